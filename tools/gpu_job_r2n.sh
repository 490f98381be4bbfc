#!/bin/bash
# wave-specialised GRU kernels (gru_cluster4.h): parity, bit identity with the v2 layout, bench A/B
OUT=gpurun_out/r2n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_gru_eigh.py -m gpu -x -q -k "gru and not wide and not miopen and not segments" --durations=3 > $OUT/pytest_gru.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gru.log
tail -4 $OUT/pytest_gru.log
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_data.py -m gpu -x -q -k "oracle_parity or golden or determin or train" --durations=3 > $OUT/pytest_par.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_par.log
tail -3 $OUT/pytest_par.log
bench() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"
}
STEMGNN_HIP_LIB=$PWD/build/ab/lib_prof.so timeout 200 python tools/gru_phase_prof.py 2>&1 | grep "gru " | tail -32 | sort | tee $OUT/gru_phase4.log
bench v4 A=1
bench v4off STEMGNN_GRU_V4=0
for v in $(ls build/ab/ | grep "^lib_v4"); do bench ${v%.so} STEMGNN_HIP_LIB=$PWD/build/ab/$v; done
bench v4b A=1
