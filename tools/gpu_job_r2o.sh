#!/bin/bash
# A/B of build/ab/lib_*.so variants (bench only)
OUT=gpurun_out/r2o; mkdir -p $OUT
bench() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"
}
bench base A=1
for v in $(ls build/ab/ | grep "^lib_v4"); do bench ${v%.so} STEMGNN_HIP_LIB=$PWD/build/ab/$v; done
bench base2 A=1
if [ -f build/ab/lib_prof.so ]; then STEMGNN_HIP_LIB=$PWD/build/ab/lib_prof.so timeout 200 python tools/gru_phase_prof.py 2>&1 | grep "gru " | tail -32 | sort | tee $OUT/gru_phase.log; fi
