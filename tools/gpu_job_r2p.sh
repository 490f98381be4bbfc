#!/bin/bash
# forward cluster size of the v4 GRU (STEMGNN_GRU_FWD_P)
OUT=gpurun_out/r2p; mkdir -p $OUT
for P in 7 5; do
STEMGNN_GRU_FWD_P=$P timeout 600 python -m pytest tests/test_hip_gru_eigh.py -m gpu -x -q -k "test_gru_fwd_bwd_vs_torch_cpu and 2] or bit_identical" > $OUT/pytest_p$P.log 2>&1; echo "pytest P=$P rc=$?"; tail -2 $OUT/pytest_p$P.log
done
bench() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"
}
bench base A=1
for P in 5 6 7; do bench fwdp$P STEMGNN_GRU_FWD_P=$P; done
bench base2 A=1
