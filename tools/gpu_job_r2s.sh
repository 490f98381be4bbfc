#!/bin/bash
# dW_ih | db_ih accumulated inside the wave-specialised backward (STEMGNN_GRU_FOLD_IH)
OUT=gpurun_out/r2s; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_gru_eigh.py -m gpu -x -q -k "test_gru_fwd_bwd_vs_torch_cpu or bit_identical" > $OUT/pytest_gru.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gru.log
tail -3 $OUT/pytest_gru.log
bench() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"
}
bench fold A=1
bench nofold STEMGNN_GRU_FOLD_IH=0
bench fold2 A=1
