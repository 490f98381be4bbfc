#!/bin/bash
# phase counters of the cluster GRU + broadcast-split (NR) variants built ahead under build/ab/
OUT=gpurun_out/r2m; mkdir -p $OUT
STEMGNN_HIP_LIB=$PWD/build/ab/lib_prof.so timeout 300 python tools/gru_phase_prof.py > $OUT/gru_phase.log 2>&1; echo "rc=$?" >> $OUT/gru_phase.log
tail -12 $OUT/gru_phase.log
bench() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"
}
bench base A=1
for v in f8b8 f8b24 f8b32 f0b16 f16b16 f8b0; do bench $v STEMGNN_HIP_LIB=$PWD/build/ab/lib_$v.so; done
bench base2 A=1
