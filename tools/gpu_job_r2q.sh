#!/bin/bash
OUT=gpurun_out/r2q; mkdir -p $OUT
bench() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"
}
bench base A=1
for v in $(ls build/ab/ | grep "^lib_v4"); do bench ${v%.so} STEMGNN_HIP_LIB=$PWD/build/ab/$v; done
for n in 16 24 48 64; do bench nsplit$n STEMGNN_GRU_NSPLIT=$n; done
bench base2 A=1
