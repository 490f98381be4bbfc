#!/bin/bash
OUT=gpurun_out/r2t; mkdir -p $OUT
bench() { tag=$1; shift
  env "$@" timeout 100 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"; }
bench base A=1
bench nohog STEMGNN_GRU_LDS_HOG=0
bench nodefer STEMGNN_DEFER_B1=0
