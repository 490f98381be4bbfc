#!/bin/bash
OUT=gpurun_out/r2f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_gru_eigh.py -m gpu -x -q -k "eigh or eig_route" --durations=8 > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?" >> $OUT/pytest1.log
tail -3 $OUT/pytest1.log
timeout 300 python tools/eig_time.py > $OUT/eig_time.log 2>&1; cat $OUT/eig_time.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_eig -- python $GRAFT_REPO_ROOT/tools/eig_time.py > $GRAFT_REPO_ROOT/$OUT/prof_eig.log 2>&1 )
F=$(find $OUT/prof_eig -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && python tools/kstats.py $F 1 40 > $OUT/eig_kernel_stats.txt; grep -i "eig_\|Eig\|Cheb" $OUT/eig_kernel_stats.txt
find $OUT/prof_eig -name '*kernel_trace.csv' -size +20M -delete
STEMGNN_GRU_WG_BM64=1 timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_wgbm64.json 2>/dev/null; python -c "
import json; print('dW_hh BM64: ms/step %.4f' % json.load(open('$OUT/bench_wgbm64.json'))['ms_per_step'])"
timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_base.json 2>/dev/null; python -c "
import json; print('base: ms/step %.4f' % json.load(open('$OUT/bench_base.json'))['ms_per_step'])"
