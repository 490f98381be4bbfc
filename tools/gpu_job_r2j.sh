#!/bin/bash
OUT=gpurun_out/r2j; mkdir -p $OUT
export TMPDIR=/tmp
# A/B: fast gate math (alternative build) -- parity first, then speed
export STEMGNN_HIP_LIB=$GRAFT_REPO_ROOT/stemgnn_amd/libstemgnn_hip_fastgates.so
timeout 600 python -m pytest tests/test_hip_gru_eigh.py tests/test_hip_parity.py tests/test_hip_data.py -m gpu -x -q -k "gru_fwd_bwd or oracle_parity or golden or train_loop or dropout" --durations=4 > $OUT/pytest_fastgates.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fastgates.log
tail -3 $OUT/pytest_fastgates.log
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_fastgates.json 2>/dev/null
python -c "
import json; print('fastgates: ms/step %.4f' % json.load(open('$OUT/bench_fastgates.json'))['ms_per_step'])"
unset STEMGNN_HIP_LIB
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_base.json 2>/dev/null
python -c "
import json; print('base: ms/step %.4f' % json.load(open('$OUT/bench_base.json'))['ms_per_step'])"
# PMC passes (kernel-trace only, one counter per pass)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-graph > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/r02_pmc_traffic.json > $OUT/r02_pmc_fetch_write.md 2>&1
tail -3 $OUT/r02_pmc_fetch_write.md
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +30M -delete
