#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/gpu_job.sh <tag> [pytest -k expr]
# runs the GPU test tier, a bench line and a kernel-trace profile; everything lands under gpurun_out/<tag>/
TAG=${1:-job}; KEXPR=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$KEXPR" --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
else
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
fi
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 3000 $OUT/bench.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1 )
F=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && python tools/kstats.py $F 46 60 > $OUT/kernel_stats.txt
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
