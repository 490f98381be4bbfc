#!/bin/bash
# step timelines (rocprofv3 kernel trace of in-step launches) for several environment settings
#   bash tools/timeline_ab.sh <tag> "<ENV=.. ENV=..>" "<...>" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  ( cd /tmp && env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$i -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline > $OUT/p$i.log 2>&1 )
  python $R/tools/step_timeline.py $OUT/p$i > $OUT/timeline_$i.txt 2>&1
  echo "=== [$i] $cfg : $(head -1 $OUT/timeline_$i.txt)   $(grep -o '"ms_per_step": [0-9.]*' $OUT/p$i.log | head -1)"
  awk 'NR>1 && $1+0 > '${FROM:-900}'' $OUT/timeline_$i.txt | cut -c1-100
  rm -rf $OUT/p$i
done
