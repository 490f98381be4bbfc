#!/bin/bash
# One parameterised job script for the GPU box (through gpurun):
#     bash tools/gpu_job.sh <tag> <stage> [<stage> ...]
# stages (outputs under gpurun_out/<tag>/):
#   tests[=<pytest -k expr>]   the -m gpu tier (or a subset)
#   bench[=<extra args>]       the default bench line -> bench.json
#   prof                       rocprofv3 kernel trace of IN-STEP launches only (--no-roofline: no isolated timing loops)
#                              -> kernel_stats.txt (us / step) + step_timeline.txt
#   prof_iso                   kernel trace of the isolated GEMM-family loops only (tools/family_time.py)
#   pmc                        separate --pmc passes (kernel-trace only): FETCH_SIZE, WRITE_SIZE, MFMA busy / SQ busy
#   run=<command>              anything else, logged to run<i>.log
TAG=${1:-job}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH_PROF="python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline"
I=0
for STAGE in "$@"; do
  I=$((I+1))
  NAME=${STAGE%%=*}; ARG=""; [[ "$STAGE" == *=* ]] && ARG=${STAGE#*=}
  case $NAME in
    tests)
      if [ -n "$ARG" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -k "$ARG" --durations=10 > $OUT/pytest.log 2>&1
      else timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.log 2>&1; fi
      echo "pytest rc=$?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log ;;
    bench)
      timeout 900 python bench.py $ARG > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
      python tools/bench_brief.py $OUT/bench.json ;;
    prof)
      ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- $BENCH_PROF > $R/$OUT/prof.log 2>&1 )
      F=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
      [ -n "$F" ] && python tools/kstats.py $F auto 70 > $OUT/kernel_stats.txt
      python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1; head -1 $OUT/step_timeline.txt
      find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete ;;
    prof_iso)
      ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_iso -- python $R/tools/family_time.py > $R/$OUT/prof_iso.log 2>&1 )
      F=$(find $OUT/prof_iso -name '*kernel_stats.csv' | head -1)
      [ -n "$F" ] && python tools/kstats.py $F 1 30 > $OUT/kernel_stats_isolated.txt
      tail -5 $OUT/prof_iso.log; find $OUT/prof_iso -name '*kernel_trace.csv' -size +20M -delete ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
        D=pmc_$(echo $c | cut -d' ' -f1)
        ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/$D -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-graph > $R/$OUT/$D.log 2>&1 )
      done
      python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES $OUT/pmc_traffic.json > $OUT/pmc_summary.md 2>&1
      tail -4 $OUT/pmc_summary.md
      find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +30M -delete ;;
    run)
      bash -c "$ARG" > $OUT/run$I.log 2>&1; echo "run rc=$?"; tail -25 $OUT/run$I.log ;;
    *) echo "unknown stage $STAGE" ;;
  esac
done
