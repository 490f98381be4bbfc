#!/bin/bash
# final artefacts of a round: GRU parity subset, the default bench line, rocprofv3 kernel stats + step timeline
TAG=${1:-fin}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_gru_eigh.py -m gpu -x -q -k "test_gru_fwd_bwd_vs_torch_cpu or bit_identical" > $OUT/pytest_gru.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gru.log
tail -2 $OUT/pytest_gru.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print('ms/step %.4f value %.0f'%(d['ms_per_step'],d['value']));[print(o['config'][:28],'%.3f'%o['ms_per_step']) for o in d['other_configs']]"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1 )
F=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && python tools/kstats.py $F 46 60 > $OUT/kernel_stats.txt
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
