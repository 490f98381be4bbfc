#!/bin/bash
OUT=gpurun_out/r2e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_gru_eigh.py tests/test_hip_splitgemm.py -m gpu -x -q -k "eigh or eig_route or split or time_segments" --durations=12 -s > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?" >> $OUT/pytest1.log
tail -4 $OUT/pytest1.log
timeout 300 python tools/eig_time.py > $OUT/eig_time.log 2>&1; cat $OUT/eig_time.log
timeout 300 python tools/split_gemm_experiment.py > $OUT/split_gemm.log 2>&1; cat $OUT/split_gemm.log
for cfg in "1 1" "2 1" "2 0" "4 0" "1 0"; do set -- $cfg
  STEMGNN_GRU_SEGMENTS=$1 STEMGNN_GRU_LDS_HOG=$2 timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_seg$1_hog$2.json 2> $OUT/bench_seg$1_hog$2.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_seg$1_hog$2.json"))
print("segments $1 hog $2: ms/step %.4f"%d["ms_per_step"])
PY
done
STEMGNN_GRU_TAIL_PAR=0 timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_tailpar0.json 2>/dev/null; python -c "
import json; print('tail_par 0: ms/step %.4f' % json.load(open('$OUT/bench_tailpar0.json'))['ms_per_step'])"
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "large_config and 2048" --durations=5 -s > $OUT/pytest2.log 2>&1; echo "pytest2 rc=$?" >> $OUT/pytest2.log
tail -4 $OUT/pytest2.log
