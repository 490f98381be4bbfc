#!/bin/bash
OUT=gpurun_out/r2k; mkdir -p $OUT
STEMGNN_G2_BK32=7 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "oracle_parity or golden or direct_grad or stock_block or odd" --durations=3 > $OUT/pytest_bk32.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_bk32.log
tail -3 $OUT/pytest_bk32.log
for cfg in "0 32" "4 32" "4 16" "4 8" "1 32" "2 32" "7 32" "7 16"; do set -- $cfg
  STEMGNN_G2_BK32=$1 STEMGNN_NSPLIT=$2 timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_bk$1_ns$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/bench_bk$1_ns$2.json"))
print("bk32 mask $1 nsplit $2: ms/step %.4f"%d["ms_per_step"], {k:round(v["sum_us_per_step"],1) for k,v in d["roofline_families"].items()})
PY
done
