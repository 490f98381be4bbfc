#!/bin/bash
OUT=gpurun_out/r2c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_data.py tests/test_hip_gru_eigh.py -m gpu -x -q -k "large_config or dropout_training or wide or rolling or train_loop" --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for cfg in "32 3" "16 3" "8 3" "8 7" "16 7" "32 7"; do set -- $cfg
  STEMGNN_NSPLIT=$1 STEMGNN_G2_BM64=$2 timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_ns$1_bm$2.json 2> $OUT/bench_ns$1_bm$2.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_ns$1_bm$2.json"))
print("nsplit $1 bm64 $2: ms/step %.4f"%d["ms_per_step"], {k:round(v["sum_us_per_step"],1) for k,v in d["roofline_families"].items()})
PY
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err
python - <<PY
import json
d=json.load(open("$OUT/bench_full.json"))
for o in d["other_configs"]: print(o.get("config"), o.get("ms_per_step"), o.get("error"))
PY
