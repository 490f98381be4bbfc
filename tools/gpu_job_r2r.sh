#!/bin/bash
OUT=gpurun_out/r2r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_gru_eigh.py -m gpu -x -q -k "test_gru_fwd_bwd_vs_torch_cpu or bit_identical or per_owner" --durations=3 > $OUT/pytest_gru.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gru.log
tail -3 $OUT/pytest_gru.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("headline ms/step %.4f"%d["ms_per_step"])
for o in d["other_configs"]: print(o["config"], "%.3f ms"%o["ms_per_step"])
PY
STEMGNN_GRU_V4=0 STEMGNN_BENCH_ONLY=2 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_v4off.json 2> $OUT/bench_v4off.err
python - <<PY
import json
d=json.load(open("$OUT/bench_v4off.json"))
print("V4=0: headline ms/step %.4f"%d["ms_per_step"])
for o in d["other_configs"]: print(o["config"], "%.3f ms"%o["ms_per_step"])
PY
