#!/bin/bash
OUT=gpurun_out/r2h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_gru_eigh.py tests/test_hip_parity.py -m gpu -x -q -k "(eigh and not 1024 and not 2048) or eig_route or gru_fwd_bwd or time_segments or oracle_parity or dropout_matches or direct_grad" --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 200 python tools/eig_time.py > $OUT/eig_time.log 2>&1; cat $OUT/eig_time.log
for v in 1 0; do
  STEMGNN_GRU_FAST_XCD=$v timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_fastxcd$v.json 2>/dev/null
  python -c "
import json; print('fast_xcd $v: ms/step %.4f' % json.load(open('$OUT/bench_fastxcd$v.json'))['ms_per_step'])"
  STEMGNN_GRU_FAST_XCD=$v timeout 200 python tools/gru_wide_time.py 2>&1 | grep "N=228\|N=358" | sed "s/^/fast_xcd $v: /"
done
