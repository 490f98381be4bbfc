K="split or oracle_parity_fwd_bwd or exported_mask or reference_golden or reproduces_reference_run or train_step_with_fused or two_rank or exact_mode or full_size_properties"
B4="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline"
bash tools/gpu_job.sh r3c "run=timeout 1300 python -m pytest tests -m gpu -q -s -k \"$K\" --durations=6 2>&1 | grep -v '^$' | tail -60" "bench=--no-cpu-baseline --no-other-configs" prof \
  "run=STEMGNN_DTYPE=bf16x3 python bench.py --no-cpu-baseline --no-other-configs | python tools/bench_brief.py /dev/stdin; STEMGNN_DTYPE=bf16x2 python bench.py --no-cpu-baseline --no-other-configs | python tools/bench_brief.py /dev/stdin" \
  "run=export STEMGNN_BENCH_WORKLOAD=2048,48,12,5,16; for v in STEMGNN_X=0 STEMGNN_WG_FUSED=0 STEMGNN_WG_CFG=16,3,2 STEMGNN_DTYPE=bf16x2 STEMGNN_DTYPE=bf16x3; do echo \$v; env \$v $B4 | python tools/bench_brief.py /dev/stdin | head -1; done"
