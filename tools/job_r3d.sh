K="six_workgroup or gru_fwd_bwd_vs_torch or hand_tuned or oracle_parity_fwd_bwd or reproduces_reference_run or two_rank or full_size_properties or large_config"
B4="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline"
bash tools/gpu_job.sh r3d "run=timeout 1300 python -m pytest tests -m gpu -q -s -k \"$K\" --durations=6 2>&1 | grep -v '^$' | tail -40" "bench=--no-cpu-baseline" prof \
  "run=STEMGNN_BENCH_WORKLOAD=358,12,3,5,32 bash tools/gpu_job.sh r3d_358 prof"
