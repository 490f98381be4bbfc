K="oracle_parity_fwd_bwd or exported_mask or reference_golden or gru_fwd_bwd_vs_torch or wave_specialised or owner_is_bit or six_workgroup or two_rank or exact_mode or full_size_properties or reproduces_reference_run or stock_block_layer or direct_grad"
BB="python bench.py --no-cpu-baseline --no-other-configs --no-roofline"
bash tools/gpu_job.sh r3e "run=timeout 900 python -m pytest tests -m gpu -q -k \"$K\" --durations=4 2>&1 | grep -v '^$' | tail -12" "bench=--no-cpu-baseline --no-other-configs" prof \
  "run=for v in STEMGNN_WG_CU0=100 STEMGNN_WG_CU0=80 STEMGNN_WG_CU0=65 STEMGNN_GFT_FIRST=0 STEMGNN_GRU_GI_STREAM=0; do echo \$v; env \$v $BB | python tools/bench_brief.py /dev/stdin | head -1; done"
