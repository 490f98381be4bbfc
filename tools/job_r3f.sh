K="oracle_parity_fwd_bwd or exported_mask or gru_fwd_bwd_vs_torch or wave_specialised or owner_is_bit or full_size_properties or reproduces_reference_run"
BB="python bench.py --no-cpu-baseline --no-other-configs --no-roofline"
bash tools/gpu_job.sh r3f "run=timeout 900 python -m pytest tests -m gpu -q -k \"$K\" --durations=4 2>&1 | grep -v '^$' | tail -8" prof \
  "run=for v in STEMGNN_X=1 STEMGNN_GFT_FIRST=0 STEMGNN_GRU_GI_STREAM=0 STEMGNN_X=2 STEMGNN_GFT_FIRST=0 STEMGNN_EARLY_FORK=0; do echo \$v; env \$v $BB | python tools/bench_brief.py /dev/stdin | head -1; done"
