for cfg in 16,4,2 16,6,1 16,4,1 16,3,2 16,3,3 32,3,1; do
  echo "== WG_CFG $cfg"
  STEMGNN_WG_CFG=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python tools/bench_brief.py /tmp/b.json
done
echo "== fused off"
STEMGNN_WG_FUSED=0 timeout 300 python bench.py --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err; python tools/bench_brief.py /tmp/b.json
