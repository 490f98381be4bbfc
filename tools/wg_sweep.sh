# A/B of the weight-gradient scheduling knobs (bench lines only)
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline --no-roofline > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; python tools/bench_brief.py /tmp/b.json; }
run A=0
run STEMGNN_WG_CU0=100
run STEMGNN_WG_CU0=25
run STEMGNN_WG_CU0=50 STEMGNN_WG_CU1=50
run STEMGNN_WG_CFG=16,4,2
run STEMGNN_WG_CFG=32,3,1
run STEMGNN_GRU_WG_FUSED=0
run STEMGNN_WG_FUSED=0
