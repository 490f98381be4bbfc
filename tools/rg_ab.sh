run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; python tools/bench_brief.py /tmp/b.json; }
run STEMGNN_RG=1
run STEMGNN_RG=0
