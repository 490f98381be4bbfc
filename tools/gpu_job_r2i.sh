#!/bin/bash
OUT=gpurun_out/r2i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_gru_eigh.py tests/test_hip_parity.py -m gpu -x -q -k "(eigh and not 1024 and not 2048) or gru_fwd_bwd or time_segments or oracle_parity or direct_grad" --durations=4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for lib in "" plainbar nosleep; do
  if [ -n "$lib" ]; then export STEMGNN_HIP_LIB=$GRAFT_REPO_ROOT/stemgnn_amd/libstemgnn_hip_$lib.so; else unset STEMGNN_HIP_LIB; fi
  timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_$lib.json 2>/dev/null
  python -c "
import json; print('lib [$lib]: ms/step %.4f' % json.load(open('$OUT/bench_$lib.json'))['ms_per_step'])"
  timeout 200 python tools/gru_wide_time.py 2>&1 | grep "N=228" | sed "s/^/lib [$lib]: /"
done
unset STEMGNN_HIP_LIB
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_eig -- python $GRAFT_REPO_ROOT/tools/eig_time.py > $GRAFT_REPO_ROOT/$OUT/prof_eig.log 2>&1 )
grep "N=" $OUT/prof_eig.log
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2i/prof_eig/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"].split("(")[0].replace("void ","")[:26],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if any(k in r["Kernel_Name"] for k in ("eig_tridiag","eig_bisect","eig_invit","eig_backtransform","EigRebuild"))]
seen=set()
for i in range(0,len(seq)-4):
    if seq[i][0].startswith("eig_tridiag"):
        key=tuple(round(x[1],-1) for x in seq[i:i+5])
        line=" | ".join(f"{n} {d:8.1f}" for n,d in seq[i:i+5])
        if key not in seen: seen.add(key); print(line)
PY
find $OUT/prof_eig -name '*kernel_trace.csv' -size +5M -delete
