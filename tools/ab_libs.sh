#!/bin/bash
# A/B of library build variants on the headline bench:  bash tools/ab_libs.sh <tag> [<tag> ...]   (tag "base" = the product .so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for tag in "$@"; do
  lib=$R/stemgnn_amd/libstemgnn_hip_$tag.so; [ "$tag" = base ] && lib=$R/stemgnn_amd/libstemgnn_hip.so
  STEMGNN_HIP_LIB=$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-roofline > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  echo "$tag: $(python tools/bench_brief.py /tmp/b.json | head -1)"
done; done
