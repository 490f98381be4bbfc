#!/bin/bash
# A/B of library build variants on the wide-cluster GRU (tools/gru_wide_time.py):  bash tools/wide_ab.sh <tag> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for tag in "$@"; do
  lib=$R/stemgnn_amd/libstemgnn_hip_$tag.so; [ "$tag" = base ] && lib=$R/stemgnn_amd/libstemgnn_hip.so
  echo "== $tag"; STEMGNN_HIP_LIB=$lib timeout 200 python tools/gru_wide_time.py 2>&1 | grep "N=1024\|N=2048"
done
