#!/bin/bash
# A/B of GRU build variants (stemgnn_amd/libstemgnn_hip_<tag>.so from tools/build_variant.sh) with tests/helpers/gru_probe.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
for shape in "32 12 228" "32 12 358" "32 12 140"; do
  for tag in "$@"; do
    lib=$R/stemgnn_amd/libstemgnn_hip${tag:+_$tag}.so
    [ "$tag" = base ] && lib=$R/stemgnn_amd/libstemgnn_hip.so
    STEMGNN_HIP_LIB=$lib python tests/helpers/gru_probe.py $shape 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%-28s %s  fwd %7.1f us  bwd %7.1f us  h %s' % (d['lib'], '$shape', d['fwd_us'], d['bwd_us'], d['h'][:8]))
"
  done
done
