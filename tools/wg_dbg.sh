#!/bin/bash
# Phase ablation of the fused weight-gradient kernel (csrc/wgrad.h).  Needs a library built with -DSG_WG_DEBUG:
#     bash tools/build_variant.sh dbg -DSG_WG_DEBUG        (-> stemgnn_amd/libstemgnn_hip_dbg.so)
# bits: 1 = stop after the K loop (no publish / reduce / store), 2 = no MFMA, 4 = no DMA inside the K loop,
#       8 = publish the partial but never reduce.  Results are WRONG by design: timing only.
export STEMGNN_HIP_LIB=${GRAFT_REPO_ROOT:-$(pwd)}/stemgnn_amd/libstemgnn_hip_dbg.so
for cfg in 16,6,1; do
  for d in 0 8 1 3 5 7; do
    echo "== cfg $cfg dbg $d"; STEMGNN_WG_DEBUG=$d python tools/block_time.py 2>&1 | grep "spectral_glu_bwd wgrad"
  done
done
