#!/bin/bash
# Phase ablation of the sg_gemm2 GLU kernels (needs a library built with  make -C stemgnn_amd/csrc CXXFLAGS+=-DSG_G2_DEBUG).
# bits: 1 = no epilogue, 2 = no MFMA, 4 = no global loads inside the K loop.  Results are WRONG by design: timing only.
for d in 0 1 2 3 4 7; do echo "== dbg $d"; STEMGNN_G2_DEBUG=$d python tools/block_time.py 2>&1 | grep "spectral_glu"; done
