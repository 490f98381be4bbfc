#!/bin/bash
# Build (here, cross-compiled) and run (on the GPU box, through gpurun) the stand-alone timing probes:
#     bash tools/probe/run_probe.sh build            # hipcc -> tools/probe/build/ (git-ignored, travels with gpurun)
#     gpurun -- 'bash tools/probe/run_probe.sh run'
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd "$(dirname "$0")"
if [ "$1" = build ]; then
  mkdir -p build
  for f in wg_probe g2_probe g2_probe96 glu_fused_probe; do
    [ -f $f.hip ] && $HIPCC -O3 -std=c++17 --offload-arch=gfx950 -I ../../stemgnn_amd/csrc $f.hip -o build/$f
  done
else
  for f in build/*; do [ -x $f ] && echo "== $f" && timeout 60 $f; done
fi
