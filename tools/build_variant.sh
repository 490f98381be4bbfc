#!/bin/bash
# Build a VARIANT of the library next to the product one (A/B and ablation runs; selected with STEMGNN_HIP_LIB):
#     bash tools/build_variant.sh <tag> [extra hipcc flags ...]     ->  stemgnn_amd/libstemgnn_hip_<tag>.so
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p $TMP/stemgnn_amd $TMP/include
cp -r $ROOT/stemgnn_amd/csrc $TMP/stemgnn_amd/csrc
cp $ROOT/include/*.h $TMP/include/
rm -f $TMP/stemgnn_amd/csrc/*.o
make -C $TMP/stemgnn_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function $*" > $TMP/build.log 2>&1 || { tail -20 $TMP/build.log; exit 1; }
cp $TMP/stemgnn_amd/libstemgnn_hip.so $ROOT/stemgnn_amd/libstemgnn_hip_$TAG.so
rm -rf $TMP
echo "built stemgnn_amd/libstemgnn_hip_$TAG.so"
