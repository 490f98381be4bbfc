#!/bin/bash
# GRU mat-vec variants: wave-per-owner forward (three gates per broadcast) and LDS-row broadcast of the polled values
OUT=gpurun_out/r2l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_gru_eigh.py -m gpu -x -q -k "gru and not wide and not miopen and not segments" --durations=3 > $OUT/pytest_gru.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gru.log
tail -3 $OUT/pytest_gru.log
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_data.py -m gpu -x -q -k "oracle_parity or golden or determin or train" --durations=3 > $OUT/pytest_par.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_par.log
tail -3 $OUT/pytest_par.log
bench() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python -c "import json;d=json.load(open('$OUT/bench_$tag.json'));print('$tag: ms/step %.4f'%d['ms_per_step'])"
}
bench new A=1
bench fwd3off STEMGNN_GRU_FWD3=0
# A/B builds of the broadcast split (no LDS rows at all = round-2 mat-vec with the new forward)
cp stemgnn_amd/libstemgnn_hip.so /tmp/lib_keep.so
cd stemgnn_amd/csrc
for v in "64 64" "8 64" "64 16"; do set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DGRU_NR_FWD=$1 -DGRU_NR_BWD=$2 -c gru.hip -o /tmp/gru_v.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC block.o front.o pack.o /tmp/gru_v.o eigh.o tail.o data.o splitgemm.o -o ../libstemgnn_hip.so
  cd ../..; bench nrf$1_nrb$2 A=1; cd stemgnn_amd/csrc
done
cd ../..
cp /tmp/lib_keep.so stemgnn_amd/libstemgnn_hip.so
