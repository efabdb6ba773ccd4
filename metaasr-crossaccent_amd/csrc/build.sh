#!/bin/bash
# Build libmasr.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value"
pids=()
for f in gemm conv attention rowops optim fold data ctc decode fbank pitch lstm lstm_rec blstm comm engine; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ folds.h -nt build/$f.o ] || [ ../../include/masr.h -nt build/$f.o ] || [ ../../include/masr_test.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmasr.so build/gemm.o build/conv.o build/attention.o build/rowops.o build/optim.o build/fold.o build/data.o build/ctc.o build/decode.o build/fbank.o build/pitch.o build/lstm.o build/lstm_rec.o build/blstm.o build/comm.o build/engine.o -ldl
echo "built $OUT/libmasr.so"
