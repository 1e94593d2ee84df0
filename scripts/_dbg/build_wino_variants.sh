#!/bin/bash
# build container: libfq_hip_wa<bits>.so for each ablation value given -- only fq_conv_wino_f32 is recompiled, the other objects are the product's
set -e
cd "$(dirname "$0")/../../pytorch-quantity_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-result"
OTHERS=$(ls build/obj/*.o | grep -v fq_conv_wino_f32)
for b in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS $WINO_EXTRA -DFQ_WINO_ABLATE=$b -c -o build/wino_wa$b.o fq_conv_wino_f32.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../lib/libfq_hip_wa$b.so $OTHERS build/wino_wa$b.o ) &
done
wait
