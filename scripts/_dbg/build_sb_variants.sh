#!/bin/bash
# build container: libfq_hip_sb<bits>.so for each FQ_SB_ABLATE value given -- only fq_conv1x1_f32 is recompiled
set -e
cd "$(dirname "$0")/../../pytorch-quantity_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
OTHERS=$(ls build/obj/*.o | grep -v fq_conv1x1_f32)
for b in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DFQ_SB_ABLATE=$b -c -o build/c1_sb$b.o fq_conv1x1_f32.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../lib/libfq_hip_sb$b.so $OTHERS build/c1_sb$b.o ) &
done
wait
