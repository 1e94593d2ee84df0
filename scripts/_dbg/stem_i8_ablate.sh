#!/bin/bash
# GPU box: the int8 stem alone (scripts/_dbg/stem_i8_probe.py) on the product build and on every other build present in lib/
# (libfq_hip_stem_*.so -- e.g. `make OUT=../lib/libfq_hip_stem_x.so VARIANT=_x EXTRA=-D...` of an experiment), alternating three times
cd /root/repo
L=pytorch-quantity_amd/lib
for rep in 1 2 3; do
  for v in libfq_hip.so $(ls $L | grep "libfq_hip_stem"); do python scripts/ab_lib.py $L/$v scripts/_dbg/stem_i8_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done
