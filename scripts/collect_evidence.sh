#!/bin/bash
# GPU box: everything the round's DESIGN.md / profiles/ numbers come from, in one gpurun call.
# usage: scripts/collect_evidence.sh <tag>          (writes gpurun_out/ev/ and gpurun_out/prof_<tag>*)
set -u
TAG=${1:-rXX}
mkdir -p gpurun_out/ev
scripts/profile_recon.sh ${TAG}_res 128 resident > gpurun_out/ev/recon_resident.log 2>&1
scripts/profile_recon.sh ${TAG}_res256 256 resident > gpurun_out/ev/recon_resident_256.log 2>&1
scripts/profile_recon.sh ${TAG}_plain 128 plain > gpurun_out/ev/recon_plain.log 2>&1
python scripts/conv_bench.py 128 i8 > gpurun_out/ev/conv_bench_i8.txt 2>&1
python scripts/conv_bench.py 128 > gpurun_out/ev/conv_bench_f32.txt 2>&1
scripts/profile_conv_pmc.sh ${TAG}_l2 256 14 256 3 1 1 i8 > gpurun_out/ev/conv_pmc_256_14_256_3x3.txt 2>&1
scripts/profile_conv_pmc.sh ${TAG}_l1 64 56 256 1 1 0 i8 > gpurun_out/ev/conv_pmc_64_56_256_1x1.txt 2>&1
mkdir -p scripts/_bin
[ -x scripts/_bin/l2_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/l2_probe scripts/l2_probe.hip 2>/dev/null
[ -x scripts/_bin/hbm_read_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/hbm_read_probe scripts/hbm_read_probe.hip 2>/dev/null
scripts/_bin/hbm_read_probe > gpurun_out/ev/hbm_read_probe.txt 2>&1
{ python scripts/microbench.py 128; python scripts/microbench.py 128 --real; python scripts/microbench.py 128 --single; python scripts/microbench.py 128 --rotate 6; } 2>&1 | grep -v amdgpu.ids > gpurun_out/ev/microbench.txt
for b in 256 128 32 16 1; do python scripts/recon_probe.py $b 30 resident graph 2>&1 | tail -2; done > gpurun_out/ev/recon_graph.txt
python scripts/per_channel_probe.py 8 128 --phases 2>&1 | tail -2 > gpurun_out/ev/per_channel.txt
scripts/profile_round.sh $TAG > gpurun_out/ev/profile_round.log 2>&1
python bench.py > gpurun_out/ev/bench.json 2> gpurun_out/ev/bench.err
tail -3 gpurun_out/ev/recon_graph.txt; tail -c 600 gpurun_out/ev/bench.json
