set -u
mkdir -p gpurun_out/ev
scripts/profile_recon.sh ev_res 128 resident > gpurun_out/ev/recon_resident.log 2>&1
scripts/profile_recon.sh ev_plain 128 plain > gpurun_out/ev/recon_plain.log 2>&1
python scripts/conv_bench.py 128 i8 > gpurun_out/ev/conv_bench_i8.txt 2>&1
python scripts/conv_bench.py 128 > gpurun_out/ev/conv_bench_f32.txt 2>&1
scripts/profile_conv_pmc.sh ev_l2 256 14 256 3 1 1 i8 > gpurun_out/ev/conv_pmc_256_14_256_3x3.txt 2>&1
scripts/profile_conv_pmc.sh ev_l1 64 56 256 1 1 0 i8 > gpurun_out/ev/conv_pmc_64_56_256_1x1.txt 2>&1
# conv_trace.py needs the debug library: make -C pytorch-quantity_amd/csrc ../lib/libfq_hip_trace.so
FQ_CONV_DMA=1 python scripts/conv_trace.py 256 14 256 3 1 1 > gpurun_out/ev/conv_trace_256_14_256_3x3.txt 2>&1
FQ_CONV_DMA=1 FQ_CONV_STAGES=3 python scripts/conv_trace.py 256 14 256 3 1 1 > gpurun_out/ev/conv_trace_256_14_256_3x3_ring3.txt 2>&1
[ -x scripts/_bin/l2_probe ] || { mkdir -p scripts/_bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/l2_probe scripts/l2_probe.hip 2>/dev/null; }
scripts/_bin/l2_probe 2 256 > gpurun_out/ev/l2_probe.txt 2>&1; scripts/_bin/l2_probe 4 256 >> gpurun_out/ev/l2_probe.txt 2>&1
for b in 128 32 16 1; do python scripts/recon_probe.py $b 30 resident graph 2>&1 | tail -2; done > gpurun_out/ev/recon_graph.txt
scripts/profile_round.sh r01e > gpurun_out/ev/profile_round.log 2>&1
python bench.py > gpurun_out/ev/bench.json 2> gpurun_out/ev/bench.err
tail -3 gpurun_out/ev/recon_graph.txt; tail -c 600 gpurun_out/ev/bench.json
