# GPU box: the round's closing evidence on the committed tree -- smoke, the GPU suite (fp32 and split-bf16 kernels), the driver command
cd /root/repo
mkdir -p gpurun_out/r05z
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r05z/smoke.txt
python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/r05z/gpu_suite.txt
FQ_CONV_SPLIT_BF16=1 python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/r05z/gpu_suite_split_bf16.txt
python bench.py 2> gpurun_out/r05z/bench.err | tail -1 > gpurun_out/r05z/bench.json
cat gpurun_out/r05z/smoke.txt gpurun_out/r05z/gpu_suite.txt gpurun_out/r05z/gpu_suite_split_bf16.txt; cut -c1-400 gpurun_out/r05z/bench.json
