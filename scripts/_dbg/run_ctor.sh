# GPU box: the constructor's cost in fresh processes, then the whole GPU suite
cd /root/repo
mkdir -p gpurun_out/r05k
for i in 1 2 3; do python scripts/_dbg/ctor_probe.py 2>&1 | tail -1; done | tee gpurun_out/r05k/ctor_probe_final.txt
python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee gpurun_out/r05k/gpu_suite.txt
