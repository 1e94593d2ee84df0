# GPU box: the random-topology fuzzers at evidence size (scripts/model_fuzz.py, scripts/recon_fuzz.py)
cd /root/repo
mkdir -p gpurun_out/r05fz
{
for s in 41 42 43; do python scripts/model_fuzz.py 300 $s; done
python scripts/model_fuzz.py 300 44 odd
python scripts/model_fuzz.py 200 45 channels
python scripts/model_fuzz.py 100 46 channels odd
for s in 47 48 49; do python scripts/recon_fuzz.py 300 $s; done
python scripts/recon_fuzz.py 300 50 odd
} 2>&1 | grep -v "amdgpu.ids\|max_img_num\|MIOpen" > gpurun_out/r05fz/fuzz.txt
tail -20 gpurun_out/r05fz/fuzz.txt
