#!/bin/bash
# GPU box: round-6 evidence for the pair / chain histograms (pass 1 leaves residual sums to pass 2): the activation-cache fuzzer on
# random topologies -- every cache size / plan must give the statistics of the calibration without a cache, bit for bit.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for args in "300 51 cache" "300 52 cache odd" "200 53 cache share" "200 54 cache bn"; do
  python scripts/model_fuzz.py $args 2>&1 | grep -v "^max_img_num\|bit:\|^Collect\|^interval\|^\[\|^Eltwise\|^Concat\|bit conv" | tail -4
done
python scripts/model_fuzz.py 300 55 2>&1 | tail -2
python scripts/recon_fuzz.py 300 56 2>&1 | tail -2
