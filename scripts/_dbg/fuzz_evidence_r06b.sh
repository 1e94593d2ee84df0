#!/bin/bash
# GPU box: the round-5 evidence set again on round 6's final kernels (plain / odd / share / bn / channels variants)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for args in "300 61 odd" "300 62 share" "300 63 bn" "200 64 channels" "100 65 channels odd"; do
  python scripts/model_fuzz.py $args 2>&1 | grep "model_fuzz\|model [0-9]" | tail -4
done
for args in "300 66 odd" "300 67"; do
  python scripts/recon_fuzz.py $args 2>&1 | grep "recon_fuzz\|model [0-9]" | tail -4
done
python scripts/conv_fuzz.py 1500 68 2>&1 | tail -2
python scripts/calib_fuzz.py 200 69 2>&1 | tail -2
python scripts/kl_fuzz.py 600 70 2>&1 | tail -2
python scripts/block_tail_fuzz.py 1500 71 2>&1 | tail -2
python scripts/float_conv_fuzz.py 2000 72 2>&1 | tail -2
