"""include/fq_log.h: the fast path (table reduction + Ziv's rounding test) must return exactly the bits of the
double-double evaluation fq_log_dd -- the log the KL sweep and the oracle have always used -- on every argument;
tests/fq_log_check.c compares them on random bit patterns, on the range the KL sweep lives in, around 1 (where
log cancels), at every table boundary and on the special values.

Error budget of the fast path (why 2^-64 is a safe test radius), for x = 2^k z, r = z*invc - 1 = r_hi + r_lo (exact):
  * k ln2 + log(1/invc): three-part ln2 (error < 2^-110 |k|) and a double-double table entry (< 2^-107);
  * -r_hi^2/2 from an exact two-product; r_hi^3 (1/3 - ... - r^7/10) in double: |value| < 2^-22.6, rounding error
    < 2^-75; truncation r^11/11 < 2^-80; first-order r_lo term exact to 2^-130;
  * every partial sum keeps its low word (two-sums); the low words are added in double: errors < 2^-88.
  Absolute error < 2^-74; |log x| >= 2^-8.4 outside the two slices that touch 1 -> relative < 2^-65.6; inside them
  every term scales with r: relative < 2^-53 r^2/3 < 2^-68.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fq_log") / "fq_log_check")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-o", exe,
                           os.path.join(ROOT, "tests", "fq_log_check.c"), "-lm"])
    return exe


def test_fast_log_equals_double_double_log(harness):
    out = subprocess.run([harness, "1500000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    last = out.stdout.strip().splitlines()[-1].split()
    checked, mismatches, fallbacks = int(last[1]), int(last[3]), int(last[5])
    assert checked > 7_000_000 and mismatches == 0
    assert fallbacks < 0.01 * checked, (fallbacks, checked)          # the cheap path decides > 99 % of the time


def test_table_is_reproducible():
    """include/fq_log_table.h is what scripts/gen_fq_log_table.py writes (90-digit decimals, deterministic)."""
    path = os.path.join(ROOT, "include", "fq_log_table.h")
    generated = subprocess.check_output(["python3", os.path.join(ROOT, "scripts", "gen_fq_log_table.py"), "-"], cwd=ROOT)
    assert generated.decode() == open(path).read()
