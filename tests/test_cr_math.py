"""cr_math.h on the CPU: the round-once exp/log and the fused log_add evaluation of the CTC kernel
against the host libm (what the reference's Float=float path calls, tensor.h:78-89)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cr_math_against_libm():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "cr_math_check")
        subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cr_math_check.cc"), "-lm"])
        out = subprocess.check_output([exe, "1500000", "20260924"], text=True).split()
    n, m_exp, m_log, m_two, m_fused, big_two, big_fused, fused_not_cr, corners = map(int, out)
    # glibc's expf/logf are correctly rounded in all but a fraction of a percent of arguments
    assert m_exp < 2e-3 * n and m_log < 3e-3 * n
    # the fused log_add is the correctly rounded ln(fl(fl(exp d) + 1)) ...
    assert fused_not_cr == 0 and corners == 0
    # ... and therefore at least as close to the libm composition as two correctly rounded calls
    assert m_fused <= m_two + 10 and big_fused <= big_two + 2
