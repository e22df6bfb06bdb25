"""csrc/lsd_align_win.h on the CPU: the header the HIP rectangle counter (k_nfa_count) uses to turn the reference's isAligned test
(OpenCV lsd.cpp, restated at oracle/lsd_oracle.cpp:68-76) into integer intervals of fp32 angle bit patterns, compiled by g++ with the same
-ffp-contract=off and compared with the reference predicate on every angle the gradient table can hold (fastAtan2 of all integer gradients in
[-510, 510]^2), the neighbours of every interval end point and random patterns -- for thetas anywhere region2rect can put them, glued to the
0 / 2pi seams and to the pruning edges, and tolerances pi/8 * 2^-h as well as arbitrary ones below pi/2.  The GPU twin is
tests/test_lines_gpu.py::test_align_windows_selftest."""
import os, subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_align_windows_equal_reference_predicate(tmp_path):
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "align_win_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "oracle"),
                           os.path.join(HERE, "sim", "align_win_test.cpp"), "-o", exe])
    for seed in (1, 2):
        r = subprocess.run([exe, "3000", str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        last = r.stdout.strip().splitlines()[-1]          # "mismatches: 0 of N tests, max windows W, three-window cases 0"
        f = last.replace(",", "").split()
        assert int(f[1]) == 0 and int(f[3]) > 100_000_000 and int(f[7]) <= 2 and int(f[-1]) == 0, last
