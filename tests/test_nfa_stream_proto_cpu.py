"""The hand-over of candidate rectangles from the cluster form's main wave to a concurrent NFA stage (the streaming NFA stage of calls of up to 64 frames: csrc/lsd_cluster.h cl_main<G, true>, csrc/lsd_nfa.h k_nfa_stream) as a CPU model with real threads (tests/sim/nfa_stream_proto.cpp): every record is
processed exactly once, by a thread that saw its final contents -- with consumers that wait as long as it takes, with consumers that give up (the launch behind the
core takes what they left), and NOT when the counter is published before the records (negative control: the test can see a broken protocol)."""
import os, subprocess
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def proto(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("nfa_stream") / "nfa_stream_proto")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(HERE, "sim", "nfa_stream_proto.cpp"), "-o", exe])
    return exe


def _bad(exe, *args):
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300).stdout
    return int(out.strip().splitlines()[-1].split()[2]), out


@pytest.mark.parametrize("runs,max_records,consumers,expire", [(300, 400, 8, 0), (300, 400, 8, 1), (100, 3000, 16, 1), (200, 40, 3, 0)])
def test_every_rectangle_is_evaluated_exactly_once(proto, runs, max_records, consumers, expire):
    bad, out = _bad(proto, runs, max_records, consumers, expire)
    assert bad == 0, out


def test_counter_before_records_is_caught(proto):
    """negative control, deterministic: the faulty producer announces record n / 2 and holds it back until a consumer has read it (tests/sim/nfa_stream_proto.cpp),
    so EVERY run with at least one record is bad -- not "some of 300" as in round 4, which failed 2 of 5 times on an 8-core box"""
    runs = 40
    bad, out = _bad(proto, runs, 400, 8, 0, 1)
    assert bad >= runs - 1, out          # (run 0 has no record)
