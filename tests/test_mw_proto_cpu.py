"""The helper-wave protocol of the LSD core -- the cluster form of csrc/lsd_cluster.h (mode 1 of the model: monotonic map, check (b) alone) and the multi-wave form it came
from (rounds 2-4, removed in round 5: checks (b) and (c); docs/history/DESIGN_rounds_1-4.md 5c, 5e) -- as a CPU model with real threads and real races
(tests/sim/mw_proto.cpp, on top of the oracle's LSD): one main thread replaying flsd() in order as the only writer of the used-map, helper
threads running the per-seed body ahead on a read-only view + private marks, results taken after the two checks the kernel makes -- (b) every
accepted point still unused, (c) no refine released pixels near the result since its sample.  Compared with the sequential run seed by seed
(position, region size, emitted, the rectangle's doubles) and on the final used-map.  Threads yield at random pixel reads (`chaos`), and the main
thread runs a share of the seeds itself regardless (more refines of its own -> more releases) so that views go stale in every possible way.
With either check switched off the comparison fails: the test can see a broken protocol."""
import os, subprocess, sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from synth import synth_frame


@pytest.fixture(scope="module")
def proto(tmp_path_factory):
    d = tmp_path_factory.mktemp("mw_proto")
    exe = str(d / "mw_proto")
    root = os.path.dirname(HERE)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-I" + os.path.join(root, "oracle"),
                           os.path.join(HERE, "sim", "mw_proto.cpp"), "-o", exe])
    frames = []
    for k, (seed, w, h) in enumerate([(2000, 640, 480), (77, 400, 300), (1235, 800, 600)]):
        p = str(d / ("f%d.raw" % k)); synth_frame(seed, w=w, h=h).tofile(p); frames.append((p, w, h))
    return exe, frames


def _run(exe, frame, helpers, reps, chaos, checks=3, own=0, mode=0, stale=0):
    p, w, h = frame
    r = subprocess.run([exe, p, str(w), str(h), str(helpers), str(reps), str(chaos), str(checks), str(own), str(mode), str(stale)], capture_output=True, text=True, timeout=600)
    last = r.stdout.strip().splitlines()[-1]          # "different runs: X of N"
    return int(last.split()[2]), r.stdout


@pytest.mark.parametrize("helpers,chaos,own", [(1, 0, 0), (4, 0, 0), (6, 10, 0), (6, 10, 40), (12, 3, 60), (3, 50, 20)])
def test_multiwave_protocol_equals_sequential(proto, helpers, chaos, own):
    exe, frames = proto
    for f in frames:
        bad, out = _run(exe, f, helpers, 6, chaos, 3, own)
        assert bad == 0, out


def test_multiwave_protocol_checks_are_necessary(proto):
    """negative controls: without (b) nothing holds; without (c) the runs in which the main thread refines a lot differ"""
    exe, frames = proto
    bad_none = sum(_run(exe, f, 6, 4, 10, 0, 0)[0] for f in frames)
    bad_only_c = sum(_run(exe, f, 6, 4, 10, 2, 0)[0] for f in frames)
    assert bad_none > 0 and bad_only_c > 0, (bad_none, bad_only_c)
    bad_only_b = 0; runs = 0
    for attempt in range(4):            # (c) matters only when a release races a helper's view: scheduling-dependent, so a few rounds
        bad_only_b += sum(_run(exe, f, 6, 10, 10, 1, 40)[0] for f in frames); runs += 10 * len(frames)
        if bad_only_b:
            break
    assert bad_only_b > 0, "check (c) never mattered in %d runs: the model does not exercise releases" % runs


# ---- the cluster form (csrc/lsd_cluster.h): helpers on other compute units see the map through caches of any age; the main
# ---- thread never releases a pixel in the map (private growth + one commit), and check (b) alone validates a result
@pytest.mark.parametrize("helpers,chaos,own,stale", [(4, 0, 0, 50), (6, 10, 30, 400), (12, 3, 60, 100000), (3, 50, 20, 7), (6, 10, 0, 0)])
def test_cluster_protocol_equals_sequential(proto, helpers, chaos, own, stale):
    exe, frames = proto
    for f in frames:
        bad, out = _run(exe, f, helpers, 6, chaos, 3, own, 1, stale)
        assert bad == 0, out


def test_cluster_protocol_needs_check_b_and_the_monotonic_map(proto):
    """negative controls: (1) without check (b) the cluster protocol differs from the sequential run; (2) the multi-wave protocol -- releases in
    the map, checks (b) + (c) -- differs on stale views (a helper that cached a line while a region of the main thread was marked keeps seeing
    it USED after refine() released it, and no release event after its sample tells): the reason the cluster form keeps the map monotonic"""
    exe, frames = proto
    assert sum(_run(exe, f, 6, 4, 10, 0, 30, 1, 50)[0] for f in frames) > 0
    bad = 0; runs = 0
    for attempt in range(16):           # scheduling-dependent (about two rounds in three show it): a few rounds
        bad += _run(exe, frames[0], 12, 10, 3, 3, 80, 0, 100000)[0]; runs += 10
        if bad:
            break
    assert bad > 0, "the multi-wave protocol never differed on stale views in %d runs" % runs
