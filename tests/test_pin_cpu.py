"""Pinning the oracle to the reference's own source (oracle/ref_pin).

* `pin-stub` (runs HERE, no OpenCV needed): /root/reference/src/ORBextractor.cc compiled unmodified against a stub cv:: layer whose leaves are
  oracle/cvleaf.h, plus line-range slices of src/ORBmatcher.cc / src/LSDmatcher.cpp / src/Frame.cc against stand-in Frame types -- every byte
  the reference's in-repo code produces must equal the oracle's.  Where the reference tree is absent (the GPU box) the committed report
  oracle/ref_pin/pin_report_stub.json is checked instead, and it must have been made with the oracle sources that are in the tree now.
* `pin` (needs OpenCV 3.4 + contrib + Eigen: the leaves): skips cleanly here; a report, when one exists, must be clean."""
import json, os, subprocess, sys
import pytest
import pkg

REF = "/root/reference"
PIN = os.path.join(pkg.ROOT, "oracle", "ref_pin")
sys.path.insert(0, PIN)
from merge_reports import oracle_sources_sha256


def _check(rep):
    assert rep["oracle_sources_sha256"] == oracle_sources_sha256(), "the pin report was made with other oracle sources: run `make -C oracle/ref_pin pin-stub` and commit oracle/ref_pin/pin_report_stub.json"
    bad = {k: v["pinned"] for k, v in rep["orb_extractor"]["fixtures"].items() if not (v["pinned"]["kp_equal"] and v["pinned"]["desc_equal"] and v["scale_tables_equal"])}
    assert not bad, bad
    bad = {k: v for k, v in rep["slices"]["cases"].items() if not v["equal"]}
    assert not bad, bad
    assert rep["all_equal"]
    assert len(rep["orb_extractor"]["fixtures"]) >= 16 and len(rep["slices"]["cases"]) >= 239
    covered = " ".join(rep["slices"]["cases"])
    for fn in ("DescriptorDistance", "GetFeaturesInArea", "GetLinesInArea", "SearchForInitialization", "SerachForInitialize", "SearchByProjection(F, MapPoints)",
               "SearchByProjection(Cur, Last)", "SearchByProjection(F, MapLines)", "SearchByBoW(KF, F)", "SearchByBoW(KF, KF)", "ExtractLineSegment", "SearchByProjection(KF, F)", "SearchByDescriptor(KF, F)",
               "SearchByDescriptor(KF, KF)", "LSDmatcher::SearchForTriangulation", "ORBmatcher::SearchForTriangulation", "DBoW2 loadFromTextFile + transform", "ComputeDistinctiveDescriptors"):
        assert fn in covered, fn
    # the error bars of decisions D1 / D4 are part of the report (DESIGN.md section 2 quotes them)
    eb = rep["orb_extractor"]["error_bars"]
    assert set(eb) == {"ref_orb_stub_fma", "ref_orb_stub_malloc", "ref_orb_stub_libm", "ref_orb_stub_asbuilt", "ref_orb_stub_gauss340"} and all(v["keypoints"] > 10000 for v in eb.values())


def test_reference_compiled_here_equals_oracle():
    if os.path.exists(os.path.join(REF, "src", "ORBextractor.cc")):
        r = subprocess.run(["make", "-s", "-C", PIN, "pin-stub", "REF=" + REF], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        _check(json.load(open(os.path.join(pkg.ROOT, "oracle", "_ref", "pin_report_stub.json"))))
    _check(json.load(open(os.path.join(PIN, "pin_report_stub.json"))))          # the committed copy (what travels)


def test_pin_recipe_runs_or_skips_cleanly():
    r = subprocess.run(["make", "-s", "-C", PIN, "pin", "pin-stub", "REF=/nonexistent-reference-tree"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.count("SKIP") >= 3, r.stdout + r.stderr          # without the reference tree it says so and succeeds


def test_pin_report_is_clean_when_present():
    rep = os.path.join(pkg.ROOT, "oracle", "_ref", "pin_report.json")
    if not os.path.exists(rep):
        pytest.skip("OpenCV leaves UNPINNED: no machine with OpenCV 3.4 + opencv_contrib has run `make -C oracle/ref_pin pin` on this tree yet")
    r = json.load(open(rep))
    bad = {k: [a for a, v in f.items() if not v["equal"]] for k, f in r["fixtures"].items()}
    assert r["all_equal"], {k: v for k, v in bad.items() if v}
