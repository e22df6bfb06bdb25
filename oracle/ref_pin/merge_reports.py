#!/usr/bin/env python3
"""Merges the two halves of `make pin-stub` into one report and stamps it with the SHA-256 of the oracle's sources, so that a committed copy
(oracle/ref_pin/pin_report_stub.json: the reference tree does not travel to the GPU box) can be tied to the oracle it was made with:
tests/test_pin_cpu.py recomputes the hash wherever it runs."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))


def oracle_sources_sha256():
    d = os.path.join(HERE, "..")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".cpp", ".h", ".inc")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    orb, sl = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    rep = {"oracle_sources_sha256": oracle_sources_sha256(), "all_equal": bool(orb["all_equal"] and sl["all_equal"]),
           "pinned_rows": "a1 a2(driver) a3 a5 a6 a8 a9 (ORBextractor.cc whole file); a11 (orchestration) a14 a15 a16 a17 a18 a19 a20 a21 (line-range slices); f2 SearchForTriangulation x2; f3 ComputeDistinctiveDescriptors x2; f4 ComputeBoW (the vendored DBoW2, whole files)",
           "d9_d10_undefined_in_the_reference": sl.get("d9_trailing_newline"),
           "d3_error_bar": sl.get("d3_error_bar"), "d2_error_bar_oracle_only": sl.get("d2_error_bar_oracle_only"), "d7_error_bar_oracle_only": sl.get("d7_error_bar_oracle_only"),
           "unpinned_leaves": "cv::FAST, cv::resize, cv::copyMakeBorder, cv::GaussianBlur, cv::fastAtan2 (oracle/cvleaf.h), cv::BFMatcher::knnMatch, cv::gemm (pose algebra), Eigen::Vector3d, "
                              "cv::line_descriptor::LSDDetector + BinaryDescriptor (oracle/lsd_oracle.cpp, lbd_oracle.cpp): UPSTREAM-RECALL",
           "orb_extractor": orb, "slices": sl}
    for out in sys.argv[3:]:
        json.dump(rep, open(out, "w"), indent=1)
    print("pin-stub:", "ALL EQUAL" if rep["all_equal"] else "DIFFERENCES", "-> ", ", ".join(sys.argv[3:]))
