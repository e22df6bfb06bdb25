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



# Per stated decision of the UNPINNED half: which observable of the author's binary (libStructureSLAM.so against OpenCV 3.4 + contrib) settles it, so that the first person
# with that build needs minutes, not a day.  "here" = what this repo's default produces on the fixture named; sizes from the error bars in this report / INTEGRATION.md section 6.
SETTLING_OBSERVABLES = {
    "D11_nfa_first_term": {"observable": "number of segments LSDDetector::detect returns on images/input.png (gray, 640x480) BEFORE the top-N cut of ExtractLineSegment.cpp:44-51",
                           "default_predicts": "3 694-scale counts over the 7 line fixtures (2.0-2.7 x the other form); tests/golden/oracle_golden.npz holds the default's segments", "other_form_predicts": "1 606-scale counts",
                           "flip": "sslam_lines_set_nfa_variant(ln, 0) / SSLAM_LSD_NFA_VARIANT=0; oracle: orc_set_nfa_variant(0)"},
    "D12_lbd_bit_order": {"observable": "byte 0 of any row of mLdesc against the same row computed with the other order: the two are bit-reversals of each other (0x80 >> i against 1 << i)",
                          "default_predicts": "comparison 0 of a row lands in bit 7", "other_form_predicts": "comparison 0 lands in bit 0",
                          "flip": "sslam_lines_set_lbd_bit_order(ln, 0) / SSLAM_LBD_BIT_ORDER=0; oracle: orc_set_lbd_bit_order(0)"},
    "D7_lsd_rescale": {"observable": "the 512x384 image LSD works on (dump `scaled_image` in lsd.cpp, or compare segment endpoints: every segment changes some endpoint bit under the other form)",
                       "default_predicts": "INTER_LINEAR_EXACT bytes (q8 coefficients, one rounding): oracle/cvleaf.h resize_linear_exact", "other_form_predicts": "INTER_LINEAR bytes (11-bit coefficient pairs, two-stage rounding)",
                       "flip": "sslam_lines_set_resize_variant(ln, 1) / SSLAM_LSD_RESIZE_VARIANT=1"},
    "D2_seed_order": {"observable": "segments of a frame with many equal-gradient pixels (synthetic fixtures): 184 of 1 606 differ between the stable order and libstdc++'s std::sort", "default_predicts": "raster order inside a bin",
                      "other_form_predicts": "std::sort's permutation (depends on the libstdc++ the author linked)", "flip": "sslam_lines_set_seed_order(ln, 1) / SSLAM_LSD_SEED_ORDER=1"},
    "D6_gaussian_blur_8u": {"observable": "cv::GaussianBlur(img, 7x7, sigma 2) of any 8-bit image against oracle/cvleaf.h gaussian_blur_8u: 3.7 % of rBRIEF bits follow",
                            "default_predicts": "OpenCV >= 3.4.1 fixed-point taps", "other_form_predicts": "3.4.0 float taps x 256, each rounded", "flip": "sslam_orb_set_blur_variant / sslam_lines_set_blur_variant (1)"},
    "LSD_output_offset": {"observable": "fractional part of segment endpoints of an axis-aligned step edge: x.625 / x.0 patterns", "default_predicts": "(coordinate + 0.5) / 0.8 (oracle/lsd_oracle.cpp:404; csrc/lsd_nfa.h k_nfa_finish)",
                          "other_form_predicts": "coordinate / 0.8 (von Gioi's lsd.c without the half-pixel shift)", "flip": "one constant on each side (the two lines named); no switch"},
    "LBD_pre_blur": {"observable": "LBD bytes of a frame with and without a 5x5 sigma-1 GaussianBlur in front of the Sobel (BinaryDescriptor::computeGaussianPyramid at octave 0)",
                     "default_predicts": "blur applied (oracle/lbd_oracle.cpp; csrc/lbd.h k_blur_sobel)", "other_form_predicts": "Sobel of the raw image", "flip": "taps {0,0,256,0,0} in lines_build_plan (csrc/lines.hip, t5) and lbd_oracle.cpp's blur call"},
    "KeyLine_numOfPixels": {"observable": "KeyLine::numOfPixels of a diagonal line: max(|dx|, |dy|) + 1 (cv::LineIterator, 8-connected) against a Bresenham / Euclidean count",
                            "default_predicts": "Chebyshev count (oracle/lsd_oracle.cpp:440)", "other_form_predicts": "another count: numOfPixels only, no descriptor bit depends on it", "flip": "the one expression named, both sides (csrc/lbd.h k_keylines)"},
}


if __name__ == "__main__":
    orb, sl = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    rep = {"oracle_sources_sha256": oracle_sources_sha256(), "all_equal": bool(orb["all_equal"] and sl["all_equal"]),
           "pinned_rows": "a1 a2(driver) a3 a5 a6 a8 a9 (ORBextractor.cc whole file); a11 (orchestration) a14 a15 a16 a17 a18 a19 a20 a21 (line-range slices); f2 SearchForTriangulation x2; f3 ComputeDistinctiveDescriptors x2; f4 ComputeBoW (the vendored DBoW2, whole files)",
           "d9_d10_undefined_in_the_reference": sl.get("d9_trailing_newline"),
           "d3_error_bar": sl.get("d3_error_bar"), "d2_error_bar_oracle_only": sl.get("d2_error_bar_oracle_only"), "d7_error_bar_oracle_only": sl.get("d7_error_bar_oracle_only"),
           "unpinned_leaves": "cv::FAST, cv::resize, cv::copyMakeBorder, cv::GaussianBlur, cv::fastAtan2 (oracle/cvleaf.h), cv::BFMatcher::knnMatch, cv::gemm (pose algebra), Eigen::Vector3d, "
                              "cv::line_descriptor::LSDDetector + BinaryDescriptor (oracle/lsd_oracle.cpp, lbd_oracle.cpp): UPSTREAM-RECALL",
           "settling_observables": SETTLING_OBSERVABLES, "orb_extractor": orb, "slices": sl}
    for out in sys.argv[3:]:
        json.dump(rep, open(out, "w"), indent=1)
    print("pin-stub:", "ALL EQUAL" if rep["all_equal"] else "DIFFERENCES", "-> ", ", ".join(sys.argv[3:]))
