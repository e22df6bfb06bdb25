#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD container only).

1. icl_input_gray.npz — the single real 640x480 frame of the reference tree
   (/root/reference/images/input.png, an ICL-NUIM office frame) converted to 8-bit gray with
   OpenCV's fixed-point RGB2GRAY (4899 R + 9617 G + 1868 B + 8192) >> 14.  The conversion sits above
   the drop-in boundary (reference src/Tracking.cc:148-161), so this only fixes the fixture bytes.
2. oracle_golden.npz — outputs of the CPU oracle on that frame and on seeded synthetic frames.
   The reference ships no tests/golden vectors and cannot be built here (no OpenCV), so these pin the
   ORACLE against regressions; parity with the real OpenCV-backed binary stays UNPINNED.
"""
import os, sys, hashlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib
from synth import synth_frame, warp_prev


def main():
    out = {}
    ref_png = "/root/reference/images/input.png"
    if os.path.exists(ref_png):
        from PIL import Image
        rgb = np.asarray(Image.open(ref_png).convert("RGB"), dtype=np.int64)
        gray = ((4899 * rgb[..., 0] + 9617 * rgb[..., 1] + 1868 * rgb[..., 2] + 8192) >> 14).astype(np.uint8)
        assert gray.shape == (480, 640)
        np.savez_compressed(os.path.join(HERE, "icl_input_gray.npz"), gray=gray)
    gray = np.load(os.path.join(HERE, "icl_input_gray.npz"))["gray"]
    orc = oracle_lib.Oracle()
    frames = {"icl": gray, "synth1234": synth_frame(1234), "synth_small": synth_frame(4321, w=320, h=240)}
    for name, img in frames.items():
        nfeat = 500 if name == "synth_small" else 1000
        kp, desc = orc.orb_extract(img, nfeat)
        out[name + "_kp"] = kp.view(np.uint8).reshape(len(kp), 28)
        out[name + "_desc"] = desc
        kl, ld, fn, raw = orc.lines_extract(img, 40 if name == "icl" else 200)
        out[name + "_kl"] = kl.view(np.uint8).reshape(len(kl), 68)
        out[name + "_ldesc"] = ld
        out[name + "_linefn"] = fn
        out[name + "_segs"] = raw
        out[name + "_sha"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8)
    cur = frames["synth1234"]; prev = warp_prev(cur)
    kp1, d1 = orc.orb_extract(prev, 1000); kp2, d2 = orc.orb_extract(cur, 1000)
    pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
    m12, pmo, n = orc.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, True)
    out["match_m12"] = m12; out["match_n"] = np.array([n])
    l1 = orc.lines_extract(prev, 200); l2 = orc.lines_extract(cur, 200)
    pairs, mad, mad12 = orc.line_match(l1[1], l2[1], 0.5, False)
    out["lmatch_pairs"] = pairs; out["lmatch_mad"] = np.array([mad, mad12])
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
