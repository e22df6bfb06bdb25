#!/opt/conda/bin/python3.9
"""Independent cross-check fixtures from scikit-image 0.18.3 (present only in the BUILD container's conda python:
`/opt/conda/bin/python3.9 tests/golden/make_skimage_fixtures.py`).  scikit-image implements the published FAST segment test and
the ORB intensity-centroid orientation on its own; neither its scoring nor its sampling is OpenCV's, so only what is
definitionally common is stored:

  fast_mask_t{20,7}  pixels with >= 9 contiguous ring pixels all brighter than v+t or all darker than v-t (corner_fast(img, 9, t) > 0)
  fast_maxth         largest threshold at which the pixel is still such a corner (0 if not a corner at 7): this IS OpenCV's
                     cornerScore by definition (fast_score.cpp: "the maximum threshold for which the pixel remains a corner")
  orient_xy / orient_deg   intensity-centroid angle atan2(m01, m10) over the radius-15 disc (OFAST_MASK = the same umax table) at every
                     FAST corner >= 16 px from the border, exact arctan2 in degrees; OpenCV's fastAtan2 is within 0.3 deg of it

The frames are stored too, so the test does not depend on the generator."""
import os, sys
import numpy as np
from skimage.feature import corner_fast, corner_orientations
from skimage.feature.orb import OFAST_MASK

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from synth import synth_frame, noise_frame

out = {}
frames = {"synth11": synth_frame(11, w=320, h=240), "synth12": synth_frame(12, w=256, h=192), "noise5": noise_frame(5, w=160, h=120)}
for name, img in frames.items():
    f = img.astype(np.float64)                       # integer-valued doubles: the comparisons stay exact
    out[name + "_img"] = img
    maxth = np.zeros(img.shape, np.uint8)
    for t in range(7, 256):
        m = corner_fast(f, 9, float(t)) > 0
        if not m.any():
            break
        maxth[m] = t
        if t in (7, 20):
            out["%s_fast_mask_t%d" % (name, t)] = np.packbits(m)
    out[name + "_fast_maxth"] = maxth
    ys, xs = np.nonzero(maxth)
    keep = (xs >= 16) & (xs < img.shape[1] - 16) & (ys >= 16) & (ys < img.shape[0] - 16)
    corners = np.stack([ys[keep], xs[keep]], axis=1)
    ang = corner_orientations(f, corners, OFAST_MASK)             # radians, atan2(m01, m10)
    out[name + "_orient_xy"] = corners[:, ::-1].astype(np.int16)
    out[name + "_orient_deg"] = (np.degrees(ang) % 360.0).astype(np.float32)
# the 256 rBRIEF test pairs as scikit-image ships them (its copy of OpenCV's bit_pattern_31_: an independent transcription of the table at
# src/ORBextractor.cc:150-408)
import skimage.feature
out["orb_positions"] = np.loadtxt(os.path.join(os.path.dirname(skimage.feature.__file__), "orb_descriptor_positions.txt")).astype(np.int8)
np.savez_compressed(os.path.join(HERE, "skimage_fast_orient.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
