"""Seeded synthetic frames for the front-end tests and bench.

Frame = mid-grey background + random filled rectangles / rotated rectangles /
ellipses with uniform intensities + line strokes + additive Gaussian noise
(sigma 2), clipped to u8.  Everything is integer / numpy so the same seed gives
the same bytes in the build container and on the GPU box (same image, same numpy).
"""
import numpy as np


def _fill_poly_mask(h, w, cx, cy, hw, hh, ang):
    yy, xx = np.mgrid[0:h, 0:w]
    c, s = np.cos(ang), np.sin(ang)
    u = (xx - cx) * c + (yy - cy) * s
    v = -(xx - cx) * s + (yy - cy) * c
    return (np.abs(u) <= hw) & (np.abs(v) <= hh)


def synth_frame(seed=1234, w=640, h=480, nshapes=None, nstrokes=None, noise=2.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    area = (w * h) / (640 * 480)
    nshapes = int(60 * area) if nshapes is None else nshapes
    nstrokes = int(40 * area) if nstrokes is None else nstrokes
    img = np.full((h, w), 110.0, dtype=np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(nshapes):
        kind = rng.integers(0, 3)
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        hw, hh = rng.uniform(8, 70), rng.uniform(8, 70)
        val = rng.uniform(20, 235)
        if kind == 0:
            m = (np.abs(xx - cx) <= hw) & (np.abs(yy - cy) <= hh)
        elif kind == 1:
            m = _fill_poly_mask(h, w, cx, cy, hw, hh, rng.uniform(0, np.pi))
        else:
            m = ((xx - cx) / hw) ** 2 + ((yy - cy) / hh) ** 2 <= 1.0
        img[m] = val
    for _ in range(nstrokes):
        x0, y0 = rng.uniform(0, w), rng.uniform(0, h)
        ang, ln = rng.uniform(0, np.pi), rng.uniform(30, 250)
        wd = rng.integers(1, 4)
        val = rng.uniform(20, 235)
        c, s = np.cos(ang), np.sin(ang)
        u = (xx - x0) * c + (yy - y0) * s
        v = -(xx - x0) * s + (yy - y0) * c
        m = (u >= 0) & (u <= ln) & (np.abs(v) <= wd * 0.5)
        img[m] = val
    if noise > 0:
        img = img + rng.normal(0.0, noise, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp_prev(img, dx=3.0, dy=-2.0, deg=1.5):
    """Previous frame = same scene translated by (dx,dy) px and rotated `deg` about the centre
    (bilinear, reflect at the border)."""
    h, w = img.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a = np.deg2rad(deg)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    xs = (xx - cx) * np.cos(a) + (yy - cy) * np.sin(a) + cx - dx
    ys = -(xx - cx) * np.sin(a) + (yy - cy) * np.cos(a) + cy - dy
    xs = np.clip(xs, 0, w - 1.001)
    ys = np.clip(ys, 0, h - 1.001)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    fx, fy = xs - x0, ys - y0
    f = img.astype(np.float64)
    v = (f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy)
         + f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def noise_frame(seed=7, w=640, h=480):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(h, w), dtype=np.uint8)


def const_frame(val=128, w=640, h=480):
    return np.full((h, w), val, dtype=np.uint8)


def ramp_frame(seed=3, w=640, h=480):
    """Two thick smooth edges: LSD regions of several thousand pixels (exercises the region-list spill past the LDS queue)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 60 + 120 / (1 + np.exp(-((xx * 0.02 + yy) - 0.42 * h) / 2.5)) + 50 / (1 + np.exp(-((xx - 0.05 * yy) - 0.62 * w) / 3.0))
    rng = np.random.Generator(np.random.PCG64(seed))
    img = img + rng.normal(0, 1.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synthetic_vocab(rng, k=10, L=4):
    """a ragged DBoW2-style tree in DBoW2's node numbering (root 0, parent id < child id): (L, child_ptr, children, node_desc, word_id, weight)"""
    ptr = [0]; children = []; desc = [np.zeros(32, np.uint8)]; level_of = [0]
    parent_desc = {0: rng.integers(0, 256, 32, dtype=np.uint8)}
    frontier = [0]; nid = 1
    child_lists = {}
    for lvl in range(1, L + 1):
        nxt = []
        for p in frontier:
            kk = k if not (lvl == L and p % 7 == 3) else 0           # some level L-1 nodes stay leaves (ragged tree)
            ids = list(range(nid, nid + kk)); nid += kk
            child_lists[p] = ids
            for c in ids:
                flips = rng.random(256) < 0.5 / lvl
                parent_desc[c] = parent_desc[p] ^ np.packbits(flips)
                level_of.append(lvl)
            nxt += ids
        frontier = nxt
    n = nid
    ptr = np.zeros(n + 1, np.int32); ch = []
    for i in range(n):
        ids = child_lists.get(i, [])
        ch += ids; ptr[i + 1] = len(ch)
    desc = np.stack([parent_desc[i] for i in range(n)])
    leaf = np.array([len(child_lists.get(i, [])) == 0 for i in range(n)])
    word = np.full(n, -1, np.int32); word[leaf] = np.arange(leaf.sum(), dtype=np.int32)
    weight = np.zeros(n, np.float64); weight[leaf] = np.where(rng.random(leaf.sum()) < 0.05, 0.0, rng.uniform(0.1, 9.0, leaf.sum()))
    return L, ptr, np.array(ch, np.int32), desc, word, weight


def write_vocab_text(path, k, L, ptr, children, desc, weight, scoring=0, weighting=0, weight_fmt="%r", trailing_newline=True):
    """the file TemplatedVocabulary::saveToTextFile writes (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1428-1455): header
    "k L  scoring weighting", then "parent isLeaf d0 .. d31 weight" per node from id 1"""
    n = len(ptr) - 1
    parent = np.zeros(n, np.int64)
    for p in range(n):
        parent[children[ptr[p]:ptr[p + 1]]] = p
    lines = ["%d %d  %d %d" % (k, L, scoring, weighting)]
    for i in range(1, n):
        leaf = int(ptr[i + 1] == ptr[i])
        lines.append("%d %d %s %s" % (parent[i], leaf, " ".join(str(int(b)) for b in desc[i]), weight_fmt % float(weight[i])))
    with open(path, "w") as f:
        f.write("\n".join(lines) + ("\n" if trailing_newline else ""))
