"""CPU checks of integer identities the HIP kernels rely on (restated here in numpy; the kernels themselves are covered by the `-m gpu` parity tests):
the key of the matrix-core knn-2 (match_knn.h), k_fast_cells' magic divisions (orb.hip), the match-any ranks of k_lsd_scatter (lsd_front.h), the
reflected border dwords of the blur kernels' branch-free row loads (lsd_front.h: row12_fix)."""
import numpy as np


def test_knn2_matrix_core_key_orders_like_distance_then_index():
    # key = acc + 32 (127 - tile), acc = 4096 (256 - 2 h) + 4096 * 256 + (31 - row-in-tile); larger key = better candidate
    rng = np.random.default_rng(1)
    h = rng.integers(0, 257, size=4000); idx = rng.permutation(4096)[:4000]           # Hamming distance, train index (distinct)
    tile, row = idx // 32, idx % 32
    key = 4096 * (256 - 2 * h) + 4096 * 256 + (31 - row) + 32 * (127 - tile)
    assert key.min() > 0 and key.max() < 2 ** 31                                       # 0 is "no candidate"; fits the i32 accumulator
    order = np.argsort(-key, kind="stable")
    want = np.lexsort((idx, h))                                                         # BFMatcher: distance, then the lower train index
    np.testing.assert_array_equal(order, want)
    # decode (the kernel's canon()): distance and index back out of the key
    T = 127 - ((key >> 5) & 127); ti = T * 32 + (31 - (key & 31))
    np.testing.assert_array_equal(ti, idx)
    np.testing.assert_array_equal((512 - (key >> 12)) >> 1, h)


def test_pm64_expansion_is_a_hamming_product():
    # bit 1 -> +64, bit 0 -> -64 on both sides: sum over 256 bits = 4096 (256 - 2 h), exact in int32; spread4 = (n * 0x00204081) & 0x01010101
    for n in range(16):
        m = (n * 0x00204081) & 0x01010101
        assert [(m >> (8 * i)) & 0xFF for i in range(4)] == [(n >> i) & 1 for i in range(4)]
        v = ((m << 7) ^ 0xC0C0C0C0) & 0xFFFFFFFF
        assert [np.int8(np.uint8((v >> (8 * i)) & 0xFF)) for i in range(4)] == [64 if (n >> i) & 1 else -64 for i in range(4)]
    rng = np.random.default_rng(2)
    a = rng.integers(0, 2, size=(50, 256)); b = rng.integers(0, 2, size=(50, 256))
    dot = ((a * 128 - 64) * (b * 128 - 64)).sum(axis=1)
    np.testing.assert_array_equal(dot, 4096 * (256 - 2 * (a ^ b).sum(axis=1)))


def test_fast_cells_magic_division_is_exact():
    # floor(i / d) == (i * (2^19 / d + 1)) >> 19 for every cell width / dword-group count and every index a cell can hold; the product fits 32 bits
    for d in range(1, 65):
        M = (1 << 19) // d + 1
        i = np.arange(0, 4400, dtype=np.int64)
        np.testing.assert_array_equal((i * M) >> 19, i // d)
        assert M < 1 << 24 and int((i[i < 66 * d] * M).max()) < 1 << 32


def test_match_any_ranks_equal_stable_counting_sort():
    # k_lsd_scatter: per group of 64 entries, rank = entries of the same bin on lower lanes (one ballot per key bit); positions = cursor[bin] + rank
    rng = np.random.default_rng(3)
    bins = np.concatenate([rng.integers(0, 1024, size=700), rng.integers(0, 8, size=500)])      # many duplicates
    cursor = np.zeros(1024, np.int64); counts = np.bincount(bins, minlength=1024)
    start = np.concatenate([[0], np.cumsum(counts[::-1])[:-1]])[::-1]                             # descending bins first (k_lsd_scan)
    cursor[:] = start
    pos = np.empty(len(bins), np.int64)
    for g0 in range(0, len(bins), 64):
        grp = bins[g0:g0 + 64]
        peers = np.ones((len(grp), len(grp)), bool)
        for bit in range(10):
            own = (grp >> bit) & 1
            peers &= own[:, None] == own[None, :]
        rank = np.array([peers[l, :l].sum() for l in range(len(grp))]); tot = peers.sum(axis=1)
        pos[g0:g0 + 64] = cursor[grp] + rank
        last = rank == tot - 1
        cursor[grp[last]] = pos[g0:g0 + 64][last] + 1
    want = np.empty(len(bins), np.int64)
    want[np.lexsort((np.arange(len(bins)), -bins))] = np.arange(len(bins))                         # stable: descending bin, then input order
    np.testing.assert_array_equal(pos, want)


def test_reflected_border_dwords():
    # row12_fix: columns -4..-1 <- 4, 3, 2, 1 from v_perm(d2, d1, 0x01020304); columns w..w+3 <- w-2, w-3, w-4, w-5 from v_perm(d1, d0, 0x03040506)
    def perm(s0, s1, sel):      # v_perm_b32: bytes 0-3 = s1, 4-7 = s0
        b = list(s1) + list(s0)
        return [b[(sel >> (8 * i)) & 0xFF] for i in range(4)]
    row = list(range(100, 116))                                                  # a 16-pixel row
    refl = lambda x: -x if x < 0 else (2 * (len(row) - 1) - x if x >= len(row) else x)      # BORDER_REFLECT_101
    d1, d2 = row[0:4], row[4:8]
    assert perm(d2, d1, 0x01020304)[1:] == [row[refl(c)] for c in (-3, -2, -1)]             # (column -4 is not used by the 7-tap windows)
    d0, d1 = row[8:12], row[12:16]
    assert perm(d1, d0, 0x03040506)[:3] == [row[refl(c)] for c in (16, 17, 18)]
