"""Pins the CPU oracle's component functions against the parts of the REFERENCE that compile
from their own sources (oracle/_ref/libga_refparts.so, built by oracle/Makefile from
/root/reference/{WordSlice.h,NodeSlice.h,AlignmentCorrectnessEstimation.cpp}).

The engine itself (GraphAligner.h) is not buildable in this image, so these are the only
reference-executed pins the oracle has; everything above them is "parity unpinned"."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob

ref = ob.reflib()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libga_refparts.so not built (reference tree absent)")


def _col(vp, vn, before, rows=64, partial=0, bex=0, eex=1):
    vp, vn = int(vp), int(vn)
    end = before + bin(vp).count("1") - bin(vn).count("1")
    return np.array([np.uint64(vp).astype(np.int64), np.uint64(vn).astype(np.int64), end, before, rows, partial, bex, eex], dtype=np.int64)


def _rand_col(rng, before_lo=0, before_hi=60, density=None):
    # a valid column: VP & VN disjoint, vertical deltas in {-1,0,+1}
    d = density if density is not None else rng.choice([0.05, 0.3, 0.5, 0.9])
    bits = rng.random(64)
    vp = sum(1 << i for i in range(64) if bits[i] < d * 0.6)
    vn = sum(1 << i for i in range(64) if d * 0.6 <= bits[i] < d)
    return vp, vn, int(rng.integers(before_lo, before_hi)) + 64


def test_column_value_matches_reference():
    rng = np.random.default_rng(1)
    L = ob.lib()
    for _ in range(300):
        vp, vn, b = _rand_col(rng)
        c = _col(vp, vn, b)
        for row in (0, 1, 31, 62, 63, int(rng.integers(0, 64))):
            assert L.gao_column_value(ob._p(c), row) == ref.ref_column_value(ob._p(c), row)


def test_merge_fully_confirmed_matches_reference():
    """the acyclic-band case: both columns have all 64 rows confirmed (WordSlice.h:361-421)"""
    rng = np.random.default_rng(2)
    L = ob.lib()
    n_ok = 0
    for it in range(4000):
        a = _col(*_rand_col(rng), bex=int(rng.integers(0, 2)))
        if it % 3 == 0:
            # the vertical re-entry shape: all +1 from a lower start (GraphAligner.h:1506-1508)
            b = _col((1 << 64) - 1, 0, int(a[3]) - int(rng.integers(1, 40)), bex=int(rng.integers(0, 2)))
        else:
            b = _col(*_rand_col(rng), bex=int(rng.integers(0, 2)))
        o1 = np.zeros(8, dtype=np.int64)
        o2 = np.zeros(8, dtype=np.int64)
        s2 = ref.ref_merge_columns(ob._p(a), ob._p(b), ob._p(o2))
        s1 = L.gao_merge_columns(ob._p(a), ob._p(b), ob._p(o1))
        if s2 != 0:
            continue   # the reference asserted on this random input
        assert s1 == 0
        assert (o1 == o2).all(), (a, b, o1, o2)
        n_ok += 1
    assert n_ok > 3000


def test_merge_partially_confirmed_matches_reference():
    """cyclic components: confirmedRowsInMerged (WordSlice.h:423-510) on partially confirmed columns"""
    rng = np.random.default_rng(3)
    L = ob.lib()
    n_ok = 0
    for it in range(20000):
        ra, rb = int(rng.integers(0, 65)), int(rng.integers(0, 65))
        pa = int(rng.integers(0, 2)) if ra < 64 else 0
        pb = int(rng.integers(0, 2)) if rb < 64 else 0
        a = _col(*_rand_col(rng, 0, 12), rows=ra, partial=pa, bex=int(rng.integers(0, 2)))
        b = _col(*_rand_col(rng, 0, 12), rows=rb, partial=pb, bex=int(rng.integers(0, 2)))
        o1 = np.zeros(8, dtype=np.int64)
        o2 = np.zeros(8, dtype=np.int64)
        s2 = ref.ref_merge_columns(ob._p(a), ob._p(b), ob._p(o2))
        if s2 != 0:
            continue
        s1 = L.gao_merge_columns(ob._p(a), ob._p(b), ob._p(o1))
        assert s1 == 0, (a, b)
        assert (o1 == o2).all(), (a, b, o1, o2)
        n_ok += 1
    assert n_ok > 2000


def test_hmm_matches_reference_bitwise():
    """the Viterbi recurrence in double precision must agree to the last bit (it steers control flow)"""
    rng = np.random.default_rng(4)
    L = ob.lib()
    for trial in range(50):
        n = 200
        mism = rng.integers(0, 30 if trial % 2 else 65, size=n).astype(np.int32)
        c1, w1, f1 = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.uint8)
        c2, w2, f2 = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.uint8)
        L.gao_hmm_chain(ob._p(mism), n, ob._p(c1), ob._p(w1), ob._p(f1))
        ref.ref_hmm_chain(ob._p(mism), n, ob._p(c2), ob._p(w2), ob._p(f2))
        assert (c1.view(np.uint64) == c2.view(np.uint64)).all()
        assert (w1.view(np.uint64) == w2.view(np.uint64)).all()
        assert (f1 == f2).all()


def test_frozen_slice_iteration_order_matches_reference():
    """band order of slice s+1 starts from the hash-map iteration order of frozen slice s
    (NodeSlice.h:724-740 feeding GraphAligner.h:1117)"""
    rng = np.random.default_rng(6)
    L = ob.lib()
    L.gao_frozen_order.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    for trial in range(300):
        n = int(rng.integers(1, 200))
        universe = int(rng.choice([64, 1000, 100000, 5000000]))
        nodes = rng.choice(universe, size=min(n, universe), replace=False).astype(np.int64)
        o1 = np.zeros(len(nodes), dtype=np.int64)
        o2 = np.zeros(len(nodes), dtype=np.int64)
        assert L.gao_frozen_order(ob._p(nodes), len(nodes), universe, ob._p(o1)) == len(nodes)
        assert ref.ref_frozen_order(ob._p(nodes), len(nodes), universe, ob._p(o2)) == len(nodes)
        assert (o1 == o2).all()


def test_freeze_thaw_matches_reference():
    """NodeSlice.h:300-376: what the next slice sees of a frozen column"""
    rng = np.random.default_rng(5)
    for trial in range(100):
        n = int(rng.integers(1, 40))
        cols = np.stack([_col(*_rand_col(rng, 0, 200), bex=int(rng.integers(0, 2))) for _ in range(n)])
        for mode in (1, 2):
            out = np.zeros_like(cols)
            assert ref.ref_freeze_thaw(ob._p(cols), n, mode, ob._p(out)) == 0
            for i in range(n):
                vp, vn = np.uint64(cols[i, 0]), np.uint64(cols[i, 1])
                if mode == 2:
                    # TinySlice: end score exact, only the last vertical bit survives
                    p, m = int(vp >> np.uint64(63)), int(vn >> np.uint64(63))
                    assert out[i, 2] == cols[i, 2]
                    assert out[i, 3] == cols[i, 2] - p + m
                    assert np.uint64(out[i, 0]) == np.uint64(p) << np.uint64(63)
                    assert np.uint64(out[i, 1]) == np.uint64(m) << np.uint64(63)
                else:
                    # SmallSlice: full bits and exact scoreBeforeStart; scoreEnd is left 0 (NodeSlice.h:304)
                    assert out[i, 0] == cols[i, 0] and out[i, 1] == cols[i, 1]
                    assert out[i, 3] == cols[i, 3]
                    assert out[i, 2] == 0
                assert out[i, 4] == 64 and out[i, 5] == 0 and out[i, 6] == 0 and out[i, 7] == 1


def test_rank_queries_and_word_helpers_match_reference():
    """WordConfiguration::{popcount, ChunkPopcounts, MortonLow/High, BitPosition} (WordSlice.h:27-130) as confirmedRowsInMerged uses
    them (:479-488): the oracle answers the same rank query with a plain loop"""
    rng = np.random.default_rng(11)
    L = ob.lib()
    L.gao_interleaved_rank.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]
    ref.ref_interleaved_rank.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]
    ref.ref_popcount.argtypes = [C.c_uint64]
    ref.ref_chunk_popcounts.argtypes = [C.c_uint64]
    ref.ref_chunk_popcounts.restype = C.c_uint64
    for f in (ref.ref_morton_low, ref.ref_morton_high):
        f.argtypes = [C.c_uint64, C.c_uint64]
        f.restype = C.c_uint64
    for trial in range(3000):
        vp = int(rng.integers(0, 1 << 63, dtype=np.uint64)) | (int(rng.integers(0, 2)) << 63)
        vn = int(rng.integers(0, 1 << 63, dtype=np.uint64)) & ~vp
        assert ref.ref_popcount(vp) == bin(vp).count("1")
        cp = ref.ref_chunk_popcounts(vp)
        assert [(cp >> (8 * k)) & 0xff for k in range(8)] == [bin((vp >> (8 * k)) & 0xff).count("1") for k in range(8)]
        a, b = vp & 0xffffffff, vn & 0xffffffff
        want = sum(((a >> i) & 1) << (2 * i) | ((b >> i) & 1) << (2 * i + 1) for i in range(32))
        assert ref.ref_morton_low(vp, vn) == want
        assert ref.ref_morton_high(vp, vn) == sum((((vp >> 32) >> i) & 1) << (2 * i) | (((vn >> 32) >> i) & 1) << (2 * i + 1) for i in range(32))
        lo = int(rng.integers(0, 63))
        hi = int(rng.integers(lo + 1, 65))
        units = bin(vp & ((1 << hi) - 1) & ~((1 << lo) - 1)).count("1") + bin(~vn & ((1 << hi) - 1) & ~((1 << lo) - 1) & ((1 << 64) - 1)).count("1")
        for rank in (0, 1, int(rng.integers(0, 130)), units - 1 if units else 0, units):
            assert L.gao_interleaved_rank(vp, vn, lo, hi, rank) == ref.ref_interleaved_rank(vp, vn, lo, hi, rank), (hex(vp), hex(vn), lo, hi, rank)


def test_work_stack_matches_reference_unique_queue():
    """the LIFO work list with membership flags that drives the per-component relaxation (UniqueQueue.h:6-69, GraphAligner.h:2364-2397)"""
    rng = np.random.default_rng(12)
    L = ob.lib()
    L.gao_work_stack.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    ref.ref_unique_queue.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    for trial in range(300):
        universe = int(rng.integers(1, 50))
        ops = np.where(rng.random(200) < 0.35, -1, rng.integers(0, universe, size=200)).astype(np.int64)
        o1 = np.zeros(300, dtype=np.int64)
        o2 = np.zeros(300, dtype=np.int64)
        n1 = L.gao_work_stack(ob._p(ops), len(ops), universe, ob._p(o1))
        n2 = ref.ref_unique_queue(ob._p(ops), len(ops), universe, ob._p(o2))
        assert n1 == n2 and (o1[:n1] == o2[:n2]).all()


def test_set_value_matches_reference():
    """the sparse method's cell writes (WordSlice::setValue, WordSlice.h:231-337): chains of (row, value) with consecutive rows,
    gaps and values that break the reference's own asserts -- same column after every chain, same place of the first assertion"""
    rng = np.random.default_rng(11)
    L = ob.lib()
    n_full = n_cut = 0
    for it in range(6000):
        rows, values = [], []
        r = int(rng.integers(0, 8))
        v = int(rng.integers(0, 50))
        style = it % 4
        while r < 64:
            rows.append(r)
            values.append(v)
            step = 1 if (style == 0 or rng.random() < 0.7) else int(rng.integers(2, 12))
            r += step
            if style == 3:
                v += int(rng.integers(-3, 4))          # often outside what the column allows: the asserts must fire at the same cell
            else:
                v += int(rng.integers(-1, 2)) if step == 1 else int(rng.integers(-step, step + 1))
        ra = np.array(rows, dtype=np.int32)
        va = np.array(values, dtype=np.int32)
        o1 = np.zeros(8, dtype=np.int64)
        o2 = np.zeros(8, dtype=np.int64)
        uninit = 10048
        d2 = ref.ref_set_values(ob._p(ra), ob._p(va), len(rows), uninit, ob._p(o2))
        d1 = L.gao_set_values(ob._p(ra), ob._p(va), len(rows), uninit, ob._p(o1))
        assert d1 == d2, (rows, values, d1, d2)
        assert (o1 == o2).all(), (rows, values, o1, o2)
        if d2 == len(rows):
            n_full += 1
        else:
            n_cut += 1
    assert n_full > 1500 and n_cut > 500
