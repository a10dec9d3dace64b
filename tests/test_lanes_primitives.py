"""The column primitives and order emulations of the lanes = reads program (graphaligner_amd/csrc/ga_lanes.h, PRODUCT code run
through the host build of the same header) against the reference's own WordSlice (oracle/_ref, built from /root/reference) and
against the real libstdc++ containers:

  column_merge    == WordSlice::mergeWith on fully confirmed columns            (WordSlice.h:361-421)
  column_reenter  == mergeWith(getSourceSliceFromScore(score above))            (GraphAligner.h:1504-1509, 1541-1546)
  column_step     == the oracle's getNextSlice restatement (itself checked cell by cell in test_cell_dp.py)
  hash_order      == iteration order of the frozen NodeSlice / std::unordered_map (NodeSlice.h:724-740)
  heap            == pop order of std::priority_queue<.., std::greater<>>       (GraphAligner.h:1115)"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import parity_common as pc
from test_oracle_refparts import _col, _rand_col

ref = ob.reflib()


@pytest.fixture(scope="module")
def emul():
    return C.CDLL(pc.emul_lib_path())


def _c3(vp, vn, before):
    return np.array([vp, vn, before & ((1 << 64) - 1)], dtype=np.uint64)


@pytest.mark.skipif(ref is None, reason="oracle/_ref/libga_refparts.so not built (reference tree absent)")
def test_column_merge_and_reentry_match_reference_wordslice(emul):
    rng = np.random.default_rng(21)
    n_ok = 0
    for it in range(3000):
        avp, avn, ab = _rand_col(rng)
        if it % 2 == 0:
            bvp, bvn, bb = (1 << 64) - 1, 0, ab - int(rng.integers(1, 60))          # the run coming down from the cell above
        else:
            bvp, bvn, bb = _rand_col(rng)
        a, b = _col(avp, avn, ab), _col(bvp, bvn, bb)
        want = np.zeros(8, dtype=np.int64)
        if ref.ref_merge_columns(ob._p(a), ob._p(b), ob._p(want)) != 0:
            continue
        n_ok += 1
        out = np.zeros(3, dtype=np.uint64)
        emul.ga_emul_lanes_merge(ob._p(_c3(avp, avn, ab)), ob._p(_c3(bvp, bvn, bb)), ob._p(out))
        assert (int(out[0]), int(out[1]), int(np.int64(out[2]))) == (int(np.uint64(want[0])), int(np.uint64(want[1])), int(want[3]))
        if it % 2 == 0:
            out2 = np.zeros(3, dtype=np.uint64)
            emul.ga_emul_lanes_reenter(ob._p(_c3(avp, avn, ab)), ab - bb, ob._p(out2))
            assert (out2 == out).all(), (hex(avp), hex(avn), ab, bb)
    assert n_ok > 2500


def test_column_step_matches_oracle_step(emul):
    rng = np.random.default_rng(22)
    L = ob.lib()
    for it in range(3000):
        vp, vn, b = _rand_col(rng)
        eq = int(rng.integers(0, 1 << 63, dtype=np.uint64)) | (int(rng.integers(0, 2)) << 63)
        left_exists, diag_in = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        up_left = int(rng.integers(0, 2))
        prev_row_eq = int(rng.integers(0, 2))
        # the cell above the left column: any end score not below the left column's row j-1 score (GraphAligner.h:1366)
        above_end = b + int(rng.integers(0, 3))
        avp = (1 << 63) if rng.random() < 0.5 else 0
        avn = (1 << 63) if (avp == 0 and rng.random() < 0.3) else 0
        left = _col(vp, vn, b, bex=left_exists)
        above = np.array([np.uint64(avp).astype(np.int64), np.uint64(avn).astype(np.int64), above_end, 0, 64, 0, 0, 1], dtype=np.int64)
        want = np.zeros(8, dtype=np.int64)
        if L.gao_step_column(eq, ob._p(left), 1, up_left, diag_in, prev_row_eq, ob._p(above), ob._p(want)) != 0:
            continue
        calc = b + 1
        if up_left:
            calc = min(calc, above_end - (1 if avp else 0) + (1 if avn else 0) + (0 if prev_row_eq else 1))
        out = np.zeros(3, dtype=np.uint64)
        emul.ga_emul_lanes_step(ob._p(_c3(vp, vn, b)), C.c_uint64(eq), int(not (left_exists and diag_in)), calc, ob._p(out))
        assert (int(out[0]), int(out[1]), int(np.int64(out[2]))) == (int(np.uint64(want[0])), int(np.uint64(want[1])), int(want[3])), it


def test_lane_table_orders_match_real_containers(emul):
    L = ob.lib()
    L.gao_frozen_order.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    L.gao_pq_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(23)
    for trial in range(400):
        n = int(rng.integers(1, 57))
        universe = 150000000 if trial < 3 else int(rng.choice([300, 5000, 200000]))      # (node indices of whole-genome graphs are large)
        keys = rng.choice(universe, size=n, replace=False).astype(np.uint32)
        out = np.zeros(n, dtype=np.int32)
        assert emul.ga_emul_lanes_hash_order(ob._p(keys), n, ob._p(out)) == n
        o = np.zeros(n, dtype=np.int64)
        assert L.gao_frozen_order(ob._p(keys.astype(np.int64)), n, universe, ob._p(o)) == n
        assert (keys[out] == o).all()
        if ref is not None and trial % 8 == 0:
            o2 = np.zeros(n, dtype=np.int64)
            assert ref.ref_frozen_order(ob._p(keys.astype(np.int64)), n, universe, ob._p(o2)) == n
            assert (keys[out] == o2).all()
    for trial in range(300):
        n_ops = int(rng.integers(1, 200))
        nodes = rng.integers(0, 1000, size=n_ops).astype(np.uint32)
        prios = rng.integers(0, int(rng.choice([3, 10, 100])), size=n_ops).astype(np.int32)
        prios[rng.random(n_ops) < 0.4] = -1
        a = np.zeros(n_ops, dtype=np.uint32)
        b = np.zeros(n_ops, dtype=np.uint32)
        ka = emul.ga_emul_lanes_heap(ob._p(nodes), ob._p(prios), n_ops, ob._p(a))
        if ka < 0:
            continue                               # more than the table's 112 entries at once
        kb = L.gao_pq_order(ob._p(nodes), ob._p(prios), n_ops, ob._p(b))
        assert ka == kb and (a[:ka] == b[:kb]).all()
