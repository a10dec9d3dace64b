"""The oracle's restatement of the reference's fallbacks for bands of 200 000 cells and more: the sparse method
(calculateSliceAlternate / setValue / finalizeAlternateSlice, GraphAligner.h:2148-2329, 2130-2146, 2523-2552) and the backtrace
override (:167-354, 2721-2764, 2810-2825).  What pins them: WordSlice::setValue against the reference's own header
(test_oracle_refparts.py); here every cell a sparse slice wrote against the rule of the reference's written-down checker
verifySliceAlternate (tests/cell_dp_checker.py), the bit-vector slices around them against the bit-vector rule, the traceback's own
assertions (every step finds a predecessor whose score fits), and the reference's sharp edges in this corner."""
import numpy as np
import pytest

from graphaligner_amd import synth
import cell_dp_checker as cd
import oracle_binding as ob
from test_cell_dp import _DG


def _check(g, read, seed, bw, ramp=0):
    dg = _DG(g.nodes, g.edges)
    og = ob.OracleGraph(g.nodes, g.edges)
    res = og.align(read, [seed], bw, ramp, record=True)
    part = read + "N" * ((64 - len(read) % 64) % 64)
    prev = {"initial": True, "node": dg.index_of[2 * seed[0]]}
    sparse_cells = dense_cells = 0
    seen = {}
    for rec in res["slice_records"]:
        # (a ramp redo computes a slice twice: every version is checked against the version of the slice above it that it was computed from)
        prev = seen.get(rec["j"] - 64, prev) if rec["j"] > 0 else {"initial": True, "node": dg.index_of[2 * seed[0]]}
        if rec["sparse"]:
            sparse_cells += cd.check_sparse_slice(dg, part, prev, rec, len(part))
        elif not prev.get("sparse") and len(rec["vp"]) < 8000:        # (the bit-vector checker walks columns in Python: small bands only)
            n = cd.check_slice(dg, part, prev, rec, len(part))
            dense_cells += max(n, 0)
        seen[rec["j"]] = rec
    return res, sparse_cells, dense_cells


@pytest.mark.parametrize("branches,branch_len,shared,stem,bw,ramp", [(8, 30000, 150, 600, 35, 0), (8, 30000, 400, 1000, 10, 0), (5, 50000, 100, 333, 35, 60),
                                                                     (12, 20000, 200, 500, 20, 45)])
def test_sparse_slices_and_override(branches, branch_len, shared, stem, bw, ramp):
    g = synth.FanGraph(head_len=200, stem_len=stem, n_branches=branches, branch_len=branch_len, shared=shared, seed=branches)
    rng = np.random.default_rng(branches * 7 + bw)
    total_sparse = n_ok = n_quirk = 0
    for k in range(4):
        read, seed = g.read_through(int(rng.integers(0, branches)), 0, 1800 + 300 * k, rng)
        res, sparse_cells, dense_cells = _check(g, read, seed, bw, ramp)
        if res["status"] == 1 and "overrideLastJ" in res["message"]:
            # a checkpoint in the slice right before an override window: getSlicesFromTable's assert(overrideLastJ > startSlice * 64)
            # (:2862) fires in the reference; the slices were still checked cell by cell above
            n_quirk += 1
            continue
        assert res["status"] == 0 and not res["failed"], res["message"]
        assert res["sparse_slices"] >= 2 and res["override_windows"] >= 1 and res["override_traces"] >= 1, res
        assert [m[0] // 2 for m in res["mappings"]][:2] == [1, 2]                 # head, stem, then one of the branches
        assert sparse_cells > 2000 and dense_cells > 20000
        total_sparse += sparse_cells
        n_ok += 1
    assert n_ok >= 2 and total_sparse > 10000, (n_ok, n_quirk, total_sparse)


def test_reference_sharp_edges_around_the_sparse_method():
    g = synth.FanGraph(head_len=200, stem_len=800, n_branches=8, branch_len=30000, shared=300, seed=3)
    og = ob.OracleGraph(g.nodes, g.edges)
    rng = np.random.default_rng(1)
    # a seed on the stem: slice 0 projects every branch, goes sparse at the ramp width (slice-0 quirk, :2612), which is 0 without -B:
    # calculables[1] of a one-element vector in the reference (:2220) -- reported as an assertion
    read = synth.add_errors(np.concatenate([g.stem, g.branches[2]])[:1500], 0.03, 0.03, 0.03, rng).tobytes().decode()
    res = og.align(read, [(2, 0, False)], 35, 0)
    assert res["status"] == 1 and "bandwidth 0" in res["message"]
    # with -B the same slice 0 opens an override window at row 0; the traceback then asks for the slices between the initial slice
    # and that window: assert(overrideLastJ > startSlice * 64) (:2862)
    res = og.align(read, [(2, 0, False)], 35, 70)
    assert res["status"] == 1 and "overrideLastJ" in res["message"], res["message"]


def test_a_node_of_200_kbp_asserts_in_the_frozen_scores():
    """the literal single-contig case: the sparse method leaves the node's untouched columns at min + length + bandwidth + 1
    (:2546-2549), which no longer fits the 16-bit offsets of the frozen slices: assert(... < 65535) (NodeSlice.h:344, 372)"""
    rng = np.random.default_rng(4)
    contig = synth.random_genome(210000, 77)
    # (a node between the seed node and the long one: slice 0 always projects the seed node's out-neighbours, and would go sparse at
    # the ramp width 0)
    nodes = [(1, contig[:300].tobytes().decode()), (2, contig[300:900].tobytes().decode()), (3, contig[900:].tobytes().decode())]
    edges = [(1, False, 2, False), (2, False, 3, False)]
    og = ob.OracleGraph(nodes, edges)
    read = synth.add_errors(contig[:1500], 0.03, 0.03, 0.03, rng).tobytes().decode()
    res = og.align(read, [(1, 0, False)], 35, 0)
    assert res["status"] == 1 and "65535" in res["message"], res["message"]
