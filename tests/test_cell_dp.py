"""Per-slice column scores of the oracle against an independent cell-by-cell recomputation (tests/cell_dp_checker.py, after the
reference's own written-down checker verifySliceBitvector, GraphAligner.h:1601-1679): linear, SNP / indel bubble and short-node
graphs, forward and backward parts, bands that gain and lose nodes.  Every cell of every column of every slice is compared."""
import numpy as np
import pytest

from graphaligner_amd import synth
import cell_dp_checker as cd
import oracle_binding as ob


def _check_read(g, dg, og, read, seed, bw):
    res = og.align(read, [seed], bw, record=True)
    assert res["status"] == 0, res["message"]
    node_id, pos, rev = seed
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    cells = 0
    slices_by_dir = {}
    for rec in res["slice_records"]:
        slices_by_dir.setdefault(rec["direction"], []).append(rec)
    for direction, recs in slices_by_dir.items():
        # getSplitAlignment (GraphAligner.h:2969-3024): forward part = read[pos:], backward part = revcomp(read[:pos]); padded with N
        if direction == 0:
            part = read[pos:]
            seed_digraph_id = 2 * node_id + (1 if rev else 0)
        else:
            part = "".join(comp[c] for c in reversed(read[:pos]))
            seed_digraph_id = 2 * node_id + (0 if rev else 1)
        part = part + "N" * ((64 - len(part) % 64) % 64)
        seed_index = dg.index_of[seed_digraph_id]
        prev = {"initial": True, "node": seed_index}
        for rec in sorted(recs, key=lambda r: r["j"]):
            n = cd.check_slice(dg, part, prev, rec, len(part))
            assert n != 0
            if n > 0:
                cells += n
            prev = rec
    return cells


class _DG(cd.Digraph):
    def __init__(self, nodes, edges):
        super().__init__(nodes, edges)
        self.index_of = {}
        k = 1
        for nid, _ in nodes:
            self.index_of[2 * nid] = k
            self.index_of[2 * nid + 1] = k + 1
            k += 2


@pytest.mark.parametrize("node_len,snp,indel,mid", [(64, 0, 0, False), (32, 60, 400, False), (16, 45, 300, True), (5, 30, 200, False)])
def test_oracle_slices_equal_cell_by_cell_dp(node_len, snp, indel, mid):
    g = synth.SynthGraph(synth.random_genome(30000, 5 + node_len), node_len=node_len, snp_every=snp, indel_every=indel, seed=9)
    reads, seeds = synth.simulate_reads(g, 4, 700, seed=3 + node_len, mid_seed=mid)
    dg = _DG(g.nodes, g.edges)
    og = ob.OracleGraph(g.nodes, g.edges)
    total = 0
    for r, s in zip(reads, seeds):
        total += _check_read(g, dg, og, r, s, 35)
    assert total > 250000
