"""The simulate -> align -> compare loop the reference runs by hand (SimulateReads -> Aligner -> CompareAlignments), with the
deterministic simulator of graphaligner_amd/synth.py and the reference's own criterion (CompareAlignments.cpp:13-44, 86)."""
from graphaligner_amd import binding, compare, synth
import parity_common as pc


def test_identity_arithmetic():
    sizes = {1: 10, 2: 20, 3: 30, 4: 5}
    assert compare.alignment_identity([1, 2, 3], [2, 3, 4], sizes) == (50, 10, 5)
    assert compare.alignment_identity([1, 2, 2], [2], sizes) == (20, 30, 0)          # mappings are summed, sets are intersected
    r = compare.compare({"a": [1, 2, 3], "b": [1]}, {"a": [2, 3, 4], "c": [4]}, sizes)
    assert (r["good"], r["bad"]) == (1, 2)                                           # a: 50 / 65 = 0.77; b and c are unmatched


def test_simulated_reads_come_back_to_where_they_were_drawn():
    lib = pc.emul_lib_path()
    g = synth.bubble_graph(60000, node_len=32, seed=71)
    truth = []
    reads, seeds = synth.simulate_reads(g, 16, 1500, seed=72, truth=truth)
    t2 = []
    r2, s2 = synth.simulate_reads(g, 8, 1500, seed=73, mid_seed=True, truth=t2)
    reads, seeds, truth = reads + r2, seeds + s2, truth + t2
    sizes = {nid: len(seq) for nid, seq in g.nodes}
    # the recorded walks spell the reads' error-free sequences: every step follows an edge of the graph
    edges = set((f, t) for f, _, t, _ in g.edges)
    for p, s in zip(truth, seeds):
        fwd = p if not s[2] else p[::-1]
        assert all((a, b) in edges for a, b in zip(fwd, fwd[1:]))
    res = binding.Graph(g.nodes, g.edges, lib_path=lib).align(reads, seeds, 35)
    rep = compare.compare({"r%d" % i: p for i, p in enumerate(truth)},
                          {"r%d" % i: compare.predicted_nodes(r) for i, r in enumerate(res) if not r["failed"]}, sizes)
    assert rep["good"] >= 22 and rep["good"] + rep["bad"] == 24
