"""Optional node splitting in the GFA loader: a graph of long segments -- the literal "single-contig GFA", which the
reference leaves to its sparse method because one band already holds >= 200 000 bp (GraphAlignerCommon.h:10) -- is cut into chains of
pieces so that the bit-vector path applies.  Checked: the cut graph aligns exactly like the same pieces given as an ordinary graph (and
like the oracle on it), and ga_results_unsplit puts the result back on the file's nodes."""
import numpy as np

from graphaligner_amd import binding, synth
import oracle_binding as ob
import parity_common as pc


def _long_segment_gfa(n_segments, seg_len, seed):
    rng = np.random.default_rng(seed)
    segs = ["".join("ACGT"[i] for i in rng.integers(0, 4, size=seg_len)) for _ in range(n_segments)]
    lines = ["H\tVN:Z:1.0"] + ["S\t%d\t%s" % (i + 1, s) for i, s in enumerate(segs)]
    # a chain with one inversion edge: segment 2 is entered through its end from segment 1's end
    for i in range(1, n_segments):
        lines.append("L\t%d\t+\t%d\t+\t0M" % (i, i + 1))
    lines.append("L\t1\t+\t2\t-\t0M")
    return segs, "\n".join(lines) + "\n"


def _pieces_graph(segs, max_len, n_segments):
    """the same cutting done by hand: nodes and edges of the piece graph, with the loader's id rule (first piece keeps the id)"""
    nodes, edges, ends = [], [], {}
    nxt = n_segments + 1
    for i, s in enumerate(segs):
        sid, prev = i + 1, None
        for at in range(0, len(s), max_len):
            pid = sid if at == 0 else nxt
            if at:
                nxt += 1
            nodes.append((pid, s[at:at + max_len]))
            if prev is not None:
                edges.append((prev, False, pid, False))
            prev = pid
        ends[sid] = (sid, prev)
    for i in range(1, n_segments):
        edges.append((ends[i][1], False, ends[i + 1][0], False))
    edges.append((ends[1][1], False, ends[2][1], True))
    return nodes, edges


def test_split_loader_matches_hand_cut_graph_and_unsplit_restores_ids():
    lib = pc.emul_lib_path()
    segs, gfa = _long_segment_gfa(3, 5000, 81)
    nodes, edges = _pieces_graph(segs, 64, 3)
    rng = np.random.default_rng(82)
    reads, seeds = [], []
    for k in range(6):
        seg = int(rng.integers(0, 3))
        start = int(rng.integers(0, 5000 - 1300)) // 64 * 64
        body = np.frombuffer(segs[seg][start:start + 1200].encode(), dtype=np.uint8)
        if k % 2 == 0:
            reads.append(synth.add_errors(body, 0.03, 0.03, 0.03, rng).tobytes().decode())
            piece = [n for n, _ in nodes if n == seg + 1][0] if start == 0 else None
            seeds.append((None, 0, False, seg, start))
        else:
            reads.append(synth.add_errors(synth.revcomp_bytes(body), 0.03, 0.03, 0.03, rng).tobytes().decode())
            seeds.append((None, 0, True, seg, start + 1199))
    split = binding.Graph(gfa=gfa, lib_path=lib, split=64)
    plain = binding.Graph(nodes, edges, lib_path=lib)
    assert split.node_count == plain.node_count and split.bp == plain.bp
    # seeds name pieces: the piece holding the read's first base
    piece_of = {}
    for (pid, s) in nodes:
        piece_of[pid] = s
    per_seg, nxt = {}, 4
    for i, s in enumerate(segs):
        ids = [i + 1] + list(range(nxt, nxt + (len(s) - 1) // 64))
        nxt += (len(s) - 1) // 64
        per_seg[i] = ids
    real_seeds = [(per_seg[seg][pos // 64], 0, rev) for (_, _, rev, seg, pos) in seeds]
    a = split.prepare(reads, real_seeds, 35, 0, flags=binding.GA_F_TRACE)
    a.run()
    ra = a.collect()
    rb = plain.align(reads, real_seeds, 35, flags=binding.GA_F_TRACE)
    og = ob.OracleGraph(nodes, edges)
    for x, y, r, sd in zip(ra, rb, reads, real_seeds):
        assert x["status"] == 0 and not x["failed"]
        assert x["score"] == y["score"] and x["mappings"] == y["mappings"]
        pc.compare_read(x, og.align(r, [sd], 35), "split")
    merged = a.collect(unsplit=True)
    for x, m, (_, _, rev, seg, pos) in zip(ra, merged, seeds):
        assert m["score"] == x["score"]
        assert [mm[0] // 2 for mm in m["mappings"]] == [seg + 1]            # one mapping on the file's segment
        assert m["mappings"][0][1] == int(rev)
        assert sum(mm[5] for mm in x["mappings"]) == m["mappings"][0][5]    # read bases add up
        assert sum(mm[4] for mm in x["mappings"]) == m["mappings"][0][4]
        assert "".join(mm[6] for mm in x["mappings"]) == m["mappings"][0][6]
        # the offset is where the read starts inside the segment (reverse strand: counted from the segment's end)
        want = pos - (pos % 64) + x["mappings"][0][2] if not rev else (5000 - 1 - pos) - ((5000 - 1 - pos) % 64) + x["mappings"][0][2]
        assert abs(m["mappings"][0][2] - want) <= 64
        assert (m["trace"][:, 0] == seg + 1).all()
