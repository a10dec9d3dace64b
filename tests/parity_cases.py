"""Parity cases shared by the GPU tests (real library) and the emulated-device tests (the same
device program executed on the host, tests/emul).  Every case compares the C ABI's results with
the CPU oracle on identical seeded inputs, bit for bit."""
import numpy as np

from graphaligner_amd import binding, synth
import parity_common as pc
import oracle_binding as ob


def case_wave_primitives_on_hardware(lib_path=None):
    """the DPP scan / shift the program is built from, checked through a whole alignment that
    only matches when they behave as the host emulation does"""
    g = synth.linear_graph(30000, node_len=64, seed=3)
    reads, seeds = synth.simulate_reads(g, 8, 1200, seed=9)
    pc.check_parity(g.nodes, g.edges, reads, seeds, 35, lib_path=lib_path, ctx="linear")


RANDOM_GRAPHS = [(64, 0, 0, 0), (64, 100, 1000, 0), (32, 40, 300, 3000), (8, 15, 60, 0), (5, 40, 0, 0)]


def case_random_graphs(node_len, snp, indel, sv, lib_path=None):
    rng = np.random.default_rng(node_len * 1000 + snp)
    g = synth.SynthGraph(synth.random_genome(25000, 500 + node_len), node_len=node_len, snp_every=snp, indel_every=indel, sv_every=sv, seed=node_len)
    for bw, err, length, mid in [(35, 0.04, 2500, False), (35, 0.04, 2500, True), (10, 0.02, 1000, False), (64, 0.1, 1500, True), (2, 0.0, 700, False)]:
        reads, seeds = synth.simulate_reads(g, 12, length, sub=err, ins=err, dele=err, seed=int(rng.integers(1 << 30)), mid_seed=mid)
        devs, oras = pc.run_both(g.nodes, g.edges, reads, seeds, bw, lib_path=lib_path)
        n_cmp = 0
        for i, (d, o) in enumerate(zip(devs, oras)):
            if d["status"] == 10:
                continue      # band wider than the widest kernel variant: reported, not silently wrong
            pc.compare_read(d, o, "nl%d bw%d read %d" % (node_len, bw, i))
            n_cmp += 1
        assert n_cmp >= len(reads) // 2


CYCLIC_GRAPHS = [(4, 10, 6, 1, 2), (8, 35, 8, 2, 5), (16, 35, 5, 3, 3), (32, 80, 9, 0, 8), (12, 35, 12, 2, 1)]


def case_cyclic_graphs(node_len, bw, back_edges, self_loops, max_span, lib_path=None):
    """bands with strongly connected components (tandem-repeat back edges, self loops): the
    reference relaxes such components with row confirmation (GraphAligner.h:2362-2397); node
    minima, band choice and the trace start depend on the visiting order, so only a faithful
    restatement passes.  Reads are random walks that go round the repeats."""
    g = synth.cyclic_graph(5000, node_len=node_len, seed=node_len + bw, back_edges=back_edges, self_loops=self_loops, max_span=max_span)
    n_ok = n_cyclic_jobs = 0
    for length, mid in [(400, False), (900, True), (1800, False), (1800, True)]:
        reads, seeds = synth.walk_reads(g, 8, length, seed=length + node_len, mid_seed=mid, first_nodes=max(1, len(g.nodes) // 3))
        devs, oras = pc.check_parity(g.nodes, g.edges, reads, seeds, bw, lib_path=lib_path, ctx="cyclic nl%d bw%d len%d" % (node_len, bw, length))
        n_ok += sum(1 for d in devs if d["status"] == 0 and not d["failed"])
        # the same batch again through the batch interface: the jobs must have needed the wide (cycle-capable) variant
        gg = binding.Graph(g.nodes, g.edges, lib_path=lib_path)
        b = gg.prepare(reads, seeds, bw)
        b.run()
        n_cyclic_jobs += b.stats()["jobs_retried"]
    assert n_ok >= 16, n_ok
    assert n_cyclic_jobs > 0


RAMP_CASES = [(8, 5, 35, 0.03), (16, 10, 40, 0.06), (32, 3, 20, 0.03), (64, 15, 80, 0.1)]


def damaged_reads(reads, rng):
    """cut, insert or replace a stretch of 8..180 bases a few times per read: a narrow band loses
    the path there and the reference goes back and widens it (-B, GraphAligner.h:2648-2719)"""
    out = []
    for r in reads:
        r = list(r)
        for _ in range(int(rng.integers(1, 4))):
            if len(r) < 500:
                break
            p = int(rng.integers(200, len(r) - 200))
            kind = int(rng.integers(3))
            n = int(rng.integers(8, 60))
            junk = lambda k: ["ACGT"[int(x)] for x in rng.integers(0, 4, k)]
            if kind == 0:
                del r[p:p + n]
            elif kind == 1:
                r[p:p] = junk(n)
            else:
                r[p:p + n * 3] = junk(n * 3)
        out.append("".join(r))
    return out


def case_ramp_redo(node_len, bw, ramp, err, lib_path=None):
    """-B ramp bandwidth: when the HMM turns "wrong" the reference returns to a remembered slice
    and recomputes with the wide band; its checkpoint list is not rewound with it, so the slices
    its traceback recomputes can differ from the ones it kept (and sometimes assert) -- all of
    which has to come out the same."""
    rng = np.random.default_rng(node_len * 7 + bw)
    g = synth.SynthGraph(synth.random_genome(30000, 900 + node_len), node_len=node_len, snp_every=60, indel_every=400, seed=node_len)
    n_ok = 0
    for length, mid in [(1500, False), (3000, True), (6000, False)]:
        reads, seeds = synth.simulate_reads(g, 8, length, sub=err, ins=err, dele=err, seed=length + bw, mid_seed=mid)
        reads = damaged_reads(reads, rng)
        devs, oras = pc.check_parity(g.nodes, g.edges, reads, seeds, bw, ramp=ramp, lib_path=lib_path, ctx="ramp nl%d bw%d/%d len%d" % (node_len, bw, ramp, length))
        n_ok += sum(1 for d in devs if d["status"] == 0 and not d["failed"])
    assert n_ok >= 12, n_ok


def case_gfa_overlap(lib_path=None):
    """de Bruijn-style GFA: consecutive segments share k bases and the L lines say `kM`.  The loader trims k bases off
    every node (the backward node is cut from the reverse complement of the WHOLE segment, BigraphToDigraph.cpp:58-68) and
    the engine extends the backward part of a read k bases past the seed and keeps k bases of the forward part out of the
    trace (GraphAligner.h:2991, 3054-3095)."""
    import oracle_binding as ob
    rng = np.random.default_rng(3)
    for k, node_len in ((15, 60), (31, 80)):
        gs = synth.random_genome(20000, 70 + k).tobytes().decode()
        step = node_len - k
        segs = []
        i, nid = 0, 1
        while i + node_len <= len(gs):
            segs.append((nid, gs[i:i + node_len]))
            i += step
            nid += 1
        links = [(segs[a][0], False, segs[a + 1][0], False) for a in range(len(segs) - 1)]
        gfa = "H\tVN:Z:1.0\n" + "".join("S\t%d\t%s\n" % x for x in segs) + "".join("L\t%d\t+\t%d\t+\t%dM\n" % (f, t, k) for f, _, t, _ in links)
        reads, seeds = [], []
        noisy = lambda a, b: synth.add_errors(np.frombuffer(gs[a:b].encode(), dtype=np.uint8), 0.03, 0.03, 0.03, rng).tobytes().decode()
        for t in range(10):
            a = int(rng.integers(5, len(segs) - 40))
            if t % 2 == 0:
                reads.append(noisy(a * step, a * step + 1200))
                seeds.append((segs[a][0], 0, False))
            else:
                b = a + 10
                pre = noisy(a * step, b * step)
                reads.append(pre + noisy(b * step, b * step + 700))
                seeds.append((segs[b][0], len(pre), False))
        g = binding.Graph(gfa=gfa, lib_path=lib_path)
        devs = g.align(reads, seeds, 35, 0, flags=binding.GA_F_TRACE)
        plain = g.align(reads, seeds, 35, 0, flags=0)          # (no TraceItem lists: the forward-only assembly path for the seeds at 0)
        og = ob.OracleGraph.from_gfa_segments(segs, links, k)
        n_ok = 0
        for i, (r, sd) in enumerate(zip(reads, seeds)):
            o = og.align(r, [sd], 35)
            pc.compare_read(devs[i], o, "overlap %d read %d" % (k, i))
            pc.compare_read(plain[i], dict(o, trace=np.zeros((0, 7), dtype=np.int64)), "overlap %d read %d without trace items" % (k, i))
            n_ok += int(o["status"] == 0 and not o["failed"])
        assert n_ok >= 8, n_ok


def case_inversion_edges(lib_path=None):
    """bidirected edges that enter a node at its end or leave it from its start (from_start / to_end of vg.Edge,
    BigraphToDigraph.cpp:32-56): every seventh segment is stored reverse-complemented, so reads run through nodes in both
    orientations; seeds on inverted nodes carry reverse = true, and some reads come from the other strand."""
    rng = np.random.default_rng(5)
    gs = synth.random_genome(30000, 99).tobytes().decode()
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda x: "".join(comp[c] for c in reversed(x))
    seg = 40
    pieces = [gs[i:i + seg] for i in range(0, len(gs) - seg, seg)]
    inverted = [i % 7 == 3 for i in range(len(pieces))]
    nodes = [(i + 1, rc(p) if inverted[i] else p) for i, p in enumerate(pieces)]
    edges = [(i + 1, inverted[i], i + 2, inverted[i + 1]) for i in range(len(pieces) - 1)]
    noisy = lambda a, b: synth.add_errors(np.frombuffer(gs[a:b].encode(), dtype=np.uint8), 0.03, 0.03, 0.03, rng).tobytes().decode()
    reads, seeds = [], []
    for t in range(16):
        a = int(rng.integers(3, len(pieces) - 40))
        if t % 2 == 0:
            reads.append(noisy(a * seg, a * seg + 1200))
            seeds.append((a + 1, 0, inverted[a]))
        else:
            b = a + 12
            pre = noisy(a * seg, b * seg)
            reads.append(pre + noisy(b * seg, b * seg + 700))
            seeds.append((b + 1, len(pre), inverted[b]))
        if t % 4 == 3:
            n, pos, _ = seeds[-1]
            reads[-1] = rc(reads[-1])
            seeds[-1] = (n - 1, len(reads[-1]) - pos, not inverted[n - 2])      # the node before, read from the other strand
    devs, oras = pc.check_parity(nodes, edges, reads, seeds, 35, lib_path=lib_path, ctx="inversions")
    assert sum(1 for d in devs if d["status"] == 0 and not d["failed"]) >= 14
    assert {m[1] for d in devs for m in d["mappings"]} == {0, 1}


def case_reference_graph_plumbing(lib_path=None):
    """the reference's own test graph (test/gwws_fail_ex1.vg, decoded fixture) and 10 reads of 256-290 bp cut
    from its longest path with s = i = d = 0.03 (seed 1), seeded at the path's first node; plus the error-free longest-path read,
    whose score SURVEY 8(c) recorded from the real engine (30)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "tests", "golden", "ref_gwws_fail_ex1.json")))
    nodes = [tuple(x) for x in d["nodes"]]
    edges = [tuple(x) for x in d["edges"]]
    path, full = d["longest_path"], d["longest_path_read"]
    rng = np.random.default_rng(1)
    reads, seeds = [full], [(path[0], 0, False)]
    for _ in range(10):
        n = int(rng.integers(256, 291))
        reads.append(synth.add_errors(np.frombuffer(full[:n].encode(), dtype=np.uint8), 0.03, 0.03, 0.03, rng).tobytes().decode())
        seeds.append((path[0], 0, False))
    devs, oras = pc.check_parity(nodes, edges, reads, seeds, 35, lib_path=lib_path, ctx="gwws")
    assert devs[0]["status"] == 0 and devs[0]["score"] == 30
    assert sum(1 for x in devs if x["status"] == 0 and not x["failed"]) >= 8


def case_short_and_edge_reads(lib_path=None):
    """directions shorter than 193 bp hit assert(samplingFrequency > 1) in the reference
    (GraphAligner.h:906); seeds at the last base align backwards only (:3006)"""
    g = synth.bubble_graph(20000, node_len=32, seed=21)
    reads, seeds = synth.simulate_reads(g, 6, 1000, seed=2)
    cases_r, cases_s = [], []
    for r, s in zip(reads, seeds):
        cases_r += [r[:100], r[:192], r[:193], r[:256], r, r, r]
        cases_s += [s, s, s, s, (s[0], 1, s[2]), (s[0], len(r) - 1, s[2]), (s[0], 400, s[2])]
    pc.check_parity(g.nodes, g.edges, cases_r, cases_s, 35, lib_path=lib_path, ctx="edge")
    # degenerate inputs: empty read and seed position past the end (assert(matchSequencePosition < sequence.size()), :2972),
    # a one-base read, a read of N only (matches everything: score 0), a seed on the last base of the other strand
    n0 = g.nodes[3][0]
    tiny_r = ["", "A", "ACGT" * 70, "N" * 400, "ACGT" * 70]
    tiny_s = [(n0, 0, False), (n0, 0, False), (n0, 280, False), (n0, 0, False), (n0, 279, True)]
    devs, _ = pc.check_parity(g.nodes, g.edges, tiny_r, tiny_s, 35, lib_path=lib_path, ctx="degenerate")
    assert [d["status"] for d in devs] == [1, 0, 1, 0, 0] and devs[3]["score"] == 0 and not devs[3]["failed"]


def case_iupac_n_and_invalid_characters(lib_path=None):
    g = synth.bubble_graph(20000, node_len=32, seed=22)
    reads, seeds = synth.simulate_reads(g, 8, 1200, seed=3, mid_seed=True)
    rng = np.random.default_rng(5)
    out = []
    for k, r in enumerate(reads):
        b = bytearray(r.encode())
        for _ in range(30):
            b[int(rng.integers(len(b)))] = ord("NRYKMSWBDVnacgt"[int(rng.integers(15))])
        if k == 5:
            b[700] = ord("H")      # reverse complement of 'H' asserts in the reference (CommonUtils.cpp:128-132)
        if k == 6:
            b[900] = ord("X")      # characterMatch default branch (GraphAligner.h:2104)
        if k == 7:
            b[100] = ord("U")
        out.append(b.decode())
    pc.check_parity(g.nodes, g.edges, out, seeds, 35, lib_path=lib_path, ctx="iupac")


def case_multiple_seeds_per_read(lib_path=None):
    g = synth.bubble_graph(30000, node_len=32, seed=23)
    reads, seeds = synth.simulate_reads(g, 6, 1500, seed=4)
    other, oseeds = synth.simulate_reads(g, 6, 1500, seed=40, mid_seed=True)
    multi = []
    for s, o in zip(seeds, oseeds):
        multi.append([s, (o[0], 700, o[2]), s, (s[0], 0, not s[2])])
    pc.check_parity(g.nodes, g.edges, reads, multi, 35, lib_path=lib_path, ctx="multiseed")


def case_results_without_trace_items(lib_path=None):
    """the same comparisons with flags = 0 (no TraceItem lists: what bench.py and a production caller ask for): the results are
    assembled without the trace-item pass, except for reads with a non-IUPAC character, which the reference's eager TraceItem
    construction (GraphAligner.h:463) still turns into an assertion"""
    g = synth.bubble_graph(30000, node_len=32, seed=27)
    reads, seeds = synth.simulate_reads(g, 10, 1500, seed=6, mid_seed=True)
    other, oseeds = synth.simulate_reads(g, 10, 1500, seed=60)
    b = bytearray(reads[3].encode()); b[400] = ord("X"); reads[3] = b.decode()
    b = bytearray(reads[4].encode()); b[10] = ord("n"); b[900] = ord("R"); reads[4] = b.decode()
    multi = [[s, (o[0], 600, o[2])] if k % 3 == 0 else s for k, (s, o) in enumerate(zip(seeds, oseeds))]
    devs, oras = pc.run_both(g.nodes, g.edges, reads, multi, 35, lib_path=lib_path, trace=False)
    assert any(o["status"] != 0 for o in oras) and any(not o["failed"] for o in oras)
    for i, (d, o) in enumerate(zip(devs, oras)):
        assert d["trace"].shape[0] == 0
        pc.compare_read(d, dict(o, trace=np.zeros((0, 7), dtype=np.int64)), "no trace items, read %d" % i)
    # one seed at the read's first base: the shape the result assembly has its own path for (one forward job, node runs noted while
    # the moves are replayed) -- with IUPAC characters, a read too short to align, a damaged read that stops early, 1-bp SNP nodes
    for graph in (g, synth.SynthGraph(synth.random_genome(20000, 71), node_len=16, snp_every=30, indel_every=300, seed=72), synth.linear_graph(20000, node_len=64, seed=73)):
        reads, seeds = synth.simulate_reads(graph, 12, 1800, seed=9)
        b = bytearray(reads[1].encode()); b[5] = ord("N"); b[700] = ord("y"); reads[1] = b.decode()
        reads[2] = reads[2][:150]
        reads[5] = damaged_reads([reads[5]], np.random.default_rng(3))[0]
        b = bytearray(reads[7].encode()); b[900] = ord("X"); reads[7] = b.decode()
        devs, oras = pc.run_both(graph.nodes, graph.edges, reads, seeds, 35, lib_path=lib_path, trace=False)
        for i, (d, o) in enumerate(zip(devs, oras)):
            pc.compare_read(d, dict(o, trace=np.zeros((0, 7), dtype=np.int64)), "forward only, read %d" % i)
        # ... and without the read that has a non-IUPAC character the whole batch qualifies for node runs instead of moves from the
        # traceback when the graph's nodes are long (ga_batch_stats.reserved says so)
        clean = [r for k, r in enumerate(reads) if k != 7]
        cseeds = [s for k, s in enumerate(seeds) if k != 7]
        gg = binding.Graph(graph.nodes, graph.edges, lib_path=lib_path)
        batch = gg.prepare(clean, cseeds, 35, 0, flags=0)
        batch.run()
        # (node runs only on graphs of long nodes, mean >= 40 bp: the 64-bp chain here; the other two keep moves)
        assert batch.stats()["reserved"] == (1 if all(len(seq) >= 40 for _, seq in graph.nodes[:-1]) else 0)
        og = ob.OracleGraph(graph.nodes, graph.edges)
        for i, d in enumerate(batch.collect()):
            pc.compare_read(d, dict(og.align(clean[i], [cseeds[i]], 35), trace=np.zeros((0, 7), dtype=np.int64)), "node runs, read %d" % i)


def case_unknown_seed_node_reports_bad_seed(lib_path=None):
    g = synth.linear_graph(5000, node_len=64, seed=1)
    reads, seeds = synth.simulate_reads(g, 2, 600, seed=1)
    gg = binding.Graph(g.nodes, g.edges, lib_path=lib_path)
    res = gg.align(reads, [(10 ** 6, 0, False), seeds[1]], 35)
    assert res[0]["status"] == 3 and res[0]["failed"]
    assert res[1]["status"] == 0 and not res[1]["failed"]


def case_gfa_loader_matches_node_edge_api(lib_path=None):
    g = synth.bubble_graph(15000, node_len=32, seed=24)
    reads, seeds = synth.simulate_reads(g, 6, 1200, seed=8)
    a = binding.Graph(g.nodes, g.edges, lib_path=lib_path).align(reads, seeds, 35)
    b = binding.Graph(gfa=g.gfa(), lib_path=lib_path).align(reads, seeds, 35)
    for x, y in zip(a, b):
        assert x["score"] == y["score"] and x["mappings"] == y["mappings"]


def case_full_size_properties(lib_path=None):
    """at benchmark read length the oracle is too slow to sweep, so check size-independent
    properties on 10 kb reads: determinism, score == edits implied by the trace, path
    continuity, and agreement with the oracle on a small subset"""
    g = synth.bubble_graph(400000, node_len=64, seed=44)
    reads, seeds = synth.simulate_reads(g, 128, 10000, seed=46)
    gg = binding.Graph(g.nodes, g.edges, lib_path=lib_path)
    r1 = gg.align(reads, seeds, 35, flags=binding.GA_F_TRACE)
    r2 = gg.align(reads, seeds, 35, flags=binding.GA_F_TRACE)
    for a, b in zip(r1, r2):
        assert a["score"] == b["score"] and a["mappings"] == b["mappings"]
    n_ok = 0
    for r, read in zip(r1, reads):
        if r["failed"]:
            continue
        n_ok += 1
        t = r["trace"]
        # every mismatch / insertion / deletion costs one; padded N rows past the read end are free or insertions
        edits = int(((t[:, 4] == 2) | (t[:, 4] == 3) | (t[:, 4] == 4)).sum())
        assert edits <= r["score"] <= edits + 64 + 1
        assert (np.diff(t[:, 3]) >= 0).all()
        assert sum(m[5] for m in r["mappings"]) <= len(read)
    assert n_ok >= 120
    import oracle_binding as ob
    og = ob.OracleGraph(g.nodes, g.edges)
    for i in range(0, 128, 16):
        pc.compare_read(r1[i], og.align(reads[i], [seeds[i]], 35), "full-size %d" % i)


SPARSE_FANS = [(8, 30000, 150, 600, 35, 0), (8, 30000, 400, 1000, 10, 0), (5, 50000, 100, 333, 35, 60), (12, 20000, 200, 500, 20, 45), (40, 6000, 250, 400, 35, 0)]


def case_sparse_method_and_override(branches, branch_len, shared, stem, bw, ramp, lib_path=None):
    """bands of 200 000 cells and more: the reference leaves its bit vectors for calculateSliceAlternate (GraphAligner.h:2148-2329)
    and keeps the window's slices to work the traceback out at once (BacktraceOverride, :167-354).  Fan graphs (synth.FanGraph): a
    stem that ends in many long branches with a common beginning, so that the projected band holds all of them.  Reads through the
    fan forwards (sparse slices between bit-vector ones, in both orders), with a seed past the fan (the backward part meets it from the
    other strand: a many-to-one join), and the sharp edges the reference has here."""
    import oracle_binding as ob
    g = synth.FanGraph(head_len=200, stem_len=stem, n_branches=branches, branch_len=branch_len, shared=shared, seed=branches)
    rng = np.random.default_rng(branches * 11 + bw)
    reads, seeds = [], []
    for k in range(4):
        r, s = g.read_through(int(rng.integers(0, branches)), 0, 1400 + 450 * k, rng)
        reads.append(r); seeds.append(s)
    # seeds inside a branch, 600-1100 bases into it: the backward part runs back over the branch's start, the stem and the head
    for k in range(3):
        b = int(rng.integers(0, branches))
        depth = 600 + 250 * k
        path = np.concatenate([g.head, g.stem, g.branches[b][:depth + 900]])
        pre = synth.add_errors(path[:len(g.head) + len(g.stem) + depth], 0.03, 0.03, 0.03, rng).tobytes().decode()
        post = synth.add_errors(path[len(g.head) + len(g.stem) + depth:], 0.03, 0.03, 0.03, rng).tobytes().decode()
        reads.append(pre + post); seeds.append((3 + b, len(pre), False))
    devs, oras = pc.check_parity(g.nodes, g.edges, reads, seeds, bw, ramp=ramp, lib_path=lib_path, ctx="fan %d x %d bw%d/%d" % (branches, branch_len, bw, ramp))
    n_sparse = sum(o["sparse_slices"] for o in oras)
    n_windows = sum(o["override_traces"] for o in oras)
    n_ok = sum(1 for d in devs if d["status"] == 0 and not d["failed"])
    assert n_sparse >= 8 and n_windows >= 1 and n_ok >= 3, (n_sparse, n_windows, n_ok)
    return devs, oras


def case_sparse_sharp_edges(lib_path=None):
    """what the reference does around the sparse method that is not an alignment: undefined behaviour at bandwidth 0, the assertion
    when an override window starts at slice 0, and the 16-bit frozen scores that a long node's untouched columns do not fit"""
    g = synth.FanGraph(head_len=200, stem_len=800, n_branches=8, branch_len=30000, shared=300, seed=3)
    rng = np.random.default_rng(1)
    read = synth.add_errors(np.concatenate([g.stem, g.branches[2]])[:1500], 0.03, 0.03, 0.03, rng).tobytes().decode()
    # a seed on the stem: slice 0 projects every branch and goes sparse at the ramp width (slice-0 quirk, :2612)
    devs, oras = pc.check_parity(g.nodes, g.edges, [read, read], [(2, 0, False), (2, 0, False)], 35, ramp=0, lib_path=lib_path, ctx="sparse at bandwidth 0")
    assert oras[0]["status"] == 1 and devs[0]["status"] == 1
    devs, oras = pc.check_parity(g.nodes, g.edges, [read], [(2, 0, False)], 35, ramp=70, lib_path=lib_path, ctx="override window at slice 0")
    assert oras[0]["status"] == 1 and "overrideLastJ" in oras[0]["message"]
    # one node of 209 100 bp behind two short ones (the literal single-contig case, unsplit): the sparse slice that reaches it leaves
    # its untouched columns at minimum + length + bandwidth + 1 (:2546-2549) -- assert(... < 65535) in the frozen slice (NodeSlice.h:372)
    contig = synth.random_genome(210000, 77)
    nodes = [(1, contig[:300].tobytes().decode()), (2, contig[300:900].tobytes().decode()), (3, contig[900:].tobytes().decode())]
    edges = [(1, False, 2, False), (2, False, 3, False)]
    reads = [synth.add_errors(contig[a:a + 1500], 0.03, 0.03, 0.03, rng).tobytes().decode() for a in (0, 100)]
    devs, oras = pc.check_parity(nodes, edges, reads, [(1, 0, False), (1, 100, False)], 35, ramp=0, lib_path=lib_path, ctx="a 209 kbp node")
    assert all(o["status"] == 1 and "65535" in o["message"] for o in oras)
    # ... and a seed ON that node without a ramp width: bandwidth 0 again
    devs, oras = pc.check_parity(nodes, edges, [synth.add_errors(contig[5000:6500], 0.03, 0.03, 0.03, rng).tobytes().decode()], [(3, 0, False)], 35, ramp=0, lib_path=lib_path, ctx="seed on a 209 kbp node")
    assert oras[0]["status"] == 1 and devs[0]["status"] == 1


def case_long_reads_on_short_nodes(lib_path=None):
    """50 kb reads on a chain of 8-bp nodes (the C5 shape at a size the oracle covers): bands of ~30 nodes, so the lanes = reads
    kernel runs as <56,16> with arenas of millions of rows -- block numbers past 2^19, one block per node, lanes without a node in a
    round (their staged image goes to the spare block behind the arena)"""
    g = synth.linear_graph(400000, node_len=8, seed=5)
    reads, seeds = synth.simulate_reads(g, 6, 50000, seed=6)
    for trace in (True, False):
        devs, oras = pc.run_both(g.nodes, g.edges, reads, seeds, 35, lib_path=lib_path, trace=trace)
        for i, (d, o) in enumerate(zip(devs, oras)):
            pc.compare_read(d, o if trace else dict(o, trace=np.zeros((0, 7), dtype=np.int64)), "50 kb read %d on 8-bp nodes" % i)
    assert all(o["status"] == 0 for o in oras)


def case_batch_run_twice(lib_path=None):
    """a prepared batch can be run again (bench.py alternates two): every pass must start from scratch -- the second run's results equal
    the first's and the oracle's, on a long-node and on a short-node graph (lanes = reads first / wave per read first)"""
    from graphaligner_amd import binding
    for graph in (synth.linear_graph(20000, node_len=64, seed=81), synth.SynthGraph(synth.random_genome(20000, 82), node_len=16, snp_every=30, indel_every=300, seed=83)):
        reads, seeds = synth.simulate_reads(graph, 12, 1500, seed=84)
        g = binding.Graph(graph.nodes, graph.edges, lib_path=lib_path)
        og = ob.OracleGraph(graph.nodes, graph.edges)
        for flags in (binding.GA_F_TRACE, 0):
            b = g.prepare(reads, [[s] for s in seeds], 35, 0, flags)
            b.run()
            first = b.collect()
            b.run()
            second = b.collect()
            for i, (x, y) in enumerate(zip(first, second)):
                assert x["status"] == y["status"] and x["score"] == y["score"] and x["mappings"] == y["mappings"], ("second run differs", i)
                o = og.align(reads[i], [seeds[i]], 35)
                pc.compare_read(y, o if flags else dict(o, trace=np.zeros((0, 7), dtype=np.int64)), "second run, read %d" % i)


def case_trace_pool_overflow(lib_path=None, monkeypatch=None):
    """a trace pool far too small for the batch: jobs that find no room report a capacity miss; every job that reports success has
    its moves intact (claims commit only when they fit, so no two regions overlap) and equals the oracle"""
    import os
    g = synth.linear_graph(60000, node_len=64, seed=5)
    reads, seeds = synth.simulate_reads(g, 96, 1500, seed=21)
    os.environ["GA_TEST_TRACE_POOL_BYTES"] = str(40000)
    try:
        devs, oras = pc.run_both(g.nodes, g.edges, reads, seeds, 35, lib_path=lib_path)
    finally:
        del os.environ["GA_TEST_TRACE_POOL_BYTES"]
    n_ok = n_cap = 0
    for i, (d, o) in enumerate(zip(devs, oras)):
        if d["status"] == 10:
            n_cap += 1
            continue
        pc.compare_read(d, o, "small trace pool, read %d" % i)
        n_ok += 1
    assert n_ok >= 8 and n_cap >= 8, (n_ok, n_cap)
