"""The BASELINE.json configurations, each exercised on the real MI355X through the C ABI:

  C3  yeast-like pangenome GFA (several chromosomes, SNP / indel / SV bubbles, nodes <= 64 bp), 10 kb ONT-error reads,
      seeds in the MIDDLE of the read (backward + forward extension, GraphAligner.h:2969-3024)
  C4  chr22-like graph (32-bp nodes, a SNP every ~45 bp) written as gzip-framed vg.Graph chunks and loaded with
      ga_graph_load_vg, 15 kb PacBio-CLR-error reads (s=.02, i=.08, d=.05)
  C5  whole-genome SHAPE: a graph with more than 2^27 directed nodes (built natively on the box), 50 kb reads placed beyond
      node index 2^27, checked against the oracle on the local window of the graph

Sizes are scaled down from the BASELINE read counts (the oracle is the slow side); every read the device reports failed, and
every read that was not finished by the first kernel pass, is in the subset compared with the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from graphaligner_amd import binding, synth
import oracle_binding as ob
import parity_common as pc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need a real MI355X"


@pytest.fixture(autouse=True, params=["lanes-first", "by-graph-shape"])
def _first_pass(request, monkeypatch):
    """both first-pass kernels on every configuration: the lanes = reads kernel forced, and the library's own choice by graph shape"""
    if request.param == "lanes-first":
        monkeypatch.setenv("GA_LANES", "1")
    else:
        monkeypatch.delenv("GA_LANES", raising=False)


def _properties(res, reads, seeds, min_ok):
    """size-independent checks on every read: rows monotone, path length bounded by the read, and -- for reads seeded at their
    first base, where the reference's TraceItem types are meaningful (the items of a backward part are typed after the trace was
    mirrored, and the oracle shows the same excess there) -- the score consistent with the edits of the trace"""
    n_ok = 0
    for r, read, sd in zip(res, reads, seeds):
        if r["failed"]:
            continue
        n_ok += 1
        t = r["trace"]
        if sd[1] == 0:
            edits = int(((t[:, 4] == 2) | (t[:, 4] == 3) | (t[:, 4] == 4)).sum())
            assert edits <= r["score"] <= edits + 64 + 1, (edits, r["score"])
        assert (np.diff(t[t[:, 4] != 5][:, 3]) >= 0).all()
        assert sum(m[5] for m in r["mappings"]) <= len(read)
        assert 0 <= r["score"] < len(read)
    assert n_ok >= min_ok, (n_ok, min_ok)


def _compare_subset(res, og, reads, seeds, bw, every, cap, ctx):
    """oracle comparison for a regular sample plus every failed read and every read a later kernel pass had to finish"""
    special = [i for i, r in enumerate(res) if r["failed"] or r["status"] != 0 or r["kernel_pass"] != 0]
    pick = sorted(set(list(range(0, len(res), every)) + special[:cap]))
    for i in pick:
        pc.compare_read(res[i], og.align(reads[i], [seeds[i]], bw), "%s read %d" % (ctx, i))
    return len(pick), len(special)


def test_c3_pangenome_gfa_mid_seeds():
    g = synth.pangenome_graph(2400000, chromosomes=16, node_len=64, seed=44)
    mid_r, mid_s = synth.simulate_reads_multi(g, 96, 10000, seed=46, mid_seed=True)
    fw_r, fw_s = synth.simulate_reads_multi(g, 32, 10000, seed=47)
    reads, seeds = mid_r + fw_r, mid_s + fw_s
    gg = binding.Graph(gfa=g.gfa())
    res = gg.align(reads, seeds, 35, flags=binding.GA_F_TRACE)
    again = gg.align(reads, seeds, 35, flags=binding.GA_F_TRACE)
    for a, b in zip(res, again):
        assert a["score"] == b["score"] and a["mappings"] == b["mappings"]
    _properties(res, reads, seeds, 120)
    # the mid-seeded reads really have a backward part: the alignment starts before the seed position
    assert sum(1 for r, s in zip(res[:96], mid_s) if not r["failed"] and r["query_position"] < s[1]) >= 80
    og = ob.OracleGraph(g.nodes, g.edges)
    n, special = _compare_subset(res, og, reads, seeds, 35, 10, 16, "C3")
    assert n >= 13


def test_c4_chr22_like_vg_clr_reads():
    g = synth.SynthGraph(synth.random_genome(3000000, 47), node_len=32, snp_every=45, indel_every=500, seed=48)
    data = g.vg_bytes(chunk_nodes=1000)
    reads, seeds = synth.simulate_reads(g, 72, 15000, sub=0.02, ins=0.08, dele=0.05, seed=49)
    mid_r, mid_s = synth.simulate_reads(g, 24, 15000, sub=0.02, ins=0.08, dele=0.05, seed=50, mid_seed=True)
    reads, seeds = reads + mid_r, seeds + mid_s
    gg = binding.Graph(vg=data)
    assert gg.node_count == 2 * len(g.nodes) + 2
    res = gg.align(reads, seeds, 35, flags=binding.GA_F_TRACE)
    _properties(res, reads, seeds, 80)
    og = ob.OracleGraph(g.nodes, g.edges)
    n, special = _compare_subset(res, og, reads, seeds, 35, 8, 24, "C4")
    assert n >= 12


def _scalegen():
    so = os.path.join(ROOT, "tests", "_build", "libga_scalegen.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "native")])
    SG = C.CDLL(so)
    SG.ga_scalegen_chain.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    SG.ga_scalegen_region.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p]
    return SG


def test_c5_shape_more_than_2_27_directed_nodes():
    """node records past index 2^27 (64-bit record addressing), 50 kb reads, both strands"""
    import psutil
    if psutil.virtual_memory().available < 60 * 2 ** 30:
        pytest.skip("needs ~40 GB of host memory for the graph")
    SG = _scalegen()
    L = binding.load()
    seed, node_len = 50, 8
    n_nodes = (1 << 26) + 150000                   # bidirected; 2 * n + 2 directed nodes > 2^27
    gg = object.__new__(binding.Graph)
    gg.L = L
    gg.h = L.ga_graph_create()
    assert SG.ga_scalegen_chain(gg.h, seed, n_nodes, node_len) == 0
    assert L.ga_graph_node_count(gg.h) == 2 * n_nodes + 2 > (1 << 27)
    binding._check(L, L.ga_graph_upload(gg.h, 0), "ga_graph_upload")
    rng = np.random.default_rng(52)
    total = n_nodes * node_len
    read_len = 50000
    starts = [total - 70000, total - 200000, total - 130000, 64, total // 2, total - 400000]
    reads, seeds, windows = [], [], []
    for k, st in enumerate(starts):
        st -= st % node_len
        buf = C.create_string_buffer(read_len)
        SG.ga_scalegen_region(seed, st, read_len, buf)
        body = np.frombuffer(buf.raw, dtype=np.uint8)
        if k % 2 == 0:
            r = synth.add_errors(body, 0.04, 0.04, 0.04, rng)
            reads.append(r.tobytes().decode())
            seeds.append((st // node_len + 1, 0, False))
        else:
            r = synth.add_errors(synth.revcomp_bytes(body), 0.04, 0.04, 0.04, rng)
            reads.append(r.tobytes().decode())
            seeds.append(((st + read_len - 1) // node_len + 1, 0, True))
        windows.append(st)
    res = gg.align(reads, seeds, 35, flags=binding.GA_F_TRACE)
    _properties(res, reads, seeds, len(reads))
    for i, st in enumerate(windows):
        # the same stretch of the chain as a small graph for the oracle (node ids kept); the band never comes near its ends
        lo = max(0, st - 4096) // node_len
        hi = min(n_nodes, (st + read_len + 8192) // node_len)
        buf = C.create_string_buffer((hi - lo) * node_len)
        SG.ga_scalegen_region(seed, lo * node_len, (hi - lo) * node_len, buf)
        text = buf.raw.decode()
        nodes = [(lo + 1 + j, text[j * node_len:(j + 1) * node_len]) for j in range(hi - lo)]
        edges = [(lo + 1 + j, False, lo + 2 + j, False) for j in range(hi - lo - 1)]
        og = ob.OracleGraph(nodes, edges)
        pc.compare_read(res[i], og.align(reads[i], [seeds[i]], 35), "C5 read %d" % i)


def test_single_contig_gfa_needs_split_then_matches_oracle_of_cut_graph():
    """a GFA with one 300 kbp segment.  Unsplit, slice 0 already holds >= 200 000 bp and the reference hands it to its
    sparse method at the ramp width (slice-0 quirk) -- 0 without -B, where calculateSliceAlternate indexes past its bucket list
    (GraphAligner.h:2220): both sides report an assertion.  Loaded with split=64 the bit-vector path applies, results equal the oracle's
    on the hand-cut chain and ga_results_unsplit names the file's segment again."""
    rng = np.random.default_rng(91)
    n, cut = 300000, 64
    contig = "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    gfa = "H\tVN:Z:1.0\nS\t1\t%s\n" % contig
    reads, seeds, starts = [], [], []
    for k in range(96):
        st = int(rng.integers(0, n - 6000)) // cut * cut
        body = np.frombuffer(contig[st:st + 5000].encode(), dtype=np.uint8)
        reads.append(synth.add_errors(body, 0.03, 0.03, 0.03, rng).tobytes().decode())
        starts.append(st)
    whole = binding.Graph(gfa=gfa)
    res = whole.align(reads[:4], [(1, 0, False)] * 4, 35)
    whole_oracle = ob.OracleGraph([(1, contig)], [])
    for r, x in zip(res, reads[:4]):
        o = whole_oracle.align(x, [(1, 0, False)], 35)
        assert o["status"] == 1 and "bandwidth 0" in o["message"]
        pc.compare_read(r, o, "unsplit single contig")
    pieces = [(1 if j == 0 else 1 + j, contig[j * cut:(j + 1) * cut]) for j in range((n + cut - 1) // cut)]
    edges = [(pieces[j][0], False, pieces[j + 1][0], False) for j in range(len(pieces) - 1)]
    split = binding.Graph(gfa=gfa, split=cut)
    assert split.node_count == 2 * len(pieces) + 2
    seeds = [(pieces[st // cut][0], 0, False) for st in starts]
    b = split.prepare(reads, seeds, 35, 0, flags=binding.GA_F_TRACE)
    b.run()
    res = b.collect()
    og = ob.OracleGraph(pieces, edges)
    _properties(res, reads, seeds, len(reads))
    for i in range(0, len(reads), 6):
        pc.compare_read(res[i], og.align(reads[i], [seeds[i]], 35), "split read %d" % i)
    merged = b.collect(unsplit=True)
    for r, m, st in zip(res, merged, starts):
        assert len(m["mappings"]) == 1 and m["mappings"][0][0] == 2 and m["score"] == r["score"]
        assert m["mappings"][0][5] == sum(x[5] for x in r["mappings"])
        assert abs(m["mappings"][0][2] - st) <= 64
        assert (m["trace"][:, 0] == 1).all() and (np.diff(m["trace"][m["trace"][:, 4] != 5][:, 2].astype(np.int64)) >= 0).all()


def test_c2_full_length_reads_on_the_64bp_chain():
    """BASELINE configuration 2 at full READ length: 10 kb ONT-error reads on a chain of 64-bp nodes (a 1.2 Mbp stretch of the genome
    bench.py uses, band 35, one seed at the first base) -- 256 reads, EVERY one compared with the oracle, with and without TraceItem
    lists (the second is the path bench.py times: node runs straight from the traceback), under both first-pass choices"""
    g = synth.linear_graph(1200000, node_len=64, seed=42)
    reads, seeds = synth.simulate_reads(g, 256, 10000, sub=0.04, ins=0.04, dele=0.04, seed=43)
    og = ob.OracleGraph(g.nodes, g.edges)
    graph = binding.Graph(gfa=g.gfa())
    with_items = graph.align(reads, seeds, 35, 0, flags=binding.GA_F_TRACE)
    plain = graph.align(reads, seeds, 35, 0, flags=0)
    n_ok = 0
    for i, (r, s) in enumerate(zip(reads, seeds)):
        o = og.align(r, [s], 35)
        pc.compare_read(with_items[i], o, "C2 read %d" % i)
        pc.compare_read(plain[i], dict(o, trace=np.zeros((0, 7), dtype=np.int64)), "C2 read %d without trace items" % i)
        n_ok += 0 if o["failed"] else 1
    assert n_ok >= 250
    _properties(with_items, reads, seeds, 250)
