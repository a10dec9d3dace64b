"""File formats either side of the hot path: vg.Graph chunks and seed GAM in,
GAM out -- implemented on zlib alone.  Checked against the reference's own binary data files
(tests/golden/ref_*.vg, *.gam: copies of /root/reference/test/...) and by round trip through an
independent decoder (tools/make_golden.py)."""
import json
import os
import sys

import numpy as np

from graphaligner_amd import binding, synth
import parity_common as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg          # the independent Python decoder


def test_vg_loader_matches_decoded_fixture():
    lib = pc.emul_lib_path()
    for vg, js in (("ref_gwws_fail_ex1.vg", "ref_gwws_fail_ex1.json"), ("ref_smallexample_sub_test.vg", "ref_smallexample.json")):
        data = open(os.path.join(GOLDEN, vg), "rb").read()
        d = json.load(open(os.path.join(GOLDEN, js)))
        a = binding.Graph(vg=data, lib_path=lib)
        b = binding.Graph([tuple(x) for x in d["nodes"]], [tuple(x) for x in d["edges"]], lib_path=lib)
        assert a.node_count == b.node_count == 2 * len(d["nodes"]) + 2
        assert a.bp == b.bp
    d = json.load(open(os.path.join(GOLDEN, "ref_gwws_fail_ex1.json")))
    g = binding.Graph(vg=open(os.path.join(GOLDEN, "ref_gwws_fail_ex1.vg"), "rb").read(), lib_path=lib)
    r = g.align([d["longest_path_read"]], [(d["longest_path"][0], 0, False)], 35)[0]
    assert r["status"] == 0 and r["score"] == 30


def test_seed_gam_decoder_on_the_reference_fixture():
    lib = pc.emul_lib_path()
    data = open(os.path.join(GOLDEN, "ref_smallexample_seedalignment.gam"), "rb").read()
    assert binding.decode_seed_gam(data, lib) == [("read1", (6738, 0, False))]


def test_gam_writer_round_trip():
    lib = pc.emul_lib_path()
    g = synth.bubble_graph(15000, node_len=32, seed=61)
    reads, seeds = synth.simulate_reads(g, 6, 900, seed=62, mid_seed=True)
    reads[2] = reads[2][:120]       # fails -> not written
    graph = binding.Graph(g.nodes, g.edges, lib_path=lib)
    batch = graph.prepare(reads, seeds, 35)
    batch.run()
    res = batch.collect()
    gam = batch.collect_gam(halve_node_ids=True)
    tmp = os.path.join(ROOT, "tests", "_build", "roundtrip.gam")
    open(tmp, "wb").write(gam)
    msgs = list(mg._messages(tmp))
    ok = [i for i, r in enumerate(res) if not r["failed"]]
    assert len(msgs) == len(ok) == 5
    for m, i in zip(msgs, ok):
        top = {}
        for f, v in mg._fields(m):
            top.setdefault(f, v)
        assert top[1].decode() == reads[i] and top[3].decode() == "read%d" % i
        assert top.get(6, 0) == res[i]["score"] and top.get(7, 0) == res[i]["query_position"]
        maps = []
        for f, v in mg._fields(top[2]):
            if f != 2:
                continue
            md = {}
            for f2, v2 in mg._fields(v):
                md[f2] = v2
            pos = dict(mg._fields(md[1]))
            ed = dict(mg._fields(md[2]))
            maps.append((pos.get(1, 0), pos.get(4, 0), pos.get(2, 0), md.get(5, 0), ed.get(1, 0), ed.get(2, 0), ed.get(3, b"").decode()))
        want = [(m_[0] // 2, m_[1], m_[2], m_[3], m_[4], m_[5], m_[6]) for m_ in res[i]["mappings"]]
        assert maps == want


def test_vg_loader_on_chunked_graph_with_reverse_edges():
    """a synthetic graph written as several gzip-framed groups of vg.Graph chunks (stream.hpp:24-118: varint count, then
    varint length + message; nodes and edges spread over chunks, edges with from_start / to_end), by an encoder
    that shares nothing with the library; the loaded graph must align like the one built through the node / edge calls"""
    import gzip
    from graphaligner_amd import aligner as al
    import parity_cases as cases
    lib = pc.emul_lib_path()
    g = synth.bubble_graph(20000, node_len=32, seed=61)
    edges = list(g.edges)
    # flip a few segments so that from_start / to_end appear
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    flip = {nid for k, (nid, _) in enumerate(g.nodes) if k % 11 == 5}
    nodes = [(nid, "".join(comp[c] for c in reversed(s)) if nid in flip else s) for nid, s in g.nodes]
    edges = [(f, (fs != (f in flip)), t, (te != (t in flip))) for f, fs, t, te in edges]

    def node_msg(nid, seq):
        return al._bytes_field(1, seq.encode()) + al._int_field(3, nid)

    def edge_msg(f, fs, t, te):
        return al._int_field(1, f) + al._int_field(2, t) + al._int_field(3, int(fs)) + al._int_field(4, int(te))

    chunks = []
    per = 300
    for lo in range(0, len(nodes), per):
        body = b"".join(al._message(1, node_msg(n, s)) for n, s in nodes[lo:lo + per])
        chunks.append(body)
    for lo in range(0, len(edges), 500):
        chunks.append(b"".join(al._message(2, edge_msg(*e)) for e in edges[lo:lo + 500]))
    data = b""
    for lo in range(0, len(chunks), 3):                     # three Graph messages per group, one gzip member per group
        grp = chunks[lo:lo + 3]
        data += gzip.compress(al._varint(len(grp)) + b"".join(al._varint(len(c)) + c for c in grp))
    a = binding.Graph(vg=data, lib_path=lib)
    b = binding.Graph(nodes, edges, lib_path=lib)
    assert a.node_count == b.node_count and a.bp == b.bp
    reads, seeds = synth.simulate_reads(g, 10, 1000, seed=8, mid_seed=True)
    seeds = [(n, p, (r != (n in flip))) for n, p, r in seeds]
    ra = a.align(reads, seeds, 35)
    rb = b.align(reads, seeds, 35)
    for x, y in zip(ra, rb):
        assert x["status"] == y["status"] and x["score"] == y["score"] and x["mappings"] == y["mappings"]
    assert sum(1 for x in ra if x["status"] == 0 and not x["failed"]) >= 8
    devs, oras = pc.check_parity(nodes, edges, reads, seeds, 35, lib_path=lib, ctx="vg chunks")
