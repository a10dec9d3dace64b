"""The C ABI library must load and export every function include/graphaligner_amd.h declares.
No compute calls are made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "graphaligner_amd.h")
LIB = os.path.join(ROOT, "graphaligner_amd", "libgraphaligner_amd.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ga_[a-z_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ["ga_graph_create", "ga_graph_add_node", "ga_graph_add_edge", "ga_graph_finalize", "ga_graph_upload", "ga_align_batch",
                 "ga_results_free", "ga_batch_prepare", "ga_batch_run", "ga_batch_collect"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__ as entry
        entry.build_product()
    lib = ctypes.CDLL(LIB)
    for name in declared_functions():
        assert hasattr(lib, name), "libgraphaligner_amd.so does not export %s" % name


def test_no_cpu_fallback_without_device():
    """graph upload must fail loudly when no GPU is usable (this container has none)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from graphaligner_amd import binding
    with pytest.raises(RuntimeError):
        binding.Graph([(1, "ACGTACGTAC")], [])


def test_product_library_does_not_contain_the_oracle_or_the_emulation():
    data = open(LIB, "rb").read()
    assert b"gao_align" not in data and b"ga_emul_" not in data


def test_digraph_level_calls_and_one_shot_entry_point():
    """the calls the reference-side binding of INTEGRATION.md uses: ga_graph_add_node / ga_graph_add_edge with digraph ids
    (AlignmentGraph::AddNode / AddEdgeNodeId) and the one-shot ga_align_batch, against the bigraph calls + batch interface.
    Runs the device program through the host emulation (this container has no GPU)."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphaligner_amd import binding, synth
    import parity_common as pc
    lib = pc.emul_lib_path()
    L = binding.load(lib)
    g = synth.bubble_graph(15000, node_len=32, seed=31)
    reads, seeds = synth.simulate_reads(g, 6, 900, seed=12, mid_seed=True)
    ref = binding.Graph(g.nodes, g.edges, lib_path=lib).align(reads, seeds, 35)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    h = L.ga_graph_create()
    for nid, seq in g.nodes:
        rc = "".join(comp[c] for c in reversed(seq))
        assert L.ga_graph_add_node(h, 2 * nid, seq.encode(), len(seq), 0) == 0
        assert L.ga_graph_add_node(h, 2 * nid + 1, rc.encode(), len(rc), 1) == 0
    for f, fs, t, te in g.edges:
        # BigraphToDigraph.cpp:32-56: right end of `from` -> right end of `to`, and the mirrored edge
        f_left, f_right = (2 * f, 2 * f + 1) if fs else (2 * f + 1, 2 * f)
        t_left, t_right = (2 * t, 2 * t + 1) if te else (2 * t + 1, 2 * t)
        assert L.ga_graph_add_edge(h, f_right, t_right) == 0
        assert L.ga_graph_add_edge(h, t_left, f_left) == 0
    assert L.ga_graph_finalize(h, 0) == 0
    assert L.ga_graph_upload(h, 0) == 0
    n = len(reads)
    keep = [r.encode() for r in reads]
    arr = (binding.GaRead * n)()
    for i in range(n):
        arr[i].name = b"r%d" % i
        arr[i].sequence = keep[i]
        arr[i].length = len(keep[i])
    sarr = (binding.GaSeed * n)()
    offs = (C.c_size_t * (n + 1))(*range(n + 1))
    for i, (node, pos, rev) in enumerate(seeds):
        sarr[i].node_id, sarr[i].read_pos, sarr[i].reverse = int(node), int(pos), int(bool(rev))
    out = C.POINTER(binding.GaResults)()
    assert L.ga_align_batch(h, arr, n, sarr, offs, 35, 0, 0, C.byref(out)) == 0
    got = binding._unpack(out.contents)
    L.ga_results_free(out)
    L.ga_graph_destroy(h)
    for a, b in zip(ref, got):
        assert a["status"] == b["status"] and a["score"] == b["score"] and a["mappings"] == b["mappings"]
    assert sum(1 for a in got if a["status"] == 0 and not a["failed"]) >= 5


def test_finished_graph_mirrored_verbatim():
    """ga_graph_set_neighbors: the reference-side binding copies a FINISHED AlignmentGraph -- inNeighbors and outNeighbors of every
    node as they stand.  The edges here are added in an order that is not sorted by target (every node's lists end up in an order a
    replay grouped by target cannot give), once through AddEdgeNodeId-style calls and once as finished lists: same results, and a
    replay of the in-lists through ga_graph_add_edge (what INTEGRATION.md used to print) gives different out-lists."""
    import ctypes as C
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphaligner_amd import binding, synth
    import parity_common as pc
    lib = pc.emul_lib_path()
    L = binding.load(lib)
    g = synth.SynthGraph(synth.random_genome(12000, 3), node_len=16, snp_every=25, indel_every=120, seed=4)
    reads, seeds = synth.simulate_reads(g, 8, 800, seed=6)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rng = np.random.default_rng(2)
    digraph_edges = []
    for f, fs, t, te in g.edges:
        f_left, f_right = (2 * f, 2 * f + 1) if fs else (2 * f + 1, 2 * f)
        t_left, t_right = (2 * t, 2 * t + 1) if te else (2 * t + 1, 2 * t)
        digraph_edges += [(f_right, t_right), (t_left, f_left)]
    order = rng.permutation(len(digraph_edges))
    digraph_edges = [digraph_edges[i] for i in order]              # "file order": anything but sorted by target

    def nodes_into(h):
        for nid, seq in g.nodes:
            rc = "".join(comp[c] for c in reversed(seq))
            assert L.ga_graph_add_node(h, 2 * nid, seq.encode(), len(seq), 0) == 0
            assert L.ga_graph_add_node(h, 2 * nid + 1, rc.encode(), len(rc), 1) == 0

    def run(h):
        assert L.ga_graph_finalize(h, 0) == 0 and L.ga_graph_upload(h, 0) == 0
        n = len(reads)
        keep = [r.encode() for r in reads]
        arr = (binding.GaRead * n)()
        for i in range(n):
            arr[i].name, arr[i].sequence, arr[i].length = b"r%d" % i, keep[i], len(keep[i])
        sarr = (binding.GaSeed * n)()
        offs = (C.c_size_t * (n + 1))(*range(n + 1))
        for i, (node, pos, rev) in enumerate(seeds):
            sarr[i].node_id, sarr[i].read_pos, sarr[i].reverse = int(node), int(pos), int(bool(rev))
        out = C.POINTER(binding.GaResults)()
        assert L.ga_align_batch(h, arr, n, sarr, offs, 35, 0, binding.GA_F_TRACE, C.byref(out)) == 0
        got = binding._unpack(out.contents)
        L.ga_results_free(out)
        L.ga_graph_destroy(h)
        return got

    # (a) the graph as the reference builds it: AddEdgeNodeId in file order
    h = L.ga_graph_create()
    nodes_into(h)
    for a, b in digraph_edges:
        assert L.ga_graph_add_edge(h, a, b) == 0
    built = run(h)
    # (b) the finished lists (what AlignmentGraph::inNeighbors / outNeighbors hold after (a)), handed over verbatim
    inn, outn = {}, {}
    for a, b in digraph_edges:
        if a not in inn.setdefault(b, []):
            inn[b].append(a)
        if b not in outn.setdefault(a, []):
            outn[a].append(b)
    h = L.ga_graph_create()
    nodes_into(h)
    for nid, _ in g.nodes:
        for d in (2 * nid, 2 * nid + 1):
            i = np.array(inn.get(d, []), dtype=np.int64)
            o = np.array(outn.get(d, []), dtype=np.int64)
            assert L.ga_graph_set_neighbors(h, d, i.ctypes.data_as(C.c_void_p), len(i), o.ctypes.data_as(C.c_void_p), len(o)) == 0
    mirrored = run(h)
    for a, b in zip(built, mirrored):
        assert a["status"] == b["status"] and a["score"] == b["score"] and a["mappings"] == b["mappings"] and (a["trace"] == b["trace"]).all()
    assert sum(1 for a in mirrored if a["status"] == 0 and not a["failed"]) >= 6
    # the replay grouped by target changes out-neighbour orders: not the same graph
    replayed = {}
    for b in sorted(inn):
        for a in inn[b]:
            replayed.setdefault(a, []).append(b)
    assert any(replayed[a] != outn[a] for a in outn)
    # and the oracle agrees with the mirrored graph
    import oracle_binding as ob
    og = ob.OracleGraph.__new__(ob.OracleGraph)
    OL = ob.lib()
    og.h = OL.gao_graph_new()
    for nid, seq in g.nodes:
        rc = "".join(comp[c] for c in reversed(seq))
        assert OL.gao_graph_add_node(og.h, 2 * nid, seq.encode(), 0) == 0 and OL.gao_graph_add_node(og.h, 2 * nid + 1, rc.encode(), 1) == 0
    for a, b in digraph_edges:
        assert OL.gao_graph_add_edge(og.h, a, b) == 0
    OL.gao_graph_finalize(og.h)
    for i, (r, sd) in enumerate(zip(reads, seeds)):
        pc.compare_read(mirrored[i], og.align(r, [sd], 35), "mirrored graph read %d" % i)


def test_kernel_resource_table_of_the_build():
    """build() keeps the compiler's per-kernel resource usage (build/kernel_resources.json) and refuses lanes = reads kernels with a
    private segment; here: the table exists for the shipped library and says what DESIGN.md says about the first-pass kernel"""
    import json
    import __graft_entry__ as entry
    if not os.path.exists(entry.RESOURCES_JSON) or os.path.getmtime(entry.RESOURCES_JSON) + 5 < os.path.getmtime(LIB):
        entry.build_product(force=True)
    table = json.load(open(entry.RESOURCES_JSON))
    lanes = {k: v for k, v in table.items() if k.startswith("ga_lanes_kernel")}
    assert len(lanes) == 3 and len([k for k in table if k.startswith("ga_extend_kernel")]) == 5
    for name, r in lanes.items():
        assert int(r["ScratchSize [bytes/lane]"]) == 0, (name, r)
    first = [v for k, v in lanes.items() if k.startswith("ga_lanes_kernel<10, 64>")][0]
    assert int(first["VGPRs Spill"]) == 0 and int(first["LDS Size [bytes/block]"]) <= 40960
