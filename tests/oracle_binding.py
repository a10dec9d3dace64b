"""ctypes binding over the CPU oracle (oracle/_build/libga_oracle.so) and, when present, the
reference parts (oracle/_ref/libga_refparts.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libga_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libga_refparts.so")

STATUS = {0: "OK", 1: "ASSERTION", 2: "UNSUPPORTED", 3: "BAD_SEED"}


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.gao_graph_new.restype = C.c_void_p
        L.gao_graph_free.argtypes = [C.c_void_p]
        L.gao_graph_add_node.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.gao_graph_add_edge.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.gao_graph_add_bigraph_node.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.gao_graph_add_bigraph_edge.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.gao_graph_set_overlap.argtypes = [C.c_void_p, C.c_int]
        L.gao_graph_finalize.argtypes = [C.c_void_p]
        L.gao_graph_nodes.argtypes = [C.c_void_p]
        L.gao_graph_nodes.restype = C.c_int64
        L.gao_graph_bp.argtypes = [C.c_void_p]
        L.gao_graph_bp.restype = C.c_int64
        L.gao_align.restype = C.c_void_p
        L.gao_align.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.gao_result_free.argtypes = [C.c_void_p]
        L.gao_result_summary.argtypes = [C.c_void_p, C.c_void_p]
        L.gao_result_message.argtypes = [C.c_void_p]
        L.gao_result_message.restype = C.c_char_p
        L.gao_result_mappings.argtypes = [C.c_void_p, C.c_void_p]
        L.gao_result_mapping_seq.argtypes = [C.c_void_p, C.c_int]
        L.gao_result_mapping_seq.restype = C.c_char_p
        L.gao_result_trace.argtypes = [C.c_void_p, C.c_void_p]
        L.gao_result_raw_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gao_slice_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gao_slice_nodes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gao_slice_minindex.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gao_slice_columns.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.gao_slice_sparse_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.gao_merge_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.gao_column_value.argtypes = [C.c_void_p, C.c_int]
        L.gao_set_values.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.gao_step_column.argtypes = [C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.gao_hmm_chain.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gao_char_match.argtypes = [C.c_int, C.c_int]
        L.gao_reverse_complement.argtypes = [C.c_char_p, C.c_char_p]
        L.gao_bench.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def reflib():
    """reference parts; None when the prebuilt library is absent (it cannot be rebuilt off-container)"""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        R = C.CDLL(REF_SO)
        R.ref_merge_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_column_value.argtypes = [C.c_void_p, C.c_int]
        if hasattr(R, "ref_set_values"):
            R.ref_set_values.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        R.ref_hmm_chain.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_frozen_order.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
        R.ref_freeze_thaw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _ref = R
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleGraph:
    """bidirected graph -> oracle's digraph through the loaders' conversion (BigraphToDigraph.cpp)"""

    def __init__(self, nodes, edges, overlap=0):
        """nodes: iterable of (id, sequence); edges: iterable of (from, from_start, to, to_end)"""
        L = lib()
        self.h = L.gao_graph_new()
        L.gao_graph_set_overlap(self.h, overlap)
        for nid, seq in nodes:
            st = L.gao_graph_add_bigraph_node(self.h, int(nid), seq.encode())
            if st:
                raise ValueError("add node failed: %s" % STATUS.get(st, st))
        for f, fs, t, te in edges:
            st = L.gao_graph_add_bigraph_edge(self.h, int(f), int(fs), int(t), int(te))
            if st:
                raise ValueError("add edge failed: %s" % STATUS.get(st, st))
        L.gao_graph_finalize(self.h)

    @classmethod
    def from_gfa_segments(cls, segments, links, overlap):
        """the GFA loader's conversion (BigraphToDigraph.cpp:58-104, 137-189): forward node = the segment minus its last
        `overlap` bases, backward node = the reverse complement of the WHOLE segment minus ITS last `overlap` bases;
        links: (from, from_is_minus, to, to_is_minus)"""
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        L = lib()
        self = cls.__new__(cls)
        self.h = L.gao_graph_new()
        L.gao_graph_set_overlap(self.h, overlap)
        for nid, seq in segments:
            rc = "".join(comp[c] for c in reversed(seq))
            for did, s, rev in ((2 * nid, seq[:len(seq) - overlap], 0), (2 * nid + 1, rc[:len(seq) - overlap], 1)):
                st = L.gao_graph_add_node(self.h, int(did), s.encode(), rev)
                if st:
                    raise ValueError("add node failed: %s" % STATUS.get(st, st))
        for f, fs, t, te in links:
            st = L.gao_graph_add_bigraph_edge(self.h, int(f), int(fs), int(t), int(te))
            if st:
                raise ValueError("add edge failed: %s" % STATUS.get(st, st))
        L.gao_graph_finalize(self.h)
        return self

    def __del__(self):
        try:
            lib().gao_graph_free(self.h)
        except Exception:
            pass

    def align(self, seq, seeds, bw, ramp=0, record=False, name="read"):
        L = lib()
        sd = np.array([[s[0], s[1], int(s[2])] for s in seeds], dtype=np.int64).reshape(-1)
        r = L.gao_align(self.h, name.encode(), seq.encode(), bw, ramp, _p(sd), len(seeds), int(record))
        try:
            return _unpack_result(L, r)
        finally:
            L.gao_result_free(r)

    def bench(self, reads, seeds, bw, ramp, threads):
        L = lib()
        blob = "".join(reads).encode()
        offs = np.zeros(len(reads) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(r) for r in reads])
        sd = np.array([[s[0], s[1], int(s[2])] for s in seeds], dtype=np.int64).reshape(-1)
        out = np.zeros(5, dtype=np.float64)
        L.gao_bench(self.h, blob, _p(offs), _p(sd), len(reads), bw, ramp, threads, _p(out))
        return dict(seconds=out[0], aligned_bp=out[1], reads_ok=int(out[2]), columns=out[3], score_sum=out[4])


def _unpack_result(L, r):
    s = np.zeros(18, dtype=np.int64)
    L.gao_result_summary(r, _p(s))
    res = dict(status=int(s[0]), failed=bool(s[1]), score=int(s[2]), alignment_start=int(s[3]), alignment_end=int(s[4]),
               query_position=int(s[5]), fw_score=int(s[10]), bw_score=int(s[11]), columns=int(s[12]), slices=int(s[13]), sparse_slices=int(s[15]), override_windows=int(s[16]), override_traces=int(s[17]),
               message=L.gao_result_message(r).decode())
    nm, nt, nf, nb, ns = int(s[6]), int(s[7]), int(s[8]), int(s[9]), int(s[14])
    m = np.zeros((nm, 6), dtype=np.int64)
    if nm:
        L.gao_result_mappings(r, _p(m))
    res["mappings"] = [tuple(int(x) for x in m[i]) + (L.gao_result_mapping_seq(r, i).decode(),) for i in range(nm)]
    t = np.zeros((nt, 7), dtype=np.int64)
    if nt:
        L.gao_result_trace(r, _p(t))
    res["trace"] = t
    fw = np.zeros((nf, 2), dtype=np.int64)
    bw = np.zeros((nb, 2), dtype=np.int64)
    if nf:
        L.gao_result_raw_trace(r, 0, _p(fw))
    if nb:
        L.gao_result_raw_trace(r, 1, _p(bw))
    res["fw_trace"], res["bw_trace"] = fw, bw
    slices = []
    for i in range(ns):
        info = np.zeros(7, dtype=np.int64)
        L.gao_slice_info(r, i, _p(info))
        nn, nc, nmi = int(info[3]), int(info[4]), int(info[6])
        nodes = np.zeros(nn, dtype=np.int64)
        mi = np.zeros(nmi, dtype=np.int64)
        vp = np.zeros(nc, dtype=np.uint64)
        vn = np.zeros(nc, dtype=np.uint64)
        before = np.zeros(nc, dtype=np.int32)
        end = np.zeros(nc, dtype=np.int32)
        ex = np.zeros(nc, dtype=np.uint8)
        L.gao_slice_nodes(r, i, _p(nodes))
        L.gao_slice_minindex(r, i, _p(mi))
        L.gao_slice_columns(r, i, _p(vp), _p(vn), _p(before), _p(end), _p(ex))
        written = np.zeros(nc, dtype=np.uint64)
        eex = np.zeros(nc, dtype=np.uint8)
        sparse = L.gao_slice_sparse_info(r, i, _p(written), _p(eex))
        slices.append(dict(direction=int(info[0]), j=int(info[1]), bandwidth=int(info[2]), nodes=nodes, min_score=int(info[5]),
                           min_index=mi, vp=vp, vn=vn, before=before, end=end, before_exists=ex, sparse=bool(sparse), written=written, end_exists=eex))
    res["slice_records"] = slices
    return res
