"""ctypes binding over the C ABI declared in include/graphaligner_amd.h.

The product library is graphaligner_amd/libgraphaligner_amd.so (HIP, gfx950).  It has no CPU
path: creating a Graph on a machine without a usable GPU raises.  (tests/ may point this
binding at tests/_build/libga_emul.so, a host emulation of the device program used to check
the device logic in the GPU-less build container.)"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_SO = os.path.join(HERE, "libgraphaligner_amd.so")

STATUS = {0: "OK", 1: "ASSERTION", 2: "UNSUPPORTED_BAND", 3: "BAD_SEED", 10: "CAPACITY", 20: "UNSUPPORTED_CYCLE", 21: "UNSUPPORTED_RAMP",
          100: "E_INVALID", 101: "E_NO_DEVICE", 102: "E_DEVICE", 103: "E_NOT_FINALIZED"}
GA_F_TRACE = 1
GA_S_OK = 0
GA_S_ASSERTION = 1


class GaRead(C.Structure):
    _fields_ = [("name", C.c_char_p), ("sequence", C.c_char_p), ("length", C.c_size_t)]


class GaSeed(C.Structure):
    _fields_ = [("node_id", C.c_int64), ("read_pos", C.c_uint64), ("reverse", C.c_int32), ("reserved", C.c_int32)]


class GaMapping(C.Structure):
    _fields_ = [("node_id", C.c_int64), ("is_reverse", C.c_int32), ("rank", C.c_int32), ("offset", C.c_int64),
                ("from_length", C.c_int64), ("to_length", C.c_int64), ("edit_seq_off", C.c_uint64)]


class GaTraceItem(C.Structure):
    _fields_ = [("node_id", C.c_int32), ("reverse", C.c_int32), ("offset", C.c_uint64), ("read_pos", C.c_uint64),
                ("type", C.c_int32), ("graph_char", C.c_char), ("read_char", C.c_char), ("pad", C.c_char * 2)]


class GaReadResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("failed", C.c_int32), ("score", C.c_int32), ("reserved", C.c_int32),
                ("alignment_start", C.c_uint64), ("alignment_end", C.c_uint64), ("query_position", C.c_uint64),
                ("first_mapping", C.c_uint64), ("n_mappings", C.c_uint64), ("first_trace", C.c_uint64), ("n_trace", C.c_uint64),
                ("column_updates", C.c_uint64)]


class GaResults(C.Structure):
    _fields_ = [("n_reads", C.c_size_t), ("reads", C.POINTER(GaReadResult)), ("n_mappings", C.c_size_t), ("mappings", C.POINTER(GaMapping)),
                ("n_edit_bytes", C.c_size_t), ("edit_bytes", C.POINTER(C.c_char)), ("n_trace", C.c_size_t), ("trace", C.POINTER(GaTraceItem))]


class GaBatchStats(C.Structure):
    _fields_ = [("n_jobs", C.c_uint64), ("column_updates", C.c_uint64), ("slices", C.c_uint64), ("jobs_retried", C.c_uint64),
                ("kernel_ms", C.c_double), ("prep_kernel_ms", C.c_double), ("slots", C.c_uint32), ("waves_per_cu", C.c_uint32),
                ("scratch_bytes", C.c_uint64), ("stamps", C.c_uint64 * 8), ("main_kernel_ms", C.c_double), ("main_variant", C.c_int32),
                ("reserved", C.c_int32)]


class GaNamedSeed(C.Structure):
    _fields_ = [("read_name", C.c_char_p), ("seed", GaSeed)]


EXPORTS = ["ga_graph_create", "ga_graph_destroy", "ga_graph_add_node", "ga_graph_add_edge", "ga_graph_add_bigraph_node",
           "ga_graph_add_bigraph_edge", "ga_graph_finalize", "ga_graph_load_gfa", "ga_graph_upload", "ga_graph_node_count", "ga_graph_bp",
           "ga_align_batch", "ga_results_free", "ga_batch_prepare", "ga_batch_run", "ga_batch_collect", "ga_batch_free", "ga_batch_stats",
           "ga_graph_set_neighbors", "ga_graph_load_gfa_split", "ga_graph_split_lookup", "ga_results_unsplit", "ga_graph_load_vg", "ga_gam_decode_seeds", "ga_results_encode_gam", "ga_bytes_free", "ga_status_string", "ga_version"]

_libs = {}


def load(path=None):
    path = path or PRODUCT_SO
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    L = C.CDLL(path)
    L.ga_graph_create.restype = C.c_void_p
    L.ga_graph_destroy.argtypes = [C.c_void_p]
    L.ga_graph_add_node.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t, C.c_int]
    L.ga_graph_add_edge.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.ga_graph_set_neighbors.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ga_graph_add_bigraph_node.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
    L.ga_graph_add_bigraph_edge.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int]
    L.ga_graph_finalize.argtypes = [C.c_void_p, C.c_int]
    L.ga_graph_load_gfa.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.ga_graph_load_gfa_split.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32]
    L.ga_graph_split_lookup.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ga_results_unsplit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ga_graph_upload.argtypes = [C.c_void_p, C.c_int]
    L.ga_graph_node_count.argtypes = [C.c_void_p]
    L.ga_graph_node_count.restype = C.c_int64
    L.ga_graph_bp.argtypes = [C.c_void_p]
    L.ga_graph_bp.restype = C.c_int64
    L.ga_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
    L.ga_results_free.argtypes = [C.c_void_p]
    L.ga_batch_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
    L.ga_batch_run.argtypes = [C.c_void_p]
    L.ga_batch_collect.argtypes = [C.c_void_p, C.c_void_p]
    L.ga_batch_free.argtypes = [C.c_void_p]
    L.ga_batch_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.ga_graph_load_vg.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.ga_gam_decode_seeds.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ga_results_encode_gam.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ga_bytes_free.argtypes = [C.c_void_p]
    L.ga_status_string.argtypes = [C.c_int]
    L.ga_status_string.restype = C.c_char_p
    L.ga_version.restype = C.c_char_p
    _libs[path] = L
    return L


def _check(L, s, what):
    if s != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, L.ga_status_string(s).decode(), s))


class Graph:
    """AlignmentGraph built the way the reference's loaders build it, then copied to HBM"""

    def __init__(self, nodes=None, edges=None, overlap=0, gfa=None, vg=None, device=0, lib_path=None, split=0):
        self.L = load(lib_path)
        self.h = self.L.ga_graph_create()
        if vg is not None:
            _check(self.L, self.L.ga_graph_load_vg(self.h, vg, len(vg)), "ga_graph_load_vg")
        elif gfa is not None:
            data = gfa.encode() if isinstance(gfa, str) else gfa
            if split:
                _check(self.L, self.L.ga_graph_load_gfa_split(self.h, data, len(data), int(split)), "ga_graph_load_gfa_split")
            else:
                _check(self.L, self.L.ga_graph_load_gfa(self.h, data, len(data)), "ga_graph_load_gfa")
        else:
            for nid, seq in nodes:
                b = seq.encode() if isinstance(seq, str) else seq
                _check(self.L, self.L.ga_graph_add_bigraph_node(self.h, int(nid), b, len(b)), "ga_graph_add_bigraph_node")
            for f, fs, t, te in edges:
                _check(self.L, self.L.ga_graph_add_bigraph_edge(self.h, int(f), int(fs), int(t), int(te)), "ga_graph_add_bigraph_edge")
            _check(self.L, self.L.ga_graph_finalize(self.h, overlap), "ga_graph_finalize")
        _check(self.L, self.L.ga_graph_upload(self.h, device), "ga_graph_upload")

    def __del__(self):
        try:
            self.L.ga_graph_destroy(self.h)
        except Exception:
            pass

    @property
    def node_count(self):
        return self.L.ga_graph_node_count(self.h)

    @property
    def bp(self):
        return self.L.ga_graph_bp(self.h)

    def prepare(self, reads, seeds, bw, ramp=0, flags=0, names=None):
        """reads: list of str (or a ReadSet, then seeds is ignored); seeds: list (one entry per read) of lists of (node, pos, reverse) or a single tuple"""
        return Batch(self, reads, seeds, bw, ramp, flags, names)

    def align(self, reads, seeds, bw, ramp=0, flags=0):
        b = self.prepare(reads, seeds, bw, ramp, flags)
        b.run()
        return b.collect()


class ReadSet:
    """reads and seeds in the form the C ABI takes them (arrays of ga_read_t / ga_seed_t + CSR offsets): what an application that
    calls the library from C or C++ already holds.  Built once, it can be handed to Graph.prepare any number of times."""

    def __init__(self, reads, seeds, names=None):
        n = len(reads)
        self._keep = [r.encode() if isinstance(r, str) else r for r in reads]
        arr = (GaRead * max(n, 1))()
        self._names = [(names[i].encode() if names is not None else b"read%d" % i) for i in range(n)]
        for i, r in enumerate(self._keep):
            arr[i].name = self._names[i]
            arr[i].sequence = r
            arr[i].length = len(r)
        flat = []
        offs = np.zeros(n + 1, dtype=np.uint64)
        for i, s in enumerate(seeds):
            lst = [s] if (len(s) == 3 and not isinstance(s[0], (tuple, list))) else list(s)
            flat.extend(lst)
            offs[i + 1] = len(flat)
        sarr = (GaSeed * max(len(flat), 1))()
        for i, (node, pos, rev) in enumerate(flat):
            sarr[i].node_id = int(node)
            sarr[i].read_pos = int(pos)
            sarr[i].reverse = int(bool(rev))
        self.arr, self.sarr, self.offs = arr, sarr, offs
        self.n_reads = n
        self.total_bp = sum(len(r) for r in self._keep)


class Batch:
    def __init__(self, graph, reads, seeds, bw, ramp, flags, names=None):
        self.g = graph
        L = self.L = graph.L
        rs = reads if isinstance(reads, ReadSet) else ReadSet(reads, seeds, names)
        self._rs = rs                      # (the arrays must outlive the batch: GAM encoding reads the names)
        self._arr, self._sarr, self._offs = rs.arr, rs.sarr, rs.offs
        self.n_reads = rs.n_reads
        self.total_bp = rs.total_bp
        h = C.c_void_p()
        _check(L, L.ga_batch_prepare(graph.h, rs.arr, rs.n_reads, rs.sarr, rs.offs.ctypes.data_as(C.c_void_p), bw, ramp, flags, C.byref(h)), "ga_batch_prepare")
        self.h = h

    def run(self):
        _check(self.L, self.L.ga_batch_run(self.h), "ga_batch_run")

    def stats(self):
        st = GaBatchStats()
        _check(self.L, self.L.ga_batch_stats(self.h, C.byref(st)), "ga_batch_stats")
        d = {k: getattr(st, k) for k, _ in st._fields_}
        d["stamps"] = list(st.stamps)
        return d

    def collect_gam(self, halve_node_ids=True):
        """align results as the bytes of a GAM file (what the reference's driver writes, Aligner.cpp:301-314)"""
        out = C.POINTER(GaResults)()
        _check(self.L, self.L.ga_batch_collect(self.h, C.byref(out)), "ga_batch_collect")
        try:
            buf, n = C.c_void_p(), C.c_size_t()
            _check(self.L, self.L.ga_results_encode_gam(out, self._arr, int(halve_node_ids), C.byref(buf), C.byref(n)), "ga_results_encode_gam")
            data = C.string_at(buf, n.value)
            self.L.ga_bytes_free(buf)
            return data
        finally:
            self.L.ga_results_free(out)

    def collect(self, summary=False, unsplit=False):
        """summary=True: per-read numpy record array only (status, failed, score, n_mappings, ...), no Python lists;
        unsplit=True: results of a graph loaded with split=... on the nodes of the GFA file (ga_results_unsplit)"""
        out = C.POINTER(GaResults)()
        _check(self.L, self.L.ga_batch_collect(self.h, C.byref(out)), "ga_batch_collect")
        if unsplit:
            merged = C.POINTER(GaResults)()
            try:
                _check(self.L, self.L.ga_results_unsplit(self.g.h, out, C.byref(merged)), "ga_results_unsplit")
            finally:
                self.L.ga_results_free(out)
            out = merged
        try:
            if summary:
                R = out.contents
                dt = np.dtype([("status", "<i4"), ("failed", "<i4"), ("score", "<i4"), ("reserved", "<i4"), ("alignment_start", "<u8"),
                               ("alignment_end", "<u8"), ("query_position", "<u8"), ("first_mapping", "<u8"), ("n_mappings", "<u8"),
                               ("first_trace", "<u8"), ("n_trace", "<u8"), ("column_updates", "<u8")])
                assert dt.itemsize == C.sizeof(GaReadResult)
                buf = C.string_at(R.reads, R.n_reads * dt.itemsize) if R.n_reads else b""
                return np.frombuffer(buf, dtype=dt).copy()
            return _unpack(out.contents)
        finally:
            self.L.ga_results_free(out)

    def close(self):
        """free the batch now (its device buffers and its copy of the reads go back to the graph's pools)"""
        if self.h:
            self.L.ga_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def status_string(status, lib_path=None):
    return load(lib_path).ga_status_string(int(status)).decode()


def decode_seed_gam(data, lib_path=None):
    """[(read name, (node, pos, reverse)), ...] from the bytes of a seed GAM file (Aligner.cpp:253-271)"""
    L = load(lib_path)
    arr, n = C.POINTER(GaNamedSeed)(), C.c_size_t()
    _check(L, L.ga_gam_decode_seeds(data, len(data), C.byref(arr), C.byref(n)), "ga_gam_decode_seeds")
    out = [(arr[i].read_name.decode(), (arr[i].seed.node_id, arr[i].seed.read_pos, bool(arr[i].seed.reverse))) for i in range(n.value)]
    L.ga_bytes_free(arr)
    return out


def _unpack(R):
    reads = []
    edit = C.string_at(R.edit_bytes, R.n_edit_bytes) if R.n_edit_bytes else b""
    for i in range(R.n_reads):
        r = R.reads[i]
        maps = []
        for k in range(r.first_mapping, r.first_mapping + r.n_mappings):
            m = R.mappings[k]
            maps.append((m.node_id, m.is_reverse, m.offset, m.rank, m.from_length, m.to_length,
                         edit[m.edit_seq_off:m.edit_seq_off + m.to_length].decode()))
        tr = np.zeros((r.n_trace, 7), dtype=np.int64)
        for k in range(r.n_trace):
            t = R.trace[r.first_trace + k]
            tr[k] = (t.node_id, t.offset, t.reverse, t.read_pos, t.type, ord(t.graph_char), ord(t.read_char))
        reads.append(dict(status=r.status, failed=bool(r.failed), score=r.score, alignment_start=r.alignment_start, alignment_end=r.alignment_end,
                          query_position=r.query_position, mappings=maps, trace=tr, columns=r.column_updates, kernel_pass=r.reserved))
    return reads
