"""Deterministic synthetic graphs and reads (no real genomes are available offline).

Scales and parameters follow SURVEY.md section 8(d): seeded random genomes, variation bubbles
at 1000GP-like density, node length capped (the reference never splits nodes and abandons
the bit-vector method once a band holds >= 200000 bp, GraphAlignerCommon.h:10), reads drawn
from haplotype walks with the error model of the reference's SimulateReads.cpp:12-41 under a
fixed seed (the reference's own simulator is time-seeded, SimulateReads.cpp:131, so it cannot
define fixtures).

Graphs are *bidirected* (vg style): nodes (id, sequence), edges (from, from_start, to,
to_end).  All edges produced here are end->start (from_start = to_end = False).
"""
import numpy as np

_ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[a] = b


def revcomp_bytes(a):
    return _COMP[a[::-1]]


def random_genome(n, seed):
    rng = np.random.default_rng(seed)
    return _ALPHA[rng.integers(0, 4, size=n, dtype=np.uint8)]


class SynthGraph:
    """a variation graph over one backbone genome; node ids are 1.. in topological order"""

    def __init__(self, genome, node_len=64, snp_every=0, indel_every=0, sv_every=0, seed=0, max_indel=20, first_id=1):
        self.genome = genome
        self.node_len = node_len
        rng = np.random.default_rng(seed)
        n = len(genome)
        # ---- choose variant sites (sorted, at least 2 backbone bases apart) -----------------
        variants = []   # (pos, kind, ref_len, alt bytes)
        pos = 0
        rates = [(snp_every, "snp"), (indel_every, "indel"), (sv_every, "sv")]
        total_rate = sum(1.0 / r for r, _ in rates if r)
        if total_rate > 0:
            kinds = [k for r, k in rates if r]
            probs = np.array([1.0 / r for r, _ in rates if r]) / total_rate
            pos = 1 + int(rng.geometric(total_rate))
            while pos < n - 600:
                kind = kinds[int(rng.choice(len(kinds), p=probs))]
                if kind == "snp":
                    ref = genome[pos]
                    alt = _ALPHA[(int(np.searchsorted(_ALPHA, ref)) + 1 + int(rng.integers(0, 3))) % 4]
                    variants.append((pos, "snp", 1, np.array([alt], dtype=np.uint8)))
                    pos += 1
                else:
                    ln = int(rng.integers(1, max_indel + 1)) if kind == "indel" else int(rng.integers(50, 501))
                    if rng.random() < 0.5:
                        variants.append((pos, "del", ln, np.zeros(0, dtype=np.uint8)))
                        pos += ln
                    else:
                        variants.append((pos, "ins", 0, _ALPHA[rng.integers(0, 4, size=ln, dtype=np.uint8)]))
                pos += 2 + int(rng.geometric(total_rate))
        self.variants = variants
        # ---- build nodes / edges -------------------------------------------------------------
        self.nodes = []          # (id, str)
        self.edges = []          # (from, False, to, False)
        self.node_at = np.zeros(n, dtype=np.int32)     # backbone position -> node id (ref allele path)
        self.node_off = np.zeros(n, dtype=np.int32)    # offset inside that node
        self._next_id = first_id
        self.var_nodes = []      # per variant: (node ids of the reference allele, node ids of the alternative allele)
        tails = []

        def add_chain(seq_bytes, tails_in, backbone_start=None):
            """chop into <= node_len nodes, link tails_in -> first, return id of last"""
            last = None
            for k in range(0, len(seq_bytes), node_len):
                piece = seq_bytes[k:k + node_len]
                nid = self._next_id
                self._next_id += 1
                self.nodes.append((nid, piece.tobytes().decode()))
                if last is None:
                    for t in tails_in:
                        self.edges.append((t, False, nid, False))
                else:
                    self.edges.append((last, False, nid, False))
                if backbone_start is not None:
                    a = backbone_start + k
                    self.node_at[a:a + len(piece)] = nid
                    self.node_off[a:a + len(piece)] = np.arange(len(piece), dtype=np.int32)
                last = nid
            return last

        cur = 0
        for (p, kind, ref_len, alt) in variants:
            if p > cur:
                tails = [add_chain(genome[cur:p], tails, cur)]
            id0 = self._next_id
            if kind == "snp":
                r = add_chain(genome[p:p + 1], tails, p)
                a = add_chain(alt, tails)
                self.var_nodes.append(([r], [a]))
                tails = [r, a]
                cur = p + 1
            elif kind == "del":
                d = add_chain(genome[p:p + ref_len], tails, p)
                self.var_nodes.append((list(range(id0, self._next_id)), []))
                tails = tails + [d]
                cur = p + ref_len
            else:
                i = add_chain(alt, tails)
                self.var_nodes.append(([], list(range(id0, self._next_id))))
                tails = tails + [i]
                cur = p
        if cur < n:
            add_chain(genome[cur:n], tails, cur)
        self.var_pos = np.array([v[0] for v in variants], dtype=np.int64)

    # ---- export -------------------------------------------------------------------------------
    def gfa(self, overlap=0):
        lines = ["H\tVN:Z:1.0"]
        for nid, seq in self.nodes:
            lines.append("S\t%d\t%s" % (nid, seq))
        for f, fs, t, te in self.edges:
            lines.append("L\t%d\t%s\t%d\t%s\t%dM" % (f, "-" if fs else "+", t, "-" if te else "+", overlap))
        return "\n".join(lines) + "\n"

    def vg_bytes(self, chunk_nodes=1000):
        """the graph as the reference's vg loader reads it (CommonUtils / stream.hpp:24-118): gzip-framed groups of vg.Graph messages,
        chunks of at most `chunk_nodes` nodes, each chunk carrying the edges that leave its nodes"""
        return vg_bytes(self.nodes, self.edges, chunk_nodes)

    # ---- haplotypes & reads -----------------------------------------------------------------------
    def _backbone_nodes(self, lo, hi, path):
        if path is not None and hi > lo:
            ids = self.node_at[lo:hi]
            keep = np.concatenate([[True], ids[1:] != ids[:-1]])
            for x in ids[keep]:
                if not path or path[-1] != int(x):
                    path.append(int(x))

    def haplotype_window(self, start, length, rng, path=None):
        """bytes of one random haplotype starting at backbone position `start` (which must be a
        backbone/ref position), at least `length` long when the genome allows"""
        out = []
        have = 0
        cur = start
        lo = int(np.searchsorted(self.var_pos, start, side="left"))
        k = lo
        n = len(self.genome)
        while have < length and cur < n:
            nxt = self.variants[k][0] if k < len(self.variants) else n
            if nxt > cur:
                take = min(nxt - cur, length - have)
                out.append(self.genome[cur:cur + take])
                self._backbone_nodes(cur, cur + take, path)
                have += take
                cur += take
                if have >= length:
                    break
            if k >= len(self.variants):
                continue
            p, kind, ref_len, alt = self.variants[k]
            k += 1
            use_alt = rng.random() < 0.5
            if path is not None:
                path.extend(self.var_nodes[k - 1][1 if use_alt else 0])
            if kind == "snp":
                out.append(alt if use_alt else self.genome[p:p + 1])
                have += 1
                cur = p + 1
            elif kind == "del":
                if not use_alt:
                    out.append(self.genome[p:p + ref_len])
                    have += ref_len
                cur = p + ref_len
            else:
                if use_alt:
                    out.append(alt)
                    have += len(alt)
                cur = p
        hap = np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)
        return hap[:length]

    def _backbone_start(self, p):
        """move p right until it is a plain backbone position (not inside a variant site)"""
        while True:
            k = int(np.searchsorted(self.var_pos, p, side="right")) - 1
            if k >= 0:
                vp, kind, ref_len, _ = self.variants[k]
                span = 1 if kind == "snp" else ref_len
                if vp <= p < vp + span:
                    p = vp + span
                    continue
            return p


def add_errors(seq, sub, ins, dele, rng):
    """SimulateReads.cpp:12-41: per base: drop w.p. dele; else substitute w.p. sub by a uniform
    base (may equal the original); then w.p. ins/10 insert uniform(0..19) uniform bases."""
    n = len(seq)
    keep = rng.random(n) >= dele
    subm = rng.random(n) < sub
    s = seq.copy()
    ns = int(subm.sum())
    if ns:
        s[subm] = _ALPHA[rng.integers(0, 4, size=ns, dtype=np.uint8)]
    insm = rng.random(n) < ins / 10.0
    if not insm.any():
        return s[keep]
    pieces = []
    last = 0
    for i in np.nonzero(insm)[0]:
        seg_keep = keep[last:i + 1]
        pieces.append(s[last:i + 1][seg_keep])
        ln = int(rng.integers(0, 20))
        if ln:
            pieces.append(_ALPHA[rng.integers(0, 4, size=ln, dtype=np.uint8)])
        last = i + 1
    pieces.append(s[last:][keep[last:]])
    return np.concatenate(pieces)


def simulate_reads(graph, n_reads, length, sub=0.04, ins=0.04, dele=0.04, seed=1, both_strands=True, mid_seed=False, min_part=2, truth=None):
    """returns (reads, seeds): reads are str, seeds are (bigraph node id, read position, reverse).  `truth`: a list that receives,
    per read, the node ids of the walk it was drawn from (what SimulateReads.cpp writes as its truth GAM, :86-112).
    The seed names the node holding the read's first base (position 0) or, with mid_seed, the
    node holding the base at the middle of the read (exercises the backward extension)."""
    rng = np.random.default_rng(seed)
    g = graph
    n = len(g.genome)
    reads, seeds = [], []
    attempts = 0
    while len(reads) < n_reads:
        attempts += 1
        if attempts > 50 * n_reads + 1000:
            raise RuntimeError("simulate_reads: cannot place %d reads of %d bp on this graph" % (n_reads, length))
        start = g._backbone_start(int(rng.integers(0, max(1, n - length - 1200))))
        hpath = [] if truth is not None else None
        hap = g.haplotype_window(start, length, rng, hpath)
        if len(hap) < length:
            continue
        reverse = both_strands and rng.random() < 0.5
        # the haplotype ends somewhere; for seeding we need a backbone anchor at the read's
        # first base.  Forward reads: `start`.  Reverse reads: the last base of the window must
        # be a backbone base, so rebuild the window to end at a backbone position.
        if not reverse:
            if mid_seed:
                half = length // 2
                # first half = haplotype from start up to a backbone anchor, second half from it
                anchor = g._backbone_start(start + half)
                p1 = [] if truth is not None else None
                first = _hap_until(g, start, anchor, rng, p1)
                second = g.haplotype_window(anchor, length - half, rng, p1)
                a = add_errors(first, sub, ins, dele, rng)
                b = add_errors(second, sub, ins, dele, rng)
                if len(a) < min_part or len(b) < min_part:
                    continue
                reads.append(np.concatenate([a, b]).tobytes().decode())
                seeds.append((int(g.node_at[anchor]), len(a), False))
                if truth is not None:
                    truth.append(p1)
            else:
                r = add_errors(hap, sub, ins, dele, rng)
                if len(r) < min_part:
                    continue
                reads.append(r.tobytes().decode())
                seeds.append((int(g.node_at[start]), 0, False))
                if truth is not None:
                    truth.append(hpath)
        else:
            # walk forward from `start`, stop at a backbone anchor near start+length
            anchor = g._backbone_start(start + length)
            if anchor >= n - 1:
                continue
            bpath = [] if truth is not None else None
            body = _hap_until(g, start, anchor + 1, rng, bpath)     # includes the anchor base
            if len(body) < min_part:
                continue
            rc = revcomp_bytes(body)
            if mid_seed:
                mid_anchor = g._backbone_start(start + length // 2)
                p2 = [] if truth is not None else None
                left = _hap_until(g, start, mid_anchor + 1, rng, p2)
                right = _hap_until(g, mid_anchor + 1, anchor + 1, rng, p2)
                a = add_errors(revcomp_bytes(right), sub, ins, dele, rng)
                b = add_errors(revcomp_bytes(left), sub, ins, dele, rng)
                if len(a) < min_part or len(b) < min_part:
                    continue
                reads.append(np.concatenate([a, b]).tobytes().decode())
                seeds.append((int(g.node_at[mid_anchor]), len(a), True))
                if truth is not None:
                    truth.append(p2[::-1])
            else:
                r = add_errors(rc, sub, ins, dele, rng)
                if len(r) < min_part:
                    continue
                reads.append(r.tobytes().decode())
                seeds.append((int(g.node_at[anchor]), 0, True))
                if truth is not None:
                    truth.append(bpath[::-1])
    return reads, seeds


def _hap_until(g, start, stop, rng, path=None):
    """random haplotype covering backbone interval [start, stop) (both backbone positions)"""
    out = []
    cur = start
    k = int(np.searchsorted(g.var_pos, start, side="left"))
    while cur < stop:
        nxt = g.variants[k][0] if k < len(g.variants) else stop
        nxt = min(nxt, stop)
        if nxt > cur:
            out.append(g.genome[cur:nxt])
            g._backbone_nodes(cur, nxt, path)
            cur = nxt
            if cur >= stop:
                break
        if k >= len(g.variants):
            break
        p, kind, ref_len, alt = g.variants[k]
        if p >= stop:
            break
        k += 1
        use_alt = rng.random() < 0.5
        if path is not None:
            path.extend(g.var_nodes[k - 1][1 if use_alt else 0])
        if kind == "snp":
            out.append(alt if use_alt else g.genome[p:p + 1])
            cur = p + 1
        elif kind == "del":
            if not use_alt:
                out.append(g.genome[p:p + ref_len])
            cur = p + ref_len
        else:
            if use_alt:
                out.append(alt)
            cur = p
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)


def linear_graph(n, node_len=64, seed=42):
    """E. coli-like single chain of node_len-bp nodes"""
    return SynthGraph(random_genome(n, seed), node_len=node_len)


def bubble_graph(n, node_len=64, seed=44, snp_every=100, indel_every=1000, sv_every=50000):
    """yeast-like pangenome: SNP / short-indel / SV bubbles"""
    return SynthGraph(random_genome(n, seed), node_len=node_len, snp_every=snp_every, indel_every=indel_every, sv_every=sv_every, seed=seed + 1)


# ---- graphs with cycles (tandem repeats, self loops) -----------------------------------------------
def cyclic_graph(n, node_len=16, seed=7, back_edges=6, self_loops=2, max_span=6, snp_every=60):
    """a variation graph plus `back_edges` edges from a node to one of the `max_span` nodes before
    it (a tandem repeat unit) and `self_loops` nodes that follow themselves (a homopolymer-like
    repeat): the band subgraph of a slice crossing them has strongly connected components"""
    rng = np.random.default_rng(seed + 1000)
    g = SynthGraph(random_genome(n, seed), node_len=node_len, snp_every=snp_every, seed=seed)
    ids = [i for i, _ in g.nodes]
    have = set((f, t) for f, _, t, _ in g.edges)
    for _ in range(back_edges):
        a = int(rng.integers(max_span + 1, len(ids) - 2))
        b = a - int(rng.integers(1, max_span + 1))
        if (ids[a], ids[b]) not in have:
            have.add((ids[a], ids[b]))
            g.edges.append((ids[a], False, ids[b], False))
    for _ in range(self_loops):
        a = int(rng.integers(2, len(ids) - 2))
        if (ids[a], ids[a]) not in have:
            have.add((ids[a], ids[a]))
            g.edges.append((ids[a], False, ids[a], False))
    return g


def walk_reads(graph, n_reads, length, sub=0.03, ins=0.03, dele=0.03, seed=1, both_strands=True, mid_seed=False, first_nodes=None):
    """reads spelled by random walks along the forward edges (so repeats are traversed a random
    number of times), with SimulateReads-style errors; seeds as in simulate_reads"""
    rng = np.random.default_rng(seed)
    seq = dict(graph.nodes)
    out = {}
    for f, _, t, _ in graph.edges:
        out.setdefault(f, []).append(t)
    ids = [i for i, _ in graph.nodes]
    limit = first_nodes if first_nodes is not None else max(1, len(ids) // 2)
    reads, seeds = [], []
    attempts = 0
    while len(reads) < n_reads:
        attempts += 1
        if attempts > 50 * n_reads + 1000:
            raise RuntimeError("walk_reads: cannot place %d reads of %d bp" % (n_reads, length))
        cur = ids[int(rng.integers(0, limit))]
        path, total = [], 0
        while total < length and cur is not None:
            path.append(cur)
            total += len(seq[cur])
            nxt = out.get(cur)
            cur = nxt[int(rng.integers(len(nxt)))] if nxt else None
        if total < length or len(path) < 3:
            continue
        as_bytes = lambda nodes: np.frombuffer("".join(seq[x] for x in nodes).encode(), dtype=np.uint8)
        k = len(path) // 2 if mid_seed else 0
        reverse = both_strands and rng.random() < 0.5
        if not reverse:
            a = add_errors(as_bytes(path[:k]), sub, ins, dele, rng) if k else np.zeros(0, dtype=np.uint8)
            b = add_errors(as_bytes(path[k:]), sub, ins, dele, rng)
            if len(b) < 2 or (k and len(a) < 2):
                continue
            reads.append(np.concatenate([a, b]).tobytes().decode())
            seeds.append((int(path[k]), len(a), False))
        else:
            # the reverse read starts with the reverse complement of the walk's tail
            k = len(path) // 2 if mid_seed else len(path) - 1
            tail = path[k + 1:]
            a = add_errors(revcomp_bytes(as_bytes(tail)), sub, ins, dele, rng) if tail else np.zeros(0, dtype=np.uint8)
            b = add_errors(revcomp_bytes(as_bytes(path[:k + 1])), sub, ins, dele, rng)
            if len(b) < 2 or (tail and len(a) < 2):
                continue
            reads.append(np.concatenate([a, b]).tobytes().decode())
            seeds.append((int(path[k]), len(a), True))
    return reads, seeds


# ---- vg.Graph stream writer (for the chr22-like configuration; shares nothing with the library's decoder) ----------------
def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def vg_bytes(nodes, edges, chunk_nodes=1000):
    """Graph{1: Node{1: sequence, 3: id}, 2: Edge{1: from, 2: to, 3: from_start, 4: to_end}} (vg.pb.h:149-173, 262-284),
    framed as stream.hpp writes it: per group a gzip member holding varint count, then (varint length, message) * count"""
    import gzip
    by_from = {}
    for e in edges:
        by_from.setdefault(e[0], []).append(e)
    members = []
    for lo in range(0, len(nodes), chunk_nodes):
        body = bytearray()
        for nid, seq in nodes[lo:lo + chunk_nodes]:
            sb = seq.encode() if isinstance(seq, str) else seq
            msg = _field(1, 2, _varint(len(sb)) + sb) + _field(3, 0, _varint(nid))
            body += _field(1, 2, _varint(len(msg)) + msg)
        for nid, _ in nodes[lo:lo + chunk_nodes]:
            for f, fs, t, te in by_from.get(nid, ()):
                msg = _field(1, 0, _varint(f)) + _field(2, 0, _varint(t))
                if fs:
                    msg += _field(3, 0, _varint(1))
                if te:
                    msg += _field(4, 0, _varint(1))
                body += _field(2, 2, _varint(len(msg)) + msg)
        members.append(gzip.compress(_varint(1) + _varint(len(body)) + bytes(body), compresslevel=1))
    return b"".join(members)


# ---- several chromosomes (a 12.1 Mbp genome in 16 sequences) ---------------------------------------------------
class MultiGraph:
    """independent SynthGraphs with disjoint node id ranges, presented as one graph"""

    def __init__(self, parts):
        self.parts = parts
        self.nodes = [n for p in parts for n in p.nodes]
        self.edges = [e for p in parts for e in p.edges]

    def gfa(self, overlap=0):
        lines = ["H\tVN:Z:1.0"]
        for nid, seq in self.nodes:
            lines.append("S\t%d\t%s" % (nid, seq))
        for f, fs, t, te in self.edges:
            lines.append("L\t%d\t%s\t%d\t%s\t%dM" % (f, "-" if fs else "+", t, "-" if te else "+", overlap))
        return "\n".join(lines) + "\n"

    def vg_bytes(self, chunk_nodes=1000):
        return vg_bytes(self.nodes, self.edges, chunk_nodes)


def pangenome_graph(total_bp, chromosomes=16, node_len=64, seed=44, snp_every=100, indel_every=1000, sv_every=50000):
    """yeast-like pangenome in `chromosomes` sequences of decreasing length"""
    weights = np.linspace(1.6, 0.4, chromosomes)
    lens = np.maximum(20000, (weights / weights.sum() * total_bp).astype(np.int64))
    parts, first = [], 1
    for c, n in enumerate(lens):
        g = SynthGraph(random_genome(int(n), seed + 100 * c), node_len=node_len, snp_every=snp_every, indel_every=indel_every, sv_every=sv_every,
                       seed=seed + 100 * c + 1, first_id=first)
        first = g._next_id
        parts.append(g)
    return MultiGraph(parts)


def simulate_reads_multi(mg, n_reads, length, seed=1, **kw):
    """reads drawn from the chromosomes in proportion to their length"""
    lens = np.array([len(p.genome) for p in mg.parts], dtype=np.float64)
    share = np.maximum(1, np.round(lens / lens.sum() * n_reads).astype(np.int64))
    while share.sum() > n_reads:
        share[int(np.argmax(share))] -= 1
    while share.sum() < n_reads:
        share[int(np.argmax(lens))] += 1
    reads, seeds = [], []
    for c, (p, k) in enumerate(zip(mg.parts, share)):
        r, s = simulate_reads(p, int(k), length, seed=seed + 7919 * c, **kw)
        reads += r
        seeds += s
    return reads, seeds


# ---- graphs whose bands reach 200 000 cells: the reference's sparse method and backtrace override ----------------
class FanGraph:
    """a head node, a stem, then `n_branches` long branches that start with the same `shared` bases and go their own ways
    after them, each followed by a tail node.  When an alignment nears the stem's end the projected band holds every branch
    whole (GraphAligner.h:1110-1159 works at node granularity), i.e. n_branches x branch_len cells: the case the reference hands
    to calculateSliceAlternate (:2483).  (A seed ON the stem would make slice 0 sparse already: the initial slice scores every
    column of the seed node 0, so its out-neighbours are always projected.)
    Node ids: head 1, stem 2, branches 3 .. n+2, tails n+3 .. 2n+2."""

    def __init__(self, head_len=300, stem_len=3000, n_branches=8, branch_len=30000, shared=150, tail_len=500, seed=1):
        rng = np.random.default_rng(seed)
        rnd = lambda n: _ALPHA[rng.integers(0, 4, size=n, dtype=np.uint8)]
        self.head = rnd(head_len)
        self.stem = rnd(stem_len)
        prefix = rnd(shared)
        self.branches = [np.concatenate([prefix, rnd(branch_len - shared)]) for _ in range(n_branches)]
        self.tails = [rnd(tail_len) for _ in range(n_branches)]
        self.nodes = [(1, self.head.tobytes().decode()), (2, self.stem.tobytes().decode())]
        self.edges = [(1, False, 2, False)]
        for k in range(n_branches):
            self.nodes.append((3 + k, self.branches[k].tobytes().decode()))
            self.edges.append((2, False, 3 + k, False))
        for k in range(n_branches):
            self.nodes.append((3 + n_branches + k, self.tails[k].tobytes().decode()))
            self.edges.append((3 + k, False, 3 + n_branches + k, False))

    def read_through(self, branch, stem_from, length, rng, sub=0.03, ins=0.03, dele=0.03, head_from=0):
        """a read over the head (from `head_from`), the stem from `stem_from` bases in (0: all of it; otherwise the read jumps
        there, which the aligner sees as one long deletion) and on into `branch`; seed = (head, 0, forward)"""
        path = np.concatenate([self.head[head_from:], self.stem[stem_from:], self.branches[branch]])[:length]
        return add_errors(path, sub, ins, dele, rng).tobytes().decode(), (1, 0, False)

    def read_from_branch(self, branch, start, length, rng, sub=0.03, ins=0.03, dele=0.03):
        """a read that starts inside a branch and crosses into its tail; seed on the branch node"""
        path = np.concatenate([self.branches[branch][start:], self.tails[branch]])[:length]
        return add_errors(path, sub, ins, dele, rng).tobytes().decode(), (3 + branch, 0, False)
