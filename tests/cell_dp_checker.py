"""A third formulation of the slice fill, independent of both the oracle's bit vectors and the device program: every cell of
every band column of a slice computed cell by cell from the rules the reference's own (stale, no longer compilable) checker
spells out -- verifySliceBitvector / getWordSliceCellByCell, GraphAligner.h:1601-1679, restated as SURVEY.md Appendix A:

  row j-1 of a node's columns   S(0,j-1) = min([n in PB] P.end(n,0), over in-neighbours m: [m in CB] S(m.last,j-1)+1, [m in PB] P.end(m.last)+1)
                                S(i,j-1) = min(S(i-1,j-1)+1, [n in PB] P.end(n,i));      E(i) = n in PB and P.end(n,i) == S(i,j-1)
  rows j..j+63                  S(w,r) = min(S(w,r-1)+1, S(left,r)+1, S(left,r-1) + mismatch), where the diagonal INTO row j only counts a
                                match when the cell above the left column exists (E(left), and at a node start the in-neighbour was in PB)
  in-neighbour only in PB       its last column is the vertical run P.end(m.last) + 1, 2, ...; only the row-j diagonal may match
  source nodes                  j == 0 and n in PB: P.end(n,0) + mismatch(read[0]) then +1 per row; n in PB: P.end(n,0) + 1, 2, ...;
                                else (len + 1) at rows j-1 and j, then +1 per row

(PB / CB = previous / current band, P.end = previous slice's last row.)  Acyclic bands only.  Pure numpy, for small cases."""
import numpy as np

_MATCH = {"A": "A", "C": "C", "G": "G", "T": "T", "N": "ACGT", "R": "AG", "Y": "CT", "K": "GT", "M": "CA", "S": "CG", "W": "AT",
          "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG"}


def char_match(read_char, graph_char):
    return graph_char in _MATCH[read_char.upper()]


class Digraph:
    """the digraph the loaders build from a bidirected graph (BigraphToDigraph.cpp:27-104): node index 0 is the dummy start,
    bidirected node k (in input order) becomes indices 1 + 2k (forward) and 2 + 2k (reverse complement)"""

    def __init__(self, nodes, edges):
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        self.seq = [""]
        index = {}
        for nid, s in nodes:
            index[2 * nid] = len(self.seq)
            self.seq.append(s)
            index[2 * nid + 1] = len(self.seq)
            self.seq.append("".join(comp[c] for c in reversed(s)))
        self.inn = [[] for _ in self.seq]
        for f, fs, t, te in edges:
            from_left, from_right = (2 * f, 2 * f + 1) if fs else (2 * f + 1, 2 * f)
            to_left, to_right = (2 * t, 2 * t + 1) if te else (2 * t + 1, 2 * t)
            for a, b in ((from_right, to_right), (to_left, from_left)):
                ia, ib = index[a], index[b]
                if ia not in self.inn[ib]:
                    self.inn[ib].append(ia)


def _column_scores(rec):
    """(columns x 65) scores of rows j-1 .. j+63 from a recorded slice's words"""
    vp = rec["vp"][:, None] >> np.arange(64, dtype=np.uint64)[None, :] & np.uint64(1)
    vn = rec["vn"][:, None] >> np.arange(64, dtype=np.uint64)[None, :] & np.uint64(1)
    d = vp.astype(np.int64) - vn.astype(np.int64)
    out = np.empty((len(rec["vp"]), 65), dtype=np.int64)
    out[:, 0] = rec["before"]
    out[:, 1:] = rec["before"][:, None] + np.cumsum(d, axis=1)
    return out


def _node_columns(g, band):
    off, at = {}, 0
    for n in band:
        off[int(n)] = at
        at += len(g.seq[int(n)])
    return off


def check_slice(g, part, prev_rec, rec, padded_len):
    """recompute every cell of `rec` (a slice record of the oracle, direction part `part` = the padded read part) from the previous
    slice's last row and compare; returns the number of cells checked.  prev_rec None = the initial slice (seed node, all zero)."""
    j = rec["j"]
    rows = part[j:j + 64]
    band = [int(n) for n in rec["nodes"]]
    cb = set(band)
    ccol = _node_columns(g, band)
    if prev_rec is None:
        seed = band_seed = None
    want = _column_scores(rec)
    if prev_rec is not None and prev_rec.get("initial"):
        pb_end = {prev_rec["node"]: np.zeros(len(g.seq[prev_rec["node"]]), dtype=np.int64)}
    else:
        pcol = _node_columns(g, [int(n) for n in prev_rec["nodes"]])
        pb_end = {n: prev_rec["end"][c:c + len(g.seq[n])].astype(np.int64) for n, c in pcol.items()}
    # an order in which every in-band in-neighbour comes first
    done, order = set(), []
    pending = list(band)
    while pending:
        progressed = False
        for n in list(pending):
            if all((m not in cb) or (m in done) for m in g.inn[n]):
                order.append(n); done.add(n); pending.remove(n); progressed = True
        if not progressed:
            return -1                       # a cycle in the band: not this checker's case
    S = {}                                   # node -> (len x 65) scores
    E = {}                                   # node -> cell above column i exists
    idx = np.arange(64, dtype=np.int64)
    INF = 1 << 40

    def step(left, left_exists, diag_ok, base, cap_first_only):
        """rows j..j+63 of a column from the column to its left (65 values), before the vertical chain"""
        mism = np.array([0 if char_match(rows[r], base) else 1 for r in range(64)], dtype=np.int64)
        if cap_first_only:
            mism[1:] = 1
        if not (left_exists and diag_ok):
            mism[0] = 1
        return np.minimum(left[1:] + 1, left[:-1] + mism)

    for n in order:
        L = len(g.seq[n])
        in_pb = n in pb_end
        sc = np.zeros((L, 65), dtype=np.int64)
        ex = np.zeros(L, dtype=bool)
        ins = [m for m in g.inn[n] if m in cb or m in pb_end]
        # ---- row j-1 ----
        z = pb_end[n][0] if in_pb else INF
        for m in ins:
            if m in cb:
                z = min(z, S[m][-1, 0] + 1)
            if m in pb_end:
                z = min(z, pb_end[m][-1] + 1)
        if not ins and not in_pb:
            z = padded_len + 1
        sc[0, 0] = z
        for i in range(1, L):
            sc[i, 0] = min(sc[i - 1, 0] + 1, pb_end[n][i] if in_pb else INF)
        if in_pb:
            ex = pb_end[n] == sc[:, 0]
        # ---- column 0 ----
        if not ins:
            if j == 0 and in_pb:
                first = sc[0, 0] + (0 if char_match(rows[0], g.seq[n][0]) else 1)
                cand = np.concatenate([[first], np.full(63, INF)])
                ex0 = True
            elif in_pb:
                cand = np.full(64, INF)
                ex0 = True
            else:
                cand = np.concatenate([[sc[0, 0]], np.full(63, INF)])          # S(j) = S(j-1): the source column's first delta is 0
                ex0 = False
            ex[0] = ex0
        else:
            cand = np.full(64, INF)
            for m in ins:
                if m in cb:
                    left, lex = S[m][-1], bool(E[m][-1])
                    cand = np.minimum(cand, step(left, lex, m in pb_end, g.seq[n][0], False))
                else:
                    left = pb_end[m][-1] + np.arange(65, dtype=np.int64)
                    cand = np.minimum(cand, step(left, True, True, g.seq[n][0], True))
        col = np.minimum.accumulate(np.concatenate([[sc[0, 0]], cand]) - np.arange(65)) + np.arange(65)
        sc[0] = col
        for w in range(1, L):
            cand = step(sc[w - 1], bool(ex[w - 1]), True, g.seq[n][w], False)
            sc[w] = np.minimum.accumulate(np.concatenate([[sc[w, 0]], cand]) - np.arange(65)) + np.arange(65)
        S[n], E[n] = sc, ex
        got = want[ccol[n]:ccol[n] + L]
        if not (got == sc).all():
            w, r = np.argwhere(got != sc)[0]
            raise AssertionError("slice j=%d node %d column %d row %d: oracle %d, cell-by-cell %d" % (j, n, w, j - 1 + r, got[w, r], sc[w, r]))
    return sum(len(g.seq[n]) for n in band) * 65


# ---- slices computed by the sparse method ----------------------------------------------------------------------------
def _scores_of(rec, cols):
    """(len(cols) x 64) scores of rows j .. j+63 of the given columns of a recorded slice"""
    vp = rec["vp"][cols][:, None] >> np.arange(64, dtype=np.uint64)[None, :] & np.uint64(1)
    vn = rec["vn"][cols][:, None] >> np.arange(64, dtype=np.uint64)[None, :] & np.uint64(1)
    return rec["before"][cols].astype(np.int64)[:, None] + np.cumsum(vp.astype(np.int64) - vn.astype(np.int64), axis=1)


def check_sparse_slice(g, part, prev_rec, rec, padded_len):
    """every cell the sparse method wrote (calculateSliceAlternate, GraphAligner.h:2148-2329) against the rule the reference's own
    (stale) checker verifySliceAlternate spells out (:1681-1747): an existing cell equals min(above + 1, left + 1, diagonal +
    mismatch), where a neighbour that does not exist counts as the read's length; a cell exists when the method wrote it
    (confirmedRows.exists), a cell of the previous slice's last row when its scoreEndExists is set; at a node's first column
    "left" and "diagonal" are the minima over the in-neighbours' last columns.  (:1720 reads the previous slice's value at a node's
    first column without asking whether it exists; the method itself only starts from existing cells, :2170-2201, so existence is
    asked for here as everywhere else.)  Returns the number of cells checked."""
    j = rec["j"]
    rows = part[j:j + 64]
    big = padded_len
    band = [int(n) for n in rec["nodes"]]
    ccol = _node_columns(g, band)
    if prev_rec.get("initial"):
        prev_nodes = {prev_rec["node"]: (np.zeros(len(g.seq[prev_rec["node"]]), dtype=np.int64), np.ones(len(g.seq[prev_rec["node"]]), dtype=bool))}
    else:
        pcol = _node_columns(g, [int(n) for n in prev_rec["nodes"]])
        prev_nodes = {n: (prev_rec["end"][c:c + len(g.seq[n])].astype(np.int64), prev_rec["end_exists"][c:c + len(g.seq[n])].astype(bool)) for n, c in pcol.items()}

    def prev_last_row(n, offs):
        """previous slice's row j-1 at the given offsets of node n, `big` where the cell does not exist"""
        if n not in prev_nodes:
            return np.full(len(offs), big, dtype=np.int64)
        end, ex = prev_nodes[n]
        return np.where(ex[offs], end[offs], big)

    def cur_rows(n, offs):
        """(len(offs) x 64) values of this slice at the given offsets of node n, `big` where the cell was not written"""
        if n not in ccol:
            return np.full((len(offs), 64), big, dtype=np.int64)
        cols = ccol[n] + np.asarray(offs, dtype=np.int64)
        wr = (rec["written"][cols][:, None] >> np.arange(64, dtype=np.uint64)[None, :] & np.uint64(1)).astype(bool)
        return np.where(wr, _scores_of(rec, cols), big)

    checked = 0
    for n in band:
        L = len(g.seq[n])
        wr_all = rec["written"][ccol[n]:ccol[n] + L]
        offs = np.nonzero(wr_all)[0]
        if len(offs) == 0:
            continue
        here = cur_rows(n, offs)
        exists = here != big
        row_sets = np.array([sum(1 << "ACGT".index(c) for c in _MATCH[ch.upper()]) for ch in rows], dtype=np.int64)
        base_bit = np.array(["ACGT".index(g.seq[n][o]) for o in offs], dtype=np.int64)
        mism = 1 - ((row_sets[None, :] >> base_bit[:, None]) & 1)
        # the column to the left: the same node's, or the best of the in-neighbours' last columns
        left = np.full((len(offs), 64), big, dtype=np.int64)
        left_above = np.full(len(offs), big, dtype=np.int64)
        inner = offs > 0
        if inner.any():
            left[inner] = cur_rows(n, offs[inner] - 1)
            left_above[inner] = prev_last_row(n, offs[inner] - 1)
        if (~inner).any():
            for m in g.inn[n]:
                last = len(g.seq[m]) - 1
                left[~inner] = np.minimum(left[~inner], cur_rows(m, [last]))
                left_above[~inner] = np.minimum(left_above[~inner], prev_last_row(m, np.array([last])))
        above0 = prev_last_row(n, offs)
        vert = np.concatenate([above0[:, None], here[:, :-1]], axis=1)
        diag = np.concatenate([left_above[:, None], left[:, :-1]], axis=1)
        rule = np.minimum(np.minimum(vert + 1, left + 1), diag + mism)
        bad = exists & (here != rule)
        if bad.any():
            k, r = np.argwhere(bad)[0]
            raise AssertionError("sparse slice j=%d node %d offset %d row %d: oracle %d, rule %d" % (j, n, offs[k], j + r, here[k, r], rule[k, r]))
        checked += int(exists.sum())
    return checked
