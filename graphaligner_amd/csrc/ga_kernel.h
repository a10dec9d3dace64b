// ga_kernel.h -- the per-read extension program: one wavefront owns one read direction.
//
// What it computes is the reference's seeded extension (GraphAligner.h:2571-2856 first pass,
// :894-1021 traceback), restated for a 64-lane wave:
//   * lanes = the 64 read rows of a slice.  A graph column is 64 scores, one per lane; the
//     column recurrence is   S(w,r) = min(S(w-1,r)+1, S(w-1,r-1)+mismatch, S(w,r-1)+1)
//     evaluated as a horizontal/diagonal candidate per lane (one wave_shr DPP move) followed by
//     a wave-wide prefix-min that resolves the vertical chain (GraphAligner.h:1349-1427 is the
//     bit-vector form of the same recurrence).  Column merges (WordSlice.h:361-421) are
//     plain per-lane minima.  VP/VN words fall out of two ballots.
//   * band projection, processing order and traceback are wave-uniform scalar code whose
//     data structures live in LDS; they reproduce the reference's ORDER (hash-map iteration
//     of the frozen slice, std::priority_queue tie order, Tarjan emission order) because the
//     trace start is the LAST minimum in processing order (GraphAligner.h:922,931).
//   * one pass: every slice's VP/VN/score words are kept in HBM (20 B per column) and the
//     traceback reads them directly, instead of the reference's sqrt-checkpoint + recompute.
//
// The same source builds for gfx950 (product, wave primitives of ga_wave.h) and for the host (tests only: tests/emul substitutes its
// own back end for ga_wave.h through GA_WAVE_HEADER).
#pragma once
#include "ga_types.h"
#ifndef GA_WAVE_HEADER
#define GA_WAVE_HEADER "ga_wave.h"
#endif
#include GA_WAVE_HEADER

namespace gak {
using namespace gaw;

constexpr int W = 64;
constexpr uint32_t kCutoff = 200000;         // GraphAlignerCommon.h:10
constexpr int kSliceHdrWords = 12;          // nNodes, nCols, minScore, minSlot, minOffset, flags, logCorrect(2), logWrong(2), slice, reserved
constexpr int kNbr = 4;                      // neighbours per band node cached in LDS

template <int MAXN> struct Limits
{
	static constexpr int kBuckets = MAXN <= 13 ? 13 : MAXN <= 29 ? 29 : MAXN <= 59 ? 59 : MAXN <= 127 ? 127 : 257;   // libstdc++ growth: 13,29,59,127,257
	static constexpr int kHeap = 4 * MAXN;
	static_assert(MAXN <= 257, "bucket schedule only covers 257 band nodes");
};

// LDS-resident per-wave state
template <int MAXN> struct WaveState
{
	// previous band, in band (insertion) order
	uint32_t pn_node[MAXN];
	int32_t pn_min[MAXN];        // node minimum of scoreEnd (NodeSlice MapItem<2>)
	int32_t pn_lastEnd[MAXN];    // scoreEnd of the node's last column
	int32_t pn_lastEnd2[MAXN];   // score one row above it (scoreEnd -/+ last VP/VN bit)
	uint32_t pn_colBase[MAXN];   // first column of the node inside the previous end buffer
	uint32_t pn_len[MAXN];
	uint8_t pn_outDeg[MAXN];     // > kNbr: the list lives in HBM only
	uint32_t pn_outNbr[MAXN * 4];
	// current band, in band order
	uint32_t cn_node[MAXN];
	uint32_t cn_colBase[MAXN];
	uint32_t cn_len[MAXN];
	int16_t cn_prev[MAXN];       // slot in the previous band or -1
	int32_t cn_min[MAXN];
	int32_t cn_lastEnd[MAXN];
	int32_t cn_lastEnd2[MAXN];
	uint64_t cn_lastVP[MAXN];
	uint64_t cn_lastVN[MAXN];
	int32_t cn_lastBefore[MAXN];
	uint8_t cn_lastExists[MAXN];
	// topology of the band nodes, fetched once per slice with all lanes in parallel
	uint32_t cn_startLo[MAXN], cn_startHi[MAXN];   // first column of the node in the graph
	uint8_t cn_inDeg[MAXN], cn_outDeg[MAXN];      // > kNbr: the list lives in HBM only
	uint32_t cn_outNbr[MAXN * 4];
	// where each cached neighbour sits in the current / previous band (-1 = not there), looked up once per slice
	int16_t cn_inSlotC[MAXN * 4], cn_inSlotP[MAXN * 4], cn_outSlotC[MAXN * 4];
	uint8_t color[MAXN];
	int16_t post[MAXN];          // Tarjan emission order
	int16_t low[MAXN];           // Tarjan low-link / component number (bands with cycles only)
	int16_t comp[MAXN];
	// scratch: hash-order emulation / heap / DFS stack
	int16_t h_next[MAXN];
	int16_t h_before[Limits<MAXN>::kBuckets];
	int16_t h_order[MAXN];
	uint32_t heap_node[Limits<MAXN>::kHeap];
	int32_t heap_prio[Limits<MAXN>::kHeap];
	int16_t st_slot[MAXN];
	uint32_t st_cur[MAXN];
};

struct Slot
{
	uint32_t* end_prev;      // per column of the previous slice: scoreEnd << 3 | scoreEndExists << 2 | VN bit 63 << 1 | VP bit 63 (the
	                         // reference's TinySlice, NodeSlice.h:26-31; scoreEndExists is only ever false after a sparse slice, GraphAligner.h:2536)
	uint32_t* end_cur;
	uint32_t* arena;         // slice records
	uint32_t* slice_off;     // [max_slices] word offset of each slice record
	uint8_t* slice_flags;    // [max_slices] bit0 currentlyCorrect, bit1 falseFromCorrect
	uint8_t* trace;          // [trace_cap] staging for the traceback moves of the current job
	uint32_t* ckpt;          // [max_slices + 2] checkpoint records of the reference's DPTable (ramp bookkeeping, wide variants)
	uint32_t* below_off;     // [max_slices + 1] record standing for slice s when the traceback crosses from slice s + 1
	uint8_t* sparse;         // scratch of the sparse method (ga_sparse.h), only in the kernel variant that carries it; else nullptr
	uint32_t* ovr;           // [2 * (max_slices + 1)] backtrace-override windows (first slice, last slice), same variant
	uint32_t sparse_max_bw;  // the bandwidth the sparse scratch was laid out for
};

GA_FN int ctz64(uint64_t m) { return __builtin_ctzll(m); }

// ---- graph access (wave-uniform) -------------------------------------------------------------
GA_FN uint32_t g_len(const GaDevGraph& g, uint32_t n) { return g.node_rec[(uint64_t)n * GA_NODE_REC_WORDS + 2]; }
// the node's 64-byte record, one word per lane (lanes 0..15): a single request
GA_FN VI g_record(const GaDevGraph& g, uint32_t n) { return load_lanes(g.node_rec + (uint64_t)n * GA_NODE_REC_WORDS, GA_NODE_REC_WORDS, 0); }
GA_FN int g_base(const GaDevGraph& g, uint64_t col) { return (int)((g.seq2[col >> 4] >> ((col & 15) * 2)) & 3); }

// slot of `key` in list[0..count) or -1: 64 entries per step, one ballot
GA_FN int find_slot(const uint32_t* list, int count, uint32_t key)
{
	for (int base = 0; base < count; base += LANES)
	{
		VI x = load_lanes(list + base, count - base, -1);
		uint64_t m = ballot(x == (int)key);
		if (m) return base + ctz64(m);
	}
	return -1;
}

// ---- std::unordered_map<size_t,..> iteration order after inserting keys[0..n) one by one ----------
// (NodeSlice.h:728-738 builds the frozen slice's map exactly so; GraphAligner.h:1117 iterates it).
// libstdc++: identity hash, bucket = key % B, B grows 13,29,59,127,257 when size would exceed it;
// a node entering an empty bucket becomes the list head, otherwise it goes to the front of its
// bucket's run; a rehash re-inserts the nodes in their current iteration order.
template <int MAXN> GA_FN void hash_insert(WaveState<MAXN>& ws, const uint32_t* keys, int i, int B, int& head)
{
	int b = (int)(keys[i] % (uint32_t)B);
	int before = ws.h_before[b];
	if (before != -1)
	{
		if (before == -2) { ws.h_next[i] = (int16_t)head; head = i; }
		else { ws.h_next[i] = ws.h_next[before]; ws.h_next[before] = (int16_t)i; }
	}
	else
	{
		ws.h_next[i] = (int16_t)head;
		if (head != -1) ws.h_before[keys[head] % (uint32_t)B] = (int16_t)i;
		head = i;
		ws.h_before[b] = -2;
	}
}

template <int MAXN> GA_FN void hash_order(WaveState<MAXN>& ws, const uint32_t* keys, int n)
{
	int B = 13, head = -1;
	for (int b = 0; b < B; b++) ws.h_before[b] = -1;
	for (int i = 0; i < n; i++)
	{
		int grown = i == 13 ? 29 : i == 29 ? 59 : i == 59 ? 127 : i == 127 ? 257 : 0;
		if (grown)
		{
			int k = 0;
			for (int p = head; p >= 0; p = ws.h_next[p]) ws.h_order[k++] = (int16_t)p;
			B = grown;
			head = -1;
			for (int b = 0; b < B; b++) ws.h_before[b] = -1;
			for (int q = 0; q < k; q++) hash_insert(ws, keys, ws.h_order[q], B, head);
		}
		hash_insert(ws, keys, i, B, head);
	}
	int k = 0;
	for (int p = head; p >= 0; p = ws.h_next[p]) ws.h_order[k++] = (int16_t)p;
}

// The same order computed with lanes = elements (n <= 64).  Inserting a sequence of distinct keys into an empty
// table with B buckets leaves the list grouped by bucket, the groups ordered by the time their bucket was first
// used (latest first) and each group by insertion time (latest first): a node entering an empty bucket goes to
// the list head, any other node to the front of its bucket's group.  A rehash re-inserts the current list, in
// list order, into the grown table, after which insertion continues; so every growth stage is one such
// grouping of the sequence "current list, then the keys added before the next growth".
template <int B> GA_FN VI hash_stage_rank(const VI key, int m)
{
	const VI lane = lane_iota();
	const VI b = vmod<B>(key);
	// f = first position in the sequence whose key shares my bucket
	VI f = lane;
	for (int j = m - 1; j >= 0; j--) f = select(b == read_lane(b, j), VI(j), f);
	// rank = how many elements precede me: later-opened groups, then later members of my own group
	VI rank = VI(0);
	for (int j = 0; j < m; j++)
	{
		const int fj = read_lane(f, j);
		rank = rank + select((f < fj) , VI(1), select((f == fj) && (lane < j), VI(1), VI(0)));
	}
	return rank;
}
template <int MAXN> GA_FN void hash_order_lanes(WaveState<MAXN>& ws, const uint32_t* keys, int n)
{
	const VI lane = lane_iota();
	int done = 0;                                                // elements already in the list (h_order[0..done) = list order)
	for (int stage = 0; done < n; stage++)
	{
		const int upto = stage == 0 ? 13 : stage == 1 ? 29 : stage == 2 ? 59 : 127;     // the table grows when the 14th, 30th, 60th key arrives
		const int m = n < upto ? n : upto;
		const VB live = lane < m;
		const VI elem = select(lane < done, load_lanes(ws.h_order, done, 0), lane);
		const VI key = select(live, gather(keys, select(live, elem, VI(0))), VI(0));
		VI rank;
		if (stage == 0) rank = hash_stage_rank<13>(key, m);
		else if (stage == 1) rank = hash_stage_rank<29>(key, m);
		else if (stage == 2) rank = hash_stage_rank<59>(key, m);
		else rank = hash_stage_rank<127>(key, m);
		wave_order();
		scatter(ws.h_order, rank, elem, live);
		wave_order();
		done = m;
	}
}

// ---- std::priority_queue<.., std::greater<>> over (node, priority), libstdc++ heap algorithms -----
// (GraphAligner.h:1115; the pop order among equal priorities decides DPSlice::nodes order)
template <int MAXN> GA_FN void heap_sift_up(WaveState<MAXN>& ws, int hole, int top, uint32_t node, int prio)
{
	int parent = (hole - 1) / 2;
	while (hole > top && ws.heap_prio[parent] > prio)
	{
		ws.heap_prio[hole] = ws.heap_prio[parent];
		ws.heap_node[hole] = ws.heap_node[parent];
		hole = parent;
		parent = (hole - 1) / 2;
	}
	ws.heap_prio[hole] = prio;
	ws.heap_node[hole] = node;
}
template <int MAXN> GA_FN bool heap_push(WaveState<MAXN>& ws, int& size, uint32_t node, int prio)
{
	if (size >= Limits<MAXN>::kHeap) return false;
	size++;
	heap_sift_up(ws, size - 1, 0, node, prio);
	return true;
}
template <int MAXN> GA_FN void heap_pop(WaveState<MAXN>& ws, int& size)
{
	// std::pop_heap then pop_back: the last element is re-inserted from the root (__adjust_heap)
	int len = size - 1;
	uint32_t node = ws.heap_node[len];
	int prio = ws.heap_prio[len];
	size = len;
	if (len == 0) return;
	int hole = 0, child = 0;
	while (child < (len - 1) / 2)
	{
		child = 2 * (child + 1);
		if (ws.heap_prio[child] > ws.heap_prio[child - 1]) child--;
		ws.heap_prio[hole] = ws.heap_prio[child];
		ws.heap_node[hole] = ws.heap_node[child];
		hole = child;
	}
	if ((len & 1) == 0 && child == (len - 2) / 2)
	{
		child = 2 * (child + 1);
		ws.heap_prio[hole] = ws.heap_prio[child - 1];
		ws.heap_node[hole] = ws.heap_node[child - 1];
		hole = child - 1;
	}
	heap_sift_up(ws, hole, 0, node, prio);
}

// The same heap with its array spread over the lanes of a few registers (entry k = lane k & 63 of register k >> 6):
// a sift then costs v_readlane / v_writelane moves instead of dependent LDS round trips.  Used while the heap fits.
template <int NREG> struct LaneHeap
{
	VI node[NREG], prio[NREG];
	GA_FN int getPrio(int k) const { VI v = prio[0]; for (int i = 1; i < NREG; i++) if ((k >> 6) == i) v = prio[i]; return read_lane(v, k & 63); }
	GA_FN int getNode(int k) const { VI v = node[0]; for (int i = 1; i < NREG; i++) if ((k >> 6) == i) v = node[i]; return read_lane(v, k & 63); }
	GA_FN void set(int k, int n, int p)
	{
		for (int i = 0; i < NREG; i++)
			if ((k >> 6) == i) { node[i] = write_lane(node[i], n, k & 63); prio[i] = write_lane(prio[i], p, k & 63); }
	}
	GA_FN void siftUp(int hole, int n, int p)
	{
		int parent = (hole - 1) / 2;
		while (hole > 0)
		{
			const int pp = getPrio(parent);
			if (!(pp > p)) break;
			set(hole, getNode(parent), pp);
			hole = parent;
			parent = (hole - 1) / 2;
		}
		set(hole, n, p);
	}
	GA_FN bool push(int& size, uint32_t n, int p)
	{
		if (size >= NREG * LANES) return false;
		size++;
		siftUp(size - 1, (int)n, p);
		return true;
	}
	GA_FN void pop(int& size)
	{
		// std::pop_heap then pop_back: the last element is re-inserted from the root (__adjust_heap)
		const int len = size - 1;
		const int n = getNode(len), p = getPrio(len);
		size = len;
		if (len == 0) return;
		int hole = 0, child = 0;
		while (child < (len - 1) / 2)
		{
			child = 2 * (child + 1);
			int cp = getPrio(child);
			const int lp = getPrio(child - 1);
			if (cp > lp) { child--; cp = lp; }
			set(hole, getNode(child), cp);
			hole = child;
		}
		if ((len & 1) == 0 && child == (len - 2) / 2)
		{
			child = 2 * (child + 1);
			set(hole, getNode(child - 1), getPrio(child - 1));
			hole = child - 1;
		}
		siftUp(hole, n, p);
	}
};

// ---- band selection at node granularity (GraphAligner.h:1110-1159) ---------------------------------
template <int MAXN> GA_FN int project_band(const GaDevGraph& g, WaveState<MAXN>& ws, int pn, int prevMin, int bandwidth, int& cn, uint32_t& totalCols)
{
	const int expand = bandwidth + W;
	cn = 0;
	totalCols = 0;
	int heapSize = 0;
	constexpr bool kLaneHeap = Limits<MAXN>::kHeap <= 4 * LANES;
	LaneHeap<kLaneHeap ? Limits<MAXN>::kHeap / LANES : 1> lh;
	if constexpr (kLaneHeap) for (int i = 0; i < Limits<MAXN>::kHeap / LANES; i++) { lh.node[i] = VI(0); lh.prio[i] = VI(0); }
	auto push = [&](uint32_t node, int prio) -> bool {
		if constexpr (kLaneHeap) return lh.push(heapSize, node, prio);
		else return heap_push(ws, heapSize, node, prio);
	};
	if (pn <= LANES) hash_order_lanes(ws, ws.pn_node, pn); else hash_order(ws, ws.pn_node, pn);
	auto add = [&](uint32_t node, int prevSlot, uint32_t len) -> bool {
		if (cn >= MAXN) return false;
		ws.cn_node[cn] = node;
		ws.cn_prev[cn] = (int16_t)prevSlot;
		ws.cn_len[cn] = len;
		ws.cn_colBase[cn] = totalCols;
		totalCols += len;
		cn++;
		return true;
	};
	auto pushFromGraph = [&](const VI& rec, uint32_t node, int prio) -> bool {
		const uint32_t deg = (uint32_t)read_lane(rec, 3) >> 16;
		if (deg <= (uint32_t)kNbr)
		{
			for (uint32_t e = 0; e < deg; e++) if (!push((uint32_t)read_lane(rec, 4 + (int)e), prio)) return false;
			return true;
		}
		for (uint32_t e = g.out_off[node]; e < g.out_off[node + 1]; e++) if (!push(g.out_nbr[e], prio)) return false;
		return true;
	};
	for (int k = 0; k < pn; k++)
	{
		int s = ws.h_order[k];
		if (ws.pn_min[s] > prevMin + bandwidth) continue;
		uint32_t node = ws.pn_node[s];
		if (!add(node, s, ws.pn_len[s])) return GA_CAP_NODES;
		if (totalCols >= kCutoff) return GA_UNSUPPORTED_BAND;
		int endScore = ws.pn_lastEnd[s];
		if (endScore > prevMin + expand) continue;
		int deg = ws.pn_outDeg[s];
		if (deg <= kNbr)
		{
			for (int e = 0; e < deg; e++)
				if (!push(ws.pn_outNbr[s * kNbr + e], endScore - prevMin + 1)) return GA_CAP_HEAP;
		}
		else if (!pushFromGraph(g_record(g, node), node, endScore - prevMin + 1)) return GA_CAP_HEAP;
	}
	if (cn == 0) return GA_ASSERTION;                                   // assert(distances.size() > 0) (:1138)
	while (heapSize > 0)
	{
		uint32_t node;
		int prio;
		if constexpr (kLaneHeap) { node = (uint32_t)lh.getNode(0); prio = lh.getPrio(0); }
		else { node = ws.heap_node[0]; prio = ws.heap_prio[0]; }
		if (prio > expand) break;
		if constexpr (kLaneHeap) lh.pop(heapSize); else heap_pop(ws, heapSize);
		if (find_slot(ws.cn_node, cn, node) >= 0) continue;             // already at a distance <= prio
		int ps = find_slot(ws.pn_node, pn, node);
		if (ps >= 0 && ws.pn_outDeg[ps] <= kNbr)
		{
			const uint32_t len = ws.pn_len[ps];
			if (!add(node, ps, len)) return GA_CAP_NODES;
			if (totalCols >= kCutoff) return GA_UNSUPPORTED_BAND;
			for (int e = 0; e < ws.pn_outDeg[ps]; e++)
				if (!push(ws.pn_outNbr[ps * kNbr + e], prio + (int)len)) return GA_CAP_HEAP;
		}
		else
		{
			// a node new to the band: its length and out-list arrive together in its record
			const VI rec = g_record(g, node);
			const uint32_t len = (uint32_t)read_lane(rec, 2);
			if (!add(node, ps, len)) return GA_CAP_NODES;
			if (totalCols >= kCutoff) return GA_UNSUPPORTED_BAND;
			if (!pushFromGraph(rec, node, prio + (int)len)) return GA_CAP_HEAP;
		}
	}
	return GA_OK;
}

// ---- fetch the band nodes' topology into LDS: lanes = band nodes, one round trip to HBM/L2 ------------
// The neighbours' positions in the current and the previous band are looked up here as well, for all band
// nodes at once (one pass over the band lists, every lane comparing its own neighbours), so that the
// per-node code finds them with one LDS read instead of a search.
template <int MAXN> GA_FN void load_topology(const GaDevGraph& g, WaveState<MAXN>& ws, int pn, int cn)
{
	const VI lane = lane_iota();
	for (int base = 0; base < cn; base += LANES)
	{
		VB live = lane < (cn - base);
		VI slot = lane + base;
		VI node = select(live, load_lanes(ws.cn_node + base, cn - base, 0), VI(0));
		// (64-bit record addressing: whole-genome graphs have more than 2^27 directed nodes)
		VI lo = gather_rec(g.node_rec, node, 0);
		VI hi = gather_rec(g.node_rec, node, 1);
		VI degs = gather_rec(g.node_rec, node, 3);
		VI inDeg = degs & 0xffff, outDeg = (degs >> 16) & 0xffff;
		scatter(ws.cn_startLo, slot, lo, live);
		scatter(ws.cn_startHi, slot, hi, live);
		scatter(ws.cn_inDeg, slot, vmin(inDeg, VI(255)), live);
		scatter(ws.cn_outDeg, slot, vmin(outDeg, VI(255)), live);
		VI inN[kNbr], outN[kNbr], inC[kNbr], inP[kNbr], outC[kNbr];
		for (int k = 0; k < kNbr; k++)
		{
			VB hasIn = live && (VI(k) < inDeg);
			VB hasOut = live && (VI(k) < outDeg);
			inN[k] = select(hasIn, gather_rec(g.node_rec, node, 8 + k), VI(-1));
			outN[k] = select(hasOut, gather_rec(g.node_rec, node, 4 + k), VI(-1));
			scatter(ws.cn_outNbr, (slot << 2) + k, outN[k], hasOut);
			inC[k] = VI(-1); inP[k] = VI(-1); outC[k] = VI(-1);
		}
		for (int jb = 0; jb < cn; jb += LANES)
		{
			const int m = cn - jb < LANES ? cn - jb : LANES;
			const VI nodesJ = load_lanes(ws.cn_node + jb, m, -2);
			for (int j = 0; j < m; j++)
			{
				const int nj = read_lane(nodesJ, j);
				for (int k = 0; k < kNbr; k++)
				{
					inC[k] = select(inN[k] == nj, VI(jb + j), inC[k]);
					outC[k] = select(outN[k] == nj, VI(jb + j), outC[k]);
				}
			}
		}
		for (int jb = 0; jb < pn; jb += LANES)
		{
			const int m = pn - jb < LANES ? pn - jb : LANES;
			const VI nodesJ = load_lanes(ws.pn_node + jb, m, -2);
			for (int j = 0; j < m; j++)
			{
				const int nj = read_lane(nodesJ, j);
				for (int k = 0; k < kNbr; k++) inP[k] = select(inN[k] == nj, VI(jb + j), inP[k]);
			}
		}
		for (int k = 0; k < kNbr; k++)
		{
			scatter(ws.cn_inSlotC, (slot << 2) + k, inC[k], live);
			scatter(ws.cn_inSlotP, (slot << 2) + k, inP[k], live);
			scatter(ws.cn_outSlotC, (slot << 2) + k, outC[k], live);
		}
	}
}

// where the k-th in-neighbour of band slot s sits in the current (cs) and the previous (pm) band, -1 = absent
template <int MAXN> GA_FN void in_slots(const GaDevGraph& g, const WaveState<MAXN>& ws, int pn, int cn, int s, int k, int& cs, int& pm)
{
	if (ws.cn_inDeg[s] <= kNbr) { cs = ws.cn_inSlotC[s * kNbr + k]; pm = ws.cn_inSlotP[s * kNbr + k]; return; }
	const uint32_t m = g.in_nbr[g.in_off[ws.cn_node[s]] + k];
	cs = find_slot(ws.cn_node, cn, m);
	pm = find_slot(ws.pn_node, pn, m);
}
// where the k-th out-neighbour of band slot s sits in the current band, -1 = absent
template <int MAXN> GA_FN int out_slot(const GaDevGraph& g, const WaveState<MAXN>& ws, int cn, int s, int k)
{
	if (ws.cn_outDeg[s] <= kNbr) return ws.cn_outSlotC[s * kNbr + k];
	return find_slot(ws.cn_node, cn, g.out_nbr[g.out_off[ws.cn_node[s]] + k]);
}
template <int MAXN> GA_FN int in_degree(const GaDevGraph& g, const WaveState<MAXN>& ws, int s)
{
	if (ws.cn_inDeg[s] <= kNbr) return ws.cn_inDeg[s];
	return (int)(g.in_off[ws.cn_node[s] + 1] - g.in_off[ws.cn_node[s]]);
}
template <int MAXN> GA_FN int out_degree(const GaDevGraph& g, const WaveState<MAXN>& ws, int s)
{
	if (ws.cn_outDeg[s] <= kNbr) return ws.cn_outDeg[s];
	return (int)(g.out_off[ws.cn_node[s] + 1] - g.out_off[ws.cn_node[s]]);
}

// ---- processing order: reverse Tarjan emission order over the band subgraph (:1836-1901, :2360) ----
// On a DAG every node is its own component and is emitted when its DFS finishes; an edge to a
// node still on the DFS stack means a cycle, which this kernel does not handle.
template <int MAXN> GA_FN int processing_order(const GaDevGraph& g, WaveState<MAXN>& ws, int cn)
{
	for (int c = 0; c < cn; c += LANES) store_lanes(ws.color + c, cn - c, VI(0));
	int emitted = 0;
	for (int root = 0; root < cn; root++)
	{
		if (ws.color[root] != 0) continue;
		int sp = 0;
		ws.color[root] = 1;
		ws.st_slot[0] = (int16_t)root;
		ws.st_cur[0] = 0;
		sp = 1;
		while (sp > 0)
		{
			int v = ws.st_slot[sp - 1];
			int cur = (int)ws.st_cur[sp - 1];
			if (cur < out_degree(g, ws, v))
			{
				int x = out_slot(g, ws, cn, v, cur);
				if (x < 0 || ws.color[x] == 2) { ws.st_cur[sp - 1] = (uint32_t)(cur + 1); continue; }
				if (ws.color[x] == 1) return GA_UNSUPPORTED_CYCLE;
				ws.color[x] = 1;
				ws.st_slot[sp] = (int16_t)x;
				ws.st_cur[sp] = 0;
				sp++;
				continue;
			}
			ws.color[v] = 2;
			ws.post[emitted++] = (int16_t)v;
			sp--;
			if (sp > 0) ws.st_cur[sp - 1] += 1;
		}
	}
	return GA_OK;
}

// ---- slice record layout inside the slot arena ---------------------------------------------------------
struct SliceRec
{
	uint32_t* hdr;       // kSliceHdrWords
	uint32_t* nodes;     // [nNodes]
	uint32_t* colBase;   // [nNodes]
	int32_t* nodeMin;    // [nNodes] the node minimum the next slice's band selection reads (NodeSlice MapItem<2>)
	uint64_t* vp;        // [nCols]
	uint64_t* vn;        // [nCols]
	int32_t* before;     // [nCols]
};
GA_FN uint64_t slice_words(uint32_t nNodes, uint32_t nCols)
{
	uint64_t w = kSliceHdrWords + 3ull * nNodes;
	w += w & 1;                               // 8-byte alignment for the 64-bit planes
	return w + 5ull * nCols + (nCols & 1);
}
GA_FN SliceRec slice_at(uint32_t* arena, uint64_t off, uint32_t nNodes, uint32_t nCols)
{
	SliceRec r;
	r.hdr = arena + off;
	r.nodes = r.hdr + kSliceHdrWords;
	r.colBase = r.nodes + nNodes;
	r.nodeMin = (int32_t*)(r.colBase + nNodes);
	uint64_t w = kSliceHdrWords + 3ull * nNodes;
	w += w & 1;
	r.vp = (uint64_t*)(arena + off + w);
	r.vn = r.vp + nCols;
	r.before = (int32_t*)(r.vn + nCols);
	return r;
}

// ---- fill one slice: every band node in processing order (GraphAligner.h:2331-2451, 1457-1573) --------
// Two lane layouts alternate:
//  * lanes = the 64 read rows.  Lane r holds T_r = S_r - r for row j+r of the current column
//    (S = cell score).  The column recurrence  S'_r = min(S_r + 1, S_{r-1} + mismatch_r, S'_{r-1} + 1)
//    becomes   g_r = min(T_r + 1, T_{r-1} - eq_r);   T'_r = min(prefix_min(g)_r, before' + 1)
//    with T_{-1} = before + 1 (before = score at row j-1): one DPP shift, a few VALU ops and a
//    six-step DPP scan per column.  Vertical deltas are T_r - T_{r-1} + 1, so VP / VN are two
//    compares whose 64-bit results are the words the reference keeps (WordSlice.h:194-195).
//  * lanes = up to 64 consecutive COLUMNS of the node (a chunk).  Everything the reference derives
//    column by column for the virtual row j-1 -- forceComponentZeroRow's chain (:1939-1944),
//    scoreBeforeStart / scoreBeforeExists of getNextSlice (:1358-1370) and the vertical re-entry
//    test (:1541-1546) -- is a min-plus recurrence along the node and is evaluated for the whole
//    chunk with the same DPP scan, then read back one lane per column.
// returns status; outputs slice min and the LAST column (in processing order) that attains it
template <int MAXN>
GA_FN int fill_slice(const GaDevGraph& g, WaveState<MAXN>& ws, const Slot& slot, const SliceRec& rec, const VI rowCode, const int rowAboveCode, uint32_t nRows,
                     uint32_t j, int pn, int cn, int& sliceMin, int& minSlot, uint32_t& minOffset)
{
	const VI lane = lane_iota();
	const VI notLane0 = select(lane == 0, VI(0), VI(1));
	const VU lowMask = low_mask_through_lane();
	// match bits of the row for bases 0..3, and in bits 4..7 the same with lane 0 forced to "no match"
	const VI rowCode2 = (rowCode & 15) | (((rowCode & 15) * notLane0) << 4);
	if (ballot((rowCode & GA_ROW_INVALID) != 0)) return GA_ASSERTION;          // characterMatch default branch (:2104-2106)
	const int rawAbove = j > 0 ? (rowAboveCode >> 4) & 7 : 7;                    // read char of row j-1, exact-compare code
	sliceMin = INF;
	minSlot = -1;
	minOffset = 0;

	// operands of a node's first chunk (graph bases, previous-slice end words: one column per lane) are
	// requested one node ahead, so their HBM/L2 latency hides behind the previous node's columns
	VI nextBaseV = VI(0), nextPendV = VI(0);
	auto requestNode = [&](int oiNext) {
		const int sn = ws.post[oiNext];
		const uint64_t fc = ((uint64_t)ws.cn_startHi[sn] << 32) | ws.cn_startLo[sn];
		const int psn = ws.cn_prev[sn];
		VI atn = lane + (int)(fc & 15);
		nextBaseV = (gather(g.seq2 + (fc >> 4), atn >> 4) >> ((atn & 15) << 1)) & 3;
		nextPendV = psn >= 0 ? load_lanes(slot.end_prev + ws.pn_colBase[psn], (int)ws.cn_len[sn], 0) : VI(0);
	};
	if (cn > 0) requestNode(cn - 1);
	for (int oi = cn - 1; oi >= 0; oi--)
	{
		const int s = ws.post[oi];
		const uint32_t len = ws.cn_len[s];
		const uint64_t firstCol = ((uint64_t)ws.cn_startHi[s] << 32) | ws.cn_startLo[s];
		const int ps = ws.cn_prev[s];
		const bool inPrev = ps >= 0;
		const uint32_t* pend = slot.end_prev + (inPrev ? ws.pn_colBase[ps] : 0);
		const uint32_t outBase = ws.cn_colBase[s];
		const uint32_t bit0 = (uint32_t)(firstCol & 15);
		const uint32_t* seqWords = g.seq2 + (firstCol >> 4);
		const bool aboveAlways = j == 0 && inPrev;                              // "previousEq" (:1503): raw char ==, not characterMatch

		VI at;
		VI baseV = nextBaseV;
		VI pendRawV = nextPendV;
		if (oi > 0) requestNode(oi - 1);

		// --- row j-1 of column 0 (forceComponentZeroRow for a single acyclic node, :1916-1937) ---
		const int inDeg = in_degree(g, ws, s);
		const int pend0raw = read_lane(pendRawV, 0);
		const int pend0 = inPrev ? (pend0raw >> 3) : INF;
		const bool pex0 = inPrev && ((pend0raw >> 2) & 1) != 0;                  // the cell above column 0 exists (scoreEndExists)
		int zero0 = pend0;
		bool hasIn = false;
		for (int e = 0; e < inDeg; e++)
		{
			int cs, pm;
			in_slots(g, ws, pn, cn, s, e, cs, pm);
			if (cs < 0 && pm < 0) continue;
			hasIn = true;
			if (cs >= 0) zero0 = zero0 < ws.cn_lastBefore[cs] + 1 ? zero0 : ws.cn_lastBefore[cs] + 1;
			if (pm >= 0) zero0 = zero0 < ws.pn_lastEnd[pm] + 1 ? zero0 : ws.pn_lastEnd[pm] + 1;
		}
		const int base0 = read_lane(baseV, 0);
		const VI eqLane0 = bit_extract(rowCode, base0);
		const bool aboveEq0 = aboveAlways || (j > 0 && rawAbove == base0);
		const bool exists0 = inPrev && pend0 == zero0 && pex0;                   // scoreBeforeExists from :1989
		VI T;
		int before0;
		bool existsFirst;
		if (!hasIn)
		{
			// source node (:1475-1499): a vertical run from the cell above
			if (j == 0 && inPrev) { T = VI(pend0 + 1 - read_lane(eqLane0, 0)); before0 = pend0; existsFirst = true; }
			else if (inPrev) { T = VI(pend0 + 1); before0 = pend0; existsFirst = pex0; }              // (:1333-1337)
			else { T = VI((int)(nRows + 1)); before0 = (int)(nRows + 1); existsFirst = false; }
		}
		else
		{
			// node start: cell-wise min over the in-neighbours' last columns advanced one step (:1270-1315)
			VI G = VI(INF);
			int calc = INF;
			for (int e = 0; e < inDeg; e++)
			{
				int cs, pm;
				in_slots(g, ws, pn, cn, s, e, cs, pm);
				if (cs < 0 && pm < 0) continue;
				VI leftT, eq;
				int leftBefore;
				bool leftExists;
				if (cs >= 0)
				{
					leftBefore = ws.cn_lastBefore[cs];
					leftT = vpopc(ws.cn_lastVP[cs] & lowMask) - vpopc(ws.cn_lastVN[cs] & lowMask) + leftBefore - lane;
					leftExists = ws.cn_lastExists[cs] != 0;
					eq = eqLane0;
				}
				else
				{
					// neighbour only in the previous band: vertical source column, only row j may match (:1294-1301)
					leftBefore = ws.pn_lastEnd[pm];
					leftT = VI(leftBefore + 1);
					leftExists = true;
					eq = select(lane == 0, eqLane0, VI(0));
				}
				if (!(leftExists && pm >= 0)) eq = eq & notLane0;                       // Eq bit 0 masked (:1358,1360)
				VI shl = shr1(leftT, leftBefore + 1);
				G = vmin(G, vmin(leftT + 1, shl - eq));
				int viaLeft = leftBefore + 1;
				if (exists0 && pm >= 0)
				{
					int viaDiag = ws.pn_lastEnd2[pm] + (aboveEq0 ? 0 : 1);
					viaLeft = viaLeft < viaDiag ? viaLeft : viaDiag;
				}
				calc = calc < viaLeft ? calc : viaLeft;
			}
			bool reenter = inPrev && calc > pend0;                               // vertical re-entry (:1504-1509)
			before0 = reenter ? pend0 : calc;
			existsFirst = reenter ? pex0 : exists0;                              // mergable.scoreBeforeExists = oldSlice[0].scoreEndExists (:1507)
			T = vmin(prefix_min(G), VI(before0 + 1));
		}

		int nodeMin = INF;
		int carryZero = INF, carryBefore = INF, carryAbove2 = 0;
		bool carryExists = false;
		VI sh = VI(0), Tp1 = VI(0);
		uint64_t vp = 0, vn = 0;
		int lastBefore = before0, lastEnd = 0, lastZero = zero0;
		bool lastExists = existsFirst;
		for (uint32_t w0 = 0; w0 < len; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			const bool first = w0 == 0;
			if (!first)
			{
				at = lane + (int)(bit0 + w0);
				baseV = (gather(seqWords, at >> 4) >> ((at & 15) << 1)) & 3;
				pendRawV = inPrev ? load_lanes(pend + w0, n, 0) : VI(0);
			}
			// ---- the row j-1 bookkeeping of the whole chunk, lanes = columns ----
			const VB live = lane < n;
			const VI pendV = inPrev ? select(live, pendRawV >> 3, VI(INF)) : VI(INF);
			const VI pexV = (pendRawV >> 2) & 1;                                 // scoreEndExists of the cell above
			const VI above2own = (pendRawV >> 3) - (pendRawV & 1) + ((pendRawV >> 1) & 1);      // score at row j-2 of the own column
			const VI above2left = shr1(above2own, carryAbove2);
			const VI aboveEqV = aboveAlways ? VI(1) : (j > 0 ? select(baseV == rawAbove, VI(1), VI(0)) : VI(0));
			// zero row (:1939-1944): zero[w] = min(zero[w-1] + 1, pend[w])
			const VI zeroIn = first ? select(lane == 0, VI(zero0), pendV) : pendV;
			const VI zeroV = vmin(prefix_min(zeroIn - lane) + lane, first ? VI(INF) : lane + (carryZero + 1));
			const VI existsWV = inPrev ? select(pendV == zeroV, pexV, VI(0)) : VI(0);
			const VI viaDiagV = select(existsWV != 0, above2left + 1 - aboveEqV, VI(INF));   // :1369
			// scoreBeforeStart (:1361-1370, then re-entry :1541-1546): before[w] = min(before[w-1] + 1, viaDiag[w], pend[w])
			const VI cIn = first ? select(lane == 0, VI(before0), vmin(pendV, viaDiagV)) : vmin(pendV, viaDiagV);
			const VI beforeV = vmin(prefix_min(cIn - lane) + lane, first ? VI(INF) : lane + (carryBefore + 1));
			const VI beforeLeft = shr1(beforeV, carryBefore);
			const VI calcV = vmin(beforeLeft + 1, viaDiagV);
			const VB reenterV = (calcV > pendV) && inPrev;
			VI existsV = select(reenterV, pexV, existsWV);                     // final scoreBeforeExists of every column of the chunk (:1544)
			if (first) existsV = select(lane == 0, VI(existsFirst ? 1 : 0), existsV);
			// One scalar per column steers the inner loop: the bit offset into rowCode2 = graph base, +4 when the
			// column to the left has no existing cell above it (then lane 0 must not see a match, :1358).
			const VI offV = baseV + select(shr1(existsV, carryExists ? 1 : 0) != 0, VI(0), VI(4));
			const VI beforeP1V = beforeV + 1;

			VI accVpLo = VI(0), accVpHi = VI(0), accVnLo = VI(0), accVnHi = VI(0);
			// The column loop is software-pipelined by one column: while the DPP scan of column c runs (a
			// chain of dependent instructions), the words of column c-1 are formed and put away.
			int c = 0;
			if (!first)
			{
				// first column of a later chunk: step from the last column of the previous chunk
				const int bp1 = read_lane(beforeP1V, 0);
				const VI eq = bit_extract(rowCode2, read_lane(offV, 0));
				T = vmin(prefix_min(vmin(Tp1, sh - eq)), VI(bp1));
			}
			{
				const int bp1 = read_lane(beforeP1V, 0);
				sh = shr1(T, bp1);
				Tp1 = T + 1;
			}
			// the two per-column scalars (T of the virtual row j-1, bit offset into rowCode2) are fetched one
			// column ahead with a uniform ds_bpermute: they arrive in VGPRs without spending VALU cycles.  The loop
			// is unrolled by two so that the fetched pair alternates between two register pairs instead of being copied.
			auto column = [&](const VI& bp1v, const VI& offv, int col) {
				// ---- column w0+col from the column to its left (calculateNode :1533-1546, getNextSlice :1349-1427) ----
				const VI eq = bit_extract_v(rowCode2, offv);
				// the cap before' + 1 is the same for every row, so it can go inside the scan: one v_min3 feeds six fused v_min_dpp
				const VI G = vmin(vmin(Tp1, sh - eq), bp1v);
				// ---- emit column col-1: vertical deltas against the row above (row j-1 holds T = before + 1) ----
				vp = ballot(T == sh);                                            // delta +1
				vn = ballot(Tp1 < sh);                                           // delta -1
				accVpLo = write_lane(accVpLo, (int)(uint32_t)vp, col - 1);
				accVpHi = write_lane(accVpHi, (int)(uint32_t)(vp >> 32), col - 1);
				accVnLo = write_lane(accVnLo, (int)(uint32_t)vn, col - 1);
				accVnHi = write_lane(accVnHi, (int)(uint32_t)(vn >> 32), col - 1);
				T = prefix_min(G);
				sh = shr1v(T, bp1v);
				Tp1 = T + 1;
			};
			const int nU = wave_uniform(n);
			VI bpA = lane_broadcast(beforeP1V, 1), offA = lane_broadcast(offV, 1);
			for (c = 1; c + 1 < nU; c += 2)
			{
				const VI bpB = lane_broadcast(beforeP1V, c + 1), offB = lane_broadcast(offV, c + 1);
				column(bpA, offA, c);
				bpA = lane_broadcast(beforeP1V, c + 2);
				offA = lane_broadcast(offV, c + 2);
				column(bpB, offB, c + 1);
			}
			if (c < nU) column(bpA, offA, c);
			vp = ballot(T == sh);
			vn = ballot(Tp1 < sh);
			accVpLo = write_lane(accVpLo, (int)(uint32_t)vp, n - 1);
			accVpHi = write_lane(accVpHi, (int)(uint32_t)(vp >> 32), n - 1);
			accVnLo = write_lane(accVnLo, (int)(uint32_t)vn, n - 1);
			accVnHi = write_lane(accVnHi, (int)(uint32_t)(vn >> 32), n - 1);
			// ---- the chunk leaves as four coalesced stores (VP, VN 8 B; before, packed end 4 B) ----
			const VU vpV = make_vu(accVpLo, accVpHi), vnV = make_vu(accVnLo, accVnHi);
			const VI endV = beforeV + vpopc(vpV) - vpopc(vnV);                   // scoreEnd = scoreBeforeStart + popcount(VP) - popcount(VN)
			const VI packedV = (endV << 3) | 4 | ((accVpHi >> 31) & 1) | (((accVnHi >> 31) & 1) << 1);     // (a bit-vector column's end cell always exists)
			store_lanes(rec.vp + outBase + w0, n, vpV);
			store_lanes(rec.vn + outBase + w0, n, vnV);
			store_lanes(rec.before + outBase + w0, n, beforeV);
			store_lanes(slot.end_cur + outBase + w0, n, packedV);
			lastEnd = read_lane(endV, n - 1);
			// minimum end score of the chunk and the LAST column attaining it (:1551-1559, 2410-2418)
			const VI endLive = select(live, endV, VI(INF));
			const int chunkMin = read_lane(prefix_min(endLive), LANES - 1);
			const uint64_t atMin = ballot(endLive == chunkMin);
			nodeMin = chunkMin < nodeMin ? chunkMin : nodeMin;
			if (chunkMin <= sliceMin) { sliceMin = chunkMin; minSlot = s; minOffset = w0 + (uint32_t)(63 - __builtin_clzll(atMin)); }
			// carries into the next chunk
			carryZero = read_lane(zeroV, n - 1);
			carryBefore = read_lane(beforeV, n - 1);
			carryAbove2 = read_lane(above2own, n - 1);
			carryExists = read_lane(existsV, n - 1) != 0;
			lastBefore = carryBefore; lastZero = carryZero; lastExists = carryExists;
		}
		if (lastBefore != lastZero) return GA_ASSERTION;                         // assert(newEnd.scoreBeforeStart == oldEnd.scoreBeforeStart) (:2385)
		if (GA_LANE0)
		{
			const int end = lastEnd;
			ws.cn_lastVP[s] = vp; ws.cn_lastVN[s] = vn; ws.cn_lastBefore[s] = lastBefore; ws.cn_lastExists[s] = lastExists ? 1 : 0;
			ws.cn_min[s] = nodeMin; ws.cn_lastEnd[s] = end; ws.cn_lastEnd2[s] = end - (int)(vp >> 63) + (int)(vn >> 63);
		}
	}
	return GA_OK;
}

// =================================================================================================
// Bands with cycles (GraphAligner.h:2362-2397).  Inside a strongly connected component a column
// depends on itself through the cycle, so the reference relaxes the component's nodes off a work
// stack and keeps, per column, how many of the 64 rows are already final ("confirmed"); a node's
// stored minimum and the argmin list are whatever its LAST visit saw, so the visiting order and
// the confirmation counts are part of the result.  This path therefore follows the reference's
// own bit-vector form step by step: wave-uniform scalar code over the VP/VN words (the 64-bit
// adds run on the scalar unit), with the cell-wise column merge done across the 64 lanes.  It is
// compiled into the wide kernel variants only; a slice takes it when its band has a cycle.
// =================================================================================================
struct GCol { uint64_t vp, vn; int before; int rows; bool partial, exists; };

GA_FN int gcol_end(const GCol& c) { return c.before + __builtin_popcountll(c.vp) - __builtin_popcountll(c.vn); }
GA_FN bool conf_less(int ar, bool ap, int br, bool bp) { return ar < br || (ar == br && !ap && bp); }          // WordSlice.h:150-153
GA_FN bool conf_greater(int ar, bool ap, int br, bool bp) { return ar > br || (ar == br && ap && !bp); }       // WordSlice.h:146-149
GA_FN int bit_of(uint64_t w, int i) { return (int)((w >> (i & 63)) & 1); }

// while a slice with cycles is being filled, end_cur[] holds rows | partial << 7 | exists << 8 per column
GA_FN GCol gcol_load(const SliceRec& rec, const uint32_t* meta, uint32_t idx)
{
	GCol c;
	c.vp = rec.vp[idx]; c.vn = rec.vn[idx]; c.before = rec.before[idx];
	const uint32_t m = meta[idx];
	c.rows = (int)(m & 127); c.partial = ((m >> 7) & 1) != 0; c.exists = ((m >> 8) & 1) != 0;
	return c;
}
GA_FN void gcol_store(const SliceRec& rec, uint32_t* meta, uint32_t idx, const GCol& c)
{
	if (GA_LANE0)
	{
		rec.vp[idx] = c.vp; rec.vn[idx] = c.vn; rec.before[idx] = c.before;
		meta[idx] = (uint32_t)c.rows | (c.partial ? 128u : 0u) | (c.exists ? 256u : 0u);
	}
}

// one column step with row confirmation (getNextSlice, GraphAligner.h:1349-1427).  aboveEnd / aboveEnd2 = rows
// j-1 / j-2 of the column to the left in the previous slice (only looked at when upLeftIn)
GA_FN GCol gcol_step(uint64_t eq, GCol c, bool upIn, bool upLeftIn, bool diagIn, bool aboveEq, int aboveEnd, int aboveEnd2, int& status)
{
	const int oldBefore = c.before;
	const int cr = c.rows;
	const uint64_t atConfirmed = 1ull << (cr & 63);            // the reference shifts by 64 when everything is confirmed: x86 wraps
	const uint64_t belowConfirmed = 1ull << ((cr - 1) & 63);
	bool oneMore = false;
	if (!c.exists || !diagIn) eq &= ~1ull;                      // :1358,1360
	c.exists = upIn;
	if (!upLeftIn) c.before += 1;
	else
	{
		if (c.before > aboveEnd) status = GA_ASSERTION;         // :1366
		const int viaDiagonal = aboveEnd2 + (aboveEq ? 0 : 1);
		c.before = c.before + 1 < viaDiagonal ? c.before + 1 : viaDiagonal;
	}
	const int hin = c.before - oldBefore;
	const uint64_t xv = eq | c.vn;
	if (hin < 0) eq |= 1;
	const uint64_t xh = (((eq & c.vp) + c.vp) ^ c.vp) | eq;
	uint64_t ph = c.vn | ~(xh | c.vp);
	uint64_t mh = c.vp & xh;
	int diagDiff = hin;
	if (cr > 0) diagDiff = ((ph & belowConfirmed) ? 1 : 0) - ((mh & belowConfirmed) ? 1 : 0);
	if (cr > 0 && (mh & belowConfirmed)) oneMore = true;
	else if (cr == 0 && hin == -1) oneMore = true;
	if (c.partial && (~ph & atConfirmed)) oneMore = true;
	ph <<= 1;
	mh <<= 1;
	if (hin < 0) mh |= 1; else if (hin > 0) ph |= 1;
	c.vp = mh | ~(xv | ph);
	c.vn = ph & xv;
	diagDiff += ((c.vp & atConfirmed) ? 1 : 0) - ((c.vn & atConfirmed) ? 1 : 0);
	if (diagDiff <= 0) oneMore = true;
	else if (c.vn & atConfirmed) oneMore = true;
	if (oneMore)
	{
		if (c.rows < W) c.rows += 1;
		c.partial = false;
	}
	else if (!c.partial && c.rows < W) c.partial = true;
	return c;
}

// index, in the sequence "VP bit of row i, then not-VN bit of row i" over rows [lo, hi), of the rank-th set
// unit (BitPosition over the interleaved words, WordSlice.h:45-130,479-488); past the end -> 128 + excess
GA_FN int interleaved_rank(uint64_t vp, uint64_t vn, int lo, int hi, int rank)
{
	int seen = 0;
	for (int i = lo; i < hi; i++)
	{
		if (bit_of(vp, i)) { if (seen == rank) return 2 * i; seen++; }
		if (!bit_of(vn, i)) { if (seen == rank) return 2 * i + 1; seen++; }
	}
	return 128 + (rank - seen);
}

// confirmed rows of the cell-wise minimum of two columns (WordSlice.h:423-510)
GA_FN void merged_confirmation(GCol l, GCol r, int& rows, bool& partial, int& status)
{
	if (l.rows == r.rows && l.partial == r.partial) { rows = l.rows; partial = l.partial; return; }
	if (conf_greater(r.rows, r.partial, l.rows, l.partial)) { GCol t = l; l = r; r = t; }
	const uint64_t low = r.rows >= W ? ~0ull : ~(~0ull << r.rows);
	int ls = l.before + __builtin_popcountll(l.vp & low) - __builtin_popcountll(l.vn & low);
	int rs = r.before + __builtin_popcountll(r.vp & low) - __builtin_popcountll(r.vn & low);
	if (r.rows == l.rows)
	{
		rs -= 1;
		if (!bit_of(l.vp, l.rows)) ls -= 1;
		rows = l.rows; partial = ls <= rs;
		return;
	}
	ls += bit_of(l.vp, r.rows) - bit_of(l.vn, r.rows);
	if (!(r.partial && bit_of(r.vp, r.rows))) rs -= 1;
	if (ls == rs + 1) { rows = r.rows; partial = true; return; }
	if (ls > rs + 1) { rows = r.rows; partial = r.partial; return; }
	if (l.rows > r.rows + 1)
	{
		if (ls > rs) status = GA_ASSERTION;
		const int lo = r.rows + 1, hi = l.rows;
		const int p = interleaved_rank(l.vp, l.vn, lo, hi, rs - ls);
		if (p / 2 < l.rows)
		{
			const int q = interleaved_rank(l.vp, l.vn, lo, hi, rs - ls + 1);
			rows = p / 2; partial = q / 2 > p / 2;
			return;
		}
		const uint64_t span = (hi >= W ? ~0ull : ~(~0ull << hi)) & (~0ull << lo);
		ls += __builtin_popcountll(l.vp & span) - __builtin_popcountll(l.vn & span);
		rs -= l.rows - r.rows - 1;
	}
	rows = l.rows; partial = l.partial;
	if (!l.partial) return;
	rs -= 1;
	if (bit_of(l.vp, l.rows) && !(ls <= rs)) partial = false;
}

// cell-wise minimum of two columns over rows j-1 .. j+63 (mergeTwoSlices, WordSlice.h:361-421): lanes = rows
GA_FN GCol gcol_merge(GCol a, GCol b, const VU& lowMask, int& status)
{
	if (a.before > b.before) { GCol t = a; a = b; b = t; }
	GCol out;
	merged_confirmation(a, b, out.rows, out.partial, status);
	const VI sa = vpopc(a.vp & lowMask) - vpopc(a.vn & lowMask) + a.before;
	const VI sb = vpopc(b.vp & lowMask) - vpopc(b.vn & lowMask) + b.before;
	const VI m = vmin(sa, sb);
	const VI up = shr1(m, a.before);
	out.vp = ballot(m == up + 1);
	out.vn = ballot(m == up - 1);
	out.before = a.before;
	out.exists = a.before < b.before ? a.exists : (a.exists || b.exists);    // :398-409
	return out;
}

// ---- Tarjan over the band subgraph with components (GraphAligner.h:1751-1901): post[] = emission order,
// comp[slot] = component number in emission order
template <int MAXN> GA_FN int scc_order(const GaDevGraph& g, WaveState<MAXN>& ws, int cn, int& nComps)
{
	for (int c = 0; c < cn; c += LANES) store_lanes(ws.color + c, cn - c, VI(0));
	wave_order();
	int16_t* index = ws.h_next;
	int16_t* tstack = ws.h_order;
	int emitted = 0, counter = 0, tsp = 0;
	nComps = 0;
	for (int root = 0; root < cn; root++)
	{
		if (ws.color[root] != 0) continue;
		int sp = 0;
		auto open = [&](int v) {
			ws.color[v] = 1;                                   // on the component stack
			index[v] = (int16_t)counter; ws.low[v] = (int16_t)counter; counter++;
			tstack[tsp++] = (int16_t)v;
			ws.st_slot[sp] = (int16_t)v; ws.st_cur[sp] = 0; sp++;
		};
		open(root);
		while (sp > 0)
		{
			const int v = ws.st_slot[sp - 1];
			const int cur = (int)ws.st_cur[sp - 1];
			if (cur < out_degree(g, ws, v))
			{
				const int x = out_slot(g, ws, cn, v, cur);
				if (x >= 0 && ws.color[x] == 0) { open(x); continue; }
				if (x >= 0 && ws.color[x] == 1 && index[x] < ws.low[v]) ws.low[v] = index[x];
				ws.st_cur[sp - 1] = (uint32_t)(cur + 1);
				continue;
			}
			sp--;
			if (ws.low[v] == index[v])
			{
				int back;
				do
				{
					back = tstack[--tsp];
					ws.color[back] = 2;
					ws.comp[back] = (int16_t)nComps;
					ws.post[emitted++] = (int16_t)back;
				} while (back != v);
				nComps++;
			}
			if (sp > 0)
			{
				const int parent = ws.st_slot[sp - 1];
				if (ws.low[v] < ws.low[parent]) ws.low[parent] = ws.low[v];
				ws.st_cur[sp - 1] += 1;
			}
		}
	}
	return GA_OK;
}

// ---- exact row j-1 of one component (forceComponentZeroRow, GraphAligner.h:1903-1995): shortest paths from the
// columns fed from outside the component; lanes = columns along a node
template <int MAXN>
GA_FN int zero_row_component(const GaDevGraph& g, WaveState<MAXN>& ws, const Slot& slot, const SliceRec& rec, int pn, int cn, int lo, int hi, int ci)
{
	const VI lane = lane_iota();
	int heapSize = 0;
	for (int k = lo; k < hi; k++)
	{
		const int s = ws.post[k];
		const uint32_t len = ws.cn_len[s];
		const int ps = ws.cn_prev[s];
		const bool inPrev = ps >= 0;
		const uint32_t* pend = slot.end_prev + (inPrev ? ws.pn_colBase[ps] : 0);
		const uint32_t outBase = ws.cn_colBase[s];
		int zero0 = inPrev ? (int)(pend[0] >> 3) : INF;
		const int inDeg = in_degree(g, ws, s);
		for (int e = 0; e < inDeg; e++)
		{
			int cs, pm;
			in_slots(g, ws, pn, cn, s, e, cs, pm);
			if (cs < 0 && pm < 0) continue;
			if (cs >= 0 && ws.comp[cs] == ci) continue;
			if (cs >= 0) zero0 = zero0 < ws.cn_lastBefore[cs] + 1 ? zero0 : ws.cn_lastBefore[cs] + 1;
			if (pm >= 0) zero0 = zero0 < ws.pn_lastEnd[pm] + 1 ? zero0 : ws.pn_lastEnd[pm] + 1;
		}
		int carry = INF;
		for (uint32_t w0 = 0; w0 < len; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			VI zeroV = VI(INF);
			if (zero0 != INF)
			{
				const VB live = lane < n;
				const VI pendV = inPrev ? select(live, load_lanes(pend + w0, n, 0) >> 3, VI(INF)) : VI(INF);
				const VI zin = w0 == 0 ? select(lane == 0, VI(zero0), pendV) : pendV;
				zeroV = vmin(prefix_min(zin - lane) + lane, w0 == 0 ? VI(INF) : lane + (carry + 1));
				carry = read_lane(zeroV, n - 1);
			}
			store_lanes(rec.before + outBase + w0, n, zeroV);
		}
		if (zero0 == INF) continue;
		const int outDeg = out_degree(g, ws, s);
		for (int e = 0; e < outDeg; e++)
		{
			const int x = out_slot(g, ws, cn, s, e);
			if (x < 0 || ws.comp[x] != ci) continue;
			if (!heap_push(ws, heapSize, (uint32_t)x, carry + 1)) return GA_CAP_HEAP;
		}
	}
	wave_sync();
	while (heapSize > 0)
	{
		const int s = (int)ws.heap_node[0];
		const int score = ws.heap_prio[0];
		heap_pop(ws, heapSize);
		const uint32_t len = ws.cn_len[s];
		const uint32_t outBase = ws.cn_colBase[s];
		bool reachedEnd = true;
		for (uint32_t w0 = 0; w0 < len && reachedEnd; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			const VI have = load_lanes(rec.before + outBase + w0, n, 0);
			const VI cand = lane + (score + (int)w0);
			const uint64_t stop = ballot((lane < n) && (have < cand + 1));
			const int upto = stop ? ctz64(stop) : n;
			store_lanes(rec.before + outBase + w0, upto, cand);
			if (stop) reachedEnd = false;
		}
		wave_sync();
		if (!reachedEnd) continue;
		const int outDeg = out_degree(g, ws, s);
		for (int e = 0; e < outDeg; e++)
		{
			const int x = out_slot(g, ws, cn, s, e);
			if (x < 0 || ws.comp[x] != ci) continue;
			if (!heap_push(ws, heapSize, (uint32_t)x, score + (int)len)) return GA_CAP_HEAP;
		}
	}
	// every column restarts as an unconfirmed vertical run under its exact row j-1 score (:1975-1993)
	for (int k = lo; k < hi; k++)
	{
		const int s = ws.post[k];
		const uint32_t len = ws.cn_len[s];
		const int ps = ws.cn_prev[s];
		const bool inPrev = ps >= 0;
		const uint32_t* pend = slot.end_prev + (inPrev ? ws.pn_colBase[ps] : 0);
		const uint32_t outBase = ws.cn_colBase[s];
		int last = INF;
		for (uint32_t w0 = 0; w0 < len; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			const VI b = load_lanes(rec.before + outBase + w0, n, 0);
			if (ballot((lane < n) && (b == INF))) return GA_ASSERTION;
			const VI pendRaw = inPrev ? load_lanes(pend + w0, n, 0) : VI(-8);
			const VI pendV = pendRaw >> 3;
			store_lanes(rec.vp + outBase + w0, n, VU(~0ull));
			store_lanes(rec.vn + outBase + w0, n, VU(0ull));
			store_lanes(slot.end_cur + outBase + w0, n, select((pendV == b) && ((pendRaw & 4) != 0), VI(256), VI(0)));      // scoreBeforeExists (:1989)
			last = read_lane(b, n - 1);
		}
		if (GA_LANE0) ws.cn_lastBefore[s] = last;
	}
	wave_sync();
	return GA_OK;
}

// ---- one visit of one node (calculateNode, GraphAligner.h:1457-1573) -----------------------------------------------
// callMin / callLast: minimum scoreEnd over the fully confirmed columns this visit touched and the last such column
template <int MAXN>
GA_FN int fill_node_general(const GaDevGraph& g, WaveState<MAXN>& ws, const Slot& slot, const SliceRec& rec, const uint64_t* eqOf, const VI& rowCode,
                            int rawAbove, uint32_t nRows, uint32_t j, int prevMin, int pn, int cn, int s, int& callMin, uint32_t& callLast)
{
	const VU lowMask = low_mask_through_lane();
	int status = GA_OK;
	callMin = INF;
	callLast = 0;
	const uint32_t len = ws.cn_len[s];
	const uint64_t firstCol = ((uint64_t)ws.cn_startHi[s] << 32) | ws.cn_startLo[s];
	const int ps = ws.cn_prev[s];
	const bool inPrev = ps >= 0;
	const uint32_t* pend = slot.end_prev + (inPrev ? ws.pn_colBase[ps] : 0);
	const uint32_t outBase = ws.cn_colBase[s];
	uint32_t* meta = slot.end_cur;
	const bool aboveAlways = j == 0 && inPrev;
	auto aboveEqAt = [&](int base) { return aboveAlways || (j > 0 && rawAbove == base); };
	auto note = [&](const GCol& c, uint32_t w) {
		if (c.rows != W) return;
		const int end = gcol_end(c);
		if (end < callMin) { callMin = end; callLast = w; }
		else if (end == callMin) callLast = w;
	};
	auto verticalEntry = [&](GCol& c, uint32_t w) {
		// re-entry from the cell above when that beats what came from the left (:1504-1509, 1541-1546)
		if (!inPrev) return;
		const int oEnd = (int)(pend[w] >> 3);
		if (c.before > oEnd)
		{
			GCol src;
			src.vp = ~0ull; src.vn = 0; src.before = oEnd; src.rows = W; src.partial = false; src.exists = (pend[w] & 4) != 0;
			c = gcol_merge(c, src, lowMask, status);
		}
	};
	auto sanity = [&](const GCol& c, uint32_t w) {
		// assertSliceCorrectness (:1437-1455)
		const int end = gcol_end(c);
		if (c.before < 0 || end < 0 || (c.vp & c.vn) != 0) status = GA_ASSERTION;
		if (inPrev && c.before > (int)(pend[w] >> 3)) status = GA_ASSERTION;
		if (c.rows == W && (end < prevMin || c.before < prevMin)) status = GA_ASSERTION;
	};

	GCol mine = gcol_load(rec, meta, outBase);
	const int rows0 = mine.rows;
	const bool partial0 = mine.partial;
	if (rows0 == W) return GA_OK;
	const int base0 = g_base(g, firstCol);
	const int inDeg = in_degree(g, ws, s);
	bool source = true;
	for (int e = 0; e < inDeg; e++)
	{
		int cs, pm;
		in_slots(g, ws, pn, cn, s, e, cs, pm);
		if (cs >= 0 || pm >= 0) { source = false; break; }
	}
	GCol c;
	if (source)
	{
		// :1317-1337
		c.vn = 0; c.rows = W; c.partial = false;
		if (j == 0 && inPrev)
		{
			const uint64_t firstVp = (eqOf[base0] & 1) ? 0 : 1;
			c.vp = (~0ull & ~1ull) | firstVp; c.before = (int)(pend[0] >> 3); c.exists = true;
		}
		else if (inPrev) { c.vp = ~0ull; c.before = (int)(pend[0] >> 3); c.exists = (pend[0] & 4) != 0; }
		else { c.vp = ~0ull & ~1ull; c.before = (int)(nRows + 1); c.exists = false; }
	}
	else
	{
		// first column from the in-neighbours' last columns (:1270-1315)
		bool any = false;
		const bool aboveEq0 = aboveEqAt(base0);
		for (int e = 0; e < inDeg; e++)
		{
			int cs, pm;
			in_slots(g, ws, pn, cn, s, e, cs, pm);
			if (cs < 0 && pm < 0) continue;
			uint64_t eqHere = eqOf[base0];
			const bool haveAbove = pm >= 0;
			GCol left;
			if (cs >= 0) left = gcol_load(rec, meta, ws.cn_colBase[cs] + ws.cn_len[cs] - 1);
			else
			{
				left.vp = ~0ull; left.vn = 0; left.before = ws.pn_lastEnd[pm]; left.rows = W; left.partial = false; left.exists = true;
				eqHere &= 1;                                                         // :1294-1301
			}
			const GCol here = gcol_step(eqHere, left, mine.exists, mine.exists && haveAbove, haveAbove, aboveEq0,
			                            haveAbove ? ws.pn_lastEnd[pm] : 0, haveAbove ? ws.pn_lastEnd2[pm] : 0, status);
			if (!any) { c = here; any = true; }
			else c = gcol_merge(c, here, lowMask, status);
		}
		if (!any) return GA_ASSERTION;
		verticalEntry(c, 0);
	}
	note(c, 0);
	sanity(c, 0);
	gcol_store(rec, meta, outBase, c);
	if (conf_less(c.rows, c.partial, rows0, partial0)) status = GA_ASSERTION;
	if (status != GA_OK) return status;
	if (c.rows == rows0 && c.partial == partial0) { wave_sync(); return GA_OK; }
	for (uint32_t w = 1; w < len; w++)
	{
		const GCol was = gcol_load(rec, meta, outBase + w);
		if (was.rows == W) break;
		const int base = g_base(g, firstCol + w);
		const bool e = was.exists;
		const int leftBefore = c.before;
		int aboveEnd = 0, aboveEnd2 = 0;
		if (inPrev) { const uint32_t pr = pend[w - 1]; aboveEnd = (int)(pr >> 3); aboveEnd2 = aboveEnd - (int)(pr & 1) + (int)((pr >> 1) & 1); }
		c = gcol_step(eqOf[base], c, e, e, c.exists, aboveEqAt(base), aboveEnd, aboveEnd2, status);
		verticalEntry(c, w);
		if (!(inPrev || c.before == (int)j || c.before == leftBefore + 1)) status = GA_ASSERTION;    // :1548
		sanity(c, w);
		note(c, w);
		gcol_store(rec, meta, outBase + w, c);
		if (status != GA_OK) return status;
		if (c.rows == was.rows && c.partial == was.partial) break;
	}
	(void)rowCode;
	wave_sync();
	return status;
}

// ---- one slice whose band has a cycle (calculateSlice, GraphAligner.h:2331-2451) ---------------------------------
template <int MAXN>
GA_FN int fill_slice_general(const GaDevGraph& g, WaveState<MAXN>& ws, const Slot& slot, const SliceRec& rec, const VI rowCode, const int rowAboveCode,
                             uint32_t nRows, uint32_t j, int prevMin, int pn, int cn, int nComps, int& sliceMin, int& minSlot, uint32_t& minOffset)
{
	const VI lane = lane_iota();
	if (ballot((rowCode & GA_ROW_INVALID) != 0)) return GA_ASSERTION;
	const int rawAbove = j > 0 ? (rowAboveCode >> 4) & 7 : 7;
	uint64_t eqOf[4];
	for (int b = 0; b < 4; b++) eqOf[b] = ballot(bit_extract(rowCode, b) != 0);
	sliceMin = INF;
	minSlot = -1;
	minOffset = 0;
	int hi = cn;
	for (int ci = nComps - 1; ci >= 0; ci--)
	{
		int lo = hi;
		while (lo > 0 && ws.comp[ws.post[lo - 1]] == ci) lo--;
		int status = zero_row_component(g, ws, slot, rec, pn, cn, lo, hi, ci);
		if (status != GA_OK) return status;
		// the work stack (UniqueQueue): color[] = "is on the stack"
		int sp = 0;
		for (int k = lo; k < hi; k++) { const int s = ws.post[k]; ws.st_slot[sp++] = (int16_t)s; ws.color[s] = 1; }
		while (sp > 0)
		{
			const int s = ws.st_slot[--sp];
			ws.color[s] = 0;
			const uint32_t lastIdx = ws.cn_colBase[s] + ws.cn_len[s] - 1;
			const GCol oldEnd = gcol_load(rec, slot.end_cur, lastIdx);
			int callMin;
			uint32_t callLast;
			status = fill_node_general(g, ws, slot, rec, eqOf, rowCode, rawAbove, nRows, j, prevMin, pn, cn, s, callMin, callLast);
			if (status != GA_OK) return status;
			if (GA_LANE0) ws.cn_min[s] = callMin;
			const GCol newEnd = gcol_load(rec, slot.end_cur, lastIdx);
			if (newEnd.before != oldEnd.before) return GA_ASSERTION;                                       // :2385
			if (conf_less(newEnd.rows, newEnd.partial, oldEnd.rows, oldEnd.partial)) return GA_ASSERTION;  // :2386
			if (newEnd.before < (int)nRows && conf_greater(newEnd.rows, newEnd.partial, oldEnd.rows, oldEnd.partial))
			{
				const int outDeg = out_degree(g, ws, s);
				for (int e = 0; e < outDeg; e++)
				{
					const int x = out_slot(g, ws, cn, s, e);
					if (x < 0 || ws.comp[x] != ci || ws.color[x] != 0) continue;
					if ((slot.end_cur[ws.cn_colBase[x]] & 127) < (uint32_t)W) { ws.st_slot[sp++] = (int16_t)x; ws.color[x] = 1; }
				}
			}
			if (callMin != INF && callMin <= sliceMin) { sliceMin = callMin; minSlot = s; minOffset = callLast; }
		}
		for (int k = lo; k < hi; k++)
			if ((slot.end_cur[ws.cn_colBase[ws.post[k]]] & 127) != (uint32_t)W) return GA_ASSERTION;      // :2422-2425
		hi = lo;
	}
	// ---- leave the slice in the form the next slice and the traceback read ----
	for (int s = 0; s < cn; s++)
	{
		const uint32_t len = ws.cn_len[s];
		const uint32_t outBase = ws.cn_colBase[s];
		uint64_t vp = 0, vn = 0;
		int before = 0, end = 0;
		for (uint32_t w0 = 0; w0 < len; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			const VU vpV = load_lanes_u64(rec.vp + outBase + w0, n), vnV = load_lanes_u64(rec.vn + outBase + w0, n);
			const VI beforeV = load_lanes(rec.before + outBase + w0, n, 0);
			const VI endV = beforeV + vpopc(vpV) - vpopc(vnV);
			const VI packedV = (endV << 3) | 4 | vpopc(vpV & VU(1ull << 63)) | (vpopc(vnV & VU(1ull << 63)) << 1);
			store_lanes(slot.end_cur + outBase + w0, n, packedV);
			vp = read_lane(vpV, n - 1); vn = read_lane(vnV, n - 1);
			before = read_lane(beforeV, n - 1); end = read_lane(endV, n - 1);
		}
		if (GA_LANE0)
		{
			ws.cn_lastVP[s] = vp; ws.cn_lastVN[s] = vn; ws.cn_lastBefore[s] = before; ws.cn_lastExists[s] = 0;
			ws.cn_lastEnd[s] = end; ws.cn_lastEnd2[s] = end - (int)(vp >> 63) + (int)(vn >> 63);
		}
	}
	(void)lane;
	return GA_OK;
}

// ---- cell value from the stored words (WordSlice.h:223-229; getValueOrMax GraphAligner.h:2008-2017) ------
// the band node list of the slice being traced is kept in LDS (tabNodes / tabBase), so finding a node costs no memory round trip
GA_FN int stored_value(const SliceRec& r, const uint32_t* tabNodes, const uint32_t* tabBase, uint32_t nNodes, uint32_t node, uint32_t offset, int rowInSlice, int big)
{
	int slot = find_slot(tabNodes, (int)nNodes, node);
	if (slot < 0) return big;
	uint32_t idx = tabBase[slot] + offset;
	uint64_t vp = r.vp[idx], vn = r.vn[idx];
	uint64_t mask = rowInSlice < 63 ? ~(~0ull << (rowInSlice + 1)) : ~0ull;
	return r.before[idx] + __builtin_popcountll(vp & mask) - __builtin_popcountll(vn & mask);
}

}  // namespace gak
#include "ga_sparse.h"
namespace gak {

// ---- the whole job --------------------------------------------------------------------------------------------
// Slice records carry everything needed to continue from them: the narrow variant only ever continues from
// the slice it has just finished; the wide variants also go back to an earlier one (the ramp redo of
// getSqrtSlices, GraphAligner.h:2648-2719) and re-run windows from checkpoints (getSlicesFromTable, :2858-2943)
// when a redo has left the reference's checkpoint list inconsistent with the slices it finally kept.
constexpr uint32_t kSeedRecord = 0xffffffffu;      // "record" of the initial slice (the seed node at score 0, j = -64)

template <int MAXN, bool GENERAL, bool SPARSE = false>
GA_FN void run_job(const GaLaunch& L, WaveState<MAXN>& ws, const Slot& slotIn, uint32_t jobIndex)
{
	constexpr bool kWide = GENERAL;             // cycles and ramp redos: compiled into the general variants only
	constexpr bool kSparse = SPARSE && GENERAL; // bands of >= 200 000 cells (sparse method, backtrace override): the last variant of the ladder only
	const GaDevGraph& g = L.graph;
	const GaJob job = L.jobs[jobIndex];
	const GaHmmTables& hmm = *L.hmm;
	const uint8_t* rows = L.rows + job.rows_off;
	Slot slot = slotIn;
	GaJobOut out;
	out.status = GA_OK; out.score = 0x7fffffff; out.n_valid = 0; out.n_run = 0; out.trace_len = 0; out.max_band_nodes = 0; out.n_columns = 0; out.trace_off = 0;
	out.start_node = 0; out.start_offset = 0; out.start_row = 0; out.reserved2 = 0; out.n_node_steps = 0; out.reserved3 = 0;
	for (int i = 0; i < 8; i++) out.stamps[i] = 0;
	uint64_t tA = stamp(), tB;
#define GA_LAP(i) do { tB = stamp(); out.stamps[i] += tB - tA; tA = tB; } while (0)
	const uint32_t numSlices = job.n_rows / W;
	int status = GA_OK;

	// ---- the state a slice is computed from: previous band tables in LDS, packed end scores in end_prev ----
	int pn = 0;
	int prevMin = 0;
	double logCorrect = hmm.init_correct, logWrong = hmm.init_wrong;
	const uint32_t seedLen = g_len(g, job.seed_node);
	auto loadSeedState = [&]() {
		// initial slice: the whole seed node at score 0 (GraphAligner.h:2945-2960)
		wave_sync();
		if (GA_LANE0)
		{
			ws.pn_node[0] = job.seed_node; ws.pn_min[0] = 0; ws.pn_lastEnd[0] = 0; ws.pn_lastEnd2[0] = 0; ws.pn_colBase[0] = 0;
			ws.pn_len[0] = seedLen;
			ws.pn_outDeg[0] = 255;          // the seed node's out-list is read from HBM once
		}
		for (uint32_t c = 0; c < seedLen; c += LANES) store_lanes(slot.end_prev + c, (int)(seedLen - c), VI(4));      // score 0, scoreEndExists
		pn = 1; prevMin = 0; logCorrect = hmm.init_correct; logWrong = hmm.init_wrong;
		wave_sync();
	};
	auto loadRecordState = [&](uint32_t off) {
		// continue from a stored slice: what the reference keeps of it is its frozen end scores (NodeSlice.h:353-376)
		wave_sync();
		const uint32_t nN = slot.arena[off], nC = slot.arena[off + 1];
		const SliceRec r = slice_at(slot.arena, off, nN, nC);
		for (uint32_t q = 0; q < nN; q++)
		{
			const uint32_t base = r.colBase[q];
			const uint32_t next = q + 1 < nN ? r.colBase[q + 1] : nC;
			const uint32_t idx = next - 1;
			const uint64_t vp = r.vp[idx], vn = r.vn[idx];
			const int e = r.before[idx] + __builtin_popcountll(vp) - __builtin_popcountll(vn);
			if (GA_LANE0)
			{
				ws.pn_node[q] = r.nodes[q]; ws.pn_colBase[q] = base; ws.pn_len[q] = next - base; ws.pn_min[q] = r.nodeMin[q];
				ws.pn_lastEnd[q] = e; ws.pn_lastEnd2[q] = e - (int)(vp >> 63) + (int)(vn >> 63);
				ws.pn_outDeg[q] = 255;      // out-lists come from HBM
			}
		}
		for (uint32_t c = 0; c < nC; c += LANES)
		{
			const int k = (int)(nC - c);
			const VU vpV = load_lanes_u64(r.vp + c, k), vnV = load_lanes_u64(r.vn + c, k);
			const VI endV = load_lanes(r.before + c, k, 0) + vpopc(vpV) - vpopc(vnV);
			store_lanes(slot.end_prev + c, k, (endV << 3) | 4 | vpopc(vpV & VU(1ull << 63)) | (vpopc(vnV & VU(1ull << 63)) << 1));
		}
		pn = (int)nN; prevMin = (int)r.hdr[2];
		union { double d; uint32_t w[2]; } a, b;
		a.w[0] = r.hdr[6]; a.w[1] = r.hdr[7]; b.w[0] = r.hdr[8]; b.w[1] = r.hdr[9];
		logCorrect = a.d; logWrong = b.d;
		wave_sync();
	};

	// a seed node of >= 200 000 bp is the whole band of the second slice: the reference goes sparse there (GraphAligner.h:2483)
	// (with the sparse method compiled in: slice 0 of such a seed runs at the ramp width, :2612; without one that is a bandwidth of 0 and
	// undefined behaviour in the reference, reported as an assertion; with one, every column of the seed node is a cell of row 0 --
	// more than this program's tables hold, reported as a capacity miss)
	if (seedLen >= kCutoff) status = kSparse ? (L.ramp_bw < 1 ? GA_ASSERTION : GA_CAP_COLS) : GA_UNSUPPORTED_BAND;
	else if (seedLen > L.cap_cols) status = GA_CAP_COLS;
	if (status == GA_OK) loadSeedState();
	VI rowNext = load_lanes(rows, W, 0);
	uint32_t rowNextSlice = 0;
	int rowAboveKept = 0;                      // code of the last row of slice rowAboveFor - 1
	uint32_t rowAboveFor = 0;
	uint64_t arenaTop = 0;
	uint32_t nPushed = 0;          // bandwidthPerSlice.size()
	uint32_t nRun = 0;
	const bool rampPossible = L.ramp_bw > L.initial_bw;

	// ---- one slice from the loaded state: band, order, fill; the record is written at arenaTop (not yet claimed) ----
	int cn = 0, sliceMin = 0;
	uint64_t need = 0;
	// what the slice just computed was: its cells (DPSlice::numCells) and, for a slice of the sparse method, what its frozen forms could not hold
	uint32_t freshCells = 0;
	bool freshSparse = false, freshEndsTooFar = false, freshBeforeTooFar = false;
	auto runSlice = [&](uint32_t slice, int bandwidth) -> int {
		uint32_t totalCols = 0;
		cn = 0;
		freshSparse = false; freshEndsTooFar = false; freshBeforeTooFar = false;
		GA_LAP(0);
		int st = project_band(g, ws, pn, prevMin, bandwidth, cn, totalCols);
		GA_LAP(1);
		if constexpr (kSparse)
		{
			if (st == GA_UNSUPPORTED_BAND)
			{
				// ---- the band has 200 000 cells or more: the sparse method (pickMethodAndExtendFill, :2499-2520) ----
				wave_sync();
				if ((uint32_t)bandwidth > slot.sparse_max_bw) return GA_CAP_HEAP;
				const SparseMem sm = sparse_mem_at(slot.sparse, slot.sparse_max_bw);
				const SparseResult sr = sparse_fill(g, ws, slot, sm, rows + (uint64_t)slice * W, job.n_rows, slice * W, pn, prevMin, bandwidth, cn);
				rowNextSlice = 0xffffffffu;                                           // (the row codes were not taken from the prefetch)
				if (sr.status != GA_OK) return sr.status;
				if (sr.oddWord) return GA_CAP_COLS;                                   // (see SparseResult::oddWord: reported, never a different answer)
				if (sr.minScore < prevMin) return GA_ASSERTION;                      // :2508
				sliceMin = sr.minScore;
				freshSparse = true; freshEndsTooFar = sr.endsTooFar; freshBeforeTooFar = sr.beforeTooFar;
				freshCells = sr.numCells;
				out.n_columns += sr.numCells;
				out.max_band_nodes = out.max_band_nodes > (uint32_t)cn ? out.max_band_nodes : (uint32_t)cn;
				need = 0;
				if (sr.endsTooFar) return GA_OK;                                      // the slice cannot outlive this iteration (see the loop): no record
				if (sr.numCells > L.cap_cols) return GA_CAP_COLS;
				need = slice_words((uint32_t)cn, sr.numCells) + (sr.numCells + 3) / 4;
				if (arenaTop + need > L.arena_words || arenaTop + need >= 0xffffffffull) return GA_CAP_ARENA;
				SliceRec rec = slice_at(slot.arena, arenaTop, (uint32_t)cn, sr.numCells);
				sparse_materialize(g, ws, slot, sm, rec, (uint8_t*)(rec.before + sr.numCells), cn, sr.nWords, bandwidth);
				if (GA_LANE0)
				{
					rec.hdr[0] = (uint32_t)cn; rec.hdr[1] = sr.numCells; rec.hdr[2] = (uint32_t)sliceMin; rec.hdr[3] = (uint32_t)sr.minSlot; rec.hdr[4] = sr.minOffset; rec.hdr[5] = 0;
					rec.hdr[11] = 1u | (sr.beforeTooFar ? 2u : 0u);
				}
				return GA_OK;
			}
		}
		if (st != GA_OK) return st;
		freshCells = totalCols;
		if (totalCols > L.cap_cols) return GA_CAP_COLS;
		wave_order();
		load_topology(g, ws, pn, cn);
		wave_order();
		// this slice's row codes normally arrived during the previous slice; request the next slice's now
		if (rowNextSlice != slice) rowNext = load_lanes(rows + (uint64_t)slice * W, W, 0);
		const VI rowCode = rowNext;
		const int rowAboveCode = rowAboveFor == slice ? rowAboveKept : (slice > 0 ? (int)rows[(uint64_t)slice * W - 1] : 0);
		rowAboveKept = read_lane(rowCode, W - 1);
		rowAboveFor = slice + 1;
		if (slice + 1 < numSlices) { rowNext = load_lanes(rows + (uint64_t)(slice + 1) * W, W, 0); rowNextSlice = slice + 1; }
		GA_LAP(2);
		st = processing_order(g, ws, cn);
		int nComps = 0;
		bool cyclicBand = false;
		if constexpr (kWide)
		{
			// a band with a cycle is filled by the confirmation-tracking path (wide kernel variants only; the
			// narrow variant reports GA_UNSUPPORTED_CYCLE and the job is rerun by a wide one)
			if (st == GA_UNSUPPORTED_CYCLE) { wave_order(); st = scc_order(g, ws, cn, nComps); cyclicBand = true; }
		}
		GA_LAP(3);
		if (st != GA_OK) return st;
		need = slice_words((uint32_t)cn, totalCols);
		if (arenaTop + need > L.arena_words) return GA_CAP_ARENA;
		if (arenaTop + need >= 0xffffffffull) return GA_CAP_ARENA;
		SliceRec rec = slice_at(slot.arena, arenaTop, (uint32_t)cn, totalCols);
		for (int c = 0; c < cn; c += LANES)
		{
			store_lanes(rec.nodes + c, cn - c, load_lanes(ws.cn_node + c, cn - c, 0));
			store_lanes(rec.colBase + c, cn - c, load_lanes(ws.cn_colBase + c, cn - c, 0));
		}
		wave_order();
		int minSlot;
		uint32_t minOffset;
		GA_LAP(0);
		if constexpr (kWide)
		{
			if (cyclicBand) st = fill_slice_general(g, ws, slot, rec, rowCode, rowAboveCode, job.n_rows, slice * W, prevMin, pn, cn, nComps, sliceMin, minSlot, minOffset);
			else st = fill_slice(g, ws, slot, rec, rowCode, rowAboveCode, job.n_rows, slice * W, pn, cn, sliceMin, minSlot, minOffset);
		}
		else st = fill_slice(g, ws, slot, rec, rowCode, rowAboveCode, job.n_rows, slice * W, pn, cn, sliceMin, minSlot, minOffset);
		GA_LAP(4);
		if (st != GA_OK) return st;
		if (sliceMin < prevMin) return GA_ASSERTION;                              // :2469
		wave_order();
		for (int c = 0; c < cn; c += LANES) store_lanes(rec.nodeMin + c, cn - c, load_lanes(ws.cn_min + c, cn - c, 0));
		if (GA_LANE0)
		{
			rec.hdr[0] = (uint32_t)cn; rec.hdr[1] = totalCols; rec.hdr[2] = (uint32_t)sliceMin; rec.hdr[3] = (uint32_t)minSlot; rec.hdr[4] = minOffset; rec.hdr[5] = 0;
			rec.hdr[11] = 0;
		}
		out.n_columns += totalCols;
		out.max_band_nodes = out.max_band_nodes > (uint32_t)cn ? out.max_band_nodes : (uint32_t)cn;
		return GA_OK;
	};
	// the slice just computed becomes the state
	auto adoptSlice = [&]() {
		wave_order();
		for (int c = 0; c < cn; c += LANES)
		{
			int k = cn - c;
			store_lanes(ws.pn_node + c, k, load_lanes(ws.cn_node + c, k, 0));
			store_lanes(ws.pn_min + c, k, load_lanes(ws.cn_min + c, k, 0));
			store_lanes(ws.pn_lastEnd + c, k, load_lanes(ws.cn_lastEnd + c, k, 0));
			store_lanes(ws.pn_lastEnd2 + c, k, load_lanes(ws.cn_lastEnd2 + c, k, 0));
			store_lanes(ws.pn_colBase + c, k, load_lanes(ws.cn_colBase + c, k, 0));
			store_lanes(ws.pn_len + c, k, load_lanes(ws.cn_len + c, k, 0));
			store_lanes(ws.pn_outDeg + c, k, load_lanes(ws.cn_outDeg + c, k, 0));
		}
		for (int c = 0; c < cn * kNbr; c += LANES) store_lanes(ws.pn_outNbr + c, cn * kNbr - c, load_lanes(ws.cn_outNbr + c, cn * kNbr - c, 0));
		pn = cn;
		prevMin = sliceMin;
		uint32_t* t = slot.end_prev; slot.end_prev = slot.end_cur; slot.end_cur = t;
		wave_sync();
	};

	// ---- first pass (getSqrtSlices, GraphAligner.h:2571-2856) ----
	// checkpoint bookkeeping of the reference, needed only to reproduce what a ramp redo does to it
	uint32_t sampling = 0;
	while ((sampling + 1) * (sampling + 1) <= numSlices) sampling++;            // (int)sqrt(len / 64) (:2962-2967)
	uint32_t lastRec = kSeedRecord, rampRec = kSeedRecord, storeRec = kSeedRecord;
	uint32_t storeMem = 28, rampUntil = 0, rampRedoIndex = 0xffffffffu, nCkpt = 0;
	bool redone = false;
	auto recSlice = [&](uint32_t recOff) -> uint32_t { return recOff == kSeedRecord ? 0xffffffffu : slot.arena[recOff + 10]; };   // slice index of a record (-1 for the seed)
	// DPSlice::j as the reference compares it: unsigned, the seed slice's -64 is the largest value there is
	auto recJ = [&](uint32_t recOff) -> uint64_t { return recOff == kSeedRecord ? ~0ull - 63 : (uint64_t)slot.arena[recOff + 10] * W; };
	// backtrace-override bookkeeping of getSqrtSlices (:2604-2606, 2721-2764, 2810-2825): the open window = slices ovFirst .. ovFirst + ovCount - 1
	uint32_t lastNumCells = 0, ovPreRec = kSeedRecord, ovFirst = 0, ovCount = 0, nOv = 0;
	bool overriding = false, anyOverride = false;
	(void)recJ; (void)ovPreRec; (void)ovFirst; (void)ovCount; (void)nOv; (void)overriding; (void)anyOverride; (void)lastNumCells;
	// what building the override of the open window checks (ga_sparse.h), then the window joins the list and the checkpoints inside it go
	auto closeWindow = [&]() -> int {
		if constexpr (kSparse)
		{
			if (ovCount == 0 || lastRec == kSeedRecord || recSlice(lastRec) != ovFirst + ovCount - 1) return GA_ASSERTION;   // assert(lastSlice.j == backtraceOverrideTemps.back().j)
			wave_sync();
			const SparseMem sm = sparse_mem_at(slot.sparse, slot.sparse_max_bw);
			const int st = explore_override(g, ws, slot, sm, slot.slice_off, ovFirst, ovCount, ovPreRec == kSeedRecord ? 0u : ovPreRec, ovPreRec == kSeedRecord,
			                                job.seed_node, rows, (int)job.n_rows);
			if (st != GA_OK) return st;
			if (GA_LANE0) { slot.ovr[2 * nOv] = ovFirst; slot.ovr[2 * nOv + 1] = ovFirst + ovCount - 1; }
			nOv++;
			anyOverride = true;
			overriding = false;
			const uint64_t startj = (uint64_t)ovFirst * W, endj = (uint64_t)(ovFirst + ovCount - 1) * W;
			while (nCkpt > 0 && recJ(slot.ckpt[nCkpt - 1]) >= startj && recJ(slot.ckpt[nCkpt - 1]) <= endj) nCkpt--;
			ovCount = 0;
			wave_sync();
		}
		return GA_OK;
	};
	for (uint32_t slice = 0; slice < numSlices && status == GA_OK; slice++)
	{
		// slice 0 always runs at the ramp width because rampUntil(0) >= slice(0) (:2603,2612)
		const bool useRamp = rampUntil >= slice;
		const int bandwidth = useRamp ? L.ramp_bw : L.initial_bw;
		status = runSlice(slice, bandwidth);
		if (status != GA_OK) break;
		const uint32_t thisRec = (uint32_t)arenaTop;
		nRun++;
		// ---- HMM step (AlignmentCorrectnessEstimation.cpp:71-89): additions and comparisons only ----
		int mism = sliceMin - prevMin;
		if (mism > 64) { status = GA_ASSERTION; break; }
		double cc = logCorrect + hmm.c2c, fc = logWrong + hmm.f2c, cf = logCorrect + hmm.c2f, ff = logWrong + hmm.f2f;
		bool correctFromCorrect = cc >= fc;
		bool falseFromCorrect = cf >= ff;
		const double newCorrect = (cc > fc ? cc : fc) + hmm.correct_mult[mism];
		const double newWrong = (cf > ff ? cf : ff) + hmm.wrong_mult[mism];
		bool currentlyCorrect = newCorrect > newWrong;
		if constexpr (kSparse) { if (rampUntil == slice && freshCells >= kCutoff) rampUntil++; }      // :2626-2629
		if constexpr (kWide)
		{
			if (rampPossible && ((slice > 0 && rampUntil == slice - 1) || (rampUntil < slice && currentlyCorrect && falseFromCorrect)) && (!kSparse || lastNumCells < kCutoff))
			{
				rampRec = lastRec;                                               // :2630-2634
				rampRedoIndex = slice - 1;
			}
		}
		if (!correctFromCorrect) break;                                          // :2640-2647 (this slice is not kept)
		if (!currentlyCorrect && rampUntil < slice && rampPossible)
		{
			if constexpr (!kWide) { status = GA_UNSUPPORTED_RAMP; break; }       // rerun by a wide variant
			else
			{
				// ---- go back to the remembered slice and come forward again at the ramp width (:2648-2719) ----
				rampUntil = slice;
				const uint32_t back = rampRedoIndex;
				rampRedoIndex = slice;
				const uint32_t backRec = rampRec;
				rampRec = lastRec;
				lastRec = backRec;
				if (backRec == kSeedRecord) loadSeedState(); else loadRecordState(backRec);
				nPushed = back + 1;                                              // bandwidthPerSlice / correctness shrink to back + 1 entries
				while (nCkpt > 1 && slot.ckpt[nCkpt - 1] != kSeedRecord && recSlice(slot.ckpt[nCkpt - 1]) > back) nCkpt--;
				if constexpr (kSparse)
				{
					lastNumCells = backRec == kSeedRecord ? 0u : slot.arena[backRec + 1];
					if (overriding)                                              // :2673-2700
					{
						if (recJ(ovPreRec) > recJ(lastRec)) { overriding = false; ovCount = 0; }
						// "shorten": the reference swaps an empty slice (j = SIZE_MAX) into the back of its list, does not pop it and tests the
						// back's j again -- once entered, that loop never ends (:2690-2694).  Reported as an assertion.
						else if (ovCount > 0 && (uint64_t)(ovFirst + ovCount - 1) * W > recJ(lastRec)) { status = GA_ASSERTION; break; }
					}
					while (nOv > 0 && (uint64_t)slot.ovr[2 * (nOv - 1) + 1] * W > recJ(lastRec)) nOv--;
				}
				redone = true;
				slice = back;                                                    // the loop increment makes it back + 1
				continue;
			}
		}
		const uint32_t thisMem = freshCells * 4 + (uint32_t)cn * 28;             // estimatedMemory (:136-139)
		bool closedWindow = false;
		if constexpr (kSparse)
		{
			// ---- the slice is kept: what the reference now freezes of it, and the override window it opens, extends or closes (:2721-2764) ----
			bool pushTemps = false, closing = false;
			if (!overriding && freshCells >= kCutoff && lastNumCells < kCutoff) { ovPreRec = lastRec; overriding = true; ovFirst = slice; ovCount = 0; pushTemps = true; }
			else if (overriding) { if (freshCells < kCutoff) closing = true; else pushTemps = true; }
			// lastSlice = newSlice.getFrozenSqrtEndScores() always follows (:2805), getFrozenScores for a window's list (:2729, 2762): both assert
			// that the slice's scores lie within 16 bits of their minimum (NodeSlice.h:344, 372) -- which a long node's untouched columns break
			if (freshSparse && (freshEndsTooFar || (pushTemps && freshBeforeTooFar))) { status = GA_ASSERTION; break; }
			if (closing)
			{
				status = closeWindow();
				if (status != GA_OK) break;
				if (GA_LANE0) slot.ckpt[nCkpt] = lastRec;                        // result.slices.push_back(lastSlice) (:2752)
				nCkpt++;
				closedWindow = true;
				wave_order();
			}
			if (pushTemps) ovCount++;
		}
		if (GA_LANE0)
		{
			uint32_t* hdr = slot.arena + arenaTop;
			union { double d; uint32_t w[2]; } a, b;
			a.d = newCorrect; b.d = newWrong;
			hdr[5] = (uint32_t)((currentlyCorrect ? 1 : 0) | (falseFromCorrect ? 2 : 0) | (useRamp ? 4 : 0));
			hdr[6] = a.w[0]; hdr[7] = a.w[1]; hdr[8] = b.w[0]; hdr[9] = b.w[1]; hdr[10] = slice;
			slot.slice_off[slice] = thisRec;
			slot.slice_flags[slice] = (uint8_t)((currentlyCorrect ? 1 : 0) | (falseFromCorrect ? 2 : 0) | (useRamp ? 4 : 0) | ((freshSparse && freshBeforeTooFar) ? 16 : 0));
		}
		if (nPushed != slice) { status = GA_ASSERTION; break; }                  // :2768
		nPushed++;
		arenaTop += need;
		logCorrect = newCorrect; logWrong = newWrong;
		if (closedWindow) { storeRec = thisRec; storeMem = thisMem; }            // storeSlice = newSlice.getFrozenSqrtEndScores() (:2756)
		if constexpr (kWide)
		{
			if (rampPossible || kSparse)
			{
				// checkpoint = cheapest slice of each sqrt window (:2772-2786)
				if (slice % sampling == 0)
				{
					if (nCkpt == 0 || recSlice(storeRec) != recSlice(slot.ckpt[nCkpt - 1]))
					{
						if (GA_LANE0) slot.ckpt[nCkpt] = storeRec;
						nCkpt++;
						storeRec = thisRec; storeMem = thisMem;
					}
				}
				if (thisMem < storeMem) { storeRec = thisRec; storeMem = thisMem; }
				wave_order();
			}
		}
		lastRec = thisRec;
		lastNumCells = freshCells;
		adoptSlice();
	}
	if constexpr (kSparse)
	{
		// a window still open when the slices end (:2810-2825); its checkpoints go, nothing is pushed
		if (status == GA_OK && overriding) status = closeWindow();
	}
	out.n_run = nRun;
	GA_LAP(0);

	// ---- drop the wrongly aligned tail (removeWronglyAlignedEnd, :2554-2569) ----
	uint32_t kept = nPushed;
	if (status == GA_OK && kept > 0)
	{
		wave_sync();
		bool ok = slot.slice_flags[kept - 1] & 1;
		while (!ok)
		{
			kept--;
			if (kept == 0) break;
			ok = (slot.slice_flags[kept - 1] & 2) != 0;
		}
	}
	out.n_valid = status == GA_OK ? kept : 0;

	// ---- after a ramp redo the slices the traceback sees are those the reference would recompute from its
	// checkpoints (getSlicesFromTable :2858-2943); a checkpoint taken before the redo can disagree with the slices kept ----
	const uint32_t* inTab = slot.slice_off;       // record traced through, per slice
	const uint32_t* belowTab = slot.slice_off;    // record standing for the slice above a slice boundary
	uint32_t startRec = kept > 0 ? slot.slice_off[kept - 1] : 0;
	if constexpr (kWide)
	{
		const bool tableTouched = redone || (kSparse && anyOverride);             // (otherwise the checkpoint list is in order by construction)
		if (status == GA_OK && tableTouched)
		{
			// the checks at the end of getSqrtSlices (:2833-2854) look at the whole checkpoint list, before the wrongly aligned tail
			// is trimmed (removeWronglyAlignedEnd is the caller's next step, :3002,3018) and whether or not anything is kept
			wave_sync();
			if (nCkpt == 0) status = GA_ASSERTION;                                // :2834
			for (uint32_t i = 1; i < nCkpt && status == GA_OK; i++)
			{
				const uint32_t a = slot.ckpt[i - 1], b = slot.ckpt[i];
				if (i >= 2 && !(recJ(b) > recJ(a))) status = GA_ASSERTION;                           // :2835-2838
				const int ma = a == kSeedRecord ? 0 : (int)slot.arena[a + 2], mb = b == kSeedRecord ? 0 : (int)slot.arena[b + 2];
				if (mb < ma) status = GA_ASSERTION;                                                // :2839-2842
			}
			if constexpr (kSparse)
			{
				for (uint32_t i = 1; i < nOv && status == GA_OK; i++) if (!(slot.ovr[2 * i] > slot.ovr[2 * (i - 1) + 1])) status = GA_ASSERTION;   // :2850-2853
			}
			if (status != GA_OK) out.n_valid = 0;
		}
		if (status == GA_OK && tableTouched && kept > 0 && numSlices >= 4)
		{
			wave_sync();
			const auto firstPassColumns = out.n_columns;                          // the column-update count is the first pass's (cellsProcessed)
			const auto firstPassNodes = out.max_band_nodes;
			// trimmed checkpoints (:2566-2568; the overrides are not trimmed)
			while (status == GA_OK && nCkpt > 1 && recJ(slot.ckpt[nCkpt - 1]) >= (uint64_t)kept * W) nCkpt--;
			for (uint32_t sIdx = 0; sIdx < kept; sIdx++) if (GA_LANE0) slot.below_off[sIdx] = slot.slice_off[sIdx];
			wave_sync();
			// getTraceFromTable's walk over the checkpoints, last to first (:917-947): between two of them the reference recomputes the slices
			// (asserting what getSlicesFromTable asserts, freezing every recomputed slice), and where a checkpoint is the last slice of the
			// next override window it follows that window's links instead and recomputes nothing of it
			uint32_t usedLo = 0xffffffffu;                                        // lastBacktraceOverrideStartJ / 64
			int ovIdx = (int)nOv - 1;
			for (uint32_t i = nCkpt; i-- > 0 && status == GA_OK;)
			{
				const uint32_t ck = slot.ckpt[i];
				const uint32_t ckSlice = recSlice(ck);                               // 0xffffffff for the seed
				const uint32_t firstSlice = ckSlice + 1;                             // wraps to 0 for the seed
				uint32_t endSlice = i + 1 == nCkpt ? kept : recSlice(slot.ckpt[i + 1]) + 1;
				if (ck != kSeedRecord && GA_LANE0) slot.below_off[ckSlice] = ck;
				if (firstSlice == kept)
				{
					if (i + 1 != nCkpt) status = GA_ASSERTION;                       // :911
					continue;
				}
				if constexpr (kSparse)
				{
					if (!(usedLo > firstSlice)) { status = GA_ASSERTION; break; }     // assert(overrideLastJ > startSlice * 64) (:2862)
					if (endSlice >= usedLo) endSlice = usedLo;                       // :2865
				}
				if (!(endSlice > firstSlice) || endSlice > kept) { status = GA_ASSERTION; break; }   // :2866-2867
				if constexpr (kSparse)
				{
					// result.push_back(newSlice.getFrozenScores()) for every recomputed slice (:2908; NodeSlice.h:344)
					wave_sync();
					for (uint32_t sl = firstSlice; sl < endSlice; sl++) if (slot.slice_flags[sl] & 16) status = GA_ASSERTION;
					if (status != GA_OK) break;
				}
				const bool consistent = ck == kSeedRecord || ck == slot.slice_off[ckSlice];
				if (!consistent && redone)
				{
					// the recompute would NOT reproduce the kept slices: a checkpoint taken before a ramp redo
					loadRecordState(ck);
					for (uint32_t sl = firstSlice; sl < endSlice && status == GA_OK; sl++)
					{
						const int bandwidth = (slot.slice_flags[sl] & 4) ? L.ramp_bw : L.initial_bw;
						status = runSlice(sl, bandwidth);
						if (status != GA_OK) break;
						if (freshSparse && (freshEndsTooFar || freshBeforeTooFar)) { status = GA_ASSERTION; break; }
						if (GA_LANE0) { slot.arena[arenaTop + 10] = sl; slot.slice_off[sl] = (uint32_t)arenaTop; if (sl + 1 < endSlice || i + 1 == nCkpt) slot.below_off[sl] = (uint32_t)arenaTop; }
						arenaTop += need;
						adoptSlice();
					}
				}
				if constexpr (kSparse)
				{
					if (status == GA_OK && ck != kSeedRecord && ovIdx >= 0 && ckSlice == slot.ovr[2 * ovIdx + 1])
					{
						// GetBacktrace through the window (:939-946): its slices are the first pass's; flag them for the traceback's row-63 rule
						const uint32_t lo = slot.ovr[2 * ovIdx];
						if (GA_LANE0) for (uint32_t sl = lo; sl <= ckSlice; sl++) slot.slice_flags[sl] |= 8;
						usedLo = lo;
						ovIdx--;
					}
				}
			}
			wave_sync();
			if (status == GA_OK)
			{
				// the trace starts at the last checkpoint when that is the last kept slice itself (:908-916), else at the last slice of the last window
				const uint32_t lastCk = slot.ckpt[nCkpt - 1];
				startRec = (lastCk != kSeedRecord && recSlice(lastCk) + 1 == kept) ? lastCk : slot.slice_off[kept - 1];
			}
			belowTab = slot.below_off;
			out.n_columns = firstPassColumns;
			out.max_band_nodes = firstPassNodes;
		}
	}

	// ---- traceback (getTraceFromTable :894-957 with pickBacktracePredecessor :493-591) ----
	if (status == GA_OK && kept > 0)
	{
		if (numSlices < 4) status = GA_ASSERTION;                                // assert(slice.samplingFrequency > 1) (:906)
	}
	if (status == GA_OK && kept > 0)
	{
		uint8_t* tr = slot.trace;
		const int big = (int)job.n_rows;                                         // getValueOrMax default = sequence.size()
		uint32_t sIdx = kept - 1;
		uint32_t off = inTab[sIdx];
		uint32_t nN = slot.arena[off], nC = slot.arena[off + 1];
		SliceRec cur = slice_at(slot.arena, off, nN, nC);
		const SliceRec from = slice_at(slot.arena, startRec, slot.arena[startRec], slot.arena[startRec + 1]);   // = cur unless a ramp redo left a stale checkpoint
		out.score = (int32_t)from.hdr[2];
		uint32_t node = from.nodes[from.hdr[3]];
		uint32_t offset = from.hdr[4];
		uint32_t row = sIdx * W + (W - 1);
		uint32_t len = 0, nodeSteps = 0;
		out.start_node = node; out.start_offset = offset; out.start_row = row;
		SliceRec prv = cur;
		uint32_t pN = 0;
		// node lists (node id, first column in the record) of the slice traced through and of the one above it, in LDS
		uint32_t* curNodes = ws.cn_node; uint32_t* curBase = ws.cn_colBase;
		uint32_t* prvNodes = ws.pn_node; uint32_t* prvBase = ws.pn_colBase;
		auto loadTable = [&](uint32_t* tn, uint32_t* tb, const SliceRec& r, uint32_t n) {
			wave_order();
			for (uint32_t c = 0; c < n; c += LANES)
			{
				store_lanes(tn + c, (int)(n - c), load_lanes(r.nodes + c, (int)(n - c), 0));
				store_lanes(tb + c, (int)(n - c), load_lanes(r.colBase + c, (int)(n - c), 0));
			}
			wave_order();
		};
		auto loadPrev = [&]() {
			if (sIdx == 0) return;
			uint32_t o = belowTab[sIdx - 1];
			pN = slot.arena[o];
			prv = slice_at(slot.arena, o, pN, slot.arena[o + 1]);
			loadTable(prvNodes, prvBase, prv, pN);
		};
		wave_sync();
		loadTable(curNodes, curBase, cur, nN);
		loadPrev();
		auto valuePrevLastRow = [&](uint32_t n, uint32_t o2) -> int {
			// row 63 of the slice before; before slice 0 that is the all-zero seed slice
			if (sIdx == 0) return n == job.seed_node ? 0 : big;
			return stored_value(prv, prvNodes, prvBase, pN, n, o2, W - 1, big);
		};
		// A window = up to 64 consecutive columns of one node in one slice, one column per lane, so a
		// run of steps inside a node costs no memory round trips: every lane evaluates its column
		// at the wanted row (WordSlice.h:223-229) and the few values a step needs are read back.
		struct Window { VU vp, vn; VI before; uint32_t slice, node; int lo, n; bool valid, present; };
		Window cw, pw;
		cw.valid = false; pw.valid = false;
		cw.vp = VU(0); cw.vn = VU(0); cw.before = VI(0); pw.vp = VU(0); pw.vn = VU(0); pw.before = VI(0);
		cw.slice = cw.node = pw.slice = pw.node = 0; cw.lo = cw.n = pw.lo = pw.n = 0; cw.present = pw.present = false;
		VI winBases = VI(0);
		VI rowv = load_lanes(rows + sIdx * W, W, 0);
		uint32_t rowvSlice = sIdx;
		const VI lane = lane_iota();
		VI winRec = VI(0);                         // graph record of the node the current window lies in
		uint32_t winRecNode = 0xffffffffu;
		auto loadWindow = [&](Window& w, const SliceRec& rec, const uint32_t* tn, const uint32_t* tb, uint32_t recNodes, uint32_t sliceIdx, uint32_t n, uint32_t hiOffset) {
			w.slice = sliceIdx; w.node = n;
			w.lo = hiOffset >= (uint32_t)(LANES - 1) ? (int)(hiOffset - (LANES - 1)) : 0;
			w.n = (int)hiOffset - w.lo + 1;
			int sl = find_slot(tn, (int)recNodes, n);
			w.present = sl >= 0;
			w.valid = true;
			if (sl >= 0)
			{
				uint32_t at = tb[sl] + (uint32_t)w.lo;
				w.vp = load_lanes_u64(rec.vp + at, w.n);
				w.vn = load_lanes_u64(rec.vn + at, w.n);
				w.before = load_lanes(rec.before + at, w.n, 0);
			}
		};
		while (true)
		{
			if (row == 0xffffffffu) break;                                       // reached the row before the first one
			if (len + LANES >= L.trace_cap) { status = GA_CAP_TRACE; break; }
			int r = (int)(row - sIdx * W);
			if (rowvSlice != sIdx) { rowv = load_lanes(rows + sIdx * W, W, 0); rowvSlice = sIdx; }
			if constexpr (kSparse)
			{
				// inside an override window the reference follows precomputed links, and a cell of a slice's last row without an end
				// score has none (BacktraceItem::end, :311-317; assert(!current.end) :211; the entry cell must be among them, :202-209)
				if (r == W - 1 && (slot.slice_flags[sIdx] & 8))
				{
					const int sx = find_slot(curNodes, (int)nN, node);
					if (sx < 0) { status = GA_ASSERTION; break; }
					const uint8_t* ex = (const uint8_t*)(cur.before + cur.hdr[1]);
					if (ex[curBase[sx] + offset] == 0) { status = GA_ASSERTION; break; }
				}
			}
			// ---- make the current window cover this column (and its left neighbour when there is one) ----
			bool covers = cw.valid && cw.slice == sIdx && cw.node == node && (int)offset >= cw.lo && (int)offset < cw.lo + cw.n && !((int)offset == cw.lo && offset > 0);
			if (!covers)
			{
				if (pw.valid && pw.slice == sIdx && pw.node == node && (int)offset >= pw.lo && (int)offset < pw.lo + pw.n && !((int)offset == pw.lo && offset > 0))
				{
					cw = pw;                                                   // the window fetched for the slice boundary becomes current
				}
				else
				{
					if (winRecNode != node) { winRec = g_record(g, node); winRecNode = node; }      // travels together with the window's words
					loadWindow(cw, cur, curNodes, curBase, nN, sIdx, node, offset);
					const uint64_t firstCol = (((uint64_t)(uint32_t)read_lane(winRec, 1) << 32) | (uint32_t)read_lane(winRec, 0)) + (uint32_t)cw.lo;
					VI at = lane + (int)(firstCol & 15);
					winBases = (gather(g.seq2 + (firstCol >> 4), at >> 4) >> ((at & 15) << 1)) & 3;
				}
				pw.valid = false;
			}
			if (winRecNode != node) { winRec = g_record(g, node); winRecNode = node; }
			if (!cw.present) { status = GA_ASSERTION; break; }                   // assert(slice.scores.hasNode(nodeIndex)) (:498)
			int rel = (int)offset - cw.lo;
			uint64_t maskR = r < 63 ? ~(~0ull << (r + 1)) : ~0ull;
			VI valR = cw.before + vpopc(maskR & cw.vp) - vpopc(maskR & cw.vn);
			int here = read_lane(valR, rel);
			if (row == 0 && node == job.seed_node && (here == 0 || here == 1)) { row = 0xffffffffu; continue; }   // free start (:500)
			// ---- a run of diagonal steps inside the window, all at once ----
			// Lane L looks at the diagonal cell (column L, row L + r - rel).  A step from column L+1
			// to L is diagonal iff the horizontal move is not taken (:541-546) and the diagonal cell
			// has the matching score (:556-571); the run is the stretch of such lanes below `rel`.
			if (rel >= 1 && r >= 1)
			{
				const VI rho = lane + (r - rel);
				const VU mA = mask_low_bits(rho + 1);
				const VI A = cw.before + vpopc(cw.vp & mA) - vpopc(cw.vn & mA);
				// one row further down: add that row's vertical delta
				const VI B = A + bit64_at(cw.vp, rho + 1) - bit64_at(cw.vn, rho + 1);
				const VI mOwn = (lane_gather(rowv, rho) >> winBases) & 1;
				const VI hereSrc = shl1(A, 0), mSrc = shl1(mOwn, 0);
				const uint64_t cond = ballot(B > hereSrc - 1) & ballot(A == hereSrc - 1 + mSrc) & ballot((lane < rel) && (rho > -1));
				const uint64_t broken = ~(cond << (64 - rel));
				int run = broken == 0 ? 64 : __builtin_clzll(broken);
				run = run < rel ? run : rel;
				if (run >= 2)
				{
					// the whole run is `run` diagonal moves inside the node
					store_lanes(tr + len, run, VI(GA_MOVE_DIAG));
					len += (uint32_t)run;
					offset -= (uint32_t)run;
					row -= (uint32_t)run;
					// the run ended because the next step is not a plain diagonal one: take that step right away from the new
					// cell (same window, same slice) unless the window has to move first
					rel -= run;
					r -= run;
					if (rel == 0 && offset > 0) continue;
					maskR = r < 63 ? ~(~0ull << (r + 1)) : ~0ull;
					valR = cw.before + vpopc(maskR & cw.vp) - vpopc(maskR & cw.vn);
					here = read_lane(valR, rel);
					if (row == 0 && node == job.seed_node && (here == 0 || here == 1)) { row = 0xffffffffu; continue; }
					if (len + LANES >= L.trace_cap) { status = GA_CAP_TRACE; break; }
				}
			}
			const int rowCode = read_lane(rowv, r);
			const int base = read_lane(winBases, rel);
			const bool match = (rowCode >> base) & 1;
			// values one row up: same window for r > 0, the slice above (its row 63) for r == 0
			VI valUp;
			bool upKnown = true;
			if (r > 0)
			{
				const uint64_t maskU = ~(~0ull << r);
				valUp = cw.before + vpopc(maskU & cw.vp) - vpopc(maskU & cw.vn);
			}
			else if (sIdx == 0)
			{
				valUp = VI(node == job.seed_node ? 0 : big);
			}
			else
			{
				if (!(pw.valid && pw.slice == sIdx - 1 && pw.node == node && pw.lo == cw.lo && pw.n == cw.n))
					loadWindow(pw, prv, prvNodes, prvBase, pN, sIdx - 1, node, (uint32_t)(cw.lo + cw.n - 1));
				if (pw.present) valUp = pw.before + vpopc(pw.vp) - vpopc(pw.vn);
				else valUp = VI(big);
			}
			(void)upKnown;
			int res = 0;
			int viaNeighbour = 0;                                                // ordinal of the in-neighbour a move enters (0 inside a node)
			const uint32_t curNode = node, curOffset = offset;
			auto decide = [&](int horizontal, int diagonal, uint32_t un, uint32_t uo) -> int {
				if (horizontal < here - 1) return -1;
				if (horizontal == here - 1) { node = un; offset = uo; return 1; }
				if (match)
				{
					if (diagonal < here) return -1;
					if (diagonal == here) { node = un; offset = uo; row = row - 1; return 2; }
				}
				else
				{
					if (diagonal < here - 1) return -1;
					if (diagonal == here - 1) { node = un; offset = uo; row = row - 1; return 2; }
				}
				return 0;
			};
			if (curOffset == 0)
			{
				const int inDeg = read_lane(winRec, 3) & 0xffff;
				if (inDeg <= kNbr)
				{
					// the last columns of all in-neighbours (this slice, and the slice above when at its first row) in one round trip:
					// lane k < inDeg looks at in-neighbour k; the record holds the neighbours and their lengths
					const VI nbr = lane_gather(winRec, vmin(lane, VI(kNbr - 1)) + 8);
					const VI nbrLast = lane_gather(winRec, vmin(lane, VI(kNbr - 1)) + 12) - 1;
					VI idxCur = VI(-1), idxPrv = VI(-1);
					for (int k = 0; k < inDeg; k++)
					{
						const uint32_t m = (uint32_t)read_lane(nbr, k);
						const int sc = find_slot(curNodes, (int)nN, m);
						if (sc >= 0) idxCur = select(lane == k, VI((int)curBase[sc]) + nbrLast, idxCur);
						if (r == 0 && sIdx > 0)
						{
							const int sp = find_slot(prvNodes, (int)pN, m);
							if (sp >= 0) idxPrv = select(lane == k, VI((int)prvBase[sp]) + nbrLast, idxPrv);
						}
					}
					const VB haveC = idxCur > -1;
					const VI safeC = select(haveC, idxCur, VI(0));
					const VU vpC = select(haveC, gather64(cur.vp, safeC), VU(0)), vnC = select(haveC, gather64(cur.vn, safeC), VU(0));
					const VI beC = gather(cur.before, safeC);
					const VI horV = select(haveC, beC + vpopc(vpC & VU(maskR)) - vpopc(vnC & VU(maskR)), VI(big));
					VI diaV;
					if (r > 0)
					{
						const uint64_t maskU = ~(~0ull << r);
						diaV = select(haveC, beC + vpopc(vpC & VU(maskU)) - vpopc(vnC & VU(maskU)), VI(big));
					}
					else if (sIdx == 0) diaV = select(nbr == (int)job.seed_node, VI(0), VI(big));
					else
					{
						const VB haveP = idxPrv > -1;
						const VI safeP = select(haveP, idxPrv, VI(0));
						diaV = select(haveP, gather(prv.before, safeP) + vpopc(gather64(prv.vp, safeP)) - vpopc(gather64(prv.vn, safeP)), VI(big));
					}
					for (int k = 0; k < inDeg && res == 0; k++)
					{
						viaNeighbour = k;
						res = decide(read_lane(horV, k), read_lane(diaV, k), (uint32_t)read_lane(nbr, k), (uint32_t)read_lane(nbrLast, k));
					}
				}
				else
				{
					for (uint32_t e = g.in_off[curNode]; e < g.in_off[curNode + 1] && res == 0; e++)
					{
						uint32_t m = g.in_nbr[e];
						uint32_t mo = g_len(g, m) - 1;
						viaNeighbour = (int)(e - g.in_off[curNode]);
						int horizontal = stored_value(cur, curNodes, curBase, nN, m, mo, r, big);
						int diagonal = 0;
						if (horizontal > here - 1) diagonal = r == 0 ? valuePrevLastRow(m, mo) : stored_value(cur, curNodes, curBase, nN, m, mo, r - 1, big);
						res = decide(horizontal, diagonal, m, mo);
					}
				}
			}
			else res = decide(read_lane(valR, rel - 1), read_lane(valUp, rel - 1), curNode, curOffset - 1);
			if (curOffset == 0 && (res == 1 || res == 2)) nodeSteps++;           // left the node through its first column
			if (res < 0) { status = GA_ASSERTION; break; }
			if (res == 0)
			{
				int up = read_lane(valUp, rel);
				if (up != here - 1) { status = GA_ASSERTION; break; }            // assert(false) (:588)
				row = row - 1;
				res = 3;
			}
			// put the move away (the step onto the row before the first one is not part of the trace, :949-950)
			if (row != 0xffffffffu)
			{
				if (viaNeighbour > 62) { status = GA_CAP_TRACE; break; }
				const int code = res == 1 ? GA_MOVE_LEFT : res == 2 ? GA_MOVE_DIAG : GA_MOVE_UP;
				if (GA_LANE0) tr[len] = (uint8_t)(code | (res == 3 ? 0 : (viaNeighbour << 2)));
				len++;
			}
			if (res >= 2 && r == 0 && row != 0xffffffffu)
			{
				// stepped into the slice above
				sIdx--;
				if (belowTab[sIdx] == inTab[sIdx])
				{
					cur = prv; nN = pN;
					uint32_t* t = curNodes; curNodes = prvNodes; prvNodes = t;
					t = curBase; curBase = prvBase; prvBase = t;
				}
				else
				{
					// the slice is traced through in another version than the one the boundary step looked at
					const uint32_t o = inTab[sIdx];
					nN = slot.arena[o];
					cur = slice_at(slot.arena, o, nN, slot.arena[o + 1]);
					loadTable(curNodes, curBase, cur, nN);
					pw.valid = false;
				}
				loadPrev();
			}
		}
		GA_LAP(5);
		// hand the steps over: claim exactly `len` entries of the pool and copy them, 64 at a time
		if (status == GA_OK)
		{
			wave_sync();
			const uint32_t words = (len + 3) / 4;
			const uint64_t at = wave_claim(L.trace_top, (uint64_t)words * 4, L.trace_pool_cap);      // (commits only when it fits)
			if (at == ~0ull) status = GA_CAP_TRACE;
			else
			{
				uint32_t* dst = (uint32_t*)(L.traces + at);
				const uint32_t* src = (const uint32_t*)tr;
				for (uint32_t c = 0; c < words; c += LANES) store_lanes(dst + c, (int)(words - c), load_lanes(src + c, (int)(words - c), 0));
				out.trace_off = at;
				out.trace_len = len;
				out.n_node_steps = nodeSteps;
			}
		}
	}
	GA_LAP(6);
#undef GA_LAP
	out.status = status;
	if (GA_LANE0) L.outs[jobIndex] = out;
}

}  // namespace gak
