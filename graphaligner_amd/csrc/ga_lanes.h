// ga_lanes.h -- the extension program with LANES = READS: every lane of a wavefront owns one read direction
// (one extension job) and carries it through band selection, slice fill and traceback on its own; the 64 lanes
// of a wave advance slice by slice in lockstep.
//
// Why: the wave-per-read program (ga_kernel.h, lanes = rows) spends ~19 vector instructions per graph column on
// a six-step cross-lane scan and is bound by vector-instruction issue at 6.8 % of the HBM roofline.  Here a
// column is the reference's own bit-vector step (getNextSlice, GraphAligner.h:1349-1427) on a lane's private
// 64-bit VP/VN words: ~100 instructions advance 64 reads by one column each.
//
// What a lane computes is exactly what ga_kernel.h computes for jobs whose bands have no cycle and that never
// take the ramp redo; anything else (a cyclic band, a ramp redo, a band wider than the lane's LDS tables, node
// degrees above four, reads shorter than four slices ...) ends the lane's job with a status that sends it to
// the wave-per-read ladder, which carries those paths.  Results are bit-identical either way.
//
// Memory, all sized per wave:
//   LDS      per-lane band tables, [word][lane] interleaved (bank = lane: never a conflict), 10 * MAXN words per lane
//   HBM      end words of the previous / current slice  [column][lane]                       (4 B per column; measured against
//            [lane][column]: the lane-major form is 12 % slower, profiles/r3_ab_arena.txt)
//            slice records: VP, VN, scoreBeforeStart, end word = 24 B per column, in blocks of R columns per lane
//            per-slice headers and band node lists for the traceback, [slice][k][lane]
//            traceback moves, 4 per word, [k][lane]
//
// The same source builds for gfx950 and, with GA_EMULATE, for the host (tests only): a lane's program is plain
// scalar code; the few wave-level decisions (loop while ANY lane has work, the slice's row range = MAX over
// lanes) sit in the driver at the bottom.
#pragma once
#include <stdint.h>

#include "ga_types.h"

namespace gal {

#ifdef GA_EMULATE
#define GAL_FN inline
#else
#define GAL_FN __device__ __forceinline__
#endif

#if defined(GA_STAMPS) && !defined(GA_EMULATE)
GAL_FN uint64_t lap_clock() { return __builtin_readcyclecounter(); }
#else
GAL_FN uint64_t lap_clock() { return 0; }
#endif

constexpr int W = 64;
constexpr int INF = 0x3fffffff;
constexpr uint32_t kCutoff = 200000;          // GraphAlignerCommon.h:10
constexpr int kRecBytes = 24;                 // VP 8, VN 8, scoreBeforeStart 4, end word 4
constexpr int kHdrWords = 8;                  // per slice and lane: nNodes, nCols, minScore, minSlot, minOffset, flags, rowBase, spare
constexpr uint32_t kNone = 0xffu;

// end word of a column: scoreEnd << 3 | scoreBeforeExists << 2 | VN bit 63 << 1 | VP bit 63
GAL_FN int ew_end(uint32_t e) { return (int)(e >> 3); }
GAL_FN int ew_end2(uint32_t e) { return (int)(e >> 3) - (int)(e & 1) + (int)((e >> 1) & 1); }       // score one row above the end (row 62)

// launch parameters of the lanes kernel
struct GaLanesLaunch
{
	GaDevGraph graph;
	const GaHmmTables* hmm;
	const uint64_t* eq;            // per job and slice: 4 match words (A, C, G, T) + 1 meta word (bits 0-2 exact code of row 63, bit 3 an invalid row)
	const GaJob* jobs;
	GaJobOut* outs;
	const uint32_t* job_list;      // wave g takes jobs job_list[g * lanes_per_wave + lane]; nullptr = identity
	uint32_t n_jobs;
	uint32_t lanes_per_wave;       // 64, or fewer to put more waves on a SIMD
	uint32_t* next_group;          // device work counter
	uint8_t* scratch;              // [n_waves][wave_bytes]
	uint64_t wave_bytes;
	uint8_t* traces;               // trace pool (bytes), shared with the ladder kernels
	uint64_t* trace_top;
	uint64_t trace_pool_cap;
	uint32_t cap_cols;             // columns per slice (end planes)
	uint32_t cap_rows;             // arena rows per wave: the columns of all its lanes' bands over all slices (blocks of 8 per lane and node)
	uint32_t max_slices;
	uint32_t cap_moves;            // traceback moves per lane (staging)
	int32_t initial_bw, ramp_bw;
	uint32_t emit_runs;            // 1: the traceback hands back node runs instead of moves (batches whose results need no cell lists)
	uint32_t reserved;
};

// byte offsets inside a wave's scratch region
struct WaveLayout { uint64_t endA, endB, hdr, snodes, moves, arena, bytes; };
template <int N>
inline
#ifndef GA_EMULATE
__host__ __device__
#endif
WaveLayout wave_layout(uint32_t capCols, uint32_t capRows, uint32_t maxSlices, uint32_t capMoves, uint32_t ls = 64)
{
	// ls = lane stride of the planes = how many lanes of a wave carry jobs (64, or fewer in the variants that trade lanes for LDS per lane)
	auto up = [](uint64_t x) { return (x + 255) & ~255ull; };
	WaveLayout l;
	uint64_t at = 0;
	const uint64_t row = 4ull * ls;
	l.endA = at; at = up(at + row * (capCols + 64));     // (+ slack: operands are requested up to two chunks past a node's end)
	l.endB = at; at = up(at + row * (capCols + 64));
	l.hdr = at; at = up(at + row * kHdrWords * (maxSlices + 1));
	l.snodes = at; at = up(at + row * 2 * N * (maxSlices + 2));
	l.moves = at; at = up(at + row * ((capMoves + 3) / 4 + 16));
	l.arena = at; at = up(at + (uint64_t)kRecBytes * (capRows + 16) + 64ull * 8 * kRecBytes);     // (+ the spare block image behind the rows)
	l.bytes = at;
	return l;
}

// where the record of arena row `row` lives.  Rows are handed out to lanes in blocks of R when a node is started (fill_slice), so a
// node's columns are consecutive rows of one lane and a row belongs to exactly one lane.
template <int R> GAL_FN uint64_t rec_off(uint32_t row, int lane, uint32_t ls)
{
	(void)lane; (void)ls;
	return (uint64_t)row * kRecBytes;
}

// ---- per-lane LDS tables: word i of lane l sits at lds[i * LW + l] ------------------------------------------------
#ifndef GAL_WINCAP
#define GAL_WINCAP 24
#endif
constexpr int kStageWordsLane = 50;          // 32-bit words per lane of the staging image that follows the tables (64 lanes x kStageWords64 x 2 / 64)
template <int N> struct Lay
{
	static constexpr int H = 2 * N;                   // heap entries
	static constexpr int HB = N <= 13 ? 13 : N <= 29 ? 29 : N <= 59 ? 59 : 127;      // buckets the frozen map can grow to with N keys
	static constexpr int P_NODE = 0, P_END = N, P_PACK = 2 * N;                      // previous band: node, end word of its last column, colBase | (min - sliceMin) << 16
	static constexpr int C_NODE = 3 * N, C_GEO = 4 * N, C_PINFO = 5 * N;             // current band: node, colBase | len << 16, prev colBase | prev slot << 16
	static constexpr int X = 6 * N;
	// while the band is selected: hash-order bytes, heap
	static constexpr int X_HASH = X, X_HEAPN = X + N, X_HEAPP = X + N + H;
	static constexpr int HB_ORDER = 0, HB_NEXT = N, HB_BEFORE = 2 * N;                // byte offsets inside X_HASH
	// while the slice is filled: node minimum, end word of the node's last column, out-neighbour slots, order bytes
	static constexpr int C_MIN = X, C_END = X + N, C_OUT = X + 2 * N, X_ORD = X + 3 * N;
	static constexpr int OB_POST = 0, OB_COLOR = N, OB_STSLOT = 2 * N, OB_STCUR = 3 * N;
	// during the traceback: node lists of the slice traced through and of the slice above it
	static constexpr int T_CN = 0, T_CB = N, T_PN = 2 * N, T_PB = 3 * N;
	// window of column records (5 words each): the rest of the tables and, behind them, the words of the staging image (kStageWordsLane
	// per lane at any lane stride), which the traceback does not use
	static constexpr int T_WIN = 4 * N, T_WINCOLS = (6 * N + kStageWordsLane) / 5 < GAL_WINCAP ? (6 * N + kStageWordsLane) / 5 : GAL_WINCAP;
	static_assert(4 * N + T_WINCOLS * 5 <= 10 * N + kStageWordsLane && T_WINCOLS >= 8, "traceback window does not fit");
	static constexpr int WORDS = 10 * N;
	static_assert((2 * N + HB + 3) / 4 <= N, "hash bytes do not fit");
};

static_assert(kStageWordsLane == 50 && GAL_WINCAP <= 32, "kStageWords64 * 2; the window's bases are one 64-bit word");
struct Lds
{
	uint32_t* base;     // already offset by the lane
	int lw;             // lane stride in words
	GAL_FN uint32_t rd(int i) const { return base[i * lw]; }
	GAL_FN void wr(int i, uint32_t v) const { base[i * lw] = v; }
	GAL_FN uint32_t rdh(int word, int k) const { return ((const uint16_t*)(base + (word + (k >> 1)) * lw))[k & 1]; }
	GAL_FN void wrh(int word, int k, uint32_t v) const { ((uint16_t*)(base + (word + (k >> 1)) * lw))[k & 1] = (uint16_t)v; }
	GAL_FN uint32_t rdb(int word, int k) const { return ((const uint8_t*)(base + (word + (k >> 2)) * lw))[k & 3]; }
	GAL_FN void wrb(int word, int k, uint32_t v) const { ((uint8_t*)(base + (word + (k >> 2)) * lw))[k & 3] = (uint8_t)v; }
};

// a lane's view of its wave's HBM scratch: plane[k * ls] is entry k of this lane
struct LaneMem
{
	Lds lds;
	uint32_t* endPrev;
	uint32_t* endCur;
	uint32_t* hdr;
	uint32_t* snodes;
	uint32_t* moves;
	uint8_t* arena;
	uint64_t* stage;              // device: the wave's LDS image of the open block of 8 arena rows
	uint32_t* laneBlocks;         // device: LDS, per lane two words: the first arena block of the node in hand, its number of blocks
	uint32_t usedChunks;          // device: 16-byte chunks of a block image that belong to lanes with a job (12 per lane)
	int lane;
	uint32_t flushPart, flushLane; // device: this thread's place in the block flush (tid % 12, tid / 12)
	int tid;                      // device: the thread's index in the wave (= lane, except in the variants where only the first `ls` lanes carry jobs
	                              // and the others shadow lane 0: every thread still takes part in the block flush)
	uint32_t ls;                  // lane stride of the planes and of the arena's blocks (a constant of the kernel variant: 64, 32 or 16)
};

// what a lane carries from slice to slice
struct LaneState
{
	uint32_t job, nRows, numSlices, seedNode, traceRows;
	const uint64_t* eq;
	int status;
	bool live;                    // still in the slice loop
	int pn, cn;
	int prevMin;
	uint32_t totalCols;
	uint32_t rowBase;             // arena row of this slice's first column
	uint64_t e0, e1, e2, e3;
	int rawAbove;                 // exact-compare code of the read character of row j-1 (7 = none)
	int sliceMin, minSlot;
	uint32_t minOffset;
	double logCorrect, logWrong;
	uint32_t nPushed, nRun, maxBandNodes, kept;
	uint64_t nColumns;
	uint32_t rampUntil;
	bool useRamp;
	uint64_t blaps[5];            // diagnostic builds: cycles inside the band phase (map order, previous band, heap, slots, processing order)
	uint64_t laps[8];             // diagnostic builds: cycles inside the traceback (fast steps, general steps, hand-over; [4..6] parts of the general step, [7] rounds | fast iterations << 32)
};

// ---- graph access ---------------------------------------------------------------------------------------------
GAL_FN const uint32_t* g_rec(const GaDevGraph& g, uint32_t node) { return g.node_rec + (uint64_t)node * GA_NODE_REC_WORDS; }

GAL_FN int find_in(const Lds& l, int table, int count, uint32_t key)
{
	int found = -1;
	for (int k = 0; k < count; k++) if (l.rd(table + k) == key) found = found < 0 ? k : found;
	return found;
}
// A node list of a narrow band (N <= 15) held in registers while the band is put together: a lookup is N compares instead of a
// chain of LDS reads each waited for.  Entries past the count hold a value no node index has.
constexpr uint32_t kNoNode = 0xffffffffu;
template <int N> struct NodeRegs
{
	static constexpr bool kOn = N <= 15;
	uint32_t v[kOn ? N : 1];
	GAL_FN void load(const Lds& l, int table, int count)
	{
		if constexpr (kOn)
		{
#pragma unroll
			for (int k = 0; k < N; k++) v[k] = l.rd(table + k);
#pragma unroll
			for (int k = 0; k < N; k++) v[k] = k < count ? v[k] : kNoNode;
		}
	}
	GAL_FN void clear()
	{
		if constexpr (kOn)
		{
#pragma unroll
			for (int k = 0; k < N; k++) v[k] = kNoNode;
		}
	}
	GAL_FN void set(int at, uint32_t node)
	{
		if constexpr (kOn)
		{
#pragma unroll
			for (int k = 0; k < N; k++) v[k] = k == at ? node : v[k];
		}
	}
	GAL_FN int find(const Lds& l, int table, int count, uint32_t key) const
	{
		if constexpr (kOn)
		{
			int found = -1;
#pragma unroll
			for (int k = N - 1; k >= 0; k--) found = v[k] == key ? k : found;
			return found;
		}
		else return find_in(l, table, count, key);
	}
};

// ---- std::unordered_map<size_t,..> iteration order after inserting the previous band's nodes one by one --------
// (NodeSlice.h:728-738 builds the frozen slice's map so; GraphAligner.h:1117 iterates it).  libstdc++: identity hash,
// bucket = key % B, B grows 13, 29, 59, 127 when the size would exceed it; a node entering an empty bucket becomes the
// list head, otherwise it goes behind its bucket's "before" node; a rehash re-inserts the nodes in iteration order.
template <int N> GAL_FN void hash_insert(const Lds& l, int i, uint32_t B, int& head)
{
	typedef Lay<N> LY;
	const uint32_t b = l.rd(LY::P_NODE + i) % B;
	const uint32_t before = l.rdb(LY::X_HASH, LY::HB_BEFORE + (int)b);
	if (before != 0xffu)
	{
		if (before == 0xfeu) { l.wrb(LY::X_HASH, LY::HB_NEXT + i, (uint32_t)head & 0xffu); head = i; }
		else { l.wrb(LY::X_HASH, LY::HB_NEXT + i, l.rdb(LY::X_HASH, LY::HB_NEXT + (int)before)); l.wrb(LY::X_HASH, LY::HB_NEXT + (int)before, (uint32_t)i); }
	}
	else
	{
		l.wrb(LY::X_HASH, LY::HB_NEXT + i, (uint32_t)head & 0xffu);
		if (head >= 0) l.wrb(LY::X_HASH, LY::HB_BEFORE + (int)(l.rd(LY::P_NODE + head) % B), (uint32_t)i);
		head = i;
		l.wrb(LY::X_HASH, LY::HB_BEFORE + (int)b, 0xfeu);
	}
}
template <int N> GAL_FN void hash_order(const Lds& l, int n)
{
	typedef Lay<N> LY;
	uint32_t B = 13;
	int head = -1;
	for (int b = 0; b < 13; b++) l.wrb(LY::X_HASH, LY::HB_BEFORE + b, 0xffu);
	for (int i = 0; i < n; i++)
	{
		const uint32_t grown = i == 13 ? 29u : i == 29 ? 59u : i == 59 ? 127u : 0u;
		if (grown)
		{
			int k = 0;
			for (int p = head; p >= 0; ) { l.wrb(LY::X_HASH, LY::HB_ORDER + k, (uint32_t)p); k++; const uint32_t nx = l.rdb(LY::X_HASH, LY::HB_NEXT + p); p = nx == 0xffu ? -1 : (int)nx; }
			B = grown;
			head = -1;
			for (uint32_t b = 0; b < B; b++) l.wrb(LY::X_HASH, LY::HB_BEFORE + (int)b, 0xffu);
			for (int q = 0; q < k; q++) hash_insert<N>(l, (int)l.rdb(LY::X_HASH, LY::HB_ORDER + q), B, head);
		}
		hash_insert<N>(l, i, B, head);
	}
	int k = 0;
	for (int p = head; p >= 0; ) { l.wrb(LY::X_HASH, LY::HB_ORDER + k, (uint32_t)p); k++; const uint32_t nx = l.rdb(LY::X_HASH, LY::HB_NEXT + p); p = nx == 0xffu ? -1 : (int)nx; }
}

// ---- std::priority_queue<.., std::greater<>> over (node, priority): libstdc++'s __push_heap / __adjust_heap ------
// (GraphAligner.h:1115; the pop order among equal priorities decides DPSlice::nodes order)
template <int N> GAL_FN void heap_sift_up(const Lds& l, int hole, uint32_t node, uint32_t prio)
{
	typedef Lay<N> LY;
	int parent = (hole - 1) / 2;
	while (hole > 0 && l.rdh(LY::X_HEAPP, parent) > prio)
	{
		l.wrh(LY::X_HEAPP, hole, l.rdh(LY::X_HEAPP, parent));
		l.wr(LY::X_HEAPN + hole, l.rd(LY::X_HEAPN + parent));
		hole = parent;
		parent = (hole - 1) / 2;
	}
	l.wrh(LY::X_HEAPP, hole, prio);
	l.wr(LY::X_HEAPN + hole, node);
}
template <int N> GAL_FN bool heap_push(const Lds& l, int& size, uint32_t node, int prio)
{
	if (size >= Lay<N>::H || prio < 0 || prio > 0xffff) return false;
	size++;
	heap_sift_up<N>(l, size - 1, node, (uint32_t)prio);
	return true;
}
template <int N> GAL_FN void heap_pop(const Lds& l, int& size)
{
	typedef Lay<N> LY;
	// std::pop_heap then pop_back: the last element is re-inserted from the root
	const int len = size - 1;
	const uint32_t node = l.rd(LY::X_HEAPN + len), prio = l.rdh(LY::X_HEAPP, len);
	size = len;
	if (len == 0) return;
	int hole = 0, child = 0;
	while (child < (len - 1) / 2)
	{
		child = 2 * (child + 1);
		if (l.rdh(LY::X_HEAPP, child) > l.rdh(LY::X_HEAPP, child - 1)) child--;
		l.wrh(LY::X_HEAPP, hole, l.rdh(LY::X_HEAPP, child));
		l.wr(LY::X_HEAPN + hole, l.rd(LY::X_HEAPN + child));
		hole = child;
	}
	if ((len & 1) == 0 && child == (len - 2) / 2)
	{
		child = 2 * (child + 1);
		l.wrh(LY::X_HEAPP, hole, l.rdh(LY::X_HEAPP, child - 1));
		l.wr(LY::X_HEAPN + hole, l.rd(LY::X_HEAPN + child - 1));
		hole = child - 1;
	}
	heap_sift_up<N>(l, hole, node, prio);
}

// ---- band selection at node granularity (projectForwardFromMinScore, GraphAligner.h:1110-1159) -------------------
template <int N> GAL_FN int project_band(const GaDevGraph& g, const LaneMem& m, LaneState& st, int bandwidth)
{
	typedef Lay<N> LY;
	const Lds& l = m.lds;
	const int pn = st.pn, prevMin = st.prevMin, expand = bandwidth + W;
	int cn = 0, heapSize = 0;
	uint32_t totalCols = 0;
	uint64_t bt = lap_clock();
	NodeRegs<N> prevNodes, curNodes;
	prevNodes.load(l, LY::P_NODE, pn);
	curNodes.clear();
	hash_order<N>(l, pn);
	{ const uint64_t t2 = lap_clock(); st.blaps[0] += t2 - bt; bt = t2; }
	auto add = [&](uint32_t node, int prevSlot, uint32_t len, uint32_t prevBase) -> int {
		if (cn >= N) return GA_CAP_NODES;
		if (len > 0xffffu || totalCols + len > 0xffffu) return GA_CAP_COLS;
		curNodes.set(cn, node);
		l.wr(LY::C_NODE + cn, node);
		l.wr(LY::C_GEO + cn, totalCols | (len << 16));
		l.wr(LY::C_PINFO + cn, prevSlot >= 0 ? (prevBase | ((uint32_t)prevSlot << 16)) : (kNone << 16));
		totalCols += len;
		cn++;
		return totalCols >= kCutoff ? GA_UNSUPPORTED_BAND : GA_OK;
	};
	// words 2 .. 7 of a node's graph record: length, degrees, first four out-neighbours
	struct Rec6 { uint32_t w[6]; };
	auto pushOut = [&](const Rec6& r, uint32_t node, int prio) -> int {
		const uint32_t deg = r.w[1] >> 16;
		if (deg <= 4)
		{
			// (the neighbour is picked by compares: indexing the record with the loop counter would put it in scratch memory)
			for (uint32_t e = 0; e < deg; e++)
			{
				const uint32_t nbr = e == 0 ? r.w[2] : e == 1 ? r.w[3] : e == 2 ? r.w[4] : r.w[5];
				if (!heap_push<N>(l, heapSize, nbr, prio)) return GA_CAP_HEAP;
			}
		}
		else for (uint32_t e = g.out_off[node]; e < g.out_off[node + 1]; e++) if (!heap_push<N>(l, heapSize, g.out_nbr[e], prio)) return GA_CAP_HEAP;
		return GA_OK;
	};
	auto loadRec6 = [&](uint32_t node, Rec6& r) {
		const uint32_t* rec = g_rec(g, node);
#pragma unroll
		for (int i = 0; i < 6; i++) r.w[i] = rec[2 + i];
	};
	// the previous band in map order; the record of the node after the one in hand is requested while that one is handled
	Rec6 cur, nxt;
	for (int i = 0; i < 6; i++) { cur.w[i] = 0; nxt.w[i] = 0; }
	if (pn > 0) loadRec6(l.rd(LY::P_NODE + (int)l.rdb(LY::X_HASH, LY::HB_ORDER)), cur);
	for (int k = 0; k < pn; k++)
	{
		const int s = (int)l.rdb(LY::X_HASH, LY::HB_ORDER + k);
		if (k + 1 < pn) loadRec6(l.rd(LY::P_NODE + (int)l.rdb(LY::X_HASH, LY::HB_ORDER + k + 1)), nxt);
		const Rec6 rec = cur;
		cur = nxt;
		const uint32_t pack = l.rd(LY::P_PACK + s);
		if ((int)(pack >> 16) > bandwidth) continue;                              // node.minScore <= minScore + bandwidth (:1119)
		const uint32_t node = l.rd(LY::P_NODE + s);
		int rc = add(node, s, rec.w[0], pack & 0xffffu);
		if (rc != GA_OK) return rc;
		const int endScore = ew_end(l.rd(LY::P_END + s));
		if (endScore > prevMin + expand) continue;
		rc = pushOut(rec, node, endScore - prevMin + 1);
		if (rc != GA_OK) return rc;
	}
	if (cn == 0) return GA_ASSERTION;                                             // assert(distances.size() > 0) (:1138)
	{ const uint64_t t2 = lap_clock(); st.blaps[1] += t2 - bt; bt = t2; }
	while (heapSize > 0)
	{
		const uint32_t node = l.rd(LY::X_HEAPN);
		const int prio = (int)l.rdh(LY::X_HEAPP, 0);
		if (prio > expand) break;
		heap_pop<N>(l, heapSize);
		if (curNodes.find(l, LY::C_NODE, cn, node) >= 0) continue;                // already at a distance <= prio
		const int ps = prevNodes.find(l, LY::P_NODE, pn, node);
		Rec6 rec;
		loadRec6(node, rec);
		const uint32_t len = rec.w[0];
		int rc = add(node, ps, len, ps >= 0 ? (l.rd(LY::P_PACK + ps) & 0xffffu) : 0u);
		if (rc != GA_OK) return rc;
		rc = pushOut(rec, node, prio + (int)len);
		if (rc != GA_OK) return rc;
	}
	st.cn = cn;
	st.totalCols = totalCols;
	{ const uint64_t t2 = lap_clock(); st.blaps[2] += t2 - bt; bt = t2; }
	return GA_OK;
}

// ---- out-neighbour slots and the processing order: reverse Tarjan emission order over the band subgraph ---------
// (GraphAligner.h:1836-1901, :2360).  On a DAG every node is its own component, emitted when its DFS finishes.
template <int N> GAL_FN int band_order(const GaDevGraph& g, const LaneMem& m, LaneState& st)
{
	typedef Lay<N> LY;
	const Lds& l = m.lds;
	const int cn = st.cn;
	const int pn = st.pn;
	// words 3 .. 11 of the node records (degrees, out- and in-neighbours), the next node's requested while this one is looked up
	uint32_t cur[9], nxt[9];
	auto loadRec9 = [&](int s, uint32_t (&r)[9]) {
		const uint32_t* rec = g_rec(g, l.rd(LY::C_NODE + s));
#pragma unroll
		for (int i = 0; i < 9; i++) r[i] = rec[3 + i];
	};
	for (int i = 0; i < 9; i++) { cur[i] = 0; nxt[i] = 0; }
	uint64_t bt = lap_clock();
	constexpr bool kRegs = NodeRegs<N>::kOn;          // narrow bands: node lists, out-neighbour slots and the DFS state stay in registers
	NodeRegs<N> curNodes, prevNodes;
	curNodes.load(l, LY::C_NODE, cn);
	prevNodes.load(l, LY::P_NODE, pn);
	uint64_t outAll[4] = {0, 0, 0, 0};                // kRegs: 4 out-neighbour slots of 4 bits (15 = none) per node, 4 nodes per word
	// kRegs: is the band one simple path?  the only in-band out-neighbour per node (4 bits each, 15 = none), nodes with an in-band in-neighbour
	uint64_t succAll = ~0ull;
	uint32_t hasPred = 0;
	bool simple = true;
	if (cn > 0) loadRec9(0, cur);
	for (int s = 0; s < cn; s++)
	{
		if (s + 1 < cn) loadRec9(s + 1, nxt);
		uint32_t rec[12];
#pragma unroll
		for (int i = 0; i < 9; i++) { rec[3 + i] = cur[i]; cur[i] = nxt[i]; }
		const uint32_t outDeg = rec[3] >> 16, inDeg = rec[3] & 0xffffu;
		if (outDeg > 4 || inDeg > 4) return GA_PUNT;
		uint32_t slots = 0, inCur = 0, inPrv = 0, nibbles = 0;
#pragma unroll
		for (uint32_t e = 0; e < 4; e++)
		{
			int x = -1, ic = -1, ip = -1;
			if (e < outDeg) x = curNodes.find(l, LY::C_NODE, cn, rec[4 + e]);
			if (e < inDeg) { ic = curNodes.find(l, LY::C_NODE, cn, rec[8 + e]); ip = prevNodes.find(l, LY::P_NODE, pn, rec[8 + e]); }
			slots |= (x < 0 ? kNone : (uint32_t)x) << (8 * e);
			nibbles |= (x < 0 ? 15u : (uint32_t)x) << (4 * e);
			inCur |= (ic < 0 ? kNone : (uint32_t)ic) << (8 * e);
			inPrv |= (ip < 0 ? kNone : (uint32_t)ip) << (8 * e);
		}
		if constexpr (kRegs)
		{
			const uint64_t field = (uint64_t)nibbles << (16 * (s & 3));
#pragma unroll
			for (int w = 0; w < 4; w++) outAll[w] |= (s >> 2) == w ? field : 0ull;
			uint32_t nOut = 0, nIn = 0, only = 15u;
#pragma unroll
			for (uint32_t e = 0; e < 4; e++)
			{
				const uint32_t x = (nibbles >> (4 * e)) & 15u;
				if (x != 15u) { nOut++; only = x; }
				if (((inCur >> (8 * e)) & 0xffu) != kNone) nIn++;
			}
			simple = simple && nOut <= 1 && nIn <= 1;
			succAll = (succAll & ~(15ull << (4 * s))) | ((uint64_t)only << (4 * s));
			hasPred |= nIn ? 1u << s : 0u;
		}
		else
		{
			l.wr(LY::C_OUT + s, slots);
			l.wrb(LY::X_ORD, LY::OB_COLOR + s, 0);
		}
		// where the node's in-neighbours sit in the current / the previous band: read once when the fill starts the node
		// (P_PACK is free by now; C_MIN[s] is only written when node s is finished)
		l.wr(LY::P_PACK + s, inCur);
		l.wr(LY::C_MIN + s, inPrv);
	}
	int emitted = 0;
	{ const uint64_t t2 = lap_clock(); st.blaps[3] += t2 - bt; bt = t2; }
	if constexpr (kRegs)
	{
		// A band that is one simple path (every node at most one in-band out- and in-neighbour, one node without an in-band
		// in-neighbour, all nodes reached from it) has one emission order whatever the search's roots are: a node is emitted when its only
		// way on has been, so the path comes out from its end to its head.  The search below is only run when some lane's band is not that.
		bool path = false;
		uint64_t along = 0;                                 // the path's slots from its head, 4 bits each
		{
			const uint32_t heads = ~hasPred & ((1u << cn) - 1u);
			if (simple && heads != 0 && (heads & (heads - 1u)) == 0)
			{
				int v = __builtin_ctz(heads), k = 0;
				bool ended = false;
				while (k < cn && !ended)
				{
					along |= (uint64_t)v << (4 * k);
					k++;
					const int nx = (int)(succAll >> (4 * v)) & 15;
					ended = nx == 15;
					v = nx;
				}
				path = ended && k == cn;
			}
		}
#ifdef GA_EMULATE
		const bool search = !path;
#else
		const bool search = __ballot(!path) != 0;
#endif
		if (!search)
		{
			for (int k = 0; k < cn; k++) l.wrb(LY::X_ORD, LY::OB_POST + k, (uint32_t)(along >> (4 * (cn - 1 - k))) & 15u);
			{ const uint64_t t2 = lap_clock(); st.blaps[4] += t2 - bt; bt = t2; }
			return GA_OK;
		}
		// the same depth-first search with its state in registers: colour 2 bits per slot, the stack's slots and edge cursors 4 bits per level
		uint32_t color = 0;
		uint64_t stSlot = 0, stCur = 0;
		int sp = 0, root = 0;
		while (true)
		{
			if (sp == 0)
			{
				if (root >= cn) break;
				if (((color >> (2 * root)) & 3u) == 0)
				{
					color |= 1u << (2 * root);
					stSlot = (uint64_t)root; stCur = 0;
					sp = 1;
				}
				root++;
				continue;
			}
			const int t = 4 * (sp - 1);
			const int v = (int)(stSlot >> t) & 15;
			const int c = (int)(stCur >> t) & 15;
			if (c < 4)
			{
				const uint64_t word = (v >> 2) == 0 ? outAll[0] : (v >> 2) == 1 ? outAll[1] : (v >> 2) == 2 ? outAll[2] : outAll[3];
				const uint32_t x = (uint32_t)(word >> (16 * (v & 3) + 4 * c)) & 15u;
				const uint32_t colx = x == 15u ? 2u : (color >> (2 * x)) & 3u;
				if (colx == 2) { stCur += 1ull << t; continue; }
				if (colx == 1) return GA_UNSUPPORTED_CYCLE;
				color |= 1u << (2 * x);
				stSlot = (stSlot & ~(15ull << (t + 4))) | ((uint64_t)x << (t + 4));
				stCur &= ~(15ull << (t + 4));
				sp++;
				continue;
			}
			color = (color & ~(3u << (2 * v))) | (2u << (2 * v));
			l.wrb(LY::X_ORD, LY::OB_POST + emitted, (uint32_t)v);
			emitted++;
			sp--;
			if (sp > 0) stCur += 1ull << (t - 4);
		}
	}
	else
	{
		for (int root = 0; root < cn; root++)
		{
			if (l.rdb(LY::X_ORD, LY::OB_COLOR + root) != 0) continue;
			int sp = 1;
			l.wrb(LY::X_ORD, LY::OB_COLOR + root, 1);
			l.wrb(LY::X_ORD, LY::OB_STSLOT, (uint32_t)root);
			l.wrb(LY::X_ORD, LY::OB_STCUR, 0);
			while (sp > 0)
			{
				const int v = (int)l.rdb(LY::X_ORD, LY::OB_STSLOT + sp - 1);
				const int cur = (int)l.rdb(LY::X_ORD, LY::OB_STCUR + sp - 1);
				if (cur < 4)
				{
					const uint32_t x = (l.rd(LY::C_OUT + v) >> (8 * cur)) & 0xffu;
					const uint32_t color = x == kNone ? 2u : l.rdb(LY::X_ORD, LY::OB_COLOR + (int)x);
					if (color == 2) { l.wrb(LY::X_ORD, LY::OB_STCUR + sp - 1, (uint32_t)(cur + 1)); continue; }
					if (color == 1) return GA_UNSUPPORTED_CYCLE;
					l.wrb(LY::X_ORD, LY::OB_COLOR + (int)x, 1);
					l.wrb(LY::X_ORD, LY::OB_STSLOT + sp, x);
					l.wrb(LY::X_ORD, LY::OB_STCUR + sp, 0);
					sp++;
					continue;
				}
				l.wrb(LY::X_ORD, LY::OB_COLOR + v, 2);
				l.wrb(LY::X_ORD, LY::OB_POST + emitted, (uint32_t)v);
				emitted++;
				sp--;
				if (sp > 0) l.wrb(LY::X_ORD, LY::OB_STCUR + sp - 1, l.rdb(LY::X_ORD, LY::OB_STCUR + sp - 1) + 1);
			}
		}
	}
	{ const uint64_t t2 = lap_clock(); st.blaps[4] += t2 - bt; bt = t2; }
	return GA_OK;
}

// ---- one column step in the reference's bit-vector form ------------------------------------------------------------
// getNextSlice (GraphAligner.h:1349-1427) without the row confirmation only cycles need, then the vertical re-entry of
// calculateNode (:1541-1546) as the cell-wise minimum with the run coming down from the cell above (score `calc - d`):
// with T_r = S_r - r the run is constant and the column loses one unit at every row without VP and one more with VN,
// so the run wins down to the row where d units are lost.
struct Col { uint64_t vp, vn; int before; };

GAL_FN void column_step(Col& c, uint64_t eq, bool noDiag, int calc)
{
	if (noDiag) eq &= ~1ull;
	const int hin = calc - c.before;
	const uint64_t neg = (uint32_t)hin >> 31, pos = (uint32_t)(-hin) >> 31;
	const uint64_t xv = eq | c.vn;
	eq |= neg;
	const uint64_t xh = (((eq & c.vp) + c.vp) ^ c.vp) | eq;
	uint64_t ph = c.vn | ~(xh | c.vp);
	uint64_t mh = c.vp & xh;
	ph = (ph << 1) | pos;
	mh = (mh << 1) | neg;
	c.vp = mh | ~(xv | ph);
	c.vn = ph & xv;
	c.before = calc;
}
GAL_FN void column_reenter(Col& c, int d)
{
	int lost = 0, p = 64;
	bool exact = false;
	uint64_t mz = ~c.vp;
	while (mz)
	{
		const int i = __builtin_ctzll(mz);
		mz &= mz - 1;
		lost += 1 + (int)((c.vn >> i) & 1);
		if (lost >= d) { p = i; exact = lost == d; break; }
	}
	const uint64_t below = p >= 64 ? ~0ull : ((1ull << p) - 1);
	const uint64_t keep = p >= 63 ? 0ull : (~0ull << (p + 1));
	c.vp = below | (c.vp & keep) | ((p < 64 && exact) ? (1ull << p) : 0ull);
	c.vn = c.vn & keep;
	c.before -= d;
}
// cell-wise minimum of two columns over rows j-1 .. j+63 (mergeTwoSlices, WordSlice.h:361-421), row by row
GAL_FN void column_merge(Col& a, const Col& b)
{
	int sa = a.before, sb = b.before;
	int prev = sa < sb ? sa : sb;
	const int outBefore = prev;
	uint64_t ovp = 0, ovn = 0;
	for (int r = 0; r < 64; r++)
	{
		sa += (int)((a.vp >> r) & 1) - (int)((a.vn >> r) & 1);
		sb += (int)((b.vp >> r) & 1) - (int)((b.vn >> r) & 1);
		const int mn = sa < sb ? sa : sb;
		ovp |= (uint64_t)(mn == prev + 1) << r;
		ovn |= (uint64_t)(mn == prev - 1) << r;
		prev = mn;
	}
	a.vp = ovp; a.vn = ovn; a.before = outBefore;
}
GAL_FN int col_end(const Col& c) { return c.before + __builtin_popcountll(c.vp) - __builtin_popcountll(c.vn); }
GAL_FN int col_value(uint64_t vp, uint64_t vn, int before, int row)            // WordSlice::getValue (WordSlice.h:223-229)
{
	const uint64_t mask = row < 63 ? ~(~0ull << (row + 1)) : ~0ull;
	return before + __builtin_popcountll(vp & mask) - __builtin_popcountll(vn & mask);
}

// a store of data nobody reads again soon
#if defined(GA_EMULATE)
#define GAL_NT_STORE(p, v) (*(p) = (v))
#else
#define GAL_NT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif
// (records are 8-byte aligned: three 64-bit words)
template <int R> GAL_FN void rec_store(const LaneMem& m, uint32_t row, const Col& c, uint32_t endWord)
{
	uint64_t* p = (uint64_t*)(m.arena + rec_off<R>(row, m.lane, m.ls));
	p[0] = c.vp; p[1] = c.vn; p[2] = (uint64_t)(uint32_t)c.before | ((uint64_t)endWord << 32);
}
template <int R> GAL_FN void rec_load(const LaneMem& m, uint32_t row, Col& c, uint32_t& endWord)
{
	const uint64_t* p = (const uint64_t*)(m.arena + rec_off<R>(row, m.lane, m.ls));
	c.vp = p[0];
	c.vn = p[1];
	const uint64_t t = p[2];
	c.before = (int)(uint32_t)t;
	endWord = (uint32_t)(t >> 32);
}
// (the traceback's form: no end word -- a requested word nobody looks at is a register the compiler hands out again at once, and
// then has to wait for the request before it may write to it)
template <int R> GAL_FN void rec_load_col(const LaneMem& m, uint32_t row, Col& c)
{
	const uint64_t* p = (const uint64_t*)(m.arena + rec_off<R>(row, m.lane, m.ls));
	c.vp = p[0];
	c.vn = p[1];
	c.before = (int)((const uint32_t*)p)[4];
}

#ifndef GA_EMULATE
// the wave's 8 x 64 records of arena block `block` leave as twelve coalesced 1 KB stores: 16-byte chunk q of the 12 KB block
// belongs to lane q / 12 (rec_off<8>: lane-major inside a block)
template <int LW> GAL_FN void stage_flush(const LaneMem& m, uint32_t chunk)
{
#ifdef GA_EXPERIMENT_NOFLUSH
	return;                                     // (diagnostic builds only: how long the fill takes when its records never leave the LDS)
#endif
	__builtin_amdgcn_wave_barrier();
	// chunk `chunk` of the nodes in hand: lane ln's 8 records (192 B) go to ITS block first + chunk.  Twelve threads write one lane's
	// block (16 B each); thread t serves lanes t / 12, t / 12 + 5, ... -- five lanes per store instruction, every LDS address a constant of
	// the thread plus an immediate.  A lane whose node has fewer chunks is not active any more and its image still holds its LAST chunk:
	// it is written to that last block again (the same bytes); a lane without a node this round points at the spare block (laneBlocks).
	const uint32_t part = m.flushPart, g = m.flushLane;
	const bool worker = m.tid < 60;
#pragma unroll
	for (int j = 0; j < (LW + 4) / 5; j++)
	{
		const uint32_t ln = (worker && g + 5u * (uint32_t)j < (uint32_t)LW) ? g + 5u * (uint32_t)j : 0u;
		const uint64_t* sp = m.stage + ln * 25 + part * 2;
		const uint64_t a = sp[0], b = sp[1];
		const uint32_t blk0 = m.laneBlocks[2 * ln], last = m.laneBlocks[2 * ln + 1] - 1u;
		const uint32_t at16 = (blk0 + (chunk < last ? chunk : last)) * 12u + part;                // in 16-byte units from the arena's start
		uint64_t* d = (uint64_t*)(m.arena + (uint64_t)at16 * 16);
		// (non-temporal: written once and read by the traceback much later -- 31.0 -> 29.7 ms against plain stores; the same hint on
		// the end words, which the next slice reads back, or on any of the loads costs 3-6 ms, profiles/r3_ab_nontemporal.txt)
#ifdef GA_PLAIN_ARENA_STORES
		d[0] = a; d[1] = b;                          // (A/B builds)
#else
		GAL_NT_STORE(d, a); GAL_NT_STORE(d + 1, b);
#endif
	}
	__builtin_amdgcn_wave_barrier();
}
// blocks for the nodes the lanes start together: lane's first block = top + (blocks of the lanes before it); returns the wave's total
GAL_FN uint32_t wave_alloc(uint32_t mine, uint32_t top, uint32_t& first)
{
	uint32_t incl = mine;
	for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64); if ((int)(threadIdx.x & 63) >= off) incl += o; }
	first = top + incl - mine;
	return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}
#endif

template <int N> constexpr bool kEqInLds = NodeRegs<N>::kOn && N >= 8;       // (needs 8 free words per lane: C_OUT is only written by the wide variants)
template <int N> GAL_FN uint64_t eq_from_lds(const Lds& l, int base)
{
	return (uint64_t)l.rd(Lay<N>::C_OUT + 2 * base) | ((uint64_t)l.rd(Lay<N>::C_OUT + 2 * base + 1) << 32);
}
GAL_FN uint64_t eq_for(uint64_t e0, uint64_t e1, uint64_t e2, uint64_t e3, int base)
{
	const uint64_t lo = (base & 1) ? e1 : e0, hi = (base & 1) ? e3 : e2;
	return (base & 2) ? hi : lo;
}
GAL_FN int g_base(const GaDevGraph& g, uint64_t col) { return (int)((g.seq2[col >> 4] >> ((col & 15) * 2)) & 3); }

// ---- fill one slice: every band node in processing order (calculateSlice / calculateNode, GraphAligner.h:2331-2451,
// 1457-1573), one column per step.  The bookkeeping of the virtual row j-1 (forceComponentZeroRow's chain :1939-1944,
// scoreBeforeStart / scoreBeforeExists of getNextSlice :1358-1370, the re-entry test :1541-1546) runs along the node
// with the columns.  A lone wave is bound by instruction issue and by exposed latency, so: the columns of a node go in
// chunks of U whose operands (previous end words, graph bases) are requested one chunk ahead; the next node's graph record
// and first previous end word are requested while this node is computed; and a node entered from the node just finished
// takes that node's last column from registers.
// Inside a node the vertical re-entry needs no merge: the cell above is at most one below the left column's row j-1
// (adjacent cells of the previous slice's last row), so the step simply starts from min(calc, above).
//
// Where the records go.  The lanes of a wave take their k-th node together and step its columns together, so step t of the
// wave produces (up to) one record per lane; those 64 records are row t of the wave's arena.  Stored lane by lane they
// would be 64 partial cache-line writes per store instruction, and the L2's request rate, not bytes, is what such a kernel
// runs into.  Instead a lane drops its record into an LDS image of the current block of 8 rows, and when the block is complete
// the whole wave writes its 12 KB as twelve fully coalesced 1 KB stores.  In memory a lane's 8 columns of a block are
// contiguous (192 B), which is what the traceback wants: it reads 16 columns of one lane at a time.
// (The host emulation stores directly; rows there are simply per lane.)
constexpr int kStageWords64 = 25;            // LDS image: lane stride 200 B (conflict-free 8-byte writes), 8 slots of 3 words
#ifndef GA_EMULATE
GAL_FN uint32_t wave_max(uint32_t v)
{
	for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = v > o ? v : o; }
	return v;
}
#else
GAL_FN uint32_t wave_max(uint32_t v) { return v; }
#endif

template <int N, int U, int LW = 64>
GAL_FN void fill_slice(const GaDevGraph& g, const LaneMem& m, LaneState& st, uint32_t slice, bool active, uint32_t& blockTop, uint32_t capRows, uint32_t capCols)
{
	typedef Lay<N> LY;
	const Lds& l = m.lds;
	const bool j0 = slice == 0;
	// (register copies: the match words must not be picked by an indexed load from the lane state)
	const uint64_t e0 = st.e0, e1 = st.e1, e2 = st.e2, e3 = st.e3;
	const int rawAbove = st.rawAbove;
	const int nRowsP1 = (int)st.nRows + 1;
	int ord = active ? st.cn - 1 : -1;
	int sliceMin = INF, minSlot = -1;
	uint32_t minOffset = 0;
	int status = GA_OK;
	uint32_t tick = blockTop;                     // the arena's next free block of U rows (device: of the wave; emulation: of the lane's own arena)
	// requested ahead for the node at `ord`
	uint32_t nFirstLo = 0, nFirstHi = 0, nPend0 = 0;
	auto request = [&](int o) {
		// (unconditional: a lane without a next node asks for node 0's record and a dummy end word)
		const int s2 = o >= 0 ? (int)l.rdb(LY::X_ORD, LY::OB_POST + o) : 0;
		const uint32_t* rec2 = g_rec(g, o >= 0 ? l.rd(LY::C_NODE + s2) : 0u);
		nFirstLo = rec2[0]; nFirstHi = rec2[1];
		const uint32_t pinfo2 = o >= 0 ? l.rd(LY::C_PINFO + s2) : (kNone << 16);
		nPend0 = m.endPrev[(uint64_t)((pinfo2 >> 16) != kNone ? (pinfo2 & 0xffffu) : 0u) * m.ls];
	};
	request(ord);
	// the node finished last: its last column stays in c / exists
	Col c; c.vp = 0; c.vn = 0; c.before = 0;
	bool exists = false;
	int doneSlot = -1;
	// a record leaves: into the block image (device: slot = the step's place in its chunk) or straight to the arena (emulation)
	auto put = [&](bool on, uint32_t row, int slot, const Col& col, uint32_t endWord) {
#ifdef GA_EMULATE
		(void)slot;
		if (on) rec_store<8>(m, row, col, endWord);
#else
		(void)row;
		if (on)
		{
			uint64_t* sp = m.stage + m.lane * kStageWords64 + slot * 3;
			sp[0] = col.vp; sp[1] = col.vn; sp[2] = (uint64_t)(uint32_t)col.before | ((uint64_t)endWord << 32);
		}
#endif
	};
	// a finished column of THIS slice read back (a node entered from a node that is not the one just finished)
	auto readBack = [&](uint32_t row, Col& col, uint32_t& endWord) {
#ifndef GA_EMULATE
		__threadfence_block();                     // the block was written by other lanes of this wave
#endif
		rec_load<8>(m, row, col, endWord);
	};
#ifdef GA_EMULATE
	while (ord >= 0)
#else
	while (__ballot(ord >= 0))
#endif
	{
		const bool act = ord >= 0;
		const int s = act ? (int)l.rdb(LY::X_ORD, LY::OB_POST + ord) : 0;
		const uint32_t geo = act ? l.rd(LY::C_GEO + s) : 0u, pinfo = act ? l.rd(LY::C_PINFO + s) : (kNone << 16);
		const uint32_t inCur = act ? l.rd(LY::P_PACK + s) : 0xffffffffu, inPrv = act ? l.rd(LY::C_MIN + s) : 0xffffffffu;
		const uint32_t colBase = geo & 0xffffu, len = geo >> 16;
		const bool inPrev = (pinfo >> 16) != kNone;
		const uint32_t pbase = pinfo & 0xffffu;
		const uint64_t firstCol = ((uint64_t)nFirstHi << 32) | nFirstLo;
		const uint32_t pend0raw = nPend0;
		request(ord - 1);
		const bool aboveAlways = j0 && inPrev;                                   // "previousEq" (:1503): raw char ==, not characterMatch
		const uint32_t* pend = m.endPrev + (uint64_t)pbase * m.ls;
		static_assert(U == 8, "a chunk is one block of 8 arena rows");
		const uint32_t nChunks = (wave_max(len) + U - 1) / U;                    // the wave's nodes go in chunks of U columns
		// every lane takes the blocks of U arena rows its own node needs (lanes whose node is shorter, or that have none, take fewer):
		// a node's columns are consecutive rows, and no row is spent on a lane that has nothing to put there
		const uint32_t myBlocks = act ? (len + U - 1) / U : 0u;
		uint32_t firstBlock = tick;
#ifdef GA_EMULATE
		const uint32_t allBlocks = myBlocks;
#else
		const uint32_t allBlocks = wave_alloc(myBlocks, tick, firstBlock);
#endif
		if ((uint64_t)(tick + allBlocks + 1) * U > capRows) { if (act) status = GA_CAP_ARENA; ord = -1; break; }
		const uint32_t t0 = firstBlock * U;                                      // arena row of the node's column 0
#ifndef GA_EMULATE
		__builtin_amdgcn_wave_barrier();
		// (a lane without a node this round points at the spare block behind the arena's rows)
		if (m.tid < LW) { m.laneBlocks[2 * m.lane] = myBlocks ? firstBlock : (capRows >> 3) + 1u; m.laneBlocks[2 * m.lane + 1] = myBlocks ? myBlocks : 1u; }
		__builtin_amdgcn_wave_barrier();
#endif
		// operands of the first chunk (columns 0 .. U-1; column 0 itself comes from the node start below).  Requests are unconditional
		// (lanes without a previous column read a valid dummy row) so that the compiler can count them.
		const uint32_t* pendSafe = (act && inPrev) ? pend : m.endPrev;
		const uint32_t* seqSafe = g.seq2 + (act ? (firstCol >> 4) : 0);
		const uint32_t sh0 = act ? (uint32_t)(firstCol & 15) * 2 : 0u;           // chunks are 8 columns: the base word's phase alternates with firstCol's
		uint32_t pe[U];
		uint64_t bw;
#pragma unroll
		for (int i = 0; i < U; i++) pe[i] = pendSafe[(uint64_t)i * m.ls];
		bw = (uint64_t)seqSafe[0] | ((uint64_t)seqSafe[1] << 32);
		int nodeMin = INF, zero = 0, above2 = 0;
		uint32_t endWord = 0;
		const uint32_t dummyCol = capCols + 32;                                  // where lanes without a column this step send their end word
		auto emit = [&](bool on, uint32_t w, int slot) {
			if (on)
			{
				const int end = col_end(c);
				endWord = ((uint32_t)end << 3) | (exists ? 4u : 0u) | ((uint32_t)(c.vn >> 63) << 1) | (uint32_t)(c.vp >> 63);
				nodeMin = end < nodeMin ? end : nodeMin;
				if (end <= sliceMin) { sliceMin = end; minSlot = s; minOffset = w; }  // the LAST column attaining the minimum (:1551-1559, 2410-2418)
			}
			m.endCur[(uint64_t)(on ? colBase + w : dummyCol) * m.ls] = endWord;
			put(on, t0 + w, slot, c, endWord);
		};
		if (act)
		{
			// ---- row j-1 of column 0 (forceComponentZeroRow for a single acyclic node, :1916-1937) ----
			const int pend0 = inPrev ? ew_end(pend0raw) : INF;
			int zero0 = pend0;
			bool hasIn = false;
			Col left[4];
			bool leftEx[4];
#pragma unroll
			for (int e = 0; e < 4; e++)
			{
				const uint32_t ic = (inCur >> (8 * e)) & 0xffu, ip = (inPrv >> (8 * e)) & 0xffu;
				left[e].vp = 0; left[e].vn = 0; left[e].before = 0; leftEx[e] = false;
				if (ic == kNone && ip == kNone) continue;
				hasIn = true;
				if (ic != kNone)
				{
					if ((int)ic == doneSlot) { left[e] = c; leftEx[e] = exists; }
					else
					{
						// (C_PINFO of a started node holds the arena row of its column 0)
						uint32_t ew;
						readBack(l.rd(LY::C_PINFO + (int)ic) + (l.rd(LY::C_GEO + (int)ic) >> 16) - 1, left[e], ew);
						leftEx[e] = ((ew >> 2) & 1) != 0;
					}
					zero0 = zero0 < left[e].before + 1 ? zero0 : left[e].before + 1;
				}
				if (ip != kNone)
				{
					const int pe1 = ew_end(l.rd(LY::P_END + (int)ip)) + 1;
					zero0 = zero0 < pe1 ? zero0 : pe1;
				}
			}
			const int base0 = (int)(bw >> sh0) & 3;
			const uint64_t eq0 = eq_for(e0, e1, e2, e3, base0);
			const bool aboveEq0 = aboveAlways || (!j0 && rawAbove == base0);
			const bool exists0 = inPrev && pend0 == zero0;                       // scoreBeforeExists from :1989 (scoreEndExists is always true on this path)
			if (!hasIn)
			{
				// source node (:1475-1499): a vertical run from the cell above
				if (j0 && inPrev) { c.vp = ~1ull | (uint64_t)(((eq0 & 1) != 0) ? 0 : 1); c.vn = 0; c.before = pend0; exists = true; }
				else if (inPrev) { c.vp = ~0ull; c.vn = 0; c.before = pend0; exists = true; }
				else { c.vp = ~1ull; c.vn = 0; c.before = nRowsP1; exists = false; }
			}
			else
			{
				// node start: cell-wise minimum over the in-neighbours' last columns advanced one step (getNodeStartSlice :1270-1315)
				bool first = true;
#pragma unroll
				for (int e = 0; e < 4; e++)
				{
					const uint32_t ic = (inCur >> (8 * e)) & 0xffu, ip = (inPrv >> (8 * e)) & 0xffu;
					if (ic == kNone && ip == kNone) continue;
					Col x = left[e];
					uint64_t eq = eq0;
					bool leftExists = leftEx[e];
					if (ic == kNone)
					{
						// neighbour only in the previous band: vertical source column, only row j may match (:1294-1301)
						x.vp = ~0ull; x.vn = 0; x.before = ew_end(l.rd(LY::P_END + (int)ip));
						leftExists = true;
						eq &= 1ull;
					}
					int calc = x.before + 1;
					if (exists0 && ip != kNone)
					{
						const int viaDiag = ew_end2(l.rd(LY::P_END + (int)ip)) + (aboveEq0 ? 0 : 1);
						calc = calc < viaDiag ? calc : viaDiag;
					}
					column_step(x, eq, !(leftExists && ip != kNone), calc);
					if (first) { c = x; first = false; } else column_merge(c, x);
				}
				exists = exists0;
				if (inPrev && c.before > pend0) { column_reenter(c, c.before - pend0); exists = true; }     // vertical re-entry (:1504-1509)
			}
			zero = zero0;
			above2 = ew_end2(pend0raw);
			l.wr(LY::C_PINFO + s, t0);                                           // from now on: where the node's records are
		}
		// ---- the node's columns, U per chunk ----
		for (uint32_t j = 0; j < nChunks; j++)
		{
			const uint32_t w0 = j * U;
			const uint32_t sh = (sh0 + 2 * w0) & 31u;                            // bit position of column w0's base inside bw
			// operands of the next chunk
			uint32_t pe2[U];
			uint64_t bw2;
#pragma unroll
			for (int i = 0; i < U; i++) pe2[i] = pendSafe[(uint64_t)(w0 + U + (uint32_t)i) * m.ls];
			{ const uint32_t* q = seqSafe + (((sh0 >> 1) + w0 + U) >> 4); bw2 = (uint64_t)q[0] | ((uint64_t)q[1] << 32); }
			// (narrow variant: the chunk's match words are read from the LDS table up front, so that no column waits for its own)
			uint64_t eqw[U];
			if constexpr (kEqInLds<N>)
			{
#pragma unroll
				for (int i = 0; i < U; i++) eqw[i] = eq_from_lds<N>(l, (int)(bw >> (sh + 2 * i)) & 3);
			}
#pragma unroll
			for (int i = 0; i < U; i++)
			{
				const uint32_t w = w0 + (uint32_t)i;
				const bool on = act && w < len;
				if (on && w > 0)
				{
					const int base = (int)(bw >> (sh + 2 * i)) & 3;
					const uint32_t peRaw = pe[i];
					const int pv = inPrev ? ew_end(peRaw) : INF;
					const int z1 = zero + 1;
					zero = z1 < pv ? z1 : pv;                                      // :1939-1944
					const bool existsW = inPrev && pv == zero;
					const bool aboveEq = aboveAlways || (!j0 && rawAbove == base);
					int calc = c.before + 1;
					if (existsW) { const int viaDiag = above2 + 1 - (aboveEq ? 1 : 0); calc = calc < viaDiag ? calc : viaDiag; }    // :1361-1370
					const bool reenter = calc > pv;                                // :1541-1546 (pv = INF when the node is new to the band)
					column_step(c, kEqInLds<N> ? eqw[i] : eq_for(e0, e1, e2, e3, base), !exists, reenter ? pv : calc);
					exists = reenter || existsW;
					above2 = ew_end2(peRaw);
				}
				emit(on, w, i);
			}
#ifndef GA_EMULATE
			stage_flush<LW>(m, j);                                                                // the chunk's blocks of U rows are complete
#endif
#pragma unroll
			for (int i = 0; i < U; i++) pe[i] = pe2[i];
			bw = bw2;
		}
		tick += allBlocks;
		if (act)
		{
			if (c.before != zero) { status = GA_ASSERTION; ord = -1; }            // assert(newEnd.scoreBeforeStart == oldEnd.scoreBeforeStart) (:2385)
			else
			{
				l.wr(LY::C_MIN + s, (uint32_t)nodeMin);
				l.wr(LY::C_END + s, endWord);
				doneSlot = s;
				ord--;
			}
		}
	}
	blockTop = tick;
	if (active)
	{
		st.sliceMin = sliceMin; st.minSlot = minSlot; st.minOffset = minOffset;
		if (status != GA_OK) st.status = status;
	}
}

// ---- the slice loop of one lane, cut at the points where the wave decides something together ---------------------
template <int N> GAL_FN void lane_begin(const GaLanesLaunch& L, const LaneMem& m, LaneState& st, uint32_t jobIndex, bool hasJob)
{
	typedef Lay<N> LY;
	st.job = jobIndex;
	st.status = hasJob ? GA_OK : GA_NOT_RUN;
	st.live = false;
	st.pn = 0; st.cn = 0; st.prevMin = 0; st.totalCols = 0; st.rowBase = 0;
	st.e0 = st.e1 = st.e2 = st.e3 = 0; st.rawAbove = 7;
	st.sliceMin = 0; st.minSlot = 0; st.minOffset = 0;
	st.nPushed = 0; st.nRun = 0; st.maxBandNodes = 0; st.kept = 0; st.nColumns = 0; st.rampUntil = 0; st.useRamp = false;
	st.nRows = 0; st.numSlices = 0; st.seedNode = 0; st.traceRows = 0; st.eq = L.eq;
	st.logCorrect = 0; st.logWrong = 0;
	for (int i = 0; i < 8; i++) st.laps[i] = 0;
	for (int i = 0; i < 5; i++) st.blaps[i] = 0;
	if (!hasJob) return;
	const GaJob job = L.jobs[jobIndex];
	st.nRows = job.n_rows;
	st.numSlices = job.n_rows / W;
	st.seedNode = job.seed_node;
	st.traceRows = job.trace_rows;
	st.eq = L.eq + (job.rows_off / W) * 5;
	st.logCorrect = L.hmm->init_correct;
	st.logWrong = L.hmm->init_wrong;
	// the traceback asserts samplingFrequency > 1 (:906) for fewer than four slices: such reads take the ladder, which reports it
	if (st.numSlices < 4 || st.numSlices > L.max_slices) { st.status = GA_PUNT; return; }
	const uint32_t seedLen = g_rec(L.graph, job.seed_node)[2];
	if (seedLen >= kCutoff) { st.status = GA_UNSUPPORTED_BAND; return; }       // the second slice's band is this node: sparse method in the reference
	if (seedLen > L.cap_cols || seedLen > 0xffffu) { st.status = GA_CAP_COLS; return; }
	// initial slice: the whole seed node at score 0 (GraphAligner.h:2945-2960)
	m.lds.wr(LY::P_NODE, job.seed_node);
	m.lds.wr(LY::P_END, 0u | 4u);
	m.lds.wr(LY::P_PACK, 0);
	for (uint32_t c = 0; c < seedLen; c++) m.endPrev[(uint64_t)c * m.ls] = 4u;       // score 0, scoreEndExists
	st.pn = 1;
	st.live = true;
}

// band of the next slice; returns with st.totalCols set (0 when the lane is not computing this slice)
template <int N> GAL_FN void lane_band(const GaLanesLaunch& L, const LaneMem& m, LaneState& st, uint32_t slice)
{
	st.totalCols = 0;
	if (!st.live) return;
	if (slice >= st.numSlices) { st.live = false; return; }
	// slice 0 always runs at the ramp width because rampUntil(0) >= slice(0) (:2603,2612)
	st.useRamp = st.rampUntil >= slice;
	const int bandwidth = st.useRamp ? L.ramp_bw : L.initial_bw;
	const uint64_t* e = st.eq + (uint64_t)slice * 5;
	st.e0 = e[0]; st.e1 = e[1]; st.e2 = e[2]; st.e3 = e[3];
	const uint32_t meta = (uint32_t)e[4];
	st.rawAbove = slice > 0 ? (int)((uint32_t)e[-1] & 7u) : 7;
	int rc = project_band<N>(L.graph, m, st, bandwidth);
	if (rc == GA_OK && st.totalCols > L.cap_cols) rc = GA_CAP_COLS;
	if (rc == GA_OK) rc = band_order<N>(L.graph, m, st);
	if (rc == GA_OK && (meta & 8u)) rc = GA_ASSERTION;                            // characterMatch default branch (:2104-2106)
	if (rc != GA_OK) { st.status = rc; st.live = false; st.totalCols = 0; }
	if constexpr (kEqInLds<N>)
	{
		// the slice's four match words as a table (the words of C_OUT, free in the narrow variant once the band is put together): the
		// fill picks a column's word by its base with one LDS read instead of a chain of selects
		const Lds& l = m.lds;
#pragma unroll
		for (int k = 0; k < 4; k++) { l.wr(Lay<N>::C_OUT + 2 * k, (uint32_t)e[k]); l.wr(Lay<N>::C_OUT + 2 * k + 1, (uint32_t)(e[k] >> 32)); }
	}
}

// after the fill: HMM step, stop test, bookkeeping, and the slice becomes the state (getSqrtSlices, :2571-2856)
template <int N> GAL_FN void lane_end_slice(const GaLanesLaunch& L, LaneMem& m, LaneState& st, uint32_t slice)
{
	typedef Lay<N> LY;
	if (!st.live) return;
	const Lds& l = m.lds;
	if (st.status != GA_OK) { st.live = false; return; }
	if (st.sliceMin < st.prevMin) { st.status = GA_ASSERTION; st.live = false; return; }        // :2469
	st.nRun++;
	st.nColumns += st.totalCols;
	st.maxBandNodes = st.maxBandNodes > (uint32_t)st.cn ? st.maxBandNodes : (uint32_t)st.cn;
	// ---- HMM step (AlignmentCorrectnessEstimation.cpp:71-89): additions and comparisons only ----
	const GaHmmTables& hmm = *L.hmm;
	const int mism = st.sliceMin - st.prevMin;
	if (mism > 64) { st.status = GA_ASSERTION; st.live = false; return; }
	const double cc = st.logCorrect + hmm.c2c, fc = st.logWrong + hmm.f2c, cf = st.logCorrect + hmm.c2f, ff = st.logWrong + hmm.f2f;
	const bool correctFromCorrect = cc >= fc;
	const bool falseFromCorrect = cf >= ff;
	const double newCorrect = (cc > fc ? cc : fc) + hmm.correct_mult[mism];
	const double newWrong = (cf > ff ? cf : ff) + hmm.wrong_mult[mism];
	const bool currentlyCorrect = newCorrect > newWrong;
	if (!correctFromCorrect) { st.live = false; return; }                                         // :2640-2647 (this slice is not kept)
	const bool rampPossible = L.ramp_bw > L.initial_bw;
	if (!currentlyCorrect && st.rampUntil < slice && rampPossible) { st.status = GA_UNSUPPORTED_RAMP; st.live = false; return; }   // the redo (:2648-2719) is the ladder's
	// the slice is kept: header and node list for the traceback
	const int cn = st.cn;
	uint32_t* h = m.hdr + (uint64_t)slice * kHdrWords * m.ls;
	h[0] = (uint32_t)cn; h[1 * m.ls] = st.totalCols; h[2 * m.ls] = (uint32_t)st.sliceMin; h[3 * m.ls] = (uint32_t)st.minSlot; h[4 * m.ls] = st.minOffset;
	h[5 * m.ls] = (uint32_t)((currentlyCorrect ? 1 : 0) | (falseFromCorrect ? 2 : 0));
	h[6 * m.ls] = 0;                                                                   // (node rows below are absolute)
	uint32_t* sn = m.snodes + (uint64_t)slice * 2 * N * m.ls;
	for (int k = 0; k < cn; k++)
	{
		const uint32_t node = l.rd(LY::C_NODE + k), geo = l.rd(LY::C_GEO + k);
		sn[(uint64_t)(2 * k) * m.ls] = node;
		sn[(uint64_t)(2 * k + 1) * m.ls] = l.rd(LY::C_PINFO + k);                       // arena row of the node's column 0
		// ... and the slice becomes the state the next one is computed from
		l.wr(LY::P_NODE + k, node);
		l.wr(LY::P_END + k, l.rd(LY::C_END + k));
		const uint32_t rel = l.rd(LY::C_MIN + k) - (uint32_t)st.sliceMin;
		if (rel > 0xffffu) { st.status = GA_PUNT; st.live = false; return; }
		l.wr(LY::P_PACK + k, (geo & 0xffffu) | (rel << 16));
	}
	st.nPushed++;
	st.logCorrect = newCorrect; st.logWrong = newWrong;
	st.pn = cn;
	st.prevMin = st.sliceMin;
	uint32_t* t = m.endPrev; m.endPrev = m.endCur; m.endCur = t;
}

#ifndef GA_EMULATE
// claim `bytes` of the pool for every lane that calls (the callers are the active lanes of one wave, converged): true + offset when it fit
GAL_FN uint64_t claim_cas(uint64_t* top, uint64_t cap, uint64_t bytes)
{
	unsigned long long seen = *(volatile unsigned long long*)top;
	while (seen + bytes <= cap)
	{
		const unsigned long long prev = atomicCAS((unsigned long long*)top, seen, seen + (unsigned long long)bytes);
		if (prev == seen) return seen;
		seen = prev;
	}
	return ~0ull;
}
GAL_FN bool claim_for_wave(uint64_t* top, uint64_t cap, uint64_t bytes, uint64_t& at)
{
	const uint64_t active = __ballot(1);
	const int lane = (int)(threadIdx.x & 63);
	// exclusive prefix of `bytes` over the active lanes, and the wave's total
	uint64_t before = 0, total = 0;
	for (uint64_t m = active; m; m &= m - 1)
	{
		const int src = __builtin_ctzll(m);
		const uint64_t b = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(bytes >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)bytes, src, 64);
		if (src < lane) before += b;
		total += b;
	}
	const int leader = __builtin_ctzll(active);
	uint64_t base = ~0ull;
	if (lane == leader) base = claim_cas(top, cap, total);
	base = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(base >> 32), leader, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, leader, 64);
	if (base != ~0ull) { at = base + before; return true; }
	// the wave's total does not fit: lane by lane, whatever still does
	bool ok = false;
	for (uint64_t m = active; m; m &= m - 1)
	{
		if (lane == __builtin_ctzll(m)) { const uint64_t a = claim_cas(top, cap, bytes); if (a != ~0ull) { at = a; ok = true; } }
	}
	return ok;
}
#endif

// ---- traceback (getTraceFromTable :894-957 with pickBacktracePredecessor :493-591) and the job's output ----------
template <int N> GAL_FN void lane_finish(const GaLanesLaunch& L, const LaneMem& m, LaneState& st, bool hasJob)
{
	typedef Lay<N> LY;
	if (!hasJob) return;
	const GaDevGraph& g = L.graph;
	const Lds& l = m.lds;
	GaJobOut out;
	out.status = st.status; out.score = 0x7fffffff; out.n_valid = 0; out.n_run = st.nRun; out.trace_len = 0; out.max_band_nodes = st.maxBandNodes;
	out.n_columns = st.nColumns; out.trace_off = 0; out.start_node = 0; out.start_offset = 0; out.start_row = 0; out.reserved2 = 0;
	out.n_node_steps = 0; out.reserved3 = 0;
	for (int i = 0; i < 8; i++) out.stamps[i] = 0;
	int status = st.status;
	// ---- drop the wrongly aligned tail (removeWronglyAlignedEnd, :2554-2569) ----
	uint32_t kept = st.nPushed;
	if (status == GA_OK && kept > 0)
	{
		bool ok = (m.hdr[((uint64_t)(kept - 1) * kHdrWords + 5) * m.ls] & 1u) != 0;
		while (!ok)
		{
			kept--;
			if (kept == 0) break;
			ok = (m.hdr[((uint64_t)(kept - 1) * kHdrWords + 5) * m.ls] & 2u) != 0;
		}
	}
	out.n_valid = status == GA_OK ? kept : 0;
	uint32_t len = 0, nodeSteps = 0;
	if (status == GA_OK && kept > 0)
	{
		const int big = (int)st.nRows;                                            // getValueOrMax default = sequence.size()
		uint32_t sIdx = kept - 1;
		uint32_t nN = 0, pN = 0, curRow = 0, prvRow = 0;
		uint32_t aN = 0, aRow = 0;                                                // header of the slice two above, requested one slice change ahead
		int tCN = LY::T_CN, tCB = LY::T_CB, tPN = LY::T_PN, tPB = LY::T_PB;
		auto loadHeader = [&](uint32_t sl, uint32_t& count, uint32_t& rowBase) {
			const uint32_t* h = m.hdr + (uint64_t)sl * kHdrWords * m.ls;
			count = h[0];
			rowBase = h[6 * m.ls];
		};
		auto loadTable = [&](int tn, int tb, uint32_t sl, uint32_t count) {
			const uint32_t* sn = m.snodes + (uint64_t)sl * 2 * N * m.ls;
			for (uint32_t k = 0; k < count; k += 4)
			{
				uint32_t a[8];
#pragma unroll
				for (int i = 0; i < 8; i++) a[i] = sn[(uint64_t)(2 * k + (uint32_t)i) * m.ls];      // (rows past `count` are inside the plane)
#pragma unroll
				for (int i = 0; i < 4; i++) if (k + (uint32_t)i < count) { l.wr(tn + (int)k + i, a[2 * i]); l.wr(tb + (int)k + i, a[2 * i + 1]); }
			}
		};
		auto loadTableFrom = [&](int tn, int tb, uint32_t sl, uint32_t count, uint32_t from) {
			const uint32_t* sn = m.snodes + (uint64_t)sl * 2 * N * m.ls;
			for (uint32_t k = from; k < count; k++) { const uint32_t a0 = sn[(uint64_t)(2 * k) * m.ls], a1 = sn[(uint64_t)(2 * k + 1) * m.ls]; l.wr(tn + (int)k, a0); l.wr(tb + (int)k, a1); }
		};
		// what the traceback keeps of a node's graph record: first column, in-degree, first four in-neighbours with their lengths
		struct NodeRec { uint64_t firstCol; uint32_t inDeg; uint32_t nb[4], nbLen[4]; };
		auto loadRec = [&](uint32_t n, NodeRec& nr) {
			const uint32_t* rec = g_rec(g, n);
			nr.firstCol = ((uint64_t)rec[1] << 32) | rec[0];
			nr.inDeg = rec[3] & 0xffffu;
#pragma unroll
			for (int k = 0; k < 4; k++) { nr.nb[k] = rec[8 + k]; nr.nbLen[k] = rec[12 + k]; }
		};
		loadHeader(sIdx, nN, curRow);
		if (sIdx > 0) loadHeader(sIdx - 1, pN, prvRow);
		if (sIdx > 1) loadHeader(sIdx - 2, aN, aRow);
		loadTable(tCN, tCB, sIdx, nN);
		if (sIdx > 0) loadTable(tPN, tPB, sIdx - 1, pN);
		const uint32_t* h = m.hdr + (uint64_t)sIdx * kHdrWords * m.ls;
		out.score = (int32_t)h[2 * m.ls];
		uint32_t node = l.rd(tCN + (int)h[3 * m.ls]);
		int offset = (int)h[4 * m.ls];             // (a column of `node`; below 0 while a tight step has gone into the in-neighbour, see below)
		uint32_t row = sIdx * W + (W - 1);
		out.start_node = node; out.start_offset = (uint32_t)offset; out.start_row = row;
		// node runs instead of moves (L.emit_runs): a run is opened at the first cell met in a node (its last cell on the read) once
		// the trace is below the rows that do not count, and written out with its first cell when the path leaves the node
		bool tracing = true;
		const bool emitRuns = L.emit_runs != 0;
		const uint32_t traceRows = st.traceRows;
		const uint32_t capRunWords = (L.cap_moves + 3) / 4 + 8;
		uint32_t nRuns = 0, runLastOff = (uint32_t)offset, runLastRow = row;
		bool started = row < traceRows;
		auto emitRun = [&](uint32_t n, uint32_t firstOff, uint32_t firstRow) {
			if (5 * (nRuns + 1) > capRunWords) { status = GA_CAP_TRACE; tracing = false; return; }
			uint32_t* d = m.moves + (uint64_t)(5 * nRuns) * m.ls;
			d[0] = n; d[1 * m.ls] = firstOff; d[2 * m.ls] = firstRow; d[3 * m.ls] = runLastOff; d[4 * m.ls] = runLastRow;
			nRuns++;
		};
		uint64_t e[4];
		auto loadEq = [&](uint32_t sl) { const uint64_t* q = st.eq + (uint64_t)sl * 5; e[0] = q[0]; e[1] = q[1]; e[2] = q[2]; e[3] = q[3]; };
		loadEq(sIdx);
		// the record of column (n, off) in the slice whose tables are (tn, tb); false when the node is not in that slice
		auto recordIn = [&](int tn, int tb, uint32_t count, uint32_t rowBase, uint32_t n, uint32_t off, Col& c) -> bool {
			const int sl = find_in(l, tn, (int)count, n);
			c.vp = 0; c.vn = 0; c.before = 0;
			if (sl < 0) return false;
			rec_load_col<8>(m, rowBase + l.rd(tb + sl) + off, c);
			return true;
		};
		uint32_t pack = 0;
#ifdef GA_DEBUG_SITE
		uint32_t site = 0;                            // diagnostic builds: which of the traceback's assertions fired (summed per site in the pass statistics)
#define GAL_SITE(n) site = (n)
#else
#define GAL_SITE(n)
#endif
		// Most steps stay inside a slice and move along a chain of nodes.  Those run in a tight loop on two records held in registers
		// (the current column and the one to its left) that are fed from a window of kWin consecutive columns kept in LDS.  The window
		// holds columns of the current node and -- when that node has exactly one in-neighbour, and the neighbour is in the slice --
		// below column 0 the last columns of that in-neighbour: a step from column 0 into it is then the same arithmetic as a step inside
		// a node (pickBacktracePredecessor with one in-neighbour, :512-556, tries the same three cells in the same order), and the
		// tight loop takes it; the lane's coordinates follow at the next round.  Everything else is a ROUND of the wave after at most
		// kBurst tight iterations: lanes that stand at a slice's first row, or at column 0 of a node whose way back is not one
		// neighbour in the slice, take a DECISION (the general step: all in-neighbours, the slice above, the reference's asserts); every
		// lane then has its window topped up.  Rare per lane is not rare per wave, so a lane that needs a decision WAITS while the others
		// step, and the decision code runs for all waiting lanes at once; lanes meet a slice's first row at about the same time (a row
		// per step, except for the few steps to the left), which is the one thing left that makes a lane wait.  (The order in which
		// lanes step does not change what any of them computes; the host emulation simply runs each lane on its own.)
		constexpr int kWin = Lay<N>::T_WINCOLS;
		constexpr int tWIN = Lay<N>::T_WIN;
#ifndef GAL_BURST_SLACK
#define GAL_BURST_SLACK 4
#endif
		constexpr int kBurst = kWin - GAL_BURST_SLACK;
		Col q0, q1;
		q0.vp = q0.vn = 0; q0.before = 0; q1 = q0;
		bool needSetup = true;
		uint32_t slotRow = 0, recNode = 0xffffffffu, inDeg = 0;
		uint32_t nb[4] = {0, 0, 0, 0}, nbLen[4] = {0, 0, 0, 0};
		uint64_t firstCol = 0;
		NodeRec nx = NodeRec();                       // the graph record of nxNode: the first in-neighbour of the node the lane was in when it was requested
		uint32_t nxNode = 0xffffffffu, nxSlotRow = 0; // ... and the arena row of its column 0 in the current slice, when the window reaches into it
		uint64_t wbases = 0;                          // the graph bases of the window's columns, 2 bits each from wLo up
		int wLo = 1, wHi = 0;                         // columns the window holds: [wLo, wHi] of `node`, columns below 0 being the in-neighbour's
		                                              // (column c < 0 = its column nbLen[0] + c); empty when wLo > wHi
		bool crossed = false, xStarted = false;       // a tight step went from column 0 into the in-neighbour: the rows around that step, and
		uint32_t xRowBefore = 0, xRowAfter = 0;       // whether a run was open
		uint32_t capMoves = L.cap_moves;
		uint32_t roundsLeft = 2 * L.cap_moves + 4096;
#ifndef GA_EMULATE
		asm volatile("" : "+v"(capMoves));            // (kept in a vector register: as a scalar it is spilled with the launch block and read back lane by lane in every step)
#endif
		auto winRead = [&](int o, Col& c) {
			const int at = tWIN + (o - wLo) * 5;
			c.vp = ((uint64_t)l.rd(at + 1) << 32) | l.rd(at);
			c.vn = ((uint64_t)l.rd(at + 3) << 32) | l.rd(at + 2);
			c.before = (int)l.rd(at + 4);
		};
		auto putMove = [&](int res, int via) {
			if (emitRuns) return;
			const uint32_t code = res == 1 ? GA_MOVE_LEFT : res == 2 ? GA_MOVE_DIAG : GA_MOVE_UP;
			pack |= (code | (res == 3 ? 0u : ((uint32_t)via << 2))) << (8 * (len & 3));
			if ((len & 3) == 3) { m.moves[(uint64_t)(len >> 2) * m.ls] = pack; pack = 0; }
			len++;
		};
		// 32 bases from graph column `col` on, 2 bits each
		auto bases32 = [&](uint64_t col, uint32_t w0, uint32_t w1, uint32_t w2) -> uint64_t {
			const uint32_t sh = 2 * (uint32_t)(col & 15);
			return (((uint64_t)w0 | ((uint64_t)w1 << 32)) >> sh) | (sh ? (uint64_t)w2 << (64 - sh) : 0ull);
		};
#ifdef GA_EMULATE
#define GAL_ANY(x) (x)
#else
#define GAL_ANY(x) (__ballot(x) != 0)
#endif
		bool firstRound = true;                       // the first round sets the lane up (no lane has a window yet)
		uint64_t lapT = lap_clock();
		while (GAL_ANY(tracing))
		{
			const bool first = firstRound;
			firstRound = false;
			{ const uint64_t t2 = lap_clock(); st.laps[1] += t2 - lapT; lapT = t2; }
			// ---- tight steps: inside the slice (r > 0), both columns in the window ----
			// (the row masks and the current cell's score are carried from step to step: a step costs one cell value of the left column)
			if (!first)
			{
				int r = (int)(row - sIdx * W);
				uint64_t mR = r < 0 ? 0ull : r < 63 ? ~(~0ull << (r + 1)) : ~0ull;      // rows 0 .. r
				int here = q0.before + __builtin_popcountll(q0.vp & mR) - __builtin_popcountll(q0.vn & mR);
				// (straight-line, predicated: a lane that cannot step keeps its state through the selects; the column after next is
				// requested from the window one step ahead and only looked at when the step has moved a column left)
				Col q2;
				winRead(((offset >= wLo + 2) & (offset <= wHi)) ? offset - 2 : wLo, q2);
				for (int it = 0; it < kBurst; it++)
				{
					const bool fast = tracing & (r > 0) & (offset > wLo) & (offset <= wHi) & (len + 8 < capMoves);       // (& not &&: no branches)
					if (!GAL_ANY(fast)) break;
#ifdef GA_STAMPS
					st.laps[7] += 1ull << 32;
#endif
					// the three cells a step can go to: left (row r of the left column), diagonal (its row r - 1), up (row r - 1 of this column);
					// row r - 1 of a column is its row r minus what the column's vertical bit r says
					const int sr = r & 63;
					const int horizontal = q1.before + __builtin_popcountll(q1.vp & mR) - __builtin_popcountll(q1.vn & mR);
					const int diagonal = horizontal - (int)((uint32_t)(q1.vp >> sr) & 1u) + (int)((uint32_t)(q1.vn >> sr) & 1u);
					const int up = here - (int)((uint32_t)(q0.vp >> sr) & 1u) + (int)((uint32_t)(q0.vn >> sr) & 1u);
					const int base = (int)(wbases >> (2 * ((offset - wLo) & 31))) & 3;
					const int want = here - 1 + (int)((e[base] >> sr) & 1);            // the diagonal cell's score if the step is diagonal
					const bool left = horizontal == here - 1;
					const bool diag = !left & (diagonal == want);
					// the reference's asserts on the way (:557-588): a neighbour below what the recurrence allows, or no predecessor at all
					const bool bad = (horizontal < here - 1) | (!left & (diagonal < want)) | (!left & !diag & (up != here - 1));
					const bool ok = fast & !bad;
					status = fast & bad ? GA_ASSERTION : status;
#ifdef GA_DEBUG_SITE
					site = fast & bad ? ((horizontal < here - 1) ? 1 : (!left & (diagonal < want)) ? 2 : 3) : site;
#endif
					tracing = tracing & !(fast & bad);
					const bool colMove = ok & (left | diag), rowMove = ok & !left;
					const bool crossNow = colMove & (offset == 0);                      // into the in-neighbour (only a window that reaches below 0 gets here)
					xRowBefore = crossNow ? row : xRowBefore;
					xStarted = crossNow ? started : xStarted;
					crossed = crossed | crossNow;
					here = ok ? (left ? horizontal : diag ? diagonal : up) : here;
					row -= rowMove ? 1u : 0u;
					r -= rowMove ? 1 : 0;
					mR = rowMove ? mR >> 1 : mR;
					xRowAfter = crossNow ? row : xRowAfter;
					offset -= colMove ? 1 : 0;
					q0.vp = colMove ? q1.vp : q0.vp; q0.vn = colMove ? q1.vn : q0.vn; q0.before = colMove ? q1.before : q0.before;
					q1.vp = colMove ? q2.vp : q1.vp; q1.vn = colMove ? q2.vn : q1.vn; q1.before = colMove ? q2.before : q1.before;
					winRead(((offset >= wLo + 2) & (offset <= wHi)) ? offset - 2 : wLo, q2);
					if (emitRuns)
					{
						// (the first cell below the rows that do not count opens the first run)
						const bool begin = ok & !started & (row < traceRows);
						runLastOff = begin ? (uint32_t)offset : runLastOff;
						runLastRow = begin ? row : runLastRow;
						started = started | begin;
					}
					else if (ok) putMove(left ? 1 : diag ? 2 : 3, 0);
				}
			}
			{ const uint64_t t2 = lap_clock(); st.laps[0] += t2 - lapT; lapT = t2; }
			// ---- the round, for every lane still tracing ----
			// Two round trips to HBM per round of the wave: everything a lane has to see before it can decide (its window when the step
			// left it, the in-neighbours' last columns at a node's first column, the columns of the slice above at a slice's first row,
			// and the record of the first in-neighbour, where the path most likely goes) is requested together; after the decision, the
			// new window, the tables of a new slice and the base words go together again.  Lanes that need nothing request a row that
			// is always there: requests are wave-wide instructions either way, and a request behind a branch that only some lanes
			// take makes the compiler wait for everything outstanding where the branch ends.  (The decision part as a whole sits behind a
			// wave-uniform branch: most rounds only top the windows up.)
#ifdef GA_STAMPS
			st.laps[7] += 1;
#endif
			if (tracing)
			{
				const uint64_t g0 = lap_clock();
				if (row == 0xffffffffu) { tracing = false; continue; }               // reached the row before the first one
				if (len + 8 >= capMoves) { status = GA_CAP_TRACE; tracing = false; continue; }
				if (roundsLeft-- == 0) { status = GA_PUNT; tracing = false; continue; }          // (every round moves its lane; the bound is what ends the loop should one ever not)
				const uint32_t safeRow = 0;
				// a tight step took the lane into the in-neighbour: its coordinates follow (what the decision below does for a step it takes itself)
				if (crossed)
				{
					const uint32_t oldNode = node, L0 = nbLen[0];
					node = nb[0]; offset += (int)L0; wLo += (int)L0; wHi += (int)L0;
					nodeSteps++;
					slotRow = nxSlotRow;
					firstCol = nx.firstCol; inDeg = nx.inDeg;                           // (a window only reaches into a neighbour whose record is here)
#pragma unroll
					for (int k = 0; k < 4; k++) { nb[k] = nx.nb[k]; nbLen[k] = nx.nbLen[k]; }
					recNode = node;
					crossed = false;
					// (the run of the node left is written AFTER the lane's state has moved, and a run that does not fit ends the lane below:
					// with `if (!tracing) continue;` between the two, hipcc 7.2 kept the OLD nb[] / nbLen[] for every lane that had a run to
					// write -- profiles/r3_miscompile_normalization_isa.txt)
					if (emitRuns)
					{
						if (xStarted) { emitRun(oldNode, 0, xRowBefore); runLastOff = L0 - 1; runLastRow = xRowAfter; }
						else if (started) runLastOff += L0;                              // (the run was opened in the new node, at a column counted from the old one)
					}
				}
				if (!tracing) continue;
				if (inDeg > 4) { status = GA_PUNT; tracing = false; continue; }
				const int r = (int)(row - sIdx * W);
				// can the window reach into the in-neighbour?  one in-neighbour, not the node itself, its record here, its columns in this slice
				const bool chained = !first && inDeg == 1 && nb[0] != node && nxNode == nb[0];
				const bool decision = !first && !(r > 0 && (offset > 0 || (chained && find_in(l, tCN, (int)nN, nb[0]) >= 0)));
				bool changed = false;
				NodeRec spec = NodeRec();
				uint32_t specNode = 0xffffffffu;
				if (GAL_ANY(decision))
				{
				if (decision)
				{
				const bool needWin = !(offset >= wLo && offset <= wHi && (offset == 0 || offset > wLo));
				const bool atStart = offset == 0;
				const bool atTop = r == 0 && sIdx > 0;
				// (1) the window around the current column
				Col wc[kWin];
				uint32_t sq0 = 0, sq1 = 0, sq2 = 0;
				const int nLoA = offset >= kWin - 1 ? offset - (kWin - 1) : 0;
				{
#pragma unroll
					for (int i = 0; i < kWin; i++) rec_load_col<8>(m, needWin && nLoA + i <= offset ? slotRow + (uint32_t)(nLoA + i) : safeRow, wc[i]);
					const uint64_t col = needWin ? firstCol + (uint64_t)nLoA : 0ull;
					const uint32_t* q = g.seq2 + (col >> 4);
					sq0 = q[0]; sq1 = q[1]; sq2 = q[2];
				}
				// (2) the last columns of the in-neighbours in this slice, and the graph record of the first of them
				Col nc[4];
#pragma unroll
				for (int k = 0; k < 4; k++) { nc[k].vp = nc[k].vn = 0; nc[k].before = 0; }
				bool nIn[4] = {false, false, false, false};
				{
					uint32_t nbRow[4];
#pragma unroll
					for (int k = 0; k < 4; k++)
					{
						nbRow[k] = safeRow;
						if (atStart && (uint32_t)k < inDeg)
						{
							const int sl = find_in(l, tCN, (int)nN, nb[k]);
							if (sl >= 0) { nIn[k] = true; nbRow[k] = curRow + l.rd(tCB + sl) + nbLen[k] - 1; }
						}
					}
#pragma unroll
					for (int k = 0; k < 4; k++) rec_load_col<8>(m, nbRow[k], nc[k]);
					if (atStart && inDeg > 0 && nb[0] != nxNode) specNode = nb[0];
					loadRec(specNode != 0xffffffffu ? specNode : node, spec);
				}
				// (3) at a slice's first row: this column (U) and the column a diagonal step would reach (A) in the slice above
				Col ca, cu;
				ca.vp = ca.vn = 0; ca.before = 0; cu = ca;
				bool aIn = false, uIn = false;
				{
					uint32_t rowA = safeRow, rowU = safeRow;
					if (atTop)
					{
						const int slU = find_in(l, tPN, (int)pN, node);
						if (slU >= 0) { uIn = true; rowU = prvRow + l.rd(tPB + slU) + (uint32_t)offset; }
						if (offset > 0) { aIn = uIn; rowA = uIn ? rowU - 1 : safeRow; }
						else if (inDeg > 0)
						{
							const int slA = find_in(l, tPN, (int)pN, nb[0]);
							if (slA >= 0) { aIn = true; rowA = prvRow + l.rd(tPB + slA) + nbLen[0] - 1; }
						}
					}
					rec_load_col<8>(m, rowA, ca);
					rec_load_col<8>(m, rowU, cu);
				}
				// ---- everything has arrived: the window first ----
				if (needWin)
				{
					wLo = nLoA; wHi = offset;
					wbases = bases32(firstCol + (uint64_t)wLo, sq0, sq1, sq2);
#pragma unroll
					for (int i = 0; i < kWin; i++)
					{
						if (wLo + i <= wHi)
						{
							const int at = tWIN + i * 5;
							l.wr(at, (uint32_t)wc[i].vp); l.wr(at + 1, (uint32_t)(wc[i].vp >> 32)); l.wr(at + 2, (uint32_t)wc[i].vn); l.wr(at + 3, (uint32_t)(wc[i].vn >> 32)); l.wr(at + 4, (uint32_t)wc[i].before);
						}
					}
					winRead(offset, q0);
					if (offset > wLo) winRead(offset - 1, q1);
				}
				const int here = col_value(q0.vp, q0.vn, q0.before, r);
				if (emitRuns && !started && row < traceRows) { started = true; runLastOff = (uint32_t)offset; runLastRow = row; }
				if (row == 0 && node == st.seedNode && (here == 0 || here == 1))                                                      // free start (:500)
				{
					if (emitRuns && started) emitRun(node, (uint32_t)offset, row);
					row = 0xffffffffu; tracing = false; continue;
				}
				const uint32_t rowBefore = row;
				const int base = (int)(wbases >> (2 * (offset - wLo))) & 3;
				const bool match = ((e[base] >> r) & 1) != 0;
				int res = 0, via = 0;
				const uint32_t curNode = node, curOffset = (uint32_t)offset;
				auto decide = [&](int horizontal, int diagonal, uint32_t un, uint32_t uo) -> int {
					if (horizontal < here - 1) return -1;
					if (horizontal == here - 1) { node = un; offset = (int)uo; return 1; }
					if (match)
					{
						if (diagonal < here) return -1;
						if (diagonal == here) { node = un; offset = (int)uo; row = row - 1; return 2; }
					}
					else
					{
						if (diagonal < here - 1) return -1;
						if (diagonal == here - 1) { node = un; offset = (int)uo; row = row - 1; return 2; }
					}
					return 0;
				};
				// value in the last row of the slice above (the all-zero seed slice before slice 0) of a column requested above ...
				auto aboveOf = [&](bool in, const Col& c, uint32_t n) -> int {
					if (sIdx == 0) return n == st.seedNode ? 0 : big;
					return in ? col_value(c.vp, c.vn, c.before, W - 1) : big;
				};
				// ... and of one that was not (in-neighbours after the first): its own round trip
				auto valueAbove = [&](uint32_t n, uint32_t o) -> int {
					if (sIdx == 0) return n == st.seedNode ? 0 : big;
					Col c;
					if (!recordIn(tPN, tPB, pN, prvRow, n, o, c)) return big;
					return col_value(c.vp, c.vn, c.before, W - 1);
				};
				if (curOffset == 0)
				{
					// the in-neighbours, in the reference's order
#pragma unroll
					for (int k = 0; k < 4; k++)
					{
						if ((uint32_t)k < inDeg && res == 0)
						{
							const uint32_t mo = nbLen[k] - 1;
							via = k;
							const int horizontal = nIn[k] ? col_value(nc[k].vp, nc[k].vn, nc[k].before, r) : big;
							int diagonal = 0;
							if (horizontal > here - 1) diagonal = r > 0 ? (nIn[k] ? col_value(nc[k].vp, nc[k].vn, nc[k].before, r - 1) : big) : k == 0 ? aboveOf(aIn, ca, nb[0]) : valueAbove(nb[k], mo);
							res = decide(horizontal, diagonal, nb[k], mo);
						}
					}
					if (res == 1 || res == 2) { needSetup = true; nodeSteps++; }       // entered another node
				}
				else
				{
					const int horizontal = col_value(q1.vp, q1.vn, q1.before, r);
					int diagonal = 0;
					if (horizontal > here - 1) diagonal = r > 0 ? col_value(q1.vp, q1.vn, q1.before, r - 1) : aboveOf(aIn, ca, curNode);
					res = decide(horizontal, diagonal, curNode, curOffset - 1);
				}
				if (res < 0) { GAL_SITE(5); status = GA_ASSERTION; tracing = false; continue; }
				if (res == 0)
				{
					const int up = r > 0 ? col_value(q0.vp, q0.vn, q0.before, r - 1) : aboveOf(uIn, cu, curNode);
					if (up != here - 1) { GAL_SITE(6); status = GA_ASSERTION; tracing = false; continue; }              // assert(false) (:588)
					row = row - 1;
					res = 3;
				}
				// put the move away (the step onto the row before the first one is not part of the trace, :949-950)
				if (row == 0xffffffffu)
				{
					if (emitRuns && started) emitRun(curNode, curOffset, rowBefore);     // the trace ends in the cell the step left
					tracing = false; continue;
				}
				putMove(res, via);
				if (emitRuns)
				{
					if (curOffset == 0 && res != 3 && node != curNode)
					{
						// (a self loop stays in its run, as consecutive cells of one node do in traceToAlignment :817-821)
						if (started) { emitRun(curNode, curOffset, rowBefore); if (!tracing) continue; }
						runLastOff = (uint32_t)offset; runLastRow = row;
					}
					if (!started && row < traceRows) { started = true; runLastOff = (uint32_t)offset; runLastRow = row; }
				}
				changed = res >= 2 && r == 0;                                           // stepped into the slice above
				if (changed)
				{
					sIdx--;
					int t = tCN; tCN = tPN; tPN = t;
					t = tCB; tCB = tPB; tPB = t;
					nN = pN; curRow = prvRow;
					pN = aN; prvRow = aRow;
					needSetup = true;
				}
				}
				}
				const uint64_t g1 = lap_clock();
				// ---- where the lane stands now: the node's columns in the slice, a new node's record, the window, a new slice's tables ----
				if (needSetup)
				{
					const int slot = find_in(l, tCN, (int)nN, node);
					if (slot < 0) { GAL_SITE(4); status = GA_ASSERTION; tracing = false; continue; }       // assert(slice.scores.hasNode(nodeIndex)) (:498)
					slotRow = curRow + l.rd(tCB + slot);
					wLo = 1; wHi = 0;
				}
				if (node != recNode && node == specNode)
				{
					firstCol = spec.firstCol; inDeg = spec.inDeg;
#pragma unroll
					for (int k = 0; k < 4; k++) { nb[k] = spec.nb[k]; nbLen[k] = spec.nbLen[k]; }
					recNode = node;
				}
				if (node != recNode && node == nxNode)
				{
					firstCol = nx.firstCol; inDeg = nx.inDeg;
#pragma unroll
					for (int k = 0; k < 4; k++) { nb[k] = nx.nb[k]; nbLen[k] = nx.nbLen[k]; }
					recNode = node;
				}
				const bool needRec = node != recNode;
				needSetup = false;
				// the window: every column up to the current one that fits, reaching into the in-neighbour when the node is chained to it
				int extSlot = -1;
				if (!needRec && inDeg == 1 && nb[0] != node && nxNode == nb[0]) extSlot = find_in(l, tCN, (int)nN, nb[0]);
				const int lowLimit = extSlot >= 0 ? -(int)nbLen[0] : 0;
				if (extSlot >= 0) nxSlotRow = curRow + l.rd(tCB + extSlot);
				const bool refill = !(offset >= wLo && offset <= wHi) || (wLo > lowLimit && offset - wLo < kWin - 1);
				const int nLo = offset - (kWin - 1) > lowLimit ? offset - (kWin - 1) : lowLimit;
				Col wd[kWin];
				{
#pragma unroll
					for (int i = 0; i < kWin; i++)
					{
						const int c = nLo + i;
						rec_load_col<8>(m, refill && c <= offset ? (c >= 0 ? slotRow + (uint32_t)c : nxSlotRow + nbLen[0] - (uint32_t)(-c)) : safeRow, wd[i]);
					}
				}
				const int xLo = nLo > 0 ? nLo : 0;                                      // the window's first column of the node itself
				const bool seqNow = refill && !needRec, seqNx = refill && nLo < 0;
				uint32_t sq0 = 0, sq1 = 0, sq2 = 0, sy0 = 0, sy1 = 0, sy2 = 0;
				const uint64_t colX = seqNow ? firstCol + (uint64_t)xLo : 0ull;
				const uint64_t colY = seqNx ? nx.firstCol + nbLen[0] - (uint64_t)(-nLo) : 0ull;
				{
					const uint32_t* q = g.seq2 + (colX >> 4);
					sq0 = q[0]; sq1 = q[1]; sq2 = q[2];
					const uint32_t* y = g.seq2 + (colY >> 4);
					sy0 = y[0]; sy1 = y[1]; sy2 = y[2];
				}
				// the node's own record (entered through another in-neighbour than the first), and the record of ITS first in-neighbour
				const bool wantNx = !needRec && inDeg >= 1 && nxNode != nb[0];
				NodeRec fresh = NodeRec(), nxt = NodeRec();
				loadRec(node, fresh);
				loadRec(wantNx ? nb[0] : node, nxt);
				uint32_t tbl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hN = 0, hRow = 0;
				uint64_t ne[4] = {0, 0, 0, 0};
				if (GAL_ANY(changed))
				{
					// the slice above the new one: the first four entries of its node list here, the rest (wide bands) in the loop below
					const uint32_t* sn = m.snodes + (uint64_t)(changed && sIdx > 0 ? sIdx - 1 : 0) * 2 * N * m.ls;
#pragma unroll
					for (int i = 0; i < 8; i++) tbl[i] = sn[(uint64_t)i * m.ls];
					const uint32_t* h = m.hdr + (uint64_t)(changed && sIdx > 1 ? sIdx - 2 : 0) * kHdrWords * m.ls;
					hN = h[0]; hRow = h[6 * m.ls];
					const uint64_t* q = st.eq + (uint64_t)(changed ? sIdx : 0) * 5;
					ne[0] = q[0]; ne[1] = q[1]; ne[2] = q[2]; ne[3] = q[3];
				}
				const uint64_t g2 = lap_clock();
				// ---- arrived ----
				if (changed)
				{
					if (sIdx > 0)
					{
#pragma unroll
						for (int i = 0; i < 4; i++) if ((uint32_t)i < pN) { l.wr(tPN + i, tbl[2 * i]); l.wr(tPB + i, tbl[2 * i + 1]); }
						if (pN > 4) loadTableFrom(tPN, tPB, sIdx - 1, pN, 4);
					}
					if (sIdx > 1) { aN = hN; aRow = hRow; }
					e[0] = ne[0]; e[1] = ne[1]; e[2] = ne[2]; e[3] = ne[3];
				}
				if (wantNx) { nx = nxt; nxNode = nb[0]; }
				if (needRec)
				{
					firstCol = fresh.firstCol; inDeg = fresh.inDeg;
#pragma unroll
					for (int k = 0; k < 4; k++) { nb[k] = fresh.nb[k]; nbLen[k] = fresh.nbLen[k]; }
					recNode = node;
				}
				if (GAL_ANY(refill && needRec))
				{
					// (a node entered through another in-neighbour than the first: its base words could only be requested now)
					const uint64_t col = refill && needRec ? firstCol + (uint64_t)xLo : 0ull;
					const uint32_t* q = g.seq2 + (col >> 4);
					const uint32_t t0 = q[0], t1 = q[1], t2 = q[2];
					if (refill && needRec) { sq0 = t0; sq1 = t1; sq2 = t2; }
				}
				if (refill)
				{
					wLo = nLo; wHi = offset;
					const uint64_t bx = bases32(firstCol + (uint64_t)xLo, sq0, sq1, sq2);
					const int ny = xLo - nLo;                                           // columns of the in-neighbour in the window
					wbases = ny > 0 ? ((bases32(colY, sy0, sy1, sy2) & ~(~0ull << (2 * ny))) | (bx << (2 * ny))) : bx;
#pragma unroll
					for (int i = 0; i < kWin; i++)
					{
						if (wLo + i <= wHi)
						{
							const int at = tWIN + i * 5;
							l.wr(at, (uint32_t)wd[i].vp); l.wr(at + 1, (uint32_t)(wd[i].vp >> 32)); l.wr(at + 2, (uint32_t)wd[i].vn); l.wr(at + 3, (uint32_t)(wd[i].vn >> 32)); l.wr(at + 4, (uint32_t)wd[i].before);
						}
					}
				}
				winRead(offset, q0);
				if (offset > wLo) winRead(offset - 1, q1);
				{ const uint64_t g3 = lap_clock(); st.laps[4] += g1 - g0; st.laps[5] += g2 - g1; st.laps[6] += g3 - g2; }
			}
		}
#undef GAL_ANY
#undef GAL_SITE
#ifdef GA_DEBUG_SITE
		if (status != GA_OK) out.stamps[site & 7] = 1;
#endif
		{ const uint64_t t2 = lap_clock(); st.laps[1] += t2 - lapT; lapT = t2; }
		if (!emitRuns && (len & 3)) m.moves[(uint64_t)(len >> 2) * m.ls] = pack;
		st.laps[3] = lap_clock();
		// hand the moves (or runs) over: claim their bytes (rounded to words) of the pool and copy them
		if (status == GA_OK)
		{
			const uint32_t words = emitRuns ? 5 * nRuns : (len + 3) / 4;
			// the claim only ever commits when it fits (compare-and-swap): an overshoot that is rolled back later could leave the pool's
			// top below a region another lane claimed in between.  The lanes of a wave get here together: one of them claims for all
			// (a compare-and-swap per lane on one address is tens of thousands of lanes retrying against each other), and only a wave
			// whose total does not fit falls back to lane-by-lane claims.
			uint64_t at = 0;
			bool claimed = false;
#ifdef GA_EMULATE
			if (*L.trace_top + (uint64_t)words * 4 <= L.trace_pool_cap) { at = *L.trace_top; *L.trace_top += (uint64_t)words * 4; claimed = true; }
#else
			claimed = claim_for_wave(L.trace_top, L.trace_pool_cap, (uint64_t)words * 4, at);
#endif
			if (!claimed) status = GA_CAP_TRACE;
			else
			{
				uint32_t* dst = (uint32_t*)(L.traces + at);
				for (uint32_t k = 0; k < words; k += 8)
				{
					uint32_t a[8];
#pragma unroll
					for (int i = 0; i < 8; i++) a[i] = m.moves[(uint64_t)(k + (uint32_t)i) * m.ls];     // (the staging plane has a row of slack past `words`)
#pragma unroll
					for (int i = 0; i < 8; i++) if (k + (uint32_t)i < words) dst[k + (uint32_t)i] = a[i];
				}
				out.trace_off = at;
				out.trace_len = emitRuns ? nRuns : len;
				out.reserved3 = emitRuns ? 1u : 0u;
				out.n_node_steps = nodeSteps;
			}
		}
	}
	{ const uint64_t t2 = lap_clock(); st.laps[2] += t2 - st.laps[3]; }
	out.status = status;
	if (status != GA_OK) { out.n_valid = 0; out.score = 0x7fffffff; out.trace_len = 0; }
	L.outs[st.job] = out;
}

}  // namespace gal
