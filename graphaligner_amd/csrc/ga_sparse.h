// ga_sparse.h -- the reference's fallbacks for bands of 200 000 cells and more, on the device: the sparse method
// (calculateSliceAlternate / setValue / finalizeAlternateSlice, GraphAligner.h:2148-2329, 2130-2146, 2523-2552; WordSlice.h:231-337)
// and what the backtrace override (:167-354) adds to a traceback that already keeps every slice.
//
// Included by ga_kernel.h (namespace gak); compiled into the last kernel of the ladder only.  This is a fallback: a band that wide
// is a tangle, or a fan of long nodes, and the reference itself leaves its bit vectors for a cell-by-cell bucket queue there.  The
// queue's ORDER is part of the result (the first touch of a node decides the slice's node order and with it the next slice's map
// order; the last entry of the final bucket is where the traceback starts), so the queue is run as what it is: a sequential program,
// by lane 0, over per-slot tables in HBM.  Everything around it -- collecting the previous slice's usable end cells, turning the
// touched columns into an ordinary slice record (every column of every touched node, untouched ones at the fill value of :2546-2549)
// -- is done by the whole wave.  From then on the slice is a slice like any other: the next band is projected from it, a bit-vector
// slice can follow it, the traceback walks through it.
#pragma once

namespace gak {

constexpr uint32_t kSetSize = 1u << 14;          // cells processed in one row (open addressing, generation-stamped)
constexpr uint32_t kMapSize = 1u << 17;          // (node slot, offset) -> touched word
constexpr uint32_t kSparseWords = 1u << 16;      // touched words per slice
constexpr uint32_t kSparseEntries = 1u << 19;    // queue entries per row set, shared evenly by the bandwidth + 1 buckets

struct SparseMem
{
	uint32_t* gen;        // [4] running generation numbers of the two tables (kept across the slot's jobs; zeroed by the host at launch)
	uint32_t* cnt;        // [2][nb] entries per bucket
	uint64_t* ent;        // [2][kSparseEntries] node << 32 | offset inside the node
	uint64_t* setKey; uint32_t* setGen;                       // [kSetSize]
	uint64_t* mapKey; uint32_t* mapGen; uint32_t* mapVal;     // [kMapSize]
	uint64_t* wVp; uint64_t* wVn; int32_t* wBefore; int32_t* wEnd; uint32_t* wSlot; uint32_t* wOff; uint32_t* wRows;   // [kSparseWords]
	uint32_t* list;       // [2 * kSparseWords] work lists (usable end cells of the previous slice; the override's reachable cells)
};
inline
#ifndef GA_EMULATE
__host__ __device__
#endif
uint64_t sparse_mem_bytes(uint32_t maxBandwidth)
{
	auto up = [](uint64_t x) { return (x + 255) & ~255ull; };
	uint64_t at = 256;
	at += up(8ull * ((uint64_t)maxBandwidth + 1));
	at += up(16ull * kSparseEntries);
	at += up(12ull * kSetSize);
	at += up(16ull * kMapSize);
	at += up(36ull * kSparseWords);
	at += up(8ull * kSparseWords);
	return at;
}
inline
#ifndef GA_EMULATE
__host__ __device__
#endif
SparseMem sparse_mem_at(uint8_t* base, uint32_t maxBandwidth)
{
	auto up = [](uint64_t x) { return (x + 255) & ~255ull; };
	SparseMem m;
	uint64_t at = 0;
	m.gen = (uint32_t*)(base + at); at += 256;
	m.cnt = (uint32_t*)(base + at); at += up(8ull * ((uint64_t)maxBandwidth + 1));
	m.ent = (uint64_t*)(base + at); at += up(16ull * kSparseEntries);
	m.setKey = (uint64_t*)(base + at); m.setGen = (uint32_t*)(base + at + 8ull * kSetSize); at += up(12ull * kSetSize);
	m.mapKey = (uint64_t*)(base + at); m.mapGen = (uint32_t*)(base + at + 8ull * kMapSize); m.mapVal = m.mapGen + kMapSize; at += up(16ull * kMapSize);
	m.wVp = (uint64_t*)(base + at); m.wVn = m.wVp + kSparseWords; m.wBefore = (int32_t*)(m.wVn + kSparseWords); m.wEnd = m.wBefore + kSparseWords;
	m.wSlot = (uint32_t*)(m.wEnd + kSparseWords); m.wOff = m.wSlot + kSparseWords; m.wRows = m.wOff + kSparseWords; at += up(36ull * kSparseWords);
	m.list = (uint32_t*)(base + at);
	return m;
}

// what the sparse fill leaves for its caller (wave-uniform)
struct SparseResult
{
	int status;
	int minScore;             // the slice's minimum (row 63)
	int minSlot;              // touched-node slot and offset of the LAST entry of the final bucket: where a traceback would start (:2320-2326, :922)
	uint32_t minOffset;
	uint32_t nWords;          // touched columns
	uint32_t numCells;        // sum of the touched nodes' lengths (DPSlice::numCells, :2550)
	bool endsTooFar;          // getFrozenSqrtEndScores would assert: scoreEnd - minimum does not fit 16 bits (NodeSlice.h:372)
	bool beforeTooFar;        // getFrozenScores would assert (NodeSlice.h:344)
	bool oddWord;             // a written column whose scoreEnd equals the "uninitialised" marker (:2546): not representable here
};

GA_FN uint32_t mix64(uint64_t k) { k ^= k >> 29; k *= 0x9e3779b97f4a7c15ull; k ^= k >> 32; return (uint32_t)k; }

// WordSlice::setValue on a touched word (WordSlice.h:231-337); rows = confirmedRows.rows | partial << 8
GA_FN int sparse_set_value(uint64_t& vp, uint64_t& vn, int& before, int& end, uint32_t& rowsWord, int row, int value)
{
	const bool partial = (rowsWord >> 8) != 0;
	const int rows = (int)(rowsWord & 0xffu);
	if (!partial)
	{
		before = value + row + 1;
		if (row < W - 1) { vn = ~(~0ull << (row + 1)); vp = ~0ull << (row + 1); }
		else { vn = ~0ull; vp = 0; }
		end = value + W - row - 1;
		rowsWord = (uint32_t)row | 0x100u;
		return GA_OK;
	}
	if (!(rows < row)) return GA_ASSERTION;
	if (rows == row - 1)
	{
		const int old = end - (W - rows - 1);
		const uint64_t lowMask = rows < 63 ? ~(~0ull << (rows + 1)) : ~0ull;
		if (old != before + __builtin_popcountll(vp & lowMask) - __builtin_popcountll(vn & lowMask)) return GA_ASSERTION;
		if (value < old - 1 || value > old + 1) return GA_ASSERTION;
		const uint64_t m = 1ull << row;
		if (value == old - 1) { vn |= m; vp &= ~m; end -= 2; }
		else if (value == old) { vn &= ~m; vp &= ~m; end -= 1; }
		else { vp |= m; vn &= ~m; }
		rowsWord = (uint32_t)row | 0x100u;
		return GA_OK;
	}
	// a gap (:281-336): rows rows+1 .. row rise by one each, then every row down to `row` is capped by the run going up from the new cell.
	// Scores are rebuilt row by row without an array: s_i of the old column, min with value + row - i, deltas against the previous result.
	int s = before, prev = before;
	uint64_t nvp = vp, nvn = vn;
	for (int i = 0; i <= row; i++)
	{
		if (i <= rows) s += (int)((vp >> i) & 1) - (int)((vn >> i) & 1); else s += 1;
		const int cap = value + row - i;
		const int v = s < cap ? s : cap;
		const int delta = v - prev;
		if (delta < -1 || delta > 1) return GA_ASSERTION;
		const uint64_t m = 1ull << i;
		if (delta == -1) { nvp &= ~m; nvn |= m; }
		else if (delta == 0) { nvp &= ~m; nvn &= ~m; }
		else { nvp |= m; nvn &= ~m; }
		prev = v;
	}
	vp = nvp; vn = nvn;
	end = prev + W - 1 - row;
	rowsWord = (uint32_t)row | 0x100u;
	return GA_OK;
}

// ---- one slice by the sparse method ---------------------------------------------------------------------------------------------
// in: the previous slice as the bit-vector path leaves it (pn_* tables, slot.end_prev); rows = the job's row codes at this slice.
// out: ws.cn_node / cn_len / cn_min / st_cur (touched columns per node) for the touched nodes in first-touch order, the touched words
// in sm.w*, and the summary.  Nothing is written to the arena here.
template <int MAXN>
GA_FN SparseResult sparse_fill(const GaDevGraph& g, WaveState<MAXN>& ws, const Slot& slot, const SparseMem& sm, const uint8_t* rows, uint32_t nRows,
                               uint32_t j, int pn, int prevMin, int bandwidth, int& cnOut)
{
	SparseResult res;
	res.status = GA_OK; res.minScore = 0; res.minSlot = 0; res.minOffset = 0; res.nWords = 0; res.numCells = 0;
	res.endsTooFar = false; res.beforeTooFar = false; res.oddWord = false;
	cnOut = 0;
	// (:2220 indexes calculables[1] of a one-element vector when the bandwidth is 0 -- slice 0 of a run without a ramp width: undefined
	// behaviour in the reference, reported as an assertion)
	if (bandwidth < 1) { res.status = GA_ASSERTION; return res; }
	const VI lane = lane_iota();
	{
		const VI rc = load_lanes(rows, W, 0);
		if (ballot((rc & GA_ROW_INVALID) != 0)) { res.status = GA_ASSERTION; return res; }      // characterMatch's default branch (:2104-2106)
	}
	const uint32_t nb = (uint32_t)bandwidth + 1;
	const uint32_t capB = kSparseEntries / nb;
	// ---- the previous slice's end cells a path can continue from (:2163-2219): score < minimum + bandwidth and scoreEndExists, in the
	// previous slice's map order, columns ascending.  list[2k] = previous slot << 24 | ..., kept as (slot, offset, end) triples.
	if (pn <= LANES) hash_order_lanes(ws, ws.pn_node, pn); else hash_order(ws, ws.pn_node, pn);
	wave_sync();
	uint32_t nSeeds = 0;
	bool seedsFull = false;
	for (int k = 0; k < pn && !seedsFull; k++)
	{
		const int s = ws.h_order[k];
		const uint32_t len = ws.pn_len[s];
		const uint32_t* pend = slot.end_prev + ws.pn_colBase[s];
		for (uint32_t w0 = 0; w0 < len; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			const VI raw = load_lanes(pend + w0, n, 0);
			const uint64_t usable = ballot((lane < n) && ((raw >> 3) < prevMin + bandwidth) && ((raw & 4) != 0));
			if (!usable) continue;
			const int count = __builtin_popcountll(usable);
			if (nSeeds + (uint32_t)count > (2 * kSparseWords) / 3) { seedsFull = true; break; }
			// lane L's place = the set bits below it
			const VU below = VU(usable) & mask_low_bits(lane);
			const VI at = vpopc(below) * 3 + (int)(nSeeds * 3);
			const VB mine = (lane < n) && ((raw >> 3) < prevMin + bandwidth) && ((raw & 4) != 0);
			scatter(sm.list, at, VI(s), mine);
			scatter(sm.list, at + 1, lane + (int)w0, mine);
			scatter(sm.list, at + 2, raw >> 3, mine);
			nSeeds += (uint32_t)count;
		}
	}
	wave_sync();
	if (seedsFull) { res.status = GA_CAP_HEAP; return res; }

	int status = GA_OK, minScore = prevMin, minSlot = 0, cn = 0;
	uint32_t minOffset = 0, nWords = 0, numCells = 0;
	int endsTooFar = 0, beforeTooFar = 0, oddWord = 0;
	if (GA_LANE0)
	{
		uint32_t setGen = sm.gen[0], mapGen = sm.gen[1] + 1;
		uint32_t* cntNow = sm.cnt;
		uint32_t* cntNext = sm.cnt + nb;
		uint64_t* entNow = sm.ent;
		uint64_t* entNext = sm.ent + kSparseEntries;
		for (uint32_t b = 0; b < nb; b++) { cntNow[b] = 0; cntNext[b] = 0; }
		auto push = [&](uint32_t* cnt, uint64_t* ent, int bucket, uint32_t node, uint32_t off) {
			if (bucket < 0 || bucket > bandwidth) { status = GA_ASSERTION; return; }                 // (out of range: undefined behaviour in the reference)
			const uint32_t c = cnt[bucket];
			if (c >= capB) { status = GA_CAP_HEAP; return; }
			ent[(uint64_t)bucket * capB + c] = ((uint64_t)node << 32) | off;
			cnt[bucket] = c + 1;
		};
		// a node's record, remembered for the cells that follow in the same node
		uint32_t recNode = 0xffffffffu, recLen = 0, recOutDeg = 0, recOut[4] = {0, 0, 0, 0};
		uint64_t recFirst = 0;
		auto nodeInfo = [&](uint32_t node) {
			if (node == recNode) return;
			const uint32_t* r = g.node_rec + (uint64_t)node * GA_NODE_REC_WORDS;
			recNode = node; recFirst = ((uint64_t)r[1] << 32) | r[0]; recLen = r[2]; recOutDeg = r[3] >> 16;
			for (int e = 0; e < 4; e++) recOut[e] = r[4 + e];
		};
		auto matchAt = [&](int row, uint64_t column) { return ((rows[row] >> g_base(g, column)) & 1) != 0; };
		auto firstColOf = [&](uint32_t node) { const uint32_t* r = g.node_rec + (uint64_t)node * GA_NODE_REC_WORDS; return ((uint64_t)r[1] << 32) | r[0]; };
		// ---- row j from the previous slice (:2163-2219) ----
		for (uint32_t q = 0; q < nSeeds && status == GA_OK; q++)
		{
			const int s = (int)sm.list[3 * q];
			const uint32_t off = sm.list[3 * q + 1];
			const int rel = (int)sm.list[3 * q + 2] - prevMin;
			const uint32_t node = ws.pn_node[s];
			nodeInfo(node);
			if (j == 0)
				push(cntNow, entNow, rel + (matchAt(0, recFirst + off) ? 0 : 1), node, off);
			else
			{
				if (rel < 0) { status = GA_ASSERTION; break; }                                         // assert(scoreEnd >= previousSlice.minScore) (:2189)
				push(cntNow, entNow, rel + 1, node, off);
				if (off + 1 < recLen) push(cntNow, entNow, rel + (matchAt(0, recFirst + off + 1) ? 0 : 1), node, off + 1);
				else
				{
					const uint32_t deg = recOutDeg;
					for (uint32_t e = 0; e < deg && status == GA_OK; e++)
					{
						const uint32_t nbr = deg <= 4 ? recOut[e] : g.out_nbr[g.out_off[node] + e];
						push(cntNow, entNow, rel + (matchAt(0, firstColOf(nbr)) ? 0 : 1), nbr, 0);
					}
				}
			}
		}
		if (status == GA_OK && cntNow[0] == 0 && cntNow[1] == 0) status = GA_ASSERTION;              // :2220
		// ---- the rows (:2224-2316) ----
		uint32_t lastSlotNode = 0xffffffffu;
		int lastSlot = -1;
		for (int row = 0; row < W && status == GA_OK; row++)
		{
			const int plus = cntNow[0] == 0 ? -1 : 0;
			setGen++;
			auto seen = [&](uint64_t key) -> bool {
				uint32_t h = mix64(key) & (kSetSize - 1);
				while (sm.setGen[h] == setGen) { if (sm.setKey[h] == key) return true; h = (h + 1) & (kSetSize - 1); }
				return false;
			};
			uint32_t inRow = 0;
			for (int sp = 0; sp < bandwidth && status == GA_OK; sp++)
			{
				for (uint32_t k = 0; k < cntNow[sp] && status == GA_OK; k++)
				{
					const uint64_t key = entNow[(uint64_t)sp * capB + k];
					const uint32_t node = (uint32_t)(key >> 32), off = (uint32_t)key;
					{
						// processed[] (:2234-2236)
						uint32_t h = mix64(key) & (kSetSize - 1);
						bool dup = false;
						while (sm.setGen[h] == setGen) { if (sm.setKey[h] == key) { dup = true; break; } h = (h + 1) & (kSetSize - 1); }
						if (dup) continue;
						if (++inRow > kSetSize / 2) { status = GA_CAP_HEAP; break; }
						sm.setKey[h] = key; sm.setGen[h] = setGen;
					}
					nodeInfo(node);
					if (off >= recLen) { status = GA_ASSERTION; break; }
					// the node's slot in this slice, first touch = addNode (:2132-2141)
					int slotN = -1;
					if (node == lastSlotNode) slotN = lastSlot;
					else
					{
						for (int t = 0; t < cn; t++) if (ws.cn_node[t] == node) { slotN = t; break; }
						if (slotN < 0)
						{
							if (cn >= MAXN) { status = GA_CAP_NODES; break; }
							slotN = cn++;
							ws.cn_node[slotN] = node; ws.cn_len[slotN] = recLen; ws.st_cur[slotN] = 0;
							numCells += recLen;
						}
						lastSlotNode = node; lastSlot = slotN;
					}
					// the column's word
					const uint64_t wkey = ((uint64_t)(uint32_t)slotN << 32) | off;
					uint32_t h = mix64(wkey) & (kMapSize - 1);
					uint32_t wi = 0xffffffffu;
					while (sm.mapGen[h] == mapGen) { if (sm.mapKey[h] == wkey) { wi = sm.mapVal[h]; break; } h = (h + 1) & (kMapSize - 1); }
					uint64_t vp = 0, vn = 0;
					int before = (int)nRows, end = (int)nRows;
					uint32_t rowsWord = 0;
					if (wi == 0xffffffffu)
					{
						if (nWords >= kSparseWords) { status = GA_CAP_COLS; break; }
						wi = nWords++;
						sm.mapKey[h] = wkey; sm.mapGen[h] = mapGen; sm.mapVal[h] = wi;
						sm.wSlot[wi] = (uint32_t)slotN; sm.wOff[wi] = off;
						ws.st_cur[slotN] += 1;
					}
					else { vp = sm.wVp[wi]; vn = sm.wVn[wi]; before = sm.wBefore[wi]; end = sm.wEnd[wi]; rowsWord = sm.wRows[wi]; }
					status = sparse_set_value(vp, vn, before, end, rowsWord, row, minScore + sp);
					if (status != GA_OK) break;
					sm.wVp[wi] = vp; sm.wVn[wi] = vn; sm.wBefore[wi] = before; sm.wEnd[wi] = end; sm.wRows[wi] = rowsWord;
					// onwards: the cell below, the cell to the right in this row, the diagonal one (:2259-2297)
					push(cntNext, entNext, sp + 1 + plus, node, off);
					auto onward = [&](uint32_t n2, uint32_t o2, uint64_t column) {
						if (!seen(((uint64_t)n2 << 32) | o2)) push(cntNow, entNow, sp + 1, n2, o2);
						if (row < W - 1) push(cntNext, entNext, sp + plus + (matchAt(row + 1, column) ? 0 : 1), n2, o2);
					};
					if (off + 1 == recLen)
					{
						const uint32_t deg = recOutDeg;
						for (uint32_t e = 0; e < deg && status == GA_OK; e++)
						{
							const uint32_t nbr = deg <= 4 ? recOut[e] : g.out_nbr[g.out_off[node] + e];
							onward(nbr, 0, firstColOf(nbr));
						}
					}
					else onward(node, off + 1, recFirst + off + 1);
				}
			}
			if (status != GA_OK) break;
			if (cntNow[0] == 0) minScore++;
			if (row < W - 1)
			{
				uint32_t* tc = cntNow; cntNow = cntNext; cntNext = tc;
				uint64_t* te = entNow; entNow = entNext; entNext = te;
				for (uint32_t b = 0; b < nb; b++) cntNext[b] = 0;
			}
		}
		if (status == GA_OK)
		{
			// the cells of the last row at the minimum; the traceback starts from the last of them (:2318-2326)
			const int bucket = cntNow[0] == 0 ? 1 : 0;
			if (cntNow[bucket] == 0) status = GA_ASSERTION;
			else
			{
				const uint64_t key = entNow[(uint64_t)bucket * capB + cntNow[bucket] - 1];
				minOffset = (uint32_t)key;
				minSlot = -1;
				for (int t = 0; t < cn; t++) if (ws.cn_node[t] == (uint32_t)(key >> 32)) { minSlot = t; break; }
				if (minSlot < 0) status = GA_ASSERTION;
			}
		}
		if (status == GA_OK)
		{
			// finalizeAlternateSlice (:2523-2552): node minima, the fill value of the untouched columns, and what the frozen forms could hold
			const int uninit = (int)nRows;
			for (int t = 0; t < cn; t++) ws.cn_min[t] = ws.st_cur[t] < ws.cn_len[t] ? uninit : 0x7fffffff;
			for (uint32_t w = 0; w < nWords; w++)
			{
				const uint32_t t = sm.wSlot[w];
				const int e = sm.wEnd[w];
				if (e == uninit) oddWord = 1;
				if (e < ws.cn_min[t]) ws.cn_min[t] = e;
			}
			int loEnd = 0x7fffffff, hiEnd = -0x7fffffff, loBefore = 0x7fffffff, hiBefore = -0x7fffffff;
			for (int t = 0; t < cn; t++)
			{
				if (ws.st_cur[t] < ws.cn_len[t])
				{
					const int fill = ws.cn_min[t] + (int)ws.cn_len[t] + bandwidth + 1;
					loEnd = loEnd < fill ? loEnd : fill; hiEnd = hiEnd > fill ? hiEnd : fill;
					loBefore = loBefore < fill ? loBefore : fill; hiBefore = hiBefore > fill ? hiBefore : fill;
				}
			}
			for (uint32_t w = 0; w < nWords; w++)
			{
				const int e = sm.wEnd[w], b = sm.wBefore[w];
				loEnd = loEnd < e ? loEnd : e; hiEnd = hiEnd > e ? hiEnd : e;
				loBefore = loBefore < b ? loBefore : b; hiBefore = hiBefore > b ? hiBefore : b;
			}
			endsTooFar = (hiEnd - loEnd >= 65535) ? 1 : 0;                                            // NodeSlice.h:372
			beforeTooFar = (hiBefore - loBefore >= 65535) ? 1 : 0;                                    // NodeSlice.h:344
		}
		sm.gen[0] = setGen; sm.gen[1] = mapGen;
	}
	wave_sync();
	res.status = wave_uniform(status);
	res.minScore = wave_uniform(minScore);
	res.minSlot = wave_uniform(minSlot);
	res.minOffset = (uint32_t)wave_uniform((int)minOffset);
	res.nWords = (uint32_t)wave_uniform((int)nWords);
	res.numCells = (uint32_t)wave_uniform((int)numCells);
	res.endsTooFar = wave_uniform(endsTooFar) != 0;
	res.beforeTooFar = wave_uniform(beforeTooFar) != 0;
	res.oddWord = wave_uniform(oddWord) != 0;
	cnOut = wave_uniform(cn);
	return res;
}

// ---- the sparse slice as an ordinary slice record: every column of every touched node ---------------------------------------------
// rec: record of cn nodes and numCells columns at the arena's top; also leaves the end words in slot.end_cur and the node tables the
// next slice's band is projected from.  exists[] (one byte per column, behind the record's planes) = scoreEndExists (:2536).
template <int MAXN>
GA_FN void sparse_materialize(const GaDevGraph& g, WaveState<MAXN>& ws, const Slot& slot, const SparseMem& sm, const SliceRec& rec, uint8_t* exists,
                              int cn, uint32_t nWords, int bandwidth)
{
	const VI lane = lane_iota();
	// column bases in touch order
	if (GA_LANE0)
	{
		uint32_t at = 0;
		for (int t = 0; t < cn; t++) { ws.cn_colBase[t] = at; at += ws.cn_len[t]; ws.cn_prev[t] = -1; ws.cn_outDeg[t] = 255; }
	}
	wave_sync();
	for (int c = 0; c < cn; c += LANES)
	{
		store_lanes(rec.nodes + c, cn - c, load_lanes(ws.cn_node + c, cn - c, 0));
		store_lanes(rec.colBase + c, cn - c, load_lanes(ws.cn_colBase + c, cn - c, 0));
		store_lanes(rec.nodeMin + c, cn - c, load_lanes(ws.cn_min + c, cn - c, 0));
	}
	// untouched columns: VP = VN = 0, scoreBeforeStart = scoreEnd = node minimum + node length + bandwidth + 1, no end cell (:2546-2549, :2536)
	for (int t = 0; t < cn; t++)
	{
		const uint32_t len = ws.cn_len[t], base = ws.cn_colBase[t];
		const int fill = ws.cn_min[t] + (int)len + bandwidth + 1;
		for (uint32_t w0 = 0; w0 < len; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			store_lanes(rec.vp + base + w0, n, VU(0ull));
			store_lanes(rec.vn + base + w0, n, VU(0ull));
			store_lanes(rec.before + base + w0, n, VI(fill));
			store_lanes(slot.end_cur + base + w0, n, VI(fill << 3));
			store_lanes(exists + base + w0, n, VI(0));
		}
		if (GA_LANE0) { ws.cn_lastEnd[t] = fill; ws.cn_lastEnd2[t] = fill; }
	}
	wave_sync();
	// the written columns
	for (uint32_t w0 = 0; w0 < nWords; w0 += LANES)
	{
		const int n = (int)(nWords - w0 < (uint32_t)LANES ? nWords - w0 : (uint32_t)LANES);
		const VB live = lane < n;
		const VI t = load_lanes(sm.wSlot + w0, n, 0);
		const VI off = load_lanes(sm.wOff + w0, n, 0);
		const VI idx = gather(ws.cn_colBase, t) + off;
		const VU vp = load_lanes_u64(sm.wVp + w0, n), vn = load_lanes_u64(sm.wVn + w0, n);
		const VI before = load_lanes(sm.wBefore + w0, n, 0), end = load_lanes(sm.wEnd + w0, n, 0);
		const VI ex = select((load_lanes(sm.wRows + w0, n, 0) & 0xff) == W - 1, VI(1), VI(0));
		scatter64(rec.vp, idx, vp, live);
		scatter64(rec.vn, idx, vn, live);
		scatter(rec.before, idx, before, live);
		scatter(slot.end_cur, idx, (end << 3) | (ex << 2) | vpopc(vp & VU(1ull << 63)) | (vpopc(vn & VU(1ull << 63)) << 1), live);
		scatter(exists, idx, ex, live);
		// a node's last column, for the next slice's projection
		const VB isLast = live && (off + 1 == gather(ws.cn_len, t));
		scatter(ws.cn_lastEnd, t, end, isLast);
		scatter(ws.cn_lastEnd2, t, end - vpopc(vp & VU(1ull << 63)) + vpopc(vn & VU(1ull << 63)), isLast);
	}
	wave_sync();
}

// ---- what building a BacktraceOverride checks (GraphAligner.h:236-343) ---------------------------------------------------------------
// The reference works out, while a window of >= 200 000-cell slices is still in memory, the predecessor of every cell that can be
// reached backwards from an existing end cell of the window's last slice, and lets the traceback follow those links later.  This
// program keeps every slice, so its traceback needs no links; what remains of the override is (a) the traceback's own rule inside a
// window it uses -- a cell in a slice's last row must exist (:211, :202-209), enforced in run_job -- and (b) that pickBacktracePredecessor
// asserts on EVERY reachable cell, on the path or not.  (b) is this function: the same walk, lane 0, cell by cell.
struct RecView { const uint32_t* nodes; const uint32_t* colBase; uint32_t nNodes; const uint64_t* vp; const uint64_t* vn; const int32_t* before; const uint8_t* exists; };
GA_FN RecView rec_view(uint32_t* arena, uint32_t off)
{
	RecView v;
	const uint32_t nN = arena[off], nC = arena[off + 1];
	const SliceRec r = slice_at(arena, off, nN, nC);
	v.nodes = r.nodes; v.colBase = r.colBase; v.nNodes = nN; v.vp = r.vp; v.vn = r.vn; v.before = r.before;
	v.exists = (const uint8_t*)(r.before + nC);
	return v;
}
GA_FN int view_slot(const RecView& v, uint32_t node) { for (uint32_t t = 0; t < v.nNodes; t++) if (v.nodes[t] == node) return (int)t; return -1; }
GA_FN int view_value(const RecView& v, int slotN, uint32_t off, int row)
{
	const uint32_t idx = v.colBase[slotN] + off;
	const uint64_t mask = row < 63 ? ~(~0ull << (row + 1)) : ~0ull;
	return v.before[idx] + __builtin_popcountll(v.vp[idx] & mask) - __builtin_popcountll(v.vn[idx] & mask);
}

// predecessor of (node, off, r) in the slice `cur` (r = row inside the slice); `above` = the slice before it (seedAbove: that is the
// all-zero seed slice).  Returns 0 left, 1 diagonal, 2 up, < 0 an assertion; the cell entered in pn / po.
GA_FN int pick_pred_scalar(const GaDevGraph& g, const RecView& cur, const RecView& above, bool seedAbove, uint32_t seedNode, const uint8_t* rowCodes,
                           uint32_t globalRow, int big, uint32_t node, uint32_t off, int r, uint32_t& pn, uint32_t& po, bool& freeStart)
{
	freeStart = false;
	const int sl = view_slot(cur, node);
	if (sl < 0) return -1;                                                                  // assert(slice.scores.hasNode(nodeIndex)) (:498)
	const int here = view_value(cur, sl, off, r);
	auto aboveValue = [&](uint32_t n, uint32_t o) -> int {
		if (seedAbove) return n == seedNode ? 0 : big;
		const int t = view_slot(above, n);
		return t < 0 ? big : view_value(above, t, o, W - 1);
	};
	if (globalRow == 0 && (seedAbove ? node == seedNode : view_slot(above, node) >= 0) && (here == 0 || here == 1)) { freeStart = true; pn = node; po = off; return 2; }   // :500
	const uint32_t* rec = g.node_rec + (uint64_t)node * GA_NODE_REC_WORDS;
	const uint64_t column = (((uint64_t)rec[1] << 32) | rec[0]) + off;
	const bool match = ((rowCodes[globalRow] >> g_base(g, column)) & 1) != 0;
	auto tryFrom = [&](uint32_t un, uint32_t uo, int& out) -> bool {
		const int t = view_slot(cur, un);
		const int horizontal = t < 0 ? big : view_value(cur, t, uo, r);
		if (horizontal < here - 1) { out = -1; return true; }
		if (horizontal == here - 1) { pn = un; po = uo; out = 0; return true; }
		const int diagonal = r == 0 ? aboveValue(un, uo) : (t < 0 ? big : view_value(cur, t, uo, r - 1));
		const int want = match ? here : here - 1;
		if (diagonal < want) { out = -1; return true; }
		if (diagonal == want) { pn = un; po = uo; out = 1; return true; }
		return false;
	};
	int out = 0;
	if (off == 0)
	{
		const uint32_t inDeg = rec[3] & 0xffffu;
		for (uint32_t e = 0; e < inDeg; e++)
		{
			const uint32_t m = inDeg <= 4 ? rec[8 + e] : g.in_nbr[g.in_off[node] + e];
			const uint32_t mo = (inDeg <= 4 ? rec[12 + e] : g.node_rec[(uint64_t)m * GA_NODE_REC_WORDS + 2]) - 1;
			if (tryFrom(m, mo, out)) return out;
		}
	}
	else if (tryFrom(node, off - 1, out)) return out;
	const int up = r == 0 ? aboveValue(node, off) : view_value(cur, sl, off, r - 1);
	if (up < here - 1) return -1;
	if (up == here - 1) { pn = node; po = off; return 2; }
	return -1;                                                                              // assert(false) (:588)
}

// windowRecs[k] = arena offset of the window's k-th slice (first slice index = firstSlice), preRec = the slice before the window
// (kSeedRecordMark: the seed slice).  Returns GA_OK or the status the construction would have ended the read with.
template <int MAXN>
GA_FN int explore_override(const GaDevGraph& g, WaveState<MAXN>& ws, const Slot& slot, const SparseMem& sm, const uint32_t* sliceOff, uint32_t firstSlice, uint32_t count,
                           uint32_t preRec, bool preIsSeed, uint32_t seedNode, const uint8_t* rowCodes, int big)
{
	const VI lane = lane_iota();
	// the existing end cells of the window's last slice, in the record's order (the order does not matter: every cell is visited once)
	const uint32_t lastOff = sliceOff[firstSlice + count - 1];
	const uint32_t nN = slot.arena[lastOff], nC = slot.arena[lastOff + 1];
	const SliceRec lastRec = slice_at(slot.arena, lastOff, nN, nC);
	const uint8_t* lastExists = (const uint8_t*)(lastRec.before + nC);
	uint32_t nStart = 0;
	bool full = false;
	for (uint32_t t = 0; t < nN && !full; t++)
	{
		const uint32_t base = lastRec.colBase[t], len = (t + 1 < nN ? lastRec.colBase[t + 1] : nC) - base;
		const uint32_t node = lastRec.nodes[t];
		for (uint32_t w0 = 0; w0 < len; w0 += LANES)
		{
			const int n = (int)(len - w0 < (uint32_t)LANES ? len - w0 : (uint32_t)LANES);
			const VI ex = load_lanes(lastExists + base + w0, n, 0);
			const uint64_t m = ballot((lane < n) && (ex != 0));
			if (!m) continue;
			const int c = __builtin_popcountll(m);
			if (nStart + (uint32_t)c > kSparseWords / 2) { full = true; break; }
			const VI at = vpopc(VU(m) & mask_low_bits(lane)) * 2 + (int)(nStart * 2);
			const VB mine = (lane < n) && (ex != 0);
			scatter(sm.list, at, VI((int)node), mine);
			scatter(sm.list, at + 1, lane + (int)w0, mine);
			nStart += (uint32_t)c;
		}
	}
	wave_sync();
	if (full) return GA_CAP_HEAP;
	int status = GA_OK;
	if (GA_LANE0)
	{
		uint32_t gen = sm.gen[0];
		uint32_t* cur = sm.list;                      // (node, offset) pairs
		uint32_t* nxt = sm.list + kSparseWords;
		uint32_t nCur = nStart, nNxt = 0;
		const uint32_t nRowsWin = count * W;
		// visited cells of the row being walked and of the row above it: one table, entries stamped gen + 1 + row (older stamps are free slots)
		uint32_t liveA = gen + 1 + (nRowsWin - 1), liveB = gen + 1 + (nRowsWin - 2);
		auto visit = [&](uint32_t row, uint32_t node, uint32_t off) -> bool {       // true when the cell is new
			const uint64_t key = ((uint64_t)node << 32) | off;
			const uint32_t stamp = gen + 1 + row;
			uint32_t h = mix64(key ^ ((uint64_t)row << 40)) & (kSetSize - 1);
			uint32_t probes = 0;
			while (sm.setGen[h] == liveA || sm.setGen[h] == liveB)
			{
				if (sm.setGen[h] == stamp && sm.setKey[h] == key) return false;
				h = (h + 1) & (kSetSize - 1);
				if (++probes > kSetSize - 2) { status = GA_CAP_HEAP; return false; }
			}
			sm.setKey[h] = key; sm.setGen[h] = stamp;
			return true;
		};
		for (uint32_t q = 0; q < nCur; q++) visit(nRowsWin - 1, cur[2 * q], cur[2 * q + 1]);
		uint32_t live = nCur;                          // entries of the two rows in the table (bounded well below its size)
		for (uint32_t row = nRowsWin; row-- > 0 && status == GA_OK;)
		{
			const uint32_t si = row / W;
			const int r = (int)(row % W);
			const RecView view = rec_view(slot.arena, sliceOff[firstSlice + si]);
			const bool seedAbove = si == 0 && preIsSeed;
			RecView above = view;
			if (!(si == 0 && preIsSeed)) above = rec_view(slot.arena, si > 0 ? sliceOff[firstSlice + si - 1] : preRec);
			const uint32_t globalRow = (firstSlice + si) * W + (uint32_t)r;
			liveA = gen + 1 + row; liveB = row > 0 ? gen + row : liveA;
			nNxt = 0;
			for (uint32_t q = 0; q < nCur && status == GA_OK; q++)
			{
				const uint32_t node = cur[2 * q], off = cur[2 * q + 1];
				if (row > 0 && r == W - 1)
				{
					// a cell without an end score is registered and not followed (:243-250)
					const int t = view_slot(view, node);
					if (t < 0) { status = GA_ASSERTION; break; }
					if (view.exists[view.colBase[t] + off] == 0) continue;
				}
				uint32_t pn2 = 0, po2 = 0;
				bool freeStart = false;
				const int res = pick_pred_scalar(g, view, above, seedAbove, seedNode, rowCodes, globalRow, big, node, off, r, pn2, po2, freeStart);
				if (res < 0) { status = GA_ASSERTION; break; }
				if (res == 0)
				{
					if (visit(row, pn2, po2))
					{
						if (nCur >= kSparseWords / 2 || ++live > kSetSize / 2) { status = GA_CAP_HEAP; break; }
						cur[2 * nCur] = pn2; cur[2 * nCur + 1] = po2; nCur++;
					}
				}
				else if (row > 0 && !freeStart)
				{
					if (visit(row - 1, pn2, po2))
					{
						if (nNxt >= kSparseWords / 2 || ++live > kSetSize / 2) { status = GA_CAP_HEAP; break; }
						nxt[2 * nNxt] = pn2; nxt[2 * nNxt + 1] = po2; nNxt++;
					}
				}
			}
			live = nNxt;
			uint32_t* t = cur; cur = nxt; nxt = t;
			nCur = nNxt;
		}
		sm.gen[0] = gen + 1 + nRowsWin + 1;
	}
	wave_sync();
	(void)ws;
	return wave_uniform(status);
}

}  // namespace gak
