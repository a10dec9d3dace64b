// refparts.cpp -- thin extern "C" shims over the parts of the REFERENCE that compile from their
// own self-contained source files (no protobuf, no boost): WordSlice.h, NodeSlice.h,
// AlignmentCorrectnessEstimation.{h,cpp}, ThreadReadAssertion.{h,cpp}.
//
// Built by oracle/Makefile into oracle/_ref/libga_refparts.so from the sources where they lie
// under /root/reference (nothing is copied into this repository).  Used ONLY by tests to pin
// the matching functions of the CPU oracle (oracle/ga_oracle.cpp).  The reference's engine
// itself (GraphAligner.h) needs protobuf-generated vg.pb.h and is NOT buildable in this image.
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "ThreadReadAssertion.h"          // defines the throwing assert() the headers below rely on
#include "WordSlice.h"
#include "NodeSlice.h"
#include "AlignmentCorrectnessEstimation.h"
#include "UniqueQueue.h"

typedef WordSlice<size_t, int, uint64_t> RefWord;

static RefWord unpack(const int64_t* c)
{
	RefWord w;
	w.VP = (uint64_t)c[0]; w.VN = (uint64_t)c[1]; w.scoreEnd = (int)c[2]; w.scoreBeforeStart = (int)c[3];
	w.confirmedRows.rows = (char)c[4]; w.confirmedRows.partial = c[5] != 0;
	w.scoreBeforeExists = c[6] != 0; w.scoreEndExists = c[7] != 0;
	return w;
}
static void pack(const RefWord& w, int64_t* c)
{
	c[0] = (int64_t)w.VP; c[1] = (int64_t)w.VN; c[2] = w.scoreEnd; c[3] = w.scoreBeforeStart;
	c[4] = w.confirmedRows.rows; c[5] = w.confirmedRows.partial; c[6] = w.scoreBeforeExists; c[7] = w.scoreEndExists;
}

extern "C" {

// WordSlice::mergeWith (WordSlice.h:202-206)
int ref_merge_columns(const int64_t* a, const int64_t* b, int64_t* out)
{
	try { pack(unpack(a).mergeWith(unpack(b)), out); return 0; }
	catch (const ThreadReadAssertion::AssertionFailure&) { return 1; }
}
// WordSlice::getValue (WordSlice.h:223-229)
int ref_column_value(const int64_t* c, int row) { return unpack(c).getValue(row); }

// WordSlice::setValue (WordSlice.h:231-337) applied, cell after cell, to a column initialised the way the sparse method's
// first touch of a node does (GraphAligner.h:2137-2140): {0, 0, uninitialized, uninitialized, 0, false}, partial = false.
// Returns the number of cells applied before an assertion (n when none failed); `out` is the column after the last good one.
int ref_set_values(const int* rows, const int* values, int n, int uninitialized, int64_t* out)
{
	RefWord w {0, 0, uninitialized, uninitialized, 0, false};
	w.confirmedRows.partial = false;
	int done = 0;
	for (; done < n; done++)
	{
		RefWord next = w;
		try { next.setValue(rows[done], values[done]); }
		catch (const ThreadReadAssertion::AssertionFailure&) { break; }
		w = next;
	}
	pack(w, out);
	return done;
}

// AlignmentCorrectnessEstimationState::NextState chained from the default state
void ref_hmm_chain(const int* mismatches, int n, double* correct, double* wrong, uint8_t* flags)
{
	AlignmentCorrectnessEstimationState h;
	for (int i = 0; i < n; i++)
	{
		h = h.NextState(mismatches[i], 64);
		correct[i] = h.CorrectLogOdds(); wrong[i] = h.FalseLogOdds();
		flags[i] = (uint8_t)((h.CorrectFromCorrect() ? 1 : 0) | (h.FalseFromCorrect() ? 2 : 0) | (h.CurrentlyCorrect() ? 4 : 0));
	}
}

// NodeSlice: add `n` nodes (in the given order, one column each) through the dense vectorMap,
// freeze to sqrt-end-scores, and report the iteration order of the frozen copy
// (NodeSlice.h:584-599, 724-740, 680-723)
int ref_frozen_order(const int64_t* nodes, int n, int64_t graphNodes, int64_t* out)
{
	try
	{
		std::vector<NodeSlice<RefWord>::MapItem> dense((size_t)graphNodes, NodeSlice<RefWord>::MapItem{0, 0, 0});
		NodeSlice<RefWord> live(&dense);
		for (int i = 0; i < n; i++) live.addNode((size_t)nodes[i], 1);
		const NodeSlice<RefWord> frozen = live.getFrozenSqrtEndScores();
		int k = 0;
		for (auto it = frozen.begin(); it != frozen.end(); ++it) out[k++] = (int64_t)(*it).first;
		return k;
	}
	catch (const ThreadReadAssertion::AssertionFailure&) { return -1; }
}

// WordContainer freeze / thaw: columns in, thawed columns out. mode 1 = getFrozenScores, 2 = getFrozenSqrtEndScores
int ref_freeze_thaw(const int64_t* cols, int n, int mode, int64_t* out)
{
	try
	{
		typedef WordContainer<size_t, int, uint64_t> C;
		C live;
		live.resize((size_t)n);
		for (int i = 0; i < n; i++) live[(size_t)i] = unpack(cols + 8 * i);
		const C frozen = mode == 1 ? live.getFrozenScores() : live.getFrozenSqrtEndScores();
		for (int i = 0; i < n; i++) pack(frozen[(size_t)i], out + 8 * i);
		return 0;
	}
	catch (const ThreadReadAssertion::AssertionFailure&) { return 1; }
}

// WordConfiguration<uint64_t> helpers behind confirmedRowsInMerged (WordSlice.h:27-130)
int ref_popcount(uint64_t x) { return WordConfiguration<uint64_t>::popcount(x); }
uint64_t ref_chunk_popcounts(uint64_t x) { return WordConfiguration<uint64_t>::ChunkPopcounts(x); }
uint64_t ref_morton_low(uint64_t a, uint64_t b) { return WordConfiguration<uint64_t>::MortonLow(a, b); }
uint64_t ref_morton_high(uint64_t a, uint64_t b) { return WordConfiguration<uint64_t>::MortonHigh(a, b); }
// the rank query of WordSlice.h:479-488: position of the rank-th set bit in the row-interleaved (VP, ~VN) words restricted to rows [lo, hi)
int ref_interleaved_rank(uint64_t vp, uint64_t vn, int lo, int hi, int rank)
{
	try
	{
		uint64_t mask = hi < 64 ? ~(~0ull << hi) : ~0ull;
		mask &= lo < 64 ? (~0ull << lo) : 0ull;
		const uint64_t low = vp & mask, high = ~vn & mask;
		return WordConfiguration<uint64_t>::BitPosition(WordConfiguration<uint64_t>::MortonLow(low, high), WordConfiguration<uint64_t>::MortonHigh(low, high), rank);
	}
	catch (const ThreadReadAssertion::AssertionFailure&) { return -1; }
}

// UniqueQueue<size_t> (UniqueQueue.h:6-69) driven by a list of operations: op >= 0 insert(op), op == -1 pop.  Returns the popped items.
int ref_unique_queue(const int64_t* ops, int nOps, int64_t universe, int64_t* popped)
{
	try
	{
		UniqueQueue<size_t> q((size_t)universe);
		int k = 0;
		for (int i = 0; i < nOps; i++)
		{
			if (ops[i] >= 0) q.insert((size_t)ops[i]);
			else if (q.size() > 0) { popped[k++] = (int64_t)q.top(); q.pop(); }
		}
		while (q.size() > 0) { popped[k++] = (int64_t)q.top(); q.pop(); }
		return k;
	}
	catch (const ThreadReadAssertion::AssertionFailure&) { return -1; }
}

}  // extern "C"
