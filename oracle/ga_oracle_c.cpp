// ga_oracle_c.cpp -- flat C entry points over the CPU oracle, for ctypes-based tests and for
// bench.py's cpu_baseline leg.  TEST INFRASTRUCTURE ONLY (see ga_oracle.hpp).
#include "ga_oracle.hpp"

#include <atomic>
#include <chrono>
#include <cstring>
#include <queue>
#include <thread>

using namespace gao;

namespace {
struct ResultBox { AlignResult r; std::vector<SliceRecord> slices; };
template <typename F> int guarded(F f)
{
	try { f(); return 0; }
	catch (const Failure& e) { return (int)e.status; }
	catch (...) { return 99; }
}
}

extern "C" {

void* gao_graph_new() { return new Graph(); }
void gao_graph_free(void* g) { delete (Graph*)g; }
int gao_graph_add_node(void* g, int digraphId, const char* seq, int reverse) { return guarded([&] { ((Graph*)g)->addNode(digraphId, seq, reverse != 0); }); }
int gao_graph_add_edge(void* g, int from, int to) { return guarded([&] { ((Graph*)g)->addEdge(from, to); }); }
int gao_graph_add_bigraph_node(void* g, int id, const char* seq) { return guarded([&] { ((Graph*)g)->addBigraphNode(id, seq); }); }
int gao_graph_add_bigraph_edge(void* g, int from, int fromStart, int to, int toEnd) { return guarded([&] { ((Graph*)g)->addBigraphEdge(from, fromStart != 0, to, toEnd != 0); }); }
void gao_graph_set_overlap(void* g, int overlap) { ((Graph*)g)->dbgOverlap = overlap; }
int gao_graph_finalize(void* g) { return guarded([&] { ((Graph*)g)->finalize(); }); }
int64_t gao_graph_nodes(void* g) { return (int64_t)((Graph*)g)->nodeCount(); }
int64_t gao_graph_bp(void* g) { return (int64_t)((Graph*)g)->bp(); }

// seeds: triples (bigraph node id, read position, reverse)
void* gao_align(void* g, const char* name, const char* seq, int bw, int rampBw, const int64_t* seeds, int nSeeds, int recordSlices)
{
	std::vector<Seed> sv;
	for (int i = 0; i < nSeeds; i++) sv.emplace_back((int)seeds[3 * i], (size_t)seeds[3 * i + 1], seeds[3 * i + 2] != 0);
	ResultBox* box = new ResultBox();
	box->r = alignOneWay(*(Graph*)g, name, seq, bw, rampBw, sv, recordSlices ? &box->slices : nullptr);
	return box;
}
void gao_result_free(void* r) { delete (ResultBox*)r; }

void gao_result_summary(void* rb, int64_t* out)
{
	const AlignResult& r = ((ResultBox*)rb)->r;
	out[0] = r.status; out[1] = r.failed; out[2] = r.score; out[3] = (int64_t)r.alignmentStart; out[4] = (int64_t)r.alignmentEnd;
	out[5] = (int64_t)r.queryPosition; out[6] = (int64_t)r.mappings.size(); out[7] = (int64_t)r.trace.size();
	out[8] = (int64_t)r.fwTrace.size(); out[9] = (int64_t)r.bwTrace.size(); out[10] = r.fwScore; out[11] = r.bwScore;
	out[12] = (int64_t)r.columnsFirstPass; out[13] = (int64_t)r.slicesFirstPass; out[14] = (int64_t)((ResultBox*)rb)->slices.size();
	out[15] = (int64_t)r.sparseSlices; out[16] = (int64_t)r.overrideWindows; out[17] = (int64_t)r.overrideTraces;
}
const char* gao_result_message(void* rb) { return ((ResultBox*)rb)->r.message.c_str(); }
// 6 values per mapping: node_id (digraph), is_reverse, offset, rank, from_length, to_length
void gao_result_mappings(void* rb, int64_t* out)
{
	const AlignResult& r = ((ResultBox*)rb)->r;
	for (size_t i = 0; i < r.mappings.size(); i++)
	{
		const Mapping& m = r.mappings[i];
		int64_t* o = out + 6 * i;
		o[0] = m.nodeId; o[1] = m.isReverse; o[2] = m.offset; o[3] = m.rank; o[4] = m.fromLength; o[5] = m.toLength;
	}
}
const char* gao_result_mapping_seq(void* rb, int i) { return ((ResultBox*)rb)->r.mappings[i].editSeq.c_str(); }
// 7 values per item: nodeID, offset, reverse, readpos, type, graphChar, readChar
void gao_result_trace(void* rb, int64_t* out)
{
	const AlignResult& r = ((ResultBox*)rb)->r;
	for (size_t i = 0; i < r.trace.size(); i++)
	{
		const TraceItem& t = r.trace[i];
		int64_t* o = out + 7 * i;
		o[0] = t.nodeID; o[1] = (int64_t)t.offset; o[2] = t.reverse; o[3] = (int64_t)t.readpos; o[4] = t.type; o[5] = t.graphChar; o[6] = t.readChar;
	}
}
void gao_result_raw_trace(void* rb, int which, int64_t* out)
{
	const AlignResult& r = ((ResultBox*)rb)->r;
	const auto& t = which == 0 ? r.fwTrace : r.bwTrace;
	for (size_t i = 0; i < t.size(); i++) { out[2 * i] = (int64_t)t[i].first; out[2 * i + 1] = (int64_t)t[i].second; }
}
// slice records: info = direction, j, bandwidth, nNodes, nColumns, minScore, nMinIndex
void gao_slice_info(void* rb, int i, int64_t* out)
{
	const SliceRecord& s = ((ResultBox*)rb)->slices[i];
	out[0] = s.direction; out[1] = (int64_t)s.j; out[2] = s.bandwidth; out[3] = (int64_t)s.nodes.size(); out[4] = (int64_t)s.columns.size();
	out[5] = s.minScore; out[6] = (int64_t)s.minIndex.size();
}
void gao_slice_nodes(void* rb, int i, int64_t* out)
{
	const SliceRecord& s = ((ResultBox*)rb)->slices[i];
	for (size_t k = 0; k < s.nodes.size(); k++) out[k] = (int64_t)s.nodes[k];
}
void gao_slice_minindex(void* rb, int i, int64_t* out)
{
	const SliceRecord& s = ((ResultBox*)rb)->slices[i];
	for (size_t k = 0; k < s.minIndex.size(); k++) out[k] = (int64_t)s.minIndex[k];
}
void gao_slice_columns(void* rb, int i, uint64_t* vp, uint64_t* vn, int32_t* before, int32_t* end, uint8_t* beforeExists)
{
	const SliceRecord& s = ((ResultBox*)rb)->slices[i];
	for (size_t k = 0; k < s.columns.size(); k++)
	{
		vp[k] = s.columns[k].vp; vn[k] = s.columns[k].vn; before[k] = s.columns[k].before; end[k] = s.columns[k].end;
		beforeExists[k] = s.columns[k].beforeExists;
	}
}

// what the sparse method leaves per column: rows written, scoreEndExists; returns 1 when slice i was computed by the sparse method
int gao_slice_sparse_info(void* rb, int i, uint64_t* written, uint8_t* endExists)
{
	const SliceRecord& s = ((ResultBox*)rb)->slices[i];
	for (size_t k = 0; k < s.columns.size(); k++) { written[k] = s.columns[k].written; endExists[k] = s.columns[k].endExists; }
	return s.sparse ? 1 : 0;
}

// ---- component-level entry points (pinned against oracle/_ref) ---------------------------------
// column layout: vp, vn, end, before, rows, partial, beforeExists, endExists  (8 x int64, vp/vn bit-cast)
static Column unpackColumn(const int64_t* c)
{
	Column x;
	x.vp = (u64)c[0]; x.vn = (u64)c[1]; x.end = (int)c[2]; x.before = (int)c[3]; x.rows = (int)c[4];
	x.partial = c[5] != 0; x.beforeExists = c[6] != 0; x.endExists = c[7] != 0;
	return x;
}
static void packColumn(const Column& x, int64_t* c)
{
	c[0] = (int64_t)x.vp; c[1] = (int64_t)x.vn; c[2] = x.end; c[3] = x.before; c[4] = x.rows; c[5] = x.partial; c[6] = x.beforeExists; c[7] = x.endExists;
}
int gao_merge_columns(const int64_t* a, const int64_t* b, int64_t* out)
{
	return guarded([&] { packColumn(mergeColumns(unpackColumn(a), unpackColumn(b)), out); });
}
int gao_column_value(const int64_t* c, int row) { return columnValue(unpackColumn(c), row); }
// setCell chained from the sparse method's freshly touched column (same contract as ref_set_values in refparts.cpp)
int gao_set_values(const int* rows, const int* values, int n, int uninitialized, int64_t* out)
{
	Column w;
	w.vp = 0; w.vn = 0; w.end = uninitialized; w.before = uninitialized; w.rows = 0; w.partial = false; w.beforeExists = false; w.endExists = true;
	int done = 0;
	for (; done < n; done++)
	{
		Column next = w;
		try { setCell(next, rows[done], values[done]); }
		catch (const Failure&) { break; }
		w = next;
	}
	packColumn(w, out);
	return done;
}
int gao_step_column(uint64_t eq, const int64_t* left, int upIn, int upLeftIn, int diagIn, int prevRowEq, const int64_t* above, int64_t* out)
{
	return guarded([&] { packColumn(stepColumn(eq, unpackColumn(left), upIn, upLeftIn, diagIn, prevRowEq, unpackColumn(above), std::numeric_limits<int>::min()), out); });
}
void gao_hmm_chain(const int* mismatches, int n, double* correct, double* wrong, uint8_t* flags)
{
	Hmm h;
	for (int i = 0; i < n; i++)
	{
		h = h.next(mismatches[i], 64);
		correct[i] = h.correct; wrong[i] = h.wrong;
		flags[i] = (uint8_t)((h.correctFromCorrect ? 1 : 0) | (h.falseFromCorrect ? 2 : 0) | (h.currentlyCorrect() ? 4 : 0));
	}
}
int gao_frozen_order(const int64_t* nodes, int n, int64_t graphNodes, int64_t* out)
{
	std::vector<size_t> v(nodes, nodes + n);
	auto o = frozenIterationOrder(v, (size_t)graphNodes);
	for (size_t i = 0; i < o.size(); i++) out[i] = (int64_t)o[i];
	return (int)o.size();
}
// std::priority_queue<.., std::greater<>> comparing priorities only (GraphAligner.h:1094-1115):
// ops with prio >= 0 push (node, prio), prio < 0 pop; whatever is left is popped at the end.
// The pop order among equal priorities is what the device-side heap has to reproduce.
int gao_pq_order(const uint32_t* nodes, const int32_t* prios, int nOps, uint32_t* popped)
{
	struct Item { uint32_t node; int prio; bool operator>(const Item& o) const { return prio > o.prio; } };
	std::priority_queue<Item, std::vector<Item>, std::greater<Item>> q;
	int k = 0;
	for (int i = 0; i < nOps; i++)
	{
		if (prios[i] >= 0) q.push(Item{nodes[i], prios[i]});
		else if (!q.empty()) { popped[k++] = q.top().node; q.pop(); }
	}
	while (!q.empty()) { popped[k++] = q.top().node; q.pop(); }
	return k;
}
int gao_char_match(int readChar, int graphChar)
{
	try { return charMatch((char)readChar, (char)graphChar) ? 1 : 0; } catch (const Failure&) { return -1; }
}
int gao_reverse_complement(const char* in, char* out)
{
	return guarded([&] { std::string r = reverseComplement(in); memcpy(out, r.c_str(), r.size() + 1); });
}

// ---- multi-threaded batch run, the CPU baseline (scheduler as Aligner.cpp:285-298: N threads
//      popping reads from a shared queue; only the alignOneWay calls are timed) ---------------------
// reads: concatenated bytes, offsets[n+1]; seeds: one triple per read.
// out[0]=seconds, out[1]=aligned bp (reads with failed==0), out[2]=reads ok, out[3]=first-pass columns, out[4]=sum of scores
void gao_bench(void* g, const char* reads, const int64_t* offsets, const int64_t* seeds, int n, int bw, int rampBw, int threads, double* out)
{
	const Graph& graph = *(Graph*)g;
	std::atomic<int> next(0);
	std::vector<double> bp(threads, 0), ok(threads, 0), cols(threads, 0), scores(threads, 0);
	auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; t++)
	{
		pool.emplace_back([&, t] {
			while (true)
			{
				int i = next.fetch_add(1);
				if (i >= n) break;
				std::string seq(reads + offsets[i], reads + offsets[i + 1]);
				std::vector<Seed> sv{Seed((int)seeds[3 * i], (size_t)seeds[3 * i + 1], seeds[3 * i + 2] != 0)};
				AlignResult r = alignOneWay(graph, "r", seq, bw, rampBw, sv, nullptr);
				cols[t] += (double)r.columnsFirstPass;
				if (!r.failed) { bp[t] += (double)seq.size(); ok[t] += 1; scores[t] += r.score; }
			}
		});
	}
	for (auto& th : pool) th.join();
	auto t1 = std::chrono::steady_clock::now();
	out[0] = std::chrono::duration<double>(t1 - t0).count();
	out[1] = out[2] = out[3] = out[4] = 0;
	for (int t = 0; t < threads; t++) { out[1] += bp[t]; out[2] += ok[t]; out[3] += cols[t]; out[4] += scores[t]; }
}

int gao_interleaved_rank(uint64_t vp, uint64_t vn, int lo, int hi, int rank) { return interleavedRankForTest(vp, vn, lo, hi, rank); }
int gao_work_stack(const int64_t* ops, int nOps, int64_t universe, int64_t* popped)
{
	std::vector<long long> v(ops, ops + nOps);
	std::vector<size_t> out = workStackForTest(v, (size_t)universe);
	for (size_t i = 0; i < out.size(); i++) popped[i] = (int64_t)out[i];
	return (int)out.size();
}

}  // extern "C"
