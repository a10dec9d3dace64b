// ga_host.cpp -- host side of the C ABI (include/graphaligner_amd.h): the graph model the
// reference's loaders build (AlignmentGraph.cpp, BigraphToDigraph.cpp), splitting reads into
// extension jobs (GraphAligner.h:2969-3024), and turning the device's raw traces back into
// AlignmentResults (GraphAligner.h:408-491, 594-847, 3026-3098).  No alignment arithmetic
// happens here; the extension program runs behind ga_backend.h on the GPU.
#include <sched.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <memory>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>
#include <atomic>
#include <string_view>
#include <sys/mman.h>
#include <mutex>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/graphaligner_amd.h"
#include "ga_backend.h"

namespace {

constexpr int W = 64;

// ---- character tables ------------------------------------------------------------------------
struct CharTables
{
	uint8_t rowCode[256];    // bits 0-3: which of A,C,G,T the read char matches; bits 4-6: exact code; bit 7: not IUPAC
	uint8_t rowCodeRc[256];  // rowCode of the complement
	uint8_t complement[256]; // 0 = the reference's ReverseComplement asserts on this char
	uint8_t baseCode[256];   // A0 C1 G2 T3, 255 otherwise
	CharTables()
	{
		for (int c = 0; c < 256; c++) { rowCode[c] = GA_ROW_INVALID; complement[c] = 0; baseCode[c] = 255; }
		auto setMatch = [&](char upper, int mask) {
			rowCode[(uint8_t)upper] = (uint8_t)(mask | (7 << 4));
			rowCode[(uint8_t)(upper + 32)] = (uint8_t)(mask | (7 << 4));
		};
		const int A = 1, C = 2, G = 4, T = 8;
		// characterMatch (GraphAligner.h:2039-2110)
		setMatch('A', A); setMatch('C', C); setMatch('G', G); setMatch('T', T); setMatch('N', A | C | G | T);
		setMatch('R', A | G); setMatch('Y', C | T); setMatch('K', G | T); setMatch('M', C | A); setMatch('S', C | G); setMatch('W', A | T);
		setMatch('B', C | G | T); setMatch('D', A | G | T); setMatch('H', A | C | T); setMatch('V', A | C | G);
		// the "previousEq" comparison is a raw char == against the upper-case graph base (:1503,1540)
		rowCode[(uint8_t)'A'] = (uint8_t)(A | (0 << 4)); rowCode[(uint8_t)'C'] = (uint8_t)(C | (1 << 4));
		rowCode[(uint8_t)'G'] = (uint8_t)(G | (2 << 4)); rowCode[(uint8_t)'T'] = (uint8_t)(T | (3 << 4));
		baseCode[(uint8_t)'A'] = 0; baseCode[(uint8_t)'C'] = 1; baseCode[(uint8_t)'G'] = 2; baseCode[(uint8_t)'T'] = 3;
		// ReverseComplement (CommonUtils.cpp:60-136): 'H'/'h' fall through to assert(false); 'U' -> 'A'
		const char* from = "ACTGNURYKMSWBVD";
		const char* to = "TGACNAYRMKSWVBH";
		for (int i = 0; from[i]; i++) { complement[(uint8_t)from[i]] = (uint8_t)to[i]; complement[(uint8_t)(from[i] + 32)] = (uint8_t)to[i]; }
		for (int c = 0; c < 256; c++) rowCodeRc[c] = rowCode[complement[c]];
	}
};
const CharTables& tables() { static const CharTables t; return t; }

GaHmmTables buildHmm()
{
	// AlignmentCorrectnessEstimation.cpp:6-36, 61-69, 81-83 -- same libm calls, same order
	GaHmmTables t;
	const double cMis = log(0.2), cMat = log(1.0 - 0.2), fMis = log(0.5), fMat = log(1.0 - 0.5);
	t.f2c = log(0.00001); t.f2f = log(1.0 - 0.00001);
	t.c2f = log(0.000000000000001); t.c2c = log(1.0 - 0.000000000000001);
	double lf[65];
	lf[0] = 0;
	for (int i = 1; i <= 64; i++) lf[i] = lf[i - 1] + log(i);
	for (int m = 0; m <= 64; m++)
	{
		double choose = lf[64] - lf[m] - lf[64 - m];
		t.correct_mult[m] = choose + m * cMis + (64 - m) * cMat;
		t.wrong_mult[m] = choose + m * fMis + (64 - m) * fMat;
	}
	t.init_correct = log(0.8);
	t.init_wrong = log(0.2);
	return t;
}

}  // namespace

// ================================================================================================
// graph
// ================================================================================================
struct ga_graph
{
	int dbgOverlap = 0;
	bool finalized = false;
	std::vector<uint64_t> nodeStart;
	std::unordered_map<int64_t, uint32_t> lookup;
	std::vector<int64_t> ids;
	std::vector<std::pair<uint32_t, uint32_t>> edgeList;      // (from, to) node indices in the order the edges were added; the CSR lists are built at Finalize
	// nodes whose finished neighbour lists were handed over verbatim (ga_graph_set_neighbors): node index -> (in-list, out-list)
	std::unordered_map<uint32_t, std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> givenLists;
	std::vector<uint8_t> reverse;
	std::vector<uint8_t> bases;         // 0..3, 4 for the dummy columns
	GaFlatGraph flat;
	GaHmmTables hmm;
	GaBackendGraph* device = nullptr;
	// optional node splitting (ga_graph_load_gfa_split): piece bigraph id -> the node it was cut from
	struct Piece { int64_t orig; uint64_t start, len, origLen; };
	std::unordered_map<int64_t, Piece> pieces;

	ga_graph()
	{
		// dummy start node, one column (AlignmentGraph.cpp:22-30)
		ids.push_back(0); nodeStart.push_back(0); reverse.push_back(0); bases.push_back(4);
	}
	uint32_t nodeCount() const { return (uint32_t)nodeStart.size(); }
	uint64_t nodeEnd(uint32_t n) const { return n + 1 == nodeStart.size() ? bases.size() : nodeStart[n + 1]; }
	uint32_t nodeLen(uint32_t n) const { return (uint32_t)(nodeEnd(n) - nodeStart[n]); }
	char baseChar(uint32_t node, uint32_t offset) const { uint8_t b = bases[nodeStart[node] + offset]; return b < 4 ? "ACGT"[b] : '-'; }
	// neighbour lists in insertion order, without double edges (AlignmentGraph.cpp:104-105); valid once finalized
	uint32_t inDegree(uint32_t n) const { return flat.in_off[n + 1] - flat.in_off[n]; }
	uint32_t inNeighbor(uint32_t n, uint32_t k) const { return flat.in_nbr[flat.in_off[n] + k]; }
	bool hasOutNeighbor(uint32_t n, uint32_t to) const
	{
		for (uint32_t e = flat.out_off[n]; e < flat.out_off[n + 1]; e++) if (flat.out_nbr[e] == to) return true;
		return false;
	}
	int reverseNode(uint32_t n, uint32_t& outNode) const
	{
		// GetReverseNode (AlignmentGraph.cpp:199-214)
		int64_t big = ids[n] / 2;
		auto it = lookup.find(ids[n] % 2 == 1 ? big * 2 : big * 2 + 1);
		if (it == lookup.end() || it->second == n || nodeLen(it->second) != nodeLen(n)) return GA_S_ASSERTION;
		outNode = it->second;
		return GA_S_OK;
	}
};

static int addNode(ga_graph* g, int64_t id, const char* seq, size_t len, bool rev)
{
	if (g->finalized) return GA_E_INVALID;
	if (g->lookup.count(id)) return GA_S_OK;            // duplicates ignored (AlignmentGraph.cpp:51)
	for (size_t i = 0; i < len; i++) if (tables().baseCode[(uint8_t)seq[i]] > 3) return GA_E_INVALID;   // graph is ACGT only (:83-85)
	if (g->nodeStart.size() >= 0x7ffffff0u) return GA_E_INVALID;
	g->lookup[id] = (uint32_t)g->nodeStart.size();
	g->ids.push_back(id);
	g->nodeStart.push_back(g->bases.size());
	g->reverse.push_back(rev ? 1 : 0);
	for (size_t i = 0; i < len; i++) g->bases.push_back(tables().baseCode[(uint8_t)seq[i]]);
	return GA_S_OK;
}

static int addEdge(ga_graph* g, int64_t from, int64_t to)
{
	if (g->finalized) return GA_E_INVALID;
	auto f = g->lookup.find(from), t = g->lookup.find(to);
	if (f == g->lookup.end() || t == g->lookup.end()) return GA_E_INVALID;
	g->edgeList.emplace_back(f->second, t->second);                                               // double edges are dropped at Finalize (:104-105)
	return GA_S_OK;
}

// a node's inNeighbors / outNeighbors exactly as a finished AlignmentGraph holds them (AlignmentGraph.h:49-50)
static int setNeighbors(ga_graph* g, int64_t id, const int64_t* in, size_t nIn, const int64_t* out, size_t nOut)
{
	if (g->finalized) return GA_E_INVALID;
	auto me = g->lookup.find(id);
	if (me == g->lookup.end()) return GA_E_INVALID;
	std::pair<std::vector<uint32_t>, std::vector<uint32_t>> lists;
	for (size_t k = 0; k < nIn; k++) { auto it = g->lookup.find(in[k]); if (it == g->lookup.end()) return GA_E_INVALID; lists.first.push_back(it->second); }
	for (size_t k = 0; k < nOut; k++) { auto it = g->lookup.find(out[k]); if (it == g->lookup.end()) return GA_E_INVALID; lists.second.push_back(it->second); }
	g->givenLists[me->second] = std::move(lists);
	return GA_S_OK;
}

static std::string revcompACGT(const char* seq, size_t len)
{
	std::string r(len, 'N');
	for (size_t i = 0; i < len; i++) r[i] = (char)tables().complement[(uint8_t)seq[len - 1 - i]];
	return r;
}

static int finalizeGraph(ga_graph* g, int overlap)
{
	if (g->finalized) return GA_E_INVALID;
	g->dbgOverlap = overlap;
	// dummy end node (AlignmentGraph.cpp:110-118)
	g->ids.push_back(0); g->nodeStart.push_back(g->bases.size()); g->reverse.push_back(0); g->bases.push_back(4);
	g->finalized = true;
	const uint32_t n = g->nodeCount();
	GaFlatGraph& f = g->flat;
	f.node_start.assign(g->nodeStart.begin(), g->nodeStart.end());
	f.node_start.push_back(g->bases.size());
	f.seq2.assign((g->bases.size() + 15) / 16 + 8, 0);      // (+ slack: the kernels request base words a little past a node's end)
	for (size_t i = 0; i < g->bases.size(); i++) f.seq2[i >> 4] |= (uint32_t)(g->bases[i] & 3) << ((i & 15) * 2);
	// neighbour lists: every node's in- and out-list in the order its edges were added, a repeated edge kept once (the reference
	// checks std::find before every push_back, AlignmentGraph.cpp:104-105).  Counting sort by node, then per-node de-duplication.
	{
		const auto& E = g->edgeList;
		std::vector<uint32_t> inCount(n + 1, 0), outCount(n + 1, 0);
		for (const auto& e : E) { outCount[e.first + 1]++; inCount[e.second + 1]++; }
		for (uint32_t i = 0; i < n; i++) { inCount[i + 1] += inCount[i]; outCount[i + 1] += outCount[i]; }
		std::vector<uint32_t> inAll(E.size() + 1), outAll(E.size() + 1);
		{
			std::vector<uint32_t> inAt(inCount.begin(), inCount.end() - 1), outAt(outCount.begin(), outCount.end() - 1);
			for (const auto& e : E) { outAll[outAt[e.first]++] = e.second; inAll[inAt[e.second]++] = e.first; }
		}
		auto compact = [&](const std::vector<uint32_t>& count, const std::vector<uint32_t>& all, std::vector<uint32_t>& off, std::vector<uint32_t>& nbr, bool inLists) {
			off.assign(n + 1, 0);
			nbr.clear();
			nbr.reserve(all.size() + 1);
			for (uint32_t i = 0; i < n; i++)
			{
				const size_t first = nbr.size();
				if (!g->givenLists.empty())
				{
					auto given = g->givenLists.find(i);
					if (given != g->givenLists.end())
					{
						// a finished list, taken as it is
						const std::vector<uint32_t>& lst = inLists ? given->second.first : given->second.second;
						nbr.insert(nbr.end(), lst.begin(), lst.end());
						off[i + 1] = (uint32_t)nbr.size();
						continue;
					}
				}
				for (uint32_t k = count[i]; k < count[i + 1]; k++)
				{
					bool seen = false;
					for (size_t q = first; q < nbr.size(); q++) if (nbr[q] == all[k]) { seen = true; break; }
					if (!seen) nbr.push_back(all[k]);
				}
				off[i + 1] = (uint32_t)nbr.size();
			}
			nbr.push_back(0);
		};
		compact(inCount, inAll, f.in_off, f.in_nbr, true);
		compact(outCount, outAll, f.out_off, f.out_nbr, false);
		std::vector<std::pair<uint32_t, uint32_t>>().swap(g->edgeList);
		g->givenLists.clear();
	}
	g->hmm = buildHmm();
	return GA_S_OK;
}

// ================================================================================================
// jobs and results
// ================================================================================================
namespace {

// a read as the batch keeps it: a view into the batch's one buffer of read copies
typedef std::string_view ReadSeq;

struct Pos { uint32_t node, offset; uint64_t row; };
typedef std::vector<Pos> Trace;

struct SeedPlan
{
	bool valid = false;          // lookups succeeded and the position is inside the read
	int early = GA_S_OK;         // status fixed before any device work (bad seed, assertion while splitting)
	uint32_t seedNodeIndex = 0;  // nodeLookup.at(id * 2), what the "already aligned" test compares (GraphAligner.h:423-425)
	int64_t fwJob = -1, bwJob = -1;
	uint64_t pos = 0;
};

// threads this process can really run at once: the affinity mask, capped by a cgroup CPU quota.  More threads than that are worse than
// useless under a quota: a burst of 64 threads on 16 cores' worth of quota spends the period's budget in a quarter of the period and the
// whole process -- the thread that waits for the GPU included -- is throttled for the rest of it.
static size_t usableCores()
{
	static const size_t cached = []() -> size_t {
		size_t n = std::thread::hardware_concurrency();
#ifdef __linux__
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min<size_t>(n ? n : (size_t)c, (size_t)c); }
		if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))
		{
			char quota[32]; long long period = 0;
			if (fscanf(f, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
				n = std::min<size_t>(n, (size_t)std::max<long long>(1, (atoll(quota) + period / 2) / period));
			fclose(f);
		}
		else if (FILE* q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))
		{
			long long quota = -1, period = 0;
			if (fscanf(q, "%lld", &quota) != 1) quota = -1;
			fclose(q);
			if (FILE* pf = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(pf, "%lld", &period) != 1) period = 0; fclose(pf); }
			if (quota > 0 && period > 0) n = std::min<size_t>(n, (size_t)std::max<long long>(1, (quota + period / 2) / period));
		}
#endif
		return std::max<size_t>(1, n);
	}();
	return cached;
}

struct ReadPlan { size_t firstSeed = 0, nSeeds = 0; };

// (a mapping's edit sequence is a piece of the read: kept as a span of it, copied once into the results)
struct SeqSpan { uint64_t pos, len; };
struct Partial { bool failed = true; int32_t score = 0; std::vector<ga_mapping_t> maps; std::vector<SeqSpan> seqs; };

}  // namespace

struct ga_batch
{
	const ga_graph* g = nullptr;
	std::vector<std::string> names;
	std::vector<ReadSeq> seqs;            // into seqBuf
	char* seqBuf = nullptr;               // the batch's own copy of the reads, one buffer; shared with the results collected from the batch:
	std::shared_ptr<char> seqKeep;        // an Edit's sequence is a piece of the read (GraphAligner.h:829, 845), so results point into it
	size_t seqBufBytes = 0;
	std::vector<ReadPlan> reads;
	std::vector<SeedPlan> seeds;
	struct RowFill { size_t read; uint64_t off, n, padded, pos; bool backward; };
	std::vector<RowFill> fills;         // where every job's rows come from
	std::vector<uint8_t> rows;          // row codes, built when a kernel that wants them is about to run (see buildRows)
	std::atomic<int> anyInvalidRow{0};    // some read has a character outside IUPAC (its results need the TraceItem pass)
	std::unique_ptr<std::atomic<uint8_t>[]> readInvalid;   // ... per read: a row of one of its jobs has such a character (noted while the match words are built)
	std::unique_ptr<uint64_t[]> eq;       // match words per slice (not cleared first: every slice is written by the host threads)
	size_t eqWords = 0;
	std::vector<GaJob> jobs;
	GaRunConfig cfg;
	uint32_t flags = 0;
	GaBackendBatch* dev = nullptr;
	bool ran = false;
	uint64_t columnUpdates = 0, slicesRun = 0, rowsTotal = 0;
	~ga_batch() { delete dev; }
};

// run `fn(fill, read sequence)` for every job's rows on the host threads
template <typename F> static void forEachFill(ga_batch* b, F fn)
{
	const auto& fills = b->fills;
	size_t nThreads = usableCores();
	if (const char* e = getenv("GA_HOST_THREADS")) nThreads = (size_t)atoi(e);
	nThreads = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nThreads, 64), fills.size() / 64 + 1));
	std::vector<std::thread> pool;
	const size_t per = (fills.size() + nThreads - 1) / nThreads;
	for (size_t t = 0; t < nThreads; t++)
	{
		const size_t lo = std::min(fills.size(), t * per), hi = std::min(fills.size(), lo + per);
		if (lo < hi) pool.emplace_back([&, lo, hi]() { for (size_t k = lo; k < hi; k++) fn(fills[k], b->seqs[fills[k].read]); });
	}
	for (auto& th : pool) th.join();
}

// the row codes (one byte per padded read base: match set, exact-compare code, validity): only the wave-per-read kernels read them
static void buildRows(ga_batch* b)
{
	if (!b->rows.empty()) return;
	const CharTables& T = tables();
	b->rows.assign(b->rowsTotal + 64, 0);       // (+ slack so a 64-byte row load never leaves the buffer)
	forEachFill(b, [&](const ga_batch::RowFill& f, const ReadSeq& seq) {
		const uint8_t padCode = T.rowCode[(uint8_t)'N'];
		uint8_t* dst = b->rows.data() + f.off;
		if (f.backward) for (uint64_t r = 0; r < f.n; r++) dst[r] = T.rowCode[T.complement[(uint8_t)seq[f.n - 1 - r]]];
		else for (uint64_t r = 0; r < f.n; r++) dst[r] = T.rowCode[(uint8_t)seq[f.pos + r]];
		for (uint64_t r = f.n; r < f.padded; r++) dst[r] = padCode;
	});
}

namespace {

// Large result arrays are recycled: unmapping a gigabyte when results are freed and faulting a fresh one in for the next batch costs
// more than assembling the results.  A few buffers are kept (GA_RESULT_POOL_MB, default 4096; 0 = none), best fit first.
class BufferPool
{
	struct Item { void* p; size_t cap; };
	std::mutex lock;
	std::vector<Item> items;
	size_t held = 0;
	static size_t limit()
	{
		static const size_t v = []() { const char* e = getenv("GA_RESULT_POOL_MB"); return (size_t)(e ? atoll(e) : 4096) << 20; }();
		return v;
	}
public:
	void* take(size_t bytes, size_t& cap)
	{
		{
			std::lock_guard<std::mutex> g(lock);
			int best = -1;
			for (size_t i = 0; i < items.size(); i++)
				if (items[i].cap >= bytes && items[i].cap <= 2 * bytes + (1u << 20) && (best < 0 || items[i].cap < items[(size_t)best].cap)) best = (int)i;
			if (best >= 0)
			{
				Item it = items[(size_t)best];
				items.erase(items.begin() + best);
				held -= it.cap;
				cap = it.cap;
				return it.p;
			}
		}
		cap = bytes + bytes / 16 + 64;
		return malloc(cap);
	}
	void give(void* p, size_t cap)
	{
		if (!p) return;
		{
			std::lock_guard<std::mutex> g(lock);
			if (cap >= (1u << 20) && held + cap <= limit() && items.size() < 16) { items.push_back(Item{p, cap}); held += cap; return; }
		}
		free(p);
	}
};
static BufferPool& resultPool() { static BufferPool* pool = new BufferPool();  return *pool; }    // (never destroyed: results may be freed at exit)
struct PooledBuffer
{
	void* p = nullptr;
	size_t cap = 0;
	void* get(size_t bytes) { resultPool().give(p, cap); p = resultPool().take(bytes, cap); return p; }
	~PooledBuffer() { resultPool().give(p, cap); }
};

struct ResultsOwner
{
	ga_results_t pub;
	std::vector<ga_read_result_t> reads;
	std::vector<ga_mapping_t> mappings;
	std::vector<char> edits;
	std::vector<ga_trace_item_t> trace;
	// the arrays of a whole batch (hundreds of MB, not cleared first): from the process-wide pool below, back to it with the results
	PooledBuffer allReads, allMappings, allTrace;
	std::shared_ptr<char> seqKeep;        // edit_bytes of a batch's results = the batch's copy of the reads, kept alive here
};

int mapDeviceStatus(int s)
{
	switch (s)
	{
		case GA_OK: return GA_S_OK;
		case GA_ASSERTION: return GA_S_ASSERTION;
		case GA_UNSUPPORTED_BAND: return GA_S_UNSUPPORTED_BAND;
		case GA_UNSUPPORTED_CYCLE: return GA_S_UNSUPPORTED_CYCLE;
		case GA_UNSUPPORTED_RAMP: return GA_S_UNSUPPORTED_RAMP;
		case GA_CAP_NODES: case GA_CAP_COLS: case GA_CAP_ARENA: case GA_CAP_TRACE: case GA_CAP_HEAP: case GA_PUNT: return GA_S_CAPACITY;
		default: return GA_E_DEVICE;
	}
}

// traceToAlignment (GraphAligner.h:782-847).  Node indices come straight from the device trace.
// std::string::substr's clamping (the reference builds the pieces with substr, GraphAligner.h:829,845)
static SeqSpan spanOf(const ReadSeq& s, uint64_t pos, uint64_t len)
{
	if (pos > s.size()) pos = s.size();
	return SeqSpan{pos, std::min<uint64_t>(len, s.size() - pos)};
}

Partial toMappings(const ga_graph& g, const ReadSeq& sequence, int32_t score, const Trace& trace)
{
	Partial res;
	res.score = score;
	res.failed = false;
	if (trace.empty()) { res.failed = true; return res; }
	size_t pos = 0;
	uint32_t oldNode = trace[0].node;
	while (oldNode == 0)                                    // dummyNodeStart
	{
		pos++;
		if (pos == trace.size()) { res.failed = true; res.score = std::numeric_limits<int32_t>::max(); return res; }
		oldNode = trace[pos].node;
	}
	const uint64_t dummyEndAsIndex = g.bases.size() - 1;   // (sic) the reference compares a node index with a column index (:802,816)
	if (oldNode == dummyEndAsIndex) { res.failed = true; res.score = std::numeric_limits<int32_t>::max(); return res; }
	int rank = 0;
	ga_mapping_t m;
	memset(&m, 0, sizeof(m));
	m.rank = rank; m.node_id = g.ids[oldNode]; m.is_reverse = g.reverse[oldNode]; m.offset = trace[pos].offset;
	Pos nodeStart = trace[pos], nodeEnd = trace[pos], beforeNode = trace[pos];
	auto column = [&](const Pos& p) { return g.nodeStart[p.node] + p.offset; };
	for (; pos < trace.size(); pos++)
	{
		if (trace[pos].node == dummyEndAsIndex) break;
		if (trace[pos].node == oldNode) { nodeEnd = trace[pos]; continue; }
		m.from_length = (int64_t)(column(nodeEnd) - column(nodeStart) + 1);
		m.to_length = (int64_t)(nodeEnd.row - beforeNode.row);
		res.maps.push_back(m);
		res.seqs.push_back(spanOf(sequence, nodeStart.row, nodeEnd.row - beforeNode.row));
		oldNode = trace[pos].node;
		beforeNode = nodeEnd;
		nodeStart = trace[pos];
		nodeEnd = trace[pos];
		rank++;
		memset(&m, 0, sizeof(m));
		m.rank = rank; m.node_id = g.ids[oldNode]; m.is_reverse = g.reverse[oldNode];
	}
	m.from_length = (int64_t)(column(nodeEnd) - column(nodeStart));        // no +1 on the last mapping (:843)
	m.to_length = (int64_t)(nodeEnd.row - beforeNode.row);
	res.maps.push_back(m);
	res.seqs.push_back(spanOf(sequence, nodeStart.row, nodeEnd.row - beforeNode.row));
	return res;
}

// mergeAlignments (GraphAligner.h:648-688): backward part first, then forward
Partial mergePartials(const ga_graph& g, const Partial& first, const Partial& second)
{
	if (first.failed) return second;
	if (second.failed) return first;
	if (first.maps.empty()) return second;
	if (second.maps.empty()) return first;
	Partial out;
	out.failed = false;
	out.maps = first.maps;
	out.seqs = first.seqs;
	out.score = first.score + second.score;
	size_t startAt = 0;
	const ga_mapping_t& a = first.maps.back();
	const ga_mapping_t& b = second.maps.front();
	uint32_t an = g.lookup.at(a.node_id), bn = g.lookup.at(b.node_id);
	if (a.node_id == b.node_id && a.is_reverse == b.is_reverse) startAt = 1;
	else if (g.hasOutNeighbor(an, bn)) startAt = 0;
	for (size_t i = startAt; i < second.maps.size(); i++) { out.maps.push_back(second.maps[i]); out.seqs.push_back(second.seqs[i]); }
	return out;
}

bool readMatches(char readChar, char graphChar)
{
	uint8_t code = tables().rowCode[(uint8_t)readChar];
	uint8_t b = tables().baseCode[(uint8_t)graphChar];
	return b < 4 && ((code >> b) & 1);
}

// getTraceInfoInner (GraphAligner.h:718-780).  Returns false where the reference's characterMatch
// would assert: a diagonal step over a read character that is not IUPAC (e.g. 'U'), which can
// survive the backward split because ReverseComplement maps it (CommonUtils.cpp:85-88).
bool traceItemsInner(const ga_graph& g, const ReadSeq& seq, const Trace& tr, std::vector<ga_trace_item_t>& out)
{
	for (size_t i = 1; i < tr.size(); i++)
	{
		const Pos& now = tr[i];
		const Pos& old = tr[i - 1];
		bool sameColumn = now.node == old.node && now.offset == old.offset;
		bool diagonal = now.row != old.row;
		if (sameColumn)
		{
			bool selfLoop = now.row == old.row + 1 && g.nodeLen(now.node) == 1 && g.hasOutNeighbor(now.node, now.node);
			if (!selfLoop) diagonal = false;
		}
		ga_trace_item_t it;
		memset(&it, 0, sizeof(it));
		it.node_id = (int32_t)(g.ids[now.node] / 2);
		it.reverse = g.ids[now.node] % 2 == 1;
		it.offset = now.offset;
		it.read_pos = now.row;
		it.graph_char = g.baseChar(now.node, now.offset);
		it.read_char = seq[now.row];
		if (now.row == old.row) it.type = 4;
		else if (sameColumn && !diagonal) it.type = 3;
		else
		{
			if (tables().rowCode[(uint8_t)seq[now.row]] & GA_ROW_INVALID) return false;
			it.type = readMatches(seq[now.row], it.graph_char) ? 1 : 2;
		}
		out.push_back(it);
	}
	return true;
}

// getTraceInfo (GraphAligner.h:690-716)
bool traceItems(const ga_graph& g, const ReadSeq& seq, const Trace& bw, const Trace& fw, std::vector<ga_trace_item_t>& out)
{
	if (!bw.empty() && !traceItemsInner(g, seq, bw, out)) return false;
	if (!bw.empty() && !fw.empty())
	{
		ga_trace_item_t it;
		memset(&it, 0, sizeof(it));
		it.type = 5;
		it.node_id = (int32_t)(g.ids[fw[0].node] / 2);
		it.reverse = fw[0].node % 2 == 1;                   // node INDEX parity, as in the reference (:704)
		it.offset = fw[0].offset;
		it.read_pos = fw[0].row;
		it.graph_char = g.baseChar(fw[0].node, fw[0].offset);
		it.read_char = seq[fw[0].row];
		out.push_back(it);
	}
	if (!fw.empty() && !traceItemsInner(g, seq, fw, out)) return false;
	return true;
}

// addAlignmentNodes (GraphAligner.h:594-634)
void noteTried(std::vector<std::tuple<uint64_t, uint64_t, uint32_t>>& tried, const Trace& t)
{
	if (t.empty()) return;
	uint32_t oldNode = t[0].node;
	uint64_t lo = t[0].row, hi = t[0].row;
	for (size_t i = 1; i < t.size(); i++)
	{
		if (t[i].node != oldNode)
		{
			tried.emplace_back(lo, hi, oldNode);
			lo = t[i].row;
			oldNode = t[i].node;
		}
		hi = t[i].row;
	}
	tried.emplace_back(lo, hi, oldNode);
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

ga_graph_t* ga_graph_create(void) { return new ga_graph(); }
void ga_graph_destroy(ga_graph_t* g) { if (g) { delete g->device; delete g; } }
int ga_graph_add_node(ga_graph_t* g, int64_t id, const char* seq, size_t len, int rev) { return g ? addNode(g, id, seq, len, rev != 0) : GA_E_INVALID; }
int ga_graph_add_edge(ga_graph_t* g, int64_t from, int64_t to) { return g ? addEdge(g, from, to) : GA_E_INVALID; }
int ga_graph_set_neighbors(ga_graph_t* g, int64_t id, const int64_t* in, size_t nIn, const int64_t* out, size_t nOut)
{
	return g && (in || nIn == 0) && (out || nOut == 0) ? setNeighbors(g, id, in, nIn, out, nOut) : GA_E_INVALID;
}
int ga_graph_add_bigraph_node(ga_graph_t* g, int64_t id, const char* seq, size_t len)
{
	if (!g) return GA_E_INVALID;
	int s = addNode(g, id * 2, seq, len, false);
	if (s) return s;
	std::string rc = revcompACGT(seq, len);
	return addNode(g, id * 2 + 1, rc.data(), rc.size(), true);
}
int ga_graph_add_bigraph_edge(ga_graph_t* g, int64_t from, int fromStart, int64_t to, int toEnd)
{
	if (!g) return GA_E_INVALID;
	// BigraphToDigraph.cpp:32-56 / 70-104
	int64_t fromLeft = fromStart ? from * 2 : from * 2 + 1, fromRight = fromStart ? from * 2 + 1 : from * 2;
	int64_t toLeft = toEnd ? to * 2 : to * 2 + 1, toRight = toEnd ? to * 2 + 1 : to * 2;
	int s = addEdge(g, fromRight, toRight);
	if (s) return s;
	return addEdge(g, toLeft, fromLeft);
}
int ga_graph_finalize(ga_graph_t* g, int overlap) { return g ? finalizeGraph(g, overlap) : GA_E_INVALID; }

static int loadGfa(ga_graph_t* g, const char* text, size_t len, uint32_t maxNodeLen)
{
	// DirectedGraph::StreamGFAGraphFromFile (BigraphToDigraph.cpp:137-189): three passes over the lines
	if (!g || g->finalized) return GA_E_INVALID;
	std::vector<std::pair<const char*, size_t>> lines;
	for (size_t i = 0; i < len;)
	{
		size_t e = i;
		while (e < len && text[e] != '\n') e++;
		if (e > i) lines.emplace_back(text + i, e - i);
		i = e + 1;
	}
	int overlap = 0;
	int64_t maxId = 0;
	for (auto& l : lines)
	{
		if (l.first[0] == 'S' && maxNodeLen) { std::stringstream str(std::string(l.first, std::min<size_t>(l.second, 64))); std::string d; int64_t id = 0; str >> d >> id; maxId = std::max(maxId, id); }
		if (l.first[0] != 'L') continue;
		std::stringstream str(std::string(l.first, l.second));
		std::string d1, d2, d3, d4, d5, ov;
		str >> d1 >> d2 >> d3 >> d4 >> d5 >> ov;
		if (ov.size() < 2) return GA_E_INVALID;
		int o = std::stoi(ov.substr(0, ov.size() - 1));
		if (!(overlap == 0 || overlap == o)) return GA_E_INVALID;
		overlap = o;
	}
	if (maxNodeLen && overlap != 0) return GA_E_INVALID;       // pieces of a node overlap by nothing: splitting is for blunt graphs
	// first / last piece of every split node: the end an edge attaches to
	std::unordered_map<int64_t, std::pair<int64_t, int64_t>> ends;
	int64_t nextId = maxId + 1;
	for (auto& l : lines)
	{
		if (l.first[0] != 'S') continue;
		std::stringstream str(std::string(l.first, l.second));
		std::string d, seq;
		int64_t id;
		str >> d >> id >> seq;
		if ((int)seq.size() <= overlap) return GA_E_INVALID;
		if (maxNodeLen && seq.size() > maxNodeLen)
		{
			// a chain of pieces, the first keeping the node's id; the reverse strand's nodes follow from the pieces like any node's
			int64_t prev = -1;
			for (size_t at = 0; at < seq.size(); at += maxNodeLen)
			{
				const size_t n = std::min<size_t>(maxNodeLen, seq.size() - at);
				const int64_t pid = at == 0 ? id : nextId++;
				int st = ga_graph_add_bigraph_node(g, pid, seq.data() + at, n);
				if (st) return st;
				g->pieces[pid] = ga_graph::Piece{id, at, n, seq.size()};
				if (prev >= 0) { st = ga_graph_add_bigraph_edge(g, prev, 0, pid, 0); if (st) return st; }
				prev = pid;
			}
			ends[id] = std::make_pair(id, prev);
			continue;
		}
		size_t keep = seq.size() - overlap;
		int s = addNode(g, id * 2, seq.data(), keep, false);
		if (s) return s;
		std::string rc = revcompACGT(seq.data(), seq.size());
		s = addNode(g, id * 2 + 1, rc.data(), keep, true);
		if (s) return s;
	}
	for (auto& l : lines)
	{
		if (l.first[0] != 'L') continue;
		std::stringstream str(std::string(l.first, l.second));
		std::string d, fs, te;
		int64_t from, to;
		str >> d >> from >> fs >> to >> te;
		if ((fs != "+" && fs != "-") || (te != "+" && te != "-")) return GA_E_INVALID;
		// an edge leaves `from` at its end ("+") or its start ("-") and enters `to` at its start ("+") or its end ("-")
		auto ef = ends.find(from), et = ends.find(to);
		if (ef != ends.end()) from = fs == "-" ? ef->second.first : ef->second.second;
		if (et != ends.end()) to = te == "-" ? et->second.second : et->second.first;
		int s = ga_graph_add_bigraph_edge(g, from, fs == "-", to, te == "-");
		if (s) return s;
	}
	return finalizeGraph(g, overlap);
}

int ga_graph_load_gfa(ga_graph_t* g, const char* text, size_t len) { return loadGfa(g, text, len, 0); }
int ga_graph_load_gfa_split(ga_graph_t* g, const char* text, size_t len, uint32_t max_node_len) { return max_node_len ? loadGfa(g, text, len, max_node_len) : GA_E_INVALID; }

int ga_graph_split_lookup(const ga_graph_t* g, int64_t bigraph_id, int64_t* orig_id, uint64_t* start, uint64_t* orig_len)
{
	if (!g) return GA_E_INVALID;
	auto it = g->pieces.find(bigraph_id);
	if (it == g->pieces.end()) { if (orig_id) *orig_id = bigraph_id; if (start) *start = 0; if (orig_len) *orig_len = 0; return 1; }
	if (orig_id) *orig_id = it->second.orig;
	if (start) *start = it->second.start;
	if (orig_len) *orig_len = it->second.origLen;
	return GA_S_OK;
}

int ga_graph_upload(ga_graph_t* g, int device)
{
	if (!g) return GA_E_INVALID;
	if (!g->finalized) return GA_E_NOT_FINALIZED;
	delete g->device;
	int status = GA_S_OK;
	g->device = ga_backend_upload_graph(g->flat, g->hmm, device, &status);
	return g->device ? GA_S_OK : (status ? status : GA_E_DEVICE);
}
int64_t ga_graph_node_count(const ga_graph_t* g) { return g ? g->nodeCount() : 0; }
int64_t ga_graph_bp(const ga_graph_t* g) { return g ? (int64_t)g->bases.size() : 0; }

int ga_batch_prepare(const ga_graph_t* g, const ga_read_t* reads, size_t nReads, const ga_seed_t* seeds, const size_t* seedOffsets,
                     int initialBandwidth, int rampBandwidth, uint32_t flags, ga_batch_t** out)
{
	if (!g || !out || (!reads && nReads) || !seedOffsets) return GA_E_INVALID;
	if (!g->finalized) return GA_E_NOT_FINALIZED;
	if (!g->device) return GA_E_NO_DEVICE;
	const CharTables& T = tables();
	ga_batch* b = new ga_batch();
	b->g = g;
	b->flags = flags;
	b->cfg.initial_bw = initialBandwidth;
	b->cfg.ramp_bw = rampBandwidth;
	b->reads.resize(nReads);
	b->names.resize(nReads);
	b->seqs.resize(nReads);
	const auto tp0 = std::chrono::steady_clock::now();
	auto pad64 = [](uint64_t n) { return (n + W - 1) / W * W; };
	// the jobs are planned first (sizes and offsets only); their row codes, one byte per read base, are written afterwards by all host threads
	typedef ga_batch::RowFill RowFill;
	std::vector<RowFill>& fills = b->fills;
	uint64_t rowsTotal = 0;
	{
		// the batch keeps its own copy of the reads (the caller's buffers may go away before ga_batch_collect): ONE buffer, so that the
		// copy is a few large first-touch faults (huge pages where the system gives them) instead of one allocation per read, filled
		// on the host threads
		std::vector<uint64_t> at(nReads + 1, 0);
		for (size_t i = 0; i < nReads; i++) at[i + 1] = at[i] + reads[i].length;
		b->seqBufBytes = (size_t)at[nReads] + 64;
		// (from the back end: the product hands out pinned blocks of a pool, so that the upload of the reads is one DMA transfer)
		b->seqKeep = g->device->hostBlock(b->seqBufBytes);
		if (!b->seqKeep) { delete b; return GA_E_INVALID; }
		b->seqBuf = b->seqKeep.get();
		size_t nThreads = usableCores();
		if (const char* e = getenv("GA_HOST_THREADS")) nThreads = (size_t)atoi(e);
		nThreads = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nThreads, 16), nReads / 256 + 1));
		std::vector<std::thread> pool;
		const size_t per = (nReads + nThreads - 1) / nThreads;
		for (size_t t = 0; t < nThreads; t++)
		{
			const size_t lo = std::min(nReads, t * per), hi = std::min(nReads, lo + per);
			if (lo < hi) pool.emplace_back([&, lo, hi]() {
				for (size_t i = lo; i < hi; i++)
				{
					if (reads[i].length) memcpy(b->seqBuf + at[i], reads[i].sequence, reads[i].length);
					b->seqs[i] = ReadSeq(b->seqBuf + at[i], reads[i].length);
					b->names[i] = reads[i].name ? reads[i].name : "";
				}
			});
		}
		for (auto& th : pool) th.join();
	}
	const auto tpc = std::chrono::steady_clock::now();
	{
		const size_t nSeeds = seedOffsets[nReads] - seedOffsets[0];
		b->seeds.reserve(nSeeds);
		b->jobs.reserve(2 * nSeeds);
		fills.reserve(2 * nSeeds);
	}
	for (size_t i = 0; i < nReads; i++)
	{
		const ReadSeq& seq = b->seqs[i];
		b->reads[i].firstSeed = b->seeds.size();
		b->reads[i].nSeeds = seedOffsets[i + 1] - seedOffsets[i];
		for (size_t k = seedOffsets[i]; k < seedOffsets[i + 1]; k++)
		{
			SeedPlan sp;
			sp.pos = seeds[k].read_pos;
			const int64_t id = seeds[k].node_id;
			auto plain = g->lookup.find(id * 2);
			auto flipped = g->lookup.find(id * 2 + 1);
			if (plain == g->lookup.end() || flipped == g->lookup.end()) { sp.early = GA_S_BAD_SEED; b->seeds.push_back(sp); continue; }
			sp.seedNodeIndex = plain->second;
			// getSplitAlignment (GraphAligner.h:2969-3024)
			if (!(sp.pos < seq.size())) { sp.early = GA_S_ASSERTION; b->seeds.push_back(sp); continue; }
			uint32_t fwNode = seeds[k].reverse ? flipped->second : plain->second;
			uint32_t bwNode = seeds[k].reverse ? plain->second : flipped->second;
			if (g->nodeLen(fwNode) != g->nodeLen(bwNode)) { sp.early = GA_S_ASSERTION; b->seeds.push_back(sp); continue; }
			sp.valid = true;
			if (sp.pos > 0)
			{
				if (seq.size() < sp.pos + (uint64_t)g->dbgOverlap) { sp.early = GA_S_ASSERTION; sp.valid = false; b->seeds.push_back(sp); continue; }
				uint64_t n = sp.pos + g->dbgOverlap;
				// ReverseComplement asserts on anything outside its table, eagerly over the whole prefix (CommonUtils.cpp:60-136)
				bool bad = false;
				for (uint64_t r = 0; r < n; r++) if (!T.complement[(uint8_t)seq[r]]) { bad = true; break; }
				if (bad) { sp.early = GA_S_ASSERTION; sp.valid = false; b->seeds.push_back(sp); continue; }
				GaJob job;
				job.rows_off = rowsTotal;
				job.n_rows = (uint32_t)pad64(n);
				job.seed_node = bwNode;
				job.trace_rows = (uint32_t)(n - g->dbgOverlap); job.reserved = 0;       // = sp.pos (:3069)
				rowsTotal += job.n_rows;
				fills.push_back(RowFill{i, job.rows_off, n, job.n_rows, 0, true});
				sp.bwJob = (int64_t)b->jobs.size();
				b->jobs.push_back(job);
			}
			if (sp.pos < seq.size() - 1)
			{
				uint64_t n = seq.size() - sp.pos;
				GaJob job;
				job.rows_off = rowsTotal;
				job.n_rows = (uint32_t)pad64(n);
				job.seed_node = fwNode;
				job.trace_rows = n >= (uint64_t)g->dbgOverlap ? (uint32_t)(n - g->dbgOverlap) : 0u; job.reserved = 0;   // (:3051)
				rowsTotal += job.n_rows;
				fills.push_back(RowFill{i, job.rows_off, n, job.n_rows, sp.pos, false});
				sp.fwJob = (int64_t)b->jobs.size();
				b->jobs.push_back(job);
			}
			b->seeds.push_back(sp);
		}
	}
	for (const GaJob& j : b->jobs)
	{
		b->cfg.max_rows = std::max(b->cfg.max_rows, j.n_rows);
		b->cfg.max_slices = std::max(b->cfg.max_slices, j.n_rows / W);
	}
	b->rowsTotal = rowsTotal;
	const auto tp1 = std::chrono::steady_clock::now();
	b->eqWords = (rowsTotal / W + 1) * 5;       // match words per slice for the lanes = reads kernel
	b->readInvalid.reset(new std::atomic<uint8_t>[nReads + 1]);
	for (size_t i = 0; i <= nReads; i++) b->readInvalid[i].store(0, std::memory_order_relaxed);
	// The product's back end builds the match words itself, by a kernel over the uploaded reads (the words are 0.6 bytes per base of
	// table look-ups and bit gathering: 26 ms on 16 host threads for the benchmark batch, nothing on the device); the host builder
	// below is what the host emulation of tests/emul gets its words from.
	const bool deviceWords = g->device->buildsMatchWords();
	std::vector<GaEqFill> eqFills;
	if (deviceWords)
	{
		eqFills.resize(fills.size());
		for (size_t k = 0; k < fills.size(); k++)
		{
			const RowFill& f = fills[k];
			eqFills[k] = GaEqFill{(uint64_t)(b->seqs[f.read].data() - b->seqBuf), f.off / W, (uint32_t)f.pos, (uint32_t)f.n, (uint32_t)f.padded, f.backward ? 1u : 0u};
		}
	}
	else
	{
	b->eq.reset(new uint64_t[b->eqWords]);
	for (int k = 0; k < 5; k++) b->eq[b->eqWords - 5 + k] = 0;              // (the slack slice behind the last job)
	forEachFill(b, [&](const RowFill& f, const ReadSeq& seq) {
		// 64 row codes at a time, straight into the slice's match words
		const uint8_t padCode = T.rowCode[(uint8_t)'N'];
		const uint8_t* fw = (const uint8_t*)seq.data() + f.pos;
		const uint8_t* bw = (const uint8_t*)seq.data() + f.n - 1;
		uint8_t buf[W];
		for (uint64_t r0 = 0; r0 < f.padded; r0 += W)
		{
			const uint64_t full = r0 + W <= f.n ? (uint64_t)W : r0 < f.n ? f.n - r0 : 0;
			if (f.backward) for (uint64_t k = 0; k < full; k++) buf[k] = T.rowCodeRc[*(bw - (r0 + k))];
			else for (uint64_t k = 0; k < full; k++) buf[k] = T.rowCode[fw[r0 + k]];
			for (uint64_t k = full; k < (uint64_t)W; k++) buf[k] = padCode;
			uint64_t* words = b->eq.get() + (f.off + r0) / W * 5;
			ga_build_eq_words(buf, W, words);
			if (words[4] & 8u) { b->anyInvalidRow.store(1, std::memory_order_relaxed); b->readInvalid[f.read].store(1, std::memory_order_relaxed); }
		}
	});
	}
	// Node runs instead of moves from the traceback when no result of the batch needs a cell list: no TraceItem lists wanted, every
	// read with one seed at its first base (one forward job: no backward part to mirror, no later seed to test against the cells
	// of an earlier one, GraphAligner.h:423-429), no character whose TraceItem the reference would assert on
	{
		// ... and a graph of long nodes (the shape the lanes = reads kernel is the first pass for): a path then changes node every few
		// dozen rows and its runs are a fraction of its moves; on graphs chopped into short nodes moves are the smaller output
		const double meanNode = g->nodeCount() > 2 ? (double)g->bases.size() / (double)(g->nodeCount() - 2) : 64.0;
		// (with the words built by the back end, whether a row is such a character is known when the batch has been created: the back
		// end clears the flag then)
		bool runs = (flags & GA_F_TRACE) == 0 && b->anyInvalidRow.load() == 0 && meanNode >= 40 && !(getenv("GA_RUNS") && atoi(getenv("GA_RUNS")) == 0);
		for (size_t i = 0; i < nReads && runs; i++) if (b->reads[i].nSeeds > 1) runs = false;
		for (const SeedPlan& sp : b->seeds) if (sp.bwJob >= 0 || (sp.fwJob >= 0 && sp.pos != 0)) { runs = false; break; }
		b->cfg.emit_runs = runs ? 1u : 0u;
	}
	int status = GA_S_OK;
	const auto tp2 = std::chrono::steady_clock::now();
	GaEqSource src;
	src.seq = b->seqBuf; src.seqBytes = b->seqBufBytes;
	src.fills = eqFills.data(); src.nFills = eqFills.size();
	src.rowCode = T.rowCode; src.rowCodeRc = T.rowCodeRc; src.padCode = T.rowCode[(uint8_t)'N'];
	b->dev = ga_backend_create_batch(g->device, [b]() -> const std::vector<uint8_t>& { buildRows(b); return b->rows; }, b->eq.get(), deviceWords ? &src : nullptr, b->eqWords,
	                                 b->jobs, b->cfg, &status);
	if (!b->dev) { delete b; return status ? status : GA_E_DEVICE; }
	if (const std::vector<uint8_t>* bad = b->dev->invalidFills())
	{
		for (size_t k = 0; k < fills.size() && k < bad->size(); k++)
			if ((*bad)[k]) { b->anyInvalidRow.store(1, std::memory_order_relaxed); b->readInvalid[fills[k].read].store(1, std::memory_order_relaxed); }
	}
	b->cfg.emit_runs = b->dev->emittingRuns() ? 1u : 0u;
	if (getenv("GA_DEBUG_COLLECT"))
	{
		const auto tp3 = std::chrono::steady_clock::now();
		auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		fprintf(stderr, "graphaligner_amd: prepare: read copies %.1f ms, job plan %.1f ms, match words %.1f ms, device batch (alloc + upload) %.1f ms\n", ms(tp0, tpc), ms(tpc, tp1), ms(tp1, tp2), ms(tp2, tp3));
	}
	*out = b;
	return GA_S_OK;
}

int ga_batch_run(ga_batch_t* b)
{
	if (!b || !b->dev) return GA_E_INVALID;
	int s = b->dev->run();
	b->ran = s == GA_S_OK;
	return s;
}

int ga_batch_stats(const ga_batch_t* b, ga_batch_stats_t* out)
{
	if (!b || !out || !b->dev) return GA_E_INVALID;
	memset(out, 0, sizeof(*out));
	GaRunStats st = b->dev->stats();
	out->n_jobs = b->jobs.size();
	out->kernel_ms = st.kernel_ms;
	out->main_kernel_ms = st.main_ms;
	out->main_variant = st.main_variant;
	out->slots = st.slots;
	out->waves_per_cu = st.waves_per_cu;
	out->scratch_bytes = st.scratch_bytes;
	out->jobs_retried = st.jobs_retried;
	for (int k = 0; k < 8; k++) out->stamps[k] = st.stamps[k];
	out->column_updates = b->columnUpdates;
	out->reserved = (int32_t)b->cfg.emit_runs;
	out->slices = b->slicesRun;
	return GA_S_OK;
}

int ga_batch_collect(ga_batch_t* b, ga_results_t** out)
{
	if (!b || !out || !b->dev || !b->ran) return GA_E_INVALID;
	const ga_graph& g = *b->g;
	std::vector<GaJobOut> outs;
	const uint8_t* moves = nullptr;
	uint64_t nMoveBytes = 0;
	const auto t0 = std::chrono::steady_clock::now();
	int s = b->dev->fetch(outs, &moves, &nMoveBytes);
	if (s) return s;
	const auto t1 = std::chrono::steady_clock::now();
	b->columnUpdates = 0;
	b->slicesRun = 0;
	for (const GaJobOut& o : outs) b->slicesRun += o.n_run;         // (column updates are summed per read below: seeds the reference would skip do not count)
	ResultsOwner* R = new ResultsOwner();
	const int32_t kMax = std::numeric_limits<int32_t>::max();

	// The device hands back, per job, the cell the traceback starts in and one byte per backward move
	// (GA_MOVE_*; a move out of a node's first column names the in-neighbour it enters).  Replaying
	// them gives the reference's trace, which runs from row 0 upwards (getTraceFromTable :949-952).
	std::atomic<int> overflow{0};
	auto deviceTrace = [&](int64_t job) {
		Trace t;
		const GaJobOut& o = outs[job];
		if (o.n_valid == 0) return t;
		if (o.reserved3 == 1) { overflow.store(1); return t; }      // (node runs, not moves: only batches that need no cell list get them)
		const uint8_t* mv = moves + o.trace_off;
		t.resize((size_t)o.trace_len + 1);
		Pos p{o.start_node, o.start_offset, o.start_row};
		size_t at = t.size() - 1;
		t[at] = p;
		for (uint32_t i = 0; i < o.trace_len; i++)
		{
			const int code = mv[i] & 3, via = mv[i] >> 2;
			if (code != GA_MOVE_LEFT) p.row -= 1;
			if (code != GA_MOVE_UP)
			{
				if (p.offset > 0) p.offset -= 1;
				else
				{
					p.node = g.inNeighbor(p.node, (uint32_t)via);
					p.offset = g.nodeLen(p.node) - 1;
				}
			}
			t[--at] = p;
		}
		return t;
	};

	// Every read's results go straight into the batch's arrays, at places fixed before the threads start: a read's path has at most
	// (moves that left a node through its first column, counted by the traceback) + 1 node runs per job, its edit sequences are pieces
	// of the read, its trace items at most one per move.  The arrays have gaps where a read needs less; first_mapping / n_mappings,
	// edit_seq_off and first_trace / n_trace say where each read's entries are.
	const size_t nReadsAll = b->reads.size();
	const bool wantTraceAll = (b->flags & GA_F_TRACE) != 0;
	std::vector<uint64_t> mapAt(nReadsAll + 1, 0), editAt(nReadsAll + 1, 0), traceAt(nReadsAll + 1, 0);
	for (size_t ri = 0; ri < nReadsAll; ri++)
	{
		const ReadPlan& rp = b->reads[ri];
		uint64_t runs = 1, items = 0;
		for (size_t k = rp.firstSeed; k < rp.firstSeed + rp.nSeeds; k++)
			for (int64_t job : {b->seeds[k].bwJob, b->seeds[k].fwJob})
				if (job >= 0 && outs[job].status == GA_OK) { runs += (uint64_t)outs[job].n_node_steps + 1; items += (uint64_t)outs[job].trace_len + 2; }
		mapAt[ri + 1] = mapAt[ri] + runs;
		editAt[ri + 1] = editAt[ri] + b->seqs[ri].size();
		traceAt[ri + 1] = traceAt[ri] + (wantTraceAll ? items : 0);
	}
	ga_read_result_t* const allReads = (ga_read_result_t*)R->allReads.get((nReadsAll + 1) * sizeof(ga_read_result_t));
	ga_mapping_t* const allMappings = (ga_mapping_t*)R->allMappings.get((mapAt[nReadsAll] + 1) * sizeof(ga_mapping_t));
	// (no copy of the edit sequences: they are pieces of the reads, and the results share the batch's read buffer)
	const char* const allEdits = b->seqBuf;
	R->seqKeep = b->seqKeep;
	ga_trace_item_t* const allTrace = (ga_trace_item_t*)R->allTrace.get((traceAt[nReadsAll] + 1) * sizeof(ga_trace_item_t));
	if (!allReads || !allMappings || !allEdits || !allTrace) { delete R; b->dev->fetchDone(); return GA_E_INVALID; }
	std::atomic<uint64_t> columnUpdatesAll{0}, forwardOnlyReads{0};

	// The common shape -- one seed at the read's first base, so one forward job, and no TraceItem list wanted -- without the detour
	// over a cell list: the moves are replayed once, node runs are noted as they end, and the mappings are written from the runs.
	// Same result as the general code below (getPiecewiseTracesFromSplit :3039-3098 without a backward part, traceToAlignment
	// :782-847, mergeAlignments with a failed first part); anything unusual (a dummy node on the path) is left to that code.
	struct NodeRun { uint32_t node, firstOffset, lastOffset; uint64_t firstRow, lastRow; };
	auto forwardOnly = [&](size_t ri, ga_read_result_t& rr) -> bool {
		const ReadPlan& rp = b->reads[ri];
		if (rp.nSeeds != 1 || wantTraceAll) return false;
		const SeedPlan& sp = b->seeds[rp.firstSeed];
		if (sp.early != GA_S_OK || sp.bwJob >= 0 || sp.fwJob < 0 || sp.pos != 0) return false;
		const GaJobOut& o = outs[sp.fwJob];
		if (o.status != GA_OK) return false;
		rr.reserved = (int32_t)o.reserved2;                                      // which kernel pass finished the job (0 = the first)
		const ReadSeq& seq = b->seqs[ri];
		// (the reference's eager TraceItem pass asserts on characters outside IUPAC: with the one forward job from the first base the
		// job's rows are the whole read, and the match-word pass has noted whether one of them is such a character)
		if (b->readInvalid[ri].load(std::memory_order_relaxed)) return false;
		if (o.n_valid == 0) { rr.column_updates += o.n_columns; rr.status = GA_S_OK; return true; }     // nothing kept: both parts fail (rr stays failed)
		const uint64_t traceable = seq.size() - sp.pos - g.dbgOverlap;
		const uint64_t dummyEndAsIndex = g.bases.size() - 1;
		thread_local std::vector<NodeRun> runs;
		runs.clear();
		if (o.reserved3 == 1)
		{
			// the traceback's own node runs (five words each), from the read's end to its start like the replay's
			const uint32_t* rec = (const uint32_t*)(moves + o.trace_off);
			runs.resize(o.trace_len);
			for (uint32_t k = 0; k < o.trace_len; k++, rec += 5) runs[k] = NodeRun{rec[0], rec[1], rec[3], rec[2], rec[4]};
		}
		else
		{
		const uint8_t* mv = moves + o.trace_off;
		const uint32_t nMoves = o.trace_len;
		Pos p{o.start_node, o.start_offset, o.start_row};
		auto step = [&](uint8_t m) {
			const int code = m & 3, via = m >> 2;
			if (code != GA_MOVE_LEFT) p.row -= 1;
			if (code != GA_MOVE_UP)
			{
				if (p.offset > 0) p.offset -= 1;
				else
				{
					p.node = g.inNeighbor(p.node, (uint32_t)via);
					p.offset = g.nodeLen(p.node) - 1;
				}
			}
		};
		uint32_t i = 0;
		while (p.row >= traceable && i < nMoves) step(mv[i++]);                  // the trace's tail beyond the read's end is dropped (:3051-3055)
		if (p.row < traceable)
		{
			// a run is opened at its last cell (the first one met on the way back) and closed, with its first cell, when the path
			// leaves the node; eight diagonal steps inside a node are taken at once
			runs.push_back(NodeRun{p.node, p.offset, p.offset, p.row, p.row});
			const uint64_t eightDiagonals = 0x0101010101010101ull * (uint64_t)GA_MOVE_DIAG;
			while (i < nMoves)
			{
				if (i + 8 <= nMoves && p.offset >= 8)
				{
					uint64_t w;
					memcpy(&w, mv + i, 8);
					if (w == eightDiagonals) { p.offset -= 8; p.row -= 8; i += 8; continue; }
				}
				const Pos before = p;
				step(mv[i++]);
				if (p.node != before.node)
				{
					runs.back().firstOffset = before.offset; runs.back().firstRow = before.row;
					runs.push_back(NodeRun{p.node, p.offset, p.offset, p.row, p.row});
				}
			}
			runs.back().firstOffset = p.offset; runs.back().firstRow = p.row;
		}
		}
		// traceToAlignment's dummy nodes (:786-800, 816): cells of the start dummy at the read's start are skipped, a path that begins
		// in the end dummy fails, and the path is cut where it first enters the end dummy
		while (!runs.empty() && runs.back().node == 0) runs.pop_back();
		if (!runs.empty() && runs.back().node == dummyEndAsIndex) runs.clear();
		for (size_t k = runs.size(); k-- > 0;)
			if (runs[k].node == dummyEndAsIndex) { runs.erase(runs.begin(), runs.begin() + (long)k + 1); break; }
		rr.column_updates += o.n_columns;
		rr.status = GA_S_OK;
		if (runs.empty()) return true;                                         // an empty trace fails (:786-790)
		const size_t nRuns = runs.size();
		if (nRuns > mapAt[ri + 1] - mapAt[ri]) { overflow.store(1); rr.status = GA_E_DEVICE; return true; }
		// the runs were noted from the read's end to its start
		uint64_t editTop = editAt[ri];
		uint64_t beforeRow = runs[nRuns - 1].firstRow;
		for (size_t k = 0; k < nRuns; k++)
		{
			const NodeRun& r = runs[nRuns - 1 - k];
			ga_mapping_t m;
			memset(&m, 0, sizeof(m));
			m.rank = (int32_t)k;
			m.node_id = g.ids[r.node];
			m.is_reverse = g.reverse[r.node];
			m.offset = k == 0 ? r.firstOffset : 0;
			m.from_length = (int64_t)r.lastOffset - (int64_t)r.firstOffset + (k + 1 < nRuns ? 1 : 0);        // no +1 on the last mapping (:843)
			m.to_length = (int64_t)(r.lastRow - beforeRow);
			const SeqSpan piece = spanOf(seq, r.firstRow, r.lastRow - beforeRow);
			m.edit_seq_off = (uint64_t)(seq.data() - allEdits) + piece.pos;
			if (piece.len > editAt[ri + 1] - editTop) { overflow.store(1); rr.status = GA_E_DEVICE; return true; }
			editTop += piece.len;
			allMappings[mapAt[ri] + k] = m;
			beforeRow = r.lastRow;
		}
		rr.failed = 0;
		rr.score = o.score;
		rr.n_mappings = nRuns;
		rr.n_trace = 0;
		rr.query_position = 0;
		rr.alignment_start = 0;
		rr.alignment_end = (uint64_t)o.n_valid * W;
		return true;
	};

	auto work = [&](size_t lo, size_t hi) {
	std::vector<ga_trace_item_t> items;
	uint64_t columnUpdates = 0;
	for (size_t ri = lo; ri < hi; ri++)
	{
		ga_read_result_t& rr = allReads[ri];
		memset(&rr, 0, sizeof(rr));
		rr.failed = 1;
		rr.score = kMax;
		rr.first_mapping = mapAt[ri];
		rr.first_trace = traceAt[ri];
		const ReadSeq& seq = b->seqs[ri];
		const ReadPlan& rp = b->reads[ri];
		if (rp.nSeeds == 0) { rr.status = GA_S_ASSERTION; continue; }          // assert(seedHits.size() > 0) (:412)
		if (forwardOnly(ri, rr)) { columnUpdates += rr.column_updates; forwardOnlyReads++; continue; }
		std::vector<std::tuple<uint64_t, uint64_t, uint32_t>> tried;
		bool have = false;
		uint64_t bestEstimate = 0, bestPos = 0;
		Trace bestFw, bestBw;
		int32_t bestFwScore = 0, bestBwScore = 0;
		int status = GA_S_OK;
		for (size_t k = rp.firstSeed; k < rp.firstSeed + rp.nSeeds && status == GA_S_OK; k++)
		{
			const SeedPlan& sp = b->seeds[k];
			if (sp.early == GA_S_BAD_SEED) { status = GA_S_BAD_SEED; break; }
			bool covered = false;
			for (auto& t : tried) if (std::get<0>(t) <= sp.pos && std::get<1>(t) >= sp.pos && std::get<2>(t) == sp.seedNodeIndex) { covered = true; break; }
			if (covered) continue;                                               // "seed already aligned" (:425-429)
			if (sp.early != GA_S_OK) { status = sp.early; break; }
			uint64_t nValidFw = 0, nValidBw = 0;
			for (int64_t job : {sp.bwJob, sp.fwJob})
			{
				if (job < 0) continue;
				rr.column_updates += outs[job].n_columns;
				rr.reserved = std::max<int32_t>(rr.reserved, (int32_t)outs[job].reserved2);
				if (outs[job].status != GA_OK) status = mapDeviceStatus(outs[job].status);
			}
			if (status != GA_S_OK) break;
			if (sp.fwJob >= 0) nValidFw = outs[sp.fwJob].n_valid;
			if (sp.bwJob >= 0) nValidBw = outs[sp.bwJob].n_valid;
			// getPiecewiseTracesFromSplit (:3039-3098)
			Trace fw, bw;
			int32_t fwScore = 0, bwScore = 0;
			if (sp.fwJob >= 0 && nValidFw > 0)
			{
				uint64_t traceable = seq.size() - sp.pos - g.dbgOverlap;
				fw = deviceTrace(sp.fwJob);
				fwScore = outs[sp.fwJob].score;
				while (!fw.empty() && fw.back().row >= traceable) fw.pop_back();
			}
			if (sp.bwJob >= 0 && nValidBw > 0)
			{
				uint64_t traceable = sp.pos;
				bw = deviceTrace(sp.bwJob);
				bwScore = outs[sp.bwJob].score;
				while (!bw.empty() && bw.back().row >= traceable) bw.pop_back();
				// reverseTrace (:3026-3037)
				std::reverse(bw.begin(), bw.end());
				uint64_t endRow = sp.pos - 1;
				for (auto& p : bw)
				{
					uint32_t other;
					if (g.reverseNode(p.node, other) != GA_S_OK || p.row > endRow) { status = GA_S_ASSERTION; break; }
					p.offset = g.nodeLen(other) - 1 - p.offset;
					p.node = other;
					p.row = endRow - p.row;
				}
				for (auto& p : fw) p.row += sp.pos;                            // only inside this branch (:3091-3094)
			}
			if (status != GA_S_OK) break;
			noteTried(tried, fw);
			noteTried(tried, bw);
			uint64_t estimate = (nValidFw + nValidBw) * W;
			if (!have || estimate > bestEstimate)
			{
				bestFw = std::move(fw); bestBw = std::move(bw); bestFwScore = fwScore; bestBwScore = bwScore;
				bestEstimate = estimate; bestPos = sp.pos; have = true;
			}
		}
		rr.status = status;
		columnUpdates += rr.column_updates;
		if (status != GA_S_OK || !have) continue;
		// the reference always builds the TraceItem list (:463) and characterMatch asserts on a
		// non-IUPAC read character; only reads that contain one need the scan when no trace is wanted
		bool wantTrace = wantTraceAll;
		bool suspicious = false;
		if (!wantTrace) for (char c : seq) if (tables().rowCode[(uint8_t)c] & GA_ROW_INVALID) { suspicious = true; break; }
		items.clear();
		if (wantTrace || suspicious)
		{
			bool fine = traceItems(g, seq, bestBw, bestFw, items);
			if (!wantTrace || !fine) items.clear();
			if (!fine) { rr.status = GA_S_ASSERTION; continue; }
		}
		Partial fwp = toMappings(g, seq, bestFwScore, bestFw);
		Partial bwp = toMappings(g, seq, bestBwScore, bestBw);
		if (fwp.failed && bwp.failed) continue;
		Partial merged = mergePartials(g, bwp, fwp);
		uint64_t editBytes = 0;
		for (const SeqSpan& piece : merged.seqs) editBytes += piece.len;
		if (merged.maps.size() > mapAt[ri + 1] - mapAt[ri] || editBytes > editAt[ri + 1] - editAt[ri] || items.size() > traceAt[ri + 1] - traceAt[ri])
		{
			overflow.store(1);              // (a bound above is wrong: fail loudly instead of writing past a read's place)
			rr.status = GA_E_DEVICE;
			continue;
		}
		rr.failed = 0;
		rr.score = merged.score;
		rr.n_mappings = merged.maps.size();
		uint64_t editTop = editAt[ri];
		for (size_t i = 0; i < merged.maps.size(); i++)
		{
			ga_mapping_t m = merged.maps[i];
			m.edit_seq_off = (uint64_t)(seq.data() - allEdits) + merged.seqs[i].pos;
			editTop += merged.seqs[i].len;
			allMappings[mapAt[ri] + i] = m;
		}
		rr.n_trace = items.size();
		if (!items.empty()) memcpy(allTrace + traceAt[ri], items.data(), items.size() * sizeof(items[0]));
		uint64_t lastAligned = !bestBw.empty() ? bestBw[0].row : bestPos;
		rr.query_position = lastAligned;
		rr.alignment_start = lastAligned;
		rr.alignment_end = lastAligned + bestEstimate;
	}
	columnUpdatesAll += columnUpdates;
	};
	// reads are independent: assembled on several host threads
	size_t nThreads = usableCores();
	if (const char* e = getenv("GA_HOST_THREADS")) nThreads = (size_t)atoi(e);
	nThreads = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nThreads, 64), nReadsAll / 64 + 1));
	{
		// (small pieces handed out from a counter: reads sorted by nothing in particular, but their lengths differ)
		std::atomic<size_t> next{0};
		const size_t piece = std::max<size_t>(16, nReadsAll / (nThreads * 16) + 1);
		std::vector<std::thread> pool;
		for (size_t t = 0; t < nThreads; t++)
			pool.emplace_back([&]() {
				while (true)
				{
					const size_t lo = next.fetch_add(piece);
					if (lo >= nReadsAll) break;
					work(lo, std::min(nReadsAll, lo + piece));
				}
			});
		for (auto& th : pool) th.join();
	}
	b->columnUpdates += columnUpdatesAll.load();
	b->dev->fetchDone();
	const auto t2 = std::chrono::steady_clock::now();
	if (overflow.load()) { delete R; return GA_E_DEVICE; }
	R->pub.n_reads = nReadsAll; R->pub.reads = allReads;
	R->pub.n_mappings = mapAt[nReadsAll]; R->pub.mappings = allMappings;
	R->pub.n_edit_bytes = b->seqBufBytes; R->pub.edit_bytes = allEdits;
	R->pub.n_trace = traceAt[nReadsAll]; R->pub.trace = allTrace;
	*out = &R->pub;
	if (getenv("GA_DEBUG_COLLECT"))
	{
		const auto t3 = std::chrono::steady_clock::now();
		auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		fprintf(stderr, "graphaligner_amd: collect: fetch %.1f ms (%zu trace bytes), assembly %.1f ms on %zu threads (%llu of %zu reads on the forward-only path), hand-over %.1f ms\n", ms(t0, t1), (size_t)nMoveBytes, ms(t1, t2), nThreads, (unsigned long long)forwardOnlyReads.load(), nReadsAll, ms(t2, t3));
	}
	return GA_S_OK;
}

void ga_batch_free(ga_batch_t* b) { delete b; }

void ga_results_free(ga_results_t* r)
{
	if (!r) return;
	delete reinterpret_cast<ResultsOwner*>(r);     // `pub` is the owner's first member
}

// results over a graph loaded with ga_graph_load_gfa_split, expressed on the nodes of the GFA file: consecutive mappings on pieces of
// one node become one mapping (lengths add up, the edit sequences are contiguous), offsets and trace items move to the node's coordinates
int ga_results_unsplit(const ga_graph_t* g, const ga_results_t* in, ga_results_t** out)
{
	if (!g || !in || !out) return GA_E_INVALID;
	ResultsOwner* R = new ResultsOwner();
	// (every mapping's edit sequence is copied: the pieces of merged mappings are joined here, whatever lies between them in `in`)
	R->edits.reserve(in->n_edit_bytes / 4 + 64);
	auto piece = [&](int64_t digraphId) -> const ga_graph::Piece* {
		auto it = g->pieces.find(digraphId / 2);
		return it == g->pieces.end() ? nullptr : &it->second;
	};
	for (size_t i = 0; i < in->n_reads; i++)
	{
		ga_read_result_t rr = in->reads[i];
		const size_t firstMap = R->mappings.size(), firstTrace = R->trace.size();
		const ga_graph::Piece* prevPiece = nullptr;
		for (uint64_t k = 0; k < rr.n_mappings; k++)
		{
			ga_mapping_t m = in->mappings[rr.first_mapping + k];
			const char* const editFrom = in->edit_bytes + m.edit_seq_off;
			const size_t editLen = (size_t)m.to_length;
			const ga_graph::Piece* p = piece(m.node_id);
			if (p)
			{
				const bool rev = (m.node_id & 1) != 0;
				if (R->mappings.size() > firstMap)
				{
					ga_mapping_t& last = R->mappings.back();
					const bool adjacent = prevPiece && prevPiece->orig == p->orig && last.node_id == p->orig * 2 + (rev ? 1 : 0) &&
					                      (rev ? p->start + p->len == prevPiece->start : prevPiece->start + prevPiece->len == p->start);
					if (adjacent)
					{
						last.from_length += m.from_length;
						last.to_length += m.to_length;
						R->edits.insert(R->edits.end(), editFrom, editFrom + editLen);
						prevPiece = p;
						continue;
					}
				}
				if (k == 0) m.offset = (int64_t)(rev ? p->origLen - (p->start + p->len) : p->start) + m.offset;
				m.node_id = p->orig * 2 + (rev ? 1 : 0);
			}
			prevPiece = p;
			m.rank = (int32_t)(R->mappings.size() - firstMap);
			m.edit_seq_off = R->edits.size();
			R->edits.insert(R->edits.end(), editFrom, editFrom + editLen);
			R->mappings.push_back(m);
		}
		for (uint64_t k = 0; k < rr.n_trace; k++)
		{
			ga_trace_item_t t = in->trace[rr.first_trace + k];
			auto it = g->pieces.find((int64_t)t.node_id);
			if (it != g->pieces.end())
			{
				const ga_graph::Piece& p = it->second;
				t.offset = (t.reverse ? p.origLen - (p.start + p.len) : p.start) + t.offset;
				t.node_id = (int32_t)p.orig;
			}
			R->trace.push_back(t);
		}
		rr.first_mapping = firstMap; rr.n_mappings = R->mappings.size() - firstMap;
		rr.first_trace = firstTrace; rr.n_trace = R->trace.size() - firstTrace;
		R->reads.push_back(rr);
	}
	R->pub.n_reads = R->reads.size(); R->pub.reads = R->reads.data();
	R->pub.n_mappings = R->mappings.size(); R->pub.mappings = R->mappings.data();
	R->pub.n_edit_bytes = R->edits.size(); R->pub.edit_bytes = R->edits.data();
	R->pub.n_trace = R->trace.size(); R->pub.trace = R->trace.data();
	*out = &R->pub;
	return GA_S_OK;
}

int ga_align_batch(const ga_graph_t* g, const ga_read_t* reads, size_t nReads, const ga_seed_t* seeds, const size_t* seedOffsets,
                   int initialBandwidth, int rampBandwidth, uint32_t flags, ga_results_t** out)
{
	ga_batch_t* b = nullptr;
	int s = ga_batch_prepare(g, reads, nReads, seeds, seedOffsets, initialBandwidth, rampBandwidth, flags, &b);
	if (s) return s;
	s = ga_batch_run(b);
	if (!s) s = ga_batch_collect(b, out);
	ga_batch_free(b);
	return s;
}

const char* ga_status_string(int s)
{
	switch (s)
	{
		case GA_S_OK: return "ok";
		case GA_S_ASSERTION: return "reference assertion";
		case GA_S_UNSUPPORTED_BAND: return "band >= 200000 bp and no device memory for the pass that carries the sparse method";
		case GA_S_BAD_SEED: return "seed node not in graph";
		case GA_S_CAPACITY: return "device buffer capacity";
		case GA_S_UNSUPPORTED_CYCLE: return "cyclic band left unresolved by the kernel ladder";
		case GA_S_UNSUPPORTED_RAMP: return "ramp redo left unresolved by the kernel ladder";
		case GA_E_INVALID: return "invalid argument";
		case GA_E_NO_DEVICE: return "no gfx950 device / graph not uploaded";
		case GA_E_DEVICE: return "device error";
		case GA_E_NOT_FINALIZED: return "graph not finalized";
	}
	return "unknown";
}
const char* ga_version(void) { return "graphaligner_amd 0.1 (gfx950)"; }

}  // extern "C"
