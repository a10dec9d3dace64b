// ga_backend_emul.cpp -- TEST-ONLY back end: runs the device extension program
// (graphaligner_amd/csrc/ga_kernel.h) on the host with every wave64 primitive emulated
// (GA_EMULATE in ga_wave.h), one job after the other.
//
// Purpose: this container has no GPU, so the exact device logic is checked here against the
// CPU oracle before it is run on a real MI355X (tests marked `gpu` repeat the same checks
// through the real library).  This file is linked only into tests/_build/libga_emul.so; the
// product library (graphaligner_amd/libgraphaligner_amd.so) is built from ga_device.hip and
// has no host execution path.
#define GA_EMULATE 1
#include <algorithm>
#include <cstring>
#include <memory>

#include "../../graphaligner_amd/csrc/ga_backend.h"
#include "../../graphaligner_amd/csrc/ga_kernel.h"
#include "../../graphaligner_amd/csrc/ga_lanes.h"
#include <cstdlib>
#include <cstdio>

namespace {

struct EmulGraph : GaBackendGraph
{
	GaFlatGraph flat;
	std::vector<uint32_t> nodeRec;
	GaHmmTables hmm;
	GaDevGraph dev;
};

struct EmulBatch : GaBackendBatch
{
	EmulGraph* g;
	GaRowsProvider rowsProvider;
	std::vector<uint8_t> rows;
	std::vector<uint64_t> eq;
	std::vector<GaJob> jobs;
	GaRunConfig cfg;
	std::vector<GaJobOut> outs;
	uint64_t lanesDone = 0;
	std::vector<uint8_t> pool;
	uint64_t poolTop = 0;
	uint64_t retried = 0;

	template <int MAXN, bool GENERAL, bool SPARSE = false> void runOne(uint32_t job, uint32_t capCols, uint64_t arenaWords, uint32_t traceCap)
	{
		std::vector<uint32_t> endA(capCols), endB(capCols), arena(arenaWords), sliceOff(cfg.max_slices + 1);
		std::vector<uint8_t> flags(cfg.max_slices + 1);
		std::vector<uint8_t> staging(traceCap + 64);
		std::vector<uint32_t> ckpt(cfg.max_slices + 2), below(cfg.max_slices + 1);
		const uint32_t maxBw = (uint32_t)std::max(std::max(cfg.initial_bw, cfg.ramp_bw), 1);
		std::vector<uint8_t> sparse(SPARSE ? gak::sparse_mem_bytes(maxBw) : 0);
		std::vector<uint32_t> ovr(SPARSE ? 2 * (cfg.max_slices + 2) : 0);
		gak::Slot slot{endA.data(), endB.data(), arena.data(), sliceOff.data(), flags.data(), staging.data(), ckpt.data(), below.data(), SPARSE ? sparse.data() : nullptr, SPARSE ? ovr.data() : nullptr, maxBw};
		GaLaunch L;
		memset(&L, 0, sizeof(L));
		L.graph = g->dev; L.hmm = &g->hmm; L.rows = rows.data(); L.jobs = jobs.data(); L.outs = outs.data();
		L.traces = pool.data(); L.trace_top = &poolTop; L.trace_pool_cap = pool.size();
		L.n_jobs = (uint32_t)jobs.size(); L.trace_cap = traceCap; L.cap_cols = capCols; L.max_slices = cfg.max_slices;
		L.arena_words = arenaWords; L.initial_bw = cfg.initial_bw; L.ramp_bw = cfg.ramp_bw;
		auto ws = std::make_unique<gak::WaveState<MAXN>>();
		gak::run_job<MAXN, GENERAL, SPARSE>(L, *ws, slot, job);
	}

	// the lanes = reads program (ga_lanes.h): a wave's 64 lanes are run one after the other through each phase; the points
	// where the real wave decides something together (any lane still live, the slice's row range) sit between the phases
	template <int N> void runLanesGroup(const std::vector<uint32_t>& group, uint32_t capCols, uint32_t capRows, uint32_t capMoves)
	{
		using namespace gal;
		GaLanesLaunch L;
		memset(&L, 0, sizeof(L));
		L.graph = g->dev; L.hmm = &g->hmm; L.eq = eq.data(); L.jobs = jobs.data(); L.outs = outs.data();
		L.n_jobs = (uint32_t)jobs.size(); L.lanes_per_wave = 64;
		L.traces = pool.data(); L.trace_top = &poolTop; L.trace_pool_cap = pool.size();
		L.cap_cols = capCols; L.cap_rows = capRows; L.max_slices = cfg.max_slices; L.cap_moves = capMoves;
		L.initial_bw = cfg.initial_bw; L.ramp_bw = cfg.ramp_bw;
		L.emit_runs = cfg.emit_runs;
		const WaveLayout lay = wave_layout<N>(capCols, capRows, cfg.max_slices, capMoves);
		// (on the device the lanes of a wave take their arena blocks from one pool; run one after the other, every lane gets an arena of its own)
		const uint64_t laneArena = lay.bytes - lay.arena;
		std::vector<uint8_t> scratch(lay.arena + 64 * laneArena + 256);
		std::vector<uint32_t> lds((size_t)(Lay<N>::WORDS + kStageWordsLane) * 64);     // tables + the words of the staging image behind them
		std::vector<LaneMem> mem(64);
		std::vector<LaneState> st(64);
		for (int lane = 0; lane < 64; lane++)
		{
			LaneMem& m = mem[lane];
			m.lane = lane; m.tid = lane; m.ls = 64;
			m.lds.base = lds.data() + lane; m.lds.lw = 64;
			m.endPrev = (uint32_t*)(scratch.data() + lay.endA) + lane;
			m.endCur = (uint32_t*)(scratch.data() + lay.endB) + lane;
			m.hdr = (uint32_t*)(scratch.data() + lay.hdr) + lane;
			m.snodes = (uint32_t*)(scratch.data() + lay.snodes) + lane;
			m.moves = (uint32_t*)(scratch.data() + lay.moves) + lane;
			m.arena = scratch.data() + lay.arena + (uint64_t)lane * laneArena;
			m.stage = nullptr;
			const bool has = lane < (int)group.size();
			lane_begin<N>(L, m, st[lane], has ? group[lane] : 0, has);
		}
		// (on the device one arena row belongs to one step of the wave; a lane run on its own simply counts its own steps)
		std::vector<uint32_t> rowTop(64, 0);
		for (uint32_t slice = 0; ; slice++)
		{
			bool any = false;
			for (int lane = 0; lane < 64; lane++)
			{
				lane_band<N>(L, mem[lane], st[lane], slice);
				any = any || st[lane].live;
			}
			if (!any) break;
			for (int lane = 0; lane < 64; lane++) fill_slice<N, 8>(L.graph, mem[lane], st[lane], slice, st[lane].live, rowTop[lane], L.cap_rows, L.cap_cols);
			for (int lane = 0; lane < 64; lane++) lane_end_slice<N>(L, mem[lane], st[lane], slice);
		}
		for (int lane = 0; lane < 64; lane++) lane_finish<N>(L, mem[lane], st[lane], lane < (int)group.size());
	}

	int run() override
	{
		outs.assign(jobs.size(), GaJobOut{});
		uint64_t totalRows = 0;
		for (auto& j : jobs) totalRows += j.n_rows;
		pool.assign(totalRows * 3 + 4096 * jobs.size() + 64, 0);
		if (const char* t = getenv("GA_TEST_TRACE_POOL_BYTES")) pool.assign((size_t)atoll(t) & ~(size_t)3, 0);
		poolTop = 0;
		retried = 0;
		// first pass: the lanes = reads program, groups of 64 jobs (longest first, as the device queue hands them out), with
		// deliberately small capacities; what it declines or cannot hold climbs the wave-per-read ladder below
		const bool lanesFirst = !(getenv("GA_EMUL_NO_LANES") && atoi(getenv("GA_EMUL_NO_LANES")));
		if (lanesFirst)
		{
			// jobs a lanes variant cannot hold (band wider than its LDS tables) move on to the next one, regrouped
			std::vector<uint32_t> order(jobs.size());
			for (uint32_t i = 0; i < jobs.size(); i++) order[i] = i;
			std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return jobs[a].n_rows > jobs[b].n_rows; });
			for (int pass = 0; pass < 3 && !order.empty(); pass++)
			{
				for (size_t at = 0; at < order.size(); at += 64)
				{
					std::vector<uint32_t> group(order.begin() + at, order.begin() + std::min(order.size(), at + 64));
					const uint32_t maxRows = jobs[group[0]].n_rows;
					const uint32_t capRows = (maxRows / 64) * (pass == 0 ? 600 : 2500) + 64, capMoves = maxRows * (pass == 0 ? 2 : 3) + 512;
					if (pass == 0) runLanesGroup<10>(group, 2048, capRows, capMoves);
					else if (pass == 1) runLanesGroup<24>(group, 4096, capRows, capMoves);
					else runLanesGroup<56>(group, 8192, capRows, capMoves);
				}
				std::vector<uint32_t> again;
				for (uint32_t j : order) if (outs[j].status == GA_CAP_NODES || outs[j].status == GA_CAP_HEAP || outs[j].status == GA_CAP_COLS || outs[j].status == GA_CAP_ARENA || outs[j].status == GA_CAP_TRACE) again.push_back(j);
				order.swap(again);
			}
		}
		lanesDone = 0;
		if (rows.empty()) rows = rowsProvider();         // the wave-per-read kernels below read the row codes
		if (getenv("GA_EMUL_DEBUG") && lanesFirst) { int hist[100] = {0}; for (auto& o : outs) hist[o.status < 100 ? o.status : 99]++; fprintf(stderr, "emul: lanes statuses:"); for (int i = 0; i < 100; i++) if (hist[i]) fprintf(stderr, " %d:%d", i, hist[i]); fprintf(stderr, "\n"); }
		for (uint32_t j = 0; j < jobs.size(); j++)
		{
			uint32_t slices = jobs[j].n_rows / 64;
			auto finalStatus = [](int s) { return s == GA_OK || s == GA_ASSERTION || s == GA_BAD_SEED; };
			if (lanesFirst && finalStatus(outs[j].status)) { lanesDone++; continue; }
			if (lanesFirst && outs[j].status == GA_UNSUPPORTED_BAND)
			{
				retried++;
				runOne<256, true, true>(j, 2000000, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 256 + 6 * 300000), jobs[j].n_rows * 8 + 4096);
				continue;
			}
			// deliberately small first-try capacities so the retry ladder is exercised too
			runOne<32, false>(j, 2048, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 40 + 5 * 700), jobs[j].n_rows * 2 + 512);
			auto capacity = [](int s) { return s == GA_CAP_NODES || s == GA_CAP_COLS || s == GA_CAP_ARENA || s == GA_CAP_TRACE || s == GA_CAP_HEAP; };
			auto general = [](int s) { return s == GA_UNSUPPORTED_CYCLE || s == GA_UNSUPPORTED_RAMP; };
			if (capacity(outs[j].status) || general(outs[j].status)) retried++;
			if (general(outs[j].status))
				runOne<64, true>(j, 4096, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 64 + 5 * 1500) * 2, jobs[j].n_rows * 3 + 1024);
			if (capacity(outs[j].status) || general(outs[j].status))
				runOne<256, true>(j, 200000, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 256 + 5 * 20000) * 3, jobs[j].n_rows * 8 + 4096);
			// a band of 200 000 cells or more: the variant that carries the sparse method and the backtrace override (ga_sparse.h)
			if (outs[j].status == GA_UNSUPPORTED_BAND)
			{
				retried++;
				runOne<256, true, true>(j, 2000000, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 256 + 6 * 300000), jobs[j].n_rows * 8 + 4096);
			}
		}
		if (getenv("GA_EMUL_DEBUG")) { int hist[100] = {0}; for (auto& o : outs) hist[o.status < 100 ? o.status : 99]++; fprintf(stderr, "emul: %zu jobs, %llu finished by the lanes program, %llu retried; final statuses:", jobs.size(), (unsigned long long)lanesDone, (unsigned long long)retried); for (int i = 0; i < 100; i++) if (hist[i]) fprintf(stderr, " %d:%d", i, hist[i]); fprintf(stderr, "\n"); }
		return 0;
	}
	int fetch(std::vector<GaJobOut>& o, const uint8_t** traces, uint64_t* nBytes) override
	{
		o = outs;
		*traces = pool.data();
		*nBytes = poolTop;
		return 0;
	}
	GaRunStats stats() const override { GaRunStats s; s.jobs_retried = retried; s.slots = 1; return s; }
	bool emittingRuns() const override { return cfg.emit_runs != 0; }
};

}  // namespace

GaBackendGraph* ga_backend_upload_graph(const GaFlatGraph& flat, const GaHmmTables& hmm, int, int* status)
{
	EmulGraph* g = new EmulGraph();
	g->flat = flat;
	g->hmm = hmm;
	g->dev.n_nodes = (uint32_t)(flat.node_start.size() - 1);
	g->dev.reserved = 0;
	g->dev.node_start = g->flat.node_start.data();
	g->dev.seq2 = g->flat.seq2.data();
	g->dev.in_off = g->flat.in_off.data();
	g->dev.in_nbr = g->flat.in_nbr.data();
	g->dev.out_off = g->flat.out_off.data();
	g->dev.out_nbr = g->flat.out_nbr.data();
	g->nodeRec = ga_build_node_records(g->flat);
	g->dev.node_rec = g->nodeRec.data();
	*status = 0;
	return g;
}

GaBackendBatch* ga_backend_create_batch(GaBackendGraph* g, GaRowsProvider rows, const uint64_t* eq, const GaEqSource*, size_t eqWords, const std::vector<GaJob>& jobs, const GaRunConfig& cfg, int* status)
{
	EmulBatch* b = new EmulBatch();
	b->g = static_cast<EmulGraph*>(g);
	b->rowsProvider = rows;
	b->eq.assign(eq, eq + eqWords);
	b->jobs = jobs;
	b->cfg = cfg;
	*status = 0;
	return b;
}

// component hooks for unit tests of the order-emulation pieces
extern "C" int ga_emul_hash_order(const uint32_t* keys, int n, int32_t* out)
{
	auto ws = std::make_unique<gak::WaveState<256>>();
	if (n > 256) return -1;
	gak::hash_order(*ws, keys, n);
	for (int i = 0; i < n; i++) out[i] = ws->h_order[i];
	return n;
}

extern "C" int ga_emul_hash_order_lanes(const uint32_t* keys, int n, int32_t* out)
{
	auto ws = std::make_unique<gak::WaveState<64>>();
	if (n > 64) return -1;
	gak::hash_order_lanes(*ws, keys, n);
	for (int i = 0; i < n; i++) out[i] = ws->h_order[i];
	return n;
}

// the same through the lane-resident heap
extern "C" int ga_emul_heap_lanes(const uint32_t* nodes, const int32_t* prios, int nOps, uint32_t* popped)
{
	gak::LaneHeap<4> h;
	for (int i = 0; i < 4; i++) { h.node[i] = gaw::VI(0); h.prio[i] = gaw::VI(0); }
	int size = 0, k = 0;
	for (int i = 0; i < nOps; i++)
	{
		if (prios[i] >= 0) { if (!h.push(size, nodes[i], prios[i])) return -1; }
		else if (size > 0) { popped[k++] = (uint32_t)h.getNode(0); h.pop(size); }
	}
	while (size > 0) { popped[k++] = (uint32_t)h.getNode(0); h.pop(size); }
	return k;
}

// push (node, prio) pairs then pop everything; ops: prio >= 0 push, prio < 0 pop.  returns pop order
extern "C" int ga_emul_heap(const uint32_t* nodes, const int32_t* prios, int nOps, uint32_t* popped)
{
	auto ws = std::make_unique<gak::WaveState<256>>();
	int size = 0, k = 0;
	for (int i = 0; i < nOps; i++)
	{
		if (prios[i] >= 0) { if (!gak::heap_push(*ws, size, nodes[i], prios[i])) return -1; }
		else if (size > 0) { popped[k++] = ws->heap_node[0]; gak::heap_pop(*ws, size); }
	}
	while (size > 0) { popped[k++] = ws->heap_node[0]; gak::heap_pop(*ws, size); }
	return k;
}

// ---- component hooks for the lanes = reads program (graphaligner_amd/csrc/ga_lanes.h) ---------------------------------------
// column = {vp, vn, before}
extern "C" void ga_emul_lanes_merge(const uint64_t* a, const uint64_t* b, uint64_t* out)
{
	gal::Col x{a[0], a[1], (int)(int64_t)a[2]}, y{b[0], b[1], (int)(int64_t)b[2]};
	gal::column_merge(x, y);
	out[0] = x.vp; out[1] = x.vn; out[2] = (uint64_t)(int64_t)x.before;
}
// the vertical re-entry: cell-wise minimum with the run coming down from a cell d below the column's row j-1 score
extern "C" void ga_emul_lanes_reenter(const uint64_t* a, int d, uint64_t* out)
{
	gal::Col x{a[0], a[1], (int)(int64_t)a[2]};
	gal::column_reenter(x, d);
	out[0] = x.vp; out[1] = x.vn; out[2] = (uint64_t)(int64_t)x.before;
}
// one column step (Eq word, "no diagonal into row j", new row j-1 score)
extern "C" void ga_emul_lanes_step(const uint64_t* a, uint64_t eq, int noDiag, int calc, uint64_t* out)
{
	gal::Col x{a[0], a[1], (int)(int64_t)a[2]};
	gal::column_step(x, eq, noDiag != 0, calc);
	out[0] = x.vp; out[1] = x.vn; out[2] = (uint64_t)(int64_t)x.before;
}
// map iteration order and heap pop order of the per-lane tables (one lane, N = 56)
extern "C" int ga_emul_lanes_hash_order(const uint32_t* keys, int n, int32_t* out)
{
	typedef gal::Lay<56> LY;
	if (n > 56) return -1;
	std::vector<uint32_t> lds((size_t)LY::WORDS);
	gal::Lds l{lds.data(), 1};
	for (int i = 0; i < n; i++) l.wr(LY::P_NODE + i, keys[i]);
	gal::hash_order<56>(l, n);
	for (int i = 0; i < n; i++) out[i] = (int32_t)l.rdb(LY::X_HASH, LY::HB_ORDER + i);
	return n;
}
extern "C" int ga_emul_lanes_heap(const uint32_t* nodes, const int32_t* prios, int nOps, uint32_t* popped)
{
	typedef gal::Lay<56> LY;
	std::vector<uint32_t> lds((size_t)LY::WORDS);
	gal::Lds l{lds.data(), 1};
	int size = 0, k = 0;
	for (int i = 0; i < nOps; i++)
	{
		if (prios[i] >= 0) { if (!gal::heap_push<56>(l, size, nodes[i], prios[i])) return -1; }
		else if (size > 0) { popped[k++] = l.rd(LY::X_HEAPN); gal::heap_pop<56>(l, size); }
	}
	while (size > 0) { popped[k++] = l.rd(LY::X_HEAPN); gal::heap_pop<56>(l, size); }
	return k;
}
