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
	std::vector<uint8_t> rows;
	std::vector<GaJob> jobs;
	GaRunConfig cfg;
	std::vector<GaJobOut> outs;
	std::vector<uint8_t> pool;
	uint64_t poolTop = 0;
	uint64_t retried = 0;

	template <int MAXN, bool GENERAL> void runOne(uint32_t job, uint32_t capCols, uint64_t arenaWords, uint32_t traceCap)
	{
		std::vector<uint32_t> endA(capCols), endB(capCols), arena(arenaWords), sliceOff(cfg.max_slices + 1);
		std::vector<uint8_t> flags(cfg.max_slices + 1);
		std::vector<uint8_t> staging(traceCap + 64);
		std::vector<uint32_t> ckpt(cfg.max_slices + 2), below(cfg.max_slices + 1);
		gak::Slot slot{endA.data(), endB.data(), arena.data(), sliceOff.data(), flags.data(), staging.data(), ckpt.data(), below.data()};
		GaLaunch L;
		memset(&L, 0, sizeof(L));
		L.graph = g->dev; L.hmm = &g->hmm; L.rows = rows.data(); L.jobs = jobs.data(); L.outs = outs.data();
		L.traces = pool.data(); L.trace_top = &poolTop; L.trace_pool_cap = pool.size();
		L.n_jobs = (uint32_t)jobs.size(); L.trace_cap = traceCap; L.cap_cols = capCols; L.max_slices = cfg.max_slices;
		L.arena_words = arenaWords; L.initial_bw = cfg.initial_bw; L.ramp_bw = cfg.ramp_bw;
		auto ws = std::make_unique<gak::WaveState<MAXN>>();
		gak::run_job<MAXN, GENERAL>(L, *ws, slot, job);
	}

	int run() override
	{
		outs.assign(jobs.size(), GaJobOut{});
		uint64_t totalRows = 0;
		for (auto& j : jobs) totalRows += j.n_rows;
		pool.assign(totalRows * 3 + 4096 * jobs.size() + 64, 0);
		poolTop = 0;
		retried = 0;
		for (uint32_t j = 0; j < jobs.size(); j++)
		{
			uint32_t slices = jobs[j].n_rows / 64;
			// deliberately small first-try capacities so the retry ladder is exercised too
			runOne<32, false>(j, 2048, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 40 + 5 * 700), jobs[j].n_rows * 2 + 512);
			auto capacity = [](int s) { return s == GA_CAP_NODES || s == GA_CAP_COLS || s == GA_CAP_ARENA || s == GA_CAP_TRACE || s == GA_CAP_HEAP; };
			auto general = [](int s) { return s == GA_UNSUPPORTED_CYCLE || s == GA_UNSUPPORTED_RAMP; };
			if (capacity(outs[j].status) || general(outs[j].status)) retried++;
			if (general(outs[j].status))
				runOne<64, true>(j, 4096, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 64 + 5 * 1500) * 2, jobs[j].n_rows * 3 + 1024);
			if (capacity(outs[j].status) || general(outs[j].status))
				runOne<256, true>(j, 200000, 64 + (uint64_t)slices * (gak::kSliceHdrWords + 3 * 256 + 5 * 20000) * 3, jobs[j].n_rows * 8 + 4096);
		}
		return 0;
	}
	int fetch(std::vector<GaJobOut>& o, std::vector<uint8_t>& traces) override
	{
		o = outs;
		traces.assign(pool.begin(), pool.begin() + poolTop);
		return 0;
	}
	GaRunStats stats() const override { GaRunStats s; s.jobs_retried = retried; s.slots = 1; return s; }
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

GaBackendBatch* ga_backend_create_batch(GaBackendGraph* g, const std::vector<uint8_t>& rows, const std::vector<GaJob>& jobs, const GaRunConfig& cfg, int* status)
{
	EmulBatch* b = new EmulBatch();
	b->g = static_cast<EmulGraph*>(g);
	b->rows = rows;
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
