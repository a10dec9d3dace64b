// ga_device.hip -- gfx950 back end: keeps the flattened graph resident in HBM and runs the
// extension program (ga_kernel.h) as a persistent launch: one 64-lane workgroup (= one
// wavefront) per slot, each slot pulling read directions from a device-side queue and owning a
// private region of HBM for its slices' VP/VN words.  Reads are independent, so there is no
// inter-workgroup communication besides the queue counter and the trace-pool bump counter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/graphaligner_amd.h"
#include "ga_backend.h"
#include "ga_kernel.h"

namespace {

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "graphaligner_amd: %s failed: %s\n", #call, hipGetErrorString(e_)); return GA_E_DEVICE; } } while (0)

struct SlotLayout
{
	uint64_t endPrev, endCur, sliceOff, arena, trace, flags, ckpt, belowOff, bytes;
};
__host__ __device__ inline SlotLayout slotLayout(uint32_t capCols, uint32_t maxSlices, uint64_t arenaWords, uint32_t traceCap)
{
	auto up = [](uint64_t x) { return (x + 255) & ~255ull; };
	SlotLayout l;
	uint64_t at = 0;
	l.endPrev = at; at = up(at + 4ull * capCols);
	l.endCur = at; at = up(at + 4ull * capCols);
	l.sliceOff = at; at = up(at + 4ull * (maxSlices + 1));
	l.arena = at; at = up(at + 4ull * arenaWords);
	l.trace = at; at = up(at + 1ull * traceCap + 64);
	l.flags = at; at = up(at + maxSlices + 1);
	l.ckpt = at; at = up(at + 4ull * (maxSlices + 2));
	l.belowOff = at; at = up(at + 4ull * (maxSlices + 1));
	l.bytes = at;
	return l;
}

#ifndef GA_WAVES_EU
#define GA_WAVES_EU 4
#endif
template <int MAXN, bool GENERAL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GA_WAVES_EU, 8))) ga_extend_kernel(GaLaunch L)
{
	__shared__ gak::WaveState<MAXN> ws;
	const SlotLayout lay = slotLayout(L.cap_cols, L.max_slices, L.arena_words, L.trace_cap);
	uint8_t* base = L.scratch + (uint64_t)blockIdx.x * L.slot_bytes;
	gak::Slot slot;
	slot.end_prev = (uint32_t*)(base + lay.endPrev);
	slot.end_cur = (uint32_t*)(base + lay.endCur);
	slot.slice_off = (uint32_t*)(base + lay.sliceOff);
	slot.arena = (uint32_t*)(base + lay.arena);
	slot.trace = base + lay.trace;
	slot.slice_flags = base + lay.flags;
	slot.ckpt = (uint32_t*)(base + lay.ckpt);
	slot.below_off = (uint32_t*)(base + lay.belowOff);
	while (true)
	{
		uint32_t k = gaw::wave_atomic_add(L.next_job, 1u);
		if (k >= L.n_jobs) break;                      // every wave reaches this exit once the queue is drained
		uint32_t job = L.job_list ? L.job_list[k] : k;
		gak::run_job<MAXN, GENERAL>(L, ws, slot, job);
		__syncthreads();
	}
}

struct DevGraph : GaBackendGraph
{
	int device = 0;
	GaDevGraph g;
	GaHmmTables* hmm = nullptr;
	std::vector<void*> allocs;
	int cus = 0;
	~DevGraph() override { hipSetDevice(device); for (void* p : allocs) hipFree(p); }
	template <typename T> int put(const std::vector<T>& v, const T** out)
	{
		void* p = nullptr;
		HIP_OK(hipMalloc(&p, std::max<size_t>(v.size() * sizeof(T), 16)));
		allocs.push_back(p);
		HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
		*out = (const T*)p;
		return 0;
	}
};

struct DevBatch : GaBackendBatch
{
	DevGraph* g = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t evStart = nullptr, evStop = nullptr;
	std::vector<void*> allocs;
	std::vector<GaJob> jobs;
	GaRunConfig cfg;
	GaLaunch L;                 // main launch
	uint32_t slots = 0, wavesPerCu = 0;
	bool narrow = true;         // first pass with the 32-node variant (more waves per CU); misses go to the 256-node variant
	std::vector<GaJobOut> outs;
	std::vector<uint32_t> orderHost;   // job order of the main launch (longest first) when lengths differ
	GaRunStats st;
	// retry pass (wide variant), built lazily
	uint8_t* retryScratch = nullptr;
	uint32_t* retryList = nullptr;
	size_t retryScratchBytes = 0;

	~DevBatch() override
	{
		hipSetDevice(g->device);
		for (void* p : allocs) hipFree(p);
		if (retryScratch) hipFree(retryScratch);
		if (retryList) hipFree(retryList);
		if (evStart) hipEventDestroy(evStart);
		if (evStop) hipEventDestroy(evStop);
		if (stream) hipStreamDestroy(stream);
	}
	template <typename T> int alloc(T** out, size_t count)
	{
		void* p = nullptr;
		HIP_OK(hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)));
		allocs.push_back(p);
		*out = (T*)p;
		return 0;
	}

	int init(const std::vector<uint8_t>& rows, const std::vector<GaJob>& jobsIn)
	{
		HIP_OK(hipSetDevice(g->device));
		HIP_OK(hipStreamCreate(&stream));
		HIP_OK(hipEventCreate(&evStart));
		HIP_OK(hipEventCreate(&evStop));
		jobs = jobsIn;
		memset(&L, 0, sizeof(L));
		L.graph = g->g;
		L.hmm = g->hmm;
		L.n_jobs = (uint32_t)jobs.size();
		L.initial_bw = cfg.initial_bw;
		L.ramp_bw = cfg.ramp_bw;
		L.max_slices = std::max<uint32_t>(cfg.max_slices, 1);
		uint8_t* dRows; GaJob* dJobs;
		if (alloc(&dRows, rows.size())) return GA_E_DEVICE;
		if (alloc(&dJobs, jobs.size())) return GA_E_DEVICE;
		HIP_OK(hipMemcpyAsync(dRows, rows.data(), rows.size(), hipMemcpyHostToDevice, stream));
		HIP_OK(hipMemcpyAsync(dJobs, jobs.data(), jobs.size() * sizeof(GaJob), hipMemcpyHostToDevice, stream));
		L.rows = dRows;
		L.jobs = dJobs;
		// the device queue hands jobs out longest first: the short ones fill the tail of the launch
		bool uniform = true;
		for (auto& j : jobs) if (j.n_rows != jobs[0].n_rows) { uniform = false; break; }
		if (!uniform)
		{
			orderHost.resize(jobs.size());
			for (uint32_t i = 0; i < jobs.size(); i++) orderHost[i] = i;
			std::stable_sort(orderHost.begin(), orderHost.end(), [&](uint32_t a, uint32_t b) { return jobs[a].n_rows > jobs[b].n_rows; });
			uint32_t* dOrder;
			if (alloc(&dOrder, orderHost.size())) return GA_E_DEVICE;
			HIP_OK(hipMemcpyAsync(dOrder, orderHost.data(), orderHost.size() * 4, hipMemcpyHostToDevice, stream));
			L.job_list = dOrder;
		}
		if (alloc(&L.outs, jobs.size())) return GA_E_DEVICE;
		if (alloc(&L.next_job, 4)) return GA_E_DEVICE;
		if (alloc(&L.trace_top, 2)) return GA_E_DEVICE;
		uint64_t totalRows = 0;
		for (auto& j : jobs) totalRows += j.n_rows;
		L.trace_pool_cap = ((totalRows + totalRows / 2 + 256ull * jobs.size() + 4096) + 3) & ~3ull;
		if (alloc(&L.traces, L.trace_pool_cap)) return GA_E_DEVICE;
		// slot geometry: bands of ~300-700 columns are the rule (b = 35 on variation graphs);
		// anything wider fails with a capacity status and is rerun by the wide variant
		L.cap_cols = 4096;
		L.trace_cap = cfg.max_rows * 2 + 1024;
		L.arena_words = 64 + (uint64_t)L.max_slices * (gak::kSliceHdrWords + 3 * 64 + 5 * 800);
		SlotLayout lay = slotLayout(L.cap_cols, L.max_slices, L.arena_words, L.trace_cap);
		L.slot_bytes = lay.bytes;
		size_t freeB = 0, totalB = 0;
		HIP_OK(hipMemGetInfo(&freeB, &totalB));
		wavesPerCu = getenv("GA_WAVES_PER_CU") ? (uint32_t)atoi(getenv("GA_WAVES_PER_CU")) : 24;
		narrow = !(getenv("GA_NARROW") && atoi(getenv("GA_NARROW")) == 0);
		uint64_t want = (uint64_t)g->cus * wavesPerCu;
		uint64_t fit = (uint64_t)(freeB * 0.8) / std::max<uint64_t>(lay.bytes, 1);
		slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(want, fit), std::max<size_t>(jobs.size(), 1)));
		// (jobs are pulled from a device queue, so the slots need no balancing by hand; a few more workgroups than fit at once
		// start as the first ones drain the queue and even out the tail)
		if (alloc(&L.scratch, (size_t)slots * lay.bytes)) return GA_E_DEVICE;
		st.slots = slots;
		st.waves_per_cu = wavesPerCu;
		st.scratch_bytes = (uint64_t)slots * lay.bytes;
		HIP_OK(hipStreamSynchronize(stream));
		return 0;
	}

	int run() override
	{
		HIP_OK(hipSetDevice(g->device));
		if (jobs.empty()) { outs.clear(); return 0; }
		HIP_OK(hipMemsetAsync(L.next_job, 0, 16, stream));
		HIP_OK(hipMemsetAsync(L.trace_top, 0, 16, stream));
		HIP_OK(hipEventRecord(evStart, stream));
		if (narrow) hipLaunchKernelGGL((ga_extend_kernel<32, false>), dim3(slots), dim3(64), 0, stream, L);
		else hipLaunchKernelGGL((ga_extend_kernel<64, false>), dim3(slots), dim3(64), 0, stream, L);
		HIP_OK(hipGetLastError());
		HIP_OK(hipEventRecord(evStop, stream));
		outs.resize(jobs.size());
		HIP_OK(hipMemcpyAsync(outs.data(), L.outs, outs.size() * sizeof(GaJobOut), hipMemcpyDeviceToHost, stream));
		HIP_OK(hipStreamSynchronize(stream));
		float ms = 0;
		HIP_OK(hipEventElapsedTime(&ms, evStart, evStop));
		st.kernel_ms = ms;
		if (getenv("GA_DEBUG_PASSES")) fprintf(stderr, "graphaligner_amd: main pass: %zu jobs on %u slots, %.2f ms\n", jobs.size(), slots, ms);
		for (int k = 0; k < 8; k++) { st.stamps[k] = 0; for (auto& o : outs) st.stamps[k] += o.stamps[k]; }
		// ---- what the lean variant could not finish climbs a ladder: 64 band nodes in LDS; then the general variants, which
		// also carry the paths for bands with cycles and for ramp redos; last 256 band nodes with large buffers ----
		st.jobs_retried = 0;
		int rc = retryPass<64, false>(8192, 3 * 64 + 5 * 2048, 3, 12, true, false);
		if (rc) return rc;
		rc = retryPass<64, true>(8192, 3 * 64 + 5 * 4096, 4, 8, false, true);      // only what needs the extra paths: capacity misses go straight on
		if (rc) return rc;
		rc = retryPass<256, true>(65536, 3 * 256 + 5 * 8192, 6, 4, true, true);
		return rc;
	}

	static bool isCapacity(int s) { return s == GA_CAP_NODES || s == GA_CAP_COLS || s == GA_CAP_ARENA || s == GA_CAP_TRACE || s == GA_CAP_HEAP; }
	// bands with cycles and ramp redos: only the general variants carry those paths
	static bool needsGeneral(int s) { return s == GA_UNSUPPORTED_CYCLE || s == GA_UNSUPPORTED_RAMP; }

	template <int MAXN, bool GENERAL> int retryPass(uint32_t capCols, uint64_t arenaWordsPerSlice, uint32_t traceMul, uint32_t wavesPerCuRetry, bool takeCapacity, bool takeGeneral)
	{
		std::vector<uint32_t> again;
		for (uint32_t i = 0; i < outs.size(); i++) if ((takeCapacity && isCapacity(outs[i].status)) || (takeGeneral && needsGeneral(outs[i].status))) again.push_back(i);
		if (again.empty()) return 0;
		st.jobs_retried += again.size();
		GaLaunch R = L;
		uint32_t maxRows = 0;
		for (uint32_t i : again) maxRows = std::max(maxRows, jobs[i].n_rows);
		R.cap_cols = capCols;
		R.trace_cap = maxRows * traceMul + 4096;
		R.arena_words = 64 + (uint64_t)(maxRows / 64) * (gak::kSliceHdrWords + arenaWordsPerSlice);
		SlotLayout lay = slotLayout(R.cap_cols, R.max_slices, R.arena_words, R.trace_cap);
		R.slot_bytes = lay.bytes;
		size_t freeB = 0, totalB = 0;
		HIP_OK(hipMemGetInfo(&freeB, &totalB));
		uint64_t fit = (uint64_t)((freeB + retryScratchBytes) * 0.8) / lay.bytes;
		uint32_t rslots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>((uint64_t)g->cus * wavesPerCuRetry, fit), again.size()));
		if ((size_t)rslots * lay.bytes > retryScratchBytes)
		{
			if (retryScratch) hipFree(retryScratch);
			retryScratch = nullptr;
			retryScratchBytes = 0;
			HIP_OK(hipMalloc((void**)&retryScratch, (size_t)rslots * lay.bytes));
			retryScratchBytes = (size_t)rslots * lay.bytes;
		}
		if (retryList) hipFree(retryList);
		retryList = nullptr;
		HIP_OK(hipMalloc((void**)&retryList, again.size() * 4));
		HIP_OK(hipMemcpyAsync(retryList, again.data(), again.size() * 4, hipMemcpyHostToDevice, stream));
		R.scratch = retryScratch;
		R.job_list = retryList;
		R.n_jobs = (uint32_t)again.size();
		HIP_OK(hipMemsetAsync(R.next_job, 0, 16, stream));
		hipEvent_t a, b;
		HIP_OK(hipEventCreate(&a));
		HIP_OK(hipEventCreate(&b));
		HIP_OK(hipEventRecord(a, stream));
		hipLaunchKernelGGL((ga_extend_kernel<MAXN, GENERAL>), dim3(rslots), dim3(64), 0, stream, R);
		HIP_OK(hipGetLastError());
		HIP_OK(hipEventRecord(b, stream));
		HIP_OK(hipMemcpyAsync(outs.data(), L.outs, outs.size() * sizeof(GaJobOut), hipMemcpyDeviceToHost, stream));
		HIP_OK(hipStreamSynchronize(stream));
		float ms2 = 0;
		HIP_OK(hipEventElapsedTime(&ms2, a, b));
		st.kernel_ms += ms2;
		if (getenv("GA_DEBUG_PASSES")) fprintf(stderr, "graphaligner_amd: retry pass <%d,%d>: %zu jobs on %u slots, %.2f ms\n", MAXN, (int)GENERAL, again.size(), rslots, ms2);
		hipEventDestroy(a);
		hipEventDestroy(b);
		return 0;
	}

	int fetch(std::vector<GaJobOut>& o, std::vector<uint8_t>& traces) override
	{
		HIP_OK(hipSetDevice(g->device));
		o = outs;
		uint64_t top = 0;
		HIP_OK(hipMemcpy(&top, L.trace_top, 8, hipMemcpyDeviceToHost));
		top = std::min<uint64_t>(top, L.trace_pool_cap);
		traces.resize(top);
		if (top) HIP_OK(hipMemcpy(traces.data(), L.traces, top, hipMemcpyDeviceToHost));
		return 0;
	}
	GaRunStats stats() const override { return st; }
};

}  // namespace

GaBackendGraph* ga_backend_upload_graph(const GaFlatGraph& flat, const GaHmmTables& hmm, int device, int* status)
{
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) { *status = GA_E_NO_DEVICE; return nullptr; }
	if (hipSetDevice(device) != hipSuccess) { *status = GA_E_NO_DEVICE; return nullptr; }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { *status = GA_E_NO_DEVICE; return nullptr; }
	// node records are addressed with 32-bit lane arithmetic (node << 4 words): up to 2^27 directed nodes
	if (flat.node_start.size() - 1 >= (1ull << 27)) { *status = GA_E_INVALID; return nullptr; }
	DevGraph* g = new DevGraph();
	g->device = device;
	g->cus = prop.multiProcessorCount;
	g->g.n_nodes = (uint32_t)(flat.node_start.size() - 1);
	g->g.reserved = getenv("GA_DIAG_NO_TRACEBACK") ? 1u : 0u;     // diagnostic: counters of the fill phases alone (results are then incomplete)
	int bad = 0;
	bad |= g->put(flat.node_start, &g->g.node_start);
	bad |= g->put(flat.seq2, &g->g.seq2);
	bad |= g->put(flat.in_off, &g->g.in_off);
	bad |= g->put(flat.in_nbr, &g->g.in_nbr);
	bad |= g->put(flat.out_off, &g->g.out_off);
	bad |= g->put(flat.out_nbr, &g->g.out_nbr);
	bad |= g->put(ga_build_node_records(flat), &g->g.node_rec);
	std::vector<GaHmmTables> h(1, hmm);
	const GaHmmTables* dh = nullptr;
	bad |= g->put(h, &dh);
	g->hmm = const_cast<GaHmmTables*>(dh);
	if (bad) { delete g; *status = GA_E_DEVICE; return nullptr; }
	*status = 0;
	return g;
}

GaBackendBatch* ga_backend_create_batch(GaBackendGraph* graph, const std::vector<uint8_t>& rows, const std::vector<GaJob>& jobs,
                                        const GaRunConfig& cfg, int* status)
{
	DevBatch* b = new DevBatch();
	b->g = static_cast<DevGraph*>(graph);
	b->cfg = cfg;
	int s = b->init(rows, jobs);
	if (s) { delete b; *status = s; return nullptr; }
	*status = 0;
	return b;
}
