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
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/graphaligner_amd.h"
#include "ga_backend.h"
#include "ga_kernel.h"
#include "ga_lanes.h"

namespace {

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "graphaligner_amd: %s failed: %s\n", #call, hipGetErrorString(e_)); return GA_E_DEVICE; } } while (0)

struct SlotLayout
{
	uint64_t endPrev, endCur, sliceOff, arena, trace, flags, ckpt, belowOff, sparse, ovr, bytes;
};
__host__ __device__ inline SlotLayout slotLayout(uint32_t capCols, uint32_t maxSlices, uint64_t arenaWords, uint32_t traceCap, uint32_t sparseBw = 0)
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
	// the variant that carries the sparse method: its tables (first, so that the host can clear them) and the override windows
	l.sparse = at; if (sparseBw) at = up(at + gak::sparse_mem_bytes(sparseBw));
	l.ovr = at; if (sparseBw) at = up(at + 8ull * (maxSlices + 2));
	l.bytes = at;
	return l;
}

#ifndef GA_WAVES_EU
#define GA_WAVES_EU 4
#endif
template <int MAXN, bool GENERAL, bool SPARSE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GA_WAVES_EU, 8))) ga_extend_kernel(GaLaunch L)
{
	__shared__ gak::WaveState<MAXN> ws;
	const SlotLayout lay = slotLayout(L.cap_cols, L.max_slices, L.arena_words, L.trace_cap, SPARSE ? L.sparse_bw : 0u);
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
	slot.sparse = SPARSE ? base + lay.sparse : nullptr;
	slot.ovr = SPARSE ? (uint32_t*)(base + lay.ovr) : nullptr;
	slot.sparse_max_bw = SPARSE ? L.sparse_bw : 0u;
	while (true)
	{
		uint32_t k = gaw::wave_atomic_add(L.next_job, 1u);
		if (k >= L.n_jobs) break;                      // every wave reaches this exit once the queue is drained
		uint32_t job = L.job_list ? L.job_list[k] : k;
		gak::run_job<MAXN, GENERAL, SPARSE>(L, ws, slot, job);
		__syncthreads();
	}
}

// ---- lanes = reads (ga_lanes.h): every lane owns one job, the wave's lanes advance slice by slice together -----------
// N band nodes per lane in LDS (10 N words per lane), slice records in blocks of R columns per lane, LW = lane stride of
// the LDS tables = how many lanes of a wave carry jobs.
template <int N, int LW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) ga_lanes_kernel(gal::GaLanesLaunch L)
{
	using namespace gal;
	__shared__ uint32_t lds[Lay<N>::WORDS * LW + LW * kStageWords64 * 2 + 128];  // tables, the staging image of LW lanes, the lanes' arena blocks
	const int lane = (int)threadIdx.x;
	const WaveLayout lay = wave_layout<N>(L.cap_cols, L.cap_rows, L.max_slices, L.cap_moves, LW);
	uint8_t* base = L.scratch + (uint64_t)blockIdx.x * L.wave_bytes;
	while (true)
	{
		const uint32_t grp = gaw::wave_atomic_add(L.next_group, 1u);
		const uint64_t first = (uint64_t)grp * L.lanes_per_wave;
		if (first >= L.n_jobs) break;                      // every wave reaches this exit once the queue is drained
		const bool hasJob = lane < (int)L.lanes_per_wave && lane < LW && first + (uint32_t)lane < L.n_jobs;
		const uint32_t jobIndex = hasJob ? (L.job_list ? L.job_list[first + (uint32_t)lane] : (uint32_t)(first + (uint32_t)lane)) : 0u;
		LaneMem m;
		m.lane = lane < LW ? lane : 0;      // (lanes beyond the variant's LW carry no job: they shadow lane 0's addresses and store nothing)
		m.tid = lane;
		m.flushPart = (uint32_t)lane % 12u; m.flushLane = (uint32_t)lane / 12u;
		m.ls = LW;
		m.lds.base = lds + (lane < LW ? lane : 0);
		m.lds.lw = LW;
		m.stage = (uint64_t*)(lds + Lay<N>::WORDS * LW);
		m.laneBlocks = lds + Lay<N>::WORDS * LW + LW * kStageWords64 * 2;
		m.usedChunks = 12u * (L.lanes_per_wave < (uint32_t)LW ? L.lanes_per_wave : (uint32_t)LW);
		m.endPrev = (uint32_t*)(base + lay.endA) + m.lane;
		m.endCur = (uint32_t*)(base + lay.endB) + m.lane;
		m.hdr = (uint32_t*)(base + lay.hdr) + m.lane;
		m.snodes = (uint32_t*)(base + lay.snodes) + m.lane;
		m.moves = (uint32_t*)(base + lay.moves) + m.lane;
		m.arena = base + lay.arena;
		LaneState st;
#if GA_STAMPS == 3
		const uint64_t wall0 = wall_clock64(), cyc0 = gaw::stamp();
#endif
		uint64_t tA = gaw::stamp(), tB, acc[4] = {0, 0, 0, 0};
#define GAL_LAP(i) do { tB = gaw::stamp(); acc[i] += tB - tA; tA = tB; } while (0)
		lane_begin<N>(L, m, st, jobIndex, hasJob);
		uint32_t blockTop = 0;                             // wave-uniform: the arena's next free block of 8 rows
		for (uint32_t slice = 0; ; slice++)
		{
			lane_band<N>(L, m, st, slice);
			GAL_LAP(0);
			if (!__ballot(st.live)) break;
			fill_slice<N, 8, LW>(L.graph, m, st, slice, st.live, blockTop, L.cap_rows, L.cap_cols);
			GAL_LAP(1);
			lane_end_slice<N>(L, m, st, slice);
			GAL_LAP(2);
		}
		lane_finish<N>(L, m, st, hasJob);
		GAL_LAP(3);
#undef GAL_LAP
#ifdef GA_STAMPS
		// diagnostic build: the wave's cycles per phase, booked on its first job (names in bench.py)
		if (hasJob && lane == 0) { GaJobOut* o = L.outs + st.job; o->stamps[1] = acc[0]; o->stamps[4] = acc[1]; o->stamps[0] = acc[2]; o->stamps[5] = acc[3]; o->stamps[2] = st.laps[0]; o->stamps[3] = st.laps[1]; o->stamps[6] = st.laps[2]; o->stamps[7] = st.laps[7];
#if GA_STAMPS == 4
			// fourth diagnostic layout: the parts of the band phase in [0..4], the band phase in [6], the fill in [7]
			o->stamps[6] = acc[0]; o->stamps[7] = acc[1];
			for (int i = 0; i < 5; i++) o->stamps[i] = st.blaps[i];
#endif
#if GA_STAMPS == 3
			// third diagnostic layout: when the wave ran, on the constant 100 MHz clock, and how many shader cycles that was
			o->stamps[0] = wall0; o->stamps[1] = wall_clock64(); o->stamps[4] = gaw::stamp() - cyc0;
#endif
#if GA_STAMPS == 2
			// second diagnostic layout: the parts of the traceback's general step instead of end_slice / band / fill
			o->stamps[0] = st.laps[4]; o->stamps[1] = st.laps[5]; o->stamps[4] = st.laps[6];
#endif
		}
#endif
		__syncthreads();
	}
}

// ---- the match words of the lanes = reads kernel, built from the reads' bytes (ga_backend.h: GaEqSource) -------------------------
// One wave per job (its rows come from one read, forwards from the seed or backwards before it); lane r of the wave holds row r of the
// slice in hand: the character's row code through a 512-byte table in LDS, four ballots = the slice's match words against A, C, G, T
// (characterMatch, GraphAligner.h:2039-2110; what the reference builds as EqVector, :2338-2351), one more for "a row outside IUPAC".
// The row codes themselves (one byte per padded row: what the wave-per-read kernels read) are written on the way.
__global__ void __launch_bounds__(64) ga_eq_words_kernel(const uint8_t* __restrict__ seq, const GaEqFill* __restrict__ fills, uint32_t nFills, const uint8_t* __restrict__ lut,
                                                         uint32_t padCode, uint64_t* __restrict__ eq, uint8_t* __restrict__ flags, uint8_t* __restrict__ rows)
{
	__shared__ uint8_t t[512];
	const uint32_t lane = threadIdx.x;
	for (uint32_t i = lane; i < 512; i += 64) t[i] = lut[i];
	__syncthreads();
	for (uint32_t fi = blockIdx.x; fi < nFills; fi += gridDim.x)
	{
		const GaEqFill f = fills[fi];
		const uint8_t* s = seq + f.seq_off;
		bool bad = false;
		for (uint32_t r0 = 0; r0 < f.padded; r0 += 64)
		{
			const uint32_t k = r0 + lane;
			uint32_t code = padCode;
			if (k < f.n) code = f.backward ? t[256 + s[f.n - 1 - k]] : t[s[f.pos + k]];
			rows[f.eq_slice * 64 + k] = (uint8_t)code;
			const uint64_t e0 = __ballot(code & 1u), e1 = __ballot(code & 2u), e2 = __ballot(code & 4u), e3 = __ballot(code & 8u);
			const uint64_t inv = __ballot(code & GA_ROW_INVALID);
			const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)code, 63);
			const uint64_t meta = (uint64_t)((last >> 4) & 7u) | (inv ? 8u : 0u);     // exact-compare code of the slice's last row | a row outside IUPAC
			if (lane < 5) eq[(f.eq_slice + r0 / 64) * 5 + lane] = lane == 0 ? e0 : lane == 1 ? e1 : lane == 2 ? e2 : lane == 3 ? e3 : meta;
			bad = bad || inv != 0;
		}
		if (lane == 0) flags[fi] = bad ? 1 : 0;
	}
}

// every job's output record starts a run as "not run" (what a job keeps when no pass ever picks it up)
__global__ void ga_outs_init_kernel(GaJobOut* outs, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	GaJobOut o;
	memset(&o, 0, sizeof(o));
	o.status = GA_NOT_RUN;
	outs[i] = o;
}

// pinned host blocks for the batches' copies of the reads (page-locking half a GB per batch costs more than copying it): a block goes
// back to the pool when the last holder -- the batch, or results whose edit sequences point into it -- lets go of it
struct PinnedPool
{
	std::mutex lock;
	std::vector<std::pair<size_t, char*>> idle;
	~PinnedPool() { for (auto& b : idle) hipHostFree(b.second); }
};

struct DevGraph : GaBackendGraph
{
	std::shared_ptr<PinnedPool> pinned = std::make_shared<PinnedPool>();
	bool buildsMatchWords() const override { return true; }
	std::shared_ptr<char> hostBlock(size_t bytes) override
	{
		const size_t two = (size_t)2 << 20;
		bytes = (bytes + two - 1) & ~(two - 1);
		std::shared_ptr<PinnedPool> pool = pinned;
		char* p = nullptr;
		size_t cap = 0;
		{
			std::lock_guard<std::mutex> guard(pool->lock);
			size_t best = pool->idle.size();
			for (size_t i = 0; i < pool->idle.size(); i++)
				if (pool->idle[i].first >= bytes && pool->idle[i].first <= 2 * bytes + two && (best == pool->idle.size() || pool->idle[i].first < pool->idle[best].first)) best = i;
			if (best != pool->idle.size()) { cap = pool->idle[best].first; p = pool->idle[best].second; pool->idle.erase(pool->idle.begin() + (long)best); }
		}
		if (!p)
		{
			hipSetDevice(device);
			if (hipHostMalloc((void**)&p, bytes, hipHostMallocDefault) != hipSuccess) return GaBackendGraph::hostBlock(bytes);      // (pageable memory: the upload is then a staged copy)
			cap = bytes;
		}
		return std::shared_ptr<char>(p, [pool, cap](char* q) {
			std::lock_guard<std::mutex> guard(pool->lock);
			if (pool->idle.size() < 6) pool->idle.emplace_back(cap, q); else hipHostFree(q);
		});
	}

	int device = 0;
	GaDevGraph g;
	GaHmmTables* hmm = nullptr;
	std::vector<void*> allocs;
	int cus = 0;
	uint64_t totalBp = 0;
	// scratch pool owned by the graph: batches reuse it instead of allocating tens of GB each
	uint8_t* pool = nullptr;
	size_t poolBytes = 0;
	bool poolBusy = false;
	~DevGraph() override
	{
		hipSetDevice(device);
		for (void* p : allocs) hipFree(p);
		for (auto& b : idleBlocks) hipFree(b.second);
		if (pool) hipFree(pool);
		if (hostPool) hipHostFree(hostPool);
	}
	// device buffers of the batches (match words, jobs, outputs, trace pool): handed back here instead of hipFree'd, because hipFree
	// waits for the whole device -- i.e. for the kernels of whatever batch is running -- and the stages of consecutive batches are
	// meant to overlap (sharding.run_overlapped).  A later batch of similar size takes them again.
	std::mutex blockLock;
	std::vector<std::pair<size_t, void*>> idleBlocks;
	void* takeBlock(size_t bytes)
	{
		{
			std::lock_guard<std::mutex> lock(blockLock);
			size_t best = idleBlocks.size();
			for (size_t i = 0; i < idleBlocks.size(); i++)
				if (idleBlocks[i].first >= bytes && idleBlocks[i].first <= bytes + bytes / 2 + 4096 && (best == idleBlocks.size() || idleBlocks[i].first < idleBlocks[best].first)) best = i;
			if (best != idleBlocks.size())
			{
				void* p = idleBlocks[best].second;
				idleBlocks.erase(idleBlocks.begin() + (long)best);
				return p;
			}
		}
		void* p = nullptr;
		if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
		return p;
	}
	void giveBlock(void* p, size_t bytes)
	{
		std::lock_guard<std::mutex> lock(blockLock);
		// (bounded: beyond a few dozen idle blocks the oldest go back to the device)
		if (idleBlocks.size() >= 48) { hipFree(idleBlocks.front().second); idleBlocks.erase(idleBlocks.begin()); }
		idleBlocks.emplace_back(bytes, p);
	}
	// (one graph can serve batches run from several host threads: the pool changes hands under a lock)
	std::mutex poolLock;
	uint8_t* takePool(size_t bytes)
	{
		std::lock_guard<std::mutex> lock(poolLock);
		if (poolBusy) return nullptr;
		if (bytes > poolBytes)
		{
			if (pool) hipFree(pool);
			pool = nullptr; poolBytes = 0;
			if (hipMalloc((void**)&pool, bytes) != hipSuccess) { pool = nullptr; return nullptr; }
			poolBytes = bytes;
		}
		poolBusy = true;
		return pool;
	}
	void givePool() { std::lock_guard<std::mutex> lock(poolLock); poolBusy = false; }
	size_t poolBytesIfFree() { std::lock_guard<std::mutex> lock(poolLock); return poolBusy ? 0 : poolBytes; }
	// pinned host buffer for the moves on their way back (page-locking half a GB per batch would cost more than the copy)
	std::mutex hostLock;
	uint8_t* hostPool = nullptr;
	size_t hostPoolBytes = 0;
	bool hostPoolBusy = false;
	uint8_t* takeHost(size_t bytes)
	{
		std::lock_guard<std::mutex> lock(hostLock);
		if (hostPoolBusy) return nullptr;
		if (bytes > hostPoolBytes)
		{
			if (hostPool) hipHostFree(hostPool);
			hostPool = nullptr; hostPoolBytes = 0;
			const size_t want = bytes + bytes / 8;
			if (hipHostMalloc((void**)&hostPool, want, hipHostMallocDefault) != hipSuccess) { hostPool = nullptr; return nullptr; }
			hostPoolBytes = want;
		}
		hostPoolBusy = true;
		return hostPool;
	}
	void giveHost() { std::lock_guard<std::mutex> lock(hostLock); hostPoolBusy = false; }
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

// the few switches the library reads from the environment, once per batch: GA_LANES (1 / 0: force which kernel goes first -- the GPU
// parity tests run every case both ways), GA_DEBUG_PASSES (one line per launch on stderr) and a test hook for the trace pool's size
struct Knobs
{
	int lanes = -1;                 // -1: by graph shape
	int spread = 1;                 // 1: a small batch is spread over all wave slots; 0: full waves; k > 1: k jobs per wave (GA_LANES_SPREAD, for experiments)
	bool debugPasses = false;
	uint64_t tracePoolBytes = 0;    // 0: sized from the batch
	Knobs()
	{
		if (const char* e = getenv("GA_LANES")) lanes = atoi(e) != 0 ? 1 : 0;
		debugPasses = getenv("GA_DEBUG_PASSES") != nullptr;
		if (const char* e = getenv("GA_LANES_SPREAD")) spread = atoi(e);
		if (const char* t = getenv("GA_TEST_TRACE_POOL_BYTES")) tracePoolBytes = (uint64_t)atoll(t) & ~3ull;
	}
};

struct DevBatch : GaBackendBatch
{
	DevGraph* g = nullptr;
	Knobs knobs;
	hipStream_t stream = nullptr;
	hipEvent_t evA = nullptr, evB = nullptr;
	std::vector<std::pair<size_t, void*>> allocs;     // (bytes, block) from the graph's block list
	std::vector<GaJob> jobs;
	GaRunConfig cfg;
	GaLaunch L;                 // what every pass shares (graph, reads, jobs, outputs, trace pool); scratch geometry is set per pass
	const uint64_t* dEq = nullptr;
	uint32_t* dList = nullptr;  // job list of the current pass
	size_t dListCap = 0;
	std::vector<GaJobOut> outs;
	bool outsLocked = false;
	std::vector<uint32_t> orderHost;   // all jobs, longest first (the device queues hand them out in this order)
	std::vector<uint8_t> passOf;       // which pass of the last run finished each job (0 = the first)
	int passNo = 0;
	GaRunStats st;
	GaRowsProvider rowsProvider;       // row codes for the wave-per-read kernels, uploaded when the first of them is about to run
	uint8_t* privateScratch = nullptr; // only when the graph's pool is taken by another batch
	size_t privateBytes = 0;

	~DevBatch() override
	{
		hipSetDevice(g->device);
		if (hostFromPool) g->giveHost();
		if (outsLocked) hipHostUnregister(outs.data());
		for (auto& a : allocs) g->giveBlock(a.second, a.first);
		if (privateScratch) hipFree(privateScratch);
		if (evA) hipEventDestroy(evA);
		if (evB) hipEventDestroy(evB);
		if (stream) hipStreamDestroy(stream);
	}
	template <typename T> int alloc(T** out, size_t count)
	{
		const size_t bytes = (std::max<size_t>(count * sizeof(T), 16) + 255) & ~(size_t)255;
		void* p = g->takeBlock(bytes);
		if (!p) { fprintf(stderr, "graphaligner_amd: no device memory for %zu bytes\n", bytes); return GA_E_DEVICE; }
		allocs.emplace_back(bytes, p);
		*out = (T*)p;
		return 0;
	}

	int ensureRows()
	{
		if (L.rows) return 0;
		if (builtRows) { L.rows = builtRows; return 0; }                          // (written by the match-word kernel)
		const std::vector<uint8_t>& rows = rowsProvider();
		uint8_t* dRows;
		if (alloc(&dRows, rows.size())) return GA_E_DEVICE;
		HIP_OK(hipMemcpyAsync(dRows, rows.data(), rows.size(), hipMemcpyHostToDevice, stream));
		L.rows = dRows;
		return 0;
	}

	uint8_t* builtRows = nullptr;      // the row codes, when the match-word kernel wrote them
	std::vector<uint8_t> badFills;     // per fill of the GaEqSource: a row outside IUPAC
	const std::vector<uint8_t>* invalidFills() const override { return &badFills; }
	bool emittingRuns() const override { return cfg.emit_runs != 0; }

	int init(const uint64_t* eq, const GaEqSource* src, size_t eqWords, const std::vector<GaJob>& jobsIn)
	{
		HIP_OK(hipSetDevice(g->device));
		// (non-blocking: a copy or a kernel of this batch must not wait for another batch's kernel through the null stream)
		HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
		HIP_OK(hipEventCreate(&evA));
		HIP_OK(hipEventCreate(&evB));
		jobs = jobsIn;
		memset(&L, 0, sizeof(L));
		L.graph = g->g;
		L.hmm = g->hmm;
		L.n_jobs = (uint32_t)jobs.size();
		L.initial_bw = cfg.initial_bw;
		L.ramp_bw = cfg.ramp_bw;
		L.max_slices = std::max<uint32_t>(cfg.max_slices, 1);
		GaJob* dJobs; uint64_t* eqDev;
		if (alloc(&eqDev, eqWords)) return GA_E_DEVICE;
		if (alloc(&dJobs, jobs.size())) return GA_E_DEVICE;
		if (!src && !eq) return GA_E_INVALID;
		if (src)
		{
			// the reads go up as they are (one transfer from the batch's pinned copy) and a kernel turns them into the match words
			uint8_t *dSeq, *dLut, *dFlags, *dRows; GaEqFill* dFills;
			const size_t rowBytes = (eqWords / 5) * 64 + 64;                         // (+ slack so a 64-byte row load never leaves the buffer)
			if (alloc(&dSeq, src->seqBytes + 64) || alloc(&dFills, src->nFills) || alloc(&dLut, 512) || alloc(&dFlags, src->nFills) || alloc(&dRows, rowBytes)) return GA_E_DEVICE;
			HIP_OK(hipMemsetAsync(dRows + rowBytes - 128, 0, 128, stream));
			builtRows = dRows;
			uint8_t lut[512];
			memcpy(lut, src->rowCode, 256); memcpy(lut + 256, src->rowCodeRc, 256);
			HIP_OK(hipMemcpyAsync(dSeq, src->seq, src->seqBytes, hipMemcpyHostToDevice, stream));
			HIP_OK(hipMemcpyAsync(dFills, src->fills, src->nFills * sizeof(GaEqFill), hipMemcpyHostToDevice, stream));
			HIP_OK(hipMemcpyAsync(dLut, lut, 512, hipMemcpyHostToDevice, stream));
			HIP_OK(hipMemsetAsync(eqDev + (eqWords - 5), 0, 40, stream));             // (the slack slice behind the last job)
			badFills.assign(src->nFills, 0);
			if (src->nFills)
			{
				const uint32_t blocks = (uint32_t)std::min<size_t>(src->nFills, (size_t)g->cus * 64);
				hipLaunchKernelGGL(ga_eq_words_kernel, dim3(blocks), dim3(64), 0, stream, dSeq, dFills, (uint32_t)src->nFills, dLut, (uint32_t)src->padCode, eqDev, dFlags, dRows);
				HIP_OK(hipGetLastError());
				HIP_OK(hipMemcpyAsync(badFills.data(), dFlags, src->nFills, hipMemcpyDeviceToHost, stream));
			}
			HIP_OK(hipStreamSynchronize(stream));                                      // (lut is on this stack; the flags decide emit_runs below)
			for (uint8_t f : badFills) if (f) { cfg.emit_runs = 0; break; }
		}
		else HIP_OK(hipMemcpyAsync(eqDev, eq, eqWords * 8, hipMemcpyHostToDevice, stream));
		HIP_OK(hipMemcpyAsync(dJobs, jobs.data(), jobs.size() * sizeof(GaJob), hipMemcpyHostToDevice, stream));
		L.rows = nullptr;
		L.jobs = dJobs;
		dEq = eqDev;
		// the device queues hand jobs out longest first: the short ones fill the tail of a launch, and the 64 jobs a lanes = reads
		// wave carries together have similar lengths
		orderHost.resize(jobs.size());
		for (uint32_t i = 0; i < jobs.size(); i++) orderHost[i] = i;
		std::stable_sort(orderHost.begin(), orderHost.end(), [&](uint32_t a, uint32_t b) { return jobs[a].n_rows > jobs[b].n_rows; });
		dListCap = std::max<size_t>(jobs.size(), 1);
		if (alloc(&dList, dListCap)) return GA_E_DEVICE;
		if (alloc(&L.outs, jobs.size())) return GA_E_DEVICE;
		// (the host copy of the records is page-locked: they come back after every pass, 128 bytes per job)
		outs.assign(jobs.size(), GaJobOut{});
		for (auto& o : outs) o.status = GA_NOT_RUN;
		outsLocked = !outs.empty() && hipHostRegister(outs.data(), outs.size() * sizeof(GaJobOut), hipHostRegisterDefault) == hipSuccess;
		if (alloc(&L.next_job, 4)) return GA_E_DEVICE;
		if (alloc(&L.trace_top, 2)) return GA_E_DEVICE;
		uint64_t totalRows = 0;
		for (auto& j : jobs) totalRows += j.n_rows;
		// one byte per move (a path makes at most ~1.5 moves per row), or 20 bytes per node run: room for a node change every 5 rows
		// (a job that does not fit reports GA_CAP_TRACE and is rerun by the ladder, which writes moves)
		L.trace_pool_cap = ((totalRows * (cfg.emit_runs ? 4 : 1) + totalRows / 2 + 256ull * jobs.size() + 4096) + 3) & ~3ull;
		// (test hook: a deliberately small pool, so that the claims' overflow path runs -- tests/parity_cases.py)
		if (knobs.tracePoolBytes) L.trace_pool_cap = knobs.tracePoolBytes;
		if (alloc(&L.traces, L.trace_pool_cap)) return GA_E_DEVICE;
		HIP_OK(hipStreamSynchronize(stream));
		return 0;
	}

	// scratch for one pass: the graph's pool, or a private allocation while another batch holds the pool
	uint8_t* takeScratch(size_t bytes, bool& fromPool)
	{
		uint8_t* p = g->takePool(bytes);
		fromPool = p != nullptr;
		if (p) return p;
		if (bytes > privateBytes)
		{
			if (privateScratch) hipFree(privateScratch);
			privateScratch = nullptr; privateBytes = 0;
			if (hipMalloc((void**)&privateScratch, bytes) != hipSuccess) { privateScratch = nullptr; return nullptr; }
			privateBytes = bytes;
		}
		return privateScratch;
	}
	size_t scratchBudget() const
	{
		size_t freeB = 0, totalB = 0;
		if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return 0;
		const size_t poolFree = g->poolBytesIfFree();
		return (size_t)((freeB + (poolFree ? poolFree : privateBytes)) * 0.85);
	}
	int uploadList(const std::vector<uint32_t>& list)
	{
		HIP_OK(hipMemcpyAsync(dList, list.data(), list.size() * 4, hipMemcpyHostToDevice, stream));
		return 0;
	}
	int afterPass(float& ms)
	{
		HIP_OK(hipGetLastError());
		HIP_OK(hipEventRecord(evB, stream));
		HIP_OK(hipMemcpyAsync(outs.data(), L.outs, outs.size() * sizeof(GaJobOut), hipMemcpyDeviceToHost, stream));
		HIP_OK(hipStreamSynchronize(stream));
		HIP_OK(hipEventElapsedTime(&ms, evA, evB));
		st.kernel_ms += ms;
		return 0;
	}

	static bool isCapacity(int s) { return s == GA_CAP_NODES || s == GA_CAP_COLS || s == GA_CAP_ARENA || s == GA_CAP_TRACE || s == GA_CAP_HEAP || s == GA_PUNT || s == GA_NOT_RUN; }
	// bands with cycles and ramp redos: only the general wave-per-read variants carry those paths
	static bool needsGeneral(int s) { return s == GA_UNSUPPORTED_CYCLE || s == GA_UNSUPPORTED_RAMP; }
	// what a lanes = reads variant with larger tables can still take
	static bool widerLanes(int s) { return s == GA_CAP_NODES || s == GA_CAP_COLS || s == GA_CAP_ARENA || s == GA_CAP_TRACE || s == GA_CAP_HEAP; }

	// one launch of the lanes = reads kernel over `list` (job indices, longest first)
	template <int N, int LW> int lanesPass(const std::vector<uint32_t>& list, uint32_t rowsPerSlice, bool first)
	{
		if (list.empty()) return 0;
		for (uint32_t i : list) passOf[i] = (uint8_t)passNo;
		passNo++;
		gal::GaLanesLaunch P;
		memset(&P, 0, sizeof(P));
		P.graph = L.graph; P.hmm = L.hmm; P.eq = dEq; P.jobs = L.jobs; P.outs = L.outs;
		P.job_list = dList; P.n_jobs = (uint32_t)list.size();
		P.lanes_per_wave = LW;
		P.next_group = L.next_job;
		P.traces = L.traces; P.trace_top = L.trace_top; P.trace_pool_cap = L.trace_pool_cap;
		P.initial_bw = L.initial_bw; P.ramp_bw = L.ramp_bw;
		P.emit_runs = cfg.emit_runs;
		uint32_t maxRows = 0;
		for (uint32_t i : list) maxRows = std::max(maxRows, jobs[i].n_rows);
		P.max_slices = std::max<uint32_t>(maxRows / 64, 1);
		P.cap_cols = std::min<uint32_t>(N * 256u, 0xff00u);
		// (node runs are five words each: the staging plane holds a run per ten rows -- six times what a path over 64-bp nodes makes; a
		// job that needs more reports GA_CAP_TRACE and climbs on)
		P.cap_moves = maxRows * 2 + 1024;
		const uint32_t ldsBytes = gal::Lay<N>::WORDS * LW * 4 + LW * gal::kStageWords64 * 8 + 512;
		const uint32_t wavesPerCu = std::max<uint32_t>(1, std::min<uint32_t>(8, 163840u / ldsBytes));
		// arena rows of a wave = the band columns of its lanes (in blocks of 8 per node), `rowsPerSlice` per lane and slice: sized for the
		// lanes a wave will really carry
		auto layoutFor = [&](uint32_t lanes) {
			P.cap_rows = (uint32_t)std::min<uint64_t>(((uint64_t)P.max_slices * rowsPerSlice * lanes + 64 + 7u) & ~7ull, 0x7ffffff0ull);
			return gal::wave_layout<N>(P.cap_cols, P.cap_rows, P.max_slices, P.cap_moves, LW);
		};
		gal::WaveLayout lay = layoutFor(LW);
		const uint64_t fit = scratchBudget() / std::max<uint64_t>(lay.bytes, 1);
		const uint64_t slotsHere = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)g->cus * wavesPerCu, std::max<uint64_t>(fit, 64)));
		// a SMALL batch is spread over all wave slots: a wave's steps cost the same with fewer lanes, and fewer lanes wait for each
		// other less.  A batch that gives at least six of ten slots a full wave runs in full waves: round 3's kernel is more sensitive
		// to the waves it shares its CU's memory path with than to the lanes it waits for (50 000 reads: 782 full waves 28.5 ms,
		// 1 021 waves of 49 reads 30.2 ms on the same box, profiles/r3_ab_lanes_per_wave.txt; round 2's kernel: the other way round)
		uint32_t lanesPer = LW;
		if (knobs.spread > 1) lanesPer = (uint32_t)std::min<int>(LW, knobs.spread);
		else if (knobs.spread && list.size() * 10 < slotsHere * LW * 6)
			lanesPer = (uint32_t)std::min<uint64_t>(LW, std::max<uint64_t>(8, (list.size() + slotsHere - 1) / slotsHere));
		P.lanes_per_wave = lanesPer;
		lay = layoutFor(lanesPer);
		P.wave_bytes = lay.bytes;
		const uint64_t groups = (list.size() + lanesPer - 1) / lanesPer;
		const uint32_t waves = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(slotsHere, groups));
		bool fromPool = false;
		uint8_t* scratch = takeScratch((size_t)waves * lay.bytes, fromPool);
		if (!scratch) return 0;                            // no memory for this variant: the jobs keep their status and climb on
		P.scratch = scratch;
		int rc = uploadList(list);
		if (!rc)
		{
			hipMemsetAsync(L.next_job, 0, 16, stream);
			hipEventRecord(evA, stream);
			hipLaunchKernelGGL((ga_lanes_kernel<N, LW>), dim3(waves), dim3(64), 0, stream, P);
			float ms = 0;
			rc = afterPass(ms);
			if (first) { st.main_ms = ms; st.main_variant = N * 1000 + 80 + (LW == 64 ? 0 : 1); st.slots = waves; st.waves_per_cu = wavesPerCu; st.scratch_bytes = (uint64_t)waves * lay.bytes; }
			if (knobs.debugPasses) fprintf(stderr, "graphaligner_amd: lanes pass <%d,%d>: %zu jobs on %u waves (%.1f GB scratch), %.2f ms\n", N, LW, list.size(), waves, waves * (double)lay.bytes / 1e9, ms);
#if GA_STAMPS == 3
			{
				uint64_t lo = ~0ull, hi = 0, n = 0; double life = 0, cyc = 0, lateStart = 0;
				for (auto& o : outs) if (o.stamps[1]) { lo = std::min(lo, o.stamps[0]); hi = std::max(hi, o.stamps[1]); }
				for (auto& o : outs) if (o.stamps[1]) { n++; life += (double)(o.stamps[1] - o.stamps[0]); cyc += (double)o.stamps[4]; lateStart = std::max(lateStart, (double)(o.stamps[0] - lo)); }
				{
					// the spread of the waves' lifetimes, and what goes with a long one (fill steps are not known per wave; traceback rounds are)
					std::vector<std::pair<double, uint64_t>> lives;
					for (auto& o : outs) if (o.stamps[1]) lives.push_back({(double)(o.stamps[1] - o.stamps[0]) / 1e5, o.stamps[7]});
					std::sort(lives.begin(), lives.end());
					auto at = [&](double q) { return lives[(size_t)(q * (lives.size() - 1))]; };
					if (!lives.empty()) fprintf(stderr, "graphaligner_amd: wave life ms p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f; traceback rounds of the p10 / p50 / p99 / max wave: %llu / %llu / %llu / %llu, fast iterations %llu / %llu / %llu / %llu\n",
						at(0.1).first, at(0.5).first, at(0.9).first, at(0.99).first, at(1.0).first,
						(unsigned long long)(at(0.1).second & 0xffffffffu), (unsigned long long)(at(0.5).second & 0xffffffffu), (unsigned long long)(at(0.99).second & 0xffffffffu), (unsigned long long)(at(1.0).second & 0xffffffffu),
						(unsigned long long)(at(0.1).second >> 32), (unsigned long long)(at(0.5).second >> 32), (unsigned long long)(at(0.99).second >> 32), (unsigned long long)(at(1.0).second >> 32));
				}
				if (n) fprintf(stderr, "graphaligner_amd: %llu waves: first start to last end %.3f ms, mean wave life %.3f ms = %.1f M shader cycles (%.2f GHz), latest start %.3f ms after the first\n",
				               (unsigned long long)n, (hi - lo) / 1e5, life / n / 1e5, cyc / n / 1e6, cyc / life / 10.0, lateStart / 1e5);
			}
#endif
		}
		if (fromPool) g->givePool();
		return rc;
	}

	template <int MAXN, bool GENERAL, bool SPARSE = false> int retryPass(uint32_t capCols, uint64_t arenaWordsPerSlice, uint32_t traceMul, uint32_t wavesPerCuRetry, bool takeCapacity, bool takeGeneral,
	                                                                    uint64_t arenaWordsExtra = 0)
	{
		std::vector<uint32_t> again;
		for (uint32_t i : orderHost)
			if ((takeCapacity && isCapacity(outs[i].status)) || (takeGeneral && needsGeneral(outs[i].status)) || (SPARSE && outs[i].status == GA_UNSUPPORTED_BAND)) again.push_back(i);
		if (again.empty()) return 0;
		if (ensureRows()) return GA_E_DEVICE;
		for (uint32_t i : again) passOf[i] = (uint8_t)passNo;
		passNo++;
		st.jobs_retried += again.size();
		GaLaunch Rl = L;
		uint32_t maxRows = 0;
		for (uint32_t i : again) maxRows = std::max(maxRows, jobs[i].n_rows);
		Rl.cap_cols = capCols;
		Rl.trace_cap = maxRows * traceMul + 4096;
		Rl.max_slices = std::max<uint32_t>(maxRows / 64, 1);
		Rl.arena_words = std::min<uint64_t>(64 + (uint64_t)(maxRows / 64) * (gak::kSliceHdrWords + arenaWordsPerSlice) + arenaWordsExtra, 0xfffffff0ull);
		Rl.sparse_bw = SPARSE ? (uint32_t)std::max(std::max(L.initial_bw, L.ramp_bw), 1) : 0u;
		SlotLayout lay = slotLayout(Rl.cap_cols, Rl.max_slices, Rl.arena_words, Rl.trace_cap, Rl.sparse_bw);
		Rl.slot_bytes = lay.bytes;
		const uint64_t fit = scratchBudget() / lay.bytes;
		uint32_t rslots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>((uint64_t)g->cus * wavesPerCuRetry, fit), again.size()));
		bool fromPool = false;
		uint8_t* scratch = takeScratch((size_t)rslots * lay.bytes, fromPool);
		if (!scratch) return 0;                            // the affected jobs keep their capacity status
		Rl.scratch = scratch;
		if (passNo == 1) { st.slots = rslots; st.waves_per_cu = wavesPerCuRetry; st.scratch_bytes = (uint64_t)rslots * lay.bytes; }
		Rl.job_list = dList;
		Rl.n_jobs = (uint32_t)again.size();
		int rc = uploadList(again);
		if (!rc)
		{
			hipMemsetAsync(Rl.next_job, 0, 16, stream);
			// (the sparse method's tables are generation-stamped: they start from zero once per launch)
			if (SPARSE) hipMemset2DAsync(scratch + lay.sparse, lay.bytes, 0, gak::sparse_mem_bytes(Rl.sparse_bw), rslots, stream);
			hipEventRecord(evA, stream);
			hipLaunchKernelGGL((ga_extend_kernel<MAXN, GENERAL, SPARSE>), dim3(rslots), dim3(64), 0, stream, Rl);
			float ms = 0;
			rc = afterPass(ms);
			if (knobs.debugPasses) fprintf(stderr, "graphaligner_amd: wave-per-read pass <%d,%d>: %zu jobs on %u slots, %.2f ms\n", MAXN, (int)GENERAL, again.size(), rslots, ms);
		}
		if (fromPool) g->givePool();
		return rc;
	}

	int run() override
	{
		HIP_OK(hipSetDevice(g->device));
		st = GaRunStats();
		passOf.assign(jobs.size(), 0);
		passNo = 0;
		if (jobs.empty()) return 0;
		// (the records are initialised on the device and come back whole after every pass: nothing to upload here; the host copy's
		// statuses are what the passes select their jobs by, so a batch that is run again starts from "not run" here too)
		for (auto& o : outs) o.status = GA_NOT_RUN;
		hipLaunchKernelGGL(ga_outs_init_kernel, dim3((uint32_t)((jobs.size() + 255) / 256)), dim3(256), 0, stream, L.outs, (uint32_t)jobs.size());
		HIP_OK(hipGetLastError());
		HIP_OK(hipMemsetAsync(L.trace_top, 0, 16, stream));
		// ---- first the lanes = reads kernel: one job per lane.  Its LDS tables hold 16, 32 or 64 band nodes per lane (4, 2, 1
		// waves per CU); the starting size follows the graph's mean node length, and jobs a size cannot hold move to the next ----
		// It is the first pass where its lockstep pays: the lanes of a wave take their k-th band node together, so on graphs of long,
		// equally long nodes (mean node length >= 40 bp: linear and sparsely branching graphs) every lane is busy; on graphs chopped
		// into short uneven nodes the wave-per-read kernel is still the faster one and goes first.  GA_LANES=1 / 0 forces the choice.
		const double meanNode = g->g.n_nodes > 2 ? (double)g->totalBp / (double)(g->g.n_nodes - 2) : 64.0;
		const bool useLanes = knobs.lanes >= 0 ? knobs.lanes != 0 : meanNode >= 40;
		int rc = 0;
		if (useLanes)
		{
			const int startN = meanNode >= 40 ? 0 : meanNode >= 14 ? 1 : 2;
			std::vector<uint32_t> list = orderHost;
			bool first = true;
			for (int n = startN; n <= 2 && !list.empty() && !rc; n++)
			{
				// later sizes are only worth a launch of their own for enough jobs to fill waves; stragglers take the wave-per-read ladder
				if (!first && list.size() < 512) break;
				// (10 / 24 / 56 band nodes per lane = 4 / 2 / 1 waves per CU next to the 12.8 KB block image)
				// (rows per lane and slice: the band's columns rounded up to blocks of 8 per node, with room for the widest slices)
				if (n == 0) rc = lanesPass<10, 64>(list, 288, first);
				// (the wider tables trade lanes for LDS per lane: 32 and 16 lanes per wave keep four waves on every CU -- the same number of
				// reads in flight as 64-lane waves that leave three SIMDs of four idle, in four times as many instruction streams)
				else if (n == 1) rc = lanesPass<24, 32>(list, 480, first);
				else rc = lanesPass<56, 16>(list, 768, first);
				first = false;
				std::vector<uint32_t> again;
				for (uint32_t i : list) if (widerLanes(outs[i].status)) again.push_back(i);
				list.swap(again);
			}
			if (rc) return rc;
		}
		// ---- what is left climbs the wave-per-read ladder: 64 band nodes in LDS; then the general variants, which also carry the
		// paths for bands with cycles and for ramp redos; last 256 band nodes with large buffers ----
		st.jobs_retried = 0;
		if (!useLanes)
		{
			// the lean wave-per-read variant over everything (32 band nodes in LDS, 24 waves per CU)
#ifndef GA_FIRST_N
#define GA_FIRST_N 32
#endif
			rc = retryPass<GA_FIRST_N, false>(4096, 3 * 64 + 5 * 800, 2, 24, true, false);
			if (rc) return rc;
			st.main_ms = st.kernel_ms;
			st.main_variant = -GA_FIRST_N;                 // (negative: the wave-per-read kernel with that many band nodes in LDS)
			st.jobs_retried = 0;
		}
		rc = retryPass<64, false>(8192, 3 * 64 + 5 * 2048, 3, 12, true, false);
		if (rc) return rc;
		rc = retryPass<64, true>(8192, 3 * 64 + 5 * 4096, 4, 8, false, true);      // only what needs the extra paths: capacity misses go straight on
		if (rc) return rc;
		rc = retryPass<256, true>(65536, 3 * 256 + 5 * 8192, 6, 4, true, true);
		if (rc) return rc;
		// bands of 200 000 cells and more (the reference's sparse method and backtrace override, ga_sparse.h): a fallback, run for the
		// jobs that met such a band, with room for every column of every node such a slice touches
		// (also the last resort for what the passes before could not hold: bit-vector bands may have up to 199 999 columns)
		rc = retryPass<256, true, true>(1u << 20, 3 * 256 + 5 * 20000, 8, 1, true, true, 48ull << 20);
		for (int k = 0; k < 8; k++) { st.stamps[k] = 0; for (auto& o : outs) st.stamps[k] += o.stamps[k]; }
		return rc;
	}

	uint8_t* hostTraces = nullptr;       // the graph's pinned buffer (hostFromPool) or a private one
	bool hostFromPool = false;
	std::unique_ptr<uint8_t[]> hostPrivate;
	size_t hostPrivateBytes = 0;
	int fetch(std::vector<GaJobOut>& o, const uint8_t** traces, uint64_t* nBytes) override
	{
		HIP_OK(hipSetDevice(g->device));
		fetchDone();
		o = outs;
		for (size_t i = 0; i < o.size(); i++) o[i].reserved2 = passOf[i];      // 0 = finished by the first pass
		uint64_t top = 0;
		// (on the batch's own stream: the null stream would make this wait for whatever kernel another batch is running)
		HIP_OK(hipMemcpyAsync(&top, L.trace_top, 8, hipMemcpyDeviceToHost, stream));
		HIP_OK(hipStreamSynchronize(stream));
		top = std::min<uint64_t>(top, L.trace_pool_cap);
		hostTraces = g->takeHost(top + 64);
		hostFromPool = hostTraces != nullptr;
		if (!hostTraces)
		{
			// another batch of this graph is being assembled from the pinned buffer: pageable memory, not initialised first
			if (hostPrivateBytes < top + 64) { hostPrivate.reset(new uint8_t[top + 64]); hostPrivateBytes = top + 64; }
			hostTraces = hostPrivate.get();
		}
		if (top)
		{
			HIP_OK(hipMemcpyAsync(hostTraces, L.traces, top, hipMemcpyDeviceToHost, stream));
			HIP_OK(hipStreamSynchronize(stream));
		}
		*traces = hostTraces;
		*nBytes = top;
		return 0;
	}
	void fetchDone() override
	{
		if (hostFromPool) g->giveHost();
		hostFromPool = false;
		hostTraces = nullptr;
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
	if (flat.node_start.size() - 1 >= 0xfffffff0ull) { *status = GA_E_INVALID; return nullptr; }      // node indices are 32-bit
	DevGraph* g = new DevGraph();
	g->device = device;
	g->cus = prop.multiProcessorCount;
	g->g.n_nodes = (uint32_t)(flat.node_start.size() - 1);
	g->g.reserved = 0;
	g->totalBp = flat.node_start.back();
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

GaBackendBatch* ga_backend_create_batch(GaBackendGraph* graph, GaRowsProvider rows, const uint64_t* eq, const GaEqSource* src, size_t eqWords, const std::vector<GaJob>& jobs,
                                        const GaRunConfig& cfg, int* status)
{
	DevBatch* b = new DevBatch();
	b->g = static_cast<DevGraph*>(graph);
	b->cfg = cfg;
	b->rowsProvider = rows;
	int s = b->init(eq, src, eqWords, jobs);
	if (s) { delete b; *status = s; return nullptr; }
	*status = 0;
	return b;
}
