// ga_wave.h -- the handful of wave64 primitives the alignment program is written in, for gfx950: a per-lane value IS a
// register; the cross-lane operations are DPP moves (wave_shr / row_shr / row_bcast), v_readlane and ballots.  64 lanes = the 64 read
// rows of one slice.
//
// (The tests check the same program on the host: tests/emul/ga_wave_emul.h defines this vocabulary with a per-lane value as an
// array of 64 and every primitive as a loop, and the test build includes it INSTEAD of this file through GA_WAVE_HEADER.  That
// back end is test infrastructure and lives with the tests.)
#pragma once
#include <stdint.h>

namespace gaw {

constexpr int LANES = 64;
constexpr int INF = 0x3fffffff;

// =========================================================================================
// gfx950 device
// =========================================================================================
#define GA_FN __device__ __forceinline__
#define GA_LANE0 (threadIdx.x == 0)

typedef bool VB;
typedef int VI;
typedef uint64_t VU;

// v_mov_b32 dpp wave_shl:1 -- lane 63 has no source lane and keeps `fill`
GA_FN VI shl1(VI x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x130, 0xf, 0xf, false); }
GA_FN VI lane_gather(VI x, VI idx) { return __builtin_amdgcn_ds_bpermute((idx & 63) << 2, x); }
// v_readlane costs ~5 SIMD cycles on gfx950 (tools/ubench_valu.hip); a uniform-address ds_bpermute gives the
// same value in a VGPR on the LDS pipe instead
GA_FN VI lane_broadcast(VI x, int lane) { return __builtin_amdgcn_ds_bpermute((lane & 63) << 2, x); }
GA_FN VI shr1v(VI x, VI fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
GA_FN VI bit_extract_v(VI x, VI bit) { return (int)__builtin_amdgcn_ubfe((unsigned)x, (unsigned)bit, 1u); }
GA_FN VU mask_low_bits(VI nbits) { return nbits <= 0 ? 0ull : nbits >= 64 ? ~0ull : ((1ull << nbits) - 1); }
GA_FN VI bit64_at(VU w, VI pos) { return (pos < 0 || pos > 63) ? 0 : (int)((w >> pos) & 1); }
GA_FN VI lane_iota() { return (int)threadIdx.x; }
GA_FN VI vmin(VI a, VI b) { return a < b ? a : b; }
GA_FN VI select(VB c, VI a, VI b) { return c ? a : b; }
GA_FN VU select(VB c, VU a, VU b) { return c ? a : b; }
GA_FN VI bit_extract(VI x, int bit) { return (int)__builtin_amdgcn_ubfe((unsigned)x, (unsigned)bit, 1u); }
GA_FN VI vpopc(VU a) { return __builtin_popcountll(a); }
template <int B> GA_FN VI vmod(VI x) { return (int)((unsigned)x % (unsigned)B); }      // constant divisor: a multiply-high and a subtract
GA_FN VU low_mask_through_lane() { return threadIdx.x == 63 ? ~0ull : ((2ull << threadIdx.x) - 1); }

// v_mov_b32 dpp wave_shr:1 -- lane 0 has no source lane and keeps `fill`
GA_FN VI shr1(VI x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
// wave64 inclusive scan: row_shr 1,2,4,8 inside each row of 16, then row_bcast:15 into rows
// 1 and 3, then row_bcast:31 into rows 2 and 3
GA_FN VI prefix_min(VI v)
{
	// old = INT_MAX is the identity of signed min, which lets the DPP combiner fold each move into v_min_i32_dpp
	const int ID = 0x7fffffff;
	v = vmin(v, __builtin_amdgcn_update_dpp(ID, v, 0x111, 0xf, 0xf, false));
	v = vmin(v, __builtin_amdgcn_update_dpp(ID, v, 0x112, 0xf, 0xf, false));
	v = vmin(v, __builtin_amdgcn_update_dpp(ID, v, 0x114, 0xf, 0xf, false));
	v = vmin(v, __builtin_amdgcn_update_dpp(ID, v, 0x118, 0xf, 0xf, false));
	v = vmin(v, __builtin_amdgcn_update_dpp(ID, v, 0x142, 0xa, 0xf, false));
	v = vmin(v, __builtin_amdgcn_update_dpp(ID, v, 0x143, 0xc, 0xf, false));
	return v;
}
GA_FN uint64_t ballot(VB c) { return __ballot(c); }
GA_FN int read_lane(VI x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
// v_writelane_b32: this clang has no builtin for it, so bind the LLVM intrinsic by name (the
// compiler still schedules it and pads its hazards, unlike inline asm)
extern "C" __device__ int ga_llvm_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
GA_FN VI write_lane(VI x, int value, int lane) { return ga_llvm_writelane_i32(value, lane, x); }
GA_FN VU make_vu(VI lo, VI hi) { return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; }
GA_FN uint64_t read_lane(VU x, int lane)
{
	uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
	uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
	return ((uint64_t)hi << 32) | lo;
}
template <typename T> GA_FN VI load_lanes(const T* p, int count, int fill) { return (int)threadIdx.x < count ? (int)p[threadIdx.x] : fill; }
template <typename T> GA_FN void store_lanes(T* p, int count, VI x) { if ((int)threadIdx.x < count) p[threadIdx.x] = (T)x; }
GA_FN void store_lanes(uint64_t* p, int count, VU x) { if ((int)threadIdx.x < count) p[threadIdx.x] = x; }
GA_FN VU load_lanes_u64(const uint64_t* p, int count) { return (int)threadIdx.x < count ? p[threadIdx.x] : 0ull; }
template <typename T> GA_FN VI gather(const T* p, VI idx) { return (int)p[idx]; }
GA_FN VU gather64(const uint64_t* p, VI idx) { return p[idx]; }
GA_FN VI gather_rec(const uint32_t* p, VI idx, int word) { return (int)p[(uint64_t)(uint32_t)idx * 16 + (uint32_t)word]; }
template <typename T> GA_FN void scatter(T* p, VI idx, VI x, VB m) { if (m) p[idx] = (T)x; }
GA_FN void scatter64(uint64_t* p, VI idx, VU x, VB m) { if (m) p[idx] = x; }
// one wave per workgroup: orders this wave's LDS / global traffic (s_waitcnt + s_barrier)
GA_FN int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
GA_FN void wave_sync() { __syncthreads(); }
// compiler-only ordering point: a single wave executes its LDS traffic in order, no wait is needed
GA_FN void wave_order() { __builtin_amdgcn_wave_barrier(); }
#ifdef GA_STAMPS
GA_FN uint64_t stamp() { return __builtin_readcyclecounter(); }
#else
GA_FN uint64_t stamp() { return 0; }
#endif
// lane 0 performs the device-scope atomic, the old value is broadcast to the wave
GA_FN uint32_t wave_atomic_add(uint32_t* p, uint32_t v)
{
	uint32_t r = 0;
	if (threadIdx.x == 0) r = atomicAdd(p, v);
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
GA_FN uint64_t wave_atomic_add(uint64_t* p, uint64_t v)
{
	unsigned long long r = 0;
	if (threadIdx.x == 0) r = atomicAdd((unsigned long long*)p, (unsigned long long)v);
	uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
	uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r >> 32));
	return ((uint64_t)hi << 32) | lo;
}

GA_FN uint64_t wave_claim(uint64_t* p, uint64_t bytes, uint64_t cap)
{
	unsigned long long r = ~0ull;
	if (threadIdx.x == 0)
	{
		unsigned long long seen = *(volatile unsigned long long*)p;
		while (seen + bytes <= cap)
		{
			const unsigned long long prev = atomicCAS((unsigned long long*)p, seen, seen + (unsigned long long)bytes);
			if (prev == seen) { r = seen; break; }
			seen = prev;
		}
	}
	uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
	uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r >> 32));
	return ((uint64_t)hi << 32) | lo;
}

}  // namespace gaw
