// ubench_fill_regime.hip -- the lanes = reads fill loop in the regime the benchmark batch really offers:
// 50 000 reads = 782 waves of 64 reads on 1024 SIMDs (less than one wave per SIMD), every wave running a long
// dependent chain of column steps.  Measures, per record layout, what one column step costs when nothing else hides
// its latency:  R = how many consecutive columns of one lane share a contiguous block ([row / R][lane][R][24 B]).
//   R = 1: every step is one coalesced 1.5 KB row;  R = 8: every lane writes into its own 192-byte block.
// Also a traceback-shaped loop: one dependent 24-byte record fetch per step, per lane at its own row.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_fill_regime tools/ubench_fill_regime.hip && tools/ubench_fill_regime
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bitvector_column_step.h"

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R>
__device__ __forceinline__ size_t rec_off(uint32_t row, int lane)           // byte offset of (row, lane) inside a wave's arena
{
	return ((size_t)(row / R) * 64 + lane) * (R * 24) + (size_t)(row % R) * 24;
}

template <int R, int LANES>
__global__ void __launch_bounds__(64) k_fill(const uint32_t* __restrict__ seq2, size_t genomeWords, const uint64_t* __restrict__ eqTab,
                                              uint8_t* __restrict__ arena, size_t arenaBytesPerWave, uint32_t* __restrict__ endBuf, int steps, int colsPerSlice)
{
	const int lane = threadIdx.x;
	if (lane >= LANES) return;
	const size_t blk = blockIdx.x;
	uint8_t* my = arena + blk * arenaBytesPerWave;
	uint32_t* endPrev = endBuf + blk * 2 * (size_t)colsPerSlice * 64;
	uint32_t* endCur = endPrev + (size_t)colsPerSlice * 64;
	uint64_t e0 = eqTab[(blk * 4 + 0) * 64 + lane], e1 = eqTab[(blk * 4 + 1) * 64 + lane], e2 = eqTab[(blk * 4 + 2) * 64 + lane], e3 = eqTab[(blk * 4 + 3) * 64 + lane];
	uint64_t vp = ~0ull, vn = 0;
	int before = 1000 + lane;
	size_t gcol = ((size_t)(blk * 64 + lane) * 7919u * 64u) % (genomeWords * 16 - (size_t)steps - 64);
	uint32_t bases = 0;
	uint32_t row = 0;
	int c = 0;
	for (int k = 0; k < steps; k++)
	{
		if ((gcol & 15) == 0 || k == 0) bases = seq2[gcol >> 4];
		const int b = (bases >> ((gcol & 15) * 2)) & 3;
		gcol++;
		const uint32_t pe = endPrev[(size_t)c * 64 + lane];
		int calc = before + 1;
		const int above = (int)(pe >> 3) + before - 3;                       // keeps |hin| <= 1 while still depending on the load
		int d = 0;
		if (calc > above && (pe & 4)) { d = calc - above; d = d > 2 ? 2 : d; }
		const int flags = b | ((pe & 8) ? 4 : 0) | (d << 3);
		bitvector_column_step(vp, vn, before, calc, flags, e0, e1, e2, e3);
		const int end = before + __builtin_popcountll(vp) - __builtin_popcountll(vn);
		const uint32_t ew = ((uint32_t)end << 3) | (uint32_t)(vp >> 63) | ((uint32_t)(vn >> 63) << 1);
		uint8_t* rec = my + rec_off<R>(row, lane);
		*(uint4*)rec = make_uint4((uint32_t)vp, (uint32_t)(vp >> 32), (uint32_t)vn, (uint32_t)(vn >> 32));
		*(uint2*)(rec + 16) = make_uint2((uint32_t)before, ew);
		endCur[(size_t)c * 64 + lane] = ew;
		row++;
		if (++c == colsPerSlice)
		{
			c = 0;
			uint32_t* t = endPrev; endPrev = endCur; endCur = t;
			e0 = e0 * 0x9E3779B97F4A7C15ull + 1; e1 ^= e0 >> 7; e2 += e1; e3 ^= e2 << 3;       // "next slice's" match words
		}
	}
}


// the same loop with the operands of the next chunk of U columns requested one chunk ahead (what the real kernel does)
template <int R, int LANES, int U>
__global__ void __launch_bounds__(64) k_fill_pf(const uint32_t* __restrict__ seq2, size_t genomeWords, const uint64_t* __restrict__ eqTab,
                                              uint8_t* __restrict__ arena, size_t arenaBytesPerWave, uint32_t* __restrict__ endBuf, int steps, int colsPerSlice)
{
	const int lane = threadIdx.x;
	if (lane >= LANES) return;
	const size_t blk = blockIdx.x;
	uint8_t* my = arena + blk * arenaBytesPerWave;
	uint32_t* endPrev = endBuf + blk * 2 * (size_t)colsPerSlice * 64;
	uint32_t* endCur = endPrev + (size_t)colsPerSlice * 64;
	uint64_t e0 = eqTab[(blk * 4 + 0) * 64 + lane], e1 = eqTab[(blk * 4 + 1) * 64 + lane], e2 = eqTab[(blk * 4 + 2) * 64 + lane], e3 = eqTab[(blk * 4 + 3) * 64 + lane];
	uint64_t vp = ~0ull, vn = 0;
	int before = 1000 + lane;
	size_t gcol = ((size_t)(blk * 64 + lane) * 7919u * 64u) % (genomeWords * 16 - (size_t)steps - 64);
	uint32_t row = 0;
	int c = 0;
	uint32_t cur[U], nxt[U];
	uint64_t bcur, bnxt;
	for (int i = 0; i < U; i++) cur[i] = endPrev[(size_t)(c + i) * 64 + lane];
	bcur = (uint64_t)seq2[gcol >> 4] | ((uint64_t)seq2[(gcol >> 4) + 1] << 32);
	for (int k = 0; k < steps; k += U)
	{
		// colsPerSlice is a multiple of U here, so a chunk never crosses the slice end
		const bool last = c + U == colsPerSlice;
		const uint32_t* src = last ? endCur : endPrev;
		const int cn = last ? 0 : c + U;
		for (int i = 0; i < U; i++) nxt[i] = src[(size_t)(cn + i) * 64 + lane];
		bnxt = (uint64_t)seq2[(gcol + U) >> 4] | ((uint64_t)seq2[((gcol + U) >> 4) + 1] << 32);
#pragma unroll
		for (int i = 0; i < U; i++)
		{
			const int b = (int)(bcur >> (((gcol & 15) + i) * 2)) & 3;
			const uint32_t pe = cur[i];
			int calc = before + 1;
			const int above = (int)(pe >> 3) + before - 3;
			const bool re = calc > above && (pe & 4);
			bitvector_column_step(vp, vn, before, re ? above : calc, b | ((pe & 8) ? 4 : 0), e0, e1, e2, e3);
			const int end = before + __builtin_popcountll(vp) - __builtin_popcountll(vn);
			const uint32_t ew = ((uint32_t)end << 3) | (uint32_t)(vp >> 63) | ((uint32_t)(vn >> 63) << 1);
			uint8_t* rec = my + rec_off<R>(row, lane);
			*(uint4*)rec = make_uint4((uint32_t)vp, (uint32_t)(vp >> 32), (uint32_t)vn, (uint32_t)(vn >> 32));
			*(uint2*)(rec + 16) = make_uint2((uint32_t)before, ew);
			endCur[(size_t)(c + i) * 64 + lane] = ew;
			row++;
		}
		gcol += U;
		c += U;
		if (c == colsPerSlice)
		{
			c = 0;
			uint32_t* t = endPrev; endPrev = endCur; endCur = t;
			e0 = e0 * 0x9E3779B97F4A7C15ull + 1; e1 ^= e0 >> 7; e2 += e1; e3 ^= e2 << 3;
		}
		for (int i = 0; i < U; i++) cur[i] = nxt[i];
		bcur = bnxt;
	}
}

// the prefetched loop with the records staged in LDS and written out cooperatively: every 8 steps the wave's 8 x 64 records
// (12 KB, blocks of 8 columns per lane = the R = 8 layout) leave as twelve fully coalesced 1 KB stores
template <int U>
__global__ void __launch_bounds__(64) k_fill_staged(const uint32_t* __restrict__ seq2, size_t genomeWords, const uint64_t* __restrict__ eqTab,
                                              uint8_t* __restrict__ arena, size_t arenaBytesPerWave, uint32_t* __restrict__ endBuf, int steps, int colsPerSlice)
{
	__shared__ uint64_t stage[64 * 25];          // lane stride 200 B (25 x 8): conflict-free 8-byte writes
	const int lane = threadIdx.x;
	const size_t blk = blockIdx.x;
	uint8_t* my = arena + blk * arenaBytesPerWave;
	uint32_t* endPrev = endBuf + blk * 2 * (size_t)colsPerSlice * 64;
	uint32_t* endCur = endPrev + (size_t)colsPerSlice * 64;
	uint64_t e0 = eqTab[(blk * 4 + 0) * 64 + lane], e1 = eqTab[(blk * 4 + 1) * 64 + lane], e2 = eqTab[(blk * 4 + 2) * 64 + lane], e3 = eqTab[(blk * 4 + 3) * 64 + lane];
	uint64_t vp = ~0ull, vn = 0;
	int before = 1000 + lane;
	size_t gcol = ((size_t)(blk * 64 + lane) * 7919u * 64u) % (genomeWords * 16 - (size_t)steps - 64);
	uint32_t row = 0;
	int c = 0;
	uint32_t cur[U], nxt[U];
	uint64_t bcur, bnxt;
	for (int i = 0; i < U; i++) cur[i] = endPrev[(size_t)(c + i) * 64 + lane];
	bcur = (uint64_t)seq2[gcol >> 4] | ((uint64_t)seq2[(gcol >> 4) + 1] << 32);
	for (int k = 0; k < steps; k += U)
	{
		const bool last = c + U == colsPerSlice;
		const uint32_t* src = last ? endCur : endPrev;
		const int cn = last ? 0 : c + U;
		for (int i = 0; i < U; i++) nxt[i] = src[(size_t)(cn + i) * 64 + lane];
		bnxt = (uint64_t)seq2[(gcol + U) >> 4] | ((uint64_t)seq2[((gcol + U) >> 4) + 1] << 32);
#pragma unroll
		for (int i = 0; i < U; i++)
		{
			const int b = (int)(bcur >> (((gcol & 15) + i) * 2)) & 3;
			const uint32_t pe = cur[i];
			int calc = before + 1;
			const int above = (int)(pe >> 3) + before - 3;
			const bool re = calc > above && (pe & 4);
			const uint64_t eq0 = (b & 2) ? ((b & 1) ? e3 : e2) : ((b & 1) ? e1 : e0);
			bitvector_column_step(vp, vn, before, re ? above : calc, b | ((pe & 8) ? 4 : 0), e0, e1, e2, e3);
			(void)eq0;
			const int end = before + __builtin_popcountll(vp) - __builtin_popcountll(vn);
			const uint32_t ew = ((uint32_t)end << 3) | (uint32_t)(vp >> 63) | ((uint32_t)(vn >> 63) << 1);
			uint64_t* st = stage + lane * 25 + i * 3;
			st[0] = vp; st[1] = vn; st[2] = (uint64_t)(uint32_t)before | ((uint64_t)ew << 32);
			endCur[(size_t)(c + i) * 64 + lane] = ew;
		}
		// cooperative write of the 8-row block: chunk q (16 B) of the 12 KB image belongs to lane q / 12
		__builtin_amdgcn_wave_barrier();
		uint8_t* dst = my + (size_t)(row / 8) * (64 * 192);
#pragma unroll
		for (int j = 0; j < 12; j++)
		{
			const uint32_t q = (uint32_t)lane + 64u * j;
			const uint32_t ln = q / 12u, part = q % 12u;
			const uint64_t* sp = stage + ln * 25 + part * 2;
			const uint64_t a = sp[0], bq = sp[1];
			*(uint4*)(dst + (size_t)q * 16) = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)bq, (uint32_t)(bq >> 32));
		}
		__builtin_amdgcn_wave_barrier();
		row += U;
		gcol += U;
		c += U;
		if (c == colsPerSlice)
		{
			c = 0;
			uint32_t* t = endPrev; endPrev = endCur; endCur = t;
			e0 = e0 * 0x9E3779B97F4A7C15ull + 1; e1 ^= e0 >> 7; e2 += e1; e3 ^= e2 << 3;
		}
		for (int i = 0; i < U; i++) cur[i] = nxt[i];
		bcur = bnxt;
	}
}

// the staged loop with the end words in blocks of 8 columns per lane ([column / 8][lane][8 words]): a chunk reads its 8 previous end
// words with two 16-byte loads and writes its own with two 16-byte stores instead of 8 + 8 four-byte accesses
template <int U>
__global__ void __launch_bounds__(64) k_fill_staged_blocked(const uint32_t* __restrict__ seq2, size_t genomeWords, const uint64_t* __restrict__ eqTab,
                                              uint8_t* __restrict__ arena, size_t arenaBytesPerWave, uint32_t* __restrict__ endBuf, int steps, int colsPerSlice)
{
	static_assert(U == 8, "blocks of 8");
	__shared__ uint64_t stage[64 * 25];
	const int lane = threadIdx.x;
	const size_t blk = blockIdx.x;
	uint8_t* my = arena + blk * arenaBytesPerWave;
	uint32_t* endPrev = endBuf + blk * 2 * (size_t)colsPerSlice * 64;
	uint32_t* endCur = endPrev + (size_t)colsPerSlice * 64;
	uint64_t e0 = eqTab[(blk * 4 + 0) * 64 + lane], e1 = eqTab[(blk * 4 + 1) * 64 + lane], e2 = eqTab[(blk * 4 + 2) * 64 + lane], e3 = eqTab[(blk * 4 + 3) * 64 + lane];
	uint64_t vp = ~0ull, vn = 0;
	int before = 1000 + lane;
	size_t gcol = ((size_t)(blk * 64 + lane) * 7919u * 64u) % (genomeWords * 16 - (size_t)steps - 64);
	uint32_t row = 0;
	int c = 0;
	uint4 curA, curB, nxtA, nxtB;
	uint64_t bcur, bnxt;
	auto blockAt = [&](uint32_t* plane, int col) { return (uint4*)(plane + ((size_t)(col >> 3) * 64 + lane) * 8); };
	curA = blockAt(endPrev, c)[0]; curB = blockAt(endPrev, c)[1];
	bcur = (uint64_t)seq2[gcol >> 4] | ((uint64_t)seq2[(gcol >> 4) + 1] << 32);
	for (int k = 0; k < steps; k += U)
	{
		const bool last = c + U == colsPerSlice;
		uint32_t* src = last ? endCur : endPrev;
		const int cn = last ? 0 : c + U;
		nxtA = blockAt(src, cn)[0]; nxtB = blockAt(src, cn)[1];
		bnxt = (uint64_t)seq2[(gcol + U) >> 4] | ((uint64_t)seq2[((gcol + U) >> 4) + 1] << 32);
		const uint32_t cur[8] = {curA.x, curA.y, curA.z, curA.w, curB.x, curB.y, curB.z, curB.w};
		uint32_t ews[8];
#pragma unroll
		for (int i = 0; i < U; i++)
		{
			const int b = (int)(bcur >> (((gcol & 15) + i) * 2)) & 3;
			const uint32_t pe = cur[i];
			int calc = before + 1;
			const int above = (int)(pe >> 3) + before - 3;
			const bool re = calc > above && (pe & 4);
			bitvector_column_step(vp, vn, before, re ? above : calc, b | ((pe & 8) ? 4 : 0), e0, e1, e2, e3);
			const int end = before + __builtin_popcountll(vp) - __builtin_popcountll(vn);
			const uint32_t ew = ((uint32_t)end << 3) | (uint32_t)(vp >> 63) | ((uint32_t)(vn >> 63) << 1);
			uint64_t* st = stage + lane * 25 + i * 3;
			st[0] = vp; st[1] = vn; st[2] = (uint64_t)(uint32_t)before | ((uint64_t)ew << 32);
			ews[i] = ew;
		}
		blockAt(endCur, c)[0] = make_uint4(ews[0], ews[1], ews[2], ews[3]);
		blockAt(endCur, c)[1] = make_uint4(ews[4], ews[5], ews[6], ews[7]);
		__builtin_amdgcn_wave_barrier();
		uint8_t* dst = my + (size_t)(row / 8) * (64 * 192);
#pragma unroll
		for (int j = 0; j < 12; j++)
		{
			const uint32_t q = (uint32_t)lane + 64u * j;
			const uint32_t ln = q / 12u, part = q % 12u;
			const uint64_t* sp = stage + ln * 25 + part * 2;
			const uint64_t a = sp[0], bq = sp[1];
			*(uint4*)(dst + (size_t)q * 16) = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)bq, (uint32_t)(bq >> 32));
		}
		__builtin_amdgcn_wave_barrier();
		row += U;
		gcol += U;
		c += U;
		if (c == colsPerSlice)
		{
			c = 0;
			uint32_t* t = endPrev; endPrev = endCur; endCur = t;
			e0 = e0 * 0x9E3779B97F4A7C15ull + 1; e1 ^= e0 >> 7; e2 += e1; e3 ^= e2 << 3;
		}
		curA = nxtA; curB = nxtB;
		bcur = bnxt;
	}
}

// traceback shape: one dependent record fetch per step (the next row depends on the fetched words)
template <int R>
__global__ void __launch_bounds__(64) k_trace(const uint8_t* __restrict__ arena, size_t arenaBytesPerWave, uint32_t* __restrict__ out, int steps, uint32_t rows)
{
	const int lane = threadIdx.x;
	const size_t blk = blockIdx.x;
	const uint8_t* my = arena + blk * arenaBytesPerWave;
	uint32_t row = rows - 1 - (uint32_t)lane * 3;
	uint32_t acc = 0;
	for (int k = 0; k < steps; k++)
	{
		const uint8_t* rec = my + rec_off<R>(row, lane);
		const uint4 a = *(const uint4*)rec;
		const uint2 b = *(const uint2*)(rec + 16);
		const uint64_t vp = ((uint64_t)a.y << 32) | a.x, vn = ((uint64_t)a.w << 32) | a.z;
		const int v = (int)b.x + __builtin_popcountll(vp & 0xffffffffull) - __builtin_popcountll(vn & 0xffffull);
		acc += (uint32_t)v;
		row = row - 1 - ((uint32_t)v & 0u);       // data-dependent (always one column to the left)
		if (row > rows) row = rows - 1;
	}
	out[blk * 64 + lane] = acc;
}

template <int R, int LANES> int run(int cus, const uint32_t* seq2, size_t genomeWords, int nReads, int steps, int colsPerSlice)
{
	const int waves = (nReads + LANES - 1) / LANES;
	const size_t arenaBytesPerWave = ((size_t)steps + 8) * 64 * 24;
	uint8_t* arena; uint64_t* eqTab; uint32_t* endBuf; uint32_t* out;
	OK(hipMalloc((void**)&arena, arenaBytesPerWave * waves));
	OK(hipMalloc((void**)&eqTab, (size_t)waves * 4 * 64 * 8));
	OK(hipMalloc((void**)&endBuf, (size_t)waves * 2 * colsPerSlice * 64 * 4));
	OK(hipMalloc((void**)&out, (size_t)waves * 64 * 4));
	OK(hipMemset(endBuf, 0x11, (size_t)waves * 2 * colsPerSlice * 64 * 4));
	std::vector<uint64_t> he((size_t)waves * 4 * 64);
	for (auto& x : he) x = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ (uint64_t)rand();
	OK(hipMemcpy(eqTab, he.data(), he.size() * 8, hipMemcpyHostToDevice));
	hipEvent_t a, b;
	OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
	float best = 1e30f, bestT = 1e30f;
	for (int rep = 0; rep < 3; rep++)
	{
		OK(hipEventRecord(a, 0));
		hipLaunchKernelGGL((k_fill<R, LANES>), dim3(waves), dim3(64), 0, 0, seq2, genomeWords, eqTab, arena, arenaBytesPerWave, endBuf, steps, colsPerSlice);
		OK(hipEventRecord(b, 0));
		OK(hipEventSynchronize(b));
		float ms = 0; OK(hipEventElapsedTime(&ms, a, b));
		best = ms < best ? ms : best;
	}
	float bestP = 1e30f;
	for (int rep = 0; rep < 3; rep++)
	{
		OK(hipEventRecord(a, 0));
		hipLaunchKernelGGL((k_fill_pf<R, LANES, 8>), dim3(waves), dim3(64), 0, 0, seq2, genomeWords, eqTab, arena, arenaBytesPerWave, endBuf, steps, colsPerSlice);
		OK(hipEventRecord(b, 0));
		OK(hipEventSynchronize(b));
		float ms = 0; OK(hipEventElapsedTime(&ms, a, b));
		bestP = ms < bestP ? ms : bestP;
	}
	printf("R=%d lanes/wave=%2d waves=%4d: PREFETCHED fill %8.3f ms = %6.1f G/s, %6.1f GB/s at 28 B (%.1f %% of 8 TB/s), %.0f cycles per wave step at 2.4 GHz\n",
	       R, LANES, waves, bestP, (double)nReads * steps / bestP / 1e6, (double)nReads * steps * 28 / bestP / 1e6, (double)nReads * steps * 28 / bestP / 1e6 / 80.0, bestP * 1e-3 * 2.4e9 / steps);
	if (LANES == 64 && R == 8)
	{
		float bestS = 1e30f;
		for (int rep = 0; rep < 3; rep++)
		{
			OK(hipEventRecord(a, 0));
			hipLaunchKernelGGL((k_fill_staged<8>), dim3(waves), dim3(64), 0, 0, seq2, genomeWords, eqTab, arena, arenaBytesPerWave, endBuf, steps, colsPerSlice);
			OK(hipEventRecord(b, 0));
			OK(hipEventSynchronize(b));
			float ms = 0; OK(hipEventElapsedTime(&ms, a, b));
			bestS = ms < bestS ? ms : bestS;
		}
		printf("R=8 via LDS staging, coalesced block writes, unrolled steps: %8.3f ms = %.0f cycles per wave step at 2.4 GHz\n", bestS, bestS * 1e-3 * 2.4e9 / steps);
		float bestB = 1e30f;
		for (int rep = 0; rep < 3; rep++)
		{
			OK(hipEventRecord(a, 0));
			hipLaunchKernelGGL((k_fill_staged_blocked<8>), dim3(waves), dim3(64), 0, 0, seq2, genomeWords, eqTab, arena, arenaBytesPerWave, endBuf, steps, colsPerSlice);
			OK(hipEventRecord(b, 0));
			OK(hipEventSynchronize(b));
			float ms = 0; OK(hipEventElapsedTime(&ms, a, b));
			bestB = ms < bestB ? ms : bestB;
		}
		printf("  ... and the end words in blocks of 8 columns per lane (two 16-byte accesses per chunk each way): %8.3f ms = %.0f cycles per wave step at 2.4 GHz\n", bestB, bestB * 1e-3 * 2.4e9 / steps);
	}
	if (LANES == 64)
		for (int rep = 0; rep < 2; rep++)
		{
			OK(hipEventRecord(a, 0));
			hipLaunchKernelGGL((k_trace<R>), dim3(waves), dim3(64), 0, 0, arena, arenaBytesPerWave, out, steps / 3, (uint32_t)steps);
			OK(hipEventRecord(b, 0));
			OK(hipEventSynchronize(b));
			float ms = 0; OK(hipEventElapsedTime(&ms, a, b));
			bestT = ms < bestT ? ms : bestT;
		}
	const double updates = (double)nReads * steps;
	printf("R=%d lanes/wave=%2d waves=%4d: fill %8.3f ms for %.3g column updates = %6.1f G/s, %6.1f GB/s at 28 B (%.1f %% of 8 TB/s), %.0f cycles per wave step at 2.4 GHz",
	       R, LANES, waves, best, updates, updates / best / 1e6, updates * 28 / best / 1e6, updates * 28 / best / 1e6 / 80.0, best * 1e-3 * 2.4e9 / steps);
	if (LANES == 64) printf(" | trace-shaped %8.3f ms for %d dependent steps = %.0f cycles per step", bestT, steps / 3, bestT * 1e-3 * 2.4e9 / (steps / 3));
	printf("\n");
	fflush(stdout);
	hipFree(arena); hipFree(eqTab); hipFree(endBuf); hipFree(out);
	return 0;
}

int main(int argc, char** argv)
{
	hipDeviceProp_t p;
	OK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	const int nReads = argc > 1 ? atoi(argv[1]) : 50000;
	const int steps = argc > 2 ? atoi(argv[2]) : 8192;          // a 10 kb read has 32 400; the cost per step is what is measured
	const int colsPerSlice = 208;
	const size_t genomeWords = 9283306 / 16 + 64;
	uint32_t* seq2;
	OK(hipMalloc((void**)&seq2, genomeWords * 4));
	std::vector<uint32_t> hs(genomeWords);
	for (auto& x : hs) x = (uint32_t)rand() * 2654435761u;
	OK(hipMemcpy(seq2, hs.data(), genomeWords * 4, hipMemcpyHostToDevice));
	printf("%d CUs; %d reads, %d steps each\n", cus, nReads, steps);
	if (run<1, 64>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	if (run<2, 64>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	if (run<4, 64>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	if (run<8, 64>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	if (run<2, 32>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	if (run<1, 32>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	if (run<4, 32>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	if (run<8, 32>(cus, seq2, genomeWords, nReads, steps, colsPerSlice)) return 1;
	return 0;
}
