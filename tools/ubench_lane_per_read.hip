// ubench_lane_per_read.hip -- what the column recurrence costs in the OTHER mapping: one read per lane.
// Every lane owns a read's 64-row column state as Myers words (VP, VN: 64-bit per lane) and steps it with the
// reference's bit-vector recurrence (GraphAligner.h:1349-1427 without row confirmation); the waves of a block
// advance in lockstep over band columns, and the words leave interleaved [column][lane] so that every store
// is one coalesced 512-byte row.  This is only the inner loop (no band selection, no node starts, no
// traceback): it bounds what DESIGN.md section 9's "lanes = reads" design could reach, against the same
// 28 B-per-column-update roofline.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lane_per_read tools/ubench_lane_per_read.hip && tools/ubench_lane_per_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) k_columns(const uint32_t* __restrict__ bases,   // [columns][64 lanes] 2-bit base per (column, read), packed 16 per word along columns
                                                 const uint64_t* __restrict__ eqTab,  // [block][4][64 lanes] match words of the lane's 64 read rows
                                                 const uint32_t* __restrict__ endPrev, // [block][columns][64] previous slice's end word (read)
                                                 uint64_t* __restrict__ vpOut, uint64_t* __restrict__ vnOut, uint32_t* __restrict__ endOut, int columns, int store)
{
	const int lane = threadIdx.x;
	const size_t blk = blockIdx.x;
	const uint64_t e0 = eqTab[(blk * 4 + 0) * 64 + lane], e1 = eqTab[(blk * 4 + 1) * 64 + lane], e2 = eqTab[(blk * 4 + 2) * 64 + lane], e3 = eqTab[(blk * 4 + 3) * 64 + lane];
	uint64_t vp = ~0ull, vn = 0;
	int before = lane, end = lane + 64;
	const size_t plane = (size_t)columns * 64;
	for (int c = 0; c < columns; c++)
	{
		const uint32_t w = bases[(size_t)(c >> 4) * 64 + lane];
		const int b = (w >> ((c & 15) * 2)) & 3;
		const uint32_t pe = endPrev[blk * plane + (size_t)c * 64 + lane];
		uint64_t eq = (b & 2) ? ((b & 1) ? e3 : e2) : ((b & 1) ? e1 : e0);
		// scoreBeforeStart of this column: from the left, or re-entered from the previous slice's end score
		int calc = before + 1;
		const int above = (int)(pe >> 2);
		calc = calc < above ? calc : above;
		const int hin = calc - before;
		const uint64_t neg = (uint32_t)hin >> 31, pos = (uint32_t)(-hin) >> 31;
		const uint64_t xv = eq | vn;
		eq |= neg;
		const uint64_t xh = (((eq & vp) + vp) ^ vp) | eq;
		uint64_t ph = vn | ~(xh | vp);
		uint64_t mh = vp & xh;
		end += (int)(ph >> 63) - (int)(mh >> 63);
		ph = (ph << 1) | pos;
		mh = (mh << 1) | neg;
		vp = mh | ~(xv | ph);
		vn = ph & xv;
		before = calc;
		if (store)
		{
			vpOut[blk * plane + (size_t)c * 64 + lane] = vp;
			vnOut[blk * plane + (size_t)c * 64 + lane] = vn;
			endOut[blk * plane + (size_t)c * 64 + lane] = ((uint32_t)end << 2) | (uint32_t)(vp >> 63) | ((uint32_t)(vn >> 63) << 1);
		}
	}
	if (!store) { vpOut[blk * 64 + lane] = vp; vnOut[blk * 64 + lane] = vn; endOut[blk * 64 + lane] = (uint32_t)end + (uint32_t)before; }
}

#include "bitvector_column_step.h"
// the same with what a real column program carries: one program word per (step, lane) -- base, no-diagonal flag, hin, re-entry depth --
// and the full step of tools/bitvector_column_step.h (re-entry on about a third of the columns, as in the aligner)
__global__ void __launch_bounds__(64) k_program(const uint32_t* __restrict__ prog,    // [columns][64 lanes]
                                                const uint64_t* __restrict__ eqTab, const uint32_t* __restrict__ endPrev,
                                                uint64_t* __restrict__ vpOut, uint64_t* __restrict__ vnOut, uint32_t* __restrict__ endOut, int columns)
{
	const int lane = threadIdx.x;
	const size_t blk = blockIdx.x;
	const uint64_t e0 = eqTab[(blk * 4 + 0) * 64 + lane], e1 = eqTab[(blk * 4 + 1) * 64 + lane], e2 = eqTab[(blk * 4 + 2) * 64 + lane], e3 = eqTab[(blk * 4 + 3) * 64 + lane];
	uint64_t vp = ~0ull, vn = 0;
	int before = 1000 + lane;
	const size_t plane = (size_t)columns * 64;
	for (int c = 0; c < columns; c++)
	{
		const uint32_t w = prog[(size_t)c * 64 + lane];
		const uint32_t pe = endPrev[blk * plane + (size_t)c * 64 + lane];
		const int calc = before + (int)((w >> 8) & 3) - 1 + (int)(pe & 0);          // hin from the program (the previous end word is loaded as in the real thing)
		bitvector_column_step(vp, vn, before, calc, (int)(w & 31), e0, e1, e2, e3);
		const int end = before + __builtin_popcountll(vp) - __builtin_popcountll(vn);
		vpOut[blk * plane + (size_t)c * 64 + lane] = vp;
		vnOut[blk * plane + (size_t)c * 64 + lane] = vn;
		endOut[blk * plane + (size_t)c * 64 + lane] = ((uint32_t)end << 2) | (uint32_t)(vp >> 63) | ((uint32_t)(vn >> 63) << 1);
	}
}

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
	hipDeviceProp_t p;
	OK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	const int columns = 4096;                        // band columns of ~16 slices per launch and block
	for (int wavesPerSimd : {2, 4, 8})
	{
		const int blocks = cus * 4 * wavesPerSimd;
		const size_t plane = (size_t)columns * 64;
		uint32_t *bases, *endPrev, *endOut;
		uint64_t *eqTab, *vpOut, *vnOut;
		OK(hipMalloc((void**)&bases, (size_t)(columns / 16) * 64 * 4));
		OK(hipMalloc((void**)&eqTab, (size_t)blocks * 4 * 64 * 8));
		OK(hipMalloc((void**)&endPrev, (size_t)blocks * plane * 4));
		OK(hipMalloc((void**)&endOut, (size_t)blocks * plane * 4));
		OK(hipMalloc((void**)&vpOut, (size_t)blocks * plane * 8));
		OK(hipMalloc((void**)&vnOut, (size_t)blocks * plane * 8));
		std::vector<uint32_t> hb((size_t)(columns / 16) * 64);
		for (auto& x : hb) x = (uint32_t)rand() * 2654435761u;
		OK(hipMemcpy(bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
		std::vector<uint64_t> he((size_t)blocks * 4 * 64);
		for (auto& x : he) x = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ (uint64_t)rand();
		OK(hipMemcpy(eqTab, he.data(), he.size() * 8, hipMemcpyHostToDevice));
		OK(hipMemset(endPrev, 0x3f, (size_t)blocks * plane * 4));
		hipEvent_t e0, e1;
		OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
		for (int store = 1; store >= 0; store--)
		{
			float best = 1e30f;
			for (int rep = 0; rep < 3; rep++)
			{
				OK(hipEventRecord(e0, 0));
				hipLaunchKernelGGL(k_columns, dim3(blocks), dim3(64), 0, 0, bases, eqTab, endPrev, vpOut, vnOut, endOut, columns, store);
				OK(hipEventRecord(e1, 0));
				OK(hipEventSynchronize(e1));
				float ms = 0;
				OK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best;
			}
			const double updates = (double)blocks * 64 * columns;
			const double rate = updates / (best * 1e-3);
			printf("%d waves/SIMD, %s: %7.3f ms for %.3g column updates -> %6.1f G column updates/s, %7.1f GB/s at 28 B each (%.1f %% of 8 TB/s), %.2f SIMD-cycles per column update\n",
			       wavesPerSimd, store ? "words stored" : "compute only", best, updates, rate / 1e9, rate * 28 / 1e9, rate * 28 / 8e12 * 100, (double)cus * 4 * p.clockRate * 1e3 / rate);
			fflush(stdout);
		}
		{
			// program words: base 2 bits, no-diagonal 1 bit (1 in 8), re-entry depth d in bits 3-4 (d = 1 or 2 on about a third of the columns), hin + 1 in bits 8-9
			uint32_t* prog;
			OK(hipMalloc((void**)&prog, plane * 4));
			std::vector<uint32_t> hp(plane);
			for (auto& x : hp)
			{
				const uint32_t r = (uint32_t)rand();
				const uint32_t d = (r >> 8) % 3 == 0 ? 1 + ((r >> 12) & 1) : 0;
				x = (r & 3) | (((r >> 2) & 7) == 0 ? 4u : 0u) | (d << 3) | ((1 + (d ? 0 : ((r >> 16) % 3) - 1 + 0)) << 8);
			}
			OK(hipMemcpy(prog, hp.data(), plane * 4, hipMemcpyHostToDevice));
			float best = 1e30f;
			for (int rep = 0; rep < 3; rep++)
			{
				OK(hipEventRecord(e0, 0));
				hipLaunchKernelGGL(k_program, dim3(blocks), dim3(64), 0, 0, prog, eqTab, endPrev, vpOut, vnOut, endOut, columns);
				OK(hipEventRecord(e1, 0));
				OK(hipEventSynchronize(e1));
				float ms = 0;
				OK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best;
			}
			const double updates = (double)blocks * 64 * columns;
			const double rate = updates / (best * 1e-3);
			printf("%d waves/SIMD, program words + re-entry, words stored: %7.3f ms -> %6.1f G column updates/s, %7.1f GB/s at 28 B each (%.1f %% of 8 TB/s), %.2f SIMD-cycles per column update\n",
			       wavesPerSimd, best, rate / 1e9, rate * 28 / 1e9, rate * 28 / 8e12 * 100, (double)cus * 4 * p.clockRate * 1e3 / rate);
			fflush(stdout);
			hipFree(prog);
		}
		hipFree(bases); hipFree(eqTab); hipFree(endPrev); hipFree(endOut); hipFree(vpOut); hipFree(vnOut);
	}
	return 0;
}
