// ubench_trace_step.hip -- what one fast traceback step of the lanes = reads kernel costs a lone wave (one wave per SIMD, 782 waves,
// every lane walking a diagonal through a synthetic window in LDS), with parts of the step switched off one at a time.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_trace_step.hip -o tools/ubench_trace_step && tools/ubench_trace_step
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Col { uint64_t vp, vn; int before; };
constexpr int kWin = 22, kWords = 150;

// V bits: 1 = no move store, 2 = no window read (the column is reused), 4 = no match-word select, 8 = masks by arithmetic (v_bfi) instead of v_cndmask
template <int V> __global__ void __launch_bounds__(64) k(uint32_t* out, int reps, uint32_t* moves, long long* cyc)
{
	__shared__ uint32_t lds[kWords * 64];
	const int lane = threadIdx.x;
	uint32_t* base = lds + lane;
	const int K = 40;
	// column c of the window: D[r][c] = |r - c - K|
	for (int c = 0; c < kWin; c++)
	{
		const int d = c + K;                                  // the row where the column's value is 0
		const uint64_t vn = d >= 63 ? ~0ull : ((1ull << (d + 1)) - 1);   // rows 0 .. d: -1
		const uint64_t vp = ~vn;
		base[(40 + c * 5 + 0) * 64] = (uint32_t)vp; base[(40 + c * 5 + 1) * 64] = (uint32_t)(vp >> 32);
		base[(40 + c * 5 + 2) * 64] = (uint32_t)vn; base[(40 + c * 5 + 3) * 64] = (uint32_t)(vn >> 32);
		base[(40 + c * 5 + 4) * 64] = (uint32_t)(d + 1);
	}
	__syncthreads();
	auto winRead = [&](uint32_t idx, Col& c) {
		const int at = 40 + (int)idx * 5;
		c.vp = ((uint64_t)base[(at + 1) * 64] << 32) | base[at * 64];
		c.vn = ((uint64_t)base[(at + 3) * 64] << 32) | base[(at + 2) * 64];
		c.before = (int)base[(at + 4) * 64];
	};
	uint64_t e[4] = {~0ull, ~0ull ^ (uint64_t)(reps == -1), ~0ull ^ (uint64_t)(reps == -2), ~0ull ^ (uint64_t)(reps == -3)};
	const uint64_t wbases = 0x1b1b1b1b1b1b1b1bull + (uint64_t)lane;
	uint32_t* mv = moves + (size_t)blockIdx.x * 64 * 4096 + lane;
	const uint32_t wLo = 0, wHi = kWin - 1, cap = 1u << 30;
	uint32_t len = 0, pack = 0, total = 0;
	bool tracing = true;
	int status = 0;
	long long acc = 0;
	int hereSum = 0;
	for (int rep = 0; rep < reps; rep++)
	{
		uint32_t offset = wHi, row = offset + K;
		Col q0, q1, q2;
		winRead(offset, q0); winRead(offset - 1, q1);
		int r = (int)row;
		uint64_t mR = r < 63 ? ~(~0ull << (r + 1)) : ~0ull, mU = ~(~0ull << r);
		int here = q0.before + __builtin_popcountll(q0.vp & mR) - __builtin_popcountll(q0.vn & mR);
		winRead((((offset >= wLo + 2) & (offset <= wHi))) ? offset - 2 : wLo, q2);
		const long long t0 = __builtin_readcyclecounter();
		while (true)
		{
			const bool fast = tracing & (r > 0) & (offset > wLo) & (offset <= wHi) & (len + 8 < cap);
			if (!__ballot(fast)) break;
			const int horizontal = q1.before + __builtin_popcountll(q1.vp & mR) - __builtin_popcountll(q1.vn & mR);
			const int diagonal = q1.before + __builtin_popcountll(q1.vp & mU) - __builtin_popcountll(q1.vn & mU);
			const int up = q0.before + __builtin_popcountll(q0.vp & mU) - __builtin_popcountll(q0.vn & mU);
			int want;
			if (V & 4) want = here;
			else
			{
				const int b = (int)(wbases >> (2 * ((offset - wLo) & 31))) & 3;
				want = here - 1 + (int)((e[b] >> (r & 63)) & 1);
			}
			const bool left = horizontal == here - 1;
			const bool diag = !left & (diagonal == want);
			const bool bad = (horizontal < here - 1) | (!left & (diagonal < want)) | (!left & !diag & (up != here - 1));
			const bool ok = fast & !bad;
			status = fast & bad ? 1 : status;
			tracing = tracing & !(fast & bad);
			const bool colMove = ok & (left | diag), rowMove = ok & !left;
			here = ok ? (left ? horizontal : diag ? diagonal : up) : here;
			if (V & 8)
			{
				const uint32_t rm = 0u - (uint32_t)rowMove, cm = 0u - (uint32_t)colMove;
				row += rm; r += (int)rm;
				const uint64_t rm64 = ((uint64_t)rm << 32) | rm, cm64 = ((uint64_t)cm << 32) | cm;
				mR = (mU & rm64) | (mR & ~rm64);
				mU = ((mU >> 1) & rm64) | (mU & ~rm64);
				offset += cm;
				q0.vp = (q1.vp & cm64) | (q0.vp & ~cm64); q0.vn = (q1.vn & cm64) | (q0.vn & ~cm64); q0.before = (int)(((uint32_t)q1.before & cm) | ((uint32_t)q0.before & ~cm));
				q1.vp = (q2.vp & cm64) | (q1.vp & ~cm64); q1.vn = (q2.vn & cm64) | (q1.vn & ~cm64); q1.before = (int)(((uint32_t)q2.before & cm) | ((uint32_t)q1.before & ~cm));
			}
			else
			{
				row -= rowMove ? 1u : 0u;
				r -= rowMove ? 1 : 0;
				mR = rowMove ? mU : mR;
				mU = rowMove ? mU >> 1 : mU;
				offset -= colMove ? 1u : 0u;
				q0.vp = colMove ? q1.vp : q0.vp; q0.vn = colMove ? q1.vn : q0.vn; q0.before = colMove ? q1.before : q0.before;
				q1.vp = colMove ? q2.vp : q1.vp; q1.vn = colMove ? q2.vn : q1.vn; q1.before = colMove ? q2.before : q1.before;
			}
			if (!(V & 2)) winRead((((offset >= wLo + 2) & (offset <= wHi))) ? offset - 2 : wLo, q2);
			if (ok)
			{
				const uint32_t code = left ? 1 : diag ? 2 : 3;
				pack |= code << (8 * (len & 3));
				if ((len & 3) == 3) { if (!(V & 1)) mv[(size_t)((len >> 2) & 4095) * 64] = pack; pack = 0; }
				len++;
			}
			total++;
		}
		acc += __builtin_readcyclecounter() - t0;
		hereSum += here;
	}
	out[blockIdx.x * 64 + lane] = len + pack + (uint32_t)status + total + (uint32_t)hereSum;
	if (lane == 0) { cyc[blockIdx.x * 2] = acc; cyc[blockIdx.x * 2 + 1] = total; }
}

template <int V> void run(const char* name)
{
	const int blocks = 782, reps = 2000;
	uint32_t* out; uint32_t* moves; long long* cyc;
	hipMalloc(&out, blocks * 64 * 4); hipMalloc(&moves, (size_t)blocks * 64 * 4096 * 4); hipMalloc(&cyc, blocks * 16);
	k<V><<<blocks, 64>>>(out, 10, moves, cyc);
	hipDeviceSynchronize();
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a);
	k<V><<<blocks, 64>>>(out, reps, moves, cyc);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	std::vector<long long> h(blocks * 2);
	hipMemcpy(h.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
	double units = 0, iters = 0; for (int i = 0; i < blocks; i++) { units += h[2 * i]; iters += h[2 * i + 1]; }
	printf("%-44s %.0f iterations per wave, %.1f counter units per iteration, kernel %.3f ms = %.1f ns per iteration (all of the kernel)\n", name, iters / blocks, units / iters, ms, ms * 1e6 / (iters / blocks));
	hipFree(out); hipFree(moves); hipFree(cyc);
}

int main()
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	run<0>("the step as in the kernel");
	run<1>("no move store");
	run<2>("no window read");
	run<3>("no store, no window read");
	run<4>("no match-word select");
	run<7>("no store, no read, no select");
	run<8>("selects by arithmetic masks");
	run<15>("arithmetic masks, nothing else");
	return 0;
}
