// ubench_valu.hip -- per-instruction issue cost of the handful of instructions the column loop is
// made of, on gfx950, one wave per SIMD and eight waves per SIMD.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND> __global__ void k(int* out, int iters, long long* cyc)
{
	int a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7;
	int s1 = iters & 63;
	long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++)
	{
		if (KIND == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
		if (KIND == 1) { REP64(asm volatile("v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a));) }
		if (KIND == 2) { REP64(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));) }
		if (KIND == 3) { REP64(asm volatile("v_mov_b32_dpp %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(a) : "v"(b));) }
		if (KIND == 4) { REP64(asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(c) : "v"(a), "s"(s1));) }
		if (KIND == 5) { REP64(asm volatile("s_mov_b32 m0, %2\n v_writelane_b32 %0, %1, m0" : "+v"(a) : "s"(c), "s"(s1));) }
		if (KIND == 6) { REP64(asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");) }
		if (KIND == 7) { REP64(asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(a) : "v"(b), "s"(s1));) }
		if (KIND == 8) { REP64(asm volatile("s_add_i32 %0, %0, 1" : "+s"(c));) }
		if (KIND == 9) { REP64(asm volatile("s_nop 1");) }
		if (KIND == 10) { REP64(asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(c));) }
		if (KIND == 11) { REP64(asm volatile("v_add_u32 %0, %0, %1\n s_add_i32 %2, %2, 1" : "+v"(a), "+v"(b), "+s"(c));) }
	}
	long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(const char* name, int wavesPerSimd)
{
	int blocks = 256 * 4 * wavesPerSimd, iters = 200;
	int* out; long long* cyc;
	hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
	k<KIND><<<blocks, 64>>>(out, iters, cyc);
	hipDeviceSynchronize();
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a);
	k<KIND><<<blocks, 64>>>(out, iters, cyc);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	std::vector<long long> h(blocks);
	hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
	double avg = 0; for (auto v : h) avg += v; avg /= blocks;
	double perInstrWave = avg / (iters * 64.0);
	printf("%-28s waves/SIMD %d: %.2f cycles per instr per wave, %.2f cycles per instr per SIMD, kernel %.3f ms\n", name, wavesPerSimd, perInstrWave, perInstrWave / wavesPerSimd, ms);
	hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv)
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	int only = argc > 1 ? atoi(argv[1]) : -1;
	for (int w : {1, 2, 4, 8})
	{
		if (only > 0 && w != only) continue;
		run<0>("v_add_u32", w);
		run<1>("v_min_i32_dpp row_shr+nop1", w);
		run<2>("v_mov_dpp wave_shr", w);
		run<3>("v_mov_dpp row_bcast31", w);
		run<4>("v_readlane (sgpr idx)", w);
		run<5>("s_mov m0 + v_writelane", w);
		run<6>("v_cmp_eq -> vcc", w);
		run<7>("v_bfe_u32 (sgpr off)", w);
		run<10>("v_min3_i32", w);
		printf("\n");
	}
	return 0;
}
