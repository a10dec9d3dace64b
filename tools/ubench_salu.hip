// ubench_salu.hip -- how fast does a SIMD retire scalar-ALU work next to vector-ALU work?
// Same 64-bit logic chain (and/add/xor/or/shift/nor: the shape of a Myers step) once on wave-uniform
// values (compiles to s_* instructions) and once on per-lane values (v_*), at 1..8 waves per SIMD, and
// both kinds of waves mixed on the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_salu tools/ubench_salu.hip && tools/ubench_salu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define STEP(x, y, z, w) { x = (x & y) + z; y = (y ^ x) | w; z = (z << 1) | (x >> 63); w = ~(w | y); }

__global__ void __launch_bounds__(64) k_scalar(uint64_t a, uint64_t b, int iters, uint64_t* out)
{
	uint64_t x = a, y = b, z = a ^ b, w = a + 3;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int k = 0; k < 16; k++) STEP(x, y, z, w)
	}
	if (threadIdx.x == 0) out[blockIdx.x] = x ^ y ^ z ^ w;
}
__global__ void __launch_bounds__(64) k_vector(uint64_t a, uint64_t b, int iters, uint64_t* out)
{
	uint64_t x = a + threadIdx.x, y = b ^ threadIdx.x, z = a ^ b, w = a + 3 * threadIdx.x;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int k = 0; k < 16; k++) STEP(x, y, z, w)
	}
	out[blockIdx.x * 64 + threadIdx.x] = x ^ y ^ z ^ w;
}
// alternate layers of one-wave-per-SIMD blocks scalar / vector: both kinds resident on every SIMD
__global__ void __launch_bounds__(64) k_mixed(uint64_t a, uint64_t b, int iters, uint64_t* out, int layer)
{
	if ((blockIdx.x / layer) & 1)
	{
		uint64_t x = a + threadIdx.x, y = b ^ threadIdx.x, z = a ^ b, w = a + 3 * threadIdx.x;
		for (int i = 0; i < iters; i++)
		{
#pragma unroll
			for (int k = 0; k < 16; k++) STEP(x, y, z, w)
		}
		out[blockIdx.x * 64 + threadIdx.x] = x ^ y ^ z ^ w;
	}
	else
	{
		uint64_t x = a, y = b, z = a ^ b, w = a + 3;
		for (int i = 0; i < iters; i++)
		{
#pragma unroll
			for (int k = 0; k < 16; k++) STEP(x, y, z, w)
		}
		if (threadIdx.x == 0) out[blockIdx.x * 64] = x ^ y ^ z ^ w;
	}
}

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
	hipDeviceProp_t p;
	OK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	const double mhz = p.clockRate / 1000.0;
	uint64_t* out;
	OK(hipMalloc((void**)&out, (size_t)cus * 64 * 64 * 8));
	hipEvent_t e0, e1;
	OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
	const int iters = 20000;
	printf("CUs %d clock %.0f MHz; one STEP = 4 chain statements on 64-bit values, 16 STEPs per loop trip, %d trips\n", cus, mhz, iters);
	for (int kind = 0; kind < 3; kind++)
		for (int wavesPerSimd : {1, 2, 4, 8})
		{
			const int grid = cus * 4 * wavesPerSimd;
			for (int rep = 0; rep < 2; rep++)
			{
				OK(hipEventRecord(e0, 0));
				if (kind == 0) hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(64), 0, 0, 0x1234567ull, 0x9876543ull, iters, out);
				else if (kind == 1) hipLaunchKernelGGL(k_vector, dim3(grid), dim3(64), 0, 0, 0x1234567ull, 0x9876543ull, iters, out);
				else hipLaunchKernelGGL(k_mixed, dim3(grid), dim3(64), 0, 0, 0x1234567ull, 0x9876543ull, iters, out, cus * 4);
				OK(hipEventRecord(e1, 0));
				OK(hipEventSynchronize(e1));
			}
			float ms = 0;
			OK(hipEventElapsedTime(&ms, e0, e1));
			const double steps = (double)iters * 16;
			const double cyc = ms * 1e-3 * mhz * 1e6;
			printf("%-7s %d waves/SIMD: %8.3f ms  -> %6.2f clock cycles per STEP per wave, %6.2f per STEP per SIMD\n", kind == 0 ? "scalar" : kind == 1 ? "vector" : "mixed",
			       wavesPerSimd, ms, cyc / steps, cyc / steps / wavesPerSimd);
			fflush(stdout);
		}
	return 0;
}
