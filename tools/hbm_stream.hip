// hbm_stream.hip -- achievable HBM bandwidth on this box: a plain streaming copy (16 B per lane,
// grid-stride), timed with HIP events.  SURVEY 8(d) asks for roofline fractions against both the
// vendor peak (8 TB/s) and this measured figure.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_stream tools/hbm_stream.hip && tools/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) read_kernel(const uint4* __restrict__ src, uint4* __restrict__ sink, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	uint4 acc = make_uint4(0, 0, 0, 0);
	for (; i < n; i += stride) { uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
	if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;      // never true in practice; keeps the loads alive
}

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
	const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 4096) << 20;
	const size_t n = bytes / 16;
	uint4 *a, *b;
	OK(hipMalloc((void**)&a, bytes));
	OK(hipMalloc((void**)&b, bytes));
	OK(hipMemset(a, 1, bytes));
	OK(hipMemset(b, 2, bytes));
	hipEvent_t e0, e1;
	OK(hipEventCreate(&e0));
	OK(hipEventCreate(&e1));
	for (int blocksPerCu : {4, 8, 16, 32})
	{
		const int grid = 256 * blocksPerCu;
		for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
		OK(hipEventRecord(e0, 0));
		const int iters = 10;
		for (int rep = 0; rep < iters; rep++) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
		OK(hipEventRecord(e1, 0));
		OK(hipEventSynchronize(e1));
		float ms = 0;
		OK(hipEventElapsedTime(&ms, e0, e1));
		printf("copy  %zu MiB grid %5d x256: %8.1f GB/s (read + write)\n", bytes >> 20, grid, 2.0 * bytes * iters / (ms * 1e-3) / 1e9);
		OK(hipEventRecord(e0, 0));
		for (int rep = 0; rep < iters; rep++) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
		OK(hipEventRecord(e1, 0));
		OK(hipEventSynchronize(e1));
		OK(hipEventElapsedTime(&ms, e0, e1));
		printf("read  %zu MiB grid %5d x256: %8.1f GB/s\n", bytes >> 20, grid, 1.0 * bytes * iters / (ms * 1e-3) / 1e9);
		fflush(stdout);
	}
	return 0;
}
