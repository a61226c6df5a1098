// Probe: what does the chip sustain for plain streaming WRITES (the analysis kernel's output: 12.6 GB of spectra per step), for
// reads, and for a copy?  One launch of 2048 workgroups x 256 threads per case, 16-byte accesses, 4 GiB per array.
// Build: hipcc --offload-arch=gfx950 -O2 -o hbm_write_rate hbm_write_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void writeK(float4 *dst, size_t n, float v) {
	for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) dst[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void write8K(float2 *dst, size_t n, float v) { // 8-byte stores at a 16-byte stride, two passes: the analysis pattern
	const size_t half = n/2;
	for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x*blockDim.x) { dst[2*i] = make_float2(v, v); }
	for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x*blockDim.x) { dst[2*i + 1] = make_float2(v, -v); }
}
__global__ __launch_bounds__(256) void readK(const float4 *src, size_t n, float *sink) {
	float acc = 0.f;
	for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void copyK(const float4 *src, float4 *dst, size_t n) {
	for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) dst[i] = src[i];
}

int main() {
	const size_t bytes = (size_t)4 << 30, n = bytes/16;
	float4 *a, *b; float *sink;
	hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 64);
	hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int grids[] = {1024, 2048, 8192};
	for (int grid : grids) {
		for (int kind = 0; kind < 4; ++kind) {
			float best = 1e9f;
			for (int rep = 0; rep < 4; ++rep) {
				hipEventRecord(e0);
				if (kind == 0) hipLaunchKernelGGL(writeK, dim3(grid), dim3(256), 0, 0, a, n, 1.0f + rep);
				if (kind == 1) hipLaunchKernelGGL(write8K, dim3(grid), dim3(256), 0, 0, (float2 *)a, 2*n, 1.0f + rep);
				if (kind == 2) hipLaunchKernelGGL(readK, dim3(grid), dim3(256), 0, 0, a, n, sink);
				if (kind == 3) hipLaunchKernelGGL(copyK, dim3(grid), dim3(256), 0, 0, a, b, n);
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1);
				if (ms < best) best = ms;
			}
			const char *names[] = {"write 16 B/lane", "write 8 B/lane at 16-B stride, even then odd", "read 16 B/lane", "copy (read + write)"};
			const double moved = (kind == 3 ? 2.0 : 1.0)*bytes;
			printf("grid %5d  %-46s %7.3f ms  %6.2f TB/s\n", grid, names[kind], best, moved/best*1e-9);
		}
	}
	return 0;
}
