// Probe: what does the staging traffic of kVocoder's producers cost by itself?  One workgroup per CU (the LDS request keeps a
// second one away), W waves, each issuing LOADS wide loads per "block" whose 64 lanes cover short contiguous runs (a row's
// window) at row-pitch distances, the windows advancing 64 bytes per block -- the producers' pattern without their arithmetic.
// Prints the time per block per CU for variations of: run length, alignment, load width, advance (0 = every block re-reads the
// same lines: L1 hits), number of waves.
// Build: hipcc --offload-arch=gfx950 -O2 -o stage_load_rate stage_load_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int LOADS = 6;

template <int BYTES>
__global__ __launch_bounds__(1024) void probe(const char *base, size_t wgStride, size_t runStride, int runLen, int advance, int misalign, int iters, float *sink) {
	extern __shared__ float4 lds[];
	using V = typename std::conditional<BYTES == 16, float4, float2>::type;
	const int wave = threadIdx.x >> 6, k = threadIdx.x & 63;
	const int runsPerWave = (LOADS*64 + runLen - 1)/runLen;
	const char *p[LOADS];
	for (int i = 0; i < LOADS; ++i) {
		const int q = k + 64*i, run = q/runLen, off = (q%runLen)*BYTES;
		p[i] = base + blockIdx.x*wgStride + (size_t)(wave*runsPerWave + run)*runStride + off + misalign;
	}
	V v[LOADS], w[LOADS];
	float acc = 0.f;
	for (int i = 0; i < LOADS; ++i) v[i] = *reinterpret_cast<const V *>(p[i]);
	for (int n = 1; n <= iters; ++n) {
		for (int i = 0; i < LOADS; ++i) w[i] = *reinterpret_cast<const V *>(p[i] + (size_t)n*advance); // next block's windows
		for (int i = 0; i < LOADS; ++i) acc += v[i].x + v[i].y;                                          // "park" the current ones
		for (int i = 0; i < LOADS; ++i) v[i] = w[i];
	}
	if (acc == 12345.678f) sink[0] = acc;
}

int main() {
	const size_t runStride = 24704, wgStride = (size_t)16*96*runStride + (1 << 20); // a row of 3088 bins; room for 16 waves x 96 runs, and for the pure-streaming case
	const size_t total = 256*wgStride + (1 << 20);
	char *dev; float *sink;
	hipMalloc(&dev, total); hipMalloc(&sink, 64);
	hipMemset(dev, 0, total);
	hipFuncSetAttribute((const void *)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024);
	hipFuncSetAttribute((const void *)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct Case { const char *name; int bytes, waves, runLen, advance, misalign; };
	const Case cases[] = {
		{"producers' pattern: 8 waves, 16 B, runs of 12, +64 B/block, aligned", 16, 8, 12, 64, 0},
		{"  same, 8-byte misaligned", 16, 8, 12, 64, 8},
		{"  same, no advance (L1 hits)", 16, 8, 12, 0, 0},
		{"  runs of 6 (PV/ROT windows)", 16, 8, 6, 64, 0},
		{"  runs of 6, misaligned", 16, 8, 6, 64, 8},
		{"  runs of 64 (1 KB contiguous per instruction)", 16, 8, 64, 64, 0},
		{"  runs of 64, advance 1024 (pure streaming)", 16, 8, 64, 1024, 0},
		{"  runs of 8, advance 128 (one whole line per run and block)", 16, 8, 8, 128, 0},
		{"  runs of 4, advance 64 (half a line per run and block)", 16, 8, 4, 64, 0},
		{"  8-byte loads, runs of 24", 8, 8, 24, 64, 0},
		{"  4 waves", 16, 4, 12, 64, 0},
		{"  16 waves", 16, 16, 12, 64, 0},
		{"  2 waves", 16, 2, 12, 64, 0},
		{"  1 wave", 16, 1, 12, 64, 0},
	};
	const int iters = 380;
	for (const Case &c : cases) {
		float best = 1e9f;
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(e0);
			if (c.bytes == 16) hipLaunchKernelGGL(probe<16>, dim3(256), dim3(64*c.waves), 150*1024, 0, dev, wgStride, runStride, c.runLen, c.advance, c.misalign, iters, sink);
			else hipLaunchKernelGGL(probe<8>, dim3(256), dim3(64*c.waves), 150*1024, 0, dev, wgStride, runStride, c.runLen, c.advance, c.misalign, iters, sink);
			hipEventRecord(e1);
			if (hipEventSynchronize(e1) != hipSuccess) { printf("failed\n"); return 1; }
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (ms < best) best = ms;
		}
		const double usPerBlock = best*1e3/iters, instrs = (double)c.waves*LOADS, bytes = instrs*64*c.bytes;
		printf("%-70s %7.3f us/block  = %6.0f ns per load instruction, %6.1f GB/s per CU, %5.2f TB/s chip\n", c.name, usPerBlock, usPerBlock*1e3/instrs, bytes/usPerBlock*1e-3, bytes*256/usPerBlock*1e-6);
	}
	return 0;
}
