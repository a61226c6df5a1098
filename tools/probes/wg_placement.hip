// Probe: where does the dispatcher put the workgroups of a grid that is as large as the number of CUs, when two of them
// WOULD fit on one CU?  (kVocoder launches one workgroup per stream; if 256 workgroups of <= 80 KB land on 128 CUs, two
// recurrences share a CU's VALU, if they land on 256 CUs nothing changes until S > 256.)
// HW_ID (reg 4): CU_ID[11:8] SH_ID[12] SE_ID[15:13]; XCC_ID (reg 20): [3:0].
// Build: hipcc --offload-arch=gfx950 -O2 -o wg_placement wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(1024) void probe(unsigned *out, int spin) {
	extern __shared__ unsigned char lds[];
	if (threadIdx.x == 0) {
		const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
		const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
		out[blockIdx.x] = ((xcc & 15) << 16) | (hw & 0xffff);
	}
	// stay resident long enough for the whole grid to be dispatched beside us
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(10);
	if (threadIdx.x == 5000) lds[0] = 1;
}

int main() {
	unsigned *dev;
	hipMalloc(&dev, 4096*4);
	hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
	const int ldsSizes[] = {150*1024, 78*1024, 50*1024, 0};
	const int grids[] = {256, 320, 512};
	const int threads[] = {1024, 640};
	for (int th : threads) for (int lds : ldsSizes) for (int grid : grids) {
		hipMemset(dev, 0xff, 4096*4);
		hipLaunchKernelGGL(probe, dim3(grid), dim3(th), lds, 0, dev, 400000);
		if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed (threads %d lds %d grid %d)\n", th, lds, grid); continue; }
		std::vector<unsigned> h(grid);
		hipMemcpy(h.data(), dev, grid*4, hipMemcpyDeviceToHost);
		std::map<unsigned, int> perCu;
		int perXcc[16] = {};
		for (int b = 0; b < grid; ++b) {
			const unsigned xcc = (h[b] >> 16) & 15, cu = (h[b] >> 8) & 15, sh = (h[b] >> 12) & 1, se = (h[b] >> 13) & 7;
			perCu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
			perXcc[xcc]++;
		}
		int hist[8] = {};
		for (auto &kv : perCu) hist[kv.second < 7 ? kv.second : 7]++;
		printf("threads %4d lds %6d grid %3d: %3zu distinct CUs; CUs holding 1/2/3/4 workgroups: %d/%d/%d/%d; per XCC:", th, lds, grid, perCu.size(), hist[1], hist[2], hist[3], hist[4]);
		for (int x = 0; x < 8; ++x) printf(" %d", perXcc[x]);
		printf("\n");
	}
	return 0;
}
