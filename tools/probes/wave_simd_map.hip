// Probe: which SIMD does wave w of a 1024-thread workgroup land on?  (HW_REG_HW_ID, gfx9 layout: WAVE_ID[3:0] SIMD_ID[5:4]
// PIPE_ID[7:6] CU_ID[11:8] SH_ID[12] SE_ID[15:13].)  Same launch shape as kVocoder: 16 waves, 4 per SIMD, one workgroup per CU
// (LDS-limited).  Build: hipcc --offload-arch=gfx950 -O2 -o wave_simd_map wave_simd_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe(unsigned *out) {
	extern __shared__ unsigned char lds[];
	if ((threadIdx.x & 63) == 0) {
		const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
		out[blockIdx.x*16 + (threadIdx.x >> 6)] = id;
	}
	if (threadIdx.x == 5000) lds[0] = 1;
}

int main() {
	const int blocks = 1024;
	unsigned *dev;
	hipMalloc(&dev, blocks*16*4);
	hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024);
	for (int lds : {0, 150*1024}) {
		hipMemset(dev, 0, blocks*16*4);
		hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), lds, 0, dev);
		hipDeviceSynchronize();
		std::vector<unsigned> h(blocks*16);
		hipMemcpy(h.data(), dev, blocks*16*4, hipMemcpyDeviceToHost);
		int identity = 0, perSimdOk = 0;
		int hist[16][4] = {};
		for (int b = 0; b < blocks; ++b) {
			bool id = true;
			int cnt[4] = {};
			for (int w = 0; w < 16; ++w) {
				const int simd = (h[b*16 + w] >> 4) & 3;
				hist[w][simd]++;
				cnt[simd]++;
				if (simd != ((w + ((h[b*16] >> 4) & 3)) & 3)) id = false;
			}
			identity += id;
			perSimdOk += (cnt[0] == 4 && cnt[1] == 4 && cnt[2] == 4 && cnt[3] == 4);
		}
		printf("lds %d: %d of %d workgroups round-robin from wave 0's SIMD; %d with 4 waves on every SIMD\n", lds, identity, blocks, perSimdOk);
		for (int w = 0; w < 16; ++w) printf("  wave %2d: SIMD0 %4d  SIMD1 %4d  SIMD2 %4d  SIMD3 %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
		printf("  first workgroups:");
		for (int b = 0; b < 4; ++b) { printf(" ["); for (int w = 0; w < 16; ++w) printf("%d", (h[b*16 + w] >> 4) & 3); printf("]"); }
		printf("\n");
	}
	return 0;
}
