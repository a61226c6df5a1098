// Probe: issue cost of the VALU instruction kinds the recurrence wave and the record producers are made of, on one SIMD.
// Each test runs an unrolled body of NI instructions REP times and stamps s_memtime around it; printed: shader cycles per
// instruction for (a) a dependent chain, (b) four independent chains, with 1, 2 and 4 waves resident on the SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum { T_FMA_DEP, T_FMA_IND, T_PKFMA_DEP, T_PKFMA_IND, T_PKMUL_OPSEL_IND, T_CNDMASK_IND, T_DPP_IND, T_RSQ_IND, T_RSQ_DEP, T_MUL_DEP, T_PKFMA_NEG_DEP, T_MOV_IND, T_CNDMASK_SGPR, T_BFI_IND, T_CMP_VCC, T_CMP_SGPR, T_FMA_SGPRSRC, T_FMA_LITERAL, T_CNDMASK_DEP, T_DSREAD_B128, T_DSREAD_B64, T_EXECMOV, T_COUNT };
static const char *kNames[T_COUNT] = {"v_fma_f32 dependent", "v_fma_f32 4 chains", "v_pk_fma_f32 dependent", "v_pk_fma_f32 4 chains", "v_pk_mul_f32 op_sel 4 chains",
                                      "v_cndmask_b32 4 chains", "v_mov_b32 dpp wave_shr:1 4 chains", "v_rsq_f32 4 chains", "v_rsq_f32 dependent", "v_mul_f32 dependent",
                                      "v_pk_fma_f32 op_sel+neg dependent", "v_mov_b32 4 chains", "v_cndmask_b32_e64 sgpr-pair mask 4 chains", "v_bfi_b32 4 chains", "v_cmp_gt_f32 -> vcc (x4)", "v_cmp_gt_f32_e64 -> sgpr pair (x4)", "v_fma_f32 with an SGPR source 4 chains", "v_fma_f32 with a literal 4 chains", "v_cndmask_b32 vcc dependent", "ds_read_b128 (x4, waited per 4)", "ds_read_b64 (x4, waited per 4)", "v_mov_b32 under a partial exec mask 4 chains"};

template <int TEST>
__global__ __launch_bounds__(1024) void probe(unsigned long long *out, float seed, int reps) {
	float a = seed + threadIdx.x, b = 1.0001f, c = 0.5f, d = a + 1, e = a + 2, f = a + 3;
	typedef float v4f __attribute__((ext_vector_type(4)));
	v4f q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
	const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)reps;
	__shared__ float ldsBuf[4096];
	ldsBuf[threadIdx.x] = a; ldsBuf[threadIdx.x + 1024] = a; ldsBuf[threadIdx.x + 2048] = a; ldsBuf[threadIdx.x + 3072] = a;
	__syncthreads();
	const unsigned ldsAddr = (unsigned)(size_t)ldsBuf + (threadIdx.x & 63)*16;
	v2f pa = {a, d}, pb = {b, b}, pc = {c, c}, pd = {e, f}, pe = {f, e}, pf = {d, a};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int r = 0; r < reps; ++r) {
		if (TEST == T_FMA_DEP) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
		if (TEST == T_FMA_IND) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));) }
		if (TEST == T_PKFMA_DEP) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa) : "v"(pb), "v"(pc));) }
		if (TEST == T_PKFMA_IND) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(pa), "+v"(pd), "+v"(pe), "+v"(pf) : "v"(pb), "v"(pc));) }
		if (TEST == T_PKMUL_OPSEL_IND) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[0,1]" : "+v"(pa), "+v"(pd), "+v"(pe), "+v"(pf) : "v"(pb));) }
		if (TEST == T_CNDMASK_IND) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b) : "vcc");) }
		if (TEST == T_DPP_IND) { REP16(asm volatile("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));) }
		if (TEST == T_RSQ_IND) { REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));) }
		if (TEST == T_RSQ_DEP) { REP64(asm volatile("v_rsq_f32 %0, %0" : "+v"(a));) }
		if (TEST == T_MUL_DEP) { REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
		if (TEST == T_PKFMA_NEG_DEP) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(pa) : "v"(pb), "v"(pc));) }
		if (TEST == T_CNDMASK_SGPR) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "s"(mask));) }
		if (TEST == T_BFI_IND) { REP16(asm volatile("v_bfi_b32 %0, %5, %0, %4\n v_bfi_b32 %1, %5, %1, %4\n v_bfi_b32 %2, %5, %2, %4\n v_bfi_b32 %3, %5, %3, %4" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));) }
		if (TEST == T_CMP_VCC) { REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b) : "vcc");) }
		if (TEST == T_CMP_SGPR) { REP16(asm volatile("v_cmp_gt_f32_e64 %5, %0, %4\n v_cmp_gt_f32_e64 %5, %1, %4\n v_cmp_gt_f32_e64 %5, %2, %4\n v_cmp_gt_f32_e64 %5, %3, %4" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "s"(mask));) }
		if (TEST == T_FMA_SGPRSRC) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "s"(seed), "v"(c));) }
		if (TEST == T_FMA_LITERAL) { REP16(asm volatile("v_fmac_f32 %0, 0x3f800347, %4\n v_fmac_f32 %1, 0x3f800347, %4\n v_fmac_f32 %2, 0x3f800347, %4\n v_fmac_f32 %3, 0x3f800347, %4" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c));) }
		if (TEST == T_CNDMASK_DEP) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
		if (TEST == T_DSREAD_B128) { REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(ldsAddr) : "memory");) }
		if (TEST == T_DSREAD_B64) { REP16(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)" : "=v"(pa), "=v"(pd), "=v"(pe), "=v"(pf) : "v"(ldsAddr) : "memory");) }
		if (TEST == T_EXECMOV) { REP16(asm volatile("s_mov_b64 s[70:71], exec\n s_and_b64 exec, exec, %5\n v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n s_mov_b64 exec, s[70:71]" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "s"(mask) : "s70", "s71");) }
		if (TEST == T_MOV_IND) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));) }
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if ((threadIdx.x & 63) == 0) out[(blockIdx.x*16 + (threadIdx.x >> 6))] = t1 - t0;
	if (q0.x + q1.y + q2.z + q3.w + a + d + e + f + pa.x + pa.y + pd.x + pe.y + pf.x == 12345.678f) out[1000] = 1; // keep the results alive
}

template <int TEST>
static void run(unsigned long long *dev) {
	const int reps = 64;
	// waves per SIMD: 64 threads = one wave on one SIMD; 256 = one per SIMD; 512 = two per SIMD; 1024 = four per SIMD
	const int shapes[4] = {64, 256, 512, 1024};
	double res[4];
	for (int i = 0; i < 4; ++i) {
		hipMemset(dev, 0, 4096*8);
		hipLaunchKernelGGL(probe<TEST>, dim3(1), dim3(shapes[i]), 0, 0, dev, 1.0f, reps); // warm-up (instruction cache)
		hipLaunchKernelGGL(probe<TEST>, dim3(1), dim3(shapes[i]), 0, 0, dev, 1.0f, reps);
		hipDeviceSynchronize();
		std::vector<unsigned long long> h(16);
		hipMemcpy(h.data(), dev, 16*8, hipMemcpyDeviceToHost);
		double worst = 0;
		for (int w = 0; w < shapes[i]/64; ++w) worst = h[w] > worst ? double(h[w]) : worst;
		res[i] = worst/(64.0*reps);
	}
	printf("%-36s cycles/instr per wave: 1 wave %.2f | 1 per SIMD %.2f | 2 per SIMD %.2f | 4 per SIMD %.2f\n", kNames[TEST], res[0], res[1], res[2], res[3]);
}

int main() {
	unsigned long long *dev;
	hipMalloc(&dev, 4096*8);
	run<T_FMA_DEP>(dev); run<T_FMA_IND>(dev); run<T_MUL_DEP>(dev); run<T_PKFMA_DEP>(dev); run<T_PKFMA_IND>(dev); run<T_PKFMA_NEG_DEP>(dev);
	run<T_PKMUL_OPSEL_IND>(dev); run<T_CNDMASK_IND>(dev); run<T_DPP_IND>(dev); run<T_MOV_IND>(dev); run<T_RSQ_IND>(dev); run<T_RSQ_DEP>(dev);
	run<T_CNDMASK_DEP>(dev); run<T_CNDMASK_SGPR>(dev); run<T_EXECMOV>(dev); run<T_BFI_IND>(dev); run<T_CMP_VCC>(dev); run<T_CMP_SGPR>(dev); run<T_FMA_SGPRSRC>(dev); run<T_FMA_LITERAL>(dev); run<T_DSREAD_B128>(dev); run<T_DSREAD_B64>(dev);
	int clk = 0;
	hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	printf("(s_memtime ticks; device clock attribute %d kHz)\n", clk);
	return 0;
}
