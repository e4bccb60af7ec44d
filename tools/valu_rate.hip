// tools/valu_rate.hip — issue cost of the vector / scalar instructions the traversal's inner step is made of, on one SIMD
// with 1, 2, 4 and 8 resident waves: cycles of SIMD time per wave-instruction (s_memtime deltas over long unrolled runs of
// independent instructions).  Build + run on the GPU box:
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
// What it answers: is a 64-lane VALU instruction 2 or 4 cycles of a SIMD when several waves issue; what a packed fp32
// instruction, a v_cndmask with an SGPR mask, a VOPC compare, a ds_bpermute cost beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstring>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                                                         \
	__global__ void __launch_bounds__(64) name(unsigned long long* out, int iters, float seed) {                    \
		float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
		float b0 = seed * 2, b1 = seed * 3;                                                                         \
		unsigned long long m = __ballot(threadIdx.x & 1);                                                           \
		unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                       \
		for (int i = 0; i < iters; i++) { REP8(body) }                                                              \
		unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                       \
		if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                            \
		if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 == 12345.f) out[0] = m;                                 \
	}

// 8 independent instructions per body, 8 bodies per iteration = 64 instructions per iteration
KERNEL(k_add, asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %8, %0, %9\n v_max3_f32 %1, %8, %1, %9\n v_max3_f32 %2, %8, %2, %9\n v_max3_f32 %3, %8, %3, %9\n v_max3_f32 %4, %8, %4, %9\n v_max3_f32 %5, %8, %5, %9\n v_max3_f32 %6, %8, %6, %9\n v_max3_f32 %7, %8, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %8, %0, %9\n v_cndmask_b32 %1, %8, %1, %9\n v_cndmask_b32 %2, %8, %2, %9\n v_cndmask_b32 %3, %8, %3, %9\n v_cndmask_b32 %4, %8, %4, %9\n v_cndmask_b32 %5, %8, %5, %9\n v_cndmask_b32 %6, %8, %6, %9\n v_cndmask_b32 %7, %8, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(m));)
KERNEL(k_cmp_sgpr, asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
KERNEL(k_salu, asm volatile("s_and_b64 s[20:21], s[20:21], %8\n s_or_b64 s[22:23], s[22:23], %8\n s_and_b64 s[24:25], s[24:25], %8\n s_or_b64 s[26:27], s[26:27], %8\n s_and_b64 s[20:21], s[20:21], %8\n s_or_b64 s[22:23], s[22:23], %8\n s_and_b64 s[24:25], s[24:25], %8\n s_or_b64 s[26:27], s[26:27], %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
KERNEL(k_valu_salu, asm volatile("v_add_f32 %0, %8, %0\n s_and_b64 s[20:21], s[20:21], %9\n v_add_f32 %1, %8, %1\n s_or_b64 s[22:23], s[22:23], %9\n v_add_f32 %2, %8, %2\n s_and_b64 s[24:25], s[24:25], %9\n v_add_f32 %3, %8, %3\n s_or_b64 s[26:27], s[26:27], %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
KERNEL(k_bperm, asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_dpp, asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)

KERNEL(k_add_e64, asm volatile("v_add_f32_e64 %0, %8, %0\n v_add_f32_e64 %1, %8, %1\n v_add_f32_e64 %2, %8, %2\n v_add_f32_e64 %3, %8, %3\n v_add_f32_e64 %4, %8, %4\n v_add_f32_e64 %5, %8, %5\n v_add_f32_e64 %6, %8, %6\n v_add_f32_e64 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_min_e32, asm volatile("v_min_f32_e32 %0, %8, %0\n v_min_f32_e32 %1, %8, %1\n v_min_f32_e32 %2, %8, %2\n v_min_f32_e32 %3, %8, %3\n v_min_f32_e32 %4, %8, %4\n v_min_f32_e32 %5, %8, %5\n v_min_f32_e32 %6, %8, %6\n v_min_f32_e32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
KERNEL(k_sub_sgpr, asm volatile("v_sub_f32_e32 %0, %8, %0\n v_sub_f32_e32 %1, %8, %1\n v_sub_f32_e32 %2, %8, %2\n v_sub_f32_e32 %3, %8, %3\n v_sub_f32_e32 %4, %8, %4\n v_sub_f32_e32 %5, %8, %5\n v_sub_f32_e32 %6, %8, %6\n v_sub_f32_e32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));)
KERNEL(k_cndmask_vcc, asm volatile("v_cndmask_b32_e32 %0, %8, %0, vcc\n v_cndmask_b32_e32 %1, %8, %1, vcc\n v_cndmask_b32_e32 %2, %8, %2, vcc\n v_cndmask_b32_e32 %3, %8, %3, vcc\n v_cndmask_b32_e32 %4, %8, %4, vcc\n v_cndmask_b32_e32 %5, %8, %5, vcc\n v_cndmask_b32_e32 %6, %8, %6, vcc\n v_cndmask_b32_e32 %7, %8, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");)
KERNEL(k_cmp_vcc, asm volatile("v_cmp_lt_f32_e32 vcc, %0, %8\n v_cmp_lt_f32_e32 vcc, %1, %8\n v_cmp_lt_f32_e32 vcc, %2, %8\n v_cmp_lt_f32_e32 vcc, %3, %8\n v_cmp_lt_f32_e32 vcc, %4, %8\n v_cmp_lt_f32_e32 vcc, %5, %8\n v_cmp_lt_f32_e32 vcc, %6, %8\n v_cmp_lt_f32_e32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");)
KERNEL(k_and, asm volatile("v_and_b32_e32 %0, %8, %0\n v_and_b32_e32 %1, %8, %1\n v_and_b32_e32 %2, %8, %2\n v_and_b32_e32 %3, %8, %3\n v_and_b32_e32 %4, %8, %4\n v_and_b32_e32 %5, %8, %5\n v_and_b32_e32 %6, %8, %6\n v_and_b32_e32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_xor_e64, asm volatile("v_xor_b32_e64 %0, %8, %0\n v_xor_b32_e64 %1, %8, %1\n v_xor_b32_e64 %2, %8, %2\n v_xor_b32_e64 %3, %8, %3\n v_xor_b32_e64 %4, %8, %4\n v_xor_b32_e64 %5, %8, %5\n v_xor_b32_e64 %6, %8, %6\n v_xor_b32_e64 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %8, 3, %0\n v_lshl_add_u32 %1, %8, 3, %1\n v_lshl_add_u32 %2, %8, 3, %2\n v_lshl_add_u32 %3, %8, 3, %3\n v_lshl_add_u32 %4, %8, 3, %4\n v_lshl_add_u32 %5, %8, 3, %5\n v_lshl_add_u32 %6, %8, 3, %6\n v_lshl_add_u32 %7, %8, 3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_mov, asm volatile("v_mov_b32_e32 %0, %8\n v_mov_b32_e32 %1, %8\n v_mov_b32_e32 %2, %8\n v_mov_b32_e32 %3, %8\n v_mov_b32_e32 %4, %8\n v_mov_b32_e32 %5, %8\n v_mov_b32_e32 %6, %8\n v_mov_b32_e32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_max3_same, asm volatile("v_max3_f32 %0, %0, %0, %8\n v_max3_f32 %1, %1, %1, %8\n v_max3_f32 %2, %2, %2, %8\n v_max3_f32 %3, %3, %3, %8\n v_max3_f32 %4, %4, %4, %8\n v_max3_f32 %5, %5, %5, %8\n v_max3_f32 %6, %6, %6, %8\n v_max3_f32 %7, %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %8, %0, %9\n v_med3_f32 %1, %8, %1, %9\n v_med3_f32 %2, %8, %2, %9\n v_med3_f32 %3, %8, %3, %9\n v_med3_f32 %4, %8, %4, %9\n v_med3_f32 %5, %8, %5, %9\n v_med3_f32 %6, %8, %6, %9\n v_med3_f32 %7, %8, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
KERNEL(k_mbcnt, asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_lo_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_lo_u32_b32 %3, -1, %3\n v_mbcnt_lo_u32_b32 %4, -1, %4\n v_mbcnt_lo_u32_b32 %5, -1, %5\n v_mbcnt_lo_u32_b32 %6, -1, %6\n v_mbcnt_lo_u32_b32 %7, -1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL(k_readlane, asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s20, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s20, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s20, %5\n v_readfirstlane_b32 s20, %6\n v_readfirstlane_b32 s20, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20");)
KERNEL(k_salu2, asm volatile("s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec\n s_and_b64 s[24:25], s[24:25], exec\n s_or_b64 s[26:27], s[26:27], exec\n s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec\n s_and_b64 s[24:25], s[24:25], exec\n s_or_b64 s[26:27], s[26:27], exec" : "+v"(a0) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
KERNEL(k_valu_salu2, asm volatile("v_add_f32 %0, %4, %0\n s_and_b64 s[20:21], s[20:21], exec\n v_add_f32 %1, %4, %1\n s_or_b64 s[22:23], s[22:23], exec\n v_add_f32 %2, %4, %2\n s_and_b64 s[24:25], s[24:25], exec\n v_add_f32 %3, %4, %3\n s_or_b64 s[26:27], s[26:27], exec" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
KERNEL(k_bcnt, asm volatile("s_bcnt1_i32_b64 s20, exec\n s_bcnt1_i32_b64 s21, exec\n s_bcnt1_i32_b64 s22, exec\n s_bcnt1_i32_b64 s23, exec\n s_bcnt1_i32_b64 s20, exec\n s_bcnt1_i32_b64 s21, exec\n s_bcnt1_i32_b64 s22, exec\n s_bcnt1_i32_b64 s23, exec" : "+v"(a0) : : "s20", "s21", "s22", "s23", "scc");)
__global__ void __launch_bounds__(64) k_lds(unsigned long long* out, int iters, float seed) {
	__shared__ unsigned long long buf[64 * 16];
	unsigned addr = threadIdx.x * 8;
	unsigned long long v = (unsigned long long)seed;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < iters; i++) {
		REP8(asm volatile("ds_write_b64 %1, %0\n ds_write_b64 %1, %0 offset:512\n ds_write_b64 %1, %0 offset:1024\n ds_write_b64 %1, %0 offset:1536\n ds_read_b64 %0, %1\n ds_read_b64 %0, %1 offset:512\n ds_read_b64 %0, %1 offset:1024\n ds_read_b64 %0, %1 offset:1536\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(addr) : "memory");)
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
	if (v == 12345ull) out[0] = buf[1];
}
KERNEL(k_mix_bcnt, asm volatile("v_add_f32 %0, %4, %0\n s_bcnt1_i32_b64 s20, exec\n v_add_f32 %1, %4, %1\n s_bcnt1_i32_b64 s21, exec\n v_add_f32 %2, %4, %2\n s_bcnt1_i32_b64 s22, exec\n v_add_f32 %3, %4, %3\n s_bcnt1_i32_b64 s23, exec" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_mix_slow_bcnt, asm volatile("v_max3_f32 %0, %4, %0, %5\n s_bcnt1_i32_b64 s20, exec\n v_max3_f32 %1, %4, %1, %5\n s_bcnt1_i32_b64 s21, exec\n v_max3_f32 %2, %4, %2, %5\n s_bcnt1_i32_b64 s22, exec\n v_max3_f32 %3, %4, %3, %5\n s_bcnt1_i32_b64 s23, exec" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_sand, asm volatile("s_and_b32 s20, s20, s24\n s_or_b32 s21, s21, s24\n s_and_b32 s22, s22, s24\n s_or_b32 s23, s23, s24\n s_and_b32 s20, s20, s24\n s_or_b32 s21, s21, s24\n s_and_b32 s22, s22, s24\n s_or_b32 s23, s23, s24" : "+v"(a0) : : "s20", "s21", "s22", "s23", "s24", "scc");)
KERNEL(k_sand64, asm volatile("s_and_b64 s[20:21], s[20:21], s[28:29]\n s_or_b64 s[22:23], s[22:23], s[28:29]\n s_and_b64 s[24:25], s[24:25], s[28:29]\n s_or_b64 s[26:27], s[26:27], s[28:29]\n s_and_b64 s[20:21], s[20:21], s[28:29]\n s_or_b64 s[22:23], s[22:23], s[28:29]\n s_and_b64 s[24:25], s[24:25], s[28:29]\n s_or_b64 s[26:27], s[26:27], s[28:29]" : "+v"(a0) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");)
KERNEL(k_saveexec, asm volatile("s_and_saveexec_b64 s[20:21], s[28:29]\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], s[28:29]\n s_or_b64 exec, exec, s[22:23]\n s_and_saveexec_b64 s[24:25], s[28:29]\n s_or_b64 exec, exec, s[24:25]\n s_and_saveexec_b64 s[26:27], s[28:29]\n s_or_b64 exec, exec, s[26:27]" : "+v"(a0) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");)
KERNEL(k_branch, asm volatile("s_cbranch_execz 1f\n1:\n s_cbranch_execz 2f\n2:\n s_cbranch_execz 3f\n3:\n s_cbranch_execz 4f\n4:\n s_cbranch_execz 5f\n5:\n s_cbranch_execz 6f\n6:\n s_cbranch_execz 7f\n7:\n s_cbranch_execz 8f\n8:" : "+v"(a0));)
KERNEL(k_waitcnt, asm volatile("s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)" : "+v"(a0));)
KERNEL(k_nop, asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0" : "+v"(a0));)
// packed fp32: register pairs
#define KERNEL_PK(name, body)                                                                                      \
	__global__ void __launch_bounds__(64) name(unsigned long long* out, int iters, float seed) {                    \
		typedef float f2 __attribute__((ext_vector_type(2)));                                                       \
		f2 a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed + 5}, a3 = {seed + 6, seed + 7};  \
		f2 a4 = {seed, seed + 1.5f}, a5 = {seed + 2.5f, seed + 3}, a6 = {seed + 4.5f, seed + 5}, a7 = {seed + 6.5f, seed + 7}; \
		f2 b0 = {seed * 2, seed * 3};                                                                               \
		unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                       \
		for (int i = 0; i < iters; i++) { REP8(body) }                                                              \
		unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                       \
		if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                            \
		if (a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y == 12345.f) out[0] = 1;                           \
	}
KERNEL_PK(k_pk_add, asm volatile("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
KERNEL_PK(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)


// ---- dependent gathers: every active lane walks a random cycle through a table of 64-byte records (four dwordx4 loads
//      per step, the next index comes out of the record), as a traversal step does with its node.  Cycles per step of one
//      wave and steps per microsecond of the chip, by table size, waves per SIMD and active lanes.
template <int REC>
__global__ void __launch_bounds__(256) k_chase_rec(const float4* __restrict__ tab, unsigned nrec, int steps, unsigned active, unsigned long long* out, unsigned* sink) {
	const unsigned lane = threadIdx.x & 63u;
	unsigned cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nrec;
	unsigned acc = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	if (lane < active) {
		for (int i = 0; i < steps; i++) {
			const float4* q = tab + REC * (size_t)cur;
			float4 v[REC];
#pragma unroll
			for (int k = 0; k < REC; k++) v[k] = q[k];
#pragma unroll
			for (int k = 1; k < REC; k++) acc += __float_as_uint(v[k].x) ^ __float_as_uint(v[k].w);
			cur = __float_as_uint(v[0].x);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
	if (acc == 0x12345678u) sink[0] = cur;
}

__global__ void __launch_bounds__(256) k_chase(const float4* __restrict__ tab, unsigned nrec, int steps, unsigned active, unsigned long long* out, unsigned* sink) {
	const unsigned lane = threadIdx.x & 63u;
	unsigned cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nrec;
	unsigned acc = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	if (lane < active) {
		for (int i = 0; i < steps; i++) {
			const float4* q = tab + 4 * (size_t)cur;
			const float4 a = q[0], b = q[1], c = q[2], d = q[3];
			acc += __float_as_uint(a.y) ^ __float_as_uint(b.x) ^ __float_as_uint(c.x) ^ __float_as_uint(d.x);
			cur = __float_as_uint(a.x);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
	if (acc == 0x12345678u) sink[0] = cur;
}

template <int REC>
static void chase_rec_bench(size_t mb) {
	const unsigned nrec = (unsigned)(mb * 1024 * 1024 / (16 * REC));
	std::vector<unsigned> perm(nrec);
	for (unsigned i = 0; i < nrec; i++) perm[i] = i;
	unsigned long long s = 88172645463325252ull;
	for (unsigned i = nrec - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; unsigned j = (unsigned)(s % (i + 1)); std::swap(perm[i], perm[j]); }
	std::vector<float> host((size_t)nrec * 4 * REC, 0.f);
	for (unsigned i = 0; i < nrec; i++) { unsigned nxt = perm[(i + 1) % nrec]; memcpy(&host[(size_t)perm[i] * 4 * REC], &nxt, 4); }
	float4* tab; unsigned long long* out; unsigned* sink;
	hipMalloc((void**)&tab, (size_t)nrec * 16 * REC); hipMalloc((void**)&out, 1 << 22); hipMalloc((void**)&sink, 64);
	hipMemcpy(tab, host.data(), (size_t)nrec * 16 * REC, hipMemcpyHostToDevice);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int blocks = prop.multiProcessorCount * 4, steps = 2000;
	hipLaunchKernelGGL(k_chase_rec<REC>, dim3(blocks), dim3(256), 0, 0, tab, nrec, 50, 64u, out, sink);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k_chase_rec<REC>, dim3(blocks), dim3(256), 0, 0, tab, nrec, steps, 64u, out, sink);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	const double lane_steps = (double)blocks * 4 * 64 * steps;
	printf("%5zu MB table, %3d-byte records (4 waves per SIMD, 64 lanes): %7.2f G records/s  %7.0f GB/s\n", mb, 16 * REC, lane_steps / (ms * 1e-3) / 1e9, lane_steps * 16 * REC / (ms * 1e-3) / 1e9);
	hipFree(tab); hipFree(out); hipFree(sink);
}

static void chase_bench() {
	printf("\ndependent gathers of larger records (aligned to their size)\n");
	for (size_t mb : {16, 212, 2048}) { chase_rec_bench<4>(mb); chase_rec_bench<8>(mb); chase_rec_bench<12>(mb); chase_rec_bench<16>(mb); }
	printf("\ndependent 64-byte gathers (4 x global_load_dwordx4 per step, next index from the record)\n");
	printf("%-10s %6s %6s %14s %16s %14s\n", "table", "waves", "lanes", "ticks/step", "Gsteps/s (lanes)", "GB/s (64 B)");
	for (size_t mb : {16, 212, 2048}) {
		const unsigned nrec = (unsigned)(mb * 1024 * 1024 / 64);
		std::vector<unsigned> perm(nrec);
		for (unsigned i = 0; i < nrec; i++) perm[i] = i;
		unsigned long long s = 88172645463325252ull;
		for (unsigned i = nrec - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; unsigned j = (unsigned)(s % (i + 1)); std::swap(perm[i], perm[j]); }
		std::vector<float> host((size_t)nrec * 16, 0.f);
		// one random cycle: record perm[i] points to perm[i+1]
		for (unsigned i = 0; i < nrec; i++) { unsigned nxt = perm[(i + 1) % nrec]; memcpy(&host[(size_t)perm[i] * 16], &nxt, 4); }
		float4* tab; unsigned long long* out; unsigned* sink;
		hipMalloc((void**)&tab, (size_t)nrec * 64); hipMalloc((void**)&out, 1 << 22); hipMalloc((void**)&sink, 64);
		hipMemcpy(tab, host.data(), (size_t)nrec * 64, hipMemcpyHostToDevice);
		hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
		const int cus = prop.multiProcessorCount;
		for (int wps : {2, 4, 7, 8}) for (unsigned active : {16u, 32u, 64u}) {
			const int blocks = cus * wps;                       // 256-thread blocks: 4 waves each, wps blocks per CU = wps waves per SIMD
			const int steps = 2000;
			hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(256), 0, 0, tab, nrec, 50, active, out, sink);
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(256), 0, 0, tab, nrec, steps, active, out, sink);
			hipEventRecord(e1, 0); hipEventSynchronize(e1);
			float ms = 0; hipEventElapsedTime(&ms, e0, e1);
			std::vector<unsigned long long> h((size_t)blocks * 4);
			hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
			std::sort(h.begin(), h.end());
			const double lane_steps = (double)blocks * 4 * active * steps;
			printf("%-10s %6d %6u %14.0f %16.2f %14.0f\n", (std::to_string(mb) + " MB").c_str(), wps, active, (double)h[h.size() / 2] / steps, lane_steps / (ms * 1e-3) / 1e9, lane_steps * 64 / (ms * 1e-3) / 1e9);
		}
		hipFree(tab); hipFree(out); hipFree(sink);
	}
}


// ---- vector-memory issue: loads that hit L1 (every lane re-reads its own 64 bytes), 8 independent loads then a wait.
//      Wall nanoseconds per wave-instruction and CU: what the address unit / L1 charge per instruction, by width and by
//      the number of active lanes.
template <int WIDTH>
__global__ void __launch_bounds__(256) k_vmem(const float4* __restrict__ tab, int iters, unsigned active, float* sink) {
	const unsigned lane = threadIdx.x & 63u;
	const float4* p = tab + 4 * (size_t)((blockIdx.x * 256u + threadIdx.x) & 4095u);
	float acc = 0.f;
	if (lane < active) {
		for (int i = 0; i < iters; i++) {
			if (WIDTH == 4) { float4 a, b, c, d;
				asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n global_load_dwordx4 %3, %4, off offset:48\n"
				             "global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n global_load_dwordx4 %3, %4, off offset:48\n s_waitcnt vmcnt(0)"
				             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
				acc += a.x + b.x + c.x + d.x; }
			else if (WIDTH == 2) { float2 a, b, c, d;
				asm volatile("global_load_dwordx2 %0, %4, off\n global_load_dwordx2 %1, %4, off offset:16\n global_load_dwordx2 %2, %4, off offset:32\n global_load_dwordx2 %3, %4, off offset:48\n"
				             "global_load_dwordx2 %0, %4, off\n global_load_dwordx2 %1, %4, off offset:16\n global_load_dwordx2 %2, %4, off offset:32\n global_load_dwordx2 %3, %4, off offset:48\n s_waitcnt vmcnt(0)"
				             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
				acc += a.x + b.x + c.x + d.x; }
			else { float a, b, c, d;
				asm volatile("global_load_dword %0, %4, off\n global_load_dword %1, %4, off offset:16\n global_load_dword %2, %4, off offset:32\n global_load_dword %3, %4, off offset:48\n"
				             "global_load_dword %0, %4, off\n global_load_dword %1, %4, off offset:16\n global_load_dword %2, %4, off offset:32\n global_load_dword %3, %4, off offset:48\n s_waitcnt vmcnt(0)"
				             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
				acc += a + b + c + d; }
		}
	}
	if (acc == 12345.f) sink[0] = acc;
}
template <int WIDTH>
static void vmem_bench(const char* name) {
	float4* tab; float* sink;
	hipMalloc((void**)&tab, 4096 * 64 + 4096); hipMalloc((void**)&sink, 64);
	hipMemset(tab, 0, 4096 * 64 + 4096);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	printf("%-24s", name);
	for (int wps : {2, 7}) for (unsigned active : {16u, 32u, 64u}) {
		const int blocks = cus * wps, iters = 4000;
		hipLaunchKernelGGL(k_vmem<WIDTH>, dim3(blocks), dim3(256), 0, 0, tab, 10, active, sink);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_vmem<WIDTH>, dim3(blocks), dim3(256), 0, 0, tab, iters, active, sink);
		hipEventRecord(e1, 0); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		// instructions per CU: wps blocks x 4 waves x iters x 8
		printf("  %dw/%2u lanes %6.2f", wps, active, ms * 1e6 / ((double)wps * 4 * iters * 8));
	}
	printf("   (ns per wave-instruction and CU)\n");
	hipFree(tab); hipFree(sink);
}

typedef void (*kern_t)(unsigned long long*, int, float);

int main() {
	unsigned long long* d = nullptr;
	hipMalloc((void**)&d, 1 << 20);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	struct { const char* name; kern_t k; } ks[] = {{"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_max3_f32", k_max3}, {"v_cndmask_b32 (sgpr mask)", k_cndmask}, {"v_cmp_lt_f32 -> sgpr pair", k_cmp_sgpr},
	                                               {"s_and/or_b64", k_salu}, {"v_add_f32 + s_and_b64 alternating (pairs)", k_valu_salu}, {"ds_bpermute_b32 (8 then wait)", k_bperm},
	                                               {"v_mov_b32_dpp quad_perm", k_dpp}, {"v_rcp_f32", k_rcp}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul},
	                                               {"v_add_f32_e64", k_add_e64}, {"v_min_f32_e32", k_min_e32}, {"v_fma_f32", k_fma}, {"v_sub_f32_e32 (sgpr src0)", k_sub_sgpr}, {"v_cndmask_b32_e32 (vcc)", k_cndmask_vcc},
	                                               {"v_cmp_lt_f32_e32 -> vcc", k_cmp_vcc}, {"v_and_b32_e32", k_and}, {"v_xor_b32_e64", k_xor_e64}, {"v_lshl_add_u32", k_lshl_add}, {"v_mov_b32_e32", k_mov},
	                                               {"v_max3_f32 (two operands the same register)", k_max3_same}, {"v_med3_f32", k_med3}, {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_readfirstlane_b32", k_readlane},
	                                               {"s_and/or_b64 (sgprs, exec)", k_salu2}, {"v_add_f32 + s_and_b64 alternating (count = pairs*2)", k_valu_salu2}, {"s_bcnt1_i32_b64", k_bcnt}, {"v_add_f32 + s_bcnt1 alternating (per instruction)", k_mix_bcnt}, {"v_max3_f32 + s_bcnt1 alternating (per instruction)", k_mix_slow_bcnt},
	                                               {"s_and/or_b32 (sgprs)", k_sand}, {"s_and/or_b64 (sgprs)", k_sand64}, {"s_and_saveexec + s_or exec (per instruction)", k_saveexec},
	                                               {"s_cbranch_execz not taken", k_branch}, {"s_waitcnt (nothing outstanding)", k_waitcnt}, {"s_nop 0", k_nop}, {"ds_write_b64 x4 + ds_read_b64 x4 + wait (per op)", k_lds}};
	const int iters = 2000;
	printf("%d CUs; cycles of one SIMD per wave-instruction (64 lanes), by resident waves per SIMD\n", cus);
	printf("%-44s %8s %8s %8s %8s\n", "instruction", "1 wave", "2", "4", "8");
	for (auto& e : ks) {
		printf("%-44s", e.name);
		double wall[4]; int nw = 0;
		for (int wps : {1, 2, 4, 8}) {
			// blocks of 64 threads: 4 * wps waves per CU = wps per SIMD
			const int blocks = cus * 4 * wps;
			hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0f);      // warm up
			hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
			hipDeviceSynchronize();
			std::vector<unsigned long long> h(blocks);
			hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
			std::sort(h.begin(), h.end());
			const double med = (double)h[blocks / 2];
			const double per_instr_wave = med / (iters * 64.0);
			printf(" %8.2f", per_instr_wave / wps);
			{   // the same by the wall clock (hipEvents): nanoseconds of one SIMD per wave-instruction
				hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
				hipEventRecord(e0, 0);
				hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d, iters * 4, 1.0f);
				hipEventRecord(e1, 0); hipEventSynchronize(e1);
				float ms = 0; hipEventElapsedTime(&ms, e0, e1);
				wall[nw++] = ms * 1e6 / (iters * 4 * 64.0) / wps;
				hipEventDestroy(e0); hipEventDestroy(e1);
			}
		}
		printf("   | wall ns per instruction and SIMD: %6.3f %6.3f %6.3f %6.3f\n", wall[0], wall[1], wall[2], wall[3]);
	}
	printf("\nvector-memory loads that hit L1, by width, waves per SIMD and active lanes\n");
	vmem_bench<4>("global_load_dwordx4"); vmem_bench<2>("global_load_dwordx2"); vmem_bench<1>("global_load_dword");
	chase_bench();
	return 0;
}
