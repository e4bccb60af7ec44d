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

typedef void (*kern_t)(unsigned long long*, int, float);

int main() {
	unsigned long long* d = nullptr;
	hipMalloc((void**)&d, 1 << 20);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	struct { const char* name; kern_t k; } ks[] = {{"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_max3_f32", k_max3}, {"v_cndmask_b32 (sgpr mask)", k_cndmask}, {"v_cmp_lt_f32 -> sgpr pair", k_cmp_sgpr},
	                                               {"s_and/or_b64", k_salu}, {"v_add_f32 + s_and_b64 alternating (pairs)", k_valu_salu}, {"ds_bpermute_b32 (8 then wait)", k_bperm},
	                                               {"v_mov_b32_dpp quad_perm", k_dpp}, {"v_rcp_f32", k_rcp}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul}};
	const int iters = 2000;
	printf("%d CUs; cycles of one SIMD per wave-instruction (64 lanes), by resident waves per SIMD\n", cus);
	printf("%-44s %8s %8s %8s %8s\n", "instruction", "1 wave", "2", "4", "8");
	for (auto& e : ks) {
		printf("%-44s", e.name);
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
			// s_memtime ticks at 100 MHz on this chip?  report both raw ticks per instruction of ONE wave and the SIMD share
			const double per_instr_wave = med / (iters * 64.0);
			printf(" %8.2f", per_instr_wave / wps);
		}
		printf("   (ticks per instruction of one wave / waves per SIMD)\n");
	}
	// tick calibration: s_memtime vs s_memrealtime (100 MHz)
	return 0;
}
