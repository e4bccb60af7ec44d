// mipt_powf.h — powf bit-exact with the host libm the reference links (glibc 2.35 e_powf.c, the "optimized routines"
// scheme: log2(x) from a 16-entry table and a degree-5 polynomial, y*log2(x) in fp64, 2^t from a 32-entry table and a
// degree-3 polynomial, everything in double and one final rounding to float).  The reference runs the variant libm
// selects on FMA-capable x86-64 (__powf_fma); the fused operations below are exactly the ones of that build, read off
// its disassembly, and the table / polynomial constants are those of its .rodata (__powf_log2_data, __exp2f_data).
// Plain C++ so that tests/native/powf_check.cpp compiles the SAME source with g++ and compares it with libm.
//
// Covered here: 0 < x < inf (normal or subnormal), y finite and non-zero.  Everything else (x <= 0, x or y zero / inf /
// NaN) returns false and is left to the caller's general powf: those results are exact special values in any libm.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#if defined(__HIPCC__)
#define MIPT_PHD __host__ __device__ __forceinline__
#else
#define MIPT_PHD static inline
#endif

MIPT_PHD uint32_t mipt_pw_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MIPT_PHD float mipt_pw_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
MIPT_PHD uint64_t mipt_pw_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
MIPT_PHD double mipt_pw_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

MIPT_PHD bool mipt_powf_main(float x, float y, float& out) {
	// __powf_log2_data.tab: {invc, logc}
	const double T[16][2] = {
		{0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2},
		{0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
		{0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
		{0x1.0000000000000p+0, 0x0.0p+0}, {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
		{0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
		{0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
	const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
	// __exp2f_data.tab: asuint64(2^(i/32)) - (i << 47)
	const uint64_t E[32] = {
		0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
		0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
		0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
		0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
		0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
		0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
	const double SHIFT = 0x1.8p+47, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;

	uint32_t ix = mipt_pw_asuint(x);
	const uint32_t iy = mipt_pw_asuint(y);
	if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) return false;            // y is zero, inf or NaN
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
		if ((ix & 0x80000000u) || 2u * ix - 1u >= 2u * 0x7f800000u - 1u) return false;   // x negative, zero, inf or NaN
		ix = mipt_pw_asuint(x * 0x1p23f);                                 // subnormal x: normalise
		ix &= 0x7fffffffu;
		ix -= 23u << 23;
	}
	// log2_inline
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (int)((tmp >> 19) & 15u);
	const uint32_t top = tmp & 0xff800000u;
	const uint32_t iz = ix - top;
	const int k = (int32_t)top >> 23;
	const double invc = T[i][0], logc = T[i][1];
	const double z = (double)mipt_pw_asfloat(iz);
	const double r = fma(z, invc, -1.0);
	const double y0 = logc + (double)k;
	const double p01 = fma(r, A0, A1);
	const double p23 = fma(r, A2, A3);
	const double r2 = r * r;
	const double q = fma(r, A4, y0);
	const double r4 = r2 * r2;
	const double s = fma(r2, p23, q);
	const double logx = fma(p01, r4, s);
	const double ylogx = (double)y * logx;
	if (((mipt_pw_asuint64(ylogx) >> 47) & 0xffffu) >= 0x80bfu) {        // |y*log2(x)| >= 126
		if (ylogx > 0x1.fffffffd1d571p+6) { out = mipt_pw_asfloat(0x7f800000u); return true; }   // __math_oflowf
		if (ylogx <= -150.0) { out = 0.f; return true; }                  // __math_uflowf
		if (ylogx < -149.0) { out = mipt_pw_asfloat(1u); return true; }    // __math_may_uflowf: 0x1.4p-75f * 0x1.4p-75f rounds to 0x1p-149f
	}
	// exp2_inline
	double kd = ylogx + SHIFT;
	const uint64_t ki = mipt_pw_asuint64(kd);
	kd -= SHIFT;
	const double rr = ylogx - kd;
	const uint64_t t = E[ki & 31u] + (ki << 47);
	const double zz = fma(rr, C0, C1);
	const double rr2 = rr * rr;
	double yy = fma(rr, C2, 1.0);
	yy = fma(zz, rr2, yy);
	yy = yy * mipt_pw_asdouble(t);
	out = (float)yy;
	return true;
}
