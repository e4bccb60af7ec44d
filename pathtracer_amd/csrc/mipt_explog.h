// mipt_explog.h — expf, logf and tanf bit-exact with the host libm the reference links (glibc 2.35), for the fog
// branch of getColor (fogContribution / int_exponential, Raytracer.cpp:20-192).
//   expf, logf: e_expf.c / e_logf.c ("optimized routines": everything in double, one rounding to float) in the variants
//   libm selects on FMA hardware (__expf_fma, __logf_fma); the fused operations are the ones of that build (read off its
//   disassembly), the constants those of its .rodata (__exp2f_data, __logf_data).
//   tanf: s_tanf.c (not an ifunc: plain SSE code) = fp64 quadrant reduction as in sinf/cosf, then fdlibm's float
//   kernel k_tanf.c with the correction term y; |x| < 120 (beyond: the caller's general tanf).
// Plain C++ so that tests/native/explog_check.cpp compiles the SAME source with g++ and compares it with libm on every
// float (expf, logf) / every float with |x| < 120 (tanf).  Must be compiled without FP contraction.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#if defined(__HIPCC__)
#define MIPT_EHD __host__ __device__ __forceinline__
#else
#define MIPT_EHD static inline
#endif

MIPT_EHD uint32_t mipt_el_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MIPT_EHD float mipt_el_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
MIPT_EHD uint64_t mipt_el_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
MIPT_EHD double mipt_el_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

MIPT_EHD float mipt_expf(float x) {
	// __exp2f_data.tab: asuint64(2^(i/32)) - (i << 47)
	const uint64_t E[32] = {
		0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
		0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
		0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
		0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
		0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
		0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
	const double SHIFT = 0x1.8p+52, InvLn2N = 0x1.71547652b82fep+5;
	const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
	const double xd = (double)x;
	const uint32_t abstop = (mipt_el_asuint(x) >> 20) & 0x7ff;
	if (abstop >= 0x42b) {                                   // |x| >= 88 or NaN
		if (mipt_el_asuint(x) == 0xff800000u) return 0.0f;
		if (abstop >= 0x7f8) return x + x;
		if (x > 0x1.62e42ep6f) return INFINITY;              // overflow
		if (x < -0x1.9fe368p6f) return 0.0f;                 // underflow
		if (x < -0x1.9d1d9ep6f) return mipt_el_asfloat(1u);   // __math_may_uflowf: 0x1.4p-75f * 0x1.4p-75f = the smallest subnormal
	}
	double kd = fma(InvLn2N, xd, SHIFT);                     // z + SHIFT with z never rounded on its own
	const uint64_t ki = mipt_el_asuint64(kd);
	kd -= SHIFT;
	const double r = fma(InvLn2N, xd, -kd);
	uint64_t t = E[ki % 32];
	t += ki << 47;
	const double s = mipt_el_asdouble(t);
	const double z = fma(C0, r, C1);
	const double r2 = r * r;
	double y = fma(C2, r, 1.0);
	y = fma(z, r2, y);
	y = y * s;
	return (float)y;
}

MIPT_EHD float mipt_logf(float x) {
	const double T[16][2] = {   // __logf_data.tab: {invc, logc}
		{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2},
		{0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
		{0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
		{0x1.0000000000000p+0, 0x0.0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
		{0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
		{0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
	const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
	uint32_t ix = mipt_el_asuint(x);
	if (ix == 0x3f800000u) return 0.0f;
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {     // x < 0x1p-126, inf or NaN
		if (ix * 2 == 0) return -INFINITY;
		if (ix == 0x7f800000u) return x;
		if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / 0.0f;   // __math_invalidf
		ix = mipt_el_asuint(x * 0x1p23f);                     // subnormal: normalise
		ix -= 23u << 23;
	}
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (tmp >> 19) % 16;
	const int k = (int32_t)tmp >> 23;
	const uint32_t iz = ix - (tmp & 0xff800000u);
	const double invc = T[i][0], logc = T[i][1];
	const double z = (double)mipt_el_asfloat(iz);
	const double r = fma(z, invc, -1.0);
	const double y0 = fma((double)k, Ln2, logc);
	const double r2 = r * r;
	double y = fma(A1, r, A2);
	y = fma(A0, r2, y);
	y = fma(y, r2, y0 + r);
	return (float)y;
}

// __kernel_tanf (k_tanf.c): tan(x + y) on [-pi/4, pi/4], iy = 1: tan, iy = -1: -1/tan.  Float arithmetic, no fusion.
MIPT_EHD float mipt_kernel_tanf(float x, float y, int iy) {
	const float one = 1.0f, pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
	const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f,
	            T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f,
	            T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f, T12 = 2.5907305826e-05f;
	float z, r, v, w, s;
	const int32_t hx = (int32_t)mipt_el_asuint(x);
	const int32_t ix = hx & 0x7fffffff;
	if (ix < 0x39000000) {                                   // |x| < 2**-13
		if ((int)x == 0) {
			if ((ix | (iy + 1)) == 0) return one / fabsf(x);
			else if (iy == 1) return x;
			else return -one / x;
		}
	}
	if (ix >= 0x3f2ca140) {                                  // |x| >= 0.6744
		if (hx < 0) { x = -x; y = -y; }
		z = pio4 - x;
		w = pio4lo - y;
		x = z + w; y = 0.0f;
		if (fabsf(x) < 0x1p-13f) return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
	}
	z = x * x;
	w = z * z;
	r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
	v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
	s = z * x;
	r = y + z * (s * (r + v) + y);
	r += T0 * s;
	w = x + r;
	if (ix >= 0x3f2ca140) {
		v = (float)iy;
		return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
	}
	if (iy == 1) return w;
	// -1/(x+r), accurately
	float a, t;
	z = mipt_el_asfloat(mipt_el_asuint(w) & 0xfffff000u);
	v = r - (z - x);
	t = a = -1.0f / w;
	t = mipt_el_asfloat(mipt_el_asuint(t) & 0xfffff000u);
	s = 1.0f + t * z;
	return t + a * (s + t * v);
}
// tanf for |x| < 120 (returns false beyond, and for inf / NaN)
MIPT_EHD bool mipt_tanf_main(float x, float& out) {
	const uint32_t ix = mipt_el_asuint(x) & 0x7fffffffu;
	if (ix <= 0x3f490fdau) { out = mipt_kernel_tanf(x, 0.0f, 1); return true; }
	if (((ix >> 20) & 0x7ff) >= ((mipt_el_asuint(120.0f) >> 20) & 0x7ff)) return false;
	const double xd = (double)x;
	const double r = xd * 0x1.45F306DC9C883p+23;            // reduce_fast (s_sincosf.h)
	const int n = ((int32_t)r + 0x800000) >> 24;
	const double dx = xd - (double)n * 0x1.921FB54442D18p0;
	const float y0 = (float)dx;
	const float y1 = (float)(dx - (double)y0);
	out = mipt_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
	return true;
}
