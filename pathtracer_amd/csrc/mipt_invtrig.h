// mipt_invtrig.h — acosf / atanf / atan2f bit-exact with the host libm the reference links (glibc 2.35: the fdlibm
// float routines e_acosf.c, s_atanf.c, e_atan2f.c — plain single-precision +, -, *, /, sqrt in a fixed order, no FMA
// variant exists for them).  Operation order and constants are those of the installed libm.so.6 (read off its
// disassembly and .rodata); tests/native/invtrig_check.cpp compiles the SAME source with g++ and compares it with libm
// on every float in [-1, 1] (acosf), on every float (atanf, sampled exhaustively over exponents) and on random and
// structured pairs (atan2f).  Must be compiled without FP contraction; division and sqrt must be correctly rounded.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#if defined(__HIPCC__)
#define MIPT_IHD __host__ __device__ __forceinline__
#else
#define MIPT_IHD static inline
#endif

MIPT_IHD uint32_t mipt_it_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MIPT_IHD float mipt_it_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

MIPT_IHD float mipt_acosf(float x) {
	const float pi = 0x1.921fb4p+1f, pi_lo2 = 0x1.4442dp-23f, pio2_hi = 0x1.921fb4p+0f, pio2_lo = 0x1.4442dp-24f;
	const float p5 = 0x1.23de1p-15f, p4 = 0x1.9efe08p-11f, p3 = 0x1.48228cp-5f, p2 = 0x1.9c155p-3f, p1 = 0x1.4d612p-2f, p0 = 0x1.555556p-3f;
	const float q4 = 0x1.3b8c5cp-4f, q3 = 0x1.6066c2p-1f, q2 = 0x1.02ae5ap+1f, q1 = 0x1.33a272p+1f;
	const uint32_t hx = mipt_it_asuint(x), ix = hx & 0x7fffffffu;
	if (ix == 0x3f800000u) return (hx >> 31) ? pi_lo2 + pi : 0.f;
	if (ix > 0x3f800000u) return (x - x) / (x - x);
	if (ix < 0x3f000000u) {                                              // |x| < 0.5
		if (ix <= 0x32800000u) return pio2_lo + pio2_hi;
		const float z = x * x;
		const float p = (((((p5 * z + p4) * z - p3) * z + p2) * z - p1) * z + p0) * z;
		const float q = (((q4 * z - q3) * z + q2) * z - q1) * z + 1.f;
		const float r = p / q;
		return pio2_hi - (x - (pio2_lo - x * r));
	}
	if (hx >> 31) {                                                      // x < -0.5
		const float z = (x + 1.f) * 0.5f;
		const float p = (((((p5 * z + p4) * z - p3) * z + p2) * z - p1) * z + p0) * z;
		const float q = (((q4 * z - q3) * z + q2) * z - q1) * z + 1.f;
		const float s = sqrtf(z);
		const float r = p / q;
		const float t = (r * s - pio2_lo) + s;
		return pi - (t + t);
	}
	const float z = (1.f - x) * 0.5f;                                    // x > 0.5
	const float s = sqrtf(z);
	const float df = mipt_it_asfloat(mipt_it_asuint(s) & 0xfffff000u);
	const float p = (((((p5 * z + p4) * z - p3) * z + p2) * z - p1) * z + p0) * z;
	const float q = (((q4 * z - q3) * z + q2) * z - q1) * z + 1.f;
	const float r = p / q;
	const float c = (z - df * df) / (s + df);
	const float t = (r * s + c) + df;
	return t + t;
}

MIPT_IHD float mipt_atanf(float x) {
	const float hi3 = 0x1.921fb4p+0f, lo3 = 0x1.4442dp-24f;
	const float aT0 = 0x1.555556p-2f, aT2 = 0x1.24924ap-3f, aT4 = 0x1.745cdcp-4f, aT6 = 0x1.10d66ap-4f, aT8 = 0x1.97b4b2p-5f, aT10 = 0x1.0ad3aep-6f;
	const float aT9 = -0x1.2b4442p-5f, b7 = 0x1.dde2d6p-5f, b5 = 0x1.3b0f2ap-4f, b3 = 0x1.c71c7p-4f, b1 = 0x1.99999ap-3f;   // |aT7|, |aT5|, |aT3|, |aT1|: subtracted
	const uint32_t hx = mipt_it_asuint(x), ix = hx & 0x7fffffffu;
	if (ix >= 0x4c000000u) {                                             // |x| >= 2^25
		if (ix > 0x7f800000u) return x + x;
		return (hx >> 31) ? -hi3 - lo3 : lo3 + hi3;
	}
	float hi = 0.f, lo = 0.f, xr = x;
	int id = -1;
	if (ix < 0x3ee00000u) {                                              // |x| < 0.4375
		if (ix < 0x31000000u) return x;                                   // huge + x > one
	} else {
		const float ax = mipt_it_asfloat(ix);
		if (ix < 0x3f980000u) {
			if (ix < 0x3f300000u) { id = 0; xr = ((ax + ax) - 1.f) / (ax + 2.f); hi = 0x1.dac67p-2f; lo = 0x1.586ed2p-28f; }
			else { id = 1; xr = (ax - 1.f) / (ax + 1.f); hi = 0x1.921fb4p-1f; lo = 0x1.4442dp-25f; }
		} else {
			if (ix < 0x401c0000u) { id = 2; xr = (ax - 1.5f) / (ax * 1.5f + 1.f); hi = 0x1.f730bcp-1f; lo = 0x1.281f68p-25f; }
			else { id = 3; xr = -1.f / ax; hi = hi3; lo = lo3; }
		}
	}
	const float z = xr * xr, w = z * z;
	const float s1 = (((((aT10 * w + aT8) * w + aT6) * w + aT4) * w + aT2) * w + aT0) * z;
	const float s2 = ((((aT9 * w - b7) * w - b5) * w - b3) * w - b1) * w;
	const float m = (s1 + s2) * xr;
	if (id < 0) return xr - m;
	const float r = hi - ((m - lo) - xr);
	return (hx >> 31) ? -r : r;
}

MIPT_IHD float mipt_atan2f(float y, float x) {
	const float tiny = 0x1.4484cp-100f, pi_o_4 = 0x1.921fb6p-1f, pi_o_2 = 0x1.921fb6p+0f, pi = 0x1.921fb6p+1f, neg_pi_lo = 0x1.777a5cp-24f, half_neg_pi_lo = 0x1.777a5cp-25f;
	const uint32_t hx = mipt_it_asuint(x), hy = mipt_it_asuint(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
	if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;
	if (hx == 0x3f800000u) return mipt_atanf(y);
	const uint32_t m = ((hy >> 31) & 1u) | ((hx >> 30) & 2u);
	if (iy == 0) {
		if (m == 2) return tiny + pi;
		if (m == 3) return -pi - tiny;
		return y;
	}
	if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : tiny + pi_o_2;
	if (ix == 0x7f800000u) {
		if (iy == 0x7f800000u) {
			if (m == 0) return tiny + pi_o_4;
			if (m == 1) return -pi_o_4 - tiny;
			if (m == 2) return 3.f * pi_o_4 + tiny;
			return -3.f * pi_o_4 - tiny;
		}
		if (m == 0) return 0.f;
		if (m == 1) return -0.f;
		if (m == 2) return tiny + pi;
		return -pi - tiny;
	}
	if (iy == 0x7f800000u) return (hy >> 31) ? -pi_o_2 - tiny : tiny + pi_o_2;
	const int32_t d = (int32_t)(iy - ix);
	float z;
	if (d > 0x1e7fffff) z = pi_o_2 - half_neg_pi_lo;                     // |y/x| > 2^60
	else if ((hx >> 31) && (d >> 23) < -60) z = 0.f;                      // |y|/x < -2^-60
	else z = mipt_atanf(fabsf(y / x));
	if (m == 0) return z;
	if (m == 1) return mipt_it_asfloat(mipt_it_asuint(z) + 0x80000000u);
	if (m == 2) return pi - (neg_pi_lo + z);
	return (z + neg_pi_lo) - pi;
}
