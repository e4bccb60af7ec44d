// mipt_sincos.h — sinf / cosf bit-exact with the host libm the reference links (glibc 2.35
// s_sinf.c / s_cosf.c, the "optimized routines" scheme: fp64 quadrant reduction n = round(x*2/pi),
// then a degree-7 sine or degree-8 cosine minimax polynomial of the reduced argument) for |x| < 120.
// Plain C++ (no HIP intrinsics) so that tests/native/sincos_check.cpp compiles the SAME source with
// g++ and compares it with libm on every float in [0, 6.5], the range the path uses (arguments are
// float(2*pi)*u, u in [0,1]).  Must be compiled without FP contraction.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define MIPT_HD __host__ __device__ __forceinline__
#else
#define MIPT_HD static inline
#endif

MIPT_HD float mipt_sincos_poly(double x, double x2, int n, bool neg) {
	// neg selects the table with negated cosine coefficients (quadrant bit n & 2)
	if ((n & 1) == 0) {
		const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
		double x3 = x * x2;
		double t1 = s2 + x2 * s3;
		double x7 = x3 * x2;
		double s = x + x3 * s1;
		return (float)(s + x7 * t1);
	} else {
		// glibc's second table holds the five cosine coefficients negated.  Every operation below is odd in the coefficient set
		// (round-to-nearest is symmetric: fl(-a) = -fl(a)), so the polynomial of the negated table is the negated polynomial, bit
		// for bit, and so is its conversion to float (the value is never 0: |x| <= pi/4): one sign flip instead of five selected
		// doubles — which cost the shade kernels ten registers and, at 128 registers, sixteen spilled values per vertex.
		const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
		double x4 = x2 * x2;
		double t2 = c3 + x2 * c4;
		double t1 = c0 + x2 * c1;
		double x6 = x4 * x2;
		double c = t1 + x4 * c2;
		const float r = (float)(c + x6 * t2);
		return neg ? -r : r;
	}
}
MIPT_HD uint32_t mipt_abstop12(float x) { return (__builtin_bit_cast(uint32_t, x) >> 20) & 0x7ff; }
template <bool COS>
MIPT_HD float mipt_sincosf(float y) {
	double x = (double)y;
	if (mipt_abstop12(y) < mipt_abstop12(0x1.921FB6p-1f)) {          // |y| < pi/4
		double x2 = x * x;
		if (mipt_abstop12(y) < mipt_abstop12(0x1p-12f)) return COS ? 1.0f : y;
		return mipt_sincos_poly(x, x2, COS ? 1 : 0, false);
	}
	// |y| < 120: fast reduction (hpi_inv is 2/pi * 2^24, quadrant in bits 24..31)
	double r = x * 0x1.45F306DC9C883p+23;
	int n = ((int)r + 0x800000) >> 24;
	x = x - (double)n * 0x1.921FB54442D18p0;
	double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign table {1,-1,-1,1}
	return mipt_sincos_poly(x * s, x * x, COS ? (n ^ 1) : n, (n & 2) != 0);
}

// sinf(y) AND cosf(y) of one argument (the samplers always want both: Vector.h:582-600): the same reduction and the same two polynomials as the
// single functions above, each evaluated ONCE.  Taken one after the other the two calls evaluate four polynomials on a wave whose lanes
// differ in the quadrant (each call runs the sine polynomial for its even-quadrant lanes and the cosine polynomial for the odd ones).
MIPT_HD void mipt_sincosf_pair(float y, float& sin_out, float& cos_out) {
	double x = (double)y;
	int n = 0;
	double xs = x;
	bool neg = false;
	if (mipt_abstop12(y) < mipt_abstop12(0x1.921FB6p-1f)) {          // |y| < pi/4
		if (mipt_abstop12(y) < mipt_abstop12(0x1p-12f)) { sin_out = y; cos_out = 1.0f; return; }
	} else {
		double r = x * 0x1.45F306DC9C883p+23;
		n = ((int)r + 0x800000) >> 24;
		x = x - (double)n * 0x1.921FB54442D18p0;
		const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
		xs = x * sg;
		neg = (n & 2) != 0;
	}
	const double x2 = x * x;
	const float S = mipt_sincos_poly(xs, x2, 0, false);             // the sine polynomial of the signed reduced argument
	const float C0 = mipt_sincos_poly(xs, x2, 1, false);            // the cosine polynomial (even in x)
	const float C = neg ? -C0 : C0;
	sin_out = (n & 1) == 0 ? S : C;
	cos_out = (n & 1) == 0 ? C : S;
}
