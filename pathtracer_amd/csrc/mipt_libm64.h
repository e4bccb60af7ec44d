// mipt_libm64.h — double-precision exp / pow / sincos / acos / atan2 bit-exact with the host libm the reference links (glibc 2.35, x86-64).
//
// The reference's path calls them in three places: `exp` in the subsurface profile weight (Raytracer.cpp:381), `cos` / `sin` /
// `pow` in random_Phong (BRDF.h:41-46), `cos` / `sin` / `acos` / `atan2` in the MERL half / difference-angle transform
// (MERLBRDFRead.cpp:49-127).
// Every cos / sin there comes as a pair on one argument, which GCC compiles to ONE call of sincos() (the compiled reference
// and the oracle import sincos, pow, exp, acos, atan2 from libm and nothing else of this kind: objdump -d | grep call).
// glibc picks, through ifunc, the variants of exp and pow built with -mfma on every x86-64 CPU that has FMA and AVX2
// (__exp_fma, __pow_fma: sysdeps/ieee754/dbl-64/e_exp.c, e_pow.c, Szabolcs Nagy's table-driven routines, with the
// contractions GCC chose), and likewise __ieee754_acos_fma / __ieee754_atan2_fma (e_asin.c, e_atan2.c, IBM's accurate
// library as of glibc 2.34+: the multi-precision fall-backs are gone, every range returns its first estimate); sincos
// has no such variant and is the plain build of IBM's accurate kernels (s_sincos.c).
// What is below restates those algorithms with every fused operation written as an explicit fma(), in the places the
// disassembly of the installed libm.so.6 has them (objdump -d: 0x76470, 0x768b0, 0x77960, 0x78060 of Ubuntu's 2.35-0ubuntu3.x build), and
// with none in sincos; the tables come out of the same file (tests/native/gen_libm64_tables.py -> mipt_libm64_tables.h).
// Plain C++ (no HIP intrinsics), compiled without FP contraction, so that tests/native/libm64_check.cpp builds the SAME
// source with g++ and compares it with libm on billions of arguments.
//
// Coverage: the main paths and the argument ranges the path can reach; whatever is outside (|x| > 1.05e8 for sincos,
// non-finite or non-positive bases and out-of-range exponents for pow) returns false from the *_main() form and the caller
// uses the device library, whose results for those exact special values are not in question.  acos and atan2 are complete
// (every argument, special values included).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define MIPT_L64 __host__ __device__ __forceinline__
#define MIPT_L64_TABLE static __device__ const
#else
#define MIPT_L64 static inline
#define MIPT_L64_TABLE static const
#endif
#include "mipt_libm64_tables.h"

MIPT_L64 uint64_t l64_bits(double x) { return __builtin_bit_cast(uint64_t, x); }
MIPT_L64 double l64_dbl(uint64_t u) { return __builtin_bit_cast(double, u); }
MIPT_L64 double l64_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
MIPT_L64 double l64_abs(double x) { return l64_dbl(l64_bits(x) & 0x7fffffffffffffffull); }
MIPT_L64 double l64_copysign(double mag, double sgn) { return l64_dbl((l64_bits(mag) & 0x7fffffffffffffffull) | (l64_bits(sgn) & 0x8000000000000000ull)); }

// ---------------------------------------------------------------- exp (e_exp.c: N = 128, degree-5 polynomial)
#define L64_INVLN2N 0x1.71547652b82fep+7
#define L64_SHIFT 0x1.8p+52
#define L64_NEGLN2HIN (-0x1.62e42fefa0000p-8)
#define L64_NEGLN2LON (-0x1.cf79abc9e3b3ap-47)
#define L64_C2 0x1.ffffffffffdbdp-2
#define L64_C3 0x1.555555555543cp-3
#define L64_C4 0x1.55555cf172b91p-5
#define L64_C5 0x1.1111167a4d017p-7

// specialcase() of e_exp.c: the result's exponent would over- or underflow the scale factor
MIPT_L64 double l64_exp_special(double tmp, uint64_t sbits, uint64_t ki) {
	if ((ki & 0x80000000ull) == 0) {                     // k > 0
		sbits -= 1009ull << 52;
		const double scale = l64_dbl(sbits);
		return 0x1p1009 * l64_fma(scale, tmp, scale);
	}
	sbits += 1022ull << 52;                              // k < 0: care in the subnormal range
	const double scale = l64_dbl(sbits);
	const double st = scale * tmp;
	double y = scale + st;
	if (y < 1.0) {
		double lo = (scale - y) + st;
		const double hi = 1.0 + y;
		lo = ((1.0 - hi) + y) + lo;
		y = (hi + lo) - 1.0;
		if (y == 0.0) y = 0.0;
	}
	return 0x1p-1022 * y;
}
// exp_inline() of e_pow.c == the body of exp() with xtail = 0 and sign_bias = 0
MIPT_L64 double l64_exp_core(double x, double xtail, bool with_tail) {
	uint32_t abstop = (uint32_t)(l64_bits(x) >> 52) & 0x7ffu;
	if (abstop - 0x3c9u >= 0x3fu) {
		if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;                         // |x| < 2^-54
		if (abstop >= 0x409u) {                                                      // |x| >= 1024, inf, nan
			if (l64_bits(x) == 0xfff0000000000000ull) return 0.0;
			if (abstop >= 0x7ffu) return 1.0 + x;
			return (l64_bits(x) >> 63) ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;    // __math_uflow / __math_oflow
		}
		abstop = 0;                                                                  // 512 <= |x| < 1024: special-cased below
	}
	double kd = l64_fma(x, L64_INVLN2N, L64_SHIFT);
	const uint64_t ki = l64_bits(kd);
	kd -= L64_SHIFT;
	double r = l64_fma(kd, L64_NEGLN2HIN, x);
	r = l64_fma(kd, L64_NEGLN2LON, r);
	if (with_tail) r = xtail + r;
	const uint64_t idx = 2 * (ki % 128);
	const uint64_t top = ki << 45;
	const double tail = l64_dbl(mipt_l64_exp_tab[idx]);
	const uint64_t sbits = mipt_l64_exp_tab[idx + 1] + top;
	const double p = l64_fma(L64_C3, r, L64_C2);
	const double q = r + tail;
	const double r2 = r * r;
	const double s = l64_fma(r, L64_C5, L64_C4);
	const double t = l64_fma(p, r2, q);
	const double r4 = r2 * r2;
	const double tmp = l64_fma(r4, s, t);
	if (abstop == 0) return l64_exp_special(tmp, sbits, ki);
	const double scale = l64_dbl(sbits);
	return l64_fma(scale, tmp, scale);
}
MIPT_L64 double mipt_exp64(double x) { return l64_exp_core(x, 0.0, false); }

// ---------------------------------------------------------------- pow (e_pow.c: log_inline with a 128-entry table, then exp_inline)
#define L64_LN2HI 0x1.62e42fefa3800p-1
#define L64_LN2LO 0x1.ef35793c76730p-45
#define L64_A0 (-0x1.0000000000000p-1)
#define L64_A1 (-0x1.5555555555560p-1)
#define L64_A2 0x1.0000000000006p-1
#define L64_A3 0x1.999999959554ep-1
#define L64_A4 (-0x1.555555529a47ap-1)
#define L64_A5 (-0x1.2495b9b4845e9p+0)
#define L64_A6 0x1.0002b8b263fc3p+0
// x positive and normal, |y| in [2^-65, 2^63): everything the path can ask for (x in (0, 1], y in (0, 1]).  false otherwise.
MIPT_L64 bool mipt_pow64_main(double x, double y, double& out) {
	const uint64_t ix = l64_bits(x), iy = l64_bits(y);
	const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
	if (topx - 1u >= 0x7ffu - 1u || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu) return false;
	const uint64_t tmp = ix - 0x3fe6955500000000ull;
	const int i = (int)((tmp >> 45) % 128);
	const int k = (int)((int64_t)tmp >> 52);
	const uint64_t iz = ix - (tmp & (0xfffull << 52));
	const double z = l64_dbl(iz), kd = (double)k;
	const double invc = l64_dbl(mipt_l64_pow_log_tab[4 * i]), logc = l64_dbl(mipt_l64_pow_log_tab[4 * i + 2]), logctail = l64_dbl(mipt_l64_pow_log_tab[4 * i + 3]);
	const double r = l64_fma(z, invc, -1.0);
	const double t1 = l64_fma(kd, L64_LN2HI, logc);
	const double lo1 = l64_fma(kd, L64_LN2LO, logctail);
	const double ar = L64_A0 * r;
	const double p1 = l64_fma(r, L64_A2, L64_A1);
	const double p2 = l64_fma(r, L64_A4, L64_A3);
	const double t2 = t1 + r;
	const double ar2 = r * ar;
	const double lo2a = t1 - t2;
	const double ar3 = r * ar2;
	const double lo3 = l64_fma(ar, r, -ar2);
	const double lo2 = lo2a + r;
	const double p3 = l64_fma(r, L64_A6, L64_A5);
	const double hi = t2 + ar2;
	const double hd = t2 - hi;
	const double p23 = l64_fma(p3, ar2, p2);
	const double lo4 = hd + ar2;
	const double pin = l64_fma(ar2, p23, p1);
	double lo = lo1 + lo2;
	lo = lo + lo3;
	lo = lo + lo4;
	lo = l64_fma(ar3, pin, lo);
	const double lhi = hi + lo;
	const double ltail = (hi - lhi) + lo;
	const double ehi = y * lhi;
	const double et = l64_fma(lhi, y, -ehi);
	const double elo = l64_fma(y, ltail, et);
	out = l64_exp_core(ehi, elo, true);
	return true;
}

// ---------------------------------------------------------------- sin / cos (s_sin.c, the IBM Accurate Mathematical Library routines)
#define L64_BIG 0x1.8p+45
#define L64_TOINT 0x1.8p+52
#define L64_HPINV 0x1.45f306dc9c883p-1
#define L64_MP1 0x1.921fb58000000p+0
#define L64_MP2 (-0x1.dde973c000000p-27)
#define L64_PP3 (-0x1.cb3b398000000p-55)
#define L64_PP4 (-0x1.d747f23e32ed7p-83)
#define L64_HP0 0x1.921fb54442d18p+0
#define L64_HP1 0x1.1a62633145c07p-54
#define L64_S1 (-0x1.5555555555555p-3)
#define L64_S2 0x1.1111111110ecep-7
#define L64_S3 (-0x1.a01a019db08b8p-13)
#define L64_S4 0x1.71de27b9a7ed9p-19
#define L64_S5 (-0x1.addffc2fcdf59p-26)
#define L64_SN3 (-0x1.5555555555515p-3)
#define L64_SN5 0x1.11110e829872fp-7
#define L64_CS2 0x1.0p-1
#define L64_CS4 (-0x1.5555555555535p-5)
#define L64_CS6 0x1.6c16bedd9e239p-10

// Where sincos / acos / atan2 read their tables.  Default: the library's constant arrays.  A kernel that spends its time in these functions may hand
// in copies it made in LDS (k_wf_merl_eval, mipt_wavefront.h: ~60 dependent table reads per BRDF evaluation; every function below is inlined, so the
// compiler sees the address space and emits ds_read).  Same words, same arithmetic.
#define MIPT_L64_SINCOS_WORDS 440
#define MIPT_L64_ASNCS_WORDS 2568
#define MIPT_L64_INROOT_WORDS 128
#define MIPT_L64_CIJ_WORDS 1687
struct L64Tables { const uint64_t *sincos, *asncs, *inroot, *cij; };
MIPT_L64 L64Tables l64_tables() { L64Tables T; T.sincos = mipt_l64_sincos_tab; T.asncs = mipt_l64_asncs_tab; T.inroot = mipt_l64_inroot_tab; T.cij = mipt_l64_cij_tab; return T; }
struct L64Tab { double sn, ssn, cs, ccs; };
MIPT_L64 L64Tab l64_lookup(double u, const L64Tables& TB) {
	const int k = (int)(uint32_t)l64_bits(u) * 4;        // u.i[LOW_HALF] * 4
	L64Tab t;
	t.sn = l64_dbl(TB.sincos[k]); t.ssn = l64_dbl(TB.sincos[k + 1]);
	t.cs = l64_dbl(TB.sincos[k + 2]); t.ccs = l64_dbl(TB.sincos[k + 3]);
	return t;
}
// ---------------------------------------------------------------- sincos (s_sincos.c)
// glibc's sincos() is not sin() next to cos(): it has no FMA variant (sysdeps/x86_64/fpu/multiarch has none for
// s_sincos.c), so it is the plain-SSE2 build of IBM's kernels — no fused operation anywhere — and it takes the
// 0.855 <= |x| < 2.426 range through a renormalised (a, da) pair for BOTH results, where sin() uses (hp0 - |x|, hp1)
// directly.  About 2 results in 10 000 differ from those of sin() / cos() in the last bit (tests/native/libm64_check.cpp
// counts them), so it is sincos that is restated here.
MIPT_L64 double l64p_do_cos(double x, double dx, const L64Tables& TB) {
	if (x < 0) dx = -dx;
	const double ax = l64_abs(x);
	const double u = L64_BIG + ax;
	const double xr = (ax - (u - L64_BIG)) + dx;
	const double xx = xr * xr;
	const double s = xr + (xr * xx) * (L64_SN3 + xx * L64_SN5);
	const double c = xx * (L64_CS2 + xx * (L64_CS4 + xx * L64_CS6));
	const L64Tab T = l64_lookup(u, TB);
	const double cor = ((T.ccs - s * T.ssn) - T.cs * c) - T.sn * s;
	return T.cs + cor;
}
MIPT_L64 double l64p_do_sin(double x, double dx, const L64Tables& TB) {
	const double xold = x;
	const double ax = l64_abs(x);
	if (ax < 0.126) {
		const double xx = x * x;
		const double poly = ((((L64_S5 * xx + L64_S4) * xx + L64_S3) * xx + L64_S2) * xx) + L64_S1;
		const double t = (poly * x - 0.5 * dx) * xx + dx;
		return x + t;
	}
	if (!(0.0 < x)) dx = -dx;
	const double u = L64_BIG + ax;
	const double xr = ax - (u - L64_BIG);
	const double xx = xr * xr;
	const double s = xr + (dx + (xr * xx) * (L64_SN3 + xx * L64_SN5));
	const double c = xr * dx + xx * (L64_CS2 + xx * (L64_CS4 + xx * L64_CS6));
	const L64Tab T = l64_lookup(u, TB);
	const double cor = ((T.ssn + s * T.ccs) - T.sn * c) + T.cs * s;
	return l64_copysign(T.sn + cor, xold);
}
MIPT_L64 int l64p_reduce(double x, double& a, double& da) {
	const double t = x * L64_HPINV + L64_TOINT;
	const double xn = t - L64_TOINT;
	const double y = (x - xn * L64_MP1) - xn * L64_MP2;
	const int n = (int)(uint32_t)l64_bits(t) & 3;
	double t1 = xn * L64_PP3;
	const double t2 = y - t1;
	double db = (y - t2) - t1;
	t1 = xn * L64_PP4;
	const double b = t2 - t1;
	db += (t2 - b) - t1;
	a = b; da = db;
	return n;
}
MIPT_L64 double l64p_do_sincos(double a, double da, int n, const L64Tables& TB) {
	const double r = (n & 1) ? l64p_do_cos(a, da, TB) : l64p_do_sin(a, da, TB);
	return (n & 2) ? -r : r;
}
// sincos(x, &s, &c) for |x| < 105414350; false for larger, infinite or NaN arguments
MIPT_L64 bool mipt_sincos64_main(double x, double& sn, double& cs, const L64Tables& TB) {
	const int32_t k = (int32_t)(l64_bits(x) >> 32) & 0x7fffffff;
	if (k < 0x400368fd) {
		if (k < 0x3e400000) { sn = x; cs = 1.0; return true; }
		if (k < 0x3feb6000) { sn = l64p_do_sin(x, 0.0, TB); cs = l64p_do_cos(x, 0.0, TB); return true; }
		const double y = L64_HP0 - l64_abs(x);
		const double a = y + L64_HP1;
		const double da = (y - a) + L64_HP1;
		sn = l64_copysign(l64p_do_cos(a, da, TB), x);
		cs = l64p_do_sin(a, da, TB);
		return true;
	}
	if (k < 0x419921fb) {
		double a, da;
		const int n = l64p_reduce(x, a, da);
		sn = l64p_do_sincos(a, da, n, TB);
		cs = l64p_do_sincos(a, da, n + 1, TB);
		return true;
	}
	return false;
}
MIPT_L64 bool mipt_sincos64_main(double x, double& sn, double& cs) { return mipt_sincos64_main(x, sn, cs, l64_tables()); }

// ---------------------------------------------------------------- acos (e_asin.c, IBM Accurate Mathematical Library; glibc 2.35 = after
// the multi-precision fall-backs were removed: every range returns its first estimate)
// __ieee754_acos_fma at 0x77960 of the installed libm: the interval [0.125, 0.96875) is cut into pieces with a Taylor
// expansion of asin around the piece's x0 stored in `asncs` (rows of 11 .. 15 doubles by range), acos = pi/2 -+ asin;
// |x| < 0.125 is a polynomial, |x| >= 0.96875 goes through 2 asin(sqrt((1 - |x|) / 2)) with a table-started square root.
#define L64_AC_F1 0x1.55555555554f9p-3
#define L64_AC_F2 0x1.333333336127dp-4
#define L64_AC_F3 0x1.6db6dae42c0e4p-5
#define L64_AC_F4 0x1.f1c7e04f4ad99p-6
#define L64_AC_F5 0x1.6e442c822d419p-6
#define L64_AC_F6 0x1.292d80f453c72p-6
#define L64_AC_RT0 0x1.fffffffecc1ddp-1
#define L64_AC_RT1 0x1.fffffff757304p-2
#define L64_AC_RT2 0x1.800496769c91ap-2
#define L64_AC_RT3 0x1.4006318d1dab9p-2
#define L64_PI 0x1.921fb54442d18p+1
#define L64_PI_LO 0x1.1a62633145c07p-53
#define l64_asn(i) l64_dbl(TB.asncs[i])
// one piece of [0.125, 0.96875): row n of `stride` doubles = {x0, c1, c2 .. c(stride-5), c_xx, asin(x0) tail, asin(x0)}
MIPT_L64 double l64_acos_piece(double x, bool positive, int n, int stride, const L64Tables& TB) {
	const double xx = (positive ? x : -x) - l64_asn(n);
	double p = l64_asn(n + stride - 5);
	for (int j = stride - 6; j >= 2; j--) p = l64_fma(xx, p, l64_asn(n + j));
	p = l64_fma(xx * xx, p, l64_asn(n + stride - 4));
	const double t = l64_fma(xx, l64_asn(n + 1), p);
	const double y = l64_asn(n + stride - 3);
	if (positive) return (L64_HP1 - t) + (L64_HP0 - y);
	return (t + L64_HP1) + (y + L64_HP0);
}
MIPT_L64 double mipt_acos64(double x, const L64Tables& TB) {
	const uint64_t bits = l64_bits(x);
	const int32_t m = (int32_t)(bits >> 32);
	const int32_t k = m & 0x7fffffff;
	const uint32_t lx = (uint32_t)bits;
	if (k <= 0x3c87ffff) return L64_HP0;                                    // |x| < 2^-55
	if (k <= 0x3fbfffff) {                                                  // |x| < 0.125
		const double x2 = x * x;
		double p = l64_fma(x2, L64_AC_F6, L64_AC_F5);
		p = l64_fma(x2, p, L64_AC_F4); p = l64_fma(x2, p, L64_AC_F3); p = l64_fma(x2, p, L64_AC_F2); p = l64_fma(x2, p, L64_AC_F1);
		const double r = L64_HP0 - x;
		const double c = ((L64_HP0 - r) - x) + L64_HP1;
		const double cor = l64_fma(-p, x * x2, c);
		return r + cor;
	}
	const bool pos = m > 0;
	if (k <= 0x3feeffff) {                                                  // [0.125, 0.96875): one piece of one of five ranges
		// (row and row length picked with selects, ONE evaluation: lanes of a wave that fall into different ranges do not
		// run five copies of the chain one after the other)
		int n, stride;
		if (k <= 0x3fcfffff) { n = 11 * ((k >> 15) & 0x1f); stride = 11; }              // < 0.25
		else if (k <= 0x3fdfffff) { n = 11 * ((k >> 14) & 0x3f) + 352; stride = 11; }   // < 0.5
		else if (k <= 0x3fe7ffff) { n = 12 * ((k >> 13) & 0x7f) + 1056; stride = 12; }  // < 0.75
		else if (k <= 0x3fed7fff) { n = 13 * ((k >> 13) & 0x7f) + 992; stride = 13; }   // < 0.921875
		else if (k <= 0x3fee7fff) { n = 14 * ((k >> 13) & 0x7f) + 884; stride = 14; }   // < 0.953125
		else { n = 15 * ((k >> 13) & 0x7f) + 768; stride = 15; }                        // < 0.96875
		return l64_acos_piece(x, pos, n, stride, TB);
	}
	if (k <= 0x3fefffff) {                                                  // < 1
		const double z = (pos ? 1.0 - x : x + 1.0) * 0.5;
		const uint64_t zb = l64_bits(z);
		double t = l64_dbl(TB.inroot[(zb >> 46) & 0x7f]) * l64_dbl((uint64_t)(1023 + (0x1ff - (int)(zb >> 53))) << 52);   // inroot[] * powtwo[]
		const double r = l64_fma(-(t * t), z, 1.0);
		double q = l64_fma(r, L64_AC_RT3, L64_AC_RT2);
		q = l64_fma(r, q, L64_AC_RT1); q = l64_fma(r, q, L64_AC_RT0);
		t = q * t;
		const double c = z * t;
		const double e = l64_fma(-c, t * 0.5, 1.5);
		const double y0 = l64_fma(c, 0x1.0p+27, c);
		const double y = l64_fma(-0x1.0p+27, c, y0);
		const double den = l64_fma(e, c, y);
		const double cc = l64_fma(-y, y, z) / den;
		double p = l64_fma(z, L64_AC_F6, L64_AC_F5);
		p = l64_fma(z, p, L64_AC_F4); p = l64_fma(z, p, L64_AC_F3); p = l64_fma(z, p, L64_AC_F2); p = l64_fma(z, p, L64_AC_F1);
		const double w = (p * z) * (y + cc);
		if (m < 0) { const double s = ((L64_HP1 - cc) - w) + (L64_HP0 - y); return s + s; }
		const double s = (cc + w) + y;
		return s + s;
	}
	if (k == 0x3ff00000 && lx == 0) return pos ? 0.0 : L64_PI;              // +-1
	if (k > 0x7ff00000 || (k == 0x7ff00000 && lx != 0)) return x + x;      // NaN
	const double u = x - x;                                                 // |x| > 1 (or infinite): invalid
	return u / u;
}
#undef l64_asn
MIPT_L64 double mipt_acos64(double x) { return mipt_acos64(x, l64_tables()); }

// ---------------------------------------------------------------- atan2 (e_atan2.c, same library, same state: first estimates only)
// __ieee754_atan2_fma at 0x78060: u = min(|x|,|y|) / max(|x|,|y|) with its rounding error du (one fused multiply), then
// atan(u + du) from a degree-13 polynomial for u < 1/16 or from the Taylor expansion around the nearest of 241 points
// (`cij`), combined with 0, pi/2 or pi in two parts.  The rounding-mode bracket of the original is a no-op here (the
// reference never leaves round-to-nearest).
#define L64_AT_D3 (-0x1.5555555555555p-2)
#define L64_AT_D5 0x1.99999999997fdp-3
#define L64_AT_D7 (-0x1.24924923f7603p-3)
#define L64_AT_D9 0x1.c71c6e5129a3bp-4
#define L64_AT_D11 (-0x1.7458022b13c25p-4)
#define L64_AT_D13 0x1.375f08b31cbcep-4
MIPT_L64 double l64_at_poly(double v) {          // d3 + v (d5 + v (d7 + v (d9 + v (d11 + v d13))))
	double p = l64_fma(v, L64_AT_D13, L64_AT_D11);
	p = l64_fma(v, p, L64_AT_D9); p = l64_fma(v, p, L64_AT_D7); p = l64_fma(v, p, L64_AT_D5);
	return l64_fma(v, p, L64_AT_D3);
}
MIPT_L64 const uint64_t* l64_at_row(double u, const L64Tables& TB) {  // i = (TWO52 + 256 u) - TWO52 (round to nearest even), row i - 16
	const double r = l64_fma(u, 256.0, 0x1.0p+52) - 0x1.0p+52;
	return TB.cij + 7 * ((int)r - 16);
}
MIPT_L64 double l64_at_tail(const uint64_t* row, double v) {   // c2 + v (c3 + v (c4 + v (c5 + v c6)))
	double p = l64_fma(v, l64_dbl(row[6]), l64_dbl(row[5]));
	p = l64_fma(v, p, l64_dbl(row[4])); p = l64_fma(v, p, l64_dbl(row[3]));
	return l64_fma(v, p, l64_dbl(row[2]));
}
MIPT_L64 double mipt_atan264(double y, double x, const L64Tables& TB) {
	const uint64_t xb = l64_bits(x), yb = l64_bits(y);
	const int32_t ux = (int32_t)(xb >> 32), uy = (int32_t)(yb >> 32);
	const uint32_t dx = (uint32_t)xb, dy = (uint32_t)yb;
	const double mhpi = -L64_HP0, mopi = -L64_PI;
	const bool x_inf = (ux & 0x7ff00000) == 0x7ff00000, y_inf = (uy & 0x7ff00000) == 0x7ff00000;
	if (x_inf && (((ux & 0x000fffff) | dx) != 0)) return x + y;            // x is NaN
	if (y_inf && (((uy & 0x000fffff) | dy) != 0)) return y + y;            // y is NaN
	if (dy == 0 && uy == 0) return ux < 0 ? L64_PI : 0.0;                   // y = +0
	if (dy == 0 && (uint32_t)uy == 0x80000000u) return ux < 0 ? mopi : -0.0;   // y = -0
	if (x == 0.0) return uy < 0 ? mhpi : L64_HP0;
	if (x_inf) {
		if (ux > 0) return y_inf ? (uy < 0 ? -0x1.921fb54442d18p-1 : 0x1.921fb54442d18p-1) : (uy < 0 ? -0.0 : 0.0);       // x = +inf
		return y_inf ? (uy < 0 ? -0x1.2d97c7f3321d2p+1 : 0x1.2d97c7f3321d2p+1) : (uy < 0 ? mopi : L64_PI);                 // x = -inf
	}
	if (y_inf) return uy < 0 ? mhpi : L64_HP0;
	double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
	const int32_t de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
	if (de > 0x038fffff) return (0.0 < y) ? L64_HP0 : mhpi;                 // |y/x| > 2^57
	if (de < (int32_t)0xfc700001) {                                         // |y/x| < 2^-57
		if (!(x > 0.0)) return (0.0 < y) ? L64_PI : mopi;
		return l64_copysign(ay / ax, y);
	}
	if (ax < 0x1.0p-500 || ay < 0x1.0p-500) { ax *= 0x1.0p+500; ay *= 0x1.0p+500; }
	if (ax > 0x1.0p+500 || ay > 0x1.0p+500) { ax *= 0x1.0p-500; ay *= 0x1.0p-500; }
	// u = smaller / larger magnitude, du = its rounding error
	const bool y_smaller = ax > ay;
	const double num = y_smaller ? ay : ax, den = y_smaller ? ax : ay;
	const double u = num / den;
	const double v = den * u, vv = l64_fma(den, u, -v);
	const double du = ((num - v) - vv) / den;
	const bool small = 0.0625 > u;
	const double v2 = u * u;
	double z;
	if (y_smaller && x > 0.0) {                                             // (i) atan(ay / ax)
		if (small) z = u + l64_fma(u * v2, l64_at_poly(v2), du);
		else {
			const uint64_t* row = l64_at_row(u, TB);
			const double t3 = u - l64_dbl(row[0]);
			const double w = du + t3;
			const double dw = (l64_abs(t3) > l64_abs(du)) ? (t3 - w) + du : (du - w) + t3;
			const double t2 = l64_dbl(row[2]);
			double p = l64_fma(w, l64_dbl(row[6]), l64_dbl(row[5]));
			p = l64_fma(w, p, l64_dbl(row[4])); p = l64_fma(w, p, l64_dbl(row[3]));
			p = (w * w) * p;
			p = l64_fma(dw, t2, p);
			z = l64_fma(w, t2, p) + l64_dbl(row[1]);
		}
		return l64_copysign(z, y);
	}
	// (ii) x > 0: pi/2 - atan(ax / ay)   (iii) x < 0, |y| > |x|: pi/2 + atan(ax / ay)   (iv) x < 0: pi - atan(ay / ax).
	// One evaluation for the three: C -+ atan(u) with a - b written as a + (-b) (the same operation), so that lanes of a wave
	// in different quadrants share the chain.
	const bool third = !(x > 0.0) && ay > ax;
	const double C = (x > 0.0 || third) ? L64_HP0 : L64_PI, C_lo = (x > 0.0 || third) ? L64_HP1 : L64_PI_LO;
	if (small) {
		const double zz = (u * v2) * l64_at_poly(v2);
		const double q = third ? u : -u;
		const double t2 = C + q;
		const double cor = (C > l64_abs(u)) ? (C - t2) + q : (q - t2) + C;
		z = (((cor + C_lo) + (third ? du : -du)) + (third ? zz : -zz)) + t2;
	} else {
		const uint64_t* row = l64_at_row(u, TB);
		const double w = (u - l64_dbl(row[0])) + du;
		const double zz = l64_fma(third ? w : -w, l64_at_tail(row, w), C_lo);
		z = (C + (third ? l64_dbl(row[1]) : -l64_dbl(row[1]))) + zz;
	}
	return l64_copysign(z, y);
}
MIPT_L64 double mipt_atan264(double y, double x) { return mipt_atan264(y, x, l64_tables()); }
