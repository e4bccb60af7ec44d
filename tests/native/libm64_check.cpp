// Compares csrc/mipt_libm64.h (the source the HIP kernels compile) with the host libm's double-precision exp / pow / sincos / acos / atan2.
// Prints "<evaluations> <exp bad> <pow bad> <sincos bad> <pairs where libm's sincos differs from its own sin / cos> <acos bad> <atan2 bad>".  The host must have FMA + AVX2 (glibc then runs the same
// __*_fma variants the header restates).
// Build: g++ -O2 -fopenmp -ffp-contract=off -mfma tests/native/libm64_check.cpp -lm
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include "../../pathtracer_amd/csrc/mipt_libm64.h"

static inline bool same(double a, double b) { return !memcmp(&a, &b, 8) || (a != a && b != b); }
static inline uint64_t rnd(uint64_t& s) { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
static inline double u01(uint64_t& s) { return (double)(rnd(s) >> 11) * 0x1p-53; }

int main(int argc, char** argv) {
	const long n = argc > 1 ? atol(argv[1]) : 100000000L;
	long bad_exp = 0, bad_pow = 0, bad_sc = 0, sc_differs = 0, total = 0, bad_acos = 0, bad_atan2 = 0;
	double fe = 0, fpx = 0, fpy = 0, fsc = 0, fac = 0, fay = 0, fax = 0;
	double (*volatile p_acos)(double) = acos; double (*volatile p_atan2)(double, double) = atan2;
	double (*volatile p_sin)(double) = sin; double (*volatile p_cos)(double) = cos; void (*volatile p_sincos)(double, double*, double*) = sincos;
	{	// special values: every pair of them for atan2, each for acos
		const double sp[] = {0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.5, -1.5, 0x1p-1074, -0x1p-1074, 0x1p-1022, -0x1p-1022, 0x1.fffffffffffffp+1023, -0x1.fffffffffffffp+1023,
		                     __builtin_inf(), -__builtin_inf(), __builtin_nan(""), 0x1p-600, -0x1p-600, 0x1p+600, -0x1p+600, 0.0625, 0x1.fffffffffffffp-5, 0x1.0000000000001p-4, 0x1.fffffffffffffp-1, -0x1.fffffffffffffp-1,
		                     0.125, 0x1.fffffffffffffp-4, 0.96875, 0x1.effffffffffffp-1, 0x1p-55, 0x1.7ffffffffffffp-55, 0x1p-56};
		const int ns = (int)(sizeof(sp) / sizeof(sp[0]));
		for (int i = 0; i < ns; i++) {
			if (!same(p_acos(sp[i]), mipt_acos64(sp[i]))) { bad_acos++; fac = sp[i]; }
			total++;
			for (int j = 0; j < ns; j++) { if (!same(p_atan2(sp[i], sp[j]), mipt_atan264(sp[i], sp[j]))) { bad_atan2++; fay = sp[i]; fax = sp[j]; } total++; }
		}
	}
#pragma omp parallel reduction(+ : bad_exp, bad_pow, bad_sc, sc_differs, total, bad_acos, bad_atan2)
	{
		uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(omp_get_thread_num() + 1);
#pragma omp for schedule(static)
		for (long it = 0; it < n; it++) {
			// ---- exp: the whole finite range incl. the subnormal results, small arguments, the path's domain (-(float)/4.5)
			double xs[4];
			xs[0] = -750.0 + 1460.0 * u01(s);
			xs[1] = (u01(s) - 0.5) * exp2(-60.0 * u01(s));
			xs[2] = (double)(-(float)(120.0 * u01(s))) / (2. * (double)1.5f * (double)1.5f);
			xs[3] = -745.2 + 40.0 * u01(s);
			for (double x : xs) { if (!same(exp(x), mipt_exp64(x))) { bad_exp++; fe = x; } total++; }
			// ---- pow: the path's domain (float base in (0, 1], exponent 1 / (ne + 1)) and general positive bases / exponents
			{
				const double x1 = (double)(float)u01(s), y1 = 1. / (double)((float)(1000.0 * u01(s) * u01(s)) + 1.f);
				const double x2 = exp2(600.0 * (u01(s) - 0.5)), y2 = (u01(s) - 0.5) * 4.0;
				const double x3 = 1.0 + (u01(s) - 0.5) * exp2(-40.0 * u01(s)), y3 = exp2(40.0 * u01(s)) * (u01(s) - 0.5);
				const double px[3] = {x1, x2, x3}, py[3] = {y1, y2, y3};
				for (int k = 0; k < 3; k++) {
					double r;
					if (mipt_pow64_main(px[k], py[k], r)) { if (!same(pow(px[k], py[k]), r)) { bad_pow++; fpx = px[k]; fpy = py[k]; } total++; }
				}
			}
			// ---- sin / cos: the path's domains (2 pi r with r a float in [0, 1]; float angles) and everything up to 1.05e8
			{
				double as[5];
				as[0] = 2 * 3.14159265358979323846 * (double)(float)u01(s);
				as[1] = (double)(float)(6.3 * u01(s));
				as[2] = (u01(s) - 0.5) * 16.0;
				as[3] = (u01(s) - 0.5) * exp2(28.0 * u01(s));
				as[4] = (u01(s) - 0.5) * exp2(-30.0 * u01(s));
				for (double a : as) {
					// (the libm calls go through volatile function pointers: with both sin(a) and cos(a) in sight GCC would call sincos() for the pair)
					double rs, rc;
					const double ls = p_sin(a), lc = p_cos(a);
					if (mipt_sincos64_main(a, rs, rc)) {
						double qs, qc;
						p_sincos(a, &qs, &qc);
						if (!same(qs, rs) || !same(qc, rc)) { bad_sc++; fsc = a; }
						if (!same(qs, ls) || !same(qc, lc)) sc_differs++;
						total += 2;
					}
				}
			}
			// ---- acos: [-1, 1] uniformly, towards 0, towards +-1, the path's domain (a component of a normalised double vector), outside
			{
				double cs[6];
				cs[0] = 2.0 * u01(s) - 1.0;
				cs[1] = (u01(s) - 0.5) * exp2(-60.0 * u01(s));
				cs[2] = (u01(s) < 0.5 ? -1.0 : 1.0) * (1.0 - exp2(-54.0 * u01(s)));
				{ const double a = u01(s) - 0.5, b = u01(s) - 0.5, c = u01(s) - 0.5; cs[3] = c / sqrt(a * a + b * b + c * c); }
				cs[4] = (u01(s) - 0.5) * 4.0;
				cs[5] = (double)(float)(2.0 * u01(s) - 1.0);
				for (double c : cs) { if (!same(p_acos(c), mipt_acos64(c))) { bad_acos++; fac = c; } total++; }
			}
			// ---- atan2: all quadrants, ratios on both sides of 1/16 and 1, extreme ratios, tiny and huge operands, zeros
			{
				double ys[6], xs2[6];
				const double ang = 6.283185307179586 * u01(s), rad = exp2(40.0 * (u01(s) - 0.5));
				ys[0] = rad * sin(ang); xs2[0] = rad * cos(ang);
				ys[1] = u01(s) - 0.5; xs2[1] = u01(s) - 0.5;
				ys[2] = (u01(s) - 0.5) * exp2(-70.0 * u01(s)); xs2[2] = (u01(s) - 0.5) * exp2(-70.0 * u01(s));
				ys[3] = (u01(s) - 0.5) * exp2(1000.0 * (u01(s) - 0.5)); xs2[3] = (u01(s) - 0.5) * exp2(1000.0 * (u01(s) - 0.5));
				ys[4] = (double)(float)(u01(s) - 0.5); xs2[4] = (double)(float)(u01(s) - 0.5);
				ys[5] = (it & 1) ? 0.0 : (u01(s) - 0.5); xs2[5] = (it & 2) ? -0.0 : (u01(s) - 0.5);
				for (int k = 0; k < 6; k++) { if (!same(p_atan2(ys[k], xs2[k]), mipt_atan264(ys[k], xs2[k]))) { bad_atan2++; fay = ys[k]; fax = xs2[k]; } total++; }
			}
		}
	}
	printf("%ld %ld %ld %ld %ld %ld %ld\n", total, bad_exp, bad_pow, bad_sc, sc_differs, bad_acos, bad_atan2);
	if (bad_acos) fprintf(stderr, "acos mismatch e.g. at %a: libm %a ours %a\n", fac, acos(fac), mipt_acos64(fac));
	if (bad_atan2) fprintf(stderr, "atan2 mismatch e.g. at (%a, %a): libm %a ours %a\n", fay, fax, atan2(fay, fax), mipt_atan264(fay, fax));
	if (bad_exp) fprintf(stderr, "exp mismatch e.g. at %a: libm %a ours %a\n", fe, exp(fe), mipt_exp64(fe));
	if (bad_pow) { double r = 0; mipt_pow64_main(fpx, fpy, r); fprintf(stderr, "pow mismatch e.g. at (%a, %a): libm %a ours %a\n", fpx, fpy, pow(fpx, fpy), r); }
	if (bad_sc) { double a = 0, b = 0, qs, qc; mipt_sincos64_main(fsc, a, b); sincos(fsc, &qs, &qc); fprintf(stderr, "sincos mismatch e.g. at %a: libm (%a, %a) ours (%a, %a)\n", fsc, qs, qc, a, b); }
	return (bad_exp || bad_pow || bad_sc || bad_acos || bad_atan2) ? 1 : 0;
}
