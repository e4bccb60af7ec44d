// Compares csrc/mipt_invtrig.h (the source the HIP kernels compile) with the host libm: acosf on every float in
// [-1, 1]; atanf on every float with |x| < 2^26 taken at a stride of 7 over the bit patterns plus all of [0, 4]; atan2f
// on N random pairs (uniform in [-1,1]^2, the env-map lookup's domain, and log-uniform magnitudes with random signs) and
// on the axes / equal-magnitude special points.  Prints "<evaluations> <acos bad> <atan bad> <atan2 bad>".
// Build: g++ -O2 -fopenmp -ffp-contract=off invtrig_check.cpp -lm
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include "../../pathtracer_amd/csrc/mipt_invtrig.h"

static inline uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline bool same(float a, float b) { return !memcmp(&a, &b, 4) || (a != a && b != b); }

int main(int argc, char** argv) {
	long nrand = argc > 1 ? atol(argv[1]) : 200000000L;
	long bad_acos = 0, bad_atan = 0, bad_atan2 = 0, total = 0;
#pragma omp parallel for reduction(+ : bad_acos, total) schedule(static)
	for (uint32_t u = 0; u <= 0x3f800000u; u++) {
		float x; memcpy(&x, &u, 4);
		if (!same(acosf(x), mipt_acosf(x))) bad_acos++;
		if (!same(acosf(-x), mipt_acosf(-x))) bad_acos++;
		total += 2;
	}
#pragma omp parallel for reduction(+ : bad_atan, total) schedule(static)
	for (uint32_t u = 0; u <= 0x4d000000u; u++) {
		if (u > 0x40800000u && (u % 7u)) continue;
		float x; memcpy(&x, &u, 4);
		if (!same(atanf(x), mipt_atanf(x))) bad_atan++;
		if (!same(atanf(-x), mipt_atanf(-x))) bad_atan++;
		total += 2;
	}
#pragma omp parallel reduction(+ : bad_atan2, total)
	{
		uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(1 + omp_get_thread_num());
		const long per = nrand / omp_get_num_threads();
		for (long n = 0; n < per; n++) {
			double u1 = (rng(s) >> 11) * 0x1p-53, u2 = (rng(s) >> 11) * 0x1p-53;
			float y, x;
			if (n & 1) { y = (float)(2 * u1 - 1); x = (float)(2 * u2 - 1); }
			else { y = (float)(exp(-70 + 140 * u1) * ((rng(s) & 1) ? 1 : -1)); x = (float)(exp(-70 + 140 * u2) * ((rng(s) & 1) ? 1 : -1)); }
			if (!same(atan2f(y, x), mipt_atan2f(y, x))) bad_atan2++;
			total++;
		}
	}
	const float sp[] = {0.f, -0.f, 1.f, -1.f, 0.5f, -0.5f, 1e-30f, -1e-30f, 1e30f, -1e30f, INFINITY, -INFINITY, 3.f, 1e-45f};
	for (float y : sp) for (float x : sp) { if (!same(atan2f(y, x), mipt_atan2f(y, x))) bad_atan2++; total++; }
	printf("%ld %ld %ld %ld\n", total, bad_acos, bad_atan, bad_atan2);
	return 0;
}
