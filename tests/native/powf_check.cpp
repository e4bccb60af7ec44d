// Compares the product's powf (pathtracer_amd/csrc/mipt_powf.h, the same source the HIP kernels compile) with the
// host libm: every float x in (0, 2] for the exponents the path uses with a constant (5.f: Schlick; 2.2f), and N random
// (x, y) pairs with x log-uniform in [1e-30, 4] and y in [-300, 300].  Prints "<evaluations> <mismatches> <unhandled>".
// Build: g++ -O2 -fopenmp -ffp-contract=off powf_check.cpp -lm
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include "../../pathtracer_amd/csrc/mipt_powf.h"

static inline uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

int main(int argc, char** argv) {
	long nrand = argc > 1 ? atol(argv[1]) : 100000000L;
	int full = argc > 2 ? atoi(argv[2]) : 1;
	long bad = 0, unhandled = 0, total = 0;
	if (full) {
		const float ys[2] = {5.f, 2.2f};
		uint32_t uhi; float two = 2.f; memcpy(&uhi, &two, 4);
#pragma omp parallel for reduction(+ : bad, unhandled, total) schedule(static)
		for (uint32_t u = 1; u <= uhi; u++) {
			float x; memcpy(&x, &u, 4);
			for (int j = 0; j < 2; j++) {
				float a = powf(x, ys[j]), b;
				if (!mipt_powf_main(x, ys[j], b)) { unhandled++; continue; }
				if (memcmp(&a, &b, 4)) bad++;
				total++;
			}
		}
	}
#pragma omp parallel reduction(+ : bad, unhandled, total)
	{
		uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(1 + omp_get_thread_num());
		const long per = nrand / omp_get_num_threads();
		for (long n = 0; n < per; n++) {
			double u1 = (rng(s) >> 11) * 0x1p-53, u2 = (rng(s) >> 11) * 0x1p-53;
			float x = (float)exp(log(1e-30) + u1 * (log(4.0) - log(1e-30)));
			float y = (float)(-300.0 + 600.0 * u2);
			if (n & 1) { x = (float)(u1 * 1.0000001); y = (float)(u2 * 200.0); }        // the Phong lobe's domain
			float a = powf(x, y), b;
			if (!mipt_powf_main(x, y, b)) { unhandled++; continue; }
			if (memcmp(&a, &b, 4)) bad++;
			total++;
		}
	}
	printf("%ld %ld %ld\n", total, bad, unhandled);
	return 0;
}
