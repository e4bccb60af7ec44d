// Compares the product's sinf/cosf (pathtracer_amd/csrc/mipt_sincos.h, the same source the HIP
// kernels compile) with the host libm on EVERY float in [0, hi].  Prints the mismatch counts.
// Build: g++ -O2 -fopenmp -ffp-contract=off sincos_check.cpp -lm
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../pathtracer_amd/csrc/mipt_sincos.h"

int main(int argc, char** argv) {
	float hi = argc > 1 ? (float)atof(argv[1]) : 6.5f;
	uint32_t uhi; memcpy(&uhi, &hi, 4);
	long bad_s = 0, bad_c = 0;
#pragma omp parallel for reduction(+ : bad_s, bad_c) schedule(static)
	for (uint32_t u = 0; u <= uhi; u++) {
		float x; memcpy(&x, &u, 4);
		float a = sinf(x), b = mipt_sincosf<false>(x);
		if (memcmp(&a, &b, 4)) bad_s++;
		a = cosf(x); b = mipt_sincosf<true>(x);
		if (memcmp(&a, &b, 4)) bad_c++;
		float ps, pc;                              // the fused form (both of one argument, each polynomial once)
		mipt_sincosf_pair(x, ps, pc);
		a = sinf(x); if (memcmp(&a, &ps, 4)) bad_s++;
		a = cosf(x); if (memcmp(&a, &pc, 4)) bad_c++;
	}
	printf("%u %ld %ld\n", uhi + 1, bad_s, bad_c);
	return 0;
}
