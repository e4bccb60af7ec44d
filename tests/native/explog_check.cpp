// Compares csrc/mipt_explog.h (the source the HIP kernels compile) with the host libm: expf and logf on EVERY float,
// tanf on every float with |x| < 120.  Prints "<evaluations> <expf bad> <logf bad> <tanf bad>".
// Build: g++ -O2 -fopenmp -ffp-contract=off explog_check.cpp -lm     (fma() must map to the hardware instruction: -mfma)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include "../../pathtracer_amd/csrc/mipt_explog.h"

static inline bool same(float a, float b) { return !memcmp(&a, &b, 4) || (a != a && b != b); }

int main(int argc, char** argv) {
	const unsigned stride = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
	long bad_exp = 0, bad_log = 0, bad_tan = 0, total = 0;
	unsigned first_exp = 0, first_log = 0, first_tan = 0;
#pragma omp parallel for reduction(+ : bad_exp, bad_log, bad_tan, total) schedule(static)
	for (long long v = 0; v <= 0xffffffffLL; v += stride) {
		const uint32_t u = (uint32_t)v;
		float x; memcpy(&x, &u, 4);
		if (!same(expf(x), mipt_expf(x))) { bad_exp++; first_exp = u; }
		if (!same(logf(x), mipt_logf(x))) { bad_log++; first_log = u; }
		total += 2;
		float t;
		if (mipt_tanf_main(x, t)) { if (!same(tanf(x), t)) { bad_tan++; first_tan = u; } total++; }
	}
	printf("%ld %ld %ld %ld\n", total, bad_exp, bad_log, bad_tan);
	if (bad_exp) fprintf(stderr, "expf mismatch e.g. at 0x%08x\n", first_exp);
	if (bad_log) fprintf(stderr, "logf mismatch e.g. at 0x%08x\n", first_log);
	if (bad_tan) fprintf(stderr, "tanf mismatch e.g. at 0x%08x\n", first_tan);
	return 0;
}
