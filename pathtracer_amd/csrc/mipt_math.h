// mipt_math.h — device math for the path-tracing hot path (gfx950).
//
// Parity rules: the reference's `Vector` is float but many expressions are
// evaluated in double (un-suffixed literals, M_PI).  Each helper below
// spells the evaluation order and the promotion points of the reference expression it cites.
// The translation unit is compiled with -ffp-contract=off: IEEE + - * / sqrt are correctly
// rounded on gfx950 exactly as on x86-64 SSE, so those expressions are bit-identical.
// Transcendentals: sinf / cosf / powf / acosf / atanf / atan2f and the double-precision exp / pow / sincos are
// re-implemented with the algorithms of the host libm the reference links against (mipt_sincos.h, mipt_powf.h,
// mipt_invtrig.h, mipt_libm64.h; each verified against libm on billions of inputs by tests/native/), so that direction
// sampling, the Phong lobe and its fp64 sampling frame, the Fresnel term, the subsurface weight and the environment-map
// lookup are bit-identical too.  Only the fp64 acos / atan2 of the MERL half / difference-angle transform still go
// through the ROCm device library; they feed nothing but table indices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mipt_sincos.h"
#include "mipt_powf.h"
#include "mipt_invtrig.h"
#include "mipt_libm64.h"

#define MIPT_DEV __device__ __forceinline__

#define MIPT_PI 3.14159265358979323846       /* M_PI (double) */
#define MIPT_TWO_PI_TRUNC 6.28318530718      /* Vector.h:16-18 M_TWO_PI */

struct f3 { float x, y, z; };

MIPT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MIPT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }            // Vector.h:491
MIPT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }            // Vector.h:495
MIPT_DEV f3 operator*(float a, f3 b) { return mk3(a * b.x, a * b.y, a * b.z); }               // Vector.h:499
MIPT_DEV f3 operator*(f3 b, float a) { return mk3(a * b.x, a * b.y, a * b.z); }               // Vector.h:503 (a*b[k])
MIPT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }            // Vector.h:519
MIPT_DEV f3 operator/(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }               // Vector.h:507
MIPT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }                                 // Vector.h:515
MIPT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                  // Vector.h:545
MIPT_DEV float norm2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                      // Vector.h:366
MIPT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }  // Vector.h:562
MIPT_DEV float sqr(float x) { return x * x; }

// Vector.h:294-309 invSqRoot: Quake rsqrt + 2 Newton steps with the 32-bit integer semantics of
// the reference's native LLP64 toolchain (see oracle/Makefile fix 4).
MIPT_DEV float inv_sq_root(float n) {
	int i = __float_as_int(n);
	i = 0x5f3759df - (i >> 1);
	float y = __int_as_float(i);
	y = y * (1.5F - ((n * 0.5F) * y * y));
	y = y * (1.5F - ((n * 0.5F) * y * y));
	return y;
}
MIPT_DEV f3 normalize(f3 a) { float n = sqrtf(norm2(a)); return mk3(a.x / n, a.y / n, a.z / n); }        // Vector.h:369
MIPT_DEV f3 fast_normalize(f3 a) { float inv = inv_sq_root(norm2(a)); return mk3(a.x * inv, a.y * inv, a.z * inv); }  // Vector.h:376
MIPT_DEV f3 reflect(f3 d, f3 N) { return d - (2.f * dot(d, N)) * N; }                                            // Vector.h:388

// ---------------------------------------------------------------- pcg32 (pcg_random.hpp)
// setseq_xsh_rr_64_32, default increment, seed ctor state = (seed+inc)*mult+inc, output on the
// previous state (pcg_random.hpp:157-158, 484-486, 1663, 1866).
#define MIPT_PCG_MULT 6364136223846793005ULL
#define MIPT_PCG_INC 1442695040888963407ULL
MIPT_DEV uint64_t pcg_seed(uint64_t seed) { return (seed + MIPT_PCG_INC) * MIPT_PCG_MULT + MIPT_PCG_INC; }
MIPT_DEV uint32_t pcg_next(uint64_t& state) {
	uint64_t old = state;
	state = old * MIPT_PCG_MULT + MIPT_PCG_INC;
	uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xs >> rot) | (xs << ((0u - rot) & 31u));
}
// the engine after four draws (an LCG: four steps are one multiply-add with M^4 and INC (M^3 + M^2 + M + 1), modulo 2^64)
MIPT_DEV uint64_t pcg_skip4(uint64_t state) {
	constexpr uint64_t M = MIPT_PCG_MULT, I = MIPT_PCG_INC, M2 = M * M, M4 = M2 * M2, C4 = I * (M2 * M + M2 + M + 1ULL);
	return state * M4 + C4;
}
// engine()*invmax, invmax = 1.f/engine.max() = 2^-32 (Raytracer.h:28); may return exactly 1.0f
MIPT_DEV float pcg_uniform(uint64_t& state) { return (float)pcg_next(state) * 2.3283064365386963e-10f; }

// ---------------------------------------------------------------- sinf / cosf
// Bit-exact with the host libm (see mipt_sincos.h; checked on every float of the range the path uses).
MIPT_DEV float pt_sinf(float y) { return mipt_sincosf<false>(y); }
MIPT_DEV float pt_cosf(float y) { return mipt_sincosf<true>(y); }
MIPT_DEV void pt_sincosf(float y, float& s, float& c) { mipt_sincosf_pair(y, s, c); }

// double-precision exp / pow / sincos of the host libm (mipt_libm64.h; 1.6 G arguments checked against libm by
// tests/native/libm64_check.cpp).  Arguments outside the restated ranges (never reached by the path) go to the device library.
__device__ __attribute__((noinline)) void sincos64_special(double a, double& s, double& c) { s = sin(a); c = cos(a); }
__device__ __attribute__((noinline)) double pow64_special(double x, double y) { return pow(x, y); }
MIPT_DEV void pt_sincos64(double a, double& s, double& c) { if (!mipt_sincos64_main(a, s, c)) sincos64_special(a, s, c); }
MIPT_DEV void pt_sincos64(double a, double& s, double& c, const L64Tables& TB) { if (!mipt_sincos64_main(a, s, c, TB)) sincos64_special(a, s, c); }
MIPT_DEV double pt_pow64(double x, double y) { double r; if (mipt_pow64_main(x, y, r)) return r; return pow64_special(x, y); }
MIPT_DEV double pt_exp64(double x) { return mipt_exp64(x); }
MIPT_DEV double pt_acos64(double x) { return mipt_acos64(x); }
MIPT_DEV double pt_atan264(double y, double x) { return mipt_atan264(y, x); }
MIPT_DEV double pt_acos64(double x, const L64Tables& TB) { return mipt_acos64(x, TB); }
MIPT_DEV double pt_atan264(double y, double x, const L64Tables& TB) { return mipt_atan264(y, x, TB); }

// powf: the host libm's algorithm, bit for bit (mipt_powf.h), for positive finite x and finite non-zero y; the exact
// special values (pow(x,0) = 1, pow(1,y) = 1, zero / inf / NaN / negative bases) come from the device library.
__device__ __attribute__((noinline)) float powf_special(float x, float y) { return powf(x, y); }
// powf_general is a LEAF (it hands the arguments it does not cover back as NaN — its own results are never NaN: a positive finite
// base and a finite exponent give a number, an infinity or zero — and the inlined caller takes them to powf_special): a function that
// calls another one saves its return address through a vector register in scratch, two vector-memory instructions per call, six
// calls per glossy vertex.
__device__ __attribute__((noinline)) float powf_general(float x, float y) {
	float r;
	if (mipt_powf_main(x, y, r)) return r;
	return __int_as_float(0x7fc00000);
}
MIPT_DEV float pt_powf(float x, float y) {
	if (y == 0.f) return 1.f;
	if (x == 1.f) return 1.f;
	const float r = powf_general(x, y);
	if (r == r) return r;
	return powf_special(x, y);
}

// Raytracer.cpp:1294-1299 fast_exp (Schraudolph, on a double)
MIPT_DEV double fast_exp(double y) {
	int hi = (int)(1512775 * y + 1072632447);
	return __hiloint2double(hi, 0);
}

// Texture::wrap (BRDF.h:270-275)
MIPT_DEV float tex_wrap(float u) { u -= (float)(int)u; if (u < 0) u += 1; return u; }
