/* TEST INFRASTRUCTURE ONLY — see pt_oracle.h.
 *
 * Plain-C restatement of the reference's hot path.  Style is deliberately the reference's own
 * (array-of-structs scene, scalar loops, a per-ray explicit stack): this file is the CHECKER,
 * not the product.  All `file:line` citations are into the reference checkout.
 *
 * Floating-point discipline: the reference's `Vector` is float, but
 * un-suffixed literals and M_PI are double, so several expressions are evaluated in double and
 * narrowed on assignment.  Every such promotion point is reproduced here explicitly.  Compile
 * with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 *
 * Parity status: PINNED against the compiled reference (tests/test_oracle_vs_reference.py) and
 * the committed golden vectors (tests/golden/).
 */
#define _GNU_SOURCE
#include "pt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <omp.h>

#ifndef M_PI
#define M_PI 3.1415926535897932
#endif
#define O_TWO_PI 6.28318530718 /* Vector.h:16-18 (truncated literal, used by PhongBRDF::eval) */

/* ------------------------------------------------------------------ vectors (Vector.h:346-541) */
typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(float a, v3 b) { return V(a * b.x, a * b.y, a * b.z); }       /* a*V and V*a */
static inline v3 vdivs(v3 a, float b) { return V(a.x / b, a.y / b, a.z / b); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float vnorm2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float vget(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
static inline void vset(v3* a, int k, float v) { if (k == 0) a->x = v; else if (k == 1) a->y = v; else a->z = v; }
static inline float sqrf(float x) { return x * x; }
static v3 vmin3(v3 a, v3 b);
static v3 vmax3(v3 a, v3 b);
static inline double sqr_d(double x) { return x * x; }

/* Vector.h:294-309 invSqRoot.  The reference reads the float through a `long*`; the intended
 * (and, as compiled, observed) semantics are the classic 32-bit ones. */
static inline float inv_sq_root(float n) {
	float y = n;
	int32_t i;
	memcpy(&i, &y, 4);
	i = 0x5f3759df - (i >> 1);
	memcpy(&y, &i, 4);
	y = y * (1.5F - ((n * 0.5F) * y * y));
	y = y * (1.5F - ((n * 0.5F) * y * y));
	return y;
}
/* Vector.h:369-375 normalize (exact) and :376-382 fast_normalize */
static inline v3 vnormalize(v3 a) { float n = sqrtf(vnorm2(a)); return V(a.x / n, a.y / n, a.z / n); }
static inline v3 vfast_normalize(v3 a) { float inv = inv_sq_root(vnorm2(a)); return V(a.x * inv, a.y * inv, a.z * inv); }
/* Vector.h:388-391 reflect: *this - 2*dot(*this,N)*N */
static inline v3 vreflect(v3 d, v3 N) { return vsub(d, vscale(2.f * vdot(d, N), N)); }

/* ------------------------------------------------------------------ RNG (pcg_random.hpp) */
/* pcg32 = setseq_xsh_rr_64_32 (pcg_random.hpp:1663,1866); mult/inc :157-158; seed ctor
 * state = bump(seed + inc) (:484-486); output on the previous state (XSH-RR). */
#define PCG_MULT 6364136223846793005ULL
#define PCG_INC  1442695040888963407ULL
typedef struct { uint64_t state; } pcg32_t;
static inline void pcg_seed(pcg32_t* g, uint64_t seed) { g->state = (seed + PCG_INC) * PCG_MULT + PCG_INC; }
static inline uint32_t pcg_next(pcg32_t* g) {
	uint64_t old = g->state;
	g->state = old * PCG_MULT + PCG_INC;
	uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xs >> rot) | (xs << ((-rot) & 31u));
}
/* Raytracer.h:28 invmax = 1.f/engine.max() = 2^-32; u = engine()*invmax (can be exactly 1.0f) */
#define INVMAX 2.3283064365386963e-10f
static inline float pcg_uniform(pcg32_t* g) { return (float)pcg_next(g) * INVMAX; }

/* ------------------------------------------------------------------ scene structs */
typedef struct {           /* BRDF.h:7-20 MaterialValues */
	v3 shadingN, Kd, Ks, Ne, Ke, Ksub;
	int transp;
	float refr_index;
} o_mat;
static void mat_default(o_mat* m) { /* BRDF.h:9-16 */
	m->shadingN = V(0, 1, 0); m->Kd = V(0.5f, 0.5f, 0.5f); m->Ne = V(100, 100, 100);
	m->Ks = V(0, 0, 0); m->Ke = V(0, 0, 0); m->Ksub = V(0, 0, 0); m->transp = 0; m->refr_index = 0;
}

typedef struct {           /* BRDF.h:252-426 Texture (float RGB, nearest fetch) */
	v3 multiplier;
	size_t W, H;
	float* values;
} o_tex;
enum { T_KD = 0, T_KS = 1, T_NORMAL = 2, T_ALPHA = 3, T_NE = 4, T_TRANSP = 5, T_REFR = 6, T_KSUB = 7, T_NSLOTS = 8 };

typedef struct { int isleaf, fg, fd; v3 bmin, bmax; } o_node;          /* TriangleMesh.h:6-13 */
typedef struct {                                                        /* TriangleMesh.h:67-111 */
	v3 A, u, v, N; float m11, m12, m22, invdetm; float uvs[3][2]; v3 normals[3];
} o_tri;
typedef struct { int vtx[3], uv[3], n[3], group; } o_idx;             /* TriangleMesh.h:53-65 */

typedef struct {
	int nv, nn, nt, nf;
	v3 *vertices, *normals, *uvs;
	o_idx* indices;
	o_tri* soup;
	v3* tangent_soup;
	int* perm;
	o_node* nodes; int nnodes, cap_nodes;
	v3 root_min, root_max;      /* bvh.bbox */
	v3 bb_min, bb_max;          /* TriMesh::bbox */
	int interp_normals;
} o_mesh;

enum { OT_SPHERE = 1, OT_PLANE = 2, OT_TRIMESH = 0 };
typedef struct {
	int type;
	int miroir, flip_normals;
	int ghost;                       /* Object::ghost (Geometry.h:721): invisible, but receives shadows / reflects (compositing over a photo) */
	float scale; v3 max_translation; float mat_rotation[9]; v3 rotation_center;
	float trans[12], inv[12], rot[9];                /* Geometry.h:322-360 */
	int ntex[T_NSLOTS]; o_tex* tex[T_NSLOTS];
	const double* merl;          /* IsoMERLBRDF::data or NULL (PhongBRDF) */
	/* sphere */ v3 O; float R, R2; int has_envmap; unsigned char* envtex; int envW, envH;
	/* plane  */ v3 A, vecN;
	/* mesh   */ o_mesh* mesh;
} o_obj;

typedef struct { v3 origin, direction; } o_ray;

struct o_ctx {
	int W, H, nrays, nb_bounces;
	float sigma_filter; int filter_size, filter_total_width;
	float* filter_integral;
	v3 cam_pos, cam_dir, cam_up; float fov, focus, aperture;
	o_obj* objs; int nobj, cap_obj;
	float intensite_lumiere, envmap_intensity;
	int is_lenticular, lenticular_nb_images, lenticular_pixel_width; float lenticular_max_angle;   /* Camera, Vector.h:831-834 */
	float* background; int backgroundW, backgroundH;
	float fog_density, fog_absorption, fog_density_decay, fog_absorption_decay, phase_aniso;   /* Geometry.h:1371-1377 */
	int fog_type, fog_phase_type;   /* Scene::background (Geometry.h:1355-1366), top row first, x196964.699 */
	float* randomPerPixel; float* samples2d;
	v3 centerLight; float radiusLight, lightPower, lum_scale;
	int current_frame; float double_frustum_start_t;
};

/* ------------------------------------------------------------------ work counters */
static _Thread_local uint64_t tl_cnt[8];
static uint64_t g_cnt[8];
static void cnt_flush(void) {
	for (int k = 0; k < 8; k++) { if (tl_cnt[k]) { __atomic_fetch_add(&g_cnt[k], tl_cnt[k], __ATOMIC_RELAXED); tl_cnt[k] = 0; } }
}
/* Traversal event trace (diagnostic; tests/tools/sched_sim.py): when a buffer is installed, every TriMesh traversal of the
   calling thread appends 0xFF kind(0 closest / 1 any-hit) [0xFE = the root box rejected the ray], then one byte per node
   popped: an inner node = the number of stack entries that remain behind it (what a stack that only stores far children
   holds), a leaf = 0x80 | its triangle count; o_trace_mark() appends 0xFD (a new camera path). */
static _Thread_local uint8_t* tl_trace = NULL;
static _Thread_local size_t tl_trace_len = 0, tl_trace_cap = 0;
void o_trace_set(uint8_t* buf, size_t cap) { tl_trace = buf; tl_trace_cap = cap; tl_trace_len = 0; }
size_t o_trace_len(void) { return tl_trace_len; }
static inline void trace_byte(uint8_t b) { if (tl_trace) { if (tl_trace_len < tl_trace_cap) tl_trace[tl_trace_len] = b; tl_trace_len++; } }
void o_counters_reset(void) { cnt_flush(); memset(g_cnt, 0, sizeof g_cnt); }
void o_counters_get(uint64_t* out8) { cnt_flush(); memcpy(out8, g_cnt, sizeof g_cnt); }

/* ------------------------------------------------------------------ textures */
static float tex_wrap(float u) { u -= (int)u; if (u < 0) u += 1; return u; }   /* BRDF.h:270-275 */
static v3 tex_getVec(const o_tex* t, float u, float v) {                      /* BRDF.h:293-308 */
	if (t->W > 0) {
		int x = u * (t->W - 1);   /* float * size_t -> float, truncation */
		int y = v * (t->H - 1);
		int idx = (y * t->W + x) * 3;
		return V(t->values[idx] * t->multiplier.x, t->values[idx + 1] * t->multiplier.y, t->values[idx + 2] * t->multiplier.z);
	}
	return t->multiplier;
}
static int tex_getBool(const o_tex* t, float u, float v) {                    /* BRDF.h:335-346 */
	if (t->W > 0) {
		int x = u * (t->W - 1); int y = v * (t->H - 1); int idx = (y * t->W + x) * 3;
		float cr = t->values[idx] * t->multiplier.x;
		return cr < 0.5f;
	}
	return t->multiplier.x < 0.5f;
}
static float tex_getValRed(const o_tex* t, float u, float v) {                /* BRDF.h:379-391 */
	if (t->W > 0) {
		int x = u * (t->W - 1); int y = v * (t->H - 1); int idx = (y * t->W + x) * 3;
		return t->values[idx] * t->multiplier.x;
	}
	return t->multiplier.x;
}
static v3 tex_getNormal(const o_tex* t, float u, float v) {                   /* BRDF.h:347-357 */
	if (t->W > 0) {
		int x = u * (t->W - 1); int y = v * (t->H - 1); int idx = (y * t->W + x) * 3;
		return V(t->values[idx], t->values[idx + 1], t->values[idx + 2]);
	}
	return V(0.f, 0.f, 1.f);
}

/* Geometry.h:399-445 Object::queryMaterial */
static void query_material(const o_obj* o, int idx, float u, float v, o_mat* mat) {
	u = tex_wrap(u);
	v = tex_wrap(v);
	size_t sidx = (size_t)(long)idx;  /* int compared with size_t: negative -> huge */
	if (sidx >= (size_t)o->ntex[T_KD]) mat->Kd = V(1, 1, 1); else mat->Kd = tex_getVec(&o->tex[T_KD][idx], u, v);
	if (sidx >= (size_t)o->ntex[T_KS]) mat->Ks = V(0, 0, 0); else mat->Ks = tex_getVec(&o->tex[T_KS][idx], u, v);
	if (sidx >= (size_t)o->ntex[T_KSUB]) mat->Ksub = V(0, 0, 0); else mat->Ksub = tex_getVec(&o->tex[T_KSUB][idx], u, v);
	if (sidx >= (size_t)o->ntex[T_NE]) mat->Ne = V(1, 1, 1); else mat->Ne = tex_getVec(&o->tex[T_NE][idx], u, v);
	if (sidx >= (size_t)o->ntex[T_TRANSP]) mat->transp = 0; else mat->transp = tex_getBool(&o->tex[T_TRANSP][idx], u, v);
	if (sidx >= (size_t)o->ntex[T_REFR]) mat->refr_index = 1.3; else mat->refr_index = tex_getValRed(&o->tex[T_REFR][idx], u, v);
	mat->Ke = V(0, 0, 0);
}

static o_tex* obj_push_tex(o_obj* o, int slot, v3 mult) {
	o->tex[slot] = (o_tex*)realloc(o->tex[slot], sizeof(o_tex) * (o->ntex[slot] + 1));
	o_tex* t = &o->tex[slot][o->ntex[slot]++];
	t->multiplier = mult; t->W = 0; t->H = 0; t->values = NULL;
	return t;
}

/* ------------------------------------------------------------------ object transforms */
/* Geometry.h:322-360 Object::build_matrix (is_recording=false, no keyframes -> rotation =
 * mat_rotation, scale = scale, translation = max_translation; Geometry.h:281-320) */
static void build_matrix(o_obj* o) {
	const float* m = o->mat_rotation;
	float mt[9];
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mt[j * 3 + i] = m[i * 3 + j];  /* Vector.h:96-103 */
	float s = o->scale;
	v3 tr = o->max_translation;
	for (int i = 0; i < 3; i++) {
		v3 v2 = V(m[0 * 3 + i], m[1 * 3 + i], m[2 * 3 + i]);
		o->trans[0 * 4 + i] = v2.x * s; o->trans[1 * 4 + i] = v2.y * s; o->trans[2 * 4 + i] = v2.z * s;
		o->rot[0 * 3 + i] = v2.x; o->rot[1 * 3 + i] = v2.y; o->rot[2 * 3 + i] = v2.z;
		v2 = V(mt[0 * 3 + i], mt[1 * 3 + i], mt[2 * 3 + i]);
		o->inv[0 * 4 + i] = v2.x / s; o->inv[1 * 4 + i] = v2.y / s; o->inv[2 * 4 + i] = v2.z / s;
	}
	/* Matrix33 * Vector (Vector.h:438-450): v = 0; v += mat[i*3+j]*b[j] */
	v3 b = vneg(o->rotation_center), v2;
	float r[3];
	for (int i = 0; i < 3; i++) { float v = 0; v += m[i * 3 + 0] * b.x; v += m[i * 3 + 1] * b.y; v += m[i * 3 + 2] * b.z; r[i] = v; }
	v2 = V(r[0], r[1], r[2]);
	o->trans[0 * 4 + 3] = v2.x * s + o->rotation_center.x + tr.x;
	o->trans[1 * 4 + 3] = v2.y * s + o->rotation_center.y + tr.y;
	o->trans[2 * 4 + 3] = v2.z * s + o->rotation_center.z + tr.z;
	b = vsub(vneg(o->rotation_center), tr);
	for (int i = 0; i < 3; i++) { float v = 0; v += mt[i * 3 + 0] * b.x; v += mt[i * 3 + 1] * b.y; v += mt[i * 3 + 2] * b.z; r[i] = v; }
	v2 = V(r[0], r[1], r[2]);
	o->inv[0 * 4 + 3] = v2.x / s + o->rotation_center.x;
	o->inv[1 * 4 + 3] = v2.y / s + o->rotation_center.y;
	o->inv[2 * 4 + 3] = v2.z / s + o->rotation_center.z;
}
/* Geometry.h:362-396 */
static v3 apply_transformation(const o_obj* o, v3 v) {
	const float* t = o->trans;
	return V(t[0] * v.x + t[1] * v.y + t[2] * v.z + t[3], t[4] * v.x + t[5] * v.y + t[6] * v.z + t[7], t[8] * v.x + t[9] * v.y + t[10] * v.z + t[11]);
}
static v3 apply_rotation(const o_obj* o, v3 v) {
	const float* r = o->rot;
	return V(r[0] * v.x + r[1] * v.y + r[2] * v.z, r[3] * v.x + r[4] * v.y + r[5] * v.z, r[6] * v.x + r[7] * v.y + r[8] * v.z);
}
static v3 apply_inverse_transformation(const o_obj* o, v3 v) {
	const float* t = o->inv;
	return V(t[0] * v.x + t[1] * v.y + t[2] * v.z + t[3], t[4] * v.x + t[5] * v.y + t[6] * v.z + t[7], t[8] * v.x + t[9] * v.y + t[10] * v.z + t[11]);
}
static v3 apply_inverse_rotation_scaling(const o_obj* o, v3 v) {
	const float* t = o->inv;
	return V(t[0] * v.x + t[1] * v.y + t[2] * v.z, t[4] * v.x + t[5] * v.y + t[6] * v.z, t[8] * v.x + t[9] * v.y + t[10] * v.z);
}

/* ------------------------------------------------------------------ AABB slab tests */
/* Geometry.h:114-142 BBoxT::intersection_invd */
static int box_invd(v3 bmin, v3 bmax, v3 o, v3 invd, const char signs[3], float* t) {
	float t_max;
	t_max = ((signs[0] ? bmax.x : bmin.x) - o.x) * invd.x;
	if (t_max < 0) return 0;
	*t = ((signs[0] ? bmin.x : bmax.x) - o.x) * invd.x;
	float t_min_y, t_max_y;
	t_max_y = ((signs[1] ? bmax.y : bmin.y) - o.y) * invd.y;
	if (t_max_y < 0) return 0;
	t_min_y = ((signs[1] ? bmin.y : bmax.y) - o.y) * invd.y;
	if (t_min_y > t_max || t_max_y < *t) return 0;
	if (t_min_y > *t) *t = t_min_y;
	if (t_max_y < t_max) t_max = t_max_y;
	float t_min_z, t_max_z;
	t_max_z = ((signs[2] ? bmax.z : bmin.z) - o.z) * invd.z;
	if (t_max_z < 0) return 0;
	t_min_z = ((signs[2] ? bmin.z : bmax.z) - o.z) * invd.z;
	if (*t > t_max_z || t_min_z > t_max) return 0;
	if (t_min_z > *t) *t = t_min_z;
	if (*t < 0) *t = 0;
	return 1;
}
/* Geometry.h:144-173 intersection_invd_positive_x */
static int box_invd_posx(v3 bmin, v3 bmax, v3 o, v3 invd, const char signs[3], float* t) {
	float t_max;
	t_max = (bmax.x - o.x);
	if (t_max < 0) return 0;
	t_max *= invd.x;
	*t = (bmin.x - o.x) * invd.x;
	float t_min_y, t_max_y;
	t_max_y = ((signs[1] ? bmax.y : bmin.y) - o.y) * invd.y;
	if (t_max_y < 0) return 0;
	t_min_y = ((signs[1] ? bmin.y : bmax.y) - o.y) * invd.y;
	if (t_min_y > t_max || t_max_y < *t) return 0;
	if (t_min_y > *t) *t = t_min_y;
	if (t_max_y < t_max) t_max = t_max_y;
	float t_min_z, t_max_z;
	t_max_z = ((signs[2] ? bmax.z : bmin.z) - o.z) * invd.z;
	if (t_max_z < 0) return 0;
	t_min_z = ((signs[2] ? bmin.z : bmax.z) - o.z) * invd.z;
	if (*t > t_max_z || t_min_z > t_max) return 0;
	if (t_min_z > *t) *t = t_min_z;
	if (*t < 0) *t = 0;
	return 1;
}
/* Geometry.h:175-204 intersection_invd_negative_x */
static int box_invd_negx(v3 bmin, v3 bmax, v3 o, v3 invd, const char signs[3], float* t) {
	float t_max;
	t_max = (bmin.x - o.x);
	if (t_max > 0) return 0;
	t_max *= invd.x;
	*t = (bmax.x - o.x) * invd.x;
	float t_min_y, t_max_y;
	t_max_y = ((signs[1] ? bmax.y : bmin.y) - o.y) * invd.y;
	if (t_max_y < 0) return 0;
	t_min_y = ((signs[1] ? bmin.y : bmax.y) - o.y) * invd.y;
	if (t_min_y > t_max || t_max_y < *t) return 0;
	if (t_min_y > *t) *t = t_min_y;
	if (t_max_y < t_max) t_max = t_max_y;
	float t_min_z, t_max_z;
	t_max_z = ((signs[2] ? bmax.z : bmin.z) - o.z) * invd.z;
	if (t_max_z < 0) return 0;
	t_min_z = ((signs[2] ? bmin.z : bmax.z) - o.z) * invd.z;
	if (*t > t_max_z || t_min_z > t_max) return 0;
	if (t_min_z > *t) *t = t_min_z;
	if (*t < 0) *t = 0;
	return 1;
}

/* ------------------------------------------------------------------ triangle (TriangleMesh.h:82-104) */
static int tri_intersection(const o_tri* T, const o_ray* d, v3* P, float* t, float* alpha, float* beta, float* gamma) {
	*t = vdot(vsub(T->A, d->origin), T->N) / vdot(d->direction, T->N);
	if (*t < 0 || *t != *t) return 0;
	*P = vadd(d->origin, vscale(*t, d->direction));
	v3 w = vsub(*P, T->A);
	float b11 = vdot(w, T->u);
	float b21 = vdot(w, T->v);
	float detb = b11 * T->m22 - b21 * T->m12;
	*beta = detb * T->invdetm;
	if (*beta < 0) return 0;
	float detg = b21 * T->m11 - b11 * T->m12;
	*gamma = detg * T->invdetm;
	if (*gamma < 0) return 0;
	*alpha = 1 - *beta - *gamma;
	if (*alpha < 0) return 0;
	return 1;
}

/* ------------------------------------------------------------------ TriMesh::getMaterial (TriangleMesh.cpp:919-1026) */
static void mesh_get_material(const o_obj* o, int triId, float alpha, float beta, float gamma, o_mat* mat) {
	const o_mesh* g = o->mesh;
	float u = 0, v = 0;
	int textureId = g->indices[triId].group;
	int has_uv = 0;
	const o_idx* tri = &g->indices[triId];
	const o_tri* ts = &g->soup[triId];
	if ((g->nt != 0) && (tri->group >= 0) && (tri->uv[0] >= 0)) {
		if (!((size_t)tri->uv[0] >= (size_t)g->nt)) {
			u = (ts->uvs[0][0] * alpha + ts->uvs[1][0] * beta + ts->uvs[2][0] * gamma);
			v = (ts->uvs[0][1] * alpha + ts->uvs[1][1] * beta + ts->uvs[2][1] * gamma);
			has_uv = 1;
		}
	}
	query_material(o, textureId, u, v, mat);
	if (!g->interp_normals || (tri->n[0] == -1)) {
		mat->shadingN = ts->N;
	} else {
		mat->shadingN = vadd(vadd(vscale(alpha, ts->normals[0]), vscale(beta, ts->normals[1])), vscale(gamma, ts->normals[2]));
	}
	mat->shadingN = vnormalize(mat->shadingN);
	if ((o->ntex[T_NORMAL] != 0) && has_uv && ((size_t)(long)textureId < (size_t)o->ntex[T_NORMAL])) {
		v3 tangent = vadd(vadd(vscale(alpha, g->tangent_soup[triId * 3]), vscale(beta, g->tangent_soup[triId * 3 + 1])), vscale(gamma, g->tangent_soup[triId * 3 + 2]));
		tangent = vnormalize(tangent);
		v3 bitangent = vcross(mat->shadingN, tangent);
		v3 NsLocal = tex_getNormal(&o->tex[T_NORMAL][textureId], u, v);
		v3 Ns = vadd(vadd(vscale(NsLocal.x, tangent), vscale(NsLocal.y, bitangent)), vscale(NsLocal.z, mat->shadingN));
		if (Ns.x == 0. && Ns.y == 0 && Ns.z == 0) Ns = mat->shadingN;
		Ns = vnormalize(Ns);
		mat->shadingN = Ns;
	}
	if (o->flip_normals) mat->shadingN = vneg(mat->shadingN);
	/* vertexcolors / facecolors / display_edges: OUT OF SCOPE */
}

/* alpha-map test inside the leaf loop (TriangleMesh.cpp:1198-1205 / 1300-1307) */
static int mesh_alpha_rejects(const o_obj* o, int i, float alpha, float beta, float gamma) {
	const o_mesh* g = o->mesh;
	int textureId = g->indices[i].group;
	const o_idx* ix = &g->indices[i];
	if (g->nt > 0 && (size_t)o->ntex[T_ALPHA] > (size_t)(long)textureId && ix->uv[0] >= 0 && ix->uv[1] >= 0 && ix->uv[2] >= 0) {
		float u = g->uvs[ix->uv[0]].x * alpha + g->uvs[ix->uv[1]].x * beta + g->uvs[ix->uv[2]].x * gamma;
		float v = g->uvs[ix->uv[0]].y * alpha + g->uvs[ix->uv[1]].y * beta + g->uvs[ix->uv[2]].y * gamma;
		u = tex_wrap(u);
		v = tex_wrap(v);
		if (tex_getValRed(&o->tex[T_ALPHA][textureId], u, v) < 0.5) return 1;
	}
	return 0;
}

/* ------------------------------------------------------------------ TriMesh::intersection (TriangleMesh.cpp:1133-1235) */
static int mesh_intersection(const o_obj* o, const o_ray* d, v3* P, float* t, o_mat* mat, float cur_best_t, int* triangle_id) {
	const o_mesh* g = o->mesh;
	*t = cur_best_t;
	int has_inter = 0;
	float t_box_left, t_box_right;
	int best_index = -1;
	int goleft, goright;
	v3 localP = V(0, 0, 0);
	float localt, alpha = 0, beta = 0, gamma = 0;
	uint64_t c_box = 0, c_node = 0, c_tri = 0;

	o_ray invd; invd.origin = d->origin;
	invd.direction = V(1.f / d->direction.x, 1.f / d->direction.y, 1.f / d->direction.z);  /* 1./x narrowed == 1.f/x */
	char signs[3];
	signs[0] = (invd.direction.x >= 0) ? 1 : 0;
	signs[1] = (invd.direction.y >= 0) ? 1 : 0;
	signs[2] = (invd.direction.z >= 0) ? 1 : 0;

	tl_cnt[6]++;
	c_box++;
	trace_byte(0xFF); trace_byte(0);
	if (!box_invd(g->root_min, g->root_max, invd.origin, invd.direction, signs, &t_box_left)) { tl_cnt[0] += c_box; trace_byte(0xFE); return 0; }
	if (t_box_left > cur_best_t) { tl_cnt[0] += c_box; trace_byte(0xFE); return 0; }

	int l[50];
	float tnear[50];
	int idx_back = -1;
	l[++idx_back] = 0;
	tnear[idx_back] = t_box_left;

	while (idx_back >= 0) {
		if (tnear[idx_back] > *t) { idx_back--; continue; }
		const int current = l[idx_back--];
		c_node++;
		const int fg = g->nodes[current].fg;
		const int fd = g->nodes[current].fd;
		if (tl_trace) trace_byte(g->nodes[current].isleaf ? (uint8_t)(0x80 | (fd - fg > 63 ? 63 : fd - fg)) : (uint8_t)(idx_back + 1));
		if (!g->nodes[current].isleaf) {
			c_box += 2;
			if (signs[0] == 1) {
				goleft = (box_invd_posx(g->nodes[fg].bmin, g->nodes[fg].bmax, invd.origin, invd.direction, signs, &t_box_left) && t_box_left < *t);
				goright = (box_invd_posx(g->nodes[fd].bmin, g->nodes[fd].bmax, invd.origin, invd.direction, signs, &t_box_right) && t_box_right < *t);
			} else {
				goleft = (box_invd_negx(g->nodes[fg].bmin, g->nodes[fg].bmax, invd.origin, invd.direction, signs, &t_box_left) && t_box_left < *t);
				goright = (box_invd_negx(g->nodes[fd].bmin, g->nodes[fd].bmax, invd.origin, invd.direction, signs, &t_box_right) && t_box_right < *t);
			}
			if (goleft && goright) {
				if (t_box_left < t_box_right) {
					l[++idx_back] = fd; tnear[idx_back] = t_box_right;
					l[++idx_back] = fg; tnear[idx_back] = t_box_left;
				} else {
					l[++idx_back] = fg; tnear[idx_back] = t_box_left;
					l[++idx_back] = fd; tnear[idx_back] = t_box_right;
				}
			} else {
				if (goleft) { l[++idx_back] = fg; tnear[idx_back] = t_box_left; }
				if (goright) { l[++idx_back] = fd; tnear[idx_back] = t_box_right; }
			}
		} else {
			for (int i = fg; i < fd; i++) {
				c_tri++;
				if (tri_intersection(&g->soup[i], d, &localP, &localt, &alpha, &beta, &gamma)) {
					if (localt < *t) {
						if (mesh_alpha_rejects(o, i, alpha, beta, gamma)) continue;
						has_inter = 1;
						best_index = i;
						*t = localt;
					}
				}
			}
		}
	}
	tl_cnt[0] += c_box; tl_cnt[1] += c_node; tl_cnt[2] += c_tri;

	if (has_inter) {
		int i = best_index;
		*triangle_id = best_index;
		tri_intersection(&g->soup[i], d, &localP, &localt, &alpha, &beta, &gamma);
		if (isnan(alpha) && isnan(beta) && isnan(gamma)) { alpha = 1; beta = 0; gamma = 0; }
		if (isnan(alpha)) alpha = 0;
		if (isnan(beta)) beta = 0;
		if (isnan(gamma)) gamma = 0;
		if (isinf(alpha)) alpha = 1;
		if (isinf(beta)) beta = 1;
		if (isinf(gamma)) gamma = 1;
		*P = localP;
		mesh_get_material(o, i, alpha, beta, gamma, mat);
	}
	return has_inter;
}


/* ------------------------------------------------------------------ diagnostic: visiting orders of the any-hit traversal
   (test infrastructure for tools/anyhit_study.py; it changes nothing the oracle returns).  intersection_shadow's result is
   "does a reachable occluder (t < 0.999 dist) exist"; the orders below visit the tree without the reference's near / far
   ordering and without its `tnear > t` prune and count what they would cost:
     [0] calls  [1] reference: inner pops  [2] leaf pops  [3] triangle tests
     unordered binary, nearer child first:  [4] inner  [5] leaves  [6] triangles  [7] results differing from the reference
     unordered binary, left child first:    [8] inner  [9] leaves  [10] triangles
     four-wide (grandchildren tested directly), nearest slot first: [11] wide steps  [12] leaves  [13] triangles  [14] differing
     four-wide, first passing slot:         [15] wide steps  [16] leaves  [17] triangles
     [18] rays that passed a box with t_box >= 0.998 dist (the only ones whose result can depend on the order)
     [19] of those, rays that found an occluder (they are replayed in the reference's order)
     [20] max stack depth of the four-wide walk  [21] four-wide slot tests  [22] reference box tests  [23] occluded (reference) */
static _Thread_local uint64_t tl_any_loc[10];   /* local-entry variant (below): [0] rays [1] found in the local walk [2] steps / [3] leaves / [4] triangles of the local walk, [5..7] of the walk from the root that follows a local miss [8] result differs */
static uint64_t g_any_loc[10];
static int g_any_loc_height = 6;              /* binary levels between the entry node and the leaf that holds the ray's origin */
static _Thread_local uint64_t tl_any[24];
_Thread_local uint64_t tl_any_q4[6];
static uint64_t g_any_q4[6];
static uint64_t g_any[24];
static int g_any_study = 0;
void o_anyhit_study(int on) { g_any_study = on; if (on) { memset(g_any, 0, sizeof g_any); memset(g_any_q4, 0, sizeof g_any_q4); } }
void o_anyhit_study_local_height(int h) { g_any_loc_height = h; memset(g_any_loc, 0, sizeof g_any_loc); }
void o_anyhit_study_get_local(uint64_t* out10) {
	#pragma omp parallel
	{ for (int k = 0; k < 10; k++) { __atomic_fetch_add(&g_any_loc[k], tl_any_loc[k], __ATOMIC_RELAXED); tl_any_loc[k] = 0; } }
	memcpy(out10, g_any_loc, sizeof g_any_loc);
}
void o_anyhit_study_get_q4(uint64_t* out6) {
	#pragma omp parallel
	{ for (int k = 0; k < 6; k++) { __atomic_fetch_add(&g_any_q4[k], tl_any_q4[k], __ATOMIC_RELAXED); tl_any_q4[k] = 0; } }
	memcpy(out6, g_any_q4, sizeof g_any_q4);
}
void o_anyhit_study_get(uint64_t* out24) {
	#pragma omp parallel
	{ for (int k = 0; k < 24; k++) { if (k == 20) { uint64_t v = tl_any[k], cur = __atomic_load_n(&g_any[k], __ATOMIC_RELAXED); while (v > cur && !__atomic_compare_exchange_n(&g_any[k], &cur, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} } else __atomic_fetch_add(&g_any[k], tl_any[k], __ATOMIC_RELAXED); tl_any[k] = 0; } }
	memcpy(out24, g_any, sizeof g_any);
}
static int study_leaf(const o_obj* o, const o_ray* d, int fg, int fd, float dist_light, uint64_t* ntri) {
	const o_mesh* g = o->mesh;
	v3 localP; float localt, alpha, beta, gamma;
	for (int i = fg; i < fd; i++) {
		(*ntri)++;
		if (tri_intersection(&g->soup[i], d, &localP, &localt, &alpha, &beta, &gamma)) {
			if (mesh_alpha_rejects(o, i, alpha, beta, gamma)) continue;
			if (localt < dist_light * 0.999) return 1;
		}
	}
	return 0;
}
static void anyhit_study(const o_obj* o, const o_ray* d, float dist_light, int ref_result, uint64_t ref_inner, uint64_t ref_leaf, uint64_t ref_tri, uint64_t ref_box) {
	const o_mesh* g = o->mesh;
	o_ray invd; invd.origin = d->origin;
	invd.direction = V(1.f / d->direction.x, 1.f / d->direction.y, 1.f / d->direction.z);
	char signs[3] = { invd.direction.x >= 0, invd.direction.y >= 0, invd.direction.z >= 0 };
	tl_any[0]++; tl_any[1] += ref_inner; tl_any[2] += ref_leaf; tl_any[3] += ref_tri; tl_any[22] += ref_box; tl_any[23] += ref_result != 0;
	float tb;
	if (!box_invd(g->root_min, g->root_max, invd.origin, invd.direction, signs, &tb) || tb > dist_light) return;
	int stack[256];
	for (int variant = 0; variant < 2; variant++) {          /* unordered binary */
		int sp = 0, found = 0, flagged = 0;
		uint64_t inner = 0, leaves = 0, tris = 0;
		stack[sp++] = 0;
		while (sp > 0 && !found) {
			const int cur = stack[--sp];
			const o_node* n = &g->nodes[cur];
			if (n->isleaf) { leaves++; found = study_leaf(o, d, n->fg, n->fd, dist_light, &tris); continue; }
			inner++;
			float tl, tr;
			const int gl = box_invd(g->nodes[n->fg].bmin, g->nodes[n->fg].bmax, invd.origin, invd.direction, signs, &tl) && tl < dist_light;
			const int gr = box_invd(g->nodes[n->fd].bmin, g->nodes[n->fd].bmax, invd.origin, invd.direction, signs, &tr) && tr < dist_light;
			if ((gl && tl >= 0.998f * dist_light) || (gr && tr >= 0.998f * dist_light)) flagged = 1;
			if (gl && gr) {
				if (variant == 0 ? (tl < tr) : 1) { stack[sp++] = n->fd; stack[sp++] = n->fg; }
				else { stack[sp++] = n->fg; stack[sp++] = n->fd; }
			} else if (gl) stack[sp++] = n->fg;
			else if (gr) stack[sp++] = n->fd;
		}
		if (variant == 0) {
			tl_any[4] += inner; tl_any[5] += leaves; tl_any[6] += tris;
			if (found != (ref_result != 0) && !flagged) tl_any[7]++;
			if (flagged) { tl_any[18]++; if (found) tl_any[19]++; }
		} else { tl_any[8] += inner; tl_any[9] += leaves; tl_any[10] += tris; }
	}
	{          /* four-wide with 8-bit boxes (mipt_anyhit.h, DQuadNode): planes origin + q * 2^e per axis, rounded outwards; first passing slot */
		int sp = 0, found = 0;
		uint64_t wide = 0, leaves = 0, tris = 0, unverified = 0;
		stack[sp++] = 0;
		if (g->nodes[0].isleaf) { sp = 0; }
		while (sp > 0 && !found) {
			const int cur = stack[--sp];
			const o_node* n = &g->nodes[cur];
			if (n->isleaf) {
				leaves++;
				if (study_leaf(o, d, n->fg, n->fd, dist_light, &tris)) {
					float t;                                      /* an occluder counts if the leaf's own float box is reached */
					if (box_invd(n->bmin, n->bmax, invd.origin, invd.direction, signs, &t) && t < dist_light) found = 1; else unverified++;
				}
				continue;
			}
			wide++;
			int slot[4], ns = 0;
			const int ch[2] = { n->fg, n->fd };
			for (int c = 0; c < 2; c++) {
				const o_node* m = &g->nodes[ch[c]];
				if (m->isleaf) slot[ns++] = ch[c];
				else { slot[ns++] = m->fg; slot[ns++] = m->fd; }
			}
			v3 nmin = g->nodes[slot[0]].bmin, nmax = g->nodes[slot[0]].bmax;
			for (int k = 1; k < ns; k++) { nmin = vmin3(nmin, g->nodes[slot[k]].bmin); nmax = vmax3(nmax, g->nodes[slot[k]].bmax); }
			float scale[3];
			for (int a = 0; a < 3; a++) {
				int e = -100;
				const float ext = vget(nmax, a) - vget(nmin, a);
				if (ext > 0) { e = (int)ceilf(log2f(ext / 255.f)) - 1; if (e < -100) e = -100; }
				while (fmaf(255.f, ldexpf(1.f, e), vget(nmin, a)) < vget(nmax, a)) e++;
				scale[a] = ldexpf(1.f, e);
			}
			int pass[4], np = 0;
			for (int k = 0; k < ns; k++) {
				v3 qmin, qmax;
				for (int a = 0; a < 3; a++) {
					const float org = vget(nmin, a), lo = vget(g->nodes[slot[k]].bmin, a), hi = vget(g->nodes[slot[k]].bmax, a);
					int ql = (int)floorf((lo - org) / scale[a]); if (ql < 0) ql = 0; if (ql > 255) ql = 255;
					while (ql > 0 && fmaf((float)ql, scale[a], org) > lo) ql--;
					int qh = (int)ceilf((hi - org) / scale[a]); if (qh < 0) qh = 0; if (qh > 255) qh = 255;
					while (qh < 255 && fmaf((float)qh, scale[a], org) < hi) qh++;
					vset(&qmin, a, fmaf((float)ql, scale[a], org)); vset(&qmax, a, fmaf((float)qh, scale[a], org));
				}
				float t;
				if (box_invd(qmin, qmax, invd.origin, invd.direction, signs, &t) && t < dist_light) pass[np++] = slot[k];
			}
			/* (the kernel keeps the first passing slot in a register and pushes the others: depth of the LDS stack = sp - 1 of this walk) */
			for (int k = np - 1; k >= 0; k--) { stack[sp++] = pass[k]; if (k > 0 && sp - 1 > 16) tl_any_q4[5]++; }
		}
		tl_any_q4[0] += wide; tl_any_q4[1] += leaves; tl_any_q4[2] += tris; tl_any_q4[3] += unverified;
		if (!g->nodes[0].isleaf && found != (ref_result != 0)) tl_any_q4[4]++;
	}
	{          /* four-wide, first passing slot, LOCAL ENTRY: the walk starts at an ancestor of the leaf next to the ray's origin (a shadow ray leaves a
	              surface of this mesh more often than not, and what shadows a bumpy surface is the next bump); only when that subtree holds no
	              occluder does the walk start again from the root, skipping the subtree it has seen.  Any reachable occluder decides the ray:
	              the order-independence argument of the any-hit kernel does not care where the walk enters. */
		int path[64], depth = 0, cur = 0;
		while (!g->nodes[cur].isleaf && depth < 63) {
			path[depth++] = cur;
			const o_node* a = &g->nodes[g->nodes[cur].fg]; const o_node* b = &g->nodes[g->nodes[cur].fd];
			float da = 0, db = 0;
			for (int ax = 0; ax < 3; ax++) {
				const float x = vget(d->origin, ax);
				const float ea = fmaxf(fmaxf(vget(a->bmin, ax) - x, x - vget(a->bmax, ax)), 0.f), eb = fmaxf(fmaxf(vget(b->bmin, ax) - x, x - vget(b->bmax, ax)), 0.f);
				da += ea * ea; db += eb * eb;
			}
			cur = da <= db ? g->nodes[cur].fg : g->nodes[cur].fd;
		}
		int ed = depth - g_any_loc_height; if (ed < 0) ed = 0; ed &= ~1;      /* wide nodes sit at even binary depths */
		const int entry = depth > 0 ? path[ed] : 0;
		int found = 0;
		tl_any_loc[0]++;
		for (int phase = 0; phase < 2 && !found; phase++) {
			if (phase == 1 && entry == 0) break;                   /* the local walk was the whole tree */
			int sp = 0;
			uint64_t wide = 0, leaves = 0, tris = 0;
			stack[sp++] = phase == 0 ? entry : 0;
			if (g->nodes[0].isleaf) { sp = 0; }
			while (sp > 0 && !found) {
				const int c0 = stack[--sp];
				const o_node* n = &g->nodes[c0];
				if (n->isleaf) {
					leaves++;
					if (study_leaf(o, d, n->fg, n->fd, dist_light, &tris)) { float t; if (box_invd(n->bmin, n->bmax, invd.origin, invd.direction, signs, &t) && t < dist_light) found = 1; }
					continue;
				}
				wide++;
				int slot[4], ns = 0;
				const int ch[2] = { n->fg, n->fd };
				for (int c = 0; c < 2; c++) {
					const o_node* m = &g->nodes[ch[c]];
					if (m->isleaf) slot[ns++] = ch[c];
					else { slot[ns++] = m->fg; slot[ns++] = m->fd; }
				}
				for (int k = ns - 1; k >= 0; k--) {
					float t;
					if (phase == 1 && slot[k] == entry) continue;        /* seen by the local walk */
					if (box_invd(g->nodes[slot[k]].bmin, g->nodes[slot[k]].bmax, invd.origin, invd.direction, signs, &t) && t < dist_light) stack[sp++] = slot[k];
				}
			}
			tl_any_loc[2 + 3 * phase] += wide; tl_any_loc[3 + 3 * phase] += leaves; tl_any_loc[4 + 3 * phase] += tris;
			if (phase == 0 && found) tl_any_loc[1]++;
		}
		if (!g->nodes[0].isleaf && found != (ref_result != 0)) tl_any_loc[8]++;
	}
	for (int variant = 0; variant < 2; variant++) {          /* four-wide */
		int sp = 0, found = 0, flagged = 0, maxsp = 0;
		uint64_t wide = 0, leaves = 0, tris = 0, tests = 0;
		stack[sp++] = 0;
		if (g->nodes[0].isleaf) { leaves++; found = study_leaf(o, d, g->nodes[0].fg, g->nodes[0].fd, dist_light, &tris); sp = 0; }
		while (sp > 0 && !found) {
			const int cur = stack[--sp];
			const o_node* n = &g->nodes[cur];
			if (n->isleaf) { leaves++; found = study_leaf(o, d, n->fg, n->fd, dist_light, &tris); continue; }
			wide++;
			int slot[4], ns = 0;
			const int ch[2] = { n->fg, n->fd };
			for (int c = 0; c < 2; c++) {
				const o_node* m = &g->nodes[ch[c]];
				if (m->isleaf) slot[ns++] = ch[c];
				else { slot[ns++] = m->fg; slot[ns++] = m->fd; }
			}
			int pass[4]; float ts[4]; int np = 0;
			for (int k = 0; k < ns; k++) {
				float t;
				tests++;
				if (box_invd(g->nodes[slot[k]].bmin, g->nodes[slot[k]].bmax, invd.origin, invd.direction, signs, &t) && t < dist_light) {
					if (t >= 0.998f * dist_light) flagged = 1;
					pass[np] = slot[k]; ts[np] = t; np++;
				}
			}
			if (variant == 0 && np > 1) {                       /* the nearest passing slot is taken first: it goes to the top */
				int best = 0;
				for (int k = 1; k < np; k++) if (ts[k] < ts[best]) best = k;
				const int tmp = pass[best]; pass[best] = pass[0]; pass[0] = tmp;
			}
			for (int k = np - 1; k >= 0; k--) stack[sp++] = pass[k];
			if (sp > maxsp) maxsp = sp;
		}
		if (variant == 0) {
			tl_any[11] += wide; tl_any[12] += leaves; tl_any[13] += tris; tl_any[21] += tests;
			if (found != (ref_result != 0) && !flagged) tl_any[14]++;
			if ((uint64_t)maxsp > tl_any[20]) tl_any[20] = (uint64_t)maxsp;
		} else { tl_any[15] += wide; tl_any[16] += leaves; tl_any[17] += tris; }
	}
}

/* ------------------------------------------------------------------ TriMesh::intersection_shadow (TriangleMesh.cpp:1239-1319) */
static int mesh_intersection_shadow_impl(const o_obj* o, const o_ray* d, float* t, float cur_best_t, float dist_light, uint64_t* study4) {
	const o_mesh* g = o->mesh;
	*t = cur_best_t;
	int has_inter = 0;
	float t_box_left, t_box_right;
	int goleft, goright;
	v3 localP;
	float localt, alpha, beta, gamma;
	uint64_t c_box = 0, c_node = 0, c_tri = 0;

	o_ray invd; invd.origin = d->origin;
	invd.direction = V(1.f / d->direction.x, 1.f / d->direction.y, 1.f / d->direction.z);
	char signs[3];
	signs[0] = (invd.direction.x >= 0) ? 1 : 0;
	signs[1] = (invd.direction.y >= 0) ? 1 : 0;
	signs[2] = (invd.direction.z >= 0) ? 1 : 0;

	tl_cnt[7]++;
	c_box++;
	trace_byte(0xFF); trace_byte(1);
	if (!box_invd(g->root_min, g->root_max, invd.origin, invd.direction, signs, &t_box_left)) { tl_cnt[3] += c_box; trace_byte(0xFE); return 0; }
	if (t_box_left > cur_best_t || t_box_left > dist_light) { tl_cnt[3] += c_box; trace_byte(0xFE); return 0; }

	int l[50];
	float tnear[50];
	int idx_back = -1;
	l[++idx_back] = 0;
	tnear[idx_back] = t_box_left;

	while (idx_back >= 0) {
		if (tnear[idx_back] > *t) { idx_back--; continue; }
		const int current = l[idx_back--];
		c_node++;
		const int fg = g->nodes[current].fg;
		const int fd = g->nodes[current].fd;
		if (tl_trace) trace_byte(g->nodes[current].isleaf ? (uint8_t)(0x80 | (fd - fg > 63 ? 63 : fd - fg)) : (uint8_t)(idx_back + 1));
		if (!g->nodes[current].isleaf) {
			c_box += 2;
			goleft = (box_invd(g->nodes[fg].bmin, g->nodes[fg].bmax, invd.origin, invd.direction, signs, &t_box_left) && (t_box_left < *t) && (t_box_left < dist_light));
			goright = (box_invd(g->nodes[fd].bmin, g->nodes[fd].bmax, invd.origin, invd.direction, signs, &t_box_right) && (t_box_right < *t) && (t_box_right < dist_light));
			if (goleft && goright) {
				if (t_box_left < t_box_right) {
					l[++idx_back] = fd; tnear[idx_back] = t_box_right;
					l[++idx_back] = fg; tnear[idx_back] = t_box_left;
				} else {
					l[++idx_back] = fg; tnear[idx_back] = t_box_left;
					l[++idx_back] = fd; tnear[idx_back] = t_box_right;
				}
			} else {
				if (goleft) { l[++idx_back] = fg; tnear[idx_back] = t_box_left; }
				if (goright) { l[++idx_back] = fd; tnear[idx_back] = t_box_right; }
			}
		} else {
			if (study4) study4[1]++;
			for (int i = fg; i < fd; i++) {
				c_tri++;
				if (tri_intersection(&g->soup[i], d, &localP, &localt, &alpha, &beta, &gamma)) {
					if (localt < *t) {
						if (mesh_alpha_rejects(o, i, alpha, beta, gamma)) continue;
						has_inter = 1;
						*t = localt;
						if (*t < dist_light * 0.999) { tl_cnt[3] += c_box; tl_cnt[4] += c_node; tl_cnt[5] += c_tri; if (study4) { study4[0] = c_node; study4[2] = c_tri; study4[3] = c_box; } return 1; }  /* double compare */
					}
				}
			}
		}
	}
	tl_cnt[3] += c_box; tl_cnt[4] += c_node; tl_cnt[5] += c_tri;
	if (study4) { study4[0] = c_node; study4[2] = c_tri; study4[3] = c_box; }
	return has_inter;
}
static int mesh_intersection_shadow(const o_obj* o, const o_ray* d, float* t, float cur_best_t, float dist_light) {
	if (!g_any_study) return mesh_intersection_shadow_impl(o, d, t, cur_best_t, dist_light, NULL);
	uint64_t s4[4] = { 0, 0, 0, 0 };
	const int r = mesh_intersection_shadow_impl(o, d, t, cur_best_t, dist_light, s4);
	anyhit_study(o, d, dist_light, r && (*t < dist_light * 0.999), s4[0] - s4[1], s4[1], s4[2], s4[3]);
	return r;
}

/* ------------------------------------------------------------------ Sphere (Geometry.h:918-992, 1071-1094) */
static int sphere_intersection(const o_obj* s, const o_ray* d, v3* P, float* t, o_mat* mat, int* triangle_id) {
	float b = vdot(d->direction, vsub(d->origin, s->O));
	float a = vnorm2(d->direction);
	float c = vnorm2(vsub(d->origin, s->O)) - s->R2;
	float delta = b * b - a * c;
	if (delta < 0) return 0;
	float sqDelta = sqrtf(delta);
	float inva = 1.f / a;
	float t2 = (-b + sqDelta) * inva;
	if (t2 < 0) return 0;
	float t1 = (-b - sqDelta) * inva;
	if (t1 > 0) *t = t1; else *t = t2;
	*P = vadd(d->origin, vscale(*t, d->direction));
	v3 N = vsub(*P, s->O);
	if (s->has_envmap) {
		N = vfast_normalize(N);
		float theta = 1.f - acosf(N.y) / (float)M_PI;
		float phi = (atan2f(-N.z, N.x) + M_PI) / (2.f * (float)M_PI);   /* double sum, float denominator */
		query_material(s, 0, theta, phi, mat);
		mat->shadingN = vneg(N);
		int idx = 3 * ((int)(theta * (s->envH - 1.f)) * s->envW + (int)(phi * (s->envW - 1.f)));
		if (idx < 0 || idx >= 3 * s->envW * s->envH) mat->Ke = V(0, 0, 0);
		else mat->Ke = vscale((100000.f / 255.f), V(s->envtex[idx + 0], s->envtex[idx + 1], s->envtex[idx + 2]));
		*triangle_id = -1;
		return 1;
	}
	if (s->ntex[T_KD] != 0 || s->ntex[T_KS] != 0 || s->ntex[T_NE] != 0 || s->ntex[T_TRANSP] != 0 || s->ntex[T_REFR] != 0) {
		N = vfast_normalize(N);
		float theta = 1.f - acosf(N.y) / (float)M_PI;
		float phi = (atan2f(-N.z, N.x) + (float)M_PI) / (2.f * (float)M_PI);
		query_material(s, 0, theta, phi, mat);
	}
	mat->shadingN = N;
	mat->Ke = V(0.f, 0.f, 0.f);
	if (s->flip_normals) mat->shadingN = vneg(mat->shadingN);
	*triangle_id = -1;
	return 1;
}
static int sphere_intersection_shadow(const o_obj* s, const o_ray* d, float* t) {
	float b = vdot(d->direction, vsub(d->origin, s->O));
	float a = vnorm2(d->direction);
	float c = vnorm2(vsub(d->origin, s->O)) - s->R2;
	float delta = b * b - a * c;
	if (delta < 0) return 0;
	float sqDelta = sqrtf(delta);
	float inva = 1.f / a;   /* 1./(a) narrowed == 1.f/a */
	float t2 = (-b + sqDelta) * inva;
	if (t2 < 0) return 0;
	float t1 = (-b - sqDelta) * inva;
	if (t1 > 0) *t = t1; else *t = t2;
	return 1;
}

/* ------------------------------------------------------------------ Plane (Geometry.h:1142-1157, 1185-1191) */
static int plane_intersection(const o_obj* p, const o_ray* d, v3* P, float* t, o_mat* mat, int* triangle_id) {
	mat->shadingN = p->vecN;
	float ddot = vdot(d->direction, p->vecN);
	if (fabsf(ddot) < 1E-9) return 0;
	*t = vdot(vsub(p->A, d->origin), p->vecN) / ddot;
	if (*t <= 0.) return 0;
	*P = vadd(d->origin, vscale(*t, d->direction));
	*triangle_id = -1;
	float u = P->x * 0.1f;
	float v = P->z * 0.1f;
	query_material(p, 0, u, v, mat);
	return 1;
}
static int plane_intersection_shadow(const o_obj* p, const o_ray* d, float* t) {
	float ddot = vdot(d->direction, p->vecN);
	if (fabsf(ddot) < 1E-9) return 0;
	*t = vdot(vsub(p->A, d->origin), p->vecN) / ddot;
	if (*t <= 0.) return 0;
	return 1;
}

/* ------------------------------------------------------------------ Scene::intersection (Geometry.cpp:589-688) */
static int scene_intersection(const o_ctx* c, const o_ray* d, v3* P, int* sphere_id, float* min_t, o_mat* mat, int* triangle_id) {
	int has_inter = 0;
	*min_t = INFINITY;   /* min_t = 1E99 narrowed to float */
	v3 localP = V(0, 0, 0);
	o_mat localmat; mat_default(&localmat);
	float t = 0;
	for (int i = 0; i < c->nobj; i++) {
		const o_obj* o = &c->objs[i];
		o_ray tr;
		tr.direction = apply_inverse_rotation_scaling(o, d->direction);
		tr.origin = apply_inverse_transformation(o, d->origin);
		int local_has_inter;
		if (o->type == OT_SPHERE) local_has_inter = sphere_intersection(o, &tr, &localP, &t, &localmat, triangle_id);
		else if (o->type == OT_PLANE) local_has_inter = plane_intersection(o, &tr, &localP, &t, &localmat, triangle_id);
		else local_has_inter = mesh_intersection(o, &tr, &localP, &t, &localmat, *min_t, triangle_id);
		if (local_has_inter) {
			if (t < *min_t) {
				has_inter = 1;
				*min_t = t;
				*P = localP;
				*sphere_id = i;
				*mat = localmat;
			}
		}
	}
	if (has_inter) {
		*P = apply_transformation(&c->objs[*sphere_id], *P);
		mat->shadingN = apply_rotation(&c->objs[*sphere_id], mat->shadingN);
	}
	mat->shadingN = vfast_normalize(mat->shadingN);
	return has_inter;
}

/* ------------------------------------------------------------------ Scene::intersection_shadow (Geometry.cpp:691-744) */
static int scene_intersection_shadow_g(const o_ctx* c, const o_ray* d, float dist_light, int avoid_ghosts) {
	float min_t = INFINITY;
	for (int i = 0; i < c->nobj; i++) {
		const o_obj* o = &c->objs[i];
		if (avoid_ghosts && o->ghost) continue;                     /* Geometry.cpp:722 */
		o_ray tr;
		tr.direction = apply_inverse_rotation_scaling(o, d->direction);
		tr.origin = apply_inverse_transformation(o, d->origin);
		float t = 0;
		int local_has_inter;
		if (o->type == OT_SPHERE) local_has_inter = sphere_intersection_shadow(o, &tr, &t);
		else if (o->type == OT_PLANE) local_has_inter = plane_intersection_shadow(o, &tr, &t);
		else local_has_inter = mesh_intersection_shadow(o, &tr, &t, min_t, dist_light);
		if (local_has_inter) {
			if (t < dist_light * 0.999) return 1;   /* float promoted, double compare */
		}
	}
	return 0;
}

static int scene_intersection_shadow(const o_ctx* c, const o_ray* d, float dist_light) { return scene_intersection_shadow_g(c, d, dist_light, 0); }

/* ------------------------------------------------------------------ samplers */
/* Vector.h:567-579 getTangent */
static v3 get_tangent(v3 N) {
	v3 tangent1;
	v3 absN = V(fabsf(N.x), fabsf(N.y), fabsf(N.z));
	if (absN.x <= absN.y && absN.x <= absN.z) tangent1 = V(0, -N.z, N.y);
	else if (absN.y <= absN.x && absN.y <= absN.z) tangent1 = V(-N.z, 0, N.x);
	else tangent1 = V(-N.y, N.x, 0);
	return vnormalize(tangent1);
}
/* Vector.h:582-589 random_cos(N, r1, r2) */
static v3 random_cos12(v3 N, float r1, float r2) {
	float sr2 = sqrtf(1.f - r2);
	v3 loc = V(cosf((float)(2. * M_PI) * r1) * sr2, sinf((float)(2. * M_PI) * r1) * sr2, sqrtf(r2));
	v3 tangent1 = get_tangent(N);
	v3 tangent2 = vcross(tangent1, N);
	return vadd(vadd(vscale(loc.z, N), vscale(loc.x, tangent1)), vscale(loc.y, tangent2));
}
/* Vector.h:591-600 random_cos(N): two engine draws */
static v3 random_cos_rng(v3 N, pcg32_t* rng) {
	float r1 = pcg_uniform(rng);
	float r2 = pcg_uniform(rng);
	return random_cos12(N, r1, r2);
}

/* BRDF.h:41-61 PhongBRDF::random_Phong */
static v3 random_phong(v3 R, float phong_exponent, float r1, float r2) {
	float facteur = sqrtf(1 - powf(r2, 2.f / (phong_exponent + 1.f)));
	v3 loc = V((float)(cos(2 * M_PI * r1) * facteur), (float)(sin(2 * M_PI * r1) * facteur), (float)pow(r2, 1. / (phong_exponent + 1)));
	v3 tangent1;
	v3 absR = V(fabsf(R.x), fabsf(R.y), fabsf(R.z));
	if (absR.x <= absR.y && absR.x <= absR.z) tangent1 = V(0, -R.z, R.y);
	else if (absR.y <= absR.x && absR.y <= absR.z) tangent1 = V(-R.z, 0, R.x);
	else tangent1 = V(-R.y, R.x, 0);
	tangent1 = vnormalize(tangent1);
	v3 tangent2 = vcross(tangent1, R);
	return vadd(vadd(vscale(loc.z, R), vscale(loc.x, tangent1)), vscale(loc.y, tangent2));
}
/* BRDF.h:63-86 PhongBRDF::sample (r1,r2 given; lobe pick draws once from the engine) */
static v3 phong_sample(const o_mat* mat, v3 wo, v3 N, float* pdf, float r1, float r2, int* has_sampled_diffuse, pcg32_t* rng) {
	float avgNe = (mat->Ne.x + mat->Ne.y + mat->Ne.z) / 3.f;
	v3 dir;
	float p = 1 - (mat->Ks.x + mat->Ks.y + mat->Ks.z) / 3.f;
	v3 R = vreflect(vneg(wo), N);
	if ((float)pcg_next(rng) / 4294967296.f < p) {   /* engine()/(float)engine.max() */
		*has_sampled_diffuse = 1;
		dir = random_cos12(N, r1, r2);
	} else {
		*has_sampled_diffuse = 0;
		dir = random_phong(R, avgNe, r1, r2);
	}
	float proba_phong = (avgNe + 1) / (2.f * M_PI) * powf(vdot(R, dir), avgNe);
	float proba_globale = p * vdot(N, dir) / (M_PI) + (1.f - p) * proba_phong;
	*pdf = proba_globale;
	return dir;
}
/* BRDF.h:88-96 PhongBRDF::eval */
static v3 phong_eval(const o_mat* mat, v3 wi, v3 wo, v3 N) {
	v3 reflechi = vreflect(vneg(wo), N);
	float d = vdot(reflechi, wi);
	if (d < 0) return vdivs(mat->Kd, (float)M_PI);
	v3 lobe = V((float)(powf(d, mat->Ne.x) * (mat->Ne.x + 2.f) / O_TWO_PI),
	            (float)(powf(d, mat->Ne.y) * (mat->Ne.y + 2.f) / O_TWO_PI),
	            (float)(powf(d, mat->Ne.z) * (mat->Ne.z + 2.f) / O_TWO_PI));
	return vadd(vdivs(mat->Kd, (float)M_PI), vmul(lobe, mat->Ks));
}


/* ------------------------------------------------------------------ IsoMERLBRDF (BRDF.h:192-247, MERLBRDFRead.cpp:29-206) */
#define MERL_TH 90
#define MERL_TD 90
#define MERL_PD 360
static void merl_rotate_vector(const double* vector, const double* axis, double angle, double* out) {   /* MERLBRDFRead.cpp:49-72 */
	double temp;
	double cross[3];
	double cos_ang = cos(angle);
	double sin_ang = sin(angle);
	out[0] = vector[0] * cos_ang; out[1] = vector[1] * cos_ang; out[2] = vector[2] * cos_ang;
	temp = axis[0] * vector[0] + axis[1] * vector[1] + axis[2] * vector[2];
	temp = temp * (1.0 - cos_ang);
	out[0] += axis[0] * temp; out[1] += axis[1] * temp; out[2] += axis[2] * temp;
	cross[0] = axis[1] * vector[2] - axis[2] * vector[1];
	cross[1] = axis[2] * vector[0] - axis[0] * vector[2];
	cross[2] = axis[0] * vector[1] - axis[1] * vector[0];
	out[0] += cross[0] * sin_ang; out[1] += cross[1] * sin_ang; out[2] += cross[2] * sin_ang;
}
static void merl_normalize(double* v) { double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] = v[0] / len; v[1] = v[1] / len; v[2] = v[2] / len; }
static void merl_lookup(const double* brdf, double theta_in, double fi_in, double theta_out, double fi_out, double* r, double* g, double* b) {
	/* std_coords_to_half_diff_coords (MERLBRDFRead.cpp:76-127) */
	double in_vec_z = cos(theta_in), proj_in_vec = sin(theta_in);
	double in_vec_x = proj_in_vec * cos(fi_in), in_vec_y = proj_in_vec * sin(fi_in);
	double in[3] = { in_vec_x, in_vec_y, in_vec_z };
	merl_normalize(in);
	double out_vec_z = cos(theta_out), proj_out_vec = sin(theta_out);
	double out_vec_x = proj_out_vec * cos(fi_out), out_vec_y = proj_out_vec * sin(fi_out);
	double half[3] = { (in_vec_x + out_vec_x) / 2.0f, (in_vec_y + out_vec_y) / 2.0f, (in_vec_z + out_vec_z) / 2.0f };
	merl_normalize(half);
	double theta_half = acos(half[2]);
	double fi_half = atan2(half[1], half[0]);
	double bi_normal[3] = { 0.0, 1.0, 0.0 }, normal[3] = { 0.0, 0.0, 1.0 }, temp[3], diff[3];
	merl_rotate_vector(in, normal, -fi_half, temp);
	merl_rotate_vector(temp, bi_normal, -theta_half, diff);
	double theta_diff = acos(diff[2]);
	double fi_diff = atan2(diff[1], diff[0]);
	/* index functions (MERLBRDFRead.cpp:134-180) */
	int th_idx;
	if (theta_half <= 0.0) th_idx = 0;
	else {
		double theta_half_deg = ((theta_half / (M_PI / 2.0)) * MERL_TH);
		double t = sqrt(theta_half_deg * MERL_TH);
		th_idx = (int)t;
		if (th_idx < 0) th_idx = 0;
		if (th_idx >= MERL_TH) th_idx = MERL_TH - 1;
	}
	int td = (int)(theta_diff / (M_PI * 0.5) * MERL_TD);
	int td_idx = td < 0 ? 0 : (td < MERL_TD - 1 ? td : MERL_TD - 1);
	if (fi_diff < 0.0) fi_diff += M_PI;
	int pd = (int)(fi_diff / M_PI * MERL_PD / 2);
	int pd_idx = pd < 0 ? 0 : (pd < MERL_PD / 2 - 1 ? pd : MERL_PD / 2 - 1);
	int ind = pd_idx + td_idx * MERL_PD / 2 + th_idx * MERL_PD / 2 * MERL_TD;
	*r = brdf[ind] * (1.0 / 1500.0);
	*g = brdf[ind + MERL_TH * MERL_TD * MERL_PD / 2] * (1.15 / 1500.0);
	*b = brdf[ind + MERL_TH * MERL_TD * MERL_PD] * (1.66 / 1500.0);
}
static v3 merl_eval(const double* data, v3 wi, v3 wo, v3 N) {   /* BRDF.h:204-246 */
	v3 tangent1;
	v3 absN = V(fabsf(N.x), fabsf(N.y), fabsf(N.z));
	if (absN.x <= absN.y && absN.x <= absN.z) tangent1 = V(0, -N.z, N.y);
	else if (absN.y <= absN.x && absN.y <= absN.z) tangent1 = V(-N.z, 0, N.x);
	else tangent1 = V(-N.y, N.x, 0);
	tangent1 = vnormalize(tangent1);
	v3 tangent2 = vcross(tangent1, N);
	v3 wil = V(vdot(wi, tangent1), vdot(wi, tangent2), vdot(wi, N));
	v3 wol = V(vdot(wo, tangent1), vdot(wo, tangent2), vdot(wo, N));
	float thetai = acosf(wil.z);
	if (thetai >= M_PI / 2) return V(0., 0., 0.);
	float thetao = acosf(wol.z);
	if (thetao >= M_PI / 2) return V(0., 0., 0.);
	float phio = atan2f(wol.y, wol.x);
	if (phio < 0) phio += 2 * M_PI;
	float phii = atan2f(wil.y, wil.x);
	if (phii < 0) phii += 2 * M_PI;
	double r, g, b;
	merl_lookup(data, thetai, phii, thetao, phio, &r, &g, &b);
	return V((float)r, (float)g, (float)b);
}

/* ------------------------------------------------------------------ camera (Vector.h:792-825, non-lenticular) */
static o_ray generate_direction(const o_ctx* c, float init_t, int i, int j, float dx_sensor, float dy_sensor, float dx_aperture, float dy_aperture, int W, int H) {
	float k = W / (2 * tanf(c->fov / 2));
	v3 camera_right = vcross(c->cam_dir, c->cam_up);
	v3 C1 = c->cam_pos;
	v3 dv;
	if (c->is_lenticular) {                                        /* Vector.h:799-812 */
		float L = c->focus * tanf(c->lenticular_max_angle / 2) / (c->lenticular_nb_images / 2.0);
		int offset = -((j / c->lenticular_pixel_width) % c->lenticular_nb_images - c->lenticular_nb_images / 2);
		v3 P = vadd(c->cam_pos, vscale(c->focus, V(0, 0, 1)));
		C1 = vadd(c->cam_pos, vscale(offset * L, camera_right));
		v3 v1 = vnormalize(vsub(P, C1));
		v3 PprojCam1 = vadd(vscale(k / vdot(v1, c->cam_dir), v1), C1);
		float pixProjCam1_j = PprojCam1.x + W / 2 - 0.5;
		float pixProjCam1_i = PprojCam1.y + H / 2 - 0.5;
		dv = V((j - pixProjCam1_j) + dx_sensor, (i - pixProjCam1_i) + dy_sensor, k);
	} else dv = V((float)(j - W / 2 + 0.5 + dx_sensor), (float)(i - H / 2 + 0.5 + dy_sensor), k);  /* int, then double sum, narrowed */
	dv = vnormalize(dv);
	dv = vadd(vadd(vscale(dv.x, camera_right), vscale(dv.y, c->cam_up)), vscale(dv.z, c->cam_dir));
	v3 destination = vadd(C1, vscale(c->focus / fabsf(vdot(dv, c->cam_dir)), dv));
	v3 new_origin = vadd(vadd(C1, vscale(dx_aperture, camera_right)), vscale(dy_aperture, c->cam_up));
	v3 new_direction = vnormalize(vsub(destination, new_origin));
	o_ray r;
	r.origin = vadd(new_origin, vdivs(vscale(init_t, new_direction), vdot(new_direction, c->cam_dir)));
	r.direction = new_direction;
	return r;
}

/* ------------------------------------------------------------------ getColor (Raytracer.cpp:196-664), in-scope branches */
/* The whole loop incl. the fog, subsurface, ghost and background-photo branches; subsurface colours are
 * restated for meshes only (Sphere / Plane::reservoir_sampling_intersection are not). */
/* normalValue / albedoValue: the denoiser inputs of Raytracer.cpp:255-258 (shading normal and Kd of the FIRST hit;
   left untouched without one, so they keep the zeros `Vector normal, albedo;` starts from, :1628).  May be NULL. */
/* ------------------------------------------------------------------ subsurface probe: a uniformly random one of the intersections in [min_t, max_t)
   TriMesh::reservoir_sampling_intersection (TriangleMesh.cpp:1321-1426): the closest-hit traversal order with a fixed far
   bound; every accepted triangle draws one number, so the visiting order decides the draws. */
static int mesh_reservoir_intersection(const o_obj* o, const o_ray* d, v3* P, float* t, o_mat* mat, int* triangle_id, int* current_nb_intersections,
                                       float min_t, float max_t, pcg32_t* rng) {
	const o_mesh* g = o->mesh;
	int has_inter = 0;
	float t_box_left, t_box_right;
	int best_index = -1;
	int goleft, goright;
	v3 localP = V(0, 0, 0);
	float localt, alpha = 0, beta = 0, gamma = 0;
	o_ray invd; invd.origin = d->origin;
	invd.direction = V(1.f / d->direction.x, 1.f / d->direction.y, 1.f / d->direction.z);
	char signs[3];
	signs[0] = (invd.direction.x >= 0) ? 1 : 0;
	signs[1] = (invd.direction.y >= 0) ? 1 : 0;
	signs[2] = (invd.direction.z >= 0) ? 1 : 0;
	if (!box_invd(g->root_min, g->root_max, invd.origin, invd.direction, signs, &t_box_left)) return 0;
	if (t_box_left > max_t) return 0;
	int l[50];
	float tnear[50];
	int idx_back = -1;
	l[++idx_back] = 0;
	tnear[idx_back] = t_box_left;
	while (idx_back >= 0) {
		if (tnear[idx_back] > max_t) { idx_back--; continue; }
		const int current = l[idx_back--];
		const int fg = g->nodes[current].fg;
		const int fd = g->nodes[current].fd;
		if (!g->nodes[current].isleaf) {
			if (signs[0] == 1) {
				goleft = (box_invd_posx(g->nodes[fg].bmin, g->nodes[fg].bmax, invd.origin, invd.direction, signs, &t_box_left) && t_box_left < max_t);
				goright = (box_invd_posx(g->nodes[fd].bmin, g->nodes[fd].bmax, invd.origin, invd.direction, signs, &t_box_right) && t_box_right < max_t);
			} else {
				goleft = (box_invd_negx(g->nodes[fg].bmin, g->nodes[fg].bmax, invd.origin, invd.direction, signs, &t_box_left) && t_box_left < max_t);
				goright = (box_invd_negx(g->nodes[fd].bmin, g->nodes[fd].bmax, invd.origin, invd.direction, signs, &t_box_right) && t_box_right < max_t);
			}
			if (goleft && goright) {
				if (t_box_left < t_box_right) {
					l[++idx_back] = fd; tnear[idx_back] = t_box_right;
					l[++idx_back] = fg; tnear[idx_back] = t_box_left;
				} else {
					l[++idx_back] = fg; tnear[idx_back] = t_box_left;
					l[++idx_back] = fd; tnear[idx_back] = t_box_right;
				}
			} else {
				if (goleft) { l[++idx_back] = fg; tnear[idx_back] = t_box_left; }
				if (goright) { l[++idx_back] = fd; tnear[idx_back] = t_box_right; }
			}
		} else {
			for (int i = fg; i < fd; i++) {
				if (tri_intersection(&g->soup[i], d, &localP, &localt, &alpha, &beta, &gamma)) {
					if (localt < max_t && localt >= min_t) {
						if (mesh_alpha_rejects(o, i, alpha, beta, gamma)) continue;
						(*current_nb_intersections)++;
						float r1 = pcg_uniform(rng);
						if (r1 < 1. / *current_nb_intersections) {
							has_inter = 1;
							best_index = i;
							*t = localt;
						}
					}
				}
			}
		}
	}
	if (has_inter) {
		int i = best_index;
		*triangle_id = best_index;
		tri_intersection(&g->soup[i], d, &localP, &localt, &alpha, &beta, &gamma);
		if (isnan(alpha) && isnan(beta) && isnan(gamma)) { alpha = 1; beta = 0; gamma = 0; }
		if (isnan(alpha)) alpha = 0;
		if (isnan(beta)) beta = 0;
		if (isnan(gamma)) gamma = 0;
		if (isinf(alpha)) alpha = 1;
		if (isinf(beta)) beta = 1;
		if (isinf(gamma)) gamma = 1;
		*P = localP;
		mesh_get_material(o, i, alpha, beta, gamma, mat);
	}
	return has_inter;
}
/* Plane::reservoir_sampling_intersection (Geometry.h:1159-1183) */
static int plane_reservoir_intersection(const o_obj* p, const o_ray* d, v3* P, float* t, o_mat* mat, int* triangle_id, int* current_nb_intersections,
                                        float min_t, float max_t, pcg32_t* rng) {
	mat->shadingN = p->vecN;
	float ddot = vdot(d->direction, p->vecN);
	if (fabsf(ddot) < 1E-9) return 0;
	float curt = vdot(vsub(p->A, d->origin), p->vecN) / ddot;
	if (curt < min_t || curt >= max_t) return 0;
	(*current_nb_intersections)++;
	float r1 = pcg_uniform(rng);
	if (r1 >= 1. / *current_nb_intersections) return 0;
	*t = curt;
	*P = vadd(d->origin, vscale(*t, d->direction));
	*triangle_id = -1;
	float u = P->x * 0.1f;
	float v = P->z * 0.1f;
	query_material(p, 0, u, v, mat);
	return 1;
}
/* Sphere::reservoir_sampling_intersection (Geometry.h:994-1068): the sphere's roots in [min_t, max_t), in root order, each drawing from
   the engine; the kept one's material is looked up at spherical coordinates computed with DOUBLE intermediates (`1 - acos(N[1]) / M_PI`,
   `(atan2(...) + M_PI) / (2.*M_PI)` narrowed to float) — unlike Sphere::intersection's all-float ones (:976-977). */
static int sphere_reservoir_intersection(const o_obj* s, const o_ray* d, v3* P, float* t, o_mat* mat, int* triangle_id, int* current_nb_intersections,
                                         float min_t, float max_t, pcg32_t* rng) {
	if (s->has_envmap) return 0;
	float b = vdot(d->direction, vsub(d->origin, s->O));
	float a = vnorm2(d->direction);
	float c = vnorm2(vsub(d->origin, s->O)) - s->R2;
	float delta = b * b - a * c;
	if (delta < 0) return 0;
	float sqDelta = sqrtf(delta);
	float inva = 1.f / a;
	float t2 = (-b + sqDelta) * inva;
	if (t2 < min_t) return 0;
	float t1 = (-b - sqDelta) * inva;
	int has_inter = 0;
	if (t1 >= min_t) {
		if (t1 < max_t) {
			(*current_nb_intersections)++;
			float r1 = pcg_uniform(rng);
			if (r1 < 1.f / *current_nb_intersections) { *t = t1; has_inter = 1; }
		}
		if (t2 < max_t) {
			(*current_nb_intersections)++;
			float r1 = pcg_uniform(rng);
			if (r1 < 1.f / *current_nb_intersections) { *t = t2; has_inter = 1; }
		}
	} else {
		if (t2 < max_t) {
			(*current_nb_intersections)++;
			float r1 = pcg_uniform(rng);
			if (r1 < 1.f / *current_nb_intersections) { *t = t2; has_inter = 1; }
		}
	}
	if (!has_inter) return 0;
	*P = vadd(d->origin, vscale(*t, d->direction));
	v3 N = vsub(*P, s->O);
	if (s->ntex[T_KD] != 0 || s->ntex[T_KS] != 0 || s->ntex[T_NE] != 0 || s->ntex[T_TRANSP] != 0 || s->ntex[T_REFR] != 0) {
		N = vfast_normalize(N);
		float theta = 1 - acosf(N.y) / M_PI;
		float phi = (atan2f(-N.z, N.x) + M_PI) / (2. * M_PI);
		query_material(s, 0, theta, phi, mat);
	}
	mat->shadingN = N;
	mat->Ke = V(0.f, 0.f, 0.f);
	if (s->flip_normals) mat->shadingN = vneg(mat->shadingN);
	*triangle_id = -1;
	return 1;
}
/* Scene::get_random_intersection (Geometry.cpp:339-470) restricted to one object (sphere_id != -1), which is how the
   subsurface branch calls it. */
static int scene_get_random_intersection(const o_ctx* c, const o_ray* d, v3* P, int sphere_id, float* min_t, o_mat* mat, int* triangle_id, float tmin, float tmax, pcg32_t* rng) {
	int has_inter = 0;
	*min_t = INFINITY;
	int nb_intersections = 0;
	const o_obj* o = &c->objs[sphere_id];
	o_ray tr;
	tr.direction = apply_inverse_rotation_scaling(o, d->direction);
	tr.origin = apply_inverse_transformation(o, d->origin);
	if (o->type == OT_TRIMESH) has_inter = mesh_reservoir_intersection(o, &tr, P, min_t, mat, triangle_id, &nb_intersections, tmin, tmax, rng);
	else if (o->type == OT_PLANE) has_inter = plane_reservoir_intersection(o, &tr, P, min_t, mat, triangle_id, &nb_intersections, tmin, tmax, rng);
	else has_inter = sphere_reservoir_intersection(o, &tr, P, min_t, mat, triangle_id, &nb_intersections, tmin, tmax, rng);
	if (has_inter) {
		*P = apply_transformation(o, *P);
		mat->shadingN = apply_rotation(o, mat->shadingN);
	}
	mat->shadingN = vfast_normalize(mat->shadingN);
	return has_inter;
}

/* ------------------------------------------------------------------ fog: single scattering (Raytracer.cpp:20-192) */
static float int_exponential(float y0, float ysol, float beta, float s, float uy) {          /* :20-40 */
	float result;
	if (fabsf(uy * beta) < 0.0001) {
		result = expf(-beta * (y0 - ysol)) * (s);
	} else {
		result = (expf(-beta * (y0 - ysol)) - expf(-beta * (y0 + s * uy - ysol))) / (uy * beta);
	}
	return result;
}
static v3 random_uniform_sphere(pcg32_t* rng) {                                                /* Vector.h:604-615, T = float */
	float r1 = pcg_uniform(rng);
	float r2 = pcg_uniform(rng);
	v3 result;
	result.x = 2.f * cosf((float)(2. * M_PI) * r1) * sqrtf(r2 * (1 - r2));
	result.y = 2.f * sinf((float)(2. * M_PI) * r1) * sqrtf(r2 * (1 - r2));
	result.z = 1.f - 2.f * r2;
	return result;
}
typedef struct { v3 weight; o_ray r; int depth; int show_lights, showenvmap, hadSS; } o_contrib;   /* Raytracer.h:15-23 */
/* One in-scattering event on the segment [0, t] of ray r: equi-angular (towards the sampled light point) or exponential
   distance sampling, direction uniform or towards the light (p = 1/2 each), one closest-hit query for the visibility of
   the light sample.  *attenuationFactor (the transmittance of the segment) is only written once the event is above
   the ground (:114): the caller keeps the previous value otherwise, as the reference's local does. */
static int fog_contribution(const o_ctx* c, const o_ray* r, v3 sampleLightPos, float t, v3 curWeight, int nbrebonds, int showLight, int hadSS,
                            o_contrib* newContrib, float* attenuationFactor, pcg32_t* rng, uint64_t* nrays2) {
	if (vnorm2(curWeight) < 1E-12) return 0;
	v3 rayDirection = r->direction;
	float p_uniform = 0.5f;
	int is_uniform_fog = (c->fog_type == 0);
	float alpha = c->fog_absorption;
	float sigmaT = c->fog_absorption_decay;
	int phase = c->fog_phase_type;
	float groundLevel = c->objs[2].max_translation.y;            /* objects[2]->get_translation()[1], no key frames */
	float int_ext;
	if (is_uniform_fog) int_ext = alpha * t * 0.05;
	else int_ext = alpha * int_exponential(r->origin.y, groundLevel, sigmaT, t, rayDirection.y);
	float T = expf(-int_ext);
	float proba_t, random_t;
	float clamped_t = 1000.f < t ? 1000.f : t;                   /* std::min(1000.f, t) */
	float a = vdot(vsub(sampleLightPos, r->origin), r->direction);
	if (a > 0) {                                                  /* equi-angular sampling (:71-84) */
		v3 projP = vadd(r->origin, vscale(a, r->direction));
		float D = sqrtf(vnorm2(vsub(sampleLightPos, projP)));
		float thetaA = -atan2f(a, D);
		float b = t - a;
		float thetaB = atan2f(b, D);
		float x = pcg_uniform(rng);
		random_t = D * tanf((1 - x) * thetaA + x * thetaB);
		proba_t = D / ((thetaB - thetaA) * (D * D + random_t * random_t));
		random_t += a;
	} else {                                                      /* :85-99 */
		float alpha2 = 5.f / clamped_t;
		do {
			random_t = -logf(pcg_uniform(rng)) / alpha2;
		} while (random_t > clamped_t);
		float normalization = 1.f / alpha2 * (1.f - expf(-alpha2 * clamped_t));
		proba_t = expf(-alpha2 * random_t) / normalization;
	}
	float int_ext_partielle;
	if (is_uniform_fog) int_ext_partielle = alpha * random_t * 0.05;
	else int_ext_partielle = alpha * int_exponential(r->origin.y, groundLevel, sigmaT, random_t, rayDirection.y);
	v3 random_P = vadd(r->origin, vscale(random_t, rayDirection));
	if (random_P.y < groundLevel) return 0;                       /* :114 */
	v3 random_dir;
	float proba_dir;
	v3 point_aleatoire = V(0, 0, 0);
	v3 axeOP = vnormalize(vsub(random_P, c->centerLight));
	int is_uniform;
	if (pcg_uniform(rng) < p_uniform) {
		random_dir = random_uniform_sphere(rng);
		is_uniform = 1;
	} else {
		v3 dir_aleatoire = random_cos_rng(axeOP, rng);
		point_aleatoire = vadd(vscale(c->radiusLight, dir_aleatoire), c->centerLight);
		random_dir = vnormalize(vsub(point_aleatoire, random_P));
		is_uniform = 0;
	}
	float phase_func = 0;
	float k = c->phase_aniso;
	switch (phase) {
	case 0: phase_func = 1. / (4. * M_PI); break;
	case 1: phase_func = (1 - k * k) / (4. * M_PI * (1 + k * vdot(random_dir, vneg(rayDirection)))); break;
	case 2: phase_func = 3 / (16 * M_PI) * (1 + sqrf(vdot(random_dir, rayDirection))); break;
	}
	o_ray L_ray; L_ray.origin = random_P; L_ray.direction = random_dir;
	o_mat interMat; mat_default(&interMat);
	v3 interP = V(0, 0, 0);
	int interid = -1, intertri = -1;
	float intert = 0;
	int interinter = scene_intersection(c, &L_ray, &interP, &interid, &intert, &interMat, &intertri);
	if (nrays2) nrays2[0]++;
	v3 interN = interMat.shadingN;
	float Vis;
	if (is_uniform) Vis = 1;
	else {
		float d_light2 = vnorm2(vsub(point_aleatoire, random_P));
		if (interinter && intert * intert < d_light2 * 0.99) Vis = 0; else Vis = 1;
	}
	*attenuationFactor = T;
	if (Vis == 0) return 0;
	float pdf_uniform = 1. / (4. * M_PI);
	float J = vdot(interN, vneg(random_dir)) / vnorm2(vsub(interP, random_P));
	float pdf_light = (interinter && interid == 0) ? (vdot(vnormalize(vsub(interP, c->centerLight)), axeOP) / (M_PI * sqrf(c->radiusLight)) / J) : 0.;
	proba_dir = p_uniform * pdf_uniform + (1 - p_uniform) * pdf_light;
	float ext;
	if (is_uniform_fog) ext = c->fog_density * 0.05;
	else ext = c->fog_density * expf(-c->fog_density_decay * (random_P.y - groundLevel));
	v3 newweight = vscale((phase_func * ext * expf(-int_ext_partielle) / (proba_t * proba_dir)), curWeight);
	newContrib->weight = newweight; newContrib->r = L_ray; newContrib->depth = nbrebonds - 1; newContrib->show_lights = showLight; newContrib->showenvmap = 1; newContrib->hadSS = hadSS;
	return 1;
}

#define O_SIZE_CIRC_ARRAY 200                                                                /* Raytracer.h:114 */
static v3 background_pixel(const o_ctx* c, int screenI, int screenJ) {                       /* :261-265 */
	int bi = (int)(screenI / (float)c->H * c->backgroundH); if (bi < 0) bi = 0; if (bi > c->backgroundH - 1) bi = c->backgroundH - 1;
	int bj = (int)(screenJ / (float)c->W * c->backgroundW); if (bj < 0) bj = 0; if (bj > c->backgroundW - 1) bj = c->backgroundW - 1;
	const float* px = c->background + ((size_t)bi * c->backgroundW + bj) * 3;
	return V(px[0], px[1], px[2]);
}
static v3 get_color_aov(const o_ctx* c, o_ray r, int sampleID, int screenI, int screenJ, pcg32_t* rng, uint64_t* nrays2, v3* normalValue, v3* albedoValue) {
	v3 color = V(0, 0, 0);
	o_mat mat; mat_default(&mat);
	v3 P = V(0, 0, 0);
	int sphere_id = -1, tri_id = -1;
	float t;
	/* the contributions of a sample wait in a circular FIFO (:213-238); without ghosts (and fog) a vertex queues at most one */
	o_contrib contribs[O_SIZE_CIRC_ARRAY];
	int contribIndexStart = 0, contribIndexEnd = 1;
	contribs[0].weight = V(1.f, 1.f, 1.f); contribs[0].r = r; contribs[0].depth = c->nb_bounces; contribs[0].show_lights = 1; contribs[0].showenvmap = 1; contribs[0].hadSS = 0;
#define PUSH(w_, ray_, depth_, lights_, env_) PUSHS(w_, ray_, depth_, lights_, env_, has_had_subsurface_interaction)
#define PUSHS(w_, ray_, depth_, lights_, env_, ss_) do { o_contrib* q_ = &contribs[contribIndexEnd]; q_->weight = (w_); q_->r = (ray_); q_->depth = (depth_); q_->show_lights = (lights_); q_->showenvmap = (env_); q_->hadSS = (ss_); \
		contribIndexEnd++; if (contribIndexEnd >= O_SIZE_CIRC_ARRAY) contribIndexEnd = 0; } while (0)
	const int has_fog = c->fog_density > 1E-8;                      /* :207 */
	float attenuationFactor = 0;                                    /* :206 (uninitialised there) */
	o_contrib newContrib;
#define FOG(ray_, lightpos_) do { if (fog_contribution(c, &(ray_), (lightpos_), t, pathWeight, nbrebonds, show_lights, has_had_subsurface_interaction, &newContrib, &attenuationFactor, rng, nrays2)) { \
		contribs[contribIndexEnd] = newContrib; contribIndexEnd++; if (contribIndexEnd >= O_SIZE_CIRC_ARRAY) contribIndexEnd = 0; } } while (0)
	const int has_dome = 1;                                         /* sphereEnv: object 1 is the environment sphere (loadScene) */
	const int has_backgroundimage = c->backgroundW > 0 && c->background != NULL;   /* :220 */
	while (contribIndexStart != contribIndexEnd) {
		const o_contrib cur = contribs[contribIndexStart];
		o_ray currentRay = cur.r;
		int nbrebonds = cur.depth;
		v3 pathWeight = cur.weight;
		int show_lights = cur.show_lights, show_envmap = cur.showenvmap;
		const int has_had_subsurface_interaction = cur.hadSS;
		contribIndexStart++; if (contribIndexStart >= O_SIZE_CIRC_ARRAY) contribIndexStart = 0;
		if (nbrebonds == 0) continue;                               /* :240 */
		if (vnorm2(pathWeight) < sqrf(0.01f)) continue;             /* :241 */
		int has_inter = scene_intersection(c, &currentRay, &P, &sphere_id, &t, &mat, &tri_id);   /* :251 */
		if (nrays2) nrays2[0]++;
		v3 N = mat.shadingN;
		if (has_inter && nbrebonds == c->nb_bounces) {              /* :255-258 */
			if (normalValue) *normalValue = N;
			if (albedoValue) *albedoValue = mat.Kd;
		}
		if (nbrebonds == c->nb_bounces && has_backgroundimage && (!has_inter || (has_inter && sphere_id == 1 && has_dome))) {   /* :260-268 */
			color = vadd(color, vmul(pathWeight, background_pixel(c, screenI, screenJ)));
			continue;
		}
		v3 rayDirection = currentRay.direction;
		if (!has_inter) { if (c->fog_density == 0) continue; else break; }   /* :654-657 */
		if (sphere_id == 1) {                                       /* :275-301 (no_envmap = false) */
			if (!show_envmap) { if (has_fog) FOG(currentRay, c->centerLight); continue; }
			if (has_fog) {
				FOG(currentRay, c->centerLight);
				color = vadd(color, vmul(vscale(c->envmap_intensity, vscale(attenuationFactor, pathWeight)), mat.Ke));
			} else color = vadd(color, vmul(vscale(c->envmap_intensity, pathWeight), mat.Ke));
			continue;
		}
		if (sphere_id == 0) {                                       /* :303-316 */
			v3 currentContrib = show_lights ? V(c->lightPower, c->lightPower, c->lightPower) : V(0.f, 0.f, 0.f);
			if (has_fog) {
				FOG(currentRay, c->centerLight);
				color = vadd(color, vmul(vscale(attenuationFactor, pathWeight), currentContrib));
			} else color = vadd(color, vmul(pathWeight, currentContrib));
			continue;
		}
		const o_obj* obj = &c->objs[sphere_id];
		const int is_subsurface = vnorm2(mat.Ksub) > 1E-8;           /* :271 */
		const float subsProba = (has_had_subsurface_interaction || !is_subsurface) ? 0.f : 0.6f;   /* :318 */
		const float inv1MSubsProba = 1.f / (1.f - subsProba);
		v3 subsW = V(inv1MSubsProba, inv1MSubsProba, inv1MSubsProba);
		int sub_interaction = 0;
		if (is_subsurface && (pcg_uniform(rng) < subsProba)) {      /* :324-404: leave through a random point of the same object nearby */
			sub_interaction = 1;
			const float invSubsProba = 1.f / subsProba;
			subsW = V(invSubsProba, invSubsProba, invSubsProba);
			const float sigmasub = 1.5f;
			const float diskR = sqrtf(12.46f) * sigmasub;
			float integ = 1.f - expf(-diskR * diskR / (2.f * sigmasub * sigmasub));
			float randR = sigmasub * sqrtf(-2.f * logf(1.f - pcg_uniform(rng) * integ));
			float randangle = pcg_uniform(rng) * 2.f * (float)M_PI;
			float gauss0 = randR * sinf(randangle), gauss1 = randR * cosf(randangle), gauss2 = randR;
			float gaussval = (1. / (sigmasub * sigmasub * 2.f * (float)M_PI)) * expf(-(gauss2 * gauss2) / (2.f * sigmasub * sigmasub));
			float pdfgauss = gaussval / integ;
			v3 Tg = get_tangent(N);
			v3 Tg2 = vcross(N, Tg);
			v3 PtaboveP = vadd(vadd(vadd(P, vscale(gauss0, Tg)), vscale(gauss1, Tg2)), vscale(diskR, N));
			float r1s = pcg_uniform(rng);
			v3 axis = vneg(N);
			float tmax;
			float hh = sqrtf(diskR * diskR - gauss2 * gauss2);
			v3 subsOrigin = vadd(PtaboveP, vscale((diskR - hh), vneg(N)));
			float wAxis;
			if (r1s < 0.5f) { wAxis = 0.5f; tmax = 2.f * hh; }
			else {
				wAxis = 0.25f;
				tmax = 2.f * gauss2;
				if (r1s < 0.75f) axis = Tg; else axis = Tg2;
				float r2s = pcg_uniform(rng);
				if (r2s < 0.5f) subsOrigin = vsub(subsOrigin, vscale(hh, N));
			}
			o_mat subsmat; mat_default(&subsmat);
			int substriid = -1;
			float subst;
			v3 localP2 = V(0, 0, 0);
			o_ray probe; probe.origin = subsOrigin; probe.direction = axis;
			int subsinter = scene_get_random_intersection(c, &probe, &localP2, sphere_id, &subst, &subsmat, &substriid, 0, tmax, rng);
			if (subsinter) {
				float chris = exp(-vnorm2(vsub(P, localP2)) / (2. * sigmasub * sigmasub));
				float sumpdfs = sqr_d(0.5 * vdot(subsmat.shadingN, N)) + sqr_d(0.25 * vdot(subsmat.shadingN, Tg)) + sqr_d(0.25 * vdot(subsmat.shadingN, Tg2));
				float pdfdisk = wAxis * fabsf(vdot(axis, subsmat.shadingN)) / sumpdfs;
				subsW = vscale(pdfdisk / fmaxf(pdfgauss, 0.05f) * chris, subsW);
				rayDirection = vnormalize(vsub(localP2, P));
				P = vadd(localP2, vscale(0.005f, subsmat.shadingN));
				if (r1s < 0.5f) subsW = vscale(2.f, subsW); else subsW = vscale(4.f, subsW);
				subsW = vmul(subsW, vdivs(mat.Ksub, (float)M_PI));
				mat = subsmat;
				N = mat.shadingN;
				tri_id = substriid;
			}
		}
		const int next_hadSS = sub_interaction ? 1 : has_had_subsurface_interaction;
		color = vadd(color, vscale(c->envmap_intensity, vmul(pathWeight, mat.Ke)));   /* :411 */
		if (obj->miroir) {                                          /* :413-436 */
			o_ray rayon_miroir;
			rayon_miroir.origin = vadd(P, vscale(0.001f, N));
			rayon_miroir.direction = vreflect(rayDirection, N);
			if (has_fog) { FOG(currentRay, c->centerLight); PUSH(vscale(attenuationFactor, pathWeight), rayon_miroir, nbrebonds - 1, show_lights, 1); }
			else PUSH(pathWeight, rayon_miroir, nbrebonds - 1, show_lights, 1);
			continue;
		}
		if (mat.transp) {                                           /* :438-489 */
			float n1 = 1.f;
			float n2 = mat.refr_index;
			v3 normale = N;
			o_ray new_ray;
			int entering = 1;
			if (vdot(rayDirection, N) > 0) { n1 = mat.refr_index; n2 = 1; normale = vneg(N); entering = 0; }
			float radical = 1.f - sqrf(n1 / n2) * (1.f - sqrf(vdot(normale, rayDirection)));
			if (radical > 0) {
				v3 direction_refracte = vsub(vscale((n1 / n2), vsub(rayDirection, vscale(vdot(rayDirection, normale), normale))), vscale(sqrtf(radical), normale));
				float R0 = sqrf((n1 - n2) / (n1 + n2));
				float R;
				if (entering) R = R0 + (1 - R0) * powf(1.f + vdot(rayDirection, N), 5.f);
				else R = R0 + (1 - R0) * powf(1.f - vdot(direction_refracte, N), 5.f);
				if (pcg_uniform(rng) < R) {
					new_ray.origin = vadd(P, vscale(0.001f, normale)); new_ray.direction = vreflect(rayDirection, N);
				} else {
					new_ray.origin = vsub(P, vscale(0.001f, normale)); new_ray.direction = direction_refracte;
				}
			} else {
				new_ray.origin = vadd(P, vscale(0.001f, normale)); new_ray.direction = vreflect(rayDirection, N);
			}
			if (has_fog) { FOG(currentRay, c->centerLight); PUSH(vscale(attenuationFactor, pathWeight), new_ray, nbrebonds - 1, show_lights, 1); }   /* :473-481 */
			else PUSH(pathWeight, new_ray, nbrebonds - 1, show_lights, 1);   /* :483-486: showenvmap takes its default, true */
			continue;
		}
		/* diffuse / glossy (:490-632) */
		v3 axeOP = vfast_normalize(vsub(P, c->centerLight));
		v3 dir_aleatoire = random_cos_rng(axeOP, rng);              /* :499, no_envmap = false */
		v3 point_aleatoire = vadd(vscale(c->radiusLight, dir_aleatoire), c->centerLight);
		v3 wi = vfast_normalize(vsub(point_aleatoire, P));
		float d_light2 = vnorm2(vsub(point_aleatoire, P));
		v3 Np = dir_aleatoire;
		o_ray ray_light; ray_light.origin = vadd(P, vscale(0.01f, wi)); ray_light.direction = wi;
		int isShadowed;
		if (vdot(mat.shadingN, wi) < 0) isShadowed = 1;
		else { isShadowed = scene_intersection_shadow_g(c, &ray_light, sqrtf(d_light2) - 0.01f, 1); if (nrays2) nrays2[1]++; }   /* :513: ghosts cast no shadow */
		v3 currentContrib = V(0, 0, 0);
		if (!isShadowed) {
			if (obj->ghost) {                                       /* :522-536: the path goes straight on through the ghost, at the same depth */
				v3 offset = vdot(N, rayDirection) > 0 ? N : vneg(N);
				currentRay.origin = vadd(vadd(P, vscale(0.001f, rayDirection)), vscale(0.001f, offset));   /* :531: currentRay itself is replaced, */
				currentRay.direction = rayDirection;                                                         /* the fog event below (:557) runs along it */
				PUSH(pathWeight, currentRay, nbrebonds, show_lights, show_envmap);
			}
			v3 BRDF;
			if (sub_interaction) BRDF = vdivs(mat.Ksub, (float)M_PI);   /* :540-541 */
			else BRDF = obj->merl ? merl_eval(obj->merl, wi, vneg(rayDirection), N) : phong_eval(&mat, wi, vneg(rayDirection), N);
			float J = vdot(Np, vneg(wi)) / d_light2;
			float proba = vdot(axeOP, dir_aleatoire) / (M_PI * c->radiusLight * c->radiusLight);   /* double, narrowed */
			if (!obj->ghost && proba > 0.f) {                       /* :547-553: no direct light on a ghost */
				currentContrib = vadd(currentContrib, vmul(vscale((c->lightPower * fmaxf(0.f, vdot(N, wi)) * J / proba), subsW), BRDF));
			}
		}
		if (has_fog) {                                              /* :557-565 */
			FOG(currentRay, point_aleatoire);
			color = vadd(color, vmul(vscale(attenuationFactor, pathWeight), currentContrib));
		} else color = vadd(color, vmul(pathWeight, currentContrib));      /* :566 */
		/* indirect (:570-632) */
		float proba_globale;
		int has_sampled_diffuse;
		float tmp;
		float r1 = modff(c->randomPerPixel[(screenI * c->W + screenJ) * 2 + 0] + c->samples2d[sampleID * 2 + 0], &tmp);
		float r2 = modff(c->randomPerPixel[(screenI * c->W + screenJ) * 2 + 1] + c->samples2d[sampleID * 2 + 1], &tmp);
		v3 direction_aleatoire;
		if (sub_interaction) {                                      /* :584-587 */
			direction_aleatoire = random_cos12(mat.shadingN, r1, r2);
			proba_globale = vdot(N, direction_aleatoire) / (float)M_PI;
			has_sampled_diffuse = 1;
		} else if (obj->merl) {   /* IsoMERLBRDF::sample (BRDF.h:198-203): cosine lobe, no engine draw */
			direction_aleatoire = random_cos12(N, r1, r2);
			proba_globale = vdot(N, direction_aleatoire) / (M_PI);
			has_sampled_diffuse = 0;
		} else direction_aleatoire = phong_sample(&mat, vneg(rayDirection), N, &proba_globale, r1, r2, &has_sampled_diffuse, rng);
		if (vdot(direction_aleatoire, N) < 0 || vdot(direction_aleatoire, vreflect(rayDirection, N)) < 0 || proba_globale <= 0) continue;   /* :593 */
		v3 BRDFindirect;
		if (sub_interaction) BRDFindirect = vdivs(mat.Ksub, (float)M_PI);   /* :603-604 */
		else BRDFindirect = obj->merl ? merl_eval(obj->merl, direction_aleatoire, vneg(rayDirection), N) : phong_eval(&mat, direction_aleatoire, vneg(rayDirection), N);
		v3 newpathWeight = vscale((vdot(N, direction_aleatoire) / proba_globale), vmul(vmul(pathWeight, subsW), BRDFindirect));   /* :611 */
		if (obj->ghost && has_backgroundimage) {                    /* :614-621: the photo shows through, tinted by what the ghost receives */
			v3 bg = background_pixel(c, screenI, screenJ);
			newpathWeight = vmul(newpathWeight, V(bg.x / 196964.699f, bg.y / 196964.699f, bg.z / 196964.699f));
		}
		o_ray rayon_aleatoire;
		rayon_aleatoire.origin = vadd(P, vscale(0.01f, direction_aleatoire));
		rayon_aleatoire.direction = direction_aleatoire;
		if (has_fog) PUSHS(vscale(attenuationFactor, newpathWeight), rayon_aleatoire, nbrebonds - 1, 0, (show_envmap && isShadowed && has_sampled_diffuse) || !obj->ghost, next_hadSS);   /* :626 */
		else PUSHS(newpathWeight, rayon_aleatoire, nbrebonds - 1, 0, (show_envmap && isShadowed && has_sampled_diffuse) || !obj->ghost, next_hadSS);   /* :629 */
	}
#undef PUSH
#undef PUSHS
#undef FOG
	return color;
}

/* ------------------------------------------------------------------ BVH build (TriangleMesh.cpp:843-885, 1029-1130) */
static void bbox_of(const o_mesh* g, int i0, int i1, v3* bmin, v3* bmax) {           /* build_bbox :843-858 */
	*bmax = g->vertices[g->indices[i0].vtx[0]];
	*bmin = g->vertices[g->indices[i0].vtx[0]];
	for (int i = i0; i < i1; i++) for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) {
		float v = vget(g->vertices[g->indices[i].vtx[c]], k);
		vset(bmin, k, fminf(vget(*bmin, k), v));
		vset(bmax, k, fmaxf(vget(*bmax, k), v));
	}
}
static v3 tri_center(const o_mesh* g, int i) {
	return vdivs(vadd(vadd(g->vertices[g->indices[i].vtx[0]], g->vertices[g->indices[i].vtx[1]]), g->vertices[g->indices[i].vtx[2]]), 3.f);
}
static void centers_bbox_of(const o_mesh* g, int i0, int i1, v3* bmin, v3* bmax) {   /* build_centers_bbox :861-875 */
	*bmax = tri_center(g, i0);
	*bmin = tri_center(g, i0);
	for (int i = i0; i < i1; i++) {
		v3 center = tri_center(g, i);
		for (int k = 0; k < 3; k++) {
			vset(bmin, k, fminf(vget(*bmin, k), vget(center, k)));
			vset(bmax, k, fmaxf(vget(*bmax, k), vget(center, k)));
		}
	}
}
static float bb_area(v3 bmin, v3 bmax) {                                             /* Geometry.h:55-58 */
	v3 s = vsub(bmax, bmin);
	return 2 * (s.x * s.y + s.x * s.z + s.y * s.z);
}
static v3 vmin3(v3 a, v3 b) { return V(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static v3 vmax3(v3 a, v3 b) { return V(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

/* One candidate plane of the loop at TriangleMesh.cpp:1062-1100: cost area_L*n_L + area_R*n_R */
static float split_cost(const o_mesh* g, int i0, int i1, int split_dim, float split_val) {
	v3 lmin = V(1E10, 1E10, 1E10), lmax = V(-1E10, -1E10, -1E10), rmin = V(1E10, 1E10, 1E10), rmax = V(-1E10, -1E10, -1E10);
	int nl = 0, nr = 0;
	for (int i = i0; i < i1; i++) {
		const o_idx* ix = &g->indices[i];
		float center_split_dim = (vget(g->vertices[ix->vtx[0]], split_dim) + vget(g->vertices[ix->vtx[1]], split_dim) + vget(g->vertices[ix->vtx[2]], split_dim)) / 3.f;  /* /3. narrowed == /3.f */
		if (center_split_dim <= split_val) {
			for (int c = 0; c < 3; c++) lmin = vmin3(lmin, g->vertices[ix->vtx[c]]);
			for (int c = 0; c < 3; c++) lmax = vmax3(lmax, g->vertices[ix->vtx[c]]);
			nl++;
		} else {
			for (int c = 0; c < 3; c++) rmin = vmin3(rmin, g->vertices[ix->vtx[c]]);
			for (int c = 0; c < 3; c++) rmax = vmax3(rmax, g->vertices[ix->vtx[c]]);
			nr++;
		}
	}
	return bb_area(lmin, lmax) * nl + bb_area(rmin, rmax) * nr;
}

/* build_bvh_recur (TriangleMesh.cpp:1029-1130).  The tree and the node order are the reference's; to keep
 * multi-million-triangle test scenes practical, a large right subtree is built as an OpenMP task into its
 * own node array (child references relative to it) and spliced behind the left subtree, which gives the
 * same preorder positions as the reference's push_back order; the 16 candidate planes of a large node are
 * costed concurrently and the first minimum in plane order is taken, as the serial loop does. */
typedef struct { o_node* a; int n, cap; } nodevec;
static void nv_push(nodevec* v, o_node n) {
	if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->a = (o_node*)realloc(v->a, sizeof(o_node) * (size_t)v->cap); }
	v->a[v->n++] = n;
}
static int g_build_fork = 1 << 15, g_build_planes = 1 << 18;
void o_set_build_thresholds(int fork_tris, int planes_tris) { g_build_fork = fork_tris; g_build_planes = planes_tris; }

static void build_bvh_recur(o_mesh* g, nodevec* out, int i0, int i1, int depth) {
	const int node = out->n;
	o_node n;
	bbox_of(g, i0, i1, &n.bmin, &n.bmax);
	n.fg = i0; n.fd = i1; n.isleaf = 1;
	nv_push(out, n);

	v3 cmin, cmax;
	centers_bbox_of(g, i0, i1, &cmin, &cmax);
	v3 diag = vsub(cmax, cmin);
	int split_dim;
	if ((diag.x >= diag.y) && (diag.x >= diag.z)) split_dim = 0;
	else if ((diag.y >= diag.x) && (diag.y >= diag.z)) split_dim = 1;
	else split_dim = 2;

	enum { max_tests = 16 };
	float cost[max_tests], factor[max_tests];
	for (int t = 0; t < max_tests; t++) factor[t] = (t + 1) / (float)(max_tests + 1);
	#pragma omp taskloop shared(cost, factor) if(i1 - i0 >= g_build_planes)
	for (int t = 0; t < max_tests; t++) cost[t] = split_cost(g, i0, i1, split_dim, vget(cmin, split_dim) + vget(diag, split_dim) * factor[t]);
	float best_split_factor = 0.5;
	float best_area_bb = INFINITY;   /* 1E50 narrowed */
	for (int t = 0; t < max_tests; t++) if (cost[t] < best_area_bb) { best_split_factor = factor[t]; best_area_bb = cost[t]; }
	float split_val = vget(cmin, split_dim) + vget(diag, split_dim) * best_split_factor;
	int pivot = i0 - 1;
	for (int i = i0; i < i1; i++) {
		const o_idx* ix = &g->indices[i];
		float center_split_dim = (vget(g->vertices[ix->vtx[0]], split_dim) + vget(g->vertices[ix->vtx[1]], split_dim) + vget(g->vertices[ix->vtx[2]], split_dim)) / 3.f;
		if (center_split_dim <= split_val) {
			pivot++;
			o_idx tmp = g->indices[i]; g->indices[i] = g->indices[pivot]; g->indices[pivot] = tmp;
			int tp = g->perm[i]; g->perm[i] = g->perm[pivot]; g->perm[pivot] = tp;
		}
	}
	if (pivot < i0 || pivot >= i1 - 1 || i1 <= i0 + 4) return;
	out->a[node].isleaf = 0;
	if (i1 - (pivot + 1) >= g_build_fork && pivot + 1 - i0 >= g_build_fork) {
		nodevec right = {0, 0, 0};
		#pragma omp task shared(right)
		build_bvh_recur(g, &right, pivot + 1, i1, depth + 1);
		out->a[node].fg = out->n;
		build_bvh_recur(g, out, i0, pivot + 1, depth + 1);
		#pragma omp taskwait
		const int off = out->n;
		out->a[node].fd = off;
		for (int k = 0; k < right.n; k++) { o_node r = right.a[k]; if (!r.isleaf) { r.fg += off; r.fd += off; } nv_push(out, r); }
		free(right.a);
	} else {
		out->a[node].fg = out->n;
		build_bvh_recur(g, out, i0, pivot + 1, depth + 1);
		out->a[node].fd = out->n;
		build_bvh_recur(g, out, pivot + 1, i1, depth + 1);
	}
}

/* setup_tangents (TriangleMesh.cpp:601-711), only what getMaterial's normal-map branch reads */
static void setup_tangents(o_mesh* g) {
	v3* tan1 = (v3*)calloc(g->nv, sizeof(v3));
	v3* tan2 = (v3*)calloc(g->nv, sizeof(v3));
	for (int i = 0; i < g->nf; i++) {
		int a = g->indices[i].vtx[0], b = g->indices[i].vtx[1], c = g->indices[i].vtx[2];
		if (g->indices[i].uv[0] == -1) continue;
		if (g->indices[i].uv[1] == -1) continue;
		if (g->indices[i].uv[2] == -1) continue;
		v3 vA = vsub(g->vertices[b], g->vertices[a]);
		v3 vB = vsub(g->vertices[c], g->vertices[a]);
		v3 sA = vsub(g->uvs[g->indices[i].uv[1]], g->uvs[g->indices[i].uv[0]]);
		v3 sB = vsub(g->uvs[g->indices[i].uv[2]], g->uvs[g->indices[i].uv[0]]);
		float det = (sA.x * sB.y - sB.x * sA.y);
		v3 sdir, tdir;
		if (det != 0) {
			sdir = vdivs(vsub(vscale(sB.y, vA), vscale(sA.y, vB)), det);
			tdir = vdivs(vsub(vscale(sA.x, vB), vscale(sB.x, vA)), det);
		} else {
			sdir = vscale(0.00001f, vA);
			tdir = vscale(0.00001f, vB);
		}
		tan1[a] = vadd(tan1[a], sdir); tan1[b] = vadd(tan1[b], sdir); tan1[c] = vadd(tan1[c], sdir);
		tan2[a] = vadd(tan2[a], tdir); tan2[b] = vadd(tan2[b], tdir); tan2[c] = vadd(tan2[c], tdir);
	}
	int* v2n = (int*)calloc(g->nv, sizeof(int));
	for (int i = 0; i < g->nf; i++) for (int c = 0; c < 3; c++) v2n[g->indices[i].vtx[c]] = g->indices[i].n[c];
	v3* tangents = (v3*)calloc(g->nv, sizeof(v3));
	for (int i = 0; i < g->nv; i++) {
		v3 N = vnormalize(g->normals[v2n[i]]);
		tangents[i] = vnormalize(vsub(tan1[i], vscale(vdot(tan1[i], N), N)));
	}
	g->tangent_soup = (v3*)malloc(sizeof(v3) * 3 * (size_t)g->nf);
	for (int i = 0; i < g->nf; i++) for (int c = 0; c < 3; c++) g->tangent_soup[i * 3 + c] = tangents[g->indices[i].vtx[c]];
	free(tan1); free(tan2); free(v2n); free(tangents);
}

/* ------------------------------------------------------------------ API: construction */
static o_obj* push_obj(o_ctx* c) {
	if (c->nobj == c->cap_obj) { c->cap_obj = c->cap_obj ? 2 * c->cap_obj : 8; c->objs = (o_obj*)realloc(c->objs, sizeof(o_obj) * c->cap_obj); }
	o_obj* o = &c->objs[c->nobj++];
	memset(o, 0, sizeof *o);
	o->scale = 1;                                              /* Geometry.h:244-251 Object() */
	o->mat_rotation[0] = o->mat_rotation[4] = o->mat_rotation[8] = 1;
	return o;
}

o_ctx* o_create(void) {
	o_ctx* c = (o_ctx*)calloc(1, sizeof *c);
	/* Raytracer::loadScene (Raytracer.cpp:1238-1274) */
	c->W = 1000; c->H = 800; c->nrays = 100; c->nb_bounces = 3; c->sigma_filter = 0.5f;
	c->cam_pos = V(0, 0, 50); c->cam_dir = V(0, 0, -1); c->cam_up = V(0, 1, 0);
	c->fov = 35 * M_PI / 180; c->focus = 50; c->aperture = 0.1;
	o_obj* slum = push_obj(c); slum->type = OT_SPHERE; slum->O = V(10, 23, 15); slum->R = 10; slum->R2 = slum->R * slum->R; slum->rotation_center = slum->O;
	o_obj* s2 = push_obj(c); s2->type = OT_SPHERE; s2->O = V(0, 0, 0); s2->R = 1000000; s2->R2 = s2->R * s2->R; s2->rotation_center = s2->O; s2->flip_normals = 1;
	o_obj* pl = push_obj(c); pl->type = OT_PLANE; pl->A = V(0, 0, 0); pl->vecN = V(0., 1., 0.); pl->max_translation = V(0., -27.3, 0.);
	c->intensite_lumiere = 1000000000 * 4. * M_PI / (4. * M_PI * slum->R * slum->R * M_PI);
	c->envmap_intensity = 1;
	/* cam.rotate(0, -22deg, 1) is applied by the caller through o_set_camera */
	return c;
}

/* s.addObject(new Sphere(O, R, mirror, normal_swapped)) (Geometry.h:849-873) */
int o_add_sphere(o_ctx* c, const float* O, float R, int mirror, int flip_normals) {
	o_obj* sp = push_obj(c);
	sp->type = OT_SPHERE; sp->O = V(O[0], O[1], O[2]); sp->R = R; sp->R2 = R * R; sp->rotation_center = sp->O;
	sp->miroir = mirror != 0; sp->flip_normals = flip_normals != 0;
	return c->nobj - 1;
}

static void free_mesh(o_mesh* g) {
	if (!g) return;
	free(g->vertices); free(g->normals); free(g->uvs); free(g->indices); free(g->soup); free(g->tangent_soup); free(g->perm); free(g->nodes); free(g);
}
void o_destroy(o_ctx* c) {
	for (int i = 0; i < c->nobj; i++) {
		o_obj* o = &c->objs[i];
		for (int s = 0; s < T_NSLOTS; s++) { for (int k = 0; k < o->ntex[s]; k++) free(o->tex[s][k].values); free(o->tex[s]); }
		free(o->envtex);
		free((void*)o->merl);
		free_mesh(o->mesh);
	}
	free(c->objs); free(c->randomPerPixel); free(c->samples2d); free(c->filter_integral); free(c);
}

void o_set_render(o_ctx* c, int W, int H, int nrays, int nb_bounces, float sigma_filter) {
	c->W = W; c->H = H; c->nrays = nrays; c->nb_bounces = nb_bounces; c->sigma_filter = sigma_filter;
}
void o_set_camera(o_ctx* c, const float* pos, const float* dir, const float* up, float fov, float focus, float aperture) {
	c->cam_pos = V(pos[0], pos[1], pos[2]); c->cam_dir = V(dir[0], dir[1], dir[2]); c->cam_up = V(up[0], up[1], up[2]);
	c->fov = fov; c->focus = focus; c->aperture = aperture;
}
void o_set_light(o_ctx* c, const float* center, float R, float intensite_lumiere) {
	o_obj* l = &c->objs[0];
	l->O = V(center[0], center[1], center[2]); l->R = R; l->R2 = R * R; l->rotation_center = l->O;
	c->intensite_lumiere = intensite_lumiere;
}
void o_set_envmap_intensity(o_ctx* c, float v) { c->envmap_intensity = v; }

/* TriMesh::init (TriangleMesh.cpp:718-841) + readOBJ's default material lists (:470-490) */
int o_add_mesh(o_ctx* c, int nv, const float* verts, int nn, const float* normals, int nt, const float* uvs,
               int nf, const int* fv, const int* fn, const int* ft, float scale, int center) {
	o_mesh* g = (o_mesh*)calloc(1, sizeof *g);
	g->nv = nv; g->nn = nn; g->nt = nt; g->nf = nf;
	g->vertices = (v3*)malloc(sizeof(v3) * (size_t)(nv > 0 ? nv : 1));
	g->normals = (v3*)malloc(sizeof(v3) * (size_t)(nn > 0 ? nn : 1));
	g->uvs = (v3*)malloc(sizeof(v3) * (size_t)(nt > 0 ? nt : 1));
	for (int i = 0; i < nv; i++) g->vertices[i] = V(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
	for (int i = 0; i < nn; i++) g->normals[i] = V(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
	for (int i = 0; i < nt; i++) g->uvs[i] = V(uvs[2 * i], uvs[2 * i + 1], 0);
	g->indices = (o_idx*)malloc(sizeof(o_idx) * (size_t)nf);
	for (int i = 0; i < nf; i++) {
		o_idx* ix = &g->indices[i];
		for (int k = 0; k < 3; k++) { ix->vtx[k] = fv[3 * i + k]; ix->n[k] = fn ? fn[3 * i + k] : -1; ix->uv[k] = ft ? ft[3 * i + k] : -1; }
		ix->group = 0;   /* no usemtl -> group 0 "Default" (:462-467) */
	}
	g->interp_normals = 1;
	/* :742-751 axis swap (x,y,z)->(-z,y,x) */
	for (int i = 0; i < nv; i++) { float t = g->vertices[i].x; g->vertices[i].x = g->vertices[i].z; g->vertices[i].z = t; g->vertices[i].x = -g->vertices[i].x; }
	for (int i = 0; i < nn; i++) { float t = g->normals[i].x; g->normals[i].x = g->normals[i].z; g->normals[i].z = t; g->normals[i].x = -g->normals[i].x; }
	v3 bmin = V(1E9, 1E9, 1E9), bmax = V(-1E9, -1E9, -1E9);
	for (int i = 0; i < nv; i++) { bmin = vmin3(bmin, g->vertices[i]); bmax = vmax3(bmax, g->vertices[i]); }
	if (center) {   /* :760-770, scaling = 1, offset = 0 */
		float s = fmaxf(bmax.x - bmin.x, fmaxf(bmax.y - bmin.y, bmax.z - bmin.z));
		v3 cc = vscale(0.5f, vadd(bmin, bmax));
		for (int i = 0; i < nv; i++) {
			g->vertices[i].x = (g->vertices[i].x - cc.x) / s * 1.f + 0.f;
			g->vertices[i].y = (g->vertices[i].y - cc.y) / s * 1.f + 0.f;
			g->vertices[i].z = (g->vertices[i].z - cc.z) / s * 1.f + 0.f;
		}
	}
	g->perm = (int*)malloc(sizeof(int) * (size_t)nf);
	for (int i = 0; i < nf; i++) g->perm[i] = i;
	/* build_bvh (:878-885) */
	bbox_of(g, 0, nf, &g->root_min, &g->root_max);
	{
		nodevec nv = {0, 0, 0};
		#pragma omp parallel
		#pragma omp single
		build_bvh_recur(g, &nv, 0, nf, 0);
		g->nodes = nv.a; g->nnodes = nv.n; g->cap_nodes = nv.cap;
	}
	bbox_of(g, 0, nf, &g->bb_min, &g->bb_max);
	/* triangle soup (:812-829), after the reorder */
	g->soup = (o_tri*)calloc((size_t)nf, sizeof(o_tri));
	for (int i = 0; i < nf; i++) {
		o_tri* T = &g->soup[i];
		v3 A = g->vertices[g->indices[i].vtx[0]], B = g->vertices[g->indices[i].vtx[1]], C = g->vertices[g->indices[i].vtx[2]];
		T->A = A; T->u = vsub(B, A); T->v = vsub(C, A); T->N = vcross(T->u, T->v);   /* TriangleMesh.h:70-78 */
		T->m11 = vnorm2(T->u); T->m22 = vnorm2(T->v); T->m12 = vdot(T->u, T->v);
		T->invdetm = 1.f / (T->m11 * T->m22 - T->m12 * T->m12);
		if (nn != 0) for (int k = 0; k < 3; k++) T->normals[k] = g->normals[g->indices[i].n[k]];
		if (nt != 0) for (int k = 0; k < 3; k++) { T->uvs[k][0] = g->uvs[g->indices[i].uv[k]].x; T->uvs[k][1] = g->uvs[g->indices[i].uv[k]].y; }
	}
	if (nt != 0) setup_tangents(g);
	else { g->tangent_soup = NULL; }

	o_obj* o = push_obj(c);
	o->type = OT_TRIMESH; o->mesh = g;
	o->rotation_center = vscale(0.5f, vadd(g->bb_min, g->bb_max));   /* :831-835 */
	/* readOBJ default lists, one entry per group (:470-480) */
	obj_push_tex(o, T_KD, V(0.5, 0.5, 0.5)); obj_push_tex(o, T_KS, V(0., 0., 0.)); obj_push_tex(o, T_NE, V(0., 0., 0.));
	obj_push_tex(o, T_NORMAL, V(0., 0., 1.)); obj_push_tex(o, T_ALPHA, V(1., 1., 1.)); obj_push_tex(o, T_REFR, V(1.3, 1.3, 1.3));
	obj_push_tex(o, T_TRANSP, V(1., 1., 1.)); obj_push_tex(o, T_KSUB, V(0., 0., 0.));
	/* GUI placement (mainApp.cpp:2402-2410): scale, bottom of the bbox on the ground plane */
	o->scale = scale;
	o->max_translation = V(0, c->objs[2].max_translation.y - (g->bb_min.y) * o->scale, 0);
	return c->nobj - 1;
}

/* Object::add_col_subsurface (a plane has no material lists until the GUI adds one) */
void o_add_col_subsurface(o_ctx* c, int obj, const float* rgb) { obj_push_tex(&c->objs[obj], T_KSUB, V(rgb[0], rgb[1], rgb[2])); }
/* Object::subsurface[grp] as a constant colour (add_col_subsurface, Geometry.h) */
void o_set_group_subsurface(o_ctx* c, int obj, int grp, const float* rgb) {
	o_obj* o = &c->objs[obj];
	if (grp >= 0 && grp < o->ntex[T_KSUB]) o->tex[T_KSUB][grp].multiplier = V(rgb[0], rgb[1], rgb[2]);
}
void o_set_lenticular(o_ctx* c, int on, int nb_images, float max_angle, int pixel_width) {
	c->is_lenticular = on != 0; c->lenticular_nb_images = nb_images; c->lenticular_max_angle = max_angle; c->lenticular_pixel_width = pixel_width;
}
/* Scene::fog_* (Geometry.h:1371-1377) */
void o_set_fog(o_ctx* c, float density, float absorption, float density_decay, float absorption_decay, int type, int phase_type, float phase_aniso) {
	c->fog_density = density; c->fog_absorption = absorption; c->fog_density_decay = density_decay; c->fog_absorption_decay = absorption_decay;
	c->fog_type = type; c->fog_phase_type = phase_type; c->phase_aniso = phase_aniso;
}
void o_set_object_ghost(o_ctx* c, int obj, int ghost) { c->objs[obj].ghost = ghost != 0; }
/* Scene::background as load_background leaves it (Geometry.h:1355-1363): W*H*3 floats, rows as in the file, already
   pow(v/255, gamma) * 196964.699 */
void o_set_background(o_ctx* c, const float* rgb, int W, int H) {
	free(c->background); c->background = NULL; c->backgroundW = c->backgroundH = 0;
	if (rgb && W > 0 && H > 0) {
		c->background = (float*)malloc(sizeof(float) * (size_t)W * H * 3);
		memcpy(c->background, rgb, sizeof(float) * (size_t)W * H * 3);
		c->backgroundW = W; c->backgroundH = H;
	}
}
void o_set_object_flags(o_ctx* c, int obj, int miroir, int flip_normals) { c->objs[obj].miroir = miroir != 0; c->objs[obj].flip_normals = flip_normals != 0; }
void o_set_group_material(o_ctx* c, int obj, int grp, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr) {
	o_obj* o = &c->objs[obj];
	if (grp < o->ntex[T_KD]) o->tex[T_KD][grp].multiplier = V(Kd[0], Kd[1], Kd[2]);
	if (grp < o->ntex[T_KS]) o->tex[T_KS][grp].multiplier = V(Ks[0], Ks[1], Ks[2]);
	if (grp < o->ntex[T_NE]) o->tex[T_NE][grp].multiplier = V(Ne[0], Ne[1], Ne[2]);
	if (grp < o->ntex[T_TRANSP]) o->tex[T_TRANSP][grp].multiplier = V(transp_col, transp_col, transp_col);
	if (grp < o->ntex[T_REFR]) o->tex[T_REFR][grp].multiplier = V(refr, refr, refr);
}
void o_add_group_material(o_ctx* c, int obj, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr) {
	o_obj* o = &c->objs[obj];
	obj_push_tex(o, T_KD, V(Kd[0], Kd[1], Kd[2])); obj_push_tex(o, T_KS, V(Ks[0], Ks[1], Ks[2])); obj_push_tex(o, T_NE, V(Ne[0], Ne[1], Ne[2]));
	obj_push_tex(o, T_TRANSP, V(transp_col, transp_col, transp_col)); obj_push_tex(o, T_REFR, V(refr, refr, refr));
}
void o_set_group_texture(o_ctx* c, int obj, int grp, int slot, int W, int H, const unsigned char* rgb) {
	o_obj* o = &c->objs[obj];
	if (grp >= o->ntex[slot]) return;
	o_tex* t = &o->tex[slot][grp];
	free(t->values);
	t->W = W; t->H = H;
	t->values = (float*)malloc(sizeof(float) * (size_t)W * H * 3);
	/* Object::set_alphamap / set_roughnessmap build a fresh Texture with multiplier (1,1,1)
	 * (Geometry.cpp:138-146); set_texture / set_specularmap / set_normalmap keep the multiplier (:60-86) */
	if (slot == T_ALPHA || slot == T_NE) t->multiplier = V(1., 1., 1.);
	/* load_image: copy then flip rows (utils.cpp:104-118) */
	for (int i = 0; i < H; i++) for (int j = 0; j < W; j++) for (int k = 0; k < 3; k++)
		t->values[((size_t)i * W + j) * 3 + k] = rgb[((size_t)(H - 1 - i) * W + j) * 3 + k];
	if (slot == T_NORMAL) {     /* Texture::loadNormals (BRDF.h:406-418) */
		for (size_t i = 0; i < (size_t)W * H; i++) {
			v3 v = vnormalize(V(t->values[i * 3] - 128, t->values[i * 3 + 1] - 128, t->values[i * 3 + 2] - 128));
			t->values[i * 3] = v.x; t->values[i * 3 + 1] = v.y; t->values[i * 3 + 2] = v.z;
		}
	} else {                    /* Texture::loadColors: /255.f, powf(.,2.2f) (BRDF.h:393-404) */
		for (size_t i = 0; i < (size_t)W * H * 3; i++) { float v = t->values[i]; v /= 255.f; t->values[i] = powf(v, 2.2f); }
	}
}
void o_set_brdf_merl(o_ctx* c, int obj, const double* table) {
	size_t n = (size_t)3 * MERL_TH * MERL_TD * MERL_PD / 2;
	double* t = (double*)malloc(n * sizeof(double));
	memcpy(t, table, n * sizeof(double));
	free((void*)c->objs[obj].merl);
	c->objs[obj].merl = t;
}
void o_merl_eval(const double* table, int n, const float* wi3, const float* wo3, const float* N3, float* out3) {
	for (int i = 0; i < n; i++) {
		v3 v = merl_eval(table, V(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), V(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), V(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]));
		out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
	}
}
void o_set_envmap(o_ctx* c, int W, int H, const unsigned char* rgb) {
	o_obj* s = &c->objs[1];
	free(s->envtex);
	s->envtex = (unsigned char*)malloc((size_t)W * H * 3);
	for (int i = 0; i < H; i++) memcpy(s->envtex + (size_t)i * W * 3, rgb + (size_t)(H - 1 - i) * W * 3, (size_t)W * 3);
	s->envW = W; s->envH = H; s->has_envmap = 1; s->flip_normals = 1;
}

/* ------------------------------------------------------------------ prepare_render (Raytracer.cpp:1276-1391) */
static double fast_exp(double y) {                                                   /* :1294-1299 */
	double d;
	int32_t w[2];
	w[0] = 0;
	w[1] = (int32_t)(1512775 * y + 1072632447);
	memcpy(&d, w, 8);
	return d;
}
static uint32_t reverse_bits(uint32_t n) {                                           /* :1302-1309 */
	n = (n << 16) | (n >> 16);
	n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
	n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
	n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
	n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
	return n;
}
static void extensible_lattice2d(uint32_t id, float* x, float* y) {                  /* :1311-1319 */
	uint32_t rid = reverse_bits(id);
	float phi_id = rid * pow(2.0, -32);
	float tmp;
	*x = modff((float)(phi_id * 1 + 0.456789123), &tmp);        /* only modf(float,float*) is viable: arg narrowed */
	*y = modff((float)(phi_id * 182667 + 0.123456789), &tmp);
}
static float sum_area_table(const float* sat, int sat_width, int i0, int i1, int j0, int j1) {   /* :1276-1291 */
	float term1 = 0; if (i0 > 0) term1 = sat[(i0 - 1) * sat_width + j1];
	float term2 = 0; if (j0 > 0) term2 = sat[i1 * sat_width + j0 - 1];
	float term3 = 0; if (i0 > 0 && j0 > 0) term3 = sat[(i0 - 1) * sat_width + j0 - 1];
	return sat[i1 * sat_width + j1] - term1 - term2 + term3;
}

void o_prepare(o_ctx* c) {
	pcg32_t e0; pcg_seed(&e0, 0);                                /* :1325-1327 engine[0] = pcg32(0) */
	free(c->randomPerPixel);
	c->randomPerPixel = (float*)malloc(sizeof(float) * 2 * (size_t)c->W * c->H);
	for (int i = 0; i < c->W * c->H; i++) {                      /* :1340-1344 */
		c->randomPerPixel[2 * i] = pcg_uniform(&e0);
		c->randomPerPixel[2 * i + 1] = pcg_uniform(&e0);
	}
	free(c->samples2d);
	c->samples2d = (float*)malloc(sizeof(float) * 2 * (size_t)c->nrays);
	for (int i = 0; i < c->nrays; i++) extensible_lattice2d((uint32_t)i, &c->samples2d[2 * i], &c->samples2d[2 * i + 1]);
	/* :1354-1374 filter tables */
	float sigma_filter = c->sigma_filter;
	c->filter_size = (int)ceilf(sigma_filter * 2);
	c->filter_total_width = 2 * c->filter_size + 1;
	int fs = c->filter_size, ftw = c->filter_total_width;
	free(c->filter_integral);
	c->filter_integral = (float*)malloc(sizeof(float) * (size_t)ftw * ftw);
	for (int i = -fs; i <= fs; i++) for (int j = -fs; j <= fs; j++) {
		float integ = 0;
		for (int i2 = -fs; i2 <= i; i2++) for (int j2 = -fs; j2 <= j; j2++) {
			float w = fast_exp(-(i2 * i2 + j2 * j2) / (2. * sigma_filter * sigma_filter)) / (sigma_filter * sigma_filter * 2. * M_PI);
			integ += w;
		}
		c->filter_integral[(i + fs) * ftw + (j + fs)] = integ;
	}
	for (int i = 0; i < c->nobj; i++) build_matrix(&c->objs[i]);   /* Scene::prepare_render, Geometry.cpp:280-284 */
	c->centerLight = apply_transformation(&c->objs[0], c->objs[0].O);   /* :1377 */
	c->lum_scale = c->objs[0].scale;
	c->radiusLight = c->lum_scale * c->objs[0].R;
	c->lightPower = c->intensite_lumiere / sqrf(c->lum_scale);
}

/* ------------------------------------------------------------------ dumps */
void o_get_light(o_ctx* c, float* out5) { out5[0] = c->centerLight.x; out5[1] = c->centerLight.y; out5[2] = c->centerLight.z; out5[3] = c->radiusLight; out5[4] = c->lightPower; }
void o_get_tables(o_ctx* c, float* rpp, float* s2d, float* fi, int* filter_size) {
	if (rpp) memcpy(rpp, c->randomPerPixel, sizeof(float) * 2 * (size_t)c->W * c->H);
	if (s2d) memcpy(s2d, c->samples2d, sizeof(float) * 2 * (size_t)c->nrays);
	if (fi) memcpy(fi, c->filter_integral, sizeof(float) * (size_t)c->filter_total_width * c->filter_total_width);
	if (filter_size) *filter_size = c->filter_size;
}
void o_get_object_matrices(o_ctx* c, int obj, float* t, float* inv, float* r) {
	memcpy(t, c->objs[obj].trans, 48); memcpy(inv, c->objs[obj].inv, 48); memcpy(r, c->objs[obj].rot, 36);
}
void o_mesh_counts(o_ctx* c, int obj, int* ntri, int* nnodes, int* nverts, int* nnormals, int* nuvs) {
	o_mesh* g = c->objs[obj].mesh;
	*ntri = g->nf; *nnodes = g->nnodes; *nverts = g->nv; *nnormals = g->nn; *nuvs = g->nt;
}
void o_mesh_dump(o_ctx* c, int obj, int* perm, int* nodes_i, float* nodes_bb, float* soup, int* groups, float* root_bb) {
	o_mesh* g = c->objs[obj].mesh;
	for (int i = 0; i < g->nf; i++) {
		if (perm) perm[i] = g->perm[i];
		if (groups) groups[i] = g->indices[i].group;
		if (soup) {
			const o_tri* T = &g->soup[i];
			float* o = soup + (size_t)i * 31;
			o[0] = T->A.x; o[1] = T->A.y; o[2] = T->A.z; o[3] = T->u.x; o[4] = T->u.y; o[5] = T->u.z;
			o[6] = T->v.x; o[7] = T->v.y; o[8] = T->v.z; o[9] = T->N.x; o[10] = T->N.y; o[11] = T->N.z;
			o[12] = T->m11; o[13] = T->m12; o[14] = T->m22; o[15] = T->invdetm;
			for (int k = 0; k < 3; k++) { o[16 + 2 * k] = T->uvs[k][0]; o[17 + 2 * k] = T->uvs[k][1]; }
			for (int k = 0; k < 3; k++) { o[22 + 3 * k] = T->normals[k].x; o[23 + 3 * k] = T->normals[k].y; o[24 + 3 * k] = T->normals[k].z; }
		}
	}
	for (int i = 0; i < g->nnodes; i++) {
		if (nodes_i) { nodes_i[3 * i] = g->nodes[i].isleaf; nodes_i[3 * i + 1] = g->nodes[i].fg; nodes_i[3 * i + 2] = g->nodes[i].fd; }
		if (nodes_bb) { float* b = nodes_bb + 6 * (size_t)i; b[0] = g->nodes[i].bmin.x; b[1] = g->nodes[i].bmin.y; b[2] = g->nodes[i].bmin.z; b[3] = g->nodes[i].bmax.x; b[4] = g->nodes[i].bmax.y; b[5] = g->nodes[i].bmax.z; }
	}
	if (root_bb) { root_bb[0] = g->root_min.x; root_bb[1] = g->root_min.y; root_bb[2] = g->root_min.z; root_bb[3] = g->root_max.x; root_bb[4] = g->root_max.y; root_bb[5] = g->root_max.z; }
}

/* ------------------------------------------------------------------ leaf-function API */
void o_pcg32(uint64_t seed, int n, uint32_t* out) { pcg32_t g; pcg_seed(&g, seed); for (int i = 0; i < n; i++) out[i] = pcg_next(&g); }
void o_lattice(int n, float* out_xy) { for (int i = 0; i < n; i++) extensible_lattice2d((uint32_t)i, &out_xy[2 * i], &out_xy[2 * i + 1]); }
void o_invsqroot(int n, const float* in, float* out) { for (int i = 0; i < n; i++) out[i] = inv_sq_root(in[i]); }
void o_fast_normalize(int n, const float* in3, float* out3) {
	for (int i = 0; i < n; i++) { v3 v = vfast_normalize(V(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2])); out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z; }
}
void o_fast_exp(int n, const double* in, double* out) { for (int i = 0; i < n; i++) out[i] = fast_exp(in[i]); }
void o_random_cos(int n, const float* N3, const float* r12, float* out3) {
	for (int i = 0; i < n; i++) { v3 d = random_cos12(V(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]), r12[2 * i], r12[2 * i + 1]); out3[3 * i] = d.x; out3[3 * i + 1] = d.y; out3[3 * i + 2] = d.z; }
}
void o_camera_rays(o_ctx* c, int n, const int* ij, const float* jit4, float* out6) {
	for (int q = 0; q < n; q++) {
		o_ray r = generate_direction(c, c->double_frustum_start_t, ij[2 * q], ij[2 * q + 1], jit4[4 * q], jit4[4 * q + 1], jit4[4 * q + 2], jit4[4 * q + 3], c->W, c->H);
		out6[6 * q] = r.origin.x; out6[6 * q + 1] = r.origin.y; out6[6 * q + 2] = r.origin.z;
		out6[6 * q + 3] = r.direction.x; out6[6 * q + 4] = r.direction.y; out6[6 * q + 5] = r.direction.z;
	}
}
void o_intersect(o_ctx* c, int n, const float* rays6, int* out_i, float* out_f) {
	for (int q = 0; q < n; q++) {
		o_ray r; r.origin = V(rays6[6 * q], rays6[6 * q + 1], rays6[6 * q + 2]); r.direction = V(rays6[6 * q + 3], rays6[6 * q + 4], rays6[6 * q + 5]);
		v3 P = V(0, 0, 0); o_mat mat; mat_default(&mat); int id = -1, tri = -1; float t;
		int h = scene_intersection(c, &r, &P, &id, &t, &mat, &tri);
		out_i[3 * q] = h ? 1 : 0; out_i[3 * q + 1] = h ? id : -1; out_i[3 * q + 2] = h ? tri : -1;
		float* o = out_f + 20 * (size_t)q;
		o[0] = t; o[1] = P.x; o[2] = P.y; o[3] = P.z;
		o[4] = mat.shadingN.x; o[5] = mat.shadingN.y; o[6] = mat.shadingN.z;
		o[7] = mat.Kd.x; o[8] = mat.Kd.y; o[9] = mat.Kd.z; o[10] = mat.Ks.x; o[11] = mat.Ks.y; o[12] = mat.Ks.z;
		o[13] = mat.Ne.x; o[14] = mat.Ne.y; o[15] = mat.Ne.z; o[16] = mat.Ke.x; o[17] = mat.Ke.y; o[18] = mat.Ke.z;
		o[19] = h ? (mat.transp ? -mat.refr_index : mat.refr_index) : 0.f;
	}
	cnt_flush();
}
void o_intersect_shadow(o_ctx* c, int n, const float* rays6, const float* dist_light, int* occluded) {
	for (int q = 0; q < n; q++) {
		o_ray r; r.origin = V(rays6[6 * q], rays6[6 * q + 1], rays6[6 * q + 2]); r.direction = V(rays6[6 * q + 3], rays6[6 * q + 4], rays6[6 * q + 5]);
		occluded[q] = scene_intersection_shadow(c, &r, dist_light[q]);
	}
	cnt_flush();
}
void o_phong_sample(int n, const float* mat9, const float* wo3, const float* N3, const float* r12, const uint64_t* seed, float* out5) {
	for (int i = 0; i < n; i++) {
		o_mat m; mat_default(&m);
		m.Kd = V(mat9[9 * i], mat9[9 * i + 1], mat9[9 * i + 2]); m.Ks = V(mat9[9 * i + 3], mat9[9 * i + 4], mat9[9 * i + 5]); m.Ne = V(mat9[9 * i + 6], mat9[9 * i + 7], mat9[9 * i + 8]);
		pcg32_t g; pcg_seed(&g, seed[i]);
		float pdf; int diff;
		v3 d = phong_sample(&m, V(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), V(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]), &pdf, r12[2 * i], r12[2 * i + 1], &diff, &g);
		out5[5 * i] = d.x; out5[5 * i + 1] = d.y; out5[5 * i + 2] = d.z; out5[5 * i + 3] = pdf; out5[5 * i + 4] = diff ? 1.f : 0.f;
	}
}
void o_phong_eval(int n, const float* mat9, const float* wi3, const float* wo3, const float* N3, float* out3) {
	for (int i = 0; i < n; i++) {
		o_mat m; mat_default(&m);
		m.Kd = V(mat9[9 * i], mat9[9 * i + 1], mat9[9 * i + 2]); m.Ks = V(mat9[9 * i + 3], mat9[9 * i + 4], mat9[9 * i + 5]); m.Ne = V(mat9[9 * i + 6], mat9[9 * i + 7], mat9[9 * i + 8]);
		v3 v = phong_eval(&m, V(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), V(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), V(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]));
		out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
	}
}

static v3 get_color(const o_ctx* c, o_ray r, int sampleID, int screenI, int screenJ, pcg32_t* rng, uint64_t* nrays2) {
	return get_color_aov(c, r, sampleID, screenI, screenJ, rng, nrays2, NULL, NULL);
}

/* ------------------------------------------------------------------ radiance API */
/* one (pixel, sample): seeding rule + the 4 camera draws of Raytracer.cpp:1462-1466 */
static v3 sample_radiance(const o_ctx* c, int i, int j, int k, float* dx_out, float* dy_out, uint64_t* nrays2) {
	pcg32_t rng;
	uint64_t p = (uint64_t)i * (uint64_t)c->W + (uint64_t)j;
	pcg_seed(&rng, p * 65536ull + (uint64_t)k);
	float dx = pcg_uniform(&rng) - 0.5f;
	float dy = pcg_uniform(&rng) - 0.5f;
	float dx_aperture = (pcg_uniform(&rng) - 0.5f) * c->aperture;
	float dy_aperture = (pcg_uniform(&rng) - 0.5f) * c->aperture;
	o_ray r = generate_direction(c, c->double_frustum_start_t, i, j, dx, dy, dx_aperture, dy_aperture, c->W, c->H);
	*dx_out = dx; *dy_out = dy;
	return get_color(c, r, k, i, j, &rng, nrays2);
}

/* per-sample denoiser inputs: getColor's normalValue / albedoValue beside the colour */
static v3 sample_radiance_aov(const o_ctx* c, int i, int j, int k, v3* normal, v3* albedo) {
	pcg32_t rng;
	uint64_t p = (uint64_t)i * (uint64_t)c->W + (uint64_t)j;
	pcg_seed(&rng, p * 65536ull + (uint64_t)k);
	float dx = pcg_uniform(&rng) - 0.5f;
	float dy = pcg_uniform(&rng) - 0.5f;
	float dx_aperture = (pcg_uniform(&rng) - 0.5f) * c->aperture;
	float dy_aperture = (pcg_uniform(&rng) - 0.5f) * c->aperture;
	o_ray r = generate_direction(c, c->double_frustum_start_t, i, j, dx, dy, dx_aperture, dy_aperture, c->W, c->H);
	*normal = V(0, 0, 0); *albedo = V(0, 0, 0);                   /* Vector normal, albedo; (:1628) */
	return get_color_aov(c, r, k, i, j, &rng, NULL, normal, albedo);
}
void o_getcolor_samples_aov(o_ctx* c, int npix, const int* ij, int k0, int k1, float* out_rgb, float* out_normal, float* out_albedo) {
	#pragma omp parallel for schedule(dynamic, 256) if(npix >= 4096)
	for (int q = 0; q < npix; q++) for (int k = k0; k < k1; k++) {
		v3 n, a;
		v3 col = sample_radiance_aov(c, ij[2 * q], ij[2 * q + 1], k, &n, &a);
		size_t o = (size_t)q * (size_t)(k1 - k0) + (size_t)(k - k0);
		out_rgb[3 * o] = col.x; out_rgb[3 * o + 1] = col.y; out_rgb[3 * o + 2] = col.z;
		out_normal[3 * o] = n.x; out_normal[3 * o + 1] = n.y; out_normal[3 * o + 2] = n.z;
		out_albedo[3 * o] = a.x; out_albedo[3 * o + 1] = a.y; out_albedo[3 * o + 2] = a.z;
	}
	#pragma omp parallel
	cnt_flush();
}
/* has_denoiser accumulation of render_image_nopreviz (Raytracer.cpp:1631-1645): no splat, every sample adds its colour,
   normal and albedo to its own pixel and 1 to the sample count.  Sums only (the caller divides, :1689-1696); `normal`
   receives the sum of the shading normals — the reference's normalImage adds imagedoublethreads instead (:1680). */
void o_render_denoiser_inputs(o_ctx* c, float* imagedouble, float* sample_count, float* albedo, float* normal) {
	const int W = c->W, H = c->H;
	memset(imagedouble, 0, sizeof(float) * (size_t)W * H * 3);
	memset(albedo, 0, sizeof(float) * (size_t)W * H * 3);
	memset(normal, 0, sizeof(float) * (size_t)W * H * 3);
	memset(sample_count, 0, sizeof(float) * (size_t)W * H);
	#pragma omp parallel for schedule(dynamic, 1)
	for (int i = 0; i < H; i++) for (int j = 0; j < W; j++) for (int k = 0; k < c->nrays; k++) {
		v3 n, a;
		v3 color = sample_radiance_aov(c, i, j, k, &n, &a);
		const int idx = ((H - i - 1) * W + j) * 3;
		imagedouble[idx + 0] += color.x; imagedouble[idx + 1] += color.y; imagedouble[idx + 2] += color.z;
		sample_count[(H - i - 1) * W + j] += 1;
		normal[idx + 0] += n.x; normal[idx + 1] += n.y; normal[idx + 2] += n.z;
		albedo[idx + 0] += a.x; albedo[idx + 1] += a.y; albedo[idx + 2] += a.z;
	}
	#pragma omp parallel
	cnt_flush();
}

/* diagnostic: the traversal events of the samples k0 .. k1-1 of the given pixels, one camera path after the other on the calling
   thread (sample index outermost, as the device numbers its path slots).  Returns the bytes the trace needs (<= cap: complete). */
unsigned long long o_trace_samples(o_ctx* c, int npix, const int* ij, int k0, int k1, uint8_t* buf, unsigned long long cap) {
	o_trace_set(buf, cap);
	for (int k = k0; k < k1; k++) for (int q = 0; q < npix; q++) {
		float dx, dy;
		trace_byte(0xFD);
		(void)sample_radiance(c, ij[2 * q], ij[2 * q + 1], k, &dx, &dy, NULL);
	}
	size_t n = tl_trace_len;
	o_trace_set(NULL, 0);
	return n;
}

void o_getcolor_samples(o_ctx* c, int npix, const int* ij, int k0, int k1, float* out_rgb, float* out_dxdy) {
	/* samples are independent (own pcg32 stream each): large requests run on all host threads */
	#pragma omp parallel for schedule(dynamic, 256) if(npix >= 4096)
	for (int q = 0; q < npix; q++) for (int k = k0; k < k1; k++) {
		float dx, dy;
		v3 col = sample_radiance(c, ij[2 * q], ij[2 * q + 1], k, &dx, &dy, NULL);
		size_t o = (size_t)q * (size_t)(k1 - k0) + (size_t)(k - k0);
		out_rgb[3 * o] = col.x; out_rgb[3 * o + 1] = col.y; out_rgb[3 * o + 2] = col.z;
		if (out_dxdy) { out_dxdy[2 * o] = dx; out_dxdy[2 * o + 1] = dy; }
	}
	#pragma omp parallel
	cnt_flush();
}

/* splat of one sample (Raytracer.cpp:1477-1497) into row-flipped buffers */
static inline void splat(const o_ctx* c, float* imagedouble, float* sample_count, int i, int j, float dx, float dy, v3 color,
                         int bmin_i, int bmax_i, int bmin_j, int bmax_j, float denom1, float denom2) {
	const int W = c->W, H = c->H;
	for (int i2 = bmin_i; i2 <= bmax_i; i2++) for (int j2 = bmin_j; j2 <= bmax_j; j2++) {
		float w = fast_exp(-(sqrf(i2 - i - dy) + sqrf(j2 - j - dx)) * denom2) * denom1;
		imagedouble[((H - i2 - 1) * W + j2) * 3 + 0] += color.x * w;
		imagedouble[((H - i2 - 1) * W + j2) * 3 + 1] += color.y * w;
		imagedouble[((H - i2 - 1) * W + j2) * 3 + 2] += color.z * w;
		sample_count[(H - i2 - 1) * W + j2] += w;
	}
}
static inline void splat_setup(const o_ctx* c, int i, int j, int* bmin_i, int* bmax_i, int* bmin_j, int* bmax_j, float* denom1) {
	const int fs = c->filter_size, ftw = c->filter_total_width;
	*bmin_i = i - fs > 0 ? i - fs : 0;
	*bmax_i = i + fs < c->H - 1 ? i + fs : c->H - 1;
	*bmin_j = j - fs > 0 ? j - fs : 0;
	*bmax_j = j + fs < c->W - 1 ? j + fs : c->W - 1;
	float ratio = 1.f / sum_area_table(c->filter_integral, ftw, *bmin_i - i + fs, *bmax_i - i + fs, *bmin_j - j + fs, *bmax_j - j + fs);
	*denom1 = ratio / (c->sigma_filter * c->sigma_filter * 2. * M_PI);
}

void o_render_seeded(o_ctx* c, float* imagedouble, float* sample_count) {
	const int W = c->W, H = c->H;
	float denom2 = 1.f / (2. * c->sigma_filter * c->sigma_filter);   /* Raytracer.cpp:1430 */
	memset(imagedouble, 0, sizeof(float) * (size_t)W * H * 3);
	memset(sample_count, 0, sizeof(float) * (size_t)W * H);
	for (int i = 0; i < H; i++) for (int j = 0; j < W; j++) {
		int bmin_i, bmax_i, bmin_j, bmax_j; float denom1;
		splat_setup(c, i, j, &bmin_i, &bmax_i, &bmin_j, &bmax_j, &denom1);
		for (int k = 0; k < c->nrays; k++) {
			float dx, dy;
			v3 color = sample_radiance(c, i, j, k, &dx, &dy, NULL);
			splat(c, imagedouble, sample_count, i, j, dx, dy, color, bmin_i, bmax_i, bmin_j, bmax_j, denom1, denom2);
		}
	}
	cnt_flush();
}

double o_render_omp(o_ctx* c, int threads, float* imagedouble, float* sample_count, uint64_t* rays_out) {
	const int W = c->W, H = c->H;
	const int batchWidth = 4, batchHeight = 4;                    /* Raytracer.cpp:1569-1572 */
	const int nbBatchX = (int)ceilf(W / (float)batchWidth);
	const int nbBatchY = (int)ceilf(H / (float)batchHeight);
	float denom2 = 1.f / (2. * c->sigma_filter * c->sigma_filter);
	if (threads < 1) threads = 1;
	float* img_t = (float*)calloc((size_t)W * H * 3 * threads, sizeof(float));
	float* cnt_t = (float*)calloc((size_t)W * H * threads, sizeof(float));
	uint64_t rays[2] = { 0, 0 };
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel num_threads(threads)
	{
		int tid = omp_get_thread_num();
		float* cur_img = img_t + (size_t)tid * W * H * 3;
		float* cur_cnt = cnt_t + (size_t)tid * W * H;
		uint64_t my_rays[2] = { 0, 0 };
#pragma omp for schedule(dynamic, 1)
		for (int batchid = 0; batchid < nbBatchX * nbBatchY; batchid++) {
			int batchi = batchid / nbBatchX, batchj = batchid % nbBatchX;
			int i_end = batchi * batchHeight + batchHeight < H ? batchi * batchHeight + batchHeight : H;
			int j_end = batchj * batchWidth + batchWidth < W ? batchj * batchWidth + batchWidth : W;
			for (int i = batchi * batchHeight; i < i_end; i++) for (int j = batchj * batchWidth; j < j_end; j++) {
				int bmin_i, bmax_i, bmin_j, bmax_j; float denom1;
				splat_setup(c, i, j, &bmin_i, &bmax_i, &bmin_j, &bmax_j, &denom1);
				for (int k = 0; k < c->nrays; k++) {
					float dx, dy;
					v3 color = sample_radiance(c, i, j, k, &dx, &dy, my_rays);
					splat(c, cur_img, cur_cnt, i, j, dx, dy, color, bmin_i, bmax_i, bmin_j, bmax_j, denom1, denom2);
				}
			}
		}
		__atomic_fetch_add(&rays[0], my_rays[0], __ATOMIC_RELAXED);
		__atomic_fetch_add(&rays[1], my_rays[1], __ATOMIC_RELAXED);
		cnt_flush();
	}
	/* serial reduce (Raytracer.cpp:1669-1685) */
	memset(imagedouble, 0, sizeof(float) * (size_t)W * H * 3);
	memset(sample_count, 0, sizeof(float) * (size_t)W * H);
	for (int th = 0; th < threads; th++) for (size_t i = 0; i < (size_t)W * H; i++) {
		imagedouble[i * 3] += img_t[(size_t)th * W * H * 3 + i * 3];
		imagedouble[i * 3 + 1] += img_t[(size_t)th * W * H * 3 + i * 3 + 1];
		imagedouble[i * 3 + 2] += img_t[(size_t)th * W * H * 3 + i * 3 + 2];
		sample_count[i] += cnt_t[(size_t)th * W * H + i];
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(img_t); free(cnt_t);
	if (rays_out) { rays_out[0] = rays[0]; rays_out[1] = rays[1]; }
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

int o_max_threads(void) { return omp_get_num_procs(); }
