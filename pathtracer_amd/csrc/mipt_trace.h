// mipt_trace.h — device-side scene intersection: AABB slab tests, triangle test, ordered BVH
// traversal (closest hit and any hit), sphere / plane primitives, material lookup.
//
// The traversal reproduces the reference's visiting ORDER (near child first, ties: right child
// first; TriangleMesh.cpp:1180-1187) and its pruning rules exactly, because the closest hit keeps
// the FIRST triangle that reaches the minimum t (strict <, TriangleMesh.cpp:1197): a different
// order could pick a different coplanar / shared-edge triangle (SURVEY.md §7 "hard parts").
#pragma once
#include "mipt_math.h"
#include "mipt_scene.h"

struct Ray { f3 o, d; };

struct Mat {                       // MaterialValues (BRDF.h:7-20)
	f3 shadingN, Kd, Ks, Ne, Ke;
	f3 Ksub;                       // filled by scene_intersect_inherit only (elsewhere the vertex logic asks hit_ksub)
	bool transp;
	float refr_index;
	// of the object that was hit (not MaterialValues: read with the object's first 64 bytes by hit_material_obj, so that the
	// vertex logic does not go back to the descriptor for them)
	int miroir;                    // DObject::miroir: bit 0 mirror, bit 1 ghost (ghost scenes are the queue pipeline's: only it looks at the bit)
	const double* merl;
};

// The first 64 bytes of a DObject (mipt_scene.h) in registers: four 16-byte loads issued together.
struct ObjHot {
	int type, miroir, flip_normals, interp_normals, nuvs, ngroups, ntex_normal, alpha_test;
	const DTriShade* shade; const DGroupMat* gmat; const float* tangent_soup; const double* merl;
};
MIPT_DEV ObjHot load_obj_hot(const DObject& o) {
	typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
	const __attribute__((address_space(1))) v4u_* q = (const __attribute__((address_space(1))) v4u_*)(&o);
	const v4u_ a = q[0], b = q[1], c = q[2], d = q[3];
	ObjHot h;
	h.type = (int)a.x; h.miroir = (int)a.y; h.flip_normals = (int)a.z; h.interp_normals = (int)a.w;
	h.nuvs = (int)b.x; h.ngroups = (int)b.y; h.ntex_normal = (int)b.z; h.alpha_test = (int)b.w;
	h.shade = (const DTriShade*)(((uint64_t)c.y << 32) | c.x); h.gmat = (const DGroupMat*)(((uint64_t)c.w << 32) | c.z);
	h.tangent_soup = (const float*)(((uint64_t)d.y << 32) | d.x); h.merl = (const double*)(((uint64_t)d.w << 32) | d.z);
	return h;
}

// Closest-hit record carried through Scene::intersection; the material is evaluated once, for
// the winning object (see scene_intersect).
struct Hit {
	int obj;        // -1 = miss
	int tri;
	float t;
	float beta, gamma;
};

// The fourth word of a closest-hit record in HBM (wf.hit[].w, and the .w a fresh ray carries in its direction):
//   MIPT_HIT_MISS                     nothing hit
//   MIPT_HIT_ANALYTIC | object        a sphere / plane: any object index (Scene::objects is an unbounded vector, Geometry.h:1306-1309)
//   scene-wide triangle index         a mesh triangle (bit 31 clear; < 2^26): the object is the mesh whose range of the scene's
//                                     triangle buffer holds it — the only mesh in every BASELINE config (no search, no load),
//                                     else a binary search over the meshes' first triangles (DScene::mesh_first)
// Until round 6 the word was object << 27 | mesh-local triangle: 31 objects at most.
#define MIPT_HIT_MISS 0xffffffffu
#define MIPT_HIT_ANALYTIC 0x80000000u
MIPT_DEV bool hit_unpack(const DScene* __restrict__ sc, unsigned packed, int& obj, int& tri) {
	obj = -1; tri = -1;
	if (packed == MIPT_HIT_MISS) return false;
	if (packed & MIPT_HIT_ANALYTIC) { obj = (int)(packed & 0x7fffffffu); return true; }
	if (sc->n_meshes == 1) { obj = sc->first_mesh; tri = (int)packed; return true; }       // (the scene's first mesh starts at triangle 0)
	int lo = 0, hi = sc->n_meshes - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sc->mesh_first[mid].x <= packed) lo = mid; else hi = mid - 1; }
	const uint2 e = sc->mesh_first[lo];
	obj = (int)e.y; tri = (int)(packed - e.x);
	return true;
}
MIPT_DEV unsigned hit_pack(const DScene* __restrict__ sc, int obj, int tri) {      // tri: mesh-local, or < 0 for a sphere / plane
	return tri < 0 ? (MIPT_HIT_ANALYTIC | (unsigned)obj) : (sc->obj[obj].tri_base + (unsigned)tri);
}
MIPT_DEV bool object_has_merl(const DScene* __restrict__ sc, int obj) {             // the first 32 objects in a register mask (no load), the rest in their records
	return obj < 32 ? ((sc->merl_mask >> obj) & 1u) != 0 : sc->obj[obj].brdf_kind == 1;
}

MIPT_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// ---------------------------------------------------------------- object transforms (Geometry.h:362-396)
MIPT_DEV f3 xf_point(const float* m, f3 v) {
	return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3], m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7], m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]);
}
MIPT_DEV f3 xf_dir(const float* m, f3 v) {
	return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
MIPT_DEV f3 xf_rot(const float* r, f3 v) {
	return mk3(r[0] * v.x + r[1] * v.y + r[2] * v.z, r[3] * v.x + r[4] * v.y + r[5] * v.z, r[6] * v.x + r[7] * v.y + r[8] * v.z);
}

// ---------------------------------------------------------------- AABB slab tests
// One routine for the reference's three variants.  XSPLIT = true gives
// intersection_invd_positive_x / _negative_x (Geometry.h:144-204: the x slab is rejected on the
// sign of the un-scaled difference), used by the closest-hit traversal; XSPLIT = false gives the
// generic intersection_invd (Geometry.h:114-142), used for the root box and by the shadow
// traversal.  sx/sy/sz = signs[k] = (1/d[k] >= 0).  The statement order of the reference is kept
// (NaN / inf behaviour of each comparison is part of the result).
template <bool XSPLIT>
MIPT_DEV bool box_test(f3 bmin, f3 bmax, f3 o, f3 invd, bool sx, bool sy, bool sz, float& t_out) {
	bool rej;
	float t_max, t;
	float farx = sx ? bmax.x : bmin.x, nearx = sx ? bmin.x : bmax.x;
	if (XSPLIT) {
		t_max = farx - o.x;
		rej = sx ? (t_max < 0) : (t_max > 0);
		t_max *= invd.x;
	} else {
		t_max = (farx - o.x) * invd.x;
		rej = (t_max < 0);
	}
	t = (nearx - o.x) * invd.x;
	float t_max_y = ((sy ? bmax.y : bmin.y) - o.y) * invd.y;
	rej |= (t_max_y < 0);
	float t_min_y = ((sy ? bmin.y : bmax.y) - o.y) * invd.y;
	rej |= (t_min_y > t_max) | (t_max_y < t);
	if (t_min_y > t) t = t_min_y;
	if (t_max_y < t_max) t_max = t_max_y;
	float t_max_z = ((sz ? bmax.z : bmin.z) - o.z) * invd.z;
	rej |= (t_max_z < 0);
	float t_min_z = ((sz ? bmin.z : bmax.z) - o.z) * invd.z;
	rej |= (t > t_max_z) | (t_min_z > t_max);
	if (t_min_z > t) t = t_min_z;
	if (t < 0) t = 0;
	t_out = t;
	return !rej;
}

// The same slab test for the persistent traversal kernels, on a DFatNode's (min, max) pairs: both planes of an axis
// go through one packed subtract and one packed multiply — the SAME fp32 operations on the same operands as
// intersection_invd / _positive_x / _negative_x — and the reference's chain of early-outs is evaluated in closed
// form.  With near_k <= far_k on every axis (monotone rounding of (plane - o) * invd) the chain
//     far_x < 0 | far_y < 0 | near_y > far_x | far_y < near_x | far_z < 0 | max(near_x, near_y) > far_z | near_z > min(far_x, far_y)
// is exactly  min(far) < 0 | max(near) > min(far)  and the returned t is max(near, 0); the x-split variants test the
// sign of the un-multiplied (far_x - o.x) instead of far_x < 0 (Geometry.h:146-204), which is kept literally.
// NOT valid when a product is NaN (0 * inf: a direction component is exactly 0 and the origin lies on a slab
// plane): callers route rays with an infinite invd component through box_test above.
typedef float mipt_f2 __attribute__((ext_vector_type(2)));
// The ray comes as register pairs (o.x, o.y), (invd.x, invd.y), (o.z, invd.z): the packed instructions broadcast
// one half of a pair through op_sel, so no duplicated operands are needed.
template <bool XSPLIT>
MIPT_DEV bool box_test_pairs(mipt_f2 X, mipt_f2 Y, mipt_f2 Z, mipt_f2 o_xy, mipt_f2 i_xy, mipt_f2 oz_iz, bool sx, bool sy, bool sz, float& t_out) {
	const mipt_f2 rx = X - __builtin_shufflevector(o_xy, o_xy, 0, 0), ry = Y - __builtin_shufflevector(o_xy, o_xy, 1, 1), rz = Z - __builtin_shufflevector(oz_iz, oz_iz, 0, 0);
	const mipt_f2 tx = rx * __builtin_shufflevector(i_xy, i_xy, 0, 0), ty = ry * __builtin_shufflevector(i_xy, i_xy, 1, 1), tz = rz * __builtin_shufflevector(oz_iz, oz_iz, 1, 1);
	const float nx = sx ? tx.x : tx.y, fx = sx ? tx.y : tx.x;
	const float ny = sy ? ty.x : ty.y, fy = sy ? ty.y : ty.x;
	const float nz = sz ? tz.x : tz.y, fz = sz ? tz.y : tz.x;
	const float t_enter = fmaxf(fmaxf(nx, ny), nz);
	const float t_exit = fminf(fminf(fx, fy), fz);
	bool ok = !(t_enter > t_exit);
	if (XSPLIT) {
		// the x slab is rejected on the sign of the UN-multiplied difference of its far plane (Geometry.h:146-204): rx.y < 0 for a ray that
		// goes up in x, rx.x > 0 for one that goes down, i.e. -rx.x < 0 — one select and one compare (written as a choice between the two
		// comparisons it compiled to seven vector instructions per box: both compares, their results as 0 / 1 integers, a select, a test)
		const bool rejx = (sx ? rx.y : -rx.x) < 0;
		ok = ok && !rejx && !(fminf(fy, fz) < 0);
	} else {
		ok = ok & !(t_exit < 0);
	}
	t_out = t_enter < 0 ? 0.f : t_enter;
	return ok;
}

// ---------------------------------------------------------------- Triangle::intersection (TriangleMesh.h:82-104)
template <bool DERIVE = false>
MIPT_DEV bool tri_test(const DTriIsect* __restrict__ T, f3 o, f3 d, float& t, float& beta, float& gamma) {
	const float4* q = reinterpret_cast<const float4*>(T);
	const float4 q0 = q[0], q1 = q[1], q2 = q[2];
	const f3 A = mk3(q0.x, q0.y, q0.z), u = mk3(q0.w, q1.x, q1.y), v = mk3(q1.z, q1.w, q2.x);
	const float invdetm = q2.y, m11 = q2.z, m12 = q2.w;
	f3 N; float m22;
	if (DERIVE) { N = cross(u, v); m22 = norm2(v); }
	else { const float4 q3 = q[3]; m22 = q3.x; N = mk3(q3.y, q3.z, q3.w); }
	t = dot(A - o, N) / dot(d, N);
	if (t < 0 || t != t) return false;
	f3 P = o + t * d;
	f3 w = P - A;
	float b11 = dot(w, u);
	float b21 = dot(w, v);
	float detb = b11 * m22 - b21 * m12;
	beta = detb * invdetm;
	if (beta < 0) return false;
	float detg = b21 * m11 - b11 * m12;
	gamma = detg * invdetm;
	if (gamma < 0) return false;
	float alpha = 1 - beta - gamma;
	if (alpha < 0) return false;
	return true;
}

// ---------------------------------------------------------------- textures / queryMaterial
MIPT_DEV int tex_index(const DTex& t, float u, float v) {     // BRDF.h:296-298
	int x = (int)(u * (float)(uint64_t)(t.W - 1));
	int y = (int)(v * (float)(uint64_t)(t.H - 1));
	return (y * t.W + x) * 3;
}
// (the texture descriptors and their texels are reached through pointers loaded from memory, which the compiler has to
// treat as generic: read through global-address-space views they are global_load instead of flat_load instructions)
typedef const __attribute__((address_space(1))) DTex glb_DTex;
typedef const __attribute__((address_space(1))) float glb_cfloat;
MIPT_DEV f3 tex_getVec(const DTex& tg, float u, float v) {     // BRDF.h:293-308
	glb_DTex& t = (glb_DTex&)tg;
	const float m0 = t.mult[0], m1 = t.mult[1], m2 = t.mult[2];
	if (t.W > 0) { DTex d; d.W = t.W; d.H = t.H; int idx = tex_index(d, u, v); glb_cfloat* val = (glb_cfloat*)t.values; return mk3(val[idx] * m0, val[idx + 1] * m1, val[idx + 2] * m2); }
	return mk3(m0, m1, m2);
}
MIPT_DEV float tex_getValRed(const DTex& tg, float u, float v) {   // BRDF.h:379-391
	glb_DTex& t = (glb_DTex&)tg;
	if (t.W > 0) { DTex d; d.W = t.W; d.H = t.H; int idx = tex_index(d, u, v); return ((glb_cfloat*)t.values)[idx] * t.mult[0]; }
	return t.mult[0];
}
MIPT_DEV f3 tex_getNormal(const DTex& tg, float u, float v) {  // BRDF.h:347-357
	glb_DTex& t = (glb_DTex&)tg;
	if (t.W > 0) { DTex d; d.W = t.W; d.H = t.H; int idx = tex_index(d, u, v); glb_cfloat* val = (glb_cfloat*)t.values; return mk3(val[idx], val[idx + 1], val[idx + 2]); }
	return mk3(0.f, 0.f, 1.f);
}
// Object::queryMaterial (Geometry.h:399-445).  idx is compared as size_t in the reference, so a
// negative group selects the defaults.
#ifndef MIPT_GROUP_TABLE
#define MIPT_GROUP_TABLE 1
#endif
MIPT_DEV void query_material(const DObject& o, const DGroupMat* gmat, int ngroups, int idx, float u, float v, Mat& mat) {
	u = tex_wrap(u);
	v = tex_wrap(v);
#if MIPT_GROUP_TABLE
	// the group's record (four 16-byte loads); only the slots whose entry is an image other than Kd's go through the entry's descriptor
	typedef float v4f_ __attribute__((ext_vector_type(4)));
	const unsigned ng = (unsigned)ngroups;
	const unsigned gi = (unsigned)idx < ng ? (unsigned)idx : ng;
	const __attribute__((address_space(1))) v4f_* q = (const __attribute__((address_space(1))) v4f_*)(gmat + gi);
	const v4f_ r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3];
	mat.Kd = mk3(r0.x, r0.y, r0.z); mat.Ks = mk3(r0.w, r1.x, r1.y); mat.Ne = mk3(r1.z, r1.w, r2.x);
	mat.transp = r2.y < 0.5f;                                                                                        // getBool, BRDF.h:335-346
	mat.refr_index = r2.z;
	const unsigned images = __float_as_uint(r2.w);
	if (images != 0) {
		if (images & (1u << MT_KD)) {                                                                                // Texture::getVec (BRDF.h:293-308): texel * multiplier
			DTex d; d.W = __float_as_int(r3.z); d.H = __float_as_int(r3.w);
			glb_cfloat* val = (glb_cfloat*)(((uint64_t)__float_as_uint(r3.y) << 32) | __float_as_uint(r3.x));
			const int ti = tex_index(d, u, v);
			mat.Kd = mk3(val[ti] * r0.x, val[ti + 1] * r0.y, val[ti + 2] * r0.z);
		}
		if (images & (1u << MT_KS)) mat.Ks = tex_getVec(o.tex[MT_KS][idx], u, v);
		if (images & (1u << MT_NE)) mat.Ne = tex_getVec(o.tex[MT_NE][idx], u, v);
		if (images & (1u << MT_TRANSP)) mat.transp = tex_getValRed(o.tex[MT_TRANSP][idx], u, v) < 0.5f;
		if (images & (1u << MT_REFR)) mat.refr_index = tex_getValRed(o.tex[MT_REFR][idx], u, v);
	}
#else
	unsigned ui = (unsigned)idx;
	mat.Kd = (ui >= (unsigned)o.ntex[MT_KD]) ? mk3(1, 1, 1) : tex_getVec(o.tex[MT_KD][idx], u, v);
	mat.Ks = (ui >= (unsigned)o.ntex[MT_KS]) ? mk3(0, 0, 0) : tex_getVec(o.tex[MT_KS][idx], u, v);
	mat.Ne = (ui >= (unsigned)o.ntex[MT_NE]) ? mk3(1, 1, 1) : tex_getVec(o.tex[MT_NE][idx], u, v);
	mat.transp = (ui >= (unsigned)o.ntex[MT_TRANSP]) ? false : (tex_getValRed(o.tex[MT_TRANSP][idx], u, v) < 0.5f);   // getBool, BRDF.h:335-346
	mat.refr_index = (ui >= (unsigned)o.ntex[MT_REFR]) ? 1.3f : tex_getValRed(o.tex[MT_REFR][idx], u, v);
#endif
	mat.Ke = mk3(0, 0, 0);
	// Ksub is only read by the subsurface branch (out of scope); upload rejects non-zero Ksub.
}
MIPT_DEV void query_material(const DObject& o, int idx, float u, float v, Mat& mat) { query_material(o, o.gmat, o.ngroups, idx, u, v, mat); }

// alpha-map test of the leaf loop (TriangleMesh.cpp:1198-1205 / 1300-1307)
// (not inlined: a rare path, and keeping its control flow out of the traversal loops avoids the hipcc
// 7.2 mis-scheduling described at the call sites)
__device__ __attribute__((noinline)) bool alpha_rejects(const DObject& o, int i, float alpha, float beta, float gamma) {
	int group = o.shade[i].group;
	if (group >= 0) group &= MIPT_GROUP_MASK;
	const int* ix = o.uvidx + 3 * (size_t)i;
	if ((unsigned)o.ntex[MT_ALPHA] > (unsigned)group && ix[0] >= 0 && ix[1] >= 0 && ix[2] >= 0) {
		const float* a = o.uvs + 3 * (size_t)ix[0];
		const float* b = o.uvs + 3 * (size_t)ix[1];
		const float* c = o.uvs + 3 * (size_t)ix[2];
		float u = a[0] * alpha + b[0] * beta + c[0] * gamma;
		float v = a[1] * alpha + b[1] * beta + c[1] * gamma;
		u = tex_wrap(u);
		v = tex_wrap(v);
		if ((double)tex_getValRed(o.tex[MT_ALPHA][group], u, v) < 0.5) return true;
	}
	return false;
}

// ---------------------------------------------------------------- BVH traversal
// Per-lane traversal stack.  The reference pushes (node, tnear) pairs (TriangleMesh.cpp:1153-1190);
// we continue directly into the near child instead of pushing and re-popping it (equivalent: t
// does not change between its push and its pop), so only far children are stored.
#ifdef MIPT_PROFILE_SIMD
// diagnostic build only (tools/simd_prof.py).  (events, active lanes) pairs: 0 inner step, 2 leaf phase, 4 leaf triangle
// iteration, 6 object-loop pass, 8 outer iteration, 10 refill; wave cycles: 12 refill + object loop, 13 inner phase, 14 leaf phase
// shade stage (fast tier): 16 sub-chunk (lanes that hold a vertex), 18 lanes whose vertex is the diffuse one (light sample +
// continuation), 20 environment hits, 22 misses / light hits, 24 vertices sent to the slow tier
__device__ unsigned long long g_simd_prof[40];
MIPT_DEV unsigned lane_id_() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
#define MIPT_PROF_COUNT(slot) { unsigned long long m_ = __ballot(1); if (lane_id_() == (unsigned)(__ffsll((long long)m_) - 1)) { atomicAdd(&g_simd_prof[slot], 1ull); atomicAdd(&g_simd_prof[(slot) + 1], (unsigned long long)__popcll(m_)); } }
#define MIPT_PROF_CLOCK(var) long long var = clock64()
#define MIPT_PROF_CYCLES(slot, from, to) { if (lane_id_() == 0) atomicAdd(&g_simd_prof[slot], (unsigned long long)((to) - (from))); }
#else
#define MIPT_PROF_COUNT(slot) {}
#define MIPT_PROF_CLOCK(var) {}
#define MIPT_PROF_CYCLES(slot, from, to) {}
#endif
#define MIPT_STACK_DEPTH 48
struct ScratchStack {            // private (scratch) memory: any occupancy, slower pops
	uint32_t ref[MIPT_STACK_DEPTH];
	float tnear[MIPT_STACK_DEPTH];
	MIPT_DEV void push(int sp, uint32_t r, float t) { ref[sp] = r; tnear[sp] = t; }
	MIPT_DEV void pop(int sp, uint32_t& r, float& t) const { r = ref[sp]; t = tnear[sp]; }
};
// LDS-resident stack of the persistent traversal kernels: entry sp of a lane lives at
// lds[sp * blockDim.x + threadIdx.x] (one 8-byte word; conflict-free for ds_read/write_b64 when the
// lanes of a group use the same sp).  Entries beyond MIPT_LDS_STACK spill to a per-thread column of a
// global scratch buffer that is normally never touched.  Both pointers carry their address space in
// the type so that pushes and pops compile to ds_* / global_* instructions, not flat ones.
#ifndef MIPT_LDS_STACK
#define MIPT_LDS_STACK 10         // (round 4: 0.2 % of the inner steps of configs[2] push at depth >= 10 and reach the global spill column; 8 entries: 3 %, +0.5 % time)
#endif
#define MIPT_SPILL_STACK (MIPT_STACK_DEPTH - MIPT_LDS_STACK)
typedef __attribute__((address_space(3))) uint2 lds_uint2;
typedef __attribute__((address_space(1))) uint2 glb_uint2;
struct LdsStack {
	lds_uint2* base;              // &lds[threadIdx.x]
	glb_uint2* spill;             // &spill[global thread id]
	int stride;                   // blockDim.x
	int spill_stride;             // total threads of the grid
	MIPT_DEV void push(int sp, uint32_t r, float t) {
		uint2 e = make_uint2(r, __float_as_uint(t));
		if (sp < MIPT_LDS_STACK) { lds_uint2* p = base + sp * stride; p->x = e.x; p->y = e.y; }
		else { glb_uint2* p = spill + (size_t)(sp - MIPT_LDS_STACK) * spill_stride; p->x = e.x; p->y = e.y; }
	}
	MIPT_DEV void pop(int sp, uint32_t& r, float& t) const {
		uint32_t x, y;
		if (sp < MIPT_LDS_STACK) { const lds_uint2* p = base + sp * stride; x = p->x; y = p->y; }
		else { const glb_uint2* p = spill + (size_t)(sp - MIPT_LDS_STACK) * spill_stride; x = p->x; y = p->y; }
		r = x; t = __uint_as_float(y);
	}
};

struct TravCounters { uint32_t box, node, tri; };

// TriMesh::intersection (TriangleMesh.cpp:1133-1214; SHADOW=false) and
// TriMesh::intersection_shadow (:1239-1319; SHADOW=true) on the ray (o,d) given in the object's
// local frame.  Closest: returns whether a triangle with t < cur_best_t exists, and t / triangle /
// barycentrics of the first one reaching the minimum.  Shadow: returns the reference's
// has_inter, with t of the last accepted triangle (the caller compares it with 0.999*dist).
template <bool SHADOW, class STK>
MIPT_DEV bool mesh_traverse(const DObject& o, f3 org, f3 d, float cur_best_t, float dist_light,
                            float& t_out, int& tri_out, float& beta_out, float& gamma_out, STK& stk) {
	float t = cur_best_t;
	bool has_inter = false;
	f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);      // 1./d narrowed to float == 1.f/d
	bool sx = invd.x >= 0, sy = invd.y >= 0, sz = invd.z >= 0;
	float t_root;
	if (!box_test<false>(ld3(o.root_min), ld3(o.root_max), org, invd, sx, sy, sz, t_root)) return false;
	if (t_root > cur_best_t) return false;
	if (SHADOW && t_root > dist_light) return false;

	int sp = 0;
	const uint32_t NONE = 0x7fffffffu;                   // not a valid inner index, no leaf bit
	uint32_t cur = o.root_ref;
	const float4* __restrict__ nodes = reinterpret_cast<const float4*>(o.nodes);
	// next stack entry that is still worth visiting (TriangleMesh.cpp:1160-1163), or NONE
	auto pop_next = [&]() -> uint32_t {
		while (sp > 0) {
			--sp;
			uint32_t r; float tn;
			stk.pop(sp, r, tn);
			if (!(tn > t)) return r;
		}
		return NONE;
	};
	// "while-while" schedule: all lanes of the wave first descend through inner nodes until each
	// holds a leaf (or is done), then all process their leaves: the two kinds of work, which need
	// different code, are not interleaved lane by lane.  Per lane the visiting order is unchanged.
	for (;;) {
		while (cur != NONE && !(cur & MIPT_LEAF_BIT)) {
			MIPT_PROF_COUNT(0)
			const float4* q = nodes + 4 * (size_t)cur;
			float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
			f3 lmin = mk3(q0.x, q0.z, q1.x), lmax = mk3(q0.y, q0.w, q1.y);      // DFatNode: (min, max) pairs per axis
			f3 rmin = mk3(q1.z, q2.x, q2.z), rmax = mk3(q1.w, q2.y, q2.w);
			uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
			float tl, tr;
			bool goleft, goright;
			if (SHADOW) {
				goleft = box_test<false>(lmin, lmax, org, invd, sx, sy, sz, tl) && (tl < t) && (tl < dist_light);
				goright = box_test<false>(rmin, rmax, org, invd, sx, sy, sz, tr) && (tr < t) && (tr < dist_light);
			} else {
				goleft = box_test<true>(lmin, lmax, org, invd, sx, sy, sz, tl) && (tl < t);
				goright = box_test<true>(rmin, rmax, org, invd, sx, sy, sz, tr) && (tr < t);
			}
			if (goleft && goright) {
				if (tl < tr) { stk.push(sp, rref, tr); sp++; cur = lref; }
				else { stk.push(sp, lref, tl); sp++; cur = rref; }
			} else if (goleft) cur = lref;
			else if (goright) cur = rref;
			else cur = pop_next();
		}
		if (cur == NONE) break;
		MIPT_PROF_COUNT(2)
		int first = (int)(cur & MIPT_LEAF_FIRST_MASK);
		const int count = mipt_leaf_count(cur, o.fat_leaves, o.n_fat_leaves);
		for (int i = first; i < first + count; i++) {
			float lt, lb, lg;
			if (tri_test(o.tris + i, org, d, lt, lb, lg)) {
				// Accept test written as one flag: with the alpha test nested inside `if (lt < t)` hipcc 7.2
				// (gfx950, -O3) moved the `t` / `beta` updates into the alpha_test == 0 arm only
				// (seen in the ISA of k_wf_extend; the cut-out golden scene caught it).
				bool accept = lt < t;
				if (accept && o.alpha_test) accept = !alpha_rejects(o, i - (int)o.tri_base, 1 - lb - lg, lb, lg);
				if (accept) {
					has_inter = true;
					t = lt; tri_out = i - (int)o.tri_base; beta_out = lb; gamma_out = lg;
					if (SHADOW && ((double)t < (double)dist_light * 0.999)) { t_out = t; return true; }   // :1309
				}
			}
		}
		cur = pop_next();
	}
	t_out = t;
	return has_inter;
}

// ---------------------------------------------------------------- Sphere / Plane (Geometry.h:918-992, 1071-1094, 1142-1157, 1185-1191)
MIPT_DEV bool sphere_test(const DObject& s, f3 o, f3 d, float& t) {
	f3 oc = o - ld3(s.O);
	float b = dot(d, oc);
	float a = norm2(d);
	float c = norm2(oc) - s.R2;
	float delta = b * b - a * c;
	if (delta < 0) return false;
	float sqDelta = sqrtf(delta);
	float inva = 1.f / a;
	float t2 = (-b + sqDelta) * inva;
	if (t2 < 0) return false;
	float t1 = (-b - sqDelta) * inva;
	t = (t1 > 0) ? t1 : t2;
	return true;
}
MIPT_DEV bool plane_test(const DObject& p, f3 o, f3 d, float& t) {
	f3 N = ld3(p.vecN);
	float ddot = dot(d, N);
	if ((double)fabsf(ddot) < 1E-9) return false;
	t = dot(ld3(p.A) - o, N) / ddot;
	if (t <= 0.f) return false;
	return true;
}

// ---------------------------------------------------------------- TriMesh::getMaterial (TriangleMesh.cpp:919-970)
MIPT_DEV void mesh_material(const DObject& o, const ObjHot& hot, int tri, float alpha, float beta, float gamma, Mat& mat) {
	typedef float v4f_ __attribute__((ext_vector_type(4)));
	const __attribute__((address_space(1))) v4f_* q = (const __attribute__((address_space(1))) v4f_*)(hot.shade + tri);
	const v4f_ r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3];
	const float4 q0 = make_float4(r0.x, r0.y, r0.z, r0.w), q1 = make_float4(r1.x, r1.y, r1.z, r1.w), q2 = make_float4(r2.x, r2.y, r2.z, r2.w), q3 = make_float4(r3.x, r3.y, r3.z, r3.w);
	f3 n0 = mk3(q0.x, q0.y, q0.z), n1 = mk3(q0.w, q1.x, q1.y), n2 = mk3(q1.z, q1.w, q2.x);
	float uv00 = q2.y, uv01 = q2.z, uv10 = q2.w, uv11 = q3.x, uv20 = q3.y, uv21 = q3.z;
	// group word: bit 30 = "the first UV index of the triangle is valid" (TriangleMesh.cpp:934), set at upload so that
	// the test needs no extra (dependent, random) fetch from the index array
	const int graw = __float_as_int(q3.w);
	const int group = graw >= 0 ? (graw & MIPT_GROUP_MASK) : graw;
	float u = 0, v = 0;
	bool has_uv = false;
	if (hot.nuvs != 0 && graw >= 0 && (graw & MIPT_GROUP_UV_OK)) {
		u = (uv00 * alpha + uv10 * beta + uv20 * gamma);
		v = (uv01 * alpha + uv11 * beta + uv21 * gamma);
		has_uv = true;
	}
	query_material(o, hot.gmat, hot.ngroups, group, u, v, mat);
	f3 N;
	if (!hot.interp_normals) {
		const DTriIsect& T = o.tris[o.tri_base + tri];
		N = ld3(T.N);
	} else {
		N = n0 * alpha + n1 * beta + n2 * gamma;
	}
	N = normalize(N);
	if (hot.ntex_normal != 0 && has_uv && (unsigned)group < (unsigned)hot.ntex_normal && hot.tangent_soup != nullptr) {
		const float* ts = hot.tangent_soup + 9 * (size_t)tri;
		f3 tangent = ld3(ts) * alpha + ld3(ts + 3) * beta + ld3(ts + 6) * gamma;
		tangent = normalize(tangent);
		f3 bitangent = cross(N, tangent);
		f3 NsLocal = tex_getNormal(o.tex[MT_NORMAL][group], u, v);
		f3 Ns = NsLocal.x * tangent + NsLocal.y * bitangent + NsLocal.z * N;
		if (Ns.x == 0.f && Ns.y == 0.f && Ns.z == 0.f) Ns = N;
		N = normalize(Ns);
	}
	if (hot.flip_normals) N = -N;
	mat.shadingN = N;
}
MIPT_DEV void mesh_material(const DObject& o, int tri, float alpha, float beta, float gamma, Mat& mat) { const ObjHot hot = load_obj_hot(o); mesh_material(o, hot, tri, alpha, beta, gamma, mat); }

// Sphere material (Geometry.h:948-991)
// inherit: `mat` is the ONE MaterialValues of Scene::intersection's loop (scene_intersect_inherit below) — a sphere without lists leaves
// in it what the object tested before wrote; otherwise such a sphere (the light; upload lets no other one through) gets the defaults
// probe: the hit comes from Sphere::reservoir_sampling_intersection (the subsurface probe, Geometry.h:1053-1068), whose spherical
// coordinates are computed with double intermediates (`1 - acos(N[1]) / M_PI`, `(atan2(..) + M_PI) / (2.*M_PI)`) where
// Sphere::intersection's are all float (:976-977)
MIPT_DEV void sphere_material(const DObject& s, f3 Plocal, Mat& mat, bool inherit = false, bool probe = false) {
	f3 N = Plocal - ld3(s.O);
	// MaterialValues() defaults for the fields a texture-less sphere never writes (BRDF.h:9-16)
	if (!inherit) { mat.Kd = mk3(0.5f, 0.5f, 0.5f); mat.Ks = mk3(0, 0, 0); mat.Ne = mk3(100, 100, 100); mat.transp = false; mat.refr_index = 0.f; }
	if (s.has_envmap) {
		N = fast_normalize(N);
		float theta = 1.f - mipt_acosf(N.y) / (float)MIPT_PI;
		float phi = (float)(((double)mipt_atan2f(-N.z, N.x) + MIPT_PI) / (double)(2.f * (float)MIPT_PI));
		query_material(s, 0, theta, phi, mat);
		mat.shadingN = -N;
		int idx = 3 * ((int)(theta * ((float)s.envH - 1.f)) * s.envW + (int)(phi * ((float)s.envW - 1.f)));
		if (idx < 0 || idx >= 3 * s.envW * s.envH) mat.Ke = mk3(0, 0, 0);
		else mat.Ke = mk3((float)s.envtex[idx], (float)s.envtex[idx + 1], (float)s.envtex[idx + 2]) * (100000.f / 255.f);
		return;
	}
	// a sphere with material lists (Geometry.h:975-981): the lists are looked up at the spherical coordinates of the NORMALISED
	// normal, which then is the one handed on (a sphere without lists keeps P - O as it is: Scene::intersection normalises later)
	if (s.ntex[MT_KD] != 0 || s.ntex[MT_KS] != 0 || s.ntex[MT_NE] != 0 || s.ntex[MT_TRANSP] != 0 || s.ntex[MT_REFR] != 0) {
		N = fast_normalize(N);
		float theta = 1.f - mipt_acosf(N.y) / (float)MIPT_PI;
		float phi = (mipt_atan2f(-N.z, N.x) + (float)MIPT_PI) / (2.f * (float)MIPT_PI);
		if (probe) {
			theta = (float)(1. - (double)mipt_acosf(N.y) / MIPT_PI);
			phi = (float)(((double)mipt_atan2f(-N.z, N.x) + MIPT_PI) / (2. * MIPT_PI));
		}
		query_material(s, 0, theta, phi, mat);
	}
	mat.shadingN = s.flip_normals ? -N : N;
	mat.Ke = mk3(0, 0, 0);
}

// ---------------------------------------------------------------- Scene::intersection (Geometry.cpp:589-688)
// Returns the closest hit over all objects (strict <, objects in index order) and, for a hit,
// the world-space point and the MaterialValues.  The reference fills a scratch `localmat` while
// it loops and copies it on every improvement; here the winner's material is evaluated once
// after the loop from (object, t, triangle, barycentrics), which yields the same values for every
// field its intersection() routine writes.
template <class STK>
MIPT_DEV bool scene_closest(const DScene* __restrict__ sc, Ray r, Hit& h, STK& stk) {
	h.obj = -1; h.tri = -1;
	float min_t = __int_as_float(0x7f800000);          // min_t = 1E99 narrowed: +inf
	h.beta = 0; h.gamma = 0;
	const int nobj = sc->nobj;
	for (int i = 0; i < nobj; i++) {
		const DObject& o = sc->obj[i];
		f3 d = xf_dir(o.inv, r.d);
		f3 org = xf_point(o.inv, r.o);
		float t; int tri = -1; float b = 0, g = 0;
		bool hit;
		if (o.type == 1) hit = sphere_test(o, org, d, t);
		else if (o.type == 2) hit = plane_test(o, org, d, t);
		else hit = mesh_traverse<false>(o, org, d, min_t, 0.f, t, tri, b, g, stk);
		if (hit && t < min_t) { min_t = t; h.obj = i; h.tri = tri; h.beta = b; h.gamma = g; }
	}
	h.t = min_t;
	return h.obj >= 0;
}

// World-space hit point and MaterialValues of the winning object (tail of Scene::intersection,
// Geometry.cpp:668-684, plus the winner's own material code).
MIPT_DEV void hit_material_obj(const DObject& o, Ray r, const Hit& h, f3& P, Mat& mat, bool inherit = false, bool probe = false) {
	const ObjHot hot = load_obj_hot(o);                  // (issued together with the matrix loads below)
	f3 d = xf_dir(o.inv, r.d);
	f3 org = xf_point(o.inv, r.o);
	f3 Pl = org + h.t * d;                               // P = d.origin + t*d.direction in the object's frame
	mat.miroir = hot.miroir; mat.merl = hot.merl;
	if (hot.type == 1) sphere_material(o, Pl, mat, inherit, probe);
	else if (hot.type == 2) { mat.shadingN = ld3(o.vecN); query_material(o, hot.gmat, hot.ngroups, 0, Pl.x * 0.1f, Pl.z * 0.1f, mat); }
	else {
		float beta = h.beta, gamma = h.gamma, alpha = 1 - beta - gamma;
		// NaN / Inf clean-up of the winner's barycentrics (TriangleMesh.cpp:1219-1226)
		if (isnan(alpha) && isnan(beta) && isnan(gamma)) { alpha = 1; beta = 0; gamma = 0; }
		if (isnan(alpha)) alpha = 0;
		if (isnan(beta)) beta = 0;
		if (isnan(gamma)) gamma = 0;
		if (isinf(alpha)) alpha = 1;
		if (isinf(beta)) beta = 1;
		if (isinf(gamma)) gamma = 1;
		mesh_material(o, hot, h.tri, alpha, beta, gamma, mat);
	}
	P = xf_point(o.trans, Pl);
	mat.shadingN = fast_normalize(xf_rot(o.rot, mat.shadingN));
}
MIPT_DEV void hit_material(const DScene* __restrict__ sc, Ray r, const Hit& h, f3& P, Mat& mat) { hit_material_obj(sc->obj[h.obj], r, h, P, mat); }
// (A wave-uniform loop over the objects, with scalar loads of the description, was measured slower in the shade stage:
// the material chains of the distinct objects in a wave then run one after the other.)

// MaterialValues::Ksub of a hit (Object::queryMaterial, Geometry.h:418-424): subsurface[group] at the hit's texture
// coordinates for a mesh (TriMesh::getMaterial), subsurface[0] at (x, z)/10 for a plane (Geometry.h:1150-1155); a sphere with
// lists: subsurface[0] at the spherical coordinates of the normal.  `Pl` = the hit point in the object's frame.
MIPT_DEV f3 hit_ksub(const DObject& o, const Hit& h, f3 Pl, bool probe = false) {
	if (o.type == 1) {
		// a sphere: queryMaterial(0, theta, phi) runs only when one of the Kd / Ks / Ne / transparency / index lists exists (Geometry.h:975, 1056)
		// — a sphere without them leaves Ksub as it was (upload keeps such scenes away) —, at the coordinates sphere_material computes
		if (o.ntex[MT_KSUB] <= 0 || !(o.has_envmap || o.ntex[MT_KD] != 0 || o.ntex[MT_KS] != 0 || o.ntex[MT_NE] != 0 || o.ntex[MT_TRANSP] != 0 || o.ntex[MT_REFR] != 0)) return mk3(0, 0, 0);
		const f3 N = fast_normalize(Pl - ld3(o.O));
		float theta = 1.f - mipt_acosf(N.y) / (float)MIPT_PI;
		float phi = o.has_envmap ? (float)(((double)mipt_atan2f(-N.z, N.x) + MIPT_PI) / (double)(2.f * (float)MIPT_PI)) : (mipt_atan2f(-N.z, N.x) + (float)MIPT_PI) / (2.f * (float)MIPT_PI);
		if (probe) { theta = (float)(1. - (double)mipt_acosf(N.y) / MIPT_PI); phi = (float)(((double)mipt_atan2f(-N.z, N.x) + MIPT_PI) / (2. * MIPT_PI)); }
		return tex_getVec(o.tex[MT_KSUB][0], tex_wrap(theta), tex_wrap(phi));
	}
	if (o.type == 2) {
		if (o.ntex[MT_KSUB] <= 0) return mk3(0, 0, 0);
		return tex_getVec(o.tex[MT_KSUB][0], tex_wrap(Pl.x * 0.1f), tex_wrap(Pl.z * 0.1f));
	}
	if (o.type != 0 || h.tri < 0) return mk3(0, 0, 0);
	const float4* q = reinterpret_cast<const float4*>(o.shade + h.tri);
	const float4 q2 = q[2], q3 = q[3];
	const int graw = __float_as_int(q3.w);
	const int group = graw >= 0 ? (graw & MIPT_GROUP_MASK) : graw;
	if ((unsigned)group >= (unsigned)o.ntex[MT_KSUB]) return mk3(0, 0, 0);
	float beta = h.beta, gamma = h.gamma, alpha = 1 - beta - gamma;        // as hit_material_obj / mesh_material
	if (isnan(alpha) && isnan(beta) && isnan(gamma)) { alpha = 1; beta = 0; gamma = 0; }
	if (isnan(alpha)) alpha = 0;
	if (isnan(beta)) beta = 0;
	if (isnan(gamma)) gamma = 0;
	if (isinf(alpha)) alpha = 1;
	if (isinf(beta)) beta = 1;
	if (isinf(gamma)) gamma = 1;
	float u = 0, v = 0;
	if (o.nuvs != 0 && graw >= 0 && (graw & MIPT_GROUP_UV_OK)) {
		u = (q2.y * alpha + q2.w * beta + q3.y * gamma);
		v = (q2.z * alpha + q3.x * beta + q3.z * gamma);
	}
	return tex_getVec(o.tex[MT_KSUB][group], tex_wrap(u), tex_wrap(v));
}

// Scene::intersection as the reference runs it, for scenes with a sphere that has no material lists (DScene::inherit_material):
// ONE MaterialValues serves all objects of the loop (`localmat`, Geometry.cpp:596), every object that reports a hit — closer than the
// best so far or not — writes its material into it (a mesh reports only closer hits, TriangleMesh.cpp:1198), a sphere without lists
// writes the normal and Ke only (Geometry.h:983-986), and the winner takes a copy at the moment it wins (:611-621).  So such a
// sphere is shaded with the Kd / Ks / Ne / transparency / index of the last object before it in the list that the ray also hit,
// at THAT object's hit point — or with MaterialValues()'s defaults when there was none (the light and, without an environment map,
// object 1 write nothing either).
template <class STK>
__device__ __attribute__((noinline)) bool scene_intersect_inherit(const DScene* __restrict__ sc, Ray r, Hit& h, f3& P, Mat& mat, STK& stk) {
	h.obj = -1; h.tri = -1; h.beta = 0; h.gamma = 0;
	float min_t = __int_as_float(0x7f800000);
	Mat localmat;                          // MaterialValues() (BRDF.h:9-16); transp and refr_index are not initialised there: false / 0 as in the oracle
	localmat.shadingN = mk3(0, 1, 0); localmat.Kd = mk3(0.5f, 0.5f, 0.5f); localmat.Ks = mk3(0, 0, 0); localmat.Ne = mk3(100, 100, 100); localmat.Ke = mk3(0, 0, 0);
	localmat.transp = false; localmat.refr_index = 0.f; localmat.miroir = 0; localmat.merl = nullptr; localmat.Ksub = mk3(0, 0, 0);
	const int nobj = sc->nobj;
	for (int i = 0; i < nobj; i++) {
		const DObject& o = sc->obj[i];
		const f3 d = xf_dir(o.inv, r.d);
		const f3 org = xf_point(o.inv, r.o);
		float t; int tri = -1; float b = 0, g = 0;
		bool hit;
		if (o.type == 1) hit = sphere_test(o, org, d, t);
		else if (o.type == 2) hit = plane_test(o, org, d, t);
		else hit = mesh_traverse<false>(o, org, d, min_t, 0.f, t, tri, b, g, stk);
		if (!hit) continue;
		Hit hi; hi.obj = i; hi.tri = tri; hi.t = t; hi.beta = b; hi.gamma = g;
		f3 Pi;
		hit_material_obj(o, r, hi, Pi, localmat, true);
		// Ksub rides in the same MaterialValues: every object whose material code runs queryMaterial writes it (0 beyond its list), a
		// sphere without lists leaves what the object before it wrote (Geometry.h:975)
		if (!(o.type == 1 && !o.has_envmap && !(o.ntex[MT_KD] != 0 || o.ntex[MT_KS] != 0 || o.ntex[MT_NE] != 0 || o.ntex[MT_TRANSP] != 0 || o.ntex[MT_REFR] != 0))) localmat.Ksub = hit_ksub(o, hi, org + t * d);
		if (t < min_t) { min_t = t; h = hi; P = Pi; mat = localmat; }
	}
	h.t = min_t;
	return h.obj >= 0;
}

template <class STK>
MIPT_DEV bool scene_intersect(const DScene* __restrict__ sc, Ray r, Hit& h, f3& P, Mat& mat, STK& stk) {
	if (sc->inherit_material) return scene_intersect_inherit(sc, r, h, P, mat, stk);
	if (!scene_closest(sc, r, h, stk)) return false;
	hit_material(sc, r, h, P, mat);
	return true;
}

// ---------------------------------------------------------------- Scene::intersection_shadow (Geometry.cpp:691-744)
template <class STK, bool AVOID_GHOSTS = false>   // AVOID_GHOSTS: getColor's shadow rays skip ghost objects (Geometry.cpp:722, Raytracer.cpp:513)
MIPT_DEV bool scene_occluded(const DScene* __restrict__ sc, Ray r, float dist_light, STK& stk) {
	const int nobj = sc->nobj;
	const float inf = __int_as_float(0x7f800000);
	bool occluded = false;
	for (int i = 0; i < nobj; i++) {
		if (occluded) break;
		const DObject& o = sc->obj[i];
		if (AVOID_GHOSTS && o.ghost) continue;
		f3 d = xf_dir(o.inv, r.d);
		f3 org = xf_point(o.inv, r.o);
		float t; int tri; float b, g;
		bool hit;
		if (o.type == 1) hit = sphere_test(o, org, d, t);
		else if (o.type == 2) hit = plane_test(o, org, d, t);
		else hit = mesh_traverse<true>(o, org, d, inf, dist_light, t, tri, b, g, stk);
		if (hit && ((double)t < (double)dist_light * 0.999)) occluded = true;
	}
	return occluded;
}
