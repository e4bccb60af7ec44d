// mipt_unified.h — the persistent traversal as ONE fetch per lane and round (round 3).
//
// mipt_persistent.h separates the work of a wave into phases (all lanes descend inner nodes, then the leaves are tested,
// then finished lanes are refilled) because the three kinds of work need different code.  The counters of round 2 say what
// that costs on this machine: the kernel waits for memory 60 % of its wave cycles, a wave step is one dependent round trip
// of ~1.5-2 k cycles whatever it computes, and only 32.5 of 64 lanes take part in an inner step — the others hold a
// leaf, wait for the refill or have just finished.  VALU is 40 % busy.  So the scarce thing is ROUND TRIPS, not
// instructions, and a lane that does nothing during one is the waste.
//
// Here every lane that holds anything does one unit of work per round, and all units are the same memory operation:
// a 64-byte record fetched as q0..q3 from (pA, pB) —
//     a ray in a mesh       its current inner node (both children's boxes)  or  the next triangle of its current leaf
//     a lane just handed a queue entry          the entry (the path id)
//     a lane that knows its path id             the ray (origin, direction: two 16-byte records, pA / pB)
//     a shadow ray that found the light         the path's colour and the pending direct term (pA / pB)
// — followed by the consumers of each kind under their exec masks.  Leaves are walked one triangle per round by the lane
// that owns them (the sequential loop of TriangleMesh.cpp:1192-1212 as it is: any leaf size, alpha test per lane), the refill
// is a three-round pipeline that runs beside the traversal of the other lanes instead of stopping the wave, and the
// registers of a ray's slot hold whatever its stage needs (the colour of a finishing shadow ray sits where its origin was).
//
// Every ray still performs exactly the reference's sequence of operations: objects in index order, ordered stack traversal
// per mesh (near child first, ties -> right, far child pushed with its tnear, pop-skip on tnear > t), triangles of a leaf in
// index order with strict '<'.  Only WHICH round a step happens in changes.
#pragma once

#ifndef MIPT_U_WAVES
#define MIPT_U_WAVES 8                  // waves per SIMD the kernels are compiled for (64 VGPRs; LDS: the stack only, 20 KB per 256 threads)
#endif

typedef const __attribute__((address_space(4))) float cst_float;
typedef const __attribute__((address_space(4))) int cst_int;
typedef const __attribute__((address_space(4))) uint32_t cst_u32;

// The LDS stack of mipt_trace.h with its per-lane addresses recomputed at each use from wave-uniform bases and the lane
// number (two mbcnt), instead of held in three registers per lane for the life of the kernel: pushes and pops are rare next
// to the instructions that would otherwise spill.
struct LdsStackU {
	unsigned lds_wave;            // byte address of the wave's first entry-0 slot in LDS (uniform)
	glb_uint2* spill_wave;        // the wave's first spill column (uniform)
	int spill_stride;             // total threads of the grid
	MIPT_DEV void push(int sp, uint32_t r, float t) const {
		const unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		if (sp < MIPT_LDS_STACK) { lds_uint2* p = (lds_uint2*)(lds_wave + l * 8u + (unsigned)sp * (MIPT_TRAV_BLOCK * 8u)); p->x = r; p->y = __float_as_uint(t); }
		else { glb_uint2* p = spill_wave + l + (size_t)(sp - MIPT_LDS_STACK) * spill_stride; p->x = r; p->y = __float_as_uint(t); }
	}
	MIPT_DEV void pop(int sp, uint32_t& r, float& t) const {
		const unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		uint32_t x, y;
		if (sp < MIPT_LDS_STACK) { const lds_uint2* p = (const lds_uint2*)(lds_wave + l * 8u + (unsigned)sp * (MIPT_TRAV_BLOCK * 8u)); x = p->x; y = p->y; }
		else { const glb_uint2* p = spill_wave + l + (size_t)(sp - MIPT_LDS_STACK) * spill_stride; x = p->x; y = p->y; }
		r = x; t = __uint_as_float(y);
	}
};
#define MIPT_DECLARE_LDS_STACK_U(stk, spill_buf) \
	__shared__ uint2 lds_stack_[MIPT_LDS_STACK * MIPT_TRAV_BLOCK]; \
	LdsStackU stk; \
	{ const unsigned w_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); \
	  stk.lds_wave = (unsigned)(uintptr_t)((lds_uint2*)lds_stack_) + w_ * 512u; \
	  stk.spill_wave = (glb_uint2*)(spill_buf) + (size_t)blockIdx.x * MIPT_TRAV_BLOCK + w_ * 64u; stk.spill_stride = (int)(gridDim.x * MIPT_TRAV_BLOCK); }

enum { US_IDLE = 0, US_NEWID = 1, US_NEWRAY = 2, US_READY = 3, US_ALIVE = 4, US_FIN = 5 };

template <bool SHADOW>
__device__ __forceinline__ void traverse_queue_u(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, const DWave& wf,
                                                 const TravQueue tq, int setup_min, int flags, const LdsStackU& stk) {
	const bool force_literal = (flags >> 16) & 1;     // test hook: every ray takes the literal slab chain
	const int alive_low = flags & 0xffff;             // below this many traversing lanes a ready ray is set up at once
	const unsigned n = tq.n_ptr ? *tq.n_ptr : tq.n_imm;
	unsigned* head = tq.head;
	const unsigned* __restrict__ list = tq.list;
	const bool identity = tq.identity;
	const unsigned nwaves = gridDim.x * (MIPT_TRAV_BLOCK / 64), wave_id = blockIdx.x * (MIPT_TRAV_BLOCK / 64) + (threadIdx.x >> 6);
	const unsigned pull_chunk = max(64u, min(MIPT_PULL_CHUNK, (n / (nwaves * MIPT_PULL_DIV)) & ~63u));
	const int nobj = sc->nobj, first_mesh = sc->first_mesh;
	const bool any_alpha = sc->any_alpha != 0;
	const float inf = __int_as_float(0x7f800000);

	// ---- the lane's slot
	int state = US_IDLE;
	float ox = 0, oy = 0, oz = 0;          // READY: world origin; ALIVE: origin in the mesh's frame; FIN: the path's colour
	float ix = 0, iy = 0, iz = 0;          // ALIVE: 1 / direction in the mesh's frame; FIN: the pending direct term
	float dx = 0, dy = 0, dz = 0;          // READY: world direction; ALIVE: direction in the mesh's frame
	float t = 0;                           // closest: best t over the objects visited so far; shadow: t of the current mesh
	float dist = 0;                        // shadow: dist_light
	uint32_t cur = MIPT_NONE;              // ALIVE: inner node or leaf (first triangle still to test, triangles left - 1)
	int sp = 0;
	int obj = 0;                           // object being traversed / next object to visit
	unsigned id = 0;                       // path id (NEWID: the queue index)
	unsigned tri_base = 0;                 // closest: first triangle record of the mesh being traversed
	bool lit = false;                      // the ray takes the literal slab chain (a direction component is exactly 0)
	// ---- the wave's share of the queue
	unsigned chunk_next = wave_id * pull_chunk, chunk_end = min(chunk_next + pull_chunk, n);
	bool drained = chunk_next >= n;
	if (drained) chunk_next = chunk_end = 0;

	auto pop_next = [&]() -> uint32_t {
		while (sp > 0) {
			--sp;
			uint32_t r; float tn;
			stk.pop(sp, r, tn);
			if (!(tn > t)) return r;
		}
		return MIPT_NONE;
	};

	for (;;) {
		// ---- queue entries for idle lanes
		{
			const unsigned long long idle = __ballot(state == US_IDLE);
			if (!drained && idle != 0) {
				if (chunk_next >= chunk_end) {
					unsigned base = 0;
					if (lane_id() == 0) base = atomicAdd(head, pull_chunk) + nwaves * pull_chunk;     // (the first chunk of a wave is static: chunk number = wave number)
					base = __builtin_amdgcn_readfirstlane(base);
					if (base >= n) { drained = true; chunk_next = chunk_end = 0; }
					else { chunk_next = base; chunk_end = min(base + pull_chunk, n); }
				}
				const unsigned take = min((unsigned)__popcll(idle), chunk_end - chunk_next);
				const unsigned prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));      // idle lanes below this one
				if (state == US_IDLE && prefix < take) { id = chunk_next + prefix; state = US_NEWID; obj = first_mesh; }
				chunk_next += take;
			}
			if (__ballot(state != US_IDLE) == 0) { if (drained) break; else continue; }
		}
		// ---- one fetch per lane: 64 bytes as q0..q3 from (pA, pB).  Written as the four instructions they are (the compiler
		//      otherwise narrows every load to the components its consumer reads and sinks it into that consumer's branch:
		//      a dozen partial loads and a wait in front of each consumer).  It does not count these loads in its s_waitcnt
		//      bookkeeping, so the wait is spelled out too; loads return in order, so its own counts stay conservative.
		mipt_v4f q0, q1, q2, q3;
		{
			const float4 *pA = nodes, *pB = nodes;
			if (state == US_ALIVE) {
				pA = (cur & MIPT_LEAF_BIT) ? reinterpret_cast<const float4*>(tris + (cur & MIPT_LEAF_FIRST_MASK)) : nodes + 4 * (size_t)cur;
				pB = pA + 1;
			} else if (state == US_NEWID) {
				pA = identity ? (const float4*)(wf.wgt + id) : reinterpret_cast<const float4*>(list + (id & ~3u));      // (queue lists are 16-byte aligned and padded)
				pB = pA;
			} else if (state == US_NEWRAY) {
				pA = (SHADOW ? wf.sh_o : wf.ray_o) + id; pB = (SHADOW ? wf.sh_d : wf.ray_d) + id;
			} else if (SHADOW && state == US_FIN) {
				pA = wf.out.col + id; pB = wf.sh_c + id;
			}
			asm volatile("" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3));            // (undefined where a lane does not load)
			if (state != US_IDLE && state != US_READY)
				asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "+v"(q0), "+v"(q1) : "v"(pA), "v"(pB) : "memory");
			if (state == US_ALIVE)
				asm volatile("global_load_dwordx4 %0, %2, off offset:32\n\tglobal_load_dwordx4 %1, %2, off offset:48" : "+v"(q2), "+v"(q3) : "v"(pA) : "memory");
			asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "memory");
		}
		// ---- consumers
		if (state == US_ALIVE) {
			if (!(cur & MIPT_LEAF_BIT)) {
				// inner node: both children's boxes (TriangleMesh.cpp:1172-1190)
				const uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
				const f3 s_org = mk3(ox, oy, oz), s_invd = mk3(ix, iy, iz);
				const bool sx = ix >= 0, sy = iy >= 0, sz = iz >= 0;
				float tl, tr;
				bool goleft, goright;
				if (__ballot(lit) != 0) {
					const f3 lmin = mk3(q0.x, q0.z, q1.x), lmax = mk3(q0.y, q0.w, q1.y);
					const f3 rmin = mk3(q1.z, q2.x, q2.z), rmax = mk3(q1.w, q2.y, q2.w);
					goleft = box_test<!SHADOW>(lmin, lmax, s_org, s_invd, sx, sy, sz, tl);
					goright = box_test<!SHADOW>(rmin, rmax, s_org, s_invd, sx, sy, sz, tr);
				} else {
					goleft = box_test_closed<!SHADOW>(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, sx, sy, sz, tl);
					goright = box_test_closed<!SHADOW>(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, sx, sy, sz, tr);
				}
				goleft = goleft && (tl < t); goright = goright && (tr < t);
				if (SHADOW) { goleft = goleft && (tl < dist); goright = goright && (tr < dist); }
				const bool left_first = tl < tr;
				if (goleft && goright) {
					stk.push(sp, left_first ? rref : lref, left_first ? tr : tl); sp++;
					cur = left_first ? lref : rref;
				} else if (goleft) cur = lref;
				else if (goright) cur = rref;
				else cur = pop_next();
			} else {
				// leaf: the next triangle of the sequential loop (TriangleMesh.cpp:1192-1212 / 1294-1316)
				const f3 A = mk3(q0.x, q0.y, q0.z), u = mk3(q0.w, q1.x, q1.y), v = mk3(q1.z, q1.w, q2.x);
				const float invdetm = q2.y, m11 = q2.z, m12 = q2.w, m22 = q3.x;
				const f3 N = mk3(q3.y, q3.z, q3.w);
				const f3 o = mk3(ox, oy, oz), d = mk3(dx, dy, dz);
				bool hit = false;
				float lt, lb = 0.f, lg = 0.f;
				lt = dot(A - o, N) / dot(d, N);                               // Triangle::intersection (TriangleMesh.h:82-104)
				if (!(lt < 0 || lt != lt)) {
					const f3 P = o + lt * d;
					const f3 w = P - A;
					const float b11 = dot(w, u), b21 = dot(w, v);
					const float detb = b11 * m22 - b21 * m12;
					lb = detb * invdetm;
					const float detg = b21 * m11 - b11 * m12;
					lg = detg * invdetm;
					const float alpha = 1 - lb - lg;
					hit = !(lb < 0) && !(lg < 0) && !(alpha < 0);
				}
				const uint32_t first = cur & MIPT_LEAF_FIRST_MASK;
				bool accept = hit && lt < t;
				if (any_alpha && accept) {
					const DObject& o_ = sc->obj[obj];
					if (o_.alpha_test) accept = !alpha_rejects(o_, (int)(first - o_.tri_base), 1 - lb - lg, lb, lg);
				}
				bool decided = false;
				if (accept) {
					t = lt;
					if (SHADOW) decided = (double)lt < (double)dist * 0.999;                              // TriangleMesh.cpp:1309
					else wf.hit[id] = make_float4(lt, lb, lg, __uint_as_float(((unsigned)obj << 27) | (first - tri_base)));
				}
				const uint32_t left = (cur >> 26) & 31u;
				if (SHADOW && decided) { cur = MIPT_NONE; sp = 0; obj = nobj; }                              // occluded: nothing more to visit
				else if (left != 0) cur = MIPT_LEAF_BIT | ((left - 1u) << 26) | (first + 1u);
				else cur = pop_next();
			}
			if (cur == MIPT_NONE) {
				lit = false;
				// mesh finished.  A shadow hit with t >= 0.999 * dist is not an occluder (Geometry.cpp:736): next object.
				if (SHADOW && obj == nobj) { if (tq.vis) tq.vis[id] = 0.f; state = US_IDLE; }
				else if (obj + 1 < nobj) { obj++; state = US_NEWRAY; }                                      // the ray continues in world space: fetch it again
				else if (SHADOW) { if (tq.vis) { tq.vis[id] = 1.f; state = US_IDLE; } else state = US_FIN; }
				else state = US_IDLE;                                                                       // closest hit: the record was written through
			}
		} else if (state == US_NEWID) {
			bool valid = true;
			if (identity) valid = (__float_as_uint(q1.w) & MIPT_WF_VALID) != 0;
			else { const unsigned k = id & 3u; id = __float_as_uint(k == 0 ? q1.x : (k == 1 ? q1.y : (k == 2 ? q1.z : q1.w))); }
			state = valid ? US_NEWRAY : US_IDLE;
		} else if (state == US_NEWRAY) {
			ox = q0.x; oy = q0.y; oz = q0.z; dx = q1.x; dy = q1.y; dz = q1.z;
			if (SHADOW) dist = q0.w;
			else if (obj == first_mesh) {
				// a fresh ray: the stage that created it has already visited the analytic objects in front of the first mesh
				// (mipt_wavefront.h) and left (t, object) with the ray; the hit record is written through from here on
				t = q0.w;
				wf.hit[id] = make_float4(q0.w, 0.f, 0.f, q1.w);
			}
			state = US_READY;
		} else if (SHADOW && state == US_FIN) {
			wf.out.col[id] = make_float4(q0.x + q1.x, q0.y + q1.y, q0.z + q1.z, 0.f);                            // Raytracer.cpp:566
			state = US_IDLE;
		}
		// ---- object loop for rays that are ready (wave-uniform object index: the description comes by scalar loads).
		//      It costs ~100 vector instructions whatever the number of lanes, so ready rays wait for company unless the
		//      wave runs low on traversing lanes.
		{
			const int nready = __popcll(__ballot(state == US_READY));
			if (nready != 0 && (nready >= setup_min || __popcll(__ballot(state == US_ALIVE)) < alive_low)) {
				const bool mine = state == US_READY;
				const f3 ro = mk3(ox, oy, oz), rd = mk3(dx, dy, dz);
				for (int i = first_mesh; i < nobj; i++) {
					if (mine && state == US_READY && obj == i) {
						obj = i + 1;
						// the object's description through the scalar cache (uniform address, constant address space)
						const DObject* op = &sc->obj[i];
						const int otype = *(cst_int*)&op->type;
						const bool ghost = SHADOW && tq.skip_ghosts && *(cst_int*)&op->ghost != 0;     // getColor's shadow rays pass through ghost objects (Geometry.cpp:722, Raytracer.cpp:513)
						if (!ghost && otype == 0) {
							// TriMesh: set up the traversal (TriangleMesh.cpp:1133-1157 / 1239-1263)
							cst_float* m = (cst_float*)op->inv;
							const f3 d = mk3(m[0] * rd.x + m[1] * rd.y + m[2] * rd.z, m[4] * rd.x + m[5] * rd.y + m[6] * rd.z, m[8] * rd.x + m[9] * rd.y + m[10] * rd.z);                       // xf_dir
							const f3 org = mk3(m[0] * ro.x + m[1] * ro.y + m[2] * ro.z + m[3], m[4] * ro.x + m[5] * ro.y + m[6] * ro.z + m[7], m[8] * ro.x + m[9] * ro.y + m[10] * ro.z + m[11]);   // xf_point
							const f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
							cst_float* bmin = (cst_float*)op->root_min; cst_float* bmax = (cst_float*)op->root_max;
							float t_root;
							const float cur_best_t = SHADOW ? inf : t;
							bool enter = box_test<false>(mk3(bmin[0], bmin[1], bmin[2]), mk3(bmax[0], bmax[1], bmax[2]), org, invd, invd.x >= 0, invd.y >= 0, invd.z >= 0, t_root);
							if (enter && t_root > cur_best_t) enter = false;
							if (SHADOW && enter && t_root > dist) enter = false;
							if (enter) {
								ox = org.x; oy = org.y; oz = org.z; ix = invd.x; iy = invd.y; iz = invd.z; dx = d.x; dy = d.y; dz = d.z;
								if (SHADOW) t = inf;
								cur = *(cst_u32*)&op->root_ref; sp = 0; obj = i; state = US_ALIVE;
								if (!SHADOW) tri_base = *(cst_u32*)&op->tri_base;
								lit = force_literal || fabsf(invd.x) == inf || fabsf(invd.y) == inf || fabsf(invd.z) == inf;
							}
						} else if (!ghost && !SHADOW) {
							// a sphere / plane behind the first mesh (shadow rays: tested when the request was made)
							const DObject& o = *op;
							const f3 d = xf_dir(o.inv, rd);
							const f3 org = xf_point(o.inv, ro);
							float th;
							const bool hit = (o.type == 1) ? sphere_test(o, org, d, th) : plane_test(o, org, d, th);
							if (hit && th < t) { t = th; wf.hit[id] = make_float4(th, 0.f, 0.f, __uint_as_float(((unsigned)i << 27) | MIPT_HIT_NOTRI)); }
						}
					}
				}
				if (mine && state == US_READY) {                                           // object list exhausted: the ray is decided
					if (SHADOW) { if (tq.vis) { tq.vis[id] = 1.f; state = US_IDLE; } else state = US_FIN; }
					else state = US_IDLE;
				}
			}
		}
	}
}

// MODE 0: closest-hit queue of depth b.  MODE 1: shadow queue of depth b.
template <int MODE>
__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(MIPT_U_WAVES))) k_wf_traverse_u(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, DWave wf, int b, unsigned n0, int setup_min, int flags) {
	MIPT_DECLARE_LDS_STACK_U(stk, wf.spill);
	TravQueue q;
	if (MODE == 0) { q.list = wf.list[b & 1]; q.n_ptr = b == 0 ? nullptr : &wf.counters[MIPT_CNT_PAIR(b - 1) + 1]; q.n_imm = n0; q.head = &wf.counters[MIPT_CNT_EXT_HEAD(b)]; q.identity = b == 0; q.vis = nullptr; q.skip_ghosts = false; }
	else { q.list = wf.list_sh; q.n_ptr = &wf.counters[MIPT_CNT_PAIR(b)]; q.n_imm = 0; q.head = &wf.counters[MIPT_CNT_SH_HEAD(b)]; q.identity = false; q.vis = nullptr; q.skip_ghosts = false; }
	traverse_queue_u<MODE == 1>(sc, nodes, tris, wf, q, setup_min, flags, stk);
}

// The same on an explicitly described queue (the contribution-queue pipeline, mipt_queue_wave.h).
template <bool SHADOW>
__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(MIPT_U_WAVES))) k_q_traverse_u(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, DWave wf, TravQueue tq, int setup_min, int flags) {
	MIPT_DECLARE_LDS_STACK_U(stk, wf.spill);
	traverse_queue_u<SHADOW>(sc, nodes, tris, wf, tq, setup_min, flags, stk);
}
