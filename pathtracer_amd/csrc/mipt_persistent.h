// mipt_persistent.h — persistent-threads traversal with dynamic ray fetch (the extend and shadow
// stages of the wavefront pipeline).
//
// Measured on the 133k-triangle bench scene with one ray per lane for the lifetime of a wave
// (tools/simd_prof.py): on average only 15.6 of 64 lanes were active in an inner-node step and 18.2
// in a leaf step — rays that miss the mesh, leave it early or find an occluder idle their lane
// until the slowest ray of the wave is done.  Here a lane that finishes its ray is handed the next
// ray of the queue as soon as enough lanes are idle (active-lane compaction by __ballot / popcount
// prefix), so the 64-wide traversal steps stay populated.
//
// Every ray still performs exactly the reference's sequence of operations (objects in index order,
// ordered stack traversal per mesh, same pruning); only WHICH lane runs it and WHEN changes.
#pragma once

#define MIPT_REFILL_THRESHOLD 20        // refill as soon as this many lanes are idle
#define MIPT_PULL_CHUNK 1024u           // ids reserved per global atomic (sub-allocated wave-locally)

struct LaneState {
	f3 org, d, invd;        // ray in the current mesh's frame
	float t;                // closest: best t over the objects visited so far; shadow: t of the current mesh
	float beta, gamma;      // closest: barycentrics of the best triangle
	float dist;             // shadow: dist_light
	uint32_t cur;           // current node reference (scene-wide) or NONE
	int sp;
	int obj;                // object being traversed / next object to visit
	int best;               // closest: packed best hit (MIPT_HIT_MISS = none); shadow: 1 = occluded
	unsigned id;            // path id
};

#define MIPT_NONE 0x7fffffffu

// One object of Scene::intersection / intersection_shadow for the lanes whose next object is `i`
// (i is wave-uniform, so the object's description is fetched with scalar loads).  Returns true when
// the lane has to start traversing mesh i (its traversal state is then set up).
template <bool SHADOW>
__device__ __forceinline__ bool visit_object(const DObject& o, int i, f3 ro, f3 rd, LaneState& st) {
	f3 d = xf_dir(o.inv, rd);
	f3 org = xf_point(o.inv, ro);
	if (o.type != 0) {
		float t;
		bool hit = (o.type == 1) ? sphere_test(o, org, d, t) : plane_test(o, org, d, t);
		if (SHADOW) {
			if (hit && ((double)t < (double)st.dist * 0.999)) st.best = 1;                            // Geometry.cpp:736-740
		} else {
			if (hit && t < st.t) { st.t = t; st.best = (int)(((unsigned)i << 27) | MIPT_HIT_NOTRI); st.beta = 0; st.gamma = 0; }
		}
		return false;
	}
	// TriMesh: set up the traversal (TriangleMesh.cpp:1133-1157 / 1239-1263)
	f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
	bool sx = invd.x >= 0, sy = invd.y >= 0, sz = invd.z >= 0;
	float t_root;
	const float cur_best_t = SHADOW ? __int_as_float(0x7f800000) : st.t;
	bool enter = box_test<false>(ld3(o.root_min), ld3(o.root_max), org, invd, sx, sy, sz, t_root);
	if (enter && t_root > cur_best_t) enter = false;
	if (SHADOW && enter && t_root > st.dist) enter = false;
	if (!enter) return false;
	st.org = org; st.d = d; st.invd = invd;
	if (SHADOW) st.t = cur_best_t;
	st.cur = o.root_ref; st.sp = 0;
	return true;
}

#ifndef MIPT_TRAVERSE_WAVES
#define MIPT_TRAVERSE_WAVES 4
#endif
template <bool SHADOW>
__global__ void __launch_bounds__(MIPT_BLOCK) __attribute__((amdgpu_waves_per_eu(MIPT_TRAVERSE_WAVES))) k_wf_traverse(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, DWave wf, int b, unsigned n0, int refill_threshold, int inner_min) {
	MIPT_DECLARE_LDS_STACK(stk, wf.spill);
	const unsigned n = SHADOW ? wf.counters[4 * b + 2] : (b == 0 ? n0 : wf.counters[4 * b]);
	unsigned* head = &wf.counters[4 * b + (SHADOW ? 3 : 1)];
	const unsigned* __restrict__ list = SHADOW ? wf.list_sh : wf.list[b & 1];
	const bool identity = !SHADOW && b == 0;
	// nodes / tris are kernel arguments (not read from *sc) so that the compiler knows they are global
	// ids reserved per global atomic: large enough to keep the same-address atomic rate low, small
	// enough that every wave of the grid gets several chunks (tail balance)
	const unsigned pull_chunk = max(64u, min(MIPT_PULL_CHUNK, (n / (gridDim.x * (MIPT_BLOCK / 64) * 8u)) & ~63u));
	const unsigned lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;

	LaneState st;
	st.cur = MIPT_NONE; st.sp = 0; st.obj = 0; st.best = 0; st.id = 0; st.t = 0; st.beta = 0; st.gamma = 0; st.dist = 0;
	st.org = mk3(0, 0, 0); st.d = mk3(0, 0, 0); st.invd = mk3(0, 0, 0);
	bool alive = false;                  // the lane holds a ray that is inside a mesh traversal
	bool need = false;                   // the lane holds a ray that must visit its next object(s)
	unsigned chunk_next = 0, chunk_end = 0;   // wave-uniform: ids reserved from the global queue
	bool drained = false;
	const int nobj = sc->nobj;

	auto pop_next = [&]() -> uint32_t {
		while (st.sp > 0) {
			--st.sp;
			uint32_t r; float tn;
			stk.pop(st.sp, r, tn);
			if (!(tn > st.t)) return r;
		}
		return MIPT_NONE;
	};

	for (;;) {
		MIPT_PROF_CLOCK(c0);
		if (alive) MIPT_PROF_COUNT(8)
		// ---- refill idle lanes from the queue
		unsigned long long idle = __ballot(!alive && !need);
		int nidle = __popcll(idle);
		if (!drained && nidle >= refill_threshold) {
			if (chunk_next >= chunk_end) {
				unsigned base = 0;
				if (lane == 0) base = atomicAdd(head, pull_chunk);
				base = __builtin_amdgcn_readfirstlane(base);
				if (base >= n) { drained = true; chunk_next = chunk_end = 0; }
				else { chunk_next = base; chunk_end = min(base + pull_chunk, n); }
			}
			unsigned take = min((unsigned)nidle, chunk_end - chunk_next);
			if (!alive && !need) {
				unsigned prefix = (unsigned)__popcll(idle & below);
				if (prefix < take) {
					unsigned idx = chunk_next + prefix;
					unsigned id = identity ? idx : list[idx];
					bool valid = true;
					if (identity) valid = (__float_as_uint(wf.wgt[id].w) & MIPT_WF_VALID) != 0;
					if (valid) {
						MIPT_PROF_COUNT(10)
						st.id = id; st.obj = 0; need = true;
						if (SHADOW) { st.dist = wf.sh_o[id].w; st.best = 0; }
						else { st.t = __int_as_float(0x7f800000); st.best = (int)MIPT_HIT_MISS; st.beta = 0; st.gamma = 0; }
					}
				}
			}
			chunk_next += take;
		}
		// ---- object loop (wave-uniform index): new rays start at object 0, rays that just left a mesh
		//      continue behind it; a ray stops at the first mesh it has to traverse
		if (__ballot(need)) {
			f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
			if (need) {
				float4 o4 = SHADOW ? wf.sh_o[st.id] : wf.ray_o[st.id];
				float4 d4 = SHADOW ? wf.sh_d[st.id] : wf.ray_d[st.id];
				ro = mk3(o4.x, o4.y, o4.z); rd = mk3(d4.x, d4.y, d4.z);
			}
			if (need) MIPT_PROF_COUNT(6)
			for (int i = 0; i < nobj; i++) {
				if (need && st.obj == i) {
					if (visit_object<SHADOW>(sc->obj[i], i, ro, rd, st)) { need = false; alive = true; }
					else if (SHADOW && st.best) st.obj = nobj;          // occluded: decided
					else st.obj = i + 1;
				}
			}
			if (need) {                                               // object list exhausted: the ray is decided
				if (SHADOW) {
					if (!st.best) {
						float4 c = wf.out.col[st.id], pc = wf.sh_c[st.id];
						wf.out.col[st.id] = make_float4(c.x + pc.x, c.y + pc.y, c.z + pc.z, 0.f);    // Raytracer.cpp:566
					}
				} else {
					wf.hit[st.id] = make_float4(st.t, st.beta, st.gamma, __uint_as_float((unsigned)st.best));
				}
				need = false;
			}
		}
		{
			const int nalive = __popcll(__ballot(alive));
			if (nalive == 0) { if (drained) break; else continue; }
			if (!drained && 64 - nalive >= refill_threshold) continue;      // rays that missed every mesh left their lanes idle again: top up first
		}
		MIPT_PROF_CLOCK(c1);
		MIPT_PROF_CYCLES(12, c0, c1)

		// ---- inner-node phase: every live lane descends until it holds a leaf or runs out of nodes
		//      (the phase also ends when fewer than inner_min lanes are still descending while others
		//      already wait with a leaf: the stragglers simply resume in the next round)
		{
			const bool sx = st.invd.x >= 0, sy = st.invd.y >= 0, sz = st.invd.z >= 0;     // signs[k] (TriangleMesh.cpp:1145)
			for (;;) {
				const bool inner = alive && st.cur != MIPT_NONE && !(st.cur & MIPT_LEAF_BIT);
				const unsigned long long mi = __ballot(inner);
				if (mi == 0) break;
				if (__popcll(mi) < inner_min && __ballot(alive && !inner) != 0) break;
				if (!inner) continue;
				MIPT_PROF_COUNT(0)
				const float4* q = nodes + 4 * (size_t)st.cur;
				float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
				f3 lmin = mk3(q0.x, q0.y, q0.z), lmax = mk3(q0.w, q1.x, q1.y);
				f3 rmin = mk3(q1.z, q1.w, q2.x), rmax = mk3(q2.y, q2.z, q2.w);
				uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
				float tl, tr;
				bool goleft, goright;
				if (SHADOW) {
					goleft = box_test<false>(lmin, lmax, st.org, st.invd, sx, sy, sz, tl) && (tl < st.t) && (tl < st.dist);
					goright = box_test<false>(rmin, rmax, st.org, st.invd, sx, sy, sz, tr) && (tr < st.t) && (tr < st.dist);
				} else {
					goleft = box_test<true>(lmin, lmax, st.org, st.invd, sx, sy, sz, tl) && (tl < st.t);
					goright = box_test<true>(rmin, rmax, st.org, st.invd, sx, sy, sz, tr) && (tr < st.t);
				}
				if (goleft && goright) {
					if (tl < tr) { stk.push(st.sp, rref, tr); st.sp++; st.cur = lref; }
					else { stk.push(st.sp, lref, tl); st.sp++; st.cur = rref; }
				} else if (goleft) st.cur = lref;
				else if (goright) st.cur = rref;
				else st.cur = pop_next();
			}
		}
		MIPT_PROF_CLOCK(c2);
		MIPT_PROF_CYCLES(13, c1, c2)
		// ---- leaf phase
		if (alive && st.cur != MIPT_NONE && (st.cur & MIPT_LEAF_BIT)) {
			MIPT_PROF_COUNT(2)
			int first = (int)(st.cur & MIPT_LEAF_FIRST_MASK);
			int count = (int)((st.cur >> 26) & 31u) + 1;
			bool decided = false;
			for (int i = first; i < first + count; i++) {
				MIPT_PROF_COUNT(4)
				float lt, lb, lg;
				if (tri_test(tris + i, st.org, st.d, lt, lb, lg)) {
					bool accept = lt < st.t;
					int local = 0;                                   // mesh-local triangle index, only needed for accepted hits
					if (accept) {
						const DObject& o = sc->obj[st.obj];
						local = i - (int)o.tri_base;
						if (o.alpha_test) accept = !alpha_rejects(o, local, 1 - lb - lg, lb, lg);
					}
					if (accept) {
						st.t = lt;
						if (SHADOW) {
							if ((double)lt < (double)st.dist * 0.999) { decided = true; break; }           // TriangleMesh.cpp:1309
						} else {
							st.best = (int)(((unsigned)st.obj << 27) | (unsigned)local); st.beta = lb; st.gamma = lg;
						}
					}
				}
			}
			if (SHADOW && decided) { st.best = 1; st.cur = MIPT_NONE; st.sp = 0; }
			else st.cur = pop_next();
		}
		MIPT_PROF_CLOCK(c3);
		MIPT_PROF_CYCLES(14, c2, c3)
		// ---- mesh finished: the ray goes on with the objects behind it (next iteration's object loop)
		if (alive && st.cur == MIPT_NONE) {
			alive = false; need = true;
			// shadow: a hit with t >= 0.999*dist is not an occluder (Geometry.cpp:736) -> next object
			st.obj = (SHADOW && st.best) ? nobj : st.obj + 1;
		}
	}
}
