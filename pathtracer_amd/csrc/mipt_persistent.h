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
// Every ray still performs exactly the reference's sequence of operations (objects in index order —
// the analytic ones in front of the first mesh already by the stage that created the ray —, ordered
// stack traversal per mesh, same pruning); only WHICH lane runs it and WHEN changes.
#pragma once

#define MIPT_REFILL_THRESHOLD 36        // refill as soon as this many lanes are idle (a refill runs the object loop for few lanes: measured optimum)
#ifndef MIPT_EXTEND_WAVES
#define MIPT_EXTEND_WAVES 7             // closest-hit kernel: 7 waves/SIMD with the derived triangle terms, made possible by MIPT_HIT_WRITE_THROUGH (round 1: 6 waves with the derived terms beat 7 with the fourth load)
#endif
#ifndef MIPT_DERIVE_SHADOW
#define MIPT_DERIVE_SHADOW 1           // the any-hit traversal loads 48 of the 64 bytes of a triangle record and derives N, m22 (-7 % stage time)
#endif
#ifndef MIPT_DERIVE_EXTEND
#define MIPT_DERIVE_EXTEND 1           // the closest-hit traversal too, at 6 waves per SIMD (84 registers; at 7 waves / 72 registers deriving spills: +15 %)
#endif
#ifndef MIPT_TRAV_BLOCK
#define MIPT_TRAV_BLOCK 256
#endif
#ifndef MIPT_PULL_CHUNK
#define MIPT_PULL_CHUNK 512u            // ids reserved per global atomic (sub-allocated wave-locally); swept 128 .. 4096 on C2: 512 and 256 best, 1024 +0.5 %, 4096 +3.7 % (tails), 128 +2 %
#endif
// MIPT_PREFETCH_IDS: the 64 queue entries behind the refill cursor are requested right after a refill (lane l: entry cursor + l)
// and handed to the lanes that take them at the next refill with one ds_bpermute: a refill then waits for ONE dependent round
// trip (the ray of the id) instead of two (the id, then the ray).  Measured on C2: extend -0.3 %, shadow +0.8 %: the refill's
// latency is covered by the other six waves of the SIMD.  Off.
#ifndef MIPT_PREFETCH_IDS
#define MIPT_PREFETCH_IDS 0
#endif
// MIPT_UNIFORM_FETCH: see the inner-node phase of traverse_queue.  Measured: EXPERIMENTS.md (round 6).
#ifndef MIPT_UNIFORM_FETCH
#define MIPT_UNIFORM_FETCH 0
#endif
#ifndef MIPT_PULL_DIV
#define MIPT_PULL_DIV 8u                // chunks per wave of the grid when the queue is short
#endif

struct LaneState {
	mipt_f2 o_xy, i_xy, oz_iz;   // ray in the current mesh's frame: (org.x, org.y), (invd.x, invd.y), (org.z, invd.z) as register pairs
	f3 d;                        // for the packed slab test (box_test_pairs)
	float t;                // closest: best t over the objects visited so far; shadow: t of the current mesh
	float beta, gamma;      // closest: barycentrics of the best triangle
	float dist;             // shadow: dist_light
	uint32_t cur;           // current node reference (scene-wide) or NONE
	int sp;
	int obj;                // object being traversed / next object to visit
	int best;               // closest: packed best hit (MIPT_HIT_MISS = none); shadow: 1 = occluded
	unsigned id;            // path id
	uint64_t rng;           // reservoir: the engine of the sample that asked
	int count;              // reservoir: intersections accepted so far
};

// Round 4 built a per-wave READY LIST on top of this loop: rays fetched, transformed and root-tested by every lane of a
// fill (idle lanes for themselves, busy lanes for a 28-entry LDS list), handed to lanes that run out of nodes at the one wave-level
// point in front of the inner phase, the inner phase ended at 32 descending lanes.  It did what the scheduling simulator
// (tests/tools/sched_sim.c) predicted — 44.8 instead of 38 lanes per inner step, vector-memory instructions -15 %, vector -10 %, scalar
// -9.5 % per launch, parity suite green — and the closest-hit stage of configs[2] ran 2.3 % SLOWER (any-hit +5 %, configs[3] +6 %):
// L1->L2 requests +17 %, L2 misses +8 %, tag-conflict stalls +77 %.  The loop does not run at an instruction rate (one more load per
// inner step, +20 % vector-memory instructions, costs 6 %): DESIGN.md 4e, profiles/r4_b_*; the code is at git tag
// r4-ready-list-experiment.  Kept from it: the lane state in st.cur (below) and v_mbcnt prefix counts.
// A lane's state is kept in st.cur beside the node reference (no separate flags: at 72 registers two flags cost two registers):
// an inner node (< MIPT_ST_NEED), a leaf (bit 31), or one of
#define MIPT_ST_NEED 0x7ffffffdu        // the lane's ray must visit its next object(s)
#define MIPT_ST_IDLE 0x7ffffffeu        // the lane holds no ray
#define MIPT_NONE 0x7fffffffu           // the ray has just left its mesh (it counts as alive until the end of the outer iteration)
#define MIPT_L_ALIVE (st.cur - MIPT_ST_NEED >= 2u)
#define MIPT_L_NEED (st.cur == MIPT_ST_NEED)
#define MIPT_L_IDLE (st.cur == MIPT_ST_IDLE)
// MIPT_HIT_WRITE_THROUGH: the closest-hit kernel does not carry (beta, gamma, best) of the best hit in registers: a ray's record
// is written when the ray is fetched (what the analytic objects left) and rewritten at every accepted hit.  That frees the
// registers that let the kernel run 7 waves per SIMD with the derived triangle terms (6 spilled values, all loop constants
// reloaded outside the inner loop): extend -1.9 % on C1 / C2, +0.8 % on C3 (more accepted hits per ray inside the glass).
// (Keeping `best` to skip the first write for rays that accept something: more spills, slower.)
#ifndef MIPT_HIT_WRITE_THROUGH
#define MIPT_HIT_WRITE_THROUGH 1
#endif

// One object of Scene::intersection / intersection_shadow for the lanes whose next object is `i`
// (i is wave-uniform, so the object's description is fetched with scalar loads).  Returns true when
// the lane has to start traversing mesh i (its traversal state is then set up).
template <bool SHADOW>
__device__ __forceinline__ bool visit_object(const DObject& o, int i, f3 ro, f3 rd, LaneState& st, bool skip_ghosts) {
	if (SHADOW && skip_ghosts && o.ghost) return false;          // getColor's shadow rays pass through ghost objects (Geometry.cpp:722, Raytracer.cpp:513)
	f3 d = xf_dir(o.inv, rd);
	f3 org = xf_point(o.inv, ro);
	if (o.type != 0) {
		if (SHADOW) return false;                         // spheres / planes were tested when the request was made
		float t;
		bool hit = (o.type == 1) ? sphere_test(o, org, d, t) : plane_test(o, org, d, t);
		{
			if (hit && t < st.t) { st.t = t; st.best = (int)(MIPT_HIT_ANALYTIC | (unsigned)i); st.beta = 0; st.gamma = 0; }     // (write-through: by the caller)
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
	st.o_xy = (mipt_f2){org.x, org.y}; st.i_xy = (mipt_f2){invd.x, invd.y}; st.oz_iz = (mipt_f2){org.z, invd.z}; st.d = d;
	if (SHADOW) st.t = cur_best_t;
	st.cur = o.root_ref; st.sp = 0;
	return true;
}

#ifndef MIPT_TRAVERSE_WAVES
#define MIPT_TRAVERSE_WAVES 7
#endif
// Which rays a launch works on and where an any-hit result goes.
struct TravQueue {
	const unsigned* list;        // path ids (ignored when identity)
	const unsigned* n_ptr;       // number of entries, read on the device (nullptr: n_imm)
	unsigned n_imm;
	unsigned* head;              // shared chunk counter of this queue
	bool identity;               // entry k is path id k (depth 0 of the wavefront pipeline)
	float* vis;                  // shadow rays: nullptr = add the pending direct term to the path's colour when the light sample is
	                             // visible (wavefront pipeline); else write 1.f (visible) / 0.f (occluded) to vis[id] (contribution-queue pipeline)
	bool skip_ghosts;            // shadow rays ignore ghost objects
	bool valid_in_ray;           // identity queues: a slot without a path is marked by a NaN in its ray origin's .w (MIPT_WF_DEAD_RAY, the wavefront
	                             // pipeline since round 5) instead of by the VALID bit of its weight word (the contribution-queue pipeline)
};
// One queue of one depth: SHADOW = false the closest-hit rays of depth b (Scene::intersection), SHADOW = true the
// light-sample rays of depth b (Scene::intersection_shadow).  Called by every wave of the grid; returns when the
// queue is drained and all rays this wave fetched are finished.
// RESV (with SHADOW = false): the subsurface probes of the contribution-queue pipeline, TriMesh::reservoir_sampling_intersection
// (TriangleMesh.cpp:1321-1426, restated per thread as mesh_reservoir in mipt_compositing.h): the request is a ray in the
// frame of ONE mesh (wf.ray_o.w = max_t, wf.ray_d.w = the object), the traversal is the closest-hit one with the far bound
// fixed at max_t, and every triangle hit in [0, max_t) draws one number from the sample's engine (wf.rng) in visiting order and
// replaces the kept one with probability 1/count.  Result: wf.hit (w = the mesh-local triangle or MIPT_HIT_MISS), wf.rng.
template <bool SHADOW, bool RESV = false>
__device__ __forceinline__ void traverse_queue(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, const DWave& wf,
                                               const TravQueue tq, int refill_threshold, int inner_min_flags, LdsStack& stk, unsigned char* leafmap) {
	const int inner_min = inner_min_flags & 0xffff;
	const bool force_literal = (inner_min_flags >> 16) & 1;     // test hook: every ray takes the literal slab chain
	const unsigned lane_limit = ((inner_min_flags >> 17) & 127) ? ((inner_min_flags >> 17) & 127) : 64u;   // probe: only the first lanes take rays
	const unsigned n = tq.n_ptr ? *tq.n_ptr : tq.n_imm;
	unsigned* head = tq.head;
	const unsigned* __restrict__ list = tq.list;
	const bool identity = tq.identity;
	// nodes / tris are kernel arguments (not read from *sc) so that the compiler knows they are global
	// ids reserved per global atomic: large enough to keep the same-address atomic rate low (one costs ~11 ns
	// chip-wide), small enough that every wave of the grid gets several chunks (tail balance)
	// (wave-uniform values are made scalar explicitly: derived from threadIdx or through min / max they would sit in vector registers,
	//  and at 72 registers the compiler kept four of them — and the leaf map's address — in scratch, reloaded in the refill and in
	//  every leaf round: scratch accesses are vector-memory instructions, the resource this kernel runs out of)
	const unsigned nwaves = gridDim.x * (MIPT_TRAV_BLOCK / 64), wave_id = blockIdx.x * (MIPT_TRAV_BLOCK / 64) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	bool first_pull = true;
	const unsigned pull_chunk = (unsigned)__builtin_amdgcn_readfirstlane((int)max(64u, min(MIPT_PULL_CHUNK, (n / (gridDim.x * (MIPT_TRAV_BLOCK / 64) * MIPT_PULL_DIV)) & ~63u)));
	const unsigned lane = lane_id();
	// set bits of a wave mask below the calling lane (v_mbcnt takes the mask from scalar registers: no per-lane `lanes below me` mask is kept)
	auto below_count = [](unsigned long long m) -> unsigned { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };

	LaneState st;
	st.cur = MIPT_ST_IDLE; st.sp = 0; st.obj = 0; st.best = 0; st.id = 0; st.t = 0; st.beta = 0; st.gamma = 0; st.dist = 0; st.rng = 0; st.count = 0;
	st.o_xy = (mipt_f2){0.f, 0.f}; st.i_xy = (mipt_f2){0.f, 0.f}; st.oz_iz = (mipt_f2){0.f, 0.f}; st.d = mk3(0, 0, 0);
	unsigned chunk_next = 0, chunk_end = 0;   // wave-uniform: ids reserved from the global queue
	unsigned ahead = 0, ahead_at = 0xffffffffu;   // MIPT_PREFETCH_IDS: entry ahead_at + lane of the list (valid while ahead_at == chunk_next)
	bool drained = false;
	const int nobj = sc->nobj, first_mesh = sc->first_mesh;
	const bool any_alpha = sc->any_alpha != 0;
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
		if (MIPT_L_ALIVE) MIPT_PROF_COUNT(8)
		// ---- refill idle lanes from the queue
		unsigned long long idle = __ballot(MIPT_L_IDLE && lane < lane_limit);
		int nidle = __popcll(idle);
		if (!drained && nidle >= refill_threshold) {
			if (chunk_next >= chunk_end) {
				// The first chunk of every wave is assigned statically (chunk number = wave number), later ones come
				// from the shared counter: 8192 waves hitting one address at launch cost ~90 us (one same-address
				// atomic ~11 ns), and a queue shorter than the grid's first chunks needs no atomic at all.
				unsigned base = 0;
				if (first_pull) { base = wave_id * pull_chunk; first_pull = false; }
				else if (lane == 0) base = atomicAdd(head, pull_chunk) + nwaves * pull_chunk;
				base = __builtin_amdgcn_readfirstlane(base);
				if (base >= n) { drained = true; chunk_next = chunk_end = 0; }
				else { chunk_next = base; chunk_end = min(base + pull_chunk, n); }
			}
			unsigned take = min((unsigned)nidle, chunk_end - chunk_next);
			const unsigned my_prefix = below_count(idle);
			unsigned id_ahead = 0;
			const bool use_ahead = MIPT_PREFETCH_IDS && !identity && ahead_at == chunk_next;       // wave-uniform
			if (use_ahead) id_ahead = (unsigned)__builtin_amdgcn_ds_bpermute((int)((my_prefix & 63u) << 2), (int)ahead);
			if (MIPT_L_IDLE && lane < lane_limit) {
				unsigned prefix = my_prefix;
				if (prefix < take) {
					unsigned idx = chunk_next + prefix;
					unsigned id = identity ? idx : (use_ahead ? id_ahead : list[idx]);
					bool valid = true;
					if (identity && !tq.valid_in_ray) valid = (__float_as_uint(wf.wgt[id].w) & MIPT_WF_VALID) != 0;
					if (valid) {
						MIPT_PROF_COUNT(10)
						// the stage that created the ray has already visited the analytic objects (mipt_wavefront.h):
						// closest-hit rays arrive with the (t, object) found in front of the first mesh, shadow rays
						// arrive only if no sphere / plane occludes them
						st.id = id; st.obj = first_mesh; st.cur = MIPT_ST_NEED;
						if (SHADOW) st.best = 0;                                         // (dist_light arrives with the ray: object loop below)
						else if (!RESV) { st.beta = 0; st.gamma = 0; }               // st.t / st.best arrive with the ray (object loop below)
					}
				}
			}
			chunk_next += take;
			if (MIPT_PREFETCH_IDS && !identity) {
				const unsigned e = chunk_next + lane;
				ahead = e < chunk_end ? list[e] : 0u;
				ahead_at = chunk_next;
			}
		}
		// ---- object loop (wave-uniform index): new rays start at object 0, rays that just left a mesh
		//      continue behind it; a ray stops at the first mesh it has to traverse
		if (RESV) {
			if (MIPT_L_NEED) {
				const float4 o4 = wf.ray_o[st.id], d4 = wf.ray_d[st.id];
				const DObject& o = sc->obj[__float_as_uint(d4.w)];
				const f3 org = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
				const f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
				float t_root;
				bool enter = box_test<false>(ld3(o.root_min), ld3(o.root_max), org, invd, invd.x >= 0, invd.y >= 0, invd.z >= 0, t_root);
				if (enter && t_root > o4.w) enter = false;
				st.cur = MIPT_ST_IDLE;
				if (enter) {
					st.o_xy = (mipt_f2){org.x, org.y}; st.i_xy = (mipt_f2){invd.x, invd.y}; st.oz_iz = (mipt_f2){org.z, invd.z}; st.d = d;
					st.t = o4.w; st.cur = o.root_ref; st.sp = 0; st.obj = (int)__float_as_uint(d4.w);
					st.count = 0; st.best = (int)MIPT_HIT_MISS; st.dist = 0.f; st.beta = 0.f; st.gamma = 0.f;
					{ const uint2 rs = wf.rng[st.id]; st.rng = (uint64_t)rs.x | ((uint64_t)rs.y << 32); }
				} else wf.hit[st.id] = make_float4(0.f, 0.f, 0.f, __uint_as_float(MIPT_HIT_MISS));      // no draw: the engine stays as it is
			}
		} else if (__ballot(MIPT_L_NEED) != 0) {
			f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
			// (a ray that has left its last object only has to be settled below: it is not fetched again — 6 % of the kernel's
			// vector-memory instructions on a one-mesh scene, and the kernel runs at the CU's rate of those: DESIGN.md 4d)
			if (MIPT_L_NEED && (st.obj < nobj || (!SHADOW && st.obj == first_mesh))) {     // (a scene without a mesh: a fresh closest-hit ray still brings its record)
				float4 o4 = SHADOW ? wf.sh_o[st.id] : wf.ray_o[st.id];
				float4 d4 = SHADOW ? wf.sh_d[st.id] : wf.ray_d[st.id];
				ro = mk3(o4.x, o4.y, o4.z); rd = mk3(d4.x, d4.y, d4.z);
				if (SHADOW) st.dist = o4.w;
				if (!SHADOW && st.obj == first_mesh) {                                                       // a fresh ray: what the analytic objects left
					if (identity && tq.valid_in_ray && o4.w != o4.w) st.cur = MIPT_ST_IDLE;                  // no path in this slot (MIPT_WF_DEAD_RAY)
					else {
						st.t = o4.w; st.best = (int)__float_as_uint(d4.w);
						if (MIPT_HIT_WRITE_THROUGH) wf.hit[st.id] = make_float4(o4.w, 0.f, 0.f, d4.w);
					}
				}
			}
			if (MIPT_L_NEED) MIPT_PROF_COUNT(6)
			for (int i = first_mesh; i < nobj; i++) {
				if (MIPT_L_NEED && st.obj == i) {
					const float t_before = st.t;
					const bool enter_mesh = visit_object<SHADOW>(sc->obj[i], i, ro, rd, st, tq.skip_ghosts);
					if (MIPT_HIT_WRITE_THROUGH && !SHADOW && sc->obj[i].type != 0 && st.t < t_before) wf.hit[st.id] = make_float4(st.t, 0.f, 0.f, __uint_as_float(MIPT_HIT_ANALYTIC | (unsigned)i));
					if (enter_mesh) {}                                   // (st.cur is the mesh's root now)
					else if (SHADOW && st.best) st.obj = nobj;          // occluded: decided
					else st.obj = i + 1;
				}
			}
			if (MIPT_L_NEED) {                                        // object list exhausted: the ray is decided
				if (SHADOW) {
					if (tq.vis) tq.vis[st.id] = st.best ? 0.f : 1.f;
					else if (!st.best) {
						float4 c = wf.out.col[st.id], pc = wf.sh_c[st.id];
						wf.out.col[st.id] = make_float4(c.x + pc.x, c.y + pc.y, c.z + pc.z, 0.f);    // Raytracer.cpp:566
					}
				} else if (!MIPT_HIT_WRITE_THROUGH) {
					wf.hit[st.id] = make_float4(st.t, st.beta, st.gamma, __uint_as_float((unsigned)st.best));
				}
				st.cur = MIPT_ST_IDLE;
			}
		}
		{
			const int nalive = __popcll(__ballot(MIPT_L_ALIVE));
			if (nalive == 0) { if (drained) break; else continue; }
			if (!drained && (int)lane_limit - nalive >= refill_threshold) continue;      // rays that missed every mesh left their lanes idle again: top up first
		}
		MIPT_PROF_CLOCK(c1);
		MIPT_PROF_CYCLES(12, c0, c1)

		// ---- inner-node phase: every live lane descends until it holds a leaf or runs out of nodes
		//      (the phase also ends when fewer than inner_min lanes are still descending while others
		//      already wait with a leaf: the stragglers simply resume in the next round)
		{
			const f3 s_org = mk3(st.o_xy.x, st.o_xy.y, st.oz_iz.x), s_invd = mk3(st.i_xy.x, st.i_xy.y, st.oz_iz.y);
			const bool sx = s_invd.x >= 0, sy = s_invd.y >= 0, sz = s_invd.z >= 0;     // signs[k] (TriangleMesh.cpp:1145)
			// a direction component that is exactly 0 can make a slab product NaN: such rays (and the lanes that step
			// together with them) use the literal early-out chain
			const float inf = __int_as_float(0x7f800000);
			const bool literal = MIPT_L_ALIVE && (force_literal || fabsf(s_invd.x) == inf || fabsf(s_invd.y) == inf || fabsf(s_invd.z) == inf);
			for (;;) {
				const bool inner = st.cur < MIPT_ST_NEED;
				const unsigned long long mi = __ballot(inner);
				if (mi == 0) break;
				if (__popcll(mi) < inner_min && __ballot(st.cur >= MIPT_NONE) != 0) break;      // (a leaf, or a ray that has run out of nodes)
				if (!inner) continue;
				MIPT_PROF_COUNT(0)
#ifdef MIPT_PROFILE_SIMD
				{   // how often the descending lanes of a wave stand on ONE node (the step a scalar fetch could serve)
					const unsigned c0_ = (unsigned)__builtin_amdgcn_readfirstlane((int)st.cur);
					const unsigned long long same_ = __ballot(st.cur == c0_);
					if (same_ == mi) MIPT_PROF_COUNT(32)
					else if (2 * __popcll(same_) >= __popcll(mi)) MIPT_PROF_COUNT(34)
				}
#endif
				float4 q0, q1, q2, q3;
#if MIPT_UNIFORM_FETCH
				// Every descending lane on ONE node (camera rays of an 8 x 8 pixel block at the top of the tree: 23 % of the inner steps of depth 0 on
				// configs[2], tools/simd_prof.py): the node comes through the scalar cache, one s_load_dwordx16 instead of four vector-memory
				// instructions — the resource this kernel runs out of.  Same node, same arithmetic: the visit sequence is unchanged.
				const uint32_t cur0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.cur);
				if (!SHADOW && !RESV && __builtin_amdgcn_readfirstlane((int)(__ballot(st.cur == cur0) == mi))) {
					typedef float mipt_f16 __attribute__((ext_vector_type(16)));
					mipt_f16 nd;
					const float4* qs = nodes + 4 * (size_t)cur0;
					asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(nd) : "s"(qs) : "memory");
					q0 = make_float4(nd[0], nd[1], nd[2], nd[3]); q1 = make_float4(nd[4], nd[5], nd[6], nd[7]);
					q2 = make_float4(nd[8], nd[9], nd[10], nd[11]); q3 = make_float4(nd[12], nd[13], nd[14], nd[15]);
				} else
#endif
				{
					const float4* q = nodes + 4 * (size_t)st.cur;
					q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
#if defined(MIPT_PROBE_EXTRA_LOAD)
					// measurement builds only (tools/build_variant.sh): what one more 16-byte access per lane and inner step costs — 1: from the node's
					// own line (a fifth L1 lookup, no new line), 2: from the line behind it (a lookup AND a line the step does not need),
					// 3: the node's line again as ONE more instruction of 4 bytes per lane (instruction cost without the data return)
					{
						float4 extra;
						const float4* qe = q + (MIPT_PROBE_EXTRA_LOAD == 2 ? 4 : 2);
						if (MIPT_PROBE_EXTRA_LOAD == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); extra = make_float4(0.f, 0.f, 0.f, 0.f); }      // the control: the wait alone
						else if (MIPT_PROBE_EXTRA_LOAD == 3) { float e1; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(e1) : "v"(qe) : "memory"); extra = make_float4(e1, 0.f, 0.f, 0.f); }
						else asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(extra) : "v"(qe) : "memory");
						if (__float_as_uint(extra.x) == 0x7fc12345u) q3.w = extra.y;          // never true: keeps the load alive
					}
#endif
				}
				uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
				float tl, tr;
				bool goleft, goright;
				if (__ballot(literal) != 0) {
					f3 lmin = mk3(q0.x, q0.z, q1.x), lmax = mk3(q0.y, q0.w, q1.y);
					f3 rmin = mk3(q1.z, q2.x, q2.z), rmax = mk3(q1.w, q2.y, q2.w);
					goleft = box_test<!SHADOW>(lmin, lmax, s_org, s_invd, sx, sy, sz, tl);
					goright = box_test<!SHADOW>(rmin, rmax, s_org, s_invd, sx, sy, sz, tr);
				} else {
					const mipt_f2 LX = {q0.x, q0.y}, LY = {q0.z, q0.w}, LZ = {q1.x, q1.y}, RX = {q1.z, q1.w}, RY = {q2.x, q2.y}, RZ = {q2.z, q2.w};
					goleft = box_test_pairs<!SHADOW>(LX, LY, LZ, st.o_xy, st.i_xy, st.oz_iz, sx, sy, sz, tl);
					goright = box_test_pairs<!SHADOW>(RX, RY, RZ, st.o_xy, st.i_xy, st.oz_iz, sx, sy, sz, tr);
				}
				goleft = goleft && (tl < st.t); goright = goright && (tr < st.t);
				if (SHADOW) { goleft = goleft && (tl < st.dist); goright = goright && (tr < st.dist); }
				if (goleft && goright) {
					if (st.sp >= 6) MIPT_PROF_COUNT(26)
					if (st.sp >= 8) MIPT_PROF_COUNT(28)
					if (st.sp >= 10) MIPT_PROF_COUNT(30)
					if (tl < tr) { stk.push(st.sp, rref, tr); st.sp++; st.cur = lref; }
					else { stk.push(st.sp, lref, tl); st.sp++; st.cur = rref; }
				} else if (goleft) st.cur = lref;
				else if (goright) st.cur = rref;
				else st.cur = pop_next();
			}
		}
		MIPT_PROF_CLOCK(c2);
		MIPT_PROF_CYCLES(13, c1, c2)
		// ---- leaf phase.  The (ray, triangle) tests of all lanes that hold a leaf are packed densely over the
		//      wave: a leaf has at most 4 triangles and typically a quarter of the lanes hold one, so the per-lane
		//      loop would run 4 rounds at ~15 of 64 lanes (measured).  Test j of the packed list runs on lane j:
		//      it fetches its owner's ray with ds_bpermute, and each owner then walks the results of ITS triangles
		//      in leaf order with the reference's strict '<' (TriangleMesh.cpp:1198), so the accepted triangle is
		//      the one the sequential loop accepts.  Leaves with more triangles (degenerate splits) and scenes with
		//      alpha-tested meshes take the per-lane loop below.
		{
			const bool leaf = (int)st.cur < 0;
			const int first = (int)(st.cur & MIPT_LEAF_FIRST_MASK);
			const int count = leaf ? (int)((st.cur >> 26) & 31u) + 1 : 0;
			bool per_lane = leaf;
			if (leaf) MIPT_PROF_COUNT(2)
			if (!any_alpha) {
				const int cnt = count <= 4 ? count : 0;
				const unsigned long long b1 = __ballot(cnt >= 1), b2 = __ballot(cnt >= 2), b3 = __ballot(cnt >= 3), b4 = __ballot(cnt >= 4);
				if (b1 != 0) {
					unsigned pfx = 0;                 // tests of the lanes below this one (v_mbcnt accumulates)
					for (const unsigned long long m : {b1, b2, b3, b4}) pfx = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, pfx));
					const int prefix = (int)pfx;
					const int total = __popcll(b1) + __popcll(b2) + __popcll(b3) + __popcll(b4);
					for (int k = 0; k < 4; k++) if (k < cnt) leafmap[prefix + k] = (unsigned char)(lane | ((unsigned)k << 6));
					__builtin_amdgcn_wave_barrier();
					float cur_t = st.t, wb = 0.f, wg = 0.f;
					int win = -1;
					bool decided = false;
					for (int base = 0; base < total; base += 64) {
						const int j = base + (int)lane;
						const unsigned m = j < total ? (unsigned)leafmap[j] : 0u;
						const int src = (int)((m & 63u) << 2), slot = (int)(m >> 6);
						const int f = __builtin_amdgcn_ds_bpermute(src, first);
						f3 ro, rd;
						ro.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(st.o_xy.x)));
						ro.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(st.o_xy.y)));
						ro.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(st.oz_iz.x)));
						rd.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(st.d.x)));
						rd.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(st.d.y)));
						rd.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(st.d.z)));
						float lt = __int_as_float(0x7f800000), lb = 0.f, lg = 0.f;
						if (j < total) {
							MIPT_PROF_COUNT(4)
							float a, bb, gg;
							if (tri_test<SHADOW ? (MIPT_DERIVE_SHADOW != 0) : (MIPT_DERIVE_EXTEND != 0)>(tris + f + slot, ro, rd, a, bb, gg)) { lt = a; lb = bb; lg = gg; }
						}
						int wj = 0;
						bool upd = false;
#pragma unroll
						for (int k = 0; k < 4; k++) {
							const int jj = prefix + k - base;
							const float v = __int_as_float(__builtin_amdgcn_ds_bpermute(min(max(jj, 0), 63) << 2, __float_as_int(lt)));
							if (RESV) {
								if (k < cnt && jj >= 0 && jj < 64 && v < cur_t && v >= 0.f) {                       // TriangleMesh.cpp:1399-1412
									st.count++;
									const float r1 = pcg_uniform(st.rng);
									if ((double)r1 < 1. / (double)st.count) { st.dist = v; win = k; wj = jj; upd = true; }
								}
							} else if (k < cnt && jj >= 0 && jj < 64 && v < cur_t) {
								cur_t = v; win = k; wj = jj; upd = true;
								if (SHADOW && ((double)v < (double)st.dist * 0.999)) decided = true;               // TriangleMesh.cpp:1309
							}
						}
						if (!SHADOW) {
							const float vb = __int_as_float(__builtin_amdgcn_ds_bpermute(wj << 2, __float_as_int(lb)));
							const float vg = __int_as_float(__builtin_amdgcn_ds_bpermute(wj << 2, __float_as_int(lg)));
							if (upd) { wb = vb; wg = vg; }
						}
					}
					if (cnt > 0) {
						per_lane = false;
						if (win >= 0) {
							if (!RESV) st.t = cur_t;
							if (!SHADOW) {
								// (the scene-wide triangle index identifies the mesh: mipt_trace.h, hit_unpack)
								if (MIPT_HIT_WRITE_THROUGH && !RESV) wf.hit[st.id] = make_float4(cur_t, wb, wg, __uint_as_float((unsigned)(first + win)));
								else { st.best = first + win; st.beta = wb; st.gamma = wg; }
							}
						}
						if (SHADOW && decided) { st.best = 1; st.cur = MIPT_NONE; st.sp = 0; }
						else st.cur = pop_next();
					}
				}
			}
			if (per_lane) {
				bool decided = false;
				const int last = first + (count < MIPT_LEAF_MAX_TRIS ? count : mipt_leaf_count_scan((uint32_t)first, sc->fat_leaves, sc->n_fat_leaves));
				for (int i = first; i < last; i++) {
					MIPT_PROF_COUNT(4)
					float lt, lb, lg;
					if (tri_test<SHADOW ? (MIPT_DERIVE_SHADOW != 0) : (MIPT_DERIVE_EXTEND != 0)>(tris + i, mk3(st.o_xy.x, st.o_xy.y, st.oz_iz.x), st.d, lt, lb, lg)) {
						bool accept = lt < st.t && (!RESV || lt >= 0.f);
						int local = 0;                                   // mesh-local triangle index, only needed for accepted hits
						if (accept) {
							const DObject& o = sc->obj[st.obj];
							local = i - (int)o.tri_base;
							if (o.alpha_test) accept = !alpha_rejects(o, local, 1 - lb - lg, lb, lg);
						}
						if (accept && RESV) {
							st.count++;
							const float r1 = pcg_uniform(st.rng);
							if ((double)r1 < 1. / (double)st.count) { st.dist = lt; st.best = i; st.beta = lb; st.gamma = lg; }
						} else if (accept) {
							st.t = lt;
							if (SHADOW) {
								if ((double)lt < (double)st.dist * 0.999) { decided = true; break; }           // TriangleMesh.cpp:1309
							} else if (MIPT_HIT_WRITE_THROUGH && !RESV) {
								wf.hit[st.id] = make_float4(lt, lb, lg, __uint_as_float((unsigned)i));
							} else {
								st.best = i; st.beta = lb; st.gamma = lg;
							}
						}
					}
				}
				if (SHADOW && decided) { st.best = 1; st.cur = MIPT_NONE; st.sp = 0; }
				else st.cur = pop_next();
			}
		}
		MIPT_PROF_CLOCK(c3);
		MIPT_PROF_CYCLES(14, c2, c3)
		// ---- mesh finished: the ray goes on with the objects behind it (next iteration's object loop)
		if (RESV) {
			if (st.cur == MIPT_NONE) {
				st.cur = MIPT_ST_IDLE;
				const unsigned tri = (unsigned)st.best == MIPT_HIT_MISS ? MIPT_HIT_MISS : (unsigned)st.best - sc->obj[st.obj].tri_base;      // the probe's answer: the mesh-local triangle
				wf.hit[st.id] = make_float4(st.dist, st.beta, st.gamma, __uint_as_float(tri));
				wf.rng[st.id] = make_uint2((unsigned)st.rng, (unsigned)(st.rng >> 32));
			}
		} else if (st.cur == MIPT_NONE) {
			st.cur = MIPT_ST_NEED;
			// shadow: a hit with t >= 0.999*dist is not an occluder (Geometry.cpp:736) -> next object
			st.obj = (SHADOW && st.best) ? nobj : st.obj + 1;
		}
	}
}

// MODE 0: closest-hit queue of depth b.  MODE 1: shadow queue of depth b.  MODE 2: shadow queue of depth b, then the
// closest-hit queue of depth b + 1 — both were filled by shade(b) and are independent of each other, so one launch serves
// both and the drain phase of the first queue (few rays left, most lanes idle) is covered by waves already working on the
// second: a pass has nb_bounces + 1 traversal launches instead of 2 nb_bounces.
template <int MODE>
__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? MIPT_EXTEND_WAVES : MIPT_TRAVERSE_WAVES))) k_wf_traverse(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, DWave wf, int b, unsigned n0, int refill_threshold, int inner_min_flags) {
	MIPT_DECLARE_LDS_STACK(stk, wf.spill, MIPT_TRAV_BLOCK);
	unsigned char* leafmap = lds_leafmap_ + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 256;
	auto extend_q = [&](int bb) { TravQueue q; q.list = wf.list[bb & 1]; q.n_ptr = bb == 0 ? nullptr : &wf.counters[MIPT_CNT_PAIR(bb - 1) + 1]; q.n_imm = n0; q.head = &wf.counters[MIPT_CNT_EXT_HEAD(bb)]; q.identity = bb == 0; q.vis = nullptr; q.skip_ghosts = false; q.valid_in_ray = true; return q; };
	auto shadow_q = [&](int bb) { TravQueue q; q.list = wf.list_sh; q.n_ptr = &wf.counters[MIPT_CNT_PAIR(bb)]; q.n_imm = 0; q.head = &wf.counters[MIPT_CNT_SH_HEAD(bb)]; q.identity = false; q.vis = nullptr; q.skip_ghosts = false; q.valid_in_ray = false; return q; };
	if (MODE == 1 || MODE == 2) traverse_queue<true>(sc, nodes, tris, wf, shadow_q(b), refill_threshold, inner_min_flags, stk, leafmap);
	if (MODE == 0) traverse_queue<false>(sc, nodes, tris, wf, extend_q(b), refill_threshold, inner_min_flags, stk, leafmap);
	if (MODE == 2) traverse_queue<false>(sc, nodes, tris, wf, extend_q(b + 1), refill_threshold, inner_min_flags, stk, leafmap);
}

// The same traversal on an explicitly described queue (the contribution-queue pipeline, mipt_queue_wave.h): closest hits
// (SHADOW = false: wf.ray_o / ray_d -> wf.hit) or any hits (SHADOW = true: wf.sh_o / sh_d -> tq.vis).
template <bool SHADOW>
__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(SHADOW ? MIPT_TRAVERSE_WAVES : MIPT_EXTEND_WAVES))) k_q_traverse(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, DWave wf, TravQueue tq, int refill_threshold, int inner_min_flags) {
	MIPT_DECLARE_LDS_STACK(stk, wf.spill, MIPT_TRAV_BLOCK);
	unsigned char* leafmap = lds_leafmap_ + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 256;
	traverse_queue<SHADOW>(sc, nodes, tris, wf, tq, refill_threshold, inner_min_flags, stk, leafmap);
}

// The subsurface probes of one round of the contribution-queue pipeline (mipt_queue_wave.h).
__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(5))) k_q_probe(const DScene* __restrict__ sc, const float4* __restrict__ nodes, const DTriIsect* __restrict__ tris, DWave wf, TravQueue tq, int refill_threshold, int inner_min_flags) {
	MIPT_DECLARE_LDS_STACK(stk, wf.spill, MIPT_TRAV_BLOCK);
	unsigned char* leafmap = lds_leafmap_ + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 256;
	traverse_queue<false, true>(sc, nodes, tris, wf, tq, refill_threshold, inner_min_flags, stk, leafmap);
}
