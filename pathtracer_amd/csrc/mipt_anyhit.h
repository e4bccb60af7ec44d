// mipt_anyhit.h — the any-hit (shadow) stage of the wavefront pipeline as an ORDER-FREE traversal of four-wide nodes with 8-bit boxes.
//
// What the reference computes.  Scene::intersection_shadow (Geometry.cpp:691-744) returns true as soon as some object reports a hit
// with t < 0.999 dist_light; it hands every mesh cur_best_t = 1E99 (min_t is never updated), and TriMesh::intersection_shadow
// (TriangleMesh.cpp:1239-1319) returns at the first accepted triangle with t < 0.999 dist_light.  A triangle with t < 0.999 dist is
// always accepted when it is tested (the running t only ever holds values >= 0.999 dist until then), and the alpha test does not
// depend on the order.  So the result is
//       "is there an occluder (t < 0.999 dist, alpha map permitting) in a leaf the traversal REACHES",
// and a leaf is reached iff every box on the way passes `slab test && t_box < dist_light && t_box < running t`, the running t being
// some accepted t in [0.999 dist, dist).  Boxes nest exactly (a node's box is the union of its triangles' boxes, TriangleMesh.cpp:843-858)
// and (plane - o) * invd is monotone in the plane under round-to-nearest, so a box that passes implies that every box around it passes
// with a smaller-or-equal distance:
//       leaf L is reached  <=>  L's OWN box passes with t_box(L) < dist_light  [and no box on the way had t_box >= the running t, which
//                               needs t_box(L) >= 0.999 dist].
// What this file does instead.  It may therefore visit ANY superset of those leaves in ANY order, as long as an occluder only counts
// when the leaf it was found in passes the reference's own test on the reference's own (float) box:
//   * no near / far ordering, no t_near on the stack (4-byte entries), no running t;
//   * FOUR-WIDE nodes with 8-BIT planes (DQuadNode, 64 B): the boxes of the four GRANDCHILDREN of a binary node (a child that is a leaf
//     takes one slot), each plane stored as origin + q * 2^e, rounded OUTWARDS and checked at build time with the very fma the kernel
//     evaluates: the decoded box contains the float box, so (same monotonicity) it passes whenever the float box does.  One node is
//     four 16-byte loads for four boxes where the ordered kernel needs four for two — and the traversal kernels run at the rate the
//     CU issues vector-memory instructions (the exact-box form of this kernel, 7 loads per 4 boxes, halved the
//     dependent rounds per ray, cut vector instructions by 21 %, scalar ones by 30 %, L2 requests by 32 % — and ran 3 % faster,
//     as its 3.5 % fewer vector-memory instructions predict; profiles/r5_b_*).  tests/tools/anyhit_study.py: 13.5 wide steps per ray
//     on configs[2] against 25.7 binary ones; the 8-bit planes cost 1.4 % more steps than exact ones;
//   * when a leaf's triangles yield an occluder, the leaf's float box (leaf_box[first triangle], 32 B) is fetched and tested exactly as
//     TriangleMesh.cpp:1278 tests it.  Fails: the reference never tests this leaf — the occluder does not count, the walk goes on.
//     Passes with t_box < 0.998f dist: no box on the way can have been skipped for the running t — the reference reaches the leaf (or
//     returned true before): occluded.  Passes with t_box >= 0.998f dist: the ray's id goes to a replay list that the ORDERED kernel
//     (traverse_queue<true>, mipt_persistent.h) works off in the reference's order.
// "No occluder found" is exact as it stands: every leaf the reference reaches was visited.  Rays with an infinite inverse-direction
// component (the packed slab test is not valid for them, mipt_trace.h) go to the replay list as well.  On the bench scenes the replay
// list stays empty.
#pragma once

struct DQuadNode {             // 64 B, 64-B aligned: four 16-byte words
	float origin[3];           // word 0: the planes' origin (the minimum corner of the four boxes) ...
	uint32_t exps;             //         ... and the biased exponent byte of each axis' power-of-two step (x | y << 8 | z << 16)
	uint32_t lohi[3][2];       // word 1, first half of word 2: per axis the four slots' lower planes (byte k = slot k), then their upper planes
	uint32_t _pad[2];
	uint32_t ref[4];           // word 3: inner = index of the slot's own DQuadNode (= its binary inner node); bit 31: leaf (first triangle | count-1 << 26)
};
static_assert(sizeof(DQuadNode) == 64, "one 64-byte line per quad node");
// the decoded plane: origin + q * step, ONE rounding (q * step is exact); the builder checks its bytes with this very function
__host__ __device__ __forceinline__ float quad_plane(float q, float step, float origin) { return __builtin_fmaf(q, step, origin); }

// The slab test of box_test_pairs<false> (mipt_trace.h; BBoxT::intersection_invd, Geometry.h:114-142) on planes that arrive as (near, far)
// pairs per axis — near = the minimum when the ray's inverse direction is >= 0 on that axis, else the maximum — instead of (min, max) pairs
// and three sign flags: the same subtractions and multiplications, without the six selects.
MIPT_DEV bool box_test_sorted(mipt_f2 X, mipt_f2 Y, mipt_f2 Z, mipt_f2 o_xy, mipt_f2 i_xy, mipt_f2 oz_iz, float& t_out) {
	const mipt_f2 rx = X - __builtin_shufflevector(o_xy, o_xy, 0, 0), ry = Y - __builtin_shufflevector(o_xy, o_xy, 1, 1), rz = Z - __builtin_shufflevector(oz_iz, oz_iz, 0, 0);
	const mipt_f2 tx = rx * __builtin_shufflevector(i_xy, i_xy, 0, 0), ty = ry * __builtin_shufflevector(i_xy, i_xy, 1, 1), tz = rz * __builtin_shufflevector(oz_iz, oz_iz, 1, 1);
	const float t_enter = fmaxf(fmaxf(tx.x, ty.x), tz.x);
	const float t_exit = fminf(fminf(tx.y, ty.y), tz.y);
	// (BBoxT::intersection_invd's `!(t_enter > t_exit) & !(t_exit < 0)` and `t_enter < 0 ? 0 : t_enter` in one maximum: the two forms agree unless
	// t_enter is a NaN, which the rays of this kernel cannot produce — an infinite inverse direction goes to the replay list — and where
	// this form passes MORE boxes, which the order-free walk may always do.  Two vector instructions per box less.)
	t_out = fmaxf(t_enter, 0.f);
	return !(t_out > t_exit);
}
// byte K of a word as a float (the compiler selects v_cvt_f32_ubyteK)
template <int K> __device__ __forceinline__ float quad_byte(uint32_t w) { return (float)((w >> (8 * K)) & 255u); }

// The order-free traversal rests on boxes that NEST: every child's box inside its parent's, which is what TriMesh::build_bvh produces (a
// node's box is the union of its triangles' boxes, TriangleMesh.cpp:843-858).  A tree handed in through the C ABI is the caller's: this
// check runs over every inner node at upload; a scene with a box that sticks out of its parent's, holds a NaN or an INFINITE plane keeps the
// ordered kernel (the 8-bit planes of k_quad_nodes are origin + q * 2^e: an infinite plane has no such cover — the exponent search ends
// without containing it and the decoded box would reject rays the float box accepts).
__global__ void k_check_nesting(const DFatNode* __restrict__ fat, size_t n, int* __restrict__ bad) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const DFatNode f = fat[i];
	bool ok = true;
	for (int side = 0; side < 2; side++) {
		const uint32_t cref = side ? f.rref : f.lref;
		const float (*cb)[2] = side ? f.r : f.l;
		for (int a = 0; a < 3; a++) ok = ok && cb[a][0] <= cb[a][1] && fabsf(cb[a][0]) <= 3.402823466e38f && fabsf(cb[a][1]) <= 3.402823466e38f;
		if ((cref & MIPT_LEAF_BIT) || cref >= n) continue;
		const DFatNode c = fat[cref];
		for (int a = 0; a < 3; a++) ok = ok && c.l[a][0] >= cb[a][0] && c.l[a][1] <= cb[a][1] && c.r[a][0] >= cb[a][0] && c.r[a][1] <= cb[a][1];
	}
	if (!ok) *bad = 1;
}

// Which binary inner nodes become quad nodes: the roots, and from a quad node the inner ones among its (up to four) slots — every second
// inner level of the tree, shifted wherever a leaf child shortens a side.  One pass marks the slots of the nodes marked so far; a tree of at
// most MIPT_STACK_DEPTH inner levels needs MIPT_STACK_DEPTH / 2 passes (children follow their parents in the node array, so a pass often
// reaches further: the loop on the host stops when a pass changes nothing).
__global__ void k_quad_mark_pass(const DFatNode* __restrict__ fat, uint32_t* __restrict__ mark, size_t n, int* __restrict__ changed) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || mark[i] != 1u) return;
	mark[i] = 2u;                                                   // expanded
	const DFatNode f = fat[i];
	bool any = false;
	for (int side = 0; side < 2; side++) {
		const uint32_t cref = side ? f.rref : f.lref;
		if ((cref & MIPT_LEAF_BIT) || cref >= n) continue;
		const DFatNode c = fat[cref];
		for (int s2 = 0; s2 < 2; s2++) {
			const uint32_t g = s2 ? c.rref : c.lref;
			if (!(g & MIPT_LEAF_BIT) && g < n && mark[g] == 0u) { mark[g] = 1u; any = true; }
		}
	}
	if (any) *changed = 1;
}
__global__ void k_quad_mark_flags(uint32_t* __restrict__ mark, size_t n) {      // marks -> 0 / 1 for the scan that numbers the quad nodes
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) mark[i] = mark[i] ? 1u : 0u;
}
// The quad node of every marked fat node, at position index[i] of a dense array (the order of the fat nodes, i.e. the reference's depth-first
// order, is kept: two quad nodes share a 128-byte line), and the float box of every leaf at leaf_box[8 * first triangle] as (min, max) pairs
// per axis.  is_quad[i] / index[i]: the mark of fat node i and the exclusive scan of the marks.
__global__ void k_quad_nodes(const DFatNode* __restrict__ fat, const uint32_t* __restrict__ is_quad, const uint32_t* __restrict__ index, DQuadNode* __restrict__ quad, float* __restrict__ leaf_box, size_t n) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || !is_quad[i]) return;
	const DFatNode f = fat[i];
	float box[4][3][2];
	uint32_t ref[4];
	int ns = 0;
	auto put = [&](const float (*b)[2], uint32_t r) {
		for (int a = 0; a < 3; a++) { box[ns][a][0] = b[a][0]; box[ns][a][1] = b[a][1]; }
		ref[ns] = (r & MIPT_LEAF_BIT) ? r : index[r]; ns++;
		if (r & MIPT_LEAF_BIT) { float* lb = leaf_box + 8 * (size_t)(r & MIPT_LEAF_FIRST_MASK); for (int a = 0; a < 3; a++) { lb[2 * a] = b[a][0]; lb[2 * a + 1] = b[a][1]; } lb[6] = lb[7] = 0.f; }
	};
	for (int side = 0; side < 2; side++) {
		const uint32_t cref = side ? f.rref : f.lref;
		if ((cref & MIPT_LEAF_BIT) || cref >= n) put(side ? f.r : f.l, cref);              // (cref >= n cannot happen in a checked tree)
		else { const DFatNode c = fat[cref]; put(c.l, c.lref); put(c.r, c.rref); }
	}
	DQuadNode w;
	w.exps = 0; w._pad[0] = w._pad[1] = 0;
	for (int a = 0; a < 3; a++) {
		float lo = box[0][a][0], hi = box[0][a][1];
		for (int k = 1; k < ns; k++) { lo = fminf(lo, box[k][a][0]); hi = fmaxf(hi, box[k][a][1]); }
		// the smallest power-of-two step whose 255th multiple reaches the upper end (boxes are finite: checked when the tree was made)
		int e = 1;                                                        // biased exponent byte: step = 2^(e - 127); at least 2^-126
		while (e < 254 && !(quad_plane(255.f, __uint_as_float((uint32_t)e << 23), lo) >= hi)) e++;
		const float step = __uint_as_float((uint32_t)e << 23);
		w.origin[a] = lo; w.exps |= (uint32_t)e << (8 * a);
		uint32_t lows = 0, highs = 0;
		for (int k = 0; k < 4; k++) {
			int ql = 255, qh = 0;                                             // an unused slot: lower plane above the upper one, never passes
			if (k < ns) {
				ql = (int)floorf((box[k][a][0] - lo) / step); ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
				while (ql > 0 && quad_plane((float)ql, step, lo) > box[k][a][0]) ql--;          // (q = 0 decodes to lo itself, <= every slot's minimum)
				qh = (int)ceilf((box[k][a][1] - lo) / step); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
				while (qh < 255 && quad_plane((float)qh, step, lo) < box[k][a][1]) qh++;        // (q = 255 decodes to >= hi by the choice of e)
			}
			lows |= (uint32_t)ql << (8 * k); highs |= (uint32_t)qh << (8 * k);
		}
		w.lohi[a][0] = lows; w.lohi[a][1] = highs;
	}
	for (int k = 0; k < 4; k++) w.ref[k] = k < ns ? ref[k] : MIPT_LEAF_BIT;
	quad[index[i]] = w;
}
// leaf_box before k_quad_nodes: every entry the whole space (a mesh whose ROOT is a leaf has no node that would write its box; its root
// box is tested by the object loop, and the entry of a triangle that does not start a leaf is never read)
__global__ void k_leaf_box_init(float* __restrict__ leaf_box, size_t ntri) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= 8 * ntri) return;
	const float inf = __int_as_float(0x7f800000);
	leaf_box[i] = (i & 7) >= 6 ? 0.f : ((i & 1) ? inf : -inf);
}

#ifndef MIPT_ANYHIT_WAVES
#define MIPT_ANYHIT_WAVES 7            // 72 registers, no scratch.  8 waves (64 registers: 16 spilled values around the leaf phase and the object loop) is 8 % slower on configs[2]
#endif                                 // (336 against 311 ms per step); the exact-box form of the kernel fitted 64 registers without scratch and gained nothing from the 8th wave either (371 / 366)
#ifndef MIPT_ANY_LDS_STACK
#define MIPT_ANY_LDS_STACK 16          // 4-byte entries in LDS per lane: 16 KB per block of 256 + 4 KB of ray slots + 1 KB of leaf maps, 7 blocks per CU = 147 of 160 KB
#endif
// (the spill columns are the ones of the ordered kernels, read as 4-byte entries: twice as many)
#define MIPT_ANY_SPILL_STACK (2 * MIPT_SPILL_STACK)
// Capacity: a quad step pushes at most 3 entries and a tree of MIPT_STACK_DEPTH inner levels has MIPT_STACK_DEPTH / 2 quad levels on any
// root-to-leaf path; the 4-byte columns overlay the uint2 columns of the ordered kernels, which the host sizes for n_cus * 8 blocks of this size.
static_assert(MIPT_ANY_LDS_STACK + MIPT_ANY_SPILL_STACK >= 3 * (MIPT_STACK_DEPTH / 2), "LdsStack4 cannot hold the pending slots of the deepest tree the upload accepts");
static_assert(MIPT_ANY_SPILL_STACK * sizeof(unsigned) <= MIPT_SPILL_STACK * sizeof(uint2), "the any-hit kernel's spill columns must fit the buffer sized for the ordered kernels");
typedef __attribute__((address_space(3))) unsigned lds_uint1;
typedef __attribute__((address_space(1))) unsigned glb_uint1;
// Entry sp of a lane: LDS word (sp * block + thread) for sp < MIPT_ANY_LDS_STACK, else word ((sp - MIPT_ANY_LDS_STACK) * grid threads + global
// thread) of the spill columns.  Nothing per lane is kept in registers: the wave's bases are scalar and the lane index is recomputed with
// v_mbcnt at every access (at 64 registers the per-lane addresses were the first values the compiler spilled — a scratch reload with a
// wait for ALL outstanding loads in front of every push).
struct LdsStack4 {
	lds_uint1* wave_base;         // &lds[first thread of the wave]            (wave-uniform)
	glb_uint1* spill_wave;        // &spill[first global thread of the wave]   (wave-uniform)
	unsigned spill_stride;        // total threads of the grid
	MIPT_DEV void push(int sp, uint32_t r) {
		const unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		if (sp < MIPT_ANY_LDS_STACK) wave_base[(unsigned)sp * MIPT_TRAV_BLOCK + l] = r;
		else spill_wave[(unsigned)(sp - MIPT_ANY_LDS_STACK) * spill_stride + l] = r;
	}
	MIPT_DEV uint32_t pop(int sp) const {
		const unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		if (sp < MIPT_ANY_LDS_STACK) return wave_base[(unsigned)sp * MIPT_TRAV_BLOCK + l];
		return spill_wave[(unsigned)(sp - MIPT_ANY_LDS_STACK) * spill_stride + l];
	}
};

// Which shadow rays a launch works on, where the answers go and where the rays it may not decide go.
struct AnyQueue {
	const unsigned* list;        // path ids
	const unsigned* n_ptr;       // number of entries (on the device)
	unsigned* head;              // shared chunk counter
	float* vis;                  // nullptr: add the pending direct term to the path's colour when visible; else 1.f / 0.f to vis[id]
	bool skip_ghosts;
	unsigned* replay_list;       // ids the ordered kernel decides afterwards
	unsigned* replay_n;
	unsigned long long* replay_total;   // running total of a render (diagnostics: mipt_debug_anyhit_replayed)
};

// flag bits of inner_min_flags beyond those of traverse_queue: bit 24 = every leaf counts as "reached near the ray's far end"
// (test hook: every occluded ray goes through the replay list)
// What the leaf tests need of a ray and the node steps do not — its direction in the mesh's frame and the occlusion bound — waits in
// LDS (16 bytes per lane, `rayslot` = the wave's 64 slots) instead of in four registers: a test lane reads its OWNER's slot with one
// ds_read_b128 where it would gather four registers with ds_bpermute.
// (the lane index where it is only needed on rare paths: recomputed on the spot and opaque to the optimiser, which otherwise hoists the
// derived LDS address out of the kernel's loop into a register that it then spills)
__device__ __forceinline__ unsigned lane_id_now() { unsigned l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; }
typedef float lds_f4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lds_f4_ lds_float4v;
template <bool DERIVE>
__device__ __forceinline__ void anyhit_queue(const DScene* __restrict__ sc, const float4* __restrict__ quad, const float4* __restrict__ leaf_box, const DTriIsect* __restrict__ tris, const DWave& wf,
                                             const AnyQueue tq, int refill_threshold, int inner_min_flags, LdsStack4& stk, unsigned char* leafmap, lds_float4v* rayslot) {
	const int inner_min = inner_min_flags & 0xffff;
	const bool force_replay = (inner_min_flags >> 16) & 1;      // test hook `literal_slab`: every ray is decided by the ordered kernel's literal chain
	const unsigned lane_limit = ((inner_min_flags >> 17) & 127) ? ((inner_min_flags >> 17) & 127) : 64u;
	const bool flag_all = (inner_min_flags >> 24) & 1;
	const unsigned n = *tq.n_ptr;
	unsigned* head = tq.head;
	const unsigned* __restrict__ list = tq.list;
	const unsigned nwaves = gridDim.x * (MIPT_TRAV_BLOCK / 64), wave_id = blockIdx.x * (MIPT_TRAV_BLOCK / 64) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	bool first_pull = true;
	const unsigned pull_chunk = (unsigned)__builtin_amdgcn_readfirstlane((int)max(64u, min(MIPT_PULL_CHUNK, (n / (gridDim.x * (MIPT_TRAV_BLOCK / 64) * MIPT_PULL_DIV)) & ~63u)));
	const unsigned lane = lane_id();
	auto below_count = [](unsigned long long m) -> unsigned { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };

	// lane state (cur: a wide node, a leaf (bit 31) or MIPT_ST_NEED / MIPT_ST_IDLE / MIPT_NONE as in traverse_queue)
	mipt_f2 o_xy = {0.f, 0.f}, i_xy = {0.f, 0.f}, oz_iz = {0.f, 0.f};
	float dist = 0.f;                      // (occ_below, in the ray slot: the smallest float F with (double)F >= (double)dist * 0.999, so that  t < F  <=>  (double)t < dist * 0.999  (TriangleMesh.cpp:1309)
	uint32_t cur = MIPT_ST_IDLE;
	int sp = 0, obj = 0;
	unsigned id = 0;
	unsigned chunk_next = 0, chunk_end = 0;
	bool drained = false;
	const int nobj = sc->nobj, first_mesh = sc->first_mesh;
	const bool any_alpha = sc->any_alpha != 0;
	const float inf = __int_as_float(0x7f800000);
	auto to_replay = [&]() { const unsigned k = atomicAdd(tq.replay_n, 1u); tq.replay_list[k] = id; atomicAdd(tq.replay_total, 1ull); };

	for (;;) {
		// ---- refill idle lanes from the queue (as traverse_queue)
		const unsigned long long idle = __ballot(cur == MIPT_ST_IDLE && lane < lane_limit);
		const int nidle = __popcll(idle);
		if (!drained && nidle >= refill_threshold) {
			if (chunk_next >= chunk_end) {
				unsigned base = 0;
				if (first_pull) { base = wave_id * pull_chunk; first_pull = false; }
				else if (lane == 0) base = atomicAdd(head, pull_chunk) + nwaves * pull_chunk;
				base = __builtin_amdgcn_readfirstlane(base);
				if (base >= n) { drained = true; chunk_next = chunk_end = 0; }
				else { chunk_next = base; chunk_end = min(base + pull_chunk, n); }
			}
			const unsigned take = min((unsigned)nidle, chunk_end - chunk_next);
			const unsigned prefix = below_count(idle);
			if (cur == MIPT_ST_IDLE && lane < lane_limit && prefix < take) {
				id = list[chunk_next + prefix]; obj = first_mesh; cur = MIPT_ST_NEED;
			}
			chunk_next += take;
		}
		// ---- object loop (wave-uniform index): a ray stops at the first mesh whose root box it enters
		if (__ballot(cur == MIPT_ST_NEED) != 0) {
			f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
			float occ_below = 0.f;
			if (cur == MIPT_ST_NEED && obj < nobj) {
				const float4 o4 = wf.sh_o[id], d4 = wf.sh_d[id];
				ro = mk3(o4.x, o4.y, o4.z); rd = mk3(d4.x, d4.y, d4.z);
				dist = o4.w;
				const double D = (double)dist * 0.999;
				float F = (float)D;
				if ((double)F < D) F = __uint_as_float(F >= 0.f ? __float_as_uint(F) + 1u : __float_as_uint(F) - 1u);
				occ_below = F;
			}
			for (int i = first_mesh; i < nobj; i++) {
				if (cur == MIPT_ST_NEED && obj == i) {
					const DObject& o = sc->obj[i];
					obj = i + 1;
					if (o.type != 0 || (tq.skip_ghosts && o.ghost)) continue;           // spheres / planes were tested when the request was made; ghosts: Geometry.cpp:722
					const f3 d = xf_dir(o.inv, rd), org = xf_point(o.inv, ro);
					const f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
					float t_root;
					bool enter = box_test<false>(ld3(o.root_min), ld3(o.root_max), org, invd, invd.x >= 0, invd.y >= 0, invd.z >= 0, t_root);
					if (enter && t_root > dist) enter = false;                            // TriangleMesh.cpp:1257 (cur_best_t is 1E99)
					if (!enter) continue;
					if (force_replay || fabsf(invd.x) == inf || fabsf(invd.y) == inf || fabsf(invd.z) == inf) { to_replay(); cur = MIPT_ST_IDLE; continue; }
					o_xy = (mipt_f2){org.x, org.y}; i_xy = (mipt_f2){invd.x, invd.y}; oz_iz = (mipt_f2){org.z, invd.z};
					rayslot[lane_id_now()] = (lds_f4_){d.x, d.y, d.z, occ_below};
					cur = o.quad_root; sp = 0; obj = i;
				}
			}
			if (cur == MIPT_ST_NEED) {                                        // no object left: the light sample is visible
				if (tq.vis) tq.vis[id] = 1.f;
				else {
					const float4 c = wf.out.col[id], pc = wf.sh_c[id];
					wf.out.col[id] = make_float4(c.x + pc.x, c.y + pc.y, c.z + pc.z, 0.f);    // Raytracer.cpp:566
				}
				cur = MIPT_ST_IDLE;
			}
		}
		{
			const int nalive = __popcll(__ballot(cur - MIPT_ST_NEED >= 2u));
			if (nalive == 0) { if (drained) break; else continue; }
			if (!drained && (int)lane_limit - nalive >= refill_threshold) continue;
		}
		// ---- wide-node phase: every live lane descends until it holds a leaf or runs out of nodes
		{
			const bool sx = i_xy.x >= 0, sy = i_xy.y >= 0, sz = oz_iz.y >= 0;
			for (;;) {
				const bool inner = cur < MIPT_ST_NEED;
				const unsigned long long mi = __ballot(inner);
				if (mi == 0) break;
				if (__popcll(mi) < inner_min && __ballot(cur >= MIPT_NONE) != 0) break;
				if (!inner) continue;
				const float4* q = quad + 4 * (size_t)cur;
				const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
				const uint32_t ex = __float_as_uint(q0.w);
				const float stx = __uint_as_float((ex & 255u) << 23), sty = __uint_as_float(((ex >> 8) & 255u) << 23), stz = __uint_as_float(((ex >> 16) & 255u) << 23);
				// the four slots' near and far planes per axis, chosen by the ray's sign on the PACKED words (two selects per axis for all four
				// slots; choosing after the decode, as box_test_pairs does, is 24 selects per step — and the kernel's vector pipe is as busy as its
				// vector-memory path).  The same operations on the same operands as the slab test of mipt_trace.h, in the same order.
				const uint32_t lx = __float_as_uint(q1.x), hx = __float_as_uint(q1.y), ly = __float_as_uint(q1.z), hy = __float_as_uint(q1.w), lz = __float_as_uint(q2.x), hz = __float_as_uint(q2.y);
				const uint32_t nwx = sx ? lx : hx, fwx = sx ? hx : lx, nwy = sy ? ly : hy, fwy = sy ? hy : ly, nwz = sz ? lz : hz, fwz = sz ? hz : lz;
				float t0, t1, t2, t3;
#define MIPT_QUAD_SLOT(K, T) box_test_sorted((mipt_f2){quad_plane(quad_byte<K>(nwx), stx, q0.x), quad_plane(quad_byte<K>(fwx), stx, q0.x)}, \
                                             (mipt_f2){quad_plane(quad_byte<K>(nwy), sty, q0.y), quad_plane(quad_byte<K>(fwy), sty, q0.y)}, \
                                             (mipt_f2){quad_plane(quad_byte<K>(nwz), stz, q0.z), quad_plane(quad_byte<K>(fwz), stz, q0.z)}, o_xy, i_xy, oz_iz, T)
				bool p0 = MIPT_QUAD_SLOT(0, t0), p1 = MIPT_QUAD_SLOT(1, t1), p2 = MIPT_QUAD_SLOT(2, t2), p3 = MIPT_QUAD_SLOT(3, t3);
#undef MIPT_QUAD_SLOT
				p0 = p0 && (t0 < dist); p1 = p1 && (t1 < dist); p2 = p2 && (t2 < dist); p3 = p3 && (t3 < dist);      // TriangleMesh.cpp:1278-1279 without `< t`, on boxes that contain the reference's
				const uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y), r2 = __float_as_uint(q3.z), r3 = __float_as_uint(q3.w);
				// the first passing slot is taken, the others wait on the stack (lower slots on top)
				if (p3 && (p0 || p1 || p2)) { stk.push(sp, r3); sp++; }
				if (p2 && (p0 || p1)) { stk.push(sp, r2); sp++; }
				if (p1 && p0) { stk.push(sp, r1); sp++; }
				if (p0) cur = r0;
				else if (p1) cur = r1;
				else if (p2) cur = r2;
				else if (p3) cur = r3;
				else if (sp > 0) { --sp; cur = stk.pop(sp); }
				else cur = MIPT_NONE;
			}
		}
		// ---- leaf phase: the (ray, triangle) tests of all lanes that hold a leaf, packed densely over the wave (as traverse_queue);
		//      an owner only needs to know whether ANY of its tests found an occluder: one ballot instead of the walk in leaf order
		{
			const bool leaf = (int)cur < 0;
			const int first = (int)(cur & MIPT_LEAF_FIRST_MASK);
			const int count = leaf ? (int)((cur >> 26) & 31u) + 1 : 0;
			bool per_lane = leaf, occluded = false;
			if (!any_alpha) {
				const int cnt = count <= 4 ? count : 0;
				const unsigned long long b1 = __ballot(cnt >= 1), b2 = __ballot(cnt >= 2), b3 = __ballot(cnt >= 3), b4 = __ballot(cnt >= 4);
				if (b1 != 0) {
					unsigned pfx = 0;
					for (const unsigned long long m : {b1, b2, b3, b4}) pfx = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, pfx));
					const int prefix = (int)pfx;
					const int total = __popcll(b1) + __popcll(b2) + __popcll(b3) + __popcll(b4);
					for (int k = 0; k < 4; k++) if (k < cnt) leafmap[prefix + k] = (unsigned char)(lane | ((unsigned)k << 6));
					__builtin_amdgcn_wave_barrier();
					for (int base = 0; base < total; base += 64) {
						const int j = base + (int)lane;
						const unsigned m = j < total ? (unsigned)leafmap[j] : 0u;
						const int src = (int)((m & 63u) << 2), slot = (int)(m >> 6);
						const int f = __builtin_amdgcn_ds_bpermute(src, first);
						f3 ro, rd;
						ro.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o_xy.x)));
						ro.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o_xy.y)));
						ro.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(oz_iz.x)));
						const lds_f4_ rs = rayslot[m & 63u];
						rd = mk3(rs.x, rs.y, rs.z);
						const float below = rs.w;
						bool occ = false;
						if (j < total) {
							float a, bb, gg;
							if (tri_test<DERIVE>(tris + f + slot, ro, rd, a, bb, gg)) occ = a < below;
						}
						const unsigned long long om = __ballot(occ);
						// this lane's tests are entries prefix .. prefix + cnt - 1 of the packed list
						const int lo = prefix - base;
						if (cnt > 0 && lo < 64 && lo + cnt > 0) {
							const unsigned long long mine = lo >= 0 ? (((1ull << cnt) - 1ull) << lo) : (((1ull << cnt) - 1ull) >> (-lo));
							if (om & mine) occluded = true;
						}
					}
					if (cnt > 0) per_lane = false;
				}
			}
			if (per_lane) {
				const lds_f4_ rs = rayslot[lane_id_now()];
				const int last = first + (count < MIPT_LEAF_MAX_TRIS ? count : mipt_leaf_count_scan((uint32_t)first, sc->fat_leaves, sc->n_fat_leaves));
				for (int i = first; i < last; i++) {
					float lt, lb, lg;
					if (tri_test<DERIVE>(tris + i, mk3(o_xy.x, o_xy.y, oz_iz.x), mk3(rs.x, rs.y, rs.z), lt, lb, lg) && lt < rs.w) {
						const DObject& o = sc->obj[obj];
						if (o.alpha_test && alpha_rejects(o, i - (int)o.tri_base, 1 - lb - lg, lb, lg)) continue;      // TriangleMesh.cpp:1300-1307
						occluded = true;
						break;
					}
				}
			}
			if (leaf) {
				if (occluded) {
					// the leaf's own float box, tested as the reference tests it on its way down (TriangleMesh.cpp:1278-1279)
					const float4* lb = leaf_box + 2 * (size_t)first;
					const float4 b0 = lb[0], b1 = lb[1];
					float tb;
					const bool reached = box_test_pairs<false>((mipt_f2){b0.x, b0.y}, (mipt_f2){b0.z, b0.w}, (mipt_f2){b1.x, b1.y}, o_xy, i_xy, oz_iz, i_xy.x >= 0, i_xy.y >= 0, oz_iz.y >= 0, tb) && tb < dist;
					const bool far_end = flag_all || tb >= 0.998f * dist;             // (decided here: tb does not live on through the branches below)
					if (!reached) occluded = false;                                   // the reference never looks into this leaf
					else if (far_end) to_replay();                                     // a box on the way may have been skipped for the running t: the ordered kernel decides
					else if (tq.vis) tq.vis[id] = 0.f;                                 // (an occluded ray adds nothing to its path's colour)
				}
				if (occluded) { cur = MIPT_ST_IDLE; sp = 0; }
				else if (sp > 0) { --sp; cur = stk.pop(sp); }
				else cur = MIPT_NONE;
			}
		}
		// ---- mesh finished without an occluder: on to the objects behind it
		if (cur == MIPT_NONE) { cur = MIPT_ST_NEED; obj = obj + 1; }
	}
}

// The shadow queue of depth b of the wavefront pipeline (the rays shade(b) asked for).
__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(MIPT_ANYHIT_WAVES))) k_wf_anyhit(const DScene* __restrict__ sc, const float4* __restrict__ quad, const float4* __restrict__ leaf_box, const DTriIsect* __restrict__ tris, DWave wf, int b,
                                                                                                                   unsigned* replay_list, unsigned long long* replay_total, int refill_threshold, int inner_min_flags) {
	__shared__ unsigned lds_stack4_[MIPT_ANY_LDS_STACK * MIPT_TRAV_BLOCK];
	__shared__ unsigned char lds_leafmap4_[4 * MIPT_TRAV_BLOCK];
	__shared__ float4 lds_rayslot4_[MIPT_TRAV_BLOCK];
	LdsStack4 stk;
	const unsigned wave_in_block = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	stk.wave_base = (lds_uint1*)lds_stack4_ + wave_in_block * 64u;
	stk.spill_wave = (glb_uint1*)wf.spill + (size_t)blockIdx.x * MIPT_TRAV_BLOCK + wave_in_block * 64u; stk.spill_stride = gridDim.x * MIPT_TRAV_BLOCK;
	unsigned char* leafmap = lds_leafmap4_ + wave_in_block * 256;
	AnyQueue q;
	q.list = wf.list_sh; q.n_ptr = &wf.counters[MIPT_CNT_PAIR(b)]; q.head = &wf.counters[MIPT_CNT_SH_HEAD(b)]; q.vis = nullptr; q.skip_ghosts = false;
	q.replay_list = replay_list; q.replay_n = &wf.counters[MIPT_CNT_REPLAY(b)]; q.replay_total = replay_total;
	anyhit_queue<MIPT_DERIVE_SHADOW != 0>(sc, quad, leaf_box, tris, wf, q, refill_threshold, inner_min_flags, stk, leafmap, (lds_float4v*)lds_rayslot4_ + wave_in_block * 64u);
}

// The same stage on an explicitly described queue (the contribution-queue pipeline, mipt_queue_wave.h).
__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(MIPT_ANYHIT_WAVES))) k_q_anyhit(const DScene* __restrict__ sc, const float4* __restrict__ quad, const float4* __restrict__ leaf_box, const DTriIsect* __restrict__ tris, DWave wf,
                                                                                                                  AnyQueue q, int refill_threshold, int inner_min_flags) {
	__shared__ unsigned lds_stack4_[MIPT_ANY_LDS_STACK * MIPT_TRAV_BLOCK];
	__shared__ unsigned char lds_leafmap4_[4 * MIPT_TRAV_BLOCK];
	__shared__ float4 lds_rayslot4_[MIPT_TRAV_BLOCK];
	LdsStack4 stk;
	const unsigned wave_in_block = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	stk.wave_base = (lds_uint1*)lds_stack4_ + wave_in_block * 64u;
	stk.spill_wave = (glb_uint1*)wf.spill + (size_t)blockIdx.x * MIPT_TRAV_BLOCK + wave_in_block * 64u; stk.spill_stride = gridDim.x * MIPT_TRAV_BLOCK;
	unsigned char* leafmap = lds_leafmap4_ + wave_in_block * 256;
	anyhit_queue<MIPT_DERIVE_SHADOW != 0>(sc, quad, leaf_box, tris, wf, q, refill_threshold, inner_min_flags, stk, leafmap, (lds_float4v*)lds_rayslot4_ + wave_in_block * 64u);
}
