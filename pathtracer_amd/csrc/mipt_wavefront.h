// mipt_wavefront.h — pipeline 1: the getColor loop as a wavefront of per-stage kernels over
// queues of path ids (included by mipt.hip after the common kernels).
//
//   generate -> [ extend (closest hit) -> shade (+NEE request, continuation) -> shadow (any hit) ] x depth
//
// Why: the per-path kernel needs ~185 VGPRs (2 waves/SIMD); the traversal alone needs ~75, so the
// split triples the number of rays in flight per CU while the BVH fetches are outstanding, and
// paths that ended stop occupying lanes (the queues are compacted with wave-aggregated appends).
//
// Path state lives in HBM as float4-packed arrays indexed by path id (a path keeps its id for the
// whole pass, only the id lists are compacted); every stage is a persistent kernel whose waves pull
// chunks of ids (the first one statically, later ones from a device-side counter), so the launch
// geometry does not depend on queue sizes that only the device knows.
//
// Float semantics: every stage calls the same device functions as the per-path kernel; the
// additions into a path's colour happen in the reference's order (emission of vertex b, direct
// term of vertex b, then vertex b+1) because shadow(b) precedes shade(b+1) on the stream.
#pragma once

struct DWave {
	float4 *ray_o, *ray_d;      // current ray
	float4* wgt;                // xyz = path weight, w = bits: depth | show_lights << 16
	uint2* rng;                 // pcg32 state
	float4* hit;                // x = t, y = beta, z = gamma, w = bits: packed object / triangle
	float4 *sh_o, *sh_d, *sh_c; // shadow request: origin (w = dist_light), direction, weight*contrib
	unsigned* list[2];          // path ids to extend at even / odd depth
	unsigned* list_sh;          // path ids with a pending shadow ray
	unsigned* list_slow;        // path ids whose vertex the fast shade tier deferred to the general one
	// shade tier 5 (round 6): the measured-BRDF evaluations of a depth as a stage of their own (k_wf_merl_eval).  What an evaluation needs beyond the arrays
	// above waits per path id: hit[] (dead once the vertex has read it) = (wo, the factor of request A), mq_a = (N, the factor of request B), mq_b = (the
	// path weight at the vertex, continuation depth | object << 16); list_mrq = the requests of the depth: path id | kind << 31 (two per path at most)
	float4 *mq_a, *mq_b;
	unsigned* list_mrq;
	uint2* spill;               // traversal-stack overflow columns (persistent kernels)
	unsigned* counters;         // queue sizes and heads, one per 128-byte line (MIPT_CNT_* below)
	DSamples out;
};
// bytes of DWave state per path id: 7 float4 + rng + 4 id lists (the pass is sized with it, mipt.hip render_impl)
#define MIPT_WF_STATE_BYTES (7 * sizeof(float4) + sizeof(uint2) + 4 * sizeof(unsigned))
#define MIPT_WF_MERL_SPLIT_BYTES (2 * sizeof(float4) + 2 * sizeof(unsigned))      // shade tier 5: mq_a, mq_b, two request entries
// Path state is written once and read once per depth, 10 GB per pass: it is accessed with the non-temporal
// (streaming) cache policy so that it does not displace the BVH from L2 / Infinity Cache.
#ifndef MIPT_STREAM_STATE
#define MIPT_STREAM_STATE 1
#endif
typedef float mipt_v4f __attribute__((ext_vector_type(4)));
typedef float mipt_v2f __attribute__((ext_vector_type(2)));
typedef unsigned mipt_v2u __attribute__((ext_vector_type(2)));
#if MIPT_STREAM_STATE
__device__ __forceinline__ float4 wf_ld(const float4* p) { mipt_v4f v = __builtin_nontemporal_load((const mipt_v4f*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 wf_ld(const uint2* p) { mipt_v2u v = __builtin_nontemporal_load((const mipt_v2u*)p); return make_uint2(v.x, v.y); }
__device__ __forceinline__ float2 wf_ld(const float2* p) { mipt_v2f v = __builtin_nontemporal_load((const mipt_v2f*)p); return make_float2(v.x, v.y); }
__device__ __forceinline__ void wf_st(float4* p, float4 a) { mipt_v4f v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, (mipt_v4f*)p); }
__device__ __forceinline__ void wf_st(uint2* p, uint2 a) { mipt_v2u v = {a.x, a.y}; __builtin_nontemporal_store(v, (mipt_v2u*)p); }
__device__ __forceinline__ void wf_st(float2* p, float2 a) { mipt_v2f v = {a.x, a.y}; __builtin_nontemporal_store(v, (mipt_v2f*)p); }
#else
template <class T> __device__ __forceinline__ T wf_ld(const T* p) { return *p; }
template <class T> __device__ __forceinline__ void wf_st(T* p, T a) { *p = a; }
#endif
#define MIPT_WF_MAX_DEPTH 255
// counters layout: [4b..4b+3] per depth (see DWave), then per depth: fast-shade head, then n_slow, then slow head
#define MIPT_WF_CNT_SHADE_HEAD (4 * (MIPT_WF_MAX_DEPTH + 2))
#define MIPT_WF_CNT_NSLOW (MIPT_WF_CNT_SHADE_HEAD + (MIPT_WF_MAX_DEPTH + 2))
#define MIPT_WF_CNT_SLOW_HEAD (MIPT_WF_CNT_NSLOW + (MIPT_WF_MAX_DEPTH + 2))
#define MIPT_WF_CNT_NMRQ (MIPT_WF_CNT_SLOW_HEAD + (MIPT_WF_MAX_DEPTH + 2))      // tier 5: evaluation requests of depth b, then the head of their queue
#define MIPT_WF_CNT_MRQ_HEAD (MIPT_WF_CNT_NMRQ + (MIPT_WF_MAX_DEPTH + 2))
#define MIPT_WF_NCOUNTERS (MIPT_WF_CNT_MRQ_HEAD + (MIPT_WF_MAX_DEPTH + 2))
// every counter sits on its own 128-byte line: atomics on one line are serialised (~11 ns each chip-wide), and a shade
// launch appends to two queues (shadow requests, continuing paths) once per chunk each
#ifndef MIPT_CNT_STRIDE
#define MIPT_CNT_STRIDE 32
#endif
#define MIPT_CNT(i) ((i) * MIPT_CNT_STRIDE)
// per depth b: line 4b holds the PAIR {n_shadow(b), n_extend(b+1)} — the two queues shade(b) appends to — as one 64-bit
// word so that one atomic reserves space in both; line 4b+1 = extend head, 4b+2 = shadow head
#define MIPT_CNT_PAIR(b) MIPT_CNT(4 * (b))
#define MIPT_CNT_EXT_HEAD(b) MIPT_CNT(4 * (b) + 1)
#define MIPT_CNT_SH_HEAD(b) MIPT_CNT(4 * (b) + 2)
// line 4b+3: word 0 = shadow rays of depth b the order-free any-hit kernel left to the ordered one (mipt_anyhit.h), word 8 = that list's head
#define MIPT_CNT_REPLAY(b) MIPT_CNT(4 * (b) + 3)
#define MIPT_N_SHADOW(wf, b) ((wf).counters[MIPT_CNT_PAIR(b)])
#define MIPT_N_EXTEND(wf, b, n0) ((b) == 0 ? (n0) : (wf).counters[MIPT_CNT_PAIR((b) - 1) + 1])
#define MIPT_WF_COUNTERS (MIPT_WF_NCOUNTERS * MIPT_CNT_STRIDE)
// (MIPT_HIT_MISS / MIPT_HIT_ANALYTIC and hit_unpack: mipt_trace.h, beside struct Hit)

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// One same-address atomic costs ~11 ns chip-wide (MI355X_MICROARCH.md "dequeue": one word saturates at
// ~88 ops/us), so queue traffic is batched: a wave pulls MIPT_WF_UNROLL*64 entries per atomic and
// appends the survivors of all its sub-chunks with one atomic per destination queue.
// Round 4: 8 sub-chunks (512 entries) per atomic instead of 4.  A shade launch over 300 M vertices makes 1.2 M chunks at 256 entries,
// and each chunk costs an atomic on the queue head, one on the pair of destination sizes and (fast tier) one on the slow list's size:
// at ~11 ns per same-address atomic that is a floor of ~13 ms per launch whatever the vertices cost.  configs[2] / [1] sat just above
// it (their fast tier is bound by its arithmetic: 372 / 303 ms per step at 4, 8 and 16 sub-chunks; 502 ms at 2), configs[4] sat ON it: its fast
// tier hands two thirds of its vertices on and does little else (generate + shade 849 -> 776 ms per step at 8, 781 at 16;
// profiles/r4_i_shade_stage_experiments.txt).
#ifndef MIPT_WF_UNROLL
#define MIPT_WF_UNROLL 8
#endif
#define MIPT_WF_CHUNK (64u * MIPT_WF_UNROLL)

// Chunks of a queue of n entries for one wave.  The first chunk is static (chunk number = wave number): all waves
// of a launch hitting one counter at start is ~90 us even for an empty queue.  Later chunks come from the shared
// counter, and the atomic for chunk i+1 is issued when chunk i is handed out, so its round trip (a few us, comparable
// to the work of one chunk) overlaps with the work instead of stalling the wave between chunks.
template <unsigned CHUNK>
struct QueuePullerT {
	unsigned next;        // lane 0: base of the chunk to hand out next
	__device__ __forceinline__ void init() { next = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * CHUNK; }
	__device__ __forceinline__ bool pull(unsigned* __restrict__ head, unsigned n, unsigned& base) {
		base = __builtin_amdgcn_readfirstlane(next);
		if (base >= n) return false;
		if (lane_id() == 0) next = atomicAdd(head, CHUNK) + gridDim.x * (blockDim.x >> 6) * CHUNK;
		return true;
	}
};
typedef QueuePullerT<MIPT_WF_CHUNK> QueuePuller;

// wave-aggregated append for the lanes whose bit u of `bits` is set (sub-chunk u of the chunk at
// `base` of the source list; src == nullptr means the identity list): one atomic for all sub-chunks
// (NU: the sub-chunks the caller's chunk has — the stages of the contribution queue keep 4, mipt_queue_wave.h)
template <int NU = MIPT_WF_UNROLL>
__device__ __forceinline__ void queue_push(unsigned* __restrict__ list, unsigned* __restrict__ count, unsigned bits,
                                           const unsigned* __restrict__ src, unsigned src_base) {
	unsigned long long m[NU];
	unsigned total = 0;
#pragma unroll
	for (int u = 0; u < NU; u++) { m[u] = __ballot((bits >> u) & 1u); total += (unsigned)__popcll(m[u]); }
	if (total == 0) return;
	unsigned lane = lane_id();
	unsigned base = 0;
	if (lane == 0) base = atomicAdd(count, total);
	base = __builtin_amdgcn_readfirstlane(base);
	unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
	for (int u = 0; u < NU; u++) {
		if ((bits >> u) & 1u) {
			unsigned idx = src_base + 64u * u + lane;
			list[base + (unsigned)__popcll(m[u] & below)] = src ? src[idx] : idx;
		}
		base += (unsigned)__popcll(m[u]);
	}
}

// The analytic part of Scene::intersection / intersection_shadow is evaluated by the stage that CREATES a ray
// (generate / shade, all lanes busy) instead of by the traversal kernels, where a refill serves few lanes:
//  * closest hit: the objects in front of the first mesh of the object list (light sphere, env sphere, ground plane in
//    every loadScene() scene) are tested in list order with the reference's strict '<' and the running (t, object)
//    is stored with the ray (the .w of its origin / direction float4s); the traversal continues from the first mesh with that bound, exactly as the
//    object loop of Geometry.cpp:600-650 would have arrived there;
//  * shadow ray: occlusion is an OR over the objects (Geometry.cpp:700-741), so every sphere / plane is tested here
//    whatever its position in the list; an occluded request is dropped, the others only traverse the meshes.
// Same device functions (xf_dir / xf_point / sphere_test / plane_test) as the in-kernel object loop: bit-identical.
// t is never a NaN: it starts at +infinity and is only replaced through `tt < t`, which is false for a NaN (degenerate spheres / planes, a NaN camera) —
// depth 0 marks a slot WITHOUT a path by a NaN in this very word (MIPT_WF_DEAD_RAY), so a live path can never be mistaken for one.
MIPT_DEV void analytic_prefix_closest(const DScene* __restrict__ sc, f3 ro, f3 rd, float& t, unsigned& best) {
	t = __int_as_float(0x7f800000); best = MIPT_HIT_MISS;
	const int n = sc->first_mesh;
	for (int i = 0; i < n; i++) {
		const DObject& o = sc->obj[i];
		f3 d = xf_dir(o.inv, rd);
		f3 org = xf_point(o.inv, ro);
		float tt;
		bool hit = (o.type == 1) ? sphere_test(o, org, d, tt) : plane_test(o, org, d, tt);
		if (hit && tt < t) { t = tt; best = MIPT_HIT_ANALYTIC | (unsigned)i; }
	}
}
MIPT_DEV bool analytic_occluded(const DScene* __restrict__ sc, f3 ro, f3 rd, float dist) {
	const int n = sc->nobj;
	bool occ = false;
	for (int i = 0; i < n; i++) {
		const DObject& o = sc->obj[i];
		if (o.type == 0) continue;
		f3 d = xf_dir(o.inv, rd);
		f3 org = xf_point(o.inv, ro);
		float tt;
		bool hit = (o.type == 1) ? sphere_test(o, org, d, tt) : plane_test(o, org, d, tt);
		if (hit && ((double)tt < (double)dist * 0.999)) occ = true;                                       // Geometry.cpp:736-740
	}
	return occ;
}

// true when the shadow ray fails the root-box test of every mesh (the two early returns of TriMesh::intersection_shadow,
// TriangleMesh.cpp:1262-1263, with cur_best_t = inf): no mesh can occlude it
#ifndef MIPT_SHADE_ROOT_TEST
#define MIPT_SHADE_ROOT_TEST 1
#endif
MIPT_DEV bool meshes_missed(const DScene* __restrict__ sc, f3 ro, f3 rd, float dist) {
	const int n = sc->nobj;
	bool missed = true;
	for (int i = sc->first_mesh; i < n; i++) {
		const DObject& o = sc->obj[i];
		if (o.type != 0) continue;
		const f3 d = xf_dir(o.inv, rd);
		const f3 org = xf_point(o.inv, ro);
		const f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
		float t_root;
		if (box_test<false>(ld3(o.root_min), ld3(o.root_max), org, invd, invd.x >= 0, invd.y >= 0, invd.z >= 0, t_root) && !(t_root > dist)) missed = false;
	}
	return missed;
}

// the same for two destination queues whose sizes share one 64-bit word (low: list_a, high: list_b): one atomic
template <int NU = MIPT_WF_UNROLL>
__device__ __forceinline__ void queue_push2(unsigned* __restrict__ list_a, unsigned* __restrict__ list_b, unsigned long long* __restrict__ count2,
                                            unsigned bits_a, unsigned bits_b, const unsigned* __restrict__ src, unsigned src_base) {
	unsigned long long ma[NU], mb[NU];
	unsigned total_a = 0, total_b = 0;
#pragma unroll
	for (int u = 0; u < NU; u++) {
		ma[u] = __ballot((bits_a >> u) & 1u); total_a += (unsigned)__popcll(ma[u]);
		mb[u] = __ballot((bits_b >> u) & 1u); total_b += (unsigned)__popcll(mb[u]);
	}
	if ((total_a | total_b) == 0) return;
	unsigned lane = lane_id();
	unsigned long long base2 = 0;
	if (lane == 0) base2 = atomicAdd(count2, (unsigned long long)total_a | ((unsigned long long)total_b << 32));
	unsigned base_a = __builtin_amdgcn_readfirstlane((unsigned)base2), base_b = __builtin_amdgcn_readfirstlane((unsigned)(base2 >> 32));
	unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
	for (int u = 0; u < NU; u++) {
		unsigned idx = src_base + 64u * u + lane;
		unsigned id = 0;
		if (((bits_a | bits_b) >> u) & 1u) id = src ? src[idx] : idx;
		if ((bits_a >> u) & 1u) list_a[base_a + (unsigned)__popcll(ma[u] & below)] = id;
		if ((bits_b >> u) & 1u) list_b[base_b + (unsigned)__popcll(mb[u] & below)] = id;
		base_a += (unsigned)__popcll(ma[u]); base_b += (unsigned)__popcll(mb[u]);
	}
}

// queue_push2 with `n_extra` more entries for list_b, taken from a wave-private LDS buffer (shade tier 4: the paths whose
// continuation was decided by an evaluation trip): still one atomic
__device__ __forceinline__ void queue_push2x(unsigned* __restrict__ list_a, unsigned* __restrict__ list_b, unsigned long long* __restrict__ count2,
                                             unsigned bits_a, unsigned bits_b, const unsigned* __restrict__ src, unsigned src_base, const unsigned* extra, unsigned n_extra) {
	unsigned long long ma[MIPT_WF_UNROLL], mb[MIPT_WF_UNROLL];
	unsigned total_a = 0, total_b = 0;
#pragma unroll
	for (int u = 0; u < MIPT_WF_UNROLL; u++) {
		ma[u] = __ballot((bits_a >> u) & 1u); total_a += (unsigned)__popcll(ma[u]);
		mb[u] = __ballot((bits_b >> u) & 1u); total_b += (unsigned)__popcll(mb[u]);
	}
	if ((total_a | total_b | n_extra) == 0) return;
	unsigned lane = lane_id();
	unsigned long long base2 = 0;
	if (lane == 0) base2 = atomicAdd(count2, (unsigned long long)total_a | ((unsigned long long)(total_b + n_extra) << 32));
	unsigned base_a = __builtin_amdgcn_readfirstlane((unsigned)base2), base_b = __builtin_amdgcn_readfirstlane((unsigned)(base2 >> 32));
	unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
	for (int u = 0; u < MIPT_WF_UNROLL; u++) {
		unsigned idx = src_base + 64u * u + lane;
		unsigned id = 0;
		if (((bits_a | bits_b) >> u) & 1u) id = src ? src[idx] : idx;
		if ((bits_a >> u) & 1u) list_a[base_a + (unsigned)__popcll(ma[u] & below)] = id;
		if ((bits_b >> u) & 1u) list_b[base_b + (unsigned)__popcll(mb[u] & below)] = id;
		base_a += (unsigned)__popcll(ma[u]); base_b += (unsigned)__popcll(mb[u]);
	}
	for (unsigned k = lane; k < n_extra; k += 64u) list_b[base_b + k] = extra[k];
}

// bit 31 of wgt.w marks a path slot that holds a live path (slots of 8x8 blocks that stick out of
// the image never do); depth 0 of a pass uses the identity list, so generation needs no queue.
#define MIPT_WF_VALID 0x80000000u
#define MIPT_WF_DEAD_RAY 0xffffffffu      // ray_o.w of a path slot that holds no path at depth 0 (a NaN: the analytic prefix never produces one)

__global__ void __launch_bounds__(MIPT_BLOCK) k_wf_generate(const DScene* __restrict__ sc, DRender R, DPass ps, DWave wf, DCounters* __restrict__ cnt, int store_jitter) {
	long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	long long total = (long long)ps.npix_slots * (ps.k1 - ps.k0);
	// Round 5: what a path starts with is not stored — weight (1, 1, 1), depth nb_bounces, show_lights and the engine after path_begin's four
	// draws (pcg_skip4) are recomputed by the stages of depth 0 from the path id; a slot that holds no path (pixel outside the image, a
	// render of depth 0) is marked by a NaN in the .w of its ray origin, which extend(0) and shade(0) load anyway (MIPT_WF_DEAD_RAY).
	bool valid = false, alive = false;
	if (tid < total) {
		int kk = (int)(tid / ps.npix_slots);
		int slot = (int)(tid % ps.npix_slots);
		int blk = slot >> 6, in = slot & 63;
		int i = ps.blocks[2 * blk] + (in >> 3), j = ps.blocks[2 * blk + 1] + (in & 7);
		if (i < R.H && j < R.W) {
			valid = true;
			PathState p; float dx, dy;
			path_begin(R, i, j, ps.k0 + kk, p, dx, dy);
			wf_st(&wf.out.col[tid], make_float4(0.f, 0.f, 0.f, 0.f));
			if (store_jitter) wf_st(&wf.out.dxdy[tid], make_float2(dx, dy));      // (the column-scan splat recomputes it: MIPT_RESOLVE_RECOMPUTE_JITTER)
			alive = path_alive(p);
			if (alive) {
				float t0; unsigned best0;
				analytic_prefix_closest(sc, p.ray.o, p.ray.d, t0, best0);       // rides in the unused .w of the ray's two float4s
				wf_st(&wf.ray_o[tid], make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, t0));
				wf_st(&wf.ray_d[tid], make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, __uint_as_float(best0)));
			}
		}
		if (!alive) wf_st(&wf.ray_o[tid], make_float4(0.f, 0.f, 0.f, __uint_as_float(MIPT_WF_DEAD_RAY)));
	}
	(void)valid; (void)cnt;     // paths are counted on the host (valid pixels x samples)
}

// extend: Scene::intersection without the material (closest object / triangle / t / barycentrics).
// Depth 0 walks the identity list of all n0 path slots and skips the dead ones.
__global__ void __launch_bounds__(MIPT_BLOCK) k_wf_extend(const DScene* __restrict__ sc, DWave wf, int b, unsigned n0) {
	MIPT_DECLARE_STACK(stk);
	const unsigned n = MIPT_N_EXTEND(wf, b, n0);
	unsigned* head = &wf.counters[MIPT_CNT_EXT_HEAD(b)];
	const unsigned* __restrict__ list = wf.list[b & 1];
	unsigned base;
	QueuePuller q; q.init();
	while (q.pull(head, n, base)) {
#pragma unroll 1
		for (int u = 0; u < MIPT_WF_UNROLL; u++) {
			unsigned idx = base + 64u * u + lane_id();
			if (idx >= n) continue;
			unsigned id = b == 0 ? idx : list[idx];
			float4 o = wf_ld(&wf.ray_o[id]), d = wf_ld(&wf.ray_d[id]);
			if (b == 0 && o.w != o.w) continue;                           // no path in this slot (MIPT_WF_DEAD_RAY)
			Ray r; r.o = mk3(o.x, o.y, o.z); r.d = mk3(d.x, d.y, d.z);
			Hit h;
			bool hit = scene_closest(sc, r, h, stk);
			unsigned packed = hit ? hit_pack(sc, h.obj, h.tri) : MIPT_HIT_MISS;
			wf_st(&wf.hit[id], make_float4(h.t, h.beta, h.gamma, __uint_as_float(packed)));
		}
	}
}

// shade: material of the hit, emission, next-event-estimation request, continuation sampling.
// TIER 0: every vertex with the general code.  TIER 1: fast tier over the same queue; vertices it cannot
// handle go to list_slow.  TIER 2: the general code over list_slow for scenes without a measured BRDF, TIER 3: with.
#ifndef MIPT_PERTURB
#define MIPT_PERTURB 0                  // measurement builds only (tools/build_variant.sh): leaves a piece of the shade stage out to price it
#endif
#ifndef MIPT_SHADE_ROLLED
#define MIPT_SHADE_ROLLED 1
#endif
// Round 1 requested the state of sub-chunk u+1 while sub-chunk u was shaded (a software pipeline over the first of the three
// dependent round trips of a vertex).  That holds 18 registers across the whole vertex code: 143 instead of 126 in the fast
// tier, i.e. 3 waves per SIMD instead of 4 — and a fourth wave without a single spilled value hides more latency than the
// pipeline did (C2: generate + shade 482 -> 449 ms per step without the prefetch at 3 waves, -> 438 ms at 4; the measurement
// of round 1 that "4 waves gain nothing" was taken with the prefetch in place and 28 spilled registers).  The general tiers
// fall from 169 / 193 to 154 / 169 registers: 3 waves instead of 2 (C4, MERL tier: 1 096 -> 947 ms; glossy C1: 661 -> 596 ms).
#ifndef MIPT_SHADE_WAVES
#define MIPT_SHADE_WAVES 4
#endif
#ifndef MIPT_SHADE_RECOMPUTE_CAMERA
#define MIPT_SHADE_RECOMPUTE_CAMERA 0   // 1 = depth 0: the shade stage recomputes the camera ray instead of fetching it.  Measured and off: 32 of ~250 bytes per
                                        // vertex less, but path_begin costs the fast tier 14 spilled registers: generate + shade 347 against 331 ms on configs[2]
#endif
#ifndef MIPT_SHADE_PREFETCH
#define MIPT_SHADE_PREFETCH 0
#endif
#ifndef MIPT_SHADE2_WAVES
#define MIPT_SHADE2_WAVES 3
#endif
#ifndef MIPT_SHADE3_WAVES
#define MIPT_SHADE3_WAVES 3             // tiers 0 and 3 (general code with the measured BRDF)
#endif
#ifndef MIPT_SHADE_EARLY_DEFER
#define MIPT_SHADE_EARLY_DEFER 1
#endif
#ifndef MIPT_SHADE4_WAVES
#define MIPT_SHADE4_WAVES 3             // tier 4 (general code, measured-BRDF evaluations batched)
#endif
// TIER 4 (round 4): tier 3 with the table evaluations of the measured BRDF taken OUT of the vertex code.  In tier 3 a vertex calls
// merl_eval twice, each time with the lanes its own branch left (light above the horizon: ~55 %; a continuation that survives
// Raytracer.cpp:593 and is not the path's last: ~35 %), and the evaluation is ~85 % of the tier's vector instructions: the tier ran
// at 33 of 64 lanes.  Here a vertex only FILES its requests (path_vertex_merl_requests, mipt_shade.h) in a per-wave list in LDS — 15
// words each: the direction, the factor, wo, N, the path weight, the path id, kind / object / depth — and whenever 64 have come
// together the wave evaluates them in ONE trip with every lane busy, whichever vertices (of this chunk or an earlier one) they
// belong to.  The lane that evaluates a request also finishes it: A writes weight * contrib where the shadow stage expects it,
// B computes the new path weight, tests path_alive (Raytracer.cpp:240-241) and, if the path lives, writes the weight and files the
// path id for the next depth's queue (appended with the chunk's other continuations, one atomic per chunk as before).  Same
// operations on the same operands as path_vertex: bit-identical.  The one copy of the evaluation is inlined at the top level of the
// sub-chunk loop, where nothing of a vertex is live: no call, no callee-saved registers through scratch.
// A vertex with a request A always goes through the shadow queue (tier 3 adds the light term at once when the shadow ray misses
// the root box of every mesh; the shadow stage then adds the same product to the same colour).
#define MIPT_MERL_RQ 128                // request slots per wave: < 64 carried over + at most 64 filed at a time
#define MIPT_MERL_RQ_WORDS 15
#define MIPT_MERL_ALIVE (64 + 64 * MIPT_WF_UNROLL)   // path ids of continuing paths per wave between two chunk ends: <= 63 carried + one chunk's
#define MIPT_MERL_LDS_WORDS (MIPT_MERL_RQ * MIPT_MERL_RQ_WORDS + MIPT_MERL_ALIVE)
// MIPT_SHADE_GLDS (experiment of round 4, off): the path state of sub-chunk u+1 is requested while sub-chunk u is shaded, as round 1's
// software pipeline did — but by LDS-DMA (global_load_lds: the data lands in LDS, no register is held while it is in flight),
// tiers 1 to 3.  Per wave: 5 x 64 float4 (weight, origin, direction, hit, colour) + 64 uint2.  Bit-identical (GPU suite and 300 fuzz
// scenes), and within the run-to-run spread of the stage on every config (configs[2]: 373 / 373 ms with, 383 / 363 without;
// configs[1] 316 / 308 and 308 / 316; configs[3] 591 / 591; configs[4] 866 / 864): the first of a vertex's dependent round trips
// is not what the stage waits for (profiles/r4_i_shade_stage_experiments.txt).
#ifndef MIPT_SHADE_GLDS
#define MIPT_SHADE_GLDS 0
#endif
#define MIPT_GLDS_WORDS (5 * 256 + 128)
#define MIPT_MRQ_STAGE (2 * 64 * MIPT_WF_UNROLL)       // tier 5: request entries a wave files per chunk at most (staged in LDS: one atomic per chunk)
#define MIPT_SHADE_LDS_BYTES(TIER) ((TIER) == 5 ? (MIPT_BLOCK / 64) * MIPT_MRQ_STAGE * 4 : (TIER) == 4 ? (MIPT_BLOCK / 64) * MIPT_MERL_LDS_WORDS * 4 : ((MIPT_SHADE_GLDS && ((TIER) == 1 || (TIER) == 2 || (TIER) == 3)) ? (MIPT_BLOCK / 64) * MIPT_GLDS_WORDS * 4 : 0))
#define MIPT_SHADE4_LDS_BYTES MIPT_SHADE_LDS_BYTES(4)
extern __shared__ unsigned mipt_shade_lds[];
template <int TIER, bool INITIAL = false>       // INITIAL: the build for depth 0 (b == 0), where a path's state is recomputed instead of fetched
__global__ void __launch_bounds__(MIPT_BLOCK) __attribute__((amdgpu_waves_per_eu(TIER == 1 ? MIPT_SHADE_WAVES : (TIER == 2 || TIER == 5 ? MIPT_SHADE2_WAVES : (TIER == 4 ? MIPT_SHADE4_WAVES : MIPT_SHADE3_WAVES))))) k_wf_shade(const DScene* __restrict__ sc, DRender R, DPass ps, DWave wf, int b, unsigned n0, DCounters* __restrict__ cnt) {
	static_assert((TIER != 4 && TIER != 5) || MIPT_SHADE_ROLLED, "tiers 4 and 5 are written into the rolled form of the sub-chunk loop");
	// (tier 4's code over the WHOLE queue of a depth, no fast tier in front of it, was measured too: configs[4] generate + shade 864 ms
	// against 771 with the fast tier, and the traversal stages 3 % slower on the queues it leaves)
	// (a classification pass in front of the fast tier — hits on measured-BRDF objects straight to list_slow, the fast tier over the rest
	// with all its lanes — was measured as well, git tag r4-classify-experiment: the pass costs 7.3 ms per launch, one atomic per 512
	// entries IS its time, and the fast tier over a third of the vertices still takes 12.3 ms of its 13.2: configs[4] 875 ms against 787)
	constexpr bool SLOW_LIST = TIER >= 2;
	constexpr bool BATCH = TIER == 4;                 // the evaluations in trips of 64 inside this kernel
	constexpr bool SPLIT = TIER == 5;                 // the evaluations by k_wf_merl_eval, launched behind this kernel: a vertex only files its requests
	constexpr bool REQ = BATCH || SPLIT;              // a measured-BRDF vertex files requests (path_vertex_merl_requests) instead of evaluating
	const unsigned n = SLOW_LIST ? wf.counters[MIPT_CNT(MIPT_WF_CNT_NSLOW + b)] : MIPT_N_EXTEND(wf, b, n0);
	unsigned* head = &wf.counters[MIPT_CNT((SLOW_LIST ? MIPT_WF_CNT_SLOW_HEAD : MIPT_WF_CNT_SHADE_HEAD) + b)];
	const unsigned* __restrict__ list = SLOW_LIST ? wf.list_slow : wf.list[b & 1];
	constexpr bool identity = !SLOW_LIST && INITIAL;
	unsigned* __restrict__ next = wf.list[(b + 1) & 1];
	unsigned n_closest = 0, n_shadow = 0;
	unsigned base;
	QueuePuller q; q.init();
	// Software pipeline over the sub-chunks of a chunk: the path ids of all sub-chunks are read first, and the state
	// of sub-chunk u+1 is requested before sub-chunk u is shaded — with 3 waves per SIMD the stage is bound by its
	// three dependent HBM round trips per vertex (state, shading record, texel), this hides the first one.
	// depth 0: weight, flags and engine of a path are what path_begin leaves — recomputed, not fetched (k_wf_generate does not store them)
	constexpr bool initial = INITIAL;
	// (MIPT_SHADE_RECOMPUTE_CAMERA: the camera ray too — path_begin run again from the path id; measured slower, off)
	constexpr bool camera = initial && MIPT_SHADE_RECOMPUTE_CAMERA;
	struct In { float4 w, o, d, hr, col; uint2 rs; };
	auto fetch = [&](unsigned id, bool ok, In& in) {
		if (ok) {
			if (!initial) in.w = wf_ld(&wf.wgt[id]);
			if (!camera) { in.o = wf_ld(&wf.ray_o[id]); in.d = wf_ld(&wf.ray_d[id]); }
			in.hr = wf_ld(&wf.hit[id]);
			if (TIER != 1) in.col = wf_ld(&wf.out.col[id]);          // the fast tier touches the colour only when a vertex adds to it
			if (!initial) in.rs = wf_ld(&wf.rng[id]);
		}
	};
	// tier 4: this wave's request list and its list of continuing path ids (LDS), both wave-uniform counts
	unsigned* const mrq_stage = SPLIT ? mipt_shade_lds + (threadIdx.x >> 6) * MIPT_MRQ_STAGE : nullptr;      // tier 5: this wave's request entries of the chunk
	unsigned n_mrq = 0;
	unsigned* const rq = BATCH ? mipt_shade_lds + (threadIdx.x >> 6) * MIPT_MERL_LDS_WORDS : nullptr;
	unsigned* const alive_buf = BATCH ? rq + MIPT_MERL_RQ * MIPT_MERL_RQ_WORDS : nullptr;
	unsigned rq_count = 0, n_alive = 0;
	for (;;) {
		// tier 4 runs the body once more after the last chunk, without a chunk, to evaluate the requests that are left
		const bool got = q.pull(head, n, base);
		if (!got && (!BATCH || rq_count == 0)) break;
		unsigned cont_bits = 0, cast_bits = 0, slow_bits = 0;
#if MIPT_SHADE_ROLLED
		// The loop over the sub-chunks is NOT unrolled: one copy of the vertex code is ~4 600 instructions (37 KB), four copies
		// do not fit the 64 KB instruction cache two CUs share.  The path id of sub-chunk u+2 and the state of sub-chunk
		// u+1 are requested while sub-chunk u is shaded.
		auto load_id = [&](int u) -> unsigned {
			const unsigned idx = base + 64u * (unsigned)u + lane_id();
			return (u < (int)MIPT_WF_UNROLL && idx < n) ? (identity ? idx : list[idx]) : 0xffffffffu;
		};
		unsigned id_cur = load_id(0), id_nxt = load_id(1), id_nn = 0xffffffffu;
		In cur, nxt;
		cur.w = cur.o = cur.d = cur.hr = cur.col = make_float4(0.f, 0.f, 0.f, 0.f); cur.rs = make_uint2(0u, 0u); nxt = cur;
		if (MIPT_SHADE_PREFETCH) fetch(id_cur, id_cur != 0xffffffffu, cur);
		constexpr bool GLDS = MIPT_SHADE_GLDS && (TIER == 1 || TIER == 2 || TIER == 3);
		typedef __attribute__((address_space(3))) unsigned lds_u32;
		typedef const __attribute__((address_space(1))) void* gptr_t;
		lds_u32* const gl = (lds_u32*)(mipt_shade_lds + (GLDS ? (threadIdx.x >> 6) * MIPT_GLDS_WORDS : 0));
		auto glds_issue = [&](unsigned pid) {
			if (pid != 0xffffffffu) {
				if (!initial) __builtin_amdgcn_global_load_lds((gptr_t)&wf.wgt[pid], gl + 0, 16, 0, 2);
				if (!camera) {
					__builtin_amdgcn_global_load_lds((gptr_t)&wf.ray_o[pid], gl + 256, 16, 0, 2);
					__builtin_amdgcn_global_load_lds((gptr_t)&wf.ray_d[pid], gl + 512, 16, 0, 2);
				}
				__builtin_amdgcn_global_load_lds((gptr_t)&wf.hit[pid], gl + 768, 16, 0, 2);
				if (TIER != 1) __builtin_amdgcn_global_load_lds((gptr_t)&wf.out.col[pid], gl + 1024, 16, 0, 2);
				if (!initial) {
					__builtin_amdgcn_global_load_lds((gptr_t)&wf.rng[pid], gl + 1280, 4, 0, 2);
					__builtin_amdgcn_global_load_lds((gptr_t)((const unsigned*)&wf.rng[pid] + 1), gl + 1344, 4, 0, 2);
				}
			}
		};
		if (GLDS) glds_issue(id_cur);
#pragma unroll 1
		for (int u = 0; u < (BATCH && !got ? 1 : (int)MIPT_WF_UNROLL); u++) {
			const unsigned id = id_cur;
			id_nn = load_id(u + 2);
			// tier 4: what this lane's vertex files (bit 0: request A, bit 1: request B)
			unsigned req_bits = 0, req_meta = 0;
			f3 req_xa = mk3(0, 0, 0), req_xb = mk3(0, 0, 0), req_wo = mk3(0, 0, 0), req_n = mk3(0, 0, 0), req_w = mk3(0, 0, 0);
			float req_sa = 0.f, req_fb = 0.f;
			if (GLDS) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (id_cur != 0xffffffffu) {
					const unsigned l = lane_id();
					typedef __attribute__((address_space(3))) mipt_v4f lds_f4;
					const lds_f4* g4 = (const lds_f4*)gl;
					const mipt_v4f a0 = g4[l], a1 = g4[64 + l], a2 = g4[128 + l], a3 = g4[192 + l];
					if (!initial) cur.w = make_float4(a0.x, a0.y, a0.z, a0.w);
					cur.o = make_float4(a1.x, a1.y, a1.z, a1.w); cur.d = make_float4(a2.x, a2.y, a2.z, a2.w); cur.hr = make_float4(a3.x, a3.y, a3.z, a3.w);
					if (TIER != 1) { const mipt_v4f a4 = g4[256 + l]; cur.col = make_float4(a4.x, a4.y, a4.z, a4.w); }
					if (!initial) cur.rs = make_uint2(gl[1280 + l], gl[1344 + l]);
				}
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				glds_issue(id_nxt);                                                   // lands while this sub-chunk is shaded
			} else if (MIPT_SHADE_PREFETCH) fetch(id_nxt, id_nxt != 0xffffffffu, nxt);       // in flight while this sub-chunk is shaded; moved into `cur` at the end of the body
			else fetch(id_cur, id_cur != 0xffffffffu, cur);
			const In sin = cur;
			do {
			if (id == 0xffffffffu) break;
#else
		unsigned ids[MIPT_WF_UNROLL];
#pragma unroll
		for (int u = 0; u < MIPT_WF_UNROLL; u++) {
			const unsigned idx = base + 64u * u + lane_id();
			ids[u] = idx < n ? (identity ? idx : list[idx]) : 0xffffffffu;
		}
		In cur, nxt;
		cur.w = cur.o = cur.d = cur.hr = cur.col = make_float4(0.f, 0.f, 0.f, 0.f); cur.rs = make_uint2(0u, 0u); nxt = cur;
		fetch(ids[0], ids[0] != 0xffffffffu, cur);
#pragma unroll
		for (int u = 0; u < MIPT_WF_UNROLL; u++) {
			const unsigned id = ids[u];
			if (u + 1 < MIPT_WF_UNROLL) fetch(ids[u + 1], ids[u + 1] != 0xffffffffu, nxt);
			const In sin = cur;
			cur = nxt;
			do {
			if (id == 0xffffffffu) break;
#endif
			const float4 w = sin.w, o = sin.o, d = sin.d, hr = sin.hr, col = sin.col;
			const uint2 rs = sin.rs;
			if (initial && !camera && o.w != o.w) break;                      // no path in this slot (MIPT_WF_DEAD_RAY)
			// pixel of this path (for the per-pixel Cranley-Patterson rotation, and at depth 0 for its engine) and its sample index
			const int kk = (int)(id / (unsigned)ps.npix_slots), slot = (int)(id % (unsigned)ps.npix_slots);
			const int blk = slot >> 6, in = slot & 63;
			const int pi = ps.blocks[2 * blk] + (in >> 3), pj = ps.blocks[2 * blk + 1] + (in & 7);
			if (camera && (pi >= R.H || pj >= R.W || R.nb_bounces <= 0)) break;      // the slots k_wf_generate marks dead: outside the image, or a render of depth 0
			const unsigned fl = initial ? (MIPT_WF_VALID | (unsigned)R.nb_bounces | 0x10000u) : __float_as_uint(w.w);       // path_begin: depth nb_bounces, show_lights
			PathState p;
			p.ray.o = mk3(o.x, o.y, o.z); p.ray.d = mk3(d.x, d.y, d.z);
			p.weight = initial ? mk3(1.f, 1.f, 1.f) : mk3(w.x, w.y, w.z); p.color = mk3(col.x, col.y, col.z);
			p.rng = initial ? pcg_skip4(pcg_seed(((uint64_t)pi * (uint64_t)R.W + (uint64_t)pj) * R.seed_stride + (uint64_t)(ps.k0 + kk)))      // the engine behind path_begin's four draws
			                : ((uint64_t)rs.x | ((uint64_t)rs.y << 32));
			p.depth = (int)(fl & 0xffffu); p.show_lights = (fl & 0x10000u) != 0;
			if (camera) {                                                      // what k_wf_generate computed for this slot, again: ray, weight (1, 1, 1), engine after its four draws, depth, show_lights
				float jx, jy;
				path_begin(R, pi, pj, ps.k0 + kk, p, jx, jy);
				p.color = mk3(col.x, col.y, col.z);
			}
			unsigned packed = __float_as_uint(hr.w);
			Hit h; h.t = hr.x; h.beta = hr.y; h.gamma = hr.z;
			const bool has_inter = hit_unpack(sc, packed, h.obj, h.tri);
			f3 P = mk3(0, 0, 0); Mat m;
			m.shadingN = mk3(0, 1, 0); m.Kd = mk3(0.5f, 0.5f, 0.5f); m.Ks = mk3(0, 0, 0); m.Ne = mk3(100, 100, 100); m.Ke = mk3(0, 0, 0); m.transp = false; m.refr_index = 0;
			if (TIER == 1) {
				MIPT_PROF_COUNT(16)
				if (has_inter && h.obj == 1) MIPT_PROF_COUNT(20)
				if (!has_inter || h.obj == 0) MIPT_PROF_COUNT(22)
				// a hit on an object with a measured BRDF is the general tier's whatever its material says: handed on BEFORE the
				// material is fetched (round 4; until then this tier fetched the material of such a vertex — shading record, group
				// record, the lot — only to find mat.merl set and defer it, and the general tier fetched it again)
				if (MIPT_SHADE_EARLY_DEFER && has_inter && object_has_merl(sc, h.obj)) { MIPT_PROF_COUNT(24) slow_bits |= 1u << u; break; }
			}
			if (has_inter) hit_material(sc, p.ray, h, P, m);
			ShadowRequest sh; f3 wv;
			bool c;
			if (TIER == 1) {
				int r = path_vertex_fast(sc, R, p, has_inter, h, P, m, pi * R.W + pj, ps.k0 + kk, sh, wv);
				if (r == VERTEX_DEFER) { MIPT_PROF_COUNT(24) slow_bits |= 1u << u; break; }
				if (sh.diffuse) MIPT_PROF_COUNT(18)
				c = r == VERTEX_CONTINUE;
			} else if (REQ && has_inter && h.obj != 0 && h.obj != 1 && !(m.miroir & 1) && !m.transp && m.merl != nullptr) {
				// exactly the vertices path_vertex takes through the measured BRDF
				wv = p.weight;
				req_wo = -p.ray.d; req_n = m.shadingN; req_w = p.weight;
				bool ra, rb;
				path_vertex_merl_requests(R, p, P, m, pi * R.W + pj, ps.k0 + kk, sh, ra, req_xa, req_sa, rb, req_xb, req_fb);
				req_bits = (ra ? 1u : 0u) | (rb ? 2u : 0u);
				req_meta = ((unsigned)p.depth << 1) | ((unsigned)h.obj << 16);     // bit 0: the kind; p.depth (15 bits: mipt_render refuses deeper paths on a scene with a
				                                                                    // measured BRDF): already the continuation's; the object: 16 bits (MIPT_MAX_OBJECTS)
				c = false;                                                        // (request B decides; see below)
			} else c = path_vertex<TIER != 2 && !REQ>(sc, R, p, has_inter, h, P, m, pi * R.W + pj, ps.k0 + kk, sh, wv);
			n_closest++;
			// The shadow request of this vertex: an analytic occluder settles it here (never queued); so does a ray that
			// misses the root box of every mesh — TriMesh::intersection_shadow returns before it visits a node
			// (TriangleMesh.cpp:1262-1263), the light sample is visible and its term is added now, in the reference's order
			// (emission :411, then direct light :566).  Everything else goes to the shadow stage.
			bool sh_queue = false;
			if (sh.diffuse && sh.cast) {
				n_shadow++;                                               // counted like the reference counts intersection_shadow calls
#if MIPT_PERTURB == 1
				sh_queue = true;                                          // measurement probe: no analytic / root tests for the shadow request
#else
				if (!analytic_occluded(sc, sh.ray.o, sh.ray.d, sh.dist)) {
					if (REQ && (req_bits & 1u)) sh_queue = true;          // its weight * contrib is written by the lane that evaluates request A
					else if (MIPT_SHADE_ROOT_TEST && meshes_missed(sc, sh.ray.o, sh.ray.d, sh.dist)) p.color = p.color + wv * sh.contrib;
					else sh_queue = true;
				} else req_bits &= ~1u;                                       // nobody needs the value
#endif
			}
			if (TIER == 1) {
				// the vertex ran with colour 0, so p.color is exactly the term it adds (0 + x = x); on a non-emissive surface
				// that term is 0 and colour + 0 = colour: the path's colour is neither read nor written (it is never -0)
				if (p.color.x != 0.f || p.color.y != 0.f || p.color.z != 0.f) {
					const float4 c0 = wf_ld(&wf.out.col[id]);
					wf_st(&wf.out.col[id], make_float4(c0.x + p.color.x, c0.y + p.color.y, c0.z + p.color.z, 0.f));
				}
			} else wf_st(&wf.out.col[id], make_float4(p.color.x, p.color.y, p.color.z, 0.f));
			if (sh_queue) {
				cast_bits |= 1u << u;
				f3 pc = wv * sh.contrib;                              // added by k_wf_shadow if the light sample is visible
				wf_st(&wf.sh_o[id], make_float4(sh.ray.o.x, sh.ray.o.y, sh.ray.o.z, sh.dist));
				wf_st(&wf.sh_d[id], make_float4(sh.ray.d.x, sh.ray.d.y, sh.ray.d.z, 0.f));
				if (!(REQ && (req_bits & 1u))) wf_st(&wf.sh_c[id], make_float4(pc.x, pc.y, pc.z, 0.f));
			}
			c = c && path_alive(p);                                   // Raytracer.cpp:240-241 at the top of the next iteration
			if (c || (REQ && (req_bits & 2u))) {
				float t0; unsigned best0;
#if MIPT_PERTURB == 2
				t0 = __int_as_float(0x7f800000); best0 = MIPT_HIT_MISS;         // measurement probe: no analytic prefix for the continuation ray
#else
				analytic_prefix_closest(sc, p.ray.o, p.ray.d, t0, best0);       // rides in the unused .w of the ray's two float4s
#endif
				wf_st(&wf.ray_o[id], make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, t0));
				wf_st(&wf.ray_d[id], make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, __uint_as_float(best0)));
				// (a vertex with a request B: the weight — and whether the path goes on — is written by the lane that evaluates it)
				if (c) wf_st(&wf.wgt[id], make_float4(p.weight.x, p.weight.y, p.weight.z, __uint_as_float(MIPT_WF_VALID | (unsigned)p.depth | (p.show_lights ? 0x10000u : 0u))));
				wf_st(&wf.rng[id], make_uint2((unsigned)p.rng, (unsigned)(p.rng >> 32)));
			}
			if (c) cont_bits |= 1u << u;
			} while (0);
#if MIPT_SHADE_ROLLED
			if (MIPT_SHADE_PREFETCH) cur = nxt;
			id_cur = id_nxt; id_nxt = id_nn;
#endif
			if (SPLIT) {
				// Tier 5: what the evaluation stage needs of this vertex goes to the per-path arrays, the requests themselves (path id | kind) to the
				// wave's staging list in LDS; the chunk's entries reach the depth's request queue with ONE atomic at the end of the chunk.
				if (req_bits) {
					wf_st(&wf.hit[id], make_float4(req_wo.x, req_wo.y, req_wo.z, req_sa));
					wf_st(&wf.mq_a[id], make_float4(req_n.x, req_n.y, req_n.z, req_fb));
					wf_st(&wf.mq_b[id], make_float4(req_w.x, req_w.y, req_w.z, __uint_as_float(req_meta >> 1)));      // depth | object << 15
				}
				const unsigned long long ma = __ballot(req_bits & 1u), mb = __ballot(req_bits & 2u);
				if (req_bits & 1u) mrq_stage[n_mrq + __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u))] = id;
				n_mrq += (unsigned)__popcll(ma);
				if (req_bits & 2u) mrq_stage[n_mrq + __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u))] = id | 0x80000000u;
				n_mrq += (unsigned)__popcll(mb);
			}
			if (BATCH) {
				// File this sub-chunk's requests (A of all lanes, then B of all lanes), and evaluate the 64 filed last whenever
				// that many have come together.  A request is complete in itself, so the order of evaluation does not matter.
#pragma unroll 1
				for (int kind = 0; kind < 2; kind++) {
					const bool has = (req_bits >> kind) & 1u;
					const unsigned long long hm = __ballot(has);
					if (has) {
						const unsigned e = rq_count + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
						const f3 x = kind ? req_xb : req_xa;
						rq[0 * MIPT_MERL_RQ + e] = __float_as_uint(x.x); rq[1 * MIPT_MERL_RQ + e] = __float_as_uint(x.y); rq[2 * MIPT_MERL_RQ + e] = __float_as_uint(x.z);
						rq[3 * MIPT_MERL_RQ + e] = __float_as_uint(kind ? req_fb : req_sa);
						rq[4 * MIPT_MERL_RQ + e] = __float_as_uint(req_wo.x); rq[5 * MIPT_MERL_RQ + e] = __float_as_uint(req_wo.y); rq[6 * MIPT_MERL_RQ + e] = __float_as_uint(req_wo.z);
						rq[7 * MIPT_MERL_RQ + e] = __float_as_uint(req_n.x); rq[8 * MIPT_MERL_RQ + e] = __float_as_uint(req_n.y); rq[9 * MIPT_MERL_RQ + e] = __float_as_uint(req_n.z);
						rq[10 * MIPT_MERL_RQ + e] = __float_as_uint(req_w.x); rq[11 * MIPT_MERL_RQ + e] = __float_as_uint(req_w.y); rq[12 * MIPT_MERL_RQ + e] = __float_as_uint(req_w.z);
						rq[13 * MIPT_MERL_RQ + e] = id; rq[14 * MIPT_MERL_RQ + e] = req_meta | (unsigned)kind;
					}
					rq_count += (unsigned)__popcll(hm);
					const bool drain = !got && kind == 1;                      // after the last chunk: whatever is left
					if (rq_count >= 64u || (drain && rq_count > 0u)) {
						const unsigned take = rq_count < 64u ? rq_count : 64u;
						rq_count -= take;
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
						unsigned alive_id = 0xffffffffu;
						if (lane_id() < take) {
							const unsigned e = rq_count + lane_id();
							const f3 x = mk3(__uint_as_float(rq[0 * MIPT_MERL_RQ + e]), __uint_as_float(rq[1 * MIPT_MERL_RQ + e]), __uint_as_float(rq[2 * MIPT_MERL_RQ + e]));
							const float fac = __uint_as_float(rq[3 * MIPT_MERL_RQ + e]);
							const f3 wo = mk3(__uint_as_float(rq[4 * MIPT_MERL_RQ + e]), __uint_as_float(rq[5 * MIPT_MERL_RQ + e]), __uint_as_float(rq[6 * MIPT_MERL_RQ + e]));
							const f3 nn = mk3(__uint_as_float(rq[7 * MIPT_MERL_RQ + e]), __uint_as_float(rq[8 * MIPT_MERL_RQ + e]), __uint_as_float(rq[9 * MIPT_MERL_RQ + e]));
							const f3 w0 = mk3(__uint_as_float(rq[10 * MIPT_MERL_RQ + e]), __uint_as_float(rq[11 * MIPT_MERL_RQ + e]), __uint_as_float(rq[12 * MIPT_MERL_RQ + e]));
							const unsigned rid = rq[13 * MIPT_MERL_RQ + e], meta = rq[14 * MIPT_MERL_RQ + e];
							const f3 brdf = merl_eval_inline(sc->obj[meta >> 16].merl, x, wo, nn);
							if (!(meta & 1u)) {                                   // A: Raytracer.cpp:548, then weight * contrib for the shadow stage
								const f3 contrib = mk3(0, 0, 0) + (mk3(1.f, 1.f, 1.f) * fac) * brdf;
								const f3 pc = w0 * contrib;
								wf_st(&wf.sh_c[rid], make_float4(pc.x, pc.y, pc.z, 0.f));
							} else {                                              // B: :611, then :240-241
								PathState np;
								np.weight = ((w0 * mk3(1.f, 1.f, 1.f)) * brdf) * fac;
								np.depth = (int)((meta >> 1) & 0x7fffu);
								if (path_alive(np)) {
									wf_st(&wf.wgt[rid], make_float4(np.weight.x, np.weight.y, np.weight.z, __uint_as_float(MIPT_WF_VALID | (unsigned)np.depth)));
									alive_id = rid;
								}
							}
						}
						const unsigned long long am = __ballot(alive_id != 0xffffffffu);
						if (alive_id != 0xffffffffu) alive_buf[n_alive + __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u))] = alive_id;
						n_alive += (unsigned)__popcll(am);
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					}
				}
			}
		}
		const unsigned* src = identity ? nullptr : list;
		if (BATCH) { queue_push2x(wf.list_sh, next, reinterpret_cast<unsigned long long*>(&wf.counters[MIPT_CNT_PAIR(b)]), cast_bits, cont_bits, src, base, alive_buf, n_alive); n_alive = 0; }
		else queue_push2(wf.list_sh, next, reinterpret_cast<unsigned long long*>(&wf.counters[MIPT_CNT_PAIR(b)]), cast_bits, cont_bits, src, base);
		if (SPLIT && n_mrq) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			unsigned mbase = 0;
			if (lane_id() == 0) mbase = atomicAdd(&wf.counters[MIPT_CNT(MIPT_WF_CNT_NMRQ + b)], n_mrq);
			mbase = __builtin_amdgcn_readfirstlane(mbase);
			for (unsigned k = lane_id(); k < n_mrq; k += 64u) wf.list_mrq[mbase + k] = mrq_stage[k];
			__builtin_amdgcn_wave_barrier();
			n_mrq = 0;
		}
		if (TIER == 1) queue_push(wf.list_slow, &wf.counters[MIPT_CNT(MIPT_WF_CNT_NSLOW + b)], slow_bits, src, base);
		if (BATCH && !got) break;
	}
	DCounters* my = MIPT_MY_COUNTERS(cnt);
	wave_add(&my->rays_closest, n_closest);
	wave_add(&my->rays_shadow, n_shadow);
}

// Shade tier 5, second half: the measured-BRDF evaluations of depth b (IsoMERLBRDF::eval, BRDF.h:204-246; MERLBRDFRead.cpp:76-206) as a stage of their
// own.  Tier 4 runs them in trips of 64 inside the vertex kernel, whose 168 registers (31 values spilled) allow 3 waves per SIMD; its waves wait for
// 64 % of their cycles and a third of its vector instructions are fp64 chains (profiles/r6_h_c4_fp64_mix_pmc_summary.txt).  Here a lane takes ONE request
// of the depth's queue (every lane busy, as in a trip), evaluates it and finishes it exactly as the trip's lane does:
//   A (next-event estimation)  sh_c[id] = w * (0 + ((1,1,1) * sa) * brdf)                                   Raytracer.cpp:548
//   B (continuation)           weight = ((w * (1,1,1)) * brdf) * fb; path_alive (:240-241) -> wgt[id], the id appended to the next depth's queue   :611
// Same operations on the same operands as path_vertex: bit-identical; the order of the next depth's queue differs, which no result depends on.
// The tables of sincos / acos / atan2 (mipt_libm64.h: 38.6 KB together) are copied to LDS when a block starts: an evaluation reads about sixty table words
// through dependent loads, and from the constant arrays every one of them is a vector-memory round trip.
#ifndef MIPT_MERL_EVAL_WAVES
#define MIPT_MERL_EVAL_WAVES 4
#endif
#ifndef MIPT_MERL_EVAL_BLOCK
#define MIPT_MERL_EVAL_BLOCK 512         // two blocks of 8 waves per CU at 4 waves per SIMD: 2 x 38.6 KB of tables
#endif
#ifndef MIPT_MERL_LDS_TABLES
#define MIPT_MERL_LDS_TABLES 1
#endif
#define MIPT_MERL_TABLE_WORDS (MIPT_L64_SINCOS_WORDS + MIPT_L64_ASNCS_WORDS + MIPT_L64_INROOT_WORDS + MIPT_L64_CIJ_WORDS + 1)      // (+1: cij rows are 7 words, keep the sum even)
__global__ void __launch_bounds__(MIPT_MERL_EVAL_BLOCK) __attribute__((amdgpu_waves_per_eu(MIPT_MERL_EVAL_WAVES))) k_wf_merl_eval(const DScene* __restrict__ sc, DWave wf, int b) {
	L64Tables TB = l64_tables();
#if MIPT_MERL_LDS_TABLES
	__shared__ uint64_t merl_tables[MIPT_MERL_TABLE_WORDS];
	{
		uint64_t* const t_sincos = merl_tables, *const t_asncs = t_sincos + MIPT_L64_SINCOS_WORDS, *const t_inroot = t_asncs + MIPT_L64_ASNCS_WORDS, *const t_cij = t_inroot + MIPT_L64_INROOT_WORDS;
		for (unsigned k = threadIdx.x; k < MIPT_L64_SINCOS_WORDS; k += MIPT_MERL_EVAL_BLOCK) t_sincos[k] = TB.sincos[k];
		for (unsigned k = threadIdx.x; k < MIPT_L64_ASNCS_WORDS; k += MIPT_MERL_EVAL_BLOCK) t_asncs[k] = TB.asncs[k];
		for (unsigned k = threadIdx.x; k < MIPT_L64_INROOT_WORDS; k += MIPT_MERL_EVAL_BLOCK) t_inroot[k] = TB.inroot[k];
		for (unsigned k = threadIdx.x; k < MIPT_L64_CIJ_WORDS; k += MIPT_MERL_EVAL_BLOCK) t_cij[k] = TB.cij[k];
		__syncthreads();
		TB.sincos = t_sincos; TB.asncs = t_asncs; TB.inroot = t_inroot; TB.cij = t_cij;
	}
#endif
	const unsigned n = wf.counters[MIPT_CNT(MIPT_WF_CNT_NMRQ + b)];
	unsigned* head = &wf.counters[MIPT_CNT(MIPT_WF_CNT_MRQ_HEAD + b)];
	const unsigned* __restrict__ list = wf.list_mrq;
	unsigned* __restrict__ next = wf.list[(b + 1) & 1];
	unsigned base;
	QueuePuller q; q.init();
	while (q.pull(head, n, base)) {
		unsigned alive_bits = 0;
#pragma unroll 1
		for (int u = 0; u < (int)MIPT_WF_UNROLL; u++) {
			const unsigned idx = base + 64u * (unsigned)u + lane_id();
			if (idx >= n) continue;
			const unsigned e = list[idx], id = e & 0x7fffffffu;
			const bool kind_b = (e >> 31) != 0;
			const float4 h4 = wf_ld(&wf.hit[id]), a4 = wf_ld(&wf.mq_a[id]), b4 = wf_ld(&wf.mq_b[id]);
			const float4 x4 = kind_b ? wf_ld(&wf.ray_d[id]) : wf_ld(&wf.sh_d[id]);
			const unsigned meta = __float_as_uint(b4.w);
			const f3 brdf = merl_eval_inline(sc->obj[meta >> 15].merl, mk3(x4.x, x4.y, x4.z), mk3(h4.x, h4.y, h4.z), mk3(a4.x, a4.y, a4.z), TB);
			const f3 w0 = mk3(b4.x, b4.y, b4.z);
			if (!kind_b) {
				const f3 contrib = mk3(0, 0, 0) + (mk3(1.f, 1.f, 1.f) * h4.w) * brdf;
				const f3 pc = w0 * contrib;
				wf_st(&wf.sh_c[id], make_float4(pc.x, pc.y, pc.z, 0.f));
			} else {
				PathState np;
				np.weight = ((w0 * mk3(1.f, 1.f, 1.f)) * brdf) * a4.w;
				np.depth = (int)(meta & 0x7fffu);
				if (path_alive(np)) {
					wf_st(&wf.wgt[id], make_float4(np.weight.x, np.weight.y, np.weight.z, __uint_as_float(MIPT_WF_VALID | (unsigned)np.depth)));
					alive_bits |= 1u << u;
				}
			}
		}
		// the continuing paths of this chunk: one atomic on the next depth's count (the high word of the pair shade(b) appends to)
		unsigned long long m[MIPT_WF_UNROLL];
		unsigned total = 0;
#pragma unroll
		for (int u = 0; u < (int)MIPT_WF_UNROLL; u++) { m[u] = __ballot((alive_bits >> u) & 1u); total += (unsigned)__popcll(m[u]); }
		if (total) {
			unsigned nb = 0;
			if (lane_id() == 0) nb = atomicAdd(&wf.counters[MIPT_CNT_PAIR(b) + 1], total);
			nb = __builtin_amdgcn_readfirstlane(nb);
#pragma unroll
			for (int u = 0; u < (int)MIPT_WF_UNROLL; u++) {
				if ((alive_bits >> u) & 1u) next[nb + __builtin_amdgcn_mbcnt_hi((unsigned)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[u], 0u))] = list[base + 64u * (unsigned)u + lane_id()] & 0x7fffffffu;
				nb += (unsigned)__popcll(m[u]);
			}
		}
	}
}

// shadow: Scene::intersection_shadow; a visible light sample adds weight*contrib to the path colour
__global__ void __launch_bounds__(MIPT_BLOCK) k_wf_shadow(const DScene* __restrict__ sc, DWave wf, int b) {
	MIPT_DECLARE_STACK(stk);
	const unsigned n = MIPT_N_SHADOW(wf, b);
	unsigned* head = &wf.counters[MIPT_CNT_SH_HEAD(b)];
	const unsigned* __restrict__ list = wf.list_sh;
	unsigned base;
	QueuePuller q; q.init();
	while (q.pull(head, n, base)) {
#pragma unroll 1
		for (int u = 0; u < MIPT_WF_UNROLL; u++) {
			unsigned idx = base + 64u * u + lane_id();
			if (idx >= n) continue;
			unsigned id = list[idx];
			float4 o = wf_ld(&wf.sh_o[id]), d = wf_ld(&wf.sh_d[id]);
			Ray r; r.o = mk3(o.x, o.y, o.z); r.d = mk3(d.x, d.y, d.z);
			if (!scene_occluded(sc, r, o.w, stk)) {
				float4 c = wf_ld(&wf.out.col[id]), pc = wf_ld(&wf.sh_c[id]);
				wf_st(&wf.out.col[id], make_float4(c.x + pc.x, c.y + pc.y, c.z + pc.z, 0.f));    // Raytracer.cpp:566
			}
		}
	}
}

// ---------------------------------------------------------------- ray reordering (option sort_rays)
// The closest-hit queue of depth b+1 is written by shade(b) in path-id order: neighbouring entries start at neighbouring
// surface points but leave in unrelated directions.  A stable counting sort by the octant of the direction (8 bins, order
// inside a bin kept) puts rays that take the same near / far decisions at every node side by side, so the lanes of a wave
// walk the tree together for longer.  Which lane runs a ray never changes its result (mipt_persistent.h).
#define MIPT_SORT_BINS 8
#define MIPT_SORT_BLOCK 256
MIPT_DEV unsigned sort_key(const DWave& wf, unsigned id) {
	const float4 d = wf.ray_d[id];
	return (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
}
MIPT_DEV void sort_segment(unsigned n, unsigned& lo, unsigned& hi) {
	const unsigned seg = ((n + gridDim.x - 1) / gridDim.x + MIPT_SORT_BLOCK - 1) / MIPT_SORT_BLOCK * MIPT_SORT_BLOCK;
	lo = min(n, blockIdx.x * seg); hi = min(n, lo + seg);
}
__global__ void __launch_bounds__(MIPT_SORT_BLOCK) k_sort_hist(DWave wf, const unsigned* __restrict__ list, const unsigned* __restrict__ n_ptr, unsigned* __restrict__ hist) {
	__shared__ unsigned cnt[MIPT_SORT_BINS];
	if (threadIdx.x < MIPT_SORT_BINS) cnt[threadIdx.x] = 0;
	__syncthreads();
	unsigned lo, hi;
	sort_segment(*n_ptr, lo, hi);
	for (unsigned i = lo + threadIdx.x; i < hi; i += MIPT_SORT_BLOCK) {
		const unsigned k = sort_key(wf, list[i]);
#pragma unroll
		for (unsigned b = 0; b < MIPT_SORT_BINS; b++) {
			const unsigned long long m = __ballot(k == b);
			if (lane_id() == 0 && m) atomicAdd(&cnt[b], (unsigned)__popcll(m));
		}
	}
	__syncthreads();
	if (threadIdx.x < MIPT_SORT_BINS) hist[threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];     // bin-major
}
// exclusive scan of the bin-major histogram (one block)
__global__ void __launch_bounds__(1024) k_sort_scan(unsigned* __restrict__ hist, unsigned total) {
	__shared__ unsigned part[1024];
	__shared__ unsigned carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (unsigned base = 0; base < total; base += 1024) {
		const unsigned i = base + threadIdx.x;
		const unsigned v = i < total ? hist[i] : 0u;
		part[threadIdx.x] = v;
		__syncthreads();
		for (unsigned ofs = 1; ofs < 1024; ofs <<= 1) {
			const unsigned a = threadIdx.x >= ofs ? part[threadIdx.x - ofs] : 0u;
			__syncthreads();
			part[threadIdx.x] += a;
			__syncthreads();
		}
		if (i < total) hist[i] = carry + part[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry += part[1023];
		__syncthreads();
	}
}
__global__ void __launch_bounds__(MIPT_SORT_BLOCK) k_sort_scatter(DWave wf, const unsigned* __restrict__ list, const unsigned* __restrict__ n_ptr, const unsigned* __restrict__ offsets, unsigned* __restrict__ out) {
	__shared__ unsigned base[MIPT_SORT_BINS];
	__shared__ unsigned wcnt[MIPT_SORT_BLOCK / 64][MIPT_SORT_BINS];
	if (threadIdx.x < MIPT_SORT_BINS) base[threadIdx.x] = offsets[threadIdx.x * gridDim.x + blockIdx.x];
	__syncthreads();
	unsigned lo, hi;
	sort_segment(*n_ptr, lo, hi);
	const unsigned w = threadIdx.x >> 6, lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	for (unsigned t = lo; t < hi; t += MIPT_SORT_BLOCK) {
		const unsigned i = t + threadIdx.x;
		const bool on = i < hi;
		const unsigned id = on ? list[i] : 0u;
		const unsigned k = on ? sort_key(wf, id) : 0xffu;
		unsigned rank = 0;
#pragma unroll
		for (unsigned b = 0; b < MIPT_SORT_BINS; b++) {
			const unsigned long long m = __ballot(k == b);
			if (k == b) rank = (unsigned)__popcll(m & below);
			if (lane == 0) wcnt[w][b] = (unsigned)__popcll(m);
		}
		__syncthreads();
		if (on) {
			unsigned o = base[k] + rank;
			for (unsigned ww = 0; ww < w; ww++) o += wcnt[ww][k];
			out[o] = id;
		}
		__syncthreads();
		if (threadIdx.x < MIPT_SORT_BINS) { unsigned s = 0; for (unsigned ww = 0; ww < MIPT_SORT_BLOCK / 64; ww++) s += wcnt[ww][threadIdx.x]; base[threadIdx.x] += s; }
		__syncthreads();
	}
}
