// mipt_queue_wave.h — getColor's contribution queue as a wavefront of stage kernels (pipeline 2 since round 2).
//
// mipt_compositing.h keeps the whole loop of one sample in one thread (trace_path_queue: getColor statement by statement).
// That is the simplest correct form and it stays, as the fallback below and as the statement of what this file must
// compute; but a thread that traverses the BVH inside a 270-register loop runs at the speed of the per-path kernel of
// pipeline 0, a fifth of the wavefront stages.  Here the loop is cut at its ray queries:
//
//   logic -> [ closest hits | any hits ] -> logic -> ...          one ROUND = one query of every live sample
//
// * One engine serves all contributions of a sample, so the FIFO order decides which random numbers a contribution sees
//   (Raytracer.cpp:213-238).  A sample therefore has ONE contribution in flight: its queries are issued one per round, in
//   the order the reference issues them, and every draw happens in the logic stage in the reference's statement order.
//   The parallelism is across samples (millions per pass), not inside one.
// * The logic stage (k_q_logic) is trace_path_queue cut into segments: POP (next contribution, camera-side tests) ->
//   main closest hit -> A1 (what the ray saw, material, subsurface probe, light sample) -> any hit -> A2 (ghost
//   pass-through / direct term) -> [fog: in-scattering event -> closest hit of its direction -> F1] -> A3 (continuation).
//   What a segment needs from an earlier one lives in a per-sample frame in HBM; everything is recomputed that is cheaper
//   to recompute than to store.  The arithmetic of every segment is the statement sequence of trace_path_queue.
// * The queries run on the persistent traversal kernels of pipeline 1 (k_q_traverse: the same traverse_queue, explicit
//   queue descriptor; shadow rays skip ghost objects, Geometry.cpp:722).  As there, the stage that creates a ray
//   tests the analytic objects itself.
// * The subsurface probe of a mesh (TriMesh::reservoir_sampling_intersection draws from the sample's engine inside its leaf
//   loop) is a third kind of query: A1 stops where getColor calls get_random_intersection, k_q_probe (the persistent
//   traversal in its reservoir mode, mipt_persistent.h) runs it with the sample's engine and A1 is entered again: it recomputes the deterministic head of the vertex from the saved closest hit,
//   takes the numbers it had drawn from the frame and goes on behind the call.
// * The reference's FIFO has 200 entries (Raytracer.h:114); 200 x 48 B per sample would cap a pass at 2 M samples and
//   the persistent kernels need far larger batches.  A sample gets a ring of `DQueueWave::ring` entries here (option `queue_ring`,
//   default MIPT_QW_RING, at most MIPT_QW_FIFO); the rare sample
//   that needs more is abandoned (nothing of it is kept) and rendered afterwards by trace_path_queue with the full
//   200-entry ring (k_render_paths_queue_list), which starts it again from its seed: same result either way.
//   Round 4: the ring's MEMORY is `ring` entries too (it was MIPT_QW_FIFO = 32 whatever the option said: 1 536 of a sample's 1 950 bytes,
//   which cut the 1080p x 64 spp frame of the rate probe into two passes).  On the probe's three scenes no sample ever holds more than 8
//   pending contributions (ring 8: 0 samples through the fallback, ring 4: 333 of 133 M in the fog scene); the default of 16 entries
//   makes the frame ONE pass of 157 GB.
#pragma once

#ifndef MIPT_QW_FIFO
#define MIPT_QW_FIFO 32                  // the largest ring a sample can be given
#endif
#ifndef MIPT_QW_RING
#define MIPT_QW_RING 16                  // the default ring
#endif
#ifndef MIPT_QW_LOGIC_WAVES
#define MIPT_QW_LOGIC_WAVES 3            // general builds of the logic stage (188-237 registers unconstrained; round 3, after sinf / cosf lost their selected constants: 3 waves with 20-70 spilled values beat 2 without — subsurface logic 59.5 -> 51.8 ms, fog 110.7 -> 107.8, ghost +-0); + 1 for the any-hit-list stage of the build without the fog code
#endif
#ifndef MIPT_QW_FAST_WAVES
#ifndef MIPT_QW_UNROLL
#define MIPT_QW_UNROLL 4                 // sub-chunks of 64 entries per queue atomic in the logic stage
#endif
#define MIPT_QW_FAST_WAVES 4             // the fast tier of the closest-hit-list stage (scenes without fog / subsurface groups)
#endif
#ifndef MIPT_QW_FAST
#define MIPT_QW_FAST 1
#endif
#define MIPT_QW_FRAME 13                  // float4 slots of the per-sample frame
// The per-sample state (1.9 KB) is touched once per call: streaming cache policy, as the path state of pipeline 1
#ifndef MIPT_QW_STREAM
#define MIPT_QW_STREAM 1
#endif
#if MIPT_QW_STREAM
#define QW_LD(p) wf_ld(p)
#define QW_ST(p, v) wf_st((p), (v))
#else
#define QW_LD(p) (*(p))
#define QW_ST(p, v) (*(p) = (v))
#endif
enum { QW_POP = 0, QW_A1 = 1, QW_A2 = 2, QW_F1 = 3, QW_DONE = 4, QW_PROBE = 5, QW_T5 = 6 };
// counters of one round (cyclic, 4 round slots): every word on its own 128-byte line
#define MIPT_QW_SLOT_WORDS (10 * 32)
#define MIPT_QW_PAIR(s) ((s) * MIPT_QW_SLOT_WORDS)                // 64-bit {n_shadow, n_closest} of the round's request lists
#define MIPT_QW_HEAD_CLOSEST(s) ((s) * MIPT_QW_SLOT_WORDS + 32)
#define MIPT_QW_HEAD_SHADOW(s) ((s) * MIPT_QW_SLOT_WORDS + 64)
#define MIPT_QW_HEAD_LOGIC_A(s) ((s) * MIPT_QW_SLOT_WORDS + 96)   // logic stage over the previous round's closest list
#define MIPT_QW_HEAD_LOGIC_B(s) ((s) * MIPT_QW_SLOT_WORDS + 128)  // ... and over its shadow list
#define MIPT_QW_N_PROBE(s) ((s) * MIPT_QW_SLOT_WORDS + 160)       // entries of the round's probe list; the next word: entries of the list of
#define MIPT_QW_N_SHADOW_ADD(s) ((s) * MIPT_QW_SLOT_WORDS + 161)  // any-hit requests nobody waits for (the traversal adds the pending term)
#define MIPT_QW_HEAD_SHADOW_ADD(s) ((s) * MIPT_QW_SLOT_WORDS + 224)
#define MIPT_QW_HEAD_PROBE(s) ((s) * MIPT_QW_SLOT_WORDS + 256)
#define MIPT_QW_HEAD_LOGIC_C(s) ((s) * MIPT_QW_SLOT_WORDS + 192)  // logic stage over the previous round's probe list
#define MIPT_QW_N_SLOW(s) ((s) * MIPT_QW_SLOT_WORDS + 162)        // samples the fast tier of the round's logic stage left to the general build
#define MIPT_QW_HEAD_LOGIC_SLOW(s) ((s) * MIPT_QW_SLOT_WORDS + 288)
#define MIPT_QW_N_OVERFLOW (4 * MIPT_QW_SLOT_WORDS)
#define MIPT_QW_COUNTERS (4 * MIPT_QW_SLOT_WORDS + 32)

struct DQueueWave {
	QContrib* fifo;                  // [N][ring]
	float4 *cur_w, *cur_o, *cur_d;   // the contribution being processed (w.w: its depth / flag bits)
	float4* acc;                     // xyz: the sample's colour so far; w: attenuationFactor (Raytracer.cpp:206, kept across contributions).
	                                 // The same array as wf.out.col: the any-hit stage adds a pending direct term to it (below)
	unsigned* ctl;                   // head | count << 8 | phase << 16 | fog site << 24
	float4* fr;                      // frame, slot-major: fr[slot * N + id]
	float* vis;                      // any-hit results
	unsigned* live[2];               // ids with a closest-hit request, by round parity
	unsigned* shl[2];                // ids with an any-hit request
	unsigned* prl[2];                // ids with a subsurface probe request
	unsigned* sha[2];                // ids with an any-hit request whose answer only decides whether wf.sh_c is added to the colour
	unsigned* slow;                  // ids the fast tier of the logic stage left to the general build (consumed in the same round)
	unsigned* overflow;              // ids whose ring overflowed
	unsigned* counters;
	float4 *aov_n, *aov_kd;          // denoiser inputs or null
	unsigned N;
	unsigned ring;                   // pending contributions a sample may hold here = entries of its ring in `fifo` (<= MIPT_QW_FIFO)
};

// shadow rays of getColor ignore ghost objects (avoid_ghosts = true, Raytracer.cpp:513; Geometry.cpp:722)
MIPT_DEV bool qw_analytic_occluded(const DScene* __restrict__ sc, f3 ro, f3 rd, float dist) {
	const int n = sc->nobj;
	bool occ = false;
	for (int i = 0; i < n; i++) {
		const DObject& o = sc->obj[i];
		if (o.type == 0 || o.ghost) continue;
		f3 d = xf_dir(o.inv, rd);
		f3 org = xf_point(o.inv, ro);
		float tt;
		bool hit = (o.type == 1) ? sphere_test(o, org, d, tt) : plane_test(o, org, d, tt);
		if (hit && ((double)tt < (double)dist * 0.999)) occ = true;
	}
	return occ;
}
MIPT_DEV bool qw_meshes_missed(const DScene* __restrict__ sc, f3 ro, f3 rd, float dist) {
	const int n = sc->nobj;
	bool missed = true;
	for (int i = sc->first_mesh; i < n; i++) {
		const DObject& o = sc->obj[i];
		if (o.type != 0 || o.ghost) continue;
		const f3 d = xf_dir(o.inv, rd);
		const f3 org = xf_point(o.inv, ro);
		const f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
		float t_root;
		if (box_test<false>(ld3(o.root_min), ld3(o.root_max), org, invd, invd.x >= 0, invd.y >= 0, invd.z >= 0, t_root) && !(t_root > dist)) missed = false;
	}
	return missed;
}

// The camera contribution of every sample, already popped (Raytracer.cpp:231-246 for the first entry of the FIFO: depth and
// weight pass the tests of :240-241 by construction) and with its closest-hit request made: round 0 is the closest-hit
// traversal over all path slots (identity queue; bit 31 of cur_w.w marks the slots that hold a sample, as MIPT_WF_VALID of
// pipeline 1's wgt — wf.wgt IS qw.cur_w here).
__global__ void __launch_bounds__(MIPT_BLOCK) k_q_begin(const DScene* __restrict__ sc, DRender R, DPass ps, DWave wf, DQueueWave qw, DCounters* __restrict__ cnt) {
	const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const long long total = (long long)ps.npix_slots * (ps.k1 - ps.k0);
	if (tid >= total) return;
	const int kk = (int)(tid / ps.npix_slots), slot = (int)(tid % ps.npix_slots);
	const int blk = slot >> 6, in = slot & 63;
	const int i = ps.blocks[2 * blk] + (in >> 3), j = ps.blocks[2 * blk + 1] + (in & 7);
	if (qw.aov_n) { qw.aov_n[tid] = make_float4(0.f, 0.f, 0.f, 0.f); qw.aov_kd[tid] = make_float4(0.f, 0.f, 0.f, 0.f); }
	wf.out.col[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
	if (!(i < R.H && j < R.W)) { qw.ctl[tid] = (unsigned)QW_DONE << 16; qw.cur_w[tid] = make_float4(0.f, 0.f, 0.f, 0.f); wf.out.dxdy[tid] = make_float2(0.f, 0.f); return; }
	PathState p; float dx, dy;
	path_begin(R, i, j, ps.k0 + kk, p, dx, dy);
	wf.out.dxdy[tid] = make_float2(dx, dy);
	wf.rng[tid] = make_uint2((unsigned)p.rng, (unsigned)(p.rng >> 32));
	const bool dead = (R.nb_bounces & 0xffff) == 0;                  // :240 (depth 0: nothing is traced)
	qw.cur_w[tid] = make_float4(1.f, 1.f, 1.f, __uint_as_float((unsigned)(R.nb_bounces & 0xffff) | 0x10000u | 0x20000u | (dead ? 0u : MIPT_WF_VALID)));
	if (dead) { qw.ctl[tid] = (unsigned)QW_DONE << 16; return; }
	float t0; unsigned best0;
	analytic_prefix_closest(sc, p.ray.o, p.ray.d, t0, best0);
	wf.ray_o[tid] = make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, t0);
	wf.ray_d[tid] = make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, __uint_as_float(best0));
	qw.ctl[tid] = (unsigned)QW_A1 << 16;                             // empty ring, waiting for the closest hit
	(void)cnt;                                                       // the camera rays are counted on the host (valid pixels x samples)
}

// The state of one sample while its segments run.
struct QwSample {
	uint64_t rng;
	f3 color; float att;
	unsigned head, count;
	bool overflow;
};

MIPT_DEV bool has_fog_early(const DRender& R) { return R.fog_density > 1E-8; }
// One sample: run segments until the next ray query.  Returns -1 (abandoned) or the requests it made as bits: 1 (closest
// hit, wf.ray_o / ray_d), 2 (any hit, wf.sh_o / sh_d, the sample waits for the answer), 4 (subsurface probe, wf.ray_o /
// ray_d: origin and direction in the object's frame, .w = tmax / the object), 8 (any hit, wf.sh_o / sh_d, nobody waits:
// the traversal stage adds wf.sh_c to the colour if the light sample is visible); 0 = finished.
//
// With fog the same request is made early too (bit 8 with the closest-hit request of the fog event, both answered in one
// round; qw.vis is read where the direct term is added, TAIL of site 5), so a vertex costs two rounds instead of three.
// Bit 8 without fog is the common vertex: an opaque, non-ghost surface.  The answer of its light-sample query decides
// nothing but whether the direct term is added (Raytracer.cpp:538-566; the env flag of the continuation, :626-629, reads
// isShadowed only for ghosts) and the query draws no random numbers, so the vertex goes on to its continuation in the same
// call and its any-hit query runs in the same round as the closest-hit query of the next contribution; the term is added
// to the colour by the any-hit stage, i.e. before the next vertex adds anything: the order of the additions is the
// reference's.
//
// FAST (round 3): the tier of pipeline 1's k_wf_shade<1> for this pipeline — a build for the closest-hit list of scenes without fog
// and without subsurface groups that knows only what most vertices are: a miss, the light, the environment sphere, a mirror,
// a dielectric, or a Lambert vertex (Ks = 0, Ne >= 0, no measured BRDF; of a ghost: up to its any-hit request).  Everything else
// returns 16 BEFORE anything of the sample has been modified (nothing is stored, no counter is touched, the engine is advanced
// on a copy) and the general build takes the sample in the same round (list_slow).  The Lambert vertex is path_vertex_fast's
// (mipt_shade.h: eval = Kd / pi, the diffuse lobe picked, pdf = dot / pi): the arithmetic of the general code with the terms
// that are exactly zero left out.  Without the Phong lobe (fp64 pow), the measured BRDF, the ghost / fog / subsurface segments
// and the frame the build needs 128 registers instead of 206: 4 waves per SIMD instead of 2.
// LAMBERT (round 4): a build of the general stage for scenes whose materials are all Lambert (upload: no measured BRDF, every specular list a
// constant 0, every exponent list a constant >= 0).  The two BRDF evaluations and the lobe pick of a vertex that IS plain (checked per vertex,
// like the fast tier: Ks = 0, Ne >= 0, no table) are path_vertex_fast's — eval = Kd / pi, the diffuse lobe unless the engine returns
// u = 1.0f, pdf = dot / pi —; a vertex that is not (in a scene chosen for this build: the 2^-25 lobe event) abandons its sample to the
// one-thread-per-sample loop exactly like a ring overflow does (the sample is recomputed from its first ray).  Round 3 measured where the
// general builds' registers go by compiling them with pieces left out: 220 -> 189 (fog, closest-hit list) and 200 -> 156 (fog, any-hit
// list) without the glossy / measured BRDF inline.
template <bool SUBS, bool SHADOW_LIST, bool FOG, bool FAST = false, bool LAMBERT = false>
__device__ __forceinline__ int qw_advance(const DScene* __restrict__ sc, const DRender& R, const DPass& ps, const DWave& wf, const DQueueWave& qw, const unsigned id,
                                       unsigned& n_closest, unsigned& n_shadow) {
	const unsigned N = qw.N;
	unsigned ctl = qw.ctl[id];
	int phase = (int)((ctl >> 16) & 0xffu);
	int site = (int)(ctl >> 24);
	if (phase == QW_DONE) return 0;
	static_assert(!FAST || (!SUBS && !SHADOW_LIST && !FOG), "the fast tier serves the closest-hit list of scenes without fog and subsurface groups");
	if (FAST && phase != QW_A1) return 16;
	QwSample S;
	{ const uint2 rs = QW_LD(&wf.rng[id]); S.rng = (uint64_t)rs.x | ((uint64_t)rs.y << 32); }
	// The colour is read when something is added to it and written back when it has changed.  Adding a zero vector never
	// changes it, not even in the sign of a zero: the colour starts as +0 and x + y is -0 only for (-0) + (-0), so no
	// component is ever -0 and c + (+-0) == c bit for bit; most vertices add m.Ke = 0 (:411) and nothing else in their call.
	// attenuationFactor lives in .w and matters with fog only (read and written with the colour there).
	S.color = mk3(0, 0, 0); S.att = 0.f;
	bool color_loaded = false, color_dirty = false;
	auto load_color = [&]() { if (!color_loaded) { const float4 a = QW_LD(&qw.acc[id]); S.color = mk3(a.x, a.y, a.z); S.att = a.w; color_loaded = true; } };
	auto add_color = [&](f3 v) {
		if (v.x == 0.f && v.y == 0.f && v.z == 0.f) return;
		load_color();
		S.color = S.color + v; color_dirty = true;
	};
	if (FOG && has_fog_early(R)) load_color();
	S.head = ctl & 0xffu; S.count = (ctl >> 8) & 0xffu; S.overflow = false;
	QContrib* const fifo = qw.fifo + (size_t)id * qw.ring;
	auto FRL = [&](int slot) -> float4 { return QW_LD(&qw.fr[(size_t)slot * N + id]); };
	auto FRS = [&](int slot, float4 v) { QW_ST(&qw.fr[(size_t)slot * N + id], v); };
	const bool has_bg = R.backgroundW > 0 && R.background != nullptr;       // :220
	const bool has_fog = FOG && R.fog_density > 1E-8;                       // :207 (FOG = false: the build for scenes without fog, no fog code in it)
	const f3 cl = ld3(R.centerLight);
	// pixel / sample index of this path id
	const int kk = (int)(id / (unsigned)ps.npix_slots), slot_ = (int)(id % (unsigned)ps.npix_slots);
	const int blk = slot_ >> 6, in_ = slot_ & 63;
	const int pi = ps.blocks[2 * blk] + (in_ >> 3), pj = ps.blocks[2 * blk + 1] + (in_ & 7);
	const int pix = pi * R.W + pj, k = ps.k0 + kk;

	QContrib front; front.w = front.o = front.d = make_float4(0.f, 0.f, 0.f, 0.f);
	bool have_front = false;
	auto push = [&](f3 w, const Ray& r, int depth, bool lights, bool env, bool hadSS) {
		if (S.count >= qw.ring) { S.overflow = true; return; }
		QContrib c;
		c.w = make_float4(w.x, w.y, w.z, __uint_as_float((unsigned)(depth & 0xffff) | (lights ? 0x10000u : 0u) | (env ? 0x20000u : 0u) | (hadSS ? 0x40000u : 0u)));
		c.o = make_float4(r.o.x, r.o.y, r.o.z, 0.f); c.d = make_float4(r.d.x, r.d.y, r.d.z, 0.f);
		if (S.count == 0) { front = c; have_front = true; }         // into an empty ring: handed to POP in registers (written by save() if the call ends first)
		else { unsigned at = S.head + S.count; if (at >= qw.ring) at -= qw.ring; QContrib* const e = fifo + at; QW_ST(&e->w, c.w); QW_ST(&e->o, c.o); QW_ST(&e->d, c.d); }   // (head < ring, count < ring)
		S.count++;
	};
	auto save = [&](int ph, int st) {
		if (have_front) { QContrib* const e = fifo + S.head; QW_ST(&e->w, front.w); QW_ST(&e->o, front.o); QW_ST(&e->d, front.d); have_front = false; }
		QW_ST(&wf.rng[id], make_uint2((unsigned)S.rng, (unsigned)(S.rng >> 32)));
		if (color_dirty) QW_ST(&qw.acc[id], make_float4(S.color.x, S.color.y, S.color.z, S.att));
		qw.ctl[id] = (S.head & 0xffu) | ((S.count & 0xffu) << 8) | ((unsigned)ph << 16) | ((unsigned)st << 24);
	};
	auto finish = [&]() { save(QW_DONE, 0); };       // qw.acc IS wf.out.col (.w is not read by the splat)
	auto request_closest = [&](const Ray& r) {
		float t0; unsigned best0;
		analytic_prefix_closest(sc, r.o, r.d, t0, best0);
		QW_ST(&wf.ray_o[id], make_float4(r.o.x, r.o.y, r.o.z, t0));
		QW_ST(&wf.ray_d[id], make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(best0)));
		n_closest++;
	};

	// the contribution in flight (valid from QW_A1 on)
	f3 pathWeight = mk3(0, 0, 0); Ray currentRay; currentRay.o = mk3(0, 0, 0); currentRay.d = mk3(0, 0, 1);
	int nbrebonds = 0; bool show_lights = false, show_envmap = false, hadSS = false;
	// weight and flags live in cur_w; the ray itself is read from the closest-hit request (wf.ray_o / ray_d hold it from POP
	// until the next request of the sample): only the subsurface probe replaces the request while the ray is still needed,
	// so its yield parks the ray in cur_o / cur_d and its answer puts it back.  F1 / TAIL / A3 never read the ray.
	if (phase != QW_POP) {
		const float4 w = QW_LD(&qw.cur_w[id]);
		const unsigned bits = __float_as_uint(w.w);
		pathWeight = mk3(w.x, w.y, w.z);
		nbrebonds = (int)(bits & 0xffffu); show_lights = (bits & 0x10000u) != 0; show_envmap = (bits & 0x20000u) != 0; hadSS = (bits & 0x40000u) != 0;
		if (phase == QW_A1 || (phase == QW_A2 && has_fog)) {
			const float4 o = QW_LD(&wf.ray_o[id]), d = QW_LD(&wf.ray_d[id]);
			currentRay.o = mk3(o.x, o.y, o.z); currentRay.d = mk3(d.x, d.y, d.z);
		} else if (phase == QW_PROBE) {
			const float4 o = QW_LD(&qw.cur_o[id]), d = QW_LD(&qw.cur_d[id]);
			currentRay.o = mk3(o.x, o.y, o.z); currentRay.d = mk3(d.x, d.y, d.z);
			QW_ST(&wf.ray_o[id], o); QW_ST(&wf.ray_d[id], d);
		}
	}

	// vertex locals that cross segment boundaries (frame)
	f3 P = mk3(0, 0, 0), Nn = mk3(0, 1, 0), rayDirection = mk3(0, 0, 1), Ksub = mk3(0, 0, 0), subsW = mk3(1, 1, 1), dir_l = mk3(0, 0, 1), wi = mk3(0, 0, 1), contrib = mk3(0, 0, 0);
	Mat m; m.shadingN = mk3(0, 1, 0); m.Kd = mk3(0.5f, 0.5f, 0.5f); m.Ks = mk3(0, 0, 0); m.Ne = mk3(100, 100, 100); m.Ke = mk3(0, 0, 0); m.transp = false; m.refr_index = 0;
	float t_main = 0.f, d_light2 = 0.f;
	int objid = 0; bool sub_interaction = false, isShadowed = false;
	int pending_add = 0;                  // 8: this call left an any-hit request whose term the traversal stage adds
	bool deferred = false;                // fog: the any-hit request of the vertex is in flight, its answer (qw.vis) is read where the direct term is added
	// with_light: the light sample (dir_l, wi) too — only A2 entered from an any-hit answer needs it; SUBS-less builds have subsW = 1
	auto save_vertex = [&](bool with_light) {
		FRS(0, make_float4(P.x, P.y, P.z, t_main));
		FRS(1, make_float4(Nn.x, Nn.y, Nn.z, __uint_as_float((unsigned)objid | (sub_interaction ? 0x100u : 0u) | (isShadowed ? 0x200u : 0u) | (deferred ? 0x400u : 0u))));
		FRS(2, make_float4(rayDirection.x, rayDirection.y, rayDirection.z, d_light2));
		FRS(3, make_float4(m.Kd.x, m.Kd.y, m.Kd.z, m.Ks.x));
		FRS(4, make_float4(m.Ks.y, m.Ks.z, m.Ne.x, m.Ne.y));
		FRS(5, make_float4(m.Ne.z, Ksub.x, Ksub.y, Ksub.z));
		if (SUBS) FRS(6, make_float4(subsW.x, subsW.y, subsW.z, 0.f));
		if (with_light) { FRS(7, make_float4(dir_l.x, dir_l.y, dir_l.z, 0.f)); FRS(8, make_float4(wi.x, wi.y, wi.z, 0.f)); }
	};
	auto load_vertex = [&](bool with_light) {
		const float4 a = FRL(0), b = FRL(1), c = FRL(2), d = FRL(3), e = FRL(4), f = FRL(5);
		const float4 g = SUBS ? FRL(6) : make_float4(1.f, 1.f, 1.f, 0.f);
		const float4 h = with_light ? FRL(7) : make_float4(0.f, 0.f, 1.f, 0.f), i2 = with_light ? FRL(8) : make_float4(0.f, 0.f, 1.f, 0.f);
		P = mk3(a.x, a.y, a.z); t_main = a.w;
		Nn = mk3(b.x, b.y, b.z); { const unsigned fl = __float_as_uint(b.w); objid = (int)(fl & 0xffu); sub_interaction = (fl & 0x100u) != 0; isShadowed = (fl & 0x200u) != 0; deferred = (fl & 0x400u) != 0; }
		rayDirection = mk3(c.x, c.y, c.z); d_light2 = c.w;
		m.Kd = mk3(d.x, d.y, d.z); m.Ks = mk3(d.w, e.x, e.y); m.Ne = mk3(e.z, e.w, f.x); Ksub = mk3(f.y, f.z, f.w);
		subsW = mk3(g.x, g.y, g.z); dir_l = mk3(h.x, h.y, h.z); wi = mk3(i2.x, i2.y, i2.z);
		m.shadingN = Nn;
	};

	// fog: the first half of fogContribution (mipt_compositing.h fog_contribution up to its visibility query).  Returns
	// true when the closest-hit request of the in-scattering direction has been written (the caller yields with QW_F1).
	auto fog_begin = [&](int st, const Ray& r, f3 sampleLightPos) -> bool {
		if (S.overflow) return false;                  // the sample is being abandoned (loop head)
		const f3 curWeight = pathWeight;
		const float t = t_main;
		if (norm2(curWeight) < 1E-12) return false;
		const f3 rd = r.d;
		const float p_uniform = 0.5f;
		const bool is_uniform_fog = R.fog_type == 0;
		const float alpha = R.fog_absorption, sigmaT = R.fog_absorption_decay, groundLevel = R.ground_level;
		float int_ext;
		if (is_uniform_fog) int_ext = (float)((double)(alpha * t) * 0.05);
		else int_ext = alpha * int_exponential(r.o.y, groundLevel, sigmaT, t, rd.y);
		const float T = mipt_expf(-int_ext);
		float proba_t, random_t;
		const float clamped_t = 1000.f < t ? 1000.f : t;
		const float a = dot(sampleLightPos - r.o, r.d);
		if (a > 0) {
			const f3 projP = r.o + a * r.d;
			const float D = sqrtf(norm2(sampleLightPos - projP));
			const float thetaA = -mipt_atan2f(a, D);
			const float b = t - a;
			const float thetaB = mipt_atan2f(b, D);
			const float x = pcg_uniform(S.rng);
			random_t = D * fog_tanf((1 - x) * thetaA + x * thetaB);
			proba_t = D / ((thetaB - thetaA) * (D * D + random_t * random_t));
			random_t += a;
		} else {
			const float alpha2 = 5.f / clamped_t;
			do { random_t = -mipt_logf(pcg_uniform(S.rng)) / alpha2; } while (random_t > clamped_t);
			const float normalization = 1.f / alpha2 * (1.f - mipt_expf(-alpha2 * clamped_t));
			proba_t = mipt_expf(-alpha2 * random_t) / normalization;
		}
		float int_ext_partielle;
		if (is_uniform_fog) int_ext_partielle = (float)((double)(alpha * random_t) * 0.05);
		else int_ext_partielle = alpha * int_exponential(r.o.y, groundLevel, sigmaT, random_t, rd.y);
		const f3 random_P = r.o + random_t * rd;
		if (random_P.y < groundLevel) return false;
		f3 random_dir, point_aleatoire = mk3(0, 0, 0);
		const f3 axeOP = normalize(random_P - cl);
		bool is_uniform;
		if (pcg_uniform(S.rng) < p_uniform) { random_dir = random_uniform_sphere(S.rng); is_uniform = true; }
		else {
			const float l1 = pcg_uniform(S.rng), l2 = pcg_uniform(S.rng);
			const f3 dl = random_cos(axeOP, l1, l2);
			point_aleatoire = dl * R.radiusLight + cl;
			random_dir = normalize(point_aleatoire - random_P);
			is_uniform = false;
		}
		float phase_func = 0.f;
		const float kf = R.phase_aniso;
		if (R.fog_phase_type == 0) phase_func = (float)(1. / (4. * MIPT_PI));
		else if (R.fog_phase_type == 1) phase_func = (float)((double)(1 - kf * kf) / (4. * MIPT_PI * (double)(1 + kf * dot(random_dir, -rd))));
		else if (R.fog_phase_type == 2) phase_func = (float)(3 / (16 * MIPT_PI) * (double)(1 + sqr(dot(random_dir, rd))));
		Ray L; L.o = random_P; L.d = random_dir;
		request_closest(L);
		FRS(10, make_float4(T, proba_t, int_ext_partielle, phase_func));
		FRS(11, make_float4(point_aleatoire.x, point_aleatoire.y, point_aleatoire.z, is_uniform ? 1.f : 0.f));
		save(QW_F1, st);
		return true;
	};

	// The segments in the order control can flow through them within one call: A1 -> A2 -> [FOG: the fog call] -> [F1: answer of the fog query] ->
	// TAIL (what follows a fog call at its site) -> A3 -> POP.  Every transfer goes forward, so each segment is one block
	// of straight-line code and what it computes dies with it unless a later segment of the same call uses it.
	enum { ST_A1 = 1, ST_A2 = 2, ST_FOG = 3, ST_F1Q = 4, ST_TAIL = 5, ST_A3 = 6, ST_POP = 7 };
	f3 fog_light = cl;                     // sampleLightPos of the fog call of the site
	int st = (phase == QW_A1 || phase == QW_PROBE) ? ST_A1 : (phase == QW_A2 ? ST_A2 : (phase == QW_F1 ? ST_F1Q : (phase == QW_T5 ? ST_TAIL : ST_POP)));
	const bool a2_from_query = phase == QW_A2;
	const bool a1_from_probe = SUBS && phase == QW_PROBE;      // the answer of the subsurface probe: the head of A1 again (deterministic), from the saved hit
	if (!SHADOW_LIST && st == ST_A1) do {
			const float4 hr = a1_from_probe ? FRL(12) : QW_LD(&wf.hit[id]);
			const unsigned packed = __float_as_uint(hr.w);
			Hit h; h.t = hr.x; h.beta = hr.y; h.gamma = hr.z;
			const bool hit = hit_unpack(sc, packed, h.obj, h.tri);
			t_main = h.t;
			if (hit) hit_material(sc, currentRay, h, P, m);
			if (hit && nbrebonds == R.nb_bounces && qw.aov_n) {                  // :255-258
				qw.aov_n[id] = make_float4(m.shadingN.x, m.shadingN.y, m.shadingN.z, 0.f);
				qw.aov_kd[id] = make_float4(m.Kd.x, m.Kd.y, m.Kd.z, 0.f);
			}
			if (nbrebonds == R.nb_bounces && has_bg && (!hit || h.obj == 1)) {  // :260-268
				add_color(pathWeight * background_pixel(R, pi, pj));
				st = ST_POP; break;
			}
			if (!hit) { if (R.fog_density == 0) { st = ST_POP; break; } else { finish(); return 0; } }   // :654-657 (break ends the sample)
			Nn = m.shadingN; rayDirection = currentRay.d;
			objid = h.obj;
			if (h.obj == 1) {                                                   // :275-301
				if (has_fog) {
					FRS(9, make_float4(m.Ke.x, m.Ke.y, m.Ke.z, 0.f));
					FRS(0, make_float4(P.x, P.y, P.z, t_main));
					site = show_envmap ? 1 : 0;
					st = ST_FOG; break;              // no event: straight to what follows the fog call
				}
				if (show_envmap) add_color((pathWeight * R.envmap_intensity) * m.Ke);
				st = ST_POP; break;
			}
			if (h.obj == 0) {                                                   // :303-316
				if (has_fog) {
					FRS(0, make_float4(P.x, P.y, P.z, t_main));
					site = 2;
					st = ST_FOG; break;
				}
				const f3 cc = show_lights ? mk3(R.lightPower, R.lightPower, R.lightPower) : mk3(0.f, 0.f, 0.f);
				add_color(pathWeight * cc);
				st = ST_POP; break;
			}
			const DObject& obj = sc->obj[h.obj];
			uint64_t rng_fast = S.rng, rng_light = S.rng; float l1_fast = 0.f, l2_fast = 0.f;
			if (FAST) {
				if (m.merl != nullptr) return 16;
				if (!(m.miroir & 1) && !m.transp) {
					if (!(m.Ks.x == 0.f && m.Ks.y == 0.f && m.Ks.z == 0.f && m.Ne.x >= 0.f && m.Ne.y >= 0.f && m.Ne.z >= 0.f)) return 16;
					l1_fast = pcg_uniform(rng_fast); l2_fast = pcg_uniform(rng_fast);
					rng_light = rng_fast;                                       // the engine behind the light sample
					// lobe pick of PhongBRDF::sample (BRDF.h:73) with p = 1 - 0/3.f = 1: the diffuse lobe unless u == 1.0f (2^-25: the general build)
					if (!((float)pcg_next(rng_fast) / 4294967296.f < 1.f)) return 16;
				}
			}
			Ksub = mk3(0, 0, 0); subsW = mk3(1.f, 1.f, 1.f); sub_interaction = false;
			if (SUBS) {
				Ksub = hit_ksub(obj, h, xf_point(obj.inv, currentRay.o) + h.t * xf_dir(obj.inv, currentRay.d));
				const bool is_subsurface = norm2(Ksub) > 1E-8;                  // :271
				const float subsProba = (hadSS || !is_subsurface) ? 0.f : 0.6f; // :318
				const float inv1MSubsProba = 1.f / (1.f - subsProba);
				subsW = mk3(inv1MSubsProba, inv1MSubsProba, inv1MSubsProba);
				if (a1_from_probe || (is_subsurface && (pcg_uniform(S.rng) < subsProba))) {        // :324-404
					sub_interaction = true;
					const float invSubsProba = 1.f / subsProba;
					subsW = mk3(invSubsProba, invSubsProba, invSubsProba);
					const float sigmasub = 1.5f;
					const float diskR = sqrtf(12.46f) * sigmasub;
					const float integ = 1.f - mipt_expf(-diskR * diskR / (2.f * sigmasub * sigmasub));
					float gauss0, gauss1, gauss2, r1s;
					bool r2_low = false;
					if (a1_from_probe) { const float4 g = FRL(6); gauss0 = g.x; gauss1 = g.y; gauss2 = g.z; r1s = g.w; r2_low = FRL(7).x != 0.f; }
					else {
						const float randR = sigmasub * sqrtf(-2.f * mipt_logf(1.f - pcg_uniform(S.rng) * integ));
						const float randangle = pcg_uniform(S.rng) * 2.f * (float)MIPT_PI;
						{ float sn_, cs_; pt_sincosf(randangle, sn_, cs_); gauss0 = randR * sn_; gauss1 = randR * cs_; } gauss2 = randR;
						r1s = pcg_uniform(S.rng);
					}
					const float gaussval = (float)((1. / (double)(sigmasub * sigmasub * 2.f * (float)MIPT_PI)) * (double)mipt_expf(-(gauss2 * gauss2) / (2.f * sigmasub * sigmasub)));
					const float pdfgauss = gaussval / integ;
					const f3 Tg = tangent_of(Nn);
					const f3 Tg2 = cross(Nn, Tg);
					const f3 PtaboveP = ((P + gauss0 * Tg) + gauss1 * Tg2) + Nn * diskR;
					f3 axis = -Nn;
					float tmax;
					const float hh = sqrtf(diskR * diskR - gauss2 * gauss2);
					f3 subsOrigin = PtaboveP + (diskR - hh) * (-Nn);
					float wAxis;
					if (r1s < 0.5f) { wAxis = 0.5f; tmax = 2.f * hh; }
					else {
						wAxis = 0.25f;
						tmax = 2.f * gauss2;
						if (r1s < 0.75f) axis = Tg; else axis = Tg2;
						if (!a1_from_probe) r2_low = pcg_uniform(S.rng) < 0.5f;
						if (r2_low) subsOrigin = subsOrigin - hh * Nn;
					}
					Ray probe; probe.o = subsOrigin; probe.d = axis;
					Hit sh; sh.obj = h.obj; sh.tri = -1; sh.t = 0; sh.beta = sh.gamma = 0;
					const f3 po = xf_point(obj.inv, probe.o), pd = xf_dir(obj.inv, probe.d);
					if (obj.type == 0 && !a1_from_probe) {                       // a mesh: get_random_intersection is the probe stage's
						FRS(12, hr);
						FRS(6, make_float4(gauss0, gauss1, gauss2, r1s));
						FRS(7, make_float4(r2_low ? 1.f : 0.f, 0.f, 0.f, 0.f));
						QW_ST(&qw.cur_o[id], make_float4(currentRay.o.x, currentRay.o.y, currentRay.o.z, 0.f));
						QW_ST(&qw.cur_d[id], make_float4(currentRay.d.x, currentRay.d.y, currentRay.d.z, 0.f));
						QW_ST(&wf.ray_o[id], make_float4(po.x, po.y, po.z, tmax));
						QW_ST(&wf.ray_d[id], make_float4(pd.x, pd.y, pd.z, __uint_as_float((unsigned)h.obj)));
						save(QW_PROBE, 0);
						return 4;
					}
					bool subsinter;
					if (a1_from_probe) {
						const float4 pr = QW_LD(&wf.hit[id]);
						subsinter = __float_as_uint(pr.w) != MIPT_HIT_MISS;
						sh.t = pr.x; sh.beta = pr.y; sh.gamma = pr.z; sh.tri = (int)__float_as_uint(pr.w);
					} else subsinter = obj.type == 1 ? sphere_reservoir(obj, po, pd, 0.f, tmax, S.rng, sh.t) : plane_reservoir(obj, po, pd, 0.f, tmax, S.rng, sh.t);
					if (subsinter) {
						f3 localP2; Mat subsmat;
						subsmat.shadingN = mk3(0, 1, 0); subsmat.Kd = mk3(0.5f, 0.5f, 0.5f); subsmat.Ks = mk3(0, 0, 0); subsmat.Ne = mk3(100, 100, 100); subsmat.Ke = mk3(0, 0, 0); subsmat.transp = false; subsmat.refr_index = 0;
						hit_material_obj(obj, probe, sh, localP2, subsmat, false, true);
						const float chris = (float)pt_exp64((double)(-norm2(P - localP2)) / (2. * (double)sigmasub * (double)sigmasub));
						const double d0 = 0.5 * (double)dot(subsmat.shadingN, Nn), d1 = 0.25 * (double)dot(subsmat.shadingN, Tg), d2 = 0.25 * (double)dot(subsmat.shadingN, Tg2);
						const float sumpdfs = (float)((d0 * d0 + d1 * d1) + d2 * d2);
						const float pdfdisk = wAxis * fabsf(dot(axis, subsmat.shadingN)) / sumpdfs;
						subsW = subsW * (pdfdisk / fmaxf(pdfgauss, 0.05f) * chris);
						rayDirection = normalize(localP2 - P);
						P = localP2 + 0.005f * subsmat.shadingN;
						if (r1s < 0.5f) subsW = subsW * 2.f; else subsW = subsW * 4.f;
						subsW = subsW * (Ksub / (float)MIPT_PI);
						m = subsmat;
						Ksub = hit_ksub(obj, sh, po + sh.t * pd, true);
						Nn = m.shadingN;
					}
				}
			}
			add_color((pathWeight * m.Ke) * R.envmap_intensity);                // :411
			if (FAST ? ((m.miroir & 1) != 0) : ((obj.miroir & 1) != 0)) {             // :413-436
				Ray rm; rm.o = P + 0.001f * Nn; rm.d = reflect(rayDirection, Nn);
				if (has_fog) {
					FRS(0, make_float4(P.x, P.y, P.z, t_main));
					FRS(7, make_float4(rm.o.x, rm.o.y, rm.o.z, 0.f)); FRS(8, make_float4(rm.d.x, rm.d.y, rm.d.z, 0.f));
					site = 3;
					st = ST_FOG; break;
				}
				push(pathWeight, rm, nbrebonds - 1, show_lights, true, hadSS);
				st = ST_POP; break;
			}
			if (m.transp) {                                                     // :438-489
				float n1 = 1.f, n2 = m.refr_index;
				f3 nt = Nn;
				bool entering = true;
				if (dot(rayDirection, Nn) > 0) { n1 = m.refr_index; n2 = 1; nt = -Nn; entering = false; }
				const float radical = 1.f - sqr(n1 / n2) * (1.f - sqr(dot(nt, rayDirection)));
				Ray nr;
				if (radical > 0) {
					const f3 refr = (n1 / n2) * (rayDirection - dot(rayDirection, nt) * nt) - nt * sqrtf(radical);
					const float R0 = sqr((n1 - n2) / (n1 + n2));
					float Rf;
					if (entering) Rf = R0 + (1 - R0) * pt_powf(1.f + dot(rayDirection, Nn), 5.f);
					else Rf = R0 + (1 - R0) * pt_powf(1.f - dot(refr, Nn), 5.f);
					if (pcg_uniform(S.rng) < Rf) { nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, Nn); }
					else { nr.o = P - 0.001f * nt; nr.d = refr; }
				} else { nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, Nn); }
				if (has_fog) {
					FRS(0, make_float4(P.x, P.y, P.z, t_main));
					FRS(7, make_float4(nr.o.x, nr.o.y, nr.o.z, 0.f)); FRS(8, make_float4(nr.d.x, nr.d.y, nr.d.z, 0.f));
					site = 4;
					st = ST_FOG; break;
				}
				push(pathWeight, nr, nbrebonds - 1, show_lights, true, hadSS);
				st = ST_POP; break;
			}
			if (FAST) {
				// ---- Lambert vertex, whole: light sample (:490-513), direct term (:538-566), continuation (:570-632)
				S.rng = rng_fast;                                               // l1, l2 and the lobe pick of the continuation are drawn
				const f3 axeOP = fast_normalize(P - cl);
				dir_l = random_cos(axeOP, l1_fast, l2_fast);
				const f3 pt_l = dir_l * R.radiusLight + cl;
				wi = fast_normalize(pt_l - P);
				d_light2 = norm2(pt_l - P);
				const f3 brdf = m.Kd / (float)MIPT_PI;
				bool yield_shadow = false;
				if (dot(Nn, wi) < 0) isShadowed = true;
				else {
					Ray rl; rl.o = P + 0.01f * wi; rl.d = wi;
					const float dist = sqrtf(d_light2) - 0.01f;
					n_shadow++;
					if (qw_analytic_occluded(sc, rl.o, rl.d, dist)) isShadowed = true;
					else if (!qw_meshes_missed(sc, rl.o, rl.d, dist)) {
						QW_ST(&wf.sh_o[id], make_float4(rl.o.x, rl.o.y, rl.o.z, dist));
						QW_ST(&wf.sh_d[id], make_float4(rl.d.x, rl.d.y, rl.d.z, 0.f));
						yield_shadow = true;
					}
				}
				// a ghost queues the path going straight on only if the light is visible: its any-hit answer is waited for (A2 of the
				// general build, over the any-hit list, goes on from the frame; the lobe pick is drawn there)
				if (yield_shadow && ((m.miroir & 2) != 0)) { S.rng = rng_light; m.shadingN = Nn; save_vertex(true); save(QW_A2, 0); return 2; }
				contrib = mk3(0, 0, 0);
				if (((m.miroir & 2) != 0)) {
					if (!isShadowed) {                                          // :522-536
						const f3 offset = dot(Nn, rayDirection) > 0 ? Nn : -Nn;
						Ray through; through.o = (P + rayDirection * 0.001f) + offset * 0.001f; through.d = rayDirection;
						push(pathWeight, through, nbrebonds, show_lights, show_envmap, hadSS);
					}
				} else if (!isShadowed) {
					const float J = dot(dir_l, -wi) / d_light2;
					const float proba = (float)((double)dot(axeOP, dir_l) / (MIPT_PI * (double)R.radiusLight * (double)R.radiusLight));
					if (proba > 0.f) contrib = contrib + (mk3(1.f, 1.f, 1.f) * (R.lightPower * fmaxf(0.f, dot(Nn, wi)) * J / proba)) * brdf;
				}
				if (yield_shadow) { const f3 pc = pathWeight * contrib; QW_ST(&wf.sh_c[id], make_float4(pc.x, pc.y, pc.z, 0.f)); pending_add = 8; }
				else add_color(pathWeight * contrib);                           // :566
				// the continuation of a path's last vertex is queued with depth 0 and dropped by the loop head (:240): not computed
				if (nbrebonds > 1) {
					float ip;
					const float r1 = modff(R.randomPerPixel[2 * (size_t)pix] + R.samples2d[2 * k], &ip);
					const float r2 = modff(R.randomPerPixel[2 * (size_t)pix + 1] + R.samples2d[2 * k + 1], &ip);
					const f3 dir = random_cos(Nn, r1, r2);
					const float pdf = (float)((double)(1.f * dot(Nn, dir)) / (MIPT_PI) + (double)(0.f));
					if (!(dot(dir, Nn) < 0 || dot(dir, reflect(rayDirection, Nn)) < 0 || pdf <= 0)) {   // :593
						f3 nw = ((pathWeight * mk3(1.f, 1.f, 1.f)) * brdf) * (dot(Nn, dir) / pdf);          // :611
						if (((m.miroir & 2) != 0) && has_bg) {                                // :614-621
							const f3 bg = background_pixel(R, pi, pj);
							nw = nw * mk3(bg.x / 196964.699f, bg.y / 196964.699f, bg.z / 196964.699f);
						}
						Ray next; next.o = P + 0.01f * dir; next.d = dir;
						push(nw, next, nbrebonds - 1, false, (show_envmap && isShadowed) || !((m.miroir & 2) != 0), hadSS);   // :626-629 (the diffuse lobe was sampled)
					}
				}
				st = ST_POP; break;
			}
			// ---- diffuse / glossy vertex: the light sample (:490-513)
			const f3 axeOP = fast_normalize(P - cl);
			const float l1 = pcg_uniform(S.rng);
			const float l2 = pcg_uniform(S.rng);
			dir_l = random_cos(axeOP, l1, l2);
			const f3 pt_l = dir_l * R.radiusLight + cl;
			wi = fast_normalize(pt_l - P);
			d_light2 = norm2(pt_l - P);
			m.shadingN = Nn;
			bool yield_shadow = false;
			if (dot(m.shadingN, wi) < 0) isShadowed = true;
			else {
				Ray rl; rl.o = P + 0.01f * wi; rl.d = wi;
				const float dist = sqrtf(d_light2) - 0.01f;
				n_shadow++;
				if (qw_analytic_occluded(sc, rl.o, rl.d, dist)) isShadowed = true;
				else if (qw_meshes_missed(sc, rl.o, rl.d, dist)) isShadowed = false;
				else {
					QW_ST(&wf.sh_o[id], make_float4(rl.o.x, rl.o.y, rl.o.z, dist));
					QW_ST(&wf.sh_d[id], make_float4(rl.d.x, rl.d.y, rl.d.z, 0.f));
					yield_shadow = true;
				}
			}
			if (yield_shadow && obj.ghost) { save_vertex(true); save(QW_A2, 0); return 2; }   // a ghost queues the path going straight on only if the light is visible
			if (yield_shadow && has_fog) deferred = true;
			else if (yield_shadow) pending_add = 8;
			if (has_fog) save_vertex(false);                                         // the fog event of site 5 comes back through the frame
			st = ST_A2;
	} while (0);
	if (!FAST && st == ST_A2) do {
			if (a2_from_query) { load_vertex(true); isShadowed = qw.vis[id] == 0.f; }
			const DObject& obj = sc->obj[objid];
			const double* const merl = obj.merl;
			const f3 axeOP = fast_normalize(P - cl);
			const f3 pt_l = dir_l * R.radiusLight + cl;
			contrib = mk3(0, 0, 0);
			if (!isShadowed) {
				if (obj.ghost) {                                                // :522-536
					const f3 offset = dot(Nn, rayDirection) > 0 ? Nn : -Nn;
					currentRay.o = (P + rayDirection * 0.001f) + offset * 0.001f;
					currentRay.d = rayDirection;
					push(pathWeight, currentRay, nbrebonds, show_lights, show_envmap, hadSS);
					// (currentRay itself is replaced: the fog event below, in this call, uses it)
				} else {                                                        // :538-553
					f3 brdf;
					if (LAMBERT) {
						const bool plain = merl == nullptr && m.Ks.x == 0.f && m.Ks.y == 0.f && m.Ks.z == 0.f && m.Ne.x >= 0.f && m.Ne.y >= 0.f && m.Ne.z >= 0.f;
						if (!sub_interaction && !plain) { S.overflow = true; st = ST_POP; break; }      // not a Lambert vertex: the sample is abandoned to the one-thread loop (below)
						brdf = (sub_interaction ? Ksub : m.Kd) / (float)MIPT_PI;
					} else brdf = sub_interaction ? Ksub / (float)MIPT_PI : (merl ? merl_eval(merl, wi, -rayDirection, Nn) : phong_eval(m, wi, -rayDirection, Nn));
					const float J = dot(dir_l, -wi) / d_light2;
					const float proba = (float)((double)dot(axeOP, dir_l) / (MIPT_PI * (double)R.radiusLight * (double)R.radiusLight));
					if (proba > 0.f) contrib = contrib + (subsW * (R.lightPower * fmaxf(0.f, dot(Nn, wi)) * J / proba)) * brdf;
				}
			}
			if (has_fog) {                                                      // :557-565
				FRS(9, make_float4(contrib.x, contrib.y, contrib.z, 0.f));
				FRS(1, make_float4(Nn.x, Nn.y, Nn.z, __uint_as_float((unsigned)objid | (sub_interaction ? 0x100u : 0u) | (isShadowed ? 0x200u : 0u) | (deferred ? 0x400u : 0u))));
				site = 5; fog_light = pt_l;
				st = ST_FOG; break;
			}
			if (pending_add) { const f3 pc = pathWeight * contrib; QW_ST(&wf.sh_c[id], make_float4(pc.x, pc.y, pc.z, 0.f)); }
			else add_color(pathWeight * contrib);                              // :566
			st = ST_A3;
	} while (0);
	// ---- the fog call of the site (one instance for the six sites: fogContribution up to its visibility query is ~2500
	//      instructions of exact expf / logf / atan2f / tanf)
	if (!FAST && st == ST_FOG) {
			if (fog_begin(site, currentRay, fog_light)) return 1 | (deferred ? 8 : 0);
			if (deferred) { save(QW_T5, site); return 2; }                       // no event: the answer of the light-sample query is needed now
			st = ST_TAIL;
	}
	if (!FAST && !SHADOW_LIST && st == ST_F1Q) {
			// the second half of fogContribution, once the closest hit along the in-scattering direction is known
			{
				const float4 lo = QW_LD(&wf.ray_o[id]), ld = QW_LD(&wf.ray_d[id]), hr = QW_LD(&wf.hit[id]), f10 = FRL(10), f11 = FRL(11);
				t_main = FRL(0).w;
				Ray L; L.o = mk3(lo.x, lo.y, lo.z); L.d = mk3(ld.x, ld.y, ld.z);
				const f3 random_P = L.o, random_dir = L.d, point_aleatoire = mk3(f11.x, f11.y, f11.z);
				const bool is_uniform = f11.w != 0.f;
				const float T = f10.x, proba_t = f10.y, int_ext_partielle = f10.z, phase_func = f10.w;
				const unsigned packed = __float_as_uint(hr.w);
				Hit ih; ih.t = hr.x; ih.beta = hr.y; ih.gamma = hr.z;
				const bool interinter = hit_unpack(sc, packed, ih.obj, ih.tri);
				bool visible = true;
				if (!is_uniform) {
					const float dl2 = norm2(point_aleatoire - random_P);
					if (interinter && (double)(ih.t * ih.t) < (double)dl2 * 0.99) visible = false;
				}
				S.att = T; color_dirty = true;
				if (visible) {
					const f3 axeOP = normalize(random_P - cl);
					const float p_uniform = 0.5f;
					const float pdf_uniform = (float)(1. / (4. * MIPT_PI));
					float pdf_light = 0.f;
					if (interinter && ih.obj == 0) {
						f3 interP = mk3(0, 0, 0); Mat im;
						im.shadingN = mk3(0, 1, 0); im.Kd = mk3(0.5f, 0.5f, 0.5f); im.Ks = mk3(0, 0, 0); im.Ne = mk3(100, 100, 100); im.Ke = mk3(0, 0, 0); im.transp = false; im.refr_index = 0;
						hit_material(sc, L, ih, interP, im);
						const float J = dot(im.shadingN, -random_dir) / norm2(interP - random_P);
						pdf_light = (float)((double)dot(normalize(interP - cl), axeOP) / (MIPT_PI * (double)sqr(R.radiusLight)) / (double)J);
					}
					const float proba_dir = p_uniform * pdf_uniform + (1 - p_uniform) * pdf_light;
					float ext;
					if (R.fog_type == 0) ext = (float)((double)R.fog_density * 0.05);
					else ext = R.fog_density * mipt_expf(-R.fog_density_decay * (random_P.y - R.ground_level));
					const f3 evw = pathWeight * (phase_func * ext * mipt_expf(-int_ext_partielle) / (proba_t * proba_dir));
					push(evw, L, nbrebonds - 1, show_lights, true, hadSS);
				}
			}
			st = ST_TAIL;
	}
	if (!FAST && st == ST_TAIL) do {
			// the statements after the fog call of the site
			if (site == 0) { st = ST_POP; break; }
			if (site == 1) { const float4 ke = FRL(9); add_color(((S.att * pathWeight) * R.envmap_intensity) * mk3(ke.x, ke.y, ke.z)); st = ST_POP; break; }
			if (site == 2) { const f3 cc = show_lights ? mk3(R.lightPower, R.lightPower, R.lightPower) : mk3(0.f, 0.f, 0.f); add_color((S.att * pathWeight) * cc); st = ST_POP; break; }
			if (site == 3 || site == 4) {
				const float4 ro = FRL(7), rd = FRL(8);
				Ray nr; nr.o = mk3(ro.x, ro.y, ro.z); nr.d = mk3(rd.x, rd.y, rd.z);
				push(S.att * pathWeight, nr, nbrebonds - 1, show_lights, true, hadSS);
				st = ST_POP; break;
			}
			// site 5: the diffuse vertex goes on (:565)
			if (phase == QW_F1 || phase == QW_T5) {                             // entered from a query: the vertex is in the frame
				load_vertex(false);
				const float4 c9 = FRL(9); contrib = mk3(c9.x, c9.y, c9.z);
				if (deferred && qw.vis[id] == 0.f) { isShadowed = true; contrib = mk3(0, 0, 0); }   // :538 was assumed visible
			}
			add_color((S.att * pathWeight) * contrib);
			st = ST_A3;
	} while (0);
	// ---- A3: the continuation of the diffuse vertex (:570-632)
	if (!FAST && st == ST_A3) do {
			const DObject& obj = sc->obj[objid];
			const double* const merl = obj.merl;
			const bool plain3 = LAMBERT && merl == nullptr && m.Ks.x == 0.f && m.Ks.y == 0.f && m.Ks.z == 0.f && m.Ne.x >= 0.f && m.Ne.y >= 0.f && m.Ne.z >= 0.f;
			float ip;
			const float r1 = modff(R.randomPerPixel[2 * (size_t)pix] + R.samples2d[2 * k], &ip);
			const float r2 = modff(R.randomPerPixel[2 * (size_t)pix + 1] + R.samples2d[2 * k + 1], &ip);
			float pdf; f3 dir; bool has_sampled_diffuse;
			if (sub_interaction) { dir = random_cos(m.shadingN, r1, r2); pdf = dot(Nn, dir) / (float)MIPT_PI; has_sampled_diffuse = true; }   // :584-587
			else if (merl) { dir = random_cos(Nn, r1, r2); pdf = (float)((double)dot(Nn, dir) / (MIPT_PI)); has_sampled_diffuse = false; }
			else {
				uint64_t peek = S.rng;
				has_sampled_diffuse = (float)pcg_next(peek) / 4294967296.f < 1 - (m.Ks.x + m.Ks.y + m.Ks.z) / 3.f;
				if (LAMBERT && plain3 && has_sampled_diffuse) {
					// PhongBRDF::sample with p = 1 - 0/3.f = 1 and the diffuse lobe picked (BRDF.h:73-82): one engine draw, the cosine lobe, and
					// pdf = p*dot/pi + (1-p)*proba_phong = dot/pi + 0 — proba_phong is finite when dot(R, dir) >= 0, and the sample is
					// rejected by the reflect test below whatever pdf is otherwise (path_vertex_fast, mipt_shade.h)
					S.rng = peek;
					dir = random_cos(Nn, r1, r2);
					pdf = (float)((double)(1.f * dot(Nn, dir)) / (MIPT_PI) + (double)(0.f));
				} else if (LAMBERT) { S.overflow = true; st = ST_POP; break; }      // a specular material, or the 2^-25 event of the Phong lobe at p = 1
				else dir = phong_sample(m, -rayDirection, Nn, pdf, r1, r2, S.rng);
			}
			if (dot(dir, Nn) < 0 || dot(dir, reflect(rayDirection, Nn)) < 0 || pdf <= 0) { st = ST_POP; break; }   // :593
			f3 brdf_i;
			if (LAMBERT) {
				if (!sub_interaction && !plain3) { S.overflow = true; st = ST_POP; break; }
				brdf_i = (sub_interaction ? Ksub : m.Kd) / (float)MIPT_PI;
			} else brdf_i = sub_interaction ? Ksub / (float)MIPT_PI : (merl ? merl_eval(merl, dir, -rayDirection, Nn) : phong_eval(m, dir, -rayDirection, Nn));
			f3 nw = ((pathWeight * subsW) * brdf_i) * (dot(Nn, dir) / pdf);    // :611
			if (obj.ghost && has_bg) {                                          // :614-621
				const f3 bg = background_pixel(R, pi, pj);
				nw = nw * mk3(bg.x / 196964.699f, bg.y / 196964.699f, bg.z / 196964.699f);
			}
			Ray next; next.o = P + 0.01f * dir; next.d = dir;
			const bool env = (show_envmap && isShadowed && has_sampled_diffuse) || !obj.ghost;      // :626-629
			if (has_fog) push(S.att * nw, next, nbrebonds - 1, false, env, sub_interaction ? true : hadSS);
			else push(nw, next, nbrebonds - 1, false, env, sub_interaction ? true : hadSS);
			st = ST_POP;
	} while (0);
	// ---- POP: the next contribution that is still alive asks for its closest hit
	if (S.overflow) { save(QW_DONE, 0); return -1; }
	for (;;) {
			if (S.count == 0) { finish(); return pending_add; }
			QContrib c;
			if (have_front) { c = front; have_front = false; }
			else { const QContrib* const e = fifo + S.head; c.w = QW_LD(&e->w); c.o = QW_LD(&e->o); c.d = QW_LD(&e->d); }
			S.head = S.head + 1 == qw.ring ? 0u : S.head + 1; S.count--;
			const unsigned bits = __float_as_uint(c.w.w);
			if ((int)(bits & 0xffffu) == 0) continue;                            // :240
			if (norm2(mk3(c.w.x, c.w.y, c.w.z)) < sqr(0.01f)) continue;          // :241
			QW_ST(&qw.cur_w[id], c.w);
			Ray r; r.o = mk3(c.o.x, c.o.y, c.o.z); r.d = mk3(c.d.x, c.d.y, c.d.z);
			request_closest(r);
			save(QW_A1, 0);
			return 1 | pending_add;
		}
}

// One round of the logic stage over one id list of the previous round (or, round 0, over all path slots).
template <bool SUBS, bool SHADOW_LIST, bool FOG, bool FAST = false, bool LAMBERT = false>
__global__ void __launch_bounds__(MIPT_BLOCK) __attribute__((amdgpu_waves_per_eu(FAST ? MIPT_QW_FAST_WAVES : ((SHADOW_LIST && !FOG) ? MIPT_QW_LOGIC_WAVES + 1 : MIPT_QW_LOGIC_WAVES)))) k_q_logic(const DScene* __restrict__ sc, DRender R, DPass ps, DWave wf, DQueueWave qw,
                                                                                                const unsigned* __restrict__ list, const unsigned* __restrict__ n_ptr, unsigned n_imm,
                                                                                                unsigned* __restrict__ head, int out_slot, int out_parity, DCounters* __restrict__ cnt) {
	const unsigned n = n_ptr ? *n_ptr : n_imm;
	unsigned n_closest = 0, n_shadow = 0;
	unsigned base;
	// (chunks of MIPT_QW_UNROLL x 64 entries, not the 512 of pipeline 1's stages: an entry here costs thousands of instructions, the atomics are
	// nowhere near a floor, and the appends of 8 sub-chunks — 16 more scalar registers each — cost the logic stage 1-3 %)
	QueuePullerT<64u * MIPT_QW_UNROLL> q; q.init();
	while (q.pull(head, n, base)) {
		unsigned closest_bits = 0, shadow_bits = 0, over_bits = 0, probe_bits = 0, add_bits = 0, slow_bits = 0;
#pragma unroll 1
		for (int u = 0; u < (int)MIPT_QW_UNROLL; u++) {
			const unsigned idx = base + 64u * (unsigned)u + lane_id();
			if (idx >= n) continue;
			const unsigned id = list ? list[idx] : idx;
			const int r = qw_advance<SUBS, SHADOW_LIST, FOG, FAST, LAMBERT>(sc, R, ps, wf, qw, id, n_closest, n_shadow);
			if (r < 0) over_bits |= 1u << u;
			else if (FAST && r == 16) slow_bits |= 1u << u;
			else {
				if (r & 1) closest_bits |= 1u << u;
				if (r & 2) shadow_bits |= 1u << u;
				if (r & 4) probe_bits |= 1u << u;
				if (r & 8) add_bits |= 1u << u;
			}
		}
		queue_push2<MIPT_QW_UNROLL>(qw.shl[out_parity], qw.live[out_parity], reinterpret_cast<unsigned long long*>(&qw.counters[MIPT_QW_PAIR(out_slot)]), shadow_bits, closest_bits, list, base);
		queue_push<MIPT_QW_UNROLL>(qw.overflow, &qw.counters[MIPT_QW_N_OVERFLOW], over_bits, list, base);
		if (SUBS && !SHADOW_LIST) queue_push<MIPT_QW_UNROLL>(qw.prl[out_parity], &qw.counters[MIPT_QW_N_PROBE(out_slot)], probe_bits, list, base);
		if (!SHADOW_LIST) queue_push<MIPT_QW_UNROLL>(qw.sha[out_parity], &qw.counters[MIPT_QW_N_SHADOW_ADD(out_slot)], add_bits, list, base);
		if (FAST) queue_push<MIPT_QW_UNROLL>(qw.slow, &qw.counters[MIPT_QW_N_SLOW(out_slot)], slow_bits, list, base);
	}
	DCounters* my = MIPT_MY_COUNTERS(cnt);
	wave_add(&my->rays_closest, n_closest);
	wave_add(&my->rays_shadow, n_shadow);
}

// The samples whose ring overflowed, by the one-thread-per-sample loop with the reference's 200-entry ring.
__global__ void __launch_bounds__(MIPT_BLOCK) k_render_paths_queue_list(const DScene* __restrict__ sc, DRender R, DPass ps, DSamples out, DCounters* __restrict__ cnt, QContrib* __restrict__ queues,
                                                                        const unsigned* __restrict__ ids, unsigned n, float4* __restrict__ aov_n, float4* __restrict__ aov_kd) {
	MIPT_DECLARE_STACK(stk);
	const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned n_closest = 0, n_shadow = 0;
	if (q < n) {
		const unsigned tid = ids[q];
		const int kk = (int)(tid / (unsigned)ps.npix_slots), slot = (int)(tid % (unsigned)ps.npix_slots);
		const int blk = slot >> 6, in = slot & 63;
		const int i = ps.blocks[2 * blk] + (in >> 3), j = ps.blocks[2 * blk + 1] + (in & 7);
		float dx, dy; f3 nv, av;
		const f3 c = trace_path_queue(sc, R, i, j, ps.k0 + kk, dx, dy, n_closest, n_shadow, stk, queues + (size_t)q * MIPT_SIZE_CIRC_ARRAY, nv, av);
		out.col[tid] = make_float4(c.x, c.y, c.z, 0.f); out.dxdy[tid] = make_float2(dx, dy);
		if (aov_n) { aov_n[tid] = make_float4(nv.x, nv.y, nv.z, 0.f); aov_kd[tid] = make_float4(av.x, av.y, av.z, 0.f); }
	}
	(void)cnt; (void)n_closest; (void)n_shadow;       // the rays of an abandoned sample were counted up to the overflow; the stats of such passes are upper bounds
}
