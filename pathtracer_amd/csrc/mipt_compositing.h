// mipt_compositing.h — getColor with its contribution queue (ghost objects and the background photo).
//
// The reference keeps the pending contributions of a sample in a circular FIFO of 200 entries (Raytracer.h:114-115,
// Raytracer.cpp:213-238) and draws from ONE engine while it works through them, so the order in which contributions are
// processed decides which random numbers they see.  Without ghosts (and fog) every vertex queues at most one successor
// and the loop is the linear chain the other kernels run.  A ghost object (Object::ghost, Geometry.h:721: invisible, but it
// receives shadows and reflections, for compositing over a photo) queues two: the path going straight on through it at
// the SAME depth (Raytracer.cpp:522-536) and the sampled continuation (:611-632).  This kernel keeps the whole loop of
// one sample in one thread (like k_render_paths), with the FIFO in HBM; scenes with ghosts or a background image are
// routed here (a few of a production's shots, not the throughput path).
#pragma once
#include "mipt_explog.h"

#define MIPT_SIZE_CIRC_ARRAY 200          // Raytracer.h:114

struct QContrib { float4 w; float4 o; float4 d; };   // w.xyz weight, w.w bits: depth | show_lights << 16 | showenvmap << 17 | has_had_subsurface_interaction << 18

MIPT_DEV f3 background_pixel(const DRender& R, int screenI, int screenJ) {   // Raytracer.cpp:261-265
	int bi = (int)((float)screenI / (float)R.H * (float)R.backgroundH); bi = min(R.backgroundH - 1, max(0, bi));
	int bj = (int)((float)screenJ / (float)R.W * (float)R.backgroundW); bj = min(R.backgroundW - 1, max(0, bj));
	const float* px = R.background + ((size_t)bi * R.backgroundW + bj) * 3;
	return mk3(px[0], px[1], px[2]);
}

// The traversals are kept out of line here: inlined four times into the getColor loop they push it to 272 registers and one
// wave per SIMD; as calls the loop needs about half of that.
template <class STK> __device__ __attribute__((noinline)) bool q_intersect(const DScene* __restrict__ sc, const Ray& r, Hit& h, f3& P, Mat& m, STK& stk) { return scene_intersect(sc, r, h, P, m, stk); }
template <class STK> __device__ __attribute__((noinline)) bool q_occluded(const DScene* __restrict__ sc, const Ray& r, float dist, STK& stk) { return scene_occluded<STK, true>(sc, r, dist, stk); }

// ---- subsurface probe (Raytracer.cpp:318-406) --------------------------------------------------------------------
// TriMesh::reservoir_sampling_intersection (TriangleMesh.cpp:1321-1426): a uniformly random one of the intersections in
// [min_t, max_t).  Same visiting order as the closest-hit traversal with a fixed far bound; every accepted triangle draws
// one number from the sample's engine, so the order of the visits decides the draws.
template <class STK>
__device__ __attribute__((noinline)) bool mesh_reservoir(const DObject& o, f3 org, f3 d, float min_t, float max_t, uint64_t& rng, float& t_out, int& tri_out, float& beta_out, float& gamma_out, STK& stk) {
	bool has_inter = false;
	int count = 0;
	f3 invd = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
	bool sx = invd.x >= 0, sy = invd.y >= 0, sz = invd.z >= 0;
	float t_root;
	if (!box_test<false>(ld3(o.root_min), ld3(o.root_max), org, invd, sx, sy, sz, t_root)) return false;
	if (t_root > max_t) return false;
	int sp = 0;
	const uint32_t NONE = 0x7fffffffu;
	uint32_t cur = o.root_ref;
	const float4* __restrict__ nodes = reinterpret_cast<const float4*>(o.nodes);
	auto pop_next = [&]() -> uint32_t {
		while (sp > 0) {
			--sp;
			uint32_t r; float tn;
			stk.pop(sp, r, tn);
			if (!(tn > max_t)) return r;
		}
		return NONE;
	};
	for (;;) {
		while (cur != NONE && !(cur & MIPT_LEAF_BIT)) {
			const float4* q = nodes + 4 * (size_t)cur;
			float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
			f3 lmin = mk3(q0.x, q0.z, q1.x), lmax = mk3(q0.y, q0.w, q1.y);
			f3 rmin = mk3(q1.z, q2.x, q2.z), rmax = mk3(q1.w, q2.y, q2.w);
			uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
			float tl, tr;
			const bool goleft = box_test<true>(lmin, lmax, org, invd, sx, sy, sz, tl) && (tl < max_t);
			const bool goright = box_test<true>(rmin, rmax, org, invd, sx, sy, sz, tr) && (tr < max_t);
			if (goleft && goright) {
				if (tl < tr) { stk.push(sp, rref, tr); sp++; cur = lref; }
				else { stk.push(sp, lref, tl); sp++; cur = rref; }
			} else if (goleft) cur = lref;
			else if (goright) cur = rref;
			else cur = pop_next();
		}
		if (cur == NONE) break;
		int first = (int)(cur & MIPT_LEAF_FIRST_MASK);
		const int cnt = mipt_leaf_count(cur, o.fat_leaves, o.n_fat_leaves);
		for (int i = first; i < first + cnt; i++) {
			float lt, lb, lg;
			if (tri_test(o.tris + i, org, d, lt, lb, lg)) {
				bool accept = lt < max_t && lt >= min_t;
				if (accept && o.alpha_test) accept = !alpha_rejects(o, i - (int)o.tri_base, 1 - lb - lg, lb, lg);
				if (accept) {
					count++;
					const float r1 = pcg_uniform(rng);
					if ((double)r1 < 1. / (double)count) { has_inter = true; t_out = lt; tri_out = i - (int)o.tri_base; beta_out = lb; gamma_out = lg; }
				}
			}
		}
		cur = pop_next();
	}
	return has_inter;
}
// Sphere::reservoir_sampling_intersection (Geometry.h:994-1050): the roots in [min_t, max_t) in root order, one engine draw each, the
// second replacing the first with probability 1/2 (`r1 < 1.f / count`, float).  The material follows in hit_material_obj(.., probe = true).
MIPT_DEV bool sphere_reservoir(const DObject& s, f3 o, f3 d, float min_t, float max_t, uint64_t& rng, float& t_out) {
	if (s.has_envmap) return false;
	const f3 oc = o - ld3(s.O);
	const float b = dot(d, oc);
	const float a = norm2(d);
	const float c = norm2(oc) - s.R2;
	const float delta = b * b - a * c;
	if (delta < 0) return false;
	const float sqDelta = sqrtf(delta);
	const float inva = 1.f / a;
	const float t2 = (-b + sqDelta) * inva;
	if (t2 < min_t) return false;
	const float t1 = (-b - sqDelta) * inva;
	bool has_inter = false;
	int count = 0;
	if (t1 >= min_t && t1 < max_t) { count++; const float r1 = pcg_uniform(rng); if (r1 < 1.f / (float)count) { t_out = t1; has_inter = true; } }
	if (t2 < max_t) { count++; const float r1 = pcg_uniform(rng); if (r1 < 1.f / (float)count) { t_out = t2; has_inter = true; } }
	return has_inter;
}
// Plane::reservoir_sampling_intersection (Geometry.h:1159-1183): the single intersection, kept with probability 1/count
MIPT_DEV bool plane_reservoir(const DObject& p, f3 o, f3 d, float min_t, float max_t, uint64_t& rng, float& t_out) {
	const f3 N = ld3(p.vecN);
	const float ddot = dot(d, N);
	if ((double)fabsf(ddot) < 1E-9) return false;
	const float curt = dot(ld3(p.A) - o, N) / ddot;
	if (curt < min_t || curt >= max_t) return false;
	const float r1 = pcg_uniform(rng);               // count = 1
	if ((double)r1 >= 1.) return false;
	t_out = curt;
	return true;
}

// ---- fog: single scattering (Raytracer.cpp:20-192) --------------------------------------------------------------
MIPT_DEV float fog_tanf(float x) { float r; if (mipt_tanf_main(x, r)) return r; return tanf(x); }   // |x| <= pi/2 here: always the exact branch
MIPT_DEV float int_exponential(float y0, float ysol, float beta, float s, float uy) {             // :20-40
	float result;
	if ((double)fabsf(uy * beta) < 0.0001) result = mipt_expf(-beta * (y0 - ysol)) * (s);
	else result = (mipt_expf(-beta * (y0 - ysol)) - mipt_expf(-beta * (y0 + s * uy - ysol))) / (uy * beta);
	return result;
}
MIPT_DEV f3 random_uniform_sphere(uint64_t& rng) {                                                // Vector.h:604-615, T = float
	const float r1 = pcg_uniform(rng);
	const float r2 = pcg_uniform(rng);
	float sn, cs;
	pt_sincosf((float)(2. * MIPT_PI) * r1, sn, cs);
	return mk3(2.f * cs * sqrtf(r2 * (1 - r2)), 2.f * sn * sqrtf(r2 * (1 - r2)), 1.f - 2.f * r2);
}
struct FogEvent { f3 weight; Ray ray; };
// One in-scattering event on [0, t] of ray r (fogContribution).  `attenuation` (the transmittance of the segment) is
// written only once the event is known to lie above the ground (:114) — the caller's variable keeps its previous
// value otherwise, like the reference's local.
template <class STK>
__device__ __noinline__ bool fog_contribution(const DScene* __restrict__ sc, const DRender& R, const Ray& r, f3 sampleLightPos, float t, f3 curWeight,
                                              FogEvent& ev, float& attenuation, uint64_t& rng, unsigned& n_closest, STK& stk) {
	if (norm2(curWeight) < 1E-12) return false;
	const f3 rayDirection = r.d;
	const float p_uniform = 0.5f;
	const bool is_uniform_fog = R.fog_type == 0;
	const float alpha = R.fog_absorption, sigmaT = R.fog_absorption_decay, groundLevel = R.ground_level;
	float int_ext;
	if (is_uniform_fog) int_ext = (float)((double)(alpha * t) * 0.05);
	else int_ext = alpha * int_exponential(r.o.y, groundLevel, sigmaT, t, rayDirection.y);
	const float T = mipt_expf(-int_ext);
	float proba_t, random_t;
	const float clamped_t = 1000.f < t ? 1000.f : t;
	const f3 cl = ld3(R.centerLight);
	const float a = dot(sampleLightPos - r.o, r.d);
	if (a > 0) {                                                            // equi-angular sampling (:71-84)
		const f3 projP = r.o + a * r.d;
		const float D = sqrtf(norm2(sampleLightPos - projP));
		const float thetaA = -mipt_atan2f(a, D);
		const float b = t - a;
		const float thetaB = mipt_atan2f(b, D);
		const float x = pcg_uniform(rng);
		random_t = D * fog_tanf((1 - x) * thetaA + x * thetaB);
		proba_t = D / ((thetaB - thetaA) * (D * D + random_t * random_t));
		random_t += a;
	} else {                                                                // :85-99
		const float alpha2 = 5.f / clamped_t;
		do { random_t = -mipt_logf(pcg_uniform(rng)) / alpha2; } while (random_t > clamped_t);
		const float normalization = 1.f / alpha2 * (1.f - mipt_expf(-alpha2 * clamped_t));
		proba_t = mipt_expf(-alpha2 * random_t) / normalization;
	}
	float int_ext_partielle;
	if (is_uniform_fog) int_ext_partielle = (float)((double)(alpha * random_t) * 0.05);
	else int_ext_partielle = alpha * int_exponential(r.o.y, groundLevel, sigmaT, random_t, rayDirection.y);
	const f3 random_P = r.o + random_t * rayDirection;
	if (random_P.y < groundLevel) return false;                             // :114
	f3 random_dir, point_aleatoire = mk3(0, 0, 0);
	const f3 axeOP = normalize(random_P - cl);
	bool is_uniform;
	if (pcg_uniform(rng) < p_uniform) { random_dir = random_uniform_sphere(rng); is_uniform = true; }
	else {
		const float l1 = pcg_uniform(rng), l2 = pcg_uniform(rng);
		const f3 dir_l = random_cos(axeOP, l1, l2);
		point_aleatoire = dir_l * R.radiusLight + cl;
		random_dir = normalize(point_aleatoire - random_P);
		is_uniform = false;
	}
	float phase_func = 0.f;
	const float k = R.phase_aniso;
	if (R.fog_phase_type == 0) phase_func = (float)(1. / (4. * MIPT_PI));
	else if (R.fog_phase_type == 1) phase_func = (float)((double)(1 - k * k) / (4. * MIPT_PI * (double)(1 + k * dot(random_dir, -rayDirection))));
	else if (R.fog_phase_type == 2) phase_func = (float)(3 / (16 * MIPT_PI) * (double)(1 + sqr(dot(random_dir, rayDirection))));
	Ray L; L.o = random_P; L.d = random_dir;
	Hit ih; f3 interP = mk3(0, 0, 0); Mat im;
	im.shadingN = mk3(0, 1, 0); im.Kd = mk3(0.5f, 0.5f, 0.5f); im.Ks = mk3(0, 0, 0); im.Ne = mk3(100, 100, 100); im.Ke = mk3(0, 0, 0); im.transp = false; im.refr_index = 0;
	const bool interinter = q_intersect(sc, L, ih, interP, im, stk);
	n_closest++;
	bool visible = true;
	if (!is_uniform) {
		const float d_light2 = norm2(point_aleatoire - random_P);
		if (interinter && (double)(ih.t * ih.t) < (double)d_light2 * 0.99) visible = false;
	}
	attenuation = T;
	if (!visible) return false;
	const float pdf_uniform = (float)(1. / (4. * MIPT_PI));
	float pdf_light = 0.f;
	if (interinter && ih.obj == 0) {
		const float J = dot(im.shadingN, -random_dir) / norm2(interP - random_P);
		pdf_light = (float)((double)dot(normalize(interP - cl), axeOP) / (MIPT_PI * (double)sqr(R.radiusLight)) / (double)J);
	}
	const float proba_dir = p_uniform * pdf_uniform + (1 - p_uniform) * pdf_light;
	float ext;
	if (is_uniform_fog) ext = (float)((double)R.fog_density * 0.05);
	else ext = R.fog_density * mipt_expf(-R.fog_density_decay * (random_P.y - groundLevel));
	ev.weight = curWeight * (phase_func * ext * mipt_expf(-int_ext_partielle) / (proba_t * proba_dir));
	ev.ray = L;
	return true;
}

// getColor, literally: pop a contribution, trace it, add what it sees, queue its successors (Raytracer.cpp:196-664 with
// subsProba = 0, no_envmap = false, has_precomputed_rays = false).
template <class STK>
__device__ __noinline__ f3 trace_path_queue(const DScene* __restrict__ sc, const DRender& R, int i, int j, int k, float& dx, float& dy,
                                            unsigned& n_closest, unsigned& n_shadow, STK& stk, QContrib* __restrict__ q, f3& normalValue, f3& albedoValue) {
	normalValue = mk3(0, 0, 0); albedoValue = mk3(0, 0, 0);            // Vector normal, albedo; (:1628)
	PathState ps;
	path_begin(R, i, j, k, ps, dx, dy);
	const int pix = i * R.W + j;
	const bool has_bg = R.backgroundW > 0 && R.background != nullptr;       // :220
	const bool has_fog = R.fog_density > 1E-8;                              // :207
	float attenuationFactor = 0.f;                                          // :206 (uninitialised in the reference)
	f3 color = mk3(0, 0, 0);
	int start = 0, end = 1;
	auto push = [&](f3 w, const Ray& r, int depth, bool lights, bool env, bool hadSS) {
		QContrib c;
		c.w = make_float4(w.x, w.y, w.z, __uint_as_float((unsigned)(depth & 0xffff) | (lights ? 0x10000u : 0u) | (env ? 0x20000u : 0u) | (hadSS ? 0x40000u : 0u)));
		c.o = make_float4(r.o.x, r.o.y, r.o.z, 0.f); c.d = make_float4(r.d.x, r.d.y, r.d.z, 0.f);
		q[end] = c;
		end++; if (end >= MIPT_SIZE_CIRC_ARRAY) end = 0;
	};
	q[0].w = make_float4(1.f, 1.f, 1.f, __uint_as_float((unsigned)(R.nb_bounces & 0xffff) | 0x10000u | 0x20000u));
	q[0].o = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, 0.f); q[0].d = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.f);
	const f3 cl = ld3(R.centerLight);
	while (start != end) {
		const QContrib cur = q[start];
		start++; if (start >= MIPT_SIZE_CIRC_ARRAY) start = 0;
		const unsigned bits = __float_as_uint(cur.w.w);
		Ray currentRay; currentRay.o = mk3(cur.o.x, cur.o.y, cur.o.z); currentRay.d = mk3(cur.d.x, cur.d.y, cur.d.z);
		const f3 pathWeight = mk3(cur.w.x, cur.w.y, cur.w.z);
		const int nbrebonds = (int)(bits & 0xffffu);
		const bool show_lights = (bits & 0x10000u) != 0, show_envmap = (bits & 0x20000u) != 0, hadSS = (bits & 0x40000u) != 0;
		if (nbrebonds == 0) continue;                                       // :240
		if (norm2(pathWeight) < sqr(0.01f)) continue;                       // :241
		Hit h; f3 P = mk3(0, 0, 0); Mat m;
		const bool hit = q_intersect(sc, currentRay, h, P, m, stk);
		n_closest++;
		const float t = h.t;
		// fog event along the ray just traced, towards `lightpos`; queues the in-scattered path (showenvmap = true)
		auto fog = [&](const Ray& ray, f3 lightpos) {
			FogEvent ev;
			if (fog_contribution(sc, R, ray, lightpos, t, pathWeight, ev, attenuationFactor, ps.rng, n_closest, stk)) push(ev.weight, ev.ray, nbrebonds - 1, show_lights, true, hadSS);
		};
		if (hit && nbrebonds == R.nb_bounces) { normalValue = m.shadingN; albedoValue = m.Kd; }   // :255-258 (every contribution still at the first depth)
		if (nbrebonds == R.nb_bounces && has_bg && (!hit || h.obj == 1)) {  // :260-268: a camera ray that leaves the scene shows the photo
			color = color + pathWeight * background_pixel(R, i, j);
			continue;
		}
		if (!hit) { if (R.fog_density == 0) continue; else break; }         // :654-657
		f3 N = m.shadingN, rayDirection = currentRay.d;
		if (h.obj == 1) {                                                   // :275-301
			if (!show_envmap) { if (has_fog) fog(currentRay, cl); continue; }
			if (has_fog) { fog(currentRay, cl); color = color + ((attenuationFactor * pathWeight) * R.envmap_intensity) * m.Ke; }
			else color = color + (pathWeight * R.envmap_intensity) * m.Ke;
			continue;
		}
		if (h.obj == 0) {                                                   // :303-316
			const f3 cc = show_lights ? mk3(R.lightPower, R.lightPower, R.lightPower) : mk3(0.f, 0.f, 0.f);
			if (has_fog) { fog(currentRay, cl); color = color + (attenuationFactor * pathWeight) * cc; }
			else color = color + pathWeight * cc;
			continue;
		}
		const DObject& obj = sc->obj[h.obj];
		const double* const merl = obj.merl;
		f3 Ksub = sc->inherit_material ? m.Ksub : hit_ksub(obj, h, xf_point(obj.inv, currentRay.o) + h.t * xf_dir(obj.inv, currentRay.d));     // (inherit: what Scene::intersection's one MaterialValues held for the winner)
		const bool is_subsurface = norm2(Ksub) > 1E-8;                      // :271
		const float subsProba = (hadSS || !is_subsurface) ? 0.f : 0.6f;     // :318
		const float inv1MSubsProba = 1.f / (1.f - subsProba);
		f3 subsW = mk3(inv1MSubsProba, inv1MSubsProba, inv1MSubsProba);
		bool sub_interaction = false;
		if (is_subsurface && (pcg_uniform(ps.rng) < subsProba)) {           // :324-404: leave through a random point of the same object nearby
			sub_interaction = true;
			const float invSubsProba = 1.f / subsProba;
			subsW = mk3(invSubsProba, invSubsProba, invSubsProba);
			const float sigmasub = 1.5f;
			const float diskR = sqrtf(12.46f) * sigmasub;
			const float integ = 1.f - mipt_expf(-diskR * diskR / (2.f * sigmasub * sigmasub));
			const float randR = sigmasub * sqrtf(-2.f * mipt_logf(1.f - pcg_uniform(ps.rng) * integ));
			const float randangle = pcg_uniform(ps.rng) * 2.f * (float)MIPT_PI;
			float sn_, cs_;
			pt_sincosf(randangle, sn_, cs_);
			const float gauss0 = randR * sn_, gauss1 = randR * cs_, gauss2 = randR;
			const float gaussval = (float)((1. / (double)(sigmasub * sigmasub * 2.f * (float)MIPT_PI)) * (double)mipt_expf(-(gauss2 * gauss2) / (2.f * sigmasub * sigmasub)));
			const float pdfgauss = gaussval / integ;
			const f3 Tg = tangent_of(N);
			const f3 Tg2 = cross(N, Tg);
			const f3 PtaboveP = ((P + gauss0 * Tg) + gauss1 * Tg2) + N * diskR;
			const float r1s = pcg_uniform(ps.rng);
			f3 axis = -N;
			float tmax;
			const float hh = sqrtf(diskR * diskR - gauss2 * gauss2);
			f3 subsOrigin = PtaboveP + (diskR - hh) * (-N);
			float wAxis;
			if (r1s < 0.5f) { wAxis = 0.5f; tmax = 2.f * hh; }
			else {
				wAxis = 0.25f;
				tmax = 2.f * gauss2;
				if (r1s < 0.75f) axis = Tg; else axis = Tg2;
				const float r2s = pcg_uniform(ps.rng);
				if (r2s < 0.5f) subsOrigin = subsOrigin - hh * N;
			}
			// Scene::get_random_intersection on this object only (Geometry.cpp:339-470)
			Ray probe; probe.o = subsOrigin; probe.d = axis;
			Hit sh; sh.obj = h.obj; sh.tri = -1; sh.t = 0; sh.beta = sh.gamma = 0;
			const f3 po = xf_point(obj.inv, probe.o), pd = xf_dir(obj.inv, probe.d);
			const bool subsinter = obj.type == 2 ? plane_reservoir(obj, po, pd, 0.f, tmax, ps.rng, sh.t)
			                     : obj.type == 1 ? sphere_reservoir(obj, po, pd, 0.f, tmax, ps.rng, sh.t)
			                                     : mesh_reservoir(obj, po, pd, 0.f, tmax, ps.rng, sh.t, sh.tri, sh.beta, sh.gamma, stk);
			if (subsinter) {
				f3 localP2; Mat subsmat;
				subsmat.shadingN = mk3(0, 1, 0); subsmat.Kd = mk3(0.5f, 0.5f, 0.5f); subsmat.Ks = mk3(0, 0, 0); subsmat.Ne = mk3(100, 100, 100); subsmat.Ke = mk3(0, 0, 0); subsmat.transp = false; subsmat.refr_index = 0;
				hit_material_obj(obj, probe, sh, localP2, subsmat, false, true);
				const float chris = (float)pt_exp64((double)(-norm2(P - localP2)) / (2. * (double)sigmasub * (double)sigmasub));
				const double d0 = 0.5 * (double)dot(subsmat.shadingN, N), d1 = 0.25 * (double)dot(subsmat.shadingN, Tg), d2 = 0.25 * (double)dot(subsmat.shadingN, Tg2);
				const float sumpdfs = (float)((d0 * d0 + d1 * d1) + d2 * d2);
				const float pdfdisk = wAxis * fabsf(dot(axis, subsmat.shadingN)) / sumpdfs;
				subsW = subsW * (pdfdisk / fmaxf(pdfgauss, 0.05f) * chris);
				rayDirection = normalize(localP2 - P);
				P = localP2 + 0.005f * subsmat.shadingN;
				if (r1s < 0.5f) subsW = subsW * 2.f; else subsW = subsW * 4.f;
				subsW = subsW * (Ksub / (float)MIPT_PI);
				m = subsmat;
				Ksub = hit_ksub(obj, sh, po + sh.t * pd, true);
				N = m.shadingN;
			}
		}
		color = color + (pathWeight * m.Ke) * R.envmap_intensity;           // :411
		if (obj.miroir & 1) {                                               // :413-436
			Ray rm; rm.o = P + 0.001f * N; rm.d = reflect(rayDirection, N);
			if (has_fog) { fog(currentRay, cl); push(attenuationFactor * pathWeight, rm, nbrebonds - 1, show_lights, true, hadSS); }
			else push(pathWeight, rm, nbrebonds - 1, show_lights, true, hadSS);
			continue;
		}
		if (m.transp) {                                                     // :438-489
			float n1 = 1.f, n2 = m.refr_index;
			f3 nt = N;
			bool entering = true;
			if (dot(rayDirection, N) > 0) { n1 = m.refr_index; n2 = 1; nt = -N; entering = false; }
			const float radical = 1.f - sqr(n1 / n2) * (1.f - sqr(dot(nt, rayDirection)));
			Ray nr;
			if (radical > 0) {
				const f3 refr = (n1 / n2) * (rayDirection - dot(rayDirection, nt) * nt) - nt * sqrtf(radical);
				const float R0 = sqr((n1 - n2) / (n1 + n2));
				float Rf;
				if (entering) Rf = R0 + (1 - R0) * pt_powf(1.f + dot(rayDirection, N), 5.f);
				else Rf = R0 + (1 - R0) * pt_powf(1.f - dot(refr, N), 5.f);
				if (pcg_uniform(ps.rng) < Rf) { nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, N); }
				else { nr.o = P - 0.001f * nt; nr.d = refr; }
			} else { nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, N); }
			if (has_fog) { fog(currentRay, cl); push(attenuationFactor * pathWeight, nr, nbrebonds - 1, show_lights, true, hadSS); }
			else push(pathWeight, nr, nbrebonds - 1, show_lights, true, hadSS);
			continue;
		}
		// ---- diffuse / glossy vertex (:490-632)
		const f3 axeOP = fast_normalize(P - cl);
		const float l1 = pcg_uniform(ps.rng);
		const float l2 = pcg_uniform(ps.rng);
		const f3 dir_l = random_cos(axeOP, l1, l2);
		const f3 pt_l = dir_l * R.radiusLight + cl;
		const f3 wi = fast_normalize(pt_l - P);
		const float d_light2 = norm2(pt_l - P);
		bool isShadowed;
		if (dot(m.shadingN, wi) < 0) isShadowed = true;
		else {
			Ray rl; rl.o = P + 0.01f * wi; rl.d = wi;
			n_shadow++;
			isShadowed = q_occluded(sc, rl, sqrtf(d_light2) - 0.01f, stk);   // :513: ghosts cast no shadow
		}
		f3 currentContrib = mk3(0, 0, 0);
		if (!isShadowed) {
			if (obj.ghost) {                                                // :522-536: straight on through the ghost, same depth; currentRay itself is replaced
				const f3 offset = dot(N, rayDirection) > 0 ? N : -N;
				currentRay.o = (P + rayDirection * 0.001f) + offset * 0.001f;
				currentRay.d = rayDirection;
				push(pathWeight, currentRay, nbrebonds, show_lights, show_envmap, hadSS);
			} else {                                                        // :538-553 (no direct light on a ghost)
				const f3 brdf = sub_interaction ? Ksub / (float)MIPT_PI : (merl ? merl_eval(merl, wi, -rayDirection, N) : phong_eval(m, wi, -rayDirection, N));   // :540-544
				const float J = dot(dir_l, -wi) / d_light2;
				const float proba = (float)((double)dot(axeOP, dir_l) / (MIPT_PI * (double)R.radiusLight * (double)R.radiusLight));
				if (proba > 0.f) currentContrib = currentContrib + (subsW * (R.lightPower * fmaxf(0.f, dot(N, wi)) * J / proba)) * brdf;
			}
		}
		if (has_fog) { fog(currentRay, pt_l); color = color + (attenuationFactor * pathWeight) * currentContrib; }   // :557-565
		else color = color + pathWeight * currentContrib;                   // :566
		float ip;
		const float r1 = modff(R.randomPerPixel[2 * (size_t)pix] + R.samples2d[2 * k], &ip);
		const float r2 = modff(R.randomPerPixel[2 * (size_t)pix + 1] + R.samples2d[2 * k + 1], &ip);
		float pdf; f3 dir; bool has_sampled_diffuse;
		if (sub_interaction) { dir = random_cos(m.shadingN, r1, r2); pdf = dot(N, dir) / (float)MIPT_PI; has_sampled_diffuse = true; }   // :584-587
		else if (merl) { dir = random_cos(N, r1, r2); pdf = (float)((double)dot(N, dir) / (MIPT_PI)); has_sampled_diffuse = false; }
		else {
			uint64_t peek = ps.rng;                                          // the lobe pick of PhongBRDF::sample (BRDF.h:73-78)
			has_sampled_diffuse = (float)pcg_next(peek) / 4294967296.f < 1 - (m.Ks.x + m.Ks.y + m.Ks.z) / 3.f;
			dir = phong_sample(m, -rayDirection, N, pdf, r1, r2, ps.rng);
		}
		if (dot(dir, N) < 0 || dot(dir, reflect(rayDirection, N)) < 0 || pdf <= 0) continue;   // :593
		const f3 brdf_i = sub_interaction ? Ksub / (float)MIPT_PI : (merl ? merl_eval(merl, dir, -rayDirection, N) : phong_eval(m, dir, -rayDirection, N));   // :603-607
		f3 nw = ((pathWeight * subsW) * brdf_i) * (dot(N, dir) / pdf);                          // :611
		if (obj.ghost && has_bg) {                                          // :614-621
			const f3 bg = background_pixel(R, i, j);
			nw = nw * mk3(bg.x / 196964.699f, bg.y / 196964.699f, bg.z / 196964.699f);
		}
		Ray next; next.o = P + 0.01f * dir; next.d = dir;
		const bool env = (show_envmap && isShadowed && has_sampled_diffuse) || !obj.ghost;      // :626-629
		if (has_fog) push(attenuationFactor * nw, next, nbrebonds - 1, false, env, sub_interaction ? true : hadSS);
		else push(nw, next, nbrebonds - 1, false, env, sub_interaction ? true : hadSS);
	}
	return color;
}

// one thread per (pixel, sample), as k_render_paths; `queues` holds MIPT_SIZE_CIRC_ARRAY entries per thread
__global__ void __launch_bounds__(MIPT_BLOCK) k_render_paths_queue(const DScene* __restrict__ sc, DRender R, DPass ps, DSamples out, DCounters* __restrict__ cnt, QContrib* __restrict__ queues,
                                                                   float4* __restrict__ aov_n, float4* __restrict__ aov_kd) {
	MIPT_DECLARE_STACK(stk);
	long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	long long total = (long long)ps.npix_slots * (ps.k1 - ps.k0);
	unsigned n_closest = 0, n_shadow = 0, n_paths = 0;
	if (tid < total) {
		int kk = (int)(tid / ps.npix_slots);
		int slot = (int)(tid % ps.npix_slots);
		int blk = slot >> 6, in = slot & 63;
		int i = ps.blocks[2 * blk] + (in >> 3), j = ps.blocks[2 * blk + 1] + (in & 7);
		if (aov_n) { aov_n[tid] = make_float4(0.f, 0.f, 0.f, 0.f); aov_kd[tid] = make_float4(0.f, 0.f, 0.f, 0.f); }
		if (i < R.H && j < R.W) {
			float dx, dy;
			f3 nv, av;
			f3 c = trace_path_queue(sc, R, i, j, ps.k0 + kk, dx, dy, n_closest, n_shadow, stk, queues + (size_t)tid * MIPT_SIZE_CIRC_ARRAY, nv, av);
			out.col[tid] = make_float4(c.x, c.y, c.z, 0.f); out.dxdy[tid] = make_float2(dx, dy);
			if (aov_n) { aov_n[tid] = make_float4(nv.x, nv.y, nv.z, 0.f); aov_kd[tid] = make_float4(av.x, av.y, av.z, 0.f); }
			n_paths = 1;
		}
	}
	DCounters* my = MIPT_MY_COUNTERS(cnt);
	wave_add(&my->paths, n_paths);
	wave_add(&my->rays_closest, n_closest);
	wave_add(&my->rays_shadow, n_shadow);
}
