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

#define MIPT_SIZE_CIRC_ARRAY 200          // Raytracer.h:114

struct QContrib { float4 w; float4 o; float4 d; };   // w.xyz weight, w.w bits: depth | show_lights << 16 | showenvmap << 17

MIPT_DEV f3 background_pixel(const DRender& R, int screenI, int screenJ) {   // Raytracer.cpp:261-265
	int bi = (int)((float)screenI / (float)R.H * (float)R.backgroundH); bi = min(R.backgroundH - 1, max(0, bi));
	int bj = (int)((float)screenJ / (float)R.W * (float)R.backgroundW); bj = min(R.backgroundW - 1, max(0, bj));
	const float* px = R.background + ((size_t)bi * R.backgroundW + bj) * 3;
	return mk3(px[0], px[1], px[2]);
}

template <class STK>
__device__ __noinline__ f3 trace_path_queue(const DScene* __restrict__ sc, const DRender& R, int i, int j, int k, float& dx, float& dy,
                                            unsigned& n_closest, unsigned& n_shadow, STK& stk, QContrib* __restrict__ q) {
	PathState ps;
	path_begin(R, i, j, k, ps, dx, dy);
	const int pix = i * R.W + j;
	const bool has_bg = R.backgroundW > 0 && R.background != nullptr;       // :220
	int start = 0, end = 1;
	auto push = [&](f3 w, const Ray& r, int depth, bool lights, bool env) {
		QContrib c;
		c.w = make_float4(w.x, w.y, w.z, __uint_as_float((unsigned)(depth & 0xffff) | (lights ? 0x10000u : 0u) | (env ? 0x20000u : 0u)));
		c.o = make_float4(r.o.x, r.o.y, r.o.z, 0.f); c.d = make_float4(r.d.x, r.d.y, r.d.z, 0.f);
		q[end] = c;
		end++; if (end >= MIPT_SIZE_CIRC_ARRAY) end = 0;
	};
	q[0].w = make_float4(1.f, 1.f, 1.f, __uint_as_float((unsigned)(R.nb_bounces & 0xffff) | 0x10000u | 0x20000u));
	q[0].o = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, 0.f); q[0].d = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.f);
	while (start != end) {
		const QContrib cur = q[start];
		start++; if (start >= MIPT_SIZE_CIRC_ARRAY) start = 0;
		const unsigned bits = __float_as_uint(cur.w.w);
		ps.ray.o = mk3(cur.o.x, cur.o.y, cur.o.z); ps.ray.d = mk3(cur.d.x, cur.d.y, cur.d.z);
		ps.weight = mk3(cur.w.x, cur.w.y, cur.w.z);
		ps.depth = (int)(bits & 0xffffu); ps.show_lights = (bits & 0x10000u) != 0;
		const bool show_envmap = (bits & 0x20000u) != 0;
		if (!path_alive(ps)) continue;                                      // :240-241
		const int nbrebonds = ps.depth;
		Hit h; f3 P = mk3(0, 0, 0); Mat m;
		const bool hit = scene_intersect(sc, ps.ray, h, P, m, stk);
		n_closest++;
		if (nbrebonds == R.nb_bounces && has_bg && (!hit || h.obj == 1)) {  // :260-268: a camera ray that leaves the scene shows the photo
			ps.color = ps.color + ps.weight * background_pixel(R, i, j);
			continue;
		}
		if (!hit) continue;
		if (h.obj == 1 && !show_envmap) continue;                           // :276-287
		const DObject& obj = sc->obj[h.obj];
		if (h.obj < 2 || !obj.ghost || obj.miroir || m.transp) {            // everything but the diffuse / glossy vertex of a ghost: as in the linear loop
			ShadowRequest sh; f3 wv;
			const bool cont = path_vertex(sc, R, ps, hit, h, P, m, pix, k, sh, wv);
			if (sh.diffuse) {
				f3 contrib = sh.contrib;
				if (sh.cast) { n_shadow++; if (scene_occluded<STK, true>(sc, sh.ray, sh.dist, stk)) contrib = mk3(0, 0, 0); }   // :513: ghosts cast no shadow
				else contrib = mk3(0, 0, 0);
				ps.color = ps.color + wv * contrib;                          // :566
			}
			if (cont) push(ps.weight, ps.ray, ps.depth, ps.show_lights, true);   // showenvmap: default argument (:425, :485) or `... || !ghost` (:629)
			continue;
		}
		// ---- diffuse / glossy vertex on a ghost object (:490-632)
		const f3 N = m.shadingN, rayDirection = ps.ray.d, pathWeight = ps.weight;
		const double* const merl = obj.merl;
		ps.color = ps.color + (pathWeight * m.Ke) * R.envmap_intensity;      // :411
		const f3 cl = ld3(R.centerLight);
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
			isShadowed = scene_occluded<STK, true>(sc, rl, sqrtf(d_light2) - 0.01f, stk);
		}
		if (!isShadowed) {                                                  // :522-536: straight on through the ghost, same depth
			const f3 offset = dot(N, rayDirection) > 0 ? N : -N;
			Ray through; through.o = (P + rayDirection * 0.001f) + offset * 0.001f; through.d = rayDirection;
			push(pathWeight, through, nbrebonds, ps.show_lights, show_envmap);
		}
		ps.color = ps.color + pathWeight * mk3(0.f, 0.f, 0.f);              // :547-566: no direct light on a ghost
		float ip;
		const float r1 = modff(R.randomPerPixel[2 * (size_t)pix] + R.samples2d[2 * k], &ip);
		const float r2 = modff(R.randomPerPixel[2 * (size_t)pix + 1] + R.samples2d[2 * k + 1], &ip);
		float pdf; f3 dir; bool has_sampled_diffuse;
		if (merl) { dir = random_cos(N, r1, r2); pdf = (float)((double)dot(N, dir) / (MIPT_PI)); has_sampled_diffuse = false; }
		else {
			uint64_t peek = ps.rng;                                          // the lobe pick of PhongBRDF::sample (BRDF.h:73-78)
			has_sampled_diffuse = (float)pcg_next(peek) / 4294967296.f < 1 - (m.Ks.x + m.Ks.y + m.Ks.z) / 3.f;
			dir = phong_sample(m, -rayDirection, N, pdf, r1, r2, ps.rng);
		}
		if (dot(dir, N) < 0 || dot(dir, reflect(rayDirection, N)) < 0 || pdf <= 0) continue;   // :593
		const f3 brdf_i = merl ? merl_eval(merl, dir, -rayDirection, N) : phong_eval(m, dir, -rayDirection, N);
		f3 nw = ((pathWeight * mk3(1.f, 1.f, 1.f)) * brdf_i) * (dot(N, dir) / pdf);             // :611
		if (has_bg) {                                                       // :614-621
			const f3 bg = background_pixel(R, i, j);
			nw = nw * mk3(bg.x / 196964.699f, bg.y / 196964.699f, bg.z / 196964.699f);
		}
		Ray next; next.o = P + 0.01f * dir; next.d = dir;
		push(nw, next, nbrebonds - 1, false, show_envmap && isShadowed && has_sampled_diffuse);   // :629
	}
	return ps.color;
}

// one thread per (pixel, sample), as k_render_paths; `queues` holds MIPT_SIZE_CIRC_ARRAY entries per thread
__global__ void __launch_bounds__(MIPT_BLOCK) k_render_paths_queue(const DScene* __restrict__ sc, DRender R, DPass ps, DSamples out, DCounters* __restrict__ cnt, QContrib* __restrict__ queues) {
	MIPT_DECLARE_STACK(stk);
	long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	long long total = (long long)ps.npix_slots * (ps.k1 - ps.k0);
	unsigned n_closest = 0, n_shadow = 0, n_paths = 0;
	if (tid < total) {
		int kk = (int)(tid / ps.npix_slots);
		int slot = (int)(tid % ps.npix_slots);
		int blk = slot >> 6, in = slot & 63;
		int i = ps.blocks[2 * blk] + (in >> 3), j = ps.blocks[2 * blk + 1] + (in & 7);
		if (i < R.H && j < R.W) {
			float dx, dy;
			f3 c = trace_path_queue(sc, R, i, j, ps.k0 + kk, dx, dy, n_closest, n_shadow, stk, queues + (size_t)tid * MIPT_SIZE_CIRC_ARRAY);
			out.col[tid] = make_float4(c.x, c.y, c.z, 0.f); out.dxdy[tid] = make_float2(dx, dy);
			n_paths = 1;
		}
	}
	DCounters* my = MIPT_MY_COUNTERS(cnt);
	wave_add(&my->paths, n_paths);
	wave_add(&my->rays_closest, n_closest);
	wave_add(&my->rays_shadow, n_shadow);
}
