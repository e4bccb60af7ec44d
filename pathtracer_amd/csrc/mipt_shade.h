// mipt_shade.h — device-side camera, samplers, Phong BRDF and the vertex logic of
// Raytracer::getColor (Raytracer.cpp:196-664, in-scope branches: light, environment sphere,
// mirror, Fresnel dielectric, diffuse/glossy with next-event estimation).
#pragma once
#include "mipt_trace.h"

// ---------------------------------------------------------------- Camera::generateDirection (Vector.h:792-825)
MIPT_DEV Ray camera_ray(const DRender& R, int i, int j, float dx_sensor, float dy_sensor, float dx_aperture, float dy_aperture) {
	f3 pos = ld3(R.cam_pos), dir = ld3(R.cam_dir), up = ld3(R.cam_up), right = ld3(R.cam_right);
	f3 dv, C1 = pos;
	if (R.lent_on) {                                                      // Vector.h:799-812: one of lent_nb cameras per pixel column
		const int offset = -((j / R.lent_pw) % R.lent_nb - R.lent_nb / 2);
		const f3 Pf = pos + R.focus * mk3(0, 0, 1);
		C1 = pos + ((float)offset * R.lent_L) * right;
		const f3 v1 = normalize(Pf - C1);
		const f3 proj = (R.cam_k / dot(v1, dir)) * v1 + C1;
		const float pj = (float)((double)(proj.x + (float)(R.W / 2)) - 0.5), pi = (float)((double)(proj.y + (float)(R.H / 2)) - 0.5);
		dv = mk3(((float)j - pj) + dx_sensor, ((float)i - pi) + dy_sensor, R.cam_k);
	} else {
		// Vector(j - W/2 + 0.5 + dx, i - H/2 + 0.5 + dy, k): integer W/2, double sum, narrowed
		dv = mk3((float)((double)(j - R.W / 2) + 0.5 + (double)dx_sensor), (float)((double)(i - R.H / 2) + 0.5 + (double)dy_sensor), R.cam_k);
	}
	dv = normalize(dv);
	dv = right * dv.x + up * dv.y + dir * dv.z;
	f3 destination = C1 + (R.focus / fabsf(dot(dv, dir))) * dv;
	f3 new_origin = C1 + dx_aperture * right + dy_aperture * up;
	f3 new_direction = normalize(destination - new_origin);
	Ray r;
	r.o = new_origin + (R.init_t * new_direction) / dot(new_direction, dir);
	r.d = new_direction;
	return r;
}

// ---------------------------------------------------------------- samplers (Vector.h:567-600)
MIPT_DEV f3 tangent_of(f3 N) {            // getTangent
	float ax = fabsf(N.x), ay = fabsf(N.y), az = fabsf(N.z);
	f3 t;
	if (ax <= ay && ax <= az) t = mk3(0, -N.z, N.y);
	else if (ay <= ax && ay <= az) t = mk3(-N.z, 0, N.x);
	else t = mk3(-N.y, N.x, 0);
	return normalize(t);
}
MIPT_DEV f3 random_cos(f3 N, float r1, float r2) {
#if defined(MIPT_PERTURB) && MIPT_PERTURB == 3
	return N;                                                               // measurement probe: no sampling frame
#endif
	float sr2 = sqrtf(1.f - r2);
	const float twopi = (float)(2. * MIPT_PI);
	float sn, cs;
	pt_sincosf(twopi * r1, sn, cs);
	f3 loc = mk3(cs * sr2, sn * sr2, sqrtf(r2));
	f3 t1 = tangent_of(N);
	f3 t2 = cross(t1, N);
	return loc.z * N + loc.x * t1 + loc.y * t2;
}

// ---------------------------------------------------------------- PhongBRDF (BRDF.h:41-96)
__device__ __attribute__((noinline)) f3 random_phong(f3 R, float phong_exponent, float r1, float r2) {
	float facteur = sqrtf(1 - pt_powf(r2, 2.f / (phong_exponent + 1.f)));
	double ang = 2 * MIPT_PI * (double)r1;
	double ca, sa;
	pt_sincos64(ang, sa, ca);                 // cos(ang), sin(ang): one sincos() call in the compiled reference
	f3 loc = mk3((float)(ca * (double)facteur), (float)(sa * (double)facteur), (float)pt_pow64((double)r2, 1. / (double)(phong_exponent + 1)));
	f3 t1 = tangent_of(R);
	f3 t2 = cross(t1, R);
	return loc.z * R + loc.x * t1 + loc.y * t2;
}
MIPT_DEV f3 phong_sample(const Mat& mat, f3 wo, f3 N, float& pdf, float r1, float r2, uint64_t& rng) {
	float avgNe = (mat.Ne.x + mat.Ne.y + mat.Ne.z) / 3.f;
	float p = 1 - (mat.Ks.x + mat.Ks.y + mat.Ks.z) / 3.f;
	f3 R = reflect(-wo, N);
	f3 dir;
	if ((float)pcg_next(rng) / 4294967296.f < p) dir = random_cos(N, r1, r2);   // engine()/(float)engine.max()
	else dir = random_phong(R, avgNe, r1, r2);
	float proba_phong = (float)((double)(avgNe + 1) / (2.f * MIPT_PI) * (double)pt_powf(dot(R, dir), avgNe));
	float proba_globale = (float)((double)(p * dot(N, dir)) / (MIPT_PI) + (double)((1.f - p) * proba_phong));
	pdf = proba_globale;
	return dir;
}
MIPT_DEV f3 phong_eval(const Mat& mat, f3 wi, f3 wo, f3 N) {
	f3 reflechi = reflect(-wo, N);
	float d = dot(reflechi, wi);
	f3 diffuse = mat.Kd / (float)MIPT_PI;
	if (d < 0) return diffuse;
	f3 lobe = mk3((float)((double)(pt_powf(d, mat.Ne.x) * (mat.Ne.x + 2.f)) / MIPT_TWO_PI_TRUNC),
	              (float)((double)(pt_powf(d, mat.Ne.y) * (mat.Ne.y + 2.f)) / MIPT_TWO_PI_TRUNC),
	              (float)((double)(pt_powf(d, mat.Ne.z) * (mat.Ne.z + 2.f)) / MIPT_TWO_PI_TRUNC));
	return diffuse + lobe * mat.Ks;
}

// ---------------------------------------------------------------- IsoMERLBRDF (BRDF.h:192-247, MERLBRDFRead.cpp:29-206)
// fp64 half/difference-angle transform and table lookup, as the reference.  cos / sin / acos / atan2 are the host libm's
// sincos(), acos(), atan2() restated (mipt_libm64.h): the angles, and with them the table cells, are the reference's by
// construction.
MIPT_DEV void merl_rotate(const double* v, const double* axis, double angle, double* out, const L64Tables& TB) {   // rotate_vector :49-72
	double ca, sa;
	pt_sincos64(angle, sa, ca, TB);
	out[0] = v[0] * ca; out[1] = v[1] * ca; out[2] = v[2] * ca;
	double temp = axis[0] * v[0] + axis[1] * v[1] + axis[2] * v[2];
	temp = temp * (1.0 - ca);
	out[0] += axis[0] * temp; out[1] += axis[1] * temp; out[2] += axis[2] * temp;
	double cx = axis[1] * v[2] - axis[2] * v[1], cy = axis[2] * v[0] - axis[0] * v[2], cz = axis[0] * v[1] - axis[1] * v[0];
	out[0] += cx * sa; out[1] += cy * sa; out[2] += cz * sa;
}
// merl_eval_inline: the body; merl_eval: the same out of line — what every caller uses.  As a function it keeps 65 registers and saves 12
// callee-saved ones through scratch (24 accesses per call, two calls per diffuse vertex).  Round 4 measured the alternative on
// configs[4] (profiles/r4_c_c4_merl_inlined_variants_sweep.txt; code at git tag r4-merl-inline-experiment): both evaluations through
// ONE inlined copy (a two-trip loop, everything that needs no BRDF value finished in front of it) — 67 spilled values at 3 waves
// per SIMD, generate + shade 869 -> 1 089 ms per step; at 2 waves (256 registers, nothing spilled) 1 028 ms.  The stage wants its
// third wave more than it minds the 48 scratch accesses per vertex; a function cannot be given a register budget
// (amdgpu_num_vgpr is for kernels), and one call for both evaluations made the allocator take 248 registers for the callee.
#define MIPT_MERL_CELL 4
#define MIPT_MERL_CELLS (90 * 90 * 180)
// planar table as the reference holds it (red, green, blue planes) -> cells of {r, g, b, 0}
__global__ void k_merl_interleave(const double* __restrict__ planar, double* __restrict__ cells) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= MIPT_MERL_CELLS) return;
	cells[MIPT_MERL_CELL * (size_t)i] = planar[i]; cells[MIPT_MERL_CELL * (size_t)i + 1] = planar[i + MIPT_MERL_CELLS];
	cells[MIPT_MERL_CELL * (size_t)i + 2] = planar[i + 2 * MIPT_MERL_CELLS]; cells[MIPT_MERL_CELL * (size_t)i + 3] = 0.0;
}
// TB: where the fp64 libm functions read their tables (mipt_libm64.h): the library's arrays, or a kernel's copies in LDS
MIPT_DEV f3 merl_eval_inline(const double* __restrict__ data, f3 wi, f3 wo, f3 N, const L64Tables& TB) {
	f3 t1 = tangent_of(N);
	f3 t2 = cross(t1, N);
	f3 wil = mk3(dot(wi, t1), dot(wi, t2), dot(wi, N));
	f3 wol = mk3(dot(wo, t1), dot(wo, t2), dot(wo, N));
	float thetai = mipt_acosf(wil.z);
	if ((double)thetai >= MIPT_PI / 2) return mk3(0, 0, 0);
	float thetao = mipt_acosf(wol.z);
	if ((double)thetao >= MIPT_PI / 2) return mk3(0, 0, 0);
	float phio = mipt_atan2f(wol.y, wol.x);
	if (phio < 0) phio = (float)((double)phio + 2 * MIPT_PI);
	float phii = mipt_atan2f(wil.y, wil.x);
	if (phii < 0) phii = (float)((double)phii + 2 * MIPT_PI);
	// std_coords_to_half_diff_coords (:76-127)
	double theta_in = thetai, fi_in = phii, theta_out = thetao, fi_out = phio;
	double in_z, pin, cfi, sfi;
	pt_sincos64(theta_in, pin, in_z, TB);
	pt_sincos64(fi_in, sfi, cfi, TB);
	double in_x = pin * cfi, in_y = pin * sfi;
	double in[3] = {in_x, in_y, in_z};
	{ double len = sqrt(in[0] * in[0] + in[1] * in[1] + in[2] * in[2]); in[0] = in[0] / len; in[1] = in[1] / len; in[2] = in[2] / len; }
	double out_z, pout, cfo, sfo;
	pt_sincos64(theta_out, pout, out_z, TB);
	pt_sincos64(fi_out, sfo, cfo, TB);
	double out_x = pout * cfo, out_y = pout * sfo;
	double half[3] = {(in_x + out_x) / 2.0, (in_y + out_y) / 2.0, (in_z + out_z) / 2.0};
	{ double len = sqrt(half[0] * half[0] + half[1] * half[1] + half[2] * half[2]); half[0] = half[0] / len; half[1] = half[1] / len; half[2] = half[2] / len; }
	double theta_half = pt_acos64(half[2], TB);
	double fi_half = pt_atan264(half[1], half[0], TB);
	const double bi_normal[3] = {0.0, 1.0, 0.0}, normal[3] = {0.0, 0.0, 1.0};
	double temp[3], diff[3];
	merl_rotate(in, normal, -fi_half, temp, TB);
	merl_rotate(temp, bi_normal, -theta_half, diff, TB);
	double theta_diff = pt_acos64(diff[2], TB);
	double fi_diff = pt_atan264(diff[1], diff[0], TB);
	// index functions (:134-180)
	int th_idx;
	if (theta_half <= 0.0) th_idx = 0;
	else {
		double deg = ((theta_half / (MIPT_PI / 2.0)) * 90);
		double t = sqrt(deg * 90);
		th_idx = (int)t;
		if (th_idx < 0) th_idx = 0;
		if (th_idx >= 90) th_idx = 89;
	}
	int td = (int)(theta_diff / (MIPT_PI * 0.5) * 90);
	int td_idx = td < 0 ? 0 : (td < 89 ? td : 89);
	if (fi_diff < 0.0) fi_diff += MIPT_PI;
	int pd = (int)(fi_diff / MIPT_PI * 360 / 2);
	int pd_idx = pd < 0 ? 0 : (pd < 179 ? pd : 179);
	int ind = pd_idx + td_idx * 180 + th_idx * 180 * 90;
	// lookup_brdf_val reads brdf[ind], brdf[ind + 90*90*180], brdf[ind + 90*90*360] (MERLBRDFRead.cpp:196-203): three cache lines
	// 11.7 MB apart.  The device copy of the table keeps the three values of a cell side by side (32-byte cells, MIPT_MERL_CELL
	// doubles each: mipt_upload_scene, k_merl_interleave): one line per evaluation.
	typedef double v2d_ __attribute__((ext_vector_type(2)));
	const __attribute__((address_space(1))) v2d_* cell = (const __attribute__((address_space(1))) v2d_*)(data + (size_t)MIPT_MERL_CELL * (size_t)ind);
	const v2d_ rg = cell[0], bx = cell[1];
	double r = rg.x * (1.0 / 1500.0);
	double g = rg.y * (1.15 / 1500.0);
	double b = bx.x * (1.66 / 1500.0);
	return mk3((float)r, (float)g, (float)b);
}
MIPT_DEV f3 merl_eval_inline(const double* __restrict__ data, f3 wi, f3 wo, f3 N) { return merl_eval_inline(data, wi, wo, N, l64_tables()); }
__device__ __attribute__((noinline)) f3 merl_eval(const double* __restrict__ data, f3 wi, f3 wo, f3 N) { return merl_eval_inline(data, wi, wo, N); }

// ---------------------------------------------------------------- path state
struct PathState {
	Ray ray;
	f3 weight;
	f3 color;
	uint64_t rng;
	int depth;          // nbrebonds
	bool show_lights;
};

// Camera part of one (pixel, sample): seeding rule + the four draws of Raytracer.cpp:1462-1466.
MIPT_DEV void path_begin(const DRender& R, int i, int j, int k, PathState& ps, float& dx, float& dy) {
	uint64_t p = (uint64_t)i * (uint64_t)R.W + (uint64_t)j;
	ps.rng = pcg_seed(p * R.seed_stride + (uint64_t)k);
	dx = pcg_uniform(ps.rng) - 0.5f;
	dy = pcg_uniform(ps.rng) - 0.5f;
	float dx_ap = (pcg_uniform(ps.rng) - 0.5f) * R.aperture;
	float dy_ap = (pcg_uniform(ps.rng) - 0.5f) * R.aperture;
	ps.ray = camera_ray(R, i, j, dx, dy, dx_ap, dy_ap);
	ps.weight = mk3(1.f, 1.f, 1.f);
	ps.color = mk3(0, 0, 0);
	ps.depth = R.nb_bounces;
	ps.show_lights = true;
}

// What a vertex asks the scheduler to do next.
struct ShadowRequest {
	bool diffuse;       // the vertex took the diffuse/glossy branch: color += weight*contrib is due
	bool cast;          // a shadow ray must be traced (otherwise the sample is "shadowed": contrib = 0)
	Ray ray;
	float dist;         // dist_light argument of Scene::intersection_shadow
	f3 contrib;         // currentContrib if the light sample is visible
};

// One vertex of getColor AFTER the closest-hit query: adds emission, prepares the next-event
// estimation (shadow request), samples the continuation ray and updates weight/depth.
// Returns true while the path continues.  `weight_at_vertex` is the weight the direct term must
// be multiplied with once the shadow query is resolved (color += pathWeight*currentContrib).
// MERL = false: the scene has no measured BRDF (the caller knows from the upload): the fp64 table evaluation is not
// compiled into the kernel, which is what its register count is otherwise sized for.
template <bool MERL = true>
MIPT_DEV bool path_vertex(const DScene* __restrict__ sc, const DRender& R, PathState& ps, bool has_inter, const Hit& h, f3 P, const Mat& mat,
                          int pix, int sampleID, ShadowRequest& sh, f3& weight_at_vertex) {
	sh.diffuse = false;
	sh.cast = false;
	sh.contrib = mk3(0, 0, 0);
	weight_at_vertex = ps.weight;
	if (!has_inter) return false;                                        // :654-657
	f3 N = mat.shadingN;
	f3 rayDirection = ps.ray.d;
	if (h.obj == 1) {                                                    // :275-301 environment sphere
		ps.color = ps.color + (ps.weight * R.envmap_intensity) * mat.Ke;
		return false;
	}
	if (h.obj == 0) {                                                    // :303-316 light sphere
		f3 cc = ps.show_lights ? mk3(R.lightPower, R.lightPower, R.lightPower) : mk3(0.f, 0.f, 0.f);
		ps.color = ps.color + ps.weight * cc;
		return false;
	}
	const double* const merl = MERL ? mat.merl : nullptr;               // (the object's flags came with the material: hit_material_obj)
	ps.color = ps.color + (ps.weight * mat.Ke) * R.envmap_intensity;     // :411
	if (mat.miroir & 1) {                                                // :413-436 (bit 1 of miroir is Object::ghost: mipt_scene.h)
		ps.ray.o = P + 0.001f * N;
		ps.ray.d = reflect(rayDirection, N);
		ps.depth--;
		return true;
	}
	if (mat.transp) {                                                    // :438-489
		float n1 = 1.f, n2 = mat.refr_index;
		f3 nt = N;
		bool entering = true;
		if (dot(rayDirection, N) > 0) { n1 = mat.refr_index; n2 = 1; nt = -N; entering = false; }
		float radical = 1.f - sqr(n1 / n2) * (1.f - sqr(dot(nt, rayDirection)));
		Ray nr;
		if (radical > 0) {
			f3 refr = (n1 / n2) * (rayDirection - dot(rayDirection, nt) * nt) - nt * sqrtf(radical);
			float R0 = sqr((n1 - n2) / (n1 + n2));
			float Rf;
			if (entering) Rf = R0 + (1 - R0) * pt_powf(1.f + dot(rayDirection, N), 5.f);
			else Rf = R0 + (1 - R0) * pt_powf(1.f - dot(refr, N), 5.f);
			if (pcg_uniform(ps.rng) < Rf) { nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, N); }
			else { nr.o = P - 0.001f * nt; nr.d = refr; }
		} else {
			nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, N);
		}
		ps.ray = nr;
		ps.depth--;
		return true;
	}
	// diffuse / glossy: next-event estimation on the sphere light (:494-566)
	sh.diffuse = true;
	f3 cl = ld3(R.centerLight);
	f3 axeOP = fast_normalize(P - cl);
	float l1 = pcg_uniform(ps.rng);
	float l2 = pcg_uniform(ps.rng);
	f3 dir_l = random_cos(axeOP, l1, l2);
	f3 pt_l = dir_l * R.radiusLight + cl;
	f3 wi = fast_normalize(pt_l - P);
	float d_light2 = norm2(pt_l - P);
	if (!(dot(mat.shadingN, wi) < 0)) {
		f3 brdf = merl ? merl_eval(merl, wi, -rayDirection, N) : phong_eval(mat, wi, -rayDirection, N);
		float J = dot(dir_l, -wi) / d_light2;
		float proba = (float)((double)dot(axeOP, dir_l) / (MIPT_PI * (double)R.radiusLight * (double)R.radiusLight));
		if (proba > 0.f) sh.contrib = sh.contrib + (mk3(1.f, 1.f, 1.f) * (R.lightPower * fmaxf(0.f, dot(N, wi)) * J / proba)) * brdf;
		sh.cast = true;
		sh.ray.o = P + 0.01f * wi;
		sh.ray.d = wi;
		sh.dist = sqrtf(d_light2) - 0.01f;
	}
	// indirect (:570-632): the same Cranley-Patterson-rotated lattice point at every depth.  Not for the last vertex of a path:
	// its continuation would be queued with depth 0 and dropped by the loop head (:240) before anything of it reaches the colour.
	if (ps.depth <= 1) return false;
	float ip;
	float r1 = modff(R.randomPerPixel[2 * (size_t)pix] + R.samples2d[2 * sampleID], &ip);
	float r2 = modff(R.randomPerPixel[2 * (size_t)pix + 1] + R.samples2d[2 * sampleID + 1], &ip);
	float pdf;
	f3 dir;
	if (merl) {                                       // IsoMERLBRDF::sample (BRDF.h:198-203): cosine lobe, no engine draw
		dir = random_cos(N, r1, r2);
		pdf = (float)((double)dot(N, dir) / (MIPT_PI));
	} else dir = phong_sample(mat, -rayDirection, N, pdf, r1, r2, ps.rng);
	if (dot(dir, N) < 0 || dot(dir, reflect(rayDirection, N)) < 0 || pdf <= 0) return false;   // :593
	f3 brdf_i = merl ? merl_eval(merl, dir, -rayDirection, N) : phong_eval(mat, dir, -rayDirection, N);
	ps.weight = ((ps.weight * mk3(1.f, 1.f, 1.f)) * brdf_i) * (dot(N, dir) / pdf);            // :611
	ps.ray.o = P + 0.01f * dir;
	ps.ray.d = dir;
	ps.show_lights = false;
	ps.depth--;
	return true;
}


// The diffuse branch of path_vertex for a surface with a measured BRDF, with its two table evaluations LEFT OUT and handed
// back as requests (shade tier 4, mipt_wavefront.h: the requests of many vertices are evaluated together, 64 to a trip).
// Both requests are known before either value is needed: wi comes from the light sample, dir from the lattice point
// (IsoMERLBRDF::sample draws nothing from the engine).  What the values are multiplied with travels with the request:
//   A (next-event estimation): contrib = 0 + ((1,1,1) * sa) * brdf,   sa = lightPower * max(0, N.wi) * J / proba     (:548)
//   B (continuation):          weight  = ((w * (1,1,1)) * brdf) * fb, fb = N.dir / pdf                               (:611)
// Only called for a vertex that path_vertex would take through the measured BRDF: a hit on an object other than the light and
// the environment sphere, neither mirror nor transparent, mat.merl != nullptr.  ps.weight is NOT updated (request B does it).
MIPT_DEV void path_vertex_merl_requests(const DRender& R, PathState& ps, f3 P, const Mat& mat, int pix, int sampleID, ShadowRequest& sh,
                                        bool& reqA, f3& xa, float& sa, bool& reqB, f3& xb, float& fb) {
	sh.diffuse = true;
	sh.cast = false;
	sh.contrib = mk3(0, 0, 0);
	reqA = false; reqB = false;
	const f3 N = mat.shadingN;
	const f3 rayDirection = ps.ray.d;
	ps.color = ps.color + (ps.weight * mat.Ke) * R.envmap_intensity;     // :411
	f3 cl = ld3(R.centerLight);
	f3 axeOP = fast_normalize(P - cl);
	float l1 = pcg_uniform(ps.rng);
	float l2 = pcg_uniform(ps.rng);
	f3 dir_l = random_cos(axeOP, l1, l2);
	f3 pt_l = dir_l * R.radiusLight + cl;
	f3 wi = fast_normalize(pt_l - P);
	float d_light2 = norm2(pt_l - P);
	if (!(dot(mat.shadingN, wi) < 0)) {
		float J = dot(dir_l, -wi) / d_light2;
		float proba = (float)((double)dot(axeOP, dir_l) / (MIPT_PI * (double)R.radiusLight * (double)R.radiusLight));
		if (proba > 0.f) { reqA = true; xa = wi; sa = R.lightPower * fmaxf(0.f, dot(N, wi)) * J / proba; }
		sh.cast = true;
		sh.ray.o = P + 0.01f * wi;
		sh.ray.d = wi;
		sh.dist = sqrtf(d_light2) - 0.01f;
	}
	if (ps.depth <= 1) return;
	float ip;
	float r1 = modff(R.randomPerPixel[2 * (size_t)pix] + R.samples2d[2 * sampleID], &ip);
	float r2 = modff(R.randomPerPixel[2 * (size_t)pix + 1] + R.samples2d[2 * sampleID + 1], &ip);
	f3 dir = random_cos(N, r1, r2);                                      // IsoMERLBRDF::sample (BRDF.h:198-203)
	float pdf = (float)((double)dot(N, dir) / (MIPT_PI));
	if (dot(dir, N) < 0 || dot(dir, reflect(rayDirection, N)) < 0 || pdf <= 0) return;   // :593
	reqB = true; xb = dir; fb = dot(N, dir) / pdf;
	ps.ray.o = P + 0.01f * dir;
	ps.ray.d = dir;
	ps.show_lights = false;
	ps.depth--;
}

// Fast tier of the shade stage: the same vertex logic restricted to what a plain diffuse vertex needs
// (miss, light / environment sphere, Lambert-only Phong material with the diffuse lobe picked).
// Anything else — mirror, dielectric, measured BRDF, a specular coefficient, a negative exponent, or the
// 2^-25 event of the lobe pick choosing the Phong lobe at p = 1 — returns VERTEX_DEFER *before any
// state is modified*, and the general kernel redoes that vertex.  For the vertices it does handle the
// arithmetic is the reference's with the terms that are exactly zero left out:
//   eval  = Kd/pi + lobe*Ks            with Ks = 0 and 0 <= lobe < inf   ->  Kd/pi      (x + (+0) = x)
//   pdf   = p*dot/pi + (1-p)*proba_phong with p = 1 and proba_phong finite -> dot/pi    (needs dot(R,dir) >= 0;
//           otherwise the sample is rejected by the reflect test whatever pdf is)
enum { VERTEX_END = 0, VERTEX_CONTINUE = 1, VERTEX_DEFER = 2 };
MIPT_DEV int path_vertex_fast(const DScene* __restrict__ sc, const DRender& R, PathState& ps, bool has_inter, const Hit& h, f3 P, const Mat& mat,
                              int pix, int sampleID, ShadowRequest& sh, f3& weight_at_vertex) {
	sh.diffuse = false;
	sh.cast = false;
	sh.contrib = mk3(0, 0, 0);
	weight_at_vertex = ps.weight;
	if (!has_inter) return VERTEX_END;
	f3 N = mat.shadingN;
	f3 rayDirection = ps.ray.d;
	if (h.obj == 1) { ps.color = ps.color + (ps.weight * R.envmap_intensity) * mat.Ke; return VERTEX_END; }
	if (h.obj == 0) {
		f3 cc = ps.show_lights ? mk3(R.lightPower, R.lightPower, R.lightPower) : mk3(0.f, 0.f, 0.f);
		ps.color = ps.color + ps.weight * cc;
		return VERTEX_END;
	}
	if ((mat.miroir & 1) || mat.transp) {                                // mirror (:413-436) / Fresnel dielectric (:438-489): no NEE, no shadow ray
		ps.color = ps.color + (ps.weight * mat.Ke) * R.envmap_intensity;  // :411
		if (mat.miroir & 1) {
			ps.ray.o = P + 0.001f * N;
			ps.ray.d = reflect(rayDirection, N);
			ps.depth--;
			return VERTEX_CONTINUE;
		}
		float n1 = 1.f, n2 = mat.refr_index;
		f3 nt = N;
		bool entering = true;
		if (dot(rayDirection, N) > 0) { n1 = mat.refr_index; n2 = 1; nt = -N; entering = false; }
		float radical = 1.f - sqr(n1 / n2) * (1.f - sqr(dot(nt, rayDirection)));
		Ray nr;
		if (radical > 0) {
			f3 refr = (n1 / n2) * (rayDirection - dot(rayDirection, nt) * nt) - nt * sqrtf(radical);
			float R0 = sqr((n1 - n2) / (n1 + n2));
			float Rf;
			if (entering) Rf = R0 + (1 - R0) * pt_powf(1.f + dot(rayDirection, N), 5.f);
			else Rf = R0 + (1 - R0) * pt_powf(1.f - dot(refr, N), 5.f);
			if (pcg_uniform(ps.rng) < Rf) { nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, N); }
			else { nr.o = P - 0.001f * nt; nr.d = refr; }
		} else {
			nr.o = P + 0.001f * nt; nr.d = reflect(rayDirection, N);
		}
		ps.ray = nr;
		ps.depth--;
		return VERTEX_CONTINUE;
	}
	const bool plain = mat.merl == nullptr &&
	                   mat.Ks.x == 0.f && mat.Ks.y == 0.f && mat.Ks.z == 0.f && mat.Ne.x >= 0.f && mat.Ne.y >= 0.f && mat.Ne.z >= 0.f;
	if (!plain) return VERTEX_DEFER;
	uint64_t rng = ps.rng;
	f3 cl = ld3(R.centerLight);
	f3 axeOP = fast_normalize(P - cl);
	float l1 = pcg_uniform(rng);
	float l2 = pcg_uniform(rng);
	// lobe pick of PhongBRDF::sample (BRDF.h:73) with p = 1 - 0/3.f = 1: the diffuse lobe unless u == 1.0f
	if (!((float)pcg_next(rng) / 4294967296.f < 1.f)) return VERTEX_DEFER;
	f3 color = ps.color + (ps.weight * mat.Ke) * R.envmap_intensity;     // :411
	sh.diffuse = true;
	f3 dir_l = random_cos(axeOP, l1, l2);
	f3 pt_l = dir_l * R.radiusLight + cl;
	f3 wi = fast_normalize(pt_l - P);
	float d_light2 = norm2(pt_l - P);
	const f3 brdf = mat.Kd / (float)MIPT_PI;
	if (!(dot(mat.shadingN, wi) < 0)) {
		float J = dot(dir_l, -wi) / d_light2;
		float proba = (float)((double)dot(axeOP, dir_l) / (MIPT_PI * (double)R.radiusLight * (double)R.radiusLight));
		if (proba > 0.f) sh.contrib = sh.contrib + (mk3(1.f, 1.f, 1.f) * (R.lightPower * fmaxf(0.f, dot(N, wi)) * J / proba)) * brdf;
		sh.cast = true;
		sh.ray.o = P + 0.01f * wi;
		sh.ray.d = wi;
		sh.dist = sqrtf(d_light2) - 0.01f;
	}
	ps.color = color;
	ps.rng = rng;
	// The last vertex of a path: getColor still samples a continuation and queues it with depth 0, and the loop head drops it
	// (Raytracer.cpp:240) — nothing of it reaches the colour, so it is not computed.
	if (ps.depth <= 1) return VERTEX_END;
	float ip;
	float r1 = modff(R.randomPerPixel[2 * (size_t)pix] + R.samples2d[2 * sampleID], &ip);
	float r2 = modff(R.randomPerPixel[2 * (size_t)pix + 1] + R.samples2d[2 * sampleID + 1], &ip);
	f3 dir = random_cos(N, r1, r2);
	float pdf = (float)((double)(1.f * dot(N, dir)) / (MIPT_PI) + (double)(0.f));
	if (dot(dir, N) < 0 || dot(dir, reflect(rayDirection, N)) < 0 || pdf <= 0) return VERTEX_END;   // :593
	ps.weight = ((ps.weight * mk3(1.f, 1.f, 1.f)) * brdf) * (dot(N, dir) / pdf);                   // :611
	ps.ray.o = P + 0.01f * dir;
	ps.ray.d = dir;
	ps.show_lights = false;
	ps.depth--;
	return VERTEX_CONTINUE;
}

// Termination tests at the top of the loop (Raytracer.cpp:240-241)
MIPT_DEV bool path_alive(const PathState& ps) {
	if (ps.depth == 0) return false;
	if (norm2(ps.weight) < sqr(0.01f)) return false;
	return true;
}
