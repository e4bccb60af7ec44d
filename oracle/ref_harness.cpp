// TEST INFRASTRUCTURE ONLY — harness around the *compiled reference* (nbonneel/pathtracer).
//
// This file is ours; it contains none of the reference's source.  It is compiled by
// oracle/Makefile as ONE translation unit together with the reference's own .cpp files
// (included by name from a scratch copy of /root/reference that carries the three textual
// GCC-compatibility fixes of SURVEY.md §8c) so that every function shares a single
// `engine[]` RNG array (SURVEY.md §5 "RNG").  The output is oracle/_ref/libptref.so
// (git-ignored).  It is used only by tests/ (to pin oracle/pt_oracle.c against the real
// reference and to generate tests/golden/*), and by bench.py's cpu_baseline leg
// (kind "reference").  Product code never links or loads it.
//
// The harness drives only the reference's public members:
//   Raytracer::loadScene / prepare_render / getColor / render_image_nopreviz / render_image
//   Scene::intersection / intersection_shadow, Camera::generateDirection,
//   TriMesh ctor + bvh/triangleSoup/indices members, PhongBRDF::sample/eval, pcg32.
#include "chrono.h"
#include "Vector.cpp"
#include "Geometry.cpp"
#include "TriangleMesh.cpp"
#include "PointSet.cpp"
#include "Raytracer.cpp"
// Raytracer.cpp's render_image_nopreviz() is left unterminated when
// USE_OPENIMAGEDENOISER is undefined (SURVEY.md §8c fix 3); the Makefile appends the
// closing brace to the scratch copy, so nothing to do here.

#include <chrono>
#include <cstdint>
#include <cstring>

extern "C" {

struct RefCtx {
	Raytracer* rt;
};

// The reference sizes its per-thread state for at most 64 OpenMP threads (engine[64] Vector.h:29,
// contribsArray[64][..] Raytracer.h:115) and indexes it with omp_get_max_threads() in its
// constructors: on a host with more hardware threads it overruns those arrays.  Cap at 64.
static int ref_thread_cap() { return omp_get_num_procs() < 64 ? omp_get_num_procs() : 64; }

RefCtx* ref_create() {
	if (omp_get_max_threads() > 64) omp_set_num_threads(64);
	RefCtx* c = new RefCtx;
	c->rt = new Raytracer();  // heap: contribsArray is ~600 KB
	Raytracer* rt = c->rt;
	rt->loadScene();
	// fields the reference leaves uninitialised
	rt->s.fog_density = 0; rt->s.fog_absorption = 0; rt->s.fog_density_decay = 0; rt->s.fog_absorption_decay = 0;
	rt->s.fog_type = 0; rt->s.fog_phase_type = 0; rt->s.phase_aniso = 0; rt->s.nbframes = 1;
	rt->last_nrays = -1; rt->lastfilter = -1;
	rt->autosave = false;
	rt->is_recording = false;
	rt->has_denoiser = false;
	return c;
}

void ref_destroy(RefCtx* c) {
	// Scene objects are owned by Scene::clear in the reference; keep it simple.
	delete c->rt;
	delete c;
}

// the reference's image decoder as Texture::loadColors calls it (utils.cpp:98-165: stb_image, 3 channels, rows flipped)
int ref_load_image(const char* file, unsigned char* out, int capacity, int* W, int* H) {
	std::vector<unsigned char> val; size_t w = 0, h = 0;
	if (!load_image(file, val, w, h, false)) return -1;
	if ((int)val.size() > capacity) return -2;
	memcpy(out, &val[0], val.size()); *W = (int)w; *H = (int)h;
	return 0;
}

// the reference's image writer, save_image (utils.cpp:177-234), for 8-bit and for float pixels (maxval as given)
void ref_save_image_u8(const char* file, const unsigned char* rgb, int W, int H) { save_image(file, rgb, W, H); }
void ref_save_image_f32(const char* file, const float* rgb, int W, int H, float maxval) { save_image(file, rgb, W, H, maxval); }

// .scn scene files through the reference's own Raytracer::save_scene / load_scene (Raytracer.cpp:1096-1236)
void ref_save_scene(RefCtx* c, const char* file) { c->rt->save_scene(file); }
void ref_load_scene(RefCtx* c, const char* file) {
	Raytracer* rt = c->rt;
	rt->load_scene(file, NULL);
	rt->last_nrays = -1; rt->lastfilter = -1; rt->randomPerPixel.clear();
	rt->has_denoiser = false;
}
// key-framed transforms (Geometry.h:258-320): the frame they are evaluated at, Object::add_keyframe, the transform it records
void ref_set_frame(RefCtx* c, int frame) { c->rt->s.current_frame = frame; }
void ref_add_keyframe(RefCtx* c, int obj, int frame) { c->rt->s.objects[obj]->add_keyframe(frame); }
void ref_set_object_transform(RefCtx* c, int obj, const float* t, const float* r, float scale) {
	Object* o = c->rt->s.objects[obj];
	o->max_translation = Vector(t[0], t[1], t[2]);
	for (int k = 0; k < 9; k++) o->mat_rotation[k] = r[k];
	o->scale = scale;
}
int ref_num_objects(RefCtx* c) { return (int)c->rt->s.objects.size(); }
void ref_get_scene_header(RefCtx* c, float* o) {
	Raytracer& r = *c->rt;
	int k = 0;
	o[k++] = (float)r.W; o[k++] = (float)r.H; o[k++] = (float)r.nrays; o[k++] = (float)r.nb_bounces;
	for (int q = 0; q < 3; q++) o[k++] = r.cam.position[q];
	for (int q = 0; q < 3; q++) o[k++] = r.cam.direction[q];
	for (int q = 0; q < 3; q++) o[k++] = r.cam.up[q];
	o[k++] = r.cam.fov; o[k++] = r.cam.focus_distance; o[k++] = r.cam.aperture; o[k++] = r.sigma_filter; o[k++] = r.gamma;
	o[k++] = r.s.intensite_lumiere; o[k++] = r.s.envmap_intensity; o[k++] = r.s.double_frustum_start_t;
	while (k < 32) o[k++] = 0.f;
}
void ref_get_object_state(RefCtx* c, int obj, float* o, int* fl) {
	Object* ob = c->rt->s.objects[obj];
	int k = 0;
	for (int q = 0; q < 3; q++) o[k++] = ob->max_translation[q];
	for (int q = 0; q < 9; q++) o[k++] = ob->mat_rotation[q];
	for (int q = 0; q < 3; q++) o[k++] = ob->rotation_center[q];
	o[k++] = ob->scale;
	for (int q = 0; q < 8; q++) o[16 + q] = 0.f;
	fl[0] = ob->type == OT_SPHERE ? 1 : (ob->type == OT_PLANE ? 2 : 0); fl[1] = ob->miroir; fl[2] = ob->ghost; fl[3] = ob->flip_normals; fl[4] = ob->interp_normals; fl[5] = 0; fl[6] = fl[7] = 0;
	if (ob->type == OT_SPHERE) { Sphere* sp = dynamic_cast<Sphere*>(ob); for (int q = 0; q < 3; q++) o[16 + q] = sp->O[q]; o[19] = sp->R; fl[5] = sp->has_envmap; }
	if (ob->type == OT_PLANE) { Plane* pl = dynamic_cast<Plane*>(ob); for (int q = 0; q < 3; q++) { o[16 + q] = pl->A[q]; o[19 + q] = pl->vecN[q]; } }
}

void ref_set_render(RefCtx* c, int W, int H, int nrays, int nb_bounces, float sigma_filter) {
	Raytracer* rt = c->rt;
	rt->W = W; rt->H = H; rt->nrays = nrays; rt->nb_bounces = nb_bounces; rt->sigma_filter = sigma_filter;
	rt->last_nrays = -1; rt->lastfilter = -1;
	rt->randomPerPixel.clear();
}

// Camera given directly (no cam.rotate): position, direction, up, fov (radians), focus, aperture.
void ref_set_camera(RefCtx* c, const float* pos, const float* dir, const float* up, float fov, float focus, float aperture) {
	Raytracer* rt = c->rt;
	rt->cam = Camera(Vector(pos[0], pos[1], pos[2]), Vector(dir[0], dir[1], dir[2]), Vector(up[0], up[1], up[2]));
	rt->cam.fov = fov;
	rt->cam.focus_distance = focus;
	rt->cam.aperture = aperture;
}

void ref_get_camera(RefCtx* c, float* out12) {
	Raytracer* rt = c->rt;
	for (int k = 0; k < 3; k++) { out12[k] = rt->cam.position[k]; out12[3 + k] = rt->cam.direction[k]; out12[6 + k] = rt->cam.up[k]; }
	out12[9] = rt->cam.fov; out12[10] = rt->cam.focus_distance; out12[11] = rt->cam.aperture;
}

// Light sphere: centre, radius, and Scene::intensite_lumiere given directly (the caller evaluates
// the reference's formula 1e9*4pi/(4pi*R*R*pi) (Raytracer.cpp:1270) times the GUI slider factor
// (mainApp.cpp:775) in double and narrows to float, as the reference's float member does).
void ref_set_light(RefCtx* c, const float* center, float R, float intensite_lumiere) {
	Raytracer* rt = c->rt;
	rt->s.lumiere->O = Vector(center[0], center[1], center[2]);
	rt->s.lumiere->R = R;
	rt->s.lumiere->R2 = R * R;
	rt->s.lumiere->rotation_center = rt->s.lumiere->O;
	rt->s.intensite_lumiere = intensite_lumiere;
}

void ref_set_envmap_intensity(RefCtx* c, float v) { c->rt->s.envmap_intensity = v; }

// Load an OBJ exactly as the GUI drag-and-drop does (mainApp.cpp:2402-2410):
// TriMesh(&scene, file, 1, (0,0,0), false, NULL, false, center), scale, bottom on the plane.
int ref_add_mesh(RefCtx* c, const char* objfile, float scale, int center) {
	Raytracer* rt = c->rt;
	TriMesh* g = new TriMesh(&rt->s, objfile, 1, Vector(0, 0, 0), false, NULL, false, center != 0);
	g->scale = scale;
	g->display_edges = false;
	g->max_translation = Vector(0, rt->s.objects[2]->get_translation(rt->s.current_time, rt->is_recording)[1] - (g->bbox.bounds[0][1])*g->scale, 0);
	rt->s.addObject(g);
	return (int)rt->s.objects.size() - 1;
}

// OBJ text for ref_add_mesh from in-memory arrays (same text as pathtracer_amd.scenes.write_obj, which
// takes 15 s for 2.5 M triangles in numpy): "v", "vn", optional "vt", faces "a//a" or "a/t/n", 1-based.
int ref_write_obj(const char* path, int nv, const float* v, int nn, const float* n, int nt, const float* uv,
                  int nf, const int* fv, const int* fn, const int* ft) {
	FILE* f = fopen(path, "w");
	if (!f) return -1;
	static char buf[1 << 20];
	setvbuf(f, buf, _IOFBF, sizeof buf);
	for (int i = 0; i < nv; i++) fprintf(f, "v %.9g %.9g %.9g\n", v[3 * i], v[3 * i + 1], v[3 * i + 2]);
	for (int i = 0; i < nn; i++) fprintf(f, "vn %.9g %.9g %.9g\n", n[3 * i], n[3 * i + 1], n[3 * i + 2]);
	for (int i = 0; i < nt; i++) fprintf(f, "vt %.9g %.9g\n", uv[2 * i], uv[2 * i + 1]);
	for (int i = 0; i < nf; i++) {
		if (nt > 0 && ft) fprintf(f, "f %d/%d/%d %d/%d/%d %d/%d/%d\n", fv[3 * i] + 1, ft[3 * i] + 1, fn[3 * i] + 1, fv[3 * i + 1] + 1, ft[3 * i + 1] + 1, fn[3 * i + 1] + 1, fv[3 * i + 2] + 1, ft[3 * i + 2] + 1, fn[3 * i + 2] + 1);
		else fprintf(f, "f %d//%d %d//%d %d//%d\n", fv[3 * i] + 1, fn[3 * i] + 1, fv[3 * i + 1] + 1, fn[3 * i + 1] + 1, fv[3 * i + 2] + 1, fn[3 * i + 2] + 1);
	}
	fclose(f);
	return 0;
}

// Per-object switches the GUI exposes (mirror flag, constant transparency / refraction index
// / Kd / Ks / Ne multipliers for one material group).
int ref_add_sphere(RefCtx* c, const float* O, float R, int mirror, int flip_normals) {
	c->rt->s.addObject(new Sphere(Vector(O[0], O[1], O[2]), R, mirror != 0, flip_normals != 0));
	return (int)c->rt->s.objects.size() - 1;
}
void ref_set_object_flags(RefCtx* c, int obj, int miroir, int flip_normals) {
	c->rt->s.objects[obj]->miroir = miroir != 0;
	c->rt->s.objects[obj]->flip_normals = flip_normals != 0;
}
void ref_set_fog(RefCtx* c, float density, float absorption, float density_decay, float absorption_decay, int type, int phase_type, float phase_aniso) {
	Scene& s = c->rt->s;
	s.fog_density = density; s.fog_absorption = absorption; s.fog_density_decay = density_decay; s.fog_absorption_decay = absorption_decay;
	s.fog_type = type; s.fog_phase_type = phase_type; s.phase_aniso = phase_aniso;
}
void ref_add_col_subsurface(RefCtx* c, int obj, const float* rgb) { c->rt->s.objects[obj]->add_col_subsurface(Vector(rgb[0], rgb[1], rgb[2])); }
void ref_set_group_subsurface(RefCtx* c, int obj, int grp, const float* rgb) {
	Object* o = c->rt->s.objects[obj];
	if (grp >= 0 && grp < (int)o->subsurface.size()) o->subsurface[grp].multiplier = Vector(rgb[0], rgb[1], rgb[2]);
}
void ref_set_lenticular(RefCtx* c, int on, int nb_images, float max_angle, int pixel_width) {
	Camera& cam = c->rt->cam;
	cam.is_lenticular = on != 0; cam.lenticular_nb_images = nb_images; cam.lenticular_max_angle = max_angle; cam.lenticular_pixel_width = pixel_width;
}
void ref_set_object_ghost(RefCtx* c, int obj, int ghost) { c->rt->s.objects[obj]->ghost = ghost != 0; }
// Scene::background as Scene::load_background leaves it (Geometry.h:1355-1363), set directly (the loader reads BMP files)
void ref_set_background(RefCtx* c, const float* rgb, int W, int H) {
	Scene& s = c->rt->s;
	s.clear_background();
	if (rgb && W > 0 && H > 0) { s.background.assign(rgb, rgb + (size_t)W * H * 3); s.backgroundW = W; s.backgroundH = H; }
}
void ref_load_background(RefCtx* c, const char* file) { c->rt->s.load_background(file, c->rt->gamma); }
int ref_get_background(RefCtx* c, float* out, int capacity, int* W, int* H) {
	const Scene& s = c->rt->s;
	*W = s.backgroundW; *H = s.backgroundH;
	if ((int)s.background.size() > capacity) return -1;
	if (out && !s.background.empty()) memcpy(out, s.background.data(), s.background.size() * sizeof(float));
	return (int)s.background.size();
}
void ref_set_group_material(RefCtx* c, int obj, int grp, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr) {
	Object* o = c->rt->s.objects[obj];
	if (grp < (int)o->textures.size()) o->textures[grp].multiplier = Vector(Kd[0], Kd[1], Kd[2]);
	if (grp < (int)o->specularmap.size()) o->specularmap[grp].multiplier = Vector(Ks[0], Ks[1], Ks[2]);
	if (grp < (int)o->roughnessmap.size()) o->roughnessmap[grp].multiplier = Vector(Ne[0], Ne[1], Ne[2]);
	if (grp < (int)o->transparent_map.size()) o->transparent_map[grp].multiplier = Vector(transp_col, transp_col, transp_col);
	if (grp < (int)o->refr_index_map.size()) o->refr_index_map[grp].multiplier = Vector(refr, refr, refr);
}
// give the ground plane (or any object without material lists) one constant material group
void ref_add_group_material(RefCtx* c, int obj, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr) {
	Object* o = c->rt->s.objects[obj];
	o->add_col_texture(Vector(Kd[0], Kd[1], Kd[2]));
	o->add_col_specular(Vector(Ks[0], Ks[1], Ks[2]));
	o->add_col_roughness(Vector(Ne[0], Ne[1], Ne[2]));
	o->add_col_transp(transp_col);
	o->add_col_refr(refr);
}

// Image textures through the reference's own loaders (stb_image reads the binary PPM the test wrote).
// slot: 0 Kd (Object::set_texture), 1 Ks (set_specularmap), 2 normal map (set_normalmap),
// 3 alpha (set_alphamap), 4 roughness / Ne (set_roughnessmap)   (Geometry.cpp:60-146)
void ref_set_group_texture_file(RefCtx* c, int obj, int grp, int slot, const char* file) {
	Object* o = c->rt->s.objects[obj];
	switch (slot) {
	case 0: o->set_texture(file, grp); break;
	case 1: o->set_specularmap(file, grp); break;
	case 2: o->set_normalmap(file, grp); break;
	case 3: o->set_alphamap(file, grp); break;
	case 4: o->set_roughnessmap(file, grp); break;
	case 7: o->set_subsurface(file, grp); break;
	}
}
// mainApp.cpp:2593: ((Sphere*)objects[1])->load_envmap(file)
void ref_set_envmap_file(RefCtx* c, const char* file) { ((Sphere*)c->rt->s.objects[1])->load_envmap(file); }

// mainApp.cpp:2436: objects[selected]->brdf = new IsoMERLBRDF(file)  (MERL ".binary" layout)
void ref_set_brdf_merl_file(RefCtx* c, int obj, const char* file) { c->rt->s.objects[obj]->brdf = new IsoMERLBRDF(std::string(file)); }
// IsoMERLBRDF::eval on explicit tuples (BRDF.h:204-246)
void ref_merl_eval(const char* file, int n, const float* wi3, const float* wo3, const float* N3, float* out3) {
	IsoMERLBRDF brdf{std::string(file)};
	MaterialValues m;
	for (int i = 0; i < n; i++) {
		Vector v = brdf.eval(m, Vector(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), Vector(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), Vector(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]));
		out3[3 * i] = v[0]; out3[3 * i + 1] = v[1]; out3[3 * i + 2] = v[2];
	}
}

void ref_prepare(RefCtx* c) {
	omp_set_num_threads(1);
	c->rt->prepare_render(c->rt->s.current_frame);
}

// ---- state dumps --------------------------------------------------------------------

void ref_get_light(RefCtx* c, float* out5) {  // centerLight[3], radiusLight, lightPower (after prepare)
	Raytracer* rt = c->rt;
	out5[0] = rt->centerLight[0]; out5[1] = rt->centerLight[1]; out5[2] = rt->centerLight[2];
	out5[3] = rt->radiusLight; out5[4] = rt->lightPower;
}

void ref_get_tables(RefCtx* c, float* randomPerPixel /*W*H*2*/, float* samples2d /*nrays*2*/, float* filter_integral, int* filter_size) {
	Raytracer* rt = c->rt;
	if (randomPerPixel) for (int i = 0; i < rt->W*rt->H; i++) { randomPerPixel[2 * i] = rt->randomPerPixel[i][0]; randomPerPixel[2 * i + 1] = rt->randomPerPixel[i][1]; }
	if (samples2d) for (int i = 0; i < rt->nrays; i++) { samples2d[2 * i] = rt->samples2d[i][0]; samples2d[2 * i + 1] = rt->samples2d[i][1]; }
	if (filter_integral) for (size_t i = 0; i < rt->filter_integral.size(); i++) filter_integral[i] = rt->filter_integral[i];
	if (filter_size) *filter_size = rt->filter_size;
}

void ref_get_object_matrices(RefCtx* c, int obj, float* trans12, float* inv12, float* rot9) {
	Object* o = c->rt->s.objects[obj];
	memcpy(trans12, o->trans_matrix, 12 * sizeof(float));
	memcpy(inv12, o->inv_trans_matrix, 12 * sizeof(float));
	memcpy(rot9, o->rot_matrix, 9 * sizeof(float));
}

// material lists of one group as readOBJ / the MTL parser left them: multipliers of Kd, Ks, Ne (3 each), alpha, refr,
// transp (1 each); W, H of the Kd / Ks / normal / alpha images; and the decoded float image of one slot
int ref_num_groups(RefCtx* c, int obj) { return (int)c->rt->s.objects[obj]->textures.size(); }
void ref_get_group_material(RefCtx* c, int obj, int grp, float* out12, int* wh8) {
	Object* o = c->rt->s.objects[obj];
	for (int k = 0; k < 3; k++) { out12[k] = o->textures[grp].multiplier[k]; out12[3 + k] = o->specularmap[grp].multiplier[k]; out12[6 + k] = o->roughnessmap[grp].multiplier[k]; }
	out12[9] = o->alphamap[grp].multiplier[0]; out12[10] = o->refr_index_map[grp].multiplier[0]; out12[11] = o->transparent_map[grp].multiplier[0];
	const Texture* t[4] = {&o->textures[grp], &o->specularmap[grp], &o->normal_map[grp], &o->alphamap[grp]};
	for (int k = 0; k < 4; k++) { wh8[2 * k] = (int)t[k]->W; wh8[2 * k + 1] = (int)t[k]->H; }
}
const float* ref_group_texture_values(RefCtx* c, int obj, int grp, int slot) {
	Object* o = c->rt->s.objects[obj];
	const Texture* t[4] = {&o->textures[grp], &o->specularmap[grp], &o->normal_map[grp], &o->alphamap[grp]};
	return t[slot]->values.empty() ? (const float*)0 : &t[slot]->values[0];
}
void ref_mesh_counts(RefCtx* c, int obj, int* ntri, int* nnodes, int* nverts, int* nnormals, int* nuvs) {
	TriMesh* g = c->rt->s.castToMesh[obj];
	*ntri = (int)g->indices.size(); *nnodes = (int)g->bvh.nodes.size();
	*nverts = (int)g->vertices.size(); *nnormals = (int)g->normals.size(); *nuvs = (int)g->uvs.size();
}

// perm[ntri]; nodes_i[nnodes*3] = isleaf,fg,fd ; nodes_bb[nnodes*6] ; soup[ntri*31] =
// A,u,v,N,m11,m12,m22,invdetm, uvs[6], normals[9] ; groups[ntri]; root_bb[6]
void ref_mesh_dump(RefCtx* c, int obj, int* perm, int* nodes_i, float* nodes_bb, float* soup, int* groups, float* root_bb) {
	TriMesh* g = c->rt->s.castToMesh[obj];
	int ntri = (int)g->indices.size();
	for (int i = 0; i < ntri; i++) {
		if (perm) perm[i] = g->permuted_triangle_index[i];
		if (groups) groups[i] = g->indices[i].group;
		if (soup) {
			const Triangle& T = g->triangleSoup[i];
			float* o = soup + (size_t)i * 31;
			for (int k = 0; k < 3; k++) { o[k] = T.A[k]; o[3 + k] = T.u[k]; o[6 + k] = T.v[k]; o[9 + k] = T.N[k]; }
			o[12] = T.m11; o[13] = T.m12; o[14] = T.m22; o[15] = T.invdetm;
			for (int k = 0; k < 3; k++) { o[16 + 2 * k] = T.uvs[k][0]; o[17 + 2 * k] = T.uvs[k][1]; }
			for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) o[22 + 3 * k + l] = T.normals[k][l];
		}
	}
	int nn = (int)g->bvh.nodes.size();
	for (int i = 0; i < nn; i++) {
		if (nodes_i) { nodes_i[3 * i] = g->bvh.nodes[i].isleaf ? 1 : 0; nodes_i[3 * i + 1] = g->bvh.nodes[i].fg; nodes_i[3 * i + 2] = g->bvh.nodes[i].fd; }
		if (nodes_bb) for (int k = 0; k < 3; k++) { nodes_bb[6 * i + k] = g->bvh.nodes[i].bbox.bounds[0][k]; nodes_bb[6 * i + 3 + k] = g->bvh.nodes[i].bbox.bounds[1][k]; }
	}
	if (root_bb) for (int k = 0; k < 3; k++) { root_bb[k] = g->bvh.bbox.bounds[0][k]; root_bb[3 + k] = g->bvh.bbox.bounds[1][k]; }
}

// ---- leaf functions ----------------------------------------------------------------

void ref_pcg32(uint64_t seed, int n, uint32_t* out) {
	pcg32 e(seed);
	for (int i = 0; i < n; i++) out[i] = e();
}

void ref_lattice(int n, float* out_xy) {
	for (int i = 0; i < n; i++) { Vector v = extensibleLattice2d(i); out_xy[2 * i] = v[0]; out_xy[2 * i + 1] = v[1]; }
}

void ref_invsqroot(int n, const float* in, float* out) {
	for (int i = 0; i < n; i++) out[i] = invSqRoot(in[i]);
}
void ref_fast_normalize(int n, const float* in3, float* out3) {
	for (int i = 0; i < n; i++) { Vector v(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]); v.fast_normalize(); out3[3 * i] = v[0]; out3[3 * i + 1] = v[1]; out3[3 * i + 2] = v[2]; }
}

void ref_fast_exp(int n, const double* in, double* out) {
	for (int i = 0; i < n; i++) out[i] = fast_exp(in[i]);
}

// random_cos(N, r1, r2) (Vector.h:582-589)
void ref_random_cos(int n, const float* N3, const float* r12, float* out3) {
	for (int i = 0; i < n; i++) {
		Vector d = random_cos(Vector(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]), r12[2 * i], r12[2 * i + 1]);
		out3[3 * i] = d[0]; out3[3 * i + 1] = d[1]; out3[3 * i + 2] = d[2];
	}
}

// Camera::generateDirection for pixel (i,j) with explicit jitters; out = origin[3], dir[3]
void ref_camera_rays(RefCtx* c, int n, const int* ij, const float* jit4, float* out6) {
	Raytracer* rt = c->rt;
	for (int q = 0; q < n; q++) {
		Ray r = rt->cam.generateDirection(rt->s.double_frustum_start_t, ij[2 * q], ij[2 * q + 1], rt->s.current_frame,
			jit4[4 * q], jit4[4 * q + 1], jit4[4 * q + 2], jit4[4 * q + 3], rt->W, rt->H);
		for (int k = 0; k < 3; k++) { out6[6 * q + k] = r.origin[k]; out6[6 * q + 3 + k] = r.direction[k]; }
	}
}

// Scene::intersection on n rays (origin[3],dir[3]).  out_i[n*3] = hit, object id, triangle id;
// out_f[n*20] = t, P[3], shadingN[3], Kd[3], Ks[3], Ne[3], Ke[3], refr_index ; out_transp[n]
void ref_intersect(RefCtx* c, int n, const float* rays6, int* out_i, float* out_f) {
	Raytracer* rt = c->rt;
	for (int q = 0; q < n; q++) {
		Ray r(Vector(rays6[6 * q], rays6[6 * q + 1], rays6[6 * q + 2]), Vector(rays6[6 * q + 3], rays6[6 * q + 4], rays6[6 * q + 5]), 0.f);
		Vector P(0, 0, 0); MaterialValues mat; mat.transp = false; mat.refr_index = 0; int id = -1, tri = -1; float t;
		bool h = rt->s.intersection(r, P, id, t, mat, tri, false, false);
		out_i[3 * q] = h ? 1 : 0; out_i[3 * q + 1] = h ? id : -1; out_i[3 * q + 2] = h ? tri : -1;
		float* o = out_f + 20 * (size_t)q;
		o[0] = t;
		for (int k = 0; k < 3; k++) { o[1 + k] = P[k]; o[4 + k] = mat.shadingN[k]; o[7 + k] = mat.Kd[k]; o[10 + k] = mat.Ks[k]; o[13 + k] = mat.Ne[k]; o[16 + k] = mat.Ke[k]; }
		o[19] = h ? (mat.transp ? -mat.refr_index : mat.refr_index) : 0.f;
	}
}

void ref_intersect_shadow(RefCtx* c, int n, const float* rays6, const float* dist_light, int* occluded) {
	Raytracer* rt = c->rt;
	for (int q = 0; q < n; q++) {
		Ray r(Vector(rays6[6 * q], rays6[6 * q + 1], rays6[6 * q + 2]), Vector(rays6[6 * q + 3], rays6[6 * q + 4], rays6[6 * q + 5]), 0.f);
		float t;
		occluded[q] = rt->s.intersection_shadow(r, t, dist_light[q], true, false) ? 1 : 0;
	}
}

// PhongBRDF::sample (with explicit r1,r2; the lobe pick draws from engine[0] reseeded with
// `seed`) and PhongBRDF::eval.  mat9 = Kd,Ks,Ne.  out = dir[3], pdf, sampled_diffuse
void ref_phong_sample(int n, const float* mat9, const float* wo3, const float* N3, const float* r12, const uint64_t* seed, float* out5) {
	PhongBRDF brdf;
	omp_set_num_threads(1);
	for (int i = 0; i < n; i++) {
		MaterialValues m;
		for (int k = 0; k < 3; k++) { m.Kd[k] = mat9[9 * i + k]; m.Ks[k] = mat9[9 * i + 3 + k]; m.Ne[k] = mat9[9 * i + 6 + k]; }
		engine[0] = pcg32(seed[i]);
		float pdf; bool diff;
		Vector d = brdf.sample(m, Vector(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), Vector(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]), pdf, r12[2 * i], r12[2 * i + 1], diff);
		out5[5 * i] = d[0]; out5[5 * i + 1] = d[1]; out5[5 * i + 2] = d[2]; out5[5 * i + 3] = pdf; out5[5 * i + 4] = diff ? 1.f : 0.f;
	}
}
void ref_phong_eval(int n, const float* mat9, const float* wi3, const float* wo3, const float* N3, float* out3) {
	PhongBRDF brdf;
	for (int i = 0; i < n; i++) {
		MaterialValues m;
		for (int k = 0; k < 3; k++) { m.Kd[k] = mat9[9 * i + k]; m.Ks[k] = mat9[9 * i + 3 + k]; m.Ne[k] = mat9[9 * i + 6 + k]; }
		Vector v = brdf.eval(m, Vector(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), Vector(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), Vector(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]));
		out3[3 * i] = v[0]; out3[3 * i + 1] = v[1]; out3[3 * i + 2] = v[2];
	}
}

// ---- per-(pixel,sample) radiance with the build's seeding rule ------
// For each listed pixel (i,j) and each k in [k0,k1): engine[0] = pcg32(seed_base(p,k)) with
// p = i*W+j, seed = p*65536 + k; 4 camera draws exactly as Raytracer.cpp:1462-1466;
// cam.generateDirection; getColor(r,k,nb_bounces,i,j,n,a,false,false).
// out_rgb[(q*(k1-k0)+(k-k0))*3], out_dxdy[...*2] (the sensor jitter used by the splat).
void ref_getcolor_samples(RefCtx* c, int npix, const int* ij, int k0, int k1, float* out_rgb, float* out_dxdy) {
	Raytracer* rt = c->rt;
	omp_set_num_threads(1);
	const float invmax = rt->invmax;
	for (int q = 0; q < npix; q++) {
		int i = ij[2 * q], j = ij[2 * q + 1];
		uint64_t p = (uint64_t)i * (uint64_t)rt->W + (uint64_t)j;
		for (int k = k0; k < k1; k++) {
			engine[0] = pcg32(p * 65536ull + (uint64_t)k);
			float dx = engine[0]()*invmax - 0.5f;
			float dy = engine[0]()*invmax - 0.5f;
			float dx_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float dy_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float time = rt->s.current_frame;
			Ray r = rt->cam.generateDirection(rt->s.double_frustum_start_t, i, j, time, dx, dy, dx_aperture, dy_aperture, rt->W, rt->H);
			Vector normal, albedo;
			Vector color = rt->getColor(r, k, rt->nb_bounces, i, j, normal, albedo, false, false);
			size_t o = (size_t)q * (size_t)(k1 - k0) + (size_t)(k - k0);
			out_rgb[3 * o] = color[0]; out_rgb[3 * o + 1] = color[1]; out_rgb[3 * o + 2] = color[2];
			if (out_dxdy) { out_dxdy[2 * o] = dx; out_dxdy[2 * o + 1] = dy; }
		}
	}
}

// The denoiser inputs getColor hands back (normalValue / albedoValue of the first hit, Raytracer.cpp:255-258), per sample
// and accumulated as render_image_nopreviz does with has_denoiser (:1631-1645: no splat, count += 1).
void ref_getcolor_samples_aov(RefCtx* c, int npix, const int* ij, int k0, int k1, float* out_rgb, float* out_normal, float* out_albedo) {
	Raytracer* rt = c->rt;
	omp_set_num_threads(1);
	const float invmax = rt->invmax;
	for (int q = 0; q < npix; q++) {
		int i = ij[2 * q], j = ij[2 * q + 1];
		uint64_t p = (uint64_t)i * (uint64_t)rt->W + (uint64_t)j;
		for (int k = k0; k < k1; k++) {
			engine[0] = pcg32(p * 65536ull + (uint64_t)k);
			float dx = engine[0]()*invmax - 0.5f;
			float dy = engine[0]()*invmax - 0.5f;
			float dx_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float dy_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float time = rt->s.current_frame;
			Ray r = rt->cam.generateDirection(rt->s.double_frustum_start_t, i, j, time, dx, dy, dx_aperture, dy_aperture, rt->W, rt->H);
			Vector normal, albedo;
			Vector color = rt->getColor(r, k, rt->nb_bounces, i, j, normal, albedo, false, false);
			size_t o = (size_t)q * (size_t)(k1 - k0) + (size_t)(k - k0);
			for (int a = 0; a < 3; a++) { out_rgb[3 * o + a] = color[a]; out_normal[3 * o + a] = normal[a]; out_albedo[3 * o + a] = albedo[a]; }
		}
	}
}
void ref_render_denoiser_inputs(RefCtx* c, float* imagedouble, float* sample_count, float* albedoImage, float* normalImage) {
	Raytracer* rt = c->rt;
	omp_set_num_threads(1);
	const int W = rt->W, H = rt->H;
	const float invmax = rt->invmax;
	memset(imagedouble, 0, sizeof(float)*(size_t)W*H * 3);
	memset(albedoImage, 0, sizeof(float)*(size_t)W*H * 3);
	memset(normalImage, 0, sizeof(float)*(size_t)W*H * 3);
	memset(sample_count, 0, sizeof(float)*(size_t)W*H);
	for (int i = 0; i < H; i++) for (int j = 0; j < W; j++) {
		uint64_t p = (uint64_t)i * (uint64_t)W + (uint64_t)j;
		for (int k = 0; k < rt->nrays; k++) {
			engine[0] = pcg32(p * 65536ull + (uint64_t)k);
			float dx = engine[0]()*invmax - 0.5f;
			float dy = engine[0]()*invmax - 0.5f;
			float dx_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float dy_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float time = rt->s.current_frame;
			Ray r = rt->cam.generateDirection(rt->s.double_frustum_start_t, i, j, time, dx, dy, dx_aperture, dy_aperture, W, H);
			Vector normal, albedo;
			Vector color = rt->getColor(r, k, rt->nb_bounces, i, j, normal, albedo, false, false);
			int idx = ((H - i - 1)*W + j) * 3;                       // Raytracer.cpp:1632-1645
			imagedouble[idx + 0] += color[0]; imagedouble[idx + 1] += color[1]; imagedouble[idx + 2] += color[2];
			sample_count[(H - i - 1)*W + j] += 1;
			normalImage[idx + 0] += normal[0]; normalImage[idx + 1] += normal[1]; normalImage[idx + 2] += normal[2];
			albedoImage[idx + 0] += albedo[0]; albedoImage[idx + 1] += albedo[1]; albedoImage[idx + 2] += albedo[2];
		}
	}
}

// Full image with the seeding rule above and the reference's own splat arithmetic
// (Raytracer.cpp:1477-1497), visiting pixels row-major and k innermost (single thread).
// imagedouble[W*H*3] (row-flipped as the reference), sample_count[W*H].
void ref_render_seeded(RefCtx* c, float* imagedouble, float* sample_count) {
	Raytracer* rt = c->rt;
	omp_set_num_threads(1);
	const int W = rt->W, H = rt->H;
	const float invmax = rt->invmax;
	const float sigma_filter = rt->sigma_filter;
	const int filter_size = rt->filter_size, filter_total_width = rt->filter_total_width;
	float denom2 = 1.f / (2.*sigma_filter*sigma_filter);
	memset(imagedouble, 0, sizeof(float)*(size_t)W*H * 3);
	memset(sample_count, 0, sizeof(float)*(size_t)W*H);
	for (int i = 0; i < H; i++) for (int j = 0; j < W; j++) {
		uint64_t p = (uint64_t)i * (uint64_t)W + (uint64_t)j;
		int bmin_i = std::max(0, i - filter_size);
		int bmax_i = std::min(i + filter_size, H - 1);
		int bmin_j = std::max(0, j - filter_size);
		int bmax_j = std::min(j + filter_size, W - 1);
		float ratio = 1.f / sum_area_table(&rt->filter_integral[0], filter_total_width, bmin_i - i + filter_size, bmax_i - i + filter_size, bmin_j - j + filter_size, bmax_j - j + filter_size);
		float denom1 = ratio / (sigma_filter*sigma_filter*2.*M_PI);
		for (int k = 0; k < rt->nrays; k++) {
			engine[0] = pcg32(p * 65536ull + (uint64_t)k);
			float dx = engine[0]()*invmax - 0.5f;
			float dy = engine[0]()*invmax - 0.5f;
			float dx_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float dy_aperture = (engine[0]()*invmax - 0.5f) * rt->cam.aperture;
			float time = rt->s.current_frame;
			Ray r = rt->cam.generateDirection(rt->s.double_frustum_start_t, i, j, time, dx, dy, dx_aperture, dy_aperture, W, H);
			Vector normal, albedo;
			Vector color = rt->getColor(r, k, rt->nb_bounces, i, j, normal, albedo, false, false);
			for (int i2 = bmin_i; i2 <= bmax_i; i2++) {
				for (int j2 = bmin_j; j2 <= bmax_j; j2++) {
					float w = fast_exp(-(sqr(i2 - i - dy) + sqr(j2 - j - dx)) *denom2) *denom1;
					imagedouble[((H - i2 - 1)*W + j2) * 3 + 0] += color[0] * w;
					imagedouble[((H - i2 - 1)*W + j2) * 3 + 1] += color[1] * w;
					imagedouble[((H - i2 - 1)*W + j2) * 3 + 2] += color[2] * w;
					sample_count[(H - i2 - 1)*W + j2] += w;
				}
			}
		}
	}
}

// ---- stock entry points, timed (cpu_baseline kind "reference") -------------------------
// Runs the reference's own render_image_nopreviz() (its fastest path) on `threads` OpenMP
// threads; returns wall seconds of the call (includes its prepare_render + tone-map).
double ref_time_render_nopreviz(RefCtx* c, int threads, float* imagedouble_out) {
	Raytracer* rt = c->rt;
	omp_set_num_threads(threads > 64 ? 64 : threads);
	rt->clear_image();
	auto t0 = std::chrono::steady_clock::now();
	rt->render_image_nopreviz();
	auto t1 = std::chrono::steady_clock::now();
	if (imagedouble_out) memcpy(imagedouble_out, &rt->imagedouble[0], sizeof(float)*(size_t)rt->W*rt->H * 3);
	return std::chrono::duration<double>(t1 - t0).count();
}
double ref_time_render_image(RefCtx* c, int threads, float* imagedouble_out, float* sample_count_out) {
	Raytracer* rt = c->rt;
	omp_set_num_threads(threads > 64 ? 64 : threads);
	rt->clear_image();
	rt->stopped = false;
	auto t0 = std::chrono::steady_clock::now();
	rt->render_image();
	auto t1 = std::chrono::steady_clock::now();
	if (imagedouble_out) memcpy(imagedouble_out, &rt->imagedouble[0], sizeof(float)*(size_t)rt->W*rt->H * 3);
	if (sample_count_out) memcpy(sample_count_out, &rt->sample_count[0], sizeof(float)*(size_t)rt->W*rt->H);
	return std::chrono::duration<double>(t1 - t0).count();
}

int ref_max_threads() { return ref_thread_cap(); }

// the figures TriMesh::build_bvh leaves for the GUI (mainApp.cpp:974, 1422-1425): max depth, average depth, node count, largest leaf
void ref_mesh_bvh_figures(RefCtx* c, int obj, float* out4) {
	TriMesh* g = c->rt->s.castToMesh[obj];
	out4[0] = (float)g->bvh_depth; out4[1] = g->bvh_avg_depth; out4[2] = (float)g->bvh_nb_nodes; out4[3] = (float)g->max_bvh_triangles;
}

// an object type outside the hot path (Geometry.h:731-847): a scene holding one makes mipt_upload_scene answer
// MIPT_ERR_UNSUPPORTED, and the USE_MIPT build of the reference keeps its stock loop for it
int ref_add_cylinder(RefCtx* c, const float* A, const float* B, float R) {
	c->rt->s.addObject(new Cylinder(Vector(A[0], A[1], A[2]), Vector(B[0], B[1], B[2]), R));
	return (int)c->rt->s.objects.size() - 1;
}

#ifdef USE_MIPT
// ---- only in oracle/_ref/libptref_mipt.so: the reference compiled with the USE_MIPT switch of integration/use_mipt --------
// Everything above then runs the reference's own classes with libmipt.so under Raytracer::render_image[_nopreviz],
// Scene::intersection and TriMesh::build_bvh.  The functions below expose the members the binding adds, so that a test can
// tell a frame the GPU rendered from one the stock loop rendered.
int ref_mipt_built_with_switch() { return 1; }
int ref_mipt_upload(RefCtx* c) { c->rt->mipt_upload(); return c->rt->mipt_status; }     // what render_image does after prepare_render
int ref_mipt_status(RefCtx* c) { return c->rt->mipt_status; }
int ref_mipt_resident(RefCtx* c) { return c->rt->s.mipt_resident ? 1 : 0; }
void ref_mipt_set_lookahead(RefCtx* c, int n) { c->rt->mipt_lookahead = n; }
void ref_mipt_set_dirty(RefCtx* c) { c->rt->mipt_scene_dirty = true; }
const char* ref_mipt_error(RefCtx* c) { return c->rt->mipt ? mipt_last_error(c->rt->mipt) : "no context"; }
int ref_mipt_stats(RefCtx* c, uint64_t* out6) {     // paths, closest-hit rays, shadow rays, pipeline, passes, replayed any-hit rays
	if (!c->rt->mipt) return MIPT_ERR_NO_SCENE;
	mipt_stats st; int rc = mipt_get_stats(c->rt->mipt, &st);
	if (rc != MIPT_OK) return rc;
	out6[0] = st.paths; out6[1] = st.rays_closest; out6[2] = st.rays_shadow; out6[3] = st.pipeline; out6[4] = st.passes; out6[5] = 0;
	mipt_debug_anyhit_replayed(c->rt->mipt, &out6[5]);
	return MIPT_OK;
}
// Raytracer::render_image() while another thread — the GUI's, in the reference (mainApp.cpp:913) — calls stopRender() as
// soon as `stop_at` samples have been published.  Returns realtime_ray_iter as render_image left it.
}  // extern "C"
#include <thread>
#include <atomic>
#include <unistd.h>
extern "C" {
int ref_mipt_render_image_stop_at(RefCtx* c, int stop_at, float* imagedouble_out, float* sample_count_out) {
	Raytracer* rt = c->rt;
	rt->clear_image();
	rt->stopped = false;
	std::atomic<bool> finished(false);
	std::thread gui([&]() { while (!finished.load()) { if (rt->realtime_ray_iter >= stop_at) { rt->stopRender(); return; } usleep(20); } });
	rt->realtime_ray_iter = 0;
	rt->render_image();
	finished.store(true);
	gui.join();
	memcpy(imagedouble_out, &rt->imagedouble[0], sizeof(float)*(size_t)rt->W*rt->H * 3);
	memcpy(sample_count_out, &rt->sample_count[0], sizeof(float)*(size_t)rt->W*rt->H);
	return rt->realtime_ray_iter;
}
// render_image_nopreviz() with has_denoiser: the three images the denoiser would read (colour and albedo divided by the
// count, normals normalised: Raytracer.cpp:1687-1696)
void ref_mipt_render_nopreviz_denoiser(RefCtx* c, float* imagedouble, float* sample_count, float* albedo, float* normal) {
	Raytracer* rt = c->rt;
	rt->clear_image();
	rt->has_denoiser = true;
	rt->render_image_nopreviz();
	rt->has_denoiser = false;
	const size_t n = (size_t)rt->W * rt->H;
	memcpy(imagedouble, &rt->imagedouble[0], sizeof(float) * n * 3); memcpy(sample_count, &rt->sample_count[0], sizeof(float) * n);
	memcpy(albedo, &rt->albedoImage[0], sizeof(float) * n * 3); memcpy(normal, &rt->normalImage[0], sizeof(float) * n * 3);
}
// the 8-bit frame of the last render (Raytracer::image)
void ref_mipt_get_image_u8(RefCtx* c, unsigned char* out) { memcpy(out, &c->rt->image[0], (size_t)c->rt->W * c->rt->H * 3); }
#endif

}  // extern "C"
