/* TEST INFRASTRUCTURE ONLY.
 *
 * pt_oracle: a plain-C, CPU-only restatement of the hot path of nbonneel/pathtracer
 * (Raytracer::getColor + Scene::intersection[_shadow] + TriMesh BVH build/traversal +
 * Triangle::intersection + Phong BRDF + camera + samplers + splat).  Every function cites
 * the reference file:line it follows (paths relative to the reference checkout).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped product (pathtracer_amd/, libmipt.so) never includes, links or calls it.
 *
 * Parity status: PINNED — tests/test_oracle_vs_reference.py checks this file bit-for-bit
 * against the compiled reference (oracle/_ref/libptref.so, built from /root/reference by
 * oracle/Makefile) and against the golden vectors in tests/golden/ generated from it.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct o_ctx o_ctx;

o_ctx* o_create(void);                 /* == new Raytracer + loadScene() (Raytracer.cpp:1238-1274) */
void   o_destroy(o_ctx*);
void   o_set_render(o_ctx*, int W, int H, int nrays, int nb_bounces, float sigma_filter);
void   o_set_camera(o_ctx*, const float* pos, const float* dir, const float* up, float fov, float focus, float aperture);
void   o_set_light(o_ctx*, const float* center, float R, float intensite_lumiere);
void   o_set_envmap_intensity(o_ctx*, float v);
int    o_add_sphere(o_ctx*, const float* O3, float R, int mirror, int flip_normals);   /* another Sphere object; returns its index */
/* TriMesh::TriMesh(scene, obj, 1, (0,0,0), false, NULL, false, center) on in-memory OBJ arrays,
 * followed by the GUI placement (mainApp.cpp:2402-2410).  ft/uvs may be NULL. Returns object id. */
int    o_add_mesh(o_ctx*, int nv, const float* verts, int nn, const float* normals, int nt, const float* uvs,
                  int nf, const int* fv, const int* fn, const int* ft, float scale, int center);
/* test hook: triangle counts above which the (tree-identical) parallel build forks subtrees / costs planes concurrently */
void   o_set_build_thresholds(int fork_tris, int planes_tris);
void   o_set_object_flags(o_ctx*, int obj, int miroir, int flip_normals);
void   o_set_group_material(o_ctx*, int obj, int grp, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr);
void   o_add_group_material(o_ctx*, int obj, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr);
/* Texture images: 8-bit RGB rows top-to-bottom as stored in the file; decoded like
 * Texture::loadColors (BRDF.h:393-404) + load_image's row flip (utils.cpp:112-118).
 * slot: 0 Kd, 1 Ks, 3 alpha, 4 Ne (roughness).  */
void   o_set_group_texture(o_ctx*, int obj, int grp, int slot, int W, int H, const unsigned char* rgb);
/* Sphere::load_envmap on object 1 (Geometry.h:912-916): 8-bit RGB, rows as in file */
void   o_set_envmap(o_ctx*, int W, int H, const unsigned char* rgb);
/* IsoMERLBRDF on one object (BRDF.h:192-247): table = 3 x 90*90*180 doubles (MERL .binary payload) */
void   o_set_brdf_merl(o_ctx*, int obj, const double* table);
void   o_merl_eval(const double* table, int n, const float* wi3, const float* wo3, const float* N3, float* out3);
void   o_prepare(o_ctx*);              /* Raytracer::prepare_render (Raytracer.cpp:1321-1391) */

/* dumps (same layouts as oracle/ref_harness.cpp) */
void   o_get_light(o_ctx*, float* out5);
void   o_get_tables(o_ctx*, float* randomPerPixel, float* samples2d, float* filter_integral, int* filter_size);
void   o_get_object_matrices(o_ctx*, int obj, float* trans12, float* inv12, float* rot9);
void   o_mesh_counts(o_ctx*, int obj, int* ntri, int* nnodes, int* nverts, int* nnormals, int* nuvs);
void   o_mesh_dump(o_ctx*, int obj, int* perm, int* nodes_i, float* nodes_bb, float* soup, int* groups, float* root_bb);

/* leaf functions */
void   o_pcg32(uint64_t seed, int n, uint32_t* out);
void   o_lattice(int n, float* out_xy);
void   o_invsqroot(int n, const float* in, float* out);
void   o_fast_normalize(int n, const float* in3, float* out3);
void   o_fast_exp(int n, const double* in, double* out);
void   o_random_cos(int n, const float* N3, const float* r12, float* out3);
void   o_camera_rays(o_ctx*, int n, const int* ij, const float* jit4, float* out6);
void   o_intersect(o_ctx*, int n, const float* rays6, int* out_i, float* out_f);
void   o_intersect_shadow(o_ctx*, int n, const float* rays6, const float* dist_light, int* occluded);
void   o_phong_sample(int n, const float* mat9, const float* wo3, const float* N3, const float* r12, const uint64_t* seed, float* out5);
void   o_phong_eval(int n, const float* mat9, const float* wi3, const float* wo3, const float* N3, float* out3);

/* radiance: per-(pixel,sample) stream pcg32(p*65536+k), p = i*W+j */
void   o_getcolor_samples(o_ctx*, int npix, const int* ij, int k0, int k1, float* out_rgb, float* out_dxdy);
/* diagnostic (tests/tools/sched_sim.py): traversal events of those samples, path after path; returns the bytes needed */
unsigned long long o_trace_samples(o_ctx*, int npix, const int* ij, int k0, int k1, uint8_t* buf, unsigned long long cap);
void   o_render_seeded(o_ctx*, float* imagedouble, float* sample_count);
void   o_set_object_ghost(o_ctx*, int obj, int ghost);
void   o_set_lenticular(o_ctx*, int on, int nb_images, float max_angle, int pixel_width);
void   o_set_group_subsurface(o_ctx*, int obj, int grp, const float* rgb);
void   o_add_col_subsurface(o_ctx*, int obj, const float* rgb);
void   o_set_fog(o_ctx*, float density, float absorption, float density_decay, float absorption_decay, int type, int phase_type, float phase_aniso);
void   o_set_background(o_ctx*, const float* rgb, int W, int H);
void   o_getcolor_samples_aov(o_ctx*, int npix, const int* ij, int k0, int k1, float* out_rgb, float* out_normal, float* out_albedo);
void   o_render_denoiser_inputs(o_ctx*, float* imagedouble, float* sample_count, float* albedo, float* normal);
/* nopreviz-style schedule (4x4 pixel batches, dynamic,1, per-thread framebuffers then serial
 * reduce; Raytracer.cpp:1581-1685) with the seeding rule above.  Returns wall seconds of the
 * render only.  rays_out[2] = closest-hit casts, shadow casts. */
double o_render_omp(o_ctx*, int threads, float* imagedouble, float* sample_count, uint64_t* rays_out);
int    o_max_threads(void);

/* traversal work counters of the reference's ordered traversal:
 * out[0..2] closest-hit: box tests, nodes popped, triangle tests; out[3..5] same for shadow rays;
 * out[6] closest-hit mesh casts; out[7] shadow mesh casts.  Reset with o_counters_reset. */
void   o_counters_reset(void);
void   o_counters_get(uint64_t* out8);

#ifdef __cplusplus
}
#endif
#endif
