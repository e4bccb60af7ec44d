/* mipt.h — C ABI of the MI355X path-tracing core (libmipt.so).
 *
 * Drop-in boundary for ONE hot path of nbonneel/pathtracer: the per-pixel / per-sample radiance
 * loop (Raytracer::getColor + Scene::intersection[_shadow] + TriMesh BVH traversal +
 * Triangle::intersection + Phong BRDF + camera-ray generation + splat).  Everything else of the
 * reference (GUI, file I/O, BVH *build*, scene editing) stays on the host and stays the caller's.
 *
 * Each entry point cites the reference interface it replaces (file:line into the reference
 * checkout).  The reference has no FFI of its own; the seam is the public surface of
 * `class Raytracer` (Raytracer.h:25-121) and `class Scene` (Geometry.h:1238-1400), the precedent
 * being its USE_EMBREE compile-time switch.  INTEGRATION.md shows the C++ binding a maintainer
 * adds on the reference side.
 *
 * Conventions: plain C, pointers + sizes, no C++/torch types.  Every function returns an int
 * status (MIPT_OK = 0); mipt_last_error() gives the text.  Host arrays passed in are copied
 * before the call returns; the library never keeps a caller pointer.  One context = one host
 * thread at a time; it drives one GPU, or several (mipt_create with n > 1: one worker thread and one
 * stream per device inside the library, partial framebuffers summed by one RCCL reduce); several
 * contexts may coexist (one process per GPU with the caller's own collective is the other way to scale).
 * There is NO CPU fallback: without a usable HIP device mipt_create fails with
 * MIPT_ERR_NO_DEVICE.
 */
#ifndef MIPT_H
#define MIPT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_ABI_VERSION 3

enum {
	MIPT_OK = 0,
	MIPT_ERR_INVALID = 1,      /* bad argument / inconsistent description */
	MIPT_ERR_NO_DEVICE = 2,    /* no HIP device, or device id out of range */
	MIPT_ERR_HIP = 3,          /* a HIP runtime call failed (text in mipt_last_error) */
	MIPT_ERR_UNSUPPORTED = 4,  /* scene uses a feature that is not built (a subsurface colour on a sphere, object types other than the three ...) */
	MIPT_ERR_NO_SCENE = 5,     /* render/trace before mipt_upload_scene */
	MIPT_ERR_CANCELLED = 6     /* *cancel became non-zero between passes (Raytracer::stopRender) */
};

typedef struct mipt_ctx mipt_ctx;

/* ---- scene description: POD mirror of what the path reads -------------------------------- */

/* Texture (BRDF.h:252-426): float RGB, already decoded by the host exactly as
 * Texture::loadColors does (rows flipped, /255.f, powf(.,2.2f)); W == 0 means "constant
 * multiplier" (the reference's "Null" texture).  `values` = &Texture::values[0]. */
typedef struct mipt_texture {
	float multiplier[3];
	int32_t W, H;
	const float* values;      /* W*H*3 floats or NULL */
} mipt_texture;

/* BVHNodesT<float> (TriangleMesh.h:6-13), 36 bytes, byte-compatible: pass &bvh.nodes[0]. */
typedef struct mipt_bvh_node {
	uint8_t isleaf; uint8_t _pad[3];
	int32_t fg, fd;           /* inner: child node indices; leaf: triangle range [fg,fd) */
	float bbox_min[3], bbox_max[3];
} mipt_bvh_node;

/* Triangle (TriangleMesh.h:67-111), 124 bytes, byte-compatible: pass &triangleSoup[0]. */
typedef struct mipt_triangle {
	float A[3], u[3], v[3], N[3];
	float m11, m12, m22, invdetm;
	float uvs[3][2];
	float normals[3][3];
} mipt_triangle;

/* TriangleIndices (TriangleMesh.h:53-65), 44 bytes, byte-compatible: pass &indices[0].
 * The path reads group, uvi/uvj/uvk and ni only. */
typedef struct mipt_triangle_indices {
	int32_t vtxi, vtxj, vtxk;
	int32_t uvi, uvj, uvk;
	int32_t ni, nj, nk;
	int32_t group;
	uint8_t showEdges[3]; uint8_t _pad;
} mipt_triangle_indices;

/* The parts of TriMesh (TriangleMesh.h:113-258) the traversal and getMaterial read. */
typedef struct mipt_mesh {
	int32_t n_triangles, n_nodes, n_uvs;
	const mipt_bvh_node* nodes;              /* bvh.nodes */
	float bvh_bbox_min[3], bvh_bbox_max[3];  /* bvh.bbox */
	const mipt_triangle* triangleSoup;
	const mipt_triangle_indices* indices;
	const float* uvs;                        /* TriMesh::uvs as Vector[n_uvs] (3 floats each) or NULL */
	const float* tangentSoup;                /* Vector[3*n_triangles] or NULL (normal maps only) */
	const struct mipt_device_mesh* device_mesh;   /* ABI 3: NULL, or the handle of mipt_device_mesh_build for THIS mesh: mipt_upload_scene then takes the
	                                            * tree and the triangle records from the device that built them (device-to-device copies) and
	                                            * nodes / triangleSoup / indices may be NULL; n_triangles / n_nodes must be the handle's */
} mipt_mesh;

enum { MIPT_OBJ_TRIMESH = 0, MIPT_OBJ_SPHERE = 1, MIPT_OBJ_PLANE = 2 };   /* ObjectType, Geometry.h:29 */
enum { MIPT_BRDF_PHONG = 0, MIPT_BRDF_MERL = 1 };

/* Object (Geometry.h:240-735) + Sphere (:849-1103) / Plane (:1127-1217) / TriMesh fields. */
typedef struct mipt_object {
	int32_t type;
	int32_t miroir, ghost, flip_normals, interp_normals;   /* ghost (Geometry.h:721): the scene is rendered by the contribution-queue kernel */
	float trans_matrix[12], inv_trans_matrix[12], rot_matrix[9];   /* after Object::build_matrix */
	int32_t brdf_kind;
	const double* merl_data;                 /* IsoMERLBRDF::data (3*90*90*180 doubles) or NULL */
	/* the eight per-group texture lists of Object (Geometry.h:731) */
	int32_t n_textures, n_specularmap, n_alphamap, n_roughnessmap, n_normal_map, n_subsurface, n_transparent_map, n_refr_index_map;
	const mipt_texture *textures, *specularmap, *alphamap, *roughnessmap, *normal_map, *subsurface, *transparent_map, *refr_index_map;
	/* Sphere */
	float O[3], R;
	int32_t has_envmap, envW, envH;
	const uint8_t* envtex;                   /* Sphere::envtex, RGB8, envW*envH*3 */
	/* Plane */
	float A[3], vecN[3];
	/* TriMesh */
	const mipt_mesh* mesh;
} mipt_object;

/* Scene (Geometry.h:1238-1400).  As in Raytracer::loadScene (Raytracer.cpp:1257-1269) object 0
 * is the light sphere (Scene::lumiere), object 1 the environment sphere, object 2.. the rest.  A sphere among
 * "the rest" without material lists of its own (and without the mirror flag) is shaded as the reference shades it: with the
 * material of the last object before it in the list that the ray also hit, at that object's hit point (one `localmat` for
 * the loop of Scene::intersection, Geometry.cpp:596).  Such a scene is rendered by the one-thread-per-sample kernel (about a
 * fifth of the wavefront pipeline's rate); together with subsurface colours it is refused (MIPT_ERR_UNSUPPORTED). */
typedef struct mipt_scene_desc {
	int32_t n_objects;
	const mipt_object* objects;
	/* Scene::background (Geometry.h:1355-1366): the photo camera rays see when they leave the scene and that ghost
	 * objects let through (Raytracer.cpp:260-268, 614-621); backgroundW x backgroundH x 3 floats, rows as in the file,
	 * values as load_background leaves them (pow(v/255, gamma) * 196964.699), or NULL / 0 / 0. */
	const float* background;
	int32_t backgroundW, backgroundH;
	/* Fog (Geometry.h:1371-1377): single scattering in a uniform (fog_type 0) or height-exponential (1) medium with an
	 * isotropic (fog_phase_type 0), Schlick (1, phase_aniso) or Rayleigh (2) phase function — fogContribution,
	 * Raytracer.cpp:45-192.  fog_density <= 1e-8: no fog.  fog_ground_level = objects[2]->get_translation(time)[1], the
	 * height of the floor the exponential medium starts at (Raytracer.cpp:55). */
	float fog_density, fog_absorption, fog_density_decay, fog_absorption_decay, phase_aniso, fog_ground_level;
	int32_t fog_type, fog_phase_type;
} mipt_scene_desc;

/* Per-render inputs: the Raytracer members getColor / render_image read after
 * Raytracer::prepare_render (Raytracer.cpp:1321-1391) has filled its tables. */
typedef struct mipt_render_params {
	int32_t W, H;
	int32_t nrays;                 /* samples per pixel */
	int32_t nb_bounces;
	float cam_position[3], cam_direction[3], cam_up[3];    /* Camera (Vector.h:700-842) */
	float cam_fov, cam_focus_distance, cam_aperture;
	float double_frustum_start_t;  /* Scene::double_frustum_start_t */
	float sigma_filter;
	int32_t filter_size;           /* ceil(2*sigma) */
	const float* filter_integral;  /* (2*filter_size+1)^2 summed-area table (Raytracer.cpp:1358-1369) */
	const float* samples2d;        /* Raytracer::samples2d, Vector[nrays] (3 floats each; x,y used) */
	const float* randomPerPixel;   /* Raytracer::randomPerPixel, Vector[W*H] (3 floats each) */
	float centerLight[3], radiusLight, lightPower;         /* Raytracer.cpp:1377-1380 */
	float envmap_intensity;        /* Scene::envmap_intensity */
	/* Sampling rule (the reference's own stream assignment is an accident of link order and
	 * thread scheduling): sample k of pixel p = i*W+j draws from
	 * pcg32(p*seed_stride + k).  seed_stride = 65536 reproduces the oracle/golden vectors. */
	uint64_t seed_stride;
	int32_t sample_begin, sample_end;   /* render samples k in [begin,end); 0,nrays = all */
	/* Work partition across GPUs (one context per GPU): pixel tiles of tile_size x tile_size,
	 * tile t belongs to rank t % tile_nranks.  tile_nranks = 1 renders everything. */
	int32_t tile_size, tile_rank, tile_nranks;
	/* Lenticular camera (Camera::generateDirection, Vector.h:799-812): pixel column j is seen from one of
	 * lenticular_nb_images cameras shifted along camera_right.  is_lenticular = 0: the pinhole of the other branch. */
	int32_t is_lenticular, lenticular_nb_images, lenticular_pixel_width;
	float lenticular_max_angle;
} mipt_render_params;

typedef struct mipt_ray { float origin[3]; float direction[3]; } mipt_ray;

/* Outputs of Scene::intersection (Geometry.h:1340): has_inter, sphere_id, triangle_id, min_t,
 * P, and the MaterialValues (BRDF.h:7-20). */
typedef struct mipt_hit {
	int32_t has_inter, object_id, triangle_id;
	float t;
	float P[3];
	float shadingN[3], Kd[3], Ks[3], Ne[3], Ke[3];
	int32_t transp;
	float refr_index;
} mipt_hit;

/* Counters and timings of the most recent mipt_render* call. */
typedef struct mipt_stats {
	uint64_t paths;               /* camera paths traced */
	uint64_t rays_closest;        /* Scene::intersection calls */
	uint64_t rays_shadow;         /* Scene::intersection_shadow calls */
	uint64_t mesh_casts_closest;  /* TriMesh::intersection calls */
	uint64_t mesh_casts_shadow;   /* TriMesh::intersection_shadow calls */
	double   render_ms;           /* HIP-event time of the whole call on its stream */
	double   traverse_ms;         /* HIP-event time summed over the launches of the dominant kernel:
	                                 pipeline 0: the per-path kernel; pipeline 1: the traversal kernel — closest-hit
	                                 launches only when traverse_merged == 0, every traversal launch (closest-hit,
	                                 any-hit and the merged any-hit(b) + closest-hit(b+1) launches) when it is 1 */
	double   shadow_ms;           /* pipeline 1, traverse_merged == 0: summed over the any-hit (shadow) launches */
	double   shade_ms;            /* pipeline 1: summed over generate + shade launches */
	uint32_t traverse_launches;
	uint32_t shadow_launches;
	uint32_t passes;
	uint32_t pipeline;            /* pipeline that produced these numbers: 0 per-path kernel, 1 wavefront stages, 2 the contribution-queue
	                                 kernel (scenes with ghost objects, a background photo, fog or subsurface colours) */
	uint32_t traverse_merged;     /* 1 = option "merge_traverse" was in effect (see traverse_ms) */
	uint32_t reserved;            /* contribution-queue pipeline: samples of the last pass that needed more pending contributions than the
	                                 wavefront stages keep per sample and were rendered by the one-thread-per-sample loop instead */
	double   resolve_ms;          /* summed over the splat (resolve) launches */
} mipt_stats;

typedef void (*mipt_progress_cb)(void* user, int samples_done, int samples_total);

/* ---- entry points ------------------------------------------------------------------------ */

/* Opens the n devices `device_ids[0..n)` as ONE context.  Replaces nothing in the reference (it has no device); called
 * from Raytracer::Raytracer().
 *   n == 1  one GPU.
 *   n  > 1  the reference's model — one process, its threads render disjoint pixel batches into per-thread framebuffers
 *           that are summed at the end (Raytracer.cpp:1581-1685) — with devices in place of threads: the scene is
 *           replicated (mipt_upload_scene uploads to every device), mipt_render / mipt_render_device give member i the
 *           tiles t % (n * tile_nranks) == tile_rank * n + i, every member renders all samples of its tiles into its own
 *           full-size partial framebuffer on its own stream and host thread, and ONE reduce (RCCL ncclReduce, sum, fp32,
 *           root = device_ids[0], over xGMI) forms the frame, which is added to the caller's buffer.  Progress calls and
 *           cancellation work as with one device (the reduce then runs after every pass-sized chunk).  The ray-level and
 *           per-sample entry points (mipt_trace*, mipt_sample_*, mipt_render_denoiser_inputs) run on device_ids[0].
 *           RCCL (librccl.so.1) is loaded with dlopen when the first group is created; mipt_group_reduce_kind() tells
 *           whether it is in use.  A device may be listed more than once (a one-GPU box exercising this path): such a
 *           group sums its framebuffers with device copies and adds instead, as does option "reduce" = 2. */
int mipt_create(const int* device_ids, int n, mipt_ctx** out);
void mipt_destroy(mipt_ctx* ctx);
const char* mipt_last_error(const mipt_ctx* ctx);
int mipt_abi_version(void);
int mipt_group_size(const mipt_ctx* ctx);                  /* number of devices behind the context */
const char* mipt_group_reduce_kind(const mipt_ctx* ctx);   /* "" for one device, else how the partial framebuffers are summed */
/* Loads RCCL and runs one single-rank ncclReduce on the context's device (checks the dlopen'ed symbols and their
 * signatures on a box with one GPU).  MIPT_ERR_UNSUPPORTED if the library cannot be loaded. */
int mipt_rccl_selftest(mipt_ctx* ctx);

/* Copies the scene to HBM and converts it to the traversal layout.  Called where
 * the reference calls Scene::prepare_render (Geometry.cpp:280-307), i.e. after every
 * Object::build_matrix, from Raytracer::prepare_render (Raytracer.cpp:1375). */
int mipt_upload_scene(mipt_ctx* ctx, const mipt_scene_desc* scene);

/* The sample loop of Raytracer::render_image (Raytracer.cpp:1444-1531) /
 * render_image_nopreviz (:1581-1685): for every owned pixel and sample: camera jitter,
 * Camera::generateDirection, getColor, Gaussian splat.  ADDS into the caller's host buffers
 * accum_rgb = Raytracer::imagedouble (W*H*3, row-flipped: pixel (i,j) at ((H-i-1)*W+j)*3) and
 * accum_w = Raytracer::sample_count (W*H); the caller zero-fills them as prepare_render does.
 * `cb` (may be NULL) is called after each pass, after the caller's buffers have received the sums so far (the
 * reference's GUI thread reads them while render_image runs); `cancel` (may be NULL) is polled between passes like
 * Raytracer::stopped (Raytracer.cpp:1452): the call then returns MIPT_ERR_CANCELLED with the finished passes in
 * the buffers.  The pass size is min(option "paths_per_pass", remaining samples). */
int mipt_render(mipt_ctx* ctx, const mipt_render_params* p, float* accum_rgb, float* accum_w,
                mipt_progress_cb cb, void* cb_user, volatile int* cancel);

/* Same, but accumulates into a DEVICE buffer of W*H*4 floats ([W*H*3 rgb | W*H weights], same
 * pixel order; on device_ids[0]) on HIP stream `hip_stream` (a hipStream_t, NULL = default stream) and returns
 * without synchronising the host.  One process per GPU: the caller reduces these buffers itself (bench.py: one RCCL
 * all-reduce), which replaces the per-thread buffer sum of Raytracer.cpp:1669-1685.  A group (n > 1) has already
 * summed its members' framebuffers into the buffer when the work queued on `hip_stream` completes. */
int mipt_render_device(mipt_ctx* ctx, const mipt_render_params* p, float* d_accum_rgbw, void* hip_stream);

/* Scene::intersection (Geometry.cpp:589-688) on n rays. */
int mipt_trace(mipt_ctx* ctx, const mipt_ray* rays, int n, mipt_hit* hits);
/* Scene::intersection_shadow (Geometry.cpp:691-744) on n rays; occluded[k] = return value. */
int mipt_trace_shadow(mipt_ctx* ctx, const mipt_ray* rays, const float* dist_light, int n, int32_t* occluded);

/* Parity hook: Raytracer::getColor (Raytracer.cpp:196-664) for pixels pixels_ij[2*q..] = (i,j) and
 * samples k in [k0,k1), WITHOUT the splat.  out_rgb[(q*(k1-k0)+(k-k0))*3], out_dxdy[..*2] = the
 * sensor jitter (dx,dy) the splat would use. */
int mipt_sample_radiance(mipt_ctx* ctx, const mipt_render_params* p, const int32_t* pixels_ij, int npix,
                         int k0, int k1, float* out_rgb, float* out_dxdy);

/* The has_denoiser branch of render_image_nopreviz (Raytracer.cpp:1631-1645, 1676-1683): no splat — every sample adds
 * its colour to its own pixel and 1 to the sample count — plus the two auxiliary images of the denoiser: the sums of
 * getColor's albedoValue (Kd of the first hit) and normalValue (its shading normal), (0,0,0) for a sample that hits
 * nothing (Raytracer.cpp:255-258, 1628).  All four arrays are in the reference's layout (idx = ((H-i-1)*W+j)*3) and are
 * added to, like mipt_render's.  The caller divides by the count / normalises as Raytracer.cpp:1689-1696 does; note
 * that the reference fills normalImage from imagedoublethreads (:1680), this entry returns the normals. */
int mipt_render_denoiser_inputs(mipt_ctx* ctx, const mipt_render_params* p, float* accum_rgb, float* accum_w, float* albedo_rgb, float* normal_xyz);

/* Test aid beside mipt_sample_radiance: getColor's three outputs (colour, normalValue, albedoValue) per (pixel, sample). */
int mipt_sample_denoiser_inputs(mipt_ctx* ctx, const mipt_render_params* p, const int32_t* pixels_ij, int npix, int k0, int k1,
                                float* out_rgb, float* out_normal, float* out_albedo);

/* Work partition used by mipt_render*: the rank in [0, tile_nranks) that renders pixel (i, j) of a
 * W-pixel-wide image, or -1 for bad parameters.  Pure host function (no device needed). */
int mipt_tile_owner(int W, int tile_size, int tile_nranks, int i, int j);

/* Measurement aid (bench.py): achievable HBM read bandwidth of the context's device, a grid-stride sum over `bytes` of
 * device memory with 16-byte loads, `repeats` launches timed with HIP events.  Not part of the reference's surface. */
int mipt_measure_stream_read(mipt_ctx* ctx, uint64_t bytes, int repeats, double* gb_per_s);
/* The same for the traversal kernels' access pattern: `records` 64-byte records (four 16-byte loads per lane, like a fat BVH
 * node) gathered at pseudo-random 64-byte-aligned offsets of a `buffer_bytes` buffer; gb_per_s counts 64 bytes per record.
 * Under `rocprofv3 --pmc FETCH_SIZE` it calibrates that counter for gathers (tools/fetch_calibration.py). */
int mipt_measure_gather_read(mipt_ctx* ctx, uint64_t buffer_bytes, uint64_t records, int repeats, double* gb_per_s);
/* Rate, in 10^9 fetches per second, of DEPENDENT random 64-byte fetches from a table of `table_bytes` (rounded down to a power
   of two of records): every lane of the chip walks a random cycle, four 16-byte loads per step.  The ceiling of the memory
   system behind L2 for the access pattern of a BVH traversal step; bench.py prices the traversal kernel's L2 misses against it.
   (Measurement aid like the two above: no counterpart in the reference.) */
int mipt_measure_dependent_gather(mipt_ctx* ctx, uint64_t table_bytes, int steps, int repeats, double* gfetches_per_s);
/* Nanoseconds a compute unit spends per vector-memory wave-instruction (16-byte loads that hit L1, `active_lanes` of 64 lanes
   active, 7 waves per SIMD issuing).  On this chip the figure barely depends on the load width or on the number of lanes: it
   is an instruction rate, and the persistent traversal kernels run at ~90 % of it; bench.py prices their instruction count
   against it.  (Measurement aid: no counterpart in the reference.) */
int mipt_measure_vmem_issue(mipt_ctx* ctx, int active_lanes, int iters, double* ns_per_instruction_and_cu);

/* Diagnostics of the any-hit stage: the number of shadow rays of the context's last render (pipeline 1) that the order-free
   traversal did not decide itself and handed to the ordered one — rays that found an occluder in a leaf whose box lies within
   0.2 % of the ray's far end (the only ones whose answer can depend on the visiting order of TriMesh::intersection_shadow,
   TriangleMesh.cpp:1239-1319) and rays with an infinite inverse-direction component.  No counterpart in the reference. */
int mipt_debug_anyhit_replayed(mipt_ctx* ctx, uint64_t* out);
/* Which any-hit kernel the shadow rays of the resident scene run on: "order-free", or "ordered: <why>" — a tree handed in through
   mipt_mesh::nodes whose boxes do not nest, are empty or hold a NaN or an infinity keeps the ordered traversal of
   TriMesh::intersection_shadow for every shadow ray (checked on the device at upload).  No counterpart in the reference. */
const char* mipt_debug_anyhit_kind(const mipt_ctx* ctx);

/* TriMesh::build_bvh / build_bvh_recur (TriangleMesh.cpp:878-885, 1029-1130) on the GPU: the same nodes at the same
 * positions of the node vector and the same reordering of the triangles as the reference's serial recursion (node boxes
 * compare equal as floats; a coordinate that is +0 in some vertices and -0 in others may come out with the other sign).
 *   vertices          nverts x 3 floats, as TriMesh::init leaves them (axis swap, centering) before build_bvh
 *   tri_vtx, stride   the three vertex indices of triangle i are the ints at tri_vtx + i*tri_stride_bytes: pass
 *                     &indices[0].vtxi and sizeof(TriangleIndices) (44)
 *   out_nodes         room for node_capacity nodes (2*ntri always suffices); *out_n_nodes receives the count
 *   out_perm          ntri ints: position i of the reordered mesh holds input triangle out_perm[i]; the caller applies
 *                     it to `indices` (and to permuted_triangle_index) exactly as the reference's swaps would have
 *   out_seconds       optional: device time of the build (HIP events), without the transfers
 * No context is needed (the scene does not exist yet when TriMesh::init runs); the text of a failure is
 * mipt_build_bvh_error() (thread local).  MIPT_ERR_NO_DEVICE without a GPU: there is no CPU build behind this entry. */
int mipt_build_bvh(int device_id, const float* vertices, int nverts, const void* tri_vtx, int tri_stride_bytes, int ntri,
                   mipt_bvh_node* out_nodes, int node_capacity, int* out_n_nodes, int32_t* out_perm, double* out_seconds);
const char* mipt_build_bvh_error(void);

/* TriMesh::init's build_bvh AND the records the traversal reads, made on the device and LEFT there (round 4).  The analogue of
 * TriangleMesh.cpp:718-885: build_bvh (as mipt_build_bvh: same tree, same triangle order), then per reordered triangle the Triangle
 * constructor's terms (TriangleMesh.h:70-78) and the gather of corner normals / UVs (:812-829) — written straight into the layout
 * the kernels traverse (64-byte fat nodes holding both children's boxes, 64-byte intersection and shading records) instead of
 * coming back to the host as bvh.nodes / triangleSoup and going up again repacked.  Pass the handle in mipt_mesh::device_mesh.
 *   vertices / normals / uvs   nverts / nnormals / nuvs x 3 floats as TriMesh::init leaves them before build_bvh (normals, uvs: may be 0)
 *   indices                    the TriangleIndices records in INPUT order (the permutation stays on the device)
 * mipt_device_mesh_download gives the reference's views when somebody wants them: bvh.nodes (node_capacity >= info.n_nodes) and the
 * permutation (perm[i] = input triangle at position i); either pointer may be NULL.  The handle belongs to device_id; a context on
 * another device (a group's other members) copies from it peer to peer; a context on the SAME device whose scene has this one mesh
 * renders from the handle's buffers in place (no second copy of 128 bytes per triangle; the scene holds a reference).  The handle is
 * reference-counted: mipt_device_mesh_free drops the caller's reference any time after the last mipt_upload_scene that names it has
 * returned; the buffers go when the last scene using them is replaced or its context destroyed.  Errors: mipt_build_bvh_error(). */
typedef struct mipt_device_mesh mipt_device_mesh;
typedef struct mipt_device_mesh_info {
	int32_t n_triangles, n_nodes, n_inner, device_id;
	int32_t has_tangents, _pad;               /* setup_tangents (TriangleMesh.cpp:572-711) ran on the device too (mesh with UVs and normals) */
	double build_seconds, records_seconds;    /* device time of the tree build / of the record (+ tangent) kernels (HIP events) */
} mipt_device_mesh_info;
int mipt_device_mesh_build(int device_id, const float* vertices, int nverts, const float* normals, int nnormals, const float* uvs, int nuvs,
                           const mipt_triangle_indices* indices, int ntri, mipt_device_mesh** out, mipt_device_mesh_info* info);
int mipt_device_mesh_download(const mipt_device_mesh* m, mipt_bvh_node* nodes, int node_capacity, int32_t* perm);
int mipt_device_mesh_download_tangents(const mipt_device_mesh* m, float* tangent_soup /* 9 floats per triangle: TriMesh::tangentSoup */);
void mipt_device_mesh_free(mipt_device_mesh* m);

/* Statistics of the last render call (rays counted like the oracle does, kernel time from HIP
 * events on the render stream). */
int mipt_get_stats(mipt_ctx* ctx, mipt_stats* out);

/* Tunables (name/value); unknown names return MIPT_ERR_INVALID.
 *   "pipeline"        0 = per-path kernel, 1 = wavefront queues (default); scenes with ghost objects, a background photo, fog or
 *                     subsurface colours always run on the contribution-queue kernel (reported as pipeline 2)
 *   "refill"          pipeline 1: 1 = traversal stages refill idle lanes from the queue (default), 0 = one ray per lane
 *   "merge_traverse"  pipeline 1: 1 = the shadow rays of depth b and the closest-hit rays of depth b+1 share one
 *                     launch of the traversal kernel, 0 = one launch per queue (default; measured equal)
 *   "resolve_slices"  ranks of a tile partition: the splat of a pass runs in this many slices along the sample index, summed in slice
 *                     order (0 = default: 3 / owned fraction of the frame, at most 24 — measured on configs[2] at 4 and 8 ranks; 1 = the single-rank kernel form)
 *   "sort_rays"       pipeline 1: 1 = the closest-hit queue of every depth >= 1 is reordered by direction octant before it is
 *                     traversed (a measured negative, DESIGN.md §4: -11 %; same results), 0 = path-id order (default)
 *   "fast_shade"      pipeline 1: 1 = two-tier shade stage (default), 0 = general shade kernel only
 *   "merl_batch"      pipeline 1, scenes with a measured BRDF: 2 = the general shade tier files the table evaluations of its vertices and a stage
 *                     of its own evaluates them, the tables of sincos / acos / atan2 in LDS (default since round 6: the stage 14 % faster on
 *                     configs[4]; 40 bytes more pass state per path), 1 = filed and run 64 to a trip inside the tier, 0 = every vertex
 *                     evaluates its own (same results in all three)
 *   "refill_threshold", "inner_min"  scheduling parameters of the persistent traversal
 *   "lane_limit"      measurement probe: the persistent traversal hands rays to the first N lanes of a wave only (0 = all 64)
 *   "literal_slab"    test hook: 1 = the persistent traversal evaluates the slab test's early-out chain literally for
 *                     every ray (normally only for rays with a zero direction component)
 *   "queue_wavefront" scenes with ghost objects / a background photo / fog / subsurface colours: 1 = getColor's contribution queue as
 *                     wavefront stages (default), 0 = one thread per sample with the queue in HBM (the round-1 kernel; same results)
 *   "queue_ring"      pending contributions a sample may hold in the wavefront stages of the contribution queue, which is also the memory of its
 *                     ring (default 16, at most 32; 48 bytes each); samples that need more are rendered by the one-thread-per-sample loop with
 *                     the reference's 200-entry ring (same results).  Small values are a test hook for that fallback
 *   "queue_fast_tier" wavefront stages of the contribution queue, scenes without fog and subsurface colours: 1 = the closest-hit list goes
 *                     through a fast tier first and the general build takes what it leaves (default), 0 = general build only (same results)
 *   "reduce"          groups only: 0 = RCCL when its communicators exist (default; a reduce that cannot be enqueued falls back to 2 and
 *                     mipt_group_reduce_kind says why), 1 = RCCL or fail, 2 = device copies + adds
 *   "resolve_packed"  ranks of a tile partition: 1 = a wave of the column-scan splat takes 64 columns that receive something from this rank's
 *                     pixels (default), 0 = 64 adjacent columns of the frame, of which a rank of 8 owns 32 at most (same results)
 *   "resolve_rows"    splat kernel: destination rows per band of the column-scan kernel (default 12; 0 = the per-pixel gather
 *                     kernel, which is also what filter radii other than 1 and 2 use).  Both add in the reference's order
 *   "invalidate_tables" 1 = the prepare_render tables were modified in place: upload them again
 *   "paths_per_pass"  upper bound on paths in flight per pass (default 2^30, ~172 GB of path state).  Whatever its value, a pass
 *                     is sized so that its state fits in ~80 % of the device memory that is free when the render starts
 *                     (hipMemGetInfo) and holds at most 2^31 paths; if the allocation still fails the pass is halved and retried
 *   "pass_memory_limit" test hook: > 0 = size the pass as if only this many bytes were free (0 = ask the device)
 *   "samples_per_pass" > 0: a pass renders at most this many samples per pixel; with 1 and a progress callback
 *                     mipt_render behaves like the sample loop of Raytracer::render_image (the caller's buffers hold the
 *                     running sums after every sample) without paying an upload and an allocation per sample; 0 = off
 *   "progressive_lookahead" renders with a progress callback: publish groups rendered per pass (1 .. 64; 0 = default: as many as make a pass hold 64 M paths); the caller still sees
 *                     exactly the sums through the group it is told about, the snapshots of a pass travel while the next one renders
 *   "anyhit_wide"     1 = shadow rays go through the order-free any-hit traversal over 8-bit four-wide nodes, with the rays whose answer
 *                     could depend on the reference's visiting order replayed by the ordered kernel (default),
 *                     0 = every shadow ray through the ordered kernel (same results).  Scenes whose uploaded tree does not nest use 0
 *   "anyhit_flag_all" test hook: 1 = every occluded shadow ray is replayed in order
 *   "device_mesh_as_remote" test hook: 1 = mipt_upload_scene treats a mipt_device_mesh of its own device as another device's
 *                     (hipMemcpyPeer into its own buffers instead of reading the records in place): what the other members of a group do */
int mipt_set_option(mipt_ctx* ctx, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */
