// mipt_scene.h — HBM-resident scene layout shared by the host uploader and the kernels.
//
// Layout choices:
//  * BVH: the reference's 36-byte nodes (isleaf, fg, fd, own box) are re-packed at upload into
//    64-byte "fat" nodes that hold BOTH children's boxes and references, because the reference's
//    traversal tests both children of every node it pops (TriangleMesh.cpp:1172-1178): one
//    aligned 64-byte fetch per popped node instead of one 36-byte node + two more for the boxes.
//  * Triangles: 64-byte intersection record (A,u,v,N,m11,m12,m22,invdetm = the 16 floats
//    Triangle::intersection reads, TriangleMesh.h:82-107), separate from the 64-byte shading
//    record (corner normals, UVs, group) that is only fetched once per path vertex.
#pragma once
#include <stdint.h>

#define MIPT_MAX_OBJECTS 65535   // a sanity bound on mipt_scene_desc::n_objects, not a layout limit: the object records are a device array of n_objects entries and a
                                 // hit record names its object by index or by triangle (mipt_trace.h, hit_unpack).  Until round 6: 31 (five bits of the hit word).
#define MIPT_TEX_SLOTS 8
// slot ids = the reference's Texture type codes (BRDF.h:256-264) + 7 for the subsurface list
enum { MT_KD = 0, MT_KS = 1, MT_NORMAL = 2, MT_ALPHA = 3, MT_NE = 4, MT_TRANSP = 5, MT_REFR = 6, MT_KSUB = 7 };

// Child reference of a fat node / traversal stack entry:
//   ref >= 0           : inner node, index into DMesh::nodes (units of 64 B)
//   ref <  0 (bit 31)  : leaf, bits 0..25 = first triangle, bits 26..30 = (count-1)
// A leaf of 32 OR MORE triangles (build_bvh_recur stops splitting when every centre falls on one side, TriangleMesh.cpp:1118: such
// leaves are unbounded) carries 31 in its count field and its real count in the scene's table of fat leaves (DScene / DObject::fat_leaves,
// sorted by first triangle; empty for ordinary meshes).  Until round 6 such a mesh was refused.
#define MIPT_LEAF_BIT 0x80000000u
#define MIPT_LEAF_MAX_TRIS 32            // count field saturates here: >= this many -> look the count up
#define MIPT_LEAF_FIRST_MASK 0x03ffffffu

struct DTex {              // Texture (BRDF.h:252-426)
	float mult[3];
	int W, H;
	int _pad;
	const float* values;
};

struct DFatNode {          // 64 B, 64-B aligned.  Slabs are stored per axis as (min, max) pairs so that one packed
	float l[3][2];         // fp32 instruction (v_pk_add_f32 / v_pk_mul_f32) handles both planes of an axis
	float r[3][2];
	uint32_t lref, rref;
	uint32_t _pad[2];
};

// Intersection record, 64 B.  A test that loads only the first 48 bytes derives N = cross(u, v) and m22 = |v|^2 with the
// operations Triangle's constructor used (TriangleMesh.h:70-78: same bits) and saves one vector-memory instruction.
struct DTriIsect { float A[3], u[3], v[3]; float invdetm, m11, m12, m22; float N[3]; };
#define MIPT_GROUP_UV_OK 0x40000000      // DTriShade::group bit: indices[tri].uvi is a valid UV index
#define MIPT_GROUP_MASK 0x3fffffff
struct DTriShade { float normals[9]; float uvs[6]; int group; };                    // 64 B (group: material group | MIPT_GROUP_UV_OK)

// What Object::queryMaterial (Geometry.h:399-445) returns for ONE material group of one object, flattened at upload: per
// slot the constant colour of its list entry (Texture::multiplier when the entry has no image) or the reference's default
// when the list is shorter than the group index.  Slots whose entry is an image (bit s of image_mask) still go through the
// entry's descriptor.  Round 3: a vertex read its five slots through 25 dependent little loads (list length, list pointer,
// three fields of the entry, per slot) and the CU issues one vector-memory instruction per ~10 ns whatever its width
// (profiles/r3_b_instruction_issue_rates.txt): three 16-byte loads of this record replace them.
struct DGroupMat {
	float Kd[3], Ks[3], Ne[3];
	float transp_val;          // mat.transp = transp_val < 0.5f (default list: 1.0 -> false)
	float refr;                // default 1.3
	uint32_t image_mask;
	const float* kd_values;    // the Kd entry's image when bit MT_KD of image_mask is set (texel * Kd[] = Texture::getVec): no trip through the entry's descriptor
	int kdW, kdH;
};

// The first 64 bytes are what the shade stage needs of an object before it can fetch anything else (four 16-byte loads issued
// together with the three matrices, one dependent round trip instead of six: the stage runs 4 waves per SIMD and waits on its
// chain of dependent loads).
struct alignas(64) DObject {   // (64-byte aligned and sized: the first 64 bytes of every object are ONE cache line and its 16-byte loads are aligned — at 440 bytes every other object's were not, and which ones changed with every field added in front of DScene::obj)
	int type, miroir, flip_normals, interp_normals;   // miroir: bit 0 = Object::miroir, bit 1 = Object::ghost (both reach the shade stage with the first 16 bytes)
	int nuvs, ngroups, ntex_normal, alpha_test;   // ntex_normal = ntex[MT_NORMAL]; alpha_test: an alpha texture list exists and the mesh has UVs (TriangleMesh.cpp:1200)
	const DTriShade* shade; const DGroupMat* gmat; // gmat[ngroups + 1]: groups 0 .. ngroups-1 (ngroups = the longest of the Kd / Ks / Ne / transp / refr lists), then the all-defaults record
	const float* tangent_soup;                     // Vector[3*ntri] or null
	const double* merl;
	float inv[12], trans[12], rot[9];
	int brdf_kind;
	int ntex[MIPT_TEX_SLOTS];
	const DTex* tex[MIPT_TEX_SLOTS];
	// Sphere
	float O[3], R, R2;
	int has_envmap, envW, envH;
	const uint8_t* envtex;
	// Plane
	float A[3], vecN[3];
	// TriMesh
	const DFatNode* nodes;     // = DScene::all_nodes (child references are scene-wide)
	const DTriIsect* tris;     // = DScene::all_tris (leaf references are scene-wide; mesh-local index = i - tri_base)
	uint32_t node_base, tri_base;
	float root_min[3], root_max[3];
	uint32_t root_ref;         // reference of node 0 (inner 0, or a leaf ref when the root is a leaf)
	uint32_t quad_root;        // the same for the any-hit stage: the root's quad node (mipt_anyhit.h), or the leaf ref
	int ntri;
	int ghost;                 // Object::ghost (Geometry.h:721): only the queue kernel (mipt_compositing.h) renders such scenes
	const float* uvs;          // Vector[nuvs]
	const int* uvidx;          // 3 ints per triangle (uvi,uvj,uvk), only when alpha_test
	const uint2* fat_leaves;   // = DScene::fat_leaves (the per-thread traversals reach the table through their object)
	int n_fat_leaves;
};
// triangles of the leaf a child reference names
__host__ __device__ inline int mipt_leaf_count(uint32_t ref, const uint2* fat, int nfat) {
	const int c = (int)((ref >> 26) & 31u) + 1;
	if (c < MIPT_LEAF_MAX_TRIS) return c;
	const uint32_t first = ref & MIPT_LEAF_FIRST_MASK;
	int lo = 0, hi = nfat - 1;
	while (lo <= hi) { const int mid = (lo + hi) >> 1; const uint32_t f = fat[mid].x; if (f == first) return (int)fat[mid].y; if (f < first) lo = mid + 1; else hi = mid - 1; }
	return c;                  // (not in the table: cannot happen for a tree this library encoded)
}
// The same for the persistent traversal kernels, which have no register to spare (one more live value spilled in the any-hit kernel): a
// linear walk over the table with wave-uniform (scalar) loads.  The table is empty for ordinary meshes; a mesh needs 32 triangles of
// one centroid to add an entry.
__device__ __forceinline__ int mipt_leaf_count_scan(uint32_t first, const uint2* __restrict__ fat, int nfat) {
	int c = MIPT_LEAF_MAX_TRIS;
	for (int k = 0; k < nfat; k++) { const uint2 e = fat[k]; if (e.x == first) c = (int)e.y; }
	return c;
}

// load_obj_hot() and the group-table branch of query_material() (mipt_trace.h) read these records with raw 16-byte loads at fixed
// offsets; the host fills them by field name.  A reordered field must fail here, not corrupt every material lookup.
#include <stddef.h>
static_assert(offsetof(DObject, type) == 0 && offsetof(DObject, nuvs) == 16 && offsetof(DObject, shade) == 32 && offsetof(DObject, gmat) == 40 &&
              offsetof(DObject, tangent_soup) == 48 && offsetof(DObject, merl) == 56 && offsetof(DObject, inv) == 64, "DObject: the shade stage loads its first 64 bytes as four 16-byte words");
static_assert(sizeof(DObject) % 64 == 0 && alignof(DObject) == 64, "DObject: whole cache lines, so that DScene::obj[i] keeps the first 64 bytes of every object in one line");
static_assert(sizeof(DGroupMat) == 64 && offsetof(DGroupMat, transp_val) == 36 && offsetof(DGroupMat, image_mask) == 44 && offsetof(DGroupMat, kd_values) == 48 &&
              offsetof(DGroupMat, kdW) == 56, "DGroupMat: one 64-byte record per material group, read as four 16-byte words (image_mask in r2.w)");
static_assert(sizeof(DFatNode) == 64 && sizeof(DTriIsect) == 64 && sizeof(DTriShade) == 64, "64-byte records");

struct DScene {
	int nobj;
	int any_alpha;               // some mesh rejects hits by an alpha map inside its leaf loop (TriangleMesh.cpp:1198-1205)
	int first_mesh;              // index of the first TriMesh object (nobj if none): the objects before it are analytic
	int inherit_material;        // some sphere beyond objects 0 / 1 has no material lists and is not a mirror: Scene::intersection's ONE MaterialValues for all objects
	                             // of its loop decides what it is shaded with (Geometry.cpp:596); such scenes are rendered by the one-thread-per-sample kernel (mipt_trace.h)
	unsigned merl_mask;          // bit i: object i < 32 carries a measured BRDF (the fast shade tier hands its hits on before it looks at their material; objects >= 32: DObject::brdf_kind)
	int n_meshes;                // TriMesh objects of the scene
	const DFatNode* all_nodes;   // fat nodes of every mesh, one buffer (wave-uniform base for the persistent traversal)
	const DTriIsect* all_tris;   // intersection records of every mesh, one buffer
	const uint2* fat_leaves;     // (first triangle, count) of every leaf with >= MIPT_LEAF_MAX_TRIS triangles, ascending; usually empty
	int n_fat_leaves;
	const uint2* mesh_first;     // n_meshes entries in object order = ascending first triangle: (first triangle of the mesh in all_tris, its object index)
	DObject obj[];               // nobj records BEHIND the header, in the same allocation (the reference's Scene::objects is an unbounded vector, Geometry.h:1306-1309;
	                             // until round 6: obj[31]).  Not a pointer to a second buffer: the kernels take the scene as `const DScene* __restrict__`, and only
	                             // what is reached from THAT pointer by address arithmetic is known not to alias their stores — behind a loaded pointer the
	                             // object records were re-read after every store (generate + shade +30 % on configs[2], measured)
};
static_assert(offsetof(DScene, obj) % 64 == 0, "the object records start on a cache line");

// Per-render constants (kernel argument, by value).
struct DRender {
	int W, H, nrays, nb_bounces;
	float cam_pos[3], cam_dir[3], cam_up[3], cam_right[3];
	float cam_k;               // W / (2*tan(fov/2)) (Vector.h:793), evaluated on the host
	int lent_on, lent_nb, lent_pw; float lent_L;   // lenticular camera (Vector.h:799-812); L = focus*tan(max_angle/2)/(nb/2.0), on the host
	float focus, aperture, init_t;
	float centerLight[3], radiusLight, lightPower, envmap_intensity;
	float sigma_filter; int filter_size;
	const float* filter_integral;
	const float* samples2d;      // nrays * 2
	const float* randomPerPixel; // W*H * 2
	uint64_t seed_stride;
	const float* background;     // Scene::background (Geometry.h:1355-1366): backgroundW x backgroundH x 3 floats or null
	int backgroundW, backgroundH;
	// Scene::fog_* (Geometry.h:1371-1377) and the height of the floor, objects[2]->get_translation()[1] (Raytracer.cpp:55)
	float fog_density, fog_absorption, fog_density_decay, fog_absorption_decay, phase_aniso, ground_level;
	int fog_type, fog_phase_type;
};

// One render pass: samples [k0,k1) of every owned 8x8 pixel block.
struct DPass {
	int k0, k1;
	int nblocks;               // owned 8x8 blocks
	const int* blocks;         // (i0, j0) per block
	const int* pix2slot;       // W*H: slot of the pixel inside this rank's block list, -1 if not owned
	int npix_slots;            // nblocks * 64
	const int* dest;           // pixels that receive a splat from an owned pixel (owned pixels dilated by the
	int ndest;                 // filter radius), or null = every pixel of the image (single rank)
	const int* scan_off;       // ranks of a partition, column-scan splat: the destination columns of band b that hold a pixel of `dest` are
	const int* scan_cols;      // scan_cols[scan_off[b] .. scan_off[b + 1]) (a wave then scans 64 columns WITH work); null = every column
};
