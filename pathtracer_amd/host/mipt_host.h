// mipt_host.h — host side of the drop-in boundary: C++ mirror of the reference's operator
// interface for the hot path (class names, member names and argument meaning follow
// Raytracer.h:25-121, Geometry.h:240-445 / 849-1217 / 1238-1400, TriangleMesh.h:113-258), with
// the radiance loop delegated to libmipt.so through include/mipt.h.
//
// What stays on the host, as in the reference: scene construction, TriMesh::init (axis swap,
// normalisation, BVH build, triangle soup, tangents), Object::build_matrix,
// Raytracer::prepare_render (lattice, per-pixel rotations, filter tables, light constants) and
// the tone map.  What moves to the GPU: everything inside the sample loops of
// render_image / render_image_nopreviz, and Scene::intersection / intersection_shadow.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>
#include <sys/mman.h>
#include <cstdlib>
#include <new>

#include "../../include/mipt.h"

namespace mipt_host {

// bumped by everything that rewrites bulk scene data in place (meshes, texture / environment / background images, MERL tables)
extern uint64_t g_content_epoch;

struct Vector {
	float c[3];
	Vector(float x = 0, float y = 0, float z = 0) { c[0] = x; c[1] = y; c[2] = z; }
	float& operator[](int i) { return c[i]; }
	const float& operator[](int i) const { return c[i]; }
};

struct Texture {                      // BRDF.h:252-426
	Vector multiplier{1, 1, 1};
	size_t W = 0, H = 0;
	std::vector<float> values;
	std::string filename;             // image the values came from ("" / "Null": constant), as written to .scn files
	// Texture::loadColors on an 8-bit RGB image given top row first (what stb_image returns):
	// load_image's row flip (utils.cpp:112-118), /255.f and powf(.,2.2f) (BRDF.h:393-404).
	void loadColorsRGB8(const unsigned char* rgb, int w, int h);
	// Texture::loadNormals (BRDF.h:406-418): (v - 128) normalised, no gamma.
	void loadNormalsRGB8(const unsigned char* rgb, int w, int h);
};

struct BVHNodes { bool isleaf; int fg, fd; float bbox[6]; };   // TriangleMesh.h:6-13 (36 bytes)

// std::vector whose resize() leaves new POD elements uninitialised: the node / index vectors of a 23.7 M-triangle mesh
// are gigabytes that the GPU build fills right away, zeroing them first costs more than the build
// Allocations of 8 MB and more are 2 MB-aligned and marked for transparent huge pages: the threads of TriMesh::init touch
// these buffers for the first time while they fill them, and with 4 KB pages the page faults cost more than the stores
// (90 MB of tangents: 19.6 ms, 311 MB of triangle records: 14 ms on the bench host).
template <class T> struct default_init_allocator : std::allocator<T> {
	template <class U> struct rebind { using other = default_init_allocator<U>; };
	T* allocate(size_t n) {
		const size_t bytes = n * sizeof(T);
		if (bytes >= (size_t)8 << 20) {
			void* p = nullptr;
			if (posix_memalign(&p, (size_t)2 << 20, (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) == 0 && p) { madvise(p, bytes, MADV_HUGEPAGE); return static_cast<T*>(p); }
		}
		void* p = malloc(bytes ? bytes : 1);
		if (!p) throw std::bad_alloc();
		return static_cast<T*>(p);
	}
	void deallocate(T* p, size_t) noexcept { free(p); }
	// (default construction of a trivially copyable element is skipped altogether: `Vector` has a zeroing default constructor, and the
	//  vertex / normal arrays of a 23.7 M-triangle mesh were zero-filled by ONE thread — page faults included — before the threads of the
	//  constructor overwrote them: 70 ms.  Every PodVec in this library is written completely before it is read.)
	template <class U> void construct(U* p) noexcept { if constexpr (!(std::is_trivially_copyable<U>::value && std::is_trivially_destructible<U>::value)) ::new (static_cast<void*>(p)) U; else (void)p; }
	template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using PodVec = std::vector<T, default_init_allocator<T>>;
static_assert(sizeof(BVHNodes) == sizeof(mipt_bvh_node), "BVH node layout");

enum ObjectType { OT_TRIMESH = MIPT_OBJ_TRIMESH, OT_SPHERE = MIPT_OBJ_SPHERE, OT_PLANE = MIPT_OBJ_PLANE };

class Object {                        // Geometry.h:240-735
public:
	virtual ~Object() {}
	Object();
	void build_matrix(float frame = 0.f);   // Geometry.h:322-360 (is_recording = false): scale / translation / rotation of the key frames at `frame`
	// key-framed transforms (Geometry.h:258-320): the value of the last key frame at or after the end, of the first one before
	// the beginning, linear (scale, translation) or quaternion-slerp (rotation) interpolation in between; no key frames: the
	// object's own scale / max_translation / mat_rotation
	float get_scale(float frame) const;
	Vector get_translation(float frame) const;
	void get_rotation(float frame, float out9[9]) const;
	void add_keyframe(int frame);     // Geometry.h:316-320: the current transform becomes the key frame of `frame`
	std::map<float, float> scale_keyframes;
	std::map<float, Vector> translation_keyframes;
	std::map<float, std::array<float, 9>> rotation_keyframes;
	Vector apply_transformation(const Vector& v) const;
	void add_col_texture(const Vector& c) { textures.push_back(constant(c)); }
	void add_col_specular(const Vector& c) { specularmap.push_back(constant(c)); }
	void add_col_roughness(const Vector& c) { roughnessmap.push_back(constant(c)); }
	void add_col_alpha(float c) { alphamap.push_back(constant(Vector(c, c, c))); }
	void add_col_refr(float c) { refr_index_map.push_back(constant(Vector(c, c, c))); }
	void add_col_transp(float c) { transparent_map.push_back(constant(Vector(c, c, c))); }
	void add_col_subsurface(const Vector& c) { subsurface.push_back(constant(c)); }
	void add_null_normalmap() { normal_map.push_back(constant(Vector(0, 0, 1))); }

	std::string name;
	ObjectType type = OT_TRIMESH;
	bool miroir = false, ghost = false, flip_normals = false, interp_normals = true;
	float scale = 1;
	Vector max_translation, rotation_center;
	float mat_rotation[9];
	float trans_matrix[12], inv_trans_matrix[12], rot_matrix[9];
	std::vector<double> merl_data;    // IsoMERLBRDF::data (3 x 90*90*180) when the object's brdf is a MERL table, else empty (PhongBRDF)
	std::vector<Texture> textures, specularmap, alphamap, roughnessmap, normal_map, subsurface, transparent_map, refr_index_map;
private:
	static Texture constant(const Vector& c) { Texture t; t.multiplier = c; return t; }
};

class Sphere : public Object {        // Geometry.h:849-1103
public:
	Sphere(const Vector& origin, float rayon);
	void load_envmap_rgb8(const unsigned char* rgb, int w, int h);   // Sphere::load_envmap (:912-916), rows as in the file
	Vector O; float R = 0; bool has_envmap = false;
	std::vector<unsigned char> envtex; int envW = 0, envH = 0;
	std::string envmapfilename;
};

class Plane : public Object {         // Geometry.h:1127-1217
public:
	Plane(const Vector& A, const Vector& N);
	Vector A, vecN;
};

class TriMesh : public Object {       // TriangleMesh.h:113-258
public:
	// TriMesh::init (TriangleMesh.cpp:718-841) on in-memory OBJ arrays (what readOBJ would have
	// parsed): scaling = 1, offset = 0, preserve_input = false.
	TriMesh(int nv, const float* verts, int nn, const float* normals, int nt, const float* uvs,
	        int nf, const int* fv, const int* fn, const int* ft, bool center);
	// TriMesh(scene, obj, 1, (0,0,0), false, NULL, false, center) (TriangleMesh.cpp:714-716): readOBJ + MTL + init.
	// `loaded` is false (and load_error says why) when the file cannot be read or holds no face.
	TriMesh(const char* obj, bool center, bool load_textures = true);
	bool is_centered = true;
	bool loaded = true;
	std::string load_error;
	std::map<std::string, int> groupNames;   // usemtl name -> material group (TriangleMesh.h:228)
	PodVec<Vector> vertices, normals, uvs;    // (not zero-filled on resize: the constructors write every element on their threads)
	PodVec<mipt_triangle_indices> indices;
	PodVec<mipt_triangle> triangleSoup;       // (not zero-filled on resize: TriMesh::init writes every record on its threads)
	PodVec<Vector> tangentSoup;
	PodVec<int> permuted_triangle_index;
	struct { float bbox[6]; PodVec<BVHNodes> nodes; } bvh;
	float bbox[6];
	int bvh_builder = 0;                 // who built bvh.nodes: 0 = the host recursion, 1 = mipt_build_bvh on the GPU, 2 = mipt_device_mesh_build (tree + records stay on the device)
	double bvh_build_seconds = 0, bvh_device_seconds = 0;
	// Round 4: with builder 2 the tree, the reordered Triangle records and the tangents exist on the DEVICE after init; the members
	// above that mirror the reference's host arrays — bvh.nodes, triangleSoup, tangentSoup, the REORDERED indices and
	// permuted_triangle_index — are views that sync_host() downloads / derives on first use (until then `indices` is in input order and
	// the others are empty).  Everything inside this library that reads them calls sync_host() first; so must outside code.
	~TriMesh();
	void sync_host();                    // materialise bvh.nodes, the permutation, the reordered indices and triangleSoup (no-op when they are current)
	void sync_tangents();                // + tangentSoup (downloaded from the device, or setup_tangents on the host)
	int node_count() const { return device_mesh ? device_nodes : (int)bvh.nodes.size(); }
	const mipt_device_mesh* device_handle() const { return device_mesh; }
private:
	mipt_device_mesh* device_mesh = nullptr;
	int device_nodes = 0;
	bool host_views_current = true, tangents_current = true;
	bool build_device_resident();
	void build_triangle_soup();
	bool bvh_gpu_unavailable = false;
	bool build_bvh_gpu();
	bool readOBJ(const char* obj, bool load_textures);
	void add_default_group_materials(int ngroups);
	void finish_init(bool center);
	void build_bbox(int i0, int i1, float* out6) const;
	void build_centers_bbox(int i0, int i1, float* out6) const;
	float split_cost(int i0, int i1, int split_dim, float split_val) const;
	void build_bvh_recur(PodVec<BVHNodes>& out, int i0, int i1, int depth);
	void setup_tangents();
};

struct Camera {                       // Vector.h:700-842 (fields the path reads)
	Vector position{0, 0, 50}, direction{0, 0, -1}, up{0, 1, 0};
	float fov = 0, focus_distance = 50, aperture = 0.1f;
	// lenticular prints: pixel column j is seen from one of lenticular_nb_images shifted cameras (Vector.h:720-723, 799-812)
	bool is_lenticular = false;
	int lenticular_nb_images = 10, lenticular_pixel_width = 1;
	float lenticular_max_angle = (float)(35 * 3.14159265358979323846 / 180. * 0.25);
	// camera array (light-field renders): driven by the caller, one render per view; kept for the scene files
	bool isArray = false; int nbviewX = 1, nbviewY = 1; float maxSpacingX = 0, maxSpacingY = 0;
};

class Raytracer;

class Scene {                         // Geometry.h:1238-1400
public:
	~Scene();
	void addObject(Object* o) { objects.push_back(o); }
	void prepare_render();            // build_matrix on every object (Geometry.cpp:280-284)
	// Scene::intersection / intersection_shadow (Geometry.h:1340-1344) through mipt_trace*.
	bool intersection(const mipt_ray& d, Vector& P, int& sphere_id, float& min_t, mipt_hit& mat, int& triangle_id) const;
	bool intersection_shadow(const mipt_ray& d, float& min_t, float dist_light) const;
	// Scene::background (Geometry.h:1348-1367): the photo behind ghost objects.  load_background decodes the file
	// (load_image: rows flipped) and stores pow(v/255., gamma) * 196964.699; false + reason when the file cannot be read.
	void clear_background() { background.clear(); backgroundW = backgroundH = 0; backgroundfilename.clear(); }
	bool load_background(const char* filename, float gamma, std::string& why);
	// fog (Geometry.h:1371-1377)
	float fog_density = 0, fog_absorption = 0, fog_density_decay = 0, fog_absorption_decay = 0, phase_aniso = 0;
	int fog_type = 0, fog_phase_type = 0;
	std::vector<float> background;
	int backgroundW = 0, backgroundH = 0;
	std::string backgroundfilename;
	std::vector<Object*> objects;
	Sphere* lumiere = nullptr;
	float intensite_lumiere = 0, envmap_intensity = 1;
	float double_frustum_start_t = 0;
	int current_frame = 0, nbframes = 1;
	Raytracer* owner = nullptr;
};

class Raytracer {                     // Raytracer.h:25-121
public:
	Raytracer();
	~Raytracer();
	void loadScene();                 // Raytracer.cpp:1238-1274
	bool load_scene(const char* filename, const char* replacedNames = nullptr);   // Raytracer.cpp:1148-1236 (.scn text format; replacedNames: Geometry.h:524-526); false + last_error() on refusal
	bool save_scene(const char* filename) const;   // Raytracer.cpp:1096-1145
	void prepare_render(float time);  // Raytracer.cpp:1321-1391 + scene upload
	void render_image();              // Raytracer.cpp:1424-1563: progressive, one pass per sample
	void render_image_nopreviz();     // Raytracer.cpp:1565-1718: offline, image divided by sample_count
	void clear_image();
	void stopRender() { stopped = 1; }
	// render_image / render_image_nopreviz upload the scene only when it changed since the last upload: a fingerprint of
	// every description the ABI receives (object parameters, matrices, material lists, array addresses and sizes) plus a
	// counter that the mutators of bulk data bump (meshes, texture / environment / background images, MERL tables).
	// Code that edits such arrays in place through the public members calls scene_changed().
	void scene_changed() { uploaded_ = false; }
	// the GPU this Raytracer renders on, and its share of the image (multi-GPU: one process per GPU)
	int open_device(int device_id);
	int open_devices(const int* device_ids, int n);   // mipt_create(device_ids, n)
	int upload_scene_if_changed();    // mipt_upload_scene unless the resident copy is current; returns the mipt status
	void set_partition(int tile_size, int rank, int nranks) { tile_size_ = tile_size; tile_rank_ = rank; tile_nranks_ = nranks; }
	const char* last_error() const;
	void set_error(const std::string& e) { err_ = e; }

	int W = 1000, H = 800;
	int nrays = 100, last_nrays = -1;
	Camera cam;
	float sigma_filter = 0.5f, lastfilter = -1;
	int filter_size = 0, filter_total_width = 0;
	int nb_bounces = 3;
	float gamma = 2.2f;
	Scene s;
	volatile int stopped = 0;
	int current_nb_rays = 0;
	int progressive_lookahead = 0;     // render_image: one-sample publishes rendered per pass, 0 = the library's choice (not a member of the reference: its loop renders them one by one)
	std::vector<unsigned char> image;
	std::vector<float> imagedouble;
	std::vector<float> sample_count;
	// has_denoiser (Raytracer.h:88): render_image_nopreviz accumulates without the splat and fills the denoiser's
	// auxiliary images (Raytracer.cpp:1631-1645, 1676-1696).  albedoImage = mean Kd of the first hits;
	// normalImage = what the reference computes (the colour sums, normalised: it adds imagedoublethreads at :1680),
	// shadingNormalImage = the normalised sum of the first hits' shading normals (what :1680 presumably meant).
	bool has_denoiser = false;
	std::vector<float> albedoImage, normalImage, shadingNormalImage;
	std::vector<float> filter_value, filter_integral;
	std::vector<Vector> samples2d, randomPerPixel;
	Vector centerLight;
	float lum_scale = 1, radiusLight = 0, lightPower = 0;
	uint64_t seed_stride = 65536;

	// the C-ABI view of the current state (valid after prepare_render)
	mipt_ctx* ctx = nullptr;
	mipt_scene_desc scene_desc{};
	mipt_render_params render_params{};
	int last_status = 0;
private:
	void build_descs();
	void tone_map(bool divide_by_count);
	std::vector<mipt_object> desc_objects_;
	std::vector<mipt_mesh> desc_meshes_;
	std::vector<std::vector<mipt_texture>> desc_tex_;
	int tile_size_ = 32, tile_rank_ = 0, tile_nranks_ = 1;
	bool uploaded_ = false;           // the device holds a scene uploaded from this object
	uint64_t uploaded_fingerprint_ = 0;
	uint64_t scene_fingerprint() const;
	std::string err_;
};

}  // namespace mipt_host

// ---- flat C view for ctypes (tests, bench.py) --------------------------------------------
extern "C" {
typedef struct mh_raytracer mh_raytracer;
mh_raytracer* mh_create(void);                           // new Raytracer + loadScene()
void mh_destroy(mh_raytracer*);
int  mh_open_device(mh_raytracer*, int device_id);        // mipt_create; returns mipt status
int  mh_open_devices(mh_raytracer*, const int* device_ids, int n);   // mipt_create(device_ids, n): n > 1 = every render call uses all of them
void mh_set_partition(mh_raytracer*, int tile_size, int rank, int nranks);
void mh_set_render(mh_raytracer*, int W, int H, int nrays, int nb_bounces, float sigma_filter);
void mh_set_camera(mh_raytracer*, const float* pos, const float* dir, const float* up, float fov, float focus, float aperture);
void mh_set_light(mh_raytracer*, const float* center, float R, float intensite_lumiere);
void mh_set_envmap_intensity(mh_raytracer*, float v);
int  mh_read_image(const char* file, unsigned char* rgb_out, int capacity, int* W, int* H, char* err, int errlen);   // PPM / PNG as stb_image delivers them (3 channels)
int  mh_load_scene(mh_raytracer*, const char* scn_file);   // Raytracer::load_scene; 0 or -1 (mh_last_error)
void mh_set_frame(mh_raytracer*, int frame);                 // Scene::current_frame: the time key-framed transforms are evaluated at
void mh_add_keyframe(mh_raytracer*, int obj, int frame);      // Object::add_keyframe(frame)
void mh_set_object_transform(mh_raytracer*, int obj, const float* translation3, const float* rotation9, float scale);   // max_translation, mat_rotation, scale
int  mh_load_scene_subst(mh_raytracer*, const char* scn_file, const char* replacedNames);   // Raytracer::load_scene(filename, replacedNames): the '#' of mesh names
int  mh_save_image(const char* file, const unsigned char* rgb, int W, int H, char* err, int errlen);   // save_image (utils.cpp:178-234) for 8-bit RGB: .png / .bmp / .tga / .ppm by extension; -1 + text otherwise
int mh_save_image_f32(const char* file, const float* rgb, int W, int H, float maxval, char* err, int errlen);   // save_image<float>: .hdr (Radiance RGBE) keeps the floats
int mh_image_format_supported(const char* file, int for_float);   // a writer exists for this output name (name only: nothing is created)
int  mh_save_scene(mh_raytracer*, const char* scn_file);   // Raytracer::save_scene
int  mh_num_objects(mh_raytracer*);
void mh_get_scene_header(mh_raytracer*, float* out32);     // W,H,nrays,bounces, cam pos/dir/up, fov, focus, aperture, sigma, gamma, intensite_lum, intensite_envmap, frustum t
void mh_get_object_state(mh_raytracer*, int obj, float* out24, int* flags8);   // translation 3, rotation 9, center 3, scale, sphere O 3 + R / plane A 3 + N 3 ; type, miroir, ghost, flip, interp, has_envmap
int  mh_add_mesh_obj(mh_raytracer*, const char* obj_file, float scale, int center);   // TriMesh(&s, file, ...) + GUI placement; -1 on failure (mh_last_error)
void mh_get_group_material(mh_raytracer*, int obj, int grp, float* out12, int* wh8);
int  mh_num_groups(mh_raytracer*, int obj);
const float* mh_group_texture_values(mh_raytracer*, int obj, int grp, int slot);
int  mh_add_mesh(mh_raytracer*, int nv, const float* verts, int nn, const float* normals, int nt, const float* uvs,
                 int nf, const int* fv, const int* fn, const int* ft, float scale, int center);
void mh_set_obj_slicing(int slice_bytes, int max_slices);    // test hook: readOBJ parses slices of the file concurrently (default 1 MiB, one per hardware thread)
void mh_set_bvh_builder(int mode, int device);              // 0 host recursion, 1 GPU, 2 GPU if present else host (default)
void mh_set_device_resident(int on);                        // GPU builder: 1 (default) the tree and the records stay on the device (mipt_device_mesh_build), 0 fetched back (mipt_build_bvh)
int  mh_mesh_bvh_builder(mh_raytracer*, int obj, double* seconds, double* device_seconds);   // 0 host / 1 GPU built this mesh
void mh_set_build_thresholds(int fork_tris, int planes_tris);   // test hook: when the (tree-identical) parallel BVH build forks
void mh_set_object_flags(mh_raytracer*, int obj, int miroir, int flip_normals);
void mh_set_group_material(mh_raytracer*, int obj, int grp, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr);
int  mh_add_sphere(mh_raytracer*, const float* O3, float R, int mirror, int flip_normals);   // another Sphere object (Geometry.h:849-873); returns its index
void mh_add_group_material(mh_raytracer*, int obj, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr);
void mh_set_group_texture(mh_raytracer*, int obj, int grp, int slot, int W, int H, const unsigned char* rgb);
void mh_set_envmap(mh_raytracer*, int W, int H, const unsigned char* rgb);
void mh_set_brdf_merl(mh_raytracer*, int obj, const double* table);   // objects[obj]->brdf = new IsoMERLBRDF(...) (mainApp.cpp:2436)
int  mh_set_brdf_merl_file(mh_raytracer*, int obj, const char* merl_binary_file);   // IsoMERLBRDF(file): read_brdf (MERLBRDFRead.cpp:212-236)
const double* mh_merl_data(mh_raytracer*, int obj);      // IsoMERLBRDF::data of the object, or null
int  mh_prepare(mh_raytracer*, int upload);               // prepare_render; upload=0 skips the device (CPU tests)
int  mh_render_image(mh_raytracer*);
void mh_set_progressive_lookahead(mh_raytracer*, int n);
int  mh_render_image_nopreviz(mh_raytracer*);
const char* mh_last_error(mh_raytracer*);
// views
void* mh_ctx(mh_raytracer*);                              // mipt_ctx*
const void* mh_scene_desc(mh_raytracer*);                 // const mipt_scene_desc*
const void* mh_render_params(mh_raytracer*);              // const mipt_render_params*
float* mh_imagedouble(mh_raytracer*);
int  mh_get_background(mh_raytracer*, float* out, int capacity, int* W, int* H);
void mh_set_lenticular(mh_raytracer*, int on, int nb_images, float max_angle, int pixel_width);   // Camera::is_lenticular & co
void mh_set_has_denoiser(mh_raytracer*, int on);
void mh_set_fog(mh_raytracer*, float density, float absorption, float density_decay, float absorption_decay, int type, int phase_type, float phase_aniso);   // Scene::fog_*
void mh_add_col_subsurface(mh_raytracer*, int obj, const float* rgb);             // Object::add_col_subsurface
void mh_set_group_subsurface(mh_raytracer*, int obj, int grp, const float* rgb);   // Object::subsurface[grp].multiplier
void mh_set_object_ghost(mh_raytracer*, int obj, int ghost);                 // Object::ghost
int  mh_load_background(mh_raytracer*, const char* file);                     // Scene::load_background(file, gamma); -1 + mh_last_error on failure
void mh_set_background(mh_raytracer*, const float* rgb, int W, int H);        // Scene::background set directly (NULL clears it)
float* mh_denoiser_image(mh_raytracer*, int which);          // 0 albedoImage, 1 normalImage (as the reference), 2 shadingNormalImage
float* mh_sample_count(mh_raytracer*);
unsigned char* mh_image(mh_raytracer*);
// dumps with the layouts of oracle/ref_harness.cpp (so one test body serves all three)
void mh_get_light(mh_raytracer*, float* out5);
void mh_get_tables(mh_raytracer*, float* randomPerPixel, float* samples2d, float* filter_integral, int* filter_size);
void mh_get_object_matrices(mh_raytracer*, int obj, float* trans12, float* inv12, float* rot9);
void mh_mesh_counts(mh_raytracer*, int obj, int* ntri, int* nnodes, int* nverts, int* nnormals, int* nuvs);
void mh_mesh_dump(mh_raytracer*, int obj, int* perm, int* nodes_i, float* nodes_bb, float* soup, int* groups, float* root_bb);
int mh_mesh_tangents(mh_raytracer*, int obj, float* out9_per_triangle);     // TriMesh::tangentSoup (synchronised from the device first); returns the number of floats written (0: a mesh without UVs)
}
