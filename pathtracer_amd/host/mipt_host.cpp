// mipt_host.cpp — see mipt_host.h.  Host-side restatement of the parts of the reference that stay
// on the CPU (file:line citations are into the reference checkout).  Compile with
// -ffp-contract=off: several results (BVH splits, matrices, tables) must match the reference's
// float arithmetic exactly because the device path consumes them.
#include "mipt_host.h"
#include "mipt_jpeg.h"
#include "mipt_imgwrite.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <array>
#include <mutex>
#include <memory>
#include <chrono>
#include <thread>
#include <zlib.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace mipt_host {

// ---------------------------------------------------------------- small vector helpers (Vector.h)
static inline Vector add(const Vector& a, const Vector& b) { return Vector(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
static inline Vector sub(const Vector& a, const Vector& b) { return Vector(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
static inline Vector mul(float a, const Vector& b) { return Vector(a * b[0], a * b[1], a * b[2]); }
static inline Vector divs(const Vector& a, float b) { return Vector(a[0] / b, a[1] / b, a[2] / b); }
static inline float dot(const Vector& a, const Vector& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline float norm2(const Vector& a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }
static inline Vector cross(const Vector& a, const Vector& b) { return Vector(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]); }
static inline Vector normalized(const Vector& a) { float n = std::sqrt(norm2(a)); return Vector(a[0] / n, a[1] / n, a[2] / n); }

// ---------------------------------------------------------------- Texture
uint64_t g_content_epoch = 1;

void Texture::loadColorsRGB8(const unsigned char* rgb, int w, int h) {
	g_content_epoch++;
	W = (size_t)w; H = (size_t)h;
	values.resize(W * H * 3);
	for (int i = 0; i < h; i++)
		for (int j = 0; j < w; j++)
			for (int k = 0; k < 3; k++) {
				float v = rgb[((size_t)(h - 1 - i) * w + j) * 3 + k];   // load_image flips the rows (utils.cpp:112-118)
				v /= 255.f;
				v = std::pow(v, 2.2f);                                    // BRDF.h:399-400
				values[((size_t)i * w + j) * 3 + k] = v;
			}
}

void Texture::loadNormalsRGB8(const unsigned char* rgb, int w, int h) {
	g_content_epoch++;
	W = (size_t)w; H = (size_t)h;
	values.resize(W * H * 3);
	for (int i = 0; i < h; i++)
		for (int j = 0; j < w; j++) {
			const unsigned char* px = rgb + ((size_t)(h - 1 - i) * w + j) * 3;
			Vector v((float)px[0] - 128, (float)px[1] - 128, (float)px[2] - 128);
			v = normalized(v);
			for (int k = 0; k < 3; k++) values[((size_t)i * w + j) * 3 + k] = v[k];
		}
}

// ---------------------------------------------------------------- Object
Object::Object() {
	for (int i = 0; i < 9; i++) mat_rotation[i] = (i % 4 == 0) ? 1.f : 0.f;   // Matrix() is the identity (Vector.h:88-94)
	memset(trans_matrix, 0, sizeof trans_matrix); memset(inv_trans_matrix, 0, sizeof inv_trans_matrix); memset(rot_matrix, 0, sizeof rot_matrix);
}

// Key frames (Geometry.h:258-320).  std::map::upper_bound and the arithmetic are the reference's; the rotation goes through
// Matrix::toQuaternion / Slerp / fromQuaternion (Vector.h:104-158, 223-269) with T = float: the unsuffixed literals make the
// square roots and the products of fromQuaternion double expressions narrowed on assignment, acos / sin of a float are the
// float overloads.
float Object::get_scale(float frame) const {
	auto it1 = scale_keyframes.upper_bound(frame);
	auto it2 = it1;
	if (it1 == scale_keyframes.end()) return scale_keyframes.size() != 0 ? scale_keyframes.rbegin()->second : scale;
	if (it1 == scale_keyframes.begin()) return it1->second;
	it1--;
	const float t = (frame - it1->first) / (it2->first - it1->first);
	return (1.f - t) * it1->second + t * it2->second;
}
Vector Object::get_translation(float frame) const {
	auto it1 = translation_keyframes.upper_bound(frame);
	auto it2 = it1;
	if (it1 == translation_keyframes.end()) return translation_keyframes.size() != 0 ? translation_keyframes.rbegin()->second : max_translation;
	if (it1 == translation_keyframes.begin()) return it1->second;
	it1--;
	const float t = (frame - it1->first) / (it2->first - it1->first);
	const float a = 1 - t;                                      // (1 - t) * v1 + t * v2, component-wise (Vector.h:499, 491)
	return Vector(a * it1->second[0] + t * it2->second[0], a * it1->second[1] + t * it2->second[1], a * it1->second[2] + t * it2->second[2]);
}
namespace {
struct Quat { float w, x, y, z; };
Quat to_quaternion(const float* v) {                            // Matrix::toQuaternion: (*this)(i, j) = values[i * 3 + j]
	const float m00 = v[0], m01 = v[3], m02 = v[6], m10 = v[1], m11 = v[4], m12 = v[7], m20 = v[2], m21 = v[5], m22 = v[8];
	const float tr = m00 + m11 + m22;
	Quat q;
	if (tr > 0) {
		const float S = (float)(std::sqrt(tr + 1.0) * 2);
		q.w = (float)(0.25 * S); q.x = (m21 - m12) / S; q.y = (m02 - m20) / S; q.z = (m10 - m01) / S;
	} else if ((m00 > m11) & (m00 > m22)) {
		const float S = (float)(std::sqrt(1.0 + m00 - m11 - m22) * 2);
		q.w = (m21 - m12) / S; q.x = (float)(0.25 * S); q.y = (m01 + m10) / S; q.z = (m02 + m20) / S;
	} else if (m11 > m22) {
		const float S = (float)(std::sqrt(1.0 + m11 - m00 - m22) * 2);
		q.w = (m02 - m20) / S; q.x = (m01 + m10) / S; q.y = (float)(0.25 * S); q.z = (m12 + m21) / S;
	} else {
		const float S = (float)(std::sqrt(1.0 + m22 - m00 - m11) * 2);
		q.w = (m10 - m01) / S; q.x = (m02 + m20) / S; q.y = (m12 + m21) / S; q.z = (float)(0.25 * S);
	}
	return q;
}
Quat slerp(Quat q1, Quat q2, float t) {                         // Vector.h:223-258
	float w2 = q2.w, x2 = q2.x, y2 = q2.y, z2 = q2.z;
	const float w1 = q1.w, x1 = q1.x, y1 = q1.y, z1 = q1.z;
	if (w1 * w2 + x1 * x2 + y1 * y2 + z1 * z2 < 0) { w2 = -w2; x2 = -x2; y2 = -y2; z2 = -z2; }
	const float theta = std::acos(w1 * w2 + x1 * x2 + y1 * y2 + z1 * z2);
	float mult1, mult2;
	if (theta > 0.000001) { mult1 = std::sin((1 - t) * theta) / std::sin(theta); mult2 = std::sin(t * theta) / std::sin(theta); }
	else { mult1 = 1 - t; mult2 = t; }
	Quat r;
	r.w = mult1 * w1 + mult2 * w2; r.x = mult1 * x1 + mult2 * x2; r.y = mult1 * y1 + mult2 * y2; r.z = mult1 * z1 + mult2 * z2;
	return r;
}
void from_quaternion(Quat q, float* v) {                        // Matrix::fromQuaternion
	const float w = q.w, x = q.x, y = q.y, z = q.z;
	v[0] = w * w + x * x - y * y - z * z;
	v[1] = (float)(2.0 * x * y + 2.0 * w * z);
	v[2] = (float)(2.0 * x * z - 2.0 * y * w);
	v[3] = (float)(2.0 * x * y - 2.0 * w * z);
	v[4] = w * w - x * x + y * y - z * z;
	v[5] = (float)(2.0 * y * z + 2.0 * w * x);
	v[6] = (float)(2.0 * x * z + 2.0 * w * y);
	v[7] = (float)(2.0 * y * z - 2.0 * w * x);
	v[8] = w * w - x * x - y * y + z * z;
}
}  // namespace
void Object::get_rotation(float frame, float out9[9]) const {
	auto it1 = rotation_keyframes.upper_bound(frame);
	auto it2 = it1;
	if (it1 == rotation_keyframes.end()) { memcpy(out9, rotation_keyframes.size() != 0 ? rotation_keyframes.rbegin()->second.data() : mat_rotation, 36); return; }
	if (it1 == rotation_keyframes.begin()) { memcpy(out9, it1->second.data(), 36); return; }
	it1--;
	const float t = (frame - it1->first) / (it2->first - it1->first);
	from_quaternion(slerp(to_quaternion(it1->second.data()), to_quaternion(it2->second.data()), t), out9);
}
void Object::add_keyframe(int frame) {
	std::array<float, 9> r; memcpy(r.data(), mat_rotation, 36);
	rotation_keyframes[(float)frame] = r;
	translation_keyframes[(float)frame] = max_translation;
	scale_keyframes[(float)frame] = scale;
}

void Object::build_matrix(float frame) {   // Geometry.h:322-360
	float m[9];
	get_rotation(frame, m);
	float mt[9];
	for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mt[j * 3 + i] = m[i * 3 + j];
	const float s = get_scale(frame);
	const Vector tr = get_translation(frame);
	for (int i = 0; i < 3; i++) {
		Vector v2(m[0 * 3 + i], m[1 * 3 + i], m[2 * 3 + i]);
		trans_matrix[0 * 4 + i] = v2[0] * s; trans_matrix[1 * 4 + i] = v2[1] * s; trans_matrix[2 * 4 + i] = v2[2] * s;
		rot_matrix[0 * 3 + i] = v2[0]; rot_matrix[1 * 3 + i] = v2[1]; rot_matrix[2 * 3 + i] = v2[2];
		v2 = Vector(mt[0 * 3 + i], mt[1 * 3 + i], mt[2 * 3 + i]);
		inv_trans_matrix[0 * 4 + i] = v2[0] / s; inv_trans_matrix[1 * 4 + i] = v2[1] / s; inv_trans_matrix[2 * 4 + i] = v2[2] / s;
	}
	auto matvec = [](const float* M, const Vector& b) {   // Matrix<3,3>*Vector (Vector.h:438-450)
		Vector r;
		for (int i = 0; i < 3; i++) { float v = 0; for (int j = 0; j < 3; j++) v += M[i * 3 + j] * b[j]; r[i] = v; }
		return r;
	};
	Vector nrc(-rotation_center[0], -rotation_center[1], -rotation_center[2]);
	Vector v2 = matvec(m, nrc);
	for (int k = 0; k < 3; k++) trans_matrix[k * 4 + 3] = v2[k] * s + rotation_center[k] + tr[k];
	v2 = matvec(mt, sub(nrc, tr));
	for (int k = 0; k < 3; k++) inv_trans_matrix[k * 4 + 3] = v2[k] / s + rotation_center[k];
}

Vector Object::apply_transformation(const Vector& v) const {   // Geometry.h:362-368
	const float* t = trans_matrix;
	return Vector(t[0] * v[0] + t[1] * v[1] + t[2] * v[2] + t[3], t[4] * v[0] + t[5] * v[1] + t[6] * v[2] + t[7], t[8] * v[0] + t[9] * v[1] + t[10] * v[2] + t[11]);
}

Sphere::Sphere(const Vector& origin, float rayon) {   // Sphere::init (Geometry.h:856-873)
	type = OT_SPHERE; O = origin; R = rayon; rotation_center = origin; name = "Sphere";
}
void Sphere::load_envmap_rgb8(const unsigned char* rgb, int w, int h) {
	g_content_epoch++;
	envtex.resize((size_t)w * h * 3);
	for (int i = 0; i < h; i++) memcpy(&envtex[(size_t)i * w * 3], rgb + (size_t)(h - 1 - i) * w * 3, (size_t)w * 3);
	envW = w; envH = h; has_envmap = true; flip_normals = true;
}
Plane::Plane(const Vector& A_, const Vector& N) { type = OT_PLANE; A = A_; vecN = N; name = "Plane"; }

// ---------------------------------------------------------------- TriMesh
// contiguous chunks of [0, n) on the host's hardware threads (mesh-sized loops whose iterations are independent)
template <class F>
static void parallel_for(int n, F f) {
	// (at most MIPT_HOST_THREADS, default 32: these loops are bound by memory, and starting 256 threads — the GPU box's hardware_concurrency —
	// costs a few milliseconds per loop, six loops per TriMesh::init)
	static const int cap = [] { const char* e = getenv("MIPT_HOST_THREADS"); const int v = e ? atoi(e) : 32; return v > 0 ? v : 32; }();
	const int nt = std::max(1, std::min(std::min((int)std::thread::hardware_concurrency(), cap), n / 65536));
	if (nt <= 1) { f(0, n); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < nt; t++) th.emplace_back([=] { f((int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt)); });
	for (auto& x : th) x.join();
}

TriMesh::TriMesh(int nv, const float* verts, int nn, const float* norms, int nt, const float* uv,
                 int nf, const int* fv, const int* fn, const int* ft, bool center) {
	type = OT_TRIMESH; interp_normals = true; name = "mesh";
	const auto t_ctor = std::chrono::steady_clock::now();
	vertices.resize(nv); normals.resize(nn); uvs.resize(nt);
	parallel_for(nv, [&](int a, int b) { for (int i = a; i < b; i++) vertices[i] = Vector(verts[3 * (size_t)i], verts[3 * (size_t)i + 1], verts[3 * (size_t)i + 2]); });
	parallel_for(nn, [&](int a, int b) { for (int i = a; i < b; i++) normals[i] = Vector(norms[3 * (size_t)i], norms[3 * (size_t)i + 1], norms[3 * (size_t)i + 2]); });
	parallel_for(nt, [&](int a, int b) { for (int i = a; i < b; i++) uvs[i] = Vector(uv[2 * (size_t)i], uv[2 * (size_t)i + 1], 0); });
	indices.resize(nf);
	parallel_for(nf, [&](int a, int b) {
	for (int i = a; i < b; i++) {
		const size_t j = 3 * (size_t)i;
		mipt_triangle_indices& t = indices[i];
		memset(&t, 0, sizeof t);
		t.vtxi = fv[j]; t.vtxj = fv[j + 1]; t.vtxk = fv[j + 2];
		t.ni = fn ? fn[j] : -1; t.nj = fn ? fn[j + 1] : -1; t.nk = fn ? fn[j + 2] : -1;
		t.uvi = ft ? ft[j] : -1; t.uvj = ft ? ft[j + 1] : -1; t.uvk = ft ? ft[j + 2] : -1;
		t.group = 0;   // no usemtl: every face in group 0, "Default" (TriangleMesh.cpp:462-467)
		t.showEdges[0] = t.showEdges[1] = t.showEdges[2] = 1;
	}
	});
	groupNames["Default"] = 0;
	add_default_group_materials(1);
	if (getenv("MIPT_BUILD_TRACE")) fprintf(stderr, "[TriMesh::TriMesh] %-26s %8.2f ms\n", "arrays -> members", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ctor).count());
	finish_init(center);
}

// which builder TriMesh::init uses: 0 = host recursion, 1 = GPU (an error if there is no device), 2 = GPU when a device
// is present, host recursion otherwise (default)
static int g_bvh_builder_mode = 2, g_bvh_builder_device = 0;
static int g_device_resident = 1;     // builder on the GPU: leave the tree and the records there (mipt_device_mesh_build) instead of fetching them back
extern "C" void mh_set_device_resident(int on) { g_device_resident = on; }
extern "C" void mh_set_obj_slicing(int slice_bytes, int max_slices);   // test hook: how readOBJ cuts the text into concurrently parsed slices
extern "C" void mh_set_bvh_builder(int mode, int device) { g_bvh_builder_mode = mode; g_bvh_builder_device = device; }

// readOBJ's per-group default material lists (TriangleMesh.cpp:470-480)
void TriMesh::add_default_group_materials(int ngroups) {
	for (int g = 0; g < ngroups; g++) {
		add_col_texture(Vector(0.5f, 0.5f, 0.5f)); add_col_specular(Vector(0, 0, 0)); add_col_roughness(Vector(0, 0, 0));
		add_null_normalmap(); add_col_alpha(1.f); add_col_refr(1.3f); add_col_transp(1.f); add_col_subsurface(Vector(0, 0, 0));
	}
}

// TriMesh::init after the file has been read (TriangleMesh.cpp:742-841), scaling = 1, offset = 0, preserve_input = false
void TriMesh::finish_init(bool center) {
	g_content_epoch++;
	const bool trace = getenv("MIPT_BUILD_TRACE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto phase = [&](const char* what) {
		if (!trace) return;
		const auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[TriMesh::init] %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
		t_last = t;
	};
	const int nn = (int)normals.size(), nt = (int)uvs.size(), nf = (int)indices.size();
	// axis swap (x,y,z) -> (-z,y,x) (TriangleMesh.cpp:742-751)
	const int nvtx = (int)vertices.size();
	parallel_for(nn, [&](int a, int b) { for (int i = a; i < b; i++) { Vector& v = normals[i]; std::swap(v[0], v[2]); v[0] = -v[0]; } });
	float bmin[3] = {1E9f, 1E9f, 1E9f}, bmax[3] = {-1E9f, -1E9f, -1E9f};
	{   // min / max over chunks, joined in chunk order: std::min / std::max keep the earlier of equal values either way
		// (the swap of the vertices rides in the same pass: one trip through 142 MB instead of two at 23.7 M triangles)
		std::mutex mu;
		std::map<int, std::array<float, 6>> part;
		parallel_for(nvtx, [&](int a, int b) {
			std::array<float, 6> q = {1E9f, 1E9f, 1E9f, -1E9f, -1E9f, -1E9f};
			for (int i = a; i < b; i++) { Vector& v = vertices[i]; std::swap(v[0], v[2]); v[0] = -v[0]; for (int k = 0; k < 3; k++) { q[k] = std::min(q[k], v[k]); q[3 + k] = std::max(q[3 + k], v[k]); } }
			std::lock_guard<std::mutex> g(mu);
			part[a] = q;
		});
		for (const auto& kv : part) for (int k = 0; k < 3; k++) { bmin[k] = std::min(bmin[k], kv.second[k]); bmax[k] = std::max(bmax[k], kv.second[3 + k]); }
	}
	if (center) {   // :760-770 with scaling = 1, offset = 0
		float s = std::max(bmax[0] - bmin[0], std::max(bmax[1] - bmin[1], bmax[2] - bmin[2]));
		float c[3] = {(bmin[0] + bmax[0]) * 0.5f, (bmin[1] + bmax[1]) * 0.5f, (bmin[2] + bmax[2]) * 0.5f};
		parallel_for(nvtx, [&](int a, int b) { for (int i = a; i < b; i++) for (int k = 0; k < 3; k++) vertices[i][k] = (vertices[i][k] - c[k]) / s * 1.f + 0.f; });
	}
	phase("axis swap, bounds, centring");
	// build_bvh (:878-885): on the GPU (mipt_build_bvh, same tree and triangle order) or with the host recursion below
	build_bbox(0, nf, bvh.bbox);
	const auto t_build = std::chrono::steady_clock::now();
	bvh_builder = 0;
	if (g_bvh_builder_mode != 0 && g_device_resident && build_device_resident()) {
		// the tree, the reordered Triangle records and the tangents are on the device and stay there; the host arrays that mirror
		// them are filled by sync_host() / sync_tangents() when something reads them
		bvh_build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
		phase("mipt_device_mesh_build");
		memcpy(bbox, bvh.bbox, sizeof bbox);           // build_bbox(0, nf, bbox) (:804): the box of all triangles again, whatever their order
		rotation_center = Vector((bbox[0] + bbox[3]) * 0.5f, (bbox[1] + bbox[4]) * 0.5f, (bbox[2] + bbox[5]) * 0.5f);   // :831-835
		phase("bounds");
		return;
	}
	permuted_triangle_index.resize(nf);       // (with the tree on the device this happens in sync_host(), when somebody asks for the permutation)
	parallel_for(nf, [&](int a, int b) { for (int i = a; i < b; i++) permuted_triangle_index[i] = i; });
	if (g_bvh_builder_mode != 0 && !build_bvh_gpu()) {
		if (g_bvh_builder_mode == 1 || !bvh_gpu_unavailable) { loaded = false; return; }   // forced, or bad input: say so
	}
	if (!bvh_builder) {
		bvh.nodes.clear();
		bvh.nodes.reserve((size_t)nf * 2);
		build_bvh_recur(bvh.nodes, 0, nf, 0);
	}
	bvh_build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
	phase("build_bvh");
	build_bbox(0, nf, bbox);
	build_triangle_soup();
	rotation_center = Vector((bbox[0] + bbox[3]) * 0.5f, (bbox[1] + bbox[4]) * 0.5f, (bbox[2] + bbox[5]) * 0.5f);   // :831-835
	phase("triangle soup");
	if (nt != 0) setup_tangents();
	phase("tangents");
}

// triangle soup, after the reorder (:812-829; Triangle ctor TriangleMesh.h:70-78)
void TriMesh::build_triangle_soup() {
	const int nn = (int)normals.size(), nt = (int)uvs.size(), nf = (int)indices.size();
	triangleSoup.resize(nf);
	parallel_for(nf, [&](int i0, int i1) {
	for (int i = i0; i < i1; i++) {
		mipt_triangle& T = triangleSoup[i];
		memset(&T, 0, sizeof T);
		const Vector &A = vertices[indices[i].vtxi], &B = vertices[indices[i].vtxj], &C = vertices[indices[i].vtxk];
		Vector u = sub(B, A), v = sub(C, A), N = cross(u, v);
		for (int k = 0; k < 3; k++) { T.A[k] = A[k]; T.u[k] = u[k]; T.v[k] = v[k]; T.N[k] = N[k]; }
		T.m11 = norm2(u); T.m22 = norm2(v); T.m12 = dot(u, v);
		T.invdetm = 1.f / (T.m11 * T.m22 - T.m12 * T.m12);   // 1./x narrowed == 1.f/x
		if (nn != 0) {
			const int nidx[3] = {indices[i].ni, indices[i].nj, indices[i].nk};
			for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) T.normals[k][l] = normals[nidx[k]][l];
		}
		if (nt != 0) {
			const int tidx[3] = {indices[i].uvi, indices[i].uvj, indices[i].uvk};
			for (int k = 0; k < 3; k++) { T.uvs[k][0] = uvs[tidx[k]][0]; T.uvs[k][1] = uvs[tidx[k]][1]; }
		}
	}
	});
}

TriMesh::~TriMesh() { if (device_mesh) mipt_device_mesh_free(device_mesh); }

// mipt_device_mesh_build (include/mipt.h): build_bvh, the Triangle records and setup_tangents on the device, nothing fetched back.
bool TriMesh::build_device_resident() {
	const int nf = (int)indices.size();
	mipt_device_mesh_info info;
	memset(&info, 0, sizeof info);
	mipt_device_mesh* dm = nullptr;
	const int rc = mipt_device_mesh_build(g_bvh_builder_device, &vertices[0][0], (int)vertices.size(), normals.empty() ? nullptr : &normals[0][0], (int)normals.size(),
	                                      uvs.empty() ? nullptr : &uvs[0][0], (int)uvs.size(), indices.data(), nf, &dm, &info);
	if (rc != MIPT_OK) {
		// no device / out of memory: the next builder in line takes over; a tree the traversal cannot hold (too deep, fat leaves) is
		// reported by the upload of the host-built arrays exactly as before
		load_error = std::string("mipt_device_mesh_build: ") + mipt_build_bvh_error();
		return false;
	}
	device_mesh = dm; device_nodes = info.n_nodes;
	host_views_current = false;
	tangents_current = uvs.empty();      // (a mesh without UVs has no tangentSoup)
	bvh_builder = 2;
	bvh_device_seconds = info.build_seconds + info.records_seconds;
	bvh.nodes.clear(); triangleSoup.clear(); tangentSoup.clear();
	load_error.clear();
	return true;
}

void TriMesh::sync_host() {
	if (host_views_current) return;
	const int nf = (int)indices.size();
	bvh.nodes.resize((size_t)device_nodes);
	std::vector<int32_t> perm(nf);
	if (mipt_device_mesh_download(device_mesh, reinterpret_cast<mipt_bvh_node*>(bvh.nodes.data()), device_nodes, perm.data()) != MIPT_OK) {
		load_error = std::string("mipt_device_mesh_download: ") + mipt_build_bvh_error();
		bvh.nodes.clear();
		return;
	}
	PodVec<mipt_triangle_indices> dst(nf);
	permuted_triangle_index.resize(nf);
	parallel_for(nf, [&](int i0, int i1) { for (int i = i0; i < i1; i++) { dst[i] = indices[perm[i]]; permuted_triangle_index[i] = perm[i]; } });
	indices.swap(dst);
	build_triangle_soup();
	host_views_current = true;
}

void TriMesh::sync_tangents() {
	sync_host();
	if (tangents_current) return;
	const int nf = (int)indices.size();
	tangentSoup.resize((size_t)nf * 3);
	if (mipt_device_mesh_download_tangents(device_mesh, &tangentSoup[0][0]) != MIPT_OK) setup_tangents();      // (a mesh with UVs but no normals: the host loop)
	tangents_current = true;
}

// ---------------------------------------------------------------- OBJ / MTL ingestion
// Same file semantics as TriMesh::readOBJ (TriangleMesh.cpp:240-569), written as a small scanner instead of sscanf
// cascades.  What is reproduced, because it decides which triangles / materials come out:
//  * physical lines are cut at 254 characters (fgets(line, 255)); only the first two characters select the record;
//  * "v x y z [r g b]" (colours ignored here), "vn", "vt u v", "usemtl name" (groups numbered by first appearance),
//    "mtllib file";
//  * a face corner is v/t/n, v/t, v or v//n with 1-based or negative (relative) indices — the first three corners must
//    share one form (tried in that order), further corners are matched one by one (v/t/n, v/t, v//n, v) — and
//    polygons are fanned as (c0, c_{k-1}, c_k); the text is split into slices that are parsed concurrently;
//  * no usemtl anywhere: every face in group 0 ("Default");
//  * MTL: records are recognised at column 0 only ("Kd", "Ks", "Ns", "map_Kd", "map_Ks", "map_Bump", "map_d",
//    "newmtl"), "Ns" with one value is replicated, a material name the OBJ never used lands in group 0
//    (std::map::operator[] default), illum is never evaluated.
// Images: binary PPM (P6, maxval 255) — the one format of the reference's decoder (stb_image) that needs no codec;
// anything else is reported through load_error and leaves the constant colour in place.
namespace {
std::string dir_of(const std::string& path) { size_t k = path.find_last_of("/\\"); return path.substr(0, k + 1); }   // extractFilePathWithEndingSlash
std::string rest_of_line(const char* line, size_t skip) {   // sscanf(line, "keyword %[^\n]"): blanks after the keyword are skipped
	const char* p = line + std::min(skip - 1, strlen(line));
	while (*p == ' ' || *p == '\t') p++;
	std::string r(p);
	size_t k = r.find('\n');
	if (k != std::string::npos) r.erase(k);
	return r;
}
bool read_ppm(const std::string& file, std::vector<unsigned char>& rgb, int& W, int& H) {
	FILE* f = fopen(file.c_str(), "rb");
	if (!f) return false;
	auto token = [&](int& out) {
		int ch = fgetc(f);
		for (;;) {
			while (ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t') ch = fgetc(f);
			if (ch == '#') { while (ch != '\n' && ch != EOF) ch = fgetc(f); continue; }
			break;
		}
		if (ch < '0' || ch > '9') return false;
		out = 0;
		while (ch >= '0' && ch <= '9') { out = out * 10 + (ch - '0'); ch = fgetc(f); }
		return true;                                 // the single whitespace after the token is consumed
	};
	bool ok = fgetc(f) == 'P' && fgetc(f) == '6';
	int maxv = 0;
	ok = ok && token(W) && token(H) && token(maxv) && maxv == 255 && W > 0 && H > 0;
	if (ok) { rgb.resize((size_t)W * H * 3); ok = fread(rgb.data(), 1, rgb.size(), f) == rgb.size(); }
	fclose(f);
	return ok;
}

// PNG as the reference's decoder (stb_image, asked for 3 channels) delivers it: non-interlaced, bit depth 1 / 2 / 4 / 8;
// grey levels below 8 bits are scaled to 0..255, grey is replicated to RGB, a palette is expanded, alpha (channel or tRNS)
// is dropped.  Decompression by zlib's inflate, the five scanline filters undone here.  16-bit and Adam7 files are refused.
bool read_png(const std::string& file, std::vector<unsigned char>& rgb, int& W, int& H, std::string& why) {
	FILE* f = fopen(file.c_str(), "rb");
	if (!f) { why = "cannot open"; return false; }
	std::vector<unsigned char> buf;
	{ unsigned char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n); }
	fclose(f);
	static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	if (buf.size() < 33 || memcmp(buf.data(), sig, 8) != 0) { why = "not a PNG file"; return false; }
	auto be32 = [&](size_t o) { return ((unsigned)buf[o] << 24) | ((unsigned)buf[o + 1] << 16) | ((unsigned)buf[o + 2] << 8) | (unsigned)buf[o + 3]; };
	int depth = 0, ctype = 0, interlace = 0;
	std::vector<unsigned char> idat, plte;
	size_t o = 8;
	bool seen_end = false;
	while (o + 12 <= buf.size() && !seen_end) {
		const unsigned len = be32(o);
		const char* tag = (const char*)&buf[o + 4];
		if (o + 12 + (size_t)len > buf.size()) { why = "truncated chunk"; return false; }
		const unsigned char* d = &buf[o + 8];
		if (!memcmp(tag, "IHDR", 4)) {
			if (len != 13) { why = "bad IHDR"; return false; }
			W = (int)be32(o + 8); H = (int)be32(o + 12); depth = d[8]; ctype = d[9]; interlace = d[12];
			if (d[10] != 0 || d[11] != 0) { why = "unknown compression / filter method"; return false; }
		} else if (!memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
		else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
		else if (!memcmp(tag, "IEND", 4)) seen_end = true;
		o += 12 + (size_t)len;
	}
	if (W <= 0 || H <= 0 || idat.empty()) { why = "no image data"; return false; }
	if (interlace != 0) { why = "Adam7-interlaced PNG is not decoded here"; return false; }
	if (depth == 16) { why = "16-bit PNG (the reference reads it through CImg) is not decoded here"; return false; }
	int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
	if (!chans || !(depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) { why = "unsupported colour type / bit depth"; return false; }
	if (ctype == 3 && plte.size() < 3) { why = "palette image without PLTE"; return false; }
	const size_t row_bytes = ((size_t)W * chans * depth + 7) / 8, bpp = std::max<size_t>(1, (size_t)chans * depth / 8);
	std::vector<unsigned char> raw((row_bytes + 1) * (size_t)H);
	{
		z_stream zs; memset(&zs, 0, sizeof zs);
		if (inflateInit(&zs) != Z_OK) { why = "zlib init failed"; return false; }
		zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size(); zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
		const int rc = inflate(&zs, Z_FINISH);
		const bool complete = zs.total_out == raw.size();
		inflateEnd(&zs);
		if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || !complete) { why = "corrupt image data"; return false; }
	}
	std::vector<unsigned char> prev(row_bytes, 0), cur(row_bytes);
	rgb.assign((size_t)W * H * 3, 0);
	for (int y = 0; y < H; y++) {
		const unsigned char* in = &raw[(row_bytes + 1) * (size_t)y];
		const int ft = in[0];
		if (ft > 4) { why = "bad scanline filter"; return false; }
		for (size_t x = 0; x < row_bytes; x++) {
			const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
			int v = in[1 + x];
			if (ft == 1) v += a;
			else if (ft == 2) v += b;
			else if (ft == 3) v += (a + b) >> 1;
			else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
			cur[x] = (unsigned char)v;
		}
		unsigned char* out = &rgb[(size_t)y * W * 3];
		static const int scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};      // grey levels to 0..255 (not applied to palette indices)
		for (int x = 0; x < W; x++) {
			auto sample = [&](int k) -> int {                  // k-th sample of the row
				if (depth == 8) return cur[k];
				const int per = 8 / depth, byte = cur[k / per], shift = 8 - depth * (k % per + 1);
				return (byte >> shift) & ((1 << depth) - 1);
			};
			if (ctype == 0 || ctype == 4) { const int g = sample(x * chans) * (ctype == 0 ? scale[depth] : 1); out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = (unsigned char)g; }
			else if (ctype == 3) {
				const size_t idx = (size_t)sample(x);
				if (3 * idx + 2 >= plte.size()) { why = "palette index out of range"; return false; }
				out[3 * x] = plte[3 * idx]; out[3 * x + 1] = plte[3 * idx + 1]; out[3 * x + 2] = plte[3 * idx + 2];
			} else { out[3 * x] = cur[(size_t)x * chans]; out[3 * x + 1] = cur[(size_t)x * chans + 1]; out[3 * x + 2] = cur[(size_t)x * chans + 2]; }
		}
		prev.swap(cur);
	}
	return true;
}

// Windows BMP, the uncompressed kinds (BI_RGB): 24 / 32 bits per pixel and 8-bit palettes, bottom-up or top-down rows;
// delivered top row first like stb_image does.  Compressed, bit-field and 16-bit files are refused.
bool read_bmp(const std::string& file, std::vector<unsigned char>& rgb, int& W, int& H, std::string& why) {
	FILE* f = fopen(file.c_str(), "rb");
	if (!f) { why = "cannot open"; return false; }
	std::vector<unsigned char> b;
	{ unsigned char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n); }
	fclose(f);
	auto le32 = [&](size_t o) { return (unsigned)b[o] | ((unsigned)b[o + 1] << 8) | ((unsigned)b[o + 2] << 16) | ((unsigned)b[o + 3] << 24); };
	auto le16 = [&](size_t o) { return (unsigned)b[o] | ((unsigned)b[o + 1] << 8); };
	if (b.size() < 54 || b[0] != 'B' || b[1] != 'M') { why = "not a BMP file"; return false; }
	const unsigned offset = le32(10), hsz = le32(14);
	if (hsz < 40) { why = "OS/2 BMP headers are not decoded here"; return false; }
	W = (int)le32(18);
	int h = (int)le32(22);
	const bool flip = h > 0;                                   // positive height: rows stored bottom-up
	H = h < 0 ? -h : h;
	const unsigned bpp = le16(28), comp = le32(30);
	if (comp != 0 || !(bpp == 24 || bpp == 32 || bpp == 8)) { why = "only uncompressed 8 / 24 / 32-bit BMP is decoded here"; return false; }
	if (W <= 0 || H <= 0) { why = "bad BMP size"; return false; }
	const size_t stride = (((size_t)W * bpp + 31) / 32) * 4;
	if ((size_t)offset + stride * H > b.size()) { why = "truncated BMP"; return false; }
	const unsigned char* pal = &b[14 + hsz];
	unsigned ncol = le32(46); if (bpp == 8 && ncol == 0) ncol = 256;
	if (bpp == 8 && 14 + (size_t)hsz + 4 * (size_t)ncol > b.size()) { why = "truncated BMP palette"; return false; }
	rgb.resize((size_t)W * H * 3);
	for (int y = 0; y < H; y++) {
		const unsigned char* row = &b[offset + stride * (size_t)(flip ? H - 1 - y : y)];
		unsigned char* out = &rgb[(size_t)y * W * 3];
		for (int x = 0; x < W; x++) {
			if (bpp == 8) { const unsigned i = row[x]; if (i >= ncol) { why = "BMP palette index out of range"; return false; } out[3 * x] = pal[4 * i + 2]; out[3 * x + 1] = pal[4 * i + 1]; out[3 * x + 2] = pal[4 * i]; }
			else { const unsigned char* px = row + (size_t)x * (bpp / 8); out[3 * x] = px[2]; out[3 * x + 1] = px[1]; out[3 * x + 2] = px[0]; }
		}
	}
	return true;
}

// an 8-bit RGB image by content: binary PPM, PNG or uncompressed BMP (what the reference reads through stb_image without a lossy codec)
bool read_image_rgb8(const std::string& file, std::vector<unsigned char>& rgb, int& W, int& H, std::string& why) {
	FILE* f = fopen(file.c_str(), "rb");
	if (!f) { why = "cannot open"; return false; }
	unsigned char magic[4] = {0, 0, 0, 0};
	const size_t got = fread(magic, 1, 4, f);
	fclose(f);
	if (got >= 2 && magic[0] == 'P' && magic[1] == '6') { if (read_ppm(file, rgb, W, H)) return true; why = "malformed binary PPM (P6, maxval 255 expected)"; return false; }
	if (got == 4 && magic[0] == 0x89 && magic[1] == 'P' && magic[2] == 'N' && magic[3] == 'G') return read_png(file, rgb, W, H, why);
	if (got >= 2 && magic[0] == 'B' && magic[1] == 'M') return read_bmp(file, rgb, W, H, why);
	if (got >= 2 && magic[0] == 0xff && magic[1] == 0xd8) {   // JPEG (baseline / progressive Huffman): mipt_jpeg.h
		std::vector<unsigned char> buf;
		FILE* g = fopen(file.c_str(), "rb");
		if (!g) { why = "cannot open"; return false; }
		{ unsigned char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, g)) > 0) buf.insert(buf.end(), tmp, tmp + n); }
		fclose(g);
		return mipt_jpeg::decode(buf.data(), buf.size(), rgb, W, H, why);
	}
	why = "only JPEG, PNG, binary PPM and uncompressed BMP images are decoded here (TGA / HDR / GIF need the reference's codecs)";
	return false;
}

// save_image for 8-bit RGB (utils.cpp:178-234): the container is chosen by the file name's extension, found anywhere in
// the lower-cased name like the reference's `ls.find(".png")`, in its order of tests (.hdr, .bmp, .tga, .jpg, .png).
// PNG (8-bit RGB, zlib deflate, per-row choice among the five scanline filters), 24-bit BMP, uncompressed TGA, baseline JPEG
// at quality 100 (mipt_imgwrite.h: the bytes of the reference's encoder) and binary PPM are written here.  `.hdr` takes float
// pixels (write_image_f32 below; the reference's 8-bit instantiation reinterprets its bytes as floats and reads out of
// bounds, utils.cpp:184-190); what the reference writes through CImg (anything else) is refused loudly instead of being
// written in another format under that name.
enum ImageFormat { F_NONE, F_HDR, F_BMP, F_TGA, F_JPG, F_PNG, F_PPM };
static ImageFormat image_format_of(const std::string& file) {
	std::string ls(file);
	for (char& ch : ls) ch = (char)tolower((unsigned char)ch);
	auto has = [&](const char* ext) { return ls.find(ext) != std::string::npos; };
	return has(".hdr") ? F_HDR : has(".bmp") ? F_BMP : has(".tga") ? F_TGA : has(".jpg") ? F_JPG : has(".png") ? F_PNG : has(".ppm") ? F_PPM : F_NONE;
}
static bool write_file(const std::string& file, const std::vector<unsigned char>& out, std::string& why) {
	FILE* f = fopen(file.c_str(), "wb");
	if (!f) { why = "cannot create"; return false; }
	const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
	if (fclose(f) != 0 || !ok) { why = "write failed"; return false; }
	return true;
}
bool write_image_rgb8(const std::string& file, const unsigned char* rgb, int W, int H, std::string& why) {
	if (W <= 0 || H <= 0 || !rgb) { why = "empty image"; return false; }
	const ImageFormat fmt = image_format_of(file);
	if (fmt == F_HDR) { why = "Radiance .hdr stores float pixels: hand the float image to save_image (mh_save_image_f32)"; return false; }
	if (fmt == F_NONE) { why = "unknown image extension (written here: .png, .bmp, .tga, .jpg, .ppm, and .hdr from float pixels)"; return false; }
	if (fmt == F_JPG) {
		if (W > 65535 || H > 65535) { why = "JPEG dimensions are 16-bit"; return false; }
		std::vector<unsigned char> jpg;
		mipt_imgwrite::encode_jpeg_q100(rgb, W, H, jpg);
		return write_file(file, jpg, why);
	}
	std::vector<unsigned char> out;
	auto put = [&](const void* p, size_t n) { out.insert(out.end(), (const unsigned char*)p, (const unsigned char*)p + n); };
	auto le16 = [&](unsigned v) { unsigned char b[2] = {(unsigned char)v, (unsigned char)(v >> 8)}; put(b, 2); };
	auto le32 = [&](unsigned v) { unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; put(b, 4); };
	const size_t row = (size_t)W * 3;
	if (fmt == F_PPM) {
		char hdr[64]; const int n = snprintf(hdr, sizeof hdr, "P6\n%d %d\n255\n", W, H);
		put(hdr, (size_t)n); put(rgb, row * H);
	} else if (fmt == F_BMP) {                           // BITMAPINFOHEADER, 24 bits, rows bottom-up, BGR, padded to 4 bytes
		const size_t pad = (4 - row % 4) % 4, img = (row + pad) * H;
		put("BM", 2); le32((unsigned)(54 + img)); le32(0); le32(54);
		le32(40); le32((unsigned)W); le32((unsigned)H); le16(1); le16(24); le32(0); le32((unsigned)img); le32(0); le32(0); le32(0); le32(0);
		const unsigned char zero[3] = {0, 0, 0};
		for (int i = H - 1; i >= 0; i--) {
			for (int j = 0; j < W; j++) { const unsigned char* q = rgb + (size_t)i * row + 3 * j; const unsigned char bgr[3] = {q[2], q[1], q[0]}; put(bgr, 3); }
			put(zero, pad);
		}
	} else if (fmt == F_TGA) {                           // type 2 (uncompressed true colour), top-left origin, BGR
		const unsigned char hdr[18] = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (unsigned char)W, (unsigned char)(W >> 8), (unsigned char)H, (unsigned char)(H >> 8), 24, 0x20};
		put(hdr, 18);
		for (size_t k = 0; k < (size_t)W * H; k++) { const unsigned char bgr[3] = {rgb[3 * k + 2], rgb[3 * k + 1], rgb[3 * k]}; put(bgr, 3); }
	} else {
		// scanlines with the filter (None / Sub / Up / Average / Paeth) whose output has the smallest sum of |signed byte|
		std::vector<unsigned char> raw((row + 1) * H), cand(row);
		std::vector<unsigned char> zero_row(row, 0);
		for (int i = 0; i < H; i++) {
			const unsigned char* cur = rgb + (size_t)i * row;
			const unsigned char* up = i ? cur - row : zero_row.data();
			int best_f = 0; unsigned long best_sum = ~0ul;
			for (int f = 0; f < 5; f++) {
				unsigned long sum = 0;
				for (size_t x = 0; x < row; x++) {
					const int a = x >= 3 ? cur[x - 3] : 0, b = up[x], c = x >= 3 ? up[x - 3] : 0;
					int pred = 0;
					if (f == 1) pred = a; else if (f == 2) pred = b; else if (f == 3) pred = (a + b) >> 1;
					else if (f == 4) { const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
					const unsigned char v = (unsigned char)(cur[x] - pred);
					cand[x] = v; sum += (unsigned long)abs((int)(signed char)v);
				}
				if (sum < best_sum) { best_sum = sum; best_f = f; raw[(size_t)i * (row + 1)] = (unsigned char)f; memcpy(&raw[(size_t)i * (row + 1) + 1], cand.data(), row); }
			}
			(void)best_f;
		}
		uLongf zlen = compressBound((uLong)raw.size());
		std::vector<unsigned char> z(zlen);
		if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { why = "zlib compress failed"; return false; }
		auto chunk = [&](const char* tag, const unsigned char* data, size_t n) {
			const unsigned char len[4] = {(unsigned char)(n >> 24), (unsigned char)(n >> 16), (unsigned char)(n >> 8), (unsigned char)n};
			put(len, 4);
			const size_t at = out.size();
			put(tag, 4); if (n) put(data, n);
			const unsigned long crc = crc32(0L, out.data() + at, (uInt)(n + 4));
			const unsigned char cb[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
			put(cb, 4);
		};
		const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
		put(sig, 8);
		const unsigned char ihdr[13] = {(unsigned char)(W >> 24), (unsigned char)(W >> 16), (unsigned char)(W >> 8), (unsigned char)W,
		                                (unsigned char)(H >> 24), (unsigned char)(H >> 16), (unsigned char)(H >> 8), (unsigned char)H, 8, 2, 0, 0, 0};
		chunk("IHDR", ihdr, 13);
		chunk("IDAT", z.data(), (size_t)zlen);
		chunk("IEND", nullptr, 0);
	}
	FILE* f = fopen(file.c_str(), "wb");
	if (!f) { why = "cannot write " + file; return false; }
	const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
	if (fclose(f) != 0 || !ok) { why = "short write to " + file; return false; }
	return true;
}
}  // namespace

namespace {
// What one slice of the OBJ text contributes.  Slices start after a newline, so the 254-character record cutting of
// fgets(line, 255) falls the same way as in a front-to-back read; everything that depends on what came before the
// slice (relative indices, the current usemtl group) is recorded symbolically and resolved when the slices are joined.
// sscanf restricted to what the face records need: in `fmt`, 'u' is %u (leading white space skipped, optional sign,
// decimal digits), 'n' is %n, a blank matches any amount of white space, anything else must match literally.
// Returns the number of %u conversions (-1 when the input ends before the first one).
int scan_units(const char* s, const char* fmt, int* out, int* consumed) {
	const char* p = s;
	int n = 0;
	auto space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
	for (; *fmt; fmt++) {
		if (*fmt == 'u') {
			while (space(*p)) p++;
			if (*p == 0) return n ? n : -1;
			const char* d = p;
			if (*d == '+' || *d == '-') d++;
			if (*d < '0' || *d > '9') return n;
			char* e;
			out[n++] = (int)(unsigned)strtoul(p, &e, 10);
			p = e;
		} else if (*fmt == 'n') *consumed = (int)(p - s);
		else if (*fmt == ' ') { while (space(*p)) p++; }
		else { if (*p != *fmt) return n; p++; }
	}
	return n;
}

int g_obj_slice_bytes = 1 << 20, g_obj_max_slices = 0;
struct ObjSlice {
	std::vector<Vector> v, vn, vt;
	PodVec<mipt_triangle_indices> faces;       // indices local to the slice where the bit in `relmask` is set
	std::vector<uint16_t> relmask;             // bit k: field k of (vtxi,vtxj,vtxk,uvi,uvj,uvk,ni,nj,nk) is relative to the running count
	std::vector<int> face_group;               // index into `usemtl` of the group in force (-1: the one inherited from the previous slice)
	std::vector<std::string> usemtl;           // names in order of appearance
	std::string matfile; bool has_matfile = false;
	bool vertex_colours = false;
};

void parse_obj_slice(const char* p, const char* end, ObjSlice& o) {
	char line[255];
	while (p < end) {
		size_t len = 0;                            // fgets(line, 255, f)
		while (p < end && len < 254) { const char ch = *p++; line[len++] = ch; if (ch == '\n') break; }
		line[len] = 0;
		if (line[0] == 'u' && line[1] == 's') o.usemtl.push_back(rest_of_line(line, 7));
		if (line[0] == 'm' && line[1] == 't' && line[2] == 'l') { o.matfile = rest_of_line(line, 7); o.has_matfile = true; }
		if (line[0] == 'v' && line[1] == ' ') {
			float x = 0, y = 0, z = 0, cr, cg, cb;
			if (sscanf(line, "v %f %f %f %f %f %f", &x, &y, &z, &cr, &cg, &cb) == 6) { o.vertex_colours = true; return; }   // vertex colours feed getMaterial (TriangleMesh.cpp:980-1000): outside the hot path
			o.v.push_back(Vector(x, y, z));
		}
		if (line[0] == 'v' && line[1] == 'n') { float x = 0, y = 0, z = 0; sscanf(line, "vn %f %f %f", &x, &y, &z); o.vn.push_back(Vector(x, y, z)); }
		if (line[0] == 'v' && line[1] == 't') { float x = 0, y = 0; sscanf(line, "vt %f %f", &x, &y); o.vt.push_back(Vector(x, y, 0)); }
		if (line[0] == 'f') {
			// the sscanf cascades of TriangleMesh.cpp:327-457, form by form.  The first three corners must share one
			// form (v/t/n, v/t, v, v//n — tried in that order); every further corner is matched on its own, again in the
			// order v/t/n, v/t, v//n, v, so that e.g. the tail "44/" that the 254-character cut leaves of "44//1" still
			// yields the vertex 44, exactly as in the reference.
			const char* q = line + 1;
			int a[9], off = 0;
			bool form_t = false, form_n = false;
			int i0, i2, j0 = 0, j2 = 0, k0 = 0, k2 = 0, i1, j1 = 0, k1 = 0;
			if (scan_units(q, "u/u/u u/u/u u/u/un", a, &off) == 9) { form_t = form_n = true; i0 = a[0]; j0 = a[1]; k0 = a[2]; i1 = a[3]; j1 = a[4]; k1 = a[5]; i2 = a[6]; j2 = a[7]; k2 = a[8]; }
			else if (scan_units(q, "u/u u/u u/un", a, &off) == 6) { form_t = true; i0 = a[0]; j0 = a[1]; i1 = a[2]; j1 = a[3]; i2 = a[4]; j2 = a[5]; }
			else if (scan_units(q, "u u un", a, &off) == 3) { i0 = a[0]; i1 = a[1]; i2 = a[2]; }
			else if (scan_units(q, "u//u u//u u//un", a, &off) == 6) { form_n = true; i0 = a[0]; k0 = a[1]; i1 = a[2]; k1 = a[3]; i2 = a[4]; k2 = a[5]; }
			else continue;   // the reference goes on with whatever its variables hold (undefined); such lines are dropped here
			auto emit = [&](bool first, int vi, int vj, int vk, bool with_t, int ti, int tj, int tk, bool with_n, int ni, int nj, int nk, const char* after) {
				mipt_triangle_indices t;
				memset(&t, 0, sizeof t);
				t.showEdges[0] = first; t.showEdges[1] = 1;
				t.showEdges[2] = (after[0] == '\n') || (after[0] == ' ' && after[1] == '\n');
				unsigned mask = 0;
				auto rel = [&](int i, size_t n, int bit) { if (i < 0) { mask |= 1u << bit; return (int)n + i; } return i - 1; };
				t.vtxi = rel(vi, o.v.size(), 0); t.vtxj = rel(vj, o.v.size(), 1); t.vtxk = rel(vk, o.v.size(), 2);
				t.uvi = t.uvj = t.uvk = -1; t.ni = t.nj = t.nk = -1;
				if (with_t) { t.uvi = rel(ti, o.vt.size(), 3); t.uvj = rel(tj, o.vt.size(), 4); t.uvk = rel(tk, o.vt.size(), 5); }
				if (with_n) { t.ni = rel(ni, o.vn.size(), 6); t.nj = rel(nj, o.vn.size(), 7); t.nk = rel(nk, o.vn.size(), 8); }
				o.faces.push_back(t);
				o.relmask.push_back((uint16_t)mask);
				o.face_group.push_back((int)o.usemtl.size() - 1);
			};
			emit(true, i0, i1, i2, form_t, j0, j1, j2, form_n, k0, k1, k2, q + off);
			q += off;
			for (;;) {                                 // fan: (c0, previous, next)  (:391-457)
				if (*q == '\n' || *q == '\0') break;
				if (scan_units(q, "u/u/un", a, &off) == 3) { emit(false, i0, i2, a[0], true, j0, j2, a[1], true, k0, k2, a[2], q + off); q += off; i2 = a[0]; j2 = a[1]; k2 = a[2]; }
				else if (scan_units(q, "u/un", a, &off) == 2) { emit(false, i0, i2, a[0], true, j0, j2, a[1], false, 0, 0, 0, q + off); q += off; i2 = a[0]; j2 = a[1]; }
				else if (scan_units(q, "u//un", a, &off) == 2) { emit(false, i0, i2, a[0], false, 0, 0, 0, true, k0, k2, a[1], q + off); q += off; i2 = a[0]; k2 = a[1]; }
				else if (scan_units(q, "un", a, &off) == 1) { emit(false, i0, i2, a[0], false, 0, 0, 0, false, 0, 0, 0, q + off); q += off; i2 = a[0]; }
				else q++;
			}
		}
	}
}
}   // namespace
extern "C" void mh_set_obj_slicing(int slice_bytes, int max_slices) { g_obj_slice_bytes = std::max(1, slice_bytes); g_obj_max_slices = max_slices; }

bool TriMesh::readOBJ(const char* obj, bool load_textures) {
	std::vector<char> text;
	{
		FILE* f = fopen(obj, "rb");
		if (!f) { load_error = std::string("cannot open ") + obj; return false; }
		fseek(f, 0, SEEK_END);
		const long sz = ftell(f);
		fseek(f, 0, SEEK_SET);
		text.resize(sz > 0 ? (size_t)sz : 0);
		const size_t got = text.empty() ? 0 : fread(text.data(), 1, text.size(), f);
		fclose(f);
		text.resize(got);
	}
	// slices of about equal size, each starting right after a newline, parsed on the host's hardware threads
	const int nslices = std::max(1, std::min(g_obj_max_slices > 0 ? g_obj_max_slices : (int)std::thread::hardware_concurrency(), (int)(text.size() / (size_t)g_obj_slice_bytes)));
	std::vector<size_t> cut(nslices + 1, text.size());
	cut[0] = 0;
	for (int k = 1; k < nslices; k++) {
		size_t c = std::max(cut[k - 1], text.size() * k / nslices);
		while (c < text.size() && c > 0 && text[c - 1] != '\n') c++;
		cut[k] = c;
	}
	std::vector<ObjSlice> sl(nslices);
	{
		std::vector<std::thread> th;
		for (int k = 1; k < nslices; k++) th.emplace_back([&, k] { parse_obj_slice(text.data() + cut[k], text.data() + cut[k + 1], sl[k]); });
		parse_obj_slice(text.data() + cut[0], text.data() + cut[1], sl[0]);
		for (auto& x : th) x.join();
	}
	for (const ObjSlice& o : sl) if (o.vertex_colours) { load_error = "per-vertex colours are not supported"; return false; }
	// join: running counts for relative indices, groups numbered by first appearance, the last mtllib wins
	std::string matfile;
	std::vector<size_t> bv(nslices + 1, 0), bt(nslices + 1, 0), bn(nslices + 1, 0), bf(nslices + 1, 0);
	std::vector<std::vector<int>> group_id(nslices);
	std::vector<int> inherited(nslices, -1);
	int curGroup = -1;
	for (int k = 0; k < nslices; k++) {
		const ObjSlice& o = sl[k];
		bv[k + 1] = bv[k] + o.v.size(); bt[k + 1] = bt[k] + o.vt.size(); bn[k + 1] = bn[k] + o.vn.size(); bf[k + 1] = bf[k] + o.faces.size();
		inherited[k] = curGroup;
		for (const std::string& grp : o.usemtl) {
			auto it = groupNames.find(grp);
			if (it != groupNames.end()) curGroup = it->second;
			else { curGroup = (int)groupNames.size(); groupNames[grp] = curGroup; }
			group_id[k].push_back(curGroup);
		}
		if (o.has_matfile) matfile = o.matfile;
	}
	vertices.resize(bv[nslices]); uvs.resize(bt[nslices]); normals.resize(bn[nslices]); indices.resize(bf[nslices]);
	{
		auto join = [&](int k) {
			const ObjSlice& o = sl[k];
			std::copy(o.v.begin(), o.v.end(), vertices.begin() + bv[k]);
			std::copy(o.vt.begin(), o.vt.end(), uvs.begin() + bt[k]);
			std::copy(o.vn.begin(), o.vn.end(), normals.begin() + bn[k]);
			for (size_t i = 0; i < o.faces.size(); i++) {
				mipt_triangle_indices t = o.faces[i];
				const unsigned m = o.relmask[i];
				if (m) {
					int* fld[9] = {&t.vtxi, &t.vtxj, &t.vtxk, &t.uvi, &t.uvj, &t.uvk, &t.ni, &t.nj, &t.nk};
					for (int b = 0; b < 9; b++) if (m & (1u << b)) *fld[b] += (int)(b < 3 ? bv[k] : (b < 6 ? bt[k] : bn[k]));
				}
				t.group = o.face_group[i] < 0 ? inherited[k] : group_id[k][o.face_group[i]];
				indices[bf[k] + i] = t;
			}
		};
		std::vector<std::thread> th;
		for (int k = 1; k < nslices; k++) th.emplace_back(join, k);
		join(0);
		for (auto& x : th) x.join();
	}
	if (groupNames.empty()) {
		for (auto& t : indices) t.group = 0;
		groupNames["Default"] = 0;
	}
	if (!load_textures) return true;                  // init(load_textures = false): the caller supplies the material lists (.scn files)
	add_default_group_materials((int)groupNames.size());
	if (matfile.empty()) return true;
	FILE* m = fopen((dir_of(obj) + matfile).c_str(), "r");
	if (!m) return true;                               // a missing MTL is not an error in the reference either
	int grp = 0;
	char line[255];
	auto image = [&](Texture& tex, const std::string& file, bool normals_map) {
		std::vector<unsigned char> rgb; int W = 0, H = 0;
		std::string why;
		if (!read_image_rgb8(dir_of(obj) + file, rgb, W, H, why)) { load_error = "texture " + file + ": " + why; return; }
		if (normals_map) tex.loadNormalsRGB8(rgb.data(), W, H); else tex.loadColorsRGB8(rgb.data(), W, H);
		tex.filename = dir_of(obj) + file;              // Texture::filename, what save_scene writes
	};
	while (fgets(line, 255, m)) {
		if (line[0] == 'n' && line[1] == 'e' && line[2] == 'w') {
			auto it = groupNames.find(rest_of_line(line, 7));
			grp = it != groupNames.end() ? it->second : 0;
		}
		if (line[0] == 'm' && line[4] == 'K' && line[5] == 'd') image(textures[grp], rest_of_line(line, 7), false);
		if (line[0] == 'm' && line[4] == 'K' && line[5] == 's') image(specularmap[grp], rest_of_line(line, 7), false);
		if (line[0] == 'm' && line[4] == 'B' && line[5] == 'u') image(normal_map[grp], rest_of_line(line, 9), true);
		if (line[0] == 'm' && line[1] == 'a' && line[4] == 'd') image(alphamap[grp], rest_of_line(line, 6), false);
		if (line[0] == 'K' && line[1] == 'd') { Vector k; sscanf(line, "Kd %f %f %f", &k[0], &k[1], &k[2]); textures[grp].multiplier = k; }
		if (line[0] == 'K' && line[1] == 's') { Vector k; sscanf(line, "Ks %f %f %f", &k[0], &k[1], &k[2]); specularmap[grp].multiplier = k; }
		if (line[0] == 'N' && line[1] == 's') {
			Vector k;
			int got = sscanf(line, "Ns %f %f %f", &k[0], &k[1], &k[2]);
			if (got == 1) k = Vector(k[0], k[0], k[0]);
			roughnessmap[grp].multiplier = k;
		}
	}
	fclose(m);
	return true;
}

TriMesh::TriMesh(const char* obj, bool center, bool load_textures) {
	type = OT_TRIMESH; interp_normals = true; name = obj; is_centered = center;
	loaded = readOBJ(obj, load_textures);
	if (loaded && !indices.empty()) finish_init(center);
	else loaded = false;
}

void TriMesh::build_bbox(int i0, int i1, float* o) const {   // :843-858
	const Vector& f = vertices[indices[i0].vtxi];
	for (int k = 0; k < 3; k++) { o[k] = f[k]; o[3 + k] = f[k]; }
	auto range = [&](int a, int b, float* q) {
		for (int i = a; i < b; i++) {
			const int vi[3] = {indices[i].vtxi, indices[i].vtxj, indices[i].vtxk};
			for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) { q[k] = std::min(q[k], vertices[vi[c]][k]); q[3 + k] = std::max(q[3 + k], vertices[vi[c]][k]); }
		}
	};
	if (i1 - i0 < (1 << 18)) { range(i0, i1, o); return; }
	// whole-mesh boxes (build_bvh, the object's bbox): chunks on threads, joined in chunk order (same result: min / max
	// keep the earlier of equal values either way)
	std::mutex mu;
	std::map<int, std::array<float, 6>> part;
	parallel_for(i1 - i0, [&](int a, int b) {
		std::array<float, 6> q = {o[0], o[1], o[2], o[3], o[4], o[5]};
		range(i0 + a, i0 + b, q.data());
		std::lock_guard<std::mutex> g(mu);
		part[a] = q;
	});
	for (const auto& kv : part) for (int k = 0; k < 3; k++) { o[k] = std::min(o[k], kv.second[k]); o[3 + k] = std::max(o[3 + k], kv.second[3 + k]); }
}
void TriMesh::build_centers_bbox(int i0, int i1, float* o) const {   // :861-875
	auto center = [&](int i) { return divs(add(add(vertices[indices[i].vtxi], vertices[indices[i].vtxj]), vertices[indices[i].vtxk]), 3.f); };
	Vector c0 = center(i0);
	for (int k = 0; k < 3; k++) { o[k] = c0[k]; o[3 + k] = c0[k]; }
	for (int i = i0; i < i1; i++) {
		Vector c = center(i);
		for (int k = 0; k < 3; k++) { o[k] = std::min(o[k], c[k]); o[3 + k] = std::max(o[3 + k], c[k]); }
	}
}

// mipt_build_bvh (include/mipt.h) in place of the recursion: the device returns the node vector and the permutation
// the reference's swaps produce; it is applied to `indices` / `permuted_triangle_index` here.
bool TriMesh::build_bvh_gpu() {
	const int nf = (int)indices.size();
	const bool trace = getenv("MIPT_BUILD_TRACE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto phase = [&](const char* what) {
		if (!trace) return;
		const auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[build_bvh_gpu] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
		t_last = t;
	};
	bvh.nodes.resize((size_t)nf * 2);            // not zeroed (PodVec): pages are touched by the download only
	std::vector<int32_t> perm(nf);
	int nn = 0;
	double dev_s = 0;
	const int rc = mipt_build_bvh(g_bvh_builder_device, &vertices[0][0], (int)vertices.size(), &indices[0].vtxi, (int)sizeof(mipt_triangle_indices), nf,
	                              reinterpret_cast<mipt_bvh_node*>(bvh.nodes.data()), nf * 2, &nn, perm.data(), &dev_s);
	phase("mipt_build_bvh");
	if (rc != MIPT_OK) {
		bvh_gpu_unavailable = (rc != MIPT_ERR_INVALID);   // no device, out of memory, a degenerate mesh: the host recursion builds the same tree
		load_error = std::string("mipt_build_bvh: ") + mipt_build_bvh_error();
		bvh.nodes.clear();
		return false;
	}
	bvh.nodes.resize(nn);
	PodVec<mipt_triangle_indices> dst(nf);
	parallel_for(nf, [&](int i0, int i1) { for (int i = i0; i < i1; i++) { dst[i] = indices[perm[i]]; permuted_triangle_index[i] = perm[i]; } });
	indices.swap(dst);
	phase("permute indices");
	bvh_builder = 1;
	bvh_device_seconds = dev_s;
	load_error.clear();
	return true;
}

// build_bvh_recur (TriangleMesh.cpp:1029-1130): longest centroid axis, 16 candidate planes,
// cost area_L*n_L + area_R*n_R, in-place partition, <= 4 triangles per leaf, nodes in preorder.
//
// Same tree as the reference, built in parallel: the two children of a node work on disjoint index
// ranges, so a large right subtree is built on another thread into its own node vector (child
// references relative to that vector) and spliced behind the left subtree afterwards — preorder
// positions are then exactly what the serial push_back order gives.  The 16 candidate planes of a
// large node are evaluated concurrently as well (the first minimum in plane order wins, as in the loop).
namespace {
int kParallelSubtree = 1 << 15;      // ranges above this many triangles fork
int kParallelPlanes = 1 << 18;       // ranges above this evaluate the candidate planes on threads
}
extern "C" void mh_set_build_thresholds(int fork_tris, int planes_tris) { kParallelSubtree = fork_tris; kParallelPlanes = planes_tris; }

float TriMesh::split_cost(int i0, int i1, int split_dim, float split_val) const {
	float lmin[3] = {1E10f, 1E10f, 1E10f}, lmax[3] = {-1E10f, -1E10f, -1E10f}, rmin[3] = {1E10f, 1E10f, 1E10f}, rmax[3] = {-1E10f, -1E10f, -1E10f};
	int nl = 0, nr = 0;
	for (int i = i0; i < i1; i++) {
		const int vi[3] = {indices[i].vtxi, indices[i].vtxj, indices[i].vtxk};
		const float c = (vertices[vi[0]][split_dim] + vertices[vi[1]][split_dim] + vertices[vi[2]][split_dim]) / 3.f;   // (a+b+c)/3. narrowed == /3.f
		if (c <= split_val) {
			for (int c2 = 0; c2 < 3; c2++) for (int k = 0; k < 3; k++) { lmin[k] = std::min(lmin[k], vertices[vi[c2]][k]); lmax[k] = std::max(lmax[k], vertices[vi[c2]][k]); }
			nl++;
		} else {
			for (int c2 = 0; c2 < 3; c2++) for (int k = 0; k < 3; k++) { rmin[k] = std::min(rmin[k], vertices[vi[c2]][k]); rmax[k] = std::max(rmax[k], vertices[vi[c2]][k]); }
			nr++;
		}
	}
	auto area = [](const float* mn, const float* mx) { float s0 = mx[0] - mn[0], s1 = mx[1] - mn[1], s2 = mx[2] - mn[2]; return 2 * (s0 * s1 + s0 * s2 + s1 * s2); };
	return area(lmin, lmax) * nl + area(rmin, rmax) * nr;
}

void TriMesh::build_bvh_recur(PodVec<BVHNodes>& out, int i0, int i1, int depth) {
	const int node = (int)out.size();
	BVHNodes n;
	build_bbox(i0, i1, n.bbox);
	n.fg = i0; n.fd = i1; n.isleaf = true;
	out.push_back(n);
	float cb[6];
	build_centers_bbox(i0, i1, cb);
	const float diag[3] = {cb[3] - cb[0], cb[4] - cb[1], cb[5] - cb[2]};
	int split_dim;
	if (diag[0] >= diag[1] && diag[0] >= diag[2]) split_dim = 0;
	else if (diag[1] >= diag[0] && diag[1] >= diag[2]) split_dim = 1;
	else split_dim = 2;
	const int max_tests = 16;
	float cost[max_tests], factor[max_tests];
	for (int t = 0; t < max_tests; t++) factor[t] = (t + 1) / (float)(max_tests + 1);
	if (i1 - i0 >= kParallelPlanes) {
		std::vector<std::thread> th;
		for (int t = 0; t < max_tests; t++) th.emplace_back([&, t] { cost[t] = split_cost(i0, i1, split_dim, cb[split_dim] + diag[split_dim] * factor[t]); });
		for (auto& x : th) x.join();
	} else {
		for (int t = 0; t < max_tests; t++) cost[t] = split_cost(i0, i1, split_dim, cb[split_dim] + diag[split_dim] * factor[t]);
	}
	float best_split_factor = 0.5f;
	float best_area_bb = std::numeric_limits<float>::infinity();   // 1E50 narrowed
	for (int t = 0; t < max_tests; t++) if (cost[t] < best_area_bb) { best_split_factor = factor[t]; best_area_bb = cost[t]; }
	float split_val = cb[split_dim] + diag[split_dim] * best_split_factor;
	int pivot = i0 - 1;
	for (int i = i0; i < i1; i++) {
		const float c = (vertices[indices[i].vtxi][split_dim] + vertices[indices[i].vtxj][split_dim] + vertices[indices[i].vtxk][split_dim]) / 3.f;
		if (c <= split_val) {
			pivot++;
			std::swap(indices[i], indices[pivot]);
			std::swap(permuted_triangle_index[i], permuted_triangle_index[pivot]);
		}
	}
	if (pivot < i0 || pivot >= i1 - 1 || i1 <= i0 + 4) return;
	out[node].isleaf = false;
	if (i1 - (pivot + 1) >= kParallelSubtree && pivot + 1 - i0 >= kParallelSubtree) {
		PodVec<BVHNodes> right;
		std::thread worker([&] { build_bvh_recur(right, pivot + 1, i1, depth + 1); });
		out[node].fg = (int)out.size();
		build_bvh_recur(out, i0, pivot + 1, depth + 1);
		worker.join();
		const int off = (int)out.size();
		out[node].fd = off;
		for (BVHNodes& r : right) { if (!r.isleaf) { r.fg += off; r.fd += off; } out.push_back(r); }
	} else {
		out[node].fg = (int)out.size();
		build_bvh_recur(out, i0, pivot + 1, depth + 1);
		out[node].fd = (int)out.size();
		build_bvh_recur(out, pivot + 1, i1, depth + 1);
	}
}

// setup_tangents (TriangleMesh.cpp:601-711): only tangentSoup is read by the path (normal maps).
void TriMesh::setup_tangents() {
	// TriangleMesh.cpp:572-640.  The reference walks the faces once and adds every face's (sdir, tdir) into its three
	// vertices: per vertex the sum runs in ascending face order, which fixes its rounding.  Here the per-face vectors are
	// computed on all threads, a vertex -> incident corners table (ascending face order) is filled in one serial pass, and the
	// vertices then sum their own lists concurrently — the same additions in the same order.
	const int nv = (int)vertices.size(), nf = (int)indices.size();
	const bool trace = getenv("MIPT_BUILD_TRACE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto phase = [&](const char* what) {
		if (!trace) return;
		const auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[setup_tangents] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
		t_last = t;
	};
	PodVec<Vector> sdir(nf), tdir(nf);                       // (PodVec: no serial zero fill of 30 MB each; every used entry is written below)
	PodVec<unsigned char> has(nf);
	parallel_for(nf, [&](int a, int b) {
		for (int i = a; i < b; i++) {
			const auto& t = indices[i];
			has[i] = !(t.uvi == -1 || t.uvj == -1 || t.uvk == -1);
			if (!has[i]) continue;
			Vector vA = sub(vertices[t.vtxj], vertices[t.vtxi]), vB = sub(vertices[t.vtxk], vertices[t.vtxi]);
			Vector sA = sub(uvs[t.uvj], uvs[t.uvi]), sB = sub(uvs[t.uvk], uvs[t.uvi]);
			float det = (sA[0] * sB[1] - sB[0] * sA[1]);
			if (det != 0) { sdir[i] = divs(sub(mul(sB[1], vA), mul(sA[1], vB)), det); tdir[i] = divs(sub(mul(sA[0], vB), mul(sB[0], vA)), det); }
			else { sdir[i] = mul(0.00001f, vA); tdir[i] = mul(0.00001f, vB); }
		}
	});
	phase("per-face vectors");
	// vertex -> incident corners (3 * face + k): counted and filled on all threads with atomic cursors, so a vertex's list comes
	// out in arbitrary order; every vertex sorts its own short list before it sums (ascending face order = the reference's)
	std::vector<int> first((size_t)nv + 1, 0);
	parallel_for(nf, [&](int a, int b) {
		for (int i = a; i < b; i++) { __atomic_fetch_add(&first[indices[i].vtxi + 1], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&first[indices[i].vtxj + 1], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&first[indices[i].vtxk + 1], 1, __ATOMIC_RELAXED); }
	});
	phase("count");
	for (int v = 0; v < nv; v++) first[v + 1] += first[v];
	std::vector<int> fill(first.begin(), first.end() - 1);
	PodVec<int> corner((size_t)nf * 3);
	parallel_for(nf, [&](int a, int b) {
		for (int i = a; i < b; i++) {
			corner[__atomic_fetch_add(&fill[indices[i].vtxi], 1, __ATOMIC_RELAXED)] = 3 * i;
			corner[__atomic_fetch_add(&fill[indices[i].vtxj], 1, __ATOMIC_RELAXED)] = 3 * i + 1;
			corner[__atomic_fetch_add(&fill[indices[i].vtxk], 1, __ATOMIC_RELAXED)] = 3 * i + 2;
		}
	});
	phase("prefix + fill");
	PodVec<Vector> tangents(nv);
	parallel_for(nv, [&](int a, int b) {
		for (int v = a; v < b; v++) {
			Vector t1, t2;
			int nidx = 0;                                           // v2n: the normal index of the last corner that names the vertex (vertices no face names keep 0)
			std::sort(corner.begin() + first[v], corner.begin() + first[v + 1]);
			for (int e = first[v]; e < first[v + 1]; e++) {
				const int f = corner[e] / 3, k = corner[e] % 3;
				if (has[f]) { t1 = add(t1, sdir[f]); t2 = add(t2, tdir[f]); }
				nidx = k == 0 ? indices[f].ni : (k == 1 ? indices[f].nj : indices[f].nk);
			}
			(void)t2;
			Vector N = normalized(normals[nidx]);
			tangents[v] = normalized(sub(t1, mul(dot(t1, N), N)));
		}
	});
	phase("per-vertex sums");
	tangentSoup.resize((size_t)nf * 3);
	parallel_for(nf, [&](int a, int b) { for (int i = a; i < b; i++) { tangentSoup[3 * (size_t)i] = tangents[indices[i].vtxi]; tangentSoup[3 * (size_t)i + 1] = tangents[indices[i].vtxj]; tangentSoup[3 * (size_t)i + 2] = tangents[indices[i].vtxk]; } });
	phase("tangent soup");
}

// ---------------------------------------------------------------- Scene
Scene::~Scene() { for (Object* o : objects) delete o; }
void Scene::prepare_render() { for (Object* o : objects) o->build_matrix((float)current_frame); }   // Geometry.cpp:280-284

bool Scene::intersection(const mipt_ray& d, Vector& P, int& sphere_id, float& min_t, mipt_hit& mat, int& triangle_id) const {
	mipt_hit h;
	if (!owner || !owner->ctx || mipt_trace(owner->ctx, &d, 1, &h) != MIPT_OK) { memset(&mat, 0, sizeof mat); min_t = std::numeric_limits<float>::infinity(); return false; }
	mat = h; min_t = h.t;
	if (h.has_inter) { P = Vector(h.P[0], h.P[1], h.P[2]); sphere_id = h.object_id; triangle_id = h.triangle_id; }
	return h.has_inter != 0;
}
bool Scene::intersection_shadow(const mipt_ray& d, float& min_t, float dist_light) const {
	int32_t occ = 0;
	min_t = std::numeric_limits<float>::infinity();
	if (!owner || !owner->ctx || mipt_trace_shadow(owner->ctx, &d, &dist_light, 1, &occ) != MIPT_OK) return false;
	return occ != 0;
}

// ---------------------------------------------------------------- Raytracer
static double fast_exp(double y) {   // Raytracer.cpp:1294-1299
	double d; int32_t w[2]; w[0] = 0; w[1] = (int32_t)(1512775 * y + 1072632447); memcpy(&d, w, 8); return d;
}
static uint32_t ReverseBits(uint32_t n) {   // Raytracer.cpp:1302-1309
	n = (n << 16) | (n >> 16);
	n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
	n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
	n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
	n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
	return n;
}
static Vector extensibleLattice2d(uint32_t id) {   // Raytracer.cpp:1311-1319
	uint32_t rid = ReverseBits(id);
	float phi_id = (float)(rid * std::pow(2.0, -32));
	float tmp;
	float x = std::modf((float)(phi_id * 1 + 0.456789123), &tmp);
	float y = std::modf((float)(phi_id * 182667 + 0.123456789), &tmp);
	return Vector(x, y, 0);
}
// pcg32 (pcg_random.hpp: setseq_xsh_rr_64_32, default increment)
struct pcg32 {
	uint64_t state;
	explicit pcg32(uint64_t seed) { const uint64_t inc = 1442695040888963407ULL, mult = 6364136223846793005ULL; state = (seed + inc) * mult + inc; }
	uint32_t operator()() {
		uint64_t old = state;
		state = old * 6364136223846793005ULL + 1442695040888963407ULL;
		uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((0u - rot) & 31u));
	}
};

Raytracer::Raytracer() { s.owner = this; }
Raytracer::~Raytracer() { if (ctx) mipt_destroy(ctx); }

void Raytracer::loadScene() {   // Raytracer.cpp:1238-1274
	W = 1000; H = 800; nrays = 100;
	cam = Camera();
	cam.fov = (float)(35 * M_PI / 180);
	cam.focus_distance = 50; cam.aperture = 0.1f;
	sigma_filter = 0.5f; nb_bounces = 3;
	Sphere* slum = new Sphere(Vector(10, 23, 15), 10);
	Sphere* s2 = new Sphere(Vector(0, 0, 0), 1000000); s2->flip_normals = true;
	Plane* plane = new Plane(Vector(0, 0, 0), Vector(0.f, 1.f, 0.f));
	plane->max_translation = Vector(0.f, -27.3f, 0.f);
	s.addObject(slum); s.addObject(s2); s.addObject(plane);
	s.lumiere = slum;
	s.intensite_lumiere = (float)(1000000000 * 4. * M_PI / (4. * M_PI * s.lumiere->R * s.lumiere->R * M_PI));
	s.envmap_intensity = 1;
	// cam.rotate(0, -22 deg, 1) (Vector.h:725-750): cosf/sinf of float(-22*pi/180)
	float ay = (float)(-22 * M_PI / 180);
	float c = std::cos(ay), sn = std::sin(ay);
	Vector d = cam.direction, u = cam.up;
	cam.direction = Vector(d[0], c * d[1] - sn * d[2], sn * d[1] + c * d[2]);
	cam.up = Vector(u[0], c * u[1] - sn * u[2], sn * u[1] + c * u[2]);
}

// ---------------------------------------------------------------- .scn scene files
// The reference's own text format (Raytracer::save_scene / load_scene, Raytracer.cpp:1096-1236; Object::save_to_file /
// load_from_file, Geometry.h:455-662; Sphere / Plane / TriMesh tails, Geometry.h:875-908, 1193-1213, TriangleMesh.h:
// 132-162).  Values are written with "%f" (six decimals) and read back from those decimals, exactly like the reference,
// so a scene loaded here and there is the same scene.  The optional / backward-compatible records of load_scene are
// accepted.  What the hot path does not cover is refused loudly rather than dropped: a
// key-framed transforms, PointSet objects, per-face colour files.
namespace {
struct ScnReader {
	FILE* f;
	std::string err;
	char line[1024];
	bool next() {                                       // next non-empty line, without its newline
		while (fgets(line, sizeof line, f)) {
			size_t n = strlen(line);
			while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
			const char* p = line;
			while (*p == ' ' || *p == '\t') p++;
			if (*p) { if (p != line) memmove(line, p, strlen(p) + 1); return true; }
		}
		return false;
	}
	bool starts(const char* key) const { return strncmp(line, key, strlen(key)) == 0; }
	const char* after(const char* key) const { const char* p = line + strlen(key); while (*p == ' ') p++; return p; }
	bool fail(const std::string& what) { if (err.empty()) err = what + " (at: \"" + line + "\")"; return false; }
	bool expect(const char* key) { if (!next() || !starts(key)) return fail(std::string("expected \"") + key + "\""); return true; }
	bool getu(const char* key, int& v) { if (!expect(key)) return false; v = (int)strtoul(after(key), nullptr, 10); return true; }
	bool getf(const char* key, float& v) { if (!expect(key)) return false; v = strtof(after(key), nullptr); return true; }
	bool floats(const char* p, float* out, int n) {      // n numbers separated by anything that is not part of a number
		for (int k = 0; k < n; k++) {
			while (*p && !(*p == '-' || *p == '+' || *p == '.' || (*p >= '0' && *p <= '9') || *p == 'n' || *p == 'i')) p++;
			char* e; out[k] = strtof(p, &e);
			if (e == p) return false;
			p = e;
		}
		return true;
	}
};

void scn_texture_list(ScnReader& R, const char* count_key, bool count_already_read, std::vector<Texture>& list, int kind, const std::string& dir, bool& ok) {
	// kind: 0 colour image (gamma decoded), 2 normal map, 3 alpha (a bare number is a constant), 5 / 6 scalar multiplier "multiplier: %f)"
	int n = 0;
	if (!ok) return;
	if (count_already_read) n = (int)strtoul(R.after(count_key), nullptr, 10);
	else if (!R.getu(count_key, n)) { ok = false; return; }
	for (int i = 0; i < n && ok; i++) {
		if (!R.expect("texture:")) { ok = false; return; }
		std::string name = R.after("texture:");
		Texture t;
		t.filename = name;
		t.multiplier = (kind == 1) ? Vector(0, 0, 0) : (kind == 2 ? Vector(0, 0, 1) : Vector(1, 1, 1));
		float col[3];
		char* e = nullptr;
		if ((kind == 0 || kind == 1 || kind == 4) && name.compare(0, 6, "Color:") == 0 && R.floats(name.c_str() + 6, col, 3)) {
			t.multiplier = kind == 4 ? Vector(col[0], col[1], col[2]) : Vector(col[0] / 255.f, col[1] / 255.f, col[2] / 255.f);   // legacy "Color: (r, g, b)"
		} else if (kind == 3 && (strtof(name.c_str(), &e), e != name.c_str())) {
			float c = strtof(name.c_str(), nullptr); t.multiplier = Vector(c, c, c);
		} else if (name != "Null" && !name.empty()) {
			std::vector<unsigned char> rgb; int W = 0, H = 0;
			std::string file = (name[0] == '/' ? name : dir + name);
			FILE* probe = fopen(file.c_str(), "rb");
			if (probe) {                                   // a missing file leaves the constant, as load_image returning false does
				fclose(probe);
				std::string why;
				if (!read_image_rgb8(file, rgb, W, H, why)) { R.fail("texture " + name + ": " + why); ok = false; return; }
				if (kind == 2) t.loadNormalsRGB8(rgb.data(), W, H); else t.loadColorsRGB8(rgb.data(), W, H);
			}
		}
		if (!R.expect("multiplier:")) { ok = false; return; }
		float m[3];
		if (kind >= 5) { if (!R.floats(R.after("multiplier:"), m, 1)) { R.fail("bad multiplier"); ok = false; return; } t.multiplier[0] = m[0]; }
		else { if (!R.floats(R.after("multiplier:"), m, 3)) { R.fail("bad multiplier"); ok = false; return; } t.multiplier = Vector(m[0], m[1], m[2]); }
		list.push_back(t);
	}
}

bool scn_object_common(ScnReader& R, Object* o, const std::string& dir, const char* replacedNames = nullptr) {      // Object::load_from_file
	if (!R.expect("name:")) return false;
	o->name = R.after("name:");
	if (replacedNames) {                                // Geometry.h:524-526: the first '#' of the name becomes replacedNames
		const size_t at = o->name.find("#");            // (no '#': std::string::replace throws there and the program ends)
		if (at == std::string::npos) return R.fail("object name without a '#' to substitute \"" + std::string(replacedNames) + "\" for");
		o->name.replace(at, 1, std::string(replacedNames));
	}
	int b = 0;
	if (!R.getu("miroir:", b)) return false;
	o->miroir = b != 0;
	if (!R.next()) return R.fail("truncated object");
	if (R.starts("ghost:")) { o->ghost = strtoul(R.after("ghost:"), nullptr, 10) != 0; if (!R.expect("translation:")) return false; }
	else if (!R.starts("translation:")) return R.fail("expected translation");
	float v[9];
	if (!R.floats(R.after("translation:"), v, 3)) return R.fail("bad translation");
	o->max_translation = Vector(v[0], v[1], v[2]);
	if (!R.expect("rotation:") || !R.floats(R.after("rotation:"), v, 9)) return R.fail("bad rotation");
	for (int k = 0; k < 9; k++) o->mat_rotation[k] = v[k];
	if (!R.expect("center:") || !R.floats(R.after("center:"), v, 3)) return R.fail("bad center");
	o->rotation_center = Vector(v[0], v[1], v[2]);
	if (!R.getf("scale:", o->scale)) return false;
	if (!R.getu("display_edges:", b)) return false;
	if (!R.getu("interp_normals:", b)) return false;
	o->interp_normals = b != 0;
	if (!R.getu("flip_normals:", b)) return false;
	o->flip_normals = b != 0;
	if (!R.next()) return R.fail("truncated object");
	bool have_count = true;
	o->scale_keyframes.clear(); o->translation_keyframes.clear(); o->rotation_keyframes.clear();
	if (R.starts("nb_transforms:")) {                   // Geometry.h:549-571: n scales, n translations, n rotations, each behind its frame
		const int n = (int)strtoul(R.after("nb_transforms:"), nullptr, 10);
		float kv[10];
		for (int i = 0; i < n; i++) { if (!R.next() || !R.floats(R.line, kv, 2)) return R.fail("bad scale key frame"); o->scale_keyframes[kv[0]] = kv[1]; }
		for (int i = 0; i < n; i++) { if (!R.next() || !R.floats(R.line, kv, 4)) return R.fail("bad translation key frame"); o->translation_keyframes[kv[0]] = Vector(kv[1], kv[2], kv[3]); }
		for (int i = 0; i < n; i++) {
			if (!R.next() || !R.floats(R.line, kv, 10)) return R.fail("bad rotation key frame");
			std::array<float, 9> r; for (int k = 0; k < 9; k++) r[k] = kv[1 + k];
			o->rotation_keyframes[kv[0]] = r;
		}
		have_count = false;
	} else if (!R.starts("nb_textures:")) return R.fail("expected nb_textures");
	bool ok = true;
	scn_texture_list(R, "nb_textures:", have_count, o->textures, 0, dir, ok);
	scn_texture_list(R, "nb_normalmaps:", false, o->normal_map, 2, dir, ok);
	if (!ok) return false;
	if (!R.next()) return R.fail("truncated object");
	if (R.starts("nb_subsurfaces:")) { scn_texture_list(R, "nb_subsurfaces:", true, o->subsurface, 0, dir, ok); scn_texture_list(R, "nb_specularmaps:", false, o->specularmap, 1, dir, ok); }
	else if (R.starts("nb_specularmaps:")) scn_texture_list(R, "nb_specularmaps:", true, o->specularmap, 1, dir, ok);
	else return R.fail("expected nb_specularmaps");
	scn_texture_list(R, "nb_alphamaps:", false, o->alphamap, 3, dir, ok);
	scn_texture_list(R, "nb_expmaps:", false, o->roughnessmap, 4, dir, ok);
	scn_texture_list(R, "nb_transpmaps:", false, o->transparent_map, 5, dir, ok);
	scn_texture_list(R, "nb_refrindexmaps:", false, o->refr_index_map, 6, dir, ok);
	return ok;
}
}  // namespace

bool Raytracer::load_scene(const char* filename, const char* replacedNames) {
	ScnReader R; R.f = fopen(filename, "r");
	if (!R.f) { err_ = std::string("cannot open ") + filename; return false; }
	const std::string dir = dir_of(filename);
	auto bail = [&](const std::string& why) { if (R.f) fclose(R.f); err_ = std::string(filename) + ": " + (why.empty() ? R.err : why); return false; };
	for (Object* o : s.objects) delete o;
	s.objects.clear(); s.lumiere = nullptr;
	float v[9];
	if (!R.expect("W,H:") || !R.floats(R.after("W,H:"), v, 2)) return bail("");
	W = (int)v[0]; H = (int)v[1];
	if (!R.getu("nrays:", nrays)) return bail("");
	if (!R.next()) return bail("truncated header");
	if (R.starts("nbframes:")) { s.nbframes = (int)strtoul(R.after("nbframes:"), nullptr, 10); if (!R.expect("Cam:")) return bail(""); }
	else if (!R.starts("Cam:")) return bail("expected Cam");
	if (!R.floats(R.after("Cam:"), v, 9)) return bail("bad Cam");
	cam.position = Vector(v[0], v[1], v[2]); cam.direction = Vector(v[3], v[4], v[5]); cam.up = Vector(v[6], v[7], v[8]);
	if (!R.getf("fov:", cam.fov) || !R.getf("focus:", cam.focus_distance) || !R.getf("aperture:", cam.aperture) || !R.getf("sigma_filter:", sigma_filter) || !R.getf("gamma:", gamma)) return bail("");
	if (!R.next()) return bail("truncated header");
	if (R.starts("is_lenticular:")) {
		cam.is_lenticular = strtoul(R.after("is_lenticular:"), nullptr, 10) != 0;
		int u = 0; float fl = 0;
		if (!R.getu("lenticular_nb_images:", cam.lenticular_nb_images) || !R.getf("lenticular_max_angle:", cam.lenticular_max_angle) || !R.getu("lenticular_pixel_width:", cam.lenticular_pixel_width)) return bail("");
		// camera arrays: the GUI moves the camera from view to view and calls the renderer once per view (mainApp.cpp:886-905);
		// inside Raytracer the fields only name the exported files, so they are kept and written back
		if (!R.getu("isArray:", u)) return bail("");
		cam.isArray = u != 0;
		if (!R.getu("nbviewX:", cam.nbviewX) || !R.getu("nbviewY:", cam.nbviewY) || !R.getf("maxSpacingX:", cam.maxSpacingX) || !R.getf("maxSpacingY:", cam.maxSpacingY)) return bail("");
		(void)fl;
		if (!R.getu("bounces:", nb_bounces)) return bail("");
	} else if (R.starts("bounces:")) nb_bounces = (int)strtoul(R.after("bounces:"), nullptr, 10);
	else return bail("expected bounces");
	if (!R.next()) return bail("truncated header");
	if (R.starts("has_denoiser:")) { has_denoiser = strtoul(R.after("has_denoiser:"), nullptr, 10) != 0; if (!R.getf("intensite_lum:", s.intensite_lumiere)) return bail(""); }
	else if (R.starts("intensite_lum:")) s.intensite_lumiere = strtof(R.after("intensite_lum:"), nullptr);
	else return bail("expected intensite_lum");
	if (!R.getf("intensite_envmap:", s.envmap_intensity)) return bail("");
	if (!R.next()) return bail("truncated header");
	s.clear_background();                                                         // Raytracer.cpp:1205-1211
	if (R.starts("background:")) {
		std::string why;
		if (!s.load_background(R.after("background:"), gamma, why)) return bail("background " + std::string(R.after("background:")) + ": " + why);
		if (!R.next()) return bail("truncated header");
	}
	if (!R.starts("nbobjects:")) return bail("expected nbobjects");
	const int nbo = (int)strtoul(R.after("nbobjects:"), nullptr, 10);
	for (int i = 0; i < nbo; i++) {
		if (!R.next()) return bail("truncated object list");
		if (R.starts("NEW SPHERE")) {
			Sphere* sp = new Sphere(Vector(0, 0, 0), 0);
			s.addObject(sp);
			if (!scn_object_common(R, sp, dir)) return bail("");
			int has_env = 0;
			if (!R.getu("is_envmap:", has_env) || !R.expect("envmapfilename:")) return bail("");
			const std::string envfile = R.after("envmapfilename:");
			if (!R.expect("O:") || !R.floats(R.after("O:"), v, 3)) return bail("bad O");
			sp->O = Vector(v[0], v[1], v[2]);
			if (!R.getf("R:", sp->R)) return bail("");
			sp->rotation_center = sp->O; sp->name = "Sphere";           // Sphere::init (Geometry.h:856-873)
			if (has_env) {
				std::vector<unsigned char> rgb; int w = 0, h = 0;
				std::string why;
				if (!read_image_rgb8(envfile[0] == '/' ? envfile : dir + envfile, rgb, w, h, why)) return bail("environment map " + envfile + ": " + why);
				sp->load_envmap_rgb8(rgb.data(), w, h);
				sp->envmapfilename = envfile;
			}
		} else if (R.starts("NEW PLANE")) {
			Plane* pl = new Plane(Vector(0, 0, 0), Vector(0, 1, 0));
			s.addObject(pl);
			if (!scn_object_common(R, pl, dir)) return bail("");
			if (!R.expect("Point:") || !R.floats(R.after("Point:"), v, 3)) return bail("bad Point");
			pl->A = Vector(v[0], v[1], v[2]);
			if (!R.expect("N:") || !R.floats(R.after("N:"), v, 3)) return bail("bad N");
			pl->vecN = Vector(v[0], v[1], v[2]);
			pl->name = "Plane";
		} else if (R.starts("NEW MESH")) {
			Object tmp;                                                   // Object::load_from_file runs before TriMesh::init
			if (!scn_object_common(R, &tmp, dir, replacedNames)) return bail("");      // only meshes (and PointSets) substitute: Geometry.cpp:11-25
			if (!R.next()) return bail("truncated mesh");
			bool centered = true;
			int hascsv = 0;
			if (R.starts("is_centered:")) { centered = strtoul(R.after("is_centered:"), nullptr, 10) == 1; if (!R.getu("has_csv:", hascsv)) return bail(""); }
			else if (R.starts("has_csv:")) hascsv = (int)strtoul(R.after("has_csv:"), nullptr, 10);
			else return bail("expected has_csv");
			if (hascsv) return bail("per-face colour files are outside the hot path");
			R.next();                                                     // "csv_file: "
			const std::string file = tmp.name[0] == '/' ? tmp.name : dir + tmp.name;
			TriMesh* g = new TriMesh(file.c_str(), centered, /*load_textures=*/false);
			if (!g->loaded) { std::string why = g->load_error; delete g; return bail(why.empty() ? "no faces in " + file : why); }
			// the transform and material lists of the file replace what init() derived (TriMesh::create_from_file: the
			// lists were read first, init(load_textures = false) then keeps them; rot_center = the stored centre)
			g->name = tmp.name; g->miroir = tmp.miroir; g->ghost = tmp.ghost; g->flip_normals = tmp.flip_normals; g->interp_normals = true;
			g->scale = tmp.scale; g->max_translation = tmp.max_translation; g->rotation_center = tmp.rotation_center;
			memcpy(g->mat_rotation, tmp.mat_rotation, sizeof tmp.mat_rotation);
			g->scale_keyframes = tmp.scale_keyframes; g->translation_keyframes = tmp.translation_keyframes; g->rotation_keyframes = tmp.rotation_keyframes;
			g->textures = tmp.textures; g->normal_map = tmp.normal_map; g->subsurface = tmp.subsurface; g->specularmap = tmp.specularmap;
			g->alphamap = tmp.alphamap; g->roughnessmap = tmp.roughnessmap; g->transparent_map = tmp.transparent_map; g->refr_index_map = tmp.refr_index_map;
			s.addObject(g);
		} else return bail("object kind outside the hot path");
	}
	// fog block (:1217-1232; older files stop after fog_type or fog_phase_type)
	if (R.next() && R.starts("fog_density:")) s.fog_density = strtof(R.after("fog_density:"), nullptr);
	while (R.next()) {
		if (R.starts("fog_absorption:")) s.fog_absorption = strtof(R.after("fog_absorption:"), nullptr);
		else if (R.starts("fog_density_decay:")) s.fog_density_decay = strtof(R.after("fog_density_decay:"), nullptr);
		else if (R.starts("fog_absorption_decay:")) s.fog_absorption_decay = strtof(R.after("fog_absorption_decay:"), nullptr);
		else if (R.starts("fog_type:")) s.fog_type = (int)strtoul(R.after("fog_type:"), nullptr, 10);
		else if (R.starts("fog_phase_type:")) s.fog_phase_type = (int)strtoul(R.after("fog_phase_type:"), nullptr, 10);
		else if (R.starts("double_frustum_start_t:")) s.double_frustum_start_t = strtof(R.after("double_frustum_start_t:"), nullptr);
	}
	fclose(R.f); R.f = nullptr;
	if (s.objects.empty() || s.objects[0]->type != OT_SPHERE) { err_ = std::string(filename) + ": object 0 must be the light sphere"; return false; }
	s.lumiere = static_cast<Sphere*>(s.objects[0]);
	last_nrays = -1; lastfilter = -1; randomPerPixel.clear(); clear_image();
	err_.clear();
	return true;
}

bool Raytracer::save_scene(const char* filename) const {
	FILE* f = fopen(filename, "w+");
	if (!f) return false;
	fprintf(f, "W,H: %u, %u\n", W, H);
	fprintf(f, "nrays: %u\n", nrays);
	fprintf(f, "nbframes: %u\n", (unsigned)s.nbframes);
	fprintf(f, "Cam: (%f, %f, %f), (%f, %f, %f), (%f, %f, %f)\n", cam.position[0], cam.position[1], cam.position[2], cam.direction[0], cam.direction[1], cam.direction[2], cam.up[0], cam.up[1], cam.up[2]);
	fprintf(f, "fov: %f\nfocus: %f\naperture: %f\nsigma_filter: %f\ngamma: %f\n", cam.fov, cam.focus_distance, cam.aperture, sigma_filter, gamma);
	fprintf(f, "is_lenticular: %u\nlenticular_nb_images: %u\nlenticular_max_angle: %f\nlenticular_pixel_width: %u\nisArray: %u\nnbviewX: %u\nnbviewY: %u\nmaxSpacingX: %f\nmaxSpacingY: %f\n",
	        cam.is_lenticular ? 1u : 0u, (unsigned)cam.lenticular_nb_images, cam.lenticular_max_angle, (unsigned)cam.lenticular_pixel_width,
	        cam.isArray ? 1u : 0u, (unsigned)cam.nbviewX, (unsigned)cam.nbviewY, cam.maxSpacingX, cam.maxSpacingY);
	fprintf(f, "bounces: %u\nhas_denoiser: %u\n", nb_bounces, has_denoiser ? 1u : 0u);
	fprintf(f, "intensite_lum: %f\nintensite_envmap: %f\n", s.intensite_lumiere, s.envmap_intensity);
	if (s.backgroundfilename.size() > 0) fprintf(f, "background: %s\n", s.backgroundfilename.c_str());   // :1125-1126
	fprintf(f, "nbobjects: %u\n", (unsigned)s.objects.size());
	auto list3 = [&](const char* key, const std::vector<Texture>& l) {
		fprintf(f, "%s: %u\n", key, (unsigned)l.size());
		for (const Texture& t : l) fprintf(f, "texture: %s\nmultiplier: (%f, %f, %f)\n", t.filename.empty() ? "Null" : t.filename.c_str(), t.multiplier[0], t.multiplier[1], t.multiplier[2]);
	};
	auto list1 = [&](const char* key, const std::vector<Texture>& l) {
		fprintf(f, "%s: %u\n", key, (unsigned)l.size());
		for (const Texture& t : l) fprintf(f, "texture: %s\nmultiplier: %f)\n", t.filename.empty() ? "Null" : t.filename.c_str(), t.multiplier[0]);
	};
	for (const Object* o : s.objects) {
		fprintf(f, o->type == OT_SPHERE ? "NEW SPHERE\n" : o->type == OT_PLANE ? "NEW PLANE\n" : "NEW MESH\n");
		fprintf(f, "name: %s\nmiroir: %u\nghost: %u\n", o->name.c_str(), o->miroir ? 1 : 0, o->ghost ? 1 : 0);
		fprintf(f, "translation: (%f, %f, %f)\n", o->max_translation[0], o->max_translation[1], o->max_translation[2]);
		const float* m = o->mat_rotation;
		fprintf(f, "rotation: (%f, %f, %f, %f, %f, %f, %f, %f, %f)\n", m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]);
		fprintf(f, "center: (%f, %f, %f)\n", o->rotation_center[0], o->rotation_center[1], o->rotation_center[2]);
		fprintf(f, "scale: %f\ndisplay_edges: 0\ninterp_normals: %u\nflip_normals: %u\nnb_transforms: %u\n", o->scale, o->interp_normals ? 1 : 0, o->flip_normals ? 1 : 0, (unsigned)o->translation_keyframes.size());
		for (const auto& kf : o->scale_keyframes) fprintf(f, "%f %f\n", kf.first, kf.second);                                  // Geometry.h:467-475
		for (const auto& kf : o->translation_keyframes) fprintf(f, "%f %f, %f, %f\n", kf.first, kf.second[0], kf.second[1], kf.second[2]);
		for (const auto& kf : o->rotation_keyframes) { const float* q = kf.second.data(); fprintf(f, "%f %f, %f, %f, %f, %f, %f, %f, %f, %f\n", kf.first, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]); }
		list3("nb_textures", o->textures); list3("nb_normalmaps", o->normal_map); list3("nb_subsurfaces", o->subsurface); list3("nb_specularmaps", o->specularmap);
		list3("nb_alphamaps", o->alphamap); list3("nb_expmaps", o->roughnessmap); list1("nb_transpmaps", o->transparent_map); list1("nb_refrindexmaps", o->refr_index_map);
		if (o->type == OT_SPHERE) {
			const Sphere* sp = static_cast<const Sphere*>(o);
			fprintf(f, "is_envmap: %u\nenvmapfilename: %s\nO: (%f, %f, %f)\nR: %f\n", sp->has_envmap ? 1 : 0, sp->envmapfilename.c_str(), sp->O[0], sp->O[1], sp->O[2], sp->R);
		} else if (o->type == OT_PLANE) {
			const Plane* pl = static_cast<const Plane*>(o);
			fprintf(f, "Point: (%f, %f, %f)\nN: (%f, %f, %f)\n", pl->A[0], pl->A[1], pl->A[2], pl->vecN[0], pl->vecN[1], pl->vecN[2]);
		} else {
			fprintf(f, "is_centered: %u\nhas_csv: 0\ncsv_file: \n", static_cast<const TriMesh*>(o)->is_centered ? 1 : 0);
		}
	}
	fprintf(f, "fog_density: %f\nfog_absorption: %f\nfog_density_decay: %f\nfog_absorption_decay: %f\nfog_type: %u\nfog_phase_type: %u\n",
	        s.fog_density, s.fog_absorption, s.fog_density_decay, s.fog_absorption_decay, (unsigned)s.fog_type, (unsigned)s.fog_phase_type);
	fprintf(f, "double_frustum_start_t: %f\n", s.double_frustum_start_t);
	fclose(f);
	return true;
}

bool Scene::load_background(const char* filename, float gamma, std::string& why) {   // Geometry.h:1355-1363
	g_content_epoch++;
	std::vector<unsigned char> bg; int w = 0, h = 0;
	if (!read_image_rgb8(filename, bg, w, h, why)) return false;
	backgroundfilename = filename;
	backgroundW = w; backgroundH = h;
	background.resize(bg.size());
	for (int i = 0; i < h; i++) for (int j = 0; j < w * 3; j++)                // load_image flips the rows (utils.cpp:112-118)
		background[(size_t)i * w * 3 + j] = (float)(std::pow(bg[(size_t)(h - 1 - i) * w * 3 + j] / 255., gamma) * 196964.699);
	return true;
}

void Raytracer::clear_image() {
	image.assign((size_t)W * H * 3, 0);
	imagedouble.assign((size_t)W * H * 3, 0.f);
	sample_count.assign((size_t)W * H, 0.f);
}

int Raytracer::open_device(int device_id) { return open_devices(&device_id, 1); }
// Several devices behind the one context: render_image / render_image_nopreviz then use all of them (tiles dealt to the
// devices, one RCCL reduce of the partial framebuffers inside mipt_render) with no other change on this side.
int Raytracer::open_devices(const int* device_ids, int n) {
	if (ctx) { mipt_destroy(ctx); ctx = nullptr; }
	uploaded_ = false;
	last_status = mipt_create(device_ids, n, &ctx);
	if (last_status != MIPT_OK) err_ = "mipt_create failed (no usable HIP device: this library has no CPU path)";
	else g_bvh_builder_device = device_ids[0];   // meshes loaded by this process build their BVH on the (first) GPU it renders on
	return last_status;
}
const char* Raytracer::last_error() const { return (ctx && last_status != MIPT_OK && err_.empty()) ? mipt_last_error(ctx) : err_.c_str(); }

void Raytracer::prepare_render(float time) {   // Raytracer.cpp:1321-1391
	pcg32 engine0(0);                       // engine[0] = pcg32(0) (:1325-1327)
	const float invmax = 1.f / 4294967296.f;   // Raytracer.h:28
	if (randomPerPixel.size() != (size_t)W * H) {
		randomPerPixel.resize((size_t)W * H);
		for (size_t i = 0; i < (size_t)W * H; i++) { randomPerPixel[i][0] = engine0() * invmax; randomPerPixel[i][1] = engine0() * invmax; }
	}
	if (image.size() != (size_t)W * H * 3) clear_image();
	if (nrays != last_nrays) {
		samples2d.resize(nrays);
		for (int i = 0; i < nrays; i++) samples2d[i] = extensibleLattice2d((uint32_t)i);
		last_nrays = nrays;
	}
	if (sigma_filter != lastfilter) {       // :1354-1374
		filter_size = (int)std::ceil(sigma_filter * 2);
		filter_total_width = 2 * filter_size + 1;
		filter_integral.assign((size_t)filter_total_width * filter_total_width, 0.f);
		filter_value.assign((size_t)filter_total_width * filter_total_width, 0.f);
		for (int i = -filter_size; i <= filter_size; i++) for (int j = -filter_size; j <= filter_size; j++) {
			float integ = 0;
			for (int i2 = -filter_size; i2 <= i; i2++) for (int j2 = -filter_size; j2 <= j; j2++) {
				float w = (float)(fast_exp(-(i2 * i2 + j2 * j2) / (2. * sigma_filter * sigma_filter)) / (sigma_filter * sigma_filter * 2. * M_PI));
				integ += w;
			}
			filter_integral[(i + filter_size) * filter_total_width + (j + filter_size)] = integ;
			filter_value[(i + filter_size) * filter_total_width + (j + filter_size)] = (float)(std::exp(-(i * i + j * j) / (2. * sigma_filter * sigma_filter)) / (sigma_filter * sigma_filter * 2. * M_PI));
		}
		lastfilter = sigma_filter;
	}
	s.prepare_render();
	centerLight = s.lumiere->apply_transformation(s.lumiere->O);   // :1377-1380
	lum_scale = s.lumiere->get_scale(time);
	radiusLight = lum_scale * s.lumiere->R;
	lightPower = s.intensite_lumiere / (lum_scale * lum_scale);
	std::fill(sample_count.begin(), sample_count.end(), 0.f);
	std::fill(imagedouble.begin(), imagedouble.end(), 0.f);
	build_descs();
}

static void tex_list(const std::vector<Texture>& in, std::vector<mipt_texture>& out) {
	out.resize(in.size());
	for (size_t k = 0; k < in.size(); k++) {
		for (int c = 0; c < 3; c++) out[k].multiplier[c] = in[k].multiplier[c];
		out[k].W = (int32_t)in[k].W; out[k].H = (int32_t)in[k].H;
		out[k].values = in[k].W > 0 ? in[k].values.data() : nullptr;
	}
}

// The reference-side binding of INTEGRATION.md: fill the POD descriptions from the live objects.
void Raytracer::build_descs() {
	const size_t n = s.objects.size();
	desc_objects_.assign(n, mipt_object{});
	desc_meshes_.assign(n, mipt_mesh{});
	desc_tex_.assign(n * 8, {});
	for (size_t i = 0; i < n; i++) {
		Object* o = s.objects[i];
		mipt_object& d = desc_objects_[i];
		d.type = o->type; d.miroir = o->miroir; d.ghost = o->ghost; d.flip_normals = o->flip_normals; d.interp_normals = o->interp_normals;
		memcpy(d.trans_matrix, o->trans_matrix, 48); memcpy(d.inv_trans_matrix, o->inv_trans_matrix, 48); memcpy(d.rot_matrix, o->rot_matrix, 36);
		d.brdf_kind = o->merl_data.empty() ? MIPT_BRDF_PHONG : MIPT_BRDF_MERL;
		d.merl_data = o->merl_data.empty() ? nullptr : o->merl_data.data();
		const std::vector<Texture>* lists[8] = {&o->textures, &o->specularmap, &o->alphamap, &o->roughnessmap, &o->normal_map, &o->subsurface, &o->transparent_map, &o->refr_index_map};
		for (int l = 0; l < 8; l++) tex_list(*lists[l], desc_tex_[i * 8 + l]);
		auto ptr = [&](int l) { return desc_tex_[i * 8 + l].empty() ? nullptr : desc_tex_[i * 8 + l].data(); };
		d.n_textures = (int)o->textures.size(); d.textures = ptr(0);
		d.n_specularmap = (int)o->specularmap.size(); d.specularmap = ptr(1);
		d.n_alphamap = (int)o->alphamap.size(); d.alphamap = ptr(2);
		d.n_roughnessmap = (int)o->roughnessmap.size(); d.roughnessmap = ptr(3);
		d.n_normal_map = (int)o->normal_map.size(); d.normal_map = ptr(4);
		d.n_subsurface = (int)o->subsurface.size(); d.subsurface = ptr(5);
		d.n_transparent_map = (int)o->transparent_map.size(); d.transparent_map = ptr(6);
		d.n_refr_index_map = (int)o->refr_index_map.size(); d.refr_index_map = ptr(7);
		if (o->type == OT_SPHERE) {
			Sphere* sp = static_cast<Sphere*>(o);
			for (int k = 0; k < 3; k++) d.O[k] = sp->O[k];
			d.R = sp->R; d.has_envmap = sp->has_envmap; d.envW = sp->envW; d.envH = sp->envH; d.envtex = sp->has_envmap ? sp->envtex.data() : nullptr;
		} else if (o->type == OT_PLANE) {
			Plane* pl = static_cast<Plane*>(o);
			for (int k = 0; k < 3; k++) { d.A[k] = pl->A[k]; d.vecN[k] = pl->vecN[k]; }
		} else {
			TriMesh* g = static_cast<TriMesh*>(o);
			mipt_mesh& m = desc_meshes_[i];
			memset(&m, 0, sizeof m);
			m.n_triangles = (int)g->indices.size(); m.n_nodes = g->node_count(); m.n_uvs = (int)g->uvs.size();
			memcpy(m.bvh_bbox_min, g->bvh.bbox, 12); memcpy(m.bvh_bbox_max, g->bvh.bbox + 3, 12);
			m.uvs = g->uvs.empty() ? nullptr : &g->uvs[0][0];
			if (g->device_handle()) {
				// the records are on the device already (mipt_device_mesh_build): the upload copies them device to device; the host views
				// (bvh.nodes, triangleSoup, the reordered indices) are not needed and not materialised
				m.device_mesh = g->device_handle();
				if (!g->uvs.empty() && g->normals.empty()) { g->sync_tangents(); m.tangentSoup = g->tangentSoup.empty() ? nullptr : &g->tangentSoup[0][0]; }   // (UVs without normals: tangents from the host loop)
			} else {
				m.nodes = reinterpret_cast<const mipt_bvh_node*>(g->bvh.nodes.data());
				m.triangleSoup = g->triangleSoup.data(); m.indices = g->indices.data();
				m.tangentSoup = g->tangentSoup.empty() ? nullptr : &g->tangentSoup[0][0];
			}
			d.mesh = &m;
		}
	}
	scene_desc.n_objects = (int)n; scene_desc.objects = desc_objects_.data();
	const bool has_bg = s.backgroundW > 0 && s.background.size() == (size_t)s.backgroundW * s.backgroundH * 3;   // Raytracer.cpp:220
	scene_desc.background = has_bg ? s.background.data() : nullptr;
	scene_desc.backgroundW = has_bg ? s.backgroundW : 0; scene_desc.backgroundH = has_bg ? s.backgroundH : 0;
	scene_desc.fog_density = s.fog_density; scene_desc.fog_absorption = s.fog_absorption; scene_desc.fog_density_decay = s.fog_density_decay;
	scene_desc.fog_absorption_decay = s.fog_absorption_decay; scene_desc.phase_aniso = s.phase_aniso; scene_desc.fog_type = s.fog_type; scene_desc.fog_phase_type = s.fog_phase_type;
	scene_desc.fog_ground_level = n > 2 ? s.objects[2]->get_translation((float)s.current_frame)[1] : 0.f;   // objects[2]->get_translation(time, is_recording)[1] (Raytracer.cpp:55)
	mipt_render_params& p = render_params;
	memset(&p, 0, sizeof p);
	p.W = W; p.H = H; p.nrays = nrays; p.nb_bounces = nb_bounces;
	for (int k = 0; k < 3; k++) { p.cam_position[k] = cam.position[k]; p.cam_direction[k] = cam.direction[k]; p.cam_up[k] = cam.up[k]; p.centerLight[k] = centerLight[k]; }
	p.cam_fov = cam.fov; p.cam_focus_distance = cam.focus_distance; p.cam_aperture = cam.aperture;
	p.double_frustum_start_t = s.double_frustum_start_t;
	p.sigma_filter = sigma_filter; p.filter_size = filter_size; p.filter_integral = filter_integral.data();
	p.samples2d = &samples2d[0][0]; p.randomPerPixel = &randomPerPixel[0][0];
	p.radiusLight = radiusLight; p.lightPower = lightPower; p.envmap_intensity = s.envmap_intensity;
	p.seed_stride = seed_stride; p.sample_begin = 0; p.sample_end = nrays;
	p.tile_size = tile_size_; p.tile_rank = tile_rank_; p.tile_nranks = tile_nranks_;
	p.is_lenticular = cam.is_lenticular ? 1 : 0; p.lenticular_nb_images = cam.lenticular_nb_images;
	p.lenticular_pixel_width = cam.lenticular_pixel_width; p.lenticular_max_angle = cam.lenticular_max_angle;
}

void Raytracer::tone_map(bool divided) {   // Raytracer.cpp:1540-1547 / 1701-1708
	const int W_ = W, H_ = H;
	const int nt = std::max(1, std::min((int)std::thread::hardware_concurrency(), H_ / 16));   // #pragma omp parallel for over the rows in the reference
	auto rows = [&](int i0, int i1) {
		for (int i = i0; i < i1; i++) for (int j = 0; j < W_; j++) for (int c = 0; c < 3; c++) {
			size_t idx = ((size_t)(H_ - i - 1) * W_ + j) * 3 + c;
			double v = divided ? imagedouble[idx] / 196964.7 : imagedouble[idx] / 196964.7 / std::max(sample_count[(size_t)(H_ - i - 1) * W_ + j], 1.f);
			image[idx] = (unsigned char)std::min(255., std::max(0., 255. * std::pow(v, (double)(1 / gamma))));
		}
	};
	if (nt == 1) { rows(0, H_); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < nt; t++) th.emplace_back(rows, (int)((long long)H_ * t / nt), (int)((long long)H_ * (t + 1) / nt));
	for (auto& x : th) x.join();
}

// FNV-1a over everything mipt_upload_scene receives by value, the addresses / sizes of what it receives by pointer, and
// the epoch of in-place rewrites of bulk data (the staging vectors of the material lists are hashed by content: they are
// rebuilt by every prepare_render)
uint64_t Raytracer::scene_fingerprint() const {
	uint64_t hsh = 1469598103934665603ull;
	auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { hsh ^= b[i]; hsh *= 1099511628211ull; } };
	mix(&g_content_epoch, sizeof g_content_epoch);
	mipt_scene_desc sd = scene_desc; sd.objects = nullptr;
	mix(&sd, sizeof sd);
	for (size_t i = 0; i < desc_objects_.size(); i++) {
		mipt_object o = desc_objects_[i];
		o.textures = o.specularmap = o.alphamap = o.roughnessmap = o.normal_map = o.subsurface = o.transparent_map = o.refr_index_map = nullptr;
		const mipt_mesh* m = o.mesh; o.mesh = nullptr;
		mix(&o, sizeof o);
		for (int l = 0; l < 8; l++) { const auto& v = desc_tex_[i * 8 + l]; if (!v.empty()) mix(v.data(), v.size() * sizeof(mipt_texture)); }
		if (m) mix(m, sizeof *m);
	}
	return hsh;
}
int Raytracer::upload_scene_if_changed() {
	const uint64_t fp = scene_fingerprint();
	if (uploaded_ && fp == uploaded_fingerprint_) return MIPT_OK;
	const int rc = mipt_upload_scene(ctx, &scene_desc);
	uploaded_ = (rc == MIPT_OK); uploaded_fingerprint_ = fp;
	return rc;
}

// Progressive render: one pass per sample index, buffers valid after every pass, `stopped`
// polled between passes (Raytracer.cpp:1444-1453).
void Raytracer::render_image() {
	prepare_render((float)s.current_frame);
	if (!ctx) { last_status = MIPT_ERR_NO_DEVICE; err_ = "no device opened"; return; }
	err_.clear();
	if ((last_status = upload_scene_if_changed()) != MIPT_OK) return;
	stopped = 0;
	// the sample loop of :1444-1531 as ONE call: one pass per sample, imagedouble / sample_count hold the running sums
	// after every pass (the GUI thread reads them while this runs), stopRender() ends it between two passes
	mipt_set_option(ctx, "samples_per_pass", 1);
	mipt_set_option(ctx, "progressive_lookahead", progressive_lookahead >= 0 ? progressive_lookahead : 0);   // (0, the default: the library sizes a pass to 64 M paths — 31 samples, ~35 ms of rendering at 1080p; every sample is still splatted, published and reported on its own)
	current_nb_rays = 0;
	last_status = mipt_render(ctx, &render_params, imagedouble.data(), sample_count.data(),
	                          [](void* self, int done, int) { static_cast<Raytracer*>(self)->current_nb_rays = done; }, this, &stopped);
	mipt_set_option(ctx, "samples_per_pass", 0);
	if (last_status == MIPT_ERR_CANCELLED) { last_status = MIPT_OK; return; }   // if (stopped) return; (:1452)
	if (last_status != MIPT_OK) return;
	tone_map(false);
	stopped = 1;
}

// Offline render: all samples in as few passes as memory allows, then imagedouble /= sample_count
// (Raytracer.cpp:1687-1694) and the tone map.
void Raytracer::render_image_nopreviz() {
	prepare_render((float)s.current_frame);
	if (!ctx) { last_status = MIPT_ERR_NO_DEVICE; err_ = "no device opened"; return; }
	err_.clear();
	if ((last_status = upload_scene_if_changed()) != MIPT_OK) return;
	stopped = 0;
	if (has_denoiser) {   // :1631-1645, 1676-1696; the denoiser itself (OpenImageDenoise, :1721-1737) is not part of the path
		const size_t npx = (size_t)W * H;
		albedoImage.assign(npx * 3, 0.f); normalImage.assign(npx * 3, 0.f); shadingNormalImage.assign(npx * 3, 0.f);
		if ((last_status = mipt_render_denoiser_inputs(ctx, &render_params, imagedouble.data(), sample_count.data(), albedoImage.data(), shadingNormalImage.data())) != MIPT_OK) return;
		for (size_t i = 0; i < npx * 3; i++) normalImage[i] += imagedouble[i];   // normalImage += imagedoublethreads (:1680)
		for (size_t i = 0; i < npx; i++) {
			const float nn = std::sqrt(normalImage[i * 3] * normalImage[i * 3] + normalImage[i * 3 + 1] * normalImage[i * 3 + 1] + normalImage[i * 3 + 2] * normalImage[i * 3 + 2]);
			const float ns = std::sqrt(shadingNormalImage[i * 3] * shadingNormalImage[i * 3] + shadingNormalImage[i * 3 + 1] * shadingNormalImage[i * 3 + 1] + shadingNormalImage[i * 3 + 2] * shadingNormalImage[i * 3 + 2]);
			for (int j = 0; j < 3; j++) {
				imagedouble[i * 3 + j] /= sample_count[i];
				albedoImage[i * 3 + j] /= sample_count[i];
				normalImage[i * 3 + j] /= nn;
				shadingNormalImage[i * 3 + j] /= ns;
			}
		}
		tone_map(true);
		return;
	}
	if ((last_status = mipt_render(ctx, &render_params, imagedouble.data(), sample_count.data(), nullptr, nullptr, &stopped)) != MIPT_OK) return;
	for (size_t i = 0; i < (size_t)W * H; i++) for (int j = 0; j < 3; j++) imagedouble[i * 3 + j] /= sample_count[i];
	tone_map(true);
}

}  // namespace mipt_host

// ---------------------------------------------------------------- flat C view
using namespace mipt_host;
struct mh_raytracer { Raytracer rt; };

extern "C" {
mh_raytracer* mh_create(void) { mh_raytracer* h = new mh_raytracer; h->rt.loadScene(); return h; }
void mh_destroy(mh_raytracer* h) { delete h; }
int mh_open_device(mh_raytracer* h, int device_id) { return h->rt.open_device(device_id); }
int mh_open_devices(mh_raytracer* h, const int* device_ids, int n) { return h->rt.open_devices(device_ids, n); }
void mh_set_partition(mh_raytracer* h, int ts, int rank, int nranks) { h->rt.set_partition(ts, rank, nranks); }
void mh_set_render(mh_raytracer* h, int W, int H, int nrays, int nb_bounces, float sigma) {
	Raytracer& r = h->rt; r.W = W; r.H = H; r.nrays = nrays; r.nb_bounces = nb_bounces; r.sigma_filter = sigma;
	r.last_nrays = -1; r.lastfilter = -1; r.randomPerPixel.clear(); r.clear_image();
}
void mh_set_camera(mh_raytracer* h, const float* pos, const float* dir, const float* up, float fov, float focus, float aperture) {
	Camera& c = h->rt.cam;
	c.position = Vector(pos[0], pos[1], pos[2]); c.direction = Vector(dir[0], dir[1], dir[2]); c.up = Vector(up[0], up[1], up[2]);
	c.fov = fov; c.focus_distance = focus; c.aperture = aperture;
}
void mh_set_light(mh_raytracer* h, const float* center, float R, float intensite) {
	Sphere* l = h->rt.s.lumiere; l->O = Vector(center[0], center[1], center[2]); l->R = R; l->rotation_center = l->O; h->rt.s.intensite_lumiere = intensite;
}
void mh_set_fog(mh_raytracer* h, float density, float absorption, float density_decay, float absorption_decay, int type, int phase_type, float phase_aniso) {
	Scene& s = h->rt.s;
	s.fog_density = density; s.fog_absorption = absorption; s.fog_density_decay = density_decay; s.fog_absorption_decay = absorption_decay;
	s.fog_type = type; s.fog_phase_type = phase_type; s.phase_aniso = phase_aniso;
}
void mh_set_object_ghost(mh_raytracer* h, int obj, int ghost) { h->rt.s.objects[obj]->ghost = ghost != 0; }
int mh_load_background(mh_raytracer* h, const char* file) {
	std::string why;
	if (!h->rt.s.load_background(file, h->rt.gamma, why)) { h->rt.set_error(std::string("background ") + file + ": " + why); return -1; }
	return 0;
}
void mh_set_background(mh_raytracer* h, const float* rgb, int W, int H) {
	Scene& s = h->rt.s;
	g_content_epoch++;
	s.clear_background();
	if (rgb && W > 0 && H > 0) { s.background.assign(rgb, rgb + (size_t)W * H * 3); s.backgroundW = W; s.backgroundH = H; }
}
int mh_get_background(mh_raytracer* h, float* out, int capacity, int* W, int* H) {
	const Scene& s = h->rt.s;
	*W = s.backgroundW; *H = s.backgroundH;
	if ((int)s.background.size() > capacity) return -1;
	if (out && !s.background.empty()) memcpy(out, s.background.data(), s.background.size() * sizeof(float));
	return (int)s.background.size();
}
void mh_set_lenticular(mh_raytracer* h, int on, int nb_images, float max_angle, int pixel_width) {
	Camera& c = h->rt.cam;
	c.is_lenticular = on != 0; c.lenticular_nb_images = nb_images; c.lenticular_max_angle = max_angle; c.lenticular_pixel_width = pixel_width;
}
void mh_set_has_denoiser(mh_raytracer* h, int on) { h->rt.has_denoiser = on != 0; }
float* mh_denoiser_image(mh_raytracer* h, int which) { std::vector<float>& v = which == 0 ? h->rt.albedoImage : (which == 1 ? h->rt.normalImage : h->rt.shadingNormalImage); return v.empty() ? nullptr : v.data(); }
void mh_set_envmap_intensity(mh_raytracer* h, float v) { h->rt.s.envmap_intensity = v; }
// decode an image file the way Texture::loadColors' load_image does (stb_image, 3 channels, rows as in the file): for tests
int mh_read_image(const char* file, unsigned char* rgb_out, int capacity, int* W, int* H, char* err, int errlen) {
	std::vector<unsigned char> rgb; std::string why;
	if (!read_image_rgb8(file, rgb, *W, *H, why)) { if (err && errlen > 0) { strncpy(err, why.c_str(), errlen - 1); err[errlen - 1] = 0; } return -1; }
	if ((int)rgb.size() > capacity) return -2;
	memcpy(rgb_out, rgb.data(), rgb.size());
	return 0;
}
int mh_load_scene(mh_raytracer* h, const char* scn) { return h->rt.load_scene(scn) ? 0 : -1; }
void mh_set_frame(mh_raytracer* h, int frame) { h->rt.s.current_frame = frame; }
void mh_add_keyframe(mh_raytracer* h, int obj, int frame) { h->rt.s.objects[obj]->add_keyframe(frame); }
void mh_set_object_transform(mh_raytracer* h, int obj, const float* t, const float* r, float scale) {
	Object* o = h->rt.s.objects[obj];
	o->max_translation = Vector(t[0], t[1], t[2]); memcpy(o->mat_rotation, r, 36); o->scale = scale;
}
int mh_load_scene_subst(mh_raytracer* h, const char* scn, const char* replacedNames) { return h->rt.load_scene(scn, replacedNames) ? 0 : -1; }
int mh_save_image(const char* file, const unsigned char* rgb, int W, int H, char* err, int errlen) {
	std::string why;
	if (write_image_rgb8(file, rgb, W, H, why)) return 0;
	if (err && errlen > 0) { strncpy(err, why.c_str(), errlen - 1); err[errlen - 1] = 0; }
	return -1;
}
// save_image<float> (utils.cpp:177-211): `.hdr` stores the floats as they are (EncodeRadianceHDR); every other container
// gets min(255, max(0, val * (255. / maxval))) truncated to a byte, then the 8-bit writer of that container.
int mh_save_image_f32(const char* file, const float* rgb, int W, int H, float maxval, char* err, int errlen) {
	std::string why;
	bool ok = false;
	if (W <= 0 || H <= 0 || !rgb) why = "empty image";
	else if (image_format_of(file) == F_HDR) {
		std::vector<unsigned char> out;
		mipt_imgwrite::encode_radiance_hdr(rgb, W, H, 3, out);
		ok = write_file(file, out, why);
	} else {
		std::vector<unsigned char> scaled((size_t)W * H * 3);
		for (size_t i = 0; i < scaled.size(); i++) scaled[i] = (unsigned char)std::min(255., std::max(0., rgb[i] * (255. / maxval)));
		ok = write_image_rgb8(file, scaled.data(), W, H, why);
	}
	if (ok) return 0;
	if (err && errlen > 0) { strncpy(err, why.c_str(), errlen - 1); err[errlen - 1] = 0; }
	return -1;
}
// Is there a writer for this output name (8-bit pixels; for_float: float pixels)?  Looks at the name only: nothing is created.
int mh_image_format_supported(const char* file, int for_float) {
	const ImageFormat f = image_format_of(file ? file : "");
	return f != F_NONE && (for_float || f != F_HDR);
}
int mh_save_scene(mh_raytracer* h, const char* scn) { return h->rt.save_scene(scn) ? 0 : -1; }
int mh_num_objects(mh_raytracer* h) { return (int)h->rt.s.objects.size(); }
void mh_get_scene_header(mh_raytracer* h, float* o) {
	const Raytracer& r = h->rt;
	int k = 0;
	o[k++] = (float)r.W; o[k++] = (float)r.H; o[k++] = (float)r.nrays; o[k++] = (float)r.nb_bounces;
	for (int c = 0; c < 3; c++) o[k++] = r.cam.position[c];
	for (int c = 0; c < 3; c++) o[k++] = r.cam.direction[c];
	for (int c = 0; c < 3; c++) o[k++] = r.cam.up[c];
	o[k++] = r.cam.fov; o[k++] = r.cam.focus_distance; o[k++] = r.cam.aperture; o[k++] = r.sigma_filter; o[k++] = r.gamma;
	o[k++] = r.s.intensite_lumiere; o[k++] = r.s.envmap_intensity; o[k++] = r.s.double_frustum_start_t;
	while (k < 32) o[k++] = 0.f;
}
void mh_get_object_state(mh_raytracer* h, int obj, float* o, int* fl) {
	const Object* ob = h->rt.s.objects[obj];
	int k = 0;
	for (int c = 0; c < 3; c++) o[k++] = ob->max_translation[c];
	for (int c = 0; c < 9; c++) o[k++] = ob->mat_rotation[c];
	for (int c = 0; c < 3; c++) o[k++] = ob->rotation_center[c];
	o[k++] = ob->scale;
	for (int c = 0; c < 8; c++) o[16 + c] = 0.f;
	fl[0] = (int)ob->type; fl[1] = ob->miroir; fl[2] = ob->ghost; fl[3] = ob->flip_normals; fl[4] = ob->interp_normals; fl[5] = 0; fl[6] = fl[7] = 0;
	if (ob->type == OT_SPHERE) { const Sphere* sp = static_cast<const Sphere*>(ob); for (int c = 0; c < 3; c++) o[16 + c] = sp->O[c]; o[19] = sp->R; fl[5] = sp->has_envmap; }
	if (ob->type == OT_PLANE) { const Plane* pl = static_cast<const Plane*>(ob); for (int c = 0; c < 3; c++) { o[16 + c] = pl->A[c]; o[19 + c] = pl->vecN[c]; } }
}
int mh_add_mesh_obj(mh_raytracer* h, const char* obj_file, float scale, int center) {
	Raytracer& r = h->rt;
	TriMesh* g = new TriMesh(obj_file, center != 0);
	if (!g->loaded) { r.set_error(g->load_error.empty() ? std::string("no faces in ") + obj_file : g->load_error); delete g; return -1; }
	g->scale = scale;   // GUI placement, mainApp.cpp:2402-2410
	g->max_translation = Vector(0, r.s.objects[2]->max_translation[1] - g->bbox[1] * g->scale, 0);
	r.s.addObject(g);
	return (int)r.s.objects.size() - 1;
}
// multipliers of one material group: Kd, Ks, Ne (3 each), alpha, refr, transp (1 each) + W, H of the Kd / Ks / normal / alpha images
void mh_get_group_material(mh_raytracer* h, int obj, int grp, float* out12, int* wh8) {
	Object* o = h->rt.s.objects[obj];
	// (an object that is not a mesh has only the lists that were given to it: a missing entry reads as zeros)
	static const Texture none = [] { Texture t; t.multiplier = Vector(0, 0, 0); return t; }();
	auto at = [&](const std::vector<Texture>& l) -> const Texture& { return grp >= 0 && grp < (int)l.size() ? l[grp] : none; };
	for (int k = 0; k < 3; k++) { out12[k] = at(o->textures).multiplier[k]; out12[3 + k] = at(o->specularmap).multiplier[k]; out12[6 + k] = at(o->roughnessmap).multiplier[k]; }
	out12[9] = at(o->alphamap).multiplier[0]; out12[10] = at(o->refr_index_map).multiplier[0]; out12[11] = at(o->transparent_map).multiplier[0];
	const Texture* t[4] = {&at(o->textures), &at(o->specularmap), &at(o->normal_map), &at(o->alphamap)};
	for (int k = 0; k < 4; k++) { wh8[2 * k] = (int)t[k]->W; wh8[2 * k + 1] = (int)t[k]->H; }
}
int mh_num_groups(mh_raytracer* h, int obj) { return (int)h->rt.s.objects[obj]->textures.size(); }
const float* mh_group_texture_values(mh_raytracer* h, int obj, int grp, int slot) {   // slot: 0 Kd, 1 Ks, 2 normal, 3 alpha
	Object* o = h->rt.s.objects[obj];
	const Texture* t[4] = {&o->textures[grp], &o->specularmap[grp], &o->normal_map[grp], &o->alphamap[grp]};
	return t[slot]->values.empty() ? nullptr : t[slot]->values.data();
}
int mh_add_mesh(mh_raytracer* h, int nv, const float* verts, int nn, const float* normals, int nt, const float* uvs, int nf, const int* fv, const int* fn, const int* ft, float scale, int center) {
	Raytracer& r = h->rt;
	TriMesh* g = new TriMesh(nv, verts, nn, normals, nt, uvs, nf, fv, fn, ft, center != 0);
	if (!g->loaded) { r.set_error(g->load_error); delete g; return -1; }
	g->scale = scale;   // GUI placement, mainApp.cpp:2402-2410
	g->max_translation = Vector(0, r.s.objects[2]->max_translation[1] - g->bbox[1] * g->scale, 0);
	r.s.addObject(g);
	return (int)r.s.objects.size() - 1;
}
void mh_set_object_flags(mh_raytracer* h, int obj, int miroir, int flip) { h->rt.s.objects[obj]->miroir = miroir != 0; h->rt.s.objects[obj]->flip_normals = flip != 0; }
void mh_add_col_subsurface(mh_raytracer* h, int obj, const float* rgb) { h->rt.s.objects[obj]->add_col_subsurface(Vector(rgb[0], rgb[1], rgb[2])); }
void mh_set_group_subsurface(mh_raytracer* h, int obj, int grp, const float* rgb) {   // Object::subsurface[grp] as a constant colour
	Object* o = h->rt.s.objects[obj];
	if (grp >= 0 && grp < (int)o->subsurface.size()) o->subsurface[grp].multiplier = Vector(rgb[0], rgb[1], rgb[2]);
}
void mh_set_group_material(mh_raytracer* h, int obj, int grp, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr) {
	Object* o = h->rt.s.objects[obj];
	if (grp < (int)o->textures.size()) o->textures[grp].multiplier = Vector(Kd[0], Kd[1], Kd[2]);
	if (grp < (int)o->specularmap.size()) o->specularmap[grp].multiplier = Vector(Ks[0], Ks[1], Ks[2]);
	if (grp < (int)o->roughnessmap.size()) o->roughnessmap[grp].multiplier = Vector(Ne[0], Ne[1], Ne[2]);
	if (grp < (int)o->transparent_map.size()) o->transparent_map[grp].multiplier = Vector(transp_col, transp_col, transp_col);
	if (grp < (int)o->refr_index_map.size()) o->refr_index_map[grp].multiplier = Vector(refr, refr, refr);
}
int mh_add_sphere(mh_raytracer* h, const float* O, float R, int mirror, int flip_normals) {   // s.addObject(new Sphere(O, R, mirror, normal_swapped))
	Sphere* sp = new Sphere(Vector(O[0], O[1], O[2]), R);
	sp->miroir = mirror != 0; sp->flip_normals = flip_normals != 0;
	h->rt.s.addObject(sp);
	return (int)h->rt.s.objects.size() - 1;
}
void mh_add_group_material(mh_raytracer* h, int obj, const float* Kd, const float* Ks, const float* Ne, float transp_col, float refr) {
	Object* o = h->rt.s.objects[obj];
	o->add_col_texture(Vector(Kd[0], Kd[1], Kd[2])); o->add_col_specular(Vector(Ks[0], Ks[1], Ks[2])); o->add_col_roughness(Vector(Ne[0], Ne[1], Ne[2]));
	o->add_col_transp(transp_col); o->add_col_refr(refr);
}
void mh_set_group_texture(mh_raytracer* h, int obj, int grp, int slot, int W, int H, const unsigned char* rgb) {
	Object* o = h->rt.s.objects[obj];
	std::vector<Texture>* lists[8] = {&o->textures, &o->specularmap, &o->normal_map, &o->alphamap, &o->roughnessmap, &o->transparent_map, &o->refr_index_map, &o->subsurface};
	if (slot < 0 || slot > 7 || grp < 0 || grp >= (int)lists[slot]->size()) return;
	Texture& t = (*lists[slot])[grp];
	// set_alphamap / set_roughnessmap build a fresh Texture with multiplier 1 (Geometry.cpp:138-146)
	if (slot == 3 || slot == 4) t.multiplier = Vector(1, 1, 1);
	if (slot == 2) t.loadNormalsRGB8(rgb, W, H); else t.loadColorsRGB8(rgb, W, H);
}
void mh_set_envmap(mh_raytracer* h, int W, int H, const unsigned char* rgb) { static_cast<Sphere*>(h->rt.s.objects[1])->load_envmap_rgb8(rgb, W, H); }
void mh_set_brdf_merl(mh_raytracer* h, int obj, const double* table) { g_content_epoch++; h->rt.s.objects[obj]->merl_data.assign(table, table + (size_t)3 * 90 * 90 * 180); }
// objects[obj]->brdf = new IsoMERLBRDF(file): the MERL ".binary" layout read like read_brdf (MERLBRDFRead.cpp:212-236):
// three int32 dimensions whose product must be 90*90*180 (= BRDF_SAMPLING_RES_THETA_H * _THETA_D * _PHI_D / 2), then
// 3 planes of that many doubles.  Returns 0, or -1 with mh_last_error set (the reference prints and carries on with no BRDF).
int mh_set_brdf_merl_file(mh_raytracer* h, int obj, const char* file) {
	FILE* f = fopen(file, "rb");
	if (!f) { h->rt.set_error(std::string("cannot open ") + file); return -1; }
	int dims[3] = {0, 0, 0};
	const size_t n = (size_t)90 * 90 * 180;
	bool ok = fread(dims, sizeof(int), 3, f) == 3 && (long long)dims[0] * dims[1] * dims[2] == (long long)n;
	std::vector<double> data;
	if (ok) { data.resize(3 * n); ok = fread(data.data(), sizeof(double), 3 * n, f) == 3 * n; }
	fclose(f);
	if (!ok) { h->rt.set_error(std::string("not a MERL .binary file (dimensions / length): ") + file); return -1; }
	h->rt.s.objects[obj]->merl_data.swap(data);
	return 0;
}
const double* mh_merl_data(mh_raytracer* h, int obj) { const auto& d = h->rt.s.objects[obj]->merl_data; return d.empty() ? nullptr : d.data(); }
int mh_prepare(mh_raytracer* h, int upload) {
	Raytracer& r = h->rt;
	r.prepare_render((float)r.s.current_frame);
	if (!upload) return MIPT_OK;
	if (!r.ctx) return MIPT_ERR_NO_DEVICE;
	r.scene_changed();                                  // an explicit prepare always uploads
	r.last_status = r.upload_scene_if_changed();
	return r.last_status;
}
int mh_render_image(mh_raytracer* h) { h->rt.render_image(); return h->rt.last_status; }
void mh_set_progressive_lookahead(mh_raytracer* h, int n) { h->rt.progressive_lookahead = n; }
int mh_render_image_nopreviz(mh_raytracer* h) { h->rt.render_image_nopreviz(); return h->rt.last_status; }
const char* mh_last_error(mh_raytracer* h) { return h->rt.last_error(); }
void* mh_ctx(mh_raytracer* h) { return h->rt.ctx; }
const void* mh_scene_desc(mh_raytracer* h) { return &h->rt.scene_desc; }
const void* mh_render_params(mh_raytracer* h) { return &h->rt.render_params; }
float* mh_imagedouble(mh_raytracer* h) { return h->rt.imagedouble.data(); }
float* mh_sample_count(mh_raytracer* h) { return h->rt.sample_count.data(); }
unsigned char* mh_image(mh_raytracer* h) { return h->rt.image.data(); }

void mh_get_light(mh_raytracer* h, float* o) { Raytracer& r = h->rt; o[0] = r.centerLight[0]; o[1] = r.centerLight[1]; o[2] = r.centerLight[2]; o[3] = r.radiusLight; o[4] = r.lightPower; }
void mh_get_tables(mh_raytracer* h, float* rpp, float* s2d, float* fi, int* fs) {
	Raytracer& r = h->rt;
	if (rpp) for (size_t i = 0; i < r.randomPerPixel.size(); i++) { rpp[2 * i] = r.randomPerPixel[i][0]; rpp[2 * i + 1] = r.randomPerPixel[i][1]; }
	if (s2d) for (size_t i = 0; i < r.samples2d.size(); i++) { s2d[2 * i] = r.samples2d[i][0]; s2d[2 * i + 1] = r.samples2d[i][1]; }
	if (fi) memcpy(fi, r.filter_integral.data(), r.filter_integral.size() * 4);
	if (fs) *fs = r.filter_size;
}
void mh_get_object_matrices(mh_raytracer* h, int obj, float* t, float* inv, float* rot) {
	Object* o = h->rt.s.objects[obj]; memcpy(t, o->trans_matrix, 48); memcpy(inv, o->inv_trans_matrix, 48); memcpy(rot, o->rot_matrix, 36);
}
void mh_mesh_counts(mh_raytracer* h, int obj, int* ntri, int* nnodes, int* nverts, int* nnormals, int* nuvs) {
	TriMesh* g = static_cast<TriMesh*>(h->rt.s.objects[obj]);
	*ntri = (int)g->indices.size(); *nnodes = g->node_count(); *nverts = (int)g->vertices.size(); *nnormals = (int)g->normals.size(); *nuvs = (int)g->uvs.size();
}
int mh_mesh_bvh_builder(mh_raytracer* h, int obj, double* seconds, double* device_seconds) {
	TriMesh* g = static_cast<TriMesh*>(h->rt.s.objects[obj]);
	if (seconds) *seconds = g->bvh_build_seconds;
	if (device_seconds) *device_seconds = g->bvh_device_seconds;
	return g->bvh_builder;
}
int mh_mesh_tangents(mh_raytracer* h, int obj, float* out) {
	TriMesh* g = static_cast<TriMesh*>(h->rt.s.objects[obj]);
	g->sync_tangents();
	if (g->tangentSoup.empty()) return 0;
	const size_t n = g->tangentSoup.size() * 3;
	if (out) memcpy(out, &g->tangentSoup[0][0], n * sizeof(float));
	return (int)n;
}
void mh_mesh_dump(mh_raytracer* h, int obj, int* perm, int* nodes_i, float* nodes_bb, float* soup, int* groups, float* root_bb) {
	TriMesh* g = static_cast<TriMesh*>(h->rt.s.objects[obj]);
	g->sync_host();                       // (a mesh built on the device: this is where its reference-layout views are fetched)
	const int nt = (int)g->indices.size();
	for (int i = 0; i < nt; i++) {
		if (perm) perm[i] = g->permuted_triangle_index[i];
		if (groups) groups[i] = g->indices[i].group;
		if (soup) memcpy(soup + (size_t)i * 31, &g->triangleSoup[i], 124);
	}
	for (size_t i = 0; i < g->bvh.nodes.size(); i++) {
		if (nodes_i) { nodes_i[3 * i] = g->bvh.nodes[i].isleaf ? 1 : 0; nodes_i[3 * i + 1] = g->bvh.nodes[i].fg; nodes_i[3 * i + 2] = g->bvh.nodes[i].fd; }
		if (nodes_bb) memcpy(nodes_bb + 6 * i, g->bvh.nodes[i].bbox, 24);
	}
	if (root_bb) memcpy(root_bb, g->bvh.bbox, 24);
}
}
