// mipt_mesh_device.h — mipt_build_bvh and mipt_device_mesh_*: host side of the BVH construction on the device and of meshes that stay there (kernels: mipt_build.h)
// (part of the one translation unit csrc/mipt.hip: included there, after the context and the render loop it uses)

#pragma once
// =====================================================================================
// BVH construction on the device (mipt_build.h): same nodes, same positions, same triangle order as
// TriMesh::build_bvh (TriangleMesh.cpp:878-885, 1029-1130)
// =====================================================================================
static thread_local std::string g_build_err;
extern "C" const char* mipt_build_bvh_error(void) { return g_build_err.c_str(); }

namespace {
struct DevPool {     // device allocations of one build, released on every exit path
	std::vector<void*> p;
	~DevPool() { release(); }
	void release() { for (void* q : p) if (q) hipFree(q); p.clear(); }
	template <class T> bool get(T** out, size_t count) {
		void* q = nullptr;
		if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
		p.push_back(q); *out = (T*)q; return true;
	}
};
int build_fail(int code, const char* fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	g_build_err = buf;
	return code;
}
}
#define BHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return build_fail(MIPT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

namespace {
// what a finished build leaves on the device (owned by `pool` until the caller detaches it)
struct BuiltTree {
	DevPool pool;
	bvhb::ONode* d_out = nullptr;      // the reference's node vector (depth-first)
	uint32_t* d_perm = nullptr;        // position i of the reordered mesh holds input triangle d_perm[i]
	float* d_vtx = nullptr;            // the vertices as uploaded
	char* d_tri = nullptr;             // the caller's triangle records as uploaded (tri_stride_bytes apart)
	int total = 0;                     // nodes
	double device_seconds = 0;
};
}
// The build itself (the level-synchronous phase, the small subtrees, numbering and emission): everything stays on the device.
static int bvh_build_core(int device_id, const float* vertices, int nverts, const void* tri_vtx, int tri_stride_bytes, int ntri, size_t tri_upload_bytes, BuiltTree& bt) {
	using namespace bvhb;
	static_assert(sizeof(ONode) == sizeof(mipt_bvh_node), "node layout");
	g_build_err.clear();
	const bool trace = getenv("MIPT_BUILD_TRACE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto phase = [&](const char* what) {
		if (!trace) return;
		hipDeviceSynchronize();
		const auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[mipt_build_bvh] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
		t_last = t;
	};
	if (!vertices || nverts <= 0 || !tri_vtx || tri_stride_bytes < 12 || (tri_stride_bytes & 3) || ntri <= 0) return build_fail(MIPT_ERR_INVALID, "bad arguments");
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return build_fail(MIPT_ERR_NO_DEVICE, "no usable HIP device");
	BHIP(hipSetDevice(device_id));
	const int n = ntri;
	hipEvent_t e0, e1;
	BHIP(hipEventCreate(&e0)); BHIP(hipEventCreate(&e1));
	struct EvGuard { hipEvent_t a, b; ~EvGuard() { hipEventDestroy(a); hipEventDestroy(b); } } evg{e0, e1};
	DevPool& pool = bt.pool;
	const int maxseg = n / (BVHB_SMALL + 1) + 2;
	float* d_vtx; char* d_tv; float4* d_rec; int* d_bad; uint32_t *d_order[2], *d_S, *d_tpos, *d_acc, *d_bins, *d_bsum; int* d_segof[2]; uint8_t* d_pf;
	Seg* d_segs[2]; float* d_planes; LNode* d_ln; int* d_smalls; Counters* d_cnt; ONode *d_sn, *d_out;
	const int nscanblk = (n + BVHB_SCAN_TILE - 1) / BVHB_SCAN_TILE;
	bool ok = pool.get(&d_vtx, (size_t)nverts * 3) && pool.get(&d_tv, std::max((size_t)n * tri_stride_bytes, tri_upload_bytes)) && pool.get(&d_bad, 1) && pool.get(&d_rec, (size_t)n * 3)
	       && pool.get(&d_order[0], n) && pool.get(&d_order[1], n) && pool.get(&d_segof[0], n) && pool.get(&d_segof[1], n)
	       && pool.get(&d_S, (size_t)n + 1) && pool.get(&d_tpos, n) && pool.get(&d_pf, n) && pool.get(&d_bsum, nscanblk)
	       && pool.get(&d_segs[0], maxseg) && pool.get(&d_segs[1], maxseg) && pool.get(&d_acc, (size_t)maxseg * 12)
	       && pool.get(&d_bins, (size_t)maxseg * BVHB_BINWORDS) && pool.get(&d_planes, (size_t)maxseg * (BVHB_NPLANES + 2))
	       && pool.get(&d_ln, (size_t)2 * n + 2) && pool.get(&d_smalls, (size_t)n + 1) && pool.get(&d_cnt, 1)
	       && pool.get(&d_sn, (size_t)2 * n + 2) && pool.get(&d_out, (size_t)2 * n + 2);
	if (!ok) return build_fail(MIPT_ERR_HIP, "hipMalloc failed for a %d-triangle build", n);
	phase("hipMalloc");
	BHIP(hipMemcpy(d_vtx, vertices, (size_t)nverts * 12, hipMemcpyHostToDevice));
	BHIP(hipMemcpy(d_tv, tri_vtx, tri_upload_bytes ? tri_upload_bytes : (size_t)(n - 1) * tri_stride_bytes + 12, hipMemcpyHostToDevice));   // the records as they are (TriangleIndices: 44 bytes apart)
	BHIP(hipMemsetAsync(d_bad, 0, 4, 0));
	phase("upload");
	BHIP(hipEventRecord(e0, 0));
	const bool large_root = n > BVHB_SMALL;
	const unsigned pos_blocks = (unsigned)((n + 255) / 256);
	hipLaunchKernelGGL(k_prepare, dim3(pos_blocks), dim3(256), 0, 0, d_vtx, nverts, d_tv, tri_stride_bytes, n, d_rec, d_order[0], d_segof[0], large_root ? 0 : -1, d_bad);
	{
		LNode root; memset(&root, 0, sizeof root);
		root.i0 = 0; root.i1 = n; root.left = root.right = -1; root.kind = large_root ? K_PENDING : K_SMALL;
		BHIP(hipMemcpyAsync(d_ln, &root, sizeof root, hipMemcpyHostToDevice, 0));
		Seg s0; memset(&s0, 0, sizeof s0); s0.node = 0; s0.i0 = 0; s0.i1 = n;
		BHIP(hipMemcpyAsync(d_segs[0], &s0, sizeof s0, hipMemcpyHostToDevice, 0));
		Counters c0 = {1, 0, large_root ? 0 : 1, 0};
		BHIP(hipMemcpyAsync(d_cnt, &c0, sizeof c0, hipMemcpyHostToDevice, 0));
		if (!large_root) { int z = 0; BHIP(hipMemcpyAsync(d_smalls, &z, 4, hipMemcpyHostToDevice, 0)); }
		int bad = 0;
		BHIP(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));   // also: the staging variables go out of scope
		if (bad) return build_fail(MIPT_ERR_INVALID, "triangle vertex index out of range");
	}
	// level-synchronous phase
	std::vector<int> level_begin{0};       // LNode id ranges per level
	int nseg = large_root ? 1 : 0, ln_count = 1, nsmall = large_root ? 0 : 1, cur = 0, levels = 0;
	const int gpw = std::max(1, std::min(64, n / (64 * 8192)));
	const unsigned wave_blocks = (unsigned)(((long long)n + 64LL * gpw - 1) / (64LL * gpw));
	while (nseg > 0) {
		if (++levels > 100000) return build_fail(MIPT_ERR_UNSUPPORTED, "degenerate mesh: more than 100000 BVH levels");
		const unsigned seg_blocks = (unsigned)((nseg + 127) / 128);
		const size_t init_n = (size_t)nseg * BVHB_BINWORDS;
		Seg* segs = d_segs[cur]; Seg* next = d_segs[cur ^ 1];
		hipLaunchKernelGGL(k_lvl_init, dim3((unsigned)((init_n + 255) / 256)), dim3(256), 0, 0, nseg, d_acc, d_bins);
		hipLaunchKernelGGL(k_lvl_bounds, dim3(wave_blocks), dim3(64), 0, 0, d_rec, d_order[cur], d_segof[cur], n, gpw, d_acc);
		hipLaunchKernelGGL(k_lvl_planes, dim3(seg_blocks), dim3(128), 0, 0, nseg, segs, d_acc, d_ln, d_planes);
		hipLaunchKernelGGL(k_lvl_bin, dim3(wave_blocks), dim3(64), 0, 0, d_rec, d_order[cur], d_segof[cur], n, gpw, segs, d_planes, d_bins);
		hipLaunchKernelGGL(k_lvl_choose, dim3(seg_blocks), dim3(128), 0, 0, nseg, segs, d_bins, d_planes);
		hipLaunchKernelGGL(k_scan_sums, dim3(nscanblk), dim3(256), 0, 0, d_rec, d_order[cur], d_segof[cur], n, segs, d_pf, d_bsum);
		hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, 0, d_bsum, nscanblk);
		hipLaunchKernelGGL(k_scan_apply, dim3(nscanblk), dim3(256), 0, 0, d_pf, n, d_bsum, d_S);
		hipLaunchKernelGGL(k_lvl_scatter_true, dim3(pos_blocks), dim3(256), 0, 0, d_order[cur], d_segof[cur], n, segs, d_pf, d_S, d_order[cur ^ 1], d_tpos);
		hipLaunchKernelGGL(k_lvl_scatter_false, dim3(pos_blocks), dim3(256), 0, 0, d_order[cur], d_segof[cur], n, segs, d_pf, d_S, d_order[cur ^ 1], d_tpos, &d_cnt->unresolved);
		hipLaunchKernelGGL(k_lvl_children, dim3(seg_blocks), dim3(128), 0, 0, nseg, segs, d_S, d_ln, next, d_smalls, d_cnt);
		hipLaunchKernelGGL(k_lvl_resegment, dim3(pos_blocks), dim3(256), 0, 0, d_segof[cur], n, segs, d_segof[cur ^ 1]);
		Counters h;
		BHIP(hipMemcpy(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost));
		for (int round = 0; h.unresolved; round++) {   // long jump chains: double the jump table and walk again
			if (round > 40) return build_fail(MIPT_ERR_HIP, "partition did not converge");
			hipLaunchKernelGGL(k_lvl_double, dim3(pos_blocks), dim3(256), 0, 0, d_segof[cur], n, segs, d_S, d_tpos);
			BHIP(hipMemsetAsync(&d_cnt->unresolved, 0, 4, 0));
			hipLaunchKernelGGL(k_lvl_scatter_false, dim3(pos_blocks), dim3(256), 0, 0, d_order[cur], d_segof[cur], n, segs, d_pf, d_S, d_order[cur ^ 1], d_tpos, &d_cnt->unresolved);
			BHIP(hipMemcpy(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost));
		}
		level_begin.push_back(ln_count);
		ln_count = h.ln_count; nsmall = h.nsmall; nseg = h.nseg_next;
		if (nseg > maxseg) return build_fail(MIPT_ERR_HIP, "segment list overflow");
		BHIP(hipMemsetAsync(&d_cnt->nseg_next, 0, 4, 0));
		cur ^= 1;
	}
	level_begin.push_back(ln_count);
	phase("levels");
	if (nsmall > 0) hipLaunchKernelGGL(k_small_subtrees, dim3((unsigned)((nsmall + 63) / 64)), dim3(64), 0, 0, nsmall, d_smalls, d_ln, d_rec, d_order[cur], d_sn);
	phase("small subtrees");
	const int nlev = (int)level_begin.size() - 1;
	for (int L = nlev - 1; L >= 0; L--) {
		const int b = level_begin[L], e = level_begin[L + 1];
		if (e > b) hipLaunchKernelGGL(k_sizes, dim3((unsigned)((e - b + 255) / 256)), dim3(256), 0, 0, d_ln, b, e);
	}
	for (int L = 0; L < nlev; L++) {
		const int b = level_begin[L], e = level_begin[L + 1];
		if (e > b) hipLaunchKernelGGL(k_preorder, dim3((unsigned)((e - b + 255) / 256)), dim3(256), 0, 0, d_ln, b, e);
	}
	hipLaunchKernelGGL(k_emit, dim3((unsigned)((ln_count + 127) / 128)), dim3(128), 0, 0, d_ln, ln_count, d_sn, d_out);
	BHIP(hipEventRecord(e1, 0));
	LNode root;
	BHIP(hipMemcpy(&root, d_ln, sizeof root, hipMemcpyDeviceToHost));
	BHIP(hipGetLastError());
	const int total = root.size;
	if (total <= 0 || total > 2 * n) return build_fail(MIPT_ERR_HIP, "inconsistent node count %d", total);
	phase("numbering");
	bt.d_out = d_out; bt.d_perm = d_order[cur]; bt.d_vtx = d_vtx; bt.d_tri = d_tv; bt.total = total;
	{ float ms = 0.f; hipEventElapsedTime(&ms, e0, e1); bt.device_seconds = ms * 1e-3; }
	return MIPT_OK;
}

extern "C" int mipt_build_bvh(int device_id, const float* vertices, int nverts, const void* tri_vtx, int tri_stride_bytes, int ntri,
                              mipt_bvh_node* out_nodes, int node_capacity, int* out_n_nodes, int32_t* out_perm, double* out_seconds) {
	if (!out_nodes || !out_n_nodes || !out_perm) { g_build_err = "bad arguments"; return MIPT_ERR_INVALID; }
	BuiltTree bt;
	const int rc = bvh_build_core(device_id, vertices, nverts, tri_vtx, tri_stride_bytes, ntri, 0, bt);
	if (rc) return rc;
	if (bt.total > node_capacity) return build_fail(MIPT_ERR_INVALID, "node_capacity %d is too small for %d nodes", node_capacity, bt.total);
	BHIP(hipMemcpy(out_nodes, bt.d_out, (size_t)bt.total * sizeof(bvhb::ONode), hipMemcpyDeviceToHost));
	BHIP(hipMemcpy(out_perm, bt.d_perm, (size_t)ntri * 4, hipMemcpyDeviceToHost));
	*out_n_nodes = bt.total;
	if (out_seconds) *out_seconds = bt.device_seconds;
	return MIPT_OK;
}

// =====================================================================================
// A mesh whose tree AND traversal records are made on the device and stay there (round 4)
// =====================================================================================
// TriMesh::init used to fetch the tree back (58 MB of nodes + the permutation at 2.5 M triangles), build 310 MB of Triangle records on
// the host, and mipt_upload_scene re-packed both into fat nodes and 64-byte records and sent 420 MB up again.  Here the device that built
// the tree derives the traversal's records from it; mipt_upload_scene adopts them with device-to-device copies (mipt_mesh::device_mesh);
// the reference-layout views (bvh.nodes, the permutation) are downloaded only when somebody asks (mipt_device_mesh_download).
struct mipt_device_mesh {
	uint32_t magic = 0x4d444d31u;      // 'MDM1'
	int device = 0;
	int ntri = 0, nnodes = 0, nfat = 0, nuvs = 0;
	uint32_t root_ref = 0;
	bvhb::ONode* d_nodes = nullptr; uint32_t* d_perm = nullptr;
	DFatNode* d_fat = nullptr; DTriIsect* d_ti = nullptr; DTriShade* d_ts = nullptr; int* d_uvidx = nullptr;
	float* d_tangent = nullptr;        // TriMesh::tangentSoup (9 floats per triangle) when the mesh has UVs and normals
	mutable std::atomic<int> refs{1};  // the creator's reference + one per scene that uses the buffers in place (mipt_upload_scene on the same device)
};

extern "C" void mipt_device_mesh_free(mipt_device_mesh* m) {
	if (!m || m->magic != 0x4d444d31u) return;
	if (m->refs.fetch_sub(1) > 1) return;          // a scene still renders from these buffers: they go when it is replaced or its context destroyed
	int prev = 0; hipGetDevice(&prev);
	hipSetDevice(m->device);
	for (void* p : {(void*)m->d_nodes, (void*)m->d_perm, (void*)m->d_fat, (void*)m->d_ti, (void*)m->d_ts, (void*)m->d_uvidx, (void*)m->d_tangent}) if (p) hipFree(p);
	hipSetDevice(prev);
	m->magic = 0;
	delete m;
}

extern "C" int mipt_device_mesh_build(int device_id, const float* vertices, int nverts, const float* normals, int nnormals, const float* uvs, int nuvs,
                                      const mipt_triangle_indices* indices, int ntri, mipt_device_mesh** out, mipt_device_mesh_info* info) {
	using namespace bvhb;
	if (!out || !indices || (nnormals > 0 && !normals) || (nuvs > 0 && !uvs) || nnormals < 0 || nuvs < 0) { g_build_err = "bad arguments"; return MIPT_ERR_INVALID; }
	*out = nullptr;
	if ((unsigned)ntri > MIPT_LEAF_FIRST_MASK) return build_fail(MIPT_ERR_UNSUPPORTED, "mesh has more than 2^26 triangles");
	const bool trace = getenv("MIPT_BUILD_TRACE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto phase = [&](const char* what) {
		if (!trace) return;
		hipDeviceSynchronize();
		const auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[mipt_device_mesh_build] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
		t_last = t;
	};
	BuiltTree bt;
	int rc = bvh_build_core(device_id, vertices, nverts, &indices[0].vtxi, (int)sizeof(mipt_triangle_indices), ntri, (size_t)ntri * sizeof(mipt_triangle_indices), bt);
	if (rc) return rc;
	phase("tree");
	const int total = bt.total;
	hipEvent_t e0, e1;
	BHIP(hipEventCreate(&e0)); BHIP(hipEventCreate(&e1));
	struct EvGuard { hipEvent_t a, b; ~EvGuard() { hipEventDestroy(a); hipEventDestroy(b); } } evg{e0, e1};
	// (until the end the device buffers belong to the pools `tmp` / `keep` / `bt.pool`; the handle's pointers are only views: a failure path deletes the plain struct)
	std::unique_ptr<mipt_device_mesh> m(new mipt_device_mesh);
	m->device = device_id; m->ntri = ntri; m->nnodes = total; m->nuvs = nuvs;
	DevPool tmp;                        // scratch of this stage
	DevPool keep;                       // what the handle will own
	float *d_normals = nullptr, *d_uvs = nullptr; uint32_t *d_irank = nullptr, *d_bsum = nullptr, *d_root = nullptr; uint8_t* d_depth = nullptr; int* d_bad = nullptr;
	const int nrb = (total + BVHB_RANK_TILE - 1) / BVHB_RANK_TILE;
	bool ok = tmp.get(&d_irank, (size_t)total) && tmp.get(&d_bsum, (size_t)nrb) && tmp.get(&d_depth, (size_t)total) && tmp.get(&d_bad, 4) && tmp.get(&d_root, 1)
	       && (nnormals == 0 || tmp.get(&d_normals, (size_t)nnormals * 3)) && (nuvs == 0 || tmp.get(&d_uvs, (size_t)nuvs * 3));
	if (!ok) return build_fail(MIPT_ERR_HIP, "hipMalloc failed for the records of a %d-triangle mesh", ntri);
	if (nnormals) BHIP(hipMemcpyAsync(d_normals, normals, (size_t)nnormals * 12, hipMemcpyHostToDevice, 0));
	if (nuvs) BHIP(hipMemcpyAsync(d_uvs, uvs, (size_t)nuvs * 12, hipMemcpyHostToDevice, 0));
	BHIP(hipMemsetAsync(d_bad, 0, 16, 0));
	phase("normals / uvs upload");
	BHIP(hipEventRecord(e0, 0));
	hipLaunchKernelGGL(k_rank_sums, dim3((unsigned)nrb), dim3(256), 0, 0, bt.d_out, total, d_bsum);
	hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, 0, d_bsum, nrb);
	hipLaunchKernelGGL(k_rank_apply, dim3((unsigned)nrb), dim3(256), 0, 0, bt.d_out, total, d_bsum, d_irank);
	const unsigned nb = (unsigned)((total + 255) / 256);
	hipLaunchKernelGGL(k_depth_init, dim3(nb), dim3(256), 0, 0, d_depth, total);
	for (int level = 1; level <= MIPT_STACK_DEPTH; level++) hipLaunchKernelGGL(k_depth_pass, dim3(nb), dim3(256), 0, 0, bt.d_out, total, d_depth, level);
	// the number of inner nodes = rank of a virtual node behind the last one: read back with the flags below
	uint32_t last_rank = 0; ONode last_node;
	BHIP(hipMemcpy(&last_rank, d_irank + (total - 1), 4, hipMemcpyDeviceToHost));
	BHIP(hipMemcpy(&last_node, bt.d_out + (total - 1), sizeof last_node, hipMemcpyDeviceToHost));
	const int nfat = (int)last_rank + (last_node.isleaf ? 0 : 1);
	m->nfat = nfat;
	phase("ranks + depths");
	ok = keep.get(&m->d_fat, (size_t)std::max(nfat, 1)) && keep.get(&m->d_ti, (size_t)ntri) && keep.get(&m->d_ts, (size_t)ntri) && (nuvs == 0 || keep.get(&m->d_uvidx, (size_t)ntri * 3));
	if (!ok) return build_fail(MIPT_ERR_HIP, "hipMalloc failed for the records of a %d-triangle mesh", ntri);
	BHIP(hipMemsetAsync(m->d_fat, 0, sizeof(DFatNode), 0));
	hipLaunchKernelGGL(k_fat_nodes, dim3(nb), dim3(256), 0, 0, bt.d_out, total, ntri, d_irank, d_depth, (int)MIPT_STACK_DEPTH, m->d_fat, d_bad, d_root);
	hipLaunchKernelGGL(k_tri_records, dim3((unsigned)((ntri + 255) / 256)), dim3(256), 0, 0, bt.d_vtx, d_normals, nnormals, d_uvs, nuvs, bt.d_tri, (int)sizeof(mipt_triangle_indices),
	                   bt.d_perm, ntri, m->d_ti, m->d_ts, m->d_uvidx, d_bad);
	phase("fat nodes + records");
	if (nuvs > 0 && nnormals > 0) {      // setup_tangents (the host runs it whenever the mesh has UVs; it reads the vertex normals)
		float *d_sdir = nullptr, *d_tan = nullptr; uint8_t* d_has = nullptr; uint32_t *d_first = nullptr, *d_fill = nullptr, *d_corner = nullptr, *d_vsum = nullptr;
		const int nvb = (nverts + BVHB_RANK_TILE - 1) / BVHB_RANK_TILE;
		ok = tmp.get(&d_sdir, (size_t)ntri * 3) && tmp.get(&d_has, (size_t)ntri) && tmp.get(&d_first, (size_t)nverts + 1) && tmp.get(&d_fill, (size_t)nverts + 1)
		  && tmp.get(&d_corner, (size_t)ntri * 3) && tmp.get(&d_tan, (size_t)nverts * 3) && tmp.get(&d_vsum, (size_t)nvb) && keep.get(&m->d_tangent, (size_t)ntri * 9);
		if (!ok) return build_fail(MIPT_ERR_HIP, "hipMalloc failed for the tangents of a %d-triangle mesh", ntri);
		const unsigned tb = (unsigned)((ntri + 255) / 256);
		BHIP(hipMemsetAsync(d_first, 0, ((size_t)nverts + 1) * 4, 0));
		hipLaunchKernelGGL(k_tan_face, dim3(tb), dim3(256), 0, 0, bt.d_vtx, d_uvs, nuvs, bt.d_tri, (int)sizeof(mipt_triangle_indices), bt.d_perm, ntri, d_sdir, d_has, d_first);
		hipLaunchKernelGGL(k_u32_sums, dim3((unsigned)nvb), dim3(256), 0, 0, d_first, (size_t)nverts, d_vsum);
		hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, 0, d_vsum, nvb);
		hipLaunchKernelGGL(k_u32_apply, dim3((unsigned)nvb), dim3(256), 0, 0, d_first, (size_t)nverts, d_vsum);
		BHIP(hipMemcpyAsync(d_fill, d_first, (size_t)nverts * 4, hipMemcpyDeviceToDevice, 0));
		hipLaunchKernelGGL(k_tan_fill, dim3(tb), dim3(256), 0, 0, bt.d_tri, (int)sizeof(mipt_triangle_indices), bt.d_perm, ntri, d_fill, d_corner);
		hipLaunchKernelGGL(k_tan_vertex, dim3((unsigned)((nverts + 255) / 256)), dim3(256), 0, 0, d_normals, nnormals, bt.d_tri, (int)sizeof(mipt_triangle_indices), bt.d_perm,
		                   d_first, d_fill, d_corner, d_sdir, d_has, nverts, d_tan);
		hipLaunchKernelGGL(k_tan_soup, dim3(tb), dim3(256), 0, 0, bt.d_tri, (int)sizeof(mipt_triangle_indices), bt.d_perm, ntri, d_tan, m->d_tangent);
	}
	phase("tangents");
	BHIP(hipEventRecord(e1, 0));
	int bad[4] = {0, 0, 0, 0};
	BHIP(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
	BHIP(hipMemcpy(&m->root_ref, d_root, 4, hipMemcpyDeviceToHost));
	BHIP(hipGetLastError());
	auto drop = [&]() { m->d_fat = nullptr; m->d_ti = nullptr; m->d_ts = nullptr; m->d_uvidx = nullptr; m->d_tangent = nullptr; };     // (still owned by `keep`)
	if (bad[0]) { drop(); return build_fail(MIPT_ERR_INVALID, "BVH child index out of order / leaf range out of bounds"); }
	if (bad[1]) { drop(); return build_fail(MIPT_ERR_UNSUPPORTED, "BVH leaf with %d triangles: meshes with leaves of %d or more triangles are not kept device-resident (upload them through mipt_mesh::nodes)", bad[1], MIPT_LEAF_MAX_TRIS); }
	if (bad[2]) { drop(); return build_fail(MIPT_ERR_UNSUPPORTED, "BVH with more than %d levels of inner nodes: the traversal stack holds %d", MIPT_STACK_DEPTH, MIPT_STACK_DEPTH); }
	if (bad[3]) { drop(); return build_fail(MIPT_ERR_UNSUPPORTED, "material group index above 2^30"); }
	// the handle keeps the reference-layout nodes and the permutation for mipt_device_mesh_download; everything else of the build goes
	auto detach = [](DevPool& p, void* q) { for (auto& x : p.p) if (x == q) { x = nullptr; return; } };
	m->d_nodes = bt.d_out; m->d_perm = bt.d_perm;
	detach(bt.pool, bt.d_out); detach(bt.pool, bt.d_perm);
	for (void* q : {(void*)m->d_fat, (void*)m->d_ti, (void*)m->d_ts, (void*)m->d_uvidx, (void*)m->d_tangent}) if (q) detach(keep, q);
	if (info) {
		float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
		info->n_triangles = ntri; info->n_nodes = total; info->n_inner = nfat; info->device_id = device_id;
		info->build_seconds = bt.device_seconds; info->records_seconds = ms * 1e-3; info->has_tangents = m->d_tangent ? 1 : 0;
	}
	*out = m.release();
	tmp.release(); keep.release(); bt.pool.release();
	phase("hipFree");
	return MIPT_OK;
}

extern "C" int mipt_device_mesh_download(const mipt_device_mesh* m, mipt_bvh_node* nodes, int node_capacity, int32_t* perm) {
	if (!m || m->magic != 0x4d444d31u) { g_build_err = "not a device mesh"; return MIPT_ERR_INVALID; }
	int prev = 0; hipGetDevice(&prev);
	BHIP(hipSetDevice(m->device));
	if (nodes) {
		if (node_capacity < m->nnodes) return build_fail(MIPT_ERR_INVALID, "node_capacity %d is too small for %d nodes", node_capacity, m->nnodes);
		BHIP(hipMemcpy(nodes, m->d_nodes, (size_t)m->nnodes * sizeof(bvhb::ONode), hipMemcpyDeviceToHost));
	}
	if (perm) BHIP(hipMemcpy(perm, m->d_perm, (size_t)m->ntri * 4, hipMemcpyDeviceToHost));
	hipSetDevice(prev);
	return MIPT_OK;
}
extern "C" int mipt_device_mesh_download_tangents(const mipt_device_mesh* m, float* tangent_soup) {
	if (!m || m->magic != 0x4d444d31u || !tangent_soup) { g_build_err = "not a device mesh"; return MIPT_ERR_INVALID; }
	if (!m->d_tangent) return build_fail(MIPT_ERR_INVALID, "the device mesh has no tangents (a mesh without UVs or without normals)");
	int prev = 0; hipGetDevice(&prev);
	BHIP(hipSetDevice(m->device));
	BHIP(hipMemcpy(tangent_soup, m->d_tangent, (size_t)m->ntri * 9 * sizeof(float), hipMemcpyDeviceToHost));
	hipSetDevice(prev);
	return MIPT_OK;
}

// mipt_upload_scene on a mesh that names a device handle: no host arrays are read; the scene's buffers are filled by device-to-device
// copies (peer copies when the handle lives on another device of a group), child references moved to the mesh's place in the scene.
static int adopt_device_mesh(mipt_ctx* c, const mipt_mesh* m, DObject& d, MeshStaging& stg) {
	const mipt_device_mesh* dm = m->device_mesh;
	if (dm->magic != 0x4d444d31u) return fail(c, MIPT_ERR_INVALID, "mipt_mesh::device_mesh is not a handle of mipt_device_mesh_build");
	if (m->n_triangles != dm->ntri || m->n_nodes != dm->nnodes) return fail(c, MIPT_ERR_INVALID, "mipt_mesh counts (%d triangles, %d nodes) are not the device mesh's (%d, %d)", m->n_triangles, m->n_nodes, dm->ntri, dm->nnodes);
	if (stg.nt_total + (size_t)dm->ntri > MIPT_LEAF_FIRST_MASK) return fail(c, MIPT_ERR_UNSUPPORTED, "scene has more than 2^26 triangles");
	MeshChunk chunk;
	chunk.dev = dm; chunk.nfat = (size_t)std::max(dm->nfat, 1); chunk.nt = (size_t)dm->ntri;
	chunk.node_base = (uint32_t)stg.nfat_total; chunk.tri_base = (uint32_t)stg.nt_total;
	d.node_base = chunk.node_base; d.tri_base = chunk.tri_base;
	d.root_ref = (dm->root_ref & MIPT_LEAF_BIT) ? dm->root_ref + chunk.tri_base : chunk.node_base;        // (an inner root is inner node 0 of the mesh)
	memcpy(d.root_min, m->bvh_bbox_min, 12); memcpy(d.root_max, m->bvh_bbox_max, 12);
	stg.nfat_total += chunk.nfat; stg.nt_total += chunk.nt;
	stg.chunks.push_back(std::move(chunk));
	d.ntri = dm->ntri;
	const bool has_uv = m->n_uvs > 0 && m->uvs && dm->nuvs == m->n_uvs;
	d.nuvs = has_uv ? m->n_uvs : 0;
	d.uvs = nullptr; d.uvidx = nullptr; d.tangent_soup = nullptr;
	if (m->n_uvs > 0 && m->uvs && dm->nuvs != m->n_uvs) return fail(c, MIPT_ERR_INVALID, "mipt_mesh::n_uvs is not the device mesh's");
	if (has_uv) {
		int rc;
		if ((rc = upload(c, m->uvs, (size_t)m->n_uvs * 3, &d.uvs))) return rc;
		const bool here = dm->device == c->device && !c->opt_device_mesh_as_remote;
		if (here) { dm->refs.fetch_add(1); c->scene_shared.push_back(dm); }          // the index triples and the tangents are read in place
		if (here) d.uvidx = dm->d_uvidx;
		else {
			void* p = nullptr;
			HIPCHK(c, hipMalloc(&p, (size_t)dm->ntri * 3 * sizeof(int)));
			c->scene_allocs.push_back(p);
			HIPCHK(c, hipMemcpyPeer(p, c->device, dm->d_uvidx, dm->device, (size_t)dm->ntri * 3 * sizeof(int)));
			d.uvidx = (const int*)p;
		}
		if (dm->d_tangent && here) d.tangent_soup = dm->d_tangent;      // setup_tangents ran on the device: 36 bytes per triangle that never cross PCIe
		else if (dm->d_tangent) {
			void* q = nullptr;
			HIPCHK(c, hipMalloc(&q, (size_t)dm->ntri * 9 * sizeof(float)));
			c->scene_allocs.push_back(q);
			HIPCHK(c, hipMemcpyPeer(q, c->device, dm->d_tangent, dm->device, (size_t)dm->ntri * 9 * sizeof(float)));
			d.tangent_soup = (const float*)q;
		} else if (m->tangentSoup && (rc = upload(c, m->tangentSoup, (size_t)dm->ntri * 9, &d.tangent_soup))) return rc;
	}
	return MIPT_OK;
}
static bool device_mesh_on(const mipt_device_mesh* dm, const mipt_ctx* c) { return dm->device == c->device && !c->opt_device_mesh_as_remote; }
static int share_device_chunk(mipt_ctx* c, const mipt_device_mesh* dm, const DFatNode** dn, const DTriIsect** dt, const DTriShade** dsh) {
	dm->refs.fetch_add(1);
	c->scene_shared.push_back(dm);
	*dn = dm->d_fat; *dt = dm->d_ti; *dsh = dm->d_ts;
	return MIPT_OK;
}
static int copy_device_chunk(mipt_ctx* c, const mipt_device_mesh* dm, DFatNode* dn, DTriIsect* dt, DTriShade* dsh, uint32_t node_base, uint32_t tri_base) {
	const size_t nfat = (size_t)std::max(dm->nfat, 1), nt = (size_t)dm->ntri;
	const bool same = dm->device == c->device && !c->opt_device_mesh_as_remote;
	auto copy = [&](void* dst, const void* src, size_t bytes) { return same ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, c->device, src, dm->device, bytes); };
	HIPCHK(c, copy(dt, dm->d_ti, nt * sizeof(DTriIsect)));
	HIPCHK(c, copy(dsh, dm->d_ts, nt * sizeof(DTriShade)));
	HIPCHK(c, copy(dn, dm->d_fat, nfat * sizeof(DFatNode)));
	if (node_base || tri_base) {       // not the scene's first mesh: the copy above is rewritten in place with scene-wide references
		hipLaunchKernelGGL(bvhb::k_rebase_nodes, dim3((unsigned)((nfat + 255) / 256)), dim3(256), 0, 0, dn, nfat, node_base, tri_base);
		HIPCHK(c, hipGetLastError());
		HIPCHK(c, hipDeviceSynchronize());
	}
	return MIPT_OK;
}

