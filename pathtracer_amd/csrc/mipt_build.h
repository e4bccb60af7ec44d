// mipt_build.h — construction of the reference's BVH on the GPU.
//
// The tree is the one TriMesh::build_bvh / build_bvh_recur make (TriangleMesh.cpp:878-885, 1029-1130):
//   node box = box of the vertices of triangles [i0,i1); split axis = longest axis of the box of the centroids
//   ((A+B)+C)/3.f; 16 candidate planes at cb + diag*(t+1)/17; cost = area(L)*n_L + area(R)*n_R with a triangle on the
//   left when centroid <= plane; first strict minimum wins; in-place partition; leaf when one side is empty or
//   i1 <= i0+4; nodes pushed in preorder.
// Everything except the partition is a min/max/count reduction, so it does not depend on the order triangles are
// visited in and maps onto level-synchronous data-parallel passes.  The partition is order dependent:
//     pivot = i0-1; for i in [i0,i1): if pred(i) { pivot++; swap(a[i], a[pivot]); }
// keeps the pred-true elements in their order, but the pred-false ones behave like a FIFO whose front is moved to the
// back by every true element that arrives while it is non-empty.  Closed form used here: with T[k] the position of the
// k-th true element, the true element at x ends at i0 + rank(x), and a false element at x keeps jumping
// x -> T[x - i0] while x - i0 < n_true (it is the one sitting at the pivot slot when that true element arrives).
//
// Segments larger than BVHB_SMALL triangles are processed level by level, all segments of a level at once (one pass
// over the triangle positions per step, atomics into per-segment accumulators with per-wave run aggregation).
// Segments of at most BVHB_SMALL triangles are finished by one thread each, which runs the reference's serial recursion
// literally on its own index range.  Node numbers are assigned afterwards from subtree sizes (preorder = the
// reference's push_back order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BVHB_SMALL 32          // subtrees of at most this many triangles are built by one thread
#define BVHB_NPLANES 16
#define BVHB_NBINS 17
#define BVHB_BINWORDS (BVHB_NBINS * 7)
#define BVHB_WALK_CAP 64       // jumps a false element may take before the pointer-doubling fallback is used

namespace bvhb {

enum { K_PENDING = 0, K_INNER = 1, K_LEAF = 2, K_SMALL = 3 };

struct LNode {                 // node of the level-synchronous phase (level order)
	float bb[6];
	int i0, i1;
	int left, right;           // LNode ids of the children (K_INNER)
	int kind;
	int size;                  // nodes in the subtree
	int pre;                   // preorder number = index in the reference's node vector
	int _pad;
};

struct Seg {                   // a segment of the current level
	int node;                  // LNode id
	int i0, i1;
	int dim;
	float split;
	int ntrue;
	int lseg, rseg;            // segment ids of the children in the next level (-1: not a large segment)
};

struct ONode { uint8_t isleaf; uint8_t _pad[3]; int32_t fg, fd; float bmin[3], bmax[3]; };   // = mipt_bvh_node

// order-preserving float <-> uint32 map for atomicMin / atomicMax
__device__ __forceinline__ uint32_t fenc(float f) { uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float fdec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e ^ 0x80000000u) : ~e); }
__device__ __forceinline__ float fmin_ref(float o, float v) { return (v < o) ? v : o; }   // std::min(o, v)
__device__ __forceinline__ float fmax_ref(float o, float v) { return (o < v) ? v : o; }   // std::max(o, v)

// Per-triangle record: centroid + box of the three vertices, 48 B.
//   r0 = (cx, cy, cz, minx), r1 = (miny, minz, maxx, maxy), r2 = (maxz, -, -, -)
struct TriRec { float c[3], mn[3], mx[3]; };
__device__ __forceinline__ TriRec load_rec(const float4* __restrict__ rec, uint32_t id) {
	const float4 a = rec[3 * (size_t)id], b = rec[3 * (size_t)id + 1], c = rec[3 * (size_t)id + 2];
	TriRec r;
	r.c[0] = a.x; r.c[1] = a.y; r.c[2] = a.z; r.mn[0] = a.w; r.mn[1] = b.x; r.mn[2] = b.y; r.mx[0] = b.z; r.mx[1] = b.w; r.mx[2] = c.x;
	return r;
}
__device__ __forceinline__ float load_centroid(const float4* __restrict__ rec, uint32_t id, int dim) {
	const float4 a = rec[3 * (size_t)id];
	return dim == 0 ? a.x : (dim == 1 ? a.y : a.z);
}

__global__ void k_prepare(const float* __restrict__ vtx, int nverts, const char* __restrict__ tri_vtx, int stride, int n, float4* __restrict__ rec,
                          uint32_t* __restrict__ order, int* __restrict__ segof, int rootseg, int* __restrict__ bad) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int* q = (const int*)(tri_vtx + (size_t)i * stride);
	int a = q[0], b = q[1], c = q[2];
	if ((unsigned)a >= (unsigned)nverts || (unsigned)b >= (unsigned)nverts || (unsigned)c >= (unsigned)nverts) { atomicOr(bad, 1); a = b = c = 0; }
	float cen[3], mn[3], mx[3];
	for (int k = 0; k < 3; k++) {
		const float A = vtx[3 * (size_t)a + k], B = vtx[3 * (size_t)b + k], C = vtx[3 * (size_t)c + k];
		cen[k] = ((A + B) + C) / 3.f;
		mn[k] = fmin_ref(fmin_ref(A, B), C);
		mx[k] = fmax_ref(fmax_ref(A, B), C);
	}
	rec[3 * (size_t)i] = make_float4(cen[0], cen[1], cen[2], mn[0]);
	rec[3 * (size_t)i + 1] = make_float4(mn[1], mn[2], mx[0], mx[1]);
	rec[3 * (size_t)i + 2] = make_float4(mx[2], 0.f, 0.f, 0.f);
	order[i] = (uint32_t)i;
	segof[i] = rootseg;
}

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// ---- step A: box of the vertices and box of the centroids of every segment ----------------------------------------
// One wave per block; a wave walks `gpw` groups of 64 consecutive positions.  While all 64 positions of a group belong
// to the segment of the current run the lanes accumulate privately; the run is flushed (wave reduction + 12 atomics)
// when the segment changes.  Groups that straddle segments are reduced by a segmented shuffle scan (one set of atomics per run).
__global__ __launch_bounds__(64) void k_lvl_bounds(const float4* __restrict__ rec, const uint32_t* __restrict__ order, const int* __restrict__ segof,
                                                   int n, int gpw, uint32_t* __restrict__ acc) {
	const int lane = threadIdx.x;
	const long long base = (long long)blockIdx.x * gpw * 64;
	int run = -1;
	float mn[3], mx[3], cmn[3], cmx[3];
	auto reset = [&] { for (int k = 0; k < 3; k++) { mn[k] = cmn[k] = INFINITY; mx[k] = cmx[k] = -INFINITY; } };
	auto flush = [&] {
		if (run < 0) return;
		uint32_t* a = acc + 12 * (size_t)run;
		for (int k = 0; k < 3; k++) {
			const float v0 = wave_min(mn[k]), v1 = wave_max(mx[k]), v2 = wave_min(cmn[k]), v3 = wave_max(cmx[k]);
			if (lane == 0) { atomicMin(a + k, fenc(v0)); atomicMax(a + 3 + k, fenc(v1)); atomicMin(a + 6 + k, fenc(v2)); atomicMax(a + 9 + k, fenc(v3)); }
		}
	};
	reset();
	for (int g = 0; g < gpw; g++) {
		const long long x = base + (long long)g * 64 + lane;
		const int s = (x < n) ? segof[x] : -1;
		const int s0 = __builtin_amdgcn_readfirstlane(s);
		if (__all(s == s0)) {
			if (s0 != run) { flush(); reset(); run = s0; }
			if (s0 >= 0) {
				const TriRec r = load_rec(rec, order[x]);
				for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], r.mn[k]); mx[k] = fmaxf(mx[k], r.mx[k]); cmn[k] = fminf(cmn[k], r.c[k]); cmx[k] = fmaxf(cmx[k], r.c[k]); }
			}
		} else {
			// several segments in the group: segmented inclusive scan over the runs of equal segment, the last lane of a
			// run holds the run's boxes and issues the atomics
			flush(); reset(); run = -1;
			const int prev = __shfl_up(s, 1), nxt = __shfl_down(s, 1);
			const bool head = lane == 0 || prev != s, tail = lane == 63 || nxt != s;
			const unsigned long long heads = __ballot(head);
			const int start = 63 - __clzll(heads & ((2ull << lane) - 1ull));
			float lo[6], hi[6];
			for (int k = 0; k < 6; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
			if (s >= 0) {
				const TriRec r = load_rec(rec, order[x]);
				for (int k = 0; k < 3; k++) { lo[k] = r.mn[k]; hi[k] = r.mx[k]; lo[3 + k] = hi[3 + k] = r.c[k]; }
			}
			for (int o = 1; o < 64; o <<= 1) {
				const bool take = lane - o >= start;
				for (int k = 0; k < 6; k++) {
					const float a = __shfl_up(lo[k], o), b = __shfl_up(hi[k], o);
					if (take) { lo[k] = fminf(lo[k], a); hi[k] = fmaxf(hi[k], b); }
				}
			}
			if (tail && s >= 0) {
				uint32_t* a = acc + 12 * (size_t)s;
				for (int k = 0; k < 3; k++) { atomicMin(a + k, fenc(lo[k])); atomicMax(a + 3 + k, fenc(hi[k])); atomicMin(a + 6 + k, fenc(lo[3 + k])); atomicMax(a + 9 + k, fenc(hi[3 + k])); }
			}
		}
	}
	flush();
}

__global__ void k_lvl_init(int nseg, uint32_t* __restrict__ acc, uint32_t* __restrict__ bins) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < (size_t)nseg * 12) acc[i] = ((i % 12) % 6 < 3) ? 0xffffffffu : 0u;
	if (i < (size_t)nseg * BVHB_BINWORDS) {
		const int w = (int)(i % 7);   // 0..2 min, 3..5 max, 6 count; the reference starts from +-1E10 (TriangleMesh.cpp:1061-1064)
		bins[i] = w < 3 ? fenc(1E10f) : (w < 6 ? fenc(-1E10f) : 0u);
	}
}

// ---- step B: node box, split axis and the 16 candidate planes of every segment -------------------------------------
__global__ void k_lvl_planes(int nseg, Seg* __restrict__ segs, const uint32_t* __restrict__ acc, LNode* __restrict__ ln, float* __restrict__ planes) {
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	const uint32_t* a = acc + 12 * (size_t)s;
	LNode& nd = ln[segs[s].node];
	for (int k = 0; k < 6; k++) nd.bb[k] = fdec(a[k]);
	float cb[6];
	for (int k = 0; k < 6; k++) cb[k] = fdec(a[6 + k]);
	const float diag[3] = {cb[3] - cb[0], cb[4] - cb[1], cb[5] - cb[2]};
	int dim;
	if (diag[0] >= diag[1] && diag[0] >= diag[2]) dim = 0;
	else if (diag[1] >= diag[0] && diag[1] >= diag[2]) dim = 1;
	else dim = 2;
	segs[s].dim = dim;
	float* p = planes + (size_t)s * (BVHB_NPLANES + 2);
	for (int t = 0; t < BVHB_NPLANES; t++) p[t] = cb[dim] + diag[dim] * ((t + 1) / (float)(BVHB_NPLANES + 1));
	p[BVHB_NPLANES] = cb[dim];
	p[BVHB_NPLANES + 1] = diag[dim];
}

// ---- step C: triangles binned by the number of planes their centroid lies beyond ----------------------------------
// centroid <= plane[t] is monotone in t (planes are non-decreasing), so the left set of plane t is bins 0..t.
#define BVHB_SLOTS 3           // a group of 64 positions meets at most 3 segments of more than BVHB_SMALL (>= 33) triangles
__global__ __launch_bounds__(64) void k_lvl_bin(const float4* __restrict__ rec, const uint32_t* __restrict__ order, const int* __restrict__ segof,
                                                int n, int gpw, const Seg* __restrict__ segs, const float* __restrict__ planes, uint32_t* __restrict__ bins) {
	__shared__ uint32_t lb[BVHB_SLOTS][BVHB_BINWORDS];
	__shared__ int slot_seg[BVHB_SLOTS];
	const int lane = threadIdx.x;
	const long long base = (long long)blockIdx.x * gpw * 64;
	int run = -1, dim = 0;
	float sv[BVHB_NPLANES];
	auto init_lds = [&](int nslots) {
		for (int i = lane; i < nslots * BVHB_BINWORDS; i += 64) { const int w = i % 7; (&lb[0][0])[i] = w < 3 ? fenc(1E10f) : (w < 6 ? fenc(-1E10f) : 0u); }
		__syncthreads();
	};
	auto flush_slot = [&](int slot, int seg) {     // LDS bins of one run -> the segment's bins
		uint32_t* b = bins + (size_t)seg * BVHB_BINWORDS;
		for (int i = lane; i < BVHB_BINWORDS; i += 64) {
			const int w = i % 7;
			const uint32_t v = lb[slot][i];
			if (w < 3) { if (v != fenc(1E10f)) atomicMin(b + i, v); }
			else if (w < 6) { if (v != fenc(-1E10f)) atomicMax(b + i, v); }
			else if (v) atomicAdd(b + i, v);
		}
	};
	auto flush = [&] {
		if (run < 0) return;
		__syncthreads();
		flush_slot(0, run);
		__syncthreads();
		init_lds(1);
	};
	init_lds(BVHB_SLOTS);
	for (int g = 0; g < gpw; g++) {
		const long long x = base + (long long)g * 64 + lane;
		const int s = (x < n) ? segof[x] : -1;
		const int s0 = __builtin_amdgcn_readfirstlane(s);
		if (__all(s == s0)) {
			if (s0 != run) {
				flush(); run = s0;
				if (s0 >= 0) { dim = segs[s0].dim; for (int t = 0; t < BVHB_NPLANES; t++) sv[t] = planes[(size_t)s0 * (BVHB_NPLANES + 2) + t]; }
			}
			if (s0 >= 0) {
				const TriRec r = load_rec(rec, order[x]);
				const float c = dim == 0 ? r.c[0] : (dim == 1 ? r.c[1] : r.c[2]);
				int b = 0;
				for (int t = 0; t < BVHB_NPLANES; t++) b += (c <= sv[t]) ? 0 : 1;
				uint32_t* q = &lb[0][b * 7];
				for (int k = 0; k < 3; k++) { atomicMin(q + k, fenc(r.mn[k])); atomicMax(q + 3 + k, fenc(r.mx[k])); }
				atomicAdd(q + 6, 1u);
			}
		} else {
			// several segments in the group: every run of an active segment gets an LDS slot
			flush(); run = -1;
			const int prev = __shfl_up(s, 1);
			const bool head = (lane == 0 || prev != s) && s >= 0;
			const unsigned long long heads = __ballot(head);
			const int slot = __popcll(heads & ((2ull << lane) - 1ull)) - 1;
			const int nslots = __popcll(heads);
			if (head && slot < BVHB_SLOTS) slot_seg[slot] = s;
			if (s >= 0) {
				const TriRec r = load_rec(rec, order[x]);
				const int d = segs[s].dim;
				const float c = d == 0 ? r.c[0] : (d == 1 ? r.c[1] : r.c[2]);
				const float* p = planes + (size_t)s * (BVHB_NPLANES + 2);
				int b = 0;
				for (int t = 0; t < BVHB_NPLANES; t++) b += (c <= p[t]) ? 0 : 1;
				uint32_t* q = slot < BVHB_SLOTS ? &lb[slot][b * 7] : bins + (size_t)s * BVHB_BINWORDS + b * 7;
				for (int k = 0; k < 3; k++) { atomicMin(q + k, fenc(r.mn[k])); atomicMax(q + 3 + k, fenc(r.mx[k])); }
				atomicAdd(q + 6, 1u);
			}
			__syncthreads();
			const int used = nslots < BVHB_SLOTS ? nslots : BVHB_SLOTS;
			for (int k = 0; k < used; k++) flush_slot(k, slot_seg[k]);
			__syncthreads();
			init_lds(used);
		}
	}
	flush();
}

__device__ __forceinline__ float box_area(const float* mn, const float* mx) {
	const float s0 = mx[0] - mn[0], s1 = mx[1] - mn[1], s2 = mx[2] - mn[2];
	return 2 * (s0 * s1 + s0 * s2 + s1 * s2);
}

// ---- step D: cost of the 16 planes from the bins, first strict minimum (TriangleMesh.cpp:1052-1090) ----------------
__global__ void k_lvl_choose(int nseg, Seg* __restrict__ segs, const uint32_t* __restrict__ bins, const float* __restrict__ planes) {
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	const uint32_t* b = bins + (size_t)s * BVHB_BINWORDS;
	const float* p = planes + (size_t)s * (BVHB_NPLANES + 2);
	float best_factor = 0.5f, best = INFINITY;   // 1E50 narrowed to float
	for (int t = 0; t < BVHB_NPLANES; t++) {
		float lmn[3] = {1E10f, 1E10f, 1E10f}, lmx[3] = {-1E10f, -1E10f, -1E10f}, rmn[3] = {1E10f, 1E10f, 1E10f}, rmx[3] = {-1E10f, -1E10f, -1E10f};
		int nl = 0, nr = 0;
		for (int j = 0; j < BVHB_NBINS; j++) {
			const uint32_t* q = b + j * 7;
			if (j <= t) { for (int k = 0; k < 3; k++) { lmn[k] = fminf(lmn[k], fdec(q[k])); lmx[k] = fmaxf(lmx[k], fdec(q[3 + k])); } nl += (int)q[6]; }
			else        { for (int k = 0; k < 3; k++) { rmn[k] = fminf(rmn[k], fdec(q[k])); rmx[k] = fmaxf(rmx[k], fdec(q[3 + k])); } nr += (int)q[6]; }
		}
		const float cost = box_area(lmn, lmx) * nl + box_area(rmn, rmx) * nr;
		if (cost < best) { best = cost; best_factor = (t + 1) / (float)(BVHB_NPLANES + 1); }
	}
	segs[s].split = p[BVHB_NPLANES] + p[BVHB_NPLANES + 1] * best_factor;
}

// ---- step E/F: pred (centroid <= split) and its exclusive prefix sum over all positions ---------------------------
#define BVHB_SCAN_ROWS 16
#define BVHB_SCAN_TILE (256 * BVHB_SCAN_ROWS)

__global__ __launch_bounds__(256) void k_scan_sums(const float4* __restrict__ rec, const uint32_t* __restrict__ order, const int* __restrict__ segof, int n,
                                                   const Seg* __restrict__ segs, uint8_t* __restrict__ pf, uint32_t* __restrict__ bsum) {
	__shared__ uint32_t ws[4];
	const size_t base = (size_t)blockIdx.x * BVHB_SCAN_TILE;
	uint32_t cnt = 0;
	for (int j = 0; j < BVHB_SCAN_ROWS; j++) {
		const size_t x = base + (size_t)j * 256 + threadIdx.x;
		if (x >= (size_t)n) break;
		const int s = segof[x];
		uint8_t p = 0;
		if (s >= 0) p = load_centroid(rec, order[x], segs[s].dim) <= segs[s].split ? 1 : 0;
		pf[x] = p;
		cnt += p;
	}
	for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(1024) void k_scan_top(uint32_t* __restrict__ bsum, int nb) {   // exclusive scan of the block sums, one block
	__shared__ uint32_t ws[16];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int base = 0; base < nb; base += 1024) {
		const int i = base + threadIdx.x;
		const uint32_t v = i < nb ? bsum[i] : 0;
		uint32_t incl = v;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((threadIdx.x & 63) >= o) incl += t; }
		if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
		__syncthreads();
		uint32_t woff = 0;
		for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += ws[w];
		const uint32_t c = carry;
		if (i < nb) bsum[i] = c + woff + incl - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry = c + woff + incl;
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint8_t* __restrict__ pf, int n, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ S) {
	__shared__ uint32_t ws[4];
	const size_t base = (size_t)blockIdx.x * BVHB_SCAN_TILE;
	uint32_t carry = bsum[blockIdx.x];
	for (int j = 0; j < BVHB_SCAN_ROWS; j++) {
		const size_t x = base + (size_t)j * 256 + threadIdx.x;
		const uint32_t v = x < (size_t)n ? pf[x] : 0;
		uint32_t incl = v;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((threadIdx.x & 63) >= o) incl += t; }
		if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
		__syncthreads();
		uint32_t woff = 0;
		for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += ws[w];
		const uint32_t total = ws[0] + ws[1] + ws[2] + ws[3];
		if (x < (size_t)n) {
			S[x] = carry + woff + incl - v;
			if (x == (size_t)n - 1) S[n] = carry + woff + incl;
		}
		carry += total;
		__syncthreads();
	}
}

// ---- step G/H: the reference's in-place partition as a scatter -----------------------------------------------------
__global__ void k_lvl_scatter_true(const uint32_t* __restrict__ order, const int* __restrict__ segof, int n, const Seg* __restrict__ segs,
                                   const uint8_t* __restrict__ pf, const uint32_t* __restrict__ S, uint32_t* __restrict__ order2, uint32_t* __restrict__ tpos) {
	const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= (size_t)n) return;
	const int s = segof[x];
	if (s < 0) { order2[x] = order[x]; return; }
	if (!pf[x]) return;
	const int i0 = segs[s].i0;
	const uint32_t d = (uint32_t)i0 + (S[x] - S[i0]);
	order2[d] = order[x];
	tpos[d] = (uint32_t)x;
}

__global__ void k_lvl_scatter_false(const uint32_t* __restrict__ order, const int* __restrict__ segof, int n, const Seg* __restrict__ segs,
                                    const uint8_t* __restrict__ pf, const uint32_t* __restrict__ S, uint32_t* __restrict__ order2,
                                    const uint32_t* __restrict__ tpos, int* __restrict__ unresolved) {
	const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= (size_t)n) return;
	const int s = segof[x];
	if (s < 0 || pf[x]) return;
	const uint32_t lim = (uint32_t)segs[s].i0 + (S[segs[s].i1] - S[segs[s].i0]);
	uint32_t y = (uint32_t)x;
	for (int step = 0; y < lim; step++) {
		if (step == BVHB_WALK_CAP) { atomicOr(unresolved, 1); return; }
		y = tpos[y];
	}
	order2[y] = order[x];
}

// pointer doubling over the jump table (fallback for long chains: a few false elements in front of many true ones)
__global__ void k_lvl_double(const int* __restrict__ segof, int n, const Seg* __restrict__ segs, const uint32_t* __restrict__ S, uint32_t* __restrict__ tpos) {
	const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= (size_t)n) return;
	const int s = segof[x];
	if (s < 0) return;
	const uint32_t lim = (uint32_t)segs[s].i0 + (S[segs[s].i1] - S[segs[s].i0]);
	if (x >= lim) return;
	const uint32_t y = tpos[x];
	if (y < lim && y != x) tpos[x] = tpos[y];   // any value read here is a later point of the same chain
}

// ---- step I/J: leaf test, children, next level's segments ---------------------------------------------------------
struct Counters { int ln_count, nseg_next, nsmall, unresolved; };

__global__ void k_lvl_children(int nseg, Seg* __restrict__ segs, const uint32_t* __restrict__ S, LNode* __restrict__ ln, Seg* __restrict__ next,
                               int* __restrict__ smalls, Counters* __restrict__ cnt) {
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	Seg& g = segs[s];
	const int nt = (int)(S[g.i1] - S[g.i0]), n = g.i1 - g.i0;
	g.ntrue = nt; g.lseg = g.rseg = -1;
	LNode& nd = ln[g.node];
	if (nt == 0 || nt == n || n <= 4) { nd.kind = K_LEAF; nd.size = 1; return; }   // pivot < i0 || pivot >= i1-1 || i1 <= i0+4
	nd.kind = K_INNER;
	const int l = atomicAdd(&cnt->ln_count, 2);
	nd.left = l; nd.right = l + 1;
	for (int c = 0; c < 2; c++) {
		const int a = c ? g.i0 + nt : g.i0, b = c ? g.i1 : g.i0 + nt;
		LNode& ch = ln[l + c];
		ch.i0 = a; ch.i1 = b; ch.left = ch.right = -1; ch.size = 0; ch.pre = 0;
		if (b - a > BVHB_SMALL) {
			ch.kind = K_PENDING;
			const int q = atomicAdd(&cnt->nseg_next, 1);
			next[q].node = l + c; next[q].i0 = a; next[q].i1 = b;
			if (c) g.rseg = q; else g.lseg = q;
		} else {
			ch.kind = K_SMALL;
			smalls[atomicAdd(&cnt->nsmall, 1)] = l + c;
		}
	}
}

__global__ void k_lvl_resegment(const int* __restrict__ segof, int n, const Seg* __restrict__ segs, int* __restrict__ segof2) {
	const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= (size_t)n) return;
	const int s = segof[x];
	int r = -1;
	if (s >= 0) r = ((int)x < segs[s].i0 + segs[s].ntrue) ? segs[s].lseg : segs[s].rseg;
	segof2[x] = r;
}

// ---- small subtrees: the reference's recursion, one thread per subtree --------------------------------------------
// Nodes of more than BVHB_LITERAL triangles cost their 16 planes from 17 bins kept in LDS (one column per thread, so
// the records are read three times per node instead of eighteen); smaller nodes run the reference's loops as they are.
#define BVHB_LITERAL 8
__global__ __launch_bounds__(64) void k_small_subtrees(int nsmall, const int* __restrict__ smalls, LNode* __restrict__ ln, const float4* __restrict__ rec,
                                                       uint32_t* __restrict__ order, ONode* __restrict__ sn) {
	__shared__ float sbox[BVHB_NBINS * 6][64];
	__shared__ int scnt[BVHB_NBINS][64];
	const int lane = threadIdx.x;
	const int q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= nsmall) return;
	LNode& root = ln[smalls[q]];
	ONode* out = sn + 2 * (size_t)root.i0;         // a subtree over m triangles has at most 2m-1 nodes
	int count = 0;
	int stk[BVHB_SMALL + 2][3];                     // (i0, i1, parent << 1 | side)
	int sp = 0;
	stk[0][0] = root.i0; stk[0][1] = root.i1; stk[0][2] = -1; sp = 1;
	while (sp > 0) {
		sp--;
		const int i0 = stk[sp][0], i1 = stk[sp][1], link = stk[sp][2];
		const int node = count++;
		if (link >= 0) { if (link & 1) out[link >> 1].fd = node; else out[link >> 1].fg = node; }
		// build_bbox + build_centers_bbox (TriangleMesh.cpp:843-875)
		float bmn[3], bmx[3], cmn[3], cmx[3];
		{
			const TriRec r = load_rec(rec, order[i0]);
			for (int k = 0; k < 3; k++) { bmn[k] = r.mn[k]; bmx[k] = r.mx[k]; cmn[k] = cmx[k] = r.c[k]; }
		}
		for (int i = i0 + 1; i < i1; i++) {
			const TriRec r = load_rec(rec, order[i]);
			for (int k = 0; k < 3; k++) { bmn[k] = fmin_ref(bmn[k], r.mn[k]); bmx[k] = fmax_ref(bmx[k], r.mx[k]); cmn[k] = fmin_ref(cmn[k], r.c[k]); cmx[k] = fmax_ref(cmx[k], r.c[k]); }
		}
		ONode o;
		o.isleaf = 1; o._pad[0] = o._pad[1] = o._pad[2] = 0; o.fg = i0; o.fd = i1;
		for (int k = 0; k < 3; k++) { o.bmin[k] = bmn[k]; o.bmax[k] = bmx[k]; }
		const float diag[3] = {cmx[0] - cmn[0], cmx[1] - cmn[1], cmx[2] - cmn[2]};
		int dim;
		if (diag[0] >= diag[1] && diag[0] >= diag[2]) dim = 0;
		else if (diag[1] >= diag[0] && diag[1] >= diag[2]) dim = 1;
		else dim = 2;
		const float cb = dim == 0 ? cmn[0] : (dim == 1 ? cmn[1] : cmn[2]);
		const float dg = dim == 0 ? diag[0] : (dim == 1 ? diag[1] : diag[2]);
		float best_factor = 0.5f, best = INFINITY;
		if (i1 - i0 <= BVHB_LITERAL) {
			for (int t = 0; t < BVHB_NPLANES; t++) {
				const float factor = (t + 1) / (float)(BVHB_NPLANES + 1);
				const float sv = cb + dg * factor;
				float lmn[3] = {1E10f, 1E10f, 1E10f}, lmx[3] = {-1E10f, -1E10f, -1E10f}, rmn[3] = {1E10f, 1E10f, 1E10f}, rmx[3] = {-1E10f, -1E10f, -1E10f};
				int nl = 0, nr = 0;
				for (int i = i0; i < i1; i++) {
					const TriRec r = load_rec(rec, order[i]);
					const float c = dim == 0 ? r.c[0] : (dim == 1 ? r.c[1] : r.c[2]);
					if (c <= sv) { for (int k = 0; k < 3; k++) { lmn[k] = fmin_ref(lmn[k], r.mn[k]); lmx[k] = fmax_ref(lmx[k], r.mx[k]); } nl++; }
					else         { for (int k = 0; k < 3; k++) { rmn[k] = fmin_ref(rmn[k], r.mn[k]); rmx[k] = fmax_ref(rmx[k], r.mx[k]); } nr++; }
				}
				const float cost = box_area(lmn, lmx) * nl + box_area(rmn, rmx) * nr;
				if (cost < best) { best = cost; best_factor = factor; }
			}
		} else {
			float sv[BVHB_NPLANES];
#pragma unroll
			for (int t = 0; t < BVHB_NPLANES; t++) sv[t] = cb + dg * ((t + 1) / (float)(BVHB_NPLANES + 1));
			for (int j = 0; j < BVHB_NBINS; j++) {
				for (int k = 0; k < 3; k++) { sbox[j * 6 + k][lane] = 1E10f; sbox[j * 6 + 3 + k][lane] = -1E10f; }
				scnt[j][lane] = 0;
			}
			for (int i = i0; i < i1; i++) {
				const TriRec r = load_rec(rec, order[i]);
				const float c = dim == 0 ? r.c[0] : (dim == 1 ? r.c[1] : r.c[2]);
				int b = 0;
#pragma unroll
				for (int t = 0; t < BVHB_NPLANES; t++) b += (c <= sv[t]) ? 0 : 1;
				for (int k = 0; k < 3; k++) {
					sbox[b * 6 + k][lane] = fmin_ref(sbox[b * 6 + k][lane], r.mn[k]);
					sbox[b * 6 + 3 + k][lane] = fmax_ref(sbox[b * 6 + 3 + k][lane], r.mx[k]);
				}
				scnt[b][lane]++;
			}
			float cl[BVHB_NPLANES];
			{
				float mn[3] = {1E10f, 1E10f, 1E10f}, mx[3] = {-1E10f, -1E10f, -1E10f};
				int cnt = 0;
#pragma unroll
				for (int t = 0; t < BVHB_NPLANES; t++) {     // left set of plane t = bins 0..t
					for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], sbox[t * 6 + k][lane]); mx[k] = fmaxf(mx[k], sbox[t * 6 + 3 + k][lane]); }
					cnt += scnt[t][lane];
					cl[t] = box_area(mn, mx) * cnt;
				}
			}
			{
				float mn[3] = {1E10f, 1E10f, 1E10f}, mx[3] = {-1E10f, -1E10f, -1E10f};
				int cnt = 0;
#pragma unroll
				for (int t = BVHB_NPLANES - 1; t >= 0; t--) {   // right set of plane t = bins t+1..16
					for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], sbox[(t + 1) * 6 + k][lane]); mx[k] = fmaxf(mx[k], sbox[(t + 1) * 6 + 3 + k][lane]); }
					cnt += scnt[t + 1][lane];
					cl[t] = cl[t] + box_area(mn, mx) * cnt;
				}
			}
#pragma unroll
			for (int t = 0; t < BVHB_NPLANES; t++) if (cl[t] < best) { best = cl[t]; best_factor = (t + 1) / (float)(BVHB_NPLANES + 1); }
		}
		const float split = cb + dg * best_factor;
		int pivot = i0 - 1;
		for (int i = i0; i < i1; i++) {
			const uint32_t id = order[i];
			if (load_centroid(rec, id, dim) <= split) { pivot++; const uint32_t o2 = order[pivot]; order[pivot] = id; order[i] = o2; }
		}
		if (!(pivot < i0 || pivot >= i1 - 1 || i1 <= i0 + 4)) {
			o.isleaf = 0;
			stk[sp][0] = pivot + 1; stk[sp][1] = i1; stk[sp][2] = (node << 1) | 1; sp++;   // right is popped after the whole left subtree
			stk[sp][0] = i0; stk[sp][1] = pivot + 1; stk[sp][2] = (node << 1); sp++;
		}
		out[node] = o;
	}
	root.size = count;
}

// ---- node numbering: subtree sizes bottom-up, preorder numbers top-down, emission ---------------------------------
__global__ void k_sizes(LNode* __restrict__ ln, int begin, int end) {
	const int i = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= end) return;
	if (ln[i].kind == K_INNER) ln[i].size = 1 + ln[ln[i].left].size + ln[ln[i].right].size;
}
__global__ void k_preorder(LNode* __restrict__ ln, int begin, int end) {
	const int i = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= end) return;
	if (ln[i].kind == K_INNER) { ln[ln[i].left].pre = ln[i].pre + 1; ln[ln[i].right].pre = ln[i].pre + 1 + ln[ln[i].left].size; }
}
__global__ void k_emit(const LNode* __restrict__ ln, int nln, const ONode* __restrict__ sn, ONode* __restrict__ out) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nln) return;
	const LNode& nd = ln[i];
	if (nd.kind == K_SMALL) {
		const ONode* src = sn + 2 * (size_t)nd.i0;
		for (int j = 0; j < nd.size; j++) {
			ONode o = src[j];
			if (!o.isleaf) { o.fg += nd.pre; o.fd += nd.pre; }
			out[nd.pre + j] = o;
		}
		return;
	}
	ONode o;
	o._pad[0] = o._pad[1] = o._pad[2] = 0;
	for (int k = 0; k < 3; k++) { o.bmin[k] = nd.bb[k]; o.bmax[k] = nd.bb[3 + k]; }
	if (nd.kind == K_INNER) { o.isleaf = 0; o.fg = ln[nd.left].pre; o.fd = ln[nd.right].pre; }
	else { o.isleaf = 1; o.fg = nd.i0; o.fd = nd.i1; }
	out[nd.pre] = o;
}

// ---- the traversal's own records, made on the device from the built tree (round 4: mipt_device_mesh_build) --------------------
// What convert_mesh (mipt.hip) makes on the host from downloaded nodes and host-side Triangle records, made where the tree
// already is: fat nodes (both children's boxes + references, inner nodes in the reference's depth-first order), the 64-byte
// intersection and shading records of the permuted triangles, and the UV index triples of alpha-tested meshes.

// inner rank of every node of the reference's node vector = number of inner nodes in front of it (exclusive scan of !isleaf)
#define BVHB_RANK_TILE 1024
__global__ __launch_bounds__(256) void k_rank_sums(const ONode* __restrict__ nodes, int total, uint32_t* __restrict__ bsum) {
	__shared__ uint32_t ws[4];
	const size_t base = (size_t)blockIdx.x * BVHB_RANK_TILE;
	uint32_t c = 0;
	for (int k = 0; k < BVHB_RANK_TILE / 256; k++) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < (size_t)total && !nodes[i].isleaf) c++; }
	for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void k_rank_apply(const ONode* __restrict__ nodes, int total, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ irank) {
	// (one wave walks the tile in node order: 4 x 256 nodes, ballot prefix inside each group of 64)
	__shared__ uint32_t wsum[4][4];
	const size_t base = (size_t)blockIdx.x * BVHB_RANK_TILE;
	const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t mine[4]; uint32_t cnt[4];
	for (int k = 0; k < 4; k++) {
		const size_t i = base + (size_t)k * 256 + threadIdx.x;
		const bool inner = i < (size_t)total && !nodes[i].isleaf;
		const unsigned long long m = __ballot(inner);
		mine[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		cnt[k] = (uint32_t)__popcll(m);
		if (lane == 0) wsum[k][w] = cnt[k];
	}
	__syncthreads();
	uint32_t off = bsum[blockIdx.x];
	for (int k = 0; k < 4; k++) {
		uint32_t before = off;
		for (unsigned ww = 0; ww < w; ww++) before += wsum[k][ww];
		const size_t i = base + (size_t)k * 256 + threadIdx.x;
		if (i < (size_t)total) irank[i] = before + mine[k];
		off += wsum[k][0] + wsum[k][1] + wsum[k][2] + wsum[k][3];
	}
}
// levels of inner nodes: depth[root] = 1, every pass hands depth + 1 to the inner children of the nodes that have one.  After P passes
// every inner node of level <= P + 1 has its depth; a tree with more levels than the traversal stack holds leaves inner nodes at 0.
__global__ void k_depth_init(uint8_t* __restrict__ depth, int total) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < (size_t)total) depth[i] = i == 0 ? 1 : 0;
}
__global__ void k_depth_pass(const ONode* __restrict__ nodes, int total, uint8_t* __restrict__ depth, int level) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)total || depth[i] != level || nodes[i].isleaf) return;
	depth[nodes[i].fg] = (uint8_t)(level + 1); depth[nodes[i].fd] = (uint8_t)(level + 1);
}
// bad[0]: a child index out of order / a leaf range out of bounds, bad[1]: a leaf with more than 32 triangles, bad[2]: an inner node no pass
// reached (tree deeper than `max_levels` levels of inner nodes), bad[3]: a material group above 2^30
__global__ void k_fat_nodes(const ONode* __restrict__ nodes, int total, int ntri, const uint32_t* __restrict__ irank, const uint8_t* __restrict__ depth, int max_levels,
                            DFatNode* __restrict__ fat, int* __restrict__ bad, uint32_t* __restrict__ root_ref) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)total) return;
	const ONode nd = nodes[i];
	auto child_ref = [&](int c) -> uint32_t {
		const ONode ch = nodes[c];
		if (!ch.isleaf) return irank[c];
		const int cnt = ch.fd - ch.fg;
		if (ch.fg < 0 || ch.fd > ntri || cnt <= 0) { atomicOr(&bad[0], 1); return 0u; }
		if (cnt >= MIPT_LEAF_MAX_TRIS) { atomicMax(&bad[1], cnt); return 0u; }      // (a fat leaf: the mesh goes through the host-side conversion, which files its count)
		return MIPT_LEAF_BIT | ((uint32_t)(cnt - 1) << 26) | (uint32_t)ch.fg;
	};
	if (i == 0) *root_ref = nd.isleaf ? (nd.fd - nd.fg >= MIPT_LEAF_MAX_TRIS || nd.fd - nd.fg <= 0 ? ((nd.fd - nd.fg <= 0 ? atomicOr(&bad[0], 1) : atomicMax(&bad[1], nd.fd - nd.fg)), 0u) : (MIPT_LEAF_BIT | ((uint32_t)(nd.fd - nd.fg - 1) << 26) | (uint32_t)nd.fg)) : 0u;
	if (nd.isleaf) return;
	if (nd.fg <= (int)i || nd.fg >= total || nd.fd <= (int)i || nd.fd >= total) { atomicOr(&bad[0], 1); return; }
	if (depth[i] == 0 || depth[i] > max_levels) atomicOr(&bad[2], 1);
	DFatNode f;
	const ONode l = nodes[nd.fg], r = nodes[nd.fd];
	for (int k = 0; k < 3; k++) { f.l[k][0] = l.bmin[k]; f.l[k][1] = l.bmax[k]; f.r[k][0] = r.bmin[k]; f.r[k][1] = r.bmax[k]; }
	f.lref = child_ref(nd.fg); f.rref = child_ref(nd.fd); f._pad[0] = f._pad[1] = 0;
	fat[irank[i]] = f;
}
// Triangle's constructor (TriangleMesh.h:70-78) and the gather of corner normals / UVs (TriangleMesh.cpp:812-829) for position i of the
// reordered mesh = input triangle order[i].  tri = the caller's TriangleIndices records (44 bytes: vtx ijk, uv ijk, n ijk, group, faceID).
__global__ void k_tri_records(const float* __restrict__ vtx, const float* __restrict__ normals, int nnormals, const float* __restrict__ uvs, int nuvs,
                              const char* __restrict__ tri, int stride, const uint32_t* __restrict__ order, int n,
                              DTriIsect* __restrict__ ti, DTriShade* __restrict__ ts, int* __restrict__ uvidx, int* __restrict__ bad) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)n) return;
	const int* t = reinterpret_cast<const int*>(tri + (size_t)order[i] * stride);
	const int vi = t[0], vj = t[1], vk = t[2], ui = t[3], uj = t[4], uk = t[5], ni = t[6], nj = t[7], nk = t[8], group = t[9];
	const float A[3] = {vtx[3 * (size_t)vi], vtx[3 * (size_t)vi + 1], vtx[3 * (size_t)vi + 2]};
	float u[3], v[3];
	for (int k = 0; k < 3; k++) { u[k] = vtx[3 * (size_t)vj + k] - A[k]; v[k] = vtx[3 * (size_t)vk + k] - A[k]; }
	DTriIsect I;
	for (int k = 0; k < 3; k++) { I.A[k] = A[k]; I.u[k] = u[k]; I.v[k] = v[k]; }
	I.N[0] = u[1] * v[2] - u[2] * v[1]; I.N[1] = u[2] * v[0] - u[0] * v[2]; I.N[2] = u[0] * v[1] - u[1] * v[0];
	I.m11 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
	I.m22 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	I.m12 = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
	I.invdetm = 1.f / (I.m11 * I.m22 - I.m12 * I.m12);
	ti[i] = I;
	DTriShade S;
	for (int k = 0; k < 9; k++) S.normals[k] = 0.f;
	if (nnormals != 0) {
		const int nidx[3] = {ni, nj, nk};
		for (int c = 0; c < 3; c++) if ((unsigned)nidx[c] < (unsigned)nnormals) for (int k = 0; k < 3; k++) S.normals[3 * c + k] = normals[3 * (size_t)nidx[c] + k];      // (an index outside the list: zeros, where the host loop would read outside it)
	}
	for (int k = 0; k < 6; k++) S.uvs[k] = 0.f;
	if (nuvs != 0) {
		const int tidx[3] = {ui, uj, uk};
		for (int c = 0; c < 3; c++) if ((unsigned)tidx[c] < (unsigned)nuvs) { S.uvs[2 * c] = uvs[3 * (size_t)tidx[c]]; S.uvs[2 * c + 1] = uvs[3 * (size_t)tidx[c] + 1]; }
		if (uvidx) { uvidx[3 * i] = ui; uvidx[3 * i + 1] = uj; uvidx[3 * i + 2] = uk; }
	}
	S.group = group;
	if (group > MIPT_GROUP_MASK) atomicOr(&bad[3], 1);
	if (group >= 0 && nuvs != 0 && ui >= 0 && ui < nuvs) S.group |= MIPT_GROUP_UV_OK;
	ts[i] = S;
}
// ---- TriMesh::setup_tangents (TriangleMesh.cpp:572-711) on the reordered mesh: per face (sdir, tdir) from the UV gradients, per vertex
// the sum of its faces' sdir IN ASCENDING FACE ORDER (the reference walks the faces once: the order fixes the rounding), Gram-Schmidt
// against the vertex normal, then the three corner tangents of every face.  Same operations as the host version (host/mipt_host.cpp).
// (a UV index outside the list would be an out-of-bounds DEVICE read — a memory fault that ends the process, where the host loop
// reads garbage: such a face counts as a face without UVs.  Vertex indices were range-checked by k_prepare before the tree was built.)
__global__ void k_tan_face(const float* __restrict__ vtx, const float* __restrict__ uvs, int nuvs, const char* __restrict__ tri, int stride, const uint32_t* __restrict__ order, int n,
                           float* __restrict__ sdir, uint8_t* __restrict__ has, uint32_t* __restrict__ first) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)n) return;
	const int* t = reinterpret_cast<const int*>(tri + (size_t)order[i] * stride);
	const int vi = t[0], vj = t[1], vk = t[2], ui = t[3], uj = t[4], uk = t[5];
	atomicAdd(&first[vi], 1u); atomicAdd(&first[vj], 1u); atomicAdd(&first[vk], 1u);      // incident corners per vertex (scanned into list starts)
	const bool h = (unsigned)ui < (unsigned)nuvs && (unsigned)uj < (unsigned)nuvs && (unsigned)uk < (unsigned)nuvs;
	has[i] = h ? 1 : 0;
	if (!h) return;
	float vA[3], vB[3];
	for (int k = 0; k < 3; k++) { vA[k] = vtx[3 * (size_t)vj + k] - vtx[3 * (size_t)vi + k]; vB[k] = vtx[3 * (size_t)vk + k] - vtx[3 * (size_t)vi + k]; }
	const float sA0 = uvs[3 * (size_t)uj] - uvs[3 * (size_t)ui], sA1 = uvs[3 * (size_t)uj + 1] - uvs[3 * (size_t)ui + 1];
	const float sB0 = uvs[3 * (size_t)uk] - uvs[3 * (size_t)ui], sB1 = uvs[3 * (size_t)uk + 1] - uvs[3 * (size_t)ui + 1];
	const float det = sA0 * sB1 - sB0 * sA1;
	for (int k = 0; k < 3; k++) sdir[3 * i + k] = det != 0 ? (sB1 * vA[k] - sA1 * vB[k]) / det : 0.00001f * vA[k];
}
// generic exclusive scan of uint32 (tile sums, k_scan_top on the sums, apply): first[] of the vertex -> corner table
__global__ __launch_bounds__(256) void k_u32_sums(const uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ bsum) {
	__shared__ uint32_t ws[4];
	const size_t base = (size_t)blockIdx.x * BVHB_RANK_TILE;
	uint32_t c = 0;
	for (int k = 0; k < BVHB_RANK_TILE / 256; k++) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n) c += a[i]; }
	for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void k_u32_apply(uint32_t* __restrict__ a, size_t n, const uint32_t* __restrict__ bsum) {      // in place: a[i] <- sum of a[0 .. i-1]
	__shared__ uint32_t wsum[4];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0) carry = bsum[blockIdx.x];
	__syncthreads();
	const size_t base = (size_t)blockIdx.x * BVHB_RANK_TILE;
	for (int k = 0; k < BVHB_RANK_TILE / 256; k++) {
		const size_t i = base + (size_t)k * 256 + threadIdx.x;
		const uint32_t v = i < n ? a[i] : 0u;
		uint32_t incl = v;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((threadIdx.x & 63) >= (unsigned)o) incl += t; }
		if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
		__syncthreads();
		uint32_t woff = 0;
		for (unsigned w = 0; w < (threadIdx.x >> 6); w++) woff += wsum[w];
		const uint32_t c = carry;
		if (i < n) a[i] = c + woff + incl - v;
		__syncthreads();
		if (threadIdx.x == 255) carry = c + woff + incl;
		__syncthreads();
	}
}
__global__ void k_tan_fill(const char* __restrict__ tri, int stride, const uint32_t* __restrict__ order, int n, uint32_t* __restrict__ fill, uint32_t* __restrict__ corner) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)n) return;
	const int* t = reinterpret_cast<const int*>(tri + (size_t)order[i] * stride);
	for (int k = 0; k < 3; k++) corner[atomicAdd(&fill[t[k]], 1u)] = 3u * (uint32_t)i + (uint32_t)k;
}
// first[] is the table's start per vertex, fill[] (advanced by k_tan_fill) its end.  A vertex's list arrives in arbitrary order: it is
// sorted here (heap sort: the poles of a UV sphere have a thousand incident faces) before the sum runs in ascending face order.
__global__ void k_tan_vertex(const float* __restrict__ normals, int nnormals, const char* __restrict__ tri, int stride, const uint32_t* __restrict__ order,
                             const uint32_t* __restrict__ first, const uint32_t* __restrict__ fill, uint32_t* __restrict__ corner, const float* __restrict__ sdir,
                             const uint8_t* __restrict__ has, int nv, float* __restrict__ tangents) {
	const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= (size_t)nv) return;
	uint32_t* a = corner + first[v];
	const int m = (int)(fill[v] - first[v]);
	auto sift = [&](int root, int end) {
		for (;;) {
			int child = 2 * root + 1;
			if (child >= end) return;
			if (child + 1 < end && a[child] < a[child + 1]) child++;
			if (a[root] >= a[child]) return;
			const uint32_t t = a[root]; a[root] = a[child]; a[child] = t;
			root = child;
		}
	};
	for (int r = m / 2 - 1; r >= 0; r--) sift(r, m);
	for (int e = m - 1; e > 0; e--) { const uint32_t t = a[0]; a[0] = a[e]; a[e] = t; sift(0, e); }
	float t1[3] = {0.f, 0.f, 0.f};
	int nidx = 0;
	for (int e = 0; e < m; e++) {
		const uint32_t f = a[e] / 3u, k = a[e] % 3u;
		if (has[f]) for (int c = 0; c < 3; c++) t1[c] = t1[c] + sdir[3 * (size_t)f + c];
		nidx = reinterpret_cast<const int*>(tri + (size_t)order[f] * stride)[6 + k];
	}
	float N[3] = {0.f, 0.f, 0.f};
	if ((unsigned)nidx < (unsigned)nnormals) for (int c = 0; c < 3; c++) N[c] = normals[3 * (size_t)nidx + c];
	{ const float len = sqrtf(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]); N[0] = N[0] / len; N[1] = N[1] / len; N[2] = N[2] / len; }
	const float d = t1[0] * N[0] + t1[1] * N[1] + t1[2] * N[2];
	float r[3] = {t1[0] - d * N[0], t1[1] - d * N[1], t1[2] - d * N[2]};
	{ const float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]); r[0] = r[0] / len; r[1] = r[1] / len; r[2] = r[2] / len; }
	for (int c = 0; c < 3; c++) tangents[3 * v + c] = r[c];
}
__global__ void k_tan_soup(const char* __restrict__ tri, int stride, const uint32_t* __restrict__ order, int n, const float* __restrict__ tangents, float* __restrict__ soup) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)n) return;
	const int* t = reinterpret_cast<const int*>(tri + (size_t)order[i] * stride);
	for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) soup[9 * i + 3 * k + c] = tangents[3 * (size_t)t[k] + c];
}

// a mesh that is not the scene's first: child references are scene-wide (inner: + node_base; leaf: first triangle + tri_base)
// (in place: ONE pointer, not a restrict-qualified source and destination that alias — ADVICE r4)
__global__ void k_rebase_nodes(DFatNode* nodes, size_t n, uint32_t node_base, uint32_t tri_base) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t l = nodes[i].lref, r = nodes[i].rref;
	nodes[i].lref = (l & MIPT_LEAF_BIT) ? l + tri_base : l + node_base;
	nodes[i].rref = (r & MIPT_LEAF_BIT) ? r + tri_base : r + node_base;
}

}   // namespace bvhb
