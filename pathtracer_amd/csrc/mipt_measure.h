// mipt_measure.h — measurement aids behind the C ABI (mipt_measure_*: ceilings that bench.py measures on the device itself) and diagnostics
// (part of the one translation unit csrc/mipt.hip: included there, after the context and the render loop it uses)

#pragma once
// Achievable HBM read bandwidth of this device, for the roofline's denominator (SURVEY.md §8d asks for the measured
// figure beside the 8 TB/s data-sheet peak): a grid-stride sum over `bytes` of device memory with 16-byte loads.
__global__ void __launch_bounds__(256) k_stream_read(const float4* __restrict__ src, size_t n4, float* __restrict__ sink) {
	float acc = 0.f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		float4 v = src[i];
		acc += v.x + v.y + v.z + v.w;
	}
	if (acc == 123.456f) *sink = acc;               // never true for the zero-filled buffer; keeps the loads alive
}
extern "C" int mipt_measure_stream_read(mipt_ctx* c, uint64_t bytes, int repeats, double* gb_per_s) {
	if (!c || !gb_per_s || bytes < (1u << 20) || repeats < 1) return fail(c, MIPT_ERR_INVALID, "bad arguments");
	HIPCHK(c, hipSetDevice(c->device));
	float4* buf = nullptr; float* sink = nullptr;
	HIPCHK(c, hipMalloc(&buf, bytes));
	if (hipMalloc(&sink, 4) != hipSuccess) { hipFree(buf); return fail(c, MIPT_ERR_HIP, "hipMalloc failed"); }
	hipMemset(buf, 0, bytes);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const unsigned grid = (unsigned)c->n_cus * 8u;
	hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, 0, buf, (size_t)(bytes / 16), sink);   // warm-up
	hipEventRecord(e0, 0);
	for (int r = 0; r < repeats; r++) hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, 0, buf, (size_t)(bytes / 16), sink);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1); hipFree(buf); hipFree(sink);
	if (!(ms > 0.f)) return fail(c, MIPT_ERR_HIP, "timing failed");
	*gb_per_s = (double)bytes * repeats / (ms * 1e-3) / 1e9;
	return MIPT_OK;
}

// The access pattern of the traversal kernels on its own, with a known byte count: every lane reads whole 64-byte records
// (four 16-byte loads, like a fat BVH node) at pseudo-random 64-byte-aligned offsets of a buffer far larger than the
// Infinity Cache.  Run under `rocprofv3 --pmc FETCH_SIZE` it tells how many bytes that counter reports per gathered byte
// (tools/fetch_calibration.py): for wide streaming reads the factor is 1/2 (MI355X_MICROARCH.md), for gathers it was unknown.
__global__ void __launch_bounds__(256) k_gather_read(const float4* __restrict__ src, unsigned long long nrec, int iters, float* __restrict__ sink) {
	float acc = 0.f;
	unsigned long long x = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
	for (int it = 0; it < iters; it++) {
		x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;   // splitmix64
		const float4* q = src + 4 * (x % nrec);
		const float4 a = q[0], b = q[1], c = q[2], d = q[3];
		acc += a.x + b.y + c.z + d.w;
	}
	if (acc == 123.456f) *sink = acc;
}
extern "C" int mipt_measure_gather_read(mipt_ctx* c, uint64_t buffer_bytes, uint64_t records, int repeats, double* gb_per_s) {
	if (!c || !gb_per_s || buffer_bytes < (1u << 20) || records < 1 || repeats < 1) return fail(c, MIPT_ERR_INVALID, "bad arguments");
	HIPCHK(c, hipSetDevice(c->device));
	float4* buf = nullptr; float* sink = nullptr;
	HIPCHK(c, hipMalloc(&buf, buffer_bytes));
	if (hipMalloc(&sink, 4) != hipSuccess) { hipFree(buf); return fail(c, MIPT_ERR_HIP, "hipMalloc failed"); }
	hipMemset(buf, 0, buffer_bytes);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const unsigned grid = (unsigned)c->n_cus * 8u;
	const int iters = (int)std::max<uint64_t>(1, records / ((uint64_t)grid * 256));
	hipLaunchKernelGGL(k_gather_read, dim3(grid), dim3(256), 0, 0, buf, (unsigned long long)(buffer_bytes / 64), std::min(iters, 8), sink);   // warm-up
	hipEventRecord(e0, 0);
	for (int r = 0; r < repeats; r++) hipLaunchKernelGGL(k_gather_read, dim3(grid), dim3(256), 0, 0, buf, (unsigned long long)(buffer_bytes / 64), iters, sink);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1); hipFree(buf); hipFree(sink);
	if (!(ms > 0.f)) return fail(c, MIPT_ERR_HIP, "timing failed");
	*gb_per_s = (double)grid * 256.0 * iters * 64.0 * repeats / (ms * 1e-3) / 1e9;
	return MIPT_OK;
}

// Dependent random fetches: every lane walks one random cycle through a table of 64-byte records, four 16-byte loads per
// step, the next index out of the record: the access pattern of a traversal step that misses the caches.  Its rate is the
// ceiling of the memory system BEHIND L2 for this pattern (tools/valu_rate.hip: the same whatever the number of waves or
// active lanes, the same for 128-byte records: a fixed rate of 128-byte line fetches); bench.py prices the traversal's L2
// misses against it.  The table is one cycle through every record, built on the device: next = (a i + c) mod 2^k has full period
// exactly when c is odd and a = 1 (mod 4) (Hull-Dobell); both constants below satisfy it (0x9E3779B1 = 1 mod 4, and odd).
__global__ void __launch_bounds__(256) k_chase_init(float4* __restrict__ tab, unsigned nrec_pow2) {
	const unsigned mask = nrec_pow2 - 1u;
	for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nrec_pow2; i += gridDim.x * blockDim.x) {
		const unsigned nxt = (i * 2654435761u + 0x9e3779b1u) & mask;                 // a = 1 mod 4, c odd: one cycle of length 2^k
		tab[4 * (size_t)i] = make_float4(__uint_as_float(nxt), 0.f, 0.f, 0.f);
		tab[4 * (size_t)i + 1] = tab[4 * (size_t)i + 2] = tab[4 * (size_t)i + 3] = make_float4(1.f, 2.f, 3.f, 4.f);
	}
}
__global__ void __launch_bounds__(256) k_chase(const float4* __restrict__ tab, unsigned nrec_pow2, int steps, unsigned* __restrict__ sink) {
	unsigned cur = ((blockIdx.x * 256u + threadIdx.x) * 2246822519u) & (nrec_pow2 - 1u);
	unsigned acc = 0;
	for (int i = 0; i < steps; i++) {
		const float4* q = tab + 4 * (size_t)cur;
		const float4 a = q[0], b = q[1], c2 = q[2], d = q[3];
		acc += __float_as_uint(a.y) ^ __float_as_uint(b.x) ^ __float_as_uint(c2.x) ^ __float_as_uint(d.x);
		cur = __float_as_uint(a.x);
	}
	if (acc == 0x12345678u) sink[0] = cur;
}
extern "C" int mipt_measure_dependent_gather(mipt_ctx* c, uint64_t table_bytes, int steps, int repeats, double* glines_per_s) {
	if (!c || !glines_per_s || table_bytes < (1u << 20) || steps < 1 || repeats < 1) return fail(c, MIPT_ERR_INVALID, "bad arguments");
	HIPCHK(c, hipSetDevice(c->device));
	unsigned nrec = 1u;
	while ((uint64_t)nrec * 2u * 64u <= table_bytes && nrec < (1u << 30)) nrec *= 2u;    // the largest power of two of 64-byte records that fits
	float4* tab = nullptr; unsigned* sink = nullptr;
	HIPCHK(c, hipMalloc(&tab, (size_t)nrec * 64));
	if (hipMalloc(&sink, 4) != hipSuccess) { hipFree(tab); return fail(c, MIPT_ERR_HIP, "hipMalloc failed"); }
	hipLaunchKernelGGL(k_chase_init, dim3((unsigned)c->n_cus * 8u), dim3(256), 0, 0, tab, nrec);
	const unsigned grid = (unsigned)c->n_cus * 4u;                 // 4 waves per SIMD: the rate does not depend on it (2 .. 8 measured)
	hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, tab, nrec, std::min(steps, 64), sink);   // warm-up
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, 0);
	for (int r = 0; r < repeats; r++) hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, tab, nrec, steps, sink);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1); hipFree(tab); hipFree(sink);
	if (!(ms > 0.f)) return fail(c, MIPT_ERR_HIP, "timing failed");
	*glines_per_s = (double)grid * 256.0 * steps * repeats / (ms * 1e-3) / 1e9;
	return MIPT_OK;
}

// What a CU charges per vector-memory wave-instruction: loads that hit L1 (every lane re-reads its own 64-byte record of a
// 256 KB table, four global_load_dwordx4 per step, eight in flight), `active` of 64 lanes, 7 waves per SIMD.  On MI355X this is
// ~10 ns per instruction and CU at 32 active lanes and ~13 ns at 64, the same for 8- and 16-byte loads and for 2 or 7 waves per
// SIMD (tools/valu_rate.hip): a rate of INSTRUCTIONS, not of bytes or lanes.  The traversal kernels issue 1.0 G of them per
// launch; bench.py prices that count against this figure (roofline.frac_vmem_issue).
__global__ void __launch_bounds__(256) k_vmem_issue(const float4* __restrict__ tab, int iters, unsigned active, float* __restrict__ sink) {
	const unsigned lane = threadIdx.x & 63u;
	const float4* p = tab + 4 * (size_t)((blockIdx.x * 256u + threadIdx.x) & 4095u);
	float acc = 0.f;
	if (lane < active) {
		for (int i = 0; i < iters; i++) {
			float4 a, b, c, d;
			asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n global_load_dwordx4 %3, %4, off offset:48\n"
			             "global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n global_load_dwordx4 %3, %4, off offset:48\n s_waitcnt vmcnt(0)"
			             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
			acc += a.x + b.x + c.x + d.x;
		}
	}
	if (acc == 12345.f) sink[0] = acc;
}
// Diagnostics of the any-hit stage (mipt_anyhit.h): shadow rays of the last render that the order-free kernel handed to the ordered one.
extern "C" int mipt_debug_anyhit_replayed(mipt_ctx* c, uint64_t* out) {
	if (!c || !out || !c->d_cnt) return fail(c, MIPT_ERR_INVALID, "no render yet");
	HIPCHK(c, hipSetDevice(c->device));
	unsigned long long v = 0;
	HIPCHK(c, hipMemcpy(&v, &c->d_cnt[0]._pad[0], sizeof v, hipMemcpyDeviceToHost));
	*out = v;
	return MIPT_OK;
}
// Which any-hit kernel the resident scene's shadow rays run on, and why.
extern "C" const char* mipt_debug_anyhit_kind(const mipt_ctx* c) {
	if (!c || !c->d_scene) return "no scene";
	if (c->anyhit_ordered_because_not_nested) return "ordered: a box of the uploaded tree does not nest in its parent's, is empty, or is not finite";
	if (!c->d_quad_nodes) return "ordered: the scene has no mesh";
	if (!c->opt_anyhit_wide) return "ordered: option anyhit_wide = 0";
	return "order-free";
}
extern "C" int mipt_measure_vmem_issue(mipt_ctx* c, int active_lanes, int iters, double* ns_per_instruction_and_cu) {
	if (!c || !ns_per_instruction_and_cu || active_lanes < 1 || active_lanes > 64 || iters < 1) return fail(c, MIPT_ERR_INVALID, "bad arguments");
	HIPCHK(c, hipSetDevice(c->device));
	float4* tab = nullptr; float* sink = nullptr;
	HIPCHK(c, hipMalloc(&tab, 4096 * 64 + 4096));
	if (hipMalloc(&sink, 4) != hipSuccess) { hipFree(tab); return fail(c, MIPT_ERR_HIP, "hipMalloc failed"); }
	hipMemset(tab, 0, 4096 * 64 + 4096);
	const int wps = 7;
	const unsigned grid = (unsigned)c->n_cus * wps;
	hipLaunchKernelGGL(k_vmem_issue, dim3(grid), dim3(256), 0, 0, tab, 16, (unsigned)active_lanes, sink);      // warm-up
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k_vmem_issue, dim3(grid), dim3(256), 0, 0, tab, iters, (unsigned)active_lanes, sink);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1); hipFree(tab); hipFree(sink);
	if (!(ms > 0.f)) return fail(c, MIPT_ERR_HIP, "timing failed");
	*ns_per_instruction_and_cu = (double)ms * 1e6 / ((double)wps * 4 * iters * 8);      // per CU: wps blocks x 4 waves x iters x 8 instructions
	return MIPT_OK;
}

