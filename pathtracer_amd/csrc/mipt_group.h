// mipt_group.h — several devices behind one context: the render across the members of a group and the reduce of their partial framebuffers (types and RCCL loader: mipt.hip, in front of mipt_create)
// (part of the one translation unit csrc/mipt.hip: included there, after the context and the render loop it uses)

#ifdef MIPT_GROUP_RENDER_PART
// Samples [k0, k1) on every member of the group, each on its share of the tiles, summed into d_accum (member 0's device) on `st`.
static int group_render_range(mipt_ctx* c, const mipt_render_params* p, int k0, int k1, float* d_accum, hipStream_t st) {
	mipt_group* g = c->group;
	const int n = (int)g->member.size();
	const size_t count = (size_t)p->W * p->H * 4, bytes = count * sizeof(float);
	const int nr = p->tile_nranks > 0 ? p->tile_nranks : 1;
	if (p->tile_rank < 0 || p->tile_rank >= nr) return fail(c, MIPT_ERR_INVALID, "tile_rank outside [0,tile_nranks)");
	std::vector<int> rcs(n, MIPT_OK);
	auto work = [&](int i) {
		mipt_ctx* m = g->member[i];
		if (hipSetDevice(m->device) != hipSuccess) { rcs[i] = fail(m, MIPT_ERR_HIP, "hipSetDevice(%d) failed", m->device); return; }
		if (g->acc_bytes[i] < bytes) {
			if (g->acc[i]) { hipFree(g->acc[i]); g->acc[i] = nullptr; g->acc_bytes[i] = 0; }
			if (hipMalloc((void**)&g->acc[i], bytes) != hipSuccess) { rcs[i] = fail(m, MIPT_ERR_HIP, "hipMalloc of a partial framebuffer failed"); return; }
			g->acc_bytes[i] = bytes;
		}
		if (hipMemsetAsync(g->acc[i], 0, bytes, g->stream[i]) != hipSuccess) { rcs[i] = fail(m, MIPT_ERR_HIP, "hipMemsetAsync failed"); return; }
		mipt_render_params q = *p;
		q.sample_begin = k0; q.sample_end = k1;
		q.tile_nranks = nr * n; q.tile_rank = p->tile_rank * n + i;      // the caller's own partition (one process per node, say) is refined by the group's
		rcs[i] = render_impl(m, &q, g->acc[i], g->stream[i], nullptr, nullptr, nullptr, nullptr);
		if (rcs[i] == MIPT_OK && hipEventRecord(g->done[i], g->stream[i]) != hipSuccess) rcs[i] = fail(m, MIPT_ERR_HIP, "hipEventRecord failed");
	};
	{
		std::vector<std::thread> th;
		for (int i = 1; i < n; i++) th.emplace_back(work, i);
		work(0);
		for (auto& t : th) t.join();
	}
	for (int i = 0; i < n; i++) if (rcs[i]) { if (i) fail(c, rcs[i], "device %d: %s", g->member[i]->device, g->member[i]->err.c_str()); hipSetDevice(c->device); return rcs[i]; }
	// the framebuffer reduce: the analogue of the per-thread buffer sum of Raytracer.cpp:1669-1685
	bool reduced = false;
	if (!g->comm.empty() && g->opt_reduce != 2) {
		RcclApi& api = rccl_api();
		int r = api.GroupStart();
		int enqueued = 0;                   // members whose ncclReduce was accepted into the group
		for (int i = 0; i < n && r == 0; i++) {
			hipSetDevice(g->member[i]->device);
			r = api.Reduce(g->acc[i], g->acc[i], count, MIPT_NCCL_FLOAT32, MIPT_NCCL_SUM, 0, g->comm[i], g->stream[i]);
			if (r == 0) enqueued++;
		}
		const int r2 = api.GroupEnd();
		hipSetDevice(c->device);
		if (r == 0 && r2 == 0) reduced = true;
		else if (g->opt_reduce == 1 || enqueued > 0)
			// part of the group may have been launched: acc[0] may hold a partial sum and a stream may sit in a collective its peers never
			// join — summing the same in-place buffers again by copies would double-count or hang, so the failure is the caller's
			return fail(c, MIPT_ERR_HIP, "ncclReduce failed (%d of %d members enqueued): %s", enqueued, n, api.GetErrorString(r ? r : r2));
		else {   // ncclGroupStart or the FIRST enqueue failed: nothing of the collective exists; this range and the following ones are summed by copies, the note says why
			g->reduce_note = std::string("copy reduce: ncclReduce failed: ") + api.GetErrorString(r ? r : r2);
			g->opt_reduce = 2;
		}
	}
	if (!reduced) {
		HIPCHK(c, hipSetDevice(c->device));
		if (g->tmp0_bytes < bytes) {
			if (g->tmp0) { hipFree(g->tmp0); g->tmp0 = nullptr; g->tmp0_bytes = 0; }
			HIPCHK(c, hipMalloc((void**)&g->tmp0, bytes));
			g->tmp0_bytes = bytes;
		}
		for (int i = 1; i < n; i++) {
			HIPCHK(c, hipStreamWaitEvent(g->stream[0], g->done[i], 0));
			HIPCHK(c, hipMemcpyPeerAsync(g->tmp0, c->device, g->acc[i], g->member[i]->device, bytes, g->stream[0]));
			// member i's partial framebuffer is cleared and refilled by the next range on ITS stream: that stream waits until
			// this copy has read it (without it a second range could zero acc[i] under the copy: ranges are only separated by a
			// host synchronisation when a progress callback is set)
			HIPCHK(c, hipEventRecord(g->copied[i], g->stream[0]));
			HIPCHK(c, hipStreamWaitEvent(g->stream[i], g->copied[i], 0));
			hipLaunchKernelGGL(k_accumulate, dim3((unsigned)c->n_cus * 4u), dim3(256), 0, g->stream[0], g->acc[0], (const float*)g->tmp0, count);
		}
	}
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipEventRecord(g->done[0], g->stream[0]));
	HIPCHK(c, hipStreamWaitEvent(st, g->done[0], 0));
	hipLaunchKernelGGL(k_accumulate, dim3((unsigned)c->n_cus * 4u), dim3(256), 0, st, d_accum, (const float*)g->acc[0], count);
	HIPCHK(c, hipGetLastError());
	// member 0's partial framebuffer is reused by the next range: its stream waits until the caller's stream has read it
	HIPCHK(c, hipEventRecord(g->done[0], st));
	HIPCHK(c, hipStreamWaitEvent(g->stream[0], g->done[0], 0));
	return MIPT_OK;
}

// mipt_render* on a group: the whole sample range at once, or — when the caller wants progress calls or may cancel — in
// chunks of about one pass per device with the reduce after each chunk, so that the caller's buffer always holds complete sums.
static int group_render(mipt_ctx* c, const mipt_render_params* p, float* d_accum, hipStream_t st, mipt_progress_cb cb, void* cb_user, volatile int* cancel) {
	if (!p) return fail(c, MIPT_ERR_INVALID, "null render params");
	if (p->W <= 0 || p->H <= 0 || p->nrays <= 0) return fail(c, MIPT_ERR_INVALID, "bad image size / sample count");
	int kb = p->sample_begin, ke = p->sample_end;
	if (kb == 0 && ke == 0) ke = p->nrays;
	if (kb < 0 || ke > p->nrays || kb > ke) return fail(c, MIPT_ERR_INVALID, "sample range outside [0,nrays]");
	if (kb == ke) return MIPT_OK;
	const int n = (int)c->group->member.size(), nr = p->tile_nranks > 0 ? p->tile_nranks : 1;
	int chunk = ke - kb;
	if (cb || cancel) {
		const int64_t px_per_device = std::max<int64_t>(64, (int64_t)p->W * p->H / ((int64_t)nr * n));
		chunk = (int)std::max<int64_t>(1, std::min<int64_t>(chunk, c->opt_paths_per_pass / px_per_device));
		if (c->opt_samples_per_pass > 0) chunk = (int)std::min<int64_t>(chunk, c->opt_samples_per_pass);
	}
	for (int k0 = kb; k0 < ke; k0 += chunk) {
		if (cancel && *cancel) { hipStreamSynchronize(st); return fail(c, MIPT_ERR_CANCELLED, "cancelled"); }
		const int k1 = std::min(ke, k0 + chunk);
		int rc = group_render_range(c, p, k0, k1, d_accum, st);
		if (rc) return rc;
		if (cb) { hipStreamSynchronize(st); cb(cb_user, k1 - kb, ke - kb); }
	}
	return MIPT_OK;
}

#else
// How a group sums its partial framebuffers ("RCCL ncclReduce(sum, fp32, root 0)" or "copy reduce: <why>"); "" for a single device.
extern "C" const char* mipt_group_reduce_kind(const mipt_ctx* c) { return (c && c->group) ? c->group->reduce_note.c_str() : ""; }
extern "C" int mipt_group_size(const mipt_ctx* c) { return !c ? 0 : (c->group ? (int)c->group->member.size() : 1); }

// Loads RCCL and runs one single-rank ncclReduce on the context's device: checks, on a box with one GPU, that the library
// the group path depends on can be loaded and called with the signatures used here.
extern "C" int mipt_rccl_selftest(mipt_ctx* c) {
	if (!c) return MIPT_ERR_INVALID;
	RcclApi& api = rccl_api();
	if (!api.lib) return fail(c, MIPT_ERR_UNSUPPORTED, "%s", api.why.c_str());
	HIPCHK(c, hipSetDevice(c->device));
	mipt_nccl_comm cm = nullptr;
	int dev = c->device;
	int r = api.CommInitAll(&cm, 1, &dev);
	if (r != 0) return fail(c, MIPT_ERR_HIP, "ncclCommInitAll: %s", api.GetErrorString(r));
	const size_t count = 1 << 20;
	float *a = nullptr, *b = nullptr;
	hipStream_t st = nullptr;
	int rc = MIPT_OK;
	std::vector<float> h(count), back(count);
	for (size_t i = 0; i < count; i++) h[i] = (float)(i % 977) * 0.25f;
	if (hipMalloc((void**)&a, count * 4) != hipSuccess || hipMalloc((void**)&b, count * 4) != hipSuccess || hipStreamCreate(&st) != hipSuccess) rc = fail(c, MIPT_ERR_HIP, "allocation failed");
	if (!rc && (hipMemcpy(a, h.data(), count * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(b, 0, count * 4) != hipSuccess)) rc = fail(c, MIPT_ERR_HIP, "copy failed");
	if (!rc) {
		api.GroupStart();
		r = api.Reduce(a, b, count, MIPT_NCCL_FLOAT32, MIPT_NCCL_SUM, 0, cm, st);
		const int r2 = api.GroupEnd();
		if (r != 0 || r2 != 0) rc = fail(c, MIPT_ERR_HIP, "ncclReduce: %s", api.GetErrorString(r ? r : r2));
	}
	if (!rc && (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(back.data(), b, count * 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = fail(c, MIPT_ERR_HIP, "reduce did not complete");
	if (!rc && memcmp(h.data(), back.data(), count * 4) != 0) rc = fail(c, MIPT_ERR_HIP, "single-rank ncclReduce returned other bytes");
	if (a) hipFree(a);
	if (b) hipFree(b);
	if (st) hipStreamDestroy(st);
	api.CommDestroy(cm);
	return rc;
}

#endif
