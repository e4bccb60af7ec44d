#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): the configuration BASELINE.json's metric is quoted on ("1080p x 1024 spp",
"the 2.5M-triangle scene") = configs[2]: 2 508 800-triangle blob with a 2048x2048 Kd texture and a
4096x2048 environment map, Phong BRDF, 1920x1080, depth 4, default loadScene() light and camera; it
fits one GPU.  --workload c1 / c3 / c4 select the other configs (c1 = 133 128-triangle diffuse blob).
One STEP = the hot path (camera rays -> getColor -> splat) over the whole frame at --spp-per-step samples per pixel.
With N GPUs the frame's 32x32-pixel tiles are dealt round-robin to the ranks (one process per GPU, scene replicated) and
the per-rank accumulators are summed by ONE all-reduce at the end (RCCL).
  --scaling strong (default)  the job is fixed: a step is the metric's frame, 1024 spp at 1080p (256 at 4K), whatever N is;
        a rank renders all samples of its 1/N of the tiles.  The library cuts a step into passes of at most 2^30 paths
        (~172 GB of path state, sized for 288 GB of HBM): 2 passes of 512 spp on one GPU, one pass of 1024 spp on its
        eighth of the frame on each of 8 — every pass stays large (the per-launch ramp and drain of the persistent
        kernels is amortised over a large batch: 3.72 Grays/s at 64 spp per pass, 3.82 at 128, 3.94 at 256, 3.98 at 512).
  --scaling weak              the job grows with N: a step renders N x 256 spp, so every rank keeps one 531 M-path pass per
        step as in the single-GPU run.
At N > 1 the line of the other mode (two steps, measured after the timed region) is attached as "other_scaling".
--in-process N drives N devices from ONE process through mipt_create(device_ids, N) (the C++ host's way: worker threads and
the RCCL reduce inside the library) instead of one process per GPU; "0,0" lists a device twice (one-GPU functional test).

value = rays (closest-hit + shadow, counted like the oracle counts them) of all ranks / wall time
of the K timed steps (+ the final reduce), inputs resident in HBM, barrier + synchronize on both
sides, max over ranks.

roofline (dominant kernel = the closest-hit traversal, k_wf_traverse<0>; its mean launch duration is measured here with
HIP events on the render stream).  Every fraction names its denominator:
  frac  (bound "hbm")   SURVEY.md 8(d) / the task contract: achieved = ALGORITHMIC bytes per launch / duration, peak = HBM 8 TB/s.
                        Bytes per ray come from the CPU oracle's counters of the reference's ordered traversal on a bounded sample of
                        the same scene and camera (B_ray = 24*n_box + 8*n_node + 64*n_tri), times the rays one launch casts.  It
                        EXCEEDS 1 (frac_note): most node fetches are served by L1 / L2 / the Infinity Cache and never cross HBM.
  traffic, frac_hbm_measured   HBM bytes per launch from the PMC counters (FETCH_SIZE corrected with the gather factor of
                        tools/fetch_calibration.py, + WRITE_SIZE), and those / duration / 8 TB/s
  issue_model           the ONE account of what the kernel's time follows: busy share of a CU's vector-memory path
                        (wave-instructions per CU x 11.5 ns) and of a SIMD's vector pipe (instructions x 1.7 ns), constants fitted on seven
                        profiled builds (profiles/r5_traversal_time_model.txt, tests/test_time_model.py), instruction counts from the PMC run
  device_rates_measured_in_this_run   raw rates mipt_measure_* takes on this device (ns per vector-memory wave-instruction and CU at the
                        kernel's lane count; dependent random line fetches per ns): for cross-checking the constants, no fraction is formed
Per-ray counter values come from the committed PMC run of the same workload (profiles/pmc_counters.json; separate --pmc passes).
derived_from_pmc_run.same_library_build compares __graft_entry__.source_hash() (csrc/* + include/mipt.h + hipcc flags) with the hash
that run recorded; when they differ everything derived from the counters is null and derived_from_pmc_run.stale is true.
roofline_shade_kernel: algorithmic path-state bytes per vertex x vertices / stage time against the HBM peak, and its measured traffic.

cpu_baseline (rank 0, N=1 only): the compiled reference's own render_image_nopreviz() on all host
cores when oracle/_ref/libptref.so is present (kind "reference"), otherwise the oracle's threaded
restatement (kind "port"), on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPP_PER_PASS = 256          # a quarter of the step at 1080p (the library itself cuts a step into passes of at most 2^30 paths, ~172 GB of path state); scaled down with the pixel count
SEED_STRIDE = 65536         # sample k of pixel p draws from pcg32(p * 65536 + k): a run may use at most 65536 samples per pixel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c1", "c1g", "c2", "c3", "c4"], help="BASELINE.json configs[1..4]")
    ap.add_argument("--spp-per-step", type=int, default=0, help="samples per pixel per step (default: strong 1024 at 1080p / 256 at 4K; weak 256 / 64 per rank)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="strong: the job per step is fixed (default); weak: it grows with the number of GPUs")
    ap.add_argument("--in-process", default="", help="device list, e.g. 0,1,2,3: ONE process drives them through mipt_create(ids, n) (world size must be 1)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--grid", type=int, default=None, help="override the blob tessellation n (2 n^2 triangles)")
    ap.add_argument("--pipeline", type=int, default=-1, help="-1 = library default")
    ap.add_argument("--opt", action="append", default=[], help="library tunable name=value (repeatable)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for functional tests)")
    ap.add_argument("--share-gpu", action="store_true", help="functional test: all ranks use GPU 0 (needs --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc", action="store_true", help="profiling run: skip the CPU legs")
    return ap.parse_args()


def oracle_bytes_per_ray(mesh, mat, cfg_full):
    """Mean algorithmic bytes per closest-hit / shadow ray on a bounded sample (same scene, same
    camera, 1/8 resolution, 2 spp), from the oracle's counters of the reference traversal."""
    import copy
    from oracle.binding import Oracle
    cfg = copy.copy(cfg_full)
    cfg.W, cfg.H, cfg.spp = max(8, cfg_full.W // 8), max(8, cfg_full.H // 8), 2
    from pathtracer_amd import scenes
    O = Oracle()
    O.apply_config(cfg)
    scenes.install(O, mesh, mat)
    O.prepare()
    O.counters_reset()
    t, img, cnt, rays = O.render_omp(O.cdll.o_max_threads())
    c = O.counters().astype(float)
    b_closest = (24 * c[0] + 8 * c[1] + 64 * c[2]) / max(1.0, float(rays[0]))
    b_shadow = (24 * c[3] + 8 * c[4] + 64 * c[5]) / max(1.0, float(rays[1]))
    # one 64-byte line per inner node visited (it holds both children's boxes: two box tests) and one per triangle record
    l_closest = (c[0] / 2 + c[2]) / max(1.0, float(rays[0]))
    l_shadow = (c[3] / 2 + c[5]) / max(1.0, float(rays[1]))
    return dict(bytes_closest=b_closest, bytes_shadow=b_shadow, lines_closest=l_closest, lines_shadow=l_shadow, rays_per_path=float(rays[0] + rays[1]) / (cfg.W * cfg.H * cfg.spp),
                sample=f"{cfg.W}x{cfg.H}x{cfg.spp}spp")


def cpu_baseline(mesh, mat, cfg_full, rays_per_path):
    import copy
    from oracle import binding
    from pathtracer_amd import scenes
    cfg = copy.copy(cfg_full)
    cfg.W, cfg.H = cfg_full.W // 4, cfg_full.H // 4
    cores = min(64, os.cpu_count() or 1)   # the reference supports at most 64 OpenMP threads
    # size the sample for roughly 15 s of CPU work: ~0.15 Mpaths/s per thread measured on this scene
    est_rate = 0.15e6 * cores
    cfg.spp = int(max(2, min(4096, 15.0 * est_rate / (cfg.W * cfg.H))))
    if binding.ref_available():
        R = binding.Ref()
        R.apply_config(cfg)
        scenes.install(R, mesh, mat)
        threads = R.max_threads()
        secs, _ = R.time_render_nopreviz(threads)
        kind = "reference"
    else:
        O = binding.Oracle()
        O.apply_config(cfg)
        scenes.install(O, mesh, mat)
        O.prepare()
        threads = O.cdll.o_max_threads()
        secs, _, _, _ = O.render_omp(threads)
        kind = "port"
    mpaths = cfg.W * cfg.H * cfg.spp / secs / 1e6
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=mpaths * rays_per_path, unit="Mrays/s", cores=int(threads), kind=kind, cpu_model=model, host_logical_cpus=os.cpu_count(),
                sample=f"{cfg.W}x{cfg.H}x{cfg.spp}spp of the same scene/camera/depth, {secs:.1f}s wall; "
                       f"{mpaths:.3f} Mpaths/s x {rays_per_path:.2f} rays/path (oracle count)",
                mpaths_per_s=mpaths)


# The headline fraction, FROZEN (it had meant four different things in four rounds).  tests/test_bench_contract.py asserts the text.
ROOFLINE_FRAC_DEFINITION = ("SURVEY 8(d): algorithmic bytes per launch (24 n_box + 8 n_node + 64 n_tri from the oracle's counters of the reference's ordered traversal, "
                            "x the rays of one launch) / mean launch time (HIP events on the render stream) / HBM peak 8 TB/s (MI355X_MICROARCH.md)")
# What the traversal kernels' time follows (fitted on seven builds of rounds 4 and 5, profiles/r5_traversal_time_model.txt):
# a compute unit issues one vector-memory wave-instruction per ~11.5 ns, a SIMD one vector instruction per ~1.7 ns
TA_NS_PER_VMEM_INSTRUCTION = 11.5
SIMD_NS_PER_VALU_INSTRUCTION = 1.7
PMC_COUNTERS = "pmc_counters.json"      # profiles/: per-ray counters of the traversal kernels of the build named inside (tools/pmc_to_json.py)


def assert_physical(out):
    """No fraction that claims to be MEASURED HBM traffic may exceed what the device can stream (ADVICE / VERDICT r5: the line once said 1.16)."""
    for key in ("roofline", "roofline_shadow_kernel", "roofline_shade_kernel"):
        f = (out.get(key) or {}).get("frac_hbm_measured")
        if f is not None and not (0.0 <= f <= 1.0):
            raise SystemExit("bench.py: %s.frac_hbm_measured = %.3f is not a physical HBM fraction: refusing to print the line" % (key, f))


STAGE = {"name": "start", "reduce": None, "n_gpus": None}      # where a failing run was, for the diagnostic line below


def stage(name, **kw):
    STAGE["name"] = name
    STAGE.update(kw)


def main():
    args = parse()
    STAGE["n_gpus"] = args.gpus
    try:
        run(args)
    except SystemExit as e:
        # a refusal or a failed check: the text is the message; still leave ONE JSON line that says where the run stopped, so that a failed
        # multi-GPU lease tells which of its paths broke (value null: nothing was measured).  Exit code stays non-zero.
        if e.code not in (0, None) and int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(failure_line(args, str(e.code))))
        raise
    except Exception as e:      # noqa: BLE001 — every failure of the product path lands here: library errors (capi.MiptError), RCCL, torch.distributed
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(failure_line(args, "%s: %s" % (type(e).__name__, e))))
        raise SystemExit("bench.py failed in stage '%s': %s: %s" % (STAGE["name"], type(e).__name__, e))


def failure_line(args, text):
    return {"metric": "Msamples/s (primary+secondary rays) at 1080p\u00d71024spp; 1/2/4/8-GPU scaling", "value": None, "unit": "Mrays/s", "n_gpus": STAGE["n_gpus"],
            "steps": args.steps, "warmup": args.warmup, "error": text[-1500:], "stage": STAGE["name"], "reduce": STAGE["reduce"],
            "stages_in_order": ["devices", "process group (torch.distributed / RCCL)", "build", "mipt_create (group: ncclCommInitAll)", "scene: TriMesh::init on the device",
                                "upload (group: hipMemcpyPeer replication)", "warm-up steps", "timed steps", "framebuffer reduce", "checks"]}


def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ      # under torch.distributed.run: one process per GPU
    in_process = [int(x) for x in args.in_process.split(",")] if args.in_process else []
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if in_process and world > 1:
        raise SystemExit("--in-process drives its devices from one process: do not launch it under torch.distributed.run")
    if launched and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks: they must agree")
    if not launched and not in_process and args.gpus > 1:
        # the plain command line: ONE process drives the N devices through the product's own multi-GPU entry,
        # mipt_create(ids, N) -> worker thread + stream + partial framebuffer per device, one ncclReduce (RCCL) to device 0
        in_process = list(range(args.gpus))
    if in_process and len(in_process) != args.gpus:
        if args.gpus != 1:
            raise SystemExit(f"--gpus {args.gpus} but --in-process lists {len(in_process)} devices")
        args.gpus = len(in_process)
    # torch.cuda.device_count() does not initialise the GPU
    stage("devices")
    n_dev = torch.cuda.device_count()
    need = (max(in_process) + 1) if in_process else (local_rank + 1 if launched and not args.share_gpu else 1)
    if n_dev < need:
        raise SystemExit(f"bench.py --gpus {args.gpus} needs {need} visible device(s), this box has {n_dev}: refusing to measure fewer GPUs than asked for")
    if args.share_gpu:
        local_rank = 0
    if in_process:
        local_rank = in_process[0]                       # the caller's accumulator and stream live on the group's first device
    if world > 1:
        stage("process group (torch.distributed / RCCL)", reduce=args.backend + " all_reduce")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    on_device = world == 1 or args.backend == "nccl"      # gloo reduces host copies
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as ge
    stage("build")
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from pathtracer_amd import capi, scenes

    job_gpus = len(in_process) if in_process else world
    assert job_gpus == args.gpus, (job_gpus, args.gpus)
    dims = {"c4": (3840, 2160)}.get(args.workload, (1920, 1080))
    npx = (args.width or dims[0]) * (args.height or dims[1])
    spp_pass = SPP_PER_PASS
    while spp_pass > 1 and npx * spp_pass > (1 << 29) + (1 << 24):
        spp_pass //= 2

    def samples_per_step(scaling):
        if args.spp_per_step > 0:
            return args.spp_per_step * (job_gpus if scaling == "weak" else 1)
        return spp_pass * job_gpus if scaling == "weak" else 4 * spp_pass      # strong: the metric's frame (1024 spp at 1080p)

    SPS = samples_per_step(args.scaling)
    other = "weak" if args.scaling == "strong" else "strong"
    OTHER_STEPS = 2 if job_gpus > 1 else 0
    total_spp = SPS * (args.steps + args.warmup) + samples_per_step(other) * (OTHER_STEPS + 1)
    if total_spp > SEED_STRIDE:
        raise SystemExit(f"{total_spp} samples per pixel in this run, but the streams of neighbouring pixels are {SEED_STRIDE} apart: use fewer steps")
    mesh, cfg, mat, wl_text = scenes.workload(args.workload, args.width, args.height, total_spp, args.grid)
    args.width, args.height = cfg.W, cfg.H

    stage("mipt_create (group: ncclCommInitAll)")
    rt = capi.HostRaytracer(device=in_process if in_process else local_rank)
    if in_process:
        STAGE["reduce"] = rt.group_reduce_kind()
    if in_process and len(set(in_process)) == len(in_process) and len(in_process) > 1:
        # distinct devices: the group's framebuffer reduce must be RCCL's ncclReduce or the run fails (option reduce = 1 keeps an RCCL
        # failure instead of falling back to peer copies: a first multi-GPU lease cannot silently measure the copy reduce)
        rt.set_option("reduce", 1)
    rt.apply_config(cfg)
    # tiles of the partition: 32 x 32 pixels dealt round-robin (the library's default; since the splat of a rank packs the columns it reaches,
    # larger tiles buy nothing: all ranks probed on one GPU predict 7.69 at 8 GPUs on configs[4], 7.50 with the 64-pixel tiles used before)
    tile = 32
    rt.set_partition(tile, rank, world)
    stage("scene: TriMesh::init on the device")
    t0 = time.time()
    mesh_obj = rt.add_mesh(mesh)                 # TriMesh::init: axis swap, BVH (on the GPU), triangle soup, tangents
    t_build = time.time() - t0
    scenes.install_material(rt, mesh_obj, mat)   # material lists, textures, environment map
    bvh_who, bvh_s, bvh_dev_s = rt.mesh_bvh_builder(mesh_obj)
    stage("upload (group: hipMemcpyPeer replication)")
    t0 = time.time()
    rt.prepare()
    t_prepare = time.time() - t0
    if args.pipeline >= 0:
        rt.set_option("pipeline", args.pipeline)
    for kv in args.opt:
        k, v = kv.split("=")
        rt.set_option(k, int(v))

    import ctypes as C
    accum = torch.zeros(args.width * args.height * 4, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    P = rt.params       # live view of the host mirror's mipt_render_params; patched per step
    assert P.W == args.width and P.nrays == total_spp and P.seed_stride == SEED_STRIDE and P.tile_nranks == world
    assert total_spp <= P.seed_stride

    def step(s, sps=None, base=0):
        sps = sps or SPS
        P.sample_begin, P.sample_end = base + s * sps, base + (s + 1) * sps
        rt.render_device(accum.data_ptr(), stream)
        st = rt.stats()    # synchronises; cheap next to a step
        return st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    stage("warm-up steps")
    for s in range(args.warmup):
        step(s)
    sync()
    if in_process:
        STAGE["reduce"] = rt.group_reduce_kind()
    stage("timed steps")
    t0 = time.perf_counter()
    rays = paths = 0
    kern_ms = sh_ms = shade_ms = resolve_ms = 0.0
    launches = sh_launches = 0
    rays_c = rays_s = 0
    pipeline = -1
    merged = 0
    for s in range(args.warmup, args.warmup + args.steps):
        st = step(s)
        rays_c += st["rays_closest"]; rays_s += st["rays_shadow"]; paths += st["paths"]
        kern_ms += st["traverse_ms"]; launches += st["traverse_launches"]
        sh_ms += st["shadow_ms"]; sh_launches += st["shadow_launches"]; shade_ms += st["shade_ms"]; resolve_ms += st["resolve_ms"]
        pipeline = st["pipeline"]; merged = st["traverse_merged"]
    def reduce_frame():
        if world > 1:                                    # the framebuffer reduce (RCCL over xGMI); --in-process: done inside the library
            if on_device:
                dist.all_reduce(accum, op=dist.ReduceOp.SUM)
            else:
                host = accum.cpu(); dist.all_reduce(host, op=dist.ReduceOp.SUM); accum.copy_(host)

    stage("framebuffer reduce")
    reduce_frame()
    sync()
    elapsed = time.perf_counter() - t0
    stage("checks")
    other_line = None
    if OTHER_STEPS:                                      # the other scaling mode, beside the timed region
        o_sps, base = samples_per_step(other), SPS * (args.steps + args.warmup)
        accum2 = accum.clone()
        step(0, o_sps, base)
        sync()
        t1 = time.perf_counter()
        o_rays = 0
        for s2 in range(1, OTHER_STEPS + 1):
            st = step(s2, o_sps, base)
            o_rays += st["rays_closest"] + st["rays_shadow"]
        reduce_frame()
        sync()
        o_el = time.perf_counter() - t1
        tt = torch.tensor([o_el, float(o_rays)], dtype=torch.float64, device=dev if on_device else "cpu")
        if world > 1:
            tm = tt.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ts = tt.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
            o_el, o_rays = float(tm[0]), float(ts[1])
        other_line = {"scaling": other, "value": o_rays / o_el / 1e6, "unit": "Mrays/s", "steps": OTHER_STEPS, "spp_per_step": o_sps, "ms_per_step": 1e3 * o_el / OTHER_STEPS}
        accum = accum2
    t = torch.tensor([elapsed, float(rays_c), float(rays_s), float(paths), kern_ms, float(launches)], dtype=torch.float64, device=dev if on_device else "cpu")
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0]); rays_c, rays_s, paths = float(tsum[1]), float(tsum[2]), float(tsum[3])
    rays = rays_c + rays_s

    if in_process and len(set(in_process)) == len(in_process) and len(in_process) > 1 and not rt.group_reduce_kind().startswith("RCCL"):
        raise SystemExit("bench.py --gpus %d on distinct devices must reduce with RCCL, the library reports: %s" % (args.gpus, rt.group_reduce_kind()))
    if rank == 0:
        img = accum[: args.width * args.height * 3]
        finite = bool(torch.isfinite(img).all().item())
        out = {
            "metric": "Msamples/s (primary+secondary rays) at 1080p\u00d71024spp; 1/2/4/8-GPU scaling",
            "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": job_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, args.steps),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl_text}, {args.width}x{args.height}, {SPS * args.steps} spp timed ({SPS} spp/step), depth {cfg.nb_bounces}",
                       "parallelism": f"tiles{tile}x{job_gpus}" + (f", one process, mipt_create(n={rt.group_size()}) on devices {in_process}: {rt.group_reduce_kind()}" if in_process else (f", one process per GPU, {world} ranks, {args.backend} all-reduce of the framebuffers" + (" (RCCL)" if args.backend == "nccl" else "") if world > 1 else "")),
                       "ranks_in_reduce": (rt.group_size() if in_process else world), "reduce": (rt.group_reduce_kind() if in_process else (args.backend + " all_reduce" if world > 1 else "none")),
                       "pipeline": int(pipeline)},
            "mpaths_per_s": paths / elapsed / 1e6, "rays_per_path": rays / max(1.0, paths),
            "host_bvh_build_s": t_build,   # TriMesh::init as a whole (axis swap, BVH, triangle soup, tangents)
            "bvh_build": {"builder": bvh_who, "seconds": round(bvh_s, 4), "device_seconds": round(bvh_dev_s, 4), "triangles": int(mesh.ntri)},
            "prepare_s": t_prepare, "finite": finite,
            "launch_stats": {"rays_closest": rays_c, "rays_shadow": rays_s, "extend_launches": int(launches), "shadow_launches": int(sh_launches)},
        }
        if other_line:
            out["other_scaling"] = other_line
        if world == 1 and args.pmc and pipeline == 1:
            out["stage_ms_per_step"] = {("traverse" if merged else "extend"): kern_ms / args.steps, "shadow": sh_ms / args.steps, "generate+shade": shade_ms / args.steps, "resolve": resolve_ms / args.steps}
        if world == 1 and not in_process and not args.pmc:
            ob = oracle_bytes_per_ray(mesh, mat, cfg)
            my_launches = max(1, launches)
            ms_per_launch = kern_ms / my_launches
            if pipeline == 0:     # one kernel casts both kinds of rays
                kernel = "k_render_paths"
                bytes_per_launch = (rays_c * ob["bytes_closest"] + rays_s * ob["bytes_shadow"]) / my_launches
            elif merged:          # dominant kernel = the traversal kernel, whose launches serve closest-hit and any-hit queues
                kernel = "k_wf_traverse (closest-hit and any-hit queues; shadow(b) + extend(b+1) share a launch)"
                bytes_per_launch = (rays_c * ob["bytes_closest"] + rays_s * ob["bytes_shadow"]) / my_launches
            else:                 # dominant kernel = closest-hit traversal
                kernel = "k_wf_traverse<0> (closest-hit / extend stage)"
                bytes_per_launch = rays_c * ob["bytes_closest"] / my_launches
            alg_hbm = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
            rays_per_launch = (rays_c if (pipeline and not merged) else rays_c + rays_s) / my_launches
            try:
                stream = rt.measure_stream_read(8 << 30, 5)      # achievable read bandwidth of this device
            except Exception:
                stream = None
            n_cus = 256
            clock_ghz = 2.4
            try:
                props = torch.cuda.get_device_properties(dev)
                n_cus = props.multi_processor_count
                clock_ghz = getattr(props, "clock_rate", 2400000) / 1e6 or 2.4
            except Exception:
                pass
            # ---- the roofline object (task contract / SURVEY 8d): achieved = ALGORITHMIC bytes per launch / launch time, peak = HBM 8 TB/s,
            # traffic = HBM bytes per launch from the PMC counters.  Everything else names its own denominator; what is scaled from the committed PMC run
            # of the same workload (profiles/pmc_counters.json: per ray of that run x the rays of this one) is dropped to null and
            # flagged stale when this run's library is not the build that was profiled (hash of csrc/* + flags, __graft_entry__.source_hash).
            scene_bytes = int(mesh.ntri) * 64 + int(mesh.ntri) * 64          # ~ one fat node per triangle pair + one record per triangle
            try:
                cap_g = rt.measure_dependent_gather(max(64 << 20, scene_bytes), 2000, 2)      # 10^9 line fetches per second
            except Exception:
                cap_g = None
            peak_l1 = n_cus * 64 * clock_ghz                            # GB/s: one 64-byte line lookup per CU and cycle
            secs = ms_per_launch * 1e-3
            rf = {"bound": "hbm", "achieved": alg_hbm, "peak": 8000.0, "unit": "GB/s", "frac": alg_hbm / 8000.0, "traffic": None,
                  "frac_definition": ROOFLINE_FRAC_DEFINITION,
                  "frac_note": "a value > 1 means cache-served: most node fetches hit L1 / L2 / the Infinity Cache and never cross HBM; the bytes that do are `traffic` "
                               "(frac_hbm_measured = traffic / launch time / 8 TB/s)",
                  "kernel": kernel, "ms_per_launch": ms_per_launch, "launches": int(launches), "rays_per_launch": rays_per_launch,
                  "algorithmic_bytes_per_launch": bytes_per_launch, "frac_algorithmic_hbm": alg_hbm / 8000.0,
                  "peak_measured_stream_read": stream,
                  "bytes_per_closest_ray": ob["bytes_closest"], "bytes_per_shadow_ray": ob["bytes_shadow"],
                  "l1_lookups_per_ray_algorithmic": ob["lines_closest"], "oracle_sample": ob["sample"],
                  "frac_hbm_measured": None}
            ceil = {}
            pmc_file = os.path.join(ROOT, "profiles", PMC_COUNTERS)
            pk = None
            try:   # per-ray counter values of the dominant kernel from the committed PMC run of the same workload
                pall = json.load(open(pmc_file))
                pj = pall[args.workload]
                pk = pj["kernels"]["k_wf_traverse<2>" if merged else ("k_wf_traverse<0>" if pipeline == 1 else "k_render_paths")]
                same = pall.get("_build", {}).get("source_sha256_16") == ge.source_hash() and not os.environ.get("MIPT_LIB_OVERRIDE")
                rf["derived_from_pmc_run"] = {"file": "profiles/pmc_counters.json", "source": pj["source"], "git_commit": pall.get("_build", {}).get("git_commit"),
                                              "same_library_build": same, "stale": not same,
                                              "keyed_on": "sha256 of pathtracer_amd/csrc/*, include/mipt.h and the hipcc flags (the binary is not bit-reproducible)"}
                if not same:
                    pk = None
            except Exception as e:
                rf["pmc_note"] = "profiles/pmc_counters.json has no entry for this workload / kernel (%s: %s)" % (type(e).__name__, e)
            lanes_per_instr = 32
            if pk:
                rf["traffic"] = pk["hbm_bytes_per_ray"] * rays_per_launch
                rf["frac_hbm_measured"] = rf["traffic"] / secs / 8e12
                rf["hbm_traffic_over_algorithmic_bytes"] = rf["traffic"] / bytes_per_launch
                rf["l2_misses_per_ray"] = pk["l2_misses_per_ray"]
                rf["l1_lookups_per_ray_measured"] = pk["tcp_accesses_per_ray"]
                # What the kernel's time follows: the 28 waves of a CU circulate between its texture-address unit (one
                # vector-memory wave-instruction per ~11.5 ns, whatever its width and almost whatever its lanes), their SIMD's vector pipe (~1.7 ns
                # per instruction) and a fixed latency per traversal step; the busy shares of the two servers:
                rf["issue_model"] = {"vmem_issue_busy": pk["vmem_per_ray"] * rays_per_launch / n_cus * TA_NS_PER_VMEM_INSTRUCTION * 1e-9 / secs,
                                     "valu_issue_busy": pk["valu_per_ray"] * rays_per_launch / (4 * n_cus) * SIMD_NS_PER_VALU_INSTRUCTION * 1e-9 / secs,
                                     "salu_issue_busy": 4 * pk["salu_per_ray"] * rays_per_launch / (4 * n_cus * clock_ghz * 1e9 * secs),
                                     "vmem_instructions_per_ray": pk["vmem_per_ray"], "valu_instructions_per_ray": pk["valu_per_ray"],
                                     "denominator": "issue time of one CU's vector-memory path at %.1f ns per wave-instruction / of a SIMD's vector pipe at %.1f ns per instruction (constants fitted on seven kernel builds, profiles/r5_traversal_time_model.txt); scalar: 4 cycles per instruction" % (TA_NS_PER_VMEM_INSTRUCTION, SIMD_NS_PER_VALU_INSTRUCTION),
                                     "active_lanes_per_vector_instruction": pk.get("active_lanes_per_vector_instruction"), "wait_share_of_wave_cycles": pk.get("wait_share_of_wave_cycles")}
                if pk.get("active_lanes_per_vector_instruction"):
                    lanes_per_instr = int(max(1, min(64, round(pk["active_lanes_per_vector_instruction"]))))
                try:   # cross-check of the model's first constant on THIS device: the cost of a vector-memory wave-instruction per CU at the kernel's mean number of active lanes
                    ceil["vmem_ns_per_wave_instruction_and_cu"] = rt.measure_vmem_issue(lanes_per_instr, 3000)
                    ceil["vmem_measured_at_active_lanes"] = lanes_per_instr
                except Exception as e:
                    ceil["vmem_note"] = "%s: %s" % (type(e).__name__, e)
            if cap_g:
                ceil["dependent_random_line_fetches_per_ns"] = cap_g
                ceil["dependent_fetch_table_mb"] = max(64 << 20, scene_bytes) >> 20
            ceil["source"] = ("measured in this run by mipt_measure_vmem_issue / mipt_measure_dependent_gather (csrc/mipt_measure.h): raw rates of the device, reported for "
                              "cross-checking the constants of issue_model; no fraction is formed from them (one account of the kernel's time: issue_model)")
            rf["device_rates_measured_in_this_run"] = ceil
            out["roofline"] = rf
            if pipeline == 1 and sh_launches:
                sh_alg = rays_s * ob["bytes_shadow"] / (sh_ms * 1e-3) / 1e9
                rs = {"kernel": "k_wf_anyhit (any-hit / shadow stage: order-free four-wide traversal, csrc/mipt_anyhit.h) + its ordered replay", "frac_algorithmic_hbm": sh_alg / 8000.0, "ms_per_launch": sh_ms / sh_launches,
                      "launches": int(sh_launches), "rays_per_launch": rays_s / sh_launches, "l1_lookups_per_ray_algorithmic": ob["lines_shadow"]}
                try:
                    if not pk: raise KeyError("no current PMC run")
                    pks = json.load(open(pmc_file))[args.workload]["kernels"]["k_wf_anyhit"]
                    secs_s = sh_ms / sh_launches * 1e-3
                    rs["l2_misses_per_ray"] = pks["l2_misses_per_ray"]
                    rs["vmem_issue_busy"] = pks["vmem_per_ray"] * rays_s / sh_launches / n_cus * TA_NS_PER_VMEM_INSTRUCTION * 1e-9 / secs_s
                    rs["vmem_instructions_per_ray"] = pks["vmem_per_ray"]
                    rs["frac_hbm_measured"] = pks["hbm_bytes_per_ray"] * rays_s / sh_launches / secs_s / 8e12
                except Exception:
                    pass
                out["roofline_shadow_kernel"] = rs
            if pipeline == 1 and shade_ms > 0:
                # the shade stage: algorithmic path-state bytes per vertex (what a vertex must read and write: ray 32, weight 16, engine 8, hit 16,
                # colour 16 in; ray 32, weight 16, engine 8, shadow request 48, colour 16, queue entries 8 out = 88 + 128, plus one 64-byte
                # shading record) x vertices / stage time, against the HBM peak
                verts = rays_c                                            # one shade vertex per closest-hit ray
                state_bytes = 88 + 128 + 64
                shade_secs = (shade_ms - 0.0) * 1e-3
                rsh = {"kernel": "k_wf_generate + k_wf_shade<fast tier, general tier> (per step)", "bound": "hbm", "unit": "GB/s", "peak": 8000.0, "algorithmic_bytes_per_vertex": state_bytes,
                       "frac_definition": "algorithmic path-state bytes per vertex x vertices / stage time / HBM peak 8 TB/s",
                       "vertices_per_step": verts / args.steps, "ms_per_step": shade_ms / args.steps,
                       "achieved": verts * state_bytes / shade_secs / 1e9, "frac": verts * state_bytes / shade_secs / 8e12}
                try:
                    if not pk: raise KeyError("no current PMC run")
                    pks = json.load(open(pmc_file))[args.workload]["kernels"]
                    # tools/pmc_to_json.py charges every build of the stage with ITS OWN launches per step (the depth-0 builds k_wf_shade<tier>[depth0]
                    # run once per pass, the others nb_bounces - 1 times: until round 6 both were multiplied by all of the stage's launches, which
                    # printed 1.16) and divides by the shade vertices of the profiled step; here: x the vertices of this run
                    per_step = json.load(open(pmc_file))[args.workload]["stage_generate_shade"]["hbm_bytes_per_vertex"] * verts / args.steps
                    rsh["traffic"] = per_step
                    rsh["frac_hbm_measured"] = per_step * args.steps / shade_secs / 8e12
                    rsh["traffic_over_algorithmic_bytes"] = per_step * args.steps / (verts * state_bytes)
                except Exception:
                    pass
                # Scenes whose shade stage is fp64 arithmetic (the measured-BRDF tiers of configs[4]: glibc's acos / atan2 / sincos restated, 61 correctly rounded
                # divisions): the bytes above do not describe it.  From the committed counter run of the same build (tools/pmc_fp64.sh -> profiles/fp64_counters.json):
                # fp64 operations per shade vertex x the vertices of this run / stage time, against the fp64 vector peak.
                try:
                    fall = json.load(open(os.path.join(ROOT, "profiles", "fp64_counters.json")))
                    fj = fall[args.workload]
                    same64 = fall.get("_build", {}).get("source_sha256_16") == ge.source_hash() and not os.environ.get("MIPT_LIB_OVERRIDE")
                    st64 = fj["stage_generate_shade"]
                    if st64["fp64_flop_per_vertex"] > 50.0:       # (a Phong scene has a few fp64 operations per vertex — promotion points of the reference's float code — and stays on the byte roofline)
                        peak64 = 78.6                              # TFLOP/s: MI355X fp64 vector FMA peak = half the fp32 vector peak of MI355X_MICROARCH.md (157.3)
                        ach = st64["fp64_flop_per_vertex"] * verts / shade_secs / 1e12 if same64 else None
                        tier = {k: v for k, v in fj["kernels"].items() if k.startswith(("k_wf_shade<5>", "k_wf_shade<4>", "k_wf_shade<3>", "k_wf_merl_eval"))}
                        rsh = {"kernel": rsh["kernel"] + " — measured-BRDF tier k_wf_shade<5> + k_wf_merl_eval (k_wf_shade<3/4> with merl_batch 0 / 1)", "bound": "fp64", "unit": "TFLOP/s", "peak": peak64, "achieved": ach, "frac": (ach / peak64) if ach is not None else None,
                               "frac_definition": "fp64 operations (ADD + MUL + 2 FMA + TRANS wave-instructions x mean active lanes, SQ_INSTS_VALU_*_F64 of the committed counter run) per shade vertex x vertices of this run / stage time / fp64 vector peak 78.6 TFLOP/s",
                               "fp64_flop_per_vertex": st64["fp64_flop_per_vertex"], "frac_if_every_fp64_instruction_had_64_lanes": (st64["fp64_issue_slot_flop_per_vertex"] * verts / shade_secs / 1e12 / peak64) if same64 else None,
                               "vertices_per_step": verts / args.steps, "ms_per_step": shade_ms / args.steps,
                               "measured_brdf_tiers": {k: {q: v[q] for q in ("launches_per_step", "fp64_share_of_vector_instructions", "active_lanes_per_vector_instruction", "vector_instructions_per_simd_and_cycle", "wait_share_of_wave_cycles")} for k, v in tier.items()},
                               "derived_from_pmc_run": {"file": "profiles/fp64_counters.json", "source": fj["source"], "same_library_build": same64, "stale": not same64},
                               "why_not_bytes": "the tier evaluates IsoMERLBRDF::eval (BRDF.h:204-246, MERLBRDFRead.cpp:76-206) in fp64 with the host libm's algorithms: about a third of its vector instructions are fp64, "
                                                "its waves wait (dependent chains at 3 waves per SIMD, 168 registers) for most of their cycles; the byte figures are kept under `hbm`",
                               "hbm": rsh}
                except Exception:
                    pass
                out["roofline_shade_kernel"] = rsh
            out["stage_ms_per_step"] = {("traverse" if merged else "extend"): kern_ms / args.steps, "shadow": sh_ms / args.steps, "generate+shade": shade_ms / args.steps, "resolve": resolve_ms / args.steps}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(mesh, mat, cfg, ob["rays_per_path"])
        assert_physical(out)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
