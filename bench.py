#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): the configuration BASELINE.json's metric is quoted on ("1080p x 1024 spp",
"the 2.5M-triangle scene") = configs[2]: 2 508 800-triangle blob with a 2048x2048 Kd texture and a
4096x2048 environment map, Phong BRDF, 1920x1080, depth 4, default loadScene() light and camera; it
fits one GPU.  --workload c1 / c3 / c4 select the other configs (c1 = 133 128-triangle diffuse blob).
One STEP = one pass of the hot path (camera rays -> getColor -> splat) over the whole frame at
--spp-per-step samples per pixel (default 256: 531 M paths, ~85 GB of path state in flight — sized for
288 GB of HBM; the per-launch ramp and drain of the persistent kernels is amortised over a large batch: 3.72 Grays/s
at 64 spp per step, 3.82 at 128, 3.94 at 256, 3.98 at 512);
the default K = 4 steps x 256 spp is exactly the 1024 spp of the config.  With N GPUs the frame's
32x32-pixel tiles are dealt round-robin to the ranks (one process per GPU, scene replicated), a step
renders N x --spp-per-step samples per pixel so that every rank keeps the same number of paths in
flight per pass as the single-GPU run ("scaling": "weak": per-GPU work per step is fixed), and the
per-rank accumulators are summed by ONE all-reduce at the end (RCCL).

value = rays (closest-hit + shadow, counted like the oracle counts them) of all ranks / wall time
of the K timed steps (+ the final reduce), inputs resident in HBM, barrier + synchronize on both
sides, max over ranks.

roofline: the dominant kernel's ALGORITHMIC bytes per launch / its mean launch duration (HIP events
on the render stream, measured here).  Algorithmic bytes per ray come from the CPU oracle's counters
of the reference's ordered traversal on a bounded sample of the same scene and camera
(B_ray = 24*n_box + 8*n_node + 64*n_tri), times the rays one launch casts.

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

SPP_PER_STEP = 256          # at 1080p; scaled down with the pixel count so that a pass keeps <= 2^29 paths (~86 GB of path state)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c1", "c1g", "c2", "c3", "c4"], help="BASELINE.json configs[1..4]")
    ap.add_argument("--spp-per-step", type=int, default=0, help="samples per pixel per step (default: 256 at 1080p, 64 at 4K)")
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
    return dict(bytes_closest=b_closest, bytes_shadow=b_shadow, rays_per_path=float(rays[0] + rays[1]) / (cfg.W * cfg.H * cfg.spp),
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


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.share_gpu:
        local_rank = 0
    if world > 1:
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
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from pathtracer_amd import capi, scenes

    if args.spp_per_step <= 0:
        dims = {"c4": (3840, 2160)}.get(args.workload, (1920, 1080))
        npx = (args.width or dims[0]) * (args.height or dims[1])
        args.spp_per_step = SPP_PER_STEP
        while args.spp_per_step > 1 and npx * args.spp_per_step > (1 << 29) + (1 << 24):
            args.spp_per_step //= 2
    SPS = args.spp_per_step * world     # weak scaling: a rank owns 1/world of the pixels and renders world x the samples per step
    total_spp = SPS * (args.steps + args.warmup)
    mesh, cfg, mat, wl_text = scenes.workload(args.workload, args.width, args.height, total_spp, args.grid)
    args.width, args.height = cfg.W, cfg.H

    rt = capi.HostRaytracer(device=local_rank)
    rt.apply_config(cfg)
    rt.set_partition(32, rank, world)
    t0 = time.time()
    mesh_obj = scenes.install(rt, mesh, mat)
    t_build = time.time() - t0
    bvh_who, bvh_s, bvh_dev_s = rt.mesh_bvh_builder(mesh_obj)
    t0 = time.time()
    rt.prepare()
    t_prepare = time.time() - t0
    if args.pipeline >= 0:
        rt.set_option("pipeline", args.pipeline)
    rt.set_option("paths_per_pass", args.width * args.height * SPS)
    for kv in args.opt:
        k, v = kv.split("=")
        rt.set_option(k, int(v))

    import ctypes as C
    accum = torch.zeros(args.width * args.height * 4, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    P = rt.params       # live view of the host mirror's mipt_render_params; patched per step
    assert P.W == args.width and P.nrays == total_spp and P.seed_stride == 65536 and P.tile_nranks == world

    def step(s):
        P.sample_begin, P.sample_end = s * SPS, (s + 1) * SPS
        rt.render_device(accum.data_ptr(), stream)
        st = rt.stats()    # synchronises; cheap next to a step
        return st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    sync()
    t0 = time.perf_counter()
    rays = paths = 0
    kern_ms = sh_ms = shade_ms = 0.0
    launches = sh_launches = 0
    rays_c = rays_s = 0
    pipeline = -1
    merged = 0
    for s in range(args.warmup, args.warmup + args.steps):
        st = step(s)
        rays_c += st["rays_closest"]; rays_s += st["rays_shadow"]; paths += st["paths"]
        kern_ms += st["traverse_ms"]; launches += st["traverse_launches"]
        sh_ms += st["shadow_ms"]; sh_launches += st["shadow_launches"]; shade_ms += st["shade_ms"]
        pipeline = st["pipeline"]; merged = st["traverse_merged"]
    if world > 1:                                        # the framebuffer reduce (RCCL over xGMI)
        if on_device:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)
        else:
            host = accum.cpu(); dist.all_reduce(host, op=dist.ReduceOp.SUM); accum.copy_(host)
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(rays_c), float(rays_s), float(paths), kern_ms, float(launches)], dtype=torch.float64, device=dev if on_device else "cpu")
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0]); rays_c, rays_s, paths = float(tsum[1]), float(tsum[2]), float(tsum[3])
    rays = rays_c + rays_s

    if rank == 0:
        img = accum[: args.width * args.height * 3]
        finite = bool(torch.isfinite(img).all().item())
        out = {
            "metric": "Msamples/s (primary+secondary rays) at 1080p\u00d71024spp; 1/2/4/8-GPU scaling",
            "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl_text}, {args.width}x{args.height}, {SPS * args.steps} spp timed ({SPS} spp/step), depth {cfg.nb_bounces}",
                       "parallelism": f"tiles32x{world}", "pipeline": int(pipeline)},
            "mpaths_per_s": paths / elapsed / 1e6, "rays_per_path": rays / max(1.0, paths),
            "host_bvh_build_s": t_build,   # TriMesh::init as a whole (axis swap, BVH, triangle soup, tangents)
            "bvh_build": {"builder": bvh_who, "seconds": round(bvh_s, 4), "device_seconds": round(bvh_dev_s, 4), "triangles": int(mesh.ntri)},
            "prepare_s": t_prepare, "finite": finite,
        }
        if world == 1 and args.pmc and pipeline == 1:
            out["stage_ms_per_step"] = {("traverse" if merged else "extend"): kern_ms / args.steps, "shadow": sh_ms / args.steps, "generate+shade": shade_ms / args.steps}
        if world == 1 and not args.pmc:
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
            achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
            try:
                stream = rt.measure_stream_read(8 << 30, 5)      # achievable read bandwidth of this device
            except Exception:
                stream = None
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                               "peak_measured_stream_read": stream, "frac_of_measured": (achieved / stream) if stream else None,
                               "traffic": None, "kernel": kernel,
                               "bytes_per_closest_ray": ob["bytes_closest"], "bytes_per_shadow_ray": ob["bytes_shadow"],
                               "ms_per_launch": ms_per_launch, "launches": int(launches), "oracle_sample": ob["sample"],
                               "rays_per_launch": (rays_c if (pipeline and not merged) else rays_c + rays_s) / my_launches}
            try:   # HBM bytes per launch of the dominant kernel from the committed PMC run (same workload)
                tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))[args.workload]
                tr = tj["kernels"]["k_wf_traverse<2>" if merged else ("k_wf_traverse<0>" if pipeline == 1 else "k_render_paths")]
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch_high"]
                out["roofline"]["traffic_note"] = "rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch of this kernel on this workload, " + tj["source"]
            except Exception:
                pass
            if pipeline == 1 and sh_launches:
                sh_ach = rays_s * ob["bytes_shadow"] / (sh_ms * 1e-3) / 1e9
                out["roofline_shadow_kernel"] = {"kernel": "k_wf_traverse<true> (any-hit / shadow stage)", "achieved": sh_ach, "frac": sh_ach / 8000.0, "ms_per_launch": sh_ms / sh_launches,
                                                 "launches": int(sh_launches)}
            out["stage_ms_per_step"] = {("traverse" if merged else "extend"): kern_ms / args.steps, "shadow": sh_ms / args.steps, "generate+shade": shade_ms / args.steps}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(mesh, mat, cfg, ob["rays_per_path"])
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
