#!/usr/bin/env python3
"""tools/scale_preflight.py — the paths a multi-GPU bench run needs, one at a time, before `bench.py --gpus N` needs them all at once.

    python tools/scale_preflight.py            # N = torch.cuda.device_count()   (no GPU is initialised before the stages start)

The driver's scaling run (bench.py --gpus 2 / 4 / 8) is the first time three things execute on real hardware: ncclCommInitAll(n > 1) +
the n-rank ncclReduce inside the library (csrc/mipt_group.h), the hipMemcpyPeer replication of a mesh that was built on device 0
(csrc/mipt_mesh_device.h), and torch.distributed's RCCL all-reduce of the framebuffers.  This script runs them in that order, each in
its OWN child process with a timeout (a hang or a crash of one stage cannot take the others with it), and prints ONE JSON line per
stage: {"stage", "ok", "seconds", ...what the stage measured... | "error"}.  The last lines are bench.py itself (2 steps), in both
multi-GPU forms.  Exit code: number of failed stages.

On a box with ONE GPU the stages run in their degenerate forms (a group that lists device 0 twice sums its framebuffers with device
copies instead of RCCL; two gloo ranks share the GPU) and say so ("degenerate": true): that checks the script, not the interconnect.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


# ------------------------------------------------------------------------------------------------ stages (each runs in a child)
def stage_rccl_selftest(n):
    """mipt_rccl_selftest on every device: dlopen of librccl, the symbols and their signatures, a one-rank ncclReduce."""
    from pathtracer_amd import capi
    for d in range(n):
        rt = capi.HostRaytracer(device=d)
        rt.rccl_selftest()
        rt.close()
    return {"devices": n}


def _small_frame(devices, name="blob32", W=1920, H=1080, spp=1):
    """One frame of a golden scene at W x H x spp on `devices` (an int or a list): (image, weights, reduce kind)."""
    from make_golden import golden_scene
    from pathtracer_amd import capi, scenes
    mesh, cfg, mat = golden_scene(name)
    cfg.W, cfg.H, cfg.spp = W, H, spp
    rt = capi.HostRaytracer(device=devices)
    if isinstance(devices, list) and len(set(devices)) == len(devices) and len(devices) > 1:
        rt.set_option("reduce", 1)              # RCCL or fail: no silent fall-back to peer copies
    rt.apply_config(cfg)
    oid = rt.add_mesh(mesh)
    scenes.install_material(rt, oid, mat)
    rt.prepare()
    img, cnt = rt.render()
    kind = rt.group_reduce_kind() if isinstance(devices, list) else ""
    return img, cnt, kind, rt.mesh_on_device(oid)


def stage_group_reduce(n):
    """ncclCommInitAll(n) + ONE n-rank ncclReduce of a 1080p framebuffer (W*H*4 floats = 33 MB) inside the library: a 2 048-triangle scene at
    1 spp, so the frame's work is nothing beside the reduce; the sum must be the single-device frame."""
    import numpy as np
    devs = list(range(n)) if n > 1 else [0, 0]
    t0 = time.perf_counter()
    img, cnt, kind, _ = _small_frame(devs)
    t_group = time.perf_counter() - t0
    ref_img, ref_cnt, _, _ = _small_frame(0)
    err = float(np.abs(img / np.maximum(cnt, 1e-30)[..., None] - ref_img / np.maximum(ref_cnt, 1e-30)[..., None]).max() / 196964.7)
    out = {"devices": devs, "reduce": kind, "frame_bytes": int(img.size * 4 + cnt.size * 4), "seconds_group_frame": t_group, "max_err_vs_one_device_over_white": err, "degenerate": n < 2}
    if n > 1 and not kind.startswith("RCCL"):
        raise RuntimeError("the group did not reduce with RCCL: " + kind)
    if not err < 1e-5:
        raise RuntimeError("group frame differs from the single-device frame: %g" % err)
    return out


def stage_peer_replication(n):
    """A mesh built on device 0 (mipt_device_mesh_build) replicated to the other members by hipMemcpyPeer at upload; n < 2: the test hook
    device_mesh_as_remote (a peer copy within one device).  The textured golden scene (UV triples, tangents, textures travel too)."""
    import numpy as np
    from make_golden import golden_scene
    from pathtracer_amd import capi, scenes
    frames = {}
    for form in ("in place on device 0", "replicated"):
        mesh, cfg, mat = golden_scene("textured")
        devs = 0 if form.startswith("in place") else (list(range(n)) if n > 1 else 0)
        rt = capi.HostRaytracer(device=devs)
        rt.apply_config(cfg)
        oid = rt.add_mesh(mesh)
        assert rt.mesh_on_device(oid)
        scenes.install_material(rt, oid, mat)
        if form == "replicated" and n < 2:
            rt.set_option("device_mesh_as_remote", 1)
        rt.prepare()
        img, cnt = rt.render()
        frames[form] = (img, cnt)
    a, b = frames["in place on device 0"], frames["replicated"]
    err = float(np.abs(a[0] / np.maximum(a[1], 1e-30)[..., None] - b[0] / np.maximum(b[1], 1e-30)[..., None]).max() / 196964.7)
    if not err < 1e-5:
        raise RuntimeError("frame from the replicated mesh differs: %g" % err)
    return {"devices": list(range(n)) if n > 1 else [0], "max_err_over_white": err, "degenerate": n < 2}


def stage_torch_allreduce_rank():
    """(under torch.distributed.run) one all-reduce of a 33 MB tensor: bench.py's one-process-per-GPU reduce."""
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("PREFLIGHT_BACKEND", "nccl")
    dev = torch.device("cuda", local if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    x = torch.full((1920 * 1080 * 4,), float(rank + 1), dtype=torch.float32, device=dev)
    times = []
    for it in range(4):
        y = x.clone() if backend == "nccl" else x.cpu()
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    want = world * (world + 1) / 2
    ok = bool((y == want).all().item())
    if rank == 0:
        print(json.dumps({"inner": True, "ok": ok, "world": world, "backend": backend, "bytes": int(x.numel() * 4), "seconds_per_all_reduce": min(times[1:]),
                          "gb_per_s": x.numel() * 4 / min(times[1:]) / 1e9}))
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


STAGES = {"rccl_selftest": stage_rccl_selftest, "group_reduce": stage_group_reduce, "peer_replication": stage_peer_replication}


# ------------------------------------------------------------------------------------------------ driver
def run_child(name, cmd, timeout, env=None):
    t0 = time.perf_counter()
    rec = {"stage": name}
    try:
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        inner = json.loads(lines[-1]) if lines else {}
        rec.update({k: v for k, v in inner.items() if k != "inner"})
        rec["ok"] = r.returncode == 0 and inner.get("ok", True) is not False and (inner.get("value", 1) is not None)
        if not rec["ok"]:
            rec["error"] = inner.get("error") or (r.stderr.strip().splitlines() or ["exit code %d" % r.returncode])[-1][-600:]
            rec["exit_code"] = r.returncode
    except subprocess.TimeoutExpired:
        rec.update(ok=False, error="timed out after %d s" % timeout)
    rec["seconds"] = round(time.perf_counter() - t0, 2)
    print(json.dumps(rec), flush=True)
    return rec["ok"]


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--stage":
        if sys.argv[2] == "torch_allreduce_rank":
            stage_torch_allreduce_rank()
            return
        out = STAGES[sys.argv[2]](int(sys.argv[3]))
        out["inner"] = True
        print(json.dumps(out))
        return
    import torch
    n = torch.cuda.device_count()                 # does not initialise the GPU
    if n < 1:
        print(json.dumps({"stage": "devices", "ok": False, "error": "no GPU visible"}))
        raise SystemExit(1)
    print(json.dumps({"stage": "devices", "ok": True, "visible": n, "degenerate": n < 2}), flush=True)
    py, me = sys.executable, os.path.abspath(__file__)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    failed = 0
    for name in ("rccl_selftest", "group_reduce", "peer_replication"):
        failed += not run_child(name, [py, me, "--stage", name, str(n)], 600, env)
    world = max(n, 2)
    backend = "nccl" if n > 1 else "gloo"
    env2 = dict(env); env2["PREFLIGHT_BACKEND"] = backend
    launch = [py, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1"]
    failed += not run_child("torch_all_reduce_33MB (%s, %d ranks)" % (backend, world), launch + ["--master-port", "29541", me, "--stage", "torch_allreduce_rank"], 600, env2)
    small = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if n > 1:
        failed += not run_child("bench.py --gpus %d (one process, mipt_create(ids, %d))" % (n, n), [py, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + small, 1500, env)
        failed += not run_child("bench.py --gpus %d (one process per GPU, RCCL all-reduce)" % n, launch + ["--master-port", "29542", os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + small, 1500, env)
    else:
        failed += not run_child("bench.py --in-process 0,0 (degenerate group on one GPU)", [py, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--in-process", "0,0"] + small, 900, env)
        failed += not run_child("bench.py two gloo ranks sharing the GPU (degenerate)", launch + ["--master-port", "29542", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"] + small, 900, env)
    print(json.dumps({"stage": "summary", "ok": failed == 0, "failed_stages": failed, "devices": n}))
    raise SystemExit(failed)


if __name__ == "__main__":
    main()
