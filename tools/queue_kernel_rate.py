"""Throughput of the contribution-queue kernel (ghost / photo / fog / subsurface scenes): configs[1]'s scene at 1080p with
the feature switched on, N spp through mipt_render_device-equivalent host call; prints Mrays/s from the ABI's counters.
usage: python tools/queue_kernel_rate.py [spp] [option=value ...] [only=feature]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes   # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opts = dict(kv.split("=") for kv in sys.argv[2:])        # e.g. queue_wavefront=0 : the one-thread-per-sample kernel; only=fog : one feature
only = opts.pop("only", None)
for feature in ("none", "photo+ghostfloor", "fog", "subsurface"):
    if only and only not in feature:
        continue
    mesh, cfg, mat, text = scenes.workload("c1", 1920, 1080, spp, None)
    H = capi.HostRaytracer(device=0)
    H.apply_config(cfg)
    oid = scenes.install(H, mesh, mat)
    if feature == "photo+ghostfloor":
        H.set_object_ghost(2, True)
        H.set_background((np.random.default_rng(1).uniform(0, 1, (270, 480, 3)) ** 2.2 * 196964.699).astype(np.float32))
    if feature == "fog":
        H.set_fog(0.5, 0.4, 0.02, 0.03, 1, 1, 0.4)
    if feature == "subsurface":
        H.set_group_subsurface(oid, 0, (0.8, 0.5, 0.3))
    H.prepare()
    for kopt, v in opts.items():
        H.set_option(kopt, int(v))
    H.render()                                    # warm-up (code object load, buffers)
    dt, st = 1e30, None
    for rep in range(3):                          # best of three (the host side of a render is ~10 ms of Python + readback)
        t0 = time.time()
        img, cnt = H.render()
        d = time.time() - t0
        if d < dt:
            dt, st = d, H.stats()
    rays = st["rays_closest"] + st["rays_shadow"]
    print(json.dumps({"feature": feature, "pipeline": st["pipeline"], "spp": spp, "Mrays_per_s": round(rays / dt / 1e6, 1), "seconds": round(dt, 3),
                      "rays_per_path": round(rays / max(1, st["paths"]), 2), "passes": st["passes"], "samples_through_the_fallback": st["reserved"], "kernel_ms": {"closest": round(st["traverse_ms"], 1), "shadow": round(st["shadow_ms"], 1), "logic": round(st["shade_ms"], 1), "splat": round(st["resolve_ms"], 1)}, "finite": bool(np.isfinite(img).all())}), flush=True)
