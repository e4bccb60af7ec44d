"""Would a spatially narrower working set per launch make the traversal faster?  configs[2] rendered once as usual (every
pass covers the whole frame) and once tile by tile (the multi-GPU partition with as many "ranks" as tiles, rendered one
after the other on this GPU: every launch then works on one tile_size x tile_size square of the image at all spp).  Same
paths, same rays; prints the stage times of both.  usage: python tools/locality_probe.py [spp] [tile_size]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes   # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ts = int(sys.argv[2]) if len(sys.argv) > 2 else 270
mesh, cfg, mat, text = scenes.workload("c2", 1920, 1080, spp, None)
H = capi.HostRaytracer(device=0)
H.apply_config(cfg)
scenes.install(H, mesh, mat)
H.prepare()


def run(parts):
    tot = {"traverse_ms": 0.0, "shadow_ms": 0.0, "shade_ms": 0.0, "resolve_ms": 0.0, "rays": 0, "wall": 0.0}
    for r in range(parts):
        pr = H.params
        pr.tile_size, pr.tile_rank, pr.tile_nranks = (ts if parts > 1 else 32), r, parts
        t0 = time.time()
        H.render()
        tot["wall"] += time.time() - t0
        st = H.stats()
        for k in ("traverse_ms", "shadow_ms", "shade_ms", "resolve_ms"):
            tot[k] += st[k]
        tot["rays"] += st["rays_closest"] + st["rays_shadow"]
    return {k: (round(v, 1) if isinstance(v, float) else v) for k, v in tot.items()}


run(1)
print(json.dumps({"mode": "whole frame per pass", "spp": spp, **run(1)}), flush=True)
ntiles = ((1920 + ts - 1) // ts) * ((1080 + ts - 1) // ts)
print(json.dumps({"mode": "one %dx%d tile per render" % (ts, ts), "tiles": ntiles, "spp": spp, **run(ntiles)}), flush=True)
