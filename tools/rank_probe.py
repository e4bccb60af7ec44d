"""What one rank of an N-rank tile partition costs on one GPU (configs[2], a step = 1080p x 1024 spp): stage times of rank 0 for
N = 1, 2, 4, 8 (N x its time / the N = 1 time = the scaling loss that is not the reduce), and at N = 8 with the splat unsliced
(resolve_slices = 1).  usage: python tools/rank_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from pathtracer_amd import capi, scenes
mesh, cfg, mat, text = scenes.workload("c2", 1920, 1080, 1024, None)
H = capi.HostRaytracer(device=0)
H.apply_config(cfg); scenes.install(H, mesh, mat); H.prepare()
for nr, zs in [(1, 0), (2, 0), (4, 0), (8, 0), (8, 1)] if len(sys.argv) < 2 else [(1, int(v)) for v in sys.argv[1:]]:
    H.set_option('resolve_slices', zs)
    pr = H.params
    pr.tile_size, pr.tile_rank, pr.tile_nranks = 32, 0, nr
    H.render()
    t0 = time.time(); H.render(); dt = time.time() - t0
    st = H.stats()
    print(json.dumps({"nranks": nr, "slices": zs, "wall_ms": round(dt * 1e3, 1), "render_ms": round(st["render_ms"], 1), "extend": round(st["traverse_ms"], 1), "shadow": round(st["shadow_ms"], 1), "shade": round(st["shade_ms"], 1), "resolve": round(st["resolve_ms"], 1), "passes": st["passes"], "rays": st["rays_closest"] + st["rays_shadow"]}), flush=True)
