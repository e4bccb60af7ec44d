"""What one rank of an N-rank tile partition costs on one GPU (a step of the workload: 1080p x 1024 spp; configs[4]: 4K x 256 spp):
stage times of rank 0 for N = 1, 2, 4, 8 (N x its time / the N = 1 time = the scaling loss that is not the reduce), and at N = 8
with the splat unsliced (resolve_slices = 1).  usage: python tools/rank_probe.py [c2|c3|c4] [slices ...]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from pathtracer_amd import capi, scenes
wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("c1", "c2", "c3", "c4") else "c2"
rest = [a for a in sys.argv[1:] if a != wl]
dims = (3840, 2160, 256) if wl == "c4" else (1920, 1080, 1024)
mesh, cfg, mat, text = scenes.workload(wl, dims[0], dims[1], dims[2], None)
H = capi.HostRaytracer(device=0)
H.apply_config(cfg); scenes.install(H, mesh, mat); H.prepare()
t1 = None
for nr, zs in [(1, 0), (2, 0), (4, 0), (8, 0), (8, 1)] if not rest else [(1, int(v)) for v in rest]:
    H.set_option('resolve_slices', zs)
    pr = H.params
    pr.tile_size, pr.tile_rank, pr.tile_nranks = 32, 0, nr
    H.render()
    t0 = time.time(); H.render(); dt = time.time() - t0
    st = H.stats()
    if nr == 1 and t1 is None: t1 = st["render_ms"]
    print(json.dumps({"workload": wl, "nranks": nr, "slices": zs, "wall_ms": round(dt * 1e3, 1), "render_ms": round(st["render_ms"], 1), "n_x_t_over_t1": round(nr * st["render_ms"] / t1, 4), "extend": round(st["traverse_ms"], 1), "shadow": round(st["shadow_ms"], 1), "shade": round(st["shade_ms"], 1), "resolve": round(st["resolve_ms"], 1), "passes": st["passes"], "rays": st["rays_closest"] + st["rays_shadow"]}), flush=True)
