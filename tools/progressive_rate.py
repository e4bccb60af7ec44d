"""Cost of Raytracer::render_image (one pass per sample, buffers published after every pass) on configs[1]'s scene at
1080p: per-pass cost from the difference of a 16-sample and a 64-sample call (the rest is prepare_render + scene upload)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes   # noqa: E402

opts = [a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("spp=")]        # e.g. progressive_lookahead=8
pair = [a[4:] for a in sys.argv[1:] if a.startswith("spp=")]                                   # spp=64,256: the two render lengths whose difference is taken (default 16,64)
lo, hi = (int(x) for x in pair[0].split(",")) if pair else (16, 64)
wall = {}
for spp in (lo, hi):
    mesh, cfg, mat, text = scenes.workload("c1", 1920, 1080, spp, None)
    H = capi.HostRaytracer(device=0)
    H.apply_config(cfg)
    scenes.install(H, mesh, mat)
    H.prepare()
    for k, v in opts:
        H.set_option(k, int(v))
    look = dict(opts).get("progressive_lookahead")
    H.render_image(look)        # warm-up
    t0 = time.time(); H.render_image(look); wall[spp] = time.time() - t0
    st = H.stats()
    rays = st["rays_closest"] + st["rays_shadow"]
    print("render_image %d spp: %.1f ms wall, GPU span %.1f ms, %.1f M rays" % (spp, wall[spp] * 1e3, st["render_ms"], rays / 1e6))
per_pass = (wall[hi] - wall[lo]) / (hi - lo)
print(dict(opts), "spp %d,%d: per sample of Raytracer::render_image: %.2f ms (%.0f Mrays/s); fixed per call: %.0f ms; the %d-sample call as a whole: %.0f Mrays/s" % (lo, hi, per_pass * 1e3, rays / hi / per_pass / 1e6, (wall[lo] - lo * per_pass) * 1e3, hi, rays / wall[hi] / 1e6))
