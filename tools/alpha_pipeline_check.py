import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import numpy as np
from helpers import *
from pathtracer_amd import capi, scenes
def run(pl, img, refill=0):
    rt = capi.HostRaytracer(device=0)
    cfg = scenes.config_c1(64, 36, 8); cfg.nb_bounces = 1
    rt.apply_config(cfg); oid = rt.add_mesh(scenes.blob_mesh(24, with_uv=True))
    rt.set_group_material(oid, 0, (0.8,)*3, (0,)*3, (0,)*3)
    rt.set_group_texture(oid, 0, 3, img)
    rt.prepare(); rt.set_option("pipeline", pl); rt.set_option("refill", refill)
    return rt.sample_radiance(all_pixels(cfg), 0, cfg.spp)[0], cfg
full = np.full((8, 8, 3), 255, np.uint8); none = np.zeros((8, 8, 3), np.uint8)
half = full.copy(); half[:, :4] = 0
for name, img in (("opaque", full), ("all holes", none), ("half", half), ("grid", scenes.alpha_texture())):
    a = run(0, img)[0]; b = run(1, img)[0]; b2 = run(1, img)[0]; c = run(1, img, 1)[0]
    print(name, "p0 vs p1:", int((~bits_equal(a, b).all(-1)).sum()), "p1 rerun differs:", int((~bits_equal(b, b2).all(-1)).sum()), "p0 vs p1-refill:", int((~bits_equal(a, c).all(-1)).sum()))
