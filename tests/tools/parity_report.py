"""Fraction of per-sample radiance values that are bit-identical to the reference's (golden fixtures) and the max
normalised error, per golden scene.  usage: python tests/tools/parity_report.py"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
from helpers import bits_equal, WHITE
import make_golden as mg
from pathtracer_amd import capi, scenes
for name in ("cornell", "blob32", "glossy", "glass", "textured", "cutout", "merl"):
    g = np.load(os.path.join(mg.OUT, f"scene_{name}.npz"))
    H = capi.HostRaytracer(device=0)
    mesh, cfg, oid = mg.setup(H, name)
    rgb, _ = H.sample_radiance(mg.all_pixels(cfg), 0, cfg.spp)
    print("%-10s bit-identical %.5f  max |err| / white %.3g" % (name, bits_equal(rgb, g["sample_rgb"]).mean(), np.abs(rgb.astype(np.float64) - g["sample_rgb"]).max() / WHITE))
g = np.load(os.path.join(mg.OUT, "objscene.npz"))
H = capi.HostRaytracer(device=0); cfg = scenes.config_c1(64, 36, 4); H.apply_config(cfg)
H.add_mesh_obj(scenes.write_obj_scene(tempfile.mkdtemp())); H.prepare()
rgb, _ = H.sample_radiance(mg.all_pixels(cfg), 0, cfg.spp)
print("%-10s bit-identical %.5f  max |err| / white %.3g" % ("objscene", bits_equal(rgb, g["sample_rgb"]).mean(), np.abs(rgb.astype(np.float64) - g["sample_rgb"]).max() / WHITE))
