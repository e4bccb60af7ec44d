"""The reference's OWN Raytracer::render_image_nopreviz() / render_image() with the USE_MIPT switch (oracle/_ref/libptref_mipt.so: nbonneel/pathtracer compiled
with integration/use_mipt against libmipt.so) on configs[1]'s scene at 1080p: what a maintainer who applies the switch gets, wall clock of the member function
(prepare_render + mipt_upload of the host arrays + render + division / tone map), against the same frame through the host mirror.
TEST INFRASTRUCTURE (uses oracle/binding.py); usage: python tools/reference_binding_rate.py [spp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding          # noqa: E402
from pathtracer_amd import capi, scenes   # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh, cfg, mat, text = scenes.workload("c1", 1920, 1080, spp, None)
X = binding.RefMipt()
X.apply_config(cfg)
t0 = time.time(); scenes.install(X, mesh, mat); t_init = time.time() - t0      # the reference's readOBJ + TriMesh::init (build_bvh -> mipt_build_bvh)
X.time_render_nopreviz(8)           # warm-up (context creation, first upload)
assert X.status() == 0, X.error()
t, img = X.time_render_nopreviz(8)
st = X.stats()
rays = st["rays_closest"] + st["rays_shadow"]
print("reference classes + USE_MIPT: %s, 1920x1080x%d: Raytracer::render_image_nopreviz() %.3f s wall = %.0f Mrays/s (%d passes; TriMesh ctor incl. readOBJ %.2f s)"
      % (text, spp, t, rays / t / 1e6, st["passes"], t_init))
t2, img2, cnt2 = X.time_render_image(8)
print("reference classes + USE_MIPT: Raytracer::render_image() (a publish per sample) %.3f s wall = %.0f Mrays/s" % (t2, rays / t2 / 1e6))
H = capi.HostRaytracer(device=0)
H.apply_config(cfg); scenes.install(H, mesh, mat); H.prepare()
H.render_image_nopreviz()
t0 = time.time(); H.render_image_nopreviz(); th = time.time() - t0
print("host mirror: Raytracer::render_image_nopreviz() %.3f s wall = %.0f Mrays/s" % (th, rays / th / 1e6))
