import sys, os, ctypes as C, shutil
sys.path.insert(0, os.getcwd())
# diagnostic: swap in the profiling build of the library
from pathtracer_amd import capi
capi.LIBMIPT = capi.LIBMIPT.replace("libmipt.so", "libmipt_prof.so")
import numpy as np
from pathtracer_amd import capi, scenes
cfg = scenes.config_c1(1920, 1080, 4)
mesh = scenes.blob_mesh(258)
rt = capi.HostRaytracer(device=0)
rt.apply_config(cfg); rt.add_mesh(mesh); rt.prepare()
rt.set_option("pipeline", 1)
out = (C.c_ulonglong * 4)()
for refill, thr, imin in ((0, 20, 0), (1, 20, 0), (1, 20, 8), (1, 20, 16), (1, 20, 24), (1, 20, 32), (1, 8, 24), (1, 32, 24)):
  rt.set_option("refill", refill); rt.set_option("refill_threshold", thr); rt.set_option("inner_min", imin)
  rt.mipt.mipt_debug_simd_profile(out, 1)
  rt.render()
  rt.mipt.mipt_debug_simd_profile(out, 1)
  st = rt.stats()
  print("refill", refill, "threshold", thr, "inner_min", imin, "extend ms %.2f shadow ms %.2f" % (st["traverse_ms"], st["shadow_ms"]))
  print("inner: wave-iters %d mean active lanes %.1f ; leaf: wave-iters %d mean active lanes %.1f" % (out[0], out[1]/max(1,out[0]), out[2], out[3]/max(1,out[2])))
  print("rays", st["rays_closest"], st["rays_shadow"], "inner steps/ray %.1f leaf steps/ray %.2f" % (out[1]/(st["rays_closest"]+st["rays_shadow"]), out[3]/(st["rays_closest"]+st["rays_shadow"])))

