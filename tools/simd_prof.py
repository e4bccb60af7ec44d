import sys, os, ctypes as C, shutil
sys.path.insert(0, os.getcwd())
# diagnostic: swap in the profiling build of the library
shutil.copy("pathtracer_amd/libmipt.so", "/tmp/libmipt_orig.so")
shutil.copy("pathtracer_amd/libmipt_prof.so", "pathtracer_amd/libmipt.so")
import numpy as np
from pathtracer_amd import capi, scenes
cfg = scenes.config_c1(1920, 1080, 4)
mesh = scenes.blob_mesh(258)
rt = capi.HostRaytracer(device=0)
rt.apply_config(cfg); rt.add_mesh(mesh); rt.prepare()
rt.set_option("pipeline", 1)
out = (C.c_ulonglong * 4)()
rt.mipt.mipt_debug_simd_profile(out, 1)
rt.render()
rt.mipt.mipt_debug_simd_profile(out, 1)
st = rt.stats()
print("inner: wave-iters %d mean active lanes %.1f ; leaf: wave-iters %d mean active lanes %.1f" % (out[0], out[1]/max(1,out[0]), out[2], out[3]/max(1,out[2])))
print("rays", st["rays_closest"], st["rays_shadow"], "inner steps/ray %.1f leaf steps/ray %.2f" % (out[1]/(st["rays_closest"]+st["rays_shadow"]), out[3]/(st["rays_closest"]+st["rays_shadow"])))
shutil.copy("/tmp/libmipt_orig.so", "pathtracer_amd/libmipt.so")
