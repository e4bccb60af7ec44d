"""Diagnostic: lane utilisation and phase shares of the persistent traversal kernels (profiling build of the library:
tools/build_variant.sh prof -DMIPT_PROFILE_SIMD).  usage: python tools/simd_prof.py [c1|c2] [name=value ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["MIPT_LIB_OVERRIDE"] = os.path.join(os.getcwd(), "pathtracer_amd", "libmipt_prof.so")
from pathtracer_amd import capi, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "c1"
spp = 4
for a in list(sys.argv[2:]):
    if a.startswith("spp="):          # samples per pixel of the profiled render (4: a queue so short that ramp and drain dominate; 64: the bench's regime)
        spp = int(a[4:]); sys.argv.remove(a)
if os.environ.get("MIPT_PROF_LIB"):
    os.environ["MIPT_LIB_OVERRIDE"] = os.path.join(os.getcwd(), "pathtracer_amd", os.environ["MIPT_PROF_LIB"])
depth = None
for a in list(sys.argv[2:]):
    if a.startswith("depth="):        # nb_bounces of the profiled render (1: only the camera rays' extend and their shadow rays)
        depth = int(a[6:]); sys.argv.remove(a)
mesh, cfg, mat, text = scenes.workload(wl, spp=spp)
if depth:
    cfg.nb_bounces = depth
rt = capi.HostRaytracer(device=0)
rt.apply_config(cfg); scenes.install(rt, mesh, mat); rt.prepare()
rt.set_option("pipeline", 1)
for kv in sys.argv[2:]:
    k, v = kv.split("="); rt.set_option(k, int(v))
out = (C.c_ulonglong * 40)()
rt.mipt.mipt_debug_simd_profile(out, 1)
rt.render()
rt.mipt.mipt_debug_simd_profile(out, 1)
st = rt.stats()
o = list(out)
rays = st["rays_closest"] + st["rays_shadow"]
print(text, "spp", spp, sys.argv[2:], "extend ms %.2f shadow ms %.2f (instrumented build)" % (st["traverse_ms"], st["shadow_ms"]))
names = {0: "inner step", 2: "leaf phase", 4: "leaf triangle iteration", 6: "object-loop pass", 8: "outer iteration (lanes alive)", 10: "refill", 26: "push at stack depth >= 6", 28: "push at stack depth >= 8", 30: "push at stack depth >= 10 (spill)"}
for k, nm in names.items():
    print("  %-32s wave events %12d  mean active lanes %5.1f  lane events per ray %6.2f" % (nm, o[k], o[k + 1] / max(1, o[k]), o[k + 1] / rays))
print("  inner steps with every descending lane on ONE node: %.1f%% of the steps (%.1f lanes); first lane's node shared by at least half: %.1f%% more"
      % (100.0 * o[32] / max(1, o[0]), o[33] / max(1, o[32]), 100.0 * o[34] / max(1, o[0])))
for k, nm in {16: "shade<1> sub-chunk", 18: "  diffuse vertex", 20: "  environment hit", 22: "  miss / light hit", 24: "  to the slow tier"}.items():
    print("  %-32s wave events %12d  mean active lanes %5.1f  share of the vertices %5.1f%%" % (nm, o[k], o[k + 1] / max(1, o[k]), 100.0 * o[k + 1] / max(1, o[17])))
tot = o[12] + o[13] + o[14]
print("  wave cycles: refill+objects %.1f%%  inner %.1f%%  leaf %.1f%%" % (100 * o[12] / tot, 100 * o[13] / tot, 100 * o[14] / tot))
