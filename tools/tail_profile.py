"""How a launch of the closest-hit kernel ramps up and drains (profiling build: tools/tail_profile.sh): per wave its start and end time.
usage: python tools/tail_profile.py [c2|c3] [depth=1]      prints, for the whole frame (1 rank) and for rank 0 of 8:
the launch's duration, the share of wave-time lost before the waves start and after they end, and when the last 10 % / 1 % of the waves end."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIPT_LIB_OVERRIDE"] = os.path.join(ROOT, "pathtracer_amd", "libmipt_tail.so")
from pathtracer_amd import capi, scenes
wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("depth") else "c2"
depth = 1
for a in sys.argv[1:]:
    if a.startswith("depth="): depth = int(a[6:])
mesh, cfg, mat, text = scenes.workload(wl, 1920, 1080, 1024, None)
H = capi.HostRaytracer(device=0)
H.apply_config(cfg); scenes.install(H, mesh, mat); H.prepare()
NW = 16384
buf = (C.c_ulonglong * (2 * NW))()
f = H.mipt.mipt_debug_tail_profile
f.argtypes = [C.c_void_p, C.c_int, C.c_int]
for nr in (1, 8):
    pr = H.params
    pr.tile_size, pr.tile_rank, pr.tile_nranks = 32, 0, nr
    H.render()
    f(buf, NW, depth)                      # clear, choose the depth
    H.render()
    st = H.stats()
    f(buf, NW, depth)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 2).astype(np.float64)
    a = a[a[:, 1] > 0] * 0.01              # microseconds (100 MHz)
    t0, t1 = a[:, 0].min(), a[:, 1].max()
    dur = t1 - t0
    ends = np.sort(t1 - a[:, 1])           # how long before the end of the launch a wave ended
    starts = a[:, 0] - t0
    print("%s, %d rank(s), depth %d: %d waves, launch %.0f us (the LAST pass's launch at this depth; extend stage %.1f ms for %d launches)" % (wl, nr, depth, a.shape[0], dur, st["traverse_ms"], st["traverse_launches"]))
    print("   wave-time lost before a wave starts: %.2f %% of the launch (mean start +%.1f us, last +%.1f us)" % (100 * starts.mean() / dur, starts.mean(), starts.max()))
    print("   wave-time lost after a wave ends:    %.2f %% of the launch (mean %.1f us = the drain; 50 %% of the waves end within the last %.0f us, 10 %% before -%.0f us)" % (100 * ends.mean() / dur, ends.mean(), np.percentile(ends, 50), np.percentile(ends, 90)))
    print("   last waves: the final 1 %% of the waves end in the last %.0f us, the final 0.1 %% in the last %.0f us" % (np.percentile(ends, 1), np.percentile(ends, 0.1)))
