"""TEST INFRASTRUCTURE (uses the oracle): what would an ORDER-FREE any-hit traversal cost?

TriMesh::intersection_shadow (TriangleMesh.cpp:1239-1319) answers "is there a reachable triangle with t < 0.999 dist"; the
oracle's diagnostic (pt_oracle.c, anyhit_study) walks every shadow ray of the sampled paths a second, third ... time in
other visiting orders and counts node fetches, leaves and triangle tests beside the reference's own.

usage: python tests/tools/anyhit_study.py <c1|c2|c3|c4> [width=240] [height=135] [spp=2] [grid] [local_height=6] [nb_bounces]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle          # noqa: E402
from pathtracer_amd import scenes           # noqa: E402

wl = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 240
Hh = int(sys.argv[3]) if len(sys.argv) > 3 else 135
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 2
grid = int(sys.argv[5]) if len(sys.argv) > 5 and int(sys.argv[5]) > 0 else None
local_height = int(sys.argv[6]) if len(sys.argv) > 6 else 6
mesh, cfg, mat, text = scenes.workload(wl, W, Hh, spp, grid)
if len(sys.argv) > 7:
    cfg.nb_bounces = int(sys.argv[7])          # 1: the shadow rays of the camera rays' hits only
O = Oracle()
O.apply_config(cfg)
scenes.install(O, mesh, mat)
O.prepare()
O.cdll.o_anyhit_study(1)
O.cdll.o_anyhit_study_local_height(local_height)
ij = np.ascontiguousarray(np.array([(i, j) for i in range(cfg.H) for j in range(cfg.W)], np.int32))
out = np.zeros((ij.shape[0] * spp, 3), np.float32)
f = O.cdll.o_getcolor_samples
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
f(O.ctx, ij.shape[0], ij.ctypes.data, 0, spp, out.ctypes.data, None)
c = np.zeros(24, np.uint64)
O.cdll.o_anyhit_study_get(c.ctypes.data_as(C.c_void_p))
q4 = np.zeros(6, np.uint64)
O.cdll.o_anyhit_study_get_q4(q4.ctypes.data_as(C.c_void_p))
q4 = [int(x) for x in q4]
loc = np.zeros(10, np.uint64)
O.cdll.o_anyhit_study_get_local(loc.ctypes.data_as(C.c_void_p))
loc = [int(x) for x in loc]
O.cdll.o_anyhit_study(0)
c = [int(x) for x in c]
n = max(c[0], 1)
r = lambda x: round(x / n, 3)
print(json.dumps({
    "scene": text, "frame": "%dx%dx%d" % (W, Hh, spp), "shadow_rays_on_mesh": c[0], "occluded_fraction": r(c[23]),
    "reference_ordered": {"inner": r(c[1]), "leaves": r(c[2]), "triangles": r(c[3]), "box_tests": r(c[22]), "rounds": r(c[1] + c[2])},
    "unordered_near_first": {"inner": r(c[4]), "leaves": r(c[5]), "triangles": r(c[6]), "rounds": r(c[4] + c[5]), "differing_unflagged": c[7]},
    "unordered_left_first": {"inner": r(c[8]), "leaves": r(c[9]), "triangles": r(c[10]), "rounds": r(c[8] + c[9])},
    "four_wide_nearest_first": {"wide_steps": r(c[11]), "leaves": r(c[12]), "triangles": r(c[13]), "slot_tests": r(c[21]), "rounds": r(c[11] + c[12]), "differing_unflagged": c[14], "max_stack": c[20]},
    "four_wide_first_slot": {"wide_steps": r(c[15]), "leaves": r(c[16]), "triangles": r(c[17]), "rounds": r(c[15] + c[16])},
    "four_wide_8bit_boxes_first_slot": {"wide_steps": r(q4[0]), "leaves": r(q4[1]), "triangles": r(q4[2]), "rounds": r(q4[0] + q4[1]), "occluders_in_leaves_the_reference_does_not_reach": q4[3], "differing": q4[4]},
    "four_wide_local_entry (walk starts %d binary levels above the leaf next to the origin, from the root only after a local miss; per ray that enters the mesh's box)" % local_height:
        {"rays": loc[0], "decided_by_the_local_walk": round(loc[1] / max(loc[0], 1), 3), "wide_steps_local": round(loc[2] / max(loc[0], 1), 2), "wide_steps_from_the_root": round(loc[5] / max(loc[0], 1), 2),
         "four_wide_first_slot_on_the_same_rays": round(c[15] / max(loc[0], 1), 2), "differing": loc[8]},
    "flagged_rays": c[18], "flagged_and_occluded (replayed)": c[19],
}, indent=1))
