"""CPU check (oracle only, no GPU) of the argument the order-free any-hit stage rests on (csrc/mipt_anyhit.h):

  TriMesh::intersection_shadow's answer (TriangleMesh.cpp:1239-1319) does not depend on the visiting order, the `tnear > t` prune or the
  children's own box tests, EXCEPT for rays that pass a box within 0.2 % of their far end — whatever the order, whether two tree levels are
  taken per step (grandchildren tested directly) and whether the boxes are the float ones or 8-bit ones rounded outwards, as long as an
  occluder only counts when its leaf's own float box is reached.

The oracle's diagnostic (pt_oracle.c, anyhit_study) walks every shadow ray of the sampled paths again in those orders beside the reference's
own and counts the rays whose answer differs without the ray being flagged.  It must be zero; the flagged rays are the ones the GPU stage
hands to the ordered kernel."""
import ctypes as C

import numpy as np
import pytest

from helpers import all_pixels
from oracle.binding import Oracle
from pathtracer_amd import scenes


def study(cfg, meshes):
    O = Oracle()
    O.apply_config(cfg)
    for mesh, kw in meshes:
        O.add_mesh(mesh, **kw)
    O.prepare()
    O.cdll.o_anyhit_study(1)
    O.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
    c = np.zeros(24, np.uint64); q4 = np.zeros(6, np.uint64)
    O.cdll.o_anyhit_study_get(c.ctypes.data_as(C.c_void_p))
    O.cdll.o_anyhit_study_get_q4(q4.ctypes.data_as(C.c_void_p))
    O.cdll.o_anyhit_study(0)
    return [int(x) for x in c], [int(x) for x in q4]


@pytest.mark.parametrize("case", ["blob", "fat_leaves_two_meshes", "shell_at_the_far_end"])
def test_answer_does_not_depend_on_the_order(case):
    cfg = scenes.config_c1(64, 40, 4)
    if case == "blob":
        meshes = [(scenes.blob_mesh(48, fine_detail=True), {})]
    elif case == "fat_leaves_two_meshes":
        cfg.nb_bounces = 5
        meshes = [(scenes.fat_leaf_mesh(), dict(scale=30.0)), (scenes.blob_mesh(24, fine_detail=True), dict(scale=14.0))]
    else:
        from test_anyhit import shell_around
        shell = shell_around((cfg.light_center[0], 0.0, cfg.light_center[2]), cfg.light_radius + 0.04, 96)
        cfg.light_center = (cfg.light_center[0], float(-27.3 - shell.vertices[:, 1].min()), cfg.light_center[2])
        meshes = [(scenes.blob_mesh(32), {}), (shell, dict(scale=1.0, center=False))]
    c, q4 = study(cfg, meshes)
    assert c[0] > 1000, "the scene casts shadow rays at meshes"
    assert c[7] == 0, "order-free binary traversal: an unflagged ray whose answer differs from the reference's"
    assert c[14] == 0, "four-wide traversal (grandchildren tested directly): an unflagged ray differs"
    assert q4[4] == 0, "four-wide traversal on 8-bit boxes with the leaf's float box verified: a ray differs"
    # two levels per step roughly halve the dependent rounds of a ray; 8-bit boxes cost a few per cent more steps than float ones
    assert c[11] + c[12] < 0.7 * (c[1] + c[2])
    assert q4[0] <= 1.1 * c[15] + 10
    if case == "shell_at_the_far_end":
        assert c[18] > 100 and c[19] > 10, "this scene was built to put occluders on either side of 0.999 dist: flagged rays, some of them occluded"
