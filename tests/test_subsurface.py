"""subsurface scattering — the probe through a Gaussian disk around the hit point, the reservoir
traversal Scene::get_random_intersection / TriMesh::reservoir_sampling_intersection that picks one of the intersections
along it with one engine draw per intersection (Raytracer.cpp:318-406, Geometry.cpp:339-470, TriangleMesh.cpp:1321-1426).
tests/golden/subsurface.npz comes from the compiled reference (tests/golden/make_golden.py --subsurface)."""
import os
import sys

import numpy as np
import pytest

from helpers import WHITE, assert_bits
from pathtracer_amd import capi, scenes

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import SSS_KINDS, all_pixels, subsurface_scene  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subsurface.npz")


@pytest.mark.parametrize("kind", SSS_KINDS)
def test_oracle_subsurface_matches_reference_golden(kind):
    from oracle.binding import Oracle
    g = np.load(GOLD)
    O = Oracle()
    cfg = subsurface_scene(O, kind)
    rgb, _ = O.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
    assert_bits(rgb, g[kind + "_rgb"], "per-sample radiance")


def test_subsurface_fixture_is_not_vacuous():
    g = np.load(GOLD)
    assert (g["ss_rgb"] != g["plain_rgb"]).any(-1).mean() > 0.2      # the blob covers about a third of the frame


def test_oracle_subsurface_against_live_reference():
    from oracle import binding
    if not binding.ref_available():
        pytest.skip("compiled reference not present")
    outs = []
    for X in (binding.Ref(), binding.Oracle()):
        cfg = subsurface_scene(X, "ssdeep")
        outs.append(X.getcolor_samples(all_pixels(cfg)[::3], 0, 3)[0])
    assert_bits(outs[1], outs[0], "subsurface, depth 8")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", SSS_KINDS)
def test_gpu_subsurface_per_sample(kind):
    """The fp64 exp of the branch (the profile weight `chris`, Raytracer.cpp:381) is glibc's algorithm (csrc/mipt_libm64.h):
    every sample bit for bit."""
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    cfg = subsurface_scene(H, kind)
    rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
    want = g[kind + "_rgb"]
    same = (rgb.view(np.uint32) == want.view(np.uint32)).all(-1).mean()
    err = np.abs(rgb.astype(np.float64) - want).max() / WHITE
    print("%s: bit-identical fraction %.6f, max |err|/white %.3e" % (kind, same, err))
    assert same == 1.0 and err == 0.0, (same, err)


@pytest.mark.gpu
def test_gpu_refuses_subsurface_on_a_sphere():
    """A sphere with material lists is outside the path (textured spheres are refused), so is its subsurface colour."""
    import ctypes as C
    H = capi.HostRaytracer(device=0)
    H.apply_config(scenes.config_c1(16, 16, 1))
    H.add_mesh(scenes.blob_mesh(8))
    H.prepare()
    desc = C.cast(H.host.mh_scene_desc(H.h), C.POINTER(capi.MiptSceneDesc)).contents
    KSUB = 5                                      # position of `subsurface` among the eight lists of mipt_object
    tex = capi.MiptTexture()
    tex.multiplier[0], tex.multiplier[1], tex.multiplier[2] = 0.5, 0.4, 0.3
    sphere = desc.objects[0]
    old = (sphere.n_lists[KSUB], sphere.lists[KSUB])
    sphere.n_lists[KSUB], sphere.lists[KSUB] = 1, C.pointer(tex)
    try:
        rc = H.mipt.mipt_upload_scene(H.ctx, C.byref(desc))
        assert rc == capi.MIPT_ERR_UNSUPPORTED and b"sphere" in H.mipt.mipt_last_error(H.ctx)
    finally:
        sphere.n_lists[KSUB], sphere.lists[KSUB] = old
