"""SURVEY.md §8 f4 (last clause): the denoiser inputs of render_image_nopreviz's has_denoiser branch — getColor's
normalValue / albedoValue (shading normal and Kd of the first hit, Raytracer.cpp:255-258) and the unsplatted
accumulation (Raytracer.cpp:1631-1645).  tests/golden/denoiser_inputs.npz comes from the compiled reference
(tests/golden/make_golden.py --denoiser-inputs)."""
import os
import sys

import numpy as np
import pytest

from helpers import assert_bits
from pathtracer_amd import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import all_pixels, setup  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "denoiser_inputs.npz")
SCENES = ("textured", "cutout", "glass")


@pytest.mark.parametrize("name", SCENES)
def test_oracle_denoiser_inputs_match_reference_golden(name):
    from oracle.binding import Oracle
    g = np.load(GOLD)
    O = Oracle()
    mesh, cfg, oid = setup(O, name)
    rgb, nrm, alb = O.getcolor_samples_aov(all_pixels(cfg), 0, 2)
    assert_bits(rgb, g[name + "_rgb"], "colour")
    assert_bits(nrm, g[name + "_normal"], "normalValue")
    assert_bits(alb, g[name + "_albedo"], "albedoValue")
    hit = np.abs(g[name + "_normal"]).sum(-1) > 0
    assert 0.5 < hit.mean() <= 1.0                      # every camera ray of these scenes ends on the mesh, the floor or the env sphere
    assert np.allclose(np.linalg.norm(g[name + "_normal"][hit], axis=-1), 1, atol=1e-3)
    if name == "textured":
        img, cnt, a, n = O.render_denoiser_inputs()
        for got, key in ((img, "_img"), (cnt, "_cnt"), (a, "_albedo_sum"), (n, "_normal_sum")):
            assert_bits(got, g[name + key], "accumulated" + key)
        assert (cnt == cfg.spp).all()


def test_oracle_denoiser_inputs_against_live_reference():
    from oracle import binding
    if not binding.ref_available():
        pytest.skip("compiled reference not present")
    outs = []
    for X in (binding.Ref(), binding.Oracle()):
        mesh, cfg, oid = setup(X, "merl")
        outs.append(X.getcolor_samples_aov(all_pixels(cfg)[::7], 0, 3) + X.render_denoiser_inputs())
    for a, b, what in zip(outs[0], outs[1], ("rgb", "normal", "albedo", "img", "cnt", "albedo sum", "normal sum")):
        assert_bits(b, a, what)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_gpu_denoiser_inputs_per_sample(name):
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup(H, name)
    rgb, nrm, alb = H.getcolor_samples_aov(all_pixels(cfg), 0, 2)
    assert_bits(rgb, g[name + "_rgb"], "colour")
    assert_bits(nrm, g[name + "_normal"], "normalValue")
    assert_bits(alb, g[name + "_albedo"], "albedoValue")


@pytest.mark.gpu
def test_gpu_denoiser_accumulation_and_host_mirror():
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup(H, "textured")
    img, cnt, alb, nrm = H.render_denoiser_inputs()
    assert_bits(img, g["textured_img"], "imagedouble (no splat)")
    assert_bits(cnt, g["textured_cnt"], "sample_count")
    assert_bits(alb, g["textured_albedo_sum"], "albedo sums")
    assert_bits(nrm, g["textured_normal_sum"], "normal sums")
    # several passes: the same sums up to the order of the additions
    H.set_option("paths_per_pass", cfg.W * cfg.H * 3)
    img2, cnt2, alb2, nrm2 = H.render_denoiser_inputs()
    H.set_option("paths_per_pass", 1 << 29)
    assert np.array_equal(cnt2, cnt)
    for a, b in ((img2, img), (alb2, alb), (nrm2, nrm)):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-4 * max(1.0, float(np.abs(b).max())))
    # Raytracer::render_image_nopreviz with has_denoiser (Raytracer.cpp:1689-1696)
    H.set_has_denoiser(True)
    mean, cnt3, u8 = H.render_image_nopreviz()
    albedo, normal_ref, normal = H.denoiser_images()
    assert_bits(mean, img / cnt[..., None], "imagedouble / sample_count")
    assert_bits(albedo, alb / cnt[..., None], "albedoImage")
    nn = np.sqrt((nrm[..., 0] * nrm[..., 0] + nrm[..., 1] * nrm[..., 1]) + nrm[..., 2] * nrm[..., 2])
    assert np.allclose(normal, nrm / nn[..., None], atol=1e-6)
    cn = np.sqrt((img[..., 0] * img[..., 0] + img[..., 1] * img[..., 1]) + img[..., 2] * img[..., 2])
    with np.errstate(invalid="ignore"):                 # a black pixel gives 0/0 there, as in the reference
        assert np.allclose(normal_ref, img / cn[..., None], atol=1e-6, equal_nan=True)   # the reference's normalImage: colour sums, normalised (:1680)


def _bare_mirror_scene(X):
    """A MIRROR sphere without material lists in front of the floor and a mesh: the radiance never reads its material, the denoiser's
    albedo input at a first hit on it is the Kd Scene::intersection's one MaterialValues held (Geometry.cpp:596)."""
    from pathtracer_amd import scenes
    cfg = scenes.config_c1(40, 28, 2)
    cfg.nb_bounces = 3
    X.apply_config(cfg)
    m = X.add_mesh(scenes.blob_mesh(10), scale=14.0)
    X.set_group_material(m, 0, (0.7, 0.3, 0.2), (0.1, 0.1, 0.1), (20., 20., 20.))
    X.add_sphere((0, -18, 14), 8.0, mirror=True)
    X.add_sphere((-16, -21, 4), 5.0, mirror=True, flip_normals=True)
    X.prepare()
    return cfg


def test_oracle_albedo_on_a_mirror_sphere_without_lists_against_live_reference():
    from oracle import binding
    if not binding.ref_available():
        pytest.skip("compiled reference not present")
    outs = []
    for X in (binding.Ref(), binding.Oracle()):
        cfg = _bare_mirror_scene(X)
        outs.append(X.getcolor_samples_aov(all_pixels(cfg), 0, 2))
    for a, b, what in zip(outs[0], outs[1], ("rgb", "normal", "albedo")):
        assert_bits(b, a, what)
    assert not (outs[0][2] == 0.5).all(-1).any()                      # never MaterialValues()'s default Kd: the sphere shows the Kd of the floor / the mesh / the environment tested before it


@pytest.mark.gpu
def test_gpu_albedo_on_a_mirror_sphere_without_lists():
    """such a scene keeps the wavefront stages for its radiance (the mirror branch reads no material); the calls that hand out
    the albedo render it the way the reference's loop runs.  Per sample, the accumulated images, and the plain render unchanged."""
    from oracle.binding import Oracle
    O, H = Oracle(), capi.HostRaytracer(device=0)
    cfg = _bare_mirror_scene(O)
    _bare_mirror_scene(H)
    pix = all_pixels(cfg)
    for a, b, what in zip(H.getcolor_samples_aov(pix, 0, 2), O.getcolor_samples_aov(pix, 0, 2), ("rgb", "normal", "albedo")):
        assert_bits(a, b, what)
    for a, b, what in zip(H.render_denoiser_inputs(), O.render_denoiser_inputs(), ("img", "cnt", "albedo sum", "normal sum")):
        assert_bits(a, b, what)
    assert_bits(H.getcolor_samples(pix, 0, 2)[0], O.getcolor_samples(pix, 0, 2)[0], "radiance through the wavefront stages")
    assert H.stats()["pipeline"] == 1                      # the plain render stays on the wavefront pipeline
