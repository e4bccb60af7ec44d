"""ghost objects and the background photo — getColor's contribution queue (Raytracer.cpp:213-238),
the pass-through of ghosts (:522-536), their missing direct term (:547-553), the photo behind camera rays and behind
ghosts (:260-268, :614-621), the showenvmap flag (:629) and shadow rays that ignore ghosts (Geometry.cpp:722).
tests/golden/compositing.npz comes from the compiled reference (tests/golden/make_golden.py --compositing)."""
import os
import struct
import sys

import numpy as np
import pytest

from helpers import assert_bits
from pathtracer_amd import capi, scenes

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import COMPOSITING_KINDS, all_pixels, background_photo, compositing_scene  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compositing.npz")


@pytest.mark.parametrize("kind", COMPOSITING_KINDS)
def test_oracle_compositing_matches_reference_golden(kind):
    from oracle.binding import Oracle
    g = np.load(GOLD)
    O = Oracle()
    cfg = compositing_scene(O, kind)
    rgb, _ = O.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
    assert_bits(rgb, g[kind + "_rgb"], "per-sample radiance")
    if kind == "both":
        img, cnt = O.render_seeded()
        assert_bits(img, g["both_img"], "splatted image")


def test_ghosts_change_the_picture():
    """The fixtures are not vacuous: the photo, the ghost floor and the ghost mesh each change many samples."""
    g = np.load(GOLD)
    assert (g["plane_rgb"] != g["bgonly_rgb"]).any(-1).mean() > 0.2
    assert (g["both_rgb"] != g["plane_rgb"]).any(-1).mean() > 0.1
    assert (g["mesh_rgb"] != g["bgonly_rgb"]).any(-1).mean() > 0.2
    assert (g["planenobg_rgb"] != g["plane_rgb"]).any(-1).mean() > 0.1


def write_bmp(path, rgb8):
    """24-bit uncompressed BMP, rows bottom-up, BGR, padded to 4 bytes."""
    h, w, _ = rgb8.shape
    row = (w * 3 + 3) & ~3
    data = bytearray()
    for i in range(h - 1, -1, -1):
        line = rgb8[i, :, ::-1].tobytes()
        data += line + b"\0" * (row - len(line))
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + len(data), 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(data), 2835, 2835, 0, 0))
        f.write(data)


def test_load_background_like_the_reference(tmp_path):
    """Scene::load_background (Geometry.h:1355-1363): load_image's row order, pow(v/255., gamma) * 196964.699."""
    from oracle import binding
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (13, 21, 3), dtype=np.uint8)
    p = str(tmp_path / "photo.bmp")
    write_bmp(p, img)
    H = capi.HostRaytracer()
    H.load_background(p)
    mine = H.get_background()
    assert mine.shape == (13, 21, 3)
    want = (np.power(img[::-1].astype(np.float64) / 255., np.float64(np.float32(2.2))) * 196964.699).astype(np.float32)
    assert np.allclose(mine, want, rtol=1e-6)
    if binding.ref_available():
        R = binding.Ref()
        R.load_background(p)
        assert_bits(mine, R.get_background(), "Scene::background")
    with pytest.raises(capi.MiptError):
        H.load_background(str(tmp_path / "missing.bmp"))


def test_compositing_against_live_reference():
    from oracle import binding
    if not binding.ref_available():
        pytest.skip("compiled reference not present")
    outs = []
    for X in (binding.Ref(), binding.Oracle()):
        cfg = compositing_scene(X, "both")
        outs.append(X.getcolor_samples(all_pixels(cfg)[::5], 0, 3)[0])
    assert_bits(outs[1], outs[0], "ghost mesh + ghost floor + photo")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", COMPOSITING_KINDS)
def test_gpu_compositing_per_sample(kind):
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    cfg = compositing_scene(H, kind)
    rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
    assert_bits(rgb, g[kind + "_rgb"], "per-sample radiance")


@pytest.mark.gpu
def test_gpu_compositing_image_and_passes():
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    cfg = compositing_scene(H, "both")
    img, cnt = H.render()
    assert_bits(img, g["both_img"], "splatted image")
    assert_bits(cnt, g["both_cnt"], "sample_count")
    st = H.stats()
    assert st["rays_closest"] > cfg.W * cfg.H * cfg.spp and st["rays_shadow"] > 0
    # denoiser inputs of a ghost scene: the last first-depth hit of the sample's contributions (Raytracer.cpp:255-258)
    from oracle.binding import Oracle
    O = Oracle()
    compositing_scene(O, "both")
    want = O.getcolor_samples_aov(all_pixels(cfg), 0, 2)
    got = H.getcolor_samples_aov(all_pixels(cfg), 0, 2)
    for a, b, what in zip(got, want, ("colour", "normalValue", "albedoValue")):
        assert_bits(a, b, what)
    oimg = O.render_denoiser_inputs()
    gimg = H.render_denoiser_inputs()
    for a, b, what in zip(gimg, oimg, ("imagedouble", "sample_count", "albedo sums", "normal sums")):
        assert_bits(a, b, what)
    # a scene without ghosts on the same context goes back to the wavefront pipeline
    H2 = capi.HostRaytracer(device=0)
    cfg2 = compositing_scene(H2, "planenobg")
    H2.set_object_ghost(2, False)
    H2.prepare()
    H2.render()
    assert H2.stats()["pipeline"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [0, 1])
def test_gpu_compositing_depth_zero_and_one(depth):
    """nb_bounces = 0: the camera contribution is dropped by the depth test of the loop head (Raytracer.cpp:240), nothing is
    traced and the sample is black — k_q_begin decides that itself since it pops the camera contribution; nb_bounces = 1: one
    vertex, whose successors die at the same test."""
    from oracle.binding import Oracle
    outs = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        cfg = compositing_scene(X, "both")
        X.set_render(cfg.W, cfg.H, cfg.spp, depth)
        X.prepare()
        outs.append(X.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0])
    assert_bits(outs[1], outs[0], f"per-sample radiance at depth {depth}")
    assert outs[1].any() == (depth > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", COMPOSITING_KINDS)
def test_gpu_logic_stage_tiers_agree(kind):
    """Round 3: the closest-hit-list logic stage runs as a fast tier (Lambert / mirror / dielectric / miss vertices, ghosts up to
    their any-hit request) plus the general build over what it leaves (csrc/mipt_queue_wave.h, FAST).  Both orders of work —
    tiers, general build alone, and the tiers with a two-entry ring (samples abandoned to the 200-entry fallback) — give
    the golden radiance bit for bit.  Round 4: a sample's ring takes `queue_ring` entries of pass memory (default 16): the rings
    of 2, 3 (not a power of two: the wrap is a compare, not a mask) and 32 entries, the largest."""
    g = np.load(GOLD)
    for opts in ({}, {"queue_fast_tier": 0}, {"queue_ring": 2}, {"queue_ring": 3}, {"queue_ring": 32}):
        H = capi.HostRaytracer(device=0)
        cfg = compositing_scene(H, kind)
        for k, v in opts.items():
            H.set_option(k, v)
        rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
        assert_bits(rgb, g[kind + "_rgb"], f"per-sample radiance with {opts or 'the default tiers'}")


@pytest.mark.gpu
@pytest.mark.parametrize("lambert", [0, 2])
def test_gpu_lambert_builds_of_the_logic_stage(lambert):
    """Round 4: the builds of the queue's logic stage that inline only the Lambert vertex (k_q_logic<.., LAMBERT>; chosen at upload for
    scenes whose materials are all Lambert) against the general builds (`queue_lambert` = 0) and forced on every scene (= 2: a glossy
    vertex abandons its sample to the one-thread-per-sample loop), on the seven ghost / photo goldens."""
    g = np.load(GOLD)
    for kind in COMPOSITING_KINDS:
        H = capi.HostRaytracer(device=0)
        cfg = compositing_scene(H, kind)
        H.set_option("queue_lambert", lambert)
        rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
        assert_bits(rgb, g[kind + "_rgb"], f"per-sample radiance, compositing scene {kind}, queue_lambert {lambert}")
