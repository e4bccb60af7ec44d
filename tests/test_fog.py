"""fog — single scattering in a uniform or height-exponential medium (fogContribution and
int_exponential, Raytracer.cpp:20-192; the call sites in getColor :275-316, :413-436, :473-486, :557-565, :626).
tests/golden/fog.npz comes from the compiled reference (tests/golden/make_golden.py --fog)."""
import os
import sys

import numpy as np
import pytest

from helpers import assert_bits
from pathtracer_amd import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import FOG_KINDS, all_pixels, fog_scene  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fog.npz")


@pytest.mark.parametrize("kind", FOG_KINDS)
def test_oracle_fog_matches_reference_golden(kind):
    from oracle.binding import Oracle
    g = np.load(GOLD)
    O = Oracle()
    cfg = fog_scene(O, kind)
    rgb, _ = O.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
    assert_bits(rgb, g[kind + "_rgb"], "per-sample radiance")
    assert np.isfinite(rgb).all() and rgb.mean() > 0


def test_oracle_fog_against_live_reference():
    from oracle import binding
    if not binding.ref_available():
        pytest.skip("compiled reference not present")
    outs = []
    for X in (binding.Ref(), binding.Oracle()):
        cfg = fog_scene(X, "rayleigh")
        outs.append(X.getcolor_samples(all_pixels(cfg)[::3], 0, 3)[0])
    assert_bits(outs[1], outs[0], "exponential fog, Rayleigh phase")


def test_scene_file_keeps_the_fog_block(tmp_path):
    H = capi.HostRaytracer()
    fog_scene(H, "exp")
    p = str(tmp_path / "fog.scn")
    H.save_scene(p)
    text = open(p).read()
    assert "fog_density: 0.800000" in text and "fog_type: 1" in text and "fog_absorption_decay: 0.040000" in text


@pytest.mark.gpu
@pytest.mark.parametrize("kind", FOG_KINDS)
def test_gpu_fog_per_sample(kind):
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    cfg = fog_scene(H, kind)
    rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
    same = (rgb.view(np.uint32) == g[kind + "_rgb"].view(np.uint32)).all(-1).mean()
    assert same == 1.0, "bit-identical fraction %.6f, max |err|/white %.3e" % (same, np.abs(rgb - g[kind + "_rgb"]).max() / 196964.7)


@pytest.mark.gpu
def test_gpu_fog_image_against_the_oracle():
    from oracle.binding import Oracle
    outs = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        fog_scene(X, "exp")
        outs.append(X.render_seeded())
    assert_bits(outs[1][0], outs[0][0], "splatted image in exponential fog")
    assert_bits(outs[1][1], outs[0][1], "sample_count")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ghostfog", "glassfog", "exp"])
def test_gpu_queue_forms_agree_and_overflow_falls_back(kind):
    """The three ways the contribution queue can run give the same bits: the wavefront stages (default), the wavefront stages
    with a ring of 2 pending contributions per sample — most samples of a fog scene need more and are then rendered by the
    one-thread-per-sample loop with the reference's 200-entry ring — and that loop alone."""
    g = np.load(GOLD)
    want = g[kind + "_rgb"]
    seen = {}
    for name, opts in (("wavefront", {}), ("ring2", {"queue_ring": 2}), ("thread", {"queue_wavefront": 0})):
        H = capi.HostRaytracer(device=0)
        cfg = fog_scene(H, kind)
        for k, v in opts.items():
            H.set_option(k, v)
        rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
        st = H.stats()
        assert st["pipeline"] == 2
        seen[name] = st["reserved"]
        assert_bits(rgb, want, f"per-sample radiance, fog scene {kind}, {name}")
        img, cnt = H.render()                               # and through the splat
        if name == "wavefront":
            first = (img, cnt)
        else:
            assert_bits(img, first[0], f"image, {name}")
            assert_bits(cnt, first[1], f"weights, {name}")
    assert seen["wavefront"] == 0 and seen["thread"] == 0 and seen["ring2"] > 0, seen


@pytest.mark.gpu
@pytest.mark.parametrize("kind,lambert", [("uniform", 0), ("exp", 0), ("ghostfog", 0), ("glossyfog", 2), ("mirrorfog", 2), ("glassfog", 2)])
def test_gpu_lambert_builds_of_the_logic_stage_in_fog(kind, lambert):
    """Round 4: fog scenes whose materials are all Lambert run builds of the logic stage that inline only the Lambert vertex
    (k_q_logic<.., LAMBERT>); a vertex that is not one abandons its sample to the one-thread-per-sample loop.  `queue_lambert` = 0:
    the general builds on a Lambert scene (same bits as the default, which is the Lambert build there); = 2: the Lambert builds forced
    on scenes with a glossy / mirror / glass mesh — the abandonment path must reproduce the goldens through the fallback."""
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    cfg = fog_scene(H, kind)
    H.set_option("queue_lambert", lambert)
    rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
    st = H.stats()
    assert st["pipeline"] == 2
    assert_bits(rgb, g[kind + "_rgb"], f"per-sample radiance, fog scene {kind}, queue_lambert {lambert}")
    if kind == "glossyfog":
        assert st["reserved"] > 0                           # glossy vertices went through the fallback
