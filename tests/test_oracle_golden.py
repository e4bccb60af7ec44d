"""CPU tests: oracle/pt_oracle.c against the golden vectors generated from the compiled reference
(tests/golden/make_golden.py).  These run everywhere, including the GPU box where the reference
itself does not exist.  Everything is compared BIT FOR BIT."""
import numpy as np
import pytest

from helpers import assert_bits, check_scene_against, load_golden, setup_scene
from oracle.binding import Oracle


def test_leaf_functions():
    g = load_golden("leaf_functions.npz")
    O = Oracle()
    for s in (0, 1, 42, 123456789012345):
        assert_bits(O.pcg32(s, 16), g[f"pcg32_{s}"], f"pcg32({s})")
    assert_bits(O.lattice(64), g["lattice"], "extensibleLattice2d")
    assert_bits(O.invsqroot(g["invsqroot_in"]), g["invsqroot_out"], "invSqRoot")
    assert_bits(O.fast_exp(g["fast_exp_in"]), g["fast_exp_out"], "fast_exp")
    assert_bits(O.random_cos(g["random_cos_N"], g["random_cos_r"]), g["random_cos_out"], "random_cos")
    assert_bits(O.phong_sample(g["phong_mat9"], g["phong_wo"], g["phong_N"], g["phong_r12"], g["phong_seeds"]),
                g["phong_sample"], "PhongBRDF::sample")
    assert_bits(O.phong_eval(g["phong_mat9"], g["phong_wi"], g["phong_wo"], g["phong_N"]), g["phong_eval"], "PhongBRDF::eval")


@pytest.mark.parametrize("name", ["cornell", "blob32", "glossy", "glass", "textured", "cutout", "merl"])
def test_scene(name):
    g = load_golden(f"scene_{name}.npz")
    O = Oracle()
    O._scene = setup_scene(O, name)
    check_scene_against(O, g, name, "oracle")


def test_c0_full_image():
    """BASELINE.json configs[0] (12-triangle Cornell, 256x256, 64 spp, depth 4): whole splatted
    image, bit for bit."""
    g = load_golden("c0_image.npz")
    O = Oracle()
    O._scene = setup_scene(O, "c0full")
    img, cnt = O.render_seeded()
    assert_bits(cnt, g["count"], "c0 splat weights")
    assert_bits(img, g["image"], "c0 image")


def test_omp_schedule_matches_serial():
    """The nopreviz-style threaded schedule (used for the CPU baseline) only changes the float
    summation order of the splat."""
    O = Oracle()
    mesh, cfg, oid = O._scene = setup_scene(O, "blob32")
    img, cnt = O.render_seeded()
    t, img2, cnt2, rays = O.render_omp(4)
    assert rays[0] > 0 and rays[1] > 0
    np.testing.assert_allclose(img2, img, rtol=2e-5, atol=1e-2)
    np.testing.assert_allclose(cnt2, cnt, rtol=2e-5, atol=1e-6)
