"""GPU tests at BASELINE.json's full geometry / image size (configs[1]: 133 128 triangles, 1920x1080),
through properties that do not need the oracle to render the whole frame:

* the oracle agrees bit for bit on a random subset of (pixel, sample) pairs of the full-size frame;
* the two schedulers (per-path kernel, wavefront queues) produce the same image bit for bit and cast
  the same number of rays;
* linearity: doubling the light intensity doubles every accumulator exactly (a power-of-two scale is
  exact in floating point, and no control-flow decision of the path depends on the light power);
* the splat weights of a frame do not depend on the scene.
"""
import numpy as np
import pytest

from helpers import WHITE, assert_bits, bits_equal
from pathtracer_amd import capi, scenes

pytestmark = pytest.mark.gpu

SPP = 2


@pytest.fixture(scope="module")
def c1():
    return scenes.blob_mesh(258), scenes.config_c1(1920, 1080, SPP)


def render(mesh, cfg, **opts):
    rt = capi.HostRaytracer(device=0)
    rt.apply_config(cfg)
    rt.add_mesh(mesh)
    rt.prepare()
    for k, v in opts.items():
        rt.set_option(k, v)
    img, cnt = rt.render()
    return rt, img, cnt


def test_oracle_subset_of_full_frame(c1):
    from oracle.binding import Oracle
    mesh, cfg = c1
    rng = np.random.default_rng(11)
    pix = np.stack([rng.integers(0, cfg.H, 1500), rng.integers(0, cfg.W, 1500)], 1).astype(np.int32)
    O = Oracle()
    O.apply_config(cfg)
    O.add_mesh(mesh)
    O.prepare()
    want, want_j = O.getcolor_samples(pix, 0, SPP)
    rt = capi.HostRaytracer(device=0)
    rt.apply_config(cfg)
    rt.add_mesh(mesh)
    rt.prepare()
    for pipeline in (0, 1):
        rt.set_option("pipeline", pipeline)
        got, got_j = rt.sample_radiance(pix, 0, SPP)
        assert_bits(got_j, want_j, "jitter")
        assert_bits(got, want, f"per-sample radiance, pipeline {pipeline}")


def test_pipelines_agree_full_frame(c1):
    mesh, cfg = c1
    rt0, img0, cnt0 = render(mesh, cfg, pipeline=0)
    rt1, img1, cnt1 = render(mesh, cfg, pipeline=1, merge_traverse=1)
    rt2, img2, cnt2 = render(mesh, cfg, pipeline=1, refill=0)
    rt3, img3, cnt3 = render(mesh, cfg, pipeline=1, merge_traverse=0, refill_threshold=8, inner_min=40)
    s0, s1, s2, s3 = rt0.stats(), rt1.stats(), rt2.stats(), rt3.stats()
    for k in ("paths", "rays_closest", "rays_shadow"):
        assert s0[k] == s1[k] == s2[k] == s3[k], k
    assert s0["paths"] == cfg.W * cfg.H * SPP
    assert s1["traverse_merged"] == 1 and s3["traverse_merged"] == 0
    assert s1["traverse_launches"] == cfg.nb_bounces + 1 and s3["traverse_launches"] == cfg.nb_bounces == s3["shadow_launches"]
    assert_bits(img1, img0, "image: wavefront vs per-path")
    assert_bits(img2, img0, "image: wavefront without refill vs per-path")
    assert_bits(img3, img0, "image: wavefront, one launch per queue, other scheduling parameters vs per-path")
    assert_bits(cnt1, cnt0, "weights")
    assert np.isfinite(img0).all()   # (negative terms exist in the reference too: J is not clamped, Raytracer.cpp:545)
    assert 0.01 < (img0 / cnt0[..., None]).mean() / WHITE < 1.0


def test_light_linearity_and_weights(c1):
    mesh, cfg = c1
    import copy
    cfg2 = copy.copy(cfg)
    cfg2.light_scale = 2.0 * cfg.light_scale
    _, img, cnt = render(mesh, cfg)
    _, img2, cnt2 = render(mesh, cfg2)
    assert_bits(img2, 2.0 * img, "image(2 x light) == 2 x image(light)")
    assert_bits(cnt2, cnt, "splat weights do not depend on the light")
    # interior pixels receive the full 3x3 filter mass of SPP samples each from 9 sources
    _, img3, cnt3 = render(scenes.cornell_mesh(), cfg)
    assert_bits(cnt3, cnt, "splat weights do not depend on the scene")


def test_c2_workload_oracle_subset():
    """configs[2] at reduced tessellation (80 000 triangles so the oracle builds in seconds), full-size
    textures, env map and frame: oracle on a random subset of (pixel, sample) pairs, bit for bit (acosf / atan2f of the
    env-map lookup are glibc's algorithms, csrc/mipt_invtrig.h) — the every-pixel test below asserts the same at full size."""
    from oracle.binding import Oracle
    mesh, cfg, mat, _ = scenes.workload("c2", spp=SPP, grid=200)
    rng = np.random.default_rng(12)
    pix = np.stack([rng.integers(0, cfg.H, 1500), rng.integers(0, cfg.W, 1500)], 1).astype(np.int32)
    O = Oracle()
    O.apply_config(cfg)
    scenes.install(O, mesh, mat)
    O.prepare()
    want, want_j = O.getcolor_samples(pix, 0, SPP)
    rt = capi.HostRaytracer(device=0)
    rt.apply_config(cfg)
    scenes.install(rt, mesh, mat)
    rt.prepare()
    got, got_j = rt.sample_radiance(pix, 0, SPP)
    assert_bits(got_j, want_j, "jitter")
    err = np.abs(got.astype(np.float64) - want).max() / WHITE
    assert err < 1e-4, err                                   # north-star tolerance: per-pixel L-inf < 1e-4 on radiance / 196964.7
    assert_bits(got, want, "per-sample radiance")


@pytest.mark.parametrize("wl,grid", [("c3", 200), ("c4", 160)])
def test_c3_c4_workloads_oracle_subset(wl, grid):
    """configs[3] (dielectric, depth 12) and configs[4] (MERL table, thin-lens depth of field, 3840x2160) at reduced
    tessellation but full frame size: oracle on a random subset of (pixel, sample) pairs, both schedulers, bit for bit
    (Schlick's powf, sinf / cosf are the host libm's algorithms; the MERL evaluation is fp64 arithmetic and table reads)."""
    from oracle.binding import Oracle
    mesh, cfg, mat, _ = scenes.workload(wl, spp=SPP, grid=grid)
    rng = np.random.default_rng(13)
    pix = np.stack([rng.integers(0, cfg.H, 1200), rng.integers(0, cfg.W, 1200)], 1).astype(np.int32)
    O = Oracle()
    O.apply_config(cfg)
    scenes.install(O, mesh, mat)
    O.prepare()
    want, want_j = O.getcolor_samples(pix, 0, SPP)
    rt = capi.HostRaytracer(device=0)
    rt.apply_config(cfg)
    scenes.install(rt, mesh, mat)
    rt.prepare()
    for pipeline in (1, 0):
        rt.set_option("pipeline", pipeline)
        got, got_j = rt.sample_radiance(pix, 0, SPP)
        assert_bits(got_j, want_j, "jitter")
        err = np.abs(got.astype(np.float64) - want).max() / WHITE
        assert err < 1e-4, (wl, pipeline, err)              # north-star tolerance
        assert_bits(got, want, f"{wl}: per-sample radiance, pipeline {pipeline}")


@pytest.mark.parametrize("wl", ["c1", "c2", "c3"])
def test_every_pixel_of_the_full_size_frame(wl):
    """configs[1..3] at FULL size (c2: 2 508 800 triangles, 2048x2048 Kd texture, 4096x2048 env map, 1920x1080; c3: the
    same mesh as a dielectric, depth 12) — one sample of EVERY pixel through the oracle (all host threads) against the
    HIP path, bit for bit (2.07 M paths each)."""
    from oracle.binding import Oracle
    mesh, cfg, mat, _ = scenes.workload(wl, spp=1)
    O = Oracle()
    O.apply_config(cfg)
    scenes.install(O, mesh, mat)
    O.prepare()
    pix = np.stack(np.meshgrid(np.arange(cfg.H), np.arange(cfg.W), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)
    want, want_j = O.getcolor_samples(pix, 0, 1)
    rt = capi.HostRaytracer(device=0)
    rt.apply_config(cfg)
    scenes.install(rt, mesh, mat)
    rt.prepare()
    got, got_j = rt.sample_radiance(pix, 0, 1)
    assert_bits(got_j, want_j, "jitter")
    assert np.abs(got.astype(np.float64) - want).max() / WHITE < 1e-4
    assert_bits(got, want, "per-sample radiance of all 2 073 600 pixels")
    assert 0.01 < want.mean() / WHITE < 1.0


@pytest.mark.parametrize("feature", ["fog+ghost+photo", "subsurface"])
def test_every_pixel_full_size_queue_kernel(feature):
    """configs[1]'s scene at 1920x1080 with the features of the contribution-queue kernel: one sample of every pixel through
    the oracle against the HIP path (2.07 M samples each; exponential fog with the Schlick phase function + a ghost floor
    over a background photo, and a subsurface colour on the mesh)."""
    from oracle.binding import Oracle
    mesh, cfg, mat, _ = scenes.workload("c1", spp=1)
    photo = (np.random.default_rng(5).uniform(0, 1, (90, 160, 3)) ** 2.2 * 196964.699).astype(np.float32)
    out = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        X.apply_config(cfg)
        oid = scenes.install(X, mesh, mat)
        if feature == "subsurface":
            X.set_group_subsurface(oid, 0, (0.8, 0.5, 0.3))
        else:
            X.set_object_ghost(2, True)
            X.set_background(photo)
            X.set_fog(0.4, 0.3, 0.02, 0.03, 1, 1, 0.5)
        X.prepare()
        out.append(X)
    pix = np.stack(np.meshgrid(np.arange(cfg.H), np.arange(cfg.W), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)
    want, _ = out[0].getcolor_samples(pix, 0, 1)
    got, _ = out[1].sample_radiance(pix, 0, 1)
    err = np.abs(got.astype(np.float64) - want).max() / WHITE
    assert err < 1e-4, (feature, err)
    assert_bits(got, want, f"per-sample radiance of all 2 073 600 pixels, {feature}")      # the fp64 exp of the subsurface weight is glibc's (csrc/mipt_libm64.h)
    assert out[1].stats()["pipeline"] == 2
    assert 0.01 < want.mean() / WHITE < 2.0


def test_c4_real_geometry_oracle_subset():
    """configs[4] with its REAL geometry: 23 697 288 triangles (15 M nodes: the one config whose BVH exceeds the Infinity
    Cache), the tree built on the GPU (mipt_build_bvh), MERL table, thin-lens depth of field, 3840x2160.  The oracle builds
    its own tree with the reference's serial recursion and evaluates 2 000 random (pixel, sample) pairs; both schedulers
    must return the same radiance bit for bit.  (The oracle's threaded host build of the 23.7 M-triangle tree is most of its ~10 s.)"""
    from oracle.binding import Oracle
    mesh, cfg, mat, text = scenes.workload("c4", spp=SPP)
    assert mesh.ntri > 23_000_000, text
    rng = np.random.default_rng(14)
    pix = np.stack([rng.integers(0, cfg.H, 1000), rng.integers(0, cfg.W, 1000)], 1).astype(np.int32)      # x SPP = 2 000 samples
    capi.set_bvh_builder("gpu", 0)
    try:
        rt = capi.HostRaytracer(device=0)
        rt.apply_config(cfg)
        oid = scenes.install(rt, mesh, mat)
        who, secs, dev_secs = rt.mesh_bvh_builder(oid)
        assert who == "gpu" and dev_secs < 1.0
        rt.prepare()
        got = {}
        for pipeline in (1, 0):
            rt.set_option("pipeline", pipeline)
            got[pipeline] = rt.sample_radiance(pix, 0, SPP)
        rt.close()
    finally:
        capi.set_bvh_builder("auto", 0)
    O = Oracle()
    O.apply_config(cfg)
    scenes.install(O, mesh, mat)
    O.prepare()
    want, want_j = O.getcolor_samples(pix, 0, SPP)
    for pipeline in (1, 0):
        g, gj = got[pipeline]
        assert_bits(gj, want_j, "jitter")
        err = np.abs(g.astype(np.float64) - want).max() / WHITE
        assert err < 1e-4, (pipeline, err)
        assert_bits(g, want, f"c4 real geometry: per-sample radiance, pipeline {pipeline}")
    assert 0.005 < want.mean() / WHITE < 2.0
