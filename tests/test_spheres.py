"""Spheres beside the light (object 0) and the environment (object 1) (Geometry.h:849-992): ordinary scene
objects in front of or behind the meshes in the object list — constant colour, mirror, glossy, image-textured (material
lists looked up at the spherical coordinates of the normalised normal), glass, flipped normals.  A sphere without
material lists (kind `bare`) is shaded by Scene::intersection with the material of the last object before it in the list that the
ray also hit (one MaterialValues for all objects of its loop, Geometry.cpp:596): such scenes run through the one-thread-per-sample
kernel, which walks the objects the way the reference does (csrc/mipt_trace.h scene_intersect_inherit).
tests/golden/spheres.npz comes from the compiled reference (tests/golden/make_golden.py --spheres)."""
import os
import sys

import numpy as np
import pytest

from helpers import WHITE, assert_bits
from pathtracer_amd import capi, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import SPHERE_KINDS, all_pixels, sphere_scene  # noqa: E402

GOLD = os.path.join(HERE, "golden", "spheres.npz")


@pytest.mark.parametrize("kind", SPHERE_KINDS)
def test_oracle_spheres_match_reference_golden(kind):
    from oracle.binding import Oracle
    g = np.load(GOLD)
    O = Oracle()
    cfg = sphere_scene(O, kind)
    rgb, _ = O.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
    assert_bits(rgb, g[kind + "_rgb"], "per-sample radiance")
    assert (rgb != 0).any(-1).mean() > 0.5


def test_sphere_scene_round_trips_through_scn(tmp_path):
    """NEW SPHERE records beyond the first two: saved and loaded like any object."""
    H = capi.HostRaytracer()
    cfg = scenes.config_c1(48, 30, 3)
    H.apply_config(cfg)
    a = H.add_sphere((3, -4, 5), 2.5, mirror=True)
    b = H.add_sphere((-6, 1, 2), 4.0, flip_normals=True)
    H.add_group_material(b, (0.9, 0.2, 0.1), (0.3, 0.3, 0.3), (20, 20, 20), 1.0, 1.3)
    p = str(tmp_path / "s.scn")
    H.save_scene(p)
    H2 = capi.HostRaytracer()
    H2.load_scene(p)
    assert H2.num_objects() == H.num_objects() == 5
    for k in (a, b):
        s1, f1 = H.object_state(k); s2, f2 = H2.object_state(k)
        assert np.allclose(s1, s2, atol=1e-5) and np.array_equal(f1, f2)
    assert np.allclose(H2.group_materials(b)[0][0], H.group_materials(b)[0][0], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", SPHERE_KINDS)
@pytest.mark.parametrize("pipeline", [1, 0])
def test_gpu_spheres_per_sample(kind, pipeline):
    g = np.load(GOLD)
    H = capi.HostRaytracer(device=0)
    cfg = sphere_scene(H, kind)
    H.set_option("pipeline", pipeline)
    rgb, _ = H.sample_radiance(all_pixels(cfg), 0, cfg.spp)
    want = g[kind + "_rgb"]
    assert np.abs(rgb.astype(np.float64) - want).max() / WHITE < 1e-4
    assert_bits(rgb, want, f"per-sample radiance, spheres '{kind}', pipeline {pipeline}")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gpu_random_sphere_scenes_against_oracle(seed):
    """Random spheres (positions, radii, mirror / glass / glossy / textured / plain, list position relative to the mesh),
    with fog on one seed (the contribution-queue pipeline): HIP path vs oracle, bit for bit, and through the splat."""
    from oracle.binding import Oracle
    outs = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        rng = np.random.default_rng(100 + seed)
        cfg = scenes.config_c1(56, 36, 2)
        cfg.nb_bounces = int(rng.integers(2, 6))
        X.apply_config(cfg)
        nsph = int(rng.integers(1, 5))
        mesh_at = int(rng.integers(0, nsph + 1))
        for k in range(nsph + 1):
            if k == mesh_at:
                X.add_mesh(scenes.blob_mesh(12 + 2 * seed), scale=float(rng.uniform(10, 25)))
                continue
            c = (float(rng.uniform(-25, 25)), float(rng.uniform(-26, -5)), float(rng.uniform(-10, 20)))
            kind = int(rng.integers(0, 5))
            o = X.add_sphere(c, float(rng.uniform(2, 9)), mirror=(kind == 1), flip_normals=bool(rng.integers(0, 2)) and kind == 0)
            if kind == 0:
                X.add_group_material(o, tuple(rng.uniform(0.1, 1, 3)), (0, 0, 0), (0, 0, 0), 1.0, 1.3)
            elif kind == 2:
                X.add_group_material(o, (1, 1, 1), (0, 0, 0), (0, 0, 0), 0.0, float(rng.uniform(1.1, 1.8)))
            elif kind == 3:
                X.add_group_material(o, tuple(rng.uniform(0.1, 1, 3)), tuple(rng.uniform(0, 0.5, 3)), tuple(rng.uniform(1, 60, 3)), 1.0, 1.3)
            elif kind == 4:
                X.add_group_material(o, (1, 1, 1), (0, 0, 0), (0, 0, 0), 1.0, 1.3)
                X.set_group_texture(o, 0, 0, scenes.checker_texture(16, 8, seed, 2))
        if seed == 3:
            X.set_fog(0.3, 0.2, 0.02, 0.03, 1, 1, 0.3)
        X.prepare()
        outs.append((X.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0], X.render_seeded()))
    assert_bits(outs[1][0], outs[0][0], "per-sample radiance")
    assert_bits(outs[1][1][1], outs[0][1][1], "splat weights")
    assert_bits(outs[1][1][0], outs[0][1][0], "splatted image")


@pytest.mark.gpu
def test_sphere_without_material_lists_takes_the_material_of_the_object_before_it():
    """Rounds 1 and 2 refused such a scene.  The reference shades the sphere with what the ground plane (object 2, hit by every ray
    that points down) or the environment sphere left in Scene::intersection's MaterialValues: image, splat and the ray API
    (mipt_trace: the same loop) against the oracle, with and without fog; the scene runs on the queue kernel's one-thread form."""
    from oracle.binding import Oracle
    for fog in (False, True):
        outs = []
        for X in (Oracle(), capi.HostRaytracer(device=0)):
            cfg = scenes.config_c1(40, 28, 2)
            cfg.nb_bounces = 4
            X.apply_config(cfg)
            X.add_sphere((0, -20, 5), 9.0)
            X.add_sphere((12, -10, 14), 5.0, flip_normals=True)
            if fog:
                X.set_fog(0.5, 0.4)
            X.prepare()
            px = np.stack(np.meshgrid(np.arange(cfg.H), np.arange(cfg.W), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)
            outs.append((X.getcolor_samples(px, 0, cfg.spp)[0], X.render_seeded()))
            if isinstance(X, capi.HostRaytracer):
                assert X.stats()["pipeline"] == 2
        assert_bits(outs[1][0], outs[0][0], f"per-sample radiance (fog {fog})")
        assert_bits(outs[1][1][0], outs[0][1][0], f"splatted image (fog {fog})")


@pytest.mark.gpu
def test_subsurface_colour_beside_spheres_with_material_lists():
    """Rounds 1-2 refused every extra sphere in a scene with subsurface colours.  Only a sphere WITHOUT lists inherits Ksub (the shared
    MaterialValues of Scene::intersection); a sphere with lists writes Ksub = 0 (Geometry.h:399-445).  HIP path (wavefront stages
    with the subsurface probe) against the oracle; the list-less case (a mirror: getColor reads Ksub before the mirror branch) was
    refused until round 4 and now runs on the one-thread-per-sample kernel, Ksub inherited like Kd / Ks / Ne."""
    from oracle.binding import Oracle
    outs = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        cfg = scenes.config_c1(40, 28, 3)
        cfg.nb_bounces = 4
        X.apply_config(cfg)
        a = X.add_sphere((-14, -16, 6), 7.0)
        X.add_group_material(a, (0.2, 0.6, 0.9), (0, 0, 0), (0, 0, 0), 1.0, 1.3)
        m = X.add_mesh(scenes.blob_mesh(14), scale=18.0)
        X.set_group_subsurface(m, 0, (0.8, 0.5, 0.3))
        b = X.add_sphere((15, -15, 3), 6.0)
        X.add_group_material(b, (0.5, 0.5, 0.1), (0.4, 0.4, 0.4), (50, 50, 50), 1.0, 1.3)
        X.prepare()
        outs.append(X.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0])
    assert_bits(outs[1], outs[0], "per-sample radiance, subsurface mesh between two spheres")
    outs = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        cfg = scenes.config_c1(32, 24, 3)
        X.apply_config(cfg)
        m = X.add_mesh(scenes.blob_mesh(8))
        X.set_group_subsurface(m, 0, (0.8, 0.5, 0.3))
        X.add_sphere((0, -20, 5), 4.0, mirror=True)
        X.add_sphere((12, -18, 9), 5.0)
        X.prepare()
        outs.append(X.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0])
    assert_bits(outs[1], outs[0], "per-sample radiance, spheres without material lists beside a subsurface mesh")
