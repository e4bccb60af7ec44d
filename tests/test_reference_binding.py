"""The drop-in boundary on the REFERENCE'S OWN classes (SURVEY.md §8 row b).

oracle/_ref/libptref_mipt.so is nbonneel/pathtracer itself — its Raytracer.cpp / Geometry.cpp / TriangleMesh.cpp from
/root/reference, compiled by oracle/Makefile (target `refmipt`) — with the USE_MIPT switch of integration/use_mipt applied:
Raytracer::render_image(), Raytracer::render_image_nopreviz(), Scene::intersection() and TriMesh::build_bvh() keep their
signatures and run on pathtracer_amd/libmipt.so.  The tests drive those members through the same harness that generated the
golden vectors (oracle/ref_harness.cpp) and hold the results to the goldens of the unmodified reference.

CPU part (no GPU): the switch applies to the reference as it is today, the library loads, and without a device the binding
reports MIPT_ERR_NO_DEVICE and renders nothing (it never renders a frame on the CPU behind the caller's back)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from helpers import ROOT, WHITE, all_pixels, assert_bits, load_golden, setup_scene
from oracle import binding

needs_lib = pytest.mark.skipif(not binding.ref_mipt_available(), reason="oracle/_ref/libptref_mipt.so not built (needs /root/reference at build time)")
MIPT_OK, MIPT_ERR_NO_DEVICE, MIPT_ERR_UNSUPPORTED = 0, 2, 4
USE_MIPT = os.path.join(ROOT, "integration", "use_mipt")


def normalised(img, cnt):
    return img / np.maximum(cnt, 1e-30)[..., None] / WHITE


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout exists in the build container only")
def test_switch_applies_to_the_reference_as_it_is():
    """Every anchor of apply.sh exists exactly once; a second application is refused; only `#ifdef USE_MIPT` blocks are added."""
    d = tempfile.mkdtemp(prefix="use_mipt_")
    try:
        for f in ("Raytracer.h", "Raytracer.cpp", "Geometry.h", "Geometry.cpp", "TriangleMesh.cpp"):
            shutil.copy(os.path.join("/root/reference", f), d)
            os.chmod(os.path.join(d, f), 0o644)
        r = subprocess.run(["sh", os.path.join(USE_MIPT, "apply.sh"), d], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for f in ("Raytracer.h", "Raytracer.cpp", "Geometry.h", "Geometry.cpp", "TriangleMesh.cpp"):
            before = open(os.path.join("/root/reference", f), "rb").read().decode("latin-1").splitlines()
            after = open(os.path.join(d, f), "rb").read().decode("latin-1").splitlines()
            # removing every #ifdef USE_MIPT ... #endif block (keeping its #else branch) gives the file back
            out, depth, in_else = [], 0, False
            for line in after:
                if line.strip() == "#ifdef USE_MIPT":
                    depth, in_else = 1, False
                elif depth and line.strip() == "#else":
                    in_else = True
                elif depth and line.strip() == "#endif":
                    depth = 0
                elif not depth or in_else:
                    out.append(line)
            assert out == before, f
        r2 = subprocess.run(["sh", os.path.join(USE_MIPT, "apply.sh"), d], capture_output=True, text=True)
        assert r2.returncode != 0 and "already" in r2.stderr
    finally:
        shutil.rmtree(d)


@needs_lib
def test_library_loads_and_refuses_to_render_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU half")
    X = binding.RefMipt()
    scene = setup_scene(X, "cornell")          # TriMesh::build_bvh: mipt_build_bvh says NO_DEVICE -> the reference's own recursion
    X._scene = scene
    g = load_golden("scene_cornell.npz")
    d = X.mesh_dump(scene[2])
    for key in ("perm", "nodes_i", "nodes_bb"):
        assert_bits(d[key], g[key], key)
    X.prepare()
    assert X.upload() == MIPT_ERR_NO_DEVICE and not X.resident()
    t, img = X.time_render_nopreviz(1)
    assert X.status() == MIPT_ERR_NO_DEVICE
    assert not img.any(), "the binding rendered on the CPU although the library reported no device"
    hi, hf = X.intersect(g["rays"][:64])       # not resident: Scene::intersection is the reference's own loop
    assert_bits(hi, g["hit_i"][:64], "hit ids")


# ------------------------------------------------------------------------------------------------ GPU
def ref_on_gpu(name):
    X = binding.RefMipt()
    scene = setup_scene(X, name)
    X._scene = scene
    return X, scene


@needs_lib
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "blob32", "glossy", "glass", "textured", "cutout", "merl"])
def test_reference_classes_render_on_the_library(name):
    g = load_golden(f"scene_{name}.npz")
    X, (mesh, cfg, oid) = ref_on_gpu(name)
    # TriMesh::build_bvh ran mipt_build_bvh: the reference's tree, triangle order and Triangle records
    d = X.mesh_dump(oid)
    for key in ("perm", "nodes_i", "nodes_bb", "groups", "root_bb"):
        assert_bits(d[key], g[key], f"mesh.{key}")
    assert_bits(d["soup"][:, :16], g["soup16"], "triangleSoup")
    # Raytracer::render_image_nopreviz(): imagedouble / sample_count, tone-mapped frame
    t, img = X.time_render_nopreviz(4)
    assert X.status() == MIPT_OK, X.error()
    st = X.stats()
    assert st["paths"] == cfg.W * cfg.H * cfg.spp and st["pipeline"] == 1
    ref = g["image"] / g["count"][..., None]
    assert_bits(img, ref, "render_image_nopreviz: imagedouble")
    u8 = X.image_u8()
    expect_u8 = np.minimum(255., np.maximum(0., 255. * np.power(ref.astype(np.float64) / 196964.7, np.float64(1 / np.float32(2.2))))).astype(np.uint8)
    assert np.abs(u8.astype(int) - expect_u8.astype(int)).max() <= 1
    assert (u8 == expect_u8).mean() > 0.999
    # Raytracer::render_image(): one publish per sample, sums in sample-major order
    t, img2, cnt2 = X.time_render_image(4)
    assert X.status() == MIPT_OK, X.error()
    np.testing.assert_allclose(cnt2, g["count"], rtol=2e-6)
    assert np.abs(normalised(img2, cnt2) - normalised(g["image"], g["count"])).max() < 1e-5
    # Scene::intersection on the resident scene = mipt_trace, one ray per call (mouse picking)
    assert X.resident()
    n = 512
    hi, hf = X.intersect(g["rays"][:n])
    assert_bits(hi[:, 0], g["hit_i"][:n, 0], "has_inter")
    hit = hi[:, 0] == 1
    assert_bits(hi[hit, 1], g["hit_i"][:n][hit, 1], "object id")
    mesh_hit = hit & (hi[:, 1] >= 3)
    assert_bits(hi[mesh_hit, 2], g["hit_i"][:n][mesh_hit, 2], "triangle id")
    assert_bits(hf[hit, :7], g["hit_f"][:n][hit, :7], "t / P / shadingN")
    shaded = hit & (hi[:, 1] >= 2)
    assert_bits(hf[shaded, 7:19], g["hit_f"][:n][shaded, 7:19], "material")


@needs_lib
@pytest.mark.gpu
def test_c0_frame_through_the_reference_classes():
    """BASELINE.json configs[0] at full size through Raytracer::render_image_nopreviz()."""
    g = load_golden("c0_image.npz")
    X, (mesh, cfg, oid) = ref_on_gpu("c0full")
    t, img = X.time_render_nopreviz(4)
    assert X.status() == MIPT_OK, X.error()
    assert_bits(img, g["image"] / g["count"][..., None], "C0 frame")


@needs_lib
@pytest.mark.gpu
def test_bvh_figures_of_the_device_build():
    """The numbers the GUI prints about a tree (mainApp.cpp:974) are those of the reference's recursion."""
    X, (mesh, cfg, oid) = ref_on_gpu("blob32")
    R = binding.Ref()
    setup_scene(R, "blob32")
    assert_bits(X.mesh_bvh_figures(oid)[1:], R.mesh_bvh_figures(oid)[1:], "avg depth / node count / largest leaf")
    assert X.mesh_bvh_figures(oid)[0] >= 1          # (the reference never initialises bvh_depth: its own value is garbage-or-max)


@needs_lib
@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [1, 8, 0])
def test_stop_render_between_two_samples(lookahead):
    """stopRender() from another thread while render_image() runs: the call returns with exactly the published sums."""
    X, (mesh, cfg, oid) = ref_on_gpu("blob32")
    X.set_render(cfg.W, cfg.H, 64, cfg.nb_bounces, cfg.sigma_filter)
    X.set_lookahead(lookahead)
    it, img, cnt = X.render_image_stop_at(5)
    assert X.status() == 6, "MIPT_ERR_CANCELLED expected"
    assert 5 <= it < 64, it
    # the same frame rendered to exactly `it` samples (the tables do not depend on nrays)
    Y, _ = ref_on_gpu("blob32")
    Y.set_render(cfg.W, cfg.H, it, cfg.nb_bounces, cfg.sigma_filter)
    Y.set_lookahead(lookahead)
    t, img2, cnt2 = Y.time_render_image(4)
    assert_bits(cnt, cnt2, "sample_count after the stop")
    assert_bits(img, img2, "imagedouble after the stop")


@needs_lib
@pytest.mark.gpu
def test_denoiser_branch_of_nopreviz():
    """has_denoiser (Raytracer.cpp:1631-1645, 1676-1696) through the reference's members against the reference's own loop."""
    X, (mesh, cfg, oid) = ref_on_gpu("cornell")
    X.set_render(32, 32, 4, cfg.nb_bounces, cfg.sigma_filter)
    img, cnt, alb, nrm = X.render_nopreviz_denoiser()
    assert X.status() == MIPT_OK, X.error()
    R = binding.Ref()
    setup_scene(R, "cornell")
    R.set_render(32, 32, 4, cfg.nb_bounces, cfg.sigma_filter)
    R.prepare()
    rimg, rcnt, ralb, rnrm = R.render_denoiser_inputs()
    assert_bits(cnt, rcnt, "sample_count")
    assert_bits(img, rimg / rcnt[..., None], "colour")
    assert_bits(alb, ralb / rcnt[..., None], "albedo")
    norm = np.sqrt(rnrm[..., 0] * rnrm[..., 0] + rnrm[..., 1] * rnrm[..., 1] + rnrm[..., 2] * rnrm[..., 2])
    with np.errstate(invalid="ignore", divide="ignore"):
        assert_bits(nrm, rnrm / norm[..., None], "normal")


@needs_lib
@pytest.mark.gpu
def test_unchanged_scene_is_not_uploaded_again_and_an_edit_is():
    X, (mesh, cfg, oid) = ref_on_gpu("blob32")
    t, img = X.time_render_nopreviz(4)
    import time
    t0 = time.perf_counter(); X.prepare(); assert X.upload() == MIPT_OK; t_same = time.perf_counter() - t0
    X.set_group_material(oid, 0, (0.9, 0.1, 0.1), (0, 0, 0), (0, 0, 0))
    t, img2 = X.time_render_nopreviz(4)
    assert X.status() == MIPT_OK
    assert np.abs(img2 - img).max() > 1.0, "the edited material did not reach the device"
    print(f"upload of an unchanged description: {t_same * 1e3:.2f} ms")


@needs_lib
@pytest.mark.gpu
def test_scene_outside_the_hot_path_keeps_the_stock_loop():
    """A cylinder (Geometry.h:731): mipt_upload answers MIPT_ERR_UNSUPPORTED and render_image_nopreviz() is the reference's own."""
    X, (mesh, cfg, oid) = ref_on_gpu("cornell")
    X.set_render(24, 24, 2, cfg.nb_bounces, cfg.sigma_filter)
    X.add_cylinder((0, -5, 0), (0, 5, 0), 2.0)
    t, img = X.time_render_nopreviz(2)
    assert X.status() == MIPT_ERR_UNSUPPORTED and not X.resident()
    assert np.isfinite(img).all() and img.any()
