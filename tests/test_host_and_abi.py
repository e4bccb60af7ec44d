"""CPU tests of the PRODUCT's host side (no GPU needed, no oracle involved):

* libmipt.so loads and exports every symbol include/mipt.h declares; without a GPU mipt_create
  reports MIPT_ERR_NO_DEVICE (there is no CPU fallback to fall into);
* the host mirror (TriMesh::init + BVH build, build_matrix, prepare_render tables, light constants,
  loadScene camera) reproduces the golden vectors generated from the compiled reference, bit for bit.
"""
import ctypes as C
import re
import os

import numpy as np
import pytest

from helpers import ROOT, assert_bits, load_golden, setup_scene
from pathtracer_amd import capi, scenes


def test_header_symbols_exported():
    mipt, host = capi.load()
    header = open(os.path.join(ROOT, "include", "mipt.h")).read()
    declared = set(re.findall(r"\b(mipt_[a-z_]+)\s*\(", header)) - {"mipt_progress_cb"}
    assert declared == set(capi.MIPT_SYMBOLS), declared ^ set(capi.MIPT_SYMBOLS)
    for s in declared:
        assert hasattr(mipt, s), s
    assert mipt.mipt_abi_version() == 3


def test_no_device_means_error_not_fallback():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    mipt, _ = capi.load()
    ctx = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert mipt.mipt_create(dev, 1, C.byref(ctx)) == capi.MIPT_ERR_NO_DEVICE
    assert not ctx.value
    two = (C.c_int * 2)(0, 1)
    assert mipt.mipt_create(two, 2, C.byref(ctx)) == capi.MIPT_ERR_NO_DEVICE and not ctx.value      # a group needs its devices too
    assert mipt.mipt_create(two, 0, C.byref(ctx)) == 1
    assert mipt.mipt_group_size(None) == 0 and mipt.mipt_group_reduce_kind(None) == b""
    with pytest.raises(capi.MiptError):
        capi.HostRaytracer(device=0)
    rt = capi.HostRaytracer()          # host-only use is fine ...
    rt.apply_config(scenes.config_c0())
    rt.add_mesh(scenes.cornell_mesh())
    rt.prepare()
    with pytest.raises(capi.MiptError):  # ... but nothing renders without the device
        rt.render()
    with pytest.raises(capi.MiptError):
        rt.render_image_nopreviz()


@pytest.mark.parametrize("name", ["cornell", "blob32", "glossy", "textured", "cutout"])
def test_host_side_matches_reference_goldens(name):
    g = load_golden(f"scene_{name}.npz")
    H = capi.HostRaytracer()
    mesh, cfg, oid = setup_scene(H, name)
    assert_bits(H.light(), g["light"], "light constants")
    rpp, s2d, fi, fs = H.tables()
    assert_bits(rpp, g["randomPerPixel"], "randomPerPixel")
    assert_bits(s2d, g["samples2d"], "samples2d")
    assert_bits(fi, g["filter_integral"], "filter_integral")
    assert fs == int(g["filter_size"])
    for k in range(oid + 1):
        for arr, key in zip(H.object_matrices(k), ("trans", "inv", "rot")):
            assert_bits(arr, g[f"obj{k}_{key}"], f"obj{k}.{key}")
    d = H.mesh_dump(oid)
    for key in ("perm", "nodes_i", "nodes_bb", "groups", "root_bb"):
        assert_bits(d[key], g[key], "mesh." + key)
    assert_bits(d["soup"][:, :16], g["soup16"], "triangleSoup intersection record")
    assert_bits(d["soup"][:, 22:31], g["soup_normals"], "triangleSoup normals")


def test_loadscene_camera():
    import ctypes
    mipt, host = capi.load()
    H = capi.HostRaytracer()
    H.set_render(8, 8, 1, 1)
    H.prepare()
    p = ctypes.cast(H.render_params, ctypes.POINTER(ctypes.c_float))
    # mipt_render_params: 4 ints, then cam_position[3], cam_direction[3], cam_up[3], fov, focus, aperture
    cam = np.array([p[4 + k] for k in range(12)], np.float32)
    d, u = scenes.default_camera_rotated()
    assert_bits(cam[3:6], np.asarray(d, np.float32), "camera direction after cam.rotate")
    assert_bits(cam[6:9], np.asarray(u, np.float32), "camera up after cam.rotate")
    assert_bits(cam[9:10], np.asarray([scenes.RenderConfig().fov], np.float32), "fov")


def test_render_params_layout():
    """The ctypes mirror of mipt_render_params matches the C struct (checked through the values the
    host side wrote into it)."""
    H = capi.HostRaytracer()
    cfg = scenes.config_c0()
    H.apply_config(cfg)
    H.set_partition(16, 1, 3)
    H.add_mesh(scenes.cornell_mesh())
    H.prepare()
    P = H.params
    assert (P.W, P.H, P.nrays, P.nb_bounces) == (cfg.W, cfg.H, cfg.spp, cfg.nb_bounces)
    assert tuple(P.cam_position) == tuple(np.float32(cfg.cam_pos))
    assert P.cam_aperture == np.float32(cfg.aperture) and P.filter_size == 1 and P.sigma_filter == 0.5
    assert P.seed_stride == 65536 and (P.sample_begin, P.sample_end) == (0, cfg.spp)
    assert (P.tile_size, P.tile_rank, P.tile_nranks) == (16, 1, 3)
    assert_bits(np.array(list(P.centerLight) + [P.radiusLight, P.lightPower], np.float32), H.light(), "light block")


def test_parallel_bvh_build_gives_the_serial_tree():
    """The host mirror (std::thread) and the oracle (OpenMP tasks) fork large subtrees and cost the 16
    candidate planes of large nodes concurrently; the tree, node order and triangle permutation must be
    those of the serial build (which the goldens pin to the reference)."""
    from oracle.binding import Oracle
    mipt, host = capi.load()
    mesh = scenes.blob_mesh(48, with_uv=True)          # 4608 triangles
    cfg = scenes.config_c1(16, 16, 1)
    dumps = []
    for fork, planes in ((1 << 30, 1 << 30), (64, 512)):
        host.mh_set_build_thresholds(fork, planes)
        H = capi.HostRaytracer(); H.apply_config(cfg)
        dumps.append(("host", fork, H.mesh_dump(H.add_mesh(mesh))))
        O = Oracle(); O.cdll.o_set_build_thresholds(fork, planes); O.apply_config(cfg)
        dumps.append(("oracle", fork, O.mesh_dump(O.add_mesh(mesh))))
    host.mh_set_build_thresholds(1 << 15, 1 << 18)
    O.cdll.o_set_build_thresholds(1 << 15, 1 << 18)
    ref = dumps[0][2]
    assert ref["nodes_i"].shape[0] > 2000
    for who, fork, d in dumps[1:]:
        for key in ("perm", "nodes_i", "nodes_bb", "soup", "root_bb"):
            assert_bits(d[key], ref[key], f"{who} fork={fork} mesh.{key}")


def test_merl_binary_file_reader(tmp_path):
    """IsoMERLBRDF(file): three int32 dimensions + 3 x 90*90*180 doubles (MERLBRDFRead.cpp:212-236); a file with other
    dimensions or a short payload is refused."""
    import ctypes
    mipt, host = capi.load()
    table = scenes.synthetic_merl_table()
    good = tmp_path / "m.binary"
    with open(good, "wb") as f:
        f.write(np.array([90, 90, 180], np.int32).tobytes()); f.write(np.ascontiguousarray(table, np.float64).tobytes())
    H = capi.HostRaytracer()
    H.apply_config(scenes.config_c1(8, 8, 1))
    oid = H.add_mesh(scenes.blob_mesh(6))
    H.set_brdf_merl_file(oid, good)
    H.prepare()
    # the table reaches the C-ABI description bit for bit
    H2 = capi.HostRaytracer()
    H2.apply_config(scenes.config_c1(8, 8, 1))
    oid2 = H2.add_mesh(scenes.blob_mesh(6))
    H2.set_brdf_merl(oid2, table)
    host.mh_merl_data.restype = ctypes.POINTER(ctypes.c_double)
    a = np.ctypeslib.as_array(host.mh_merl_data(H.h, oid), shape=(3 * 90 * 90 * 180,))
    b = np.ctypeslib.as_array(host.mh_merl_data(H2.h, oid2), shape=(3 * 90 * 90 * 180,))
    assert_bits(a, b, "MERL table read from the file")
    bad = tmp_path / "bad.binary"
    with open(bad, "wb") as f:
        f.write(np.array([90, 90, 90], np.int32).tobytes()); f.write(np.zeros(10, np.float64).tobytes())
    with pytest.raises(capi.MiptError):
        H.set_brdf_merl_file(oid, bad)
    with pytest.raises(capi.MiptError):
        H.set_brdf_merl_file(oid, tmp_path / "missing.binary")


def test_scene_desc_ctypes_layout_matches_header():
    """The ctypes mirror of mipt_object / mipt_mesh / mipt_bvh_node (used by tests that hand-build descriptions) has the
    layout the C compiler gives include/mipt.h."""
    import subprocess, tempfile, os, ctypes
    src = "\n".join(['#include <stdio.h>', '#include <stddef.h>', '#include "mipt.h"',
                     'int main(){printf("%zu %zu %zu %zu %zu %zu %zu", sizeof(mipt_bvh_node), sizeof(mipt_mesh), sizeof(mipt_object),',
                     '  offsetof(mipt_object, mesh), offsetof(mipt_object, O), offsetof(mipt_mesh, nodes), sizeof(mipt_scene_desc));return 0;}', ''])
    d = tempfile.mkdtemp()
    open(os.path.join(d, "t.c"), "w").write(src)
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")], check=True)
    got = [int(x) for x in subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(capi.MiptBvhNode), ctypes.sizeof(capi.MiptMesh), ctypes.sizeof(capi.MiptObject), capi.MiptObject.mesh.offset,
            capi.MiptObject.O.offset, capi.MiptMesh.nodes.offset, ctypes.sizeof(capi.MiptSceneDesc)]
    assert got == want, (got, want)
