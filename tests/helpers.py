"""Shared comparison helpers for the parity tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

WHITE = 196964.7   # the reference's white level (Raytracer.cpp:1543)


def bits_equal(a, b):
    """Element-wise bit equality (NaN == NaN, +0 != -0) for float32/float64/int arrays."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    if a.dtype == np.float64:
        return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
    return a == b


def assert_bits(a, b, what):
    eq = bits_equal(np.asarray(a), np.asarray(b))
    assert eq.all(), f"{what}: {int((~eq).sum())}/{eq.size} elements differ bitwise"


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def setup_scene(X, name):
    from make_golden import setup
    return setup(X, name)


def all_pixels(cfg):
    return np.stack(np.meshgrid(np.arange(cfg.H), np.arange(cfg.W), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)


def check_scene_against(X, g, name, what):
    """Runs every per-scene golden check of tests/golden/scene_<name>.npz against implementation X
    (an oracle.binding object already set up on that scene)."""
    mesh, cfg, oid = X._scene
    assert_bits(X.light(), g["light"], f"{what}:{name}:light")
    rpp, s2d, fi, fs = X.tables()
    assert_bits(rpp, g["randomPerPixel"], f"{what}:{name}:randomPerPixel")
    assert_bits(s2d, g["samples2d"], f"{what}:{name}:samples2d")
    assert_bits(fi, g["filter_integral"], f"{what}:{name}:filter_integral")
    assert fs == int(g["filter_size"])
    for k in range(oid + 1):
        for arr, key in zip(X.object_matrices(k), ("trans", "inv", "rot")):
            assert_bits(arr, g[f"obj{k}_{key}"], f"{what}:{name}:obj{k}.{key}")
    d = X.mesh_dump(oid)
    for key in ("perm", "nodes_i", "nodes_bb", "groups", "root_bb"):
        assert_bits(d[key], g[key], f"{what}:{name}:mesh.{key}")
    assert_bits(d["soup"][:, :16], g["soup16"], f"{what}:{name}:soup16")
    assert_bits(d["soup"][:, 22:31], g["soup_normals"], f"{what}:{name}:soup normals")
    assert_bits(X.camera_rays(g["cam_ij"], g["cam_jit"]), g["cam_rays"], f"{what}:{name}:camera rays")
    hi, hf = X.intersect(g["rays"])
    assert_bits(hi, g["hit_i"], f"{what}:{name}:hit ids")
    gf = g["hit_f"]
    # t, P, shadingN are defined for every hit; the material columns only for objects that run
    # queryMaterial (plane, meshes): the reference leaves them at whatever `localmat` held.
    hit = hi[:, 0] == 1
    assert_bits(hf[hit, :7], gf[hit, :7], f"{what}:{name}:hit t/P/N")
    shaded = hit & (hi[:, 1] >= 2)
    assert_bits(hf[shaded, 7:19], gf[shaded, 7:19], f"{what}:{name}:hit material")
    assert_bits(X.intersect_shadow(g["rays"], g["shadow_dist"]), g["shadow_occluded"], f"{what}:{name}:shadow")
    rgb, dxdy = X.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
    assert_bits(dxdy, g["sample_dxdy"], f"{what}:{name}:sample jitter")
    assert_bits(rgb, g["sample_rgb"], f"{what}:{name}:per-sample radiance")
    img, cnt = X.render_seeded()
    assert_bits(cnt, g["count"], f"{what}:{name}:splat weights")
    assert_bits(img, g["image"], f"{what}:{name}:splatted image")


def spawn_ranks(fn, nprocs, *args):
    """torch.multiprocessing.spawn(fn, args=(nprocs, port, *args)) on a free local port.  The port is found by binding to 0 and closed
    before the children open it (another process can take it in between, and gloo opens further ports of its own): a rendezvous that
    fails with a SOCKET error is attempted once more on a fresh port; any other failure of a rank is the test's."""
    import socket
    import torch.multiprocessing as mp
    for attempt in (0, 1):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        try:
            mp.spawn(fn, args=(nprocs, port, *args), nprocs=nprocs, join=True)
            return
        except Exception as e:                                      # ProcessRaisedException carries the rank's traceback as text
            text = str(e)
            # only a port that was taken between the probe and the rendezvous earns a second attempt ("socket" / "timed out" also match a
            # collective that hangs after the rendezvous, which must fail the test)
            port_taken = any(k in text for k in ("Address already in use", "EADDRINUSE"))
            if attempt == 1 or not port_taken:
                raise
            import warnings
            warnings.warn("rendezvous port %d was taken by another process, retrying once on a fresh port" % port)
            print("rendezvous failed on port %d, trying another one:\n%s" % (port, text[-600:]))
