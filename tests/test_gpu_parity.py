"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI of
include/mipt.h (pathtracer_amd/capi.py is a thin ctypes view of it); the expected values are the
golden vectors generated from the compiled reference, and the oracle on fresh random scenes.

Bars: ray-level results (hit / object / triangle / t / P / normal / occlusion) are pure IEEE
+,-,*,/,sqrt arithmetic and must be BIT-EXACT.  Per-sample radiance and the single-pass splatted
image must be BIT-EXACT on every golden scene: sinf / cosf / powf / acosf / atan2f / expf / logf and the
fp64 exp / pow / sincos / acos / atan2 (random_Phong, the MERL transform, the subsurface weight) are the host
libm's algorithms (csrc/mipt_sincos.h, mipt_powf.h, mipt_invtrig.h, mipt_explog.h, mipt_libm64.h): no
transcendental of the path comes from the device library.  The north-star tolerance, per-pixel
L_inf < 1e-4 on radiance / 196964.7, is asserted beside the bit checks.
"""
import numpy as np
import pytest

from helpers import WHITE, all_pixels, assert_bits, bits_equal, load_golden, setup_scene
from pathtracer_amd import capi, scenes

pytestmark = pytest.mark.gpu

TOL = 1e-4   # per-pixel L_inf on radiance / 196964.7 (BASELINE.json north_star)


def gpu(name, **opts):
    rt = capi.HostRaytracer(device=0)
    scene = setup_scene(rt, name)
    for k, v in opts.items():
        rt.set_option(k, v)
    return rt, scene


def normalised(img, cnt):
    return img / np.maximum(cnt, 1e-30)[..., None] / WHITE


@pytest.mark.parametrize("name", ["cornell", "blob32", "glossy", "glass", "textured", "cutout", "merl"])
def test_rays_bit_exact(name):
    g = load_golden(f"scene_{name}.npz")
    rt, _ = gpu(name)
    hi, hf = rt.intersect(g["rays"])
    assert_bits(hi[:, 0], g["hit_i"][:, 0], "has_inter")
    hit = hi[:, 0] == 1
    assert_bits(hi[hit, 1], g["hit_i"][hit, 1], "object id")
    mesh_hit = hit & (hi[:, 1] >= 3)
    assert_bits(hi[mesh_hit, 2], g["hit_i"][mesh_hit, 2], "triangle id")
    assert_bits(hf[hit, :7], g["hit_f"][hit, :7], "t / P / shadingN")
    shaded = hit & (hi[:, 1] >= 2)
    assert_bits(hf[shaded, 7:19], g["hit_f"][shaded, 7:19], "material")
    assert_bits(rt.intersect_shadow(g["rays"], g["shadow_dist"]), g["shadow_occluded"], "occlusion")


@pytest.mark.parametrize("name,exact", [("cornell", True), ("blob32", True), ("glossy", True), ("glass", True), ("textured", True), ("cutout", True), ("merl", True)])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_per_sample_radiance(name, exact, pipeline):
    g = load_golden(f"scene_{name}.npz")
    rt, (mesh, cfg, oid) = gpu(name, pipeline=pipeline)
    rgb, dxdy = rt.sample_radiance(all_pixels(cfg), 0, cfg.spp)
    assert_bits(dxdy, g["sample_dxdy"], "sensor jitter")
    same = bits_equal(rgb, g["sample_rgb"]).all(-1)
    err = np.abs(rgb - g["sample_rgb"]).max(-1) / WHITE
    print(f"{name}: bit-identical samples {same.mean():.6f}, max |err|/white {err.max():.3e}")
    if exact:
        assert same.all(), f"{(~same).sum()} of {same.size} samples differ"
    else:
        assert same.mean() > 0.9
        # a last-ulp difference in powf may, rarely, flip a discrete decision of one sample; the
        # per-pixel mean must still meet the tolerance
        pix_err = np.abs(rgb.mean(1) - g["sample_rgb"].mean(1)).max() / WHITE
        assert pix_err < TOL, pix_err


@pytest.mark.parametrize("name,exact", [("cornell", True), ("blob32", True), ("glossy", True), ("glass", True), ("textured", True), ("cutout", True), ("merl", True)])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_rendered_image(name, exact, pipeline):
    g = load_golden(f"scene_{name}.npz")
    rt, (mesh, cfg, oid) = gpu(name, pipeline=pipeline)
    img, cnt = rt.render()
    st = rt.stats()
    assert st["paths"] == cfg.W * cfg.H * cfg.spp and st["pipeline"] == pipeline
    assert_bits(cnt, g["count"], "splat weights")
    err = np.abs(normalised(img, cnt) - normalised(g["image"], g["count"])).max()
    print(f"{name}: per-pixel L_inf {err:.3e}; rays closest {st['rays_closest']} shadow {st['rays_shadow']}")
    if exact:
        assert_bits(img, g["image"], "splatted image (single pass: same summation order as the reference loop)")
    assert err < TOL


@pytest.mark.parametrize("pipeline", [0, 1])
def test_c0_full_image(pipeline):
    """BASELINE.json configs[0]: 12-triangle Cornell scene, 256x256, 64 spp, depth 4."""
    g = load_golden("c0_image.npz")
    rt, (mesh, cfg, oid) = gpu("c0full", pipeline=pipeline)
    img, cnt = rt.render()
    err = np.abs(normalised(img, cnt) - normalised(g["image"], g["count"])).max()
    print(f"C0 per-pixel L_inf {err:.3e}")
    assert err < TOL
    assert_bits(img, g["image"], "C0 image")


def test_pipelines_agree_on_ray_counts():
    """Both schedulers cast exactly the rays the reference's loop casts."""
    counts = []
    for pipeline in (0, 1):
        rt, (mesh, cfg, oid) = gpu("glass", pipeline=pipeline)
        img, cnt = rt.render()
        st = rt.stats()
        counts.append((st["paths"], st["rays_closest"], st["rays_shadow"], img.tobytes()))
    assert counts[0] == counts[1]


@pytest.mark.parametrize("pipeline", [0, 1])
def test_multi_pass_and_partition_sum(pipeline):
    """Several passes and a 3-way tile partition change only the float summation order."""
    g = load_golden("scene_blob32.npz")
    rt, (mesh, cfg, oid) = gpu("blob32", paths_per_pass=64 * 64, pipeline=pipeline)
    img, cnt = rt.render()
    assert rt.stats()["passes"] > 1
    ref = normalised(g["image"], g["count"])
    assert np.abs(normalised(img, cnt) - ref).max() < 1e-5
    acc_i, acc_c = np.zeros_like(img), np.zeros_like(cnt)
    paths = 0
    for rank in range(3):
        r2 = capi.HostRaytracer(device=0)
        r2.set_partition(16, rank, 3)
        setup_scene(r2, "blob32")
        r2.set_option("pipeline", pipeline)
        i2, c2 = r2.render()
        paths += r2.stats()["paths"]
        acc_i += i2
        acc_c += c2
    assert paths == cfg.W * cfg.H * cfg.spp
    assert np.abs(normalised(acc_i, acc_c) - ref).max() < 1e-5
    np.testing.assert_allclose(acc_c, g["count"], rtol=1e-5)


def test_reference_entry_points():
    """Raytracer::render_image_nopreviz / render_image of the host mirror."""
    g = load_golden("scene_blob32.npz")
    rt, (mesh, cfg, oid) = gpu("blob32")
    img, cnt, u8 = rt.render_image_nopreviz()     # imagedouble already divided by sample_count
    ref = g["image"] / g["count"][..., None]
    np.testing.assert_allclose(img, ref, rtol=1e-5, atol=1e-3)
    expect_u8 = np.clip(255.0 * np.power(ref.astype(np.float64) / 196964.7, 1 / np.float32(2.2)), 0, 255).astype(np.uint8)
    assert np.abs(u8.astype(int) - expect_u8.astype(int)).max() <= 1
    img2, cnt2, u82 = rt.render_image()           # progressive: one pass per sample
    assert np.abs(normalised(img2, cnt2) - normalised(g["image"], g["count"])).max() < 1e-5
    assert np.abs(u82.astype(int) - expect_u8.astype(int)).max() <= 1


@pytest.mark.parametrize("seed", [3, 4])
def test_random_scene_against_oracle(seed):
    from oracle.binding import Oracle
    rng = np.random.default_rng(seed)
    mesh = scenes.blob_mesh(40 + 8 * seed, fine_detail=bool(seed & 1))
    cfg = scenes.config_c1(96, 64, 6)
    cfg.nb_bounces = 5
    cfg.aperture = float(rng.uniform(0.0, 0.5))
    cfg.light_center = tuple(float(x) for x in rng.uniform(-20, 30, 3))
    out = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        X.apply_config(cfg)
        X.add_mesh(mesh)
        X.prepare()
        out.append(X.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0])
    assert_bits(out[0], out[1], "per-sample radiance, default materials")


def test_error_paths():
    rt = capi.HostRaytracer(device=0)
    with pytest.raises(capi.MiptError):
        rt.trace(np.zeros((1, 6), np.float32))      # no scene uploaded
    rt.apply_config(scenes.config_c0())
    rt.add_mesh(scenes.cornell_mesh())
    rt.prepare()
    with pytest.raises(capi.MiptError):
        rt.set_option("no_such_option", 1)
    with pytest.raises(capi.MiptError):
        rt.sample_radiance(np.array([[10 ** 6, 0]], np.int32), 0, 1)   # pixel outside the image


def test_two_meshes_fat_leaves_ties_and_literal_slab():
    from oracle.binding import Oracle
    cfg = scenes.config_c1(96, 64, 6)
    cfg.nb_bounces = 5
    fat, small = scenes.fat_leaf_mesh(), scenes.blob_mesh(24, fine_detail=True)
    O, G = Oracle(), capi.HostRaytracer(device=0)
    for X in (O, G):
        X.apply_config(cfg)
        a = X.add_mesh(fat, scale=30.0)
        X.add_mesh(small, scale=14.0)          # a second mesh behind the first in the object list
        X.prepare()
    d = G.mesh_dump(a)
    leaves = d["nodes_i"][d["nodes_i"][:, 0] == 1]
    assert (leaves[:, 2] - leaves[:, 1]).max() > 4
    pix = all_pixels(cfg)
    want = O.getcolor_samples(pix, 0, cfg.spp)[0]
    for opts in ({"pipeline": 1}, {"pipeline": 1, "literal_slab": 1}, {"pipeline": 1, "refill": 0}, {"pipeline": 1, "sort_rays": 1}, {"pipeline": 0}):
        for k, v in opts.items():
            G.set_option(k, v)
        assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, f"per-sample radiance {opts}")
        G.set_option("literal_slab", 0); G.set_option("refill", 1); G.set_option("sort_rays", 0)


def test_progress_callback_and_cancel():
    """mipt_render's threading contract: the caller's buffers hold the running sums at every progress
    call, and raising the cancel flag ends the render between passes with the finished passes in the buffers."""
    rt, (mesh, cfg, oid) = gpu("blob32")
    slots = ((cfg.W + 7) // 8) * ((cfg.H + 7) // 8) * 64             # path slots per sample: whole 8x8 pixel blocks
    rt.set_option("paths_per_pass", 2 * slots)                     # 2 samples per pass -> spp / 2 passes
    rc, img, cnt, calls = rt.render_progressive()
    assert rc == capi.MIPT_OK
    assert [c[0] for c in calls] == list(range(2, cfg.spp + 1, 2)) and all(c[1] == cfg.spp for c in calls)
    sums = [c[2] for c in calls]
    assert all(b > a for a, b in zip(sums, sums[1:])) and abs(sums[-1] - float(cnt.sum(dtype=np.float64))) < 1e-3 * sums[-1]
    g = load_golden("scene_blob32.npz")
    assert np.allclose(cnt, g["count"], rtol=1e-5, atol=0)         # several passes: same samples, other summation order
    err = np.abs(normalised(img, cnt) - normalised(g["image"], g["count"])).max()
    assert err < TOL                                               # several passes: same samples, other summation order
    # cancel after the first pass: status CANCELLED, buffers = exactly the first 2 samples
    rc2, img2, cnt2, calls2 = rt.render_progressive(cancel_after=1)
    assert rc2 == capi.MIPT_ERR_CANCELLED and len(calls2) == 1 and calls2[0][0] == 2
    rt.params.sample_begin, rt.params.sample_end = 0, 2
    img_ref, cnt_ref = rt.render()
    assert_bits(img2, img_ref, "image after cancel == render of the finished samples")
    assert_bits(cnt2, cnt_ref, "weights after cancel")


@pytest.mark.parametrize("pipeline", [0, 1])
def test_scene_without_mesh_and_tiny_images(pipeline):
    """Edge cases of the queues: no TriMesh at all (every ray is decided by the analytic objects and nothing reaches the
    traversal kernels), images smaller than one 8x8 pixel block, a single sample, depth 1."""
    from oracle.binding import Oracle
    for (W, H, spp, depth) in ((48, 32, 3, 4), (5, 3, 1, 1), (1, 1, 2, 3)):
        cfg = scenes.config_c1(W, H, spp)
        cfg.nb_bounces = depth
        O, G = Oracle(), capi.HostRaytracer(device=0)
        for X in (O, G):
            X.apply_config(cfg)
            X.prepare()
        G.set_option("pipeline", pipeline)
        pix = all_pixels(cfg)
        assert_bits(G.getcolor_samples(pix, 0, spp)[0], O.getcolor_samples(pix, 0, spp)[0], f"no mesh, {W}x{H}x{spp}, depth {depth}")
        img, cnt = G.render()
        oimg, ocnt = O.render_seeded()
        assert_bits(cnt, ocnt, "weights")
        assert_bits(img, oimg, "image")


def test_degenerate_deep_bvh_is_refused_not_overrun():
    """A chain-shaped BVH deeper than the traversal stack (hand-built node array through the C ABI) must be refused at
    upload; the reference's own fixed-size stack is undefined behaviour there (TriangleMesh.cpp:1153)."""
    import ctypes as C
    capi.set_device_resident(False)                           # (the scene description must carry host arrays: the test swaps the node array)
    try:
        rt = capi.HostRaytracer(device=0)
        rt.apply_config(scenes.config_c1(8, 8, 1))
        oid = rt.add_mesh(scenes.blob_mesh(8))
    finally:
        capi.set_device_resident(True)
    rt.prepare()                                              # a valid upload first
    # now hand the ABI a 60-level chain: node i = inner(i+1, leaf), built over the first triangles of the same mesh
    desc = C.cast(rt.host.mh_scene_desc(rt.h), C.POINTER(capi.MiptSceneDesc)).contents
    mesh = desc.objects[oid].mesh.contents
    depth = 60
    nodes = (capi.MiptBvhNode * (2 * depth + 1))()
    for i in range(depth):
        nodes[2 * i].isleaf = 0; nodes[2 * i].fg = 2 * i + 1; nodes[2 * i].fd = 2 * i + 2
        nodes[2 * i + 1].isleaf = 1; nodes[2 * i + 1].fg = i % 4; nodes[2 * i + 1].fd = i % 4 + 1
        for k in range(3):
            nodes[2 * i].bbox_min[k] = nodes[2 * i + 1].bbox_min[k] = -1.0
            nodes[2 * i].bbox_max[k] = nodes[2 * i + 1].bbox_max[k] = 1.0
    last = nodes[2 * depth]
    last.isleaf = 1; last.fg = 0; last.fd = 1
    for k in range(3):
        last.bbox_min[k], last.bbox_max[k] = -1.0, 1.0
    old_nodes, old_n = mesh.nodes, mesh.n_nodes
    mesh.nodes, mesh.n_nodes = C.cast(nodes, type(mesh.nodes)), 2 * depth + 1
    try:
        rc = rt.mipt.mipt_upload_scene(rt.ctx, C.byref(desc))
        assert rc == capi.MIPT_ERR_UNSUPPORTED, rc
        assert b"traversal stack" in rt.mipt.mipt_last_error(rt.ctx)
    finally:
        mesh.nodes, mesh.n_nodes = old_nodes, old_n


def test_scene_is_uploaded_again_only_when_it_changed():
    """The host mirror's render calls skip mipt_upload_scene while the scene's fingerprint (every description the ABI gets +
    the epoch of in-place rewrites of bulk data) is unchanged, and upload again after a material edit or a new texture image."""
    rt, (mesh, cfg, oid) = gpu("textured")
    a, _, _ = rt.render_image_nopreviz()
    b, _, _ = rt.render_image_nopreviz()
    assert np.array_equal(a, b)
    rt.set_group_material(oid, 0, (0.2, 0.9, 0.3), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))      # a multiplier: the descriptions change
    c, _, _ = rt.render_image_nopreviz()
    assert not np.array_equal(a, c)
    rt.set_group_texture(oid, 0, 0, scenes.checker_texture(64, 32, 99, 4))               # same size, new pixels: only the epoch changes
    d, _, _ = rt.render_image_nopreviz()
    assert not np.array_equal(c, d)
    fresh, _ = gpu("textured")                                                          # the same edits on a new context
    fresh.set_group_material(oid, 0, (0.2, 0.9, 0.3), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    fresh.set_group_texture(oid, 0, 0, scenes.checker_texture(64, 32, 99, 4))
    e, _, _ = fresh.render_image_nopreviz()
    assert_bits(d, e, "image after the edits")


def test_filter_sigma_change_on_one_context():
    """prepare_render refills filter_integral IN PLACE when sigma changes but ceil(2 sigma) does not (Raytracer.cpp:1354-1369:
    same address, same size), so the device copy of the tables is keyed on sigma and on the table's content too: a
    second render on the same context with another sigma must match a fresh oracle at that sigma, bit for bit."""
    from oracle.binding import Oracle
    mesh = scenes.blob_mesh(24)
    G = capi.HostRaytracer(device=0)
    for sigma in (0.5, 0.4, 1.0, 0.8, 0.5):
        cfg = scenes.config_c1(64, 40, 3)
        cfg.sigma_filter = sigma
        O = Oracle()
        O.apply_config(cfg)
        O.add_mesh(mesh)
        O.prepare()
        G.apply_config(cfg)
        if sigma == 0.5 and G.num_objects() < 4:
            G.add_mesh(mesh)
        G.prepare()
        img, cnt = G.render()
        oimg, ocnt = O.render_seeded()
        assert_bits(cnt, ocnt, f"splat weights at sigma {sigma}")
        assert_bits(img, oimg, f"image at sigma {sigma}")


def test_pass_is_sized_for_the_free_memory():
    """A pass never asks for more state than the device has free (hipMemGetInfo; here a pretended 24 MB through the
    test hook): the render splits into more passes instead of failing, and the image only changes in summation order."""
    g = load_golden("scene_blob32.npz")
    rt, (mesh, cfg, oid) = gpu("blob32")
    img0, cnt0 = rt.render()
    assert rt.stats()["passes"] == 1
    slots = ((cfg.W + 7) // 8) * ((cfg.H + 7) // 8) * 64
    rt.set_option("pass_memory_limit", int((2.5 * slots * 160 + 240000) / 0.8))     # room for about two samples per pixel and pass (160 B of state per path)
    img, cnt = rt.render()
    assert 1 < rt.stats()["passes"] < cfg.spp
    assert np.abs(normalised(img, cnt) - normalised(img0, cnt0)).max() < 1e-5
    rt.set_option("pass_memory_limit", 1)                                 # not even one sample fits: one sample per pass is still tried
    img, cnt = rt.render()
    assert rt.stats()["passes"] == cfg.spp
    assert np.abs(normalised(img, cnt) - normalised(g["image"], g["count"])).max() < 1e-5
    rt.set_option("pass_memory_limit", 0)


def test_partition_rank_splat_slices_are_deterministic():
    """A rank of a tile partition cuts the samples of a pass into slices along the sample index (the splat would leave most
    of the chip idle otherwise) and adds the partial images in slice order: the same frame on every run, the frame of the
    unsliced kernel up to the summation order, and the ranks still sum to the whole frame."""
    g = load_golden("scene_blob32.npz")
    ref = normalised(g["image"], g["count"])
    frames = {}
    for slices in (0, 0, 1, 5):
        acc_i, acc_c = None, None
        for rank in range(4):
            r = capi.HostRaytracer(device=0)
            r.set_partition(8, rank, 4)
            setup_scene(r, "blob32")
            r.set_option("resolve_slices", slices)
            i2, c2 = r.render()
            acc_i = i2 if acc_i is None else acc_i + i2
            acc_c = c2 if acc_c is None else acc_c + c2
        assert np.abs(normalised(acc_i, acc_c) - ref).max() < 1e-5
        frames.setdefault(slices, []).append(acc_i.tobytes())
    assert frames[0][0] == frames[0][1]


@pytest.mark.parametrize("tile,nranks,slices,rows", [(8, 4, 0, 12), (16, 8, 0, 12), (32, 3, 0, 12), (8, 4, 1, 12), (8, 5, 3, 5), (16, 2, 0, 0)])
def test_partition_rank_splat_packed_columns_change_nothing(tile, nranks, slices, rows):
    """Round 5: a wave of a rank's column-scan splat takes 64 columns that receive something from the rank's pixels (option
    resolve_packed, default) instead of 64 adjacent columns of the frame.  Every destination pixel still gets its terms in the same
    order from the same slices: the partial frame of every rank is the unpacked kernel's bit for bit — pixels the rank does not reach
    included (they stay what they were) —, with and without slices, for band heights that do and do not divide the tiles, and
    with the per-pixel gather kernel (rows = 0), which has no columns to pack."""
    for rank in range(nranks):
        out = {}
        for packed in (1, 0):
            r = capi.HostRaytracer(device=0)
            r.set_partition(tile, rank, nranks)
            setup_scene(r, "blob32")
            r.set_option("resolve_slices", slices)
            r.set_option("resolve_rows", rows)
            r.set_option("resolve_packed", packed)
            out[packed] = r.render()
        assert np.array_equal(out[1][0].view(np.uint32), out[0][0].view(np.uint32)) and np.array_equal(out[1][1].view(np.uint32), out[0][1].view(np.uint32)), rank


@pytest.mark.parametrize("seed", range(6))
def test_partition_rank_splat_packed_columns_on_random_partitions(seed):
    """The same comparison on frames whose sides are not multiples of anything (8-pixel blocks, tiles and bands stick out of the image),
    both filter radii the column scan is built for (sigma 0.5 -> 1 pixel, sigma 1.0 -> 2 pixels), random tile sizes, rank counts, band
    heights and slice counts: every rank's partial frame equals the unpacked kernel's bit for bit, and the ranks add up to the whole frame."""
    rng = np.random.default_rng(100 + seed)
    W, H = int(rng.integers(33, 150)), int(rng.integers(17, 90))
    tile, nranks = int(rng.choice([8, 16, 24, 32, 40])), int(rng.integers(2, 9))
    rows, slices = int(rng.choice([3, 7, 12, 16, 32])), int(rng.choice([0, 0, 1, 2, 5]))
    cfg = scenes.config_c1(W, H, 6)
    cfg.sigma_filter = float(rng.choice([0.5, 1.0]))
    mesh = scenes.blob_mesh(12)
    one = capi.HostRaytracer(device=0)
    one.apply_config(cfg); one.add_mesh(mesh); one.prepare()
    img1, cnt1 = one.render()
    acc_i, acc_c = np.zeros_like(img1), np.zeros_like(cnt1)
    for rank in range(nranks):
        out = {}
        for packed in (1, 0):
            r = capi.HostRaytracer(device=0)
            r.set_partition(tile, rank, nranks)
            r.apply_config(cfg); r.add_mesh(mesh); r.prepare()
            r.set_option("resolve_slices", slices); r.set_option("resolve_rows", rows); r.set_option("resolve_packed", packed)
            out[packed] = r.render()
        what = (W, H, tile, nranks, rows, slices, cfg.sigma_filter, rank)
        assert np.array_equal(out[1][0].view(np.uint32), out[0][0].view(np.uint32)) and np.array_equal(out[1][1].view(np.uint32), out[0][1].view(np.uint32)), what
        acc_i += out[1][0]; acc_c += out[1][1]
    np.testing.assert_allclose(acc_c, cnt1, rtol=1e-5)
    assert np.abs(normalised(acc_i, acc_c) - normalised(img1, cnt1)).max() < 1e-5


@pytest.mark.parametrize("pipeline", [1, 0])
def test_two_hundred_objects_forty_meshes(pipeline):
    """Scene::objects is an unbounded vector (Geometry.h:1306-1309).  Until round 6 a hit record had five bits for its object: 31 objects.
    Now: light, environment, ground plane, 157 spheres of every kind and 40 small meshes (one of them, far beyond object 32, with a measured
    BRDF; one alpha-tested) — bit for bit against the oracle at depth 3 and depth 0, rays of every mesh by Scene::intersection, and the
    image through the splat.  A hit names its mesh by the scene-wide triangle index (csrc/mipt_trace.h, hit_unpack)."""
    from oracle.binding import Oracle
    merl_objs = []

    def build(X):
        cfg = scenes.config_c1(64, 40, 2)
        cfg.nb_bounces = 3
        X.apply_config(cfg)
        rng = np.random.default_rng(11)
        meshes = []
        merl_objs.clear()
        for k in range(197):
            c = (float(rng.uniform(-32, 32)), float(rng.uniform(-26, 4)), float(rng.uniform(-20, 28)))
            if k % 5 == 4 or k == 196:                         # 40 meshes spread over the list (their triangle ranges ascend with the object index)
                m = scenes.blob_mesh(4 + k % 7, with_uv=(k == 99))
                sc_ = float(rng.uniform(3, 7))
                # placed through its vertices (TriMesh::init with center = false keeps them; it swaps the axes (x, y, z) -> (-z, y, x),
                # TriangleMesh.cpp:742-751: world (x, z) = (-z_in, x_in)); add_mesh then rests the mesh on the ground plane
                v = m.vertices.astype(np.float64) * sc_ + np.array([c[2], 0.0, -c[0]])
                m = scenes.MeshData(v.astype(np.float32), m.normals, m.uvs, m.faces_v, m.faces_n, m.faces_t, "blob_%d" % k)
                o = X.add_mesh(m, scale=1.0, center=False)
                meshes.append(o)
                if k == 99:
                    X.set_group_material(o, 0, (0.8, 0.8, 0.8), (0, 0, 0), (0, 0, 0))
                    X.set_group_texture(o, 0, 3, scenes.alpha_texture())       # alpha-tested leaves (the per-lane leaf loop)
                if k in (24, 49, 149, 196):
                    X.set_brdf_merl(o, scenes.synthetic_merl_table())          # object indices below and above 32 (object_has_merl; the request word of the batched tier)
                    merl_objs.append(o)
                continue
            o = X.add_sphere(c, float(rng.uniform(1.5, 4.5)), mirror=(k % 4 == 1))
            if k % 4 == 0: X.add_group_material(o, tuple(rng.uniform(0.1, 1, 3)), (0, 0, 0), (0, 0, 0), 1.0, 1.3)
            if k % 4 == 2: X.add_group_material(o, tuple(rng.uniform(0.1, 1, 3)), (0.3, 0.3, 0.3), (40, 40, 40), 1.0, 1.3)
            if k % 4 == 3: X.add_group_material(o, (1, 1, 1), (0, 0, 0), (0, 0, 0), 0.0, 1.4)
        return cfg, meshes

    O, G = Oracle(), capi.HostRaytracer(device=0)
    for X in (O, G):
        cfg, meshes = build(X)
        X.prepare()
    assert G.num_objects() == 200 and len(meshes) == 40
    G.set_option("pipeline", pipeline)
    pix = all_pixels(cfg)
    want = O.getcolor_samples(pix, 0, cfg.spp)[0]
    assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, "200 objects, depth 3")
    assert (want != 0).any(-1).mean() > 0.3
    # Scene::intersection: camera-like rays, object and (mesh-local) triangle ids of every kind of object
    rng = np.random.default_rng(5)
    rays = np.concatenate([np.tile(np.array(cfg.cam_pos, np.float32), (3000, 1)), rng.normal(size=(3000, 3)).astype(np.float32) * (0.35, 0.25, 0.1) + np.array(cfg.cam_dir, np.float32)], 1).astype(np.float32)
    gi, gf = G.intersect(rays); oi, of = O.intersect(rays)
    assert_bits(gi, oi, "hit / object / triangle ids")               # (triangle_id: -1 as soon as a sphere BEHIND the winning mesh in the list is hit, Geometry.cpp:589-650)
    on_mesh = (oi[:, 0] == 1) & np.isin(oi[:, 1], meshes)
    assert len(set(oi[on_mesh, 1].tolist()) & set(o for o in merl_objs if o > 32)) >= 1, "no ray reaches a measured-BRDF mesh beyond object 32"
    assert len(set(oi[on_mesh, 1].tolist())) >= 10 and (oi[on_mesh, 2] >= 0).sum() > 100 and (oi[on_mesh, 2] < 0).sum() > 50, "the rays reach too few meshes for the test to mean anything"
    assert_bits(gf[oi[:, 0] == 1, :7], of[oi[:, 0] == 1, :7], "t / P / normal")
    img, cnt = G.render(); oimg, ocnt = O.render_seeded()
    assert_bits(cnt, ocnt, "splat weights"); assert_bits(img, oimg, "image")
    for X in (O, G):
        X.set_render(cfg.W, cfg.H, cfg.spp, 0)
        X.prepare()
    black = G.getcolor_samples(pix, 0, cfg.spp)[0]
    assert_bits(black, O.getcolor_samples(pix, 0, cfg.spp)[0], "depth 0")
    assert not black.any()


@pytest.mark.parametrize("resident", [True, False])
def test_leaves_of_a_hundred_triangles(resident):
    """build_bvh_recur's leaves are unbounded (a bundle of triangles with one centroid cannot be split, TriangleMesh.cpp:1118); a leaf
    reference here has five bits for the count.  Leaves of 32 and more file their count in the scene's table (csrc/mipt_scene.h,
    mipt_leaf_count).  Until round 6 such a mesh was refused.  Closest hits, any hits in all three forms of the stage, both pipelines,
    the contribution-queue kernel's per-thread traversal, a second mesh behind it (the table is scene-wide) — bit for bit against the
    oracle.  `resident`: the device build declines such a mesh (its records are made by the host-side conversion either way)."""
    from oracle.binding import Oracle
    capi.set_device_resident(resident)
    try:
        cfg = scenes.config_c1(72, 48, 3)
        cfg.nb_bounces = 4
        big, small = scenes.huge_leaf_mesh(), scenes.huge_leaf_mesh(8, 40, 2)
        O, G = Oracle(), capi.HostRaytracer(device=0)
        for X in (O, G):
            X.apply_config(cfg)
            a = X.add_mesh(small, scale=12.0)
            b = X.add_mesh(big, scale=30.0)
            X.prepare()
        for oid, want_max in ((a, 12), (b, 90)):
            d = G.mesh_dump(oid)
            leaves = d["nodes_i"][d["nodes_i"][:, 0] == 1]
            assert (leaves[:, 2] - leaves[:, 1]).max() >= want_max
            assert_bits(d["nodes_i"], O.mesh_dump(oid)["nodes_i"], "the reference's tree")
        pix = all_pixels(cfg)
        want = O.getcolor_samples(pix, 0, cfg.spp)[0]
        for opts in ({"pipeline": 1}, {"pipeline": 1, "anyhit_wide": 0}, {"pipeline": 1, "anyhit_flag_all": 1}, {"pipeline": 1, "refill": 0}, {"pipeline": 0}):
            for k, v in opts.items():
                G.set_option(k, v)
            assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, f"per-sample radiance {opts}")
            G.set_option("anyhit_wide", 1); G.set_option("anyhit_flag_all", 0); G.set_option("refill", 1)
        G.set_option("pipeline", 1)
        rng = np.random.default_rng(3)
        rays = np.concatenate([np.tile(np.array(cfg.cam_pos, np.float32), (4000, 1)), rng.normal(size=(4000, 3)).astype(np.float32) * (0.3, 0.3, 0.1) + np.array(cfg.cam_dir, np.float32)], 1).astype(np.float32)
        gi, gf = G.intersect(rays); oi, of = O.intersect(rays)
        assert_bits(gi, oi, "hit / object / triangle ids")
        dist = rng.uniform(5, 80, 4000).astype(np.float32)
        assert_bits(G.intersect_shadow(rays, dist), O.intersect_shadow(rays, dist), "occlusion")
        # fog: the contribution-queue pipeline (its logic stages and, with queue_wavefront = 0, the one-thread-per-sample loop)
        for X in (O, G):
            X.set_fog(0.01, 0.002)
            X.prepare()
        want = O.getcolor_samples(pix, 0, cfg.spp)[0]
        for qw in (1, 0):
            G.set_option("queue_wavefront", qw)
            assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, f"fog, queue_wavefront {qw}")
    finally:
        capi.set_device_resident(True)


@pytest.mark.parametrize("batch", [0, 1, 2])
def test_measured_brdf_tiers_against_golden_and_oracle(batch):
    """Scenes with a measured BRDF, general shade tier in both forms: every vertex evaluating its own table entries (`merl_batch` 0)
    and the evaluations filed in LDS and run 64 to a trip (1, the default since round 4).  The golden scene of the compiled reference,
    then a scene where the measured surface sits beside a glossy Phong mesh, a mirror sphere and a glass sphere (the tier takes the
    other vertices through the general code), at a depth where requests of several chunks share a trip — against the oracle, per
    sample and through the splat; ray counts equal between the two forms."""
    from oracle.binding import Oracle
    g = load_golden("scene_merl.npz")
    rt, (mesh, cfg, oid) = gpu("merl", merl_batch=batch)
    rgb, _ = rt.sample_radiance(all_pixels(cfg), 0, cfg.spp)
    assert_bits(rgb, g["sample_rgb"], f"golden MERL scene, merl_batch {batch}")
    outs = []
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        cfg = scenes.config_c1(72, 48, 5)
        cfg.nb_bounces = 6
        X.apply_config(cfg)
        a = X.add_mesh(scenes.blob_mesh(28), scale=16.0)
        X.set_brdf_merl(a, scenes.synthetic_merl_table())
        b = X.add_mesh(scenes.blob_mesh(12), scale=7.0)
        X.set_group_material(b, 0, (0.7, 0.3, 0.2), (0.4, 0.4, 0.4), (30, 30, 30))
        X.add_sphere((16, -14, 8), 5.0, mirror=True)
        s = X.add_sphere((-15, -16, 10), 4.0)
        X.add_group_material(s, (1, 1, 1), (0, 0, 0), (0, 0, 0), 0.0, 1.4)
        if isinstance(X, capi.HostRaytracer):
            X.set_option("merl_batch", batch)
        X.prepare()
        outs.append((X.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0], X.render_seeded()))
        if isinstance(X, capi.HostRaytracer):
            st = X.stats()
            assert st["pipeline"] == 1
            outs.append((st["rays_closest"], st["rays_shadow"]))
    assert_bits(outs[1][0], outs[0][0], f"per-sample radiance, merl_batch {batch}")
    assert_bits(outs[1][1][1], outs[0][1][1], "splat weights")
    assert_bits(outs[1][1][0], outs[0][1][0], "splatted image")
    assert (outs[0][0] != 0).any(-1).mean() > 0.5
    key = "_merl_tier_ray_counts"
    prev = globals().setdefault(key, outs[2])
    assert prev == outs[2], "the two forms of the tier count different rays"
