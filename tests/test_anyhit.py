"""The any-hit (shadow) stage as an order-free traversal of four-wide 8-bit nodes (csrc/mipt_anyhit.h) against the reference's ordered
TriMesh::intersection_shadow (TriangleMesh.cpp:1239-1319, Geometry.cpp:691-744), through the C-ABI, bit for bit:

* the three forms of the stage — order-free + replay list (default), every occluded ray replayed in order (`anyhit_flag_all`),
  the ordered kernel for every ray (`anyhit_wide` = 0) — give the same per-sample radiance as the oracle / the reference's goldens;
* scenes that exercise what the order-free kernel must not decide itself: light samples that end inside the mesh (leaves whose box lies
  within 0.2 % of the ray's far end: the replay list is NOT empty and the answers still match), alpha-tested leaves, leaves with more
  than four triangles, two meshes, axis-parallel rays (infinite inverse direction: the literal slab chain of the ordered kernel).
"""
import numpy as np
import pytest

from helpers import all_pixels, assert_bits, load_golden, setup_scene
from pathtracer_amd import capi, scenes

pytestmark = pytest.mark.gpu

FORMS = ({"anyhit_wide": 1, "anyhit_flag_all": 0}, {"anyhit_wide": 1, "anyhit_flag_all": 1}, {"anyhit_wide": 0, "anyhit_flag_all": 0})


@pytest.mark.parametrize("name", ["blob32", "glossy", "textured", "cutout", "glass"])
def test_forms_agree_with_the_reference_goldens(name):
    g = load_golden(f"scene_{name}.npz")
    rt = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup_scene(rt, name)
    rt.set_option("pipeline", 1)
    replayed = []
    for form in FORMS:
        for k, v in form.items():
            rt.set_option(k, v)
        rgb, _ = rt.sample_radiance(all_pixels(cfg), 0, cfg.spp)
        assert_bits(rgb, g["sample_rgb"], f"{name}: per-sample radiance, {form}")
        replayed.append(rt.anyhit_replayed())
    st = rt.stats()
    print(f"{name}: shadow rays {st['rays_shadow']}, replayed {replayed}")
    assert replayed[2] == 0                      # the ordered kernel for every ray: nothing to hand over
    assert replayed[1] >= replayed[0]            # flag_all: every ray that found an occluder
    assert replayed[1] > 0


def shell_around(center, radius, n):
    """A UV sphere of 2 n^2 triangles around `center` (world coordinates), as input to TriMesh::init with center = False, scale = 1:
    init swaps the axes (x, y, z) -> (-z, y, x) (TriangleMesh.cpp:742-751), so the input is the inverse image."""
    th = np.linspace(0.02, np.pi - 0.02, n + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, n + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    world = np.asarray(center, np.float64) + radius * d
    unswap = lambda w: np.stack([w[:, 2], w[:, 1], -w[:, 0]], -1)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (ii * (n + 1) + jj).ravel()
    f = np.empty((2 * n * n, 3), np.int32)
    f[0::2] = np.stack([a, a + 1, a + n + 2], -1)
    f[1::2] = np.stack([a, a + n + 2, a + n + 1], -1)
    return scenes.MeshData(unswap(world).astype(np.float32), unswap(d).astype(np.float32), None, f, f.copy(), None, "shell%d" % n)


def test_occluders_at_the_far_end_of_the_ray_fill_the_replay_list():
    """A finely tessellated shell 0.15 % of the typical light distance outside the light sphere: every light sample is hidden by a triangle
    with t / dist in about [0.998, 0.9995] — on either side of TriMesh::intersection_shadow's 0.999 (TriangleMesh.cpp:1309), in leaves
    whose boxes begin within 0.2 % of the ray's far end.  This is the one regime where the reference's answer depends on its visiting
    order (hits with t >= 0.999 dist lower its running t and prune boxes behind it): the order-free kernel must hand such rays over."""
    from oracle.binding import Oracle
    cfg = scenes.config_c1(96, 64, 8)
    shell = shell_around((cfg.light_center[0], 0.0, cfg.light_center[2]), cfg.light_radius + 0.04, 160)
    # add_mesh stands a mesh on the ground plane (y = -27.3: the GUI's placement, mainApp.cpp:2402-2410): the light goes where the shell's centre lands
    cfg.light_center = (cfg.light_center[0], float(-27.3 - shell.vertices[:, 1].min()), cfg.light_center[2])
    blob = scenes.blob_mesh(32)
    O, G = Oracle(), capi.HostRaytracer(device=0)
    for X in (O, G):
        X.apply_config(cfg)
        X.add_mesh(blob)
        X.add_mesh(shell, scale=1.0, center=False)
        X.prepare()
    pix = all_pixels(cfg)
    want = O.getcolor_samples(pix, 0, cfg.spp)[0]
    G.set_option("pipeline", 1)
    counts = []
    for form in FORMS:
        for k, v in form.items():
            G.set_option(k, v)
        assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, f"per-sample radiance {form}")
        counts.append(G.anyhit_replayed())
    lit = (want > 0).any(-1).mean()
    print("replayed:", counts, "of", G.stats()["rays_shadow"], "shadow rays; samples with light:", lit)
    assert counts[0] > 100, "the scene was built to put occluders at the far end of shadow rays"
    assert counts[0] <= counts[1]
    assert 0.0 < lit < 1.0


def test_two_meshes_fat_leaves_and_ties():
    from oracle.binding import Oracle
    cfg = scenes.config_c1(96, 64, 6)
    cfg.nb_bounces = 5
    fat, small = scenes.fat_leaf_mesh(), scenes.blob_mesh(24, fine_detail=True)
    O, G = Oracle(), capi.HostRaytracer(device=0)
    for X in (O, G):
        X.apply_config(cfg)
        X.add_mesh(fat, scale=30.0)
        X.add_mesh(small, scale=14.0)
        X.prepare()
    pix = all_pixels(cfg)
    want = O.getcolor_samples(pix, 0, cfg.spp)[0]
    G.set_option("pipeline", 1)
    for form in FORMS + ({"anyhit_wide": 1, "anyhit_flag_all": 0, "literal_slab": 1},):
        for k, v in form.items():
            G.set_option(k, v)
        assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, f"per-sample radiance {form}")
    G.set_option("literal_slab", 0)


def test_tiny_meshes_root_leaf_and_three_slot_nodes():
    """Meshes of 1 .. 9 triangles: a root that is a leaf (no quad node at all), nodes with two or three used slots."""
    from oracle.binding import Oracle
    base = scenes.cornell_mesh()
    for ntri in (1, 2, 3, 5, 9):
        mesh = scenes.MeshData(base.vertices, base.normals, None, np.ascontiguousarray(base.faces_v[:ntri]), np.ascontiguousarray(base.faces_n[:ntri]), None, "tiny%d" % ntri)
        cfg = scenes.config_c0()
        cfg.W, cfg.H, cfg.spp = 48, 48, 4
        O, G = Oracle(), capi.HostRaytracer(device=0)
        for X in (O, G):
            X.apply_config(cfg)
            X.add_mesh(mesh)
            X.prepare()
        pix = all_pixels(cfg)
        want = O.getcolor_samples(pix, 0, cfg.spp)[0]
        G.set_option("pipeline", 1)
        for form in FORMS:
            for k, v in form.items():
                G.set_option(k, v)
            assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, f"{ntri} triangles, {form}")


def test_every_pixel_of_a_mid_size_frame():
    """configs[1]'s scene at 480 x 270 x 4: every sample against the oracle, default form; the replay list stays (almost) empty."""
    from oracle.binding import Oracle
    mesh, cfg, mat, text = scenes.workload("c1", 480, 270, 4, None)
    O, G = Oracle(), capi.HostRaytracer(device=0)
    for X in (O, G):
        X.apply_config(cfg)
        scenes.install(X, mesh, mat)
        X.prepare()
    pix = all_pixels(cfg)
    want = O.getcolor_samples(pix, 0, cfg.spp)[0]
    G.set_option("pipeline", 1)
    assert_bits(G.getcolor_samples(pix, 0, cfg.spp)[0], want, "per-sample radiance")
    st = G.stats()
    rep = G.anyhit_replayed()
    print(f"replayed {rep} of {st['rays_shadow']} shadow rays")
    assert rep <= st["rays_shadow"] // 1000


@pytest.mark.parametrize("what", ["infinite inner box", "infinite leaf box", "child outside its parent"])
def test_tree_handed_in_through_the_abi_that_the_byte_boxes_cannot_cover_keeps_the_ordered_kernel(what):
    """mipt_mesh::nodes is the caller's.  A box with an infinite plane (or one that sticks out of its parent's) has no 8-bit cover;
    the upload check must send such a scene to the ordered any-hit kernel — results as the reference's traversal of THAT tree gives them,
    which for a box that only grew are those of the unmodified tree."""
    import ctypes as C
    g = load_golden("scene_blob32.npz")
    capi.set_device_resident(False)            # the mirror then describes the mesh by host arrays: mipt_mesh::nodes
    try:
        rt = capi.HostRaytracer(device=0)
        mesh, cfg, oid = setup_scene(rt, "blob32")
        assert rt.anyhit_kind() == "order-free"
        desc = C.cast(rt.scene_desc, C.POINTER(capi.MiptSceneDesc)).contents
        m = desc.objects[oid].mesh.contents
        assert m.nodes and not m.device_mesh
        nodes = np.ctypeslib.as_array(C.cast(m.nodes, C.POINTER(C.c_uint32)), shape=(m.n_nodes, 9)).copy()
        f = nodes.view(np.float32)
        isleaf = (nodes[:, 0] & 255) == 1
        inner = np.flatnonzero(~isleaf)
        if what == "infinite inner box":
            f[inner[len(inner) // 2], 3] = -np.inf                    # bbox_min.x of an inner node in the middle of the tree
        elif what == "infinite leaf box":
            f[np.flatnonzero(isleaf)[7], 7] = np.inf                  # bbox_max.y of a leaf
        else:
            k = inner[3]; child = int(nodes[k, 1])
            f[child, 6] = f[k, 6] + 1.0                               # the left child's bbox_max.x beyond its parent's
        m.nodes = C.cast(nodes.ctypes.data, C.POINTER(capi.MiptBvhNode))
        rc = rt.mipt.mipt_upload_scene(rt.ctx, C.byref(desc))
        assert rc == capi.MIPT_OK, rt.mipt.mipt_last_error(rt.ctx)
        assert rt.anyhit_kind().startswith("ordered: a box of the uploaded tree"), rt.anyhit_kind()
        assert_bits(rt.intersect_shadow(g["rays"], g["shadow_dist"]), g["shadow_occluded"], "occlusion")
        rt.set_option("pipeline", 1)
        rgb, _ = rt.sample_radiance(all_pixels(cfg), 0, cfg.spp)
        assert_bits(rgb, g["sample_rgb"], "per-sample radiance on the ordered any-hit kernel")
        assert rt.anyhit_replayed() == 0
    finally:
        capi.set_device_resident(True)
