"""the BVH built on the GPU (mipt_build_bvh) is the reference's tree — same nodes at the same positions
of the node vector, same triangle order — checked against the host mirror's recursion, which the CPU suite pins to the
oracle and, through the golden fixtures, to the reference's own TriMesh::build_bvh (TriangleMesh.cpp:878-885, 1029-1130).
Node boxes are compared as floats (==): a coordinate that is +0 in some vertices and -0 in others may carry the other
zero, everything else bit for bit."""
import numpy as np
import pytest

from pathtracer_amd import capi, scenes

pytestmark = pytest.mark.gpu


def both_trees(mesh, center=True):
    cfg = scenes.config_c1(16, 16, 1)
    out = {}
    try:
        for mode in ("host", "gpu"):
            capi.set_bvh_builder(mode)
            H = capi.HostRaytracer()
            H.apply_config(cfg)
            obj = H.add_mesh(mesh, center=center)
            assert obj >= 0, H.host.mh_last_error(H.h)
            assert H.mesh_bvh_builder(obj)[0] == mode
            out[mode] = H.mesh_dump(obj)
    finally:
        capi.set_bvh_builder("auto")
    return out["host"], out["gpu"]


def assert_same_tree(mesh, center=True):
    h, g = both_trees(mesh, center)
    assert h["nodes_i"].shape == g["nodes_i"].shape, (h["nodes_i"].shape, g["nodes_i"].shape)
    assert np.array_equal(h["perm"], g["perm"]), "triangle order differs at %d positions" % int((h["perm"] != g["perm"]).sum())
    assert np.array_equal(h["nodes_i"], g["nodes_i"]), "node topology / numbering differs"
    assert np.array_equal(h["nodes_bb"], g["nodes_bb"]), "node boxes differ"
    assert np.array_equal(h["root_bb"], g["root_bb"])
    assert np.array_equal(h["soup"].view(np.uint32), g["soup"].view(np.uint32)), "triangle soup differs"
    assert np.array_equal(h["groups"], g["groups"])
    return h


def soup_mesh(ntri, seed, spread=1.0, size=0.02):
    """Unordered triangle soup: consecutive triangles are nowhere near each other, so every partition moves data."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (ntri, 1, 3))
    v = (c + rng.normal(0, size, (ntri, 3, 3))).reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * ntri, dtype=np.int32).reshape(ntri, 3)
    nrm = np.tile(np.float32([[0, 1, 0]]), (1, 1))
    return scenes.MeshData(v, nrm, None, f, np.zeros_like(f), None, "soup%d" % ntri)


@pytest.mark.parametrize("name", ["cornell12", "blob3", "blob4", "blob5", "blob48uv", "blob200", "fine400", "fatleaf20"])
def test_gpu_bvh_is_the_reference_tree(name):
    mesh = {"cornell12": lambda: scenes.cornell_mesh(), "blob3": lambda: scenes.blob_mesh(3), "blob4": lambda: scenes.blob_mesh(4),
            "blob5": lambda: scenes.blob_mesh(5), "blob48uv": lambda: scenes.blob_mesh(48, with_uv=True),
            "blob200": lambda: scenes.blob_mesh(200), "fine400": lambda: scenes.blob_mesh(400, fine_detail=True),
            "fatleaf20": lambda: scenes.fat_leaf_mesh(20)}[name]()
    h = assert_same_tree(mesh)
    assert h["nodes_i"][0, 0] == (1 if mesh.ntri <= 4 else 0)


@pytest.mark.parametrize("ntri,seed", [(1, 0), (2, 1), (33, 2), (1000, 3), (200000, 4)])
def test_gpu_bvh_on_unordered_soups(ntri, seed):
    assert_same_tree(soup_mesh(ntri, seed), center=False)


def test_gpu_bvh_partition_with_long_jump_chains():
    """A few far-away triangles at the front of the array, all others behind: in the reference's swap loop each of them
    is moved once per following triangle (a chain of ~100000 jumps), which the device resolves by pointer doubling."""
    m = soup_mesh(100000, 5)
    v = m.vertices.copy()
    v[0:9] += np.float32([500.0, 0, 0])        # triangles 0..2 far out on the split axis
    v[9 * 50:9 * 50 + 3] += np.float32([0, 0, 300.0])
    assert_same_tree(scenes.MeshData(v, m.normals, None, m.faces_v, m.faces_n, None, "outliers"), center=False)


def test_gpu_bvh_unsplittable_segment_becomes_a_leaf():
    """More than BVHB_SMALL triangles with one common centroid: pivot ends at i1-1, the node stays a leaf (TriangleMesh.cpp:1107)."""
    rng = np.random.default_rng(6)
    base = soup_mesh(500, 7)
    tri = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    copies = np.concatenate([np.roll(tri, k % 3, axis=0) for k in range(40)])        # same three vertices, same centroid
    v = np.concatenate([base.vertices, copies + np.float32([3, 0, 0])])
    f = np.arange(v.shape[0], dtype=np.int32).reshape(-1, 3)
    f = f[rng.permutation(f.shape[0])]
    h = assert_same_tree(scenes.MeshData(v, base.normals, None, np.ascontiguousarray(f), np.zeros_like(f), None, "coincident"), center=False)
    leaves = h["nodes_i"][h["nodes_i"][:, 0] == 1]
    assert (leaves[:, 2] - leaves[:, 1]).max() >= 40


def test_gpu_bvh_entry_refuses_bad_input():
    v = np.zeros((3, 3), np.float32)
    with pytest.raises(capi.MiptError):
        capi.build_bvh(v, np.int32([[0, 1, 3]]))
    nodes_i, nodes_bb, perm, sec = capi.build_bvh(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]), np.int32([[0, 1, 2]]))
    assert nodes_i.tolist() == [[1, 0, 1]] and perm.tolist() == [0] and nodes_bb.tolist() == [[0, 0, 0, 1, 1, 0]]


def test_default_builder_on_a_gpu_box_is_the_gpu():
    H = capi.HostRaytracer()
    H.apply_config(scenes.config_c1(16, 16, 1))
    obj = H.add_mesh(scenes.blob_mesh(24))
    assert H.mesh_bvh_builder(obj)[0] == "gpu"


@pytest.mark.parametrize("name", ["cornell", "blob32", "glossy", "glass", "textured", "cutout", "merl"])
def test_gpu_bvh_against_the_reference_goldens(name):
    """The GPU-built tree of every golden scene, compared DIRECTLY with what the compiled reference's TriMesh::build_bvh
    produced (tests/golden/scene_*.npz: perm / nodes_i / nodes_bb / root_bb / soup written by tests/golden/make_golden.py
    from oracle/_ref) — not through the host recursion."""
    from helpers import load_golden, setup_scene
    g = load_golden(f"scene_{name}.npz")
    capi.set_bvh_builder("gpu", 0)
    try:
        H = capi.HostRaytracer()
        mesh, cfg, oid = setup_scene_no_device(H, name)
        assert H.mesh_bvh_builder(oid)[0] == "gpu"
        d = H.mesh_dump(oid)
    finally:
        capi.set_bvh_builder("auto", 0)
    assert np.array_equal(d["perm"], g["perm"]), "triangle order"
    assert np.array_equal(d["nodes_i"], g["nodes_i"]), "node topology / numbering"
    assert np.array_equal(d["nodes_bb"], g["nodes_bb"]), "node boxes (compared as floats: +0 == -0)"
    assert np.array_equal(d["root_bb"], g["root_bb"])
    assert np.array_equal(d["soup"][:, :16].view(np.uint32), g["soup16"].view(np.uint32)), "triangle records in tree order"
    assert np.array_equal(d["groups"], g["groups"])


@pytest.mark.parametrize("name", ["cornell", "blob32", "glossy", "glass", "textured", "cutout", "merl"])
def test_device_resident_mesh_against_the_reference_goldens(name):
    """Round 4: TriMesh::init leaves the tree, the Triangle records and the tangents on the device (mipt_device_mesh_build).  The
    reference-layout views the host mirror fetches on demand are the compiled reference's, like the round-3 path's above."""
    from helpers import load_golden
    g = load_golden(f"scene_{name}.npz")
    capi.set_bvh_builder("gpu", 0)
    try:
        H = capi.HostRaytracer()
        mesh, cfg, oid = setup_scene_no_device(H, name)
        assert H.mesh_bvh_builder(oid)[0] == "gpu" and H.mesh_on_device(oid)
        d = H.mesh_dump(oid)              # sync_host(): downloads bvh.nodes + the permutation, derives the reordered indices and triangleSoup
    finally:
        capi.set_bvh_builder("auto", 0)
    assert np.array_equal(d["perm"], g["perm"]) and np.array_equal(d["nodes_i"], g["nodes_i"]) and np.array_equal(d["nodes_bb"], g["nodes_bb"])
    assert np.array_equal(d["soup"][:, :16].view(np.uint32), g["soup16"].view(np.uint32)) and np.array_equal(d["groups"], g["groups"])


@pytest.mark.parametrize("name", ["textured", "cutout", "glossy", "merl"])
def test_device_made_records_render_like_host_made_ones(name):
    """The records the device derives from its tree (fat nodes, intersection / shading records, UV index triples, tangents) against
    those convert_mesh packs on the host from the downloaded tree: per-sample radiance of the golden scene, bit for bit — it covers
    textures through the UVs, the alpha test through the index triples, normal interpolation, the tangent frame (every OBJ-style mesh
    carries a null normal map: TriangleMesh.cpp:952-970 runs whenever the list exists)."""
    from make_golden import golden_scene
    out = {}
    for resident in (True, False):
        capi.set_device_resident(resident)
        try:
            H = capi.HostRaytracer(device=0)
            mesh, cfg, mat = golden_scene(name)
            H.apply_config(cfg)
            oid = H.add_mesh(mesh)
            assert H.mesh_on_device(oid) == resident
            scenes.install_material(H, oid, mat)
            H.prepare()
            pix = np.stack(np.meshgrid(np.arange(0, cfg.H, 3), np.arange(0, cfg.W, 3), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)
            out[resident] = H.sample_radiance(pix, 0, min(cfg.spp, 4))[0]
        finally:
            capi.set_device_resident(True)
    assert np.array_equal(out[True].view(np.uint32), out[False].view(np.uint32))


@pytest.mark.parametrize("name,devices", [("textured", 0), ("cutout", 0), ("merl", 0), ("textured", [0, 0])])
def test_device_resident_mesh_replicated_with_peer_copies(name, devices):
    """What the other members of a group do with a mesh that was built on member 0's device: hipMemcpyPeer of the records, the index
    triples and the tangents into their own buffers (csrc/mipt_mesh_device.h).  A one-GPU box has no other device, so the option
    device_mesh_as_remote makes a context treat its own device's mesh as a remote one (a peer copy within one device is legal HIP);
    the scene must render exactly like the one that reads the records in place — alone and as a group of two."""
    from make_golden import golden_scene
    out = {}
    for remote in (0, 1):
        H = capi.HostRaytracer(device=devices)
        mesh, cfg, mat = golden_scene(name)
        H.apply_config(cfg)
        oid = H.add_mesh(mesh)
        assert H.mesh_on_device(oid)
        scenes.install_material(H, oid, mat)
        H.set_option("device_mesh_as_remote", remote)
        H.prepare()
        if isinstance(devices, list):
            out[remote] = np.concatenate([a.reshape(-1) for a in H.render()])
        else:
            pix = np.stack(np.meshgrid(np.arange(0, cfg.H, 3), np.arange(0, cfg.W, 3), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)
            out[remote] = H.sample_radiance(pix, 0, min(cfg.spp, 4))[0]
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


def test_device_tangents_are_the_host_loop_s():
    """setup_tangents on the device (per-vertex sums in ascending face order, a UV sphere's poles with a thousand incident faces)
    against the host loop, bit for bit, through the lazily downloaded tangentSoup."""
    H = capi.HostRaytracer()
    H.apply_config(scenes.config_c1(16, 16, 1))
    mesh = scenes.blob_mesh(96, with_uv=True)
    a = H.add_mesh(mesh)
    capi.set_device_resident(False)
    try:
        b = H.add_mesh(mesh)
    finally:
        capi.set_device_resident(True)
    assert H.mesh_on_device(a) and not H.mesh_on_device(b)
    ta, tb = H.mesh_tangents(a), H.mesh_tangents(b)
    assert ta.shape == tb.shape and ta.shape[0] == mesh.ntri * 3
    assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32))


def setup_scene_no_device(H, name):
    """tests/golden/make_golden.py's setup() without the upload at the end (the context has no device; only TriMesh::init runs)."""
    from make_golden import golden_scene
    mesh, cfg, mat = golden_scene(name)
    H.apply_config(cfg)
    oid = H.add_mesh(mesh)
    return mesh, cfg, oid
