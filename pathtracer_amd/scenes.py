"""Synthetic scenes for the hot path (SURVEY.md §8d "Synthetic inputs").

The reference ships no scenes, so every mesh here is procedural:

* ``cornell_mesh()``  – 12-triangle inward-normal open box + one floating panel (config C0).
* ``blob_mesh(n)``    – displaced UV-sphere with analytic normals (n=258 -> 133 128 triangles,
  config C1; n=1120 -> 2 508 800 triangles, configs C2/C3; n=3444 -> 23.7 M, config C4).

A mesh is a ``MeshData`` of float32 / int32 arrays laid out exactly like what the reference's
``TriMesh::readOBJ`` (TriangleMesh.cpp:240-458) produces from the equivalent OBJ text, and
``write_obj`` writes that text with ``%.9g`` so that ``sscanf("%f")`` recovers the same float32
bit patterns: the reference (through oracle/_ref), the oracle and the HIP path all start from
identical inputs.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np


@dataclass
class MeshData:
    vertices: np.ndarray            # (nv,3) float32, OBJ order (before the reference's axis swap)
    normals: np.ndarray             # (nn,3) float32
    uvs: Optional[np.ndarray]       # (nt,2) float32 or None
    faces_v: np.ndarray             # (nf,3) int32, 0-based
    faces_n: np.ndarray             # (nf,3) int32
    faces_t: Optional[np.ndarray]   # (nf,3) int32 or None
    name: str = "mesh"

    @property
    def ntri(self) -> int:
        return int(self.faces_v.shape[0])


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))


def cornell_mesh() -> MeshData:
    """Open box (floor, ceiling, back, left, right: 10 triangles, normals pointing inward)
    plus one floating quad (2 triangles, normal up) — 12 triangles.  Shading in the reference
    is one-sided (Raytracer.cpp:510-511, 593), hence the inward normals."""
    # box corners, unit cube centred at origin; opening towards +x in OBJ space
    # (the reference maps OBJ (x,y,z) -> (-z,y,x), TriangleMesh.cpp:742-751, so OBJ +x = world +z,
    # which faces the default camera at z=+50 looking down -z).
    v = [
        (-0.5, -0.5, -0.5), (0.5, -0.5, -0.5), (0.5, -0.5, 0.5), (-0.5, -0.5, 0.5),   # floor 0-3
        (-0.5, 0.5, -0.5), (0.5, 0.5, -0.5), (0.5, 0.5, 0.5), (-0.5, 0.5, 0.5),       # ceiling 4-7
        (-0.15, -0.2, -0.25), (0.2, -0.2, -0.2), (0.25, -0.15, 0.2), (-0.2, -0.15, 0.25),  # panel 8-11
    ]
    n = [(0, 1, 0), (0, -1, 0), (1, 0, 0), (0, 0, 1), (0, 0, -1), (0.05, 1, 0.02)]
    nn = np.asarray(n, dtype=np.float64)
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    quads = [
        ((0, 3, 2, 1), 0),   # floor, normal +y
        ((4, 5, 6, 7), 1),   # ceiling, normal -y
        ((0, 4, 7, 3), 2),   # back wall x=-0.5, normal +x
        ((0, 1, 5, 4), 3),   # wall z=-0.5, normal +z
        ((3, 7, 6, 2), 4),   # wall z=+0.5, normal -z
        ((8, 11, 10, 9), 5),  # floating panel, normal ~+y
    ]
    fv, fn = [], []
    for (a, b, c, d), ni in quads:
        fv += [(a, b, c), (a, c, d)]
        fn += [(ni, ni, ni), (ni, ni, ni)]
    return MeshData(_f32(v), _f32(nn), None, np.asarray(fv, np.int32), np.asarray(fn, np.int32), None, "cornell12")


def blob_mesh(n: int, fine_detail: bool = False, with_uv: bool = False) -> MeshData:
    """Displaced UV sphere r(θ,φ)=1+0.15 sin5θ cos7φ+0.05 sin(23θ+3φ)[+0.01 sin131θ sin97φ],
    θ∈[0.02,π−0.02], n×n quads -> 2n² triangles, analytic vertex normals."""
    th = np.linspace(0.02, np.pi - 0.02, n + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, n + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = 1 + 0.15 * np.sin(5 * T) * np.cos(7 * P) + 0.05 * np.sin(23 * T + 3 * P)
    rt = 0.75 * np.cos(5 * T) * np.cos(7 * P) + 1.15 * np.cos(23 * T + 3 * P)
    rp = -1.05 * np.sin(5 * T) * np.sin(7 * P) + 0.15 * np.cos(23 * T + 3 * P)
    if fine_detail:
        r = r + 0.01 * np.sin(131 * T) * np.sin(97 * P)
        rt = rt + 1.31 * np.cos(131 * T) * np.sin(97 * P)
        rp = rp + 0.97 * np.sin(131 * T) * np.cos(97 * P)
    st, ct, sp, cp = np.sin(T), np.cos(T), np.sin(P), np.cos(P)
    d = np.stack([st * cp, ct, st * sp], -1)
    dT = np.stack([ct * cp, -st, ct * sp], -1)
    dP = np.stack([-st * sp, np.zeros_like(T), st * cp], -1)
    pos = r[..., None] * d
    pT = rt[..., None] * d + r[..., None] * dT
    pP = rp[..., None] * d + r[..., None] * dP
    nrm = np.cross(pP, pT)   # outward for this parametrisation
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-30)
    verts = _f32(pos.reshape(-1, 3))
    norms = _f32(nrm.reshape(-1, 3))
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (ii * (n + 1) + jj).ravel()
    b = a + 1
    c = a + (n + 1)
    e = c + 1
    # two triangles per quad, counter-clockwise seen from outside
    f = np.empty((2 * n * n, 3), np.int32)
    f[0::2] = np.stack([a, b, e], -1)
    f[1::2] = np.stack([a, e, c], -1)
    uvs = None
    ft = None
    if with_uv:
        uvs = _f32(np.stack([P / (2 * np.pi), T / np.pi], -1).reshape(-1, 2))
        ft = f.copy()
    return MeshData(verts, norms, uvs, f, f.copy(), ft, f"blob{n}" + ("f" if fine_detail else ""))


def fat_leaf_mesh(n: int = 20) -> MeshData:
    """Blob with every 7th face repeated six times with OTHER vertex normals: coincident centroids cannot be
    split, so the BVH gets leaves with more than 4 triangles, and the copies tie exactly in t — the reference
    keeps the first one in leaf order (strict '<'), which is visible in the shading normal."""
    m = blob_mesh(n)
    sel = np.arange(0, m.ntri, 7)
    fv = np.concatenate([m.faces_v] + [m.faces_v[sel]] * 6)
    fn = np.concatenate([m.faces_n] + [np.roll(m.faces_n[sel], k + 1, axis=0) for k in range(6)])
    return MeshData(m.vertices, m.normals, None, np.ascontiguousarray(fv, np.int32), np.ascontiguousarray(fn, np.int32), None, "fatleaf%d" % n)


def huge_leaf_mesh(n: int = 12, fan: int = 100, fans: int = 3) -> MeshData:
    """Blob plus `fans` bundles of `fan` DIFFERENT triangles whose centroids coincide exactly (integer offsets around a common centre:
    (A + B + C) / 3 is the same float for all of them): build_bvh_recur cannot split such a bundle (TriangleMesh.cpp:1118), so the tree
    gets leaves of `fan` triangles — beyond the 32 a leaf reference's count field holds.  Each bundle is a star of thin blades, so rays
    hit different members of it (and the reference keeps the first in leaf order on a tie)."""
    m = blob_mesh(n)
    rng = np.random.default_rng(5)
    verts, faces = [m.vertices], [m.faces_v]
    base = m.vertices.shape[0]
    for b in range(fans):
        c = np.array([[-1.0, 0.5, 0.75], [0.5, 1.0, -0.5], [0.25, -0.75, 1.0]][b % 3]) * 2.0 + b // 3      # exact in float
        for k in range(fan):
            a = rng.integers(-8, 9, 3).astype(np.float64) / 16.0                                          # multiples of 1/16: every sum below is exact
            d = rng.integers(-8, 9, 3).astype(np.float64) / 16.0
            if not np.any(np.cross(a, d)):
                d = d + np.array([0.0625, 0.125, 0.0])
            tri = np.stack([c + a, c + d, c - a - d])
            verts.append(tri.astype(np.float32)); faces.append(np.array([[base, base + 1, base + 2]], np.int32)); base += 3
    v = np.concatenate(verts).astype(np.float32)
    f = np.ascontiguousarray(np.concatenate(faces), np.int32)
    nrm = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-6)
    return MeshData(v, nrm.astype(np.float32), None, f, f.copy(), None, "hugeleaf%d_%d" % (n, fan))


def write_obj(mesh: MeshData, path: str) -> None:
    """OBJ text with ``vn`` and ``f a//a`` (or ``a/t/n``) faces.  SURVEY.md §4 pitfall 1: an OBJ
    without ``vn`` renders black in the reference, so normals are always written."""
    with open(path, "w") as f:
        np.savetxt(f, mesh.vertices, fmt="v %.9g %.9g %.9g")
        np.savetxt(f, mesh.normals, fmt="vn %.9g %.9g %.9g")
        if mesh.uvs is not None:
            np.savetxt(f, mesh.uvs, fmt="vt %.9g %.9g")
            cols = np.stack([mesh.faces_v[:, 0], mesh.faces_t[:, 0], mesh.faces_n[:, 0],
                             mesh.faces_v[:, 1], mesh.faces_t[:, 1], mesh.faces_n[:, 1],
                             mesh.faces_v[:, 2], mesh.faces_t[:, 2], mesh.faces_n[:, 2]], -1) + 1
            np.savetxt(f, cols, fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")
        else:
            cols = np.stack([mesh.faces_v[:, 0], mesh.faces_n[:, 0], mesh.faces_v[:, 1], mesh.faces_n[:, 1],
                             mesh.faces_v[:, 2], mesh.faces_n[:, 2]], -1) + 1
            np.savetxt(f, cols, fmt="f %d//%d %d//%d %d//%d")


@dataclass
class RenderConfig:
    """Mirror of the Raytracer fields the hot path reads (Raytracer.h:73-111)."""
    W: int = 256
    H: int = 256
    spp: int = 64
    nb_bounces: int = 4
    sigma_filter: float = 0.5
    cam_pos: tuple = (0.0, 0.0, 50.0)
    cam_dir: tuple = (0.0, 0.0, -1.0)
    cam_up: tuple = (0.0, 1.0, 0.0)
    fov: float = float(np.float32(35 * np.pi / 180))
    focus: float = 50.0
    aperture: float = 0.1
    light_center: tuple = (10.0, 23.0, 15.0)
    light_radius: float = 10.0
    light_scale: float = 1.0
    envmap_intensity: float = 1.0
    mesh_scale: float = 30.0


def default_camera_rotated():
    """The reference's loadScene() camera after cam.rotate(0, -22 deg, 1) (Raytracer.cpp:1250,1273;
    Vector.h:725-750).  The float32 values are the ones the compiled reference produces (libm
    cosf/sinf of float(-22*pi/180)); tests/test_oracle_vs_reference.py checks them against it."""
    c = float.fromhex("0x1.dab7d8p-1")   # cosf(22 deg)
    s = float.fromhex("0x1.7f98dep-2")   # sinf(22 deg)
    return (0.0, -s, -c), (0.0, c, -s)


def config_c0() -> RenderConfig:
    """BASELINE.json configs[0]: 12-triangle Cornell-style scene, 256x256, 64 spp, depth 4."""
    return RenderConfig(W=256, H=256, spp=64, nb_bounces=4, cam_pos=(0.0, -12.3, 55.0), aperture=0.01,
                        light_center=(0.0, -3.0, 0.0), light_radius=3.0, light_scale=0.04)


def config_c1(W=1920, H=1080, spp=256) -> RenderConfig:
    """BASELINE.json configs[1]: 133k-triangle diffuse blob, 1080p, 256 spp, depth 4, default
    loadScene() light and camera."""
    d, u = default_camera_rotated()
    return RenderConfig(W=W, H=H, spp=spp, nb_bounces=4, cam_dir=d, cam_up=u)


def checker_texture(W=64, H=32, seed=7, cells=8):
    """Procedural RGB8 albedo: coloured checker modulated by value noise (config C2's Kd texture)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    chk = ((xx * cells // W + yy * cells // H) & 1).astype(np.float64)
    coarse = rng.uniform(0.55, 1.0, (cells + 1, cells + 1, 3))
    noise = coarse[(yy * cells // H)[..., None], (xx * cells // W)[..., None], np.arange(3)]
    base = np.where(chk[..., None] > 0, np.array([0.85, 0.35, 0.25]), np.array([0.25, 0.55, 0.85]))
    return np.clip(base * noise * 255.0, 0, 255).astype(np.uint8)


def sky_envmap(W=128, H=64):
    """Procedural RGB8 environment map: vertical gradient plus a bright patch (config C2's env map)."""
    yy, xx = np.mgrid[0:H, 0:W]
    t = yy / (H - 1.0)
    img = np.stack([40 + 60 * t, 70 + 90 * t, 140 + 100 * t], -1)
    sun = np.exp(-(((xx - 0.3 * W) / (0.05 * W)) ** 2 + ((yy - 0.75 * H) / (0.08 * H)) ** 2))
    img += 120 * sun[..., None]
    return np.clip(img, 0, 255).astype(np.uint8)


def alpha_texture(W=32, H=32):
    """RGB8 cut-out mask: opaque (255) except a grid of round holes (0)."""
    yy, xx = np.mgrid[0:H, 0:W]
    hole = (((xx % 8) - 3.5) ** 2 + ((yy % 8) - 3.5) ** 2) < 5.0
    a = np.where(hole, 0, 255).astype(np.uint8)
    return np.stack([a, a, a], -1)


def bump_texture(W=32, H=32, seed=3):
    """RGB8 tangent-space normal map (128 = 0), gentle random tilt."""
    rng = np.random.default_rng(seed)
    n = np.stack([rng.normal(0, 25, (H, W)), rng.normal(0, 25, (H, W)), np.full((H, W), 110.0)], -1) + 128
    return np.clip(n, 0, 255).astype(np.uint8)


def synthetic_merl_table(kd=(0.25, 0.2, 0.15), ks=0.4, shininess=60.0):
    """Analytic Blinn-like lobe sampled into the MERL isotropic layout (config C4's measured BRDF):
    3 colour planes x 90 theta_half x 90 theta_diff x 180 phi_diff doubles, stored pre-divided by
    the per-channel scales the reader multiplies back (1, 1.15, 1.66)/1500."""
    th_i = np.arange(90)
    theta_half = (th_i + 0.5) ** 2 / 90.0 * (np.pi / 2) / 90.0     # inverse of theta_half_index (sqrt mapping)
    theta_diff = (np.arange(90) + 0.5) / 90.0 * (np.pi / 2)
    spec = ks * (shininess + 2) / (2 * np.pi) * np.cos(theta_half) ** shininess
    fres = 0.04 + 0.96 * (1 - np.cos(theta_diff)) ** 5
    val = spec[:, None, None] * (0.5 + fres)[None, :, None] * np.ones(180)[None, None, :]
    planes = []
    for c, scale in zip(range(3), (1.0 / 1500.0, 1.15 / 1500.0, 1.66 / 1500.0)):
        planes.append((kd[c] / np.pi + val) / scale)
    return np.ascontiguousarray(np.stack(planes, 0), np.float64)


# ---------------------------------------------------------------- BASELINE.json configs as workloads
def workload(name: str, width: int = None, height: int = None, spp: int = None, grid: int = None):
    """(mesh, cfg, material, description) of BASELINE.json configs[1..4] (SURVEY.md §8d synthetic inputs).
    `material` is None (OBJ/MTL defaults) or a dict understood by install()."""
    name = name.lower()
    if name == "c1":      # 133k-triangle diffuse blob, 1080p, 256 spp, depth 4
        g = grid or 258
        cfg = config_c1(width or 1920, height or 1080, spp or 256)
        return blob_mesh(g), cfg, None, f"configs[1]: {2 * g * g}-triangle diffuse blob, Phong BRDF"
    if name == "c1g":     # configs[1] with a glossy Phong lobe on the mesh (not a BASELINE config: exercises the general shade tier)
        g = grid or 258
        cfg = config_c1(width or 1920, height or 1080, spp or 256)
        mat = dict(Kd=(0.5, 0.4, 0.3), Ks=(0.3, 0.3, 0.3), Ne=(50.0, 50.0, 50.0))
        return blob_mesh(g), cfg, mat, f"configs[1] with a glossy lobe: {2 * g * g}-triangle blob, Phong Ks 0.3 Ne 50"
    if name == "c2":      # 2.5M triangles, Kd texture, env map, 1024 spp
        g = grid or 1120
        cfg = config_c1(width or 1920, height or 1080, spp or 1024)
        mat = dict(Kd=(1.0, 1.0, 1.0), Ks=(0.0, 0.0, 0.0), Ne=(0.0, 0.0, 0.0),
                   tex={0: checker_texture(2048, 2048, 7, 64)}, envmap=sky_envmap(4096, 2048))
        return blob_mesh(g, with_uv=True), cfg, mat, f"configs[2]: {2 * g * g}-triangle blob, 2048x2048 Kd texture, 4096x2048 env map, Phong BRDF"
    if name == "c3":      # same mesh, fully transparent (Fresnel dielectric n = 1.3), depth 12
        g = grid or 1120
        cfg = config_c1(width or 1920, height or 1080, spp or 1024)
        cfg.nb_bounces = 12
        mat = dict(Kd=(0.5, 0.5, 0.5), Ks=(0.0, 0.0, 0.0), Ne=(0.0, 0.0, 0.0), transp=0.0, refr=1.3)
        return blob_mesh(g), cfg, mat, f"configs[3]: {2 * g * g}-triangle dielectric blob (n=1.3), depth 12"
    if name == "c4":      # 23.7M triangles, MERL-layout measured BRDF, depth of field, 4K, 4096 spp
        g = grid or 3444
        cfg = config_c1(width or 3840, height or 2160, spp or 4096)
        cfg.aperture = 0.5
        cfg.focus = 50.0
        mat = dict(Kd=(0.5, 0.5, 0.5), Ks=(0.0, 0.0, 0.0), Ne=(0.0, 0.0, 0.0), merl=synthetic_merl_table())
        return blob_mesh(g, fine_detail=True), cfg, mat, f"configs[4]: {2 * g * g}-triangle blob, MERL-layout BRDF table, aperture 0.5"
    raise KeyError(name)


def install(X, mesh, mat):
    """Adds the workload's mesh and material to X: a capi.HostRaytracer, an oracle.binding.Oracle or a
    binding.Ref (they share these method names).  Returns the object id."""
    oid = X.add_mesh(mesh)
    install_material(X, oid, mat)
    return oid


def install_material(X, oid, mat):
    """The workload's material, textures, BRDF table and environment map on object `oid` of X."""
    if mat is not None:
        X.set_group_material(oid, 0, mat["Kd"], mat["Ks"], mat["Ne"], mat.get("transp", 1.0), mat.get("refr", 1.3))
        for slot, img in mat.get("tex", {}).items():
            X.set_group_texture(oid, 0, slot, img)
        if "merl" in mat:
            X.set_brdf_merl(oid, mat["merl"])
        if "envmap" in mat:
            X.set_envmap(mat["envmap"])


# ---------------------------------------------------------------- an OBJ / MTL / PPM scene on disk
def write_ppm(path: str, rgb8: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb8.shape[1], rgb8.shape[0]))
        f.write(np.ascontiguousarray(rgb8, np.uint8).tobytes())


def write_obj_scene(directory: str, n: int = 10) -> str:
    """A blob written the way real OBJ files are: quads and a pentagon fan (polygon triangulation), 1-based and
    negative (relative) indices, v/t/n and v//n corners, three usemtl groups, an MTL with Kd / Ks / Ns (one and three
    values), map_Kd / map_Bump / map_d images (binary PPM), an indented record and a material the OBJ never uses.
    Returns the path of the .obj."""
    import os
    m = blob_mesh(n, with_uv=True)
    nv = m.vertices.shape[0]
    L = ["# synthetic OBJ scene", "mtllib scene.mtl"]
    L += ["v %.9g %.9g %.9g" % tuple(v) for v in m.vertices.tolist()]
    L += ["vn %.9g %.9g %.9g" % tuple(v) for v in m.normals.tolist()]
    L += ["vt %.9g %.9g" % tuple(v) for v in m.uvs.tolist()]
    quad = lambda i, j: (i * (n + 1) + j, i * (n + 1) + j + 1, (i + 1) * (n + 1) + j + 1, (i + 1) * (n + 1) + j)
    for i in range(n):
        if i == 0:
            L.append("usemtl skin")
        elif i == n // 3:
            L.append("usemtl   bumpy")          # blanks after the keyword are skipped by the reference's sscanf
        elif i == 2 * n // 3:
            L.append("usemtl plain")
        j = 0
        while j < n:
            a, b, c, d = quad(i, j)
            if i < n // 3:                      # quads, 1-based v/t/n
                L.append("f " + " ".join("%d/%d/%d" % (k + 1, k + 1, k + 1) for k in (a, b, c, d)))
                j += 1
            elif i < 2 * n // 3:                # quads, negative indices (relative to the end of the arrays)
                L.append("f " + " ".join("%d/%d/%d" % (k - nv, k - nv, k - nv) for k in (a, b, c, d)) + " ")
                j += 1
            elif j + 1 < n:                     # two quads merged into one hexagon fan, v//n corners
                a2, b2, c2, d2 = quad(i, j + 1)
                L.append("f " + " ".join("%d//%d" % (k + 1, k + 1) for k in (a, b, b2, c2, c, d)))
                j += 2
            else:
                L.append("f " + " ".join("%d//%d" % (k + 1, k + 1) for k in (a, b, c, d)))
                j += 1
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "scene.obj"), "w") as f:
        f.write("\n".join(L) + "\n")
    with open(os.path.join(directory, "scene.mtl"), "w") as f:
        f.write("newmtl skin\nKd 0.9 0.8 0.7\nKs 0.2 0.2 0.1\nNs 35\nmap_Kd kd.ppm\n\tKd 9 9 9\n"
                "newmtl bumpy\nKd 0.3 0.6 0.9\nNs 10 20 30\nmap_Bump bump.ppm\nmap_d alpha.ppm\n"
                "newmtl plain\nKd 0.5 0.4 0.3\n"
                "newmtl never_used\nKs 0.05 0.06 0.07\n")
    write_ppm(os.path.join(directory, "kd.ppm"), checker_texture(32, 16, 3, 4))
    write_ppm(os.path.join(directory, "bump.ppm"), bump_texture())
    write_ppm(os.path.join(directory, "alpha.ppm"), alpha_texture())
    return os.path.join(directory, "scene.obj")
