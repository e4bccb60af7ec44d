"""Generates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libptref.so).

Run in the build container only (needs /root/reference to have been compiled by
`make -f oracle/Makefile ref`):   python tests/golden/make_golden.py

The files hold data only — inputs and the reference's outputs (SURVEY.md §8c list i–viii) — so that
the oracle and the HIP path can be checked on the GPU box, where the reference does not exist.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Ref  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def golden_scene(name):
    """(mesh, cfg, materials) of the named golden scene; shared with the tests."""
    if name == "cornell":
        cfg = scenes.config_c0()
        cfg.W, cfg.H, cfg.spp = 64, 64, 16
        return scenes.cornell_mesh(), cfg, None
    if name == "blob32":
        cfg = scenes.config_c1(96, 54, 8)
        return scenes.blob_mesh(32), cfg, None
    if name == "glossy":   # Phong lobe + mirror-ish plane material
        cfg = scenes.config_c1(64, 36, 8)
        return scenes.blob_mesh(24), cfg, dict(Kd=(0.4, 0.3, 0.2), Ks=(0.5, 0.5, 0.4), Ne=(40.0, 60.0, 80.0))
    if name == "glass":    # Fresnel dielectric, deep paths
        cfg = scenes.config_c1(64, 36, 8)
        cfg.nb_bounces = 8
        return scenes.blob_mesh(24), cfg, dict(Kd=(0.5, 0.5, 0.5), Ks=(0, 0, 0), Ne=(0, 0, 0), transp=0.0, refr=1.3)
    if name == "textured":   # config C2 in small: UV mesh, Kd / Ks / Ne image textures, env map
        cfg = scenes.config_c1(64, 36, 8)
        return scenes.blob_mesh(24, with_uv=True), cfg, dict(Kd=(1.0, 1.0, 1.0), Ks=(0.3, 0.3, 0.3), Ne=(20.0, 20.0, 20.0),
                                                                   tex={0: scenes.checker_texture()}, envmap=scenes.sky_envmap())
    if name == "cutout":     # alpha-map rejection inside the leaf loop + tangent-space normal map
        cfg = scenes.config_c1(64, 36, 8)
        return scenes.blob_mesh(24, with_uv=True), cfg, dict(Kd=(0.8, 0.8, 0.8), Ks=(0, 0, 0), Ne=(0, 0, 0),
                                                                   tex={3: scenes.alpha_texture(), 2: scenes.bump_texture(), 0: scenes.checker_texture(16, 16, 2, 4)})
    if name == "merl":       # config C4 in small: measured (MERL-layout) BRDF + thin-lens depth of field
        cfg = scenes.config_c1(64, 36, 8)
        cfg.aperture = 0.5
        return scenes.blob_mesh(24), cfg, dict(Kd=(0.5, 0.5, 0.5), Ks=(0, 0, 0), Ne=(0, 0, 0), merl=scenes.synthetic_merl_table())
    if name == "c0full":
        return scenes.cornell_mesh(), scenes.config_c0(), None
    raise KeyError(name)


def setup(X, name):
    mesh, cfg, mat = golden_scene(name)
    X.apply_config(cfg)
    oid = X.add_mesh(mesh)
    if mat is not None:
        X.set_group_material(oid, 0, mat["Kd"], mat["Ks"], mat["Ne"], mat.get("transp", 1.0), mat.get("refr", 1.3))
        for slot, img in mat.get("tex", {}).items():
            X.set_group_texture(oid, 0, slot, img)
        if "merl" in mat:
            X.set_brdf_merl(oid, mat["merl"])
        if "envmap" in mat:
            X.set_envmap(mat["envmap"])
    X.prepare()
    return mesh, cfg, oid


def all_pixels(cfg):
    return np.stack(np.meshgrid(np.arange(cfg.H), np.arange(cfg.W), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)


def main():
    rng = np.random.default_rng(20261002)
    R = Ref()
    g = {}
    # (i) pcg32, (ii) lattice, helpers
    for s in (0, 1, 42, 123456789012345):
        g[f"pcg32_{s}"] = R.pcg32(s, 16)
    g["lattice"] = R.lattice(64)
    x = np.concatenate([rng.uniform(0, 1e4, 500), 10.0 ** rng.uniform(-30, 30, 500), [0.0, 1.0]]).astype(np.float32)
    g["invsqroot_in"], g["invsqroot_out"] = x, R.invsqroot(x)
    y = rng.uniform(-20, 1, 1000)
    g["fast_exp_in"], g["fast_exp_out"] = y, R.fast_exp(y)
    v = rng.normal(size=(1000, 3)).astype(np.float32)
    N = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    r12 = rng.uniform(0, 1, (1000, 2)).astype(np.float32)
    g["random_cos_N"], g["random_cos_r"], g["random_cos_out"] = N, r12, R.random_cos(N, r12)
    # (vi) Phong sample / eval on random tuples
    n = 256
    mat9 = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0, 0.9, (n, 3)), rng.uniform(0, 200, (n, 3))], 1).astype(np.float32)
    mat9[: n // 4, 3:6] = 0      # pure diffuse
    mat9[: n // 8, 6:9] = 0      # Ne = 0 (OBJ default)
    Nn = N[:n]
    wo = rng.normal(size=(n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True); wo = np.where((wo * Nn).sum(1, keepdims=True) < 0, -wo, wo).astype(np.float32)
    wi = rng.normal(size=(n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True); wi = np.where((wi * Nn).sum(1, keepdims=True) < 0, -wi, wi).astype(np.float32)
    seeds = rng.integers(0, 2 ** 40, n).astype(np.uint64)
    g["phong_mat9"], g["phong_wo"], g["phong_wi"], g["phong_N"], g["phong_r12"], g["phong_seeds"] = mat9, wo, wi, Nn, r12[:n], seeds
    g["phong_sample"] = R.phong_sample(mat9, wo, Nn, r12[:n], seeds)
    g["phong_eval"] = R.phong_eval(mat9, wi, wo, Nn)
    np.savez_compressed(os.path.join(OUT, "leaf_functions.npz"), **g)

    # per-scene goldens: (iii) camera, (iv) BVH, (v) rays, (vii) per-sample radiance
    for name in ("cornell", "blob32", "glossy", "glass", "textured", "cutout", "merl"):
        R = Ref()
        mesh, cfg, oid = setup(R, name)
        g = {}
        g["light"] = R.light()
        rpp, s2d, fi, fs = R.tables()
        g["randomPerPixel"], g["samples2d"], g["filter_integral"], g["filter_size"] = rpp, s2d, fi, np.int32(fs)
        for k in range(oid + 1):
            t, inv, r = R.object_matrices(k)
            g[f"obj{k}_trans"], g[f"obj{k}_inv"], g[f"obj{k}_rot"] = t, inv, r
        d = R.mesh_dump(oid)
        g["perm"], g["nodes_i"], g["nodes_bb"], g["groups"], g["root_bb"] = d["perm"], d["nodes_i"], d["nodes_bb"], d["groups"], d["root_bb"]
        g["soup16"] = d["soup"][:, :16]
        g["soup_normals"] = d["soup"][:, 22:31]
        ij = all_pixels(cfg)
        sel = rng.choice(ij.shape[0], 2048, replace=False)
        jit = rng.uniform(-0.5, 0.5, (2048, 4)).astype(np.float32)
        jit[:, 2:] *= np.float32(cfg.aperture)
        g["cam_ij"], g["cam_jit"] = ij[sel], jit
        rays = R.camera_rays(ij[sel], jit)
        g["cam_rays"] = rays
        hi, hf = R.intersect(rays)
        P = hf[:, 1:4]
        d2 = rng.normal(size=P.shape).astype(np.float32)
        d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        rays2 = np.concatenate([np.where(hi[:, :1] > 0, P + np.float32(0.01) * d2, rays[:, :3]), d2], 1).astype(np.float32)
        allrays = np.concatenate([rays, rays2], 0)
        hi, hf = R.intersect(allrays)
        g["rays"], g["hit_i"], g["hit_f"] = allrays, hi, hf[:, :19]
        dist = rng.uniform(1, 80, allrays.shape[0]).astype(np.float32)
        g["shadow_dist"], g["shadow_occluded"] = dist, R.intersect_shadow(allrays, dist)
        rgb, dxdy = R.getcolor_samples(ij, 0, cfg.spp)
        g["sample_rgb"], g["sample_dxdy"] = rgb, dxdy
        img, cnt = R.render_seeded()
        g["image"], g["count"] = img, cnt
        np.savez_compressed(os.path.join(OUT, f"scene_{name}.npz"), **g)
        print(name, "mean radiance / white =", float(rgb.mean() / 196964.7))

    # OBJ / MTL / PPM ingestion: the scene files are regenerated by scenes.write_obj_scene, the
    # reference reads them with its own readOBJ + stb_image; what it made of them is the fixture
    import tempfile
    R = Ref()
    cfg = scenes.config_c1(64, 36, 4)
    R.apply_config(cfg)
    oid = R.add_mesh_obj(scenes.write_obj_scene(tempfile.mkdtemp(prefix="ptref_objscene_")))
    R.prepare()
    d = R.mesh_dump(oid)
    g = dict(perm=d["perm"], nodes_i=d["nodes_i"], nodes_bb=d["nodes_bb"], groups=d["groups"], root_bb=d["root_bb"],
             soup16=d["soup"][:, :16], soup_normals=d["soup"][:, 22:31], soup_uvs=d["soup"][:, 16:22])
    for k, (m, wh) in enumerate(R.group_materials(oid)):
        g[f"mat{k}"], g[f"mat{k}_wh"] = m, wh
        for slot in range(4):
            t = R.group_texture(oid, k, slot)
            if t is not None:
                g[f"tex{k}_{slot}"] = t
    rgb, dxdy = R.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
    g["sample_rgb"], g["sample_dxdy"] = rgb, dxdy
    np.savez_compressed(os.path.join(OUT, "objscene.npz"), **g)
    print("objscene mean radiance / white =", float(rgb.mean() / 196964.7))

    # .scn scene files: the reference SAVES the OBJ scene with a changed camera / depth, that text is
    # the fixture (tests/golden/objscene.scn); a second reference instance LOADS it and renders: expected state + radiance
    d = tempfile.mkdtemp(prefix="ptref_scn_")
    cwd = os.getcwd()
    os.chdir(d)                      # the reference resolves the relative file names of a .scn against the working directory
    try:
        scenes.write_obj_scene(d)
        R = Ref()
        cfg = scenes.config_c1(64, 36, 4)
        cfg.aperture, cfg.nb_bounces = 0.25, 5
        R.apply_config(cfg)
        R.add_mesh_obj("scene.obj")
        R.save_scene("objscene.scn")
        text = open("objscene.scn").read()
        R2 = Ref()
        R2.load_scene("objscene.scn")
        R2.prepare()
        g = dict(header=R2.scene_header())
        for k in range(R2.num_objects()):
            g[f"obj{k}_state"], g[f"obj{k}_flags"] = R2.object_state(k)
        rgb, dxdy = R2.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
        g["sample_rgb"], g["sample_dxdy"] = rgb, dxdy
    finally:
        os.chdir(cwd)
    open(os.path.join(OUT, "objscene.scn"), "w").write(text)
    np.savez_compressed(os.path.join(OUT, "objscene_scn.npz"), **g)
    print("objscene.scn mean radiance / white =", float(rgb.mean() / 196964.7))

    # (viii) full C0 image 256x256x64spp, stored normalised
    R = Ref()
    mesh, cfg, oid = setup(R, "c0full")
    img, cnt = R.render_seeded()
    np.savez_compressed(os.path.join(OUT, "c0_image.npz"), image=img, count=cnt)
    print("c0 mean", float((img / np.maximum(cnt, 1)[..., None]).mean() / 196964.7))


if __name__ == "__main__" and "--spheres" not in sys.argv and "--keyframes" not in sys.argv and "--denoiser-inputs" not in sys.argv and "--compositing" not in sys.argv and "--fog" not in sys.argv and "--subsurface" not in sys.argv and "--jpeg" not in sys.argv and "--lenticular" not in sys.argv and "--image-writers" not in sys.argv:
    main()


def main_denoiser_inputs():
    """tests/golden/denoiser_inputs.npz: getColor's normalValue / albedoValue per sample and the has_denoiser sums
    (Raytracer.cpp:255-258, 1631-1645) of three golden scenes, from the compiled reference."""
    g = {}
    for name in ("textured", "cutout", "glass"):
        R = Ref()
        mesh, cfg, oid = setup(R, name)
        rgb, nrm, alb = R.getcolor_samples_aov(all_pixels(cfg), 0, 2)
        g[name + "_rgb"], g[name + "_normal"], g[name + "_albedo"] = rgb, nrm, alb
        if name == "textured":
            img, cnt, a, n = R.render_denoiser_inputs()
            g[name + "_img"], g[name + "_cnt"], g[name + "_albedo_sum"], g[name + "_normal_sum"] = img, cnt, a, n
    np.savez_compressed(os.path.join(OUT, "denoiser_inputs.npz"), **g)


if __name__ == "__main__" and "--denoiser-inputs" in sys.argv:
    main_denoiser_inputs()


# ---------------------------------------------------------------- ghost objects / background photo
COMPOSITING_KINDS = ("bgonly", "plane", "mesh", "both", "planenobg", "glossyghost", "mirrorghost")


def background_photo(W=48, H=32):
    """Scene::background as load_background leaves it: pow(v/255., 2.2) * 196964.699 of an 8-bit test pattern."""
    y, x = np.mgrid[0:H, 0:W]
    img = np.stack([(x * 5) % 256, (y * 7) % 256, ((x + y) * 3) % 256], -1).astype(np.float64)
    return (np.power(img / 255., 2.2) * 196964.699).astype(np.float32)


def compositing_scene(X, kind):
    """A blob over the floor plane with ghost objects and / or a background photo; X: Ref, Oracle or HostRaytracer."""
    cfg = scenes.config_c1(64, 36, 4)
    cfg.nb_bounces = 4
    X.apply_config(cfg)
    oid = X.add_mesh(scenes.blob_mesh(16))
    if kind in ("plane", "both", "planenobg"):
        X.set_object_ghost(2, True)                      # the floor catches the blob's shadow on the photo
    if kind in ("mesh", "both"):
        o2 = X.add_mesh(scenes.blob_mesh(8), scale=45.0)  # a closed ghost mesh around the blob: paths pass through it at the same depth
        X.set_object_ghost(o2, True)
    if kind == "glossyghost":
        X.set_group_material(oid, 0, (0.4, 0.3, 0.2), (0.5, 0.5, 0.4), (40., 60., 80.))
        X.set_object_ghost(oid, True)
    if kind == "mirrorghost":
        X.set_object_flags(oid, True, False)             # a ghost that is a mirror: the mirror branch comes first (:413)
        X.set_object_ghost(oid, True)
        X.set_object_ghost(2, True)
    if kind != "planenobg":
        X.set_background(background_photo())
    X.prepare()
    return cfg


FOG_KINDS = ("uniform", "dense", "exp", "schlick", "rayleigh", "glossyfog", "mirrorfog", "glassfog", "ghostfog")


def fog_scene(X, kind):
    """Blob over the floor in a participating medium (fogContribution, Raytracer.cpp:45-192)."""
    cfg = scenes.config_c1(64, 36, 4)
    cfg.nb_bounces = 4
    X.apply_config(cfg)
    oid = X.add_mesh(scenes.blob_mesh(16))
    fog = dict(uniform=(0.5, 0.4), dense=(3.0, 2.0), exp=(0.8, 0.6, 0.05, 0.04, 1, 0, 0.0), schlick=(0.5, 0.4, 0, 0, 0, 1, 0.6),
               rayleigh=(0.5, 0.4, 0.02, 0.03, 1, 2, 0.0)).get(kind, (0.5, 0.4))
    if kind == "glossyfog":
        X.set_group_material(oid, 0, (0.4, 0.3, 0.2), (0.5, 0.5, 0.4), (40., 60., 80.))
    if kind == "mirrorfog":
        X.set_object_flags(oid, True, False)
    if kind == "glassfog":
        X.set_group_material(oid, 0, (0.5, 0.5, 0.5), (0, 0, 0), (0, 0, 0), 0.0, 1.3)
    if kind == "ghostfog":
        X.set_object_ghost(2, True)
        X.set_background(background_photo())
        o2 = X.add_mesh(scenes.blob_mesh(8), scale=45.0)
        X.set_object_ghost(o2, True)
    X.set_fog(*fog)
    X.prepare()
    return cfg


def main_fog():
    """tests/golden/fog.npz: per-sample radiance of the fog scenes from the compiled reference."""
    g = {}
    for kind in FOG_KINDS:
        R = Ref()
        cfg = fog_scene(R, kind)
        rgb, dxdy = R.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
        g[kind + "_rgb"] = rgb
        print(kind, "mean radiance / white", float(rgb.mean() / 196964.7))
    np.savez_compressed(os.path.join(OUT, "fog.npz"), **g)


SSS_KINDS = ("ss", "ssglossy", "ssfog", "ssghost", "sstex", "ssdeep", "ssplane", "ssimage", "sssphere", "ssspheretex", "ssbare")


def subsurface_scene(X, kind):
    """A blob with a subsurface colour (Raytracer.cpp:318-406: with probability 0.6 the path leaves through a random nearby
    point of the same object found by Scene::get_random_intersection)."""
    cfg = scenes.config_c1(64, 36, 4)
    cfg.nb_bounces = 8 if kind == "ssdeep" else 4
    X.apply_config(cfg)
    oid = X.add_mesh(scenes.blob_mesh(16, with_uv=(kind in ("sstex", "ssimage")), fine_detail=(kind == "ssdeep")))
    X.set_group_subsurface(oid, 0, (0.8, 0.5, 0.3))
    if kind == "ssplane":                                # the floor scatters as well (Plane::reservoir_sampling_intersection)
        X.add_col_subsurface(2, (0.7, 0.6, 0.2))
    if kind == "ssimage":                                # the subsurface colour comes from an image (Object::set_subsurface)
        X.set_group_subsurface(oid, 0, (1.0, 1.0, 1.0))
        X.set_group_texture(oid, 0, 7, scenes.checker_texture(32, 16, 9, 4))
    if kind == "ssglossy":
        X.set_group_material(oid, 0, (0.4, 0.3, 0.2), (0.5, 0.5, 0.4), (40., 60., 80.))
    if kind == "ssfog":
        X.set_fog(0.5, 0.4)
    if kind == "ssghost":
        X.set_object_ghost(2, True)
        X.set_background(background_photo())
    if kind == "sstex":
        X.set_group_texture(oid, 0, 0, scenes.checker_texture())
        X.set_envmap(scenes.sky_envmap())
    if kind in ("sssphere", "ssspheretex"):               # round 4: a sphere with material lists scatters too (Sphere::reservoir_sampling_intersection,
        sp = X.add_sphere((16, -16, 6), 9.0)              # Geometry.h:994-1068: both roots in root order, spherical coordinates with double intermediates)
        X.add_group_material(sp, (0.8, 0.7, 0.6), (0.1, 0.1, 0.1), (30, 30, 30), 1.0, 1.3)
        X.add_col_subsurface(sp, (0.9, 0.4, 0.2))
        sp2 = X.add_sphere((-18, -19, 10), 7.0, flip_normals=True)
        X.add_group_material(sp2, (0.5, 0.6, 0.9), (0, 0, 0), (0, 0, 0), 1.0, 1.3)
        X.add_col_subsurface(sp2, (0.3, 0.5, 0.9))
        if kind == "ssspheretex":                         # the sphere's Kd and subsurface colour from images, looked up at (theta, phi)
            X.set_group_texture(sp, 0, 0, scenes.checker_texture(32, 16, 5, 4))
            X.set_group_subsurface(sp, 0, (1.0, 1.0, 1.0))
            X.set_group_texture(sp, 0, 7, scenes.checker_texture(32, 16, 9, 4))
    if kind == "ssbare":                                  # spheres WITHOUT lists (one a mirror) beside subsurface colours: Ksub is inherited from the
        X.add_col_subsurface(2, (0.7, 0.6, 0.2))          # object tested before them, like Kd / Ks / Ne (Geometry.cpp:596), and read before the mirror branch
        X.add_sphere((16, -16, 6), 9.0)
        X.add_sphere((-18, -19, 10), 7.0, mirror=True)
    X.prepare()
    return cfg


def main_subsurface():
    """tests/golden/subsurface.npz: per-sample radiance of the subsurface scenes from the compiled reference."""
    g = {}
    for kind in SSS_KINDS:
        R = Ref()
        cfg = subsurface_scene(R, kind)
        rgb, dxdy = R.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
        g[kind + "_rgb"] = rgb
        print(kind, "mean radiance / white", float(rgb.mean() / 196964.7))
    R = Ref()
    cfg = subsurface_scene(R, "ss")
    R2 = Ref()                       # the same scene without the subsurface colour, to show the branch is taken
    cfg2 = scenes.config_c1(64, 36, 4); cfg2.nb_bounces = 4
    R2.apply_config(cfg2); R2.add_mesh(scenes.blob_mesh(16)); R2.prepare()
    g["plain_rgb"] = R2.getcolor_samples(all_pixels(cfg2), 0, cfg2.spp)[0]
    np.savez_compressed(os.path.join(OUT, "subsurface.npz"), **g)


def main_compositing():
    """tests/golden/compositing.npz: per-sample radiance of the ghost / background scenes from the compiled reference."""
    g = {}
    for kind in COMPOSITING_KINDS:
        R = Ref()
        cfg = compositing_scene(R, kind)
        rgb, dxdy = R.getcolor_samples(all_pixels(cfg), 0, cfg.spp)
        g[kind + "_rgb"] = rgb
        if kind == "both":
            img, cnt = R.render_seeded()
            g["both_img"], g["both_cnt"] = img, cnt
        print(kind, "mean radiance / white", float(rgb.mean() / 196964.7))
    np.savez_compressed(os.path.join(OUT, "compositing.npz"), **g)


if __name__ == "__main__" and "--compositing" in sys.argv:
    main_compositing()
if __name__ == "__main__" and "--fog" in sys.argv:
    main_fog()
if __name__ == "__main__" and "--subsurface" in sys.argv:
    main_subsurface()


def main_jpeg():
    """tests/golden/jpeg_cases.npz: small JPEG files (written by Pillow from a synthetic pattern: baseline / progressive, 4:4:4 /
    4:2:2 / 4:2:0, grey, restart markers, optimised tables, odd sizes) and the pixels the reference's load_image (stb_image)
    makes of them, rows as in the file."""
    import ctypes as C
    import io
    from PIL import Image
    rng = np.random.default_rng(11)

    def pattern(h, w):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(x / 7.) * np.cos(y / 5.), 127 + 100 * np.cos(x / 3. + y / 11.), (x * 3 + y * 5) % 256], -1)
        return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    R = Ref()
    g = {}
    cases = [(37, 53, 0, 90, False, False, 0), (37, 53, 1, 35, False, False, 0), (37, 53, 2, 75, False, False, 0), (37, 53, 2, 75, True, False, 0),
             (16, 16, 1, 90, True, False, 0), (1, 1, 2, 80, False, False, 0), (8, 9, 2, 35, True, False, 0), (40, 56, 0, 80, False, True, 0),
             (40, 56, 0, 80, True, True, 0), (33, 47, 2, 75, False, False, 2), (33, 47, 1, 75, True, False, 1), (3, 120, 2, 60, False, False, 0)]
    tmp = os.path.join(OUT, "_tmp.jpg")
    for n, (h, w, sub, q, prog, grey, rst) in enumerate(cases):
        img = pattern(h, w)
        im = Image.fromarray(img[..., 0] if grey else img, "L" if grey else "RGB")
        kw = dict(quality=q, progressive=prog, optimize=(q == 35))
        if not grey:
            kw["subsampling"] = sub
        if rst:
            kw["restart_marker_rows"] = rst
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        open(tmp, "wb").write(buf.getvalue())
        W, H = C.c_int(0), C.c_int(0)
        out = (C.c_ubyte * (1 << 20))()
        assert R.lib.ref_load_image(tmp.encode(), out, len(out), C.byref(W), C.byref(H)) == 0
        g[f"file{n}"] = np.frombuffer(buf.getvalue(), np.uint8)
        g[f"rgb{n}"] = np.frombuffer(out, np.uint8, W.value * H.value * 3).reshape(H.value, W.value, 3)[::-1].copy()   # load_image flips the rows
    os.remove(tmp)
    g["count"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(OUT, "jpeg_cases.npz"), **g)
    print("jpeg cases", len(cases))


if __name__ == "__main__" and "--jpeg" in sys.argv:
    main_jpeg()


LENTICULAR_KINDS = ("default", "wide", "dof")


def lenticular_scene(X, kind):
    """Camera::generateDirection with is_lenticular (Vector.h:799-812): pixel column j is seen from one of nb_images cameras."""
    cfg = scenes.config_c1(64, 36, 3)
    cfg.nb_bounces = 3
    if kind == "dof":
        cfg.aperture = 0.5
    X.apply_config(cfg)
    X.add_mesh(scenes.blob_mesh(16))
    X.set_lenticular(True, *((10, 35 * np.pi / 180. * 0.25, 1) if kind != "wide" else (6, 0.4, 3)))
    X.prepare()
    return cfg


def main_lenticular():
    g = {}
    for kind in LENTICULAR_KINDS:
        R = Ref()
        cfg = lenticular_scene(R, kind)
        g[kind + "_rgb"] = R.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0]
    R = Ref()
    cfg = scenes.config_c1(64, 36, 3); cfg.nb_bounces = 3
    R.apply_config(cfg); R.add_mesh(scenes.blob_mesh(16)); R.prepare()
    g["pinhole_rgb"] = R.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0]
    np.savez_compressed(os.path.join(OUT, "lenticular.npz"), **g)
    print("lenticular goldens written")


if __name__ == "__main__" and "--lenticular" in sys.argv:
    main_lenticular()


# ---- key-framed transforms (Geometry.h:258-320): scale / translation interpolated linearly, rotation by quaternion slerp
KEYFRAME_FRAMES = (0, 2, 3, 5, 7, 9, 12, 40)


def _rot(axis, angle):
    a = np.asarray(axis, np.float64); a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return (np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K).astype(np.float32).reshape(9)


def keyframe_scene(X, objfile):
    """The OBJ scene with key frames on the mesh (three rotations that take Matrix::toQuaternion through its four branches,
    translations, scales), on the light (translation + scale: centerLight, radiusLight, lightPower follow) and on the
    ground plane (translation: the fog's ground level follows).  X: the compiled reference or the host mirror."""
    cfg = scenes.config_c1(48, 30, 2)
    cfg.nb_bounces = 3
    X.apply_config(cfg)
    oid = X.add_mesh_obj(objfile)
    ident = np.eye(3, dtype=np.float32).reshape(9)
    for frame, (t, r, sc) in ((2, ((0, 0, 0), _rot((0, 1, 0), 0.3), 30.0)), (5, ((4, 1, -3), _rot((1, 0, 0), 3.0), 24.0)),
                              (9, ((-2, 3, 5), _rot((0, 1, 0), 3.1), 33.0)), (12, ((0, 6, 0), _rot((0.2, 0.1, 1), 2.9), 27.0))):
        base, _ = X.object_state(oid)
        X.set_object_transform(oid, np.asarray(t, np.float32) + base[:3] * (frame == 2), r, sc)
        X.add_keyframe(oid, frame)
    for frame, (t, sc) in ((2, ((0, 0, 0), 1.0)), (9, ((5, 2, -4), 1.5))):
        X.set_object_transform(0, t, ident, sc)
        X.add_keyframe(0, frame)
    for frame, ty in ((0, 0.0), (12, -3.0)):
        X.set_object_transform(2, (0, ty, 0), ident, 1.0)
        X.add_keyframe(2, frame)
    return cfg, oid


def main_keyframes():
    import tempfile
    d = tempfile.mkdtemp(prefix="ptref_keyframes_")
    cwd = os.getcwd()
    os.chdir(d)
    try:
        scenes.write_obj_scene(d)
        R = Ref()
        cfg, oid = keyframe_scene(R, "scene.obj")
        R.set_fog(0.3, 0.2, 0.02, 0.03, 1, 0, 0.0)
        R.save_scene("keyframes.scn")
        text = open("keyframes.scn").read()
        R2 = Ref()                               # what a second instance makes of the file: the expected values
        R2.load_scene("keyframes.scn")
        g = {}
        for frame in KEYFRAME_FRAMES:
            R2.set_frame(frame)
            R2.prepare()
            g[f"light_{frame}"] = R2.light()
            for k in range(R2.num_objects()):
                t, inv, r = R2.object_matrices(k)
                g[f"f{frame}_obj{k}_trans"], g[f"f{frame}_obj{k}_inv"], g[f"f{frame}_obj{k}_rot"] = t, inv, r
        for frame in (3, 7):
            R2.set_frame(frame)
            R2.prepare()
            g[f"rgb_{frame}"] = R2.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0]
    finally:
        os.chdir(cwd)
    open(os.path.join(OUT, "keyframes.scn"), "w").write(text)
    np.savez_compressed(os.path.join(OUT, "keyframes.npz"), **g)
    print("keyframes goldens written; mean radiance / white at frame 3:", float(g["rgb_3"].mean() / 196964.7))


if __name__ == "__main__" and "--keyframes" in sys.argv:
    main_keyframes()


# ---- spheres beside the light (0) and the environment (1) (Geometry.h:849-992)
SPHERE_KINDS = ("mixed", "glossy", "nomesh", "bare")


def sphere_scene(X, kind):
    """Spheres as ordinary scene objects, in front of and behind the mesh in the object list: a constant diffuse one, a mirror,
    an image-textured glossy one (lists looked up at the spherical coordinates of the normal), glass; `nomesh`: no TriMesh
    at all, flipped normals.  `bare`: spheres WITHOUT material lists before and behind the mesh and behind a textured sphere —
    Scene::intersection shades such a sphere with the material of the last object before it in the list that the ray also hit
    (its one MaterialValues for all objects of the loop, Geometry.cpp:596), at that object's own hit point."""
    cfg = scenes.config_c1(48, 30, 3)
    cfg.nb_bounces = 5 if kind == "mixed" else 4
    X.apply_config(cfg)
    a = X.add_sphere((-12, -17, 8), 9.0)
    if kind == "bare":
        m = X.add_mesh(scenes.blob_mesh(16, with_uv=True), scale=20.0)
        X.set_group_material(m, 0, (0.7, 0.8, 0.3), (0.2, 0.2, 0.2), (30., 30., 30.))
        X.set_group_texture(m, 0, 0, scenes.checker_texture())
        b = X.add_sphere((14, -14, 2), 8.0)
        c = X.add_sphere((0, -20, 18), 6.0)
        X.add_group_material(c, (0.9, 0.2, 0.1), (0.3, 0.3, 0.3), (20, 20, 20), 1.0, 1.3)
        X.set_group_texture(c, 0, 0, scenes.checker_texture(32, 16, 5, 4))
        X.add_sphere((-20, -18, -6), 7.0, flip_normals=True)
        X.add_sphere((4, -22, 26), 3.0)
        X.prepare()
        return cfg
    X.add_group_material(a, (0.2, 0.6, 0.9), (0, 0, 0), (0, 0, 0), 1.0, 1.3)
    if kind != "nomesh":
        X.add_mesh(scenes.blob_mesh(16), scale=20.0)
    b = X.add_sphere((14, -14, 2), 8.0, mirror=(kind != "glossy"))
    if kind == "glossy":
        X.add_group_material(b, (0.5, 0.5, 0.1), (0.4, 0.4, 0.4), (50, 50, 50), 1.0, 1.3)
    c = X.add_sphere((0, -20, 18), 6.0, flip_normals=(kind == "nomesh"))
    X.add_group_material(c, (0.9, 0.2, 0.1), (0.3, 0.3, 0.3) if kind != "mixed" else (0.0, 0.0, 0.0), (20, 20, 20), 1.0, 1.3)
    X.set_group_texture(c, 0, 0, scenes.checker_texture(32, 16, 5, 4))
    d = X.add_sphere((-20, -18, -6), 7.0)
    X.add_group_material(d, (1, 1, 1), (0, 0, 0), (0, 0, 0), 0.0, 1.5)      # transparent_map < 0.5: glass
    X.prepare()
    return cfg


def main_spheres():
    g = {}
    for kind in SPHERE_KINDS:
        R = Ref()
        cfg = sphere_scene(R, kind)
        g[kind + "_rgb"] = R.getcolor_samples(all_pixels(cfg), 0, cfg.spp)[0]
        print(kind, "mean radiance / white =", float(g[kind + "_rgb"].mean() / 196964.7))
    np.savez_compressed(os.path.join(OUT, "spheres.npz"), **g)


if __name__ == "__main__" and "--spheres" in sys.argv:
    main_spheres()



def main_image_writers():
    """tests/golden/image_writers.npz: the FILES the reference's save_image (utils.cpp:177-234) writes — `.jpg` through its
    vendored encoder at quality 100 for 8-bit pixels, `.hdr` through EncodeRadianceHDR for float pixels — for a few small
    synthetic images (sizes that are not multiples of 8, a single pixel, rows longer than one 127-byte literal chunk,
    zero / tiny / huge float values), plus the `.jpg` the float instantiation writes (its scaling to bytes)."""
    import ctypes as C
    rng = np.random.default_rng(21)
    R = Ref()
    g = {}
    tmpdir = os.path.join(OUT, "_tmp_writers")
    os.makedirs(tmpdir, exist_ok=True)

    def pattern(h, w):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(x / 7.) * np.cos(y / 5.), 127 + 100 * np.cos(x / 3. + y / 11.), (x * 3 + y * 5) % 256], -1)
        return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    sizes = [(37, 53), (8, 8), (1, 1), (16, 24), (9, 130), (64, 3)]
    for n, (h, w) in enumerate(sizes):
        img = np.ascontiguousarray(pattern(h, w))
        if n == 3:
            img[:] = rng.integers(0, 256, img.shape, dtype=np.uint8)           # noise: every coefficient alive, long codes, 0xFF stuffing
        f = os.path.join(tmpdir, f"u8_{n}.jpg")
        R.lib.ref_save_image_u8(f.encode(), img.ctypes.data_as(C.c_void_p), w, h)
        g[f"u8_{n}"] = img
        g[f"jpg_{n}"] = np.frombuffer(open(f, "rb").read(), np.uint8)
    for n, (h, w) in enumerate([(5, 7), (3, 130), (1, 1), (4, 300)]):
        fl = np.exp(rng.normal(0, 4, (h, w, 3))).astype(np.float32)
        fl[rng.random((h, w)) < 0.15] = 0.0
        fl[rng.random((h, w)) < 0.1] *= np.float32(1e-18)
        fl[rng.random((h, w)) < 0.05] *= np.float32(1e12)
        if n == 0:
            fl[0, 0] = (-1.0, 0.5, 0.25)                                        # a negative channel beside positive ones
            fl[0, 1] = (1.0, 1.0, 1.0)
        fl = np.ascontiguousarray(fl)
        f = os.path.join(tmpdir, f"f32_{n}.hdr")
        R.lib.ref_save_image_f32.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        R.lib.ref_save_image_f32(f.encode(), fl.ctypes.data_as(C.c_void_p), w, h, C.c_float(255.0))
        g[f"f32_{n}"] = fl
        g[f"hdr_{n}"] = np.frombuffer(open(f, "rb").read(), np.uint8)
    # save_image<float> into an 8-bit container: val * (255. / maxval), clamped, truncated
    fl = np.ascontiguousarray((rng.random((11, 13, 3)) * 1.3 - 0.1).astype(np.float32))
    f = os.path.join(tmpdir, "f32_scaled.jpg")
    R.lib.ref_save_image_f32(f.encode(), fl.ctypes.data_as(C.c_void_p), 13, 11, C.c_float(1.0))
    g["f32_scaled"] = fl
    g["jpg_f32_scaled"] = np.frombuffer(open(f, "rb").read(), np.uint8)
    import shutil
    shutil.rmtree(tmpdir)
    g["n_u8"] = np.int32(len(sizes)); g["n_f32"] = np.int32(4)
    np.savez_compressed(os.path.join(OUT, "image_writers.npz"), **g)
    print("image writers:", {k: v.size for k, v in g.items() if k.startswith(("jpg", "hdr"))})


if __name__ == "__main__" and "--image-writers" in sys.argv:
    main_image_writers()
