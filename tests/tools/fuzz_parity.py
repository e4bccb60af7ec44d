"""Differential fuzzing of the HIP path against the oracle: random scenes (tessellation, one or two meshes, scales,
camera, light, aperture, depth, materials incl. glossy / mirror / dielectric / textures / MERL), per-sample radiance
compared bit for bit on both pipelines.  usage: python tests/tools/fuzz_parity.py [n_scenes] [seed] [--queue]   (test infrastructure: the oracle is the checker)
--queue: every scene also draws from the features of the contribution-queue kernel (ghost objects, background photo, fog in
both media with the three phase functions, subsurface colours), alone and combined.
--spheres: every scene also holds 1-3 random spheres (constant, glossy, textured, mirror, glass) before / after the mesh.
--bare-spheres: the same, and a sphere may have no material lists at all (such scenes run on the one-thread-per-sample kernel).
--kind=<diffuse|glossy|mirror|glass|textured|merl|two|fat>: every scene's mesh gets this material kind (default: drawn per scene).
--many: 30-90 spheres and 3-8 more meshes per scene (fat leaves among them).
--merl-tiers: pipeline 1 is checked with the three forms of the measured-BRDF tier (`merl_batch` 1, 0 and 2)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import bits_equal, WHITE
from pathtracer_amd import capi, scenes
from oracle.binding import Oracle

QUEUE = "--queue" in sys.argv
SPHERES = "--spheres" in sys.argv or "--bare-spheres" in sys.argv
BARE = "--bare-spheres" in sys.argv      # spheres may also come WITHOUT material lists (kind 5: shaded with the material of the object tested before, Geometry.cpp:596)
KIND = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--kind=")), None)
MERL_TIERS = "--merl-tiers" in sys.argv
MANY = "--many" in sys.argv          # round 6: 30-90 spheres and 3-8 more meshes (one of them sometimes with leaves of 40-100 triangles): object tables far beyond 31 entries, fat leaves
argv = [a for a in sys.argv if not a.startswith("--")]
n_scenes = int(argv[1]) if len(argv) > 1 else 20
rng = np.random.default_rng(int(argv[2]) if len(argv) > 2 else 1)
bad = 0
for it in range(n_scenes):
    W, H, spp = int(rng.integers(8, 97)), int(rng.integers(8, 65)), int(rng.integers(1, 5))
    cfg = scenes.config_c1(W, H, spp)
    cfg.nb_bounces = int(rng.integers(1, 9))
    cfg.aperture = float(rng.choice([0.0, 0.1, 0.5, 2.0]))
    cfg.light_center = tuple(float(x) for x in rng.uniform(-30, 40, 3))
    cfg.light_radius = float(rng.uniform(2, 15))
    if rng.random() < 0.5:
        ang = rng.uniform(-0.6, 0.2)
        cfg.cam_pos = (float(rng.uniform(-20, 20)), float(rng.uniform(-10, 20)), float(rng.uniform(30, 70)))
        cfg.cam_dir = (0.0, float(np.sin(ang)), float(-np.cos(ang)))
        cfg.cam_up = (0.0, float(np.cos(ang)), float(np.sin(ang)))
    kind = rng.choice(["diffuse", "glossy", "mirror", "glass", "textured", "merl", "two", "fat"])
    if KIND: kind = KIND
    n = int(rng.integers(6, 70))
    mesh = scenes.blob_mesh(n, fine_detail=bool(rng.integers(0, 2)), with_uv=(kind == "textured"))
    if kind == "fat":
        mesh = scenes.fat_leaf_mesh(int(rng.integers(8, 30)))
    scale = float(rng.choice([30.0, 5.0, 0.5, 80.0]))
    out = []
    sph = []
    if SPHERES or MANY:
        for k in range(int(rng.integers(30, 90)) if MANY else int(rng.integers(1, 4))):
            sph.append(dict(c=tuple(float(v) for v in rng.uniform((-35, -25, -25), (35, 25, 30))), r=float(rng.uniform(1, 12)), kind=int(rng.integers(0, 6 if BARE else 5)),
                            first=bool(rng.random() < 0.3), flip=bool(rng.random() < 0.15), Kd=rng.uniform(0.05, 1, 3), Ks=rng.uniform(0, 0.6, 3), Ne=rng.uniform(1, 200, 3)))
    sph_lists = True      # (rounds 1-3: subsurface colours only beside spheres WITH material lists; since round 4 a sphere without — mirror or not — inherits Ksub on the one-thread kernel)
    for q in sph:         # round 4: a sphere with lists may carry a subsurface colour of its own (Sphere::reservoir_sampling_intersection)
        q["ksub"] = tuple(float(v) for v in rng.uniform(0.05, 0.9, 3)) if (QUEUE and q["kind"] in (0, 2, 3) and rng.random() < 0.35) else None
    extra_meshes = []
    if MANY:
        for k in range(int(rng.integers(3, 9))):
            em = scenes.huge_leaf_mesh(int(rng.integers(6, 12)), int(rng.integers(40, 100)), int(rng.integers(1, 3))) if rng.random() < 0.25 else scenes.blob_mesh(int(rng.integers(4, 14)))
            extra_meshes.append((em, float(rng.uniform(2, 9)), (float(rng.uniform(-30, 30)), 0.0, float(rng.uniform(-20, 25)))))
    def put_spheres(X, first):
        for q in sph:
            if q["first"] != first: continue
            o = X.add_sphere(q["c"], q["r"], mirror=(q["kind"] == 1), flip_normals=q["flip"])
            if q["kind"] == 0: X.add_group_material(o, q["Kd"], (0, 0, 0), (0, 0, 0), 1.0, 1.3)
            if q["kind"] == 2: X.add_group_material(o, q["Kd"], q["Ks"], q["Ne"], 1.0, 1.3)
            if q["kind"] == 3:
                X.add_group_material(o, (1, 1, 1), q["Ks"] * 0.5, q["Ne"], 1.0, 1.3)
                X.set_group_texture(o, 0, 0, scenes.checker_texture(32, 16, 5, 4))
            if q["kind"] == 4: X.add_group_material(o, (1, 1, 1), (0, 0, 0), (0, 0, 0), 0.0, float(1.1 + q["Ks"][0]))
            if q["ksub"]: X.add_col_subsurface(o, q["ksub"])
    for X in (Oracle(), capi.HostRaytracer(device=0)):
        X.apply_config(cfg)
        put_spheres(X, True)
        oid = X.add_mesh(mesh, scale=scale)
        put_spheres(X, False)
        if MANY:
            for (em, es, ec) in extra_meshes:
                v = em.vertices.astype(np.float64) * es + np.array([ec[2], 0.0, -ec[0]])      # placed through the vertices (world (x, z) = (-z_in, x_in)); add_mesh rests it on the ground
                X.add_mesh(scenes.MeshData(v.astype(np.float32), em.normals, em.uvs, em.faces_v, em.faces_n, em.faces_t, em.name), scale=1.0, center=False)
        out.append((X, oid))
    # materials must be identical on both sides: draw once, apply twice
    Kd, Ks, Ne = rng.uniform(0, 1, 3), rng.uniform(0, 0.9, 3), rng.uniform(0, 300, 3)
    for X, oid in out:
        if kind == "glossy":
            X.set_group_material(oid, 0, Kd, Ks, Ne)
            X.add_group_material(2, Kd[::-1], Ks * 0.5, Ne)
        elif kind == "mirror":
            X.set_object_flags(oid, True, False)
        elif kind == "glass":
            X.set_group_material(oid, 0, Kd, (0, 0, 0), (0, 0, 0), 0.0, float(1.0 + Ks[0]))
        elif kind == "textured":
            X.set_group_material(oid, 0, (1, 1, 1), Ks * 0.4, Ne * 0.2)
            X.set_group_texture(oid, 0, 0, scenes.checker_texture(32, 16, 5, 4))
            X.set_envmap(scenes.sky_envmap(64, 32))
        elif kind == "merl":
            X.set_brdf_merl(oid, scenes.synthetic_merl_table())
        elif kind == "two":
            X.add_mesh(scenes.blob_mesh(12), scale=scale * 0.4)
    feats = []
    if QUEUE:
        photo = (rng.uniform(0, 1, (int(rng.integers(2, 40)), int(rng.integers(2, 40)), 3)) ** 2.2 * 196964.699).astype(np.float32)
        fog = (float(rng.uniform(0.05, 3.0)), float(rng.uniform(0.05, 2.0)), float(rng.uniform(0, 0.08)), float(rng.uniform(0, 0.08)),
               int(rng.integers(0, 2)), int(rng.integers(0, 3)), float(rng.uniform(-0.8, 0.8)))
        ksub = tuple(float(v) for v in rng.uniform(0.05, 0.9, 3))
        pick = rng.random(5)
        for X, oid in out:
            if pick[0] < 0.45: X.set_object_ghost(2, True)
            if pick[1] < 0.25: X.set_object_ghost(oid, True)
            if pick[2] < 0.6: X.set_background(photo)
            if pick[3] < 0.5: X.set_fog(*fog)
            if pick[4] < 0.4 and kind not in ("two",) and sph_lists: X.set_group_subsurface(oid, 0, ksub)
        feats = [n for n, on in (("ghostfloor", pick[0] < 0.45), ("ghostmesh", pick[1] < 0.25), ("photo", pick[2] < 0.6), ("fog%d/%d" % (fog[4], fog[5]), pick[3] < 0.5), ("sss", pick[4] < 0.4 and kind != "two" and sph_lists)) if on]
    for X, oid in out:
        X.prepare()
    O, G = out[0][0], out[1][0]
    pix = np.stack(np.meshgrid(np.arange(H), np.arange(W), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)
    want = O.getcolor_samples(pix, 0, spp)[0]
    line = "%2d %-8s n=%-3d scale %-5g %3dx%-3d spp %d depth %d aperture %-4g" % (it, kind, n, scale, W, H, spp, cfg.nb_bounces, cfg.aperture)
    line += " " + "+".join(feats) + (" spheres " + "".join("cmgtdb"[q["kind"]] + ("s" if q["ksub"] else "") for q in sph) if sph else "")
    for pipeline in ((1,) if (feats or any(q["ksub"] for q in sph)) else (1, 0)):
        G.set_option("pipeline", pipeline)
        for batch in ((1, 0, 2) if (MERL_TIERS and pipeline == 1) else (None,)):
            if batch is not None: G.set_option("merl_batch", batch)
            got = G.getcolor_samples(pix, 0, spp)[0]
            same = bits_equal(got, want).all(-1).mean()
            err = np.abs(got.astype(np.float64) - want).max() / WHITE
            line += "  p%d%s: identical %.6f max|err|/white %.2g" % (pipeline, "" if batch is None else "/batch%d" % batch, same, err)
            if same < 1.0:
                bad += 1
    print(line, flush=True)
print("scenes with any differing sample:", bad)
