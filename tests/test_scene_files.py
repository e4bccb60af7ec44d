""".scn scene files: Raytracer::load_scene / save_scene of the host mirror against the reference's own
(tests/golden/objscene.scn was WRITTEN by the reference's save_scene; objscene_scn.npz is what a second reference
instance made of it after load_scene: header, per-object state, per-sample radiance)."""
import os
import shutil

import numpy as np
import pytest

from helpers import WHITE, assert_bits
from pathtracer_amd import capi, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
SCN = os.path.join(HERE, "golden", "objscene.scn")
GOLD = os.path.join(HERE, "golden", "objscene_scn.npz")


def stage(tmp_path):
    """The files a .scn refers to (mesh, MTL images) next to a copy of it: names inside are relative."""
    scenes.write_obj_scene(str(tmp_path))
    dst = os.path.join(str(tmp_path), "objscene.scn")
    shutil.copy(SCN, dst)
    return dst


def check_state(X, g):
    assert_bits(X.scene_header(), g["header"], "scene header")
    n = X.num_objects()
    assert n == sum(1 for k in g.files if k.endswith("_state"))
    for k in range(n):
        st, fl = X.object_state(k)
        assert_bits(st, g[f"obj{k}_state"], f"object {k} transform / shape")
        assert np.array_equal(fl != 0, g[f"obj{k}_flags"] != 0), f"object {k} flags"


def test_load_scene_matches_reference_golden(tmp_path):
    H = capi.HostRaytracer()
    H.load_scene(stage(tmp_path))
    check_state(H, np.load(GOLD))
    mats = H.group_materials(3)
    assert len(mats) == 3 and mats[0][1][0].tolist() == [32, 16]       # the Kd image of group 0 came through the .scn


def test_save_scene_round_trips_through_the_reference(tmp_path):
    """What the host mirror saves, the reference loads to the same scene (and the other way round)."""
    from oracle import binding
    if not binding.ref_available():
        pytest.skip("compiled reference not present")
    scn = stage(tmp_path)
    H = capi.HostRaytracer()
    H.load_scene(scn)
    mine = os.path.join(str(tmp_path), "host.scn")
    H.save_scene(mine)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))                 # the reference resolves the relative names against the working directory
    try:
        R = binding.Ref()
        R.load_scene("host.scn")
        g = np.load(GOLD)
        assert_bits(R.scene_header(), g["header"], "header after host save -> reference load")
        for k in range(R.num_objects()):
            st, fl = R.object_state(k)
            assert_bits(st, g[f"obj{k}_state"], f"object {k}")
        a, b = R.mesh_dump(3), H.mesh_dump(3)
        for key in ("perm", "nodes_i", "nodes_bb", "groups"):
            assert_bits(a[key], b[key], "mesh." + key)
        for (ma, wa), (mb, wb) in zip(R.group_materials(3), H.group_materials(3)):
            assert_bits(ma, mb, "material multipliers")
            assert np.array_equal(wa, wb)
    finally:
        os.chdir(cwd)


def test_unsupported_scene_features_are_refused(tmp_path):
    scn = stage(tmp_path)
    text = open(scn).read()
    H = capi.HostRaytracer()
    for old, new, what in (("nb_transforms: 0", "nb_transforms: 2", "key frame"),          # two key frames announced, none given
                           ("has_csv: 0", "has_csv: 1", "per-face colour"), ("NEW MESH", "NEW POINTSET", "outside the hot path")):
        bad = os.path.join(str(tmp_path), "bad.scn")
        open(bad, "w").write(text.replace(old, new, 1))
        with pytest.raises(capi.MiptError, match=what):
            H.load_scene(bad)
    with pytest.raises(capi.MiptError):
        H.load_scene(os.path.join(str(tmp_path), "missing.scn"))


def test_load_scene_name_substitution(tmp_path):
    """Raytracer::load_scene(filename, replacedNames) (Raytracer.cpp:1149, 1214; Geometry.h:524-526; the third argument
    of the reference's command line, mainApp.cpp:41-42): the '#' in a mesh name is replaced before the mesh file is read."""
    scn = stage(tmp_path)
    text = open(scn).read()
    mesh_line = [l for l in text.splitlines() if l.startswith("name:") and l.strip().endswith(".obj")]
    assert len(mesh_line) == 1
    fname = mesh_line[0].split(":", 1)[1].strip()
    stem = fname[:-4]
    templ = os.path.join(str(tmp_path), "templ.scn")
    open(templ, "w").write(text.replace(mesh_line[0], "name: " + stem[:-3] + "#" + stem[-1:] + ".obj", 1))
    H = capi.HostRaytracer()
    H.load_scene(templ, stem[-3:-1])
    check_state(H, np.load(GOLD))
    a = H.mesh_dump(3)
    H2 = capi.HostRaytracer()
    H2.load_scene(scn)
    for key in ("perm", "nodes_i", "nodes_bb", "groups"):
        assert_bits(a[key], H2.mesh_dump(3)[key], "mesh." + key)
    with pytest.raises(capi.MiptError):                 # wrong substitution: the file does not exist
        H.load_scene(templ, "zz")
    with pytest.raises(capi.MiptError, match="#"):      # no '#' to substitute: the reference's std::string::replace throws
        H.load_scene(scn, "x")


@pytest.mark.gpu
def test_scene_file_radiance_and_cli(tmp_path):
    """scene.scn -> host mirror -> C ABI -> HIP path: per-sample radiance of the reference, bit for bit; and the
    command-line front end renders the same scene file to the same bytes as the Python-driven mirror."""
    import subprocess
    g = np.load(GOLD)
    scn = stage(tmp_path)
    H = capi.HostRaytracer(device=0)
    H.load_scene(scn)
    H.prepare()
    pix = np.stack(np.meshgrid(np.arange(H.H), np.arange(H.W), indexing="ij"), -1).reshape(-1, 2).astype(np.int32)
    rgb, dxdy = H.sample_radiance(pix, 0, H.spp)
    assert_bits(dxdy, g["sample_dxdy"], "jitter")
    assert np.abs(rgb.astype(np.float64) - g["sample_rgb"]).max() / WHITE < 1e-4
    assert_bits(rgb, g["sample_rgb"], "per-sample radiance of the loaded scene")
    out = tmp_path / "out.ppm"
    exe = os.path.join(os.path.dirname(capi.LIBHOST), "mipt_render")
    r = subprocess.run([exe, scn, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    img, cnt, u8 = H.render_image_nopreviz()
    header = b"P6\n%d %d\n255\n" % (H.W, H.H)
    data = out.read_bytes()
    assert data.startswith(header)
    assert np.array_equal(np.frombuffer(data[len(header):], np.uint8).reshape(H.H, H.W, 3), u8)
    # the reference's command line: `scene.scn out.png [nameSubst]` writes a PNG (save_image dispatches on the extension)
    from test_image_readers import host_read
    png = tmp_path / "out.png"
    r = subprocess.run([exe, scn, str(png)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    assert np.array_equal(host_read(png), u8)
    # .jpg (baseline, quality 100: the reference's encoder byte for byte, tests/test_image_writers.py) and .hdr are written too
    jpg = tmp_path / "out.jpg"
    r = subprocess.run([exe, scn, str(jpg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dec = host_read(jpg)
    assert dec.shape == u8.shape and np.abs(dec.astype(int) - u8.astype(int)).max() <= 6
    hdr = tmp_path / "out.hdr"
    r = subprocess.run([exe, scn, str(hdr)], capture_output=True, text=True)
    assert r.returncode == 0 and hdr.read_bytes().startswith(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (H.H, H.W)), r.stderr
    # a name no writer exists for is refused before anything is rendered, and an existing file of that name is left alone
    gif = tmp_path / "out.gif"
    gif.write_bytes(b"keep me")
    r = subprocess.run([exe, scn, str(gif)], capture_output=True, text=True)
    assert r.returncode != 0 and "no writer" in r.stderr and gif.read_bytes() == b"keep me"
