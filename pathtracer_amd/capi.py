"""ctypes view of the product libraries.

* ``libmipt.so``       — the C-ABI of include/mipt.h (HIP kernels).
* ``libmipt_host.so``  — host-side mirror of the reference's Raytracer / Scene / TriMesh surface
                         (pathtracer_amd/host/mipt_host.h), which owns scene construction, the BVH
                         build and prepare_render, and hands POD descriptions to the C-ABI.

``HostRaytracer`` offers the same method names as oracle/binding.py so that one test body can be
run against the reference, the oracle and the HIP path.  Nothing here falls back to a CPU
implementation: without a GPU ``HostRaytracer(device=...)`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBMIPT = os.environ.get("MIPT_LIB_OVERRIDE") or os.path.join(_HERE, "libmipt.so")   # override: tuning builds (tools/)
LIBHOST = os.path.join(_HERE, "libmipt_host.so")

MIPT_OK = 0
MIPT_ERR_NO_DEVICE = 2
MIPT_ERR_CANCELLED = 6
MIPT_ERR_UNSUPPORTED = 4

# every symbol include/mipt.h declares
MIPT_SYMBOLS = ["mipt_create", "mipt_destroy", "mipt_last_error", "mipt_abi_version", "mipt_upload_scene", "mipt_render",
                "mipt_render_device", "mipt_tile_owner", "mipt_measure_stream_read", "mipt_measure_gather_read", "mipt_measure_dependent_gather", "mipt_measure_vmem_issue", "mipt_debug_anyhit_replayed", "mipt_debug_anyhit_kind", "mipt_group_size", "mipt_group_reduce_kind", "mipt_rccl_selftest", "mipt_trace", "mipt_trace_shadow", "mipt_sample_radiance", "mipt_get_stats", "mipt_set_option",
                "mipt_build_bvh", "mipt_build_bvh_error", "mipt_render_denoiser_inputs", "mipt_sample_denoiser_inputs",
                "mipt_device_mesh_build", "mipt_device_mesh_download", "mipt_device_mesh_download_tangents", "mipt_device_mesh_free"]

_f = C.c_float
_i = C.c_int


class MiptStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("mesh_casts_closest", C.c_uint64), ("mesh_casts_shadow", C.c_uint64),
                ("render_ms", C.c_double), ("traverse_ms", C.c_double), ("shadow_ms", C.c_double), ("shade_ms", C.c_double),
                ("traverse_launches", C.c_uint32), ("shadow_launches", C.c_uint32), ("passes", C.c_uint32), ("pipeline", C.c_uint32),
                ("traverse_merged", C.c_uint32), ("reserved", C.c_uint32), ("resolve_ms", C.c_double)]


class MiptTexture(C.Structure):
    _fields_ = [("multiplier", _f * 3), ("W", C.c_int32), ("H", C.c_int32), ("values", C.POINTER(_f))]


class MiptBvhNode(C.Structure):
    """mipt_bvh_node == BVHNodesT<float> (36 bytes)."""
    _fields_ = [("isleaf", C.c_uint8), ("_pad", C.c_uint8 * 3), ("fg", C.c_int32), ("fd", C.c_int32), ("bbox_min", _f * 3), ("bbox_max", _f * 3)]


class MiptMesh(C.Structure):
    _fields_ = [("n_triangles", C.c_int32), ("n_nodes", C.c_int32), ("n_uvs", C.c_int32), ("nodes", C.POINTER(MiptBvhNode)),
                ("bvh_bbox_min", _f * 3), ("bvh_bbox_max", _f * 3), ("triangleSoup", C.c_void_p), ("indices", C.c_void_p),
                ("uvs", C.POINTER(_f)), ("tangentSoup", C.POINTER(_f)), ("device_mesh", C.c_void_p)]


class MiptObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("miroir", C.c_int32), ("ghost", C.c_int32), ("flip_normals", C.c_int32), ("interp_normals", C.c_int32),
                ("trans_matrix", _f * 12), ("inv_trans_matrix", _f * 12), ("rot_matrix", _f * 9),
                ("brdf_kind", C.c_int32), ("merl_data", C.POINTER(C.c_double)),
                ("n_lists", C.c_int32 * 8), ("lists", C.POINTER(MiptTexture) * 8),
                ("O", _f * 3), ("R", _f), ("has_envmap", C.c_int32), ("envW", C.c_int32), ("envH", C.c_int32), ("envtex", C.POINTER(C.c_uint8)),
                ("A", _f * 3), ("vecN", _f * 3), ("mesh", C.POINTER(MiptMesh))]


class MiptSceneDesc(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("objects", C.POINTER(MiptObject)), ("background", C.POINTER(_f)), ("backgroundW", C.c_int32), ("backgroundH", C.c_int32),
                ("fog_density", _f), ("fog_absorption", _f), ("fog_density_decay", _f), ("fog_absorption_decay", _f), ("phase_aniso", _f), ("fog_ground_level", _f),
                ("fog_type", C.c_int32), ("fog_phase_type", C.c_int32)]


class MiptHit(C.Structure):
    _fields_ = [("has_inter", C.c_int32), ("object_id", C.c_int32), ("triangle_id", C.c_int32), ("t", _f), ("P", _f * 3),
                ("shadingN", _f * 3), ("Kd", _f * 3), ("Ks", _f * 3), ("Ne", _f * 3), ("Ke", _f * 3), ("transp", C.c_int32), ("refr_index", _f)]


class MiptRenderParams(C.Structure):
    """mipt_render_params (include/mipt.h)."""
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("nrays", C.c_int32), ("nb_bounces", C.c_int32),
                ("cam_position", _f * 3), ("cam_direction", _f * 3), ("cam_up", _f * 3),
                ("cam_fov", _f), ("cam_focus_distance", _f), ("cam_aperture", _f),
                ("double_frustum_start_t", _f), ("sigma_filter", _f), ("filter_size", C.c_int32),
                ("filter_integral", C.c_void_p), ("samples2d", C.c_void_p), ("randomPerPixel", C.c_void_p),
                ("centerLight", _f * 3), ("radiusLight", _f), ("lightPower", _f), ("envmap_intensity", _f),
                ("seed_stride", C.c_uint64), ("sample_begin", C.c_int32), ("sample_end", C.c_int32),
                ("tile_size", C.c_int32), ("tile_rank", C.c_int32), ("tile_nranks", C.c_int32),
                ("is_lenticular", C.c_int32), ("lenticular_nb_images", C.c_int32), ("lenticular_pixel_width", C.c_int32), ("lenticular_max_angle", _f)]


HIT_DTYPE = np.dtype([("has_inter", np.int32), ("object_id", np.int32), ("triangle_id", np.int32), ("t", np.float32), ("P", np.float32, 3),
                      ("shadingN", np.float32, 3), ("Kd", np.float32, 3), ("Ks", np.float32, 3), ("Ne", np.float32, 3), ("Ke", np.float32, 3),
                      ("transp", np.int32), ("refr_index", np.float32)])
assert HIT_DTYPE.itemsize == C.sizeof(MiptHit)

_libs = None


def load():
    """Loads both libraries and checks every exported symbol; raises if anything is missing."""
    global _libs
    if _libs is not None:
        return _libs
    for p in (LIBMIPT, LIBHOST):
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: run `python __graft_entry__.py` (build()) first; there is no fallback path")
    mipt = C.CDLL(LIBMIPT, mode=C.RTLD_GLOBAL)
    host = C.CDLL(LIBHOST)
    for s in MIPT_SYMBOLS:
        getattr(mipt, s)
    mipt.mipt_last_error.restype = C.c_char_p
    mipt.mipt_last_error.argtypes = [C.c_void_p]
    mipt.mipt_build_bvh_error.restype = C.c_char_p
    mipt.mipt_group_reduce_kind.restype = C.c_char_p
    mipt.mipt_debug_anyhit_kind.restype = C.c_char_p
    mipt.mipt_debug_anyhit_kind.argtypes = [C.c_void_p]
    mipt.mipt_group_reduce_kind.argtypes = [C.c_void_p]
    mipt.mipt_group_size.argtypes = [C.c_void_p]
    mipt.mipt_rccl_selftest.argtypes = [C.c_void_p]
    host.mh_create.restype = C.c_void_p
    host.mh_last_error.restype = C.c_char_p
    for name in ("mh_ctx", "mh_scene_desc", "mh_render_params", "mh_imagedouble", "mh_sample_count", "mh_image"):
        getattr(host, name).restype = C.c_void_p
    _libs = (mipt, host)
    return _libs


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def set_bvh_builder(mode, device=0):
    """Which builder the host mirror's TriMesh::init uses: 'host' (the recursion), 'gpu' (mipt_build_bvh; an error
    without a device) or 'auto' (GPU when a device is present; default)."""
    load()[1].mh_set_bvh_builder({"host": 0, "gpu": 1, "auto": 2}[mode], int(device))


def set_device_resident(on=True):
    """With a GPU builder: TriMesh::init leaves the tree, the Triangle records and the tangents on the device
    (mipt_device_mesh_build; default) or fetches bvh.nodes back and builds triangleSoup / tangentSoup on the host as in round 3."""
    load()[1].mh_set_device_resident(1 if on else 0)


def build_bvh(vertices, tri_vtx, device=0):
    """mipt_build_bvh on raw arrays: (nodes_i [n,3] = isleaf/fg/fd, nodes_bb [n,6], perm [ntri], device seconds)."""
    mipt = load()[0]
    v = np.ascontiguousarray(vertices, np.float32)
    t = np.ascontiguousarray(tri_vtx, np.int32)
    ntri = t.shape[0]
    nodes = np.zeros((2 * ntri, 9), np.uint32)
    perm = np.zeros(ntri, np.int32)
    nn, sec = C.c_int(0), C.c_double(0)
    rc = mipt.mipt_build_bvh(int(device), _p(v, _f), v.shape[0], t.ctypes.data_as(C.c_void_p), 12, ntri,
                             nodes.ctypes.data_as(C.c_void_p), nodes.shape[0], C.byref(nn), _p(perm, _i), C.byref(sec))
    if rc != 0:
        raise MiptError(rc, mipt.mipt_build_bvh_error().decode())
    nodes = nodes[: nn.value]
    nodes_i = np.stack([nodes[:, 0] & 0xff, nodes[:, 1], nodes[:, 2]], 1).astype(np.int32)
    return nodes_i, nodes[:, 3:9].copy().view(np.float32), perm, sec.value


def light_intensity(R, scale=1.0):
    """Scene::intensite_lumiere = 1e9*4pi/(4pi*R*R*pi) (Raytracer.cpp:1270) times the GUI slider factor
    (mainApp.cpp:775), in double, narrowed to the float member."""
    R = float(np.float32(R))
    return float(np.float32(scale * 1000000000 * 4. * np.pi / (4. * np.pi * R * R * np.pi)))


class MiptError(RuntimeError):
    pass


class HostRaytracer:
    """One mipt_host::Raytracer (after loadScene()), optionally bound to a GPU."""

    def __init__(self, device=None):
        self.mipt, self.host = load()
        self.h = C.c_void_p(self.host.mh_create())
        self.W = self.H = self.spp = 0
        self.device = device
        self._uploaded = False
        if device is not None:
            ids = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
            rc = self.host.mh_open_devices(self.h, (C.c_int * len(ids))(*ids), len(ids))     # mipt_create(device_ids, n)
            if rc != MIPT_OK:
                raise MiptError(f"mipt_create(devices={ids}) failed with status {rc}: {self.host.mh_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.host.mh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != MIPT_OK:
            ctx = self.host.mh_ctx(self.h)
            msg = self.mipt.mipt_last_error(ctx).decode() if ctx else self.host.mh_last_error(self.h).decode()
            raise MiptError(f"{what} failed with status {rc}: {msg}")

    # ---- setup (names as oracle/binding.py)
    def set_partition(self, tile_size, rank, nranks):
        self.host.mh_set_partition(self.h, tile_size, rank, nranks)

    def set_render(self, W, H, spp, nb_bounces, sigma_filter=0.5):
        self.W, self.H, self.spp = W, H, spp
        self.host.mh_set_render(self.h, W, H, spp, nb_bounces, _f(sigma_filter))

    def set_camera(self, pos, direction, up, fov, focus, aperture):
        a = lambda v: (_f * 3)(*v)
        self.host.mh_set_camera(self.h, a(pos), a(direction), a(up), _f(fov), _f(focus), _f(aperture))

    def set_light(self, center, R, scale=1.0):
        self.host.mh_set_light(self.h, (_f * 3)(*center), _f(R), _f(light_intensity(R, scale)))

    def set_envmap_intensity(self, v):
        self.host.mh_set_envmap_intensity(self.h, _f(v))

    def apply_config(self, cfg):
        self.set_render(cfg.W, cfg.H, cfg.spp, cfg.nb_bounces, cfg.sigma_filter)
        self.set_camera(cfg.cam_pos, cfg.cam_dir, cfg.cam_up, cfg.fov, cfg.focus, cfg.aperture)
        self.set_light(cfg.light_center, cfg.light_radius, cfg.light_scale)
        self.set_envmap_intensity(cfg.envmap_intensity)

    def set_brdf_merl_file(self, obj, path):
        """objects[obj]->brdf = new IsoMERLBRDF(path): the MERL .binary file is read by the host mirror."""
        self.host.mh_set_brdf_merl_file.restype = C.c_int
        if self.host.mh_set_brdf_merl_file(self.h, obj, str(path).encode()) != 0:
            raise MiptError(self.host.mh_last_error(self.h).decode())

    # ---- .scn scene files (Raytracer::load_scene / save_scene of the host mirror)
    def load_scene(self, path, replaced_names=None):
        """Raytracer::load_scene(filename, replacedNames) (Raytracer.cpp:1149): replaced_names takes the place of the '#' in mesh names."""
        self.host.mh_load_scene_subst.restype = C.c_int
        if self.host.mh_load_scene_subst(self.h, str(path).encode(), None if replaced_names is None else str(replaced_names).encode()) != 0:
            raise MiptError("load_scene(%s): %s" % (path, self.host.mh_last_error(self.h).decode()))
        hdr = self.scene_header()
        self.W, self.H, self.spp = int(hdr[0]), int(hdr[1]), int(hdr[2])

    def add_sphere(self, center, R, mirror=False, flip_normals=False):
        """s.addObject(new Sphere(center, R, mirror, normal_swapped)): a sphere beside the light (0) and the environment (1)."""
        o = np.ascontiguousarray(center, np.float32)
        return self.host.mh_add_sphere(self.h, _p(o, _f), _f(R), int(mirror), int(flip_normals))

    # ---- key-framed transforms (Geometry.h:258-320)
    def set_frame(self, frame):
        self.host.mh_set_frame(self.h, int(frame))

    def add_keyframe(self, obj, frame):
        self.host.mh_add_keyframe(self.h, int(obj), int(frame))

    def set_object_transform(self, obj, translation, rotation9, scale):
        t = np.ascontiguousarray(translation, np.float32); r = np.ascontiguousarray(rotation9, np.float32).reshape(9)
        self.host.mh_set_object_transform(self.h, int(obj), _p(t, _f), _p(r, _f), _f(scale))

    def save_scene(self, path):
        self.host.mh_save_scene.restype = C.c_int
        if self.host.mh_save_scene(self.h, str(path).encode()) != 0:
            raise MiptError("save_scene(%s) failed" % path)

    def num_objects(self):
        return self.host.mh_num_objects(self.h)

    def scene_header(self):
        o = np.zeros(32, np.float32)
        self.host.mh_get_scene_header(self.h, o.ctypes.data_as(C.POINTER(_f)))
        return o

    def object_state(self, obj):
        o = np.zeros(24, np.float32); fl = np.zeros(8, np.int32)
        self.host.mh_get_object_state(self.h, obj, o.ctypes.data_as(C.POINTER(_f)), fl.ctypes.data_as(C.POINTER(C.c_int)))
        return o, fl

    def add_mesh_obj(self, path, scale=30.0, center=True):
        """TriMesh(&scene, path, ...) of the reference: OBJ + MTL (+ PPM textures) read by the host mirror."""
        self.host.mh_add_mesh_obj.restype = C.c_int
        rid = self.host.mh_add_mesh_obj(self.h, str(path).encode(), _f(scale), 1 if center else 0)
        if rid < 0:
            raise MiptError("TriMesh(%s): %s" % (path, self.host.mh_last_error(self.h).decode()))
        return rid

    def group_materials(self, obj):
        """Per material group: multipliers (Kd, Ks, Ne, alpha, refr, transp) and the image sizes of Kd / Ks / normal / alpha."""
        out = []
        for g in range(self.host.mh_num_groups(self.h, obj)):
            m = np.zeros(12, np.float32); wh = np.zeros(8, np.int32)
            self.host.mh_get_group_material(self.h, obj, g, m.ctypes.data_as(C.POINTER(_f)), wh.ctypes.data_as(C.POINTER(C.c_int)))
            out.append((m, wh.reshape(4, 2)))
        return out

    def group_texture(self, obj, grp, slot):
        m, wh = self.group_materials(obj)[grp]
        W, H = int(wh[slot][0]), int(wh[slot][1])
        if W == 0:
            return None
        self.host.mh_group_texture_values.restype = C.POINTER(_f)
        p = self.host.mh_group_texture_values(self.h, obj, grp, slot)
        return np.ctypeslib.as_array(p, shape=(H, W, 3)).copy()

    def add_mesh(self, mesh, scale=30.0, center=True, tmpdir=None):
        v = np.ascontiguousarray(mesh.vertices, np.float32)
        n = np.ascontiguousarray(mesh.normals, np.float32)
        fv = np.ascontiguousarray(mesh.faces_v, np.int32)
        fn = np.ascontiguousarray(mesh.faces_n, np.int32)
        if mesh.uvs is not None:
            uv = np.ascontiguousarray(mesh.uvs, np.float32)
            ft = np.ascontiguousarray(mesh.faces_t, np.int32)
            uvp, ftp, nt = _p(uv, _f), _p(ft, _i), uv.shape[0]
        else:
            uvp, ftp, nt = None, None, 0
        return self.host.mh_add_mesh(self.h, v.shape[0], _p(v, _f), n.shape[0], _p(n, _f), nt, uvp,
                                     fv.shape[0], _p(fv, _i), _p(fn, _i), ftp, _f(scale), 1 if center else 0)

    def set_object_flags(self, obj, miroir=False, flip_normals=False):
        self.host.mh_set_object_flags(self.h, obj, int(miroir), int(flip_normals))

    def set_group_material(self, obj, grp, Kd, Ks, Ne, transp_col=1.0, refr=1.3):
        a = lambda v: (_f * 3)(*v)
        self.host.mh_set_group_material(self.h, obj, grp, a(Kd), a(Ks), a(Ne), _f(transp_col), _f(refr))

    def add_group_material(self, obj, Kd, Ks, Ne, transp_col=1.0, refr=1.3):
        a = lambda v: (_f * 3)(*v)
        self.host.mh_add_group_material(self.h, obj, a(Kd), a(Ks), a(Ne), _f(transp_col), _f(refr))

    def set_group_texture(self, obj, grp, slot, rgb8):
        rgb8 = np.ascontiguousarray(rgb8, np.uint8)
        self.host.mh_set_group_texture(self.h, obj, grp, slot, rgb8.shape[1], rgb8.shape[0], _p(rgb8, C.c_ubyte))

    def set_envmap(self, rgb8):
        rgb8 = np.ascontiguousarray(rgb8, np.uint8)
        self.host.mh_set_envmap(self.h, rgb8.shape[1], rgb8.shape[0], _p(rgb8, C.c_ubyte))

    def set_brdf_merl(self, obj, table):
        table = np.ascontiguousarray(table, np.float64).ravel()
        assert table.size == 3 * 90 * 90 * 180
        self.host.mh_set_brdf_merl(self.h, obj, _p(table, C.c_double))

    def prepare(self):
        """Raytracer::prepare_render; uploads the scene when a device is open."""
        upload = self.device is not None
        self._check(self.host.mh_prepare(self.h, 1 if upload else 0), "prepare_render / mipt_upload_scene")
        self._uploaded = upload

    # ---- host-side dumps
    def light(self):
        out = np.zeros(5, np.float32)
        self.host.mh_get_light(self.h, _p(out, _f))
        return out

    def tables(self):
        rpp = np.zeros((self.H * self.W, 2), np.float32)
        s2d = np.zeros((self.spp, 2), np.float32)
        fi = np.zeros(64 * 64, np.float32)
        fs = _i(0)
        self.host.mh_get_tables(self.h, _p(rpp, _f), _p(s2d, _f), _p(fi, _f), C.byref(fs))
        w = 2 * fs.value + 1
        return rpp, s2d, fi[: w * w].copy(), fs.value

    def object_matrices(self, obj):
        t, inv, r = np.zeros(12, np.float32), np.zeros(12, np.float32), np.zeros(9, np.float32)
        self.host.mh_get_object_matrices(self.h, obj, _p(t, _f), _p(inv, _f), _p(r, _f))
        return t, inv, r

    def mesh_bvh_builder(self, obj):
        """('gpu' | 'host', seconds TriMesh::init spent in build_bvh, device seconds of the GPU build).  'gpu' covers both GPU forms:
        mesh_on_device() says whether the tree and the records stayed on the device (mipt_device_mesh_build)."""
        a, b = C.c_double(0), C.c_double(0)
        who = self.host.mh_mesh_bvh_builder(self.h, obj, C.byref(a), C.byref(b))
        return ("gpu" if who in (1, 2) else "host"), a.value, b.value

    def mesh_on_device(self, obj):
        return self.host.mh_mesh_bvh_builder(self.h, obj, None, None) == 2

    def mesh_dump(self, obj):
        c = [_i(0) for _ in range(5)]
        self.host.mh_mesh_counts(self.h, obj, *[C.byref(x) for x in c])
        ntri, nnodes = c[0].value, c[1].value
        perm = np.zeros(ntri, np.int32)
        nodes_i = np.zeros((nnodes, 3), np.int32)
        nodes_bb = np.zeros((nnodes, 6), np.float32)
        soup = np.zeros((ntri, 31), np.float32)
        groups = np.zeros(ntri, np.int32)
        root = np.zeros(6, np.float32)
        self.host.mh_mesh_dump(self.h, obj, _p(perm, _i), _p(nodes_i, _i), _p(nodes_bb, _f), _p(soup, _f), _p(groups, _i), _p(root, _f))
        return dict(perm=perm, nodes_i=nodes_i, nodes_bb=nodes_bb, soup=soup, groups=groups, root_bb=root,
                    nverts=c[2].value, nnormals=c[3].value, nuvs=c[4].value)

    def mesh_tangents(self, obj):
        """TriMesh::tangentSoup as [3 * ntri, 3] (empty for a mesh without UVs)."""
        n = self.host.mh_mesh_tangents(self.h, obj, None)
        out = np.zeros((n // 3, 3), np.float32)
        if n:
            self.host.mh_mesh_tangents(self.h, obj, _p(out, _f))
        return out

    # ---- the C-ABI, called directly with the descriptions the host side built
    @property
    def ctx(self):
        return C.c_void_p(self.host.mh_ctx(self.h))

    @property
    def scene_desc(self):
        return C.c_void_p(self.host.mh_scene_desc(self.h))

    @property
    def render_params(self):
        return C.c_void_p(self.host.mh_render_params(self.h))

    @property
    def params(self):
        """Live ctypes view of the host mirror's mipt_render_params (valid after prepare())."""
        return C.cast(self.host.mh_render_params(self.h), C.POINTER(MiptRenderParams)).contents

    def _need_device(self):
        if not self._uploaded:
            raise MiptError("no scene uploaded to a GPU (HostRaytracer(device=...) then prepare())")

    def set_option(self, name, value):
        self._check(self.mipt.mipt_set_option(self.ctx, name.encode(), C.c_int64(int(value))), "mipt_set_option")

    def trace(self, rays6):
        self._need_device()
        rays6 = np.ascontiguousarray(rays6, np.float32)
        hits = np.zeros(rays6.shape[0], HIT_DTYPE)
        self._check(self.mipt.mipt_trace(self.ctx, rays6.ctypes.data_as(C.c_void_p), rays6.shape[0], hits.ctypes.data_as(C.c_void_p)), "mipt_trace")
        return hits

    def intersect(self, rays6):
        """Scene::intersection in the (ids, floats) layout of oracle/binding.py."""
        h = self.trace(rays6)
        oi = np.stack([h["has_inter"], h["object_id"], h["triangle_id"]], 1).astype(np.int32)
        of = np.concatenate([h["t"][:, None], h["P"], h["shadingN"], h["Kd"], h["Ks"], h["Ne"], h["Ke"],
                             np.where(h["transp"] != 0, -h["refr_index"], h["refr_index"])[:, None]], 1).astype(np.float32)
        return oi, of

    def intersect_shadow(self, rays6, dist):
        self._need_device()
        rays6 = np.ascontiguousarray(rays6, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        out = np.zeros(rays6.shape[0], np.int32)
        self._check(self.mipt.mipt_trace_shadow(self.ctx, rays6.ctypes.data_as(C.c_void_p), _p(dist, _f), rays6.shape[0], _p(out, C.c_int32)), "mipt_trace_shadow")
        return out

    def sample_radiance(self, ij, k0, k1):
        self._need_device()
        ij = np.ascontiguousarray(ij, np.int32)
        n = ij.shape[0]
        rgb = np.zeros((n, k1 - k0, 3), np.float32)
        dxdy = np.zeros((n, k1 - k0, 2), np.float32)
        self._check(self.mipt.mipt_sample_radiance(self.ctx, self.render_params, _p(ij, C.c_int32), n, k0, k1, _p(rgb, _f), _p(dxdy, _f)), "mipt_sample_radiance")
        return rgb, dxdy

    getcolor_samples = sample_radiance

    def getcolor_samples_aov(self, ij, k0, k1):
        """(rgb, normal, albedo) per sample: getColor's colour and its normalValue / albedoValue outputs (mipt_sample_denoiser_inputs)."""
        self._need_device()
        ij = np.ascontiguousarray(ij, np.int32)
        n = ij.shape[0]
        out = [np.zeros((n, k1 - k0, 3), np.float32) for _ in range(3)]
        self._check(self.mipt.mipt_sample_denoiser_inputs(self.ctx, self.render_params, _p(ij, C.c_int32), n, k0, k1, *[_p(a, _f) for a in out]), "mipt_sample_denoiser_inputs")
        return tuple(out)

    def render_denoiser_inputs(self):
        """mipt_render_denoiser_inputs into zeroed accumulators: (imagedouble, sample_count, albedo sums, normal sums)."""
        self._need_device()
        img, alb, nrm = (np.zeros((self.H, self.W, 3), np.float32) for _ in range(3))
        cnt = np.zeros((self.H, self.W), np.float32)
        self._check(self.mipt.mipt_render_denoiser_inputs(self.ctx, self.render_params, _p(img, _f), _p(cnt, _f), _p(alb, _f), _p(nrm, _f)), "mipt_render_denoiser_inputs")
        return img, cnt, alb, nrm

    def set_fog(self, density, absorption, density_decay=0.0, absorption_decay=0.0, fog_type=0, phase_type=0, phase_aniso=0.0):
        """Scene::fog_* (Geometry.h:1371-1377)."""
        self.host.mh_set_fog(self.h, _f(density), _f(absorption), _f(density_decay), _f(absorption_decay), int(fog_type), int(phase_type), _f(phase_aniso))

    def add_col_subsurface(self, obj, rgb):
        self.host.mh_add_col_subsurface(self.h, obj, (_f * 3)(*rgb))

    def set_group_subsurface(self, obj, grp, rgb):
        self.host.mh_set_group_subsurface(self.h, obj, grp, (_f * 3)(*rgb))

    def set_object_ghost(self, obj, ghost=True):
        self.host.mh_set_object_ghost(self.h, obj, int(ghost))

    def set_background(self, rgb):
        """Scene::background: float [H, W, 3], rows as in memory after load_background, x196964.699 units (None clears it)."""
        if rgb is None:
            self.host.mh_set_background(self.h, None, 0, 0)
            return
        rgb = np.ascontiguousarray(rgb, np.float32)
        self.host.mh_set_background(self.h, _p(rgb, _f), rgb.shape[1], rgb.shape[0])

    def load_background(self, path):
        if self.host.mh_load_background(self.h, os.fsencode(path)) != 0:
            raise MiptError(self.host.mh_last_error(self.h).decode())

    def get_background(self):
        W, H = _i(0), _i(0)
        buf = np.zeros(1 << 22, np.float32)
        n = self.host.mh_get_background(self.h, _p(buf, _f), buf.size, C.byref(W), C.byref(H))
        return buf[:max(n, 0)].reshape(H.value, W.value, 3).copy() if n > 0 else None

    def set_lenticular(self, on, nb_images=10, max_angle=35 * np.pi / 180. * 0.25, pixel_width=1):
        """Camera::is_lenticular & co (Vector.h:720-723, 799-812)."""
        self.host.mh_set_lenticular(self.h, int(on), int(nb_images), _f(max_angle), int(pixel_width))

    def set_has_denoiser(self, on=True):
        self.host.mh_set_has_denoiser(self.h, 1 if on else 0)

    def denoiser_images(self):
        """(albedoImage, normalImage as the reference computes it, shadingNormalImage) after render_image_nopreviz with has_denoiser."""
        self.host.mh_denoiser_image.restype = C.c_void_p
        n = self.W * self.H
        return tuple(np.ctypeslib.as_array(C.cast(self.host.mh_denoiser_image(self.h, k), C.POINTER(_f)), shape=(n * 3,)).reshape(self.H, self.W, 3).copy() for k in range(3))

    def render(self):
        """mipt_render into zeroed host accumulators: (imagedouble[H,W,3], sample_count[H,W])."""
        self._need_device()
        img = np.zeros((self.H, self.W, 3), np.float32)
        cnt = np.zeros((self.H, self.W), np.float32)
        self._check(self.mipt.mipt_render(self.ctx, self.render_params, _p(img, _f), _p(cnt, _f), None, None, None), "mipt_render")
        return img, cnt

    render_seeded = render

    def render_progressive(self, on_pass=None, cancel_after=None):
        """mipt_render with a progress callback (and, optionally, the cancel flag raised after `cancel_after` passes).
        Returns (status, imagedouble, sample_count, [(samples_done, samples_total, sum of sample_count at the call), ...])."""
        self._need_device()
        img = np.zeros((self.H, self.W, 3), np.float32)
        cnt = np.zeros((self.H, self.W), np.float32)
        calls = []
        cancel = C.c_int(0)
        CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)

        def cb(user, done, total):
            calls.append((done, total, float(cnt.sum(dtype=np.float64))))
            if on_pass is not None:
                on_pass(done, total, img, cnt)
            if cancel_after is not None and len(calls) >= cancel_after:
                cancel.value = 1
        rc = self.mipt.mipt_render(self.ctx, self.render_params, _p(img, _f), _p(cnt, _f), CB(cb), None, C.byref(cancel))
        return rc, img, cnt, calls

    def render_cancellable(self):
        """mipt_render with a cancel flag that is never raised and NO progress callback: the library renders the range in
        pass-sized chunks without a host synchronisation between them."""
        self._need_device()
        img = np.zeros((self.H, self.W, 3), np.float32)
        cnt = np.zeros((self.H, self.W), np.float32)
        cancel = C.c_int(0)
        self._check(self.mipt.mipt_render(self.ctx, self.render_params, _p(img, _f), _p(cnt, _f), None, None, C.byref(cancel)), "mipt_render")
        return img, cnt

    def render_device(self, d_accum_ptr, stream=0):
        self._need_device()
        self._check(self.mipt.mipt_render_device(self.ctx, self.render_params, C.c_void_p(d_accum_ptr), C.c_void_p(stream)), "mipt_render_device")

    def measure_stream_read(self, nbytes=8 << 30, repeats=5):
        """Achievable HBM read bandwidth of the device in GB/s (mipt_measure_stream_read)."""
        out = C.c_double(0.0)
        self._check(self.mipt.mipt_measure_stream_read(self.ctx, C.c_uint64(nbytes), int(repeats), C.byref(out)), "mipt_measure_stream_read")
        return out.value

    def group_size(self):
        return self.mipt.mipt_group_size(self.ctx)

    def group_reduce_kind(self):
        return self.mipt.mipt_group_reduce_kind(self.ctx).decode()

    def rccl_selftest(self):
        self._check(self.mipt.mipt_rccl_selftest(self.ctx), "mipt_rccl_selftest")

    def measure_gather_read(self, buffer_bytes=8 << 30, records=1 << 28, repeats=3):
        """GB/s of 64-byte records gathered at random 64-byte-aligned offsets (mipt_measure_gather_read)."""
        out = C.c_double(0.0)
        self._check(self.mipt.mipt_measure_gather_read(self.ctx, C.c_uint64(buffer_bytes), C.c_uint64(records), int(repeats), C.byref(out)), "mipt_measure_gather_read")
        return out.value

    def measure_dependent_gather(self, table_bytes=256 << 20, steps=2000, repeats=2):
        """10^9 dependent random 64-byte fetches per second from a table of that size (mipt_measure_dependent_gather)."""
        out = C.c_double(0.0)
        self._check(self.mipt.mipt_measure_dependent_gather(self.ctx, C.c_uint64(table_bytes), int(steps), int(repeats), C.byref(out)), "mipt_measure_dependent_gather")
        return out.value

    def measure_vmem_issue(self, active_lanes=32, iters=4000):
        """ns of one CU per vector-memory wave-instruction at that many active lanes (mipt_measure_vmem_issue)."""
        out = C.c_double(0.0)
        self._check(self.mipt.mipt_measure_vmem_issue(self.ctx, int(active_lanes), int(iters), C.byref(out)), "mipt_measure_vmem_issue")
        return out.value

    def anyhit_kind(self):
        return self.mipt.mipt_debug_anyhit_kind(self.ctx).decode()

    def anyhit_replayed(self):
        """Shadow rays of the last render that the order-free any-hit kernel left to the ordered one (mipt_debug_anyhit_replayed)."""
        out = C.c_uint64(0)
        self._check(self.mipt.mipt_debug_anyhit_replayed(self.ctx, C.byref(out)), "mipt_debug_anyhit_replayed")
        return out.value

    def stats(self):
        st = MiptStats()
        self._check(self.mipt.mipt_get_stats(self.ctx, C.byref(st)), "mipt_get_stats")
        return {k: getattr(st, k) for k, _ in st._fields_}

    # ---- the reference's entry points, through the host mirror
    def render_image_nopreviz(self):
        self._check(self.host.mh_render_image_nopreviz(self.h), "Raytracer::render_image_nopreviz")
        return self._images()

    def render_image(self, lookahead=None):
        """Raytracer::render_image of the host mirror (one publish per sample).  lookahead: publishes rendered per pass (the mirror's default: 8)."""
        if lookahead is not None:
            self.host.mh_set_progressive_lookahead(self.h, int(lookahead))
        self._check(self.host.mh_render_image(self.h), "Raytracer::render_image")
        return self._images()

    def _images(self):
        n = self.W * self.H
        img = np.ctypeslib.as_array(C.cast(self.host.mh_imagedouble(self.h), C.POINTER(_f)), shape=(n * 3,)).reshape(self.H, self.W, 3).copy()
        cnt = np.ctypeslib.as_array(C.cast(self.host.mh_sample_count(self.h), C.POINTER(_f)), shape=(n,)).reshape(self.H, self.W).copy()
        u8 = np.ctypeslib.as_array(C.cast(self.host.mh_image(self.h), C.POINTER(C.c_ubyte)), shape=(n * 3,)).reshape(self.H, self.W, 3).copy()
        return img, cnt, u8
