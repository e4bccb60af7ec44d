"""TEST INFRASTRUCTURE ONLY — ctypes bindings of

* oracle/_ref/libptref.so  (class ``Ref``):    the compiled reference (oracle/ref_harness.cpp);
* oracle/_ref/libptref_mipt.so (class ``RefMipt``): the same reference compiled with the USE_MIPT switch of
  integration/use_mipt, i.e. with pathtracer_amd/libmipt.so under its own Raytracer / Scene / TriMesh members;
* oracle/libptoracle.so    (class ``Oracle``): our plain-C restatement (oracle/pt_oracle.c).

Both expose the same methods so a test can run one comparison against either.  Imported only
from tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Product code (pathtracer_amd/) never imports this module."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libptref.so")
REF_MIPT_LIB_PATH = os.path.join(_HERE, "_ref", "libptref_mipt.so")
ORACLE_LIB_PATH = os.path.join(_HERE, "libptoracle.so")


def ref_available() -> bool:
    return os.path.exists(REF_LIB_PATH)


def ref_mipt_available() -> bool:
    return os.path.exists(REF_MIPT_LIB_PATH)


def oracle_available() -> bool:
    return os.path.exists(ORACLE_LIB_PATH)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


_f = C.c_float
_i = C.c_int


def light_intensity(R, scale=1.0):
    """Scene::intensite_lumiere: 1e9*4pi/(4pi*R*R*pi) (Raytracer.cpp:1270) times the GUI slider
    factor (mainApp.cpp:775), evaluated in double and narrowed to the float member."""
    R = float(np.float32(R))
    return float(np.float32(scale * 1000000000 * 4. * np.pi / (4. * np.pi * R * R * np.pi)))


class _Prefixed:
    """Resolves ``self.lib.ref_xxx`` to ``<prefix>xxx`` of the loaded library."""

    def __init__(self, cdll, prefix):
        self._cdll, self._prefix = cdll, prefix

    def __getattr__(self, name):
        assert name.startswith("ref_")
        return getattr(self._cdll, self._prefix + name[4:])


class _Base:
    """One `Raytracer` instance (after loadScene())."""
    PREFIX = "ref_"
    PATH = REF_LIB_PATH

    def __init__(self):
        self.cdll = C.CDLL(self.PATH)
        self.lib = _Prefixed(self.cdll, self.PREFIX)
        L = self.lib
        L.ref_create.restype = C.c_void_p
        self.ctx = C.c_void_p(L.ref_create())
        self.W = self.H = self.spp = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if self.ctx:
            self.lib.ref_destroy(self.ctx)
            self.ctx = None

    # ---- setup
    def set_render(self, W, H, spp, nb_bounces, sigma_filter=0.5):
        self.W, self.H, self.spp = W, H, spp
        self.lib.ref_set_render(self.ctx, W, H, spp, nb_bounces, _f(sigma_filter))

    def set_camera(self, pos, direction, up, fov, focus, aperture):
        a = lambda v: (_f * 3)(*v)
        self.lib.ref_set_camera(self.ctx, a(pos), a(direction), a(up), _f(fov), _f(focus), _f(aperture))

    def get_camera(self):
        out = np.zeros(12, np.float32)
        self.lib.ref_get_camera(self.ctx, _p(out, _f))
        return out

    def set_light(self, center, R, scale=1.0):
        self.lib.ref_set_light(self.ctx, (_f * 3)(*center), _f(R), _f(light_intensity(R, scale)))

    def set_envmap_intensity(self, v):
        self.lib.ref_set_envmap_intensity(self.ctx, _f(v))

    def add_sphere(self, center, R, mirror=False, flip_normals=False):
        """s.addObject(new Sphere(center, R, mirror, normal_swapped)): a sphere beside the light (0) and the environment (1)."""
        o = np.ascontiguousarray(center, np.float32)
        return self.lib.ref_add_sphere(self.ctx, _p(o, _f), _f(R), int(mirror), int(flip_normals))

    def set_object_flags(self, obj, miroir=False, flip_normals=False):
        self.lib.ref_set_object_flags(self.ctx, obj, int(miroir), int(flip_normals))

    def set_fog(self, density, absorption, density_decay=0.0, absorption_decay=0.0, fog_type=0, phase_type=0, phase_aniso=0.0):
        """Scene::fog_* (Geometry.h:1371-1377): fog_type 0 uniform / 1 exponential in height, phase 0 isotropic / 1 Schlick / 2 Rayleigh."""
        self.lib.ref_set_fog(self.ctx, _f(density), _f(absorption), _f(density_decay), _f(absorption_decay), int(fog_type), int(phase_type), _f(phase_aniso))

    def add_col_subsurface(self, obj, rgb):
        """Object::add_col_subsurface: appends a constant subsurface colour (planes have no material lists by default)."""
        self.lib.ref_add_col_subsurface(self.ctx, obj, (_f * 3)(*rgb))

    def set_group_subsurface(self, obj, grp, rgb):
        """Object::subsurface[grp]: a constant subsurface colour Ksub (non-zero switches the subsurface branch on)."""
        self.lib.ref_set_group_subsurface(self.ctx, obj, grp, (_f * 3)(*rgb))

    def set_lenticular(self, on, nb_images=10, max_angle=35 * np.pi / 180. * 0.25, pixel_width=1):
        """Camera::is_lenticular & co (Vector.h:720-723, 799-812)."""
        self.lib.ref_set_lenticular(self.ctx, int(on), int(nb_images), _f(max_angle), int(pixel_width))

    def set_object_ghost(self, obj, ghost=True):
        self.lib.ref_set_object_ghost(self.ctx, obj, int(ghost))

    def set_background(self, rgb):
        """Scene::background: float [H, W, 3], rows as in the file, in the renderer's x196964.699 units (None clears it)."""
        if rgb is None:
            self.lib.ref_set_background(self.ctx, None, 0, 0)
            return
        rgb = np.ascontiguousarray(rgb, np.float32)
        self.lib.ref_set_background(self.ctx, _p(rgb, _f), rgb.shape[1], rgb.shape[0])

    def load_background(self, path):
        self.lib.ref_load_background(self.ctx, os.fsencode(path))

    def get_background(self):
        W, H = _i(0), _i(0)
        buf = np.zeros(1 << 22, np.float32)
        n = self.lib.ref_get_background(self.ctx, _p(buf, _f), buf.size, C.byref(W), C.byref(H))
        return buf[:max(n, 0)].reshape(H.value, W.value, 3).copy() if n > 0 else None

    def set_group_material(self, obj, grp, Kd, Ks, Ne, transp_col=1.0, refr=1.3):
        a = lambda v: (_f * 3)(*v)
        self.lib.ref_set_group_material(self.ctx, obj, grp, a(Kd), a(Ks), a(Ne), _f(transp_col), _f(refr))

    def add_group_material(self, obj, Kd, Ks, Ne, transp_col=1.0, refr=1.3):
        a = lambda v: (_f * 3)(*v)
        self.lib.ref_add_group_material(self.ctx, obj, a(Kd), a(Ks), a(Ne), _f(transp_col), _f(refr))

    def set_group_texture(self, obj, grp, slot, rgb8):
        """Image texture (H,W,3 uint8, top row first) for one material group; slot 0 Kd, 1 Ks, 2 normal map,
        3 alpha, 4 Ne."""
        rgb8 = np.ascontiguousarray(rgb8, np.uint8)
        self._set_group_texture(obj, grp, slot, rgb8)

    def _set_group_texture(self, obj, grp, slot, rgb8):
        self.lib.ref_set_group_texture(self.ctx, obj, grp, slot, rgb8.shape[1], rgb8.shape[0], _p(rgb8, C.c_ubyte))

    def set_envmap(self, rgb8):
        rgb8 = np.ascontiguousarray(rgb8, np.uint8)
        self.lib.ref_set_envmap(self.ctx, rgb8.shape[1], rgb8.shape[0], _p(rgb8, C.c_ubyte))

    def set_brdf_merl(self, obj, table):
        """IsoMERLBRDF on one object; table = float64 array of 3*90*90*180 values (MERL .binary payload)."""
        table = np.ascontiguousarray(table, np.float64).ravel()
        assert table.size == 3 * 90 * 90 * 180
        self._set_brdf_merl(obj, table)

    def _set_brdf_merl(self, obj, table):
        self.lib.ref_set_brdf_merl(self.ctx, obj, _p(table, C.c_double))

    def apply_config(self, cfg):
        self.set_render(cfg.W, cfg.H, cfg.spp, cfg.nb_bounces, cfg.sigma_filter)
        self.set_camera(cfg.cam_pos, cfg.cam_dir, cfg.cam_up, cfg.fov, cfg.focus, cfg.aperture)
        self.set_light(cfg.light_center, cfg.light_radius, cfg.light_scale)
        self.set_envmap_intensity(cfg.envmap_intensity)

    def prepare(self):
        self.lib.ref_prepare(self.ctx)

    # ---- dumps
    def light(self):
        out = np.zeros(5, np.float32)
        self.lib.ref_get_light(self.ctx, _p(out, _f))
        return out

    def tables(self):
        rpp = np.zeros((self.H * self.W, 2), np.float32)
        s2d = np.zeros((self.spp, 2), np.float32)
        fi = np.zeros(64 * 64, np.float32)
        fs = _i(0)
        self.lib.ref_get_tables(self.ctx, _p(rpp, _f), _p(s2d, _f), _p(fi, _f), C.byref(fs))
        w = 2 * fs.value + 1
        return rpp, s2d, fi[: w * w].copy(), fs.value

    def object_matrices(self, obj):
        t, inv, r = np.zeros(12, np.float32), np.zeros(12, np.float32), np.zeros(9, np.float32)
        self.lib.ref_get_object_matrices(self.ctx, obj, _p(t, _f), _p(inv, _f), _p(r, _f))
        return t, inv, r

    def mesh_dump(self, obj):
        c = [_i(0) for _ in range(5)]
        self.lib.ref_mesh_counts(self.ctx, obj, *[C.byref(x) for x in c])
        ntri, nnodes = c[0].value, c[1].value
        perm = np.zeros(ntri, np.int32)
        nodes_i = np.zeros((nnodes, 3), np.int32)
        nodes_bb = np.zeros((nnodes, 6), np.float32)
        soup = np.zeros((ntri, 31), np.float32)
        groups = np.zeros(ntri, np.int32)
        root = np.zeros(6, np.float32)
        self.lib.ref_mesh_dump(self.ctx, obj, _p(perm, _i), _p(nodes_i, _i), _p(nodes_bb, _f), _p(soup, _f),
                               _p(groups, _i), _p(root, _f))
        return dict(perm=perm, nodes_i=nodes_i, nodes_bb=nodes_bb, soup=soup, groups=groups, root_bb=root,
                    nverts=c[2].value, nnormals=c[3].value, nuvs=c[4].value)

    # ---- leaf functions
    def pcg32(self, seed, n):
        out = np.zeros(n, np.uint32)
        self.lib.ref_pcg32(C.c_uint64(seed), n, _p(out, C.c_uint32))
        return out

    def lattice(self, n):
        out = np.zeros((n, 2), np.float32)
        self.lib.ref_lattice(n, _p(out, _f))
        return out

    def invsqroot(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        self.lib.ref_invsqroot(x.size, _p(x, _f), _p(out, _f))
        return out

    def fast_normalize(self, v):
        v = np.ascontiguousarray(v, np.float32)
        out = np.zeros_like(v)
        self.lib.ref_fast_normalize(v.shape[0], _p(v, _f), _p(out, _f))
        return out

    def fast_exp(self, x):
        x = np.ascontiguousarray(x, np.float64)
        out = np.zeros_like(x)
        self.lib.ref_fast_exp(x.size, _p(x, C.c_double), _p(out, C.c_double))
        return out

    def random_cos(self, N, r12):
        N = np.ascontiguousarray(N, np.float32)
        r12 = np.ascontiguousarray(r12, np.float32)
        out = np.zeros_like(N)
        self.lib.ref_random_cos(N.shape[0], _p(N, _f), _p(r12, _f), _p(out, _f))
        return out

    def camera_rays(self, ij, jit4):
        ij = np.ascontiguousarray(ij, np.int32)
        jit4 = np.ascontiguousarray(jit4, np.float32)
        out = np.zeros((ij.shape[0], 6), np.float32)
        self.lib.ref_camera_rays(self.ctx, ij.shape[0], _p(ij, _i), _p(jit4, _f), _p(out, _f))
        return out

    def intersect(self, rays6):
        rays6 = np.ascontiguousarray(rays6, np.float32)
        n = rays6.shape[0]
        oi = np.zeros((n, 3), np.int32)
        of = np.zeros((n, 20), np.float32)
        self.lib.ref_intersect(self.ctx, n, _p(rays6, _f), _p(oi, _i), _p(of, _f))
        return oi, of

    def intersect_shadow(self, rays6, dist):
        rays6 = np.ascontiguousarray(rays6, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        out = np.zeros(rays6.shape[0], np.int32)
        self.lib.ref_intersect_shadow(self.ctx, rays6.shape[0], _p(rays6, _f), _p(dist, _f), _p(out, _i))
        return out

    def phong_sample(self, mat9, wo, N, r12, seeds):
        mat9 = np.ascontiguousarray(mat9, np.float32); wo = np.ascontiguousarray(wo, np.float32)
        N = np.ascontiguousarray(N, np.float32); r12 = np.ascontiguousarray(r12, np.float32)
        seeds = np.ascontiguousarray(seeds, np.uint64)
        out = np.zeros((mat9.shape[0], 5), np.float32)
        self.lib.ref_phong_sample(mat9.shape[0], _p(mat9, _f), _p(wo, _f), _p(N, _f), _p(r12, _f),
                                  _p(seeds, C.c_uint64), _p(out, _f))
        return out

    def phong_eval(self, mat9, wi, wo, N):
        mat9 = np.ascontiguousarray(mat9, np.float32); wi = np.ascontiguousarray(wi, np.float32)
        wo = np.ascontiguousarray(wo, np.float32); N = np.ascontiguousarray(N, np.float32)
        out = np.zeros((mat9.shape[0], 3), np.float32)
        self.lib.ref_phong_eval(mat9.shape[0], _p(mat9, _f), _p(wi, _f), _p(wo, _f), _p(N, _f), _p(out, _f))
        return out

    # ---- radiance
    def getcolor_samples(self, ij, k0, k1):
        ij = np.ascontiguousarray(ij, np.int32)
        n = ij.shape[0]
        rgb = np.zeros((n, k1 - k0, 3), np.float32)
        dxdy = np.zeros((n, k1 - k0, 2), np.float32)
        self.lib.ref_getcolor_samples(self.ctx, n, _p(ij, _i), k0, k1, _p(rgb, _f), _p(dxdy, _f))
        return rgb, dxdy

    def getcolor_samples_aov(self, ij, k0, k1):
        """(rgb, normal, albedo) per sample: getColor's colour and its normalValue / albedoValue outputs."""
        ij = np.ascontiguousarray(ij, np.int32)
        n = ij.shape[0]
        out = [np.zeros((n, k1 - k0, 3), np.float32) for _ in range(3)]
        self.lib.ref_getcolor_samples_aov(self.ctx, n, _p(ij, _i), k0, k1, *[_p(a, _f) for a in out])
        return tuple(out)

    def render_denoiser_inputs(self):
        """(imagedouble, sample_count, albedo sums, normal sums) of the has_denoiser accumulation (no splat)."""
        img, alb, nrm = (np.zeros((self.H, self.W, 3), np.float32) for _ in range(3))
        cnt = np.zeros((self.H, self.W), np.float32)
        self.lib.ref_render_denoiser_inputs(self.ctx, _p(img, _f), _p(cnt, _f), _p(alb, _f), _p(nrm, _f))
        return img, cnt, alb, nrm

    def render_seeded(self):
        img = np.zeros((self.H, self.W, 3), np.float32)
        cnt = np.zeros((self.H, self.W), np.float32)
        self.lib.ref_render_seeded(self.ctx, _p(img, _f), _p(cnt, _f))
        return img, cnt

    def max_threads(self):
        return self.lib.ref_max_threads()


class Ref(_Base):
    """The compiled reference (oracle/_ref/libptref.so)."""
    PREFIX = "ref_"
    PATH = REF_LIB_PATH

    def __init__(self):
        super().__init__()
        self.cdll.ref_time_render_nopreviz.restype = C.c_double
        self.cdll.ref_time_render_image.restype = C.c_double

    # ---- .scn scene files (the reference's own Raytracer::save_scene / load_scene)
    def save_scene(self, path):
        self.lib.ref_save_scene(self.ctx, str(path).encode())

    def load_scene(self, path):
        self.lib.ref_load_scene(self.ctx, str(path).encode())
        hdr = self.scene_header()
        self.W, self.H, self.spp = int(hdr[0]), int(hdr[1]), int(hdr[2])

    # ---- key-framed transforms
    def set_frame(self, frame):
        self.lib.ref_set_frame(self.ctx, int(frame))

    def add_keyframe(self, obj, frame):
        self.lib.ref_add_keyframe(self.ctx, int(obj), int(frame))

    def set_object_transform(self, obj, translation, rotation9, scale):
        t = np.ascontiguousarray(translation, np.float32); r = np.ascontiguousarray(rotation9, np.float32).reshape(9)
        self.lib.ref_set_object_transform(self.ctx, int(obj), _p(t, _f), _p(r, _f), _f(scale))

    def num_objects(self):
        return self.lib.ref_num_objects(self.ctx)

    def scene_header(self):
        o = np.zeros(32, np.float32)
        self.lib.ref_get_scene_header(self.ctx, _p(o, _f))
        return o

    def object_state(self, obj):
        o = np.zeros(24, np.float32); fl = np.zeros(8, np.int32)
        self.lib.ref_get_object_state(self.ctx, obj, _p(o, _f), _p(fl, _i))
        return o, fl

    def add_mesh_obj(self, path, scale=30.0, center=True):
        """The reference's own TriMesh(&scene, path, ...): readOBJ + MTL + stb_image."""
        return self.lib.ref_add_mesh(self.ctx, str(path).encode(), _f(scale), 1 if center else 0)

    def group_materials(self, obj):
        out = []
        for g in range(self.lib.ref_num_groups(self.ctx, obj)):
            m = np.zeros(12, np.float32); wh = np.zeros(8, np.int32)
            self.lib.ref_get_group_material(self.ctx, obj, g, _p(m, _f), _p(wh, _i))
            out.append((m, wh.reshape(4, 2)))
        return out

    def group_texture(self, obj, grp, slot):
        m, wh = self.group_materials(obj)[grp]
        W, H = int(wh[slot][0]), int(wh[slot][1])
        if W == 0:
            return None
        self.lib.ref_group_texture_values.restype = C.POINTER(_f)
        p = self.lib.ref_group_texture_values(self.ctx, obj, grp, slot)
        return np.ctypeslib.as_array(p, shape=(H, W, 3)).copy()

    def add_mesh(self, mesh, scale=30.0, center=True, tmpdir=None):
        """Writes the mesh as OBJ text and lets the reference's own readOBJ parse it."""
        import tempfile
        from pathtracer_amd import scenes
        d = tmpdir or tempfile.mkdtemp(prefix="ptref_obj_")
        path = os.path.join(d, mesh.name + ".obj")
        if hasattr(self.lib, "ref_write_obj") and mesh.faces_v.shape[0] > 100000:    # same text, written from C
            v = np.ascontiguousarray(mesh.vertices, np.float32); n = np.ascontiguousarray(mesh.normals, np.float32)
            fv = np.ascontiguousarray(mesh.faces_v, np.int32); fn = np.ascontiguousarray(mesh.faces_n, np.int32)
            uv = ft = None
            if mesh.uvs is not None:
                uv = np.ascontiguousarray(mesh.uvs, np.float32); ft = np.ascontiguousarray(mesh.faces_t, np.int32)
            self.lib.ref_write_obj.restype = C.c_int
            rc = self.lib.ref_write_obj(path.encode(), v.shape[0], _p(v, _f), n.shape[0], _p(n, _f), 0 if uv is None else uv.shape[0],
                                        None if uv is None else _p(uv, _f), fv.shape[0], _p(fv, _i), _p(fn, _i), None if ft is None else _p(ft, _i))
            if rc != 0:
                raise OSError("ref_write_obj failed: " + path)
        else:
            scenes.write_obj(mesh, path)
        rid = self.lib.ref_add_mesh(self.ctx, path.encode(), _f(scale), 1 if center else 0)
        if tmpdir is None:
            try:
                os.remove(path); os.rmdir(d)
            except OSError:
                pass
        return rid

    @staticmethod
    def _write_ppm(rgb8):
        import tempfile
        f = tempfile.NamedTemporaryFile(prefix="ptref_tex_", suffix=".ppm", delete=False)
        f.write(b"P6\n%d %d\n255\n" % (rgb8.shape[1], rgb8.shape[0]))
        f.write(rgb8.tobytes())
        f.close()
        return f.name

    @staticmethod
    def _write_merl(table):
        import tempfile
        f = tempfile.NamedTemporaryFile(prefix="ptref_merl_", suffix=".binary", delete=False)
        f.write(np.array([90, 90, 180], np.int32).tobytes())
        f.write(np.ascontiguousarray(table, np.float64).tobytes())
        f.close()
        return f.name

    def _set_brdf_merl(self, obj, table):
        self.cdll.ref_set_brdf_merl_file(self.ctx, obj, self._write_merl(table).encode())

    def merl_eval(self, table, wi, wo, N):
        wi = np.ascontiguousarray(wi, np.float32); wo = np.ascontiguousarray(wo, np.float32); N = np.ascontiguousarray(N, np.float32)
        out = np.zeros_like(wi)
        self.cdll.ref_merl_eval(self._write_merl(table).encode(), wi.shape[0], _p(wi, _f), _p(wo, _f), _p(N, _f), _p(out, _f))
        return out

    def _set_group_texture(self, obj, grp, slot, rgb8):
        self.cdll.ref_set_group_texture_file(self.ctx, obj, grp, slot, self._write_ppm(rgb8).encode())

    def set_envmap(self, rgb8):
        rgb8 = np.ascontiguousarray(rgb8, np.uint8)
        self.cdll.ref_set_envmap_file(self.ctx, self._write_ppm(rgb8).encode())

    def time_render_nopreviz(self, threads):
        img = np.zeros((self.H, self.W, 3), np.float32)
        t = self.lib.ref_time_render_nopreviz(self.ctx, threads, _p(img, _f))
        return t, img

    def time_render_image(self, threads):
        img = np.zeros((self.H, self.W, 3), np.float32)
        cnt = np.zeros((self.H, self.W), np.float32)
        t = self.lib.ref_time_render_image(self.ctx, threads, _p(img, _f), _p(cnt, _f))
        return t, img, cnt


    def mesh_bvh_figures(self, obj):
        """bvh_depth, bvh_avg_depth, bvh_nb_nodes, max_bvh_triangles as TriMesh::build_bvh left them."""
        o = np.zeros(4, np.float32)
        self.cdll.ref_mesh_bvh_figures(self.ctx, obj, _p(o, _f))
        return o

    def add_cylinder(self, A, B, R):
        return self.cdll.ref_add_cylinder(self.ctx, (_f * 3)(*A), (_f * 3)(*B), _f(R))


class RefMipt(Ref):
    """The reference compiled with -DUSE_MIPT (integration/use_mipt): Raytracer::render_image[_nopreviz], Scene::intersection
    and TriMesh::build_bvh of the REFERENCE'S OWN classes run on libmipt.so.  Everything `Ref` offers works unchanged; the
    methods below read the members the binding adds."""
    PATH = REF_MIPT_LIB_PATH

    def __init__(self):
        super().__init__()
        assert self.cdll.ref_mipt_built_with_switch() == 1
        self.cdll.ref_mipt_error.restype = C.c_char_p

    def upload(self):
        """Raytracer::mipt_upload(): what render_image does after prepare_render.  Returns the MIPT_* status."""
        return self.cdll.ref_mipt_upload(self.ctx)

    def status(self):
        return self.cdll.ref_mipt_status(self.ctx)

    def resident(self):
        return bool(self.cdll.ref_mipt_resident(self.ctx))

    def error(self):
        return self.cdll.ref_mipt_error(self.ctx).decode()

    def set_lookahead(self, n):
        self.cdll.ref_mipt_set_lookahead(self.ctx, int(n))

    def set_dirty(self):
        self.cdll.ref_mipt_set_dirty(self.ctx)

    def stats(self):
        o = np.zeros(6, np.uint64)
        rc = self.cdll.ref_mipt_stats(self.ctx, _p(o, C.c_uint64))
        if rc != 0:
            return None
        return dict(paths=int(o[0]), rays_closest=int(o[1]), rays_shadow=int(o[2]), pipeline=int(o[3]), passes=int(o[4]), replayed=int(o[5]))

    def render_image_stop_at(self, stop_at):
        img = np.zeros((self.H, self.W, 3), np.float32)
        cnt = np.zeros((self.H, self.W), np.float32)
        it = self.cdll.ref_mipt_render_image_stop_at(self.ctx, int(stop_at), _p(img, _f), _p(cnt, _f))
        return it, img, cnt

    def render_nopreviz_denoiser(self):
        img, alb, nrm = (np.zeros((self.H, self.W, 3), np.float32) for _ in range(3))
        cnt = np.zeros((self.H, self.W), np.float32)
        self.cdll.ref_mipt_render_nopreviz_denoiser(self.ctx, _p(img, _f), _p(cnt, _f), _p(alb, _f), _p(nrm, _f))
        return img, cnt, alb, nrm

    def image_u8(self):
        out = np.zeros((self.H, self.W, 3), np.uint8)
        self.cdll.ref_mipt_get_image_u8(self.ctx, _p(out, C.c_ubyte))
        return out


class Oracle(_Base):
    """Our plain-C restatement (oracle/libptoracle.so)."""
    PREFIX = "o_"
    PATH = ORACLE_LIB_PATH

    def __init__(self):
        super().__init__()
        self.cdll.o_render_omp.restype = C.c_double

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
        return self.lib.ref_add_mesh(self.ctx, v.shape[0], _p(v, _f), n.shape[0], _p(n, _f), nt, uvp,
                                     fv.shape[0], _p(fv, _i), _p(fn, _i), ftp, _f(scale), 1 if center else 0)

    def merl_eval(self, table, wi, wo, N):
        table = np.ascontiguousarray(table, np.float64).ravel()
        wi = np.ascontiguousarray(wi, np.float32); wo = np.ascontiguousarray(wo, np.float32); N = np.ascontiguousarray(N, np.float32)
        out = np.zeros_like(wi)
        self.cdll.o_merl_eval(_p(table, C.c_double), wi.shape[0], _p(wi, _f), _p(wo, _f), _p(N, _f), _p(out, _f))
        return out

    def render_omp(self, threads):
        img = np.zeros((self.H, self.W, 3), np.float32)
        cnt = np.zeros((self.H, self.W), np.float32)
        rays = np.zeros(2, np.uint64)
        t = self.cdll.o_render_omp(self.ctx, threads, _p(img, _f), _p(cnt, _f), _p(rays, C.c_uint64))
        return t, img, cnt, rays

    def counters_reset(self):
        self.cdll.o_counters_reset()

    def counters(self):
        out = np.zeros(8, np.uint64)
        self.cdll.o_counters_get(_p(out, C.c_uint64))
        return out
