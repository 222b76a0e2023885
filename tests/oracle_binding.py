"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference
integrator).  Test infrastructure: imported only by tests, the smoke check and
the benchmark's cpu_baseline leg -- never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)


class OScene(C.Structure):
    _fields_ = [
        ("vertices", _f32p), ("n_vertices", C.c_uint32),
        ("indices", _f32p), ("n_triangles", C.c_uint32),
        ("vertex_data", _f32p),
        ("lights", _f32p), ("n_lights", C.c_uint32),
        ("materials", _f32p), ("n_materials", C.c_uint32),
        ("material_map", _u32p),
    ]


class OCamera(C.Structure):
    _fields_ = [
        ("eye", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
        ("width", C.c_uint32), ("height", C.c_uint32), ("tan_half_fov", C.c_float),
    ]


class OStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("paths", "segments", "light_paths", "escaped_paths", "zeroed_paths", "nan_samples")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build_if_needed() -> None:
    src = os.path.join(ORACLE_DIR, "pt_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_if_needed()
        L = C.CDLL(LIB)
        L.oracle_rng.restype = C.c_float
        L.oracle_rng.argtypes = [_u32p]
        L.oracle_raycast.restype = None
        L.oracle_raycast.argtypes = [C.POINTER(OCamera), C.c_uint32, C.c_uint32, C.c_uint32, _f32p, _f32p, _u32p]
        L.oracle_intersect.restype = C.c_int32
        L.oracle_intersect.argtypes = [C.POINTER(OScene), _f32p, _f32p, _f32p]
        L.oracle_trace.restype = None
        L.oracle_trace.argtypes = [C.POINTER(OScene), C.c_uint64, _f32p, _f32p, _f32p, _i32p]
        L.oracle_primary.restype = None
        L.oracle_primary.argtypes = [C.POINTER(OScene), C.POINTER(OCamera), C.c_uint32, _i32p, _f32p, _f32p]
        L.oracle_sample_surface.restype = C.c_int
        L.oracle_sample_surface.argtypes = [_f32p, _f32p, _f32p, _f32p, _u32p, _f32p, _f32p, _f32p]
        L.oracle_render.restype = None
        L.oracle_render.argtypes = [C.POINTER(OScene), C.POINTER(OCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                    _f32p, C.POINTER(OStats), C.c_int]
        L.oracle_render_rgba8.restype = None
        L.oracle_render_rgba8.argtypes = [C.POINTER(OScene), C.POINTER(OCamera), C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.POINTER(C.c_uint8), C.c_int]
        L.oracle_tonemap.restype = None
        L.oracle_tonemap.argtypes = [_f32p, _f32p, C.c_uint64]
        L.oracle_light_inverse.restype = C.c_int
        L.oracle_light_inverse.argtypes = [_f32p, _f32p]
        _lib = L
    return _lib


def _p(a, ty=_f32p):
    return a.ctypes.data_as(ty)


# tan(FOVY * 0.5f), FOVY = 19.5f * PI / 180.f, as the shader compiler folds it
# (pathtracer.frag.glsl:306,315): bits 0x3E2FF494.
TAN_HALF_FOV_DEFAULT = np.array([0x3E2FF494], np.uint32).view(np.float32)[0]


class OracleScene:
    """Keeps the numpy arrays alive next to the C struct that points at them."""

    def __init__(self, arrays: dict):
        self.a = {
            "vertices": np.ascontiguousarray(arrays["vertices"], np.float32).reshape(-1, 3),
            "indices": np.ascontiguousarray(arrays["indices"], np.float32).reshape(-1, 4),
            "vertex_data": np.ascontiguousarray(arrays["vertex_data"], np.float32).reshape(-1, 6),
            "lights": np.ascontiguousarray(arrays["lights"], np.float32).reshape(-1, 20),
            "materials": np.ascontiguousarray(arrays["materials"], np.float32).reshape(-1, 8),
            "material_map": np.ascontiguousarray(arrays["material_map"], np.uint32).reshape(-1),
        }
        a = self.a
        self.c = OScene(_p(a["vertices"]), a["vertices"].shape[0], _p(a["indices"]), a["indices"].shape[0],
                        _p(a["vertex_data"]), _p(a["lights"]), a["lights"].shape[0], _p(a["materials"]),
                        a["materials"].shape[0], _p(a["material_map"], _u32p))


def make_camera(eye, forward, up, right, width, height, tan_half_fov=TAN_HALF_FOV_DEFAULT) -> OCamera:
    cam = OCamera()
    for k in range(3):
        cam.eye[k], cam.forward[k], cam.up[k], cam.right[k] = (float(eye[k]), float(forward[k]), float(up[k]),
                                                               float(right[k]))
    cam.width, cam.height = int(width), int(height)
    cam.tan_half_fov = float(tan_half_fov)
    return cam


def primary(scene: OracleScene, cam: OCamera, iteration=0, want_t=True, want_dirs=True):
    n = cam.width * cam.height
    ids = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32) if want_t else None
    dirs = np.zeros((n, 3), np.float32) if want_dirs else None
    lib().oracle_primary(C.byref(scene.c), C.byref(cam), iteration, _p(ids, _i32p),
                         None if t is None else _p(t), None if dirs is None else _p(dirs))
    shape = (cam.height, cam.width)
    return ids.reshape(shape), None if t is None else t.reshape(shape), None if dirs is None else dirs.reshape(shape + (3,))


def primary_dirs_only(scene: OracleScene, cam: OCamera, iteration=0):
    """Ray directions of one iteration without intersecting anything."""
    n = cam.width * cam.height
    dirs = np.zeros((n, 3), np.float32)
    lib().oracle_primary(C.byref(scene.c), C.byref(cam), iteration, None, None, _p(dirs))
    return None, None, dirs.reshape(cam.height, cam.width, 3)


def trace(scene: OracleScene, origins, dirs):
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    n = o.shape[0]
    t = np.zeros(n, np.float32)
    ids = np.zeros(n, np.int32)
    lib().oracle_trace(C.byref(scene.c), n, _p(o), _p(d), _p(t), _p(ids, _i32p))
    return t, ids


def poisoned(scene: OracleScene, origins, dirs):
    """True for rays lying exactly in some triangle's plane (st = 0/0 there)."""
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    flags = np.zeros(o.shape[0], np.uint8)
    L = lib()
    L.oracle_poisoned.restype = None
    L.oracle_poisoned.argtypes = [C.POINTER(OScene), C.c_uint64, _f32p, _f32p, C.POINTER(C.c_uint8)]
    L.oracle_poisoned(C.byref(scene.c), o.shape[0], _p(o), _p(d), flags.ctypes.data_as(C.POINTER(C.c_uint8)))
    return flags.astype(bool)


def render(scene: OracleScene, cam: OCamera, it_begin, it_end, max_depth, nthreads=1, sum_rgba=None):
    if sum_rgba is None:
        sum_rgba = np.zeros((cam.height, cam.width, 4), np.float32)
    st = OStats()
    lib().oracle_render(C.byref(scene.c), C.byref(cam), it_begin, it_end, max_depth, _p(sum_rgba), C.byref(st),
                        nthreads)
    return sum_rgba, st.as_dict()


def render_rgba8(scene: OracleScene, cam: OCamera, it_begin, it_end, max_depth, nthreads=1, accum=None):
    if accum is None:
        accum = np.zeros((cam.height, cam.width, 4), np.uint8)
    lib().oracle_render_rgba8(C.byref(scene.c), C.byref(cam), it_begin, it_end, max_depth,
                              accum.ctypes.data_as(C.POINTER(C.c_uint8)), nthreads)
    return accum


def sample_surface(material, normal, wo, xi, seed):
    m = np.ascontiguousarray(material, np.float32)
    n = np.ascontiguousarray(normal, np.float32)
    w = np.ascontiguousarray(wo, np.float32)
    x = np.ascontiguousarray(xi, np.float32)
    s = np.ascontiguousarray(seed, np.uint32).copy()
    wi = np.zeros(3, np.float32)
    pdf = C.c_float(0)
    f = np.zeros(3, np.float32)
    early = lib().oracle_sample_surface(_p(m), _p(n), _p(w), _p(x), _p(s, _u32p), _p(wi), C.byref(pdf), _p(f))
    return wi, pdf.value, f, early, s


def rng_stream(seed, n):
    s = np.ascontiguousarray(seed, np.uint32).copy()
    out = np.zeros(n, np.float32)
    for i in range(n):
        out[i] = lib().oracle_rng(_p(s, _u32p))
    return out, s


def tonemap(rgb):
    a = np.ascontiguousarray(rgb, np.float32)
    out = np.zeros_like(a)
    lib().oracle_tonemap(_p(a), _p(out), a.size // 3)
    return out


def light_inverse(m16):
    m = np.ascontiguousarray(m16, np.float32).reshape(16)
    inv = np.zeros(16, np.float32)
    ok = lib().oracle_light_inverse(_p(m), _p(inv))
    return inv, bool(ok)
