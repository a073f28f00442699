"""ctypes binding over librtr_host.so (scene building) and librtr_hip.so (C ABI).

Every compute call goes ``HostScene`` -> ``rtr_*`` C ABI -> HIP kernels.  There
is no fallback: a missing library or a failing device call raises ``RtrError``.
"""
import ctypes as C
import os

import numpy as np

from .scenes import default_scene_spec  # noqa: F401  (part of this module's interface)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTR_HIP_SO: another build of the library (development: variants built with experiment switches)
HIP_SO = os.environ.get("RTR_HIP_SO") or os.path.join(_HERE, "librtr_hip.so")
HOST_SO = os.path.join(_HERE, "librtr_host.so")

DIFFUSE, SPECULAR, MICROFACET = 0, 1, 2

f32 = np.float32
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)


MODE_AUTO, MODE_WAVEFRONT, MODE_FUSED = 0, 1, 2  # rtr_render_params.mode (include/rtr.h)


class RtrError(RuntimeError):
    pass


class RenderParams(C.Structure):
    """rtr_render_params (include/rtr.h)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("seed", C.c_uint32), ("max_depth", C.c_int32), ("rr_prob", C.c_float),
                ("chunk_spp", C.c_int32), ("pool_size", C.c_int32),
                ("part_index", C.c_int32), ("part_count", C.c_int32), ("tile_cols", C.c_int32),
                ("profile", C.c_int32), ("mode", C.c_int32), ("streams", C.c_int32)]


class Stats(C.Structure):
    """rtr_stats (include/rtr.h)."""
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("vertices", C.c_uint64),
                ("iterations", C.c_uint64), ("chunk_spp", C.c_int32), ("pool_size", C.c_int32),
                ("n_columns", C.c_int32), ("mode", C.c_int32), ("ms_total", C.c_double),
                ("ms_trace", C.c_double), ("ms_logic", C.c_double), ("ms_other", C.c_double),
                ("trace_launches", C.c_uint64), ("ms_fused", C.c_double),
                ("fused_launches", C.c_uint64), ("ms_gather", C.c_double),
                ("n_devices", C.c_int32), ("reserved", C.c_int32),
                ("boxes_tested", C.c_uint64), ("prims_tested", C.c_uint64),
                ("ms_render_min", C.c_double), ("ms_render_max", C.c_double)]


class Camera(C.Structure):
    """rtr_camera (include/rtr.h)."""
    _fields_ = [("position", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("fov", C.c_float)]


# every symbol include/rtr.h declares
ABI_SYMBOLS = [
    "rtr_last_error", "rtr_device_count", "rtr_scene_create", "rtr_scene_destroy",
    "rtr_trace_closest", "rtr_trace_closest_device", "rtr_material_eval", "rtr_material_sample",
    "rtr_sample_light", "rtr_stream_uniform", "rtr_render_pt", "rtr_render_pt_device",
    "rtr_partition_columns", "rtr_check_connectable", "rtr_create_photon_map",
    "rtr_create_photon_map_device", "rtr_render_pt_device_noalloc", "rtr_render_pt_multi",
    "rtr_render_pt_multi_device", "rtr_trace_count_work", "rtr_self_test", "rtr_scene_read_tree",
    "rtr_scene_read_pairs", "rtr_build_bvh",
]

_hip = None
_host = None


def hip_lib():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise RtrError("librtr_hip.so is not built; run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (there is no CPU fallback)")
        L = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        L.rtr_last_error.restype = C.c_char_p
        L.rtr_device_count.argtypes = [_ip]
        L.rtr_scene_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rtr_scene_destroy.argtypes = [C.c_void_p]
        L.rtr_scene_destroy.restype = None
        L.rtr_trace_closest.argtypes = [C.c_void_p, _fp, _fp, C.c_size_t, _ip, _fp, _fp, _fp]
        L.rtr_trace_closest_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        L.rtr_trace_count_work.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64)]
        L.rtr_check_connectable.argtypes = [C.c_void_p, _fp, _fp, C.c_size_t, C.POINTER(C.c_uint8)]
        L.rtr_create_photon_map.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, _fp, _fp, _fp,
                                            C.POINTER(C.c_uint64), _up]
        L.rtr_create_photon_map_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), _up]
        L.rtr_material_eval.argtypes = [C.c_void_p, _ip, _fp, _fp, _fp, C.c_size_t, _fp, _fp]
        L.rtr_material_sample.argtypes = [C.c_void_p, _ip, _fp, _fp, _fp, C.c_size_t, _fp]
        L.rtr_sample_light.argtypes = [C.c_void_p, _fp, C.c_int32, C.c_size_t, _fp, _fp, _fp, _ip,
                                       _ip]
        L.rtr_stream_uniform.argtypes = [C.c_void_p, C.c_uint32, _up, _up, _up, _up, C.c_size_t, _fp]
        L.rtr_render_pt.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), _fp,
                                    C.POINTER(Stats)]
        L.rtr_render_pt_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams),
                                           C.c_void_p, C.POINTER(Stats)]
        L.rtr_render_pt_device_noalloc.argtypes = L.rtr_render_pt_device.argtypes
        L.rtr_render_pt_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Camera),
                                          C.POINTER(RenderParams), _fp, C.POINTER(Stats)]
        L.rtr_render_pt_multi_device.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Camera),
                                                 C.POINTER(RenderParams), C.c_void_p,
                                                 C.POINTER(Stats)]
        L.rtr_partition_columns.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ip, _ip]
        L.rtr_build_bvh.argtypes = [C.c_int, C.c_int32, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp),
                                    C.POINTER(_fp), _ip, _ip, _ip, _ip]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_SO):
            raise RtrError("librtr_host.so is not built; run __graft_entry__.build()")
        L = C.CDLL(HOST_SO)
        L.rtrh_last_error.restype = C.c_char_p
        L.rtrh_scene_new.restype = C.c_void_p
        L.rtrh_scene_new.argtypes = [_fp, _fp, _fp, C.c_float]
        L.rtrh_scene_free.argtypes = [C.c_void_p]
        L.rtrh_scene_free.restype = None
        L.rtrh_add_material.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _fp, C.c_float, C.c_float]
        L.rtrh_add_sphere.argtypes = [C.c_void_p, C.c_int, _fp, C.c_float]
        L.rtrh_add_mesh.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int, _up, C.c_int]
        L.rtrh_build.argtypes = [C.c_void_p]
        L.rtrh_flatten.argtypes = [C.c_void_p, _fp, _fp, _ip, _ip]
        L.rtrh_num_prims.argtypes = [C.c_void_p]
        L.rtrh_object_area.argtypes = [C.c_void_p, C.c_int]
        L.rtrh_object_area.restype = C.c_float
        L.rtrh_lights.argtypes = [C.c_void_p, _ip, _ip, _fp, _ip, _ip, _ip]
        L.rtrh_set_device.argtypes = [C.c_void_p, C.c_int]
        L.rtrh_device_scene.argtypes = [C.c_void_p]
        L.rtrh_device_scene.restype = C.c_void_p
        L.rtrh_material_id.argtypes = [C.c_void_p, C.c_int]
        L.rtrh_intersect.argtypes = [C.c_void_p, _fp, _fp, C.c_int64, _ip, _fp, _fp, _fp]
        L.rtrh_check_connectable.argtypes = [C.c_void_p, _fp, _fp, C.c_int64, C.POINTER(C.c_uint8)]
        L.rtrh_set_node_box.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        L.rtrh_create_photon_map.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _fp, _fp, _fp]
        L.rtrh_create_photon_map.restype = C.c_int64
        L.rtrh_set_render_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.rtrh_render.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_char_p,
                                  C.POINTER(RenderParams), _fp, C.POINTER(Stats)]
        L.rtrh_save_file.argtypes = [_fp, C.c_uint, C.c_uint, C.c_char_p]
        _host = L
    return _host


def save_ppm(rgb, height, width, file):
    """Render::saveFile of the host library (reference src/Render.cpp:3-21) on a float buffer of
    height * width pixels in the renderer's i * height + j order: writes <file>.ppm."""
    L = host_lib()
    rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    if len(rgb) != height * width:
        raise ValueError("save_ppm: buffer does not hold height * width pixels")
    if L.rtrh_save_file(rgb.ctypes.data_as(_fp), height, width, str(file).encode()) != 0:
        raise RtrError("rtrh_save_file: %s" % L.rtrh_last_error().decode())


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _u(a):
    return a.ctypes.data_as(_up)


def _vec(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _check(rc, what):
    if rc != 0:
        raise RtrError("%s failed (%d): %s" % (what, rc, hip_lib().rtr_last_error().decode()))


def device_count():
    n = C.c_int32(0)
    rc = hip_lib().rtr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def self_test(device=0):
    """rtr_self_test: number of mismatches of the kernels' arithmetic shortcuts (must be 0)."""
    n = C.c_uint64(0)
    L = hip_lib()
    L.rtr_self_test.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    _check(L.rtr_self_test(device, C.byref(n)), "rtr_self_test")
    return n.value


def build_bvh(bmin, bmax, device=0):
    """rtr_build_bvh: the reference's BVH::selfBuildBVH over boxes bmin / bmax (n x 3), built on
    the device.  Returns (node_bmin[N,3], node_bmax[N,3], left[N], right[N], object[N],
    root_order[n]) with N = 2n - 1, node 0 the root."""
    bmin = np.ascontiguousarray(np.asarray(bmin, f32).reshape(-1, 3).T)
    bmax = np.ascontiguousarray(np.asarray(bmax, f32).reshape(-1, 3).T)
    n = bmin.shape[1]
    N = max(2 * n - 1, 1)
    nlo = np.zeros((3, N), f32)
    nhi = np.zeros((3, N), f32)
    left = np.zeros(N, np.int32)
    right = np.zeros(N, np.int32)
    obj = np.zeros(N, np.int32)
    order = np.zeros(max(n, 1), np.int32)
    axes = lambda a: (_fp * 3)(*[_f(a[k]) for k in range(3)])
    _check(hip_lib().rtr_build_bvh(device, n, axes(bmin), axes(bmax), axes(nlo), axes(nhi), _i(left),
                                   _i(right), _i(obj), _i(order)), "rtr_build_bvh")
    return nlo.T.copy(), nhi.T.copy(), left, right, obj, order


def partition_columns(width, part_index, part_count, tile_cols=8):
    n = C.c_int32(0)
    _check(hip_lib().rtr_partition_columns(width, part_index, part_count, tile_cols, C.byref(n), None),
           "rtr_partition_columns")
    cols = np.zeros(n.value, np.int32)
    _check(hip_lib().rtr_partition_columns(width, part_index, part_count, tile_cols, C.byref(n),
                                           _i(cols)), "rtr_partition_columns")
    return cols


class HostScene:
    """A scene built through the host library's mirror of the reference API."""

    def __init__(self, camera):
        self.L = host_lib()
        self.cam = dict(camera)
        self.h = C.c_void_p(self.L.rtrh_scene_new(_vec(camera["pos"]), _vec(camera["front"]),
                                                  _vec(camera["up"]), C.c_float(camera["fov"])))
        self.mat_ids = {}
        self._dev = None

    def close(self):
        self.close_replicas()
        if self.h:
            self.L.rtrh_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _herr(self, what):
        raise RtrError("%s: %s" % (what, self.L.rtrh_last_error().decode()))

    # ---- construction: Material / Sphere / TrangleMesh / Scene::addObject / rebuildBVH ----
    def add_material(self, name, kind, emission=(0, 0, 0), kd=(0, 0, 0), ks=(0, 0, 0), ior=1e-5,
                     roughness=1e-5):
        self.mat_ids[name] = self.L.rtrh_add_material(self.h, kind, _vec(emission), _vec(kd),
                                                      _vec(ks), C.c_float(ior), C.c_float(roughness))
        return self.mat_ids[name]

    def add_sphere(self, material, center, radius):
        r = self.L.rtrh_add_sphere(self.h, self.mat_ids[material], _vec(center), C.c_float(radius))
        if r < 0:
            self._herr("add_sphere")
        return r

    def add_mesh(self, material, positions, indices):
        pos = np.ascontiguousarray(positions, f32)
        idx = np.ascontiguousarray(indices, np.uint32)
        r = self.L.rtrh_add_mesh(self.h, self.mat_ids[material], _f(pos), len(pos), _u(idx), len(idx))
        if r < 0:
            self._herr("add_mesh")
        return r

    def build(self):
        if self.L.rtrh_build(self.h) != 0:
            self._herr("build")
        self._invalidate_device()

    @classmethod
    def from_spec(cls, spec):
        s = cls(spec["camera"])
        for name, kind, e, kd, ks, ior, rough in spec["materials"]:
            s.add_material(name, kind, e, kd, ks, ior, rough)
        for ob in spec["objects"]:
            if ob[0] == "sphere":
                s.add_sphere(ob[1], ob[2], ob[3])
            else:
                s.add_mesh(ob[1], ob[2], ob[3])
        s.build()
        return s

    # ---- host-side introspection (no GPU) ----
    def flatten(self):
        n = self.L.rtrh_flatten(self.h, None, None, None, None)
        if n < 0:
            self._herr("flatten")
        lo = np.zeros((n, 3), f32)
        hi = np.zeros((n, 3), f32)
        child = np.zeros((n, 2), np.int32)
        prim = np.zeros(n, np.int32)
        self.L.rtrh_flatten(self.h, _f(lo), _f(hi), _i(child), _i(prim))
        return lo, hi, child, prim

    def set_node_box(self, flat_index, lo, hi):
        """Scene::setNodeBound: overwrite one node's box (numbering of flatten())."""
        lo = np.ascontiguousarray(lo, f32)
        hi = np.ascontiguousarray(hi, f32)
        if self.L.rtrh_set_node_box(self.h, int(flat_index), _f(lo), _f(hi)) != 0:
            raise RtrError("no node %d" % flat_index)
        self._invalidate_device()

    def num_prims(self):
        return self.L.rtrh_num_prims(self.h)

    def object_area(self, o):
        return f32(self.L.rtrh_object_area(self.h, o))

    def lights(self):
        n = self.L.rtrh_lights(self.h, None, None, None, None, None, None)
        kind = np.zeros(n, np.int32)
        prim = np.zeros(n, np.int32)
        area = np.zeros(n, f32)
        begin = np.zeros(n + 1, np.int32)
        nt = C.c_int32(0)
        self.L.rtrh_lights(self.h, _i(kind), _i(prim), _f(area), _i(begin), None, C.byref(nt))
        tris = np.zeros(max(nt.value, 1), np.int32)
        self.L.rtrh_lights(self.h, None, None, None, None, _i(tris), None)
        return kind, prim, area, begin, tris[:nt.value]

    def material_id(self, name):
        return self.L.rtrh_material_id(self.h, self.mat_ids[name])

    # ---- device: everything below calls the C ABI of include/rtr.h ----
    def set_device(self, index):
        self.L.rtrh_set_device(self.h, index)
        self._invalidate_device()

    def _invalidate_device(self):
        """The scene changed: every upload of it is stale, the per-device replicas of
        render_multi included (a later render_multi uploads again)."""
        self._dev = None
        self.close_replicas()

    def flatten_for_upload(self):
        """Scene::flatten of the host mirror (cached there): the arrays rtr_scene_create takes."""
        d = self.L.rtrh_scene_desc
        d.restype = C.c_void_p
        d.argtypes = [C.c_void_p]
        if not d(self.h):
            self._herr("scene desc")

    def device_scene(self):
        if self._dev is None:
            p = self.L.rtrh_device_scene(self.h)
            if not p:
                self._herr("device scene")
            self._dev = C.c_void_p(p)
        return self._dev

    def intersect(self, org, dirs):
        """rtr_trace_closest: (prim, t, pos, nrm)."""
        org = np.ascontiguousarray(org, f32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, f32).reshape(-1, 3)
        n = len(org)
        prim = np.full(n, -1, np.int32)
        t = np.zeros(n, f32)
        pos = np.zeros((n, 3), f32)
        nrm = np.zeros((n, 3), f32)
        _check(hip_lib().rtr_trace_closest(self.device_scene(), _f(org), _f(dirs), n, _i(prim),
                                           _f(t), _f(pos), _f(nrm)), "rtr_trace_closest")
        return prim, t, pos, nrm

    def read_tree(self):
        """rtr_scene_read_tree: (nodes[N,8] as float32, parent[N], leaves[L,16], (fast, nested))."""
        L = hip_lib()
        L.rtr_scene_read_tree.argtypes = [C.c_void_p, _ip, _ip, _fp, _ip, _fp, _ip]
        n, nl = C.c_int32(0), C.c_int32(0)
        _check(L.rtr_scene_read_tree(self.device_scene(), C.byref(n), C.byref(nl), None, None, None, None),
               "rtr_scene_read_tree")
        nodes = np.zeros((n.value, 8), f32)
        parent = np.zeros(n.value, np.int32)
        leaves = np.zeros((nl.value, 16), f32)
        flags = np.zeros(2, np.int32)
        _check(L.rtr_scene_read_tree(self.device_scene(), None, None, _f(nodes), _i(parent), _f(leaves),
                                     _i(flags)), "rtr_scene_read_tree")
        return nodes, parent, leaves, (int(flags[0]), int(flags[1]))

    def read_pairs(self):
        """rtr_scene_read_pairs: (records[R,16], seconds[P,12]) of the fused kernel's leaf pass."""
        L = hip_lib()
        L.rtr_scene_read_pairs.argtypes = [C.c_void_p, _ip, _ip, _fp, _fp]
        n, npairs = C.c_int32(0), C.c_int32(0)
        _check(L.rtr_scene_read_pairs(self.device_scene(), C.byref(n), C.byref(npairs), None, None),
               "rtr_scene_read_pairs")
        records = np.zeros((n.value, 16), f32)
        seconds = np.zeros((npairs.value, 12), f32)
        _check(L.rtr_scene_read_pairs(self.device_scene(), None, None, _f(records), _f(seconds)),
               "rtr_scene_read_pairs")
        return records, seconds

    def intersect_device(self, d_org, d_dir, n, d_prim, d_t, d_pos=None, d_nrm=None, stream=None):
        """rtr_trace_closest_device: device addresses as ints; enqueues, does not wait."""
        _check(hip_lib().rtr_trace_closest_device(self.device_scene(), C.c_void_p(d_org),
                                                  C.c_void_p(d_dir), n, C.c_void_p(d_prim),
                                                  C.c_void_p(d_t), C.c_void_p(d_pos),
                                                  C.c_void_p(d_nrm), C.c_void_p(stream)),
               "rtr_trace_closest_device")

    def count_work(self, d_org, d_dir, n):
        """rtr_trace_count_work: (boxes, triangles, spheres) the reference's traversal tests
        for the n rays at these device addresses."""
        out = [C.c_uint64(0) for _ in range(3)]
        _check(hip_lib().rtr_trace_count_work(self.device_scene(), C.c_void_p(d_org), C.c_void_p(d_dir),
                                              n, *[C.byref(o) for o in out]), "rtr_trace_count_work")
        return tuple(o.value for o in out)

    def create_photon_map(self, n, seed=0, via_host_api=False):
        """Render::createPhotonMap: (pos, wi, power, emission paths used); each array k x 3
        with k = n, or 0 for a scene without lights.  via_host_api: through the C++ mirror
        (Render::createPhotonMap) instead of the C ABI call directly."""
        out = [np.zeros((max(n, 1), 3), f32) for _ in range(3)]
        if via_host_api:
            k = self.L.rtrh_create_photon_map(self.h, seed, n, _f(out[0]), _f(out[1]), _f(out[2]))
            if k < 0:
                self._herr("create_photon_map")
            return [a[:k] for a in out] + [None]
        stored = C.c_uint64(0)
        used = np.zeros(1, np.uint32)
        _check(hip_lib().rtr_create_photon_map(self.device_scene(), n, seed, _f(out[0]), _f(out[1]),
                                               _f(out[2]), C.byref(stored), _u(used)),
               "rtr_create_photon_map")
        return [a[:stored.value] for a in out] + [int(used[0])]

    def check_connectable(self, a, b, via_host_api=False):
        """Scene::checkConnectable for many pairs: uint8 array, 1 where b is reached from a.
        via_host_api: through the C++ mirror (Scene::checkConnectableBatch) instead of the
        C ABI call directly."""
        a = np.ascontiguousarray(a, f32).reshape(-1, 3)
        b = np.ascontiguousarray(b, f32).reshape(-1, 3)
        out = np.zeros(len(a), np.uint8)
        op = out.ctypes.data_as(C.POINTER(C.c_uint8))
        if via_host_api:
            if host_lib().rtrh_check_connectable(self.h, _f(a), _f(b), len(a), op) != 0:
                self._herr("checkConnectable")
        else:
            _check(hip_lib().rtr_check_connectable(self.device_scene(), _f(a), _f(b), len(a), op),
                   "rtr_check_connectable")
        return out

    def material_eval(self, mat, wi, wo, n):
        wi, wo, n = [np.ascontiguousarray(a, f32) for a in (wi, wo, n)]
        ids = np.full(len(wi), self.material_id(mat), np.int32)
        out = np.zeros_like(wi)
        pdf = np.zeros(len(wi), f32)
        _check(hip_lib().rtr_material_eval(self.device_scene(), _i(ids), _f(wi), _f(wo), _f(n),
                                           len(wi), _f(out), _f(pdf)), "rtr_material_eval")
        return out, pdf

    def material_sample(self, mat, wi, n, u2):
        wi, n, u2 = [np.ascontiguousarray(a, f32) for a in (wi, n, u2)]
        ids = np.full(len(wi), self.material_id(mat), np.int32)
        out = np.zeros_like(wi)
        _check(hip_lib().rtr_material_sample(self.device_scene(), _i(ids), _f(wi), _f(n), _f(u2),
                                             len(wi), _f(out)), "rtr_material_sample")
        return out

    def sample_light(self, u):
        u = np.ascontiguousarray(u, f32)
        n, stride = u.shape
        pos = np.zeros((n, 3), f32)
        nrm = np.zeros((n, 3), f32)
        pdf = np.zeros(n, f32)
        prim = np.zeros(n, np.int32)
        used = np.zeros(n, np.int32)
        _check(hip_lib().rtr_sample_light(self.device_scene(), _f(u), stride, n, _f(pos), _f(nrm),
                                          _f(pdf), _i(prim), _i(used)), "rtr_sample_light")
        return pos, nrm, pdf, prim, used

    def stream_uniform(self, seed, pixel, sample, bounce, dim):
        arrs = [np.ascontiguousarray(a, np.uint32) for a in (pixel, sample, bounce, dim)]
        out = np.zeros(len(arrs[0]), f32)
        _check(hip_lib().rtr_stream_uniform(self.device_scene(), seed, *[_u(a) for a in arrs],
                                            len(out), _f(out)), "rtr_stream_uniform")
        return out

    def camera_struct(self):
        c = Camera()
        c.position[:] = [float(x) for x in self.cam["pos"]]
        c.front[:] = [float(x) for x in self.cam["front"]]
        c.up[:] = [float(x) for x in self.cam["up"]]
        c.fov = float(self.cam["fov"])
        return c

    def make_params(self, W, H, spp, seed=0, max_depth=0, rr_prob=0.0, chunk_spp=0, pool_size=0,
                    part_index=0, part_count=1, tile_cols=8, profile=False, mode=MODE_AUTO,
                    streams=0):
        return RenderParams(W, H, spp, seed, max_depth, rr_prob, chunk_spp, pool_size, part_index,
                            part_count, tile_cols, 1 if profile else 0, mode, streams)

    def render(self, W, H, spp, **kw):
        """rtr_render_pt into host memory: (rgb[n_cols*H,3], Stats)."""
        p = self.make_params(W, H, spp, **kw)
        ncol = len(partition_columns(W, p.part_index, p.part_count, p.tile_cols))
        rgb = np.zeros((ncol * H, 3), f32)
        st = Stats()
        cam = self.camera_struct()
        _check(hip_lib().rtr_render_pt(self.device_scene(), C.byref(cam), C.byref(p), _f(rgb),
                                       C.byref(st)), "rtr_render_pt")
        return rgb, st

    def render_device(self, W, H, spp, d_ptr, **kw):
        """rtr_render_pt_device into a device buffer (address as int)."""
        p = self.make_params(W, H, spp, **kw)
        st = Stats()
        cam = self.camera_struct()
        _check(hip_lib().rtr_render_pt_device(self.device_scene(), C.byref(cam), C.byref(p),
                                              C.c_void_p(d_ptr), C.byref(st)),
               "rtr_render_pt_device")
        return st

    def device_scenes(self, devices):
        """One rtr_scene per entry of `devices` (uploads of this scene; a device may repeat)."""
        key = tuple(int(d) for d in devices)
        if getattr(self, "_replicas_key", None) != key:
            self.close_replicas()
            d = self.L.rtrh_scene_desc
            d.restype = C.c_void_p
            d.argtypes = [C.c_void_p]
            desc = d(self.h)
            if not desc:
                self._herr("scene desc")
            out = []
            for dev in key:
                h = C.c_void_p()
                _check(hip_lib().rtr_scene_create(C.c_void_p(desc), dev, C.byref(h)), "rtr_scene_create")
                out.append(h)
            self._replicas, self._replicas_key = out, key
        return self._replicas

    def close_replicas(self):
        for h in getattr(self, "_replicas", []):
            hip_lib().rtr_scene_destroy(h)
        self._replicas, self._replicas_key = [], None

    def render_multi(self, W, H, spp, devices, d_ptr=None, **kw):
        """rtr_render_pt_multi: the whole image (rgb[W*H,3], Stats) rendered as len(devices)
        concurrent partitions; with d_ptr (device address on devices[0]) the image stays in
        device memory and only Stats is returned."""
        p = self.make_params(W, H, spp, **kw)
        if "part_count" not in kw:
            p.part_count = 0
        scenes = self.device_scenes(devices)
        arr = (C.c_void_p * len(scenes))(*[h.value for h in scenes])
        st = Stats()
        cam = self.camera_struct()
        if d_ptr is not None:
            _check(hip_lib().rtr_render_pt_multi_device(arr, len(scenes), C.byref(cam), C.byref(p),
                                                        C.c_void_p(d_ptr), C.byref(st)),
                   "rtr_render_pt_multi_device")
            return st
        rgb = np.zeros((W * H, 3), f32)
        _check(hip_lib().rtr_render_pt_multi(arr, len(scenes), C.byref(cam), C.byref(p), _f(rgb),
                                             C.byref(st)), "rtr_render_pt_multi")
        return rgb, st

    def render_device_noalloc(self, W, H, spp, d_ptr, **kw):
        """rtr_render_pt_device_noalloc into a device buffer (address as int)."""
        p = self.make_params(W, H, spp, **kw)
        st = Stats()
        cam = self.camera_struct()
        _check(hip_lib().rtr_render_pt_device_noalloc(self.device_scene(), C.byref(cam), C.byref(p),
                                                      C.c_void_p(d_ptr), C.byref(st)),
               "rtr_render_pt_device_noalloc")
        return st

    def render_via_host_api(self, W, H, spp, file=None, devices=None, **kw):
        """Render::renderPathTracing of the host library (optionally writes file.ppm);
        devices: RenderOptions::devices."""
        p = self.make_params(W, H, spp, **kw)
        devs = [int(d) for d in (devices or [])]
        if self.L.rtrh_set_render_devices(self.h, (C.c_int * max(len(devs), 1))(*devs), len(devs)) != 0:
            self._herr("set_render_devices")
        if devs:
            p.tile_cols = kw.get("tile_cols", 0)
        ncol = W if devs else len(partition_columns(W, p.part_index, p.part_count, p.tile_cols))
        rgb = np.zeros((ncol * H, 3), f32)
        st = Stats()
        rc = self.L.rtrh_render(self.h, W, H, spp, file.encode() if file else None, C.byref(p),
                                _f(rgb), C.byref(st))
        if rc != 0:
            self._herr("render")
        return rgb, st


def default_scene():
    return HostScene.from_spec(default_scene_spec())
