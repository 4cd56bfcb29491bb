"""ctypes loader of ``libpcr_hip.so`` (C ABI: ``include/pcr_hip.h``).

There is NO CPU fallback: if the shared library is missing, or no HIP device is visible, every
compute entry point raises ``RuntimeError`` (the CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PCR_HIP_SO") or os.path.join(_HERE, "libpcr_hip.so")      # (PCR_HIP_SO: diagnostic builds of tools/build_variant.sh)

PCR_OK, PCR_EINVAL, PCR_ENOMEM, PCR_EHIP, PCR_ENUMERIC, PCR_ECAPACITY = 0, -1, -2, -3, -4, -5
SEARCH_KNN, SEARCH_RADIUS, SEARCH_HYBRID = 0, 1, 2
LOSS_L2, LOSS_L1, LOSS_GM = 0, 1, 2


class PcrResult(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("fitness", C.c_double), ("inlier_rmse", C.c_double),
                ("n_correspondences", C.c_int64), ("iterations", C.c_int32), ("converged", C.c_int32)]


class PcrGicpParams(C.Structure):
    _fields_ = [("loss", C.c_int32), ("loss_k", C.c_double), ("epsilon", C.c_double), ("relative_fitness", C.c_double),
                ("relative_rmse", C.c_double), ("max_iteration", C.c_int32)]


class PcrScaleRecord(C.Structure):
    _fields_ = [("n_voxel", C.c_int64 * 2), ("n_clean", C.c_int64 * 2), ("icp", PcrResult)]


class PcrPair(C.Structure):
    _fields_ = [("src_xyz", C.c_void_p), ("src_normals", C.c_void_p), ("n_src", C.c_int64),
                ("tgt_xyz", C.c_void_p), ("tgt_normals", C.c_void_p), ("n_tgt", C.c_int64),
                ("init_T", C.c_double * 16), ("records", C.POINTER(PcrScaleRecord)), ("correspondences", C.c_void_p),
                ("status", C.c_int32), ("error", C.c_char * 120)]


class PcrFgrOption(C.Structure):
    _fields_ = [("division_factor", C.c_double), ("use_absolute_scale", C.c_int32), ("decrease_mu", C.c_int32),
                ("maximum_correspondence_distance", C.c_double), ("iteration_number", C.c_int32),
                ("tuple_scale", C.c_double), ("maximum_tuple_count", C.c_int32), ("tuple_test", C.c_int32),
                ("seed", C.c_uint64)]


class PcrFgrParams(C.Structure):
    _fields_ = [("normal_radius", C.c_double), ("normal_max_nn", C.c_int32), ("feature_radius", C.c_double),
                ("feature_max_nn", C.c_int32), ("option", PcrFgrOption)]


class PcrPairsPlan(C.Structure):
    _fields_ = [("stage", C.c_int32), ("fgr", C.POINTER(PcrFgrParams)), ("voxel_sizes", C.POINTER(C.c_double)),
                ("max_distances", C.POINTER(C.c_double)), ("n_scales", C.c_int32), ("radius_rule", C.c_int32),
                ("sor_k", C.c_int32), ("sor_std", C.c_double), ("normal_k", C.c_int32), ("gicp", C.POINTER(PcrGicpParams)),
                ("gicp_prior_from_fgr", C.c_int32), ("info_max_dist", C.c_double), ("inflight", C.c_int32), ("group", C.c_int32), ("pair_forms", C.c_int32), ("fgr_group", C.c_int32)]


class PcrPairEx(C.Structure):
    _fields_ = [("base", PcrPair), ("fgr", PcrResult), ("src_normals_out", C.c_void_p), ("tgt_normals_out", C.c_void_p),
                ("max_distances", C.c_double * 8), ("info36", C.c_double * 36)]


STAGE_GICP, STAGE_FGR, STAGE_FGR_GICP = 1, 2, 3

# every symbol include/pcr_hip.h declares (tests check the export table against this list)
EXPORTS = [
    "pcr_create", "pcr_destroy", "pcr_set_stream", "pcr_last_error", "pcr_version", "pcr_bounds",
    "pcr_voxel_down_sample", "pcr_remove_statistical_outlier", "pcr_estimate_normals", "pcr_estimate_covariances",
    "pcr_registration_generalized_icp", "pcr_multiscale_gicp", "pcr_evaluate_registration", "pcr_information_matrix",
    "pcr_compute_fpfh_feature", "pcr_registration_fgr", "pcr_debug_knn", "pcr_debug_gicp_linearize",
    "pcr_profile_enable", "pcr_profile_read", "pcr_registration_generalized_icp_cov", "pcr_register_pairs", "pcr_pool_profile",
    "pcr_registro_fgr", "pcr_register_pairs_plan", "pcr_debug_feature_nn", "pcr_set_option", "pcr_counter",
]

_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    env = dict(os.environ)
    if force:
        for f in os.listdir(os.path.join(_HERE, "csrc")):
            if f.endswith(".o"):
                os.remove(os.path.join(_HERE, "csrc", f))
    subprocess.check_call(["bash", script], env=env)
    return SO_PATH


def load():
    """Return the ctypes handle of libpcr_hip.so; raise RuntimeError if it is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(SO_PATH):
                raise RuntimeError(
                    f"libpcr_hip.so not found at {SO_PATH}: the HIP extension is not built "
                    "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
            # torch bundles its own HIP runtime: load it FIRST so that libpcr_hip.so binds to the same one
            # (two HIP runtimes in one process cannot share a device context)
            import torch  # noqa: F401
            lib = C.CDLL(SO_PATH)
            lib.pcr_last_error.restype = C.c_char_p
            lib.pcr_last_error.argtypes = [C.c_void_p]
            lib.pcr_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
            lib.pcr_destroy.argtypes = [C.c_void_p]
            lib.pcr_set_stream.argtypes = [C.c_void_p, C.c_void_p]
            if hasattr(lib, "pcr_set_option"):       # (absent only from older diagnostic builds loaded through PCR_HIP_SO)
                lib.pcr_set_option.argtypes = [C.c_char_p, C.c_longlong]
            _lib = lib
    return _lib


def counter(name: str, reset: bool = False) -> int:
    """Process-wide event counter of the library (``pcr_counter``, include/pcr_hip.h), e.g. ``counter("fgr_group_barrier_timeouts")``."""
    lib = load()
    lib.pcr_counter.argtypes = [C.c_char_p, C.c_int]; lib.pcr_counter.restype = C.c_longlong
    v = int(lib.pcr_counter(name.encode(), int(bool(reset))))
    if v < 0:
        raise ValueError(f"pcr_counter: unknown counter {name!r}")
    return v


def set_option(name: str, value: int) -> None:
    """Test / diagnostic switch of the process (``pcr_set_option``, include/pcr_hip.h), e.g. ``set_option("knn_wave", 1)``."""
    rc = load().pcr_set_option(name.encode(), int(value))
    if rc != PCR_OK:
        raise ValueError(f"pcr_set_option: unknown option {name!r}")


class Context:
    """One libpcr_hip context (scratch arena + stream) per (device, torch stream, thread).

    The cache is bounded (least recently used contexts are destroyed: each holds a device arena of hundreds of MB) and guarded
    by a lock; ``close()`` destroys a context explicitly.  ``stream_ptr == 0`` is torch's default stream = the legacy default
    stream: the library then runs on a stream of its own and fences every call against the default stream on both sides
    (``pcr_set_stream(ctx, NULL)``, include/pcr_hip.h)."""

    _cache: "dict" = {}
    _cache_lock = threading.Lock()
    MAX_CACHED = 16

    def __init__(self, device: int, stream_ptr: int):
        lib = load()
        h = C.c_void_p()
        rc = lib.pcr_create(int(device), C.byref(h))
        if rc != PCR_OK:
            raise RuntimeError(f"pcr_create(device={device}) failed with code {rc}: no usable HIP device "
                               "(the MI355X path has no CPU fallback)")
        self.handle = h
        self.device = device
        self.lib = lib
        lib.pcr_set_stream(h, C.c_void_p(stream_ptr))

    def close(self):
        """Destroy the library context (waits for its stream, frees the arena)."""
        h, self.handle = self.handle, None
        if h is not None and h.value:
            self.lib.pcr_destroy(h)

    @classmethod
    def current(cls) -> "Context":
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to torch: the MI355X registration path cannot run (no CPU fallback)")
        dev = torch.cuda.current_device()
        sp = int(torch.cuda.current_stream(dev).cuda_stream)
        key = (dev, sp, threading.get_ident())
        stale = []
        with cls._cache_lock:
            ctx = cls._cache.pop(key, None)
            if ctx is None or ctx.handle is None:
                ctx = cls(dev, sp)
            cls._cache[key] = ctx                       # most recently used last
            while len(cls._cache) > cls.MAX_CACHED:
                stale.append(cls._cache.pop(next(iter(cls._cache))))
        for c in stale:
            c.close()
        return ctx

    @classmethod
    def close_all(cls):
        with cls._cache_lock:
            ctxs = list(cls._cache.values()); cls._cache.clear()
        for c in ctxs:
            c.close()

    def check(self, rc: int, what: str):
        if rc == PCR_OK:
            return
        msg = self.lib.pcr_last_error(self.handle)
        msg = msg.decode() if msg else ""
        # Open3D raises RuntimeError for invalid arguments; mirror that (SURVEY.md §8b error convention)
        raise RuntimeError(f"{what}: {msg} (pcr status {rc})")
