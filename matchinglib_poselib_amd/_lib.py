"""ctypes binding of libmlpl_hip.so (the C ABI in include/mlpl_c.h).

The library is built in-tree by `python -m matchinglib_poselib_amd.build` (or __graft_entry__.build()).
Loading fails loudly when the shared object is missing: there is no Python/CPU fallback for any op.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MLPL_LIB_PATH") or os.path.join(_HERE, "lib", "libmlpl_hip.so")   # (MLPL_LIB_PATH: tools A/B two BUILDS of the library)

MLPL_OK = 0
MLPL_E_BAD_INPUT = -1
MLPL_E_UNSUPPORTED = -2
MLPL_E_FAILED = -3
MLPL_E_INTERNAL = -103
MLPL_E_FEW_KEYPOINTS = -4
MLPL_E_NO_DEVICE = -100
MLPL_E_HIP = -101
MLPL_E_NOMEM = -102


class MlplError(RuntimeError):
    def __init__(self, code: int, where: str, msg: str = ""):
        self.code = code
        super().__init__(f"{where} failed with code {code}: {msg}")


def library_path() -> str:
    return _LIB_PATH


_lib = None
_lib_lock = threading.Lock()

c_void_p, c_int, c_size_t, c_float, c_double, c_u32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double, C.c_uint32

# name -> (restype, argtypes); must list EVERY symbol declared in include/mlpl_c.h (tests check this)
_SIGNATURES = {
    "mlpl_ctx_create": (c_int, [c_int, C.POINTER(c_void_p)]),
    "mlpl_ctx_destroy": (None, [c_void_p]),
    "mlpl_last_error": (C.c_char_p, []),
    "mlpl_version": (C.c_char_p, []),
    "mlpl_device_count": (c_int, []),
    "mlpl_ctx_stream": (c_void_p, [c_void_p]),
    "mlpl_ctx_device": (c_int, [c_void_p]),
    "mlpl_ctx_synchronize": (c_int, [c_void_p]),
    "mlpl_set_option": (c_int, [c_void_p, C.c_char_p, c_int]),
    "mlpl_get_option": (c_int, [c_void_p, C.c_char_p, C.POINTER(c_int)]),
    "mlpl_profile_enable": (c_int, [c_void_p, c_int]),
    "mlpl_profile_reset": (c_int, [c_void_p]),
    "mlpl_profile_read": (c_int, [c_void_p, c_int, C.POINTER(c_double), C.POINTER(c_int)]),
    "mlpl_knn2_hamming": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_int, c_size_t, c_int, c_int,
                                  c_void_p, c_void_p]),
    "mlpl_knn2_l2sq_f32": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_int, c_size_t, c_int, c_int,
                                   c_void_p, c_void_p]),
    "mlpl_knn2_hamming_dev": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_size_t, c_void_p, c_int, c_size_t,
                                      c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mlpl_knn2_l2sq_f32_dev": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_size_t, c_void_p, c_int, c_size_t,
                                       c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mlpl_set_l2_path": (c_int, [c_void_p, c_int]),
    "mlpl_debug_l2_flags": (c_int, [c_void_p, c_void_p]),
    "mlpl_ratio_compact_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                       C.POINTER(c_int)]),
    "mlpl_ratio_compact_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                       C.POINTER(c_int)]),
    "mlpl_ratio_compact_i32_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                           c_void_p, c_void_p]),
    "mlpl_ratio_compact_f32_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                           c_void_p, c_void_p]),
    "mlpl_get_matches_linear": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_size_t, c_void_p, c_int, c_size_t,
                                        c_int, c_int, c_int, c_void_p, C.POINTER(c_int)]),
    "mlpl_get_matches_bruteforce_nms": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_size_t, c_void_p, c_int,
                                                c_size_t, c_int, c_int, c_int, c_void_p, C.POINTER(c_int)]),
    "mlpl_match_hamming_dev": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_size_t, c_void_p, c_int, c_size_t,
                                       c_size_t, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "mlpl_gather_match_points_dev": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p]),
    "mlpl_img_to_cam": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "mlpl_remove_lens_dist": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.POINTER(c_int)]),
    "mlpl_get_inliers_strict": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_double, c_void_p, c_void_p]),
    "mlpl_count_models": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_double, c_int, c_void_p]),
    "mlpl_median_models": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "mlpl_lmeds_essential": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_u32, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "mlpl_lmeds_essential_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_u32, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "mlpl_arrsac_essential": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mlpl_arrsac_essential_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    "mlpl_arrsac_essential_batch_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p]),
    "mlpl_arrsac_last_stats": (c_int, [c_void_p, c_void_p]),
    "mlpl_arrsac_sample_models": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "mlpl_robust_essential_refine": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "mlpl_debug_arrsac_trace": (c_int, [c_void_p, c_void_p, c_int]),
    "mlpl_usac_default_params": (None, [c_void_p, c_double]),
    "mlpl_usac_essential": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mlpl_usac_essential_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mlpl_usac_essential_batch_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "mlpl_usac_last_stats": (c_int, [c_void_p, c_void_p]),
    "mlpl_usac_last_degeneracy": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "mlpl_debug_usac_trace": (c_int, [c_void_p, c_void_p, c_int]),
    "mlpl_ransac_essential": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_int, c_int, c_u32,
                                      c_void_p, c_void_p, C.POINTER(c_int), C.POINTER(c_int)]),
    "mlpl_ransac_essential_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_int, c_int,
                                          c_u32, c_void_p, c_void_p, C.POINTER(c_int), C.POINTER(c_int), c_void_p]),
    "mlpl_solve_5pt": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "mlpl_score_models": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_double, c_void_p,
                                  c_void_p]),
    "mlpl_ransac_last_stats": (c_int, [c_void_p, c_void_p]),
    "mlpl_debug_ransac_draw": (c_int, [c_void_p, c_void_p]),
    "mlpl_debug_fiber_selftest": (C.c_longlong, [c_int, c_int, c_int]),
    "mlpl_debug_eig9": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "mlpl_debug_dk_stats": (c_int, [c_void_p, c_int, c_void_p]),
    "mlpl_debug_hamming_stamps": (c_int, [c_void_p, c_void_p, c_int]),
    "mlpl_debug_hamming_clock": (c_int, [c_void_p, c_void_p, c_int]),
    "mlpl_debug_hop_trace": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mlpl_pair_pose_dev": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                   c_int, c_double, c_int, c_u32, c_double, c_void_p, c_void_p]),
    "mlpl_pair_pose_batch_dev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                         c_int, c_double, c_int, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "mlpl_pair_pose_batch_lanes_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_double, c_int, c_double, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "mlpl_pair_pose_batch_usac_dev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                              c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "mlpl_pair_pose_batch_arrsac_dev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int,
                                                c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "mlpl_sorted_match_idx": (c_int, [c_void_p, c_int, c_void_p]),
    "mlpl_pair_batch_last_stats": (c_int, [c_void_p, c_void_p]),
    "mlpl_ransac_essential_batch_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_double, c_int, c_double, c_void_p, c_int, c_double,
                                        c_void_p, c_void_p, c_void_p]),
    "mlpl_recover_pose_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "mlpl_recover_pose_translation": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    "mlpl_recover_pose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
}


def _preload_torch_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME libamdhip64.so.7).
    If ours (/opt/rocm) were loaded first, torch would later load a second copy and find "No HIP GPUs"; device
    pointers and streams could not be shared either.  So when torch is installed, map its runtime first; our
    library's NEEDED libamdhip64.so.7 then resolves to that same object."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """dlopen libmlpl_hip.so and bind every entry point. Raises if the library has not been built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise MlplError(
                MLPL_E_NO_DEVICE,
                "load_library",
                f"{_LIB_PATH} not found -- build it with `python -m matchinglib_poselib_amd.build`; "
                "there is no CPU fallback",
            )
        _preload_torch_hip_runtime()
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = symbol missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error() -> str:
    return load_library().mlpl_last_error().decode("utf-8", "replace")


def check(rc: int, where: str) -> int:
    if rc != MLPL_OK:
        raise MlplError(rc, where, last_error())
    return rc


class Context:
    """Owns one mlpl_ctx (device workspace + private stream). One per host thread."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = c_void_p()
        rc = self._lib.mlpl_ctx_create(int(device), C.byref(h))
        if rc != MLPL_OK:
            raise MlplError(rc, "mlpl_ctx_create", last_error())
        self._h = h
        self.device = int(device)
        # MLPL_OPTIONS="name=value,...": the knobs of mlpl_set_option for every context of the process (the C++ facade reads the same
        # variable); tools use it for A/B runs of unchanged scripts.  Words without "=" are the facade's own switches.
        for kv in os.environ.get("MLPL_OPTIONS", "").split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                self.set_option(k.strip(), int(v))

    @property
    def handle(self):
        return self._h

    @property
    def lib(self):
        return self._lib

    def stream(self) -> int:
        return int(self._lib.mlpl_ctx_stream(self._h) or 0)

    def synchronize(self):
        check(self._lib.mlpl_ctx_synchronize(self._h), "mlpl_ctx_synchronize")

    def set_option(self, name: str, value: int):
        """mlpl_set_option (include/mlpl_c.h): kernel selection / tuning knobs of this context."""
        check(self._lib.mlpl_set_option(self._h, name.encode(), int(value)), "mlpl_set_option")

    def get_option(self, name: str) -> int:
        """mlpl_get_option: the current value of a tuning knob."""
        v = C.c_int(0)
        check(self._lib.mlpl_get_option(self._h, name.encode(), C.byref(v)), "mlpl_get_option")
        return v.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mlpl_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_default_ctx = {}
_default_lock = threading.Lock()


def default_context(device: int = 0) -> Context:
    key = (threading.get_ident(), int(device))
    with _default_lock:
        ctx = _default_ctx.get(key)
        if ctx is None:
            ctx = Context(device)
            _default_ctx[key] = ctx
        return ctx
