"""ctypes binding of libhdiff.so (include/hdiff.h).

The product path has no CPU fallback: if the shared library is missing, ``lib()`` raises.  Every wrapper checks the
status code and raises ``RuntimeError`` with ``hdiff_last_error()``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HDIFF_LIB", os.path.join(_HERE, "libhdiff.so"))   # HDIFF_LIB: dev override (A/B builds)
CSRC = os.path.join(_HERE, "csrc")
MAX_TAPS = 25

_lib = None


class LinearJob(C.Structure):
    """hdiff_linear_job (include/hdiff.h); an array of these is copied to device memory once per plan."""
    _fields_ = [("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("y", C.c_void_p),
                ("n", C.c_int), ("first", C.c_int)]


class DdpmLoopDesc(C.Structure):
    """hdiff_ddpm_loop_desc (include/hdiff.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("eps_c", C.c_void_p), ("eps_u", C.c_void_p), ("noise", C.c_void_p), ("x_next", C.c_void_p),
        ("coeff1", C.c_void_p), ("coeff2", C.c_void_p), ("sigma", C.c_void_p), ("step_ptr", C.c_void_p), ("T", C.c_int),
        ("w", C.c_double), ("seed", C.c_uint64), ("nan_flag", C.c_void_p), ("n", C.c_int64),
        ("x_dup0", C.c_void_p), ("x_dup1", C.c_void_p), ("t_next", C.c_void_p), ("t_count", C.c_int),
        ("done_counter", C.c_void_p),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("C0", C.c_int), ("C1", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("wp", C.c_void_p), ("bias", C.c_void_p), ("Cout", C.c_int), ("CinPad", C.c_int), ("CoutPad", C.c_int),
        ("gn_scale", C.c_void_p), ("gn_shift", C.c_void_p), ("addvec", C.c_void_p), ("residual", C.c_void_p),
        ("out", C.c_void_p), ("OH", C.c_int), ("OW", C.c_int),
        ("VH", C.c_int), ("VW", C.c_int), ("in_stride", C.c_int),
        ("out_sy", C.c_int), ("out_oy", C.c_int), ("out_sx", C.c_int), ("out_ox", C.c_int),
        ("ntaps", C.c_int), ("tap_dy", C.c_int * MAX_TAPS), ("tap_dx", C.c_int * MAX_TAPS),
        ("splitk_ws", C.c_void_p), ("splitk_floats", C.c_int64), ("wp_x3", C.c_void_p),
        ("wp_h2", C.c_void_p), ("act_scale", C.c_void_p),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("C0", C.c_int), ("C1", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("gn_scale", C.c_void_p), ("gn_shift", C.c_void_p), ("dy", C.c_void_p),
        ("Cout", C.c_int), ("CinPad", C.c_int), ("CoutPad", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
        ("VH", C.c_int), ("VW", C.c_int), ("in_stride", C.c_int),
        ("out_sy", C.c_int), ("out_oy", C.c_int), ("out_sx", C.c_int), ("out_ox", C.c_int),
        ("ntaps", C.c_int), ("tap_dy", C.c_int * MAX_TAPS), ("tap_dx", C.c_int * MAX_TAPS),
    ]


_PROTOS = {
    "hdiff_abi_version": (C.c_int, []),
    "hdiff_last_error": (C.c_char_p, []),
    "hdiff_device_count": (C.c_int, []),
    "hdiff_pack_conv_weight_x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_pack_conv_weight_x3_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_void_p]),
    "hdiff_pack_conv_weight_h2_words": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "hdiff_pack_conv_weight_h2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_gn_act_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "hdiff_set_contraction_mode": (C.c_int, [C.c_int]),
    "hdiff_get_contraction_mode": (C.c_int, []),
    "hdiff_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "hdiff_conv2d_fwd_workspace": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int64)]),
    "hdiff_gn_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "hdiff_gn_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hdiff_gn_scale_shift": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hdiff_gn_swish_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "hdiff_mha_flash_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_mha_flash_fwd_workspace": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "hdiff_mha_flash_fwd_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_int64, C.c_void_p]),
    "hdiff_mha_wide_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_mha_wide_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_gn_affine_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hdiff_gn_affine_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    "hdiff_mha_flash_bwd_workspace": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "hdiff_mha_flash_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_conv2d_wgrad_workspace": (C.c_int, [C.POINTER(WgradDesc), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "hdiff_conv2d_wgrad": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p, C.c_int, C.c_void_p]),
    "hdiff_conv_wgrad_unpack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_gn_swish_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "hdiff_bias_addvec_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hdiff_linear_rows_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_sq_err_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hdiff_dropout_mask": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "hdiff_mul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hdiff_linear_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_linear_rows_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_q_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p]),
    "hdiff_sq_err": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hdiff_ddpm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p]),
    "hdiff_ddpm_step_loop": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hdiff_fill_t": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hdiff_step_decrement": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hdiff_ddim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                  C.c_void_p]),
    "hdiff_fill_from_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_resize_nearest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_avgpool_global": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hdiff_concat2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "hdiff_clip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_void_p]),
    "hdiff_axpby": (C.c_int, [C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hdiff_opt_chunk": (C.c_int, []),
    "hdiff_grad_norm_clip_coef": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "hdiff_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_int64, C.c_void_p]),
    "hdiff_randn": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "hdiff_graph_begin": (C.c_int, [C.c_void_p]),
    "hdiff_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "hdiff_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hdiff_graph_destroy": (C.c_int, [C.c_void_p]),
    "hdiff_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "hdiff_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hdiff_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "hdiff_event_destroy": (C.c_int, [C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())


def build(verbose: bool = False, clean: bool = False) -> str:
    """Compile every HIP source for gfx950 into libhdiff.so (hipcc cross-compiles without a GPU).  ``clean`` removes the
    object files and the library first, so that the call proves compilation of every source (about 15 s with 8 jobs),
    not just a link of objects that travelled with the tree."""
    if clean:
        subprocess.run(["make", "-C", CSRC, "clean"], capture_output=True, text=True, check=True)
    res = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0 or not os.path.isfile(LIB_PATH):
        raise RuntimeError("building libhdiff.so failed")
    return LIB_PATH


MUTANT_PATH = os.path.join(CSRC, "build", "libhdiff_mutant.so")
MUTANT2_PATH = os.path.join(CSRC, "build", "libhdiff_mutant2.so")


def build_mutant(verbose: bool = False) -> str:
    """Test infrastructure (tests/test_gpu_mutation.py): the library with ONE low-order piece product damaged in each
    split-operand kernel (csrc/common.h lists the HDIFF_MUTANT bits), and a second one with the third-piece terms of dS in the
    attention backward alone.  Never loaded by the product."""
    res = subprocess.run(["make", "-C", CSRC, "-j4", "mutant", "mutant2"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0 or not os.path.isfile(MUTANT_PATH) or not os.path.isfile(MUTANT2_PATH):
        raise RuntimeError("building libhdiff_mutant.so / libhdiff_mutant2.so failed")
    return MUTANT_PATH


def lib() -> C.CDLL:
    """Load libhdiff.so; raise loudly if it is not built (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is required; there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (the owner of the device memory and
        # streams we are handed), so it must be resident before libhdiff.so resolves its libamdhip64.so.7 dependency --
        # otherwise the system runtime is loaded beside it and launches fail with "no ROCm-capable device".
        import torch
        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.isfile(hip_rt):
            C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)       # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().hdiff_last_error().decode(errors="replace")
        raise RuntimeError(f"hdiff {what} failed (status {rc}): {msg}")
