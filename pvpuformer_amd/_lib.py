"""ctypes binding of libvpu_hip.so (C ABI declared in include/vpu_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module raises at
import / first use.  (The CPU oracle under ``oracle/`` is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VPU_LIB_DIAG=1: the diagnostic build (csrc/build.sh diag: -DVPU_DIAG, time stamps in the K2 / K4P kernels) for tools/k2_stamps.py and
# tools/k4_drift.py; the product library carries no stamp code
# (VPU_LIB_FILE: an experiment library built by `csrc/build.sh x` for a same-box A/B; never set by the product or the tests)
LIB_PATH = os.path.join(_HERE, os.environ.get("VPU_LIB_FILE") or
                        ("libvpu_hip_diag.so" if os.environ.get("VPU_LIB_DIAG", "0") == "1" else "libvpu_hip.so"))

BF16, F32 = 0, 1

EPI_BIAS, EPI_PREACT, EPI_GELU, EPI_RELU, EPI_DGELU, EPI_DRELU = 1, 2, 4, 8, 16, 32
EPI_RESID, EPI_AFFINE, EPI_ACCUM, EPI_OUT_F32 = 64, 128, 256, 512
EPI_SAVE_DGELU, EPI_MULAUX = 1024, 2048


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("resid", C.c_void_p),
        ("aux", C.c_void_p), ("preact", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32), ("ldaux", C.c_int32),
        ("batch", C.c_int32), ("inner", C.c_int32),
        ("sAo", C.c_int64), ("sAi", C.c_int64), ("sBo", C.c_int64), ("sBi", C.c_int64),
        ("sCo", C.c_int64), ("sCi", C.c_int64), ("sRo", C.c_int64), ("sRi", C.c_int64),
        ("transA", C.c_int32), ("transB", C.c_int32), ("dtype", C.c_int32), ("flags", C.c_int32),
        ("resid_period", C.c_int32),
        ("alpha", C.c_float), ("post_mul", C.c_float), ("post_add", C.c_float),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("colsum", C.c_void_p),
        ("cs_tn", C.c_int32), ("cs_t0", C.c_int32), ("cs_ld", C.c_int64),
    ]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (every function returns int unless listed in _RET)
class ColsumJob(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("nrows", C.c_int32), ("ncols", C.c_int32),
                ("row_len", C.c_int32), ("out_ld", C.c_int32)]


class CastJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src2", C.c_void_p), ("dst", C.c_void_p), ("ld_src", C.c_int64), ("ld_dst", C.c_int64),
                ("rows", C.c_int64), ("cols", C.c_int32), ("cols_pad", C.c_int32), ("dst_dtype", C.c_int32),
                ("perm_g", C.c_int32), ("perm_wg", C.c_int32)]


SIGNATURES = {
    "vpu_gemm": [C.POINTER(GemmDesc), _P],
    "vpu_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P],
    "vpu_layernorm_bwd_nblk": [_L],
    "vpu_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "vpu_layernorm_fwd_pe": [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _L, _P, _I, _P],
    "vpu_layernorm_bwd2": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "vpu_colsum_batched": [C.POINTER(ColsumJob), _I, _P],
    "vpu_colsum_f32": [_P, _P, _L, _I, _F, _P],
    "vpu_colsum": [_P, _I, _P, _P, _L, _I, _F, _I, _P],
    "vpu_softmax_fwd": [_P, _I, _P, _I, _L, _I, _I, _P],
    "vpu_softmax_bwd": [_P, _I, _P, _I, _P, _L, _I, _F, _I, _P],
    "vpu_l2norm_fwd": [_P, _P, _P, _L, _I, _I, _P],
    "vpu_l2norm_bwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "vpu_attn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vpu_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vpu_xattn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vpu_xattn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vpu_xattn_fwd_split": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "vpu_xattn_bwd_split": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "vpu_attn_combine": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vpu_sum_groups": [_P, _P, _L, _I, _L, _I, _P],
    "vpu_add_bcast": [_P, _P, _P, _L, _L, _I, _P],
    "vpu_add4": [_P, _P, _P, _P, _P, _L, _I, _P],
    "vpu_cast2d": [_P, _I, _L, _P, _I, _L, _L, _I, _I, _P],
    "vpu_cast2d_batched": [C.POINTER(CastJob), _I, _P],
    "vpu_fanout_add": [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I, _L, _I, _P],
    "vpu_dropout_mask": [_P, _I, _F, C.c_uint64, _P, _P],
    "vpu_fill_f32": [_P, _F, _L, _P],
    "vpu_fill_ranges_f32": [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I, _F, _P],
    "vpu_debug_spin": [_P, _I, _L, _P],
    "vpu_debug_gemm_times": [_P],
    "vpu_sigmoid_to_channel": [_P, _P, _I, _L, _I, _I, _P],
    "vpu_act_bwd": [_P, _L, _P, _L, _P, _L, _L, _I, _I, _I, _P],
    "vpu_pue_encode": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpu_cc_roots": [_P, _P, _I, _I, _I, _P],
    "vpu_cc_table": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vpu_edt": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vpu_chamfer5": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vpu_pue_scribble_rows": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpu_draw_polyline": [_P, _P, _I, _I, _I, _I, _P],
    "vpu_mask_bbox": [_P, _F, _P, _I, _P, _I, _I, _I, _P],
    "vpu_masked_argmax": [_P, _P, _P, _I, _I, _I, _P],
    "vpu_error_masks": [_P, _P, _P, _P, _I, _I, _P],
    "vpu_disk_maps": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "vpu_patch_im2col": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpu_patch_im2col_prenorm": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpu_window_permute": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpu_pixel_shuffle2": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpu_pixel_shuffle2_gn_stats": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpu_groupnorm_apply": [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _F, _I, _I, _P],
    "vpu_pixel_unshuffle2_nblk": [_I],
    "vpu_pixel_unshuffle2_sums": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpu_groupnorm_nchunk": [],
    "vpu_groupnorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _F, _I, _I, _P],
    "vpu_groupnorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P],
    "vpu_bilinear_cl_fwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vpu_bilinear_cl_bwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vpu_head_grad_fused": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _P],
    "vpu_upsum_relu": [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _I, _I, _I, _I, _P],
    "vpu_gate_stats": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpu_gate_apply": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vpu_gate_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpu_gate_fwd_n": [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P, C.POINTER(C.c_void_p), _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpu_gate_bwd_n": [C.POINTER(C.c_void_p), _P, _P, _P, _P, _P, _P, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P, _I, _I, _I,
                       _I, _I, _I, _P],
    "vpu_convseg_fwd": [_P, _P, _P, _P, _P, _L, _L, _I, _I, _P],
    "vpu_convseg_bwd_nblk": [_L],
    "vpu_convseg_bwd": [_P, _P, _P, _P, _P, _I, _P, _P, _L, _L, _I, _I, _P],
    "vpu_upsample_ac_fwd": [_P, _P, _L, _I, _I, _I, _I, _P],
    "vpu_upsample_ac_bwd": [_P, _P, _L, _I, _I, _I, _I, _P],
    "vpu_p2cl_fwd_bwd": [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P],
    "vpu_p2cl_up_nband": [_I, _I],
    "vpu_p2cl_up_fwd_bwd": [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P],
    "vpu_nfl_dice_scratch_doubles": [_I],
    "vpu_nfl_dice_fwd_bwd": [_P, _P, _P, _P, _P, _F, _F, _I, _L, _P],
    "vpu_loss_finalize": [_P, _P, _I, _I, C.c_double, _F, _F, _F, _F, _P, _P],
    "vpu_adam_step": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    "vpu_adam_step_groups": [_P, _P, _P, _P, _P, _L, _P, _P, _P, _I, _F, _F, _F, _I, _I, _F, _P],
    "vpu_adam_step_hyper": [_P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _F, _F, _F, _F, _I, _P],
    "vpu_gemm_set_option": [C.c_char_p, _I],
    "vpu_gemm_get_option": [C.c_char_p, C.POINTER(C.c_int32)],
    "vpu_gemm_last_kernel": [],
    "vpu_attn_last_kernel": [],
    "vpu_attn_set_option": [C.c_char_p, _I],
    "vpu_gemm_grouped": [C.POINTER(GemmDesc), _I, _P],
    "vpu_last_error": [],
    "vpu_abi_version": [],
}
_RET = {"vpu_last_error": C.c_char_p, "vpu_gemm_last_kernel": C.c_char_p, "vpu_attn_last_kernel": C.c_char_p}

_lib = None


class VpuError(RuntimeError):
    pass


def load():
    """Loads the HIP extension; raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VpuError(f"{LIB_PATH} not found: build it with pvpuformer_amd/csrc/build.sh "
                       f"(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # torch must be imported first: it ships its own libamdhip64 and a process must hold exactly ONE HIP runtime -- if
    # this library is dlopen'ed before torch it binds /opt/rocm's copy and every launch on a torch stream fails with
    # "no ROCm-capable device is detected" (seen in round 1).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RET.get(name, C.c_int)
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise VpuError(f"{name} failed ({rc}): {lib.vpu_last_error().decode()}")
    return rc
