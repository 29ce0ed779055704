"""ctypes binding of libdcamd.so (C-ABI declared in include/dcamd.h).

The product path has NO fallback: if the HIP library is missing or does not load, every
entry point raises.  Nothing here imports `oracle/`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCAMD_LIB") or os.path.join(_HERE, "libdcamd.so")   # DCAMD_LIB: diagnostic builds only

ABI_VERSION = 4      # include/dcamd.h DC_ABI_VERSION
DC_F32, DC_BF16, DC_F16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_GELU_TANH = 0, 1, 2, 3
OP_QSAMPLE, OP_SINUSOID, OP_IGEMM, OP_GROUPNORM, OP_LAYERNORM, OP_ATTENTION, OP_EPS_MSE, OP_TBLOCK_FRONT = 1, 2, 3, 4, 5, 6, 7, 8

i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p


class QsampleParams(C.Structure):
    _fields_ = [("x", vp), ("eps", vp), ("alpha", vp), ("sigma", vp), ("img_of_bj", vp),
                ("out", vp), ("out_dtype", i32),
                ("n_bj", i32), ("C", i32), ("H", i32), ("W", i32), ("ld", i32), ("im2col", i32), ("patch", i32)]


class SinusoidParams(C.Structure):
    _fields_ = [("lam", vp), ("out", vp), ("n", i32), ("dim", i32), ("flip_sin_to_cos", i32), ("freq_shift", f32)]


class IgemmParams(C.Structure):
    _fields_ = [("dtype", i32), ("taps", i32), ("stride", i32), ("upsample", i32),
                ("n_img", i32), ("Hin", i32), ("Win", i32), ("Hout", i32), ("Wout", i32),
                ("src0", vp), ("map0", vp), ("C0", i32), ("ld0", i32),
                ("src1", vp), ("map1", vp), ("C1", i32), ("ld1", i32),
                ("W", vp), ("Cout", i32), ("tile_n", i32),
                ("bias", vp),
                ("rowvec", vp), ("rowvec_map", vp), ("rowvec_ld", i32), ("act", i32),
                ("gate", vp), ("gate_map", vp), ("gate_ld", i32), ("pad2_", i32),
                ("residual", vp), ("res_map", vp), ("res_dtype", i32), ("res_ld", i32),
                ("out", vp), ("out_dtype", i32), ("out_ld", i32),
                ("gn_scale", vp), ("gn_shift", vp), ("gn_silu", i32), ("pad3_", i32),
                ("src2", vp), ("map2", vp), ("W2", vp), ("C2", i32), ("ld2", i32),
                ("ln_eps", f32), ("up4", i32), ("qstats", vp),
                ("pn_out", vp), ("pn_gamma", vp), ("pn_beta", vp), ("pn_cnt", vp),
                ("pn_ld", i32), ("pn_groups", i32), ("pn_silu", i32), ("pn_eps", f32)]


class GroupnormParams(C.Structure):
    _fields_ = [("x", vp), ("map0", vp), ("x1", vp), ("map1", vp),
                ("y", vp), ("dtype", i32), ("out_dtype", i32),
                ("n", i32), ("HW", i32), ("C", i32), ("C1", i32), ("groups", i32), ("silu", i32),
                ("splits", i32), ("eps", f32),
                ("gamma", vp), ("beta", vp), ("ws", vp), ("out_scale", vp), ("out_shift", vp),
                ("qstats", vp), ("qparts", i32), ("pad_", i32)]


class LayernormParams(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("dtype", i32), ("out_dtype", i32),
                ("rows", i32), ("C", i32), ("rows_per_sample", i32), ("mod_ld", i32), ("eps", f32),
                ("gamma", vp), ("beta", vp), ("scale", vp), ("shift", vp), ("mod_map", vp)]


class AttentionParams(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp),
                ("dtype", i32), ("n", i32), ("L", i32), ("heads", i32), ("d", i32),
                ("ld_qkv", i32), ("ld_out", i32), ("scale", f32)]


class TblockFrontParams(C.Structure):
    _fields_ = [("x", vp), ("Wp", vp), ("bp", vp), ("ln_g", vp), ("ln_b", vp), ("Wqkv", vp), ("Wo", vp), ("bo", vp),
                ("rowvec", vp), ("rowvec_map", vp), ("out", vp),
                ("dtype", i32), ("n", i32), ("L", i32), ("C", i32), ("heads", i32), ("ldx", i32), ("ld_out", i32), ("rowvec_ld", i32),
                ("ln_eps", f32), ("scale", f32)]


class EpsMseParams(C.Structure):
    _fields_ = [("pred", vp), ("eps", vp), ("x", vp), ("alpha", vp), ("sigma", vp),
                ("bj_of_unit", vp), ("img_of_bj", vp), ("out_index", vp),
                ("out", vp), ("n_units", i32), ("C", i32), ("H", i32), ("W", i32), ("ld", i32), ("v_param", i32), ("patch", i32)]


class DdpmStepParams(C.Structure):
    _fields_ = [("z", vp), ("pred", vp), ("noise", vp), ("out", vp),
                ("n", i32), ("C", i32), ("H", i32), ("W", i32), ("ld", i32), ("patch", i32), ("v_param", i32),
                ("w", f32), ("alpha_t", f32), ("sigma_t", f32), ("alpha_s", f32), ("c", f32), ("sd", f32), ("one_plus_w", f32)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("pad_", i32), ("params", vp)]


# every symbol include/dcamd.h declares (tests check that the library exports all of them)
EXPORTS = ["dc_abi_version", "dc_last_error", "dc_arch", "dc_qsample", "dc_philox_normal", "dc_sinusoid",
           "dc_igemm", "dc_igemm_cout_pad", "dc_igemm_variant", "dc_igemm_gn_fusable", "dc_igemm_side_ok", "dc_igemm_ln_ok", "dc_igemm_qstats_parts", "dc_igemm_up4_ok", "dc_igemm_pn_ok", "dc_pn_timeouts", "dc_groupnorm", "dc_groupnorm_ws_floats", "dc_groupnorm_splits",
           "dc_layernorm", "dc_attention", "dc_tblock_front", "dc_tblock_front_ok", "dc_eps_mse", "dc_ddpm_step", "dc_haar_dwt2", "dc_haar_idwt2", "dc_stage_topk", "dc_reduce_argmin", "dc_stage_maps", "dc_run_plan", "dc_run_plan_timed",
           "dc_packed_bytes", "dc_pack_weights_matrix", "dc_pack_weights_conv3x3", "dc_pack_weights_up4", "dc_pack_weights_geglu",
           "dc_fold_layernorm_bias", "dc_workspace_bytes_groupnorm", "dc_workspace_bytes_igemm", "dc_workspace_bytes_attention",
           "dc_workspace_bytes_layernorm"]

_lib = None


class DcamdError(RuntimeError):
    pass


def lib():
    """Load libdcamd.so once.  Raises (never falls back) when the extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DcamdError(
            f"HIP extension {LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C diffusion-classifier_amd/csrc`). There is no CPU fallback for the scoring path.")
    L = C.CDLL(LIB_PATH)
    L.dc_abi_version.restype = i32
    L.dc_last_error.restype = C.c_char_p
    L.dc_arch.restype = C.c_char_p
    for name, argt in [("dc_qsample", [C.POINTER(QsampleParams), vp]),
                       ("dc_sinusoid", [C.POINTER(SinusoidParams), vp]),
                       ("dc_igemm", [C.POINTER(IgemmParams), vp]),
                       ("dc_groupnorm", [C.POINTER(GroupnormParams), vp]),
                       ("dc_layernorm", [C.POINTER(LayernormParams), vp]),
                       ("dc_attention", [C.POINTER(AttentionParams), vp]),
                       ("dc_tblock_front", [C.POINTER(TblockFrontParams), vp]),
                       ("dc_eps_mse", [C.POINTER(EpsMseParams), vp]),
                       ("dc_ddpm_step", [C.POINTER(DdpmStepParams), vp]),
                       ("dc_run_plan", [C.POINTER(Op), i32, vp]),
                       ("dc_run_plan_timed", [C.POINTER(Op), i32, vp, vp]),
                       ("dc_philox_normal", [vp, i64, i64, vp, u64, vp]),
                       ("dc_haar_dwt2", [vp, vp, i32, i32, i32, i32, f32, vp]),
                       ("dc_haar_idwt2", [vp, vp, i32, i32, i32, i32, f32, vp]),
                       ("dc_pack_weights_matrix", [vp, i32, i32, i32, vp, vp, vp, i32, i32, vp]),
                       ("dc_pack_weights_conv3x3", [vp, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
                       ("dc_pack_weights_up4", [vp, i32, i32, vp, i32, i32, vp]),
                       ("dc_pack_weights_geglu", [vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
                       ("dc_fold_layernorm_bias", [vp, vp, vp, i32, i32, vp, vp]),
                       ("dc_stage_topk", [vp, i32, i32, i32, i32, i32, vp, vp, vp]),
                       ("dc_reduce_argmin", [vp, i32, i32, i32, i32, vp, vp, vp]),
                       ("dc_stage_maps", [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp])]:
        fn = getattr(L, name)
        fn.argtypes = argt
        fn.restype = i32
    L.dc_igemm_ln_ok.argtypes = [C.POINTER(IgemmParams)]
    L.dc_igemm_ln_ok.restype = C.c_int32
    L.dc_igemm_qstats_parts.argtypes = [C.POINTER(IgemmParams)]
    L.dc_igemm_qstats_parts.restype = C.c_int32
    L.dc_igemm_up4_ok.argtypes = [C.POINTER(IgemmParams)]
    L.dc_igemm_up4_ok.restype = C.c_int32
    L.dc_igemm_pn_ok.argtypes = [C.POINTER(IgemmParams)]
    L.dc_igemm_pn_ok.restype = C.c_int32
    L.dc_tblock_front_ok.argtypes = [C.POINTER(TblockFrontParams)]
    L.dc_tblock_front_ok.restype = C.c_int32
    L.dc_pn_timeouts.argtypes = []
    L.dc_pn_timeouts.restype = C.c_int32
    L.dc_igemm_side_ok.argtypes = [C.POINTER(IgemmParams)]
    L.dc_igemm_side_ok.restype = C.c_int32
    L.dc_igemm_variant.argtypes = [C.POINTER(IgemmParams)]
    L.dc_igemm_variant.restype = C.c_char_p
    L.dc_igemm_gn_fusable.argtypes = [C.POINTER(IgemmParams)]
    L.dc_igemm_gn_fusable.restype = i32
    L.dc_igemm_cout_pad.argtypes = [i32, i32]
    L.dc_igemm_cout_pad.restype = i32
    L.dc_packed_bytes.argtypes = [i32, i32, i32, i32]
    L.dc_packed_bytes.restype = i64
    for name, pt in (("dc_workspace_bytes_groupnorm", GroupnormParams), ("dc_workspace_bytes_igemm", IgemmParams),
                     ("dc_workspace_bytes_attention", AttentionParams), ("dc_workspace_bytes_layernorm", LayernormParams)):
        getattr(L, name).argtypes = [C.POINTER(pt)]
        getattr(L, name).restype = i64
    L.dc_groupnorm_ws_floats.argtypes = [i32, i32, i32]
    L.dc_groupnorm_ws_floats.restype = i64
    L.dc_groupnorm_splits.argtypes = [i32, i32, i32]
    L.dc_groupnorm_splits.restype = i32
    if L.dc_abi_version() != ABI_VERSION:
        raise DcamdError(f"libdcamd ABI {L.dc_abi_version()} != {ABI_VERSION} (stale libdcamd.so? rebuild with `make -C diffusion-classifier_amd/csrc`)")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise DcamdError(f"{what} failed (status {rc}): {lib().dc_last_error().decode()}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise DcamdError("the diffusion-classifier scoring path needs an MI355X (HIP device); no CPU fallback exists")
    return lib()
