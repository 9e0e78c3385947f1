"""ctypes binding of libusdm_hip.so (C-ABI declared in include/usdm_hip.h).

No fallback: if the shared library is missing or an entry point is absent this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libusdm_hip.so")


class UsdmError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -m usdm_amd.build` (hipcc --offload-arch=gfx950). "
        "usdm_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)      # (global: the experimental library resolves usdm_set_error here)
lib.usdm_last_error.restype = C.c_char_p
lib.usdm_gemv_batch_ks_floats.restype = C.c_int64
_exp = None


def exp():
    """libusdm_hip_experimental.so (include/usdm_hip_experimental.h): kernels that measured slower than the default path, kept out of
    the product library; loaded on first use by ops.gemv_chain / ops.gemv_engine only."""
    global _exp
    if _exp is None:
        path = os.path.join(_HERE, "libusdm_hip_experimental.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: build it with `python -m usdm_amd.build`")
        e = C.CDLL(path)
        n = e.usdm_sizeof_gemv_chain_args()
        if n != C.sizeof(GemvChainArgs):
            raise ImportError(f"ABI mismatch: usdm_gemv_chain_args is {n} bytes in the library, {C.sizeof(GemvChainArgs)} in Python")
        _exp = e
    return _exp

BF16, F32 = 0, 1
ACT_NONE, ACT_GELU, ACT_SWIGLU, ACT_TANH, ACT_LOGCLAMP = 0, 1, 3, 4, 5
EPI_PLAIN, EPI_QKV_HEADS = 0, 1


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("taps", C.c_int32), ("Kc", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64), ("rowsA", C.c_int32),
        ("a_row_mul", C.c_int32), ("a_row_off", C.c_int32), ("a_row_step", C.c_int32),
        ("a_tap_stride", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("groups", C.c_int32), ("batch", C.c_int32),
        ("a_gstride", C.c_int64), ("w_gstride", C.c_int64), ("a_bstride", C.c_int64),
        ("c_gcol", C.c_int32), ("c_bstride", C.c_int64),
        ("bias", C.c_void_p), ("alpha", C.c_float), ("act", C.c_int32), ("round_bf16", C.c_int32),
        ("residual", C.c_void_p), ("res_dtype", C.c_int32), ("ldr", C.c_int64),
        ("C32", C.c_void_p), ("C16", C.c_void_p), ("ldc", C.c_int64),
        ("c_row_mul", C.c_int32), ("c_row_off", C.c_int32), ("transpose_out", C.c_int32), ("epi", C.c_int32),
        ("qkv_S", C.c_int32), ("qkv_Spad", C.c_int32), ("qkv_H", C.c_int32), ("qkv_D", C.c_int32),
        ("qkv_q", C.c_void_p), ("qkv_k", C.c_void_p), ("qkv_v", C.c_void_p),
        ("split_k", C.c_int32), ("c_split_stride", C.c_int64),
        ("stats_out", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_nt", C.c_int32), ("ln_mode", C.c_int32), ("ln_C", C.c_int32),
        ("ln_eps", C.c_float), ("ln_c", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("ln_guard", C.c_void_p), ("ln_guard_ratio", C.c_float), ("tile_sel", C.c_int32),
    ]


class NormArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_dtype", C.c_int32), ("ldx", C.c_int64),
        ("res", C.c_void_p), ("res_dtype", C.c_int32), ("ldr", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
        ("rows", C.c_int32), ("C", C.c_int32),
        ("rms", C.c_int32), ("act", C.c_int32), ("round_bf16", C.c_int32), ("premask", C.c_int32),
        ("valid_len", C.c_void_p), ("rows_per_batch", C.c_int32),
        ("out32", C.c_void_p), ("out16", C.c_void_p), ("ldo", C.c_int64),
        ("sum32", C.c_void_p), ("sum16", C.c_void_p), ("lds", C.c_int64),
        ("res2", C.c_void_p), ("n_res2", C.c_int32), ("res2_stride", C.c_int64),
    ]


class SnakeArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("T", C.c_int32), ("C", C.c_int32), ("Creal", C.c_int32), ("L", C.c_int32),
        ("alpha", C.c_void_p), ("beta", C.c_void_p), ("logscale", C.c_int32),
        ("fup", C.c_float * 12), ("fdn", C.c_float * 12),
        ("out32", C.c_void_p), ("out16", C.c_void_p), ("ldo", C.c_int64),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("dh", C.c_int32), ("B", C.c_int32), ("Hq", C.c_int32), ("Hkv", C.c_int32),
        ("Sq", C.c_int32), ("Skv", C.c_int32), ("Skv_alloc", C.c_int32), ("q_pos0", C.c_int32),
        ("alibi_col0_zero", C.c_int32), ("scale", C.c_float), ("head_order", C.c_int32),
        ("q", C.c_void_p), ("q_bs", C.c_int64), ("q_hs", C.c_int64), ("q_rs", C.c_int64),
        ("k", C.c_void_p), ("k_bs", C.c_int64), ("k_hs", C.c_int64), ("k_rs", C.c_int64),
        ("vt", C.c_void_p), ("v_bs", C.c_int64), ("v_hs", C.c_int64), ("v_ds", C.c_int64),
        ("o", C.c_void_p), ("o_bs", C.c_int64), ("o_rs", C.c_int64),
        ("kv_len", C.c_void_p), ("slopes", C.c_void_p), ("window", C.c_int32), ("variant", C.c_int32),
    ]


class VbInputArgs(C.Structure):
    _fields_ = [
        ("ids", C.c_void_p), ("y", C.c_void_p), ("cond", C.c_void_p), ("table", C.c_void_p),
        ("B_in", C.c_int32), ("dup", C.c_int32), ("S", C.c_int32), ("E", C.c_int32), ("F", C.c_int32),
        ("null_id", C.c_int32), ("use_cond", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_dtype", C.c_int32),
    ]


class VbSolverArgs(C.Structure):
    _fields_ = [
        ("vout", C.c_void_p), ("z", C.c_void_p), ("v1", C.c_void_p), ("eps", C.c_void_p), ("cond", C.c_void_p),
        ("z_in", C.c_void_p), ("z_commit", C.c_void_p), ("t_cur", C.c_void_p),
        ("B", C.c_int32), ("F", C.c_int32), ("S", C.c_int32), ("P", C.c_int32), ("cfg", C.c_int32), ("mode", C.c_int32),
        ("t_count", C.c_int32),
        ("gs", C.c_float), ("dt", C.c_float), ("c_eps", C.c_float), ("c_cond", C.c_float), ("t_next", C.c_float),
    ]


class GemvArgs(C.Structure):
    _fields_ = [
        ("W", C.c_void_p), ("ldw", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
        ("x", C.c_void_p), ("norm_w", C.c_void_p), ("eps", C.c_float),
        ("act", C.c_int32), ("round_bf16", C.c_int32),
        ("residual", C.c_void_p), ("y16", C.c_void_p), ("y32", C.c_void_p),
        ("ban", C.c_void_p), ("part_val", C.c_void_p), ("part_idx", C.c_void_p), ("idx_offset", C.c_int32),
        ("x_delta", C.c_void_p), ("x_out", C.c_void_p), ("skip", C.c_void_p),
        ("p2p", C.c_void_p), ("p2p_site", C.c_int32), ("p2p_mode", C.c_int32),
        ("mrg_pm", C.c_void_p), ("mrg_pl", C.c_void_p), ("mrg_po", C.c_void_p), ("mrg_ns", C.c_int32),
        ("cmb_gran", C.c_void_p), ("cmb_err", C.c_void_p), ("cmb_timeout_ms", C.c_int32),
    ]


class GemvChainArgs(C.Structure):
    _fields_ = [("ph", GemvArgs * 4), ("nph", C.c_int32), ("sync", C.c_void_p), ("timeout_ms", C.c_int32), ("gran", C.c_void_p),
                ("norm_nth", C.c_int32 * 4)]


class GemvBatchArgs(C.Structure):
    _fields_ = [
        ("g", GemvArgs), ("nb", C.c_int32), ("x_bs", C.c_int64), ("y_bs", C.c_int64), ("res_bs", C.c_int64), ("part_bs", C.c_int32),
        ("form", C.c_int32), ("ks_part", C.c_void_p), ("ks_cnt", C.c_void_p), ("ks_part_floats", C.c_int64),
    ]


class DecodeState(C.Structure):
    _fields_ = [
        ("next_token", C.c_void_p), ("out_tokens", C.c_void_p), ("step", C.c_void_p), ("pos", C.c_void_p),
        ("max_out", C.c_int32), ("id_offset", C.c_int32), ("advance_pos", C.c_int32), ("batch", C.c_int32),
        ("done", C.c_void_p), ("eos", C.c_void_p),
    ]


class P2pDev(C.Structure):
    _fields_ = [("base", C.c_uint64 * 8), ("rank", C.c_int32), ("world", C.c_int32), ("n_sites", C.c_int32), ("max_elems", C.c_int32),
                ("timeout_ticks", C.c_uint64)]


class SampleParams(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float), ("reserved", C.c_int32), ("seed", C.c_uint64)]


class SampleArgs(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p), ("V", C.c_int32), ("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float),
        ("seed", C.c_uint64), ("probs_out", C.c_void_p), ("dev_params", C.c_void_p), ("logits_bs", C.c_int64),
    ]


class RopeArgs(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("ld", C.c_int64), ("S", C.c_int32), ("pos0", C.c_int32), ("Hq", C.c_int32),
        ("Hkv", C.c_int32), ("ctx_max", C.c_int32), ("max_pos", C.c_int32),
        ("cos", C.c_void_p), ("sin", C.c_void_p),
        ("kcache", C.c_void_p), ("vcache", C.c_void_p), ("vt", C.c_void_p), ("vt_ld", C.c_int64),
    ]


class AttnDecodeArgs(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("pos", C.c_void_p),
        ("Hq", C.c_int32), ("Hkv", C.c_int32), ("ctx_max", C.c_int32), ("NS", C.c_int32), ("scale", C.c_float),
        ("cos", C.c_void_p), ("sin", C.c_void_p),
        ("kcache", C.c_void_p), ("vcache", C.c_void_p),
        ("pm", C.c_void_p), ("pl", C.c_void_p), ("po", C.c_void_p), ("out", C.c_void_p), ("counters", C.c_void_p),
        ("batch", C.c_int32), ("qkv_bs", C.c_int64), ("out_bs", C.c_int64), ("cache_bs", C.c_int64), ("skip", C.c_void_p),
        ("defer_merge", C.c_int32), ("window", C.c_int32), ("cmb_gran", C.c_void_p),
    ]


def check(rc, what=""):
    if rc != 0:
        raise UsdmError(f"{what} failed (rc={rc}): {lib.usdm_last_error().decode()}")


def _selfcheck():
    assert lib.usdm_abi_version() >= 1
    n = lib.usdm_sizeof_gemm_args()
    if n != C.sizeof(GemmArgs):
        raise ImportError(f"ABI mismatch: usdm_gemm_args is {n} bytes in the library, {C.sizeof(GemmArgs)} in Python")
    for name, cls in (("norm", NormArgs), ("snake", SnakeArgs), ("attn", AttnArgs), ("vb_input", VbInputArgs),
                      ("vb_solver", VbSolverArgs), ("gemv", GemvArgs), ("decode_state", DecodeState),
                      ("rope", RopeArgs), ("attn_decode", AttnDecodeArgs), ("sample", SampleArgs),
                      ("gemv_batch", GemvBatchArgs), ("p2p_dev", P2pDev)):
        n = getattr(lib, f"usdm_sizeof_{name}" if name in ("decode_state", "p2p_dev") else f"usdm_sizeof_{name}_args")()
        if n != C.sizeof(cls):
            raise ImportError(f"ABI mismatch: usdm_{name}_args is {n} bytes in the library, {C.sizeof(cls)} in Python")


_selfcheck()
