"""ctypes binding of libevhip.so (include/evhip.h, include/evhip_ops.h).

There is NO fallback: if the library has not been built (``python emotivoice_amd/csrc/build.py`` or
``__graft_entry__.build()``) importing this module raises, and ``ev_create`` itself fails when no HIP
device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVHIP_LIB", os.path.join(_HERE, "csrc", "libevhip.so"))
EV_ABI_VERSION = 7
EV_PREC_F16, EV_PREC_F32, EV_PREC_X3, EV_PREC_MX = 0, 1, 2, 3
EV_FLAG_DEVICE_INPUTS, EV_FLAG_NO_VOCODER, EV_FLAG_WANT_INT16, EV_FLAG_FORCED_DURATIONS = 1, 2, 4, 8


class ev_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_vocab", C.c_int32), ("n_speaker", C.c_int32), ("n_mels", C.c_int32),
        ("hidden", C.c_int32), ("heads", C.c_int32), ("enc_layers", C.c_int32), ("dec_layers", C.c_int32),
        ("ffn_kernel", C.c_int32), ("bert_dim", C.c_int32), ("dur_layers", C.c_int32), ("pitch_layers", C.c_int32),
        ("energy_layers", C.c_int32), ("var_kernel", C.c_int32), ("var_embed_kernel", C.c_int32), ("n_up", C.c_int32),
        ("up_rates", C.c_int32 * 8), ("up_kernels", C.c_int32 * 8), ("up_init_ch", C.c_int32), ("n_rb", C.c_int32),
        ("rb_kernels", C.c_int32 * 8), ("rb_dils", (C.c_int32 * 4) * 8), ("n_rb_dils", C.c_int32),
        ("sample_rate", C.c_int32), ("decoder_precision", C.c_int32), ("keep_stages", C.c_int32),
        ("token_rate_split", C.c_int32), ("vocoder_chunk_mb", C.c_int32), ("vocoder_streams", C.c_int32),
        ("vocoder_precision", C.c_int32), ("mx_residual", C.c_int32), ("decoder_attention", C.c_int32),
        ("fused_pairs", C.c_int32), ("mx_mrf", C.c_int32), ("decoder_ln_planes", C.c_int32), ("token_splitk", C.c_int32), ("mx_act_format", C.c_int32),
        ("mx_group", C.c_int32),
    ]


class ev_result(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("total_tokens", C.c_int32), ("total_frames", C.c_int64), ("total_samples", C.c_int64),
        ("wav", C.c_void_p), ("wav_i16", C.c_void_p), ("mel", C.c_void_p), ("durations", C.c_void_p),
        ("log_durations", C.c_void_p), ("pitch", C.c_void_p), ("energy", C.c_void_p),
        ("mel_lens", C.POINTER(C.c_int32)), ("mel_offsets", C.POINTER(C.c_int64)),
    ]


class ev_bert_config(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("intermediate", C.c_int32), ("max_position", C.c_int32), ("type_vocab", C.c_int32), ("ln_eps", C.c_float),
                ("reserved", C.c_int32 * 8)]


class ev_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int32), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


class ev_launch_record(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("taps", C.c_int32), ("dil", C.c_int32),
                ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


class ev_conv_gemm_desc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("A", C.c_void_p), ("lda", C.c_int), ("W", C.c_void_p), ("W_lo", C.c_void_p), ("bias", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("taps", C.c_int), ("dil", C.c_int), ("center", C.c_int),
        ("row_valid", C.c_void_p), ("valid_shift", C.c_int), ("row_seq", C.c_void_p), ("seq_bias", C.c_void_p),
        ("ld_seq_bias", C.c_int), ("act", C.c_int), ("act_slope", C.c_float), ("pro_lrelu", C.c_int),
        ("pro_slope", C.c_float), ("res", C.c_void_p), ("res_dtype", C.c_int), ("ldres", C.c_int),
        ("out_scale", C.c_float), ("acc32", C.c_void_p), ("ldacc", C.c_int), ("post_lrelu", C.c_int),
        ("post_slope", C.c_float), ("out16", C.c_void_p), ("out32", C.c_void_p), ("ldo", C.c_int),
        ("out32_before_post", C.c_int), ("reserved0", C.c_int),
        ("add16_a", C.c_void_p), ("add16_b", C.c_void_p), ("ldadd", C.c_int), ("ksplit", C.c_int),
        ("W_mx", C.c_void_p), ("mx_scratch", C.c_void_p), ("mx_scratch_size", C.c_size_t),
        ("mx_x4", C.c_void_p * 2), ("mx_xs", C.c_void_p * 2), ("mx_xs_stride", C.c_uint), ("polyphase_cout", C.c_int),
        ("mxo_h", C.c_void_p), ("mxo_q4", C.c_void_p * 2), ("mxo_qs", C.c_void_p * 2), ("mxo_qs_stride", C.c_uint),
        ("mxo_logC", C.c_int), ("mxo_slope", C.c_float), ("reserved3", C.c_int),
        ("res_x4", C.c_void_p), ("res_xs", C.c_void_p), ("res_xs_stride", C.c_uint), ("res_inv_slope", C.c_float),
        ("acc_h", C.c_void_p), ("acc_x4", C.c_void_p), ("acc_xs", C.c_void_p), ("acc_xs_stride", C.c_uint), ("mxo_partial", C.c_int),
    ]


class ev_res_pair_desc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
                ("M", C.c_int), ("k", C.c_int), ("dil", C.c_int), ("gmin", C.c_int), ("gmax", C.c_int),
                ("w1_mx", C.c_void_p), ("w2_mx", C.c_void_p), ("epi", ev_conv_gemm_desc)]


# every symbol include/evhip.h and include/evhip_ops.h declare: (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "ev_default_config": (None, [C.POINTER(ev_config)]),
    "ev_abi_info": (C.c_int, [C.POINTER(C.c_size_t)]),
    "ev_create": (C.c_int, [C.c_int, C.POINTER(ev_config), C.POINTER(_P)]),
    "ev_destroy": (None, [_P]),
    "ev_last_error": (C.c_char_p, [_P]),
    "ev_set_stream": (C.c_int, [_P, _P]),
    "ev_load_weights": (C.c_int, [_P, _P, C.c_size_t, C.c_char_p]),
    "ev_load_weights_device": (C.c_int, [_P, _P, C.c_size_t, C.c_char_p]),
    "ev_synthesize": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_float, C.c_uint32, C.POINTER(ev_result)]),
    "ev_set_forced_durations": (C.c_int, [_P, _P, C.c_int64]),
    "ev_vocoder": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_uint32, C.POINTER(ev_result)]),
    "ev_get_stage": (C.c_int64, [_P, C.c_char_p, _P, C.c_size_t]),
    "ev_set_profiling": (C.c_int, [_P, C.c_int]),
    "ev_get_timing": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_float)]),
    "ev_kernel_stat_count": (C.c_int, [_P]),
    "ev_get_kernel_stat": (C.c_int, [_P, C.c_int, C.POINTER(ev_kernel_stat)]),
    "ev_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "ev_launch_record_count": (C.c_int, [_P]),
    "ev_get_launch_record": (C.c_int, [_P, C.c_int, C.POINTER(ev_launch_record)]),
    "ev_default_bert_config": (None, [C.POINTER(ev_bert_config)]),
    "ev_style_load_weights": (C.c_int, [_P, C.POINTER(ev_bert_config), _P, C.c_size_t]),
    "ev_style_embed": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_uint32, _P]),
    "ev_op_conv_gemm": (C.c_int, [C.POINTER(ev_conv_gemm_desc), _P]),
    "ev_op_mx_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ev_op_resblock_pair_c32": (C.c_int, [C.POINTER(ev_res_pair_desc), _P]),
    "ev_op_resblock_pair_c64": (C.c_int, [C.POINTER(ev_res_pair_desc), _P]),
    "ev_op_resblock_pair_c32_mx": (C.c_int, [C.POINTER(ev_res_pair_desc), _P]),
    "ev_op_resblock_pair_c64_mx": (C.c_int, [C.POINTER(ev_res_pair_desc), _P]),
    "ev_op_layernorm": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_float, _P, _P, _P, _P, C.c_float, _P, _P]),
    "ev_op_layernorm_planes": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P, C.c_uint, _P]),
    "ev_op_attention": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first (python emotivoice_amd/csrc/build.py). "
                "emotivoice_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        # a stale library (or a stale copy of this file) must fail here with the rebuild hint, not with a bare "undefined symbol" from the
        # binding loop below and not by mis-parsing a descriptor later: ev_abi_info is resolved and checked BEFORE any other symbol
        mine = (C.sizeof(ev_config), C.sizeof(ev_result), C.sizeof(ev_conv_gemm_desc), C.sizeof(ev_res_pair_desc))
        hint = "rebuild with python emotivoice_amd/csrc/build.py"
        try:
            abi_info = l.ev_abi_info
        except AttributeError:
            raise ImportError(f"{LIB_PATH} predates ev_abi_info (ABI < 2); this binding is ABI {EV_ABI_VERSION}: {hint}") from None
        abi_info.restype, abi_info.argtypes = SIGNATURES["ev_abi_info"]
        sizes = (C.c_size_t * 4)()
        ver = abi_info(sizes)
        if ver != EV_ABI_VERSION or tuple(sizes) != mine:
            raise ImportError(f"{LIB_PATH}: ABI version {ver} / struct sizes {tuple(sizes)} do not match this binding "
                              f"({EV_ABI_VERSION} / {mine}): {hint}")
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                raise ImportError(f"{LIB_PATH} does not export {name} although it reports ABI {ver}: {hint}") from None
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib
