"""ctypes binding of libscnattn.so (the C ABI declared in include/scnattn.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  The library is built in-tree by ``__graft_entry__.build()`` (``make -C
indonesian-image-captioning_amd/csrc``) so that it travels with the repository snapshot.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscnattn.so")

_lib = None
_lock = threading.Lock()

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "P", "E", "A", "D", "F", "M", "S", "V", "T", "L", "has_att")]


PARAM_FIELDS = (
    "attention_encoder_att_weight", "attention_encoder_att_bias",
    "attention_decoder_att_weight", "attention_decoder_att_bias",
    "attention_full_att_weight", "attention_full_att_bias",
    "embedding_weight",
    "decode_step_weight_ia", "decode_step_weight_ib", "decode_step_weight_ic",
    "decode_step_weight_ha", "decode_step_weight_hb", "decode_step_weight_hc",
    "decode_step_bias_ih", "decode_step_bias_hh",
    "init_h_weight", "init_h_bias", "init_c_weight", "init_c_bias",
    "f_beta_weight", "f_beta_bias", "fc_weight", "fc_bias",
)


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


class Pool(C.Structure):      # scnattn_pool
    _fields_ = [("Q", C.c_int), ("qtap_max", C.c_int), ("tap_idx", C.c_void_p), ("tap_w", C.c_void_p),
                ("qtap_idx", C.c_void_p), ("qtap_w", C.c_void_p), ("col_w", C.c_void_p)]


class ConvExtra(C.Structure):      # scnattn_conv_extra
    _fields_ = [("pro", C.c_int), ("epi", C.c_int), ("pro_ss", C.c_void_p),
                ("stat_partial", C.c_void_p), ("stat_shift", C.c_void_p), ("ez", C.c_void_p), ("emean", C.c_void_p),
                ("einvstd", C.c_void_p), ("egamma", C.c_void_p), ("ebeta", C.c_void_p), ("ldz", C.c_long),
                ("stride", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("force_split", C.c_int), ("force_mi", C.c_int)]


_SIGS = {
    "scnattn_version": ([], i32),
    "scnattn_set_option": ([C.c_char_p, i32], i32),
    "scnattn_profile_collect": ([C.POINTER(C.c_double)], i32),
    "scnattn_seq_workspace": ([C.POINTER(Dims), C.POINTER(Pool), C.POINTER(sz), C.POINTER(sz)], i32),
    "scnattn_seq_fwd": ([vp, C.POINTER(Dims), C.POINTER(Params), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                         C.POINTER(Pool)], i32),
    "scnattn_seq_bwd": ([vp, C.POINTER(Dims), C.POINTER(Params), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                         C.POINTER(Params), vp, vp, C.POINTER(Pool)], i32),
    "scnattn_seq_bwd_streams": ([vp, vp, C.POINTER(Dims), C.POINTER(Params), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                 C.POINTER(Params), vp, vp, C.POINTER(Pool)], i32),
    "scnattn_sgemm": ([vp, i32, i32, i32, i32, i32, f32, vp, i64, vp, i64, f32, vp, i64, vp, vp, i32, i64, i64, i64], i32),
    "scnattn_sgemm_ws": ([vp, i32, i32, i32, i32, i32, f32, vp, i64, vp, i64, f32, vp, i64, vp, vp, i32, i64, i64, i64,
                          vp, i64], i32),
    "scnattn_cgemm": ([vp, i32, i32, i32, i32, i32, f32, vp, i64, vp, i64, f32, vp, i64, vp, vp, i32, i64, i64, i64,
                       vp, i64, C.POINTER(ConvExtra)], i32),
    "scnattn_cgemm_row_tiles": ([i32], i32),
    "scnattn_conv1x1_fwd": ([vp, i32, i32, i32, vp, vp, vp, C.POINTER(ConvExtra), vp, i64], i32),
    "scnattn_conv1x1_dgrad": ([vp, i32, i32, i32, vp, vp, i32, f32, vp, C.POINTER(ConvExtra), vp, i64], i32),
    "scnattn_conv1x1_wgrad": ([vp, i32, i32, i32, vp, vp, vp, C.POINTER(ConvExtra), vp, i64], i32),
    "scnattn_conv3x3_fwd": ([vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(ConvExtra), vp, i64], i32),
    "scnattn_conv3x3_dgrad": ([vp, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(ConvExtra), vp, i64], i32),
    "scnattn_conv3x3_wgrad": ([vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i64, i32], i32),
    "scnattn_conv3x3_dgrad_strided": ([vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i64], i32),
    "scnattn_cgemm_stat_ld": ([i32], i32),
    "scnattn_bn_finalize": ([vp, i64, i32, vp, i32, i32, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp], i32),
    "scnattn_bn_apply_fin": ([vp, i64, i32, vp, vp, i32, vp, i32, i32, vp, f32, f32, vp, vp, i32, vp, vp, vp, vp, vp, vp], i32),
    "scnattn_bn_bwd_reduce": ([vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, C.POINTER(C.c_int)], i32),
    "scnattn_bn_bwd_dx_fin": ([vp, i64, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp], i32),
    "scnattn_bf16_weights": ([vp, i32, vp, vp, i32], i32),
    "scnattn_cgemm16": ([vp, i32, i32, i32, vp, i64, vp, i64, f32, vp, i64, i32, vp, i64, C.POINTER(ConvExtra)], i32),
    "scnattn_conv3x3_fwd16": ([vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(ConvExtra), vp, i64], i32),
    "scnattn_conv3x3_dgrad16": ([vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(ConvExtra), vp, i64], i32),
    "scnattn_wgrad16_3x3": ([vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i64, i32], i32),
    "scnattn_block16_sizes": ([vp, vp, vp, vp, vp, vp], i32),
    "scnattn_block16_fwd": ([vp, vp], i32),
    "scnattn_block16_bwd": ([vp, vp, vp, vp], i32),
    "scnattn_wgrad16_rows": ([vp, i32, i32, i32, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32], i32),
    "scnattn_stem_tiles": ([i32, i32, i32], i32),
    "scnattn_stem_conv7": ([vp, i32, i32, i32, vp, i64, i64, i64, i64, vp, i64, i64, i64, i64, vp, vp, vp], i32),
    "scnattn_stem_bn_relu_maxpool": ([vp, i32, i32, i32, i32, vp, vp, vp, i32], i32),
    "scnattn_bn_stats_fold": ([vp, i32, i32, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    "scnattn_skinny_gemm": ([vp, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i32,
                             C.POINTER(i32)], i32),
    "scnattn_skinny_gemm_bf16w": ([vp, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i32,
                                   C.POINTER(i32)], i32),
    "scnattn_skinny_gemm_bf16": ([vp, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i32,
                                  C.POINTER(i32)], i32),
    "scnattn_f32_to_bf16": ([vp, i64, vp, vp], i32),
    "scnattn_attn_scores": ([vp, i32, i32, i32, vp, vp, i32, i64, i64, vp, vp, vp, vp, vp], i32),
    "scnattn_attn_context": ([vp, i32, i32, i32, vp, vp, vp, i32, i64, i64, vp, vp, i64, vp, vp, vp, vp], i32),
    "scnattn_mean_pixels": ([vp, i32, i32, i32, vp, vp], i32),
    "scnattn_attn_dalpha": ([vp, i32, i32, i32, vp, vp, vp, i64, vp], i32),
    "scnattn_attn_softmax_bwd": ([vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64], i32),
    "scnattn_attn_datt1_post_blocks": ([i32, i32], i32),
    "scnattn_attn_datt1_post": ([vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp], i32),
    "scnattn_scn_mix_fwd": ([vp, i32, i32, vp, i32, i64, i64, vp, vp, i32, i64, i64, vp, vp, vp, vp, vp], i32),
    "scnattn_lstm_fwd": ([vp, i32, i32, vp, i32, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp], i32),
    "scnattn_lstm_bwd": ([vp, i32, i32, i32, vp, vp, i32, i64, i64, vp, vp, vp, vp, vp], i32),
    "scnattn_scn_mix_bwd": ([vp, i32, i32, vp, i32, i64, i64, i64, vp, vp, vp, vp, vp, vp, i64, vp, vp], i32),
    "scnattn_gate_bwd": ([vp, i32, i32, vp, i32, i64, i64, vp, vp, vp, vp, i64], i32),
    "scnattn_transpose2d": ([vp, i32, i32, vp, i64, vp, i64], i32),
    "scnattn_colsum": ([vp, i32, i32, vp, i64, vp, f32], i32),
    "scnattn_mul_bcast": ([vp, i32, i32, i32, vp, vp, vp], i32),
    "scnattn_pool_permute_fwd": ([vp, i32, i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, vp], i32),
    "scnattn_pool_permute_bwd": ([vp, i32, i32, i32, i32, i32, i32, vp, vp, i64, i64, i64, i64], i32),
    "scnattn_caption_loss_fwd": ([vp, i32, i32, i32, i32, vp, vp, i64, vp, i64, vp, f32, vp, vp, vp, vp, vp], i32),
    "scnattn_caption_loss_bwd": ([vp, i32, i32, i32, i32, vp, vp, i64, vp, i64, vp, vp, f32, vp, vp, vp], i32),
    "scnattn_u8_gather_normalize": ([vp, vp, i64, vp, i64, i32, i64, vp, vp, i32, i32], i32),
    "scnattn_bn_workspace_floats": ([i32], i32),
    "scnattn_bn_stats": ([vp, i32, i32, vp, i32, f32, f32, vp, vp, vp, vp, vp], i32),
    "scnattn_bn_apply": ([vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, i32, vp], i32),
    "scnattn_bn_bwd": ([vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp], i32),
    "scnattn_dp_unique_id": ([C.c_char_p], i32),
    "scnattn_dp_comm_create": ([C.c_char_p, i32, i32, C.POINTER(vp)], i32),
    "scnattn_dp_comm_allreduce_bucket": ([vp, vp, vp, i64], i32),
    "scnattn_dp_comm_finish": ([vp, vp], i32),
    "scnattn_dp_comm_set_stream": ([vp, vp], i32),
    "scnattn_dp_comm_world": ([vp], i32),
    "scnattn_dp_comm_destroy": ([vp], i32),
    "scnattn_clamp_adam": ([vp, i64, vp, vp, vp, vp, f64, f64, f64, f64, i32, f64, f64], i32),
}

EXPORTS = tuple(_SIGS) + ("scnattn_last_error",)


def lib():
    """Load (once) and return the ctypes handle.  Raises RuntimeError when the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "libscnattn.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` or `make -C indonesian-image-captioning_amd/csrc`. There is no CPU fallback." % LIB_PATH)
            h = C.CDLL(LIB_PATH)
            for name, (args, res) in _SIGS.items():
                fn = getattr(h, name)
                fn.argtypes = args
                fn.restype = res
            h.scnattn_last_error.argtypes = []
            h.scnattn_last_error.restype = C.c_char_p
            # A/B runs without touching code: SCNATTN_OPTIONS="cgemm_combine=0,dec_tail=1" (scnattn_set_option names)
            for item in filter(None, os.environ.get("SCNATTN_OPTIONS", "").split(",")):
                name, _, val = item.partition("=")
                if h.scnattn_set_option(name.strip().encode(), int(val)) != 0:
                    raise RuntimeError("SCNATTN_OPTIONS: %s" % h.scnattn_last_error().decode())
            _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().scnattn_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """The current HIP stream of the tensor's device, as the void* the C ABI takes."""
    if _raw_stream is not None:     # ~10x cheaper than building a torch.cuda.Stream object per call
        return C.c_void_p(_raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("scnattn: the HIP path needs tensors on an MI355X device (got a %s tensor); "
                               "there is no CPU fallback" % t.device.type)


def f32c(t):
    """float32 + contiguous (no copy when already so)."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
