"""ctypes binding of libwxhip.so (include/wxhip.h).  There is no CPU fallback: if the
library is missing or a call fails, this raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwxhip.so")


class WxError(RuntimeError):
    pass


class ModelDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class Tuning(C.Structure):
    """wx_tuning (include/wxhip_test.h): launch-shape, scheduling and measurement knobs of the decode loop"""
    _fields_ = [("use_graph", C.c_int), ("check_every", C.c_int), ("cross_split", C.c_int), ("step_variant", C.c_int),
                ("fc2_tile_n", C.c_int), ("profile_launches", C.c_int), ("max_steps_ahead", C.c_int)]

    @classmethod
    def defaults(cls):
        return cls(1, 8, 2, 0, 0, 0, 0)            # WX_TUNING_DEFAULTS


class DecodeOpts(C.Structure):
    """wx_decode_opts (include/wxhip.h).  `tuning` NULL = the library's own configuration; set_tuning keeps the struct alive"""
    _fields_ = [
        ("prompt", C.c_int * 8), ("n_prompt", C.c_int), ("sample_len", C.c_int), ("rules", C.c_int),
        ("max_initial_ts", C.c_int), ("forced_len", C.c_int), ("eot", C.c_int), ("no_speech", C.c_int),
        ("timestamp_begin", C.c_int), ("blank0", C.c_int), ("blank1", C.c_int),
        ("suppress_mask", C.c_void_p), ("capture_qk", C.c_int), ("forced_lens", C.c_void_p),
        ("n_active", C.c_int), ("tuning", C.POINTER(Tuning))]

    def set_tuning(self, **kw):
        t = Tuning.defaults()
        for k, v in kw.items():
            setattr(t, k, int(v))
        self._tuning = t
        self.tuning = C.pointer(t)
        return t


class W2vDims(C.Structure):
    _fields_ = [("n_conv", C.c_int), ("conv_dim", C.c_int), ("conv_kernel", C.c_int * 8), ("conv_stride", C.c_int * 8)] + \
               [(n, C.c_int) for n in ("hidden", "heads", "layers", "ffn", "vocab", "pos_kernel", "pos_groups",
                                       "norm_mode", "stable_ln")]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float
_SIGS = {
    "wx_create": (_I, [_I, C.POINTER(ModelDims), _I, C.POINTER(_P)]),
    "wx_destroy": (None, [_P]),
    "wx_last_error": (C.c_char_p, [_P]),
    "wx_bind_weight": (_I, [_P, C.c_char_p, _P, C.c_size_t]),
    "wx_finalize": (_I, [_P]),
    "wx_set_alignment_heads": (_I, [_P, C.POINTER(_I), _I]),
    "wx_set_mel_filters": (_I, [_P, C.POINTER(_F), _I]),
    "wx_logmel": (_I, [_P, _P, _L, _P, _I, _P, _P, _P]),
    "wx_encode": (_I, [_P, _P, _I, _P, _P]),
    "wx_decode_greedy": (_I, [_P, _P, _I, C.POINTER(DecodeOpts), _P, _P, _P, C.POINTER(_I), _P]),
    "wx_decode_logits": (_I, [_P, _P, _I, _P, _I, _P, _P]),
    "wx_sample_step": (_I, [_P, _P, _L, _P, _I, _I, _I, C.POINTER(DecodeOpts), _P, _P, _P]),
    "wx_get_align_qk": (_I, [_P, _I, _P, _P]),
    "wx_dtw_path": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _I, _P, _P, _P]),
    "wx_ctc_align": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "wx_w2v_create": (_I, [_I, C.POINTER(W2vDims), C.POINTER(_P)]),
    "wx_w2v_destroy": (None, [_P]),
    "wx_w2v_last_error": (C.c_char_p, [_P]),
    "wx_w2v_bind_weight": (_I, [_P, C.c_char_p, _P, C.c_size_t]),
    "wx_w2v_finalize": (_I, [_P]),
    "wx_w2v_num_frames": (_I, [C.POINTER(W2vDims), _L]),
    "wx_w2v_emissions": (_I, [_P, _P, _L, C.POINTER(_I), _I, _P, _I, C.POINTER(_I), _P]),
    "wx_w2v_ctc_align": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "wx_gather_results": (_I, [_P, _P, C.c_size_t, _P, _P]),
    "wx_probe": (_I, [_P, _I, _I, _I, _I, _P]),
    "wx_test_fused_selfq": (_I, [_P, _I, _P, _P, C.POINTER(_I), _P]),
    "wx_test_raise_device_flag": (_I, [_P, _P]),
    "wx_set_encoder_cap": (_I, [_P, _I]),
    "wx_decode_stats": (_I, [_P, C.POINTER(_I), _P]),
    "wx_graph_generation": (_I, [_P]),
    "wx_launch_profile": (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_longlong), _P]),
    "wx_median7_rows": (_I, [_P, _P, _L, _I, _I, _P, _L, _P]),
    "wx_device_status": (_I, [_P, _P]),
    "wx_streams_overlap": (_I, [_I, _P, _I, _I, _P]),
    "wx_gemm_f16": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _P, _I, _P, _L, _P, _L, _I, _P]),
    "wx_skinny_f16": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _P, _P, _P, _P, _L, _P, _P, _L, _I, _I, _P]),
    "wx_skinny_mt_f16": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _P, _P, _P, _P, _L, _P, _P, _L, _I, _I, _P]),
    "wx_pack_gemv_weight": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "wx_skinny_ex": (_I, [_P, _P, _L, _I, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P]),
    "wx_skinny_q8": (_I, [_P, _P, _L, _I, _P, _P, _L, _I, _I, _P, _P, _P, _P, _L, _P, _P, _L, _I, _I, _P]),
    "wx_skinny2_f16": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _P, _I, _I, _P, _P, _L, _P, _P]),
    "wx_skinny2_ln_f16": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _P, _P, _P, _L, _P]),
    "wx_resln_f16": (_I, [_P, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "wx_layernorm_f16": (_I, [_P, _P, _L, _P, _P, _P, _L, _I, _I, _P]),
    "wx_attention_f16": (_I, [_P, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """Loads libwxhip.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64; it must be in the process before libwxhip.so is
        # loaded, otherwise the .so binds /opt/rocm's copy and the two runtimes do not share devices
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise WxError(f"{LIB_PATH} is missing: build it with `python -m whisperx_mlx_amd.build` "
                          "(there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(ctx, rc, what):
    if rc != 0:
        msg = lib().wx_last_error(ctx).decode() if ctx else "no context"
        raise WxError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())
