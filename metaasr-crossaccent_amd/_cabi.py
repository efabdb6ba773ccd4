"""ctypes binding of libmasr.so (include/masr.h; the masr_test_* entries are include/masr_test.h).  No fallback: if the HIP library is missing the
import raises -- the product path never runs on the CPU oracle."""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MASR_LIB", _HERE / "lib" / "libmasr.so"))


class MasrConfig(C.Structure):
    _fields_ = [("idim", C.c_int32), ("odim", C.c_int32), ("d_model", C.c_int32), ("nheads", C.c_int32),
                ("d_inner", C.c_int32), ("enc_layers", C.c_int32), ("dec_layers", C.c_int32),
                ("tie_weights", C.c_int32), ("dropout", C.c_float), ("pos_dropout", C.c_float),
                ("label_smoothing", C.c_float)]


vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "masr_version": (C.c_int, []),
    "masr_last_error": (C.c_char_p, []),
    "masr_create": (vp, [C.POINTER(MasrConfig)]),
    "masr_destroy": (None, [vp]),
    "masr_param_numel": (i64, [vp]),
    "masr_param_count": (i32, [vp]),
    "masr_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64)]),
    "masr_workspace_bytes": (i64, [vp, i32, i32, i32]),
    "masr_bind": (i32, [vp, vp, vp, vp, vp, i64]),
    "masr_refresh": (i32, [vp, vp]),
    "masr_set_seed": (None, [vp, C.c_uint64]),
    "masr_set_concurrency": (None, [vp, i32]),
    "masr_dropout_state": (None, [vp, C.POINTER(C.c_uint64), i32]),
    "masr_run_batch": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "masr_set_step_graphs": (None, [vp, i32]),
    "masr_set_split_wgrad_launches": (None, [vp, i32]),
    "masr_set_ksplit": (None, [vp, i32]),
    "masr_set_drop_nan_grads": (None, [vp, i32]),
    "masr_step_counters": (None, [vp, C.POINTER(i64)]),
    "masr_read_stats": (i32, [vp, C.POINTER(f32), vp]),
    "masr_stats_post": (i64, [vp, vp]),
    "masr_stats_peek": (C.POINTER(C.c_uint32), [vp, i64]),
    "masr_stats_wait": (i32, [vp, i64, C.POINTER(f32)]),
    "masr_last_logits": (i32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "masr_grad_norm": (i32, [vp, vp]),
    "masr_clip_sgd_step": (i32, [vp, vp, f32, f32, f32, i32, i32, vp]),
    "masr_clip_grads": (i32, [vp, f32, vp]),
    "masr_clip_accumulate": (i32, [vp, vp, f32, vp]),
    "masr_clip_scale_flat": (i32, [vp, i64, vp, f32, vp]),
    "masr_adam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp]),
    "masr_adam_step_guarded": (i32, [vp, vp, vp, vp, vp, i64, f32, i32, f32, i32, f32, f32, f32, f32, i32, i32, vp]),
    "masr_sum_n": (i32, [vp, vp, i32, f32, i64, vp]),
    "masr_adam_sum_step": (i32, [vp, vp, i32, f32, vp, vp, i64, f32, f32, f32, f32, i32, vp]),
    "masr_adamw_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, i32, vp]),
    "masr_radam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, i32, vp]),
    "masr_sgd_step": (i32, [vp, vp, vp, i64, f32, f32, i32, i32, vp]),
    "masr_scale": (i32, [vp, i64, f32, vp]),
    "masr_axpy": (i32, [vp, vp, i64, f32, vp]),
    "masr_copy": (i32, [vp, vp, i64, vp]),
    "masr_allreduce_unique_id": (i32, [C.c_char_p]),
    "masr_allreduce_init": (vp, [i32, i32, C.c_char_p]),
    "masr_allreduce_destroy": (None, [vp]),
    "masr_allreduce": (i32, [vp, vp, i64, vp, f32, i32, vp]),
    "masr_allreduce_wait": (i32, [vp, vp]),
    "masr_allreduce_check": (i32, [vp, i32]),
    "masr_stats_device": (vp, [vp]),
    "masr_recog": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "masr_recog_full": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "masr_edit_distance": (i64, [vp, i32, vp, i32]),
    "masr_blstm_create": (vp, [vp]),
    "masr_blstm_destroy": (None, [vp]),
    "masr_blstm_param_numel": (i64, [vp]),
    "masr_blstm_param_count": (i32, [vp]),
    "masr_blstm_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64)]),
    "masr_blstm_workspace_bytes": (i64, [vp, i32, i32, i32]),
    "masr_blstm_bind": (i32, [vp, vp, vp, vp, i64]),
    "masr_blstm_refresh": (i32, [vp, vp]),
    "masr_blstm_run_batch": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "masr_blstm_forward": (i32, [vp, vp, vp, i32, i32, vp]),
    "masr_blstm_read_stats": (i32, [vp, C.POINTER(f32), vp]),
    "masr_blstm_set_resident_recurrence": (None, [vp, i32]),
    "masr_blstm_check": (i32, [vp, vp]),
    "masr_blstm_last_logits": (i32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "masr_blstm_clip_grads": (i32, [vp, f32, vp]),
    "masr_blstm_clip_sgd_step": (i32, [vp, vp, f32, f32, f32, i32, i32, vp]),
    "masr_fbank": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "masr_fbank_pitch_work_bytes": (i64, [i64, i32, i32]),
    "masr_fbank_pitch": (i32, [vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, i64, vp]),
    "masr_gather_pad": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "masr_ctc_work_floats": (i64, [i32, i32, i32]),
    "masr_ctc_status": (i32, [vp, i32, i32, i32, vp]),
    "masr_ctc_loss": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    "masr_profile_enable": (i32, [vp, i32]),
    "masr_profile_read": (i32, [vp, C.POINTER(f32), C.POINTER(i32)]),
    "masr_test_blstm_stall": (None, [vp, i32]),          # include/masr_test.h from here on
    "masr_test_gemm": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, vp, i32, vp, i64, vp]),
    "masr_test_dropout_mask": (i32, [C.c_uint32, C.c_uint32, i64, f32, vp, vp]),
    "masr_test_gemm_dropout": (i32, [vp, i64, vp, i64, i32, i32, i32, f32, C.c_uint32, C.c_uint32, vp, i64, vp]),
    "masr_test_attention_dropout": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, C.c_uint32, C.c_uint32, vp]),
    "masr_test_gemm_epi": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, i32, f32, vp, vp, vp, vp, vp]),
    "masr_test_linear_shadows": (i32, [vp, i64, i32, i32, i32, vp, vp, vp]),
    "masr_test_conv1_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "masr_test_conv3x3": (i32, [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "masr_test_conv3x3_ex": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "masr_test_conv3x3_sign_bits": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "masr_test_conv3x3_pool_idx": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "masr_test_conv3x3_dgrad_pooled": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "masr_test_conv1_wgrad_fused_slab_floats": (i64, [i32, i32, i32]),
    "masr_test_conv1_wgrad_fused": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, vp]),
    "masr_test_conv3x3_wgrad": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "masr_test_conv3x3_wgrad_slab_floats": (i64, [i32, i32, i32, i32, i32]),
    "masr_test_wgrad_grouped": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]),
    "masr_test_ksplit_ln": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, f32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "masr_test_wgrad_grouped_n": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "masr_test_conv3x3_wgrad_pooled": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "masr_test_layernorm_slab_floats": (i64, [i32, i32]),
    "masr_test_layernorm": (i32, [vp] * 13 + [i32, i32, f32, C.c_uint32, C.c_uint32, vp]),
    "masr_test_attention_dropout_bwd": (i32, [vp] * 10 + [i32] * 6 + [f32, C.c_uint32, C.c_uint32, vp]),
    "masr_test_attention": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
}
EXPORTS = tuple(_SIGS)
PROF_NAMES = ("conv1_fwd", "conv2_fwd", "conv3_fwd", "conv4_fwd", "conv2_dgrad", "conv3_dgrad", "conv4_dgrad", "conv2_wgrad",
              "conv3_wgrad", "conv4_wgrad", "conv1_wgrad", "gemm_enc", "gemm_dec", "wgrad_enc", "wgrad_dec", "attn_enc", "attn_dec",
              "layernorm", "pool", "optim", "shadows", "misc")          # include/masr.h MASR_PROF_*

_lib = None


def lib():
    """Load libmasr.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"libmasr.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                               "There is no CPU fallback for the product path.")
        l = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class MasrError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise MasrError(f"{what} failed ({rc}): {lib().masr_last_error().decode()}")
