"""ctypes binding of libisg.so (the C ABI declared in include/isg.h).

The library is built in-tree by ``imagesegmentation_amd.build``; there is no CPU fallback: if the
shared object is missing, or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISG_LIBRARY points at another build of the same ABI (variant libraries for A/B runs of kernel changes)
LIB_PATH = os.environ.get("ISG_LIBRARY") or os.path.join(_HERE, "libisg.so")

_vp = ctypes.c_void_p
_int = ctypes.c_int
_f32 = ctypes.c_float
_f64 = ctypes.c_double
_ll = ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/isg.h one to one
SIGNATURES = {
    "isg_version": (_int, []),
    "isg_last_error": (ctypes.c_char_p, []),
    "isg_raster_iou": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "isg_shape_mask": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "isg_history_frames": (_int, [_vp, _int, _int, _int, _int, _vp, _int, _int, _vp]),
    "isg_batch_points": (_int, [_vp, _vp, _int, _vp, _vp, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "isg_env_step": (_int, [_vp, _vp, _int, _vp, _int, _vp, _vp, _int, _vp, _int, _int, _int, _vp, _vp, _vp, _int, _int,
                            _vp]),
    "isg_softmax_xent_fwd": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "isg_softmax_xent_bwd": (_int, [_vp, _vp, _vp, _vp, _f32, _int, _int, _int, _vp, _vp]),
    "isg_sequence_xent_fwd": (_int, [_vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "isg_softmax_xent_mean_bwd": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "isg_argmax_rows": (_int, [_vp, _int, _int, _int, _vp, _vp]),
    "isg_sample_rows": (_int, [_vp, _int, _int, _int, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _vp]),
    "isg_sample_actions": (_int, [_vp, _int, _int, _int, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _int, _vp, _vp]),
    "isg_decode_step": (_int, [_vp, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _int, _vp]),
    "isg_sequence_metrics": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "isg_sequence_summary": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "isg_bias_relu": (_int, [_vp, _vp, _ll, _int, _int, _vp]),
    "isg_relu_bwd_bias_workspace_bytes": (_ll, [_int]),
    "isg_relu_bwd_bias": (_int, [_vp, _vp, _ll, _int, _int, _vp, _vp, _vp, _vp]),
    "isg_relu_bwd_bias_strided": (_int, [_vp, _ll, _vp, _ll, _ll, _int, _int, _vp, _vp, _vp, _vp]),
    "isg_relu_bwd_bias_static": (_int, [_vp, _vp, _ll, _int, _int, _vp, _vp, _vp]),
    "isg_relu_bwd_bias_static_init": (_int, []),
    "isg_conv_in_packed_elems": (_ll, [_int, _int]),
    "isg_conv_in_pack": (_int, [_vp, _int, _int, _vp, _vp]),
    "isg_conv_in_pad_image": (_int, [_vp, _ll, _vp, _vp]),
    "isg_conv_in_fwd": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    "isg_conv_in_wgrad_workspace_bytes": (_ll, [_int, _int]),
    "isg_conv_in_wgrad": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _vp, _vp, _vp]),
    "isg_conv32_packed_elems": (_ll, [_int]),
    "isg_conv32_pack": (_int, [_vp, _int, _int, _int, _vp, _vp]),
    "isg_conv32_pack_padded": (_int, [_vp, _int, _int, _int, _vp, _vp]),
    "isg_conv32": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp, _vp]),
    "isg_conv32_strided": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp, _ll, _vp]),
    "isg_conv32_wgrad_workspace_bytes": (_ll, [_int]),
    "isg_conv32_wgrad": (_int, [_vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp]),
    "isg_conv16_packed_elems": (_ll, [_int]),
    "isg_conv16": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp, _vp]),
    "isg_conv32_masked_workspace_bytes": (_ll, []),
    "isg_conv32_masked": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "isg_conv16_masked": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "isg_conv16_wgrad_workspace_bytes": (_ll, [_int]),
    "isg_conv16_wgrad": (_int, [_vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp]),
    "isg_avgpool_fwd": (_int, [_vp, _ll, _int, _int, _int, _int, _int, _vp, _vp]),
    "isg_convlstm_packed_elems": (_ll, [_int, _int, _int]),
    "isg_convlstm_pack_weights": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "isg_convlstm_cell_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _int,
                                     _vp, _int, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _int, _vp]),
    "isg_convlstm_cell_fwd_strided": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _int,
                                             _vp, _int, _ll, _vp, _vp, _int, _vp, _vp, _ll, _int, _vp, _int, _vp]),
    "isg_copy_channel_slice": (_int, [_vp, _vp, _vp, _vp, _int, _ll, _ll, _ll, _vp]),
    "isg_a3c_loss_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "isg_a3c_loss_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "isg_rollout_record": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "isg_gae": (_int, [_vp, _vp, _vp, _vp, _int, _int, _f64, _f64, _vp, _vp, _vp]),
    "isg_column_sums_workspace_bytes": (_ll, [_int]),
    "isg_column_sums": (_int, [_vp, _ll, _int, _int, _vp, _vp, _vp]),
    "isg_decode_begin": (_int, [_vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "isg_decode_polygons": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp]),
    "isg_sum_slices_bias": (_int, [_vp, _int, _int, _int, _vp, _int, _vp, _vp]),
    "isg_sum_slices_bias_padded": (_int, [_vp, _int, _int, _int, _int, _vp, _int, _vp, _vp]),
    "isg_rows_dot": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "isg_pad_channels": (_int, [_vp, _int, _int, _vp, _int, _ll, _vp]),
    "isg_adam_tf": (_int, [_vp, _vp, _vp, _vp, _ll, _f64, _f64, _f64, _f64, _vp, _f64, _vp]),
    "isg_refresh_derived": (_int, [_vp, _ll, _vp, _ll, _vp, _vp, _ll, _vp, _int, _vp, _vp, _vp, _vp]),
    "isg_convlstm_pooled_fwd_supported": (_int, [_int, _int, _int, _int, _int, _int, _int, _int]),
    "isg_convlstm_cell_fwd_pooled": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _int,
                                            _vp, _int, _ll, _vp, _vp, _vp, _vp, _int, _int, _vp, _ll, _vp]),
    "isg_convlstm_pointwise_bwd": (_int, [_vp, _vp, _vp, _vp, _ll, _int, _vp, _vp, _int, _int, _int, _int, _int, _int,
                                          _vp, _int, _vp, _vp, _vp]),
    "isg_convlstm_dgrad": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _int, _vp, _int, _int, _int, _vp,
                                  _vp]),
    "isg_convlstm_wgrad_workspace_bytes": (_ll, [_int, _int]),
    "isg_convlstm_wgrad": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp, _int, _vp, _int, _ll,
                                  _vp, _int, _vp, _vp, _vp, _vp]),
    "isg_convlstm_bwd_fused_supported": (_int, [_int, _int, _int, _int, _int, _int, _int]),
    "isg_convlstm_state_bf16_supported": (_int, [_int, _int, _int, _int, _int, _int]),
    "isg_convlstm_bwd_fused_hist_bytes": (_ll, [_int, _int]),
    "isg_convlstm_bwd_fused_hist_begin": (_int, [_vp, _int, _int, _vp]),
    "isg_convlstm_bwd_fused_hist_finish": (_int, [_vp, _int, _int, _int, _int, _vp, _vp]),
    "isg_convlstm_bwd_fused": (_int, [_vp, _vp, _vp, _vp, _ll, _int, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int,
                                      _int, _int, _vp, _int, _vp, _int, _ll, _vp, _vp, _vp, _vp, _int, _int, _int, _vp,
                                      _int, _vp, _vp]),
    "isg_convlstm_bwd_fused_strided": (_int, [_vp, _vp, _vp, _int, _vp, _ll, _int, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _int,
                                              _int, _int, _int, _int, _int, _vp, _int, _vp, _vp, _vp, _int, _int, _int,
                                              _vp, _int, _vp]),
    "isg_convlstm_wgrad_finish": (_int, [_vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
}

_lib = None


class IsgError(RuntimeError):
    pass


def lib():
    """The loaded library; raises ImportError with build instructions when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libisg.so not found at %s - build it first: python -m imagesegmentation_amd.build" % LIB_PATH)
        # torch first: libisg.so needs libamdhip64.so.7 and must share the HIP runtime instance that
        # torch has loaded (same device context, same streams), not bring a second one in
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().isg_last_error().decode("utf-8", "replace")
        raise IsgError("%s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("libisg needs contiguous tensors")
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
