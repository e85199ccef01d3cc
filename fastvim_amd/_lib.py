"""ctypes binding of libfastvim_hip.so (C ABI: include/fastvim_hip.h).

The product path fails loudly when the library is absent -- there is no fallback.
"""
import ctypes
import os
import threading

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libfastvim_hip.so")

FV_F32, FV_BF16, FV_F16 = 0, 1, 2
_DTYPES = {torch.float32: FV_F32, torch.bfloat16: FV_BF16, torch.float16: FV_F16}

_lib = None

# every symbol include/fastvim_hip.h declares (tests check the .so exports all of them)
C_ABI_SYMBOLS = (
    "fv_last_error", "fv_version",
    "fv_selective_scan_fwd", "fv_selective_scan_bwd_workspace", "fv_selective_scan_bwd",
    "fv_causal_conv1d_fwd", "fv_causal_conv1d_bwd", "fv_scan_expand_skip_fwd",
    "fv_mixer_conv_pool_fwd", "fv_mixer_scan_fwd", "fv_mixer_xproj_scan_fwd_ok", "fv_mixer_xproj_scan_fwd", "fv_mixer_combine_fwd",
    "fv_mixer_bwd_blocks", "fv_mixer_combine_bwd", "fv_mixer_scan_bwd_chunks", "fv_mixer_scan_bwd_chunks_b",
    "fv_mixer_scan_bwd_ckpt_floats", "fv_mixer_scan_bwd_partials", "fv_mixer_scan_bwd", "fv_mixer_scan_bwd_dir", "fv_mixer_scan_bwd_ckpt", "fv_mixer_scan_bwd_segments", "fv_mixer_scan_bwd_seg_floats", "fv_mixer_scan_bwd_seg_partials", "fv_mixer_scan_bwd_seg_chunks", "fv_mixer_scan_bwd_seg", "fv_mixer_scan_fwd_ckpt", "fv_mixer_scan_ckpt_floats", "fv_mixer_scan_fwd_segments", "fv_mixer_scan_fwd_seg_floats", "fv_mixer_scan_fwd_seg", "fv_rows_segment_sum", "fv_rows_gather", "fv_mixer_conv_pool_bwd", "fv_mixer_conv_pool_bwd2_ok", "fv_mixer_conv_pool_bwd2", "fv_mixer_scan_bwd_xproj_ok", "fv_mixer_scan_bwd_xproj", "fv_chunk_rows_bf16", "fv_reduce_partials", "fv_reduce_partials_multi",
    "fv_add_norm_blocks", "fv_add_norm_fwd", "fv_add_norm_bwd", "fv_gemm_bf16", "fv_gemm_bf16_tn_grouped", "fv_gemm_bf16_tn_grouped_ld", "fv_mixer_xproj_fwd", "fv_mixer_xproj_bwd_slices", "fv_mixer_xproj_bwd", "fv_mixer_xproj_bwd2", "fv_mixer_xproj_bwd3_ok", "fv_mixer_xproj_bwd3", "fv_adamw_flat", "fv_soft_target_ce",
    "fv_patch_unfold", "fv_gemm_bf16_rowbias", "fv_mean_pool_fwd", "fv_mean_pool_bwd", "fv_droppath_table", "fv_scale_cast",
    "fv_column_sum", "fv_gemm_bf16_addnorm", "fv_gemm_bf16_dgrad_addnorm_blocks", "fv_gemm_bf16_dgrad_addnorm_bwd", "fv_gemm_bf16_dgrad_addnorm_bwd2", "fv_gemm_bf16_addnorm2", "fv_mixer_combine_out_proj_addnorm_ok", "fv_mixer_combine_out_proj_addnorm", "fv_gemm_f32",
    "fv_mixer_conv_pool_bwd_dgrad_ok", "fv_mixer_conv_pool_bwd_dgrad_blocks", "fv_mixer_conv_pool_bwd_dgrad", "fv_transpose_bf16_batched", "fv_gemm_bf16_tn_grouped_wide8",
)


_tls = threading.local()      # .dev: device index of the tensor this THREAD's last ``stream_of`` was asked about


class _DeviceGuarded:
    """The CDLL with every entry point wrapped in a device guard: a kernel is launched with the device of the
    tensor whose stream it was given (``stream_of``) current, like the reference's ops do with
    ``at::cuda::CUDAGuard`` (selective_scan.cpp:326-327) -- costs one ``current_device()`` query per launch, and a
    ``torch.cuda.device`` switch only when the tensors live on another GPU than the current one.  The launch device is
    per host thread (two threads driving two GPUs do not see each other's), and a call that no ``stream_of`` preceded
    in its thread runs on the current device."""

    def __init__(self, cdll):
        self._cdll = cdll
        self._wrapped = {}

    def __getattr__(self, name):
        w = self._wrapped.get(name)
        if w is None:
            fn = getattr(self._cdll, name)

            def call(*args, _fn=fn):
                dev = getattr(_tls, "dev", None)
                if dev is None or dev == torch.cuda.current_device():
                    return _fn(*args)
                with torch.cuda.device(dev):
                    return _fn(*args)

            w = self._wrapped[name] = call
        return w


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m fastvim_amd.build` "
                "(or __graft_entry__.build()). fastvim_amd has no CPU / eager fallback.")
        cdll = ctypes.CDLL(LIB_PATH)
        cdll.fv_last_error.restype = ctypes.c_char_p
        cdll.fv_selective_scan_bwd_workspace.restype = ctypes.c_size_t
        cdll.fv_mixer_scan_bwd_ckpt_floats.restype = ctypes.c_size_t
        cdll.fv_mixer_scan_ckpt_floats.restype = ctypes.c_size_t
        cdll.fv_mixer_scan_fwd_seg_floats.restype = ctypes.c_size_t
        cdll.fv_mixer_scan_bwd_seg_floats.restype = ctypes.c_size_t
        _lib = _DeviceGuarded(cdll)
    return _lib


def dtype_code(dt):
    try:
        return _DTYPES[dt]
    except KeyError:
        raise RuntimeError(f"fastvim_amd: unsupported dtype {dt}") from None


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream_of(t):
    _tls.dev = t.device.index
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fastvim_amd ops run on the GPU only (HIP kernels); got a CPU tensor")


def check(rc, what):
    if rc != 0:
        msg = lib().fv_last_error().decode()
        raise RuntimeError(f"{what}: {msg} (code {rc})")


def i32(v):
    return ctypes.c_int(int(v))
