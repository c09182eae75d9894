"""ctypes binding of include/rtfs_hip.h (librtfs_hip.so).

The product path has NO fallback: if the library is missing or a launch fails this module raises.
Tensors are handed over as raw device pointers; kernels are enqueued on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "librtfs_hip.so"

STAT_STRIDE = 16  # doubles per utterance in a gLN statistics slot (kStatStride in csrc/common.h): one 128-byte line each

P = c_void_p
I = c_int
F = c_float
LL = c_longlong
D = c_double

# name -> argtypes, in the order of include/rtfs_hip.h
SIGNATURES = {
    "rtfs_stft_fwd": [P, P, I, I, P],
    "rtfs_enc_conv_fwd": [P, P, P, P, I, I, P],
    "rtfs_bottleneck_fwd": [P, P, P, P, P, P, P, I, I, P],
    "rtfs_proj_fwd": [P, P, P, F, P, P, P, P, I, I, P],
    "rtfs_dwconv_fwd": [P, P, P, P, F, I, I, I, P, P, P, P, I, I, I, P],
    "rtfs_dwconv_mix_fwd": [P] * 12 + [I, P, P, P, P, I, I, I, I, I, P],
    "rtfs_pool_fwd": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "rtfs_dp_unfold_gemm_fwd": [P, P, P, P, P, I, I, I, I, P],
    "rtfs_sru_scan_fwd": [P, P, P, P, F, P, I, I, I, P],
    "rtfs_sru_layer_fwd": [P, P, P, P, F, P, P, P, I, I, P],
    "rtfs_sru_layer_fwd_form": [P, P, P, P, F, P, P, P, I, I, I, P],
    "rtfs_vp_param_count": [],
    "rtfs_vp_block_fwd": [P, P, P, P, I, I, P],
    "rtfs_neg_sdr_sums": [P, P, P, I, I, I, P],
    "rtfs_neg_sdr_finish": [P, I, I, I, P, P, I, I, I, P],
    "rtfs_neg_sdr_grad": [P, P, P, P, P, I, I, I, P],
    "rtfs_lip_stem_fwd": [P, P, P, P, P, P, I, I, I, I, P],
    "rtfs_lip_maxpool_fwd": [P, P, I, I, I, P],
    "rtfs_conv_nhwc_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "rtfs_lip_avgpool_fwd": [P, P, I, I, I, I, P],
    "rtfs_lip_roi_fwd": [P, P, P, P, I, I, I, I, I, I, P],
    "rtfs_gemm_rows_fwd": [P, P, P, P, I, I, I, P],
    "rtfs_dp_convt_fwd": [P, P, P, P, I, I, I, P],
    "rtfs_dp_convt_fwd_form": [P, P, P, P, I, I, I, I, P],
    "rtfs_dp_convt_fwd_to": [P, P, P, P, P, I, I, I, P],
    "rtfs_attn_qkv_fwd": [P] * 14 + [I, I, P],
    "rtfs_attn_core_fwd": [P, P, P, P, P, I, I, P],
    "rtfs_attn_out_fwd": [P, P, P, F, P, P, P, P, I, I, P],
    "rtfs_attn_out_fwd_to": [P, P, P, F, P, P, P, P, P, I, I, P],
    "rtfs_tfar_mix_fwd": [P] * 13 + [I, I, I, I, I, P],
    "rtfs_dwconv_trio_fwd": [P] * 12 + [I, I, I, P],
    "rtfs_pool_add_fwd": [P] * 6 + [I, I, P],
    "rtfs_dwconv_gadd_fwd": [P] * 9 + [I, I, I, P],
    "rtfs_resid_fwd": [P] * 16 + [P, P, P, P, P, F, P, P, I, I, I, P],
    "rtfs_resid_proj_fwd": [P] * 16 + [P, P, P, P, P, F, P, P, P, P, P, P, I, I, I, I, P],
    "rtfs_resid_caf_fwd": [P] * 16 + [P, P, P, P, P, F] + [P] * 6 + [I, I, P, P, P, P, P, I, I, I, I, P],
    "rtfs_caf_video_fwd": [P] * 11 + [I, I, P],
    "rtfs_caf_fuse_fwd": [P] * 9 + [I, I, I, P],
    "rtfs_caf_video_bwd": [P] * 20 + [I, I, P],
    "rtfs_mask_fwd": [P, F, P, P, P, P, P, I, I, P],
    "rtfs_istft_fwd": [P, P, P, I, I, P],
    # ---- backward (training step) ----
    "rtfs_gemm_rows": [P, P, P, P, I, I, I, I, P],
    "rtfs_colsum_add": [P, P, LL, I, P],
    "rtfs_axpy": [P, F, P, LL, P],
    "rtfs_sum_n": [P, I, P, LL, P],
    "rtfs_gln_bwd_reduce": [P, P, P, P, P, I, F, P, P, P, P, I, I, I, P],
    "rtfs_gln_bwd_apply": [P, P, P, P, P, I, F, P, P, I, I, I, I, P],
    "rtfs_dwconv_bwd_input": [P, P, P, I, I, I, I, I, P],
    "rtfs_dwconv_bwd_weight": [P, P, P, P, P, F, I, I, P, P, I, I, I, P],
    "rtfs_pool_bwd": [P, P, I, I, I, P],
    "rtfs_d0_tail_bwd": [P] * 11 + [I, I, I, P],
    "rtfs_mix_bwd": [P] * 12 + [I, I, I, I, I, P],
    "rtfs_mix_gln_bwd": [P] * 18 + [I, I, I, I, I, P],
    "rtfs_expand_fwd": [P] * 17 + [I, I, I, P],
    "rtfs_proj_gateway_bwd": [P, P, P, P, P, P, F, P, I, P, I, P, P, P, LL, P],
    "rtfs_proj_gateway_bwd_next": [P, P, P, P, P, P, F, P, P, P, P, P, P, LL, P],
    "rtfs_wgrad": [P, I, P, I, P, I, P, LL, I, I, I, I, I, I, I, P, P, F, P, I, P],
    "rtfs_fold_gemm_bwd": [P, P, P, I, I, I, P],
    "rtfs_convt_bwd_input": [P, P, P, I, I, I, P],
    "rtfs_convt_bwd_input_form": [P, P, P, I, I, I, I, P],
    "rtfs_sru_scan_train_fwd": [P, P, P, P, F, P, P, I, I, I, P],
    "rtfs_sru_scan_bwd": [P, P, P, P, P, F, P, P, P, P, P, I, I, I, P],
    "rtfs_sru_scan_bwd2": [P, P, P, P, P, F, P, P, P, P, P, P, I, I, I, P],
    "rtfs_sru_layer_bwd_work_floats": [I],
    "rtfs_sru_layer_bwd": [P, P, P, P, P, P, F, P, P, P, P, P, P, P, P, I, I, P],
    "rtfs_ln4d_c_bwd": [P, P, P, P, P, P, LL, P],
    "rtfs_seq_gather": [P, P, P, I, P, I, I, I, P],
    "rtfs_attn_out_norm_bwd": [P, P, F, P, P, P, P, P, I, P],
    "rtfs_attn_qkv_norm_bwd": [P] * 16 + [I, I, P],
    "rtfs_attn_core_bwd": [P] * 10 + [I, I, P],
    "rtfs_transpose_tok": [P, P, I, P],
    "rtfs_mask_bwd_elem": [P, P, P, P, P, LL, P],
    "rtfs_prelu_bwd": [P, P, F, P, I, P, LL, P],
    "rtfs_gemm_prelu_bwd": [P, P, P, F, P, P, I, I, P],
    "rtfs_gemm_gln_relu_bwd_reduce": [P] * 10 + [I, I, P],
    "rtfs_chan_stats": [P, P, P, LL, P],
    "rtfs_caf_bwd_reduce": [P] * 11 + [I, I, I, P],
    "rtfs_caf_bwd_apply": [P] * 8 + [I, I, I, I, P],
    "rtfs_istft_bwd": [P, P, P, I, I, P],
    "rtfs_spec_patches": [P, P, I, I, P],
    # ---- VP block training step (csrc/vp_train.hip) ----
    "rtfs_vp_gate_proj_fwd": [P, P, P, F, P, P, P, P, P, I, I, P],
    "rtfs_vp_dwconv_fwd": [P, P, P, P, F, I, F, P, P, P, P, P, P, P, I, I, I, I, P],
    "rtfs_vp_pool_fwd": [P, P, P, P, I, I, I, I, F, F, F, F, P, I, I, P],
    "rtfs_vp_mix_fwd": [P, P, P, P, F, P, P, P, P, P, P, P, P, F, P, P, P, P, P, I, I, I, P],
    "rtfs_vp_resid_fwd": [P, P, P, P, P, I, I, P],
    "rtfs_vp_resid_bwd": [P, P, P, P, P, P, I, I, P],
    "rtfs_vp_mix_bwd": [P, P, P, P, P, F, P, P, P, P, F, P, P, P, P, I, I, I, P],
    "rtfs_vp_bn_bwd_reduce": [P, P, P, P, P, F, P, I, I, P],
    "rtfs_vp_dwconv_bwd": [P, P, P, P, P, F, P, F, I, P, P, P, P, F, I, F, P, P, P, P, I, P, I, I, I, I, P],
    "rtfs_vp_pool_bwd": [P, P, I, I, I, I, I, I, P],
    "rtfs_vp_gate_proj_bwd": [P, P, P, P, P, F, P, F, I, P, P, P, P, P, F, P, P, P, P, P, P, P, I, I, P],
    "rtfs_vp_attn_param_count": [],
    "rtfs_vp_attn_mask_size": [I],
    "rtfs_vp_attn_fwd": [P, P, P, P, P, I, I, P],
    "rtfs_vp_attn_long_work_floats": [I],
    "rtfs_vp_attn_long_fwd": [P, P, P, P, P, I, I, P],
    "rtfs_grad_sqnorm": [P, P, P, P, P, P, I, P, P],
    "rtfs_adamw_clip_step": [P, P, P, P, P, P, I, P, D, D, D, D, D, D, D, D, P],
    "rtfs_caf_bn_prepare": [P] * 23 + [F, F, P, P, P, P, P],
    "rtfs_caf_bn_adjoint": [P] * 19 + [P],
    "rtfs_dw_adjoint": [I, P, P, P, P, P, P, P, P, P, P, F, I, P, I, I, P, I, P, P, I, I, I, P],
    "rtfs_dw_adjoint_mix": [P] * 6 + [I, I, P, P, P, P, P, F, I, P, I, I, P, I, P, I, I, I, P],
    "rtfs_mix_gln_bwd_sig": [P] * 19 + [I, I, I, I, I, P],
    "rtfs_gln_stats": [P, P, I, LL, P],
    "rtfs_norm_act_fwd": [P, P, P, P, I, F, P, I, LL, I, P],
    "rtfs_gateway_fwd": [P, P, P, F, P, LL, P],
    "rtfs_cl_to_nchw": [P, P, I, I, I, P],
    "rtfs_nchw_to_cl": [P, P, I, I, I, P],
    "rtfs_vp_attn_bwd": [P, P, P, P, P, P, P, I, I, P],
    # ---- bf16 / split-bf16 MFMA variants of the inference path (extra int `terms` before the stream) ----
    "rtfs_bottleneck_fwd_bf16": [P, P, P, P, P, P, P, I, I, I, P],
    "rtfs_proj_fwd_bf16": [P, P, P, F, P, P, P, P, I, I, I, P],
    "rtfs_dp_unfold_gemm_fwd_bf16": [P, P, P, P, P, I, I, I, I, I, P],
    "rtfs_sru_layer_fwd_bf16": [P, P, P, P, F, P, P, P, I, I, I, P],
    "rtfs_dp_convt_fwd_bf16": [P, P, P, P, I, I, I, I, P],
    "rtfs_dp_convt_fwd_to_bf16": [P, P, P, P, P, I, I, I, I, P],
    "rtfs_attn_qkv_fwd_bf16": [P] * 14 + [I, I, I, P],
    "rtfs_attn_core_fwd_bf16": [P, P, P, P, P, I, I, I, P],
    "rtfs_attn_out_fwd_bf16": [P, P, P, F, P, P, P, P, I, I, I, P],
    "rtfs_attn_out_fwd_to_bf16": [P, P, P, F, P, P, P, P, P, I, I, I, P],
    "rtfs_resid_fwd_bf16": [P] * 16 + [P, P, P, P, P, F, P, P, I, I, I, I, P],
    "rtfs_resid_proj_fwd_bf16": [P] * 16 + [P, P, P, P, P, F, P, P, P, P, P, P, I, I, I, I, I, P],
    "rtfs_resid_caf_fwd_bf16": [P] * 16 + [P, P, P, P, P, F] + [P] * 6 + [I, I, P, P, P, P, P, I, I, I, I, I, P],
    "rtfs_mask_fwd_bf16": [P, F, P, P, P, P, P, I, I, I, P],
    "rtfs_gemm_rows_fwd_bf16": [P, P, P, P, I, I, I, I, P],
    "rtfs_gemm_rows_bf16": [P, P, P, P, I, I, I, I, I, P],
    "rtfs_wgrad_bf16": [P, I, P, I, P, I, P, LL, I, I, I, I, I, I, I, P, P, F, P, I, I, P],
    "rtfs_spread_defer": [I, P],
    "rtfs_spread_flush": [P],
    "rtfs_spread_lane": [I],
    "rtfs_decoder_mask_bwd": [P, P, P, P, P, P, LL, P],
    "rtfs_decoder_mask_bwd_bf16": [P, P, P, P, P, P, LL, I, P],
    "rtfs_proj_gateway_bwd_bf16": [P, P, P, P, P, P, F, P, I, P, I, P, P, P, LL, I, P],
    "rtfs_fold_gemm_bwd_bf16": [P, P, P, I, I, I, I, P],
    "rtfs_convt_bwd_input_bf16": [P, P, P, I, I, I, I, P],
}

_lib = None
_fn_cache = {}


def library_path() -> str:
    """librtfs_hip.so next to this file, or $RTFS_HIP_LIB (lets a relocated copy of `models` find it)."""
    return os.environ.get("RTFS_HIP_LIB", os.path.join(_HERE, LIB_NAME))


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension is not built. Run `python -m rtfs_net_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback for the product path."
            )
        lib = ctypes.CDLL(path)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.argtypes = args
            fn.restype = c_int
        _lib = lib
    return _lib


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("rtfs HIP kernels need contiguous device tensors")
    return t.data_ptr()


def ptr_array(tensors):
    arr = (c_void_p * len(tensors))(*[ptr(t) for t in tensors])
    return arr


_prof_name = None
_prof_pred = None
_prof_events = []


def profile_begin(name: str, pred=None):
    """Time every launch of ONE entry point with HIP events recorded on the launch stream (bench.py roofline).
    `pred(int_args)` (optional) narrows that to the launches whose integer arguments (shapes / modes) it accepts."""
    global _prof_name, _prof_events, _prof_labels, _prof_pred
    _prof_name, _prof_events, _prof_labels, _prof_pred = name, [], [], pred


def profile_end():
    """-> list of per-launch durations in ms (synchronises)."""
    global _prof_name
    _prof_name = None
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in _prof_events]


def profile_labels():
    """labels of the launches recorded by profile_begin("*"), in the order of profile_end()'s durations"""
    return list(_prof_labels)


def call(name: str, *args):
    """Invoke an entry point; tensors are converted to pointers, the stream is appended.

    The launch goes to the device that OWNS the tensors: every tensor argument must live on the same HIP device, the call runs
    with that device current (hipGetDevice-keyed library state, csrc/spread.hip) and on torch's current stream OF THAT DEVICE -
    a model moved with `.to('cuda:1')` works without `torch.cuda.set_device(1)`.
    (Hot path: at batch 1 the forward is 143 of these calls in 2.8 ms - the per-argument checks are written for speed, round 5.)"""
    conv = []
    keep = None
    dev = -1
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda or not a.is_contiguous():
                raise ValueError("rtfs HIP kernels need contiguous device tensors")
            d = a.get_device()
            if d != dev:
                if dev >= 0:
                    raise ValueError(f"{name}: tensor arguments live on different devices (cuda:{dev} and cuda:{d})")
                dev = d
            conv.append(a.data_ptr())
        elif isinstance(a, (list, tuple)):
            for t in a:
                if t is not None:
                    d = _same_device(name, dev, t)
                    dev = d
            arr = ptr_array(a)
            if keep is None:
                keep = []
            keep.append(arr)
            conv.append(ctypes.cast(arr, c_void_p))
        else:
            conv.append(a)
    if dev < 0:
        raise ValueError(f"{name}: no device tensor among the arguments")
    if dev != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return _launch(name, conv, args, dev)
    return _launch(name, conv, args, dev)


def spread_defer(on: bool, device) -> None:
    """rtfs_spread_defer on `device`'s current stream (no tensor argument to take the device from): the training step brackets each of its two
    backward stages with it, so the parameter-gradient reducers' finish launches are batched (include/rtfs_hip.h)"""
    device = torch.device(device)
    with torch.cuda.device(device):
        rc = load().rtfs_spread_defer(1 if on else 0, torch.cuda.current_stream(device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"rtfs_spread_defer failed with code {rc}")


def spread_lane(lane: int) -> None:
    """rtfs_spread_lane: scratch lane of the following reducer launches (1 = the weight-gradient side stream of the training step)"""
    rc = load().rtfs_spread_lane(int(lane))
    if rc != 0:
        raise RuntimeError(f"rtfs_spread_lane failed with code {rc}")


def _same_device(name, dev, t):
    """device index of tensor t, checked against `dev` (-1: none yet)"""
    if not t.is_cuda:
        raise ValueError("rtfs HIP kernels need contiguous device tensors")
    d = t.get_device()
    if dev >= 0 and d != dev:
        raise ValueError(f"{name}: tensor arguments live on different devices (cuda:{dev} and cuda:{d})")
    return d


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the current stream's handle without building a torch.cuda.Stream object (~2 us per launch)


def _launch(name, conv, args, dev):
    stream = _raw_stream(dev) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
    if ((name == _prof_name or name == str(_prof_name) + "_bf16") and (_prof_pred is None or _prof_pred(tuple(a for a in args if isinstance(a, int))))) or _prof_name == "*":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # torch's current stream == the stream handed to the kernel
        rc = getattr(load(), name)(*conv, stream)
        e1.record()
        _prof_events.append((e0, e1))
        if _prof_name == "*":  # tools/train_breakdown.py: label = entry point + its integer arguments (shapes / modes)
            _prof_labels.append(name + str(tuple(a for a in args if isinstance(a, int))))
    else:
        fn = _fn_cache.get(name)
        if fn is None:
            fn = _fn_cache[name] = getattr(load(), name)
        rc = fn(*conv, stream)
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc} ({'invalid argument' if rc == -1 else 'launch failure'})")
