"""ctypes binding of libstito_hip.so (include/stito_hip.h).

PyTorch owns device memory and streams; this module only passes raw pointers.  There is no
CPU fallback: if the library is missing or no HIP device is present, calls raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

import torch  # noqa: F401  (must be imported first: brings in the process' libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STITO_LIB_PATH") or os.path.join(_HERE, "_lib", "libstito_hip.so")  # env: A/B builds of the library

FX_PARAMETRIC_EQ, FX_COMPRESSOR, FX_DISTORTION, FX_DELAY, FX_REVERB, FX_GAIN, FX_NOISE_REVERB, FX_CHORUS = range(8)
NORM_NONE, NORM_MINMAX, NORM_BATCHNORM = range(3)
FX_FLAG_NORMALIZE_AFTER = 1  # stito_fx_desc.flags bit 0
CONV_DIRECT, CONV_WINOGRAD, CONV_WINOGRAD_F4, CONV_WINOGRAD_F4_PRE, CONV_WINOGRAD_F4_SPLIT, CONV_WINOGRAD_F4_SPLIT2 = 0, 1, 2, 3, 4, 5
# 6, 7: retired in ABI version 9 (split-precision experiments that never beat the kernels they were meant to replace)
CONV_WINOGRAD_F2_REG = 8
CONV_WINOGRAD_F4_SPLIT3 = 9
MAX_FX_PARAMS = 32
E_INVALID, E_UNSUPPORTED, E_WORKSPACE, E_HIP = -1, -2, -3, -4


class FxDesc(Structure):
    _fields_ = [
        ("kind", c_int32), ("num_channels", c_int32), ("w_offset", c_int32), ("has_bypass", c_int32),
        ("fixed_mask", c_uint32), ("flags", c_uint32), ("fixed_raw", c_double * MAX_FX_PARAMS),
        ("aux_dev", c_void_p), ("aux_len", c_int64),
    ]


class Frontend(Structure):
    _fields_ = [
        ("n_fft", c_int32), ("hop", c_int32), ("n_mels", c_int32), ("norm_mode", c_int32),
        ("window_dev", c_void_p), ("twiddle_dev", c_void_p), ("mel_start_dev", c_void_p),
        ("mel_len_dev", c_void_p), ("mel_off_dev", c_void_p), ("mel_w_dev", c_void_p),
        ("bn0_scale_dev", c_void_p), ("bn0_shift_dev", c_void_p),
        ("no_center", c_int32), ("mel_w_stride", c_int32),
    ]


class Cnn14Weights(Structure):
    _fields_ = [
        ("embed_dim", c_int32), ("n_mels", c_int32), ("channels", c_int32 * 7), ("reserved", c_int32),
        ("conv_w_dev", c_void_p * 12), ("conv_wino_dev", c_void_p * 12), ("conv_wino_algo", c_int32 * 12),
        ("bn_scale_dev", c_void_p * 12), ("bn_shift_dev", c_void_p * 12),
        ("fc_mid_wt_dev", c_void_p), ("fc_mid_b_dev", c_void_p),
        ("fc_side_wt_dev", c_void_p), ("fc_side_b_dev", c_void_p),
        ("reserved_ptr", c_void_p),
        ("conv1_f2reg_w_dev", c_void_p),
        ("conv_alt_dev", c_void_p * 12), ("conv_alt_algo", c_int32 * 12),
        ("chunk_streams", c_int32), ("chunk_first_conv", c_int32), ("chunk_last_conv", c_int32), ("reserved2", c_int32),
    ]


class StitoError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol declared in include/stito_hip.h
SIGNATURES = {
    "stito_last_error": (c_char_p, []),
    "stito_version": (c_int, []),
    "stito_fx_num_params": (c_int, [c_int]),
    "stito_chorus_lfo": (c_int, [ctypes.c_double, ctypes.c_double, c_int64, c_void_p, c_void_p]),
    "stito_dasp_compressor": (c_int, [c_void_p, c_int, c_int, c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, c_void_p, c_void_p]),
    "stito_chain_out_channels": (c_int, [POINTER(FxDesc), c_int, c_int]),
    "stito_chain_num_dims": (c_int, [POINTER(FxDesc), c_int]),
    "stito_render_workspace_bytes": (c_size_t, [POINTER(FxDesc), c_int, c_int, c_int64, c_int]),
    "stito_render_population": (c_int, [POINTER(FxDesc), c_int, c_void_p, c_int, c_int64, c_void_p, c_int, c_int,
                                        c_double, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "stito_render_population_multi": (c_int, [POINTER(FxDesc), c_int, c_void_p, c_int, c_int, c_int64, c_void_p, c_int,
                                              c_int, c_double, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "stito_peak": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "stito_normalize_audio": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "stito_num_frames": (c_int64, [c_int64, c_int]),
    "stito_logmel": (c_int, [POINTER(Frontend), c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "stito_cnn14_packed_conv_floats": (c_size_t, [c_int, c_int, c_int]),
    "stito_cnn14_pack_conv": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "stito_bn_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "stito_transpose": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "stito_cnn14_workspace_bytes": (c_size_t, [POINTER(Cnn14Weights), c_int, c_int64]),
    "stito_cnn14_forward": (c_int, [POINTER(Cnn14Weights), c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "stito_debug_wino_trace": (c_int, [c_void_p]),
    "stito_conv_timing_enable": (c_int, [c_int]),
    "stito_conv_timing_read": (c_int, [POINTER(ctypes.c_double), POINTER(c_int)]),
    "stito_conv_timing_read_each": (c_int, [POINTER(ctypes.c_double), c_int, POINTER(c_int)]),
    "stito_conv_timing_read_tagged": (c_int, [POINTER(ctypes.c_double), POINTER(c_int), c_int, POINTER(c_int)]),
    "stito_cnn14_packed_conv1_f2reg_floats": (c_size_t, []),
    "stito_cnn14_pack_conv1_f2reg": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "stito_conv_block1_f2reg_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "stito_conv_block1_f2reg_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "stito_conv_block1_f2reg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "stito_conv3x3_issued_flops": (c_double, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "stito_conv3x3_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "stito_conv3x3_bn_relu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p]),
    "stito_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "stito_conv3x3_bn_relu_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "stito_num_frames_nocenter": (c_int64, [c_int64, c_int, c_int]),
    "stito_mfcc_stats": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_int, ctypes.c_float, c_void_p, c_void_p]),
    "stito_rms_crest": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "stito_lufs_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "stito_lufs": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_double, c_void_p, c_void_p, c_size_t, c_void_p]),
    "stito_barkspectrum": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "stito_spectral_centroid_workspace_bytes": (c_size_t, [c_int, c_int, c_int64]),
    "stito_spectral_centroid": (c_int, [c_void_p, c_int, c_int, c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    "stito_embed_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "stito_resample_num_samples": (c_int64, [c_int64, c_int, c_int]),
    "stito_resample_sinc": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "stito_neg_cosine": (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.c_float, c_int, c_void_p, c_void_p]),
}


def lib():
    """Load libstito_hip.so once.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StitoError(
                f"{LIB_PATH} not found: build it with `make -C st-ito_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback for this path."
            )
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int):
    if rc == 0:
        return
    msg = lib().stito_last_error().decode("utf-8", "replace")
    if rc in (E_INVALID,):
        raise ValueError(msg)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise StitoError(f"libstito_hip error {rc}: {msg}")


def require_gpu():
    if not torch.cuda.is_available():
        raise StitoError("st_ito (MI355X build): no HIP device visible; this path has no CPU fallback")


def ptr(t):
    """Device (or host) pointer of a contiguous tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor passed to libstito_hip must be contiguous"
    return c_void_p(t.data_ptr())


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def mel_tables(melW, dev):
    """Device tables of a (n_bins, n_mels) mel filterbank for stito_frontend: every band is one run of consecutive
    non-zero bins.  -> (mel_start, mel_len, mel_off, mel_w, mel_w_stride) in the interleaved layout: weight i of band m
    at mel_w[i * n_mels + m] (zero padded to the longest run), so that the lanes of a wave -- one band each -- read
    consecutive floats per step."""
    import numpy as np
    import torch
    melW = np.asarray(melW, dtype=np.float32)
    n_mels = melW.shape[1]
    starts, lens = [], []
    for m in range(n_mels):
        nz = np.nonzero(melW[:, m])[0]
        s, e = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (0, 0)
        starts.append(s); lens.append(e - s)
    table = np.zeros((max(max(lens), 1), n_mels), np.float32)
    for m in range(n_mels):
        table[: lens[m], m] = melW[starts[m]: starts[m] + lens[m], m]
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)  # noqa: E731
    return i32(starts), i32(lens), i32(list(range(n_mels))), torch.from_numpy(table).to(dev).contiguous(), n_mels
