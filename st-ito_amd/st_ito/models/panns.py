"""AFx-Rep encoder (Cnn14) of the reference (st_ito/models/panns.py:121-281) with the forward
pass on hand-written HIP kernels.

The module keeps the reference's constructor signature and state_dict keys (including
torchlibrosa's frozen front-end parameters `spectrogram_extractor.stft.conv_{real,imag}.weight`
and `logmel_extractor.melW`) so that `load_param_model` can load the published afx-rep.ckpt with
strict=True.  forward() has no PyTorch implementation: it calls libstito_hip and raises when no
GPU / library is available.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import _hip


def _hann_periodic(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _dft_conv_kernels(n_fft: int, window: np.ndarray):
    """torchlibrosa STFT kernels: Re/Im of the windowed DFT matrix, (n_fft//2+1, 1, n_fft) float32."""
    nb = n_fft // 2 + 1
    k = np.arange(nb)[:, None]
    n = np.arange(n_fft)[None, :]
    ang = -2.0 * np.pi * ((k * n) % n_fft) / n_fft
    real = (np.cos(ang) * window[None, :]).astype(np.float32)[:, None, :]
    imag = (np.sin(ang) * window[None, :]).astype(np.float32)[:, None, :]
    return torch.from_numpy(real), torch.from_numpy(imag)


def _mel_filterbank(sr: float, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel (Slaney scale, slaney norm) -> (n_mels, n_fft//2+1) float32; what
    torchlibrosa's LogmelFilterBank stores (transposed) as melW (panns.py:158-168)."""
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


class _Holder(nn.Module):
    """Namespace module so that parameter names match torchlibrosa's."""


class ConvBlock(nn.Module):
    """Parameter container for reference panns.py:25-80 (conv1/bn1/conv2/bn2)."""

    def __init__(self, in_channels: int, out_channels: int, use_batchnorm: bool = True):
        super().__init__()
        self.use_batchnorm = use_batchnorm
        self.conv1 = nn.Conv2d(in_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels) if use_batchnorm else nn.Identity()
        self.bn2 = nn.BatchNorm2d(out_channels) if use_batchnorm else nn.Identity()
        nn.init.xavier_uniform_(self.conv1.weight)
        nn.init.xavier_uniform_(self.conv2.weight)


class Cnn14(nn.Module):
    def __init__(self, embed_dim: int, sample_rate: float, window_size: int, hop_size: int, mel_bins: int,
                 fmin: float, fmax: float, use_batchnorm: bool = False, input_norm: str = "batchnorm"):
        super().__init__()
        if input_norm not in ("batchnorm", "minmax", "none"):
            raise ValueError(f"Invalid input_norm: {input_norm}")
        self.embed_dim, self.use_batchnorm, self.input_norm = embed_dim, use_batchnorm, input_norm
        self.sample_rate, self.window_size, self.hop_size, self.mel_bins = sample_rate, window_size, hop_size, mel_bins

        # torchlibrosa-compatible frozen front-end parameters
        self.spectrogram_extractor = _Holder()
        self.spectrogram_extractor.stft = _Holder()
        nb = window_size // 2 + 1
        self.spectrogram_extractor.stft.conv_real = nn.Conv1d(1, nb, window_size, stride=hop_size, bias=False)
        self.spectrogram_extractor.stft.conv_imag = nn.Conv1d(1, nb, window_size, stride=hop_size, bias=False)
        r, i = _dft_conv_kernels(window_size, _hann_periodic(window_size))
        self.spectrogram_extractor.stft.conv_real.weight.data = r
        self.spectrogram_extractor.stft.conv_imag.weight.data = i
        self.logmel_extractor = _Holder()
        self.logmel_extractor.melW = nn.Parameter(
            torch.from_numpy(_mel_filterbank(sample_rate, window_size, mel_bins, fmin, fmax).T.copy()))
        for p in list(self.spectrogram_extractor.parameters()) + list(self.logmel_extractor.parameters()):
            p.requires_grad = False

        self.bn0 = nn.BatchNorm2d(mel_bins)
        chans = [1, 64, 128, 256, 512, 1024, 2048]
        for b in range(6):
            setattr(self, f"conv_block{b + 1}", ConvBlock(chans[b], chans[b + 1], use_batchnorm))
        self.fc_mid = nn.Linear(2048, embed_dim, bias=True)
        self.fc_side = nn.Linear(2048, embed_dim, bias=True)
        nn.init.xavier_uniform_(self.fc_mid.weight)
        nn.init.xavier_uniform_(self.fc_side.weight)
        self.fc_mid.bias.data.fill_(0.0)
        self.fc_side.bias.data.fill_(0.0)

        self._packed = None      # device-side packed weights, built lazily
        self._packed_key = None
        self._ws = None
        self.max_streams_per_pass = int(os.environ.get("STITO_MAX_STREAMS", "512"))
        # 3x3 conv algorithm of the trunk: "winograd_f4" (F(4x4,3x3), default), "winograd" (F(2x2,3x3)) or "direct"
        self.conv_algo = {"direct": _hip.CONV_DIRECT, "winograd": _hip.CONV_WINOGRAD, "winograd_f4": _hip.CONV_WINOGRAD_F4}[
            os.environ.get("STITO_CONV_ALGO", "winograd_f4")]
        # F(4x4,3x3) layers with at least this many output channels run with the input transform hoisted into its own
        # pass (CONV_WINOGRAD_F4_PRE): from 512 up the 8+ workgroups sharing a pixel block stop repeating it (0 = never)
        self.conv_pre_min_cout = int(os.environ.get("STITO_CONV_PRE_MIN_COUT", "512"))
        # ... and, unless STITO_CONV_SPLIT=0, on the f16 matrix pipe with split operands (CONV_WINOGRAD_F4_SPLIT: every f32
        # operand as f16 hi + lo, three products, f32 accumulation -- as accurate as the f32 pipe, measured) where cin % 64 == 0
        self.conv_split = os.environ.get("STITO_CONV_SPLIT", "1") != "0"
        # from 256 output channels up (measured at 512 streams: 128 -> 256 channels 4.0 -> 3.6 ms, 256 -> 256 6.4 -> 5.7 ms; below,
        # the transformed input's round trip through HBM costs more than the matrix pipe saves)
        self.conv_split_min_cout = int(os.environ.get("STITO_CONV_SPLIT_MIN_COUT", "256"))
        # ... on 64 x 64 workgroup tiles in two sweeps (CONV_WINOGRAD_F4_SPLIT2) from this many INPUT channels up: its longer
        # epilogue (two sweeps, the first one's outputs parked in HBM) is repaid by a third fewer bytes per MAC only when the
        # channel loop is long (measured at 512 streams: 512 -> 512 4.45 -> 4.20 ms, 2048 -> 2048 4.05 -> 3.05; 256 -> 512 2.51 -> 2.66)
        self.conv_split2_min_cin = int(os.environ.get("STITO_CONV_SPLIT2_MIN_CIN", "512"))
        # ... and on 128 x 128 tiles in SIX sweeps (CONV_WINOGRAD_F4_SPLIT3: half the bytes per MAC again; the rows' contributions
        # meet in a scratch area) from this many input channels up where cout % 512 == 0: measured at 512 streams 512 -> 512
        # 3.73 -> 3.42 ms, 512 -> 1024 2.04 -> 1.92, 1024 -> 1024 3.02 -> 2.57, 1024 -> 2048 1.73 -> 1.42, 2048 -> 2048 3.02 -> 2.40
        # (256 -> 512: 2.35 -> 2.85, the channel loop is too short for six epilogues); 0 = never
        self.conv_split3_min_cin = int(os.environ.get("STITO_CONV_SPLIT3_MIN_CIN", "512"))
        # the 64-input-channel layers (conv_block1.conv2, conv_block2.conv1) by Winograd F(2x2,3x3) on the f16 pipe with the
        # transformed weights resident in registers and the input transform done in registers (CONV_WINOGRAD_F2_REG), unless
        # STITO_CONV_F2REG=0
        self.conv_f2reg = os.environ.get("STITO_CONV_F2REG", "1") != "0"
        # conv_block1 as ONE launch of that kernel (the first conv computed into its patch ring); STITO_CONV_FUSE1=0: two launches
        self.conv_fuse1 = os.environ.get("STITO_CONV_FUSE1", "1") != "0"
        # depth-first schedule (stito_cnn14_weights.chunk_*, ABI v10): STITO_TRUNK_CHUNK = streams per chunk (0 = layer by layer),
        # STITO_TRUNK_CHUNK_CONVS = "first:last" conv indices of the run (conv_block<b>.conv<j> = 2 (b - 1) + (j - 1))
        self.trunk_chunk = int(os.environ.get("STITO_TRUNK_CHUNK", "0"))
        first, _, last = os.environ.get("STITO_TRUNK_CHUNK_CONVS", "2:6").partition(":")
        self.trunk_chunk_convs = (int(first), int(last or first))

    # ------------------------------------------------------------------------------------
    def _invalidate(self):
        self._packed = None

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def prepare(self):
        """Pack conv weights, fold BN, transpose FC weights and build the front-end tables on
        the device (all through libstito_hip kernels / tiny host tables)."""
        _hip.require_gpu()
        dev = self._device()
        if dev.type != "cuda":
            raise _hip.StitoError("Cnn14 (MI355X build) must live on the GPU: call load_param_model(use_gpu=True) "
                                  "or model.cuda(); there is no CPU forward")
        if self.training:
            raise NotImplementedError("Cnn14 (MI355X build) is inference-only (model.eval())")
        L = _hip.lib()
        st = _hip.stream_ptr()
        keep = []
        W = _hip.Cnn14Weights()
        W.embed_dim, W.n_mels = self.embed_dim, self.mel_bins
        W.chunk_streams, (W.chunk_first_conv, W.chunk_last_conv) = self.trunk_chunk, self.trunk_chunk_convs
        chans = [1, 64, 128, 256, 512, 1024, 2048]
        for i, c in enumerate(chans):
            W.channels[i] = c
        for b in range(6):
            blk = getattr(self, f"conv_block{b + 1}")
            for j, (conv, bn) in enumerate(((blk.conv1, blk.bn1), (blk.conv2, blk.bn2))):
                w = conv.weight.detach().to(torch.float32).contiguous()
                cout, cin = w.shape[0], w.shape[1]
                packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, _hip.CONV_DIRECT), dtype=torch.float32, device=dev)
                _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, _hip.CONV_DIRECT, _hip.ptr(packed), st))
                if self.conv_algo != _hip.CONV_DIRECT and cin % 8 == 0 and cout % 64 == 0:
                    pre = self.conv_algo == _hip.CONV_WINOGRAD_F4 and 0 < self.conv_pre_min_cout <= cout
                    split = (self.conv_algo == _hip.CONV_WINOGRAD_F4 and self.conv_split and 0 < self.conv_split_min_cout <= cout and
                             cin % 64 == 0 and cout % 256 == 0 and (cout < 1024 or cout % 512 == 0))
                    algo = _hip.CONV_WINOGRAD_F4_SPLIT if split else (_hip.CONV_WINOGRAD_F4_PRE if pre else self.conv_algo)
                    if split and 0 < self.conv_split2_min_cin <= cin:
                        algo = _hip.CONV_WINOGRAD_F4_SPLIT2
                    if split and 0 < self.conv_split3_min_cin <= cin and cout % 512 == 0:
                        algo = _hip.CONV_WINOGRAD_F4_SPLIT3
                    if self.conv_algo == _hip.CONV_WINOGRAD_F4 and self.conv_split and self.conv_f2reg and cin == 64 and cout % 64 == 0:
                        algo = _hip.CONV_WINOGRAD_F2_REG
                    upk = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), dtype=torch.float32, device=dev)
                    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, algo, _hip.ptr(upk), st))
                    W.conv_wino_dev[2 * b + j] = upk.data_ptr()
                    W.conv_wino_algo[2 * b + j] = algo
                    keep.append(upk)
                    if algo == _hip.CONV_WINOGRAD_F4_SPLIT3:
                        # small batches (fewer than 3 / 4 of the six-sweep kernel's workgroups per CU: a population of 32 has 32 of
                        # them for conv_block6) run the two-sweep kernel: its packing rides along (stito_cnn14_forward chooses per call)
                        alt = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, _hip.CONV_WINOGRAD_F4_SPLIT2), dtype=torch.float32, device=dev)
                        _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, _hip.CONV_WINOGRAD_F4_SPLIT2, _hip.ptr(alt), st))
                        W.conv_alt_dev[2 * b + j] = alt.data_ptr()
                        W.conv_alt_algo[2 * b + j] = _hip.CONV_WINOGRAD_F4_SPLIT2
                        keep.append(alt)
                scale = torch.empty(cout, dtype=torch.float32, device=dev)
                shift = torch.empty(cout, dtype=torch.float32, device=dev)
                if isinstance(bn, nn.BatchNorm2d):
                    args = [t.detach().to(torch.float32).contiguous() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
                    _hip.check(L.stito_bn_fold(*[_hip.ptr(t) for t in args], float(bn.eps), cout, _hip.ptr(scale), _hip.ptr(shift), st))
                    keep += args
                else:
                    _hip.check(L.stito_bn_fold(None, None, None, None, 0.0, cout, _hip.ptr(scale), _hip.ptr(shift), st))
                idx = 2 * b + j
                W.conv_w_dev[idx], W.bn_scale_dev[idx], W.bn_shift_dev[idx] = packed.data_ptr(), scale.data_ptr(), shift.data_ptr()
                keep += [w, packed, scale, shift]
        if (self.conv_fuse1 and int(W.conv_wino_algo[1]) == _hip.CONV_WINOGRAD_F2_REG and W.conv_wino_dev[1] and
                self.conv_block1.conv1.weight.shape[:2] == (64, 1)):
            # conv_block1 in one launch (stito_conv_block1_f2reg): the first conv is computed on the matrix pipe into the second
            # conv's patch ring, in the slots where the unfused kernel issues its copies; the 64-channel full-resolution map
            # (7.9 GB at 512 streams, written once and read 1.5 times) never exists
            w1 = self.conv_block1.conv1.weight.detach().to(torch.float32).contiguous()
            fw = torch.empty(L.stito_cnn14_packed_conv1_f2reg_floats(), dtype=torch.float32, device=dev)
            _hip.check(L.stito_cnn14_pack_conv1_f2reg(_hip.ptr(w1), W.bn_scale_dev[0], W.bn_shift_dev[0], 64, _hip.ptr(fw), st))
            W.conv1_f2reg_w_dev = fw.data_ptr()
            keep += [w1, fw]
        for name in ("mid", "side"):
            fc = getattr(self, f"fc_{name}")
            w = fc.weight.detach().to(torch.float32).contiguous()           # (E, 2048)
            wt = torch.empty((w.shape[1], w.shape[0]), dtype=torch.float32, device=dev)
            _hip.check(L.stito_transpose(_hip.ptr(w), w.shape[0], w.shape[1], _hip.ptr(wt), st))
            b = fc.bias.detach().to(torch.float32).contiguous()
            setattr(W, f"fc_{name}_wt_dev", wt.data_ptr())
            setattr(W, f"fc_{name}_b_dev", b.data_ptr())
            keep += [w, wt, b]

        # ---- front end tables ---------------------------------------------------------------
        n_fft = self.window_size
        cr = self.spectrogram_extractor.stft.conv_real.weight.detach().cpu().numpy()[:, 0, :]
        ci = self.spectrogram_extractor.stft.conv_imag.weight.detach().cpu().numpy()[:, 0, :]
        window = cr[0].astype(np.float64)  # bin 0: cos(0) * window
        # the FFT path assumes the stored kernels ARE a windowed DFT; verify on a few bins
        nn_ = np.arange(n_fft)
        for kbin in (1, 7, n_fft // 4, n_fft // 2):
            ang = -2.0 * np.pi * ((kbin * nn_) % n_fft) / n_fft
            if (np.abs(cr[kbin] - np.cos(ang) * window).max() > 1e-4 or np.abs(ci[kbin] - np.sin(ang) * window).max() > 1e-4):
                raise NotImplementedError("checkpoint STFT kernels are not a windowed DFT; unsupported front-end")
        FE = _hip.Frontend()
        FE.n_fft, FE.hop, FE.n_mels = n_fft, self.hop_size, self.mel_bins
        FE.norm_mode = {"none": _hip.NORM_NONE, "minmax": _hip.NORM_MINMAX, "batchnorm": _hip.NORM_BATCHNORM}[self.input_norm]
        win_t = torch.from_numpy(window.astype(np.float32)).to(dev)
        kk = np.arange(n_fft // 2)
        tw = np.stack([np.cos(-2 * np.pi * kk / n_fft), np.sin(-2 * np.pi * kk / n_fft)], 1).astype(np.float32)
        tw_t = torch.from_numpy(tw).to(dev).contiguous()
        melW = self.logmel_extractor.melW.detach().cpu().numpy()  # (n_bins, n_mels)
        ms, ml, mo, mw, FE.mel_w_stride = _hip.mel_tables(melW, dev)
        FE.window_dev, FE.twiddle_dev = win_t.data_ptr(), tw_t.data_ptr()
        FE.mel_start_dev, FE.mel_len_dev, FE.mel_off_dev, FE.mel_w_dev = ms.data_ptr(), ml.data_ptr(), mo.data_ptr(), mw.data_ptr()
        keep += [win_t, tw_t, ms, ml, mo, mw]
        if self.input_norm == "batchnorm":
            bn = self.bn0
            sc = torch.empty(self.mel_bins, dtype=torch.float32, device=dev)
            sh = torch.empty(self.mel_bins, dtype=torch.float32, device=dev)
            args = [t.detach().to(torch.float32).contiguous() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
            _hip.check(L.stito_bn_fold(*[_hip.ptr(t) for t in args], float(bn.eps), self.mel_bins, _hip.ptr(sc), _hip.ptr(sh), st))
            FE.bn0_scale_dev, FE.bn0_shift_dev = sc.data_ptr(), sh.data_ptr()
            keep += args + [sc, sh]
        self._packed = (W, FE, keep)
        return self._packed

    def _ensure(self):
        if self._packed is None:
            self.prepare()
        return self._packed

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != self._device():
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self._device())
        return self._ws

    # ------------------------------------------------------------------------------------
    def logmel(self, audio: torch.Tensor, peaks: Optional[torch.Tensor] = None, norm_passes: int = 0,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B, C, L) float32 on the GPU -> normalised log-mel (B*C, T, M) (panns.py:213-245); `out`: the (contiguous) slice of a
        larger population's log-mel buffer to write into."""
        W, FE, _ = self._ensure()
        B, C, n = audio.shape
        if C not in (1, 2):
            raise ValueError(f"Invalid number of channels: {C}")
        L = _hip.lib()
        T = L.stito_num_frames(n, self.hop_size)
        if out is None:
            out = torch.empty((B * C, T, self.mel_bins), dtype=torch.float32, device=audio.device)
        assert out.shape == (B * C, T, self.mel_bins) and out.is_contiguous() and out.dtype == torch.float32
        assert audio.is_contiguous() and (peaks is None or peaks.is_contiguous())
        _hip.check(L.stito_logmel(FE, _hip.ptr(audio), _hip.ptr(peaks), norm_passes, B, C, n, _hip.ptr(out), _hip.stream_ptr()))
        return out

    def trunk(self, logmel: torch.Tensor, n_cand: int, channels: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """log-mel (n_cand*channels, T, M) -> raw (mid, side) fc outputs (panns.py:250-281)."""
        W, FE, _ = self._ensure()
        L = _hip.lib()
        T = logmel.shape[1]
        dev = logmel.device
        mid = torch.empty((n_cand, self.embed_dim), dtype=torch.float32, device=dev)
        side = torch.empty((n_cand, self.embed_dim), dtype=torch.float32, device=dev)
        need = L.stito_cnn14_workspace_bytes(W, n_cand * channels, T)
        ws = self._workspace(need)
        _hip.check(L.stito_cnn14_forward(W, _hip.ptr(logmel), n_cand, channels, T, _hip.ptr(mid), _hip.ptr(side),
                                         _hip.ptr(ws), ws.numel(), _hip.stream_ptr()))
        return mid, side

    def embed_raw(self, audio: torch.Tensor, peaks: Optional[torch.Tensor], norm_passes: int):
        """audio (B, C, L) on the GPU (+ per-item peaks) -> raw (mid, side), sub-batched so that the
        activation workspace stays bounded."""
        audio = audio.contiguous()
        B, C, _ = audio.shape
        step = max(1, self.max_streams_per_pass // C)
        mids, sides = [], []
        for b0 in range(0, B, step):
            a = audio[b0:b0 + step]
            p = peaks[b0:b0 + step].contiguous() if peaks is not None else None
            lm = self.logmel(a, p, norm_passes)
            m, s = self.trunk(lm, a.shape[0], C)
            mids.append(m); sides.append(s)
        return (torch.cat(mids), torch.cat(sides)) if len(mids) > 1 else (mids[0], sides[0])

    def forward(self, x: torch.Tensor):
        """input: waveform (batch_size, chs, seq_len) -> (mid_embed, side_embed)  (panns.py:209-281)."""
        _hip.require_gpu()
        if x.dim() != 3:
            raise ValueError("expected (batch, chs, seq_len)")
        if x.shape[1] not in (1, 2):
            raise ValueError(f"Invalid number of channels: {x.shape[1]}")
        dev = self._device()
        xin = x.detach().to(dev, torch.float32)
        mid, side = self.embed_raw(xin, None, 0)
        return mid, side
