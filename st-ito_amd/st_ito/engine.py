"""Host-side orchestration of the HIP hot path: plugin dict -> chain descriptor, population
rendering, embedding, loss.  Everything numeric happens in libstito_hip; torch only owns the
buffers and the stream.

Reference call sites replaced: the loop `for w in W: process_audio(...)`, the `embed_func` call
and the cosine loss in run_es.evaluate (st_ito/style_transfer.py:504-573).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _hip

CROP_LEN = 262144  # style_transfer.py:505


# --------------------------------------------------------------------------------------------
# plugin dict (reference schema) -> stito_fx_desc[]
# --------------------------------------------------------------------------------------------
def _instance_of(plugin: dict):
    if "instance" not in plugin:
        if "vst_filepath" in plugin:
            raise NotImplementedError("VST plugins (pedalboard.load_plugin) are not supported in this build; "
                                      "use --effect-type basic")
        elif "class_path" in plugin:
            plugin["instance"] = plugin["class_path"]()
        else:
            raise ValueError("Plugin must contain 'vst_filepath' or 'class_path'.")
    return plugin["instance"]


def compile_chain(plugins: Dict[str, dict], normalize_stages: bool = False) -> Tuple[ctypes.Array, int]:
    """Compile the reference's `plugins` dict into the C chain descriptor.

    Mirrors how process_audio walks the dict (style_transfer.py:65-92): every name in
    plugin["parameter_names"] consumes one slot of w; "our_bypass" is consumed and ignored;
    names listed in plugin["fixed_parameters"] take the fixed value (via Parameter.set_value)
    but still consume their slot."""
    descs = (_hip.FxDesc * max(1, len(plugins)))()
    descs._keep = []  # device buffers the descriptor points into
    off = 0
    for i, (plugin_name, plugin) in enumerate(plugins.items()):
        if "vst_filepath" in plugin and "class_path" not in plugin:
            raise NotImplementedError("VST plugins are not supported in this build; use --effect-type basic")
        inst = _instance_of(plugin)
        kind = getattr(inst, "KIND", -1)
        if kind < 0:
            raise ValueError(f"Plugin {plugin_name}: {type(inst).__name__} is not an effect of this build")
        names = list(plugin.get("parameter_names") or list(inst.parameters.keys()))
        has_bypass = 1 if (names and names[0] == "our_bypass") else 0
        real = names[has_bypass:]
        if "our_bypass" in real or real != list(inst.parameters.keys()):
            raise ValueError(f"Plugin {plugin_name}: parameter_names {names} do not match {list(inst.parameters)}")
        d = descs[i]
        d.kind, d.num_channels, d.w_offset, d.has_bypass = kind, int(plugin["num_channels"]), off, has_bypass
        if d.num_channels not in (1, 2):
            raise ValueError(f"Plugin {plugin_name}: num_channels must be 1 or 2")
        mask = 0
        for p, name in enumerate(real):
            if name in plugin.get("fixed_parameters", {}):
                prm = inst.parameters[name]
                prm.set_value(plugin["fixed_parameters"][name])  # asserts the range like the reference
                d.fixed_raw[p] = prm.raw_value
                mask |= 1 << p
        d.fixed_mask = mask
        d.flags = _hip.FX_FLAG_NORMALIZE_AFTER if normalize_stages else 0  # style_transfer.py:106-107
        if kind == _hip.FX_NOISE_REVERB:  # the band-filtered noise bank is an input of the stage
            _hip.require_gpu()
            bank = inst.noise_bank_device(torch.device("cuda", torch.cuda.current_device()))
            descs._keep.append(bank)
            d.aux_dev, d.aux_len = bank.data_ptr(), bank.shape[-1]
        off += len(names)
    return descs, off


class _Workspace:
    """Grow-only device scratch buffers keyed by name (torch owns the memory)."""

    def __init__(self):
        self._bufs: Dict[str, torch.Tensor] = {}

    def get(self, name: str, nbytes: int, device) -> torch.Tensor:
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes or b.device != device:
            b = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._bufs[name] = b
        return b


_WS = _Workspace()


def render_population(plugins: Dict[str, dict], x: torch.Tensor, W: torch.Tensor, sample_rate: float,
                      chain=None, out=None, ws_key: str = "render") -> Tuple[torch.Tensor, torch.Tensor]:
    """x (C, L) float32 on the GPU -- or (B, C, L): B inputs, candidate p reads input p // (P // B) --
    W (P, D) float64 on the GPU -> (audio (P, C', L) before peak normalisation, peaks (P,)).
    out: (audio, peaks) to render into (contiguous slices of a larger population's buffers); ws_key: name of the
    workspace this call uses (two renders in flight on different streams need different ones)."""
    _hip.require_gpu()
    L = _hip.lib()
    descs, ndims = chain if chain is not None else compile_chain(plugins)
    n_fx = len(plugins)
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() in (2, 3)
    assert W.is_cuda and W.dtype == torch.float64 and W.dim() == 2
    x = x.contiguous()
    W = W.contiguous()
    if W.shape[1] != ndims:
        raise ValueError(f"parameter vector has {W.shape[1]} dims, chain consumes {ndims}")
    n_inputs = x.shape[0] if x.dim() == 3 else 1
    C, n = x.shape[-2:]
    P = W.shape[0]
    if P % n_inputs:
        raise ValueError(f"population {P} is not a multiple of the number of inputs {n_inputs}")
    c_out = L.stito_chain_out_channels(descs, n_fx, C)
    if out is None:
        audio = torch.empty((P, c_out, n), dtype=torch.float32, device=x.device)
        peaks = torch.empty((P,), dtype=torch.float32, device=x.device)
    else:
        audio, peaks = out
        assert audio.shape == (P, c_out, n) and audio.is_contiguous() and audio.dtype == torch.float32 and peaks.shape == (P,)
    for i in range(n_fx):  # the chorus stage reads its LFO from a table that has to cover this length
        if descs[i].kind == _hip.FX_CHORUS:
            # re-fetched on every call: the shared table belongs to one (sample rate, device) and is replaced when a longer render
            # grows it, so a descriptor compiled earlier may point at the old one (or at another rate's).  The descriptor keeps
            # every table it has pointed at alive (descs._keep): a launch already queued on the stream may still read it
            from .effects import chorus_lfo_device
            t = chorus_lfo_device(sample_rate, n, x.device)
            if descs[i].aux_dev != t.data_ptr() or descs[i].aux_len != t.numel():
                descs[i].aux_dev, descs[i].aux_len = t.data_ptr(), t.numel()
                if hasattr(descs, "_keep"):
                    descs._keep.append(t)
    need = L.stito_render_workspace_bytes(descs, n_fx, C, n, P)
    ws = _WS.get(ws_key, need, x.device)
    _hip.check(L.stito_render_population_multi(descs, n_fx, _hip.ptr(x), n_inputs, C, n, _hip.ptr(W), P, ndims,
                                               float(sample_rate), _hip.ptr(audio), _hip.ptr(peaks), _hip.ptr(ws),
                                               ws.numel(), _hip.stream_ptr()))
    return audio, peaks


def normalize_audio_(audio: torch.Tensor, peaks: torch.Tensor) -> torch.Tensor:
    """In place: audio[p] /= clip(peaks[p], 1e-8)  (style_transfer.py:113)."""
    P, C, n = audio.shape
    _hip.check(_hip.lib().stito_normalize_audio(_hip.ptr(audio), P, C, n, _hip.ptr(peaks), _hip.stream_ptr()))
    return audio


def render_single(instance, x: np.ndarray, sample_rate: float) -> np.ndarray:
    """Basic*.process(x, sample_rate): one effect, current parameter values, (chs, n) -> (chs', n)."""
    _hip.require_gpu()
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError("process expects a (chs, n) array")
    # .process() is called with exactly the channels the effect should see; a stereo effect
    # called with mono audio is fed the mono signal as in pedalboard (no up-mix here).
    nch = 2 if ((x.shape[0] == 2 and instance.NUM_CHANNELS == 2) or getattr(instance, "ALWAYS_STEREO", False)) else 1
    plugins = {"fx": {"class_path": type(instance), "instance": instance, "num_channels": nch,
                      "fixed_parameters": {}, "parameter_names": list(instance.parameters.keys()),
                      "num_params": len(instance.parameters)}}
    w = np.array([[p.raw_value for p in instance.parameters.values()]], dtype=np.float64)
    dev = torch.device("cuda", torch.cuda.current_device())
    audio, _ = render_population(plugins, torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), sample_rate)
    return audio[0].cpu().numpy()


def process_audio_gpu(x: np.ndarray, w: np.ndarray, sr: int, plugins: Dict[str, dict],
                      normalize_stages: bool = False) -> np.ndarray:
    """process_audio for one parameter vector (style_transfer.py:45-115)."""
    _hip.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    xt = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    wt = torch.as_tensor(np.asarray(w, dtype=np.float64)[None, :]).to(dev)
    audio, peaks = render_population(plugins, xt, wt, sr, chain=compile_chain(plugins, normalize_stages))
    normalize_audio_(audio, peaks)
    return audio[0].cpu().numpy()


# --------------------------------------------------------------------------------------------
# population evaluation
# --------------------------------------------------------------------------------------------
class PopulationEvaluator:
    """GPU replacement of run_es.evaluate (style_transfer.py:474-573) for the AFx-Rep metric.

    One instance per run_es call: holds the (padded) input on the device, the compiled chain
    and the target embeddings.  evaluate(W) returns the fitness list; embeddings and (lazily)
    normalised audio are available for --savepop.

    Multi-pair batches (BASELINE.json configs[2]): x may hold B inputs (B, C, L) with B target
    embeddings (B, E); evaluate then takes the B populations stacked pair-major, (B * P, D), and
    scores the candidates of pair b against target b."""

    def __init__(self, x: torch.Tensor, sample_rate: int, plugins: Dict[str, dict], model, target_embeds: dict,
                 device: Optional[torch.device] = None, max_candidates_per_pass: Optional[int] = None,
                 embed_func=None, normalize_stages: bool = False, entry_weights: Optional[Dict[str, float]] = None,
                 use_graph: Optional[bool] = None, capture_after: Optional[int] = None):
        """embed_func: None or st_ito.utils.get_param_embeds -> the fused AFx-Rep path (render -> log-mel with the
        peak normalisations folded into the STFT loader -> Cnn14 -> loss).  Any other embed_func(x, model, sample_rate)
        -> dict of (P, E_k) embeddings (the MIR / MFCC metrics of st_ito.utils, or a user function working on GPU
        tensors) takes the generic path of style_transfer.py:531-571: the rendered population is peak-normalised in
        HBM, handed to embed_func as one (P, C, L) GPU tensor, and every entry of the returned dict is scored against
        the target's entry of the same name.  use_graph: None = STITO_GRAPH (default on), False = eager launches only
        (bench.py's per-launch event timing needs host-side launches)."""
        _hip.require_gpu()
        from . import utils as _utils

        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.sample_rate = sample_rate
        self.plugins = plugins
        self.model = model
        self.chain = compile_chain(plugins, normalize_stages)
        self.ndims = self.chain[1]
        assert x.dim() == 3, "input audio must be (batch, chs, seq_len)"
        self.n_inputs = x.shape[0]
        self.x_full = x.to(self.device, torch.float32).contiguous()
        self.embed_func = embed_func
        # the fused path embeds the rendered audio as 48 kHz audio; at any other rate the reference resamples inside
        # get_param_embeds (utils.py:462-463), on both sides of the distance: that goes through the generic path
        self.fused = (embed_func is None or embed_func is _utils.get_param_embeds) and int(sample_rate) == 48000
        if embed_func is None and not self.fused:
            self.embed_func = _utils.get_param_embeds
        self.targets = {k: v.detach().to(self.device, torch.float32).contiguous().view(-1, v.shape[-1])
                        for k, v in target_embeds.items()}
        for k, v in self.targets.items():
            if v.shape[0] != self.n_inputs:
                raise ValueError(f"{self.n_inputs} inputs but {v.shape[0]} target embeddings ({k})")
        if self.fused:
            self.tmid, self.tside = self.targets["mid"], self.targets["side"]
        # generic path: weight of every entry of the embedding dict inside the mean over entries (style_transfer.py:560-568
        # counts a content embedding's distance twice: dists.append(2 * dist)); default 1
        self.entry_weights = dict(entry_weights or {})
        self.max_cand = max_candidates_per_pass
        self.flags = torch.zeros((256, 2), dtype=torch.int32, device=self.device)  # NaN flags, one row per loss call
        # The plain fused call -- one population per pass, no crop, no dropout, no audio handed back: what run_es issues every
        # iteration -- is captured as ONE hipGraph (render -> log-mel -> Cnn14 -> loss: ~45 launches) and replayed; W travels
        # through a static device buffer.  STITO_GRAPH=0 keeps the eager launches.  On an idle host a replay buys little (the
        # eager launches already run ahead of the device: pop 32 6.56 against 6.60 ms per step, pop 256 43.0 against 43.1); what it
        # removes is the step's dependence on the host's launch rate: one launch per step and rank instead of ~45.
        # (Round 4 had this off: every replay after the first kept the previous replay's per-candidate peaks and stream maxima,
        # because the hipMemsetAsync nodes that zero those atomicMax targets were not ordered in front of their kernels on
        # replay.  The library now zeroes with a kernel -- csrc/common.h zero_async -- and tests/test_gpu_es.py replays the graph
        # in 50 fresh processes against the eager result, bit for bit.)
        self._graph_on = (os.environ.get("STITO_GRAPH", "1") != "0") if use_graph is None else bool(use_graph)
        # a graph is captured on the (capture_after + 1)-th eligible call with the same population size and input buffer.  The
        # capture costs ~10 ms at pop 32 and ~30 ms at pop 256 (private-pool allocations + hipGraphInstantiate of ~45 kernel nodes)
        # and a replay saves 0.2 - 1 % of a step on an idle host, so a short run never earns it back (the reference's CLI default
        # stops after ~25 iterations, its PST harness runs 32: 7.5 -> 7.9 ms per iteration with a capture at call 9, measured):
        # the first 32 calls launch eagerly, longer runs switch to replay.  STITO_GRAPH_AFTER overrides; 0 = capture on the first
        # call (bench.py: inside its warm-up)
        self.capture_after = int(os.environ.get("STITO_GRAPH_AFTER", "32")) if capture_after is None else int(capture_after)
        self._graph_calls = {}
        self._graph_evictions = 0
        self._graphs = {}      # (P, input pointer, input shape) -> (graph, W buffer, loss, mid, side, n_calls, buffers kept alive)
        self._x_padded = None

    def _input(self, random_crop: bool, rng, parallel: bool = False) -> torch.Tensor:
        """Length policy of style_transfer.py:505-518 (one crop position for all inputs of a batch).  The reference's
        parallel=True branch (499-502) hands x to the pool as it is: no padding to 262144, no crop."""
        x = self.x_full
        n = x.shape[-1]
        if parallel:
            return x
        if n > CROP_LEN:
            if random_crop:  # 506-514: start 0 unless more than 16384 samples are spare (the crop still happens)
                start = int(rng.randint(16384, n - CROP_LEN)) if (n - CROP_LEN) > 16384 else 0
                return x[..., start:start + CROP_LEN].contiguous()
            return x
        if n == CROP_LEN:
            return x
        if self._x_padded is None:  # padded once: the same buffer for every call (a captured graph reads it)
            self._x_padded = torch.nn.functional.pad(x, (0, CROP_LEN - n)).contiguous()
        return self._x_padded

    def _fused_pass(self, Wc, x, p0, p1, per, n_calls, dropout, want_audio):
        """One pass of the fused AFx-Rep path over candidates p0 .. p1 - 1 on the current stream:
        render -> log-mel + Cnn14 -> loss.  -> (loss, mid, side, audio or None, peaks, n_calls)"""
        L = _hip.lib()
        B = self.n_inputs
        b0, b1 = p0 // per, (p1 + per - 1) // per
        xin = x[0] if B == 1 else x[b0:b1]
        audio, peaks = render_population(self.plugins, xin, Wc, self.sample_rate, chain=self.chain)
        mid, side = self.model.embed_raw(audio, peaks, norm_passes=2)
        loss = torch.empty(mid.shape[0], dtype=torch.float32, device=self.device)
        # dropout (style_transfer.py:549-551) hits the embeddings only inside the distance; the cosine is
        # scale-invariant, so dropping the raw vectors and normalising afterwards is the same quantity.
        # The returned embeddings stay undropped, like the reference's output_embeds.
        md, sd = mid, side
        if dropout > 0.0:
            md = torch.nn.functional.dropout(mid, p=dropout, training=True).contiguous()
            sd = torch.nn.functional.dropout(side, p=dropout, training=True).contiguous()
        spans = [(0, 0, p1 - p0)] if B == 1 else [(b, (b - b0) * per, (b - b0 + 1) * per) for b in range(b0, b1)]
        for b, q0, q1 in spans:  # candidates of pair b against target b
            _hip.check(L.stito_embed_loss(_hip.ptr(md[q0:q1]), _hip.ptr(sd[q0:q1]), q1 - q0, mid.shape[1],
                                          _hip.ptr(self.tmid[b]), _hip.ptr(self.tside[b]), _hip.ptr(loss[q0:q1]),
                                          _hip.ptr(self.flags[n_calls % 256]), _hip.stream_ptr()))
            n_calls += 1
        if dropout > 0.0:  # NaN scrub + L2 norm of the embeddings handed back
            _hip.check(L.stito_embed_loss(_hip.ptr(mid), _hip.ptr(side), mid.shape[0], mid.shape[1], None, None, None,
                                          _hip.ptr(self.flags[255]), _hip.stream_ptr()))
        return loss, mid, side, (normalize_audio_(audio, peaks) if want_audio else None), (audio, peaks), n_calls

    def _evaluate_graph(self, Wn: np.ndarray, x, per):
        """The plain fused call as one hipGraph launch, or None while this (population size, input buffer) has been seen fewer than
        capture_after times.  W travels through a static device buffer; the outputs are copied out of the graph's buffers, so
        they stay valid across calls."""
        P = Wn.shape[0]
        key = (P, x.data_ptr(), tuple(x.shape))
        ent = self._graphs.get(key)
        if ent is None:
            seen = self._graph_calls.get(key, 0)
            self._graph_calls[key] = seen + 1
            if seen < self.capture_after:
                return None   # not yet: the caller launches eagerly
            if len(self._graphs) >= 4 and self._graph_evictions >= 8:
                return None   # more than four shapes keep rotating: every call would re-capture (10 - 30 ms each); stay eager
            Wbuf = torch.empty((P, self.ndims), dtype=torch.float64, device=self.device)
            Wbuf.copy_(torch.from_numpy(Wn))
            if seen == 0:
                # nothing of this shape has run yet: one eager pass first (not part of the graph), which builds everything lazy
                # -- packed weights, workspaces, LDS attributes -- outside the capture
                side_stream = torch.cuda.Stream(self.device)
                side_stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side_stream):
                    self._fused_pass(Wbuf, x, 0, P, per, 0, 0.0, False)
                torch.cuda.current_stream(self.device).wait_stream(side_stream)
            # capture by hand on a side stream: the torch.cuda.graph() context manager also runs gc.collect() and
            # torch.cuda.empty_cache(), i.e. it hands every cached block back to the driver (the CLI-default point: 8.2 ms per
            # iteration with it, 7.9 without, 7.5 - 7.6 eager)
            g = torch.cuda.CUDAGraph()
            cap_stream = torch.cuda.Stream(self.device)
            cap_stream.wait_stream(torch.cuda.current_stream(self.device))
            try:
                with torch.cuda.stream(cap_stream):
                    g.capture_begin(capture_error_mode="relaxed")
                    try:
                        loss, mid, side, _, keep, n_calls = self._fused_pass(Wbuf, x, 0, P, per, 0, 0.0, False)
                    finally:
                        g.capture_end()
            except Exception as e:  # noqa: BLE001 -- a capture-unsafe call on another ROCm build, an allocation failure, ...
                # the step must not die at call capture_after + 1 of a long run: drop the partial graph, switch replay off for
                # this evaluator and let the caller launch eagerly (the eager path is the one the first capture_after calls took)
                import warnings
                warnings.warn(f"st_ito: hipGraph capture of the evaluate step failed ({type(e).__name__}: {e}); continuing with eager launches")
                self._graph_on = False
                self._graphs.clear()
                del g
                torch.cuda.synchronize(self.device)
                return None
            torch.cuda.current_stream(self.device).wait_stream(cap_stream)
            # the graph holds raw pointers: everything it touches stays referenced here -- the outputs, the render and trunk
            # workspaces as they were at capture (a later, larger call replaces those objects; the graph keeps its own)
            keep = (keep, _WS._bufs.get("render"), getattr(self.model, "_ws", None), x)
            ent = (g, Wbuf, loss, mid, side, n_calls, keep)
            if len(self._graphs) >= 4:   # a few shapes at most (find_w0 batch, population, last partial shard)
                old = next(iter(self._graphs))
                self._graphs.pop(old)
                self._graph_calls.pop(old, None)   # an evicted shape starts counting again instead of re-capturing on its next call
                self._graph_evictions += 1
            self._graphs[key] = ent
        g, Wbuf, loss, mid, side, n_calls, _ = ent
        Wbuf.copy_(torch.from_numpy(Wn))
        g.replay()
        self._n_flag_rows = min(n_calls, 255)
        return loss.clone(), {"mid": mid.clone(), "side": side.clone()}, None

    def evaluate(self, W, random_crop: bool = False, rng=np.random, want_audio: bool = False, dropout: float = 0.0,
                 parallel: bool = False):
        """Fitness of every row of W -> (loss (P,), embeddings dict, normalised audio or None).  The population goes through in
        passes of at most `max_candidates_per_pass` candidates (whole pairs for a multi-pair batch), one after the other on the
        current stream; a candidate's result does not depend on how the population is cut."""
        Wn = np.asarray(W, dtype=np.float64)
        if Wn.ndim != 2 or Wn.shape[1] != self.ndims:
            raise ValueError(f"parameter vectors must be (P, {self.ndims}), got {tuple(Wn.shape)}")
        x = self._input(random_crop, rng, parallel)
        P = Wn.shape[0]
        B = self.n_inputs
        if P == 0 or P % B:
            raise ValueError(f"{P} candidates cannot be split over {B} inputs")
        per = P // B  # candidates per input
        step = min(P, self.max_cand) if self.max_cand else P
        if B > 1:  # passes hold whole pairs
            step = max(per, step // per * per)
        bounds = [(p0, min(P, p0 + step)) for p0 in range(0, P, step)]
        cropped = random_crop and not parallel and self.x_full.shape[-1] > CROP_LEN   # a new input buffer per call
        if (self._graph_on and self.fused and len(bounds) == 1 and dropout == 0.0 and not want_audio and not cropped and
                not torch.cuda.is_current_stream_capturing()):
            out = self._evaluate_graph(Wn, x, per)
            if out is not None:
                return out
        Wt = torch.from_numpy(Wn).to(self.device)
        losses, mids, sides, audios, generic_embeds = [], [], [], [], []
        n_calls = 0
        for p0, p1 in bounds:
            Wc = Wt[p0:p1].contiguous()
            if self.fused:
                loss, mid, side, audio, _, n_calls = self._fused_pass(Wc, x, p0, p1, per, n_calls, dropout, want_audio)
                mids.append(mid); sides.append(side)
            else:
                b0, b1 = p0 // per, (p1 + per - 1) // per
                xin = x[0] if B == 1 else x[b0:b1]
                audio, peaks = render_population(self.plugins, xin, Wc, self.sample_rate, chain=self.chain)
                spans = [(0, 0, p1 - p0)] if B == 1 else [(b, (b - b0) * per, (b - b0 + 1) * per) for b in range(b0, b1)]
                loss, emb = self._generic_loss(normalize_audio_(audio, peaks), spans, dropout)
                generic_embeds.append(emb)
            losses.append(loss)
            if want_audio:
                audios.append(audio)
        loss = torch.cat(losses) if len(losses) > 1 else losses[0]
        if self.fused:
            embeds = {"mid": torch.cat(mids), "side": torch.cat(sides)}
        else:
            embeds = {k: torch.cat([e[k] for e in generic_embeds]) for k in generic_embeds[0]}
        self._n_flag_rows = min(n_calls, 255)
        return loss, embeds, (torch.cat(audios) if want_audio else None)

    def _generic_loss(self, audio: torch.Tensor, spans, dropout: float):
        """style_transfer.py:531-571 for an arbitrary metric: embed_func on the normalised population (GPU tensor),
        then mean over the dict's entries of -cosine_similarity to the target entry (stito_neg_cosine)."""
        L = _hip.lib()
        embeds = self.embed_func(audio, self.model, self.sample_rate)
        if not isinstance(embeds, dict) or not embeds:
            raise ValueError("embed_func must return a non-empty dict of (batch, embed_dim) tensors")
        n = audio.shape[0]
        loss = torch.empty(n, dtype=torch.float32, device=self.device)
        out = {}
        for idx, (name, emb) in enumerate(embeds.items()):
            if name not in self.targets:
                raise KeyError(f"target embeddings have no entry {name!r}")
            emb = emb.detach().to(self.device, torch.float32).contiguous().view(n, -1)
            out[name] = emb
            tgt = self.targets[name]
            if tgt.shape[1] != emb.shape[1]:
                raise ValueError(f"{name}: candidate embeddings have {emb.shape[1]} dims, the target {tgt.shape[1]}")
            # dropout hits the style embeddings only: the reference scores the content embeddings undropped
            # (style_transfer.py:545-568: F.dropout on input_embeds[embed_name], then the content term on its own)
            drop = dropout > 0.0 and not name.startswith("__content__:")
            ed = torch.nn.functional.dropout(emb, p=dropout, training=True).contiguous() if drop else emb
            for b, q0, q1 in spans:
                _hip.check(L.stito_neg_cosine(_hip.ptr(ed[q0:q1]), q1 - q0, emb.shape[1], _hip.ptr(tgt[b]),
                                              self.entry_weights.get(name, 1.0) / len(embeds),
                                              0 if idx == 0 else 1, _hip.ptr(loss[q0:q1]), _hip.stream_ptr()))
        return loss, out

    def nan_warning(self) -> Optional[str]:
        """The reference's "Warning: NaNs found in ..._embeddings" (utils.py:491-497) for the last evaluate();
        reads the device flags, so call it after the fitness has been fetched (run_es does)."""
        fl = self.flags[: getattr(self, "_n_flag_rows", 0)].sum(dim=0).cpu()
        if fl.numel() and int(fl[0]):
            return "Warning: NaNs found in mid_embeddings"
        if fl.numel() and int(fl[1]):
            return "Warning: NaNs found in side_embeddings"
        return None
