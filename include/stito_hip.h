/*
 * stito_hip.h -- C ABI of libstito_hip.so: the MI355X (gfx950) implementation of st-ito's
 * ES "evaluate-population" hot path.
 *
 * The reference (csteinmetz1/st-ito) is pure Python and has no FFI; the entry points below
 * are what a binding for this path replaces, one per stage of
 *     run_es.evaluate()                      st_ito/style_transfer.py:474-573
 * (file:line relative to the reference repository):
 *
 *   stito_render_population   <- the per-candidate loop over process_audio()
 *                                st_ito/style_transfer.py:45-115, 512-521 and the Basic*
 *                                plugin .process() methods st_ito/effects.py:784-959
 *   stito_normalize_audio     <- x /= clip(max|x|, 1e-8)   style_transfer.py:113
 *   stito_logmel              <- peak-normalise + mid/side + Spectrogram + LogmelFilterBank +
 *                                input_norm   st_ito/utils.py:473-474, models/panns.py:213-245
 *   stito_cnn14_*             <- Cnn14 conv stack / pooling head / fc_mid, fc_side
 *                                st_ito/models/panns.py:250-281
 *   stito_embed_loss          <- NaN scrub + F.normalize + -cosine_similarity + mean
 *                                st_ito/utils.py:491-501, style_transfer.py:544-571
 *
 * Conventions: every pointer named *_dev is a device pointer owned by the caller (PyTorch in
 * the Python host); `stream` is a hipStream_t passed as void*; all work is enqueued on that
 * stream -- or, for the streaming convolutions of stito_cnn14_forward / stito_conv3x3_bn_relu_ws
 * (ABI version 10), on library-owned side streams forked from and joined to it by events, so the
 * caller sees one stream's ordering and a hipGraph capture of `stream` captures them too -- and
 * nothing synchronises.  Functions return 0 on success and a negative STITO_E_*
 * code otherwise; stito_last_error() returns a thread-local message for the last failure.
 * No torch types appear here.  There is no CPU fallback: without a HIP device the calls fail.
 */
#ifndef STITO_HIP_H
#define STITO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STITO_OK 0
#define STITO_E_INVALID -1      /* bad argument (ValueError on the Python side) */
#define STITO_E_UNSUPPORTED -2  /* valid in the reference, not built here */
#define STITO_E_WORKSPACE -3    /* workspace too small */
#define STITO_E_HIP -4          /* a HIP runtime call failed */

/* Effect kinds; parameter order inside an effect is the reference's `parameters` dict order. */
enum {
    STITO_FX_PARAMETRIC_EQ = 0, /* effects.py:800-873, 18 params, 1-channel plugin */
    STITO_FX_COMPRESSOR = 1,    /* effects.py:876-897,  4 params, 1-channel */
    STITO_FX_DISTORTION = 2,    /* effects.py:900-916,  2 params, 1-channel */
    STITO_FX_DELAY = 3,         /* effects.py:919-934,  3 params, 2-channel in run_optim.py:395 */
    STITO_FX_REVERB = 4,        /* effects.py:937-959,  4 params, 2-channel */
    STITO_FX_GAIN = 5,          /* effects.py:532-542,  1 param  (BASELINE "gain" stage) */
    STITO_FX_NOISE_REVERB = 6,  /* effects.py:558-620 (apply_reverb -> dasp noise_shaped_reverberation), 25 params
                                   (12 band gains, 12 band decays, mix; raw values used as they are), 2-channel;
                                   convolution reverb of BASELINE.json configs[4].  Needs aux_dev / aux_len. */
    STITO_FX_CHORUS = 7,        /* effects.py:962-985 (pedalboard.Chorus = juce::dsp::Chorus<float>), 5 params (rate_hz is declared and,
                                   as in the reference's process(), not passed on: 1 Hz), 1-channel; aux_dev / aux_len: the LFO table
                                   of stito_chorus_lfo, at least n_samples long */
    STITO_FX_NUM_KINDS = 8
};

#define STITO_MAX_FX_PARAMS 32
/* stito_fx_desc.flags bit 0: joint peak normalisation of every candidate after this stage --
 * process_audio(normalize_stages=True), style_transfer.py:106-107. */
#define STITO_FX_FLAG_NORMALIZE_AFTER 1u

/* One plugin of the chain == one entry of the reference's `plugins` dict
 * (run_optim.py:376-437, style_transfer.py:17-42). */
typedef struct {
    int32_t kind;         /* STITO_FX_* */
    int32_t num_channels; /* plugin["num_channels"]: 1 = run per channel, 2 = stereo (up-mixes mono) */
    int32_t w_offset;     /* index in w of this plugin's first slot */
    int32_t has_bypass;   /* 1: slot 0 is the dead "our_bypass" dimension (style_transfer.py:28, 89-92) */
    uint32_t fixed_mask;  /* bit p set: parameter p comes from fixed_raw[p], but still consumes a w slot
                             (style_transfer.py:79-84) */
    uint32_t flags;       /* STITO_FX_FLAG_* (ABI version 2 renamed this field from `reserved`; same offset) */
    double fixed_raw[STITO_MAX_FX_PARAMS]; /* raw [0,1] value = (v - min) / (max - min) */
    const float *aux_dev; /* STITO_FX_NOISE_REVERB: band-filtered noise bank (2, 12, aux_len) float32 on the device
                             (the library draws it afresh per call; here it is an input); NULL otherwise */
    int64_t aux_len;      /* STITO_FX_NOISE_REVERB: impulse-response length in taps (dasp default 65 536) */
} stito_fx_desc;

const char *stito_last_error(void);
/* ABI version: 10 (1 = first round; 2: stito_fx_desc.flags was `reserved`; 3: stito_cnn14_weights.conv_wino_algo;
 * 4: STITO_CONV_WINOGRAD_F4_PRE, stito_conv3x3_bn_relu_ws / stito_conv3x3_workspace_bytes; stito_frontend.mel_w_stride
 * was `reserved`: 0 keeps the packed-run layout of versions 1-3; 5: STITO_CONV_WINOGRAD_F4_SPLIT, _F4_SPLIT2,
 * _F4_SPLITK, stito_conv_timing_read_each; 6: STITO_CONV_DIRECT_SPLIT;
 * 7: STITO_CONV_WINOGRAD_F2_REG; 8: stito_conv_block1_f2reg + stito_cnn14_weights.conv1_f2reg_w_dev (appended);
 * 9: algorithms 6 and 7 and stito_conv_block1_fused / stito_cnn14_pack_conv1_fused retired;
 * STITO_CONV_WINOGRAD_F4_SPLIT3; 10: stito_cnn14_weights.chunk_* (appended), stito_conv_timing_read_tagged). */
int stito_version(void);

/* LFO of STITO_FX_CHORUS: lfo_dev[n] = sin(phase_n - pi) with juce::dsp::Oscillator's float phase recurrence (phase += 2 pi
 * rate / fs per sample, wrapped), n < n_samples.  The recurrence is walked on the host and copied (this call BLOCKS on `stream`
 * and must not be made while the stream is being captured); the sines are taken on the device.  Cache the table per (sample
 * rate, rate). */
int stito_chorus_lfo(double sample_rate, double rate_hz, int64_t n_samples, float *lfo_dev, void *stream);
/* dasp_pytorch.functional.compressor as st_ito/dsp.py:49-78 (apply_random_compressor) calls it: side chain = channel sum,
 * soft-knee gain computer, ONE one-pole smoothing filter with the attack constant (the library evaluates it by frequency
 * sampling on >= 2 n - 1 points = the causal recursion to FFT rounding; the release constant is unused there), make-up gain.
 * audio_dev / out_dev (n_items, channels, n_samples) float32. */
int stito_dasp_compressor(const float *audio_dev, int n_items, int channels, int64_t n_samples, double sample_rate,
                          double threshold_db, double ratio, double attack_ms, double knee_db, double makeup_gain_db,
                          float *out_dev, void *stream);
/* Number of real parameters of an effect kind (without the bypass slot), or <0. */
int stito_fx_num_params(int kind);
/* Channel count of the rendered audio for `in_channels` input channels (style_transfer.py:94-104). */
int stito_chain_out_channels(const stito_fx_desc *chain, int n_fx, int in_channels);
/* Total number of w slots the chain consumes. */
int stito_chain_num_dims(const stito_fx_desc *chain, int n_fx);

size_t stito_render_workspace_bytes(const stito_fx_desc *chain, int n_fx, int in_channels,
                                    int64_t n_samples, int pop);

/* Render every candidate of the population through the chain.
 *   x_dev        (in_channels, n_samples) float32, shared by all candidates
 *   w_dev        (pop, n_dims) float64, raw parameters in [0,1] (CMA-ES phenotypes)
 *   audio_dev    (pop, out_channels, n_samples) float32: chain output BEFORE peak normalisation
 *   peaks_dev    (pop) float32: max |audio| over channels and samples of each candidate
 * The final `x /= clip(peak, 1e-8)` of process_audio() is applied by stito_normalize_audio /
 * stito_logmel so that the audio tensor only has to be rewritten when the caller wants it. */
int stito_render_population(const stito_fx_desc *chain, int n_fx, const float *x_dev, int in_channels,
                            int64_t n_samples, const double *w_dev, int pop, int n_dims,
                            double sample_rate, float *audio_dev, float *peaks_dev, void *workspace_dev,
                            size_t workspace_bytes, void *stream);

/* Multi-pair batch (BASELINE.json configs[2]; an extension: the reference's evaluate assumes one
 * input, style_transfer.py:520, and eval_pst.py:691-765 walks its examples serially).
 *   x_dev  (n_inputs, in_channels, n_samples); pop % n_inputs == 0; candidate p reads input
 *   p / (pop / n_inputs).  Everything else as stito_render_population (= the n_inputs == 1 case). */
int stito_render_population_multi(const stito_fx_desc *chain, int n_fx, const float *x_dev, int n_inputs,
                                  int in_channels, int64_t n_samples, const double *w_dev, int pop, int n_dims,
                                  double sample_rate, float *audio_dev, float *peaks_dev, void *workspace_dev,
                                  size_t workspace_bytes, void *stream);

/* max|x| per candidate over (channels, n_samples). */
int stito_peak(const float *audio_dev, int pop, int channels, int64_t n_samples, float *peaks_dev,
               void *stream);
/* audio[p] /= clip(peaks[p], 1e-8)  (style_transfer.py:113). */
int stito_normalize_audio(float *audio_dev, int pop, int channels, int64_t n_samples,
                          const float *peaks_dev, void *stream);

/* ---- resampling in front of the path -------------------------------------------------------- */
/* torchaudio.functional.resample(x, orig_freq, new_freq) with the library defaults (sinc_interp_hann,
 * lowpass_filter_width 6, rolloff 0.99) -- st_ito/utils.py:462-463, scripts/run_optim.py:446, 526.
 * orig / newf are the two rates divided by their gcd; width = ceil(6 * orig / (min(orig, newf) * 0.99));
 * kernel_t_dev (2 * width + orig, newf) float32: the library's kernel table transposed (host-built);
 * x_dev (rows, n_in) -> out_dev (rows, n_out), n_out = stito_resample_num_samples = ceil(newf * n_in / orig). */
int64_t stito_resample_num_samples(int64_t n_in, int orig, int newf);
int stito_resample_sinc(const float *x_dev, int rows, int64_t n_in, const float *kernel_t_dev, int orig, int newf,
                        int width, float *out_dev, int64_t n_out, void *stream);

/* ---- front end --------------------------------------------------------------------------- */
enum { STITO_NORM_NONE = 0, STITO_NORM_MINMAX = 1, STITO_NORM_BATCHNORM = 2 };

typedef struct {
    int32_t n_fft;        /* window_size (power of two, <= 4096) */
    int32_t hop;          /* hop_size */
    int32_t n_mels;       /* mel_bins (<= 256) */
    int32_t norm_mode;    /* STITO_NORM_* (Cnn14.input_norm, panns.py:233-245) */
    const float *window_dev;      /* (n_fft) analysis window (periodic Hann in the reference) */
    const float *twiddle_dev;     /* (n_fft/2, 2) cos/sin of exp(-2 pi i k / n_fft) */
    const int32_t *mel_start_dev; /* (n_mels) first FFT bin of each band */
    const int32_t *mel_len_dev;   /* (n_mels) number of consecutive bins */
    const int32_t *mel_off_dev;   /* (n_mels) offset of the band's first weight in mel_w_dev */
    const float *mel_w_dev;       /* the non-zero run of melW[:, m]: weight i of band m at mel_off[m] + i * mel_w_stride */
    const float *bn0_scale_dev;   /* (n_mels) gamma/sqrt(var+eps)        (BATCHNORM only) */
    const float *bn0_shift_dev;   /* (n_mels) beta - mean*scale          (BATCHNORM only) */
    int32_t no_center;    /* 0: frames centred with reflect padding (torchlibrosa, panns.py:147-155);
                             1: frame t starts at t*hop, no padding (torchaudio MelSpectrogram(center=False),
                             utils.py:104-113) */
    int32_t mel_w_stride; /* 0 / 1: runs packed back to back; n_mels with mel_off[m] = m: table (max run length, n_mels),
                             zero padded -- the lanes of a wave then read consecutive floats (ABI version 4; was `reserved`) */
} stito_frontend;

/* T = n_samples / hop + 1 frames (center=True, reflect padding). */
int64_t stito_num_frames(int64_t n_samples, int hop);
/* T = (n_samples - n_fft) / hop + 1 frames for no_center front ends. */
int64_t stito_num_frames_nocenter(int64_t n_samples, int n_fft, int hop);

/* audio_dev (pop, channels, n_samples) + peaks_dev (pop) = max|audio[p]|; norm_passes = how many
 * `x /= clip(max|x|, 1e-8)` the reference applies before the model: 2 for rendered candidates
 * (style_transfer.py:113 then utils.py:473-474), 1 for get_param_embeds() alone, 0 = none
 * -> logmel_dev (pop*channels, T, n_mels) float32, streams interleaved [p0_mid, p0_side, p1_mid, ...]
 * (panns.py:219-227).  channels == 1: one stream per candidate. */
int stito_logmel(const stito_frontend *fe, const float *audio_dev, const float *peaks_dev,
                 int norm_passes, int pop, int channels, int64_t n_samples, float *logmel_dev, void *stream);

/* ---- Cnn14 trunk -------------------------------------------------------------------------- */
#define STITO_CNN14_NUM_CONVS 12
/* 3x3 conv algorithm: direct implicit GEMM; Winograd F(2x2,3x3) (16 MACs per 2x2 outputs instead of 36);
 * Winograd F(4x4,3x3) (36 MACs per 4x4 outputs instead of 144: 4x fewer than direct, 1.78x fewer than F(2x2,3x3)).
 * Both Winograd forms need cin % 8 == 0, cout % 64 == 0 and a feature map whose halo patch fits LDS
 * (stito_conv3x3_supported). */
enum { STITO_CONV_DIRECT = 0, STITO_CONV_WINOGRAD = 1, STITO_CONV_WINOGRAD_F4 = 2,
       /* F(4x4,3x3) with the input transform hoisted out of the convolution: one pass writes the transformed input tiles
        * to a workspace, the convolution streams them like the weights.  Same packed weights as STITO_CONV_WINOGRAD_F4,
        * same arithmetic and results; pays off when cout >= 512 (the workgroups that share a pixel block and differ only
        * in their 64 output channels no longer repeat the transform).  Needs stito_conv3x3_bn_relu_ws (ABI version 4). */
       STITO_CONV_WINOGRAD_F4_PRE = 3,
       /* The hoisted form on the f16 matrix pipe with float32 accumulation: every float32 operand (transformed input,
        * transformed weight) travels as two f16 halves hi + lo of a power-of-two multiple of itself (22 significand bits;
        * the scale is per layer for the weights and per stream for the input, chosen from the data so that nothing
        * overflows), every product is hi hi' + hi lo' + lo hi'.  Results agree with the float32 forms to float32 rounding
        * level (not bitwise); own packing (stito_cnn14_packed_conv_floats), needs cin % 64 == 0, cout % 256 == 0 and
        * stito_conv3x3_bn_relu_ws (ABI version 5). */
       STITO_CONV_WINOGRAD_F4_SPLIT = 4,
       /* The same arithmetic per product on workgroup tiles twice as large (64 tiles x 64 channels), reached by going over the
        * input channels twice -- 18 of the 36 Winograd positions per sweep, the first sweep's share of the outputs parked in
        * the workspace: a third fewer bytes copied into LDS per MAC, which is what bounds _F4_SPLIT.  Own packing; sums in
        * a different order than _F4_SPLIT (same accuracy, not the same bits). */
       STITO_CONV_WINOGRAD_F4_SPLIT2 = 5,
       /* 6 (STITO_CONV_WINOGRAD_F4_SPLITK: split-precision products with the input transform inside the convolution) and
        * 7 (STITO_CONV_DIRECT_SPLIT: direct implicit GEMM with split operands) were experiments of rounds 3 - 4 that never beat
        * the kernels they were meant to replace; RETIRED in ABI version 9 (stito_conv3x3_supported returns 0 for them, packing
        * and launching fail with STITO_E_UNSUPPORTED).  The numbers stay reserved. */
       /* Winograd F(2x2,3x3) on the f16 matrix pipe with the same split operands, for the 64-input-channel layers (conv_block1.conv2,
        * conv_block2.conv1): the transformed WEIGHTS (16 positions x 64 x 64 as f16 hi + lo = 256 KB) stay in the registers of
        * persistent workgroups for the whole launch, the input transform is done in registers straight into the MFMA operand
        * layout (no transformed input in LDS or HBM), the raw halo patches arrive by LDS-DMA.  cin == 64, cout % 64 == 0; own
        * packing; workspace = one word per stream; stito_conv3x3_bn_relu_ws (ABI version 7).  gfx950 only by construction: a
        * workgroup needs 512 registers per wave and ~144 KB of dynamic LDS (153 KB with the first conv fused);
        * stito_conv3x3_supported / stito_conv_block1_f2reg_supported return 0 on a device whose LDS per workgroup is smaller,
        * and the trunk then takes the F(4x4,3x3) kernels for those layers. */
       STITO_CONV_WINOGRAD_F2_REG = 8,
       /* The split-precision streaming convolution on 128-tile x 128-channel workgroup tiles, in SIX sweeps over the input
        * channels (one Winograd position row per sweep; the rows' contributions to the 4 x 4 outputs accumulate in a workgroup-
        * private scratch area of the workspace): half the bytes copied into LDS per MAC of _F4_SPLIT2, which is what bounds these
        * kernels.  For the deep layers (long channel loops: the five extra passes over the outputs are + 13 % of the operand
        * stream at 2 048 input channels, + 27 % at 1 024).  cin % 64 == 0, cout % 512 == 0; own packing; the same arithmetic in the
        * same order as _F4_SPLIT2: identical results (ABI version 9). */
       STITO_CONV_WINOGRAD_F4_SPLIT3 = 9 };

typedef struct {
    int32_t embed_dim;
    int32_t n_mels;
    int32_t channels[7];   /* 1,64,128,256,512,1024,2048 */
    int32_t reserved;
    /* conv weights in the packed layout produced by stito_cnn14_pack_conv; BN folded to
     * per-channel scale/shift applied after the convolution (eval mode, panns.py:67-68) */
    const float *conv_w_dev[STITO_CNN14_NUM_CONVS];    /* STITO_CONV_DIRECT packing (required) */
    const float *conv_wino_dev[STITO_CNN14_NUM_CONVS]; /* Winograd packing of conv_wino_algo[i], or NULL: used per
                                                          layer whenever the feature map fits that kernel */
    int32_t conv_wino_algo[STITO_CNN14_NUM_CONVS];     /* STITO_CONV_WINOGRAD, _F4, _F4_PRE (these two share a packing), _F4_SPLIT, _F4_SPLIT2, _F4_SPLIT3 or STITO_CONV_WINOGRAD_F2_REG */
    const float *bn_scale_dev[STITO_CNN14_NUM_CONVS];
    const float *bn_shift_dev[STITO_CNN14_NUM_CONVS];
    const float *fc_mid_wt_dev;  /* (2048, embed_dim): fc_mid.weight transposed */
    const float *fc_mid_b_dev;   /* (embed_dim) */
    const float *fc_side_wt_dev; /* (2048, embed_dim) */
    const float *fc_side_b_dev;  /* (embed_dim) */
    const float *reserved_ptr;      /* was conv1_fused_w_dev (ABI 3 - 8: the F(4x4,3x3)-based fused block, retired in ABI 9); ignored */
    const float *conv1_f2reg_w_dev; /* stito_cnn14_pack_conv1_f2reg, or NULL (ABI v8).  With it and a
                                       STITO_CONV_WINOGRAD_F2_REG packing of conv index 1 the forward runs conv_block1 as ONE
                                       launch (stito_conv_block1_f2reg): the 64-channel full-resolution map is never stored */
    /* ABI v9: a second packing per conv (or NULL) for the calls in which the first one's kernel would not fill the device.
     * Today: conv_wino_algo[i] == STITO_CONV_WINOGRAD_F4_SPLIT3 with conv_alt_algo[i] == STITO_CONV_WINOGRAD_F4_SPLIT2 -- the
     * six-sweep kernel's workgroups are four times larger, so a small batch (fewer than 3/4 workgroup per CU) runs the
     * two-sweep kernel instead.  The two kernels do the same arithmetic in the same order: identical bits (tested), so the
     * choice cannot be seen in the results. */
    const float *conv_alt_dev[STITO_CNN14_NUM_CONVS];
    int32_t conv_alt_algo[STITO_CNN14_NUM_CONVS];
    /* ABI v10: depth-first schedule of a run of convs over chunks of streams (replaces the layer-by-layer order of
     * panns.py:250-261 for that run; same results bit for bit, a stream never meets another one inside a kernel).
     * chunk_streams > 0: convs chunk_first_conv .. chunk_last_conv (indices 0 .. 11, conv_block<b>.conv<j> = 2 (b - 1) + (j - 1))
     * run on chunk_streams streams at a time with the maps between them in two chunk-sized scratch buffers that every chunk
     * reuses, so that the hand-offs of a chunk can stay in the 256 MiB Infinity Cache.  chunk_first_conv == chunk_last_conv:
     * one layer launched chunk by chunk (only its transformed input is reused).  0 = layer by layer.  Runs the forward cannot
     * schedule (they would overwrite their own input) fall back to layer by layer. */
    int32_t chunk_streams;
    int32_t chunk_first_conv;
    int32_t chunk_last_conv;
    int32_t reserved2;
} stito_cnn14_weights;

/* Number of floats of a packed conv weight for (cout, cin) and algorithm. */
size_t stito_cnn14_packed_conv_floats(int cout, int cin, int algo);
/* (cout, cin, 3, 3) PyTorch layout -> packed layout used by the MFMA kernel of `algo`
 * (Winograd: the weights are transformed, U = G g G^T, in float64, rounded once). */
int stito_cnn14_pack_conv(const float *w_oihw_dev, int cout, int cin, int algo, float *packed_dev, void *stream);
/* BN(eval) -> scale = gamma / sqrt(var + eps), shift = beta - mean * scale.  gamma_dev == NULL:
 * identity (use_batchnorm=False). */
int stito_bn_fold(const float *gamma_dev, const float *beta_dev, const float *mean_dev,
                  const float *var_dev, double eps, int n, float *scale_dev, float *shift_dev,
                  void *stream);
/* (rows, cols) -> (cols, rows) float32 transpose (fc weights). */
int stito_transpose(const float *in_dev, int rows, int cols, float *out_dev, void *stream);

size_t stito_cnn14_workspace_bytes(const stito_cnn14_weights *w, int n_streams, int64_t n_frames);

/* logmel_dev (n_streams, T, n_mels) -> mid_dev / side_dev (n_cand, embed_dim), the raw fc outputs
 * (before L2 normalisation).  channels == 2: streams are [mid, side] pairs; channels == 1:
 * side := mid (panns.py:271-274). */
int stito_cnn14_forward(const stito_cnn14_weights *w, const float *logmel_dev, int n_cand, int channels,
                        int64_t n_frames, float *mid_dev, float *side_dev, void *workspace_dev,
                        size_t workspace_bytes, void *stream);

/* Individual layers, exposed for parity tests and profiling.
 * Activations are channel-blocked: a map with C channels (C % 8 == 0) is (n, C/8, H, W, 8) float32;
 * the 1-channel input of the first conv is (n, H, W).
 * in (n, cin/8, H, W, 8) -> out (n, cout/8, H', W', 8), y = relu(conv3x3(x) * scale + shift),
 * pool != 0: 2x2 average pooling (floor). */
/* conv_block1 in one launch (panns.py:250, ConvBlock 65-80): y = pool?(relu(bn2(conv3x3(relu(bn1(conv3x3(x))))))) for a
 * 1-channel input x (n, H, W) -- the log-mel image -- on the register-resident F(2x2,3x3) kernel (STITO_CONV_WINOGRAD_F2_REG;
 * ABI v8; the F(4x4,3x3)-based stito_conv_block1_fused of ABI 3 - 8, 11.4 ms against this one's 5.4, was retired in ABI 9).
 * out_dev (n, cout/8, H', W', 8).  The first conv (c1 = 64 channels, bn1, ReLU) is evaluated on the f16 matrix pipe (4 x 4 window
 * x 32 channels x 32 pixels per product; operands split into f16 hi + lo like the second conv's, three products, f32
 * accumulate) straight into the LDS patch ring of the second conv, in the instruction slots where the unfused kernel issues
 * its patch copies.
 *   packed_w1_dev  stito_cnn14_pack_conv1_f2reg(w1 (c1,1,3,3), bn1 scale, bn1 shift): stito_cnn14_packed_conv1_f2reg_floats()
 *   packed_w2_dev  STITO_CONV_WINOGRAD_F2_REG packing of w2 (cout, c1, 3, 3);  scale2_dev / shift2_dev (cout)
 *   workspace      stito_conv_block1_f2reg_workspace_bytes (per-stream scales of the log-mel operand)
 *   amax_out_dev   NULL, or n zeroed words: per-stream maxima of the output (bit patterns) for a split-precision layer behind
 * |x| must stay below 2^29 (log-mel values do: dB); needs c1 == 64, cout % 64 == 0. */
size_t stito_cnn14_packed_conv1_f2reg_floats(void);
int stito_cnn14_pack_conv1_f2reg(const float *w_oihw_dev, const float *scale_dev, const float *shift_dev, int c1, float *packed_dev,
                                 void *stream);
int stito_conv_block1_f2reg_supported(int n, int H, int W, int c1, int cout, int pool);
size_t stito_conv_block1_f2reg_workspace_bytes(int n, int H, int W, int c1, int cout, int pool);
int stito_conv_block1_f2reg(const float *x_dev, const float *packed_w1_dev, const float *packed_w2_dev, const float *scale2_dev,
                            const float *shift2_dev, float *out_dev, int n, int H, int W, int c1, int cout, int pool,
                            void *workspace_dev, size_t workspace_bytes, void *stream, unsigned *amax_out_dev);
/* Profiling aid: with buf_dev != NULL the Winograd launches use an instrumented instantiation whose
 * workgroup 100 records s_memtime stamps of 32 chunks x 12 waves x 8 phases (int64) into buf_dev;
 * NULL (default) restores the plain kernel.  Thread-local.  See tools/wino_timeline.py. */
int stito_debug_wino_trace(long long *buf_dev);
/* Measurement aid (bench.py "roofline"): while enabled, stito_cnn14_forward brackets every f32-MFMA
 * conv launch (cin % 8 == 0: 11 of the 12 convs) with hipEventRecord on the launch stream.
 * stito_conv_timing_read waits for the recorded events, returns their summed elapsed time and the
 * number of launches, and clears the list.  The switch and the event list are thread-local (one host thread per
 * GPU, like stito_last_error): enable and read from the thread that calls stito_cnn14_forward. */
int stito_conv_timing_enable(int on);
int stito_conv_timing_read(double *total_ms, int *n_launches);
/* The same per launch, in launch order (bench.py attributes the launches to the matrix pipe they run on): fills
 * ms_each[0 .. min(cap, n) - 1], returns n in *n_launches and clears the list. */
int stito_conv_timing_read_each(double *ms_each, int cap, int *n_launches);
/* The same with the conv index (0 .. 11; 1 for conv_block1 in one launch) of every launch in conv_each (may be NULL): with a
 * chunked schedule a layer is several launches per pass (ABI v10). */
int stito_conv_timing_read_tagged(double *ms_each, int *conv_each, int cap, int *n_launches);
/* FLOPs of the MFMA instructions one launch of this shape issues with `algo`, tile padding included (0 if unsupported or
 * cin % 8 != 0; STITO_CONV_WINOGRAD_F4_SPLIT: the three f16 products per element, i.e. f16-pipe FLOPs).  Measurement aid: bench.py divides it by the launch time for `roofline.achieved`. */
double stito_conv3x3_issued_flops(int n, int H, int W, int cin, int cout, int pool, int algo);
/* 1 if stito_conv3x3_bn_relu can run this shape with `algo`, else 0. */
int stito_conv3x3_supported(int n, int H, int W, int cin, int cout, int pool, int algo);
int stito_conv3x3_bn_relu(const float *in_dev, const float *packed_w_dev, const float *scale_dev,
                          const float *shift_dev, float *out_dev, int n, int H, int W, int cin, int cout,
                          int pool, int algo, void *stream);
/* The same with a workspace, for the algorithms that need one (STITO_CONV_WINOGRAD_F4_PRE / _F4_SPLIT / _F4_SPLIT2 / _F4_SPLIT3: the transformed
 * input; STITO_CONV_WINOGRAD_F2_REG: per-stream maxima of the input; stito_conv3x3_workspace_bytes; 0 bytes / NULL for the others). */
size_t stito_conv3x3_workspace_bytes(int n, int H, int W, int cin, int cout, int pool, int algo);
int stito_conv3x3_bn_relu_ws(const float *in_dev, const float *packed_w_dev, const float *scale_dev,
                             const float *shift_dev, float *out_dev, int n, int H, int W, int cin, int cout,
                             int pool, int algo, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- hand-crafted features: st_ito/features.py (alternative metrics of the evaluation harness) ---- */
/* compute_rms_energy (features.py:235-245) and compute_crest_factor (248-264) of (n_items, channels, n) audio
 * -> rms_dev, crest_dev (n_items, channels). */
int stito_rms_crest(const float *audio_dev, int n_items, int channels, int64_t n_samples, float *rms_dev,
                    float *crest_dev, void *stream);
/* compute_lufs (features.py:267-299): the reference's per-sample cross-channel normalisation, mono duplicated, then
 * pyloudnorm.Meter(sr).integrated_loudness (ITU-R BS.1770-4; un-vendored: restated, parity unpinned): K-weighting in float64
 * (kweight_coef_dev: n_items rows of 32 doubles, six sections b0 b1 b2 a1 a2 -- the two K-weighting biquads, then identity
 * rows 1 0 0 0 0), 400 ms blocks at 75 % overlap with the HOST's integer block edges block_lo/hi_dev (n_blocks; they are
 * pyloudnorm's int() of float products), inv_block_len = 1 / (0.4 sr), absolute gate -70 LUFS, relative gate -10 LU.
 * lufs_dev (n_items) float32, -inf for silence. */
size_t stito_lufs_workspace_bytes(int n_items, int64_t n_samples, int n_blocks);
int stito_lufs(const float *audio_dev, int n_items, int channels, int64_t n_samples, const double *kweight_coef_dev,
               const int *block_lo_dev, const int *block_hi_dev, int n_blocks, double inv_block_len, float *lufs_dev,
               void *workspace_dev, size_t workspace_bytes, void *stream);
/* compute_barkspectrum (features.py:166-232).  mode 0 mono / 1 stereo / 2 mid-side; fft_size a power of two
 * <= 32768 (hop fft_size/4, rectangular window, centred, reflect pad); twiddle_dev: fft_size/2 complex
 * exp(-2 pi i k / fft_size); fb_dev (n_bands, fft_size/2 + 1): barkscale_fbanks transposed;
 * out_dev (n_items, n_signals * n_bands), L2-normalised rows. */
int stito_barkspectrum(const float *audio_dev, int n_items, int channels, int64_t n_samples, int mode, int fft_size,
                       const float *twiddle_dev, const float *fb_dev, int n_bands, float *out_dev, void *stream);
/* compute_spectral_centroid (features.py:302-333): Hann 2048 / hop 1024 magnitude STFT, centroid per frame,
 * nan_to_num, adaptive average pooling to 10, / Nyquist -> out_dev (n_items, channels * 10).
 * window_dev: 2048 floats; twiddle_dev: 1024 complex exp(-2 pi i k / 2048). */
size_t stito_spectral_centroid_workspace_bytes(int n_items, int channels, int64_t n_samples);
int stito_spectral_centroid(const float *audio_dev, int n_items, int channels, int64_t n_samples, double sample_rate,
                            const float *window_dev, const float *twiddle_dev, float *out_dev, void *workspace_dev,
                            size_t workspace_bytes, void *stream);

/* MFCC statistics of get_mfcc_feature_embeds (utils.py:116-159) on top of stito_logmel's 10 log10(mel power):
 * logmel_dev (n_items * channels, T, n_mels); per item: clamp to (max over the item's channels, bands and
 * frames) - top_db (torchaudio amplitude_to_DB); per frame: DCT with dct_dev (n_mels, n_mfcc); over frames:
 * mean, unbiased std, max per coefficient -> out_dev (n_items, channels * 3 * n_mfcc) = per channel
 * [mean | std | max], rows L2-normalised.  n_mfcc <= 32, T * n_mfcc floats must fit 128 KB of LDS. */
int stito_mfcc_stats(const float *logmel_dev, int n_items, int channels, int64_t n_frames, int n_mels,
                     const float *dct_dev, int n_mfcc, float top_db, float *out_dev, void *stream);

/* ---- embeddings -> fitness ----------------------------------------------------------------- */
/* In place: NaN scrub (utils.py:491-497), L2-normalise mid/side (n_cand, E).  If target_mid_dev
 * is not NULL also writes loss_dev (n_cand) = mean(-cos(mid, target_mid), -cos(side, target_side))
 * (style_transfer.py:544-571); targets are (1, E). */
int stito_embed_loss(float *mid_dev, float *side_dev, int n_cand, int embed_dim,
                     const float *target_mid_dev, const float *target_side_dev, float *loss_dev,
                     int32_t *flags_dev /* (2) scratch */, void *stream);

/* Generic-metric form of the loss (any embed_func returning a dict of (n_cand, E_k) embeddings,
 * style_transfer.py:544-571): loss[c] (+)= weight * -cosine_similarity(embed[c], target), eps 1e-8.
 * The host walks the dict: first entry with accumulate = 0, the others with 1, weight = 1 / n_entries
 * (the reference's mean over the stacked distances).  embed (n_cand, embed_dim), target (embed_dim). */
int stito_neg_cosine(const float *embed_dev, int n_cand, int embed_dim, const float *target_dev, float weight,
                     int accumulate, float *loss_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STITO_HIP_H */
