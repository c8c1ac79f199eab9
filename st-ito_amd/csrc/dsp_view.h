// Shared between the effect kernels (dsp.hip, convreverb.hip): where a stage reads its input from.
#pragma once
#include "common.h"

namespace stito {

static constexpr int COEF_STRIDE = 32;  // doubles per (effect, candidate)

struct InView {  // where a stage reads its input from
    const float *base;
    int64_t cand_stride;  // 0: an input x shared by a group of candidates
    int64_t ch_stride;
    int in_ch;  // channel c reads channel c % in_ch (mono -> stereo up-mix, style_transfer.py:94-95)
    int group = 1 << 30;       // candidates per input (multi-pair batches: candidate p reads input p / group)
    int64_t group_stride = 0;  // floats between consecutive inputs
};

__device__ __forceinline__ const float *in_ptr(const InView &v, int cand, int ch) {
    return v.base + (int64_t)cand * v.cand_stride + (int64_t)(cand / v.group) * v.group_stride +
           (int64_t)(ch % v.in_ch) * v.ch_stride;
}

// Noise-shaped convolution reverb stage (convreverb.hip).  noise_bank: (2, 12, n_taps) float32.
size_t conv_reverb_workspace_bytes(int n_streams, int64_t n_samples, int64_t n_taps);
int conv_reverb_stage(const InView &in, float *audio_dev, int64_t cand_stride, int pop, int64_t n_samples,
                      const double *coef, const float *noise_bank, int64_t n_taps, void *workspace, hipStream_t st);

// Six-biquad float64 cascade of the parametric EQ (dsp.hip: k_eq) on (n_cand, C, L) audio; coef: n_cand rows of COEF_STRIDE doubles
// (b0 b1 b2 a1 a2 per section).
int eq_cascade(const InView &in, float *out, int n_cand, int C, int64_t L, const double *coef, hipStream_t st);

// Chorus stage (modfx.hip): lfo_dev = the table of stito_chorus_lfo (>= n_samples).
int chorus_stage(const InView &in, float *audio_dev, int64_t cand_stride, int pop, int C, int64_t L, const double *coef,
                 const float *lfo_dev, int64_t lfo_len, double sample_rate, hipStream_t st);

// Compressor stage (compressor.hip): envelope by block composition + VCA, in place on audio_dev.
size_t compressor_workspace_bytes(int n_streams, int64_t n_samples);
int compressor_stage(const InView &in, float *audio_dev, int64_t cand_stride, int pop, int C, int64_t n_samples,
                     const double *coef, void *workspace, hipStream_t st);

}  // namespace stito
