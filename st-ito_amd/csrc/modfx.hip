// modfx.hip -- the two effects of the reference's st_ito/effects.py / st_ito/dsp.py surface that are not on the ES chains of
// BASELINE.json: BasicChorus (effects.py:962-985 -> pedalboard.Chorus = juce::dsp::Chorus<float>) as a
// chain stage, and dasp_pytorch.functional.compressor as apply_random_compressor calls it (dsp.py:49-78).  Both libraries are
// un-vendored and un-pinned: restated from their published algorithms (oracle/dsp_oracle.c: oracle_chorus,
// oracle/st_ito_oracle.py: dasp_compressor) -- parity unpinned.
#include "dsp_view.h"

#include <vector>

// hipcc contracts a * b + c into an fma by default (its __fmul_rn / __fadd_rn are plain operators, not barriers, and
// -ffp-contract=fast ignores pragmas): the JUCE arithmetic restated here rounds every product, so this file is compiled with
// -ffp-contract=off (Makefile)

namespace stito {

// ---- chorus --------------------------------------------------------------------------------------------------------
// The LFO is common to every candidate and channel (it depends on the sample rate and the -- fixed -- rate only):
// juce::dsp::Oscillator accumulates its phase in float, sample by sample.  That recurrence is inherently serial (every addition
// rounds), so the phases are walked once on the HOST (stito_chorus_lfo: ~1 ms per 2^19 samples; a single GPU thread took ~4 ms of
// device time at 10 dependent cycles per step) and copied to the table; the sines are taken on the device in parallel.  The Python
// layer caches the table per sample rate and hands it to the stage as aux_dev.  (This file is compiled with -ffp-contract=off, host
// pass included: the products and sums below round one by one, exactly like the device walk they replace.)
static void chorus_phase_walk(float rate_hz, float sample_rate, int64_t n, float *s) {
    const float two_pi = 6.283185307179586f;
    const float inc = (two_pi * rate_hz) / sample_rate;
    float phase = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        s[i] = phase;
        float next = phase + inc;
        while (next >= two_pi) next = next - two_pi;
        phase = next;
    }
}
// sin(p - pi) rounded once from float64: the correctly rounded float sine, which is what a host libm's sinf returns (the delay
// is a float32 near 300 .. 1400 samples, ulp 3e-5 .. 1.2e-4: one ulp of the LFO would move the output by 1e-5 .. 1e-4 of its peak)
__global__ __launch_bounds__(256) void k_chorus_sin(int64_t n, float *__restrict__ s) {
    const float pi = 3.14159265358979323846f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s[i] = (float)sin((double)__fsub_rn(s[i], pi));
}

// One wave per (candidate, channel).  The feedback path runs through the delay line, whose delay is never below 1 ms, so
// blocks of TB <= floor(fs / 1000) consecutive samples only read values pushed by EARLIER blocks: lane t computes sample
// b0 + t; the one-sample recurrence v[n] = x[n] - feedback * wet[n - 1] needs the neighbouring lane's wet value only.
static constexpr int CH_RING = 4096;  // floats per wave: BasicChorus never delays by more than 30 ms (checked by the launcher)
__global__ __launch_bounds__(256) void k_chorus(InView in, float *__restrict__ out, int64_t cand_stride, int C, int64_t L,
                                                const double *__restrict__ coef, const float *__restrict__ lfo, double sample_rate,
                                                int n_streams, int tb) {
    __shared__ float ring_all[4][CH_RING];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int strm = blockIdx.x * 4 + wave;
    if (strm >= n_streams) return;
    float *ring = ring_all[wave];
    const int cand = strm / C, ch = strm % C;
    const double *cf = coef + (int64_t)cand * COEF_STRIDE;
    const float centre = (float)cf[0], osc_vol = (float)cf[1], fb = (float)cf[2], wet_v = (float)cf[3];
    const float dry_v = __fsub_rn(1.0f, wet_v);
    const float *x = in_ptr(in, cand, ch);
    float *y = out + (int64_t)cand * cand_stride + (int64_t)ch * L;
    for (int i = lane; i < CH_RING; i += 64) ring[i] = 0.0f;
    float wet_last = 0.0f;  // wet value of the sample before the block
    for (int64_t b0 = 0; b0 < L; b0 += tb) {
        const int64_t n = b0 + lane;
        const bool on = lane < tb && n < L;
        float xin = 0.0f, wet = 0.0f;
        if (on) {
            xin = x[n];
            float l = __fadd_rn(__fmul_rn(20.0f, __fmul_rn(lfo[n], osc_vol)), centre);
            l = l < 1.0f ? 1.0f : l;
            const float d = (float)((double)l * sample_rate / 1000.0);
            const int di = (int)floorf(d);
            const float frac = __fsub_rn(d, (float)di);
            const int64_t i1 = n - di, i2 = n - di - 1;
            const float v1 = i1 >= 0 ? ring[i1 & (CH_RING - 1)] : 0.0f;
            const float v2 = i2 >= 0 ? ring[i2 & (CH_RING - 1)] : 0.0f;
            wet = __fadd_rn(v1, __fmul_rn(frac, __fsub_rn(v2, v1)));
        }
        float wet_prev = __shfl_up(wet, 1, 64);
        if (lane == 0) wet_prev = wet_last;
        if (on) {
            ring[n & (CH_RING - 1)] = __fsub_rn(xin, __fmul_rn(wet_prev, fb));   // v[n] = x[n] - lastOutput
            y[n] = __fadd_rn(__fmul_rn(wet, wet_v), __fmul_rn(xin, dry_v));
        }
        wet_last = __shfl(wet, tb - 1, 64);
    }
}

int chorus_stage(const InView &in, float *audio_dev, int64_t cand_stride, int pop, int C, int64_t L, const double *coef,
                 const float *lfo_dev, int64_t lfo_len, double sample_rate, hipStream_t st) {
    STITO_REQUIRE(lfo_dev != nullptr && lfo_len >= L, STITO_E_INVALID, "Chorus: the LFO table (stito_chorus_lfo, aux_dev / aux_len) is missing or short");
    int tb = (int)floor(sample_rate / 1000.0);
    tb = tb > 64 ? 64 : tb;
    // longest delay: centre_delay_ms <= 20 (the parameter's range) + 20 * 0.5 * depth <= 10 ms
    STITO_REQUIRE(tb >= 8 && 31.0 * sample_rate / 1000.0 + 64 < CH_RING, STITO_E_UNSUPPORTED, "Chorus: sample rate %.0f", sample_rate);
    const int n_streams = pop * C;
    hipLaunchKernelGGL(k_chorus, dim3((unsigned)((n_streams + 3) / 4)), dim3(256), 0, st, in, audio_dev, cand_stride, C, L, coef, lfo_dev,
                       sample_rate, n_streams, tb);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

// ---- dasp_pytorch.functional.compressor ----------------------------------------------------------------------------------
// Gain computer per sample (float32, as the library's tensor ops), then the library's smoothing: one one-pole filter applied
// "by frequency sampling" on n_fft >= 2 n - 1 points.  The sampled response H = (1 - a) / (1 - a e^{-jw}) is the transform of
// the impulse response wrapped modulo n_fft, i.e. the causal recursion's response up to a^n_fft (< e^-200 for any audio
// length), so the filter IS  g[n] = (1 - a) g_c[n] + a g[n - 1]  to float32 FFT rounding; it runs here as that linear
// recurrence in float64, time-parallel: 256 chunks per item, local pass -> scan of the chunk ends -> corrected pass + gain.
__device__ __forceinline__ float dasp_gain_computer(float side, float thr, float ratio, float knee, float eps) {
    const float x_db = 20.0f * log10f(fmaxf(fabsf(side), eps));
    float x_sc = x_db;
    if (x_db >= thr - knee / 2 && x_db <= thr + knee / 2) {
        const float t = x_db - thr + knee / 2;
        x_sc = x_db + ((1.0f / ratio) - 1.0f) * (t * t) / (2.0f * knee);
    }
    if (x_db > thr + knee / 2) x_sc = thr + (x_db - thr) / ratio;
    return x_sc - x_db;
}

__global__ __launch_bounds__(256) void k_dasp_compressor(const float *__restrict__ x, int C, int64_t L, float sample_rate, float thr, float ratio,
                                                         float attack_ms, float knee, float makeup, float *__restrict__ out) {
    __shared__ double ends[256], starts[256];
    const int item = blockIdx.x, tid = threadIdx.x;
    const float *x0 = x + (int64_t)item * C * L;
    float *o0 = out + (int64_t)item * C * L;
    const float alpha_f = expf(-logf(9.0f) / (sample_rate * (attack_ms / 1e3f)));
    const double a = (double)alpha_f, b = (double)(1.0f - alpha_f);
    const int64_t B = (L + 255) / 256, s0 = (int64_t)tid * B;
    int64_t len = L - s0;
    len = len < 0 ? 0 : (len > B ? B : len);
    auto side = [&](int64_t i) {
        float s = x0[i];
        for (int c = 1; c < C; ++c) s += x0[(int64_t)c * L + i];
        return s;
    };
    double g = 0.0;
    for (int64_t i = 0; i < len; ++i) g = fma(a, g, b * (double)dasp_gain_computer(side(s0 + i), thr, ratio, knee, 1e-8f));
    ends[tid] = g;
    __syncthreads();
    if (tid == 0) {  // g entering chunk c: the chunk ends chained by a^B (256 steps)
        const double aB = pow(a, (double)B);
        double s = 0.0;
        for (int c = 0; c < 256; ++c) {
            starts[c] = s;
            s = fma(aB, s, ends[c]);   // (every chunk but the last has length B; the last one's end is never used)
        }
    }
    __syncthreads();
    g = starts[tid];
    for (int64_t i = 0; i < len; ++i) {
        g = fma(a, g, b * (double)dasp_gain_computer(side(s0 + i), thr, ratio, knee, 1e-8f));
        const float lin = powf(10.0f, ((float)g + makeup) / 20.0f);
        for (int c = 0; c < C; ++c) o0[(int64_t)c * L + s0 + i] = x0[(int64_t)c * L + s0 + i] * lin;
    }
}

}  // namespace stito

using namespace stito;

extern "C" int stito_chorus_lfo(double sample_rate, double rate_hz, int64_t n_samples, float *lfo_dev, void *stream) {
    STITO_REQUIRE(n_samples > 0 && lfo_dev != nullptr && sample_rate > 0 && rate_hz > 0, STITO_E_INVALID, "stito_chorus_lfo: empty table");
    {   // phases on the host, one blocking copy (the table is built once per sample rate and cached by the caller)
        std::vector<float> phases((size_t)n_samples);
        chorus_phase_walk((float)rate_hz, (float)sample_rate, n_samples, phases.data());
        STITO_HIP_CHECK(hipMemcpyAsync(lfo_dev, phases.data(), sizeof(float) * (size_t)n_samples, hipMemcpyHostToDevice, (hipStream_t)stream));
        STITO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));   // `phases` dies at the end of this block
    }
    hipLaunchKernelGGL(k_chorus_sin, dim3((unsigned)((n_samples + 4095) / 4096 > 1024 ? 1024 : (n_samples + 4095) / 4096)), dim3(256), 0,
                       (hipStream_t)stream, n_samples, lfo_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_dasp_compressor(const float *audio_dev, int n_items, int channels, int64_t n_samples, double sample_rate,
                                     double threshold_db, double ratio, double attack_ms, double knee_db, double makeup_gain_db,
                                     float *out_dev, void *stream) {
    STITO_REQUIRE(n_items > 0 && n_samples > 0 && channels >= 1, STITO_E_INVALID, "stito_dasp_compressor: empty input");
    STITO_REQUIRE(ratio > 0 && knee_db > 0 && attack_ms > 0, STITO_E_INVALID, "stito_dasp_compressor: ratio / knee / attack must be positive");
    hipLaunchKernelGGL(k_dasp_compressor, dim3(n_items), dim3(256), 0, (hipStream_t)stream, audio_dev, channels, n_samples, (float)sample_rate,
                       (float)threshold_db, (float)ratio, (float)attack_ms, (float)knee_db, (float)makeup_gain_db, out_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}
