/*
 * oracle/dsp_oracle.c -- CPU restatement of the reference's audio-effect arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under st-ito_amd/ may include, link or call
 * this file; it is the checker for the HIP kernels (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).
 *
 * Every function is a scalar, sample-serial restatement with the reference's
 * own operation order, compiled with -ffp-contract=off so that no FMA fusion
 * changes the rounding sequence.
 *
 * What is pinned and what is not:
 *   - rbj_biquad / eq_cascade follow st_ito/effects.py:395-450 and 453-512
 *     (scipy.signal.lfilter = direct-form-II-transposed in float64) and ARE pinned
 *     against the imported reference (tests/golden/eq_*.npz).
 *   - compressor / distortion / gain / delay / freeverb restate the JUCE /
 *     pedalboard processors that st_ito/effects.py:876-959 instantiates.  Those
 *     packages are un-vendored, un-pinned third-party code (setup.py:38) that is
 *     absent from this image: PARITY UNPINNED for these (formulas from the
 *     published JUCE sources, restated from memory).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- RBJ biquad design: st_ito/effects.py:395-450 (function `biqaud`) ------------
 * kind: 0 = low_shelf, 1 = peaking, 2 = high_shelf.  ba[0..2] = b/a0, ba[3..5] = a/a0. */
void oracle_rbj_biquad(double gain_db, double cutoff_freq, double q_factor,
                       double sample_rate, int kind, double *ba)
{
    double A = pow(10.0, gain_db / 40.0);
    double w0 = 2.0 * M_PI * (cutoff_freq / sample_rate);
    double alpha = sin(w0) / (2.0 * q_factor);
    double cos_w0 = cos(w0);
    double sqrt_A = sqrt(A);
    double b0, b1, b2, a0, a1, a2;
    if (kind == 2) {
        b0 = A * ((A + 1) + (A - 1) * cos_w0 + 2 * sqrt_A * alpha);
        b1 = -2 * A * ((A - 1) + (A + 1) * cos_w0);
        b2 = A * ((A + 1) + (A - 1) * cos_w0 - 2 * sqrt_A * alpha);
        a0 = (A + 1) - (A - 1) * cos_w0 + 2 * sqrt_A * alpha;
        a1 = 2 * ((A - 1) - (A + 1) * cos_w0);
        a2 = (A + 1) - (A - 1) * cos_w0 - 2 * sqrt_A * alpha;
    } else if (kind == 0) {
        b0 = A * ((A + 1) - (A - 1) * cos_w0 + 2 * sqrt_A * alpha);
        b1 = 2 * A * ((A - 1) - (A + 1) * cos_w0);
        b2 = A * ((A + 1) - (A - 1) * cos_w0 - 2 * sqrt_A * alpha);
        a0 = (A + 1) + (A - 1) * cos_w0 + 2 * sqrt_A * alpha;
        a1 = -2 * ((A - 1) + (A + 1) * cos_w0);
        a2 = (A + 1) + (A - 1) * cos_w0 - 2 * sqrt_A * alpha;
    } else {
        b0 = 1 + alpha * A;
        b1 = -2 * cos_w0;
        b2 = 1 - alpha * A;
        a0 = 1 + alpha / A;
        a1 = -2 * cos_w0;
        a2 = 1 - alpha / A;
    }
    ba[0] = b0 / a0; ba[1] = b1 / a0; ba[2] = b2 / a0;
    ba[3] = a0 / a0; ba[4] = a1 / a0; ba[5] = a2 / a0;
}

/* ---- scipy.signal.lfilter for one second-order section, float64 -------------------
 * Direct form II transposed, zero initial state (effects.py:486,499,510 call lfilter
 * with no `zi`).  In place on a float64 buffer. */
static void lfilter_sos_f64(const double *ba, double *x, int64_t n)
{
    const double b0 = ba[0], b1 = ba[1], b2 = ba[2], a1 = ba[4], a2 = ba[5];
    double z0 = 0.0, z1 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double xi = x[i];
        double yi = z0 + b0 * xi;
        z0 = z1 + b1 * xi - a1 * yi;
        z1 = b2 * xi - a2 * yi;
        x[i] = yi;
    }
}

/* ---- parametric_eq: st_ito/effects.py:453-512 --------------------------------------
 * p[18] = denormalised (gain_db, cutoff_freq, q) for low-shelf, band0..3, high-shelf in
 * the BasicParametricEQ.parameters order (effects.py:822-841).  One channel, float32 in,
 * float64 cascade, one cast to float32 at the end (effects.py:512). */
void oracle_parametric_eq(const float *x, float *y, int64_t n, double sample_rate,
                          const double *p)
{
    double *buf = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) buf[i] = (double)x[i];
    for (int s = 0; s < 6; ++s) {
        double ba[6];
        int kind = (s == 0) ? 0 : (s == 5 ? 2 : 1);
        oracle_rbj_biquad(p[3 * s + 0], p[3 * s + 1], p[3 * s + 2], sample_rate, kind, ba);
        lfilter_sos_f64(ba, buf, n);
    }
    for (int64_t i = 0; i < n; ++i) y[i] = (float)buf[i];
    free(buf);
}

/* ---- sensitivity switches for two JUCE details that the default restatement leaves out ------
 * snap_block > 0: pedalboard feeds JUCE blocks of that many samples (8192) and juce::dsp::BallisticsFilter::snapToZero()
 *                 runs after each block: envelope state with |y| < 1e-8 becomes 0.
 * undenormalise:  JUCE_UNDENORMALISE(x) = { x += 0.1f; x -= 0.1f; } (JUCE_INTEL builds) on the comb filters' `last` and
 *                 `temp` and the all-pass `temp` of juce::Reverb -- a quantisation of those states to multiples of 2^-27.
 * tests/test_oracle_golden.py measures what they change; the product kernels follow the default (both off). */
static int g_snap_block = 0, g_undenormalise = 0;
void oracle_set_juce_quirks(int snap_block, int undenormalise) { g_snap_block = snap_block; g_undenormalise = undenormalise; }
static inline float undenorm(float x)
{
    if (!g_undenormalise) return x;
    volatile float t = x;
    t += 0.1f;
    t -= 0.1f;
    return t;
}

/* ---- juce::Decibels::decibelsToGain<float> ------------------------------------------ */
static float db_to_gain_f32(float db, float minus_inf_db)
{
    return db > minus_inf_db ? powf(10.0f, db * 0.05f) : 0.0f;
}

/* ---- BasicCompressor: effects.py:876-897 -> pedalboard.Compressor ->
 * juce::dsp::Compressor<float> (peak BallisticsFilter + VCA).  [parity unpinned]
 * One channel; state starts at 0 (fresh object per call, effects.py:892). */
void oracle_compressor(const float *x, float *y, int64_t n, double sample_rate,
                       double threshold_db, double ratio, double attack_ms, double release_ms)
{
    float thr = db_to_gain_f32((float)threshold_db, -200.0f);
    float thr_inv = 1.0f / thr;
    float ratio_inv = 1.0f / (float)ratio;
    float exp_factor = (float)(-2.0 * M_PI * 1000.0 / sample_rate);
    float at = (float)attack_ms, rl = (float)release_ms;
    float cte_at = at < 1.0e-3f ? 0.0f : expf(exp_factor / at);
    float cte_rl = rl < 1.0e-3f ? 0.0f : expf(exp_factor / rl);
    float yold = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        float in = x[i];
        float v = fabsf(in);
        float cte = (v > yold) ? cte_at : cte_rl;
        float env = v + cte * (yold - v);
        yold = env;
        float g = (env < thr) ? 1.0f : powf(env * thr_inv, ratio_inv - 1.0f);
        y[i] = g * in;
        if (g_snap_block > 0 && (i + 1) % g_snap_block == 0 && !(yold < -1.0e-8f || yold > 1.0e-8f)) yold = 0.0f;
    }
}

/* Envelope only (used by tests that check the HIP envelope scan separately). */
void oracle_compressor_envelope(const float *x, float *env_out, int64_t n, double sample_rate,
                                double attack_ms, double release_ms)
{
    float exp_factor = (float)(-2.0 * M_PI * 1000.0 / sample_rate);
    float at = (float)attack_ms, rl = (float)release_ms;
    float cte_at = at < 1.0e-3f ? 0.0f : expf(exp_factor / at);
    float cte_rl = rl < 1.0e-3f ? 0.0f : expf(exp_factor / rl);
    float yold = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        float v = fabsf(x[i]);
        float cte = (v > yold) ? cte_at : cte_rl;
        float env = v + cte * (yold - v);
        yold = env;
        env_out[i] = env;
    }
}

/* ---- BasicDistortion: effects.py:900-916 -> pedalboard.Distortion (dsp::Gain + tanh
 * WaveShaper) then pedalboard.Gain.  [parity unpinned] */
void oracle_distortion(const float *x, float *y, int64_t n, double drive_db, double output_gain_db)
{
    float g_in = db_to_gain_f32((float)drive_db, -100.0f);
    float g_out = db_to_gain_f32((float)output_gain_db, -100.0f);
    for (int64_t i = 0; i < n; ++i) {
        float t = tanhf(x[i] * g_in);
        y[i] = t * g_out;
    }
}

/* ---- gain stage (BASELINE "gain"): effects.py:532-542 -> dasp_pytorch.gain:
 * x * 10^(gain_db/20).  [parity unpinned] */
void oracle_gain(const float *x, float *y, int64_t n, double gain_db)
{
    float g = powf(10.0f, (float)gain_db / 20.0f);
    for (int64_t i = 0; i < n; ++i) y[i] = x[i] * g;
}

/* ---- BasicDelay: effects.py:919-934 -> pedalboard.Delay (juce DelayLine, no
 * interpolation).  [parity unpinned]  One channel. */
void oracle_delay(const float *x, float *y, int64_t n, double sample_rate,
                  double delay_seconds, double feedback, double mix)
{
    float ds = (float)delay_seconds, fb = (float)feedback, mx = (float)mix;
    if (ds == 0.0f) { if (y != x) memcpy(y, x, sizeof(float) * (size_t)n); return; }
    int64_t D = (int64_t)(int)((double)ds * sample_rate); /* (int)(delaySeconds * spec.sampleRate), sampleRate is double */
    float dry = 1.0f - mx, wet = mx;
    if (D <= 0) {
        /* pop-before-push with zero delay reads the slot about to be overwritten, which
         * still holds zeros for a buffer longer than the signal: delayOutput = 0. */
        for (int64_t i = 0; i < n; ++i) y[i] = x[i] * dry + wet * 0.0f;
        return;
    }
    float *pushed = (float *)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        float d = (i >= D) ? pushed[i - D] : 0.0f;
        float in = x[i];
        pushed[i] = in + fb * d;
        y[i] = in * dry + wet * d;
    }
    free(pushed);
}

/* ---- BasicReverb: effects.py:937-959 -> pedalboard.Reverb -> juce::Reverb (Freeverb).
 * [parity unpinned]  Stereo in place semantics; l/r in, yl/yr out. */
static const int k_comb_tunings[8] = {1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617};
static const int k_allpass_tunings[4] = {556, 441, 341, 225};

void oracle_freeverb_sizes(double sample_rate, int *comb_sizes /*[2][8]*/, int *ap_sizes /*[2][4]*/)
{
    int isr = (int)sample_rate;
    for (int j = 0; j < 8; ++j) {
        comb_sizes[j] = (isr * k_comb_tunings[j]) / 44100;
        comb_sizes[8 + j] = (isr * (k_comb_tunings[j] + 23)) / 44100;
    }
    for (int j = 0; j < 4; ++j) {
        ap_sizes[j] = (isr * k_allpass_tunings[j]) / 44100;
        ap_sizes[4 + j] = (isr * (k_allpass_tunings[j] + 23)) / 44100;
    }
}

void oracle_freeverb(const float *l, const float *r, float *yl, float *yr, int64_t n,
                     double sample_rate, double room_size, double damping_p,
                     double wet_level, double dry_level, double width)
{
    int csz[16], asz[8];
    oracle_freeverb_sizes(sample_rate, csz, asz);
    float *cbuf[16], *abuf[8];
    int cidx[16] = {0}, aidx[8] = {0};
    float clast[16] = {0};
    for (int j = 0; j < 16; ++j) cbuf[j] = (float *)calloc((size_t)csz[j], sizeof(float));
    for (int j = 0; j < 8; ++j) abuf[j] = (float *)calloc((size_t)asz[j], sizeof(float));

    const float wet = (float)wet_level * 3.0f;
    const float dry = (float)dry_level * 2.0f;
    const float wet1 = 0.5f * wet * (1.0f + (float)width);
    const float wet2 = 0.5f * wet * (1.0f - (float)width);
    const float gain = 0.015f;
    const float damp = (float)damping_p * 0.4f;
    const float feedbck = (float)room_size * 0.28f + 0.7f;

    for (int64_t i = 0; i < n; ++i) {
        const float inl = l[i], inr = r[i];
        const float input = (inl + inr) * gain;
        float out[2] = {0.0f, 0.0f};
        for (int j = 0; j < 8; ++j) {
            for (int c = 0; c < 2; ++c) {
                int k = c * 8 + j;
                float output = cbuf[k][cidx[k]];
                clast[k] = undenorm((output * (1.0f - damp)) + (clast[k] * damp));
                float temp = undenorm(input + (clast[k] * feedbck));
                cbuf[k][cidx[k]] = temp;
                cidx[k] = (cidx[k] + 1) % csz[k];
                out[c] += output;
            }
        }
        for (int j = 0; j < 4; ++j) {
            for (int c = 0; c < 2; ++c) {
                int k = c * 4 + j;
                float bv = abuf[k][aidx[k]];
                float temp = undenorm(out[c] + (bv * 0.5f));
                abuf[k][aidx[k]] = temp;
                aidx[k] = (aidx[k] + 1) % asz[k];
                out[c] = bv - out[c];
            }
        }
        yl[i] = out[0] * wet1 + out[1] * wet2 + inl * dry;
        yr[i] = out[1] * wet1 + out[0] * wet2 + inr * dry;
    }
    for (int j = 0; j < 16; ++j) free(cbuf[j]);
    for (int j = 0; j < 8; ++j) free(abuf[j]);
}

/* ---- joint peak normalisation: style_transfer.py:113 -------------------------------
 * x /= clip(max|x|, 1e-8) over all channels; returns the peak. */
float oracle_peak_normalize(float *x, int64_t n_total)
{
    float peak = 0.0f;
    for (int64_t i = 0; i < n_total; ++i) { float a = fabsf(x[i]); if (a > peak) peak = a; }
    float d = peak < 1e-8f ? 1e-8f : peak;
    for (int64_t i = 0; i < n_total; ++i) x[i] = x[i] / d;
    return peak;
}

/* ---- BasicChorus: effects.py:962-985 -> pedalboard.Chorus = juce::dsp::Chorus<float> (rate_hz is NOT passed on by the
 * reference: the library default 1.0 Hz).  [parity unpinned: restated from JUCE's published sources]  One channel.
 *   osc       juce::dsp::Oscillator<float> with std::sin, no lookup table: per sample  p = phase; phase += 2 pi rate / fs
 *             (float accumulation, wrapped by subtracting 2 pi), value = sin(p - pi)
 *   lfo       max(1, 20 * (value * depth * 0.5) + centre_delay_ms) ms   (maximumDelayModulation 20, oscVolumeMultiplier 0.5;
 *             setCentreDelay clamps to [1, 100] ms); delay in samples = (float)(lfo * fs / 1000), fs a double
 *   line      DelayLine<float, Linear>: push v[n] = x[n] - lastOutput, read v at n - delay with linear interpolation
 *             value1 + frac * (value2 - value1), value1 = v[n - floor(delay)], value2 = v[n - floor(delay) - 1]
 *   feedback  lastOutput = wet[n] * feedback
 *   mix       DryWetMixer, linear rule: y = wet * mix + x * (1 - mix)
 * The smoothed values (oscVolume, feedback, mix) are reset to their targets in prepare(): no ramps in an offline render. */
/* lfo_table: NULL (the oscillator below, this libm's sinf) or n precomputed values of sin(phase - pi) -- the delay is a float32
 * of 300 .. 1400 samples (ulp 3e-5 .. 1.2e-4), so two sine implementations that differ in the last bit (glibc's sinf is not
 * correctly rounded everywhere, nor is any other) move the output by 1e-5 .. 1e-4 of its peak; a test that wants to compare
 * everything else tightly hands both sides the same table. */
void oracle_chorus_lfo(const float *x, float *y, int64_t n, double sample_rate, double rate_hz, double centre_delay_ms,
                       double depth, double feedback, double mix, const float *lfo_table);
void oracle_chorus(const float *x, float *y, int64_t n, double sample_rate, double rate_hz, double centre_delay_ms,
                   double depth, double feedback, double mix)
{
    oracle_chorus_lfo(x, y, n, sample_rate, rate_hz, centre_delay_ms, depth, feedback, mix, (const float *)0);
}
void oracle_chorus_lfo(const float *x, float *y, int64_t n, double sample_rate, double rate_hz, double centre_delay_ms,
                       double depth, double feedback, double mix, const float *lfo_table)
{
    const float two_pi = 6.283185307179586f, pi = 3.14159265358979323846f;
    float centre = (float)centre_delay_ms;
    centre = centre < 1.0f ? 1.0f : (centre > 100.0f ? 100.0f : centre);
    const float osc_vol = (float)depth * 0.5f, fb = (float)feedback, wet_v = (float)mix, dry_v = 1.0f - (float)mix;
    const float inc = two_pi * (float)rate_hz / (float)sample_rate;
    float *v = (float *)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    float phase = 0.0f, last = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        const float p = phase;
        float next = p + inc;
        while (next >= two_pi) next -= two_pi;
        phase = next;
        const float osc = (lfo_table ? lfo_table[i] : sinf(p - pi)) * osc_vol;
        float lfo = 20.0f * osc + centre;
        lfo = lfo < 1.0f ? 1.0f : lfo;
        const float d = (float)((double)lfo * sample_rate / 1000.0);
        const int di = (int)floorf(d);
        const float frac = d - (float)di;
        v[i] = x[i] - last;
        const float v1 = (i - di >= 0) ? v[i - di] : 0.0f;
        const float v2 = (i - di - 1 >= 0) ? v[i - di - 1] : 0.0f;
        const float wet = v1 + frac * (v2 - v1);
        last = wet * fb;
        y[i] = wet * wet_v + x[i] * dry_v;
    }
    free(v);
}
