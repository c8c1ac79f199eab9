// dsp.hip -- population-parallel audio-effect chain for gfx950.
//
// Replaces the per-candidate Python loop over process_audio() (reference
// st_ito/style_transfer.py:45-115, 512-521) and the Basic* plugins' .process()
// (st_ito/effects.py:784-959).  Audio lives in HBM as (pop, channels, n) float32; every effect
// works in place on that buffer.  The time axis is serial in the reference (IIR / delay-line
// recurrences); each kernel below uses the time-parallel formulation that fits its recurrence:
//
//   parametric EQ  : 12th-order LTI cascade, float64 like scipy.lfilter (effects.py:465-512).
//                    One workgroup per stream, 256 time-chunks per stream; zero-state pass,
//                    12x12 state-transition power + chunk scan in LDS, corrected pass.
//   compressor     : the switching one-pole envelope composes over (max, +): 15-sample block functions in
//                    parallel, a short serial scan over block boundaries, envelope + VCA in parallel
//                    (compressor.hip).
//   Freeverb       : one workgroup per candidate, all delay lines resident in LDS (~112 KB);
//                    time advances in tiles shorter than the shortest delay line, so comb /
//                    all-pass updates inside a tile are independent; the comb damping one-pole
//                    is a wave-level affine scan on the DPP network; one barrier per tile with
//                    the combs of tile k and the all-passes of tile k-1 in flight.
//   delay          : feedback delay of D samples = D independent geometric recurrences.
//   distortion/gain: element-wise.
#include "common.h"
#include "dsp_view.h"

namespace stito {


// What a stage's store may absorb from the stages that follow it: a Gain effect (x * 10^(g/20), effects.py:
// 532-542) and the final per-candidate peak max|y| (style_transfer.py:113) -- both pointwise on the value being
// stored, so the fused result is bit-identical to running them as separate passes over HBM.
struct PostOp {
    const double *gain_coef = nullptr;  // coefficient rows of the fused Gain stage (COEF_STRIDE doubles per candidate)
    float *peaks = nullptr;             // (pop) zero-initialised; atomicMax on the uint view (order-independent)
};

// ------------------------------------------------------------------------------------------------
// parameter tables: (min, max) of every Parameter, reference order
//   EQ effects.py:822-841, compressor 885-888, distortion 903-904, delay 924-926,
//   reverb 946-949, gain 538.
// ------------------------------------------------------------------------------------------------
__constant__ double c_pmin[STITO_FX_NUM_KINDS][STITO_MAX_FX_PARAMS] = {
    {-24, 20, 0.1, -24, 20, 0.1, -24, 20, 0.1, -24, 20, 0.1, -24, 20, 0.1, -24, 200, 0.1},
    {-80, 1, 0.1, 10},
    {-48, -24},
    {0.01, 0.05, 0.0},
    {0, 0, 0, 0},
    {-48},
    {0},
    {0.1, 0.1, 0, 0, 0}};
__constant__ double c_pmax[STITO_FX_NUM_KINDS][STITO_MAX_FX_PARAMS] = {
    {24, 4000, 4, 24, 10000, 4, 24, 10000, 4, 24, 10000, 4, 24, 10000, 4, 24, 18000, 4},
    {0, 20, 100, 1000},
    {48, 24},
    {1.0, 1.0, 1.0},
    {1, 1, 1, 1},
    {48},
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
    {10, 20, 1, 1, 1}};
static const int h_nparams[STITO_FX_NUM_KINDS] = {18, 4, 2, 3, 4, 1, 25, 5};

struct ChainArgs {
    int n_fx;
    stito_fx_desc fx[16];
};

// juce::Decibels::decibelsToGain<float>
__device__ __forceinline__ float db_to_gain(float db, float minus_inf) {
    return db > minus_inf ? powf(10.0f, db * 0.05f) : 0.0f;
}

// RBJ biquad, effects.py:395-450.  kind: 0 low-shelf, 1 peaking, 2 high-shelf.
__device__ void rbj(double gain_db, double f, double q, double sr, int kind, double *o /*b0 b1 b2 a1 a2*/) {
    const double A = pow(10.0, gain_db / 40.0);
    const double w0 = 2.0 * M_PI * (f / sr);
    const double alpha = sin(w0) / (2.0 * q);
    const double cw = cos(w0);
    const double sA = sqrt(A);
    double b0, b1, b2, a0, a1, a2;
    if (kind == 2) {
        b0 = A * ((A + 1) + (A - 1) * cw + 2 * sA * alpha);
        b1 = -2 * A * ((A - 1) + (A + 1) * cw);
        b2 = A * ((A + 1) + (A - 1) * cw - 2 * sA * alpha);
        a0 = (A + 1) - (A - 1) * cw + 2 * sA * alpha;
        a1 = 2 * ((A - 1) - (A + 1) * cw);
        a2 = (A + 1) - (A - 1) * cw - 2 * sA * alpha;
    } else if (kind == 0) {
        b0 = A * ((A + 1) - (A - 1) * cw + 2 * sA * alpha);
        b1 = 2 * A * ((A - 1) - (A + 1) * cw);
        b2 = A * ((A + 1) - (A - 1) * cw - 2 * sA * alpha);
        a0 = (A + 1) + (A - 1) * cw + 2 * sA * alpha;
        a1 = -2 * ((A - 1) + (A + 1) * cw);
        a2 = (A + 1) + (A - 1) * cw - 2 * sA * alpha;
    } else {
        b0 = 1 + alpha * A;
        b1 = -2 * cw;
        b2 = 1 - alpha * A;
        a0 = 1 + alpha / A;
        a1 = -2 * cw;
        a2 = 1 - alpha / A;
    }
    o[0] = b0 / a0; o[1] = b1 / a0; o[2] = b2 / a0; o[3] = a1 / a0; o[4] = a2 / a0;
}

// One thread per (candidate, effect): raw [0,1] -> value (Parameter.get_value, effects.py:795-797)
// -> the constants the effect kernel needs.
__global__ void k_prepare(ChainArgs chain, const double *__restrict__ w, int P, int D, double sr,
                          double *__restrict__ coef) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * chain.n_fx) return;
    const int f = idx / P, cand = idx % P;
    const stito_fx_desc &fx = chain.fx[f];
    const int kind = fx.kind;
    double v[STITO_MAX_FX_PARAMS];
    const int np = kind == 0 ? 18 : kind == 1 ? 4 : kind == 2 ? 2 : kind == 3 ? 3 : kind == 4 ? 4 : kind == 5 ? 1 : kind == 6 ? 25 : 5;
    for (int p = 0; p < np; ++p) {
        const double raw = ((fx.fixed_mask >> p) & 1u) ? fx.fixed_raw[p]
                                                       : w[(int64_t)cand * D + fx.w_offset + fx.has_bypass + p];
        v[p] = __dadd_rn(__dmul_rn(raw, c_pmax[kind][p] - c_pmin[kind][p]), c_pmin[kind][p]);
    }
    double *o = coef + ((int64_t)f * P + cand) * COEF_STRIDE;
    if (kind == STITO_FX_PARAMETRIC_EQ) {
        for (int s = 0; s < 6; ++s) rbj(v[3 * s], v[3 * s + 1], v[3 * s + 2], sr, s == 0 ? 0 : (s == 5 ? 2 : 1), o + 5 * s);
    } else if (kind == STITO_FX_COMPRESSOR) {  // juce::dsp::Compressor<float>::update + BallisticsFilter
        const float thr = db_to_gain((float)v[0], -200.0f);
        const float expf_ = (float)(-2.0 * M_PI * 1000.0 / sr);
        const float at = (float)v[2], rl = (float)v[3];
        o[0] = thr;
        o[1] = 1.0f / thr;
        o[2] = 1.0f / (float)v[1] - 1.0f;
        // the one-pole coefficients sit next to 1 and the envelope only sees 1 - c: one ulp of c is up to 3e-5 of
        // (1 - c).  exp in double of the float32 argument, rounded once, is the correctly rounded expf the host libm
        // returns (ocml's expf is allowed 1 ulp)
        o[3] = at < 1.0e-3f ? 0.0f : (float)exp((double)(expf_ / at));
        o[4] = rl < 1.0e-3f ? 0.0f : (float)exp((double)(expf_ / rl));
    } else if (kind == STITO_FX_DISTORTION) {
        o[0] = db_to_gain((float)v[0], -100.0f);
        o[1] = db_to_gain((float)v[1], -100.0f);
    } else if (kind == STITO_FX_DELAY) {
        const float ds = (float)v[0];
        o[0] = ds == 0.0f ? 0.0 : (double)(int)((double)ds * sr);
        o[1] = (float)v[1];
        o[2] = (float)v[2];
    } else if (kind == STITO_FX_REVERB) {  // juce::Reverb::setParameters; wet=wet_dry, dry=1-wet_dry (effects.py:956-957)
        const float room = (float)v[0], dampp = (float)v[1], wetl = (float)v[2], dryl = (float)(1 - v[2]),
                    width = (float)v[3];
        const float wet = wetl * 3.0f;
        o[0] = dampp * 0.4f;
        o[1] = room * 0.28f + 0.7f;
        o[2] = 0.5f * wet * (1.0f + width);
        o[3] = 0.5f * wet * (1.0f - width);
        o[4] = dryl * 2.0f;
    } else if (kind == STITO_FX_CHORUS) {  // juce::dsp::Chorus setters (rate_hz = v[0] is not passed on by BasicChorus.process)
        const float centre = (float)v[1];
        o[0] = centre < 1.0f ? 1.0f : (centre > 100.0f ? 100.0f : centre);  // setCentreDelay: jlimit(1, 100)
        o[1] = (float)v[2] * 0.5f;                                            // oscVolume = depth * oscVolumeMultiplier
        o[2] = (float)v[3];
        o[3] = (float)v[4];
    } else if (kind == STITO_FX_NOISE_REVERB) {  // dasp noise_shaped_reverberation: gains, 10 decay + 1, mix (float32)
        for (int b = 0; b < 12; ++b) {
            o[b] = (float)v[b];
            o[12 + b] = (float)v[12 + b] * 10.0f + 1.0f;
        }
        o[24] = (float)v[24];
    } else {
        o[0] = powf(10.0f, (float)v[0] / 20.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// Parametric EQ
// ------------------------------------------------------------------------------------------------
#ifndef EQ_NC_BUILD
#define EQ_NC_BUILD 256
#endif
static constexpr int EQ_NC = EQ_NC_BUILD;  // time chunks per stream == threads per workgroup
static constexpr int EQ_TS = 32;   // samples per LDS tile row (= tile floats per thread)
static constexpr int EQ_NG = EQ_NC / 32;   // 32-lane groups of the workgroup: a group stages one 32-sample row of the tile at a time

struct EqSec { double b0, b1, b2, a1, a2; };

// scipy.signal.lfilter's direct-form-II-transposed step for the 6-section cascade
__device__ __forceinline__ double eq_step(double x, const EqSec (&c)[6], double (&z)[12]) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const double y = fma(c[s].b0, x, z[2 * s]);
        z[2 * s] = fma(-c[s].a1, y, fma(c[s].b1, x, z[2 * s + 1]));
        z[2 * s + 1] = fma(-c[s].a2, y, c[s].b2 * x);
        x = y;
    }
    return x;
}

__global__ __launch_bounds__(EQ_NC) void k_eq(InView in, PostOp post, float *__restrict__ out, int64_t out_cand_stride,
                                               int C, int64_t L, const double *__restrict__ coef) {
    // the chunk states are only alive between the two passes, while the tile is not: one buffer for both
    constexpr size_t TILE_B = sizeof(float) * EQ_NC * (EQ_TS + 1), ZST_B = sizeof(double) * 12 * EQ_NC;
    __shared__ __attribute__((aligned(16))) char tz[TILE_B > ZST_B ? TILE_B : ZST_B];
    float (*tile)[EQ_TS + 1] = (float (*)[EQ_TS + 1])tz;
    double (*zst)[EQ_NC] = (double (*)[EQ_NC])tz;
    __shared__ double mat[3][144];

    const int s = blockIdx.x;
    const int cand = s / C, ch = s % C;
    const float *__restrict__ x = in_ptr(in, cand, ch);
    float *__restrict__ y = out + (int64_t)cand * out_cand_stride + (int64_t)ch * L;
    const double *cf = coef + (int64_t)cand * COEF_STRIDE;
    EqSec sec[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        sec[k].b0 = cf[5 * k]; sec[k].b1 = cf[5 * k + 1]; sec[k].b2 = cf[5 * k + 2];
        sec[k].a1 = cf[5 * k + 3]; sec[k].a2 = cf[5 * k + 4];
    }
    const int tid = threadIdx.x;
    const float post_gain = post.gain_coef != nullptr ? (float)post.gain_coef[(int64_t)cand * COEF_STRIDE] : 1.0f;  // x * 1.0f == x
    float post_max = 0.0f;
    const int64_t B = (L + EQ_NC - 1) / EQ_NC;  // chunk length
    const int64_t start = (int64_t)tid * B;
    int64_t len = L - start;
    len = len < 0 ? 0 : (len > B ? B : len);
    const int lr = tid >> 5, lj = tid & 31;

    double z[12];
    // tile staging: every thread fetches EQ_NC/8 = 32 scattered 4-byte pieces per tile; all of them
    // are issued together into registers one tile ahead of use so HBM latency overlaps the cascade.
    float pre[EQ_TS];
    // Index arithmetic in 32 bits (the launchers refuse streams of 2^31 samples), one add and one min per load: the staging of a
    // tile, not the float64 recurrence, is where this kernel's time went (profiles/README.md; the same lesson as k_conv_first).
    // A tile is INTERIOR when all of its 256 x 32 samples exist (58 of the 59 tiles of a 10 s stream): no masks at all then.
    const unsigned vbase = (unsigned)lr * (unsigned)B + (unsigned)lj, stride8 = (unsigned)EQ_NG * (unsigned)B, last = (unsigned)L - 1u;
    auto interior = [&](int64_t t0) { return t0 + EQ_TS <= B && (int64_t)(EQ_NC - 1) * B + t0 + EQ_TS <= L; };
    auto fetch = [&](int64_t t0) {  // unconditional loads (clamped); masked in stage(), one tile later
        const unsigned o0 = vbase + (unsigned)t0;
#pragma unroll
        for (int it = 0; it < EQ_TS; ++it) pre[it] = x[min(o0 + (unsigned)it * stride8, last)];
    };
    auto stage = [&](int64_t t0) {
        if (interior(t0)) {
#pragma unroll
            for (int it = 0; it < EQ_TS; ++it) tile[it * EQ_NG + lr][lj] = pre[it];
            return;
        }
        const unsigned o0 = vbase + (unsigned)t0;
        const bool in_chunk = t0 + lj < B;
#pragma unroll
        for (int it = 0; it < EQ_TS; ++it) tile[it * EQ_NG + lr][lj] = (in_chunk && o0 + (unsigned)it * stride8 <= last) ? pre[it] : 0.0f;
    };
    // ---- pass A: zero-state response of every chunk, keep only the final state -------------
#pragma unroll
    for (int k = 0; k < 12; ++k) z[k] = 0.0;
    fetch(0);
    for (int64_t t0 = 0; t0 < B; t0 += EQ_TS) {
        __syncthreads();
        stage(t0);
        __syncthreads();
        if (t0 + EQ_TS < B) fetch(t0 + EQ_TS);
        int64_t n = len - t0;
        n = n < 0 ? 0 : (n > EQ_TS ? EQ_TS : n);
        for (int j = 0; j < (int)n; ++j) (void)eq_step((double)tile[tid][j], sec, z);
    }
    __syncthreads();   // the states go where the tile was: every thread has finished reading its last tile row
#pragma unroll
    for (int k = 0; k < 12; ++k) zst[k][tid] = z[k];

    // ---- one-step transition matrix A (x = 0) and Phi = A^B by repeated squaring ------------
    if (tid < 12) {
        double e[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) e[k] = (k == tid) ? 1.0 : 0.0;
        (void)eq_step(0.0, sec, e);
#pragma unroll
        for (int k = 0; k < 12; ++k) mat[1][k * 12 + tid] = e[k];  // base
    }
    if (tid < 144) mat[0][tid] = (tid / 12 == tid % 12) ? 1.0 : 0.0;  // result = I
    __syncthreads();
    {
        const int r = tid / 12, k = tid % 12;
        for (int64_t e = B; e > 0; e >>= 1) {
            if (e & 1) {  // result = base * result
                double acc = 0.0;
                if (tid < 144)
                    for (int m = 0; m < 12; ++m) acc = fma(mat[1][r * 12 + m], mat[0][m * 12 + k], acc);
                __syncthreads();
                if (tid < 144) mat[0][tid] = acc;
                __syncthreads();
            }
            double acc = 0.0;
            if (tid < 144)
                for (int m = 0; m < 12; ++m) acc = fma(mat[1][r * 12 + m], mat[1][m * 12 + k], acc);
            __syncthreads();
            if (tid < 144) mat[1][tid] = acc;
            __syncthreads();
        }
    }
    // ---- chunk scan: s_in[c+1] = Phi s_in[c] + z_c  (lanes 0..11 of wave 0, one row each) ------
    if (tid < 64) {
        const int i = tid < 12 ? tid : 0;
        double phi[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) phi[k] = mat[0][i * 12 + k];
        double si = 0.0;
        for (int c = 0; c < EQ_NC; ++c) {
            const double zc = zst[i][c];
            double acc = zc;
#pragma unroll
            for (int k = 0; k < 12; ++k) acc = fma(phi[k], __shfl(si, k), acc);
            if (tid < 12) zst[i][c] = si;  // state entering chunk c
            si = acc;
        }
    }
    __syncthreads();
    // ---- pass B: rerun every chunk from its true initial state, write the output -------------
#pragma unroll
    for (int k = 0; k < 12; ++k) z[k] = zst[k][tid];
    fetch(0);
    for (int64_t t0 = 0; t0 < B; t0 += EQ_TS) {
        __syncthreads();
        stage(t0);
        __syncthreads();
        if (t0 + EQ_TS < B) fetch(t0 + EQ_TS);
        int64_t n = len - t0;
        n = n < 0 ? 0 : (n > EQ_TS ? EQ_TS : n);
        for (int j = 0; j < (int)n; ++j) tile[tid][j] = (float)eq_step((double)tile[tid][j], sec, z);
        __syncthreads();
        {
            const unsigned o0 = vbase + (unsigned)t0;
            const bool all = interior(t0), in_chunk = t0 + lj < B;
#pragma unroll
            for (int it = 0; it < EQ_TS; ++it) {
                const unsigned o = o0 + (unsigned)it * stride8;
                if (all || (in_chunk && o <= last)) {
                    const float v = tile[it * EQ_NG + lr][lj] * post_gain;
                    y[o] = v;
                    post_max = fmaxf(post_max, fabsf(v));
                }
            }
        }
    }
    if (post.peaks != nullptr) {  // block max -> the candidate's peak
        __shared__ float pk_red[EQ_NC / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) post_max = fmaxf(post_max, __shfl_xor(post_max, o));
        if ((tid & 63) == 0) pk_red[tid >> 6] = post_max;
        __syncthreads();
        if (tid == 0) {
            float m = pk_red[0];
            for (int w = 1; w < EQ_NC / 64; ++w) m = fmaxf(m, pk_red[w]);
            atomicMax((unsigned int *)&post.peaks[cand], __float_as_uint(m));
        }
    }
}

// The same cascade for other callers (features.hip: the K-weighting of BS.1770 = two biquads, the other four sections identity
// rows b0 = 1): coef = n_cand rows of COEF_STRIDE doubles, out (n_cand, C, L) float32.
int eq_cascade(const InView &in, float *out, int n_cand, int C, int64_t L, const double *coef, hipStream_t st) {
    STITO_REQUIRE(L > 0 && L < ((int64_t)1 << 31) - 65536, STITO_E_UNSUPPORTED, "parametric EQ: %lld samples per stream (k_eq indexes in 32 bits)", (long long)L);
    hipLaunchKernelGGL(k_eq, dim3((unsigned)(n_cand * C)), dim3(EQ_NC), 0, st, in, PostOp{}, out, (int64_t)C * L, C, L, coef);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

// ------------------------------------------------------------------------------------------------
// Distortion (tanh waveshaper + output gain, effects.py:907-916) and gain (effects.py:532-542)
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void k_pointwise(InView in, float *__restrict__ out, int64_t cand_stride, int C,
                                                    int64_t L, const double *__restrict__ coef) {
    const int s = blockIdx.y;
    const int cand = s / C, ch = s % C;
    const float *x = in_ptr(in, cand, ch);
    float *y = out + (int64_t)cand * cand_stride + (int64_t)ch * L;
    const double *cf = coef + (int64_t)cand * COEF_STRIDE;
    const float g0 = (float)cf[0], g1 = (float)cf[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (KIND == STITO_FX_DISTORTION) y[i] = tanhf(v * g0) * g1;
        else if (KIND == STITO_FX_GAIN) y[i] = v * g0;
        else y[i] = v;  // copy / up-mix
    }
}

// ------------------------------------------------------------------------------------------------
// Delay: pedalboard.Delay (juce DelayLine, no interpolation), effects.py:929-934
//   d[n] = pushed[n-D]; pushed[n] = x[n] + fb*d[n]; y[n] = x[n]*(1-mix) + mix*d[n]
// Residue class i (mod D) is an independent recurrence over n = i, i+D, i+2D, ...
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_delay(InView in, float *__restrict__ out, int64_t cand_stride, int C,
                                                int64_t L, const double *__restrict__ coef) {
    const int cand = blockIdx.y;
    const double *cf = coef + (int64_t)cand * COEF_STRIDE;
    const int64_t D = (int64_t)cf[0];
    const float fb = (float)cf[1], mix = (float)cf[2];
    const float dry = 1.0f - mix;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const bool dup = (in.in_ch == 1 && C == 2);  // identical channels: compute once, write twice
    if (D <= 0) {  // delay_seconds == 0: pedalboard passes the input through
        for (int ch = 0; ch < C; ++ch) out[(int64_t)cand * cand_stride + (int64_t)ch * L + i] = in_ptr(in, cand, ch)[i];
        return;
    }
    if (i >= D) return;
    // One residue class = one serial chain of L / D steps, but only its two FMAs are serial: the sixteen samples of a batch are loaded
    // TOGETHER (a thread only ever re-reads positions of its own class, and it has read them before it writes them, so the in-place
    // form is safe), then walked, then stored.  One load per step behind the previous step's store was one memory round trip per step:
    // 254 us at a population of 32 with the delays of the CLI-default chain (round 6: 5 500 steps of ~500 cycles).  Channels run on
    // blockIdx.z (they were a loop in the thread).
    constexpr int DL_U = 16;
    const int ch0 = dup ? 0 : (int)blockIdx.z;
    if (ch0 >= C) return;
    {
        const int ch = ch0;
        const float *x = in_ptr(in, cand, ch);
        float *y = out + (int64_t)cand * cand_stride + (int64_t)ch * L;
        float prev = 0.0f;
        for (int64_t n = i; n < L; n += D * DL_U) {
            float v[DL_U];
#pragma unroll
            for (int u = 0; u < DL_U; ++u) {
                const int64_t m = n + u * D;
                v[u] = m < L ? x[m] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < DL_U; ++u) {
                const int64_t m = n + u * D;
                if (m < L) {
                    const float o = v[u] * dry + mix * prev;
                    prev = v[u] + fb * prev;
                    y[m] = o;
                    if (dup) y[L + m] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Freeverb: juce::Reverb::processStereo, effects.py:952-959
// ------------------------------------------------------------------------------------------------
static constexpr int RV_TT = 192;       // tile length; must be <= the shortest delay line (244 @ 48 kHz)
static constexpr int RV_RUN = 12;       // consecutive samples of a comb per lane: a comb's tile is ONE 16-lane DPP row, four combs per wave
static constexpr int RV_COMB_WAVES = 4;
static constexpr int RV_AP_WAVES = 2 * RV_TT / 64;   // one all-pass / mix thread per (channel, sample)
static constexpr int RV_STAGE_WAVES = 2;             // input staging: thread v < RV_TT / 2 moves samples 2v, 2v + 1 of both channels
static constexpr int RV_THREADS = (RV_COMB_WAVES + RV_AP_WAVES + RV_STAGE_WAVES) * 64;
static constexpr int RV_PD = 4;         // input tiles in flight per staging thread (register ring)
static constexpr int RV_PAD = 32;       // floats: half the LDS banks
static constexpr int RV_XB = 2 * RV_TT + RV_PAD;    // floats per dry-input buffer
static constexpr int RV_CB = 16 * RV_TT + RV_PAD;   // floats per comb-output buffer
static constexpr int RV_TILE_FLOATS = 2 * RV_TT + 3 * RV_XB + 2 * RV_CB;  // s_in + s_x + s_comb
static_assert(RV_RUN * 16 == RV_TT && RV_RUN % 4 == 0 && RV_THREADS <= 1024, "a comb's tile is 16 lanes x RV_RUN samples");
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct ReverbGeom {
    int comb_size[16];  // [ch*8 + j]
    int comb_off[16];   // float offsets into the LDS state area
    int ap_size[8];     // [ch*4 + j]
    int ap_off[8];
    int state_floats;
};

// DPP move: lane i reads `v` of the lane selected by CTRL; rows outside ROW_MASK and lanes whose source is out of
// range read 0
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float rv_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}

// One workgroup per candidate, every delay line in LDS, ONE barrier per 192-sample tile, three roles on separate waves
// working on three consecutive tiles at once.  The tile time is set by LATENCY, not by instruction count or bandwidth
// (measured with s_memtime stamps and by switching roles off: round 1's kernel spent 2 200 cycles per tile, of which per
// tile: two vector loads of the all-pass geometry from the kernel-argument struct + vmcnt(0), two 64-bit modulos, an
// all-pass chain that read each line only after the previous stage's write, a wait for the input on the heels of the
// output store -- vmcnt is one in-order counter -- and 3 samples per lane behind a 6-step wave scan).  Hence:
//   staging waves  only loads in their vmcnt: the input of tile k + 1 comes out of a register ring filled RV_PD tiles ahead
//                  (branch-free, so hipcc counts the waits instead of draining);
//   all-pass waves only stores: comb sums, all four all-pass reads and the dry sample are fetched together, then the chain
//                  runs on registers;
//   comb waves     a comb's tile is 16 lanes (one DPP row) x 12 consecutive samples: the damping one-pole runs serially inside
//                  the lane (11 + 12 dependent FMAs) and a 4-step scan of affine maps crosses the row's lanes on the DPP network.
// (An LDS-only barrier -- s_waitcnt lgkmcnt(0); s_barrier, without __syncthreads()'s vmcnt(0) -- was measured too: 2.16
// against 2.06 ms on the same box; hipcc then drains vmcnt in front of the LDS writes instead.)
#define RV_BARRIER() __syncthreads()
// wet / dry mix of juce::Reverb::processStereo with its two fused multiply-adds spelled out: the one-workgroup kernel and the
// per-channel form (k_reverb<true> + k_reverb_mix) must round alike -- a candidate's audio may not depend on which of them its
// batch size selects
__device__ __forceinline__ float rv_mix(float wet_own, float wet_other, float x, float wet1, float wet2, float dry) {
    return fmaf(wet_own, wet1, fmaf(wet_other, wet2, x * dry));
}
#ifndef RV_ABL
#define RV_ABL 0   // timing builds only (tools/reverb_ablate.sh): 1 = comb role idle, 2 = all-pass role idle
#endif

// HALF (round 6, small populations: a population of 32 puts 32 of these workgroups on 256 CUs): one workgroup per (candidate, CHANNEL)
// -- the channel's eight combs on two waves, its all-pass chain on three, the same staging (the comb input is the sum of both
// channels) -- writing the channel's WET signal to a scratch buffer; k_reverb_mix then forms wet_c wet1 + wet_other wet2 + x_c dry
// with the same two fused multiply-adds.  Same LDS geometry, same arithmetic per comb / all-pass: identical bits (tested), half the
// instructions per SIMD and tile.
template <bool HALF>
__global__ __launch_bounds__(RV_THREADS) void k_reverb(InView in, float *__restrict__ out, int64_t cand_stride,
                                                        int64_t L, const double *__restrict__ coef, ReverbGeom g) {
    constexpr int NCW = HALF ? RV_COMB_WAVES / 2 : RV_COMB_WAVES, NAW = HALF ? RV_AP_WAVES / 2 : RV_AP_WAVES;
    constexpr int NTHREADS = (NCW + NAW + RV_STAGE_WAVES) * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *state = smem;                          // comb + all-pass delay lines
    float *s_in = smem + g.state_floats;          // [2 buffers][RV_TT] (L+R)*gain
    // channel 1 sits 32 banks behind channel 0 in s_x and s_comb: an all-pass wave's lanes alternate channels on the same
    // sample, and without the pad both channels of a sample fall on one bank in every comb-sum read
    float *s_x = s_in + 2 * RV_TT;                // [3 buffers][2][RV_TT (+ RV_PAD for channel 1)] dry input
    float *s_comb = s_x + 3 * RV_XB;              // [2 buffers][16][RV_TT] (+ RV_PAD in front of combs 8..15) comb outputs

    const int cand = HALF ? blockIdx.x >> 1 : blockIdx.x, my_ch = HALF ? blockIdx.x & 1 : 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *cf = coef + (int64_t)cand * COEF_STRIDE;
    const float damp = (float)cf[0], fbk = (float)cf[1], wet1 = (float)cf[2], wet2 = (float)cf[3], dry = (float)cf[4];
    const float omd = 1.0f - damp;
    const float *xl = in_ptr(in, cand, 0), *xr = in_ptr(in, cand, 1);
    float *yl = out + (int64_t)cand * cand_stride, *yr = yl + L;   // (HALF: out = the wet scratch buffer, (pop, 2, L))

    for (int i = tid; i < g.state_floats; i += NTHREADS) state[i] = 0.0f;
    // tiles, rounded up to whole turns of the staging ring: every role runs ntiles + 1 barrier steps, and the staging loop has no
    // early exit inside its unrolled turn (with one, hipcc gives up counting its loads and drains vmcnt(0) every turn); the tiles
    // past the end of the signal are zeros in, nothing out
    const int ntiles = (int)((L + RV_TT - 1) / RV_TT + RV_PD - 1) / RV_PD * RV_PD + (RV_PD - 1);

    if (wv < NCW) {
        // ---- comb role: the four 16-lane rows of wave w are combs 4 w .. 4 w + 3 (HALF: of this workgroup's channel); lane l of a row owns samples 12 l .. 12 l + 11.
        // The kernel is bound by the instructions its waves issue between two barriers, summed per SIMD (tools/reverb_ablate.sh: the
        // roles' times ADD -- barriers only 0.4 ms, + all-pass 0.4, + comb 0.7 at 256 candidates -- at ~4.8 cycles per instruction and
        // SIMD): with two combs of 32 lanes x 6 samples per wave (rounds 2 - 4) the eight comb waves issued 8 x 92 instructions per
        // tile; a row per comb halves the waves for ~100 each, and the scan needs no step across rows.
        const int cidx = (HALF ? 8 * my_ch : 0) + 4 * wv + (lane >> 4), cl = lane & 15;
        const int csz = g.comb_size[cidx];
        float *cbuf = state + g.comb_off[cidx];
        int cpos = 0;
        float last_in = 0.0f;  // filterStore entering the tile (meaningful in lane 0 of the comb's row)
        float apw[4];          // damp^12, ^24, ^48, ^96: the affine maps' slopes at scan distances 1, 2, 4, 8
        {
            const float d2 = damp * damp, d4 = d2 * d2;
            float a = (d4 * d4) * d4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { apw[k] = a; a *= a; }
        }
        // the line's values for tile k + 1 are read at the end of tile k: they were written a whole delay (>= 2 tiles) ago, so
        // the LDS round trip is off the tile's dependent chain (read -> 11 FMAs -> scan -> 12 FMAs -> write)
        // A lane's run of RV_RUN consecutive slots is always read and written STRAIGHT: every line carries RV_RUN - 1 junk floats in
        // front of it and a mirror of its first RV_RUN - 1 slots (+ junk) behind it (reverb_geometry).  The run that crosses the
        // end of the circular line is written a second time RV_RUN - 1 .. 1 slots in front of slot 0 (its tail lands on the real
        // slots 0 ..), the run that starts inside the first RV_RUN - 1 slots a second time behind the end (the mirror); reads never
        // wrap.  (Rounds 1 - 4 wrapped slot by slot under `if (!straight)`: ~100 instructions that only one lane of one comb
        // needs, but with 16 combs wrapping every 6 - 8 tiles nearly EVERY tile had one, and all waves meet at the tile's barrier.)
        int base = RV_RUN * cl;  // cpos = 0
        float o[RV_RUN];
#pragma unroll
        for (int i = 0; i < RV_RUN; ++i) o[i] = 0.0f;  // the lines start empty
        for (int k = 0; k <= ntiles; ++k) {
            RV_BARRIER();
            if (k == ntiles) break;
            if (RV_ABL & 1) continue;
            const float *in_c = s_in + (k & 1) * RV_TT + RV_RUN * cl;
            float *cmb = s_comb + (k & 1) * RV_CB + cidx * RV_TT + (cidx >= 8 ? RV_PAD : 0) + RV_RUN * cl;
            float pq[RV_RUN], w[RV_RUN];
            f4 nin[RV_RUN / 4];
#pragma unroll
            for (int q = 0; q < RV_RUN / 4; ++q) nin[q] = *(const f4 *)(in_c + 4 * q);
            // the lane's 12 damping steps as one affine map of the state entering it: out = damp^12 in + b
#pragma unroll
            for (int i = 0; i < RV_RUN; ++i) pq[i] = o[i] * omd;
            float b = pq[0];
#pragma unroll
            for (int i = 1; i < RV_RUN; ++i) b = fmaf(b, damp, pq[i]);
            b = fmaf(apw[0], cl == 0 ? last_in : 0.0f, b);  // the state entering the tile rides on lane 0's map
            // inclusive scan over the comb's 16 lanes = one DPP row: shifts 1, 2, 4, 8 (lanes shifted in from outside the row read 0)
            b = fmaf(apw[0], rv_dpp<0x111>(b), b);
            b = fmaf(apw[1], rv_dpp<0x112>(b), b);
            b = fmaf(apw[2], rv_dpp<0x114>(b), b);
            b = fmaf(apw[3], rv_dpp<0x118>(b), b);
            // state entering this lane's run = the scan of the lane before it (lane 0 of the row: the tile's entering state)
            const float prev = rv_dpp<0x111>(b);  // row_shr:1
            float filt = cl == 0 ? last_in : prev;
#pragma unroll
            for (int i = 0; i < RV_RUN; ++i) {
                filt = fmaf(filt, damp, pq[i]);
                w[i] = nin[i >> 2][i & 3] + (filt * fbk);
            }
#pragma unroll
            for (int i = 0; i < RV_RUN; ++i) cbuf[base + i] = w[i];
            {
                const bool cross = base + (RV_RUN - 1) >= csz, head = base < RV_RUN - 1;
                if (cross || head) {   // one lane of a comb, once per trip around its line
                    float *c2 = cbuf + (cross ? base - csz : base + csz);
#pragma unroll
                    for (int i = 0; i < RV_RUN; ++i) c2[i] = w[i];
                }
            }
#pragma unroll
            for (int q = 0; q < RV_RUN / 4; ++q) *(f4 *)(cmb + 4 * q) = (f4){o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
            // the row's last lane holds the state leaving the tile: rotate it into lane 0 (row_ror:1; the other lanes do not use it)
            last_in = rv_dpp<0x121>(filt);
            cpos += RV_TT;
            cpos = cpos >= csz ? cpos - csz : cpos;
            base = cpos + RV_RUN * cl;
            base = base >= csz ? base - csz : base;
#pragma unroll
            for (int i = 0; i < RV_RUN; ++i) o[i] = cbuf[base + i];   // (runs into the mirror behind the line's end)
        }
    } else if (wv < NCW + NAW) {
        // ---- all-pass role: thread u = (channel u & 1, sample u >> 1) of tile k - 1: comb sum, 4 series all-passes, width mix
        // (the other channel of the sample is the neighbouring lane), wet/dry, store.  HALF: thread u = sample u of the one channel
        const int u = tid - NCW * 64, c2 = HALF ? my_ch : (u & 1), t2 = HALF ? u : (u >> 1);
        float *yc_g = c2 == 0 ? yl : yr;
        int appos[4] = {0, 0, 0, 0}, apsz[4], apoff[4];  // geometry copied out of the kernel-argument struct once
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            apsz[j] = g.ap_size[c2 * 4 + j];
            apoff[j] = g.ap_off[c2 * 4 + j];
        }
        int m3 = 2;            // (k - 1) % 3 at k = 0
        int64_t t_out = -(int64_t)RV_TT + t2;
        for (int k = 0; k <= ntiles; ++k) {
            RV_BARRIER();
            if (k >= 1 && !(RV_ABL & 2)) {
                const float *cmb = s_comb + ((k - 1) & 1) * RV_CB + c2 * (8 * RV_TT + RV_PAD) + t2;
                float cs[8], bv[4];
                float *abp[4];
#pragma unroll
                for (int j = 0; j < 8; ++j) cs[j] = cmb[j * RV_TT];
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // every line is read before any is written: one LDS round trip, not four
                    int p = appos[j] + t2;
                    p = p >= apsz[j] ? p - apsz[j] : p;
                    abp[j] = state + apoff[j] + p;
                    bv[j] = *abp[j];
                    appos[j] += RV_TT;
                    appos[j] = appos[j] >= apsz[j] ? appos[j] - apsz[j] : appos[j];
                }
                const float xd = HALF ? 0.0f : s_x[m3 * RV_XB + c2 * (RV_TT + RV_PAD) + t2];
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += cs[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *abp[j] = acc + (bv[j] * 0.5f);
                    acc = bv[j] - acc;
                }
                if (HALF) {
                    if (t_out < L) yc_g[t_out] = acc;   // the channel's wet signal; mixed by k_reverb_mix
                } else {
                    const float other = rv_dpp<0xB1>(acc);  // quad_perm [1,0,3,2]: the other channel of this sample
                    if (t_out < L) yc_g[t_out] = rv_mix(acc, other, xd, wet1, wet2, dry);
                }
            }
            m3 = m3 == 2 ? 0 : m3 + 1;
            t_out += RV_TT;
        }
    } else {
        // ---- staging role: thread v < RV_TT / 2 owns samples 2 v, 2 v + 1 of both channels.  Tile k + 1 is written to LDS in
        // step k out of the ring slot filled RV_PD steps earlier; loads are clamped, not branched around.
        const int v = tid - (NCW + NAW) * 64;
        const bool act = v < RV_TT / 2;
        const int64_t Lm1 = L - 1;
        // Register ring of RV_PD tiles, loaded by inline-asm global_load_dword and waited for by hand: (l0, l1, r0, r1) of tile t
        // in slot t % RV_PD.  Why by hand: given plain loads hipcc waits for them where it loses count -- right behind the load
        // when the end-of-signal select follows it (rounds 1 - 4: vmcnt(0) on the load just issued, one full memory latency per
        // tile in front of the barrier every other wave is waiting at), and with vmcnt(0) at the head of every unrolled turn even
        // when nothing touches the values early (it does not carry its counters over the loop's back edge).  This wave issues no
        // other vector-memory instruction, so the count is exact: a slot is consumed RV_PD fetches after its own, i.e. with
        // 4 (RV_PD - 1) younger loads in flight.  The zeroing of samples past the end of the signal happens in put().
        // The ring lives in FIXED registers v64 .. v79 (slot s: v[64 + 4 s .. 67 + 4 s]) that only these asm statements name (declared
        // as clobbers): the compiler never holds a ring value in a register of its own before the wait that makes it valid, so it
        // cannot copy or spill a register whose load is still in flight (with "=v" outputs on the loads it legally
        // could).  take() waits and moves a slot's four values into compiler-visible registers in ONE asm statement.  Build check
        // (tests/test_host_logic.py): the compiler's own code of this kernel stays below v64.
#define RV_RING_FETCH(R0_, R1_, R2_, R3_)                                                                                          \
    asm volatile("global_load_dword " R0_ ", %0, off\n\tglobal_load_dword " R1_ ", %1, off\n\tglobal_load_dword " R2_ ", %2, off\n\t" \
                 "global_load_dword " R3_ ", %3, off" : : "v"(xl + j0), "v"(xl + j1), "v"(xr + j0), "v"(xr + j1) : "memory", R0_, R1_, R2_, R3_)
#define RV_RING_TAKE(N_, R0_, R1_, R2_, R3_)                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #N_ ")\n\tv_mov_b32 %0, " R0_ "\n\tv_mov_b32 %1, " R1_ "\n\tv_mov_b32 %2, " R2_ "\n\tv_mov_b32 %3, " R3_ \
                 : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : : "memory", R0_, R1_, R2_, R3_)
        auto fetch = [&](int64_t t, int slot) {
            const int64_t i0 = t * RV_TT + 2 * (act ? v : 0), i1 = i0 + 1;
            const int64_t j0 = i0 < Lm1 ? i0 : Lm1, j1 = i1 < Lm1 ? i1 : Lm1;
            switch (slot) {   // a compile-time constant at every call site (unrolled loops)
                case 0: RV_RING_FETCH("v64", "v65", "v66", "v67"); break;
                case 1: RV_RING_FETCH("v68", "v69", "v70", "v71"); break;
                case 2: RV_RING_FETCH("v72", "v73", "v74", "v75"); break;
                default: RV_RING_FETCH("v76", "v77", "v78", "v79"); break;
            }
        };
        auto take0 = [&](int slot, float &a0, float &a1, float &b0, float &b1) {    // everything issued so far has landed
            switch (slot) {
                case 0: RV_RING_TAKE(0, "v64", "v65", "v66", "v67"); break;
                case 1: RV_RING_TAKE(0, "v68", "v69", "v70", "v71"); break;
                case 2: RV_RING_TAKE(0, "v72", "v73", "v74", "v75"); break;
                default: RV_RING_TAKE(0, "v76", "v77", "v78", "v79"); break;
            }
        };
        auto take12 = [&](int slot, float &a0, float &a1, float &b0, float &b1) {   // ... all but the twelve youngest loads
            switch (slot) {
                case 0: RV_RING_TAKE(12, "v64", "v65", "v66", "v67"); break;
                case 1: RV_RING_TAKE(12, "v68", "v69", "v70", "v71"); break;
                case 2: RV_RING_TAKE(12, "v72", "v73", "v74", "v75"); break;
                default: RV_RING_TAKE(12, "v76", "v77", "v78", "v79"); break;
            }
        };
        auto put = [&](float a0, float a1, float b0, float b1, int64_t t, int buf3, int buf2) {
            const int64_t i0 = t * RV_TT + 2 * (act ? v : 0);
            const bool in0 = i0 < L, in1 = i0 + 1 < L;
            const float l0 = in0 ? a0 : 0.0f, l1 = in1 ? a1 : 0.0f, r0 = in0 ? b0 : 0.0f, r1 = in1 ? b1 : 0.0f;
            if (act) {
                *(f2 *)(s_x + buf3 * RV_XB + 2 * v) = (f2){l0, l1};
                *(f2 *)(s_x + buf3 * RV_XB + RV_TT + RV_PAD + 2 * v) = (f2){r0, r1};
                *(f2 *)(s_in + buf2 * RV_TT + 2 * v) = (f2){(l0 + r0) * 0.015f, (l1 + r1) * 0.015f};
            }
        };
        static_assert(RV_PD == 4, "the wait count of take12 is 4 (RV_PD - 1) = 12 loads; four ring slots are named");
        float q0, q1, q2, q3;
        fetch(0, 0);
        take0(0, q0, q1, q2, q3);
        put(q0, q1, q2, q3, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < RV_PD; ++j) fetch(j + 1, (j + 1) % RV_PD);
        int m3 = 1;  // (k + 1) % 3 at k = 0
        for (int k0 = 0; k0 <= ntiles; k0 += RV_PD) {   // ntiles + 1 is a multiple of RV_PD
#pragma unroll
            for (int kk = 0; kk < RV_PD; ++kk) {
                const int k = k0 + kk;
                const int sl = (kk + 1) % RV_PD;
                RV_BARRIER();
                take12(sl, q0, q1, q2, q3);   // tile k + 1 has landed (the three fetches behind it may still be in flight)
                put(q0, q1, q2, q3, (int64_t)k + 1, m3, (k + 1) & 1);   // zeros past the end of the signal
                fetch((int64_t)k + 1 + RV_PD, sl);
                m3 = m3 == 2 ? 0 : m3 + 1;
            }
        }
#undef RV_RING_FETCH
#undef RV_RING_TAKE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of the ring may land after the wave has ended
    }
}

// per-channel form: wet (pop, 2, L) from k_reverb<true> + the stage's input -> out (in place over the input is fine: pointwise)
__global__ __launch_bounds__(256) void k_reverb_mix(InView in, const float *__restrict__ wet, float *__restrict__ out, int64_t cand_stride,
                                                     int64_t L, const double *__restrict__ coef) {
    const int cand = blockIdx.y;
    const double *cf = coef + (int64_t)cand * COEF_STRIDE;
    const float wet1 = (float)cf[2], wet2 = (float)cf[3], dry = (float)cf[4];
    const float *xl = in_ptr(in, cand, 0), *xr = in_ptr(in, cand, 1);
    const float *wl = wet + (int64_t)cand * 2 * L, *wr = wl + L;
    float *yl = out + (int64_t)cand * cand_stride, *yr = yl + L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = wl[i], b = wr[i], x0 = xl[i], x1 = xr[i];
        yl[i] = rv_mix(a, b, x0, wet1, wet2, dry);
        yr[i] = rv_mix(b, a, x1, wet1, wet2, dry);
    }
}

// ------------------------------------------------------------------------------------------------
// peak / normalise (style_transfer.py:113)
// ------------------------------------------------------------------------------------------------
// per_cand = floats scanned per candidate, stride = floats between candidates (>= per_cand: a mono
// stage of a chain that ends stereo only fills the first channel of each candidate's slot)
__global__ __launch_bounds__(256) void k_peak(const float *__restrict__ a, int64_t per_cand, int64_t stride,
                                               float *__restrict__ peaks) {
    __shared__ float red[4];
    const int cand = blockIdx.y;
    const float4 *p4 = (const float4 *)(a + (int64_t)cand * stride);
    const int64_t n4 = per_cand / 4;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = p4[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax((unsigned int *)&peaks[cand], __float_as_uint(m));  // non-negative floats order as uints
    }
}

__global__ __launch_bounds__(256) void k_peak_scalar(const float *__restrict__ a, int64_t per_cand, int64_t stride,
                                                      float *__restrict__ peaks) {
    __shared__ float red[4];
    const int cand = blockIdx.y;
    const float *p = a + (int64_t)cand * stride;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_cand; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax((unsigned int *)&peaks[cand], __float_as_uint(m));
    }
}

__global__ __launch_bounds__(256) void k_normalize(float *__restrict__ a, int64_t per_cand, int64_t stride,
                                                    const float *__restrict__ peaks) {
    const int cand = blockIdx.y;
    const float d = fmaxf(peaks[cand], 1e-8f);
    float *p = a + (int64_t)cand * stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_cand; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = p[i] / d;
}

// ================================================================================================
// host side
// ================================================================================================
static int fx_channels_after(const stito_fx_desc &fx, int c) { return (fx.num_channels == 2 && c == 1) ? 2 : c; }

static void reverb_geometry(double sr, ReverbGeom &g) {
    static const int comb_t[8] = {1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617};
    static const int ap_t[4] = {556, 441, 341, 225};
    const int isr = (int)sr;
    int off = 0;
    // a comb line in LDS: [RV_RUN - 1 junk][the line][mirror of its first RV_RUN - 1 slots + RV_RUN junk] (k_reverb, comb role)
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 8; ++j) {
            g.comb_size[c * 8 + j] = (int)(((int64_t)isr * (comb_t[j] + (c ? 23 : 0))) / 44100);
            g.comb_off[c * 8 + j] = off + (RV_RUN - 1);
            off += g.comb_size[c * 8 + j] + (RV_RUN - 1) + (2 * RV_RUN - 1);
        }
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 4; ++j) {
            g.ap_size[c * 4 + j] = (int)(((int64_t)isr * (ap_t[j] + (c ? 23 : 0))) / 44100);
            g.ap_off[c * 4 + j] = off;
            off += g.ap_size[c * 4 + j];
        }
    g.state_floats = (off + 3) & ~3;
}

static int grid_x_for(int64_t L, int streams) {
    int64_t want = (L + 255) / 256;
    int64_t cap = (256 * 16 + streams - 1) / streams;  // ~16 blocks per CU over the whole launch
    cap = cap < 1 ? 1 : cap;
    return (int)(want < cap ? want : cap);
}

}  // namespace stito

using namespace stito;

extern "C" int stito_fx_num_params(int kind) {
    if (kind < 0 || kind >= STITO_FX_NUM_KINDS) return STITO_E_INVALID;
    return h_nparams[kind];
}

extern "C" int stito_chain_out_channels(const stito_fx_desc *chain, int n_fx, int in_channels) {
    int c = in_channels;
    for (int i = 0; i < n_fx; ++i) c = fx_channels_after(chain[i], c);
    return c;
}

extern "C" int stito_chain_num_dims(const stito_fx_desc *chain, int n_fx) {
    int d = 0;
    for (int i = 0; i < n_fx; ++i) {
        if (chain[i].kind < 0 || chain[i].kind >= STITO_FX_NUM_KINDS) return STITO_E_INVALID;
        d += h_nparams[chain[i].kind] + (chain[i].has_bypass ? 1 : 0);
    }
    return d;
}

// convolution reverb: spectra of the input blocks and of every candidate's IR partitions
static size_t conv_reverb_bytes(const stito_fx_desc *chain, int n_fx, int64_t n_samples, int pop) {
    size_t cr = 0;
    for (int i = 0; i < n_fx; ++i)
        if (chain[i].kind == STITO_FX_NOISE_REVERB) {
            const size_t need = conv_reverb_workspace_bytes(pop * 2, n_samples, chain[i].aux_len > 0 ? chain[i].aux_len : 1);
            cr = need > cr ? need : cr;
        }
    return cr;
}

// Freeverb as two workgroups per candidate (k_reverb<true> + k_reverb_mix) while that still leaves every workgroup its own CU:
// a population of 32 runs 64 half-size workgroups instead of 32 (0.81 -> see profiles/round6_small_pop.txt).  STITO_REVERB_SPLIT=0 / 1 forces.
static bool reverb_split(int pop) {
    if (const char *e = getenv("STITO_REVERB_SPLIT")) return atoi(e) != 0;   // (read per call: the tests flip it)
    DeviceInfo d;
    if (device_info(d) != STITO_OK) return false;
    return 2 * pop <= d.cus;
}
static size_t reverb_split_bytes(const stito_fx_desc *chain, int n_fx, int64_t n_samples, int pop) {   // the wet signals (pop, 2, L)
    bool has = false;
    for (int i = 0; i < n_fx; ++i) has |= chain[i].kind == STITO_FX_REVERB;
    return has && reverb_split(pop) ? align_up((size_t)pop * 2 * n_samples * sizeof(float), 256) : 0;
}

extern "C" size_t stito_render_workspace_bytes(const stito_fx_desc *chain, int n_fx, int in_channels,
                                               int64_t n_samples, int pop) {
    (void)in_channels;  // the compressor's share is sized for two channels per candidate whatever the chain does
    size_t coef = align_up((size_t)(n_fx > 0 ? n_fx : 1) * pop * COEF_STRIDE * sizeof(double), 256);
    bool has_comp = false;
    for (int i = 0; i < n_fx; ++i) has_comp |= chain[i].kind == STITO_FX_COMPRESSOR;
    size_t env = has_comp ? compressor_workspace_bytes(pop * 2, n_samples) : 0;  // block functions + boundary states
    return coef + env + conv_reverb_bytes(chain, n_fx, n_samples, pop) + reverb_split_bytes(chain, n_fx, n_samples, pop) +
           align_up((size_t)pop * sizeof(float), 256) + 256;
}

static int peak_strided(const float *audio_dev, int pop, int64_t per, int64_t stride, float *peaks_dev, hipStream_t st) {
    STITO_TRY(zero_async(peaks_dev, sizeof(float) * pop, st));
    if (((uintptr_t)audio_dev & 15) == 0 && per % 4 == 0 && stride % 4 == 0) {
        dim3 grid(grid_x_for(per / 4, pop), pop);
        hipLaunchKernelGGL(k_peak, grid, dim3(256), 0, st, audio_dev, per, stride, peaks_dev);
    } else {
        dim3 grid(grid_x_for(per, pop), pop);
        hipLaunchKernelGGL(k_peak_scalar, grid, dim3(256), 0, st, audio_dev, per, stride, peaks_dev);
    }
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

static int normalize_strided(float *audio_dev, int pop, int64_t per, int64_t stride, const float *peaks_dev, hipStream_t st) {
    dim3 grid(grid_x_for(per, pop), pop);
    hipLaunchKernelGGL(k_normalize, grid, dim3(256), 0, st, audio_dev, per, stride, peaks_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_peak(const float *audio_dev, int pop, int channels, int64_t n_samples, float *peaks_dev,
                          void *stream) {
    STITO_REQUIRE(pop > 0 && channels > 0 && n_samples > 0, STITO_E_INVALID, "stito_peak: empty input");
    const int64_t per = (int64_t)channels * n_samples;
    return peak_strided(audio_dev, pop, per, per, peaks_dev, (hipStream_t)stream);
}

extern "C" int stito_normalize_audio(float *audio_dev, int pop, int channels, int64_t n_samples,
                                     const float *peaks_dev, void *stream) {
    const int64_t per = (int64_t)channels * n_samples;
    return normalize_strided(audio_dev, pop, per, per, peaks_dev, (hipStream_t)stream);
}

extern "C" int stito_render_population(const stito_fx_desc *chain, int n_fx, const float *x_dev, int in_channels,
                                       int64_t n_samples, const double *w_dev, int pop, int n_dims,
                                       double sample_rate, float *audio_dev, float *peaks_dev,
                                       void *workspace_dev, size_t workspace_bytes, void *stream) {
    return stito_render_population_multi(chain, n_fx, x_dev, 1, in_channels, n_samples, w_dev, pop, n_dims, sample_rate,
                                         audio_dev, peaks_dev, workspace_dev, workspace_bytes, stream);
}

extern "C" int stito_render_population_multi(const stito_fx_desc *chain, int n_fx, const float *x_dev, int n_inputs,
                                             int in_channels, int64_t n_samples, const double *w_dev, int pop,
                                             int n_dims, double sample_rate, float *audio_dev, float *peaks_dev,
                                             void *workspace_dev, size_t workspace_bytes, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(n_inputs >= 1 && pop % n_inputs == 0, STITO_E_INVALID,
                  "population %d is not a multiple of the number of inputs %d", pop, n_inputs);
    STITO_REQUIRE(n_fx >= 0 && n_fx <= 16, STITO_E_INVALID, "chain length %d not in [0,16]", n_fx);
    STITO_REQUIRE(in_channels == 1 || in_channels == 2, STITO_E_INVALID, "in_channels must be 1 or 2, got %d", in_channels);
    STITO_REQUIRE(pop > 0 && n_samples > 0, STITO_E_INVALID, "empty population or audio");
    const int dims = stito_chain_num_dims(chain, n_fx);
    STITO_REQUIRE(dims >= 0, STITO_E_INVALID, "Plugin must contain a known effect kind");
    STITO_REQUIRE(dims == n_dims, STITO_E_INVALID, "parameter vector has %d dims, chain consumes %d", n_dims, dims);
    const size_t need = stito_render_workspace_bytes(chain, n_fx, in_channels, n_samples, pop);
    STITO_REQUIRE(workspace_bytes >= need, STITO_E_WORKSPACE, "render workspace: have %zu need %zu", workspace_bytes, need);
    const int C_out = stito_chain_out_channels(chain, n_fx, in_channels);
    const int64_t L = n_samples;
    const int64_t cand_stride = (int64_t)C_out * L;

    char *ws = (char *)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    double *coef = (double *)ws;
    float *envbuf = (float *)(ws + align_up((size_t)(n_fx > 0 ? n_fx : 1) * pop * COEF_STRIDE * sizeof(double), 256));
    bool has_comp = false;
    for (int i = 0; i < n_fx; ++i) has_comp |= chain[i].kind == STITO_FX_COMPRESSOR;
    char *crbuf = (char *)envbuf + (has_comp ? compressor_workspace_bytes(pop * 2, n_samples) : 0);
    float *rvbuf = (float *)(crbuf + conv_reverb_bytes(chain, n_fx, n_samples, pop));   // wet signals of the per-channel Freeverb
    const bool rv_split = reverb_split_bytes(chain, n_fx, n_samples, pop) > 0;
    // per-stage peaks (normalize_stages) live in the last pop floats of the workspace
    float *stage_peaks = (float *)(ws + (need - 256 - align_up((size_t)pop * sizeof(float), 256)));

    if (n_fx > 0) {
        ChainArgs args;
        args.n_fx = n_fx;
        int off = 0;
        for (int i = 0; i < n_fx; ++i) {
            args.fx[i] = chain[i];
            STITO_REQUIRE(chain[i].w_offset == off, STITO_E_INVALID, "fx %d: w_offset %d, expected %d", i, chain[i].w_offset, off);
            STITO_REQUIRE(chain[i].num_channels == 1 || chain[i].num_channels == 2, STITO_E_INVALID, "fx %d: num_channels", i);
            off += h_nparams[chain[i].kind] + (chain[i].has_bypass ? 1 : 0);
        }
        const int n = pop * n_fx;
        hipLaunchKernelGGL(k_prepare, dim3((n + 127) / 128), dim3(128), 0, st, args, w_dev, pop, n_dims, sample_rate, coef);
        STITO_LAUNCH_CHECK();
    }

    InView in{x_dev, 0, L, in_channels, pop / n_inputs, (int64_t)in_channels * L};
    int C = in_channels;
    bool in_buffer = false, peaks_done = false;
    for (int i = 0; i < n_fx; ++i) {
        bool fused_next = false;  // stage i + 1 was absorbed by stage i's store
        const stito_fx_desc &fx = chain[i];
        const int Cn = fx_channels_after(fx, C);
        const double *cf = coef + (int64_t)i * pop * COEF_STRIDE;
        const int S = pop * Cn;
        if (Cn != C && in_buffer) {  // mono -> stereo in place: duplicate channel 0 first (style_transfer.py:94-95)
            hipLaunchKernelGGL(k_pointwise<-1>, dim3(grid_x_for(L, S), S), dim3(256), 0, st, in, audio_dev, cand_stride, Cn, L, cf);
            STITO_LAUNCH_CHECK();
            in = InView{audio_dev, cand_stride, L, Cn};
        }
        switch (fx.kind) {
            case STITO_FX_PARAMETRIC_EQ: {
                // the store of the EQ absorbs a Gain stage that follows it and, when it is then the end of the
                // chain, the final peak scan (bench chain: EQ -> gain -> peak = three passes over the audio less)
                PostOp post;
                if (i + 1 < n_fx && chain[i + 1].kind == STITO_FX_GAIN && fx_channels_after(chain[i + 1], Cn) == Cn &&
                    !(fx.flags & STITO_FX_FLAG_NORMALIZE_AFTER) && !(chain[i + 1].flags & STITO_FX_FLAG_NORMALIZE_AFTER)) {
                    post.gain_coef = coef + (int64_t)(i + 1) * pop * COEF_STRIDE;
                    fused_next = true;
                }
                if (peaks_dev != nullptr && i + (fused_next ? 2 : 1) == n_fx && Cn == C_out && !(fx.flags & STITO_FX_FLAG_NORMALIZE_AFTER)) {
                    STITO_TRY(zero_async(peaks_dev, sizeof(float) * pop, st));
                    post.peaks = peaks_dev;
                    peaks_done = true;
                }
                STITO_REQUIRE(L < ((int64_t)1 << 31) - 65536, STITO_E_UNSUPPORTED, "parametric EQ: %lld samples per stream (k_eq indexes in 32 bits)", (long long)L);
                hipLaunchKernelGGL(k_eq, dim3(S), dim3(EQ_NC), 0, st, in, post, audio_dev, cand_stride, Cn, L, cf);
                break;
            }
            case STITO_FX_COMPRESSOR: {
                const int rc = compressor_stage(in, audio_dev, cand_stride, pop, Cn, L, cf, envbuf, st);
                if (rc) return rc;
                break;
            }
            case STITO_FX_DISTORTION:
                hipLaunchKernelGGL(k_pointwise<STITO_FX_DISTORTION>, dim3(grid_x_for(L, S), S), dim3(256), 0, st, in, audio_dev, cand_stride, Cn, L, cf);
                break;
            case STITO_FX_GAIN:
                hipLaunchKernelGGL(k_pointwise<STITO_FX_GAIN>, dim3(grid_x_for(L, S), S), dim3(256), 0, st, in, audio_dev, cand_stride, Cn, L, cf);
                break;
            case STITO_FX_DELAY: {
                const int64_t dmax = (int64_t)(1.0 * sample_rate) + 1;
                const int64_t nthreads = L < dmax ? L : dmax;
                hipLaunchKernelGGL(k_delay, dim3((unsigned)((nthreads + 255) / 256), pop, (in.in_ch == 1 && Cn == 2) ? 1 : Cn), dim3(256), 0, st, in, audio_dev, cand_stride, Cn, L, cf);
                break;
            }
            case STITO_FX_REVERB: {
                STITO_REQUIRE(Cn == 2, STITO_E_INVALID, "Reverb must be declared with num_channels=2 (run_optim.py:401-406)");
                ReverbGeom g;
                reverb_geometry(sample_rate, g);
                int mins = g.ap_size[0];
                for (int k = 0; k < 8; ++k) mins = g.ap_size[k] < mins ? g.ap_size[k] : mins;
                const size_t lds = (size_t)(g.state_floats + RV_TILE_FLOATS) * sizeof(float);
                int minc = g.comb_size[0];
                for (int k = 0; k < 16; ++k) minc = g.comb_size[k] < minc ? g.comb_size[k] : minc;
                STITO_REQUIRE(mins >= RV_TT && minc >= 2 * RV_TT + RV_RUN && lds <= 160 * 1024, STITO_E_UNSUPPORTED,
                              "Reverb: sample rate %.0f needs delay lines outside the LDS-resident design", sample_rate);
                if (rv_split) {
                    constexpr int NT = (RV_COMB_WAVES / 2 + RV_AP_WAVES / 2 + RV_STAGE_WAVES) * 64;
                    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)k_reverb<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_reverb<true>, dim3(2 * pop), dim3(NT), lds, st, in, rvbuf, (int64_t)2 * L, L, cf, g);
                    STITO_LAUNCH_CHECK();
                    hipLaunchKernelGGL(k_reverb_mix, dim3(grid_x_for(L, pop), pop), dim3(256), 0, st, in, rvbuf, audio_dev, cand_stride, L, cf);
                } else {
                    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)k_reverb<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_reverb<false>, dim3(pop), dim3(RV_THREADS), lds, st, in, audio_dev, cand_stride, L, cf, g);
                }
                break;
            }
            case STITO_FX_CHORUS: {
                const int rc = chorus_stage(in, audio_dev, cand_stride, pop, Cn, L, cf, fx.aux_dev, fx.aux_len, sample_rate, st);
                if (rc) return rc;
                break;
            }
            case STITO_FX_NOISE_REVERB: {
                STITO_REQUIRE(Cn == 2, STITO_E_INVALID, "NoiseShapedReverb must be declared with num_channels=2");
                const int rc = conv_reverb_stage(in, audio_dev, cand_stride, pop, L, cf, fx.aux_dev, fx.aux_len, crbuf, st);
                if (rc) return rc;
                break;
            }
            default:
                STITO_REQUIRE(false, STITO_E_INVALID, "Plugin must contain a known effect kind (got %d)", fx.kind);
        }
        STITO_LAUNCH_CHECK();
        C = Cn;
        in = InView{audio_dev, cand_stride, L, C};
        in_buffer = true;
        if (fused_next) ++i;
        if (fx.flags & STITO_FX_FLAG_NORMALIZE_AFTER) {  // normalize_stages (style_transfer.py:106-107)
            int rc = peak_strided(audio_dev, pop, (int64_t)C * L, cand_stride, stage_peaks, st);
            if (rc) return rc;
            rc = normalize_strided(audio_dev, pop, (int64_t)C * L, cand_stride, stage_peaks, st);
            if (rc) return rc;
        }
    }
    if (!in_buffer) {  // empty chain: broadcast x
        const int S = pop * C_out;
        hipLaunchKernelGGL(k_pointwise<-1>, dim3(grid_x_for(L, S), S), dim3(256), 0, st, in, audio_dev, cand_stride, C_out, L, (const double *)workspace_dev);
        STITO_LAUNCH_CHECK();
    }
    if (peaks_dev && !peaks_done) return stito_peak(audio_dev, pop, C_out, L, peaks_dev, stream);
    return STITO_OK;
}
