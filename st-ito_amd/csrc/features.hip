// features.hip -- hand-crafted audio features of st_ito/features.py on gfx950 (SURVEY 8(f) rank 4;
// named in BASELINE.json's north_star next to dsp.py: "windowed radix-2 FFT in LDS, mel matvec").
//
// Replaces (reference file:line):
//   compute_barkspectrum       st_ito/features.py:166-232  rectangular-window STFT (n_fft 32 768, hop n_fft/4,
//                              centred, reflect pad) -> |X| -> mean over frames -> bark filterbank -> log
//   compute_rms_energy         features.py:235-245
//   compute_crest_factor       features.py:248-264 (including its per-sample cross-channel "peak normalise")
//   compute_spectral_centroid  features.py:302-333 -> torchaudio SpectralCentroid(n_fft 2048, hop 1024, Hann)
//
// One workgroup per (item, signal) walks the frames of its signal: n_fft real samples are packed as
// n_fft/2 complex points and transformed IN PLACE in LDS (radix-2 decimation in frequency, 128 KB
// for n_fft = 32 768 -- a ping-pong Stockham pair would not fit the 160 KB), the spectrum is read
// back through the bit-reversed index.  The time mean of |X| lives in registers (a thread owns its
// bins for the whole signal), so the result is deterministic and nothing but the audio is read
// from HBM: the kernels are bound by LDS butterfly traffic.
#include "common.h"
#include "dsp_view.h"

namespace stito {

enum { FEAT_MONO = 0, FEAT_STEREO = 1, FEAT_MIDSIDE = 2 };

__device__ __forceinline__ float2 ft_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ int64_t ft_reflect(int64_t i, int64_t L) {
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    return i;
}

__device__ __forceinline__ float ft_signal(const float *xl, const float *xr, int64_t i, int mode, int sig) {
    if (mode == FEAT_MONO) return xr ? (xl[i] + xr[i]) / 2.0f : xl[i];  // torch mean over the channel axis
    if (mode == FEAT_STEREO) return sig == 0 ? xl[i] : xr[i];
    return sig == 0 ? xl[i] + xr[i] : xl[i] - xr[i];                    // "mid-side" without halving (features.py:201-203)
}

// in-place radix-2 DIF FFT of N2 = 1 << LOG2 complex points in LDS; result element k ends at brev(k)
template <int NT>
__device__ void ft_fft_dif(float2 *z, const float2 *__restrict__ tw, int log2n2, int tid) {
    const int N2 = 1 << log2n2, half = N2 >> 1;
    for (int span = half, sh = 1; span >= 1; span >>= 1, ++sh) {
        // butterfly (a, a + span): twiddle exp(-2 pi i j / (2 span)) = tw[j * (N2 / span)] = tw[j << sh]
        for (int i = tid; i < half; i += NT) {
            const int j = i & (span - 1);
            const int a = ((i - j) << 1) + j;
            const float2 u = z[a], v = z[a + span];
            z[a] = make_float2(u.x + v.x, u.y + v.y);
            z[a + span] = ft_cmul(make_float2(u.x - v.x, u.y - v.y), tw[j << sh]);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int ft_brev(int k, int bits) { return (int)(__brev((unsigned)k) >> (32 - bits)); }

// |X[k]| of the 2*N2 real samples packed in z (after ft_fft_dif), k in [0, N2]
__device__ __forceinline__ float ft_mag(const float2 *z, const float2 *__restrict__ tw, int k, int log2n2) {
    const int N2 = 1 << log2n2;
    const float2 zk = z[ft_brev(k & (N2 - 1), log2n2)], zn = z[ft_brev((N2 - k) & (N2 - 1), log2n2)];
    const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    const float2 O = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
    const float2 w = k < N2 ? tw[k] : make_float2(-1.0f, 0.0f);
    const float2 wo = ft_cmul(w, O);
    const float re = E.x + wo.x, im = E.y + wo.y;
    return sqrtf(re * re + im * im);
}

// MODE 0: bark spectrum -> out (item, n_bands, n_sig) = log(fb . mean_t |X| + 1e-8)
// MODE 1: spectral centroid per frame -> out (item * n_sig, T)
template <int NT, int MODE>
__global__ __launch_bounds__(NT) void k_stft_feature(const float *__restrict__ audio, int C, int64_t L, int mode, int n_sig,
                                                      int log2n2, int hop, int64_t T, const float *__restrict__ window,
                                                      const float2 *__restrict__ tw, const float *__restrict__ fb, int n_bands,
                                                      float nyq_step, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float2 ft_lds[];
    __shared__ float red[2][NT / 64];
    float2 *z = ft_lds;
    const int N2 = 1 << log2n2;
    const int item = blockIdx.x / n_sig, sig = blockIdx.x % n_sig, tid = threadIdx.x;
    const float *xl = audio + (int64_t)item * C * L;
    const float *xr = C == 2 ? xl + L : nullptr;
    if (MODE == 1) xl += (int64_t)sig * L;  // centroid: every channel on its own
    constexpr int MAXB = 17;                // bins per thread: N2 / NT (+ the Nyquist bin on thread 0)
    float acc[MAXB];
#pragma unroll
    for (int u = 0; u < MAXB; ++u) acc[u] = 0.0f;

    for (int64_t t = 0; t < T; ++t) {
        const int64_t base = t * hop - N2;  // center=True: frame t covers [t hop - n_fft/2, t hop + n_fft/2)
        for (int m = tid; m < N2; m += NT) {
            const int64_t i0 = ft_reflect(base + 2 * m, L), i1 = ft_reflect(base + 2 * m + 1, L);
            float a0, a1;
            if (MODE == 1) { a0 = xl[i0] * window[2 * m]; a1 = xl[i1] * window[2 * m + 1]; }
            else { a0 = ft_signal(xl, xr, i0, mode, sig); a1 = ft_signal(xl, xr, i1, mode, sig); }
            z[m] = make_float2(a0, a1);
        }
        __syncthreads();
        ft_fft_dif<NT>(z, tw, log2n2, tid);
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < MAXB - 1; ++u) {
                const int k = tid + u * NT;
                if (k < N2) acc[u] += ft_mag(z, tw, k, log2n2);
            }
            if (tid == 0) acc[MAXB - 1] += ft_mag(z, tw, N2, log2n2);
        } else {  // centroid = sum_k f_k |X_k| / sum_k |X_k|, f_k = k * (sr/2) / N2 (float32 linspace)
            float num = 0.0f, den = 0.0f;
            for (int k = tid; k <= N2; k += NT) {
                const float mg = ft_mag(z, tw, k, log2n2);
                num += (float)k * nyq_step * mg;
                den += mg;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { num += __shfl_xor(num, o); den += __shfl_xor(den, o); }
            if ((tid & 63) == 0) { red[0][tid >> 6] = num; red[1][tid >> 6] = den; }
            __syncthreads();
            if (tid == 0) {
                float n2 = 0.0f, d2 = 0.0f;
                for (int w = 0; w < NT / 64; ++w) { n2 += red[0][w]; d2 += red[1][w]; }
                out[(int64_t)blockIdx.x * T + t] = n2 / d2;  // NaN for a silent frame, scrubbed by the pooling kernel
            }
        }
        __syncthreads();  // z is rewritten by the next frame
    }
    if (MODE == 0) {
        // mean over frames -> LDS (float view of z), then one filterbank row per wave at a time
        float *mean = (float *)z;
        const float inv_t = 1.0f / (float)T;
#pragma unroll
        for (int u = 0; u < MAXB - 1; ++u) {
            const int k = tid + u * NT;
            if (k < N2) mean[k] = acc[u] * inv_t;
        }
        if (tid == 0) mean[N2] = acc[MAXB - 1] * inv_t;
        __syncthreads();
        const int wv = tid >> 6, lane = tid & 63, nfreq = N2 + 1;
        for (int b = wv; b < n_bands; b += NT / 64) {
            const float *row = fb + (int64_t)b * nfreq;
            float s = 0.0f;
            for (int k = lane; k < nfreq; k += 64) s = fmaf(row[k], mean[k], s);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            // the reference concatenates the signals on the LAST axis of (bs, n_bands, 1) before flattening: band-major
            if (lane == 0) out[((int64_t)item * n_bands + b) * n_sig + sig] = logf(s + 1e-8f);
        }
    }
}

// rows (n_rows, n_cols) -> x / max(||x||_2, 1e-12)   (torch.nn.functional.normalize)
__global__ void k_l2norm_rows(float *__restrict__ x, int n_rows, int n_cols) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    float *p = x + (int64_t)r * n_cols;
    float s = 0.0f;
    for (int c = 0; c < n_cols; ++c) s = fmaf(p[c], p[c], s);
    const float d = fmaxf(sqrtf(s), 1e-12f);
    for (int c = 0; c < n_cols; ++c) p[c] = p[c] / d;
}

// per-frame centroids (n_rows, T) -> nan_to_num -> adaptive_avg_pool1d(10) -> / nyquist  => (n_rows, 10)
__global__ void k_centroid_pool(const float *__restrict__ sc, int n_rows, int64_t T, float nyquist, float *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * 10) return;
    const int r = idx / 10, i = idx % 10;
    const int64_t lo = (i * T) / 10, hi = ((i + 1) * T + 9) / 10;  // floor / ceil like adaptive pooling
    float s = 0.0f;
    for (int64_t t = lo; t < hi; ++t) {
        const float v = sc[(int64_t)r * T + t];
        s += (isnan(v) || isinf(v)) ? 0.0f : v;
    }
    out[idx] = (s / (float)(hi - lo)) / nyquist;
}

// rms (n_items, C) and crest factor (n_items, C); one workgroup per item
__global__ __launch_bounds__(256) void k_rms_crest(const float *__restrict__ audio, int C, int64_t L, float *__restrict__ rms,
                                                   float *__restrict__ crest) {
    __shared__ float red[4][4];
    const int item = blockIdx.x, tid = threadIdx.x;
    const float *x0 = audio + (int64_t)item * C * L, *x1 = C == 2 ? x0 + L : x0;
    // sums of squares of x and of the "peak-normalised" x (each sample pair divided by its larger magnitude), per channel;
    // maxima of |normalised x| per channel
    float sq[2] = {0.f, 0.f}, sqn[2] = {0.f, 0.f}, mx[2] = {0.f, 0.f};
    for (int64_t i = tid; i < L; i += 256) {
        const float a = x0[i], b = x1[i];
        const float pk = fmaxf(C == 2 ? fmaxf(fabsf(a), fabsf(b)) : fabsf(a), 1e-8f);
        const float an = a / pk, bn = b / pk;
        sq[0] = fmaf(a, a, sq[0]); sq[1] = fmaf(b, b, sq[1]);
        sqn[0] = fmaf(an, an, sqn[0]); sqn[1] = fmaf(bn, bn, sqn[1]);
        mx[0] = fmaxf(mx[0], fabsf(an)); mx[1] = fmaxf(mx[1], fabsf(bn));
    }
    float vals[6] = {sq[0], sq[1], sqn[0], sqn[1], mx[0], mx[1]};
    float tot[6];
    for (int q = 0; q < 6; ++q) {
        float v = vals[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = q < 4 ? v + __shfl_xor(v, o) : fmaxf(v, __shfl_xor(v, o));
        if ((tid & 63) == 0) red[tid >> 6][q & 3] = v;
        __syncthreads();
        tot[q] = q < 4 ? ((red[0][q & 3] + red[1][q & 3]) + (red[2][q & 3] + red[3][q & 3]))
                       : fmaxf(fmaxf(red[0][q & 3], red[1][q & 3]), fmaxf(red[2][q & 3], red[3][q & 3]));
        __syncthreads();
    }
    if (tid < C) {
        const float r = sqrtf(fmaxf(tot[tid] / (float)L, 1e-8f));
        rms[item * C + tid] = r;
        const float den = fmaxf(sqrtf(fmaxf(tot[2 + tid] / (float)L, 1e-8f)), 1e-8f);
        crest[item * C + tid] = 20.0f * log10f(fmaxf(tot[4 + tid] / den, 1e-8f));
    }
}

// ---- MFCC statistics (utils.py:116-159 on torchaudio.transforms.MFCC, restated: parity unpinned) ----
// one workgroup per (item, channel): clamp, DCT per frame into LDS, then mean / unbiased std / max over frames
__global__ __launch_bounds__(256) void k_mfcc_stats(const float *__restrict__ lm, int channels, int64_t T, int M,
                                                    const float *__restrict__ dct /*(M, K)*/, int K, float top_db,
                                                    float *__restrict__ out) {
    extern __shared__ float mf[];  // [T][K]
    __shared__ float red[4];
    const int stream = blockIdx.x, item = stream / channels, ch = stream % channels, tid = threadIdx.x;
    // top_db floor: max of the item's dB mel spectrogram over its channels, frames and bands
    float mxv = -INFINITY;
    {
        const float *pi = lm + (int64_t)item * channels * T * M;
        for (int64_t i = tid; i < (int64_t)channels * T * M; i += 256) mxv = fmaxf(mxv, pi[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mxv = fmaxf(mxv, __shfl_xor(mxv, o));
        if ((tid & 63) == 0) red[tid >> 6] = mxv;
        __syncthreads();
        mxv = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    const float floor_db = mxv - top_db;
    const float *p = lm + (int64_t)stream * T * M;
    for (int64_t q = tid; q < T * K; q += 256) {
        const int64_t t = q / K;
        const int k = (int)(q % K);
        float acc = 0.0f;
        for (int m = 0; m < M; ++m) acc = fmaf(fmaxf(p[t * M + m], floor_db), dct[m * K + k], acc);
        mf[q] = acc;
    }
    __syncthreads();
    if (tid < K) {
        float s = 0.0f, mx = -INFINITY;
        for (int64_t t = 0; t < T; ++t) { const float v = mf[t * K + tid]; s += v; mx = fmaxf(mx, v); }
        const float mean = s / (float)T;
        float ss = 0.0f;
        for (int64_t t = 0; t < T; ++t) { const float d = mf[t * K + tid] - mean; ss = fmaf(d, d, ss); }
        float *o = out + ((int64_t)item * channels + ch) * 3 * K;
        o[tid] = mean;
        o[K + tid] = sqrtf(ss / (float)(T - 1));
        o[2 * K + tid] = mx;
    }
}

}  // namespace stito

using namespace stito;

extern "C" int stito_mfcc_stats(const float *logmel_dev, int n_items, int channels, int64_t n_frames, int n_mels,
                                const float *dct_dev, int n_mfcc, float top_db, float *out_dev, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(n_items > 0 && channels > 0 && n_frames > 1 && n_mels > 0, STITO_E_INVALID, "stito_mfcc_stats: empty input");
    STITO_REQUIRE(n_mfcc > 0 && n_mfcc <= 32, STITO_E_UNSUPPORTED, "n_mfcc %d not in [1, 32]", n_mfcc);
    const size_t lds = (size_t)n_frames * n_mfcc * sizeof(float);
    STITO_REQUIRE(lds <= 128 * 1024, STITO_E_UNSUPPORTED, "MFCC statistics: %lld frames do not fit the LDS", (long long)n_frames);
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)k_mfcc_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_mfcc_stats, dim3(n_items * channels), dim3(256), lds, st, logmel_dev, channels, n_frames, n_mels, dct_dev,
                       n_mfcc, top_db, out_dev);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_l2norm_rows, dim3((n_items + 63) / 64), dim3(64), 0, st, out_dev, n_items, channels * 3 * n_mfcc);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

static int feat_log2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return (1 << l) == n ? l : -1;
}

// ---- integrated loudness (features.py:267-299 -> pyloudnorm.Meter.integrated_loudness: ITU-R BS.1770-4, restated in
// st_ito/loudness.py; parity unpinned) ---------------------------------------------------------------------------------------
// (1) the reference's per-sample cross-channel normalisation (x / max_c |x|, clamp 1e-8), mono duplicated: (n, 2, L)
__global__ __launch_bounds__(256) void k_lufs_prep(const float *__restrict__ audio, int C, int64_t L, float *__restrict__ xn) {
    const int item = blockIdx.y;
    const float *x0 = audio + (int64_t)item * C * L, *x1 = C == 2 ? x0 + L : x0;
    float *o = xn + (int64_t)item * 2 * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) {
        const float a = x0[i], b = x1[i];
        const float pk = fmaxf(C == 2 ? fmaxf(fabsf(a), fabsf(b)) : fabsf(a), 1e-8f);
        o[i] = a / pk;
        o[L + i] = b / pk;
    }
}
// (3) mean square of the K-weighted signal over gating block j = [lo_j, hi_j) (the host's int() of the block edges, so that
// they are pyloudnorm's), float64 sums in a fixed order: z (n, 2, n_blocks)
__global__ __launch_bounds__(256) void k_lufs_blocks(const float *__restrict__ y, int64_t L, const int *__restrict__ lo, const int *__restrict__ hi,
                                                     int n_blocks, double inv_len, double *__restrict__ z) {
    __shared__ double red[4];
    const int j = blockIdx.x, sc = blockIdx.y, tid = threadIdx.x;
    const float *p = y + (int64_t)sc * L;
    double acc = 0.0;
    for (int i = lo[j] + tid; i < hi[j]; i += 256) acc = fma((double)p[i], (double)p[i], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) z[(int64_t)sc * n_blocks + j] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_len;
}
// (4) gating: absolute -70 LUFS, relative -10 LU below the absolutely-gated mean; one thread per item (channel gains 1, 1)
__global__ void k_lufs_gate(const double *__restrict__ z, int n_items, int n_blocks, float *__restrict__ lufs) {
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const double *z0 = z + (int64_t)item * 2 * n_blocks, *z1 = z0 + n_blocks;
    double s0 = 0.0, s1 = 0.0;
    int cnt = 0;
    for (int j = 0; j < n_blocks; ++j) {
        const double l = -0.691 + 10.0 * log10(z0[j] + z1[j]);
        if (l >= -70.0) { s0 += z0[j]; s1 += z1[j]; ++cnt; }
    }
    float out = -INFINITY;
    if (cnt > 0) {
        const double gamma_r = -0.691 + 10.0 * log10(s0 / cnt + s1 / cnt) - 10.0;
        s0 = s1 = 0.0;
        cnt = 0;
        for (int j = 0; j < n_blocks; ++j) {
            const double l = -0.691 + 10.0 * log10(z0[j] + z1[j]);
            if (l > gamma_r && l > -70.0) { s0 += z0[j]; s1 += z1[j]; ++cnt; }
        }
        if (cnt > 0) out = (float)(-0.691 + 10.0 * log10(s0 / cnt + s1 / cnt));
    }
    lufs[item] = out;
}

extern "C" size_t stito_lufs_workspace_bytes(int n_items, int64_t n_samples, int n_blocks) {
    if (n_items <= 0 || n_samples <= 0 || n_blocks <= 0) return 0;
    return align_up((size_t)n_items * 2 * n_samples * sizeof(float), 256) + align_up((size_t)n_items * 2 * n_blocks * sizeof(double), 256);
}

extern "C" int stito_lufs(const float *audio_dev, int n_items, int channels, int64_t n_samples, const double *kweight_coef_dev,
                          const int *block_lo_dev, const int *block_hi_dev, int n_blocks, double inv_block_len, float *lufs_dev,
                          void *workspace_dev, size_t workspace_bytes, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(n_items > 0 && n_samples > 0 && n_blocks > 0, STITO_E_INVALID, "stito_lufs: empty input");
    STITO_REQUIRE(channels == 1 || channels == 2, STITO_E_INVALID, "Invalid number of channels: %d", channels);
    STITO_REQUIRE(n_samples < (1ll << 31), STITO_E_UNSUPPORTED, "stito_lufs: %lld samples", (long long)n_samples);
    STITO_REQUIRE(workspace_dev != nullptr && workspace_bytes >= stito_lufs_workspace_bytes(n_items, n_samples, n_blocks), STITO_E_WORKSPACE,
                  "stito_lufs: workspace too small");
    float *xn = (float *)workspace_dev;
    double *z = (double *)((char *)workspace_dev + align_up((size_t)n_items * 2 * n_samples * sizeof(float), 256));
    const int gx = (int)((n_samples + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(k_lufs_prep, dim3(gx < 1 ? 1 : (gx > 1024 ? 1024 : gx), n_items), dim3(256), 0, st, audio_dev, channels, n_samples, xn);
    STITO_LAUNCH_CHECK();
    InView in{xn, 2 * n_samples, n_samples, 2};
    const int rc = eq_cascade(in, xn, n_items, 2, n_samples, kweight_coef_dev, st);  // in place, like the effect chain's EQ
    if (rc) return rc;
    hipLaunchKernelGGL(k_lufs_blocks, dim3(n_blocks, n_items * 2), dim3(256), 0, st, (const float *)xn, n_samples, block_lo_dev, block_hi_dev,
                       n_blocks, inv_block_len, z);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lufs_gate, dim3((n_items + 63) / 64), dim3(64), 0, st, (const double *)z, n_items, n_blocks, lufs_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_rms_crest(const float *audio_dev, int n_items, int channels, int64_t n_samples, float *rms_dev,
                               float *crest_dev, void *stream) {
    STITO_REQUIRE(n_items > 0 && n_samples > 0, STITO_E_INVALID, "stito_rms_crest: empty input");
    STITO_REQUIRE(channels == 1 || channels == 2, STITO_E_INVALID, "Invalid number of channels: %d", channels);
    hipLaunchKernelGGL(k_rms_crest, dim3(n_items), dim3(256), 0, (hipStream_t)stream, audio_dev, channels, n_samples, rms_dev, crest_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_barkspectrum(const float *audio_dev, int n_items, int channels, int64_t n_samples, int mode, int fft_size,
                                  const float *twiddle_dev, const float *fb_dev, int n_bands, float *out_dev, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int l2 = feat_log2(fft_size);
    STITO_REQUIRE(l2 >= 7 && l2 <= 15, STITO_E_UNSUPPORTED, "bark spectrum: fft_size %d must be a power of two in [128, 32768]", fft_size);
    STITO_REQUIRE(n_items > 0 && n_bands > 0, STITO_E_INVALID, "stito_barkspectrum: empty input");
    STITO_REQUIRE(channels == 2 || (channels == 1 && mode == FEAT_MONO), STITO_E_INVALID, "mode %d needs a stereo input", mode);
    STITO_REQUIRE(mode >= 0 && mode <= 2, STITO_E_INVALID, "Invalid mode %d", mode);
    STITO_REQUIRE(n_samples > fft_size / 2, STITO_E_INVALID, "reflect padding needs n_samples > fft_size/2");
    const int n_sig = mode == FEAT_MONO ? 1 : 2, hop = fft_size / 4;
    const int64_t T = n_samples / hop + 1;
    const size_t lds = (size_t)(fft_size / 2 + 8) * sizeof(float2);
    if (fft_size >= 4096) {
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)k_stft_feature<1024, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_stft_feature<1024, 0>), dim3(n_items * n_sig), dim3(1024), lds, st, audio_dev, channels, n_samples, mode,
                           n_sig, l2 - 1, hop, T, nullptr, (const float2 *)twiddle_dev, fb_dev, n_bands, 0.0f, out_dev);
    } else {
        hipLaunchKernelGGL((k_stft_feature<256, 0>), dim3(n_items * n_sig), dim3(256), lds, st, audio_dev, channels, n_samples, mode,
                           n_sig, l2 - 1, hop, T, nullptr, (const float2 *)twiddle_dev, fb_dev, n_bands, 0.0f, out_dev);
    }
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_l2norm_rows, dim3((n_items + 63) / 64), dim3(64), 0, st, out_dev, n_items, n_sig * n_bands);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" size_t stito_spectral_centroid_workspace_bytes(int n_items, int channels, int64_t n_samples) {
    return (size_t)n_items * channels * (n_samples / 1024 + 1) * sizeof(float) + 256;
}

extern "C" int stito_spectral_centroid(const float *audio_dev, int n_items, int channels, int64_t n_samples, double sample_rate,
                                       const float *window_dev, const float *twiddle_dev, float *out_dev, void *workspace_dev,
                                       size_t workspace_bytes, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(n_items > 0, STITO_E_INVALID, "stito_spectral_centroid: empty input");
    STITO_REQUIRE(channels == 1 || channels == 2, STITO_E_INVALID, "Invalid number of channels: %d", channels);
    STITO_REQUIRE(n_samples > 1024, STITO_E_INVALID, "reflect padding needs n_samples > n_fft/2");
    STITO_REQUIRE(workspace_bytes >= stito_spectral_centroid_workspace_bytes(n_items, channels, n_samples), STITO_E_WORKSPACE,
                  "centroid workspace too small");
    const int64_t T = n_samples / 1024 + 1;
    float *sc = (float *)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    const int nyq = (int)sample_rate / 2;  // torch.linspace(0, sample_rate // 2, 1025)
    const size_t lds = (size_t)(1024 + 8) * sizeof(float2);
    // channels are independent signals: (item, channel) -> blockIdx; C passed as 1 stream layout of `channels` signals
    hipLaunchKernelGGL((k_stft_feature<256, 1>), dim3(n_items * channels), dim3(256), lds, st, audio_dev, channels, n_samples, FEAT_STEREO,
                       channels, 10, 1024, T, window_dev, (const float2 *)twiddle_dev, nullptr, 0, (float)nyq / 1024.0f, sc);
    STITO_LAUNCH_CHECK();
    const int n = n_items * channels * 10;
    hipLaunchKernelGGL(k_centroid_pool, dim3((n + 255) / 256), dim3(256), 0, st, sc, n_items * channels, T, (float)(sample_rate / 2.0), out_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}
