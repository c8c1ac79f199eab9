// frontend.hip -- peak-normalise + mid/side + STFT power + log-mel + input norm, fused.
//
// Replaces (reference file:line):
//   x[b] /= x[b].abs().max().clamp(1e-8)                     st_ito/utils.py:473-474
//   mid = (L+R)/2, side = (L-R)/2, view(B*C, L)              st_ito/models/panns.py:216-227
//   torchlibrosa Spectrogram (reflect pad, Hann, |.|^2)      panns.py:147-155, 230
//   torchlibrosa LogmelFilterBank (melW, 10 log10 clamp)     panns.py:158-168, 231
//   input_norm {batchnorm, minmax, none}                     panns.py:233-245
//
// torchlibrosa evaluates the STFT as two conv1d's with a dense (1025 x 2048) windowed DFT
// matrix (3.9 GFLOP per 10 s stream); here each frame is one real FFT: the 2048 real samples
// are packed as 1024 complex points, transformed by a radix-2 Stockham FFT in LDS and
// unpacked to the 1025 one-sided bins.  The mel projection uses the band structure of melW
// (each band is one run of consecutive bins).  One workgroup = one (candidate, frame), both
// mid and side streams; HBM traffic is the audio (read ~2x because hop = n_fft/2; the second
// read hits L2) plus the (T, n_mels) output.
#include "common.h"

namespace stito {

struct FrontendDev {
    int n_fft, hop, n_mels, norm_mode, log2_n2, no_center, mel_stride;
    const float *window;
    const float2 *twiddle;  // exp(-2 pi i k / n_fft), k < n_fft/2
    const int *mel_start, *mel_len, *mel_off;
    const float *mel_w;
    const float *bn_scale, *bn_shift;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ int64_t reflect_idx(int64_t i, int64_t L) {
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    return i;
}

static constexpr int FE_THREADS = 256;

__global__ __launch_bounds__(FE_THREADS) void k_logmel(FrontendDev fe, const float *__restrict__ audio,
                                                        const float *__restrict__ peaks, int norm_passes, int C,
                                                        int64_t L, int64_t T, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int N = fe.n_fft, N2 = N >> 1;
    float2 *bufA = (float2 *)smem_raw;           // [C][N2]
    float2 *bufB = bufA + C * N2;                // [C][N2]; after the FFT the idle one of the two holds the power spectrum

    const int cand = blockIdx.y;
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;

    // ---- normalisation divisors (process_audio's and get_param_embeds' peak norms) -----------
    float d1 = 1.0f, d2 = 1.0f;
    if (peaks != nullptr && norm_passes > 0) {
        const float pk = peaks[cand];
        d1 = fmaxf(pk, 1e-8f);
        if (norm_passes > 1) d2 = fmaxf(pk / d1, 1e-8f);  // == 1 unless the candidate is (near) silent
    }
    const bool second = norm_passes > 1 && d2 != 1.0f;    // x / 1.0f == x
    // x / d as x * (1 / d): one IEEE division per thread instead of eight per sample pair (the ~10-instruction division
    // was a quarter of this kernel's vector work); <= 1 ulp from the reference's division, 1e-7 against a 2e-5 bar
    const float r1 = 1.0f / d1, r2 = 1.0f / d2;
    const float *xl = audio + (int64_t)cand * C * L;
    const float *xr = xl + L;
    // center=True: frame t covers [t*hop - n_fft/2, t*hop + n_fft/2) with reflection at the ends;
    // no_center: [t*hop, t*hop + n_fft), always inside the signal
    const int64_t base = fe.no_center ? t * fe.hop : t * fe.hop - N2;

    // ---- load, normalise, mid/side, window, pack even/odd samples as complex ------------------
    // Interior frames (every frame but the reflected ends) fetch four samples per 16-byte load when the frame start is
    // 16-byte aligned (hop and stream offsets multiples of 4), sample pairs per 8-byte load when it is 8-byte aligned: a
    // quarter / half of the load instructions and index arithmetic of the general path (by ablation this stage was 1.0 of
    // the kernel's 2.3 ms).  Same values, same operations.
    const bool inside = base >= 0 && base + N <= L;
    const uintptr_t al = (uintptr_t)(xl + base) | (uintptr_t)(xr + base) | (uintptr_t)fe.window;
    if (inside && (al & 15) == 0 && (N2 & 1) == 0) {  // four samples = two packed points per 16-byte load
        const float4 *l4 = (const float4 *)(xl + base), *r4 = (const float4 *)(xr + base), *w4 = (const float4 *)fe.window;
        for (int j = tid; j < (N2 >> 1); j += FE_THREADS) {
            const float4 a = l4[j], w = w4[j];
            float a0 = a.x * r1, a1 = a.y * r1, a2 = a.z * r1, a3 = a.w * r1;
            if (second) { a0 = a0 * r2; a1 = a1 * r2; a2 = a2 * r2; a3 = a3 * r2; }
            if (C == 2) {
                const float4 b = r4[j];
                float b0 = b.x * r1, b1 = b.y * r1, b2 = b.z * r1, b3 = b.w * r1;
                if (second) { b0 = b0 * r2; b1 = b1 * r2; b2 = b2 * r2; b3 = b3 * r2; }
                const float m0 = (a0 + b0) * 0.5f, m1 = (a1 + b1) * 0.5f, m2 = (a2 + b2) * 0.5f, m3 = (a3 + b3) * 0.5f;  // / 2 is exact
                const float s0 = (a0 - b0) * 0.5f, s1 = (a1 - b1) * 0.5f, s2 = (a2 - b2) * 0.5f, s3 = (a3 - b3) * 0.5f;
                *(float4 *)(bufA + 2 * j) = make_float4(m0 * w.x, m1 * w.y, m2 * w.z, m3 * w.w);
                *(float4 *)(bufA + N2 + 2 * j) = make_float4(s0 * w.x, s1 * w.y, s2 * w.z, s3 * w.w);
            } else {
                *(float4 *)(bufA + 2 * j) = make_float4(a0 * w.x, a1 * w.y, a2 * w.z, a3 * w.w);
            }
        }
    } else {
    const bool fast = inside && (al & 7) == 0;
    for (int m = tid; m < N2; m += FE_THREADS) {
        float a0, a1, b0 = 0.0f, b1 = 0.0f, w0, w1;
        if (fast) {
            const float2 a = ((const float2 *)(xl + base))[m], w = ((const float2 *)fe.window)[m];
            a0 = a.x; a1 = a.y; w0 = w.x; w1 = w.y;
            if (C == 2) { const float2 b = ((const float2 *)(xr + base))[m]; b0 = b.x; b1 = b.y; }
        } else {
            const int64_t i0 = reflect_idx(base + 2 * m, L), i1 = reflect_idx(base + 2 * m + 1, L);
            w0 = fe.window[2 * m]; w1 = fe.window[2 * m + 1];
            a0 = xl[i0]; a1 = xl[i1];
            if (C == 2) { b0 = xr[i0]; b1 = xr[i1]; }
        }
        a0 = a0 * r1; a1 = a1 * r1;
        if (second) { a0 = a0 * r2; a1 = a1 * r2; }
        if (C == 2) {
            b0 = b0 * r1; b1 = b1 * r1;
            if (second) { b0 = b0 * r2; b1 = b1 * r2; }
            const float m0 = (a0 + b0) * 0.5f, m1 = (a1 + b1) * 0.5f, s0 = (a0 - b0) * 0.5f, s1 = (a1 - b1) * 0.5f;  // / 2 is exact
            bufA[m] = make_float2(m0 * w0, m1 * w1);
            bufA[N2 + m] = make_float2(s0 * w0, s1 * w1);
        } else {
            bufA[m] = make_float2(a0 * w0, a1 * w1);
        }
    }
    }
    __syncthreads();

    // ---- Stockham autosort FFT of N2 complex points per stream: radix-4 stages (half the LDS round trips and
    // barriers of radix-2), plus one radix-2 stage first when log2(N2) is odd ----------------------------------
    float2 *src = bufA, *dst = bufB;
    int p = 1, sh = fe.log2_n2;  // stage twiddles exp(-2 pi i k / (r p)) are table entries: table[t] = exp(-2 pi i t / N)
    if (fe.log2_n2 & 1) {        // radix-2, p = 1: all twiddles are 1
        const int half = N2 >> 1;
        for (int c = 0; c < C; ++c) {
            const float2 *s = src + c * N2;
            float2 *d = dst + c * N2;
            for (int i = tid; i < half; i += FE_THREADS) {
                const float2 u0 = s[i], u1 = s[i + half];
                d[2 * i] = make_float2(u0.x + u1.x, u0.y + u1.y);
                d[2 * i + 1] = make_float2(u0.x - u1.x, u0.y - u1.y);
            }
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
        p = 2; --sh;
    }
    const int quarter = N2 >> 2;
    for (; p < N2; p <<= 2, sh -= 2) {
        // butterfly i: inputs s[i + q N2/4], twiddles w^q with w = exp(-2 pi i k / (4p)) = table[k << (sh - 1)],
        // outputs d[4 (i - k) + k + q p]
        for (int i = tid; i < quarter; i += FE_THREADS) {
            const int k = i & (p - 1);
            const int j = ((i - k) << 2) + k;
            const int ti = k << (sh - 1);
            const float2 w1 = fe.twiddle[ti], w2 = fe.twiddle[2 * ti];  // shared by the mid and side streams
            const float2 w3 = cmul(w1, w2);
            for (int c = 0; c < C; ++c) {
                const float2 *s = src + c * N2;
                float2 *d = dst + c * N2;
                const float2 a0 = s[i];
                const float2 a1 = cmul(s[i + quarter], w1);
                const float2 a2 = cmul(s[i + 2 * quarter], w2);
                const float2 a3 = cmul(s[i + 3 * quarter], w3);
                const float2 b0 = make_float2(a0.x + a2.x, a0.y + a2.y), b1 = make_float2(a0.x - a2.x, a0.y - a2.y);
                const float2 b2 = make_float2(a1.x + a3.x, a1.y + a3.y), b3 = make_float2(a1.x - a3.x, a1.y - a3.y);
                d[j] = make_float2(b0.x + b2.x, b0.y + b2.y);
                d[j + p] = make_float2(b1.x + b3.y, b1.y - b3.x);       // b1 - i b3
                d[j + 2 * p] = make_float2(b0.x - b2.x, b0.y - b2.y);
                d[j + 3 * p] = make_float2(b1.x - b3.y, b1.y + b3.x);   // b1 + i b3
            }
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
    }

    // ---- unpack the real transform: X[k] = E[k] + W^k O[k], power spectrum ---------------------
    float *pw = (float *)dst;  // [C][N2 + 1] in the FFT's idle buffer (C (N2 + 1) floats <= C N2 float2): 32 KB of LDS per
                               // workgroup instead of 41, one more workgroup per CU
    for (int c = 0; c < C; ++c) {
        const float2 *Z = src + c * N2;
        for (int k = tid; k <= N2; k += FE_THREADS) {
            const float2 zk = Z[k & (N2 - 1)];
            const float2 zn = Z[(N2 - k) & (N2 - 1)];
            const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            const float2 O = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));  // (zk - conj(zn)) / (2i)
            const float2 w = (k < N2) ? fe.twiddle[k] : make_float2(-1.0f, 0.0f);
            const float2 wo = cmul(w, O);
            const float re = E.x + wo.x, im = E.y + wo.y;
            pw[c * (N2 + 1) + k] = re * re + im * im;
        }
    }
    __syncthreads();

    // ---- mel bands, 10 log10(clamp(., 1e-10)), input norm ---------------------------------------
    // four lanes per (stream, band), each summing a quarter of the band's run of bins, joined in a fixed order (quarter 0 +
    // 1, 2 + 3, then the two halves): the widest bands are ~60 bins and with one lane per band the waves that hold them
    // kept the whole workgroup waiting -- 0.7 ms of the kernel's 2.0 by ablation
    const int M = fe.n_mels;
    for (int q0 = 0; q0 < 4 * C * M; q0 += FE_THREADS) {
        const int q = q0 + tid;
        const bool live = q < 4 * C * M;    // C * M * 4 is a multiple of 4: a quad of lanes is live or not as a whole
        const int part = q & 3, cm = live ? q >> 2 : 0;
        const int c = cm / M, m = cm - c * M;
        const int st = fe.mel_start[m], ln = fe.mel_len[m];
        const int lq = (ln + 3) >> 2;
        const int beg = part * lq < ln ? part * lq : ln, end = beg + lq < ln ? beg + lq : ln;
        // weights at w[i * mel_stride]: with the interleaved table (stride = n_mels, offset = m) neighbouring bands read
        // neighbouring floats per step; with packed runs (stride 1) every lane is on its own cache line
        const float *w = fe.mel_w + fe.mel_off[m];
        const int ws = fe.mel_stride;
        const float *p = pw + c * (N2 + 1) + st;
        float acc = 0.0f;
        for (int i = beg; i < end; ++i) acc = fmaf(p[i], w[i * ws], acc);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (!live || part) continue;
        float v = 10.0f * log10f(fmaxf(acc, 1e-10f));
        if (fe.norm_mode == STITO_NORM_MINMAX) {
            v = fminf(fmaxf(v, -80.0f), 40.0f);
            v = (v + 80.0f) / 120.0f;
            v = (v * 2.0f) - 1.0f;
        } else if (fe.norm_mode == STITO_NORM_BATCHNORM) {
            v = v * fe.bn_scale[m] + fe.bn_shift[m];
        }
        out[((int64_t)(cand * C + c) * T + t) * M + m] = v;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The AFx-Rep front end (n_fft = 2048, hop = 1024) without workgroup barriers: one WAVE per frame.
// k_logmel above spends its time waiting (eight __syncthreads per frame around one butterfly per thread, every sample loaded
// twice).  Here a wave owns FW_F consecutive frames of a candidate: a hop of 1 024 samples is loaded once (lane l holds the
// sample pairs 2 l + 128 j of both channels, already normalised and turned into mid / side) and serves as the second half
// of one frame and the first half of the next; the 1 024-point complex FFT of a stream is 16 x 16 x 4 in registers --
//   z[l + 64 j] --16-point DFT over j--> y[l][k1], x w1024^(l k1) --LDS transpose--> lane (k1, a): l = a + 4 b
//   --16-point DFT over b--> u[a][k1][k2a], x w64^(a k2a) --LDS transpose--> lane l'': 4-point DFT over a
//   --> X[k1 + 16 k2a + 256 k2b] = X[l'' + 64 m], m = 0..15: sixteen bins per lane, in order --
// the conjugate partner X[1024 - k] of the real-transform unpack sits in lane 64 - l (one cross-lane read per bin), the power
// spectrum goes to the wave's LDS buffer and the mel / log / norm stage is the one of k_logmel (same order of sums).  Nothing
// is shared between waves: LDS operations of one wave execute in order, so no barrier is needed anywhere.
// ------------------------------------------------------------------------------------------------------------------------
#ifndef FW_ABL
#define FW_ABL 0  // timing experiment (1 no mel stage, 2 no FFT passes, 4 no unpack); 0 in every build that ships
#endif
#ifndef FW_FRAMES
#define FW_FRAMES 4
#endif
static constexpr int FW_F = FW_FRAMES;               // consecutive frames per wave
static constexpr int FW_ROW = 66;                    // float2 per row of the first transpose (64 + 2: lanes (k1, a) of a half-wave on 32 distinct bank pairs)
static constexpr int FW_WAVE_F2 = 16 * FW_ROW + 64;  // float2 per wave: transpose buffer (>= 1 024 + 1 floats of power spectrum), w64 twiddles
static constexpr int FW_MAX_MELS = 256;
static constexpr int FW_LDS_BYTES = (16 * 64 + (FE_THREADS / 64) * FW_WAVE_F2) * 8 + 4 * FW_MAX_MELS * 8;  // + pass-A twiddles and mel tasks, shared

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// forward radix-4 butterfly in place: a_k <- sum_n a_n (-i)^(n k)
__device__ __forceinline__ void fw_r4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
    const float2 b0 = cadd(a0, a2), b1 = csub(a0, a2), b2 = cadd(a1, a3), b3 = csub(a1, a3);
    a0 = cadd(b0, b2);
    a1 = make_float2(b1.x + b3.y, b1.y - b3.x);  // b1 - i b3
    a2 = csub(b0, b2);
    a3 = make_float2(b1.x - b3.y, b1.y + b3.x);  // b1 + i b3
}
// forward 16-point DFT in place; X[k] ends up in x[4 (k & 3) + (k >> 2)] (FW_K)
#define FW_K(k) (4 * ((k) & 3) + ((k) >> 2))
__device__ __forceinline__ void fw_dft16(float2 (&x)[16]) {
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) fw_r4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);  // A[n2][k1] in x[4 k1 + n2]
    // x[4 k1 + n2] *= w16^(n2 k1)
    x[5] = cmul(x[5], make_float2(c1, -s1));     // 1
    x[6] = cmul(x[6], make_float2(h, -h));       // 2
    x[7] = cmul(x[7], make_float2(s1, -c1));     // 3
    x[9] = cmul(x[9], make_float2(h, -h));       // 2
    x[10] = make_float2(x[10].y, -x[10].x);      // 4: -i
    x[11] = cmul(x[11], make_float2(-h, -h));    // 6
    x[13] = cmul(x[13], make_float2(s1, -c1));   // 3
    x[14] = cmul(x[14], make_float2(-h, -h));    // 6
    x[15] = cmul(x[15], make_float2(-c1, s1));   // 9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) fw_r4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);  // X[k1 + 4 k2] in x[4 k1 + k2]
}
// exp(-2 pi i t / 2048) for t < 2048 from the half table
__device__ __forceinline__ float2 fw_tw(const float2 *__restrict__ table, int t) {
    const float2 v = table[t & 1023];
    return t >= 1024 ? make_float2(-v.x, -v.y) : v;
}
#define FW_LDS_SYNC() { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

#ifndef FW_OCC
#define FW_OCC 3
#endif
__global__ __launch_bounds__(FE_THREADS, FW_OCC) void k_logmel_wave(FrontendDev fe, const float *__restrict__ audio,
                                                             const float *__restrict__ peaks, int norm_passes, int C,
                                                             int64_t L, int64_t T, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N2 = 1024, HOP = 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *tw1s = (float2 *)smem_raw;                                   // [k1][l]: w1024^(l k1), shared by the waves
    float2 *buf = tw1s + 16 * 64 + wave * FW_WAVE_F2;
    float2 *tw2s = buf + 16 * FW_ROW;                                   // [a][k2a]: w64^(a k2a)
    for (int i = threadIdx.x; i < 16 * 64; i += FE_THREADS) tw1s[i] = fw_tw(fe.twiddle, 2 * (i & 63) * (i >> 6));
    // mel tasks (band, quarter of the band's run of bins), the same for every frame and stream: (first bin | bins << 16,
    // offset of the first weight)
    uint2 *mtask = (uint2 *)(tw1s + 16 * 64 + (FE_THREADS / 64) * FW_WAVE_F2);
    const int M = fe.n_mels;
    for (int q = threadIdx.x; q < 4 * M; q += FE_THREADS) {
        const int part = q & 3, m = q >> 2;
        const int st = fe.mel_start[m], ln = fe.mel_len[m];
        const int lq = (ln + 3) >> 2;
        const int beg = part * lq < ln ? part * lq : ln, end = beg + lq < ln ? beg + lq : ln;
        mtask[q] = make_uint2((unsigned)(st + beg) | (unsigned)(end - beg) << 16, (unsigned)(fe.mel_off[m] + beg * fe.mel_stride));
    }
    __syncthreads();  // the only one: tables that are the same for every wave, frame and stream
    const int cand = blockIdx.y;
    const int64_t t0 = ((int64_t)blockIdx.x * (FE_THREADS / 64) + wave) * FW_F;
    if (t0 >= T) return;
    const int nf = (int)(T - t0 < FW_F ? T - t0 : FW_F);

    float d1 = 1.0f, d2 = 1.0f;
    if (peaks != nullptr && norm_passes > 0) {
        const float pk = peaks[cand];
        d1 = fmaxf(pk, 1e-8f);
        if (norm_passes > 1) d2 = fmaxf(pk / d1, 1e-8f);
    }
    const bool second = norm_passes > 1 && d2 != 1.0f;
    const float r1 = 1.0f / d1, r2 = 1.0f / d2;  // (as k_logmel: x * (1 / d))
    const float *xl = audio + (int64_t)cand * C * L;
    const float *xr = xl + L;
    const bool al8 = (((uintptr_t)xl | (uintptr_t)xr) & 7) == 0;

    tw2s[lane] = fw_tw(fe.twiddle, 32 * (lane >> 4) * (lane & 15));  // w64^(a k2a)

    // hop h = samples [1024 h, 1024 h + 1024) (reflected outside the signal), lane l: pairs 2 l + 128 j, j < 8 -> mid / side
    float2 hm0[8], hs0[8], hm1[8], hs1[8];
    auto load_hop = [&](int64_t h, float2 (&hm)[8], float2 (&hs)[8]) {
        const int64_t b = h * HOP;
        const bool fast = al8 && b >= 0 && b + HOP <= L;
        // interior hop: all sixteen 8-byte loads are issued before the first one is used -- one memory round trip per hop instead
        // of eight (with the loads inside the loop below, behind the uniform test, hipcc waits for each pair before it issues the next)
        if (fast) {
            const float *pl = xl + b + 2 * lane, *pr = xr + b + 2 * lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) hm[j] = *(const float2 *)(pl + 128 * j);
            if (C == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) hs[j] = *(const float2 *)(pr + 128 * j);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t i = b + 2 * lane + 128 * j;
            float a0, a1, b0 = 0.0f, b1 = 0.0f;
            if (fast) {
                a0 = hm[j].x; a1 = hm[j].y;
                if (C == 2) { b0 = hs[j].x; b1 = hs[j].y; }
            } else {
                const int64_t i0 = reflect_idx(i, L), i1 = reflect_idx(i + 1, L);
                a0 = xl[i0]; a1 = xl[i1];
                if (C == 2) { b0 = xr[i0]; b1 = xr[i1]; }
            }
            a0 = a0 * r1; a1 = a1 * r1;
            if (second) { a0 = a0 * r2; a1 = a1 * r2; }
            if (C == 2) {
                b0 = b0 * r1; b1 = b1 * r1;
                if (second) { b0 = b0 * r2; b1 = b1 * r2; }
                hm[j] = make_float2((a0 + b0) * 0.5f, (a1 + b1) * 0.5f);
                hs[j] = make_float2((a0 - b0) * 0.5f, (a1 - b1) * 0.5f);
            } else {
                hm[j] = make_float2(a0, a1);
                hs[j] = hm[j];
            }
        }
    };
    const int64_t h_first = fe.no_center ? t0 : t0 - 1;  // frame t covers hops (t - 1, t) when centred, (t, t + 1) otherwise
    load_hop(h_first, hm0, hs0);
    load_hop(h_first + 1, hm1, hs1);

    for (int f = 0; f < nf; ++f) {
        const int64_t t = t0 + f;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            // ---- window, pass A: 16-point DFT over j of z[l + 64 j] ------------------------------------------------
            float2 x[16];
            // (the window and the unpack twiddles are the same for every frame: reloaded per stream from L1 / L2 through
            // pointers the compiler cannot see through, or it keeps all 64 values in registers across the loops and spills)
            const float *win = fe.window;
            const float2 *twt = fe.twiddle;
            asm volatile("" : "+s"(win), "+s"(twt));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 w = *(const float2 *)(win + 2 * lane + 128 * j);
                const float2 v = j < 8 ? (c == 0 ? hm0[j] : hs0[j]) : (c == 0 ? hm1[j - 8] : hs1[j - 8]);
                x[j] = make_float2(v.x * w.x, v.y * w.y);
            }
            if (!(FW_ABL & 2)) {
            fw_dft16(x);
            buf[lane] = x[FW_K(0)];
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) buf[k1 * FW_ROW + lane] = cmul(x[FW_K(k1)], tw1s[k1 * 64 + lane]);
            FW_LDS_SYNC()
            // ---- pass B: lane (k1, a), 16-point DFT over b of y[a + 4 b][k1] ------------------------------------------
            {
                const int k1 = lane & 15, a = lane >> 4;
#pragma unroll
                for (int b = 0; b < 16; ++b) x[b] = buf[k1 * FW_ROW + a + 4 * b];
                fw_dft16(x);
                FW_LDS_SYNC()  // (all reads of the first layout are back before the second one is written)
                buf[(k1 + 0) * 4 + a] = x[FW_K(0)];
#pragma unroll
                for (int k2a = 1; k2a < 16; ++k2a) buf[(k1 + 16 * k2a) * 4 + a] = cmul(x[FW_K(k2a)], tw2s[a * 16 + k2a]);
            }
            FW_LDS_SYNC()
            // ---- pass C: 4-point DFT over a; x[q + 4 k2b] = X[lane + 64 (q + 4 k2b)] -----------------------------------
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 p01 = *(const float4 *)(buf + (lane + 64 * q) * 4), p23 = *(const float4 *)(buf + (lane + 64 * q) * 4 + 2);
                float2 a0 = make_float2(p01.x, p01.y), a1 = make_float2(p01.z, p01.w), a2 = make_float2(p23.x, p23.y), a3 = make_float2(p23.z, p23.w);
                fw_r4(a0, a1, a2, a3);
                x[q] = a0; x[q + 4] = a1; x[q + 8] = a2; x[q + 12] = a3;
            }
            FW_LDS_SYNC()
            }
            // ---- unpack the real transform (as k_logmel), power spectrum into the wave's buffer ---------------------------
            float *pw = (float *)buf;
            const int pl = (64 - lane) & 63;
            if (FW_ABL & 4) { for (int m = 0; m < 16; ++m) pw[lane + 64 * m] = x[m].x; }
            else
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int k = lane + 64 * m;
                const float2 zk = x[m];
                const float px = __shfl(x[15 - m].x, pl, 64), py = __shfl(x[15 - m].y, pl, 64);
                const float2 zo = x[(16 - m) & 15];  // lane 0: X[1024 - 64 m] is its own bin 16 - m (X[1024] = X[0])
                const float2 zn = lane == 0 ? zo : make_float2(px, py);
                const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
                const float2 O = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
                const float2 wo = cmul(twt[k], O);
                const float re = E.x + wo.x, im = E.y + wo.y;
                pw[k] = re * re + im * im;
            }
            if (lane == 0) {  // k = 1024: zk = zn = X[0], w = -1
                const float2 z = x[0];
                const float2 E = make_float2(0.5f * (z.x + z.x), 0.5f * (z.y - z.y));
                const float2 O = make_float2(0.5f * (z.y + z.y), -0.5f * (z.x - z.x));
                const float2 wo = cmul(make_float2(-1.0f, 0.0f), O);
                const float re = E.x + wo.x, im = E.y + wo.y;
                pw[N2] = re * re + im * im;
            }
            FW_LDS_SYNC()
            // ---- mel bands, 10 log10(clamp(., 1e-10)), input norm: four lanes per band (k_logmel's order of sums) ---------
            if (FW_ABL & 1) { if (lane < M) out[((int64_t)(cand * C + c) * T + t) * M + lane] = pw[lane * 8]; if (lane + 64 < M) out[((int64_t)(cand * C + c) * T + t) * M + lane + 64] = pw[lane * 8 + 4]; }
            else
            for (int q0 = 0; q0 < 4 * M; q0 += 64) {
                const int q = q0 + lane;
                const bool live = q < 4 * M;
                const int part = q & 3, m = live ? q >> 2 : 0;
                const uint2 mt = live ? mtask[q] : make_uint2(0u, 0u);
                const float *w = fe.mel_w + mt.y;
                const int ws = fe.mel_stride, n = (int)(mt.x >> 16);
                const float *p = pw + (mt.x & 0xffffu);
                float acc = 0.0f;
                for (int i = 0; i < n; ++i) acc = fmaf(p[i], w[i * ws], acc);
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                if (!live || part) continue;
                float v = 10.0f * log10f(fmaxf(acc, 1e-10f));
                if (fe.norm_mode == STITO_NORM_MINMAX) {
                    v = fminf(fmaxf(v, -80.0f), 40.0f);
                    v = (v + 80.0f) / 120.0f;
                    v = (v * 2.0f) - 1.0f;
                } else if (fe.norm_mode == STITO_NORM_BATCHNORM) {
                    v = v * fe.bn_scale[m] + fe.bn_shift[m];
                }
                out[((int64_t)(cand * C + c) * T + t) * M + m] = v;
            }
            FW_LDS_SYNC()  // the power spectrum has been read before the next stream's pass A overwrites it
        }
        if (f + 1 < nf) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { hm0[j] = hm1[j]; hs0[j] = hs1[j]; }
            load_hop(h_first + f + 2, hm1, hs1);
        }
    }
}

}  // namespace stito

using namespace stito;

extern "C" int64_t stito_num_frames(int64_t n_samples, int hop) { return n_samples / hop + 1; }
extern "C" int64_t stito_num_frames_nocenter(int64_t n_samples, int n_fft, int hop) {
    return n_samples >= n_fft ? (n_samples - n_fft) / hop + 1 : 0;
}

extern "C" int stito_logmel(const stito_frontend *fe, const float *audio_dev, const float *peaks_dev, int norm_passes,
                            int pop, int channels, int64_t n_samples, float *logmel_dev, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(fe != nullptr, STITO_E_INVALID, "stito_logmel: null front-end");
    STITO_REQUIRE(channels == 1 || channels == 2, STITO_E_INVALID, "Invalid number of channels: %d", channels);
    const int N = fe->n_fft;
    STITO_REQUIRE(N >= 64 && N <= 4096 && (N & (N - 1)) == 0, STITO_E_UNSUPPORTED, "n_fft %d must be a power of two in [64, 4096]", N);
    STITO_REQUIRE(fe->hop > 0 && fe->n_mels > 0, STITO_E_INVALID, "bad hop / n_mels");
    STITO_REQUIRE(fe->no_center ? n_samples >= N : n_samples > N / 2, STITO_E_INVALID,
                  "audio too short for the STFT (n_samples %lld, n_fft %d)", (long long)n_samples, N);
    STITO_REQUIRE(pop > 0, STITO_E_INVALID, "empty batch");
    FrontendDev d;
    d.n_fft = N; d.hop = fe->hop; d.n_mels = fe->n_mels; d.norm_mode = fe->norm_mode; d.no_center = fe->no_center;
    int l2 = 0;
    while ((1 << l2) < N / 2) ++l2;
    d.log2_n2 = l2;
    d.window = fe->window_dev; d.twiddle = (const float2 *)fe->twiddle_dev;
    d.mel_start = fe->mel_start_dev; d.mel_len = fe->mel_len_dev; d.mel_off = fe->mel_off_dev; d.mel_w = fe->mel_w_dev;
    d.mel_stride = fe->mel_w_stride > 1 ? fe->mel_w_stride : 1;
    d.bn_scale = fe->bn0_scale_dev; d.bn_shift = fe->bn0_shift_dev;
    STITO_REQUIRE(fe->norm_mode != STITO_NORM_BATCHNORM || (d.bn_scale && d.bn_shift), STITO_E_INVALID, "batchnorm input norm needs bn0 scale/shift");
    const int64_t T = fe->no_center ? stito_num_frames_nocenter(n_samples, N, fe->hop) : stito_num_frames(n_samples, fe->hop);
    // the AFx-Rep front end runs one wave per frame (k_logmel_wave); STITO_LOGMEL_GENERIC=1 keeps the general kernel (tests)
    const char *gen_env = getenv("STITO_LOGMEL_GENERIC");  // (read per call: the parity test switches inside one process)
    const bool generic_only = gen_env != nullptr && atoi(gen_env) != 0;
    if (N == 2048 && fe->hop == 1024 && !generic_only && n_samples >= 2048 && fe->n_mels <= FW_MAX_MELS) {
        const int wpb = FE_THREADS / 64;
        const int64_t tasks = (T + FW_F - 1) / FW_F;
        const size_t lds_w = FW_LDS_BYTES;
        hipLaunchKernelGGL(k_logmel_wave, dim3((unsigned)((tasks + wpb - 1) / wpb), pop), dim3(FE_THREADS), lds_w, st, d, audio_dev,
                           peaks_dev, norm_passes, channels, n_samples, T, logmel_dev);
        STITO_LAUNCH_CHECK();
        return STITO_OK;
    }
    const size_t lds = (size_t)channels * (N / 2) * sizeof(float2) * 2;
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)k_logmel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_logmel, dim3((unsigned)T, pop), dim3(FE_THREADS), lds, st, d, audio_dev, peaks_dev, norm_passes,
                       channels, n_samples, T, logmel_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}
