// cnn14.hip -- AFx-Rep (Cnn14) trunk on gfx950: 3x3 conv + BN + ReLU (+2x2 avg-pool) as an
// implicit GEMM on exact-f32 MFMA, pooling head, fc_mid / fc_side, L2-normalise + cosine loss.
//
// Replaces (reference file:line): ConvBlock.forward st_ito/models/panns.py:65-80, Cnn14.forward
// panns.py:250-281, and the embedding post-processing st_ito/utils.py:491-501 +
// st_ito/style_transfer.py:544-571.
//
// Precision: the reference runs the trunk in fp32 and the parity bar is 1e-4 relative on the
// embeddings, so the contraction uses v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bitwise an
// fmaf chain).  gfx950 has no TF32/xf32 path; the f32 MFMA peak (157.3 TFLOP/s) is the roofline.
//
// Layout: activations NHWC (stream, time, mel, channel) so that one pixel's channels are
// contiguous (im2col K-runs are contiguous, the epilogue stores 128 B per pixel per half-wave).
// Weights are pre-packed as [cin/8][tap][cout][8].
//
// Tiling: a workgroup computes BM = 64*WM output pixels x BN = 64*WN output channels; each of its
// WM*WN waves owns a 64 x 64 sub-tile = 2 x 2 MFMA 32x32 blocks (64 accumulator VGPRs).  The BM
// pixels are a TH x TW spatial patch of one stream (or of several streams when the feature map
// is smaller than the patch).  Per 8-input-channel chunk the (TH+2) x (TW+2) x 8 halo patch and
// the 9 x BN x 8 weight slab go to LDS once and are reused by all 9 taps: 9*4 = 36 k-steps of
// 4 MFMAs per wave (9216 MFMA cycles) between barriers; the next chunk's global loads are
// issued before the MFMA block and land in registers meanwhile.
// M index -> pixel mapping: 4 consecutive GEMM rows = one 2x2 pooling window, which the MFMA
// C layout leaves in 4 consecutive accumulator registers of one lane, so BN+ReLU+avg-pool
// happen in registers.
#include <string>

#include "common.h"

namespace stito {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int CK = 8;      // input channels per chunk
static constexpr int CPAD = 12;   // floats per LDS pixel / weight row (8 + 4 pad: conflict-free b128 reads)

struct ConvGeom {
    int S, H, W, Cin, Cout;
    int HS, NSLOT;            // rows per stream slot, stream slots per tile
    int n_row_tiles, n_col_tiles, n_m_tiles;
    int Ho, Wo;
};

// MODE 0: one LDS buffer, two barriers per chunk, 2 workgroups per CU cover each other's staging.
// MODE 1: two LDS buffers (128 KB, 1 workgroup per CU): the next chunk is written to the other
//         buffer in the middle of the MFMA stream, one barrier per chunk.
// MODE 2: timing ablation only (no staging after chunk 0; wrong results).
template <int WM, int WN, int TW, bool POOL, int MODE>
__global__ __launch_bounds__(64 * WM * WN) void k_conv3x3(const float *__restrict__ in, const float *__restrict__ wpk,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, float *__restrict__ out,
                                                          ConvGeom g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int TH = BM / TW;
    constexpr int GW = TW / 2;
    constexpr int PW = TW + 2;
    constexpr int B_ITEMS = (9 * BN * 2 + NT - 1) / NT;
    constexpr int A_ITEMS = (WM == 4) ? 4 : 3;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sB = smem;                       // [9][BN][CPAD]
    float *sA = smem + 9 * BN * CPAD;       // [NSLOT*(HS+2)][PW][CPAD]
    const int buf_floats = 9 * BN * CPAD + g.NSLOT * (g.HS + 2) * PW * CPAD;  // MODE 1: second buffer follows

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv % WM, wn = wv / WM;
    const int half = lane >> 5, l31 = lane & 31;

    // block -> (n tile, m tile); m fastest so that co-running blocks share one weight slab in L2
    const int m_tile = blockIdx.x % g.n_m_tiles;
    const int n_tile = blockIdx.x / g.n_m_tiles;
    const int n0 = n_tile * BN;
    const int ct = m_tile % g.n_col_tiles;
    const int rt = (m_tile / g.n_col_tiles) % g.n_row_tiles;
    const int sg = m_tile / (g.n_col_tiles * g.n_row_tiles);
    const int s_base = sg * g.NSLOT;
    const int h0 = rt * TH, w0 = ct * TW;
    const int HSP = g.HS + 2;
    const int npix = g.NSLOT * HSP * PW;

    // ---- per-thread staging descriptors (invariant over the K loop) ---------------------------
    int64_t a_goff[A_ITEMS];
    int a_loff[A_ITEMS];
    bool a_valid[A_ITEMS], a_inrange[A_ITEMS];
#pragma unroll
    for (int it = 0; it < A_ITEMS; ++it) {
        const int q = tid + it * NT;
        const int pix = q >> 1, hf = q & 1;
        a_inrange[it] = pix < npix;
        const int pcol = pix % PW;
        const int prow_all = pix / PW;
        const int slot = prow_all / HSP, prow = prow_all % HSP;
        const int s = s_base + slot, h = h0 + prow - 1, w = w0 + pcol - 1;
        a_valid[it] = a_inrange[it] && s < g.S && h >= 0 && h < g.H && w >= 0 && w < g.W;
        a_goff[it] = (((int64_t)s * g.H + h) * g.W + w) * g.Cin + hf * 4;
        a_loff[it] = pix * CPAD + hf * 4;
    }
    int64_t b_goff[B_ITEMS];
    int b_loff[B_ITEMS];
    bool b_valid[B_ITEMS];
#pragma unroll
    for (int it = 0; it < B_ITEMS; ++it) {
        const int q = tid + it * NT;
        b_valid[it] = q < 9 * BN * 2;
        const int tap = q / (2 * BN), r = q % (2 * BN);
        const int co = r >> 1, hf = r & 1;
        b_goff[it] = ((int64_t)tap * g.Cout + n0 + co) * CK + hf * 4;
        b_loff[it] = (tap * BN + co) * CPAD + hf * 4;
    }
    const int64_t b_chunk_stride = (int64_t)9 * g.Cout * CK;

    // ---- per-lane fragment addresses ----------------------------------------------------------
    int a_frag[2], b_frag[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int gi = (wm * 2 + mb) * 8 + (l31 >> 2);
        const int gr = gi / GW, gc = gi % GW;
        const int ph = 2 * gr + ((l31 >> 1) & 1), pw_ = 2 * gc + (l31 & 1);
        const int slot = ph / g.HS, prow = ph % g.HS;
        a_frag[mb] = ((slot * HSP + prow) * PW + pw_) * CPAD + half * 4;
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) b_frag[nb] = (wn * 64 + nb * 32 + l31) * CPAD + half * 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    float4 a_reg[A_ITEMS], b_reg[B_ITEMS];
    auto prefetch = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < A_ITEMS; ++it)
            a_reg[it] = a_valid[it] ? *(const float4 *)(in + a_goff[it] + chunk * CK) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int it = 0; it < B_ITEMS; ++it)
            b_reg[it] = b_valid[it] ? *(const float4 *)(wpk + b_goff[it] + chunk * b_chunk_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    const int n_chunks = g.Cin / CK;
    auto stage_write = [&](int boff) {
#pragma unroll
        for (int it = 0; it < A_ITEMS; ++it)
            if (a_inrange[it]) *(float4 *)(sA + boff + a_loff[it]) = a_reg[it];
#pragma unroll
        for (int it = 0; it < B_ITEMS; ++it)
            if (b_valid[it]) *(float4 *)(sB + boff + b_loff[it]) = b_reg[it];
    };
    auto mfma_tap = [&](int boff, int tap) {
        const int kh = tap / 3, kw = tap % 3;
        float4 av[2], bv[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) av[mb] = *(const float4 *)(sA + boff + a_frag[mb] + (kh * PW + kw) * CPAD);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bv[nb] = *(const float4 *)(sB + boff + b_frag[nb] + tap * BN * CPAD);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float a0 = kk == 0 ? av[0].x : kk == 1 ? av[0].y : kk == 2 ? av[0].z : av[0].w;
            const float a1 = kk == 0 ? av[1].x : kk == 1 ? av[1].y : kk == 2 ? av[1].z : av[1].w;
            const float b0 = kk == 0 ? bv[0].x : kk == 1 ? bv[0].y : kk == 2 ? bv[0].z : bv[0].w;
            const float b1 = kk == 0 ? bv[1].x : kk == 1 ? bv[1].y : kk == 2 ? bv[1].z : bv[1].w;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };

    prefetch(0);
    if (MODE == 1) {
        stage_write(0);
        __syncthreads();
        if (n_chunks > 1) prefetch(1);
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const int cur = (chunk & 1) * buf_floats, nxt = buf_floats - cur;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                mfma_tap(cur, tap);
                if (tap == 4 && chunk + 1 < n_chunks) stage_write(nxt);  // chunk+1 -> the idle buffer
            }
            __syncthreads();
            if (chunk + 2 < n_chunks) prefetch(chunk + 2);
        }
    } else {
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            if (MODE == 0 || chunk == 0) {
                __syncthreads();  // previous chunk's fragment reads are done
                stage_write(0);
                __syncthreads();
                if (MODE == 0 && chunk + 1 < n_chunks) prefetch(chunk + 1);  // in flight during the MFMA block
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) mfma_tap(0, tap);
        }
    }

    // ---- epilogue: BN (scale/shift) + ReLU (+ 2x2 average pool), NHWC store ----------------------
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int co = n0 + wn * 64 + nb * 32 + l31;
        const float sc = scale[co], sh = shift[co];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // accumulator rows 8q + 4*half + {0,1,2,3} = pooling window (group) 2q + half of this block
                const int gi = (wm * 2 + mb) * 8 + 2 * q + half;
                const int gr = gi / GW, gc = gi % GW;
                const int slot = (2 * gr) / g.HS, prow = (2 * gr) % g.HS;
                const int s = s_base + slot;
                const int h = h0 + prow, w = w0 + 2 * gc;
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaxf(fmaf(acc[mb][nb][4 * q + e], sc, sh), 0.0f);
                if (POOL) {
                    const int oh = h >> 1, ow = w >> 1;
                    if (s < g.S && oh < g.Ho && ow < g.Wo)
                        out[(((int64_t)s * g.Ho + oh) * g.Wo + ow) * g.Cout + co] = (((y[0] + y[1]) + y[2]) + y[3]) * 0.25f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int hh = h + (e >> 1), ww = w + (e & 1);
                        if (s < g.S && hh < g.H && ww < g.W)
                            out[(((int64_t)s * g.H + hh) * g.W + ww) * g.Cout + co] = y[e];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv_block1.conv1: one input channel (the log-mel image) -> 64 channels.  K = 9: no matrix
// shape to speak of; direct, output-bandwidth bound (writes S*T*M*64 floats).
// thread = (pixel slot, 4 output channels); 16 threads cover one pixel's 64 channels = 256 B.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_first(const float *__restrict__ in, const float *__restrict__ w /*[cout][9]*/,
                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                     float *__restrict__ out, int H, int W, int Cout, int64_t n_pix_per_stream) {
    const int s = blockIdx.y;
    const int cg = threadIdx.x % (Cout / 4), ps = threadIdx.x / (Cout / 4);
    const int pix_per_iter = 256 / (Cout / 4);
    float wr[4][9], sc[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(cg * 4 + c) * 9 + t];
        sc[c] = scale[cg * 4 + c];
        sh[c] = shift[cg * 4 + c];
    }
    const float *ip = in + (int64_t)s * n_pix_per_stream;
    float *op = out + (int64_t)s * n_pix_per_stream * Cout;
    for (int64_t p = (int64_t)blockIdx.x * pix_per_iter + ps; p < n_pix_per_stream; p += (int64_t)gridDim.x * pix_per_iter) {
        const int h = (int)(p / W), x = (int)(p % W);
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = x + t % 3 - 1;
            v[t] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? ip[(int64_t)hh * W + ww] : 0.0f;
        }
        float4 o;
        float r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = 0.0f;
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(v[t], wr[c][t], a);
            r[c] = fmaxf(fmaf(a, sc[c], sh[c]), 0.0f);
        }
        o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
        *(float4 *)(op + p * Cout + cg * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Pooling head (panns.py:262-266): mean over mel, then max over time + mean over time.
// x (S, H, W, C) -> feat (S, C)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head(const float *__restrict__ x, float *__restrict__ feat, int H, int W, int C) {
    const int s = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float *p = x + (int64_t)s * H * W * C + c;
    float mx = -INFINITY, sum = 0.0f;
    for (int h = 0; h < H; ++h) {
        float rs = 0.0f;
        for (int w = 0; w < W; ++w) rs += p[((int64_t)h * W + w) * C];
        const float m = rs / (float)W;
        mx = fmaxf(mx, m);
        sum += m;
    }
    feat[(int64_t)s * C + c] = mx + sum / (float)H;
}

// fc_mid / fc_side (panns.py:271-279).  feat (n_cand*channels, K); stream parity picks the layer.
// One workgroup = 8 streams of one kind x 256 outputs; weights are (K, E) so lanes read coalesced.
static constexpr int FC_SB = 8;
__global__ __launch_bounds__(256) void k_fc(const float *__restrict__ feat, const float *__restrict__ wt_mid,
                                             const float *__restrict__ b_mid, const float *__restrict__ wt_side,
                                             const float *__restrict__ b_side, float *__restrict__ mid,
                                             float *__restrict__ side, int n_cand, int channels, int K, int E) {
    extern __shared__ float sf[];  // [FC_SB][K]
    const int kind = blockIdx.z;   // 0 mid, 1 side
    const int c0 = blockIdx.y * FC_SB;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float *wt = kind == 0 ? wt_mid : wt_side;
    const float *bb = kind == 0 ? b_mid : b_side;
    float *dst = kind == 0 ? mid : side;
    for (int i = threadIdx.x; i < FC_SB * K; i += 256) {
        const int sb = i / K, k = i % K;
        const int cand = c0 + sb;
        sf[i] = cand < n_cand ? feat[((int64_t)cand * channels + kind) * K + k] : 0.0f;
    }
    __syncthreads();
    if (e >= E) return;
    float acc[FC_SB];
#pragma unroll
    for (int sb = 0; sb < FC_SB; ++sb) acc[sb] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float wv = wt[(int64_t)k * E + e];
#pragma unroll
        for (int sb = 0; sb < FC_SB; ++sb) acc[sb] = fmaf(sf[sb * K + k], wv, acc[sb]);
    }
    const float bias = bb[e];
#pragma unroll
    for (int sb = 0; sb < FC_SB; ++sb)
        if (c0 + sb < n_cand) dst[(int64_t)(c0 + sb) * E + e] = acc[sb] + bias;
}

__global__ void k_copy(const float *__restrict__ a, float *__restrict__ b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

// ------------------------------------------------------------------------------------------------
// Embedding post-processing: NaN scrub (utils.py:491-497), F.normalize (500-501),
// -cosine_similarity and mean over {mid, side} (style_transfer.py:544-571).
// ------------------------------------------------------------------------------------------------
__global__ void k_nan_flags(const float *__restrict__ mid, const float *__restrict__ side, int64_t n, int *flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (isnan(mid[i])) atomicOr(&flags[0], 1);
    if (isnan(side[i])) atomicOr(&flags[1], 1);
}

__device__ __forceinline__ float nan_to_num(float v) {
    if (isnan(v)) return 0.0f;
    if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return v;
}

__device__ __forceinline__ float block_sum_256(float v, float *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_embed_loss(float *__restrict__ mid, float *__restrict__ side, int E,
                                                     const float *__restrict__ tmid, const float *__restrict__ tside,
                                                     float *__restrict__ loss, const int *__restrict__ flags,
                                                     int side_is_mid) {
    __shared__ float red[4];
    const int cand = blockIdx.x;
    // reference: `if isnan(mid).any(): scrub mid  elif isnan(side).any(): scrub side`
    const bool scrub_mid = flags[0] != 0;
    const bool scrub_side = !scrub_mid && flags[1] != 0;
    float cosv[2];
    for (int kind = 0; kind < 2; ++kind) {
        float *e = (kind == 0 ? mid : side) + (int64_t)cand * E;
        const float *t = kind == 0 ? tmid : tside;
        const bool scrub = kind == 0 ? scrub_mid : scrub_side;
        float ss = 0.0f;
        for (int i = threadIdx.x; i < E; i += 256) {
            float v = e[i];
            if (scrub) v = nan_to_num(v);
            ss = fmaf(v, v, ss);
        }
        const float nrm = fmaxf(sqrtf(block_sum_256(ss, red)), 1e-12f);  // F.normalize eps
        float dot = 0.0f, s2 = 0.0f, t2 = 0.0f;
        for (int i = threadIdx.x; i < E; i += 256) {
            float v = e[i];
            if (scrub) v = nan_to_num(v);
            v = v / nrm;
            e[i] = v;
            if (t != nullptr) {
                const float tv = t[i];
                dot = fmaf(v, tv, dot); s2 = fmaf(v, v, s2); t2 = fmaf(tv, tv, t2);
            }
        }
        if (t != nullptr) {
            dot = block_sum_256(dot, red);
            s2 = block_sum_256(s2, red);
            t2 = block_sum_256(t2, red);
            // torch.cosine_similarity: x.y / (max(|x|, eps) * max(|y|, eps)), eps = 1e-8
            cosv[kind] = dot / (fmaxf(sqrtf(s2), 1e-8f) * fmaxf(sqrtf(t2), 1e-8f));
        }
        __syncthreads();
    }
    (void)side_is_mid;
    if (tmid != nullptr && threadIdx.x == 0) loss[cand] = ((-cosv[0]) + (-cosv[1])) / 2.0f;
}

// ------------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_conv(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)Cout * Cin * 9;
    if (i >= n) return;
    if (Cin % CK != 0) {  // first layer: [cout][9]
        o[i] = w[i];
        return;
    }
    // destination index i -> (chunk, tap, co, c8)
    const int c8 = (int)(i % CK);
    const int co = (int)((i / CK) % Cout);
    const int tap = (int)((i / ((int64_t)CK * Cout)) % 9);
    const int chunk = (int)(i / ((int64_t)CK * Cout * 9));
    const int ci = chunk * CK + c8;
    o[i] = w[((int64_t)co * Cin + ci) * 9 + tap];
}

__global__ void k_bn_fold(const float *g, const float *b, const float *m, const float *v, float eps, int n, float *scale,
                          float *shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (g == nullptr) { scale[i] = 1.0f; shift[i] = 0.0f; return; }
    const float sc = g[i] / sqrtf(v[i] + eps);
    scale[i] = sc;
    shift[i] = b[i] - m[i] * sc;
}

__global__ void k_transpose(const float *__restrict__ in, int rows, int cols, float *__restrict__ out) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        t[j][threadIdx.x] = (r < rows && c < cols) ? in[(int64_t)r * cols + c] : 0.0f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(int64_t)c * rows + r] = t[threadIdx.x][j];
    }
}

// ================================================================================================
// host side
// ================================================================================================
static int g_conv_mode = 0;

template <int WM, int WN, int TW, bool POOL, int MODE>
static int launch_conv_mode(const float *in, const float *wpk, const float *scale, const float *shift, float *out,
                       ConvGeom g, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN, TH = BM / TW, PW = TW + 2;
    int hs = TH, nslot = 1;
    if (g.H < TH) {
        hs = 2;
        while (hs < g.H) hs <<= 1;
        nslot = TH / hs;
    }
    g.HS = hs; g.NSLOT = nslot;
    g.n_row_tiles = nslot > 1 ? 1 : (g.H + TH - 1) / TH;
    g.n_col_tiles = (g.W + TW - 1) / TW;
    g.n_m_tiles = ((g.S + nslot - 1) / nslot) * g.n_row_tiles * g.n_col_tiles;
    g.Ho = g.H / 2; g.Wo = g.W / 2;
    const int npix = nslot * (hs + 2) * PW;
    STITO_REQUIRE(npix * 2 <= ((WM == 4) ? 4 : 3) * 64 * WM * WN, STITO_E_UNSUPPORTED, "conv tile: halo patch of %d pixels exceeds the staging budget", npix);
    const size_t lds = (size_t)(9 * BN * CPAD + npix * CPAD) * sizeof(float) * (MODE == 1 ? 2 : 1);
    STITO_REQUIRE(lds <= 160 * 1024, STITO_E_UNSUPPORTED, "conv tile needs %zu bytes of LDS", lds);
    auto kern = k_conv3x3<WM, WN, TW, POOL, MODE>;
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (int64_t)g.n_m_tiles * (g.Cout / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WM * WN), lds, st, in, wpk, scale, shift, out, g);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

template <int WM, int WN, int TW, bool POOL>
static int launch_conv(const float *in, const float *wpk, const float *scale, const float *shift, float *out,
                       const ConvGeom &g, hipStream_t st) {
    if (g_conv_mode == 1) return launch_conv_mode<WM, WN, TW, POOL, 1>(in, wpk, scale, shift, out, g, st);
    if (g_conv_mode == 2) return launch_conv_mode<WM, WN, TW, POOL, 2>(in, wpk, scale, shift, out, g, st);
    return launch_conv_mode<WM, WN, TW, POOL, 0>(in, wpk, scale, shift, out, g, st);
}

template <int WM, int WN, bool POOL>
static int launch_conv_tw(const float *in, const float *wpk, const float *scale, const float *shift, float *out,
                          const ConvGeom &g, hipStream_t st) {
    if (g.W >= 16) return launch_conv<WM, WN, 16, POOL>(in, wpk, scale, shift, out, g, st);
    if (g.W >= 8) return launch_conv<WM, WN, 8, POOL>(in, wpk, scale, shift, out, g, st);
    return launch_conv<WM, WN, 4, POOL>(in, wpk, scale, shift, out, g, st);
}

static int conv_first(const float *in, const float *w, const float *scale, const float *shift, float *out, int S, int H,
                      int W, int Cout, hipStream_t st) {
    STITO_REQUIRE(Cout % 4 == 0 && 256 % (Cout / 4) == 0, STITO_E_UNSUPPORTED, "first conv: cout %d", Cout);
    const int64_t npix = (int64_t)H * W;
    const int ppi = 256 / (Cout / 4);
    int64_t gx = (npix + ppi - 1) / ppi;
    const int64_t cap = (256 * 32 + S - 1) / S;
    gx = gx < cap ? gx : (cap < 1 ? 1 : cap);
    hipLaunchKernelGGL(k_conv_first, dim3((unsigned)gx, S), dim3(256), 0, st, in, w, scale, shift, out, H, W, Cout, npix);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

}  // namespace stito

using namespace stito;

extern "C" int stito_set_option(const char *name, int value) {
    if (name != nullptr && std::string(name) == "conv_mode") {
        STITO_REQUIRE(value >= 0 && value <= 2, STITO_E_INVALID, "conv_mode must be 0, 1 or 2");
        g_conv_mode = value;
        return STITO_OK;
    }
    STITO_REQUIRE(false, STITO_E_INVALID, "unknown option %s", name ? name : "(null)");
}

extern "C" size_t stito_cnn14_packed_conv_floats(int cout, int cin) { return (size_t)cout * cin * 9; }

extern "C" int stito_cnn14_pack_conv(const float *w_oihw_dev, int cout, int cin, float *packed_dev, void *stream) {
    const int64_t n = (int64_t)cout * cin * 9;
    hipLaunchKernelGGL(k_pack_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw_dev, cout, cin, packed_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_bn_fold(const float *gamma_dev, const float *beta_dev, const float *mean_dev, const float *var_dev,
                             double eps, int n, float *scale_dev, float *shift_dev, void *stream) {
    hipLaunchKernelGGL(k_bn_fold, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma_dev, beta_dev, mean_dev, var_dev, (float)eps, n, scale_dev, shift_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_transpose(const float *in_dev, int rows, int cols, float *out_dev, void *stream) {
    hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, (hipStream_t)stream, in_dev, rows, cols, out_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_conv3x3_bn_relu(const float *in_dev, const float *packed_w_dev, const float *scale_dev,
                                     const float *shift_dev, float *out_dev, int n, int H, int W, int cin, int cout,
                                     int pool, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(n > 0 && H > 0 && W > 0, STITO_E_INVALID, "conv: empty input");
    if (cin % CK != 0) {
        STITO_REQUIRE(cin == 1 && !pool, STITO_E_UNSUPPORTED, "conv: cin=%d (only 1 or a multiple of 8)", cin);
        return conv_first(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, n, H, W, cout, st);
    }
    STITO_REQUIRE(cout % 64 == 0, STITO_E_UNSUPPORTED, "conv: cout=%d must be a multiple of 64", cout);
    STITO_REQUIRE(!pool || (H >= 2 && W >= 2), STITO_E_INVALID, "Given input size: (%dx%dx%d). Output size is too small", cout, H, W);
    ConvGeom g{};
    g.S = n; g.H = H; g.W = W; g.Cin = cin; g.Cout = cout;
    if (cout % 128 == 0) {
        return pool ? launch_conv_tw<2, 2, true>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st)
                    : launch_conv_tw<2, 2, false>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st);
    }
    return pool ? launch_conv_tw<4, 1, true>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st)
                : launch_conv_tw<4, 1, false>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st);
}

static void cnn14_dims(int64_t T, int M, int H[7], int W[7]) {
    H[0] = (int)T; W[0] = M;
    for (int b = 1; b <= 5; ++b) { H[b] = H[b - 1] / 2; W[b] = W[b - 1] / 2; }
    H[6] = H[5]; W[6] = W[5];
}

extern "C" size_t stito_cnn14_workspace_bytes(const stito_cnn14_weights *w, int n_streams, int64_t n_frames) {
    int H[7], W[7];
    cnn14_dims(n_frames, w->n_mels, H, W);
    size_t a = 0, b = 0;
    for (int blk = 0; blk < 6; ++blk) {
        const size_t full = (size_t)n_streams * H[blk] * W[blk] * w->channels[blk + 1];
        const size_t pooled = (size_t)n_streams * H[blk + 1] * W[blk + 1] * w->channels[blk + 1];
        a = full > a ? full : a;
        b = pooled > b ? pooled : b;
    }
    const size_t feat = (size_t)n_streams * w->channels[6];
    return align_up(a * 4, 256) + align_up(b * 4, 256) + align_up(feat * 4, 256) + 256;
}

extern "C" int stito_cnn14_forward(const stito_cnn14_weights *w, const float *logmel_dev, int n_cand, int channels,
                                   int64_t n_frames, float *mid_dev, float *side_dev, void *workspace_dev,
                                   size_t workspace_bytes, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(channels == 1 || channels == 2, STITO_E_INVALID, "Invalid number of channels: %d", channels);
    STITO_REQUIRE(n_cand > 0, STITO_E_INVALID, "empty batch");
    const int S = n_cand * channels;
    int H[7], W[7];
    cnn14_dims(n_frames, w->n_mels, H, W);
    STITO_REQUIRE(H[5] >= 1 && W[5] >= 1, STITO_E_INVALID,
                  "Given input size: (%dx%dx%d). Calculated output size is too small (audio shorter than 5 poolings)",
                  w->channels[5], H[4], W[4]);
    STITO_REQUIRE(workspace_bytes >= stito_cnn14_workspace_bytes(w, S, n_frames), STITO_E_WORKSPACE, "cnn14 workspace too small");
    size_t a = 0, b = 0;
    for (int blk = 0; blk < 6; ++blk) {
        const size_t full = (size_t)S * H[blk] * W[blk] * w->channels[blk + 1];
        const size_t pooled = (size_t)S * H[blk + 1] * W[blk + 1] * w->channels[blk + 1];
        a = full > a ? full : a;
        b = pooled > b ? pooled : b;
    }
    char *ws = (char *)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    float *actA = (float *)ws;
    float *actB = (float *)(ws + align_up(a * 4, 256));
    float *feat = (float *)(ws + align_up(a * 4, 256) + align_up(b * 4, 256));

    const float *cur = logmel_dev;
    for (int blk = 0; blk < 6; ++blk) {
        const int cin = w->channels[blk], cout = w->channels[blk + 1];
        int rc = stito_conv3x3_bn_relu(cur, w->conv_w_dev[2 * blk], w->bn_scale_dev[2 * blk], w->bn_shift_dev[2 * blk], actA,
                                       S, H[blk], W[blk], cin, cout, 0, stream);
        if (rc) return rc;
        rc = stito_conv3x3_bn_relu(actA, w->conv_w_dev[2 * blk + 1], w->bn_scale_dev[2 * blk + 1], w->bn_shift_dev[2 * blk + 1],
                                   actB, S, H[blk], W[blk], cout, cout, blk < 5 ? 1 : 0, stream);
        if (rc) return rc;
        cur = actB;
    }
    const int C6 = w->channels[6], E = w->embed_dim;
    hipLaunchKernelGGL(k_head, dim3((C6 + 255) / 256, S), dim3(256), 0, st, actB, feat, H[6], W[6], C6);
    STITO_LAUNCH_CHECK();
    const size_t lds = (size_t)FC_SB * C6 * sizeof(float);
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)k_fc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_fc, dim3((E + 255) / 256, (n_cand + FC_SB - 1) / FC_SB, channels), dim3(256), lds, st, feat,
                       w->fc_mid_wt_dev, w->fc_mid_b_dev, w->fc_side_wt_dev, w->fc_side_b_dev, mid_dev, side_dev, n_cand,
                       channels, C6, E);
    STITO_LAUNCH_CHECK();
    if (channels == 1) {  // side_embed = mid_embed (panns.py:271-274)
        const int64_t n = (int64_t)n_cand * E;
        hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mid_dev, side_dev, n);
        STITO_LAUNCH_CHECK();
    }
    return STITO_OK;
}

extern "C" int stito_embed_loss(float *mid_dev, float *side_dev, int n_cand, int embed_dim, const float *target_mid_dev,
                                const float *target_side_dev, float *loss_dev, int32_t *flags_dev, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(n_cand > 0 && embed_dim > 0, STITO_E_INVALID, "empty embeddings");
    STITO_REQUIRE((target_mid_dev == nullptr) == (target_side_dev == nullptr), STITO_E_INVALID, "need both targets or none");
    STITO_REQUIRE(target_mid_dev == nullptr || loss_dev != nullptr, STITO_E_INVALID, "loss output missing");
    STITO_HIP_CHECK(hipMemsetAsync(flags_dev, 0, 2 * sizeof(int32_t), st));
    const int64_t n = (int64_t)n_cand * embed_dim;
    hipLaunchKernelGGL(k_nan_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mid_dev, side_dev, n, flags_dev);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_embed_loss, dim3(n_cand), dim3(256), 0, st, mid_dev, side_dev, embed_dim, target_mid_dev,
                       target_side_dev, loss_dev, flags_dev, 0);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}
