// conv_direct_split.hip -- 3x3 conv (pad 1) + BN + ReLU (+ 2x2 average pool) of the Cnn14 trunk (reference: ConvBlock.forward,
// st_ito/models/panns.py:65-80) as a DIRECT implicit GEMM on the f16 matrix pipe with split operands.
//
// Why not Winograd here.  With every f32 operand carried as two f16 halves (hi hi' + hi lo' + lo hi' on
// v_mfma_f32_32x32x16_f16 is as close to float64 as the exact-f32 MFMA) a multiply-accumulate costs 3 / 16 of what it costs
// on the f32 pipe, so the 9 MACs per output of the direct form are 1.7 f32-pipe equivalents against F(4x4,3x3)'s 2.25 -- and the
// direct form has what the Winograd kernels lack on this chip: arithmetic intensity.  An activation element (4 bytes as hi + lo)
// meets 9 x 64 weights per pass over it and a weight element 512 pixels, where a Winograd-domain element meets 32 .. 64: the
// split Winograd kernels are bound by filling LDS (0.10 - 0.25 of the f16 peak), and the f32 Winograd kernel of the 64- and
// 128-channel layers by the f32 pipe plus its transforms, its 36-position exchange epilogue and its prologue (0.43 - 0.57 of the f32 peak).  Here there is no
// transform, no position dimension (64 accumulator registers instead of 144) and no exchange: BN + ReLU + pooling happen in
// registers as in the f32 direct kernel k_conv3x3 (cnn14.hip), whose tiling in "virtual row" space this kernel keeps.
//
// Workgroup = 256 output pixels (TH rows x TW columns of the virtual-row space) x 64 output channels, 4 waves of 64 x 64
// (2 x 2 MFMA blocks), 67 KB of LDS: TWO workgroups per CU, so that one's prologue (first loads from HBM), conversions and
// epilogue run under the other's MFMAs -- the first version (512 pixels, 8 waves, 150 KB, one workgroup per CU) spent 27 % of
// its time in prologues that nothing overlapped (measured by ablation: DS_ABL).  One period = 16 input channels = one k-step of
// the MFMA per tap, in three stages (one row of taps each):
//   weights   [cin/16][cout/64][tap][hi | lo][64 couts][16 channels] f16, scaled by the layer's power of two at pack time:
//             12 KB per stage, global -> registers (a stage ahead) -> LDS, double buffered;
//   input     f32 NC8HW8 -> registers (issued two periods ahead) -> x (stream's power of two) -> hi + lo -> LDS
//             [hi | lo][patch pixel][16 channels], double buffered, two items per stage; the patch is the tile's
//             (TH + 2) x (TW + 2) halo, rows counted in input-virtual-row space (rows of neighbouring streams are neighbours; the
//             zero padding between streams is applied per lane when the fragment is read, the padding left / right of the map
//             by the items that own those pixels writing zeros);
//   products  per tap and wave 4 + 4 ds_read_b128 (the shifted patch IS the im2col operand), issued a tap ahead, and 12 MFMAs.
// STATUS: parity-green on every conv case, NOT faster than the Winograd kernels it was meant to replace (512 streams, ms:
// 64->64 at 469 x 128: 8.2 against 7.7; 64->128 at 234 x 64: 4.5 / 4.6; 128->128: 6.8 / 7.0; 128->256 at 117 x 32: 3.75 / 3.4), so the
// model does not select it (STITO_CONV_DSPLIT_MAX_CIN, default 0).  By ablation (DS_ABL) its parts ADD UP instead of
// overlapping: first loads 2.4 ms, conversion 1.0, operand reads + MFMAs 4.2 (2.8 of MFMA at the f16 peak), epilogue 0.6 -- also
// with two workgroups per CU and with their starts staggered: a wave streaming MFMAs keeps the other waves of its SIMD from
// issuing (DESIGN 4.1(1) holds for the f16 pipe too), so the second workgroup's loads and conversions wait for the first one's
// MFMA phases instead of running under them.
// No LDS-DMA here: with every load going through registers hipcc counts vmcnt itself and waits for exactly what a statement
// needs; a stage ends with s_waitcnt lgkmcnt(0) + s_barrier (not __syncthreads, which would also drain the loads in flight).
#include "common.h"
#include "conv_layout.h"

namespace stito {

typedef _Float16 dh8 __attribute__((ext_vector_type(8)));
typedef _Float16 dh4 __attribute__((ext_vector_type(4)));

#ifndef DS_ABL
#define DS_ABL 0  // timing experiment (1 no MFMAs, 2 no operand reads either, 4 no conversion / input loads, 8 no weight copies, 16 no epilogue stores); 0 in every build that ships
#endif
static constexpr int DS_THREADS = 256;
static constexpr int DS_BM = 256, DS_BN = 64;
static constexpr int DS_B_BYTES = 9 * 2 * DS_BN * 32;  // weight slab of a period: [tap][hi | lo][cout][16 f16]
static constexpr int DS_STAGE_BYTES = DS_B_BYTES / 3;  // one row of taps: 12 KB = 3 x 16 bytes per thread
static constexpr int DS_MAX_ITEMS = 6;                 // (patch pixel, channel quad) items per thread: patches up to 384 pixels
#define DS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

// power-of-two scale of a stream's activations from the bit pattern of its maximum (>= 0): amax < 2^e -> 2^(14 - e)
__host__ __device__ __forceinline__ float ds_vscale(unsigned amax_bits) {
    int e = (int)((amax_bits >> 23) & 0xff) - 126;
    if (amax_bits == 0u) e = 14;
    e = e < -40 ? -40 : (e > 60 ? 60 : e);
    return __builtin_ldexpf(1.0f, 14 - e);
}

struct DsGeom {
    int S, H, W, Cin, Cout;
    int Heff;               // output rows per stream (POOL: 2 * (H / 2), else H)
    int VR, IVR;            // S * Heff output virtual rows, S * H input virtual rows
    int n_col_tiles, n_m_tiles, n_n_tiles;
    int PR, npix;           // patch rows, patch pixels (PR * (TW + 2))
    int Ho, Wo;
    unsigned *amax_out;     // or NULL
};

template <int TW, bool POOL>
__global__ __launch_bounds__(DS_THREADS, 2) void k_conv_dsplit(const float *__restrict__ in, const char *__restrict__ wpk,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                float *__restrict__ out, DsGeom g, const unsigned *__restrict__ amax,
                                                                const float *__restrict__ u_inv_p) {
    constexpr int TH = DS_BM / TW, GW = TW / 2, PW = TW + 2;
    // patch rows are PW pixels of 32 bytes + 16: the two rows of a 2x2 window then fall on different 16-byte bank groups
    // (row pitch = 16 mod 32), and a fragment read of 16 lanes covers 16 distinct groups
    constexpr int PITCH = PW * 32 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int a_plane = g.PR * PITCH + 32;  // bytes of one half (hi or lo) of an activation buffer; the last 32: a trash pixel
    const int a_bytes = 2 * a_plane;
    char *const b_base = smem + 2 * a_bytes;  // two stage buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    // channel tile fastest: the workgroups that share a halo patch run side by side
    const int n_tile = (int)blockIdx.x % g.n_n_tiles, m_tile = (int)blockIdx.x / g.n_n_tiles;
    const int n0 = n_tile * DS_BN;
    const int ct = m_tile % g.n_col_tiles, rt = m_tile / g.n_col_tiles;
    const int vr0 = rt * TH, w0 = ct * TW;
    const int s_lo = vr0 / g.Heff;
    const int iv_lo = s_lo * g.H + (vr0 - s_lo * g.Heff) - 1;  // input virtual row of patch row 0

    // (no zero fill: every patch pixel is written every period -- by the item that owns it, with zeros where there is no pixel)
    // ---- activation items: (patch pixel, channel quad); lanes 4 p .. 4 p + 3 write the 32 bytes of pixel p -----------------------
    const int64_t plane8 = (int64_t)g.H * g.W * 8;                 // floats per channel octet of one stream
    const int s_first = iv_lo < 0 ? 0 : iv_lo / g.H;               // first stream the patch touches
    const float *in0 = in + (int64_t)s_first * g.Cin * g.H * g.W;  // per-item offsets are relative to this (32-bit)
    int it_goff[DS_MAX_ITEMS], it_dst[DS_MAX_ITEMS];
    float it_sc[DS_MAX_ITEMS];
#pragma unroll
    for (int j = 0; j < DS_MAX_ITEMS; ++j) {
        const int i = tid + DS_THREADS * j;
        const int pix = i >> 2, cq = i & 3;
        const int p = pix / PW, pcol = pix - p * PW;
        const int iv = iv_lo + p, w = w0 + pcol - 1;
        const bool ok = pix < g.npix && iv >= 0 && iv < g.IVR && w >= 0 && w < g.W;
        const int s = ok ? iv / g.H : s_first;
        const int h = ok ? iv - s * g.H : 0;
        // items without a pixel (outside the map / the batch, or past the patch) load the stream's first element and write
        // zeros -- into the padding pixel they stand for or the trash pixel: no branch around any load or store,
        // so the six loads of a period are issued back to back (with a branch per item hipcc put s_waitcnt vmcnt(0) in front of
        // every one of them: six HBM latencies in series per period)
        it_goff[j] = ok ? (int)(((int64_t)(s - s_first) * (g.Cin >> 3) + (cq >> 1)) * plane8) + (h * g.W + w) * 8 + (cq & 1) * 4 : 0;
        it_dst[j] = (pix < g.npix ? p * PITCH + pcol * 32 : g.PR * PITCH) + cq * 8;
        it_sc[j] = ok ? ds_vscale(amax[s]) : 0.0f;
    }
    // two register sets: chunk c travels in set c % 2 -- loaded at the top of period c - 2, converted during period c - 1
    f32x4 raw0[DS_MAX_ITEMS], raw1[DS_MAX_ITEMS];
#define DS_LOAD_RAW(CH, RAW)                                                                              \
    {                                                                                                     \
        const float *cb_ = in0 + (int64_t)(CH) * 2 * plane8;                                               \
        _Pragma("unroll") for (int j = 0; j < DS_MAX_ITEMS; ++j) RAW[j] = *(const f32x4 *)(cb_ + it_goff[j]); \
    }
#define DS_CONVERT_ITEM(J, RAW, BUF)                                                                      \
    {                                                                                                     \
        char *ab_ = smem + (BUF) * a_bytes;                                                                \
        const f32x4 x_ = it_sc[J] != 0.0f ? RAW[J] * it_sc[J] : (f32x4)(0.0f);                             \
        dh4 hi_, lo_;                                                                                      \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { hi_[e_] = (_Float16)x_[e_]; lo_[e_] = (_Float16)(x_[e_] - (float)hi_[e_]); } \
        *(dh4 *)(ab_ + it_dst[J]) = hi_;                                                                   \
        *(dh4 *)(ab_ + a_plane + it_dst[J]) = lo_;                                                         \
    }
    // ---- weight stage copies: thread t moves bytes 16 (t + 256 j), j < 3, of the stage's 12 KB ------------------------------
    const char *w_tile = wpk + (int64_t)n_tile * DS_B_BYTES + tid * 16;
    const int64_t w_chunk_stride = (int64_t)g.n_n_tiles * DS_B_BYTES;
    f32x4 wreg[3];
#define DS_LOAD_W(G_) /* stage G_ = 3 chunk + row of taps */                                              \
    {                                                                                                     \
        const char *wb_ = w_tile + (int64_t)((G_) / 3) * w_chunk_stride + ((G_) % 3) * DS_STAGE_BYTES;     \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) wreg[j] = *(const f32x4 *)(wb_ + j * 4096);          \
    }
#define DS_STORE_W(G_)                                                                                    \
    {                                                                                                     \
        char *bd_ = b_base + ((G_) & 1) * DS_STAGE_BYTES + tid * 16;                                        \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) *(f32x4 *)(bd_ + j * 4096) = wreg[j];                \
    }
    // ---- per-lane fragment addresses and row masks (pixel mapping of k_conv3x3: 4 consecutive GEMM rows = one 2x2 window) -------
    const int wm = wv;
    int a_frag[2], b_frag[2];
    bool m_up[2], m_dn[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int gi = (wm * 2 + mb) * 8 + (l31 >> 2);
        const int gr = gi / GW, gc = gi % GW;
        int vr = vr0 + 2 * gr + ((l31 >> 1) & 1);
        vr = vr < g.VR ? vr : g.VR - 1;  // overhang lanes: any in-range row (result discarded)
        const int s = vr / g.Heff;
        const int h = vr - s * g.Heff;
        const int pc = s * g.H + h - iv_lo;  // patch row of the centre tap (>= 1)
        const int pw_ = 2 * gc + (l31 & 1);
        a_frag[mb] = (pc - 1) * PITCH + pw_ * 32 + half * 16;
        m_up[mb] = h >= 1;
        m_dn[mb] = h + 1 < g.H;
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) b_frag[nb] = (nb * 32 + l31) * 32 + half * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const int n_chunks = g.Cin >> 4, n_stages = 3 * n_chunks;
    DS_LOAD_RAW(0, raw0)
    DS_LOAD_W(0)
    DS_LOAD_RAW(n_chunks > 1 ? 1 : 0, raw1)
#pragma unroll
    for (int j = 0; j < DS_MAX_ITEMS; ++j) DS_CONVERT_ITEM(j, raw0, 0)
    DS_STORE_W(0)
    DS_BARRIER()

// operands of tap T (of the stage's three) into register set S (the shifted patch is the im2col operand; rows outside the stream read as zero)
#define DS_READ_OPS(KH, KW, S)                                                                            \
    {                                                                                                     \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) {                                                 \
            const int o_ = a_frag[mb] + (KH) * PITCH + (KW) * 32;                                          \
            ah[S][mb] = *(const dh8 *)(ab + o_);                                                           \
            al[S][mb] = *(const dh8 *)(ab + a_plane + o_);                                                 \
            if (((KH) == 0 && !m_up[mb]) || ((KH) == 2 && !m_dn[mb])) { ah[S][mb] = (dh8)(0); al[S][mb] = (dh8)(0); } \
        }                                                                                                  \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                                 \
            const int o_ = b_frag[nb] + (KW) * 4096;                                                       \
            bh[S][nb] = *(const dh8 *)(bb + o_);                                                           \
            bl[S][nb] = *(const dh8 *)(bb + 2048 + o_);                                                    \
        }                                                                                                  \
    }
// the three products of a block go to the same accumulator: the four blocks take turns
#define DS_MFMAS(S)                                                                                       \
    if (DS_ABL & 1) {                                                                                      \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                   \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                               \
                acc[mb][nb][0] += (float)ah[S][mb][0] + (float)al[S][mb][1] + (float)bh[S][nb][2] + (float)bl[S][nb][3]; \
    } else {                                                                                               \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                   \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[S][mb], bh[S][nb], acc[mb][nb], 0, 0, 0); \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                   \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[S][mb], bl[S][nb], acc[mb][nb], 0, 0, 0); \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                   \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[S][mb], bh[S][nb], acc[mb][nb], 0, 0, 0); \
    }
// One stage: row KH of taps of chunk K_ (stage G_ = 3 K_ + KH).  The weights of stage G_ + 1 are loaded at its top and stored at
// its end; two items of chunk K_ + 1 are converted inside it.
#define DS_STAGE(K_, KH, RAW_NEAR, RAW_FAR)                                                               \
    {                                                                                                     \
        const int g_ = 3 * (K_) + (KH);                                                                    \
        /* (past the end the loads, conversions and stores repeat the last chunk / stage into buffers nobody reads: no branch \
           around a load, so that hipcc's vmcnt bookkeeping stays exact) */                                                   \
        const int gn_ = g_ + 1 < n_stages ? g_ + 1 : g_;                                                   \
        if (!(DS_ABL & 8)) DS_LOAD_W(gn_)                                                                  \
        /* the input of chunk K_ + 2 is requested BEHIND this stage's weights: vmcnt counts in order, so the wait in front of  \
           the weights' store at the end of the stage leaves these six loads in flight (issued ahead of the weights they were   \
           drained -- from HBM -- at the end of every first stage) */                                                          \
        if ((KH) == 0 && !(DS_ABL & 4)) DS_LOAD_RAW((K_) + 2 < n_chunks ? (K_) + 2 : n_chunks - 1, RAW_FAR) \
        const char *ab = smem + ((K_) & 1) * a_bytes;                                                      \
        const char *bb = b_base + (g_ & 1) * DS_STAGE_BYTES;                                               \
        dh8 ah[2][2], al[2][2], bh[2][2], bl[2][2];                                                        \
        if (!(DS_ABL & 2)) {                                                                               \
            DS_READ_OPS(KH, 0, 0)                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            DS_READ_OPS(KH, 1, 1) DS_MFMAS(0)                                                              \
            if (!(DS_ABL & 4)) DS_CONVERT_ITEM(2 * (KH), RAW_NEAR, ((K_) & 1) ^ 1)                         \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            DS_READ_OPS(KH, 2, 0) DS_MFMAS(1)                                                              \
            if (!(DS_ABL & 4)) DS_CONVERT_ITEM(2 * (KH) + 1, RAW_NEAR, ((K_) & 1) ^ 1)                     \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            DS_MFMAS(0)                                                                                    \
        }                                                                                                  \
        if (!(DS_ABL & 8)) DS_STORE_W(g_ + 1)                                                              \
        DS_BARRIER() /* weights(G_ + 1) and (after the last stage) activations(K_ + 1) written; this stage's reads are done */ \
    }
#define DS_PERIOD(K_, RAW_NEAR, RAW_FAR)                                                                  \
    { DS_STAGE(K_, 0, RAW_NEAR, RAW_FAR) DS_STAGE(K_, 1, RAW_NEAR, RAW_FAR) DS_STAGE(K_, 2, RAW_NEAR, RAW_FAR) }
    static_assert(DS_MAX_ITEMS == 6, "DS_STAGE converts two items in each of the three stages of a period");
    for (int chunk = 0; chunk < n_chunks; chunk += 2) {
        DS_PERIOD(chunk, raw1, raw0)
        if (chunk + 1 < n_chunks) DS_PERIOD(chunk + 1, raw0, raw1)
    }

    if (DS_ABL & 16) { if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.f) out[tid] = 1.f; return; }
    // ---- epilogue: 1 / (weight scale x stream scale) folded into the BN scale, ReLU (+ 2x2 average pool), channel-blocked store
    const float u_inv = u_inv_p[0];
    unsigned mx0 = 0, mx1 = 0;  // largest stored output of streams s_lo and s_lo + 1 (later streams: atomics straight away)
    auto note_max = [&](int s, float v) {
        const unsigned b = __float_as_uint(v);
        if (s == s_lo) mx0 = max(mx0, b);
        else if (s == s_lo + 1) mx1 = max(mx1, b);
        else if (b > __hip_atomic_load(g.amax_out + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g.amax_out + s, b);
    };
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = (wm * 2 + mb) * 8 + 2 * q + half;
            const int gr = gi / GW, gc = gi % GW;
            const int vr = vr0 + 2 * gr;
            int64_t obase[2];
            bool ok[2];
            int so[2];
            if (POOL) {
                const int s = vr / g.Heff;
                const int oh = (vr - s * g.Heff) >> 1, ow = (w0 >> 1) + gc;
                ok[0] = vr < g.VR && ow < g.Wo;
                so[0] = ok[0] ? s : s_lo;
                obase[0] = act_off(so[0], 0, ok[0] ? oh : 0, ok[0] ? ow : 0, g.Cout, g.Ho, g.Wo);
                ok[1] = false; so[1] = s_lo; obase[1] = 0;
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {  // the two rows of the 2x2 register group
                    const int v = vr + e;
                    const int s = v / g.H;
                    ok[e] = v < g.VR;
                    so[e] = ok[e] ? s : s_lo;
                    obase[e] = act_off(so[e], 0, ok[e] ? v - s * g.H : 0, w0 + 2 * gc, g.Cout, g.H, g.W);
                }
            }
            const float vs0 = u_inv / ds_vscale(amax[so[0]]), vs1 = POOL ? vs0 : u_inv / ds_vscale(amax[so[1]]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int co = n0 + nb * 32 + l31;
                const float sc = scale[co], sh = shift[co];
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaxf(fmaf(acc[mb][nb][4 * q + e], sc * ((e >> 1) ? vs1 : vs0), sh), 0.0f);
                if (POOL) {
                    if (ok[0]) {
                        const float v = (((y[0] + y[1]) + y[2]) + y[3]) * 0.25f;
                        out[obase[0] + (int64_t)(co >> 3) * g.Ho * g.Wo * 8 + (co & 7)] = v;
                        if (g.amax_out != nullptr) note_max(so[0], v);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ww = w0 + 2 * gc + (e & 1);
                        if (ok[e >> 1] && ww < g.W) {
                            out[obase[e >> 1] + (int64_t)(co >> 3) * plane8 + (e & 1) * 8 + (co & 7)] = y[e];
                            if (g.amax_out != nullptr) note_max(so[e >> 1], y[e]);
                        }
                    }
                }
            }
        }
    }
    if (g.amax_out != nullptr) {  // one atomic per wave and stream, skipped when the running maximum already covers it
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx0 = max(mx0, (unsigned)__shfl_xor((int)mx0, o, 64));
            mx1 = max(mx1, (unsigned)__shfl_xor((int)mx1, o, 64));
        }
        if (lane == 0) {
            if (mx0 > __hip_atomic_load(g.amax_out + s_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g.amax_out + s_lo, mx0);
            if (mx1 && s_lo + 1 < g.S && mx1 > __hip_atomic_load(g.amax_out + s_lo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(g.amax_out + s_lo + 1, mx1);
        }
    }
}

// ---- weights: [cin/16][cout/64][tap][hi | lo][64 couts][16 channels] f16 of w x (layer scale), header {max bits, 1 / scale, scale} behind
__global__ void k_dsplit_wmax(const float *__restrict__ w, int64_t n, unsigned *__restrict__ hdr) {
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(hdr, m);
}
__global__ void k_dsplit_wscale(unsigned *hdr) {  // max |w| < 2^e -> scale 2^(14 - e)
    const unsigned mb = hdr[0];
    int e = (int)((mb >> 23) & 0xff) - 126;
    if (mb == 0u) e = 14;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    hdr[1] = __float_as_uint(__builtin_ldexpf(1.0f, e - 14));
    hdr[2] = __float_as_uint(__builtin_ldexpf(1.0f, 14 - e));
}
__global__ void k_dsplit_pack(const float *__restrict__ w /*[cout][cin][3][3]*/, int Cout, int Cin, char *__restrict__ o,
                              const unsigned *__restrict__ hdr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin * 9) return;
    const int tap = (int)(i % 9), ci = (int)((i / 9) % Cin), co = (int)(i / (9 * (int64_t)Cin));
    const float ws = w[i] * __uint_as_float(hdr[2]);
    const _Float16 hi = (_Float16)ws, lo = (_Float16)(ws - (float)hi);
    char *d = o + (((int64_t)(ci >> 4) * (Cout >> 6) + (co >> 6)) * 9 + tap) * 4096 + (co & 63) * 32 + (ci & 15) * 2;
    *(_Float16 *)d = hi;
    *(_Float16 *)(d + 2048) = lo;
}

size_t dsplit_packed_floats(int cout, int cin) { return (size_t)9 * cout * cin + 64; }

int pack_dsplit(const float *w_oihw, int cout, int cin, float *packed, hipStream_t st) {
    STITO_REQUIRE(cin % 16 == 0 && cout % 64 == 0, STITO_E_UNSUPPORTED, "conv (direct, split precision): cin %d / cout %d", cin, cout);
    const int64_t n = (int64_t)cout * cin * 9;
    unsigned *hdr = (unsigned *)(packed + (size_t)9 * cout * cin);
    STITO_HIP_CHECK(hipMemsetAsync(hdr, 0, 64 * sizeof(float), st));
    hipLaunchKernelGGL(k_dsplit_wmax, dim3((unsigned)((n + 256 * 8 - 1) / (256 * 8))), dim3(256), 0, st, w_oihw, n, hdr);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dsplit_wscale, dim3(1), dim3(1), 0, st, hdr);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dsplit_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, (char *)packed, (const unsigned *)hdr);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static int dsplit_tw(const ConvShape &c) { return c.W >= 32 && c.W % 32 == 0 ? 32 : 16; }

template <int TW>
static bool dsplit_geometry(const ConvShape &c, bool pool, DsGeom &g, size_t &lds, int64_t &blocks) {
    constexpr int TH = DS_BM / TW, PW = TW + 2;
    g = DsGeom{};
    g.S = c.S; g.H = c.H; g.W = c.W; g.Cin = c.Cin; g.Cout = c.Cout;
    g.Ho = c.H / 2; g.Wo = c.W / 2;
    g.Heff = pool ? 2 * g.Ho : c.H;
    if (g.Heff < 1 || c.W < 2 || c.Cin % 16 != 0 || c.Cout % 64 != 0) return false;
    const int64_t VR = (int64_t)c.S * g.Heff, IVR = (int64_t)c.S * c.H;
    if (IVR >= (1ll << 30) || (int64_t)c.Cin * c.H * c.W * 4 >= (1ll << 31)) return false;  // 32-bit row / offset arithmetic in the kernel
    g.VR = (int)VR; g.IVR = (int)IVR;
    g.n_col_tiles = (c.W + TW - 1) / TW;
    const int64_t n_row_tiles = (VR + TH - 1) / TH;
    if (n_row_tiles * g.n_col_tiles >= (1 << 30)) return false;
    g.n_m_tiles = (int)(n_row_tiles * g.n_col_tiles);
    g.n_n_tiles = c.Cout / DS_BN;
    g.PR = TH + 2 + ((TH - 1) / g.Heff + 1) * (c.H - g.Heff);
    g.npix = g.PR * PW;
    blocks = (int64_t)g.n_m_tiles * g.n_n_tiles;
    lds = (size_t)4 * (g.PR * (PW * 32 + 16) + 32) + 2 * DS_STAGE_BYTES;
    return g.npix * 4 <= DS_MAX_ITEMS * DS_THREADS && lds <= 80 * 1024 && blocks < (1ll << 31);  // two workgroups per CU
}

bool dsplit_supported(const ConvShape &c, bool pool) {
    DsGeom g;
    size_t lds;
    int64_t blocks;
    if (c.W < 16) return false;
    return dsplit_tw(c) == 32 ? dsplit_geometry<32>(c, pool, g, lds, blocks) : dsplit_geometry<16>(c, pool, g, lds, blocks);
}

// FLOPs of the MFMA instructions one launch issues (three products per operand pair, tile padding included)
double dsplit_issued_flops(const ConvShape &c, bool pool) {
    DsGeom g;
    size_t lds;
    int64_t blocks = 0;
    const bool ok = dsplit_tw(c) == 32 ? dsplit_geometry<32>(c, pool, g, lds, blocks) : dsplit_geometry<16>(c, pool, g, lds, blocks);
    return ok ? 3.0 * 2.0 * (double)blocks * DS_BM * DS_BN * 9.0 * c.Cin : 0.0;
}

size_t dsplit_workspace_bytes(const ConvShape &c, bool pool) {
    return dsplit_supported(c, pool) ? align_up((size_t)c.S * sizeof(unsigned), 256) : 0;
}

__global__ __launch_bounds__(256) void k_dsplit_absmax(const float *__restrict__ x, int64_t per_stream, unsigned *__restrict__ amax) {
    const f32x4 *xs = (const f32x4 *)(x + (int64_t)blockIdx.y * per_stream);
    const int64_t n4 = per_stream >> 2;
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = xs[i];
        m = max(max(m, max(__float_as_uint(v[0]) & 0x7fffffffu, __float_as_uint(v[1]) & 0x7fffffffu)),
                max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax + blockIdx.y, m);
}

template <int TW, bool POOL>
static int launch_ds(const float *in, const float *wpk, const float *scale, const float *shift, float *out, const ConvShape &c,
                     void *ws, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    DsGeom g;
    size_t lds;
    int64_t blocks;
    STITO_REQUIRE((dsplit_geometry<TW>(c, POOL, g, lds, blocks)), STITO_E_UNSUPPORTED,
                  "conv (direct, split precision): %dx%d map, %d -> %d channels does not fit the kernel's staging", c.H, c.W, c.Cin, c.Cout);
    const unsigned *amax = amax_in;
    if (amax_in == nullptr) {  // called on its own: the stream maxima of the input come from a scan
        unsigned *amax_ws = (unsigned *)ws;
        amax = amax_ws;
        STITO_HIP_CHECK(hipMemsetAsync(amax_ws, 0, (size_t)c.S * sizeof(unsigned), st));
        const int64_t per_stream = (int64_t)c.Cin * c.H * c.W;
        int splits = (int)((per_stream / 4 + 256 * 16 - 1) / (256 * 16));
        const int cap = (4096 + c.S - 1) / c.S;
        splits = splits > cap ? cap : (splits < 1 ? 1 : splits);
        hipLaunchKernelGGL(k_dsplit_absmax, dim3((unsigned)splits, (unsigned)c.S), dim3(256), 0, st, in, per_stream, amax_ws);
        STITO_LAUNCH_CHECK();
    }
    g.amax_out = amax_out;
    auto kern = k_conv_dsplit<TW, POOL>;
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float *u_inv = wpk + (size_t)9 * c.Cout * c.Cin + 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(DS_THREADS), lds, st, in, (const char *)wpk, scale, shift, out, g, amax, u_inv);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

int launch_dsplit(const float *in, const float *wpk, const float *scale, const float *shift, float *out, const ConvShape &c, bool pool,
                  void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    const size_t need = dsplit_workspace_bytes(c, pool);
    STITO_REQUIRE(need > 0 && (amax_in != nullptr || (ws != nullptr && ws_bytes >= need)), STITO_E_WORKSPACE,
                  "conv (direct, split precision): workspace have %zu need %zu", ws_bytes, need);
    if (dsplit_tw(c) == 32)
        return pool ? launch_ds<32, true>(in, wpk, scale, shift, out, c, ws, st, amax_in, amax_out)
                    : launch_ds<32, false>(in, wpk, scale, shift, out, c, ws, st, amax_in, amax_out);
    return pool ? launch_ds<16, true>(in, wpk, scale, shift, out, c, ws, st, amax_in, amax_out)
                : launch_ds<16, false>(in, wpk, scale, shift, out, c, ws, st, amax_in, amax_out);
}

}  // namespace stito
