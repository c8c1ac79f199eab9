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
// Layout: activations channel-blocked NC8HW8 (see act_off): a K-chunk's halo patch is a set of dense
// rows, 32 B per pixel.
// Weights are pre-packed as [cin/4][tap][cout][4].
//
// Tiling: a workgroup computes BM = 64*WM output pixels x BN = 64*WN output channels; each of its
// WM*WN waves owns a 64 x 64 sub-tile = 2 x 2 MFMA 32x32 blocks (64 accumulator VGPRs).  The BM
// pixels are TH rows x TW columns; rows are counted in "virtual row" space (the output rows of all
// streams concatenated) so every tile is full even when a feature map has 14 or 29 rows.
// M index -> pixel mapping: 4 consecutive GEMM rows = one 2x2 pooling window, which the MFMA
// C layout leaves in 4 consecutive accumulator registers of one lane, so BN+ReLU+avg-pool
// happen in registers.  Details of the data movement are at k_conv3x3.
#include "common.h"
#include "conv_layout.h"

#include <cstdlib>
#include <utility>
#include <vector>

namespace stito {

static constexpr int CK = 4;      // input channels per chunk (16 B per pixel / per weight row)

// Same copy with everything but the per-lane byte offset on the scalar unit: global address =
// sbase (SGPR pair, wave-uniform) + voff (32-bit per-lane offset), LDS base wave-uniform.  No VALU
// instruction at all -- on a SIMD whose vector pipe is saturated by f32 MFMAs every VALU/VGPR-port
// cycle of a co-resident wave is a cycle the MFMAs do not get.
__device__ __forceinline__ void glds16_s(const float *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}

// ------------------------------------------------------------------------------------------------
// k_conv3x3: 3x3 conv (pad 1) + per-channel scale/shift + ReLU (+ 2x2 average pool).
//   * rows are tiled in "virtual row" space (all streams' output rows concatenated), so tiles
//     are always full (no padding of 14-, 29-, 58-row maps to 16/32/64); a tile may straddle
//     streams, and the zero padding above/below a stream is applied per lane when the A fragment
//     is read (kh = 0 / 2 taps only).
//   * 4 input channels per chunk; the halo patch and the 9 x BN x 4 weight slab are copied
//     HBM/L2 -> LDS by global_load_lds (16 B per lane, no VGPR round trip, no ds_write), double
//     buffered: the copies for chunk c+1 are issued right after the single barrier of chunk c and
//     land while its 72 MFMAs per wave run.  ~45 KB LDS per workgroup -> 3 workgroups per CU.
// Weights: [cin/4][tap][cout][4].
// ------------------------------------------------------------------------------------------------
struct ConvGeom {
    int S, H, W, Cin, Cout;
    int Heff;        // output rows per stream (POOL: 2*(H/2), else H)
    int64_t VR;      // S * Heff
    int64_t IVR;     // S * H input virtual rows
    int n_col_tiles, n_m_tiles;
    int PR, na_i;    // patch rows, A wave-instructions per chunk
    int Ho, Wo;
};

// LDS-DMA: 16 B per lane, global -> LDS at (wave-uniform lds_dst) + lane * 16, no VGPR round trip.
// Issued through inline asm on purpose: with the __builtin_amdgcn_global_load_lds form hipcc
// cannot prove that the copy's destination (the idle buffer) does not alias the fragment reads of
// the live buffer and drains vmcnt(0) before the first ds_read, which serialises the prefetch with
// the MFMA block.  Hidden from its bookkeeping, the copy is waited for explicitly (s_waitcnt
// vmcnt(0) + barrier at the top of the next chunk).  M0 is saved/restored inside the statement.
__device__ __forceinline__ void glds16(const float *gsrc, float *lds_dst) {
    unsigned keep;
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds)
                 : "memory");
}

template <int WM, int WN, int TW, bool POOL>
__global__ __launch_bounds__(64 * WM * WN) void k_conv3x3(const float *__restrict__ in, const float *__restrict__ wpk,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ shift, float *__restrict__ out,
                                                             ConvGeom g) {
    constexpr int NW = WM * WN;
    constexpr int NT = 64 * NW;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int TH = BM / TW;
    constexpr int GW = TW / 2;
    constexpr int PW = TW + 2;
    constexpr int CK4 = CK;
    constexpr int NB_I = 9 * BN * CK4 / 256;   // weight-slab wave-instructions per chunk (1 KB each)
    constexpr int MAX_A_I = 8;
    constexpr int MAXI = (NB_I + MAX_A_I + NW - 1) / NW;
    constexpr int B_FLOATS = 9 * BN * CK4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int buf_floats = B_FLOATS + g.na_i * 256;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv % WM, wn = wv / WM;
    const int half = lane >> 5, l31 = lane & 31;

    const int m_tile = blockIdx.x % g.n_m_tiles;
    const int n_tile = blockIdx.x / g.n_m_tiles;
    const int n0 = n_tile * BN;
    const int ct = m_tile % g.n_col_tiles;
    const int64_t rt = m_tile / g.n_col_tiles;
    const int64_t vr0 = rt * TH;
    const int w0 = ct * TW;
    const int64_t iv_lo = (vr0 / g.Heff) * g.H + (vr0 % g.Heff) - 1;
    const int npix = g.PR * PW;

    // ---- zero both A regions once: padding / out-of-range pixels are never overwritten ---------
    for (int i = tid; i < g.na_i * 256; i += NT) {
        smem[B_FLOATS + i] = 0.0f;
        smem[buf_floats + B_FLOATS + i] = 0.0f;
    }

    // ---- per-thread copy descriptors ----------------------------------------------------------------
    const float *gsrc[MAXI];
    int ldso[MAXI], gstep[MAXI];
    bool gval[MAXI];
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int ii = wv + k * NW;  // wave-uniform
        if (ii < NB_I) {
            const int q = ii * 64 + lane;
            const int tap = q / BN, co = q % BN;
            gsrc[k] = wpk + ((int64_t)tap * g.Cout + n0 + co) * CK4;
            gstep[k] = 9 * g.Cout * CK4;
            ldso[k] = ii * 256;
            gval[k] = true;
        } else {
            const int a = ii - NB_I;
            const int pix = a * 64 + lane;
            const int p = pix / PW, pcol = pix % PW;
            const int64_t iv = iv_lo + p;
            const int w = w0 + pcol - 1;
            gval[k] = a < g.na_i && pix < npix && iv >= 0 && iv < g.IVR && w >= 0 && w < g.W;
            // 4-channel chunk c: channel block c/2, half c%2 -> + (c>>1) * H*W*8 + (c&1)*4 floats
            gsrc[k] = in + (gval[k] ? act_off(iv / g.H, 0, (int)(iv % g.H), w, g.Cin, g.H, g.W) : 0);
            gstep[k] = -1;  // marks an activation slot (plane-strided)
            ldso[k] = B_FLOATS + (a < g.na_i ? a : 0) * 256;
        }
    }
    const int64_t plane8 = (int64_t)g.H * g.W * 8;  // floats per channel block of one stream
    auto issue = [&](int chunk, int boff) {
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            if (gval[k])
                glds16(gsrc[k] + (gstep[k] >= 0 ? (int64_t)chunk * gstep[k] : (int64_t)(chunk >> 1) * plane8 + (chunk & 1) * 4),
                       smem + boff + ldso[k]);
        }
    };

    // ---- per-lane fragment addresses and row masks ---------------------------------------------------
    int a_frag[2], b_frag[2];
    bool m_up[2], m_dn[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int gi = (wm * 2 + mb) * 8 + (l31 >> 2);
        const int gr = gi / GW, gc = gi % GW;
        int64_t vr = vr0 + 2 * gr + ((l31 >> 1) & 1);
        vr = vr < g.VR ? vr : g.VR - 1;  // overhang lanes: any in-range row (result discarded)
        const int64_t s = vr / g.Heff;
        const int h = (int)(vr % g.Heff);
        const int pc = (int)(s * g.H + h - iv_lo);  // patch row of the centre tap (>= 1)
        const int pw_ = 2 * gc + (l31 & 1);
        a_frag[mb] = B_FLOATS + ((pc - 1) * PW + pw_) * CK4 + half * 2;
        m_up[mb] = h >= 1;
        m_dn[mb] = h + 1 < g.H;
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) b_frag[nb] = (wn * 64 + nb * 32 + l31) * CK4 + half * 2;

    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const int n_chunks = g.Cin / CK4;
    __syncthreads();  // zero fill visible before any copy can land next to it
    issue(0, 0);
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int cur = (chunk & 1) * buf_floats;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies for `chunk` have landed
        __syncthreads();                                   // ... everyone's; and chunk-1's reads are done
        if (chunk + 1 < n_chunks) issue(chunk + 1, buf_floats - cur);
        const float *sb = smem + cur;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            float2 av[2], bv[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                av[mb] = *(const float2 *)(sb + a_frag[mb] + (kh * PW + kw) * CK4);
                if (kh == 0 && !m_up[mb]) av[mb] = make_float2(0.f, 0.f);
                if (kh == 2 && !m_dn[mb]) av[mb] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) bv[nb] = *(const float2 *)(sb + b_frag[nb] + tap * BN * CK4);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0].x, bv[0].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0].x, bv[1].x, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1].x, bv[0].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1].x, bv[1].x, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0].y, bv[0].y, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0].y, bv[1].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1].y, bv[0].y, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1].y, bv[1].y, acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue: BN + ReLU (+ 2x2 average pool), NHWC store ------------------------------------------
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = (wm * 2 + mb) * 8 + 2 * q + half;
            const int gr = gi / GW, gc = gi % GW;
            const int64_t vr = vr0 + 2 * gr;
            int64_t obase[2];
            bool ok[2];
            if (POOL) {
                const int64_t s = vr / g.Heff;
                const int oh = (int)(vr % g.Heff) >> 1, ow = (w0 >> 1) + gc;
                ok[0] = vr < g.VR && ow < g.Wo;
                obase[0] = act_off(s, 0, oh, ow, g.Cout, g.Ho, g.Wo);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {  // the two rows of the 2x2 register group
                    const int64_t v = vr + e;
                    const int64_t s = v / g.H;
                    const int h = (int)(v % g.H);
                    ok[e] = v < g.VR;
                    obase[e] = act_off(s, 0, h, w0 + 2 * gc, g.Cout, g.H, g.W);
                }
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int co = n0 + wn * 64 + nb * 32 + l31;
                const float sc = scale[co], sh = shift[co];
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaxf(fmaf(acc[mb][nb][4 * q + e], sc, sh), 0.0f);
                // channel co of the blocked layout: block co/8 is a plane of Ho*Wo (or H*W) pixels x 8
                if (POOL) {
                    if (ok[0]) out[obase[0] + (int64_t)(co >> 3) * g.Ho * g.Wo * 8 + (co & 7)] = (((y[0] + y[1]) + y[2]) + y[3]) * 0.25f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ww = w0 + 2 * gc + (e & 1);
                        if (ok[e >> 1] && ww < g.W) out[obase[e >> 1] + (int64_t)(co >> 3) * plane8 + (e & 1) * 8 + (co & 7)] = y[e];
                    }
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) (Lavin & Gray 2016): Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A per 4x4 input tile d.  The 16
// element-wise products become 16 independent GEMMs  M_p[tile, cout] = V_p[tile, cin] U_p[cin, cout], 16 MACs per 2x2
// output tile instead of 36: 2.25x fewer MFMA cycles at float32 accuracy comparable to the direct form (all transform
// coefficients are 0, +-1, +-1/2).  Workgroup = 64 tiles x 64 output channels x 16 positions, 8-channel chunks:
//   U  pre-transformed weights [cin/8][16][cout][8], copied by scalar-addressed LDS-DMA;
//   V  B^T d B of each tile's 4x4 patch, read from the channel-blocked activations (bounds-checked = zero padding,
//      stream boundaries included), double buffered in LDS as [pos][(tile + pos/4) % 64][8];
//   the 16 positions of one output meet only in the epilogue: a wave reduces its row over nu in registers (column half
//   of A^T M A), the 4 x 2 partial results per (tile, channel) are exchanged through LDS in one pass, then the row half,
//   BN, ReLU and the 2x2 average pool (one Winograd tile == one pooling window) are applied per (tile, channel).
// The first-round kernel (k_conv_wino: 8 consumer + 4 producer waves, 83 ms per trunk pass) is gone; what its LDS-stamp
// traces showed -- producers only run once the consumers sit at the barrier -- is explained and fixed in k_conv_wino8 below.
// ------------------------------------------------------------------------------------------------
struct WinoGeom {
    int S, H, W, Cin, Cout;
    int TR, TC;        // tile rows per stream / tile columns that produce output
    int64_t VTR;       // S * TR
    int n_col_blocks, n_m_blocks;
    int Ho, Wo;
    int PR, pa_i;      // halo patch rows; patch float4 count / 64 (patch buffer size)
    long long *trace;  // TRACE instantiation only: s_memtime stamps (tools/wino_timeline.py)
};
static constexpr int WK = 8;           // input channels per chunk

// ------------------------------------------------------------------------------------------------
// k_conv_wino8: the same Winograd F(2x2,3x3) tile (64 tiles x 64 output channels x 16 positions, 8-channel
// chunks, same LDS layouts and epilogue as k_conv_wino) with EIGHT SYMMETRIC WAVES and no producer role.
//
// Why (tools/ubench/mfma_coissue.hip, mfma_interleave.hip, measured on MI355X):
//   * a wave that streams independent v_mfma_f32_32x32x2_f32 starves every OTHER wave of its SIMD almost
//     completely -- VALU, LDS, SALU alike, whatever their age or s_setprio -- until it blocks on a counter or a
//     barrier (a filler wave got 3 instruction groups in 263 000 cycles).  k_conv_wino's producer waves therefore
//     only ran once the consumers sat at the chunk barrier: every chunk paid the MFMA phase PLUS a serial
//     transform tail (mfma_issued_frac 0.66).
//   * instructions of the wave's OWN stream do issue between its MFMAs: a ds_read_b128 costs ~1.5 cycles of
//     MFMA time, a VALU instruction 4-6 (the f32 matrix pipe is the f32 VALU pipe: VALU time is additive, it
//     cannot be hidden, only kept small), a ds_write_b128 ~13 cycles of the CU's LDS write path.
// So every wave does an eighth of everything, interleaved by hand into its MFMA stream:
//   MFMA   row xi = w / 2 of the Winograd domain x 64 tiles x half (w % 2) of the channels: 32 MFMAs per chunk,
//          operands of group nu+1 read (3 ds_read_b128) while group nu runs;
//   U      4 of the 32 scalar-addressed LDS-DMA copies of the next chunk's weight slab;
//   patch  2 float4 of the halo patch, HBM -> registers two periods ahead -> LDS (no VALU: scalar base + lane offset);
//   V      one transform item (tile, channel quad, row of B^T d B): 8 ds_read_b128, 16 packed VALU, 4 ds_write_b128.
// The loop is rotated so that ONE barrier per chunk is enough and nothing waits behind it: barrier(k) sits after
// the operand reads of (chunk k, nu = 3); the period that follows runs those 8 MFMAs from registers while the
// first reads of chunk k+1 are in flight, then (k+1, nu = 0..2).  V(k+1) / U(k+1) are produced during period k
// into the buffer whose last reader passed barrier(k-1).
// ------------------------------------------------------------------------------------------------
static constexpr int WINO8_THREADS = 512;

template <int TTW, bool POOL, bool TRACE = false>
__global__ __launch_bounds__(WINO8_THREADS) void k_conv_wino8(const float *__restrict__ in, const float *__restrict__ upk,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ shift, float *__restrict__ out,
                                                               WinoGeom g) {
    constexpr int TTH = 64 / TTW;
    constexpr int U_FLOATS = 16 * 64 * WK;  // [pos][cout][8]
    constexpr int V_FLOATS = 16 * 64 * WK;  // [pos][(tile + pos/4) % 64][8]
    constexpr int BUF = U_FLOATS + V_FLOATS;
    constexpr int PWC = 2 * TTW + 2;        // patch columns
    constexpr int NPL = 2;                  // float4 of patch per thread per chunk: 2 PR PWC <= 1024 (geometry check)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = g.Cout / 64;
    const int m_blk = blockIdx.x / n_tiles;  // channel tile fastest: the workgroups sharing a halo patch run side by side
    const int n0 = (blockIdx.x % n_tiles) * 64;
    const int cb = m_blk % g.n_col_blocks;
    const int rb = m_blk / g.n_col_blocks;
    const int vtr0 = rb * TTH;
    const int tc0 = cb * TTW;
    const int n_chunks = g.Cin / WK;
    const int pfl = g.pa_i * 256 + 512;  // floats per patch buffer: pixels + a row of zeros + a trash row
    float *patch0 = smem + 2 * BUF;
    const int iv_lo = (vtr0 / g.TR) * g.H + 2 * (vtr0 % g.TR) - 1;  // input virtual row (s*H + h) of patch row 0

    // ---- MFMA role: row xi of the Winograd domain, channel half nh -------------------------------------
    const int half = lane >> 5, l31 = lane & 31;
    const int xi = wv >> 1, nh = wv & 1;
    // operand offsets of nu = 0 inside a U/V buffer (floats); nu adds 64 * WK
    const int a_off0 = U_FLOATS + (4 * xi) * 64 * WK + ((l31 + xi) & 63) * WK + half * 4;
    const int a_off1 = U_FLOATS + (4 * xi) * 64 * WK + ((32 + l31 + xi) & 63) * WK + half * 4;
    const int b_off = (4 * xi) * 64 * WK + (nh * 32 + l31) * WK + half * 4;

    // ---- transform item: thread = (tile, channel quad, row t_xi of B^T d B) --------------------------------
    const int t_xi = tid & 3, t_quad = (tid >> 2) & 1, t_tile = tid >> 3;
    int rowA, rowB, vdst;
    float sgn;
    {
        const int ra = t_xi == 0 ? 0 : (t_xi == 2 ? 2 : 1), rbb = t_xi == 3 ? 3 : (t_xi == 2 ? 1 : 2);
        sgn = t_xi == 1 ? 1.0f : -1.0f;  // T[t_xi] = d[ra] + sgn * d[rbb]
        const int vtr = vtr0 + t_tile / TTW, tcl = t_tile % TTW;
        const int s_ = vtr / g.TR, tr = vtr % g.TR;
        const int pc0 = s_ * g.H + 2 * tr - 1 - iv_lo;  // patch row of this tile's first input row
        const int ha = 2 * tr - 1 + ra, hb = 2 * tr - 1 + rbb;
        const int zoff = g.pa_i * 256;                   // the row of zeros: rows outside the stream / map
        rowA = ((vtr < g.VTR && ha >= 0 && ha < g.H) ? ((pc0 + ra) * PWC + 2 * tcl) * 8 : zoff) + t_quad * 4;
        rowB = ((vtr < g.VTR && hb >= 0 && hb < g.H) ? ((pc0 + rbb) * PWC + 2 * tcl) * 8 : zoff) + t_quad * 4;
        // V plane p = 4 t_xi + nu keeps tile t at slot (t + t_xi) % 64 (bank spread of the four writers of a tile)
        vdst = U_FLOATS + (t_xi * 4) * 64 * WK + ((t_tile + t_xi) & 63) * WK + t_quad * 4;
    }

    // ---- U slab copies: 32 wave-instructions per chunk (position ii / 2, half ii % 2 of its 2 KB), 4 per wave ----
    const float *u_base = upk + (int64_t)n0 * WK;
    const int64_t u_pos_stride = (int64_t)g.Cout * WK, u_chunk_stride = 16 * u_pos_stride;  // floats
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;

    // ---- halo patch staging: float4 q = tid + 512 j of the patch ((pixel, half) linear), scalar base + lane offset ----
    const int s_first = (iv_lo < 0 ? 0 : iv_lo) / g.H;  // first stream the patch touches (wave-uniform)
    const float *p_base = in + act_off(s_first, 0, 0, 0, g.Cin, g.H, g.W);
    const int64_t plane8 = (int64_t)g.H * g.W * 8;      // floats per 8-channel plane of one stream
    unsigned p_off[NPL];                                // byte offset from p_base (+ chunk * plane8 floats)
    int p_dst[NPL];                                     // float offset in a patch buffer; lanes without a pixel hit the trash row
    {
        const int npix2 = g.PR * PWC * 2;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int q = tid + WINO8_THREADS * j, pix = q >> 1;
            const int pr = pix / PWC, pc = pix % PWC;
            const int iv = iv_lo + pr;
            const int w = 2 * tc0 - 1 + pc;
            const bool ok = q < npix2 && iv >= 0 && iv < g.S * g.H && w >= 0 && w < g.W;
            const int s_ = ok ? iv / g.H : s_first, h_ = ok ? iv % g.H : 0, w_ = ok ? w : 0;
            p_off[j] = (unsigned)(((int64_t)(s_ - s_first) * (g.Cin >> 3) * plane8 + ((int64_t)h_ * g.W + w_) * 8 + (q & 1) * 4) * 4);
            p_dst[j] = ok ? q * 4 : g.pa_i * 256 + 256 + lane * 4;
        }
    }
    f32x4 rpA[NPL], rpB[NPL];

#define W8_COPY_U1(CH, BOFF, K_) /* copy K_ (0..3) of this wave's four */                               \
    if ((CH) < n_chunks) {                                                                               \
        const int ii = wv * 4 + (K_);                                                                    \
        glds16_m0(u_base + (int64_t)(CH) * u_chunk_stride + (ii >> 1) * u_pos_stride + (ii & 1) * 256,   \
                  (unsigned)lane * 16u, lds0 + (unsigned)((BOFF) + (ii >> 1) * 64 * WK + (ii & 1) * 256) * 4u); \
    }
#define W8_COPY_U(CH, BOFF) W8_COPY_U1(CH, BOFF, 0) W8_COPY_U1(CH, BOFF, 1) W8_COPY_U1(CH, BOFF, 2) W8_COPY_U1(CH, BOFF, 3)
#define W8_LOAD_P(rp, CH)                                                                               \
    {                                                                                                   \
        const int cc_ = (CH) < n_chunks ? (CH) : n_chunks - 1;                                           \
        const char *pb_ = (const char *)(p_base + (int64_t)cc_ * plane8);                                \
        _Pragma("unroll") for (int j = 0; j < NPL; ++j) rp[j] = *(const f32x4 *)(pb_ + p_off[j]);        \
    }
#define W8_WRITE_P(rp, PBUF)                                                                            \
    {                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < NPL; ++j) *(f32x4 *)((PBUF) + p_dst[j]) = rp[j];           \
    }
// transform, in three steps that the main loop places between MFMA groups: reads, row combination, column combination + stores
#define LO2(v) __builtin_shufflevector(v, v, 0, 1)
#define HI2(v) __builtin_shufflevector(v, v, 2, 3)
#define CAT4(a, b) __builtin_shufflevector(a, b, 0, 1, 2, 3)
#define W8_T_READ(PBUF, J0) /* columns J0, J0 + 1 of the two patch rows this item combines */           \
    {                                                                                                   \
        _Pragma("unroll") for (int j = (J0); j < (J0) + 2; ++j) {                                        \
            tA[j] = *(const f32x4 *)((PBUF) + rowA + j * 8);                                             \
            tB[j & 1] = *(const f32x4 *)((PBUF) + rowB + j * 8);                                         \
        }                                                                                                \
    }
#define W8_T_ROWS(J0) /* T = A + sgn B, as fma(B, sgn, A): exact */                                      \
    {                                                                                                   \
        _Pragma("unroll") for (int j = (J0); j < (J0) + 2; ++j)                                          \
            tA[j] = CAT4(pk_fma(LO2(tB[j & 1]), sg2, LO2(tA[j])), pk_fma(HI2(tB[j & 1]), sg2, HI2(tA[j]))); \
    }
#define W8_T_STORE1(VBOFF, P_, EXPR_LO, EXPR_HI)                                                        \
    *(f32x4 *)(smem + (VBOFF) + vdst + (P_) * 64 * WK) = CAT4(EXPR_LO, EXPR_HI);
#define W8_T_STORE_0(VBOFF) W8_T_STORE1(VBOFF, 0, pk_sub(LO2(tA[0]), LO2(tA[2])), pk_sub(HI2(tA[0]), HI2(tA[2])))
#define W8_T_STORE_1(VBOFF) W8_T_STORE1(VBOFF, 1, pk_add(LO2(tA[1]), LO2(tA[2])), pk_add(HI2(tA[1]), HI2(tA[2])))
#define W8_T_STORE_2(VBOFF) W8_T_STORE1(VBOFF, 2, pk_sub(LO2(tA[2]), LO2(tA[1])), pk_sub(HI2(tA[2]), HI2(tA[1])))
#define W8_T_STORE_3(VBOFF) W8_T_STORE1(VBOFF, 3, pk_sub(LO2(tA[1]), LO2(tA[3])), pk_sub(HI2(tA[1]), HI2(tA[3])))
#define W8_T_STORE(VBOFF) W8_T_STORE_0(VBOFF) W8_T_STORE_1(VBOFF) W8_T_STORE_2(VBOFF) W8_T_STORE_3(VBOFF)
#define W8_LOAD_OPS(S, SB, NU)                                                                          \
    {                                                                                                   \
        S##a0 = *(const f32x4 *)((SB) + a_off0 + (NU) * 64 * WK);                                        \
        S##a1 = *(const f32x4 *)((SB) + a_off1 + (NU) * 64 * WK);                                        \
        S##b = *(const f32x4 *)((SB) + b_off + (NU) * 64 * WK);                                          \
    }
#define W8_MFMA1(S, NU, MB, KK)                                                                         \
    acc[NU][MB] = __builtin_amdgcn_mfma_f32_32x32x2f32(S##a##MB[KK], S##b[KK], acc[NU][MB], 0, 0, 0);
#define W8_MFMA2(S, NU, KK) W8_MFMA1(S, NU, 0, KK) W8_MFMA1(S, NU, 1, KK)
#define W8_FENCE() __builtin_amdgcn_sched_barrier(0);
#define W8_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    f32x16 acc[4][2];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][mb][r] = 0.0f;
    f32x4 tA[4], tB[2];
    const f32x2 sg2 = {sgn, sgn};
// TRACE instantiation (stito_debug_wino_trace): s_memtime stamps of the workgroups with blockIdx % 256 == 100, lane 0 of
// waves 0 and 4: [wg / 256][wave / 4][slot]: 0 entry, 1 prologue done, 2 main loop done, 3 epilogue done, 4.. barrier(k) of
// the first 12 periods (tools/wino_timeline.py)
#define W8_STAMP(SLOT)                                                                                  \
    if (TRACE && (blockIdx.x & 255) == 100 && (blockIdx.x >> 8) < 8 && lane == 0 && (wv & 3) == 0)      \
        g.trace[((blockIdx.x >> 8) * 2 + (wv >> 2)) * 16 + (SLOT)] = (long long)__builtin_readcyclecounter();
    W8_STAMP(0)
    f32x4 xa0, xa1, xb, ya0, ya1, yb;  // two operand sets: one feeds the running MFMA group, the other is being read

    // ---- prologue: patch(0), patch(1) in LDS, V(0) and U(0) in buffer 0 -------------------------------------
    W8_COPY_U(0, 0)
    W8_LOAD_P(rpA, 0)
    W8_LOAD_P(rpB, 1)
    for (int i = tid; i < 2 * pfl; i += WINO8_THREADS) patch0[i] = 0.0f;  // both patch buffers + zero / trash rows
    W8_BARRIER()
    W8_WRITE_P(rpA, patch0)        // (the compiler's wait for the loaded registers also covers the older U(0) copies)
    W8_WRITE_P(rpB, patch0 + pfl)
    W8_LOAD_P(rpA, 2)
    W8_LOAD_P(rpB, 3)
    W8_BARRIER()
    W8_T_READ(patch0, 0)
    W8_T_ROWS(0)
    W8_T_READ(patch0, 2)
    W8_T_ROWS(2)
    W8_T_STORE(0)
    W8_BARRIER()                   // B(-1): V(0), U(0) complete
    W8_STAMP(1)

// one period: k = chunk whose groups nu = 0..2 run here; FIRST: there is no (k-1, nu = 3) group in flight.
// Operand sets alternate x, y, x, y over the four groups of a period, so no register is ever copied.  One piece of
// non-MFMA work per MFMA gap (64 cycles of matrix pipe), the production of chunk k+1 in the first half of the period.
#define W8_G(S, NU, MB, KK, WORK) W8_MFMA1(S, NU, MB, KK) WORK W8_FENCE()
#define W8_PERIOD(rp, FIRST)                                                                            \
    {                                                                                                   \
        const int cur = (k & 1) * BUF, nxt = BUF - cur;                                                  \
        const float *sb = smem + cur;                                                                    \
        float *pb_w = patch0 + (k & 1) * pfl;           /* patch(k+2) goes where patch(k) was */         \
        const float *pb_r = patch0 + ((k + 1) & 1) * pfl; /* patch(k+1) */                               \
        const bool more = k + 1 < n_chunks;                                                              \
        W8_LOAD_OPS(y, sb, 0)                                                                            \
        W8_WRITE_P(rp, pb_w)                                                                             \
        W8_FENCE()                                                                                       \
        if (!(FIRST)) { W8_G(x, 3, 0, 0, W8_COPY_U1(k + 1, nxt, 0)) } else { W8_COPY_U1(k + 1, nxt, 0) } \
        if (!(FIRST)) { W8_G(x, 3, 1, 0, W8_COPY_U1(k + 1, nxt, 1)) } else { W8_COPY_U1(k + 1, nxt, 1) } \
        if (!(FIRST)) { W8_G(x, 3, 0, 1, W8_COPY_U1(k + 1, nxt, 2)) } else { W8_COPY_U1(k + 1, nxt, 2) } \
        if (!(FIRST)) { W8_G(x, 3, 1, 1, W8_COPY_U1(k + 1, nxt, 3)) } else { W8_COPY_U1(k + 1, nxt, 3) } \
        if (!(FIRST)) { W8_G(x, 3, 0, 2, W8_LOAD_P(rp, k + 4)) } else { W8_LOAD_P(rp, k + 4) }           \
        if (!(FIRST)) { W8_G(x, 3, 1, 2, if (more) W8_T_READ(pb_r, 0)) } else { if (more) W8_T_READ(pb_r, 0) } \
        if (!(FIRST)) { W8_G(x, 3, 0, 3, ) }                                                             \
        if (!(FIRST)) { W8_G(x, 3, 1, 3, ) }                                                             \
        W8_FENCE()                                                                                       \
        W8_G(y, 0, 0, 0, W8_LOAD_OPS(x, sb, 1))                                                          \
        W8_G(y, 0, 1, 0, if (more) W8_T_ROWS(0))                                                         \
        W8_G(y, 0, 0, 1, if (more) W8_T_READ(pb_r, 2))                                                   \
        W8_G(y, 0, 1, 1, )                                                                               \
        W8_G(y, 0, 0, 2, )                                                                               \
        W8_G(y, 0, 1, 2, if (more) W8_T_ROWS(2))                                                         \
        W8_G(y, 0, 0, 3, if (more) W8_T_STORE_0(nxt))                                                    \
        W8_G(y, 0, 1, 3, if (more) W8_T_STORE_1(nxt))                                                    \
        W8_G(x, 1, 0, 0, W8_LOAD_OPS(y, sb, 2))                                                          \
        W8_G(x, 1, 1, 0, if (more) W8_T_STORE_2(nxt))                                                    \
        W8_G(x, 1, 0, 1, if (more) W8_T_STORE_3(nxt))                                                    \
        W8_G(x, 1, 1, 1, )                                                                               \
        W8_G(x, 1, 0, 2, )                                                                               \
        W8_G(x, 1, 1, 2, )                                                                               \
        W8_G(x, 1, 0, 3, )                                                                               \
        W8_G(x, 1, 1, 3, )                                                                               \
        W8_G(y, 2, 0, 0, W8_LOAD_OPS(x, sb, 3))                                                          \
        W8_MFMA1(y, 2, 1, 0)                                                                             \
        W8_MFMA2(y, 2, 1)                                                                                \
        W8_MFMA2(y, 2, 2)                                                                                \
        W8_MFMA2(y, 2, 3)                                                                                \
        W8_FENCE()                                                                                       \
        /* U(k+1) landed.  vmcnt(0), not vmcnt(NPL): LDS-DMA copies and register loads do not complete in issue order on */ \
        /* this part (tools/conv_stress.py caught k_conv_wino43 consuming a slab early under a partial wait), so the patch */ \
        /* loads of this period are drained too -- they were issued 28 MFMA gaps ago.                                      */ \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        W8_BARRIER()                                     /* B(k) */                                      \
        if (TRACE && k < 12) { W8_STAMP(4 + k) }                                                         \
    }

    {
        int k = 0;
        W8_PERIOD(rpA, true)
        for (k = 1; k + 1 < n_chunks; k += 2) {
            W8_PERIOD(rpB, false)
            ++k;
            W8_PERIOD(rpA, false)
            --k;
        }
        if (k < n_chunks) W8_PERIOD(rpB, false)
    }
    W8_MFMA2(x, 3, 0)
    W8_MFMA2(x, 3, 1)
    W8_MFMA2(x, 3, 2)
    W8_MFMA2(x, 3, 3)
    W8_STAMP(2)
#undef W8_PERIOD
#undef W8_G
#undef W8_COPY_U
#undef W8_COPY_U1
#undef W8_MFMA1
#undef W8_T_STORE1
#undef W8_T_STORE_0
#undef W8_T_STORE_1
#undef W8_T_STORE_2
#undef W8_T_STORE_3
#undef W8_LOAD_P
#undef W8_WRITE_P
#undef W8_T_READ
#undef W8_T_ROWS
#undef W8_T_STORE
#undef W8_LOAD_OPS
#undef W8_MFMA2
#undef W8_FENCE

    // ---- epilogue (as k_conv_wino): column half of A^T M A in registers, exchange through LDS, row half + BN +
    // ReLU (+ 2x2 average pool) per (tile, channel)
    constexpr int XT = 72, XP = 64 * XT;
    float *xch = smem;
    W8_BARRIER()  // every wave is past its last LDS read of the main loop
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        float *xp = xch + (2 * xi) * XP + mb * 32 * XT + nh * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float m0 = acc[0][mb][r], m1 = acc[1][mb][r], m2 = acc[2][mb][r], m3 = acc[3][mb][r];
            xp[row * XT] = (m0 + m1) + m2;
            xp[XP + row * XT] = (m1 - m2) - m3;
        }
    }
    W8_BARRIER()
#undef W8_BARRIER
    const int e_co = tid & 63, e_t0 = tid >> 6;  // thread -> channel, tiles e_t0 + 8*k
    const int co = n0 + e_co;
    const float sc = scale[co], sh = shift[co];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int tl = e_t0 + 8 * it;
        float c0[4], c1[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            c0[x] = xch[(2 * x) * XP + tl * XT + e_co];
            c1[x] = xch[(2 * x + 1) * XP + tl * XT + e_co];
        }
        float y[4];
        y[0] = (c0[0] + c0[1]) + c0[2];
        y[1] = (c1[0] + c1[1]) + c1[2];
        y[2] = (c0[1] - c0[2]) - c0[3];
        y[3] = (c1[1] - c1[2]) - c1[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(fmaf(y[e], sc, sh), 0.0f);
        const int vtr = vtr0 + tl / TTW;
        const int tc = tc0 + tl % TTW;
        if (vtr < g.VTR && tc < g.TC) {
            const int s = vtr / g.TR;
            const int tr = vtr % g.TR;
            if (POOL) {
                out[act_off(s, co, tr, tc, g.Cout, g.Ho, g.Wo)] = (((y[0] + y[1]) + y[2]) + y[3]) * 0.25f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int hh = 2 * tr + (e >> 1), ww = 2 * tc + (e & 1);
                    if (hh < g.H && ww < g.W) out[act_off(s, co, hh, ww, g.Cout, g.H, g.W)] = y[e];
                }
            }
        }
    }
    W8_STAMP(3)
#undef W8_STAMP
}

// ------------------------------------------------------------------------------------------------
// conv_block1.conv1: one input channel (the log-mel image) -> 64 channels.  K = 9: no matrix shape to speak of; direct,
// bound by the write of its output (S*T*M*64 floats = 7.9 GB at 512 streams).  thread = (channel octet, pixel): the 8
// channels of a pixel are the 32 contiguous bytes of the channel-blocked layout, a wave (64 consecutive pixels of one
// octet) stores 2 KB contiguous -- the first version (thread = 4 channels, the 16 quads of a pixel side by side) scattered
// every wave store over 32 separate 32-byte pieces and reached 3.2 TB/s.
// ------------------------------------------------------------------------------------------------
// ALLOCT: a thread keeps its pixel's nine input values and walks all channel octets (weights by scalar loads per octet) instead of
// one octet per workgroup (blockIdx.z): the loads, the bounds tests and the p / W of a pixel are done once, not Cout / 8 times.
template <bool ALLOCT>
__global__ __launch_bounds__(256) void k_conv_first(const float *__restrict__ in, const float *__restrict__ w /*[cout][9]*/,
                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                     float *__restrict__ out, int H, int W, int Cout, int64_t n_pix_per_stream,
                                                     unsigned *__restrict__ amax_out) {
    const int s = blockIdx.y;
    const int oct0 = ALLOCT ? 0 : blockIdx.z, oct1 = ALLOCT ? Cout / 8 : oct0 + 1;  // channels 8 oct .. 8 oct + 7
    unsigned mx = 0;             // largest output of this thread (>= 0 after ReLU: bit patterns order like the values)
    const float *ip = in + (int64_t)s * n_pix_per_stream;
    const int npix = (int)n_pix_per_stream;  // H * W of one stream: 32-bit (a 64-bit p / W costs more than the conv)
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        const int h = p / W, x = p - h * W;
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = x + t % 3 - 1;
            v[t] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? ip[(int64_t)hh * W + ww] : 0.0f;
        }
        for (int oct = oct0; oct < oct1; ++oct) {
            const float *wo = w + oct * 72;  // wave-uniform: scalar loads
            float *op = out + (int64_t)s * n_pix_per_stream * Cout + (int64_t)oct * n_pix_per_stream * 8;  // NC8HW8 plane of this octet
            float r[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float a = 0.0f;
#pragma unroll
                for (int t = 0; t < 9; ++t) a = fmaf(v[t], wo[c * 9 + t], a);
                r[c] = fmaxf(fmaf(a, scale[oct * 8 + c], shift[oct * 8 + c]), 0.0f);
            }
            float4 o0, o1;
            o0.x = r[0]; o0.y = r[1]; o0.z = r[2]; o0.w = r[3];
            o1.x = r[4]; o1.y = r[5]; o1.z = r[6]; o1.w = r[7];
            *(float4 *)(op + (int64_t)p * 8) = o0;
            *(float4 *)(op + (int64_t)p * 8 + 4) = o1;
            if (amax_out != nullptr) {
#pragma unroll
                for (int c = 0; c < 8; ++c) mx = max(mx, __float_as_uint(r[c]));
            }
        }
    }
    if (amax_out != nullptr) {  // per-stream maximum for the split-precision layer behind (conv_layout.h): one atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
        if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(amax_out + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_out + s, mx);
    }
}

// ------------------------------------------------------------------------------------------------
// Pooling head (panns.py:262-266): mean over mel, then max over time + mean over time.
// x (S, C/8, H, W, 8) -> feat (S, C)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head(const float *__restrict__ x, float *__restrict__ feat, int H, int W, int C) {
    const int s = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float *p = x + act_off(s, c, 0, 0, C, H, W);  // NC8HW8: pixel stride 8 within the channel block
    float mx = -INFINITY, sum = 0.0f;
    for (int h = 0; h < H; ++h) {
        float rs = 0.0f;
        for (int w = 0; w < W; ++w) rs += p[((int64_t)h * W + w) * 8];
        const float m = rs / (float)W;
        mx = fmaxf(mx, m);
        sum += m;
    }
    feat[(int64_t)s * C + c] = mx + sum / (float)H;
}

// fc_mid / fc_side (panns.py:271-279).  feat (n_cand*channels, K); stream parity picks the layer.
// One workgroup = 8 streams of one kind x 64 outputs; each of its four waves walks a quarter of K (weights are (K, E): a wave
// reads 256 contiguous bytes per k, the 8 stream values of a k are two broadcast 16-byte LDS reads), the quarters are added
// in a fixed order.  (The first version gave a thread all of K for 256 outputs: 128 workgroups, 2 048 dependent steps, 250 us.)
static constexpr int FC_SB = 8;
__global__ __launch_bounds__(256) void k_fc(const float *__restrict__ feat, const float *__restrict__ wt_mid,
                                             const float *__restrict__ b_mid, const float *__restrict__ wt_side,
                                             const float *__restrict__ b_side, float *__restrict__ mid,
                                             float *__restrict__ side, int n_cand, int channels, int K, int E) {
    extern __shared__ __attribute__((aligned(16))) float sf[];  // [K][FC_SB]; afterwards the partial sums [4][64][FC_SB]
    const int kind = blockIdx.z;   // 0 mid, 1 side
    const int c0 = blockIdx.y * FC_SB;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const float *wt = kind == 0 ? wt_mid : wt_side;
    const float *bb = kind == 0 ? b_mid : b_side;
    float *dst = kind == 0 ? mid : side;
    for (int i = threadIdx.x; i < FC_SB * K; i += 256) {
        const int sb = i / K, k = i % K;
        const int cand = c0 + sb;
        sf[k * FC_SB + sb] = cand < n_cand ? feat[((int64_t)cand * channels + kind) * K + k] : 0.0f;
    }
    __syncthreads();
    const int kq = (K + 3) / 4, k0 = ks * kq, k1 = min(K, k0 + kq);
    const int ee = e < E ? e : E - 1;
    f32x4 a0 = (f32x4)(0.0f), a1 = (f32x4)(0.0f);
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const float wv = wt[(int64_t)k * E + ee];
        a0 += *(const f32x4 *)(sf + k * FC_SB) * wv;
        a1 += *(const f32x4 *)(sf + k * FC_SB + 4) * wv;
    }
    __syncthreads();
    *(f32x4 *)(sf + (ks * 64 + lane) * FC_SB) = a0;
    *(f32x4 *)(sf + (ks * 64 + lane) * FC_SB + 4) = a1;
    __syncthreads();
    if (e >= E) return;
    const float bias = bb[e];
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // thread (lane, ks) finishes streams 2 ks, 2 ks + 1
        const int sb = 2 * ks + h;
        const float v = ((sf[(0 * 64 + lane) * FC_SB + sb] + sf[(1 * 64 + lane) * FC_SB + sb]) + sf[(2 * 64 + lane) * FC_SB + sb]) +
                        sf[(3 * 64 + lane) * FC_SB + sb];
        if (c0 + sb < n_cand) dst[(int64_t)(c0 + sb) * E + e] = v + bias;
    }
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

// -cosine_similarity(emb[c], target) of one named embedding (style_transfer.py:544-559), written as
// loss[c] = weight * (-cos) or accumulated onto loss[c]: the generic-metric form of the loss (the dict an
// embed_func returns is walked by the host; the mean over its entries is weight = 1 / n_entries).
__global__ __launch_bounds__(256) void k_neg_cosine(const float *__restrict__ emb, const float *__restrict__ tgt, int E,
                                                     float weight, int accumulate, float *__restrict__ loss) {
    __shared__ float red[4];
    const float *e = emb + (int64_t)blockIdx.x * E;
    float dot = 0.0f, s2 = 0.0f, t2 = 0.0f;
    for (int i = threadIdx.x; i < E; i += 256) {
        const float v = e[i], tv = tgt[i];
        dot = fmaf(v, tv, dot); s2 = fmaf(v, v, s2); t2 = fmaf(tv, tv, t2);
    }
    dot = block_sum_256(dot, red);
    s2 = block_sum_256(s2, red);
    t2 = block_sum_256(t2, red);
    if (threadIdx.x == 0) {
        const float c = dot / (fmaxf(sqrtf(s2), 1e-8f) * fmaxf(sqrtf(t2), 1e-8f));  // torch.cosine_similarity, eps 1e-8
        loss[blockIdx.x] = (accumulate ? loss[blockIdx.x] : 0.0f) + weight * (-c);
    }
}

// ------------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_conv(const float *__restrict__ w, int Cout, int Cin, int ck, float *__restrict__ o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)Cout * Cin * 9;
    if (i >= n) return;
    if (Cin % ck != 0) {  // first layer: [cout][9]
        o[i] = w[i];
        return;
    }
    // destination index i -> (chunk, tap, co, c)
    const int c = (int)(i % ck);
    const int co = (int)((i / ck) % Cout);
    const int tap = (int)((i / ((int64_t)ck * Cout)) % 9);
    const int chunk = (int)(i / ((int64_t)ck * Cout * 9));
    const int ci = chunk * ck + c;
    o[i] = w[((int64_t)co * Cin + ci) * 9 + tap];
}

// Winograd weight transform U = G g G^T (float64, rounded once), packed [cin/8][16][cout][8].
__global__ void k_pack_wino(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin) return;
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    double gk[3][3], t[4][3], u[4][4];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gk[a][b] = (double)w[((int64_t)co * Cin + ci) * 9 + a * 3 + b];
    for (int b = 0; b < 3; ++b) {
        t[0][b] = gk[0][b];
        t[1][b] = 0.5 * (gk[0][b] + gk[1][b] + gk[2][b]);
        t[2][b] = 0.5 * (gk[0][b] - gk[1][b] + gk[2][b]);
        t[3][b] = gk[2][b];
    }
    for (int a = 0; a < 4; ++a) {
        u[a][0] = t[a][0];
        u[a][1] = 0.5 * (t[a][0] + t[a][1] + t[a][2]);
        u[a][2] = 0.5 * (t[a][0] - t[a][1] + t[a][2]);
        u[a][3] = t[a][2];
    }
    const int chunk = ci / WK, c8 = ci % WK;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            o[(((int64_t)chunk * 16 + a * 4 + b) * Cout + co) * WK + c8] = (float)u[a][b];
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
template <int WM, int WN, int TW, bool POOL>
static int launch_conv(const float *in, const float *wpk, const float *scale, const float *shift, float *out,
                          const ConvShape &g0, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN, TH = BM / TW, PW = TW + 2;
    ConvGeom g{};
    g.S = g0.S; g.H = g0.H; g.W = g0.W; g.Cin = g0.Cin; g.Cout = g0.Cout;
    g.Ho = g.H / 2; g.Wo = g.W / 2;
    g.Heff = POOL ? 2 * g.Ho : g.H;
    g.VR = (int64_t)g.S * g.Heff;
    g.IVR = (int64_t)g.S * g.H;
    g.n_col_tiles = (g.W + TW - 1) / TW;
    const int64_t n_row_tiles = (g.VR + TH - 1) / TH;
    STITO_REQUIRE(n_row_tiles * g.n_col_tiles < (1 << 30), STITO_E_UNSUPPORTED, "conv: too many tiles");
    g.n_m_tiles = (int)(n_row_tiles * g.n_col_tiles);
    g.PR = TH + 2 + ((TH - 1) / g.Heff + 1) * (g.H - g.Heff);
    g.na_i = (g.PR * PW + 63) / 64;
    STITO_REQUIRE(g.na_i <= 8, STITO_E_UNSUPPORTED, "conv tile: halo patch of %d pixels exceeds the staging budget", g.PR * PW);
    const size_t lds = (size_t)2 * (9 * BN * 4 + g.na_i * 256) * sizeof(float);
    auto kern = k_conv3x3<WM, WN, TW, POOL>;
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (int64_t)g.n_m_tiles * (g.Cout / BN);
    STITO_REQUIRE(blocks < (1ll << 31), STITO_E_UNSUPPORTED, "conv: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WM * WN), lds, st, in, wpk, scale, shift, out, g);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

template <int WM, int WN, bool POOL>
static int launch_conv_tw(const float *in, const float *wpk, const float *scale, const float *shift, float *out,
                          const ConvShape &g, hipStream_t st) {
    if (g.W >= 16) return launch_conv<WM, WN, 16, POOL>(in, wpk, scale, shift, out, g, st);
    if (g.W >= 8) return launch_conv<WM, WN, 8, POOL>(in, wpk, scale, shift, out, g, st);
    return launch_conv<WM, WN, 4, POOL>(in, wpk, scale, shift, out, g, st);
}


static thread_local long long *g_wino_trace = nullptr;  // stito_debug_wino_trace (per host thread, like stito_last_error)

template <int TTW, bool POOL>
static bool wino_geometry(const ConvShape &c, WinoGeom &g, size_t &lds, int64_t &blocks) {
    constexpr int TTH = 64 / TTW;
    g = WinoGeom{};
    g.S = c.S; g.H = c.H; g.W = c.W; g.Cin = c.Cin; g.Cout = c.Cout;
    g.Ho = c.H / 2; g.Wo = c.W / 2;
    g.TR = POOL ? g.Ho : (c.H + 1) / 2;
    g.TC = POOL ? g.Wo : (c.W + 1) / 2;
    if (g.TR < 1 || g.TC < 1) return false;
    g.VTR = (int64_t)g.S * g.TR;
    g.n_col_blocks = (g.TC + TTW - 1) / TTW;
    const int64_t nrb = (g.VTR + TTH - 1) / TTH;
    blocks = nrb * g.n_col_blocks * (g.Cout / 64);
    if (nrb * g.n_col_blocks >= (1 << 30) || blocks >= (1ll << 31)) return false;
    if ((int64_t)g.S * g.H >= (1ll << 31) - 64 || g.VTR >= (1ll << 31) - 64) return false;  // the kernel's row arithmetic is 32-bit
    g.n_m_blocks = (int)(nrb * g.n_col_blocks);
    // patch rows: 2 per tile row + 2 halo, plus the rows skipped at every stream boundary a block can straddle
    g.PR = 2 * TTH + 2 + ((TTH - 1) / g.TR + 1) * (g.H - 2 * g.TR > 0 ? g.H - 2 * g.TR : 0);
    g.pa_i = (g.PR * (2 * TTW + 2) * 2 + 63) / 64;
    lds = ((size_t)2 * (16 * 64 * WK * 2) + 2 * (g.pa_i * 256 + 512)) * sizeof(float);
    // pa_i >= 8: the epilogue's exchange (8 planes x 64 tiles x 72 floats) spills 16 KB into the patch buffers
    if (g.PR * (2 * TTW + 2) * 2 > 2 * WINO8_THREADS) return false;  // two patch float4 per thread per chunk
    return g.pa_i >= 8 && g.pa_i <= 16 && (2 * TTW + 2) * 8 <= 256 && lds <= 160 * 1024;
}

template <int TTW, bool POOL>
static int launch_wino(const float *in, const float *upk, const float *scale, const float *shift, float *out,
                       const ConvShape &c, hipStream_t st) {
    WinoGeom g;
    size_t lds;
    int64_t blocks;
    STITO_REQUIRE((wino_geometry<TTW, POOL>(c, g, lds, blocks)), STITO_E_UNSUPPORTED,
                  "conv (winograd): %dx%d map does not fit the LDS-resident halo patch", c.H, c.W);
    g.trace = g_wino_trace;
    auto kern8 = g_wino_trace ? k_conv_wino8<TTW, POOL, true> : k_conv_wino8<TTW, POOL, false>;
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern8, dim3((unsigned)blocks), dim3(WINO8_THREADS), lds, st, in, upk, scale, shift, out, g);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

static int wino_ttw(const ConvShape &c, bool pool) {
    const int tc = pool ? c.W / 2 : (c.W + 1) / 2;
    return tc >= 8 ? 8 : (tc >= 4 ? 4 : 2);
}

static bool wino_supported(const ConvShape &c, bool pool) {
    if (c.Cin % WK != 0 || c.Cout % 64 != 0) return false;
    WinoGeom g;
    size_t lds;
    int64_t blocks;
    const int ttw = wino_ttw(c, pool);
    if (pool) return ttw == 8 ? wino_geometry<8, true>(c, g, lds, blocks) : ttw == 4 ? wino_geometry<4, true>(c, g, lds, blocks) : wino_geometry<2, true>(c, g, lds, blocks);
    return ttw == 8 ? wino_geometry<8, false>(c, g, lds, blocks) : ttw == 4 ? wino_geometry<4, false>(c, g, lds, blocks) : wino_geometry<2, false>(c, g, lds, blocks);
}

template <bool POOL>
static int launch_wino_tw(const float *in, const float *upk, const float *scale, const float *shift, float *out,
                          const ConvShape &c, hipStream_t st) {
    const int ttw = wino_ttw(c, POOL);
    if (ttw == 8) return launch_wino<8, POOL>(in, upk, scale, shift, out, c, st);
    if (ttw == 4) return launch_wino<4, POOL>(in, upk, scale, shift, out, c, st);
    return launch_wino<2, POOL>(in, upk, scale, shift, out, c, st);
}

static int conv_first(const float *in, const float *w, const float *scale, const float *shift, float *out, int S, int H,
                      int W, int Cout, hipStream_t st, unsigned *amax_out = nullptr) {
    STITO_REQUIRE(Cout % 8 == 0 && Cout / 8 <= 65535, STITO_E_UNSUPPORTED, "first conv: cout %d", Cout);
    const int64_t npix = (int64_t)H * W;
    STITO_REQUIRE(npix < (1ll << 30), STITO_E_UNSUPPORTED, "first conv: %dx%d map too large", H, W);
    STITO_REQUIRE(S <= 65535, STITO_E_UNSUPPORTED, "first conv: %d streams per launch", S);
    int64_t gx = (npix + 255) / 256;
    // default: one thread per pixel walks all channel octets (2.80 -> 2.14 ms at 512 streams: the kernel was bound by the
    // instructions in front of its stores, not by the stores); STITO_CONV_FIRST_ALLOCT=0 = one octet per workgroup, for A / B runs
    static const bool alloct = [] { const char *e = getenv("STITO_CONV_FIRST_ALLOCT"); return e ? atoi(e) != 0 : true; }();
    const int64_t nz = alloct ? 1 : Cout / 8;
    const int64_t cap = (256 * 64 + (int64_t)S * nz - 1) / ((int64_t)S * nz);  // ~64 workgroups per CU in total
    gx = gx < cap ? gx : (cap < 1 ? 1 : cap);
    if (alloct) hipLaunchKernelGGL(k_conv_first<true>, dim3((unsigned)gx, S, 1), dim3(256), 0, st, in, w, scale, shift, out, H, W, Cout, npix, amax_out);
    else hipLaunchKernelGGL(k_conv_first<false>, dim3((unsigned)gx, S, Cout / 8), dim3(256), 0, st, in, w, scale, shift, out, H, W, Cout, npix, amax_out);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

}  // namespace stito

using namespace stito;

static bool wino_ok(int cout, int cin) { return cin % WK == 0 && cout % 64 == 0; }

extern "C" size_t stito_cnn14_packed_conv_floats(int cout, int cin, int algo) {
    if (algo == 6 || algo == 7) return 0;  // retired (see stito_hip.h)
    if (algo == STITO_CONV_WINOGRAD_F2_REG) return wino23r_packed_floats(cout, cin);
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT || algo == STITO_CONV_WINOGRAD_F4_SPLIT2 || algo == STITO_CONV_WINOGRAD_F4_SPLIT3) return wino43_split_packed_floats(cout, cin);
    return (size_t)cout * cin * ((algo == STITO_CONV_WINOGRAD_F4 || algo == STITO_CONV_WINOGRAD_F4_PRE) ? 36 : algo == STITO_CONV_WINOGRAD ? 16 : 9);
}

extern "C" int stito_cnn14_pack_conv(const float *w_oihw_dev, int cout, int cin, int algo, float *packed_dev, void *stream) {
    STITO_REQUIRE(algo != 6 && algo != 7, STITO_E_UNSUPPORTED, "conv: algorithm %d was retired in ABI version 9", algo);
    if (algo == STITO_CONV_WINOGRAD_F2_REG) return pack_wino23r(w_oihw_dev, cout, cin, packed_dev, (hipStream_t)stream);
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT || algo == STITO_CONV_WINOGRAD_F4_SPLIT2 || algo == STITO_CONV_WINOGRAD_F4_SPLIT3)
        return pack_wino43_split(w_oihw_dev, cout, cin, packed_dev, algo == STITO_CONV_WINOGRAD_F4_SPLIT3 ? 2 : algo == STITO_CONV_WINOGRAD_F4_SPLIT2 ? 1 : 0,
                                 (hipStream_t)stream);
    if (algo == STITO_CONV_WINOGRAD_F4 || algo == STITO_CONV_WINOGRAD_F4_PRE) {  // one packing for both
        STITO_REQUIRE(wino_ok(cout, cin), STITO_E_UNSUPPORTED, "conv (winograd): cin %d / cout %d", cin, cout);
        return pack_wino43(w_oihw_dev, cout, cin, packed_dev, (hipStream_t)stream);
    }
    if (algo == STITO_CONV_WINOGRAD) {
        STITO_REQUIRE(wino_ok(cout, cin), STITO_E_UNSUPPORTED, "conv (winograd): cin %d / cout %d", cin, cout);
        const int64_t n = (int64_t)cout * cin;
        hipLaunchKernelGGL(k_pack_wino, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw_dev, cout, cin, packed_dev);
    } else {
        const int64_t n = (int64_t)cout * cin * 9;
        hipLaunchKernelGGL(k_pack_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw_dev, cout, cin, CK, packed_dev);
    }
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" size_t stito_cnn14_packed_conv1_f2reg_floats(void) { return conv1_f2reg_packed_floats(); }

extern "C" int stito_cnn14_pack_conv1_f2reg(const float *w_oihw_dev, const float *scale_dev, const float *shift_dev, int c1,
                                            float *packed_dev, void *stream) {
    return pack_conv1_f2reg(w_oihw_dev, scale_dev, shift_dev, c1, packed_dev, (hipStream_t)stream);
}

extern "C" int stito_conv_block1_f2reg_supported(int n, int H, int W, int c1, int cout, int pool) {
    if (n <= 0 || H <= 0 || W <= 0) return 0;
    return wino23r_fused1_supported(ConvShape{n, H, W, c1, cout}, pool != 0) ? 1 : 0;
}

extern "C" size_t stito_conv_block1_f2reg_workspace_bytes(int n, int H, int W, int c1, int cout, int pool) {
    if (n <= 0 || H <= 0 || W <= 0) return 0;
    return wino23r_fused1_workspace_bytes(ConvShape{n, H, W, c1, cout}, pool != 0);
}

extern "C" int stito_conv_block1_f2reg(const float *x_dev, const float *packed_w1_dev, const float *packed_w2_dev,
                                       const float *scale2_dev, const float *shift2_dev, float *out_dev, int n, int H, int W, int c1,
                                       int cout, int pool, void *workspace_dev, size_t workspace_bytes, void *stream,
                                       unsigned *amax_out_dev) {
    STITO_REQUIRE(n > 0 && H > 0 && W > 0, STITO_E_INVALID, "conv: empty input");
    return launch_wino23r_fused1(x_dev, packed_w1_dev, packed_w2_dev, scale2_dev, shift2_dev, out_dev, ConvShape{n, H, W, c1, cout},
                                 pool != 0, workspace_dev, workspace_bytes, (hipStream_t)stream, amax_out_dev);
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

extern "C" int stito_debug_wino_trace(long long *buf_dev) {
    g_wino_trace = buf_dev;
    return STITO_OK;
}

extern "C" int stito_conv3x3_supported(int n, int H, int W, int cin, int cout, int pool, int algo) {
    if (n <= 0 || H <= 0 || W <= 0 || cout % 4 != 0) return 0;
    if (pool && (H < 2 || W < 2)) return 0;
    if (algo == STITO_CONV_WINOGRAD_F4_PRE && (cout % 256 != 0 || (cout >= 1024 && cout % 512 != 0))) return 0;  // its workgroup order deals channel tiles in fours / eights
    if (algo == 6 || algo == 7) return 0;  // retired
    if (algo == STITO_CONV_WINOGRAD_F2_REG) return wino23r_supported(ConvShape{n, H, W, cin, cout}, pool != 0) ? 1 : 0;
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT3) return wino43_split3_supported(ConvShape{n, H, W, cin, cout}, pool != 0) ? 1 : 0;
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT || algo == STITO_CONV_WINOGRAD_F4_SPLIT2)
        return (cout < 1024 || cout % 512 == 0) && wino43_split_supported(ConvShape{n, H, W, cin, cout}, pool != 0) ? 1 : 0;
    if (algo == STITO_CONV_WINOGRAD_F4 || algo == STITO_CONV_WINOGRAD_F4_PRE) return wino43_supported(ConvShape{n, H, W, cin, cout}, pool != 0) ? 1 : 0;
    if (algo == STITO_CONV_WINOGRAD) return wino_supported(ConvShape{n, H, W, cin, cout}, pool != 0) ? 1 : 0;
    if (cin == 1) return (!pool && cout % 8 == 0) ? 1 : 0;
    return (cin % 8 == 0 && cout % 64 == 0) ? 1 : 0;  // channel-blocked activations: 8 channels per block
}

// FLOPs of the MFMA instructions one launch issues, tile padding included (bench.py "roofline": the hardware-side
// numerator, <= peak by construction; the algorithmic count 2*9*cin*cout*H*W is larger for the Winograd kernels)
extern "C" double stito_conv3x3_issued_flops(int n, int H, int W, int cin, int cout, int pool, int algo) {
    if (!stito_conv3x3_supported(n, H, W, cin, cout, pool, algo) || cin % 8 != 0) return 0.0;
    ConvShape c{n, H, W, cin, cout};
    if (algo == STITO_CONV_WINOGRAD_F2_REG) return wino23r_issued_flops(c, pool != 0);
    if (algo == STITO_CONV_WINOGRAD_F4 || algo == STITO_CONV_WINOGRAD_F4_PRE) return wino43_issued_flops(c, pool != 0);
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT) return 3.0 * wino43_issued_flops(c, pool != 0);  // hi hi' + hi lo' + lo hi' on the f16 pipe
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT2) return wino43_split2_issued_flops(c, pool != 0);
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT3) return wino43_split3_issued_flops(c, pool != 0);
    if (algo == STITO_CONV_WINOGRAD) {
        WinoGeom g;
        size_t lds;
        int64_t blocks = 0;
        const int ttw = wino_ttw(c, pool != 0);
        const bool ok = pool ? (ttw == 8 ? wino_geometry<8, true>(c, g, lds, blocks) : ttw == 4 ? wino_geometry<4, true>(c, g, lds, blocks) : wino_geometry<2, true>(c, g, lds, blocks))
                             : (ttw == 8 ? wino_geometry<8, false>(c, g, lds, blocks) : ttw == 4 ? wino_geometry<4, false>(c, g, lds, blocks) : wino_geometry<2, false>(c, g, lds, blocks));
        return ok ? 2.0 * (double)blocks * 64.0 * 64.0 * 16.0 * cin : 0.0;
    }
    // direct: launch_conv's tiling -- (BM x BN) = (128 x 128) when cout % 128 == 0, else (256 x 64); TW by map width
    const int BM = cout % 128 == 0 ? 128 : 256, BN = cout % 128 == 0 ? 128 : 64;
    const int TW = W >= 16 ? 16 : (W >= 8 ? 8 : 4), TH = BM / TW;
    const int Heff = pool ? 2 * (H / 2) : H;
    const int64_t n_row_tiles = ((int64_t)n * Heff + TH - 1) / TH, n_col_tiles = (W + TW - 1) / TW;
    return 2.0 * (double)(n_row_tiles * n_col_tiles) * (cout / BN) * BM * BN * 9.0 * cin;
}

extern "C" int stito_conv3x3_bn_relu(const float *in_dev, const float *packed_w_dev, const float *scale_dev,
                                     const float *shift_dev, float *out_dev, int n, int H, int W, int cin, int cout,
                                     int pool, int algo, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(n > 0 && H > 0 && W > 0, STITO_E_INVALID, "conv: empty input");
    STITO_REQUIRE(algo != STITO_CONV_WINOGRAD_F4_PRE && algo != STITO_CONV_WINOGRAD_F4_SPLIT && algo != STITO_CONV_WINOGRAD_F4_SPLIT2 && algo != STITO_CONV_WINOGRAD_F4_SPLIT3 &&
                  algo != STITO_CONV_WINOGRAD_F2_REG, STITO_E_WORKSPACE,
                  "conv: STITO_CONV_WINOGRAD_F4_PRE / _F4_SPLIT* / STITO_CONV_WINOGRAD_F2_REG need stito_conv3x3_bn_relu_ws");
    if (cin % 8 != 0) {
        STITO_REQUIRE(cin == 1 && !pool, STITO_E_UNSUPPORTED, "conv: cin=%d (only 1 or a multiple of 8)", cin);
        return conv_first(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, n, H, W, cout, st);
    }
    STITO_REQUIRE(cout % 64 == 0, STITO_E_UNSUPPORTED, "conv: cout=%d must be a multiple of 64", cout);
    STITO_REQUIRE(!pool || (H >= 2 && W >= 2), STITO_E_INVALID, "Given input size: (%dx%dx%d). Output size is too small", cout, H, W);
    ConvShape g{n, H, W, cin, cout};
    if (algo == STITO_CONV_WINOGRAD_F4) {
        STITO_REQUIRE(wino_ok(cout, cin), STITO_E_UNSUPPORTED, "conv (winograd): cin %d / cout %d", cin, cout);
        return launch_wino43(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, pool != 0, g_wino_trace, st);
    }
    if (algo == STITO_CONV_WINOGRAD) {
        STITO_REQUIRE(wino_ok(cout, cin), STITO_E_UNSUPPORTED, "conv (winograd): cin %d / cout %d", cin, cout);
        return pool ? launch_wino_tw<true>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st)
                    : launch_wino_tw<false>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st);
    }
    if (cout % 128 == 0) {
        return pool ? launch_conv_tw<2, 2, true>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st)
                    : launch_conv_tw<2, 2, false>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st);
    }
    return pool ? launch_conv_tw<4, 1, true>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st)
                : launch_conv_tw<4, 1, false>(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, g, st);
}

extern "C" size_t stito_conv3x3_workspace_bytes(int n, int H, int W, int cin, int cout, int pool, int algo) {
    if ((algo != STITO_CONV_WINOGRAD_F4_PRE && algo != STITO_CONV_WINOGRAD_F4_SPLIT && algo != STITO_CONV_WINOGRAD_F4_SPLIT2 && algo != STITO_CONV_WINOGRAD_F4_SPLIT3 &&
         algo != STITO_CONV_WINOGRAD_F2_REG) || !stito_conv3x3_supported(n, H, W, cin, cout, pool, algo)) return 0;
    if (algo == STITO_CONV_WINOGRAD_F2_REG) return wino23r_workspace_bytes(ConvShape{n, H, W, cin, cout}, pool != 0);
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT) return wino43_split_workspace_bytes(ConvShape{n, H, W, cin, cout}, pool != 0);
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT2) return wino43_split2_workspace_bytes(ConvShape{n, H, W, cin, cout}, pool != 0);
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT3) return wino43_split3_workspace_bytes(ConvShape{n, H, W, cin, cout}, pool != 0);
    return wino43_pre_workspace_bytes(ConvShape{n, H, W, cin, cout}, pool != 0);
}

// amax_in / amax_out: see conv_layout.h (per-stream output maxima handed from one layer to the next inside the trunk)
static int conv3x3_ws(const float *in_dev, const float *packed_w_dev, const float *scale_dev, const float *shift_dev, float *out_dev,
                      int n, int H, int W, int cin, int cout, int pool, int algo, void *workspace_dev, size_t workspace_bytes,
                      void *stream, const unsigned *amax_in, unsigned *amax_out) {
    STITO_REQUIRE(algo != 6 && algo != 7, STITO_E_UNSUPPORTED, "conv: algorithm %d was retired in ABI version 9", algo);
    if (algo == STITO_CONV_WINOGRAD_F2_REG) {
        STITO_REQUIRE(n > 0 && H > 0 && W > 0, STITO_E_INVALID, "conv: empty input");
        STITO_REQUIRE(stito_conv3x3_supported(n, H, W, cin, cout, pool, algo), STITO_E_UNSUPPORTED,
                      "conv (winograd F(2x2,3x3), register-resident weights): %dx%d map, %d -> %d channels not covered (cin == 64, cout %% 64)", H, W, cin, cout);
        STITO_REQUIRE(!pool || (H >= 2 && W >= 2), STITO_E_INVALID, "Given input size: (%dx%dx%d). Output size is too small", cout, H, W);
        return launch_wino23r(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, ConvShape{n, H, W, cin, cout}, pool != 0, workspace_dev,
                              workspace_bytes, (hipStream_t)stream, amax_in, amax_out);
    }
    if (algo == STITO_CONV_DIRECT && cin == 1 && amax_out != nullptr && !pool && n > 0 && H > 0 && W > 0)
        return conv_first(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, n, H, W, cout, (hipStream_t)stream, amax_out);
    if (algo != STITO_CONV_WINOGRAD_F4_PRE && algo != STITO_CONV_WINOGRAD_F4_SPLIT && algo != STITO_CONV_WINOGRAD_F4_SPLIT2 && algo != STITO_CONV_WINOGRAD_F4_SPLIT3) {
        if (algo == STITO_CONV_WINOGRAD_F4 && amax_out != nullptr && n > 0 && H > 0 && W > 0 && cin % 8 == 0 && cout % 64 == 0 &&
            wino_ok(cout, cin) && (!pool || (H >= 2 && W >= 2)))
            return launch_wino43(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, ConvShape{n, H, W, cin, cout}, pool != 0, g_wino_trace,
                                 (hipStream_t)stream, amax_out);
        STITO_REQUIRE(amax_out == nullptr, STITO_E_INVALID, "conv: this algorithm does not report output maxima");
        return stito_conv3x3_bn_relu(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, n, H, W, cin, cout, pool, algo, stream);
    }
    STITO_REQUIRE(n > 0 && H > 0 && W > 0, STITO_E_INVALID, "conv: empty input");
    if (algo == STITO_CONV_WINOGRAD_F4_SPLIT || algo == STITO_CONV_WINOGRAD_F4_SPLIT2 || algo == STITO_CONV_WINOGRAD_F4_SPLIT3) {
        STITO_REQUIRE(stito_conv3x3_supported(n, H, W, cin, cout, pool, algo), STITO_E_UNSUPPORTED,
                      "conv (split-precision winograd F(4x4,3x3)): %dx%d map, %d -> %d channels not covered (cin %% 64, cout %% 256)", H, W, cin, cout);
        STITO_REQUIRE(!pool || (H >= 2 && W >= 2), STITO_E_INVALID, "Given input size: (%dx%dx%d). Output size is too small", cout, H, W);
        if (algo == STITO_CONV_WINOGRAD_F4_SPLIT3)
            return launch_wino43_split3(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, ConvShape{n, H, W, cin, cout}, pool != 0,
                                        workspace_dev, workspace_bytes, (hipStream_t)stream, amax_in, amax_out);
        if (algo == STITO_CONV_WINOGRAD_F4_SPLIT2)
            return launch_wino43_split2(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, ConvShape{n, H, W, cin, cout}, pool != 0,
                                        workspace_dev, workspace_bytes, (hipStream_t)stream, amax_in, amax_out);
        return launch_wino43_split(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, ConvShape{n, H, W, cin, cout}, pool != 0,
                                   workspace_dev, workspace_bytes, (hipStream_t)stream, amax_in, amax_out);
    }
    STITO_REQUIRE(cin % 8 == 0 && cout % 64 == 0 && wino_ok(cout, cin), STITO_E_UNSUPPORTED, "conv (winograd): cin %d / cout %d", cin, cout);
    STITO_REQUIRE(stito_conv3x3_supported(n, H, W, cin, cout, pool, algo), STITO_E_UNSUPPORTED,
                  "conv (winograd F(4x4,3x3), hoisted input transform): %dx%d map, %d -> %d channels not covered (cout must be a multiple of 256)", H, W, cin, cout);
    STITO_REQUIRE(!pool || (H >= 2 && W >= 2), STITO_E_INVALID, "Given input size: (%dx%dx%d). Output size is too small", cout, H, W);
    return launch_wino43_pre(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, ConvShape{n, H, W, cin, cout}, pool != 0,
                             (float *)workspace_dev, workspace_bytes, (hipStream_t)stream, amax_out);
}

extern "C" int stito_conv3x3_bn_relu_ws(const float *in_dev, const float *packed_w_dev, const float *scale_dev,
                                        const float *shift_dev, float *out_dev, int n, int H, int W, int cin, int cout,
                                        int pool, int algo, void *workspace_dev, size_t workspace_bytes, void *stream) {
    return conv3x3_ws(in_dev, packed_w_dev, scale_dev, shift_dev, out_dev, n, H, W, cin, cout, pool, algo, workspace_dev, workspace_bytes,
                      stream, nullptr, nullptr);
}

static void cnn14_dims(int64_t T, int M, int H[7], int W[7]) {
    H[0] = (int)T; W[0] = M;
    for (int b = 1; b <= 5; ++b) { H[b] = H[b - 1] / 2; W[b] = W[b - 1] / 2; }
    H[6] = H[5]; W[6] = W[5];
}

// transformed-input workspace: the largest need among the layers that run STITO_CONV_WINOGRAD_F4_PRE
static size_t cnn14_pre_bytes(const stito_cnn14_weights *w, int n_streams, const int H[7], const int W[7]) {
    size_t v = 0;
    for (int i = 0; i < STITO_CNN14_NUM_CONVS; ++i) {
        const int algo = w->conv_wino_algo[i];
        if (w->conv_wino_dev[i] == nullptr || (algo != STITO_CONV_WINOGRAD_F4_PRE && algo != STITO_CONV_WINOGRAD_F4_SPLIT && algo != STITO_CONV_WINOGRAD_F4_SPLIT2 && algo != STITO_CONV_WINOGRAD_F4_SPLIT3 &&
                                               algo != STITO_CONV_WINOGRAD_F2_REG)) continue;
        const int blk = i / 2, j = i % 2;
        const int ci = j == 0 ? w->channels[blk] : w->channels[blk + 1], pool = (j == 1 && blk < 5) ? 1 : 0;
        size_t need = stito_conv3x3_workspace_bytes(n_streams, H[blk], W[blk], ci, w->channels[blk + 1], pool, algo);
        if (w->conv_alt_dev[i] != nullptr) {
            const size_t alt = stito_conv3x3_workspace_bytes(n_streams, H[blk], W[blk], ci, w->channels[blk + 1], pool, w->conv_alt_algo[i]);
            need = alt > need ? alt : need;
        }
        v = need > v ? need : v;
    }
    if (w->conv1_f2reg_w_dev != nullptr && w->channels[0] == 1) {   // conv_block1 in one launch: per-stream scales of the log-mel operand
        const size_t need = stito_conv_block1_f2reg_workspace_bytes(n_streams, H[0], W[0], w->channels[1], w->channels[1], 1);
        v = need > v ? need : v;
    }
    return align_up(v, 256);
}

// ---- depth-first schedule of a run of conv layers over chunks of streams (ABI version 10) ----
// stito_cnn14_weights.chunk_streams > 0: the convs chunk_first_conv .. chunk_last_conv run chunk by chunk -- every layer of the run on
// the first chunk_streams streams, then on the next ones -- with the maps between them in two chunk-sized scratch buffers that every
// chunk reuses (and the transformed-input workspace at the same addresses), so that a chunk's X -> transform -> V -> conv hand-offs
// can stay inside the 256 MiB Infinity Cache instead of making a round trip through HBM.  A stream's results do not depend on
// which launch it rides in (the kernels never mix streams), so the schedule cannot be seen in the outputs (tested bitwise).
struct ChunkPlan {
    int first = -1, last = -1, streams = 0;   // first < 0: no chunked run
    size_t scratch_floats = 0;                // per scratch buffer (two of them)
};

static size_t conv_out_floats_per_stream(const stito_cnn14_weights *w, const int H[7], const int W[7], int i) {
    const int blk = i / 2, j = i % 2;
    return (j == 1 && blk < 5) ? (size_t)H[blk + 1] * W[blk + 1] * w->channels[blk + 1] : (size_t)H[blk] * W[blk] * w->channels[blk + 1];
}

static ChunkPlan cnn14_chunk_plan(const stito_cnn14_weights *w, int n_streams, const int H[7], const int W[7], bool fuse1r) {
    ChunkPlan p;
    int first = w->chunk_first_conv, last = w->chunk_last_conv;
    const int sc = w->chunk_streams;
    if (sc <= 0 || sc >= n_streams || first < 0 || last >= STITO_CNN14_NUM_CONVS || first > last) return p;
    if (fuse1r && first == 1) first = 0;            // conv_block1 is one launch: both convs or neither
    if (fuse1r && last == 0) last = 1;
    if (!fuse1r && first == 0 && w->channels[0] % 8 != 0 && last == 0) return p;   // a lone first conv has nothing to hand over
    if (first == last) {                            // a single layer: its transformed input is what stays in the cache; no scratch maps
        p.first = first; p.last = last; p.streams = sc;
        return p;
    }
    // the run's output lands in the buffer the layer-by-layer schedule would use (actA for a block's first conv, actB for its
    // second); the run's input must not be overwritten by an earlier chunk's output: it is, when both live in actB, unless a
    // stream's output is no larger than its input (chunk c's output then ends before chunk c + 1's input starts)
    if (last % 2 == 1 && first % 2 == 0 && first > 0) {
        const size_t in_ps = conv_out_floats_per_stream(w, H, W, first - 1), out_ps = conv_out_floats_per_stream(w, H, W, last);
        if (out_ps > in_ps) return p;
    }
    if (last % 2 == 0 && first % 2 == 1) return p;  // input and output both in actA (mid-block start): not scheduled
    size_t m = 0;
    for (int i = first; i < last; ++i) {
        if (fuse1r && i == 0) continue;             // never materialised
        const size_t f = conv_out_floats_per_stream(w, H, W, i);
        m = f > m ? f : m;
    }
    p.first = first; p.last = last; p.streams = sc;
    p.scratch_floats = m * (size_t)sc;
    return p;
}

static bool cnn14_fuse1r(const stito_cnn14_weights *w, int S, const int H[7], const int W[7]) {
    return w->conv1_f2reg_w_dev != nullptr && w->conv_wino_dev[1] != nullptr && w->conv_wino_algo[1] == STITO_CONV_WINOGRAD_F2_REG &&
           w->channels[0] == 1 && stito_conv_block1_f2reg_supported(S, H[0], W[0], w->channels[1], w->channels[1], 1);
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
    const ChunkPlan plan = cnn14_chunk_plan(w, n_streams, H, W, cnn14_fuse1r(w, n_streams, H, W));
    return align_up(a * 4, 256) + align_up(b * 4, 256) + align_up(feat * 4, 256) + cnn14_pre_bytes(w, n_streams, H, W) +
           STITO_CNN14_NUM_CONVS * align_up((size_t)n_streams * sizeof(unsigned), 256) + 256 +  // + per-stream output maxima, one buffer per conv (split-precision layers)
           2 * align_up(plan.scratch_floats * 4, 256);                                         // + the chunked run's two scratch maps
}

// ---- optional launch timing for bench.py: HIP events on the launch stream around the MFMA convs ----
namespace {
struct ConvTiming {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;  // created on demand, reused
    std::vector<int> conv;                                // conv index of the launch (the second conv's for conv_block1 in one launch)
    size_t used = 0;
};
thread_local ConvTiming g_conv_timing;  // one host thread drives one GPU: the state is per thread, like stito_last_error
}  // namespace

extern "C" int stito_conv_timing_enable(int on) {
    g_conv_timing.on = on != 0;
    return STITO_OK;
}

extern "C" int stito_conv_timing_read(double *total_ms, int *n_launches) {
    STITO_REQUIRE(total_ms != nullptr && n_launches != nullptr, STITO_E_INVALID, "null output");
    double tot = 0.0;
    for (size_t i = 0; i < g_conv_timing.used; ++i) {
        float ms = 0.f;
        STITO_HIP_CHECK(hipEventSynchronize(g_conv_timing.pool[i].second));
        STITO_HIP_CHECK(hipEventElapsedTime(&ms, g_conv_timing.pool[i].first, g_conv_timing.pool[i].second));
        tot += ms;
    }
    *total_ms = tot;
    *n_launches = (int)g_conv_timing.used;
    g_conv_timing.used = 0;
    return STITO_OK;
}

extern "C" int stito_conv_timing_read_each(double *ms_each, int cap, int *n_launches) {
    return stito_conv_timing_read_tagged(ms_each, nullptr, cap, n_launches);
}

extern "C" int stito_conv_timing_read_tagged(double *ms_each, int *conv_each, int cap, int *n_launches) {
    STITO_REQUIRE(ms_each != nullptr && n_launches != nullptr && cap >= 0, STITO_E_INVALID, "null output");
    for (size_t i = 0; i < g_conv_timing.used; ++i) {
        float ms = 0.f;
        STITO_HIP_CHECK(hipEventSynchronize(g_conv_timing.pool[i].second));
        STITO_HIP_CHECK(hipEventElapsedTime(&ms, g_conv_timing.pool[i].first, g_conv_timing.pool[i].second));
        if ((int)i < cap) {
            ms_each[i] = ms;
            if (conv_each != nullptr) conv_each[i] = g_conv_timing.conv[i];
        }
    }
    *n_launches = (int)g_conv_timing.used;
    g_conv_timing.used = 0;
    return STITO_OK;
}

namespace {
struct TimedLaunch {   // brackets one conv launch with events when the bench asked for them
    hipStream_t st;
    bool on;
    TimedLaunch(hipStream_t s, bool enable) : st(s), on(enable) {}
    int begin(int conv) {
        if (!on) return STITO_OK;
        if (g_conv_timing.used == g_conv_timing.pool.size()) {
            hipEvent_t e0, e1;
            STITO_HIP_CHECK(hipEventCreate(&e0));
            STITO_HIP_CHECK(hipEventCreate(&e1));
            g_conv_timing.pool.emplace_back(e0, e1);
            g_conv_timing.conv.push_back(0);
        }
        g_conv_timing.conv[g_conv_timing.used] = conv;
        STITO_HIP_CHECK(hipEventRecord(g_conv_timing.pool[g_conv_timing.used].first, st));
        return STITO_OK;
    }
    int end() {
        if (!on) return STITO_OK;
        STITO_HIP_CHECK(hipEventRecord(g_conv_timing.pool[g_conv_timing.used++].second, st));
        return STITO_OK;
    }
};

struct Trunk {
    const stito_cnn14_weights *w;
    int H[7], W[7];
    void *stream;
    void *vbuf;
    size_t vbytes;
    unsigned *amax_all;
    size_t amax_stride;
    int n_cus;
    bool fuse1r;

    unsigned *amax_of(int conv, int s0) const { return (unsigned *)((char *)amax_all + amax_stride * conv) + s0; }

    // does conv i + 1 run a kernel that scales its transformed input by per-stream maxima of conv i's output?
    bool next_wants_amax(int i, int S) const {
        if (i + 1 >= STITO_CNN14_NUM_CONVS) return false;
        const int nb = (i + 1) / 2, nj = (i + 1) % 2;
        const int nci = nj == 0 ? w->channels[nb] : w->channels[nb + 1], npool = (nj == 1 && nb < 5) ? 1 : 0;
        const int nalgo = w->conv_wino_algo[i + 1];
        return w->conv_wino_dev[i + 1] != nullptr &&
               (nalgo == STITO_CONV_WINOGRAD_F4_SPLIT || nalgo == STITO_CONV_WINOGRAD_F4_SPLIT2 || nalgo == STITO_CONV_WINOGRAD_F4_SPLIT3 ||
                nalgo == STITO_CONV_WINOGRAD_F2_REG) &&
               stito_conv3x3_supported(S, H[nb], W[nb], nci, w->channels[nb + 1], npool, nalgo);
    }

    // conv_block1 as one launch on the register-resident F(2x2,3x3) kernel (it computes the first conv into its patch ring)
    int block1(const float *in, float *out, int S, int s0, bool &have_amax) const {
        TimedLaunch t((hipStream_t)stream, g_conv_timing.on);
        STITO_TRY(t.begin(1));
        unsigned *amax_out = next_wants_amax(1, S) ? amax_of(1, s0) : nullptr;
        const int cout = w->channels[1];
        const int rc = stito_conv_block1_f2reg(in, w->conv1_f2reg_w_dev, w->conv_wino_dev[1], w->bn_scale_dev[1], w->bn_shift_dev[1], out,
                                               S, H[0], W[0], cout, cout, 1, vbuf, vbytes, stream, amax_out);
        if (rc) return rc;
        have_amax = amax_out != nullptr;
        return t.end();
    }

    // conv i on streams s0 .. s0 + S - 1: in / out point at the first of them; have_amax: in -- the producer of `in` reported
    // these streams' maxima (buffer of conv i - 1), out -- this launch reported its own
    int conv(int i, const float *in, float *out, int S, int s0, bool &have_amax) const {
        const int blk = i / 2, j = i % 2;
        const int cin = w->channels[blk], cout = w->channels[blk + 1];
        const int ci = j == 0 ? cin : cout, pool = (j == 1 && blk < 5) ? 1 : 0;
        // Winograd where a transformed weight set was supplied and the map fits; direct otherwise
        int walgo = (w->conv_wino_algo[i] == STITO_CONV_WINOGRAD_F4 || w->conv_wino_algo[i] == STITO_CONV_WINOGRAD_F4_PRE ||
                     w->conv_wino_algo[i] == STITO_CONV_WINOGRAD_F4_SPLIT || w->conv_wino_algo[i] == STITO_CONV_WINOGRAD_F4_SPLIT2 || w->conv_wino_algo[i] == STITO_CONV_WINOGRAD_F4_SPLIT3 ||
                     w->conv_wino_algo[i] == STITO_CONV_WINOGRAD_F2_REG)
                        ? w->conv_wino_algo[i] : STITO_CONV_WINOGRAD;  // (a split packing has no float32 fallback: the direct kernel takes over)
        if (walgo == STITO_CONV_WINOGRAD_F4_PRE && !stito_conv3x3_supported(S, H[blk], W[blk], ci, cout, pool, walgo))
            walgo = STITO_CONV_WINOGRAD_F4;  // same packing
        const float *wino_w = w->conv_wino_dev[i];
        if (walgo == STITO_CONV_WINOGRAD_F4_SPLIT3 && w->conv_alt_dev[i] != nullptr && w->conv_alt_algo[i] == STITO_CONV_WINOGRAD_F4_SPLIT2 &&
            4 * wino43_split3_workgroups(ConvShape{S, H[blk], W[blk], ci, cout}, pool != 0) < 3 * n_cus &&
            stito_conv3x3_supported(S, H[blk], W[blk], ci, cout, pool, STITO_CONV_WINOGRAD_F4_SPLIT2)) {
            walgo = STITO_CONV_WINOGRAD_F4_SPLIT2;   // too few of the large workgroups for this batch: the alternative packing
            wino_w = w->conv_alt_dev[i];
        }
        const bool wino = wino_w != nullptr && stito_conv3x3_supported(S, H[blk], W[blk], ci, cout, pool, walgo);
        TimedLaunch t((hipStream_t)stream, g_conv_timing.on && ci % 8 == 0);
        STITO_TRY(t.begin(i));
        const int algo_i = wino ? walgo : STITO_CONV_DIRECT;
        // can this layer's kernel report the maxima the next one wants?
        unsigned *amax_out = nullptr;
        if ((algo_i >= STITO_CONV_WINOGRAD_F4 || (algo_i == STITO_CONV_DIRECT && ci == 1 && !pool)) && next_wants_amax(i, S)) amax_out = amax_of(i, s0);
        const int rc = conv3x3_ws(in, wino ? wino_w : w->conv_w_dev[i], w->bn_scale_dev[i], w->bn_shift_dev[i], out, S, H[blk], W[blk], ci, cout, pool,
                                  algo_i, vbuf, vbytes, stream, (have_amax && i > 0) ? amax_of(i - 1, s0) : nullptr, amax_out);
        if (rc) return rc;
        have_amax = amax_out != nullptr;
        return t.end();
    }
};
}  // namespace

extern "C" int stito_cnn14_forward(const stito_cnn14_weights *w, const float *logmel_dev, int n_cand, int channels,
                                   int64_t n_frames, float *mid_dev, float *side_dev, void *workspace_dev,
                                   size_t workspace_bytes, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    STITO_REQUIRE(channels == 1 || channels == 2, STITO_E_INVALID, "Invalid number of channels: %d", channels);
    STITO_REQUIRE(n_cand > 0, STITO_E_INVALID, "empty batch");
    const int S = n_cand * channels;
    Trunk tr;
    tr.w = w;
    tr.stream = stream;
    cnn14_dims(n_frames, w->n_mels, tr.H, tr.W);
    const int *H = tr.H, *W = tr.W;
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
    tr.vbytes = cnn14_pre_bytes(w, S, H, W);
    tr.vbuf = ws + align_up(a * 4, 256) + align_up(b * 4, 256) + align_up((size_t)S * w->channels[6] * 4, 256);
    // per-stream output maxima, handed from a layer to the split-precision layer behind it: one buffer per conv, all zeroed by
    // ONE launch at the top of the pass (one per layer was ten more dispatches per pass)
    tr.amax_stride = align_up((size_t)S * sizeof(unsigned), 256);
    tr.amax_all = (unsigned *)((char *)tr.vbuf + tr.vbytes);
    STITO_TRY(zero_async(tr.amax_all, tr.amax_stride * STITO_CNN14_NUM_CONVS, st));
    DeviceInfo dinfo;
    STITO_TRY(device_info(dinfo));   // cached per device
    tr.n_cus = dinfo.cus;
    tr.fuse1r = cnn14_fuse1r(w, S, H, W) && tr.vbytes >= stito_conv_block1_f2reg_workspace_bytes(S, H[0], W[0], w->channels[1], w->channels[1], 1);
    const ChunkPlan plan = cnn14_chunk_plan(w, S, H, W, tr.fuse1r);
    float *chA = (float *)((char *)tr.amax_all + tr.amax_stride * STITO_CNN14_NUM_CONVS);   // the chunked run's two scratch maps
    float *chB = (float *)((char *)chA + align_up(plan.scratch_floats * 4, 256));

    const float *cur = logmel_dev;   // input of conv i
    bool have_amax = false;          // its producer reported the per-stream maxima
    // the buffer the layer-by-layer schedule gives conv i's output: a block's first conv writes actA, its second actB
    auto home = [&](int i) { return i % 2 == 0 ? actA : actB; };
    for (int i = 0; i < STITO_CNN14_NUM_CONVS;) {
        if (i == plan.first) {
            // ---- the chunked run: every layer of it on one chunk of streams after the other ----
            const size_t in_ps = i == 0 ? (size_t)H[0] * W[0] * (w->channels[0] >= 8 ? w->channels[0] : 1) : conv_out_floats_per_stream(w, H, W, i - 1);
            const size_t out_ps = conv_out_floats_per_stream(w, H, W, plan.last);
            float *const run_out = home(plan.last);
            bool all_amax = true;
            for (int s0 = 0; s0 < S; s0 += plan.streams) {
                const int sc = S - s0 < plan.streams ? S - s0 : plan.streams;
                const float *cin_ptr = cur + (size_t)s0 * in_ps;
                bool camax = have_amax;
                int flip = 0;
                for (int k = plan.first; k <= plan.last;) {
                    const bool b1 = k == 0 && tr.fuse1r;   // conv_block1's two convs are one launch writing conv 1's output
                    const int k_out = b1 ? 1 : k;
                    float *o = k_out == plan.last ? run_out + (size_t)s0 * out_ps : (flip ? chB : chA);
                    if (b1) STITO_TRY(tr.block1(cin_ptr, o, sc, s0, camax));
                    else STITO_TRY(tr.conv(k, cin_ptr, o, sc, s0, camax));
                    cin_ptr = o;
                    flip ^= 1;
                    k = k_out + 1;
                }
                all_amax = all_amax && camax;
            }
            have_amax = all_amax;
            cur = run_out;
            i = plan.last + 1;
            continue;
        }
        if (i == 0 && tr.fuse1r) {
            STITO_TRY(tr.block1(cur, actB, S, 0, have_amax));
            cur = actB;
            i = 2;
            continue;
        }
        STITO_TRY(tr.conv(i, cur, home(i), S, 0, have_amax));
        cur = home(i);
        ++i;
    }
    const int C6 = w->channels[6], E = w->embed_dim;
    hipLaunchKernelGGL(k_head, dim3((C6 + 255) / 256, S), dim3(256), 0, st, cur, feat, H[6], W[6], C6);
    STITO_LAUNCH_CHECK();
    const size_t lds = (size_t)FC_SB * (C6 > 256 ? C6 : 256) * sizeof(float);  // the features of 8 streams; reused for 4 x 64 x 8 partial sums
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)k_fc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_fc, dim3((E + 63) / 64, (n_cand + FC_SB - 1) / FC_SB, channels), dim3(256), lds, st, feat,
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
    STITO_TRY(zero_async(flags_dev, 2 * sizeof(int32_t), st));
    const int64_t n = (int64_t)n_cand * embed_dim;
    hipLaunchKernelGGL(k_nan_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mid_dev, side_dev, n, flags_dev);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_embed_loss, dim3(n_cand), dim3(256), 0, st, mid_dev, side_dev, embed_dim, target_mid_dev,
                       target_side_dev, loss_dev, flags_dev, 0);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

extern "C" int stito_neg_cosine(const float *embed_dev, int n_cand, int embed_dim, const float *target_dev, float weight,
                                int accumulate, float *loss_dev, void *stream) {
    STITO_REQUIRE(n_cand > 0 && embed_dim > 0, STITO_E_INVALID, "empty embeddings");
    STITO_REQUIRE(embed_dev != nullptr && target_dev != nullptr && loss_dev != nullptr, STITO_E_INVALID, "null pointer");
    hipLaunchKernelGGL(k_neg_cosine, dim3(n_cand), dim3(256), 0, (hipStream_t)stream, embed_dev, target_dev, embed_dim, weight,
                       accumulate, loss_dev);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}
