// conv_wino23r.hip -- 3x3 conv (pad 1) + BN + ReLU (+ 2x2 average pool) of the Cnn14 trunk's 64- and 128-channel layers
// (reference: ConvBlock.forward, st_ito/models/panns.py:65-80; conv_block1.conv2, conv_block2 of panns.py:250-253) by
// Winograd F(2x2, 3x3) on the f16 matrix pipe with split operands, built the other way round from conv_wino43.hip:
//
//   * the WEIGHTS stay in registers.  U = G g G^T of a 64 -> 64 layer is 16 positions x 64 x 64 x (f16 hi + lo) = 256 KB: half of a
//     CU's register file.  A workgroup is 4 waves of up to 512 registers (one per SIMD); wave i keeps the four positions (i, 0..3)
//     of all input channels and its 32 NB output channels as MFMA operands for the whole launch.  Workgroups are persistent (one per
//     CU) and walk over pixel groups: no weight slab is ever copied again -- the traffic that bound every earlier kernel of these
//     layers (36 KB of weights through L2 and LDS per 4-channel chunk of a 32-tile block) is gone.
//   * the INPUT TRANSFORM happens in registers, straight into the operand layout.  The second operand of
//     v_mfma_f32_32x32x16_f16 wants, per lane, 8 consecutive k (= input channels) of one column (= tile): lane (tile, channel
//     octet) reads the two patch rows its position row combines (8 pixels x 8 channels by ds_read_b128 from the raw f32 halo
//     patch), forms t = d[a1] +- d[a2] and the four V(i, j) = t[b1] +- t[b2] with packed f32 adds, and rounds every value
//     to scaled f16 halves hi + lo with v_fma_mixlo/hi_f16 (one instruction per half: hi = rn16(s v), lo = rn16(s v - hi)).
//     No transformed input in LDS or HBM, no operand reads from LDS at all.
//   * F(2x2,3x3) instead of F(4x4,3x3): 4 MACs per output instead of 2.25, but on the f16 pipe they are cheap (3 products per
//     MAC at 16 x the f32 rate); what decides is that 16 positions of weights fit the registers where 36 do not, and that the
//     transforms are adds only.
//   * the products D = U^T-block x V: A operand = weights (rows = output channels), B operand = V (columns = tiles), so that a
//     lane's accumulators are 4 consecutive output channels of ITS tile: the epilogue's 16-byte exchanges and stores fall out
//     of the layout.
//   * Y = A^T M A with A^T = [1 1 1 0; 0 1 -1 -1]: wave i reduces its row over j in registers (Z_i[c], two values), the sum over i
//     goes through LDS (each wave writes its Z, 16 KB, and finishes a quarter of the group's outputs: BN + ReLU (+ pool) + stores).
//
// Pixel group = 2 tile rows x 16 tile columns (output 4 x 32 pixels before pooling), halo patch 6 x 34 pixels, staged by
// LDS-DMA one k-step (16 channels = 4 channel quads, 13 KB) at a time into a ring of W23_RING = 6 entries.  Patch layout per
// entry: [quad][row parity, column parity][3 x 17 pixels] x 16 bytes -- the four parity planes make the 16 tiles of a
// ds_read_b128 lane group (a tile row; lanes are mapped to tiles group by group) read 16 consecutive 16-byte slots: no bank
// conflicts; wave w copies plane w (51 lanes per instruction, pixels outside the map masked off through EXEC and zeroed by hand).
//
// The group loop is GENERATED (tools/gen/gen_w23_body.py -> conv_wino23r_body.inc / conv_wino23r_pro.inc): one wave per SIMD
// issues about one instruction per 8 cycles (tools/ubench/w23_shadow.hip, profiles/round4_w23_shadow_ubench.txt), so what
// stands between two MFMAs is placed by hand, one asm block per MFMA "slot".  A group is four phases (one per k-step) of 24
// MFMAs; software pipeline over groups g:
//   phase 0      products of k-step 0 | the epilogue of group g - 1 (Z sums of the four waves from LDS -> Y = A^T . A -> BN + ReLU
//                (+ pool) -> stores) in the odd slots: the accumulators of three positions are dead there
//   phase 1      products of k-step 1 | vmcnt(0) + barrier, then the copies of k-steps 2, 3 of group g + 1
//   phase 2      products of k-step 2
//   phase 3      products of k-step 3 | vmcnt(0) + barrier, then the copies of k-steps 0, 1 of group g + 2; the transform of
//                position 0 of the NEXT group's k-step 0 (look-ahead, unconditional: a branch here makes hipcc copy accumulators)
//   then         drain, Z_i = sums over j of this wave's row -> LDS (16 ds_write_b128), barrier, ring rotates by 4 entries
// so an entry is rewritten two phases after its last read, every copy has two phases (~6000 cycles, measured latency under
// load ~5000) to land, and the only waits are the two vmcnt(0) + barrier pairs (the no-partial-wait rule of conv_wino43.hip:
// LDS-DMA copies complete out of order) and the barrier behind the Z writes.  The transform of position p + 1 runs in the
// slots of position p's products (unit = 4 adds + 8 v_fma_mix per B operand pair; tcomb = 2 v_pk_fma_f32 per patch column).
// Measured timeline and ablations: profiles/round4_w23_ablation.txt.
#include "conv_layout.h"

#include <cstdlib>

namespace stito {

typedef _Float16 rh8 __attribute__((ext_vector_type(8)));
typedef unsigned ru4 __attribute__((ext_vector_type(4)));

static constexpr int W23_THREADS = 256;
static constexpr int W23_PLANE = 51;                        // 3 x 17 pixels of one parity plane
static constexpr int W23_ENTRY = 4 * 4 * W23_PLANE * 16;    // bytes of one k-step of the patch: [quad][plane][51][16 B]
static constexpr int W23_RING = 6;  // k-step entries: the four of the group being read + two in flight (see conv_wino23r_body.inc)

#ifndef W23_ABL
#define W23_ABL 0  // timing-experiment bit mask (1 no transform, 2 no MFMAs, 4 no DMA, 8 no epilogue, 16 no Z exchange); 0 in every build that ships
#endif

// power-of-two scale of a stream's transformed input: |B^T d B| <= 4 max|d|, amax < 2^e -> 2^(12 - e): below 2^14
__host__ __device__ __forceinline__ float w23_vscale(unsigned amax_bits) {
    int e = (int)((amax_bits >> 23) & 0xff) - 126;  // amax = f * 2^e, f in [0.5, 1)
    if (amax_bits == 0u) e = 12;                      // all-zero stream: scale 1
    e = e < -40 ? -40 : (e > 60 ? 60 : e);
    return __builtin_ldexpf(1.0f, 12 - e);
}

struct W23Geom {
    int S, H, W, Cin, Cout;
    int Ho, Wo;           // output map (pooled when POOL)
    int TR, TC;           // tile rows / columns per stream that produce output
    int n_bands, n_txb;   // pixel groups per stream: ceil(TR / 2) x ceil(TC / 16)
    int n_groups;         // S * n_bands * n_txb
    int n_cb;             // output-channel blocks (Cout / (32 NB))
    int wg_per_cb;        // persistent workgroups per channel block
    FDiv fGPS, fTXB;      // groups per stream, n_txb as launch-constant divisors
    const unsigned *amax_in;   // per stream: largest input activation (bit pattern)
    unsigned *amax_out;        // or NULL: per stream, the largest output (atomicMax; zeroed by the caller)
    const float *u_inv;        // 1 / weight scale (header of the packed weights)
    long long *clk;            // or NULL (STITO_W23_CLK=1, a measurement aid): {shader clock, 100 MHz clock} at the start / end of one workgroup
    // FUSE1 (conv_block1 in one launch): `in` is the 1-channel log-mel image (S, H, W); the 64-channel input of this layer is
    // relu(bn1(conv3x3(in))), computed on the matrix pipe into the patch ring instead of being copied
    const char *c1w;           // first-conv operands (pack_conv1_f2reg): [channel block of 32][hi | lo][lane] x 16 B
    const unsigned *c1_sm;     // per stream: power-of-two scale of the log-mel operand (bit pattern)
    const unsigned *c1_kinv;   // per stream: 1 / (that scale x the first-conv weight scale): the ring holds K x the activations
};
static constexpr int W23_MELROW = 256;                 // bytes per row of a staged log-mel window (64 floats, 38 used)
static constexpr int W23_MELBUF = 9 * W23_MELROW;      // rows 4 band - 2 .. 4 band + 6

#define W23_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
// The products are inline asm so that the weight operand can be pinned to the AGPR half of the register file ("a": hipcc
// given the builtin keeps MFMA sources in arch VGPRs and copies 256 weight registers per group through v_accvgpr_read, with
// spills).  What the compiler's hazard recognizer no longer sees is handled here: the first product of an accumulator takes
// the inline constant 0 as its C operand (no VALU-written accumulator feeds an MFMA), dependent products on the same
// accumulator issue back to back (the matrix pipe interlocks on an exactly overlapping C), and W23_MFMA_DRAIN (>= 18 wait
// states behind a 16-pass MFMA) stands between the last product and the first VALU read of the accumulators.
#define W23_MFMA0(ACC, A_, B_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(A_), "v"(B_));
#define W23_MFMA(ACC, A_, B_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(A_), "v"(B_));
#define P2(v, h) __builtin_shufflevector(v, v, 2 * (h), 2 * (h) + 1)
#define H2(v, h) ((f32x2){(v)[2 * (h)], (v)[2 * (h) + 1]})   // the same with an index that is only constant after unrolling

// scale as a bit pattern built with integer arithmetic only (the value stays in a scalar register: it is the SGPR operand of
// every v_fma_mix of the transform)
__device__ __forceinline__ float w23_vscale_s(unsigned amax_bits) {
    int e = (int)((amax_bits >> 23) & 0xff) - 126;
    if (amax_bits == 0u) e = 12;
    e = e < -40 ? -40 : (e > 60 ? 60 : e);
    return __uint_as_float((unsigned)(127 + 12 - e) << 23);
}

#define W23_PK4(OP, A_, B_) __builtin_shufflevector(OP(P2(A_, 0), P2(B_, 0)), OP(P2(A_, 1), P2(B_, 1)), 0, 1, 2, 3)
#define W23_FENCE() __builtin_amdgcn_sched_barrier(0);
#define W23_ENT(E_) ((E_) >= W23_RING ? (E_) - W23_RING : (E_))
#ifndef W23_TRACE
#define W23_TRACE 0  // measurement build (tools/ab_build.sh trace -DW23_TRACE=1 + STITO_W23_CLK=1): s_memtime stamps of one wave at the phase
                     // starts, around the two waits and around the Z exchange of four consecutive groups; 0 in every build that ships
#endif
#if W23_TRACE
#define W23_STAMP(N_) if (clk_on && lane == 0 && gi - g_lo >= 8 && gi - g_lo < 12) g.clk[4 + (gi - g_lo - 8) * 16 + (N_)] = (long long)__builtin_readcyclecounter();
#else
#define W23_STAMP(N_)
#endif
#define W23_X() asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); W23_BARRIER()
#define W23_BH(J) __builtin_bit_cast(rh8, bhv[J])
#define W23_BL(J) __builtin_bit_cast(rh8, blv[J])
// the k-step whose patch rows are read next lives in ring entry E_ (< 2 W23_RING)
#define W23_SETP(E_) { const int eo_ = W23_ENT(E_) * W23_ENTRY; pa = ring + eo_ + rd1; pb = ring + eo_ + rd2; }
#define W23_LD(QQ, B_) { dA[(B_) & 1] = *(const f32x4 *)(pa + W23_RDOFF(B_, QQ)); dB[(B_) & 1] = *(const f32x4 *)(pb + W23_RDOFF(B_, QQ)); }

struct W23Cur { int s, band, txb; };  // pixel group = (stream, band of 2 tile rows, block of 16 tile columns)

template <int KS, int NB, bool POOL, bool FUSE1>
__global__ __launch_bounds__(W23_THREADS) void k_conv_wino23r(const float *__restrict__ in, const char *__restrict__ wpk,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                float *__restrict__ out, W23Geom g) {
    static_assert(KS == 4 && NB == 2, "the generated group loop (conv_wino23r_body.inc) is for 4 k-steps x 2 channel halves");
    constexpr int XCH = 4 * 2 * NB * 4 * 1024;  // exchange: [wave][c][n][register quad][lane] x 16 B
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const ring = smem;
    char *const xch = smem + W23_RING * W23_ENTRY;
    float *const bnp = (float *)(xch + XCH);  // [scale | shift][32 NB]
    char *const c1a = (char *)(bnp + 2 * 32 * NB);   // FUSE1: first-conv weight operands (4 KB), two log-mel windows
    char *const melbuf = c1a + 4096;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, t31 = lane & 31;

    // workgroup -> (channel block, group range): workgroups that walk the same groups with different channel blocks are 8 apart
    // (same XCD, b % 8): the patch comes from HBM once
    const int b = blockIdx.x;
    const int cb = (b >> 3) % g.n_cb;
    const int wi = (b & 7) + 8 * ((b >> 3) / g.n_cb);
    const int g_lo = (int)((int64_t)g.n_groups * wi / g.wg_per_cb), g_hi = (int)((int64_t)g.n_groups * (wi + 1) / g.wg_per_cb);
    if (g_lo >= g_hi) return;
    const bool clk_on = g.clk != nullptr && blockIdx.x == (gridDim.x >> 1) && wv == 0;
    long long clk_c0 = 0, clk_r0 = 0;
    if (clk_on) { clk_c0 = (long long)__builtin_readcyclecounter(); clk_r0 = (long long)__builtin_amdgcn_s_memrealtime(); }

    // ---- weights: [cb][wave][j][ks][n][hi | lo][lane] x 16 B, read once, pinned to the AGPRs by the products' "a" operands ------
    rh8 Wt[4][KS][NB][2];
    {
        const char *wp = wpk + ((int64_t)(cb * 4 + wv) * 4 * KS * NB * 2) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int p = 0; p < 2; ++p) Wt[j][ks][n][p] = *(const rh8 *)(wp + ((((j * KS + ks) * NB + n) * 2 + p) * 1024));
    }
    if (tid < 32 * NB) {
        bnp[tid] = scale[cb * 32 * NB + tid];
        bnp[32 * NB + tid] = shift[cb * 32 * NB + tid];
    }
    const unsigned u_inv_bits = __float_as_uint(g.u_inv[0]);  // 1 / weight scale: a power of two

    // ---- lane -> tile: the 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) each get one
    // tile row (16 consecutive 16-byte slots of a parity plane)
    const bool in_a = t31 < 4 || (t31 >= 12 && t31 < 16) || (t31 >= 20 && t31 < 28);
    const int ty = in_a ? 0 : 1;
    const int tx = in_a ? (t31 < 4 ? t31 : (t31 < 16 ? t31 - 8 : t31 - 12)) : (t31 < 12 ? t31 - 4 : (t31 < 20 ? t31 - 8 : t31 - 16));
    // transform of position row i = wv: t[b] = d[a1][b] + sg d[a2][b]   (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1])
    const int a1 = wv == 0 ? 0 : (wv == 2 ? 2 : 1), a2 = wv == 0 ? 2 : (wv == 1 ? 2 : (wv == 2 ? 1 : 3));
    const float sgf = wv == 1 ? 1.0f : -1.0f;
    const f32x2 sg2 = {sgf, sgf};
    // byte offset inside a ring entry of patch pixel (2 ty + a, 2 tx + b), quad 2 half + qq:
    //   ((quad * 4 + (a & 1) * 2 + (b & 1)) * 51 + (ty + a / 2) * 17 + tx + b / 2) * 16
    const int rd_lane = ((2 * half * 4) * W23_PLANE + ty * 17 + tx) * 16;
    const int rd1 = rd_lane + (((a1 & 1) * 2) * W23_PLANE + (a1 >> 1) * 17) * 16;
    const int rd2 = rd_lane + (((a2 & 1) * 2) * W23_PLANE + (a2 >> 1) * 17) * 16;
#define W23_RDOFF(B_, QQ_) ((((QQ_) * 4 + ((B_) & 1)) * W23_PLANE + ((B_) >> 1)) * 16)

    // ---- pixel-group cursors: advanced by compare-and-wrap, no divisions in the loop (one instruction costs this wave ~8 cycles
    // whatever it is: the per-group bookkeeping of the first version -- three pairs of divisions -- was 1 400 cycles) ---------------
#define W23_ADV(C_) { if (++(C_).txb == g.n_txb) { (C_).txb = 0; if (++(C_).band == g.n_bands) { (C_).band = 0; ++(C_).s; } } }
    W23Cur cC, cB, cA, cD = {0, 0, 0};  // the group being multiplied, the next one, the one after, the previous one
    {
        int r_;
        cC.s = fdiv(g_lo, g.fGPS, r_);
        cC.band = fdiv(r_, g.fTXB, cC.txb);
    }
    cB = cC; W23_ADV(cB)
    cA = cB; W23_ADV(cA)

    // ---- patch copies: wave w copies parity plane w; lane L < 51 -> plane pixel (L / 17, L % 17) -> patch pixel (dy, dx).  A lane
    // has a pixel unless its group touches the map's border: four launch-constant lane masks (top row of bands, bottom row,
    // leftmost column of groups, rightmost) ------------------------------------------------------------------------------------
    const int dy = 2 * (lane / 17) + (wv >> 1), dx = 2 * (lane % 17) + (wv & 1);
    const unsigned dma_voff = (unsigned)((dy * g.W + dx) * 32);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned dma_lds = lds0 + (unsigned)(wv * W23_PLANE * 16);
    char *const zf_lane = ring + wv * W23_PLANE * 16 + lane * 16;  // this lane's slot of plane wv, quad 0, entry 0
    const int64_t plane8 = (int64_t)g.H * g.W * 32;               // bytes per channel octet of one stream
    const int64_t in_ss = (int64_t)(g.Cin >> 3) * plane8;          // bytes per stream
    const int in_bs = 4 * g.W * 32, in_o0 = -(g.W + 1) * 32;       // bytes per band; patch origin (-1, -1) of group (0, 0)
    const uint64_t m_all = (1ull << W23_PLANE) - 1;
    const uint64_t m_top = __builtin_amdgcn_ballot_w64(lane < W23_PLANE && dy >= 1);
    const uint64_t m_bot = __builtin_amdgcn_ballot_w64(lane < W23_PLANE && 4 * (g.n_bands - 1) - 1 + dy < g.H);
    const uint64_t m_left = __builtin_amdgcn_ballot_w64(lane < W23_PLANE && dx >= 1);
    const uint64_t m_right = __builtin_amdgcn_ballot_w64(lane < W23_PLANE && 32 * (g.n_txb - 1) - 1 + dx < g.W);
    uint64_t dma_mask = 0;
    const char *dma_base = nullptr;
    bool dma_edge = false;
#define W23_DMA_PREP(C_)                                                                                         \
    {                                                                                                            \
        uint64_t m_ = m_all;                                                                                     \
        if ((C_).band == 0) m_ &= m_top;                                                                         \
        if ((C_).band == g.n_bands - 1) m_ &= m_bot;                                                             \
        if ((C_).txb == 0) m_ &= m_left;                                                                         \
        if ((C_).txb == g.n_txb - 1) m_ &= m_right;                                                              \
        dma_mask = m_;                                                                                           \
        dma_edge = m_ != m_all;                                                                                  \
        dma_base = (const char *)in + ((int64_t)(C_).s * in_ss + ((C_).band * in_bs + (C_).txb * 1024 + in_o0)); \
    }
// channel quad Q_ of k-step KS_ of the group prepared last -> ring entry E_ (< 2 W23_RING): one masked 1 KB copy; the lanes of the
// plane without a pixel get zeros
#define W23_DMA_Q(KS_, E_, Q_)                                                                                   \
    if (!(W23_ABL & 4)) {                                                                                        \
        const char *sb_ = dma_base + (int64_t)(2 * (KS_) + ((Q_) >> 1)) * plane8 + ((Q_) & 1) * 16;               \
        const int eo_ = W23_ENT(E_) * W23_ENTRY + (Q_) * 4 * W23_PLANE * 16;                                     \
        uint64_t keep_;                                                                                          \
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"              \
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"                                      \
                     : "=&s"(keep_) : "v"(dma_voff), "s"(sb_), "s"(dma_lds + (unsigned)eo_), "s"(dma_mask) : "memory"); \
        if (dma_edge && lane < W23_PLANE && !((dma_mask >> lane) & 1)) *(f32x4 *)(zf_lane + eo_) = (f32x4)(0.0f); \
    }

    // ---- FUSE1: the first conv on the matrix pipe.  D[channel, pixel] = sum_k A[channel, k] B[k, pixel], k = 4 r + c over the
    // 4 x 4 log-mel window at the pixel (the 3 x 3 taps of relu(bn1(conv3x3)) with bn1's scale folded in, zero weights on the
    // fourth row / column; slot 15 carries bn1's shift against a constant operand, slot 7 a zero against the same constant).
    // Wave w computes ITS parity plane (the pixels it would have copied): pixel block nb = plane pixels 32 nb .. 32 nb + 31
    // (lane % 32; window rows 2 (lane / 32), + 1 = the lane's k octet), channel block mb = 32 channels = two k-steps of the ring.
    // Operands are f16 hi + lo like the main products' (weights x a per-layer power of two, log-mel x a per-stream one):
    // three products, f32 accumulate; the ring holds K x relu(..), K = the product of the two scales -- a power of two that the
    // transform's scale takes back out (W23_SV), so no instruction is spent on it.  An accumulator register quad = 4
    // consecutive channels of the lane's pixel = one 16-byte slot of the ring's [quad][plane][pixel] layout.
    const int c1_ps1 = 32 + t31 < W23_PLANE ? 32 + t31 : W23_PLANE - 1;
    const int mrd0 = ((2 * (t31 / 17) + (wv >> 1) + 2 * half) * 64 + 2 * (t31 % 17) + (wv & 1)) * 4;
    const int mrd1 = ((2 * (c1_ps1 / 17) + (wv >> 1) + 2 * half) * 64 + 2 * (c1_ps1 % 17) + (wv & 1)) * 4;
    const unsigned mel_voff = (unsigned)lane * 4u;
    const int mel_ss = g.H * g.W * 4;              // bytes per stream of the log-mel image (wino23r_supported: H W 32 < 2^31)
    const unsigned c1_wo = lds0 + (unsigned)(((half * 4 + wv) * W23_PLANE + t31) * 16);   // LDS address of (quad lane / 32, plane wv, pixel lane % 32), entry 0
    const uint64_t mel_call = (1ull << 40) - 1;
    const uint64_t mel_cleft = __builtin_amdgcn_ballot_w64(lane >= 2);
    const uint64_t mel_cright = __builtin_amdgcn_ballot_w64(32 * (g.n_txb - 1) - 2 + lane < g.W);
    const uint64_t mel_cright2 = __builtin_amdgcn_ballot_w64(32 * (g.n_txb - 2) - 2 + lane < g.W);   // the window is 40 wide: the last but one block can cross the edge too
    uint64_t c1_keep0 = 0, c1_keep1 = 0;
    float c1_smf = 1.0f;
    int mel_pB = 1;                // window buffer of the NEXT group (cB); the group after it (cA) uses the other one
    float c1m[8];
    ru4 c1bh, c1bl;
    rh8 c1wh, c1wl;
    f32x16 c1acc;
#define W23_C1_PREP(C_)                                                                                          \
    {                                                                                                            \
        uint64_t m_ = m_all;                                                                                     \
        if ((C_).band == 0) m_ &= m_top;                                                                         \
        if ((C_).band == g.n_bands - 1) m_ &= m_bot;                                                             \
        if ((C_).txb == 0) m_ &= m_left;                                                                         \
        if ((C_).txb == g.n_txb - 1) m_ &= m_right;                                                              \
        c1_keep0 = (m_ & 0xffffffffull) | (m_ << 32);                                                            \
        c1_keep1 = (m_ >> 32) & 0x7ffffull;                                                                      \
        c1_keep1 |= c1_keep1 << 32;                                                                              \
        c1_smf = __uint_as_float(g.c1_sm[(C_).s]);                                                               \
    }
// row RR_ (wave-uniform) of the log-mel window of group C_ -> window buffer P_: one masked LDS-DMA of <= 40 dwords; what lies
// outside the map is the first conv's zero padding, written by hand.  (Rows 0..7 x columns 0..35 carry weights; row 8 and
// the columns behind only ever meet the zero weights of the 4 x 4 window: they are zeroed once, in the prologue.)
#define W23_MEL_ROW(C_, P_, RR_)                                                                                 \
    {                                                                                                            \
        const int rg_ = 4 * (C_).band - 2 + (RR_);                                                               \
        const bool rok_ = (unsigned)rg_ < (unsigned)g.H;                                                         \
        char *const mr_ = melbuf + (P_) * W23_MELBUF + (RR_) * W23_MELROW;                                       \
        if (rok_) {                                                                                              \
            const char *sb_ = mel_sb_ + (rg_ * g.W + 32 * (C_).txb - 2) * 4;   /* 32-bit inside a stream */      \
            uint64_t keep_;                                                                                      \
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"             \
                         "global_load_lds_dword %1, %2\n\ts_mov_b64 exec, %0"                                    \
                         : "=&s"(keep_) : "v"(mel_voff), "s"(sb_), "s"(lds0 + (unsigned)(mr_ - smem)), "s"(mel_cm_) : "memory"); \
        }                                                                                                        \
        if ((!rok_ || mel_cm_ != mel_call) && lane < 40 && !(rok_ && ((mel_cm_ >> lane) & 1))) *(float *)(mr_ + lane * 4) = 0.0f; \
    }
#define W23_MEL_DMA(C_, P_)                                                                                      \
    {                                                                                                            \
        uint64_t mel_cm_ = mel_call;                                                                             \
        const char *mel_sb_ = (const char *)in + (int64_t)(C_).s * mel_ss;                                       \
        if ((C_).txb == 0) mel_cm_ &= mel_cleft;                                                                 \
        if ((C_).txb == g.n_txb - 1) mel_cm_ &= mel_cright;                                                      \
        if ((C_).txb == g.n_txb - 2) mel_cm_ &= mel_cright2;                                                     \
        W23_MEL_ROW(C_, P_, wv) W23_MEL_ROW(C_, P_, wv + 4)                                                      \
    }
#define W23_C1_A(MB_) { c1wh = *(const rh8 *)(c1a + ((MB_) * 2 + 0) * 1024 + lane * 16); c1wl = *(const rh8 *)(c1a + ((MB_) * 2 + 1) * 1024 + lane * 16); }
#define W23_C1_MLD(NB_, P_)                                                                                      \
    {                                                                                                            \
        const char *mp_ = melbuf + (P_) * W23_MELBUF + ((NB_) ? mrd1 : mrd0);                                    \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) c1m[k_] = *(const float *)(mp_ + (k_ >> 2) * W23_MELROW + (k_ & 3) * 4); \
    }
// the lane's 8 window values -> scaled f16 halves; element 7 is the constant operand (the stream's scale itself: f16-exact).
// Lanes whose pixel lies outside the map scale by 0: operand 0, constant slot included -> accumulator 0 -> relu 0 = this
// layer's zero padding, with no second store.
#define W23_C1_B(NB_)                                                                                            \
    {                                                                                                            \
        float vs_;                                                                                               \
        asm volatile("v_mov_b32 %8, %18\n\tv_cndmask_b32 %8, 0, %8, %17\n\t"                                    \
                     "v_fma_mixlo_f16 %0, %9, %8, 0\n\tv_fma_mixlo_f16 %1, %11, %8, 0\n\tv_fma_mixlo_f16 %2, %13, %8, 0\n\tv_fma_mixlo_f16 %3, %15, %8, 0\n\t" \
                     "v_fma_mixhi_f16 %0, %10, %8, 0\n\tv_fma_mixhi_f16 %1, %12, %8, 0\n\tv_fma_mixhi_f16 %2, %14, %8, 0\n\tv_fma_mixhi_f16 %3, %8, 1.0, 0\n\t" \
                     "v_fma_mixlo_f16 %4, %9, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %5, %11, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t" \
                     "v_fma_mixlo_f16 %6, %13, %8, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %7, %15, %8, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t" \
                     "v_fma_mixhi_f16 %4, %10, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %5, %12, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
                     "v_fma_mixhi_f16 %6, %14, %8, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %7, 0, 0, 0"  \
                     : "=&v"(c1bh[0]), "=&v"(c1bh[1]), "=&v"(c1bh[2]), "=&v"(c1bh[3]), "=&v"(c1bl[0]), "=&v"(c1bl[1]), "=&v"(c1bl[2]), "=&v"(c1bl[3]), "=&v"(vs_) \
                     : "v"(c1m[0]), "v"(c1m[1]), "v"(c1m[2]), "v"(c1m[3]), "v"(c1m[4]), "v"(c1m[5]), "v"(c1m[6]), "v"(c1m[7]), \
                       "s"((NB_) ? c1_keep1 : c1_keep0), "s"(c1_smf));                                           \
    }
// lo' hi, hi' lo, hi' hi (the large term last) on ONE accumulator, one product per block of the main stream (a product that
// follows the one it accumulates on directly waits for it)
#define W23_C1_MM(P_)                                                                                            \
    if ((P_) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c1acc) : "v"(c1wl), "v"(__builtin_bit_cast(rh8, c1bh)));            \
    else if ((P_) == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1acc) : "v"(c1wh), "v"(__builtin_bit_cast(rh8, c1bl)));       \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1acc) : "v"(c1wh), "v"(__builtin_bit_cast(rh8, c1bh)));
#define W23_C1_DRAIN() asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c1acc));
// ReLU as a signed-integer max on the bit patterns (no canonicalising instruction in front of it), then the four register quads
// -> ring entries E_, E_ + 1 (channels 16 apart), quads 2 (g & 1) + lane / 32; in pixel block 1 only 19 of the 32 lanes have a pixel
#define W23_C1_ST(NB_, E_)                                                                                       \
    {                                                                                                            \
        asm volatile("s_nop 3" : "+v"(c1acc));                                                                   \
        f32x4 y_[4];                                                                                             \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) y_[r_ >> 2][r_ & 3] = __int_as_float(max(__float_as_int(c1acc[r_]), 0)); \
        const unsigned d0_ = c1_wo + (unsigned)(W23_ENT(E_) * W23_ENTRY + (NB_) * 512);                          \
        const unsigned d1_ = c1_wo + (unsigned)(W23_ENT((E_) + 1) * W23_ENTRY + (NB_) * 512);                    \
        if (NB_) {                                                                                               \
            uint64_t keep_;                                                                                      \
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %7\n\t"                                         \
                         "ds_write_b128 %1, %3\n\tds_write_b128 %1, %4 offset:%8\n\tds_write_b128 %2, %5\n\tds_write_b128 %2, %6 offset:%8\n\t" \
                         "s_mov_b64 exec, %0"                                                                    \
                         : "=&s"(keep_) : "v"(d0_), "v"(d1_), "v"(y_[0]), "v"(y_[1]), "v"(y_[2]), "v"(y_[3]), "s"(0x0007ffff0007ffffull), \
                           "n"(2 * 4 * W23_PLANE * 16) : "memory");                                              \
        } else {                                                                                                 \
            asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %0, %3 offset:%6\n\tds_write_b128 %1, %4\n\tds_write_b128 %1, %5 offset:%6" \
                         :: "v"(d0_), "v"(d1_), "v"(y_[0]), "v"(y_[1]), "v"(y_[2]), "v"(y_[3]), "n"(2 * 4 * W23_PLANE * 16) : "memory"); \
        }                                                                                                        \
    }
// the transform's scale: that of the stream's activations (w23_vscale_s of their bound) / K -- powers of two: the exponents add
#define W23_SV(S_) (FUSE1 ? __uint_as_float(__float_as_uint(w23_vscale_s(g.amax_in[S_])) + g.c1_kinv[S_] - 0x3f800000u) : w23_vscale_s(g.amax_in[S_]))

    f32x16 acc[4][NB];
    ru4 bhv[4], blv[4];            // B operands of the four positions: 8 channels as f16 hi / lo (2 per register)
    f32x2 tt[2][4][2];             // t[quad][patch column][channel pair]
    f32x4 dA[2], dB[2];            // patch rows a1 / a2 in flight
    float vtmp[4];
    const char *pa = ring, *pb = ring;
    if (W23_ABL & 1) {   // timing experiments only: constant operands / accumulators in place of the ablated producers
#pragma unroll
        for (int j = 0; j < 4; ++j) { bhv[j] = (ru4)(0x3c003c00u); blv[j] = (ru4)(0u); }
    }
    if (W23_ABL & 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.0f;
    }

    // ---- epilogue of a group (cD: the group before the one being multiplied): this wave's NB register quads of the outputs -----
    bool e_have = false;
    const char *ex = xch;
    f32x4 e_sc, e_sh, Zr[4];
    f32x2 Yk[2][2][2];             // [output row][output column][channel pair]
    unsigned e_esc_bits = 0x3f800000u;
    int e_esc_s = -1, e_co8 = 0;
    unsigned mx = 0;   // largest stored output of this lane since the last flush, of stream mx_s
    int mx_s = -1;
    const int o_lane = POOL ? (ty * g.Wo + tx) * 8 + 4 * half : ((2 * ty) * g.W + 2 * tx) * 8 + 4 * half;  // per-lane part of the output offset
#define W23_AMAX_FLUSH()                                                                                         \
    if (g.amax_out != nullptr && mx_s >= 0) {                                                                    \
        unsigned m_ = mx;                                                                                        \
        _Pragma("unroll") for (int o_ = 32; o_ > 0; o_ >>= 1) m_ = max(m_, (unsigned)__shfl_xor((int)m_, o_, 64)); \
        if (lane == 0 && m_ > __hip_atomic_load(g.amax_out + mx_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g.amax_out + mx_s, m_); \
    }
// 1 / (u_scale v_scale(stream)) = u_inv * 2^(e - 12): both powers of two, so the exponents add (integer arithmetic on scalars)
#define W23_E_BEGIN()                                                                                            \
    if (!(W23_ABL & 8)) {                                                                                        \
        if (cD.s != e_esc_s) {                                                                                   \
            e_esc_s = cD.s;                                                                                      \
            e_esc_bits = u_inv_bits + 0x3f800000u - __float_as_uint(w23_vscale_s(g.amax_in[cD.s]));              \
        }                                                                                                        \
        if (cD.s != mx_s) { W23_AMAX_FLUSH() mx = 0; mx_s = cD.s; }                                              \
    }
#define W23_E_SETUP(K_)                                                                                          \
    if (!(W23_ABL & 8)) {                                                                                        \
        const int u_ = wv * NB + (K_), n_ = u_ >> 2, rq_ = u_ & 3;                                               \
        ex = xch + ((n_ * 4 + rq_) * 64 + lane) * 16;                                                            \
        const int cl_ = 32 * n_ + 8 * rq_ + 4 * half;                                                            \
        e_sc = *(const f32x4 *)(bnp + cl_) * __uint_as_float(e_esc_bits);                                        \
        e_sh = *(const f32x4 *)(bnp + 32 * NB + cl_);                                                            \
        e_co8 = cb * 4 * NB + 4 * n_ + rq_;                                                                      \
    }
#define W23_E_LD(C_)                                                                                             \
    if (!(W23_ABL & 8)) {                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) Zr[i_] = *(const f32x4 *)(ex + ((i_ * 2 + (C_)) * NB * 4 * 64) * 16); \
    }
// Y[0][c] = relu(bn(Z0 + Z1 + Z2)), Y[1][c] = relu(bn(Z1 - Z2 - Z3)): per channel pair six packed instructions in ONE asm
// statement (no compiler s_nop between dependent statements), then eight v_max (asm: the compiler canonicalises in front of its own)
#define W23_E_Y(C_)                                                                                              \
    if (!(W23_ABL & 8)) {                                                                                        \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                       \
            f32x2 t0_, t1_;                                                                                      \
            asm volatile("v_pk_add_f32 %2, %4, %5\n\tv_pk_add_f32 %3, %5, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"      \
                         "v_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"      \
                         "v_pk_fma_f32 %0, %2, %8, %9\n\tv_pk_fma_f32 %1, %3, %8, %9"                            \
                         : "=&v"(Yk[0][C_][h_]), "=&v"(Yk[1][C_][h_]), "=&v"(t0_), "=&v"(t1_)                    \
                         : "v"(H2(Zr[0], h_)), "v"(H2(Zr[1], h_)), "v"(H2(Zr[2], h_)), "v"(H2(Zr[3], h_)), "v"(H2(e_sc, h_)), "v"(H2(e_sh, h_))); \
        }                                                                                                        \
        asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\t" \
                     "v_max_f32 %4, 0, %4\n\tv_max_f32 %5, 0, %5\n\tv_max_f32 %6, 0, %6\n\tv_max_f32 %7, 0, %7"  \
                     : "+v"(Yk[0][C_][0][0]), "+v"(Yk[0][C_][0][1]), "+v"(Yk[0][C_][1][0]), "+v"(Yk[0][C_][1][1]), \
                       "+v"(Yk[1][C_][0][0]), "+v"(Yk[1][C_][0][1]), "+v"(Yk[1][C_][1][0]), "+v"(Yk[1][C_][1][1])); \
    }
#define W23_MAX4(V_) max(max(__float_as_uint((V_)[0]), __float_as_uint((V_)[1])), max(__float_as_uint((V_)[2]), __float_as_uint((V_)[3])))
#define W23_Y4(R_, C_) __builtin_shufflevector(Yk[R_][C_][0], Yk[R_][C_][1], 0, 1, 2, 3)
#define W23_E_ST(K_)                                                                                             \
    if (!(W23_ABL & 8)) {                                                                                        \
        if (POOL) {                                                                                              \
            const int64_t ob_ = ((((int64_t)cD.s * (g.Cout >> 3) + e_co8) * g.Ho + 2 * cD.band) * g.Wo + 16 * cD.txb) * 8; \
            f32x2 p_[2];                                                                                         \
            const f32x2 q2_ = {0.25f, 0.25f};                                                                    \
            _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                     \
                asm volatile("v_pk_add_f32 %0, %1, %2\n\tv_pk_add_f32 %0, %0, %3\n\tv_pk_add_f32 %0, %0, %4\n\tv_pk_mul_f32 %0, %0, %5" \
                             : "=&v"(p_[h_]) : "v"(Yk[0][0][h_]), "v"(Yk[0][1][h_]), "v"(Yk[1][0][h_]), "v"(Yk[1][1][h_]), "v"(q2_)); \
            if (e_have && 2 * cD.band + ty < g.Ho && 16 * cD.txb + tx < g.Wo) {                                  \
                const f32x4 v_ = __builtin_shufflevector(p_[0], p_[1], 0, 1, 2, 3);                              \
                *(f32x4 *)(out + ob_ + o_lane) = v_;                                                             \
                mx = max(mx, W23_MAX4(v_));                                                                      \
            }                                                                                                    \
        } else {                                                                                                 \
            const int64_t ob_ = ((((int64_t)cD.s * (g.Cout >> 3) + e_co8) * g.H + 4 * cD.band) * g.W + 32 * cD.txb) * 8; \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                                     \
                _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                                 \
                    if (e_have && 4 * cD.band + 2 * ty + r_ < g.H && 32 * cD.txb + 2 * tx + c_ < g.W) {          \
                        const f32x4 v_ = W23_Y4(r_, c_);                                                         \
                        *(f32x4 *)(out + ob_ + o_lane + (r_ * g.W + c_) * 8) = v_;                               \
                        mx = max(mx, W23_MAX4(v_));                                                              \
                    }                                                                                            \
        }                                                                                                        \
    }
// Z_i[c] = sum_j M(i, j) A[j][c]:  c = 0: M0 + M1 + M2,  c = 1: M1 - M2 - M3 -> exchange buffer.  The drain (>= 18 wait states
// behind the last 16-pass MFMA) stands in front of the first VALU read of an accumulator; four packed adds per channel pair in
// one asm statement, the two 16-byte stores of a register quad right behind their sums (the LDS takes ~13 cycles per store and
// wave: the sums of the next quad issue meanwhile); the barrier behind the writes makes them visible to the next phase 0,
// where every wave finishes its share of this group's outputs.
#define W23_ACC2(J_, N_, R_) ((f32x2){acc[J_][N_][R_], acc[J_][N_][(R_) + 1]})
#define W23_ZSTORE()                                                                                             \
    W23_STAMP(8)                                                                                                 \
    if (!(W23_ABL & 16)) {                                                                                       \
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1])); \
        _Pragma("unroll") for (int n_ = 0; n_ < NB; ++n_)                                                        \
            _Pragma("unroll") for (int rq_ = 0; rq_ < 4; ++rq_) {                                                \
                f32x2 z0_[2], z1_[2];                                                                            \
                _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                 \
                    asm volatile("v_pk_add_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t" \
                                 "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5 neg_lo:[0,1] neg_hi:[0,1]"  \
                                 : "=&v"(z0_[h_]), "=&v"(z1_[h_])                                                \
                                 : "v"(W23_ACC2(0, n_, 4 * rq_ + 2 * h_)), "v"(W23_ACC2(1, n_, 4 * rq_ + 2 * h_)), \
                                   "v"(W23_ACC2(2, n_, 4 * rq_ + 2 * h_)), "v"(W23_ACC2(3, n_, 4 * rq_ + 2 * h_))); \
                *(f32x4 *)(xch + ((((wv * 2 + 0) * NB + n_) * 4 + rq_) * 64 + lane) * 16) = __builtin_shufflevector(z0_[0], z0_[1], 0, 1, 2, 3); \
                *(f32x4 *)(xch + ((((wv * 2 + 1) * NB + n_) * 4 + rq_) * 64 + lane) * 16) = __builtin_shufflevector(z1_[0], z1_[1], 0, 1, 2, 3); \
            }                                                                                                    \
    }                                                                                                            \
    W23_STAMP(9)                                                                                                 \
    W23_BARRIER()                                                                                                \
    W23_STAMP(10)

    // scale of the transformed input of the group being multiplied and of the next one (reloaded only when the stream changes)
    float sv = W23_SV(cC.s), sv_n = sv;
#define W23_NEXT_SV() { sv_n = cB.s != cC.s ? W23_SV(cB.s) : sv; }
#define W23_ROTATE()                                                                                             \
    {                                                                                                            \
        e_have = true; cD = cC; cC = cB; cB = cA; W23_ADV(cA)                                                    \
        sv = sv_n;                                                                                               \
        ent = W23_ENT(ent + 4);                                                                                  \
        mel_pB = 1 - mel_pB;                                                                                     \
    }

    // ---- prologue: the first group's four k-steps (entries 0..3) and the second group's first two (entries 4, 5); t and the
    // position-0 operands of the first k-step ------------------------------------------------------------------------------
    int ent = 0;  // ring entry of the current group's k-step 0
    if constexpr (FUSE1) {
        // first-conv operands and the windows of the first two groups -> LDS; then the first group's four k-steps (entries 0..3)
        // and the second group's first two (entries 4, 5) are computed
        *(f32x4 *)(c1a + tid * 16) = *(const f32x4 *)(g.c1w + tid * 16);
        for (int i = tid; i < 2 * W23_MELBUF / 16; i += W23_THREADS) *(f32x4 *)(melbuf + i * 16) = (f32x4)(0.0f);   // row 8, columns 40..: zero for good
        W23_BARRIER()
        W23_MEL_DMA(cC, 0)
        if (g_lo + 1 < g_hi) { W23_MEL_DMA(cB, 1) }
        W23_X()
        W23_C1_PREP(cC)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            W23_C1_A(mb)
            W23_C1_MLD(0, 0) W23_C1_B(0) W23_C1_MM(0) W23_C1_MM(1) W23_C1_MM(2) W23_C1_DRAIN() W23_C1_ST(0, 2 * mb)
            W23_C1_MLD(1, 0) W23_C1_B(1) W23_C1_MM(0) W23_C1_MM(1) W23_C1_MM(2) W23_C1_DRAIN() W23_C1_ST(1, 2 * mb)
        }
        if (g_lo + 1 < g_hi) {
            W23_C1_PREP(cB)
            W23_C1_A(0)
            W23_C1_MLD(0, 1) W23_C1_B(0) W23_C1_MM(0) W23_C1_MM(1) W23_C1_MM(2) W23_C1_DRAIN() W23_C1_ST(0, 4)
            W23_C1_MLD(1, 1) W23_C1_B(1) W23_C1_MM(0) W23_C1_MM(1) W23_C1_MM(2) W23_C1_DRAIN() W23_C1_ST(1, 4)
        }
    } else {
        W23_DMA_PREP(cC)
#pragma unroll
        for (int k = 0; k < 4; ++k) { W23_DMA_Q(k, k, 0) W23_DMA_Q(k, k, 1) W23_DMA_Q(k, k, 2) W23_DMA_Q(k, k, 3) }
        if (g_lo + 1 < g_hi) {
            W23_DMA_PREP(cB)
#pragma unroll
            for (int k = 0; k < 2; ++k) { W23_DMA_Q(k, 4 + k, 0) W23_DMA_Q(k, 4 + k, 1) W23_DMA_Q(k, 4 + k, 2) W23_DMA_Q(k, 4 + k, 3) }
        }
    }
    W23_X()
    W23_SETP(0)
#include "conv_wino23r_pro.inc"
    W23_FENCE()

    if constexpr (FUSE1) {
#include "conv_wino23r_body_f1.inc"
    } else {
#include "conv_wino23r_body.inc"
    }

    // ---- the last group's outputs (its Z are behind W23_ZSTORE's barrier; e_have is true: the loop ran at least once) -------------
    W23_E_BEGIN()
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        W23_E_SETUP(k) W23_E_LD(0) W23_E_Y(0) W23_E_LD(1) W23_E_Y(1) W23_E_ST(k)
    }
    W23_AMAX_FLUSH()
    if (clk_on && lane == 0) {
        g.clk[0] = clk_c0; g.clk[1] = (long long)__builtin_readcyclecounter();
        g.clk[2] = clk_r0; g.clk[3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// ---- weights: U = G g G^T (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], float64, rounded once to float32), scaled by the layer's
// power of two (max |U| < 2^e -> 2^(14 - e)) and split into f16 halves, in the register order of the kernel:
// [cout / (32 NB)][wave i][j][k-step][n][hi | lo][lane] x 16 B, lane (m = lane % 32, octet = lane / 32) = 8 input channels
// 16 ks + 8 octet .. of output channel 32 (NB cb + n) + m.  hdr = {max bits, 1 / scale, scale} behind the data.
template <int PASS>
__global__ void k_pack_wino23r(const float *__restrict__ w, int Cout, int Cin, int NB, char *__restrict__ o, unsigned *__restrict__ hdr) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Cout * Cin) return;
    const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    double gk[3][3], t[4][3];
    for (int a = 0; a < 3; ++a)
        for (int bb = 0; bb < 3; ++bb) gk[a][bb] = (double)w[((int64_t)co * Cin + ci) * 9 + a * 3 + bb];
    for (int a = 0; a < 4; ++a)
        for (int bb = 0; bb < 3; ++bb) t[a][bb] = G[a][0] * gk[0][bb] + G[a][1] * gk[1][bb] + G[a][2] * gk[2][bb];
    const int KS = Cin / 16;
    const int cb = co / (32 * NB), n = (co / 32) % NB, m = co & 31;
    const int ks = ci >> 4, oct = (ci >> 3) & 1, e8 = ci & 7;
    const float su = PASS == 1 ? __uint_as_float(hdr[2]) : 1.0f;
    unsigned mx = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const float u = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
            if (PASS == 0) {
                mx = max(mx, __float_as_uint(u) & 0x7fffffffu);
            } else {
                const float us = u * su;
                const _Float16 hi = (_Float16)us, lo = (_Float16)(us - (float)hi);
                char *d = o + ((((((int64_t)cb * 4 + i) * 4 + j) * KS + ks) * NB + n) * 2) * 1024 + (oct * 32 + m) * 16 + e8 * 2;
                *(_Float16 *)d = hi;
                *(_Float16 *)(d + 1024) = lo;
            }
        }
    if (PASS == 0 && mx) atomicMax(hdr, mx);
}

__global__ void k_pack_wino23r_scale(unsigned *hdr) {
    const unsigned mb = hdr[0];
    int e = (int)((mb >> 23) & 0xff) - 126;
    if (mb == 0u) e = 14;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    hdr[1] = __float_as_uint(__builtin_ldexpf(1.0f, e - 14));
    hdr[2] = __float_as_uint(__builtin_ldexpf(1.0f, 14 - e));
}

static int w23_nb(int cin) { return cin == 64 ? 2 : 0; }

size_t wino23r_packed_floats(int cout, int cin) { return (size_t)16 * cout * cin + 64; }

// dynamic LDS of k_conv_wino23r<4, 2, ., FUSE1>: patch ring + Z exchange + BN constants (+ first-conv weights and two log-mel buffers)
static size_t w23_lds_bytes(bool fuse1) {
    constexpr int NB = 2;
    return (size_t)W23_RING * W23_ENTRY + (size_t)4 * 2 * NB * 4 * 1024 + (size_t)2 * 32 * NB * sizeof(float) +
           (fuse1 ? (size_t)4096 + 2 * W23_MELBUF : 0);
}

// gfx950 only: the kernel needs ~144 KB of LDS per workgroup (153 KB with the first conv fused) and 512 registers per wave; on a
// device that does not offer that much LDS per block the shape is reported unsupported and the trunk falls back to F(4x4,3x3).
static bool w23_device_fits(bool fuse1) {
    DeviceInfo d;
    return device_info(d) == STITO_OK && (size_t)d.lds_per_block >= w23_lds_bytes(fuse1);
}

bool wino23r_supported(const ConvShape &c, bool pool) {
    if (w23_nb(c.Cin) == 0 || c.Cout % (32 * w23_nb(c.Cin)) != 0) return false;
    if (!w23_device_fits(false)) return false;
    if (pool && (c.H < 2 || c.W < 2)) return false;
    if ((int64_t)c.H * c.W * 32 >= (1ll << 31)) return false;  // 32-bit byte offsets inside a channel-octet plane
    const int tr = pool ? c.H / 2 : (c.H + 1) / 2, tc = pool ? c.W / 2 : (c.W + 1) / 2;
    const int64_t groups = (int64_t)c.S * ((tr + 1) / 2) * ((tc + 15) / 16);
    return groups > 0 && groups < (1ll << 22);  // fdiv range
}

size_t wino23r_workspace_bytes(const ConvShape &c, bool pool) {
    return wino23r_supported(c, pool) ? align_up((size_t)c.S * sizeof(unsigned), 256) : 0;  // stream maxima when the caller has none
}

// f16-pipe FLOPs the kernel issues: groups x 32 tiles x 16 positions x cin x cout x 3 products
double wino23r_issued_flops(const ConvShape &c, bool pool) {
    if (!wino23r_supported(c, pool)) return 0.0;
    const int tr = pool ? c.H / 2 : (c.H + 1) / 2, tc = pool ? c.W / 2 : (c.W + 1) / 2;
    const double groups = (double)c.S * ((tr + 1) / 2) * ((tc + 15) / 16);
    return 3.0 * 2.0 * groups * 32.0 * 16.0 * c.Cin * c.Cout;
}

int pack_wino23r(const float *w_oihw, int cout, int cin, float *packed, hipStream_t st) {
    const int nb = w23_nb(cin);
    STITO_REQUIRE(nb > 0 && cout % (32 * nb) == 0, STITO_E_UNSUPPORTED, "conv (winograd F(2x2,3x3), register-resident weights): cin %d / cout %d", cin, cout);
    const int64_t n = (int64_t)cout * cin;
    unsigned *hdr = (unsigned *)(packed + (size_t)16 * cout * cin);
    STITO_TRY(zero_async(hdr, 64 * sizeof(float), st));
    hipLaunchKernelGGL(k_pack_wino23r<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, nb, (char *)packed, hdr);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_wino23r_scale, dim3(1), dim3(1), 0, st, hdr);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_wino23r<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, nb, (char *)packed, hdr);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

__global__ __launch_bounds__(256) void k_stream_absmax23(const float *__restrict__ x, int64_t per_stream, unsigned *__restrict__ amax) {
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

struct W23Fuse1 { const char *c1w; const unsigned *sm, *kinv; };   // FUSE1 launch: `in` is the log-mel image, amax_in the bound of the first conv's output

template <bool POOL, bool FUSE1 = false>
static int launch_w23(const float *in, const float *wpk, const float *scale, const float *shift, float *out, const ConvShape &c,
                      char *ws, hipStream_t st, const unsigned *amax_in, unsigned *amax_out, const W23Fuse1 *f1 = nullptr) {
    constexpr int KS = 4, NB = 2;
    W23Geom g{};
    g.S = c.S; g.H = c.H; g.W = c.W; g.Cin = c.Cin; g.Cout = c.Cout;
    g.Ho = POOL ? c.H / 2 : c.H;
    g.Wo = POOL ? c.W / 2 : c.W;
    g.TR = POOL ? g.Ho : (c.H + 1) / 2;
    g.TC = POOL ? g.Wo : (c.W + 1) / 2;
    g.n_bands = (g.TR + 1) / 2;
    g.n_txb = (g.TC + 15) / 16;
    g.n_groups = c.S * g.n_bands * g.n_txb;
    g.n_cb = c.Cout / (32 * NB);
    g.fGPS = make_fdiv(g.n_bands * g.n_txb);
    g.fTXB = make_fdiv(g.n_txb);
    DeviceInfo dinfo;
    STITO_TRY(device_info(dinfo));   // cached per device
    // persistent workgroups, one per CU; per channel block a multiple of 8 of them (one set per XCD)
    int per_cb = (dinfo.cus / g.n_cb) & ~7;
    if (per_cb < 8) per_cb = 8;
    // tuning / test aid, read per launch: the parity tests and tools/block1_fuzz.py sweep it inside one process (a getenv, no device query)
    if (const char *e = getenv("STITO_W23_WG")) { const int v = atoi(e); if (v >= 8 && v % 8 == 0) per_cb = v; }
    g.wg_per_cb = per_cb;
    const unsigned *amax = amax_in;
    // measurement aid (tools/conv_bench.py): STITO_W23_AMAX_ONCE=1 keeps the maxima a previous call left in the same workspace
    // for the same input instead of scanning the input again (inside the trunk the producing layer reports them: no scan at all)
    static const bool amax_once = [] { const char *e = getenv("STITO_W23_AMAX_ONCE"); return e && atoi(e) != 0; }();
    static thread_local const void *amax_have_in = nullptr, *amax_have_ws = nullptr;
    if (amax_in == nullptr && amax_once && amax_have_in == (const void *)in && amax_have_ws == (const void *)ws) {
        amax = (const unsigned *)ws;
    } else if (amax_in == nullptr) {
        amax_have_in = in; amax_have_ws = ws;
        unsigned *amax_ws = (unsigned *)ws;
        amax = amax_ws;
        STITO_TRY(zero_async(amax_ws, (size_t)c.S * sizeof(unsigned), st));
        const int64_t per_stream = (int64_t)c.Cin * c.H * c.W;
        int splits = (int)((per_stream / 4 + 256 * 16 - 1) / (256 * 16));
        const int cap = (4096 + c.S - 1) / c.S;
        splits = splits > cap ? cap : (splits < 1 ? 1 : splits);
        hipLaunchKernelGGL(k_stream_absmax23, dim3((unsigned)splits, (unsigned)c.S), dim3(256), 0, st, in, per_stream, amax_ws);
        STITO_LAUNCH_CHECK();
    }
    g.amax_in = amax;
    g.amax_out = amax_out;
    g.u_inv = wpk + (size_t)16 * c.Cout * c.Cin + 1;
    if (FUSE1) { g.c1w = f1->c1w; g.c1_sm = f1->sm; g.c1_kinv = f1->kinv; }
    auto kern = k_conv_wino23r<KS, NB, POOL, FUSE1>;
    const size_t lds = w23_lds_bytes(FUSE1);
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const bool clk_probe = [] { const char *e = getenv("STITO_W23_CLK"); return e && atoi(e) != 0; }();
    static long long *clk_dev = nullptr;
    if (clk_probe && clk_dev == nullptr) { STITO_HIP_CHECK(hipMalloc(&clk_dev, 68 * sizeof(long long))); STITO_HIP_CHECK(hipMemset(clk_dev, 0, 68 * sizeof(long long))); }
    g.clk = clk_probe ? clk_dev : nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)(per_cb * g.n_cb)), dim3(W23_THREADS), lds, st, in, (const char *)wpk, scale, shift, out, g);
    STITO_LAUNCH_CHECK();
    if (clk_probe) {   // measurement aid: synchronises
        long long v[68];
        STITO_HIP_CHECK(hipStreamSynchronize(st));
        STITO_HIP_CHECK(hipMemcpy(v, clk_dev, sizeof(v), hipMemcpyDeviceToHost));
        const double us = (double)(v[3] - v[2]) / 100.0;
        fprintf(stderr, "[stito clock] k_conv_wino23r%s %dx%d %d->%d: workgroup in the middle of the grid ran %.1f us, %lld shader cycles, %.0f MHz\n", FUSE1 ? " (first conv fused)" : "", c.H, c.W,
                c.Cin, c.Cout, us, v[1] - v[0], us > 0 ? (double)(v[1] - v[0]) / us : 0.0);
#if W23_TRACE
        // stamps: 0 phase 0, 1 phase 1, 2 | 3 before / after its wait + barrier, 4 phase 2, 5 phase 3, 6 | 7 its wait, 8 Z exchange, 9 its barrier, 10 done
        for (int gq = 0; gq < 4; ++gq) {
            const long long *q = v + 4 + gq * 16;
            fprintf(stderr, "[stito trace] group %d: ph0 %lld  ph1 %lld (wait %lld)  ph2 %lld  ph3 %lld (wait %lld)  Z %lld  barrier %lld  | group %lld cycles\n", gq,
                    q[1] - q[0], q[4] - q[1], q[3] - q[2], q[5] - q[4], q[8] - q[5], q[7] - q[6], q[9] - q[8], q[10] - q[9], q[10] - q[0]);
        }
#endif
    }
    return STITO_OK;
}

int launch_wino23r(const float *in, const float *wpk, const float *scale, const float *shift, float *out, const ConvShape &c, bool pool,
                   void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    STITO_REQUIRE(wino23r_supported(c, pool), STITO_E_UNSUPPORTED,
                  "conv (winograd F(2x2,3x3), register-resident weights): %dx%d map, %d -> %d channels not covered", c.H, c.W, c.Cin, c.Cout);
    STITO_REQUIRE(amax_in != nullptr || (ws != nullptr && ws_bytes >= wino23r_workspace_bytes(c, pool)), STITO_E_WORKSPACE,
                  "conv (winograd F(2x2,3x3), register-resident weights): workspace have %zu need %zu", ws_bytes, wino23r_workspace_bytes(c, pool));
    return pool ? launch_w23<true>(in, wpk, scale, shift, out, c, (char *)ws, st, amax_in, amax_out)
                : launch_w23<false>(in, wpk, scale, shift, out, c, (char *)ws, st, amax_in, amax_out);
}

// ---- conv_block1 in one launch (FUSE1) ------------------------------------------------------------------------------------------
// First-conv operands: A[channel][k = 4 r + c] over the 4 x 4 window = bn1 scale x w1[channel][r][c] (r, c < 3), bn1 shift in
// slot 15, zero elsewhere; x a power of two (max |A| < 2^e -> 2^(14 - e)), f16 hi + lo, in the A-operand order of
// v_mfma_f32_32x32x16_f16: [channel block of 32][hi | lo][lane = 32 octet + m] x 16 B (k = 8 octet .. + 7 of channel 32 mb + m).
// hdr (behind the 4 KB): {scale bits, 1 / scale, max_ch sum_k<15 |A|, max_ch |shift|}.  One wave does it all.
__global__ __launch_bounds__(64) void k_pack_conv1_f2reg(const float *__restrict__ w, const float *__restrict__ scale, const float *__restrict__ shift,
                                                          char *__restrict__ o) {
    const int ch = threadIdx.x;
    float a[16];
    float sum = 0.0f, mxa = 0.0f;
    for (int k = 0; k < 16; ++k) {
        const int r = k >> 2, c = k & 3;
        a[k] = (r < 3 && c < 3) ? w[ch * 9 + r * 3 + c] * scale[ch] : (k == 15 ? shift[ch] : 0.0f);
        if (k < 15) sum += fabsf(a[k]);
        mxa = fmaxf(mxa, fabsf(a[k]));
    }
    float bsh = fabsf(shift[ch]);
    for (int of = 32; of > 0; of >>= 1) {
        sum = fmaxf(sum, __shfl_xor(sum, of, 64));
        mxa = fmaxf(mxa, __shfl_xor(mxa, of, 64));
        bsh = fmaxf(bsh, __shfl_xor(bsh, of, 64));
    }
    const unsigned mb_ = __float_as_uint(mxa);
    int e = (int)((mb_ >> 23) & 0xff) - 126;
    if (mb_ == 0u) e = 14;
    e = e < -40 ? -40 : (e > 40 ? 40 : e);
    const float su = __builtin_ldexpf(1.0f, 14 - e);
    for (int k = 0; k < 16; ++k) {
        const float us = a[k] * su;
        const _Float16 hi = (_Float16)us, lo = (_Float16)(us - (float)hi);
        char *d = o + ((ch >> 5) * 2) * 1024 + ((k >> 3) * 32 + (ch & 31)) * 16 + (k & 7) * 2;
        *(_Float16 *)d = hi;
        *(_Float16 *)(d + 1024) = lo;
    }
    if (ch == 0) {
        float *hdr = (float *)(o + 4096);
        hdr[0] = su;
        hdr[1] = __builtin_ldexpf(1.0f, e - 14);
        hdr[2] = sum;
        hdr[3] = bsh;
    }
}

size_t conv1_f2reg_packed_floats() { return 1024 + 64; }

int pack_conv1_f2reg(const float *w_dev, const float *scale_dev, const float *shift_dev, int c1, float *packed, hipStream_t st) {
    STITO_REQUIRE(c1 == 64, STITO_E_UNSUPPORTED, "fused first conv (register-resident F(2x2,3x3) block): %d channels", c1);
    hipLaunchKernelGGL(k_pack_conv1_f2reg, dim3(1), dim3(64), 0, st, w_dev, scale_dev, shift_dev, (char *)packed);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

// per stream: the log-mel operand's scale (max |x| < 2^e -> 2^(12 - e), kept inside f16's normal range: it is also the constant
// operand of the shift slot), a bound of the first conv's output (sum |A| max |x| + max |shift|: the scale of the transformed
// input is taken from it, as the other layers take it from the measured maximum), 1 / (log-mel scale x weight scale)
__global__ __launch_bounds__(256) void k_w23_mel_params(const float *__restrict__ x, int64_t per_stream, const float *__restrict__ hdr,
                                                        unsigned *__restrict__ amax_in, unsigned *__restrict__ sm, unsigned *__restrict__ kinv) {
    __shared__ unsigned red[4];
    const float *xs = x + (int64_t)blockIdx.x * per_stream;
    unsigned m = 0;
    for (int64_t i = threadIdx.x; i < per_stream; i += 256) m = max(m, __float_as_uint(xs[i]) & 0x7fffffffu);
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, of, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        int e = (int)((m >> 23) & 0xff) - 126;
        if (m == 0u) e = 12;
        e = e < -3 ? -3 : (e > 26 ? 26 : e);                 // 2^-14 <= scale <= 2^15
        const float s_m = __builtin_ldexpf(1.0f, 12 - e);
        const float bound = (hdr[2] * __uint_as_float(m) + hdr[3]) * 1.001f;
        amax_in[blockIdx.x] = __float_as_uint(bound);
        sm[blockIdx.x] = __float_as_uint(s_m);
        kinv[blockIdx.x] = __float_as_uint(hdr[1] / s_m);    // powers of two: exact
        if (blockIdx.x == 0) {   // the two entries behind the last stream: read (not used) by the kernel's look-ahead
            const int S = (int)gridDim.x;
            amax_in[S] = amax_in[S + 1] = 0u;
            sm[S] = sm[S + 1] = 0x3f800000u;
            kinv[S] = kinv[S + 1] = 0x3f800000u;
        }
    }
}

bool wino23r_fused1_supported(const ConvShape &c, bool pool) {   // c: the SECOND conv's shape (Cin = the first conv's channels)
    return c.Cin == 64 && wino23r_supported(c, pool) && w23_device_fits(true);
}

size_t wino23r_fused1_workspace_bytes(const ConvShape &c, bool pool) {
    return wino23r_fused1_supported(c, pool) ? 3 * align_up((size_t)(c.S + 2) * sizeof(unsigned), 256) : 0;
}

int launch_wino23r_fused1(const float *logmel, const float *c1pk, const float *wpk, const float *scale, const float *shift, float *out,
                          const ConvShape &c, bool pool, void *ws, size_t ws_bytes, hipStream_t st, unsigned *amax_out) {
    STITO_REQUIRE(wino23r_fused1_supported(c, pool), STITO_E_UNSUPPORTED,
                  "fused conv block (winograd F(2x2,3x3), register-resident weights): %dx%d map, %d -> %d channels not covered", c.H, c.W, c.Cin, c.Cout);
    STITO_REQUIRE(ws != nullptr && ws_bytes >= wino23r_fused1_workspace_bytes(c, pool), STITO_E_WORKSPACE,
                  "fused conv block (winograd F(2x2,3x3), register-resident weights): workspace have %zu need %zu", ws_bytes, wino23r_fused1_workspace_bytes(c, pool));
    const size_t part = align_up((size_t)(c.S + 2) * sizeof(unsigned), 256);
    unsigned *amax_in = (unsigned *)ws, *sm = (unsigned *)((char *)ws + part), *kinv = (unsigned *)((char *)ws + 2 * part);
    hipLaunchKernelGGL(k_w23_mel_params, dim3((unsigned)c.S), dim3(256), 0, st, logmel, (int64_t)c.H * c.W, c1pk + 1024, amax_in, sm, kinv);
    STITO_LAUNCH_CHECK();
    const W23Fuse1 f1{(const char *)c1pk, sm, kinv};
    return pool ? launch_w23<true, true>(logmel, wpk, scale, shift, out, c, (char *)ws, st, amax_in, amax_out, &f1)
                : launch_w23<false, true>(logmel, wpk, scale, shift, out, c, (char *)ws, st, amax_in, amax_out, &f1);
}

}  // namespace stito
