// conv_wino43.hip -- 3x3 conv (pad 1) + BN + ReLU (+ 2x2 average pool) of the Cnn14 trunk (reference
// st_ito/models/panns.py:25-80, 250-261) by Winograd F(4x4, 3x3) on exact-f32 MFMA.
//
// Per 6x6 input tile d (stride 4):  Y(4x4) = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A   (Lavin & Gray 2016, points
// 0, +-1, +-2, inf).  The 36 element-wise products are 36 independent GEMMs M_p[tile, cout] = V_p[tile, cin] U_p[cin, cout]:
// 36 MACs per 16 outputs = 2.25 per output, against 4 for F(2x2,3x3) (k_conv_wino8) and 9 for the direct form.  The f32
// matrix pipe is the bound of the whole step, so this is 1.78x fewer of its cycles than F(2x2,3x3).
// Accuracy: the transforms carry coefficients up to 8 and 1/24; simulated through the whole trunk in float32 against a
// float64 direct convolution (all 11 layers Winograd, synthetic AFx-Rep weights) the pooled features differ by 4e-7 of
// their maximum (F(2x2,3x3): 2e-7, float32 direct: 1e-7) -- far inside the 1e-4 bar; the per-layer parity tests hold the
// kernel to 5e-5 of the layer's output maximum against a float64 conv2d on random SIGNED inputs (the worst case for the
// cancellation in A^T M A: 3.3e-5 at cin = 2048, against 6e-6 for the direct kernel); on the real trunk (tools/
// trunk_accuracy.py, 10 s stereo, embeddings against the float64 oracle) all three algorithms sit at 1-2e-6.
//
// Workgroup = 32 tiles (512 output pixels) x 64 output channels x 36 positions, 4-channel chunks, 8 symmetric waves
// (512 threads, 2 per SIMD), built like k_conv_wino8 on the findings of tools/ubench/mfma_{coissue,interleave}.hip: a wave
// streaming f32 MFMAs starves the other waves of its SIMD, so there are no producer waves -- every wave interleaves its
// share of the production into its own MFMA stream:
//   MFMA   wave (g = w / 2, nh = w % 2): positions 9g .. 9g+8 x 32 tiles x channel half nh: 9 MFMA blocks = 144
//          accumulator VGPRs, 18 MFMAs per chunk, operands of the next three blocks read while three run;
//   U      [cin/4][36][pair][cout][2] pre-transformed weights: 36 scalar-addressed 1 KB LDS-DMA copies per chunk, 4-5 per wave;
//   patch  raw halo patch (4 channels = 16 B per pixel): LDS-DMA too, lanes without a pixel masked off through EXEC (the
//          two patch buffers are zero-filled once = the padding), issued first thing in the period BEFORE the one that
//          transforms it;
//   every period ends with s_waitcnt vmcnt(0): partial waits (vmcnt(n) leaving the youngest copies in flight) are NOT safe
//          here -- LDS-DMA loads, and register loads mixed with them, were observed to complete out of issue order (one
//          workgroup in ~10^2..10^4 consumed a U slab before it had landed; tools/conv_stress.py), so everything a
//          period issues has to land inside it and is issued in its first third;
//   V      item = (tile, row i of B^T d B, channel pair), 48 items per wave: the row combination with per-lane
//          coefficients (4 ds_read_b64 + 4 packed VALU per column), then the column combination (12 packed VALU) and 6
//          ds_write_b64.
// One barrier per chunk in a rotated loop (barrier(k) sits after the operand reads of chunk k's last block group).
// Epilogue: the 36 positions of an output meet in three passes over the position rows {1,2}, {3,4}, {0,5}: the owning
// waves write those accumulators to LDS (12 x 32 x 64 floats), thread (tile, channel quad) reduces each row over j
// (column half of A^T M A) and accumulates its contribution to the 4x4 outputs; then BN + ReLU (+ pool) and 16-byte
// stores into the channel-blocked activation layout.
#include "conv_layout.h"

#include <cstdlib>
#include <cstring>

namespace stito {

#ifndef W43_ABL
#ifndef W43_CLK
#define W43_CLK 0  // measurement build (tools/ab_build.sh clk -DW43_CLK=1): the workgroup in the middle of the grid reads s_memtime (shader
                   // clock) and s_memrealtime (constant 100 MHz) at its start and end, the launcher prints the ratio = the clock the part
                   // really ran at under this kernel; 0 in every build that ships
#endif
#if W43_CLK
#define W43_CLK_BEGIN()                                                                                                 \
    const bool clk_on = g.clk != nullptr && blockIdx.x == (gridDim.x >> 1) && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0; \
    long long clk_c0 = 0, clk_r0 = 0;                                                                                   \
    if (clk_on) { clk_c0 = (long long)__builtin_readcyclecounter(); clk_r0 = (long long)__builtin_amdgcn_s_memrealtime(); }
#define W43_CLK_END()                                                                                                   \
    if (clk_on && (threadIdx.x & 63) == 0) {                                                                            \
        g.clk[0] = clk_c0; g.clk[1] = (long long)__builtin_readcyclecounter();                                          \
        g.clk[2] = clk_r0; g.clk[3] = (long long)__builtin_amdgcn_s_memrealtime();                                      \
    }
#else
#define W43_CLK_BEGIN()
#define W43_CLK_END()
#endif
#define W43_ABL 0  // timing-experiment bit mask (1 no transform, 2 no U copies, 4 no patch copies, 8 no operand reads); 0 in every build that ships
#endif
static constexpr int W43_THREADS = 512;
static constexpr int W43_K = 4;                          // input channels per chunk
static constexpr int W43_U = 36 * 64 * W43_K;            // floats: [pos][cout][4]
static constexpr int W43_V = 36 * 32 * W43_K;            // floats: [pos][tile][4]
static constexpr int W43_BUF = W43_U + W43_V;
static constexpr int W43_XT = 68;                        // exchange: floats per tile row ([pos][tile][64 cout], +4 pad)

// split-precision streaming kernel (k_conv_wino43s): a slab = 8 position slots x 16 input channels of f16 halves hi + lo
static constexpr int S43_VPART = 8 * 2048;               // bytes: per slot [hi | lo][channel octet 0, 1][32 tiles][8 f16]
static constexpr int S43_UPART = 8 * 4096;               // bytes: per slot [hi | lo][channel octet 0, 1][64 couts][8 f16]
static constexpr int S43_SLAB = S43_VPART + S43_UPART;   // 48 KB, three of them in LDS
// two-sweep kernel (k_conv_wino43s2): a slab = 6 positions (one row of the 6 x 6) x 16 input channels x (64 tiles | 64 couts)
static constexpr int S43B_PART = 6 * 4096;               // bytes of the input half = bytes of the weight half of a slab
static constexpr int S43B_SLAB = 2 * S43B_PART;          // 48 KB
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
// Stores of the transform passes' V slabs (2 - 4.5 GB per layer, written once, read back from HBM by the convolution that follows):
// nontemporal, so that they stream past the L2 -- measured -0.25 ms per step.  (The same on the layer outputs and on the
// first conv's 7.9 GB: +1.6 / +0.7 ms per step, rejected.)
template <class T>
__device__ __forceinline__ void w43_store_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }

struct Wino43Geom {
    int S, H, W, Cin, Cout;
    int TR, TC;        // tile rows per stream / tile columns that produce output
    int64_t VTR;       // S * TR
    int n_col_blocks;
    int Ho, Wo;        // output map (pooled when POOL)
    int PR;            // halo patch rows
    int n_cgroups;     // MODE 2 only: workgroups that share a pixel block's chunks between them (1 otherwise)
    int n_mblocks;     // pixel blocks (MODE 1: the grid is padded to whole XCD rounds)
    int ct_group;      // MODE 1: channel tiles that run side by side on one XCD (a power of two dividing Cout / 64, <= 32)
    int b0;            // k_conv_wino43s: index of the launch's first workgroup item (a layer launched as two grids on two streams: 0 and the split point)
    int xcd_m;         // k_conv_wino43s3: the 8 XCDs as xcd_m pixel-block classes x 8 / xcd_m channel-tile ranges (8 = the other kernels' order)
    FDiv fH, fTR, fNCB, fNT;  // H, TR, n_col_blocks, Cout / 64 as launch-constant divisors (fdiv)
    long long *trace;  // TRACE instantiation only
    long long *clk;    // W43_CLK builds only
    unsigned *amax_out;  // or NULL: per stream, the largest output of this layer as a bit pattern (atomicMax; zeroed by the caller):
                         // what the split-precision kernels scale the NEXT layer's transformed input by (w43s_vscale)
};

// LDS layout of a halo patch buffer (16-byte pixels = the 4 channels of the chunk).  A transform read fetches, for 16 tiles
// at a time, the pixel (4 tr + k, 4 tc + l) of each tile: with the pixels stored row by row those addresses are 64 bytes
// apart along a tile row, so 32 lanes fall on 4..8 bank groups (measured: the LDS, not the MFMA pipe, set the period).
// The buffer is therefore split into 16 phase planes (pr % 4, pc % 4); inside a plane the pixel (pr / 4, pc / 4) of the 16
// tiles of a read group are consecutive (or 8 / 4 / 2 apart) -- every LDS-DMA lane can fetch any pixel, so the permutation
// is free:   slot(pr, pc) = ((pr % 4) * 4 + pc % 4) * PLANE + pos(pr / 4, pc / 4),
//   TTW = 8: pos = (pr/4) * 9 + pc/4 (one 2-way conflict in 16 tiles);  TTW <= 4: pos = (pc/4) * RH + pr/4 with RH = 12, 24, 34
//   (RH = 16 / TTW mod 16, so the TTW tile columns of a group interleave without collision).
template <int TTW>
struct W43Patch {
    static constexpr bool ROWMAJOR = TTW == 8;
    static constexpr int CW = 9;                                                // TTW == 8: positions per plane row
    static constexpr int NR = 6;                                                // TTW == 8: plane rows (PR <= 24)
    static constexpr int RH = TTW == 4 ? 12 : TTW == 2 ? 24 : 34;              // TTW <= 4: positions per plane column
    static constexpr int PLANE = ROWMAJOR ? NR * CW : (TTW + 1) * RH;          // slots per plane
    static constexpr int SLOTS = 16 * PLANE;
    static constexpr int NPL = (SLOTS + W43_THREADS - 1) / W43_THREADS;        // LDS-DMA instructions per wave per chunk
    static constexpr int PFL = SLOTS * W43_K;                                   // floats per patch buffer
    static constexpr int MAXPR4 = ROWMAJOR ? NR : RH;                           // (PR + 3) / 4 must not exceed this
    __host__ __device__ static constexpr int cstep() { return ROWMAJOR ? 1 : RH; }  // slots per tile column
    __device__ static int slot(int pr, int pc) {
        const int pos = ROWMAJOR ? (pr >> 2) * CW + (pc >> 2) : (pc >> 2) * RH + (pr >> 2);
        return ((pr & 3) * 4 + (pc & 3)) * PLANE + pos;
    }
    // float offset of column l (0..5) of a tile relative to its column 0
    __host__ __device__ static constexpr int coloff(int l) { return ((l & 3) * PLANE + (l >> 2) * cstep()) * W43_K; }
};

// -a as fma operand helper: (T3 - T1) etc. stay packed
#define P2(v, h) __builtin_shufflevector(v, v, 2 * (h), 2 * (h) + 1)

// v_pk_fma_f32 with one packed operand taken from ONE half of a register pair for both results (op_sel / op_sel_hi):
// d = x * c.lo + y   and   d = x * c.hi + y.  The row-combination coefficients are per lane (the lane's row i), two per pair.
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 x, f32x2 c, f32x2 y) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(c), "v"(y));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 x, f32x2 c, f32x2 y) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(c), "v"(y));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul_lo(f32x2 x, f32x2 c) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(c));
    return d;
}

// Split-precision kernel: power-of-two scale of a stream's transformed input, from the bit pattern of the stream's largest
// activation (>= 0, after ReLU).  |B^T d B| <= 100 max|d| (the rows of B^T sum to at most 10 in magnitude), and with
// amax < 2^e the scale 2^(8 - e) keeps every scaled element below 100 / 128 * 2^15: no f16 overflow, whatever the data.
__host__ __device__ __forceinline__ float w43s_vscale(unsigned amax_bits) {
    int e = (int)((amax_bits >> 23) & 0xff) - 126;  // amax = f * 2^e, f in [0.5, 1)
    if (amax_bits == 0u) e = 8;                       // all-zero stream: scale 1
    e = e < -40 ? -40 : (e > 60 ? 60 : e);            // (a denormal maximum or an inf / nan bit pattern: stay finite)
    return __builtin_ldexpf(1.0f, 8 - e);
}

// Largest of the thread's stored outputs (>= 0 after ReLU: the bit patterns order like the values), reduced over the 16
// lanes that share a tile (one DPP row: lanes differ in the channel quad), one atomicMax per tile into amax_out[stream].
__device__ __forceinline__ void w43_amax_out(unsigned *amax_out, int s, unsigned m) {
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, true));  // row_ror:8
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, true));  // row_ror:4
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x122, 0xf, 0xf, true));  // row_ror:2
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x121, 0xf, 0xf, true));  // row_ror:1
    // amax_out[s] only grows: a (possibly stale) load that already covers m makes the atomic unnecessary -- after a
    // stream's first few tiles nearly all of them (measured: 30 000 workgroups x 32 atomics onto 512 words cost a 6.5 ms layer
    // 3.7 ms before this check)
    // (agent-scope load = sc1: served by L2, where the atomics land; a plain load would keep hitting the CU's own L1 line)
    if ((threadIdx.x & 15) == 0 && m > __hip_atomic_load(amax_out + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_out + s, m);
}
__device__ __forceinline__ unsigned w43_max4(f32x4 v) {
    return max(max(__float_as_uint(v[0]), __float_as_uint(v[1])), max(__float_as_uint(v[2]), __float_as_uint(v[3])));
}

// ---- epilogue (shared by k_conv_wino43 and k_conv_wino43s) ---------------------------------------------------------
// Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  Three passes over the position rows
// {1,2}, {3,4}, {0,5} (the pairs whose contributions share sums and differences); reader thread = (tile, channel
// quad) keeps the 4x4 outputs of its 4 channels (64 VGPRs).  Packed arithmetic on register halves throughout.
// Wave (pg, nh) owns the accumulators of positions 9 pg .. 9 pg + 8 x 32 tiles x channels nh * 32 .. + 31.
// SPLIT: the accumulators carry the power-of-two operand scales of the split-precision kernel: 1 / (u_scale * v_scale(stream))
// is folded into the BN scale (exact: powers of two).
#define W43_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
template <int TTW, bool POOL, bool SPLIT>
__device__ __forceinline__ void w43_epilogue(float *smem, f32x16 (&acc)[9], const int tid, const int pg, const int nh,
                                             const Wino43Geom &g, const int n0, const int vtr0, const int tc0,
                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                             float *__restrict__ out, const float u_inv, const unsigned *__restrict__ amax) {
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
#define A4(a, b) __builtin_shufflevector(pk_add(P2(a, 0), P2(b, 0)), pk_add(P2(a, 1), P2(b, 1)), 0, 1, 2, 3)
#define S4(a, b) __builtin_shufflevector(pk_sub(P2(a, 0), P2(b, 0)), pk_sub(P2(a, 1), P2(b, 1)), 0, 1, 2, 3)
#define F4(c2, a, b) __builtin_shufflevector(pk_fma(c2, P2(a, 0), P2(b, 0)), pk_fma(c2, P2(a, 1), P2(b, 1)), 0, 1, 2, 3) /* c a + b */
    float *xch = smem;                      // [12 positions of the pass][32 tiles][W43_XT]
    constexpr int XP = 32 * W43_XT;
    const int e_quad = tid & 15, e_tile = tid >> 4;
    const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k8 = {8.f, 8.f};
    f32x4 Yo[4][4];
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        W43_BARRIER()  // the main loop's (or the previous pass's) LDS reads are done
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int p = 9 * pg + q;  // wave-uniform
            // exchange slot of position p in this pass: rows {1,2} -> p - 6, rows {3,4} -> p - 18, rows {0,5} -> p or p - 24
            const int slot = pass == 0 ? p - 6 : pass == 1 ? p - 18 : (p < 6 ? p : p - 24);
            const bool mine = pass == 0 ? (p >= 6 && p < 18) : pass == 1 ? (p >= 18 && p < 30) : (p < 6 || p >= 30);
            if (mine) {
                float *xp = xch + slot * XP + nh * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int trow = (r & 3) + 8 * (r >> 2) + 4 * half;
                    xp[trow * W43_XT] = acc[q][r];
                }
            }
        }
        W43_BARRIER()
        f32x4 Z[2][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            f32x4 m[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) m[j] = *(const f32x4 *)(xch + (ii * 6 + j) * XP + e_tile * W43_XT + e_quad * 4);
            // column half: Z[c] = sum_j M[i][j] A[j][c]
            const f32x4 s12 = A4(m[1], m[2]), d12 = S4(m[1], m[2]), s34 = A4(m[3], m[4]), d34 = S4(m[3], m[4]);
            Z[ii][0] = A4(A4(m[0], s12), s34);
            Z[ii][1] = F4(k2, d34, d12);
            Z[ii][2] = F4(k4, s34, s12);
            Z[ii][3] = A4(F4(k8, d34, d12), m[5]);
        }
        // row half: Y[r][c] += A^T[r][i] Z_i[c]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (pass == 0) {         // rows 1, 2: A^T columns (1,1,1,1) and (1,-1,1,-1); first pass: initialises Y
                const f32x4 sm = A4(Z[0][c], Z[1][c]), df = S4(Z[0][c], Z[1][c]);
                Yo[0][c] = sm; Yo[1][c] = df; Yo[2][c] = sm; Yo[3][c] = df;
            } else if (pass == 1) {  // rows 3, 4: (1,2,4,8) and (1,-2,4,-8)
                const f32x4 sm = A4(Z[0][c], Z[1][c]), df = S4(Z[0][c], Z[1][c]);
                Yo[0][c] = A4(Yo[0][c], sm); Yo[1][c] = F4(k2, df, Yo[1][c]); Yo[2][c] = F4(k4, sm, Yo[2][c]); Yo[3][c] = F4(k8, df, Yo[3][c]);
            } else {                 // rows 0, 5: (1,0,0,0) and (0,0,0,1)
                Yo[0][c] = A4(Yo[0][c], Z[0][c]); Yo[3][c] = A4(Yo[3][c], Z[1][c]);
            }
        }
    }
#undef A4
#undef S4
#undef F4
    // BN + ReLU (+ 2x2 average pool), 16-byte stores (4 channels) into NC8HW8
    {
        const int co = n0 + e_quad * 4;
        f32x4 sc = *(const f32x4 *)(scale + co);
        const f32x4 sh = *(const f32x4 *)(shift + co);
        const int vtr = vtr0 + e_tile / TTW;
        const int tc = tc0 + e_tile % TTW;
        unsigned smax = 0;
        int sidx = 0;
        if (vtr < g.VTR && tc < g.TC) {
            int tr;
            const int s = fdiv(vtr, g.fTR, tr);
            if constexpr (SPLIT) sc = sc * (u_inv / w43s_vscale(amax[s]));
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) Yo[r][c] = __builtin_elementwise_max(Yo[r][c] * sc + sh, (f32x4)(0.0f));
            unsigned mx = 0;
            if (POOL) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const int oh = 2 * tr + pr, ow = 2 * tc + pc;
                        if (oh < g.Ho && ow < g.Wo) {
                            const f32x4 v = (((Yo[2 * pr][2 * pc] + Yo[2 * pr][2 * pc + 1]) + Yo[2 * pr + 1][2 * pc]) + Yo[2 * pr + 1][2 * pc + 1]) * 0.25f;
                            *(f32x4 *)(out + act_off(s, co, oh, ow, g.Cout, g.Ho, g.Wo)) = v;
                            mx = max(mx, w43_max4(v));
                        }
                    }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int hh = 4 * tr + r, ww = 4 * tc + c;
                        if (hh < g.H && ww < g.W) {
                            *(f32x4 *)(out + act_off(s, co, hh, ww, g.Cout, g.H, g.W)) = Yo[r][c];
                            mx = max(mx, w43_max4(Yo[r][c]));
                        }
                    }
            }
            smax = mx;
            sidx = s;
        }
        // (outside the branch: the DPP reduction reads all 16 lanes of the tile's row; a tile is valid or not as a whole)
        if (g.amax_out != nullptr) w43_amax_out(g.amax_out, sidx, smax);
    }
}

// MODE 0: the whole convolution (patch -> V in the workgroup).  The workgroups of one pixel block that differ only in their
// 64 output channels all repeat the same input transform (8 .. 32 times for Cout >= 512), and it is paid in the same
// ALUs the f32 MFMAs run on; for those layers the transform is hoisted: MODE 2 runs the production pipeline alone (one
// workgroup per pixel block, no weights, no MFMAs) and writes every chunk's V slab -- in exactly the LDS layout -- to HBM,
// MODE 1 is the convolution with `in` = those slabs: V(k) arrives by LDS-DMA like U(k), nothing is transformed.
template <int TTW, bool POOL, bool TRACE, int MODE = 0>
__global__ __launch_bounds__(W43_THREADS, MODE >= 2 ? 4 : 2) void k_conv_wino43(const float *__restrict__ in, const float *__restrict__ upk,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, float *__restrict__ out,
                                                              Wino43Geom g) {
    constexpr int TTH = 32 / TTW;
    constexpr int PWC = 4 * TTW + 2;  // patch columns
    W43_CLK_BEGIN()
    using PL = W43Patch<TTW>;
    constexpr int NPL = PL::NPL;      // LDS-DMA instructions of patch per wave per chunk
    constexpr int PFL = PL::PFL;      // floats per patch buffer
    constexpr bool PREV = MODE == 1, VOUT = MODE >= 2, V16 = MODE == 3, V16B = MODE == 4, V16C = MODE == 5;  // 3 / 4 / 5: f16 slabs of k_conv_wino43s / s2 / s3
    // the transform passes (MODE >= 2) have no weights: their two buffers hold V only (36 KB + patches = 64 KB of LDS, and with
    // <= 128 VGPRs two workgroups share a CU: twice the HBM requests in flight of a pass that does nothing but move data)
    constexpr int BUF = VOUT ? W43_V : W43_BUF, UO = VOUT ? 0 : W43_U;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = VOUT ? g.n_cgroups : g.Cout / 64;
    int ct_;
    const int bidx = (int)blockIdx.x + (MODE == 0 ? g.b0 : 0);   // (MODE 0 may be launched as several grids: w43_multi_queue_launch)
    int m_blk = VOUT ? bidx / n_tiles : fdiv(bidx, g.fNT, ct_);  // channel tile fastest: the workgroups sharing a halo patch run side by side
    if (VOUT) ct_ = bidx % n_tiles;
    int n0 = VOUT ? 0 : ct_ * 64;
    if constexpr (PREV) {
        // MODE 1 streams 18 KB of V and 36 KB of U per period through L2, and workgroup b runs on XCD b % 8 (own L2 each).
        // With the channel tile fastest the 8+ sharers of a V slab sit on 8 different XCDs: every slab is fetched 8 times
        // (PMC: 22.8 GB for conv_block4.conv2 against 9.2 GB for MODE 0).  Here the 32 workgroups an XCD runs at a time are
        // 8 pixel blocks x 4 channel tiles: 8 V streams + 4 U streams = 288 KB per period per XCD, the minimum of 18 a + 36 b
        // over a b = 32 (612 KB before).  Pixel blocks are dealt round-robin to the XCDs; padding workgroups leave at once.
        const int b = blockIdx.x, xcd = b & 7, j = b >> 3, r = j & 31, gi = j >> 5;
        const int a = g.ct_group, n_ctg = n_tiles / a;          // a channel tiles x 32 / a pixel blocks per XCD round
        const int ct = (gi % n_ctg) * a + (r % a);
        m_blk = ((gi / n_ctg) * (32 / a) + r / a) * 8 + xcd;
        n0 = ct * 64;
        if (m_blk >= g.n_mblocks) return;
    }
    int cb;
    const int rb = fdiv(m_blk, g.fNCB, cb);
    const int vtr0 = rb * TTH;  // first virtual tile row (s * TR + tr) of the block
    const int tc0 = cb * TTW;
    // MODE 2: this workgroup transforms chunks c_base .. c_base + n_chunks - 1 of the pixel block (an even count >= 2)
    const int n_chunks_all = g.Cin / W43_K;
    const int n_chunks = VOUT ? n_chunks_all / g.n_cgroups : n_chunks_all;
    const int c_base = VOUT ? ct_ * n_chunks : 0;
    float *patch0 = smem + 2 * BUF;
    int tr0_;
    const int s0_ = fdiv(vtr0, g.fTR, tr0_);
    const int iv_lo = s0_ * g.H + 4 * tr0_ - 1;  // input virtual row (s*H + h) of patch row 0

    // MODE 3: the V slabs leave as scaled f16 halves hi + lo in the slab order of k_conv_wino43s.  Thread -> tile tid % 32 for
    // all of its (position, tile) items; the scale belongs to the tile's stream (`scale` carries the streams' maxima).
    const int v16_tile = tid & 31;
    float v16_s = 1.0f;
    if constexpr (V16 || V16B || V16C) {
        int vt_ = vtr0 + v16_tile / TTW, tr_;
        vt_ = vt_ < (int)g.VTR ? vt_ : (int)g.VTR - 1;
        v16_s = w43s_vscale(((const unsigned *)scale)[fdiv(vt_, g.fTR, tr_)]);
    }
    const int n_slabs16 = V16B ? (g.Cin >> 4) * 3 : (g.Cin >> 5) * 9;

#define W43_STAMP(SLOT)                                                                                 \
    if (TRACE && (blockIdx.x & 255) == 100 && (blockIdx.x >> 8) < 8 && lane == 0 && (wv & 3) == 0)      \
        g.trace[((blockIdx.x >> 8) * 2 + (wv >> 2)) * 16 + (SLOT)] = (long long)__builtin_readcyclecounter();
    W43_STAMP(0)

    // ---- MFMA role ----------------------------------------------------------------------------------------
    const int half = lane >> 5, l31 = lane & 31;
    const int nh = wv & 1, pg = wv >> 1;
    // operands per channel pair (= lane half), 8 bytes per row so that a half-wave reads 256 contiguous bytes:
    // V[p][cp][(tile + 16 cp) % 32][2] (the rotation keeps the transform's stores of both pairs on disjoint banks), U[p][cp][cout][2]
    const int a_off = W43_U + (9 * pg) * 32 * W43_K + half * 64 + ((l31 + 16 * half) & 31) * 2;
    const int b_off = (9 * pg) * 64 * W43_K + half * 128 + (nh * 32 + l31) * 2;

    // transform-item registers (set up AFTER the first copies are in flight: the index arithmetic hides behind their latency)
    int roff[4];     // float offsets (inside a patch buffer) of the four input rows the lane's row i combines, at tile column 0
    f32x2 cab, ccd;  // their coefficients (a, b), (c, d); 0 for rows outside the map / stream
    int vdst;
    // ---- U slab copies: position ii = wv + 8 j, 1 KB each --------------------------------------------------
    // packed weights [cin/4][36][2 pairs][cout][2]: lanes 0..31 fetch this block's 64 channels of pair 0, lanes 32..63 of pair 1
    const float *u_base = upk + (int64_t)n0 * 2;
    const int64_t u_pos_stride = (int64_t)g.Cout * W43_K, u_chunk_stride = 36 * u_pos_stride;  // floats
    const unsigned u_voff = (unsigned)((lane & 31) * 16 + (lane >> 5) * g.Cout * 8);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const float *v_base = in + (int64_t)m_blk * n_chunks_all * W43_V;  // PREV: this pixel block's V slabs

    // ---- halo patch staging by LDS-DMA: slot q = tid + 512 j of the permuted layout (W43Patch) holds pixel (pr, pc) ------
    int h_first;  // row of the first patch row that exists, inside its stream
    const int s_first = fdiv(iv_lo < 0 ? 0 : iv_lo, g.fH, h_first);
    const int64_t plane8 = (int64_t)g.H * g.W * 8;  // floats per 8-channel plane of one stream
    const float *p_base = in + act_off(s_first, 0, 0, 0, g.Cin, g.H, g.W) + (int64_t)(c_base >> 1) * plane8;
    unsigned p_off[NPL];                             // per-lane byte offset from the chunk's base
    uint64_t p_mask[NPL];                            // lanes of this wave that have a pixel (wave-uniform)
    if constexpr (!PREV) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int q = tid + W43_THREADS * j;
            const int plane = q / PL::PLANE, pos = q % PL::PLANE;
            const int pr4 = PL::ROWMAJOR ? pos / PL::CW : pos % PL::RH, pc4 = PL::ROWMAJOR ? pos % PL::CW : pos / PL::RH;
            const int pr = 4 * pr4 + (plane >> 2), pc = 4 * pc4 + (plane & 3);
            const int iv = iv_lo + pr;
            const int w = 4 * tc0 - 1 + pc;
            const bool ok = q < PL::SLOTS && pr < g.PR && pc < PWC && iv >= 0 && iv < g.S * g.H && w >= 0 && w < g.W;
            int hq_;
            const int sq_ = fdiv(ok ? iv : 0, g.fH, hq_);
            const int s_ = ok ? sq_ : s_first, h_ = ok ? hq_ : 0, w_ = ok ? w : 0;
            p_off[j] = (unsigned)(((int64_t)(s_ - s_first) * (g.Cin >> 3) * plane8 + ((int64_t)h_ * g.W + w_) * 8) * 4);
            p_mask[j] = __builtin_amdgcn_ballot_w64(ok);
        }
    }
    const unsigned lds_patch = lds0 + (unsigned)(2 * BUF) * 4u;  // byte address of patch buffer 0

#define W43_COPY_U1(CH, BOFF, J_) /* callers guarantee CH < n_chunks */                                  \
    {                                                                                                    \
        const int ii = wv + 8 * (J_) < 36 ? wv + 8 * (J_) : wv + 8 * (J_) - 8;                           \
        glds16_m0(u_base + (int64_t)(CH) * u_chunk_stride + ii * u_pos_stride, u_voff,                   \
                  lds0 + (unsigned)((BOFF) + ii * 64 * W43_K) * 4u);                                     \
    }
// PREV: V slab of chunk CH (18 KB, already in the LDS layout) -> V region of buffer BOFF: 18 copies of 1 KB, J = 0..2 per wave
#define W43_COPY_V1(CH, BOFF, J_)                                                                        \
    {                                                                                                    \
        const int ii = wv + 8 * (J_) < 18 ? wv + 8 * (J_) : wv + 8 * (J_) - 8;                           \
        glds16_m0(v_base + (int64_t)(CH) * W43_V + ii * 256, (unsigned)lane * 16u,                       \
                  lds0 + (unsigned)((BOFF) + W43_U + ii * 256) * 4u);                                    \
    }
// VOUT: V(k) (complete in the current buffer since the last barrier) -> HBM
#define W43_STORE_V(K_, CUR)                                                                             \
    {                                                                                                    \
        f32x4 *vo_ = (f32x4 *)(out + ((int64_t)m_blk * n_chunks_all + c_base + (K_)) * W43_V);                        \
        const f32x4 *vs_ = (const f32x4 *)(smem + (CUR) + UO);                                           \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                  \
            const int e_ = tid + W43_THREADS * j;                                                        \
            if (e_ < W43_V / 4) vo_[e_] = vs_[e_];                                                       \
        }                                                                                                \
    }
// V16 / V16B: V(k) -> scaled f16 halves hi + lo in the slab order of k_conv_wino43s / k_conv_wino43s2.  Two consecutive chunks
// (channels 8 m .. 8 m + 7: one 16-byte group of an operand row) leave together: the even chunk's values wait in registers
// (v16_h) for the odd one, so that every store is a whole 16-byte group and a half-wave writes 512 contiguous bytes
// (8-byte stores of single chunks: 3.4 TB/s for the transform pass; measured).  c_base and the chunk count are even.
//   V16:  position p = 9 pg + q sits in slab 9 (c / 32) + e / 2, slot 2 pg + e % 2 with e = 9 ((c % 32) / 16) + q (the order in
//         which k_conv_wino43s walks a wave's nine blocks twice per 32 channels); inside a slot [hi | lo][(c % 16) / 8][tile][c % 8];
//   V16B: 64-tile workgroups = pixel-block pairs, two sweeps over 18 positions each.  Position row i = p / 6 belongs to sweep
//         i >= 3 and is that sweep's local row [1, 2, 0][i] resp. i - 3 (rows {1, 2} and {3, 4} -- the pairs whose
//         contributions share sums and differences -- are local rows 0, 1); slab 3 (c / 16) + local row, position column
//         j = p % 6 at 4 KB each: [hi | lo][(c % 16) / 8][64 tiles][c % 8].
//   V16C (B_ == 2): 128-tile workgroups = pixel-block QUADS, six sweeps of one position row each (k_conv_wino43s3): quad
//         m_blk / 4, row i = p / 6, period 2 (c / 16) + (j >= 3) with j = p % 6: 24 KB, position j % 3 at 8 KB each:
//         [hi | lo][(c % 16) / 8][128 tiles = 32 (m_blk % 4) + tile][c % 8].
#define W43_STORE_V16X(K_, CUR, B_)                                                                      \
    {                                                                                                    \
        const int kc_ = c_base + (K_);                                                                   \
        const float *vs_ = smem + (CUR) + UO;                                                            \
        char *vo_ = (B_) == 2 ? (char *)out + ((int64_t)(m_blk >> 2) * 6 * (g.Cin >> 3) + 2 * (kc_ >> 2)) * S43B_PART + \
                               (((kc_ >> 1) & 1) * 128 + (m_blk & 3) * 32 + v16_tile) * 16                \
                  : (B_) ? (char *)out + ((int64_t)(m_blk >> 1) * 2 * n_slabs16 + 3 * (kc_ >> 2)) * S43B_PART + \
                               (((kc_ >> 1) & 1) * 64 + (m_blk & 1) * 32 + v16_tile) * 16                 \
                         : (char *)out + ((int64_t)m_blk * n_slabs16 + 9 * (kc_ >> 3)) * S43_VPART +     \
                               (((kc_ >> 1) & 1) * 32 + v16_tile) * 16;                                  \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                  \
            const int p_ = (tid >> 5) + 16 * j;                                                          \
            if (p_ < 36) {                                                                               \
                const f32x2 x01 = *(const f32x2 *)(vs_ + p_ * 128 + v16_tile * 2) * v16_s;               \
                const f32x2 x23 = *(const f32x2 *)(vs_ + p_ * 128 + 64 + ((v16_tile + 16) & 31) * 2) * v16_s; \
                if (!(kc_ & 1)) {                                                                        \
                    v16_h[j][0] = x01; v16_h[j][1] = x23;                                                \
                } else {                                                                                 \
                    const float x_[8] = {v16_h[j][0][0], v16_h[j][0][1], v16_h[j][1][0], v16_h[j][1][1], x01[0], x01[1], x23[0], x23[1]}; \
                    h8 hi_, lo_;                                                                         \
                    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) { hi_[e_] = (_Float16)x_[e_]; lo_[e_] = (_Float16)(x_[e_] - (float)hi_[e_]); } \
                    char *d_;                                                                            \
                    int lo_off_;                                                                         \
                    if ((B_) == 2) {                                                                     \
                        const int i_ = p_ / 6, j_ = p_ - 6 * i_;                                         \
                        d_ = vo_ + ((int64_t)i_ * (g.Cin >> 3) + (j_ >= 3)) * S43B_PART + (j_ % 3) * 8192; \
                        lo_off_ = 4096;                                                                  \
                    } else if (B_) {                                                                     \
                        const int i_ = p_ / 6, j_ = p_ - 6 * i_;                                         \
                        const int sw_ = i_ >= 3, lr_ = sw_ ? i_ - 3 : (i_ == 0 ? 2 : i_ - 1);            \
                        d_ = vo_ + ((int64_t)sw_ * n_slabs16 + lr_) * S43B_PART + j_ * 4096;             \
                        lo_off_ = 2048;                                                                  \
                    } else {                                                                             \
                        const int pg_ = p_ / 9, e_ = ((kc_ >> 2) & 1) * 9 + p_ - 9 * pg_;                \
                        d_ = vo_ + (int64_t)(e_ >> 1) * S43_VPART + (pg_ * 2 + (e_ & 1)) * 2048;         \
                        lo_off_ = 1024;                                                                  \
                    }                                                                                    \
                    w43_store_stream((h8 *)d_, hi_);                                                     \
                    w43_store_stream((h8 *)(d_ + lo_off_), lo_);                                         \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    }
#define W43_STORE_V16(K_, CUR) W43_STORE_V16X(K_, CUR, 0)
#define W43_STORE_V16B(K_, CUR) W43_STORE_V16X(K_, CUR, 1)
#define W43_STORE_V16C(K_, CUR) W43_STORE_V16X(K_, CUR, 2)
// patch(CH) -> patch buffer PB (0, 1): two masked LDS-DMA instructions per wave (pixels wv*64 + 512 j + lane)
#define W43_COPY_P(CH, PB)                                                                              \
    {                                                                                                   \
        const int cc_ = (CH) < n_chunks ? (CH) : n_chunks - 1;                                           \
        const float *pb_ = p_base + (int64_t)(cc_ >> 1) * plane8 + (cc_ & 1) * 4;                        \
        _Pragma("unroll") for (int j = 0; j < NPL; ++j) {                                                \
            uint64_t keep_;                                                                              \
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"  \
                         "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"                          \
                         : "=&s"(keep_)                                                                  \
                         : "v"(p_off[j]), "s"(pb_), "s"(lds_patch + (unsigned)((PB) * PFL + (wv * 64 + W43_THREADS * j) * W43_K) * 4u), \
                           "s"(p_mask[j])                                                                \
                         : "memory");                                                                    \
        }                                                                                                \
    }
// transform: row combination of column L (4 reads), T[L] = ((ca da + cb db) + cc dc) + cd dd
#define W43_T_RD(PBUF, L, R)                                                                            \
    {                                                                                                   \
        R[0] = *(const f32x2 *)((PBUF) + roff[0] + PL::coloff(L));                                       \
        R[1] = *(const f32x2 *)((PBUF) + roff[1] + PL::coloff(L));                                       \
        R[2] = *(const f32x2 *)((PBUF) + roff[2] + PL::coloff(L));                                       \
        R[3] = *(const f32x2 *)((PBUF) + roff[3] + PL::coloff(L));                                       \
    }
#define W43_T_ROW(L, R) tT[L] = pk_fma_hi(R[3], ccd, pk_fma_lo(R[2], ccd, pk_fma_hi(R[1], cab, pk_mul_lo(R[0], cab))));
// column combination (the same B^T along the columns) and stores: V[6 i + j] for j = 0..5
#define W43_T_COLS_A(VBOFF)                                                                             \
    {                                                                                                   \
        float *vb_ = smem + (VBOFF) + vdst;                                                              \
        const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, cm4 = {-4.f, -4.f};                             \
        *(f32x2 *)(vb_ + 0 * 32 * W43_K) = pk_fma(c4, tT[0], pk_fma(cm5, tT[2], tT[4]));                 \
        const f32x2 p_ = pk_fma(cm4, tT[2], tT[4]), q_ = pk_fma(cm4, tT[1], tT[3]);                      \
        *(f32x2 *)(vb_ + 1 * 32 * W43_K) = pk_add(p_, q_);                                               \
        *(f32x2 *)(vb_ + 2 * 32 * W43_K) = pk_sub(p_, q_);                                               \
    }
#define W43_T_COLS_B(VBOFF)                                                                             \
    {                                                                                                   \
        float *vb_ = smem + (VBOFF) + vdst;                                                              \
        const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, c2 = {2.f, 2.f}, cm2 = {-2.f, -2.f};            \
        const f32x2 s_ = pk_sub(tT[4], tT[2]), u_ = pk_sub(tT[3], tT[1]);                                \
        *(f32x2 *)(vb_ + 3 * 32 * W43_K) = pk_fma(c2, u_, s_);                                           \
        *(f32x2 *)(vb_ + 4 * 32 * W43_K) = pk_fma(cm2, u_, s_);                                          \
        *(f32x2 *)(vb_ + 5 * 32 * W43_K) = pk_fma(c4, tT[1], pk_fma(cm5, tT[3], tT[5]));                 \
    }
// MFMA operands of block group G (blocks 3G .. 3G+2 of this wave's nine) into register set S
#define W43_LOAD_OPS(S, SB, G)                                                                          \
    {                                                                                                   \
        _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_) {                                               \
            S##a[t_] = *(const f32x2 *)((SB) + a_off + (3 * (G) + t_) * 32 * W43_K);                     \
            S##b[t_] = *(const f32x2 *)((SB) + b_off + (3 * (G) + t_) * 64 * W43_K);                     \
        }                                                                                                \
    }
// The MFMA intrinsics have no side effects, so nothing ties them to the fences: with little else in a gap (MODE 1 has only
// copies there) hipcc lets them drift and clumps them behind the waits -- measured: the hoisted transform then buys nothing.
// An empty volatile asm that "touches" the accumulator before and after pins each one between the neighbouring pieces.
// (The MFMA itself as inline asm gave wrong sums: hipcc does not apply its MFMA hazard handling to opaque asm.)
#define W43_MFMA(S, G, T_, E_)                                                                          \
    if constexpr (!VOUT) {                                                                              \
        asm volatile("" : "+v"(acc[3 * (G) + (T_)]));                                                   \
        acc[3 * (G) + (T_)] = __builtin_amdgcn_mfma_f32_32x32x2f32(S##a[T_][E_], S##b[T_][E_], acc[3 * (G) + (T_)], 0, 0, 0); \
        asm volatile("" : "+v"(acc[3 * (G) + (T_)]));                                                   \
    }
#define W43_FENCE() __builtin_amdgcn_sched_barrier(0);
#define W43_GAP(S, G, T_, E_, WORK) W43_MFMA(S, G, T_, E_) WORK W43_FENCE()

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    f32x2 tT[6], rX[4], rY[4];
    f32x2 v16_h[3][2];  // V16 / V16B: the even chunk's values of this thread's items, waiting for the odd chunk
    f32x2 xa[3], xb[3], ya[3], yb[3], za[3], zb[3];  // z: PREV only

    // ---- prologue: patch(0), patch(1), U(0) by LDS-DMA, issued first; meanwhile every thread zeroes the slots of both patch
    // buffers that ITS copy lane never writes (= the padding: no overlap with any copy, so no barrier in between); V(0) from patch(0)
    if constexpr (PREV) {  // V(0), U(0) by LDS-DMA; nothing to transform
        W43_COPY_V1(0, 0, 0) W43_COPY_V1(0, 0, 1) W43_COPY_V1(0, 0, 2)
        W43_COPY_U1(0, 0, 0) W43_COPY_U1(0, 0, 1) W43_COPY_U1(0, 0, 2) W43_COPY_U1(0, 0, 3) W43_COPY_U1(0, 0, 4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    W43_COPY_P(0, 0)
    if (!VOUT) { W43_COPY_U1(0, 0, 0) W43_COPY_U1(0, 0, 1) W43_COPY_U1(0, 0, 2) W43_COPY_U1(0, 0, 3) W43_COPY_U1(0, 0, 4) }
    W43_COPY_P(1, 1)
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int q = tid + W43_THREADS * j;
        if (q < PL::SLOTS && !((p_mask[j] >> lane) & 1)) {
            *(f32x4 *)(patch0 + q * W43_K) = (f32x4)(0.0f);
            *(f32x4 *)(patch0 + PFL + q * W43_K) = (f32x4)(0.0f);
        }
    }
    // ---- transform item: (row i of B^T d B, tile, channel pair).  Twelve units (row i, tile half) of 16 tiles x 2 pairs:
    // lanes 0..31 of wave w carry unit w, lanes 32..47 a quarter of unit 8 + w/2, lanes 48..63 mirror lanes 32..47 (same reads,
    // same values, same addresses written: no divergent branch in the loop, and every wave carries the same 48 items).
    if constexpr (!PREV) {
        int ti, tile, cp;
        if (lane < 32) {
            ti = wv >> 1; tile = (wv & 1) * 16 + (lane & 15); cp = lane >> 4;
        } else {
            const int u = 8 + (wv >> 1), l = lane & 15;
            ti = u >> 1; tile = (u & 1) * 16 + (wv & 1) * 8 + (l & 7); cp = l >> 3;
        }
        // B^T (Lavin & Gray):  row 0: 4 d0 - 5 d2 + d4        row 1: -4 d1 - 4 d2 + d3 + d4   row 2: 4 d1 - 4 d2 - d3 + d4
        //                      row 3: -2 d1 - d2 + 2 d3 + d4  row 4: 2 d1 - d2 - 2 d3 + d4     row 5: 4 d1 - 5 d3 + d5
        int kr[4];
        float cf[4];
        switch (ti) {
            case 0: kr[0] = 0; kr[1] = 2; kr[2] = 4; kr[3] = 4; cf[0] = 4.f; cf[1] = -5.f; cf[2] = 1.f; cf[3] = 0.f; break;
            case 1: kr[0] = 1; kr[1] = 2; kr[2] = 3; kr[3] = 4; cf[0] = -4.f; cf[1] = -4.f; cf[2] = 1.f; cf[3] = 1.f; break;
            case 2: kr[0] = 1; kr[1] = 2; kr[2] = 3; kr[3] = 4; cf[0] = 4.f; cf[1] = -4.f; cf[2] = -1.f; cf[3] = 1.f; break;
            case 3: kr[0] = 1; kr[1] = 2; kr[2] = 3; kr[3] = 4; cf[0] = -2.f; cf[1] = -1.f; cf[2] = 2.f; cf[3] = 1.f; break;
            case 4: kr[0] = 1; kr[1] = 2; kr[2] = 3; kr[3] = 4; cf[0] = 2.f; cf[1] = -1.f; cf[2] = -2.f; cf[3] = 1.f; break;
            default: kr[0] = 1; kr[1] = 3; kr[2] = 5; kr[3] = 5; cf[0] = 4.f; cf[1] = -5.f; cf[2] = 1.f; cf[3] = 0.f; break;
        }
        const int vtr = vtr0 + tile / TTW, tcl = tile % TTW;
        int tr;
        const int s_ = fdiv(vtr, g.fTR, tr);
        const int pc0 = s_ * g.H + 4 * tr - 1 - iv_lo;  // patch row of this tile's first input row
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int hh = 4 * tr - 1 + kr[x];
            // rows outside the map / stream contribute nothing: coefficient 0 on a row that exists in the patch (finite data)
            const bool ok = vtr < g.VTR && hh >= 0 && hh < g.H;
            if (!ok) cf[x] = 0.f;
            const int prow = ok ? pc0 + kr[x] : 0;
            roff[x] = PL::slot(prow, 4 * tcl) * W43_K + cp * 2;
        }
        cab = (f32x2){cf[0], cf[1]};
        ccd = (f32x2){cf[2], cf[3]};
        vdst = UO + (ti * 6) * 32 * W43_K + cp * 64 + ((tile + 16 * cp) & 31) * 2;  // V[6 ti + j][cp][(tile + 16 cp) % 32]
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W43_BARRIER()
    W43_T_RD(patch0, 0, rX) W43_T_RD(patch0, 1, rY)
    W43_T_ROW(0, rX) W43_T_RD(patch0, 2, rX)
    W43_T_ROW(1, rY) W43_T_RD(patch0, 3, rY)
    W43_T_ROW(2, rX) W43_T_RD(patch0, 4, rX)
    W43_T_ROW(3, rY) W43_T_RD(patch0, 5, rY)
    W43_T_ROW(4, rX) W43_T_ROW(5, rY)
    W43_T_COLS_A(0)
    W43_T_COLS_B(0)
    }
    W43_BARRIER()  // B(-1): V(0), U(0) complete
    W43_STAMP(1)

// One period: chunk k's block groups 0 and 1 run here, preceded by group 2 of chunk k-1 from registers (not FIRST).
// S0 holds group 2 of the previous chunk on entry; the sets alternate S0, S1, S0 and the next period starts on S1.
// MORE: chunk k+1 exists -- its production (U copies, transform) and the patch copy of chunk k+2 are spread over the first
// MFMA gaps; patch(c) lives in buffer c % 2.
#define W43_OPS(X) if (!VOUT && !(W43_ABL & 8)) { X }
#define W43_UCP(X) if (!VOUT && MORE_ && !(W43_ABL & 2)) { X }
#define W43_TRF(X) if (!PREV && MORE_ && !(W43_ABL & 1)) { X }
#define W43_PERIOD4(P2, G0, G1, N2, FIRST, MORE)                                                                 \
    {                                                                                                   \
        constexpr bool MORE_ = MORE;                                                                     \
        const int cur = (k & 1) * BUF, nxt = BUF - cur;                                                  \
        const float *sb = smem + cur;                                                                    \
        const float *pb_r = patch0 + ((k + 1) & 1) * PFL;     /* patch(k+1); patch(k+2) goes where patch(k) was */ \
        if (V16) W43_STORE_V16(k, cur) else if (V16B) W43_STORE_V16B(k, cur) else if (V16C) W43_STORE_V16C(k, cur) else if (VOUT) W43_STORE_V(k, cur) \
        if (!(FIRST)) {                                                                                  \
            W43_GAP(P2, 2, 0, 0, if (MORE_ && !(W43_ABL & 4)) { if (PREV) { W43_COPY_V1(k + 1, nxt, 0) } else W43_COPY_P(k + 2, k & 1) }) \
            W43_GAP(P2, 2, 1, 0, W43_OPS(W43_LOAD_OPS(G0, sb, 0)) W43_UCP(W43_COPY_U1(k + 1, nxt, 0)))   \
            W43_GAP(P2, 2, 2, 0, W43_UCP(W43_COPY_U1(k + 1, nxt, 1)))                                    \
            W43_GAP(P2, 2, 0, 1, W43_UCP(W43_COPY_U1(k + 1, nxt, 2)))                                    \
            W43_GAP(P2, 2, 1, 1, W43_UCP(W43_COPY_U1(k + 1, nxt, 3)))                                    \
            W43_GAP(P2, 2, 2, 1, W43_UCP(W43_COPY_U1(k + 1, nxt, 4)))                                    \
        } else {                                                                                         \
            if (PREV) { W43_COPY_V1(k + 1, nxt, 0) W43_COPY_V1(k + 1, nxt, 1) W43_COPY_V1(k + 1, nxt, 2) } \
            else W43_COPY_P(k + 2, k & 1)                                                                \
            W43_OPS(W43_LOAD_OPS(G0, sb, 0))                                                             \
            W43_UCP(W43_COPY_U1(k + 1, nxt, 0) W43_COPY_U1(k + 1, nxt, 1) W43_COPY_U1(k + 1, nxt, 2)     \
                    W43_COPY_U1(k + 1, nxt, 3) W43_COPY_U1(k + 1, nxt, 4))                               \
            W43_FENCE()                                                                                  \
        }                                                                                                \
        W43_GAP(G0, 0, 0, 0, W43_OPS(W43_LOAD_OPS(G1, sb, 1)))                                           \
        W43_GAP(G0, 0, 1, 0, W43_TRF(W43_T_RD(pb_r, 0, rX) W43_T_RD(pb_r, 1, rY)) if (PREV && MORE_ && !(FIRST)) W43_COPY_V1(k + 1, nxt, 1)) \
        W43_GAP(G0, 0, 2, 0, if (PREV && MORE_ && !(FIRST)) W43_COPY_V1(k + 1, nxt, 2)) \
        W43_GAP(G0, 0, 0, 1, W43_TRF(W43_T_ROW(0, rX) W43_T_RD(pb_r, 2, rX)))                            \
        W43_GAP(G0, 0, 1, 1, W43_TRF(W43_T_ROW(1, rY) W43_T_RD(pb_r, 3, rY)))                            \
        W43_GAP(G0, 0, 2, 1, W43_TRF(W43_T_ROW(2, rX) W43_T_RD(pb_r, 4, rX)))                            \
        W43_GAP(G1, 1, 0, 0, W43_OPS(W43_LOAD_OPS(N2, sb, 2)))                                           \
        W43_GAP(G1, 1, 1, 0, W43_TRF(W43_T_ROW(3, rY) W43_T_RD(pb_r, 5, rY)))                            \
        W43_GAP(G1, 1, 2, 0, W43_TRF(W43_T_ROW(4, rX)))                                                  \
        W43_GAP(G1, 1, 0, 1, W43_TRF(W43_T_ROW(5, rY)))                                                  \
        W43_GAP(G1, 1, 1, 1, W43_TRF(W43_T_COLS_A(nxt)))                                                 \
        W43_GAP(G1, 1, 2, 1, W43_TRF(W43_T_COLS_B(nxt)))                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* U(k+1), patch(k+2) / V(k+1) landed (no partial waits: see the header) */ \
        W43_BARRIER()                                     /* B(k) */                                     \
        if (TRACE && k < 12) { W43_STAMP(4 + k) }                                                        \
    }

// two operand sets alternate (MODE 0 / 2: register budget); group 2 of the chunk waits in S1 for the next period, which names it S0
#define W43_PERIOD(S0, S1, FIRST, MORE) W43_PERIOD4(S0, S1, S0, S1, FIRST, MORE)

    {
        // n_chunks is even and >= 2 (Cin % 8 == 0).  Periods 0 .. n_chunks-2 produce the next chunk; the last one does not.
        int k = 0;
        if constexpr (PREV) {
            // one operand set per block group (there are registers to spare without the transform): every period is the same code
            W43_PERIOD4(z, x, y, z, true, true)
            for (k = 1; k + 2 < n_chunks; k += 2) {  // two periods per trip: the buffer parity is a compile-time constant
                W43_PERIOD4(z, x, y, z, false, true)
                ++k;
                W43_PERIOD4(z, x, y, z, false, true)
                --k;
            }
            W43_PERIOD4(z, x, y, z, false, false)  // k = n_chunks - 1
            W43_MFMA(z, 2, 0, 0) W43_MFMA(z, 2, 1, 0) W43_MFMA(z, 2, 2, 0)
            W43_MFMA(z, 2, 0, 1) W43_MFMA(z, 2, 1, 1) W43_MFMA(z, 2, 2, 1)
        } else {
            W43_PERIOD(x, y, true, true)  // leaves group 2 of chunk 0 in set y
            for (k = 1; k + 2 < n_chunks; k += 2) {
                W43_PERIOD(y, x, false, true)
                ++k;
                W43_PERIOD(x, y, false, true)
                --k;
            }
            W43_PERIOD(y, x, false, false)  // k = n_chunks - 1
            if constexpr (VOUT) return;  // every V slab is in HBM
            W43_MFMA(x, 2, 0, 0) W43_MFMA(x, 2, 1, 0) W43_MFMA(x, 2, 2, 0)
            W43_MFMA(x, 2, 0, 1) W43_MFMA(x, 2, 1, 1) W43_MFMA(x, 2, 2, 1)
        }
    }
    W43_STAMP(2)

    w43_epilogue<TTW, POOL, false>(smem, acc, tid, pg, nh, g, n0, vtr0, tc0, scale, shift, out, 1.0f, nullptr);
    W43_CLK_END()
    W43_STAMP(3)
}

// Winograd F(4x4,3x3) weight transform U = G g G^T (float64, rounded once), packed [cin/4][36][channel pair][cout][2].
// G (Lavin & Gray) = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1].
__global__ void k_pack_wino43(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin) return;
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double gk[3][3], t[6][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gk[a][b] = (double)w[((int64_t)co * Cin + ci) * 9 + a * 3 + b];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * gk[0][b] + G[a][1] * gk[1][b] + G[a][2] * gk[2][b];
    const int chunk = ci / W43_K, c4 = ci % W43_K;
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            const double u = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
            o[((((int64_t)chunk * 36 + a * 6 + b) * 2 + (c4 >> 1)) * Cout + co) * 2 + (c4 & 1)] = (float)u;
        }
}

#ifndef S43B_ABL
#define S43B_ABL 0  // timing experiment (k_conv_wino43s / s2): 1 = no main loop (prologue + epilogue(s) only), 2 = no slab copies in the loop, 4 = no operand reads / MFMAs; 0 in every build that ships
#endif
// ---- split-precision streaming convolution ----------------------------------------------------------------------------
// The same convolution as MODE 1 (transformed input and weights both streamed), on the f16 matrix pipe: every f32 operand x
// is carried as two f16 halves  hi = rn16(s x), lo = rn16(s x - hi)  (s a power of two: per layer for the weights, per
// stream for the transformed input, chosen from the data so that nothing overflows: w43s_vscale), 22 significand bits
// against float32's 24, and every product as  hi hi' + hi lo' + lo hi'  on v_mfma_f32_32x32x16_f16 with float32
// accumulation.  Measured against float64 on K = 2 048 .. 18 432 products of random and of heavy-tailed data
// (tools/ubench/split_mfma.hip) the result is as close as the exact-f32 MFMA's (rms 1.6e-7 / 4.4e-7 of the output maximum
// against 2.4e-7 / 6.9e-7: the f32 pipe rounds after every rank-2 update, this one after every rank-16), the dropped
// lo lo' term changes nothing in the last digit shown, and three f16 MFMAs do in 96 cycles what eight f32 MFMAs do in 512.
// With the matrix pipe out of the way the loop is bound by filling LDS (24 TB/s of L2 -> LDS over the chip = 41 B/clk/CU,
// same ubench), so it is built around the copies: a slab = 8 position slots x 16 input channels x (32 tiles + 64 couts) x
// (hi, lo) = 48 KB, three slabs in LDS; period k multiplies slab k (two blocks = 6 MFMAs per wave) while slab k + 1 is
// landing and slab k + 2 is issued.  The waves form two sets that issue on alternate periods: a wave waits vmcnt(0)
// only just before the barrier that precedes its next issue (the rule of this file: no partial vmcnt waits around LDS-DMA),
// a slab has two periods to land, and there is always one in flight.
// Wave (pg, nh) owns positions 9 pg .. 9 pg + 8 x channel half nh as in k_conv_wino43 (same epilogue); a 32-channel
// super-step walks its nine blocks twice (channels 0..15, then 16..31), two per period: 9 periods, slab 9 ss + e / 2 holds
// walk index e = 9 kg + q in slot 2 pg + e % 2.  The loop body is 18 periods (64 channels) so that ring buffer, issuing set
// and accumulator index are all compile-time.
template <int TTW, bool POOL>
__global__ __launch_bounds__(W43_THREADS) void k_conv_wino43s(const char *__restrict__ vsl, const char *__restrict__ usl,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               float *__restrict__ out, Wino43Geom g,
                                                               const unsigned *__restrict__ amax, const float *__restrict__ u_inv_p) {
    constexpr int TTH = 32 / TTW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup order: as MODE 1 (XCD b % 8 runs ct_group channel tiles x 32 / ct_group pixel blocks at a time)
    const int n_tiles = g.Cout / 64;
    const int b = (int)blockIdx.x + g.b0, xcd = b & 7, jb = b >> 3, r = jb & 31, gi = jb >> 5;
    const int a = g.ct_group, n_ctg = n_tiles / a;
    const int ct = (gi % n_ctg) * a + (r % a);
    const int m_blk = ((gi / n_ctg) * (32 / a) + r / a) * 8 + xcd;
    const int n0 = ct * 64;
    if (m_blk >= g.n_mblocks) return;
    W43_CLK_BEGIN()
    int cb;
    const int rb = fdiv(m_blk, g.fNCB, cb);
    const int vtr0 = rb * TTH, tc0 = cb * TTW;
    const int n_slabs = (g.Cin >> 5) * 9;
    const char *vbase = vsl + (int64_t)m_blk * n_slabs * S43_VPART;
    const char *ubase = usl + (int64_t)ct * n_slabs * S43_UPART;
    const int set = wv >> 2, w4 = wv & 3, pg = wv >> 1, nh = wv & 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const char *vw = vbase + w4 * 1024, *uw = ubase + w4 * 1024;
    const unsigned ldsw = lds0 + (unsigned)w4 * 1024u;
    const char *a_rd = (const char *)smem + pg * 2 * 2048 + lane * 16;
    const char *b_rd = (const char *)smem + S43_VPART + pg * 2 * 4096 + ((lane >> 5) * 64 + nh * 32 + (lane & 31)) * 16;

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;

// slab SL -> ring buffer BUF: 48 pieces of 1 KB (0..15 transformed input, 16..47 weights), 12 per wave of the issuing set
// slab SL -> ring buffer BUF: 48 pieces of 1 KB (0..15 transformed input, 16..47 weights), dealt round-robin to the four waves of
// the issuing set (piece = 4 c + w4), so that WHICH stream a copy reads is a compile-time property of c: with twelve consecutive
// pieces per wave it depended on the wave, and every copy carried ten scalar instructions of address selection in front of it --
// issued in order, in the way of the wave's own MFMAs.  vw / uw = the wave's first piece of either part.
#define S43_ISSUE(SL, BUF) S43_ISSUE_R(SL, BUF, 0, 12)
#define S43_MFMA(Q, A_, B_) acc[Q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc[Q], 0, 0, 0);
// S43_ILV (default): the period as a fixed interleave, one piece per MFMA gap (sched_barrier between gaps, every MFMA pinned:
// DESIGN.md 4.1(4)): at most two operand reads between two MFMAs (the rate at which tools/ubench/mfma_interleave.hip shows
// them free) instead of the whole period's reads at its top behind the barrier, the last block of a slab multiplied at the top
// of the NEXT period from registers (ahP ..) while that period's first reads are in flight, and the issuing set's copies in
// three groups between products instead of ahead of them.  Measured (conv_bench, 512 streams): the five two-sweep layers
// 13.07 -> 12.64 ms, profiles/README.md.
#ifndef S43_ILV
#define S43_ILV 1
#endif
#define S43_MFMA_P(Q, A_, B_) asm volatile("" : "+v"(acc[Q])); S43_MFMA(Q, A_, B_) asm volatile("" : "+v"(acc[Q]));
#define S43_GAP() __builtin_amdgcn_sched_barrier(0);
#define S43_ISSUE_R(SL, BUF, C0_, C1_)                                                                   \
    _Pragma("unroll") for (int c_ = (C0_); c_ < (C1_); ++c_) {                                            \
        const char *src_ = c_ < 4 ? vw + (int64_t)(SL) * S43_VPART + c_ * 4096                            \
                                  : uw + (int64_t)(SL) * S43_UPART + (c_ - 4) * 4096;                     \
        glds16_m0((const float *)src_, (unsigned)lane * 16u, ldsw + (unsigned)((BUF) * S43_SLAB + c_ * 4096)); \
    }
#define S43_PERIOD(K18, SL)                                                                              \
    {                                                                                                    \
        constexpr int BUF_ = (K18) % 3, NB_ = ((K18) + 2) % 3;                                            \
        constexpr int Q0_ = (2 * ((K18) % 9)) % 9, Q1_ = (2 * ((K18) % 9) + 1) % 9;                       \
        constexpr int QP_ = (2 * (((K18) + 8) % 9) + 1) % 9;   /* block 1 of the period before */         \
        const bool mine_ = set == ((K18) & 1);                                                            \
        const bool iss_ = !(S43B_ABL & 2) && mine_ && (SL) + 2 < n_slabs;                                 \
        const char *pa_ = a_rd + BUF_ * S43_SLAB, *pb_ = b_rd + BUF_ * S43_SLAB;                          \
        if (!S43_ILV) {                                                                                   \
            if (iss_) { S43_ISSUE((SL) + 2, NB_) }                                                        \
            const h8 ah0 = *(const h8 *)(pa_), al0 = *(const h8 *)(pa_ + 1024);                           \
            const h8 bh0 = *(const h8 *)(pb_), bl0 = *(const h8 *)(pb_ + 2048);                           \
            const h8 ah1 = *(const h8 *)(pa_ + 2048), al1 = *(const h8 *)(pa_ + 3072);                    \
            const h8 bh1 = *(const h8 *)(pb_ + 4096), bl1 = *(const h8 *)(pb_ + 6144);                    \
            if (!(S43B_ABL & 4)) {                                                                        \
            S43_MFMA(Q0_, al0, bh0) S43_MFMA(Q1_, al1, bh1)                                               \
            S43_MFMA(Q0_, ah0, bl0) S43_MFMA(Q1_, ah1, bl1)                                               \
            S43_MFMA(Q0_, ah0, bh0) S43_MFMA(Q1_, ah1, bh1)                                               \
            }                                                                                             \
        } else if (!(S43B_ABL & 4)) {                                                                     \
            h8 ah0, al0, bh0, bl0;                                                                        \
            const bool prev_ = (SL) > 0;                                                                  \
            S43_GAP()                                                                                     \
            if (prev_) { S43_MFMA_P(QP_, alP, bhP) } ah0 = *(const h8 *)(pa_); al0 = *(const h8 *)(pa_ + 1024); S43_GAP() \
            if (prev_) { S43_MFMA_P(QP_, ahP, blP) } bh0 = *(const h8 *)(pb_); bl0 = *(const h8 *)(pb_ + 2048); \
                if (iss_) { S43_ISSUE_R((SL) + 2, NB_, 0, 4) } S43_GAP()                                  \
            if (prev_) { S43_MFMA_P(QP_, ahP, bhP) } ahP = *(const h8 *)(pa_ + 2048); alP = *(const h8 *)(pa_ + 3072); S43_GAP() \
            S43_MFMA_P(Q0_, al0, bh0) bhP = *(const h8 *)(pb_ + 4096); blP = *(const h8 *)(pb_ + 6144);   \
                if (iss_) { S43_ISSUE_R((SL) + 2, NB_, 4, 8) } S43_GAP()                                  \
            S43_MFMA_P(Q0_, ah0, bl0) if (iss_) { S43_ISSUE_R((SL) + 2, NB_, 8, 12) } S43_GAP()           \
            S43_MFMA_P(Q0_, ah0, bh0) S43_GAP()                                                           \
        } else if (iss_) { S43_ISSUE((SL) + 2, NB_) }                                                     \
        if (!mine_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        W43_BARRIER()                                                                                     \
    }

    if (set == 0) { S43_ISSUE(0, 0) } else { S43_ISSUE(1, 1) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W43_BARRIER()
    h8 ahP, alP, bhP, blP;  // S43_ILV: operands of the previous period's second block
    for (int sl = 0; sl < ((S43B_ABL & 1) ? 0 : n_slabs); sl += 18) {  // Cin % 64 == 0
        S43_PERIOD(0, sl) S43_PERIOD(1, sl + 1) S43_PERIOD(2, sl + 2) S43_PERIOD(3, sl + 3) S43_PERIOD(4, sl + 4) S43_PERIOD(5, sl + 5)
        S43_PERIOD(6, sl + 6) S43_PERIOD(7, sl + 7) S43_PERIOD(8, sl + 8) S43_PERIOD(9, sl + 9) S43_PERIOD(10, sl + 10) S43_PERIOD(11, sl + 11)
        S43_PERIOD(12, sl + 12) S43_PERIOD(13, sl + 13) S43_PERIOD(14, sl + 14) S43_PERIOD(15, sl + 15) S43_PERIOD(16, sl + 16) S43_PERIOD(17, sl + 17)
    }
    if (S43_ILV && !(S43B_ABL & 5)) {  // the last period's second block (n_slabs % 18 == 0: accumulator 8)
        S43_MFMA_P(8, alP, bhP) S43_MFMA_P(8, ahP, blP) S43_MFMA_P(8, ahP, bhP)
    }
    w43_epilogue<TTW, POOL, true>(smem, acc, tid, pg, nh, g, n0, vtr0, tc0, scale, shift, out, u_inv_p[0], amax);
    W43_CLK_END()
}

// ---- the same on 64 x 64 workgroup tiles, in two sweeps over the positions ----------------------------------------------
// k_conv_wino43s is bound by what it copies into LDS (measured: L2 hits 41.5 B/clk/CU, L2 misses 11.3 B/clk/CU, and the two
// add up: tools/ubench/split_mfma.hip), 4 bytes per operand element for 21.3 MACs on its 32-tile x 64-channel tile.  The tile
// cannot grow while all 36 positions' accumulators have to sit in registers (36 x 32 x 64 floats = 288 of the CU's 512 KB).
// Here a workgroup owns 64 tiles (two consecutive 32-tile pixel blocks) x 64 channels and goes over the input channels
// twice: sweep 0 accumulates the position rows {1, 2, 0}, sweep 1 the rows {3, 4, 5} (18 x 64 x 64 floats each = the same
// registers).  Y = A^T M A is linear in M, so each sweep's epilogue contributes its rows' part of the 4 x 4 outputs: sweep
// 0 leaves it in a workgroup-private scratch area (f32, 256 KB, written and read back by the same threads), sweep 1 adds
// its own and finishes (BN, ReLU, pool).  Per MAC the copies shrink by a third (32 MACs per 4 bytes), the misses with them.
// Slab = one local row (6 positions) x 16 channels x (64 tiles | 64 couts) x (hi, lo) = 48 KB, ring of three, two wave sets
// as in k_conv_wino43s; wave (th, nh, pp) = (tile half, channel half, position parity) owns the columns j = pp + 2 t of every
// row: 3 blocks = 9 MFMAs per period, accumulator 3 * local row + t.  Six periods (32 channels) per loop trip.
//
// Round 6, for SMALL batches (the reference's CLI default is a population of 32 = 64 streams: conv_block6 is 2 pixel-block pairs x 32
// channel tiles = 64 workgroups of 768 periods, each streaming 18.9 MB of weights):
//   * the XCDs in two dimensions (g.xcd_m, as in k_conv_wino43s3): with the pair class = b % 8 those 64 workgroups sat on TWO of
//     the eight XCDs -- two L2s and two fabric ports pulled all 604 MB of weights (1.1 TB/s);
//   * SWSPLIT: the two sweeps of an item as TWO workgroups (second half of the grid = sweep 1).  The sweeps never shared anything
//     but the partial outputs, which sweep 0 leaves in the scratch area anyway: sweep 1's workgroup waits for its partner's flag
//     (agent-scope release / acquire) just before it reads them.  Same loads, same products, same additions in the same order:
//     IDENTICAL BITS (tested), twice the workgroups.  Sweep-0 workgroups have the lower indices, are dispatched first and wait for
//     nobody, so the wait cannot deadlock however many workgroups are resident.
template <int TTW, bool POOL, bool SWSPLIT>
__global__ __launch_bounds__(W43_THREADS) void k_conv_wino43s2(const char *__restrict__ vsl, const char *__restrict__ usl,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                float *__restrict__ out, Wino43Geom g,
                                                                const unsigned *__restrict__ amax, const float *__restrict__ u_inv_p,
                                                                f32x4 *__restrict__ partial, unsigned *__restrict__ sweep_flags) {
    constexpr int TTH = 32 / TTW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup order: as MODE 1, over pixel-block PAIRS (g.n_mblocks = number of pairs); XCD = (pair class xcd % xm, channel-tile
    // range xcd / xm), xm = 8: every XCD runs all channel tiles of its pairs
    const int xm = g.xcd_m, n_tiles = (g.Cout / 64) / (8 / xm);   // channel tiles of this XCD's range
    const int half_grid = SWSPLIT ? (int)(gridDim.x >> 1) : (int)gridDim.x;   // a multiple of 8: b % 8 is the XCD in both halves
    const int my_sweep = SWSPLIT ? (int)(blockIdx.x >= (unsigned)half_grid) : 0;
    const int b = (int)blockIdx.x - my_sweep * half_grid, xcd = b & 7, jb = b >> 3, r = jb & 31, gi = jb >> 5;
    const int a = g.ct_group, n_ctg = n_tiles / a;
    const int ct = (xcd / xm) * n_tiles + (gi % n_ctg) * a + (r % a);
    const int m_pair = ((gi / n_ctg) * (32 / a) + r / a) * xm + (xcd % xm);
    const int n0 = ct * 64;
    if (m_pair >= g.n_mblocks) return;
    W43_CLK_BEGIN()
    const int n_slabs = (g.Cin >> 4) * 3;
    const int set = wv >> 2, w4 = wv & 3, th = wv >> 2, nh = (wv >> 1) & 1, pp = wv & 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned ldsw = lds0 + (unsigned)w4 * 1024u;
    const char *a_rd = (const char *)smem + pp * 4096 + (half * 64 + th * 32 + l31) * 16;
    const char *b_rd = (const char *)smem + S43B_PART + pp * 4096 + (half * 64 + nh * 32 + l31) * 16;
    f32x4 *my_partial = partial + (int64_t)b * (2 * 16 * W43_THREADS) + tid;  // [tile half][r * 4 + c][thread]
    const float u_inv = u_inv_p[0];

    f32x16 acc[9];
#define S43B_ISSUE(SL, BUF) S43B_ISSUE_R(SL, BUF, 0, 12)
#define S43B_BLOCK(T_, Q_)                                                                               \
    {                                                                                                    \
        const h8 ah_ = *(const h8 *)(pa_ + (T_) * 8192), al_ = *(const h8 *)(pa_ + (T_) * 8192 + 2048);   \
        const h8 bh_ = *(const h8 *)(pb_ + (T_) * 8192), bl_ = *(const h8 *)(pb_ + (T_) * 8192 + 2048);   \
        S43_MFMA(Q_, al_, bh_) S43_MFMA(Q_, ah_, bl_) S43_MFMA(Q_, ah_, bh_)                              \
    }
// S43_ILV as in k_conv_wino43s: block 2 of a slab waits in registers for the next period
#define S43B_ISSUE_R(SL, BUF, C0_, C1_) /* piece = 4 c + w4 as in k_conv_wino43s: c < 6 input, else weights */ \
    _Pragma("unroll") for (int c_ = (C0_); c_ < (C1_); ++c_) {                                            \
        const char *src_ = c_ < 6 ? vw + (int64_t)(SL) * S43B_PART + c_ * 4096                            \
                                  : uw + (int64_t)(SL) * S43B_PART + (c_ - 6) * 4096;                     \
        glds16_m0((const float *)src_, (unsigned)lane * 16u, ldsw + (unsigned)((BUF) * S43B_SLAB + c_ * 4096)); \
    }
#define S43B_LDA(T_, H_, L_) H_ = *(const h8 *)(pa_ + (T_) * 8192); L_ = *(const h8 *)(pa_ + (T_) * 8192 + 2048);
#define S43B_LDB(T_, H_, L_) H_ = *(const h8 *)(pb_ + (T_) * 8192); L_ = *(const h8 *)(pb_ + (T_) * 8192 + 2048);
#define S43B_PERIOD(K6, SL)                                                                              \
    {                                                                                                    \
        constexpr int BUF_ = (K6) % 3, NB_ = ((K6) + 2) % 3, SUB_ = (K6) % 3, QP_ = 3 * (((K6) + 2) % 3) + 2; \
        const bool mine_ = set == ((K6) & 1);                                                             \
        const bool iss_ = !(S43B_ABL & 2) && mine_ && (SL) + 2 < n_slabs;                                 \
        const char *pa_ = a_rd + BUF_ * S43B_SLAB, *pb_ = b_rd + BUF_ * S43B_SLAB;                        \
        if (!S43_ILV) {                                                                                   \
            if (iss_) { S43B_ISSUE((SL) + 2, NB_) }                                                       \
            if (!(S43B_ABL & 4)) { S43B_BLOCK(0, 3 * SUB_) S43B_BLOCK(1, 3 * SUB_ + 1) S43B_BLOCK(2, 3 * SUB_ + 2) } \
        } else if (!(S43B_ABL & 4)) {                                                                     \
            h8 ah0, al0, bh0, bl0, ah1, al1, bh1, bl1;                                                    \
            const bool prev_ = (SL) > 0;                                                                  \
            S43_GAP()                                                                                     \
            if (prev_) { S43_MFMA_P(QP_, alP, bhP) } S43B_LDA(0, ah0, al0) S43_GAP()                       \
            if (prev_) { S43_MFMA_P(QP_, ahP, blP) } S43B_LDB(0, bh0, bl0) S43_GAP()                       \
            if (prev_) { S43_MFMA_P(QP_, ahP, bhP) } S43B_LDA(1, ah1, al1) if (iss_) { S43B_ISSUE_R((SL) + 2, NB_, 0, 4) } S43_GAP() \
            S43_MFMA_P(3 * SUB_, al0, bh0) S43B_LDB(1, bh1, bl1) S43_GAP()                                \
            S43_MFMA_P(3 * SUB_, ah0, bl0) S43B_LDA(2, ahP, alP) if (iss_) { S43B_ISSUE_R((SL) + 2, NB_, 4, 8) } S43_GAP() \
            S43_MFMA_P(3 * SUB_, ah0, bh0) S43B_LDB(2, bhP, blP) S43_GAP()                                \
            S43_MFMA_P(3 * SUB_ + 1, al1, bh1) if (iss_) { S43B_ISSUE_R((SL) + 2, NB_, 8, 12) } S43_GAP()  \
            S43_MFMA_P(3 * SUB_ + 1, ah1, bl1) S43_GAP()                                                  \
            S43_MFMA_P(3 * SUB_ + 1, ah1, bh1) S43_GAP()                                                  \
        } else if (iss_) { S43B_ISSUE((SL) + 2, NB_) }                                                    \
        if (!mine_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        W43_BARRIER()                                                                                     \
    }

    float *xch = smem;                      // [12 positions of the pass][32 tiles][W43_XT]
    constexpr int XP = 32 * W43_XT;
    const int e_quad = tid & 15, e_tile = tid >> 4;
    const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k8 = {8.f, 8.f};
#define A4(a, b) __builtin_shufflevector(pk_add(P2(a, 0), P2(b, 0)), pk_add(P2(a, 1), P2(b, 1)), 0, 1, 2, 3)
#define S4(a, b) __builtin_shufflevector(pk_sub(P2(a, 0), P2(b, 0)), pk_sub(P2(a, 1), P2(b, 1)), 0, 1, 2, 3)
#define F4(c2, a, b) __builtin_shufflevector(pk_fma(c2, P2(a, 0), P2(b, 0)), pk_fma(c2, P2(a, 1), P2(b, 1)), 0, 1, 2, 3) /* c a + b */
// exchange of the local rows R0 .. R0 + NR - 1 of tile half HB (written by the four waves with th == HB), then the column half
// Z[ii][c] = sum_j M[row ii][j] A[j][c] in the reader thread (tile, channel quad)
#define S43B_EXCHANGE(HB, R0, NR)                                                                        \
    W43_BARRIER()                                                                                         \
    if (th == (HB)) {                                                                                     \
        _Pragma("unroll") for (int rr_ = 0; rr_ < (NR); ++rr_)                                            \
            _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_) {                                            \
                float *xp = xch + (rr_ * 6 + pp + 2 * t_) * XP + nh * 32 + l31;                           \
                _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_)                                         \
                    xp[((e_ & 3) + 8 * (e_ >> 2) + 4 * half) * W43_XT] = acc[3 * ((R0) + rr_) + t_][e_];  \
            }                                                                                             \
    }                                                                                                     \
    W43_BARRIER()                                                                                         \
    _Pragma("unroll") for (int ii = 0; ii < (NR); ++ii) {                                                 \
        f32x4 m[6];                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) m[j] = *(const f32x4 *)(xch + (ii * 6 + j) * XP + e_tile * W43_XT + e_quad * 4); \
        const f32x4 s12 = A4(m[1], m[2]), d12 = S4(m[1], m[2]), s34 = A4(m[3], m[4]), d34 = S4(m[3], m[4]); \
        Z[ii][0] = A4(A4(m[0], s12), s34);                                                                \
        Z[ii][1] = F4(k2, d34, d12);                                                                      \
        Z[ii][2] = F4(k4, s34, s12);                                                                      \
        Z[ii][3] = A4(F4(k8, d34, d12), m[5]);                                                            \
    }

#pragma unroll
    for (int sweep = 0; sweep < 2; ++sweep) {
        if (SWSPLIT && sweep != my_sweep) continue;   // (uniform) this workgroup runs one of the two
        const char *vw = vsl + ((int64_t)m_pair * 2 + sweep) * n_slabs * S43B_PART + w4 * 1024;
        const char *uw = usl + ((int64_t)ct * 2 + sweep) * n_slabs * S43B_PART + w4 * 1024;
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
        W43_BARRIER()  // (sweep 1: the epilogue's exchange reads are done before the ring is refilled)
        if (set == 0) { S43B_ISSUE(0, 0) } else { S43B_ISSUE(1, 1) }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W43_BARRIER()
        h8 ahP, alP, bhP, blP;  // S43_ILV: operands of the previous slab's block 2
        for (int sl = 0; sl < ((S43B_ABL & 1) ? 0 : n_slabs); sl += 6) {  // Cin % 32 == 0
            S43B_PERIOD(0, sl) S43B_PERIOD(1, sl + 1) S43B_PERIOD(2, sl + 2)
            S43B_PERIOD(3, sl + 3) S43B_PERIOD(4, sl + 4) S43B_PERIOD(5, sl + 5)
        }
        if (S43_ILV && !(S43B_ABL & 5)) {  // the last slab's block 2 (n_slabs % 6 == 0: local row 2 -> accumulator 8)
            S43_MFMA_P(8, alP, bhP) S43_MFMA_P(8, ahP, blP) S43_MFMA_P(8, ahP, bhP)
        }
        if (SWSPLIT && sweep == 1) {   // the partner's partial outputs have to be there (and visible) before they are read
            if (tid == 0)
                while (__hip_atomic_load(sweep_flags + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
            W43_BARRIER()
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        // ---- this sweep's rows of Y = A^T M A, one tile half (= one 32-tile pixel block) at a time --------------------
#pragma unroll 1
        for (int hb = 0; hb < 2; ++hb) {
            f32x4 Yo[4][4], Z[2][4];
            f32x4 *pt = my_partial + hb * (16 * W43_THREADS);
            if (sweep == 1) {  // sweep 0's contribution (issued ahead of the exchange that hides its latency)
#pragma unroll
                for (int e = 0; e < 16; ++e) Yo[e >> 2][e & 3] = pt[e * W43_THREADS];
            }
            S43B_EXCHANGE(hb, 0, 2)   // local rows 0, 1: position rows (1, 2) resp. (3, 4)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 sm = A4(Z[0][c], Z[1][c]), df = S4(Z[0][c], Z[1][c]);
                if (sweep == 0) {  // A^T columns (1,1,1,1) and (1,-1,1,-1)
                    Yo[0][c] = sm; Yo[1][c] = df; Yo[2][c] = sm; Yo[3][c] = df;
                } else {           // (1,2,4,8) and (1,-2,4,-8)
                    Yo[0][c] = A4(Yo[0][c], sm); Yo[1][c] = F4(k2, df, Yo[1][c]); Yo[2][c] = F4(k4, sm, Yo[2][c]); Yo[3][c] = F4(k8, df, Yo[3][c]);
                }
            }
            S43B_EXCHANGE(hb, 2, 1)   // local row 2: position row 0 resp. 5: A^T column (1,0,0,0) resp. (0,0,0,1)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (sweep == 0) Yo[0][c] = A4(Yo[0][c], Z[0][c]);
                else Yo[3][c] = A4(Yo[3][c], Z[0][c]);
            }
            if (sweep == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) pt[e * W43_THREADS] = Yo[e >> 2][e & 3];
            } else {
                // BN + ReLU (+ 2x2 average pool), 16-byte stores (4 channels) into NC8HW8; pixel block 2 m_pair + hb
                const int m_blk = 2 * m_pair + hb;
                int cb;
                const int rb = fdiv(m_blk, g.fNCB, cb);
                const int vtr = rb * TTH + e_tile / TTW, tc = cb * TTW + e_tile % TTW;
                const int co = n0 + e_quad * 4;
                f32x4 sc = *(const f32x4 *)(scale + co);
                const f32x4 sh = *(const f32x4 *)(shift + co);
                unsigned smax = 0;
                int sidx = 0;
                if (vtr < g.VTR && tc < g.TC) {
                    int tr;
                    const int s_ = fdiv(vtr, g.fTR, tr);
                    sidx = s_;
                    sc = sc * (u_inv / w43s_vscale(amax[s_]));
#pragma unroll
                    for (int r_ = 0; r_ < 4; ++r_)
#pragma unroll
                        for (int c = 0; c < 4; ++c) Yo[r_][c] = __builtin_elementwise_max(Yo[r_][c] * sc + sh, (f32x4)(0.0f));
                    if (POOL) {
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                            for (int pc = 0; pc < 2; ++pc) {
                                const int oh = 2 * tr + pr, ow = 2 * tc + pc;
                                if (oh < g.Ho && ow < g.Wo) {
                                    const f32x4 v = (((Yo[2 * pr][2 * pc] + Yo[2 * pr][2 * pc + 1]) + Yo[2 * pr + 1][2 * pc]) + Yo[2 * pr + 1][2 * pc + 1]) * 0.25f;
                                    *(f32x4 *)(out + act_off(s_, co, oh, ow, g.Cout, g.Ho, g.Wo)) = v;
                                    smax = max(smax, w43_max4(v));
                                }
                            }
                    } else {
#pragma unroll
                        for (int r_ = 0; r_ < 4; ++r_)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int hh = 4 * tr + r_, ww = 4 * tc + c;
                                if (hh < g.H && ww < g.W) {
                                    *(f32x4 *)(out + act_off(s_, co, hh, ww, g.Cout, g.H, g.W)) = Yo[r_][c];
                                    smax = max(smax, w43_max4(Yo[r_][c]));
                                }
                            }
                    }
                }
                if (g.amax_out != nullptr) w43_amax_out(g.amax_out, sidx, smax);
            }
        }
        if (SWSPLIT && sweep == 0) {   // publish: every thread's partial stores, then the flag
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            W43_BARRIER()
            if (tid == 0) __hip_atomic_store(sweep_flags + b, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    W43_CLK_END()
#undef A4
#undef S4
#undef F4
}

// ---- the same on 128 x 128 workgroup tiles, in six sweeps of one position row each ----------------------------------------
// Both streaming kernels above are bound by what they copy from L2 into LDS (profiles/README.md: slab copies alone 1 625 cycles
// of a 1 970-cycle period; operand reads and products add little), i.e. by bytes per MAC = (tiles + couts) / (tiles x couts) of
// the workgroup tile: 96 / 2 048 (k_conv_wino43s), 128 / 4 096 (k_conv_wino43s2).  Here 256 / 16 384: half of the two-sweep
// kernel's -- measured on the loop alone before anything else was written (tools/ubench/s3_loop.hip, profiles/
// round4_s3_loop_ubench.txt): 1 850 cycles per 48 KB slab of 786 K MACs against 1 970 per 393 K, 0.56 of the f16 peak issued.
// What pays for it: only ONE position row's accumulators fit the registers (6 positions x 128 x 128 floats = 384 KB; wave
// (tile half, cout quarter) owns 64 tiles x 32 couts = 12 blocks = 192 VGPRs), so the input channels are swept six times, once
// per position row i, and Y = A^T M A -- linear in M, row i contributing A[i][r] (sum_j M[i][j] A[j][c]) to output (r, c) --
// is assembled through a workgroup-private f32 scratch area (1.25 MB, written and read back by the same lanes): a wave owns
// all six positions of the row for its sub-tile, so the column combination Z_i[c] = sum_j M[i][j] A[j][c] happens in registers
// -- NO exchange through LDS, no barrier in the epilogue -- and rows 0..4 just store their four Z values per output (stores
// only: nothing waits; accumulating the 16 outputs instead was 3.4 x the traffic and a load-add-store chain per sweep: measured,
// the first version); the sweep of row 5 reads the twenty values back, one output column at a time, combines the rows, and
// finishes (operand scales, BN, ReLU, pool, stores).
// Slab = 3 positions (half a row) x 16 channels x (128 tiles | 128 couts) x (hi, lo) = 48 KB: [input part 24 KB | weight part
// 24 KB], position jj at 8 KB: [hi 4 KB | lo 4 KB], each [channel octet][128][8 f16]; ring of three, two wave sets issuing on
// alternate periods as above; two periods per 16 channels.  MFMA A operand = weights (rows = couts), B = input (columns =
// tiles): a lane's accumulator quad = 4 consecutive couts of ITS tile, 16-byte partial-sum accesses and stores fall out.
// The arithmetic and its order are k_conv_wino43s2's -- per position the same sequence of products, Y = A^T M A with the same
// association -- so the two kernels return identical bits (tested): stito_cnn14_forward may pick either by batch size.
#ifndef S43C_ABL
#define S43C_ABL 0  // timing experiment (k_conv_wino43s3): 1 = no epilogues, 2 = no main loops, 4 = no partial stores, 8 = no partial loads; 0 in every build that ships
#endif
__device__ __forceinline__ const char *w43_uniform(const char *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const char *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
// SWSPLIT (round 6, small batches): the six sweeps of an item as SIX workgroups (sixth s of the grid = position row s): rows 0 .. 4 leave
// their Z in the scratch area as before and count themselves in (agent-scope release), the row-5 workgroup waits for the count of five
// before it reads them back.  Same products, same additions, same order: identical bits; six times the workgroups for a batch that
// would leave most CUs idle.  Rows 0 .. 4 have the lower workgroup indices and wait for nobody: no deadlock whatever is resident.
template <int TTW, bool POOL, bool SWSPLIT>
__global__ __launch_bounds__(W43_THREADS) void k_conv_wino43s3(const char *__restrict__ vsl, const char *__restrict__ usl,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                float *__restrict__ out, Wino43Geom g,
                                                                const unsigned *__restrict__ amax, const float *__restrict__ u_inv_p,
                                                                f32x4 *__restrict__ partial, unsigned *__restrict__ sweep_count) {
    constexpr int TTH = 32 / TTW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, oct = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup order: as the other streaming kernels, over pixel-block QUADS (g.n_mblocks = number of quads) and 128-cout tiles
    // ... with the XCDs (b % 8, each with its own L2) split in TWO dimensions when there are few quads: XCD = (quad class xcd % xm,
    // channel-tile range xcd / xm).  An XCD's L2 fetches the V slabs of its quads and the U slabs of its channel tiles -- the same
    // number of bytes per quad as per tile -- so quads / xm + tiles / (8 / xm) is what the eight L2s pull over the fabric each
    // (profiles/round5_conv_ea_pmc.txt: with every XCD running all 16 channel tiles of conv_block6.conv2, 8 x 604 MB of weights)
    const int xm = g.xcd_m, n_tiles = (g.Cout / 128) / (8 / xm);   // channel tiles of this XCD's range
    const int sixth = SWSPLIT ? (int)(gridDim.x / 6) : (int)gridDim.x;   // a multiple of 8: b % 8 is the XCD in every sixth
    const int my_sweep = SWSPLIT ? (int)(blockIdx.x / (unsigned)sixth) : 0;
    const int b = (int)blockIdx.x - my_sweep * sixth, xcd = b & 7, jb = b >> 3, r = jb & 31, gi = jb >> 5;
    const int a = g.ct_group, n_ctg = n_tiles / a;
    const int ct = (xcd / xm) * n_tiles + (gi % n_ctg) * a + (r % a);
    const int m_quad = ((gi / n_ctg) * (32 / a) + r / a) * xm + (xcd % xm);
    if (m_quad >= g.n_mblocks) return;
    W43_CLK_BEGIN()
    const int nP = g.Cin >> 3;   // periods per sweep: two per 16 input channels
    const int sw_lo = SWSPLIT ? my_sweep : 0, sw_hi = SWSPLIT ? my_sweep + 1 : 6;   // the sweeps this workgroup runs
    const int sl_end = sw_hi * nP;                                                  // ... and the end of its slab stream
    const int set = wv >> 2, w4 = wv & 3, th = wv & 1, cq = wv >> 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const unsigned ldsw = lds0 + (unsigned)w4 * 1024u;
    const char *v_rd = (const char *)smem + (oct * 128 + th * 64 + l31) * 16;               // + jj * 8192 (+ 4096: lo) + blk * 512
    const char *u_rd = (const char *)smem + S43B_PART + (oct * 128 + cq * 32 + l31) * 16;   // + jj * 8192 (+ 4096: lo)
    f32x4 *my_partial = partial + (int64_t)b * (8 * 2 * 4 * 20 * 64) + wv * (2 * 4 * 20 * 64) + lane;  // [wave][blk][quad][row 0..4][c][lane]
    const float u_inv = u_inv_p[0];

    f32x16 acc[12];   // [position j of the row][tile panel]
// piece = 4 c + w4 (c < 6: input part) as in the kernels above
#define S43C_ISSUE(SL, BUF)                                                                              \
    _Pragma("unroll") for (int c_ = 0; c_ < 12; ++c_) {                                                   \
        const char *src_ = c_ < 6 ? vw + (int64_t)(SL) * S43B_PART + c_ * 4096                            \
                                  : uw + (int64_t)(SL) * S43B_PART + (c_ - 6) * 4096;                     \
        glds16_m0((const float *)src_, (unsigned)lane * 16u, ldsw + (unsigned)((BUF) * S43B_SLAB + c_ * 4096)); \
    }
#define S43C_PERIOD(KK, SL, BUF_)                                                                        \
    {                                                                                                    \
        const bool mine_ = set == (KK);                                                                   \
        if (mine_ && (SL) + 2 < sl_end) { const int nb_ = (BUF_) + 2 >= 3 ? (BUF_) - 1 : (BUF_) + 2; S43C_ISSUE((SL) + 2, nb_) } \
        const char *pv_ = v_rd + (BUF_) * S43B_SLAB, *pu_ = u_rd + (BUF_) * S43B_SLAB;                    \
        _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                                \
            const h8 uh_ = *(const h8 *)(pu_ + q_ * 8192), ul_ = *(const h8 *)(pu_ + q_ * 8192 + 4096);    \
            const h8 vh0_ = *(const h8 *)(pv_ + q_ * 8192), vl0_ = *(const h8 *)(pv_ + q_ * 8192 + 4096);  \
            const h8 vh1_ = *(const h8 *)(pv_ + q_ * 8192 + 512), vl1_ = *(const h8 *)(pv_ + q_ * 8192 + 4096 + 512); \
            constexpr int a0_ = ((KK) * 3) * 2;                                                           \
            S43_MFMA(a0_ + 2 * q_, uh_, vl0_) S43_MFMA(a0_ + 2 * q_ + 1, uh_, vl1_)   /* input lo x weight hi first, as */ \
            S43_MFMA(a0_ + 2 * q_, ul_, vh0_) S43_MFMA(a0_ + 2 * q_ + 1, ul_, vh1_)   /* k_conv_wino43s2: the same bits */ \
            S43_MFMA(a0_ + 2 * q_, uh_, vh0_) S43_MFMA(a0_ + 2 * q_ + 1, uh_, vh1_)                       \
        }                                                                                                 \
        if (!mine_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        W43_BARRIER()                                                                                     \
    }

    // this lane's two tiles (one per tile panel): map position, validity, stream, BN scale with the operand scales folded in
    int t_s[2], t_tr[2], t_tc[2];
    bool t_ok[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int m_blk = 4 * m_quad + 2 * th + blk;
        int cb;
        const int rb = fdiv(m_blk, g.fNCB, cb);
        const int vtr = rb * TTH + l31 / TTW;
        t_tc[blk] = cb * TTW + l31 % TTW;
        t_ok[blk] = vtr < g.VTR && t_tc[blk] < g.TC;
        int tr = 0;
        t_s[blk] = t_ok[blk] ? fdiv(vtr, g.fTR, tr) : 0;
        t_tr[blk] = tr;
    }

    // (the 64-bit products are VALU work: back into scalar registers by hand -- hipcc hands an "s" asm operand a VGPR pair otherwise)
    const char *vw = w43_uniform(vsl + (int64_t)m_quad * 6 * nP * S43B_PART + w4 * 1024);
    const char *uw = w43_uniform(usl + (int64_t)ct * 6 * nP * S43B_PART + w4 * 1024);
    if (set == 0) { S43C_ISSUE(sw_lo * nP, 0) } else { S43C_ISSUE(sw_lo * nP + 1, 1) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W43_BARRIER()
    int buf = 0;
#pragma unroll 1
    for (int sweep = sw_lo; sweep < sw_hi; ++sweep) {
#pragma unroll
        for (int q = 0; q < 12; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
        // the six rows' slabs follow one another in both streams: ONE stream of 6 nP periods, the ring keeps filling through the
        // epilogues (which touch neither LDS nor barriers); nP % 2 == 0; the ring position is a run-time value (three loop bodies
        // for it cost registers)
        for (int sl = sweep * nP; sl < ((S43C_ABL & 2) ? 0 : (sweep + 1) * nP); sl += 2) {
            S43C_PERIOD(0, sl, buf)
            buf = buf == 2 ? 0 : buf + 1;
            S43C_PERIOD(1, sl + 1, buf)
            buf = buf == 2 ? 0 : buf + 1;
        }
        // ---- this row's column combination Z[c] = sum_j M[row][j] A[j][c] (c = 0..3) goes to the scratch area (rows 0..4: stores
        // only, nothing waits for them); the sweep of row 5 reads the five others back, one output column at a time, and finishes:
        // Y[r][c] = sum_i A[i][r] Z_i[c],  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1] ------------------------------
        if (S43C_ABL & 1) {   // (timing builds: keep the accumulators alive)
            float s_ = 0.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) s_ += acc[q][0] + acc[q][7] + acc[q][15];
            if (s_ == 12345.f) out[tid] = s_;
            continue;
        }
#define S43C_Z(BLK, GQ, Z_)   /* element by element straight from the accumulator registers (vector temporaries of them are copies) */ \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
            const float m0 = acc[0 + (BLK)][4 * (GQ) + e], m1 = acc[2 + (BLK)][4 * (GQ) + e], m2 = acc[4 + (BLK)][4 * (GQ) + e], \
                        m3 = acc[6 + (BLK)][4 * (GQ) + e], m4 = acc[8 + (BLK)][4 * (GQ) + e], m5 = acc[10 + (BLK)][4 * (GQ) + e]; \
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;                       \
            Z_[0][e] = (m0 + s12) + s34;                                                                  \
            Z_[1][e] = 2.f * d34 + d12;                                                                   \
            Z_[2][e] = 4.f * s34 + s12;                                                                   \
            Z_[3][e] = (8.f * d34 + d12) + m5;                                                            \
        }
        if (sweep < 5) {
#pragma unroll
            for (int bg = 0; bg < 8; ++bg) {
                W43_FENCE()
                f32x4 Z[4];
                S43C_Z(bg >> 2, bg & 3, Z)
                f32x4 *pt = my_partial + bg * (5 * 4 * 64);   // [row 0..4][c][lane]
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (S43C_ABL & 4) { if (Z[c][0] == 12345.f) pt[(sweep * 4 + c) * 64] = Z[c]; }   // timing builds: no partial stores
                    else pt[(sweep * 4 + c) * 64] = Z[c];
                }
            }
            if (SWSPLIT) {   // publish this row: every thread's stores, then the count
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                W43_BARRIER()
                if (tid == 0) __hip_atomic_fetch_add(sweep_count + b, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (SWSPLIT) {   // the other five rows have to be there (and visible) before they are read back
                if (tid == 0)
                    while (__hip_atomic_load(sweep_count + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 5u) __builtin_amdgcn_s_sleep(8);
                W43_BARRIER()
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            // the last row: all eight Z sets first (the accumulators are dead behind them: 128 registers for the loads that follow --
            // a (panel, quad)'s twenty stored values in flight at once; five at a time was 32 dependent round trips to HBM per workgroup)
            f32x4 Zl[8][4];
#pragma unroll
            for (int bg = 0; bg < 8; ++bg) { S43C_Z(bg >> 2, bg & 3, Zl[bg]) }
#pragma unroll
            for (int bg = 0; bg < 8; ++bg) {
                W43_FENCE()
                const int blk = bg >> 2, gq = bg & 3;
                const f32x4 *pt = my_partial + bg * (5 * 4 * 64);
                f32x4 zr[5][4];
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) zr[i][c] = (S43C_ABL & 8) ? Zl[bg][c] : pt[(i * 4 + c) * 64];   // (timing builds: no partial loads)
                const int co = ct * 128 + cq * 32 + 8 * gq + 4 * oct;
                f32x4 sc = *(const f32x4 *)(scale + co);
                const f32x4 sh = *(const f32x4 *)(shift + co);
                unsigned smax = 0;
                const int s_ = t_s[blk], tr = t_tr[blk], tc = t_tc[blk];
                if (t_ok[blk]) sc = sc * (u_inv / w43s_vscale(amax[s_]));
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    f32x4 y[2][4];   // [column 2 pc + cc][output row]
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const int c = 2 * pc + cc;
                        const f32x4 p12 = zr[1][c] + zr[2][c], q12 = zr[1][c] - zr[2][c], p34 = zr[3][c] + zr[4][c], q34 = zr[3][c] - zr[4][c];
                        y[cc][0] = (zr[0][c] + p12) + p34;
                        y[cc][1] = 2.f * q34 + q12;
                        y[cc][2] = 4.f * p34 + p12;
                        y[cc][3] = (8.f * q34 + q12) + Zl[bg][c];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) y[cc][rr] = __builtin_elementwise_max(y[cc][rr] * sc + sh, (f32x4)(0.0f));
                    }
                    if (t_ok[blk]) {
                        if (POOL) {
#pragma unroll
                            for (int pr = 0; pr < 2; ++pr) {
                                const int oh = 2 * tr + pr, ow = 2 * tc + pc;
                                if (oh < g.Ho && ow < g.Wo) {
                                    const f32x4 v = (((y[0][2 * pr] + y[1][2 * pr]) + y[0][2 * pr + 1]) + y[1][2 * pr + 1]) * 0.25f;
                                    *(f32x4 *)(out + act_off(s_, co, oh, ow, g.Cout, g.Ho, g.Wo)) = v;
                                    smax = max(smax, w43_max4(v));
                                }
                            }
                        } else {
#pragma unroll
                            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                                for (int rr = 0; rr < 4; ++rr) {
                                    const int hh = 4 * tr + rr, ww = 4 * tc + 2 * pc + cc;
                                    if (hh < g.H && ww < g.W) {
                                        *(f32x4 *)(out + act_off(s_, co, hh, ww, g.Cout, g.H, g.W)) = y[cc][rr];
                                        smax = max(smax, w43_max4(y[cc][rr]));
                                    }
                                }
                        }
                    }
                }
                // amax_out[s] only grows: a (possibly stale) agent-scope load that already covers smax makes the atomic unnecessary
                if (g.amax_out != nullptr && t_ok[blk] && smax > __hip_atomic_load(g.amax_out + s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    atomicMax(g.amax_out + s_, smax);
            }
        }
    }
    W43_CLK_END()
}

// weights of k_conv_wino43s3: [cout / 128][position row i][period 2 (c / 16) + (j >= 3)] x 24 KB, position j % 3 at 8 KB:
// [hi 4 KB | lo 4 KB], each [(c % 16) / 8][cout % 128][c % 8]
__global__ void k_pack_wino43s3(const float *__restrict__ w, int Cout, int Cin, char *__restrict__ o, const unsigned *__restrict__ hdr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin) return;
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double gk[3][3], t[6][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gk[a][b] = (double)w[((int64_t)co * Cin + ci) * 9 + a * 3 + b];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * gk[0][b] + G[a][1] * gk[1][b] + G[a][2] * gk[2][b];
    const int nP = Cin >> 3;
    const int kg = ci >> 4, h = (ci >> 3) & 1, i8 = ci & 7;
    const float su = __uint_as_float(hdr[2]);
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            const float us = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]) * su;
            const _Float16 hi = (_Float16)us, lo = (_Float16)(us - (float)hi);
            char *d = o + (((int64_t)(co >> 7) * 6 + a) * nP + 2 * kg + (b >= 3)) * S43B_PART + (b % 3) * 8192 + (h * 128 + (co & 127)) * 16 + i8 * 2;
            *(_Float16 *)d = hi;
            *(_Float16 *)(d + 4096) = lo;
        }
}

// weights of k_conv_wino43s2: as k_pack_wino43s<1> in the slab order [cout / 64][sweep][slab = 3 (c / 16) + local row][column j]
__global__ void k_pack_wino43s2(const float *__restrict__ w, int Cout, int Cin, char *__restrict__ o, const unsigned *__restrict__ hdr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin) return;
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double gk[3][3], t[6][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gk[a][b] = (double)w[((int64_t)co * Cin + ci) * 9 + a * 3 + b];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * gk[0][b] + G[a][1] * gk[1][b] + G[a][2] * gk[2][b];
    const int n_slabs = (Cin >> 4) * 3;
    const int kg = ci >> 4, h = (ci >> 3) & 1, i8 = ci & 7;
    const float su = __uint_as_float(hdr[2]);
    for (int a = 0; a < 6; ++a) {
        const int sw = a >= 3, lr = sw ? a - 3 : (a == 0 ? 2 : a - 1);
        for (int b = 0; b < 6; ++b) {
            const float us = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]) * su;
            const _Float16 hi = (_Float16)us, lo = (_Float16)(us - (float)hi);
            char *d = o + (((int64_t)(co >> 6) * 2 + sw) * n_slabs + 3 * kg + lr) * S43B_PART + b * 4096 + (h * 64 + (co & 63)) * 16 + i8 * 2;
            *(_Float16 *)d = hi;
            *(_Float16 *)(d + 2048) = lo;
        }
    }
}

// Largest activation of every stream (>= 0 after ReLU; the sign bit is dropped anyway) as a bit pattern: amax[s] is zeroed
// by the launcher, one atomicMax per wave.  grid (splits, S).
__global__ __launch_bounds__(256) void k_stream_absmax(const float *__restrict__ x, int64_t per_stream, unsigned *__restrict__ amax) {
    const f32x4 *xs = (const f32x4 *)(x + (int64_t)blockIdx.y * per_stream);
    const int64_t n4 = per_stream >> 2;  // per_stream % 8 == 0 (channel-blocked layout)
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = xs[i];
        const unsigned a0 = __float_as_uint(v[0]) & 0x7fffffffu, a1 = __float_as_uint(v[1]) & 0x7fffffffu;
        const unsigned a2 = __float_as_uint(v[2]) & 0x7fffffffu, a3 = __float_as_uint(v[3]) & 0x7fffffffu;
        m = max(max(m, max(a0, a1)), max(a2, a3));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax + blockIdx.y, m);
}

// Weights of the split-precision kernel: U = G g G^T as in k_pack_wino43 (float64, rounded once to float32), then scaled by
// the layer's power of two and split.  PASS 0: max |U| (bit pattern) into hdr[0]; PASS 1: the slabs,
// [cout / 64][slab][slot][hi | lo][channel octet][64 couts][8 f16].  hdr = {max bits, 1 / scale, scale} behind the slabs.
template <int PASS>
__global__ void k_pack_wino43s(const float *__restrict__ w, int Cout, int Cin, char *__restrict__ o, unsigned *__restrict__ hdr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin) return;
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double gk[3][3], t[6][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gk[a][b] = (double)w[((int64_t)co * Cin + ci) * 9 + a * 3 + b];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * gk[0][b] + G[a][1] * gk[1][b] + G[a][2] * gk[2][b];
    const int n_slabs = (Cin >> 5) * 9;
    const int ss = ci >> 5, kg = (ci >> 4) & 1, h = (ci >> 3) & 1, i8 = ci & 7;
    const float su = PASS == 1 ? __uint_as_float(hdr[2]) : 1.0f;
    unsigned mx = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            const float u = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
            if (PASS == 0) {
                mx = max(mx, __float_as_uint(u) & 0x7fffffffu);
            } else {
                const int p = a * 6 + b, pg = p / 9, e = kg * 9 + p - 9 * pg;
                const float us = u * su;
                const _Float16 hi = (_Float16)us, lo = (_Float16)(us - (float)hi);
                char *d = o + ((int64_t)(co >> 6) * n_slabs + 9 * ss + (e >> 1)) * S43_UPART + (pg * 2 + (e & 1)) * 4096 +
                          (h * 64 + (co & 63)) * 16 + i8 * 2;
                *(_Float16 *)d = hi;
                *(_Float16 *)(d + 2048) = lo;
            }
        }
    if (PASS == 0 && mx) atomicMax(hdr, mx);
}

__global__ void k_pack_wino43s_scale(unsigned *hdr) {
    // max |U| < 2^e  ->  scale 2^(14 - e): every scaled weight below 2^14
    const unsigned mb = hdr[0];
    int e = (int)((mb >> 23) & 0xff) - 126;
    if (mb == 0u) e = 14;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    hdr[1] = __float_as_uint(__builtin_ldexpf(1.0f, e - 14));
    hdr[2] = __float_as_uint(__builtin_ldexpf(1.0f, 14 - e));
}

size_t wino43_split_packed_floats(int cout, int cin) { return (size_t)36 * cout * cin + 64; }

int pack_wino43_split(const float *w_oihw, int cout, int cin, float *packed, int layout, hipStream_t st) {
    STITO_REQUIRE(cin % 64 == 0 && cout % (layout == 2 ? 128 : 64) == 0, STITO_E_UNSUPPORTED, "conv (split-precision winograd): cin %d / cout %d", cin, cout);
    const int64_t n = (int64_t)cout * cin;
    unsigned *hdr = (unsigned *)(packed + (size_t)36 * cout * cin);
    STITO_TRY(zero_async(hdr, 64 * sizeof(float), st));
    hipLaunchKernelGGL(k_pack_wino43s<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, (char *)packed, hdr);
    STITO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_wino43s_scale, dim3(1), dim3(1), 0, st, hdr);
    STITO_LAUNCH_CHECK();
    if (layout == 2) hipLaunchKernelGGL(k_pack_wino43s3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, (char *)packed, (const unsigned *)hdr);
    else if (layout == 1) hipLaunchKernelGGL(k_pack_wino43s2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, (char *)packed, (const unsigned *)hdr);
    else hipLaunchKernelGGL(k_pack_wino43s<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, (char *)packed, hdr);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

static int w43_ttw(const ConvShape &c, bool pool) {
    const int tc = pool ? (c.W / 2 + 1) / 2 : (c.W + 3) / 4;
    return tc >= 8 ? 8 : (tc >= 4 ? 4 : (tc >= 2 ? 2 : 1));
}

template <int TTW>
static bool w43_geometry(const ConvShape &c, bool pool, Wino43Geom &g, size_t &lds, int64_t &blocks) {
    g.b0 = 0;

    constexpr int TTH = 32 / TTW;
    g = Wino43Geom{};
    g.S = c.S; g.H = c.H; g.W = c.W; g.Cin = c.Cin; g.Cout = c.Cout;
    g.n_cgroups = 1;
    g.n_mblocks = 0;
    g.ct_group = 4;
    g.Ho = pool ? c.H / 2 : c.H;
    g.Wo = pool ? c.W / 2 : c.W;
    g.TR = pool ? (g.Ho + 1) / 2 : (c.H + 3) / 4;
    g.TC = pool ? (g.Wo + 1) / 2 : (c.W + 3) / 4;
    if (g.TR < 1 || g.TC < 1) return false;
    g.VTR = (int64_t)g.S * g.TR;
    g.n_col_blocks = (g.TC + TTW - 1) / TTW;
    g.fH = make_fdiv(g.H); g.fTR = make_fdiv(g.TR); g.fNCB = make_fdiv(g.n_col_blocks); g.fNT = make_fdiv(g.Cout / 64);
    const int64_t nrb = (g.VTR + TTH - 1) / TTH;
    blocks = nrb * g.n_col_blocks * (g.Cout / 64);
    if (nrb * g.n_col_blocks >= (1 << 30) || blocks >= (1ll << 31)) return false;
    if ((int64_t)g.S * g.H >= (1ll << 31) - 64 || g.VTR >= (1ll << 31) - 64) return false;  // 32-bit row arithmetic in the kernel
    // patch rows: 4 per tile row + 2 halo, plus the input rows skipped at every stream boundary a block can straddle
    const int skip = g.H - 4 * g.TR > 0 ? g.H - 4 * g.TR : 0;
    g.PR = 4 * TTH + 2 + ((TTH - 1) / g.TR + 1) * skip;
    if ((g.PR + 3) / 4 > W43Patch<TTW>::MAXPR4) return false;  // patch rows the permuted LDS layout has planes for
    // 32-bit byte offsets of the patch copies relative to the block's first stream
    if ((int64_t)((TTH - 1) / g.TR + 2) * (g.Cin / 8) * g.H * g.W * 32 >= (1ll << 32)) return false;
    lds = ((size_t)2 * W43_BUF + 2 * W43Patch<TTW>::PFL) * sizeof(float);
    return lds <= 160 * 1024 && (size_t)12 * 32 * W43_XT <= (size_t)2 * W43_BUF;
}

bool wino43_supported(const ConvShape &c, bool pool) {
    if (c.Cin % 8 != 0 || c.Cout % 64 != 0) return false;
    if (pool && (c.H < 2 || c.W < 2)) return false;
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    switch (w43_ttw(c, pool)) {
        case 8: return w43_geometry<8>(c, pool, g, lds, blocks);
        case 4: return w43_geometry<4>(c, pool, g, lds, blocks);
        case 2: return w43_geometry<2>(c, pool, g, lds, blocks);
        default: return w43_geometry<1>(c, pool, g, lds, blocks);
    }
}

// MFMA work the kernel issues for this shape (tile padding included): workgroups x 32 tiles x 64 channels x 36 positions x cin MACs
double wino43_issued_flops(const ConvShape &c, bool pool) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks = 0;
    bool ok;
    switch (w43_ttw(c, pool)) {
        case 8: ok = w43_geometry<8>(c, pool, g, lds, blocks); break;
        case 4: ok = w43_geometry<4>(c, pool, g, lds, blocks); break;
        case 2: ok = w43_geometry<2>(c, pool, g, lds, blocks); break;
        default: ok = w43_geometry<1>(c, pool, g, lds, blocks); break;
    }
    return ok ? 2.0 * (double)blocks * 32.0 * 64.0 * 36.0 * c.Cin : 0.0;
}

#if W43_CLK
// a ring of 256 launches, no synchronisation (the launches stay back to back); dumped when the process exits: the LAST launch of
// every (kernel, shape), i.e. the clock after that kernel has run for as long as the caller kept launching it
struct W43ClkLog { char what[32]; int H, W, Cin, Cout; };
static long long *w43_clk_ring = nullptr;
static W43ClkLog w43_clk_log[256];
static int w43_clk_n = 0;
static void w43_clk_dump() {
    if (w43_clk_n == 0) return;
    static long long v[256 * 4];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(v, w43_clk_ring, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return;
    const int first = w43_clk_n > 256 ? w43_clk_n - 256 : 0;
    for (int i = first; i < w43_clk_n; ++i) {
        const W43ClkLog &l = w43_clk_log[i & 255];
        bool last = true;  // of its (kernel, shape) in the ring
        for (int j = i + 1; j < w43_clk_n && last; ++j) {
            const W43ClkLog &m = w43_clk_log[j & 255];
            last = !(m.H == l.H && m.W == l.W && m.Cin == l.Cin && m.Cout == l.Cout && strcmp(m.what, l.what) == 0);
        }
        if (!last) continue;
        const long long *q = v + (i & 255) * 4;
        const double us = (double)(q[3] - q[2]) / 100.0;
        fprintf(stderr, "[stito clock] %-28s %dx%d %d->%d: workgroup in the middle of the grid ran %.1f us at %.0f MHz (launch %d)\n", l.what, l.H, l.W,
                l.Cin, l.Cout, us, us > 0 ? (double)(q[1] - q[0]) / us : 0.0, i);
    }
}
static long long *w43_clk_buf() {
    if (w43_clk_ring == nullptr) {
        if (hipMalloc(&w43_clk_ring, 256 * 4 * sizeof(long long)) != hipSuccess) return nullptr;
        atexit(w43_clk_dump);
    }
    return w43_clk_ring + (w43_clk_n & 255) * 4;
}
static void w43_clk_report(const char *what, const ConvShape &c, hipStream_t) {
    W43ClkLog &l = w43_clk_log[w43_clk_n & 255];
    snprintf(l.what, sizeof(l.what), "%s", what);
    l.H = c.H; l.W = c.W; l.Cin = c.Cin; l.Cout = c.Cout;
    ++w43_clk_n;
}
#define W43_CLK_ARM(G) (G).clk = w43_clk_buf();
#define W43_CLK_REPORT(WHAT, C, ST) w43_clk_report(WHAT, C, ST);
#else
#define W43_CLK_ARM(G) (G).clk = nullptr;
#define W43_CLK_REPORT(WHAT, C, ST)
#endif

// One workgroup of k_conv_wino43s / the f32 k_conv_wino43 fills a CU (138 - 148 KB of LDS, 240 - 256 registers per lane), so between a
// workgroup's last store and the next one's first copies the CU idles: W43_CLK stamps put 3 - 13 us of it between 40 - 50 us workgroups
// (DESIGN 4.3).  Keeping the workgroup alive instead (a persistent item loop) was built three times in round 6 and lost every time to the
// register allocation of the main loop.  What does NOT touch the kernel: launching the layer as N grids on N streams -- N hardware queues,
// each with a dispatcher of its own, so that when a CU falls free one of them usually has a workgroup ready.  Grid q takes the items
// [b0_q, b0_q + n_q) (whole XCD rounds of 256), the side streams fork from and join `st` by events (graph-capturable; the suite replays it),
// results are bit for bit those of one grid (an item does not know which grid it rode in).  STITO_W43_QUEUES = N for k_conv_wino43s (default 2; 1 = one
// grid), STITO_W43_QUEUES_F32 for the f32 kernel (default 1: measured neutral).
template <class LAUNCH>
static int w43_multi_queue_launch(int64_t blocks, hipStream_t st, LAUNCH &&launch, bool f32_kernel = false) {
    constexpr int MAXQ = 4, MAXDEV = 64;
    // (read per launch: the tests flip them)  Measured (tools/dual_grid_ab.sh, bench step, alternating runs on one box): k_conv_wino43s 1 -> 2 queues
    // 43.36 / 42.80 / 43.25 -> 42.97 / 42.42 / 42.80 ms; 3 and 4 queues 42.98 / 43.05; the f32 kernel neutral (43.23 / 42.82 / 43.35 with two): one grid
    const char *e_ = getenv(f32_kernel ? "STITO_W43_QUEUES_F32" : "STITO_W43_QUEUES");
    int nq = e_ ? atoi(e_) : (f32_kernel ? 1 : 2);
    nq = nq < 1 ? 1 : (nq > MAXQ ? MAXQ : nq);
    while (nq > 1 && blocks / nq < 512) --nq;   // at least two XCD rounds per grid
    if (nq <= 1) {
        launch(blocks, 0, st);
        return STITO_OK;
    }
    struct Side { hipStream_t q[MAXQ - 1]; hipEvent_t fork, join[MAXQ - 1]; bool ready; };
    static Side sides[MAXDEV];
    int dev = 0;
    STITO_HIP_CHECK(hipGetDevice(&dev));
    STITO_REQUIRE(dev >= 0 && dev < MAXDEV, STITO_E_UNSUPPORTED, "device index %d", dev);
    Side &sd = sides[dev];
    if (!sd.ready) {   // (one host thread drives a GPU: created on first use, kept for the life of the process)
        STITO_HIP_CHECK(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
        for (int i = 0; i < MAXQ - 1; ++i) {
            STITO_HIP_CHECK(hipStreamCreateWithFlags(&sd.q[i], hipStreamNonBlocking));
            STITO_HIP_CHECK(hipEventCreateWithFlags(&sd.join[i], hipEventDisableTiming));
        }
        sd.ready = true;
    }
    const int64_t per = (blocks / nq) / 256 * 256;
    STITO_HIP_CHECK(hipEventRecord(sd.fork, st));
    for (int i = 1; i < nq; ++i) STITO_HIP_CHECK(hipStreamWaitEvent(sd.q[i - 1], sd.fork, 0));
    launch(per, 0, st);
    for (int i = 1; i < nq; ++i) {
        const int64_t b0 = per * i, n = i == nq - 1 ? blocks - b0 : per;
        launch(n, (int)b0, sd.q[i - 1]);
        STITO_HIP_CHECK(hipEventRecord(sd.join[i - 1], sd.q[i - 1]));
    }
    for (int i = 1; i < nq; ++i) STITO_HIP_CHECK(hipStreamWaitEvent(st, sd.join[i - 1], 0));
    return STITO_OK;
}

template <int TTW, bool POOL>
static int launch_w43(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                      long long *trace, hipStream_t st, unsigned *amax_out) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    STITO_REQUIRE((w43_geometry<TTW>(c, POOL, g, lds, blocks)), STITO_E_UNSUPPORTED,
                  "conv (winograd F(4x4,3x3)): %dx%d map, %d channels does not fit the kernel's staging", c.H, c.W, c.Cin);
    g.trace = trace;
    g.amax_out = amax_out;
    W43_CLK_ARM(g)
    auto kern = trace ? k_conv_wino43<TTW, POOL, true> : k_conv_wino43<TTW, POOL, false>;
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (trace != nullptr || (256 % (c.Cout / 64)) != 0) {   // (the timeline build stamps by blockIdx; grids are cut at multiples of 256 items = whole pixel blocks)
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(W43_THREADS), lds, st, in, upk, scale, shift, out, g);
    } else {
        STITO_TRY(w43_multi_queue_launch(blocks, st, [&](int64_t n_blk, int b0, hipStream_t q) {
            Wino43Geom gq = g;
            gq.b0 = b0;
            hipLaunchKernelGGL(kern, dim3((unsigned)n_blk), dim3(W43_THREADS), lds, q, in, upk, scale, shift, out, gq);
        }, true));
    }
    STITO_LAUNCH_CHECK();
    W43_CLK_REPORT("k_conv_wino43 (f32 MFMA)", c, st)
    return STITO_OK;
}

int launch_wino43(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                  bool pool, long long *trace, hipStream_t st, unsigned *amax_out) {
    switch (w43_ttw(c, pool)) {
        case 8: return pool ? launch_w43<8, true>(in, upk, scale, shift, out, c, trace, st, amax_out) : launch_w43<8, false>(in, upk, scale, shift, out, c, trace, st, amax_out);
        case 4: return pool ? launch_w43<4, true>(in, upk, scale, shift, out, c, trace, st, amax_out) : launch_w43<4, false>(in, upk, scale, shift, out, c, trace, st, amax_out);
        case 2: return pool ? launch_w43<2, true>(in, upk, scale, shift, out, c, trace, st, amax_out) : launch_w43<2, false>(in, upk, scale, shift, out, c, trace, st, amax_out);
        default: return pool ? launch_w43<1, true>(in, upk, scale, shift, out, c, trace, st, amax_out) : launch_w43<1, false>(in, upk, scale, shift, out, c, trace, st, amax_out);
    }
}

// Hoisted input transform (MODE 2 then MODE 1): V slabs [pixel block][cin/4][36][pair][32][2] in `vbuf`.
template <int TTW>
static size_t w43_pre_bytes(const ConvShape &c, bool pool) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    if (!w43_geometry<TTW>(c, pool, g, lds, blocks)) return 0;
    return (size_t)(blocks / (c.Cout / 64)) * (size_t)(c.Cin / W43_K) * W43_V * sizeof(float);
}

size_t wino43_pre_workspace_bytes(const ConvShape &c, bool pool) {
    if (!wino43_supported(c, pool)) return 0;
    switch (w43_ttw(c, pool)) {
        case 8: return w43_pre_bytes<8>(c, pool);
        case 4: return w43_pre_bytes<4>(c, pool);
        case 2: return w43_pre_bytes<2>(c, pool);
        default: return w43_pre_bytes<1>(c, pool);
    }
}

template <int TTW, bool POOL>
static int launch_w43_pre(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                          float *vbuf, hipStream_t st, unsigned *amax_out) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    STITO_REQUIRE((w43_geometry<TTW>(c, POOL, g, lds, blocks)), STITO_E_UNSUPPORTED,
                  "conv (winograd F(4x4,3x3)): %dx%d map, %d channels does not fit the kernel's staging", c.H, c.W, c.Cin);
    {   // V slabs: the chunks of a pixel block split over as many workgroups as it takes to fill the chip a few times
        Wino43Geom gv = g;
        const int64_t m_blocks = blocks / (c.Cout / 64);
        const int n_chunks = c.Cin / W43_K;
        int ncg = 1;
        while (m_blocks * ncg < 1024 && n_chunks % (4 * ncg) == 0 && n_chunks / (2 * ncg) >= 4) ncg *= 2;  // even share, >= 4 chunks
        gv.n_cgroups = ncg;
        auto kern = k_conv_wino43<TTW, POOL, false, 2>;
        const size_t lds_t = ((size_t)2 * W43_V + 2 * W43Patch<TTW>::PFL) * sizeof(float);  // V buffers + patch buffers (no weights)
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
        hipLaunchKernelGGL(kern, dim3((unsigned)(m_blocks * ncg)), dim3(W43_THREADS), lds_t, st, in, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, vbuf, gv);
        STITO_LAUNCH_CHECK();
    }
    auto kern = k_conv_wino43<TTW, POOL, false, 1>;
    const size_t lds1 = (size_t)2 * W43_BUF * sizeof(float);
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    // grid in whole XCD rounds: 8 XCDs x (groups of 8 pixel blocks) x (groups of 4 channel tiles) x 32 workgroups
    const int64_t m_blocks = blocks / (c.Cout / 64);
    g.n_mblocks = (int)m_blocks;
    // channel tiles side by side on an XCD: swept 1..32 at 512 streams (tools/conv_bench.py --modes 9): 4 is best for 8 tiles
    // (conv_block4: 3.17 / 5.69 ms against 3.63 / 6.59 for 1), 8 from 16 tiles up (1-4 % over 4), 32 loses 15 % at 32 tiles
    const int n_tiles = c.Cout / 64;
    const int a = n_tiles >= 16 ? 8 : 4;
    STITO_REQUIRE(n_tiles % a == 0, STITO_E_UNSUPPORTED, "conv (hoisted input transform): cout %d", c.Cout);
    g.ct_group = a;
    g.amax_out = amax_out;
    const int bm = 32 / a;
    const int64_t m_groups = ((m_blocks + 7) / 8 + bm - 1) / bm;
    blocks = 8 * m_groups * (n_tiles / a) * 32;
    STITO_REQUIRE(blocks < (1ll << 31), STITO_E_UNSUPPORTED, "conv (hoisted input transform): grid");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(W43_THREADS), lds1, st, (const float *)vbuf, upk, scale, shift, out, g);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

int launch_wino43_pre(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                      bool pool, float *vbuf, size_t vbuf_bytes, hipStream_t st, unsigned *amax_out) {
    const size_t need = wino43_pre_workspace_bytes(c, pool);
    STITO_REQUIRE(need > 0 && vbuf != nullptr && vbuf_bytes >= need, STITO_E_WORKSPACE,
                  "conv (winograd F(4x4,3x3), hoisted input transform): workspace have %zu need %zu", vbuf_bytes, need);
    switch (w43_ttw(c, pool)) {
        case 8: return pool ? launch_w43_pre<8, true>(in, upk, scale, shift, out, c, vbuf, st, amax_out) : launch_w43_pre<8, false>(in, upk, scale, shift, out, c, vbuf, st, amax_out);
        case 4: return pool ? launch_w43_pre<4, true>(in, upk, scale, shift, out, c, vbuf, st, amax_out) : launch_w43_pre<4, false>(in, upk, scale, shift, out, c, vbuf, st, amax_out);
        case 2: return pool ? launch_w43_pre<2, true>(in, upk, scale, shift, out, c, vbuf, st, amax_out) : launch_w43_pre<2, false>(in, upk, scale, shift, out, c, vbuf, st, amax_out);
        default: return pool ? launch_w43_pre<1, true>(in, upk, scale, shift, out, c, vbuf, st, amax_out) : launch_w43_pre<1, false>(in, upk, scale, shift, out, c, vbuf, st, amax_out);
    }
}

// Split-precision variant of the hoisted path: stream maxima -> MODE 3 (V slabs as scaled f16 halves) -> k_conv_wino43s.
// Workspace: the V slabs (the same bytes as MODE 2's) followed by one unsigned per stream.
bool wino43_split_supported(const ConvShape &c, bool pool) {
    return c.Cin % 64 == 0 && c.Cout % 256 == 0 && wino43_supported(c, pool) && ((int64_t)c.Cin * c.H * c.W) % 8 == 0;
}

size_t wino43_split_workspace_bytes(const ConvShape &c, bool pool) {
    if (!wino43_split_supported(c, pool)) return 0;
    return align_up(wino43_pre_workspace_bytes(c, pool), 256) + align_up((size_t)c.S * sizeof(unsigned), 256);
}

template <int TTW, bool POOL>
static int launch_w43_split(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                            char *ws, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    STITO_REQUIRE((w43_geometry<TTW>(c, POOL, g, lds, blocks)), STITO_E_UNSUPPORTED,
                  "conv (winograd F(4x4,3x3)): %dx%d map, %d channels does not fit the kernel's staging", c.H, c.W, c.Cin);
    const int64_t m_blocks = blocks / (c.Cout / 64);
    const unsigned *amax = amax_in;
    if (amax_in == nullptr) {   // stream maxima (not supplied by the layer that produced `in`)
        unsigned *amax_ws = (unsigned *)(ws + align_up((size_t)m_blocks * (c.Cin / W43_K) * W43_V * sizeof(float), 256));
        amax = amax_ws;
        STITO_TRY(zero_async(amax_ws, (size_t)c.S * sizeof(unsigned), st));
        const int64_t per_stream = (int64_t)c.Cin * c.H * c.W;
        int splits = (int)((per_stream / 4 + 256 * 16 - 1) / (256 * 16));  // >= 16 float4 per thread
        const int cap = (4096 + c.S - 1) / c.S;                            // ~16 workgroups per CU in total
        splits = splits > cap ? cap : (splits < 1 ? 1 : splits);
        hipLaunchKernelGGL(k_stream_absmax, dim3((unsigned)splits, (unsigned)c.S), dim3(256), 0, st, in, per_stream, amax_ws);
        STITO_LAUNCH_CHECK();
    }
    {   // V slabs (MODE 3): grid as MODE 2
        Wino43Geom gv = g;
        const int n_chunks = c.Cin / W43_K;
        int ncg = 1;
        while (m_blocks * ncg < 1024 && n_chunks % (4 * ncg) == 0 && n_chunks / (2 * ncg) >= 4) ncg *= 2;
        gv.n_cgroups = ncg;
        auto kern = k_conv_wino43<TTW, POOL, false, 3>;
        const size_t lds_t = ((size_t)2 * W43_V + 2 * W43Patch<TTW>::PFL) * sizeof(float);  // V buffers + patch buffers (no weights)
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
        hipLaunchKernelGGL(kern, dim3((unsigned)(m_blocks * ncg)), dim3(W43_THREADS), lds_t, st, in, (const float *)nullptr,
                           (const float *)amax, (const float *)nullptr, (float *)ws, gv);
        STITO_LAUNCH_CHECK();
    }
    auto kern = k_conv_wino43s<TTW, POOL>;
    const size_t lds1 = (size_t)3 * S43_SLAB;
    static_assert((size_t)12 * 32 * W43_XT * sizeof(float) <= (size_t)3 * S43_SLAB, "epilogue exchange fits the slab ring");
    STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    g.n_mblocks = (int)m_blocks;
    const int n_tiles = c.Cout / 64;
    int a = n_tiles >= 16 ? 8 : (n_tiles >= 8 ? 4 : 2);  // swept again with the round's final kernels (profiles/round3_stream_loop_experiments.txt, 15.)
    if (const char *e = getenv("STITO_W43S_CTG")) { const int ae = atoi(e); if (ae >= 1 && ae <= 32 && (ae & (ae - 1)) == 0 && n_tiles % ae == 0) a = ae; }  // tuning aid (tools/conv_bench.py)
    STITO_REQUIRE(a >= 1 && a <= 32 && (a & (a - 1)) == 0 && n_tiles % a == 0, STITO_E_UNSUPPORTED, "conv (split-precision winograd): cout %d", c.Cout);
    g.ct_group = a;
    const int bm = 32 / a;
    const int64_t m_groups = ((m_blocks + 7) / 8 + bm - 1) / bm;
    blocks = 8 * m_groups * (n_tiles / a) * 32;
    STITO_REQUIRE(blocks < (1ll << 31), STITO_E_UNSUPPORTED, "conv (split-precision winograd): grid");
    g.amax_out = amax_out;
    const float *u_inv = upk + (size_t)36 * c.Cout * c.Cin + 1;
    W43_CLK_ARM(g)
    // the layer as several grids on several hardware queues (w43_multi_queue_launch): the queues' dispatchers overlap the hand-over
    STITO_TRY(w43_multi_queue_launch(blocks, st, [&](int64_t n_blk, int b0, hipStream_t q) {
        Wino43Geom gq = g;
        gq.b0 = b0;
        hipLaunchKernelGGL(kern, dim3((unsigned)n_blk), dim3(W43_THREADS), lds1, q, (const char *)ws, (const char *)upk, scale, shift, out, gq,
                           (const unsigned *)amax, u_inv);
    }));
    STITO_LAUNCH_CHECK();
    W43_CLK_REPORT("k_conv_wino43s (f16 MFMA)", c, st)
    return STITO_OK;
}

// Two-sweep variant: workspace = V slabs of the pixel-block pairs | stream maxima | per-workgroup partial outputs (256 KB each).
// -> the grid of ONE sweep set (SWSPLIT launches twice as many); xcd_m / ct_group: the kernel's workgroup order
template <int TTW>
static int64_t w43_split2_grid(const ConvShape &c, bool pool, int64_t &m_pairs, int &ct_group, int &xcd_m) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    if (!w43_geometry<TTW>(c, pool, g, lds, blocks)) return 0;
    const int64_t m_blocks = blocks / (c.Cout / 64);
    m_pairs = (m_blocks + 1) / 2;
    const int n_tiles_all = c.Cout / 64;   // a multiple of 4 (wino43_split_supported)
    // XCD split: the xm in {8, 4, 2, 1} with the fewest slab streams per L2, pairs / xm + tiles / (8 / xm); 8 on a tie
    int xm = 8;
    for (int cand = 4; cand >= 1; cand >>= 1)
        if (n_tiles_all % (8 / cand) == 0 && (m_pairs + cand - 1) / cand + n_tiles_all / (8 / cand) < (m_pairs + xm - 1) / xm + n_tiles_all / (8 / xm)) xm = cand;
    if (const char *e = getenv("STITO_W43S2_XM")) { const int xe = atoi(e); if ((xe == 8 || xe == 4 || xe == 2 || xe == 1) && n_tiles_all % (8 / xe) == 0) xm = xe; }  // tuning aid
    xcd_m = xm;
    const int n_tiles = n_tiles_all / (8 / xm);
    int a = 8;
    while (a > 1 && n_tiles % a != 0) a >>= 1;
    if (const char *e = getenv("STITO_W43S_CTG")) { const int ae = atoi(e); if (ae >= 1 && ae <= 32 && (ae & (ae - 1)) == 0 && n_tiles % ae == 0) a = ae; }
    if (a < 1 || a > 32 || (a & (a - 1)) != 0 || n_tiles % a != 0) return 0;
    ct_group = a;
    const int bm = 32 / a;
    const int64_t m_groups = ((m_pairs + xm - 1) / xm + bm - 1) / bm;
    return 8 * m_groups * (n_tiles / a) * 32;
}

static int64_t w43_split2_grid_any(const ConvShape &c, bool pool, int64_t &m_pairs, int &ct_group) {
    int xm;
    switch (w43_ttw(c, pool)) {
        case 8: return w43_split2_grid<8>(c, pool, m_pairs, ct_group, xm);
        case 4: return w43_split2_grid<4>(c, pool, m_pairs, ct_group, xm);
        case 2: return w43_split2_grid<2>(c, pool, m_pairs, ct_group, xm);
        default: return w43_split2_grid<1>(c, pool, m_pairs, ct_group, xm);
    }
}

// the two sweeps of an item as two workgroups when that still fits one round of the device (see the kernel); STITO_W43S2_SWSPLIT=0 / 1 forces
// Measured (tools/small_conv_ab.sh, 64 / 128 streams x 257 frames): 1024 -> 1024 0.348 -> 0.293 ms, 1024 -> 2048 0.346 -> 0.231, 2048 -> 2048
// 0.640 -> 0.388; with 512 input channels the halves are too short to carry the second workgroup's prologue and its wait
// (512 -> 1024: 0.201 -> 0.236), hence the channel threshold.
static bool w43_split2_sweep_split(int64_t m_pairs, int cout, int cin) {
    if (const char *e = getenv("STITO_W43S2_SWSPLIT")) return atoi(e) != 0;   // (read per launch: the tests flip it)
    DeviceInfo d;
    if (device_info(d) != STITO_OK) return false;
    return cin >= 1024 && 2 * m_pairs * (cout / 64) <= d.cus;
}

// f16-pipe FLOPs the two-sweep kernel issues: pixel-block pairs (a padded half included) x channel tiles x 64 x 64 x 36 x cin x 3 products
double wino43_split2_issued_flops(const ConvShape &c, bool pool) {
    if (!wino43_split_supported(c, pool)) return 0.0;
    int64_t m_pairs = 0;
    int a;
    if (w43_split2_grid_any(c, pool, m_pairs, a) <= 0) return 0.0;
    return 3.0 * 2.0 * (double)m_pairs * (c.Cout / 64) * 64.0 * 64.0 * 36.0 * c.Cin;
}

size_t wino43_split2_workspace_bytes(const ConvShape &c, bool pool) {
    if (!wino43_split_supported(c, pool)) return 0;
    int64_t m_pairs = 0;
    int a;
    const int64_t grid = w43_split2_grid_any(c, pool, m_pairs, a);
    if (grid <= 0 || grid >= (1ll << 31)) return 0;
    return align_up((size_t)m_pairs * 2 * (size_t)(c.Cin / 16) * 3 * S43B_PART, 256) + align_up((size_t)c.S * sizeof(unsigned), 256) +
           (size_t)grid * 2 * 16 * W43_THREADS * sizeof(f32x4) + align_up((size_t)grid * sizeof(unsigned), 256);   // ... + one flag per workgroup item (SWSPLIT)
}

template <int TTW, bool POOL>
static int launch_w43_split2(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                             char *ws, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    STITO_REQUIRE((w43_geometry<TTW>(c, POOL, g, lds, blocks)), STITO_E_UNSUPPORTED,
                  "conv (winograd F(4x4,3x3)): %dx%d map, %d channels does not fit the kernel's staging", c.H, c.W, c.Cin);
    int64_t m_pairs = 0;
    int a = 4, xm = 8;
    const int64_t grid = w43_split2_grid<TTW>(c, POOL, m_pairs, a, xm);
    STITO_REQUIRE(grid > 0 && grid < (1ll << 30), STITO_E_UNSUPPORTED, "conv (two-sweep split-precision winograd): grid / cout %d", c.Cout);
    const size_t vbytes = align_up((size_t)m_pairs * 2 * (size_t)(c.Cin / 16) * 3 * S43B_PART, 256);
    unsigned *amax_ws = (unsigned *)(ws + vbytes);
    const unsigned *amax = amax_in != nullptr ? amax_in : amax_ws;
    f32x4 *partial = (f32x4 *)(ws + vbytes + align_up((size_t)c.S * sizeof(unsigned), 256));
    if (amax_in == nullptr) {   // stream maxima (not supplied by the layer that produced `in`)
        STITO_TRY(zero_async(amax_ws, (size_t)c.S * sizeof(unsigned), st));
        const int64_t per_stream = (int64_t)c.Cin * c.H * c.W;
        int splits = (int)((per_stream / 4 + 256 * 16 - 1) / (256 * 16));
        const int cap = (4096 + c.S - 1) / c.S;
        splits = splits > cap ? cap : (splits < 1 ? 1 : splits);
        hipLaunchKernelGGL(k_stream_absmax, dim3((unsigned)splits, (unsigned)c.S), dim3(256), 0, st, in, per_stream, amax_ws);
        STITO_LAUNCH_CHECK();
    }
    {   // V slabs (MODE 4) of 2 * m_pairs pixel blocks (a block past the map transforms to zeros)
        Wino43Geom gv = g;
        const int n_chunks = c.Cin / W43_K;
        const int64_t m_blocks2 = 2 * m_pairs;
        int ncg = 1;
        while (m_blocks2 * ncg < 1024 && n_chunks % (4 * ncg) == 0 && n_chunks / (2 * ncg) >= 4) ncg *= 2;
        gv.n_cgroups = ncg;
        auto kern = k_conv_wino43<TTW, POOL, false, 4>;
        const size_t lds_t = ((size_t)2 * W43_V + 2 * W43Patch<TTW>::PFL) * sizeof(float);  // V buffers + patch buffers (no weights)
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
        hipLaunchKernelGGL(kern, dim3((unsigned)(m_blocks2 * ncg)), dim3(W43_THREADS), lds_t, st, in, (const float *)nullptr,
                           (const float *)amax, (const float *)nullptr, (float *)ws, gv);
        STITO_LAUNCH_CHECK();
    }
    const bool swsplit = w43_split2_sweep_split(m_pairs, c.Cout, c.Cin);
    unsigned *flags = (unsigned *)((char *)partial + (size_t)grid * 2 * 16 * W43_THREADS * sizeof(f32x4));
    const size_t lds1 = (size_t)3 * S43B_SLAB;
    static_assert((size_t)12 * 32 * W43_XT * sizeof(float) <= (size_t)3 * S43B_SLAB, "epilogue exchange fits the slab ring");
    g.n_mblocks = (int)m_pairs;
    g.ct_group = a;
    g.xcd_m = xm;
    g.amax_out = amax_out;
    const float *u_inv = upk + (size_t)36 * c.Cout * c.Cin + 1;
    W43_CLK_ARM(g)
    if (swsplit) {
        STITO_TRY(zero_async(flags, (size_t)grid * sizeof(unsigned), st));
        auto kern = k_conv_wino43s2<TTW, POOL, true>;
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3((unsigned)(2 * grid)), dim3(W43_THREADS), lds1, st, (const char *)ws, (const char *)upk, scale, shift, out, g,
                           (const unsigned *)amax, u_inv, partial, flags);
    } else {
        auto kern = k_conv_wino43s2<TTW, POOL, false>;
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W43_THREADS), lds1, st, (const char *)ws, (const char *)upk, scale, shift, out, g,
                           (const unsigned *)amax, u_inv, partial, flags);
    }
    STITO_LAUNCH_CHECK();
    W43_CLK_REPORT("k_conv_wino43s2 (f16 MFMA)", c, st)
    return STITO_OK;
}

int launch_wino43_split2(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                         bool pool, void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    const size_t need = wino43_split2_workspace_bytes(c, pool);
    STITO_REQUIRE(need > 0 && ws != nullptr && ws_bytes >= need, STITO_E_WORKSPACE,
                  "conv (two-sweep split-precision winograd F(4x4,3x3)): workspace have %zu need %zu", ws_bytes, need);
    char *w = (char *)ws;
    switch (w43_ttw(c, pool)) {
        case 8: return pool ? launch_w43_split2<8, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split2<8, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        case 4: return pool ? launch_w43_split2<4, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split2<4, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        case 2: return pool ? launch_w43_split2<2, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split2<2, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        default: return pool ? launch_w43_split2<1, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split2<1, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
    }
}

// Six-sweep variant (k_conv_wino43s3): workspace = V slabs of the pixel-block quads | stream maxima | per-workgroup partial outputs (1 MB each).
static int64_t w43_split3_grid_any(const ConvShape &c, bool pool, int64_t &m_quads, int &ct_group);
bool wino43_split3_supported(const ConvShape &c, bool pool) {
    if (!(c.Cin % 64 == 0 && c.Cout % 512 == 0 && wino43_supported(c, pool) && ((int64_t)c.Cin * c.H * c.W) % 8 == 0)) return false;
    int64_t m_quads = 0;
    int a;
    const int64_t grid = w43_split3_grid_any(c, pool, m_quads, a);   // "supported" means the launcher finds a grid for it
    return grid > 0 && grid < (1ll << 31);
}

template <int TTW>
static int64_t w43_split3_grid(const ConvShape &c, bool pool, int64_t &m_quads, int &ct_group, int &xcd_m) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    if (!w43_geometry<TTW>(c, pool, g, lds, blocks)) return 0;
    const int64_t m_blocks = blocks / (c.Cout / 64);
    m_quads = (m_blocks + 3) / 4;
    const int n_tiles_all = c.Cout / 128;   // a multiple of 4 (wino43_split3_supported)
    // XCD split (see the kernel): the xm in {8, 4, 2} with the fewest slab bytes per L2, quads / xm + tiles / (8 / xm); 8 on a tie
    int xm = 8;
    for (int cand = 4; cand >= 2; cand >>= 1)
        if (n_tiles_all % (8 / cand) == 0 && (m_quads + cand - 1) / cand + n_tiles_all / (8 / cand) < (m_quads + xm - 1) / xm + n_tiles_all / (8 / xm)) xm = cand;
    if (const char *e = getenv("STITO_W43S3_XM")) { const int xe = atoi(e); if ((xe == 8 || xe == 4 || xe == 2) && n_tiles_all % (8 / xe) == 0) xm = xe; }  // tuning aid
    xcd_m = xm;
    const int n_tiles = n_tiles_all / (8 / xm);
    const int64_t mq_loc = (m_quads + xm - 1) / xm;
    // channel tiles side by side in a round of 32 workgroups: a power of two that divides the tile count (cout 1536 has 12 tiles,
    // 6 per XCD half: ADVICE r5), grown while a round is not full
    int a = 4;
    while (a > 1 && n_tiles % a != 0) a >>= 1;
    while (a < n_tiles && a < 32 && 32 / a > mq_loc && n_tiles % (2 * a) == 0) a *= 2;
    if (const char *e = getenv("STITO_W43S3_CTG")) { const int ae = atoi(e); if (ae >= 1 && ae <= 32 && (ae & (ae - 1)) == 0 && n_tiles % ae == 0) a = ae; }
    if (n_tiles % a != 0) return 0;
    ct_group = a;
    const int bm = 32 / a;
    const int64_t m_groups = (mq_loc + bm - 1) / bm;
    return 8 * m_groups * (n_tiles / a) * 32;
}

static int64_t w43_split3_grid_any(const ConvShape &c, bool pool, int64_t &m_quads, int &ct_group) {
    int xm;
    switch (w43_ttw(c, pool)) {
        case 8: return w43_split3_grid<8>(c, pool, m_quads, ct_group, xm);
        case 4: return w43_split3_grid<4>(c, pool, m_quads, ct_group, xm);
        case 2: return w43_split3_grid<2>(c, pool, m_quads, ct_group, xm);
        default: return w43_split3_grid<1>(c, pool, m_quads, ct_group, xm);
    }
}

// f16-pipe FLOPs the six-sweep kernel issues: pixel-block quads (padded blocks included) x 128-cout tiles x 128 x 128 x 36 x cin x 3 products
double wino43_split3_issued_flops(const ConvShape &c, bool pool) {
    if (!wino43_split3_supported(c, pool)) return 0.0;
    int64_t m_quads = 0;
    int a;
    if (w43_split3_grid_any(c, pool, m_quads, a) <= 0) return 0.0;
    return 3.0 * 2.0 * (double)m_quads * (c.Cout / 128) * 128.0 * 128.0 * 36.0 * c.Cin;
}

// workgroups of the six-sweep kernel that have work (pixel-block quads x 128-cout tiles): below about one per CU the two-sweep
// kernel's four times as many, four times shorter workgroups finish sooner (stito_cnn14_forward's choice per call)
int64_t wino43_split3_workgroups(const ConvShape &c, bool pool) {
    if (!wino43_split3_supported(c, pool)) return 0;
    int64_t m_quads = 0;
    int a;
    if (w43_split3_grid_any(c, pool, m_quads, a) <= 0) return 0;
    return m_quads * (c.Cout / 128);
}

static size_t w43_split3_vbytes(const ConvShape &c, int64_t m_quads) { return align_up((size_t)m_quads * 6 * (size_t)(c.Cin >> 3) * S43B_PART, 256); }

size_t wino43_split3_workspace_bytes(const ConvShape &c, bool pool) {
    if (!wino43_split3_supported(c, pool)) return 0;
    int64_t m_quads = 0;
    int a;
    const int64_t grid = w43_split3_grid_any(c, pool, m_quads, a);
    if (grid <= 0 || grid >= (1ll << 31)) return 0;
    return w43_split3_vbytes(c, m_quads) + align_up((size_t)c.S * sizeof(unsigned), 256) + (size_t)grid * (8 * 2 * 4 * 20 * 64) * sizeof(f32x4) +
           align_up((size_t)grid * sizeof(unsigned), 256);   // ... + one row count per workgroup item (SWSPLIT)
}

// the six sweeps of an item as six workgroups: OFF unless STITO_W43S3_SWSPLIT=1.  Built and measured in round 6 for the small batches
// whose items alone leave CUs idle (tools/small_conv_ab.sh, profiles/round6_small_conv_ab.txt): it takes 64-stream conv_block6.conv2 from
// 1.53 to 0.51 ms -- but the two-sweep kernel with ITS sweeps split does the same layer in 0.39 (the six-sweep kernel's transform pass and
// its 1 MB of partial outputs per item do not shrink with the batch), and that is what stito_cnn14_forward takes for small batches.
static bool w43_split3_sweep_split(int64_t, int) {
    if (const char *e = getenv("STITO_W43S3_SWSPLIT")) return atoi(e) != 0;   // (read per launch: the tests flip it)
    return false;
}

template <int TTW, bool POOL>
static int launch_w43_split3(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                             char *ws, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    Wino43Geom g;
    size_t lds;
    int64_t blocks;
    STITO_REQUIRE((w43_geometry<TTW>(c, POOL, g, lds, blocks)), STITO_E_UNSUPPORTED,
                  "conv (winograd F(4x4,3x3)): %dx%d map, %d channels does not fit the kernel's staging", c.H, c.W, c.Cin);
    int64_t m_quads = 0;
    int a = 4, xm = 8;
    const int64_t grid = w43_split3_grid<TTW>(c, POOL, m_quads, a, xm);
    STITO_REQUIRE(grid > 0 && grid < (1ll << 31), STITO_E_UNSUPPORTED, "conv (six-sweep split-precision winograd): grid / cout %d", c.Cout);
    const size_t vbytes = w43_split3_vbytes(c, m_quads);
    unsigned *amax_ws = (unsigned *)(ws + vbytes);
    const unsigned *amax = amax_in != nullptr ? amax_in : amax_ws;
    f32x4 *partial = (f32x4 *)(ws + vbytes + align_up((size_t)c.S * sizeof(unsigned), 256));
    if (amax_in == nullptr) {   // stream maxima (not supplied by the layer that produced `in`)
        STITO_TRY(zero_async(amax_ws, (size_t)c.S * sizeof(unsigned), st));
        const int64_t per_stream = (int64_t)c.Cin * c.H * c.W;
        int splits = (int)((per_stream / 4 + 256 * 16 - 1) / (256 * 16));
        const int cap = (4096 + c.S - 1) / c.S;
        splits = splits > cap ? cap : (splits < 1 ? 1 : splits);
        hipLaunchKernelGGL(k_stream_absmax, dim3((unsigned)splits, (unsigned)c.S), dim3(256), 0, st, in, per_stream, amax_ws);
        STITO_LAUNCH_CHECK();
    }
    {   // V slabs (MODE 5) of 4 * m_quads pixel blocks (a block past the map transforms to zeros)
        Wino43Geom gv = g;
        const int n_chunks = c.Cin / W43_K;
        const int64_t m_blocks4 = 4 * m_quads;
        int ncg = 1;
        while (m_blocks4 * ncg < 1024 && n_chunks % (4 * ncg) == 0 && n_chunks / (2 * ncg) >= 4) ncg *= 2;
        gv.n_cgroups = ncg;
        auto kern = k_conv_wino43<TTW, POOL, false, 5>;
        const size_t lds_t = ((size_t)2 * W43_V + 2 * W43Patch<TTW>::PFL) * sizeof(float);  // V buffers + patch buffers (no weights)
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
        hipLaunchKernelGGL(kern, dim3((unsigned)(m_blocks4 * ncg)), dim3(W43_THREADS), lds_t, st, in, (const float *)nullptr,
                           (const float *)amax, (const float *)nullptr, (float *)ws, gv);
        STITO_LAUNCH_CHECK();
    }
    const size_t lds1 = (size_t)3 * S43B_SLAB;
    g.n_mblocks = (int)m_quads;
    g.ct_group = a;
    g.xcd_m = xm;
    g.amax_out = amax_out;
    const float *u_inv = upk + (size_t)36 * c.Cout * c.Cin + 1;
    unsigned *counts = (unsigned *)((char *)partial + (size_t)grid * (8 * 2 * 4 * 20 * 64) * sizeof(f32x4));
    W43_CLK_ARM(g)
    if (w43_split3_sweep_split(m_quads, c.Cout) && 6 * grid < (1ll << 31)) {
        STITO_TRY(zero_async(counts, (size_t)grid * sizeof(unsigned), st));
        auto kern = k_conv_wino43s3<TTW, POOL, true>;
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3((unsigned)(6 * grid)), dim3(W43_THREADS), lds1, st, (const char *)ws, (const char *)upk, scale, shift, out, g,
                           (const unsigned *)amax, u_inv, partial, counts);
    } else {
        auto kern = k_conv_wino43s3<TTW, POOL, false>;
        STITO_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W43_THREADS), lds1, st, (const char *)ws, (const char *)upk, scale, shift, out, g,
                           (const unsigned *)amax, u_inv, partial, counts);
    }
    STITO_LAUNCH_CHECK();
    W43_CLK_REPORT("k_conv_wino43s3 (f16 MFMA)", c, st)
    return STITO_OK;
}

int launch_wino43_split3(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                         bool pool, void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    const size_t need = wino43_split3_workspace_bytes(c, pool);
    STITO_REQUIRE(need > 0 && ws != nullptr && ws_bytes >= need, STITO_E_WORKSPACE,
                  "conv (six-sweep split-precision winograd F(4x4,3x3)): workspace have %zu need %zu", ws_bytes, need);
    char *w = (char *)ws;
    switch (w43_ttw(c, pool)) {
        case 8: return pool ? launch_w43_split3<8, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split3<8, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        case 4: return pool ? launch_w43_split3<4, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split3<4, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        case 2: return pool ? launch_w43_split3<2, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split3<2, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        default: return pool ? launch_w43_split3<1, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split3<1, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
    }
}

int launch_wino43_split(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                        bool pool, void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in, unsigned *amax_out) {
    const size_t need = wino43_split_workspace_bytes(c, pool);
    STITO_REQUIRE(need > 0 && ws != nullptr && ws_bytes >= need, STITO_E_WORKSPACE,
                  "conv (split-precision winograd F(4x4,3x3)): workspace have %zu need %zu", ws_bytes, need);
    char *w = (char *)ws;
    switch (w43_ttw(c, pool)) {
        case 8: return pool ? launch_w43_split<8, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split<8, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        case 4: return pool ? launch_w43_split<4, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split<4, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        case 2: return pool ? launch_w43_split<2, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split<2, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
        default: return pool ? launch_w43_split<1, true>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out) : launch_w43_split<1, false>(in, upk, scale, shift, out, c, w, st, amax_in, amax_out);
    }
}

int pack_wino43(const float *w_oihw, int cout, int cin, float *packed, hipStream_t st) {
    const int64_t n = (int64_t)cout * cin;
    hipLaunchKernelGGL(k_pack_wino43, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_oihw, cout, cin, packed);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

}  // namespace stito
