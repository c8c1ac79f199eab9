// compressor.hip -- juce::dsp::Compressor<float> (peak ballistics + VCA) for a population of streams,
// time-parallel.  Replaces BasicCompressor.process (reference st_ito/effects.py:873-897, pedalboard.Compressor =
// juce::dsp::Compressor: BallisticsFilter::processSample + the gain computer) inside the per-candidate loop of
// st_ito/style_transfer.py:512-521.
//
// The envelope is a switching one-pole,
//     y' = v + c (y - v),   c = (v > y) ? c_att : c_rel,   v = |x|,
// serial in time and not linear, so it has no associative scan in the usual sense.  It does have one over the
// (max, +) structure.  Both candidates are increasing affine maps of y and the selected one is the larger
// (c_att <= c_rel) or the smaller (c_att > c_rel; then run the same thing on z = -y), so one step is
//     z' = max(c_att z + a, c_rel z + r),      (a, r) = s ((1 - c_att) v, (1 - c_rel) v),  s = +-1,  env = |z|.
// A step is a max of two increasing affine maps; a composition of K steps is a max over the 2^K attack/release
// paths, and all paths with the same number j of attack steps share the slope c_att^j c_rel^(K-j): only the
// largest intercept of each slope can ever win.  A block of K samples therefore IS the function
//     F(z) = max_{j=0..K} (S_j z + B_j),        S_j = c_att^j c_rel^(K-j),
// with K+1 intercepts that build up by   B'_j = max(c_att B_{j-1} + a_n, c_rel B_j + r_n)   -- K+1 lanes, K steps,
// independent of every other block.  With K = 15 a block is a 16-lane DPP row:
//
//   k_comp_blockfn   (parallel over all blocks)  intercepts B_0..B_15 of every full 15-sample block,
//   k_comp_blockscan (serial over blocks, 4 lanes per stream) z at every block boundary: 16 terms per 15 samples
//                    instead of 15 dependent steps -- ~9 instructions of the one serial wave per block, where
//                    walking the samples costs 3 per SAMPLE (the kernel this replaces: 6.5 ms at 512 x 480 000),
//   k_comp_apply     (parallel over all blocks)  re-walks each block from its boundary state with the reference's
//                    own per-sample operations, v + c (y - v), each rounded, and applies the VCA in the same pass
//                    (no envelope buffer in HBM).
//
// Rounding: inside a block every sample goes through the reference's float operations in the reference's order (the
// first block is bit-identical to a sequential walk); the boundary states come out of a different, shorter
// sequence -- every term a sum of same-signed products, no cancellation -- and the map is contractive, so a
// boundary's few ulp decay instead of accumulating.  Parity bound of the effect stays 2e-5 of peak.
#include "common.h"
#include "dsp_view.h"
#include "comp_scan.inc"

namespace stito {

static constexpr int CB_K = 15;        // samples per block; K + 1 = 16 intercepts = one DPP row
static constexpr int CB_TILE = 128;    // blocks per workgroup of k_comp_blockfn (4 waves x 8 iterations x 4 rows)
static constexpr int CB_D = STITO_COMP_SCAN_DEPTH;  // blocks in flight per lane in k_comp_blockscan (register ring)
static constexpr int CA_TILE = 256;    // blocks per workgroup of k_comp_apply (one block per thread)
static constexpr float CB_CMIN = 1.0e-30f;  // c = 0 (times below 1e-3 ms) would turn -inf * c into NaN in the intercept
                                            // recurrence; 1e-30 z is below every ulp that matters and -inf stays -inf

typedef float cb_f2 __attribute__((ext_vector_type(2)));
typedef float cb_f4 __attribute__((ext_vector_type(4)));

struct CompGeom {
    int64_t L;
    int64_t n_blocks;   // ceil(L / K): blocks k_comp_apply walks
    int64_t n_fn;       // n_blocks - 1: blocks whose function is needed (every one but the last is full)
    int64_t fn_stride;  // floats between streams in the intercept array: (n_fn + 2 CB_D) * 16 (the ring over-reads)
    int64_t z_stride;   // floats between streams in the boundary-state array: round_up(n_fn, CB_D) + CB_D
};

static CompGeom comp_geometry(int64_t L) {
    CompGeom g;
    g.L = L;
    g.n_blocks = (L + CB_K - 1) / CB_K;
    g.n_fn = g.n_blocks - 1;
    g.fn_stride = (g.n_fn + 2 * CB_D) * 16;
    g.z_stride = (g.n_fn + CB_D - 1) / CB_D * CB_D + CB_D;
    return g;
}

size_t compressor_workspace_bytes(int n_streams, int64_t n_samples) {
    const CompGeom g = comp_geometry(n_samples);
    return align_up((size_t)n_streams * g.fn_stride * sizeof(float), 256) + align_up((size_t)n_streams * g.z_stride * sizeof(float), 256);
}

struct CompCoef {
    float thr, thr_inv, p, cat, crl, sg;
};
__device__ __forceinline__ CompCoef comp_coef(const double *coef, int cand) {
    const double *cf = coef + (int64_t)cand * COEF_STRIDE;
    CompCoef c;
    c.thr = (float)cf[0];
    c.thr_inv = (float)cf[1];
    c.p = (float)cf[2];
    c.cat = (float)cf[3];
    c.crl = (float)cf[4];
    c.sg = c.cat <= c.crl ? 1.0f : -1.0f;
    return c;
}

template <int CTRL>
__device__ __forceinline__ float dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, 0xf, 0xf, false));
}

// ------------------------------------------------------------------------------------------------
// A. intercepts of every full block.  Workgroup = CB_TILE consecutive blocks of one stream: the samples are
// expanded to (a, r) pairs in LDS once (coalesced loads), then every 16-lane row of every wave walks one
// block: lane j keeps B_j; per sample one row_shr:1 (B_{j-1}), one packed FMA, one max.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_comp_blockfn(InView in, float *__restrict__ fn, CompGeom g, int C,
                                                       const double *__restrict__ coef) {
    __shared__ cb_f2 pairs[CB_TILE * CB_K];
    const int s = blockIdx.y, cand = s / C, ch = s % C;
    const float *x = in_ptr(in, cand, ch);
    const CompCoef cc = comp_coef(coef, cand);
    const float sa = cc.sg * (1.0f - cc.cat), sr = cc.sg * (1.0f - cc.crl);
    const cb_f2 c2 = {fmaxf(cc.cat, CB_CMIN), fmaxf(cc.crl, CB_CMIN)};
    const int64_t blk0 = (int64_t)blockIdx.x * CB_TILE, t0 = blk0 * CB_K;
    for (int i = threadIdx.x; i < CB_TILE * CB_K; i += 256) {
        const float v = t0 + i < g.L ? fabsf(x[t0 + i]) : 0.0f;
        pairs[i] = (cb_f2){sa * v, sr * v};
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, row = lane >> 4, j = lane & 15;
    const float ninf = -__builtin_inff();
    float *out = fn + (int64_t)s * g.fn_stride;
#pragma unroll 2
    for (int it = 0; it < CB_TILE / 16; ++it) {
        const int bl = w * (CB_TILE / 4) + it * 4 + row;
        const cb_f2 *pr = pairs + bl * CB_K;
        float B = j == 0 ? 0.0f : ninf;
#pragma unroll
        for (int n = 0; n < CB_K; ++n) {
            const float sh = dpp<0x111>(ninf, B);  // row_shr:1, lane 0 of the row keeps -inf
            const cb_f2 t = __builtin_elementwise_fma(c2, (cb_f2){sh, B}, pr[n]);
            B = fmaxf(t.x, t.y);
        }
        if (blk0 + bl < g.n_fn) out[(blk0 + bl) * 16 + j] = B;
    }
}

// ------------------------------------------------------------------------------------------------
// B. boundary states.  Four lanes per stream, four terms per lane; the 16-term max closes over the quad with two
// DPP quad_perm steps.  Intercepts come straight from HBM through a CB_D-deep register ring (one b128 per
// block per lane); states leave as one b128 per four blocks.  z_end[k] = state after block k.  The loop body
// is generated asm (comp_scan.inc, tools/gen/gen_comp_scan_asm.py): 9.6 instructions per block.
// ------------------------------------------------------------------------------------------------
// spw = streams per wave (a power of two <= 16): with fewer than sixteen, the upper quads shadow the lower ones (same values from and to
// the same addresses, like the idle quads past the last stream) and the launch has 16 / spw times as many waves.  The scan is one
// dependent chain per stream, but at sixteen streams a wave also pulls 16 x 2 MB of intercepts through ONE CU's memory pipe (35 GB/s
// per CU at 512 streams: 0.93 ms against a chain of ~0.72); spreading the streams over more CUs takes that off the chain.
__global__ __launch_bounds__(64) void k_comp_blockscan(const float *__restrict__ fn, float *__restrict__ z_end, CompGeom g,
                                                        int C, int S, const double *__restrict__ coef, int spw) {
    const int lane = threadIdx.x, q = lane & 3;
    const int s_raw = blockIdx.x * spw + ((lane >> 2) & (spw - 1));
    const int s = s_raw < S ? s_raw : S - 1;  // idle quads shadow the last stream (same values to the same addresses)
    const CompCoef cc = comp_coef(coef, s / C);
    const double ca = fmaxf(cc.cat, CB_CMIN), cr = fmaxf(cc.crl, CB_CMIN);
    float S4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int jj = 4 * q + u;
        S4[u] = (float)(pow(ca, (double)jj) * pow(cr, (double)(CB_K - jj)));
    }
    const cb_f2 S01 = {S4[0], S4[1]}, S23 = {S4[2], S4[3]};
    const cb_f4 *src = (const cb_f4 *)(fn + (int64_t)s * g.fn_stride) + q;  // block k at src[4 k]
    float *dst = z_end + (int64_t)s * g.z_stride;
    const int64_t n_ring = (g.n_fn + CB_D - 1) / CB_D;
    STITO_COMP_SCAN_PROLOGUE(src, dst + n_ring * CB_D);
    float z = 0.0f;
    for (int64_t r = 0; r < n_ring; ++r) {
        STITO_COMP_SCAN_PASS(z, S01, S23, src, dst);
        src += 4 * CB_D;
        dst += CB_D;
    }
    STITO_COMP_SCAN_DRAIN();
}

// ------------------------------------------------------------------------------------------------
// C. envelope + VCA.  gain = env < thr ? 1 : pow(env/thr, 1/ratio - 1); y = gain * x.  The power goes through
// v_log_f32 / v_exp_f32 (1 ulp each; base >= 1, |exponent| < 1: relative error < 2e-6 against powf, inside the
// 2e-5 parity bound of the effect).  One thread = one block: its 15 samples come out of an LDS tile that was
// loaded and is stored coalesced (stride 15 floats between lanes: conflict-free), so the kernel is bound by its
// 8 bytes per sample of HBM traffic.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float vca_gain(float e, float thr, float thr_inv, float p) {
    return e < thr ? 1.0f : __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(e * thr_inv));
}

__global__ __launch_bounds__(CA_TILE) void k_comp_apply(InView in, float *out, const float *__restrict__ z_end,
                                                         CompGeom g, int64_t cand_stride, int C,
                                                         const double *__restrict__ coef) {
    __shared__ float tile[CA_TILE * CB_K];
    const int s = blockIdx.y, cand = s / C, ch = s % C;
    const float *x = in_ptr(in, cand, ch);
    float *y = out + (int64_t)cand * cand_stride + (int64_t)ch * g.L;
    const CompCoef cc = comp_coef(coef, cand);
    const int64_t blk0 = (int64_t)blockIdx.x * CA_TILE, t0 = blk0 * CB_K;
    const int n_here = (int)((g.L - t0 < CA_TILE * CB_K) ? g.L - t0 : CA_TILE * CB_K);
    for (int i = threadIdx.x; i < n_here; i += CA_TILE) tile[i] = x[t0 + i];
    const int64_t k = blk0 + threadIdx.x;
    float env = (k > 0 && k < g.n_blocks) ? fabsf(z_end[(int64_t)s * g.z_stride + k - 1]) : 0.0f;  // env = |z|
    __syncthreads();
    float *mine = tile + threadIdx.x * CB_K;
    if (threadIdx.x * CB_K < n_here) {  // samples past L inside the last block: computed on stale LDS, never stored
#pragma unroll
        for (int n = 0; n < CB_K; ++n) {
            // inside a block: the reference's own operations in the reference's order, each rounded (no contraction)
            const float xv = mine[n], v = fabsf(xv);
            const float c = v > env ? cc.cat : cc.crl;
            env = __fadd_rn(v, __fmul_rn(c, __fsub_rn(env, v)));
            mine[n] = vca_gain(env, cc.thr, cc.thr_inv, cc.p) * xv;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_here; i += CA_TILE) y[t0 + i] = tile[i];
}

int compressor_stage(const InView &in, float *audio_dev, int64_t cand_stride, int pop, int C, int64_t n_samples,
                     const double *coef, void *workspace, hipStream_t st) {
    const CompGeom g = comp_geometry(n_samples);
    const int S = pop * C;
    float *fn = (float *)workspace;
    float *z_end = (float *)((char *)workspace + align_up((size_t)S * g.fn_stride * sizeof(float), 256));
    if (g.n_fn > 0) {
        hipLaunchKernelGGL(k_comp_blockfn, dim3((unsigned)((g.n_fn + CB_TILE - 1) / CB_TILE), S), dim3(256), 0, st, in, fn, g, C, coef);
        STITO_LAUNCH_CHECK();
        // streams per wave: eight (measured at 512 streams x 480 000 samples, the whole stage: 2.06 ms with sixteen, 1.85 with eight,
        // 1.86 with four or two, 1.93 with one), fewer while that leaves three quarters of the CUs without a wave
        DeviceInfo dinfo;
        STITO_TRY(device_info(dinfo));
        int spw = 8;
        while (spw > 1 && (S + spw - 1) / spw < dinfo.cus / 4) spw >>= 1;
        if (const char *e = getenv("STITO_COMP_SPW")) { const int v = atoi(e); if (v >= 1 && v <= 16 && (v & (v - 1)) == 0) spw = v; }  // tuning aid
        hipLaunchKernelGGL(k_comp_blockscan, dim3((S + spw - 1) / spw), dim3(64), 0, st, fn, z_end, g, C, S, coef, spw);
        STITO_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_comp_apply, dim3((unsigned)((g.n_blocks + CA_TILE - 1) / CA_TILE), S), dim3(CA_TILE), 0, st, in, audio_dev,
                       z_end, g, cand_stride, C, coef);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

}  // namespace stito
