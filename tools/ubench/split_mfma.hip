// Micro-benchmarks behind the split-precision streaming convolution (conv_wino43.hip, k_conv_wino43s):
//   A. numerics: a K-deep 32x32 product from f32 operands on (1) the exact-f32 MFMA, (2) f16 MFMA with every operand split
//      in two halves hi + lo (11 + 11 significand bits, power-of-two pre-scale) and the products hi*hi + hi*lo + lo*hi,
//      (3) the same with lo*lo, (4) bf16 MFMA with three-way splits and six products -- all accumulated in f32 by the
//      matrix pipe, against a float64 reference;
//   B. issue rate of v_mfma_f32_32x32x16_f16 with one and two waves per SIMD;
//   C. LDS fill: a 512-thread workgroup streaming 48 KB slabs through a ring of three by LDS-DMA (two wave sets that
//      issue on alternate periods, so each wave waits vmcnt(0) before it issues again and a slab has two periods to land),
//      with and without the MFMA / ds_read work of the convolution, under three sharing patterns of the source streams.
//   hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// ---- A ---------------------------------------------------------------------------------------------------------
// A (32 x K) row-major, B (K x 32) row-major, D (32 x 32).  One wave.
template <int VAR>
__global__ void k_num(const float *A, const float *B, int K, float sa, float sb, float *D) {
    const int l = threadIdx.x, m = l & 31, kh = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (VAR == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * K + k + kh], B[(k + kh) * 32 + m], acc, 0, 0, 0);
    } else if (VAR == 1 || VAR == 2) {
        for (int k = 0; k < K; k += 16) {
            h8 ah, al, bh, bl;
            for (int i = 0; i < 8; ++i) {
                const float a = A[m * K + k + 8 * kh + i] * sa, b = B[(k + 8 * kh + i) * 32 + m] * sb;
                ah[i] = (_Float16)a; al[i] = (_Float16)(a - (float)ah[i]);
                bh[i] = (_Float16)b; bl[i] = (_Float16)(b - (float)bh[i]);
            }
            if (VAR == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            b8 a0, a1, a2, b0, b1, b2;
            for (int i = 0; i < 8; ++i) {
                float a = A[m * K + k + 8 * kh + i], b = B[(k + 8 * kh + i) * 32 + m];
                a0[i] = (__bf16)a; a -= (float)a0[i]; a1[i] = (__bf16)a; a -= (float)a1[i]; a2[i] = (__bf16)a;
                b0[i] = (__bf16)b; b -= (float)b0[i]; b1[i] = (__bf16)b; b -= (float)b1[i]; b2[i] = (__bf16)b;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
        }
    }
    const float inv = (VAR == 1 || VAR == 2) ? 1.0f / (sa * sb) : 1.0f;
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + m] = acc[r] * inv;
}

static float pow2_scale(const std::vector<float> &v, int target_log2) {
    float mx = 0.f;
    for (float x : v) mx = std::fmax(mx, std::fabs(x));
    if (mx == 0.f) return 1.f;
    int e;
    std::frexp(mx, &e);  // mx = f * 2^e, f in [0.5, 1)  =>  mx <= 2^e
    return std::ldexp(1.0f, target_log2 - e);
}

static void numerics(int K, int kind) {
    std::mt19937 rng(1234 + kind);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A(32 * K), B(K * 32);
    for (auto &x : A) {
        const float g = nd(rng);
        x = kind == 0 ? g : (kind == 1 ? std::fmax(g, 0.f) * std::exp(1.5f * nd(rng)) : g * std::exp(3.0f * nd(rng)));
    }
    for (auto &x : B) x = 0.02f * nd(rng) * (kind == 2 ? std::exp(2.0f * nd(rng)) : 1.f);
    std::vector<double> ref(1024, 0.0);
    for (int m = 0; m < 32; ++m)
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < 32; ++n) ref[m * 32 + n] += (double)A[m * K + k] * (double)B[k * 32 + n];
    double rmax = 0;
    for (double r : ref) rmax = std::fmax(rmax, std::fabs(r));
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const float sa = pow2_scale(A, 15), sb = pow2_scale(B, 14);
    static const char *names[] = {"f32 mfma", "f16x2, 3 products", "f16x2, 4 products", "bf16x3, 6 products"};
    static const char *kinds[] = {"signed normal", "relu x lognormal(1.5)", "heavy tails (lognormal 3 / 2)"};
    printf("K = %d, A %s  (scales 2^%d, 2^%d)\n", K, kinds[kind], (int)std::log2(sa), (int)std::log2(sb));
    for (int var = 0; var < 4; ++var) {
        if (var == 0) hipLaunchKernelGGL(k_num<0>, dim3(1), dim3(64), 0, 0, dA, dB, K, sa, sb, dD);
        if (var == 1) hipLaunchKernelGGL(k_num<1>, dim3(1), dim3(64), 0, 0, dA, dB, K, sa, sb, dD);
        if (var == 2) hipLaunchKernelGGL(k_num<2>, dim3(1), dim3(64), 0, 0, dA, dB, K, sa, sb, dD);
        if (var == 3) hipLaunchKernelGGL(k_num<3>, dim3(1), dim3(64), 0, 0, dA, dB, K, sa, sb, dD);
        std::vector<float> D(1024);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double emax = 0, e2 = 0;
        for (int i = 0; i < 1024; ++i) {
            const double e = std::fabs((double)D[i] - ref[i]);
            emax = std::fmax(emax, e);
            e2 += e * e;
        }
        printf("  %-20s max err / max|ref| = %.3e   rms err / max|ref| = %.3e\n", names[var], emax / rmax, std::sqrt(e2 / 1024) / rmax);
    }
    hipFree(dA); hipFree(dB); hipFree(dD);
}

// ---- B ---------------------------------------------------------------------------------------------------------
template <int MW>
__global__ __launch_bounds__(256 * MW) void k_rate(int nm, float *out, long long *res) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.0f + lane * 1e-3f); b[i] = (_Float16)0.5f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nm / 4; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && blockIdx.x == 7) { res[wv * 2] = t0; res[wv * 2 + 1] = t1; }
}

template <int MW>
static void rate(float *out, long long *res_d) {
    const int nm = 4096;
    long long res[16];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_rate<MW>, dim3(256), dim3(256 * MW), 0, 0, nm, out, res_d);
        hipDeviceSynchronize();
    }
    hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
    long long lo = res[0], hi = res[1];
    for (int w = 0; w < 4 * MW; ++w) { lo = res[2 * w] < lo ? res[2 * w] : lo; hi = res[2 * w + 1] > hi ? res[2 * w + 1] : hi; }
    printf("v_mfma_f32_32x32x16_f16, %d wave(s) per SIMD: %.1f cycles per MFMA per SIMD\n", MW, (double)(hi - lo) / ((double)nm * MW));
}

// ---- C ---------------------------------------------------------------------------------------------------------
// slab = SLAB bytes (one third transformed input, two thirds weights), NRING slabs in LDS, NRING - 1 wave sets: set k % NSETS
// issues slab k + NSETS in period k and waits vmcnt(0) at the end of period k + NSETS - 1, just before it issues again.
__device__ __forceinline__ void glds16(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
// WORK bit 0: ds_read_b128 of the operands, bit 1: MFMAs (3 per block, SLAB / 24 KB blocks per period and wave)
// pd > 0 (pattern 3 only): L2 prefetch by touch -- in the period in which it issues slab k + NSETS, a workgroup also touches (one
// dword per 128-byte line, result discarded) ITS SHARE of the lines of slab k + NSETS + pd: 1 / 8 of the input part (the 8
// workgroups of the XCD round that stream the same input) and 1 / 4 of the weight part, so that every line of the round's
// streams is requested from the fabric once, pd periods before the LDS-DMA copies want it.
template <int WORK, int SLAB, int NRING>
__global__ __launch_bounds__(512) void k_fill(const char *vsrc, const char *usrc, int pattern, int n_periods, float *out, long long *res, int pd) {
    constexpr int NSETS = NRING - 1, WPS = 8 / NSETS, PIECES = SLAB / 1024, PPW = PIECES / WPS, VPART = SLAB / 3, NBLK = SLAB / (24 * 1024);
    static_assert(PIECES % WPS == 0 && VPART % 1024 == 0, "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wv / WPS, ws = wv % WPS;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, r = j & 31, round = b >> 8;
    // pattern 0: every workgroup its own streams; 1: the convolution's XCD round (4 pixel blocks x 8 channel tiles per XCD,
    // all XCDs the same channel tiles), the same streams in every round (they stay in the Infinity Cache); 2: every workgroup
    // the same streams; 3: as 1 with fresh streams in every round of 256 workgroups (the real kernel)
    long long vb, ut;
    if (pattern == 0) { vb = b & 255; ut = b & 255; }
    else if (pattern == 1) { vb = xcd * 4 + (r >> 3); ut = r & 7; }
    else if (pattern == 3) { vb = (round * 8 + xcd) * 4 + (r >> 3); ut = round * 8 + (r & 7); }
    else { vb = 0; ut = 0; }
    const char *vbase = vsrc + vb * (long long)n_periods * VPART;
    const char *ubase = usrc + ut * (long long)n_periods * (SLAB - VPART);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    f32x16 acc[2];
    for (int q = 0; q < 2; ++q)
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
#define ISSUE(S, BUF)                                                                                     \
    _Pragma("unroll") for (int c = 0; c < PPW; ++c) {                                                      \
        const int piece = ws * PPW + c;                                                                    \
        const char *src = piece < VPART / 1024 ? vbase + (long long)(S) * VPART + piece * 1024             \
                                               : ubase + (long long)(S) * (SLAB - VPART) + (piece - VPART / 1024) * 1024; \
        glds16(src, (unsigned)lane * 16u, lds0 + (unsigned)((BUF) * SLAB + piece * 1024));                 \
    }
    unsigned touched = 0;
    constexpr int VL = VPART / 128 / 8, UL = (SLAB - VPART) / 128 / 4, TPW = (VL + UL + WPS - 1) / WPS;  // lines per workgroup, per wave
    static_assert(TPW <= 64, "");
    const int tl = ws * TPW + lane;  // this lane's line of the workgroup's share
    const bool t_on = pd > 0 && lane < TPW && tl < VL + UL;
    const long long t_off = tl < VL ? ((r & 7) * VL + tl) * 128ll : ((r >> 3) * UL + (tl - VL)) * 128ll;
    const char *t_base = tl < VL ? vbase : ubase;
    const long long t_stride = tl < VL ? VPART : SLAB - VPART;
#define TOUCH(S)                                                                                           \
    if (t_on && (S) < n_periods) {                                                                         \
        const char *p_ = t_base + (long long)(S) * t_stride + t_off;                                       \
        asm volatile("global_load_dword %0, %1, off" : "=v"(touched) : "v"(p_) : "memory");               \
    }
    for (int s0 = 0; s0 < NSETS; ++s0)
        if (set == s0) { ISSUE(s0, s0) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    int buf = 0, turn = 0;  // turn = k % NSETS
    for (int k = 0; k < n_periods; ++k) {
        const bool mine = turn == set;
        int nb = buf + NSETS; nb = nb >= NRING ? nb - NRING : nb;
        if (mine && k + NSETS < n_periods) { ISSUE(k + NSETS, nb) TOUCH(k + NSETS + pd) }
        if (WORK & 1) {
#pragma unroll
            for (int q = 0; q < NBLK; ++q) {
                const int slot = (wv >> 1) * NBLK + q;
                const char *vs = smem + buf * SLAB + slot * 2048, *us = smem + buf * SLAB + VPART + slot * 4096;
                const h8 ah = *(const h8 *)(vs + lane * 16), al = *(const h8 *)(vs + 1024 + lane * 16);
                const h8 bh = *(const h8 *)(us + ((lane >> 5) * 64 + (wv & 1) * 32 + (lane & 31)) * 16);
                const h8 bl = *(const h8 *)(us + 2048 + ((lane >> 5) * 64 + (wv & 1) * 32 + (lane & 31)) * 16);
                if (WORK & 2) {
                    acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[q & 1], 0, 0, 0);
                    acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[q & 1], 0, 0, 0);
                    acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[q & 1], 0, 0, 0);
                } else {
                    acc[q & 1][0] += (float)ah[0] + (float)al[1] + (float)bh[2] + (float)bl[3];
                }
            }
        }
        // the set that issues next period must have its previous slab landed
        int nt = turn + 1; nt = nt == NSETS ? 0 : nt;
        if (nt == set) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(touched));  // the touch's destination register stays reserved until it has landed
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        buf = buf == NRING - 1 ? 0 : buf + 1;
        turn = nt;
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(touched));
    float s = (float)(touched & 1);
    for (int q = 0; q < 2; ++q)
        for (int i = 0; i < 16; ++i) s += acc[q][i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) { res[2 * blockIdx.x] = t0; res[2 * blockIdx.x + 1] = t1; }
}

template <int WORK, int SLAB, int NRING>
static void fill(const char *vsrc, const char *usrc, int pattern, long long bytes_per_wg, int blocks, float *out, long long *res_d, int pd = 0) {
    const int n_periods = (int)(bytes_per_wg / SLAB);
    hipFuncSetAttribute((const void *)k_fill<WORK, SLAB, NRING>, hipFuncAttributeMaxDynamicSharedMemorySize, NRING * SLAB);
    std::vector<long long> res(2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_fill<WORK, SLAB, NRING>), dim3(blocks), dim3(512), NRING * SLAB, 0, vsrc, usrc, pattern, n_periods, out, res_d, pd);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    if (hipGetLastError() != hipSuccess) printf("launch failed\n");
    hipMemcpy(res.data(), res_d, res.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (int i = 0; i < blocks; ++i) {
        const double c = (double)(res[2 * i + 1] - res[2 * i]) / n_periods;
        sum += c;
        mx = c > mx ? c : mx;
    }
    static const char *pn[] = {"own streams", "XCD round 4 x 8 (same)", "one stream", "XCD round 4 x 8 (fresh)"};
    const double bytes = (double)blocks * n_periods * SLAB;
    if (pd) printf("[touch-prefetch %d periods ahead] ", pd);
    printf("fill work=%d slab %2d KB x %d %-24s: %7.1f cycles per period (max %7.1f) = %5.1f B/clk/CU;  %.3f ms, %.2f TB/s into LDS\n", WORK,
           SLAB / 1024, NRING, pn[pattern], sum / blocks, mx, SLAB / (sum / blocks), ms, bytes / ms / 1e9);
}

int main(int argc, char **argv) {
    const int what = argc > 1 ? atoi(argv[1]) : 7;
    if (what & 1) {
        for (int kind = 0; kind < 3; ++kind) numerics(2048, kind);
        numerics(18432, 0);
        numerics(18432, 1);
    }
    float *out;
    long long *res_d;
    hipMalloc(&out, 4096 * 512 * 4);
    hipMalloc(&res_d, 4096 * 16);
    if (what & 2) { rate<1>(out, res_d); rate<2>(out, res_d); }
    if (what & 4) {
        const long long bytes_per_wg = 288ll * 48 * 1024;  // Cin = 1024 through the 32 x 64 tile: 288 slabs of 48 KB
        const int blocks = 1024;
        char *vsrc, *usrc;
        const size_t vbytes = (size_t)256 * bytes_per_wg / 3, ubytes = (size_t)256 * bytes_per_wg * 2 / 3;
        hipMalloc(&vsrc, vbytes); hipMalloc(&usrc, ubytes);
        hipMemset(vsrc, 0, vbytes); hipMemset(usrc, 0, ubytes);
        for (int pattern : {2, 1, 3, 0}) {
            fill<0, 48 * 1024, 3>(vsrc, usrc, pattern, bytes_per_wg, blocks, out, res_d);
            fill<3, 48 * 1024, 3>(vsrc, usrc, pattern, bytes_per_wg, blocks, out, res_d);
            fill<3, 72 * 1024, 2>(vsrc, usrc, pattern, bytes_per_wg, blocks, out, res_d);
            fill<3, 24 * 1024, 3>(vsrc, usrc, pattern, bytes_per_wg, blocks, out, res_d);
            fill<3, 24 * 1024, 5>(vsrc, usrc, pattern, bytes_per_wg, blocks, out, res_d);
            fill<0, 24 * 1024, 5>(vsrc, usrc, pattern, bytes_per_wg, blocks, out, res_d);
        }
    }
    if (what & 8) {  // L2 prefetch by touch under the real kernel's sharing pattern
        const long long bytes_per_wg = 288ll * 48 * 1024;
        const int blocks = 1024;
        char *vsrc, *usrc;
        const size_t vbytes = (size_t)256 * bytes_per_wg / 3, ubytes = (size_t)256 * bytes_per_wg * 2 / 3;
        hipMalloc(&vsrc, vbytes); hipMalloc(&usrc, ubytes);
        hipMemset(vsrc, 0, vbytes); hipMemset(usrc, 0, ubytes);
        for (int pd : {0, 1, 2, 3, 4, 6, 8}) {
            fill<3, 48 * 1024, 3>(vsrc, usrc, 3, bytes_per_wg, blocks, out, res_d, pd);
            fill<0, 48 * 1024, 3>(vsrc, usrc, 3, bytes_per_wg, blocks, out, res_d, pd);
        }
    }
    return 0;
}
