// Main-loop prototype of a 128-tile x 128-channel streaming convolution tile (round 4, not a product kernel): what would a
// workgroup tile four times the two-sweep kernel's (k_conv_wino43s2: 64 x 64) cost per period?  Same machinery as
// split_mfma.hip part C -- 48 KB slabs through a ring of three by LDS-DMA, two wave sets issuing on alternate periods, operands by
// ds_read_b128 straight in MFMA layout, v_mfma_f32_32x32x16_f16 x 3 products -- but a slab is 3 Winograd positions x 16 input
// channels x (128 tiles | 128 couts) x (hi, lo), and wave (tile half, cout quarter) owns a 64 x 32 sub-tile: per position 2 A
// panels + 1 B panel (6 reads) for 6 products.  One position ROW (6 positions = 2 periods per 16 channels) per sweep, so the
// accumulators are 12 blocks = 192 registers; the six sweeps' partial outputs are not modelled (+ 13 % of the stream at 2 048
// input channels, + 27 % at 1 024).  786 K MACs per 48 KB against 393 K (two-sweep kernel) and 262 K (one-sweep kernel).
//   hipcc --offload-arch=gfx950 -O3 s3_loop.hip -o s3_loop && ./s3_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
constexpr int SLAB = 48 * 1024, PART = 24 * 1024, NRING = 3;

// WORK bit 0: operand reads, bit 1: products, bit 2: copies
// pattern 0: XCD x runs channel tiles 4 (x % 4) .. + 3 x pixel quads 8 (x / 4) .. + 7 (12 distinct streams per XCD; a weight stream
//            is wanted by 2 XCDs, an input stream by 4); 1: XCD x runs pixel quads 2 x, 2 x + 1 x all 16 channel tiles (18 distinct
//            streams; weight streams wanted by all 8 XCDs at the same time, input streams by one); 2: every workgroup its own streams
template <int WORK>
__global__ __launch_bounds__(512) void k_s3(const char *vsrc, const char *usrc, int pattern, int n_periods, float *out, long long *res) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, oct = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wv >> 2, ws = wv & 3, th = wv & 1, cq = wv >> 1;
    const int b = blockIdx.x, xcd = b & 7, r = (b >> 3) & 31;
    long long px, ct;
    if (pattern == 0) { ct = (xcd & 3) * 4 + (r & 3); px = (xcd >> 2) * 8 + (r >> 2); }
    else if (pattern == 1) { px = xcd * 2 + (r >> 4); ct = r & 15; }
    else { px = b & 255; ct = b & 255; }
    const char *vbase = vsrc + px * (long long)n_periods * PART;
    const char *ubase = usrc + ct * (long long)n_periods * PART;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    f32x16 acc[12];
    for (int q = 0; q < 12; ++q)
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    // 48 pieces of 1 KB per slab, 12 per wave of the issuing set: piece = 4 c + ws (c < 6: input part)
#define ISSUE(S, BUF)                                                                                     \
    if (WORK & 4) {                                                                                       \
        _Pragma("unroll") for (int c = 0; c < 12; ++c) {                                                   \
            const char *src = c < 6 ? vbase + (long long)(S) * PART + (4 * c + ws) * 1024                  \
                                    : ubase + (long long)(S) * PART + (4 * (c - 6) + ws) * 1024;           \
            glds16(src, (unsigned)lane * 16u, lds0 + (unsigned)((BUF) * SLAB + (4 * c + ws) * 1024));      \
        }                                                                                                  \
    }
    if (set == 0) { ISSUE(0, 0) } else { ISSUE(1, 1) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char *a_rd = smem + (oct * 128 + th * 64 + l31) * 16;          // + pos * 8192 + (lo: 4096) + blk * 512
    const char *b_rd = smem + PART + (oct * 128 + cq * 32 + l31) * 16;   // + pos * 8192 + (lo: 4096)
    const long long t0 = __builtin_readcyclecounter();
    int buf = 0;
    for (int k = 0; k < n_periods; k += 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bool mine = set == kk;
            int nb = buf + 2; nb = nb >= NRING ? nb - NRING : nb;
            if (mine && k + kk + 2 < n_periods) { ISSUE(k + kk + 2, nb) }
            if (WORK & 1) {
                const char *pa = a_rd + buf * SLAB, *pb = b_rd + buf * SLAB;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const h8 ah0 = *(const h8 *)(pa + q * 8192), al0 = *(const h8 *)(pa + q * 8192 + 4096);
                    const h8 ah1 = *(const h8 *)(pa + q * 8192 + 512), al1 = *(const h8 *)(pa + q * 8192 + 4096 + 512);
                    const h8 bh = *(const h8 *)(pb + q * 8192), bl = *(const h8 *)(pb + q * 8192 + 4096);
                    const int a0 = (kk * 3 + q) * 2;
                    if (WORK & 2) {
                        acc[a0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh, acc[a0], 0, 0, 0);
                        acc[a0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh, acc[a0 + 1], 0, 0, 0);
                        acc[a0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl, acc[a0], 0, 0, 0);
                        acc[a0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl, acc[a0 + 1], 0, 0, 0);
                        acc[a0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh, acc[a0], 0, 0, 0);
                        acc[a0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh, acc[a0 + 1], 0, 0, 0);
                    } else {
                        acc[a0][0] += (float)ah0[0] + (float)al0[1] + (float)bh[2] + (float)bl[3] + (float)ah1[4] + (float)al1[5];
                    }
                }
            }
            if (!mine) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the set that issues next period: its previous slab has landed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            buf = buf == NRING - 1 ? 0 : buf + 1;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int q = 0; q < 12; ++q)
        for (int i = 0; i < 16; ++i) s += acc[q][i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) { res[2 * blockIdx.x] = t0; res[2 * blockIdx.x + 1] = t1; }
}

template <int WORK>
static void run(const char *vsrc, const char *usrc, int pattern, int n_periods, float *out, long long *res_d) {
    const int blocks = 256;
    hipFuncSetAttribute((const void *)k_s3<WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, NRING * SLAB);
    std::vector<long long> res(2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_s3<WORK>), dim3(blocks), dim3(512), NRING * SLAB, 0, vsrc, usrc, pattern, n_periods, out, res_d);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    if (hipGetLastError() != hipSuccess) printf("launch failed\n");
    hipMemcpy(res.data(), res_d, res.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (int i = 0; i < blocks; ++i) {
        const double c = (double)(res[2 * i + 1] - res[2 * i]) / n_periods;
        sum += c; mx = c > mx ? c : mx;
    }
    static const char *pn[] = {"XCD: 4 ct x 8 px", "XCD: 16 ct x 2 px", "own streams"};
    const double cyc = sum / blocks, macs = 3.0 * 128 * 128 * 16;
    printf("work=%d %-18s: %7.1f cycles per period (max %7.1f) = %5.1f B/clk/CU, %5.2f cycles per K true MACs (two-sweep kernel: 1970 / 393 = 5.0); "
           "%.3f ms for %d periods, %.0f TFLOP/s issued\n", WORK, pn[pattern], cyc, mx, SLAB / cyc, cyc / (macs / 1e3), ms, n_periods,
           256.0 * n_periods * macs * 2 * 3 / ms / 1e9);
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2048;
    const int n_periods = 6 * (K / 16) * 2;                 // six sweeps x K / 16 channel groups x two periods per row
    const size_t stream = (size_t)n_periods * PART;          // bytes of one input (or weight) stream: 36 positions x K x 128 x 4 B
    char *v, *u; float *out; long long *res;
    hipMalloc(&v, 256 * stream); hipMalloc(&u, 256 * stream);   // (pattern 2 needs 256 streams of each; the convolution has 16)
    hipMemset(v, 0, 256 * stream); hipMemset(u, 0, 256 * stream);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&res, 512 * 8);
    printf("K = %d input channels: %d periods per workgroup, %.1f MB per stream, 256 workgroups (16 pixel quads x 16 channel tiles)\n", K, n_periods, stream / 1e6);
    for (int p = 0; p < 2; ++p) {
        run<7>(v, u, p, n_periods, out, res);
        run<4>(v, u, p, n_periods, out, res);
    }
    run<3>(v, u, 0, n_periods, out, res);
    run<1>(v, u, 0, n_periods, out, res);
    run<7>(v, u, 2, n_periods, out, res);
    return 0;
}
