// Micro-benchmark behind conv_wino23r.hip: ONE wave per SIMD (the kernel's waves use up to 512 registers), the weights operand
// of v_mfma_f32_32x32x16_f16 in AGPRs, three dependent products per accumulator, eight accumulators in rotation -- what can
// the wave issue between two products without slowing them down (32 cycles of matrix pipe each), by instruction type?
//   hipcc --offload-arch=gfx950 -O3 w23_shadow.hip -o w23_shadow && ./w23_shadow
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// TYPE: 0 v_add_f32, 1 v_fma_f32, 2 v_pk_add_f32, 3 v_pk_fma_f32, 4 v_fma_mixlo_f16 (independent destinations),
// 5 v_fma_mixlo_f16 + v_fma_mixhi_f16 pairs on one destination (the kernel's split), 6 ds_read_b128, 7 ds_write_b128,
// 8 v_max_f32, 9 global_load_lds_dwordx4 (1 KB, L2-resident source; K = copies per 8 products), 10 s_mov_b32 (SALU)
template <int TYPE>
__device__ __forceinline__ void filler(int i, float (&r)[16], f32x2 (&p)[8], unsigned (&hreg)[8], f32x4 (&q)[4], float sc, unsigned addr,
                                       const float *gsrc, unsigned ldsdst) {
    const int a = i & 7, b = (i + 3) & 7, c = (i + 5) & 7;
    if (TYPE == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[a]) : "v"(r[8 + b]), "v"(r[8 + c]));
    if (TYPE == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[a]) : "v"(r[8 + b]), "v"(r[8 + c]), "v"(r[8 + a]));
    if (TYPE == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[a & 3]) : "v"(p[4 + (b & 3)]), "v"(p[4 + (c & 3)]));
    if (TYPE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[a & 3]) : "v"(p[4 + (b & 3)]), "v"(p[4 + (c & 3)]), "v"(p[4 + (a & 3)]));
    if (TYPE == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hreg[a]) : "v"(r[8 + b]), "s"(sc));
    if (TYPE == 5) {
        if (i & 1) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hreg[(i >> 1) & 7]) : "v"(r[8 + b]), "s"(sc));
        else asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hreg[(i >> 1) & 7]) : "v"(r[8 + b]), "s"(sc));
    }
    if (TYPE == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(addr) : "memory");
    if (TYPE == 7) asm volatile("ds_write_b128 %1, %0" :: "v"(q[i & 3]), "v"(addr) : "memory");
    if (TYPE == 8) asm volatile("v_max_f32 %0, %1, %2" : "=v"(r[a]) : "v"(r[8 + b]), "v"(r[8 + c]));
    if (TYPE == 9) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(addr), "s"(gsrc), "s"(ldsdst) : "memory");
    if (TYPE == 10) { unsigned s_; asm volatile("s_mov_b32 %0, %1" : "=s"(s_) : "s"(ldsdst)); }
}

template <int K, int TYPE>
__global__ __launch_bounds__(256) void k(int nm, float *out, long long *res, const float *gsrc) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[8];
    h8 W[8], B;
    for (int i = 0; i < 8; ++i) {
        B[i] = (_Float16)0.5f;
        for (int j = 0; j < 8; ++j) W[j][i] = (_Float16)(1.0f + lane * 1e-3f + j);
    }
    float r[16];
    f32x2 p[8];
    unsigned hreg[8];
    f32x4 q[4];
    for (int i = 0; i < 16; ++i) r[i] = 1.0f + i + lane;
    for (int i = 0; i < 8; ++i) { p[i] = (f32x2){1.0f + i, 2.0f + lane}; hreg[i] = 0; }
    for (int i = 0; i < 4; ++i) q[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
    const unsigned addr = wv * 1024 + lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const float *gs = gsrc + (size_t)blockIdx.x * 4096 + wv * 256;
    const float sc = 4.0f;
    for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[j]) : "a"(W[j]), "v"(B));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nm / 24; ++it) {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(j / 3) & 7]) : "a"(W[(j / 3) & 7]), "v"(B));
            if (TYPE == 9) {
                if ((j & 7) < K) filler<TYPE>(j, r, p, hreg, q, sc, addr, gs, lds0 + 32768 + wv * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) filler<TYPE>(j * K + i, r, p, hreg, q, sc, addr, gs, lds0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TYPE == 6 || TYPE == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (TYPE == 9) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += r[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + (float)hreg[i];
    for (int i = 0; i < 4; ++i) s += q[i].x;
    for (int j = 0; j < 8; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0 && blockIdx.x == 7) { res[wv * 2] = t0; res[wv * 2 + 1] = t1; }
}

template <int K, int TYPE>
static void run(int nm, float *out, long long *res_d, const float *gsrc) {
    hipFuncSetAttribute((const void *)k<K, TYPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    long long res[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<K, TYPE>), dim3(256), dim3(256), 64 * 1024, 0, nm, out, res_d, gsrc);
        hipDeviceSynchronize();
    }
    hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
    long long lo = res[0], hi = res[1];
    for (int w = 0; w < 4; ++w) { lo = res[2 * w] < lo ? res[2 * w] : lo; hi = res[2 * w + 1] > hi ? res[2 * w + 1] : hi; }
    static const char *names[] = {"v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_fma_mixlo_f16", "mixlo+mixhi pairs", "ds_read_b128",
                                  "ds_write_b128", "v_max_f32", "LDS-DMA 1 KB / 8 MFMAs", "s_mov_b32"};
    printf("%-24s K=%d : %6.1f cyc per MFMA slot (32 = pipe-bound)\n", names[TYPE], K, (double)(hi - lo) / (double)(nm / 24 * 24));
}

template <int TYPE>
static void sweep(int nm, float *out, long long *res_d, const float *gsrc) {
    run<0, TYPE>(nm, out, res_d, gsrc);
    run<1, TYPE>(nm, out, res_d, gsrc);
    run<2, TYPE>(nm, out, res_d, gsrc);
    run<3, TYPE>(nm, out, res_d, gsrc);
    run<4, TYPE>(nm, out, res_d, gsrc);
    run<5, TYPE>(nm, out, res_d, gsrc);
    run<6, TYPE>(nm, out, res_d, gsrc);
    run<8, TYPE>(nm, out, res_d, gsrc);
}

int main() {
    float *out, *gsrc;
    long long *res_d;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&gsrc, 256 * 4096 * 4);
    hipMemset(gsrc, 0, 256 * 4096 * 4);
    hipMalloc(&res_d, 64);
    const int nm = 2400;
    sweep<0>(nm, out, res_d, gsrc);
    sweep<1>(nm, out, res_d, gsrc);
    sweep<2>(nm, out, res_d, gsrc);
    sweep<3>(nm, out, res_d, gsrc);
    sweep<4>(nm, out, res_d, gsrc);
    sweep<5>(nm, out, res_d, gsrc);
    sweep<8>(nm, out, res_d, gsrc);
    sweep<10>(nm, out, res_d, gsrc);
    sweep<6>(nm, out, res_d, gsrc);
    sweep<7>(nm, out, res_d, gsrc);
    sweep<9>(nm, out, res_d, gsrc);
    return 0;
}
