// Micro-benchmark: what can a third wave of a SIMD issue while the SIMD's two other waves stream
// v_mfma_f32_32x32x2_f32 back to back (the regime of k_conv_wino: 8 consumer + 4 producer waves)?
// 768-thread workgroups, one per CU: waves 0-7 issue NM MFMAs each on 8 independent accumulators,
// waves 8-11 loop over a filler body of one instruction kind until wave 0 raises an LDS flag.
// Output per filler kind: cycles of the MFMA waves (vs the filler-free run) and the filler wave's
// cycles per filler instruction.
//   hipcc --offload-arch=gfx950 -O3 mfma_coissue.hip -o /tmp/mfma_coissue && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND>
__device__ __forceinline__ void filler_body(float *lds, int lane, f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {
    if (KIND == 1) {  // 16 v_fma_f32, 4 independent chains of 4
        REP4(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                          : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) : "v"(b.x), "v"(b.y));)
    } else if (KIND == 2) {  // 16 v_pk_fma_f32
        f32x2 p0 = {a.x, a.y}, p1 = {a.z, a.w}, p2 = {c.x, c.y}, p3 = {c.z, c.w}, q = {b.x, b.y}, r = {b.z, b.w};
        REP4(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(r));)
        a.x = p0.x + p1.x; c.x = p2.x + p3.x;
    } else if (KIND == 3) {  // 16 v_add_f32
        REP4(asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4"
                          : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) : "v"(b.x));)
    } else if (KIND == 4) {  // 16 v_pk_add_f32
        f32x2 p0 = {a.x, a.y}, p1 = {a.z, a.w}, p2 = {c.x, c.y}, p3 = {c.z, c.w}, q = {b.x, b.y};
        REP4(asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        a.x = p0.x + p1.x; c.x = p2.x + p3.x;
    } else if (KIND == 5) {  // 16 ds_read_b128 (conflict-free, lane * 16 B), one wait
        const unsigned addr = 65536 + lane * 16;
        REP4(asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                          : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");)
    } else if (KIND == 6) {  // 16 ds_write_b128
        const unsigned addr = 65536 + lane * 16;
        REP4(asm volatile("ds_write_b128 %4, %0\n\tds_write_b128 %4, %1 offset:1024\n\tds_write_b128 %4, %2 offset:2048\n\tds_write_b128 %4, %3 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                          :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(addr) : "memory");)
    } else if (KIND == 7) {  // 16 v_mov_b32
        REP4(asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4"
                          : "=v"(a.x), "=v"(a.y), "=v"(a.z), "=v"(a.w) : "v"(b.x));)
    } else if (KIND == 8) {  // 16 s_nop (SALU-side only)
        REP16(asm volatile("s_nop 0");)
    } else if (KIND == 9) {  // transform-like group: 4 ds_read_b128, 8 v_add_f32, 4 v_sub_f32... x1 + 2 ds_write_b128 => 16 "useful" VALU
        const unsigned addr = 65536 + lane * 16;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
        asm volatile("v_sub_f32 %0, %0, %8\n\tv_sub_f32 %1, %1, %9\n\tv_sub_f32 %2, %2, %10\n\tv_sub_f32 %3, %3, %11\n\t"
                     "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %9\n\tv_add_f32 %6, %6, %10\n\tv_add_f32 %7, %7, %11\n\t"
                     "v_sub_f32 %8, %8, %4\n\tv_sub_f32 %9, %9, %5\n\tv_sub_f32 %10, %10, %6\n\tv_sub_f32 %11, %11, %7\n\t"
                     "v_sub_f32 %12, %12, %4\n\tv_sub_f32 %13, %13, %5\n\tv_sub_f32 %14, %14, %6\n\tv_sub_f32 %15, %15, %7"
                     : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w),
                       "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
        asm volatile("ds_write_b128 %4, %0 offset:4096\n\tds_write_b128 %4, %1 offset:5120\n\tds_write_b128 %4, %2 offset:6144\n\tds_write_b128 %4, %3 offset:7168"
                     :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(addr) : "memory");
    } else if (KIND == 10) {  // 16 v_fma_f32 at priority 3
        REP4(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                          : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) : "v"(b.x), "v"(b.y));)
    }
}

// MFMA16: 0 = v_mfma_f32_32x32x2_f32 (64 cyc/SIMD), 1 = v_mfma_f32_16x16x4_f32 (32 cyc/SIMD)
// FIRST: the filler waves are waves 0-3 (the oldest of the workgroup) instead of the youngest.
// MW: MFMA waves per SIMD (1 or 2).  res layout (long long): [wave][0] start, [1] end, [2] filler groups,
// [3..3+NS) stamp of filler group g (absolute s_memtime).
static constexpr int NS = 48, RW = 3 + NS;
template <int KIND, int MFMA16, bool FIRST, int MW>
__global__ __launch_bounds__(64 * (4 * MW + 4)) void k(int nm, float *out, long long *res) {
    extern __shared__ float lds[];
    constexpr int NWAVES = 4 * MW + 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    volatile int *flag = (volatile int *)lds;
    if (tid == 0) *flag = 0;
    for (int i = tid + 64; i < 20480; i += 64 * NWAVES) lds[i] = (float)i;
    __syncthreads();
    const bool is_filler = FIRST ? wv < 4 : wv >= 4 * MW;
    long long *my = res + wv * RW;
    const bool rec = lane == 0 && blockIdx.x == 7;
    if (!is_filler) {
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float a = 1.0f + lane * 1e-3f, b = 0.5f;
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < nm / 32; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (MFMA16) {
                        f32x4 *p = (f32x4 *)&acc[j];
                        p[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, p[q], 0, 0, 0);
                    } else {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                    }
                }
        }
        const long long t1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) atomicAdd((int *)lds, 1);  // flag counts finished MFMA waves
        float s = 0.f;
        for (int j = 0; j < 8; ++j)
            for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[blockIdx.x * 1024 + tid] = s;
        if (rec) { my[0] = t0; my[1] = t1; }
    } else {
        if (KIND == 0) return;
        if (KIND == 10) __builtin_amdgcn_s_setprio(3);
        f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {1e-3f, 1.f, 1e-3f, 1.f}, c = a, d = b;
        long long n_it = 0;
        const long long t0 = __builtin_readcyclecounter();
        while (true) {
            filler_body<KIND>(lds, lane, a, b, c, d);
            if (n_it < NS && rec) my[3 + n_it] = __builtin_readcyclecounter();
            n_it += 1;
            if (*flag >= 4 * MW) break;
        }
        const long long t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 1024 + tid] = a.x + b.x + c.x + d.x + a.y + a.z + a.w;
        if (rec) { my[0] = t0; my[1] = t1; my[2] = n_it; }
    }
}

template <int KIND, int MFMA16, bool FIRST, int MW>
static void run(const char *name, int nm, float *out, long long *res_d) {
    constexpr int NWAVES = 4 * MW + 4;
    hipFuncSetAttribute((const void *)k<KIND, MFMA16, FIRST, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    std::vector<long long> res(12 * RW);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(res_d, 0, res.size() * 8);
        hipLaunchKernelGGL((k<KIND, MFMA16, FIRST, MW>), dim3(256), dim3(64 * NWAVES), 128 * 1024, 0, nm, out, res_d);
        hipDeviceSynchronize();
    }
    hipMemcpy(res.data(), res_d, res.size() * 8, hipMemcpyDeviceToHost);
    long long base = -1;
    for (int w = 0; w < NWAVES; ++w)
        if (res[w * RW] && (base < 0 || res[w * RW] < base)) base = res[w * RW];
    const double ideal = (double)nm * (MFMA16 ? 32 : 64) * MW;
    printf("== %-30s mfma%s x%d/SIMD, fillers %s: ideal %.0f cyc\n", name, MFMA16 ? "16x16x4" : "32x32x2", MW, FIRST ? "oldest" : "youngest", ideal);
    for (int w = 0; w < NWAVES; ++w) {
        const bool is_filler = FIRST ? w < 4 : w >= 4 * MW;
        const long long *my = &res[w * RW];
        if (!is_filler) {
            printf("   wave %2d mfma   start %8lld end %8lld  (%.3f of ideal)\n", w, my[0] - base, my[1] - base, (my[1] - my[0]) / ideal);
        } else if (KIND && (w == (FIRST ? 0 : 4 * MW))) {
            printf("   wave %2d filler start %8lld end %8lld groups %lld -> %.1f cyc/instr; group stamps:", w, my[0] - base, my[1] - base, my[2],
                   (double)(my[1] - my[0]) / (my[2] * 16.0));
            for (int g = 0; g < NS && g < my[2]; g += (g < 8 ? 1 : 8)) printf(" %lld", my[3 + g] - base);
            printf("\n");
        }
    }
}


// k2: fixed-count filler (no polling): MEMOP 0 none, 1 s_memtime per group, 2 LDS read per group, 3 global store per group,
// 4 LDS write per group.  Reports when the filler finished its NG groups of 16 v_fma_f32.
template <int MEMOP, bool FIRST>
__global__ __launch_bounds__(768) void k2(int nm, int ng, float *out, long long *res) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 20480; i += 768) lds[i] = (float)i;
    __syncthreads();
    const bool is_filler = FIRST ? wv < 4 : wv >= 8;
    long long *my = res + wv * RW;
    const bool rec = lane == 0 && blockIdx.x == 7;
    if (!is_filler) {
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float a = 1.0f + lane * 1e-3f, b = 0.5f;
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < nm / 32; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
        const long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int j = 0; j < 8; ++j)
            for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[blockIdx.x * 1024 + tid] = s;
        if (rec) { my[0] = t0; my[1] = t1; }
    } else {
        f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {1e-3f, 1.f, 1e-3f, 1.f}, c = a, d = b;
        const long long t0 = __builtin_readcyclecounter();
        long long tm = 0;
        float acc2 = 0.f;
        for (int g = 0; g < ng; ++g) {
            filler_body<1>(lds, lane, a, b, c, d);
            if (MEMOP == 1) tm += __builtin_readcyclecounter();
            if (MEMOP == 2) acc2 += ((volatile float *)lds)[1024 + lane];
            if (MEMOP == 3) out[256 * 1024 + blockIdx.x * 1024 + tid] = a.x;
            if (MEMOP == 4) ((volatile float *)lds)[2048 + tid] = a.x;
        }
        const long long t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 1024 + tid] = a.x + b.x + c.x + d.x + a.y + a.z + a.w + acc2 + (float)tm;
        if (rec) { my[0] = t0; my[1] = t1; my[2] = ng; }
    }
}

template <int MEMOP, bool FIRST>
static void run2(const char *name, int nm, int ng, float *out, long long *res_d) {
    hipFuncSetAttribute((const void *)k2<MEMOP, FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    std::vector<long long> res(12 * RW);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(res_d, 0, res.size() * 8);
        hipLaunchKernelGGL((k2<MEMOP, FIRST>), dim3(256), dim3(768), 128 * 1024, 0, nm, ng, out, res_d);
        hipDeviceSynchronize();
    }
    hipMemcpy(res.data(), res_d, res.size() * 8, hipMemcpyDeviceToHost);
    long long base = -1;
    for (int w = 0; w < 12; ++w)
        if (res[w * RW] && (base < 0 || res[w * RW] < base)) base = res[w * RW];
    const int fw = FIRST ? 0 : 8, m0 = FIRST ? 4 : 0, m1 = FIRST ? 8 : 4;
    printf("== k2 %-22s fillers %-8s: mfma wave A end %8lld, wave B end %8lld (ideal %d / %d); filler %d groups: start %lld end %lld = %.1f cyc/instr\n", name,
           FIRST ? "oldest" : "youngest", res[m0 * RW + 1] - base, res[m1 * RW + 1] - base, nm * 64, nm * 128, ng, res[fw * RW] - base,
           res[fw * RW + 1] - base, (double)(res[fw * RW + 1] - res[fw * RW]) / (ng * 16.0));
}

int main() {
    float *out;
    long long *res_d;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&res_d, 12 * RW * 8);
    const int nm = 32 * 128;
    hipMalloc(&out, 2 * 256 * 1024 * 4);

    run2<0, false>("valu only", nm, 256, out, res_d);
    run2<0, true>("valu only", nm, 256, out, res_d);
    run2<1, false>("valu + s_memtime", nm, 256, out, res_d);
    run2<1, true>("valu + s_memtime", nm, 256, out, res_d);
    run2<2, false>("valu + lds read", nm, 256, out, res_d);
    run2<2, true>("valu + lds read", nm, 256, out, res_d);
    run2<3, false>("valu + global store", nm, 256, out, res_d);
    run2<3, true>("valu + global store", nm, 256, out, res_d);
    run2<4, false>("valu + lds write", nm, 256, out, res_d);
    run2<4, true>("valu + lds write", nm, 256, out, res_d);
    run<0, 0, false, 2>("no filler", nm, out, res_d);
    run<1, 0, false, 2>("v_fma_f32", nm, out, res_d);
    run<8, 0, false, 2>("s_nop", nm, out, res_d);
    run<10, 0, false, 2>("v_fma_f32 @prio3", nm, out, res_d);
    run<1, 0, true, 2>("v_fma_f32", nm, out, res_d);
    run<2, 0, true, 2>("v_pk_fma_f32", nm, out, res_d);
    run<5, 0, true, 2>("ds_read_b128", nm, out, res_d);
    run<6, 0, true, 2>("ds_write_b128", nm, out, res_d);
    run<9, 0, true, 2>("transform-like", nm, out, res_d);
    run<0, 0, false, 1>("no filler", nm, out, res_d);
    run<1, 0, false, 1>("v_fma_f32", nm, out, res_d);
    run<1, 0, true, 1>("v_fma_f32", nm, out, res_d);
    run<9, 0, true, 1>("transform-like", nm, out, res_d);
    run<9, 0, false, 1>("transform-like", nm, out, res_d);
    run<1, 1, false, 2>("v_fma_f32", 2 * nm, out, res_d);
    run<1, 1, true, 2>("v_fma_f32", 2 * nm, out, res_d);
    return 0;
}
