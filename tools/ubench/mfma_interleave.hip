// Micro-benchmark: how many VALU / LDS instructions of its OWN stream can a wave issue between two
// back-to-back v_mfma_f32_32x32x2_f32 (64 cycles of matrix pipe each) without slowing the MFMAs down,
// with one or two such waves per SIMD?  (mfma_coissue.hip showed that a wave streaming MFMAs starves
// every OTHER wave of its SIMD; the question here is what the wave itself may interleave.)
//   hipcc --offload-arch=gfx950 -O3 mfma_interleave.hip -o mfma_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// TYPE 0: v_fma_f32   1: v_pk_fma_f32   2: ds_read_b128   3: ds_write_b128   4: v_add_f32 (different regs)
// 5: mix: per 4 slots = 1 ds_read_b128 + 2 v_add_f32 + 1 ds_write_b128
template <int TYPE>
__device__ __forceinline__ void one(int i, f32x4 &a, f32x4 &b, f32x2 &p, f32x2 &q, unsigned addr) {
    if (TYPE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(b.x), "v"(b.y));
    if (TYPE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
    if (TYPE == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(addr) : "memory");
    if (TYPE == 3) asm volatile("ds_write_b128 %1, %0" :: "v"(a), "v"(addr) : "memory");
    if (TYPE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.y) : "v"(b.z));
    if (TYPE == 5) {
        if ((i & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(addr) : "memory");
        if ((i & 3) == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.y) : "v"(a.z));
        if ((i & 3) == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a.w) : "v"(a.z));
        if ((i & 3) == 3) asm volatile("ds_write_b128 %1, %0 offset:8192" :: "v"(a), "v"(addr) : "memory");
    }
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// F16 = true: the same experiment around v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles of matrix pipe): what may a wave issue in
// the shadow of the split-precision kernels' MFMAs?
template <int K, int TYPE, int MW, bool F16 = false>
__global__ __launch_bounds__(256 * MW) void k(int nm, float *out, long long *res) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 16384; i += 256 * MW) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float av = 1.0f + lane * 1e-3f, bv = 0.5f;
    h8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(1.0f + lane * 1e-3f); bh[i] = (_Float16)0.5f; }
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {1e-3f, 1.f, 1e-3f, 1.f};
    f32x2 p = {1.f, 2.f}, q = {1e-3f, 1.f};
    const unsigned addr = wv * 1024 + lane * 16;
    const long long r0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nm / 8; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(ah), "v"(bh));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(av), "v"(bv));
#pragma unroll
            for (int i = 0; i < K; ++i) one<TYPE>(j * K + i, a, b, p, q, addr);
        }
        if (TYPE == 2 || TYPE == 3 || TYPE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + p.x + p.y;
    for (int j = 0; j < 8; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && blockIdx.x == 7) { res[wv * 2] = t0; res[wv * 2 + 1] = t1; if (wv == 0) { res[16] = r0; res[17] = r1; } }
}

template <int K, int TYPE, int MW, bool F16 = false>
static void run(int nm, float *out, long long *res_d) {
    hipFuncSetAttribute((const void *)k<K, TYPE, MW, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    long long res[18];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<K, TYPE, MW, F16>), dim3(256), dim3(256 * MW), 64 * 1024, 0, nm, out, res_d);
        hipDeviceSynchronize();
    }
    hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
    long long lo = res[0], hi = res[1];
    for (int w = 0; w < 4 * MW; ++w) { lo = res[2 * w] < lo ? res[2 * w] : lo; hi = res[2 * w + 1] > hi ? res[2 * w + 1] : hi; }
    static const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "ds_read_b128", "ds_write_b128", "v_add_f32", "mix rd/add/sub/wr"};
    printf("%-18s K=%2d waves/SIMD=%d : %7.1f cyc per MFMA slot (%s)\n", names[TYPE], K, MW, (double)(hi - lo) / ((double)nm * MW),
           F16 ? "f16 32x32x16: 32 = pipe-bound" : "64 = pipe-bound");
}
template <int TYPE, int MW>
static void sweep16(int nm, float *out, long long *res_d) {
    run<0, TYPE, MW, true>(nm, out, res_d);
    run<1, TYPE, MW, true>(nm, out, res_d);
    run<2, TYPE, MW, true>(nm, out, res_d);
    run<3, TYPE, MW, true>(nm, out, res_d);
    run<4, TYPE, MW, true>(nm, out, res_d);
    run<6, TYPE, MW, true>(nm, out, res_d);
    run<8, TYPE, MW, true>(nm, out, res_d);
}

template <int TYPE, int MW>
static void sweep(int nm, float *out, long long *res_d) {
    run<0, TYPE, MW>(nm, out, res_d);
    run<2, TYPE, MW>(nm, out, res_d);
    run<4, TYPE, MW>(nm, out, res_d);
    run<6, TYPE, MW>(nm, out, res_d);
    run<8, TYPE, MW>(nm, out, res_d);
    run<12, TYPE, MW>(nm, out, res_d);
    run<16, TYPE, MW>(nm, out, res_d);
}

// FED: the same density of ds_read_b128 beside v_mfma_f32_32x32x16_f16, but the reads FEED the products as in the streaming
// convolutions: MFMA j multiplies the operand pair that was read D slots earlier (R reads per slot into a ring of D + 1 pairs;
// R = 2: a fresh A and a fresh B per MFMA, R = 1: a fresh A, B stays), s_waitcnt lgkmcnt(R (D - 1)) in front of every MFMA.
// Question (profiles/README.md, "What bounds the streaming convolutions' main loop"): are reads that feed MFMAs dearer than
// independent ones?
template <int D, int R, int MW, int DEP = 1, int PAT = 0>
__global__ __launch_bounds__(256 * MW) void kfed(int nm, float *out, long long *res) {
    // DEP: consecutive MFMAs that accumulate into the same registers (3 in the streaming convolutions: hi hi' + hi lo' + lo hi');
    // PAT 1: the kernels' operand address pattern (lanes 0..31 512 contiguous bytes, lanes 32..63 1 KB behind them, the waves of
    // a workgroup spread over a 48 KB slab) instead of lane * 16
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 16384; i += 256 * MW) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    constexpr int NB = 12, SLOTS = 12;  // ring of operand pairs = slots per loop trip (compile-time indices)
    h8 A[NB], B[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) { A[b][i] = (_Float16)(1.0f + lane * 1e-3f); B[b][i] = (_Float16)0.5f; }
    const unsigned addr = PAT == 0 ? wv * 2048 + lane * 16
                                   : (wv & 1) * 4096 + ((lane >> 5) * 64 + ((wv >> 2) & 1) * 32 + (lane & 31)) * 16 + ((wv >> 1) & 1) * 24576;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(A[j % NB]) : "v"(addr) : "memory");
        if (R == 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(B[j % NB]) : "v"(addr) : "memory");
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nm / SLOTS; ++it) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            if (R * (D - 1) == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (R * (D - 1) == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            else if (R * (D - 1) == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else if (R * (D - 1) == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else if (R * (D - 1) == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (R * (D - 1) == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else if (R * (D - 1) == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(j / DEP) & 3]) : "v"(A[j % NB]), "v"(B[j % NB]));
            asm volatile("ds_read_b128 %0, %1" : "=v"(A[(j + D) % NB]) : "v"(addr) : "memory");
            if (R == 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(B[(j + D) % NB]) : "v"(addr) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) s += (float)A[b][0] + (float)B[b][1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && blockIdx.x == 7) { res[wv * 2] = t0; res[wv * 2 + 1] = t1; }
}

template <int D, int R, int MW, int DEP = 1, int PAT = 0>
static void run_fed(int nm, float *out, long long *res_d) {
    nm = nm / 12 * 12;
    hipFuncSetAttribute((const void *)kfed<D, R, MW, DEP, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    long long res[16];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((kfed<D, R, MW, DEP, PAT>), dim3(256), dim3(256 * MW), 64 * 1024, 0, nm, out, res_d);
        hipDeviceSynchronize();
    }
    hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
    long long lo = res[0], hi = res[1];
    for (int w = 0; w < 4 * MW; ++w) { lo = res[2 * w] < lo ? res[2 * w] : lo; hi = res[2 * w + 1] > hi ? res[2 * w + 1] : hi; }
    printf("reads FEED the MFMAs: %d ds_read_b128 per MFMA, used %d slots later, %d MFMA(s) per accumulator in a row, %s, waves/SIMD=%d : %6.1f cyc per MFMA slot (32 = pipe-bound)\n",
           R, D, DEP, PAT ? "kernel addresses" : "lane * 16", MW, (double)(hi - lo) / ((double)nm * MW));
}

// PHASE: the streaming convolutions' period as it is compiled (S43_ILV), in asm: 9 MFMAs in three dependent triples per wave and
// barrier, 12 operand reads in the first six gaps, the first triple fed from the phase before.  BAR = 0: without the s_barrier.
template <int MW, int BAR, int DATA = 0>
__global__ __launch_bounds__(256 * MW) void kphase(int nphase, float *out, long long *res) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // DATA: the f16 values the reads fetch -- 0 ordinary, 1 all subnormal (0x0123), 2 NaN (0x7e00), 3 zeros, 4 = hi normal / lo subnormal mix
    for (int i = tid; i < 16384; i += 256 * MW) {
        if (DATA == 0) lds[i] = 1e-3f * (float)(i & 255);
        else ((unsigned *)lds)[i] = DATA == 1 ? 0x01230123u : DATA == 2 ? 0x7e007e00u : DATA == 3 ? 0u : ((i >> 9) & 1 ? 0x00410023u : 0x3a833c00u);
    }
    __syncthreads();
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h8 ah0, al0, bh0, bl0, ah1, al1, bh1, bl1, ahP, alP, bhP, blP;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ahP[i] = alP[i] = (_Float16)(1.0f + lane * 1e-3f); bhP[i] = blP[i] = (_Float16)0.5f; }
    const unsigned addr = (wv & 1) * 4096 + ((lane >> 5) * 64 + ((wv >> 2) & 1) * 32 + (lane & 31)) * 16 + ((wv >> 1) & 1) * 24576;
#define PH_RD(X, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(X) : "v"(addr) : "memory");
#define PH_MM(Q, A_, B_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[Q]) : "v"(A_), "v"(B_));
#define PH_WAIT(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");
    const long long r0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nphase; ++it) {
        PH_MM(2, alP, bhP) PH_RD(ah0, 0) PH_RD(al0, 2048)
        PH_MM(2, ahP, blP) PH_RD(bh0, 512) PH_RD(bl0, 2560)
        PH_MM(2, ahP, bhP) PH_RD(ah1, 8192) PH_RD(al1, 10240)
        PH_WAIT(3) PH_MM(0, al0, bh0) PH_RD(bh1, 8704) PH_RD(bl1, 10752)
        PH_WAIT(4) PH_MM(0, ah0, bl0) PH_RD(ahP, 16384) PH_RD(alP, 18432)
        PH_MM(0, ah0, bh0) PH_RD(bhP, 16896) PH_RD(blP, 18944)
        PH_WAIT(5) PH_MM(1, al1, bh1)
        PH_WAIT(4) PH_MM(1, ah1, bl1)
        PH_MM(1, ah1, bh1)
        if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)ah0[0] + (float)al0[0] + (float)bh0[0] + (float)bl0[0] + (float)ah1[0] + (float)al1[0] + (float)bh1[0] + (float)bl1[0];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && blockIdx.x == 7) { res[wv * 2] = t0; res[wv * 2 + 1] = t1; if (wv == 0) { res[16] = r0; res[17] = r1; } }
}

template <int MW, int BAR, int DATA = 0>
static void run_phase(int nphase, float *out, long long *res_d) {
    hipFuncSetAttribute((const void *)kphase<MW, BAR, DATA>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    long long res[18];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((kphase<MW, BAR, DATA>), dim3(256), dim3(256 * MW), 64 * 1024, 0, nphase, out, res_d);
        hipDeviceSynchronize();
    }
    hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
    long long lo = res[0], hi = res[1];
    for (int w = 0; w < 4 * MW; ++w) { lo = res[2 * w] < lo ? res[2 * w] : lo; hi = res[2 * w + 1] > hi ? res[2 * w + 1] : hi; }
    static const char *dn[] = {"ordinary values", "all f16 subnormal", "NaN", "zeros", "normal / subnormal mix"};
    const double mhz = MW == 1 ? 0.0 : (double)(res[1] - res[0]) / ((double)(res[17] - res[16]) / 100.0);
    printf("[shader clock %.0f MHz: s_memtime against s_memrealtime] ", mhz);
    printf("the kernels' period in asm (9 MFMAs in dependent triples, 12 fed reads, %s, %s), waves/SIMD=%d : %7.1f cycles per period (%d MFMAs per SIMD: %d = pipe-bound)\n",
           BAR ? "s_barrier" : "no barrier", dn[DATA], MW, (double)(hi - lo) / nphase, 9 * MW, 9 * MW * 32);
}

template <int R, int MW>
static void sweep_fed(int nm, float *out, long long *res_d) {
    run_fed<1, R, MW>(nm, out, res_d);
    run_fed<2, R, MW>(nm, out, res_d);
    run_fed<3, R, MW>(nm, out, res_d);
    run_fed<4, R, MW>(nm, out, res_d);
    run_fed<6, R, MW>(nm, out, res_d);
}

int main(int argc, char **argv) {
    float *out;
    long long *res_d;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&res_d, 32 * 8);
    const int nm = 2048;
    if (argc > 1 && argv[1][0] == 'c') {  // what clock does the part hold under a sustained stream of MFMAs?  (s_memtime = shader clock,
        // s_memrealtime = constant 100 MHz; every kernel runs on all 256 CUs, two waves per SIMD, for the given number of MFMAs per wave)
        for (int nmc : {4096, 65536, 1048576}) {
            long long res[18];
            hipLaunchKernelGGL((k<0, 0, 2, false>), dim3(256), dim3(512), 64 * 1024, 0, nmc, out, res_d);
            hipDeviceSynchronize();
            hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
            printf("v_mfma_f32_32x32x2_f32  alone, %8d per wave: %6.1f cycles each, %7.3f ms, shader clock %4.0f MHz\n", nmc,
                   (double)(res[1] - res[0]) / (2.0 * nmc), (double)(res[17] - res[16]) / 1e5, (double)(res[1] - res[0]) / ((double)(res[17] - res[16]) / 100.0));
            hipLaunchKernelGGL((k<0, 0, 2, true>), dim3(256), dim3(512), 64 * 1024, 0, nmc, out, res_d);
            hipDeviceSynchronize();
            hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
            printf("v_mfma_f32_32x32x16_f16 alone, %8d per wave: %6.1f cycles each, %7.3f ms, shader clock %4.0f MHz\n", nmc,
                   (double)(res[1] - res[0]) / (2.0 * nmc), (double)(res[17] - res[16]) / 1e5, (double)(res[1] - res[0]) / ((double)(res[17] - res[16]) / 100.0));
            hipLaunchKernelGGL((k<2, 2, 2, true>), dim3(256), dim3(512), 64 * 1024, 0, nmc, out, res_d);
            hipDeviceSynchronize();
            hipMemcpy(res, res_d, sizeof(res), hipMemcpyDeviceToHost);
            printf("  + 2 ds_read_b128 per MFMA,   %8d per wave: %6.1f cycles each, %7.3f ms, shader clock %4.0f MHz\n", nmc,
                   (double)(res[1] - res[0]) / (2.0 * nmc), (double)(res[17] - res[16]) / 1e5, (double)(res[1] - res[0]) / ((double)(res[17] - res[16]) / 100.0));
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'f') {  // only the question of the reads that feed the products (independent reads for comparison)
        run<1, 2, 2, true>(nm, out, res_d);
        run<2, 2, 2, true>(nm, out, res_d);
        sweep_fed<1, 2>(nm, out, res_d);
        sweep_fed<2, 2>(nm, out, res_d);
        sweep_fed<2, 1>(nm, out, res_d);
        run_fed<3, 2, 2, 3, 0>(nm, out, res_d);   // dependent triples
        run_fed<6, 2, 2, 3, 0>(nm, out, res_d);
        run_fed<3, 2, 2, 1, 1>(nm, out, res_d);   // the kernels' address pattern
        run_fed<3, 2, 2, 3, 1>(nm, out, res_d);   // both
        run_fed<6, 2, 2, 3, 1>(nm, out, res_d);
        run_phase<2, 1>(20000, out, res_d);
        run_phase<2, 0>(512, out, res_d);
        run_phase<1, 1>(512, out, res_d);
        run_phase<2, 1, 1>(512, out, res_d);
        run_phase<2, 1, 2>(512, out, res_d);
        run_phase<2, 1, 3>(512, out, res_d);
        run_phase<2, 1, 4>(512, out, res_d);
        return 0;
    }
    sweep<0, 1>(nm, out, res_d);
    sweep<0, 2>(nm, out, res_d);
    sweep<1, 1>(nm, out, res_d);
    sweep<1, 2>(nm, out, res_d);
    sweep<2, 1>(nm, out, res_d);
    sweep<2, 2>(nm, out, res_d);
    sweep<3, 1>(nm, out, res_d);
    sweep<3, 2>(nm, out, res_d);
    sweep<5, 1>(nm, out, res_d);
    sweep<5, 2>(nm, out, res_d);
    printf("---- around v_mfma_f32_32x32x16_f16 ----\n");
    sweep16<0, 1>(nm, out, res_d);
    sweep16<0, 2>(nm, out, res_d);
    sweep16<1, 2>(nm, out, res_d);
    sweep16<2, 1>(nm, out, res_d);
    sweep16<2, 2>(nm, out, res_d);
    sweep16<3, 2>(nm, out, res_d);
    sweep16<5, 2>(nm, out, res_d);
    return 0;
}
