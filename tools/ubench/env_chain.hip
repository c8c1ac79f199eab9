// Micro-benchmark: cycles per sample of the compressor-envelope recurrence on ONE wave
// (the serial critical path of k_comp_env).  hipcc --offload-arch=gfx950 -O3 env_chain.hip -o env_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V>
__global__ void k(const float *x, float *out, int n, float cat, float crl, long long *cyc) {
    __shared__ float tile[64 * 132];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 132; i += 64) tile[i] = x[i % 4096];
    __syncthreads();
    float y = 0.f;
    const float omc_a = 1.f - cat, omc_r = 1.f - crl;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n / 128; ++it) {
        const float *row = tile + lane * 132;
#pragma unroll 4
        for (int j = 0; j < 128; j += 4) {
            const float4 x4 = *(const float4 *)(row + j);
            float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = fabsf(xs[u]);
                if (V == 0) {  // reference form: d = y - v; select(fma(cat,d,v), fma(crl,d,v))
                    const float d = y - v;
                    const float ya = fmaf(cat, d, v), yr = fmaf(crl, d, v);
                    y = (d < 0.f) ? ya : yr;
                } else if (V == 1) {  // 2-deep: fma + max (valid when cat <= crl)
                    y = fmaxf(fmaf(cat, y, omc_a * v), fmaf(crl, y, omc_r * v));
                } else if (V == 2) {  // 1 fma only (lower bound of a dependent chain)
                    y = fmaf(crl, y, omc_r * v);
                } else {  // select on compare of y and v directly, both fmas from y
                    const float ya = fmaf(cat, y, omc_a * v), yr = fmaf(crl, y, omc_r * v);
                    y = (v > y) ? ya : yr;
                }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[lane] = y;
    if (lane == 0) *cyc = t1 - t0;
}

// packed / multi-stream variants.  PRE: (a, r) = (s(1-cat)|x|, s(1-crl)|x|) precomputed in LDS
// (what the mover waves would write); NS: independent streams per lane (ILP across streams).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool PRE, int NS>
__global__ void kp(const float *x, float *out, int n, float cat, float crl, long long *cyc) {
    extern __shared__ float tile[];  // [NS][64][LD]
    constexpr int LD = PRE ? 260 : 132;
    const int lane = threadIdx.x;
    for (int i = lane; i < NS * 64 * LD; i += 64) tile[i] = fabsf(x[i % 4096]) * (PRE ? ((i & 1) ? 1.f - crl : 1.f - cat) : 1.f);
    __syncthreads();
    float y[NS];
    for (int s = 0; s < NS; ++s) y[s] = 0.f;
    const f2 c2 = {cat, crl}, om2 = {1.f - cat, 1.f - crl};
    long long t0 = __builtin_readcyclecounter();
    f4v nq0[NS], nq1[NS];
    for (int s = 0; s < NS; ++s) {
        const float *row = tile + (s * 64 + lane) * LD;
        nq0[s] = *(const f4v *)row;
        nq1[s] = *(const f4v *)(row + 4);
    }
    for (int it = 0; it < n / 128; ++it) {
#pragma unroll 2
        for (int j = 0; j < 128; j += 4) {
            f4v cq0[NS], cq1[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {  // software pipeline: next group's LDS reads fly under this group's chain
                const float *row = tile + (s * 64 + lane) * LD;
                cq0[s] = nq0[s];
                cq1[s] = nq1[s];
                const int jn = (j + 4) & 127;
                if (PRE) {
                    nq0[s] = *(const f4v *)(row + 2 * jn);
                    nq1[s] = *(const f4v *)(row + 2 * jn + 4);
                } else {
                    nq0[s] = *(const f4v *)(row + jn);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    f2 ar;
                    if (PRE) ar = u == 0 ? (f2){cq0[s].x, cq0[s].y} : u == 1 ? (f2){cq0[s].z, cq0[s].w} : u == 2 ? (f2){cq1[s].x, cq1[s].y} : (f2){cq1[s].z, cq1[s].w};
                    else ar = om2 * (f2){cq0[s][u], cq0[s][u]};
                    const f2 t = __builtin_elementwise_fma(c2, (f2){y[s], y[s]}, ar);
                    y[s] = fmaxf(t.x, t.y);
                }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int s = 0; s < NS; ++s) acc += y[s];
    out[lane] = acc;
    if (lane == 0) *cyc = t1 - t0;
}

// NS independent streams per lane with the reference form (sub, 2 fma, cmp, cndmask)
template <int NS>
__global__ void km(const float *x, float *out, int n, float cat, float crl, long long *cyc) {
    extern __shared__ float tile[];
    const int lane = threadIdx.x;
    for (int i = lane; i < NS * 64 * 132; i += 64) tile[i] = x[i % 4096];
    __syncthreads();
    float y[NS];
    for (int s = 0; s < NS; ++s) y[s] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n / 128; ++it) {
#pragma unroll 2
        for (int j = 0; j < 128; j += 4) {
            f4v q[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) q[s] = *(const f4v *)(tile + (s * 64 + lane) * 132 + j);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float v = fabsf(q[s][u]), d = y[s] - v;
                    const float ya = fmaf(cat, d, v), yr = fmaf(crl, d, v);
                    y[s] = (d < 0.f) ? ya : yr;
                }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int s = 0; s < NS; ++s) acc += y[s];
    out[lane] = acc;
    if (lane == 0) *cyc = t1 - t0;
}

int main() {
    const int n = 1 << 20;
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = sinf(i * 0.37f) * 0.5f;
    float *dx, *dout; long long *dc;
    hipMalloc(&dx, 4096 * 4); hipMalloc(&dout, 256); hipMalloc(&dc, 8);
    hipMemcpy(dx, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    auto run = [&](auto kern, const char *name) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        kern<<<1, 64>>>(dx, dout, n, 0.9f, 0.999f, dc);
        hipEventRecord(a); kern<<<1, 64>>>(dx, dout, n, 0.9f, 0.999f, dc); hipEventRecord(b);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("%-28s %8.3f ms  %6.2f ns/sample  %7.2f counter-ticks/sample\n", name, ms, ms * 1e6 / n, (double)c / n);
    };
    run(k<0>, "V0 sub,2fma,cmp,cndmask");
    run(k<1>, "V1 2fma,max");
    run(k<2>, "V2 1 fma");
    run(k<3>, "V3 2fma,cmp(v>y),cndmask");
    auto run2 = [&](auto kern, const char *name, int ns, size_t lds) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        kern<<<1, 64, lds>>>(dx, dout, n, 0.9f, 0.999f, dc);
        hipEventRecord(a); kern<<<1, 64, lds>>>(dx, dout, n, 0.9f, 0.999f, dc); hipEventRecord(b);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s %8.3f ms  %6.2f ns/sample/stream-group  %6.2f ns per (sample, 64 streams)\n", name, ms, ms * 1e6 / n,
               ms * 1e6 / n / ns);
    };
    run2(kp<false, 1>, "V4 pk_mul,pk_fma,max", 1, 64 * 132 * 4);
    run2(kp<true, 1>, "V5 pre(a,r): pk_fma,max", 1, 64 * 260 * 4);
    run2(kp<true, 2>, "V7 pre, 2 streams/lane", 2, 2 * 64 * 260 * 4);
    run2(km<2>, "V6 ref form, 2 streams/lane", 2, 2 * 64 * 132 * 4);
    run2(km<4>, "V8 ref form, 4 streams/lane", 4, 4 * 64 * 132 * 4);
    return 0;
}
