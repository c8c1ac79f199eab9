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
    return 0;
}
