// What does a v_fma_f64 cost on gfx950?  (Is k_eq -- 30 dependent-ish float64 FMAs per sample and pass -- at its arithmetic floor?)
// N independent accumulator chains per lane, W waves per SIMD; prints cycles per wave-instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 f64_rate.hip -o f64_rate && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, bool SGPR>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
    double acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x * 1e-3 + c;
    double va = SGPR ? a : a + threadIdx.x * 1e-9, vb = SGPR ? b : b + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = fma(acc[c], va, vb);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    if (s == 1234.5) out[0] = s;
}
template <int CHAINS, bool SGPR>
void run(int wgs_per_cu) {
    double *out; hipMalloc(&out, 8);
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, SGPR>), dim3(grid), dim3(256), 0, 0, out, 0.999999, 1e-7, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, SGPR>), dim3(grid), dim3(256), 0, 0, out, 0.999999, 1e-7, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr_per_simd = (double)iters * CHAINS * wgs_per_cu;   // one wave of each workgroup per SIMD
    printf("%d chains, %s operand, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s, %.2f cycles per wave-instruction and SIMD at 2.4 GHz\n", CHAINS,
           SGPR ? "scalar" : "vector", wgs_per_cu, ms, 2.0 * iters * CHAINS * 256.0 * grid / ms / 1e9, ms * 1e-3 * 2.4e9 / winstr_per_simd);
    hipFree(out);
}
int main() {
    run<1, true>(1); run<2, true>(1); run<4, true>(1); run<8, true>(1); run<8, false>(1);
    run<1, true>(2); run<4, true>(2); run<8, true>(2); run<8, true>(4); run<8, false>(4);
    return 0;
}
