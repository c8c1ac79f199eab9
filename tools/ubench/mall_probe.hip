// Calibration for the L2-miss counters: what do TCC_EA0_RDREQ / _RDREQ_LEVEL / _RDREQ_DRAM(_32B) read for
// a stream that the Infinity Cache (MALL, 256 MiB) serves, against one that has to come from HBM?
//
// One kernel, launched several times over buffers of different sizes.  Every workgroup reads the WHOLE buffer (grid-strided
// by XCD: the eight L2s each fetch every line, like eight cout tiles on eight XCDs reading the same V slab), 16 bytes per
// lane, several times.  A buffer below 4 MiB lives in every L2 (no EA traffic after the first pass), 8 - 128 MiB misses the
// L2s but fits the MALL, 1 GiB and up comes from HBM.  The fabric cannot be told from DRAM at the TCC's counters unless the
// LATENCY of the EA reads differs: average latency = RDREQ_LEVEL / RDREQ (Little).
//
//   hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe && ./mall_probe
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// blocks b % 8 == xcd: each XCD's blocks together sweep the buffer `passes` times
template <int TAG>
__global__ __launch_bounds__(256) void k_probe(const f4 *__restrict__ buf, size_t n16, int passes, float *out, size_t limit16) {
    const int xcd = blockIdx.x & 7;
    const size_t per_xcd_blocks = gridDim.x >> 3;
    const size_t bi = blockIdx.x >> 3;
    f4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        // a different starting phase per XCD so that the eight sweeps are not in lockstep on the same lines
        size_t i = (bi * 256 + threadIdx.x + (size_t)xcd * (n16 / 8) + (size_t)p * 4096) % n16;
        for (size_t k = 0; k < limit16; k += per_xcd_blocks * 256) {
            size_t j = i + k;
            if (j >= n16) j -= n16;
            if (j < n16) {
                f4 v = buf[j];
                acc += v;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// "curve" mode: latency against load.  A 64 MiB buffer (Infinity-Cache resident) and a 2 GiB one (HBM), each swept by 8 x {4 .. 256}
// workgroups: the EA read latency (RDREQ_LEVEL / RDREQ) at 0.2 ... 6.8 TB/s for both sources -- the two curves a conv kernel's
// (rate, latency) point is read against.
static void curve(float *out) {
    const size_t sizes_mb[] = {64, 2048};
    const int per_xcd[] = {4, 8, 16, 32, 64, 128, 256};
    for (size_t s : sizes_mb) {
        const size_t bytes = s << 20, n16 = bytes / 16;
        f4 *buf;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        hipLaunchKernelGGL(k_probe<0>, dim3(2048), dim3(256), 0, 0, buf, n16, 1, out, n16);  // warm
        for (int g : per_xcd) {
            const size_t want16 = (size_t)g << 20;                       // ~16 MiB of loads per workgroup
            const size_t limit16 = want16 < n16 ? want16 : n16;
            const int passes = (int)(want16 / limit16);
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe<2>, dim3(g * 8), dim3(256), 0, 0, buf, n16, passes, out, limit16);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("curve: buffer %5zu MiB, %4d workgroups (%3d per XCD) x %3d passes: %8.3f ms  %6.2f TB/s loaded   (k_probe<2>)\n", s, g * 8, g, passes, ms,
                   8.0 * passes * limit16 * 16 / 1e9 / ms);
        }
        CHECK(hipFree(buf));
    }
}

int main(int argc, char **argv) {
    const size_t sizes_mb[] = {2, 16, 64, 128, 192, 512, 2048};
    float *out;
    CHECK(hipMalloc(&out, 64));
    if (argc > 1) { curve(out); return 0; }
    for (size_t s : sizes_mb) {
        const size_t bytes = s << 20, n16 = bytes / 16;
        f4 *buf;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        const int passes = (int)((4096 + s - 1) / s) < 2 ? 2 : (int)((4096 + s - 1) / s);  // >= 4 GiB of loads per XCD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int grid = 256 * 8;
        hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(256), 0, 0, buf, n16, 1, out, n16);  // warm
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(256), 0, 0, buf, n16, passes, out, n16);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double gb = 8.0 * passes * bytes / 1e9;
        printf("buffer %5zu MiB x %4d passes x 8 XCDs: %8.3f ms  %7.2f TB/s loaded   (timed launch = k_probe<1>)\n", s, passes, ms, gb / ms);
        CHECK(hipFree(buf));
    }
    return 0;
}
