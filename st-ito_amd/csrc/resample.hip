// resample.hip -- band-limited sinc resampling on gfx950: torchaudio.functional.resample(x, orig, new) with its
// defaults (resampling_method "sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99), the call the reference
// makes in front of the path: st_ito/utils.py:462-463 (get_param_embeds), 131, 205; scripts/run_optim.py:446, 526.
//
// With orig / new reduced by their gcd, output sample k * new + j is the dot product of phase j's kernel
// (kw = 2 * width + orig taps, built on the host in float64 exactly as the library does and rounded to float32)
// with the zero-padded input starting at k * orig - width: a polyphase FIR, one thread per output sample.
// The kernel table is passed transposed, (kw, new), so that consecutive lanes (consecutive phases j) read
// consecutive table entries; the input reads of a wave are a handful of broadcast addresses.  HBM-bound in
// principle (4 B in + 4 B out per sample) but called once per run_es / get_param_embeds, never per candidate.
#include "common.h"

namespace stito {

__global__ __launch_bounds__(256) void k_resample(const float *__restrict__ x, int64_t n_in, const float *__restrict__ kt,
                                                   int orig, int newf, int width, int kw, float *__restrict__ out,
                                                   int64_t n_out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const float *xr = x + (int64_t)blockIdx.y * n_in;
    const int64_t k = o / newf;
    const int j = (int)(o - k * newf);
    const int64_t base = k * orig - width;
    float acc = 0.0f;
    for (int i = 0; i < kw; ++i) {
        const int64_t m = base + i;
        const float v = (m >= 0 && m < n_in) ? xr[m] : 0.0f;
        acc = fmaf(kt[(int64_t)i * newf + j], v, acc);
    }
    out[(int64_t)blockIdx.y * n_out + o] = acc;
}

}  // namespace stito

using namespace stito;

extern "C" int64_t stito_resample_num_samples(int64_t n_in, int orig, int newf) {
    if (n_in < 0 || orig <= 0 || newf <= 0) return -1;
    return (newf * n_in + orig - 1) / orig;  // ceil(new * length / orig), orig / new already reduced
}

extern "C" int stito_resample_sinc(const float *x_dev, int rows, int64_t n_in, const float *kernel_t_dev, int orig, int newf,
                                   int width, float *out_dev, int64_t n_out, void *stream) {
    STITO_REQUIRE(rows > 0 && n_in > 0, STITO_E_INVALID, "resample: empty input");
    STITO_REQUIRE(orig > 0 && newf > 0 && width > 0, STITO_E_INVALID, "resample: bad ratio %d -> %d (width %d)", orig, newf, width);
    STITO_REQUIRE(n_out == stito_resample_num_samples(n_in, orig, newf), STITO_E_INVALID,
                  "resample: output length %lld, expected %lld", (long long)n_out, (long long)stito_resample_num_samples(n_in, orig, newf));
    STITO_REQUIRE(rows < 65536 && (n_out + 255) / 256 < (1ll << 31), STITO_E_UNSUPPORTED, "resample: grid too large");
    hipLaunchKernelGGL(k_resample, dim3((unsigned)((n_out + 255) / 256), rows), dim3(256), 0, (hipStream_t)stream, x_dev, n_in,
                       kernel_t_dev, orig, newf, width, 2 * width + orig, out_dev, n_out);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}
