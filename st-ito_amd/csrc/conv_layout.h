// conv_layout.h -- pieces shared by the 3x3-conv kernels of the Cnn14 trunk (cnn14.hip, conv_wino43.hip): the
// channel-blocked activation layout, native vector types, explicit packed-f32 arithmetic and the LDS-DMA copy.
#pragma once
#include "common.h"

namespace stito {

// Activation layout: channel-blocked NC8HW8 -- element (stream s, channel c, row h, column w) of a map
// with C channels lives at (((s * C/8 + c/8) * H + h) * W + w) * 8 + c%8.  Eight consecutive
// channels of a pixel are 32 B, and for one channel block consecutive pixels are contiguous: a
// K-chunk's halo patch rows are dense 32-B-per-pixel runs (NHWC would touch one 32-B piece out of
// every 4*C-byte pixel vector per chunk: 4x line over-fetch, measured as the Winograd limiter).
__device__ __forceinline__ int64_t act_off(int64_t s, int c, int h, int w, int C, int H, int W) {
    return ((((s * (C >> 3)) + (c >> 3)) * H + h) * (int64_t)W + w) * 8 + (c & 7);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ConvShape {
    int S, H, W, Cin, Cout;
};

// Packed f32 arithmetic, issued explicitly: inside the MFMA loops hipcc splits native-vector f32x4 expressions into
// scalar v_fma / v_sub (twice the VALU instructions, and every VALU instruction is taken out of the MFMA stream's time).
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {  // a + (-b): exact, bit-identical to v_sub_f32
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// LDS-DMA: 16 B per lane, global (sbase SGPR pair, wave-uniform, + voff per-lane byte offset) -> LDS at M0 + lane * 16,
// no VGPR round trip and no VALU instruction.  M0 is not saved: nothing else in the kernels that use this reads it
// (gfx950 LDS instructions do not, and they have no movrel / sendmsg / interp); hipcc rejects "m0" in a clobber list as
// reserved, so that is checked by inspection of the ISA (grep m0: only these statements write it).
__device__ __forceinline__ void glds16_m0(const float *sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// Division by a runtime divisor that is fixed per launch (map height, tile rows per stream, ...).  hipcc turns `a / d`
// into a ~35-instruction sequence; a conv workgroup does ten of them before it can issue its first copy (measured: 700 of
// its ~7 400 prologue cycles).  For 0 <= a < 2^22 and 1 <= d < 2^22 the float quotient is off by at most one, which the
// remainder check repairs exactly; larger operands take the compiler's division.
struct FDiv {
    int d;
    float rcp;
};
static inline FDiv make_fdiv(int d) { return FDiv{d, 1.0f / (float)d}; }
__device__ __forceinline__ int fdiv(int a, const FDiv f, int &rem) {
    if ((unsigned)(a | f.d) >= (1u << 22)) {
        rem = a % f.d;
        return a / f.d;
    }
    int q = (int)(__int2float_rn(a) * f.rcp);
    int r = a - q * f.d;
    if (r < 0) { --q; r += f.d; }
    else if (r >= f.d) { ++q; r -= f.d; }
    rem = r;
    return q;
}

// conv_wino43.hip
bool wino43_supported(const ConvShape &c, bool pool);
double wino43_issued_flops(const ConvShape &c, bool pool);
// amax_out (or NULL): per stream, the layer's largest output as a bit pattern (atomicMax into a buffer the caller zeroed);
// amax_in (or NULL): the same of the layer that produced `in` -- without it the split-precision launchers scan `in` themselves
int launch_wino43(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                  bool pool, long long *trace, hipStream_t st, unsigned *amax_out = nullptr);
int pack_wino43(const float *w_oihw, int cout, int cin, float *packed, hipStream_t st);
size_t wino43_pre_workspace_bytes(const ConvShape &c, bool pool);  // hoisted input transform: bytes of V slabs (0: unsupported)
int launch_wino43_pre(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                      bool pool, float *vbuf, size_t vbuf_bytes, hipStream_t st, unsigned *amax_out = nullptr);
// split-precision streaming kernel (f16 hi + lo operands, f32 accumulate)
bool wino43_split_supported(const ConvShape &c, bool pool);
size_t wino43_split_workspace_bytes(const ConvShape &c, bool pool);
size_t wino43_split_packed_floats(int cout, int cin);
int pack_wino43_split(const float *w_oihw, int cout, int cin, float *packed, int layout, hipStream_t st);  // 0: k_conv_wino43s, 1: s2, 2: s3
size_t wino43_split2_workspace_bytes(const ConvShape &c, bool pool);
double wino43_split2_issued_flops(const ConvShape &c, bool pool);
int launch_wino43_split2(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                         bool pool, void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr);
bool wino43_split3_supported(const ConvShape &c, bool pool);   // six sweeps, 128 x 128 workgroup tiles (cout % 512 == 0)
size_t wino43_split3_workspace_bytes(const ConvShape &c, bool pool);
double wino43_split3_issued_flops(const ConvShape &c, bool pool);
int64_t wino43_split3_workgroups(const ConvShape &c, bool pool);
int launch_wino43_split3(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                         bool pool, void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr);
int launch_wino43_split(const float *in, const float *upk, const float *scale, const float *shift, float *out, const ConvShape &c,
                        bool pool, void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr);
// conv_wino23r.hip: Winograd F(2x2,3x3) on the f16 matrix pipe, weights resident in registers (the 64-input-channel layers)
bool wino23r_supported(const ConvShape &c, bool pool);
double wino23r_issued_flops(const ConvShape &c, bool pool);
size_t wino23r_workspace_bytes(const ConvShape &c, bool pool);  // per-stream maxima of the input when the caller has none
size_t wino23r_packed_floats(int cout, int cin);
int pack_wino23r(const float *w_oihw, int cout, int cin, float *packed, hipStream_t st);
int launch_wino23r(const float *in, const float *wpk, const float *scale, const float *shift, float *out, const ConvShape &c, bool pool,
                   void *ws, size_t ws_bytes, hipStream_t st, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr);
// conv_block1 in one launch on that kernel: the first conv (1 -> 64 channels, bn1, ReLU) is computed on the matrix pipe into
// the patch ring in the slots where the unfused kernel issues its copies (c: the second conv's shape)
bool wino23r_fused1_supported(const ConvShape &c, bool pool);
size_t wino23r_fused1_workspace_bytes(const ConvShape &c, bool pool);
size_t conv1_f2reg_packed_floats();
int pack_conv1_f2reg(const float *w_dev, const float *scale_dev, const float *shift_dev, int c1, float *packed, hipStream_t st);
int launch_wino23r_fused1(const float *logmel, const float *c1pk, const float *wpk, const float *scale, const float *shift, float *out,
                          const ConvShape &c, bool pool, void *ws, size_t ws_bytes, hipStream_t st, unsigned *amax_out);

}  // namespace stito
