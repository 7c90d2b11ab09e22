// Helpers shared by the decoder kernels (modconv.hip: planar fp32 activations; decoder2.hip: packed split-f16 activations).
#pragma once
#include "siren_common.h"

namespace e3dge {

// max over the kAmaxSlots slots of an amax buffer (producers spread their atomics over the slots), wave-uniform
__device__ __forceinline__ float amax_read(const float* p, int lane) {
    float m = p[(lane & (kAmaxSlots - 1)) * kAmaxStride];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    return m;
}
// power-of-two operand scale for a bound on |s x|: sc = 2^(141 - eb) puts bound * sc into [2^14, 2^15);
// 1 / (128 * sc) = 2^(eb - 148) undoes it together with the weights' factor 128.
__device__ __forceinline__ unsigned scale_exponent(float bound) {
    unsigned eb = (__float_as_uint(bound) >> 23) & 255u;          // bound in [2^(eb-127), 2^(eb-126))
    return eb < 22u ? 22u : (eb > 250u ? 250u : eb);
}


// e3dge_decoder_styles (modconv.hip) with an optional block of floats to clear in the same launch (host side)
int decoder_styles_launch(const E3dgeModLayer* table, int n_layers, int total_rows, int total_co, const float* latent, int n_latent,
                          int style_dim, int batch, float* zero, int n_zero, hipStream_t st);

}  // namespace e3dge
