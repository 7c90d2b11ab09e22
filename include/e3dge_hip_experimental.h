/* Entry points and modes that exist only in -DE3DGE_EXPERIMENTAL builds of libe3dge_hip.so (tools/build_variant.sh <name>
 * -DE3DGE_EXPERIMENTAL): kernels kept for A/B measurements that are slower than the default path and/or use scratch.  Nothing
 * here is on a product path; the default build does not contain them (round-3 review: "shipped-but-dead code").
 *   - e3dge_ws_chain: the weight-stationary chain study of DESIGN.md 4.1d (tools/ws_proto.py)
 *   - precision E3DGE_PREC_F16X3_V1 of the forward launches: first-generation split-f16 kernel (4 waves x 32 points)
 *   - precision E3DGE_PREC_F16X3_G2 of the backward-type launches: the 8-wave backward / chain kernels (csrc/siren16_bwd.h)
 * In a default build the two precisions are refused with E3DGE_ERR_INVALID_ARG. */
#ifndef E3DGE_HIP_EXPERIMENTAL_H
#define E3DGE_HIP_EXPERIMENTAL_H
#include "e3dge_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* A WEIGHT-STATIONARY split-f16 chain of n_layers (<= 8) 256 x 256 layers  x <- sin(gamma * (W x) + beta)  over n_points (a
 * multiple of 128) points -- the hidden-layer core of the renderer with the operand roles swapped.
 *   film   : (n_layers, 2, 256) fp32: gamma / 128 (the image carries the factor 128), beta;   x, y : (n_points, 256) fp32
 *   grid   : workgroups (<= 0: 256, one per CU);  cycles: NULL or 512 int64 -- per workgroup shader cycles / 100-MHz ticks
 *            of layers 1.. of its first 128-point group (tools/ws_proto.py) */
int e3dge_ws_chain(const void* wimg, const float* film, const float* x, float* y, int n_layers, int n_points, int grid,
                   long long* cycles, e3dge_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
