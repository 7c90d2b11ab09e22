/* Modes that exist only in -DE3DGE_EXPERIMENTAL builds of libe3dge_hip.so (tools/build_variant.sh <name> -DE3DGE_EXPERIMENTAL):
 * kernels kept for A/B measurements that are slower than the default path and/or use scratch.  Nothing here is on a product path; the
 * default build does not contain them (round-3 review: "shipped-but-dead code").
 *   - precision E3DGE_PREC_F16X3_V1 of the forward launches: first-generation split-f16 kernel (4 waves x 32 points)
 *   - precision E3DGE_PREC_F16X3_G2 of the backward-type launches: the 8-wave backward / chain kernels (csrc/siren16_bwd.h)
 * In a default build the two precisions are refused with E3DGE_ERR_INVALID_ARG; e3dge_build_flags() & 1 says which build is loaded.
 * (The weight-stationary chain study e3dge_ws_chain of rounds 3-4 was removed in round 5: DESIGN.md 4.1d.) */
#ifndef E3DGE_HIP_EXPERIMENTAL_H
#define E3DGE_HIP_EXPERIMENTAL_H
#include "e3dge_hip.h"
#endif
