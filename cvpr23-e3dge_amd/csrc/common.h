// Shared helpers for the gfx950 kernels behind include/e3dge_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/e3dge_hip.h"

namespace e3dge {

// Thread-local error text returned by e3dge_last_error().
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(e3dge_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// After a <<<>>> launch: convert a launch error into E3DGE_ERR_LAUNCH.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return E3DGE_OK;
}

constexpr int kWave = 64;  // CDNA wavefront

// "amax buffers": max |value| of an activation tensor, tracked by its producer so that the next modulated conv can pick
// its power-of-two operand scale without another pass over the data.  kAmaxSlots slots, kAmaxStride floats (one 128-B
// line) apart; producers atomically max into slot (block index mod slots) -- one word would serialise thousands of
// atomics -- and the consumer takes the maximum over the slots.  Values are non-negative floats, compared as uint32.
constexpr int kAmaxSlots = E3DGE_AMAX_SLOTS, kAmaxStride = E3DGE_AMAX_STRIDE;
__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

}  // namespace e3dge

#define E3DGE_REQUIRE(cond, ...)                                        \
    do {                                                                \
        if (!(cond)) return ::e3dge::fail(E3DGE_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)
