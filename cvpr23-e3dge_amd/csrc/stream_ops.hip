// HBM-bound stream kernels of the StyleGAN2 up-sampler: fused bias+activation, fused noise+bias+
// activation, and modulated-conv weight preparation.  gfx950 only; fp32.
//
// Roofline: all three are pure streams (8 B/element fwd for the activations), so the only goals are
// 16 B/lane coalesced accesses, no per-element integer division, and >= 2k workgroups in flight.
#include "common.h"

namespace e3dge {

static thread_local char g_err[512] = {0};
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------------------------------------
// fused_bias_act  (reference: project/models/op/fused_bias_act_kernel.cu:19-49)
// ---------------------------------------------------------------------------------------------
enum ActMode { kLinear = 0, kZero = 1, kLrelu = 2, kLreluGrad = 3 };

template <int MODE>
__device__ __forceinline__ float act_one(float x, float ref, float alpha, float scale) {
    float y;
    if (MODE == kLinear) y = x;
    else if (MODE == kZero) y = 0.0f;
    else if (MODE == kLrelu) y = (x > 0.0f) ? x : __fmul_rn(x, alpha);
    else y = (ref > 0.0f) ? x : __fmul_rn(x, alpha);
    return __fmul_rn(y, scale);
}

constexpr int kActThreads = 256;
constexpr int kActVecPerThread = 4;                                  // float4s per thread
constexpr int kActChunk = kActThreads * kActVecPerThread * 4;        // floats per workgroup

// Storage type T: float, or _Float16 for e3dge_fused_bias_act_f16 (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// fused_bias_act_kernel.cu:79).  Arithmetic is fp32 either way; a half tensor is widened on load and rounded once (RNE) on store.
// One 16-byte vector = 4 floats or 8 halves; the row kernel below keeps its float4 form for T = float and has a twin for halves.
typedef _Float16 half8v __attribute__((ext_vector_type(8)));

template <int MODE, bool HAS_BIAS, bool HAS_REF>
__global__ void __launch_bounds__(kActThreads)
bias_act_rows_f16_kernel(_Float16* __restrict__ y, const _Float16* __restrict__ x, const _Float16* __restrict__ bias,
                         const _Float16* __restrict__ ref, float alpha, float scale, int step_b, int size_b, int chunks_per_row) {
    const int row = blockIdx.x / chunks_per_row;
    const int chunk = blockIdx.x - row * chunks_per_row;
    const float b = HAS_BIAS ? (float)bias[row % size_b] : 0.0f;
    const int64_t base = (int64_t)row * step_b;
    const int nvec = step_b >> 3;
    const half8v* x8 = reinterpret_cast<const half8v*>(x + base);
    const half8v* r8 = HAS_REF ? reinterpret_cast<const half8v*>(ref + base) : nullptr;
    half8v* y8 = reinterpret_cast<half8v*>(y + base);
    const int v0 = chunk * (kActThreads * kActVecPerThread) + threadIdx.x;
    half8v xv[kActVecPerThread], rv[kActVecPerThread];
#pragma unroll
    for (int j = 0; j < kActVecPerThread; ++j) {
        const int v = v0 + j * kActThreads;
        if (v < nvec) {
            xv[j] = x8[v];
            if (HAS_REF) rv[j] = r8[v];
        }
    }
#pragma unroll
    for (int j = 0; j < kActVecPerThread; ++j) {
        const int v = v0 + j * kActThreads;
        if (v < nvec) {
            half8v o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xe = (float)xv[j][e];
                o[e] = (_Float16)act_one<MODE>(HAS_BIAS ? xe + b : xe, HAS_REF ? (float)rv[j][e] : 0.0f, alpha, scale);
            }
            y8[v] = o;
        }
    }
}

// Row kernel: x viewed as (rows, step_b) with one bias value per row; step_b % 4 == 0.
template <int MODE, bool HAS_BIAS, bool HAS_REF>
__global__ void __launch_bounds__(kActThreads)
bias_act_rows_kernel(float* __restrict__ y, const float* __restrict__ x,
                     const float* __restrict__ bias, const float* __restrict__ ref, float alpha,
                     float scale, int step_b, int size_b, int chunks_per_row) {
    const int row = blockIdx.x / chunks_per_row;
    const int chunk = blockIdx.x - row * chunks_per_row;
    const float b = HAS_BIAS ? bias[row % size_b] : 0.0f;
    const int64_t base = (int64_t)row * step_b;
    const int nvec = step_b >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* r4 = HAS_REF ? reinterpret_cast<const float4*>(ref + base) : nullptr;
    float4* y4 = reinterpret_cast<float4*>(y + base);
    const int v0 = chunk * (kActThreads * kActVecPerThread) + threadIdx.x;
    float4 xv[kActVecPerThread], rv[kActVecPerThread];
#pragma unroll
    for (int j = 0; j < kActVecPerThread; ++j) {
        const int v = v0 + j * kActThreads;
        if (v < nvec) {
            xv[j] = x4[v];
            if (HAS_REF) rv[j] = r4[v];
        }
    }
#pragma unroll
    for (int j = 0; j < kActVecPerThread; ++j) {
        const int v = v0 + j * kActThreads;
        if (v < nvec) {
            float4 o;
            o.x = act_one<MODE>(HAS_BIAS ? xv[j].x + b : xv[j].x, HAS_REF ? rv[j].x : 0.0f, alpha, scale);
            o.y = act_one<MODE>(HAS_BIAS ? xv[j].y + b : xv[j].y, HAS_REF ? rv[j].y : 0.0f, alpha, scale);
            o.z = act_one<MODE>(HAS_BIAS ? xv[j].z + b : xv[j].z, HAS_REF ? rv[j].z : 0.0f, alpha, scale);
            o.w = act_one<MODE>(HAS_BIAS ? xv[j].w + b : xv[j].w, HAS_REF ? rv[j].w : 0.0f, alpha, scale);
            y4[v] = o;
        }
    }
}

// Generic element kernel (any step_b, e.g. the (B, C) outputs of MappingLinear / EqualLinear).
template <int MODE, bool HAS_BIAS, bool HAS_REF, typename T>
__global__ void __launch_bounds__(kActThreads)
bias_act_elem_kernel(T* __restrict__ y, const T* __restrict__ x,
                     const T* __restrict__ bias, const T* __restrict__ ref, float alpha,
                     float scale, int n, int step_b, int size_b) {
    for (int i = blockIdx.x * kActThreads + threadIdx.x; i < n; i += gridDim.x * kActThreads) {
        float v = (float)x[i];
        if (HAS_BIAS) v += (float)bias[(i / step_b) % size_b];
        y[i] = (T)act_one<MODE>(v, HAS_REF ? (float)ref[i] : 0.0f, alpha, scale);
    }
}

template <int MODE, bool HAS_BIAS, bool HAS_REF, typename T>
static int launch_bias_act(T* y, const T* x, const T* bias, const T* ref,
                           float alpha, float scale, int64_t n, int64_t step_b, int64_t size_b,
                           hipStream_t st) {
    constexpr int EPV = 16 / (int)sizeof(T);                 // elements per 16-byte vector
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                           (HAS_REF ? reinterpret_cast<uintptr_t>(ref) : 0)) & 15) == 0;
    // Without a bias the whole tensor is one "row"; keep rows < 2^31 elements.
    int64_t row_len = HAS_BIAS ? step_b : n;
    if (aligned && row_len >= EPV && (row_len % EPV) == 0 && (n % row_len) == 0) {
        const int64_t rows = n / row_len;
        const int64_t chunk = (int64_t)kActThreads * kActVecPerThread * EPV;
        const int64_t cpr = (row_len + chunk - 1) / chunk;
        const int64_t blocks = rows * cpr;
        if (blocks < (int64_t)1 << 31) {
            if constexpr (sizeof(T) == 4)
                bias_act_rows_kernel<MODE, HAS_BIAS, HAS_REF>
                    <<<dim3((unsigned)blocks), dim3(kActThreads), 0, st>>>(
                        y, x, bias, ref, alpha, scale, (int)row_len, HAS_BIAS ? (int)size_b : 1, (int)cpr);
            else
                bias_act_rows_f16_kernel<MODE, HAS_BIAS, HAS_REF>
                    <<<dim3((unsigned)blocks), dim3(kActThreads), 0, st>>>(
                        y, x, bias, ref, alpha, scale, (int)row_len, HAS_BIAS ? (int)size_b : 1, (int)cpr);
            return check_launch("fused_bias_act(rows)");
        }
    }
    int64_t blocks = (n + kActThreads - 1) / kActThreads;
    if (blocks > 8192) blocks = 8192;
    bias_act_elem_kernel<MODE, HAS_BIAS, HAS_REF, T><<<dim3((unsigned)blocks), dim3(kActThreads), 0, st>>>(
        y, x, bias, ref, alpha, scale, (int)n, HAS_BIAS ? (int)step_b : 1, HAS_BIAS ? (int)size_b : 1);
    return check_launch("fused_bias_act(elem)");
}

template <int MODE, typename T>
static int dispatch_bias_act(T* y, const T* x, const T* bias, const T* ref,
                             float alpha, float scale, int64_t n, int64_t step_b, int64_t size_b,
                             hipStream_t st) {
    const bool hb = bias != nullptr && size_b > 0;
    const bool hr = (MODE == kLreluGrad) && ref != nullptr;
    if (hb && hr) return launch_bias_act<MODE, true, true, T>(y, x, bias, ref, alpha, scale, n, step_b, size_b, st);
    if (hb) return launch_bias_act<MODE, true, false, T>(y, x, bias, ref, alpha, scale, n, step_b, size_b, st);
    if (hr) return launch_bias_act<MODE, false, true, T>(y, x, bias, ref, alpha, scale, n, step_b, size_b, st);
    return launch_bias_act<MODE, false, false, T>(y, x, bias, ref, alpha, scale, n, step_b, size_b, st);
}

// ---------------------------------------------------------------------------------------------
// noise + bias + lrelu  (reference chain: stylesdf_model.py:466 then fused_act.py:55-118)
// ---------------------------------------------------------------------------------------------
template <bool HAS_NOISE, bool HAS_BIAS>
__global__ void __launch_bounds__(kActThreads)
noise_bias_act_kernel(float* __restrict__ y, const float* __restrict__ x,
                      const float* __restrict__ noise, const float* __restrict__ noise_weight,
                      const float* __restrict__ bias, float alpha, float scale, int channels, int hw,
                      int noise_batch, int chunks_per_row) {
    const int row = blockIdx.x / chunks_per_row;  // b * channels + c
    const int chunk = blockIdx.x - row * chunks_per_row;
    const int b = row / channels;
    const int c = row - b * channels;
    const float bc = HAS_BIAS ? bias[c] : 0.0f;
    const float nw = HAS_NOISE ? noise_weight[0] : 0.0f;
    const int nvec = hw >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + (int64_t)row * hw);
    const float4* n4 = HAS_NOISE
        ? reinterpret_cast<const float4*>(noise + (int64_t)(noise_batch == 1 ? 0 : b) * hw) : nullptr;
    float4* y4 = reinterpret_cast<float4*>(y + (int64_t)row * hw);
    const int v0 = chunk * (kActThreads * kActVecPerThread) + threadIdx.x;
    float4 xv[kActVecPerThread], nv[kActVecPerThread];
#pragma unroll
    for (int j = 0; j < kActVecPerThread; ++j) {
        const int v = v0 + j * kActThreads;
        if (v < nvec) {
            xv[j] = x4[v];
            if (HAS_NOISE) nv[j] = n4[v];
        }
    }
#pragma unroll
    for (int j = 0; j < kActVecPerThread; ++j) {
        const int v = v0 + j * kActThreads;
        if (v < nvec) {
            float4 t = xv[j];
            if (HAS_NOISE) {  // image + weight * noise : a rounded product, then a rounded add
                t.x = __fadd_rn(t.x, __fmul_rn(nw, nv[j].x));
                t.y = __fadd_rn(t.y, __fmul_rn(nw, nv[j].y));
                t.z = __fadd_rn(t.z, __fmul_rn(nw, nv[j].z));
                t.w = __fadd_rn(t.w, __fmul_rn(nw, nv[j].w));
            }
            float4 o;
            o.x = act_one<kLrelu>(HAS_BIAS ? t.x + bc : t.x, 0.0f, alpha, scale);
            o.y = act_one<kLrelu>(HAS_BIAS ? t.y + bc : t.y, 0.0f, alpha, scale);
            o.z = act_one<kLrelu>(HAS_BIAS ? t.z + bc : t.z, 0.0f, alpha, scale);
            o.w = act_one<kLrelu>(HAS_BIAS ? t.w + bc : t.w, 0.0f, alpha, scale);
            y4[v] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// modulated-conv weight preparation  (reference: stylesdf_model.py:317-338)
// ---------------------------------------------------------------------------------------------
constexpr int kModThreads = 256;

__global__ void __launch_bounds__(kModThreads)
modconv_weights_kernel(float* __restrict__ out, const float* __restrict__ weight,
                       const float* __restrict__ style, float scale, int demodulate, int transpose,
                       int co, int ci, int kk) {
    __shared__ float red[kModThreads / kWave];
    __shared__ float demod_s;
    const int o = blockIdx.x % co;
    const int b = blockIdx.x / co;
    const int n = ci * kk;
    const float* w = weight + (int64_t)o * n;
    const float* s = style + (int64_t)b * ci;
    float sq = 0.0f;
    for (int e = threadIdx.x; e < n; e += kModThreads) {
        const int i = e / kk;
        const float v = __fmul_rn(__fmul_rn(scale, w[e]), s[i]);  // (scale * W) * style
        sq = fmaf(v, v, sq);
    }
    float d = 1.0f;
    if (demodulate) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int k = 0; k < kModThreads / kWave; ++k) t += red[k];
            demod_s = 1.0f / sqrtf(t + 1e-8f);
        }
        __syncthreads();
        d = demod_s;
    }
    for (int e = threadIdx.x; e < n; e += kModThreads) {
        const int i = e / kk;
        const int k = e - i * kk;
        float v = __fmul_rn(__fmul_rn(scale, w[e]), s[i]);
        if (demodulate) v = __fmul_rn(v, d);
        const int64_t dst = transpose ? (((int64_t)b * ci + i) * co + o) * kk + k
                                      : (((int64_t)b * co + o) * ci + i) * kk + k;
        out[dst] = v;
    }
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int e3dge_abi_version(void) { return 14; }
extern "C" const char* e3dge_last_error(void) { return err_buf(); }
extern "C" int e3dge_build_flags(void) {
#ifdef E3DGE_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}

extern "C" int64_t e3dge_stream_capture_id(e3dge_stream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(as_stream(stream), &st, &id) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return st == hipStreamCaptureStatusActive ? (int64_t)id + 1 : 0;
}

template <typename T>
static int fused_bias_act_any(T* y, const T* x, const T* bias, const T* ref, int act, int grad, float alpha, float scale, int64_t n,
                              int64_t step_b, int64_t size_b, e3dge_stream_t stream) {
    E3DGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "fused_bias_act: n=%lld outside int32 range", (long long)n);
    if (n == 0) return E3DGE_OK;
    E3DGE_REQUIRE(x && y, "fused_bias_act: null x/y");
    const bool has_bias = bias != nullptr && size_b > 0;
    E3DGE_REQUIRE(!has_bias || step_b >= 1, "fused_bias_act: step_b=%lld", (long long)step_b);
    hipStream_t st = as_stream(stream);
    // act*10+grad table of the reference (:35-45); anything unknown falls to 'linear' like its default.
    const int code = act * 10 + grad;
    switch (code) {
        case 12: case 32:
            return dispatch_bias_act<kZero, T>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
        case 30:
            return dispatch_bias_act<kLrelu, T>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
        case 31:
            return dispatch_bias_act<kLreluGrad, T>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
        default:
            return dispatch_bias_act<kLinear, T>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
    }
}

extern "C" int e3dge_fused_bias_act(float* y, const float* x, const float* bias, const float* ref,
                                    int act, int grad, float alpha, float scale, int64_t n,
                                    int64_t step_b, int64_t size_b, e3dge_stream_t stream) {
    return fused_bias_act_any<float>(y, x, bias, ref, act, grad, alpha, scale, n, step_b, size_b, stream);
}

extern "C" int e3dge_fused_bias_act_f16(void* y, const void* x, const void* bias, const void* ref,
                                        int act, int grad, float alpha, float scale, int64_t n,
                                        int64_t step_b, int64_t size_b, e3dge_stream_t stream) {
    return fused_bias_act_any<_Float16>(static_cast<_Float16*>(y), static_cast<const _Float16*>(x), static_cast<const _Float16*>(bias),
                                        static_cast<const _Float16*>(ref), act, grad, alpha, scale, n, step_b, size_b, stream);
}

// fp64 form (ABI 12): the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF (fused_bias_act_kernel.cu:79), so double tensors --
// what torch.autograd.gradcheck / gradgradcheck feed an op -- run its kernel with scalar_t = double arithmetic.  Same here: a plain
// element kernel in double (alpha / scale arrive as float and are widened, as `scalar_t alpha = (scalar_t)alpha` does there).
namespace e3dge {
template <int MODE, bool HAS_BIAS, bool HAS_REF>
__global__ void __launch_bounds__(kActThreads)
bias_act_elem_f64_kernel(double* __restrict__ y, const double* __restrict__ x, const double* __restrict__ bias,
                         const double* __restrict__ ref, double alpha, double scale, int n, int step_b, int size_b) {
    // (64-bit index: n may be within one grid stride of INT_MAX, where an int increment would overflow)
    for (int64_t i = (int64_t)blockIdx.x * kActThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kActThreads) {
        double v = x[i];
        if (HAS_BIAS) v += bias[(int)((i / step_b) % size_b)];
        double o;
        if (MODE == kLinear) o = v;
        else if (MODE == kZero) o = 0.0;
        else if (MODE == kLrelu) o = v > 0.0 ? v : v * alpha;
        else o = (HAS_REF ? ref[i] : 0.0) > 0.0 ? v : v * alpha;
        y[i] = o * scale;
    }
}
template <int MODE>
static int dispatch_bias_act_f64(double* y, const double* x, const double* bias, const double* ref, float alpha, float scale, int64_t n,
                                 int64_t step_b, int64_t size_b, hipStream_t st) {
    const bool hb = bias != nullptr && size_b > 0, hr = (MODE == kLreluGrad) && ref != nullptr;
    int64_t blocks = (n + kActThreads - 1) / kActThreads;
    if (blocks > 8192) blocks = 8192;
    const dim3 g((unsigned)blocks), t(kActThreads);
    const int sb = hb ? (int)step_b : 1, zb = hb ? (int)size_b : 1;
    if (hb && hr) bias_act_elem_f64_kernel<MODE, true, true><<<g, t, 0, st>>>(y, x, bias, ref, alpha, scale, (int)n, sb, zb);
    else if (hb) bias_act_elem_f64_kernel<MODE, true, false><<<g, t, 0, st>>>(y, x, bias, ref, alpha, scale, (int)n, sb, zb);
    else if (hr) bias_act_elem_f64_kernel<MODE, false, true><<<g, t, 0, st>>>(y, x, bias, ref, alpha, scale, (int)n, sb, zb);
    else bias_act_elem_f64_kernel<MODE, false, false><<<g, t, 0, st>>>(y, x, bias, ref, alpha, scale, (int)n, sb, zb);
    return check_launch("fused_bias_act(f64)");
}
}  // namespace e3dge

extern "C" int e3dge_fused_bias_act_f64(double* y, const double* x, const double* bias, const double* ref, int act, int grad, float alpha,
                                        float scale, int64_t n, int64_t step_b, int64_t size_b, e3dge_stream_t stream) {
    E3DGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "fused_bias_act_f64: n=%lld outside int32 range", (long long)n);
    if (n == 0) return E3DGE_OK;
    E3DGE_REQUIRE(x && y, "fused_bias_act_f64: null x/y");
    const bool has_bias = bias != nullptr && size_b > 0;
    E3DGE_REQUIRE(!has_bias || step_b >= 1, "fused_bias_act_f64: step_b=%lld", (long long)step_b);
    hipStream_t st = as_stream(stream);
    switch (act * 10 + grad) {       // the reference's table (:35-45)
        case 12: case 32: return dispatch_bias_act_f64<kZero>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
        case 30: return dispatch_bias_act_f64<kLrelu>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
        case 31: return dispatch_bias_act_f64<kLreluGrad>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
        default: return dispatch_bias_act_f64<kLinear>(y, x, has_bias ? bias : nullptr, ref, alpha, scale, n, step_b, size_b, st);
    }
}

extern "C" int e3dge_noise_bias_act(float* y, const float* x, const float* noise,
                                    const float* noise_weight, const float* bias, float alpha,
                                    float scale, int64_t batch, int64_t channels, int64_t hw,
                                    int64_t noise_batch, e3dge_stream_t stream) {
    const int64_t n = batch * channels * hw;
    if (n == 0) return E3DGE_OK;
    E3DGE_REQUIRE(x && y, "noise_bias_act: null x/y");
    E3DGE_REQUIRE(n < ((int64_t)1 << 31), "noise_bias_act: n=%lld outside int32 range", (long long)n);
    E3DGE_REQUIRE(hw % 4 == 0, "noise_bias_act: hw=%lld must be a multiple of 4", (long long)hw);
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                    reinterpret_cast<uintptr_t>(noise)) & 15) == 0, "noise_bias_act: pointers must be 16-B aligned");
    const bool hn = noise != nullptr;
    E3DGE_REQUIRE(!hn || (noise_weight && (noise_batch == 1 || noise_batch == batch)),
                  "noise_bias_act: noise_batch=%lld must be 1 or batch", (long long)noise_batch);
    const int64_t cpr = (hw + kActChunk - 1) / kActChunk;
    const int64_t blocks = batch * channels * cpr;
    hipStream_t st = as_stream(stream);
    dim3 g((unsigned)blocks), t(kActThreads);
    if (hn && bias) noise_bias_act_kernel<true, true><<<g, t, 0, st>>>(y, x, noise, noise_weight, bias, alpha, scale, (int)channels, (int)hw, (int)noise_batch, (int)cpr);
    else if (hn) noise_bias_act_kernel<true, false><<<g, t, 0, st>>>(y, x, noise, noise_weight, bias, alpha, scale, (int)channels, (int)hw, (int)noise_batch, (int)cpr);
    else if (bias) noise_bias_act_kernel<false, true><<<g, t, 0, st>>>(y, x, noise, noise_weight, bias, alpha, scale, (int)channels, (int)hw, (int)noise_batch, (int)cpr);
    else noise_bias_act_kernel<false, false><<<g, t, 0, st>>>(y, x, noise, noise_weight, bias, alpha, scale, (int)channels, (int)hw, (int)noise_batch, (int)cpr);
    return check_launch("noise_bias_act");
}

// ---------------------------------------------------------------------------------------------
// ToRGB (stylesdf_model.py:510-541) in one pass: 1x1 modulated conv WITHOUT demodulation (:232), + bias, + the skip image
// up-sampled by upfirdn2d(up=2, pad=(2,1)) with the 4x4 FIR (Upsample :96-119).  Bound: HBM (reads Ci floats per pixel).
// 256 threads = pixel lanes (4 adjacent pixels each, float4 loads) x channel groups, partial sums folded through LDS in group order.
// ---------------------------------------------------------------------------------------------
constexpr int kRgbThreads = 256, kRgbMaxCi = 1024;
// G channel groups x (256 / G) pixel lanes of 4 adjacent pixels.  G follows the channel count (16 for Ci >= 256, 4 for 128, 1
// below): the deep, small levels (512 x 64^2) need the channels spread over threads to have enough loads in flight -- a thread
// walking 128 channels eight at a time took 16 us for 8 MB --, the wide, shallow ones (32 x 1024^2) are a plain stream with
// every channel of a pixel in one thread and no reduction.
template <int G>
__global__ void __launch_bounds__(kRgbThreads)
torgb_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ weight,
             const float* __restrict__ style, const float* __restrict__ bias, const float* __restrict__ skip,
             const float* __restrict__ fir, float scale, int Ci, int H, int W, int blocks_per_img) {
    constexpr int PL = kRgbThreads / G;                        // pixel lanes per workgroup
    __shared__ float wm[3 * kRgbMaxCi];
    __shared__ float part[G > 1 ? G - 1 : 1][3][PL][4];
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x - b * blocks_per_img;
    const int pl = threadIdx.x % PL, grp = threadIdx.x / PL;
    const int HW = H * W;
    for (int i = threadIdx.x; i < 3 * Ci; i += kRgbThreads) {
        const int c = i / Ci, ci = i - c * Ci;
        wm[i] = __fmul_rn(__fmul_rn(scale, weight[c * Ci + ci]), style[(int64_t)b * Ci + ci]);      // (scale * W) * s, :321
    }
    __syncthreads();
    const int p0 = (blk * PL + pl) * 4;                        // HW is a multiple of 4 (checked by the launcher)
    const bool live = p0 < HW;
    float acc[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (live) {
        const int per = (Ci + G - 1) / G, c0 = grp * per, c1 = min(Ci, c0 + per);
        const float4* xp = reinterpret_cast<const float4*>(x + ((int64_t)b * Ci + c0) * HW + p0);
        constexpr int UN = 16;                                 // loads in flight per thread
        int ci = c0;
        for (; ci + UN <= c1; ci += UN) {
            float4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) v[u] = xp[(int64_t)u * (HW / 4)];
            xp += (int64_t)UN * (HW / 4);
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float w0 = wm[ci + u], w1 = wm[Ci + ci + u], w2 = wm[2 * Ci + ci + u];
                acc[0][0] = fmaf(w0, v[u].x, acc[0][0]); acc[0][1] = fmaf(w0, v[u].y, acc[0][1]); acc[0][2] = fmaf(w0, v[u].z, acc[0][2]); acc[0][3] = fmaf(w0, v[u].w, acc[0][3]);
                acc[1][0] = fmaf(w1, v[u].x, acc[1][0]); acc[1][1] = fmaf(w1, v[u].y, acc[1][1]); acc[1][2] = fmaf(w1, v[u].z, acc[1][2]); acc[1][3] = fmaf(w1, v[u].w, acc[1][3]);
                acc[2][0] = fmaf(w2, v[u].x, acc[2][0]); acc[2][1] = fmaf(w2, v[u].y, acc[2][1]); acc[2][2] = fmaf(w2, v[u].z, acc[2][2]); acc[2][3] = fmaf(w2, v[u].w, acc[2][3]);
            }
        }
        for (; ci < c1; ++ci) {
            const float4 v = *xp;
            xp += HW / 4;
            const float w0 = wm[ci], w1 = wm[Ci + ci], w2 = wm[2 * Ci + ci];
            acc[0][0] = fmaf(w0, v.x, acc[0][0]); acc[0][1] = fmaf(w0, v.y, acc[0][1]); acc[0][2] = fmaf(w0, v.z, acc[0][2]); acc[0][3] = fmaf(w0, v.w, acc[0][3]);
            acc[1][0] = fmaf(w1, v.x, acc[1][0]); acc[1][1] = fmaf(w1, v.y, acc[1][1]); acc[1][2] = fmaf(w1, v.z, acc[1][2]); acc[1][3] = fmaf(w1, v.w, acc[1][3]);
            acc[2][0] = fmaf(w2, v.x, acc[2][0]); acc[2][1] = fmaf(w2, v.y, acc[2][1]); acc[2][2] = fmaf(w2, v.z, acc[2][2]); acc[2][3] = fmaf(w2, v.w, acc[2][3]);
        }
    }
    if (G > 1) {
        if (grp > 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[grp - 1][c][pl][j] = acc[c][j];
        }
        __syncthreads();
    }
    if (grp != 0 || !live) return;
    const int oy = p0 / W, ox = p0 - oy * W;                  // 4 pixels of one row (W % 4 == 0)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float conv = acc[c][j];
            if (G > 1) {
#pragma unroll
                for (int gq = 0; gq < G - 1; ++gq) conv += part[gq][c][pl][j];       // fixed order: group 0, 1, 2, ...
            }
            o[j] = conv + bias[c];
        }
        if (skip) {
            // upfirdn2d(skip, fir, up=2, pad=(2,1)): only the taps with (o + k - 2) even meet a sample; same tap order
            // (ky, then kx) and the same fma chain as e3dge_upfirdn2d, so the sum is bit-identical to it
            const int h = H >> 1, w = W >> 1;
            const float* sp = skip + ((int64_t)b * 3 + c) * h * w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = ox + j;
                float u = 0.0f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ky = (oy & 1) + 2 * a, iy = (oy + ky - 2) >> 1;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int kx = (xx & 1) + 2 * e, ix = (xx + kx - 2) >> 1;
                        if (iy >= 0 && iy < h && ix >= 0 && ix < w) u = fmaf(sp[iy * w + ix], fir[(3 - ky) * 4 + (3 - kx)], u);
                    }
                }
                o[j] = o[j] + u;
            }
        }
        *reinterpret_cast<float4*>(y + ((int64_t)b * 3 + c) * HW + p0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int e3dge_torgb(float* y, const float* x, const float* weight, const float* style, const float* bias,
                           const float* skip, const float* fir, float scale, int batch, int ci, int height, int width,
                           e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && ci >= 1 && ci <= kRgbMaxCi && height >= 1 && width >= 1, "torgb: bad sizes (ci <= %d)", kRgbMaxCi);
    if (batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(y && x && weight && style && bias, "torgb: null pointer");
    E3DGE_REQUIRE(skip == nullptr || fir != nullptr, "torgb: skip needs the 4x4 FIR");
    E3DGE_REQUIRE(width % 4 == 0 && (skip == nullptr || (height % 2 == 0 && width % 2 == 0)), "torgb: width must be a multiple of 4 (and even extents with a skip)");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "torgb: x / y must be 16-B aligned");
    const int64_t hw = (int64_t)height * width;
    E3DGE_REQUIRE(hw * ci < ((int64_t)1 << 31), "torgb: image too large");
    hipStream_t st = as_stream(stream);
    if (ci >= 256) {
        const int bpi = (int)((hw / 4 + 15) / 16);
        torgb_kernel<16><<<dim3((unsigned)(bpi * batch)), dim3(kRgbThreads), 0, st>>>(y, x, weight, style, bias, skip, fir, scale, ci, height, width, bpi);
    } else if (ci > 64) {
        const int bpi = (int)((hw / 4 + 63) / 64);
        torgb_kernel<4><<<dim3((unsigned)(bpi * batch)), dim3(kRgbThreads), 0, st>>>(y, x, weight, style, bias, skip, fir, scale, ci, height, width, bpi);
    } else {
        const int bpi = (int)((hw / 4 + 255) / 256);
        torgb_kernel<1><<<dim3((unsigned)(bpi * batch)), dim3(kRgbThreads), 0, st>>>(y, x, weight, style, bias, skip, fir, scale, ci, height, width, bpi);
    }
    return check_launch("torgb");
}

extern "C" int e3dge_modconv_weights(float* out, const float* weight, const float* style, float scale,
                                     int demodulate, int transpose, int batch, int co, int ci, int kk,
                                     e3dge_stream_t stream) {
    E3DGE_REQUIRE(out && weight && style, "modconv_weights: null pointer");
    E3DGE_REQUIRE(batch > 0 && co > 0 && ci > 0 && kk > 0, "modconv_weights: bad sizes");
    modconv_weights_kernel<<<dim3((unsigned)(batch * co)), dim3(kModThreads), 0, as_stream(stream)>>>(
        out, weight, style, scale, demodulate, transpose, co, ci, kk);
    return check_launch("modconv_weights");
}
