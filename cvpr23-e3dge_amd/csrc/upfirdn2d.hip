// upfirdn2d for gfx950: zero-insert up-sample, pad/crop, correlate with the FLIPPED FIR, decimate.
// Reference semantics: project/models/op/upfirdn2d_kernel.cu:49-207 (CUDA) and the PyTorch restatement
// project/models/op/upfirdn2d.py:157-200.  minor_dim == 1 (how every Python caller invokes it).
//
//   U[uy][ux] = x[(uy-pad_y0)/up_y][(ux-pad_x0)/up_x]  when both quotients are exact and in range, else 0
//   y[oy][ox] = sum_{ky,kx} U[oy*down_y+ky][ox*down_x+kx] * k[kh-1-ky][kw-1-kx]
//
// Roofline: pure HBM stream, 4*(in+out) bytes per plane.  The tiled kernel stages the up-sampled tile U
// in LDS once (so every input element is fetched from HBM once, plus the (k-1)-wide halo), each lane
// produces 1x4 adjacent outputs from two ds_read_b128 per tap row, and stores 16 B.
#include "common.h"

namespace e3dge {

__host__ __device__ inline int floor_div_i(int a, int b) {
    int q = a / b;
    return (q * b > a) ? q - 1 : q;
}

struct UpfirdnGeom {
    int in_h, in_w, out_h, out_w;
    int up_x, up_y, down_x, down_y;
    int pad_x0, pad_y0;
    int kh, kw;
};

// Storage type T of x / y: float, or _Float16 for the half entry points (the reference dispatches
// AT_DISPATCH_FLOATING_TYPES_AND_HALF, upfirdn2d_kernel.cu:311).  Arithmetic is fp32 either way: a half tensor is widened on
// load and rounded ONCE (RNE) on store -- at least as accurate as the reference's scalar_t = half accumulation.
template <typename T> __device__ __forceinline__ void store4(T* dst, const float (&v)[4], bool vec, int n_ok) {
    if (vec && n_ok >= 4) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<h4*>(dst) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n_ok) dst[j] = (T)v[j];
    }
}

// ---------------------------------------------------------------------------------------------
// Tiled kernel: square up factor UP, square decimation DN, KHxKW taps; tile 32 x 64 outputs.
// ---------------------------------------------------------------------------------------------
constexpr int kTileH = 32, kTileW = 64, kUpThreads = 256;

// Optional fused tail (EPI): StyledConv's NoiseInjection + FusedLeakyReLU after the Blur of an up-sampling layer
// (stylesdf_model.py:346, :459-466, :500-507) and the amax tracking the next modulated conv wants -- one pass less over
// the activation than blur -> noise_bias_act.
struct UpfirdnEpi {
    const float* noise;       // (noise_batch, out_h * out_w) or null
    const float* noise_w;     // device scalar
    const float* bias;        // (channels) or null
    float* out_amax;          // amax buffer (common.h) or null
    float alpha, scale;
    int channels, noise_batch;
};

template <int UP, int DN, int KH, int KW, bool EPI = false, typename T = float>
__global__ void __launch_bounds__(kUpThreads)
upfirdn2d_tiled_kernel(T* __restrict__ y, const T* __restrict__ x,
                       const float* __restrict__ k, UpfirdnGeom g, int tiles_x, int tiles_y, UpfirdnEpi ep = UpfirdnEpi{}) {
    constexpr int UH = (kTileH - 1) * DN + KH;           // rows of U needed by the tile
    constexpr int UW = (kTileW - 1) * DN + KW;
    constexpr int PITCH = (UW + 3) & ~3;                 // 16-B aligned rows for ds_read_b128
    __shared__ __attribute__((aligned(16))) float u[UH * PITCH];

    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x; bid /= tiles_x;
    const int ty_i = bid % tiles_y; bid /= tiles_y;
    const int64_t plane = bid;
    const int oy0 = ty_i * kTileH, ox0 = tx_i * kTileW;
    const T* xp = x + plane * (int64_t)g.in_h * g.in_w;

    // ---- stage U (zero-inserted, padded) ----
    const int gy0 = oy0 * DN - g.pad_y0, gx0 = ox0 * DN - g.pad_x0;   // U-tile origin in up-sampled coords
    // two phases so that all of a thread's loads are in flight together (a load -> LDS-store loop serialises on the HBM
    // latency: one outstanding load per thread)
    constexpr int NST = (UH * PITCH + kUpThreads - 1) / kUpThreads;
    float sv[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int e = threadIdx.x + it * kUpThreads;
        const int r = e / PITCH, c = e - r * PITCH;
        float v = 0.0f;
        if (e < UH * PITCH && c < UW) {
            const int sy = gy0 + r, sx = gx0 + c;
            if (sy >= 0 && sx >= 0) {
                int iy = sy, ix = sx;
                bool ok = true;
                if (UP > 1) {
                    iy = sy / UP; ix = sx / UP;
                    ok = (iy * UP == sy) && (ix * UP == sx);
                }
                if (ok && iy < g.in_h && ix < g.in_w) v = (float)xp[(int64_t)iy * g.in_w + ix];
            }
        }
        sv[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int e = threadIdx.x + it * kUpThreads;
        if (e < UH * PITCH) u[e] = sv[it];
    }
    // flipped taps, wave-uniform
    float kf[KH][KW];
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int b = 0; b < KW; ++b) kf[a][b] = k[(KH - 1 - a) * KW + (KW - 1 - b)];
    __syncthreads();

    // ---- compute: lane -> 4 adjacent outputs in x, rows ty and ty+16 ----
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    T* yp = y + plane * (int64_t)g.out_h * g.out_w;
    const bool vec_ok = (g.out_w & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) & (4 * sizeof(T) - 1)) == 0);
    float amax_l = 0.0f;
    [[maybe_unused]] const float e_nw = (EPI && ep.noise) ? ep.noise_w[0] : 0.0f;
    [[maybe_unused]] const float e_b = (EPI && ep.bias) ? ep.bias[(int)(plane % ep.channels)] : 0.0f;
    [[maybe_unused]] const float* e_nz = (EPI && ep.noise) ? ep.noise + (ep.noise_batch > 1 ? (plane / ep.channels) : 0) * (int64_t)g.out_h * g.out_w : nullptr;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int oyl = ty + rr * 16;
        const int oy = oy0 + oyl;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        constexpr int NIN = 3 * DN + KW;                 // inputs spanned by 4 outputs
        constexpr int NVEC = (NIN + 3) / 4;
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const float* row = u + (oyl * DN + ky) * PITCH + tx * 4 * DN;
            float in[NVEC * 4];
#pragma unroll
            for (int v = 0; v < NVEC; ++v) {
                // the last vector of the last lane may run past UW but stays inside PITCH*UH + slack:
                // PITCH >= UW and reads beyond are multiplied by nothing (indices < NIN only are used).
                const float4 q = *reinterpret_cast<const float4*>(row + 4 * v);
                in[4 * v + 0] = q.x; in[4 * v + 1] = q.y; in[4 * v + 2] = q.z; in[4 * v + 3] = q.w;
            }
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(in[j * DN + kx], kf[ky][kx], acc[j]);
        }
        if (oy < g.out_h) {
            const int ox = ox0 + tx * 4;
            T* dst = yp + (int64_t)oy * g.out_w + ox;
            if (EPI) {
                float nzv[4] = {0.f, 0.f, 0.f, 0.f};
                if (e_nz) {
                    const float* np_ = e_nz + (int64_t)oy * g.out_w + ox;
                    if ((g.out_w & 3) == 0 && ox + 3 < g.out_w && (reinterpret_cast<uintptr_t>(e_nz) & 15) == 0) {
                        const float4 q = *reinterpret_cast<const float4*>(np_);
                        nzv[0] = q.x; nzv[1] = q.y; nzv[2] = q.z; nzv[3] = q.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (ox + j < g.out_w) nzv[j] = np_[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ox + j < g.out_w) {
                        float v = acc[j];
                        if (e_nz) v = __fadd_rn(v, __fmul_rn(e_nw, nzv[j]));   // same rounding as noise_bias_act
                        if (ep.bias) v = v + e_b;
                        v = (v > 0.0f ? v : v * ep.alpha) * ep.scale;
                        acc[j] = v;
                        amax_l = fmaxf(amax_l, fabsf(v));
                    }
                }
            }
            store4<T>(dst, acc, vec_ok, g.out_w - ox);
        }
    }
    if (EPI && ep.out_amax) {                           // one atomic per block, spread over the buffer's slots
        __shared__ float part[kUpThreads / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax_l;
        __syncthreads();
        if (threadIdx.x == 0)
            atomic_max_nonneg(ep.out_amax + ((int)blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride,
                              fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
    }
}

// ---------------------------------------------------------------------------------------------
// Blur of the up-sampling layers (up 1, down 1, 4x4 taps -- 96 % of the op's bytes): 64 x 64 outputs per
// workgroup.  Against the general tiled kernel above: the 67 x 67 input patch is staged in groups of four (one index
// computation and one ds_write_b128 per four elements instead of per element -- the staging arithmetic was as many VALU
// instructions as the 16 FMAs per output), the halo is 9.7 % of the patch instead of 14.5 %, and a lane keeps four
// consecutive output rows in registers so that every pair of ds_read_b128 feeds up to 16 outputs x 4 taps (0.9 LDS reads per
// output instead of 2).  Tap order per output is unchanged (ky outer, kx inner): bit-identical results.
// (An LDS-free variant -- every lane loading its own 7 x 8 window with unaligned 16-byte loads, no barrier -- measured
// 5 % slower at 1024^2 and the same at 128^2; not kept.  DESIGN.md 4.2.)
// ---------------------------------------------------------------------------------------------
constexpr int kBlTile = 64, kBlU = kBlTile + 3, kBlPitch = 68, kBlGroups = kBlPitch / 4;   // 67 rows x 17 groups of 4

template <bool EPI, typename T = float>
__global__ void __launch_bounds__(kUpThreads)
blur44_kernel(T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ k, UpfirdnGeom g, int tiles_x,
              int tiles_y, UpfirdnEpi ep) {
    __shared__ __attribute__((aligned(16))) float u[kBlU * kBlPitch];
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x; bid /= tiles_x;
    const int ty_i = bid % tiles_y; bid /= tiles_y;
    const int64_t plane = bid;
    const int oy0 = ty_i * kBlTile, ox0 = tx_i * kBlTile;
    const T* __restrict__ xp = x + plane * (int64_t)g.in_h * g.in_w;
    const int gy0 = oy0 - g.pad_y0, gx0 = ox0 - g.pad_x0;

    // ---- stage the patch: all loads of a thread first (in flight together), then the LDS stores ----
    constexpr int NG = kBlU * kBlGroups, NST = (NG + kUpThreads - 1) / kUpThreads;      // 1139 groups, 5 per thread
    float4 sv[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int gi = threadIdx.x + it * kUpThreads;
        const int r = gi / kBlGroups, cg = gi - r * kBlGroups;
        const int sy = gy0 + r, sx = gx0 + 4 * cg;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (gi < NG && sy >= 0 && sy < g.in_h) {
            const T* __restrict__ row = xp + (int64_t)sy * g.in_w;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sx + j >= 0 && sx + j < g.in_w) v[j] = (float)row[sx + j];
        }
        sv[it] = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int gi = threadIdx.x + it * kUpThreads;
        if (gi < NG) *reinterpret_cast<float4*>(u + 4 * gi) = sv[it];           // group gi = row r, columns 4cg..4cg+3: offset r*68 + 4cg
    }
    float kf[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) kf[a][b] = k[(3 - a) * 4 + (3 - b)];       // flipped taps, wave-uniform
    __syncthreads();

    // ---- compute: lane -> 4 adjacent outputs in x, 4 consecutive rows ----
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[rr][j] = 0.0f;
#pragma unroll
    for (int ri = 0; ri < 7; ++ri) {
        const float* row = u + (4 * ty + ri) * kBlPitch + 4 * tx;
        const float4 q0 = *reinterpret_cast<const float4*>(row), q1 = *reinterpret_cast<const float4*>(row + 4);
        const float in[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int ky = ri - rr;
            if (ky >= 0 && ky < 4) {
#pragma unroll
                for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[rr][j] = fmaf(in[j + kx], kf[ky][kx], acc[rr][j]);
            }
        }
    }
    T* __restrict__ yp = y + plane * (int64_t)g.out_h * g.out_w;
    const bool vec_ok = (g.out_w & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) & (4 * sizeof(T) - 1)) == 0);
    float amax_l = 0.0f;
    [[maybe_unused]] const float e_nw = (EPI && ep.noise) ? ep.noise_w[0] : 0.0f;
    [[maybe_unused]] const float e_b = (EPI && ep.bias) ? ep.bias[(int)(plane % ep.channels)] : 0.0f;
    [[maybe_unused]] const float* e_nz = (EPI && ep.noise) ? ep.noise + (ep.noise_batch > 1 ? (plane / ep.channels) : 0) * (int64_t)g.out_h * g.out_w : nullptr;
    const int ox = ox0 + tx * 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int oy = oy0 + 4 * ty + rr;
        if (oy >= g.out_h) continue;
        T* dst = yp + (int64_t)oy * g.out_w + ox;
        if (EPI) {
            float nzv[4] = {0.f, 0.f, 0.f, 0.f};
            if (e_nz) {
                const float* np_ = e_nz + (int64_t)oy * g.out_w + ox;
                if ((g.out_w & 3) == 0 && ox + 3 < g.out_w && (reinterpret_cast<uintptr_t>(e_nz) & 15) == 0) {
                    const float4 q = *reinterpret_cast<const float4*>(np_);
                    nzv[0] = q.x; nzv[1] = q.y; nzv[2] = q.z; nzv[3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (ox + j < g.out_w) nzv[j] = np_[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ox + j < g.out_w) {
                    float v = acc[rr][j];
                    if (e_nz) v = __fadd_rn(v, __fmul_rn(e_nw, nzv[j]));       // same rounding as noise_bias_act
                    if (ep.bias) v = v + e_b;
                    v = (v > 0.0f ? v : v * ep.alpha) * ep.scale;
                    acc[rr][j] = v;
                    amax_l = fmaxf(amax_l, fabsf(v));
                }
            }
        }
        store4<T>(dst, acc[rr], vec_ok, g.out_w - ox);
    }
    if (EPI && ep.out_amax) {                           // one atomic per block, spread over the buffer's slots
        __shared__ float part[kUpThreads / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax_l;
        __syncthreads();
        if (threadIdx.x == 0)
            atomic_max_nonneg(ep.out_amax + ((int)blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride,
                              fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
    }
}

template <bool EPI, typename T = float>
static int launch_blur44(T* y, const T* x, const float* k, const UpfirdnGeom& g, int64_t planes, const UpfirdnEpi& ep,
                         hipStream_t st, const char* what) {
    const int tiles_x = (g.out_w + kBlTile - 1) / kBlTile, tiles_y = (g.out_h + kBlTile - 1) / kBlTile;
    const int64_t blocks = (int64_t)tiles_x * tiles_y * planes;
    if (blocks >= ((int64_t)1 << 31)) return fail(E3DGE_ERR_INVALID_ARG, "%s: grid too large", what);
    blur44_kernel<EPI, T><<<dim3((unsigned)blocks), dim3(kUpThreads), 0, st>>>(y, x, k, g, tiles_x, tiles_y, ep);
    return check_launch(what);
}

// ---------------------------------------------------------------------------------------------
// Generic kernel: one output per thread, direct gather (any up/down/pad/kernel <= 32x32).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kUpThreads)
upfirdn2d_generic_kernel(T* __restrict__ y, const T* __restrict__ x,
                         const float* __restrict__ k, UpfirdnGeom g, int64_t total) {
    for (int64_t o = (int64_t)blockIdx.x * kUpThreads + threadIdx.x; o < total;
         o += (int64_t)gridDim.x * kUpThreads) {
        const int ox = (int)(o % g.out_w);
        const int64_t t = o / g.out_w;
        const int oy = (int)(t % g.out_h);
        const int64_t plane = t / g.out_h;
        const T* xp = x + plane * (int64_t)g.in_h * g.in_w;
        const int by = oy * g.down_y - g.pad_y0, bx = ox * g.down_x - g.pad_x0;
        float acc = 0.0f;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int sy = by + ky;
            if (sy < 0) continue;
            const int iy = sy / g.up_y;
            if (iy * g.up_y != sy || iy >= g.in_h) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int sx = bx + kx;
                if (sx < 0) continue;
                const int ix = sx / g.up_x;
                if (ix * g.up_x != sx || ix >= g.in_w) continue;
                acc = fmaf((float)xp[(int64_t)iy * g.in_w + ix], k[(g.kh - 1 - ky) * g.kw + (g.kw - 1 - kx)], acc);
            }
        }
        y[o] = (T)acc;
    }
}

// fp64 form (ABI 12; upfirdn2d_kernel.cu:311 dispatches double too): the generic kernel in double arithmetic, taps in double -- what
// torch.autograd.gradcheck feeds the op.
__global__ void __launch_bounds__(kUpThreads)
upfirdn2d_generic_f64_kernel(double* __restrict__ y, const double* __restrict__ x, const double* __restrict__ k, UpfirdnGeom g, int64_t total) {
    for (int64_t o = (int64_t)blockIdx.x * kUpThreads + threadIdx.x; o < total; o += (int64_t)gridDim.x * kUpThreads) {
        const int ox = (int)(o % g.out_w);
        const int64_t t = o / g.out_w;
        const int oy = (int)(t % g.out_h);
        const int64_t plane = t / g.out_h;
        const double* xp = x + plane * (int64_t)g.in_h * g.in_w;
        const int by = oy * g.down_y - g.pad_y0, bx = ox * g.down_x - g.pad_x0;
        double acc = 0.0;
        for (int ky = 0; ky < g.kh; ++ky) {
            const int sy = by + ky;
            if (sy < 0) continue;
            const int iy = sy / g.up_y;
            if (iy * g.up_y != sy || iy >= g.in_h) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int sx = bx + kx;
                if (sx < 0) continue;
                const int ix = sx / g.up_x;
                if (ix * g.up_x != sx || ix >= g.in_w) continue;
                acc = fma(xp[(int64_t)iy * g.in_w + ix], k[(g.kh - 1 - ky) * g.kw + (g.kw - 1 - kx)], acc);
            }
        }
        y[o] = acc;
    }
}

template <int UP, int DN, int KH, int KW, typename T = float>
static int launch_tiled(T* y, const T* x, const float* k, const UpfirdnGeom& g,
                        int64_t major, hipStream_t st) {
    const int tiles_x = (g.out_w + kTileW - 1) / kTileW, tiles_y = (g.out_h + kTileH - 1) / kTileH;
    const int64_t blocks = (int64_t)tiles_x * tiles_y * major;
    if (blocks >= ((int64_t)1 << 31)) return fail(E3DGE_ERR_INVALID_ARG, "upfirdn2d: grid too large");
    upfirdn2d_tiled_kernel<UP, DN, KH, KW, false, T><<<dim3((unsigned)blocks), dim3(kUpThreads), 0, st>>>(
        y, x, k, g, tiles_x, tiles_y);
    return check_launch("upfirdn2d(tiled)");
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int e3dge_upfirdn2d_out_size(int in, int up, int down, int pad0, int pad1, int k) {
    if (in <= 0 || up <= 0 || down <= 0 || k <= 0) return -1;
    const int num = in * up + pad0 + pad1 - k + down;   // upfirdn2d_kernel.cu:237-240
    if (num <= 0) return -1;
    const int out = num / down;
    return out > 0 ? out : -1;
}

template <typename T>
static int upfirdn2d_any(T* y, const T* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                         int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, e3dge_stream_t stream) {
    E3DGE_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "upfirdn2d: up/down must be >= 1");
    E3DGE_REQUIRE(kh >= 1 && kw >= 1 && kh <= 32 && kw <= 32, "upfirdn2d: kernel %dx%d outside 1..32", kh, kw);
    E3DGE_REQUIRE(major >= 0 && in_h >= 1 && in_w >= 1, "upfirdn2d: bad input extent");
    const int out_h = e3dge_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh);
    const int out_w = e3dge_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw);
    E3DGE_REQUIRE(out_h > 0 && out_w > 0, "upfirdn2d: empty output (%d x %d)", out_h, out_w);
    if (major == 0) return E3DGE_OK;
    E3DGE_REQUIRE(x && y && k, "upfirdn2d: null pointer");
    E3DGE_REQUIRE(major * (int64_t)out_h * out_w < ((int64_t)1 << 40), "upfirdn2d: output too large");
    UpfirdnGeom g{in_h, in_w, out_h, out_w, up_x, up_y, down_x, down_y, pad_x0, pad_y0, kh, kw};
    hipStream_t st = as_stream(stream);
    const bool sq = (up_x == up_y) && (down_x == down_y) && kh == 4 && kw == 4;
    if (sq && up_x == 1 && down_x == 1) return launch_blur44<false, T>(y, x, k, g, major, UpfirdnEpi{}, st, "upfirdn2d(blur)");   // Blur
    if (sq && up_x == 2 && down_x == 1) return launch_tiled<2, 1, 4, 4, T>(y, x, k, g, major, st);   // Upsample
    if (sq && up_x == 1 && down_x == 2) return launch_tiled<1, 2, 4, 4, T>(y, x, k, g, major, st);   // its gradient / Downsample
    const int64_t total = major * (int64_t)out_h * out_w;
    int64_t blocks = (total + kUpThreads - 1) / kUpThreads;
    if (blocks > 16384) blocks = 16384;
    upfirdn2d_generic_kernel<T><<<dim3((unsigned)blocks), dim3(kUpThreads), 0, st>>>(y, x, k, g, total);
    return check_launch("upfirdn2d(generic)");
}

extern "C" int e3dge_upfirdn2d(float* y, const float* x, const float* k, int64_t major, int in_h,
                               int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                               int pad_x0, int pad_x1, int pad_y0, int pad_y1, e3dge_stream_t stream) {
    return upfirdn2d_any<float>(y, x, k, major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream);
}

extern "C" int e3dge_upfirdn2d_f16(void* y, const void* x, const float* k, int64_t major, int in_h,
                                   int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                   int pad_x0, int pad_x1, int pad_y0, int pad_y1, e3dge_stream_t stream) {
    return upfirdn2d_any<_Float16>(static_cast<_Float16*>(y), static_cast<const _Float16*>(x), k, major, in_h, in_w, kh, kw, up_x, up_y,
                                   down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream);
}

extern "C" int e3dge_upfirdn2d_f64(double* y, const double* x, const double* k, int64_t major, int in_h, int in_w, int kh, int kw, int up_x,
                                   int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, e3dge_stream_t stream) {
    E3DGE_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "upfirdn2d_f64: up/down must be >= 1");
    E3DGE_REQUIRE(kh >= 1 && kw >= 1 && kh <= 32 && kw <= 32, "upfirdn2d_f64: kernel %dx%d outside 1..32", kh, kw);
    E3DGE_REQUIRE(major >= 0 && in_h >= 1 && in_w >= 1, "upfirdn2d_f64: bad input extent");
    const int out_h = e3dge_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh);
    const int out_w = e3dge_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw);
    E3DGE_REQUIRE(out_h > 0 && out_w > 0, "upfirdn2d_f64: empty output (%d x %d)", out_h, out_w);
    if (major == 0) return E3DGE_OK;
    E3DGE_REQUIRE(x && y && k, "upfirdn2d_f64: null pointer");
    E3DGE_REQUIRE(major * (int64_t)out_h * out_w < ((int64_t)1 << 40), "upfirdn2d_f64: output too large");
    UpfirdnGeom g{in_h, in_w, out_h, out_w, up_x, up_y, down_x, down_y, pad_x0, pad_y0, kh, kw};
    const int64_t total = major * (int64_t)out_h * out_w;
    int64_t blocks = (total + kUpThreads - 1) / kUpThreads;
    if (blocks > 16384) blocks = 16384;
    upfirdn2d_generic_f64_kernel<<<dim3((unsigned)blocks), dim3(kUpThreads), 0, as_stream(stream)>>>(y, x, k, g, total);
    return check_launch("upfirdn2d(f64)");
}

extern "C" int e3dge_blur_noise_bias_act(float* y, const float* x, const float* k, const float* noise, const float* noise_weight,
                                         const float* bias, float alpha, float scale, int64_t batch, int64_t channels,
                                         int in_h, int in_w, int pad0, int pad1, int64_t noise_batch, float* out_amax,
                                         e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && channels >= 1 && in_h >= 1 && in_w >= 1, "blur_noise_bias_act: bad extent");
    const int out_h = e3dge_upfirdn2d_out_size(in_h, 1, 1, pad0, pad1, 4);
    const int out_w = e3dge_upfirdn2d_out_size(in_w, 1, 1, pad0, pad1, 4);
    E3DGE_REQUIRE(out_h > 0 && out_w > 0, "blur_noise_bias_act: empty output");
    if (batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(x && y && k, "blur_noise_bias_act: null pointer");
    E3DGE_REQUIRE(noise == nullptr || (noise_weight != nullptr && (noise_batch == 1 || noise_batch == batch)),
                  "blur_noise_bias_act: noise needs noise_weight and noise_batch in {1, batch}");
    UpfirdnGeom g{in_h, in_w, out_h, out_w, 1, 1, 1, 1, pad0, pad0, 4, 4};
    UpfirdnEpi ep{noise, noise_weight, bias, out_amax, alpha, scale, (int)channels, (int)noise_batch};
    return launch_blur44<true>(y, x, k, g, batch * channels, ep, as_stream(stream), "blur_noise_bias_act");
}
