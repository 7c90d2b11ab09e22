// Weight-stationary split-f16 chain of 256 x 256 layers (prototype of the third-generation renderer core).
//
// The first two f16x3 kernels keep a wave's ACTIVATIONS in registers and stream the weights through LDS: every wave reads
// every weight chunk (2 MiB of ds_read per layer and 128 points) and the instruction stream carries ~5 instructions per
// 16-cycle MFMA.  Here the roles are swapped: each of the 8 waves keeps its 32-feature slice of the layer's WEIGHTS in
// registers (128 VGPRs of packed hi / lo f16), the activations of 128 points live in LDS as four 32-point sets, and a wave
// reads a third of a ds_read_b128 per MFMA.  While a wave's matrix pipe works on set q, its VALU runs the FiLM + sine +
// split epilogue of set q - 1 from the other accumulators; the weights of the next layer replace the current ones k-step
// by k-step during the last set.
//
// Fragments (v_mfma_f32_16x16x32_f16, siren16.h): lane l, n = l & 15, q = l >> 4
//   A (16 x 32): row n, k = 8q + j        B (32 x 16): column n, k = 8q + j        C/D: column n, rows 4q + r
// A = weights (feature 32 wave + 16 ft + n), B = activations (point 16 pt + n of the set): D[feature][point].
// LDS image of a set: [hi | lo][k-step g = 0..7][pt 2][q 4][n 16] x 16 B = the 8 features 32 g + 8 q .. + 7 of point 16 pt + n.
#define E3DGE_16_HELPERS_ONLY
#include <string.h>
#include "siren16.h"
#include "decoder_common.h"

namespace e3dge {

#ifndef E3DGE_WS_PIN
#define E3DGE_WS_PIN 1               // 1 = MFMA / epilogue interleave pinned in source order, 0 = the compiler's placement
#endif
#ifndef E3DGE_WS_ABL
#define E3DGE_WS_ABL 0               // timing ablations (wrong results): 4 = no workgroup barriers, 8 = weights loaded once,
                                     // 16 = activation fragments read once per phase, 32 = epilogue without FiLM + sine
#endif
#define WS_SYNC() do { if (!(E3DGE_WS_ABL & 4)) __syncthreads(); } while (0)

constexpr int kWsThreads = 512;
constexpr int kWsSets = 4;
constexpr int kWsSetBytes = 32 * 1024, kWsHalfBytes = 16 * 1024;
constexpr int kWsXBytes = kWsSets * kWsSetBytes;
constexpr int kWsSteps = 8;          // k-steps of 32

struct WsRegs {
    u32x4 wh[2][kWsSteps], wl[2][kWsSteps];      // this wave's 32 output features (two 16-row tiles) x K = 256, hi and lo
};

// weight image: [layer][wave 8][g 8][ft 2][hi | lo][lane 64] x 16 B
__device__ __forceinline__ void ws_load_w(WsRegs& R, const u32x4* __restrict__ wimg, int layer, int wave, int lane, int g) {
    const u32x4* __restrict__ p = wimg + ((size_t)((layer * 8 + wave) * kWsSteps + g) * 4) * 64 + lane;
    R.wh[0][g] = p[0];
    R.wl[0][g] = p[64];
    R.wh[1][g] = p[128];
    R.wl[1][g] = p[192];
}

// split2 (siren_common.h) without inline asm, so that the scheduler can place its instructions: with the multiplier -1 in a
// register the compiler selects v_fma_mix_f32 itself (a literal -1 is folded into v_cvt_f32_f16 + v_sub_f32: one more op).
__device__ __forceinline__ HiLo split2s(float x0, float x1, float m1) {
    const fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const float l0 = fmaf((float)hh.x, m1, x0), l1 = fmaf((float)hh.y, m1, x1);
    HiLo p;
    p.h = __builtin_bit_cast(unsigned, hh);
    p.l = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
    return p;
}

struct WsAcc { f32x4v t[2][2]; };     // [ft][pt]

// One phase: the K = 256 contraction of one set into accM (DO_M) with the epilogue of the previous set from accE threaded
// through it (DO_E): tile e = (ft, pt) -- four features of one point per lane -- is finished during k-steps 2e, 2e + 1.
// LOAD_W: after k-step g the registers of that k-step take the next layer's weights.
// The activation fragments of k-step g + 1 are read one k-step ahead.  Per k-step the twelve MFMAs run as three passes over
// the four accumulators (hi*hi, lo*hi, hi*lo), so a dependent MFMA is four MFMAs behind its producer.
struct WsCarry;
__device__ __forceinline__ void ws_carry(WsCarry& C, const char* xm, const float* __restrict__ film_e, int wave, int lane);

struct WsCarry {                     // read ahead of the phase that uses them (before the barrier in between)
    u32x4 xh[2], xl[2];              // activation fragments of k-step 0, [pt]
};

template <bool DO_M, bool DO_E, bool LOAD_W>
__device__ __forceinline__ void ws_phase(WsRegs& R, WsAcc& accM, const WsAcc& accE, const char* xm, char* xe,
                                         const float* __restrict__ film_e, const u32x4* __restrict__ wimg, int next_layer,
                                         int wave, int lane, float m1, WsCarry& C, const char* xm_next,
                                         const float* __restrict__ film_next) {
    const int n = lane & 15, q = lane >> 4;
    const char* xr = xm + (q * 16 + n) * 16;
    char* xw = xe + (wave * 8 + (q >> 1)) * 256 + n * 16 + 8 * (q & 1);
    const float* fe = film_e + 32 * wave + 4 * q;
    f32x4v g4 = zero4(), b4 = zero4();
    if (DO_E && !(E3DGE_WS_ABL & 32)) {
        g4 = *reinterpret_cast<const f32x4v*>(fe);
        b4 = *reinterpret_cast<const f32x4v*>(fe + kWidth);
    }
    u32x4 xh[2][2], xl[2][2];        // [buffer][pt]
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) { xh[0][pt] = C.xh[pt]; xl[0][pt] = C.xl[pt]; }
    f32x4v arg = zero4(), rv = zero4();
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    fp16x2 hh0 = __builtin_amdgcn_cvt_pkrtz(0.f, 0.f), hh1 = hh0;
    float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    // Source order is the schedule when E3DGE_WS_PIN: a sched_barrier(0) behind every MFMA and behind the slice of the
    // carried epilogue that follows it (two VALU instructions), so that both waves of a SIMD -- which run this code in
    // lockstep after every barrier -- always have VALU work in the shadow of an MFMA instead of all stalling on the matrix
    // pipe together and then all running their epilogue together.
#if E3DGE_WS_PIN
#define WS_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define WS_SB() do { } while (0)
#endif
#pragma unroll
    for (int g = 0; g < kWsSteps; ++g) {
        const int cb = (E3DGE_WS_ABL & 16) ? 0 : (g & 1);
        const int e = g >> 1, eft = e >> 1, ept = e & 1;
        const bool even = (g & 1) == 0;
        // slice j (0..11) of the epilogue work of this k-step
        auto slice = [&](int j) {
            if (!DO_E) return;
            if (E3DGE_WS_ABL & 32) {
                if (!even && j == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = accE.t[eft][ept][i];
                }
            } else if (even) {
                if (j < 8) {                                   // value i = j >> 1: FiLM + period index | reduced argument
                    const int i = j >> 1;
                    if ((j & 1) == 0) {
                        arg[i] = fmaf(g4[i], accE.t[eft][ept][i], b4[i]);
                        rv[i] = rintf(arg[i] * 0.15915494f);
                    } else {
                        rv[i] = fmaf(arg[i], 0.15915494f, -rv[i]);
                        rv[i] = fmaf(arg[i], 6.4206382e-09f, rv[i]);
                    }
                }
            } else {
                if (j < 2) { v[2 * j] = __builtin_amdgcn_sinf(rv[2 * j]); v[2 * j + 1] = __builtin_amdgcn_sinf(rv[2 * j + 1]); }
            }
            if (!even) {
                if (j == 2) { hh0 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]); hh1 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]); }
                if (j == 3) { l0 = fmaf((float)hh0.x, m1, v[0]); l1 = fmaf((float)hh0.y, m1, v[1]); }
                if (j == 4) { l2 = fmaf((float)hh1.x, m1, v[2]); l3 = fmaf((float)hh1.y, m1, v[3]); }
                if (j == 5) {
                    const fp16x2 ll0 = __builtin_amdgcn_cvt_pkrtz(l0, l1), ll1 = __builtin_amdgcn_cvt_pkrtz(l2, l3);
                    char* o = xw + ept * 1024 + eft * 512;
                    *reinterpret_cast<uint2*>(o) = make_uint2(__builtin_bit_cast(unsigned, hh0), __builtin_bit_cast(unsigned, hh1));
                    *reinterpret_cast<uint2*>(o + kWsHalfBytes) = make_uint2(__builtin_bit_cast(unsigned, ll0), __builtin_bit_cast(unsigned, ll1));
                }
                if (j == 6 && e == 1 && !(E3DGE_WS_ABL & 32)) {    // FiLM rows of the second feature tile
                    g4 = *reinterpret_cast<const f32x4v*>(fe + 16);
                    b4 = *reinterpret_cast<const f32x4v*>(fe + kWidth + 16);
                }
            }
        };
        WS_SB();
        if (DO_M && g + 1 < kWsSteps && !(E3DGE_WS_ABL & 16)) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                xh[(g + 1) & 1][pt] = *reinterpret_cast<const u32x4*>(xr + (g + 1) * 2048 + pt * 1024);
                xl[(g + 1) & 1][pt] = *reinterpret_cast<const u32x4*>(xr + (g + 1) * 2048 + pt * 1024 + kWsHalfBytes);
            }
        }
        WS_SB();
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    if (DO_M) {
                        const u32x4 wa = (pass == 1) ? R.wl[ft][g] : R.wh[ft][g];
                        const u32x4 xb = (pass == 2) ? xl[cb][pt] : xh[cb][pt];
                        accM.t[ft][pt] = mfma16x16(wa, xb, (g == 0 && pass == 0) ? zero4() : accM.t[ft][pt]);
                    }
                    WS_SB();
                    slice(pass * 4 + ft * 2 + pt);
                    WS_SB();
                }
        if (LOAD_W && !(E3DGE_WS_ABL & 8)) ws_load_w(R, wimg, next_layer, wave, lane, g);
        if (g == kWsSteps - 1) ws_carry(C, xm_next, film_next, wave, lane);
    }
#undef WS_SB
}

__device__ __forceinline__ void ws_carry(WsCarry& C, const char* xm, const float* __restrict__ film_e, int wave, int lane) {
    const int n = lane & 15, q = lane >> 4;
    const char* xr = xm + (q * 16 + n) * 16;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        C.xh[pt] = *reinterpret_cast<const u32x4*>(xr + pt * 1024);
        C.xl[pt] = *reinterpret_cast<const u32x4*>(xr + pt * 1024 + kWsHalfBytes);
    }
}

#ifdef E3DGE_EXPERIMENTAL     // the chain study of DESIGN.md 4.1d (include/e3dge_hip_experimental.h); e3dge_ws_linear below is the product use
// y = layer_{n-1}(... layer_0(x)), layer(x) = sin(gamma * (W x) + beta); film = [layer][gamma | beta][256] with the weights'
// factor 128 already divided out of gamma.  One workgroup per CU, groups of 128 points.
__global__ void __launch_bounds__(kWsThreads)
ws_chain_kernel(const u32x4* __restrict__ wimg, const float* __restrict__ film, const float* __restrict__ x0, float* __restrict__ y,
                int n_layers, int n_groups, long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const X = smem;
    float* const film_s = reinterpret_cast<float*>(smem + kWsXBytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < n_layers * 2 * kWidth; i += kWsThreads) film_s[i] = film[i];
    float m1 = -1.0f;
    asm volatile("" : "+v"(m1));
    WsRegs R;
#pragma unroll
    for (int g = 0; g < kWsSteps; ++g) ws_load_w(R, wimg, 0, wave, lane, g);
    WsAcc acc0, acc1;

    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        // stage the group's input: split to (hi, lo) in the LDS image; a thread converts the 8-feature slots (g, q) = tid >> 5 ...
        // of point tid & 31 of every set
#pragma unroll
        for (int st = 0; st < kWsSets; ++st)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pnt = tid & 31, slot = (tid >> 5) * 2 + i;             // slot = g * 4 + q
                const float* src = x0 + ((size_t)grp * 128 + st * 32 + pnt) * kWidth + slot * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                const HiLo p0 = split2(a[0], a[1]), p1 = split2(a[2], a[3]), p2 = split2(b[0], b[1]), p3 = split2(b[2], b[3]);
                char* o = X + st * kWsSetBytes + ((((slot >> 2) * 2 + (pnt >> 4)) * 4 + (slot & 3)) * 16 + (pnt & 15)) * 16;
                *reinterpret_cast<u32x4*>(o) = u32x4{p0.h, p1.h, p2.h, p3.h};
                *reinterpret_cast<u32x4*>(o + kWsHalfBytes) = u32x4{p0.l, p1.l, p2.l, p3.l};
            }
        WS_SYNC();
        WsCarry C;
        const char* const X0 = X, * const X1 = X + kWsSetBytes, * const X2 = X + 2 * kWsSetBytes, * const X3 = X + 3 * kWsSetBytes;
        ws_carry(C, X0, film_s, wave, lane);
        // layer 0: the first phase has no epilogue to carry
        {
            const float* f0 = film_s;
            const int nl = (n_layers > 1) ? 1 : 0;
            ws_phase<true, false, false>(R, acc0, acc1, X0, X, f0, wimg, nl, wave, lane, m1, C, X1, f0);
            WS_SYNC();
            ws_phase<true, true, false>(R, acc1, acc0, X1, X + 0 * kWsSetBytes, f0, wimg, nl, wave, lane, m1, C, X2, f0);
            WS_SYNC();
            ws_phase<true, true, false>(R, acc0, acc1, X2, X + 1 * kWsSetBytes, f0, wimg, nl, wave, lane, m1, C, X3, f0);
            WS_SYNC();
            ws_phase<true, true, true>(R, acc1, acc0, X3, X + 2 * kWsSetBytes, f0, wimg, nl, wave, lane, m1, C, X0, f0);
            WS_SYNC();
        }
        const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
#pragma unroll 1
        for (int L = 1; L < n_layers; ++L) {
            const float* fp = film_s + (L - 1) * 2 * kWidth;
            const float* fc = film_s + L * 2 * kWidth;
            const int nl = (L + 1 < n_layers) ? L + 1 : 0;
            ws_phase<true, true, false>(R, acc0, acc1, X0, X + 3 * kWsSetBytes, fp, wimg, nl, wave, lane, m1, C, X1, fc);
            WS_SYNC();
            ws_phase<true, true, false>(R, acc1, acc0, X1, X + 0 * kWsSetBytes, fc, wimg, nl, wave, lane, m1, C, X2, fc);
            WS_SYNC();
            ws_phase<true, true, false>(R, acc0, acc1, X2, X + 1 * kWsSetBytes, fc, wimg, nl, wave, lane, m1, C, X3, fc);
            WS_SYNC();
            ws_phase<true, true, true>(R, acc1, acc0, X3, X + 2 * kWsSetBytes, fc, wimg, nl, wave, lane, m1, C, X0, fc);
            WS_SYNC();
        }
        if (dbg && tid == 0 && grp < 256) { dbg[grp] = __builtin_readcyclecounter() - t0; dbg[256 + grp] = wall_clock64() - r0; }
        ws_phase<false, true, false>(R, acc0, acc1, X, X + 3 * kWsSetBytes, film_s + (n_layers - 1) * 2 * kWidth, wimg, 0, wave, lane, m1, C, X0, film_s);
        WS_SYNC();
#pragma unroll
        for (int st = 0; st < kWsSets; ++st)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pnt = tid & 31, slot = (tid >> 5) * 2 + i;
                const char* o = X + st * kWsSetBytes + ((((slot >> 2) * 2 + (pnt >> 4)) * 4 + (slot & 3)) * 16 + (pnt & 15)) * 16;
                const u32x4 hi = *reinterpret_cast<const u32x4*>(o), lo = *reinterpret_cast<const u32x4*>(o + kWsHalfBytes);
                float* dst = y + ((size_t)grp * 128 + st * 32 + pnt) * kWidth + slot * 8;
                f32x4 a, b;
                a[0] = f16lo(hi[0]) + f16lo(lo[0]); a[1] = f16hi(hi[0]) + f16hi(lo[0]);
                a[2] = f16lo(hi[1]) + f16lo(lo[1]); a[3] = f16hi(hi[1]) + f16hi(lo[1]);
                b[0] = f16lo(hi[2]) + f16lo(lo[2]); b[1] = f16hi(hi[2]) + f16hi(lo[2]);
                b[2] = f16lo(hi[3]) + f16lo(lo[3]); b[3] = f16hi(hi[3]) + f16hi(lo[3]);
                *reinterpret_cast<f32x4*>(dst) = a;
                *reinterpret_cast<f32x4*>(dst + 4) = b;
            }
        WS_SYNC();
    }
}
#endif  // E3DGE_EXPERIMENTAL

}  // namespace e3dge

namespace e3dge {
// =====================================================================================================================
// One 256 x 256 linear layer on rows of a matrix, weight-stationary (the layers of Fuse_sft_MLP, local_query.py):
//     y[row, off_y + f] = post( sum_k W[f][k] pre(x[row, off_x + k]) + bias[f] + colw[f] pre(m[row]) + r1[row, f] + r2[row, f] )
// pre = relu or identity; post = identity | leaky relu | the SFT fuse  D + w (D S + v)  with D = r1, S = r2.
// A workgroup keeps the whole weight image in registers for its lifetime (8 waves x 32 output features, as ws_chain_kernel)
// and walks groups of 64 rows: the rows of group g + 1 are fetched (global -> registers) before the contraction of group g
// and converted / written to the other LDS buffer after it.  Operand scale: 2^(141 - eb) with eb from the input tensor's amax
// buffer (decoder_common.h), so any magnitude works; the output's amax is tracked for the next layer.
// =====================================================================================================================
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };      // 16-byte access at 4-byte alignment (row pitch 513 floats)

struct WsLinK {
    const u32x4* wimg; const float* x; const float* amax_in; const float* bias; const float* colw; const float* m;
    const float* r1; const float* r2; float* y; float* amax_out;
    int64_t n_rows;
    int ld_x, off_x, ld_m, off_m, ld_r1, off_r1, ld_r2, off_r2, ld_y, off_y;
    int pre_relu, post;
    float slope, w_fuse;
};
constexpr int kWlRows = 64, kWlBufBytes = 2 * kWsSetBytes;

__global__ void __launch_bounds__(kWsThreads) ws_linear_kernel(const WsLinK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + 2 * kWlBufBytes);           // bias[256], colw[256]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, q = lane >> 4;
    const int64_t n_groups = (a.n_rows + kWlRows - 1) / kWlRows;
    if ((int64_t)blockIdx.x >= n_groups) return;
    const unsigned eb = a.amax_in ? scale_exponent(amax_read(a.amax_in, lane)) : 127u + 14u;
    const float in_scale = __uint_as_float((268u - eb) << 23), oscale = __uint_as_float((eb - 21u) << 23);
    if (tid < kWidth) { tab[tid] = a.bias ? a.bias[tid] : 0.0f; tab[kWidth + tid] = a.colw ? a.colw[tid] : 0.0f; }
    WsRegs R;
#pragma unroll
    for (int g = 0; g < kWsSteps; ++g) ws_load_w(R, a.wimg, 0, wave, lane, g);

    // staging: this lane's 8 columns 32 wave + 8 q .. + 7 of the rows 16 t + n (t = 0..3) of a group
    // (two of the four row tiles at a time -- half = the LDS set they belong to: 16 staging registers live across a contraction)
    f32x4 sa[2], sb[2];
    auto stage_load = [&](int64_t grp, int half) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t row = grp * kWlRows + 32 * half + 16 * t + n;
            if (row < a.n_rows) {
                const float* p = a.x + row * a.ld_x + a.off_x + 32 * wave + 8 * q;
                const F4U u0 = *reinterpret_cast<const F4U*>(p), u1 = *reinterpret_cast<const F4U*>(p + 4);
                sa[t] = f32x4{u0.v[0], u0.v[1], u0.v[2], u0.v[3]};
                sb[t] = f32x4{u1.v[0], u1.v[1], u1.v[2], u1.v[3]};
            } else {
                sa[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                sb[t] = sa[t];
            }
        }
    };
    auto stage_write = [&](int buf, int half) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 u = sa[t], w = sb[t];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.pre_relu) { u[i] = fmaxf(u[i], 0.0f); w[i] = fmaxf(w[i], 0.0f); }
                u[i] *= in_scale; w[i] *= in_scale;
            }
            const HiLo p0 = split2(u[0], u[1]), p1 = split2(u[2], u[3]), p2 = split2(w[0], w[1]), p3 = split2(w[2], w[3]);
            char* o = smem + buf * kWlBufBytes + half * kWsSetBytes + (((wave * 2 + t) * 4 + q) * 16 + n) * 16;
            *reinterpret_cast<u32x4*>(o) = u32x4{p0.h, p1.h, p2.h, p3.h};
            *reinterpret_cast<u32x4*>(o + kWsHalfBytes) = u32x4{p0.l, p1.l, p2.l, p3.l};
        }
    };
    float amax_l = 0.0f;
    stage_load(blockIdx.x, 0);
    stage_write(0, 0);
    stage_load(blockIdx.x, 1);
    stage_write(0, 1);
    __syncthreads();
    int buf = 0;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x, buf ^= 1) {
        const bool has_next = grp + gridDim.x < n_groups;
#pragma unroll 1
        for (int st = 0; st < 2; ++st) {
            if (has_next) stage_load(grp + gridDim.x, st);
            const char* xr = smem + buf * kWlBufBytes + st * kWsSetBytes + (q * 16 + n) * 16;
            WsAcc acc;
            u32x4 xh[2][2], xl[2][2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                xh[0][pt] = *reinterpret_cast<const u32x4*>(xr + pt * 1024);
                xl[0][pt] = *reinterpret_cast<const u32x4*>(xr + pt * 1024 + kWsHalfBytes);
            }
#pragma unroll
            for (int g = 0; g < kWsSteps; ++g) {
                if (g + 1 < kWsSteps) {
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        xh[(g + 1) & 1][pt] = *reinterpret_cast<const u32x4*>(xr + (g + 1) * 2048 + pt * 1024);
                        xl[(g + 1) & 1][pt] = *reinterpret_cast<const u32x4*>(xr + (g + 1) * 2048 + pt * 1024 + kWsHalfBytes);
                    }
                }
#pragma unroll
                for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) {
                            const u32x4 wa = (pass == 1) ? R.wl[ft][g] : R.wh[ft][g];
                            const u32x4 xb = (pass == 2) ? xl[g & 1][pt] : xh[g & 1][pt];
                            acc.t[ft][pt] = mfma16x16(wa, xb, (g == 0 && pass == 0) ? zero4() : acc.t[ft][pt]);
                        }
            }
            // ---- post: this lane's features 32 wave + 16 ft + 4 q .. + 3 of the rows 16 pt + n of the set ----
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int64_t row = grp * kWlRows + st * 32 + 16 * pt + n;
                const bool ok = row < a.n_rows;
                float mv = 0.0f;
                if (a.m && ok) { mv = a.m[row * a.ld_m + a.off_m]; if (a.pre_relu) mv = fmaxf(mv, 0.0f); }
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    const int f0 = 32 * wave + 16 * ft + 4 * q;
                    const f32x4v b4 = *reinterpret_cast<const f32x4v*>(tab + f0), c4 = *reinterpret_cast<const f32x4v*>(tab + kWidth + f0);
                    float v[4], d1[4] = {0.f, 0.f, 0.f, 0.f}, d2[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.r1 && ok) { const F4U u = *reinterpret_cast<const F4U*>(a.r1 + row * a.ld_r1 + a.off_r1 + f0); d1[0] = u.v[0]; d1[1] = u.v[1]; d1[2] = u.v[2]; d1[3] = u.v[3]; }
                    if (a.r2 && ok) { const F4U u = *reinterpret_cast<const F4U*>(a.r2 + row * a.ld_r2 + a.off_r2 + f0); d2[0] = u.v[0]; d2[1] = u.v[1]; d2[2] = u.v[2]; d2[3] = u.v[3]; }
                    F4U out;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = fmaf(acc.t[ft][pt][i], oscale, b4[i]);
                        t = fmaf(c4[i], mv, t);
                        if (a.post == 2) {
                            t = fmaf(a.w_fuse, fmaf(d1[i], d2[i], t), d1[i]);                  // D + w (D S + shift)
                        } else {
                            t = (t + d1[i]) + d2[i];
                            if (a.post == 1) t = fmaxf(t, t * a.slope);                         // leaky relu, 0 <= slope <= 1
                        }
                        out.v[i] = t;
                        if (ok) amax_l = fmaxf(amax_l, fabsf(t));
                    }
                    if (ok) *reinterpret_cast<F4U*>(a.y + row * a.ld_y + a.off_y + f0) = out;
                }
            }
            if (has_next) stage_write(buf ^ 1, st);
        }
        __syncthreads();
    }
    if (a.amax_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if (lane == 0) atomic_max_nonneg(a.amax_out + (((int)blockIdx.x * 8 + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_l);
    }
}
}  // namespace e3dge

namespace e3dge {
// weight image of ws_chain_kernel from fp32 weights (n_layers, 256 out, 256 in): word `wd` (two f16) of lane l (n = l & 15,
// q = l >> 4) of [layer][wave][g][ft][hi | lo] holds 128 W[32 wave + 16 ft + n][32 g + 8 q + 2 wd], + 1 -- hi = round-toward-zero
// f16 of the scaled value, lo = f16 of the remainder (as split2)
__global__ void __launch_bounds__(256) ws_pack_kernel(unsigned* __restrict__ img, const float* __restrict__ w, int64_t n_words) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_words; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        const int wd = r & 3; r >>= 2;
        const int lane = r & 63; r >>= 6;
        const int hl = r & 1; r >>= 1;
        const int ft = r & 1; r >>= 1;
        const int g = r & 7; r >>= 3;
        const int wave = r & 7; r >>= 3;
        const int layer = (int)r;
        const int f = 32 * wave + 16 * ft + (lane & 15), k = 32 * g + 8 * (lane >> 4) + 2 * wd;
        const float* src = w + ((int64_t)layer * kWidth + f) * kWidth + k;
        const HiLo p = split2(src[0] * kW16Scale, src[1] * kW16Scale);
        img[e] = hl ? p.l : p.h;
    }
}
}  // namespace e3dge

extern "C" int64_t e3dge_ws_image_bytes(int n_layers) { return (int64_t)n_layers * 8 * 8 * 2 * 2 * 64 * 16; }

extern "C" int e3dge_ws_pack(void* wimg, const float* weights, int n_layers, e3dge_stream_t stream) {
    using namespace e3dge;
    E3DGE_REQUIRE(wimg && weights && n_layers >= 1 && n_layers <= 8, "ws_pack: null pointer or n_layers not in 1..8");
    const int64_t n_words = e3dge_ws_image_bytes(n_layers) / 4;
    ws_pack_kernel<<<dim3(1024), dim3(256), 0, as_stream(stream)>>>(reinterpret_cast<unsigned*>(wimg), weights, n_words);
    return check_launch("ws_pack");
}

#ifdef E3DGE_EXPERIMENTAL
extern "C" int e3dge_ws_chain(const void* wimg, const float* film, const float* x0, float* y, int n_layers, int n_points, int grid,
                              long long* dbg, e3dge_stream_t stream) {
    using namespace e3dge;
    E3DGE_REQUIRE(wimg && film && x0 && y, "ws_chain: null pointer");
    E3DGE_REQUIRE(n_points > 0 && n_points % 128 == 0 && n_layers >= 1 && n_layers <= 8, "ws_chain: n_points must be a positive multiple of 128, 1 <= n_layers <= 8");
    if (grid <= 0) grid = 256;
    const int lds = kWsXBytes + n_layers * 2 * kWidth * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ws_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(ws_chain): %s", hipGetErrorString(e));
    ws_chain_kernel<<<dim3(grid), dim3(kWsThreads), lds, as_stream(stream)>>>(reinterpret_cast<const u32x4*>(wimg), film, x0, y, n_layers,
                                                                             n_points / 128, dbg);
    return check_launch("ws_chain");
}
#endif

static_assert(sizeof(E3dgeWsLinear) == sizeof(e3dge::WsLinK), "E3dgeWsLinear mirrors WsLinK");

extern "C" int e3dge_ws_linear(const E3dgeWsLinear* args, e3dge_stream_t stream) {
    using namespace e3dge;
    E3DGE_REQUIRE(args && args->wimg && args->x && args->y, "ws_linear: null pointer");
    E3DGE_REQUIRE(args->n_rows >= 0 && args->ld_x >= args->off_x + 256 && args->ld_y >= args->off_y + 256 && args->off_x >= 0 && args->off_y >= 0,
                  "ws_linear: a 256-column block must fit the row pitch (x: ld %d off %d, y: ld %d off %d)", args->ld_x, args->off_x, args->ld_y, args->off_y);
    E3DGE_REQUIRE(args->post >= 0 && args->post <= 2 && (args->post != 2 || (args->r1 && args->r2)), "ws_linear: post must be 0, 1 or 2 (2 needs r1 = D and r2 = S)");
    E3DGE_REQUIRE((!args->r1 || args->ld_r1 >= args->off_r1 + 256) && (!args->r2 || args->ld_r2 >= args->off_r2 + 256) && (!args->m || args->ld_m > args->off_m),
                  "ws_linear: residual / column operand does not fit its row pitch");
    E3DGE_REQUIRE(!args->colw == !args->m, "ws_linear: colw and m come together");
    E3DGE_REQUIRE(args->post != 1 || (args->slope >= 0.0f && args->slope <= 1.0f), "ws_linear: leaky slope must be in [0, 1]");
    if (args->n_rows == 0) return E3DGE_OK;
    WsLinK k;
    memcpy(&k, args, sizeof(k));
    const int lds = 2 * kWlBufBytes + 2 * kWidth * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ws_linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(ws_linear): %s", hipGetErrorString(e));
    const int64_t n_groups = (k.n_rows + kWlRows - 1) / kWlRows;
    const int grid = n_groups < 256 ? (int)n_groups : 256;
    ws_linear_kernel<<<dim3((unsigned)grid), dim3(kWsThreads), lds, as_stream(stream)>>>(k);
    return check_launch("ws_linear");
}
