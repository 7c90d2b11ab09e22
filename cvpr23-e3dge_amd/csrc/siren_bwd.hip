// Backward of the FiLM-SIREN MLP with respect to the FiLM parameters (gamma, beta of all 9 layers), i.e. the path
// the encoder's gradient takes: styles -> (gamma, beta) -> every layer (generator weights are frozen, trainer.py:1568).
//
// Reference being replaced: autograd through SirenGenerator.forward (project/utils/volume_renderer.py:168-264) as
// train_ae.py drives it (loss.backward(), trainer.py:728).  Inputs are the pre-sine arguments the forward kernel saved
// (E3dgeRenderArgs.save_args) and the gradient w.r.t. the per-point network outputs [rgb3, sdf1, feat256].
//
// Same machine as the forward kernel: one wave keeps its 32 points' 256-wide gradient in registers and chains
//     dh_{L-1} = W_L^T (gamma_L * dh_L * cos(arg_L))
// through fp32 MFMAs (gradients do not fit f16's range) against the TRANSPOSED weight image streamed through LDS;
// the C/D fragment of one GEMM is again the B operand of the next.  d(gamma), d(beta) are sums over points: each wave
// reduces its 32 lanes with a transpose-reduce and accumulates into its own slice of a partial buffer (no two waves
// share an address -> deterministic), which a small kernel then folds; a last kernel maps d(film) to d(styles).
#include "siren_common.h"

namespace e3dge {

struct SirenBwdK {
    const float* packed;
    const float* film;       // (batch, 9, 2, 256) gamma, beta as e3dge_film_params produced them
    const float* args;       // (batch, n_pts, 9, 256) saved pre-sine arguments
    const float* d_feat;     // (batch, n_pts, 256) or null
    const float* d_rgb;      // (batch, n_pts, 3) or null
    const float* d_sdf;      // (batch, n_pts) or null
    // render mode: d_feat of a point is weights[p] * d_featmap[ray(p)] (never materialised per point)
    const float* d_featmap;  // (batch * rays, 256) or null
    const float* weights;    // (batch, n_pts) compositing weights of the forward launch
    int samples;             // points per ray
    // second-order (eikonal) streams, both (batch, n_pts, 8, 256), or null
    const float* tang;       // tangent arguments ta_l of the tangent kernel
    const float* rsave;      // r_l = d sdf / d h_l of the sdf-chain kernel
    int precision;           // E3DGE_PREC_F32 / E3DGE_PREC_F16X3
    float* partials;         // (grid, 9, 2, 256): every workgroup writes its whole slice
    // optional: gradient w.r.t. the query points, dL/dx = s W_0^T (gamma_0 * adj(a_0))  (batch, n_pts, 3), or null
    float* d_pts;
    float box_scale;
    // optional second-pass texture FiLM h8' = (alpha + 1) h8 + beta in front of the view layer (:217-220):
    const float* tex_alpha;  // (batch, n_pts, 256) the forward launch's alpha, or null
    float* d_tex_alpha;      // (batch, n_pts, 256) out: dL/dalpha = dh8' * h8
    float* d_tex_beta;       // (batch, n_pts, 256) out: dL/dbeta = dh8'
    long long n_pts;
    int batch, subtiles_per_wg, wgs_per_img;
};

constexpr int kBwdLdsW = 0;
constexpr int kBwdLdsFilm = kBwdLdsW + kNBuf * kChunkFloats;     // [9][3][256] gamma, beta, 1/gamma
constexpr int kBwdLdsHead = kBwdLdsFilm + 9 * 3 * kWidth;        // w_sigma[256], w_rgb[3][256]
constexpr int kBwdLdsAcc = kBwdLdsHead + 4 * kWidth;              // [9][2][256] this workgroup's d(gamma), d(beta)
constexpr int kBwdLdsSlot = kBwdLdsAcc + 9 * 2 * kWidth;          // [2 parity][4 waves][2][32] per-tile wave sums
constexpr int kBwdLdsSlot8 = kBwdLdsSlot + 2 * 4 * 2 * 32;        // [8 tiles][4 waves][2][32] view-layer wave sums
constexpr int kBwdLdsW0 = kBwdLdsSlot8 + kNT * 4 * 2 * 32;         // [3][256] first-layer weights, column-major (d_pts)
constexpr int kBwdLdsFloats = kBwdLdsW0 + 3 * kWidth;
constexpr int kBwdLdsBytes = kBwdLdsFloats * 4;

// Sum over the 32 lanes of a half for 8 per-lane values, entirely in the VALU (no LDS round trips: one wave per SIMD,
// so any lgkmcnt stall would idle the MFMA pipe).  v_permlane16_swap exchanges the odd 16-lane rows of one register
// with the even rows of another, so x[i] + x[4+i] after the swap is the two-row sum of value i in even rows and of
// value 4+i in odd rows; four DPP row rotations finish the 16 lanes.  Result: q[i] (i < 4) on every lane = total of
// value i (lanes 0-15 of the half) or value 4+i (lanes 16-31).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void reduce8_over_lanes(const float (&v)[8], float (&q)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = v[i], hi = v[4 + i];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
        float x = lo + hi;
        x += dpp_f32<0x128>(x);    // row_ror:8
        x += dpp_f32<0x124>(x);    // row_ror:4
        x += dpp_f32<0x122>(x);    // row_ror:2
        x += dpp_f32<0x121>(x);    // row_ror:1
        q[i] = x;
    }
}

// sin and cos of a saved argument with one shared range reduction
__device__ __forceinline__ void sincos_hw_f32(float x, float& sn, float& cs) {
    const float r = revolutions_f32(x);
    sn = __builtin_amdgcn_sinf(r);
    cs = __builtin_amdgcn_cosf(r);
}

// Split-f16 operand of a backward-type GEMM.  The B operand here is a gradient: no bounded range, so each point (a
// column of B, one lane pair) gets its own power-of-two scale that brings the largest of its 256 values into [1, 2)
// before the (hi, lo) split; the GEMM is linear per column, so D / (128 * scale) is the unscaled result exactly.
// Relative to the column's largest element the representation is good to ~2^-24, the same as the forward kernel's
// activations, and the accumulation is fp32 in both.  `src` = the 256 values of the point held by this lane (standard
// layout); returns 1 / (kW16Scale * scale) for the epilogue.
// `m` = max |value| over this lane's 128 values (the producers keep it as a running max).
__device__ __forceinline__ float scale_split(const f32x16 (&src)[kNT], u32x4 (&dH)[2 * kNT], u32x4 (&dL)[2 * kNT], float m) {
    m = fmaxf(m, xhalf(m));
    const unsigned e = min((__float_as_uint(m) >> 23) & 255u, 254u);    // m in [2^(e-127), 2^(e-126)); inf/nan: scale 0 -> NaN out
    const float sc = __uint_as_float((254u - e) << 23);                 // m * sc in [1, 2)   (m == 0: sc = 2^127, harmless)
    const float inv = __uint_as_float((e > 8u ? e - 7u : 1u) << 23);    // 1 / (128 * sc) = 2^(e-134)
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            SPLIT2_TO(src[t][r] * sc, src[t][r + 1] * sc, dH[2 * t + (r >> 3)][(r & 7) >> 1], dL[2 * t + (r >> 3)][(r & 7) >> 1]);
    return inv;
}

__device__ __forceinline__ float scale_split(const f32x16 (&src)[kNT], u32x4 (&dH)[2 * kNT], u32x4 (&dL)[2 * kNT]) {
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(src[t][r]));
    return scale_split(src, dH, dL, m);
}

// EIK = false: gradient of a loss that reaches the network through (feat, rgb, sdf).
// EIK = true : additionally the loss depends on the eikonal term e = d sdf / d x (get_eikonal_term :796-802, i.e. the
//   reference's create_graph=True double backward).  With v = dL/de held fixed, dL = d(v.e) and v.e is the tangent of
//   sdf along v, so the extra terms come from differentiating the tangent pass: with r_l = d sdf / d h_l (saved by the
//   sdf-chain kernel) and ta_l = gamma_l * (W_l th_{l-1}) the tangent argument (saved by the tangent kernel),
//       adj(a_l)     = cos(a_l) adj(h_l) - sin(a_l) ta_l r_l
//       adj(gamma_l) = adj(a_l) z_l + (ta_l / gamma_l) cos(a_l) r_l ,   adj(beta_l) = adj(a_l)
//   and the same transposed chain carries adj(h) downwards.
// F16 = true: the eight GEMMs run as block-scaled split-f16 contractions (scale_split above) on the f16 matrix pipe.
// TEX = true: the forward pass applied the per-point texture FiLM between the sdf head and the view layer; GEMM 0's result
//   is then dL/dh8', which yields dL/dalpha = dh8' * sin(a_7), dL/dbeta = dh8' (stored per point) and continues down the
//   chain as (alpha + 1) * dh8'.  Only built without EIK (the eikonal losses live on the first pass).
// DPTS = true: also write dL/dx of every query point (the first layer's input gradient).
template <bool EIK, bool F16, bool TEX, bool DPTS>
__global__ void __launch_bounds__(kThreads) siren_bwd_kernel(const SirenBwdK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kBwdLdsW;
    float* const film_s = smem + kBwdLdsFilm;
    float* const head_s = smem + kBwdLdsHead;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int b = blockIdx.x / a.wgs_per_img;
    const int wg = blockIdx.x - b * a.wgs_per_img;
    const long long pt0 = (long long)wg * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;

    const float* __restrict__ packed = a.packed;
    const float* __restrict__ film_g = a.film + (int64_t)b * 9 * 2 * kWidth;
    for (int i = tid; i < 9 * kWidth; i += kThreads) {
        const int l = i >> 8, n = i & 255;
        const float g = film_g[(l * 2) * kWidth + n];
        film_s[(l * 3) * kWidth + n] = g;
        film_s[(l * 3 + 1) * kWidth + n] = film_g[(l * 2 + 1) * kWidth + n];
        film_s[(l * 3 + 2) * kWidth + n] = 1.0f / g;
    }
    for (int i = tid; i < 4 * kWidth; i += kThreads) head_s[i] = packed[kOffWSigma + i];
    float* const w0_s = smem + kBwdLdsW0;
    if (DPTS) for (int i = tid; i < 3 * kWidth; i += kThreads) {
        const int c = i >> 8, n = i & 255;           // fragment image of layer 0 (siren_pack_kernel): [t][m][lane], k = 2m + half
        w0_s[i] = packed[kOffFirst + ((n >> 5) * 2 + (c >> 1)) * 64 + (c & 1) * 32 + (n & 31)];
    }
    float* const acc_s = smem + kBwdLdsAcc;
    float* const slot_s = smem + kBwdLdsSlot;
    for (int i = tid; i < 9 * 2 * kWidth; i += kThreads) acc_s[i] = 0.0f;
    // d(gamma), d(beta) are sums over points.  Global atomics would sit in vmcnt and stall every weight-chunk wait, so:
    // each wave leaves its 32-lane sums of a tile in its LDS slot; after the next workgroup barrier (every weight chunk
    // has one) the four slots are added in fixed order into the workgroup accumulator -- deterministic, LDS only.
    int pend_layer = -1, pend_t = 0, pend_par = 0, par = 0;
    auto fold_pending = [&]() {
        if (pend_layer >= 0 && lane < 16) {
            const int v = wave * 16 + lane, gb = v >> 5, nl = v & 31;
            const float* sp = slot_s + pend_par * 256 + gb * 32 + nl;
            const float sum = ((sp[0] + sp[64]) + sp[128]) + sp[192];
            acc_s[(pend_layer * 2 + gb) * kWidth + 32 * pend_t + nl] += sum;
        }
        pend_layer = -1;
    };

    ChunkPipe pipe;
    pipe.init(wbuf, packed + (F16 ? kOffBigT16 : kOffBigT), wave, lane, 0, kChunksPerPass);
    pipe.prime();
    auto issue_piece = [&](int i) { pipe.issue_piece(i); };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 ring[kRing];
    u32x4 ringH[kRing16], ringL[kRing16];
    if (!F16) {
        ring[0] = reinterpret_cast<const f32x4*>(pipe.wcur)[lane];
        ring[1] = reinterpret_cast<const f32x4*>(pipe.wcur)[64 + lane];
    } else {
#pragma unroll
        for (int g = 0; g < kRing16 - 1; ++g) {
            ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + lane];
            ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + lane];
        }
    }

    f32x16 in[kNT], out[kNT];
    u32x4 inH[2 * kNT], inL[2 * kNT];
    float inv_scale = 1.0f;                       // F16: undoes the operand scaling of the GEMM being consumed
    // F16: running max |g| of the layer being produced (this lane's 128 values).  With the two extra streams of the EIK
    // variant one more live register tips the allocator into spilling inside the MFMA stream, so that variant takes
    // the max in a separate pass over out[] instead.
    // Round 3: the running maximum is off for every variant (E3DGE_BWD_RUNMAX=1 restores it where EIK is off): with it the three
    // default-mode variants without the eikonal streams spilled 36 / 116 / 192 bytes per lane; the separate pass costs
    // 128 v_max per layer and wave.
#ifndef E3DGE_BWD_RUNMAX
#define E3DGE_BWD_RUNMAX 0
#endif
    constexpr bool kRunMax = E3DGE_BWD_RUNMAX != 0 && F16 && !EIK;
    float gmax = 0.0f;
    // g_L of a finished layer (out[], fp32) becomes the B operand of the next GEMM
    auto next_operand = [&]() {
        if (F16) {
            inv_scale = kRunMax ? scale_split(out, inH, inL, gmax) : scale_split(out, inH, inL);
            gmax = 0.0f;
        } else {
#pragma unroll
            for (int tt = 0; tt < kNT; ++tt) {
                in[tt] = out[tt];
                asm volatile("" : "+a"(in[tt]));
            }
        }
    };
#ifdef E3DGE_BWD_TIMING
    unsigned long long t_tile = 0, t_epi = 0, t_pro = 0, t_tail = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
#define BT_NOW() __builtin_readcyclecounter()
#else
#define BT_NOW() 0ull
#endif

    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    for (int sub = 0; sub < n_sub; ++sub) {
        [[maybe_unused]] const unsigned long long tp0 = BT_NOW();
        const int lane = lane_id_fresh(), half = lane >> 5, col = lane & 31;      // (per sub-tile: see lane_id_fresh)
        const int wave = wave_s;
        const int p = sub * kTilePts + 32 * wave + col;
        const bool valid = p < npts;
        const int pc = valid ? p : (npts - 1);
        const int64_t gpt = (int64_t)b * a.n_pts + pt0 + pc;
        const float* __restrict__ ap = a.args + gpt * (9 * kWidth);
        const float* __restrict__ tp_ = EIK ? a.tang + gpt * (8 * kWidth) : nullptr;
        const float* __restrict__ rp_ = EIK ? a.rsave + gpt * (8 * kWidth) : nullptr;
        const float vmask = valid ? 1.0f : 0.0f;                      // padded lanes contribute nothing
        const float* __restrict__ txa = TEX ? a.tex_alpha + gpt * kWidth : nullptr;
        float* __restrict__ dta = TEX ? a.d_tex_alpha + gpt * kWidth : nullptr;
        float* __restrict__ dtb = TEX ? a.d_tex_beta + gpt * kWidth : nullptr;
        const float dsdf = (a.d_sdf && valid) ? a.d_sdf[gpt] : 0.0f;
        float drgb[3] = {0.f, 0.f, 0.f};
        if (a.d_rgb && valid) { drgb[0] = a.d_rgb[gpt * 3]; drgb[1] = a.d_rgb[gpt * 3 + 1]; drgb[2] = a.d_rgb[gpt * 3 + 2]; }

        // per-tile reduction of d(beta) += da, d(gamma) += da * u over this wave's 32 points, layer `layer`, tile `t`
        // registers 8*h8 .. 8*h8+7 of a tile: lane sums into this wave's slot
        auto reduce_half = [&](int h8, const float (&vb)[8], const float (&vg)[8], float* my_slot) {
            float qb[4], qg[4];
            reduce8_over_lanes(vb, qb);
            reduce8_over_lanes(vg, qg);
            // lanes 0-3 of each 16-lane row publish value (col & 3) [+4 in the odd row]
            const int i = col & 3;
            const float sb = i == 0 ? qb[0] : i == 1 ? qb[1] : i == 2 ? qb[2] : qb[3];
            const float sg = i == 0 ? qg[0] : i == 1 ? qg[1] : i == 2 ? qg[2] : qg[3];
            if ((col & 15) < 4) {
                const int nl = row_of(8 * h8 + 4 * (col >> 4) + i, half);
                my_slot[nl] = sg;
                my_slot[32 + nl] = sb;
            }
        };
        auto reduce_tile = [&](int layer, int t, const float (&rb)[16], const float (&rg)[16], float* view_slot = nullptr) {
            if (!view_slot) fold_pending();                            // previous tile: a barrier has passed since
            float* const my_slot = view_slot ? view_slot : slot_s + par * 256 + wave * 64;
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                float vb[8], vg[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { vb[i] = rb[8 * h8 + i]; vg[i] = rg[8 * h8 + i]; }
                reduce_half(h8, vb, vg, my_slot);
            }
            if (!view_slot) { pend_layer = layer; pend_t = t; pend_par = par; par ^= 1; }
        };

        // =====================================================================================
        // 1. view layer: dh_view = d_feat + Wrgb^T d_rgb ; g8 = gamma8 * dh_view * cos(arg8)
        // =====================================================================================
        {
            const float* __restrict__ fg = film_s + 8 * 3 * kWidth;
            const float* __restrict__ wr = head_s + kWidth;
            const float* __restrict__ df = a.d_feat ? a.d_feat + gpt * kWidth : nullptr;
            float wfeat = 1.0f;
            if (a.d_featmap) {
                df = a.d_featmap + (gpt / a.samples) * kWidth;
                wfeat = a.weights[gpt];
            }
            // Opaque copy of the lane's half index: every address below then has to be recomputed per sub-tile.
            // Otherwise the compiler hoists ~150 loop-invariant address registers out of the sub-tile loop, cannot
            // keep them through the chain (in[] alone is 128 VGPRs), spills them, and their reloads -- in-order in
            // vmcnt with the HBM prefetches -- serialise this whole block on memory latency.
            int half_p = half;
            asm volatile("" : "+v"(half_p));
            // the (cold, HBM) argument / gradient rows of this layer are fetched two tiles ahead of their use
            constexpr int kAhead = 2;
            f32x4 arb[kAhead + 1][4], dfb[kAhead + 1][4];
            auto fetch = [&](int t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 32 * t + 8 * q + 4 * half_p;
                    arb[t % (kAhead + 1)][q] = *reinterpret_cast<const f32x4*>(ap + 8 * kWidth + o);
                    dfb[t % (kAhead + 1)][q] = df ? *reinterpret_cast<const f32x4*>(df + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
#pragma unroll
            for (int t = 0; t < kAhead; ++t) fetch(t);
            float* const slot8 = smem + kBwdLdsSlot8;
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                if (t + kAhead < kNT) fetch(t + kAhead);
                float rb[16], rg[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 32 * t + 8 * q + 4 * half_p;
                    const f32x4 ar = arb[t % (kAhead + 1)][q];
                    const f32x4 d4 = dfb[t % (kAhead + 1)][q];
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(fg + o), b4 = *reinterpret_cast<const f32x4*>(fg + kWidth + o),
                                i4 = *reinterpret_cast<const f32x4*>(fg + 2 * kWidth + o);
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + o), w1 = *reinterpret_cast<const f32x4*>(wr + kWidth + o),
                                w2 = *reinterpret_cast<const f32x4*>(wr + 2 * kWidth + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dh = vmask * (wfeat * d4[j] + w0[j] * drgb[0] + w1[j] * drgb[1] + w2[j] * drgb[2]);
                        const float da = dh * cos_hw_f32(ar[j]);
                        rb[4 * q + j] = da;
                        rg[4 * q + j] = da * ((ar[j] - b4[j]) * i4[j]);
                        out[t][4 * q + j] = g4[j] * da;
                        if (kRunMax) gmax = fmaxf(gmax, fabsf(out[t][4 * q + j]));
                    }
                }
                asm volatile("" : "+a"(out[t]));                       // results wait in AGPRs: the VGPRs stay free for the loads
                reduce_tile(8, t, rb, rg, slot8 + (t * 4 + wave) * 64);   // own slot per (tile, wave): no barrier needed here
            }
            next_operand();
            // one barrier, then every wave folds its quarter of the eight tiles in fixed order
            __syncthreads();
            fold_pending();                                            // the previous sub-tile's last tile
            if (lane < 16) {
#pragma unroll
                for (int t = 0; t < kNT; ++t) {
                    const int v = wave * 16 + lane, gb = v >> 5, nl = v & 31;
                    const float* sp8 = slot8 + t * 256 + gb * 32 + nl;
                    acc_s[(8 * 2 + gb) * kWidth + 32 * t + nl] += ((sp8[0] + sp8[64]) + sp8[128]) + sp8[192];
                }
            }
        }

#ifdef E3DGE_BWD_TIMING
        t_pro += BT_NOW() - tp0;
#endif
        // =====================================================================================
        // 2. the chain: GEMM Gb (layer L = 8 - Gb) turns g_L into dh_{L-1}; its epilogue makes g_{L-1}
        // =====================================================================================
#pragma unroll 1
        for (int Gb = 0; Gb < kBigLayers; ++Gb) {
            const int Lm1 = 7 - Gb;                                      // layer whose argument / FiLM the epilogue uses
            const float* __restrict__ fg = film_s + Lm1 * 3 * kWidth;
            const float* __restrict__ apl = ap + Lm1 * kWidth;
            const float* __restrict__ tpl = EIK ? tp_ + Lm1 * kWidth : nullptr;
            const float* __restrict__ rpl = EIK ? rp_ + Lm1 * kWidth : nullptr;
            const float sdf_term = (Gb == 0) ? dsdf : 0.0f;              // sdf head reads the backbone output h8
            f32x16 prev;
            f32x4 argb[4];                                               // saved arguments of the tile being finished
            f32x4 argt[4];                                               // ... and of the layer's last tile (no GEMM tile follows it)
            f32x4 tgb[4], rsb[4];                                        // EIK: tangent arguments and r of the tile being finished
            // F16: the epilogue of tile t-1 is issued from inside GEMM tile t (one accumulator register per k-step, in the
            // shadow of the f16 MFMAs -- standing alone it cost more than the GEMM), so its streams are fetched a tile ahead
            f32x4 arg2[2][4], tg2[2][4], rs2[2][4];
            f32x4 e_g = {0.f, 0.f, 0.f, 0.f}, e_b = e_g, e_i = e_g, e_w = e_g;
            float rbh[8], rgh[8];
            auto epilogue = [&](int tp, const f32x16& dhv, const f32x4 (&ar)[4], f32x16& dst) {
                float rb[16], rg[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 32 * tp + 8 * q + 4 * half;
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(fg + o), b4 = *reinterpret_cast<const f32x4*>(fg + kWidth + o),
                                i4 = *reinterpret_cast<const f32x4*>(fg + 2 * kWidth + o);
                    const f32x4 ws = *reinterpret_cast<const f32x4*>(head_s + o);
                    const bool tex_here = TEX && Gb == 0;                      // this GEMM's result is dL/dh8' (view-layer input)
                    f32x4 al4 = {0.f, 0.f, 0.f, 0.f}, da4 = al4, db4 = al4;
                    if (tex_here) al4 = *reinterpret_cast<const f32x4*>(txa + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float xin = dhv[4 * q + j];
                        float tsn = 0.f, tcs = 0.f;
                        if (TEX) {
                            sincos_hw_f32(ar[q][j], tsn, tcs);
                            if (tex_here) { da4[j] = xin * tsn; db4[j] = xin; xin = __fadd_rn(al4[j], 1.0f) * xin; }
                        }
                        const float dh = fmaf(ws[j], sdf_term, xin);   // padded lanes: in[] = 0 and dsdf = 0, so dh = 0
                        float da, dg_extra = 0.0f;
                        if (TEX) {
                            da = dh * tcs;
                        } else if (EIK) {
                            float sn, cs;
                            sincos_hw_f32(ar[q][j], sn, cs);
                            const float tr = vmask * tgb[q][j] * rsb[q][j];
                            da = fmaf(dh, cs, -sn * tr);
                            dg_extra = tr * i4[j] * cs;
                        } else {
#ifndef E3DGE_BWD_ABL_NO_COS
                            da = dh * cos_hw_f32(ar[q][j]);
#else
                            da = dh * ar[q][j];
#endif
                        }
                        rb[4 * q + j] = da;
                        rg[4 * q + j] = fmaf(da, (ar[q][j] - b4[j]) * i4[j], dg_extra);
                        dst[4 * q + j] = g4[j] * da;
                        if (kRunMax) gmax = fmaxf(gmax, fabsf(dst[4 * q + j]));
                    }
                    if (tex_here && valid) {
                        *reinterpret_cast<f32x4*>(dta + o) = da4;
                        *reinterpret_cast<f32x4*>(dtb + o) = db4;
                    }
                }
#ifndef E3DGE_BWD_ABL_NO_REDUCE
                reduce_tile(Lm1, tp, rb, rg);
#else
                float keep = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) keep += rb[i] + rg[i];
                dst[0] += keep * 1e-30f;
#endif
            };
            f32x4 e_al = {0.f, 0.f, 0.f, 0.f}, e_da = e_al, e_db = e_al;
            auto epi_step = [&](int tp, int r, f32x16& dst) {           // tp, r: compile-time constants at every call site
                const int q = r >> 2, j = r & 3;
                const bool tex_here = TEX && Gb == 0;
                if (j == 0) {
                    const int o = 32 * tp + 8 * q + 4 * half;
                    e_g = *reinterpret_cast<const f32x4*>(fg + o); e_b = *reinterpret_cast<const f32x4*>(fg + kWidth + o);
                    e_i = *reinterpret_cast<const f32x4*>(fg + 2 * kWidth + o); e_w = *reinterpret_cast<const f32x4*>(head_s + o);
                    if (tex_here) e_al = *reinterpret_cast<const f32x4*>(txa + o);
                }
                const float ar = arg2[tp & 1][q][j];
                float xin = prev[r];
                float tsn = 0.f, tcs = 0.f;
                if (TEX) {
                    sincos_hw_f32(ar, tsn, tcs);
                    if (tex_here) { e_da[j] = xin * tsn; e_db[j] = xin; xin = __fadd_rn(e_al[j], 1.0f) * xin; }
                }
                const float dh = fmaf(e_w[j], sdf_term, xin);
                float da, dg_extra = 0.0f;
                if (TEX) {
                    da = dh * tcs;
                } else if (EIK) {
                    float sn, cs;
                    sincos_hw_f32(ar, sn, cs);
                    const float tr = vmask * tg2[tp & 1][q][j] * rs2[tp & 1][q][j];
                    da = fmaf(dh, cs, -sn * tr);
                    dg_extra = tr * e_i[j] * cs;
                } else {
                    da = dh * cos_hw_f32(ar);
                }
                rbh[r & 7] = da;
                rgh[r & 7] = fmaf(da, (ar - e_b[j]) * e_i[j], dg_extra);
                dst[r] = e_g[j] * da;
                if (kRunMax) gmax = fmaxf(gmax, fabsf(dst[r]));
                if (tex_here && j == 3 && valid) {
                    const int o = 32 * tp + 8 * q + 4 * half;
                    *reinterpret_cast<f32x4*>(dta + o) = e_da;
                    *reinterpret_cast<f32x4*>(dtb + o) = e_db;
                }
                if (r == 3) fold_pending();                              // this tile's chunk barrier (k-step 2) has passed
                if ((r & 7) == 7) {
                    reduce_half(r >> 3, rbh, rgh, slot_s + par * 256 + wave * 64);
                    if (r == 15) { pend_layer = Lm1; pend_t = tp; pend_par = par; par ^= 1; }
                }
            };
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                // The saved arguments of tile t-1 (HBM, cold stream) are needed by the epilogue that follows GEMM tile t.
                // They are issued right AFTER this tile's weight-chunk wait (issued before it they would stall the
                // vmcnt(0) of the wait itself) and have the rest of the tile, ~7.5k cycles of MFMAs, to arrive.
                auto sync_and_fetch = [&]() {
                    pipe.sync();
                    if (F16) {                                           // streams of tile t itself: used one GEMM tile later
#pragma unroll
                        for (int q = 0; q < 4; ++q) arg2[t & 1][q] = *reinterpret_cast<const f32x4*>(apl + 32 * t + 8 * q + 4 * half);
                        if (EIK) {
                            // The four quarter-line loads of a stream BACK TO BACK, stream after stream (round 5).  A tile's streams are 3 x 32 rows x
                            // 128 B per wave = 48 KB per CU, more than the L1 holds: interleaved (arg, ta, r per quad) a line's four accesses were
                            // spread over twelve instructions of every wave and mostly missed again -- L2 served each line up to four times.
                            // Timing ablations: no streams -1.24 ms, whole-line loads -0.76 ms, this order -0.58 ms of the
                            // 7.98-ms step at four samples per GPU (-0.38 / -0.20 / -0.10 of 2.46 ms at one).
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < 4; ++q) tg2[t & 1][q] = *reinterpret_cast<const f32x4*>(tpl + 32 * t + 8 * q + 4 * half);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < 4; ++q) rs2[t & 1][q] = *reinterpret_cast<const f32x4*>(rpl + 32 * t + 8 * q + 4 * half);
                        }
                        return;
                    }
                    if (t > 0) {
#ifndef E3DGE_BWD_ABL_NO_ARGLOAD
#pragma unroll
                        for (int q = 0; q < 4; ++q) argb[q] = *reinterpret_cast<const f32x4*>(apl + 32 * (t - 1) + 8 * q + 4 * half);
#else
                        for (int q = 0; q < 4; ++q) argb[q] = f32x4{0.1f * t, 0.2f, 0.3f + q, 0.4f};
#endif
                    }
                    if (t == kNT - 1) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) argt[q] = *reinterpret_cast<const f32x4*>(apl + 32 * (kNT - 1) + 8 * q + 4 * half);
                    }
                    if (EIK && t > 0) {     // the second-order streams of the same tile, stream after stream (see the f16 branch)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) tgb[q] = *reinterpret_cast<const f32x4*>(tpl + 32 * (t - 1) + 8 * q + 4 * half);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) rsb[q] = *reinterpret_cast<const f32x4*>(rpl + 32 * (t - 1) + 8 * q + 4 * half);
                    }
                };
                f32x16 acc = zero16();
                [[maybe_unused]] const unsigned long long c0 = BT_NOW();
                if (!F16) {
                    acc = big_tile<false, 0>(pipe.wcur, pipe.wnxt, lane, in, acc, ring, NoEpilogue(), sync_and_fetch, issue_piece);
                } else {
                    f32x16 accb = zero16();
                    if (t == 0) {
                        big_tile_f16<false>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL, NoEpilogue(), sync_and_fetch, issue_piece);
                    } else {
                        big_tile_f16<false>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL,
                                            [&](int r) { epi_step(t - 1, r, out[t - 1]); }, sync_and_fetch, issue_piece);
                        asm volatile("" : "+a"(out[t - 1]));
                    }
                    acc = (acc + accb) * inv_scale;
                }
                [[maybe_unused]] const unsigned long long c1 = BT_NOW();
                pipe.advance();
                if (!F16 && t > 0) {
                    epilogue(t - 1, prev, argb, out[t - 1]);
                    asm volatile("" : "+a"(out[t - 1]));
                }
                prev = acc;
                if (F16) asm volatile("" : "+v"(prev));
#ifdef E3DGE_BWD_TIMING
                t_tile += c1 - c0; t_epi += BT_NOW() - c1;
#endif
            }
            [[maybe_unused]] const unsigned long long ct0 = BT_NOW();
            if (F16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    argt[q] = arg2[(kNT - 1) & 1][q];
                    if (EIK) { tgb[q] = tg2[(kNT - 1) & 1][q]; rsb[q] = rs2[(kNT - 1) & 1][q]; }
                }
            } else if (EIK) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    tgb[q] = *reinterpret_cast<const f32x4*>(tpl + 32 * (kNT - 1) + 8 * q + 4 * half);
                    rsb[q] = *reinterpret_cast<const f32x4*>(rpl + 32 * (kNT - 1) + 8 * q + 4 * half);
                }
            }
            __syncthreads();       // the last two epilogues of a layer have no weight-chunk barrier between them
            epilogue(kNT - 1, prev, argt, out[kNT - 1]);
            next_operand();
#ifdef E3DGE_BWD_TIMING
            t_tail += BT_NOW() - ct0;
#endif
        }
        // ---- optional: dL/dx = s W_0^T g_0 (g_0 = gamma_0 * adj(a_0) is in out[]), like the sdf chain's last step ----
        if (DPTS) {
            int half_e = half;
            asm volatile("" : "+v"(half_e));
            float ex = 0.f, ey = 0.f, ez = 0.f;
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 32 * t + 8 * q + 4 * half_e;
                    const f32x4 wx = *reinterpret_cast<const f32x4*>(w0_s + o), wy = *reinterpret_cast<const f32x4*>(w0_s + kWidth + o),
                                wz = *reinterpret_cast<const f32x4*>(w0_s + 2 * kWidth + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = out[t][4 * q + j];
                        ex = fmaf(wx[j], g, ex); ey = fmaf(wy[j], g, ey); ez = fmaf(wz[j], g, ez);
                    }
                }
            }
            ex += xhalf(ex); ey += xhalf(ey); ez += xhalf(ez);
            if (valid && half == 0) {
                float* o = a.d_pts + gpt * 3;
                o[0] = ex * a.box_scale; o[1] = ey * a.box_scale; o[2] = ez * a.box_scale;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fold_pending();
    __syncthreads();
    float* const my_partial = a.partials + (int64_t)blockIdx.x * (9 * 2 * kWidth);
    for (int i = tid; i < 9 * 2 * kWidth; i += kThreads) my_partial[i] = acc_s[i];
#ifdef E3DGE_BWD_TIMING
    // profiling build: the first floats of this workgroup's slice carry wave 0's cycle counts (tools/bwd_timing.py)
    __syncthreads();
    if (tid == 0) {
        my_partial[0] = (float)(BT_NOW() - t_begin); my_partial[1] = (float)t_pro; my_partial[2] = (float)t_tile;
        my_partial[3] = (float)t_epi; my_partial[4] = (float)t_tail; my_partial[5] = (float)pipe.t_vm; my_partial[6] = (float)pipe.t_bar;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The two first-order chains the eikonal term needs, one kernel template (seven 256x256 GEMMs per point each):
//   TANGENT = false  "sdf chain":  r_7 = w_sigma * seed ;  r_{l-1} = W_l^T (gamma_l cos(a_l) r_l) ;
//                    saves r_l = d sdf / d h_l (l = 0..7) and writes e = d sdf / d x = s W_0^T (gamma_0 cos(a_0) r_0)
//                    -- get_eikonal_term (volume_renderer.py:796-802) without an autograd graph.
//   TANGENT = true   "tangent":    ta_0 = gamma_0 W_0 (s v) ;  ta_l = gamma_l W_l (cos(a_{l-1}) ta_{l-1}) ;
//                    saves ta_l (l = 0..7): the forward-mode derivative of the arguments along v = dL/de.
// Both feed the next GEMM with gamma cos(a) acc; they differ in direction (transposed image, descending layers / forward
// image, ascending layers), in what is stored (acc / gamma acc) and at the two ends.  Stores of a tile are issued after
// the NEXT tile's weight-chunk wait so that they never sit in front of a vmcnt(0).
// ---------------------------------------------------------------------------------------------------------------
struct SirenChainK {
    const float* packed;
    const float* film;
    const float* args;       // (batch, n_pts, 9, 256)
    const float* seed;       // TANGENT: v (batch, n_pts, 3);  else: per-point scale of r_7 (batch, n_pts) or null (= 1)
    float* save;             // (batch, n_pts, 8, 256): ta_l or r_l
    const float* rmul;       // 8-wave tangent kernel, product form: r_l of the sdf chain (batch, n_pts, 8, 256); save <- ta_l r_l
    float* eik;              // (batch, n_pts, 3), sdf chain only
    float box_scale;
    long long n_pts;
    int batch, subtiles_per_wg, wgs_per_img;
};

constexpr int kChLdsW = 0;
constexpr int kChLdsFilm = kChLdsW + kNBuf * kChunkFloats;      // [8][256] gamma of the backbone layers
constexpr int kChLdsW0 = kChLdsFilm + 8 * kWidth;               // [3][256] first-layer weights, column-major
constexpr int kChLdsHead = kChLdsW0 + 3 * kWidth;               // w_sigma[256]
constexpr int kChLdsSt = kChLdsHead + kWidth;                    // [4 waves][2 tiles][32 points][32 floats]: the store transpose (see ch_put / ch_drain)
constexpr int kChLdsFloats = kChLdsSt + 4 * 2 * 1024;
constexpr int kChLdsBytes = kChLdsFloats * 4;
static_assert(kChLdsBytes <= 160 * 1024, "LDS budget (chain)");

template <bool TANGENT, bool F16>
__global__ void __launch_bounds__(kThreads) siren_chain_kernel(const SirenChainK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kChLdsW;
    float* const gam_s = smem + kChLdsFilm;
    float* const w0_s = smem + kChLdsW0;
    float* const ws_s = smem + kChLdsHead;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int b = blockIdx.x / a.wgs_per_img;
    const int wg = blockIdx.x - b * a.wgs_per_img;
    const long long pt0 = (long long)wg * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;

    const float* __restrict__ packed = a.packed;
    const float* __restrict__ film_g = a.film + (int64_t)b * 9 * 2 * kWidth;
    for (int i = tid; i < 8 * kWidth; i += kThreads) gam_s[i] = film_g[((i >> 8) * 2) * kWidth + (i & 255)];
    for (int i = tid; i < 3 * kWidth; i += kThreads) {
        const int c = i >> 8, n = i & 255;           // fragment image of layer 0 (siren_pack_kernel): [t][m][lane], k = 2m + half
        w0_s[i] = packed[kOffFirst + ((n >> 5) * 2 + (c >> 1)) * 64 + (c & 1) * 32 + (n & 31)];
    }
    for (int i = tid; i < kWidth; i += kThreads) ws_s[i] = packed[kOffWSigma + i];

    constexpr int kChainChunks = 7 * kNT;             // layers 1..7
    ChunkPipe pipe;
    pipe.init(wbuf, packed + (TANGENT ? (F16 ? kOffBig16 : kOffBig) : (F16 ? kOffBigT16 : kOffBigT)), wave, lane,
              TANGENT ? 0 : kNT, kChainChunks);
    pipe.prime();
    auto issue_piece = [&](int i) { pipe.issue_piece(i); };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 ring[kRing];
    u32x4 ringH[kRing16], ringL[kRing16];
    if (!F16) {
        ring[0] = reinterpret_cast<const f32x4*>(pipe.wcur)[lane];
        ring[1] = reinterpret_cast<const f32x4*>(pipe.wcur)[64 + lane];
    } else {
#pragma unroll
        for (int g = 0; g < kRing16 - 1; ++g) {
            ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + lane];
            ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + lane];
        }
    }

    f32x16 in[kNT], out[kNT];
    u32x4 inH[2 * kNT], inL[2 * kNT];
    // F16 tangent chain: epilogue interleaved into the next GEMM tile + running column max.  The sdf chain keeps the
    // epilogue after the tile (interleaved, the allocator spills ~30 registers inside the MFMA stream of that variant).
    constexpr bool kInter = F16 && TANGENT;
    float inv_scale = 1.0f, gmax = 0.0f;
    auto next_operand = [&]() {
        if (F16) {
            inv_scale = kInter ? scale_split(out, inH, inL, gmax) : scale_split(out, inH, inL);
            gmax = 0.0f;
        } else {
#pragma unroll
            for (int tt = 0; tt < kNT; ++tt) {
                in[tt] = out[tt];
                asm volatile("" : "+a"(in[tt]));
            }
        }
    };

    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    for (int sub = 0; sub < n_sub; ++sub) {
        const int lane = lane_id_fresh(), half = lane >> 5, col = lane & 31;      // (per sub-tile: see lane_id_fresh)
        const int wave = wave_s;
        const int p = sub * kTilePts + 32 * wave + col;
        const bool valid = p < npts;
        const int pc = valid ? p : (npts - 1);
        const int64_t gpt = (int64_t)b * a.n_pts + pt0 + pc;
        const float* __restrict__ ap = a.args + gpt * (9 * kWidth);
        // The saved tile of a layer (16 values per lane: quads 2q + half of its point's 128-byte line) leaves through LDS: written as the lane
        // holds it, read back so that EIGHT lanes cover one point's whole line -- a store instruction then writes 8 full 128-B lines instead
        // of 32 quarter lines (timing ablations, DESIGN.md 4.6: the quarter-line stores cost 0.7 ms of the 8.7-ms step at four samples per
        // GPU).  Piece p of point c sits at slot p ^ ((c >> 1) & 7) of the point's eight 16-byte slots: conflict-free for both the
        // 16 consecutive points of a write and the 2 points x 8 pieces of a read.  Two tile buffers per wave; only this wave touches them.
        float* const stl = smem + kChLdsSt + wave * 2048;
        const int st_wr = col * 32, st_pc = 4 * half, st_sw = 4 * ((col >> 1) & 7);              // write: quad q -> float (8 q + 4 half) ^ st_sw of the point's 32
        const int dc = lane >> 3, dp = lane & 7;                                                   // drain: this lane = piece dp of points 8 i + dc
        const int n_rows_ok = npts - (sub * kTilePts + 32 * wave);                                 // rows of this wave's 32 that exist
        float* const sw0 = a.save + ((int64_t)b * a.n_pts + pt0 + sub * kTilePts + 32 * wave) * (8 * kWidth);   // wave-uniform: the wave's first row
        const int st_goff = dc * (8 * kWidth) + 4 * dp;
        auto ch_put = [&](int buf, int q, const f32x4& v) {
            *reinterpret_cast<f32x4*>(stl + buf * 1024 + st_wr + ((8 * q + st_pc) ^ st_sw)) = v;
        };
        auto ch_drain_piece = [&](int buf, int i, float* __restrict__ layer_base, int tp) {      // piece i of four: points 8 i .. 8 i + 7
            const int c = 8 * i + dc;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stl + buf * 1024 + c * 32 + 4 * (dp ^ ((4 * i + (dc >> 1)) & 7)));
            if (c < n_rows_ok) *reinterpret_cast<f32x4*>(layer_base + st_goff + i * (8 * 8 * kWidth) + 32 * tp) = v;
        };

        // ---- first layer of the chain (no GEMM) ----
        {
            int half_p = half;                       // opaque: keeps the address math of this block inside the loop
            asm volatile("" : "+v"(half_p));         // (see siren_bwd_kernel)
            const int l0 = TANGENT ? 0 : 7;
            const float* __restrict__ gl = gam_s + l0 * kWidth;
            float sx = 0.f, sy = 0.f, sz = 0.f, seed = 1.0f;
            if (TANGENT) {
                const float* vv = a.seed + gpt * 3;
                sx = vv[0] * a.box_scale; sy = vv[1] * a.box_scale; sz = vv[2] * a.box_scale;
            } else if (a.seed) {
                seed = a.seed[gpt];
            }
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 32 * t + 8 * q + 4 * half_p;
                    const f32x4 ar = *reinterpret_cast<const f32x4*>(ap + l0 * kWidth + o);
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gl + o);
                    f32x4 x4;
                    if (TANGENT) {
                        const f32x4 wx = *reinterpret_cast<const f32x4*>(w0_s + o), wy = *reinterpret_cast<const f32x4*>(w0_s + kWidth + o),
                                    wz = *reinterpret_cast<const f32x4*>(w0_s + 2 * kWidth + o);
#pragma unroll
                        for (int j = 0; j < 4; ++j) x4[j] = g4[j] * fmaf(wz[j], sz, fmaf(wy[j], sy, wx[j] * sx));   // ta_0
                    } else {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws_s + o);
#pragma unroll
                        for (int j = 0; j < 4; ++j) x4[j] = w4[j] * seed;                                          // r_7
                    }
                    ch_put(t & 1, q, x4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        out[t][4 * q + j] = cos_hw_f32(ar[j]) * (TANGENT ? x4[j] : g4[j] * x4[j]);
                        if (kInter) gmax = fmaxf(gmax, fabsf(out[t][4 * q + j]));
                    }
                }
                asm volatile("" : "+a"(out[t]));
#pragma unroll
                for (int i = 0; i < 4; ++i) ch_drain_piece(t & 1, i, sw0 + l0 * kWidth, t);
            }
            next_operand();
        }

        // ---- seven GEMMs ----
#pragma unroll 1
        for (int step = 0; step < 7; ++step) {
            const int l = TANGENT ? step + 1 : 6 - step;                 // layer whose argument / gamma the epilogue uses
            const float* __restrict__ gl = gam_s + l * kWidth;
            const float* __restrict__ apl = ap + l * kWidth;
            float* __restrict__ swl = sw0 + l * kWidth;
            f32x16 prev;
            f32x4 argb[2][4];
            auto epilogue = [&](int tp, const f32x16& accv, const f32x4 (&ar)[4], f32x16& dst) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gl + 32 * tp + 8 * q + 4 * half);
                    f32x4 s4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ga = g4[j] * accv[4 * q + j];
                        s4[j] = TANGENT ? ga : accv[4 * q + j];
                        dst[4 * q + j] = cos_hw_f32(ar[q][j]) * ga;
                        if (kInter) gmax = fmaxf(gmax, fabsf(dst[4 * q + j]));
                    }
                    ch_put(tp & 1, q, s4);                                   // stored by ch_drain_piece one chunk wait later
                }
            };
            // F16: the same work for tile tp issued from inside the next GEMM tile, one accumulator register per k-step; the
            // 16-B stores go out as each quad completes (they are retired by the next tile's chunk wait)
            f32x4 e_g = {0.f, 0.f, 0.f, 0.f}, e_st = e_g;
            auto epi_step = [&](int tp, int r, f32x16& dst) {
                const int q = r >> 2, j = r & 3;
                if (j == 0) e_g = *reinterpret_cast<const f32x4*>(gl + 32 * tp + 8 * q + 4 * half);
                const float ga = e_g[j] * prev[r];
                e_st[j] = TANGENT ? ga : prev[r];
                dst[r] = cos_hw_f32(argb[tp & 1][q][j]) * ga;
                gmax = fmaxf(gmax, fabsf(dst[r]));
                if (j == 3) ch_put(tp & 1, q, e_st);
                if (tp > 0 && r >= 4 && r < 8) ch_drain_piece((tp - 1) & 1, r - 4, swl, tp - 1);       // tile tp - 1: written during the previous GEMM tile
            };
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                auto sync_and_fetch = [&]() {
                    pipe.sync();
                    if (t > 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(argb[(t - 1) & 1][q]));
                    }
                    if (!kInter && t > 1) {                              // tile t-2 finished during the previous GEMM tile
#pragma unroll
                        for (int i = 0; i < 4; ++i) ch_drain_piece((t - 2) & 1, i, swl, t - 2);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) argb[t & 1][q] = *reinterpret_cast<const f32x4*>(apl + 32 * t + 8 * q + 4 * half);
                };
                f32x16 acc = zero16();
                if (!F16) {
                    acc = big_tile<false, 0>(pipe.wcur, pipe.wnxt, lane, in, acc, ring, NoEpilogue(), sync_and_fetch, issue_piece);
                } else {
                    f32x16 accb = zero16();
                    if (t == 0 || !kInter) {
                        big_tile_f16<false>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL, NoEpilogue(), sync_and_fetch, issue_piece);
                    } else {
                        big_tile_f16<false>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL,
                                            [&](int r) { epi_step(t - 1, r, out[t - 1]); }, sync_and_fetch, issue_piece);
                        asm volatile("" : "+a"(out[t - 1]));
                    }
                    acc = (acc + accb) * inv_scale;
                }
                pipe.advance();
                if (!kInter && t > 0) {
                    epilogue(t - 1, prev, argb[(t - 1) & 1], out[t - 1]);
                    asm volatile("" : "+a"(out[t - 1]));
                }
                prev = acc;
                if (kInter) asm volatile("" : "+v"(prev));
            }
            // tail of the layer: tile 6 is still in its LDS buffer, tile 7 has no GEMM tile after it
#pragma unroll
            for (int i = 0; i < 4; ++i) ch_drain_piece((kNT - 2) & 1, i, swl, kNT - 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(argb[(kNT - 1) & 1][q]));
            epilogue(kNT - 1, prev, argb[(kNT - 1) & 1], out[kNT - 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) ch_drain_piece((kNT - 1) & 1, i, swl, kNT - 1);
            if (step < 6) next_operand();
        }

        // ---- sdf chain: e = s W_0^T g_0 (g_0 is in out[]) ----
        if (!TANGENT) {
            int half_e = half;
            asm volatile("" : "+v"(half_e));
            float ex = 0.f, ey = 0.f, ez = 0.f;
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 32 * t + 8 * q + 4 * half_e;
                    const f32x4 wx = *reinterpret_cast<const f32x4*>(w0_s + o), wy = *reinterpret_cast<const f32x4*>(w0_s + kWidth + o),
                                wz = *reinterpret_cast<const f32x4*>(w0_s + 2 * kWidth + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = out[t][4 * q + j];
                        ex = fmaf(wx[j], g, ex); ey = fmaf(wy[j], g, ey); ez = fmaf(wz[j], g, ez);
                    }
                }
            }
            ex += xhalf(ex); ey += xhalf(ey); ez += xhalf(ez);
            if (valid && half == 0) {
                float* o = a.eik + gpt * 3;
                o[0] = ex * a.box_scale; o[1] = ey * a.box_scale; o[2] = ez * a.box_scale;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of volume_integration (project/utils/volume_renderer.py:809-943, the sdf / force_background configuration)
// down to the per-point network outputs: one wave per ray.
//   dL/dw_s   = 2 sum_c dRGB_c sigmoid(rgb_s,c) + <dFEAT, feat_s> + <dXYZ, pts_s> + dDEPTH z_s
//   w_s       = alpha_s T_s,  T_s = prod_{j<s} (1 - alpha_j + 1e-10);  force_background: w_{S-1} = 1 - sum_{j<S-1} w_j
//   dL/dalpha_s = dw_s T_s - (sum_{j>s} dw_j alpha_j T_j) / (1 - alpha_s + 1e-10)
//   alpha = 1 - exp(-sigma delta), sigma = sigmoid(-sdf/beta)/beta
// feat_s = sin(arg8_s) and rgb_s = Wrgb feat_s + b are recomputed from the saved view-layer arguments.
// ---------------------------------------------------------------------------------------------------------------
struct CompositeBwdK {
    const float* packed; const float* args; const float* sdf; const float* dists; const float* points;
    const float* weights; const float* t_vals; const float* near; const float* far;
    const float* d_rgbmap;   // (rays, 3)
    const float* d_featmap;  // (rays, 256)
    const float* d_xyzmap;   // (rays, 3) or null
    const float* d_depthmap; // (rays) or null
    const float* d_sdf_in;   // (rays, S) or null: gradient arriving at the per-point sdf output
    const float* d_weights;  // (rays, S) or null: gradient arriving at the compositing weights (hit_prob)
    float* d_rgb_pts;        // (rays, S, 3) out
    float* d_sdf_pts;        // (rays, S) out
    float sigmoid_beta;
    int S, force_bg;
    long long n_rays, rays_per_img;
    int args_blocked;        // the saved arguments are slab-major (siren_common.h): precision f16x3_g2
};

constexpr int kCbStride = 8;     // floats of LDS per sample: dw, alpha, T, dalpha/dsdf, rgb[3], d_sdf

__global__ void __launch_bounds__(kThreads) composite_bwd_kernel(const CompositeBwdK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = a.S;
    float* const ws = smem + (size_t)wave * S * kCbStride;
    const float* __restrict__ wrgb = a.packed + kOffWRgb;
    const float* __restrict__ bhead = a.packed + kOffBHead;
    f32x4 wr4[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) wr4[c] = *reinterpret_cast<const f32x4*>(wrgb + c * kWidth + 4 * lane);
    const float inv_beta = 1.0f / a.sigmoid_beta;
    const long long stride = (long long)gridDim.x * 4;
    const long long trips = (a.n_rays + stride - 1) / stride;
    for (long long it = 0; it < trips; ++it) {
        const long long ray = it * stride + (long long)blockIdx.x * 4 + wave;
        const bool active = ray < a.n_rays;
        if (active) {
            const int b = (int)(ray / a.rays_per_img);
            const float nearv = a.near[b], farv = a.far[b];
            f32x4 df4 = {0.f, 0.f, 0.f, 0.f};
            if (a.d_featmap) df4 = *reinterpret_cast<const f32x4*>(a.d_featmap + ray * kWidth + 4 * lane);
            float drgb[3] = {0.f, 0.f, 0.f}, dxyz[3] = {0.f, 0.f, 0.f};
            if (a.d_rgbmap) { drgb[0] = a.d_rgbmap[ray * 3]; drgb[1] = a.d_rgbmap[ray * 3 + 1]; drgb[2] = a.d_rgbmap[ray * 3 + 2]; }
            if (a.d_xyzmap) { dxyz[0] = a.d_xyzmap[ray * 3]; dxyz[1] = a.d_xyzmap[ray * 3 + 1]; dxyz[2] = a.d_xyzmap[ray * 3 + 2]; }
            const float ddepth = a.d_depthmap ? a.d_depthmap[ray] : 0.0f;
            // phase 1: recompute feat / rgb of every sample, four wave-wide dot products each.  The S argument rows of a ray are fetched six
            // at a time (round 6: one load in flight per wave made this kernel a chain of S cold-HBM round trips, 48-61 us for 4,096 rays)
            const long long arow0 = (long long)b * saved_rows_per_image(a.args_blocked != 0, a.rays_per_img * S) + (ray * S - (long long)b * a.rays_per_img * S);
            // phase 0 (round 6, late): the per-sample scalars of d weight -- d xyz . point + d depth z (+ d weights) -- lanes over samples, into
            // slot 0 of the sample's LDS row.  They used to be loaded by lane 0 inside the sample loop below: one exposed round trip per sample.
            for (int s = lane; s < S; s += kWave) {
                const long long gpt = ray * S + s;
                const float tv = a.t_vals[s];
                const float z = nearv * (1.0f - tv) + farv * tv;
                const float* pp = a.points + gpt * 3;
                float ex = dxyz[0] * pp[0] + dxyz[1] * pp[1] + dxyz[2] * pp[2] + ddepth * z;
                if (a.d_weights) ex += a.d_weights[gpt];
                ws[s * kCbStride + 0] = ex;
                // (what used to be phase 2 -- alpha and d(alpha)/d(sdf) -- needs only sdf and dists: its loads travel with the ones above)
                const float sg = sigmoid_f32(-a.sdf[gpt] * inv_beta);
                const float sigma = sg * inv_beta;
                const float delta = a.dists[gpt];
                const float e = __expf(-sigma * delta);
                ws[s * kCbStride + 1] = 1.0f - e;
                // d alpha / d sdf = delta e * (-sg (1 - sg) / beta^2); delta = 1e10 |d| on the last sample: e == 0 there
                ws[s * kCbStride + 3] = (e == 0.0f) ? 0.0f : delta * e * (-sg * (1.0f - sg) * inv_beta * inv_beta);
            }
            __builtin_amdgcn_wave_barrier();
            constexpr int kCbAhead = 6;
            for (int s0 = 0; s0 < S; s0 += kCbAhead) {
                f32x4 a4s[kCbAhead];
#pragma unroll
                for (int u = 0; u < kCbAhead; ++u) {
                    // the view layer's 256 arguments of the sample, features 4 lane .. 4 lane + 3 (slab-major: tile lane >> 2, q = lane & 3)
                    const int sc = min(s0 + u, S - 1);
                    a4s[u] = *reinterpret_cast<const f32x4*>(a.args + saved_elem_floats(a.args_blocked != 0, arow0 + sc, 8, 4 * lane, 9));
                }
#pragma unroll
                for (int u = 0; u < kCbAhead; ++u) {
                    const int s = s0 + u;
                    if (s >= S) break;
                    const long long gpt = ray * S + s;
                    const f32x4 a4 = a4s[u];
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float f = sin_f32(a4[j]);
                        v[0] = fmaf(df4[j], f, v[0]);
                        v[1] = fmaf(wr4[0][j], f, v[1]);
                        v[2] = fmaf(wr4[1][j], f, v[2]);
                        v[3] = fmaf(wr4[2][j], f, v[3]);
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], off, kWave);
                    }
                    if (lane == 0) {
                        float dw = v[0] + ws[s * kCbStride + 0];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float rc = v[1 + c] + bhead[1 + c];
                            ws[s * kCbStride + 4 + c] = rc;
                            dw = fmaf(2.0f * drgb[c], sigmoid_f32(rc), dw);
                        }
                        ws[s * kCbStride + 0] = dw;
                    }
                }
            }
        }
        __syncthreads();
        if (active && lane == 0) {
            // phase 3: the two sequential scans of one ray (S is a few dozen)
            float T = 1.0f, dw_last = 0.0f;
            for (int s = 0; s < S; ++s) {
                ws[s * kCbStride + 2] = T;
                T *= (1.0f - ws[s * kCbStride + 1] + 1e-10f);
            }
            if (a.force_bg) dw_last = ws[(S - 1) * kCbStride + 0];
            float suffix = 0.0f;
            for (int s = S - 1; s >= 0; --s) {
                const float al = ws[s * kCbStride + 1], Ts = ws[s * kCbStride + 2];
                const float dwe = (a.force_bg && s == S - 1) ? 0.0f : ws[s * kCbStride + 0] - dw_last;
                const float dal = dwe * Ts - suffix / (1.0f - al + 1e-10f);
                suffix = fmaf(dwe, al * Ts, suffix);
                ws[s * kCbStride + 7] = dal * ws[s * kCbStride + 3];
            }
        }
        __syncthreads();
        if (active) {
            float drgb[3] = {0.f, 0.f, 0.f};
            if (a.d_rgbmap) { drgb[0] = a.d_rgbmap[ray * 3]; drgb[1] = a.d_rgbmap[ray * 3 + 1]; drgb[2] = a.d_rgbmap[ray * 3 + 2]; }
            for (int s = lane; s < S; s += kWave) {
                const long long gpt = ray * S + s;
                const float w = a.weights[gpt];
                a.d_sdf_pts[gpt] = ws[s * kCbStride + 7] + (a.d_sdf_in ? a.d_sdf_in[gpt] : 0.0f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float sc = sigmoid_f32(ws[s * kCbStride + 4 + c]);
                    a.d_rgb_pts[gpt * 3 + c] = 2.0f * drgb[c] * w * sc * (1.0f - sc);
                }
            }
        }
        __syncthreads();
    }
}

// fold the per-workgroup partial sums: dfilm[b][l][gb][n] = sum over the image's slices.  64 elements x 16
// slice groups per block; fixed association order -> bit-reproducible.
constexpr int kRedGroups = 16;
__global__ void __launch_bounds__(64 * kRedGroups)
bwd_reduce_kernel(float* __restrict__ dfilm, const float* __restrict__ partials, int wgs_per_img) {
    __shared__ float part[kRedGroups][64];
    const int b = blockIdx.y;
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;                               // < 9*2*256
    const int n_slices = wgs_per_img;
    const float* p = partials + (int64_t)b * n_slices * (9 * 2 * kWidth) + e;
    float acc = 0.0f;
    for (int s = g; s < n_slices; s += kRedGroups) acc += p[(int64_t)s * (9 * 2 * kWidth)];
    part[g][el] = acc;
    __syncthreads();
    if (g == 0) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < kRedGroups; ++i) t += part[i][el];
        dfilm[(int64_t)b * 9 * 2 * kWidth + e] = t;
    }
}

// The 8-wave kernels (siren16_bwd.h) leave per slice S_a = sum(da a [+ ta r cos a]) in the gamma rows and S_b = sum(da) in the beta rows:
// with z = (a - beta) / gamma,  d gamma = sum(da z [+ ...] / gamma) = (S_a - beta S_b) / gamma,  d beta = S_b.  One thread folds both rows
// of a (layer, feature) over its share of the slices; fixed association order -> bit-reproducible.
// 16 (layer, feature) pairs x 64 slice groups per block: 144 blocks per image (the 8-wave kernels leave one slice per 128-point sub-tile --
// 576 per 64x64x18 image; 36 blocks per image took 130 us at four images per step).
constexpr int kFinPairs = 16, kFinGroups = 64;
__global__ void __launch_bounds__(kFinPairs * kFinGroups)
bwd_reduce_fin_kernel(float* __restrict__ dfilm, const float* __restrict__ partials, const float* __restrict__ film, int n_slices) {
    __shared__ float part[2][kFinGroups][kFinPairs];
    const int b = blockIdx.y;
    const int el = threadIdx.x & (kFinPairs - 1), g = threadIdx.x / kFinPairs;
    const int ln = blockIdx.x * kFinPairs + el;                        // (layer, feature) < 9*256
    const int l = ln >> 8, n = ln & 255;
    const float* p = partials + (int64_t)b * n_slices * (9 * 2 * kWidth) + (l * 2) * kWidth + n;
    float sa = 0.0f, sb = 0.0f;
    int s = g;
    for (; s + 3 * kFinGroups < n_slices; s += 4 * kFinGroups) {          // four slices' loads in flight, added in the same order as one by one
        float va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            va[k] = p[(int64_t)(s + k * kFinGroups) * (9 * 2 * kWidth)];
            vb[k] = p[(int64_t)(s + k * kFinGroups) * (9 * 2 * kWidth) + kWidth];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { sa += va[k]; sb += vb[k]; }
    }
    for (; s < n_slices; s += kFinGroups) {
        sa += p[(int64_t)s * (9 * 2 * kWidth)];
        sb += p[(int64_t)s * (9 * 2 * kWidth) + kWidth];
    }
    part[0][g][el] = sa; part[1][g][el] = sb;
    __syncthreads();
    if (g == 0) {
        float ta = 0.0f, tb = 0.0f;
#pragma unroll 8
        for (int i = 0; i < kFinGroups; ++i) { ta += part[0][i][el]; tb += part[1][i][el]; }
        const float* fb = film + ((int64_t)b * 9 + l) * 2 * kWidth;
        float* o = dfilm + ((int64_t)b * 9 + l) * 2 * kWidth;
        o[n] = __fdiv_rn(ta - fb[kWidth + n] * tb, fb[n]);
        o[kWidth + n] = tb;
    }
}

// d(styles)[b][l][k] = 15 * sum_n Wg_l[n][k] dgamma[b][l][n] + 0.25 * sum_n Wb_l[n][k] dbeta[b][l][n]
// (backward of LinearLayer.forward :76-80; film_params_kernel is the forward)
// Round 6: (batch x 9 layers x 4 column blocks) workgroups instead of (batch x 9) -- this launch ends every backward, 18-35 us with nine
// workgroups streaming 4.7 MB of weights.  Thread (k, n group): 64 of the 256 rows; the four groups are added in fixed order.
__global__ void __launch_bounds__(256)
film_bwd_kernel(float* __restrict__ dstyles, const float* __restrict__ dfilm, const float* __restrict__ wg,
                const float* __restrict__ wb) {
    __shared__ float part[4][64];
    const int kb = blockIdx.x & 3, l = (blockIdx.x >> 2) % 9, b = (blockIdx.x >> 2) / 9;
    const int kk = threadIdx.x & 63, ng = threadIdx.x >> 6;
    const int k = 64 * kb + kk;
    const float* __restrict__ dg = dfilm + ((int64_t)b * 9 + l) * 2 * kWidth;
    const float* __restrict__ db = dg + kWidth;
    const float* __restrict__ Wg = wg + (int64_t)l * kWidth * kWidth;
    const float* __restrict__ Wb = wb + (int64_t)l * kWidth * kWidth;
    float acc = 0.0f;
    for (int n0 = 64 * ng; n0 < 64 * ng + 64; n0 += 16) {              // lanes along k: coalesced rows; 32 loads in flight
        float wgv[16], wbv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { wgv[i] = Wg[(int64_t)(n0 + i) * kWidth + k]; wbv[i] = Wb[(int64_t)(n0 + i) * kWidth + k]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += 15.0f * wgv[i] * dg[n0 + i] + 0.25f * wbv[i] * db[n0 + i];
    }
    part[ng][kk] = acc;
    __syncthreads();
    if (ng == 0) dstyles[((int64_t)b * 9 + l) * kWidth + k] = ((part[0][kk] + part[1][kk]) + part[2][kk]) + part[3][kk];
}

}  // namespace e3dge

#include "siren16_bwd.h"      // the 8-wave x 16-point generation (E3DGE_PREC_F16X3_G2)

using namespace e3dge;

static int bwd_geometry(int batch, int64_t n_pts, int* subtiles_per_wg, int* wgs_per_img) {
    const int64_t tiles = (n_pts + kTilePts - 1) / kTilePts;
    const int spw = pick_subtiles_per_wg(tiles, batch);
    *subtiles_per_wg = spw;
    *wgs_per_img = (int)((tiles + spw - 1) / spw);
    return 0;
}

extern "C" int64_t e3dge_siren_bwd_partial_floats(int batch, int64_t n_pts) {
    if (batch <= 0 || n_pts <= 0) return 0;
    int spw, wpi;
    bwd_geometry(batch, n_pts, &spw, &wpi);
    return (int64_t)batch * wpi * spw * (9 * 2 * kWidth);      // one slice per sub-tile (8-wave kernels); per workgroup (first generation)
}

static int launch_bwd(SirenBwdK k, const float* wg, const float* wb, float* dfilm, float* dstyles, hipStream_t st) {
    const int batch = k.batch;
    const int64_t n_pts = k.n_pts;
    E3DGE_REQUIRE(k.packed && k.film && k.args && wg && wb && k.partials && dfilm && dstyles, "siren_bwd: null pointer");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(k.packed) | reinterpret_cast<uintptr_t>(k.args) | reinterpret_cast<uintptr_t>(k.d_feat) |
                    reinterpret_cast<uintptr_t>(k.d_featmap)) & 15) == 0,
                  "siren_bwd: packed/args/d_feat must be 16-B aligned");
    float* const partials = k.partials;
    E3DGE_REQUIRE(k.precision >= E3DGE_PREC_F32 && k.precision <= E3DGE_PREC_F16X3_G2, "siren_bwd: precision=%d", k.precision);
    const bool gen2 = k.precision == E3DGE_PREC_F16X3_G2;
    // first generation: tang = ta_l and rsave = r_l, both or neither; 8-wave generation: tang = the products ta_l r_l of
    // e3dge_siren_tangent_tr, rsave must be NULL
    if (gen2) E3DGE_REQUIRE(k.rsave == nullptr, "siren_bwd: precision f16x3_g2 takes the products ta*r in `tang` (e3dge_siren_tangent_tr) and no rsave");
    else E3DGE_REQUIRE((k.tang == nullptr) == (k.rsave == nullptr), "siren_bwd: tang and rsave must come together");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(k.tang) | reinterpret_cast<uintptr_t>(k.rsave)) & 15) == 0, "siren_bwd: tang/rsave must be 16-B aligned");
    const bool tex = k.tex_alpha != nullptr;
    E3DGE_REQUIRE(tex == (k.d_tex_alpha != nullptr) && tex == (k.d_tex_beta != nullptr), "siren_bwd: tex_alpha, d_tex_alpha, d_tex_beta must come together");
    E3DGE_REQUIRE(!(tex && k.tang), "siren_bwd: the eikonal double backward is not available on the tex-FiLM pass");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(k.tex_alpha) | reinterpret_cast<uintptr_t>(k.d_tex_alpha) | reinterpret_cast<uintptr_t>(k.d_tex_beta)) & 15) == 0,
                  "siren_bwd: tex pointers must be 16-B aligned");
    typedef void (*KernelFn)(const SirenBwdK);
    // [dpts][eik][f16], then the two tex variants
    static const KernelFn fns[10] = {
        &siren_bwd_kernel<false, false, false, false>, &siren_bwd_kernel<false, true, false, false>,
        &siren_bwd_kernel<true, false, false, false>, &siren_bwd_kernel<true, true, false, false>,
        &siren_bwd_kernel<false, false, false, true>, &siren_bwd_kernel<false, true, false, true>,
        &siren_bwd_kernel<true, false, false, true>, &siren_bwd_kernel<true, true, false, true>,
        &siren_bwd_kernel<false, false, true, false>, &siren_bwd_kernel<false, true, true, false>};
    E3DGE_REQUIRE(!(tex && k.d_pts), "siren_bwd: d_pts is not available on the tex-FiLM pass");
    const int f16 = k.precision != E3DGE_PREC_F32;
    // 8-wave generation (8 waves x 16 points): [dpts][eik], then the tex variant
    static const KernelFn fns16[5] = {
        &siren16_bwd_kernel<false, false, false>, &siren16_bwd_kernel<true, false, false>,
        &siren16_bwd_kernel<false, false, true>, &siren16_bwd_kernel<true, false, true>,
        &siren16_bwd_kernel<false, true, false>};
    const bool eik = k.tang != nullptr, dpts = k.d_pts != nullptr;
    const KernelFn fn = gen2 ? fns16[tex ? 4 : 2 * dpts + eik] : fns[tex ? 8 + f16 : 4 * dpts + 2 * eik + f16];
    const int lds_bytes = gen2 ? b16_lds_bytes(eik ? 2 : 1, dpts) : kBwdLdsBytes;
    k.precision = f16 ? E3DGE_PREC_F16X3 : E3DGE_PREC_F32;
    {   // the attribute is per device (and cheap): set it on the launch's device every time
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(siren_bwd): %s", hipGetErrorString(e));
    }
    bwd_geometry(batch, n_pts, &k.subtiles_per_wg, &k.wgs_per_img);
    if (gen2) E3DGE_REQUIRE((int64_t)k.subtiles_per_wg * kTilePts * 9 * kWidth * 4 < ((int64_t)1 << 31), "siren_bwd: workgroup span exceeds the 32-bit stream offsets");
    if (n_pts > 0) {
        const int64_t grid = (int64_t)k.wgs_per_img * batch;
        E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "siren_bwd: grid too large");
        fn<<<dim3((unsigned)grid), dim3(gen2 ? k16Threads : kThreads), lds_bytes, st>>>(k);
        int rc = check_launch("siren_bwd");
        if (rc) return rc;
    }
    if (gen2)
        bwd_reduce_fin_kernel<<<dim3(9 * kWidth / kFinPairs, (unsigned)batch), dim3(kFinPairs * kFinGroups), 0, st>>>(dfilm, partials, k.film, n_pts > 0 ? k.wgs_per_img * k.subtiles_per_wg : 0);
    else
        bwd_reduce_kernel<<<dim3(9 * 2 * kWidth / 64, (unsigned)batch), dim3(64 * kRedGroups), 0, st>>>(dfilm, partials, n_pts > 0 ? k.wgs_per_img : 0);
    int rc = check_launch("siren_bwd(reduce)");
    if (rc) return rc;
    film_bwd_kernel<<<dim3((unsigned)(batch * 9 * 4)), dim3(256), 0, st>>>(dstyles, dfilm, wg, wb);
    return check_launch("siren_bwd(film)");
}

extern "C" int e3dge_siren_bwd(const E3dgeSirenBwdArgs* r, e3dge_stream_t stream) {
    E3DGE_REQUIRE(r != nullptr, "siren_bwd: null args");
    E3DGE_REQUIRE(r->batch >= 0 && r->n_pts >= 0, "siren_bwd: bad sizes");
    if (r->batch == 0) return E3DGE_OK;
    SirenBwdK k{};
    k.packed = r->packed; k.film = r->film; k.args = r->args; k.d_feat = r->d_feat; k.d_rgb = r->d_rgb; k.d_sdf = r->d_sdf;
    k.partials = r->partials; k.n_pts = r->n_pts; k.batch = r->batch; k.samples = 1; k.tang = r->tang; k.rsave = r->rsave;
    k.precision = r->precision; k.d_pts = r->d_pts; k.box_scale = r->box_scale;
    k.tex_alpha = r->tex_alpha; k.d_tex_alpha = r->d_tex_alpha; k.d_tex_beta = r->d_tex_beta;
    return launch_bwd(k, r->wg, r->wb, r->dfilm, r->dstyles, as_stream(stream));
}

extern "C" int e3dge_siren_render_bwd(const E3dgeRenderBwdArgs* r, e3dge_stream_t stream) {
    E3DGE_REQUIRE(r != nullptr, "siren_render_bwd: null args");
    E3DGE_REQUIRE(r->batch >= 0 && r->height > 0 && r->width > 0, "siren_render_bwd: bad image extent");
    if (r->batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(r->n_samples >= 1 && r->n_samples <= 1024, "siren_render_bwd: n_samples=%d outside [1, 1024]", r->n_samples);
    E3DGE_REQUIRE(r->packed && r->film && r->args && r->sdf && r->dists && r->points && r->weights && r->t_vals &&
                  r->near && r->far && r->d_rgb_pts && r->d_sdf_pts, "siren_render_bwd: null pointer");
    E3DGE_REQUIRE(r->sigmoid_beta != 0.0f, "siren_render_bwd: sigmoid_beta must be non-zero");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(r->d_feat_map) | reinterpret_cast<uintptr_t>(r->args)) & 15) == 0,
                  "siren_render_bwd: args/d_feat_map must be 16-B aligned");
    hipStream_t st = as_stream(stream);
    const int64_t HW = (int64_t)r->height * r->width;
    CompositeBwdK c{};
    c.packed = r->packed; c.args = r->args; c.sdf = r->sdf; c.dists = r->dists; c.points = r->points; c.weights = r->weights;
    c.t_vals = r->t_vals; c.near = r->near; c.far = r->far;
    c.d_rgbmap = r->d_rgb_map; c.d_featmap = r->d_feat_map; c.d_xyzmap = r->d_xyz_map; c.d_depthmap = r->d_depth_map;
    c.d_sdf_in = r->d_sdf; c.d_weights = r->d_weights; c.d_rgb_pts = r->d_rgb_pts; c.d_sdf_pts = r->d_sdf_pts;
    c.sigmoid_beta = r->sigmoid_beta; c.S = r->n_samples; c.force_bg = r->force_background;
    c.n_rays = HW * r->batch; c.rays_per_img = HW;
    c.args_blocked = (r->precision == E3DGE_PREC_F16X3_G2 && kT3Blocked) ? 1 : 0;
    const size_t lds = (size_t)4 * r->n_samples * kCbStride * sizeof(float);
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 1024 * kCbStride * 4);
        if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(composite_bwd): %s", hipGetErrorString(e));
    }
    int64_t grid = (c.n_rays + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
    E3DGE_REQUIRE(r->phase >= 0 && r->phase <= 2, "siren_render_bwd: phase=%d (0 = both launches, 1 = compositing only, 2 = network only)", r->phase);
    if (r->phase != 2) {
        composite_bwd_kernel<<<dim3((unsigned)grid), dim3(kThreads), lds, st>>>(c);
        int rc = check_launch("siren_render_bwd(composite)");
        if (rc) return rc;
    }
    if (r->phase == 1) return E3DGE_OK;
    SirenBwdK k{};
    k.packed = r->packed; k.film = r->film; k.args = r->args; k.d_feat = nullptr; k.d_rgb = r->d_rgb_pts; k.d_sdf = r->d_sdf_pts;
    k.d_featmap = r->d_feat_map; k.weights = r->weights; k.samples = r->n_samples;
    k.partials = r->partials; k.n_pts = HW * r->n_samples; k.batch = r->batch; k.tang = r->tang; k.rsave = r->rsave; k.precision = r->precision;
    k.tex_alpha = r->tex_alpha; k.d_tex_alpha = r->d_tex_alpha; k.d_tex_beta = r->d_tex_beta;
    return launch_bwd(k, r->wg, r->wb, r->dfilm, r->dstyles, st);
}

template <bool TANGENT>
static int launch_chain(const float* packed, const float* film, const float* args, const float* seed, const float* rmul, float box_scale,
                        int batch, int64_t n_pts, float* save, float* eik, int precision, hipStream_t st, const char* what) {
    E3DGE_REQUIRE(precision >= E3DGE_PREC_F32 && precision <= E3DGE_PREC_F16X3_G2, "%s: precision=%d", what, precision);
    E3DGE_REQUIRE(batch >= 0 && n_pts >= 0, "%s: bad sizes", what);
    if (batch == 0 || n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(packed && film && args && save && (TANGENT ? seed != nullptr : eik != nullptr), "%s: null pointer", what);
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(args) | reinterpret_cast<uintptr_t>(save) |
                    reinterpret_cast<uintptr_t>(rmul)) & 15) == 0, "%s: packed/args/save/rsave must be 16-B aligned", what);
    const bool gen2 = precision == E3DGE_PREC_F16X3_G2;
    E3DGE_REQUIRE(rmul == nullptr || (TANGENT && gen2), "%s: the product form (ta * r) exists for the tangent pass in precision f16x3_g2 only", what);
    const bool tr = rmul != nullptr;
    typedef void (*ChainFn)(const SirenChainK);
    ChainFn fn;
    int lds_bytes = kChLdsBytes, threads = kThreads;
    if (gen2) {
        if (TANGENT) fn = tr ? &siren16_chain_kernel<TANGENT, TANGENT> : &siren16_chain_kernel<TANGENT, false>;
        else fn = &siren16_chain_kernel<TANGENT, false>;
        lds_bytes = c16_lds_bytes(tr ? 2 : 1); threads = k16Threads;
    } else {
        fn = (precision != E3DGE_PREC_F32) ? &siren_chain_kernel<TANGENT, true> : &siren_chain_kernel<TANGENT, false>;
    }
    {   // per device, cheap: set on every launch
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    }
    SirenChainK k{};
    k.packed = packed; k.film = film; k.args = args; k.seed = seed; k.save = save; k.rmul = rmul; k.eik = eik; k.box_scale = box_scale;
    k.n_pts = n_pts; k.batch = batch;
    bwd_geometry(batch, n_pts, &k.subtiles_per_wg, &k.wgs_per_img);
    if (gen2) E3DGE_REQUIRE((int64_t)k.subtiles_per_wg * kTilePts * 9 * kWidth * 4 < ((int64_t)1 << 31), "%s: workgroup span exceeds the 32-bit stream offsets", what);
    const int64_t grid = (int64_t)k.wgs_per_img * batch;
    E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "%s: grid too large", what);
    fn<<<dim3((unsigned)grid), dim3(threads), lds_bytes, st>>>(k);
    return check_launch(what);
}

extern "C" int e3dge_siren_sdf_grad(const float* packed, const float* film, const float* args, const float* seed,
                                    float box_scale, int batch, int64_t n_pts, float* rsave, float* eik,
                                    int precision, e3dge_stream_t stream) {
    return launch_chain<false>(packed, film, args, seed, nullptr, box_scale, batch, n_pts, rsave, eik, precision, as_stream(stream), "siren_sdf_grad");
}

extern "C" int e3dge_siren_tangent(const float* packed, const float* film, const float* args, const float* v,
                                   float box_scale, int batch, int64_t n_pts, float* tang, int precision,
                                   e3dge_stream_t stream) {
    return launch_chain<true>(packed, film, args, v, nullptr, box_scale, batch, n_pts, tang, nullptr, precision, as_stream(stream), "siren_tangent");
}

extern "C" int e3dge_siren_tangent_tr(const float* packed, const float* film, const float* args, const float* v, const float* rsave,
                                      float box_scale, int batch, int64_t n_pts, float* tr, int precision, e3dge_stream_t stream) {
    E3DGE_REQUIRE(rsave != nullptr, "siren_tangent_tr: rsave is required");
    return launch_chain<true>(packed, film, args, v, rsave, box_scale, batch, n_pts, tr, nullptr, precision, as_stream(stream), "siren_tangent_tr");
}
