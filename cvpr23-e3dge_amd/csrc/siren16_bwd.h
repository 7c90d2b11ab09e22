// Backward-direction kernels of the FiLM-SIREN MLP, second generation: 8 waves per workgroup (two per SIMD), 16 points per
// wave, v_mfma_f32_16x16x32_f16 on block-scaled split-f16 operands -- the layout of the forward kernel (siren16.h) applied to
//   siren16_bwd_kernel   : d(film), d(styles) [, d(points), d(texture FiLM)] of a loss on (feat, rgb, sdf) [and on the eikonal term]
//   siren16_chain_kernel : the two first-order chains the eikonal term needs (sdf chain / tangent)
// Included by siren_bwd.hip, which documents the mathematics and the reference lines (volume_renderer.py:168-264, :796-802) at
// the first-generation kernels; those stay as the fp32 path and as E3DGE_PREC_F16X3_V1.
//
// Why a second generation: the first one (one wave per SIMD, 32 points per wave) is VALU-bound around its GEMM -- per 32x32
// tile ~615 VALU instructions (FiLM / cosine epilogue, lane reductions, operand scaling) = 2.5 k cycles next to 1.5 k cycles
// of MFMA and 0.9 k of LDS fragment returns, and a single wave cannot overlap the three (profiles/r1_v6_bwd_*: MFMA pipe 25 %
// busy).  With 16 points per wave the register-resident state halves (64 packed-f16 operand registers + 64 fp32 result
// registers), two waves share a SIMD and one wave's epilogue runs under the other's MFMAs.
//
// Layout (siren16.h): lane l: point n = l & 15, q = l >> 4; register r of 16-feature tile t = feature 16t + 4q + r of point n.
// Every stream (saved arguments, tangent arguments, r, texture alpha, d_feat) is therefore one 16-B load per lane and tile.
// Sums over points (d gamma, d beta) = sums over the 16 lanes of a row: four DPP row rotations per value; every wave leaves
// its per-layer sums in its own LDS slice and the workgroup folds the eight slices in fixed order after each layer --
// deterministic, no atomics.
#pragma once
#define E3DGE_16_HELPERS_ONLY
#include "siren16.h"

namespace e3dge {

constexpr int kB16LdsW = 0;
constexpr int kB16LdsFilm = kB16LdsW + k16NBuf * k16ChunkFloats;     // [9][3][256] gamma, beta, 1/gamma
constexpr int kB16LdsHead = kB16LdsFilm + 9 * 3 * kWidth;            // w_sigma[256], w_rgb[3][256]
constexpr int kB16LdsAcc = kB16LdsHead + 4 * kWidth;                 // [9][2][256] this workgroup's d(gamma), d(beta)
constexpr int kB16LdsWave = kB16LdsAcc + 9 * 2 * kWidth;             // [8 waves][256][2] one layer's sums of each wave
constexpr int kB16LdsW0 = kB16LdsWave + 8 * 2 * kWidth;              // [3][256] first-layer weights, column-major (d_pts)
constexpr int kB16LdsFloats = kB16LdsW0 + 3 * kWidth;
constexpr int kB16LdsBytes = kB16LdsFloats * 4;
static_assert(kB16LdsBytes <= 160 * 1024, "LDS budget");
static_assert(k16Chunks % k16NBuf == 0 && (7 * k16Tiles) % k16NBuf == 0 && k16Tiles % k16NBuf == 0, "static buffer index = tile index mod k16NBuf");

constexpr int kC16LdsW = 0;
constexpr int kC16LdsFilm = kC16LdsW + k16NBuf * k16ChunkFloats;     // [8][256] gamma of the backbone layers
constexpr int kC16LdsW0 = kC16LdsFilm + 8 * kWidth;                  // [3][256]
constexpr int kC16LdsHead = kC16LdsW0 + 3 * kWidth;                  // w_sigma[256]
constexpr int kC16LdsFloats = kC16LdsHead + kWidth;
constexpr int kC16LdsBytes = kC16LdsFloats * 4;

constexpr int kB16Ring = 2;                       // k-steps of weight fragments held (registers are the scarce resource here)

__device__ __forceinline__ f32x4v ld4(const float* p) { return *reinterpret_cast<const f32x4v*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4v& v) { *reinterpret_cast<f32x4v*>(p) = v; }

// Split-f16 operand of a backward-type GEMM (see scale_split in siren_bwd.hip): the 256 values of a point are spread over the
// four lanes n, n+16, n+32, n+48; one power-of-two scale per point brings the largest into [1, 2).  `m` = max |value| over
// this lane's 64 values.  Returns 1 / (kW16Scale * scale) for the epilogue of the GEMM that consumes the operand.
__device__ __forceinline__ float scale_split16(const f32x4v (&src)[k16Tiles], u32x4 (&dH)[k16Steps], u32x4 (&dL)[k16Steps], float m) {
    m = fmaxf(m, __shfl_xor(m, 16, kWave));
    m = fmaxf(m, __shfl_xor(m, 32, kWave));
    const unsigned e = min((__float_as_uint(m) >> 23) & 255u, 254u);    // m in [2^(e-127), 2^(e-126)); inf/nan: scale 0 -> NaN out
    const float sc = __uint_as_float((254u - e) << 23);                 // m * sc in [1, 2)   (m == 0: sc = 2^127, harmless)
    const float inv = __uint_as_float((e > 8u ? e - 7u : 1u) << 23);    // 1 / (128 * sc) = 2^(e-134)
#pragma unroll
    for (int t = 0; t < k16Tiles; ++t) {
        SPLIT2_TO(src[t][0] * sc, src[t][1] * sc, dH[t >> 1][2 * (t & 1)], dL[t >> 1][2 * (t & 1)]);
        SPLIT2_TO(src[t][2] * sc, src[t][3] * sc, dH[t >> 1][2 * (t & 1) + 1], dL[t >> 1][2 * (t & 1) + 1]);
    }
    return inv;
}

__device__ __forceinline__ void sincos_hw16(float x, float& sn, float& cs) {
    const float r = revolutions_f32(x);
    sn = __builtin_amdgcn_sinf(r);
    cs = __builtin_amdgcn_cosf(r);
}

// EIK / TEX / DPTS as in siren_bwd_kernel.
template <bool EIK, bool TEX, bool DPTS>
__global__ void __launch_bounds__(k16Threads) siren16_bwd_kernel(const SirenBwdK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kB16LdsW;
    float* const film_s = smem + kB16LdsFilm;
    float* const head_s = smem + kB16LdsHead;
    float* const acc_s = smem + kB16LdsAcc;
    float* const wave_s = smem + kB16LdsWave;
    float* const w0_s = smem + kB16LdsW0;

    const int tid_k = threadIdx.x;
    const int b = blockIdx.x / a.wgs_per_img;
    const int wg = blockIdx.x - b * a.wgs_per_img;
    const long long pt0 = (long long)wg * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;

    const float* __restrict__ packed = a.packed;
    const float* __restrict__ film_g = a.film + (int64_t)b * 9 * 2 * kWidth;
    for (int i = tid_k; i < 9 * kWidth; i += k16Threads) {
        const int l = i >> 8, n = i & 255;
        const float g = film_g[(l * 2) * kWidth + n];
        film_s[(l * 3) * kWidth + n] = g;
        film_s[(l * 3 + 1) * kWidth + n] = film_g[(l * 2 + 1) * kWidth + n];
        film_s[(l * 3 + 2) * kWidth + n] = 1.0f / g;
    }
    for (int i = tid_k; i < 4 * kWidth; i += k16Threads) head_s[i] = packed[kOffWSigma + i];
    if (DPTS) for (int i = tid_k; i < 3 * kWidth; i += k16Threads) {
        const int c = i >> 8, n = i & 255;           // fragment image of layer 0 (siren_pack_kernel): [t][m][lane], k = 2m + half
        w0_s[i] = packed[kOffFirst + ((n >> 5) * 2 + (c >> 1)) * 64 + (c & 1) * 32 + (n & 31)];
    }
    for (int i = tid_k; i < 9 * 2 * kWidth; i += k16Threads) acc_s[i] = 0.0f;

    ChunkPipe16 pipe;
    pipe.init(wbuf, packed + kOffBigT16b, tid_k >> 6, tid_k & 63);
    pipe.prime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4 ringH[kB16Ring], ringL[kB16Ring];
    {
        const int lane0 = tid_k & 63;
#pragma unroll
        for (int g = 0; g < kB16Ring - 1; ++g) {
            ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + lane0];
            ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + lane0];
        }
    }

    u32x4 inH[k16Steps], inL[k16Steps];            // the GEMM operand g_L, block-scaled packed f16 (hi, lo)
    f32x4v out[k16Tiles];                          // g_{L-1} being produced, fp32 until the point's maximum is known
    float inv_scale = 1.0f, gmax = 0.0f;

    for (int sub = 0; sub < n_sub; ++sub) {
        int tid_o = tid_k;
        asm volatile("" : "+v"(tid_o));            // opaque: address math stays inside the sub-tile (no hoisted registers)
        const int tid = tid_o, lane = tid & 63, wave = tid >> 6, q = lane >> 4, col = lane & 15;
        const int p = sub * kTilePts + 16 * wave + col;
        const bool valid = p < npts;
        const int pc = valid ? p : (npts - 1);
        const int64_t gpt = (int64_t)b * a.n_pts + pt0 + pc;
#ifdef E3DGE_B16_ABL_HOTARGS      // ablation: every lane reads the same (cache-resident) row: is the kernel waiting for HBM?
        const float* __restrict__ ap = a.args + ((int64_t)b * a.n_pts + pt0) * (9 * kWidth) + 4 * q;
#else
        const float* __restrict__ ap = a.args + gpt * (9 * kWidth) + 4 * q;
#endif
        const float* __restrict__ tp_ = EIK ? a.tang + gpt * (8 * kWidth) + 4 * q : nullptr;
        const float* __restrict__ rp_ = EIK ? a.rsave + gpt * (8 * kWidth) + 4 * q : nullptr;
        const float vmask = valid ? 1.0f : 0.0f;                       // padded lanes contribute nothing
        const float* __restrict__ txa = TEX ? a.tex_alpha + gpt * kWidth + 4 * q : nullptr;
        float* __restrict__ dta = TEX ? a.d_tex_alpha + gpt * kWidth + 4 * q : nullptr;
        float* __restrict__ dtb = TEX ? a.d_tex_beta + gpt * kWidth + 4 * q : nullptr;
        const float dsdf = (a.d_sdf && valid) ? a.d_sdf[gpt] : 0.0f;
        gmax = 0.0f;
        float drgb[3] = {0.f, 0.f, 0.f};
        if (a.d_rgb && valid) { drgb[0] = a.d_rgb[gpt * 3]; drgb[1] = a.d_rgb[gpt * 3 + 1]; drgb[2] = a.d_rgb[gpt * 3 + 2]; }

        // d(beta) += da, d(gamma) += da * z over this wave's 16 points for the 16 features of tile t: row sums, then the lanes
        // n < 4 of every row publish value n into this wave's slice
        auto reduce_store = [&](int t, const float (&rb)[4], const float (&rg)[4]) {
            // slice layout [feature][gamma, beta].  The address is recomputed from the thread index at every use (a few VALU ops
            // under the other wave's MFMAs): kept live across the tile it gets spilled, and the reload's s_waitcnt vmcnt(0) in the
            // middle of a tile would also drain the stream loads and the weight DMA.
#ifdef E3DGE_B16_ABL_NOREDUCE     // ablation: no lane reductions / LDS slices (results wrong)
            gmax += (rb[0] + rb[1] + rb[2] + rb[3] + rg[0] + rg[1] + rg[2] + rg[3]) * 1e-38f;
            return;
#endif
            int tid_r = tid_k;
            asm volatile("" : "+v"(tid_r));
            float* const my_ws = wave_s + (tid_r >> 6) * (2 * kWidth) + 2 * (((tid_r >> 4) & 3) * 4 + (tid_r & 3));
            float sb[4], sg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { sb[r] = row_sum16(rb[r]); sg[r] = row_sum16(rg[r]); }
            const int i = col & 3;
            const float vb = i == 0 ? sb[0] : i == 1 ? sb[1] : i == 2 ? sb[2] : sb[3];
            const float vg = i == 0 ? sg[0] : i == 1 ? sg[1] : i == 2 ? sg[2] : sg[3];
            if (col < 4) *reinterpret_cast<float2*>(my_ws + 32 * t) = make_float2(vg, vb);   // one ds_write_b64, immediate offset
        };
        // after a workgroup barrier: thread (feature, quantity) adds the eight waves' sums of the finished layer in fixed order
        auto fold = [&](int layer) {
            const float* sp = wave_s + tid;                            // tid = 2 * feature + quantity
            float s = sp[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) s += sp[w * 2 * kWidth];
            acc_s[layer * 2 * kWidth + (tid & 1) * kWidth + (tid >> 1)] += s;
        };
        auto next_operand = [&]() {
            inv_scale = scale_split16(out, inH, inL, gmax);
            gmax = 0.0f;
        };

        // =====================================================================================
        // 1. view layer: dh_view = d_feat + Wrgb^T d_rgb ; g8 = gamma8 * dh_view * cos(arg8)
        // =====================================================================================
        if (sub > 0) __syncthreads();                                  // the previous sub-tile's last fold has read wave_s
        {
            const float* __restrict__ fg = film_s + 8 * 3 * kWidth + 4 * q;
            const float* __restrict__ wr = head_s + kWidth + 4 * q;
            const float* __restrict__ df = a.d_feat ? a.d_feat + gpt * kWidth + 4 * q : nullptr;
            float wfeat = 1.0f;
            if (a.d_featmap) {
                df = a.d_featmap + (gpt / a.samples) * kWidth + 4 * q;
                wfeat = a.weights[gpt];
            }
#pragma unroll
            for (int t4 = 0; t4 < k16Tiles; t4 += 4) {
                f32x4v arb[4], dfb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    arb[u] = ld4(ap + 8 * kWidth + 16 * (t4 + u));
                    dfb[u] = df ? ld4(df + 16 * (t4 + u)) : zero4();
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t4 + u, o = 16 * t;
                    const f32x4v g4 = ld4(fg + o), b4 = ld4(fg + kWidth + o), i4 = ld4(fg + 2 * kWidth + o);
                    const f32x4v w0 = ld4(wr + o), w1 = ld4(wr + kWidth + o), w2 = ld4(wr + 2 * kWidth + o);
                    float rb[4], rg[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dh = vmask * (wfeat * dfb[u][r] + w0[r] * drgb[0] + w1[r] * drgb[1] + w2[r] * drgb[2]);
                        const float da = dh * cos_hw_f32(arb[u][r]);
                        rb[r] = da;
                        rg[r] = da * ((arb[u][r] - b4[r]) * i4[r]);
                        out[t][r] = g4[r] * da;
                        gmax = fmaxf(gmax, fabsf(out[t][r]));
                    }
                    reduce_store(t, rb, rg);
                }
            }
            next_operand();
            __syncthreads();
            fold(8);
        }

        // =====================================================================================
        // 2. the chain: GEMM Gb (layer L = 8 - Gb) turns g_L into dh_{L-1}; its epilogue makes g_{L-1}
        // =====================================================================================
#pragma unroll 1
        for (int Gb = 0; Gb < kBigLayers; ++Gb) {
            const int Lm1 = 7 - Gb;                                      // layer whose argument / FiLM the epilogue uses
            const float* __restrict__ fg = film_s + Lm1 * 3 * kWidth + 4 * q;
            const float* __restrict__ apl = ap + Lm1 * kWidth;
            const float* __restrict__ tpl = EIK ? tp_ + Lm1 * kWidth : nullptr;
            const float* __restrict__ rpl = EIK ? rp_ + Lm1 * kWidth : nullptr;
            const float sdf_term = (Gb == 0) ? dsdf : 0.0f;              // the sdf head reads the backbone output h8
            const bool tex_here = TEX && Gb == 0;                        // this GEMM's result is dL/dh8' (view-layer input)
            f32x4v prev = zero4();
            // streams of the tile whose epilogue runs inside the NEXT GEMM tile: fetched one tile ahead, two buffers
            f32x4v arg2[2], tg2[2], rs2[2], al2[2];
            f32x4v e_g = zero4(), e_b = zero4(), e_i = zero4(), e_w = zero4(), e_da = zero4(), e_db = zero4();
            float rb[4], rg[4];
            auto epi_load = [&](int tp) {
                const int o = 16 * tp;
                e_g = ld4(fg + o); e_b = ld4(fg + kWidth + o); e_i = ld4(fg + 2 * kWidth + o); e_w = ld4(head_s + 4 * q + o);
            };
            auto epi_val = [&](int tp, int r) {                           // tp, r: compile-time constants at every call site
                const float ar = arg2[tp & 1][r];
                float xin = prev[r];
                float sn = 0.f, cs;
                if (EIK || tex_here) sincos_hw16(ar, sn, cs);
                else cs = cos_hw_f32(ar);
                if (tex_here) { e_da[r] = xin * sn; e_db[r] = xin; xin = __fadd_rn(al2[tp & 1][r], 1.0f) * xin; }
                const float dh = fmaf(e_w[r], sdf_term, xin);            // padded lanes: operand 0 and dsdf = 0, so dh = 0
                float da, dg_extra = 0.0f;
                if (EIK) {
                    const float tr = vmask * tg2[tp & 1][r] * rs2[tp & 1][r];
                    da = fmaf(dh, cs, -sn * tr);
                    dg_extra = tr * e_i[r] * cs;
                } else {
                    da = dh * cs;
                }
                rb[r] = da;
                rg[r] = fmaf(da, (ar - e_b[r]) * e_i[r], dg_extra);
                out[tp][r] = e_g[r] * da;
                gmax = fmaxf(gmax, fabsf(out[tp][r]));
            };
            auto epi_finish = [&](int tp) {
                if (tex_here && valid) { st4(dta + 16 * tp, e_da); st4(dtb + 16 * tp, e_db); }
                reduce_store(tp, rb, rg);
            };
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                // after k-step 1: chunk wait + barrier; the streams of tile t-1 (issued one tile ago) have landed with it -- the
                // empty asm makes the compiler place ITS wait for them here, where it is free, instead of in front of their first
                // use further down (there it would also drain the loads issued below and the weight DMA: the compiler does not
                // see the DMA in vmcnt).  Then this tile's streams, then the next weight chunk.
                auto hook = [&]() {
                    pipe.template sync<true>();
                    if (t > 0) {
                        asm volatile("" : "+v"(arg2[(t - 1) & 1]));
                        if (EIK) { asm volatile("" : "+v"(tg2[(t - 1) & 1])); asm volatile("" : "+v"(rs2[(t - 1) & 1])); }
                        if (TEX) asm volatile("" : "+v"(al2[(t - 1) & 1]));
                    }
                    arg2[t & 1] = ld4(apl + 16 * t);
                    if (EIK) { tg2[t & 1] = ld4(tpl + 16 * t); rs2[t & 1] = ld4(rpl + 16 * t); }
                    if (TEX) al2[t & 1] = tex_here ? ld4(txa + 16 * t) : zero4();
                    pipe.issue_chunk();
                };
                f32x4v acc = zero4(), accb = zero4();
                if (t == 0) {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [](int) {}, hook, t % k16NBuf);
                } else {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [&](int g) {
                        if (g == 0) epi_load(t - 1);
                        else if (g <= 4) epi_val(t - 1, g - 1);
                        else if (g == 5) epi_finish(t - 1);
                    }, hook, t % k16NBuf);
                }
                pipe.advance();
                prev = (acc + accb) * inv_scale;
            }
            asm volatile("" : "+v"(arg2[(k16Tiles - 1) & 1]));
            if (EIK) { asm volatile("" : "+v"(tg2[(k16Tiles - 1) & 1])); asm volatile("" : "+v"(rs2[(k16Tiles - 1) & 1])); }
            if (TEX) asm volatile("" : "+v"(al2[(k16Tiles - 1) & 1]));
            epi_load(k16Tiles - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_val(k16Tiles - 1, r);
            epi_finish(k16Tiles - 1);
            if (Gb + 1 < kBigLayers) next_operand();
            __syncthreads();
            fold(Lm1);
        }
        // ---- optional: dL/dx = s W_0^T g_0 (g_0 = gamma_0 * adj(a_0) is in out[]) ----
        if (DPTS) {
            float ex = 0.f, ey = 0.f, ez = 0.f;
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                const int o = 16 * t + 4 * q;
                const f32x4v wx = ld4(w0_s + o), wy = ld4(w0_s + kWidth + o), wz = ld4(w0_s + 2 * kWidth + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = out[t][r];
                    ex = fmaf(wx[r], g, ex); ey = fmaf(wy[r], g, ey); ez = fmaf(wz[r], g, ez);
                }
            }
            ex += __shfl_xor(ex, 16, kWave); ex += __shfl_xor(ex, 32, kWave);
            ey += __shfl_xor(ey, 16, kWave); ey += __shfl_xor(ey, 32, kWave);
            ez += __shfl_xor(ez, 16, kWave); ez += __shfl_xor(ez, 32, kWave);
            if (valid && q == 0) {
                float* o = a.d_pts + gpt * 3;
                o[0] = ex * a.box_scale; o[1] = ey * a.box_scale; o[2] = ez * a.box_scale;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* const my_partial = a.partials + (int64_t)blockIdx.x * (9 * 2 * kWidth);
    for (int i = tid_k; i < 9 * 2 * kWidth; i += k16Threads) my_partial[i] = acc_s[i];
}

// TANGENT as in siren_chain_kernel.  Seven GEMMs: forward image, layers 1..7 (tangent) or transposed image, layers 7..1 (sdf).
template <bool TANGENT>
__global__ void __launch_bounds__(k16Threads) siren16_chain_kernel(const SirenChainK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kC16LdsW;
    float* const gam_s = smem + kC16LdsFilm;
    float* const w0_s = smem + kC16LdsW0;
    float* const ws_s = smem + kC16LdsHead;

    const int tid_k = threadIdx.x;
    const int b = blockIdx.x / a.wgs_per_img;
    const int wg = blockIdx.x - b * a.wgs_per_img;
    const long long pt0 = (long long)wg * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;

    const float* __restrict__ packed = a.packed;
    const float* __restrict__ film_g = a.film + (int64_t)b * 9 * 2 * kWidth;
    for (int i = tid_k; i < 8 * kWidth; i += k16Threads) gam_s[i] = film_g[((i >> 8) * 2) * kWidth + (i & 255)];
    for (int i = tid_k; i < 3 * kWidth; i += k16Threads) {
        const int c = i >> 8, n = i & 255;
        w0_s[i] = packed[kOffFirst + ((n >> 5) * 2 + (c >> 1)) * 64 + (c & 1) * 32 + (n & 31)];
    }
    for (int i = tid_k; i < kWidth; i += k16Threads) ws_s[i] = packed[kOffWSigma + i];

    constexpr int kChainChunks = 7 * k16Tiles;
    ChunkPipe16 pipe;
    // tangent: hidden layers 1..7 are the first 7 layers of the forward image; sdf chain: skip the view layer's transposed chunks
    pipe.init(wbuf, packed + (TANGENT ? kOffBig16b : kOffBigT16b + (int64_t)k16Tiles * k16ChunkFloats), tid_k >> 6, tid_k & 63, kChainChunks);
    pipe.prime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4 ringH[kB16Ring], ringL[kB16Ring];
    {
        const int lane0 = tid_k & 63;
#pragma unroll
        for (int g = 0; g < kB16Ring - 1; ++g) {
            ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + lane0];
            ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + lane0];
        }
    }
    u32x4 inH[k16Steps], inL[k16Steps];
    f32x4v out[k16Tiles];
    float inv_scale = 1.0f, gmax = 0.0f;

    for (int sub = 0; sub < n_sub; ++sub) {
        int tid_o = tid_k;
        asm volatile("" : "+v"(tid_o));
        const int tid = tid_o, lane = tid & 63, wave = tid >> 6, q = lane >> 4, col = lane & 15;
        const int p = sub * kTilePts + 16 * wave + col;
        const bool valid = p < npts;
        const int pc = valid ? p : (npts - 1);
        const int64_t gpt = (int64_t)b * a.n_pts + pt0 + pc;
#ifdef E3DGE_B16_ABL_HOTARGS
        const float* __restrict__ ap = a.args + ((int64_t)b * a.n_pts + pt0) * (9 * kWidth) + 4 * q;
#else
        const float* __restrict__ ap = a.args + gpt * (9 * kWidth) + 4 * q;
#endif
        float* __restrict__ sp = a.save + gpt * (8 * kWidth) + 4 * q;
        gmax = 0.0f;

        // ---- first layer of the chain (no GEMM) ----
        {
            const int l0 = TANGENT ? 0 : 7;
            const float* __restrict__ gl = gam_s + l0 * kWidth + 4 * q;
            float sx = 0.f, sy = 0.f, sz = 0.f, seed = 1.0f;
            if (TANGENT) {
                const float* vv = a.seed + gpt * 3;
                sx = vv[0] * a.box_scale; sy = vv[1] * a.box_scale; sz = vv[2] * a.box_scale;
            } else if (a.seed) {
                seed = a.seed[gpt];
            }
#pragma unroll
            for (int t4 = 0; t4 < k16Tiles; t4 += 4) {
                f32x4v arb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) arb[u] = ld4(ap + l0 * kWidth + 16 * (t4 + u));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t4 + u, o = 16 * t;
                    const f32x4v g4 = ld4(gl + o);
                    f32x4v x4;
                    if (TANGENT) {
                        const f32x4v wx = ld4(w0_s + 4 * q + o), wy = ld4(w0_s + kWidth + 4 * q + o), wz = ld4(w0_s + 2 * kWidth + 4 * q + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) x4[r] = g4[r] * fmaf(wz[r], sz, fmaf(wy[r], sy, wx[r] * sx));   // ta_0
                    } else {
                        const f32x4v w4 = ld4(ws_s + 4 * q + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) x4[r] = w4[r] * seed;                                           // r_7
                    }
                    if (valid) st4(sp + l0 * kWidth + o, x4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        out[t][r] = cos_hw_f32(arb[u][r]) * (TANGENT ? x4[r] : g4[r] * x4[r]);
                        gmax = fmaxf(gmax, fabsf(out[t][r]));
                    }
                }
            }
            inv_scale = scale_split16(out, inH, inL, gmax);
            gmax = 0.0f;
        }

        // ---- seven GEMMs ----
#pragma unroll 1
        for (int step = 0; step < 7; ++step) {
            const int l = TANGENT ? step + 1 : 6 - step;                 // layer whose argument / gamma the epilogue uses
            const float* __restrict__ gl = gam_s + l * kWidth + 4 * q;
            const float* __restrict__ apl = ap + l * kWidth;
            float* __restrict__ spl = sp + l * kWidth;
            f32x4v prev = zero4();
            f32x4v arg2[2];
            f32x4v e_g = zero4(), e_st = zero4();
            auto epi_val = [&](int tp, int r) {
                const float ga = e_g[r] * prev[r];
                e_st[r] = TANGENT ? ga : prev[r];
                out[tp][r] = cos_hw_f32(arg2[tp & 1][r]) * ga;
                gmax = fmaxf(gmax, fabsf(out[tp][r]));
            };
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                auto hook = [&]() {                                      // see siren16_bwd_kernel
                    pipe.template sync<true>();
                    if (t > 0) asm volatile("" : "+v"(arg2[(t - 1) & 1]));
                    arg2[t & 1] = ld4(apl + 16 * t);
                    pipe.issue_chunk();
                };
                f32x4v acc = zero4(), accb = zero4();
                if (t == 0) {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [](int) {}, hook, t % k16NBuf);
                } else {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [&](int g) {
                        if (g == 0) e_g = ld4(gl + 16 * (t - 1));
                        else if (g <= 4) epi_val(t - 1, g - 1);
                        else if (g == 5) { if (valid) st4(spl + 16 * (t - 1), e_st); }
                    }, hook, t % k16NBuf);
                }
                pipe.advance();
                prev = (acc + accb) * inv_scale;
            }
            asm volatile("" : "+v"(arg2[(k16Tiles - 1) & 1]));
            e_g = ld4(gl + 16 * (k16Tiles - 1));
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_val(k16Tiles - 1, r);
            if (valid) st4(spl + 16 * (k16Tiles - 1), e_st);
            if (step < 6) {
                inv_scale = scale_split16(out, inH, inL, gmax);
                gmax = 0.0f;
            }
        }

        // ---- sdf chain: e = s W_0^T g_0 (g_0 is in out[]) ----
        if (!TANGENT) {
            float ex = 0.f, ey = 0.f, ez = 0.f;
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                const int o = 16 * t + 4 * q;
                const f32x4v wx = ld4(w0_s + o), wy = ld4(w0_s + kWidth + o), wz = ld4(w0_s + 2 * kWidth + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = out[t][r];
                    ex = fmaf(wx[r], g, ex); ey = fmaf(wy[r], g, ey); ez = fmaf(wz[r], g, ez);
                }
            }
            ex += __shfl_xor(ex, 16, kWave); ex += __shfl_xor(ex, 32, kWave);
            ey += __shfl_xor(ey, 16, kWave); ey += __shfl_xor(ey, 32, kWave);
            ez += __shfl_xor(ez, 16, kWave); ez += __shfl_xor(ez, 32, kWave);
            if (valid && q == 0) {
                float* o = a.eik + gpt * 3;
                o[0] = ex * a.box_scale; o[1] = ey * a.box_scale; o[2] = ez * a.box_scale;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace e3dge
