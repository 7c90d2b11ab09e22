// Backward-direction kernels of the FiLM-SIREN MLP on the forward kernel's machine (siren16.h): 8 waves per workgroup (two per
// SIMD), 16 points per wave, v_mfma_f32_16x16x32_f16 on block-scaled split-f16 operands.
//   siren16_bwd_kernel   : d(film), d(styles) [, d(points), d(texture FiLM)] of a loss on (feat, rgb, sdf) [and on the eikonal term]
//   siren16_chain_kernel : the two first-order chains the eikonal term needs (sdf chain / tangent)
// Included by siren_bwd.hip, which documents the mathematics and the reference lines (volume_renderer.py:168-264, :796-802) at
// the first-generation kernels (4 waves x 32 points); those stay as the fp32 path and as E3DGE_PREC_F16X3_V1.
//
// Round 6 ("third generation").  The round-2 version of this file had the same tiles and was not faster than the first generation:
// its waves were parked at s_waitcnt / s_barrier 46 % of their life (profiles/r2_pmc_issue_siren16_bwd_g2.txt) because
//   (a) every saved-state stream (pre-sine arguments, tangent arguments, r) was a register load issued ONE tile ahead -- in-order
//       vmcnt and compiler-visible destination registers allow no more -- against an HBM latency of several tile times, and
//   (b) __syncthreads() with compiler-visible global stores in flight is s_waitcnt vmcnt(0) + s_barrier: every tile of the two
//       chain kernels drained its own stores.
// Now:
//   * the streams arrive by LDS-DMA (global_load_lds_dwordx4), one instruction per wave, tile and stream: lane (n, q) fetches the
//     16 bytes [feature 16t + 4q .. + 3] of its point -- exactly the C/D fragment slot it will combine them with -- into a per-wave
//     ring of four 1-KiB slots, THREE tiles ahead of use.  No destination registers, so the distance is bounded by LDS only.
//   * one in-order queue, counted waits: per tile a wave issues [weight chunk x 2, streams x NS] in the hook and [stores x NO] from
//     the epilogue; the hook of tile T waits for the weight chunk of tile T+1 (issued at hook T-2) with
//     s_waitcnt vmcnt(2 + 2 NS + 2 NO) -- everything older has then retired too, in particular the streams of tile T (issued at
//     hook T-3), while the youngest two tiles' traffic stays in flight.  Nothing in the tile loop waits for vmcnt(0).
//     The count is a LOWER bound of the operations younger than the awaited one (any extra operation only makes a wait
//     stricter), so every counted operation is issued unconditionally: rows beyond the tensor are exact clones of the last valid
//     row (same addresses, same values, same stores) in the chain kernels, and zero contributions in the backward kernel.
//   * barriers are `s_waitcnt lgkmcnt(0); s_barrier` in inline asm: the compiler's barrier would add vmcnt(0) for its stores.
//   * the second-order backward reads TWO streams instead of three: ta_l and r_l only ever appear as the product ta_l r_l, which the
//     tangent kernel now forms (it reads r_l through the same ring) and stores in place of ta_l (e3dge_siren_tangent_tr).
//   * d gamma is accumulated as sum(da a) [+ ta r cos a] and finished as (S - beta d beta) / gamma by the fold kernel
//     (z = (a - beta) / gamma): two VALU operations and two LDS table reads per value less; per-layer sums leave the workgroup
//     per 128-point sub-tile (global partial slices, folded in fixed order: still no atomics, still bit-reproducible).
//
// Layout (siren16.h): lane l: point n = l & 15, q = l >> 4; register r of 16-feature tile t = feature 16t + 4q + r of point n.
#pragma once
#define E3DGE_16_HELPERS_ONLY
#include "siren16.h"

namespace e3dge {

constexpr int kT3Dist = 3;                        // stream tiles in flight ahead of the tile being consumed
constexpr int kT3Slots = 4;                       // ring slots of 1 KiB per wave and stream (= kT3Dist + the one being read)
constexpr int kT3RingFloats = 8 * kT3Slots * 256; // one stream, eight waves: 32 KiB
static_assert(k16Tiles % kT3Slots == 0, "static slot index = tile index mod kT3Slots");

// backward kernel: weights | ring (1 or 2 streams) | gamma [9][256] | w_sigma [256] | wave slices [8][256][2] | W0 [3][256] (d_pts)
constexpr int kB16LdsW = 0;
constexpr int kB16LdsRing = kB16LdsW + k16NBuf * k16ChunkFloats;
constexpr int b16_lds_gam(int ns) { return kB16LdsRing + ns * kT3RingFloats; }
constexpr int b16_lds_head(int ns) { return b16_lds_gam(ns) + 9 * kWidth; }
constexpr int b16_lds_wave(int ns) { return b16_lds_head(ns) + kWidth; }
constexpr int b16_lds_w0(int ns) { return b16_lds_wave(ns) + 8 * 2 * kWidth; }
constexpr int b16_lds_bytes(int ns, bool dpts) { return (b16_lds_w0(ns) + (dpts ? 3 * kWidth : 0)) * 4; }
static_assert(b16_lds_bytes(2, true) <= 160 * 1024, "LDS budget (backward)");
static_assert(k16Chunks % k16NBuf == 0 && (7 * k16Tiles) % k16NBuf == 0 && k16Tiles % k16NBuf == 0, "static buffer index = tile index mod k16NBuf");

// chain kernels: weights | ring | gamma [8][256] | W0 [3][256] | w_sigma [256]
constexpr int kC16LdsW = 0;
constexpr int kC16LdsRing = kC16LdsW + k16NBuf * k16ChunkFloats;
constexpr int c16_lds_gam(int ns) { return kC16LdsRing + ns * kT3RingFloats; }
constexpr int c16_lds_w0(int ns) { return c16_lds_gam(ns) + 8 * kWidth; }
constexpr int c16_lds_head(int ns) { return c16_lds_w0(ns) + 3 * kWidth; }
constexpr int c16_lds_bytes(int ns) { return (c16_lds_head(ns) + kWidth) * 4; }
static_assert(c16_lds_bytes(2) <= 160 * 1024, "LDS budget (chain)");

constexpr int kB16Ring = 2;                       // k-steps of weight fragments held (registers are the scarce resource here)

__device__ __forceinline__ f32x4v ld4(const float* p) { return *reinterpret_cast<const f32x4v*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4v& v) { *reinterpret_cast<f32x4v*>(p) = v; }

// ---- the in-order memory queue (see the header comment) ----
#ifdef E3DGE_T3_STRICT      // debugging: every counted wait drains the queue (a result that differs from the default build is a race)
template <int N> __device__ __forceinline__ void t3_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
template <int N> __device__ __forceinline__ void t3_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is six bits");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
#endif
// workgroup barrier that does not touch vmcnt (the compiler's own adds s_waitcnt vmcnt(0) when it has stores in flight)
__device__ __forceinline__ void t3_barrier() {
#if !(E3DGE_16_ABL & 4)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// One stream tile (16 points x 16 features of one layer = 1 KiB) of this wave into ring slot TILE & 3.  `gbase` (scalar) = the
// stream at the workgroup's first point, `voff` = the lane's byte offset (point row + layer + 16 q), the tile's 64 bytes go into
// the instruction's immediate -- which the hardware adds to the LDS address as well, so it is taken off the slot base.
template <int TILE>
__device__ __forceinline__ void t3_issue(const void* gbase, uint32_t voff, uint32_t ring_lds) {
    constexpr int kImm = TILE * 64;
    glds16_saddr<kImm>(gbase, voff, ring_lds + (uint32_t)((TILE & (kT3Slots - 1)) * 1024 - kImm));
}
__device__ __forceinline__ void t3_issue_tile(int tile, const void* gbase, uint32_t voff, uint32_t ring_lds) {   // tile: a constant after unrolling
    switch (tile & 15) {
        case 0: t3_issue<0>(gbase, voff, ring_lds); break;    case 1: t3_issue<1>(gbase, voff, ring_lds); break;
        case 2: t3_issue<2>(gbase, voff, ring_lds); break;    case 3: t3_issue<3>(gbase, voff, ring_lds); break;
        case 4: t3_issue<4>(gbase, voff, ring_lds); break;    case 5: t3_issue<5>(gbase, voff, ring_lds); break;
        case 6: t3_issue<6>(gbase, voff, ring_lds); break;    case 7: t3_issue<7>(gbase, voff, ring_lds); break;
        case 8: t3_issue<8>(gbase, voff, ring_lds); break;    case 9: t3_issue<9>(gbase, voff, ring_lds); break;
        case 10: t3_issue<10>(gbase, voff, ring_lds); break;  case 11: t3_issue<11>(gbase, voff, ring_lds); break;
        case 12: t3_issue<12>(gbase, voff, ring_lds); break;  case 13: t3_issue<13>(gbase, voff, ring_lds); break;
        case 14: t3_issue<14>(gbase, voff, ring_lds); break;  default: t3_issue<15>(gbase, voff, ring_lds); break;
    }
}
__device__ __forceinline__ uint32_t lds_addr_of(const float* p) {
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const float*)p;
}

// Split-f16 operand of a backward-type GEMM (see scale_split in siren_bwd.hip): the 256 values of a point are spread over the
// four lanes n, n+16, n+32, n+48; one power-of-two scale per point brings the largest into [1, 2).  `m` = max |value| over
// this lane's 64 values.  Returns 1 / (kW16Scale * scale) for the epilogue of the GEMM that consumes the operand.
__device__ __forceinline__ float scale_split16(const f32x4v (&src)[k16Tiles], u32x4 (&dH)[k16Steps], u32x4 (&dL)[k16Steps], float m) {
    {   // max over the four 16-lane groups in the VALU (the same exchanges as sum_over_q)
        const unsigned u = __builtin_bit_cast(unsigned, m);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        m = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
        const unsigned v = __builtin_bit_cast(unsigned, m);
        const auto t = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        m = fmaxf(__builtin_bit_cast(float, (unsigned)t[0]), __builtin_bit_cast(float, (unsigned)t[1]));
    }
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

// EIK / TEX / DPTS as in siren_bwd_kernel.  EIK: a.tang holds the PRODUCTS ta_l r_l (e3dge_siren_tangent_tr), a.rsave is unused.
// Partial sums: slice (workgroup, sub-tile) of a.partials, [9][2][256] = sum(da a [+ ta r cos a]), sum(da) per layer and feature.
template <bool EIK, bool TEX, bool DPTS>
__global__ void __launch_bounds__(k16Threads) siren16_bwd_kernel(const SirenBwdK a) {
    constexpr int NS = EIK ? 2 : 1;
    constexpr int kWaitN = 2 + 2 * NS;             // no regular stores in this kernel (d_pts / d_tex / partial slices are extras)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kB16LdsW;
    float* const gam_s = smem + b16_lds_gam(NS);
    float* const wsig_s = smem + b16_lds_head(NS);
    float* const wave_s = smem + b16_lds_wave(NS);
    float* const w0_s = smem + b16_lds_w0(NS);

    const int tid_k = threadIdx.x;
    const int b = blockIdx.x / a.wgs_per_img;
    const int wg = blockIdx.x - b * a.wgs_per_img;
    const long long pt0 = (long long)wg * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;

    const float* __restrict__ packed = a.packed;
    const float* __restrict__ film_g = a.film + (int64_t)b * 9 * 2 * kWidth;
    for (int i = tid_k; i < 9 * kWidth; i += k16Threads) gam_s[i] = film_g[((i >> 8) * 2) * kWidth + (i & 255)];
    for (int i = tid_k; i < kWidth; i += k16Threads) wsig_s[i] = packed[kOffWSigma + i];
    if (DPTS) for (int i = tid_k; i < 3 * kWidth; i += k16Threads) {
        const int c = i >> 8, n = i & 255;           // fragment image of layer 0 (siren_pack_kernel): [t][m][lane], k = 2m + half
        w0_s[i] = packed[kOffFirst + ((n >> 5) * 2 + (c >> 1)) * 64 + (c & 1) * 32 + (n & 31)];
    }

    const int wave_u = __builtin_amdgcn_readfirstlane(tid_k >> 6);
    const int64_t base_pt = (int64_t)b * a.n_pts + pt0;                       // the workgroup's first point
    const char* const g_args = reinterpret_cast<const char*>(a.args + base_pt * (9 * kWidth));
    const char* const g_tr = EIK ? reinterpret_cast<const char*>(a.tang + base_pt * (8 * kWidth)) : nullptr;
    const uint32_t ring_a = lds_addr_of(smem + kB16LdsRing) + (uint32_t)wave_u * 4096u;
    const uint32_t ring_t = ring_a + (uint32_t)kT3RingFloats * 4u;
    // per-lane row offsets of a sub-tile (rows beyond the tensor read the last valid row)
    auto row_of = [&](int sub) {
        const int p = sub * kTilePts + 16 * (tid_k >> 6) + (tid_k & 15);
        return p < npts ? p : npts - 1;
    };

    ChunkPipe16 pipe;
    pipe.init(wbuf, packed + kOffBigT16b, tid_k >> 6, tid_k & 63);
    pipe.prime();
    {   // streams of the first three tiles of the first sub-tile (layer 7)
        const uint32_t r0 = (uint32_t)row_of(0), qb = (uint32_t)((tid_k >> 4) & 3) * 16u;
        const uint32_t va = r0 * (9u * kWidth * 4u) + 7u * 1024u + qb, vt = r0 * (8u * kWidth * 4u) + 7u * 1024u + qb;
#pragma unroll
        for (int t = 0; t < kT3Dist; ++t) {
            t3_issue_tile(t, g_args, va, ring_a);
            if (EIK) t3_issue_tile(t, g_tr, vt, ring_t);
        }
    }
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
        const int64_t gpt = base_pt + pc;
        const float* __restrict__ ap = a.args + gpt * (9 * kWidth) + 4 * q;
        const float vmask = valid ? 1.0f : 0.0f;                       // padded lanes contribute nothing
        const float* __restrict__ txa = TEX ? a.tex_alpha + gpt * kWidth + 4 * q : nullptr;
        float* __restrict__ dta = TEX ? a.d_tex_alpha + gpt * kWidth + 4 * q : nullptr;
        float* __restrict__ dtb = TEX ? a.d_tex_beta + gpt * kWidth + 4 * q : nullptr;
        const float dsdf = (a.d_sdf && valid) ? a.d_sdf[gpt] : 0.0f;
        gmax = 0.0f;
        float drgb[3] = {0.f, 0.f, 0.f};
        if (a.d_rgb && valid) { drgb[0] = a.d_rgb[gpt * 3]; drgb[1] = a.d_rgb[gpt * 3 + 1]; drgb[2] = a.d_rgb[gpt * 3 + 2]; }
        // stream offsets of this sub-tile and of the next one (the last layer's hooks prefetch across the sub-tile boundary)
        const uint32_t vo_a = (uint32_t)pc * (9u * kWidth * 4u) + (uint32_t)q * 16u;
        const uint32_t vo_t = (uint32_t)pc * (8u * kWidth * 4u) + (uint32_t)q * 16u;
        const int pn = row_of(sub + 1 < n_sub ? sub + 1 : sub);
        const uint32_t vo_a_n = (uint32_t)pn * (9u * kWidth * 4u) + (uint32_t)q * 16u;
        const uint32_t vo_t_n = (uint32_t)pn * (8u * kWidth * 4u) + (uint32_t)q * 16u;
        const float* const ring_rd = smem + kB16LdsRing + wave * (kT3Slots * 256) + lane * 4;     // this lane's 16 bytes of slot 0, stream 0

        // sum(da), sum(da a ...) over this wave's 16 points for the 16 features of tile t: row sums, then the lanes
        // n < 4 of every row publish value n into this wave's slice
        auto reduce_store = [&](int t, const float (&rb)[4], const float (&rg)[4]) {
            // slice layout [feature][gamma, beta].  The address is recomputed from the thread index at every use (a few VALU ops
            // under the other wave's MFMAs): kept live across the tile it gets spilled.
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
        // after a workgroup barrier: thread (feature, quantity) adds the eight waves' sums of the finished layer in fixed order and
        // leaves them in this sub-tile's slice of the partial buffer
        float* const my_slice = a.partials + ((int64_t)blockIdx.x * a.subtiles_per_wg + sub) * (9 * 2 * kWidth);
        auto fold = [&](int layer) {
            const float* sp = wave_s + tid;                            // tid = 2 * feature + quantity
            float s = sp[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) s += sp[w * 2 * kWidth];
            my_slice[layer * 2 * kWidth + (tid & 1) * kWidth + (tid >> 1)] = s;
        };
        auto next_operand = [&]() {
            inv_scale = scale_split16(out, inH, inL, gmax);
            gmax = 0.0f;
        };

        // =====================================================================================
        // 1. view layer: dh_view = d_feat + Wrgb^T d_rgb ; g8 = gamma8 * dh_view * cos(arg8)
        // =====================================================================================
        if (sub > 0) t3_barrier();                                     // the previous sub-tile's last fold has read wave_s
        {
            const float* __restrict__ fg = gam_s + 8 * kWidth + 4 * q;
            const float* __restrict__ wr = packed + kOffWRgb + 4 * q;   // (3, 256), L2 / L1 resident
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
                    const f32x4v g4 = ld4(fg + o);
                    const f32x4v w0 = ld4(wr + o), w1 = ld4(wr + kWidth + o), w2 = ld4(wr + 2 * kWidth + o);
                    float rb[4], rg[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dh = vmask * (wfeat * dfb[u][r] + w0[r] * drgb[0] + w1[r] * drgb[1] + w2[r] * drgb[2]);
                        const float da = dh * cos_hw_f32(arb[u][r]);
                        rb[r] = da;
                        rg[r] = da * arb[u][r];
                        out[t][r] = g4[r] * da;
                        gmax = fmaxf(gmax, fabsf(out[t][r]));
                    }
                    reduce_store(t, rb, rg);
                }
            }
            next_operand();
            t3_barrier();
            fold(8);
        }

        // =====================================================================================
        // 2. the chain: GEMM Gb (layer L = 8 - Gb) turns g_L into dh_{L-1}; its epilogue makes g_{L-1}
        // =====================================================================================
#pragma unroll 1
        for (int Gb = 0; Gb < kBigLayers; ++Gb) {
            const int Lm1 = 7 - Gb;                                      // layer whose argument / FiLM the epilogue uses
            const float* __restrict__ fg = gam_s + Lm1 * kWidth + 4 * q;
            const float sdf_term = (Gb == 0) ? dsdf : 0.0f;              // the sdf head reads the backbone output h8
            const bool tex_here = TEX && Gb == 0;                        // this GEMM's result is dL/dh8' (view-layer input)
            // stream offsets: this layer's tiles, and the tiles the last three hooks fetch for the next layer (layer 7 of the next sub-tile at the end)
            const uint32_t va_cur = vo_a + (uint32_t)Lm1 * 1024u, vt_cur = vo_t + (uint32_t)Lm1 * 1024u;
            const uint32_t va_nxt = Gb < 7 ? va_cur - 1024u : vo_a_n + 7u * 1024u;
            const uint32_t vt_nxt = Gb < 7 ? vt_cur - 1024u : vo_t_n + 7u * 1024u;
            f32x4v prev = zero4();
            f32x4v a4 = zero4(), tr4 = zero4(), al2[2];                   // streams of the tile whose epilogue is running
            f32x4v e_g = zero4(), e_w = zero4(), e_da = zero4(), e_db = zero4();
            float rb[4], rg[4];
            auto epi_load = [&](int tp) {
                const int o = 16 * tp;
                e_g = ld4(fg + o); e_w = ld4(wsig_s + 4 * q + o);
                a4 = ld4(ring_rd + (tp & (kT3Slots - 1)) * 256);
                if (EIK) tr4 = ld4(ring_rd + kT3RingFloats + (tp & (kT3Slots - 1)) * 256);
            };
            auto epi_val = [&](int tp, int r) {                           // tp, r: compile-time constants at every call site
                const float ar = a4[r];
                float xin = prev[r];
                float sn = 0.f, cs;
                if (EIK || tex_here) sincos_hw16(ar, sn, cs);
                else cs = cos_hw_f32(ar);
                if (tex_here) { e_da[r] = xin * sn; e_db[r] = xin; xin = __fadd_rn(al2[tp & 1][r], 1.0f) * xin; }
                const float dh = fmaf(e_w[r], sdf_term, xin);            // padded lanes: operand 0 and dsdf = 0, so dh = 0
                float da;
                if (EIK) {
                    const float tr = vmask * tr4[r];                      // ta r
                    da = fmaf(dh, cs, -sn * tr);
                    rg[r] = fmaf(da, ar, tr * cs);                        // gamma d gamma + beta d beta, see the fold kernel
                } else {
                    da = dh * cs;
                    rg[r] = da * ar;
                }
                rb[r] = da;
                out[tp][r] = e_g[r] * da;
                gmax = fmaxf(gmax, fabsf(out[tp][r]));
            };
            auto epi_finish = [&](int tp) {
                if (tex_here && valid) { st4(dta + 16 * tp, e_da); st4(dtb + 16 * tp, e_db); }
                reduce_store(tp, rb, rg);
            };
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                // after k-step 1: the counted wait (weight chunk t+1 and, being older, the streams of tile t), the barrier, the next
                // weight chunk, then the streams three tiles ahead
                auto hook = [&]() {
                    t3_wait<kWaitN>();
                    t3_barrier();
                    pipe.issue_chunk();
                    t3_issue_tile(t + kT3Dist, g_args, t + kT3Dist < k16Tiles ? va_cur : va_nxt, ring_a);
                    if (EIK) t3_issue_tile(t + kT3Dist, g_tr, t + kT3Dist < k16Tiles ? vt_cur : vt_nxt, ring_t);
                    if (TEX) {
                        if (t > 0) asm volatile("" : "+v"(al2[(t - 1) & 1]));
                        al2[t & 1] = tex_here ? ld4(txa + 16 * t) : zero4();
                    }
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
            if (TEX) asm volatile("" : "+v"(al2[(k16Tiles - 1) & 1]));
            epi_load(k16Tiles - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_val(k16Tiles - 1, r);
            epi_finish(k16Tiles - 1);
            if (Gb + 1 < kBigLayers) next_operand();
            t3_barrier();
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
            ex = sum_over_q(ex); ey = sum_over_q(ey); ez = sum_over_q(ez);
            if (valid && q == 0) {
                float* o = a.d_pts + gpt * 3;
                o[0] = ex * a.box_scale; o[1] = ey * a.box_scale; o[2] = ez * a.box_scale;
            }
        }
    }
    // sub-tiles this workgroup does not have (the image's last workgroup): zero slices, the fold kernel adds every slice
    for (int sub = n_sub; sub < a.subtiles_per_wg; ++sub) {
        float* const sl = a.partials + ((int64_t)blockIdx.x * a.subtiles_per_wg + sub) * (9 * 2 * kWidth);
        for (int i = tid_k; i < 9 * 2 * kWidth; i += k16Threads) sl[i] = 0.0f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// TANGENT as in siren_chain_kernel.  Seven GEMMs: forward image, layers 1..7 (tangent) or transposed image, layers 7..1 (sdf).
// TR (tangent only): a.rmul = r_l of the sdf chain (batch, n_pts, 8, 256); what is stored is the product ta_l r_l.
template <bool TANGENT, bool TR>
__global__ void __launch_bounds__(k16Threads) siren16_chain_kernel(const SirenChainK a) {
    static_assert(TANGENT || !TR, "the product form belongs to the tangent pass");
    constexpr int NS = TR ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kC16LdsW;
    float* const gam_s = smem + c16_lds_gam(NS);
    float* const w0_s = smem + c16_lds_w0(NS);
    float* const ws_s = smem + c16_lds_head(NS);

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
    constexpr int kFirstGemmLayer = TANGENT ? 1 : 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid_k >> 6);
    const int64_t base_pt = (int64_t)b * a.n_pts + pt0;
    const char* const g_args = reinterpret_cast<const char*>(a.args + base_pt * (9 * kWidth));
    const char* const g_r = TR ? reinterpret_cast<const char*>(a.rmul + base_pt * (8 * kWidth)) : nullptr;
    const uint32_t ring_a = lds_addr_of(smem + kC16LdsRing) + (uint32_t)wave_u * 4096u;
    const uint32_t ring_r = ring_a + (uint32_t)kT3RingFloats * 4u;
    auto row_of = [&](int sub) {
        const int p = sub * kTilePts + 16 * (tid_k >> 6) + (tid_k & 15);
        return p < npts ? p : npts - 1;
    };

    ChunkPipe16 pipe;
    // tangent: hidden layers 1..7 are the first 7 layers of the forward image; sdf chain: skip the view layer's transposed chunks
    pipe.init(wbuf, packed + (TANGENT ? kOffBig16b : kOffBigT16b + (int64_t)k16Tiles * k16ChunkFloats), tid_k >> 6, tid_k & 63, kChainChunks);
    pipe.prime();
    {
        const uint32_t r0 = (uint32_t)row_of(0), qb = (uint32_t)((tid_k >> 4) & 3) * 16u;
        const uint32_t va = r0 * (9u * kWidth * 4u) + (uint32_t)kFirstGemmLayer * 1024u + qb;
        const uint32_t vr = r0 * (8u * kWidth * 4u) + (uint32_t)kFirstGemmLayer * 1024u + qb;
#pragma unroll
        for (int t = 0; t < kT3Dist; ++t) {
            t3_issue_tile(t, g_args, va, ring_a);
            if (TR) t3_issue_tile(t, g_r, vr, ring_r);
        }
    }
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
        // Rows beyond the tensor are exact clones of the last valid row: same loads, same arithmetic, the same values stored to the
        // same addresses -- every store of this kernel is unconditional (the counted waits rely on it).
        const int p = sub * kTilePts + 16 * wave + col;
        const int pc = p < npts ? p : (npts - 1);
        const int64_t gpt = base_pt + pc;
        const float* __restrict__ ap = a.args + gpt * (9 * kWidth) + 4 * q;
        const float* __restrict__ rp = TR ? a.rmul + gpt * (8 * kWidth) + 4 * q : nullptr;
        float* __restrict__ sp = a.save + gpt * (8 * kWidth) + 4 * q;
        const uint32_t vo_a = (uint32_t)pc * (9u * kWidth * 4u) + (uint32_t)q * 16u;
        const uint32_t vo_r = (uint32_t)pc * (8u * kWidth * 4u) + (uint32_t)q * 16u;
        const int pn = row_of(sub + 1 < n_sub ? sub + 1 : sub);
        const uint32_t vo_a_n = (uint32_t)pn * (9u * kWidth * 4u) + (uint32_t)q * 16u;
        const uint32_t vo_r_n = (uint32_t)pn * (8u * kWidth * 4u) + (uint32_t)q * 16u;
        const float* const ring_rd = smem + kC16LdsRing + wave * (kT3Slots * 256) + lane * 4;
        gmax = 0.0f;

        // ---- first layer of the chain (no GEMM): its arguments (and r) by ordinary loads, eight tiles in flight ----
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
            for (int t8 = 0; t8 < k16Tiles; t8 += 8) {
                f32x4v arb[8], rmb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    arb[u] = ld4(ap + l0 * kWidth + 16 * (t8 + u));
                    if (TR) rmb[u] = ld4(rp + l0 * kWidth + 16 * (t8 + u));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t8 + u, o = 16 * t;
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
                    st4(sp + l0 * kWidth + o, TR ? x4 * rmb[u] : x4);
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
            float* __restrict__ spl = sp + l * kWidth;
            const uint32_t va_cur = vo_a + (uint32_t)l * 1024u, vr_cur = vo_r + (uint32_t)l * 1024u;
            const uint32_t va_nxt = step < 6 ? (TANGENT ? va_cur + 1024u : va_cur - 1024u) : vo_a_n + (uint32_t)kFirstGemmLayer * 1024u;
            const uint32_t vr_nxt = step < 6 ? (TANGENT ? vr_cur + 1024u : vr_cur - 1024u) : vo_r_n + (uint32_t)kFirstGemmLayer * 1024u;
            f32x4v prev = zero4();
            f32x4v a4 = zero4(), r4 = zero4();
            f32x4v e_g = zero4(), e_st = zero4();
            auto epi_load = [&](int tp) {
                e_g = ld4(gl + 16 * tp);
                a4 = ld4(ring_rd + (tp & (kT3Slots - 1)) * 256);
                if (TR) r4 = ld4(ring_rd + kT3RingFloats + (tp & (kT3Slots - 1)) * 256);
            };
            auto epi_val = [&](int tp, int r) {
                const float ga = e_g[r] * prev[r];
                e_st[r] = TANGENT ? (TR ? ga * r4[r] : ga) : prev[r];
                out[tp][r] = cos_hw_f32(a4[r]) * ga;
                gmax = fmaxf(gmax, fabsf(out[tp][r]));
            };
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                auto hook = [&]() {                                      // see siren16_bwd_kernel; one store per tile here
                    if (t == 2) t3_wait<2 + 2 * NS + 1>(); else t3_wait<2 + 2 * NS + 2>();   // (tile 0 has no epilogue: one store less behind tile 2's chunk)
                    t3_barrier();
                    pipe.issue_chunk();
                    t3_issue_tile(t + kT3Dist, g_args, t + kT3Dist < k16Tiles ? va_cur : va_nxt, ring_a);
                    if (TR) t3_issue_tile(t + kT3Dist, g_r, t + kT3Dist < k16Tiles ? vr_cur : vr_nxt, ring_r);
                };
                f32x4v acc = zero4(), accb = zero4();
                if (t == 0) {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [](int) {}, hook, t % k16NBuf);
                } else {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [&](int g) {
                        if (g == 0) epi_load(t - 1);
                        else if (g <= 4) epi_val(t - 1, g - 1);
                        else if (g == 5) st4(spl + 16 * (t - 1), e_st);
                    }, hook, t % k16NBuf);
                }
                pipe.advance();
                prev = (acc + accb) * inv_scale;
            }
            epi_load(k16Tiles - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_val(k16Tiles - 1, r);
            st4(spl + 16 * (k16Tiles - 1), e_st);
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
            ex = sum_over_q(ex); ey = sum_over_q(ey); ez = sum_over_q(ez);
            if (q == 0) {
                float* o = a.eik + gpt * 3;
                o[0] = ex * a.box_scale; o[1] = ey * a.box_scale; o[2] = ez * a.box_scale;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace e3dge
