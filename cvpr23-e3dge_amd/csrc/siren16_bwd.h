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
#include <utility>

namespace e3dge {

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression in the
// body BY CONSTRUCTION.  (`#pragma unroll` over the sixteen tiles of a layer silently gave up on the two-stream kernels once the body held
// four DMA statements behind a sixteen-way switch on the tile index: the pre-unroll size estimate passed the pragma threshold, the tile
// index stayed a run-time value and out[] moved to scratch.)
template <class F, int... Is> __device__ __forceinline__ void t3_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void t3_static_for(F&& f) {
    t3_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ring slots of 1 KiB per wave and stream; a stream tile is fetched (slots - 1) tiles ahead of the tile being consumed.  One stream:
// eight slots (64 KiB of ring); two streams: four each (LDS: 64 KiB of weight buffers + 64 KiB of ring + tables).
#ifndef E3DGE_T3_SLOTS1
#define E3DGE_T3_SLOTS1 8
#endif
constexpr int t3_slots(int ns) { return ns == 1 ? E3DGE_T3_SLOTS1 : 4; }
constexpr int t3_ring_floats(int ns) { return 8 * t3_slots(ns) * 256; }      // one stream, eight waves
static_assert(k16Tiles % t3_slots(1) == 0 && k16Tiles % t3_slots(2) == 0, "static slot index = tile index mod slots");
// Who issues what (E3DGE_T3_SPLIT, default OFF -- an experiment that is kept because its result decides where the time is NOT).
// vmcnt retires in order, so a wave that waits for its (L2-resident, two tiles old) weight pieces also waits for every older stream
// load of its own: with one queue per wave the streams' latency budget is three tiles whatever the ring depth.  E3DGE_T3_SPLIT=1
// separates the queues BY WAVE -- waves 0-3 issue the whole weight chunk (4 pieces each), waves 4-7 the stream tiles of waves w-4
// and w (any wave may DMA into any LDS address), every wave waits for its own operations and the tile's barrier publishes all of
// them; a stream's budget is then its ring depth (7 tiles with one stream).  Measured (MI355X, 64x64x18, tools/r6_chain_abl.sh):
// sdf chain 0.439 vs 0.444 ms, tangent 0.437 vs 0.439, second-order backward 0.492 vs 0.480 -- nothing: the streams were not
// latency-bound, they were ACCESS-PATTERN-bound (sixteen 64-byte pieces per instruction; the slab-major layout above takes the same
// kernels to 0.325 / 0.321 / 0.468 ms, and 0.237 / 0.235 / 0.431 with cache-resident rows).
#ifndef E3DGE_T3_SPLIT
#define E3DGE_T3_SPLIT 0
#endif
constexpr bool kT3Split = E3DGE_T3_SPLIT != 0;

// backward kernel: weights | ring (1 or 2 streams) | gamma [9][256] | w_sigma [256] | wave slices [8][256][2] | W0 [3][256] (d_pts)
constexpr int kB16LdsW = 0;
constexpr int kB16LdsRing = kB16LdsW + k16NBuf * k16ChunkFloats;
constexpr int b16_lds_gam(int ns) { return kB16LdsRing + ns * t3_ring_floats(ns); }
constexpr int b16_lds_head(int ns) { return b16_lds_gam(ns) + 9 * kWidth; }
constexpr int b16_lds_wave(int ns) { return b16_lds_head(ns) + kWidth; }
constexpr int b16_lds_w0(int ns) { return b16_lds_wave(ns) + 8 * 2 * kWidth; }
constexpr int b16_lds_bytes(int ns, bool dpts) { return (b16_lds_w0(ns) + (dpts ? 3 * kWidth : 0)) * 4; }
static_assert(b16_lds_bytes(2, true) <= 160 * 1024 && b16_lds_bytes(1, true) <= 160 * 1024, "LDS budget (backward)");
static_assert(k16Chunks % k16NBuf == 0 && (7 * k16Tiles) % k16NBuf == 0 && k16Tiles % k16NBuf == 0, "static buffer index = tile index mod k16NBuf");

// chain kernels: weights | ring | gamma [8][256] | W0 [3][256] | w_sigma [256]
constexpr int kC16LdsW = 0;
constexpr int kC16LdsRing = kC16LdsW + k16NBuf * k16ChunkFloats;
constexpr int c16_lds_gam(int ns) { return kC16LdsRing + ns * t3_ring_floats(ns); }
constexpr int c16_lds_w0(int ns) { return c16_lds_gam(ns) + 8 * kWidth; }
constexpr int c16_lds_head(int ns) { return c16_lds_w0(ns) + 3 * kWidth; }
constexpr int c16_lds_bytes(int ns) { return (c16_lds_head(ns) + kWidth) * 4; }
static_assert(c16_lds_bytes(2) <= 160 * 1024 && c16_lds_bytes(1) <= 160 * 1024, "LDS budget (chain)");

constexpr int kB16Ring = 2;                       // k-steps of weight fragments held (registers are the scarce resource here)
// Timing ablations (tools/build_variant.sh -DE3DGE_T3_ABL=bits; results are wrong with any bit set):
//   1 = no stream DMA (the epilogues read whatever the ring holds)      2 = no stores of the chain kernels
//   16 = hot rows: every lane streams (and stores) the workgroup's first row -- the same instruction mix against cache-resident data
#ifndef E3DGE_T3_ABL
#define E3DGE_T3_ABL 0
#endif
// Layout of the saved state (pre-sine arguments (.., 9, 256), r_l / ta_l r_l (.., 8, 256)):
//   point-major (kT3Blocked = false): row p = the L x 256 floats of point p.  A wave's tile access touches 16 rows: 64 B in each.
//   slab-major  (kT3Blocked = true) : 16 consecutive points form a slab [L layers][16 tiles][lane (q, n) = 16 q + n][4 floats]: the
//       16 points x 16 features of one (layer, tile) are 1 KiB contiguous in exactly the order the 64 lanes hold them -- every stream
//       DMA and every store of a wave is ONE contiguous KiB (8 full lines, one DRAM page) instead of sixteen 64-byte pieces.
//       Same number of bytes (rows padded to a multiple of 16 per image); slab s starts where row 16 s starts.
#ifndef E3DGE_T3_BLOCKED
#define E3DGE_T3_BLOCKED 1      // (0 = point-major, as the first-generation kernels: A/B builds; the forward then must not be asked for slabs)
#endif
constexpr bool kT3Blocked = E3DGE_T3_BLOCKED != 0;
constexpr int kT3LayerF = kT3Blocked ? 16 * kWidth : kWidth;      // floats between consecutive layers of a point / slab
constexpr int kT3TileF = kT3Blocked ? kWidth : 16;                // floats between consecutive 16-feature tiles
// float offset of lane (n = pl & 15 of the slab, q)'s 4 values of (layer 0, tile 0), relative to the workgroup's first row; L layers per row
__device__ __forceinline__ uint32_t t3_row_floats(int pl, int q, int L) {
    return kT3Blocked ? (uint32_t)((pl >> 4) * L * (16 * kWidth) + (q * 16 + (pl & 15)) * 4) : (uint32_t)(pl * L * kWidth + q * 4);
}
constexpr int kT3StreamOps = (E3DGE_T3_ABL & 1) ? 0 : 1;      // counted operations per stream and tile / per store and tile
constexpr int kT3StoreOps = (E3DGE_T3_ABL & 2) ? 0 : 1;

__device__ __forceinline__ f32x4v ld4(const float* p) { return *reinterpret_cast<const f32x4v*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4v& v) { *reinterpret_cast<f32x4v*>(p) = v; }
__device__ __forceinline__ void st4_chain(float* p, const f32x4v& v) {      // the per-tile store of the chain kernels (see E3DGE_T3_ABL)
#if !(E3DGE_T3_ABL & 2)
    save_st4(p, v);
#else
    if (v[0] == 1.2345e-30f) *reinterpret_cast<f32x4v*>(p) = v;
#endif
}

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
// Issue roles as scalar flags, computed by SALU instructions only: (wave < 4, wave >= 4).  `wave_s` must come straight from
// __builtin_amdgcn_readfirstlane (then it IS an SGPR; a C++ select on it may be evaluated in the VALU, and an "s" asm operand silently
// accepts the resulting VGPR).  An earlier form read the flag back with v_readfirstlane inside an asm statement: the compiler cannot
// see the VALU-writes-SGPR / SALU-reads hazards of gfx950 in there, the flags came out inverted or stale on the MI355X and waves 0-3
// issued the stream tiles of waves -4..-1 (tools/ubench/asm_if.hip reproduces it in isolation).
__device__ __forceinline__ void t3_roles(int wave_s, int& w_role, int& s_role) {
    int w, st;
    asm volatile("s_cmp_lt_u32 %2, 4\n\ts_cselect_b32 %0, 1, 0\n\ts_cselect_b32 %1, 0, 1" : "=s"(w), "=s"(st) : "s"(wave_s) : "scc");
    w_role = w; s_role = st;
}
// Role-conditional forms: the scalar test and the branch live INSIDE the asm statement, so the compiler keeps seeing one straight-line
// tile (a C++ `if (role)` around the DMA split every unrolled tile into basic blocks and cost the chain kernels 60 registers).
#ifndef E3DGE_NT_LOADS
#define E3DGE_NT_LOADS 0
#endif
#if E3DGE_NT_LOADS
#define E3DGE_T3_NT " nt"
#else
#define E3DGE_T3_NT ""
#endif
template <int OFF_BYTES>
__device__ __forceinline__ void glds16_saddr_if(int flag, const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lt3skip%=\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4" E3DGE_T3_NT "\n.Lt3skip%=:"
                 :: "s"(flag), "v"(voff), "s"(sbase), "s"(lds_addr), "n"(OFF_BYTES) : "memory", "scc");
}
__device__ __forceinline__ void glds16_saddr_x4_if(int flag, const void* sbase, uint32_t voff, uint32_t lds_addr) {     // four 1-KiB pieces
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lt3skip%=\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:0" E3DGE_T3_NT "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" E3DGE_T3_NT "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048" E3DGE_T3_NT "\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" E3DGE_T3_NT "\n.Lt3skip%=:"
                 :: "s"(flag), "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "scc");
}
// s_waitcnt vmcnt(A) if flag else vmcnt(B)
template <int A, int B> __device__ __forceinline__ void t3_wait_by_role(int flag) {
#ifdef E3DGE_T3_STRICT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    static_assert(A >= 0 && A < 64 && B >= 0 && B < 64, "vmcnt is six bits");
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lt3w%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lt3e%=\n.Lt3w%=:\n\ts_waitcnt vmcnt(%2)\n.Lt3e%=:"
                 :: "s"(flag), "n"(A), "n"(B) : "memory", "scc");
#endif
}
template <int TILE, int SLOTS>
__device__ __forceinline__ void t3_issue(int flag, const void* gbase, uint32_t voff, uint32_t ring_lds) {
    // (the immediate is 12 bits: slab-major tiles are 1 KiB apart, so only tile & 3 fits and the rest goes into the lane offset)
    constexpr int kImm = kT3Blocked ? (TILE & 3) * 1024 : TILE * 64;
    constexpr uint32_t kAdd = kT3Blocked ? (uint32_t)(TILE >> 2) * 4096u : 0u;
    if (kT3Split) glds16_saddr_if<kImm>(flag, gbase, voff + kAdd, ring_lds + (uint32_t)((TILE & (SLOTS - 1)) * 1024 - kImm));
    else glds16_saddr<kImm>(gbase, voff + kAdd, ring_lds + (uint32_t)((TILE & (SLOTS - 1)) * 1024 - kImm));
}
// flag (scalar): this wave issues stream tiles (always 1 without the role split)
template <int TILE, int SLOTS>
__device__ __forceinline__ void t3_issue_tile(int flag, const void* gbase, uint32_t voff, uint32_t ring_lds) {
    if (E3DGE_T3_ABL & 1) return;
    if (E3DGE_T3_ABL & 16) voff &= (kT3Blocked ? 1023u : 63u);
    t3_issue<(TILE & 15), SLOTS>(flag, gbase, voff, ring_lds);
}
// The weight pipe of the 8-wave kernels with the issue roles above: `active` waves (0-3 when split: a quarter of the chunk each in four
// pieces; every wave an eighth in two pieces otherwise) issue, every wave keeps the bookkeeping.
struct T3WeightPipe : ChunkPipe16 {
    int active;
    __device__ __forceinline__ void init3(float* wbuf_, const float* image, int wave_u, int lane, int count_ = k16Chunks) {     // wave_u: from readfirstlane
        int s_unused;
        active = 1;
        if (kT3Split) t3_roles(wave_u, active, s_unused);
        init(wbuf_, image, kT3Split ? 2 * (wave_u & 3) : wave_u, lane, count_);     // split: img / lds_base at wave_u * 4 KiB
    }
    __device__ __forceinline__ void issue3() {
        const char* s = img + (size_t)idx * (k16ChunkFloats * 4);
        const uint32_t d = lds_base + (uint32_t)buf * (k16ChunkFloats * 4);
        if (kT3Split) {
            glds16_saddr_x4_if(active, s, voff, d);
        } else {
            glds16_saddr<0>(s, voff, d);
            glds16_saddr<1024>(s, voff, d);
        }
        idx = (idx + 1 == count) ? 0 : idx + 1;
        buf = (buf + 1 == k16NBuf) ? 0 : buf + 1;
    }
    __device__ __forceinline__ void prime3() { issue3(); issue3(); issue3(); }
};
constexpr int kT3WOps = kT3Split ? 4 : 2;         // weight pieces per issuing wave and tile
// Counted waits of the hook of tile t (see the header comment).  NS streams, NO stores per tile and wave, D = stream distance.
//   one queue per wave: the weight chunk of tile t+1 (hook t-2) is awaited; younger = streams of hook t-2, stores, all of hook t-1
//   split, weight waves: younger = stores of two gaps + the four pieces of hook t-1
//   split, stream waves: the streams of hook t-D are awaited; younger = D-1 hooks of 2 NS stream tiles + the stores of D gaps
// (tile 0 of a layer has no interleaved epilogue: the gap between hooks 0 and 1 holds no store)
template <int NS, int NO, int D> __device__ __forceinline__ void t3_hook_wait(int t, int w_role) {     // w_role: scalar 0 / 1
    constexpr int S = NS * kT3StreamOps, O = NO * kT3StoreOps;
    if (!kT3Split) {
        if (t == 2) t3_wait<2 + 2 * S + O>(); else t3_wait<2 + 2 * S + 2 * O>();
    } else {
        constexpr int kS = (D - 1) * 2 * S;
        if (t == 2 && t == D) t3_wait_by_role<4 + O, kS + (D - 1) * O>(w_role);
        else if (t == 2) t3_wait_by_role<4 + O, kS + D * O>(w_role);
        else if (t == D) t3_wait_by_role<4 + 2 * O, kS + (D - 1) * O>(w_role);
        else t3_wait_by_role<4 + 2 * O, kS + D * O>(w_role);
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
    constexpr int NS = EIK ? 2 : 1;                // streams: arguments [, ta r]; no regular stores (d_pts / d_tex / partial slices are extras)
    constexpr int kSlots = t3_slots(NS), kDist = kSlots - 1, kRingF = t3_ring_floats(NS);
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
    int w_role = 0, s_role = 1;                                               // weight waves issue no streams (scalar flags)
    if (kT3Split) t3_roles(wave_u, w_role, s_role);
    const int64_t base_pt = (int64_t)b * a.n_pts + pt0;                       // the workgroup's first point in the caller's (point-major) tensors
    const int64_t base_row = (int64_t)b * saved_rows_per_image(kT3Blocked, a.n_pts) + pt0;      // ... and its row in the saved state
    const char* const g_args = reinterpret_cast<const char*>(a.args + base_row * (9 * kWidth));
    const char* const g_tr = EIK ? reinterpret_cast<const char*>(a.tang + base_row * (8 * kWidth)) : nullptr;
    // ring bases (LDS byte address of slot 0) of the waves this wave fetches for: itself (B) and, when split, wave - 4 (A)
    const uint32_t ring_b = lds_addr_of(smem + kB16LdsRing) + (uint32_t)wave_u * (kSlots * 1024u);
    constexpr uint32_t kStream1 = (uint32_t)kRingF * 4u;                      // byte distance of the second stream's ring
    // row (relative to the workgroup's first) of this lane's column in wave `w` of sub-tile `sub`; rows beyond the tensor read the last valid row
    auto row_of = [&](int sub, int w, int tid_x) {
        const int p = sub * kTilePts + 16 * w + (tid_x & 15);
        return p < npts ? p : npts - 1;
    };
    // the stream tiles of (sub-tile, layer, tile) for the waves this wave serves
    auto issue_streams = [&](auto tile_c, int sub, int layer) {
        constexpr int tile = decltype(tile_c)::value;
        int tid_i = tid_k;
        asm volatile("" : "+v"(tid_i));             // recomputed at every use: nothing of this lives across a tile
        uint32_t ring_o = ring_b;
        asm volatile("" : "+s"(ring_o));            // (the per-tile slot addresses are loop invariants the compiler would hoist into ~60 SGPRs)
        const int q0 = (tid_i >> 4) & 3;
        const int rb = row_of(sub, tid_i >> 6, tid_i);
        t3_issue_tile<tile, kSlots>(s_role, g_args, 4u * (t3_row_floats(rb, q0, 9) + (uint32_t)layer * kT3LayerF), ring_o);
        if (EIK) t3_issue_tile<tile, kSlots>(s_role, g_tr, 4u * (t3_row_floats(rb, q0, 8) + (uint32_t)layer * kT3LayerF), ring_o + kStream1);
        if (kT3Split) {
            const int ra = row_of(sub, (tid_i >> 6) - 4, tid_i);
            t3_issue_tile<tile, kSlots>(s_role, g_args, 4u * (t3_row_floats(ra, q0, 9) + (uint32_t)layer * kT3LayerF), ring_o - 4u * (kSlots * 1024u));
            if (EIK) t3_issue_tile<tile, kSlots>(s_role, g_tr, 4u * (t3_row_floats(ra, q0, 8) + (uint32_t)layer * kT3LayerF), ring_o - 4u * (kSlots * 1024u) + kStream1);
        }
    };

    T3WeightPipe pipe;
    pipe.init3(wbuf, packed + kOffBigT16b, wave_u, tid_k & 63);
    pipe.prime3();
    t3_static_for<kDist>([&](auto tc) { issue_streams(tc, 0, 7); });          // the first tiles of the first sub-tile (layer 7)
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
        const float* __restrict__ ap = a.args + base_row * (9 * kWidth) + t3_row_floats(pc, q, 9);
        const float vmask = valid ? 1.0f : 0.0f;                       // padded lanes contribute nothing
        const float* __restrict__ txa = TEX ? a.tex_alpha + gpt * kWidth + 4 * q : nullptr;
        float* __restrict__ dta = TEX ? a.d_tex_alpha + gpt * kWidth + 4 * q : nullptr;
        float* __restrict__ dtb = TEX ? a.d_tex_beta + gpt * kWidth + 4 * q : nullptr;
        const float dsdf = (a.d_sdf && valid) ? a.d_sdf[gpt] : 0.0f;
        gmax = 0.0f;
        float drgb[3] = {0.f, 0.f, 0.f};
        if (a.d_rgb && valid) { drgb[0] = a.d_rgb[gpt * 3]; drgb[1] = a.d_rgb[gpt * 3 + 1]; drgb[2] = a.d_rgb[gpt * 3 + 2]; }
        const int sub_n = sub + 1 < n_sub ? sub + 1 : sub;             // (the last layer's hooks prefetch across the sub-tile boundary)
        const float* const ring_rd = smem + kB16LdsRing + wave * (kSlots * 256) + lane * 4;     // this lane's 16 bytes of slot 0, stream 0

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
                    arb[u] = ld4(ap + 8 * kT3LayerF + kT3TileF * (t4 + u));
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
        // (TEX: layer 8's pass is peeled off -- `first` is a compile-time constant -- so that the registers only it needs, the texture FiLM's
        // alpha tiles and the two gradient tiles it stores, are dead in the loop over the other seven layers: kept in one runtime loop they
        // cost the kernel 32 B of scratch per lane, and a scratch reload waits for every LDS-DMA piece in flight)
        auto layer = [&](const int Gb, auto first_c) __attribute__((always_inline)) {
            constexpr bool tex_here = TEX && decltype(first_c)::value;   // this GEMM's result is dL/dh8' (view-layer input)
            const int Lm1 = 7 - Gb;                                      // layer whose argument / FiLM the epilogue uses
            const float* __restrict__ fg = gam_s + Lm1 * kWidth + 4 * q;
            const float sdf_term = (Gb == 0) ? dsdf : 0.0f;              // the sdf head reads the backbone output h8
            // the last hooks of a layer fetch the first tiles of the next one (layer 7 of the next sub-tile at the end)
            const int sub_x = Gb < 7 ? sub : sub_n, lay_x = Gb < 7 ? Lm1 - 1 : 7;
            f32x4v prev = zero4();
            f32x4v a4 = zero4(), tr4 = zero4(), al2[2];                   // streams of the tile whose epilogue is running
            f32x4v e_g = zero4(), e_w = zero4(), e_da = zero4(), e_db = zero4();
            float rb[4], rg[4];
            auto epi_load = [&](int tp) {
                const int o = 16 * tp;
                e_g = ld4(fg + o); e_w = ld4(wsig_s + 4 * q + o);
                a4 = ld4(ring_rd + (tp & (kSlots - 1)) * 256);
                if (EIK) tr4 = ld4(ring_rd + kRingF + (tp & (kSlots - 1)) * 256);
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
            t3_static_for<k16Tiles>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                // after k-step 1: the counted wait, the barrier (publishes every wave's finished DMA), the next weight chunk, then the
                // streams kDist tiles ahead
                auto hook = [&]() {
                    t3_hook_wait<NS, 0, kDist>(t, w_role);
                    t3_barrier();
                    pipe.issue3();
                    constexpr int tn = (t + kDist) & 15;
                    if (t + kDist < k16Tiles) issue_streams(std::integral_constant<int, tn>{}, sub, Lm1);
                    else issue_streams(std::integral_constant<int, tn>{}, sub_x, lay_x);
                    if (tex_here) {
                        if (t > 0) asm volatile("" : "+v"(al2[(t - 1) & 1]));
                        al2[t & 1] = ld4(txa + 16 * t);
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
            });
            if (tex_here) asm volatile("" : "+v"(al2[(k16Tiles - 1) & 1]));
            epi_load(k16Tiles - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_val(k16Tiles - 1, r);
            epi_finish(k16Tiles - 1);
            if (Gb + 1 < kBigLayers) next_operand();
            t3_barrier();
            fold(Lm1);
        };
        if (TEX) layer(0, std::true_type{});
#pragma unroll 1
        for (int Gb = TEX ? 1 : 0; Gb < kBigLayers; ++Gb) layer(Gb, std::false_type{});
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
            // (the point's index is recomputed from the thread index: kept live across the eight layers it was spilled)
            int tid_d = tid_k;
            asm volatile("" : "+v"(tid_d));
            const int p_d = sub * kTilePts + 16 * (tid_d >> 6) + (tid_d & 15);
            if (p_d < npts && ((tid_d >> 4) & 3) == 0) {
                float* o = a.d_pts + (base_pt + p_d) * 3;
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
    constexpr int kSlots = t3_slots(NS), kDist = kSlots - 1, kRingF = t3_ring_floats(NS);
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
    int w_role = 0, s_role = 1;
    if (kT3Split) t3_roles(wave_u, w_role, s_role);
    const int64_t base_pt = (int64_t)b * a.n_pts + pt0;
    const int64_t base_row = (int64_t)b * saved_rows_per_image(kT3Blocked, a.n_pts) + pt0;
    const char* const g_args = reinterpret_cast<const char*>(a.args + base_row * (9 * kWidth));
    const char* const g_r = TR ? reinterpret_cast<const char*>(a.rmul + base_row * (8 * kWidth)) : nullptr;
    const uint32_t ring_b = lds_addr_of(smem + kC16LdsRing) + (uint32_t)wave_u * (kSlots * 1024u);
    constexpr uint32_t kStream1 = (uint32_t)kRingF * 4u;
    auto row_of = [&](int sub, int w, int tid_x) {
        const int p = sub * kTilePts + 16 * w + (tid_x & 15);
        return (E3DGE_T3_ABL & 16) ? 0 : (p < npts ? p : npts - 1);
    };
    auto issue_streams = [&](auto tile_c, int sub, int layer) {
        constexpr int tile = decltype(tile_c)::value;     // see siren16_bwd_kernel
        int tid_i = tid_k;
        asm volatile("" : "+v"(tid_i));
        uint32_t ring_o = ring_b;
        asm volatile("" : "+s"(ring_o));
        const int q0 = (tid_i >> 4) & 3;
        const int rb = row_of(sub, tid_i >> 6, tid_i);
        t3_issue_tile<tile, kSlots>(s_role, g_args, 4u * (t3_row_floats(rb, q0, 9) + (uint32_t)layer * kT3LayerF), ring_o);
        if (TR) t3_issue_tile<tile, kSlots>(s_role, g_r, 4u * (t3_row_floats(rb, q0, 8) + (uint32_t)layer * kT3LayerF), ring_o + kStream1);
        if (kT3Split) {
            const int ra = row_of(sub, (tid_i >> 6) - 4, tid_i);
            t3_issue_tile<tile, kSlots>(s_role, g_args, 4u * (t3_row_floats(ra, q0, 9) + (uint32_t)layer * kT3LayerF), ring_o - 4u * (kSlots * 1024u));
            if (TR) t3_issue_tile<tile, kSlots>(s_role, g_r, 4u * (t3_row_floats(ra, q0, 8) + (uint32_t)layer * kT3LayerF), ring_o - 4u * (kSlots * 1024u) + kStream1);
        }
    };

    T3WeightPipe pipe;
    // tangent: hidden layers 1..7 are the first 7 layers of the forward image; sdf chain: skip the view layer's transposed chunks
    pipe.init3(wbuf, packed + (TANGENT ? kOffBig16b : kOffBigT16b + (int64_t)k16Tiles * k16ChunkFloats), wave_u, tid_k & 63, kChainChunks);
    pipe.prime3();
    t3_static_for<kDist>([&](auto tc) { issue_streams(tc, 0, kFirstGemmLayer); });
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
        const int pc = (E3DGE_T3_ABL & 16) ? 0 : (p < npts ? p : (npts - 1));
        const int64_t gpt = base_pt + pc;
        const float* __restrict__ ap = a.args + base_row * (9 * kWidth) + t3_row_floats(pc, q, 9);
        const float* __restrict__ rp = TR ? a.rmul + base_row * (8 * kWidth) + t3_row_floats(pc, q, 8) : nullptr;
        float* __restrict__ sp = a.save + base_row * (8 * kWidth) + t3_row_floats(pc, q, 8);
        const int sub_n = sub + 1 < n_sub ? sub + 1 : sub;
        const float* const ring_rd = smem + kC16LdsRing + wave * (kSlots * 256) + lane * 4;
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
                    arb[u] = ld4(ap + l0 * kT3LayerF + kT3TileF * (t8 + u));
                    if (TR) rmb[u] = ld4(rp + l0 * kT3LayerF + kT3TileF * (t8 + u));
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
                    st4_chain(sp + l0 * kT3LayerF + kT3TileF * t, TR ? x4 * rmb[u] : x4);
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
            float* __restrict__ spl = sp + l * kT3LayerF;
            const int sub_x = step < 6 ? sub : sub_n, lay_x = step < 6 ? (TANGENT ? l + 1 : l - 1) : kFirstGemmLayer;
            f32x4v prev = zero4();
            f32x4v a4 = zero4(), r4 = zero4();
            f32x4v e_g = zero4(), e_st = zero4();
            auto epi_load = [&](int tp) {
                e_g = ld4(gl + 16 * tp);
                a4 = ld4(ring_rd + (tp & (kSlots - 1)) * 256);
                if (TR) r4 = ld4(ring_rd + kRingF + (tp & (kSlots - 1)) * 256);
            };
            auto epi_val = [&](int tp, int r) {
                const float ga = e_g[r] * prev[r];
                e_st[r] = TANGENT ? (TR ? ga * r4[r] : ga) : prev[r];
                out[tp][r] = cos_hw_f32(a4[r]) * ga;
                gmax = fmaxf(gmax, fabsf(out[tp][r]));
            };
            t3_static_for<k16Tiles>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                auto hook = [&]() {                                      // see siren16_bwd_kernel; one store per tile here
                    t3_hook_wait<NS, 1, kDist>(t, w_role);
                    t3_barrier();
                    pipe.issue3();
                    constexpr int tn = (t + kDist) & 15;
                    if (t + kDist < k16Tiles) issue_streams(std::integral_constant<int, tn>{}, sub, l);
                    else issue_streams(std::integral_constant<int, tn>{}, sub_x, lay_x);
                };
                f32x4v acc = zero4(), accb = zero4();
                if (t == 0) {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [](int) {}, hook, t % k16NBuf);
                } else {
                    tile16<false, kB16Ring>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [&](int g) {
                        if (g == 0) epi_load(t - 1);
                        else if (g <= 4) epi_val(t - 1, g - 1);
                        else if (g == 5) st4_chain(spl + kT3TileF * (t - 1), e_st);
                    }, hook, t % k16NBuf);
                }
                pipe.advance();
                prev = (acc + accb) * inv_scale;
            });
            epi_load(k16Tiles - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_val(k16Tiles - 1, r);
            st4_chain(spl + kT3TileF * (k16Tiles - 1), e_st);
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
