// The split-f16 ("f16x3") fused renderer, second generation: 8 waves per workgroup (two per SIMD), 16 points per wave,
// v_mfma_f32_16x16x32_f16.  Included by siren.hip (shares its launch code); see siren.hip for the reference lines of every
// phase and DESIGN.md 4.1c for the numbers.
//
// Why: the first f16x3 kernel (one wave per SIMD, 32 points per wave, 32x32x16 MFMAs) keeps the matrix pipe 41 % busy -- with a
// single wave on a SIMD every weight-fragment ds_read, every epilogue VALU op and every thin phase between the layers comes
// straight out of the MFMA stream.  Halving the points per wave halves the register-resident state (64 + 64 registers of
// packed f16 hi/lo activations instead of 128 + 128), so two waves fit on a SIMD and one wave's LDS reads / epilogue run
// under the other's MFMAs.  Price: every wave still reads the whole weight chunk for its (now 16) points, so LDS read
// traffic per point doubles (to ~60 % of the 256 B/clk ds_read_b128 rate) -- affordable, because the LDS itself was never
// the bottleneck, the single wave's issue stream was.
//
// Fragment conventions (validated on the GPU by e3dge_selftest_mfma16x16): lane l, n = l & 15, q = l >> 4
//   A (16 x 32): row n, k = 8q + j          B (32 x 16): column n, k = 8q + j          C/D (16 x 16): column n, rows 4q + r
// Standard layers: A = weights (16 out features x 32 k), B = activations (32 k x 16 points): D[feature 16t + 4q + r][point n].
// A lane's C/D registers of the tiles 2g, 2g+1 (features 32g + 16e + 4q + r) are its 8 k-slots of k-step g of the next layer
// -- the weight image (kOffBig16b) is packed in that k order, so activations never leave the registers.
// View layer: the same registers as the A operand, weights as B: D[point 4q + r][feature n] (features on lanes).
#pragma once
#include "siren_common.h"

namespace e3dge {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int k16Threads = 512;
constexpr int k16Tiles = 16;                     // 16-feature output tiles per layer
constexpr int k16Steps = 8;                      // k-steps of 32 per tile
constexpr int k16ChunkFloats = 16 * kWidth;      // one tile x K = 256: 16 KiB
constexpr int k16Chunks = kBigLayers * k16Tiles; // 128 chunks per 128-point sub-tile
// One workgroup barrier per tile.  (Per two tiles with a fifth LDS buffer measured the same, 0.3140 vs 0.3145 ms: the wait at
// the barrier is where the two waves of a SIMD queue for the matrix pipe, not a cost of the barrier -- removed)
#ifndef E3DGE_16_ABL
#define E3DGE_16_ABL 0      // timing ablations (wrong results): 4 = no workgroup barrier in the weight pipe, 8 = no
                            // transmittance scan, 16 = no colour compositing scan, 32 = no ordered merge of the feature partials
#endif
constexpr int k16NBuf = 4;                       // LDS weight buffers
constexpr int k16Slots = 2;                      // rays a 16-point slab can touch when S >= 16
#ifndef E3DGE_16_RING
#define E3DGE_16_RING 2     // 2, 4 and 8 measure the same (0.320 / 0.320 / 0.323 ms): the fragment reads are not latency-exposed
#endif
constexpr int k16Ring = E3DGE_16_RING;           // k-steps of (hi, lo) fragments held in registers (must divide 8)

// ---- LDS carve (floats) ----
constexpr int k16LdsW = 0;
constexpr int k16LdsFilm = k16LdsW + k16NBuf * k16ChunkFloats;        // [9][2][256]
constexpr int k16LdsHead = k16LdsFilm + 9 * 2 * kWidth;               // w_sigma[256], w_rgb[3][256], b_sigma, b_rgb[3]
constexpr int k16LdsW0 = k16LdsHead + kHeadFloats;                    // [3][256] first-layer weights, column-major
constexpr int k16LdsWvt = k16LdsW0 + 3 * kWidth;                      // [3][256] view-direction columns of the view layer, x128
constexpr int k16LdsFeat = k16LdsWvt + 3 * kWidth;                    // [kRMax][kFPitch]
constexpr int k16LdsPart = k16LdsFeat + kRMax * kFPitch;              // [8 waves][2 slots][256]
constexpr int k16LdsAlpha = ((k16LdsPart + 8 * k16Slots * kWidth + 3) / 4) * 4;
constexpr int k16LdsWgt = k16LdsAlpha + kTilePts;
constexpr int k16LdsZ = k16LdsWgt + kTilePts;
constexpr int k16LdsPts = k16LdsZ + kTilePts;                         // [128][3]
constexpr int k16LdsRgb = k16LdsPts + kTilePts * 3;                   // [128][3]
constexpr int k16LdsState = k16LdsRgb + kTilePts * 3;                 // [kRMax][12]
constexpr int k16LdsWq = ((k16LdsState + kRMax * kStateStride + 3) / 4) * 4;   // [8 waves][2 slots][16 rows]
constexpr int k16LdsVd = k16LdsWq + 8 * k16Slots * 16;                // [8 waves][16 points][4]: view direction (padded)
constexpr int k16LdsFloats = k16LdsVd + 8 * 16 * 4;
constexpr int k16LdsBytes = k16LdsFloats * 4;
static_assert(k16LdsBytes <= 160 * 1024, "LDS budget");
static_assert((k16LdsFilm % 4) == 0 && (k16LdsHead % 4) == 0 && (k16LdsW0 % 4) == 0 && (k16LdsWvt % 4) == 0, "alignment");

// one 16-byte store of saved state (E3DGE_NT_STORES=1: non-temporal -- an A/B of round 6)
#ifndef E3DGE_NT_STORES
#define E3DGE_NT_STORES 1
#endif
__device__ __forceinline__ void save_st4(float* p, const f32x4v& v) {
#if E3DGE_NT_STORES
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4v*>(p));
#else
    *reinterpret_cast<f32x4v*>(p) = v;
#endif
}
__device__ __forceinline__ f32x4v mfma16x16(u32x4 a, u32x4 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v zero4() { return f32x4v{0.f, 0.f, 0.f, 0.f}; }
// acc + w * (f16 half `HI` of the packed word p), one instruction
template <int HI> __device__ __forceinline__ float fma_mix_h(unsigned p, float w, float acc) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));
    return r;
}
// value r (0..3) of 16-feature tile t from the packed (hi, lo) activations
__device__ __forceinline__ float act16_get(const u32x4 (&aH)[k16Steps], const u32x4 (&aL)[k16Steps], int t, int r) {
    const int g = t >> 1, w = 2 * (t & 1) + (r >> 1);
    return (r & 1) ? f16hi(aH[g][w]) + f16hi(aL[g][w]) : f16lo(aH[g][w]) + f16lo(aL[g][w]);
}
template <int CTRL> __device__ __forceinline__ float dpp16(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a row (every lane of the row gets the total)
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp16<0x128>(x);    // row_ror:8
    x += dpp16<0x124>(x);    // row_ror:4
    x += dpp16<0x122>(x);    // row_ror:2
    x += dpp16<0x121>(x);    // row_ror:1
    return x;
}

// x[l] + x[l ^ 16] + x[l ^ 32] + x[l ^ 48] in every lane (the sum over the four 16-lane groups q), in the VALU: v_permlane16_swap
// exchanges the odd rows of its first operand with the even rows of the second, v_permlane32_swap the upper half of the first
// with the lower half of the second -- fed the same register twice, the two results add up to the pairwise sums.  (The same
// tree as two __shfl_xor steps, without their ds_bpermute round trips through the LDS crossbar.)
__device__ __forceinline__ float sum_over_q(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned v = __builtin_bit_cast(unsigned, y);
    const auto t = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __builtin_bit_cast(float, (unsigned)t[0]) + __builtin_bit_cast(float, (unsigned)t[1]);
}

// Wave-wide inclusive scans in the VALU (DPP; gfx9 controls row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143,
// wave_shr:1 = 0x138): four shifted steps inside every 16-lane row, then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into
// rows 2 and 3.  Lanes a step does not reach combine with the identity.  Lane 63 ends up with the reduction over the wave.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_or(float ident, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_incl_prod(float x) {
    x = __fmul_rn(x, dpp_or<0x111, 0xf>(1.0f, x));
    x = __fmul_rn(x, dpp_or<0x112, 0xf>(1.0f, x));
    x = __fmul_rn(x, dpp_or<0x114, 0xf>(1.0f, x));
    x = __fmul_rn(x, dpp_or<0x118, 0xf>(1.0f, x));
    x = __fmul_rn(x, dpp_or<0x142, 0xa>(1.0f, x));
    x = __fmul_rn(x, dpp_or<0x143, 0xc>(1.0f, x));
    return x;
}
__device__ __forceinline__ float wave_sum_last(float x) {      // the sum over the wave, valid in lane 63 (broadcast below)
    x = __fadd_rn(x, dpp_or<0x111, 0xf>(0.0f, x));
    x = __fadd_rn(x, dpp_or<0x112, 0xf>(0.0f, x));
    x = __fadd_rn(x, dpp_or<0x114, 0xf>(0.0f, x));
    x = __fadd_rn(x, dpp_or<0x118, 0xf>(0.0f, x));
    x = __fadd_rn(x, dpp_or<0x142, 0xa>(0.0f, x));
    x = __fadd_rn(x, dpp_or<0x143, 0xc>(0.0f, x));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// 16-KiB weight chunks through k16NBuf LDS buffers; every wave moves a 2-KiB slice (two LDS-DMA pieces) of each chunk.
//   tile g: sync() -> wait for everything but the chunk issued one tile ago, barrier (publishes chunk g+2, proves tile g-1 is
//   finished) -> issue chunk g+3 into the buffer tile g-1 used: two tile times to arrive.
struct ChunkPipe16 {
    const char* img;
    uint32_t voff, lds_base;
    int idx, buf, use_buf, count;
    float* wbuf;
    const float* wcur;
    const float* wnxt;
    // `image` = first chunk of the cycle, `count_` chunks per 128-point sub-tile (the cycle then starts over)
    __device__ __forceinline__ void init(float* wbuf_, const float* image, int wave, int lane, int count_ = k16Chunks) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        wbuf = wbuf_;
        count = count_;
        img = reinterpret_cast<const char*>(image) + wave_u * 2048;
        voff = (uint32_t)lane * 16u;
        lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)wbuf_ + (uint32_t)wave_u * 2048u;
        idx = 0; buf = 0; use_buf = 0;
        wcur = wbuf_; wnxt = wbuf_ + k16ChunkFloats;
    }
    __device__ __forceinline__ void issue_chunk() {
        const char* s = img + (size_t)idx * (k16ChunkFloats * 4);
        const uint32_t d = lds_base + (uint32_t)buf * (k16ChunkFloats * 4);
        glds16_saddr<0>(s, voff, d);
        glds16_saddr<1024>(s, voff, d);
        idx = (idx + 1 == count) ? 0 : idx + 1;
        buf = (buf + 1 == k16NBuf) ? 0 : buf + 1;
    }
    __device__ __forceinline__ void prime() {
        issue_chunk(); issue_chunk(); issue_chunk();
    }
#ifdef E3DGE_PHASE_TIMING
    unsigned long long t_vm = 0, t_bar = 0;
#endif
    template <bool STRICT> __device__ __forceinline__ void sync() {
#ifdef E3DGE_PHASE_TIMING
        const unsigned long long c0 = __builtin_readcyclecounter();
#endif
        // STRICT (training: the epilogue's argument stores share vmcnt): wait for everything
        if (STRICT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#ifdef E3DGE_PHASE_TIMING
        const unsigned long long c1 = __builtin_readcyclecounter();
#endif
#if !(E3DGE_16_ABL & 4)
        __syncthreads();
#endif
#ifdef E3DGE_PHASE_TIMING
        t_vm += c1 - c0; t_bar += __builtin_readcyclecounter() - c1;
#endif
    }
    __device__ __forceinline__ void advance() {
        use_buf = (use_buf + 1 == k16NBuf) ? 0 : use_buf + 1;
        wcur = wnxt;
        wnxt = wbuf + ((use_buf + 1 == k16NBuf) ? 0 : use_buf + 1) * k16ChunkFloats;
    }
};

#ifdef E3DGE_16_TRACE
// profiling build: s_memtime at the start of every k-step and after its MFMAs, for one tile of two partner waves
// (tools/trace16.py).  Inline asm, so no s_waitcnt is placed behind the stamps; they are collected after the tile.
__device__ unsigned long long g_trace16[2][24];
#define TRACE16_STAMP(i) do { if (trace) asm volatile("s_memtime %0" : "=s"(tstamp[i])); } while (0)
#else
#define TRACE16_STAMP(i) do { } while (0)
#endif

// K = 256 contraction of one 16-feature tile: 8 k-steps x (hi*hi, lo*hi, hi*lo) on two alternating accumulators.
// On entry the ring holds k-steps 0..2 of this chunk; on exit k-steps 0..2 of the next one.
// `hook()` runs once, after k-step 1: the chunk wait + workgroup barrier + next DMA issue (and whatever else has to sit there).
template <bool TRANSPOSED, int RING = k16Ring, class Epi, class Hook>
__device__ __forceinline__ void tile16(ChunkPipe16& pipe, int lane, const u32x4 (&aH)[k16Steps], const u32x4 (&aL)[k16Steps],
                                       f32x4v& acc, f32x4v& accb, u32x4 (&ringH)[RING], u32x4 (&ringL)[RING], Epi&& epi,
                                       Hook&& hook, int sbuf = -1, [[maybe_unused]] int trace = 0) {
#ifdef E3DGE_16_TRACE
    unsigned long long tstamp[24];
#endif
    // sbuf >= 0: the caller knows the buffer index statically (tile index mod k16NBuf in a fully unrolled layer): the fragment
    // reads then are one base register + immediates instead of per-buffer address registers
    const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(sbuf >= 0 ? pipe.wbuf + sbuf * k16ChunkFloats : pipe.wcur) + lane;
    const u32x4* __restrict__ wn = reinterpret_cast<const u32x4*>(sbuf >= 0 ? pipe.wbuf + ((sbuf + 1) % k16NBuf) * k16ChunkFloats : pipe.wnxt) + lane;
#pragma unroll
    for (int g = 0; g < k16Steps; ++g) {
        const int ga = g + RING - 1;
        TRACE16_STAMP(3 * g);
        ringH[ga % RING] = (ga < k16Steps) ? wp[(ga * 2 + 0) * 64] : wn[((ga - k16Steps) * 2 + 0) * 64];
        ringL[ga % RING] = (ga < k16Steps) ? wp[(ga * 2 + 1) * 64] : wn[((ga - k16Steps) * 2 + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 wh = ringH[g % RING], wl = ringL[g % RING];
        f32x4v& x0 = (g & 1) ? accb : acc;
        f32x4v& x1 = (g & 1) ? acc : accb;
        if (!TRANSPOSED) {
            x0 = mfma16x16(wh, aH[g], x0);
            x1 = mfma16x16(wl, aH[g], x1);
            x0 = mfma16x16(wh, aL[g], x0);
        } else {
            x0 = mfma16x16(aH[g], wh, x0);
            x1 = mfma16x16(aH[g], wl, x1);
            x0 = mfma16x16(aL[g], wh, x0);
        }
        TRACE16_STAMP(3 * g + 1);
        if (g == 1) hook();
        epi(g);
        TRACE16_STAMP(3 * g + 2);
    }
#ifdef E3DGE_16_TRACE
    if (trace) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 24; ++i) g_trace16[trace - 1][i] = tstamp[i];
    }
#endif
}

#ifndef E3DGE_16_HELPERS_ONLY
#ifdef E3DGE_PHASE_TIMING
#define PHASE16(i) do { if (MODE == 0 && blockIdx.x == 0 && tid == 0 && sub < 3) tstamp[sub * 6 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PHASE16(i) do { } while (0)
#endif

// CACHE (render mode without SAVE only): 1 = also write the backbone output (layer 7, packed hi / lo) to a.bb_out, one 16-KiB
// record per wave slab in register order; 2 = read it (and the composite weights a.weights_in) back instead of running layers
// 0..7, the sdf head, alpha and the transmittance scan: the second pass of an evaluated image differs from the first only
// behind the sdf head (texture FiLM -> view layer -> colour / feature compositing).  Geometry, launch shape and the point ->
// (workgroup, sub-tile, wave, lane) mapping are those of the launch that wrote the record.
constexpr int64_t k16SlabWords = 8 * 2 * 64;      // u32x4 per slab record: [g][hi | lo][lane]

template <int MODE, bool SAVE, int CACHE = 0>
__global__ void __launch_bounds__(k16Threads) siren16_kernel(const SirenK a) {
    static_assert(CACHE == 0 || (MODE == 0 && !SAVE), "the backbone hand-over exists for inference renders only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + k16LdsW;
    float* const film_s = smem + k16LdsFilm;
    float* const head_s = smem + k16LdsHead;
    float* const w0_s = smem + k16LdsW0;
    float* const wvt_s = smem + k16LdsWvt;
    float* const feat_acc = smem + k16LdsFeat;
    float* const part = smem + k16LdsPart;
    float* const alpha_s = smem + k16LdsAlpha;
    float* const wgt_s = smem + k16LdsWgt;
    float* const z_s = smem + k16LdsZ;
    float* const pts_s = smem + k16LdsPts;
    float* const rgb_s = smem + k16LdsRgb;
    float* const state = smem + k16LdsState;

    const int tid_k = threadIdx.x;
#ifdef E3DGE_PHASE_TIMING
    const unsigned long long t_entry = __builtin_readcyclecounter();
    unsigned long long t_loop0 = 0, t_loop1 = 0;
#endif
    // ---- work assignment (as siren_kernel) ----
    int b, npts, n_sub;
    int pix0 = 0, nrays = 0;
    long long pt0 = 0;
    const int S = (MODE == 0) ? a.S : 1;
    if (MODE == 0) {
        b = blockIdx.x / a.tiles_per_img;
        const int tile = blockIdx.x - b * a.tiles_per_img;
        const int HW = a.H * a.Wd;
        pix0 = tile * a.R;
        nrays = min(a.R, HW - pix0);
        npts = nrays * S;
    } else {
        b = blockIdx.x / a.wgs_per_img;
        const int wg = blockIdx.x - b * a.wgs_per_img;
        pt0 = (long long)wg * a.subtiles_per_wg * kTilePts;
        const long long rem = a.n_pts - pt0;
        npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    }
    n_sub = (npts + kTilePts - 1) / kTilePts;

    const float* __restrict__ packed = a.packed;
    const float* __restrict__ film_g = a.film + (int64_t)b * 9 * 2 * kWidth;
    // first weight chunks on their way (L2 -> LDS by DMA) while the tables below are built
    ChunkPipe16 pipe;
    if (CACHE == 2) pipe.init(wbuf, packed + kOffBig16b + (int64_t)(kBigLayers - 1) * k16Tiles * k16ChunkFloats, tid_k >> 6, tid_k & 63, k16Tiles);
    else pipe.init(wbuf, packed + kOffBig16b, tid_k >> 6, tid_k & 63);
    pipe.prime();
    // FiLM block with the layer bias folded into the offset and the weights' factor 128 divided out of gamma (layers >= 1)
    for (int i = tid_k; i < 9 * kWidth; i += k16Threads) {
        const int l = i >> 8, n = i & 255;
        const float gm = film_g[(l * 2 + 0) * kWidth + n], bt = film_g[(l * 2 + 1) * kWidth + n];
        film_s[(l * 2 + 0) * kWidth + n] = (l >= 1) ? gm * (1.0f / kW16Scale) : gm;
        film_s[(l * 2 + 1) * kWidth + n] = __fadd_rn(__fmul_rn(gm, packed[kOffBias + l * kWidth + n]), bt);
    }
    for (int i = tid_k; i < kHeadFloats; i += k16Threads) head_s[i] = packed[kOffWSigma + i];
    for (int i = tid_k; i < 3 * kWidth; i += k16Threads) {
        const int c = i >> 8, n = i & 255;           // fragment images of layer 0 / the view tail: [t][m][lane], k = 2m + half
        const int src = ((n >> 5) * 2 + (c >> 1)) * 64 + (c & 1) * 32 + (n & 31);
        w0_s[i] = packed[kOffFirst + src];
        wvt_s[i] = packed[kOffVTail + src] * kW16Scale;
    }
    float cw[12] = {0}, focal = 1.f, nearv = 0.f, farv = 0.f;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) cw[i] = a.c2w[b * 12 + i];
        focal = a.focal[b]; nearv = a.near[b]; farv = a.far[b];
        for (int i = tid_k; i < kRMax * kStateStride; i += k16Threads) state[i] = 0.0f;
        for (int i = tid_k; i < kRMax * kFPitch; i += k16Threads) feat_acc[i] = 0.0f;
    }
    const float* __restrict__ film = film_s;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4 ringH[k16Ring], ringL[k16Ring];
    {
        const int lane0 = tid_k & 63;
#pragma unroll
        for (int g = 0; g < k16Ring - 1; ++g) {
            ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + lane0];
            ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + lane0];
        }
    }

#ifdef E3DGE_16_TRACE
    auto trace_sel = [&](int L, int t) {
        const int w = __builtin_amdgcn_readfirstlane(tid_k >> 6);
        return (MODE == 0 && blockIdx.x == 7 && L == 3 && t == 6 && (w == 0 || w == 4)) ? 1 + (w >> 2) : 0;
    };
#else
    auto trace_sel = [](int, int) { return 0; };
#endif
    // SAVE: the argument stores share the in-order memory queue with the weight DMA.  Until round 6 this hook drained the queue
    // (s_waitcnt vmcnt(0), and __syncthreads() adds the same for the compiler's own stores): every tile of the saving forward waited for
    // its previous tile's stores to reach L2 -- +40 % over the plain forward.  Now a counted wait (the chunk of tile t+1 was issued at
    // hook t-2; younger than it: `younger` guaranteed operations) and a barrier in asm, as in siren16_bwd.h.
    auto fwd_hook = [&](int younger) {          // younger: compile-time constant at every call site; ignored without SAVE
        if (SAVE) {
            switch (younger) {
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            pipe.template sync<false>();
        }
        pipe.issue_chunk();
    };
    // packed f16 (hi, lo) activations of this wave's 16 points: word 2e + (r >> 1), half r & 1 of in?[g] = feature 32g + 16e + 4q + r
    u32x4 inH[k16Steps], inL[k16Steps], outH[k16Steps], outL[k16Steps];
#ifdef E3DGE_PHASE_TIMING
    unsigned long long tstamp[18];
    for (int i = 0; i < 18; ++i) tstamp[i] = 0;
#endif

#ifdef E3DGE_PHASE_TIMING
    t_loop0 = __builtin_readcyclecounter();
#endif
    for (int sub = 0; sub < n_sub; ++sub) {
        int tid_o = tid_k;
        asm volatile("" : "+v"(tid_o));                    // opaque: address math stays inside the sub-tile (no hoisted registers)
        const int tid = tid_o, lane = tid & 63, wave = tid >> 6, q = lane >> 4, col = lane & 15;
        PHASE16(0);
        // =====================================================================================
        // 1. this lane's point (replicated over the four lane groups q)
        // =====================================================================================
        const int p_sub = 16 * wave + col;
        const int p = sub * kTilePts + p_sub;
        const bool valid = p < npts;
        const int pc = valid ? p : (npts - 1);
        float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f, zval = 0.f, dist = 0.f;
        int ray_l = 0, s_idx = 0;
        int64_t gpt;
        if (MODE == 0) {
            ray_l = pc / S;
            s_idx = pc - ray_l * S;
            const int pix = pix0 + ray_l;
            const int iy = pix / a.Wd, ix = pix - iy * a.Wd;
            gpt = ((int64_t)b * a.H * a.Wd + pix) * S + s_idx;
            const float hres = (float)a.res * 0.5f;
            const float d0 = __fdiv_rn(((float)ix + 0.5f) - hres, focal);
            const float d1 = -__fdiv_rn(((float)iy + 0.5f) - hres, focal);
            const float d2 = -1.0f;
            float rd[3];
#pragma unroll
            for (int m = 0; m < 3; ++m)
                rd[m] = __fadd_rn(__fadd_rn(__fmul_rn(d0, cw[4 * m + 0]), __fmul_rn(d1, cw[4 * m + 1])), __fmul_rn(d2, cw[4 * m + 2]));
            const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)));
            vx = __fdiv_rn(d0, dn); vy = __fdiv_rn(d1, dn); vz = __fdiv_rn(d2, dn);
            const float tv = a.t_vals[s_idx];
            zval = __fadd_rn(__fmul_rn(nearv, __fsub_rn(1.0f, tv)), __fmul_rn(farv, tv));
            px = __fadd_rn(cw[3], __fmul_rn(rd[0], zval));
            py = __fadd_rn(cw[7], __fmul_rn(rd[1], zval));
            pz = __fadd_rn(cw[11], __fmul_rn(rd[2], zval));
            const float rdn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rd[0], rd[0]), __fmul_rn(rd[1], rd[1])), __fmul_rn(rd[2], rd[2])));
            if (s_idx + 1 < S) {
                const float tn = a.t_vals[s_idx + 1];
                const float zn = __fadd_rn(__fmul_rn(nearv, __fsub_rn(1.0f, tn)), __fmul_rn(farv, tn));
                dist = __fmul_rn(__fsub_rn(zn, zval), rdn);
            } else {
                dist = __fmul_rn(1e10f, rdn);
            }
            if (valid && q == 0) {
                if (a.points) { float* o = a.points + gpt * 3; o[0] = px; o[1] = py; o[2] = pz; }
                if (a.dists) a.dists[gpt] = dist;
                if (s_idx == 0) {
                    const int64_t gr = (int64_t)b * a.H * a.Wd + pix;
                    if (a.rays_d) { float* o = a.rays_d + gr * 3; o[0] = rd[0]; o[1] = rd[1]; o[2] = rd[2]; }
                    if (a.viewdirs) { float* o = a.viewdirs + gr * 3; o[0] = vx; o[1] = vy; o[2] = vz; }
                }
            }
        } else {
            gpt = (int64_t)b * a.n_pts + pt0 + pc;
            const float* pp = a.pts + gpt * 3;
            px = pp[0]; py = pp[1]; pz = pp[2];
            if (a.vdirs) { const float* vv = a.vdirs + gpt * 3; vx = vv[0]; vy = vv[1]; vz = vv[2]; }
        }
        const int64_t gpt_block = (MODE == 0) ? ((int64_t)b * a.H * a.Wd + pix0) * S : (int64_t)b * a.n_pts + pt0;
        // Saved arguments: every lane stores -- rows beyond the tensor are exact clones of the last valid point (same position, same
        // arithmetic, the same values to the same address), so the store count per tile does not depend on the data (counted waits).
        const bool sblk = SAVE && a.save_blocked != 0;
        const int64_t img_n = (MODE == 0) ? (int64_t)a.H * a.Wd * S : (int64_t)a.n_pts;
        const int64_t srow_block = (int64_t)b * saved_rows_per_image(sblk, img_n) + ((MODE == 0) ? (int64_t)pix0 * S : pt0);    // padded row of the block's first point
        float* const sv = SAVE ? a.save_args + saved_row_floats(sblk, srow_block + pc, q, 9) : nullptr;
        const int sv_ls = sblk ? kSlabLayerF : kWidth, sv_ts = sblk ? kSlabTileF : 16;
        // the view layer (features on lanes) needs the directions of the points 4q + r: through LDS
        float* const vd_s = smem + k16LdsVd + wave * 64;
        if (q == 0) { vd_s[col * 4 + 0] = vx; vd_s[col * 4 + 1] = vy; vd_s[col * 4 + 2] = vz; }

        // =====================================================================================
        // 2. layer 0 (3 -> 256) in the VALU: every lane its 4 features of each tile
        // =====================================================================================
        if (CACHE != 2) {
            const float xs = __fmul_rn(px, a.box_scale), ys = __fmul_rn(py, a.box_scale), zs = __fmul_rn(pz, a.box_scale);
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                const int o = 16 * t + 4 * q;
                const f32x4v wx = *reinterpret_cast<const f32x4v*>(w0_s + o), wy = *reinterpret_cast<const f32x4v*>(w0_s + kWidth + o),
                             wz = *reinterpret_cast<const f32x4v*>(w0_s + 2 * kWidth + o);
                const f32x4v g4 = *reinterpret_cast<const f32x4v*>(film + o), b4 = *reinterpret_cast<const f32x4v*>(film + kWidth + o);
                f32x4v arg, v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lin = fmaf(wz[r], zs, fmaf(wy[r], ys, wx[r] * xs));
                    arg[r] = fmaf(g4[r], lin, b4[r]);
                    v[r] = sin_f32(arg[r]);
                }
                if (SAVE) save_st4(sv + t * sv_ts, arg);
                SPLIT2_TO(v[0], v[1], inH[t >> 1][2 * (t & 1)], inL[t >> 1][2 * (t & 1)]);
                SPLIT2_TO(v[2], v[3], inH[t >> 1][2 * (t & 1) + 1], inL[t >> 1][2 * (t & 1) + 1]);
            }
        }

        PHASE16(1);
        // =====================================================================================
        // 3. layers 1..7: 112 tiles; the FiLM + sine epilogue of tile t-1 issues inside tile t's MFMA stream
        // =====================================================================================
        // (A role split of the two waves of a SIMD -- GEMM segment / epilogue segment in opposite phases, two barriers per tile -- was
        // built and measured 5 % slower; the code is in the history at da20f64.)
#pragma unroll 1
        for (int L = 1; L < ((CACHE == 2) ? 0 : E3DGE_SIREN_DEPTH); ++L) {
            const float* __restrict__ film_l = film + L * 2 * kWidth;
            f32x4v prev = zero4();
            auto finish = [&](int tp, const f32x4v& pv) {      // whole epilogue of tile tp (used for the layer's last tile)
                const int o = 16 * tp + 4 * q;
                const f32x4v g4 = *reinterpret_cast<const f32x4v*>(film_l + o), b4 = *reinterpret_cast<const f32x4v*>(film_l + kWidth + o);
                f32x4v arg, v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { arg[r] = fmaf(g4[r], pv[r], b4[r]); v[r] = sin_f32(arg[r]); }
                if (SAVE) save_st4(sv + L * sv_ls + tp * sv_ts, arg);
                SPLIT2_TO(v[0], v[1], outH[tp >> 1][2 * (tp & 1)], outL[tp >> 1][2 * (tp & 1)]);
                SPLIT2_TO(v[2], v[3], outH[tp >> 1][2 * (tp & 1) + 1], outL[tp >> 1][2 * (tp & 1) + 1]);
            };
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                f32x4v acc = zero4(), accb = zero4();
                if (t == 0) {
                    tile16<false>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [](int) {}, [&]() { fwd_hook(2); }, t % k16NBuf);
                } else {
                    const int o = 16 * (t - 1) + 4 * q;
                    f32x4v g4 = zero4(), b4 = zero4(), arg4 = zero4(), kf4 = zero4(), x4 = zero4();
                    tile16<false>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [&](int g) {
                        // the FiLM + sine + split of tile t-1's four values, staged over the k-steps with the four values side by
                        // side: every slice is four independent instructions deep instead of one dependent chain (a lone chain
                        // leaves the wave -- in order -- waiting on VALU latency with nothing else to issue)
                        if (g == 0) {
                            g4 = *reinterpret_cast<const f32x4v*>(film_l + o);
                            b4 = *reinterpret_cast<const f32x4v*>(film_l + kWidth + o);
                        } else if (g == 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) arg4[r] = fmaf(g4[r], prev[r], b4[r]);
#ifndef E3DGE_POLY_SINE      // sin_hw_f32 (siren_common.h) in stages: period index, reduced argument in revolutions, v_sin
#pragma unroll
                            for (int r = 0; r < 4; ++r) kf4[r] = rintf(arg4[r] * 0.15915494f);
                        } else if (g == 2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) kf4[r] = fmaf(arg4[r], 6.4206382e-09f, fmaf(arg4[r], 0.15915494f, -kf4[r]));
#endif
                        } else if (g == 3) {
#pragma unroll
#ifndef E3DGE_POLY_SINE
                            for (int r = 0; r < 4; ++r) x4[r] = __builtin_amdgcn_sinf(kf4[r]);
#else
                            for (int r = 0; r < 4; ++r) x4[r] = sin_poly_f32(arg4[r]);
#endif
                            if (SAVE) save_st4(sv + L * sv_ls + (t - 1) * sv_ts, arg4);
                        } else if (g == 4) {
                            SPLIT2_TO(x4[0], x4[1], outH[(t - 1) >> 1][2 * ((t - 1) & 1)], outL[(t - 1) >> 1][2 * ((t - 1) & 1)]);
                            SPLIT2_TO(x4[2], x4[3], outH[(t - 1) >> 1][2 * ((t - 1) & 1) + 1], outL[(t - 1) >> 1][2 * ((t - 1) & 1) + 1]);
                        }
                    // (guaranteed operations younger than chunk t+1: the two pieces of hook t-1 and the stores of tiles t-3, t-2 -- none
                    // guaranteed in front of a layer's first tiles: the view layer's stores are predicated)
                    }, [&]() { fwd_hook(t < 2 ? 2 : (t == 2 ? 3 : 4)); }, t % k16NBuf, trace_sel(L, t));
                }
                pipe.advance();
                prev = acc + accb;
            }
            finish(k16Tiles - 1, prev);
#pragma unroll
            for (int g = 0; g < k16Steps; ++g) { inH[g] = outH[g]; inL[g] = outL[g]; }
        }

        if (CACHE != 0) {
            const int64_t rec = (((int64_t)blockIdx.x * a.bb_subs + sub) * 8 + wave) * k16SlabWords + lane;
            if (CACHE == 1) {
                u32x4* __restrict__ dst = reinterpret_cast<u32x4*>(a.bb_out) + rec;
#pragma unroll
                for (int g = 0; g < k16Steps; ++g) { dst[(g * 2 + 0) * 64] = inH[g]; dst[(g * 2 + 1) * 64] = inL[g]; }
            } else {
                const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(a.bb_in) + rec;
#pragma unroll
                for (int g = 0; g < k16Steps; ++g) { inH[g] = src[(g * 2 + 0) * 64]; inL[g] = src[(g * 2 + 1) * 64]; }
            }
        }

        PHASE16(2);
        // =====================================================================================
        // 4. sdf head on the backbone output (features on registers): 64 features per lane, then over the 4 lane groups
        // =====================================================================================
        float sdf = 0.0f;
        if (CACHE == 2) {
            if (q == 0) wgt_s[p_sub] = valid ? a.weights_in[gpt] : 0.0f;      // (read back by this wave's own lanes and, after the
        } else {                                                                //  barrier behind rgb_s, by the compositing threads)
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                const f32x4v w4 = *reinterpret_cast<const f32x4v*>(head_s + 16 * t + 4 * q);
                const int g = t >> 1, w0i = 2 * (t & 1);
                acc = fma_mix_h<0>(inH[g][w0i], w4[0], acc); acc = fma_mix_h<0>(inL[g][w0i], w4[0], acc);
                acc = fma_mix_h<1>(inH[g][w0i], w4[1], acc); acc = fma_mix_h<1>(inL[g][w0i], w4[1], acc);
                acc = fma_mix_h<0>(inH[g][w0i + 1], w4[2], acc); acc = fma_mix_h<0>(inL[g][w0i + 1], w4[2], acc);
                acc = fma_mix_h<1>(inH[g][w0i + 1], w4[3], acc); acc = fma_mix_h<1>(inL[g][w0i + 1], w4[3], acc);
            }
            sdf = sum_over_q(acc) + head_s[4 * kWidth];
        }

        if (MODE == 0 && CACHE != 2) {
            const float sg = __fdiv_rn(sigmoid_f32(__fdiv_rn(-sdf, a.sigmoid_beta)), a.sigmoid_beta);
            const float alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sg, dist)));
            if (q == 0) {
                alpha_s[p_sub] = valid ? alpha : 0.0f;
                z_s[p_sub] = zval;
                pts_s[p_sub * 3 + 0] = px; pts_s[p_sub * 3 + 1] = py; pts_s[p_sub * 3 + 2] = pz;
                if (valid && a.sdf) a.sdf[gpt] = sdf;
            }
            __syncthreads();
            // transmittance scan (:869-886) as a wavefront prefix product: a wave per ray touching this sub-tile, a lane per sample
            // (64 at a time, front to back): T_s = T_in * prod_{j < s} (1 - alpha_j + 1e-10) from one inclusive DPP scan shifted by a
            // lane, the weight sum in front of the last sample and the weighted depth / position sums as DPP reductions.
            // (Same terms as the reference's cumprod / sum; the association order is the scan tree's.)
            const int sub_lo = sub * kTilePts;
            const int sub_hi = min(sub_lo + kTilePts, npts);
            const int r_first = sub_lo / S, r_last = (sub_hi - 1) / S;
            for (int i = __builtin_amdgcn_readfirstlane(wave); i <= ((E3DGE_16_ABL & 8) ? -1 : r_last - r_first); i += 8) {
                const int rl = r_first + i;
                const int s_lo = max(0, sub_lo - rl * S), s_hi = min(S, sub_hi - rl * S);
                float* st = state + rl * kStateStride;
                float T_in = (s_lo == 0) ? 1.0f : st[0];
                float wsum = (s_lo == 0) ? 0.0f : st[1];
                float dep = 0.0f, x0 = 0.0f, x1 = 0.0f, x2 = 0.0f;
                for (int c0 = s_lo; c0 < s_hi; c0 += 64) {
                    const int s = c0 + lane;
                    const bool act = s < s_hi;
                    const int ps = rl * S + min(s, s_hi - 1) - sub_lo;
                    const float al = act ? alpha_s[ps] : 0.0f;
                    const float zz = z_s[ps], p0 = pts_s[ps * 3 + 0], p1 = pts_s[ps * 3 + 1], p2 = pts_s[ps * 3 + 2];
                    const float incl = wave_incl_prod(act ? __fadd_rn(__fsub_rn(1.0f, al), 1e-10f) : 1.0f);
                    const float T = __fmul_rn(T_in, dpp_or<0x138, 0xf>(1.0f, incl));
                    float w = __fmul_rn(al, T);
                    const bool last = a.force_bg && act && s == S - 1;
                    const float wfront = wave_sum_last(last ? 0.0f : w);
                    if (last) w = __fsub_rn(1.0f, __fadd_rn(wsum, wfront));
                    wsum = __fadd_rn(wsum, wfront);
                    if (act) wgt_s[ps] = w;
                    dep = __fadd_rn(dep, wave_sum_last(__fmul_rn(w, zz)));
                    x0 = __fadd_rn(x0, wave_sum_last(__fmul_rn(w, p0)));
                    x1 = __fadd_rn(x1, wave_sum_last(__fmul_rn(w, p1)));
                    x2 = __fadd_rn(x2, wave_sum_last(__fmul_rn(w, p2)));
                    T_in = __fmul_rn(T_in, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63)));
                }
                if (lane == 0) {
                    st[0] = T_in; st[1] = wsum;
                    st[2] = __fadd_rn(st[2], dep); st[3] = __fadd_rn(st[3], x0); st[4] = __fadd_rn(st[4], x1); st[5] = __fadd_rn(st[5], x2);
                }
            }
            __syncthreads();
            if (valid && q == 0 && a.weights) a.weights[gpt] = wgt_s[p_sub];
        }

        // optional per-point texture FiLM (:217-220), after the sdf head read h
        if (MODE == 0 && a.tex_alpha) {
            const float* __restrict__ ta = a.tex_alpha + gpt * kWidth;
            const float* __restrict__ tb = a.tex_beta + gpt * kWidth;
#pragma unroll
            for (int t = 0; t < k16Tiles; ++t) {
                const f32x4v a4 = *reinterpret_cast<const f32x4v*>(ta + 16 * t + 4 * q);
                const f32x4v b4 = *reinterpret_cast<const f32x4v*>(tb + 16 * t + 4 * q);
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = __fadd_rn(__fmul_rn(__fadd_rn(a4[r], 1.0f), act16_get(inH, inL, t, r)), b4[r]);
                SPLIT2_TO(y[0], y[1], inH[t >> 1][2 * (t & 1)], inL[t >> 1][2 * (t & 1)]);
                SPLIT2_TO(y[2], y[3], inH[t >> 1][2 * (t & 1) + 1], inL[t >> 1][2 * (t & 1) + 1]);
            }
        }

        PHASE16(3);
        // =====================================================================================
        // 5. view layer (259 -> 256), TRANSPOSED: D[point 4q + r][feature 16t + col]
        // =====================================================================================
        float* const wq_s = smem + k16LdsWq + wave * (k16Slots * 16);
        int slab_first_ray = 0, slab_nslots = 0;
        const int slab_p0 = sub * kTilePts + 16 * wave;
        if (MODE == 0) {
            const int slab_hi = min(slab_p0 + 16, npts);
            if (slab_hi > slab_p0) {
                slab_first_ray = slab_p0 / S;
                slab_nslots = (slab_hi - 1) / S - slab_first_ray + 1;
            }
            const int b1 = (slab_first_ray + 1) * S - slab_p0;
            if (lane < 16) {                               // row = lane: composite weight of that point, by the ray it belongs to
                const float w = (slab_p0 + lane < npts) ? wgt_s[16 * wave + lane] : 0.0f;
                wq_s[lane] = (lane < b1) ? w : 0.0f;
                wq_s[16 + lane] = (lane >= b1) ? w : 0.0f;
            }
        }
        // (vd_s and wq_s were written by this wave's own lanes: no workgroup barrier needed, only the LDS ordering of a wave)
        const f32x4v q0 = (MODE == 0) ? *reinterpret_cast<const f32x4v*>(wq_s + 4 * q) : zero4();
        const f32x4v q1 = (MODE == 0) ? *reinterpret_cast<const f32x4v*>(wq_s + 16 + 4 * q) : zero4();
        float dvx[4], dvy[4], dvz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { dvx[r] = vd_s[(4 * q + r) * 4 + 0]; dvy[r] = vd_s[(4 * q + r) * 4 + 1]; dvz[r] = vd_s[(4 * q + r) * 4 + 2]; }
        float prgb[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) prgb[c][r] = 0.0f;
        {
            const float* __restrict__ film_v = film + 8 * 2 * kWidth;
            const float* __restrict__ wrgb = head_s + kWidth;
            f32x4v pv = zero4();
            float e_gm = 0.f, e_bt = 0.f, e_w0 = 0.f, e_w1 = 0.f, e_w2 = 0.f, fa0 = 0.f, fa1 = 0.f;
            int e_n = 0;
            auto epi_begin = [&](int tp) {
                e_n = 16 * tp + col;
                e_gm = film_v[e_n]; e_bt = film_v[kWidth + e_n];
                e_w0 = wrgb[e_n]; e_w1 = wrgb[kWidth + e_n]; e_w2 = wrgb[2 * kWidth + e_n];
                fa0 = fa1 = 0.f;
            };
            auto epi_r = [&](int r) {
                const float varg = fmaf(e_gm, pv[r], e_bt);
                const float h = sin_f32(varg);
                const int pr = slab_p0 + 4 * q + r;
                if (SAVE && pr < npts) a.save_args[saved_elem_floats(sblk, srow_block + pr, 8, e_n, 9)] = varg;
                prgb[0][r] = fmaf(e_w0, h, prgb[0][r]);
                prgb[1][r] = fmaf(e_w1, h, prgb[1][r]);
                prgb[2][r] = fmaf(e_w2, h, prgb[2][r]);
                if (MODE == 0) {
                    fa0 = fmaf(q0[r], h, fa0);
                    fa1 = fmaf(q1[r], h, fa1);
                } else if (a.raw) {
                    if (pr < npts) a.raw[((int64_t)b * a.n_pts + pt0 + pr) * 260 + 4 + e_n] = h;
                }
            };
            auto epi_end = [&]() {
                if (MODE == 0) {
                    fa0 = sum_over_q(fa0);
                    fa1 = sum_over_q(fa1);
                    if (q == 0) {
                        float* pp = part + (wave * k16Slots) * kWidth + e_n;
                        if (slab_nslots > 0) pp[0] = fa0;
                        if (slab_nslots > 1) pp[kWidth] = fa1;
                    }
                }
            };
#pragma unroll 1
            for (int t = 0; t < k16Tiles; ++t) {
                f32x4v acc = zero4(), accb = zero4();
                if (t == 0) {
                    tile16<true>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [](int) {}, [&]() { fwd_hook(2); });
                } else {
                    epi_begin(t - 1);
                    tile16<true>(pipe, lane, inH, inL, acc, accb, ringH, ringL, [&](int g) { if (g >= 1 && g <= 4) epi_r(g - 1); }, [&]() { fwd_hook(2); });
                    epi_end();
                }
                pipe.advance();
                acc = acc + accb;
                // view-direction columns (fp32, already x128 like the streamed weights)
                const int n = 16 * t + col;
                const float t0 = wvt_s[n], t1 = wvt_s[kWidth + n], t2 = wvt_s[2 * kWidth + n];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaf(t2, dvz[r], fmaf(t1, dvy[r], fmaf(t0, dvx[r], acc[r])));
                pv = acc;
            }
            epi_begin(k16Tiles - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) epi_r(r);
            epi_end();
        }

        PHASE16(4);
        // =====================================================================================
        // 6. rgb head: sum the per-lane partials over the 16 feature lanes of each lane group (DPP row rotations)
        // =====================================================================================
        float rgbv[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) rgbv[c][r] = row_sum16(prgb[c][r]) + head_s[4 * kWidth + 1 + c];
        if (MODE == 0) {
            if (col == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ps = 16 * wave + 4 * q + r;
                    rgb_s[ps * 3 + 0] = sigmoid_f32(rgbv[0][r]);
                    rgb_s[ps * 3 + 1] = sigmoid_f32(rgbv[1][r]);
                    rgb_s[ps * 3 + 2] = sigmoid_f32(rgbv[2][r]);
                }
            }
            __syncthreads();
            const int sub_lo = sub * kTilePts;
            const int sub_hi = min(sub_lo + kTilePts, npts);
            const int r_first = sub_lo / S, r_last = (sub_hi - 1) / S;
            for (int i = __builtin_amdgcn_readfirstlane(wave); i <= ((E3DGE_16_ABL & 16) ? -1 : r_last - r_first); i += 8) {   // a wave per ray, a lane per sample
                const int rl = r_first + i;
                const int s_lo = max(0, sub_lo - rl * S), s_hi = min(S, sub_hi - rl * S);
                float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
                for (int b0 = s_lo; b0 < s_hi; b0 += 64) {
                    const int s = b0 + lane;
                    const int ps = rl * S + min(s, s_hi - 1) - sub_lo;
                    const float ww = (s < s_hi) ? wgt_s[ps] : 0.0f;
                    c0 = __fadd_rn(c0, wave_sum_last(__fmul_rn(ww, rgb_s[ps * 3 + 0])));
                    c1 = __fadd_rn(c1, wave_sum_last(__fmul_rn(ww, rgb_s[ps * 3 + 1])));
                    c2 = __fadd_rn(c2, wave_sum_last(__fmul_rn(ww, rgb_s[ps * 3 + 2])));
                }
                if (lane == 0) {
                    float* st = state + rl * kStateStride;
                    st[6] = __fadd_rn(st[6], c0); st[7] = __fadd_rn(st[7], c1); st[8] = __fadd_rn(st[8], c2);
                }
            }
            if (tid < kWidth && !(E3DGE_16_ABL & 32)) {   // ordered merge of the feature partials: slab by slab, ray slot by ray slot
                const int n = tid;
                float pv[8 * k16Slots];                      // (all sixteen loads in flight at once; unused slots are never added)
#pragma unroll
                for (int j = 0; j < 8 * k16Slots; ++j) pv[j] = part[j * kWidth + n];
                int cur = -1;
                float run = 0.0f;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) {
                    const int slab_lo = sub_lo + 16 * wv;
                    const int slab_hi2 = min(slab_lo + 16, npts);
                    if (slab_hi2 > slab_lo) {
                        const int fr = slab_lo / S;
                        const int ns = (slab_hi2 - 1) / S - fr + 1;
#pragma unroll
                        for (int sl = 0; sl < k16Slots; ++sl)
                            if (sl < ns) {
                                if (fr + sl != cur) {
                                    if (cur >= 0) feat_acc[cur * kFPitch + n] += run;
                                    cur = fr + sl;
                                    run = 0.0f;
                                }
                                run += pv[wv * k16Slots + sl];
                            }
                    }
                }
                if (cur >= 0) feat_acc[cur * kFPitch + n] += run;
            }
            __syncthreads();
        } else {
            if (a.raw && col == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pr = sub * kTilePts + 16 * wave + 4 * q + r;
                    if (pr < npts) {
                        float* o = a.raw + ((int64_t)b * a.n_pts + pt0 + pr) * 260;
                        o[0] = rgbv[0][r]; o[1] = rgbv[1][r]; o[2] = rgbv[2][r];
                    }
                }
            }
            if (valid && q == 0) {
                if (a.sdf) a.sdf[gpt] = sdf;
                if (a.raw) a.raw[gpt * 260 + 3] = sdf;
            }
        }
        PHASE16(5);
    }  // sub-tiles

#ifdef E3DGE_PHASE_TIMING
    t_loop1 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA in flight when the workgroup retires
    if (MODE == 0) {
        // 7. per-ray outputs, channel-first like VolumeFeatureRenderer.forward returns them (:1957-1968)
        const int tid = tid_k;
        const int HW = a.H * a.Wd;
        if (a.features) {
            float* fo = a.features + (int64_t)b * kWidth * HW + pix0;
            for (int e = tid; e < nrays * kWidth; e += k16Threads) {
                const int n = e / nrays, rl = e - n * nrays;
                fo[(int64_t)n * HW + rl] = feat_acc[rl * kFPitch + n];
            }
        }
        for (int rl = tid; rl < nrays; rl += k16Threads) {
            const float* st = state + rl * kStateStride;
            const int pix = pix0 + rl;
            if (a.rgb) {
                float* o = a.rgb + (int64_t)b * 3 * HW + pix;
                o[0] = __fadd_rn(-1.0f, __fmul_rn(2.0f, st[6]));
                o[HW] = __fadd_rn(-1.0f, __fmul_rn(2.0f, st[7]));
                o[2 * HW] = __fadd_rn(-1.0f, __fmul_rn(2.0f, st[8]));
            }
            if (a.xyz) {
                float* o = a.xyz + (int64_t)b * 3 * HW + pix;
                o[0] = st[3]; o[HW] = st[4]; o[2 * HW] = st[5];
            }
            if (a.depth) a.depth[(int64_t)b * HW + pix] = st[2];
            if (a.mask) a.mask[(int64_t)b * HW + pix] = (st[2] < a.mask_thresh) ? 1.0f : 0.0f;
        }
#ifdef E3DGE_PHASE_TIMING
        // profiling build: the first floats of `dists` carry thread 0's per-phase cycle counts (tools/phase_timing.py)
        __syncthreads();
        float* const td = a.dists ? a.dists : a.rgb;       // (the launch on the layer-7 record has no `dists`: its counts overwrite rgb[0..22])
        if (blockIdx.x == 0 && tid == 0 && td) {
            for (int i = 0; i < 18; ++i)
                td[i] = (i % 6 == 0) ? (float)(i / 6 ? tstamp[i] - tstamp[i - 1] : 0) : (float)(tstamp[i] - tstamp[i - 1]);
            td[18] = (float)pipe.t_vm; td[19] = (float)pipe.t_bar;
            td[20] = (float)(t_loop0 - t_entry);                            // prologue: LDS tables, first weight chunks
            td[21] = (float)(t_loop1 - t_loop0);                            // all sub-tiles
            td[22] = (float)(__builtin_readcyclecounter() - t_loop1);       // per-ray outputs (up to this thread's last store)
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Layout self-test of v_mfma_f32_16x16x32_f16 with the conventions above: c (16x16, row-major) = hi(a) hi(b)^T
__global__ void __launch_bounds__(64)
selftest_mfma16x16_kernel(float* __restrict__ cmat, const float* __restrict__ amat, const float* __restrict__ bmat, int k) {
    const int lane = threadIdx.x, q = lane >> 4, n = lane & 15;
    f32x4v acc = zero4();
    for (int kb = 0; kb < k; kb += 32) {
        u32x4 a4, b4, dummy;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            SPLIT2_TO(amat[n * k + kb + 8 * q + 2 * w], amat[n * k + kb + 8 * q + 2 * w + 1], a4[w], dummy[w]);
            SPLIT2_TO(bmat[n * k + kb + 8 * q + 2 * w], bmat[n * k + kb + 8 * q + 2 * w + 1], b4[w], dummy[w]);
        }
        acc = mfma16x16(a4, b4, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cmat[(4 * q + r) * 16 + n] = acc[r];
}

#endif  // E3DGE_16_HELPERS_ONLY

}  // namespace e3dge
