// Shared device-side definitions of the SIREN kernels (forward: siren.hip, backward: siren_bwd.hip): vector types,
// the packed weight image layout, LDS carve of the forward kernel, MFMA / LDS-DMA / sine helpers and the tile routines.
#pragma once
#include "common.h"

namespace e3dge {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));       // 8 packed f16 = one f16-MFMA operand
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int kWidth = E3DGE_SIREN_WIDTH;       // 256
constexpr int kNT = kWidth / 32;                // 8 output tiles of 32 features
constexpr int kChunkFloats = 32 * kWidth;       // one output tile x K=256 : 8192 floats = 32 KiB
constexpr int kBigLayers = 8;                   // pts_linears.1..7 + views_linears[:, :256]
constexpr int kChunksPerPass = kBigLayers * kNT;  // 64 chunks = 2 MiB per 128-point sub-tile
constexpr int kNBuf = 3;                        // LDS weight buffers
constexpr int kTilePts = 128;                   // points per sub-tile (4 waves x 32)

// 128-point sub-tiles a workgroup of a points launch walks one after the other.  One workgroup owns a CU (LDS), so a launch of `grid`
// workgroups of spw sub-tiles each lasts ceil(grid / 256) rounds x spw sub-tile times: take the spw <= 8 with the shortest launch and,
// among equals, the largest (fewest per-workgroup partial sums, and the CUs of an unfilled single round stay free for a side stream).
// (Until round 5 this was min(8, ceil(tiles / 256)): 4 x 576 tiles became 288 workgroups of 8 = two rounds of 8 for 9 rounds of work.)
inline int pick_subtiles_per_wg(long long tiles_per_img, int batch) {
    int best = 1;
    long long best_span = -1;
    for (int spw = 1; spw <= 8; ++spw) {
        const long long grid = ((tiles_per_img + spw - 1) / spw) * batch;
        const long long span = ((grid + 255) / 256) * spw;
        if (best_span < 0 || span <= best_span) { best_span = span; best = spw; }
    }
    return best;
}
constexpr int kThreads = 256;
constexpr int kRMax = 16;                       // max rays per workgroup (LDS feature accumulators)
constexpr int kFPitch = kWidth + 1;             // padded pitch of the feature accumulators
constexpr int kMaxSlots = 3;                    // rays a 32-point slab can touch when S >= 16
constexpr int kMinSamples = 16;

// VALU instructions of the pipelined epilogue forced behind each MFMA with sched_group_barrier (0 = leave the
// placement to the compiler).  Measured on MI355X (round 1, tools/build_variant.sh A/B): 0 -> 0.949 ms, 2..4 ->
// 1.00-1.02 ms, with one or two accumulators alike: on gfx950 the fp32 MFMA and the fp32 VALU do not execute
// concurrently for one wave (PMC: MFMA-busy + VALU-active + waits = wave cycles in every variant), so spreading
// only adds issue bubbles.  The lever that works is FEWER VALU instructions, not better placement.
#ifndef E3DGE_SPREAD_STD
#define E3DGE_SPREAD_STD 0
#endif
#ifndef E3DGE_SPREAD_VIEW
#define E3DGE_SPREAD_VIEW 0
#endif
// f16x3 path: VALU instructions scheduled behind each f16 MFMA (there the pipes DO overlap; 0 = compiler's placement)
#ifndef E3DGE_SPREAD16
#define E3DGE_SPREAD16 5
#endif

// ---- packed weight image (floats) ----
constexpr int64_t kOffBig = 0;                                       // [8 layers][8 t][8 c][4 q][64 lane][4]
constexpr int64_t kOffFirst = kOffBig + (int64_t)kChunksPerPass * kChunkFloats;   // [8 t][2][64]
constexpr int64_t kOffVTail = kOffFirst + kNT * 2 * 64;              // [8 t][2][64]
constexpr int64_t kOffBias = kOffVTail + kNT * 2 * 64;               // [9][256]
constexpr int64_t kOffWSigma = kOffBias + 9 * kWidth;                // [256]
constexpr int64_t kOffWRgb = kOffWSigma + kWidth;                    // [3][256]
constexpr int64_t kOffBHead = kOffWRgb + 3 * kWidth;                 // b_sigma, b_rgb[3]
constexpr int64_t kOffBig16 = kOffBHead + 4;                         // f16x3 image of the 8 big layers, see below
// fp32 image of the TRANSPOSED big layers for the backward chain dh_{L-1} = W_L^T g_L, in the order it is consumed:
// [Gb = 0..7 <-> layer L = 8 - Gb][8 out-tiles of k_in][8 c][4 q][64 lanes][4] = W_L[32c + 8q + 4(l>>5) + j][32t + (l&31)]
constexpr int64_t kOffBigT = kOffBig16 + (int64_t)kChunksPerPass * kChunkFloats;
// f16x3 (hi, lo) image of the same transposed GEMMs, same chunk order and word layout as kOffBig16:
// [Gb][8 out-tiles][16 k-steps][hi|lo][64 lanes][4 words of two f16], carrying the factor kW16Scale
constexpr int64_t kOffBigT16 = kOffBigT + (int64_t)kChunksPerPass * kChunkFloats;
// f16x3 image for the 8-wave forward kernel (siren16.h, v_mfma_f32_16x16x32_f16): the same 8 big layers as 128 chunks of
// 16 KiB = one 16-feature output tile x K = 256:  [Lb][16 t][8 k-steps g][hi|lo][64 lanes][4 words of two f16]; lane l
// (n = l & 15, q = l >> 4) holds, for j = 0..7,   128 * W[16t + n][32g + 16(j >> 2) + 4q + (j & 3)]
// -- the k order in which a lane's C/D registers of two consecutive 16-feature tiles of the previous layer (rows 4q + r)
// are the 8 k-slots of the next layer's operand.
constexpr int64_t kOffBig16b = kOffBigT16 + (int64_t)kChunksPerPass * kChunkFloats;
// the transposed GEMMs in the same 16x16x32 fragment order, for the 8-wave backward / sdf-chain kernels (siren16_bwd.h):
// [Gb = 0..7 <-> layer L = 8 - Gb][16 t][8 g][hi|lo][64 lanes][4 words]; lane l holds 128 * W_L[32g + 16(j >> 2) + 4q + (j & 3)][16t + n]
constexpr int64_t kOffBigT16b = kOffBig16b + (int64_t)kChunksPerPass * kChunkFloats;
constexpr int64_t kPackedFloats = kOffBigT16b + (int64_t)kChunksPerPass * kChunkFloats;
// f16x3 image: the same 64 chunks of 32 KiB, each [16 k-steps g = 2c+s][hi, lo][64 lanes][8 f16]: lane l holds
//   128 * W[32t + (l&31)][32c + 16s + (j&3) + 8(j>>2) + 4(l>>5)],  j = 0..7
// split as hi = f16(v), lo = f16(v - hi).  The k order is the one in which a lane's C/D registers of the previous
// layer (r = 8s + j) become the 8 k-slots of a v_mfma_f32_32x32x16_f16 operand; 128 keeps `lo` out of the f16
// subnormals and is undone exactly by storing gamma / 128.
constexpr float kW16Scale = 128.0f;

// ---- LDS carve (floats) ----
constexpr int kLdsW = 0;
constexpr int kLdsFilm = kLdsW + kNBuf * kChunkFloats;               // [9][2][256] gamma/beta of this image
constexpr int kLdsHead = kLdsFilm + 9 * 2 * kWidth;                  // w_sigma[256], w_rgb[3][256], b_sigma, b_rgb[3]
constexpr int kHeadFloats = 4 * kWidth + 4;
constexpr int kLdsVTail = kLdsHead + kHeadFloats;                    // [8 t][2][64] view-layer tail fragments
constexpr int kLdsFeat = kLdsVTail + kNT * 2 * 64;                   // [kRMax][kFPitch]
constexpr int kLdsPart = kLdsFeat + kRMax * kFPitch;                 // [4][kMaxSlots][256]
constexpr int kLdsAlpha = ((kLdsPart + 4 * kMaxSlots * kWidth + 3) / 4) * 4;   // [128]
constexpr int kLdsWgt = kLdsAlpha + kTilePts;                        // [128]
constexpr int kLdsZ = kLdsWgt + kTilePts;                            // [128]
constexpr int kLdsPts = kLdsZ + kTilePts;                            // [128][3]
constexpr int kLdsRgb = kLdsPts + kTilePts * 3;                      // [128][3]
constexpr int kLdsState = kLdsRgb + kTilePts * 3;                    // [kRMax][12]: T, wsum, depth, xyz3, rgb3
constexpr int kStateStride = 12;
constexpr int kLdsWq = kLdsState + kRMax * kStateStride;             // [4 waves][3 slots][2 halves][16]: composite weights per row
constexpr int kLdsFloats = kLdsWq + 4 * kMaxSlots * 2 * 16;
constexpr int kLdsBytes = kLdsFloats * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
static_assert((kLdsFilm % 4) == 0 && (kLdsHead % 4) == 0 && (kLdsFeat % 4) == 0, "alignment");
static_assert(kOffWRgb == kOffWSigma + kWidth && kOffBHead == kOffWSigma + 4 * kWidth, "head block is contiguous");

struct SirenK {
    const float* packed;
    const float* film;         // (batch, 9, 2, 256)
    // render mode
    const float* c2w; const float* focal; const float* near; const float* far; const float* t_vals;
    const float* tex_alpha; const float* tex_beta;
    float sigmoid_beta, box_scale, mask_thresh;
    int batch, H, Wd, S, res, force_bg;
    int R, tiles_per_img;
    float *rgb, *features, *xyz, *depth, *mask, *sdf, *weights, *points, *rays_d, *viewdirs, *dists;
    // points mode
    const float* pts; const float* vdirs; long long n_pts; int subtiles_per_wg, wgs_per_img;
    float* raw;
    float* save_args;          // training: (points, 9, 256) pre-sine arguments of every layer, or null
    int save_blocked;          // save_args in the slab-major layout of the 8-wave backward-type kernels (saved_row_floats below)
    // backbone hand-over between the two passes of one evaluated image (siren16_kernel<0, false, CACHE>): pass #1 writes the packed
    // (hi, lo) output of layer 7 per 16-point slab, pass #2 (texture FiLM) reads it and the composite weights instead of
    // recomputing layers 0..7, the sdf head and the transmittance scan -- they do not depend on the texture conditions
    void* bb_out; const void* bb_in; const float* weights_in; int bb_subs;
};

void siren_record_layout(int batch, int height, int width, int n_samples, int* R, int* tiles_per_img, int* subs);   // siren.hip

// ---- layouts of the saved state (pre-sine arguments: L = 9 layers per point; r_l, ta_l r_l: L = 8) ----
//   point-major: row p = the L x 256 floats of point p (E3DGE_PREC_F32 / _F16X3 backward-type kernels).
//   slab-major (E3DGE_PREC_F16X3_G2): 16 consecutive rows form a slab [L layers][16 tiles of 16 features][lane = 16 q + n][4 floats],
//       n = row & 15, feature = 16 tile + 4 q + j: the 16 points x 16 features of one (layer, tile) are ONE contiguous KiB in exactly the
//       order the 64 lanes of an 8-wave kernel hold them (C/D fragment of v_mfma_f32_16x16x32_f16) -- a wave's store / stream DMA of a
//       tile is 8 full 128-byte lines of one DRAM page instead of sixteen 64-byte pieces in sixteen rows 9 KiB apart.  Rows per image
//       are padded to a multiple of 16; slab s starts where row 16 s would start, the tensor is otherwise the same size.
constexpr int kSlabLayerF = 16 * kWidth;          // floats between the layers of a slab
constexpr int kSlabTileF = kWidth;                // floats between the 16-feature tiles of a slab layer
__host__ __device__ __forceinline__ int64_t saved_rows_per_image(bool blocked, int64_t n) { return blocked ? ((n + 15) & ~(int64_t)15) : n; }
// float offset of the 4 values [feature 4 q .. 4 q + 3 of tile 0, layer 0] of row `row` (rows counted over the padded images)
__device__ __forceinline__ int64_t saved_row_floats(bool blocked, int64_t row, int q, int L) {
    return blocked ? (row >> 4) * L * kSlabLayerF + (q * 16 + (int)(row & 15)) * 4 : row * L * kWidth + q * 4;
}
// float offset of ONE value (row, layer, feature)
__device__ __forceinline__ int64_t saved_elem_floats(bool blocked, int64_t row, int layer, int feature, int L) {
    return blocked ? ((row >> 4) * L + layer) * kSlabLayerF + (feature >> 4) * kSlabTileF + (((feature >> 2) & 3) * 16 + (int)(row & 15)) * 4 + (feature & 3)
                   : (row * L + layer) * kWidth + feature;
}

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// C/D fragment of v_mfma_f32_32x32x2_f32: lane l, register r holds D[row_of(r, l>>5)][l&31].
__device__ __forceinline__ constexpr int row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float f16lo(unsigned p) { return (float)__builtin_bit_cast(fp16x2, p).x; }
__device__ __forceinline__ float f16hi(unsigned p) { return (float)__builtin_bit_cast(fp16x2, p).y; }
// fp32 pair -> packed f16 (hi word, lo word) with x = hi + lo up to 2^-21 |x| (v_cvt_pkrtz rounds toward zero, so the
// remainder is exact in fp32 and has the sign of x).  Simulated against float64 this split with three products
// (hi*hi + hi*lo + lo*hi, fp32 accumulate) is as accurate as plain fp32 in this network.
struct HiLo { unsigned h, l; };
__device__ __forceinline__ HiLo split2(float x0, float x1) {
    HiLo p;
    p.h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    // residual x - (float)hi in ONE instruction per value: v_fma_mix_f32 reads the f16 half directly (the compiler emits
    // v_cvt_f32_f16 + v_sub_f32 for the C expression; the product with -1 is exact, so the result is the same bits)
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(p.h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(p.h), "v"(x1));
    p.l = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
    return p;
}
// (vector elements cannot be bound to references, hence the macro)
#define SPLIT2_TO(x0, x1, H, L) do { const HiLo p_ = split2((x0), (x1)); (H) = p_.h; (L) = p_.l; } while (0)

// value r (0..15) of feature tile c from the packed (hi, lo) representation
__device__ __forceinline__ float acts16_get(const u32x4 (&aH)[2 * kNT], const u32x4 (&aL)[2 * kNT], int c, int r) {
    const int g = 2 * c + (r >> 3), k = (r & 7) >> 1;
    return (r & 1) ? f16hi(aH[g][k]) + f16hi(aL[g][k]) : f16lo(aH[g][k]) + f16lo(aL[g][k]);
}

// LDS-DMA (global_load_lds_dwordx4: 16 B per lane straight into LDS, wave-uniform LDS base in M0).  The instruction's
// immediate offset is added to BOTH the global and the LDS address (validated on gfx950), so with the chunk image laid
// out identically on both sides the pieces of a chunk differ only in that immediate.
// Addressing: scalar 64-bit base + per-lane 32-bit byte offset + immediate.  hipcc never selects this mode for
// __builtin_amdgcn_global_load_lds (it builds a 64-bit VGPR address per piece: ~10 instructions and two VGPRs each
// time); written out, a piece is s_mov m0 / s_nop / global_load_lds.  No other code in these kernels uses M0.  The
// compiler does not count these in vmcnt; every consumer waits with an explicit vmcnt(0).
template <int OFF_BYTES>
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr), "n"(OFF_BYTES) : "memory");
}

// The weight-chunk pipeline shared by every kernel: 32 KiB chunks (one 32-row output tile x K=256) of a fragment image
// stream L2 -> LDS through kNBuf buffers, eight 16-B LDS-DMA pieces per lane and chunk, handed out one per MFMA group
// pair inside the tile two chunks earlier.  sync() runs early in every tile g: each wave drains its own DMA (chunk g+1,
// issued one whole tile earlier), the barrier publishes it and proves that everybody has left tile g-1, whose buffer
// the DMA of chunk g+2 may now overwrite.  Chunk indices wrap inside [first, first+count): the two chunks fetched past
// the end of the work land in buffers nobody reads (kernels end with vmcnt(0)).
struct ChunkPipe {
    const char* img;          // image + this wave's 8 KiB slice (wave-uniform)
    const char* src;          // chunk being issued
    uint32_t voff;            // lane * 16
    uint32_t lds_base;        // LDS byte address of wbuf + this wave's slice
    uint32_t lds_dst;         // ... of the buffer being filled
    int idx, first, count, buf, use_buf;
    float* wbuf;
    const float* wcur;
    const float* wnxt;
#if defined(E3DGE_PHASE_TIMING) || defined(E3DGE_BWD_TIMING)
    unsigned long long t_vm, t_bar;
#endif
    __device__ __forceinline__ void init(float* wbuf_, const float* image, int wave, int lane, int first_, int count_) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        wbuf = wbuf_;
        img = reinterpret_cast<const char*>(image) + wave_u * 8192;
        voff = (uint32_t)lane * 16u;
        lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)wbuf_ + (uint32_t)wave_u * 8192u;
        first = first_; count = count_;
        idx = first_; buf = 0; use_buf = 0;
        src = img + (size_t)idx * (kChunkFloats * 4);
        lds_dst = lds_base;
        wcur = wbuf_; wnxt = wbuf_ + kChunkFloats;
#if defined(E3DGE_PHASE_TIMING) || defined(E3DGE_BWD_TIMING)
        t_vm = 0; t_bar = 0;
#endif
    }
    __device__ __forceinline__ void issue_piece(int i) {     // i is a compile-time constant at every call site
#ifndef E3DGE_ABL_NODMA
        const char* s = src + (i >> 2) * 4096;
        const uint32_t d = lds_dst + (uint32_t)(i >> 2) * 4096u;
        switch (i & 3) {
            case 0: glds16_saddr<0>(s, voff, d); break;
            case 1: glds16_saddr<1024>(s, voff, d); break;
            case 2: glds16_saddr<2048>(s, voff, d); break;
            default: glds16_saddr<3072>(s, voff, d); break;
        }
#endif
        if (i == 7) {
            idx = (idx + 1 == first + count) ? first : idx + 1;
            src = img + (size_t)idx * (kChunkFloats * 4);
            buf = (buf + 1 == kNBuf) ? 0 : buf + 1;
            lds_dst = lds_base + (uint32_t)buf * (kChunkFloats * 4);
        }
    }
    __device__ __forceinline__ void prime() {
        for (int c = 0; c < kNBuf - 1; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) issue_piece(i);
    }
    __device__ __forceinline__ void sync() {
#if defined(E3DGE_PHASE_TIMING) || defined(E3DGE_BWD_TIMING)
        const unsigned long long c0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c1 = __builtin_readcyclecounter();
        __syncthreads();
        t_vm += c1 - c0; t_bar += __builtin_readcyclecounter() - c1;
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
    }
    __device__ __forceinline__ void advance() {
        use_buf = (use_buf + 1 == kNBuf) ? 0 : use_buf + 1;
        wcur = wnxt;
        wnxt = wbuf + ((use_buf + 1 == kNBuf) ? 0 : use_buf + 1) * kChunkFloats;
    }
};

// Two sines, both with an exact FMA Cody-Waite range reduction (|x| < ~1e5):
//  * sin_hw_f32 (default, 5 VALU ops): x/2pi reduced to [-0.5, 0.5] revolutions, hardware v_sin_f32.  Max abs error 3e-7.
//  * sin_poly_f32 (13 VALU ops): reduce to |r| <= pi/2, degree-9 minimax odd polynomial (4.7e-9 in exact
//    arithmetic), sign from k's parity.  Max abs error 1.2e-7.
// fp32 MFMA and fp32 VALU do not overlap on gfx950, so every VALU op of the epilogue is paid in full; the
// renderer's parity against the reference is the same with either (features ~1e-5, the summation-order noise).
// -DE3DGE_POLY_SINE selects the polynomial for the kernels.
// x / 2pi - rint(x / 2pi) in [-0.5, 0.5] to one rounding: the first product only picks the period, the fused
// multiply-add re-forms it exactly, the second adds the low part of 1/2pi.  4 ops; 3.0e-8 revolutions (1.9e-7 rad) max
// error over |x| < 80 -- the subtract-2pi-multiples form needed 5 ops for 3.3e-7 rad.
__device__ __forceinline__ float revolutions_f32(float x) {
    const float kf = rintf(x * 0.15915494f);
    float r = fmaf(x, 0.15915494f, -kf);
    return fmaf(x, 6.4206382e-09f, r);
}
__device__ __forceinline__ float sin_hw_f32(float x) { return __builtin_amdgcn_sinf(revolutions_f32(x)); }
// cos of the SAVED argument (backward kernels: d/dx sin)
__device__ __forceinline__ float cos_hw_f32(float x) { return __builtin_amdgcn_cosf(revolutions_f32(x)); }
__device__ __forceinline__ float sin_poly_f32(float x) {
    const float kf = rintf(x * 0.318309886183790672f);
    float r = fmaf(-kf, 3.1415927410125732f, x);
    r = fmaf(-kf, -8.742277657347586e-08f, r);
    const float r2 = r * r;
    float p = fmaf(r2, 2.6003292532550404e-06f, -1.9806761702056974e-04f);
    p = fmaf(p, r2, 8.33301991224289e-03f);
    p = fmaf(p, r2, -1.6666656732559204e-01f);
    const float sv = fmaf(r * r2, p, r);
    const unsigned sign = ((unsigned)(int)kf) << 31;
    return __uint_as_float(__float_as_uint(sv) ^ sign);
}
__device__ __forceinline__ float sin_f32(float x) {
#ifdef E3DGE_POLY_SINE
    return sin_poly_f32(x);
#else
    return sin_hw_f32(x);
#endif
}

__device__ __forceinline__ float sigmoid_f32(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, kWave); }
// lane index of the wavefront, re-derived on the spot (two VALU ops, not hoistable): a kernel-lifetime copy of threadIdx.x costs
// a register across the register-saturated tile loops -- where it is the value the allocator picks to spill
__device__ __forceinline__ int lane_id_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// K=256 contraction of one 32-feature output tile against the wave's register-resident activations.
//   TRANSPOSED=false: D[feature][point]  (weights = A operand, activations = B operand)
//   TRANSPOSED=true : D[point][feature]  (activations = A operand, weights = B operand)
// * The weight fragments are double-buffered in registers: each ds_read_b128 of k-block c+1 is issued ahead of
//   4 MFMAs of k-block c (one k-block = 1024 cycles of cover for the LDS latency).
// * `epi(r)`, r = 0..15, is the epilogue of the PREVIOUS output tile (FiLM + sine of one accumulator register, ~20
//   VALU ops); it is called once every second 4-MFMA group so that its VALU work issues in the shadow of this
//   tile's MFMAs (the matrix pipe is busy 64 cycles per MFMA, a VALU op takes 4) instead of after them.
struct NoEpilogue { __device__ __forceinline__ void operator()(int) const {} };

constexpr int kRing = 4;          // weight fragments held in registers (2 being consumed + 2 in flight)
constexpr int kSyncPair = 2;      // MFMA-group pair (even index) after which the chunk barrier happens; DMA pieces follow

template <bool TRANSPOSED, int VALU_PER_MFMA, class Epi, class Sync, class Dma>
__device__ __forceinline__ f32x16 big_tile(const float* __restrict__ wchunk, const float* __restrict__ wnext,
                                           int lane, const f32x16 (&in)[kNT], f32x16 acc, f32x4 (&ring)[kRing],
                                           Epi&& epi, Sync&& sync, Dma&& dma) {
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(wchunk) + lane;
    const f32x4* __restrict__ wn = reinterpret_cast<const f32x4*>(wnext) + lane;
    constexpr int kGroups = kNT * 4;     // 32 groups of 4 MFMAs (one ds_read_b128 each)
    // On entry ring[0], ring[1] hold groups 0 and 1 of this chunk (fetched by the previous tile's tail or the
    // prologue); on exit they hold groups 0 and 1 of the NEXT chunk, so consecutive tiles run back to back.
#pragma unroll
    for (int gp = 0; gp < kGroups; gp += 2) {
#pragma unroll
        for (int g = gp + 2; g < gp + 4; ++g)
            ring[g % kRing] = (g < kGroups) ? wp[g * 64] : wn[(g - kGroups) * 64];
        __builtin_amdgcn_sched_barrier(0);              // keep the prefetch ahead of the MFMAs it covers
#pragma unroll
        for (int g = gp; g < gp + 2; ++g) {
            const f32x4 w4 = ring[g % kRing];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float act = in[g >> 2][4 * (g & 3) + j];
                acc = TRANSPOSED ? mfma32(act, w4[j], acc) : mfma32(w4[j], act, acc);
            }
        }
        if (gp == kSyncPair) sync();
        if (gp > kSyncPair && gp <= kSyncPair + 16) dma((gp - kSyncPair) / 2 - 1);   // one DMA piece per group pair
        epi(gp >> 1);
        if (VALU_PER_MFMA > 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);   // n VALU
            }
        }
    }
    return acc;
}

// The same contraction on the f16 matrix pipe: fp32 operands split into f16 hi + lo, three products per k-step
// (hi*hi, lo_w*hi_a, hi_w*lo_a) into one fp32 accumulator.  16 k-steps of K=16 per tile = 48 MFMAs of 32
// cycles (1536 vs 8192 for fp32), and this pipe runs concurrently with the VALU, so the epilogue hides under it.
#ifndef E3DGE_RING16
#define E3DGE_RING16 4
#endif
constexpr int kRing16 = E3DGE_RING16;   // k-steps whose (hi, lo) weight fragments are held: 1 consumed + the rest in flight
constexpr int kSyncStep16 = 2;    // k-step after which the chunk barrier + next DMA issue happen

// Ablation switches for tools/ablate.sh (timing experiments only -- results are wrong when any is defined):
//   E3DGE_ABL_NOEPI  drop the pipelined epilogue VALU     E3DGE_ABL_NODMA  drop the weight DMA
//   E3DGE_ABL_NOLDS  drop the weight-fragment LDS reads   E3DGE_ABL_NOSYNC drop the chunk barrier
template <bool TRANSPOSED, class Epi, class Sync, class Dma>
__device__ __forceinline__ void big_tile_f16(const float* __restrict__ wchunk, const float* __restrict__ wnext,
                                             int lane, const u32x4 (&aH)[2 * kNT], const u32x4 (&aL)[2 * kNT],
                                             f32x16& acc, f32x16& accb, u32x4 (&ringH)[kRing16],
                                             u32x4 (&ringL)[kRing16], Epi&& epi, Sync&& sync, Dma&& dma) {
    const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(wchunk) + lane;
    const u32x4* __restrict__ wn = reinterpret_cast<const u32x4*>(wnext) + lane;
    constexpr int kSteps = 2 * kNT;
    // entry: ring slots 0..2 hold k-steps 0..2 of this chunk; exit: k-steps 0..2 of the next chunk
#pragma unroll
    for (int g = 0; g < kSteps; ++g) {
        const int ga = g + kRing16 - 1;
#ifndef E3DGE_ABL_NOLDS
        ringH[ga % kRing16] = (ga < kSteps) ? wp[(ga * 2 + 0) * 64] : wn[((ga - kSteps) * 2 + 0) * 64];
        ringL[ga % kRing16] = (ga < kSteps) ? wp[(ga * 2 + 1) * 64] : wn[((ga - kSteps) * 2 + 1) * 64];
#endif
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 wh = ringH[g % kRing16], wl = ringL[g % kRing16];
        // Two accumulators used alternately (a b a | b a b | ...): an instruction issued between two MFMAs that chain
        // on the SAME accumulator costs ~43 cycles (the accumulate-forwarding path is lost); with the interleaved
        // epilogue every MFMA would pay it.  The caller adds the two once per tile.
        f32x16& x0 = (g & 1) ? accb : acc;
        f32x16& x1 = (g & 1) ? acc : accb;
        if (!TRANSPOSED) {
            x0 = mfma16(wh, aH[g], x0);
            x1 = mfma16(wl, aH[g], x1);
            x0 = mfma16(wh, aL[g], x0);
        } else {
            x0 = mfma16(aH[g], wh, x0);
            x1 = mfma16(aH[g], wl, x1);
            x0 = mfma16(aL[g], wh, x0);
        }
#ifndef E3DGE_ABL_NOSYNC
        if (g == kSyncStep16) sync();
#endif
#ifndef E3DGE_ABL_NODMA
        if (g > kSyncStep16 && g <= kSyncStep16 + 8) dma(g - kSyncStep16 - 1);            // one DMA piece per k-step
#endif
#ifndef E3DGE_ABL_NOEPI
        epi(g);
#endif
        if (E3DGE_SPREAD16 > 0) {
            // f16 MFMAs co-execute with the VALU when the fillers sit BETWEEN consecutive MFMAs (about five single-issue
            // instructions hide per 32-cycle MFMA): lay the epilogue out as {MFMA, n VALU, LDS read} x 3 per k-step
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, E3DGE_SPREAD16, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

// sin(gamma * acc + beta), standard layout; gamma/beta of the layer come from the LDS copy of this image's
// FiLM block ([2][256], bias already folded into beta); one fused multiply-add feeds the sine.
// `save` (may be null): where this lane's point keeps the 256 arguments of the layer ([256] floats).
__device__ __forceinline__ f32x16 film_sin_std(f32x16 acc, const float* __restrict__ film_l, int t, int half,
                                               float* __restrict__ save = nullptr) {
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + 32 * t + 8 * q + 4 * half);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(film_l + kWidth + 32 * t + 8 * q + 4 * half);
        f32x4 arg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            arg[j] = fmaf(g4[j], acc[4 * q + j], b4[j]);
            o[4 * q + j] = sin_f32(arg[j]);
        }
        if (save) *reinterpret_cast<f32x4*>(save + 32 * t + 8 * q + 4 * half) = arg;
    }
    return o;
}

__device__ __forceinline__ void set_tile(f32x16 (&dst)[kNT], int t, const f32x16& v) {
    switch (t) {
        case 0: dst[0] = v; break;
        case 1: dst[1] = v; break;
        case 2: dst[2] = v; break;
        case 3: dst[3] = v; break;
        case 4: dst[4] = v; break;
        case 5: dst[5] = v; break;
        case 6: dst[6] = v; break;
        default: dst[7] = v; break;
    }
}


}  // namespace e3dge
