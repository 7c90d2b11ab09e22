// Packed decoder pipeline ("dec2"): the StyleGAN2 up-sampler of Decoder.forward (project/models/stylesdf_model.py:741-797)
// with every activation kept in the layout the f16 matrix pipe consumes (include/e3dge_hip.h, "Packed decoder pipeline").
//
// Why (round-3 measurements of modconv.hip): with fp32 planes between the kernels every convolution step
// had to load its input patch into registers, multiply by the style, split it into f16 hi/lo and write it to LDS -- and
// every co-block repeated that for the same patch.  The MFMA phase was 25-33 % of a step.  Here the PRODUCER of an
// activation does the split once, in its epilogue, and stores 16-byte entries of eight channels per pixel (hi plane, lo
// plane, one-entry zero border); a convolution stages both operands by LDS-DMA (global_load_lds_dwordx4) and its waves
// issue nothing but DMA pieces, ds_read_b128 and MFMAs until a tile's epilogue.  The modulation moves to the weights:
// w'' = ((scale W) s) demod per sample, rounded as the reference rounds it (:319-326), rebuilt by one streaming launch
// per forward.  The operand scale of a packed tensor comes from an a-priori bound (demodulated filters have unit norm,
// the FIR taps of a phase sum to one), so a producer can scale before it has seen its own maximum.
#include "decoder_common.h"
#include <stdlib.h>
#include <math.h>
#include <type_traits>

namespace e3dge {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kPkSlab = 9 * 2 * 1024;            // bytes of one weight slab: (32 co) x (16 ci) x 9 taps x (hi, lo)

// -DE3DGE_PK_TIMING: waves 0 and NW-1 of workgroup 0 accumulate shader-cycle deltas per phase of a step (0: vmcnt + barrier,
// 1: DMA issue, 2: fragment reads + MFMAs, 3: epilogue) and leave them in the unused floats of the output amax buffer's first
// line (tools/dec2_check.py --timing).
#ifndef E3DGE_PK_OVL
#define E3DGE_PK_OVL 0
#endif
#ifndef E3DGE_PK_S1_PM
#define E3DGE_PK_S1_PM 1         // stride-1 conv: product-major MFMA order (0 = three dependent MFMAs per accumulator in a row)
#endif
#ifndef E3DGE_PK_ISSUE_TAPS
#define E3DGE_PK_ISSUE_TAPS 6    // the LDS-DMA pieces of the next step go out behind the MFMAs of this many taps (stride-1 and fused up-sampling kernels)
#endif
#ifndef E3DGE_PK_EPI_VALU
#define E3DGE_PK_EPI_VALU 8      // VALU instructions of a finished tile's epilogue scheduled behind each MFMA of the next tile
#endif
#ifdef E3DGE_PK_TIMING
#define PK_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#define PK_T_INIT unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter(); const unsigned long long tbegin = tlast
#define PK_T_DONE(NW_) do { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == (NW_) - 1) && a.out_amax) { \
        float* o_ = a.out_amax + 1 + (wave == 0 ? 0 : 8); \
        for (int i_ = 0; i_ < 4; ++i_) o_[i_] = (float)tacc[i_]; \
        o_[4] = (float)(__builtin_readcyclecounter() - tbegin); o_[5] = (float)nsteps; \
        if (wave == 0) for (int i_ = 0; i_ < 4; ++i_) a.out_amax[17 + i_] = (float)tacc[4 + i_]; } } while (0)
#else
#define PK_T(i) do { } while (0)
#define PK_T_INIT do { } while (0)
#define PK_T_DONE(NW_) do { } while (0)
#endif

__device__ __forceinline__ uint32_t lds_u32(const void* p) {
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {       // make wave-uniformity explicit for an SGPR operand
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
// q / d for 0 <= q < 2^24, d > 0 (float reciprocal + fix-up)
__device__ __forceinline__ int div_small(int q, int d, float rcp) {
    int i = (int)((float)q * rcp);
    if (i * d > q) --i;
    if ((i + 1) * d <= q) ++i;
    return i;
}
__device__ __forceinline__ float pow2_bits(unsigned biased) { return __uint_as_float(biased << 23); }
// value of the lane one below / one above in the wavefront (DPP wave_shr:1 / wave_shl:1; the end lanes read 0)
__device__ __forceinline__ float dpp_wave_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_wave_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 (B, C, R, R) <-> packed
// ---------------------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
pk_pack_kernel(uint32_t* __restrict__ out, int* __restrict__ meta, const float* __restrict__ x, const float* __restrict__ amax,
               int C, int R) {
    const int lane = threadIdx.x & 63;
    const unsigned eb = scale_exponent(amax_read(amax, lane));
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) meta[0] = (int)eb;
    const float sc = pow2_bits(268u - eb);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= R * R) return;
    const int y = p / R, xx = p - y * R, g = blockIdx.y, b = blockIdx.z, G = C >> 3;
    const int64_t HW = (int64_t)R * R;
    const float* __restrict__ src = x + ((int64_t)b * C + 8 * g) * HW + p;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[j * HW] * sc;
    u32x4 hi, lo;
#pragma unroll
    for (int w = 0; w < 4; ++w) SPLIT2_TO(v[2 * w], v[2 * w + 1], hi[w], lo[w]);
    const int64_t plane = (int64_t)(R + 2) * (R + 2);
    const int64_t e = ((int64_t)(b * G + g) * 2) * plane + (int64_t)(y + 1) * (R + 2) + xx + 1;
    reinterpret_cast<u32x4*>(out)[e] = hi;
    reinterpret_cast<u32x4*>(out)[e + plane] = lo;
}

__global__ void __launch_bounds__(256)
pk_unpack_kernel(float* __restrict__ x, const uint32_t* __restrict__ in, const int* __restrict__ meta, int C, int R) {
    const float inv = pow2_bits((unsigned)meta[0] - 14u);                 // 2^(eb - 141)
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= R * R) return;
    const int y = p / R, xx = p - y * R, g = blockIdx.y, b = blockIdx.z, G = C >> 3;
    const int64_t HW = (int64_t)R * R, plane = (int64_t)(R + 2) * (R + 2);
    const int64_t e = ((int64_t)(b * G + g) * 2) * plane + (int64_t)(y + 1) * (R + 2) + xx + 1;
    const u32x4 hi = reinterpret_cast<const u32x4*>(in)[e], lo = reinterpret_cast<const u32x4*>(in)[e + plane];
    float* __restrict__ dst = x + ((int64_t)b * C + 8 * g) * HW + p;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        dst[(2 * w) * HW] = (f16lo(hi[w]) + f16lo(lo[w])) * inv;
        dst[(2 * w + 1) * HW] = (f16hi(hi[w]) + f16hi(lo[w])) * inv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weights: wpre (once per weight update) and the per-sample images + ToRGB tables (once per forward)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pk_prepack_kernel(float* __restrict__ wpre, const float* __restrict__ w, float scale, int Co, int Ci, int n_chunks, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        const int j = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int tap = (int)(r % 9); r /= 9;
        const int c = (int)(r % n_chunks);
        const int t = (int)(r / n_chunks);
        const int co = 32 * t + (l & 31), ci = 16 * c + 8 * (l >> 5) + j;
        wpre[e] = (co < Co && ci < Ci) ? __fmul_rn(scale, w[((int64_t)co * Ci + ci) * 9 + tap]) : 0.0f;
    }
}

// Transposed arrangement for the data-gradient convolutions (round 5): rows = the forward conv's INPUT channels, reduction = its output
// channels, taps flipped for the stride-1 convs (d x = conv(d y, flip(w)^T)), as they are for the transposed-stride ones
// (d x = conv_stride2(d T, w^T)).  wpre_t[t][c][tap][lane][j] = scale * weight[16c + 8 (lane >> 5) + j][32t + (lane & 31)][flip ? 8 - tap : tap]
__global__ void __launch_bounds__(256)
pk_prepack_t_kernel(float* __restrict__ wpre, const float* __restrict__ w, float scale, int Co, int Ci, int n_chunks, int64_t n, int flip) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        const int j = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int tap = (int)(r % 9); r /= 9;
        const int c = (int)(r % n_chunks);
        const int t = (int)(r / n_chunks);
        const int ci = 32 * t + (l & 31), co = 16 * c + 8 * (l >> 5) + j;
        wpre[e] = (co < Co && ci < Ci) ? __fmul_rn(scale, w[((int64_t)co * Ci + ci) * 9 + (flip ? 8 - tap : tap)]) : 0.0f;
    }
}

// swap != 0 (transposed images): `style` is indexed by the image's ROW (the forward conv's input channel) and `demod` by its reduction
// index (the forward conv's output channel); the product keeps the forward's rounding order ((scale W) s) demod
struct PkWConv { const float* wpre; const float* style; const float* demod; uint32_t* img; int co, ci, n_chunks; int n_items; int64_t words; int swap; };
struct PkWRgb { const float* w; const float* style; float* wm; float scale; int ci; };
struct PkWTab { PkWConv conv[2 * E3DGE_DEC2_MAX_UP + 1]; PkWRgb rgb[E3DGE_DEC2_MAX_UP + 1]; int n_conv, n_rgb; };

__global__ void __launch_bounds__(256) pk_weights_kernel(const PkWTab tab) {
    const int layer = blockIdx.y, b = blockIdx.z;
    if (layer < tab.n_conv) {
        const PkWConv L = tab.conv[layer];
        const float* __restrict__ st = (L.swap ? L.demod : L.style) + (size_t)b * L.ci;      // per reduction index
        const float* __restrict__ dm = (L.swap ? L.style : L.demod) + (size_t)b * L.co;      // per row
        u32x4* __restrict__ img = reinterpret_cast<u32x4*>(L.img + (size_t)b * L.words);
        for (int item = blockIdx.x * 256 + threadIdx.x; item < L.n_items; item += gridDim.x * 256) {
            const int l = item & 63;
            int r = item >> 6;
            const int tap = r % 9; r /= 9;
            const int c = r % L.n_chunks;
            const int t = r / L.n_chunks;
            const int co = 32 * t + (l & 31), ci0 = 16 * c + 8 * (l >> 5);
            const f32x4 w0 = reinterpret_cast<const f32x4*>(L.wpre)[(size_t)item * 2], w1 = reinterpret_cast<const f32x4*>(L.wpre)[(size_t)item * 2 + 1];
            const float d = co < L.co ? dm[co] : 0.0f;
            u32x4 hi, lo;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned hw = 0, lw = 0;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int j = 2 * w + e2;
                    const float wv = j < 4 ? w0[j] : w1[j - 4];
                    const float s = (ci0 + j) < L.ci ? st[ci0 + j] : 0.0f;
                    const float v = kW16Scale * (L.swap ? __fmul_rn(__fmul_rn(wv, d), s) : __fmul_rn(__fmul_rn(wv, s), d));   // ((scale W) s) demod, then the exact x128
                    const _Float16 h = (_Float16)v;
                    const _Float16 lv = (_Float16)(v - (float)h);
                    hw |= (unsigned)__builtin_bit_cast(unsigned short, h) << (16 * e2);
                    lw |= (unsigned)__builtin_bit_cast(unsigned short, lv) << (16 * e2);
                }
                hi[w] = hw; lo[w] = lw;
            }
            const size_t frag = ((size_t)(t * L.n_chunks + c) * 9 + tap) * 2;           // [co_tile][chunk][tap][hi|lo][lane]
            img[frag * 64 + l] = hi;
            img[(frag + 1) * 64 + l] = lo;
        }
    } else if (layer - tab.n_conv < tab.n_rgb) {
        const PkWRgb L = tab.rgb[layer - tab.n_conv];
        for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * L.ci; i += gridDim.x * 256) {
            const int ci = i % L.ci;
            L.wm[(size_t)b * 3 * L.ci + i] = __fmul_rn(__fmul_rn(L.scale, L.w[i]), L.style[(size_t)b * L.ci + ci]);   // (scale W) s, :321
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// convolutions
// ---------------------------------------------------------------------------------------------------------------------
struct PkConvK {
    const unsigned char* x;        // packed input (B, Ci/8, 2, H+2, W+2) x 16 B
    const unsigned char* wimg;     // per-sample weight images
    int64_t wimg_bytes;            // bytes per sample
    const int* in_meta;            // eb of the input
    const float* in_amax;          // measured max |input| (amax buffer)
    const float* noise; const float* noise_w; const float* noise_amax; const float* bias;
    unsigned char* y;              // stride-1: packed output (B, Co/8, 2, H+2, W+2)
    float* t;                      // up-sampling: fp32 (B, Co, 2H+3, 2W+4), T(y, x) at [y + 1][x + 2]
    int* out_meta;
    float* out_amax;
    // fused ToRGB of the last convolution (RGB variants): (B, 3, Co) table, (3) bias, skip (B, 3, H/2, W/2) or null, 4x4 FIR, out (B, 3, H, W)
    const float* rgb_wm; const float* rgb_bias; const float* rgb_skip; const float* rgb_fir; float* rgb_out;
    int rgb_store;                 // with rgb_out: 1 = also store the packed activation (a level that is not the last)
    float bias_amax, knorm, slope, act_scale;
    int B, Ci, Co, H, W;
    int n_chunks, noise_batch;
    int tiles_x, tiles_y, co_blocks, n_tiles;
    // up-sampling: positions (i, j) in [0, H] x [0, W] are processed per column block [j0, j0 + cwb), flattened q = i cwb + (j - j0)
    int cw, cwl, nblk, tpf, tpl;   // block width (all but the last / the last), blocks, WG tiles per full / last block
    // data-gradient launches (BWD, round 5; decoder2_bwd.h): the "input" x is a packed GRADIENT, the weight images are the transposed ones
    const unsigned char* mask_act; // packed FORWARD activation of the output's shape: the sign of its hi half selects lrelu' (1 or slope)
    const float* rgbt_d;           // (B, 3, H, W) gradient of this level's ToRGB output or null; its (scale W) s table is rgb_wm
    const float* rgbt_amax;        // amax buffer of rgbt_d
    const float* rgbt_l1;          // device scalar: max_{b, ci} sum_c |rgb_wm[b][c][ci]|
    const float* bwd_wl1;          // device scalar: bound on max_{b, row} sum |w''| of the transposed image (operator norm, max-abs)
    float* out_f32;                // BWD = 2: fp32 (B, Co, H, W) output (the gradient of the feature map), no mask
};

// ---- epilogue of one 32-channel x 32-pixel accumulator tile of a data-gradient convolution ------------------------------------------
// d x = acc 2^-s (+ ToRGB^T d rgb) ; d pre = lrelu'(forward activation) d x sqrt 2 (fused_bias_act grad = 1: (g alpha) scale for a
// non-positive reference, op/fused_act.py:19-50 of the reference) -> f16 hi/lo split, packed store, max |.| tracked.  FINAL: the plain
// fp32 (B, Co, H, W) store of d features.  Lane (half, col) holds rows 8 g4 + 4 half + j of the tile for pixel column col.
// The sign words of the tile's four channel groups (the lane's 8 bytes of each packed hi entry): requested one step BEFORE the tile's
// last chunk, so that they have landed by that step's vmcnt(0) + barrier -- loaded inside the epilogue they waited for a cold line AND
// (vmcnt retires in order) for every LDS-DMA piece of the next step issued since.
__device__ __forceinline__ void bwd_mask_load(const PkConvK& a, int b, int cot, int oy, int ox, int half, uint2 (&mw)[4]) {
    const int WP = a.W + 2, GO = a.Co >> 3;
    const int64_t plane_b = (int64_t)(a.H + 2) * WP * 16;
    // (branch-free: clamped into the image; a load inside a divergent `if` gets a wait of its own)
    const int64_t em = ((int64_t)(b * GO + cot * 4) * 2) * plane_b + ((int64_t)(min(oy, a.H - 1) + 1) * WP + min(ox, a.W - 1) + 1) * 16 + half * 8;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) mw[g4] = *reinterpret_cast<const uint2*>(a.mask_act + em + (int64_t)g4 * 2 * plane_b);
}

template <bool FINAL>
__device__ __forceinline__ void bwd_tile_epilogue(const PkConvK& a, const f32x16& d, int b, int cot, int oy, int ox, int half,
                                                  const float* rgb_tab, int tab_stride, int tab_co0, const float (&dr)[3], const uint2 (&mw)[4],
                                                  float oscale, float sc_out, float& amax_l) {
    const bool ok = oy < a.H && ox < a.W;
    if (FINAL) {
        if (ok) {
            const int64_t hw = (int64_t)a.H * a.W;
            float* o = a.out_f32 + ((int64_t)b * a.Co + cot * 32 + 4 * half) * hw + (int64_t)oy * a.W + ox;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(int64_t)((r & 3) + 8 * (r >> 2)) * hw] = d[r] * oscale;
        }
        return;
    }
    const int WP = a.W + 2, GO = a.Co >> 3;
    const int64_t plane_b = (int64_t)(a.H + 2) * WP * 16;
    const int64_t grp = ((int64_t)(b * GO + cot * 4) * 2) * plane_b;
    const float gmul = a.act_scale * sc_out;
    float m = 0.0f;
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        unsigned hh[2][2], ll[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int g4 = 2 * gp + e, co0 = cot * 32 + 8 * g4 + 4 * half;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = d[4 * g4 + j] * oscale;
            if (rgb_tab) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(rgb_tab + c * tab_stride + (co0 - tab_co0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaf(w4[j], dr[c], v[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned hb = ((j < 2 ? mw[g4].x : mw[g4].y) >> (16 * (j & 1))) & 0xffffu;
                const bool pos = hb - 1u < 0x7fffu;                                       // f16 bits 0x0001 .. 0x7fff: > 0
                const float t = pos ? v[j] : v[j] * a.slope;
                v[j] = t * gmul;
                m = fmaxf(m, fabsf(v[j]));
            }
            SPLIT2_TO(v[0], v[1], hh[e][0], ll[e][0]);
            SPLIT2_TO(v[2], v[3], hh[e][1], ll[e][1]);
        }
        auto r0 = __builtin_amdgcn_permlane32_swap(hh[0][0], hh[1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(hh[0][1], hh[1][1], false, false);
        auto r2 = __builtin_amdgcn_permlane32_swap(ll[0][0], ll[1][0], false, false);
        auto r3 = __builtin_amdgcn_permlane32_swap(ll[0][1], ll[1][1], false, false);
        if (ok) {
            unsigned char* dst = a.y + grp + (int64_t)((2 * gp + half) * 2) * plane_b + ((int64_t)(oy + 1) * WP + ox + 1) * 16;
            u32x4 eh, el;
            eh[0] = r0[0]; eh[1] = r1[0]; eh[2] = r0[1]; eh[3] = r1[1];
            el[0] = r2[0]; el[1] = r3[0]; el[2] = r2[1]; el[3] = r3[1];
            *reinterpret_cast<u32x4*>(dst) = eh;
            *reinterpret_cast<u32x4*>(dst + plane_b) = el;
        }
    }
    if (ok) amax_l = fmaxf(amax_l, m);
}

// one LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS at lds_dst), global address = wave-uniform base + per-lane byte offset
__device__ __forceinline__ void dma_piece(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    glds16_saddr<0>(uniform_ptr(sbase), voff, (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst));
}

// XCD-aware tile order (MI355X_MICROARCH: block b runs on XCD b % 8, each XCD has its own L2): workgroup-tile t -> logical
// tile id such that every XCD walks one CONTIGUOUS range of logical ids; logical ids put the co-blocks of one pixel tile
// next to each other (they read the same input patch) and neighbouring pixel tiles after that (they share halos).
__device__ __forceinline__ int xcd_logical(int t, int n_tiles) {
    const int nq = n_tiles >> 3, nr = n_tiles & 7, xcd = t & 7, slot = t >> 3;
    return (xcd < nr ? xcd * (nq + 1) : nr * (nq + 1) + (xcd - nr) * nq) + slot;
}

// ---- stride-1 3x3, pad 1 ---------------------------------------------------------------------------------------------
// Workgroup tile: (32 NCT WCO) output channels x (NPY WY) rows x (32 NPX WX) columns; a wave owns NCT co-tiles x NPY x NPX
// pixel tiles of 32 columns.  Steps = (tile, 16-channel chunk) of a persistent workgroup; two LDS stages; per step ONE barrier:
//     wait for my DMA pieces of this step | barrier | 9 taps of MFMAs from this stage, the DMA pieces of step + 1 dealt out
//     behind the first six taps | [epilogue]
// The patch (+halo) of a step is four planes [k-half][hi|lo] of NPIX 16-byte entries, fetched as 1-KiB pieces whose lanes walk
// the patch row-major (the source address is per lane, the LDS side is linear); weight slabs are 18 pieces each.
// RGB = true (last convolution of the decoder, tile covers all output channels): ToRGB (:531-541) happens in the epilogue --
// the activation is reduced against the 3 x Co table (scale W) s from LDS, + bias + FIR-up-sampled skip -- and is never stored.
// RGB = 2 (round 4: a 64-channel level that is NOT the last): the same ToRGB in the epilogue AND the packed activation stored -- the
// stand-alone ToRGB launch re-read the whole activation (67 MB at 512^2) for three output channels.
// BWD (round 5): 1 = data-gradient launch (x = packed gradient, transposed weight image): the epilogue is bwd_tile_epilogue -- no noise /
// bias, + ToRGB^T d rgb when rgbt_d is given, lrelu' from the packed forward activation mask_act; 2 = the last one (fp32 d features).
template <int NCT, int NPY, int NPX, int WCO, int WY, int WX, int RGB, int BWD = 0>
__global__ void __launch_bounds__(64 * WCO * WY * WX) pkconv_s1_kernel(const PkConvK a) {
    static_assert(!RGB || WCO == 1, "fused ToRGB needs every output channel of a pixel in one wave");
    static_assert(!(RGB && BWD), "the fused ToRGB forms are forward kernels");
    constexpr int NW = WCO * WY * WX, NT = 64 * NW;
    constexpr int TH = NPY * WY, TW = 32 * NPX * WX, PH = TH + 2, PW = TW + 2, NPIX = PH * PW, NPP = (NPIX + 63) / 64;
    constexpr int NCTB = NCT * WCO, XPLANE = NPIX * 16, XST = 4 * XPLANE, WST = NCTB * kPkSlab, STAGE = XST + WST;
    constexpr int NWP = NCTB * 18, NPIECE = NWP + 4 * NPP, NPT = NPY * NPX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pk[];
    float* const tab = reinterpret_cast<float*>(smem_pk + 2 * STAGE);      // [Co] activation bias, then (RGB) [3][Co] table
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = wave % WX, wy = (wave / WX) % WY, wco = wave / (WX * WY);
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * a.n_chunks;
    if (nsteps <= 0) return;
    const int HP = a.H + 2, WP = a.W + 2, G = a.Ci >> 3, GO = a.Co >> 3;
    const int64_t plane_b = (int64_t)HP * WP * 16;                  // bytes of one (g, hl) plane (input and output alike)

    // operand scales: the input's eb from its producer; the output's from the a-priori bound (header comment)
    const unsigned eb_in = (unsigned)a.in_meta[0];
    const float oscale = pow2_bits(eb_in - 21u);                     // 1 / (128 * 2^(141 - eb_in))
    const float nw = a.noise ? a.noise_w[0] : 0.0f;
    float sc_out = 1.0f;
    if (BWD == 1) {
        // |d pre| <= sqrt 2 (max |d y| max_row sum |w''| + max |d rgb| max_ci sum_c |(scale W) s|): the operator norms come from
        // pk_bwd_bounds_kernel (deterministic), the maxima are measured by the producers of the two gradients
        const float rg = a.rgbt_d ? amax_read(a.rgbt_amax, lane) * a.rgbt_l1[0] : 0.0f;
        const float bound = a.act_scale * (amax_read(a.in_amax, lane) * a.bwd_wl1[0] * 1.002f + rg) * 1.001f;
        const unsigned eb_out = scale_exponent(bound);
        sc_out = pow2_bits(268u - eb_out);
        if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb_out;
    } else if (!BWD && RGB != 1) {
        const float nza = a.noise ? fabsf(nw) * amax_read(a.noise_amax, lane) : 0.0f;
        const float bound = a.act_scale * (amax_read(a.in_amax, lane) * a.knorm * 1.002f + nza + a.bias_amax) * 1.001f;
        const unsigned eb_out = scale_exponent(bound);
        sc_out = pow2_bits(268u - eb_out);
        if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb_out;
    }
    if (!BWD) for (int i = tid; i < a.Co; i += NT) tab[i] = a.bias[i];        // read after >= 1 step-top barrier
    int b_tab = -1;

    struct Pos { int k, c, b, cb, ty, tx; };
    auto tile_of = [&](Pos& p) {
        if (p.k >= my_tiles) return;
        int L = xcd_logical((int)blockIdx.x + p.k * (int)gridDim.x, a.n_tiles);
        p.cb = L % a.co_blocks; L /= a.co_blocks;
        p.tx = L % a.tiles_x; L /= a.tiles_x;
        p.ty = L % a.tiles_y; p.b = L / a.tiles_y;
    };
    auto advance = [&](Pos& p) { if (++p.c == a.n_chunks) { p.c = 0; ++p.k; tile_of(p); } };

    // The LDS-DMA pieces of a step, dealt out in STATIC slots (round 4).  Inside the main loop the pieces of step + 1 go out behind the
    // MFMAs of the first six taps: a dedicated issue phase had all eight waves stalled on the CU's one vector-memory path at the
    // same time (2.2-3.5 k cycles of a 12 k-cycle step, E3DGE_PK_TIMING).  Until round 4 piece i = wave + j NW was "a weight piece if
    // i < NWP, else patch piece i - NWP": what a slot j held depended on the wave index, so every slot carried both forms, their
    // 64-bit source bases were recomputed from (b, cb, c, ty, tx) behind every tap, and the kernel-lifetime scalars this kept alive
    // were spilled to VGPR lanes -- ~170 instructions per tap for two pieces, ~1,000 of the ~2,800 a wave issued per step
    // (54 of them MFMAs; the step was instruction-issue bound: 11-12 k cycles, both waves of a SIMD alike).  Now:
    //   weight slot j < NWS:   piece i = wave + j NW of the NCTB slabs (18 pieces each), source offset in a per-lane register
    //   patch slot (pl, r):    plane pl (static), piece pp = (wave + pl ROT) % NW + r NW of its NPP (the rotation spreads the
    //                          planes' leftover pieces over the waves), lane's patch entry packed (row << 8 | column) in a register
    // and the sources of step + 1 are computed ONCE at the top of a step (Src, pinned in scalar registers).
    constexpr int NWS = (NWP + NW - 1) / NW, PR = (NPP + NW - 1) / NW, ROT = NW >= 4 ? NW / 4 : 1;
    constexpr int NSLOT = NWS + 4 * PR, PPT = (NSLOT + E3DGE_PK_ISSUE_TAPS - 1) / E3DGE_PK_ISSUE_TAPS;
    uint32_t wvo[NWS];
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
        const int i = wave + j * NW, ct = i / 18, pc = i - ct * 18;
        wvo[j] = (uint32_t)lane * 16u + (uint32_t)((ct * a.n_chunks * 18 + pc) * 1024);
        asm volatile("" : "+v"(wvo[j]));
    }
    // (one register per plane: the entry of round r > 0 lies r NW 64 entries further on and is derived from round 0's when it is issued)
    uint32_t pkv[4];
    int rws[4];
    unsigned vmask = 0;                                  // bit pl PR + r: this wave has a piece in round r of plane pl
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
        const int pp0 = (wave + pl * ROT) % NW;
        rws[pl] = pl * XPLANE + pp0 * 1024;
        asm volatile("" : "+s"(rws[pl]));
        const int e = pp0 * 64 + lane, prow = e / PW, pcol = e - prow * PW;
        pkv[pl] = (pp0 < NPP && e < NPIX) ? (uint32_t)(prow << 8 | pcol) : 0xffffffffu;
        asm volatile("" : "+v"(pkv[pl]));
#pragma unroll
        for (int r = 0; r < PR; ++r) vmask |= (pp0 + r * NW < NPP ? 1u : 0u) << (pl * PR + r);
    }
    vmask = (unsigned)__builtin_amdgcn_readfirstlane((int)vmask);
    asm volatile("" : "+s"(vmask));
    const uint32_t plb = (uint32_t)plane_b;
    struct Src { uint32_t wlo, whi, xlo, xhi; int gy0, gx0; };
    auto src_of = [&](const Pos& ps) {
        const uint64_t w = reinterpret_cast<uint64_t>(a.wimg + (int64_t)ps.b * a.wimg_bytes + ((int64_t)(ps.cb * NCTB) * a.n_chunks + ps.c) * kPkSlab);
        const uint64_t x = reinterpret_cast<uint64_t>(a.x + ((int64_t)(ps.b * G + 2 * ps.c) * 2) * plane_b);
        Src sc;
        sc.wlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w); sc.whi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 32));
        sc.xlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x); sc.xhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32));
        sc.gy0 = ps.ty * TH; sc.gx0 = ps.tx * TW;
        asm volatile("" : "+s"(sc.wlo), "+s"(sc.whi), "+s"(sc.xlo), "+s"(sc.xhi), "+s"(sc.gy0), "+s"(sc.gx0));      // (computed here, not re-derived at every use)
        return sc;
    };
    auto issue = [&](const Src& sc, uint32_t xl, int s_lo, int s_hi) {         // xl: LDS address of the target stage
        const void* wsrc = reinterpret_cast<const void*>((uint64_t)sc.whi << 32 | sc.wlo);
        const void* xsrc = reinterpret_cast<const void*>((uint64_t)sc.xhi << 32 | sc.xlo);
#pragma unroll
        for (int sl = s_lo; sl < s_hi; ++sl) {
            if (sl < NWS) {
                if ((sl + 1) * NW <= NWP || wave + sl * NW < NWP)
                    glds16_saddr<0>(wsrc, wvo[sl], xl + (uint32_t)(XST + (wave + sl * NW) * 1024));
            } else {
                const int q = sl - NWS, pl = q & 3, r = q >> 2;
                constexpr int DQ = (NW * 64) / PW, DR = (NW * 64) % PW;
                uint32_t pk = pkv[pl];
                if (r > 0) asm volatile("" : "+v"(pk));     // (a fresh copy: derived entries hoisted out of the loop would cost the registers this saves)
                uint32_t prow = pk >> 8, pcol = pk & 255u;
#pragma unroll
                for (int i = 0; i < r; ++i) {
                    pcol += DR; prow += DQ;
                    const bool cy = pcol >= (uint32_t)PW;
                    pcol = cy ? pcol - PW : pcol; prow += cy ? 1u : 0u;
                }
                if (((vmask >> (pl * PR + r)) & 1u) && prow < (uint32_t)PH) {
                    // clamped into the padded image: tiles that overhang a small image read (and compute) garbage that is never stored
                    const int gy = min(sc.gy0 + (int)prow, HP - 1), gx = min(sc.gx0 + (int)pcol, WP - 1);
                    glds16_saddr<0>(xsrc, (uint32_t)(gy * WP + gx) * 16u + (uint32_t)pl * plb, xl + (uint32_t)(rws[pl] + r * NW * 1024));
                }
            }
        }
    };
    auto stage_lds = [&](int stage) {
        uint32_t xl = lds_u32(smem_pk) + (uint32_t)(stage * STAGE);
        asm volatile("" : "+s"(xl));
        return xl;
    };

    Pos p_cur{0, 0, 0, 0, 0, 0};
    tile_of(p_cur);
    Pos p_nx1 = p_cur; advance(p_nx1);
    issue(src_of(p_cur), stage_lds(0), 0, NSLOT);

    f32x16 acc[NCT][NPT];                                // the tile being accumulated
    f32x16 fin[NCT][NPT];                                // the finished tile: its epilogue runs under the NEXT tile's first MFMAs
    float nzr[NPT], nzf[NPT];                            // noise of the tile's pixels (being fetched / of the finished tile)
    float skr[RGB ? NPT : 1][3], skf[RGB ? NPT : 1][3];  // (RGB) the FIR-up-sampled skip image at the tile's pixels, idem
    float rgbp[RGB ? NPT : 1][3];                        // (RGB) partial channel sums of the finished tile
    float svr[RGB ? NPT : 1][3][4], fwr[RGB ? NPT : 1][4];   // (RGB) the four skip-image taps of each pixel and their FIR weights, as loaded
    Pos p_fin{0, 0, 0, 0, 0, 0};
    bool pending = false;
    float amax_l = 0.0f;
    const int prow0 = wy * NPY, pcol0 = wx * NPX * 32 + col;
    const float kmul = RGB == 1 ? a.act_scale : a.act_scale * sc_out;      // lrelu(t) * act_scale * 2^k == (lrelu(t) * act_scale) * 2^k exactly
    const float kinv = RGB == 1 ? 1.0f : 1.0f / sc_out;                    // (RGB = 2: the ToRGB sums carry 2^k too and shed it, exactly, at the end)
    // (BWD) the tile's sign words and d rgb values: requested at the top of the step BEFORE the last chunk's, pinned at the top of the last
    // chunk's step (everything has landed behind that step's vmcnt(0)); then the whole tile's epilogue, straight from the accumulators
    uint2 mwp[BWD == 1 ? NCT * NPT : 1][4];
    float drp[BWD == 1 ? NPT : 1][3];
    auto bwd_prefetch = [&]() {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int oy = p_cur.ty * TH + prow0 + pt / NPX, ox = p_cur.tx * TW + pcol0 + 32 * (pt % NPX);
            drp[pt][0] = drp[pt][1] = drp[pt][2] = 0.0f;
            if (a.rgbt_d) {
                const float* dp = a.rgbt_d + (int64_t)p_cur.b * 3 * a.H * a.W + (unsigned)(min(oy, a.H - 1) * a.W + min(ox, a.W - 1));
#pragma unroll
                for (int c = 0; c < 3; ++c) drp[pt][c] = dp[(int64_t)c * a.H * a.W];
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bwd_mask_load(a, p_cur.b, p_cur.cb * NCTB + wco * NCT + ct, oy, ox, half, mwp[ct * NPT + pt]);
        }
    };
    auto bwd_pin = [&]() {
#pragma unroll
        for (int i = 0; i < NCT * NPT; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(mwp[i][g4].x), "+v"(mwp[i][g4].y));
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) asm volatile("" : "+v"(drp[pt][0]), "+v"(drp[pt][1]), "+v"(drp[pt][2]));
    };
    auto epi_bwd = [&]() {
        const uint2 none[4] = {};
        const float zero3[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int oy = p_cur.ty * TH + prow0 + pt / NPX, ox = p_cur.tx * TW + pcol0 + 32 * (pt % NPX);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bwd_tile_epilogue<BWD == 2>(a, acc[ct][pt], p_cur.b, p_cur.cb * NCTB + wco * NCT + ct, oy, ox, half,
                                            (BWD == 1 && a.rgbt_d) ? tab + a.Co : nullptr, a.Co, 0, BWD == 1 ? drp[pt] : zero3,
                                            BWD == 1 ? mwp[ct * NPT + pt] : none, oscale, sc_out, amax_l);
        }
    };
    PK_T_INIT;

    // One slice of a finished tile's epilogue: pixel tile pt, co-tile ct, channel group g4 = 2 gp + e (four values per lane):
    // conv * 2^-s + noise_w noise + bias -> lrelu * sqrt 2 (:459-466, :500-507; same roundings), then either the f16 hi/lo split
    // -- after the second group of a pair one 16-byte store per plane -- or (RGB) the ToRGB partial sums.  NH = 4 NCT NPT slices
    // per tile, dealt out over the nine taps of the next step and interleaved with its MFMAs (sched_group_barrier below): in
    // program order behind the MFMAs they would not overlap at all -- a wave issues in order, and its partner on the SIMD runs
    // the same phase (measured: moving the epilogue into the MFMA phase without the interleave changed nothing).
    constexpr int NH = 4 * NCT * NPT;
    // E3DGE_PK_OVL: 0 never, 1 whenever two accumulator sets fit (NCT NPT <= 2; four tiles + their copy + the slices' temporaries
    // spill 252-348 B), 2 only for the fused-ToRGB kernels
    constexpr bool OVL = E3DGE_PK_OVL == 1 ? NCT * NPT <= 2 : (E3DGE_PK_OVL == 2 ? (RGB && NCT * NPT <= 2) : false);
    unsigned shw[2], slw[2];                         // words of the pair's first group, kept until the second is done
    float m8 = 0.0f;
    auto epi_slice = [&](int h) {
        const int e = h & 1, gp = (h >> 1) & 1, ct = (h >> 2) % NCT, pt = (h >> 2) / NCT;
        const int b = p_fin.b, cot = p_fin.cb * NCTB + wco * NCT + ct;
        const int oy = p_fin.ty * TH + prow0 + pt / NPX, ox = p_fin.tx * TW + pcol0 + 32 * (pt % NPX);
        const bool ok = oy < a.H && ox < a.W;
        const float nz = __fmul_rn(nw, nzf[pt]);
        const f32x16& d = fin[ct][pt];
        if (RGB && ct == 0 && gp == 0 && e == 0) { rgbp[pt][0] = 0.0f; rgbp[pt][1] = 0.0f; rgbp[pt][2] = 0.0f; }
        const int g4 = 2 * gp + e, co0 = cot * 32 + 8 * g4 + 4 * half;
        const f32x4 bs = *reinterpret_cast<const f32x4*>(tab + co0);
        float v[4];
        if (e == 0) m8 = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = __fadd_rn(fmaf(d[4 * g4 + j], oscale, nz), bs[j]);        // (acc * 2^-s is exact: one rounding, as conv + noise)
            v[j] = fmaxf(t, t * a.slope) * kmul;                                       // lrelu for 0 <= slope <= 1
            m8 = fmaxf(m8, fabsf(v[j]));
        }
        if (RGB) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(tab + a.Co + c * a.Co + co0);
                rgbp[pt][c] += fmaf(w4[3], v[3], fmaf(w4[2], v[2], fmaf(w4[1], v[1], w4[0] * v[0])));
            }
            if (ct == NCT - 1 && gp == 1 && e == 1) {   // last slice of this pixel tile: fold the halves, + bias + up-sampled skip
#pragma unroll
                for (int c = 0; c < 3; ++c) rgbp[pt][c] += __shfl_xor(rgbp[pt][c], 32, kWave);
                if (ok && half == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        a.rgb_out[((int64_t)b * 3 + c) * a.H * a.W + (int64_t)oy * a.W + ox] = (rgbp[pt][c] * kinv + a.rgb_bias[c]) + skf[pt][c];
                }
            }
            if (RGB == 1) return;
        }
        unsigned h0, l0, h1, l1;
        SPLIT2_TO(v[0], v[1], h0, l0);
        SPLIT2_TO(v[2], v[3], h1, l1);
        if (e == 0) { shw[0] = h0; shw[1] = h1; slw[0] = l0; slw[1] = l1; return; }
        if (ok) amax_l = fmaxf(amax_l, m8);
        // half exchange (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second): afterwards lanes 0-31
        // hold the whole 16-byte entry of group 2 gp, lanes 32-63 that of group 2 gp + 1 -> one 16-byte store per plane instead of
        // two 8-byte ones (the epilogue was store-issue bound, guide T21)
        auto r0 = __builtin_amdgcn_permlane32_swap(shw[0], h0, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(shw[1], h1, false, false);
        auto r2 = __builtin_amdgcn_permlane32_swap(slw[0], l0, false, false);
        auto r3 = __builtin_amdgcn_permlane32_swap(slw[1], l1, false, false);
        if (ok) {
            unsigned char* dst = a.y + ((int64_t)(b * GO + cot * 4 + 2 * gp + half) * 2) * plane_b + ((int64_t)(oy + 1) * WP + ox + 1) * 16;
            u32x4 eh, el;
            eh[0] = r0[0]; eh[1] = r1[0]; eh[2] = r0[1]; eh[3] = r1[1];
            el[0] = r2[0]; el[1] = r3[0]; el[2] = r2[1]; el[3] = r3[1];
            *reinterpret_cast<u32x4*>(dst) = eh;
            *reinterpret_cast<u32x4*>(dst + plane_b) = el;
        }
    };

    // (RGB) FIR-up-sampled skip image at pixel tile pt of the CURRENT tile from its four loaded taps.  DEFER: called behind tap 0 -- in
    // the step's prelude it put the loads' s_waitcnt, a full memory round trip (~4 k cycles per tile at the 1024^2 level,
    // E3DGE_PK_TIMING), in front of the step's MFMAs.  Not for the 64-channel store + ToRGB form: the 32 registers of raw taps
    // that stay live through tap 0 spill 108 B there.
    constexpr bool DEFER = !(RGB == 2 && NCT * NPT > 2);
    auto skip_combine = [&](int pt) {
        const int oy = p_cur.ty * TH + prow0 + pt / NPX, ox = p_cur.tx * TW + pcol0 + 32 * (pt % NPX);
        const int oyc = min(oy, a.H - 1), oxc = min(ox, a.W - 1), h2 = a.H >> 1, w2 = a.W >> 1;
        float fw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ky = (oyc & 1) + 2 * (q >> 1), iy = (oyc + ky - 2) >> 1, kx = (oxc & 1) + 2 * (q & 1), ix = (oxc + kx - 2) >> 1;
            fw[q] = (iy >= 0 && iy < h2 && ix >= 0 && ix < w2) ? fwr[pt][q] : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float uacc = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) uacc = fmaf(svr[pt][c][q], fw[q], uacc);     // (a tap with weight 0 leaves the chain's value unchanged)
            skr[pt][c] = uacc;
        }
    };

    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        // my pieces of this step have landed; after the barrier everybody's have, and nobody still reads the other stage
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        PK_T(0);
        const bool has_next = step + 1 < nsteps, last_chunk = p_cur.c == a.n_chunks - 1;
        const Src src_nx = src_of(p_nx1);                // (past the last step: unused)
        const uint32_t xl_nx = stage_lds(cur ^ 1);
        const bool do_epi = pending;                     // (a tile's epilogue runs in the step after its last chunk)
        if (BWD == 1) {                                  // (host: n_chunks >= 2)
            if (p_cur.c == a.n_chunks - 2) bwd_prefetch();
            if (last_chunk) bwd_pin();
        }
        if (p_cur.c == 0) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = zero16();
        }
        // (scale W) s of this tile's sample.  Written in the tile's SECOND step: during its first one other waves may still be reading the
        // previous sample's table for the finished tile's epilogue; the host requires n_chunks >= 2, so a barrier lies before its readers.
        // (BWD: in the tile's FIRST step -- its epilogue never overlaps a later step, so nobody reads the table across this barrier, and a
        // two-chunk tile reads it in its second step)
        if ((RGB || (BWD == 1 && a.rgbt_d)) && p_cur.c == (BWD ? 0 : 1) && p_cur.b != b_tab) {
            b_tab = p_cur.b;
            const int t2 = (int)__builtin_amdgcn_readfirstlane(tid >> 6) * 64 + lane_id_fresh();     // (a kernel-lifetime copy of tid was the value spilled here)
            for (int i = t2; i < 3 * a.Co; i += NT) tab[a.Co + i] = a.rgb_wm[(size_t)b_tab * 3 * a.Co + i];
        }
        // The epilogue's per-pixel inputs are requested here, while nothing else is in flight, and waited for behind tap 0 -- before
        // this step's DMA pieces go out: a wait inside the epilogue would also wait for every DMA piece issued since (vmcnt is in order).
        // (Branch-free: a load inside a divergent `if` gets its own s_waitcnt vmcnt(0) -- twelve serialized L2 round trips per pixel tile
        // for the skip image in the first version of this block.  Indices are clamped into the image, out-of-image taps get weight 0.)
        if (last_chunk && !BWD) {
            // One 64-bit base per image (wave-uniform, in SGPRs) and unsigned 32-bit per-lane offsets: as `p ? p[i64] : 0` every one of
            // the 26 loads per pixel tile carried its own branch and 64-bit address arithmetic (6.2 scalar instructions per MFMA in the
            // fused-ToRGB kernel, PMC).  A missing noise / skip image reads a finite dummy (the bias table) that gets weight 0.
            const bool has_nz = a.noise != nullptr, has_sk = RGB && a.rgb_skip != nullptr;
            const int h2 = a.H >> 1, w2 = a.W >> 1;
            const float* __restrict__ nbase = has_nz ? a.noise + (int64_t)(a.noise_batch > 1 ? p_cur.b : 0) * a.H * a.W : a.bias;
            const float* __restrict__ sbase = has_sk ? a.rgb_skip + (int64_t)p_cur.b * 3 * h2 * w2 : a.bias;
            const unsigned splane = has_sk ? (unsigned)(h2 * w2) : 0u;
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int oy = p_cur.ty * TH + prow0 + pt / NPX, ox = p_cur.tx * TW + pcol0 + 32 * (pt % NPX);
                const int oyc = min(oy, a.H - 1), oxc = min(ox, a.W - 1);
                nzr[pt] = nbase[has_nz ? (unsigned)(oyc * a.W + oxc) : 0u];
                if (RGB) {
                    // raw taps (skip_combine above)
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {        // upfirdn2d(skip, fir, up=2, pad=(2,1)): taps and order of e3dge_upfirdn2d
                        const int ky = (oyc & 1) + 2 * p2, iy = (oyc + ky - 2) >> 1;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int kx = (oxc & 1) + 2 * e, ix = (oxc + kx - 2) >> 1;
                            const unsigned off = has_sk ? (unsigned)(min(max(iy, 0), h2 - 1) * w2 + min(max(ix, 0), w2 - 1)) : 0u;
                            const float f = a.rgb_fir[(unsigned)((3 - ky) * 4 + (3 - kx))];
                            fwr[pt][2 * p2 + e] = has_sk ? f : 0.0f;
#pragma unroll
                            for (int c = 0; c < 3; ++c) svr[pt][c][2 * p2 + e] = sbase[c * splane + off];
                        }
                    }
                    if (!DEFER) skip_combine(pt);
                }
            }
        }
        // The nine taps exist twice -- with and without the finished tile's epilogue -- because the slices must share basic blocks
        // with the MFMAs for the scheduler to thread them in between (a runtime `if (do_epi)` around each slice kept them apart).
        auto taps = [&](auto epi_tag) {
            constexpr bool EPI = decltype(epi_tag)::value;
            const unsigned char* xb = smem_pk + cur * STAGE + (size_t)(half * 2) * XPLANE;
            const unsigned char* wb = smem_pk + cur * STAGE + XST + (size_t)(wco * NCT) * kPkSlab + lane * 16;
#ifdef E3DGE_PK_TIMING
            unsigned long long tl2 = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                u32x4 ah[NCT], al[NCT], bh[NPT], bl[NPT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    ah[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kPkSlab + (tap * 2 + 0) * 1024);
                    al[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kPkSlab + (tap * 2 + 1) * 1024);
                }
#pragma unroll
                for (int py = 0; py < NPY; ++py)
#pragma unroll
                    for (int px = 0; px < NPX; ++px) {
                        const int pix = (prow0 + py + ky) * PW + pcol0 + 32 * px + kx;
                        bh[py * NPX + px] = *reinterpret_cast<const u32x4*>(xb + pix * 16);
                        bl[py * NPX + px] = *reinterpret_cast<const u32x4*>(xb + XPLANE + pix * 16);
                    }
#if E3DGE_PK_S1_PM        // product-major: consecutive MFMAs go to different accumulators (cf. pkconv_upblur2_kernel)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = mfma16(ah[ct], bh[pt], acc[ct][pt]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = mfma16(al[ct], bh[pt], acc[ct][pt]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = mfma16(ah[ct], bl[pt], acc[ct][pt]);
#else
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        f32x16& d = acc[ct][pt];
                        d = mfma16(ah[ct], bh[pt], d);
                        d = mfma16(al[ct], bh[pt], d);
                        d = mfma16(ah[ct], bl[pt], d);
                    }
#endif
                if (EPI) {                               // this tap's slices of the finished tile, threaded between its MFMAs
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        if (h * 9 / NH == tap) epi_slice(h);
#pragma unroll
                    for (int i = 0; i < 3 * NCT * NPT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, E3DGE_PK_EPI_VALU, 0);   // then this many VALU instructions of the slices
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (tap == 0 && last_chunk && !BWD) {
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        asm volatile("" : "+v"(nzr[pt]));      // the compiler's vmcnt wait lands here
                        if (RGB && DEFER) skip_combine(pt);
                        if (RGB && !DEFER) asm volatile("" : "+v"(skr[pt][0]), "+v"(skr[pt][1]), "+v"(skr[pt][2]));
                    }
                }
                if (has_next && tap * PPT < NSLOT) issue(src_nx, xl_nx, tap * PPT, min((tap + 1) * PPT, NSLOT));
                if (tap % 3 == 2 || EPI) __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads of later taps from piling up
#ifdef E3DGE_PK_TIMING
                if (tap == 0 || tap == 8) { const unsigned long long n2_ = __builtin_readcyclecounter();
                    tacc[(EPI ? 4 : 6) + (tap == 8)] += n2_ - tl2; tl2 = n2_; }
#endif
            }
        };
        if (OVL && do_epi) taps(std::true_type{}); else taps(std::false_type{});
        PK_T(2);
        pending = false;
        if (BWD) {
            if (last_chunk) epi_bwd();
        } else if (last_chunk) {                         // hand the tile over; its epilogue runs during the next step (or after the loop)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) fin[ct][pt] = acc[ct][pt];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                nzf[pt] = nzr[pt];
                if (RGB) { skf[pt][0] = skr[pt][0]; skf[pt][1] = skr[pt][1]; skf[pt][2] = skr[pt][2]; }
            }
            p_fin = p_cur;
            pending = OVL;
            if (!OVL) {
#pragma unroll
                for (int h = 0; h < NH; ++h) epi_slice(h);
            }
        }
        PK_T(3);
        p_cur = p_nx1;
        advance(p_nx1);
    }
    if (RGB) __syncthreads();                              // the table of the last tile's sample may have been written in the last step
    if (pending) {
#pragma unroll
        for (int h = 0; h < NH; ++h) epi_slice(h);
    }
    if (RGB != 1 && BWD != 2 && a.out_amax) {
        int l2 = lane;
        asm volatile("" : "+v"(l2));          // (fresh permute addresses: sharing them with the prologue's amax_read keeps five registers alive across the kernel)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            amax_l = fmaxf(amax_l, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((l2 ^ off) << 2, __builtin_bit_cast(int, amax_l))));
        if (l2 == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * NW + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_l * kinv);
    }
    PK_T_DONE(NW);
}

// ---- stride-2 transposed 3x3 (conv_transpose2d, padding 0) by output phase ---------------------------------------------
// Position (i, j) in [0, H] x [0, W] produces out(2i + ey, 2j + ex); tap (ky, kx) feeds phase (ky & 1, kx & 1) from
// x[i - (ky == 2)][j - (kx == 2)]: 4 + 2 + 2 + 1 taps, four distinct input shifts.  Positions are FLATTENED inside column
// blocks of <= 129 columns (q = i cwb + jj), a workgroup tile is Q consecutive q: no tile is wasted on the +1 position per
// row/column (the planar kernel spent 1.55x / 1.27x the useful MFMA work at 64^2 / 128^2).  The patch is the rows
// [i_lo, i_hi + 1] x columns [j0, j0 + cwb] of the padded input.
template <int NCT, int NPT, int WCO, int WQ, int NPIXMAX>
__global__ void __launch_bounds__(64 * WCO * WQ) pkconv_up_kernel(const PkConvK a) {
    constexpr int NW = WCO * WQ, Q = 32 * NPT * WQ;
    constexpr int NCTB = NCT * WCO, XPLANE = NPIXMAX * 16, XST = 4 * XPLANE, WST = NCTB * kPkSlab, STAGE = XST + WST;
    constexpr int NWP = NCTB * 18;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pk[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave % WQ, wco = wave / WQ;
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * a.n_chunks;
    if (nsteps <= 0) return;
    const int HP = a.H + 2, WP = a.W + 2, G = a.Ci >> 3;
    const int64_t plane_b = (int64_t)HP * WP * 16;
    const int TR = 2 * a.H + 3, TP = 2 * a.W + 4;
    const float oscale = pow2_bits((unsigned)a.in_meta[0] - 21u);
    const int tiles_per_img = (a.nblk - 1) * a.tpf + a.tpl;

    struct Pos { int k, c, b, cb, j0, cwb, q0, i_lo, npix; };
    auto tile_of = [&](Pos& p) {
        if (p.k >= my_tiles) return;
        int L = xcd_logical((int)blockIdx.x + p.k * (int)gridDim.x, a.n_tiles);
        p.cb = L % a.co_blocks; L /= a.co_blocks;
        const int idx = L % tiles_per_img; p.b = L / tiles_per_img;
        int blk, qt;
        if (idx < (a.nblk - 1) * a.tpf) { blk = idx / a.tpf; qt = idx - blk * a.tpf; p.cwb = a.cw; }
        else { blk = a.nblk - 1; qt = idx - blk * a.tpf; p.cwb = a.cwl; }
        p.j0 = blk * a.cw;
        p.q0 = qt * Q;
        const int qlast = min(p.q0 + Q, (a.H + 1) * p.cwb) - 1;
        p.i_lo = p.q0 / p.cwb;
        p.npix = (qlast / p.cwb - p.i_lo + 2) * (p.cwb + 1);
    };
    auto advance = [&](Pos& p) { if (++p.c == a.n_chunks) { p.c = 0; ++p.k; tile_of(p); } };

    constexpr int NPW = (NWP + 4 * ((NPIXMAX + 63) / 64) + NW - 1) / NW, PPT = (NPW + 5) / 6;   // piece slots per wave (upper bound), per tap
    auto issue = [&](const Pos& ps, int stage, int j_lo, int j_hi) {
        const uint32_t xl = lds_u32(smem_pk + stage * STAGE), wl = xl + XST;
        const unsigned char* wsrc = a.wimg + (int64_t)ps.b * a.wimg_bytes + ((int64_t)(ps.cb * NCTB) * a.n_chunks + ps.c) * kPkSlab;
        const unsigned char* xsrc = a.x + ((int64_t)(ps.b * G + 2 * ps.c) * 2) * plane_b + ((int64_t)ps.i_lo * WP + ps.j0) * 16;
        const int pwr = ps.cwb + 1, npp = (ps.npix + 63) >> 6;
        const float rcp = 1.0f / (float)pwr;
        const int npiece = NWP + 4 * npp;
        for (int j = j_lo; j < j_hi; ++j) {
            const int i = wave + j * NW;
            if (i >= npiece) break;
            if (i < NWP) {
                const int ct = i / 18, pc = i - ct * 18;
                dma_piece(wsrc + (int64_t)ct * a.n_chunks * kPkSlab + pc * 1024, (uint32_t)lane * 16u, wl + ct * kPkSlab + pc * 1024);
            } else {
                const int p = i - NWP, pl = p / npp, pp = p - pl * npp;
                const int e = pp * 64 + lane;
                if (e < ps.npix) {
                    const int prow = div_small(e, pwr, rcp), pcol = e - prow * pwr;
                    dma_piece(xsrc + pl * plane_b, (uint32_t)(prow * WP + pcol) * 16u, xl + pl * XPLANE + pp * 1024);
                }
            }
        }
    };

    Pos p_cur{0, 0, 0, 0, 0, 1, 0, 0, 0};
    tile_of(p_cur);
    Pos p_nx1 = p_cur; advance(p_nx1);
    issue(p_cur, 0, 0, NPW);

    f32x16 acc[4][NCT][NPT];
    float amax_l = 0.0f;
    int pixb[NPT], pi[NPT], pj[NPT];        // patch index of the position's (a = 0, b = 0) entry; its (i, j); j < 0: no position
    int pwr_cur = 1;
    PK_T_INIT;

    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        PK_T(0);
        const bool has_next = step + 1 < nsteps;
        if (p_cur.c == 0) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) acc[ph][ct][pt] = zero16();
            const int cwb = p_cur.cwb, qn = (a.H + 1) * cwb;
            const float rcp = 1.0f / (float)cwb;
            pwr_cur = cwb + 1;
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int q = p_cur.q0 + (wq * NPT + pt) * 32 + col;
                const int qc = min(q, qn - 1);
                const int i = div_small(qc, cwb, rcp), jj = qc - i * cwb;
                pixb[pt] = (i - p_cur.i_lo) * pwr_cur + jj;
                pi[pt] = i;
                pj[pt] = q < qn ? p_cur.j0 + jj : -1;
            }
        }
        {
            const unsigned char* xb = smem_pk + cur * STAGE + (size_t)(half * 2) * XPLANE;
            const unsigned char* wb = smem_pk + cur * STAGE + XST + (size_t)(wco * NCT) * kPkSlab + lane * 16;
            u32x4 bh[NPT][4], bl[NPT][4];       // shift s = 2 a + b: rows i - 1 + a, columns j - 1 + b of the input
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int pix = pixb[pt] + (s >> 1) * pwr_cur + (s & 1);
                    bh[pt][s] = *reinterpret_cast<const u32x4*>(xb + pix * 16);
                    bl[pt][s] = *reinterpret_cast<const u32x4*>(xb + XPLANE + pix * 16);
                }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int s = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1), ph = (ky & 1) * 2 + (kx & 1);
                u32x4 ah[NCT], al[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    ah[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kPkSlab + (tap * 2 + 0) * 1024);
                    al[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kPkSlab + (tap * 2 + 1) * 1024);
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        f32x16& d = acc[ph][ct][pt];
                        d = mfma16(ah[ct], bh[pt][s], d);
                        d = mfma16(al[ct], bh[pt][s], d);
                        d = mfma16(ah[ct], bl[pt][s], d);
                    }
                if (has_next && tap * PPT < NPW) issue(p_nx1, cur ^ 1, tap * PPT, min((tap + 1) * PPT, NPW));
                if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
        PK_T(2);
        if (p_cur.c == a.n_chunks - 1) {                    // ---- epilogue: the four phases of a position as two float2 rows ----
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int cot = p_cur.cb * NCTB + wco * NCT + ct;
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    if (pj[pt] >= 0) {
                        float* tp = a.t + (((int64_t)p_cur.b * a.Co + cot * 32 + 4 * half) * TR + 2 * pi[pt] + 1) * TP + 2 * pj[pt] + 2;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float* tr = tp + (int64_t)((r & 3) + 8 * (r >> 2)) * TR * TP;
#pragma unroll
                            for (int ey = 0; ey < 2; ++ey) {
                                const float v0 = acc[2 * ey][ct][pt][r] * oscale, v1 = acc[2 * ey + 1][ct][pt][r] * oscale;
                                amax_l = fmaxf(amax_l, fmaxf(fabsf(v0), fabsf(v1)));
                                *reinterpret_cast<float2*>(tr + ey * TP) = make_float2(v0, v1);
                            }
                        }
                    }
                }
            }
        }
        PK_T(3);
        p_cur = p_nx1;
        advance(p_nx1);
    }
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if (lane == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * NW + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_l);
    }
    PK_T_DONE(NW);
}

// ---- up-sampling StyledConv in ONE kernel: transposed conv by output phase + blur + noise + bias + lrelu -> packed ------
// At the two highest resolutions the intermediate T = conv_transpose(x) (fp32, (2H+1)^2) is the dominant HBM traffic of the
// decoder (written by the conv, read by the blur: 268 MB of 615 MB at 1024^2).  Here it never leaves the CU: a workgroup owns a
// block of 16 x 32 POSITIONS (i0 - 1 .. i0 + 14) x (j0 - 1 .. j0 + 30) -- one position row per MFMA column tile, two rows per
// wave -- accumulates the four output phases over all input channels, then, eight output channels at a time, writes the
// 32 x 64 patch of T into LDS (zeros for positions outside the image = the blur's padding), blurs it (4x4 FIR, same tap order as
// e3dge_upfirdn2d), applies StyledConv's tail and stores 28 x 60 pixels of packed entries.  Recomputed halo: 512 / 420 positions.
// The operand scale of the output cannot come from max|T| (T is produced here): |T| <= amax_in sqrt(4 ci) (at most four taps
// of a unit-norm demodulated filter reach one output phase).
// NPT = position rows per wave: 2 -> 16 x 32 positions, 28 x 60 pixels per tile; 1 -> 8 x 32 positions, 12 x 60 pixels (more,
// smaller tiles for the deep low-resolution levels, whose 16-row tiling leaves half of the CUs without a tile).
constexpr int kUbTW = 30, kUbPC = kUbTW + 3;

template <int NPT>
__global__ void __launch_bounds__(512) pkconv_upblur_kernel(const PkConvK a, const float* __restrict__ fir) {
    constexpr int kUbTH = 8 * NPT - 2, kUbPR = kUbTH + 3, kUbNpix = kUbPR * kUbPC, kUbTlBytes = 8 * (16 * NPT) * 64 * 4;
    constexpr int TLR = 16 * NPT, ORows = 2 * kUbTH;                         // rows of the T patch / output rows of a tile
    constexpr int NW = 8, XPLANE = kUbNpix * 16, XST = 4 * XPLANE, WST = kPkSlab, STAGE = XST + WST;
    constexpr int NWP = 18, NPP = (kUbNpix + 63) / 64, NPIECE = NWP + 4 * NPP, NPW = (NPIECE + NW - 1) / NW, PPT = (NPW + 5) / 6;
    static_assert(kUbTlBytes <= 2 * STAGE, "the T patch aliases the staging buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pk[];
    float* const tl = reinterpret_cast<float*>(smem_pk);                     // [8 ch][TLR rows][64 cols], aliases the stages
    float* const bias_s = reinterpret_cast<float*>(smem_pk + 2 * STAGE);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int HP = a.H + 2, WP = a.W + 2, G = a.Ci >> 3, GO = a.Co >> 3, R = 2 * a.H;
    const int64_t plane_b = (int64_t)HP * WP * 16, oplane = (int64_t)(R + 2) * (R + 2);
    const float oscale = pow2_bits((unsigned)a.in_meta[0] - 21u);
    const float nw = a.noise ? a.noise_w[0] : 0.0f;
    const float nza = a.noise ? fabsf(nw) * amax_read(a.noise_amax, lane) : 0.0f;
    const float bound = a.act_scale * (amax_read(a.in_amax, lane) * a.knorm * 1.002f + nza + a.bias_amax) * 1.001f;   // knorm = sqrt(4 ci)
    const unsigned eb_out = scale_exponent(bound);
    const float kmul = a.act_scale * pow2_bits(268u - eb_out), kinv = 1.0f / pow2_bits(268u - eb_out);
    if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb_out;
    for (int i = tid; i < a.Co; i += 512) bias_s[i] = a.bias[i];
    float kf[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) kf[p][q] = fir[(3 - p) * 4 + (3 - q)];

    // blur-phase role of this thread: output row ry (0 .. ORows - 1), pixels 4 gx .. 4 gx + 3 (gx 0..14).  Sixteen lanes per row
    // (gx = 15 idles): with fifteen, the 16-lane groups of a ds_read_b128 straddled two rows of the 256-byte-pitch patch and a
    // quarter of the LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT 2.2e6 of 8.5e6, profiles/r3_decoder_pmc.txt).
    const int ry = tid >> 4, gx = tid & 15;
    const bool blur_thread = ry < ORows && gx < 15;
    float amax_l = 0.0f;
    PK_T_INIT;
    const int nsteps = my_tiles;

    for (int k = 0; k < my_tiles; ++k) {
        int L = xcd_logical((int)blockIdx.x + k * (int)gridDim.x, a.n_tiles);
        const int cb = L % a.co_blocks; L /= a.co_blocks;
        const int txi = L % a.tiles_x; L /= a.tiles_x;
        const int tyi = L % a.tiles_y, b = L / a.tiles_y;
        const int i0 = tyi * kUbTH, j0 = txi * kUbTW;                    // first position whose outputs this tile stores
        auto issue = [&](int c, int stage, int j_lo, int j_hi) {
            const uint32_t xl = lds_u32(smem_pk + stage * STAGE), wl = xl + XST;
            const unsigned char* wsrc = a.wimg + (int64_t)b * a.wimg_bytes + ((int64_t)cb * a.n_chunks + c) * kPkSlab;
            const unsigned char* xsrc = a.x + ((int64_t)(b * G + 2 * c) * 2) * plane_b;
            for (int j = j_lo; j < j_hi; ++j) {
                const int i = wave + j * NW;
                if (i >= NPIECE) break;
                if (i < NWP) {
                    dma_piece(wsrc + i * 1024, (uint32_t)lane * 16u, wl + i * 1024);
                } else {
                    const int p = i - NWP, pl = p / NPP, pp = p - pl * NPP;
                    const int e = pp * 64 + lane;
                    if (e < kUbNpix) {
                        const int prow = e / kUbPC, pcol = e - prow * kUbPC;
                        // padded input rows i0 - 1 + prow, clamped into the buffer: positions outside the image are zeroed below
                        const int gy = min(max(i0 - 1 + prow, 0), HP - 1), gx_ = min(max(j0 - 1 + pcol, 0), WP - 1);
                        dma_piece(xsrc + pl * plane_b, (uint32_t)(gy * WP + gx_) * 16u, xl + pl * XPLANE + pp * 1024);
                    }
                }
            }
        };
        __syncthreads();                                  // the previous tile's last blur round has finished reading tl
        issue(0, 0, 0, NPW);
        // the blur threads' noise (the same for every channel), requested while the tile's first weights are on their way
        float nzv[4] = {0.f, 0.f, 0.f, 0.f};
        const int oy = 2 * i0 + ry, ox0 = 2 * j0 + 4 * gx;
        if (blur_thread && a.noise) {
            const int oyc = min(oy, R - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                nzv[j] = a.noise[(int64_t)(a.noise_batch > 1 ? b : 0) * R * R + (int64_t)oyc * R + min(ox0 + j, R - 1)];
        }

        f32x16 acc[4][NPT];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[ph][pt] = zero16();
        for (int c = 0; c < a.n_chunks; ++c) {
            const int cur = c & 1;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            PK_T(0);
            const bool has_next = c + 1 < a.n_chunks;
            const unsigned char* xb = smem_pk + cur * STAGE + (size_t)(half * 2) * XPLANE;
            const unsigned char* wb = smem_pk + cur * STAGE + XST + lane * 16;
            u32x4 bh[NPT][4], bl[NPT][4];       // shift s = 2 a + b: input rows i - 1 + a, columns j - 1 + b
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) {
                    const int pix = (wave * NPT + pt + (sft >> 1)) * kUbPC + col + (sft & 1);
                    bh[pt][sft] = *reinterpret_cast<const u32x4*>(xb + pix * 16);
                    bl[pt][sft] = *reinterpret_cast<const u32x4*>(xb + XPLANE + pix * 16);
                }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int sft = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1), ph = (ky & 1) * 2 + (kx & 1);
                const u32x4 ah = *reinterpret_cast<const u32x4*>(wb + (tap * 2 + 0) * 1024);
                const u32x4 al = *reinterpret_cast<const u32x4*>(wb + (tap * 2 + 1) * 1024);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    f32x16& d = acc[ph][pt];
                    d = mfma16(ah, bh[pt][sft], d);
                    d = mfma16(al, bh[pt][sft], d);
                    d = mfma16(ah, bl[pt][sft], d);
                }
                if (has_next && tap * PPT < NPW) issue(c + 1, cur ^ 1, tap * PPT, min((tap + 1) * PPT, NPW));
                if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
            PK_T(1);
        }
        // ---- epilogue: eight output channels per round through the LDS patch of T ----
        // The patch is stored with CHANNEL PAIRS interleaved -- tl[pair][row][col][2] -- so that the FIR and the tail run on
        // v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two channels per instruction; the even-aligned register pairs come straight out
        // of the ds_read_b128): 256 instead of 512 FMA instructions per thread and round.  (Packed fp32 next to MFMAs is an
        // anti-lever, MI355X_MICROARCH; this phase has no MFMAs.)
        bool pos_ok[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int pi = i0 - 1 + wave * NPT + pt, pj = j0 - 1 + col;
            pos_ok[pt] = pi >= 0 && pi <= a.H && pj >= 0 && pj <= a.W;
        }
        f32x2 nz2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float t_ = a.noise ? __fmul_rn(nw, nzv[j]) : 0.0f; nz2[j] = f32x2{t_, t_}; }
        bool px_ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) px_ok[j] = oy < R && ox0 + j < R;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            __syncthreads();                              // staging buffers (first round) / the previous round's readers are done
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int prow = wave * NPT + pt;
#pragma unroll
                for (int q = 0; q < 2; ++q) {             // channel pair 2 half + q of the group: channels 4 half + 2 q, + 1
                    float* trow = tl + (((size_t)(2 * half + q) * TLR + 2 * prow) * 64 + 2 * col) * 2;
#pragma unroll
                    for (int ey = 0; ey < 2; ++ey) {
                        f32x4 v4;                           // (column 2 col: channels e = 0, 1), (column 2 col + 1: e = 0, 1)
                        v4[0] = pos_ok[pt] ? acc[2 * ey][pt][4 * g4 + 2 * q] * oscale : 0.0f;
                        v4[1] = pos_ok[pt] ? acc[2 * ey][pt][4 * g4 + 2 * q + 1] * oscale : 0.0f;
                        v4[2] = pos_ok[pt] ? acc[2 * ey + 1][pt][4 * g4 + 2 * q] * oscale : 0.0f;
                        v4[3] = pos_ok[pt] ? acc[2 * ey + 1][pt][4 * g4 + 2 * q + 1] * oscale : 0.0f;
                        *reinterpret_cast<f32x4*>(trow + ey * 128) = v4;
                    }
                }
            }
            __syncthreads();
            PK_T(2);
            if (blur_thread) {
                const int gout = cb * 4 + g4;
                u32x4 hi[4], lo[4];
                float m = 0.0f;
#pragma unroll
                for (int cp = 0; cp < 4; ++cp) {
                    f32x2 ac[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky) {
                        const float* row = tl + (((size_t)cp * TLR + ry + 1 + ky) * 64 + 4 * gx) * 2;     // tl row <-> y = 2 i0 - 2 + row
                        f32x2 in[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 q4 = *reinterpret_cast<const f32x4*>(row + 4 * i);
                            in[2 * i] = f32x2{q4[0], q4[1]};
                            in[2 * i + 1] = f32x2{q4[2], q4[3]};
                        }
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            const f32x2 kk = f32x2{kf[ky][kx], kf[ky][kx]};
#pragma unroll
                            for (int j = 0; j < 4; ++j) ac[j] = __builtin_elementwise_fma(in[j + kx + 1], kk, ac[j]);   // tl column <-> x = 2 j0 - 2 + column
                        }
                    }
                    const f32x2 bv = *reinterpret_cast<const f32x2*>(bias_s + gout * 8 + 2 * cp);
                    const f32x2 sl = f32x2{a.slope, a.slope}, km = f32x2{kmul, kmul};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x2 tv = (ac[j] + nz2[j]) + bv;                    // (conv + noise) + bias, as the unfused kernels round it
                        const f32x2 ls = tv * sl;
                        tv = f32x2{fmaxf(tv[0], ls[0]), fmaxf(tv[1], ls[1])} * km;     // lrelu (0 <= slope <= 1) * act_scale * 2^k
                        if (px_ok[j]) m = fmaxf(m, fmaxf(fabsf(tv[0]), fabsf(tv[1])));
                        SPLIT2_TO(tv[0], tv[1], hi[j][cp], lo[j][cp]);
                    }
                }
                amax_l = fmaxf(amax_l, m);
                if (oy < R) {
                    u32x4* __restrict__ dst = reinterpret_cast<u32x4*>(a.y) + ((int64_t)(b * GO + gout) * 2) * oplane + (int64_t)(oy + 1) * (R + 2) + ox0 + 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (px_ok[j]) { dst[j] = hi[j]; dst[oplane + j] = lo[j]; }
                }
            }
            PK_T(3);
        }
    }
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if (lane == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * NW + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_l * kinv);
    }
    PK_T_DONE(NW);
}

// ---- the same layer, second generation (round 4): separable FIR, horizontal pass in registers, prefetch under the epilogue ----
// What round 3's counters said about pkconv_upblur_kernel (profiles/r3_decoder_pmc.txt): 7 VALU per MFMA (the 4x4 FIR: ~1,100
// VALU per thread and round of eight channels), 36 % of the LDS cycles bank conflicts, 20 % matrix-pipe busy, and the first
// chunk of every tile waited for in the open (the T patch aliased BOTH staging buffers).  Here:
//  * the FIR is applied as its two 1-D factors (Blur's kernel is make_kernel([1,3,3,1]): rank one; the host checks and passes
//    the factor, anything else takes the first-generation kernel).  The HORIZONTAL pass runs on the accumulators themselves:
//    a lane holds position column j of its row, its neighbours j-1 / j+1 are the adjacent lanes (v_mov_dpp wave_shr / wave_shl;
//    the lanes at the ends of a 32-column tile are the halo columns whose results are never stored), 8 FMAs + 3 DPP moves per
//    (row, channel) for the two output columns 2j, 2j+1 -- no LDS traffic, no 7-column windows;
//  * what goes through LDS is H, laid out [channel half][T row][output column] in 16-byte elements (four channels): writers
//    (lane = position column, two adjacent elements) and readers (lane = output column, one element) are both lane-linear --
//    conflict-free by construction for ds_write_b128 / ds_read_b128 (MI355X_MICROARCH, LDS table);
//  * the VERTICAL pass: a wave owns 4 output rows x 64 output columns x 8 channels, reads 7 H rows (2 x ds_read_b128 each),
//    4 FMAs per output, then StyledConv's tail, the f16 split and fully coalesced stores (64 consecutive 16-byte entries per
//    instruction -- the first generation's threads held four adjacent pixels each and left 48-byte gaps between lanes);
//  * the H buffer lives behind staging buffer 0, so the NEXT tile's first chunk is fetched during the epilogue.
// Per thread and round: ~180 VALU (horizontal) + ~360 (vertical + tail + split, 7 of 8 waves) instead of ~1,100.
template <int NPT, int NW, int TC>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) pkconv_upblur2_kernel(const PkConvK a, const float g0, const float g1) {
    // TC = position columns of one MFMA column tile: 32 (one position row per tile) or 16 (two rows x 16 columns: square-ish
    // workgroup tiles -- a 65-wide level needs five 14-column tiles = 80 computed columns instead of three 30-column tiles = 96)
    static_assert(TC == 32 || TC == 16, "MFMA column tile = 1 x 32 or 2 x 16 positions");
    constexpr int SR = 32 / TC, PRW = NW * NPT * SR;                         // position rows per MFMA tile / per workgroup tile
    constexpr int kUbTH = PRW - 2, kUbTW = TC - 2, kUbPR = PRW + 1, kUbPC = TC + 1, kUbNpix = kUbPR * kUbPC;
    constexpr int TLR = 2 * PRW, ORows = 2 * kUbTH, OX = 2 * TC;                // T rows of a tile / output rows / output columns incl. halo
    constexpr int NG = NW * (64 / OX);                                       // vertical pass: row groups (a wave holds 64 / OX of them)
    constexpr int RPW = (ORows + NG - 1) / NG, NVW = ORows / RPW;            // output rows per group, groups taking part
    static_assert(NVW * RPW == ORows && NVW <= NG, "vertical-pass split");
    constexpr int NT = 64 * NW, XPLANE = kUbNpix * 16, XST = 4 * XPLANE, WST = kPkSlab, STAGE = XST + WST;
    constexpr int NWP = 18, NPP = (kUbNpix + 63) / 64;
    constexpr int HB = 2 * TLR * OX * 16;                                    // H buffer: [half][T row][OX columns] x 16 B
    constexpr int HOFF = STAGE, BOFF = (2 * STAGE > STAGE + HB ? 2 * STAGE : STAGE + HB);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pk[];
    unsigned char* const hl = smem_pk + HOFF;                                // aliases staging buffer 1 (and beyond), never buffer 0
    float* const bias_s = reinterpret_cast<float*>(smem_pk + BOFF);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & (TC - 1), srow = (lane & 31) / TC;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int HP = a.H + 2, WP = a.W + 2, G = a.Ci >> 3, GO = a.Co >> 3, R = 2 * a.H;
    const int64_t plane_b = (int64_t)HP * WP * 16, oplane = (int64_t)(R + 2) * (R + 2);
    const float oscale = pow2_bits((unsigned)a.in_meta[0] - 21u);
    const float nw = a.noise ? a.noise_w[0] : 0.0f;
    const float nza = a.noise ? fabsf(nw) * amax_read(a.noise_amax, lane) : 0.0f;
    const float bound = a.act_scale * (amax_read(a.in_amax, lane) * a.knorm * 1.002f + nza + a.bias_amax) * 1.001f;   // knorm = sqrt(4 ci)
    const unsigned eb_out = (unsigned)__builtin_amdgcn_readfirstlane((int)scale_exponent(bound));     // (wave-uniform: keep it, kmul and kinv in SGPRs)
    const float kmul = a.act_scale * pow2_bits(268u - eb_out), kinv = pow2_bits(eb_out - 14u);      // 2^-(141 - eb)
    if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb_out;
    for (int i = tid; i < a.Co; i += NT) bias_s[i] = a.bias[i];
    // The 1-D factor is SYMMETRIC (g0, g1, g1, g0: the host checks; make_kernel([1,3,3,1]) is), so flipping it (upfirdn2d correlates
    // with the flipped kernel) is the identity and two constants per pass do -- four register pairs of packed taps fewer, which is
    // what kept the 2 x 16 form from fitting 256 registers.  The accumulators' power-of-two scale rides on the horizontal taps (exact).
    const float fx0 = g0 * oscale, fx1 = g1 * oscale;
    const float fy0 = g0, fy1 = g1;
    float amax_l = 0.0f;
    PK_T_INIT;
    const int nsteps = my_tiles;

    struct Tile { int b, cb, i0, j0; };
    auto decode = [&](int k) {
        int L = xcd_logical((int)blockIdx.x + k * (int)gridDim.x, a.n_tiles);
        Tile t;
        t.cb = L % a.co_blocks; L /= a.co_blocks;
        const int txi = L % a.tiles_x; L /= a.tiles_x;
        const int tyi = L % a.tiles_y;
        t.b = L / a.tiles_y;
        t.i0 = tyi * kUbTH; t.j0 = txi * kUbTW;                            // first position whose outputs this tile stores
        return t;
    };
    // LDS-DMA pieces in static slots, as in pkconv_s1_kernel: weight slot j = piece wave + j NW of the slab's 18; patch slot (pl, r) =
    // piece (wave + pl ROT) % NW + r NW of plane pl.  Which entry of the (kUbPR x kUbPC) patch a lane fetches is a kernel constant,
    // kept PACKED (row << 8 | column, one register per slot); the source address is rebuilt per chunk from it (six VALU per piece,
    // in the shadow of the MFMAs): left to LICM, row, column and offset of every piece stay live across the K loop and the
    // epilogue -- three registers each, which is what spilled in the 2 x 16 form.  The 64-bit sources of a chunk are computed once
    // (Src), not behind every tap.
    constexpr int NWS = (NWP + NW - 1) / NW, PR = (NPP + NW - 1) / NW, ROT = NW / 4;
    constexpr int NSLOT = NWS + 4 * PR, PPT = (NSLOT + E3DGE_PK_ISSUE_TAPS - 1) / E3DGE_PK_ISSUE_TAPS;
    uint32_t pkv[4];                                     // (round 0; later rounds are derived from it when they are issued)
    int rws[4];
    unsigned vmask = 0;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
        const int pp0 = (wave + pl * ROT) % NW;
        rws[pl] = pl * XPLANE + pp0 * 1024;
        asm volatile("" : "+s"(rws[pl]));
        const int e = pp0 * 64 + lane, prow = e / kUbPC, pcol = e - prow * kUbPC;
        pkv[pl] = (pp0 < NPP && e < kUbNpix) ? (uint32_t)(prow << 8 | pcol) : 0xffffffffu;
        asm volatile("" : "+v"(pkv[pl]));
#pragma unroll
        for (int r = 0; r < PR; ++r) vmask |= (pp0 + r * NW < NPP ? 1u : 0u) << (pl * PR + r);
    }
    vmask = (unsigned)__builtin_amdgcn_readfirstlane((int)vmask);
    asm volatile("" : "+s"(vmask));
    const uint32_t plb = (uint32_t)plane_b;
    struct Src { uint32_t wlo, whi, xlo, xhi; int im, jm; };
    auto src_of = [&](const Tile& t, int c) {
        const uint64_t w = reinterpret_cast<uint64_t>(a.wimg + (int64_t)t.b * a.wimg_bytes + ((int64_t)t.cb * a.n_chunks + c) * kPkSlab);
        const uint64_t x = reinterpret_cast<uint64_t>(a.x + ((int64_t)(t.b * G + 2 * c) * 2) * plane_b);
        Src sc;
        sc.wlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w); sc.whi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w >> 32));
        sc.xlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x); sc.xhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32));
        sc.im = t.i0 - 1; sc.jm = t.j0 - 1;
        asm volatile("" : "+s"(sc.wlo), "+s"(sc.whi), "+s"(sc.xlo), "+s"(sc.xhi), "+s"(sc.im), "+s"(sc.jm));
        return sc;
    };
    auto issue = [&](const Src& sc, int stage, int s_lo, int s_hi) {
        uint32_t xl = lds_u32(smem_pk) + (uint32_t)(stage * STAGE);
        asm volatile("" : "+s"(xl));
        const void* wsrc = reinterpret_cast<const void*>((uint64_t)sc.whi << 32 | sc.wlo);
        const void* xsrc = reinterpret_cast<const void*>((uint64_t)sc.xhi << 32 | sc.xlo);
#pragma unroll
        for (int sl = s_lo; sl < s_hi; ++sl) {
            if (sl < NWS) {
                if ((sl + 1) * NW <= NWP || wave + sl * NW < NWP)
                    glds16_saddr<0>(wsrc, (uint32_t)lane * 16u + (uint32_t)((wave + sl * NW) * 1024), xl + (uint32_t)(XST + (wave + sl * NW) * 1024));
            } else {
                const int q = sl - NWS, pl = q & 3, r = q >> 2;
                constexpr int DQ = (NW * 64) / kUbPC, DR = (NW * 64) % kUbPC;
                uint32_t pk = pkv[pl];
                if (r > 0) asm volatile("" : "+v"(pk));     // (a fresh copy: derived entries hoisted out of the loop would cost the registers this saves)
                uint32_t prow = pk >> 8, pcol = pk & 255u;
#pragma unroll
                for (int i = 0; i < r; ++i) {
                    pcol += DR; prow += DQ;
                    const bool cy = pcol >= (uint32_t)kUbPC;
                    pcol = cy ? pcol - kUbPC : pcol; prow += cy ? 1u : 0u;
                }
                if (((vmask >> (pl * PR + r)) & 1u) && prow < (uint32_t)kUbPR) {
                    // padded input rows i0 - 1 + prow, clamped into the buffer: everything outside the image lands on the
                    // zero border, so positions outside the image come out as T = 0 (= the blur's padding) by themselves
                    const int gy = min(max(sc.im + (int)prow, 0), HP - 1), gx_ = min(max(sc.jm + (int)pcol, 0), WP - 1);
                    glds16_saddr<0>(xsrc, (uint32_t)(gy * WP + gx_) * 16u + (uint32_t)pl * plb, xl + (uint32_t)(rws[pl] + r * NW * 1024));
                }
            }
        }
    };

    Tile cur = decode(0);
    issue(src_of(cur, 0), 0, 0, NSLOT);
    for (int k = 0; k < my_tiles; ++k) {
        const bool more = k + 1 < my_tiles;
        Tile nxt = cur;
        if (more) nxt = decode(k + 1);
        // the vertical-pass role of this thread: row group vg (output rows RPW vg .. + RPW - 1 of the tile), output column 2 j0 - 2 + vx
        int l3 = lane;
        asm volatile("" : "+v"(l3));                 // (re-derived per tile: as kernel-lifetime constants these cost two more registers than the 2 x 16 form has)
        const int vx = l3 & (OX - 1), vg = wave * (64 / OX) + l3 / OX;
        const int ox = 2 * cur.j0 - 2 + vx, oy0 = 2 * cur.i0 + RPW * vg;
        const bool lane_ok = vx >= 2 && vx < OX - 2 && ox < R && vg < NVW;
        float nzv[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) nzv[rr] = 0.0f;
        if (a.noise && vg < NVW) {
            const float* __restrict__ nb = a.noise + (int64_t)(a.noise_batch > 1 ? cur.b : 0) * R * R + min(max(ox, 0), R - 1);
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) nzv[rr] = nb[(int64_t)min(oy0 + rr, R - 1) * R];
        }

        f32x16 acc[4][NPT];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[ph][pt] = zero16();
        for (int c = 0; c < a.n_chunks; ++c) {
            const int cs = c & 1;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            PK_T(0);
            const bool has_next = c + 1 < a.n_chunks;
            const Src src_nx = src_of(cur, c + 1);           // (past the last chunk: unused)
            const unsigned char* xb = smem_pk + cs * STAGE + (size_t)(half * 2) * XPLANE;
            const unsigned char* wb = smem_pk + cs * STAGE + XST + lane * 16;
            u32x4 bh[NPT][4], bl[NPT][4];       // shift s = 2 a + b: input rows i - 1 + a, columns j - 1 + b
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) {
                    const int pix = ((wave * NPT + pt) * SR + srow + (sft >> 1)) * kUbPC + col + (sft & 1);
                    bh[pt][sft] = *reinterpret_cast<const u32x4*>(xb + pix * 16);
                    bl[pt][sft] = *reinterpret_cast<const u32x4*>(xb + XPLANE + pix * 16);
                }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int sft = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1), ph = (ky & 1) * 2 + (kx & 1);
                const u32x4 ah = *reinterpret_cast<const u32x4*>(wb + (tap * 2 + 0) * 1024);
                const u32x4 al = *reinterpret_cast<const u32x4*>(wb + (tap * 2 + 1) * 1024);
                // product-major: consecutive MFMAs go to DIFFERENT accumulators.  Tile-major (three dependent MFMAs in a row) is
                // free while two waves share the SIMD, but a wave that has its SIMD to itself (<= 256 tiles: one workgroup per CU)
                // then runs at the dependent-accumulator latency -- measured 61 cycles per MFMA at the 64^2 level instead of 32.
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ph][pt] = mfma16(ah, bh[pt][sft], acc[ph][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ph][pt] = mfma16(al, bh[pt][sft], acc[ph][pt]);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[ph][pt] = mfma16(ah, bl[pt][sft], acc[ph][pt]);
                if (has_next && tap * PPT < NSLOT) issue(src_nx, cs ^ 1, tap * PPT, min((tap + 1) * PPT, NSLOT));
                if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
            PK_T(1);
        }
        // ---- epilogue: four rounds of eight output channels ----
        // (pin the noise values here: the compiler then waits for ITS loads now, while nothing else is in flight.  Left to itself it
        // puts s_waitcnt vmcnt(n) in front of their first use -- behind the prefetch pieces and the stores, which the in-order
        // counter would then drain in the middle of the vertical pass)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) asm volatile("" : "+v"(nzv[rr]));
        const float slope = a.slope;
        const f32x2 kx0 = f32x2{fx0, fx0}, kx1 = f32x2{fx1, fx1};
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // horizontal pass on the accumulators (registers only): output columns 2 j (h0) and 2 j + 1 (h1) of every T row
            f32x4 h0[NPT][2], h1[NPT][2];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {            // two channels per instruction (v_pk_fma_f32; the DPP moves stay per register)
                        const f32x2 p0 = f32x2{acc[2 * py][pt][4 * g4 + r], acc[2 * py][pt][4 * g4 + r + 1]};
                        const f32x2 p1 = f32x2{acc[2 * py + 1][pt][4 * g4 + r], acc[2 * py + 1][pt][4 * g4 + r + 1]};
                        const f32x2 p1m = f32x2{dpp_wave_shr1(p1[0]), dpp_wave_shr1(p1[1])};
                        const f32x2 p0p = f32x2{dpp_wave_shl1(p0[0]), dpp_wave_shl1(p0[1])};
                        const f32x2 p1p = f32x2{dpp_wave_shl1(p1[0]), dpp_wave_shl1(p1[1])};
                        f32x2 u = p1m * kx0;                    // T columns 2j-1, 2j, 2j+1, 2j+2
                        u = __builtin_elementwise_fma(p0, kx1, u); u = __builtin_elementwise_fma(p1, kx1, u); u = __builtin_elementwise_fma(p0p, kx0, u);
                        f32x2 v = p0 * kx0;                     // T columns 2j, 2j+1, 2j+2, 2j+3
                        v = __builtin_elementwise_fma(p1, kx1, v); v = __builtin_elementwise_fma(p0p, kx1, v); v = __builtin_elementwise_fma(p1p, kx0, v);
                        h0[pt][py][r] = u[0]; h0[pt][py][r + 1] = u[1];
                        h1[pt][py][r] = v[0]; h1[pt][py][r + 1] = v[1];
                    }
            __syncthreads();            // round 0: every wave has left the K loop (staging buffer 1 is free); later: the previous round's readers are done
            if (g4 == 0 && more) issue(src_of(nxt, 0), 0, 0, NSLOT);          // the next tile's first chunk lands in buffer 0 during this epilogue
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int trow = 2 * ((wave * NPT + pt) * SR + srow) + py;
                    f32x4* dst = reinterpret_cast<f32x4*>(hl + ((size_t)(half * TLR + trow) * OX + 2 * col) * 16);
                    dst[0] = h0[pt][py];
                    dst[1] = h1[pt][py];
                }
            __syncthreads();
            PK_T(2);
            if (vg < NVW) {
                const int gout = cur.cb * 4 + g4;
                // vertical pass: output row rr of the group needs T rows RPW vg + rr + 1 .. + 4 (tile row t <-> y = 2 i0 - 2 + t)
                auto load_row = [&](f32x2 (&w)[4], int t) {        // H row RPW vg + 1 + t of this thread's column: [channel pair]
                    const int trow = RPW * vg + 1 + t;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x4 q = *reinterpret_cast<const f32x4*>(hl + ((size_t)(hh * TLR + trow) * OX + vx) * 16);
                        w[2 * hh] = f32x2{q[0], q[1]};
                        w[2 * hh + 1] = f32x2{q[2], q[3]};
                    }
                };
                f32x2 win[4][4];                                   // sliding window of four H rows (a full RPW + 3 row block costs 24 more registers)
                load_row(win[0], 0); load_row(win[1], 1); load_row(win[2], 2);
                f32x2 bv[4];
#pragma unroll
                for (int cp = 0; cp < 4; ++cp) bv[cp] = *reinterpret_cast<const f32x2*>(bias_s + gout * 8 + 2 * cp);
                const f32x2 sl = f32x2{slope, slope}, km = f32x2{kmul, kmul};
                const f32x2 k0 = f32x2{fy0, fy0}, k1 = f32x2{fy1, fy1};
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const float nzr = a.noise ? __fmul_rn(nw, nzv[rr]) : 0.0f;
                    const f32x2 nz2 = f32x2{nzr, nzr};
                    u32x4 hi, lo;
                    float m = 0.0f;
                    load_row(win[(rr + 3) & 3], rr + 3);
#pragma unroll
                    for (int cp = 0; cp < 4; ++cp) {
                        f32x2 o = win[rr & 3][cp] * k0;
                        o = __builtin_elementwise_fma(win[(rr + 1) & 3][cp], k1, o);
                        o = __builtin_elementwise_fma(win[(rr + 2) & 3][cp], k1, o);
                        o = __builtin_elementwise_fma(win[(rr + 3) & 3][cp], k0, o);
                        f32x2 tv = (o + nz2) + bv[cp];                       // (conv + noise) + bias, as the unfused kernels round it
                        const f32x2 ls = tv * sl;
                        tv = f32x2{fmaxf(tv[0], ls[0]), fmaxf(tv[1], ls[1])} * km;     // lrelu (0 <= slope <= 1) * act_scale * 2^k
                        m = fmaxf(m, fmaxf(fabsf(tv[0]), fabsf(tv[1])));
                        SPLIT2_TO(tv[0], tv[1], hi[cp], lo[cp]);
                    }
                    const int oy = oy0 + rr;
                    if (lane_ok && oy < R) {
                        amax_l = fmaxf(amax_l, m);
                        u32x4* __restrict__ dst = reinterpret_cast<u32x4*>(a.y) + ((int64_t)(cur.b * GO + gout) * 2) * oplane + (int64_t)(oy + 1) * (R + 2) + ox + 1;
                        dst[0] = hi;
                        dst[oplane] = lo;
                    }
                }
            }
            PK_T(3);
        }
        cur = nxt;
    }
    if (a.out_amax) {
        int l2 = lane;
        asm volatile("" : "+v"(l2));          // (fresh permute addresses: sharing them with the prologue's amax_read keeps five registers alive across the kernel)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            amax_l = fmaxf(amax_l, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((l2 ^ off) << 2, __builtin_bit_cast(int, amax_l))));
        if (l2 == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * NW + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_l * kinv);
    }
    PK_T_DONE(NW);
}

// ---------------------------------------------------------------------------------------------------------------------
// Blur of an up-sampling layer + StyledConv's tail, T (fp32, zero-bordered) -> packed:  u = lrelu(upfirdn2d(T, k, pad (1, 1))
// + noise_w noise + bias) * act_scale  (stylesdf_model.py:346, :459-466, :500-507).  Workgroup = 8 channels (one packed entry
// group) x 16 rows x 64 columns; a thread finishes 4 adjacent pixels of all 8 channels and stores their hi / lo entries.
// Tap order per output (ky, then kx; one fma chain) is the one of e3dge_upfirdn2d.
// ---------------------------------------------------------------------------------------------------------------------
struct PkBlurK {
    const float* t; unsigned char* y; const float* fir; const float* noise; const float* noise_w; const float* noise_amax;
    const float* bias; const float* t_amax; int* out_meta; float* out_amax;
    float bias_amax, slope, act_scale;
    int B, C, R, noise_batch, tiles_x, tiles_y;
};
constexpr int kPbRows = 16, kPbCols = 64, kPbU = kPbRows + 3, kPbPitch = 68;

__global__ void __launch_bounds__(256) pk_blur_kernel(const PkBlurK a) {
    __shared__ __attribute__((aligned(16))) float u[8 * kPbU * kPbPitch];
    const int tid = threadIdx.x, lane = tid & 63;
    int bid = blockIdx.x;
    const int tx_i = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty_i = bid % a.tiles_y; bid /= a.tiles_y;
    const int G = a.C >> 3, g = bid % G, b = bid / G;
    const int R = a.R, TR = R + 3, TP = R + 4;
    const int oy0 = ty_i * kPbRows, ox0 = tx_i * kPbCols;

    const float nw = a.noise ? a.noise_w[0] : 0.0f;
    const float nza = a.noise ? fabsf(nw) * amax_read(a.noise_amax, lane) : 0.0f;
    const float bound = a.act_scale * (amax_read(a.t_amax, lane) * 1.001f + nza + a.bias_amax) * 1.001f;
    const unsigned eb = scale_exponent(bound);
    const float sc = pow2_bits(268u - eb);
    if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb;

    // stage 8 planes of 19 rows x 17 groups of four columns: T rows oy0 - 1 .., columns ox0 - 2 .. (one column more than the taps
    // need: with T(y, x) stored at [y + 1][x + 2] every group is one aligned 16-byte load; the T buffer carries its own zero border)
    constexpr int NGR = kPbPitch / 4, NE = 8 * kPbU * NGR, NIT = (NE + 255) / 256;
    const float* __restrict__ tb = a.t + ((int64_t)b * a.C + 8 * g) * TR * TP;
    f32x4 sv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = min(tid + it * 256, NE - 1);
        const int ch = idx / (kPbU * NGR), rem = idx - ch * (kPbU * NGR);
        const int r = rem / NGR, cg = rem - r * NGR;
        const int row = min(oy0 + r, TR - 1), cc = min(ox0 + 4 * cg, TP - 4);
        sv[it] = *reinterpret_cast<const f32x4*>(tb + ((int64_t)ch * TR + row) * TP + cc);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        if (idx < NE) *reinterpret_cast<f32x4*>(u + 4 * idx) = sv[it];       // u[ch][r][4 cg ..]: the same linear order
    }
    float kf[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) kf[p][q] = a.fir[(3 - p) * 4 + (3 - q)];
    __syncthreads();

    const int tx = tid & 15, ty = tid >> 4;
    const int oy = oy0 + ty, ox = ox0 + 4 * tx;
    const bool row_ok = oy < R;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.noise && row_ok) {
        const float* np_ = a.noise + (int64_t)(a.noise_batch > 1 ? b : 0) * R * R + (int64_t)oy * R + ox;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (ox + j < R) nz[j] = __fmul_rn(nw, np_[j]);
    }
    float amax_l = 0.0f;
    u32x4 hi[4], lo[4];                      // per pixel j: four words = eight channels
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {         // channel pairs (2 cp, 2 cp + 1) -> word cp of every pixel
        float v[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ch = 2 * cp + e;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const float* row = u + (ch * kPbU + ty + ky) * kPbPitch + 4 * tx;
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(row), q1 = *reinterpret_cast<const f32x4*>(row + 4);
                const float in[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
                for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(in[j + kx + 1], kf[ky][kx], acc[j]);     // u column c <-> x = ox0 - 2 + c
            }
            const float bv = a.bias ? a.bias[8 * g + ch] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[j];
                if (a.noise) t = __fadd_rn(t, nz[j]);
                t = t + bv;
                t = (t > 0.0f ? t : t * a.slope) * a.act_scale;
                if (row_ok && ox + j < R) amax_l = fmaxf(amax_l, fabsf(t));
                v[e][j] = t * sc;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) SPLIT2_TO(v[0][j], v[1][j], hi[j][cp], lo[j][cp]);
    }
    // Store through LDS: a thread holds four ADJACENT pixels, so its 16-byte stores would leave 48-byte gaps between lanes (every
    // store instruction touching 32 lines for 1 KiB).  The staging buffer is free now: entries go to LDS as [plane][row][pixel]
    // and come back in linear order -- each store instruction then writes 64 consecutive entries of one output row.
    __syncthreads();
    u32x4* const eb_lds = reinterpret_cast<u32x4*>(u);              // 2 planes x 16 rows x 64 pixels x 16 B = 32 KiB
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        eb_lds[(0 * kPbRows + ty) * kPbCols + 4 * tx + j] = hi[j];
        eb_lds[(1 * kPbRows + ty) * kPbCols + 4 * tx + j] = lo[j];
    }
    __syncthreads();
    {
        const int64_t plane = (int64_t)(R + 2) * (R + 2);
        u32x4* __restrict__ dst = reinterpret_cast<u32x4*>(a.y) + ((int64_t)(b * G + g) * 2) * plane;
#pragma unroll
        for (int it = 0; it < 2 * kPbRows * kPbCols / 256; ++it) {
            const int e = tid + it * 256;
            const int hl = e / (kPbRows * kPbCols), rem = e - hl * (kPbRows * kPbCols);
            const int r = rem / kPbCols, c = rem - r * kPbCols;
            if (oy0 + r < R && ox0 + c < R) dst[hl * plane + (int64_t)(oy0 + r + 1) * (R + 2) + ox0 + c + 1] = eb_lds[e];
        }
    }
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if (lane == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * 4 + (tid >> 6)) & (kAmaxSlots - 1)) * kAmaxStride, amax_l);
    }
}

// The same operation as a PERSISTENT, software-pipelined kernel (default; E3DGE_DEC2_BLUR=1 selects the one above).  The
// one-shot form is load -> barrier -> compute -> barrier -> store per workgroup with three workgroups per CU: at most ~40 KB per
// CU are in flight and it streamed 3.0-3.1 TB/s whatever the staging / store pattern (round-3 measurements).  Here a 512-thread
// workgroup walks 8-row tiles with tile k + 1 arriving by LDS-DMA (per-lane source address, 24 pieces of 1 KiB dealt out between
// the channel computations of tile k) and the entries leaving LDS-transposed (1 KiB per store instruction); 70 KB of LDS, so two
// workgroups share a CU and cover each other's waits (cold T rows return after ~3 k cycles under load: one tile of prefetch
// per workgroup is not enough on its own -- measured with E3DGE_PK_TIMING).  A thread finishes 4 adjacent pixels of 2 channels.
constexpr int kPb2Rows = 8, kPb2U = kPb2Rows + 3, kPb2Threads = 512;
constexpr int kPb2StageBytes = 8 * kPb2U * kPbPitch * 4, kPb2EntryBytes = 2 * kPb2Rows * kPbCols * 16;
constexpr int kPb2Lds = 2 * kPb2StageBytes + kPb2EntryBytes + 4096;        // + bias table (C <= 1024)

__global__ void __launch_bounds__(kPb2Threads) pk_blur2_kernel(const PkBlurK a, int n_tiles, int tiles_y2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pb[];
    unsigned char* const eb_lds = smem_pb + 2 * kPb2StageBytes;
    float* const bias_s = reinterpret_cast<float*>(eb_lds + kPb2EntryBytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.C >> 3, R = a.R, TR = R + 3, TP = R + 4;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    const float nw = a.noise ? a.noise_w[0] : 0.0f;
    const float nza = a.noise ? fabsf(nw) * amax_read(a.noise_amax, lane) : 0.0f;
    const float bound = a.act_scale * (amax_read(a.t_amax, lane) * 1.001f + nza + a.bias_amax) * 1.001f;
    const unsigned eb = scale_exponent(bound);
    const float kmul = a.act_scale * pow2_bits(268u - eb), kinv = 1.0f / pow2_bits(268u - eb);
    if (blockIdx.x == 0 && tid == 0) a.out_meta[0] = (int)eb;
    for (int i = tid; i < a.C; i += kPb2Threads) bias_s[i] = a.bias ? a.bias[i] : 0.0f;
    float kf[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) kf[p][q] = a.fir[(3 - p) * 4 + (3 - q)];

    struct Tile { int b, g, oy0, ox0; };
    auto tile_at = [&](int k) {
        int L = xcd_logical((int)blockIdx.x + k * (int)gridDim.x, n_tiles);
        Tile t;
        const int tx_i = L % a.tiles_x; L /= a.tiles_x;
        const int ty_i = L % tiles_y2; L /= tiles_y2;
        t.g = L % G; t.b = L / G;
        t.oy0 = ty_i * kPb2Rows; t.ox0 = tx_i * kPbCols;
        return t;
    };
    constexpr int NGR = kPbPitch / 4, NE = 8 * kPb2U * NGR, NPC = (NE + 63) / 64, NPW = (NPC + 7) / 8;   // 1496 groups of four floats, 24 pieces
    // this wave's pieces j_lo <= j < j_hi (piece wave + 8 j) of tile t; T rows oy0 - 1 .., columns ox0 - 2 .. (T(y, x) lives at
    // [y + 1][x + 2]: every group of four is one aligned 16-byte element; the T buffer carries its own zero border)
    auto issue = [&](const Tile& t, int buf, int j_lo, int j_hi) {
        const float* tb = a.t + ((int64_t)t.b * a.C + 8 * t.g) * TR * TP;
        const uint32_t dst = lds_u32(smem_pb + buf * kPb2StageBytes);
        for (int j = j_lo; j < j_hi; ++j) {
            const int pc = wave + 8 * j;
            if (pc >= NPC) break;
            const int e = pc * 64 + lane;
            if (e < NE) {
                const int ch = e / (kPb2U * NGR), rem = e - ch * (kPb2U * NGR);
                const int r = rem / NGR, cg = rem - r * NGR;
                const int row = min(t.oy0 + r, TR - 1), cc = min(t.ox0 + 4 * cg, TP - 4);
                dma_piece(tb, (uint32_t)((ch * TR + row) * TP + cc) * 4u, dst + pc * 1024);
            }
        }
    };
    const int cq = tid >> 7, t7 = tid & 127, tx = t7 & 15, ty = t7 >> 4;       // channel quarter (2 channels), 4-pixel group, row
    auto load_noise = [&](const Tile& t, float (&nz)[4]) {
        const int oy = min(t.oy0 + ty, R - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = min(t.ox0 + 4 * tx + j, R - 1);
            nz[j] = a.noise ? a.noise[(int64_t)(a.noise_batch > 1 ? t.b : 0) * R * R + (int64_t)oy * R + ox] : 0.0f;
        }
    };

    Tile t_cur = tile_at(0), t_nx = t_cur;
    float nz[4], nz_next[4] = {0.f, 0.f, 0.f, 0.f};
    issue(t_cur, 0, 0, NPW);
    load_noise(t_cur, nz_next);
    float amax_l = 0.0f;
    PK_T_INIT;
    const int nsteps = my_tiles;
    for (int k = 0; k < my_tiles; ++k) {
        // tile k (and its noise) has landed, tile k - 1's stores are out; after the barrier nobody reads the buffers reused below
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        PK_T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(nz_next[j])); nz[j] = __fmul_rn(nw, nz_next[j]); }
        const bool has_next = k + 1 < my_tiles;
        if (has_next) {
            t_nx = tile_at(k + 1);
            issue(t_nx, (k + 1) & 1, 0, 1);
        }
        PK_T(1);
        // ---- compute tile k: 4 pixels x 2 channels per thread ----
        const float* u = reinterpret_cast<const float*>(smem_pb + (k & 1) * kPb2StageBytes);
        const bool ok_row = t_cur.oy0 + ty < R;
        float v[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ch = 2 * cq + e;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const float* row = u + (ch * kPb2U + ty + ky) * kPbPitch + 4 * tx;
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(row), q1 = *reinterpret_cast<const f32x4*>(row + 4);
                const float in[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
                for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(in[j + kx + 1], kf[ky][kx], acc[j]);     // u column c <-> x = ox0 - 2 + c
            }
            const float bv = bias_s[8 * t_cur.g + ch];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float tv = acc[j];
                if (a.noise) tv = __fadd_rn(tv, nz[j]);
                tv = tv + bv;
                tv = fmaxf(tv, tv * a.slope) * kmul;                   // lrelu (0 <= slope <= 1) * act_scale * 2^k
                if (ok_row && t_cur.ox0 + 4 * tx + j < R) amax_l = fmaxf(amax_l, fabsf(tv));
                v[e][j] = tv;
            }
            if (has_next) issue(t_nx, (k + 1) & 1, 1 + e * (NPW / 2), e == 1 ? NPW : 1 + (NPW / 2));
        }
        if (has_next) load_noise(t_nx, nz_next);
        PK_T(2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                  // word cq of the pixel's hi and lo entries
            unsigned hw, lw;
            SPLIT2_TO(v[0][j], v[1][j], hw, lw);
            const int pix = ty * kPbCols + 4 * tx + j;
            *reinterpret_cast<unsigned*>(eb_lds + (size_t)pix * 16 + 4 * cq) = hw;
            *reinterpret_cast<unsigned*>(eb_lds + (size_t)(kPb2Rows * kPbCols + pix) * 16 + 4 * cq) = lw;
        }
        __syncthreads();
        {   // entries [plane][row][pixel] come back in linear order: 1 KiB per store instruction
            const int64_t plane = (int64_t)(R + 2) * (R + 2);
            u32x4* __restrict__ dst = reinterpret_cast<u32x4*>(a.y) + ((int64_t)(t_cur.b * G + t_cur.g) * 2) * plane;
#pragma unroll
            for (int it = 0; it < 2 * kPb2Rows * kPbCols / kPb2Threads; ++it) {
                const int e = tid + it * kPb2Threads;
                const int hl = e / (kPb2Rows * kPbCols), rem = e - hl * (kPb2Rows * kPbCols);
                const int r = rem / kPbCols, c = rem - r * kPbCols;
                if (t_cur.oy0 + r < R && t_cur.ox0 + c < R)
                    dst[hl * plane + (int64_t)(t_cur.oy0 + r + 1) * (R + 2) + t_cur.ox0 + c + 1] = reinterpret_cast<const u32x4*>(eb_lds)[e];
            }
        }
        PK_T(3);
        t_cur = t_nx;
    }
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, off, kWave));
        if (lane == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * 8 + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_l * kinv);
    }
    PK_T_DONE(8);
}

// ---------------------------------------------------------------------------------------------------------------------
// ToRGB on a packed activation (stylesdf_model.py:531-541): 1x1 modulated conv without demodulation (table wm = (scale W) s
// from the weights launch) + bias + the FIR-up-sampled skip image.  Bound: HBM (4 B per input element).  A thread owns one
// pixel and a slice of the channel groups (16-byte entries, coalesced across pixels); slices fold through LDS in order.
// ---------------------------------------------------------------------------------------------------------------------
template <int GS>
__global__ void __launch_bounds__(256)
pk_torgb_kernel(float* __restrict__ y, const unsigned char* __restrict__ x, const int* __restrict__ meta, const float* __restrict__ wm_g,
                const float* __restrict__ bias, const float* __restrict__ skip, const float* __restrict__ fir, int Ci, int R,
                int blocks_per_img) {
    constexpr int PL = 256 / GS;
    __shared__ float wm[3 * 1024];
    __shared__ float part[GS > 1 ? GS - 1 : 1][3][PL];
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x - b * blocks_per_img;
    const int pl = threadIdx.x % PL, grp = threadIdx.x / PL;
    for (int i = threadIdx.x; i < 3 * Ci; i += 256) wm[i] = wm_g[(size_t)b * 3 * Ci + i];
    __syncthreads();
    const float inv = pow2_bits((unsigned)meta[0] - 14u);
    const int p = blk * PL + pl, HW = R * R;
    const bool live = p < HW;
    const int oy = live ? p / R : 0, ox = live ? p - oy * R : 0;
    float acc[3] = {0.f, 0.f, 0.f};
    if (live) {
        const int G = Ci >> 3, per = (G + GS - 1) / GS, g0 = grp * per, g1 = min(G, g0 + per);
        const int64_t plane = (int64_t)(R + 2) * (R + 2);
        const u32x4* __restrict__ xp = reinterpret_cast<const u32x4*>(x) + ((int64_t)(b * G + g0) * 2) * plane + (int64_t)(oy + 1) * (R + 2) + ox + 1;
        auto fold = [&](int gi, const u32x4& h, const u32x4& l) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float v0 = f16lo(h[w]) + f16lo(l[w]), v1 = f16hi(h[w]) + f16hi(l[w]);
                const int ci = 8 * gi + 2 * w;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = fmaf(wm[c * Ci + ci + 1], v1, fmaf(wm[c * Ci + ci], v0, acc[c]));
            }
        };
        int gi = g0;
        for (; gi + 4 <= g1; gi += 4) {
            u32x4 h[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { h[q] = xp[(int64_t)(2 * q) * plane]; l[q] = xp[(int64_t)(2 * q + 1) * plane]; }
            xp += 8 * plane;
#pragma unroll
            for (int q = 0; q < 4; ++q) fold(gi + q, h[q], l[q]);
        }
        for (; gi < g1; ++gi) {
            const u32x4 h = xp[0], l = xp[plane];
            xp += 2 * plane;
            fold(gi, h, l);
        }
    }
    if (GS > 1) {
        if (grp > 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) part[grp - 1][c][pl] = acc[c];
        }
        __syncthreads();
    }
    if (grp != 0 || !live) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float conv = acc[c];
        if (GS > 1) {
#pragma unroll
            for (int q = 0; q < GS - 1; ++q) conv += part[q][c][pl];
        }
        float o = conv * inv + bias[c];
        if (skip) {      // upfirdn2d(skip, fir, up=2, pad=(2,1)): same taps, same order as e3dge_torgb / e3dge_upfirdn2d
            const int h = R >> 1, w = R >> 1;
            const float* sp = skip + ((int64_t)b * 3 + c) * h * w;
            float uacc = 0.0f;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int ky = (oy & 1) + 2 * p2, iy = (oy + ky - 2) >> 1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int kx = (ox & 1) + 2 * e, ix = (ox + kx - 2) >> 1;
                    if (iy >= 0 && iy < h && ix >= 0 && ix < w) uacc = fmaf(sp[iy * w + ix], fir[(3 - ky) * 4 + (3 - kx)], uacc);
                }
            }
            o = o + uacc;
        }
        y[((int64_t)b * 3 + c) * HW + p] = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
template <int NCT, int NPY, int NPX, int WCO, int WY, int WX, int RGB = 0, int BWD = 0>
static int launch_s1(PkConvK k, hipStream_t st, const char* what) {
    constexpr int TH = NPY * WY, TW = 32 * NPX * WX, NPIX = (TH + 2) * (TW + 2), NCTB = NCT * WCO;
    constexpr int lds_stages = 2 * (4 * NPIX * 16 + NCTB * kPkSlab);
    static_assert(lds_stages <= 160 * 1024, "LDS budget");
    const int lds = lds_stages + 4 * k.Co * ((RGB || BWD == 1) ? 4 : 1);         // + bias table (+ ToRGB table)
    E3DGE_REQUIRE(lds <= 160 * 1024, "%s: LDS budget with Co=%d", what, k.Co);
    E3DGE_REQUIRE(k.Co % (32 * NCTB) == 0, "%s: Co=%d not a multiple of %d", what, k.Co, 32 * NCTB);
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.W + TW - 1) / TW;
    k.co_blocks = k.Co / (32 * NCTB);
    E3DGE_REQUIRE(!RGB || (k.co_blocks == 1 && k.n_chunks >= 2 && k.rgb_wm && k.rgb_bias && k.rgb_out), "%s: fused ToRGB needs one co-block and >= 32 input channels", what);
    E3DGE_REQUIRE(RGB != 2 || (k.y && k.out_meta), "%s: the store + ToRGB form needs the output activation", what);
    E3DGE_REQUIRE(BWD != 1 || (k.y && k.out_meta && k.mask_act && k.bwd_wl1 && k.in_amax && k.n_chunks >= 2), "%s: data-gradient launch: missing pointer, or fewer than 32 input channels", what);
    E3DGE_REQUIRE(BWD != 1 || !k.rgbt_d || (k.rgb_wm && k.rgbt_amax && k.rgbt_l1 && k.n_chunks >= 2), "%s: ToRGB^T needs its table, amax, norm and >= 32 input channels", what);
    E3DGE_REQUIRE(BWD != 2 || k.out_f32, "%s: the last data-gradient launch needs out_f32", what);
    const int64_t n_tiles = (int64_t)k.B * k.co_blocks * k.tiles_y * k.tiles_x;
    E3DGE_REQUIRE(n_tiles < ((int64_t)1 << 30), "%s: too many tiles", what);
    k.n_tiles = (int)n_tiles;
    auto fn = &pkconv_s1_kernel<NCT, NPY, NPX, WCO, WY, WX, RGB, BWD>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    int grid = 256 * ((160 * 1024) / lds >= 2 ? 2 : 1);
    if (grid > k.n_tiles) grid = k.n_tiles;
    fn<<<dim3((unsigned)grid), dim3(64 * WCO * WY * WX), lds, st>>>(k);
    return check_launch(what);
}

constexpr int kUpNpixMax = 528;
template <int NCT, int NPT, int WCO, int WQ>
static int launch_up(PkConvK k, hipStream_t st, const char* what) {
    constexpr int Q = 32 * NPT * WQ, NCTB = NCT * WCO;
    constexpr int lds = 2 * (4 * kUpNpixMax * 16 + NCTB * kPkSlab);
    static_assert(lds <= 160 * 1024, "LDS budget");
    E3DGE_REQUIRE(k.Co % (32 * NCTB) == 0, "%s: Co=%d not a multiple of %d", what, k.Co, 32 * NCTB);
    k.co_blocks = k.Co / (32 * NCTB);
    // column blocks: the fewest equal-width blocks whose patch (rows spanned by Q consecutive positions + 1, block width + 1
    // columns) fits the LDS plane.  Q = 256: 65 -> one block, 129 -> one, 257 -> 129 + 128, 513 -> 4 x 103 + 101.
    auto fits = [&](int wdt) {
        int rows = (Q - 1) / wdt + 3;
        if (rows > k.H + 2) rows = k.H + 2;
        return rows * (wdt + 1) <= kUpNpixMax;
    };
    k.nblk = 0;
    for (int nb = 1; nb <= k.W + 1 && nb <= 256; ++nb) {
        const int cw = (k.W + 1 + nb - 1) / nb, cwl = k.W + 1 - (nb - 1) * cw;
        if (cwl >= 1 && fits(cw) && fits(cwl)) { k.nblk = nb; k.cw = cw; k.cwl = cwl; break; }
    }
    E3DGE_REQUIRE(k.nblk > 0, "%s: no column blocking of %d positions fits the LDS plane", what, k.W + 1);
    k.tpf = ((k.H + 1) * k.cw + Q - 1) / Q;
    k.tpl = ((k.H + 1) * k.cwl + Q - 1) / Q;
    const int64_t n_tiles = (int64_t)k.B * k.co_blocks * ((int64_t)(k.nblk - 1) * k.tpf + k.tpl);
    E3DGE_REQUIRE(n_tiles < ((int64_t)1 << 30), "%s: too many tiles", what);
    k.n_tiles = (int)n_tiles;
    auto fn = &pkconv_up_kernel<NCT, NPT, WCO, WQ, kUpNpixMax>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    int grid = 256 * ((160 * 1024) / lds >= 2 ? 2 : 1);
    if (grid > k.n_tiles) grid = k.n_tiles;
    fn<<<dim3((unsigned)grid), dim3(64 * WCO * WQ), lds, st>>>(k);
    return check_launch(what);
}

template <int NPT>
static int launch_upblur_t(PkConvK k, const float* fir, hipStream_t st) {
    constexpr int TH = 8 * NPT - 2, lds = 2 * (4 * (TH + 3) * kUbPC * 16 + kPkSlab) + 4096;
    static_assert(lds <= 160 * 1024 && 8 * 16 * NPT * 64 * 4 <= lds - 4096, "LDS budget");
    k.co_blocks = k.Co / 32;
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.W + kUbTW - 1) / kUbTW;
    const int64_t n_tiles = (int64_t)k.B * k.co_blocks * k.tiles_y * k.tiles_x;
    E3DGE_REQUIRE(n_tiles < ((int64_t)1 << 30), "dec2 convT+blur: too many tiles");
    k.n_tiles = (int)n_tiles;
    auto fn = &pkconv_upblur_kernel<NPT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(dec2 convT+blur): %s", hipGetErrorString(e));
    const int wgs = (160 * 1024) / lds >= 2 ? 512 : 256;
    const int grid = k.n_tiles < wgs ? k.n_tiles : wgs;
    fn<<<dim3((unsigned)grid), dim3(512), lds, st>>>(k, fir);
    return check_launch("dec2 convT+blur");
}
static int shape_override(const char* name);
template <int NPT, int NW, int TC>
static int launch_upblur2_t(PkConvK k, const float* g, hipStream_t st) {
    constexpr int PRW = NW * NPT * (32 / TC), TH = PRW - 2, TW = TC - 2;
    constexpr int STAGE = 4 * (PRW + 1) * (TC + 1) * 16 + kPkSlab, HB = 2 * 2 * PRW * 2 * TC * 16;
    constexpr int lds = (2 * STAGE > STAGE + HB ? 2 * STAGE : STAGE + HB) + 4096;
    static_assert(lds <= 160 * 1024, "LDS budget");
    k.co_blocks = k.Co / 32;
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.W + TW - 1) / TW;
    const int64_t n_tiles = (int64_t)k.B * k.co_blocks * k.tiles_y * k.tiles_x;
    E3DGE_REQUIRE(n_tiles < ((int64_t)1 << 30), "dec2 convT+blur: too many tiles");
    k.n_tiles = (int)n_tiles;
    auto fn = &pkconv_upblur2_kernel<NPT, NW, TC>;
    const int wgs = (NW == 4 && (160 * 1024) / lds >= 2) ? 512 : 256;      // (eight-wave forms: 143+ registers, one workgroup per CU)
    const int grid = k.n_tiles < wgs ? k.n_tiles : wgs;
    // Two 77-KB workgroups fit a CU and the dispatcher PACKS them (measured: 264 tiles of the 64^2 level ran two per CU on 132 CUs,
    // each at half the matrix-pipe rate, while 124 CUs idled).  With at most one tile per CU to hand out, ask for more than half
    // of the LDS so that every workgroup gets a CU of its own.
    const int lds_req = ((160 * 1024) / lds >= 2 && grid <= 256 && shape_override("E3DGE_DEC2_UPBLUR_SPREAD") != 0) ? 81 * 1024 : lds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_req);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(dec2 convT+blur v2): %s", hipGetErrorString(e));
    fn<<<dim3((unsigned)grid), dim3(64 * NW), lds_req, st>>>(k, g[0], g[1]);
    return check_launch("dec2 convT+blur v2");
}
// how many rounds of `slots` concurrently resident workgroups a tiling needs (ceil), per (stored rows, stored columns) of a tile
static int upblur_rounds(const PkConvK& k, int th, int tw, int slots) {
    const int64_t n = (int64_t)k.B * (k.Co / 32) * ((k.H + th - 1) / th) * ((k.W + tw - 1) / tw);
    return (int)((n + slots - 1) / slots);
}
static int shape_override(const char* name);
// fir1d: the 1-D factor of the blur kernel when the plan says it is rank one (second-generation kernel), else NULL
static int launch_upblur(PkConvK k, const float* fir, const float* fir1d, hipStream_t st) {
    E3DGE_REQUIRE(k.Co % 32 == 0 && k.Co <= 1024 && k.y && k.out_meta, "dec2 convT+blur: bad arguments");
    if (fir1d && shape_override("E3DGE_DEC2_UPBLUR_GEN") != 1) {
        // tile shapes: 0 = eight waves x 2 position rows (14 x 30 stored positions, one workgroup per CU), 1 = eight waves x 1 row
        // (6 x 30), 2 = FOUR waves x 2 rows (6 x 30, 77 KB of LDS: two workgroups per CU, whose barriers and operand fetches
        // interleave), 3 = four waves x 2 MFMA tiles of 2 x 16 positions (14 x 14 stored, same LDS).  Automatic: 2 or 3, whichever
        // needs fewer rounds of 512 resident workgroups (64^2: 200 instead of 264 tiles -- one per CU; 512^2: 1,369 instead of
        // 1,548 -- three rounds instead of four), 3 on a tie (9 % fewer halo positions).
        // 4 = EIGHT waves x 1 MFMA tile of 2 x 16 (the same 14 x 14 tile, one workgroup per CU): when there are no more tiles than
        // CUs a four-wave workgroup has its SIMDs to itself and its LDS-DMA issue (10 pieces per wave and chunk, ~150 cycles each)
        // is no longer covered by a partner's MFMAs -- 61 cycles per MFMA measured at the 64^2 level; with two waves of the SAME
        // workgroup per SIMD it is (59.6 vs 70.0 us there, same box).
        int v = shape_override("E3DGE_DEC2_UPBLUR_SHAPE");
        if (v < 0) {
            if (upblur_rounds(k, 14, 14, 256) <= 1) v = 4;
            else v = upblur_rounds(k, 14, 14, 512) <= upblur_rounds(k, 6, 30, 512) ? 3 : 2;
        }
        switch (v) {
            case 0: return launch_upblur2_t<2, 8, 32>(k, fir1d, st);
            case 1: return launch_upblur2_t<1, 8, 32>(k, fir1d, st);
            case 2: return launch_upblur2_t<2, 4, 32>(k, fir1d, st);
            case 4: return launch_upblur2_t<1, 8, 16>(k, fir1d, st);
            default: return launch_upblur2_t<2, 4, 16>(k, fir1d, st);
        }
    }
    // tiles of 14 x 30 positions (two position rows per wave).  The 6 x 30 form (E3DGE_DEC2_UPBLUR_NPT=1) gives the 64^2 level 264
    // tiles instead of 120 for the 256 CUs but re-streams every weight slab twice as often: 115 vs 96 us there, slower everywhere.
    return shape_override("E3DGE_DEC2_UPBLUR_NPT") == 1 ? launch_upblur_t<1>(k, fir, st) : launch_upblur_t<2>(k, fir, st);
}

// fuse the blur into the transposed convolution?  Measured at every level of the 1024^2 / channel-multiplier-2 decoder
// (tools/dec2_check.py, one MI355X): 93 vs 75 + 19 us (64 -> 128), 68 vs 63 + 28, 90 vs 71 + 46, 108 vs 70 + 84 (512 -> 1024):
// never slower, so it is the default; E3DGE_DEC2_UPBLUR=0 keeps the two-kernel form (T in HBM) for A/B and tests.
static bool use_upblur(int) {
    const char* v = getenv("E3DGE_DEC2_UPBLUR");
    return !(v && *v && atoi(v) == 0);
}

static int shape_override(const char* name) {      // E3DGE_DEC2_S1 / E3DGE_DEC2_UP = variant index (tuning runs); -1: automatic
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : -1;
}

// can the last convolution also do ToRGB?  (one co-block, every channel of a pixel in one wave: the 32- and 64-channel tiles)
static bool s1_can_fuse_rgb(int co, int ci) {
    const char* v = getenv("E3DGE_DEC2_FUSE_RGB");
    if (v && *v && atoi(v) == 0) return false;
    return (co == 32 || co == 64) && ci >= 32;
}

static int conv_s1(PkConvK k, hipStream_t st) {
    if (k.rgb_out && k.rgb_store) {
        if (k.Co == 32) return launch_s1<1, 1, 2, 1, 8, 1, 2>(k, st, "dec2 conv+store+rgb<32co,8x64>");
        return launch_s1<2, 1, 2, 1, 8, 1, 2>(k, st, "dec2 conv+store+rgb<64co,8x64>");
    }
    if (k.rgb_out) {
        if (k.Co == 32) return launch_s1<1, 1, 2, 1, 8, 1, 1>(k, st, "dec2 conv+rgb<32co,8x64>");
        return launch_s1<2, 1, 2, 1, 8, 1, 1>(k, st, "dec2 conv+rgb<64co,8x64>");
    }
    int v = shape_override("E3DGE_DEC2_S1");
    const int64_t px = (int64_t)k.H * k.W;
    if (v < 0) {
        if (k.Co % 64 != 0) v = 3;
        else if (px <= 64 * 64) v = 0;
        else if (px <= 128 * 128) v = 1;
        else v = 2;
    }
    if (k.Co % 64 != 0 && v != 3 && v != 4) v = 3;
    switch (v) {
        case 0: return launch_s1<1, 1, 1, 2, 4, 1>(k, st, "dec2 conv<64co,4x32>");
        case 1: return launch_s1<1, 1, 2, 2, 4, 1>(k, st, "dec2 conv<64co,4x64>");
        case 2: return launch_s1<2, 1, 2, 1, 8, 1>(k, st, "dec2 conv<64co,8x64>");
        case 3: return launch_s1<1, 1, 2, 1, 8, 1>(k, st, "dec2 conv<32co,8x64>");
        default: return launch_s1<1, 2, 2, 1, 4, 1>(k, st, "dec2 conv<32co,8x64,4w>");
    }
}

static int conv_up(PkConvK k, hipStream_t st) {
    int v = shape_override("E3DGE_DEC2_UP");
    if (v < 0) {      // measured per layer (tools/dec2_check.py --sweep, 1024^2 / channel multiplier 2)
        if (k.Co % 64 != 0) v = 3;
        else if ((int64_t)k.H * k.W <= 64 * 64) v = 3;
        else v = 1;
    }
    if (k.Co % 64 != 0 && v != 2 && v != 3) v = 2;
    switch (v) {
        case 0: return launch_up<1, 1, 2, 4>(k, st, "dec2 convT<64co,128q>");
        case 1: return launch_up<2, 1, 1, 8>(k, st, "dec2 convT<64co,256q>");
        case 2: return launch_up<1, 2, 1, 4>(k, st, "dec2 convT<32co,256q,4w>");
        default: return launch_up<1, 1, 1, 8>(k, st, "dec2 convT<32co,256q>");
    }
}

static int launch_torgb(float* y, const unsigned char* x, const int* meta, const float* wm, const float* bias, const float* skip,
                        const float* fir, int B, int Ci, int R, hipStream_t st) {
    E3DGE_REQUIRE(Ci % 8 == 0 && Ci <= 1024, "dec2 torgb: ci=%d", Ci);
    const int64_t hw = (int64_t)R * R;
    const int gs = Ci >= 512 ? 16 : (Ci >= 256 ? 8 : (Ci >= 128 ? 4 : (Ci >= 64 ? 2 : 1)));
    const int pl = 256 / gs, bpi = (int)((hw + pl - 1) / pl);
    dim3 grid((unsigned)(bpi * B)), th(256);
    switch (gs) {
        case 16: pk_torgb_kernel<16><<<grid, th, 0, st>>>(y, x, meta, wm, bias, skip, fir, Ci, R, bpi); break;
        case 8: pk_torgb_kernel<8><<<grid, th, 0, st>>>(y, x, meta, wm, bias, skip, fir, Ci, R, bpi); break;
        case 4: pk_torgb_kernel<4><<<grid, th, 0, st>>>(y, x, meta, wm, bias, skip, fir, Ci, R, bpi); break;
        case 2: pk_torgb_kernel<2><<<grid, th, 0, st>>>(y, x, meta, wm, bias, skip, fir, Ci, R, bpi); break;
        default: pk_torgb_kernel<1><<<grid, th, 0, st>>>(y, x, meta, wm, bias, skip, fir, Ci, R, bpi); break;
    }
    return check_launch("dec2 torgb");
}

#include "decoder2_bwd.h"

static int check_conv(const E3dgeDec2Conv& c, const char* what) {
    E3DGE_REQUIRE(c.wpre && c.style && c.demod && c.wimg && c.bias, "dec2 %s: null pointer", what);
    E3DGE_REQUIRE(c.ci > 0 && c.co > 0 && c.ci % 16 == 0 && c.co % 32 == 0 && c.ci <= 1024, "dec2 %s: needs ci %% 16 == 0, co %% 32 == 0 (got %d, %d)", what, c.ci, c.co);
    E3DGE_REQUIRE(c.noise == nullptr || (c.noise_w && c.noise_amax && c.noise_batch >= 1), "dec2 %s: noise needs noise_w, noise_amax, noise_batch", what);
    return E3DGE_OK;
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int64_t e3dge_dec2_act_words(int batch, int channels, int res) {
    if (batch <= 0 || channels <= 0 || res <= 0) return 0;
    return (int64_t)batch * ((channels + 7) / 8) * 2 * (res + 2) * (res + 2) * 4;
}
extern "C" int64_t e3dge_dec2_tbuf_floats(int batch, int co, int in_res) {
    if (batch <= 0 || co <= 0 || in_res <= 0) return 0;
    return (int64_t)batch * co * (2 * in_res + 3) * (2 * in_res + 4);
}
extern "C" int e3dge_dec2_num_launches(int n_up) { return 6 + 4 * n_up; }

extern "C" int e3dge_dec2_prepack_weights(float* wpre, const float* weight, float scale, int co, int ci, e3dge_stream_t stream) {
    E3DGE_REQUIRE(wpre && weight && co > 0 && ci > 0, "dec2_prepack_weights: bad arguments");
    E3DGE_REQUIRE(co % 32 == 0 && ci % 16 == 0, "dec2_prepack_weights: needs co %% 32 == 0 and ci %% 16 == 0");
    const int64_t n = (int64_t)co * ci * 9;
    pk_prepack_kernel<<<dim3(512), dim3(256), 0, as_stream(stream)>>>(wpre, weight, scale, co, ci, ci / 16, n);
    return check_launch("dec2_prepack_weights");
}

extern "C" int e3dge_dec2_pack(uint32_t* packed, int32_t* meta, const float* x, const float* amax, int batch, int channels, int res,
                               e3dge_stream_t stream) {
    E3DGE_REQUIRE(packed && meta && x && amax && batch >= 0 && channels > 0 && channels % 8 == 0 && res > 0, "dec2_pack: bad arguments");
    if (batch == 0) return E3DGE_OK;
    pk_pack_kernel<<<dim3((unsigned)((res * res + 255) / 256), (unsigned)(channels / 8), (unsigned)batch), dim3(256), 0, as_stream(stream)>>>(packed, meta, x, amax, channels, res);
    return check_launch("dec2_pack");
}
extern "C" int e3dge_dec2_unpack(float* x, const uint32_t* packed, const int32_t* meta, int batch, int channels, int res,
                                 e3dge_stream_t stream) {
    E3DGE_REQUIRE(packed && meta && x && batch >= 0 && channels > 0 && channels % 8 == 0 && res > 0, "dec2_unpack: bad arguments");
    if (batch == 0) return E3DGE_OK;
    pk_unpack_kernel<<<dim3((unsigned)((res * res + 255) / 256), (unsigned)(channels / 8), (unsigned)batch), dim3(256), 0, as_stream(stream)>>>(x, packed, meta, channels, res);
    return check_launch("dec2_unpack");
}

extern "C" int e3dge_dec2_forward(const E3dgeDec2Plan* P, e3dge_stream_t stream) {
    E3DGE_REQUIRE(P != nullptr, "dec2_forward: null plan");
    E3DGE_REQUIRE(P->batch >= 0 && P->n_up >= 0 && P->n_up <= E3DGE_DEC2_MAX_UP && P->in_res >= 4 && P->in_ch > 0 && P->in_ch % 16 == 0,
                  "dec2_forward: bad sizes (batch %d, n_up %d, in_res %d, in_ch %d)", P->batch, P->n_up, P->in_res, P->in_ch);
    if (P->batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(P->features && P->mod_table && P->latent && P->amax && P->meta && P->fir_blur && P->fir_up, "dec2_forward: null pointer");
    int rc = check_conv(P->conv1, "conv1");
    if (rc) return rc;
    E3DGE_REQUIRE(P->conv1.ci == P->in_ch, "dec2_forward: conv1.ci != in_ch");
    E3DGE_REQUIRE(P->skip_in == nullptr, "dec2_forward: rgbd_in is not supported by the packed pipeline");
    E3DGE_REQUIRE(P->rgb1.ci == P->conv1.co, "dec2_forward: rgb1.ci != conv1.co");
    for (int u = 0, prev = P->conv1.co; u < P->n_up; ++u) {
        if ((rc = check_conv(P->up[u], "up")) != 0 || (rc = check_conv(P->conv[u], "conv")) != 0) return rc;
        E3DGE_REQUIRE(P->tbuf[u] && P->act[2 + 2 * u] && P->act[3 + 2 * u] && P->rgb[u].out && P->rgb[u].wm && P->rgb[u].weight && P->rgb[u].style && P->rgb[u].bias,
                      "dec2_forward: level %d workspace / ToRGB pointer missing", u);
        E3DGE_REQUIRE(P->up[u].ci == prev && P->conv[u].ci == P->up[u].co && P->rgb[u].ci == P->conv[u].co, "dec2_forward: level %d channel chain", u);
        E3DGE_REQUIRE((int64_t)(1 + (P->up[u].co + 7) / 8) * 8 * ((int64_t)P->in_res << (u + 1)) * ((int64_t)P->in_res << (u + 1)) * P->batch < ((int64_t)1 << 31),
                      "dec2_forward: level %d activation too large for 32-bit offsets", u);
        prev = P->conv[u].co;
    }
    E3DGE_REQUIRE(P->act[0] && P->act[1] && P->rgb1.out && P->rgb1.wm && P->rgb1.weight && P->rgb1.style && P->rgb1.bias, "dec2_forward: workspace missing");
    hipStream_t st = as_stream(stream);
    const int B = P->batch, n_l = e3dge_dec2_num_launches(P->n_up);
    const bool timing = P->kernel_ms != nullptr;
    E3DGE_REQUIRE(!timing || P->n_kernel_ms >= n_l, "dec2_forward: kernel_ms needs %d entries", n_l);
    hipEvent_t ev[8 + 4 * E3DGE_DEC2_MAX_UP];
    bool fused_away[8 + 4 * E3DGE_DEC2_MAX_UP] = {};
    int n_ev = 0;
    bool ev_failed = false;
    auto mark = [&](bool fused = false) {
        if (!timing || ev_failed) return;
        if (hipEventCreate(&ev[n_ev]) != hipSuccess) { ev_failed = true; return; }
        fused_away[n_ev] = fused;
        if (hipEventRecord(ev[n_ev++], st) != hipSuccess) ev_failed = true;
    };
    auto finish = [&](int code) {
        if (timing) {
            if (code == 0 && ev_failed) code = fail(E3DGE_ERR_LAUNCH, "dec2_forward: HIP event create / record failed (kernel_ms)");
            if (code == 0 && n_ev > 0) {
                if (hipEventSynchronize(ev[n_ev - 1]) != hipSuccess) code = fail(E3DGE_ERR_LAUNCH, "dec2: hipEventSynchronize failed (kernel_ms)");
                for (int i = 0; i + 1 < n_ev; ++i) {
                    if (hipEventElapsedTime(&P->kernel_ms[i], ev[i], ev[i + 1]) != hipSuccess) P->kernel_ms[i] = -1.0f;
                    if (fused_away[i + 1]) P->kernel_ms[i] = 0.0f;        // this launch does not exist: its work is part of the previous one
                }
            }
            for (int i = 0; i < n_ev; ++i) (void)hipEventDestroy(ev[i]);
        }
        return code;
    };
#define DEC2_STEP(expr) do { rc = (expr); if (rc) return finish(rc); mark(); } while (0)

    // The amax block is zeroed by a KERNEL (the styles launch, the first of a forward), not by hipMemsetAsync: captured into a HIP graph,
    // the memset node was observed to run out of order with the kernels behind it (round 4: replays of the inversion forward came back
    // with a handful of discrete wrong images -- the amax of the packed features or of a later activation zeroed after its producer had
    // written it; eager launches and most replays were fine, tools/graph_debug.py).  A kernel node keeps the stream order.
    mark();
    // 1. all modulation vectors + demodulation factors (+ the zeroing)
    DEC2_STEP(decoder_styles_launch(P->mod_table, P->n_mod, P->mod_rows, P->mod_co, P->latent, P->n_latent, P->style_dim, B, P->amax,
                                    E3DGE_AMAX_FLOATS * (3 * P->n_up + 2), st));
    // 2, 3. features -> packed
    const float* amax0 = P->amax;
    DEC2_STEP(e3dge_amax(P->amax, P->features, (int64_t)B * P->in_ch * P->in_res * P->in_res, stream));
    DEC2_STEP(e3dge_dec2_pack(P->act[0], P->meta, P->features, amax0, B, P->in_ch, P->in_res, stream));
    // 4. per-sample weight images + ToRGB tables
    {
        PkWTab tab{};
        auto add_conv = [&](const E3dgeDec2Conv& c) {
            PkWConv& w = tab.conv[tab.n_conv++];
            w.wpre = c.wpre; w.style = c.style; w.demod = c.demod; w.img = c.wimg; w.co = c.co; w.ci = c.ci; w.n_chunks = c.ci / 16;
            w.n_items = (c.co / 32) * (c.ci / 16) * 9 * 64;
            w.words = e3dge_modconv_packed_words(c.co, c.ci);
        };
        auto add_rgb = [&](const E3dgeDec2Rgb& r) {
            PkWRgb& w = tab.rgb[tab.n_rgb++];
            w.w = r.weight; w.style = r.style; w.wm = r.wm; w.scale = r.scale; w.ci = r.ci;
        };
        add_conv(P->conv1);
        for (int u = 0; u < P->n_up; ++u) { add_conv(P->up[u]); add_conv(P->conv[u]); }
        add_rgb(P->rgb1);
        for (int u = 0; u < P->n_up; ++u) add_rgb(P->rgb[u]);
        pk_weights_kernel<<<dim3(96, (unsigned)(tab.n_conv + tab.n_rgb), (unsigned)B), dim3(256), 0, st>>>(tab);
        DEC2_STEP(check_launch("dec2 weights"));
    }
    auto conv_args = [&](const E3dgeDec2Conv& c, int res) {
        PkConvK k{};
        k.wimg = reinterpret_cast<const unsigned char*>(c.wimg);
        k.wimg_bytes = e3dge_modconv_packed_words(c.co, c.ci) * 4;
        k.noise = c.noise; k.noise_w = c.noise_w; k.noise_amax = c.noise_amax; k.bias = c.bias; k.bias_amax = c.bias_amax;
        k.knorm = sqrtf(9.0f * (float)c.ci);
        k.slope = P->negative_slope; k.act_scale = P->act_scale;
        k.B = B; k.Ci = c.ci; k.Co = c.co; k.H = res; k.W = res; k.n_chunks = c.ci / 16; k.noise_batch = c.noise_batch;
        return k;
    };
    // 5, 6. conv1 + ToRGB
    int res = P->in_res;
    {
        PkConvK k = conv_args(P->conv1, res);
        k.x = reinterpret_cast<const unsigned char*>(P->act[0]); k.in_meta = P->meta; k.in_amax = P->amax;
        k.y = reinterpret_cast<unsigned char*>(P->act[1]); k.out_meta = P->meta + 1; k.out_amax = P->amax + E3DGE_AMAX_FLOATS;
        DEC2_STEP(conv_s1(k, st));
        DEC2_STEP(launch_torgb(P->rgb1.out, reinterpret_cast<const unsigned char*>(P->act[1]), P->meta + 1, P->rgb1.wm, P->rgb1.bias,
                               nullptr, nullptr, B, P->rgb1.ci, res, st));
    }
    const float* skip = P->rgb1.out;
    int prev_act = 1;
    for (int u = 0; u < P->n_up; ++u) {
        const E3dgeDec2Conv& cu = P->up[u];
        const E3dgeDec2Conv& cc = P->conv[u];
        float* am_t = P->amax + (int64_t)E3DGE_AMAX_FLOATS * (2 + 3 * u);
        float* am_u = am_t + E3DGE_AMAX_FLOATS;
        float* am_v = am_u + E3DGE_AMAX_FLOATS;
        if (use_upblur(res)) {   // transposed conv + blur + noise + bias + lrelu -> packed in one launch, T stays on chip
            PkConvK k = conv_args(cu, res);
            k.knorm = sqrtf(4.0f * (float)cu.ci);
            k.x = reinterpret_cast<const unsigned char*>(P->act[prev_act]); k.in_meta = P->meta + prev_act;
            k.in_amax = P->amax + (int64_t)E3DGE_AMAX_FLOATS * (prev_act == 1 ? 1 : 4 + 3 * (u - 1));
            k.y = reinterpret_cast<unsigned char*>(P->act[2 + 2 * u]); k.out_meta = P->meta + 2 + 2 * u; k.out_amax = am_u;
            DEC2_STEP(launch_upblur(k, P->fir_blur, P->fir_blur_separable ? P->fir_blur_1d : nullptr, st));
            mark(true);                   // (keeps the kernel_ms slots aligned: this level's blur entry reads 0)
            res *= 2;
        } else {
        {   // transposed conv -> T
            PkConvK k = conv_args(cu, res);
            k.noise = nullptr; k.noise_w = nullptr; k.noise_amax = nullptr;
            k.x = reinterpret_cast<const unsigned char*>(P->act[prev_act]); k.in_meta = P->meta + prev_act;
            k.in_amax = P->amax + (int64_t)E3DGE_AMAX_FLOATS * (prev_act == 1 ? 1 : 4 + 3 * (u - 1));
            k.t = P->tbuf[u]; k.out_amax = am_t;
            DEC2_STEP(conv_up(k, st));
        }
        res *= 2;
        {   // blur + noise + bias + lrelu -> packed
            PkBlurK k{};
            k.t = P->tbuf[u]; k.y = reinterpret_cast<unsigned char*>(P->act[2 + 2 * u]); k.fir = P->fir_blur;
            k.noise = cu.noise; k.noise_w = cu.noise_w; k.noise_amax = cu.noise_amax; k.bias = cu.bias; k.bias_amax = cu.bias_amax;
            k.t_amax = am_t; k.out_meta = P->meta + 2 + 2 * u; k.out_amax = am_u;
            k.slope = P->negative_slope; k.act_scale = P->act_scale;
            k.B = B; k.C = cu.co; k.R = res; k.noise_batch = cu.noise_batch;
            k.tiles_x = (res + kPbCols - 1) / kPbCols; k.tiles_y = (res + kPbRows - 1) / kPbRows;
            const int64_t blocks = (int64_t)k.tiles_x * k.tiles_y * (cu.co / 8) * B;
            if (shape_override("E3DGE_DEC2_BLUR") == 1) {
                pk_blur_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(k);
            } else {
                if (!(cu.co <= 1024 && blocks < ((int64_t)1 << 30))) return finish(fail(E3DGE_ERR_INVALID_ARG, "dec2 blur: too many channels / tiles"));
                auto fn = &pk_blur2_kernel;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kPb2Lds);
                if (e != hipSuccess) return finish(fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(dec2 blur): %s", hipGetErrorString(e)));
                const int tiles_y2 = (res + kPb2Rows - 1) / kPb2Rows;
                const int64_t tiles2 = (int64_t)k.tiles_x * tiles_y2 * (cu.co / 8) * B;
                if (!(tiles2 < ((int64_t)1 << 30))) return finish(fail(E3DGE_ERR_INVALID_ARG, "dec2 blur: too many tiles"));
                const int grid = tiles2 < 512 ? (int)tiles2 : 512;        // two 70-KB workgroups per CU
                fn<<<dim3((unsigned)grid), dim3(kPb2Threads), kPb2Lds, st>>>(k, (int)tiles2, tiles_y2);
            }
            DEC2_STEP(check_launch("dec2 blur"));
        }
        }
        // ToRGB in the convolution's epilogue whenever one tile covers every output channel (32 / 64): at the last level the activation
        // is then never stored; at an earlier level (round 4) it is stored AND reduced -- the stand-alone ToRGB re-read all of it
        const bool last_level = u == P->n_up - 1;
        const bool fuse_rgb = s1_can_fuse_rgb(cc.co, cc.ci) && (last_level || shape_override("E3DGE_DEC2_FUSE_RGB_MID") != 0);
        {   // stride-1 conv
            PkConvK k = conv_args(cc, res);
            k.x = reinterpret_cast<const unsigned char*>(P->act[2 + 2 * u]); k.in_meta = P->meta + 2 + 2 * u; k.in_amax = am_u;
            k.y = reinterpret_cast<unsigned char*>(P->act[3 + 2 * u]); k.out_meta = P->meta + 3 + 2 * u; k.out_amax = am_v;
            if (fuse_rgb) {
                k.rgb_wm = P->rgb[u].wm; k.rgb_bias = P->rgb[u].bias; k.rgb_skip = skip; k.rgb_fir = P->fir_up; k.rgb_out = P->rgb[u].out;
                k.rgb_store = (last_level && !P->save_for_backward) ? 0 : 1;      // (a backward needs the top activation's signs)
            }
            DEC2_STEP(conv_s1(k, st));
        }
        if (fuse_rgb) mark(true);     // (keeps the kernel_ms slots aligned: this level's ToRGB entry reads 0)
        else DEC2_STEP(launch_torgb(P->rgb[u].out, reinterpret_cast<const unsigned char*>(P->act[3 + 2 * u]), P->meta + 3 + 2 * u, P->rgb[u].wm,
                                    P->rgb[u].bias, skip, P->fir_up, B, P->rgb[u].ci, res, st));
        skip = P->rgb[u].out;
        prev_act = 3 + 2 * u;
    }
#undef DEC2_STEP
    return finish(E3DGE_OK);
}


// ---- backward: d image -> d features (decoder2_bwd.h) --------------------------------------------------------------------------------
extern "C" int e3dge_dec2_prepack_weights_t(float* wpre_t, const float* weight, float scale, int co, int ci, int flip, e3dge_stream_t stream) {
    E3DGE_REQUIRE(wpre_t && weight && co > 0 && ci > 0, "dec2_prepack_weights_t: bad arguments");
    E3DGE_REQUIRE(ci % 32 == 0 && co % 16 == 0, "dec2_prepack_weights_t: needs ci %% 32 == 0 and co %% 16 == 0 (the transposed image's rows are the input channels)");
    const int64_t n = (int64_t)co * ci * 9;
    pk_prepack_t_kernel<<<dim3(512), dim3(256), 0, as_stream(stream)>>>(wpre_t, weight, scale, co, ci, co / 16, n, flip);
    return check_launch("dec2_prepack_weights_t");
}
extern "C" int64_t e3dge_dec2_pbuf_words(int batch, int channels, int res) {
    if (batch <= 0 || channels <= 0 || res <= 0 || res % 2) return 0;
    return (int64_t)batch * ((channels + 7) / 8) * 2 * 4 * (res / 2 + 2) * (res / 2 + 2) * 4;
}
extern "C" int e3dge_dec2_bwd_num_launches(int n_up) { return 5 + 4 * n_up; }

namespace e3dge {
constexpr int kDsChunk = 4096;            // pixels per block of pk_dstyle_sums_kernel
// channels / resolution of packed activation i of a plan ([1] conv1 out, [2 + 2u] up out, [3 + 2u] conv out)
static void dec2_act_shape(const E3dgeDec2Plan* P, int i, int* C, int* R) {
    if (i == 1) { *C = P->conv1.co; *R = P->in_res; return; }
    const int u = (i - 2) / 2;
    *C = (i & 1) ? P->conv[u].co : P->up[u].co;
    *R = P->in_res << (u + 1);
}
}  // namespace e3dge
extern "C" int64_t e3dge_dec2_dlatent_ws_floats(const E3dgeDec2Plan* P) {
    if (!P || P->batch <= 0 || P->n_up < 0 || P->n_up > E3DGE_DEC2_MAX_UP) return 0;
    int64_t n = (int64_t)P->batch * P->in_ch;
    for (int i = 1; i <= 2 * P->n_up + 1; ++i) {
        int C, R;
        e3dge::dec2_act_shape(P, i, &C, &R);
        n += (int64_t)P->batch * (C / 8) * ((((int64_t)R * R + e3dge::kDsChunk - 1) / e3dge::kDsChunk) + 1) * 24;      // chunk partials + their fold
    }
    return n + (int64_t)P->batch * (3 * P->n_up + 2) * 1024;                                                           // dL/ds per modulation row
}

extern "C" int e3dge_dec2_backward(const E3dgeDec2Plan* P, const E3dgeDec2BwdPlan* Q, e3dge_stream_t stream) {
    E3DGE_REQUIRE(P != nullptr && Q != nullptr, "dec2_backward: null plan");
    E3DGE_REQUIRE(P->batch >= 0 && P->n_up >= 0 && P->n_up <= E3DGE_DEC2_MAX_UP && P->in_res >= 4 && P->in_ch > 0 && P->in_ch % 32 == 0,
                  "dec2_backward: bad sizes (batch %d, n_up %d, in_res %d, in_ch %d)", P->batch, P->n_up, P->in_res, P->in_ch);
    if (P->batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(P->save_for_backward, "dec2_backward: the forward of this plan did not keep its top activation (save_for_backward)");
    E3DGE_REQUIRE(Q->d_img && Q->d_features && Q->amax && Q->meta && Q->bounds && P->fir_blur && P->fir_up, "dec2_backward: null pointer");
    const int n_up = P->n_up, B = P->batch;
    auto check_bc = [&](const E3dgeDec2Conv& c, const E3dgeDec2BwdConv& q, const char* what) -> int {
        E3DGE_REQUIRE(q.wpre_t && q.wcol && q.wimg_t && c.style && c.demod, "dec2_backward %s: null pointer", what);
        E3DGE_REQUIRE(c.ci % 32 == 0 && c.co % 32 == 0, "dec2_backward %s: needs ci %% 32 == 0 and co %% 32 == 0 (got %d, %d)", what, c.ci, c.co);
        E3DGE_REQUIRE(what[0] != 'u' || c.ci % 64 == 0, "dec2_backward up: the stride-2 data-gradient kernel owns 64 channels per workgroup (ci = %d)", c.ci);
        return E3DGE_OK;
    };
    int rc = check_bc(P->conv1, Q->conv1, "conv1");
    if (rc) return rc;
    E3DGE_REQUIRE(Q->gact[1] && P->act[1] && P->rgb1.wm && (n_up == 0 || Q->drgb[0]), "dec2_backward: workspace missing");
    for (int u = 0; u < n_up; ++u) {
        if ((rc = check_bc(P->up[u], Q->up[u], "up")) != 0 || (rc = check_bc(P->conv[u], Q->conv[u], "conv")) != 0) return rc;
        E3DGE_REQUIRE(Q->gact[2 + 2 * u] && Q->gact[3 + 2 * u] && P->act[2 + 2 * u] && P->act[3 + 2 * u] && P->rgb[u].wm && Q->pbuf && (u == n_up - 1 || Q->drgb[1 + u]),
                      "dec2_backward: level %d workspace missing", u);
    }
    hipStream_t st = as_stream(stream);
    const int n_l = e3dge_dec2_bwd_num_launches(n_up);
    const bool timing = Q->kernel_ms != nullptr;
    E3DGE_REQUIRE(!timing || Q->n_kernel_ms >= n_l, "dec2_backward: kernel_ms needs %d entries", n_l);
    hipEvent_t ev[8 + 4 * E3DGE_DEC2_MAX_UP];
    int n_ev = 0;
    bool ev_failed = false;
    auto mark = [&]() {
        if (!timing || ev_failed) return;
        if (hipEventCreate(&ev[n_ev]) != hipSuccess) { ev_failed = true; return; }
        if (hipEventRecord(ev[n_ev++], st) != hipSuccess) ev_failed = true;
    };
    auto finish = [&](int code) {
        if (timing) {
            if (code == 0 && ev_failed) code = fail(E3DGE_ERR_LAUNCH, "dec2_backward: HIP event create / record failed (kernel_ms)");
            if (code == 0 && n_ev > 0) {
                if (hipEventSynchronize(ev[n_ev - 1]) != hipSuccess) code = fail(E3DGE_ERR_LAUNCH, "dec2: hipEventSynchronize failed (kernel_ms)");
                for (int i = 0; i + 1 < n_ev; ++i)
                    if (hipEventElapsedTime(&Q->kernel_ms[i], ev[i], ev[i + 1]) != hipSuccess) Q->kernel_ms[i] = -1.0f;
            }
            for (int i = 0; i < n_ev; ++i) (void)hipEventDestroy(ev[i]);
        }
        return code;
    };
#define DEC2_STEP(expr) do { rc = (expr); if (rc) return finish(rc); mark(); } while (0)
    // amax buffers: [u + 1] d rgb of level u (u = -1: rgb1's; u = n_up - 1: d img), then G2 of level u ([n_up + 1 + (u + 1)]), G1 ([2 n_up + 2 + u]), P ([3 n_up + 2 + u])
    auto am_d = [&](int u) { return Q->amax + (int64_t)E3DGE_AMAX_FLOATS * (u + 1); };
    auto am_g2 = [&](int u) { return Q->amax + (int64_t)E3DGE_AMAX_FLOATS * (n_up + 2 + u); };
    auto am_g1 = [&](int u) { return Q->amax + (int64_t)E3DGE_AMAX_FLOATS * (2 * n_up + 2 + u); };
    auto am_p = [&](int u) { return Q->amax + (int64_t)E3DGE_AMAX_FLOATS * (3 * n_up + 2 + u); };
    // meta: G2 of level u at [u + 1], G1 at [n_up + 1 + u], P at [2 n_up + 1 + u];  bounds: conv1 [0], up[u] [1 + 2u], conv[u] [2 + 2u], rgb of level u [2 n_up + 2 + u]
    auto mt_g2 = [&](int u) { return Q->meta + (u + 1); };
    auto mt_g1 = [&](int u) { return Q->meta + (n_up + 1 + u); };
    auto mt_p = [&](int u) { return Q->meta + (2 * n_up + 1 + u); };
    auto bd_rgb = [&](int u) { return Q->bounds + (2 * n_up + 2 + u); };
    const E3dgeDec2Rgb& rgb_top = n_up ? P->rgb[n_up - 1] : P->rgb1;
    auto rgb_of = [&](int u) -> const E3dgeDec2Rgb& { return u < 0 ? P->rgb1 : P->rgb[u]; };
    auto drgb_of = [&](int u) -> const float* { return u == n_up - 1 ? Q->d_img : Q->drgb[u + 1]; };

    mark();
    {   // 1. operator norms + clear the amax block
        PkBndTab tab{};
        auto add = [&](const E3dgeDec2Conv& c, const E3dgeDec2BwdConv& q) {
            PkBndConv& w = tab.conv[tab.n_conv++];
            w.style = c.style; w.demod = c.demod; w.wcol = q.wcol; w.co = c.co; w.ci = c.ci;
        };
        add(P->conv1, Q->conv1);
        for (int u = 0; u < n_up; ++u) { add(P->up[u], Q->up[u]); add(P->conv[u], Q->conv[u]); }
        for (int u = -1; u < n_up; ++u) { PkBndRgb& w = tab.rgb[tab.n_rgb++]; w.wm = rgb_of(u).wm; w.ci = rgb_of(u).ci; }
        tab.batch = B; tab.out = Q->bounds; tab.zero = Q->amax; tab.n_zero = E3DGE_AMAX_FLOATS * (4 * n_up + 2);
        pk_bwd_bounds_kernel<<<dim3((unsigned)(tab.n_conv + tab.n_rgb)), dim3(256), 0, st>>>(tab);
        DEC2_STEP(check_launch("dec2 bwd bounds"));
    }
    {   // 2. transposed per-sample weight images
        PkWTab tab{};
        auto add = [&](const E3dgeDec2Conv& c, const E3dgeDec2BwdConv& q) {
            PkWConv& w = tab.conv[tab.n_conv++];
            w.wpre = q.wpre_t; w.style = c.style; w.demod = c.demod; w.img = q.wimg_t; w.co = c.ci; w.ci = c.co; w.n_chunks = c.co / 16;
            w.n_items = (c.ci / 32) * (c.co / 16) * 9 * 64;
            w.words = e3dge_modconv_packed_words(c.ci, c.co);
            w.swap = 1;
        };
        add(P->conv1, Q->conv1);
        for (int u = 0; u < n_up; ++u) { add(P->up[u], Q->up[u]); add(P->conv[u], Q->conv[u]); }
        pk_weights_kernel<<<dim3(96, (unsigned)tab.n_conv, (unsigned)B), dim3(256), 0, st>>>(tab);
        DEC2_STEP(check_launch("dec2 bwd weights"));
    }
    const int top_res = P->in_res << n_up;
    // 3. max |d img|
    DEC2_STEP(e3dge_amax(am_d(n_up - 1), Q->d_img, (int64_t)B * 3 * top_res * top_res, stream));
    {   // 4. G2[top] = lrelu'(act2[top]) sqrt 2 . ToRGB^T d img
        const int top_act = n_up ? 3 + 2 * (n_up - 1) : 1;
        PkRgbtK k{};
        k.d_img = Q->d_img; k.wm = rgb_top.wm; k.act = reinterpret_cast<const unsigned char*>(P->act[top_act]);
        k.y = reinterpret_cast<unsigned char*>(Q->gact[top_act]);
        k.d_amax = am_d(n_up - 1); k.rgb_l1 = bd_rgb(n_up - 1); k.out_meta = mt_g2(n_up - 1); k.out_amax = am_g2(n_up - 1);
        k.act_scale = P->act_scale; k.slope = P->negative_slope; k.B = B; k.C = rgb_top.ci; k.R = top_res;
        E3DGE_REQUIRE(k.C % 8 == 0, "dec2_backward: top channel count %d", k.C);
        pk_rgbt_mask_kernel<<<dim3((unsigned)(((int64_t)top_res * top_res + 256 * kRgbtPix - 1) / (256 * kRgbtPix)), (unsigned)(k.C / 8), (unsigned)B), dim3(256), 0, st>>>(k);
        DEC2_STEP(check_launch("dec2 bwd rgbT+mask"));
    }
    auto bwd_args = [&](const E3dgeDec2Conv& c, const E3dgeDec2BwdConv& q, int res) {
        PkConvK k{};
        k.wimg = reinterpret_cast<const unsigned char*>(q.wimg_t);
        k.wimg_bytes = e3dge_modconv_packed_words(c.ci, c.co) * 4;
        k.slope = P->negative_slope; k.act_scale = P->act_scale;
        k.B = B; k.Ci = c.co; k.Co = c.ci; k.H = res; k.W = res; k.n_chunks = c.co / 16;
        return k;
    };
    int res = top_res;
    for (int u = n_up - 1; u >= 0; --u) {
        const int a2 = 3 + 2 * u, a1 = 2 + 2 * u, prev = u == 0 ? 1 : 3 + 2 * (u - 1);
        {   // d rgb of the level below
            const int h = res >> 1;
            pk_drgb_down_kernel<<<dim3((unsigned)((h * h + 255) / 256), (unsigned)(3 * B)), dim3(256), 0, st>>>(Q->drgb[u], am_d(u - 1), drgb_of(u), P->fir_up, 3 * B, res);
            DEC2_STEP(check_launch("dec2 bwd d_rgb"));
        }
        {   // G1 = lrelu'(act1) sqrt 2 . conv^T G2
            PkConvK k = bwd_args(P->conv[u], Q->conv[u], res);
            k.x = reinterpret_cast<const unsigned char*>(Q->gact[a2]); k.in_meta = mt_g2(u); k.in_amax = am_g2(u);
            k.y = reinterpret_cast<unsigned char*>(Q->gact[a1]); k.out_meta = mt_g1(u); k.out_amax = am_g1(u);
            k.mask_act = reinterpret_cast<const unsigned char*>(P->act[a1]); k.bwd_wl1 = Q->bounds + 2 + 2 * u;
            DEC2_STEP(conv_s1_bwd<1>(k, st));
        }
        {   // P = phases of Blur^T G1
            PkDblurK k{};
            k.g = reinterpret_cast<const unsigned char*>(Q->gact[a1]); k.in_meta = mt_g1(u); k.in_amax = am_g1(u); k.fir = P->fir_blur;
            k.p = reinterpret_cast<unsigned char*>(Q->pbuf); k.out_meta = mt_p(u); k.out_amax = am_p(u);
            k.B = B; k.C = P->up[u].co; k.R = res;
            const int h = res >> 1;
            k.tiles_x = (h + 1 + kDbTJ - 1) / kDbTJ; k.tiles_y = (h + 1 + kDbTI - 1) / kDbTI;
            const int64_t blocks = (int64_t)k.tiles_x * k.tiles_y * (k.C / 8) * B;
            if (!(blocks < ((int64_t)1 << 31))) return finish(fail(E3DGE_ERR_INVALID_ARG, "dec2 bwd blur: too many tiles"));
            pk_dblur_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(k);
            DEC2_STEP(check_launch("dec2 bwd blur^T"));
        }
        res >>= 1;
        {   // G2 of the level below = lrelu'(act2) sqrt 2 . (conv_stride2(P) + ToRGB^T d rgb)
            PkConvK k = bwd_args(P->up[u], Q->up[u], res);
            k.x = reinterpret_cast<const unsigned char*>(Q->pbuf); k.in_meta = mt_p(u); k.in_amax = am_p(u);
            k.y = reinterpret_cast<unsigned char*>(Q->gact[prev]); k.out_meta = mt_g2(u - 1); k.out_amax = am_g2(u - 1);
            k.mask_act = reinterpret_cast<const unsigned char*>(P->act[prev]); k.bwd_wl1 = Q->bounds + 1 + 2 * u;
            k.rgbt_d = Q->drgb[u]; k.rgb_wm = rgb_of(u - 1).wm; k.rgbt_amax = am_d(u - 1); k.rgbt_l1 = bd_rgb(u - 1);
            DEC2_STEP(launch_down<2>(k, st, "dec2 convT^T<64co,4x32>"));      // (64 output channels per workgroup: check_bc requires up.ci % 64 == 0)
        }
    }
    {   // d features = conv1^T G2[-1]
        PkConvK k = bwd_args(P->conv1, Q->conv1, res);
        k.x = reinterpret_cast<const unsigned char*>(Q->gact[1]); k.in_meta = mt_g2(-1); k.in_amax = am_g2(-1);
        k.out_f32 = Q->d_features;
        DEC2_STEP(conv_s1_bwd<2>(k, st));
    }
    if (Q->d_latent) {
        // ---- optional: d latent from per-channel sums over the tensors the chain left behind (decoder2_bwd.h, "d latent") ----
        if (!(Q->ds_part && Q->ds_part_floats >= e3dge_dec2_dlatent_ws_floats(P) && P->features && P->mod_table))
            return finish(fail(E3DGE_ERR_INVALID_ARG, "dec2_backward: d_latent needs ds_part (e3dge_dec2_dlatent_ws_floats floats), the forward's features and mod_table"));
        if (!(P->style_dim <= 1024 && P->in_ch <= 1024 && P->conv1.co <= 1024 && P->n_mod == 3 * n_up + 2))
            return finish(fail(E3DGE_ERR_INVALID_ARG, "dec2_backward: d_latent needs style_dim and channel counts <= 1024 and the 3 n_up + 2 rows of the forward's modulation table"));
        float* part[2 * E3DGE_DEC2_MAX_UP + 2] = {};
        int nch[2 * E3DGE_DEC2_MAX_UP + 2] = {};
        float* cursor = Q->ds_part;
        float* p_conv1 = cursor; cursor += (int64_t)B * P->in_ch;
        for (int i = 1; i <= 2 * n_up + 1; ++i) {
            int C, R;
            dec2_act_shape(P, i, &C, &R);
            const bool odd = (i & 1) != 0;
            const int u = odd ? (i - 3) / 2 : (i - 2) / 2;                 // level (odd: conv / rgb of level u, u = -1: conv1 / rgb1)
            const E3dgeDec2Conv& prod = i == 1 ? P->conv1 : (odd ? P->conv[u] : P->up[u]);
            PkDsSumsK k{};
            k.act = reinterpret_cast<const unsigned char*>(P->act[i]); k.g = reinterpret_cast<const unsigned char*>(Q->gact[i]);
            k.act_meta = P->meta + i; k.g_meta = odd ? mt_g2(u) : mt_g1(u);
            k.noise = prod.noise; k.noise_w = prod.noise_w; k.noise_batch = prod.noise_batch; k.bias = prod.bias;
            if (odd) { k.wm = rgb_of(u).wm; k.drgb = drgb_of(u); }
            k.slope = P->negative_slope; k.act_scale = P->act_scale; k.C = C; k.R = R; k.chunk = kDsChunk;
            k.n_chunks = (int)(((int64_t)R * R + kDsChunk - 1) / kDsChunk);
            k.part = cursor; part[i] = cursor; nch[i] = k.n_chunks;
            cursor += (int64_t)B * (C / 8) * k.n_chunks * 24;
            pk_dstyle_sums_kernel<<<dim3((unsigned)k.n_chunks, (unsigned)(C / 8), (unsigned)B), dim3(256), 0, st>>>(k);
            if ((rc = check_launch("dec2 bwd style sums")) != 0) return finish(rc);
        }
        pk_dot_planes_kernel<<<dim3((unsigned)P->in_ch, (unsigned)B), dim3(256), 0, st>>>(p_conv1, P->features, Q->d_features, P->in_ch, P->in_res * P->in_res);
        if ((rc = check_launch("dec2 bwd d features . features")) != 0) return finish(rc);
        // the chunks of every tensor folded in one launch, then dL/ds per modulation row, then modulation^T
        float* sums[2 * E3DGE_DEC2_MAX_UP + 2] = {};
        PkDsFoldK kf{};
        int blocks = 0;
        for (int i = 1; i <= 2 * n_up + 1; ++i) {
            int C, R;
            dec2_act_shape(P, i, &C, &R);
            sums[i] = cursor; cursor += (int64_t)B * (C / 8) * 24;
            kf.t[i - 1] = PkDsFoldT{part[i], sums[i], C / 8, nch[i], blocks};
            blocks += B * (C / 8);
        }
        kf.n_tensors = 2 * n_up + 1;
        pk_dstyle_fold_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(kf);
        if ((rc = check_launch("dec2 bwd style sums (fold)")) != 0) return finish(rc);
        float* ds = cursor; cursor += (int64_t)B * P->n_mod * 1024;
        PkDsK kd{};
        kd.tab = P->mod_table; kd.n_rows = P->n_mod; kd.ds = ds;
        // table rows (Decoder._mod_layers): 0 conv1, 1 rgb1, then per level up (2 + 3u), conv (3 + 3u), rgb (4 + 3u)
        kd.row[0] = PkDsRow{p_conv1, sums[1], 3};
        kd.row[1] = PkDsRow{sums[1], nullptr, 2};
        for (int u = 0; u < n_up; ++u) {
            const int prev = u == 0 ? 1 : 3 + 2 * (u - 1), a1 = 2 + 2 * u, a2 = 3 + 2 * u;
            kd.row[2 + 3 * u] = PkDsRow{sums[prev], sums[a1], 0};
            kd.row[3 + 3 * u] = PkDsRow{sums[a1], sums[a2], 0};
            kd.row[4 + 3 * u] = PkDsRow{sums[a2], nullptr, 2};
        }
        pk_dstyle_kernel<<<dim3((unsigned)P->n_mod, (unsigned)B), dim3(256), 0, st>>>(kd);
        if ((rc = check_launch("dec2 bwd d styles")) != 0) return finish(rc);
        PkDlatK k{};
        k.tab = P->mod_table; k.ds = ds; k.n_rows = P->n_mod; k.n_latent = P->n_latent; k.style_dim = P->style_dim; k.d_latent = Q->d_latent;
        pk_dlatent_kernel<<<dim3((unsigned)P->n_latent, (unsigned)B, (unsigned)((P->style_dim + 63) / 64)), dim3(256), 0, st>>>(k);
        if ((rc = check_launch("dec2 bwd d latent")) != 0) return finish(rc);
    }
#undef DEC2_STEP
    return finish(E3DGE_OK);
}
