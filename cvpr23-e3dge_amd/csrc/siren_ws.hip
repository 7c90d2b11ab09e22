// Weight-stationary split-f16 256 x 256 linear layers (e3dge_ws_linear): the layers of Fuse_sft_MLP, forward and backward.
//
// The renderer's kernels keep a wave's ACTIVATIONS in registers and stream the weights through LDS.  Here the roles are swapped: each of
// the 8 waves keeps its 32-feature slice of the layer's WEIGHTS in registers (128 VGPRs of packed hi / lo f16), the activations of a
// group of rows live in LDS, and a wave reads a third of a ds_read_b128 per MFMA.  (Rounds 3-4 also carried a study of the renderer's
// hidden-layer chain in this form, e3dge_ws_chain -- +7 % per hidden layer stand-alone, ~+4 % on the render kernel after the register
// <-> LDS hand-over; it was never merged and was removed in round 5 with DESIGN.md 4.1d's closing note: 0.41-0.42 of the f16 / 3
// roofline is final for the headline kernel at this numerics.)
//
// Fragments (v_mfma_f32_16x16x32_f16, siren16.h): lane l, n = l & 15, q = l >> 4
//   A (16 x 32): row n, k = 8q + j        B (32 x 16): column n, k = 8q + j        C/D: column n, rows 4q + r
// A = weights (feature 32 wave + 16 ft + n), B = activations (row 16 pt + n of the set): D[feature][row].
// LDS image of a set: [hi | lo][k-step g = 0..7][pt 2][q 4][n 16] x 16 B = the 8 features 32 g + 8 q .. + 7 of row 16 pt + n.
#define E3DGE_16_HELPERS_ONLY
#include <string.h>
#include "siren16.h"
#include "decoder_common.h"

namespace e3dge {

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

struct WsAcc { f32x4v t[2][2]; };     // [ft][pt]

}  // namespace e3dge

namespace e3dge {
// =====================================================================================================================
// One 256 x 256 linear layer on rows of a matrix, weight-stationary (the layers of Fuse_sft_MLP, local_query.py):
//     y[row, off_y + f] = post( sum_k W[f][k] pre(x[row, off_x + k]) + bias[f] + colw[f] pre(m[row]) + r1[row, f] + r2[row, f] )
// pre = relu or identity; post = identity | leaky relu | the SFT fuse  D + w (D S + v)  with D = r1, S = r2.
// A workgroup keeps the whole weight image in registers for its lifetime (8 waves x 32 output features)
// and walks groups of 64 rows: the rows of group g + 1 are fetched (global -> registers) before the contraction of group g
// and converted / written to the other LDS buffer after it.
// Operand scale (round 6): one power of two per ROW.  A row's 256 columns are staged by eight waves (32 columns = one k-step each),
// so no wave sees the row's maximum when it converts its block: every block is converted with the scale of its OWN maximum and leaves
// that scale in an LDS table; the contraction -- behind the group's barrier -- takes the row's scale as the smallest of the eight and
// multiplies the k-step's packed f16 fragments by the (power-of-two, <= 1) ratio on their way from LDS to the matrix pipe (four
// v_pk_mul_f16 per fragment, exact; a block more than 2^24 below its row's maximum becomes zero, as it should).  One accumulator set,
// no extra barrier.  (A first form kept per-block partial products and rescaled them in fp32: 2x slower -- every k-step's MFMAs
// became a dependent chain in front of sixteen VALU fmas -- and 352 B of scratch.)
// Until round 6 the scale was one per TENSOR (from its amax buffer): right for activations, wrong for the gradients of the backward
// chain, whose rows span many orders of magnitude -- rows far below the tensor's maximum lost the low (lo) half of the split and the
// stage-2 step's Fuse_sft_MLP gradients sat at 1e-4 (relative L2) where fp32 is at 2e-5 (tests/test_gpu_stage2.py).
// amax_in / amax_xmul are no longer read; the output's amax is still tracked for callers that want it.
// =====================================================================================================================
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };      // 16-byte access at 4-byte alignment (row pitch 513 floats)

struct WsLinK {
    const u32x4* wimg; const float* x; const float* amax_in; const float* bias; const float* colw; const float* m;
    const float* r1; const float* r2; float* y; float* amax_out;
    int64_t n_rows;
    int ld_x, off_x, ld_m, off_m, ld_r1, off_r1, ld_r2, off_r2, ld_y, off_y;
    int pre_relu, post;
    float slope, w_fuse;
    // ABI 12 (the backward chain of Fuse_sft_MLP, local_query.py): an element-wise multiplier on the input side
    const float* xmul; const float* amax_xmul; int ld_xmul, off_xmul; float x_scale; int reserved;
};
constexpr int kWlRows = 64, kWlBufBytes = 2 * kWsSetBytes;
constexpr int kWlScaleFloats = kWlRows * 8;                 // per buffer: [row 64][block 8] inverse operand scales

__global__ void __launch_bounds__(kWsThreads) ws_linear_kernel(const WsLinK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + 2 * kWlBufBytes);           // bias[256], colw[256]
    float* const inv_s = tab + 2 * kWidth;                                         // [buffer 2][row 64][block 8]: 1 / (128 * block scale)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, q = lane >> 4;
    const int64_t n_groups = (a.n_rows + kWlRows - 1) / kWlRows;
    if ((int64_t)blockIdx.x >= n_groups) return;
    const float xs = a.x_scale == 0.0f ? 1.0f : a.x_scale;           // (0 = unset: callers that zero-initialise the struct)
    if (tid < kWidth) { tab[tid] = a.bias ? a.bias[tid] : 0.0f; tab[kWidth + tid] = a.colw ? a.colw[tid] : 0.0f; }
    WsRegs R;
#pragma unroll
    for (int g = 0; g < kWsSteps; ++g) ws_load_w(R, a.wimg, 0, wave, lane, g);

    // staging: this lane's 8 columns 32 wave + 8 q .. + 7 of the rows 16 t + n (t = 0..3) of a group
    // (two of the four row tiles at a time -- half = the LDS set they belong to: 16 staging registers live across a contraction)
    f32x4 sa[2], sb[2];
    // Every load of this kernel is issued for ALL lanes from a clamped row (rows past the end re-read the last one and are zeroed by a
    // select) and the loads of a phase go out back to back: a load inside `if (row < n_rows)` / `if (ok)` sits in a basic block of its own and
    // the compiler's wait insertion drained the queue after each one -- 60 of the kernel's 80 vmcnt waits were vmcnt(0), every round trip to
    // HBM exposed: 74-130 us per launch for 0.2-0.4 GB (round 6; the texture head's loads taught the same lesson in round 2).
    const int64_t last_row = a.n_rows - 1;
    auto stage_load = [&](int64_t grp, int half) {
        F4U u0[2], u1[2], m0[2], m1[2];
        bool ok[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t row = grp * kWlRows + 32 * half + 16 * t + n;
            ok[t] = row <= last_row;
            const int64_t rc = ok[t] ? row : last_row;
            const float* p = a.x + rc * a.ld_x + a.off_x + 32 * wave + 8 * q;
            u0[t] = *reinterpret_cast<const F4U*>(p); u1[t] = *reinterpret_cast<const F4U*>(p + 4);
        }
        if (a.xmul) {                                       // x <- x (.) xmul * x_scale  (d scale = w g (.) dec of the SFT fuse's backward)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t row = grp * kWlRows + 32 * half + 16 * t + n;
                const float* m = a.xmul + (row <= last_row ? row : last_row) * a.ld_xmul + a.off_xmul + 32 * wave + 8 * q;
                m0[t] = *reinterpret_cast<const F4U*>(m); m1[t] = *reinterpret_cast<const F4U*>(m + 4);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float z = ok[t] ? xs : 0.0f;               // (x_scale and the zeroing of rows past the end in one factor)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa[t][i] = u0[t].v[i] * (a.xmul ? m0[t].v[i] * z : z);
                sb[t][i] = u1[t].v[i] * (a.xmul ? m1[t].v[i] * z : z);
            }
        }
    };
    auto stage_write = [&](int buf, int half) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 u = sa[t], w = sb[t];
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.pre_relu) { u[i] = fmaxf(u[i], 0.0f); w[i] = fmaxf(w[i], 0.0f); }
                m = fmaxf(m, fmaxf(fabsf(u[i]), fabsf(w[i])));
            }
            {   // max over the four lanes (q) that hold this row's 32 columns of the block: the exchanges of sum_over_q (siren16.h)
                const unsigned mu = __builtin_bit_cast(unsigned, m);
                const auto r1 = __builtin_amdgcn_permlane16_swap(mu, mu, false, false);
                m = fmaxf(__builtin_bit_cast(float, (unsigned)r1[0]), __builtin_bit_cast(float, (unsigned)r1[1]));
                const unsigned mv_ = __builtin_bit_cast(unsigned, m);
                const auto r2 = __builtin_amdgcn_permlane32_swap(mv_, mv_, false, false);
                m = fmaxf(__builtin_bit_cast(float, (unsigned)r2[0]), __builtin_bit_cast(float, (unsigned)r2[1]));
            }
            // block maximum into [2^13, 2^14): m in [2^(e-127), 2^(e-126)) -> scale 2^(140 - e); 1 / (128 scale) = 2^(e - 147)
            const unsigned e = min(max((__float_as_uint(m) >> 23) & 255u, 24u), 254u);      // (an all-zero block: any scale; inf / nan: propagate)
            const float in_scale = __uint_as_float((267u - e) << 23);
            {   // (address from an opaque copy of the lane index: kept live across the kernel these few words of address arithmetic are
                //  what tips the 256-register allocation into spilling weight fragments)
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                if ((lane_o >> 4) == 0) inv_s[(buf * kWlRows + 32 * half + 16 * t + (lane_o & 15)) * 8 + wave] = __uint_as_float((e - 20u) << 23);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { u[i] *= in_scale; w[i] *= in_scale; }
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
            int n_o = n;
            asm volatile("" : "+v"(n_o));
            const float* const inv_r = inv_s + (buf * kWlRows + st * 32 + n_o) * 8;    // [pt * 128 + block]: 1 / (128 x block scale)
            float inv_row[2], rinv_row[2], inv_g[2][2];                                   // row scale (largest block maximum), its reciprocal; [k-step parity][pt]
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const f32x4v i0 = *reinterpret_cast<const f32x4v*>(inv_r + pt * 128), i1 = *reinterpret_cast<const f32x4v*>(inv_r + pt * 128 + 4);
                inv_row[pt] = fmaxf(fmaxf(fmaxf(i0[0], i0[1]), fmaxf(i0[2], i0[3])), fmaxf(fmaxf(i1[0], i1[1]), fmaxf(i1[2], i1[3])));
                rinv_row[pt] = __uint_as_float((254u << 23) - __float_as_uint(inv_row[pt]));       // exact reciprocal of a power of two
                inv_g[0][pt] = i0[0];
            }
            auto rescaled = [](u32x4 v, float ratio) -> u32x4 {      // the 8 packed f16 of a fragment x ratio, a power of two <= 1 (four v_pk_mul_f16)
                const _Float16 r = (_Float16)ratio;                    // exact
                const half8 r8 = {r, r, r, r, r, r, r, r};
                return __builtin_bit_cast(u32x4, __builtin_bit_cast(half8, v) * r8);
            };
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
                        inv_g[(g + 1) & 1][pt] = inv_r[pt * 128 + g + 1];
                    }
                }
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const float ratio = inv_g[g & 1][pt] * rinv_row[pt];
                    xh[g & 1][pt] = rescaled(xh[g & 1][pt], ratio);
                    xl[g & 1][pt] = rescaled(xl[g & 1][pt], ratio);
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
            // (all residual loads of the set first, unconditional lanes on clamped rows -- see stage_load)
            F4U d1[2][2], d2[2][2];
            float mvs[2] = {0.0f, 0.0f};
            bool okr[2];
            int64_t rcl[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int64_t row = grp * kWlRows + st * 32 + 16 * pt + n;
                okr[pt] = row <= last_row;
                rcl[pt] = okr[pt] ? row : last_row;
            }
            if (a.r1) {
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) d1[pt][ft] = *reinterpret_cast<const F4U*>(a.r1 + rcl[pt] * a.ld_r1 + a.off_r1 + 32 * wave + 16 * ft + 4 * q);
            }
            if (a.r2) {
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) d2[pt][ft] = *reinterpret_cast<const F4U*>(a.r2 + rcl[pt] * a.ld_r2 + a.off_r2 + 32 * wave + 16 * ft + 4 * q);
            }
            if (a.m) {
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) { mvs[pt] = a.m[rcl[pt] * a.ld_m + a.off_m]; if (a.pre_relu) mvs[pt] = fmaxf(mvs[pt], 0.0f); }
            }
            // (the next group's half is converted and written to LDS HERE, between the residual loads and their first use: ~150 VALU
            // instructions under the loads' round trip instead of behind it)
            if (has_next) stage_write(buf ^ 1, st);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int64_t row = rcl[pt];
                const bool ok = okr[pt];
                const float mv = mvs[pt];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    const int f0 = 32 * wave + 16 * ft + 4 * q;
                    const f32x4v b4 = *reinterpret_cast<const f32x4v*>(tab + f0), c4 = *reinterpret_cast<const f32x4v*>(tab + kWidth + f0);
                    F4U out;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float e1 = a.r1 ? d1[pt][ft].v[i] : 0.0f, e2 = a.r2 ? d2[pt][ft].v[i] : 0.0f;
                        float t = fmaf(acc.t[ft][pt][i], inv_row[pt], b4[i]);
                        t = fmaf(c4[i], mv, t);
                        if (a.post == 2) {
                            t = fmaf(a.w_fuse, fmaf(e1, e2, t), e1);                           // D + w (D S + shift)
                        } else if (a.post == 3) {
                            t = fmaf(t, e1 > 0.0f ? 1.0f : a.slope, e2);                        // act'(r1) * v + r2: the backward through lrelu / relu
                        } else if (a.post == 4) {
                            t = fmaf(e1, fmaf(a.w_fuse, e2, 1.0f), t);                         // v + D (1 + w S): d dec of the SFT fuse
                        } else {
                            t = (t + e1) + e2;
                            if (a.post == 1) t = fmaxf(t, t * a.slope);                         // leaky relu, 0 <= slope <= 1
                        }
                        out.v[i] = t;
                        if (ok) amax_l = fmaxf(amax_l, fabsf(t));
                    }
                    if (ok) *reinterpret_cast<F4U*>(a.y + row * a.ld_y + a.off_y + f0) = out;
                }
            }
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
// weight image of ws_linear_kernel from fp32 weights (n_layers, 256 out, 256 in): word `wd` (two f16) of lane l (n = l & 15,
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


namespace e3dge {
// out[row * ld_out + off_out] = sum_k a[row][k] u[k] + gate(row) * sum_k b[row][k] v[k],  gate = (g[row * ld_g + off_g] > 0) or 1 without g:
// the data gradient of ONE extra input column of two 256-wide linear layers (the visibility-mask column of Fuse_sft_MLP's 513-wide
// input: d x[:, 256] = de Ws[:, 256] + (dnet W0[:, 256]) [x[:, 256] > 0]).  One wave per row, 16 B per lane; fixed-order reduction.
__global__ void __launch_bounds__(256)
ws_rowdot2_kernel(float* __restrict__ out, int ld_out, int off_out, const float* __restrict__ a, const float* __restrict__ u,
                  const float* __restrict__ b, const float* __restrict__ v, const float* __restrict__ g, int ld_g, int off_g, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const f32x4v u4 = *reinterpret_cast<const f32x4v*>(u + 4 * lane), v4 = *reinterpret_cast<const f32x4v*>(v + 4 * lane);
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const f32x4v a4 = *reinterpret_cast<const f32x4v*>(a + row * kWidth + 4 * lane), b4 = *reinterpret_cast<const f32x4v*>(b + row * kWidth + 4 * lane);
        float sa = fmaf(a4[3], u4[3], fmaf(a4[2], u4[2], fmaf(a4[1], u4[1], a4[0] * u4[0])));
        float sb = fmaf(b4[3], v4[3], fmaf(b4[2], v4[2], fmaf(b4[1], v4[1], b4[0] * v4[0])));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sa += __shfl_xor(sa, off, kWave); sb += __shfl_xor(sb, off, kWave); }
        if (lane == 0) out[row * ld_out + off_out] = sa + ((!g || g[row * ld_g + off_g] > 0.0f) ? sb : 0.0f);
    }
}
}  // namespace e3dge

extern "C" int e3dge_ws_rowdot2(float* out, int ld_out, int off_out, const float* a, const float* u, const float* b, const float* v,
                                const float* gate, int ld_gate, int off_gate, int64_t n_rows, e3dge_stream_t stream) {
    using namespace e3dge;
    E3DGE_REQUIRE(out && a && u && b && v && n_rows >= 0 && ld_out > off_out && off_out >= 0 && (!gate || (ld_gate > off_gate && off_gate >= 0)),
                  "ws_rowdot2: bad arguments");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                  "ws_rowdot2: a, b (n_rows, 256) and u, v (256) must be 16-byte aligned");
    if (n_rows == 0) return E3DGE_OK;
    const int64_t blocks = (n_rows + 3) / 4;
    ws_rowdot2_kernel<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, as_stream(stream)>>>(out, ld_out, off_out, a, u, b, v, gate, ld_gate, off_gate, n_rows);
    return check_launch("ws_rowdot2");
}

static_assert(sizeof(E3dgeWsLinear) == sizeof(e3dge::WsLinK), "E3dgeWsLinear mirrors WsLinK");

extern "C" int e3dge_ws_linear(const E3dgeWsLinear* args, e3dge_stream_t stream) {
    using namespace e3dge;
    E3DGE_REQUIRE(args && args->wimg && args->x && args->y, "ws_linear: null pointer");
    E3DGE_REQUIRE(args->n_rows >= 0 && args->ld_x >= args->off_x + 256 && args->ld_y >= args->off_y + 256 && args->off_x >= 0 && args->off_y >= 0,
                  "ws_linear: a 256-column block must fit the row pitch (x: ld %d off %d, y: ld %d off %d)", args->ld_x, args->off_x, args->ld_y, args->off_y);
    E3DGE_REQUIRE(args->post >= 0 && args->post <= 4 && ((args->post != 2 && args->post != 4) || (args->r1 && args->r2)) && (args->post != 3 || args->r1),
                  "ws_linear: post must be 0..4 (2 and 4 need r1 = D and r2 = S, 3 needs r1 = the activation whose sign selects the slope)");
    E3DGE_REQUIRE(!args->xmul || args->ld_xmul >= args->off_xmul + 256, "ws_linear: xmul does not fit its row pitch");
    E3DGE_REQUIRE((!args->r1 || args->ld_r1 >= args->off_r1 + 256) && (!args->r2 || args->ld_r2 >= args->off_r2 + 256) && (!args->m || args->ld_m > args->off_m),
                  "ws_linear: residual / column operand does not fit its row pitch");
    E3DGE_REQUIRE(!args->colw == !args->m, "ws_linear: colw and m come together");
    E3DGE_REQUIRE((args->post != 1 && args->post != 3) || (args->slope >= 0.0f && args->slope <= 1.0f), "ws_linear: leaky slope must be in [0, 1]");
    if (args->n_rows == 0) return E3DGE_OK;
    WsLinK k;
    memcpy(&k, args, sizeof(k));
    const int lds = 2 * kWlBufBytes + 2 * kWidth * 4 + 2 * kWlScaleFloats * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ws_linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(ws_linear): %s", hipGetErrorString(e));
    const int64_t n_groups = (k.n_rows + kWlRows - 1) / kWlRows;
    const int grid = n_groups < 256 ? (int)n_groups : 256;
    ws_linear_kernel<<<dim3((unsigned)grid), dim3(kWsThreads), lds, as_stream(stream)>>>(k);
    return check_launch("ws_linear");
}
