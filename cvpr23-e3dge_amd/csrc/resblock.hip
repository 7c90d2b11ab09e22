// Local-feature -> texture-FiLM head: per point
//     out = W_s x + W_1 relu(W_0 relu(x) + b_0) + b_1 ,   x in R^Cin (Cin <= 320; the released models use 301),
//     (alpha, beta) = split(out, 256)
// Reference: ResnetBlockFC.forward (project/models/helper_modules/resnetfc.py:49-58) as
// netLocal.local_feat_to_tex_modulations_linear (vendor/pifu/lib/model/HGPIFuGANNetResidualInputResnetFC.py:84-93), called
// from SirenLocalGlobal.forward_backbone (project/utils/volume_renderer.py:327-336) on (B,H,W,S,301) local features.
//
// Same machine as the SIREN kernels: 4 waves x 32 points per workgroup, a point's whole state in one lane pair, split-f16
// MFMA contractions (hi + lo, three products, fp32 accumulate) with per-point power-of-two block scaling of the B operand
// (the inputs are unbounded), weights streamed L2 -> LDS by the shared ChunkPipe in 32-KiB chunks (32 output rows x 256 k).
// K = 320 (Cin padded) is two chunks: a full one and one of which only the first four k-steps (64 k) are non-zero and
// executed.  x and r = relu(net) are kept as packed (hi, lo) words (160 registers each); relu(x) is formed from x's words
// on the fly per k-step (integer sign masks).  Neither the (P, 301) hidden activations nor r ever touch memory; the output
// goes straight to (alpha, beta).
#include "siren_common.h"

namespace e3dge {

constexpr int kRbKin = 320;               // padded input / hidden width
constexpr int kRbTilesIn = kRbKin / 32;   // 10
constexpr int kRbStepsIn = kRbKin / 16;   // 20 k-steps of 16
constexpr int kRbOut = 512;
constexpr int kRbTilesOut = kRbOut / 32;  // 16
constexpr int kRbChunksG1 = kRbTilesIn * 2;                 // W_0: per out tile chunk A (k 0..255), chunk B (k 256..319)
constexpr int kRbChunksG2 = kRbTilesOut * 4;                // per out tile: W_s A, W_s B, W_1 A, W_1 B
constexpr int kRbChunks = kRbChunksG1 + kRbChunksG2;        // 84
constexpr int64_t kRbOffBias0 = (int64_t)kRbChunks * kChunkFloats;    // b_0 [320]
constexpr int64_t kRbOffBias1 = kRbOffBias0 + kRbKin;                 // b_1 [512]
constexpr int64_t kRbOffAux = kRbOffBias1 + kRbOut;                   // [0] max_n ||W_0[n,:]||_2, [1] max |b_0|, [2..3] pad
constexpr int64_t kRbPackedFloats = kRbOffAux + 4;

constexpr int kRbXPitch = 36;             // floats per point row of the staging tile (16-B aligned rows)
constexpr int kRbLdsW = 0;
constexpr int kRbLdsX = kRbLdsW + kNBuf * kChunkFloats;               // [128][36] staging of one 32-feature slice of x
constexpr int kRbLdsB = kRbLdsX + kTilePts * kRbXPitch;               // b_0 [320], b_1 [512]
constexpr int kRbLdsFloats = kRbLdsB + kRbKin + kRbOut;
constexpr int kRbLdsBytes = kRbLdsFloats * 4;
static_assert(kRbLdsBytes <= 160 * 1024, "LDS budget");

// ---------------------------------------------------------------------------------------------------------------
// weight image: [chunk][16 k-steps][hi|lo][64 lanes][4 words of two f16], value kW16Scale * W[n][k] with
// n = 32 t + (lane & 31), k = 256 * half_chunk + (k-slot order of kOffBig16); rows / columns beyond the real sizes are 0
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
resblock_pack_kernel(float* __restrict__ packed, const float* __restrict__ w0, const float* __restrict__ b0,
                     const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ ws, int cin) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < kRbPackedFloats; e += (int64_t)gridDim.x * 256) {
        float v;
        if (e >= kRbOffAux) {
            continue;                                  // written by resblock_norm_kernel
        } else if (e >= kRbOffBias1) {
            v = b1[e - kRbOffBias1];
        } else if (e >= kRbOffBias0) {
            const int n = (int)(e - kRbOffBias0);
            v = n < cin ? b0[n] : 0.0f;
        } else {
            int64_t r = e;
            const int k = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int hl = r & 1; r >>= 1;
            const int g = r & 15; r >>= 4;
            const int chunk = (int)r;
            int t, kc, which;                         // which: 0 = W_0, 1 = W_s, 2 = W_1
            if (chunk < kRbChunksG1) { t = chunk >> 1; kc = chunk & 1; which = 0; }
            else { const int c2 = chunk - kRbChunksG1; t = c2 >> 2; kc = c2 & 1; which = 1 + ((c2 >> 1) & 1); }
            const int n = 32 * t + (lane & 31);
            unsigned word = 0;
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j = 2 * k + e2;
                const int kk = 256 * kc + 32 * (g >> 1) + 16 * (g & 1) + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                float w = 0.0f;
                if (kk < cin) {
                    if (which == 0) { if (n < cin) w = w0[(int64_t)n * cin + kk]; }       // fc_0: (cin, cin)
                    else if (which == 1) w = ws[(int64_t)n * cin + kk];                   // shortcut: (512, cin)
                    else w = w1[(int64_t)n * cin + kk];                                   // fc_1: (512, cin)
                }
                w *= kW16Scale;
                const _Float16 hi = (_Float16)w;
                const _Float16 val = hl ? (_Float16)(w - (float)hi) : hi;
                word |= (unsigned)__builtin_bit_cast(unsigned short, val) << (16 * e2);
            }
            v = __uint_as_float(word);
        }
        packed[e] = v;
    }
}

// bound on the hidden activations: |net_n| <= ||W_0[n,:]||_2 ||relu(x)||_2 + |b_0[n]|.  The two weight-side factors:
__global__ void __launch_bounds__(256)
resblock_norm_kernel(float* __restrict__ aux, const float* __restrict__ w0, const float* __restrict__ b0, int cin) {
    __shared__ float red[2][256];
    float rn = 0.0f, bm = 0.0f;
    for (int n = threadIdx.x; n < cin; n += 256) {
        float ss = 0.0f;
        for (int k = 0; k < cin; ++k) { const float w = w0[(int64_t)n * cin + k]; ss = fmaf(w, w, ss); }
        rn = fmaxf(rn, sqrtf(ss));
        bm = fmaxf(bm, fabsf(b0[n]));
    }
    red[0][threadIdx.x] = rn; red[1][threadIdx.x] = bm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] = fmaxf(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { aux[0] = red[0][0]; aux[1] = red[1][0]; aux[2] = 0.0f; aux[3] = 0.0f; }
}

// one weight chunk against KSTEPS k-steps of a B operand produced by `opnd(g, H, L)`; the same ring / barrier / DMA protocol
// as big_tile_f16 (siren_common.h), with the chunk barrier after k-step SYNC and the 8 DMA pieces spread over the rest
template <int KSTEPS, int SYNC, class Opnd, class Sync, class Dma>
__device__ __forceinline__ void rb_tile(const float* __restrict__ wchunk, const float* __restrict__ wnext, int lane,
                                        f32x16& acc, f32x16& accb, u32x4 (&ringH)[kRing16], u32x4 (&ringL)[kRing16],
                                        Opnd&& opnd, Sync&& sync, Dma&& dma) {
    const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(wchunk) + lane;
    const u32x4* __restrict__ wn = reinterpret_cast<const u32x4*>(wnext) + lane;
    constexpr int kAvail = KSTEPS - SYNC - 1;                 // k-steps after the barrier
    constexpr int kPer = (8 + kAvail - 1) / kAvail;           // DMA pieces per such step
    static_assert(kAvail >= 1 && SYNC + kRing16 - 1 >= 0, "tile too short");
#pragma unroll
    for (int g = 0; g < KSTEPS; ++g) {
        const int ga = g + kRing16 - 1;
        // the next chunk may only be touched after this tile's barrier
        static_assert(KSTEPS - (kRing16 - 1) > SYNC, "ring would read the next chunk before the barrier");
        ringH[ga % kRing16] = (ga < KSTEPS) ? wp[(ga * 2 + 0) * 64] : wn[((ga - KSTEPS) * 2 + 0) * 64];
        ringL[ga % kRing16] = (ga < KSTEPS) ? wp[(ga * 2 + 1) * 64] : wn[((ga - KSTEPS) * 2 + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 wh = ringH[g % kRing16], wl = ringL[g % kRing16];
        u32x4 bH, bL;
        opnd(g, bH, bL);
        f32x16& x0 = (g & 1) ? accb : acc;
        f32x16& x1 = (g & 1) ? acc : accb;
        x0 = mfma16(wh, bH, x0);
        x1 = mfma16(wl, bH, x1);
        x0 = mfma16(wh, bL, x0);
        if (g == SYNC) sync();
        if (g > SYNC) {
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int piece = (g - SYNC - 1) * kPer + i;
                if (piece < 8) dma(piece);
            }
        }
    }
}

struct ResblockK {
    const float* packed;
    const float* feats;     // (n_pts, cin)
    float* alpha;           // (n_pts, 256)
    float* beta;            // (n_pts, 256)
    long long n_pts;
    int cin, subtiles_per_wg;
};

// relu on a packed (hi, lo) pair of two f16 values each: both halves are cleared where hi is negative
__device__ __forceinline__ void relu_hilo(unsigned h, unsigned l, unsigned& rh, unsigned& rl) {
    const unsigned s = h & 0x80008000u;
    const unsigned keep = ~((s - (s >> 15)) | s);
    rh = h & keep;
    rl = l & keep;
}

__global__ void __launch_bounds__(kThreads) resblock_kernel(const ResblockK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kRbLdsW;
    float* const xs = smem + kRbLdsX;
    float* const b0_s = smem + kRbLdsB;
    float* const b1_s = b0_s + kRbKin;

    const int tid_k = threadIdx.x;
    const long long pt0 = (long long)blockIdx.x * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;
    const float* __restrict__ packed = a.packed;
    for (int i = tid_k; i < kRbKin + kRbOut; i += kThreads) b0_s[i] = packed[kRbOffBias0 + i];

    ChunkPipe pipe;
    pipe.init(wbuf, packed, tid_k >> 6, tid_k & 63, 0, kRbChunks);
    pipe.prime();
    auto issue_piece = [&](int i) { pipe.issue_piece(i); };
    auto chunk_sync = [&]() { pipe.sync(); };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4 ringH[kRing16], ringL[kRing16];
#pragma unroll
    for (int g = 0; g < kRing16 - 1; ++g) {
        ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + (tid_k & 63)];
        ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + (tid_k & 63)];
    }

    for (int sub = 0; sub < n_sub; ++sub) {
        int tid_o = tid_k;                                  // opaque per-iteration lane indices (see siren.hip)
        asm volatile("" : "+v"(tid_o));
        const int tid = tid_o, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
        const int p = sub * kTilePts + 32 * wave + col;
        const bool valid = p < npts;
        const long long gpt = pt0 + (valid ? p : npts - 1);

        // ---- 1. x: coalesced 32-feature slices through LDS, each lane keeps its point's values; then (hi, lo) ----
        u32x4 xH[kRbStepsIn], xL[kRbStepsIn];
        float inv_x, xnorm;
        {
            f32x16 xf[kRbTilesIn];
            float m = 0.0f, ss = 0.0f;
            const long long sub0 = pt0 + (long long)sub * kTilePts;
#pragma unroll
            for (int ft = 0; ft < kRbTilesIn; ++ft) {
                __syncthreads();                            // previous slice consumed
                // 128 points x 32 features = 4096 floats, 16 per thread: lane <-> feature, so a wave instruction reads
                // two 128-byte row segments
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int e = i * kThreads + tid, pp = e >> 5, f = 32 * ft + (e & 31);
                    const long long gp = sub0 + pp;
                    xs[pp * kRbXPitch + (e & 31)] = (gp < pt0 + npts && f < a.cin) ? a.feats[gp * a.cin + f] : 0.0f;
                }
                __syncthreads();
                const float* row = xs + (32 * wave + col) * kRbXPitch + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 q4 = *reinterpret_cast<const f32x4*>(row + 8 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xf[ft][4 * q + j] = q4[j]; m = fmaxf(m, fabsf(q4[j])); ss = fmaf(q4[j], q4[j], ss); }
                }
                asm volatile("" : "+a"(xf[ft]));
            }
            m = fmaxf(m, xhalf(m));
            ss += xhalf(ss);
            xnorm = sqrtf(ss);
            const unsigned e = min((__float_as_uint(m) >> 23) & 255u, 254u);
            const float sc = __uint_as_float((254u - e) << 23);
            inv_x = __uint_as_float((e > 8u ? e - 7u : 1u) << 23);
#pragma unroll
            for (int t = 0; t < kRbTilesIn; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    SPLIT2_TO(xf[t][r] * sc, xf[t][r + 1] * sc, xH[2 * t + (r >> 3)][(r & 7) >> 1], xL[2 * t + (r >> 3)][(r & 7) >> 1]);
        }

        // ---- 2. net = W_0 relu(x) + b_0 ; r = relu(net) as (hi, lo) words ----
        // r has to be split tile by tile (keeping it in fp32 until its column maximum is known costs 160 more registers
        // than there are), so its scale comes from the bound max|net| <= max_n ||W_0[n,:]|| * ||x|| + max|b_0| instead of the
        // exact maximum: typically a few bits of headroom, i.e. the operand is still good to ~2^-21 of the column maximum.
        u32x4 rH[kRbStepsIn], rL[kRbStepsIn];
        float inv_r;
        {
            const float bound = fmaf(packed[kRbOffAux], xnorm, packed[kRbOffAux + 1]);
            const unsigned er = min((__float_as_uint(bound) >> 23) & 255u, 253u);   // bound < 2^(er-126)
            const float sc_r = __uint_as_float((253u - er) << 23);          // r * sc_r < 1
            inv_r = __uint_as_float((er > 8u ? er - 6u : 1u) << 23);        // 1 / (128 * sc_r)
#pragma unroll
            for (int t = 0; t < kRbTilesIn; ++t) {
                f32x16 acc = zero16(), accb = zero16();
                rb_tile<16, kSyncStep16>(pipe.wcur, pipe.wnxt, lane, acc, accb, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) { unsigned h, l; relu_hilo(xH[g][w], xL[g][w], h, l); H[w] = h; L[w] = l; }
                    }, chunk_sync, issue_piece);
                pipe.advance();
                rb_tile<4, 0>(pipe.wcur, pipe.wnxt, lane, acc, accb, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) { unsigned h, l; relu_hilo(xH[16 + g][w], xL[16 + g][w], h, l); H[w] = h; L[w] = l; }
                    }, chunk_sync, issue_piece);
                pipe.advance();
                const f32x16 sum = (acc + accb) * inv_x;
                f32x16 rv;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(b0_s + 32 * t + 8 * q + 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rv[4 * q + j] = fmaxf(sum[4 * q + j] + b4[j], 0.0f) * sc_r;
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    SPLIT2_TO(rv[r], rv[r + 1], rH[2 * t + (r >> 3)][(r & 7) >> 1], rL[2 * t + (r >> 3)][(r & 7) >> 1]);
                asm volatile("" : "+a"(rH[2 * t]), "+a"(rH[2 * t + 1]), "+a"(rL[2 * t]), "+a"(rL[2 * t + 1]));
            }
        }

        // ---- 3. out = W_s x + W_1 r + b_1 -> alpha (tiles 0..7), beta (tiles 8..15) ----
#ifdef E3DGE_RB_UNROLL2
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int t = 0; t < kRbTilesOut; ++t) {
            f32x16 acc = zero16(), accb = zero16();
            rb_tile<16, kSyncStep16>(pipe.wcur, pipe.wnxt, lane, acc, accb, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = xH[g]; L = xL[g]; }, chunk_sync, issue_piece);
            pipe.advance();
            rb_tile<4, 0>(pipe.wcur, pipe.wnxt, lane, acc, accb, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = xH[16 + g]; L = xL[16 + g]; }, chunk_sync, issue_piece);
            pipe.advance();
            f32x16 res = (acc + accb) * inv_x;
            acc = zero16(); accb = zero16();
            rb_tile<16, kSyncStep16>(pipe.wcur, pipe.wnxt, lane, acc, accb, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = rH[g]; L = rL[g]; }, chunk_sync, issue_piece);
            pipe.advance();
            rb_tile<4, 0>(pipe.wcur, pipe.wnxt, lane, acc, accb, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = rH[16 + g]; L = rL[16 + g]; }, chunk_sync, issue_piece);
            pipe.advance();
            res = res + (acc + accb) * inv_r;
            float* __restrict__ dst = (t < 8 ? a.alpha : a.beta) + gpt * kWidth + 32 * (t & 7);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(b1_s + 32 * t + 8 * q + 4 * half);
                f32x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = res[4 * q + j] + b4[j];
                if (valid) *reinterpret_cast<f32x4*>(dst + 8 * q + 4 * half) = o4;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int64_t e3dge_resblock_packed_floats(void) { return kRbPackedFloats; }

extern "C" int e3dge_resblock_pack_weights(float* packed, const float* w0, const float* b0, const float* w1,
                                           const float* b1, const float* ws, int cin, e3dge_stream_t stream) {
    E3DGE_REQUIRE(packed && w0 && b0 && w1 && b1 && ws, "resblock_pack_weights: null pointer");
    E3DGE_REQUIRE(cin >= 1 && cin <= kRbKin, "resblock_pack_weights: cin=%d outside [1, %d]", cin, kRbKin);
    resblock_pack_kernel<<<dim3(1024), dim3(256), 0, as_stream(stream)>>>(packed, w0, b0, w1, b1, ws, cin);
    int rc = check_launch("resblock_pack_weights");
    if (rc) return rc;
    resblock_norm_kernel<<<dim3(1), dim3(256), 0, as_stream(stream)>>>(packed + kRbOffAux, w0, b0, cin);
    return check_launch("resblock_pack_weights(norms)");
}

extern "C" int e3dge_tex_modulations_fwd(const float* packed, const float* feats, int cin, int64_t n_pts,
                                         float* alpha, float* beta, e3dge_stream_t stream) {
    E3DGE_REQUIRE(n_pts >= 0, "tex_modulations_fwd: bad size");
    if (n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(packed && feats && alpha && beta, "tex_modulations_fwd: null pointer");
    E3DGE_REQUIRE(cin >= 1 && cin <= kRbKin, "tex_modulations_fwd: cin=%d outside [1, %d]", cin, kRbKin);
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(alpha) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0,
                  "tex_modulations_fwd: packed/alpha/beta must be 16-B aligned");
    {   // per device, cheap: set on every launch
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kRbLdsBytes);
        if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(resblock): %s", hipGetErrorString(e));
    }
    ResblockK k{};
    k.packed = packed; k.feats = feats; k.alpha = alpha; k.beta = beta; k.n_pts = n_pts; k.cin = cin;
    const int64_t tiles = (n_pts + kTilePts - 1) / kTilePts;
    int spw = (int)((tiles + 255) / 256);
    if (spw < 1) spw = 1;
    if (spw > 8) spw = 8;
    k.subtiles_per_wg = spw;
    const int64_t grid = (tiles + spw - 1) / spw;
    E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "tex_modulations_fwd: grid too large");
    resblock_kernel<<<dim3((unsigned)grid), dim3(kThreads), kRbLdsBytes, as_stream(stream)>>>(k);
    return check_launch("tex_modulations_fwd");
}
