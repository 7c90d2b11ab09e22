// Data gradient of the texture head (round 5; included by resblock.hip inside namespace e3dge):
//     out = W_s x + W_1 relu(net) + b_1,  net = W_0 relu(x) + b_0          (ResnetBlockFC.forward, helper_modules/resnetfc.py:49-58)
//     d net = (W_1^T d out) * [net > 0]      d x = W_s^T d out + (W_0^T d net) * [x > 0]
// i.e. what autograd runs through the reference's module for the stage-2 losses (e3dge_full_runner.py:185-317) -- the reference has no
// hand-written backward here; `_resblock_backward_torch` (volume_renderer.py of this package) was the library form of the same lines.
//
// Same machine as the forward kernel (4 waves x 32 points, a point in a lane pair, split-f16 MFMA with per-point power-of-two block
// scales, weights streamed L2 -> LDS in 20-KiB chunks through RbPipe), four contractions per 128-point sub-tile:
//   G1  net = W_0 relu(x) + b_0     K 320 -> 320   x resident (160 operand registers); only the SIGNS of net are kept (sign words in LDS),
//                                                  x's own signs likewise; then x is dropped
//   G2  d net = (W_1^T d out) [net > 0]   K 512 -> 320   d out resident (256 operand registers); d net -> workspace rows (fp32, 320 wide)
//   G4  short = W_s^T d out               K 512 -> 320   d out still resident; -> second workspace
//   G3  d x = (W_0^T d net) [x > 0] + short   K 320 -> 320   d net read back as the operand (160 registers); `short` comes in by LDS-DMA two
//                                                  tiles ahead (a plain load inside the tile loop would drain the weight pipe: resblock_kernel, FILM)
// d out (512 values per point) and d net (320) cannot be resident together (416 of 512 registers before accumulators, ring and addresses),
// hence the 2.5 KB per point of workspace traffic (L2: written and read back by the same workgroup within a sub-tile).  All tile loops are
// rolled: output tiles only differ in addresses, so per-tile state that would need register indexing (the sign words) lives in LDS.
// K = 512 is 32 k-steps = 3.2 chunks of ten: the image pads it to four, the MFMAs of the padding are skipped at compile time (KACT).
#pragma once

constexpr int kRbBChunks = 120;                 // G1: 10 tiles x 2 | G2: 10 x 4 | G4: 10 x 4 | G3: 10 x 2 -- in the order they are consumed
constexpr int64_t kRbBOffBias0 = (int64_t)kRbBChunks * kRbChunkFloats;      // b_0 [320]
constexpr int64_t kRbBPackedFloats = kRbBOffBias0 + kRbKin;

// weight image of the backward: chunk layout of resblock_pack_kernel ([10 k-steps][hi|lo][64 lanes][4 words]); rows n = output of the contraction
__global__ void __launch_bounds__(256)
resblock_bwd_pack_kernel(float* __restrict__ packed, const float* __restrict__ w0, const float* __restrict__ b0,
                         const float* __restrict__ w1, const float* __restrict__ ws, int cin) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < kRbBPackedFloats; e += (int64_t)gridDim.x * 256) {
        float v;
        if (e >= kRbBOffBias0) {
            const int n = (int)(e - kRbBOffBias0);
            v = n < cin ? b0[n] : 0.0f;
        } else {
            int64_t r = e;
            const int k = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int hl = r & 1; r >>= 1;
            const int chunk = (int)(r / kRbCSteps);
            const int g = (int)(r - (int64_t)chunk * kRbCSteps);
            int t, kc, which;                         // which: 0 = W_0 (rows = net), 1 = W_1^T, 2 = W_0^T, 3 = W_s^T (rows = hidden / input feature)
            if (chunk < 20) { t = chunk >> 1; kc = chunk & 1; which = 0; }
            else if (chunk < 60) { const int c2 = chunk - 20; t = c2 >> 2; kc = c2 & 3; which = 1; }
            else if (chunk < 100) { const int c2 = chunk - 60; t = c2 >> 2; kc = c2 & 3; which = 3; }
            else { const int c2 = chunk - 100; t = c2 >> 1; kc = c2 & 1; which = 2; }
            const int n = 32 * t + (lane & 31);
            unsigned word = 0;
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j = 2 * k + e2;
                const int gg = kRbCSteps * kc + g;
                const int kk = 32 * (gg >> 1) + 16 * (gg & 1) + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                float w = 0.0f;
                if (n < cin) {
                    if (which == 0) { if (kk < cin) w = w0[(int64_t)n * cin + kk]; }
                    else if (which == 1) { if (kk < kRbOut) w = w1[(int64_t)kk * cin + n]; }
                    else if (which == 2) { if (kk < cin) w = w0[(int64_t)kk * cin + n]; }
                    else { if (kk < kRbOut) w = ws[(int64_t)kk * cin + n]; }
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

// LDS of the backward kernel: the forward's weight ring and b_0, then the sign words and the three 4-KiB `short` buffers of each wave
constexpr int kRbBLdsMask = ((kRbLdsB + kRbKin + 3) / 4) * 4;            // [2 (x | net)][5 words][256 threads]
constexpr int kRbBLdsShort = kRbBLdsMask + 2 * (kRbTilesIn / 2) * kThreads;   // [4 waves][3 buffers][4 quads][64 lanes] x 16 B
constexpr int kRbBLdsFloats = kRbBLdsShort + 4 * 3 * 1024;
constexpr int kRbBLdsBytes = kRbBLdsFloats * 4;
static_assert(kRbBLdsBytes <= 160 * 1024, "LDS budget (backward)");
constexpr int kRbWsRow = kRbKin;                                         // workspace rows: 320 floats (16-byte aligned whatever cin is)

// one weight chunk against KACT of its KSTEPS k-steps (the rest is the zero padding of K = 512): rb_tile's operand ring, chunk barrier and
// DMA hand-out; `extra(g)` runs in front of k-step g
template <int KSTEPS, int SYNC, int POS, int KACT, class Opnd, class Sync, class Dma, class Extra>
__device__ __forceinline__ void rb_tile_b(const float* __restrict__ wchunk, const float* __restrict__ wnext, int lane, f32x16& acc, f32x16& accb,
                                          u32x4 (&ringH)[kRbRing], u32x4 (&ringL)[kRbRing], Opnd&& opnd, Sync&& sync, Dma&& dma, Extra&& extra) {
    const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(wchunk) + lane;
    const u32x4* __restrict__ wn = reinterpret_cast<const u32x4*>(wnext) + lane;
    constexpr int PH = (POS * KSTEPS) % kRbRing;
    static_assert((2 * KSTEPS) % kRbRing == 0 && KSTEPS - SYNC - 1 >= kRbPieces && KSTEPS - (kRbRing - 1) > SYNC, "chunk pipeline shape");
#pragma unroll
    for (int g = 0; g < KSTEPS; ++g) {
        extra(g);
        const int ga = g + kRbRing - 1;
        ringH[(ga + PH) % kRbRing] = (ga < KSTEPS) ? wp[(ga * 2 + 0) * 64] : wn[((ga - KSTEPS) * 2 + 0) * 64];
        ringL[(ga + PH) % kRbRing] = (ga < KSTEPS) ? wp[(ga * 2 + 1) * 64] : wn[((ga - KSTEPS) * 2 + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);
        if (g < KACT) {
            const u32x4 wh = ringH[(g + PH) % kRbRing], wl = ringL[(g + PH) % kRbRing];
            u32x4 bH, bL;
            opnd(g, bH, bL);
            f32x16& x0 = (g & 1) ? accb : acc;
            x0 = mfma16(wh, bH, x0);
            f32x16& x1 = (g & 1) ? acc : accb;
            x1 = mfma16(wl, bH, x1);
            x0 = mfma16(wh, bL, x0);
        }
        if (g == SYNC) sync();
        if (g > SYNC && g - SYNC - 1 < kRbPieces) dma(g - SYNC - 1);
    }
}

struct ResblockBwdK {
    const float* packed;        // resblock_bwd_pack_kernel's image
    const float* feats;         // (n_pts, cin)
    const float* d_alpha;       // (n_pts, 256) each: d out = [d alpha | d beta]
    const float* d_beta;
    float* d_feats;             // (n_pts, cin) out
    float* ws;                  // 2 x (n_pts, 320): d net (valid on return, rows padded to 320) | W_s^T d out
    float* net_out;             // null, or (n_pts, 320): net as recomputed by G1
    float* amax4;               // null, or four amax buffers (zeroed by the caller): max |feats|, |d out|, |d net|, |net| -- the operand scales of e3dge_wgrad
    long long n_pts;
    int cin, subtiles_per_wg;
};

// the wave's largest value into an amax buffer (round 6: the kernel holds every operand of the head's parameter gradients anyway; four
// e3dge_amax passes over 100-MB tensors -- 120 us per stage-2 step -- are not needed)
__device__ __forceinline__ void rb_amax_publish(float* buf, float m, int lane, int slot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    if (lane == 0) atomic_max_nonneg(buf + (slot & (kAmaxSlots - 1)) * kAmaxStride, m);
}

// power-of-two block scale of a point whose largest magnitude is m: operand = value * sc in [1, 2), accumulator * inv = true sum (the
// weight image carries kW16Scale = 128); the forward's rule (resblock_kernel, "x")
__device__ __forceinline__ void rb_block_scale(float m, float& sc, float& inv) {
    const unsigned e = min((__float_as_uint(m) >> 23) & 255u, 254u);
    sc = __uint_as_float((254u - e) << 23);
    inv = __uint_as_float((e > 8u ? e - 7u : 1u) << 23);
}

// sign words: 32 decisions [v > 0] per word, pushed in from the right (decision i of a word ends at bit 31 - i); integer ops only
__device__ __forceinline__ void rb_push_sign(unsigned& word, float v) {
    const int t = max(__float_as_int(v), 0);                       // > 0 exactly for positive floats (-0.0 and negatives -> 0)
    word = (word << 1) | min((unsigned)t, 1u);
}
__device__ __forceinline__ float rb_keep_if(unsigned word, int i, float v) {      // v where decision i of the word was "positive", else +0
    const int m = __builtin_amdgcn_sbfe((int)word, (unsigned)(31 - i), 1u);       // 0 or -1
    return __uint_as_float(__float_as_uint(v) & (unsigned)m);
}

__global__ void __launch_bounds__(kThreads) resblock_bwd_kernel(const ResblockBwdK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kRbLdsW;
    float* const b0_s = smem + kRbLdsB;
    unsigned* const msk = reinterpret_cast<unsigned*>(smem + kRbBLdsMask);       // [x | net][word][thread]

    const int tid_k = threadIdx.x;
    const long long pt0 = (long long)blockIdx.x * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;
    const float* __restrict__ packed = a.packed;
    float* const ws_net = a.ws + pt0 * kRbWsRow;                                 // wave-uniform bases of this workgroup's rows
    float* const ws_short = a.ws + (a.n_pts + pt0) * kRbWsRow;
    for (int i = tid_k; i < kRbKin; i += kThreads) b0_s[i] = packed[kRbBOffBias0 + i];

    RbPipe pipe;
    pipe.init(wbuf, packed, tid_k >> 6, tid_k & 63, kRbBChunks);
    pipe.prime();
    auto issue_piece = [&](int i) { pipe.issue_piece(i); };
    auto chunk_sync = [&]() { pipe.sync(); };
    auto nothing = [](int) {};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4 ringH[kRbRing], ringL[kRbRing];
#pragma unroll
    for (int g = 0; g < kRbRing - 1; ++g) {
        ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + (tid_k & 63)];
        ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + (tid_k & 63)];
    }

    for (int sub = 0; sub < n_sub; ++sub) {
        int tid_o = tid_k;                                  // opaque per-iteration lane indices (see siren.hip)
        asm volatile("" : "+v"(tid_o));
        const int tid = tid_o, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
        const int p = sub * kTilePts + 32 * wave + col;
        const bool valid = p < npts;
        const int pc = valid ? p : npts - 1;                 // padded lanes shadow the last valid point (and never store)
        const long long gpt = pt0 + pc;
        const unsigned row_off = (unsigned)pc * kRbWsRow + 4u * half;           // this lane's workspace row, floats from the workgroup's base
        f32x16 P0, P1;

        // ---- 1. x -> (hi, lo) words (relu applied per k-step, as in the forward) and the sign words of x ----
        {
            u32x4 xH[kRbStepsIn], xL[kRbStepsIn];
            float inv_x;
            {
                f32x16 xf[kRbTilesIn];
                float m = 0.0f;
                // the forward's load of a cin-float row at 4-byte alignment (resblock_kernel, "1. x": clamped start + shift for the tensor's last rows)
                const long long total = a.n_pts * (long long)a.cin;
                const long long row0 = gpt * a.cin;
                if (total >= 4) {
                    const long long sl = total - 4 - row0;
                    const int slack = (int)(sl < 4096 ? sl : 4096);
                    const float* __restrict__ xr = a.feats + row0;
#pragma unroll
                    for (int ft = 0; ft < kRbTilesIn; ++ft) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int f0 = 32 * ft + 8 * q + 4 * half;
                            const F4u v = *reinterpret_cast<const F4u*>(xr + min(f0, slack));
#pragma unroll
                            for (int j = 0; j < 4; ++j) xf[ft][4 * q + j] = v.v[j];
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(slack < kRbKin - 4)) {
#pragma unroll
                        for (int ft = 0; ft < kRbTilesIn; ++ft) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int f0 = 32 * ft + 8 * q + 4 * half;
                                const int d = f0 - min(f0, slack);
                                float v[4], w[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = xf[ft][4 * q + j];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    w[j] = v[j];
                                    if (j + 1 < 4) w[j] = d == 1 ? v[j + 1] : w[j];
                                    if (j + 2 < 4) w[j] = d == 2 ? v[j + 2] : w[j];
                                    if (j + 3 < 4) w[j] = d == 3 ? v[j + 3] : w[j];
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) xf[ft][4 * q + j] = w[j];
                            }
                        }
                    }
                    const int lim = a.cin - 4 * half;
#pragma unroll
                    for (int ft = 0; ft < kRbTilesIn; ++ft)
#pragma unroll
                        for (int r = 0; r < 16; ++r) xf[ft][r] = (32 * ft + 8 * (r >> 2) + (r & 3) < lim) ? xf[ft][r] : 0.0f;
                } else {
#pragma unroll
                    for (int ft = 0; ft < kRbTilesIn; ++ft)
#pragma unroll
                        for (int r = 0; r < 16; ++r) xf[ft][r] = 0.0f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) if (half == 0 && j < a.cin) xf[0][j] = a.feats[row0 + j];
                }
#pragma unroll
                for (int w = 0; w < kRbTilesIn / 2; ++w) {
                    unsigned word = 0u;
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        const float v = xf[2 * w + (i >> 4)][i & 15];
                        m = fmaxf(m, fabsf(v));
                        rb_push_sign(word, v);
                    }
                    msk[w * kThreads + tid] = word;
                }
                m = fmaxf(m, xhalf(m));
                if (a.amax4) rb_amax_publish(a.amax4, m, lane, (int)blockIdx.x * 4 + wave);
                float sc;
                rb_block_scale(m, sc, inv_x);
#pragma unroll
                for (int t = 0; t < kRbTilesIn; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2)
                        SPLIT2_TO(xf[t][r] * sc, xf[t][r + 1] * sc, xH[2 * t + (r >> 3)][(r & 7) >> 1], xL[2 * t + (r >> 3)][(r & 7) >> 1]);
                }
            }
            // ---- 2. G1: net = W_0 relu(x) + b_0, signs only ----
            unsigned word = 0u;
            float m_n = 0.0f;
#pragma unroll 1
            for (int t = 0; t < kRbTilesIn; ++t) {
                P0 = zero16(); P1 = zero16();
                rb_tile_b<kRbCSteps, kSyncStep16, 0, kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) { unsigned h, l; relu_hilo(xH[g][w], xL[g][w], h, l); H[w] = h; L[w] = l; }
                    }, chunk_sync, issue_piece, nothing);
                pipe.advance();
                rb_tile_b<kRbCSteps, kSyncStep16, 1, kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) { unsigned h, l; relu_hilo(xH[kRbCSteps + g][w], xL[kRbCSteps + g][w], h, l); H[w] = h; L[w] = l; }
                    }, chunk_sync, issue_piece, nothing);
                pipe.advance();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(b0_s + 32 * t + 8 * q + 4 * half);
                    f32x4 nv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        nv[j] = (P0[4 * q + j] + P1[4 * q + j]) * inv_x + b4[j];      // the forward's expression for net
                        rb_push_sign(word, nv[j]);
                        m_n = fmaxf(m_n, fabsf(nv[j]));
                    }
                    if (a.net_out && valid) *reinterpret_cast<f32x4*>(a.net_out + (pt0 * kRbWsRow + row_off + 32u * t + 8 * q)) = nv;
                }
                if (t & 1) { msk[((kRbTilesIn / 2) + (t >> 1)) * kThreads + tid] = word; word = 0u; }
            }
            if (a.amax4 && a.net_out) rb_amax_publish(a.amax4 + 3 * E3DGE_AMAX_FLOATS, m_n, lane, (int)blockIdx.x * 4 + wave);
        }

        // ---- 3. d out resident: register 4q + j of tile T = column 32 T + 8 q + 4 half + j of [d alpha | d beta] ----
        u32x4 yH[2 * kRbTilesOut], yL[2 * kRbTilesOut];
        float inv_y;
        {
            const float* __restrict__ ya = a.d_alpha + gpt * kWidth + 4 * half;
            const float* __restrict__ yb = a.d_beta + gpt * kWidth + 4 * half;
            float m = 0.0f;
#pragma unroll
            for (int T = 0; T < kRbTilesOut; ++T) {                              // the point's largest |d out| first (values not kept)
                const float* __restrict__ s = (T < 8) ? ya + 32 * T : yb + 32 * (T - 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(s + 8 * q);
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
                }
            }
            m = fmaxf(m, xhalf(m));
            if (a.amax4) rb_amax_publish(a.amax4 + E3DGE_AMAX_FLOATS, m, lane, (int)blockIdx.x * 4 + wave);
            float sc_y;
            rb_block_scale(m, sc_y, inv_y);
#pragma unroll
            for (int T = 0; T < kRbTilesOut; ++T) {
                const float* __restrict__ s = (T < 8) ? ya + 32 * T : yb + 32 * (T - 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(s + 8 * q);
                    const int r = 4 * q;
                    SPLIT2_TO(w[0] * sc_y, w[1] * sc_y, yH[2 * T + (r >> 3)][(r & 7) >> 1], yL[2 * T + (r >> 3)][(r & 7) >> 1]);
                    SPLIT2_TO(w[2] * sc_y, w[3] * sc_y, yH[2 * T + (r >> 3)][((r + 2) & 7) >> 1], yL[2 * T + (r >> 3)][((r + 2) & 7) >> 1]);
                }
            }
        }
        // one output tile of a K = 512 contraction against d out: four chunks, 32 of their 40 k-steps
        auto dout_tile = [&]() {
            P0 = zero16(); P1 = zero16();
            rb_tile_b<kRbCSteps, kSyncStep16, 0, kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = yH[g]; L = yL[g]; }, chunk_sync, issue_piece, nothing);
            pipe.advance();
            rb_tile_b<kRbCSteps, kSyncStep16, 1, kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = yH[kRbCSteps + g]; L = yL[kRbCSteps + g]; }, chunk_sync, issue_piece, nothing);
            pipe.advance();
            rb_tile_b<kRbCSteps, kSyncStep16, 2, kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = yH[2 * kRbCSteps + g]; L = yL[2 * kRbCSteps + g]; }, chunk_sync, issue_piece, nothing);
            pipe.advance();
            rb_tile_b<kRbCSteps, kSyncStep16, 3, 2 * kRbTilesOut - 3 * kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                [&](int g, u32x4& H, u32x4& L) { H = yH[3 * kRbCSteps + g]; L = yL[3 * kRbCSteps + g]; }, chunk_sync, issue_piece, nothing);
            pipe.advance();
        };
        // ---- 4. G2: d net = (W_1^T d out) [net > 0] -> workspace; its largest magnitude for the operand scale of G3 ----
        float m_d = 0.0f;
#pragma unroll 1
        for (int t = 0; t < kRbTilesIn; ++t) {
            dout_tile();
            const unsigned word = msk[((kRbTilesIn / 2) + (t >> 1)) * kThreads + tid];
            float* __restrict__ dst = ws_net + (row_off + 32u * t);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o4[j] = rb_keep_if(word, 16 * (t & 1) + 4 * q + j, (P0[4 * q + j] + P1[4 * q + j]) * inv_y);
                    m_d = fmaxf(m_d, fabsf(o4[j]));
                }
                if (valid) *reinterpret_cast<f32x4*>(dst + 8 * q) = o4;
            }
        }
        if (a.amax4) rb_amax_publish(a.amax4 + 2 * E3DGE_AMAX_FLOATS, m_d, lane, (int)blockIdx.x * 4 + wave);
        // ---- 5. G4: W_s^T d out -> second workspace ----
#pragma unroll 1
        for (int t = 0; t < kRbTilesIn; ++t) {
            dout_tile();
            float* __restrict__ dst = ws_short + (row_off + 32u * t);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (P0[4 * q + j] + P1[4 * q + j]) * inv_y;
                if (valid) *reinterpret_cast<f32x4*>(dst + 8 * q) = o4;
            }
        }

        // ---- 6. G3: d x = (W_0^T d net) [x > 0] + short ----
        {
            // `short` of tile t by LDS-DMA into buffer t % 3 of this wave, two tiles ahead of its use: piece q = this lane's quad q
            const uint32_t sh_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)(smem + kRbBLdsShort) + (uint32_t)wave * (3u * 4096u);
            const unsigned sh_voff = row_off * 4u;
            auto short_piece = [&](int t, int q) {
                const uint32_t dst = sh_lds + (uint32_t)((t % 3) * 4096 + q * 1024);
                glds16_saddr<0>(ws_short, sh_voff + (unsigned)((32 * t + 8 * q) * 4), (uint32_t)__builtin_amdgcn_readfirstlane((int)dst));
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) short_piece(0, q);
#pragma unroll
            for (int q = 0; q < 4; ++q) short_piece(1, q);
            m_d = fmaxf(m_d, xhalf(m_d));
            float sc_d, inv_d;
            rb_block_scale(m_d, sc_d, inv_d);
            u32x4 dH[kRbStepsIn], dL[kRbStepsIn];
            {
                const float* __restrict__ src = ws_net + row_off;
#pragma unroll
                for (int T = 0; T < kRbTilesIn; ++T) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(src + 32 * T + 8 * q);
                        const int r = 4 * q;
                        SPLIT2_TO(w[0] * sc_d, w[1] * sc_d, dH[2 * T + (r >> 3)][(r & 7) >> 1], dL[2 * T + (r >> 3)][(r & 7) >> 1]);
                        SPLIT2_TO(w[2] * sc_d, w[3] * sc_d, dH[2 * T + (r >> 3)][((r + 2) & 7) >> 1], dL[2 * T + (r >> 3)][((r + 2) & 7) >> 1]);
                    }
                }
            }
            // the two prologue pieces are older than the d net loads, whose data has arrived: they have landed.  (Made explicit: nothing else
            // orders an LDS-DMA piece against the epilogue's ds_read one tile later; from tile 2 on the pipe's counted waits do, see short_piece.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float* __restrict__ out_row = a.d_feats + gpt * a.cin + 4 * half;
            const f32x4* const sh_rd = reinterpret_cast<const f32x4*>(smem + kRbBLdsShort + wave * (3 * 1024)) + lane;
#pragma unroll 1
            for (int t = 0; t < kRbTilesIn; ++t) {
                P0 = zero16(); P1 = zero16();
                rb_tile_b<kRbCSteps, kSyncStep16, 0, kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) { H = dH[g]; L = dL[g]; }, chunk_sync, issue_piece,
                    [&](int g) { if (g >= 1 && g <= 4 && t + 2 < kRbTilesIn) short_piece(t + 2, g - 1); });
                pipe.advance();
                rb_tile_b<kRbCSteps, kSyncStep16, 1, kRbCSteps>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) { H = dH[kRbCSteps + g]; L = dL[kRbCSteps + g]; }, chunk_sync, issue_piece, nothing);
                pipe.advance();
                const unsigned word = msk[(t >> 1) * kThreads + tid];
                const int f0 = 32 * t + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 sh = sh_rd[((t % 3) * 4 + q) * 64];
                    F4u o;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o.v[j] = rb_keep_if(word, 16 * (t & 1) + 4 * q + j, (P0[4 * q + j] + P1[4 * q + j]) * inv_d) + sh[j];
                    if (valid) {
                        if (f0 + 8 * q + 3 < a.cin) *reinterpret_cast<F4u*>(out_row + 32 * t + 8 * q) = o;
                        else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) if (f0 + 8 * q + j < a.cin) out_row[32 * t + 8 * q + j] = o.v[j];
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
