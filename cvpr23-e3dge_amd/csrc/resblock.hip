// Local-feature -> texture-FiLM head: per point
//     out = W_s x + W_1 relu(W_0 relu(x) + b_0) + b_1 ,   x in R^Cin (Cin <= 320; the released models use 301),
//     (alpha, beta) = split(out, 256)
// Reference: ResnetBlockFC.forward (project/models/helper_modules/resnetfc.py:49-58) as
// netLocal.local_feat_to_tex_modulations_linear (vendor/pifu/lib/model/HGPIFuGANNetResidualInputResnetFC.py:84-93), called
// from SirenLocalGlobal.forward_backbone (project/utils/volume_renderer.py:327-336) on (B,H,W,S,301) local features.
//
// Same machine as the SIREN kernels: 4 waves x 32 points per workgroup, a point's whole state in one lane pair, split-f16
// MFMA contractions (hi + lo, three products, fp32 accumulate) with per-point power-of-two block scaling of the B operand
// (the inputs are unbounded), weights streamed L2 -> LDS in 20-KiB chunks (32 output rows x 160 k: K = 320, Cin padded, is
// two equal chunks of ten k-steps) through a ring of kRbNBuf LDS buffers (RbPipe below).  With one wave per SIMD nothing
// else covers a chunk that has not landed, so a chunk's DMA is issued kRbNBuf - 1 chunks before it is read and the chunk
// barrier waits with a COUNTED vmcnt that leaves the younger chunks in flight.  x and r = relu(net) are kept as packed
// (hi, lo) words (160 registers each); relu(x) is formed from x's words on the fly per k-step (packed integer ops).  Neither
// the (P, 301) hidden activations nor r ever touch memory; the output goes straight to (alpha, beta).  DESIGN.md 4.7 has the
// measurements behind every choice here (tools/rb_trace.py, profiles/r2_pmc_issue_texhead.txt).
#include "siren_common.h"
#include <type_traits>

namespace e3dge {

constexpr int kRbKin = 320;               // padded input / hidden width
constexpr int kRbTilesIn = kRbKin / 32;   // 10
constexpr int kRbStepsIn = kRbKin / 16;   // 20 k-steps of 16
constexpr int kRbOut = 512;
constexpr int kRbTilesOut = kRbOut / 32;  // 16
constexpr int kRbCSteps = kRbStepsIn / 2;                   // 10 k-steps per chunk
constexpr int kRbChunkFloats = kRbCSteps * 2 * 64 * 4;      // [k-step][hi|lo][lane][4 words] = 5120 floats = 20 KiB
constexpr int kRbPieces = kRbChunkFloats * 4 / (4 * 64 * 16);   // 5 LDS-DMA pieces (16 B per lane) per wave and chunk
#ifndef E3DGE_RB_NBUF
#define E3DGE_RB_NBUF 5
#endif
constexpr int kRbNBuf = E3DGE_RB_NBUF;                      // LDS weight buffers
#ifndef E3DGE_RB_RING
#define E3DGE_RB_RING 2
#endif
constexpr int kRbRing = E3DGE_RB_RING;                      // k-steps of (hi, lo) weight fragments held: 1 consumed + the rest in flight
constexpr int kRbChunksG1 = kRbTilesIn * 2;                 // W_0: per out tile chunk A (k 0..159), chunk B (k 160..319)
constexpr int kRbChunksG2 = kRbTilesOut * 4;                // per out tile: W_s A, W_s B, W_1 A, W_1 B
constexpr int kRbChunks = kRbChunksG1 + kRbChunksG2;        // 84
constexpr int64_t kRbOffBias0 = (int64_t)kRbChunks * kRbChunkFloats;  // b_0 [320]
constexpr int64_t kRbOffBias1 = kRbOffBias0 + kRbKin;                 // b_1 [512]
constexpr int64_t kRbOffAux = kRbOffBias1 + kRbOut;                   // [0] max_n ||W_0[n,:]||_2, [1] max |b_0|, [2..3] pad
constexpr int64_t kRbPackedFloats = kRbOffAux + 4;

constexpr int kRbLdsW = 0;
constexpr int kRbLdsB = kRbLdsW + kRbNBuf * kRbChunkFloats;           // b_0 [320], b_1 [512]
constexpr int kRbLdsFloats = kRbLdsB + kRbKin + kRbOut;
constexpr int kRbLdsBytes = kRbLdsFloats * 4;
static_assert(kRbLdsBytes <= 160 * 1024, "LDS budget");
// FILM form (the head writes h' = (alpha + 1) h8 + beta straight into a layer-7 record, see resblock_kernel): per wave a 4-KiB stash of
// the alpha tile ([q][lane] x 16 B) and two 4-KiB buffers of the h8 entries of the current / next feature block ([hl][s][lane] x 16 B)
constexpr int kRbLdsStash = ((kRbLdsFloats + 3) / 4) * 4;
constexpr int kRbLdsH8 = kRbLdsStash + 4 * 1024;                      // floats: 4 waves x 4 KiB
constexpr int kRbLdsFilmFloats = kRbLdsH8 + 4 * 2 * 1024;             // 4 waves x 2 buffers x 4 KiB
constexpr int kRbLdsFilmBytes = kRbLdsFilmFloats * 4;
static_assert(kRbLdsFilmBytes <= 160 * 1024, "LDS budget (FILM)");
static_assert(kRbNBuf >= 3 && kRbPieces == 5, "chunk pipeline shape");

// ---------------------------------------------------------------------------------------------------------------
// weight image: [chunk][10 k-steps][hi|lo][64 lanes][4 words of two f16], value kW16Scale * W[n][k] with
// n = 32 t + (lane & 31), k = 160 * half_chunk + (k-slot order of kOffBig16); rows / columns beyond the real sizes are 0
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
            const int chunk = (int)(r / kRbCSteps);
            const int g = (int)(r - (int64_t)chunk * kRbCSteps);
            int t, kc, which;                         // which: 0 = W_0, 1 = W_s, 2 = W_1
            if (chunk < kRbChunksG1) { t = chunk >> 1; kc = chunk & 1; which = 0; }
            else {      // output tiles are streamed alpha_0, beta_0, alpha_1, beta_1, ...: stream position t' -> row tile 8 (t' & 1) + (t' >> 1)
                const int c2 = chunk - kRbChunksG1, tp = c2 >> 2;
                t = 8 * (tp & 1) + (tp >> 1); kc = c2 & 1; which = 1 + ((c2 >> 1) & 1);
            }
            const int n = 32 * t + (lane & 31);
            unsigned word = 0;
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j = 2 * k + e2;
                const int gg = kRbCSteps * kc + g;     // k-step of the whole contraction: two per 32-feature tile of x / r
                const int kk = 32 * (gg >> 1) + 16 * (gg & 1) + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
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

// The weight-chunk pipeline of this kernel (cf. ChunkPipe in siren_common.h, which the 8-wave SIREN kernels use with three
// 32-KiB buffers and a plain vmcnt(0)).  Chunk c lives in buffer c % kRbNBuf; wave w copies bytes [5 w, 5 w + 5) KiB of a
// chunk in five 1-KiB LDS-DMA pieces.  sync(), early in chunk c: every wave waits until ITS pieces of chunk c+1 have landed
// -- in-order completion: at most the 5 (kRbNBuf - 3) pieces of the chunks after it may still be outstanding; output stores
// and x loads in the queue only make the wait stricter, never laxer --, the barrier publishes chunk c+1 and proves that
// everybody has left chunk c-1, whose buffer the DMA of chunk c + kRbNBuf - 1 may now overwrite.  Chunk indices wrap
// after `count`; chunks fetched past the end of the work land in buffers nobody reads (the kernel ends with vmcnt(0)).
struct RbPipe {
    const char* img;          // image + this wave's 5-KiB slice (wave-uniform)
    const char* src;          // chunk being issued
    uint32_t voff;            // lane * 16
    uint32_t lds_base;        // LDS byte address of wbuf + this wave's slice
    uint32_t lds_dst;         // ... of the buffer being filled
    int idx, count, buf, use_buf;
    float* wbuf;
    const float* wcur;
    const float* wnxt;
    static constexpr int kSliceBytes = kRbPieces * 1024;
    __device__ __forceinline__ void init(float* wbuf_, const float* image, int wave, int lane, int count_) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        wbuf = wbuf_;
        img = reinterpret_cast<const char*>(image) + wave_u * kSliceBytes;
        voff = (uint32_t)lane * 16u;
        lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)wbuf_ + (uint32_t)wave_u * kSliceBytes;
        count = count_;
        idx = 0; buf = 0; use_buf = 0;
        src = img;
        lds_dst = lds_base;
        wcur = wbuf_; wnxt = wbuf_ + kRbChunkFloats;
    }
    __device__ __forceinline__ void issue_piece(int i) {     // i is a compile-time constant at every call site
        switch (i) {
            case 0: glds16_saddr<0>(src, voff, lds_dst); break;
            case 1: glds16_saddr<1024>(src, voff, lds_dst); break;
            case 2: glds16_saddr<2048>(src, voff, lds_dst); break;
            case 3: glds16_saddr<3072>(src, voff, lds_dst); break;
            default: glds16_saddr<0>(src + 4096, voff, lds_dst + 4096u); break;     // the immediate is 13-bit signed
        }
        if (i == kRbPieces - 1) {
            idx = (idx + 1 == count) ? 0 : idx + 1;
            src = img + (size_t)idx * (kRbChunkFloats * 4);
            buf = (buf + 1 == kRbNBuf) ? 0 : buf + 1;
            lds_dst = lds_base + (uint32_t)buf * (kRbChunkFloats * 4);
        }
    }
    __device__ __forceinline__ void prime() {
        for (int c = 0; c < kRbNBuf - 1; ++c)
#pragma unroll
            for (int i = 0; i < kRbPieces; ++i) issue_piece(i);
    }
    __device__ __forceinline__ void sync() {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kRbPieces * (kRbNBuf - 3)) : "memory");
        __syncthreads();
    }
    __device__ __forceinline__ void advance() {
        use_buf = (use_buf + 1 == kRbNBuf) ? 0 : use_buf + 1;
        wcur = wnxt;
        wnxt = wbuf + ((use_buf + 1 == kRbNBuf) ? 0 : use_buf + 1) * kRbChunkFloats;
    }
};

// one weight chunk against its KSTEPS k-steps of a B operand produced by `opnd(g, H, L)`; the same operand ring as
// big_tile_f16 (siren_common.h), with the chunk barrier after k-step SYNC and the DMA pieces of the chunk kRbNBuf - 1
// ahead handed out one per k-step after it
// POS is the chunk's position inside its output tile (tiles have an even number of chunks): ten k-steps per chunk do not
// divide a ring of four, so the slot of k-step g is (g + 10 POS) % kRbRing -- a compile-time rotation
template <int KSTEPS, int SYNC, int POS, class Opnd, class Sync, class Dma, class Step>
__device__ __forceinline__ void rb_tile(const float* __restrict__ wchunk, const float* __restrict__ wnext, int lane,
                                        f32x16& acc, f32x16& accb, u32x4 (&ringH)[kRbRing], u32x4 (&ringL)[kRbRing],
                                        Opnd&& opnd, Sync&& sync, Dma&& dma, Step&& step) {
    const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(wchunk) + lane;
    const u32x4* __restrict__ wn = reinterpret_cast<const u32x4*>(wnext) + lane;
    constexpr int PH = (POS * KSTEPS) % kRbRing;
    static_assert((2 * KSTEPS) % kRbRing == 0, "ring rotation must close over a pair of chunks");
    static_assert(KSTEPS - SYNC - 1 >= kRbPieces, "tile too short");
#pragma unroll
    for (int g = 0; g < KSTEPS; ++g) {
        step(POS * KSTEPS + g);
        const int ga = g + kRbRing - 1;
        // the next chunk may only be touched after this tile's barrier
        static_assert(KSTEPS - (kRbRing - 1) > SYNC, "ring would read the next chunk before the barrier");
        ringH[(ga + PH) % kRbRing] = (ga < KSTEPS) ? wp[(ga * 2 + 0) * 64] : wn[((ga - KSTEPS) * 2 + 0) * 64];
        ringL[(ga + PH) % kRbRing] = (ga < KSTEPS) ? wp[(ga * 2 + 1) * 64] : wn[((ga - KSTEPS) * 2 + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 wh = ringH[(g + PH) % kRbRing], wl = ringL[(g + PH) % kRbRing];
        u32x4 bH, bL;
        opnd(g, bH, bL);
        f32x16& x0 = (g & 1) ? accb : acc;
        x0 = mfma16(wh, bH, x0);
        f32x16& x1 = (g & 1) ? acc : accb;
        x1 = mfma16(wl, bH, x1);
        x0 = mfma16(wh, bL, x0);
        if (g == SYNC) sync();
        if (g > SYNC && g - SYNC - 1 < kRbPieces) dma(g - SYNC - 1);     // (all five right after the barrier: same time)
    }
}

struct ResblockK {
    const float* packed;
    const float* feats;     // (n_pts, cin)
    float* alpha;           // (n_pts, 256)
    float* beta;            // (n_pts, 256)
    long long n_pts;
    int cin, subtiles_per_wg;
    // FILM form: the render launch's layer-7 record (siren16.h, CACHE = 1) in, the FiLM-ed record out, and the render's tiling
    const unsigned char* bb_in; unsigned char* bb_out;
    int S, R, HW, tiles_per_img, bb_subs;
};

// relu on a packed (hi, lo) pair of two f16 values each: both halves are cleared where hi is negative.  Three VALU ops a
// word: packed integer max against 0 on the bit patterns (negative f16 <=> negative int16), packed arithmetic shift for the
// sign mask, and-not.  The first two are volatile asm ON PURPOSE: the ten output tiles of phase 2 are unrolled and apply this
// to the same x words, and as plain C++ the compiler keeps the results of the first tile alive for the others -- 160 more
// live registers, i.e. spills whose reloads wait for all weight DMA in flight.
__device__ __forceinline__ void relu_hilo(unsigned h, unsigned l, unsigned& rh, unsigned& rl) {
    unsigned neg;
    asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(rh) : "v"(h));
    asm volatile("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(neg) : "v"(h));
    rl = l & ~neg;
}

#ifdef E3DGE_RB_TRACE
// (tag, s_memtime) pairs of one wave (tools/rb_trace.py): tag = 1000 * phase + 100 * what + index
__device__ unsigned long long g_rb_trace[640];
#if E3DGE_RB_TRACE > 1
#define RB_STAMP(tag) do { if (tr_on) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); \
                                        if ((tid_k & 63) == 0 && tr_i < 320) { g_rb_trace[2 * tr_i] = (tag); g_rb_trace[2 * tr_i + 1] = t_; } \
                                        ++tr_i; } } while (0)
#else       // coarse: the four phase boundaries only, held in SGPRs until the sub-tile ends (leaves the register allocation alone)
#define RB_STAMP(tag) do { if ((tag) >= 1000) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_ts[(tag) / 1000 - 1])); \
                             if ((tag) == 4000 && tr_sub && (tid_k & 63) == 0) {                                                    \
                                 for (int i_ = 0; i_ < 4; ++i_) { g_rb_trace[2 * i_] = 1000 * (i_ + 1); g_rb_trace[2 * i_ + 1] = tr_ts[i_]; } \
                                 g_rb_trace[9] = 0; } } } while (0)
#endif
#ifndef E3DGE_RB_TRACE_T2
#define E3DGE_RB_TRACE_T2 1
#endif
#ifndef E3DGE_RB_TRACE_T3
#define E3DGE_RB_TRACE_T3 5
#endif
#else
#define RB_STAMP(tag) do { } while (0)
#endif

struct __attribute__((packed, aligned(4))) F4u { float v[4]; };        // 16-byte load at 4-byte alignment

// FILM = false: (alpha, beta) -> global memory.  FILM = true (inference, second pass of an evaluated image): the head also applies
// them -- h' = (alpha + 1) h8 + beta on the layer-7 output h8 that render pass #1 left in its record (same operation order and
// rounding as siren16_kernel's FiLM step, volume_renderer.py:217-220) -- and writes h' as a record of the same layout, which pass #2
// reads INSTEAD of record + alpha + beta: (alpha, beta) never reach HBM (2 KiB of the 3 KiB per point that made pass #2 HBM-bound).
// Output tiles are streamed alpha_i, beta_i, ...: the alpha tile waits in LDS (each lane's own 4 x 16 B) for its beta tile; the
// h8 entries of feature block i (four 16-byte entries per lane: [hi | lo] x [two lane groups of the 16x16x32 B-operand layout]) come
// in by LDS-DMA two tiles ahead -- a plain global load inside the tile loop would put an in-order vmcnt wait in front of its
// use and drain the weight pipe; DMA pieces are covered by the pipe's own counted wait, which runs at least three
// chunks behind their issue.
template <bool FILM>
__global__ void __launch_bounds__(kThreads) resblock_kernel(const ResblockK a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kRbLdsW;
    float* const b0_s = smem + kRbLdsB;
    float* const b1_s = b0_s + kRbKin;

    const int tid_k = threadIdx.x;
    const long long pt0 = (long long)blockIdx.x * a.subtiles_per_wg * kTilePts;
    const long long rem = a.n_pts - pt0;
    const int npts = (int)(rem < (long long)a.subtiles_per_wg * kTilePts ? rem : (long long)a.subtiles_per_wg * kTilePts);
    const int n_sub = (npts + kTilePts - 1) / kTilePts;
    const float* __restrict__ packed = a.packed;
    float* const out_a = a.alpha + pt0 * kWidth;            // wave-uniform bases of this workgroup's rows
    float* const out_b = a.beta + pt0 * kWidth;
    for (int i = tid_k; i < kRbKin + kRbOut; i += kThreads) b0_s[i] = packed[kRbOffBias0 + i];

    RbPipe pipe;
    pipe.init(wbuf, packed, tid_k >> 6, tid_k & 63, kRbChunks);
    pipe.prime();
    auto issue_piece = [&](int i) { pipe.issue_piece(i); };
#ifdef E3DGE_RB_TRACE
    bool tr_on = false, tr_sub = false;
    int tr_i = 0;
    unsigned long long tr_ts[4] = {0, 0, 0, 0};
    auto chunk_sync = [&]() {
        RB_STAMP(100);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kRbPieces * (kRbNBuf - 3)) : "memory");
        RB_STAMP(200);
        __syncthreads();
        RB_STAMP(300);
    };
    auto kstep = [&](int g) { RB_STAMP(g); };
#else
    auto chunk_sync = [&]() { pipe.sync(); };
    auto kstep = [](int) {};
#endif
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
        const int pc = valid ? p : npts - 1;                 // padded lanes shadow the last valid point
        const long long gpt = pt0 + pc;
        // output position as a 32-bit offset from the workgroup's first row: a 64-bit per-lane address held across the
        // tile loop gets spilled, and its reload waits for every weight DMA in flight
        const unsigned out_off = (unsigned)pc * kWidth + 4u * half;
        // FILM: byte offset of this lane's first record entry (feature block 0, hi, lane group s = 0): the render kernel's point
        // order is [image][tile of R rays][sample], a tile's points in sub-tiles of 128 = 8 slabs of 16 (siren16.h)
        unsigned rec_b = 0;
        uint32_t h8_lds = 0;
        f32x4* stash = nullptr;
        if (FILM) {
            const unsigned gp = (unsigned)gpt, per_img = (unsigned)a.HW * (unsigned)a.S;
            const unsigned b = gp / per_img, in_img = gp - b * per_img;
            const unsigned ray = in_img / (unsigned)a.S, tile = ray / (unsigned)a.R;
            const unsigned pin = in_img - tile * (unsigned)a.R * (unsigned)a.S;
            const unsigned slab = ((b * (unsigned)a.tiles_per_img + tile) * (unsigned)a.bb_subs + (pin >> 7)) * 8u + ((pin >> 4) & 7u);
            rec_b = (slab * 1024u + (pin & 15u) + 16u * (unsigned)half) * 16u;
            h8_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)(smem + kRbLdsH8) + (uint32_t)wave * 8192u;
            stash = reinterpret_cast<f32x4*>(smem + kRbLdsStash + wave * 1024) + lane;
        }

#ifdef E3DGE_RB_TRACE
        tr_sub = blockIdx.x == 7 && (tid_k >> 6) == 0 && sub == 1;
        tr_on = tr_sub;
        if (tr_on) tr_i = 0;
        RB_STAMP(1000);
#endif
        // ---- 1. x: each lane keeps its point's values; then (hi, lo) ----
        u32x4 xH[kRbStepsIn], xL[kRbStepsIn];
        float inv_x, xnorm;
        {
            f32x16 xf[kRbTilesIn];
            float m = 0.0f, ss = 0.0f;
            // this lane's point, straight from global memory: register r = 4q + j of tile ft is feature 32 ft + 8 q + 4 half + j,
            // i.e. one 16-byte load per (tile, q) at 4-byte alignment (rows are cin floats).  Branch-free on purpose: all 40
            // loads of the sub-tile are in flight together.  (Loads under a per-quad `if (quad inside the row)` each got their
            // own vmcnt(0) -- 40 serialized round trips that also drained the weight DMA: 59 of the 177 kcycles of a sub-tile.
            // The first version staged 32-feature slices through LDS: ten load -> barrier -> read rounds.)  A quad that sticks
            // out of its row reads into the next row and is zeroed afterwards; only at the very end of the tensor would that
            // leave the buffer, so the start of every read is clamped to the last legal one and the (at most one) wave that
            // was clamped shifts its elements back into place.  Padded lanes read the last valid point.
            const long long total = a.n_pts * (long long)a.cin;
            const long long row0 = gpt * a.cin;
            if (total >= 4) {
                const long long sl = total - 4 - row0;                      // last legal start, relative to this row
                const int slack = (int)(sl < 4096 ? sl : 4096);
                const float* __restrict__ xr = a.feats + row0;
#pragma unroll
                for (int ft = 0; ft < kRbTilesIn; ++ft) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 32 * ft + 8 * q + 4 * half;          // first feature of this quad
                        const F4u v = *reinterpret_cast<const F4u*>(xr + min(f0, slack));
#pragma unroll
                        for (int j = 0; j < 4; ++j) xf[ft][4 * q + j] = v.v[j];
                    }
                }
                if (__builtin_amdgcn_ballot_w64(slack < kRbKin - 4)) {      // wave-uniform, true for the tensor's last rows only
#pragma unroll
                    for (int ft = 0; ft < kRbTilesIn; ++ft) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int f0 = 32 * ft + 8 * q + 4 * half;
                            const int d = f0 - min(f0, slack);              // the read started d elements early
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
                const int lim = a.cin - 4 * half;                           // feature 32 ft + 8 q + j is real iff below lim
#pragma unroll
                for (int ft = 0; ft < kRbTilesIn; ++ft)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xf[ft][r] = (32 * ft + 8 * (r >> 2) + (r & 3) < lim) ? xf[ft][r] : 0.0f;
            } else {                                                        // fewer than four floats in the whole tensor
#pragma unroll
                for (int ft = 0; ft < kRbTilesIn; ++ft)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xf[ft][r] = 0.0f;
#pragma unroll
                for (int j = 0; j < 3; ++j) if (half == 0 && j < a.cin) xf[0][j] = a.feats[row0 + j];
            }
#pragma unroll
            for (int ft = 0; ft < kRbTilesIn; ++ft) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { m = fmaxf(m, fabsf(xf[ft][r])); ss = fmaf(xf[ft][r], xf[ft][r], ss); }
                asm volatile("" : "+a"(xf[ft]));
            }
            m = fmaxf(m, xhalf(m));
            ss += xhalf(ss);
            xnorm = sqrtf(ss);
            const unsigned e = min((__float_as_uint(m) >> 23) & 255u, 254u);
            const float sc = __uint_as_float((254u - e) << 23);
            inv_x = __uint_as_float((e > 8u ? e - 7u : 1u) << 23);
#pragma unroll
            for (int t = 0; t < kRbTilesIn; ++t) {
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    SPLIT2_TO(xf[t][r] * sc, xf[t][r + 1] * sc, xH[2 * t + (r >> 3)][(r & 7) >> 1], xL[2 * t + (r >> 3)][(r & 7) >> 1]);
                // x is used by VALU code (the relu of phase 2): keep its words in the VGPR half, next to the weight ring and
                // the epilogue temporaries; the r words (160) and the accumulators take the AGPR half
                asm volatile("" : "+v"(xH[2 * t]), "+v"(xH[2 * t + 1]), "+v"(xL[2 * t]), "+v"(xL[2 * t + 1]));
            }
        }

        RB_STAMP(2000);
#if defined(E3DGE_RB_TRACE) && E3DGE_RB_TRACE > 1
        tr_on = false;
#endif
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
            // Two accumulator pairs (P, Q).  A tile's epilogue -- 170 VALU instructions in phase 2 -- would otherwise run with the
            // matrix pipe idle (one wave per SIMD: nothing else to issue), so it is deferred: tile t accumulates into one pair
            // while the finished pair of tile t-1 is turned into r words in four slices, hooked in front of k-steps 1..4.
            auto r_slice = [&](int T, int q, const f32x16& a0, const f32x16& a1) {         // registers 4q..4q+3 of tile T
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(b0_s + 32 * T + 8 * q + 4 * half);
                float rv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[j] = fmaxf((a0[4 * q + j] + a1[4 * q + j]) * inv_x + b4[j], 0.0f) * sc_r;
                const int r = 4 * q;
                SPLIT2_TO(rv[0], rv[1], rH[2 * T + (r >> 3)][(r & 7) >> 1], rL[2 * T + (r >> 3)][(r & 7) >> 1]);
                SPLIT2_TO(rv[2], rv[3], rH[2 * T + (r >> 3)][((r + 2) & 7) >> 1], rL[2 * T + (r >> 3)][((r + 2) & 7) >> 1]);
            };
            f32x16 P0, P1, Q0, Q1;
#pragma unroll
            for (int t = 0; t < kRbTilesIn; ++t) {
#if defined(E3DGE_RB_TRACE) && E3DGE_RB_TRACE > 1
                tr_on = tr_sub && t == E3DGE_RB_TRACE_T2;
#endif
                f32x16& A0 = (t & 1) ? Q0 : P0;
                f32x16& A1 = (t & 1) ? Q1 : P1;
                const f32x16& B0 = (t & 1) ? P0 : Q0;                       // the pair tile t-1 left behind
                const f32x16& B1 = (t & 1) ? P1 : Q1;
                auto hook = [&](int g) {
                    kstep(g);
                    if (t > 0 && g >= 1 && g <= 4) r_slice(t - 1, g - 1, B0, B1);
                };
                A0 = zero16(); A1 = zero16();
                rb_tile<kRbCSteps, kSyncStep16, 0>(pipe.wcur, pipe.wnxt, lane, A0, A1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) { unsigned h, l; relu_hilo(xH[g][w], xL[g][w], h, l); H[w] = h; L[w] = l; }
                    }, chunk_sync, issue_piece, hook);
                pipe.advance();
                if (t > 0) asm volatile("" : "+a"(rH[2 * (t > 0 ? t - 1 : 0)]), "+a"(rH[2 * (t > 0 ? t - 1 : 0) + 1]),
                                             "+a"(rL[2 * (t > 0 ? t - 1 : 0)]), "+a"(rL[2 * (t > 0 ? t - 1 : 0) + 1]));
                rb_tile<kRbCSteps, kSyncStep16, 1>(pipe.wcur, pipe.wnxt, lane, A0, A1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            unsigned h, l; relu_hilo(xH[kRbCSteps + g][w], xL[kRbCSteps + g][w], h, l); H[w] = h; L[w] = l;
                        }
                    }, chunk_sync, issue_piece, hook);
                pipe.advance();
            }
            RB_STAMP(400);
#pragma unroll
            for (int q = 0; q < 4; ++q) r_slice(kRbTilesIn - 1, q, Q0, Q1);   // the last tile (odd) sat in Q
            asm volatile("" : "+a"(rH[18]), "+a"(rH[19]), "+a"(rL[18]), "+a"(rL[19]));
            RB_STAMP(500);

#ifdef E3DGE_RB_TRACE
            tr_on = tr_sub;
#endif
            RB_STAMP(3000);
            // ---- 3. out = W_s x + W_1 r + b_1 -> alpha (tiles 0..7), beta (tiles 8..15) ----
            // W_s x accumulates in P, W_1 r in Q.  P is folded into `px` in front of k-steps 20..23 (while Q runs); Q is folded,
            // biased and stored in front of k-steps 1..4 of the NEXT tile (while P runs): no k-step waits for an epilogue.
            f32x16 px;
            // registers 4q..4q+3 of stream tile tp = (tp & 1 ? beta : alpha) of feature block tp >> 1 (PAR = tp & 1, compile time)
            auto out_slice = [&](auto par, int tp, int q) {
                constexpr int PAR = decltype(par)::value;
                const int blk = tp >> 1, rt = 8 * PAR + blk;                 // row tile of W_s / W_1 / b_1
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(b1_s + 32 * rt + 8 * q + 4 * half);
                f32x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = px[4 * q + j] + (Q0[4 * q + j] + Q1[4 * q + j]) * inv_r + b4[j];
                if (!FILM) {
                    float* __restrict__ dst = (PAR ? out_b : out_a) + (out_off + 32u * blk);
                    if (valid) *reinterpret_cast<f32x4*>(dst + 8 * q) = o4;
                } else if (!PAR) {
                    stash[q * 64] = o4;                                      // alpha waits for its beta tile (this lane's own slot)
                } else {
                    const f32x4 al = stash[q * 64];
                    // h8 of features 32 blk + 8 q + 4 half + (0..3): entry (g = blk, hl, lane group s = q & 1), words 2 (q >> 1), + 1
                    const unsigned char* hb = reinterpret_cast<const unsigned char*>(smem + kRbLdsH8) + wave * 8192 + (blk & 1) * 4096 + lane * 16 + 8 * (q >> 1);
                    const uint2 wh = *reinterpret_cast<const uint2*>(hb + (0 * 2 + (q & 1)) * 1024);
                    const uint2 wl = *reinterpret_cast<const uint2*>(hb + (1 * 2 + (q & 1)) * 1024);
                    const float h0 = f16lo(wh.x) + f16lo(wl.x), h1 = f16hi(wh.x) + f16hi(wl.x);
                    const float h2 = f16lo(wh.y) + f16lo(wl.y), h3 = f16hi(wh.y) + f16hi(wl.y);
                    // the operation order of siren16_kernel's FiLM step: ((alpha + 1) h) + beta, every step rounded
                    const float y0 = __fadd_rn(__fmul_rn(__fadd_rn(al[0], 1.0f), h0), o4[0]), y1 = __fadd_rn(__fmul_rn(__fadd_rn(al[1], 1.0f), h1), o4[1]);
                    const float y2 = __fadd_rn(__fmul_rn(__fadd_rn(al[2], 1.0f), h2), o4[2]), y3 = __fadd_rn(__fmul_rn(__fadd_rn(al[3], 1.0f), h3), o4[3]);
                    uint2 oh, ol;
                    SPLIT2_TO(y0, y1, oh.x, ol.x);
                    SPLIT2_TO(y2, y3, oh.y, ol.y);
                    if (valid) {
                        unsigned char* ob = a.bb_out + (rec_b + (unsigned)(((blk * 2) * 64 + 32 * (q & 1)) * 16 + 8 * (q >> 1)));
                        *reinterpret_cast<uint2*>(ob) = oh;
                        *reinterpret_cast<uint2*>(ob + 64 * 16) = ol;
                    }
                }
            };
            // FILM: the four h8 entries per lane of feature block blk, by LDS-DMA into buffer blk & 1: piece i = (hl = i >> 1, s = i & 1)
            auto h8_piece = [&](int blk, int i) {
                const uint32_t dst = h8_lds + (uint32_t)((blk & 1) * 4096 + i * 1024);
                glds16_saddr<0>(a.bb_in, rec_b + (unsigned)(((blk * 2 + (i >> 1)) * 64 + 32 * (i & 1)) * 16), (uint32_t)__builtin_amdgcn_readfirstlane((int)dst));
            };
            using ParA = std::integral_constant<int, 0>;
            using ParB = std::integral_constant<int, 1>;
            auto fold_px = [&](int g) {
                if (g >= 20 && g <= 23) {
                    const int q = g - 20;
#pragma unroll
                    for (int j = 0; j < 4; ++j) px[4 * q + j] = (P0[4 * q + j] + P1[4 * q + j]) * inv_x;
                }
            };
            auto run_tile = [&](auto&& hook) {
                P0 = zero16(); P1 = zero16();
                rb_tile<kRbCSteps, kSyncStep16, 0>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) { H = xH[g]; L = xL[g]; }, chunk_sync, issue_piece, hook);
                pipe.advance();
                rb_tile<kRbCSteps, kSyncStep16, 1>(pipe.wcur, pipe.wnxt, lane, P0, P1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) { H = xH[kRbCSteps + g]; L = xL[kRbCSteps + g]; }, chunk_sync, issue_piece, hook);
                pipe.advance();
                RB_STAMP(600);
                Q0 = zero16(); Q1 = zero16();
                rb_tile<kRbCSteps, kSyncStep16, 2>(pipe.wcur, pipe.wnxt, lane, Q0, Q1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) { H = rH[g]; L = rL[g]; }, chunk_sync, issue_piece, hook);
                pipe.advance();
                rb_tile<kRbCSteps, kSyncStep16, 3>(pipe.wcur, pipe.wnxt, lane, Q0, Q1, ringH, ringL,
                    [&](int g, u32x4& H, u32x4& L) { H = rH[kRbCSteps + g]; L = rL[kRbCSteps + g]; }, chunk_sync, issue_piece, hook);
                pipe.advance();
                RB_STAMP(700);
            };
            // W_s x accumulates in P, W_1 r in Q.  P is folded into `px` in front of k-steps 20..23 (while Q runs); Q is folded,
            // biased and stored / stashed / FiLM-ed in front of k-steps 1..4 of the NEXT tile (while P runs): no k-step waits for
            // an epilogue.  The tile loop runs over feature blocks (alpha tile, beta tile) so that a slice's kind is a compile-time fact.
#pragma unroll 1
            for (int blk = 0; blk < kRbTilesOut / 2; ++blk) {
#if defined(E3DGE_RB_TRACE) && E3DGE_RB_TRACE > 1
                tr_on = tr_sub && 2 * blk == E3DGE_RB_TRACE_T3;
#endif
                run_tile([&](int g) {                   // alpha tile of block blk; finishes the beta tile of block blk - 1
                    kstep(g);
                    if (g >= 1 && g <= 4 && blk > 0) out_slice(ParB{}, 2 * blk - 1, g - 1);
                    if (FILM && g >= 11 && g <= 14) h8_piece(blk, g - 11);
                    fold_px(g);
                });
#if defined(E3DGE_RB_TRACE) && E3DGE_RB_TRACE > 1
                tr_on = tr_sub && 2 * blk + 1 == E3DGE_RB_TRACE_T3;
#endif
                run_tile([&](int g) {                   // beta tile of block blk; finishes its alpha tile
                    kstep(g);
                    if (g >= 1 && g <= 4) out_slice(ParA{}, 2 * blk, g - 1);
                    fold_px(g);
                });
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) out_slice(ParB{}, kRbTilesOut - 1, q);
            RB_STAMP(800);
        }
#ifdef E3DGE_RB_TRACE
        tr_on = tr_sub;
#endif
        RB_STAMP(4000);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#include "resblock_bwd.h"

}  // namespace e3dge

using namespace e3dge;

#ifdef E3DGE_RB_TRACE
extern "C" int e3dge_debug_rb_trace(unsigned long long* out640) {
    return hipMemcpyFromSymbol(out640, HIP_SYMBOL(e3dge::g_rb_trace), sizeof(unsigned long long) * 640) == hipSuccess ? 0 : 1;
}
#endif

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

static int launch_resblock(ResblockK k, bool film, hipStream_t st, const char* what) {
    const int lds = film ? kRbLdsFilmBytes : kRbLdsBytes;
    const void* fn = film ? reinterpret_cast<const void*>(&resblock_kernel<true>) : reinterpret_cast<const void*>(&resblock_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);       // per device, cheap: set on every launch
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(resblock): %s", hipGetErrorString(e));
    const int64_t tiles = (k.n_pts + kTilePts - 1) / kTilePts;
    const int spw = pick_subtiles_per_wg(tiles, 1);
    k.subtiles_per_wg = spw;
    const int64_t grid = (tiles + spw - 1) / spw;
    E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "%s: grid too large", what);
    if (film) resblock_kernel<true><<<dim3((unsigned)grid), dim3(kThreads), lds, st>>>(k);
    else resblock_kernel<false><<<dim3((unsigned)grid), dim3(kThreads), lds, st>>>(k);
    return check_launch(what);
}

extern "C" int e3dge_tex_modulations_fwd(const float* packed, const float* feats, int cin, int64_t n_pts,
                                         float* alpha, float* beta, e3dge_stream_t stream) {
    E3DGE_REQUIRE(n_pts >= 0, "tex_modulations_fwd: bad size");
    if (n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(packed && feats && alpha && beta, "tex_modulations_fwd: null pointer");
    E3DGE_REQUIRE(cin >= 1 && cin <= kRbKin, "tex_modulations_fwd: cin=%d outside [1, %d]", cin, kRbKin);
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(alpha) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0,
                  "tex_modulations_fwd: packed/alpha/beta must be 16-B aligned");
    ResblockK k{};
    k.packed = packed; k.feats = feats; k.alpha = alpha; k.beta = beta; k.n_pts = n_pts; k.cin = cin;
    return launch_resblock(k, false, as_stream(stream), "tex_modulations_fwd");
}

extern "C" int64_t e3dge_resblock_bwd_packed_floats(void) { return kRbBPackedFloats; }

extern "C" int e3dge_resblock_bwd_pack_weights(float* packed, const float* w0, const float* b0, const float* w1, const float* ws, int cin,
                                               e3dge_stream_t stream) {
    E3DGE_REQUIRE(packed && w0 && b0 && w1 && ws, "resblock_bwd_pack_weights: null pointer");
    E3DGE_REQUIRE(cin >= 1 && cin <= kRbKin, "resblock_bwd_pack_weights: cin=%d outside [1, %d]", cin, kRbKin);
    resblock_bwd_pack_kernel<<<dim3(1024), dim3(256), 0, as_stream(stream)>>>(packed, w0, b0, w1, ws, cin);
    return check_launch("resblock_bwd_pack_weights");
}

extern "C" int64_t e3dge_tex_modulations_bwd_ws_floats(int64_t n_pts) { return n_pts > 0 ? 2 * n_pts * (int64_t)kRbWsRow : 0; }

extern "C" int e3dge_tex_modulations_bwd(const float* packed_bwd, const float* feats, int cin, int64_t n_pts, const float* d_alpha,
                                         const float* d_beta, float* d_feats, float* ws, float* net_out, float* amax4, e3dge_stream_t stream) {
    E3DGE_REQUIRE(n_pts >= 0, "tex_modulations_bwd: bad size");
    if (n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(packed_bwd && feats && d_alpha && d_beta && d_feats && ws, "tex_modulations_bwd: null pointer");
    E3DGE_REQUIRE(cin >= 1 && cin <= kRbKin, "tex_modulations_bwd: cin=%d outside [1, %d]", cin, kRbKin);
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(packed_bwd) | reinterpret_cast<uintptr_t>(d_alpha) | reinterpret_cast<uintptr_t>(d_beta) | reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(net_out)) & 15) == 0,
                  "tex_modulations_bwd: packed / d_alpha / d_beta / ws / net_out must be 16-B aligned");
    ResblockBwdK k{};
    k.packed = packed_bwd; k.feats = feats; k.d_alpha = d_alpha; k.d_beta = d_beta; k.d_feats = d_feats; k.ws = ws; k.net_out = net_out; k.amax4 = amax4;
    k.n_pts = n_pts; k.cin = cin;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kRbBLdsBytes);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(resblock_bwd): %s", hipGetErrorString(e));
    const int64_t tiles = (n_pts + kTilePts - 1) / kTilePts;
    const int spw = pick_subtiles_per_wg(tiles, 1);
    k.subtiles_per_wg = spw;
    const int64_t grid = (tiles + spw - 1) / spw;
    E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "tex_modulations_bwd: grid too large");
    resblock_bwd_kernel<<<dim3((unsigned)grid), dim3(kThreads), kRbBLdsBytes, st>>>(k);
    return check_launch("tex_modulations_bwd");
}

extern "C" int e3dge_tex_film_fwd(const float* packed, const float* feats, int cin, int batch, int height, int width, int n_samples,
                                  const void* backbone_in, void* backbone_out, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && height > 0 && width > 0 && n_samples > 0, "tex_film_fwd: bad sizes");
    if (batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(packed && feats && backbone_in && backbone_out && backbone_in != backbone_out, "tex_film_fwd: null / aliased pointer");
    E3DGE_REQUIRE(cin >= 1 && cin <= kRbKin, "tex_film_fwd: cin=%d outside [1, %d]", cin, kRbKin);
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(backbone_in) | reinterpret_cast<uintptr_t>(backbone_out)) & 15) == 0,
                  "tex_film_fwd: packed / records must be 16-B aligned");
    const int64_t bytes = e3dge_siren_backbone_bytes(batch, height, width, n_samples);
    const int64_t n_pts = (int64_t)batch * height * width * n_samples;
    E3DGE_REQUIRE(bytes > 0 && bytes < ((int64_t)1 << 32) && n_pts < ((int64_t)1 << 31),
                  "tex_film_fwd: record of %lld bytes (needs 0 < bytes < 4 GiB: 32-bit offsets)", (long long)bytes);
    ResblockK k{};
    k.packed = packed; k.feats = feats; k.n_pts = n_pts; k.cin = cin;
    k.bb_in = static_cast<const unsigned char*>(backbone_in); k.bb_out = static_cast<unsigned char*>(backbone_out);
    k.S = n_samples; k.HW = height * width;
    siren_record_layout(batch, height, width, n_samples, &k.R, &k.tiles_per_img, &k.bb_subs);
    return launch_resblock(k, true, as_stream(stream), "tex_film_fwd");
}
