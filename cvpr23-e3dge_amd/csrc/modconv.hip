// Modulated 3x3 convolutions of the StyleGAN2 up-sampler as implicit GEMMs on the f16 matrix pipe.
//
// Reference being replaced: ModulatedConv2d.forward (project/models/stylesdf_model.py:317-362) as used by StyledConv
// (:469-507): per sample  w' = scale * W * s[ci];  w' *= rsqrt(sum_{ci,k} w'^2 + 1e-8);  F.conv2d(x, w', padding=1, groups=B)
// or, for the up-sampling layers, F.conv_transpose2d(x, w'^T, stride=2) followed by the FIR blur.
//
// What is different here
//   * No per-sample weight tensor.  The modulation commutes with the contraction:
//         out[co] = demod[b,co] * sum_{ci,k} (scale W[co,ci,k]) * (s[b,ci] x[ci, .+k]),
//     so the weights are packed ONCE per layer (e3dge_modconv_pack_weights) into MFMA A-fragment order, the style is
//     applied to the input patch while it is staged into LDS, and demod[b,co] (e3dge_modconv_demod: a GEMV against the
//     per-(co,ci) squared norms) scales the accumulators in the epilogue.
//   * fp32 accuracy on the f16 pipe, as in the SIREN kernels: both operands are split into f16 hi + lo, three products
//     (hi*hi, lo*hi, hi*lo) accumulate in fp32 (v_mfma_f32_32x32x16_f16).  Weights carry a factor 128; activations are
//     scaled per launch by a power of two derived from max|s| * max|x| (device scalars) so that the largest product operand
//     sits below 2^15 -- exact, undone in the epilogue.
//   * GEMM view: M = Co (32-row tiles), N = pixels (32 consecutive x of one row per MFMA column block), K = (tap, ci) with
//     16 input channels per k-step.  The input patch of a tile (+halo) lives in LDS as four planes [hi|lo][k-half][pixel]
//     of 16-byte entries (8 channels), so the B fragment of any tap is one conflict-free ds_read_b128 at a shifted pixel.
//     Weight fragments of the chunk stream L2 -> LDS by LDS-DMA.  Both are double buffered across (tile, chunk) steps of a
//     persistent workgroup, so the next tile's HBM reads overlap this tile's MFMAs.
//   * Stride-1 layers fuse StyledConv's tail (NoiseInjection + FusedLeakyReLU, :459-466, :500-507) into the epilogue.
//     The up-sampling layers run the transposed convolution by output phase (4 + 2 + 2 + 1 taps, no multiplications by the
//     zeros of a zero-insertion) and write the (2H+1)^2 map the FIR blur consumes.
#include "siren_common.h"
#include "decoder_common.h"

namespace e3dge {

constexpr int kMcChunk = 16;                          // input channels per k-step
constexpr int kMcFragBytes = 64 * 16;                 // one A fragment: 64 lanes x 8 f16
constexpr int kMcSlabBytes = 9 * 2 * kMcFragBytes;    // (co-tile, chunk): 9 taps x (hi, lo) = 18,432 B
constexpr int kMcThreads = 512;                       // 8 waves: two per SIMD
constexpr int kMcMaxCi = 1024;                        // style vector staged in LDS
// How far ahead the input patches are loaded (1 or 2 steps), per tile shape: A = 4x32 px, B = 4x64, C = 8x64 / 64 co,
// D = 8x64 / 32 co, TA / TD = the two transposed shapes.  Two steps ahead costs NIT*8 more registers but lets the two wave
// groups run their phases in opposite order (see run_step).
#ifndef E3DGE_MC_AH_A
#define E3DGE_MC_AH_A 1
#endif
#ifndef E3DGE_MC_AH_B
#define E3DGE_MC_AH_B 1
#endif
#ifndef E3DGE_MC_AH_C
#define E3DGE_MC_AH_C 1
#endif
#ifndef E3DGE_MC_AH_D
#define E3DGE_MC_AH_D 1
#endif
#ifndef E3DGE_MC_AH_TA
#define E3DGE_MC_AH_TA 1
#endif
#ifndef E3DGE_MC_AH_TD
#define E3DGE_MC_AH_TD 1
#endif
#ifndef E3DGE_MC_TAPGROUP
#define E3DGE_MC_TAPGROUP 3
#endif
#ifndef E3DGE_MC_SKEW
#define E3DGE_MC_SKEW 1
#endif
constexpr bool kMcSkew = E3DGE_MC_SKEW != 0;      // (with kMcAhead == 2) opposite phase order for waves 0-3 / 4-7              // input patches are loaded this many steps ahead (1 or 2)
#ifdef E3DGE_MC_TIMING
#define MC_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define MC_T(i) do { } while (0)
#endif

struct ModconvK {
    const float* x;          // (B, Ci, H, W)
    const unsigned* wimg;    // [co_tile][chunk][tap][hi|lo][lane][4 words]
    const float* style;      // (B, Ci) modulation s (EqualLinear output)
    const float* demod;      // (B, Co) or null (ToRGB-style, no demodulation)
    const float* in_amax;    // device scalar: max |x| over the whole input
    const float* s_amax;     // (B): max_ci |s[b, ci]|
    const float* noise;      // (noise_batch, OH*OW) or null
    const float* noise_w;    // device scalar
    const float* bias;       // (Co) or null
    float* y;                // (B, Co, OH, OW)
    float* out_amax;         // optional device scalar (atomic max of |y|), bits of a non-negative float
    float slope, act_scale;
    int act;                 // 1: lrelu(v + noise + bias) * act_scale ; 0: plain output
    int B, Ci, Co, H, W, OH, OW;
    int tiles_x, tiles_y, co_blocks, n_tiles, n_chunks, noise_batch;
};


// UP=false: stride-1 3x3, pad 1.   UP=true: stride-2 transposed 3x3 (conv_transpose2d, padding 0), tiled over input
// positions (i, j) in [0, H] x [0, W]; output (2i+ey, 2j+ex).
// A workgroup = 8 waves = TH rows x CG co-groups; a wave owns NCT co-tiles x NPT pixel tiles (32 px) of its row.
// Pipeline over the steps (tile, 16-channel chunk) of a persistent workgroup:
//   step s computes from LDS buffers [s & 1]; at its top the weight DMA of step s+1 and the global loads of the input
//   patch of step s+2 (two steps ahead: HBM latency is longer than one step's MFMAs) are issued; at its end the patch of
//   step s+1, loaded one step earlier, is modulated, split into f16 hi/lo and written to LDS buffer [(s+1) & 1].
template <bool UP, int TH, int NPT, int CG, int NCT, int kMcAhead>
__global__ void __launch_bounds__(kMcThreads) modconv_kernel(const ModconvK a) {
    static_assert(TH * CG == 8, "8 waves");
    constexpr int TW = 32 * NPT;
    constexpr int PH = UP ? TH + 1 : TH + 2, PW = UP ? TW + 1 : TW + 2;      // patch with halo
    constexpr int NPIX = PH * PW;
    constexpr int NCTB = CG * NCT;                                            // co-tiles per workgroup
    constexpr int NPH = UP ? 4 : 1;                                           // output phases per position
    constexpr int XBUF = 4 * NPIX * 16;                                       // bytes of one input buffer
    constexpr int WBUF = NCTB * kMcSlabBytes;
    constexpr int NIT = (2 * NPIX + kMcThreads - 1) / kMcThreads;             // staging items per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_mc[];
    unsigned char* const xbuf = smem_mc;                                      // 2 x XBUF
    unsigned char* const wbuf = smem_mc + 2 * XBUF;                           // 2 x WBUF
    float* const s_lds = reinterpret_cast<float*>(smem_mc + 2 * XBUF + 2 * WBUF);   // [kMcMaxCi] style * operand scale
    float* const e_lds = s_lds + kMcMaxCi;              // [2 tile parities][2][32 * NCTB]: output scale (demod / 128 / sc), bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int wrow = wave % TH, wcg = wave / TH;
    const int HW = a.H * a.W;
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * a.n_chunks;
    if (nsteps <= 0) return;
    const float xmax = amax_read(a.in_amax, lane);

    auto tile_of = [&](int k, int& b, int& cb, int& ty, int& tx) {
        int t = (int)blockIdx.x + k * (int)gridDim.x;
        tx = t % a.tiles_x; t /= a.tiles_x;
        ty = t % a.tiles_y; t /= a.tiles_y;
        cb = t % a.co_blocks; b = t / a.co_blocks;
    };
    // Position of a step: tile k (of this workgroup) and chunk c, with the tile's coordinates.  Three cursors walk the steps
    // (current, +1 for the weight DMA and the LDS stores, +kMcAhead for the global loads) and are ADVANCED, not recomputed:
    // step / n_chunks and the three divisions of tile_of for each of them were ~10 scalar divisions per step -- 13 SALU
    // instructions per MFMA in the counters (profiles/r2_pmc_issue_modconv.txt), all in the issue stream of the same waves.
    struct Pos { int k, c, b, cb, ty, tx; };
    auto pos_at = [&](int step) {
        Pos p;
        p.k = step / a.n_chunks; p.c = step - p.k * a.n_chunks;
        tile_of(p.k, p.b, p.cb, p.ty, p.tx);
        return p;
    };
    auto advance = [&](Pos& p) {
        if (++p.c == a.n_chunks) { p.c = 0; ++p.k; tile_of(p.k, p.b, p.cb, p.ty, p.tx); }
    };

    // ---- staging ----------------------------------------------------------------------------------------------------
    float preg[kMcAhead][NIT][8];
    auto issue_weights = [&](const Pos& ps, int buf) {
        const int c = ps.c, cb = ps.cb;
        // NCTB slabs of 18 KiB = 18 LDS-DMA pieces of 1 KiB each; pieces are dealt round-robin to the 8 waves
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        for (int piece = wave_u; piece < NCTB * 18; piece += 8) {
            const int ct = piece / 18, pc = piece - ct * 18;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(a.wimg) +
                                       ((size_t)(cb * NCTB + ct) * a.n_chunks + c) * kMcSlabBytes + (size_t)pc * 1024;
            const uint32_t dst = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)(wbuf + buf * WBUF) +
                                 (uint32_t)(ct * kMcSlabBytes + pc * 1024);
            // (the address is wave-uniform by construction; readfirstlane makes that explicit for the SGPR operand)
            const uint64_t sa = reinterpret_cast<uint64_t>(src);
            const uint64_t su = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sa >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sa);
            glds16_saddr<0>(reinterpret_cast<const void*>(su), (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)dst));
        }
    };
    // The geometry of a patch item (which pixel of the patch, which global element, inside the image or not) depends on the tile
    // only; a tile has n_chunks steps.  It is computed when a tile's first chunk comes by and kept in NIT registers each for the
    // load and the store side (they run one or two steps apart): with 16-32 chunks per tile on the deep layers the per-step
    // index arithmetic was 15-30 % of a step (E3DGE_MC_TIMING).
    unsigned ld_off[NIT];                 // element offset of the item within a 16-channel block of the image being loaded
    int st_pix[NIT];                      // patch pixel index of the item, or -1: not an item / outside the image (stores zeros)
    int st_h[NIT];
    auto load_input = [&](const Pos& ps, bool first, float (&pr)[NIT][8]) {
        const int c = ps.c, b = ps.b, ty = ps.ty, tx = ps.tx;
        if (c == 0 || first) {
            const int oy = ty * TH - 1, ox = tx * TW - 1;                    // patch origin (same for both variants)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = min(tid + it * kMcThreads, 2 * NPIX - 1);
                const int h = q / NPIX, pix = q - h * NPIX;
                const int prow = pix / PW, pcol = pix - prow * PW;
                const int gy = min(max(oy + prow, 0), a.H - 1), gx = min(max(ox + pcol, 0), a.W - 1);
                // wave-uniform 64-bit base + 32-bit lane offset (16 channels x HW elements < 2^31): no 64-bit address per load
                ld_off[it] = (unsigned)(8 * h * HW + gy * a.W + gx);
            }
        }
        const float* __restrict__ bp = a.x + ((size_t)b * a.Ci + c * kMcChunk) * HW;
        // every thread issues exactly NIT * 8 loads (addresses clamped into the image, out-of-image values zeroed when they
        // are consumed): the end-of-step wait below relies on that count
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) pr[it][j] = bp[ld_off[it] + (unsigned)(j * HW)];
    };
    auto store_input = [&](const Pos& ps, bool first, int buf, const float (&pr)[NIT][8]) {
        const int c = ps.c;
        if (c == 0 || first) {
            const int oy = ps.ty * TH - 1, ox = ps.tx * TW - 1;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + it * kMcThreads;
                const int h = q / NPIX, pix = q - h * NPIX;
                const int prow = pix / PW, pcol = pix - prow * PW;
                const int gy = oy + prow, gx = ox + pcol;
                const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                st_h[it] = h;
                st_pix[it] = (q < 2 * NPIX) ? (inside ? pix : -1 - pix) : (int)0x40000000;       // >= 0 inside, < 0 zero-fill, big: none
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (st_pix[it] != (int)0x40000000) {
                const bool inside = st_pix[it] >= 0;
                const int pix = inside ? st_pix[it] : -1 - st_pix[it];
                const int h = st_h[it];
                const float* sp = s_lds + c * kMcChunk + 8 * h;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
                u32x4 hi, lo;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float sa = w < 2 ? s0[2 * w] : s1[2 * w - 4], sb = w < 2 ? s0[2 * w + 1] : s1[2 * w - 3];
                    const float v0 = inside ? pr[it][2 * w] * sa : 0.0f, v1 = inside ? pr[it][2 * w + 1] * sb : 0.0f;
                    SPLIT2_TO(v0, v1, hi[w], lo[w]);
                }
                unsigned char* base = xbuf + buf * XBUF + (size_t)pix * 16;
                *reinterpret_cast<u32x4*>(base + (0 * 2 + h) * NPIX * 16) = hi;
                *reinterpret_cast<u32x4*>(base + (1 * 2 + h) * NPIX * 16) = lo;
            }
        }
    };
    // s_lds = style[b] * operand scale of sample b (all threads; callers put a barrier before the first reader)
    auto load_style = [&](int b) {
        const float sc = __uint_as_float((268u - scale_exponent(xmax * a.s_amax[b])) << 23);
        for (int i = tid; i < a.Ci; i += kMcThreads) s_lds[i] = a.style[(size_t)b * a.Ci + i] * sc;
    };

    // epilogue constants of tile k into e_lds[k & 1] (threads < 32 * NCTB; a barrier separates this from the tile's epilogue)
    auto load_epilogue_consts = [&](const Pos& ps) {
        const int k = ps.k, b = ps.b, cb = ps.cb;
        if (tid < 32 * NCTB) {
            const int co = cb * NCTB * 32 + tid;
            const float oscale = __uint_as_float((scale_exponent(xmax * a.s_amax[b]) - 21u) << 23);
            float* e = e_lds + (k & 1) * (2 * 32 * NCTB);
            e[tid] = oscale * (a.demod ? a.demod[(size_t)b * a.Co + co] : 1.0f);
            e[32 * NCTB + tid] = (a.act && a.bias) ? a.bias[co] : 0.0f;
        }
    };

    // ---- prologue: step 0 into buffers 0 (and, two steps ahead, step 1 on its way) ----------------------------------
    Pos p_cur = pos_at(0);                               // the step being computed
    Pos p_nx1 = p_cur; advance(p_nx1);                   // step + 1
    Pos p_nxa = p_nx1;                                   // step + kMcAhead
    if (kMcAhead == 2) advance(p_nxa);
    int b_lds = p_cur.b;
    load_style(b_lds);
    load_epilogue_consts(p_cur);
    issue_weights(p_cur, 0);
    load_input(p_cur, true, preg[0]);
    if (kMcAhead == 2 && nsteps > 1) load_input(p_nx1, false, preg[1]);
    __syncthreads();                                   // s_lds visible
    store_input(p_cur, true, 0, preg[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef E3DGE_MC_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long tbegin = tlast;
#endif

    f32x16 acc[NPH][NCT][NPT];
    float amax_wg = 0.0f;
    // pr_load: registers the prefetch of this step goes to; pr_store: registers holding the patch of step+1
    auto run_step = [&](int step, float (&pr_load)[NIT][8], const float (&pr_store)[NIT][8]) {
        const int cur = step & 1;
        const int k = p_cur.k, c = p_cur.c;
        const bool has1 = step + 1 < nsteps, hasp = step + kMcAhead < nsteps;
        if (has1) issue_weights(p_nx1, cur ^ 1);
        auto do_issue = [&]() {
            if (hasp) load_input(p_nxa, false, pr_load);
            MC_T(0);
        };
        auto do_compute = [&]() {
        if (c == 0) {
#pragma unroll
                for (int p = 0; p < NPH; ++p)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < NPT; ++pt) acc[p][ct][pt] = zero16();
            }
            // ---- 9 taps x (NCT x NPT) x 3 MFMAs from LDS buffer `cur` ----
            {
                const unsigned char* xb = xbuf + cur * XBUF;
                const unsigned char* wb = wbuf + cur * WBUF + (size_t)(wcg * NCT) * kMcSlabBytes + lane * 16;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    // stride-1: input offset (ky-1, kx-1); transposed: out(2i+ky) gets x[i] for ky = 0, 1 and x[i-1] for ky = 2
                    const int dy = UP ? (ky == 2 ? -1 : 0) : ky - 1, dx = UP ? (kx == 2 ? -1 : 0) : kx - 1;
                    const int ph = UP ? (ky & 1) * 2 + (kx & 1) : 0;
                    u32x4 ah[NCT], al[NCT], bh[NPT], bl[NPT];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        ah[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kMcSlabBytes + (tap * 2 + 0) * kMcFragBytes);
                        al[ct] = *reinterpret_cast<const u32x4*>(wb + ct * kMcSlabBytes + (tap * 2 + 1) * kMcFragBytes);
                    }
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        const int pix = (wrow + dy + 1) * PW + (32 * pt + col + dx + 1);
                        bh[pt] = *reinterpret_cast<const u32x4*>(xb + ((0 * 2 + half) * NPIX + pix) * 16);
                        bl[pt] = *reinterpret_cast<const u32x4*>(xb + ((1 * 2 + half) * NPIX + pix) * 16);
                    }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int pt = 0; pt < NPT; ++pt) {
                            f32x16& d = acc[ph][ct][pt];
                            d = mfma16(ah[ct], bh[pt], d);
                            d = mfma16(al[ct], bh[pt], d);
                            d = mfma16(ah[ct], bl[pt], d);
                        }
                    // the scheduler otherwise hoists the fragment reads of all nine taps (72 x 4 registers) and spills
                    if (tap % E3DGE_MC_TAPGROUP == E3DGE_MC_TAPGROUP - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            MC_T(1);
            // ---- epilogue of a finished tile ----
            if (c == a.n_chunks - 1) {
                const int b = p_cur.b, cb = p_cur.cb, ty = p_cur.ty, tx = p_cur.tx;
                const float nw = (a.noise && a.noise_w) ? a.noise_w[0] : 0.0f;
                const float* __restrict__ ec = e_lds + (k & 1) * (2 * 32 * NCTB);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int co_t = (cb * NCTB + wcg * NCT + ct) * 32;
                    const float* __restrict__ dmp = ec + (wcg * NCT + ct) * 32 + 4 * half;      // rows row_of(r, half) = 8(r>>2) + 4 half + (r&3)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        const int py = ty * TH + wrow, px = tx * TW + 32 * pt + col;     // output pixel (stride-1) / input position (UP)
#pragma unroll
                        for (int p = 0; p < NPH; ++p) {
                            const int ey = p >> 1, ex = p & 1;
                            const int oy = UP ? 2 * py + ey : py, ox = UP ? 2 * px + ex : px;
                            const bool ok = oy < a.OH && ox < a.OW;
                            const float nz = (ok && a.act && a.noise) ? nw * a.noise[(size_t)(a.noise_batch > 1 ? b : 0) * a.OH * a.OW + (size_t)oy * a.OW + ox] : 0.0f;
                            float* yp = a.y + (((size_t)b * a.Co + co_t) * a.OH + oy) * a.OW + ox;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float v = acc[p][ct][pt][r] * dmp[8 * (r >> 2) + (r & 3)];
                                if (a.act) {
                                    v = (v + nz) + dmp[32 * NCTB + 8 * (r >> 2) + (r & 3)];
                                    v = (v > 0.0f ? v : v * a.slope) * a.act_scale;
                                }
                                if (ok) {
                                    yp[(size_t)row_of(r, half) * a.OH * a.OW] = v;
                                    amax_wg = fmaxf(amax_wg, fabsf(v));
                                }
                            }
                        }
                    }
                }
            }
            MC_T(2);
        };
        auto do_convert = [&]() {
        if (has1) {
                const int c1 = p_nx1.c, b1 = p_nx1.b;
                if (b1 != b_lds) {                          // next tile belongs to another sample (rare): its style into LDS
                    __syncthreads();                        // nobody still reads the old s_lds (all readers are behind us)
                    load_style(b1);
                    b_lds = b1;
                    __syncthreads();
                }
                store_input(p_nx1, false, cur ^ 1, pr_store);
                if (c1 == 0) load_epilogue_consts(p_nx1);      // first step of the next tile follows: its epilogue constants
            }
            MC_T(3);
        };
        // Two steps ahead the staging of a step does not depend on this step's loads, so the two wave groups (one wave per
        // SIMD each) run the phases in opposite order: while waves 0-3 issue loads and convert, waves 4-7 own the matrix
        // pipe, and vice versa -- the MFMA phase of one group covers the VMEM / VALU / LDS-store phase of the other.
        if (kMcAhead == 2 && kMcSkew && wave >= 4) {
            do_compute();
            do_issue();
            do_convert();
        } else {
            do_issue();
            if (kMcAhead == 2 && kMcSkew) { do_convert(); do_compute(); }
            else { do_compute(); do_convert(); }
        }
        // weights of step s+1 (DMA, issued before this step's prefetch loads) must have landed.  Two steps ahead, the NIT*8
        // loads just issued may stay in flight (vmcnt retires in order: "at most NIT*8 outstanding" = everything older is done)
        if (kMcAhead == 2 && hasp) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIT * 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MC_T(4);
        __syncthreads();
        MC_T(5);
        p_cur = p_nx1;
        advance(p_nx1);
        if (kMcAhead == 2) advance(p_nxa); else p_nxa = p_nx1;
    };
    if (kMcAhead == 2) {
        for (int step = 0; step < nsteps; step += 2) {
            run_step(step, preg[0], preg[1]);           // loads step+2 -> preg[0]; stores step+1 from preg[1]
            if (step + 1 < nsteps) run_step(step + 1, preg[1], preg[0]);
        }
    } else {
        for (int step = 0; step < nsteps; ++step) run_step(step, preg[0], preg[0]);   // loaded at the top, stored at the end
    }
#ifdef E3DGE_MC_TIMING
    if (blockIdx.x == 0 && tid == 0 && a.out_amax) {   // profiling build: cycle sums in the unused floats of slot 0's line
        for (int i = 0; i < 6; ++i) a.out_amax[1 + i] = (float)tacc[i];
        a.out_amax[7] = (float)(__builtin_readcyclecounter() - tbegin);
        a.out_amax[8] = (float)nsteps;
    }
#endif
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax_wg = fmaxf(amax_wg, __shfl_xor(amax_wg, off, kWave));
        if (lane == 0) atomic_max_nonneg(a.out_amax + (((int)blockIdx.x * 8 + wave) & (kAmaxSlots - 1)) * kAmaxStride, amax_wg);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight image: word w of lane l of (co_tile t, chunk c, tap, hl) = f16 pair j = 2w, 2w+1 of
//     128 * scale * W[32t + (l & 31)][16c + 8(l >> 5) + j][tap]      (hi = f16(v), lo = f16(v - hi))
// plus wsq[co][ci] = sum_tap (scale W)^2 for the demodulation GEMV.  Rows / channels beyond (Co, Ci) are zero.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
modconv_pack_kernel(unsigned* __restrict__ img, float* __restrict__ wsq, const float* __restrict__ w, float scale,
                    int Co, int Ci, int co_tiles, int n_chunks) {
    const int64_t n_words = (int64_t)co_tiles * n_chunks * 9 * 2 * 64 * 4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_words; e += (int64_t)gridDim.x * 256) {
        int64_t r = e;
        const int wd = r & 3; r >>= 2;
        const int l = r & 63; r >>= 6;
        const int hl = r & 1; r >>= 1;
        const int tap = (int)(r % 9); r /= 9;
        const int c = (int)(r % n_chunks); r /= n_chunks;
        const int t = (int)r;
        const int co = 32 * t + (l & 31);
        unsigned word = 0;
        for (int e2 = 0; e2 < 2; ++e2) {
            const int ci = 16 * c + 8 * (l >> 5) + 2 * wd + e2;
            const float v = (co < Co && ci < Ci) ? kW16Scale * scale * w[((int64_t)co * Ci + ci) * 9 + tap] : 0.0f;
            const _Float16 hi = (_Float16)v;
            const _Float16 val = hl ? (_Float16)(v - (float)hi) : hi;
            word |= (unsigned)__builtin_bit_cast(unsigned short, val) << (16 * e2);
        }
        img[e] = word;
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)Co * Ci; e += (int64_t)gridDim.x * 256) {
        float s = 0.0f;
        for (int tap = 0; tap < 9; ++tap) { const float v = scale * w[e * 9 + tap]; s = fmaf(v, v, s); }
        wsq[e] = s;
    }
}

// demod[b][co] = rsqrt(sum_ci s[b,ci]^2 wsq[co][ci] + 1e-8) (one wave per output), s_amax[b] = max_ci |s[b,ci]|
__global__ void __launch_bounds__(256)
modconv_demod_kernel(float* __restrict__ demod, float* __restrict__ s_amax, const float* __restrict__ style,
                     const float* __restrict__ wsq, int Co, int Ci, int want_demod) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ s = style + (size_t)b * Ci;
    if (want_demod) {
        const int co = blockIdx.x * 4 + wave;
        if (co < Co) {
            float acc = 0.0f;
            for (int ci = lane; ci < Ci; ci += 64) { const float v = s[ci]; acc = fmaf(v * v, wsq[(size_t)co * Ci + ci], acc); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
            if (lane == 0) demod[(size_t)b * Co + co] = rsqrtf(acc + 1e-8f);
        }
    }
    if (blockIdx.x == 0 && wave == 0) {
        float m = 0.0f;
        for (int ci = lane; ci < Ci; ci += 64) m = fmaxf(m, fabsf(s[ci]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
        if (lane == 0) s_amax[b] = m;
    }
}

// max |x| of a tensor into a zero-initialised amax buffer (kAmaxSlots slots, kAmaxStride floats apart): one atomic per block
__global__ void __launch_bounds__(256) amax_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t n) {
    __shared__ float part[4];
    float m = 0.0f;
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // four independent 16-byte loads per trip (round 6: one load per trip streamed a 100-MB operand at 3.7 TB/s -- 25 us per launch, ten
    // launches per stage-2 step -- where the element-wise kernels of this library reach 5-7)
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const f32x4*>(x)[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k][0]), fabsf(v[k][1]))), fmaxf(fabsf(v[k][2]), fabsf(v[k][3])));
    }
    for (; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomic_max_nonneg(out + ((int)blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride, fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
}

// ---------------------------------------------------------------------------------------------------------------------
// All modulation vectors of a decoder in two launches (instead of one EqualLinear GEMV + two elementwise kernels + one
// demod kernel per layer): for every layer l of a device-resident table
//     s_l[b, :]   = (W_l * lin_scale) latent[b, latent_index_l, :] + bias_l * lr_mul          (EqualLinear.forward :234-244)
//     demod_l[b, co] = rsqrt(sum_ci s_l[b,ci]^2 wsq_l[co,ci] + 1e-8),   s_amax_l[b] = max_ci |s_l[b,ci]|
// One wave per output row, lanes along k.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
decoder_styles_kernel(const E3dgeModLayer* __restrict__ tab, int n_layers, int total_rows, const float* __restrict__ latent,
                      int n_latent, int style_dim, float* __restrict__ zero, int n_zero) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (the packed decoder pipeline has its amax block cleared here: this is the first launch of a forward, and a kernel of its own
    // for 4 KB of zeros cost 4.7 us of the inversion forward's timeline)
    if (zero && b == 0)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n_zero; i += gridDim.x * 256) zero[i] = 0.0f;
    const int row = blockIdx.x * 4 + wave;
    if (row >= total_rows) return;
    int l = 0;
    while (l + 1 < n_layers && row >= tab[l + 1].row_start) ++l;
    const E3dgeModLayer L = tab[l];
    const int r = row - L.row_start;
    const float* __restrict__ w = L.mod_weight + (size_t)r * style_dim;
    const float* __restrict__ x = latent + ((size_t)b * n_latent + L.latent_index) * style_dim;
    float acc = 0.0f;
    for (int k = lane; k < style_dim; k += 64) acc = fmaf(__fmul_rn(w[k], L.lin_scale), x[k], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if (lane == 0) L.style_out[(size_t)b * L.ci + r] = acc + __fmul_rn(L.mod_bias[r], L.lr_mul);
}

__global__ void __launch_bounds__(256)
decoder_demod_kernel(const E3dgeModLayer* __restrict__ tab, int n_layers, int total_co) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = blockIdx.x * 4 + wave;                  // rows [0, total_co): demod outputs; then one amax row per layer
    if (o < total_co) {
        int l = 0;
        while (l + 1 < n_layers && o >= tab[l + 1].co_start) ++l;
        const E3dgeModLayer L = tab[l];
        if (!L.demod_out) return;
        const int co = o - L.co_start;
        const float* __restrict__ s = L.style_out + (size_t)b * L.ci;
        float acc = 0.0f;
        for (int ci = lane; ci < L.ci; ci += 64) { const float v = s[ci]; acc = fmaf(v * v, L.wsq[(size_t)co * L.ci + ci], acc); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
        if (lane == 0) L.demod_out[(size_t)b * L.co + co] = rsqrtf(acc + 1e-8f);
    } else if (o - total_co < n_layers) {
        const E3dgeModLayer L = tab[o - total_co];
        if (!L.s_amax_out) return;
        const float* __restrict__ s = L.style_out + (size_t)b * L.ci;
        float m = 0.0f;
        for (int ci = lane; ci < L.ci; ci += 64) m = fmaxf(m, fabsf(s[ci]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
        if (lane == 0) L.s_amax_out[b] = m;
    }
}

template <bool UP, int TH, int NPT, int CG, int NCT, int AH>
static int launch_modconv(ModconvK k, hipStream_t st, const char* what) {
    constexpr int TW = 32 * NPT;
    constexpr int PH = UP ? TH + 1 : TH + 2, PW = UP ? TW + 1 : TW + 2;
    constexpr int lds = 2 * (4 * PH * PW * 16) + 2 * (CG * NCT * kMcSlabBytes) + kMcMaxCi * 4 + 2 * 2 * 32 * CG * NCT * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int ext_y = UP ? k.H + 1 : k.H, ext_x = UP ? k.W + 1 : k.W;       // tiled extent (input positions for UP)
    k.tiles_y = (ext_y + TH - 1) / TH;
    k.tiles_x = (ext_x + TW - 1) / TW;
    E3DGE_REQUIRE(k.Co % (32 * CG * NCT) == 0, "%s: Co=%d not a multiple of the %d-channel block", what, k.Co, 32 * CG * NCT);
    k.co_blocks = k.Co / (32 * CG * NCT);
    const int64_t n_tiles = (int64_t)k.B * k.co_blocks * k.tiles_y * k.tiles_x;
    E3DGE_REQUIRE(n_tiles < ((int64_t)1 << 30), "%s: too many tiles", what);
    k.n_tiles = (int)n_tiles;
    auto fn = &modconv_kernel<UP, TH, NPT, CG, NCT, AH>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    const int wgs_per_cu = (160 * 1024) / lds >= 2 ? 2 : 1;
    int grid = 256 * wgs_per_cu;
    if (grid > k.n_tiles) grid = k.n_tiles;
    fn<<<dim3((unsigned)grid), dim3(kMcThreads), lds, st>>>(k);
    return check_launch(what);
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int64_t e3dge_modconv_packed_words(int co, int ci) {
    if (co <= 0 || ci <= 0) return 0;
    return (int64_t)((co + 31) / 32) * ((ci + 15) / 16) * 9 * 2 * 64 * 4;
}

extern "C" int e3dge_modconv_pack_weights(uint32_t* image, float* wsq, const float* weight, float scale, int co, int ci,
                                          e3dge_stream_t stream) {
    E3DGE_REQUIRE(image && wsq && weight, "modconv_pack_weights: null pointer");
    E3DGE_REQUIRE(co > 0 && ci > 0, "modconv_pack_weights: bad sizes");
    E3DGE_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, "modconv_pack_weights: image must be 16-B aligned");
    modconv_pack_kernel<<<dim3(512), dim3(256), 0, as_stream(stream)>>>(image, wsq, weight, scale, co, ci, (co + 31) / 32, (ci + 15) / 16);
    return check_launch("modconv_pack_weights");
}

extern "C" int e3dge_modconv_demod(float* demod, float* s_amax, const float* style, const float* wsq, int batch, int co,
                                   int ci, int demodulate, e3dge_stream_t stream) {
    E3DGE_REQUIRE(s_amax && style && wsq && (demod || !demodulate), "modconv_demod: null pointer");
    E3DGE_REQUIRE(batch >= 0 && co > 0 && ci > 0, "modconv_demod: bad sizes");
    if (batch == 0) return E3DGE_OK;
    modconv_demod_kernel<<<dim3((unsigned)((co + 3) / 4), (unsigned)batch), dim3(256), 0, as_stream(stream)>>>(demod, s_amax, style, wsq, co, ci, demodulate);
    return check_launch("modconv_demod");
}

namespace e3dge {
int decoder_styles_launch(const E3dgeModLayer* table, int n_layers, int total_rows, int total_co, const float* latent, int n_latent,
                          int style_dim, int batch, float* zero, int n_zero, hipStream_t st) {
    E3DGE_REQUIRE(table && latent, "decoder_styles: null pointer");
    E3DGE_REQUIRE(n_layers >= 1 && n_layers <= 64 && total_rows >= 1 && total_co >= 0 && n_latent >= 1 && style_dim >= 1 && batch >= 0,
                  "decoder_styles: bad sizes");
    if (batch == 0) return E3DGE_OK;
    decoder_styles_kernel<<<dim3((unsigned)((total_rows + 3) / 4), (unsigned)batch), dim3(256), 0, st>>>(table, n_layers, total_rows, latent, n_latent, style_dim,
                                                                                                         zero, n_zero);
    int rc = check_launch("decoder_styles");
    if (rc) return rc;
    decoder_demod_kernel<<<dim3((unsigned)((total_co + n_layers + 3) / 4), (unsigned)batch), dim3(256), 0, st>>>(table, n_layers, total_co);
    return check_launch("decoder_styles(demod)");
}
}  // namespace e3dge

extern "C" int e3dge_decoder_styles(const E3dgeModLayer* table, int n_layers, int total_rows, int total_co, const float* latent,
                                    int n_latent, int style_dim, int batch, e3dge_stream_t stream) {
    return decoder_styles_launch(table, n_layers, total_rows, total_co, latent, n_latent, style_dim, batch, nullptr, 0, as_stream(stream));
}

// the same over the first `width` columns of rows of pitch `ld` (a column block of a wider row tensor: the gradient of a cat's first piece)
__global__ void __launch_bounds__(256) amax_rows_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t n_rows, int width, int64_t ld) {
    __shared__ float part[4];
    float m = 0.0f;
    const int64_t n = n_rows * width;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {               // eight independent loads per trip
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t e = i + k * stride, r = e / width;
            v[k] = x[r * ld + (e - r * width)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v[k]));
    }
    for (; i < n; i += stride) {
        const int64_t r = i / width;
        m = fmaxf(m, fabsf(x[r * ld + (i - r * width)]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomic_max_nonneg(out + ((int)blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride, fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
}

extern "C" int e3dge_amax_rows(float* out, const float* x, int64_t n_rows, int width, int64_t ld, e3dge_stream_t stream) {
    E3DGE_REQUIRE(out && n_rows >= 0 && width >= 0 && ld >= width && (x || n_rows * width == 0), "amax_rows: bad arguments");
    if (n_rows * width == 0) return E3DGE_OK;
    int64_t blocks = (n_rows * width + 2047) / 2048;
    if (blocks > 4096) blocks = 4096;
    amax_rows_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(out, x, n_rows, width, ld);
    return check_launch("amax_rows");
}

extern "C" int e3dge_amax(float* out, const float* x, int64_t n, e3dge_stream_t stream) {
    E3DGE_REQUIRE(out && (x || n == 0) && n >= 0, "amax: bad arguments");
    if (n == 0) return E3DGE_OK;
    E3DGE_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "amax: x must be 16-B aligned");
    int64_t blocks = (n / 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    amax_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(out, x, n);
    return check_launch("amax");
}

extern "C" int e3dge_modconv3x3(const E3dgeModconvArgs* r, e3dge_stream_t stream) {
    E3DGE_REQUIRE(r != nullptr, "modconv3x3: null args");
    E3DGE_REQUIRE(r->batch >= 0 && r->ci > 0 && r->co > 0 && r->height > 0 && r->width > 0, "modconv3x3: bad sizes");
    if (r->batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(r->x && r->wimg && r->style && r->in_amax && r->s_amax && r->y, "modconv3x3: null pointer");
    E3DGE_REQUIRE(r->ci % 16 == 0 && r->co % 32 == 0 && r->ci <= kMcMaxCi, "modconv3x3: needs Ci %% 16 == 0, Ci <= %d and Co %% 32 == 0 (got %d, %d)", kMcMaxCi, r->ci, r->co);
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(r->wimg) | reinterpret_cast<uintptr_t>(r->style)) & 15) == 0, "modconv3x3: wimg/style must be 16-B aligned");
    E3DGE_REQUIRE(!(r->upsample && r->act), "modconv3x3: the up-sampling variant has no fused activation (the FIR blur comes first)");
    E3DGE_REQUIRE(r->noise == nullptr || r->noise_w != nullptr, "modconv3x3: noise without noise_w");
    E3DGE_REQUIRE((int64_t)r->batch * r->ci * r->height * r->width < ((int64_t)1 << 31) &&
                  (int64_t)r->batch * r->co * (2 * (int64_t)r->height + 1) * (2 * (int64_t)r->width + 1) < ((int64_t)1 << 40), "modconv3x3: tensor too large");
    ModconvK k{};
    k.x = r->x; k.wimg = r->wimg; k.style = r->style; k.demod = r->demod; k.in_amax = r->in_amax; k.s_amax = r->s_amax;
    k.noise = r->noise; k.noise_w = r->noise_w; k.bias = r->bias; k.y = r->y; k.out_amax = r->out_amax;
    k.slope = r->negative_slope; k.act_scale = r->act_scale; k.act = r->act;
    k.B = r->batch; k.Ci = r->ci; k.Co = r->co; k.H = r->height; k.W = r->width;
    k.OH = r->upsample ? 2 * r->height + 1 : r->height; k.OW = r->upsample ? 2 * r->width + 1 : r->width;
    k.n_chunks = r->ci / 16; k.noise_batch = r->noise_batch;
    hipStream_t st = as_stream(stream);
    const int64_t px = (int64_t)r->height * r->width;
    // tile shapes by layer size, so that the small early layers still spread over the 256 CUs
    if (!r->upsample) {
        if (r->co % 64 != 0) return launch_modconv<false, 8, 2, 1, 1, E3DGE_MC_AH_D>(k, st, "modconv3x3<8x64,32co>");
        if (px <= 64 * 64) return launch_modconv<false, 4, 1, 2, 1, E3DGE_MC_AH_A>(k, st, "modconv3x3<4x32,64co>");
        // (above 128^2 an 8 x 64 tile with two co-tiles per wave measured a few % faster in round 2, but its instantiation spills 60 B;
        // this planar path is the fallback of the packed pipeline now, so the 4 x 64 shape serves every larger size: no scratch anywhere)
        return launch_modconv<false, 4, 2, 2, 1, E3DGE_MC_AH_B>(k, st, "modconv3x3<4x64,64co>");
    }
    if (r->co % 64 == 0) return launch_modconv<true, 4, 1, 2, 1, E3DGE_MC_AH_TA>(k, st, "modconv3x3T<4x32,64co>");
    return launch_modconv<true, 8, 1, 1, 1, E3DGE_MC_AH_TD>(k, st, "modconv3x3T<8x32,32co>");
}
