// Fused FiLM-SIREN volume renderer for gfx950 (MI355X): ray generation -> sample placement -> 9 FiLM
// sine layers -> sdf / rgb heads -> SDF->alpha -> transmittance scan -> composites, in ONE kernel.
//
// Reference path being replaced (all in project/utils/volume_renderer.py):
//   get_rays :769-794, render :1666-1701, render_rays :1183-1287, run_network :1052-1128,
//   FiLMSiren.forward :116-132, SirenGenerator.forward :168-264, volume_integration :809-943.
//
// Design (see DESIGN.md for the numbers)
//   * The MLP is a dense fp32 contraction (526,848 MAC per point) -> it runs on v_mfma_f32_32x32x2_f32.
//   * One workgroup = 4 waves = one wave per SIMD (the kernel wants ~400 of the 512 unified registers).
//     A wave owns 32 points and keeps their 256-wide activation ENTIRELY IN REGISTERS across all layers:
//     the MFMA C/D fragment of layer L (rows = features, cols = points) has exactly the lane structure of
//     a B operand (k on lane>>5, column on lane&31), so layer L's accumulators feed layer L+1's MFMAs with
//     no shuffle, no LDS round trip and no HBM traffic.  K is consumed in the fragment's own k order.
//   * Weights are re-laid once (e3dge_siren_pack_weights) into the lane-linear A-fragment image and are
//     streamed L2 -> LDS with global_load_lds (16 B/lane) in 32-KiB chunks (one 32-feature output tile x
//     K=256), triple buffered; the 4 waves of a workgroup share each chunk (ds_read_b128, conflict-free
//     by construction because the image is lane-linear).
//   * The last (view) layer is evaluated TRANSPOSED (activations as the A operand, weights as B) so its
//     output has features on lanes and points on registers: the feature composite sum_s w_s * f_s becomes
//     16 FMAs per lane instead of a cross-lane reduction.
//   * Transmittance is a sequential front-to-back scan per ray (same order as torch.cumprod), carried in
//     LDS across the 128-point sub-tiles of a workgroup's ray block, so rays x samples can be tiled
//     without aligning rays to tiles.
#include "siren_common.h"
#include "siren16.h"

namespace e3dge {

// ---------------------------------------------------------------------------------------------
// the kernel.  MODE 0 = render (rays x samples + compositing), MODE 1 = arbitrary point set (raw outputs)
// ---------------------------------------------------------------------------------------------
// Phase timing (variant builds only, -DE3DGE_PHASE_TIMING): wave 0 of workgroup 0 records s_memtime at the phase
// boundaries of each sub-tile and, at the very end, overwrites the first floats of the `dists` output with the
// per-phase cycle counts (tools/phase_timing.py reads them back).
#ifdef E3DGE_PHASE_TIMING
#define PHASE_MARK(i) do { if (MODE == 0 && blockIdx.x == 0 && tid == 0 && sub < 4) tstamp[sub * 6 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PHASE_MARK(i) do { } while (0)
#endif

// PREC 0: fp32 MFMA (v_mfma_f32_32x32x2_f32).  PREC 1: "f16x3" -- split-f16 contraction on v_mfma_f32_32x32x16_f16.
// SAVE: additionally store the pre-sine argument of every FiLM layer (what the backward kernels consume).
template <int MODE, int PREC, bool SAVE>
__global__ void __launch_bounds__(kThreads) siren_kernel(const SirenK a) {
    constexpr bool F16 = PREC == 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wbuf = smem + kLdsW;
    float* const film_s = smem + kLdsFilm;
    float* const head_s = smem + kLdsHead;
    float* const vtail_s = smem + kLdsVTail;
    float* const feat_acc = smem + kLdsFeat;
    float* const part = smem + kLdsPart;
    float* const alpha_s = smem + kLdsAlpha;
    float* const wgt_s = smem + kLdsWgt;
    float* const z_s = smem + kLdsZ;
    float* const pts_s = smem + kLdsPts;
    float* const rgb_s = smem + kLdsRgb;
    float* const state = smem + kLdsState;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;

    // ---- work assignment ----
    int b, npts, n_sub;
    int pix0 = 0, nrays = 0;            // render
    long long pt0 = 0;                  // points mode: first point of this workgroup inside image b
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
    // Per-image FiLM block in LDS with the layer bias folded into the offset:
    //   sin(gamma * (W h + b) + beta) = sin(gamma * (W h) + beta'),  beta' = gamma * b + beta
    // so the accumulators start at literal zero and the pipelined tiles issue no VMEM besides the weight DMA.
    // (Differs from the reference's rounding order by ~1 ulp of the sine argument, the same size as the
    // difference between the MFMA's and MKL's summation orders.)  Published by the first chunk barrier.
    for (int i = tid; i < 9 * kWidth; i += kThreads) {
        const int l = i >> 8, n = i & 255;
        const float gm = film_g[(l * 2 + 0) * kWidth + n], bt = film_g[(l * 2 + 1) * kWidth + n];
        // f16x3: the streamed weights carry a factor 128 (kW16Scale); gamma / 128 is exact and undoes it
        film_s[(l * 2 + 0) * kWidth + n] = (F16 && l >= 1) ? gm * (1.0f / kW16Scale) : gm;
        film_s[(l * 2 + 1) * kWidth + n] = __fadd_rn(__fmul_rn(gm, packed[kOffBias + l * kWidth + n]), bt);
    }
    const float* __restrict__ film = film_s;
    for (int i = tid; i < kHeadFloats; i += kThreads) head_s[i] = packed[kOffWSigma + i];
    for (int i = tid; i < kNT * 2 * 64; i += kThreads) vtail_s[i] = packed[kOffVTail + i];
    const float* __restrict__ bias_all = packed + kOffBias;

    // per-image camera (render mode)
    float cw[12] = {0}, focal = 1.f, nearv = 0.f, farv = 0.f;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) cw[i] = a.c2w[b * 12 + i];
        focal = a.focal[b]; nearv = a.near[b]; farv = a.far[b];
        // zero the per-ray state and the feature accumulators
        for (int i = tid; i < kRMax * kStateStride; i += kThreads) state[i] = 0.0f;
        for (int i = tid; i < kRMax * kFPitch; i += kThreads) feat_acc[i] = 0.0f;
    }

    // ---- weight chunk pipeline (ChunkPipe, siren_common.h) ----
    ChunkPipe pipe;
    pipe.init(wbuf, packed + (F16 ? kOffBig16 : kOffBig), wave, lane, 0, kChunksPerPass);
    pipe.prime();
    auto issue_piece = [&](int i) { pipe.issue_piece(i); };
    auto chunk_sync = [&]() { pipe.sync(); };
    auto advance_chunk = [&]() { pipe.advance(); };
    // first chunk (and the LDS parameter blocks staged above) visible to everyone; prime the fragment ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 ring[kRing];
    u32x4 ringH[kRing16], ringL[kRing16];
    if (!F16) {
        ring[0] = reinterpret_cast<const f32x4*>(pipe.wcur)[lane];
        ring[1] = reinterpret_cast<const f32x4*>(pipe.wcur)[64 + lane];
    } else {
#pragma unroll
        for (int g = 0; g < kRing16 - 1; ++g) {
            ringH[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 0) * 64 + lane];
            ringL[g] = reinterpret_cast<const u32x4*>(pipe.wcur)[(g * 2 + 1) * 64 + lane];
        }
    }

    // Register-resident activations of this wave's 32 points.
    //   fp32 : in[t][r]                      = feature 32t + row_of(r, half)
    //   f16x3: inH/inL[2t + (r>>3)] word (r&7)>>1, half-word r&1  (packed f16 hi / lo of the same value)
    f32x16 in[kNT], out[kNT];
    u32x4 inH[2 * kNT], inL[2 * kNT], outH[2 * kNT], outL[2 * kNT];
#ifdef E3DGE_PHASE_TIMING
    unsigned long long tstamp[24];
    for (int i = 0; i < 24; ++i) tstamp[i] = 0;
#endif

    const int tid_k = tid;
    for (int sub = 0; sub < n_sub; ++sub) {
        // Opaque per-iteration copies of the lane indices: every address derived from them is then recomputed inside
        // the sub-tile instead of being hoisted out of this loop, held across the MFMA chain (where no register is free)
        // and spilled -- the reloads are vmcnt-ordered behind the LDS-DMA pieces and stall on them.
        int tid_o = tid_k;
        asm volatile("" : "+v"(tid_o));
        const int tid = tid_o, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
        PHASE_MARK(0);
        // =====================================================================================
        // 1. this lane's point
        // =====================================================================================
        const int p_sub = 32 * wave + col;                 // index inside the sub-tile
        const int p = sub * kTilePts + p_sub;               // index inside the workgroup's block
        const bool valid = p < npts;
        const int pc = valid ? p : (npts - 1);
        float px = 0.f, py = 0.f, pz = 0.f;                 // world-space point
        float vx = 0.f, vy = 0.f, vz = 0.f;                 // view direction fed to the MLP
        float zval = 0.f, dist = 0.f;
        int ray_l = 0, s_idx = 0;
        int64_t gpt;                                        // global point index (b, ray, s) / (b, n)
        if (MODE == 0) {
            ray_l = pc / S;
            s_idx = pc - ray_l * S;
            const int pix = pix0 + ray_l;
            const int iy = pix / a.Wd, ix = pix - iy * a.Wd;
            gpt = ((int64_t)b * a.H * a.Wd + pix) * S + s_idx;
            // get_rays (:771-788): pixel centres at +0.5, camera looks down -z
            const float hres = (float)a.res * 0.5f;
            const float d0 = __fdiv_rn(((float)ix + 0.5f) - hres, focal);
            const float d1 = -__fdiv_rn(((float)iy + 0.5f) - hres, focal);
            const float d2 = -1.0f;
            float rd[3];
#pragma unroll
            for (int m = 0; m < 3; ++m)
                rd[m] = __fadd_rn(__fadd_rn(__fmul_rn(d0, cw[4 * m + 0]), __fmul_rn(d1, cw[4 * m + 1])),
                                  __fmul_rn(d2, cw[4 * m + 2]));
            // static_viewdirs: camera-space dirs, normalised in render (:1679)
            const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)));
            vx = __fdiv_rn(d0, dn); vy = __fdiv_rn(d1, dn); vz = __fdiv_rn(d2, dn);
            // z_vals (:1211) and pts (:1231)
            const float tv = a.t_vals[s_idx];
            zval = __fadd_rn(__fmul_rn(nearv, __fsub_rn(1.0f, tv)), __fmul_rn(farv, tv));
            px = __fadd_rn(cw[3], __fmul_rn(rd[0], zval));
            py = __fadd_rn(cw[7], __fmul_rn(rd[1], zval));
            pz = __fadd_rn(cw[11], __fmul_rn(rd[2], zval));
            // dists (:826-837)
            const float rdn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rd[0], rd[0]), __fmul_rn(rd[1], rd[1])), __fmul_rn(rd[2], rd[2])));
            if (s_idx + 1 < S) {
                const float tn = a.t_vals[s_idx + 1];
                const float zn = __fadd_rn(__fmul_rn(nearv, __fsub_rn(1.0f, tn)), __fmul_rn(farv, tn));
                dist = __fmul_rn(__fsub_rn(zn, zval), rdn);
            } else {
                dist = __fmul_rn(1e10f, rdn);
            }
            if (valid && half == 0) {
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

        // training: where this lane's point keeps its [9][256] pre-sine arguments (null: nothing is saved)
        float* const sv = (SAVE && valid && half >= 0) ? a.save_args + gpt * (9 * kWidth) : nullptr;
        // first point of this workgroup's block in the same numbering (block points are contiguous)
        const int64_t gpt_block = (MODE == 0) ? ((int64_t)b * a.H * a.Wd + pix0) * S : (int64_t)b * a.n_pts + pt0;

        // =====================================================================================
        // 2. layer 0 (3 -> 256): two K=2 MFMAs per output tile, operands straight from L2
        // =====================================================================================
        {
            const float xs = __fmul_rn(px, a.box_scale), ys = __fmul_rn(py, a.box_scale), zs = __fmul_rn(pz, a.box_scale);
            const float b0 = half ? ys : xs;
            const float b1 = half ? 0.0f : zs;
            const float* __restrict__ wf = packed + kOffFirst;
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                f32x16 acc = zero16();
                acc = mfma32(wf[(t * 2 + 0) * 64 + lane], b0, acc);
                acc = mfma32(wf[(t * 2 + 1) * 64 + lane], b1, acc);
                const f32x16 v0 = film_sin_std(acc, film, t, half, SAVE ? sv : nullptr);
                if (!F16) {
                    in[t] = v0;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2)
                        SPLIT2_TO(v0[r], v0[r + 1], inH[2 * t + (r >> 3)][(r & 7) >> 1], inL[2 * t + (r >> 3)][(r & 7) >> 1]);
                }
            }
        }

        PHASE_MARK(1);
        // =====================================================================================
        // 3. layers 1..7 (256 -> 256): 56 output tiles, weights streamed through LDS.  Software pipeline: the
        //    FiLM + sine epilogue of tile i-1 issues inside the MFMA stream of tile i (big_tile's epi hook), the
        //    bias of tile i+1 is fetched while tile i computes.  Only the last tile of a layer (whose result the
        //    next layer's first MFMAs need) runs its epilogue on its own.
        // =====================================================================================
        if (!F16) {
#pragma unroll 1
            for (int L = 1; L < E3DGE_SIREN_DEPTH; ++L) {
                const float* __restrict__ film_l = film + L * 2 * kWidth;
                f32x16 prev;
#pragma unroll
                for (int t = 0; t < kNT; ++t) {
                    f32x16 acc = zero16();
                    if (t == 0) {
                        acc = big_tile<false, 0>(pipe.wcur, pipe.wnxt, lane, in, acc, ring, NoEpilogue(), chunk_sync, issue_piece);
                    } else {
                        // FiLM (gamma, beta') of the 4 registers being processed, and of the NEXT 4 (fetched from LDS one quad
                        // ahead: a read issued right before its use exposes the LDS latency four times per tile)
                        const float* __restrict__ fl = film_l + 32 * (t - 1) + 4 * half;
                        f32x4 g4 = *reinterpret_cast<const f32x4*>(fl), b4 = *reinterpret_cast<const f32x4*>(fl + kWidth);
                        f32x4 g4n = g4, b4n = b4, sarg;
                        acc = big_tile<false, E3DGE_SPREAD_STD>(pipe.wcur, pipe.wnxt, lane, in, acc, ring, [&](int r) {
                            if ((r & 3) == 0 && r < 12) {
                                g4n = *reinterpret_cast<const f32x4*>(fl + 8 * ((r >> 2) + 1));
                                b4n = *reinterpret_cast<const f32x4*>(fl + kWidth + 8 * ((r >> 2) + 1));
                            }
                            const float arg = fmaf(g4[r & 3], prev[r], b4[r & 3]);
                            out[t - 1][r] = sin_f32(arg);
                            if (SAVE) {
                                sarg[r & 3] = arg;
                                if ((r & 3) == 3 && sv) *reinterpret_cast<f32x4*>(sv + L * kWidth + 32 * (t - 1) + 8 * (r >> 2) + 4 * half) = sarg;
                            }
                            if ((r & 3) == 3) { g4 = g4n; b4 = b4n; }
                        }, chunk_sync, issue_piece);
                        asm volatile("" : "+a"(out[t - 1]));   // park finished activations in the accumulator half
                    }
                    advance_chunk();
                    prev = acc;
                    asm volatile("" : "+v"(prev));           // the epilogue's VALU reads it: keep it out of the AGPRs
                }
                out[kNT - 1] = film_sin_std(prev, film_l, kNT - 1, half, (SAVE && sv) ? sv + L * kWidth : nullptr);
#pragma unroll
                for (int tt = 0; tt < kNT; ++tt) {
                    in[tt] = out[tt];
                    asm volatile("" : "+a"(in[tt]));         // MFMA B/A operands are read straight from AGPRs
                }
            }
        } else {
#pragma unroll 1
            for (int L = 1; L < E3DGE_SIREN_DEPTH; ++L) {
                const float* __restrict__ film_l = film + L * 2 * kWidth;
                f32x16 prev;
#pragma unroll
                for (int t = 0; t < kNT; ++t) {
                    f32x16 acc = zero16(), accb = zero16();
                    if (t == 0) {
                        big_tile_f16<false>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL, NoEpilogue(), chunk_sync, issue_piece);
                    } else {
                        float xe = 0.f;
                        const float* __restrict__ fl = film_l + 32 * (t - 1) + 4 * half;
                        f32x4 g4 = *reinterpret_cast<const f32x4*>(fl), b4 = *reinterpret_cast<const f32x4*>(fl + kWidth);
                        f32x4 g4n = g4, b4n = b4, sarg;
                        big_tile_f16<false>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL, [&](int r) {
                            if ((r & 3) == 0 && r < 12) {       // FiLM of the next quad, one quad ahead of its use
                                g4n = *reinterpret_cast<const f32x4*>(fl + 8 * ((r >> 2) + 1));
                                b4n = *reinterpret_cast<const f32x4*>(fl + kWidth + 8 * ((r >> 2) + 1));
                            }
                            const float arg = fmaf(g4[r & 3], prev[r], b4[r & 3]);
                            const float x = sin_f32(arg);
                            if (SAVE) {
                                sarg[r & 3] = arg;
                                if ((r & 3) == 3 && sv) *reinterpret_cast<f32x4*>(sv + L * kWidth + 32 * (t - 1) + 8 * (r >> 2) + 4 * half) = sarg;
                            }
                            if (r & 1) SPLIT2_TO(xe, x, outH[2 * (t - 1) + (r >> 3)][(r & 7) >> 1], outL[2 * (t - 1) + (r >> 3)][(r & 7) >> 1]);
                            else xe = x;
                            if ((r & 3) == 3) { g4 = g4n; b4 = b4n; }
                        }, chunk_sync, issue_piece);
                        asm volatile("" : "+a"(outH[2 * (t - 1)]), "+a"(outH[2 * (t - 1) + 1]), "+a"(outL[2 * (t - 1)]), "+a"(outL[2 * (t - 1) + 1]));
                    }
                    advance_chunk();
                    prev = acc + accb;
                    asm volatile("" : "+v"(prev));
                }
                {
                    const f32x16 v7 = film_sin_std(prev, film_l, kNT - 1, half, (SAVE && sv) ? sv + L * kWidth : nullptr);
#pragma unroll
                    for (int r = 0; r < 16; r += 2)
                        SPLIT2_TO(v7[r], v7[r + 1], outH[2 * (kNT - 1) + (r >> 3)][(r & 7) >> 1], outL[2 * (kNT - 1) + (r >> 3)][(r & 7) >> 1]);
                }
#pragma unroll
                for (int g = 0; g < 2 * kNT; ++g) {
                    inH[g] = outH[g]; inL[g] = outL[g];
                    asm volatile("" : "+a"(inH[g]), "+a"(inL[g]));
                }
            }
        }

        PHASE_MARK(2);
        // =====================================================================================
        // 4. sdf head (:206-208) on the backbone output, standard layout: features live on registers
        // =====================================================================================
        float sdf;
        {
            const float* __restrict__ ws = head_s;
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < kNT; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + 32 * c + 8 * q + 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc = fmaf(w4[j], F16 ? acts16_get(inH, inL, c, 4 * q + j) : in[c][4 * q + j], acc);
                }
            sdf = acc + xhalf(acc) + head_s[4 * kWidth];
        }

        if (MODE == 0) {
            // SDF -> density -> alpha (:804-807, :852-861)
            const float sg = __fdiv_rn(sigmoid_f32(__fdiv_rn(-sdf, a.sigmoid_beta)), a.sigmoid_beta);
            const float alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sg, dist)));
            if (half == 0) {
                alpha_s[p_sub] = valid ? alpha : 0.0f;
                z_s[p_sub] = zval;
                pts_s[p_sub * 3 + 0] = px; pts_s[p_sub * 3 + 1] = py; pts_s[p_sub * 3 + 2] = pz;
                if (valid && a.sdf) a.sdf[gpt] = sdf;
            }
            __syncthreads();
            // transmittance scan (:869-886): one thread per ray touching this sub-tile, front to back
            const int sub_lo = sub * kTilePts;
            const int sub_hi = min(sub_lo + kTilePts, npts);
            const int r_first = sub_lo / S, r_last = (sub_hi - 1) / S;
            for (int i = tid; i <= r_last - r_first; i += kThreads) {
                const int rl = r_first + i;
                const int s_lo = max(0, sub_lo - rl * S), s_hi = min(S, sub_hi - rl * S);
                float* st = state + rl * kStateStride;
                float T = (s_lo == 0) ? 1.0f : st[0];
                float wsum = (s_lo == 0) ? 0.0f : st[1];
                float dep = st[2], x0 = st[3], x1 = st[4], x2 = st[5];
                // 4 samples per trip: the 20 LDS reads of a trip are independent of the carried (T, wsum) chain and
                // issue together; only the two multiplies per sample are serial.
                for (int s0 = s_lo; s0 < s_hi; s0 += 4) {
                    float al[4], zz[4], p0[4], p1[4], p2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ps = min(rl * S + s0 + u, sub_hi - 1) - sub_lo;
                        al[u] = alpha_s[ps]; zz[u] = z_s[ps];
                        p0[u] = pts_s[ps * 3 + 0]; p1[u] = pts_s[ps * 3 + 1]; p2[u] = pts_s[ps * 3 + 2];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = s0 + u;
                        if (s < s_hi) {
                            float w = __fmul_rn(al[u], T);
                            if (a.force_bg && s == S - 1) w = __fsub_rn(1.0f, wsum);
                            else wsum = __fadd_rn(wsum, w);
                            T = __fmul_rn(T, __fadd_rn(__fsub_rn(1.0f, al[u]), 1e-10f));
                            wgt_s[rl * S + s - sub_lo] = w;
                            dep = __fadd_rn(dep, __fmul_rn(w, zz[u]));
                            x0 = __fadd_rn(x0, __fmul_rn(w, p0[u]));
                            x1 = __fadd_rn(x1, __fmul_rn(w, p1[u]));
                            x2 = __fadd_rn(x2, __fmul_rn(w, p2[u]));
                        }
                    }
                }
                st[0] = T; st[1] = wsum; st[2] = dep; st[3] = x0; st[4] = x1; st[5] = x2;
            }
            __syncthreads();
            if (valid && half == 0 && a.weights) a.weights[gpt] = wgt_s[p_sub];
        }

        // optional per-point texture FiLM (SirenGenerator.forward_tex :217-220), after the sdf head read h
        if (MODE == 0 && a.tex_alpha) {
            const float* __restrict__ ta = a.tex_alpha + gpt * kWidth;
            const float* __restrict__ tb = a.tex_beta + gpt * kWidth;
#pragma unroll
            for (int c = 0; c < kNT; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ta + 32 * c + 8 * q + 4 * half);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(tb + 32 * c + 8 * q + 4 * half);
                    if (!F16) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            in[c][4 * q + j] = __fadd_rn(__fmul_rn(__fadd_rn(a4[j], 1.0f), in[c][4 * q + j]), b4[j]);
                    } else {
                        float y[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            y[j] = __fadd_rn(__fmul_rn(__fadd_rn(a4[j], 1.0f), acts16_get(inH, inL, c, 4 * q + j)), b4[j]);
                        const int g = 2 * c + (q >> 1), k0 = 2 * (q & 1);
                        SPLIT2_TO(y[0], y[1], inH[g][k0], inL[g][k0]);
                        SPLIT2_TO(y[2], y[3], inH[g][k0 + 1], inL[g][k0 + 1]);
                    }
                }
        }

        PHASE_MARK(3);
        // =====================================================================================
        // 5. view layer (259 -> 256), TRANSPOSED: rows (registers) = points, cols (lanes) = features
        // =====================================================================================
        // per-register (= per point row) compositing weight and ray slot of this wave's 32-point slab
        // Composite weights of this wave's 32-point slab, split by the ray they belong to.  With S >= 16 a slab
        // touches at most 3 rays; their boundaries inside the slab are at rows b1 and b1 + S.
        // They are the same for all 32 lanes of a half, and 48 registers of them next to the 48 rgb partials and the
        // 128 activation registers made the view layer spill -- with every reload ordered behind the LDS-DMA pieces
        // in vmcnt.  They live in LDS instead ([slot][half][16 rows] per wave) and are read back a quad at a time.
        float* const wq_s = smem + kLdsWq + wave * (kMaxSlots * 2 * 16);
        int slab_first_ray = 0, slab_nslots = 0;
        if (MODE == 0) {
            const int slab_lo = sub * kTilePts + 32 * wave;
            const int slab_hi = min(slab_lo + 32, npts);           // exclusive; may be <= slab_lo
            if (slab_hi > slab_lo) {
                slab_first_ray = slab_lo / S;
                slab_nslots = (slab_hi - 1) / S - slab_first_ray + 1;
            }
            const int b1 = (slab_first_ray + 1) * S - slab_lo, b2 = b1 + S;
            if (col < 16) {                                       // lane (col = r, half) owns row row_of(r, half)
                const int row = row_of(col, half);
                const float w = (slab_lo + row < npts) ? wgt_s[32 * wave + row] : 0.0f;
                wq_s[(0 * 2 + half) * 16 + col] = (row < b1) ? w : 0.0f;
                wq_s[(1 * 2 + half) * 16 + col] = (row >= b1 && row < b2) ? w : 0.0f;
                wq_s[(2 * 2 + half) * 16 + col] = (row >= b2) ? w : 0.0f;
            }
        }
        float prgb[3][16];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) prgb[c][r] = 0.0f;
        {
            const float* __restrict__ film_v = film + 8 * 2 * kWidth;
            const float* __restrict__ wvt = vtail_s;
            const float* __restrict__ wrgb = head_s + kWidth;
            const float a0 = half ? vy : vx;
            const float a1 = half ? 0.0f : vz;
            const int slab_p0 = sub * kTilePts + 32 * wave;
            // epilogue of one finished view tile `tp` held in `pv`: FiLM + sine, rgb-head partials, and either the
            // per-ray feature composite partials (render) or the raw feature rows (points mode)
            f32x16 pv;
            float fa0 = 0.f, fa1 = 0.f, fa2 = 0.f;
            float e_gm = 0.f, e_bt = 0.f, e_w0 = 0.f, e_w1 = 0.f, e_w2 = 0.f;
            int e_n = 0;
            const float* __restrict__ wqh = wq_s + half * 16;
            f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0, q0n = q0, q1n = q0, q2n = q0;   // composite weights of a row quad
            auto epi_begin = [&](int tp) {
                e_n = 32 * tp + col;
                e_gm = film_v[e_n]; e_bt = film_v[kWidth + e_n];
                e_w0 = wrgb[e_n]; e_w1 = wrgb[kWidth + e_n]; e_w2 = wrgb[2 * kWidth + e_n];
                fa0 = fa1 = fa2 = 0.f;
                if (MODE == 0) {
                    q0 = *reinterpret_cast<const f32x4*>(wqh); q1 = *reinterpret_cast<const f32x4*>(wqh + 32);
                    q2 = *reinterpret_cast<const f32x4*>(wqh + 64);
                }
            };
            auto epi_r = [&](int r) {
                if (MODE == 0 && (r & 3) == 0 && r < 12) {      // next quad, one quad ahead of its use
                    q0n = *reinterpret_cast<const f32x4*>(wqh + r + 4); q1n = *reinterpret_cast<const f32x4*>(wqh + 32 + r + 4);
                    q2n = *reinterpret_cast<const f32x4*>(wqh + 64 + r + 4);
                }
                const float varg = fmaf(e_gm, pv[r], e_bt);
                const float h = sin_f32(varg);
                if (SAVE) {
                    const int pr = slab_p0 + row_of(r, half);
                    if (pr < npts) a.save_args[((gpt_block + pr) * 9 + 8) * kWidth + e_n] = varg;
                }
                prgb[0][r] = fmaf(e_w0, h, prgb[0][r]);
                prgb[1][r] = fmaf(e_w1, h, prgb[1][r]);
                prgb[2][r] = fmaf(e_w2, h, prgb[2][r]);
                if (MODE == 0) {
                    fa0 = fmaf(q0[r & 3], h, fa0);
                    fa1 = fmaf(q1[r & 3], h, fa1);
                    fa2 = fmaf(q2[r & 3], h, fa2);
                    if ((r & 3) == 3) { q0 = q0n; q1 = q1n; q2 = q2n; }
                } else if (a.raw) {
                    const int pr = slab_p0 + row_of(r, half);
                    if (pr < npts) a.raw[((int64_t)b * a.n_pts + pt0 + pr) * 260 + 4 + e_n] = h;
                }
            };
            auto epi_end = [&]() {
                if (MODE == 0) {
                    fa0 += xhalf(fa0); fa1 += xhalf(fa1); fa2 += xhalf(fa2);
                    if (half == 0) {
                        float* pp = part + (wave * kMaxSlots) * kWidth + e_n;
                        if (slab_nslots > 0) pp[0] = fa0;
                        if (slab_nslots > 1) pp[kWidth] = fa1;
                        if (slab_nslots > 2) pp[2 * kWidth] = fa2;
                    }
                }
            };
#pragma unroll 1
            for (int t = 0; t < kNT; ++t) {
                f32x16 acc = zero16();
                if (!F16) {
                    if (t == 0) {
                        acc = big_tile<true, 0>(pipe.wcur, pipe.wnxt, lane, in, acc, ring, NoEpilogue(), chunk_sync, issue_piece);
                    } else {
                        epi_begin(t - 1);
                        acc = big_tile<true, E3DGE_SPREAD_VIEW>(pipe.wcur, pipe.wnxt, lane, in, acc, ring, epi_r, chunk_sync, issue_piece);
                        epi_end();
                    }
                    acc = mfma32(a0, wvt[(t * 2 + 0) * 64 + lane], acc);
                    acc = mfma32(a1, wvt[(t * 2 + 1) * 64 + lane], acc);
                } else {
                    f32x16 accb = zero16();
                    if (t == 0) {
                        big_tile_f16<true>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL, NoEpilogue(), chunk_sync, issue_piece);
                    } else {
                        epi_begin(t - 1);
                        big_tile_f16<true>(pipe.wcur, pipe.wnxt, lane, inH, inL, acc, accb, ringH, ringL, epi_r, chunk_sync, issue_piece);
                        epi_end();
                    }
                    // view-direction tail in fp32, carrying the same 128 scale as the streamed weights
                    acc = mfma32(a0 * kW16Scale, wvt[(t * 2 + 0) * 64 + lane], acc);
                    accb = mfma32(a1 * kW16Scale, wvt[(t * 2 + 1) * 64 + lane], accb);
                    acc = acc + accb;
                }
                advance_chunk();
                pv = acc;
                asm volatile("" : "+v"(pv));
            }
            epi_begin(kNT - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) epi_r(r);
            epi_end();
        }

        PHASE_MARK(4);
        // =====================================================================================
        // 6. rgb head (:235): reduce the per-lane partials over the 32 feature lanes of each half
        // =====================================================================================
        // Transpose-reduce over the 32 feature lanes of each half: at every step a lane keeps half of its rows and
        // hands the other half to its xor-partner, so the row count halves while the lane span doubles (8+4+2+1+1
        // shuffles per channel instead of 5 x 16).  Afterwards lane `col` (and col^1) holds the total of row col >> 1.
        float rgbv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float q8[8], q4[4], q2[2], q1;
            const bool b4 = col & 16, b3 = col & 8, b2 = col & 4, b1 = col & 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float keep = b4 ? prgb[c][8 + i] : prgb[c][i], send = b4 ? prgb[c][i] : prgb[c][8 + i];
                q8[i] = keep + __shfl_xor(send, 16, kWave);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float keep = b3 ? q8[4 + i] : q8[i], send = b3 ? q8[i] : q8[4 + i];
                q4[i] = keep + __shfl_xor(send, 8, kWave);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float keep = b2 ? q4[2 + i] : q4[i], send = b2 ? q4[i] : q4[2 + i];
                q2[i] = keep + __shfl_xor(send, 4, kWave);
            }
            {
                const float keep = b1 ? q2[1] : q2[0], send = b1 ? q2[0] : q2[1];
                q1 = keep + __shfl_xor(send, 2, kWave);
            }
            q1 += __shfl_xor(q1, 1, kWave);
            rgbv[c] = q1 + head_s[4 * kWidth + 1 + c];
        }
        const int my_row = row_of(col >> 1, half);                 // the slab row whose rgb this lane now holds
        if (MODE == 0) {
            if ((col & 1) == 0) {
                const int ps = 32 * wave + my_row;
                rgb_s[ps * 3 + 0] = sigmoid_f32(rgbv[0]);
                rgb_s[ps * 3 + 1] = sigmoid_f32(rgbv[1]);
                rgb_s[ps * 3 + 2] = sigmoid_f32(rgbv[2]);
            }
            __syncthreads();
            // rgb composite (:888-890), sequential per ray, and ordered merge of the feature partials
            const int sub_lo = sub * kTilePts;
            const int sub_hi = min(sub_lo + kTilePts, npts);
            const int r_first = sub_lo / S, r_last = (sub_hi - 1) / S;
            for (int i = tid; i <= r_last - r_first; i += kThreads) {
                const int rl = r_first + i;
                const int s_lo = max(0, sub_lo - rl * S), s_hi = min(S, sub_hi - rl * S);
                float* st = state + rl * kStateStride;
                float c0 = st[6], c1 = st[7], c2 = st[8];
                for (int s0 = s_lo; s0 < s_hi; s0 += 4) {
                    float ww[4], q0[4], q1[4], q2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ps = min(rl * S + s0 + u, sub_hi - 1) - sub_lo;
                        ww[u] = (s0 + u < s_hi) ? wgt_s[ps] : 0.0f;
                        q0[u] = rgb_s[ps * 3 + 0]; q1[u] = rgb_s[ps * 3 + 1]; q2[u] = rgb_s[ps * 3 + 2];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (s0 + u < s_hi) {
                            c0 = __fadd_rn(c0, __fmul_rn(ww[u], q0[u]));
                            c1 = __fadd_rn(c1, __fmul_rn(ww[u], q1[u]));
                            c2 = __fadd_rn(c2, __fmul_rn(ww[u], q2[u]));
                        }
                }
                st[6] = c0; st[7] = c1; st[8] = c2;
            }
            {
                const int n = tid;   // 256 threads <-> 256 features
                for (int wv = 0; wv < 4; ++wv) {
                    const int slab_lo = sub_lo + 32 * wv;
                    const int slab_hi = min(slab_lo + 32, npts);
                    if (slab_hi <= slab_lo) break;
                    const int fr = slab_lo / S;
                    const int ns = (slab_hi - 1) / S - fr + 1;
                    for (int sl = 0; sl < ns; ++sl)
                        feat_acc[(fr + sl) * kFPitch + n] += part[(wv * kMaxSlots + sl) * kWidth + n];
                }
            }
            __syncthreads();
        } else {
            if (a.raw || a.sdf) {
                if (a.raw && (col & 1) == 0) {
                    const int pr = sub * kTilePts + 32 * wave + my_row;
                    if (pr < npts) {
                        float* o = a.raw + ((int64_t)b * a.n_pts + pt0 + pr) * 260;
                        o[0] = rgbv[0]; o[1] = rgbv[1]; o[2] = rgbv[2];
                    }
                }
                if (valid && half == 0) {
                    if (a.sdf) a.sdf[gpt] = sdf;
                    if (a.raw) a.raw[gpt * 260 + 3] = sdf;
                }
            }
        }
        PHASE_MARK(5);
    }  // sub-tiles

    // make sure no LDS-DMA is still in flight when the workgroup retires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (MODE == 0) {
        // =====================================================================================
        // 7. per-ray outputs, channel-first like VolumeFeatureRenderer.forward returns them (:1957-1968)
        // =====================================================================================
        const int HW = a.H * a.Wd;
        if (a.features) {
            float* fo = a.features + (int64_t)b * kWidth * HW + pix0;
            for (int e = tid; e < nrays * kWidth; e += kThreads) {
                const int n = e / nrays, rl = e - n * nrays;        // consecutive lanes -> consecutive rays
                fo[(int64_t)n * HW + rl] = feat_acc[rl * kFPitch + n];
            }
        }
        for (int rl = tid; rl < nrays; rl += kThreads) {
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
        __syncthreads();
        if (blockIdx.x == 0 && tid == 0 && a.dists)
            for (int i = 0; i < 18; ++i)
                a.dists[i] = (i % 6 == 0) ? (float)(i / 6 ? tstamp[i] - tstamp[i - 1] : 0) : (float)(tstamp[i] - tstamp[i - 1]);
        if (blockIdx.x == 0 && (tid & 63) == 0 && a.dists) {
            a.dists[18 + wave * 3 + 0] = (float)pipe.t_vm; a.dists[18 + wave * 3 + 1] = (float)pipe.t_bar; a.dists[18 + wave * 3 + 2] = 0.0f;
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the pipeline fetches two chunks past the end of the work
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
siren_pack_kernel(float* __restrict__ packed, const float* __restrict__ w_first,
                  const float* __restrict__ b_first, const float* __restrict__ w_hidden,
                  const float* __restrict__ b_hidden, const float* __restrict__ w_view,
                  const float* __restrict__ b_view, const float* __restrict__ w_rgb,
                  const float* __restrict__ b_rgb, const float* __restrict__ w_sigma,
                  const float* __restrict__ b_sigma) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < kPackedFloats; e += (int64_t)gridDim.x * 256) {
        float v;
        if (e < kOffFirst) {
            // [Lb][t][c][q][lane][j] = W_Lb[32t + (lane&31)][32c + 8q + 4(lane>>5) + j]
            int64_t r = e;
            const int j = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int q = r & 3; r >>= 2;
            const int c = r & 7; r >>= 3;
            const int t = r & 7; r >>= 3;
            const int Lb = (int)r;
            const int n = 32 * t + (lane & 31), k = 32 * c + 8 * q + 4 * (lane >> 5) + j;
            v = (Lb < 7) ? w_hidden[((int64_t)Lb * kWidth + n) * kWidth + k] : w_view[(int64_t)n * 259 + k];
        } else if (e < kOffBias) {
            // [t][m][lane]: m=0 -> (half0: col 0, half1: col 1); m=1 -> (half0: col 2, half1: 0)
            const bool vt = e >= kOffVTail;
            int r = (int)(e - (vt ? kOffVTail : kOffFirst));
            const int lane = r & 63; r >>= 6;
            const int m = r & 1; r >>= 1;
            const int t = r;
            const int n = 32 * t + (lane & 31), hf = lane >> 5;
            const int kcol = m * 2 + hf;
            if (kcol >= 3) v = 0.0f;
            else v = vt ? w_view[(int64_t)n * 259 + 256 + kcol] : w_first[n * 3 + kcol];
        } else if (e < kOffWSigma) {
            const int r = (int)(e - kOffBias);
            const int L = r / kWidth, n = r - L * kWidth;
            v = (L == 0) ? b_first[n] : (L < 8 ? b_hidden[(L - 1) * kWidth + n] : b_view[n]);
        } else if (e < kOffWRgb) {
            v = w_sigma[e - kOffWSigma];
        } else if (e < kOffBHead) {
            v = w_rgb[e - kOffWRgb];
        } else if (e < kOffBig16) {
            const int r = (int)(e - kOffBHead);
            v = (r == 0) ? b_sigma[0] : b_rgb[r - 1];
        } else if (e >= kOffBigT16b) {
            // transposed 16x16x32 image (siren_common.h): [Gb][16 t][8 g][hl][lane][word k], out row kin = 16t + (lane & 15)
            int64_t r = e - kOffBigT16b;
            const int k = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int hl = r & 1; r >>= 1;
            const int g = r & 7; r >>= 3;
            const int t = r & 15; r >>= 4;
            const int L = 8 - (int)r;
            const int kin = 16 * t + (lane & 15), q = lane >> 4;
            unsigned word = 0;
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j = 2 * k + e2;
                const int nn = 32 * g + 16 * (j >> 2) + 4 * q + (j & 3);
                const float w = kW16Scale * ((L == 8) ? w_view[(int64_t)nn * 259 + kin] : w_hidden[((int64_t)(L - 1) * kWidth + nn) * kWidth + kin]);
                const _Float16 hi = (_Float16)w;
                const _Float16 val = hl ? (_Float16)(w - (float)hi) : hi;
                word |= (unsigned)__builtin_bit_cast(unsigned short, val) << (16 * e2);
            }
            v = __uint_as_float(word);
        } else if (e >= kOffBig16b) {
            // 16x16x32 image (siren_common.h): [Lb][16 t][8 g][hl][lane][word k]
            int64_t r = e - kOffBig16b;
            const int k = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int hl = r & 1; r >>= 1;
            const int g = r & 7; r >>= 3;
            const int t = r & 15; r >>= 4;
            const int Lb = (int)r;
            const int n = 16 * t + (lane & 15), q = lane >> 4;
            unsigned word = 0;
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j = 2 * k + e2;
                const int kk = 32 * g + 16 * (j >> 2) + 4 * q + (j & 3);
                const float w = kW16Scale * ((Lb < 7) ? w_hidden[((int64_t)Lb * kWidth + n) * kWidth + kk] : w_view[(int64_t)n * 259 + kk]);
                const _Float16 hi = (_Float16)w;
                const _Float16 val = hl ? (_Float16)(w - (float)hi) : hi;
                word |= (unsigned)__builtin_bit_cast(unsigned short, val) << (16 * e2);
            }
            v = __uint_as_float(word);
        } else if (e >= kOffBigT16) {
            // transposed f16x3 image: out row kin = 32t + (lane & 31), contraction index nn in the k-slot order of kOffBig16
            int64_t r = e - kOffBigT16;
            const int k = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int hl = r & 1; r >>= 1;
            const int g = r & 15; r >>= 4;
            const int t = r & 7; r >>= 3;
            const int L = 8 - (int)r;
            const int kin = 32 * t + (lane & 31);
            unsigned word = 0;
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j = 2 * k + e2;
                const int nn = 32 * (g >> 1) + 16 * (g & 1) + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                const float w = kW16Scale * ((L == 8) ? w_view[(int64_t)nn * 259 + kin] : w_hidden[((int64_t)(L - 1) * kWidth + nn) * kWidth + kin]);
                const _Float16 hi = (_Float16)w;
                const _Float16 val = hl ? (_Float16)(w - (float)hi) : hi;
                word |= (unsigned)__builtin_bit_cast(unsigned short, val) << (16 * e2);
            }
            v = __uint_as_float(word);
        } else if (e >= kOffBigT) {
            // transposed fp32 image for the backward chain (siren_common.h)
            int64_t r = e - kOffBigT;
            const int j = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int q = r & 3; r >>= 2;
            const int c = r & 7; r >>= 3;
            const int t = r & 7; r >>= 3;
            const int L = 8 - (int)r;
            const int kin = 32 * t + (lane & 31), n = 32 * c + 8 * q + 4 * (lane >> 5) + j;
            v = (L == 8) ? w_view[(int64_t)n * 259 + kin] : w_hidden[((int64_t)(L - 1) * kWidth + n) * kWidth + kin];
        } else {
            // one 32-bit word = two f16: [Lb][t][g = 2c+s][hl][lane][word k]  ->  j = 2k, 2k+1
            int64_t r = e - kOffBig16;
            const int k = r & 3; r >>= 2;
            const int lane = r & 63; r >>= 6;
            const int hl = r & 1; r >>= 1;
            const int g = r & 15; r >>= 4;
            const int t = r & 7; r >>= 3;
            const int Lb = (int)r;
            const int n = 32 * t + (lane & 31);
            unsigned word = 0;
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j = 2 * k + e2;
                const int kk = 32 * (g >> 1) + 16 * (g & 1) + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                const float w = kW16Scale * ((Lb < 7) ? w_hidden[((int64_t)Lb * kWidth + n) * kWidth + kk] : w_view[(int64_t)n * 259 + kk]);
                const _Float16 hi = (_Float16)w;
                const _Float16 val = hl ? (_Float16)(w - (float)hi) : hi;
                word |= (unsigned)__builtin_bit_cast(unsigned short, val) << (16 * e2);
            }
            v = __uint_as_float(word);
        }
        packed[e] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// FiLM parameters: one wave per output row, lanes along K (16 B/lane coalesced), xor-reduce
// ---------------------------------------------------------------------------------------------
constexpr int kFilmRowsPerBlock = 16;      // 4 waves x 4 rows: 16 x 18 = 288 workgroups per image (was 18 -> latency-bound)

__global__ void __launch_bounds__(256)
film_params_kernel(float* __restrict__ film, const float* __restrict__ styles,
                   const float* __restrict__ wg, const float* __restrict__ bg,
                   const float* __restrict__ wb, const float* __restrict__ bb) {
    // grid: batch * 9 * 2 * (256 / kFilmRowsPerBlock)
    int id = blockIdx.x;
    const int rg = id % (kWidth / kFilmRowsPerBlock); id /= (kWidth / kFilmRowsPerBlock);
    const int which = id & 1; id >>= 1;
    const int l = id % 9;
    const int b = id / 9;
    const float* __restrict__ Wm = (which ? wb : wg) + (int64_t)l * kWidth * kWidth;
    const float* __restrict__ bv = (which ? bb : bg) + l * kWidth;
    const float* __restrict__ s = styles + ((int64_t)b * 9 + l) * kWidth;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(s + lane * 4);
    const float std_init = which ? 0.25f : 15.0f, bias_init = which ? 0.0f : 30.0f;
    float* __restrict__ o = film + (((int64_t)b * 9 + l) * 2 + which) * kWidth;
    f32x4 w4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w4[i] = *reinterpret_cast<const f32x4*>(Wm + (int64_t)(rg * kFilmRowsPerBlock + wave * 4 + i) * kWidth + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = rg * kFilmRowsPerBlock + wave * 4 + i;
        float acc = w4[i][0] * s4[0];
        acc = fmaf(w4[i][1], s4[1], acc);
        acc = fmaf(w4[i][2], s4[2], acc);
        acc = fmaf(w4[i][3], s4[3], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
        if (lane == 0) o[n] = __fadd_rn(__fmul_rn(std_init, __fadd_rn(acc, bv[n])), bias_init);   // LinearLayer.forward :76-80
    }
}

// ---------------------------------------------------------------------------------------------
// self tests
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
selftest_mfma_kernel(float* __restrict__ cmat, const float* __restrict__ amat,
                     const float* __restrict__ bmat, int k) {
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int kb = 0; kb < k; kb += 8) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(amat + col * k + kb + 4 * half);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bmat + col * k + kb + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma32(a4[j], b4[j], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cmat[row_of(r, half) * 32 + col] = acc[r];
}

// Same for v_mfma_f32_32x32x16_f16 with the k-slot convention of the f16x3 path: lane l supplies, for k-step g,
// the 8 values k = 16g + 8(l>>5) + j of row (A) / column (B) l&31.
__global__ void __launch_bounds__(64)
selftest_mfma16_kernel(float* __restrict__ cmat, const float* __restrict__ amat,
                       const float* __restrict__ bmat, int k) {
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    f32x16 acc = zero16();
    for (int kb = 0; kb < k; kb += 16) {
        u32x4 a4, b4, dummy;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            SPLIT2_TO(amat[col * k + kb + 8 * half + 2 * w], amat[col * k + kb + 8 * half + 2 * w + 1], a4[w], dummy[w]);
            SPLIT2_TO(bmat[col * k + kb + 8 * half + 2 * w], bmat[col * k + kb + 8 * half + 2 * w + 1], b4[w], dummy[w]);
        }
        acc = mfma16(a4, b4, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cmat[row_of(r, half) * 32 + col] = acc[r];
}

__global__ void selftest_sin_kernel(float* __restrict__ y, const float* __restrict__ x, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = mode ? sin_poly_f32(x[i]) : sin_f32(x[i]);
}

template <int MODE>
static int launch_siren(const SirenK& k, int precision, int64_t grid, hipStream_t st) {
    typedef void (*KernelFn)(const SirenK);
    const int save = k.save_args != nullptr;
    if (precision == E3DGE_PREC_F16X3) {
        // second-generation split-f16 kernel: 8 waves x 16 points, v_mfma_f32_16x16x32_f16 (siren16.h)
        static const KernelFn fns16[2] = {&siren16_kernel<MODE, false>, &siren16_kernel<MODE, true>};
        KernelFn fn = fns16[save];
        if constexpr (MODE == 0) {
            if (k.bb_out) fn = &siren16_kernel<0, false, 1>;
            else if (k.bb_in) fn = &siren16_kernel<0, false, 2>;
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, k16LdsBytes);
        if (e != hipSuccess)
            return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", k16LdsBytes, hipGetErrorString(e));
        fn<<<dim3((unsigned)grid), dim3(k16Threads), k16LdsBytes, st>>>(k);
        return E3DGE_OK;
    }
#ifdef E3DGE_EXPERIMENTAL   // the first-generation split-f16 forward (4 waves x 32 points; 48-52 B of scratch): A/B builds only
    static const KernelFn fns[4] = {&siren_kernel<MODE, 0, false>, &siren_kernel<MODE, 0, true>,
                                    &siren_kernel<MODE, 1, false>, &siren_kernel<MODE, 1, true>};
    const KernelFn fn = fns[2 * (precision == E3DGE_PREC_F16X3_V1) + save];
#else
    static const KernelFn fns[2] = {&siren_kernel<MODE, 0, false>, &siren_kernel<MODE, 0, true>};
    if (precision == E3DGE_PREC_F16X3_V1)
        return fail(E3DGE_ERR_INVALID_ARG, "precision f16x3_v1 (first-generation split-f16 forward) is only in -DE3DGE_EXPERIMENTAL builds");
    const KernelFn fn = fns[save];
#endif
    // the attribute is per device; setting it on every launch is cheap and keeps multi-GPU processes correct
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess)
        return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", kLdsBytes, hipGetErrorString(e));
    fn<<<dim3((unsigned)grid), dim3(kThreads), kLdsBytes, st>>>(k);
    return E3DGE_OK;
}

// rays per workgroup: the largest R <= kRMax whose R*S is a multiple of 128 if one exists (no padded
// lanes), else the R <= kRMax minimising padding; small images get fewer rays per workgroup so that the
// grid still covers the 256 CUs.
static int pick_rays_per_wg(int S, int64_t total_rays) {
    int best = 1;
    double best_cost = 1e30;
    for (int R = 1; R <= kRMax; ++R) {
        const int pts = R * S;
        const int nsub = (pts + kTilePts - 1) / kTilePts;
        const double pad = (double)(nsub * kTilePts) / pts;                 // >= 1
        const int64_t wgs = (total_rays + R - 1) / R;
        const int64_t rounds = (wgs + 255) / 256;
        const double fill = (double)(rounds * 256) / (double)wgs;           // >= 1, tail effect
        const double cost = pad * fill * (1.0 + 0.02 / nsub);               // slight preference for longer blocks
        if (cost < best_cost - 1e-12) { best_cost = cost; best = R; }
    }
    return best;
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int64_t e3dge_siren_packed_floats(void) { return kPackedFloats; }

extern "C" int e3dge_siren_pack_weights(float* packed, const float* w_first, const float* b_first,
                                        const float* w_hidden, const float* b_hidden,
                                        const float* w_view, const float* b_view, const float* w_rgb,
                                        const float* b_rgb, const float* w_sigma, const float* b_sigma,
                                        e3dge_stream_t stream) {
    E3DGE_REQUIRE(packed && w_first && b_first && w_hidden && b_hidden && w_view && b_view && w_rgb && b_rgb &&
                  w_sigma && b_sigma, "siren_pack_weights: null pointer");
    E3DGE_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "siren_pack_weights: packed must be 16-B aligned");
    siren_pack_kernel<<<dim3(512), dim3(256), 0, as_stream(stream)>>>(packed, w_first, b_first, w_hidden, b_hidden,
                                                                      w_view, b_view, w_rgb, b_rgb, w_sigma, b_sigma);
    return check_launch("siren_pack_weights");
}

extern "C" int e3dge_film_params(float* film, const float* styles, const float* wg, const float* bg,
                                 const float* wb, const float* bb, int batch, e3dge_stream_t stream) {
    E3DGE_REQUIRE(film && styles && wg && bg && wb && bb, "film_params: null pointer");
    E3DGE_REQUIRE(batch >= 0, "film_params: batch=%d", batch);
    if (batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(styles) | reinterpret_cast<uintptr_t>(wg) |
                    reinterpret_cast<uintptr_t>(wb)) & 15) == 0, "film_params: inputs must be 16-B aligned");
    film_params_kernel<<<dim3((unsigned)(batch * 9 * 2 * (kWidth / kFilmRowsPerBlock))), dim3(256), 0, as_stream(stream)>>>(film, styles, wg, bg, wb, bb);
    return check_launch("film_params");
}

extern "C" int e3dge_siren_render_fwd(const E3dgeRenderArgs* r, e3dge_stream_t stream) {
    E3DGE_REQUIRE(r != nullptr, "siren_render_fwd: null args");
    E3DGE_REQUIRE(r->batch >= 0 && r->height > 0 && r->width > 0, "siren_render_fwd: bad image extent");
    if (r->batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(r->packed && r->film && r->c2w && r->focal && r->near && r->far && r->t_vals,
                  "siren_render_fwd: null input pointer");
    E3DGE_REQUIRE(r->n_samples >= kMinSamples && r->n_samples <= 4096,
                  "siren_render_fwd: n_samples=%d outside [%d, 4096] (use e3dge_siren_points_fwd for raw queries)",
                  r->n_samples, kMinSamples);
    E3DGE_REQUIRE((r->tex_alpha == nullptr) == (r->tex_beta == nullptr), "siren_render_fwd: tex_alpha/tex_beta must come together");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(r->packed) | reinterpret_cast<uintptr_t>(r->film) |
                    reinterpret_cast<uintptr_t>(r->tex_alpha) | reinterpret_cast<uintptr_t>(r->tex_beta)) & 15) == 0,
                  "siren_render_fwd: packed/film/tex pointers must be 16-B aligned");
    E3DGE_REQUIRE(r->sigmoid_beta != 0.0f, "siren_render_fwd: sigmoid_beta must be non-zero");
    E3DGE_REQUIRE(r->precision >= E3DGE_PREC_F32 && r->precision <= E3DGE_PREC_F16X3_G2, "siren_render_fwd: precision=%d", r->precision);
    if (r->batch == 0) return E3DGE_OK;
    const int64_t HW = (int64_t)r->height * r->width;
    E3DGE_REQUIRE(HW * r->batch * (int64_t)r->n_samples < ((int64_t)1 << 40), "siren_render_fwd: too many points");
    SirenK k{};
    k.packed = r->packed; k.film = r->film; k.c2w = r->c2w; k.focal = r->focal; k.near = r->near; k.far = r->far;
    k.t_vals = r->t_vals; k.tex_alpha = r->tex_alpha; k.tex_beta = r->tex_beta;
    k.sigmoid_beta = r->sigmoid_beta; k.box_scale = r->box_scale; k.mask_thresh = r->mask_depth_thresh;
    k.batch = r->batch; k.H = r->height; k.Wd = r->width; k.S = r->n_samples; k.res = r->res; k.force_bg = r->force_background;
    k.R = pick_rays_per_wg(r->n_samples, HW * r->batch);
    if (k.R > HW) k.R = (int)HW;
    k.tiles_per_img = (int)((HW + k.R - 1) / k.R);
    k.rgb = r->rgb; k.features = r->features; k.xyz = r->xyz; k.depth = r->depth; k.mask = r->mask; k.sdf = r->sdf;
    k.weights = r->weights; k.points = r->points; k.rays_d = r->rays_d; k.viewdirs = r->viewdirs; k.dists = r->dists;
    const int64_t grid = (int64_t)k.tiles_per_img * r->batch;
    E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "siren_render_fwd: grid too large");
    k.save_args = r->save_args;
    // precision f16x3_g2 on a forward launch = the f16x3 kernel, with the saved arguments in the slab-major layout its backward-type
    // kernels read (siren_common.h); without save_args the two are the same launch
    k.save_blocked = (r->precision == E3DGE_PREC_F16X3_G2 && r->save_args) ? 1 : 0;
    const int precision = r->precision == E3DGE_PREC_F16X3_G2 ? E3DGE_PREC_F16X3 : r->precision;
    if (r->backbone_out || r->backbone_in) {
        E3DGE_REQUIRE(precision == E3DGE_PREC_F16X3 && !r->save_args, "siren_render_fwd: the backbone hand-over needs precision f16x3 and no save_args");
        E3DGE_REQUIRE(!(r->backbone_out && r->backbone_in), "siren_render_fwd: backbone_out and backbone_in are exclusive");
        E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(r->backbone_out) | reinterpret_cast<uintptr_t>(r->backbone_in)) & 15) == 0, "siren_render_fwd: backbone record must be 16-B aligned");
        if (r->backbone_in)
            E3DGE_REQUIRE(r->weights_in && !r->sdf && !r->weights && !r->xyz && !r->depth && !r->mask,
                          "siren_render_fwd: a launch that reads the backbone record needs weights_in and produces only rgb / features (and the ray geometry)");
        k.bb_out = r->backbone_out; k.bb_in = r->backbone_in; k.weights_in = r->weights_in;
        k.bb_subs = (k.R * r->n_samples + kTilePts - 1) / kTilePts;
    }
    int rc = launch_siren<0>(k, precision, grid, as_stream(stream));
    if (rc) return rc;
    return check_launch("siren_render_fwd");
}

// tiling of a render launch (rays per workgroup, workgroups per image, 128-point sub-tiles per workgroup): what the layer-7 record's
// slab index is made of; shared with the texture head's FILM form (resblock.hip)
void e3dge::siren_record_layout(int batch, int height, int width, int n_samples, int* R, int* tiles_per_img, int* subs) {
    const int64_t HW = (int64_t)height * width;
    int r = pick_rays_per_wg(n_samples, HW * batch);
    if (r > HW) r = (int)HW;
    *R = r;
    *tiles_per_img = (int)((HW + r - 1) / r);
    *subs = (r * n_samples + kTilePts - 1) / kTilePts;
}

extern "C" int64_t e3dge_siren_backbone_bytes(int batch, int height, int width, int n_samples) {
    if (batch <= 0 || height <= 0 || width <= 0 || n_samples < kMinSamples) return 0;
    const int64_t HW = (int64_t)height * width;
    int R = pick_rays_per_wg(n_samples, HW * batch);
    if (R > HW) R = (int)HW;
    const int64_t grid = ((HW + R - 1) / R) * batch, subs = (R * (int64_t)n_samples + kTilePts - 1) / kTilePts;
    return grid * subs * 8 * k16SlabWords * 16;
}

extern "C" int e3dge_siren_points_fwd(const float* packed, const float* film, const float* pts,
                                      const float* viewdirs, float box_scale, int batch, int64_t n_pts,
                                      float* sdf, float* raw, float* save_args, int precision, e3dge_stream_t stream) {
    E3DGE_REQUIRE(precision >= E3DGE_PREC_F32 && precision <= E3DGE_PREC_F16X3_G2, "siren_points_fwd: precision=%d", precision);
    E3DGE_REQUIRE(batch >= 0 && n_pts >= 0, "siren_points_fwd: bad sizes");
    if (batch == 0 || n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(packed && film && pts, "siren_points_fwd: null input pointer");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(film)) & 15) == 0,
                  "siren_points_fwd: packed/film must be 16-B aligned");
    if (batch == 0 || n_pts == 0) return E3DGE_OK;
    SirenK k{};
    k.packed = packed; k.film = film; k.pts = pts; k.vdirs = viewdirs; k.box_scale = box_scale;
    k.batch = batch; k.n_pts = n_pts; k.sdf = sdf; k.raw = raw;
    const int64_t tiles = (n_pts + kTilePts - 1) / kTilePts;
    const int spw = pick_subtiles_per_wg(tiles, batch);
    k.subtiles_per_wg = spw;
    k.wgs_per_img = (int)((tiles + spw - 1) / spw);
    const int64_t grid = (int64_t)k.wgs_per_img * batch;
    E3DGE_REQUIRE(grid < ((int64_t)1 << 31), "siren_points_fwd: grid too large");
    k.save_args = save_args;
    k.save_blocked = (precision == E3DGE_PREC_F16X3_G2 && save_args) ? 1 : 0;      // (see e3dge_siren_render_fwd)
    if (precision == E3DGE_PREC_F16X3_G2) precision = E3DGE_PREC_F16X3;
    int rc = launch_siren<1>(k, precision, grid, as_stream(stream));
    if (rc) return rc;
    return check_launch("siren_points_fwd");
}

extern "C" int e3dge_selftest_mfma(float* c, const float* a, const float* b, int k, e3dge_stream_t stream) {
    E3DGE_REQUIRE(c && a && b && k > 0 && k <= 256 && (k % 8) == 0, "selftest_mfma: bad arguments");
    selftest_mfma_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(c, a, b, k);
    return check_launch("selftest_mfma");
}

extern "C" int e3dge_selftest_mfma16(float* c, const float* a, const float* b, int k, e3dge_stream_t stream) {
    E3DGE_REQUIRE(c && a && b && k > 0 && k <= 256 && (k % 16) == 0, "selftest_mfma16: bad arguments");
    selftest_mfma16_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(c, a, b, k);
    return check_launch("selftest_mfma16");
}

#ifdef E3DGE_16_TRACE
extern "C" int e3dge_debug_trace16(unsigned long long* out48) {
    return hipMemcpyFromSymbol(out48, HIP_SYMBOL(e3dge::g_trace16), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int e3dge_selftest_mfma16x16(float* c, const float* a, const float* b, int k, e3dge_stream_t stream) {
    E3DGE_REQUIRE(c && a && b && k > 0 && k <= 256 && (k % 32) == 0, "selftest_mfma16x16: bad arguments");
    selftest_mfma16x16_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(c, a, b, k);
    return check_launch("selftest_mfma16x16");
}

extern "C" int e3dge_selftest_sin(float* y, const float* x, int n, e3dge_stream_t stream) {
    E3DGE_REQUIRE(y && x && n >= 0, "selftest_sin: bad arguments");
    if (n == 0) return E3DGE_OK;
    selftest_sin_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(y, x, n, 0);
    return check_launch("selftest_sin");
}

extern "C" int e3dge_selftest_sin_poly(float* y, const float* x, int n, e3dge_stream_t stream) {
    E3DGE_REQUIRE(y && x && n >= 0, "selftest_sin_poly: bad arguments");
    if (n == 0) return E3DGE_OK;
    selftest_sin_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(y, x, n, 1);
    return check_launch("selftest_sin_poly");
}
