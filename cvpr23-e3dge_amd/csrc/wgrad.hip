// Parameter gradients of the local branch's fully connected layers:  C (m, n) = sum_p A[p, :m]^T  f(B[p, :n]),  f = identity | relu,
// with A = the layer's output gradient and B = its input, both (n_rows, ld) fp32 rows -- what autograd runs for every nn.Linear of
// ResnetBlockFC (helper_modules/resnetfc.py:49-58) and Fuse_sft_MLP (helper_modules/sft.py:84-110) in the stage-2 step
// (e3dge_full_runner.py:185-317): `grad_output.t() @ input`, and beside it `grad_output.sum(0)` for the bias.  The contraction runs over
// the POINTS (98,304 per image at 64 x 64 x 24), the outputs are at most 512 x 512 -- a split-K problem: a workgroup owns one output block
// and one slab of points, partial blocks go to a workspace and a second launch folds the slabs in fixed order (bit-reproducible, no atomics).
//
// Round 6 (second generation).  Round 5's machine (4 waves, 128 x 128 block, one 32-point LDS stage, conversion and contraction in
// separate phases) ran at 0.23 of f16 / 3: (a) every operand column was read by two blocks whose workgroups sat on different XCDs
// (400 MB through HBM for a 256 x 256 gradient), (b) ~10 VALU instructions per MFMA, none of them overlapped with the matrix pipe.  Now:
//   * 8 waves own a block of up to 256 x 256 (wave grid 2 x 4, a wave MI x NJ tiles of 32 x 32): a 256 x 256 layer reads each operand
//     ONCE; the VALU work per MFMA halves.  Tiles that lie wholly outside (m, n) are skipped wave-uniformly (301 columns = 2 x 256 with
//     the last five of eight tile columns idle; covers with less padding -- 256 x 128 and 128 x 128 blocks, kept as E3DGE_WGRAD_SHAPE -- measured slower).
//   * steps of 16 points (one k-step of v_mfma_f32_32x32x16_f16), TWO LDS stages and TWO register sets: a step converts and stores the slice
//     of step + 1, which was requested two steps earlier (one step of ~2 k cycles did not cover HBM latency: 82 us per 256 x 256 layer)
//     (split-f16, transposed to [feature][point], row pitch 48 B: conflict-free 16-byte fragment reads; the 4-byte stores are 2-way
//     conflicted, which their 4-cycle register transfer hides), requests the slice of step + 3 and multiplies the slice of step -- ONE
//     barrier per step, the conversion of one wave runs under the MFMAs of the other wave of its SIMD.
//   * workgroup ids are mapped so that the blocks of one slab are neighbours on one XCD (they read the same rows).
//   * the column sums of A (the bias gradient) and one extra column  sum_p A[p, :] f(xcol[p])  (the visibility-mask column of the
//     513-wide Fuse_sft_MLP input: the block grid then skips that column of B -- `b_gap`) come from the values the conversion already
//     holds: per-slab partials, folded in the same fixed order.
#include <type_traits>
#include "decoder_common.h"

namespace e3dge {

constexpr int kWgPts = 16;                  // points per step (one MFMA k-step)
constexpr int kWgPitch = 48;                // bytes per feature row of one half (32 + 16 pad)
constexpr int kWgThreads = 512;

struct WgradK {
    const float* a; const float* amax_a; const float* b; const float* amax_b; const float* xcol;
    float* ws; float* ws_col;
    long long n_rows, slab;
    int lda, off_a, m, ldb, off_b, n, relu_b, mb, nb, gap_at, gap, ld_xcol, cols;       // cols: bit 0 = column sums, bit 1 = xcol
};

struct __attribute__((packed, aligned(4))) WgU4 { float v[4]; };             // 16-byte access at 4-byte alignment (rows of any pitch)
// where this thread reads: its first row of the current step and how many of its quad's four columns exist (4 = the whole quad)
struct WgSrc { const float* row; long long ld; int have; };

// thread (pp, fq): points 2 pp, 2 pp + 1 of the step, feature quad fq of the block (64 quads = 256 features)
__device__ __forceinline__ void wg_src_init(WgSrc& s, const float* base, int ld, int width, int f0, int f0_phys, int blk_w, long long p_begin, int pp, int fq) {
    s.ld = ld;
    s.row = base + (p_begin + 2 * pp) * (long long)ld + f0_phys + 4 * fq;
    const int left = width - (f0 + 4 * fq);
    s.have = (4 * fq >= blk_w || left <= 0) ? 0 : (left > 4 ? 4 : left);
}

// ROWS = false: all 16 points of the step exist (every step but the last of the last slab)
template <bool ROWS>
__device__ __forceinline__ void wg_load(f32x4 (&v)[2], const WgSrc& src, long long p0, long long p_end, int pp) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ p = src.row + e * src.ld;
        if (src.have && (!ROWS || p0 + 2 * pp + e < p_end)) {
            if (src.have == 4) {
                const WgU4 u = *reinterpret_cast<const WgU4*>(p);
                x = f32x4{u.v[0], u.v[1], u.v[2], u.v[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) if (j < src.have) x[j] = p[j];
            }
        }
        v[e] = x;
    }
}

__device__ __forceinline__ void wg_store(const f32x4 (&v)[2], unsigned char* __restrict__ hi, unsigned char* __restrict__ lo, float sc, bool relu,
                                         int pp, int fq) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0 = v[0][j] * sc, x1 = v[1][j] * sc;
        if (relu) { x0 = fmaxf(x0, 0.0f); x1 = fmaxf(x1, 0.0f); }
        const HiLo w = split2(x0, x1);
        const int off = (4 * fq + j) * kWgPitch + 4 * pp;
        *reinterpret_cast<unsigned*>(hi + off) = w.h;
        *reinterpret_cast<unsigned*>(lo + off) = w.l;
    }
}

// consecutive logical ids on one XCD (hardware deals workgroup t to XCD t & 7)
__device__ __forceinline__ int wg_xcd_logical(int t, int n) {
    const int nq = n >> 3, nr = n & 7, xcd = t & 7, slot = t >> 3;
    return (xcd < nr ? xcd * (nq + 1) : nr * (nq + 1) + (xcd - nr) * nq) + slot;
}

template <int MI, int NJ>
constexpr int wg_lds_bytes() { return 2 * 2 * (64 * MI + 128 * NJ) * kWgPitch; }

template <int MI, int NJ, int MINB>
__global__ void __launch_bounds__(kWgThreads, MINB) wgrad_kernel(const WgradK a) {
    constexpr int BM = 64 * MI, BN = 128 * NJ, HALF_A = BM * kWgPitch, HALF_B = BN * kWgPitch, STAGE = 2 * (HALF_A + HALF_B);
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int pp = tid & 7, fq = tid >> 3;                    // (a wave: 8 point pairs x 8 quads)
    const int nblk = a.mb * a.nb;
    const int L = wg_xcd_logical((int)blockIdx.x, (int)gridDim.x);
    const int blk = L % nblk, slab = L / nblk;
    const int bm = blk / a.nb, bn = blk % a.nb;
    const long long p_begin = (long long)slab * a.slab, p_end = min(a.n_rows, p_begin + a.slab);
    const unsigned ea = scale_exponent(amax_read(a.amax_a, lane)), eb = scale_exponent(amax_read(a.amax_b, lane));
    const float sa = __uint_as_float((268u - ea) << 23), sb = __uint_as_float((268u - eb) << 23);       // 2^(141 - e)
    const int wy = wave >> 2, wx = wave & 3;
    const int m_left = a.m - BM * bm, n_left = a.n - BN * bn;
    const int cols = bn == 0 ? a.cols : 0;                   // the column sums ride with the first block column

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = zero16();
    bool live_i[MI], live_j[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) live_i[i] = 32 * (MI * wy + i) < m_left;
#pragma unroll
    for (int j = 0; j < NJ; ++j) live_j[j] = 32 * (NJ * wx + j) < n_left;

    WgSrc src_a, src_b;
    const int f0b = BN * bn, f0b_phys = f0b + (a.gap && f0b >= a.gap_at ? a.gap : 0);
    wg_src_init(src_a, a.a + a.off_a, a.lda, a.m, BM * bm, BM * bm, BM, p_begin, pp, fq);
    wg_src_init(src_b, a.b + a.off_b, a.ldb, a.n, f0b, f0b_phys, BN, p_begin, pp, fq);
    const float* xc = a.xcol ? a.xcol + (p_begin + 2 * pp) * (long long)a.ld_xcol : nullptr;
    // slice of step s: register set s & 1, LDS stage s & 1 (the step loop is unrolled by two: both are compile-time)
    f32x4 ra[2][2], rb[2][2];
    float rx[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cx = {0.f, 0.f, 0.f, 0.f};
    const bool mine_a = 4 * fq < BM, mine_b = 4 * fq < BN;    // (64 quads of threads, BM / 4 and BN / 4 quads of features)

    auto load = [&](long long p0, auto set_c) {
        constexpr int set = decltype(set_c)::value;
        if (p0 + kWgPts <= p_end) { wg_load<false>(ra[set], src_a, p0, p_end, pp); wg_load<false>(rb[set], src_b, p0, p_end, pp); }
        else { wg_load<true>(ra[set], src_a, p0, p_end, pp); wg_load<true>(rb[set], src_b, p0, p_end, pp); }
        if (cols & 2) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float w = (p0 + 2 * pp + e < p_end) ? xc[e * (long long)a.ld_xcol] : 0.0f;
                rx[set][e] = a.relu_b ? fmaxf(w, 0.0f) : w;
            }
        }
        src_a.row += kWgPts * src_a.ld;
        src_b.row += kWgPts * src_b.ld;
        if (xc) xc += kWgPts * (long long)a.ld_xcol;
    };
    auto store = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
        unsigned char* const base = wg_lds + set * STAGE;
        if (mine_a) wg_store(ra[set], base, base + HALF_A, sa, false, pp, fq);
        if (mine_b) wg_store(rb[set], base + 2 * HALF_A, base + 2 * HALF_A + HALF_B, sb, a.relu_b != 0, pp, fq);
        if (cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cs[j] += ra[set][0][j] + ra[set][1][j];
                cx[j] = fmaf(ra[set][0][j], rx[set][0], fmaf(ra[set][1][j], rx[set][1], cx[j]));
            }
        }
    };
    // one step: barrier | convert + store the slice of step + 1 (requested two steps ago) | request the slice of step + 1 + PF | MFMAs of this step
    auto step = [&](long long p0, auto par_c) {
        constexpr int par = decltype(par_c)::value;
        using Other = std::integral_constant<int, par ^ 1>;
        __syncthreads();                                      // stage par is complete; every read of stage par ^ 1 (the previous step) is done
        if (p0 + kWgPts < p_end) {
            store(Other{});
            if (p0 + 3 * kWgPts < p_end) load(p0 + 3 * kWgPts, Other{});
        }
        const unsigned char* const base = wg_lds + par * STAGE;
        u32x4 fah[MI], fal[MI], fbh[NJ], fbl[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ro = (32 * (MI * wy + i) + col) * kWgPitch + 16 * half;
            fah[i] = *reinterpret_cast<const u32x4*>(base + ro);
            fal[i] = *reinterpret_cast<const u32x4*>(base + HALF_A + ro);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = (32 * (NJ * wx + j) + col) * kWgPitch + 16 * half;
            fbh[j] = *reinterpret_cast<const u32x4*>(base + 2 * HALF_A + co);
            fbl[j] = *reinterpret_cast<const u32x4*>(base + 2 * HALF_A + HALF_B + co);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (live_i[i] && live_j[j]) {
                    acc[i][j] = mfma16(fah[i], fbh[j], acc[i][j]);
                    acc[i][j] = mfma16(fal[i], fbh[j], acc[i][j]);
                    acc[i][j] = mfma16(fah[i], fbl[j], acc[i][j]);
                }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    if (p_begin < p_end) {
        load(p_begin, C0{});
        store(C0{});
        if (p_begin + kWgPts < p_end) load(p_begin + kWgPts, C1{});
        if (p_begin + 2 * kWgPts < p_end) load(p_begin + 2 * kWgPts, C0{});
    }
    for (long long p0 = p_begin; p0 < p_end; p0 += 2 * kWgPts) {
        step(p0, C0{});
        if (p0 + kWgPts < p_end) step(p0 + kWgPts, C1{});
    }
    // partial block -> workspace [slab][block][wave][i][j][r / 4][lane][r & 3]: a wave-instruction stores one contiguous KiB (row-major
    // blocks cost 128-byte pieces: 82 us per 256 x 256 layer, a third of it these stores).  The fold sums slot by slot and decodes
    // (row, column) only for the final store: register r of tile (i, j) is row 32 (MI wy + i) + row_of(r, half), column 32 (NJ wx + j) + col.
    f32x4* __restrict__ out = reinterpret_cast<f32x4*>(a.ws + ((long long)slab * nblk + blk) * (BM * BN)) + (wave * MI * NJ) * 256 + lane;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (live_i[i] && live_j[j]) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    out[((i * NJ + j) * 4 + r4) * 64] = f32x4{acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
            }
    if (cols) {
        // this thread's sums cover its two points of every step: fold the eight point pairs (lanes that differ in bits 0..2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) { cs[j] += __shfl_xor(cs[j], off, kWave); cx[j] += __shfl_xor(cx[j], off, kWave); }
        }
        if (pp == 0 && 4 * fq < BM) {
            float* __restrict__ oc = a.ws_col + ((long long)slab * a.mb + bm) * (2 * BM) + 4 * fq;
#pragma unroll
            for (int j = 0; j < 4; ++j) { oc[j] = cs[j]; oc[BM + j] = cx[j]; }
        }
    }
}

// C[m][n] = 2^(ea + eb - 282) sum over slabs (in order) of the partial blocks; behind them (blocks >= n_mat) the column partials
struct WgFoldK {
    float* c; const float* ws; const float* ws_col; float* colsum; float* ccol; const float* amax_a; const float* amax_b;
    int ldc, m, n, mb, nb, bm_sz, bn_sz, n_slabs, gap_at, gap, ld_ccol, n_mat;
};
constexpr int kWgFoldWaves = 16;
__global__ void __launch_bounds__(64 * kWgFoldWaves) wgrad_fold_kernel(const WgFoldK a) {
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= a.n_mat) {
        // column partials: a block owns 64 features; wave w sums the slabs w, w + 16, ... of both kinds, then the same fixed tree
        // (one thread walking all slabs was a chain of 256 dependent loads: +60 us per layer)
        __shared__ float cred[kWgFoldWaves][2][64];
        const int w = threadIdx.x >> 6, y = ((int)blockIdx.x - a.n_mat) * 64 + lane;
        float s0 = 0.0f, s1 = 0.0f;
        if (y < a.m) {
            const float* __restrict__ p = a.ws_col + (long long)(y / a.bm_sz) * (2 * a.bm_sz) + y % a.bm_sz;
            const long long stride = (long long)a.mb * 2 * a.bm_sz;
            for (int s = w; s < a.n_slabs; s += kWgFoldWaves) { s0 += p[s * stride]; s1 += p[s * stride + a.bm_sz]; }
        }
        cred[w][0][lane] = s0; cred[w][1][lane] = s1;
        __syncthreads();
        if (w < 2 && y < a.m) {
            float v = 0.0f;
#pragma unroll
            for (int q = 0; q < kWgFoldWaves; q += 4) v += (cred[q][w][lane] + cred[q + 1][w][lane]) + (cred[q + 2][w][lane] + cred[q + 3][w][lane]);
            if (w == 0 && a.colsum) a.colsum[y] = v;
            if (w == 1 && a.ccol) a.ccol[(long long)y * a.ld_ccol] = v;
        }
        return;
    }
    // a block: the 64 slot quads (one KiB per wave-load) of one (output block, producer wave, tile, r / 4); wave w of the fold's 16 sums the
    // slabs w, w + 16, ... (loads 8 deep), then the 16 sums in a fixed tree.  Slots of tiles outside (m, n) were never written:
    // they are skipped by the same wave-uniform test the producer used.
    __shared__ f32x4 red[kWgFoldWaves][64];
    const unsigned ea = scale_exponent(amax_read(a.amax_a, lane)), eb = scale_exponent(amax_read(a.amax_b, lane));
    const int ee = (int)ea + (int)eb - 282;                                           // 1 / (sa sb) = 2^ee, |ee| can exceed the fp32 exponent range:
    const float f1 = __uint_as_float((unsigned)(127 + ee / 2) << 23), f2 = __uint_as_float((unsigned)(127 + (ee - ee / 2)) << 23);   // two factors
    const int w = threadIdx.x >> 6;
    const int MI = a.bm_sz / 64, NJ = a.bn_sz / 128, per_blk = 8 * MI * NJ * 4;       // wave-slots (KiB pieces) per block
    const int blk = (int)blockIdx.x / per_blk, piece = (int)blockIdx.x % per_blk;
    const int r4 = piece & 3, t = piece >> 2, j = t % NJ, i = (t / NJ) % MI, pw = t / (NJ * MI), wy = pw >> 2, wx = pw & 3;
    const int bm = blk / a.nb, bn = blk % a.nb;
    const int row0 = a.bm_sz * bm + 32 * (MI * wy + i), col0 = a.bn_sz * bn + 32 * (NJ * wx + j);
    if (row0 >= a.m || col0 >= a.n) return;
    const long long bsz = (long long)a.bm_sz * a.bn_sz;
    const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(a.ws + (long long)blk * bsz) + piece * 64 + lane;
    const long long stride = (long long)a.mb * a.nb * bsz / 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = w;
    for (; s + 7 * kWgFoldWaves < a.n_slabs; s += 8 * kWgFoldWaves) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(s + kWgFoldWaves * k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; s < a.n_slabs; s += kWgFoldWaves) acc += p[s * stride];
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < kWgFoldWaves; q += 4) v += (red[q][lane] + red[q + 1][lane]) + (red[q + 2][lane] + red[q + 3][lane]);
        const int x = col0 + (lane & 31), y0 = row0 + 8 * r4 + 4 * (lane >> 5);
        if (x < a.n) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (y0 + e < a.m) a.c[(long long)(y0 + e) * a.ldc + x + (a.gap && x >= a.gap_at ? a.gap : 0)] = v[e] * f1 * f2;
        }
    }
}

struct WgGeom { int shape, bm_sz, bn_sz, mb, nb, n_slabs; int64_t slab; };
// shape 0: 256 x 256 (always, unless E3DGE_WGRAD_SHAPE forces 1: 256 x 128 or 2: 128 x 128 for an A/B).  Measured at 98,304 points
// (tools/time_wgrad.py, kernel + fold): 256 x 256 outputs 72 / 84 / 82 us for shapes 0 / 1 / 2, 301 x 301 174 / 303 / 244, 512 x 301
// 185 / 266 / 278: the launch is bound by its reads, tiles outside (m, n) cost nothing (skipped per wave), and every smaller block
// re-reads an operand panel -- a cover with less padding does not pay.
static WgGeom wgrad_geometry(int m, int n, int64_t n_rows, int gap_at) {
    static const int bms[3] = {256, 256, 128}, bns[3] = {256, 128, 128}, resident[3] = {1, 1, 2};
    static const int forced = [] { const char* v = getenv("E3DGE_WGRAD_SHAPE"); return v && v[0] >= '0' && v[0] <= '2' && !v[1] ? v[0] - '0' : -1; }();
    const int sh = forced >= 0 && gap_at % bns[forced] == 0 ? forced : 0;
    WgGeom best{sh, bms[sh], bns[sh], (m + bms[sh] - 1) / bms[sh], (n + bns[sh] - 1) / bns[sh], 0, 0};
    const int blocks = best.mb * best.nb;
    int64_t s = (256 * resident[best.shape]) / blocks;       // one round of the resident workgroups
    const int64_t max_s = (n_rows + 255) / 256;               // a slab is at least 256 points
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int64_t len = (n_rows + s - 1) / s;
    len = (len + kWgPts - 1) / kWgPts * kWgPts;
    best.slab = len;
    best.n_slabs = (int)((n_rows + len - 1) / len);
    return best;
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int64_t e3dge_wgrad_ws_floats(int m, int n, int64_t n_rows) {
    if (m <= 0 || n <= 0 || n_rows <= 0) return 0;
    const WgGeom g = wgrad_geometry(m, n, n_rows, 0);
    return (int64_t)g.n_slabs * g.mb * (g.nb * g.bm_sz * g.bn_sz + 2 * g.bm_sz);         // partial blocks, then the column partials
}

template <int MI, int NJ, int MINB>
static int wgrad_launch(const WgradK& k, int grid, hipStream_t st) {
    constexpr int lds = wg_lds_bytes<MI, NJ>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<MI, NJ, MINB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(wgrad): %s", hipGetErrorString(e));
    wgrad_kernel<MI, NJ, MINB><<<dim3((unsigned)grid), dim3(kWgThreads), lds, st>>>(k);
    return check_launch("wgrad");
}

extern "C" int e3dge_wgrad(const E3dgeWgrad* g, e3dge_stream_t stream) {
    E3DGE_REQUIRE(g != nullptr, "wgrad: null args");
    E3DGE_REQUIRE(g->m >= 0 && g->n >= 0 && g->n_rows >= 0, "wgrad: bad sizes");
    E3DGE_REQUIRE(g->b_gap >= 0 && g->b_gap_at >= 0 && (g->b_gap == 0 || (g->b_gap_at > 0 && g->b_gap_at % 256 == 0)), "wgrad: b_gap_at must be a positive multiple of 256");
    if (g->m == 0) return E3DGE_OK;
    hipStream_t st = as_stream(stream);
    const int gap = g->b_gap_at < g->n ? g->b_gap : 0;
    if (g->n_rows == 0) {
        hipError_t e = hipSuccess;
        if (g->n > 0) {
            E3DGE_REQUIRE(g->c && g->ldc >= g->n + gap, "wgrad: null c / ldc < n");
            e = hipMemset2DAsync(g->c, (size_t)g->ldc * 4, 0, (size_t)(g->n + gap) * 4, (size_t)g->m, st);
        }
        if (e == hipSuccess && g->colsum) e = hipMemsetAsync(g->colsum, 0, (size_t)g->m * 4, st);
        if (e == hipSuccess && g->ccol) e = hipMemset2DAsync(g->ccol, (size_t)(g->ld_ccol > 0 ? g->ld_ccol : 1) * 4, 0, 4, (size_t)g->m, st);
        return e == hipSuccess ? E3DGE_OK : fail(E3DGE_ERR_LAUNCH, "wgrad: hipMemset: %s", hipGetErrorString(e));
    }
    if (g->n == 0) return E3DGE_OK;
    E3DGE_REQUIRE(g->c && g->ldc >= g->n + gap, "wgrad: null c / ldc < n (+ b_gap)");
    E3DGE_REQUIRE(g->a && g->b && g->amax_a && g->amax_b && g->ws, "wgrad: null pointer");
    E3DGE_REQUIRE(g->off_a >= 0 && g->off_b >= 0 && g->lda >= g->off_a + g->m && g->ldb >= g->off_b + g->n + gap, "wgrad: columns outside the rows");
    E3DGE_REQUIRE(g->ws_floats >= e3dge_wgrad_ws_floats(g->m, g->n, g->n_rows), "wgrad: workspace too small");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(g->a) | reinterpret_cast<uintptr_t>(g->b) | reinterpret_cast<uintptr_t>(g->c) | reinterpret_cast<uintptr_t>(g->ws)) & 3) == 0,
                  "wgrad: pointers must be 4-B aligned");
    E3DGE_REQUIRE((g->xcol == nullptr) == (g->ccol == nullptr), "wgrad: xcol and ccol go together");
    E3DGE_REQUIRE(!g->xcol || (g->ld_xcol >= 1 && g->ld_ccol >= 1), "wgrad: ld_xcol / ld_ccol");
    const WgGeom ge = wgrad_geometry(g->m, g->n, g->n_rows, gap ? g->b_gap_at : 0);
    WgradK k{};
    k.a = g->a; k.amax_a = g->amax_a; k.b = g->b; k.amax_b = g->amax_b; k.xcol = g->xcol; k.ws = g->ws; k.n_rows = g->n_rows;
    k.ws_col = g->ws + (int64_t)ge.n_slabs * ge.mb * ge.nb * ge.bm_sz * ge.bn_sz;
    k.lda = g->lda; k.off_a = g->off_a; k.m = g->m; k.ldb = g->ldb; k.off_b = g->off_b; k.n = g->n; k.relu_b = g->relu_b;
    k.mb = ge.mb; k.nb = ge.nb; k.slab = ge.slab; k.gap_at = g->b_gap_at; k.gap = gap; k.ld_xcol = g->ld_xcol;
    k.cols = (g->colsum ? 1 : 0) | (g->xcol ? 2 : 0);
    const int grid = ge.n_slabs * ge.mb * ge.nb;
    int rc = ge.shape == 0 ? wgrad_launch<4, 2, 1>(k, grid, st) : ge.shape == 1 ? wgrad_launch<4, 1, 1>(k, grid, st) : wgrad_launch<2, 1, 2>(k, grid, st);
    if (rc) return rc;
    WgFoldK f{};
    f.c = g->c; f.ws = g->ws; f.ws_col = k.ws_col; f.colsum = g->colsum; f.ccol = g->ccol; f.amax_a = g->amax_a; f.amax_b = g->amax_b;
    f.ldc = g->ldc; f.m = g->m; f.n = g->n; f.mb = ge.mb; f.nb = ge.nb; f.bm_sz = ge.bm_sz; f.bn_sz = ge.bn_sz; f.n_slabs = ge.n_slabs;
    f.gap_at = g->b_gap_at; f.gap = gap; f.ld_ccol = g->ld_ccol;
    f.n_mat = ge.mb * ge.nb * 8 * (ge.bm_sz / 64) * (ge.bn_sz / 128) * 4;
    const int n_col = k.cols ? (g->m + 63) / 64 : 0;
    wgrad_fold_kernel<<<dim3((unsigned)(f.n_mat + n_col)), dim3(64 * kWgFoldWaves), 0, st>>>(f);
    return check_launch("wgrad(fold)");
}
