// Parameter gradients of the local branch's fully connected layers (round 5):  C (m, n) = sum_p A[p, :m]^T  f(B[p, :n]),  f = identity | relu,
// with A = the layer's output gradient and B = its input, both (n_rows, ld) fp32 rows -- what autograd runs for every nn.Linear of
// ResnetBlockFC (helper_modules/resnetfc.py:49-58) and Fuse_sft_MLP (helper_modules/sft.py:84-110) in the stage-2 step
// (e3dge_full_runner.py:185-317): `grad_output.t() @ input`.  The contraction runs over the POINTS (98,304 per image at 64 x 64 x 24), the outputs
// are at most 512 x 512 -- a split-K problem: a workgroup owns a 128 x 128 output block and one slab of points, partial blocks go to a workspace
// and a second launch folds the slabs in fixed order (bit-reproducible, no atomics).
//
// Machine: 4 waves (2 x 2 of 64 x 64 outputs = four 32x32 MFMA tiles each), v_mfma_f32_32x32x16_f16 on split-f16 operands (hi + lo, three
// products, fp32 accumulate; one power-of-two scale per tensor from its amax buffer), 32 points per step.  Both MFMA operands need the
// contraction index (points) contiguous per lane while memory is point-major: a step's 32 x 128 slice of each operand is loaded with 16-byte
// row accesses, split, and written to LDS TRANSPOSED as [feature][point] f16 (two points per 32-bit word; row pitch 80 B: the 16-byte fragment
// reads of 16 consecutive features hit 16 different bank groups), one stage of 40 KB, the next step's slice waiting in registers.
#include "decoder_common.h"

namespace e3dge {

constexpr int kWgBlk = 128;                 // outputs per workgroup and dimension
constexpr int kWgPts = 32;                  // points per step (two k-steps of 16)
constexpr int kWgPitch = 80;                // bytes per feature row of one half (64 + 16 pad)
constexpr int kWgHalf = kWgBlk * kWgPitch;  // one half (hi or lo) of one operand
constexpr int kWgLdsBytes = 4 * kWgHalf;    // A hi, A lo, B hi, B lo = 40 KB

struct WgradK {
    const float* a; const float* amax_a; const float* b; const float* amax_b;
    float* ws;
    long long n_rows, slab;
    int lda, off_a, m, ldb, off_b, n, relu_b, mb, nb;
};

// this thread's share of a 32-point slice of one operand: points 2 pp, 2 pp + 1; feature quads fq and fq + 16 of the block
struct WgSlice { f32x4 v[2][2]; };          // [quad][point]
struct __attribute__((packed, aligned(4))) WgU4 { float v[4]; };             // 16-byte access at 4-byte alignment (rows of any pitch)
// where this thread reads: its first row of the current step and how many of each quad's four columns exist (4 = the whole quad)
struct WgSrc { const float* row; long long ld; int have[2]; };

__device__ __forceinline__ void wg_src_init(WgSrc& s, const float* base, int ld, int width, int f0, long long p_begin, int pp, int fq) {
    s.ld = ld;
    s.row = base + (p_begin + 2 * pp) * (long long)ld + f0 + 4 * fq;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int left = width - (f0 + 4 * (fq + 16 * q));
        s.have[q] = left < 0 ? 0 : (left > 4 ? 4 : left);
    }
}

// ROWS = false: all 32 points of the step exist (every step but the last of the last slab)
template <bool ROWS>
__device__ __forceinline__ void wg_load(WgSlice& s, const WgSrc& src, long long p0, long long p_end, int pp) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const float* __restrict__ p = src.row + e * src.ld + 64 * q;
            if (!ROWS || p0 + 2 * pp + e < p_end) {
                if (src.have[q] == 4) {
                    const WgU4 u = *reinterpret_cast<const WgU4*>(p);
                    v = f32x4{u.v[0], u.v[1], u.v[2], u.v[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) if (j < src.have[q]) v[j] = p[j];
                }
            }
            s.v[q][e] = v;
        }
    }
}

__device__ __forceinline__ void wg_store(const WgSlice& s, unsigned char* __restrict__ hi, unsigned char* __restrict__ lo, float sc, bool relu,
                                         int pp, int fq) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0 = s.v[q][0][j] * sc, x1 = s.v[q][1][j] * sc;
            if (relu) { x0 = fmaxf(x0, 0.0f); x1 = fmaxf(x1, 0.0f); }
            const HiLo w = split2(x0, x1);
            const int off = (4 * (fq + 16 * q) + j) * kWgPitch + 4 * pp;
            *reinterpret_cast<unsigned*>(hi + off) = w.h;
            *reinterpret_cast<unsigned*>(lo + off) = w.l;
        }
    }
}

__global__ void __launch_bounds__(256, 2) wgrad_kernel(const WgradK a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    unsigned char* const ah = wg_lds, * const al = wg_lds + kWgHalf, * const bh = wg_lds + 2 * kWgHalf, * const bl = wg_lds + 3 * kWgHalf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int pp = tid & 15, fq = tid >> 4;                   // (a wave: 16 point pairs x 4 quads -> 64 different LDS banks per write)
    const int blk = blockIdx.x % (a.mb * a.nb), slab = blockIdx.x / (a.mb * a.nb);
    const int bm = blk / a.nb, bn = blk % a.nb;
    const long long p_begin = (long long)slab * a.slab, p_end = min(a.n_rows, p_begin + a.slab);
    const unsigned ea = scale_exponent(amax_read(a.amax_a, lane)), eb = scale_exponent(amax_read(a.amax_b, lane));
    const float sa = __uint_as_float((268u - ea) << 23), sb = __uint_as_float((268u - eb) << 23);       // 2^(141 - e)
    const float* __restrict__ pa = a.a + a.off_a;
    const float* __restrict__ pb = a.b + a.off_b;
    const int wy = wave >> 1, wx = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    WgSlice sa_r, sb_r;
    WgSrc src_a, src_b;
    wg_src_init(src_a, pa, a.lda, a.m, kWgBlk * bm, p_begin, pp, fq);
    wg_src_init(src_b, pb, a.ldb, a.n, kWgBlk * bn, p_begin, pp, fq);
    if (p_begin + kWgPts <= p_end) { wg_load<false>(sa_r, src_a, p_begin, p_end, pp); wg_load<false>(sb_r, src_b, p_begin, p_end, pp); }
    else { wg_load<true>(sa_r, src_a, p_begin, p_end, pp); wg_load<true>(sb_r, src_b, p_begin, p_end, pp); }
    for (long long p0 = p_begin; p0 < p_end; p0 += kWgPts) {
        __syncthreads();                                      // the previous step's fragment reads are done
        wg_store(sa_r, ah, al, sa, false, pp, fq);
        wg_store(sb_r, bh, bl, sb, a.relu_b != 0, pp, fq);
        __syncthreads();
        if (p0 + kWgPts < p_end) {                            // the next slice travels while this one is multiplied
            src_a.row += kWgPts * src_a.ld;
            src_b.row += kWgPts * src_b.ld;
            if (p0 + 2 * kWgPts <= p_end) { wg_load<false>(sa_r, src_a, p0 + kWgPts, p_end, pp); wg_load<false>(sb_r, src_b, p0 + kWgPts, p_end, pp); }
            else { wg_load<true>(sa_r, src_a, p0 + kWgPts, p_end, pp); wg_load<true>(sb_r, src_b, p0 + kWgPts, p_end, pp); }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ro = (32 * (2 * wy + i) + col) * kWgPitch + 32 * s + 16 * half;
                fah[i] = *reinterpret_cast<const u32x4*>(ah + ro);
                fal[i] = *reinterpret_cast<const u32x4*>(al + ro);
                const int co = (32 * (2 * wx + i) + col) * kWgPitch + 32 * s + 16 * half;
                fbh[i] = *reinterpret_cast<const u32x4*>(bh + co);
                fbl[i] = *reinterpret_cast<const u32x4*>(bl + co);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = mfma16(fah[i], fbh[j], acc[i][j]);
                    acc[i][j] = mfma16(fal[i], fbh[j], acc[i][j]);
                    acc[i][j] = mfma16(fah[i], fbl[j], acc[i][j]);
                }
        }
    }
    // partial block -> workspace [slab][block][128][128]; register r of tile (i, j): row 32 (2 wy + i) + row_of(r, half), column 32 (2 wx + j) + col
    float* __restrict__ out = a.ws + ((long long)slab * (a.mb * a.nb) + blk) * (kWgBlk * kWgBlk);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(32 * (2 * wy + i) + row_of(r, half)) * kWgBlk + 32 * (2 * wx + j) + col] = acc[i][j][r];
}

// C[m][n] = 2^(ea + eb - 282) sum over slabs (in order) of the partial blocks
__global__ void __launch_bounds__(256) wgrad_fold_kernel(float* __restrict__ c, int ldc, const float* __restrict__ ws, const float* __restrict__ amax_a,
                                                         const float* __restrict__ amax_b, int m, int n, int mb, int nb, int n_slabs) {
    const int lane = threadIdx.x & 63;
    const unsigned ea = scale_exponent(amax_read(amax_a, lane)), eb = scale_exponent(amax_read(amax_b, lane));
    const int ee = (int)ea + (int)eb - 282;                                           // 1 / (sa sb) = 2^ee, |ee| can exceed the fp32 exponent range:
    const float f1 = __uint_as_float((unsigned)(127 + ee / 2) << 23), f2 = __uint_as_float((unsigned)(127 + (ee - ee / 2)) << 23);   // two factors
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= n || y >= m) return;
    const int blk = (y / kWgBlk) * nb + x / kWgBlk;
    const float* __restrict__ p = ws + (long long)blk * (kWgBlk * kWgBlk) + (y % kWgBlk) * kWgBlk + x % kWgBlk;
    const long long stride = (long long)mb * nb * (kWgBlk * kWgBlk);
    float acc = 0.0f;
    for (int s = 0; s < n_slabs; ++s) acc += p[s * stride];
    c[(long long)y * ldc + x] = acc * f1 * f2;
}

static void wgrad_geometry(int m, int n, int64_t n_rows, int* mb, int* nb, int* n_slabs, int64_t* slab) {
    *mb = (m + kWgBlk - 1) / kWgBlk;
    *nb = (n + kWgBlk - 1) / kWgBlk;
    const int blocks = *mb * *nb;
    int64_t s = 512 / blocks;                                 // one round of the 2 x 256 resident workgroups (a 513th would cost a second one)
    const int64_t max_s = (n_rows + 255) / 256;               // a slab is at least 256 points
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int64_t len = (n_rows + s - 1) / s;
    len = (len + kWgPts - 1) / kWgPts * kWgPts;
    *slab = len;
    *n_slabs = (int)((n_rows + len - 1) / len);
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int64_t e3dge_wgrad_ws_floats(int m, int n, int64_t n_rows) {
    if (m <= 0 || n <= 0 || n_rows <= 0) return 0;
    int mb, nb, ns;
    int64_t slab;
    wgrad_geometry(m, n, n_rows, &mb, &nb, &ns, &slab);
    return (int64_t)ns * mb * nb * kWgBlk * kWgBlk;
}

extern "C" int e3dge_wgrad(const E3dgeWgrad* g, e3dge_stream_t stream) {
    E3DGE_REQUIRE(g != nullptr, "wgrad: null args");
    E3DGE_REQUIRE(g->m >= 0 && g->n >= 0 && g->n_rows >= 0, "wgrad: bad sizes");
    if (g->m == 0 || g->n == 0) return E3DGE_OK;
    E3DGE_REQUIRE(g->c && g->ldc >= g->n, "wgrad: null c / ldc < n");
    hipStream_t st = as_stream(stream);
    if (g->n_rows == 0) {
        hipError_t e = hipMemset2DAsync(g->c, (size_t)g->ldc * 4, 0, (size_t)g->n * 4, (size_t)g->m, st);
        return e == hipSuccess ? E3DGE_OK : fail(E3DGE_ERR_LAUNCH, "wgrad: hipMemset2DAsync: %s", hipGetErrorString(e));
    }
    E3DGE_REQUIRE(g->a && g->b && g->amax_a && g->amax_b && g->ws, "wgrad: null pointer");
    E3DGE_REQUIRE(g->off_a >= 0 && g->off_b >= 0 && g->lda >= g->off_a + g->m && g->ldb >= g->off_b + g->n, "wgrad: columns outside the rows");
    E3DGE_REQUIRE(g->ws_floats >= e3dge_wgrad_ws_floats(g->m, g->n, g->n_rows), "wgrad: workspace too small");
    E3DGE_REQUIRE(((reinterpret_cast<uintptr_t>(g->a) | reinterpret_cast<uintptr_t>(g->b) | reinterpret_cast<uintptr_t>(g->c) | reinterpret_cast<uintptr_t>(g->ws)) & 3) == 0,
                  "wgrad: pointers must be 4-B aligned");
    WgradK k{};
    k.a = g->a; k.amax_a = g->amax_a; k.b = g->b; k.amax_b = g->amax_b; k.ws = g->ws; k.n_rows = g->n_rows;
    k.lda = g->lda; k.off_a = g->off_a; k.m = g->m; k.ldb = g->ldb; k.off_b = g->off_b; k.n = g->n; k.relu_b = g->relu_b;
    int ns;
    int64_t slab;
    wgrad_geometry(g->m, g->n, g->n_rows, &k.mb, &k.nb, &ns, &slab);
    k.slab = slab;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kWgLdsBytes);
    if (e != hipSuccess) return fail(E3DGE_ERR_LAUNCH, "hipFuncSetAttribute(wgrad): %s", hipGetErrorString(e));
    wgrad_kernel<<<dim3((unsigned)(ns * k.mb * k.nb)), dim3(256), kWgLdsBytes, st>>>(k);
    int rc = check_launch("wgrad");
    if (rc) return rc;
    wgrad_fold_kernel<<<dim3((unsigned)((g->n + 63) / 64), (unsigned)((g->m + 3) / 4)), dim3(256), 0, st>>>(g->c, g->ldc, g->ws, g->amax_a, g->amax_b,
                                                                                                          g->m, g->n, k.mb, k.nb, ns);
    return check_launch("wgrad(fold)");
}
