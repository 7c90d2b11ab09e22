// Per-point query of the local branch's feature maps: perspective projection of world-space points into a
// view, bilinear gather of a feature map at the projected position, in-image mask, and the positional encoding of the
// points -- the inputs of the second renderer pass's texture head (project/trainers/E3DGE/e3dge_full_runner.py:185-317).
//
// Reference being replaced (per point, PyTorch ops):
//   HGPIFuNetGAN.query (vendor/pifu/lib/model/HGPIFuGANNet.py:85-151): xyz = perspective(points, calibs); y *= -1;
//       in_img = |x| <= 1 & |y| <= 1; feats = index(im_feat, xy)
//   perspective (vendor/pifu/lib/geometry.py:101-129): homo = trans + rot p;  z = -homo_z if the FIRST point of the FIRST
//       sample has homo_z < 0 else homo_z;  xy = homo_xy / z
//   index (geometry.py:64-80): grid_sample(feat, uv, bilinear, zeros padding, align_corners=False)
//   PosEncoding.forward (project/utils/misc_utils.py:148-185): [x, sin(2^k x), cos(2^k x)]_k, k = 0..n_freqs-1
//
// Layout: the feature map is CHANNEL-LAST (B, h, w, C) in HBM, so the four corners of a point are four contiguous C-float
// rows: one wave per point, 16 B per lane per corner, every byte of a fetched line is used.  (NCHW would make each of the
// 4*C corner reads of a point its own cache line.)  Outputs go straight into a slice [col_off, col_off + C) of a wider
// per-point row (leading dimension ld), so the concatenations of the reference (:255-262, :279) cost nothing.
// Bound: HBM / L2 gather, (4 C reads + C writes) * 4 B per point.
#include "common.h"

namespace e3dge {

typedef float lq_f4 __attribute__((ext_vector_type(4)));

struct LocalQueryK {
    const float* pts;      // (B, N, 3) world-space points
    const float* calibs;   // (B, 3, 4)
    const float* fmap;     // (B, h, w, C) channel-last, or null (projection / mask only)
    float* out;            // (B, N, ld): features written to [col_off, col_off + C)
    float* in_img;         // (B, N) or null: 1.0 inside the image plane, else 0.0 (may alias a column of `out` via mask_ld)
    float* proj;           // (B, N, 3) or null: projected (x, y flipped, depth)
    long long N;
    int B, C, h, w, ld, col_off, mask_ld, mask_off;
};

__global__ void __launch_bounds__(256) local_query_kernel(const LocalQueryK a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // sign convention of the whole batch: decided by the first point of the first sample (geometry.py:115-118)
    float zsign;
    {
        const float* c = a.calibs;
        const float* p = a.pts;
        const float hz = fmaf(c[8], p[0], fmaf(c[9], p[1], fmaf(c[10], p[2], c[11])));
        zsign = hz < 0.0f ? -1.0f : 1.0f;
    }
    const long long total = (long long)a.B * a.N;
    // (round 6: the next point's coordinates are requested while this one is gathered, and the four corner rows are loaded unconditionally
    // from clamped pixels with the weights of corners outside the map set to zero: as `if (corner valid) acc += w * row` every load sat in
    // a block of its own behind a vmcnt(0) -- two to five exposed round trips per point, 45 us for 100 MB of output)
    const long long stride_pt = (long long)gridDim.x * 4;
    long long pt = (long long)blockIdx.x * 4 + wave;
    float nx_ = 0.f, ny_ = 0.f, nz_ = 0.f;
    if (pt < total) { const float* p = a.pts + (size_t)pt * 3; nx_ = p[0]; ny_ = p[1]; nz_ = p[2]; }
    for (; pt < total; pt += stride_pt) {
        const int b = (int)(pt / a.N);
        const float* c = a.calibs + (size_t)b * 12;
        const float px = nx_, py = ny_, pz = nz_;
        if (pt + stride_pt < total) { const float* p = a.pts + (size_t)(pt + stride_pt) * 3; nx_ = p[0]; ny_ = p[1]; nz_ = p[2]; }
        const float hx = c[3] + (c[0] * px + c[1] * py + c[2] * pz);
        const float hy = c[7] + (c[4] * px + c[5] * py + c[6] * pz);
        const float hz = c[11] + (c[8] * px + c[9] * py + c[10] * pz);
        const float z = zsign * hz;
        const float x = hx / z, y = -(hy / z);                        // y flipped to grid_sample's convention (:109)
        const bool inside = x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f;
        if (lane == 0) {
            if (a.in_img) a.in_img[(size_t)pt * a.mask_ld + a.mask_off] = inside ? 1.0f : 0.0f;
            if (a.proj) { float* q = a.proj + (size_t)pt * 3; q[0] = x; q[1] = y; q[2] = z; }
        }
        if (!a.fmap) continue;
        // grid_sample, align_corners=False: pixel coordinate = ((g + 1) * size - 1) / 2
        const float fx = ((x + 1.0f) * (float)a.w - 1.0f) * 0.5f, fy = ((y + 1.0f) * (float)a.h - 1.0f) * 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float tx = fx - x0f, ty = fy - y0f;
        // NaN / huge coordinates: every corner out of range -> zeros (as grid_sample's zeros padding gives)
        const bool finite = fx > -2.0f && fx < (float)a.w + 1.0f && fy > -2.0f && fy < (float)a.h + 1.0f;
        const int x0 = finite ? (int)x0f : -5, y0 = finite ? (int)y0f : -5;
        const bool vx0 = x0 >= 0 && x0 < a.w, vx1 = x0 + 1 >= 0 && x0 + 1 < a.w;
        const bool vy0 = y0 >= 0 && y0 < a.h, vy1 = y0 + 1 >= 0 && y0 + 1 < a.h;
        const float w00 = (vy0 && vx0) ? (1.0f - tx) * (1.0f - ty) : 0.0f, w01 = (vy0 && vx1) ? tx * (1.0f - ty) : 0.0f;
        const float w10 = (vy1 && vx0) ? (1.0f - tx) * ty : 0.0f, w11 = (vy1 && vx1) ? tx * ty : 0.0f;
        const int xa = min(max(x0, 0), a.w - 1), xb = min(max(x0 + 1, 0), a.w - 1), ya = min(max(y0, 0), a.h - 1), yb = min(max(y0 + 1, 0), a.h - 1);
        const float* base = a.fmap + (size_t)b * a.h * a.w * a.C;
        float* o = a.out + (size_t)pt * a.ld + a.col_off;
        for (int ch = lane * 4; ch < a.C; ch += 256) {
            const lq_f4 f00 = *reinterpret_cast<const lq_f4*>(base + ((size_t)ya * a.w + xa) * a.C + ch);
            const lq_f4 f01 = *reinterpret_cast<const lq_f4*>(base + ((size_t)ya * a.w + xb) * a.C + ch);
            const lq_f4 f10 = *reinterpret_cast<const lq_f4*>(base + ((size_t)yb * a.w + xa) * a.C + ch);
            const lq_f4 f11 = *reinterpret_cast<const lq_f4*>(base + ((size_t)yb * a.w + xb) * a.C + ch);
            lq_f4 acc = {0.f, 0.f, 0.f, 0.f};
            // same accumulation order as the native kernel: nw, ne, sw, se (a corner outside the map: weight 0, a finite row from a clamped pixel)
            acc += w00 * f00;
            acc += w01 * f01;
            acc += w10 * f10;
            acc += w11 * f11;
            if ((a.ld & 3) == 0 && (a.col_off & 3) == 0) *reinterpret_cast<lq_f4*>(o + ch) = acc;
            else { o[ch] = acc[0]; o[ch + 1] = acc[1]; o[ch + 2] = acc[2]; o[ch + 3] = acc[3]; }
        }
    }
}

// Backward of the gather (reference: project/models/op/grid_sample_gradfix.py:52-89 -- grid_sampler_2d_backward for the feature map
// AND the grid -- behind HGPIFuNetGAN.query's index(), vendor/pifu/lib/model/HGPIFuGANNet.py:85-151; stage-2 training
// back-propagates into the hourglass filters' maps through it, e3dge_full_runner.py:185-317).  One wave per point, as forward:
//   d_fmap[corner] += w_corner * d_out          (atomic adds into the channel-last map: four contiguous C-float rows per point)
//   d_fx = sum_c d_out_c ((1-ty)(f01 - f00) + ty (f11 - f10)),  d_fy = sum_c d_out_c ((1-tx)(f10 - f00) + tx (f11 - f01))
//   (corners outside the map count as zeros, like grid_sample's zeros padding), then through the pixel mapping, the flip of y,
//   the perspective division and the calibration rows to d_pts.
struct LocalQueryBwdK {
    const float* pts; const float* calibs; const float* fmap;      // as forward
    const float* d_out;     // (B, N, ld): gradient of the features, columns [col_off, col_off + C)
    float* d_fmap;          // (B, h, w, C) channel-last, zero-filled by the caller, or null
    float* d_pts;           // (B, N, 3) or null
    const int* order;       // null, or a permutation of the B N points: the kernel walks order[0], order[1], ... (round 6: sorted by corner pixel)
    long long N;
    int B, C, h, w, ld, col_off;
};

// Round 6: (i) a wave walks kLqRun CONSECUTIVE points and keeps the four corner sums in registers while the corner pixel stays the
// same -- the samples of a ray project to one pixel of their own view's map (24 atomics' worth of contention per address in the first
// version) and to neighbouring pixels of the other view's --, (ii) an atomic instruction covers 64 consecutive floats (two cache lines)
// instead of 64 floats 16 bytes apart (eight lines): the stage-2 step spent 2.4 of its 10.9 ms in the two launches of the first version.
constexpr int kLqRun = 32, kLqBatch = 4;
static_assert(kLqRun <= 64, "a run's entries of `order` live in one register");

__global__ void __launch_bounds__(256) local_query_bwd_kernel(const LocalQueryBwdK a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float zsign;
    {
        const float* c = a.calibs;
        const float* p = a.pts;
        const float hz = fmaf(c[8], p[0], fmaf(c[9], p[1], fmaf(c[10], p[2], c[11])));
        zsign = hz < 0.0f ? -1.0f : 1.0f;
    }
    const long long total = (long long)a.B * a.N;
    const long long n_runs = (total + kLqRun - 1) / kLqRun;
    const int n_c4 = a.C / 4;                                   // float4 groups per row; this lane's channels: 64 j + lane, j < C / 64 (+ tail)
    const int n_j = (a.C + 63) / 64;
    for (long long run = (long long)blockIdx.x * 4 + wave; run < n_runs; run += (long long)gridDim.x * 4) {
        const long long p0 = run * kLqRun, p1 = p0 + kLqRun < total ? p0 + kLqRun : total;
        // the corner sums of the current pixel: [corner][j]
        float acc[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] = 0.0f;
        bool have = false;                                      // a pixel is being accumulated: (cb, cy, cx) = image, top-left corner (-1 .. h-1, -1 .. w-1)
        int cb = 0, cy = 0, cx = 0;
        bool cvx0 = false, cvx1 = false, cvy0 = false, cvy1 = false;
        auto flush = [&]() {
            if (!have || !a.d_fmap) return;
            float* base = a.d_fmap + (((long long)cb * a.h + cy) * a.w + cx) * (long long)a.C;     // (only valid corners are dereferenced)
            const size_t row = (size_t)a.w * a.C;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = 64 * j + lane;
                if (j < n_j && ch < a.C) {
                    if (cvy0 && cvx0) atomicAdd(base + ch, acc[0][j]);
                    if (cvy0 && cvx1) atomicAdd(base + a.C + ch, acc[1][j]);
                    if (cvy1 && cvx0) atomicAdd(base + row + ch, acc[2][j]);
                    if (cvy1 && cvx1) atomicAdd(base + row + a.C + ch, acc[3][j]);
                }
                acc[0][j] = acc[1][j] = acc[2][j] = acc[3][j] = 0.0f;
            }
        };
        // The points come in batches of kLqBatch whose coordinates and gradient rows are requested one batch AHEAD: vmcnt retires in order,
        // so a row load issued behind a pixel's atomics waited for them (and for its own HBM round trip) inside a 32-deep serial chain --
        // 195 us per launch for 100 MB of rows.  Now a batch's loads are in the queue before the previous batch's atomics.
        struct Pt { long long q; float x, y, z; float g[4]; };
        Pt nxt[kLqBatch], cur[kLqBatch];
        long long ixn[kLqBatch];
        // (the run's kLqRun <= 64 entries of `order` in ONE load, lane k holding entry k; a batch's indices are lane reads of that register)
        const int my_entry = a.order ? a.order[p0 + lane < p1 ? p0 + lane : p1 - 1] : 0;
        auto load_idx = [&](long long e0) {
#pragma unroll
            for (int k = 0; k < kLqBatch; ++k) {
                const long long e = e0 + k < p1 ? e0 + k : p1 - 1;         // (clamped: entries past the run are loaded twice, never used)
                ixn[k] = a.order ? (long long)__shfl(my_entry, (int)(e - p0), kWave) : e;
            }
        };
        auto load_batch = [&](Pt (&d)[kLqBatch]) {
#pragma unroll
            for (int k = 0; k < kLqBatch; ++k) {
                const long long q = ixn[k];
                d[k].q = q;
                const float* p = a.pts + (size_t)q * 3;
                d[k].x = p[0]; d[k].y = p[1]; d[k].z = p[2];
                const float* g = a.d_out + (size_t)q * a.ld + a.col_off;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = 64 * j + lane;
                    d[k].g[j] = (j < n_j && ch < a.C) ? g[ch] : 0.0f;
                }
            }
        };
        load_idx(p0);
        load_batch(nxt);
        if (p0 + kLqBatch < p1) load_idx(p0 + kLqBatch);
        for (long long pb = p0; pb < p1; pb += kLqBatch) {
#pragma unroll
            for (int k = 0; k < kLqBatch; ++k) cur[k] = nxt[k];
            if (pb + kLqBatch < p1) {
                load_batch(nxt);
                if (pb + 2 * kLqBatch < p1) load_idx(pb + 2 * kLqBatch);
            }
#pragma unroll
          for (int k = 0; k < kLqBatch; ++k) {
            if (pb + k >= p1) break;
            const long long pt = cur[k].q;
            const int b = (int)(pt / a.N);
            const float* c = a.calibs + (size_t)b * 12;
            const float px = cur[k].x, py = cur[k].y, pz = cur[k].z;
            const float hx = c[3] + (c[0] * px + c[1] * py + c[2] * pz);
            const float hy = c[7] + (c[4] * px + c[5] * py + c[6] * pz);
            const float hz = c[11] + (c[8] * px + c[9] * py + c[10] * pz);
            const float z = zsign * hz;
            const float x = hx / z, y = -(hy / z);
            const float fx = ((x + 1.0f) * (float)a.w - 1.0f) * 0.5f, fy = ((y + 1.0f) * (float)a.h - 1.0f) * 0.5f;
            const float x0f = floorf(fx), y0f = floorf(fy);
            const float tx = fx - x0f, ty = fy - y0f;
            const bool finite = fx > -2.0f && fx < (float)a.w + 1.0f && fy > -2.0f && fy < (float)a.h + 1.0f;
            const int x0 = finite ? (int)x0f : -5, y0 = finite ? (int)y0f : -5;
            const float w00 = (1.0f - tx) * (1.0f - ty), w01 = tx * (1.0f - ty), w10 = (1.0f - tx) * ty, w11 = tx * ty;
            const bool vx0 = x0 >= 0 && x0 < a.w, vx1 = x0 + 1 >= 0 && x0 + 1 < a.w;
            const bool vy0 = y0 >= 0 && y0 < a.h, vy1 = y0 + 1 >= 0 && y0 + 1 < a.h;
            const bool any = (vx0 || vx1) && (vy0 || vy1);
            const float* g = a.d_out + (size_t)pt * a.ld + a.col_off;
            if (a.d_fmap && any) {
                if (!have || b != cb || y0 != cy || x0 != cx) {
                    flush();
                    have = true; cb = b; cy = y0; cx = x0; cvx0 = vx0; cvx1 = vx1; cvy0 = vy0; cvy1 = vy1;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = cur[k].g[j];                         // (0 beyond C)
                    acc[0][j] = fmaf(w00, d, acc[0][j]); acc[1][j] = fmaf(w01, d, acc[1][j]);
                    acc[2][j] = fmaf(w10, d, acc[2][j]); acc[3][j] = fmaf(w11, d, acc[3][j]);
                }
            }
            if (a.d_pts) {
                const size_t moff = (size_t)b * a.h * a.w * a.C;
                const size_t o00 = moff + ((size_t)y0 * a.w + x0) * a.C, o01 = o00 + a.C, o10 = o00 + (size_t)a.w * a.C, o11 = o10 + a.C;
                float gx = 0.0f, gy = 0.0f;
                for (int c4 = lane; c4 < n_c4; c4 += 64) {
                    const int ch = 4 * c4;
                    lq_f4 d;
                    if ((a.ld & 3) == 0 && (a.col_off & 3) == 0) d = *reinterpret_cast<const lq_f4*>(g + ch);
                    else { d[0] = g[ch]; d[1] = g[ch + 1]; d[2] = g[ch + 2]; d[3] = g[ch + 3]; }
                    const lq_f4 zero = {0.f, 0.f, 0.f, 0.f};
                    const lq_f4 f00 = (vy0 && vx0) ? *reinterpret_cast<const lq_f4*>(a.fmap + o00 + ch) : zero;
                    const lq_f4 f01 = (vy0 && vx1) ? *reinterpret_cast<const lq_f4*>(a.fmap + o01 + ch) : zero;
                    const lq_f4 f10 = (vy1 && vx0) ? *reinterpret_cast<const lq_f4*>(a.fmap + o10 + ch) : zero;
                    const lq_f4 f11 = (vy1 && vx1) ? *reinterpret_cast<const lq_f4*>(a.fmap + o11 + ch) : zero;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gx = fmaf(d[j], (1.0f - ty) * (f01[j] - f00[j]) + ty * (f11[j] - f10[j]), gx);
                        gy = fmaf(d[j], (1.0f - tx) * (f10[j] - f00[j]) + tx * (f11[j] - f01[j]), gy);
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { gx += __shfl_xor(gx, off, kWave); gy += __shfl_xor(gy, off, kWave); }
                if (lane == 0) {
                    const float dx = gx * (0.5f * (float)a.w), dy = gy * (0.5f * (float)a.h);     // d fx / d x = w / 2
                    // x = hx / z, y = -hy / z, z = zsign hz
                    const float dhx = dx / z, dhy = -dy / z, dhz = zsign * (-(dx * x + dy * y) / z);
                    float* q = a.d_pts + (size_t)pt * 3;
                    q[0] = c[0] * dhx + c[4] * dhy + c[8] * dhz;
                    q[1] = c[1] * dhx + c[5] * dhy + c[9] * dhz;
                    q[2] = c[2] * dhx + c[6] * dhy + c[10] * dhz;
                }
            }
          }
        }
        flush();
    }
}

// ---- round 6: the points sorted by the pixel of their top-left corner ---------------------------------------------------------------------
// Along a ray of ANOTHER view consecutive samples land on different pixels (measured: the pixel changes at every one of the 98,304 points
// of the stage-2 step's reference-view gather, against 4 % for a view's own rays), so the run accumulation above never merges anything and
// the launch is 1.57 M device-scope atomic instructions: 290 us, six times the query view's.  A counting sort by (image, corner pixel)
// makes every pixel's ~6 points neighbours in `order`: a wave's run then flushes once per pixel.  Keys: (y0 + 1) (w + 1) + (x0 + 1) per
// image, one more bin for points that touch no pixel (they sort to the end and are skipped).
struct LqSortK { const float* pts; const float* calibs; int* keys; int* counts; int* starts; int* cursor; int* order; long long N; int B, h, w, nb; };   // cursor: per-point ranks

__device__ __forceinline__ int lq_corner_key(const float* c, const float* p, float zsign, int b, int h, int w, int nb) {
    // (the same arithmetic, in the same order, as local_query_bwd_kernel)
    const float px = p[0], py = p[1], pz = p[2];
    const float hx = c[3] + (c[0] * px + c[1] * py + c[2] * pz);
    const float hy = c[7] + (c[4] * px + c[5] * py + c[6] * pz);
    const float hz = c[11] + (c[8] * px + c[9] * py + c[10] * pz);
    const float z = zsign * hz;
    const float x = hx / z, y = -(hy / z);
    const float fx = ((x + 1.0f) * (float)w - 1.0f) * 0.5f, fy = ((y + 1.0f) * (float)h - 1.0f) * 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const bool finite = fx > -2.0f && fx < (float)w + 1.0f && fy > -2.0f && fy < (float)h + 1.0f;
    const int x0 = finite ? (int)x0f : -5, y0 = finite ? (int)y0f : -5;
    const bool vx = (x0 >= 0 && x0 < w) || (x0 + 1 >= 0 && x0 + 1 < w), vy = (y0 >= 0 && y0 < h) || (y0 + 1 >= 0 && y0 + 1 < h);
    return (vx && vy) ? b * ((h + 1) * (w + 1)) + (y0 + 1) * (w + 1) + (x0 + 1) : nb - 1;
}

__global__ void __launch_bounds__(256) lq_zero_kernel(int* __restrict__ p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0;
}

__global__ void __launch_bounds__(256) lq_key_kernel(const LqSortK a) {
    float zsign;
    {
        const float* c = a.calibs;
        const float* p = a.pts;
        const float hz = fmaf(c[8], p[0], fmaf(c[9], p[1], fmaf(c[10], p[2], c[11])));
        zsign = hz < 0.0f ? -1.0f : 1.0f;
    }
    const long long total = (long long)a.B * a.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / a.N);
        const int key = lq_corner_key(a.calibs + (size_t)b * 12, a.pts + (size_t)i * 3, zsign, b, a.h, a.w, a.nb);
        a.keys[i] = key;
        a.cursor[i] = atomicAdd(a.counts + key, 1);            // this point's rank inside its bin (the scatter then needs no atomics)
    }
}

// exclusive scan of the bin counts: one workgroup, a thread per chunk of consecutive bins
__global__ void __launch_bounds__(1024) lq_scan_kernel(const LqSortK a) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, chunk = (a.nb + 1023) / 1024, b0 = tid * chunk, b1 = min(a.nb, b0 + chunk);
    int sum = 0;
    for (int i = b0; i < b1; ++i) sum += a.counts[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int i = b0; i < b1; ++i) { a.starts[i] = run; run += a.counts[i]; }
}

__global__ void __launch_bounds__(256) lq_scatter_kernel(const LqSortK a) {
    const long long total = (long long)a.B * a.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        a.order[a.starts[a.keys[i]] + a.cursor[i]] = (int)i;
    }
}

// out[m, col_off + ...] = [x(3), sin(f_0 x)(3), cos(f_0 x)(3), sin(f_1 x)(3), ...], f_k = 2^k
__global__ void __launch_bounds__(256)
pos_encoding_kernel(float* __restrict__ out, const float* __restrict__ pts, long long M, int n_freqs, int ld, int col_off) {
    const int width = 3 * (2 * n_freqs + 1);
    const long long total = M * width;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long m = e / width;
        const int j = (int)(e - m * width);
        const int grp = j / 3, ax = j - grp * 3;
        const float x = pts[m * 3 + ax];
        float v = x;
        if (grp > 0) {
            const float f = (float)(1 << ((grp - 1) >> 1));
            v = ((grp - 1) & 1) ? cosf(f * x) : sinf(f * x);
        }
        out[m * ld + col_off + j] = v;
    }
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int e3dge_local_query(float* out, int ld, int col_off, float* in_img, int mask_ld, int mask_off, float* proj,
                                 const float* pts, const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts,
                                 int channels, int fh, int fw, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && n_pts >= 0, "local_query: bad sizes");
    if (batch == 0 || n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(pts && calibs, "local_query: null pointer");
    if (fmap_nhwc) {
        E3DGE_REQUIRE(out != nullptr && channels >= 4 && channels % 4 == 0 && fh >= 1 && fw >= 1, "local_query: feature map needs C %% 4 == 0");
        E3DGE_REQUIRE(ld >= col_off + channels && col_off >= 0, "local_query: output slice [%d, %d) outside ld=%d", col_off, col_off + channels, ld);
        E3DGE_REQUIRE((reinterpret_cast<uintptr_t>(fmap_nhwc) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "local_query: fmap / out must be 16-B aligned");
    }
    E3DGE_REQUIRE(in_img == nullptr || (mask_ld >= 1 && mask_off >= 0 && mask_off < mask_ld), "local_query: bad mask stride");
    LocalQueryK k{};
    k.pts = pts; k.calibs = calibs; k.fmap = fmap_nhwc; k.out = out; k.in_img = in_img; k.proj = proj; k.N = n_pts; k.B = batch;
    k.C = channels; k.h = fh; k.w = fw; k.ld = ld; k.col_off = col_off; k.mask_ld = mask_ld; k.mask_off = mask_off;
    int64_t blocks = ((int64_t)batch * n_pts + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    local_query_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(k);
    return check_launch("local_query");
}

static int local_query_bwd_launch(float* d_fmap_nhwc, float* d_pts, const float* d_out, int ld, int col_off, const float* pts,
                                  const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts, int channels, int fh, int fw,
                                  int* sort_ws, e3dge_stream_t stream);

extern "C" int e3dge_local_query_bwd(float* d_fmap_nhwc, float* d_pts, const float* d_out, int ld, int col_off, const float* pts,
                                     const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts, int channels, int fh, int fw,
                                     e3dge_stream_t stream) {
    return local_query_bwd_launch(d_fmap_nhwc, d_pts, d_out, ld, col_off, pts, calibs, fmap_nhwc, batch, n_pts, channels, fh, fw, nullptr, stream);
}

extern "C" int64_t e3dge_local_query_sort_ws_ints(int batch, int64_t n_pts, int fh, int fw) {
    if (batch <= 0 || n_pts <= 0 || fh <= 0 || fw <= 0) return 0;
    const int64_t nb = (int64_t)batch * (fh + 1) * (fw + 1) + 1;
    return 3 * (int64_t)batch * n_pts + 2 * nb;                   // keys, order, ranks | counts, starts
}

extern "C" int e3dge_local_query_bwd_sorted(float* d_fmap_nhwc, float* d_pts, const float* d_out, int ld, int col_off, const float* pts,
                                            const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts, int channels, int fh, int fw,
                                            int* ws, int64_t ws_ints, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && n_pts >= 0 && fh >= 1 && fw >= 1, "local_query_bwd_sorted: bad sizes");
    if (batch == 0 || n_pts == 0 || (!d_fmap_nhwc && !d_pts)) return E3DGE_OK;
    E3DGE_REQUIRE(ws && ws_ints >= e3dge_local_query_sort_ws_ints(batch, n_pts, fh, fw), "local_query_bwd_sorted: workspace too small");
    E3DGE_REQUIRE((int64_t)batch * n_pts < ((int64_t)1 << 31) && (int64_t)batch * (fh + 1) * (fw + 1) < ((int64_t)1 << 30), "local_query_bwd_sorted: too many points / pixels for 32-bit keys");
    E3DGE_REQUIRE(pts && calibs, "local_query_bwd_sorted: null pointer");
    const int64_t total = (int64_t)batch * n_pts;
    LqSortK k{};
    k.pts = pts; k.calibs = calibs; k.N = n_pts; k.B = batch; k.h = fh; k.w = fw; k.nb = batch * (fh + 1) * (fw + 1) + 1;
    k.keys = ws; k.order = ws + total; k.cursor = ws + 2 * total; k.counts = ws + 3 * total; k.starts = k.counts + k.nb;
    hipStream_t st = as_stream(stream);
    lq_zero_kernel<<<dim3((unsigned)((k.nb + 255) / 256 > 1024 ? 1024 : (k.nb + 255) / 256)), dim3(256), 0, st>>>(k.counts, k.nb);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    lq_key_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(k);
    lq_scan_kernel<<<dim3(1), dim3(1024), 0, st>>>(k);
    lq_scatter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(k);
    int rc = check_launch("local_query_bwd_sorted(sort)");
    if (rc) return rc;
    return local_query_bwd_launch(d_fmap_nhwc, d_pts, d_out, ld, col_off, pts, calibs, fmap_nhwc, batch, n_pts, channels, fh, fw, k.order, stream);
}

static int local_query_bwd_launch(float* d_fmap_nhwc, float* d_pts, const float* d_out, int ld, int col_off, const float* pts,
                                  const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts, int channels, int fh, int fw,
                                  int* sort_ws, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && n_pts >= 0, "local_query_bwd: bad sizes");
    if (batch == 0 || n_pts == 0 || (!d_fmap_nhwc && !d_pts)) return E3DGE_OK;
    E3DGE_REQUIRE(pts && calibs && d_out && fmap_nhwc, "local_query_bwd: null pointer");
    E3DGE_REQUIRE(channels >= 4 && channels % 4 == 0 && fh >= 1 && fw >= 1, "local_query_bwd: feature map needs C %% 4 == 0");
    E3DGE_REQUIRE(ld >= col_off + channels && col_off >= 0, "local_query_bwd: gradient slice [%d, %d) outside ld=%d", col_off, col_off + channels, ld);
    E3DGE_REQUIRE((reinterpret_cast<uintptr_t>(fmap_nhwc) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0, "local_query_bwd: fmap / d_out must be 16-B aligned");
    LocalQueryBwdK k{};
    k.pts = pts; k.calibs = calibs; k.fmap = fmap_nhwc; k.d_out = d_out; k.d_fmap = d_fmap_nhwc; k.d_pts = d_pts; k.N = n_pts; k.B = batch;
    k.C = channels; k.h = fh; k.w = fw; k.ld = ld; k.col_off = col_off; k.order = sort_ws;
    E3DGE_REQUIRE(channels <= 256, "local_query_bwd: at most 256 channels (four per lane)");
    int64_t blocks = (((int64_t)batch * n_pts + kLqRun - 1) / kLqRun + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    local_query_bwd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(k);
    return check_launch("local_query_bwd");
}

extern "C" int e3dge_pos_encoding(float* out, int ld, int col_off, const float* pts, int64_t n_pts, int n_freqs,
                                  e3dge_stream_t stream) {
    E3DGE_REQUIRE(n_pts >= 0 && n_freqs >= 0 && n_freqs <= 16, "pos_encoding: bad sizes");
    if (n_pts == 0) return E3DGE_OK;
    E3DGE_REQUIRE(out && pts && col_off >= 0 && ld >= col_off + 3 * (2 * n_freqs + 1), "pos_encoding: output slice outside ld");
    const int64_t total = n_pts * 3 * (2 * n_freqs + 1);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    pos_encoding_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(out, pts, n_pts, n_freqs, ld, col_off);
    return check_launch("pos_encoding");
}
