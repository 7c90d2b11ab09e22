// The 2-D reconstruction metrics of one evaluated image in one pass over (prediction, target): the per-image scalars that
// the sharded evaluation all-gathers (BASELINE.json configs[2]).
//
// Reference: Loss.calc_2d_rec_loss (project/losses/builder.py:130-184): loss_l2 = MSE, mae = L1, PSNR on the [0,1]-scaled
// images, SSIM = 1 - kornia.losses.ssim_loss(pred, gt, window 5) (Gaussian window sigma 1.5, reflect padding,
// C1 = 0.01^2, C2 = 0.03^2, loss = mean(clamp((1 - ssim) / 2, 0, 1))).  The identity / LPIPS terms need pretrained
// networks and are outside the path (reported as 0 by the host code).
//
// One 32x32-pixel tile of one channel per workgroup: both images (+2-pixel reflected halo) staged in LDS once, the five
// windowed moments of every pixel from 25 taps, per-workgroup partial sums (squared error, absolute error, SSIM loss) to a
// scratch row; a second single-workgroup launch folds the rows in a fixed order (bit-reproducible) into
//   out[0..3] = sum of squared errors, sum of absolute errors, sum of SSIM losses, element count.
// Bound: HBM, 8 B per element -- 25 MB for a 3x1024^2 pair.
#include "common.h"

namespace e3dge {

constexpr int kMtTile = 32, kMtHalo = 2, kMtPatch = kMtTile + 2 * kMtHalo;   // 36
constexpr int kMtThreads = 256;

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

__global__ void __launch_bounds__(kMtThreads)
image_metrics_tile_kernel(float* __restrict__ partial, const float* __restrict__ pred, const float* __restrict__ gt,
                          int H, int W, int tiles_x, int tiles_y, float max_val) {
    __shared__ float px[kMtPatch][kMtPatch + 1], py[kMtPatch][kMtPatch + 1];
    __shared__ float red[3][kMtThreads / 64];
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int64_t plane = bid;                                   // (batch * channels) index
    const float* __restrict__ p = pred + plane * (int64_t)H * W;
    const float* __restrict__ g = gt + plane * (int64_t)H * W;
    const int y0 = ty * kMtTile - kMtHalo, x0 = tx * kMtTile - kMtHalo;
    // the SSIM of the reference runs on the images as given ([-1,1] range, max_val = 1.0 in kornia's call)
    for (int e = threadIdx.x; e < kMtPatch * kMtPatch; e += kMtThreads) {
        const int r = e / kMtPatch, c = e - r * kMtPatch;
        const int yy = reflect_idx(y0 + r, H), xx = reflect_idx(x0 + c, W);
        px[r][c] = p[(int64_t)yy * W + xx];
        py[r][c] = g[(int64_t)yy * W + xx];
    }
    // Gaussian window, sigma 1.5, normalised (kornia get_gaussian_kernel1d)
    float g1[5];
    {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i) { g1[i] = expf(-((float)(i - 2) * (float)(i - 2)) / (2.0f * 1.5f * 1.5f)); s += g1[i]; }
#pragma unroll
        for (int i = 0; i < 5; ++i) g1[i] /= s;
    }
    __syncthreads();
    const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
    float se = 0.0f, ae = 0.0f, sl = 0.0f;
    for (int e = threadIdx.x; e < kMtTile * kMtTile; e += kMtThreads) {
        const int r = e / kMtTile, c = e - r * kMtTile;
        const int yy = ty * kMtTile + r, xx = tx * kMtTile + c;
        if (yy >= H || xx >= W) continue;
        float mx = 0.f, my = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            float rx = 0.f, ry = 0.f, rxx = 0.f, ryy = 0.f, rxy = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float a = px[r + i][c + j], b = py[r + i][c + j], w = g1[j];
                rx = fmaf(w, a, rx); ry = fmaf(w, b, ry);
                rxx = fmaf(w, a * a, rxx); ryy = fmaf(w, b * b, ryy); rxy = fmaf(w, a * b, rxy);
            }
            mx = fmaf(g1[i], rx, mx); my = fmaf(g1[i], ry, my);
            sxx = fmaf(g1[i], rxx, sxx); syy = fmaf(g1[i], ryy, syy); sxy = fmaf(g1[i], rxy, sxy);
        }
        sxx -= mx * mx; syy -= my * my; sxy -= mx * my;
        const float ssim = ((2.0f * mx * my + c1) * (2.0f * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2));
        sl += fminf(fmaxf((1.0f - ssim) * 0.5f, 0.0f), 1.0f);
        const float d = px[r + kMtHalo][c + kMtHalo] - py[r + kMtHalo][c + kMtHalo];
        se = fmaf(d, d, se);
        ae += fabsf(d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_xor(se, off, kWave); ae += __shfl_xor(ae, off, kWave); sl += __shfl_xor(sl, off, kWave);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = se; red[1][threadIdx.x >> 6] = ae; red[2][threadIdx.x >> 6] = sl; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float* rr = red[threadIdx.x];
        partial[(int64_t)blockIdx.x * 3 + threadIdx.x] = ((rr[0] + rr[1]) + rr[2]) + rr[3];
    }
}

// fixed-order fold of the per-tile rows of image `b` (double accumulation): out[b] = (sse, sae, ssim-loss sum, count)
__global__ void __launch_bounds__(256)
image_metrics_fold_kernel(float* __restrict__ out, const float* __restrict__ partial, int rows_per_image, float count) {
    __shared__ double acc[3][256];
    const int b = blockIdx.x;
    const float* pr = partial + (int64_t)b * rows_per_image * 3;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < rows_per_image; i += 256) { s0 += pr[i * 3]; s1 += pr[i * 3 + 1]; s2 += pr[i * 3 + 2]; }
    acc[0][threadIdx.x] = s0; acc[1][threadIdx.x] = s1; acc[2][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += acc[threadIdx.x][i];
        out[b * 4 + threadIdx.x] = (float)t;
    }
    if (threadIdx.x == 3) out[b * 4 + 3] = count;
}

// the eight reported columns from the per-image sums (builder.py:174-184): means over the whole batch tensor, PSNR on images
// rescaled to [0, 1] (squared error / 4); the identity / LPIPS columns are the reference's values at lambda = 0
__global__ void image_metric_row_kernel(float* __restrict__ row, const float* __restrict__ sums, int batch, float l2_lambda) {
    if (threadIdx.x != 0) return;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, n = 0.f;
    for (int b = 0; b < batch; ++b) { t0 += sums[b * 4]; t1 += sums[b * 4 + 1]; t2 += sums[b * 4 + 2]; n += sums[b * 4 + 3]; }
    const float mse = t0 / n, mae = t1 / n, ssim_loss = t2 / n;
    row[0] = mse; row[1] = 0.0f; row[2] = 0.0f; row[3] = mse * l2_lambda; row[4] = mae;
    row[5] = 10.0f * log10f(1.0f / (mse * 0.25f));
    row[6] = 1.0f - ssim_loss; row[7] = 1.0f;
}

}  // namespace e3dge

using namespace e3dge;

extern "C" int e3dge_image_metric_row(float* row, const float* sums, int batch, float l2_lambda, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 1, "image_metric_row: batch=%d", batch);
    E3DGE_REQUIRE(row && sums, "image_metric_row: null pointer");
    image_metric_row_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(row, sums, batch, l2_lambda);
    return check_launch("image_metric_row");
}

extern "C" int64_t e3dge_image_metrics_scratch_floats(int batch, int channels, int height, int width) {
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0) return 0;
    return (int64_t)batch * channels * ((height + kMtTile - 1) / kMtTile) * ((width + kMtTile - 1) / kMtTile) * 3;
}

extern "C" int e3dge_image_metrics(float* sums, float* scratch, const float* pred, const float* gt, int batch, int channels,
                                   int height, int width, float max_val, e3dge_stream_t stream) {
    E3DGE_REQUIRE(batch >= 0 && channels >= 1 && height >= 1 && width >= 1, "image_metrics: bad sizes");
    if (batch == 0) return E3DGE_OK;
    E3DGE_REQUIRE(sums && scratch && pred && gt, "image_metrics: null pointer");
    const int tiles_x = (width + kMtTile - 1) / kMtTile, tiles_y = (height + kMtTile - 1) / kMtTile;
    const int64_t blocks = (int64_t)batch * channels * tiles_x * tiles_y;
    E3DGE_REQUIRE(blocks < ((int64_t)1 << 31), "image_metrics: grid too large");
    hipStream_t st = as_stream(stream);
    image_metrics_tile_kernel<<<dim3((unsigned)blocks), dim3(kMtThreads), 0, st>>>(scratch, pred, gt, height, width, tiles_x, tiles_y, max_val);
    int rc = check_launch("image_metrics(tiles)");
    if (rc) return rc;
    image_metrics_fold_kernel<<<dim3((unsigned)batch), dim3(256), 0, st>>>(sums, scratch, channels * tiles_x * tiles_y,
                                                                             (float)((double)channels * height * width));
    return check_launch("image_metrics(fold)");
}
