// µbench: cycles per v_mfma_f32_32x32x2_f32 (and the f16 32x32x16) for 1 / 2 / 4 independent accumulator chains, one
// wave per SIMD (4 waves per workgroup, 256 workgroups), operands from VGPRs.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NACC, bool F16>
__global__ void __launch_bounds__(256) chain(float* out, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = threadIdx.x * 1e-3f, bv = 1.0f + threadIdx.x * 1e-4f;
    half8 ah, bh; for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(av + j); bh[j] = (_Float16)(bv - j); }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if (F16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[a], 0, 0, 0);
                else     acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, bool F16> void run(const char* name) {
    float* out; long long* cyc; (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 64;
    chain<NACC, F16><<<256, 256>>>(out, cyc, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); chain<NACC, F16><<<256, 256>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 32 * NACC;
    double flop = n * (F16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2) * 1024;   // 1024 waves
    printf("%-22s %d acc: %.2f cycles/MFMA (s_memtime), %.1f TFLOP/s, %.3f ms\n", name, NACC, c / n, flop / (ms * 1e-3) / 1e12, ms);
}
int main() {
    run<1, false>("f32 32x32x2"); run<2, false>("f32 32x32x2"); run<4, false>("f32 32x32x2");
    run<1, true>("f16 32x32x16"); run<2, true>("f16 32x32x16"); run<4, true>("f16 32x32x16");
    return 0;
}
