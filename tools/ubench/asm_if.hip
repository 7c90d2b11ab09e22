// Micro test of the role flags + role-conditional asm statements of csrc/siren16_bwd.h (round 6): which waves execute the guarded
// instruction?  Expected: waves 0-3 "w_role 1 s_role 0 executed 000", waves 4-7 "w_role 0 s_role 1 executed 111".
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/asm_if.hip -o /tmp/asm_if && /tmp/asm_if [bad]
// `bad` selects the first attempt (flags read back with v_readfirstlane inside an asm statement: the compiler cannot insert the
// wait states of the VALU-writes-SGPR -> SALU-reads hazard there) -- on the MI355X the flags come out inverted / stale.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ int bad_scalar(int v) { int r; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(r) : "v"(v)); return r; }
__device__ __forceinline__ void t3_roles(int wave_s, int& w_role, int& s_role) {
    int w, st;
    asm volatile("s_cmp_lt_u32 %2, 4\n\ts_cselect_b32 %0, 1, 0\n\ts_cselect_b32 %1, 0, 1" : "=s"(w), "=s"(st) : "s"(wave_s) : "scc");
    w_role = w; s_role = st;
}
template <bool BAD> __global__ void k(int* out) {
    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    int w_role, s_role;
    if (BAD) { w_role = bad_scalar(wave_u < 4 ? 1 : 0); s_role = bad_scalar(1 - w_role); }
    else t3_roles(wave_u, w_role, s_role);
    int v = 0, v2 = 0, v3 = 0;
    asm volatile("s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 .Lsk%=\n\tv_mov_b32 %0, 1\n.Lsk%=:" : "+v"(v) : "s"(s_role) : "scc");
    asm volatile("s_cmp_lg_u32 %1, 0\n\ts_cbranch_scc0 .Lsk%=\n\tv_mov_b32 %0, 1\n.Lsk%=:" : "+v"(v2) : "s"(s_role) : "scc");
    asm volatile("s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 .Lsk%=\n\ts_nop 0\n\tv_mov_b32 %0, 1\n\ts_nop 0\n.Lsk%=:\n\ts_nop 0" : "+v"(v3) : "s"(s_role) : "scc");
    int sr, wr;
    asm volatile("v_mov_b32 %0, %1" : "=v"(sr) : "s"(s_role));
    asm volatile("v_mov_b32 %0, %1" : "=v"(wr) : "s"(w_role));
    out[tid * 4 + 0] = v + 10 * v2 + 100 * v3; out[tid * 4 + 1] = sr; out[tid * 4 + 2] = wr; out[tid * 4 + 3] = wave_u;
}
int main(int argc, char** argv) {
    int* out; (void)hipMalloc(&out, 512 * 16);
    if (argc > 1 && !strcmp(argv[1], "bad")) hipLaunchKernelGGL(k<true>, dim3(1), dim3(512), 0, 0, out);
    else hipLaunchKernelGGL(k<false>, dim3(1), dim3(512), 0, 0, out);
    hipError_t e = hipDeviceSynchronize(); printf("sync: %s\n", hipGetErrorString(e));
    int r[2048]; (void)hipMemcpy(r, out, 8192, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) printf("wave %d: w_role %d  s_role %d  executed %03d  (wave_u %d)\n", w, r[w * 256 + 2], r[w * 256 + 1], r[w * 256], r[w * 256 + 3]);
    return 0;
}
