// mix_f64_f32_rate.hip -- what does a 32-bit VALU instruction cost BESIDE float64 work on gfx950?  Every wave runs 9 v_fma_f64
// (independent accumulators) per iteration plus N extra instructions of one kind:
//   f32  : v_fma_f32            i32 : v_bitop3_b32 / v_alignbit_b32 (the integer instructions of the object loop)
//   cvt  : v_cvt_f32_f64 + v_cvt_f64_f32 pairs         pk : v_pk_fma_f32
// ns per iteration per SIMD at 3 waves / SIMD; the slope per extra instruction is its marginal cost.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mix_f64_f32_rate.hip -o tools/ubench/mix_f64_f32_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__global__ __launch_bounds__(256) void k(double* out, const double* __restrict__ c, int iters) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, a8 = a0 + 8;
  const double s0 = c[0], s1 = c[1];
  float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2_ = f0 + 2, f3 = f0 + 3;
  const float g0 = (float)s0, g1 = (float)s1;
  unsigned u0 = threadIdx.x, u1 = u0 * 7u, u2 = u0 * 13u;
  f2 p0 = {f0, f1}, p1 = {f2_, f3};
  const f2 q0 = {g0, g1}, q1 = {g1, g0};
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fma(a0, s1, s0); a1 = __builtin_fma(a1, s1, s0); a2 = __builtin_fma(a2, s1, s0);
    a3 = __builtin_fma(a3, s1, s0); a4 = __builtin_fma(a4, s1, s0); a5 = __builtin_fma(a5, s1, s0);
    a6 = __builtin_fma(a6, s1, s0); a7 = __builtin_fma(a7, s1, s0); a8 = __builtin_fma(a8, s1, s0);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      if (KIND == 0) { if (n & 1) f0 = __builtin_fmaf(f0, g1, g0); else f1 = __builtin_fmaf(f1, g1, g0); }
      else if (KIND == 1) { if (n & 1) u0 = __builtin_amdgcn_alignbit(u0, u1, 31); else u1 = __builtin_amdgcn_bitop3_b32(u0, u1, u2, 0x51); }
      else if (KIND == 2) { if (n & 1) a8 = (double)f0; else f0 = (float)a7; }
      else { if (n & 1) p0 = __builtin_elementwise_fma(p0, q1, q0); else p1 = __builtin_elementwise_fma(p1, q1, q0); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + f0 + f1 + f2_ + f3 + (double)(u0 + u1) + p0.x + p0.y + p1.x + p1.y;
}

template <int KIND, int N>
void run(const char* name) {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 3;
  double *out, *c; (void)hipMalloc(&out, (size_t)blocks * 256 * 8); (void)hipMalloc(&c, 16);
  double hc[2] = {1e-9, 1.0000001}; (void)hipMemcpy(c, hc, 16, hipMemcpyHostToDevice);
  const int iters = 400000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, out, c, 1000);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, out, c, iters); (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // 3 waves per SIMD, each runs `iters` iterations: per SIMD 3 * iters iterations in ms
  printf("%-22s N=%2d  %8.3f ms  %.3f ns per iteration (9 v_fma_f64 + N extra) per SIMD\n", name, N, ms, ms * 1e6 / (3.0 * iters));
  (void)hipFree(out); (void)hipFree(c);
}

int main() {
  run<0, 0>("baseline");
  run<0, 3>("v_fma_f32"); run<0, 6>("v_fma_f32"); run<0, 12>("v_fma_f32");
  run<1, 3>("bitop3/alignbit"); run<1, 6>("bitop3/alignbit"); run<1, 12>("bitop3/alignbit");
  run<2, 2>("cvt f64<->f32"); run<2, 4>("cvt f64<->f32");
  run<3, 3>("v_pk_fma_f32"); run<3, 6>("v_pk_fma_f32");
  return 0;
}
