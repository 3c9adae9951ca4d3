// fp64_rate.hip -- microbenchmark: sustained float64 VALU issue rate on gfx950 for the instruction
// mixes the integrator's object loop uses (v_add_f64 / v_mul_f64 with an SGPR operand, no FMA).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/fp64_rate.hip -o /tmp/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* __restrict__ c, int iters) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double s0 = c[0], s1 = c[1];  // uniform -> SGPRs
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {  // 8 independent add chains
        a0 += s0; a1 += s0; a2 += s0; a3 += s0; a4 += s0; a5 += s0; a6 += s0; a7 += s0;
      } else if (MODE == 1) {  // alternating mul / add, independent chains
        a0 *= s1; a1 += s0; a2 *= s1; a3 += s0; a4 *= s1; a5 += s0; a6 *= s1; a7 += s0;
      } else if (MODE == 2) {  // one dependent chain of adds
        a0 += s0; a0 += s1; a0 += s0; a0 += s1; a0 += s0; a0 += s1; a0 += s0; a0 += s1;
      } else {  // fma
        a0 = __builtin_fma(a0, s1, s0); a1 = __builtin_fma(a1, s1, s0); a2 = __builtin_fma(a2, s1, s0); a3 = __builtin_fma(a3, s1, s0);
        a4 = __builtin_fma(a4, s1, s0); a5 = __builtin_fma(a5, s1, s0); a6 = __builtin_fma(a6, s1, s0); a7 = __builtin_fma(a7, s1, s0);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int blocks = prop.multiProcessorCount * blocks_per_cu;
  double *out, *c; hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&c, 16);
  double hc[2] = {1e-9, 1.0000001}; hipMemcpy(c, hc, 16, hipMemcpyHostToDevice);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, c, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, c, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)blocks * 256 * iters * 64.0;  // lane-instructions
  printf("%-28s waves/SIMD=%d  %.2f T lane-ops/s  (%.1f%% of 39.3)  %.1f ms\n", name, blocks_per_cu, insts / ms / 1e9, insts / ms / 1e9 / 39.3 * 100, ms);
  hipFree(out); hipFree(c);
}

int main() {
  for (int b : {1, 2, 4}) {
    run<0>("add x8 independent", b);
    run<1>("mul/add independent", b);
    run<2>("add dependent chain", b);
    run<3>("fma x8 independent", b);
  }
  return 0;
}
