// dep_latency.hip -- microbenchmark: how many shader cycles does ONE wave need per instruction of a dependent float64 chain
// (the server side of the chain hand-off, DESIGN 4.7 (HISTORY 4.10), is one such chain per bounce), and does a narrower EXEC mask or
// instruction-level parallelism change it?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/dep_latency.hip -o /tmp/dep_latency && /tmp/dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP, int ILP>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, const double* __restrict__ c, int iters, int lanes) {
  const double s0 = c[0], s1 = c[1];
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < lanes) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (OP == 0) { a0 += s0; if (ILP > 1) a1 += s0; if (ILP > 2) { a2 += s0; a3 += s0; } }
        else if (OP == 1) { a0 *= s1; if (ILP > 1) a1 *= s1; if (ILP > 2) { a2 *= s1; a3 *= s1; } }
        else if (OP == 2) { a0 = __builtin_fma(a0, s1, s0); if (ILP > 1) a1 = __builtin_fma(a1, s1, s0); if (ILP > 2) { a2 = __builtin_fma(a2, s1, s0); a3 = __builtin_fma(a3, s1, s0); } }
        else if (OP == 3) { a0 = __builtin_sqrt(a0 + 2.0); }
        else if (OP == 4) { a0 = s1 / (a0 + 2.0); }
        else { float f = (float)a0; f = __builtin_fmaf(f, 1.0000001f, 1e-9f); a0 = f; }
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  out[threadIdx.x] = a0 + a1 + a2 + a3;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP, int ILP>
void run(const char* name, int lanes) {
  double *out, *c; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&c, 16); hipMalloc(&cyc, 8);
  double hc[2] = {1e-9, 1.0000001}; hipMemcpy(c, hc, 16, hipMemcpyHostToDevice);
  const int iters = 4000;
  hipLaunchKernelGGL((k<OP, ILP>), dim3(1), dim3(64), 0, 0, out, cyc, c, 10, lanes);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, ILP>), dim3(1), dim3(64), 0, 0, out, cyc, c, iters, lanes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * ILP;
  printf("%-34s lanes %2d ILP %d: %7.2f counter ticks / instruction, %7.2f ns / instruction (kernel %.3f ms)\n", name, lanes, ILP, h / n, ms * 1e6 / n, ms);
}

int main() {
  for (int lanes : {64, 32, 16, 1}) {
    run<0, 1>("v_add_f64 dependent", lanes);
    run<1, 1>("v_mul_f64 dependent", lanes);
    run<2, 1>("v_fma_f64 dependent", lanes);
  }
  run<0, 2>("v_add_f64 two chains", 64);
  run<0, 4>("v_add_f64 four chains", 64);
  run<2, 2>("v_fma_f64 two chains", 64);
  run<2, 4>("v_fma_f64 four chains", 64);
  run<3, 1>("sqrt(x + 2) dependent (IEEE)", 64);
  run<3, 1>("sqrt(x + 2) dependent (IEEE)", 1);
  run<4, 1>("s / (x + 2) dependent (IEEE)", 64);
  run<5, 1>("cvt f64->f32, v_fma_f32, cvt back", 64);
  return 0;
}
