// mfma_f64_rate.hip -- GATE for "the FMA screen as a contraction on the float64 matrix pipe" (DESIGN 4.2).
// Question: does v_mfma_f64_16x16x4_f64 add float64 throughput ON TOP of the VALU's v_fma_f64 on gfx950, or do the two share
// one issue / one datapath?  Measured on the whole machine (every SIMD loaded), in wave-instructions and in flop:
//   A  v_fma_f64 only                       (8 independent chains per lane)
//   B  v_mfma_f64_16x16x4_f64 only          (4 independent accumulators)
//   C  both in ONE wave's stream            (1 MFMA : R fma, R = 4 / 8 / 16)
//   D  both from DIFFERENT waves of one SIMD (512-thread workgroups: waves 0-3 MFMA, waves 4-7 fma -- one of each per SIMD)
// "pair rate": the screen costs 11-14 v_*_f64 per 64 (ray, object) pairs today; as a contraction it would cost (2 MFMA per
// 256 pairs for static spheres, 4 for moving ones) + 2 v_fma_f64 + 2 integer ops per 64 pairs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

// role bits: 1 = MFMA, 2 = VALU fma; R = fma per MFMA in a mixed stream
template <int R>
__device__ __forceinline__ void body(int role, int iters, double s0, double s1, double& fsum, v4d& msum) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double ma = a0 * 1e-3, mb = a1 * 1e-3;
  if (role == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, ma, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, ma, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, mb, c3, 0, 0, 0);
      }
    }
  } else if (role == 2) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4 * R / 8 + (4 * R < 8 ? 1 : 0); ++u) {
        a0 = __builtin_fma(a0, s1, s0); a1 = __builtin_fma(a1, s1, s0); a2 = __builtin_fma(a2, s1, s0); a3 = __builtin_fma(a3, s1, s0);
        a4 = __builtin_fma(a4, s1, s0); a5 = __builtin_fma(a5, s1, s0); a6 = __builtin_fma(a6, s1, s0); a7 = __builtin_fma(a7, s1, s0);
      }
    }
  } else {  // 3: one stream, 4 MFMA + 4 R fma per iteration, interleaved
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m == 0) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0, 0, 0);
        if (m == 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, ma, c1, 0, 0, 0);
        if (m == 2) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, ma, c2, 0, 0, 0);
        if (m == 3) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, mb, c3, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          switch (r & 7) {
            case 0: a0 = __builtin_fma(a0, s1, s0); break;
            case 1: a1 = __builtin_fma(a1, s1, s0); break;
            case 2: a2 = __builtin_fma(a2, s1, s0); break;
            case 3: a3 = __builtin_fma(a3, s1, s0); break;
            case 4: a4 = __builtin_fma(a4, s1, s0); break;
            case 5: a5 = __builtin_fma(a5, s1, s0); break;
            case 6: a6 = __builtin_fma(a6, s1, s0); break;
            default: a7 = __builtin_fma(a7, s1, s0); break;
          }
        }
      }
    }
  }
  fsum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  msum = c0 + c1 + c2 + c3;
}

// MODE 0: every wave VALU; 1: every wave MFMA; 2: every wave the mixed stream; 3: waves 0..3 of the workgroup MFMA, 4..7 VALU
template <int MODE, int R, int THREADS>
__global__ __launch_bounds__(THREADS) void k(double* out, const double* __restrict__ c, int iters) {
  const double s0 = c[0], s1 = c[1];
  double f = 0;
  v4d m = {0, 0, 0, 0};
  int role = MODE == 0 ? 2 : (MODE == 1 ? 1 : (MODE == 2 ? 3 : ((threadIdx.x >> 6) < 4 ? 1 : 2)));
  role = __builtin_amdgcn_readfirstlane(role);
  body<R>(role, iters, s0, s1, f, m);
  out[blockIdx.x * blockDim.x + threadIdx.x] = f + m.x + m.y + m.z + m.w;
}

struct Result { double ms, mfma, fma; };

template <int MODE, int R, int THREADS>
Result run(const char* name, int wg_per_cu, int iters) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * wg_per_cu;
  double *out, *c; hipMalloc(&out, (size_t)blocks * THREADS * 8); hipMalloc(&c, 16);
  double hc[2] = {1e-9, 1.0000001}; hipMemcpy(c, hc, 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, R, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, c, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, R, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, c, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * (THREADS / 64);
  double n_mfma = 0, n_fma = 0;  // wave-instructions
  const int fma_per_iter = (4 * R / 8 + (4 * R < 8 ? 1 : 0)) * 8;
  if (MODE == 0) n_fma = waves * iters * fma_per_iter;
  if (MODE == 1) n_mfma = waves * iters * 16.0;
  if (MODE == 2) { n_mfma = waves * iters * 4.0; n_fma = waves * iters * 4.0 * R; }
  if (MODE == 3) { n_mfma = waves / 2 * iters * 16.0; n_fma = waves / 2 * iters * fma_per_iter; }
  const double simds = prop.multiProcessorCount * 4.0;
  const double tflop = (n_mfma * 2048.0 + n_fma * 128.0) / (ms * 1e-3) / 1e12;
  printf("%-44s wg/CU=%d  %8.3f ms  mfma %6.2f G/s/SIMD  fma %6.2f G/s/SIMD  -> %6.1f TFLOP/s  (ns per: mfma %.2f fma %.2f)\n", name, wg_per_cu, ms,
         n_mfma / (ms * 1e-3) / simds / 1e9, n_fma / (ms * 1e-3) / simds / 1e9, tflop, n_mfma > 0 ? ms * 1e6 * simds / n_mfma : 0.0,
         n_fma > 0 ? ms * 1e6 * simds / n_fma : 0.0);
  hipFree(out); hipFree(c);
  return {ms, n_mfma, n_fma};
}

int main() {
  const int it = 20000;
  for (int w : {1, 2}) {
    run<0, 8, 256>("A  v_fma_f64 only (4 waves/WG)", w, it);
    run<1, 8, 256>("B  v_mfma_f64_16x16x4 only (4 waves/WG)", w, it / 4);
    run<2, 4, 256>("C  one stream, 1 MFMA : 4 fma", w, it / 4);
    run<2, 8, 256>("C  one stream, 1 MFMA : 8 fma", w, it / 4);
    run<2, 16, 256>("C  one stream, 1 MFMA : 16 fma", w, it / 4);
  }
  // D: 512-thread workgroups, one MFMA wave + one fma wave per SIMD.  The fma wave runs 4R fma per 16 MFMA of its partner.
  run<0, 8, 512>("A' fma only, 8 waves/WG (2 per SIMD)", 1, it);
  run<1, 8, 512>("B' MFMA only, 8 waves/WG (2 per SIMD)", 1, it / 4);
  run<3, 8, 512>("D  MFMA wave + fma wave per SIMD, 16 : 32", 1, it / 4);
  run<3, 16, 512>("D  MFMA wave + fma wave per SIMD, 16 : 64", 1, it / 4);
  run<3, 32, 512>("D  MFMA wave + fma wave per SIMD, 16 : 128", 1, it / 4);
  run<3, 64, 512>("D  MFMA wave + fma wave per SIMD, 16 : 256", 1, it / 4);
  return 0;
}
