// dpp_f64_rate.hip -- can the object loop take its per-object operands from VECTOR registers instead of the scalar data path?
// gfx90a+ allow DPP on the double-precision ALU with ONE control: row_newbcast:N (lane N of every row of 16 feeds the whole row).
// A VGPR pair then holds one field of 16 objects (the same 16 in each of the 4 rows) and `v_fmac_f64_dpp acc, obj row_newbcast:j, ray`
// multiplies object j's field with every lane's ray value -- no s_load, no constant-bus operand.
//   A  v_fmac_f64 acc, v, v                 (8 independent accumulators)         -- the VALU float64 rate
//   B  v_fmac_f64_dpp ... row_newbcast:j    (same count, j cycling 0..15)        -- does DPP cost issue slots?
//   C  v_fma_f64 acc, s, v, acc with s_load_dwordx8 per 11 VALU (the current loop's shape: 32 B of scalar data per object)
//   D  the same with 40 B per object (dwordx8 + dwordx2)
// Reports ns per wave-instruction per SIMD (1 / 2.4 GHz = 0.417 ns per cycle; a float64 VALU op issues in 4 cycles = 1.67 ns).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_f64_rate.hip -o tools/ubench/dpp_f64_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMAC_DPP(acc, obj, ray, J) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(obj), "v"(ray))
#define FMAC(acc, obj, ray) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(obj), "v"(ray))

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* __restrict__ c, int iters) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double obj = c[threadIdx.x & 15] * 1e-9, ray = 1.0 + threadIdx.x * 1e-12;
  if (MODE == 0) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 2; ++u) { FMAC(a0, obj, ray); FMAC(a1, obj, ray); FMAC(a2, obj, ray); FMAC(a3, obj, ray); FMAC(a4, obj, ray); FMAC(a5, obj, ray); FMAC(a6, obj, ray); FMAC(a7, obj, ray); }
    }
  } else if (MODE == 1) {
    for (int i = 0; i < iters; ++i) {
      FMAC_DPP(a0, obj, ray, 0); FMAC_DPP(a1, obj, ray, 1); FMAC_DPP(a2, obj, ray, 2); FMAC_DPP(a3, obj, ray, 3);
      FMAC_DPP(a4, obj, ray, 4); FMAC_DPP(a5, obj, ray, 5); FMAC_DPP(a6, obj, ray, 6); FMAC_DPP(a7, obj, ray, 7);
      FMAC_DPP(a0, obj, ray, 8); FMAC_DPP(a1, obj, ray, 9); FMAC_DPP(a2, obj, ray, 10); FMAC_DPP(a3, obj, ray, 11);
      FMAC_DPP(a4, obj, ray, 12); FMAC_DPP(a5, obj, ray, 13); FMAC_DPP(a6, obj, ray, 14); FMAC_DPP(a7, obj, ray, 15);
    }
  } else {
    // scalar operands: one record (32 or 40 bytes) per 11 float64 VALU instructions, next record requested one ahead
    typedef const double __attribute__((address_space(4)))* cdptr;
    cdptr rec = (cdptr)(unsigned long long)c;
    double n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3], n4 = rec[4];
    for (int i = 0; i < iters; ++i) {
      const int base = (i & 255) * 8;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double c0 = n0, c1 = n1, c2 = n2, c3 = n3, c4 = (MODE == 3) ? n4 : 0.5;
        n0 = rec[base + 6 * (u + 1) + 0]; n1 = rec[base + 6 * (u + 1) + 1]; n2 = rec[base + 6 * (u + 1) + 2]; n3 = rec[base + 6 * (u + 1) + 3];
        if (MODE == 3) n4 = rec[base + 6 * (u + 1) + 4];
        a0 = __builtin_fma(a0, c0, ray); a1 = __builtin_fma(a1, c1, ray); a2 = __builtin_fma(a2, c2, ray); a3 = __builtin_fma(a3, c3, ray);
        a4 = __builtin_fma(a4, c4, ray); a5 = __builtin_fma(a5, c0, ray); a6 = __builtin_fma(a6, c1, ray); a7 = __builtin_fma(a7, c2, ray);
        a0 = __builtin_fma(a0, c3, ray);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char* name, int wg_per_cu, int per_iter) {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * wg_per_cu;
  double *out, *c; (void)hipMalloc(&out, (size_t)blocks * 256 * 8); (void)hipMalloc(&c, 8 * 4096);
  double hc[4096]; for (int i = 0; i < 4096; ++i) hc[i] = 1.0 + i * 1e-6;
  (void)hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
  const int iters = 40000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, c, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, c, iters); (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 * iters * per_iter;  // wave-instructions
  const double simds = prop.multiProcessorCount * 4.0;
  printf("%-52s waves/SIMD=%d  %8.3f ms  %.3f ns per float64 VALU wave-instruction per SIMD\n", name, wg_per_cu, ms, ms * 1e6 * simds / insts);
  (void)hipFree(out); (void)hipFree(c);
}

int main() {
  for (int w : {1, 2, 3, 4}) {
    run<0>("A  v_fmac_f64 (VGPR operands)", w, 16);
    run<1>("B  v_fmac_f64_dpp row_newbcast:j", w, 16);
    run<2>("C  fma with scalar operands, 32 B / 9 VALU", w, 72);
    run<3>("D  fma with scalar operands, 40 B / 9 VALU", w, 72);
  }
  return 0;
}
