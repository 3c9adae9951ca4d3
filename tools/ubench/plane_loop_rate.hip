// plane_loop_rate.hip -- what bounds stage one of the plane-screened object loop (kernel/integrate_loop_plane.inc)?  Per object and
// wave: 16 bytes through the scalar data path, v_fma_f64 x 3 (a dependent chain) + v_alignbit_b32.  Measured on the whole machine
// (256 CUs x 3 workgroups x 4 waves, as the kernel runs), ns per object and SIMD-resident wave:
//   0  the loop as it is compiled today (one s_load_dwordx16 per 4 objects, one ahead)
//   1  the same arithmetic on loop-invariant scalar operands (no loads in the loop): the VALU's own rate for this chain
//   2  the loads alone (one v_alignbit per object keeps them alive): the scalar data path's rate
//   3  v_cmp_lt_f64 |s|, thr + v_addc_co_u32 instead of the squaring fma + v_alignbit (2 fma + cmp + addc)
//   4  two objects' chains interleaved by hand (separate temporaries)
//   5  as 0 with a scheduling barrier behind every object: the chains strictly one after the other, as integrate_kernel's register
//      budget makes the compiler schedule them (0 itself comes out interleaved four deep here)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/plane_loop_rate.hip -o /tmp/plane_loop_rate && /tmp/plane_loop_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const double __attribute__((address_space(4))) * cdptr;

__device__ __forceinline__ unsigned push_bit(unsigned m, double q) {
  return __builtin_amdgcn_alignbit(m, (unsigned)((unsigned long long)__double_as_longlong(q) >> 32), 31);
}
__device__ __forceinline__ unsigned push_cmp(unsigned m, double s, double thr) {
  unsigned long long mask, carry;
  unsigned r;
  asm("v_cmp_lt_f64_e64 %0, |%1|, %2" : "=s"(mask) : "v"(s), "v"(thr));
  asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry) : "v"(m), "s"(mask));
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void k(const double* tab, int n_obj, int iters, const double* rays, unsigned* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const double nx = rays[(t & 1023) * 4 + 0], nz = rays[(t & 1023) * 4 + 1], c0 = rays[(t & 1023) * 4 + 2], negthr = rays[(t & 1023) * 4 + 3];
  unsigned m = 0, acc = 0;
  cdptr base = (cdptr)(uintptr_t)tab;
  for (int it = 0; it < iters; ++it) {
    cdptr rec = base;
    if (MODE == 1) {
      const double a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4], a5 = rec[5], a6 = rec[6], a7 = rec[7];
      for (int i = 0; i < n_obj; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double cx = (j & 3) == 0 ? a0 : ((j & 3) == 1 ? a2 : ((j & 3) == 2 ? a4 : a6));
          const double cz = (j & 3) == 0 ? a1 : ((j & 3) == 1 ? a3 : ((j & 3) == 2 ? a5 : a7));
          const double s = __builtin_fma(nx, cx, __builtin_fma(nz, cz, c0));
          m = push_bit(m, __builtin_fma(s, s, negthr));
        }
        acc ^= m;
        asm volatile("" : "+v"(m));
      }
    } else {
      double n0 = rec[0], n1 = rec[1];
      for (int i = 0; i < n_obj; i += 8) {
        if (MODE == 4) {
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const double cx = n0, cz = n1, dx = rec[2 * (j + 1)], dz = rec[2 * (j + 1) + 1];
            n0 = rec[2 * (j + 2)]; n1 = rec[2 * (j + 2) + 1];
            double s0 = __builtin_fma(nz, cz, c0), s1 = __builtin_fma(nz, dz, c0);
            asm volatile("" : "+v"(s0), "+v"(s1));
            s0 = __builtin_fma(nx, cx, s0); s1 = __builtin_fma(nx, dx, s1);
            asm volatile("" : "+v"(s0), "+v"(s1));
            double q0 = __builtin_fma(s0, s0, negthr), q1 = __builtin_fma(s1, s1, negthr);
            asm volatile("" : "+v"(q0), "+v"(q1));
            m = push_bit(push_bit(m, q0), q1);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double cx = n0, cz = n1;
            n0 = rec[2 * (j + 1)]; n1 = rec[2 * (j + 1) + 1];
            if (MODE == 2) {
              m = __builtin_amdgcn_alignbit(m, (unsigned)((unsigned long long)__double_as_longlong(cx) >> 32) ^ (unsigned)((unsigned long long)__double_as_longlong(cz) >> 32), 31);
            } else {
              const double s = __builtin_fma(nx, cx, __builtin_fma(nz, cz, c0));
              if (MODE == 3) m = push_cmp(m, s, -negthr);
              else m = push_bit(m, __builtin_fma(s, s, negthr));
              if (MODE == 5) __builtin_amdgcn_sched_barrier(0);  // one object after the other, as the kernel's loop is scheduled (one temporary)
            }
          }
        }
        rec += 16;
        acc ^= m;
      }
    }
  }
  out[t] = acc;
}

template <int MODE>
void run(const char* name, const double* tab, int n_obj, const double* rays, unsigned* out, int cus) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(cus * 3), dim3(256), 0, 0, tab, n_obj, 100, rays, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(cus * 3), dim3(256), 0, 0, tab, n_obj, iters, rays, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // each SIMD hosts 3 waves; SIMD time per object and wave = kernel time / (iters * n_obj * 3)
  printf("%-62s %8.2f ms   %6.2f ns per object and wave (SIMD time; 3 waves per SIMD)\n", name, ms, ms * 1e6 / ((double)iters * n_obj * 3));
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int n_obj = 496;
  std::vector<double> tab(2 * n_obj + 64), rays(4096);
  for (int i = 0; i < n_obj; ++i) { tab[2 * i] = -11.0 + 22.0 * ((i * 37) % 101) / 101.0; tab[2 * i + 1] = -11.0 + 22.0 * ((i * 53) % 103) / 103.0; }
  for (int i = 0; i < 1024; ++i) { rays[4 * i] = 0.6; rays[4 * i + 1] = 0.8; rays[4 * i + 2] = 0.01 * i; rays[4 * i + 3] = -0.04; }
  double *d_tab, *d_rays; unsigned* d_out;
  hipMalloc(&d_tab, tab.size() * 8); hipMalloc(&d_rays, rays.size() * 8); hipMalloc(&d_out, (size_t)prop.multiProcessorCount * 3 * 256 * 4);
  hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_rays, rays.data(), rays.size() * 8, hipMemcpyHostToDevice);
  const int cus = prop.multiProcessorCount;
  run<0>("0  the loop as compiled (s_load x16 per 4 objects, one ahead)", d_tab, n_obj, d_rays, d_out, cus);
  run<1>("1  same arithmetic, loop-invariant scalar operands (no loads)", d_tab, n_obj, d_rays, d_out, cus);
  run<2>("2  loads + one v_alignbit per object", d_tab, n_obj, d_rays, d_out, cus);
  run<3>("3  2 fma + v_cmp_lt_f64 |s| + v_addc_co_u32", d_tab, n_obj, d_rays, d_out, cus);
  run<4>("4  two objects' chains interleaved", d_tab, n_obj, d_rays, d_out, cus);
  run<5>("5  as 0, one object strictly after the other (no interleaving)", d_tab, n_obj, d_rays, d_out, cus);
  return 0;
}
