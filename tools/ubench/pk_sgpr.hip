// Does v_pk_add_f32 / v_pk_fma_f32 read BOTH halves of an SGPR-pair source on gfx950?
// (uniform values loaded through the scalar data path, as the TOR_ACCEL_F32 loop does)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4))) * cfptr;
__global__ void k(const float* u, const float* v, float* out) {
  cfptr s = (cfptr)(uintptr_t)u;
  f2 a = {s[0], s[1]};          // SGPR pair (1, 2)
  f2 b = {s[2], s[3]};          // SGPR pair (10, 20)
  int l = threadIdx.x;
  f2 x = {v[2 * l], v[2 * l + 1]};
  f2 r0 = x - a;                                    // v_pk_add_f32 v, v, s neg
  f2 r1 = __builtin_elementwise_fma(x, x, -b);      // v_pk_fma_f32 v, v, s neg
  f2 r2 = a + x;
  out[6 * l + 0] = r0.x; out[6 * l + 1] = r0.y; out[6 * l + 2] = r1.x; out[6 * l + 3] = r1.y;
  out[6 * l + 4] = r2.x; out[6 * l + 5] = r2.y;
}
int main() {
  float hu[4] = {1.f, 2.f, 10.f, 20.f}, hv[128], ho[384];
  for (int i = 0; i < 128; ++i) hv[i] = 100.f + i;
  float *u, *v, *o;
  hipMalloc(&u, 16); hipMalloc(&v, 512); hipMalloc(&o, 1536);
  hipMemcpy(u, hu, 16, hipMemcpyHostToDevice); hipMemcpy(v, hv, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, u, v, o);
  hipMemcpy(ho, o, 1536, hipMemcpyDeviceToHost);
  for (int l = 0; l < 2; ++l)
    printf("lane %d x=(%g,%g): x-a=(%g,%g) want (%g,%g); x*x-b=(%g,%g) want (%g,%g); a+x=(%g,%g) want (%g,%g)\n", l, hv[2*l], hv[2*l+1],
           ho[6*l], ho[6*l+1], hv[2*l]-1, hv[2*l+1]-2, ho[6*l+2], ho[6*l+3], hv[2*l]*hv[2*l]-10, hv[2*l+1]*hv[2*l+1]-20,
           ho[6*l+4], ho[6*l+5], hv[2*l]+1, hv[2*l+1]+2);
  return 0;
}
