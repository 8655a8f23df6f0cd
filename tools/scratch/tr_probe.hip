// probe of ds_read_b64_tr_b16 lane semantics (see cdna_hip_programming.md T10)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define P 40  // row pitch in shorts (80 B: multiple of 8 B)
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * P];
  for (int i = threadIdx.x; i < 64 * P; i += 64) lds[i] = (short)((i / P) * 100 + (i % P));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, t = l & 15, q = t >> 2, p = t & 3;
  const int r0 = 8 * (g >> 1), c0 = 16 * (g & 1);   // groups 0,1: rows 0-3 cols 0-15 / 16-31; groups 2,3: rows 8-11
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + (r0 + q) * P + c0 + 4 * p));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int g = l >> 4, t = l & 15, r0 = 8 * (g >> 1), c0 = 16 * (g & 1);
    for (int e = 0; e < 4; ++e) { int exp = (r0 + e) * 100 + c0 + t; if (h[l*4+e] != exp) { if (bad < 8) printf("lane %d e %d got %d exp %d\n", l, e, h[l*4+e], exp); bad++; } }
  }
  printf("lane0: %d %d %d %d | lane17: %d %d %d %d | lane35: %d %d %d %d  bad=%d\n", h[0],h[1],h[2],h[3], h[68],h[69],h[70],h[71], h[140],h[141],h[142],h[143], bad);
  return bad != 0;
}
