// Probe: semantics of ds_read_b64_tr_b16 on gfx950 with NON-affine per-lane addresses.
// Lane l reads 8 bytes at LDS byte perm[l]*8; LDS element value = its index.  For each result
// element we print which lane's address it was fetched through and which 16-bit element.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* perm, short* out) {
  __shared__ short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + perm[threadIdx.x] * 4));
  for (int e = 0; e < 4; e++) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d; short h[256]; int* dp; int perm[64], inv[64];
  (void)hipMalloc(&d, 512); (void)hipMalloc(&dp, 256);
  srand(7);
  for (int i = 0; i < 64; i++) perm[i] = i;
  for (int i = 63; i > 0; i--) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
  for (int i = 0; i < 64; i++) inv[perm[i]] = i;
  (void)hipMemcpy(dp, perm, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dp, d);
  (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; e++) {
      int src_lane = inv[h[l * 4 + e] / 4], src_e = h[l * 4 + e] % 4;
      int g = l & ~15, i = l & 15;
      int exp_lane = g + 4 * e + i / 4, exp_e = i % 4;
      if (src_lane != exp_lane || src_e != exp_e) bad++;
      printf(" (L%2d,e%d)", src_lane, src_e);
    }
    printf("\n");
  }
  printf("mismatches vs H1 (lane 16g+4r+i/4, elem i%%4): %d\n", bad);
  return 0;
}
