// Probe: HBM write rate of the gather-GEMM epilogue's store pattern (16 pixels x 64 B per wave instruction, pixel pitch
// 512 B) against fully contiguous 1 KiB wave stores, one 128 KiB tile per workgroup (256 workgroups = the K1 launch),
// and the same for a 64-channel tensor (pixel pitch 128 B).   hipcc --offload-arch=gfx950 -O3 probe_store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE, int NT>
__global__ __launch_bounds__(512) void k(u4* y, int pitch16, int tiles_per_block) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const u4 v = {1u, 2u, 3u, (unsigned)tid};
  for (int t = 0; t < tiles_per_block; t++) {
    u4* tile = y + ((size_t)blockIdx.x * tiles_per_block + t) * 256 * pitch16;   // 256 pixels x pitch
    if (MODE == 0) {
      // epilogue order: wave (wcI = wv / 4 -> 128 channels = 256 B, wpI = wv % 4 -> 64 pixels); sp: 64-byte segment; b: 16 pixels
      const int wcI = wv >> 2, wpI = wv & 3;
      for (int sp = 0; sp < pitch16 / 8; sp++)
        for (int b = 0; b < 4; b++) {
          const int px = wpI * 64 + b * 16 + fr;
          const int c16 = (pitch16 == 32 ? wcI * 16 : 0) + sp * 4 + fg;
          if (pitch16 == 32 || wcI == 0 || true) {
            if (pitch16 == 32) __builtin_nontemporal_store(v, &tile[px * pitch16 + c16]);
            else if (wcI == 0) __builtin_nontemporal_store(v, &tile[px * pitch16 + c16]);
          }
        }
    } else {
      // contiguous: every wave instruction writes 1 KiB in a row
      const int total16 = 256 * pitch16;
      for (int i = tid; i < total16; i += NT) __builtin_nontemporal_store(v, &tile[i]);
    }
  }
}
int main() {
  const size_t bytes = 512ull << 20;
  u4* y; (void)hipMalloc(&y, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int pitch16 : {32, 8}) {            // 512-byte pixels (256 channels), 128-byte pixels (64 channels)
    const int tile_bytes = 256 * pitch16 * 16;
    for (int tpb : {1, 4}) {
      const int blocks = 256;
      const double mb = (double)blocks * tpb * tile_bytes / 1e6;
      for (int mode = 0; mode < 2; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 20; rep++) {
          (void)hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL((k<0, 512>), dim3(blocks), dim3(512), 0, 0, y, pitch16, tpb);
          else hipLaunchKernelGGL((k<1, 512>), dim3(blocks), dim3(512), 0, 0, y, pitch16, tpb);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          if (rep > 2 && ms < best) best = ms;
        }
        printf("pixel pitch %3d B, %d tile(s)/block, %6.1f MB: %s  %.1f us  %.2f TB/s\n", pitch16 * 16, tpb, mb,
               mode == 0 ? "epilogue pattern" : "contiguous     ", best * 1e3, mb / best / 1e3 / 1e3 * 1e3);
      }
    }
  }
  return 0;
}
