// Device helpers shared by the gather-GEMM kernels.
#pragma once
#include "mt_common.h"
#include "conv_params.h"

template <bool BF16>
__device__ __forceinline__ void mma_chunk(f32x4& acc, const u32x4& a, const u32x4& b) {
  if constexpr (BF16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < 4; s++)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[s]), __uint_as_float(b[s]),
                                                 acc, 0, 0, 0);
  }
}

// bijective XCD-aware remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
// contiguous range of tiles so neighbouring pixel tiles (shared halo rows) hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

