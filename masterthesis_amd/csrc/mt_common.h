// Shared device/host helpers for libmt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mt_api.h"

#define MT_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

void mt_set_error(const char* fmt, ...);

#define MT_CHECK(cond, ...)                 \
  do {                                      \
    if (!(cond)) {                          \
      mt_set_error(__VA_ARGS__);            \
      return 1;                             \
    }                                       \
  } while (0)

#define MT_LAUNCH_CHECK()                                              \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      mt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                   hipGetErrorString(e__));                            \
      return 2;                                                        \
    }                                                                  \
  } while (0)

// ---- bf16 <-> f32 ---------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE, NaN preserving)
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
  return (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
}

// Element-type traits: V = elements per 16-byte chunk.
template <bool BF16> struct Elem;
template <> struct Elem<true> {
  static constexpr int V = 8;
  static constexpr int SZ = 2;
  __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f[2 * i] = __uint_as_float(c[i] << 16);
      f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = pack2_bf16(f[2 * i], f[2 * i + 1]);
    return c;
  }
};
template <> struct Elem<false> {
  static constexpr int V = 4;
  static constexpr int SZ = 4;
  __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; i++) f[i] = __uint_as_float(c[i]);
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = __float_as_uint(f[i]);
    return c;
  }
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case MT_ACT_RELU: return v > 0.f ? v : 0.f;
    case MT_ACT_LRELU: return v > 0.f ? v : v * slope;
    case MT_ACT_TANH: return tanhf(v);
    default: return v;
  }
}
// derivative given pre-activation z
__device__ __forceinline__ float act_grad_z(float z, int act, float slope) {
  switch (act) {
    case MT_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case MT_ACT_LRELU: return z > 0.f ? 1.f : slope;
    case MT_ACT_TANH: { float t = tanhf(z); return 1.f - t * t; }
    default: return 1.f;
  }
}
// derivative given activation output y
__device__ __forceinline__ float act_grad_y(float y, int act, float slope) {
  switch (act) {
    case MT_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case MT_ACT_LRELU: return y > 0.f ? 1.f : slope;
    case MT_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// wave64 sum via DPP-lowered shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the 16 lanes of a DPP row (lanes 16g..16g+15); every lane of the row gets the total.
// quad_perm [1,0,3,2] (xor 1), [2,3,0,1] (xor 2), row_half_mirror, row_mirror: 4 VALU ops, no LDS traffic.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
  return v;
}

// Block reduction for the streaming kernels whose threads are laid out as (pixel lane pl, 16-byte channel
// chunk cq) with cq = tid % cchunks: sum v over the threads that share cq.  When cchunks is a power of two
// <= 32 the lanes of a wave that share cq are first combined with xor-shuffles, so the serial tail handles one
// partial per WAVE instead of one per thread (at cchunks = 8 and 1024 threads: 16 instead of 128 LDS reads per
// output).  `red` needs blockDim.x floats.  The total is returned to the thread with tid < cchunks (others: 0).
__device__ __forceinline__ float block_sum_by_chunk(float v, int cchunks, float* red) {
  const int tid = threadIdx.x;
  const bool pow2 = (cchunks & (cchunks - 1)) == 0;
  int stride = cchunks;
  if (pow2 && cchunks < 64) {
    for (int o = cchunks; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    stride = 64;
  }
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  float a = 0.f;
  if (tid < cchunks) {
    const int cnt = blockDim.x / stride;
    for (int k = 0; k < cnt; k++) a += red[tid + k * stride];
  }
  return a;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
