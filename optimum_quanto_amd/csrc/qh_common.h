// Shared device helpers for libquanto_hip (gfx950 only: wave64, OCP fp8, v_dot2 bf16, MFMA 16x16x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/quanto_hip.h"

namespace qh {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kWave = 64;

// qbits GEMV load policy used when QUANTO_HIP_GEMV_VARIANT is unset (see qbits_gemv.hip): bit 0 = x / scale loads first,
// bit 1 = non-temporal weight loads.  Measured (r2, hipGraph replay, us per launch, variant 0 / 1 / 2 / 3): (1,4096,4096) 4.54 /
// 4.93 / 4.57 / 4.70, (1,4096,11008) 7.49 / 8.33 / 7.40 / 7.99, fused gate+up 2 x (1,4096,14336) 15.5 / 16.0 / 14.2 / 15.4:
// non-temporal loads pay once the stream is long, forcing the small loads to the front never does (hipcc already
// issues them first where it matters).
#ifndef QUANTO_HIP_GEMV_DEFAULT_VARIANT
#define QUANTO_HIP_GEMV_DEFAULT_VARIANT 2
#endif

// ---- element traits for the three float dtypes of the ABI ------------------------------------
template <int DT>
struct Elem;
template <>
struct Elem<QUANTO_HIP_F32> {
  using T = float;
  static __device__ __forceinline__ float to_f32(T v) { return v; }
  static __device__ __forceinline__ T from_f32(float f) { return f; }
};
template <>
struct Elem<QUANTO_HIP_F16> {
  using T = _Float16;
  static __device__ __forceinline__ float to_f32(T v) { return (float)v; }
  static __device__ __forceinline__ T from_f32(float f) { return (_Float16)f; }  // RNE
};
template <>
struct Elem<QUANTO_HIP_BF16> {
  using T = __bf16;
  static __device__ __forceinline__ float to_f32(T v) { return (float)v; }
  static __device__ __forceinline__ T from_f32(float f) { return (__bf16)f; }  // v_cvt_pk_bf16_f32, RNE
};

// ---- 8-bit weight/activation decoders ---------------------------------------------------------
// e4m3fnuz has no hardware path on gfx950 (the native fp8 is OCP e4m3fn): decode in software.
__device__ __forceinline__ float decode_e4m3fnuz(uint32_t b) {
  b &= 0xFFu;
  if (b == 0x80u) return __builtin_nanf("");
  const uint32_t e = (b >> 3) & 0xFu, m = b & 7u;
  float v = e == 0 ? (float)m * 0x1p-10f : __builtin_bit_cast(float, ((e + 119u) << 23) | (m << 20));
  return (b & 0x80u) ? -v : v;
}

// e4m3fnuz on the OCP converters (r4): an fnuz byte is the e4m3fn value of the same bits times 1/2 (exponent bias 8 instead of 7; the
// subnormals m * 2^-10 = fn's m * 2^-9 halved), except for the three patterns the two formats disagree on: 0x7F / 0xFF are +-240 in fnuz
// (NaN in fn) and 0x80 is fnuz's only NaN (-0 in fn).  The hardware converts with scale 0.5 (a power of two: exact), the three patterns are
// patched - they are rare but real: the absmax element of every row quantizes to +-240.
__device__ __forceinline__ float fnuz_fix_f32(float half_of_fn, uint32_t byte) {
  float v = half_of_fn;
  if ((byte & 0x7Fu) == 0x7Fu) v = (byte & 0x80u) ? -240.f : 240.f;
  if (byte == 0x80u) v = __builtin_nanf("");
  return v;
}
// two T elements (bytes 2p, 2p + 1 of `word`); MAX_T / NAN_T: 240.0 and a quiet NaN in T
template <uint32_t MAX_T, uint32_t NAN_T>
__device__ __forceinline__ uint32_t fnuz_fix_pair(uint32_t converted, uint32_t word, int p) {
  const uint32_t b0 = (word >> (16 * p)) & 0xFFu, b1 = (word >> (16 * p + 8)) & 0xFFu;
  uint32_t lo = converted & 0xFFFFu, hi = converted >> 16;
  if ((b0 & 0x7Fu) == 0x7Fu) lo = MAX_T | ((b0 & 0x80u) << 8);
  if ((b1 & 0x7Fu) == 0x7Fu) hi = MAX_T | ((b1 & 0x80u) << 8);
  if (b0 == 0x80u) lo = NAN_T;
  if (b1 == 0x80u) hi = NAN_T;
  return lo | (hi << 16);
}
__device__ __forceinline__ uint32_t fnuz_pair_bf16(uint32_t word, int p) {
  const uint32_t c = __builtin_bit_cast(uint32_t, p == 0 ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)word, 0.5f, false)
                                                         : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)word, 0.5f, true));
  return fnuz_fix_pair<0x4370u, 0x7FC0u>(c, word, p);
}
__device__ __forceinline__ uint32_t fnuz_pair_f16(uint32_t word, int p) {
  const uint32_t c = __builtin_bit_cast(uint32_t, p == 0 ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)word, 0.5f, false)
                                                         : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)word, 0.5f, true));
  return fnuz_fix_pair<0x5B80u, 0x7E00u>(c, word, p);
}

template <int DT>
__device__ __forceinline__ float decode8(uint8_t b);
template <>
__device__ __forceinline__ float decode8<QUANTO_HIP_I8>(uint8_t b) {
  return (float)(int8_t)b;
}
template <>
__device__ __forceinline__ float decode8<QUANTO_HIP_U8>(uint8_t b) {
  return (float)b;
}
template <>
__device__ __forceinline__ float decode8<QUANTO_HIP_F8_E4M3FN>(uint8_t b) {
  return __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
}
template <>
__device__ __forceinline__ float decode8<QUANTO_HIP_F8_E5M2>(uint8_t b) {
  return __builtin_amdgcn_cvt_f32_bf8((int)b, 0);
}
template <>
__device__ __forceinline__ float decode8<QUANTO_HIP_F8_E4M3FNUZ>(uint8_t b) {
  return decode_e4m3fnuz(b);
}

// ---- wave-level sum (64 lanes) ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Generic PackedTensor geometry for an axis-0 weight [N, K] with group size C (C == K when
// per-channel): grouped rows R = N*K/C, packed rows row_dim = ceil(R / vpi).
struct PackedGeom {
  int64_t N, K, C, G, R, row_dim;
  int bits, vpi;
};
inline PackedGeom make_geom(int64_t N, int64_t K, int bits, int group_size) {
  PackedGeom g;
  g.N = N;
  g.K = K;
  g.bits = bits;
  g.vpi = 8 / bits;
  g.C = group_size > 0 ? group_size : K;
  g.G = K / g.C;
  g.R = N * g.G;
  g.row_dim = (g.R + g.vpi - 1) / g.vpi;
  return g;
}

// Experiment knobs.  The product dispatch reads NO environment variable per call: the switch QUANTO_HIP_EXPERIMENT is read once,
// when the library first needs it; unless it was set to a non-zero value at that moment every knob is its default and env_int /
// env_ptr are a load and a branch.  With the switch on, the knobs are read on every call, so that one process can A/B variants
// (scripts/ab.py flips a variable between hipGraph captures; the GPU tests force the tile configurations the same way).
inline bool experiments_enabled() {
  static const bool on = [] {
    const char* e = getenv("QUANTO_HIP_EXPERIMENT");
    return e != nullptr && atoi(e) != 0;
  }();
  return on;
}
inline int env_int(const char* name, int dflt) {
  if (!experiments_enabled()) return dflt;
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// debugging aid: a device address handed over as text (scripts/skinny_timeline.py), null when unset
inline void* env_ptr(const char* name) {
  if (!experiments_enabled()) return nullptr;
  const char* e = getenv(name);
  return e ? reinterpret_cast<void*>(strtoull(e, nullptr, 0)) : nullptr;
}

int launch_status();  // hipGetLastError() -> quanto_hip_status (defined in c_api.hip)
void set_last_kernel(const char* name);

}  // namespace qh
