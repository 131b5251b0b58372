// Shared helpers for libarctic_hip.so (error channel, HIP checks, small device utilities).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/arctic_hip.h"

namespace aic {

void set_error(const char* fmt, ...);
bool has_device();

// Optional device timing of one named kernel: when enabled, launch sites bracket that kernel with a
// HIP event pair on the launch stream; aic_profile_read() synchronises the events and sums them.
bool profile_enabled();
void profile_begin(hipStream_t stream);
void profile_end(hipStream_t stream);

#define AIC_HIP_TRY(expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::aic::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                       __LINE__);                                                          \
      return AIC_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

#define AIC_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::aic::set_error(__VA_ARGS__);  \
      return AIC_ERR_INVALID;         \
    }                                 \
  } while (0)

#define AIC_NEED_DEVICE()                                             \
  do {                                                                \
    if (!::aic::has_device()) {                                       \
      ::aic::set_error("no HIP device visible (there is no CPU fallback)"); \
      return AIC_ERR_NO_DEVICE;                                       \
    }                                                                 \
  } while (0)

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return AIC_ERR_HIP;
  }
  return AIC_OK;
}

// ---- bf16 helpers (bit-level; device + host) -------------------------------------------------
__host__ __device__ inline float bf16_to_f32(uint16_t h) {
  union { uint32_t u; float f; } v;
  v.u = static_cast<uint32_t>(h) << 16;
  return v.f;
}
// round-to-nearest-even, NaN kept quiet (matches torch's float->bfloat16)
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): one instruction where the bit arithmetic
  // below is eight — the LSTM cell kernel, VALU-bound on its 64 workgroups, rounds ~25 times per element
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
#endif
  union { uint32_t u; float f; } v;
  v.f = f;
  if ((v.u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((v.u >> 16) | 0x0040u);
  const uint32_t lsb = (v.u >> 16) & 1u;
  v.u += 0x7fffu + lsb;
  return static_cast<uint16_t>(v.u >> 16);
}
__host__ __device__ inline float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// ---- fp16 helpers (bit-level) ------------------------------------------------------------------
__host__ __device__ inline float f16_to_f32(uint16_t h) {
  return static_cast<float>(__builtin_bit_cast(_Float16, h));
}
__host__ __device__ inline uint16_t f32_to_f16(float f) {
  return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f));  // round-to-nearest-even
}
__host__ __device__ inline float round_f16(float f) { return f16_to_f32(f32_to_f16(f)); }

}  // namespace aic
