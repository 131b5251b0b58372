// Experiment (r04): issue cost of the candidate fp8 -> bf16 expansions on gfx950, one / two / three waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/exp/instr_rate.hip -o gpurun_out/instr_rate && ./gpurun_out/instr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) {          // v_add_f32 (calibration: 4 cycles alone)
        a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, a[i]) + 1.0f);
      } else if (KIND == 1) {   // v_cvt_scalef32_pk_bf16_fp8 (2 fp8 -> 2 bf16)
        bf16x2_t r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(static_cast<int>(a[i]), 1.0f, false);
        a[i] ^= __builtin_bit_cast(uint32_t, r);
      } else if (KIND == 2) {   // v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32
        f32x2_t f = __builtin_amdgcn_cvt_pk_f32_fp8(static_cast<int>(a[i]), false);
        bf16x2_t r = {static_cast<__bf16>(f[0]), static_cast<__bf16>(f[1])};
        a[i] ^= __builtin_bit_cast(uint32_t, r);
      } else if (KIND == 3) {   // integer expansion: perm, shift, and, and_or, add (normal numbers)
        const uint32_t t = __builtin_amdgcn_perm(a[i], 0u, (i & 1) ? 0x070c060cu : 0x050c040cu);   // b1 << 24 | b0 << 8
        const uint32_t mag = (t >> 4) & 0x07f007f0u;
        const uint32_t r = ((t & 0x80008000u) | mag) + 0x3c003c00u;
        a[i] ^= r;
      } else if (KIND == 4) {   // v_exp_f32
        a[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_exp2f(__builtin_bit_cast(float, a[i])));
      } else if (KIND == 5) {   // xor only (what the cvt kinds add on top)
        a[i] ^= a[(i + 1) & 15] >> 3;
      }
    }
  }
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) x ^= a[i];
  if (x == 0x12345678u) out[threadIdx.x] = x;
}

template <int KIND>
double run(int wgs_per_cu, uint32_t* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, 16, 3u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters, 3u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (static_cast<double>(iters) * 16);   // ns per instruction group per wave
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 4096);
  const int iters = 200000;
  const char* names[] = {"v_add_f32", "cvt_scalef32_pk_bf16_fp8 + xor", "cvt_pk_f32_fp8 + cvt_pk_bf16_f32 + xor",
                         "perm/shr/and/and_or/add + xor", "v_exp_f32", "shift + xor"};
  for (int w = 1; w <= 3; ++w) {
    double t[6] = {run<0>(w, d, iters), run<1>(w, d, iters), run<2>(w, d, iters), run<3>(w, d, iters), run<4>(w, d, iters), run<5>(w, d, iters)};
    for (int i = 0; i < 6; ++i)
      printf("%d wave(s)/SIMD  %-42s %7.3f ns per group per wave  (= %5.2f x v_add_f32)\n", w, names[i], t[i], t[i] / t[0]);
  }
  return 0;
}
