// Practical 16-bit MFMA ceiling on this box: register-only chains of v_mfma_f32_32x32x16_f16 (what k_gemm_h2 issues), no memory traffic;
// prints TFLOP/s of the instruction and, over a long run, what the power-limited clock leaves of the 2.5 PFLOP/s dense peak.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void __launch_bounds__(256) k_peak(float* out, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(blockIdx.x * 1e-3f - j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
void run(int wgs, int iters, const char* name) {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)wgs * 4 * iters * 8 * NACC * (2.0 * 32 * 32 * 16);
    printf("%s wgs=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", name, wgs, iters, ms, flops / ms / 1e9);
  }
  hipFree(d);
}
int main() {
  run<4>(256, 4000, "f16 32x32x16 4acc 1wg/CU");
  run<4>(512, 4000, "f16 32x32x16 4acc 2wg/CU");
  run<4>(1024, 40000, "f16 32x32x16 4acc 4wg/CU long (power-limited clock)");
  return 0;
}
