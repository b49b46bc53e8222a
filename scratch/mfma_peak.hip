// Practical f32-MFMA ceiling on this box: register-only chains of v_mfma_f32_32x32x2_f32, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k_peak(float* out, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
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
    double flops = (double)wgs * 4 * iters * 8 * NACC * (2.0 * 32 * 32 * 2);
    printf("%s wgs=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", name, wgs, iters, ms, flops / ms / 1e9);
  }
  hipFree(d);
}
int main() {
  run<4>(256, 4000, "4acc 1wg/CU");
  run<4>(512, 4000, "4acc 2wg/CU");
  run<2>(512, 8000, "2acc 2wg/CU");
  run<4>(1024, 20000, "4acc 4wg/CU long");
  return 0;
}
