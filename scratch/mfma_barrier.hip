// How much does one s_barrier per N MFMAs cost by itself?  Register-only MFMA chains (as scratch/mfma_peak.hip) with a
// workgroup barrier every NB MFMAs, 4 waves per workgroup, W workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NB, bool LDSR>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ float4 sm[2048];
  f16v acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  if (LDSR) for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = make_float4(a, b, a, b);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    float4 fa = make_float4(a, a, a, a), fb = make_float4(b, b, b, b);
    if (LDSR) {               // 16 fragment reads per 64 MFMAs, like the conv kernel
      float4 t = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < (NB * 16) / 64; ++r) { const float4 v = sm[(threadIdx.x + r * 64 + it) & 2047]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      fa = t; fb = t;
    }
#pragma unroll
    for (int r = 0; r < NB / 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(r & 1 ? fa.x : fa.y, r & 2 ? fb.x : fb.y, acc[i], 0, 0, 0);
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}
template <int NB, bool LDSR>
void run(int wgs, int iters, const char* name) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NB, LDSR>), dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double flops = (double)wgs * 4 * iters * NB * (2.0 * 32 * 32 * 2);
  printf("%-44s wgs=%d  %.3f ms  %.1f TFLOP/s\n", name, wgs, best, flops / best / 1e9);
  hipFree(d);
}
int main() {
  run<64, false>(512, 2000, "barrier / 64 MFMAs, 2 wg/CU");
  run<32, false>(512, 4000, "barrier / 32 MFMAs, 2 wg/CU");
  run<128, false>(512, 1000, "barrier / 128 MFMAs, 2 wg/CU");
  run<64, false>(256, 2000, "barrier / 64 MFMAs, 1 wg/CU");
  run<64, false>(768, 2000, "barrier / 64 MFMAs, 3 wg/CU");
  run<64, true>(512, 2000, "barrier + 16 LDS reads / 64 MFMAs, 2 wg/CU");
  run<64, true>(768, 2000, "barrier + 16 LDS reads / 64 MFMAs, 3 wg/CU");
  return 0;
}
