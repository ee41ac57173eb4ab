// Dev micro-benchmark: issue rate of v_mfma_f32_16x16x16_f16 vs v_mfma_f32_16x16x32_f16 on gfx950 (one wave per SIMD,
// operands in registers, independent accumulators).  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K32>
__global__ __launch_bounds__(256) void rate(float* out, int iters, unsigned long long* cyc) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h4 a4 = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)threadIdx.x}, b4 = a4;
  h8 a8 = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)threadIdx.x, (_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2.f}, b8 = a8;
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (K32) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
    }
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc; 
  hipMalloc(&out, 256 * 256 * 4 * sizeof(float)); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int k32 = 0; k32 < 2; ++k32) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (k32) rate<1><<<256 * 4, 256>>>(out, iters, cyc); else rate<0><<<256 * 4, 256>>>(out, iters, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma = (double)iters * 4;
    const double flops = 256.0 * 4 * 4 * n_mfma * 2.0 * 16 * 16 * (k32 ? 32 : 16);
    printf("%s: %.3f ms, %.1f TFLOP/s chip (4 workgroups/CU), %.1f clock64 ticks per MFMA per wave\n", k32 ? "16x16x32_f16" : "16x16x16_f16", ms, flops / ms * 1e-9, (double)c / n_mfma);
  }
  return 0;
}
