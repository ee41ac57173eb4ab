// Dev microbenchmark (GPU box): store patterns for one 366 912-byte item per 256-thread workgroup (32768 items).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr size_t CH4 = 3276 * 14 / 2;   // float4 per item
// A: the estimator's pattern: 252 lanes, workgroup iteration = 4032 contiguous bytes
__global__ void patA(float4* p) {
  float4* q = p + blockIdx.x * CH4; const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  if (threadIdx.x < 252) for (size_t i = threadIdx.x; i < CH4; i += 252) q[i] = v;
}
// A0: pattern A storing zeros (what hipMemset was asked to store)
__global__ void patA0(float4* p) {
  float4* q = p + blockIdx.x * CH4; const float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x < 252) for (size_t i = threadIdx.x; i < CH4; i += 252) q[i] = v;
}
// B: each wave owns a contiguous quarter of the item and walks it 1 KB at a time
__global__ void patB(float4* p) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63; const size_t Q = CH4 / 4;   // 5733 float4 per wave
  float4* q = p + blockIdx.x * CH4 + w * Q; const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  for (size_t i = l; i < Q; i += 64) q[i] = v;
}
// C: each lane writes 64 contiguous bytes per iteration (4 float4), workgroup iteration = 16 KB
__global__ void patC(float4* p) {
  float4* q = p + blockIdx.x * CH4; const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  for (size_t i = 4 * threadIdx.x; i + 3 < CH4; i += 1024) { q[i] = v; q[i + 1] = v; q[i + 2] = v; q[i + 3] = v; }
  // (tail of CH4 % 4 = 0: 22932 = 4 * 5733, exact)
}
// D: as A with 256 lanes and the workgroup iteration unrolled x4 (16 KB in flight per workgroup)
__global__ void patD(float4* p) {
  float4* q = p + blockIdx.x * CH4; const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  size_t i = threadIdx.x;
  for (; i + 768 < CH4; i += 1024) { q[i] = v; q[i + 256] = v; q[i + 512] = v; q[i + 768] = v; }
  for (; i < CH4; i += 256) q[i] = v;
}
// E: two items per workgroup interleaved (halves the number of concurrent streams per CU for the same occupancy)
__global__ void patE(float4* p) {
  float4* q = p + (size_t)blockIdx.x * 2 * CH4; const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  if (threadIdx.x < 252) for (size_t i = threadIdx.x; i < 2 * CH4; i += 252) q[i] = v;
}
// G: grid-stride over the whole buffer (what hipMemset launches with 256 workgroups): the chip sweeps the buffer front to back
__global__ void patG(float4* p, size_t n4) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// H: pattern A with the occupancy limited through dynamic LDS (fewer concurrent item streams per CU)
__global__ void patH(float4* p) {
  extern __shared__ float pad[];
  float4* q = p + blockIdx.x * CH4; const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  if (threadIdx.x == 999) pad[0] = 1.f;
  if (threadIdx.x < 252) for (size_t i = threadIdx.x; i < CH4; i += 252) q[i] = v;
}
// S: the item's 4 KB chunks in a strided order -- chunk (k mod G) * STRIDE + k div G at step k -- so that consecutive steps of one
// stream land STRIDE * 4 KB apart (if the HBM channel interleave has that period, a stream then finishes one channel's share of
// its item -- one DRAM row -- before moving to the next channel instead of revisiting each channel every 64 KB)
template <int STRIDE>
__global__ __launch_bounds__(256, 3) void patS(float4* p) {
  extern __shared__ float pad[];
  float4* q = p + blockIdx.x * CH4; const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  if (threadIdx.x == 999) pad[0] = 1.f;
  constexpr int NCH = (int)((CH4 + 255) / 256), G = (NCH + STRIDE - 1) / STRIDE;
  for (int k = 0; k < G * STRIDE; ++k) {
    const int c = (k % G) * STRIDE + k / G;
    const size_t i = (size_t)c * 256 + threadIdx.x;
    if (c < NCH && i < CH4) q[i] = v;
  }
}
template <typename F> double time_ms(F f, int iters) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); f(); CHECK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}
int main() {
  const int n = 32768; float4* p; CHECK(hipMalloc(&p, n * CH4 * 16)); const double gb = n * CH4 * 16 / 1e9;
  for (int rep = 0; rep < 2; ++rep) {
    double t;
    t = time_ms([&] { CHECK(hipMemsetAsync(p, 0, n * CH4 * 16, 0)); }, 5); printf("hipMemset                       %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { patA<<<n, 256>>>(p); }, 5); printf("A 252 lanes, 4 KB per WG iter    %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { patA0<<<n, 256>>>(p); }, 5); printf("A0 pattern A, all-zero data      %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { CHECK(hipMemsetD32Async((hipDeviceptr_t)p, 0x3F8CCCCD, n * CH4 * 4, 0)); }, 5); printf("hipMemsetD32 of 1.1f             %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    for (int b : {128, 256, 512, 768, 1024, 2048}) {
      t = time_ms([&] { patG<<<b, 256>>>(p, n * CH4); }, 5); printf("G grid-stride, %4d workgroups     %.3f ms %.0f GB/s\n", b, t, gb / t * 1e3);
    }
    for (int kb : {150, 75, 50, 38, 24, 12}) {
      CHECK(hipFuncSetAttribute((const void*)patH, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024));
      t = time_ms([&] { patH<<<n, 256, kb * 1024>>>(p); }, 5); printf("H pattern A, %3d KB LDS (%d WG/CU)  %.3f ms %.0f GB/s\n", kb, 160 / kb > 8 ? 8 : 160 / kb, t, gb / t * 1e3);
    }
#define RUN_S(ST) do { CHECK(hipFuncSetAttribute((const void*)patS<ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 50 * 1024)); \
    t = time_ms([&] { patS<ST><<<n, 256, 50 * 1024>>>(p); }, 5); printf("S chunk order strided by %4d KB (3 WG/CU) %.3f ms %.0f GB/s\n", ST * 4, t, gb / t * 1e3); } while (0)
    RUN_S(1); RUN_S(2); RUN_S(4); RUN_S(8); RUN_S(16); RUN_S(32); RUN_S(45); RUN_S(64);
    t = time_ms([&] { patB<<<n, 256>>>(p); }, 5); printf("B wave-contiguous quarters       %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { patC<<<n, 256>>>(p); }, 5); printf("C 64 B per lane                  %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { patD<<<n, 256>>>(p); }, 5); printf("D 256 lanes x4 unrolled          %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { patE<<<n / 2, 256>>>(p); }, 5); printf("E two items per WG               %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
  }
  return 0;
}
