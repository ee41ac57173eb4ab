// Dev microbenchmark (GPU box): what store bandwidth can a write-only stream reach on this chip?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/fillbench tools/micro/fillbench.hip && /tmp/fillbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// A: grid-stride, all threads of the grid interleaved
template <bool NT_ST> __global__ void fill_gridstride(float4* p, size_t n4) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    if (NT_ST) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, (f32x4*)(p + i)); else p[i] = v;
  }
}
// B: each block owns one contiguous chunk of chunk4 float4 (like one work item), ACT active lanes
template <bool NT_ST> __global__ void fill_chunk(float4* p, size_t chunk4, int act, int unroll_dummy) {
  float4* q = p + blockIdx.x * chunk4;
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  if ((int)threadIdx.x < act)
    for (size_t i = threadIdx.x; i < chunk4; i += act) {
      if (NT_ST) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, (f32x4*)(q + i)); else q[i] = v;
    }
}
// C: persistent blocks looping over chunks (grid = k * 256)
template <bool NT_ST> __global__ void fill_chunk_persist(float4* p, size_t chunk4, int n_chunks, int act) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    float4* q = p + (size_t)c * chunk4;
    if ((int)threadIdx.x < act)
      for (size_t i = threadIdx.x; i < chunk4; i += act) {
        if (NT_ST) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, (f32x4*)(q + i)); else q[i] = v;
      }
  }
}

template <typename F> double time_ms(F f, int iters) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); f();
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main() {
  const size_t chunk4 = 3276 * 14 / 2;           // 22932 float4 = 366912 B
  const int n_chunks = 32768;
  const size_t n4 = chunk4 * n_chunks;
  float4* p; CHECK(hipMalloc(&p, n4 * 16));
  const double gb = n4 * 16 / 1e9;
  printf("buffer %.2f GB\n", gb);
  double t;
  t = time_ms([&] { CHECK(hipMemsetAsync(p, 0, n4 * 16, 0)); }, 5); printf("hipMemset            %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
  for (int blocks : {2048, 4096, 16384}) {
    t = time_ms([&] { fill_gridstride<false><<<blocks, 256>>>(p, n4); }, 5); printf("gridstride b=%-6d   %.3f ms  %.0f GB/s\n", blocks, t, gb / t * 1e3);
    t = time_ms([&] { fill_gridstride<true><<<blocks, 256>>>(p, n4); }, 5);  printf("gridstride b=%-6d nt %.3f ms  %.0f GB/s\n", blocks, t, gb / t * 1e3);
  }
  for (int act : {256, 252}) for (int thr : {256, 512}) {
    int a2 = act * (thr / 256);
    t = time_ms([&] { fill_chunk<false><<<n_chunks, thr>>>(p, chunk4, a2, 0); }, 5); printf("chunk/blk thr=%d act=%d    %.3f ms  %.0f GB/s\n", thr, a2, t, gb / t * 1e3);
    t = time_ms([&] { fill_chunk<true><<<n_chunks, thr>>>(p, chunk4, a2, 0); }, 5);  printf("chunk/blk thr=%d act=%d nt %.3f ms  %.0f GB/s\n", thr, a2, t, gb / t * 1e3);
  }
  for (int k : {2, 4, 8}) {
    t = time_ms([&] { fill_chunk_persist<false><<<256 * k, 256>>>(p, chunk4, n_chunks, 252); }, 5); printf("persist %d/CU act=252     %.3f ms  %.0f GB/s\n", k, t, gb / t * 1e3);
    t = time_ms([&] { fill_chunk_persist<true><<<256 * k, 256>>>(p, chunk4, n_chunks, 252); }, 5);  printf("persist %d/CU act=252 nt  %.3f ms  %.0f GB/s\n", k, t, gb / t * 1e3);
  }
  return 0;
}
