// Dev microbenchmark (GPU box): does gating the READS of the estimator's access pattern into chip-wide time windows help?
// Same pattern as rwmix<true, true> (tools/micro/rwmix.hip: per item two 26 KB DM-RS rows + shared pilots in, 366 912 B out, stores
// depend on the loads); before a workgroup issues its loads, it waits until the 100 MHz wall clock is inside a window of every
// period -- so that the memory controllers see the reads of all resident workgroups in bursts instead of sprinkled among the stores.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/rwphase tools/micro/rwphase.hip && /tmp/rwphase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int N_SC = 3276, N_SYM = 14, N_RE = 1638, ROW4 = 7, ACTIVE = 252;

__global__ __launch_bounds__(256) void rwphase(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out,
                                               int n_ports, unsigned period, unsigned window) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  int item = blockIdx.x;
  {
    const int per = 8 * n_ports, g = item / per, j = item - g * per;
    item = (g * 8 + (j & 7)) * n_ports + (j >> 3);
  }
  const int slot = item / n_ports;
  const float2* r = rx + (size_t)item * N_SC * N_SYM;
  if (period) {   // wave-uniform spin (bounded: at most one period)
    unsigned spins = 0;
    while ((unsigned)(wall_clock64() % period) >= window && spins < 100000u) { __builtin_amdgcn_s_sleep(8); ++spins; }
  }
  float acc = 0.f;
  for (int k = tid; k < N_RE; k += 256) {
    const float2 a = r[2 * N_SC + 2 * k], b = r[11 * N_SC + 2 * k];
    const float2 p = pil[(size_t)slot * N_RE * 2 + k], q = pil[(size_t)slot * N_RE * 2 + N_RE + k];
    acc += a.x * p.x + a.y * p.y + b.x * q.x + b.y * q.y;
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const float v = red[0];
  if (tid < ACTIVE) {
    float4* o = out + (size_t)item * (N_SC * ROW4) + tid;
    const float4 val = make_float4(v, v + 1.f, v + 2.f, (float)item);
#pragma unroll 4
    for (int s = tid / ROW4; s < N_SC; s += ACTIVE / ROW4) { *o = val; o += ACTIVE; }
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
  const int n_slots = 8192, n_ports = 4, n_items = n_slots * n_ports;
  float2 *rx, *pil; float4* out;
  CHECK(hipMalloc(&rx, (size_t)n_items * N_SC * N_SYM * 8));
  CHECK(hipMalloc(&pil, (size_t)n_slots * N_RE * 2 * 8));
  CHECK(hipMalloc(&out, (size_t)n_items * N_SC * N_SYM * 8));
  CHECK(hipMemset(rx, 0, (size_t)n_items * N_SC * N_SYM * 8));
  CHECK(hipMemset(pil, 0, (size_t)n_slots * N_RE * 2 * 8));
  const double alg = (double)n_slots * 1598688.0;
  const int lds = 50 * 1024;   // 3 workgroups per CU
  CHECK(hipFuncSetAttribute((const void*)rwphase, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int rep = 0; rep < 2; ++rep) {
    double t = time_ms([&] { rwphase<<<n_items, 256, lds>>>(rx, pil, out, n_ports, 0, 0); }, 5);
    printf("no gating                          : %.3f ms  %.0f GB/s algorithmic\n", t, alg / t / 1e6);
    for (unsigned period : {250u, 500u, 1000u, 2000u, 4000u})       // 100 MHz ticks: 2.5 ... 40 us
      for (unsigned pct : {15u, 25u, 40u}) {
        const unsigned window = period * pct / 100;
        t = time_ms([&] { rwphase<<<n_items, 256, lds>>>(rx, pil, out, n_ports, period, window); }, 5);
        printf("reads gated: period %5.1f us, window %2u %%: %.3f ms  %.0f GB/s algorithmic\n", period / 100.0, pct, t, alg / t / 1e6);
      }
  }
  return 0;
}
