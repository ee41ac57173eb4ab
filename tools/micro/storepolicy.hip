// Dev microbenchmark (GPU box): the estimator's access pattern (tools/micro/rwmix.hip: per item two comb-2 DM-RS rows + shared pilots in, 366 912
// contiguous bytes out through 252 lanes of float4, one workgroup per item, XCD map, 3 workgroups per CU) with the STORES issued under each cache
// policy the ISA offers: default, nt, sc0, sc1, sc0 sc1, sc0 sc1 nt (global_store_dwordx4 ... <bits>, CDNA3/4 ISA: scope and non-temporal hints).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/storepolicy tools/micro/storepolicy.hip && /tmp/storepolicy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int N_SC = 3276, N_SYM = 14, N_RE = 1638, ROW4 = 7, ACTIVE = 252;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int POL> __device__ __forceinline__ void st(float4* p, float4 v) {
  f32x4 x = {v.x, v.y, v.z, v.w};
  if (POL == 0) *p = v;
  else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(x) : "memory");
  else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(x) : "memory");
  else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
  else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(x) : "memory");
}

template <int POL, bool READ, bool NTLOAD = false>
__global__ __launch_bounds__(256) void k(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out, int n_ports) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  int item = blockIdx.x;
  { const int per = 8 * n_ports, g = item / per, j = item - g * per; item = (g * 8 + (j & 7)) * n_ports + (j >> 3); }
  const int slot = item / n_ports;
  const float2* r = rx + (size_t)item * N_SC * N_SYM;
  float acc = 0.f;
  if (READ)
    for (int kk = tid; kk < N_RE; kk += 256) {
      float2 a, b;
      if (NTLOAD) {   // the DM-RS rows are read once: non-temporal loads (global_load ... nt)
        const double da = __builtin_nontemporal_load(reinterpret_cast<const double*>(r + 2 * N_SC + 2 * kk)), db = __builtin_nontemporal_load(reinterpret_cast<const double*>(r + 11 * N_SC + 2 * kk));
        __builtin_memcpy(&a, &da, 8); __builtin_memcpy(&b, &db, 8);
      } else { a = r[2 * N_SC + 2 * kk]; b = r[11 * N_SC + 2 * kk]; }
      const float2 p = pil[(size_t)slot * N_RE * 2 + kk], q = pil[(size_t)slot * N_RE * 2 + N_RE + kk];
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
    for (int s = tid / ROW4; s < N_SC; s += ACTIVE / ROW4) { st<POL>(o, val); o += ACTIVE; }
  }
}

// The same traffic with ALIGNED stores: 256 lanes x 16 bytes = 4096 bytes per iteration, the item's stride padded to a multiple of 4096 bytes (the
// estimator's 252-lane iterations of 4032 bytes and its 366 912-byte items leave half of all wave stores 64 bytes off a 128-byte line).
template <bool READ, bool PAD>
__global__ __launch_bounds__(256) void ka(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out, int n_ports) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  int item = blockIdx.x;
  { const int per = 8 * n_ports, g = item / per, j = item - g * per; item = (g * 8 + (j & 7)) * n_ports + (j >> 3); }
  const int slot = item / n_ports;
  const float2* r = rx + (size_t)item * N_SC * N_SYM;
  float acc = 0.f;
  if (READ)
    for (int kk = tid; kk < N_RE; kk += 256) {
      const float2 a = r[2 * N_SC + 2 * kk], b = r[11 * N_SC + 2 * kk];
      const float2 p = pil[(size_t)slot * N_RE * 2 + kk], q = pil[(size_t)slot * N_RE * 2 + N_RE + kk];
      acc += a.x * p.x + a.y * p.y + b.x * q.x + b.y * q.y;
    }
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const float v = red[0];
  constexpr int N4 = N_SC * ROW4, STRIDE4 = PAD ? ((N4 + 255) / 256) * 256 : N4;
  float4* o = out + (size_t)item * STRIDE4 + tid;
  const float4 val = make_float4(v, v + 1.f, v + 2.f, (float)item);
#pragma unroll 4
  for (int f = tid; f < N4; f += 256) { *o = val; o += 256; }
}
template <bool READ, bool PAD> float run_a(const float2* rx, const float2* pil, float4* out, int items, int lds) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)ka<READ, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int w = 0; w < 3; ++w) ka<READ, PAD><<<items, 256, lds>>>(rx, pil, out, 4);
  CHECK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) ka<READ, PAD><<<items, 256, lds>>>(rx, pil, out, 4);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; best = ms < best ? ms : best;
  }
  return best;
}

template <int POL, bool READ, bool NTLOAD = false> float run(const float2* rx, const float2* pil, float4* out, int items, int lds) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)k<POL, READ, NTLOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int w = 0; w < 3; ++w) k<POL, READ, NTLOAD><<<items, 256, lds>>>(rx, pil, out, 4);
  CHECK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) k<POL, READ, NTLOAD><<<items, 256, lds>>>(rx, pil, out, 4);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; best = ms < best ? ms : best; sum += ms;
  }
  (void)sum;
  return best;
}

int main() {
  const int slots = 8192, ports = 4, items = slots * ports;
  float2 *rx, *pil; float4* out;
  CHECK(hipMalloc(&rx, (size_t)items * N_SC * N_SYM * 8)); CHECK(hipMalloc(&pil, (size_t)slots * N_RE * 2 * 8)); CHECK(hipMalloc(&out, (size_t)items * (((N_SC * ROW4 + 255) / 256) * 256) * 16));
  CHECK(hipMemset(rx, 0, (size_t)items * N_SC * N_SYM * 8)); CHECK(hipMemset(pil, 0, (size_t)slots * N_RE * 2 * 8));
  const int lds = 160 * 1024 / 3 - 2048;   // 3 workgroups per CU
  const char* names[6] = {"default", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"};
  float rw[6], wo[6];
  for (int round = 0; round < 2; ++round) {   // twice, in alternating order, best of both
    float a[6], b[6];
    a[0] = run<0, true>(rx, pil, out, items, lds); a[1] = run<1, true>(rx, pil, out, items, lds); a[2] = run<2, true>(rx, pil, out, items, lds);
    a[3] = run<3, true>(rx, pil, out, items, lds); a[4] = run<4, true>(rx, pil, out, items, lds); a[5] = run<5, true>(rx, pil, out, items, lds);
    b[5] = run<5, false>(rx, pil, out, items, lds); b[4] = run<4, false>(rx, pil, out, items, lds); b[3] = run<3, false>(rx, pil, out, items, lds);
    b[2] = run<2, false>(rx, pil, out, items, lds); b[1] = run<1, false>(rx, pil, out, items, lds); b[0] = run<0, false>(rx, pil, out, items, lds);
    for (int i = 0; i < 6; ++i) { rw[i] = round == 0 || a[i] < rw[i] ? a[i] : rw[i]; wo[i] = round == 0 || b[i] < wo[i] ? b[i] : wo[i]; }
  }
  {
    float t[6];
    for (int i = 0; i < 6; ++i) t[i] = 1e9f;
    for (int round = 0; round < 3; ++round) {
      const float x[6] = {run<0, true>(rx, pil, out, items, lds), run_a<true, false>(rx, pil, out, items, lds), run_a<true, true>(rx, pil, out, items, lds),
                          run<0, false>(rx, pil, out, items, lds), run_a<false, false>(rx, pil, out, items, lds), run_a<false, true>(rx, pil, out, items, lds)};
      for (int i = 0; i < 6; ++i) t[i] = x[i] < t[i] ? x[i] : t[i];
    }
    printf("252 lanes (4032 B / iteration) | 256 lanes (4096 B), items dense | 256 lanes, item stride padded to 4096 B:\n  read -> dependent write %.3f | %.3f | %.3f ms   write only %.3f | %.3f | %.3f ms\n",
           t[0], t[1], t[2], t[3], t[4], t[5]);
  }
  {
    float d = 1e9f, n = 1e9f;
    for (int round = 0; round < 3; ++round) { const float x = run<0, true, false>(rx, pil, out, items, lds), y = run<0, true, true>(rx, pil, out, items, lds); d = x < d ? x : d; n = y < n ? y : n; }
    printf("loads of the DM-RS rows default / nt (default stores): read -> dependent write %.3f / %.3f ms\n", d, n);
  }
  for (int i = 0; i < 6; ++i) printf("stores %-11s: read -> dependent write %.3f ms   write only %.3f ms\n", names[i], rw[i], wo[i]);
  return 0;
}
