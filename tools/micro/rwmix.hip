// Dev microbenchmark (GPU box): the memory system's bound for the estimator's ACCESS PATTERN with no estimation work --
// per work item: read two 26 KB DM-RS rows (every other complex64, as the comb-2 pilots are) + 26 KB of pilots shared by 4
// items, then write 366 912 contiguous bytes with 252 lanes of float4; one 256-thread workgroup per item, LDS padded so
// that 3 or 4 workgroups fit a CU.  `dep`: the written value depends on the loaded data (as the estimate does) or not.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/rwmix tools/micro/rwmix.hip && /tmp/rwmix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int N_SC = 3276, N_SYM = 14, N_RE = 1638, ROW4 = 7, ACTIVE = 252;

template <bool DEP, bool READ>
__global__ __launch_bounds__(256) void rwmix(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out,
                                             int n_ports, int xcd_map) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  int item = blockIdx.x;
  if (xcd_map) {  // ports of a slot 8 workgroups apart (same XCD), as the estimator places them
    const int per = 8 * n_ports, g = item / per, j = item - g * per;
    item = (g * 8 + (j & 7)) * n_ports + (j >> 3);
  }
  const int slot = item / n_ports;
  const float2* r = rx + (size_t)item * N_SC * N_SYM;   // [sym][sc] buffer of this item
  float acc = 0.f;
  if (READ) {
    for (int k = tid; k < N_RE; k += 256) {
      const float2 a = r[2 * N_SC + 2 * k], b = r[11 * N_SC + 2 * k];
      const float2 p = pil[(size_t)slot * N_RE * 2 + k], q = pil[(size_t)slot * N_RE * 2 + N_RE + k];
      acc += a.x * p.x + a.y * p.y + b.x * q.x + b.y * q.y;
    }
  }
  float v = 1.f;
  if (DEP) {  // block-wide dependence of every store on every load, like the estimate's
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    v = red[0];
  } else if (READ) {
    if (acc == 123.456f) v = 2.f;   // keeps the loads alive; stores may issue before they return
  }
  if (tid < ACTIVE) {
    float4* o = out + (size_t)item * (N_SC * ROW4) + tid;
    const float4 val = make_float4(v, v + 1.f, v + 2.f, (float)item);
#pragma unroll 4
    for (int s = tid / ROW4; s < N_SC; s += ACTIVE / ROW4) { *o = val; o += ACTIVE; }
  }
}

// The estimator's start-up, grafted onto the pattern above, to see which part of it (if any) explains the few per cent the
// estimator's kernel sits above `rwmix<true, true>`:  PRE = every workgroup first copies 4.4 KB of per-launch tables
// (plan + twiddles: the same addresses for all workgroups) into LDS and reads a few scalars through a pointer;
// EAGER = the item's 28 pilot loads per thread are requested back to back (unrolled, clamped index) instead of in a loop.
struct SkelPlan { int n_re, n_sc, a, b, c, d, e, f; };
template <bool PRE, bool EAGER, int REGS = 0>
__global__ __launch_bounds__(256) void skel(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out,
                                            int n_ports, const float4* __restrict__ tab, const SkelPlan* __restrict__ plan) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  int item = blockIdx.x;
  {
    const int per = 8 * n_ports, g = item / per, j = item - g * per;
    item = (g * 8 + (j & 7)) * n_ports + (j >> 3);
  }
  const int slot = item / n_ports;
  const float2* r = rx + (size_t)item * N_SC * N_SYM;
  float live[REGS > 0 ? REGS : 1];  // REGS > 0: that many VGPRs stay live from the first to the last instruction (footprint experiment)
#pragma unroll
  for (int i = 0; i < REGS; ++i) { live[i] = (float)(tid + i); asm volatile("" : "+v"(live[i])); }
  float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
  int n_re = N_RE;
  if (PRE) {
    if (tid < 276) tv = tab[tid];
    n_re = plan->n_re + (plan->e & 0);
  }
  float acc = 0.f;
  if (EAGER) {
    float2 a[7], b[7], p[7], q[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int k = tid + i * 256, kk = k < n_re ? k : 0;
      a[i] = r[2 * N_SC + 2 * kk]; b[i] = r[11 * N_SC + 2 * kk];
      p[i] = pil[(size_t)slot * N_RE * 2 + kk]; q[i] = pil[(size_t)slot * N_RE * 2 + N_RE + kk];
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (tid + i * 256 < n_re) acc += a[i].x * p[i].x + a[i].y * p[i].y + b[i].x * q[i].x + b[i].y * q[i].y;
  } else {
    for (int k = tid; k < n_re; k += 256) {
      const float2 a = r[2 * N_SC + 2 * k], b = r[11 * N_SC + 2 * k];
      const float2 p = pil[(size_t)slot * N_RE * 2 + k], q = pil[(size_t)slot * N_RE * 2 + N_RE + k];
      acc += a.x * p.x + a.y * p.y + b.x * q.x + b.y * q.y;
    }
  }
  float* lds_tab = red + 256;
  if (PRE && tid < 276) reinterpret_cast<float4*>(lds_tab)[tid] = tv;
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  float v = red[0];
  if (PRE) v += lds_tab[(tid * 7) & 1023] * 1e-30f;
  if (tid < ACTIVE) {
    float4* o = out + (size_t)item * (N_SC * ROW4) + tid;
    const float4 val = make_float4(v, v + 1.f, v + 2.f, (float)item);
#pragma unroll 4
    for (int s = tid / ROW4; s < N_SC; s += ACTIVE / ROW4) { *o = val; o += ACTIVE; }
  }
#pragma unroll
  for (int i = 0; i < REGS; ++i) asm volatile("" : : "v"(live[i]));
}

// Upper bound of a "sweep writer" redesign: persistent workgroups; each alternates between reading + reducing one item's
// pilots and writing one item's worth of output in 4 KB chunks handed out by a global ticket counter, so that the chip's
// stores advance through the output in address order while the reads stay interleaved as in the fused kernel.  (A real
// design would take the written values from a cache-resident workspace and needs generation barriers: not modelled.)
__global__ __launch_bounds__(256) void sweepmix(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out,
                                                int n_items, int n_ports, unsigned long long* ticket, unsigned long long n_chunks, int chunk4) {
  __shared__ float red[256];
  __shared__ unsigned long long first;
  const int tid = threadIdx.x;
  const int per_item = (N_SC * ROW4 + chunk4 - 1) / chunk4;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int slot = item / n_ports;
    const float2* r = rx + (size_t)item * N_SC * N_SYM;
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
    if (tid == 0) first = atomicAdd(ticket, (unsigned long long)per_item);   // this workgroup's run of consecutive chunks
    __syncthreads();
    const unsigned long long c0 = first;
    const float4 val = make_float4(v, v + 1.f, v + 2.f, (float)item);
    for (int c = 0; c < per_item; ++c) {
      const unsigned long long ch = c0 + c;
      if (ch < n_chunks) for (int i = tid; i < chunk4; i += 256) out[ch * chunk4 + i] = val;
    }
    __syncthreads();
  }
}
// the same with the tickets taken one 4 KB chunk at a time (finest interleaving of the chip's stores)
__global__ __launch_bounds__(256) void sweepmix1(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out,
                                                 int n_items, int n_ports, unsigned long long* ticket, unsigned long long n_chunks, int chunk4) {
  __shared__ float red[256];
  __shared__ unsigned long long first;
  const int tid = threadIdx.x;
  const int per_item = (N_SC * ROW4 + chunk4 - 1) / chunk4;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int slot = item / n_ports;
    const float2* r = rx + (size_t)item * N_SC * N_SYM;
    float acc = 0.f;
    for (int k = tid; k < N_RE; k += 256) {
      const float2 a = r[2 * N_SC + 2 * k], b = r[11 * N_SC + 2 * k];
      const float2 p = pil[(size_t)slot * N_RE * 2 + k], q = pil[(size_t)slot * N_RE * 2 + N_RE + k];
      acc += a.x * p.x + a.y * p.y + b.x * q.x + b.y * q.y;
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float4 val = make_float4(red[0], 1.f, 2.f, (float)item);
    for (int c = 0; c < per_item; ++c) {
      __syncthreads();
      if (tid == 0) first = atomicAdd(ticket, 1ull);
      __syncthreads();
      const unsigned long long ch = first;
      if (ch < n_chunks) for (int i = tid; i < chunk4; i += 256) out[ch * chunk4 + i] = val;
    }
    __syncthreads();
  }
}
// Lock-step generations: all resident workgroups read + reduce one item each, meet at a grid barrier, then write the
// generation's contiguous output region grid-stride (chunk j, j + G, j + 2G, ... of 4 KB): a static sweep, no tickets.
// The next generation's reads overlap the tail of this one's stores.  (Written values would come from a cache-resident
// workspace in a real design: not modelled.)
__global__ __launch_bounds__(256) void lockstep(const float2* __restrict__ rx, const float2* __restrict__ pil, float4* __restrict__ out,
                                                int n_items, int n_ports, unsigned* bar) {
  __shared__ float red[256];
  const int tid = threadIdx.x, G = gridDim.x;
  const size_t item4 = (size_t)N_SC * ROW4;
  unsigned gen = 0;
  for (int base = 0; base < n_items; base += G, ++gen) {
    const int item = base + blockIdx.x;
    float v = 0.f;
    if (item < n_items) {
      const int slot = item / n_ports;
      const float2* r = rx + (size_t)item * N_SC * N_SYM;
      float acc = 0.f;
      for (int k = tid; k < N_RE; k += 256) {
        const float2 a = r[2 * N_SC + 2 * k], b = r[11 * N_SC + 2 * k];
        const float2 p = pil[(size_t)slot * N_RE * 2 + k], q = pil[(size_t)slot * N_RE * 2 + N_RE + k];
        acc += a.x * p.x + a.y * p.y + b.x * q.x + b.y * q.y;
      }
      red[tid] = acc;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
      v = red[0];
    }
    // grid barrier: every resident workgroup arrives once per generation
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      atomicAdd(bar, 1u);
      while (atomicAdd(bar, 0u) < (gen + 1) * (unsigned)G) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const int n_here = min(G, n_items - base);
    const size_t region4 = (size_t)n_here * item4;
    float4* o = out + (size_t)base * item4;
    const float4 val = make_float4(v, 1.f, 2.f, (float)item);
    for (size_t c = (size_t)blockIdx.x * 256; c < region4; c += (size_t)G * 256) {
      const size_t i = c + tid;
      if (i < region4) o[i] = val;
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
  const int n_slots = 8192, n_ports = 4, n_items = n_slots * n_ports;
  float2 *rx, *pil; float4* out;
  CHECK(hipMalloc(&rx, (size_t)n_items * N_SC * N_SYM * 8));
  CHECK(hipMalloc(&pil, (size_t)n_slots * N_RE * 2 * 8));
  CHECK(hipMalloc(&out, (size_t)n_items * N_SC * N_SYM * 8));
  CHECK(hipMemset(rx, 0, (size_t)n_items * N_SC * N_SYM * 8));
  CHECK(hipMemset(pil, 0, (size_t)n_slots * N_RE * 2 * 8));
  const double alg = (double)n_slots * 1598688.0;
  for (int wgs : {3, 4, 6}) {
    const int lds = wgs == 3 ? 50 * 1024 : wgs == 4 ? 38 * 1024 : 24 * 1024;   // dynamic LDS that lets exactly `wgs` fit 160 KB
    CHECK(hipFuncSetAttribute((const void*)rwmix<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)rwmix<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)rwmix<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int xm : {1, 0}) {
      double t = time_ms([&] { rwmix<true, true><<<n_items, 256, lds>>>(rx, pil, out, n_ports, xm); }, 5);
      printf("%d WG/CU xcd_map=%d read -> dependent write : %.3f ms  %.0f GB/s algorithmic\n", wgs, xm, t, alg / t / 1e6);
      t = time_ms([&] { rwmix<false, true><<<n_items, 256, lds>>>(rx, pil, out, n_ports, xm); }, 5);
      printf("%d WG/CU xcd_map=%d read || independent write: %.3f ms  %.0f GB/s algorithmic\n", wgs, xm, t, alg / t / 1e6);
    }
    double t = time_ms([&] { rwmix<false, false><<<n_items, 256, lds>>>(rx, pil, out, n_ports, 1); }, 5);
    printf("%d WG/CU write only                           : %.3f ms  %.0f GB/s of stores\n", wgs, t, (double)n_items * 366912 / t / 1e6);
  }
  {
    float4* tab; SkelPlan* plan; SkelPlan hp = {N_RE, N_SC, 0, 0, 0, 0, 0, 0};
    CHECK(hipMalloc(&tab, 276 * 16)); CHECK(hipMemset(tab, 0, 276 * 16));
    CHECK(hipMalloc(&plan, sizeof(SkelPlan))); CHECK(hipMemcpy(plan, &hp, sizeof(hp), hipMemcpyHostToDevice));
    const int lds = 50 * 1024;
    CHECK(hipFuncSetAttribute((const void*)skel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)skel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)skel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)skel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)skel<true, true, 130>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 2; ++rep) {
      double t = time_ms([&] { skel<true, true, 130><<<n_items, 256, lds>>>(rx, pil, out, n_ports, tab, plan); }, 5);
      printf("skeleton 3 WG/CU: eager loads, table copy, +130 live VGPRs: %.3f ms  %.0f GB/s algorithmic\n", t, alg / t / 1e6);
      t = time_ms([&] { skel<false, false><<<n_items, 256, lds>>>(rx, pil, out, n_ports, tab, plan); }, 5);
      printf("skeleton 3 WG/CU: loop loads, no table copy    : %.3f ms  %.0f GB/s algorithmic\n", t, alg / t / 1e6);
      t = time_ms([&] { skel<true, false><<<n_items, 256, lds>>>(rx, pil, out, n_ports, tab, plan); }, 5);
      printf("skeleton 3 WG/CU: loop loads, 4.4 KB table copy: %.3f ms  %.0f GB/s algorithmic\n", t, alg / t / 1e6);
      t = time_ms([&] { skel<false, true><<<n_items, 256, lds>>>(rx, pil, out, n_ports, tab, plan); }, 5);
      printf("skeleton 3 WG/CU: eager loads, no table copy   : %.3f ms  %.0f GB/s algorithmic\n", t, alg / t / 1e6);
      t = time_ms([&] { skel<true, true><<<n_items, 256, lds>>>(rx, pil, out, n_ports, tab, plan); }, 5);
      printf("skeleton 3 WG/CU: eager loads, table copy      : %.3f ms  %.0f GB/s algorithmic\n", t, alg / t / 1e6);
    }
  }
  unsigned* bar; CHECK(hipMalloc(&bar, 4));
  for (int wgs : {1, 2, 3}) {
    double t = time_ms([&] { CHECK(hipMemsetAsync(bar, 0, 4, 0)); lockstep<<<256 * wgs, 256>>>(rx, pil, out, n_items, n_ports, bar); }, 5);
    printf("lock-step generations, static sweep, %d WG/CU: %.3f ms  %.0f GB/s algorithmic\n", wgs, t, alg / t / 1e6);
  }
  unsigned long long* ticket; CHECK(hipMalloc(&ticket, 8));
  for (int chunk4 : {256, 1024}) {   // 4 KB and 16 KB chunks
    const unsigned long long n_chunks = ((unsigned long long)n_items * N_SC * ROW4) / chunk4;
    for (int wgs : {3, 4}) {
      double t = time_ms([&] { CHECK(hipMemsetAsync(ticket, 0, 8, 0)); sweepmix<<<256 * wgs, 256>>>(rx, pil, out, n_items, n_ports, ticket, n_chunks, chunk4); }, 5);
      printf("sweep writer bound, %2d KB chunks, item-sized ticket runs, %d WG/CU: %.3f ms  %.0f GB/s algorithmic\n", chunk4 / 64, wgs, t, alg / t / 1e6);
      t = time_ms([&] { CHECK(hipMemsetAsync(ticket, 0, 8, 0)); sweepmix1<<<256 * wgs, 256>>>(rx, pil, out, n_items, n_ports, ticket, n_chunks, chunk4); }, 5);
      printf("sweep writer bound, %2d KB chunks, one ticket per chunk,    %d WG/CU: %.3f ms  %.0f GB/s algorithmic\n", chunk4 / 64, wgs, t, alg / t / 1e6);
    }
  }
  return 0;
}
