// EXTENSION (no reference counterpart, see include/ce_denoise.h): 3-layer residual Conv2d denoiser over the
// (subcarrier, symbol) plane of the channel estimate, fp16 operands on v_mfma_f32_16x16x32_f16.
//
// One 256-thread workgroup owns one (item, layer) plane and walks it in strips of DN_T subcarriers; the three layers of a
// strip run back to back out of LDS (activations [row][16 columns][16 channels] fp16, columns 0 and 15 = the zero padding
// of symbols -1 and 14), so HBM sees the plane once in and once out.  Implicit GEMM, D = W X:
//   A = weights   [c_out 16][K 32]   lane l: row l&15, k = 8(l>>4)+j   (registers, packed on the host)
//   B = activations [K 32][16 columns of one subcarrier row], K = 2 taps x 16 input channels:
//       lane l: column l&15, tap 2m + (l>>5), channels 8((l>>4)&1) .. +7  -> one ds_read_b128 per MFMA
//   D lane l: column l&15, c_out 4(l>>4)+i -> one ds_write_b64 per tile.
// Layer 2 (16 -> 16 channels, the bulk of the work): a wave owns a BAND of consecutive output rows and pairs the taps so
// that a B fragment belongs to ONE input row (or two adjacent ones) and serves every output row that touches it:
//   A_i = (row i, dx=-1 | row i, dx=0)  used by output rows i, i-1, i-2 (as ky = 0, 1, 2)
//   C_i = (row i, dx=+1 | row i+1, dx=+1)  used by output rows i (ky = 0, 1) and i-2 (ky = 2; upper half: zero weights)
// 5 MFMAs per output row as before, but 2 LDS fragment reads instead of 5 (round 1: one read per MFMA = the LDS
// array's full rate, the kernel's bound).
// Layer 3 has only 2 output channels, so its 16 MFMA rows hold (4 subcarrier rows x re|im, 8 rows used) and K runs over
// the 6 input rows x 3 symbol taps x 16 channels that a block of 4 output rows sees (9 K-steps; the weight matrix is
// banded: a row's 3 x 3 taps, zeros elsewhere) -- 9 MFMAs and 9 LDS reads per 4 rows instead of 20.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "ce_denoise.h"
#include "ce_plan.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ce_denoiser {
  int device = 0;
  half8* wfrag = nullptr;  // [15][64]: layer 1 (1 K-step), layer 2 (5), layer 3 (9, banded over a 4-row block)
  float* bias = nullptr;   // [3][16]
};

#ifndef DN_ABLATE
#define DN_ABLATE 0   // timing experiments only: 1 no global stores, 2 no global loads
#endif

namespace {

#ifndef DN_MIN_WAVES
#define DN_MIN_WAVES 3   // workgroups per CU the register allocator leaves room for (41 KB of LDS allow 3)
#endif
#ifndef DN_TWO_BODIES
#define DN_TWO_BODIES 0   // 1: a second copy of the strip body without row tests for interior strips
#endif
#ifndef DN_L3_BUF
#define DN_L3_BUF 2   // layer 3: fragment sets in flight (2 = the next block's requested before this block's MFMA chain: 36 more VGPRs; 4.53 -> 4.40 ms)
#endif
#ifndef DN_T_ROWS
#define DN_T_ROWS 32
#endif
constexpr int DN_T = DN_T_ROWS;           // output subcarriers per strip (a multiple of 16: each wave owns blocks of 4 rows)
constexpr int DN_NB3 = DN_T / 16;         // layer-3 blocks per wave
static_assert(DN_T % 16 == 0, "strip height");
constexpr int DN_NT = 256, DN_COLS = 16, DN_C = CE_DN_CHANNELS;
constexpr int X0_PIX = (DN_T + 6) * DN_COLS + 2;  // + one pad pixel in front and behind
constexpr int X1_PIX = (DN_T + 5) * DN_COLS + 2;  // + one row that only zero-weight lanes of the last C fragment read
constexpr int X2_PIX = (DN_T + 2) * DN_COLS + 2;

__device__ __forceinline__ int tap_off(int tap) {  // pixel offset of tap t = 3 (dy+1) + (dx+1) in the [row][16] image
  const int ky = tap / 3, kx = tap - 3 * ky;
  return (ky - 1) * DN_COLS + (kx - 1);
}

constexpr int ROW_BYTES16 = DN_COLS * DN_C * 2;  // one image row of a 16-channel layer: 512 B
constexpr int ROW_BYTES0 = DN_COLS * 4;          // one image row of x0 (fp16 re, im): 64 B

// fp16 + ReLU of one D fragment (c_out 4g .. 4g+3 of one pixel): round first, then a packed max -- rounding to
// nearest is monotonic and keeps the sign, so this equals fp16(ReLU(x)); 2 converts + 2 packed max instead of 8 + 2
typedef _Float16 half2x __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 masked(float2 v, bool ok) {  // v, or +0 where !ok, without a branch
  const unsigned m = ok ? 0xFFFFFFFFu : 0u;
  return make_float2(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m));
}
// `keep` = all ones, or 0 on the lanes of the two padding columns: their pixels must stay zero, and an AND is cheaper than
// an exec-masked store (s_and_saveexec + branch + restore around every tile's ds_write)
__device__ __forceinline__ half4 relu_h4(f32x4 acc, unsigned keep) {
  half2x lo = half2x{(_Float16)acc[0], (_Float16)acc[1]}, hi = half2x{(_Float16)acc[2], (_Float16)acc[3]};
  const half2x z = half2x{0, 0};
  lo = __builtin_elementwise_max(lo, z);
  hi = __builtin_elementwise_max(hi, z);
  unsigned ulo, uhi;
  __builtin_memcpy(&ulo, &lo, 4);
  __builtin_memcpy(&uhi, &hi, 4);
  ulo &= keep;
  uhi &= keep;
  __builtin_memcpy(&lo, &ulo, 4);
  __builtin_memcpy(&hi, &uhi, 4);
  return half4{lo[0], lo[1], hi[0], hi[1]};
}

// The kernel is bound by vector-instruction issue (first version: 11.6 vector instructions per MFMA), not by the matrix
// cores or the LDS, unless the per-tile bookkeeping is scalar or constant: every LDS address below is a per-lane constant plus a compile-time tile offset (the tile loops are fully
// unrolled: tile k of a wave is image row wave + 4k), row validity is wave-uniform (scalar branch), the padding columns
// are never written (they stay zero from the initial clear), and global offsets are 32-bit from a per-strip scalar base.
__global__ __launch_bounds__(DN_NT, DN_MIN_WAVES) void ce_denoise_kernel(float2* __restrict__ ch, const half8* __restrict__ wfrag,
                                                            const float* __restrict__ bias, int n_sc, int L) {
  __shared__ __attribute__((aligned(16))) half2v x0[2][X0_PIX];
  __shared__ __attribute__((aligned(16))) _Float16 x1[X1_PIX * DN_C];
  __shared__ __attribute__((aligned(16))) _Float16 x2[X2_PIX * DN_C];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int64_t plane = blockIdx.x;
  const int64_t item = plane / L;
  const int l = (int)(plane - item * L);
  float2* base = ch + item * (int64_t)n_sc * CE_DN_SYMBOLS * L + l;

  half8 w1 = wfrag[lane], w2[5], w3[9];
#pragma unroll
  for (int m = 0; m < 5; ++m) w2[m] = wfrag[(1 + m) * 64 + lane];
#pragma unroll
  for (int m = 0; m < 9; ++m) w3[m] = wfrag[(6 + m) * 64 + lane];
  f32x4 b1, b2, b3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b1[i] = bias[4 * g + i];
    b2[i] = bias[16 + 4 * g + i];
    b3[i] = g < 2 ? bias[32 + (i & 1)] : 0.f;  // layer-3 D rows: (subcarrier row 2g + (i >> 1), re | im)
  }
  // pad pixels and padding columns are read but never written: zero the images once
  for (int i = tid; i < 2 * X0_PIX; i += DN_NT) (&x0[0][0])[i] = half2v{0, 0};
  for (int i = tid; i < X1_PIX * DN_C; i += DN_NT) x1[i] = 0;
  for (int i = tid; i < X2_PIX * DN_C; i += DN_NT) x2[i] = 0;

  // ---- per-lane constants
  const bool col_ok = n >= 1 && n <= CE_DN_SYMBOLS;
  const unsigned keep = col_ok ? 0xFFFFFFFFu : 0u;
  // layer 1 reads: taps 4g .. 4g+3 of x0 at image row wave + 1 (+ 4k rows per tile)
  int rd0[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tap = 4 * g + q;
    rd0[q] = (1 + (wave + 1) * DN_COLS + n + tap_off(tap < 9 ? tap : 4)) * 4;
  }
  // The two 16-byte channel halves of pixel p are stored swapped when bit 2 of p is set: the b128 fragment reads stay
  // conflict-free and the D-fragment ds_write_b64 (16 lanes, 32-byte stride) drops from a 4-way to a 2-way bank conflict.
  // Rows advance by 16 pixels, so the swap is a per-lane constant.
  // layer 2: the wave's band of x2 image rows starts at row a2; fragment A of x1 row a2: pixel column n - 1 (lanes g < 2)
  // or n (g >= 2); fragment C: column n + 1 of row a2 (g < 2) or a2 + 1 (g >= 2); channels 8 (g & 1) .. +7
  constexpr int R2 = (DN_T + 2 + 3) / 4;   // output rows per wave; the last wave's band is moved up so that it ends with the image
  const int a2 = wave * R2 < DN_T + 2 - R2 ? wave * R2 : DN_T + 2 - R2;  // (it recomputes two of its neighbour's rows: same values, no branches in the loop)
  int rdA, rdC, wr2;
  {
    const int pa = 1 + a2 * DN_COLS + n - 1 + (g >> 1), pc = 1 + (a2 + (g >> 1)) * DN_COLS + n + 1, pw = 1 + a2 * DN_COLS + n;
    rdA = pa * (DN_C * 2) + (((g & 1) ^ ((pa >> 2) & 1)) * 16);
    rdC = pc * (DN_C * 2) + (((g & 1) ^ ((pc >> 2) & 1)) * 16);
    wr2 = pw * (DN_C * 2) + ((((g >> 1) & 1) ^ ((pw >> 2) & 1)) * 16) + (g & 1) * 8;  // D fragment of x2 row a2: channels 4g .. 4g+3
  }
  const int wpix = 1 + wave * DN_COLS + n;  // D fragment of image row wave: channels 4g .. 4g+3
  const int wr = wpix * (DN_C * 2) + ((((g >> 1) & 1) ^ ((wpix >> 2) & 1)) * 16) + (g & 1) * 8;
  // global offsets (complex elements) from the strip's first grid row r0
  constexpr int ST_N = ((DN_T + 6) * DN_COLS + DN_NT - 1) / DN_NT;
  int st_off[ST_N], st_row[ST_N];
  bool st_ok[ST_N];
#pragma unroll
  for (int k = 0; k < ST_N; ++k) {
    const int i = tid + k * DN_NT, sym = (i & 15) - 1;
    st_row[k] = (i >> 4) - 3;
    st_ok[k] = i < (DN_T + 6) * DN_COLS && sym >= 0 && sym < CE_DN_SYMBOLS;
    st_off[k] = (st_row[k] * CE_DN_SYMBOLS + sym) * L;
  }
  // layer 3: K-step j of a block reads input row j/3 + 3 (g >> 1), symbol tap j % 3, channels 8 (g & 1) .. +7; the block
  // of a wave starts at x2 image row 4 wave (+ 16 for its second block).  One base per symbol tap (the half swap depends
  // on it), rows are compile-time offsets.
  int rd3[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int pix = 1 + (4 * wave + 3 * (g >> 1)) * DN_COLS + n + dx - 1;
    rd3[dx] = pix * (DN_C * 2) + (((g & 1) ^ ((pix >> 2) & 1)) * 16);
  }
  // lanes g = 0, 1 hold the block's rows 2g and 2g + 1 (re, im each) of column n
  const int h_row = 4 * wave + 2 * g;                           // strip-relative subcarrier row of acc[0..1]
  const int h_off = (h_row * CE_DN_SYMBOLS + (n - 1)) * L;
  const bool h_lane = g < 2 && col_ok;

  float2 st[ST_N];
  auto stage_load = [&](int r0) {  // rows r0-3 .. r0+T+2 -> registers (zeros outside the grid)
    const float2* sb = base + (int64_t)r0 * CE_DN_SYMBOLS * L;
#pragma unroll
    for (int k = 0; k < ST_N; ++k) {
      // unconditional load from an in-range address + AND with the lane's validity mask: left as `if (ok) load`, every
      // load sat inside four nested exec-mask regions with the zero fill repeated in each (25 instructions per load)
      const bool ok = st_ok[k] && (unsigned)(r0 + st_row[k]) < (unsigned)n_sc;
      st[k] = masked((DN_ABLATE & 2) ? make_float2(0.f, 0.f) : sb[ok ? st_off[k] : 0], ok);
    }
  };
  auto stage_load_in = [&](int r0) {  // the same when every row is known to be inside the grid
    const float2* sb = base + (int64_t)r0 * CE_DN_SYMBOLS * L;
#pragma unroll
    for (int k = 0; k < ST_N; ++k) st[k] = masked((DN_ABLATE & 2) ? make_float2(0.f, 0.f) : sb[st_ok[k] ? st_off[k] : 0], st_ok[k]);
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < ST_N; ++k) {
      const int i = tid + k * DN_NT;
      if (i < (DN_T + 6) * DN_COLS) x0[buf][1 + i] = half2v{(_Float16)st[k].x, (_Float16)st[k].y};
    }
  };
  stage_load(0);
  __syncthreads();  // the clear above
  stage_store(0);
  __syncthreads();
  const int n_strips = (n_sc + DN_T - 1) / DN_T;
  // EDGE strips (rows above / below the grid in reach: the first and the last one or two) test every tile's row; interior
  // strips -- all but 2-3 of a 273-PRB plane's 103 -- carry no row tests at all (they were a third of the loop's instructions)
  auto strip = [&](auto edge_tag, int s) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    const int r0 = s * DN_T, buf = s & 1;
    float2* sb = base + (int64_t)r0 * CE_DN_SYMBOLS * L;
    // The next strip's input rows overlap this strip's output rows: request them (and this strip's float32 residuals)
    // now, before layer 3 overwrites them; they are consumed after layer 1 / in layer 3.
    if (s + 1 < n_strips) {
      if (!EDGE && r0 + 2 * DN_T + 3 <= n_sc) stage_load_in(r0 + DN_T); else stage_load(r0 + DN_T);  // rows r0+T-3 .. r0+2T+2
    }
    float2 hres[DN_NB3][2];  // [block][row 2g + e]: float32 residuals of this lane's outputs
#pragma unroll
    for (int b = 0; b < DN_NB3; ++b)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool ok = h_lane && (!EDGE || r0 + h_row + 16 * b + e < n_sc);
        hres[b][e] = masked((DN_ABLATE & 2) ? make_float2(0.f, 0.f) : sb[ok ? h_off + (16 * b + e) * CE_DN_SYMBOLS * L : 0], ok);
      }
    // ---- layer 1: x0 -> x1 image rows 0 .. T+3 (image row t <-> grid row r0-2+t); K = (tap, re | im)
    {
      const char* xin = reinterpret_cast<const char*>(&x0[buf][0]);
      char* xout = reinterpret_cast<char*>(x1);
      constexpr int NK1 = (DN_T + 4) / 4;
      auto finish1 = [&](f32x4 acc, int k) __attribute__((always_inline)) {
        const int row = r0 - 2 + wave + 4 * k;  // wave-uniform: outside the grid the next layer must see zero padding
        const unsigned rowkeep = (!EDGE || (row >= 0 && row < n_sc)) ? keep : 0u;   // scalar select, one AND: no branch, no zero fill
        *reinterpret_cast<half4*>(xout + wr + k * 4 * ROW_BYTES16) = relu_h4(acc, rowkeep);
      };
      f32x4 accp = b1;
      half2v px[2][4];  // tile k+1's pixels are requested before tile k's MFMA (same fence as in layer 2)
#pragma unroll
      for (int q = 0; q < 4; ++q) px[0][q] = *reinterpret_cast<const half2v*>(xin + rd0[q]);
#pragma unroll
      for (int k = 0; k < NK1; ++k) {
        if (k + 1 < NK1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) px[(k + 1) & 1][q] = *reinterpret_cast<const half2v*>(xin + rd0[q] + (k + 1) * 4 * ROW_BYTES0);
        }
        asm volatile("" ::: "memory");
        half8 b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // taps >= 9 (K padding) have zero weights: whatever finite pixel their lanes read does not matter
          b[2 * q] = px[k & 1][q][0];
          b[2 * q + 1] = px[k & 1][q][1];
        }
        const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, b, b1, 0, 0, 0);
        if (k > 0) finish1(accp, k - 1);  // the previous tile's epilogue, behind this tile's MFMA (see layer 2)
        accp = acc;
      }
      finish1(accp, NK1 - 1);
    }
    __syncthreads();
    if (s + 1 < n_strips) stage_store(buf ^ 1);  // x0[buf ^ 1] was last read by layer 1 of the previous strip
    // ---- layer 2: x1 -> x2 image rows 0 .. T+1 (image row t <-> grid row r0-1+t; x2 row t sees x1 rows t, t+1, t+2).
    // The wave walks its band row by row; the fragments of input row t+3 are requested before row t's MFMA chain (the
    // compiler fence keeps the scheduler from sinking them back next to their uses).
    {
      const char* xin = reinterpret_cast<const char*>(x1);
      char* xout = reinterpret_cast<char*>(x2);
      half8 fa[R2 + 2], fc[R2 + 2];
      auto fetch = [&](int i) __attribute__((always_inline)) {  // fragments of x1 row a2 + i
        fa[i] = *reinterpret_cast<const half8*>(xin + rdA + i * ROW_BYTES16);
        fc[i] = *reinterpret_cast<const half8*>(xin + rdC + i * ROW_BYTES16);
      };
      fetch(0);
      fetch(1);
      fetch(2);
      // Row j's conversion + ReLU + store is issued AFTER row j+1's MFMA chain: it then runs in the vector-issue slots the
      // chain leaves free instead of stalling the wave on the chain's result before the next chain may start.
      auto finish2 = [&](f32x4 acc, int j) __attribute__((always_inline)) {
        const int row = r0 - 1 + a2 + j;  // wave-uniform: outside the grid the next layer must see zero padding
        const unsigned rowkeep = (!EDGE || (row >= 0 && row < n_sc)) ? keep : 0u;
        *reinterpret_cast<half4*>(xout + wr2 + j * ROW_BYTES16) = relu_h4(acc, rowkeep);
      };
      f32x4 accp = b2;
#pragma unroll
      for (int j = 0; j < R2; ++j) {
        if (j + 3 < R2 + 2) fetch(j + 3);  // (needed by row j + 1 on)
        asm volatile("" ::: "memory");
        f32x4 acc = b2;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[0], fa[j], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[1], fa[j + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[2], fa[j + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[3], fc[j], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[4], fc[j + 2], acc, 0, 0, 0);
        if (j > 0) finish2(accp, j - 1);
        accp = acc;
      }
      finish2(accp, R2 - 1);
    }
    __syncthreads();
    // ---- layer 3 + residual: x2 -> grid rows r0 .. r0+T-1, T/16 blocks of 4 rows per wave (rows 4 wave + 16 b ..)
    {
      const char* xin = reinterpret_cast<const char*>(x2);
      auto finish3 = [&](f32x4 acc, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if ((!(DN_ABLATE & 1) || acc[0] == 123.456f) && h_lane && (!EDGE || r0 + h_row + 16 * b + e < n_sc))
            sb[h_off + (16 * b + e) * CE_DN_SYMBOLS * L] = make_float2(hres[b][e].x + acc[2 * e], hres[b][e].y + acc[2 * e + 1]);
      };
      f32x4 accp = b3;
      half8 fr[DN_L3_BUF][9];
#pragma unroll
      for (int j = 0; j < 9; ++j) fr[0][j] = *reinterpret_cast<const half8*>(xin + rd3[j % 3] + (j / 3) * ROW_BYTES16);
#pragma unroll
      for (int b = 0; b < DN_NB3; ++b) {
        if (DN_L3_BUF == 2 && b + 1 < DN_NB3) {
#pragma unroll
          for (int j = 0; j < 9; ++j) fr[(b + 1) & 1][j] = *reinterpret_cast<const half8*>(xin + rd3[j % 3] + (16 * (b + 1) + j / 3) * ROW_BYTES16);
        }
        if (DN_L3_BUF == 1 && b > 0) {
#pragma unroll
          for (int j = 0; j < 9; ++j) fr[0][j] = *reinterpret_cast<const half8*>(xin + rd3[j % 3] + (16 * b + j / 3) * ROW_BYTES16);
        }
        asm volatile("" ::: "memory");
        f32x4 acc = b3;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[j], fr[b & (DN_L3_BUF - 1)][j], acc, 0, 0, 0);
        if (b > 0) finish3(accp, b - 1);
        accp = acc;
      }
      finish3(accp, DN_NB3 - 1);
    }
    // no barrier here: layer 1 of the next strip writes x1, which every wave finished reading before the barrier
    // above; x2 is rewritten only after the next strip's first barrier, i.e. after every wave left this loop
  };
#pragma unroll 1
  for (int s = 0; s < n_strips; ++s) {
    const int r0 = s * DN_T;
    if (DN_TWO_BODIES && r0 >= 3 && r0 + DN_T + 3 <= n_sc) strip(std::false_type{}, s);
    else strip(std::true_type{}, s);
  }
}

_Float16 to_h(float v) { return (_Float16)v; }

}  // namespace

extern "C" int ce_denoiser_create(int32_t device, const float* w1, const float* b1, const float* w2, const float* b2,
                                  const float* w3, const float* b3, ce_denoiser** out) {
  if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out) return ce_fail(CE_ERR_INVALID, "null argument");
  // fragment f, lane (row = c_out = lane & 15, g = lane >> 4), element j: the weight that multiplies B's k = 8 g + j
  std::vector<_Float16> frag(15 * 64 * 8, to_h(0.f));
  auto at = [&](int f, int lane, int j) -> _Float16& { return frag[((size_t)f * 64 + lane) * 8 + j]; };
  for (int lane = 0; lane < 64; ++lane) {
    const int co = lane & 15, g = lane >> 4;
    for (int j = 0; j < 8; ++j) {  // layer 1: k = 8 g + j = 2 tap + (re | im)
      const int k = 8 * g + j, tap = k >> 1, ci = k & 1;
      if (tap < 9) at(0, lane, j) = to_h(w1[(co * 2 + ci) * 9 + tap]);
    }
    for (int j = 0; j < 8; ++j) {  // layer 2: k = 16 (g >> 1) + c_in; fragments A (ky = 0, 1, 2): taps (ky, dx=-1 | ky, dx=0);
      const int ci = 8 * (g & 1) + j, up = g >> 1;   // C0: (ky=0, dx=+1 | ky=1, dx=+1); C1: (ky=2, dx=+1 | zero)
      for (int ky = 0; ky < 3; ++ky) at(1 + ky, lane, j) = to_h(w2[(co * 16 + ci) * 9 + ky * 3 + up]);
      at(4, lane, j) = to_h(w2[(co * 16 + ci) * 9 + up * 3 + 2]);
      if (!up) at(5, lane, j) = to_h(w2[(co * 16 + ci) * 9 + 2 * 3 + 2]);
    }
    // layer 3, banded over a block of 4 output rows: D row = 2 r + (re | im), r = 0..3 (rows 8..15 unused); K-step m,
    // lane group g: input row ri = m / 3 + 3 (g >> 1) of the block's 6, symbol tap dx = m % 3, channels 8 (g & 1) + j
    if (co < 8)
      for (int m = 0; m < 9; ++m)
        for (int j = 0; j < 8; ++j) {
          const int r = co >> 1, c3 = co & 1, ri = m / 3 + 3 * (g >> 1), dx = m % 3, ci = 8 * (g & 1) + j, ky = ri - r;
          if (ky >= 0 && ky <= 2) at(6 + m, lane, j) = to_h(w3[(c3 * 16 + ci) * 9 + ky * 3 + dx]);
        }
  }
  float bias[48] = {0};
  for (int i = 0; i < 16; ++i) { bias[i] = b1[i]; bias[16 + i] = b2[i]; }
  bias[32] = b3[0];
  bias[33] = b3[1];
  ce_denoiser* d = new ce_denoiser;
  d->device = device;
  CeDeviceScope scope(device);
  hipError_t e = scope.err;
  if (e == hipSuccess) e = hipMalloc(&d->wfrag, frag.size() * sizeof(_Float16));
  if (e == hipSuccess) e = hipMalloc(&d->bias, sizeof(bias));
  if (e == hipSuccess) e = hipMemcpy(d->wfrag, frag.data(), frag.size() * sizeof(_Float16), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d->bias, bias, sizeof(bias), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    ce_denoiser_destroy(d);
    return ce_fail(CE_ERR_HIP, "denoiser upload: %s", hipGetErrorString(e));
  }
  *out = d;
  return CE_OK;
}

extern "C" void ce_denoiser_destroy(ce_denoiser* d) {
  if (!d) return;
  if (d->wfrag) (void)hipFree(d->wfrag);
  if (d->bias) (void)hipFree(d->bias);
  delete d;
}

extern "C" int ce_denoise_batch(const ce_denoiser* d, void* ch_est, int64_t n_items, int32_t n_sc, int32_t n_sym,
                                int32_t n_layers, void* stream) {
  if (!d || (!ch_est && n_items > 0)) return ce_fail(CE_ERR_INVALID, "null argument");
  if (n_sym != CE_DN_SYMBOLS) return ce_fail(CE_ERR_UNSUPPORTED, "the denoiser is built for 14-symbol grids (got %d)", n_sym);
  if (n_items < 0 || n_sc < 1 || n_layers < 1 || n_layers > CE_MAX_LAYERS) return ce_fail(CE_ERR_INVALID, "bad shape");
  const int64_t planes = n_items * n_layers;
  if (planes == 0) return CE_OK;
  if (planes > 0x7FFFFFFFll) return ce_fail(CE_ERR_UNSUPPORTED, "%lld planes in one launch", (long long)planes);
  CeDeviceScope scope(d->device);
  if (scope.err != hipSuccess) return ce_fail(CE_ERR_HIP, "device %d: %s", d->device, hipGetErrorString(scope.err));
  hipLaunchKernelGGL(ce_denoise_kernel, dim3((unsigned)planes), dim3(DN_NT), 0, (hipStream_t)stream,
                     reinterpret_cast<float2*>(ch_est), d->wfrag, d->bias, n_sc, n_layers);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CE_OK : ce_fail(CE_ERR_HIP, "denoise launch: %s", hipGetErrorString(e));
}
