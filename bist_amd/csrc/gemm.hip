// GEMM for the BiST hot path on gfx950:  C = epilogue(alpha * A . B^T), fp32 accumulate on MFMA.
//
// One 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA
// fragments of 16x16).  Every operand tile is 16 KiB of LDS per stage and is filled by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave instruction, LDS image lane-linear, any swizzle applied
// on the per-lane SOURCE address), double-buffered with one barrier per K tile and the next tile's
// DMA in flight under the MFMAs.  Two operand layouts, chosen per operand from its strides:
//
//   K-contiguous ("N"):  x.W^T of nn.Linear.  LDS image [128 rows][128 B of K], 16-byte chunks
//       XOR-swizzled by ((row>>1)&7); fragments by ds_read_b128 (conflict-free).  One 16-byte read
//       feeds one v_mfma_f32_16x16x32_bf16 (8 bf16) or four v_mfma_f32_16x16x4_f32.
//   row-contiguous ("T"): the backward products x^T.dy and dy.W read their operands in place, no
//       transpose pass.  LDS image [K rows][128 elements]; bf16 fragments by ds_read_b64_tr_b16 (the
//       hardware transposing read: 4 K-rows x 16 columns per 16-lane group), f32 fragments by
//       ds_read_b32; 16-byte pieces XOR-swizzled by K-row so both are conflict-free.
//
// Split-K: problems with few output tiles and a long K (weight gradients: 512x512 outputs, K = B*T*S)
// are cut along K over several workgroups that write fp32 partial slabs to a caller-provided
// workspace; a second small kernel sums the slabs and applies the epilogue.
//
// A generic register-staged kernel (any strides, any K, zero-filled tails) remains as the fallback.
// Workgroup ids are remapped so that each XCD (blocks b, b+8, ... share one) walks a contiguous range
// of tiles: neighbouring tiles share an A row panel, which then stays in that XCD's L2.
#include "common.hpp"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;                 // K extent of an "N" tile in bytes
constexpr int TILE_BYTES = BM * ROW_BYTES;     // 16 KiB per operand per stage
constexpr int NTHREADS = 256;
constexpr int WS_HEADER = 1024;                // floats at the head of the workspace: split-K ticket counters (kept zero between calls)

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct GemmK {   // device-side argument block (by value)
  const char* A; const char* B; char* C; const char* bias; const char* residual;
  int M, N, K;
  long a_rs, a_ks, b_rs, b_ks, ldc, ldr;
  int batch2;
  long a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2, r_bs1, r_bs2, bias_bs1, bias_bs2;
  float alpha; int act; int res_outer, res_inner;
  float drop_p; unsigned long long drop_seed; const unsigned long long* drop_ctr;
  int tiles_m, tiles_n;
  int split_k; float* ws;            // split_k > 1: raw fp32 partial tiles go to ws[z][split][M][N]
  int vec_c, vec_r;                  // 16-byte aligned output / residual rows: vector epilogue allowed
  int dbg;                           // BIST_GEMM_DBG ablation aid (0 in production): 1 exit at entry, 2 skip the K loop, 3 skip the epilogue
  // LayerNorm prologue (gemm_t64_pre_kernel, K = 512 bf16): A rows are normalised in LDS before the products and written to ln_out
  const char* ln_a; const char* ln_b; char* ln_out; long ln_ld; float ln_eps;
  int ln_mode;                       // 0: LayerNorm of the A rows (prologue); 1: LayerNorm of the OUTPUT rows (epilogue of the 256-tile kernel, N = 512)
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void step(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void step(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), c, 0, 0, 0);
  }
};

// tile id -> (z, split, tm, tn) with the XCD-contiguous remap (bijective for any grid size)
__device__ __forceinline__ void tile_coords(const GemmK& g, int& z, int& sp, int& tm, int& tn, unsigned bid = blockIdx.x,
                                            unsigned nwg = gridDim.x) {
  const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
  unsigned lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  tn = lid % g.tiles_n; lid /= g.tiles_n;
  tm = lid % g.tiles_m; lid /= g.tiles_m;
  sp = lid % g.split_k;
  z = lid / g.split_k;
}

// ---- fragment loads -----------------------------------------------------------------------------
// 16 rows starting at `row0` (multiple of 16) of the operand tile at `tile`, K step ks (64 bytes of K
// for "N" tiles, 32 bf16 / 16 f32 K-rows for "T" tiles).  The returned uint4 is what Mma<T>::step eats:
// lane (x = lane&15, kg = lane>>4) holds row x, k = 8*kg..8*kg+7 (bf16) or k = 4*kg+i for MFMA i (f32).
template <typename T, bool TR>
__device__ __forceinline__ uint4 load_frag(const char* tile, int row0, int ks, int lane) {
  const int x = lane & 15, kg = lane >> 4;
  if constexpr (!TR) {
    const int off = ((ks * 4 + kg) ^ (x >> 1)) << 4;
    return *reinterpret_cast<const uint4*>(tile + (row0 + x) * ROW_BYTES + off);
  } else if constexpr (sizeof(T) == 2) {
    // [64 k-rows][256 B]; piece (16 B = 8 rows) index c stored at c ^ (2*(k&3)) ^ (8*((k>>3)&1))
    const int q = x >> 2, p = x & 3;
    const int kb = ks * 32 + kg * 8;
    const int c = (row0 >> 3) + (p >> 1);
    const int pos = c ^ (2 * q) ^ (8 * (kg & 1));
    const char* a0 = tile + (kb + q) * 256 + pos * 16 + (p & 1) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * 256));
    uint4 r;
    r.x = __builtin_bit_cast(uint2, lo).x; r.y = __builtin_bit_cast(uint2, lo).y;
    r.z = __builtin_bit_cast(uint2, hi).x; r.w = __builtin_bit_cast(uint2, hi).y;
    return r;
  } else {
    // [32 k-rows][512 B]; piece (16 B = 4 rows) index c stored at c ^ (4*((k>>2)&7))
    const int kb = ks * 16 + kg * 4;                 // k = kb + i for MFMA i; (k>>2) = ks*4 + kg for all i
    const int pos = (((row0 + x) >> 2) ^ (4 * ((ks * 4 + kg) & 7)));
    const char* a0 = tile + kb * 512 + pos * 16 + (x & 3) * 4;
    uint4 r;
    r.x = *reinterpret_cast<const unsigned*>(a0);
    r.y = *reinterpret_cast<const unsigned*>(a0 + 512);
    r.z = *reinterpret_cast<const unsigned*>(a0 + 1024);
    r.w = *reinterpret_cast<const unsigned*>(a0 + 1536);
    return r;
  }
}

template <typename T, bool ATR, bool BTR>
__device__ __forceinline__ void compute_tile(const char* lds_a, const char* lds_b, f32x4 (&acc)[4][4], int wm, int wn, int lane,
                                             bool half = false) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if (ks == 1 && half) break;            // K tail of half a tile: only the first 64 bytes / K-rows are valid
    uint4 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = load_frag<T, ATR>(lds_a, wm * 64 + i * 16, ks, lane);
      bf[i] = load_frag<T, BTR>(lds_b, wn * 64 + i * 16, ks, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Mma<T>::step(bf[j], af[i], acc[i][j]);
  }
}

// ---- per-lane LDS-DMA source pointers ---------------------------------------------------------------
// Wave w issues instructions j = 0..3 into LDS bytes [(w*4+j)*1024, +1024) of the operand tile.
__device__ uint4 g_zero16;        // 16 zero bytes: the DMA source of every lane that falls past K on a trailing partial tile

template <typename T, bool TR>
struct Stager {
  const char* p[4];
  int kofs[4];      // this lane's K offset (elements) inside a tile, per instruction: lanes with kofs >= K remainder read zeros
  long step;
  __device__ __forceinline__ void init(const char* base, long rs, long ks_stride, int r0, int rows, int k0, int w, int lane) {
    constexpr int SZ = (int)sizeof(T);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int inst = w * 4 + j;
      if constexpr (!TR) {
        const int row = inst * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((j & 1) << 2) | (lane >> 4));          // inverse of the read swizzle
        const int rr = min(r0 + row, rows - 1);                                   // tails re-read a valid row
        p[j] = base + ((long)rr * rs + k0) * SZ + chunk * 16;
        kofs[j] = chunk * (16 / SZ);
      } else if constexpr (SZ == 2) {
        const int krow = lane >> 4, k = inst * 4 + krow;
        int c = (lane & 15) ^ (2 * krow) ^ (8 * ((inst >> 1) & 1));
        if (r0 + c * 8 >= rows) c = 0;                                            // rows % 8 == 0: whole piece in or out
        p[j] = base + ((long)(k0 + k) * ks_stride + r0 + c * 8) * SZ;
        kofs[j] = k;
      } else {
        const int krow = lane >> 5, k = inst * 2 + krow;
        int c = (lane & 31) ^ (4 * ((inst >> 1) & 7));
        if (r0 + c * 4 >= rows) c = 0;
        p[j] = base + ((long)(k0 + k) * ks_stride + r0 + c * 4) * SZ;
        kofs[j] = k;
      }
    }
    step = TR ? (long)(ROW_BYTES / SZ) * ks_stride * SZ : ROW_BYTES;
  }
  __device__ __forceinline__ void issue(char* lds_tile, int w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(p[j]), LDS_PTR(lds_tile + (w * 4 + j) * 1024), 16, 0, 0);
      p[j] += step;
    }
  }
  // trailing partial tile: only the first `krem` K elements exist
  __device__ __forceinline__ void issue_tail(char* lds_tile, int w, int krem) {
    const char* zp = reinterpret_cast<const char*>(&g_zero16);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds(GLB_PTR(kofs[j] < krem ? p[j] : zp), LDS_PTR(lds_tile + (w * 4 + j) * 1024), 16, 0, 0);
  }
};

template <typename T, typename TO>
__device__ __forceinline__ float epilogue_value(const GemmK& g, float acc, float bv, const TO* res, int m, int n, unsigned long long zoff) {
  float v = acc * g.alpha + bv;
  if (g.act == BIST_ACT_RELU) v = fmaxf(v, 0.f);
  if (g.drop_p > 0.f) {
    const unsigned long long seed = g.drop_seed + (g.drop_ctr ? g.drop_ctr[0] * 0xD1B54A32D192ED03ULL : 0ULL);
    v = drop_keep(seed, zoff + (unsigned long long)m * g.N + n, g.drop_p) ? v * (1.f / (1.f - g.drop_p)) : 0.f;
  }
  if (res) {
    const long rr = g.res_outer > 0 ? (long)(m / g.res_outer) * g.res_inner + (m % g.res_inner) : (long)m;
    const float r = to_f(res[rr * g.ldr + n]);
    v = g.act == BIST_ACT_GATE ? (r > 0.f ? v : 0.f) : v + r;
  }
  return v;
}

// v_permlane16_swap: lanes 16-31 / 48-63 of `a` trade places with lanes 0-15 / 32-47 of `b`.  Inline asm (with the two
// wait states the instruction needs after a VALU write of an operand): the hipcc 7.2 builtin drops the second result.
__device__ __forceinline__ void swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// Row-wise finish of VW consecutive output columns held by one lane (the layout an MFMA leaves when the B fragment
// is its row operand): alpha, bias, ReLU, dropout, residual, then one or two 16-byte stores.  The per-launch switches are
// wave-uniform and tested once per row piece, not once per element.
template <typename T, typename TO>
struct RowOut {
  const GemmK& g;
  TO* C;
  const T* bias;
  const TO* res;
  unsigned long long zoff, seed;
  float keep_scale;
  bool relu, dropping, gate;
  __device__ __forceinline__ RowOut(const GemmK& g_, int z1, int z2, unsigned long long zoff_) : g(g_), zoff(zoff_) {
    C = reinterpret_cast<TO*>(g.C) + z1 * g.c_bs1 + z2 * g.c_bs2;
    bias = g.bias ? reinterpret_cast<const T*>(g.bias) + z1 * g.bias_bs1 + z2 * g.bias_bs2 : nullptr;
    res = g.residual ? reinterpret_cast<const TO*>(g.residual) + z1 * g.r_bs1 + z2 * g.r_bs2 : nullptr;
    relu = g.act == BIST_ACT_RELU;
    gate = g.act == BIST_ACT_GATE;
    dropping = g.drop_p > 0.f;
    seed = g.drop_seed + ((dropping && g.drop_ctr) ? g.drop_ctr[0] * 0xD1B54A32D192ED03ULL : 0ULL);
    keep_scale = dropping ? 1.f / (1.f - g.drop_p) : 1.f;
  }
  template <int VW>
  __device__ __forceinline__ void load_bias(float (&bv)[VW], int n) const {
#pragma unroll
    for (int e = 0; e < VW; ++e) bv[e] = 0.f;
    if (!bias || n >= g.N) return;
    if (n + VW <= g.N && (n & 7) == 0 && sizeof(T) == 2 && VW == 8 && (reinterpret_cast<size_t>(bias) & 15) == 0) {
      T q[8];
      *reinterpret_cast<uint4*>(q) = *reinterpret_cast<const uint4*>(bias + n);
#pragma unroll
      for (int e = 0; e < VW; ++e) bv[e] = to_f(q[e]);
    } else {
#pragma unroll
      for (int e = 0; e < VW; ++e) if (n + e < g.N) bv[e] = to_f(bias[n + e]);
    }
  }
  // Whole-tile fast path: every row and column of the wave's tile is inside the matrix, 16-byte rows everywhere, the
  // residual (if any) maps row to row, N % 4 == 0.  Straight-line code, one instantiation per (dropout, residual) pair,
  // so that a wave fetches only the instructions it runs (the generic row() below is ~10x the code).
  __device__ __forceinline__ bool whole(int row0, int rows, int col0, int cols) const {
    return row0 + rows <= g.M && col0 + cols <= g.N && g.vec_c && (g.N & 3) == 0 && (!res || (g.vec_r && g.res_outer <= 0));
  }
  template <int VW, bool DROP, bool RES, int NB>
  __device__ __forceinline__ void row_whole(float (&v)[VW], const float (&bv)[NB], int m, int n, uint32_t thr) const {
#pragma unroll
    for (int e = 0; e < VW; ++e) v[e] = v[e] * g.alpha + bv[e];
    if (relu) {
#pragma unroll
      for (int e = 0; e < VW; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if constexpr (DROP) {
      const unsigned long long i4 = (zoff + (unsigned long long)m * g.N + n) >> 2;
#pragma unroll
      for (int q = 0; q < VW / 4; ++q) {
        const uint64_t bits = drop_bits4(seed, i4 + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = drop_keep_of(bits, e, thr) ? v[4 * q + e] * keep_scale : 0.f;
      }
    }
    if constexpr (RES) {
      TO q[VW];
      const TO* rp = res + (long)m * g.ldr + n;
#pragma unroll
      for (int c = 0; c < (int)(VW * sizeof(TO) / 16); ++c) reinterpret_cast<uint4*>(q)[c] = reinterpret_cast<const uint4*>(rp)[c];
      if (gate) {
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] = to_f(q[e]) > 0.f ? v[e] : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] += to_f(q[e]);
      }
    }
    TO o[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) o[e] = from_f<TO>(v[e]);
    TO* dst = C + (long)m * g.ldc + n;
#pragma unroll
    for (int c = 0; c < (int)(VW * sizeof(TO) / 16); ++c) reinterpret_cast<uint4*>(dst)[c] = reinterpret_cast<const uint4*>(o)[c];
  }
  template <int VW, int NB>
  __device__ __forceinline__ void row(float (&v)[VW], const float (&bv)[NB], int m, int n) const {
    static_assert(NB >= VW, "bias piece too short");
    if (m >= g.M || n >= g.N) return;
    const bool full = n + VW <= g.N;
#pragma unroll
    for (int e = 0; e < VW; ++e) v[e] = v[e] * g.alpha + bv[e];
    if (relu) {
#pragma unroll
      for (int e = 0; e < VW; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (dropping) {
#pragma unroll
      for (int e = 0; e < VW; ++e) v[e] = drop_keep(seed, zoff + (unsigned long long)m * g.N + n + e, g.drop_p) ? v[e] * keep_scale : 0.f;
    }
    if (res) {
      const long rr = g.res_outer > 0 ? (long)(m / g.res_outer) * g.res_inner + (m % g.res_inner) : (long)m;
      const TO* rp = res + rr * g.ldr + n;
      if (full && g.vec_r) {
        TO q[VW];
        if constexpr (VW * sizeof(TO) == 8) *reinterpret_cast<uint2*>(q) = *reinterpret_cast<const uint2*>(rp);
        else {
#pragma unroll
          for (int c = 0; c < (int)(VW * sizeof(TO) / 16); ++c) reinterpret_cast<uint4*>(q)[c] = reinterpret_cast<const uint4*>(rp)[c];
        }
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] = gate ? (to_f(q[e]) > 0.f ? v[e] : 0.f) : v[e] + to_f(q[e]);
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) if (n + e < g.N) v[e] = gate ? (to_f(rp[e]) > 0.f ? v[e] : 0.f) : v[e] + to_f(rp[e]);
      }
    }
    TO o[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) o[e] = from_f<TO>(v[e]);
    TO* dst = C + (long)m * g.ldc + n;
    if (full && g.vec_c) {
      if constexpr (VW * sizeof(TO) == 8) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(o);
      else {
#pragma unroll
        for (int c = 0; c < (int)(VW * sizeof(TO) / 16); ++c) reinterpret_cast<uint4*>(dst)[c] = reinterpret_cast<const uint4*>(o)[c];
      }
    } else {
#pragma unroll
      for (int e = 0; e < VW; ++e) if (n + e < g.N) dst[e] = o[e];
    }
  }
};

// Epilogue straight from the accumulators.  Every kernel below feeds the MFMA the B fragment as its ROW operand, so lane
// (lr = lane & 15, lg = lane >> 4) of fragment (i, j) holds the 4 CONSECUTIVE COLUMNS C[row0 + i*16 + lr][col0 + j*16 + lg*4 .. +3]:
// fp32 outputs are 16-byte stores as they stand; for 2-byte outputs one v_permlane16_swap per register between the
// fragments j, j+1 leaves each lane 8 consecutive columns (16 bytes).  No LDS image, no barrier: a finished wave starts
// storing while the others still compute, and at two workgroups per CU the stores run under the neighbour's K loop
// (a CU stores ~10 B/clk, so the C tile is the longest serial piece of a short product).
template <typename T, typename TO, int NI, int NJ, int MODE>     // MODE: 0..3 = whole-tile fast path (bit 0 dropout, bit 1 residual), 4 = generic
__device__ __forceinline__ void frag_rows(const RowOut<T, TO>& out, f32x4 (&acc)[NI][NJ], int row0, int col0, int lane) {
  const int lr = lane & 15, lg = lane >> 4;
  const uint32_t thr = drop_threshold(out.g.drop_p);
  if constexpr (sizeof(TO) == 2) {
    static_assert(NJ % 2 == 0, "fragment pairs");
    const int cofs = (lg & 1) * 16 + (lg >> 1) * 8;
    float bv[NJ / 2][8];
#pragma unroll
    for (int jp = 0; jp < NJ / 2; ++jp) out.load_bias(bv[jp], col0 + jp * 32 + cofs);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = acc[i][2 * jp][r], b = acc[i][2 * jp + 1][r];
          swap16(a, b);
          v[r] = a; v[4 + r] = b;
        }
        if constexpr (MODE < 4) out.template row_whole<8, (MODE & 1) != 0, (MODE & 2) != 0>(v, bv[jp], row0 + i * 16 + lr, col0 + jp * 32 + cofs, thr);
        else out.template row<8>(v, bv[jp], row0 + i * 16 + lr, col0 + jp * 32 + cofs);
      }
  } else {
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) out.load_bias(bv[j], col0 + j * 16 + lg * 4);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if constexpr (MODE < 4) out.template row_whole<4, (MODE & 1) != 0, (MODE & 2) != 0>(v, bv[j], row0 + i * 16 + lr, col0 + j * 16 + lg * 4, thr);
        else out.template row<4>(v, bv[j], row0 + i * 16 + lr, col0 + j * 16 + lg * 4);
      }
  }
}

template <typename T, typename TO, int NI, int NJ>
__device__ __forceinline__ void frag_out(const GemmK& g, f32x4 (&acc)[NI][NJ], int z1, int z2, int row0, int col0, int lane) {
  const long zlin = z1 * (long)g.batch2 + z2;
  const RowOut<T, TO> out(g, z1, z2, (unsigned long long)zlin * (unsigned long long)g.M * (unsigned long long)g.N);
  if (out.whole(row0, NI * 16, col0, NJ * 16)) {
    const int mode = (out.dropping ? 1 : 0) | (out.res ? 2 : 0);
    if (mode == 0) frag_rows<T, TO, NI, NJ, 0>(out, acc, row0, col0, lane);
    else if (mode == 1) frag_rows<T, TO, NI, NJ, 1>(out, acc, row0, col0, lane);
    else if (mode == 2) frag_rows<T, TO, NI, NJ, 2>(out, acc, row0, col0, lane);
    else frag_rows<T, TO, NI, NJ, 3>(out, acc, row0, col0, lane);
  } else {
    frag_rows<T, TO, NI, NJ, 4>(out, acc, row0, col0, lane);
  }
}

// raw fp32 partial sums of one split-K slice (the reduce kernel applies the epilogue); same fragment layout as frag_out
template <int NI, int NJ>
__device__ __forceinline__ void frag_out_partial(const GemmK& g, f32x4 (&acc)[NI][NJ], long zlin, int sp, int row0, int col0, int lane) {
  const int lr = lane & 15, lg = lane >> 4;
  float* W = g.ws + WS_HEADER + ((zlin * g.split_k + sp) * (long)g.M) * g.N;
  const bool vec = (g.N & 3) == 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = row0 + i * 16 + lr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = col0 + j * 16 + lg * 4;
      float* dst = W + (long)m * g.N + n;
      if (vec && n + 4 <= g.N) *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < g.N) dst[r] = acc[i][j][r];
      }
    }
  }
}

// the f32 [64][64] image of a 64-tile (16-byte chunk ^ 4*((row>>2)&3)) that stream_out<.., 64> reads: in-launch split-K only
__device__ __forceinline__ void frag_to_image64(f32x4 (&acc)[2][2], float* buf, int wm, int wn, int lane) {
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wm * 32 + i * 16 + lr, chunk = (wn * 32 + j * 16) / 4 + lg;
      *reinterpret_cast<float4*>(buf + row * 64 + ((chunk ^ (((row >> 2) & 3) << 2)) << 2)) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}

// Staged epilogue of the fast kernel.  The accumulators go to LDS as f32 (one [64 rows][128 cols] image per
// wave-row half, reusing the two operand stages; 16-byte chunks XOR-swizzled by row so the scattered
// fragment writes do not collide), then every thread streams ONE fixed group of 8 (bf16 out) or 4 (f32 out)
// columns down the rows: its bias values are loaded once, every row is one 16-byte residual load and one
// 16-byte store, and the row loop is fully unrolled so all loads are in flight together.  Versus storing
// straight from the MFMA layout (2-byte stores, 32-byte runs) this cuts the store instructions 8x.
template <typename T, typename TO, bool SPLIT, int TILE = 128>
__device__ __forceinline__ void stream_out(const GemmK& g, const char* lds_lo, const char* lds_hi, int z1, int z2, int sp,
                                           int m0, int n0, int tid) {
  using TS = typename std::conditional<SPLIT, float, TO>::type;      // element type actually stored
  constexpr int VW = 16 / (int)sizeof(TS);          // columns per thread: 8 (bf16) or 4 (f32)
  constexpr int IPR = TILE / VW;                    // threads per row
  constexpr int RPP = NTHREADS / IPR;               // rows per pass
  constexpr int NPASS = TILE / RPP;
  const int col0 = (tid % IPR) * VW, rbase = tid / IPR;
  const int n = n0 + col0;
  if (n >= g.N) return;
  const bool full = n + VW <= g.N;
  const long zlin = z1 * (long)g.batch2 + z2;
  const unsigned long long zoff = (unsigned long long)zlin * (unsigned long long)g.M * (unsigned long long)g.N;
  TS* out_base;
  long ld_out;
  bool vec_out;
  if constexpr (SPLIT) {
    out_base = g.ws + WS_HEADER + ((zlin * g.split_k + sp) * (long)g.M) * g.N;
    ld_out = g.N; vec_out = (g.N & 3) == 0;
  } else {
    out_base = reinterpret_cast<TO*>(g.C) + z1 * g.c_bs1 + z2 * g.c_bs2;
    ld_out = g.ldc; vec_out = g.vec_c != 0;
  }
  float bv[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) bv[e] = 0.f;
  const TO* res = nullptr;
  if constexpr (!SPLIT) {
    if (g.bias) {
      const T* bias = reinterpret_cast<const T*>(g.bias) + z1 * g.bias_bs1 + z2 * g.bias_bs2 + n;
#pragma unroll
      for (int e = 0; e < VW; ++e) if (n + e < g.N) bv[e] = to_f(bias[e]);
    }
    if (g.residual) res = reinterpret_cast<const TO*>(g.residual) + z1 * g.r_bs1 + z2 * g.r_bs2;
  }
  const unsigned long long seed = g.drop_seed + ((g.drop_p > 0.f && g.drop_ctr) ? g.drop_ctr[0] * 0xD1B54A32D192ED03ULL : 0ULL);
  const float keep_scale = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
#pragma unroll
  for (int it = 0; it < NPASS; ++it) {
    const int row = rbase + it * RPP;
    const int m = m0 + row;
    if (m >= g.M) continue;
    const float* src = TILE == 128 ? reinterpret_cast<const float*>(row < 64 ? lds_lo : lds_hi) + (row & 63) * 128
                                   : reinterpret_cast<const float*>(lds_lo) + row * 64;      // 64-tile: one [64][64] image
    const int sw = TILE == 128 ? (((row & 63) >> 2) & 7) << 2 : ((row >> 2) & 3) << 2;
    float v[VW];
#pragma unroll
    for (int q = 0; q < VW / 4; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(src + (((col0 >> 2) + q) ^ sw) * 4);
      v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
    }
    TS o[VW];
    if constexpr (SPLIT) {
#pragma unroll
      for (int e = 0; e < VW; ++e) o[e] = v[e];
    } else {
      float rv[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) rv[e] = 0.f;
      if (res) {
        const long rr = g.res_outer > 0 ? (long)(m / g.res_outer) * g.res_inner + (m % g.res_inner) : (long)m;
        const TO* rp = res + rr * g.ldr + n;
        if (full && g.vec_r) {
          const uint4 qv = *reinterpret_cast<const uint4*>(rp);
          TO qe[VW];
          *reinterpret_cast<uint4*>(qe) = qv;
#pragma unroll
          for (int e = 0; e < VW; ++e) rv[e] = to_f(qe[e]);
        } else {
#pragma unroll
          for (int e = 0; e < VW; ++e) if (n + e < g.N) rv[e] = to_f(rp[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        float x = v[e] * g.alpha + bv[e];
        if (g.act == BIST_ACT_RELU) x = fmaxf(x, 0.f);
        if (g.drop_p > 0.f) x = drop_keep(seed, zoff + (unsigned long long)m * g.N + n + e, g.drop_p) ? x * keep_scale : 0.f;
        o[e] = from_f<TO>(g.act == BIST_ACT_GATE ? (rv[e] > 0.f ? x : 0.f) : x + rv[e]);
      }
    }
    TS* dst = out_base + (long)m * ld_out + n;
    if (full && vec_out) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
    } else {
#pragma unroll
      for (int e = 0; e < VW; ++e) if (n + e < g.N) dst[e] = o[e];
    }
  }
}

template <typename T, typename TO>
__device__ __forceinline__ void epilogue_direct(const GemmK& g, f32x4 (&acc)[4][4], int z1, int z2, int sp, int m0, int n0, int wm, int wn, int lane) {
  if (g.split_k > 1) frag_out_partial<4, 4>(g, acc, z1 * (long)g.batch2 + z2, sp, m0 + wm * 64, n0 + wn * 64, lane);
  else frag_out<T, TO, 4, 4>(g, acc, z1, z2, m0 + wm * 64, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------
// fast kernel: LDS-DMA staging of both operands, double buffer, optional split-K
// ---------------------------------------------------------------------------------------------
template <typename T, typename TO, bool ATR, bool BTR, int NS>
__global__ __launch_bounds__(NTHREADS) void gemm_fast_kernel(const GemmK g) {
  // one LDS object per stage: the compiler tracks in-flight LDS-DMA per object, so the
  // fragment reads of stage s need not wait for the DMA that is filling another stage
  __shared__ __attribute__((aligned(16))) char lds0[2 * TILE_BYTES];   // [A|B]
  __shared__ __attribute__((aligned(16))) char lds1[2 * TILE_BYTES];
  __shared__ __attribute__((aligned(16))) char lds2[NS > 2 ? 2 * TILE_BYTES : 16];
  __shared__ __attribute__((aligned(16))) char lds3[NS > 2 ? 2 * TILE_BYTES : 16];
  if (g.dbg == 1) return;
  int z, sp, tm, tn;
  tile_coords(g, z, sp, tm, tn);
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);

  const int nk_all = (g.K + BK - 1) / BK;                 // a trailing partial tile counts as one
  const int k_tail = g.K % BK;                            // != 0: the last tile holds only k_tail K elements (rest zero-filled)
  const int kt0 = (int)((long)nk_all * sp / g.split_k), kt1 = (int)((long)nk_all * (sp + 1) / g.split_k);
  const int nk = g.dbg == 2 ? 0 : kt1 - kt0;

  Stager<T, ATR> sa;
  Stager<T, BTR> sb;
  sa.init(g.A + (z1 * g.a_bs1 + z2 * g.a_bs2) * (long)sizeof(T), g.a_rs, g.a_ks, m0, g.M, kt0 * BK, w, lane);
  sb.init(g.B + (z1 * g.b_bs1 + z2 * g.b_bs2) * (long)sizeof(T), g.b_rs, g.b_ks, n0, g.N, kt0 * BK, w, lane);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto is_tail = [&](int kt) { return k_tail != 0 && (kt0 + kt == nk_all - 1); };
  auto is_half = [&](int kt) { return is_tail(kt) && k_tail <= BK / 2; };      // second K step is all zeros: skip it
  auto stage_in = [&](char* st, int kt) {
    if (is_tail(kt)) { sa.issue_tail(st, w, k_tail); sb.issue_tail(st + TILE_BYTES, w, k_tail); }
    else { sa.issue(st, w); sb.issue(st + TILE_BYTES, w); }
  };

  if constexpr (NS == 2) {
    if (nk > 0) stage_in(lds0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
      __syncthreads();                     // waits vmcnt(0): tile kt has landed, tile kt-1 fully consumed
      if (kt + 1 < nk) stage_in(lds1, kt + 1);
      compute_tile<T, ATR, BTR>(lds0, lds0 + TILE_BYTES, acc, wm, wn, lane, is_half(kt));
      if (kt + 1 < nk) {
        __syncthreads();
        if (kt + 2 < nk) stage_in(lds0, kt + 2);
        compute_tile<T, ATR, BTR>(lds1, lds1 + TILE_BYTES, acc, wm, wn, lane, is_half(kt + 1));
      }
    }
  } else if constexpr (NS == 22) {
    // paired double buffer for latency-bound launches (few workgroups, one per CU): every barrier hands over
    // TWO K tiles, so the load -> barrier -> compute chain is half as long; 128 KiB of LDS, one workgroup per CU
    if (nk > 0) stage_in(lds0, 0);
    if (nk > 1) stage_in(lds1, 1);
    for (int kt = 0; kt < nk; kt += 4) {
      __syncthreads();
      if (kt + 2 < nk) stage_in(lds2, kt + 2);
      if (kt + 3 < nk) stage_in(lds3, kt + 3);
      compute_tile<T, ATR, BTR>(lds0, lds0 + TILE_BYTES, acc, wm, wn, lane, is_half(kt));
      if (kt + 1 < nk) compute_tile<T, ATR, BTR>(lds1, lds1 + TILE_BYTES, acc, wm, wn, lane, is_half(kt + 1));
      if (kt + 2 < nk) {
        __syncthreads();
        if (kt + 4 < nk) stage_in(lds0, kt + 4);
        if (kt + 5 < nk) stage_in(lds1, kt + 5);
        compute_tile<T, ATR, BTR>(lds2, lds2 + TILE_BYTES, acc, wm, wn, lane, is_half(kt + 2));
        if (kt + 3 < nk) compute_tile<T, ATR, BTR>(lds3, lds3 + TILE_BYTES, acc, wm, wn, lane, is_half(kt + 3));
      }
    }
  } else {
    // 4-stage ring for latency-bound launches (few workgroups, one per CU): three tiles of DMA stay in
    // flight across the barriers.  Each wave first waits for ITS OWN pieces of tile kt with a counted
    // vmcnt (8 DMA instructions per tile per wave), then the raw barrier makes every wave's pieces
    // visible; the stage refilled after the barrier is the one all waves finished reading before it.
    if (nk > 0) stage_in(lds0, 0);
    if (nk > 1) stage_in(lds1, 1);
    if (nk > 2) stage_in(lds2, 2);
    auto phase = [&](char* cur, char* refill, int kt) {
      const int ahead = min(kt + 2, nk - 1) - kt;          // tiles issued after kt that may stay in flight
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 3 < nk) stage_in(refill, kt + 3);
      compute_tile<T, ATR, BTR>(cur, cur + TILE_BYTES, acc, wm, wn, lane, is_half(kt));
    };
    for (int kt = 0; kt < nk; kt += 4) {
      phase(lds0, lds3, kt);
      if (kt + 1 < nk) phase(lds1, lds0, kt + 1);
      if (kt + 2 < nk) phase(lds2, lds1, kt + 2);
      if (kt + 3 < nk) phase(lds3, lds2, kt + 3);
    }
  }
  if (g.dbg == 3) { if (acc[0][0][0] == 123.456f) g.C[0] = 1; return; }
  epilogue_direct<T, TO>(g, acc, z1, z2, sp, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// 64x64-tile kernel for launches that cannot fill the chip with 128x128 tiles (the M = B*Lq = 320-row
// products of the decoder / query streams: 12 big tiles on 256 CUs).  Same LDS-DMA staging and the same
// two operand layouts, but a workgroup owns a 64x64 tile (4 waves as 2x2, 32x32 each), so the launch has 4x
// the workgroups, several of them share a CU (32 KiB of LDS each) and hide each other's load and LDS
// latency, and M = 320 is an exact multiple of the tile.  Measured with BIST_GEMM_DBG on the 128 tile at
// 12 workgroups: 1.8 us launch + 5 us serial K loop (one wave per SIMD) + 4 us epilogue.
//   "N" image: [64 rows][128 B of K], chunk ^ ((row>>1)&7)                    (as the 128 tile)
//   "T" image, bf16: [64 K-rows][128 B = 64 elements], piece ^ 2*f(k), f = ((k>>1)&1) | 2*((k>>3)&1)
//   "T" image, f32:  [32 K-rows][256 B = 64 elements], piece ^ 4*((k>>2)&3)
// ---------------------------------------------------------------------------------------------
constexpr int T64 = 64;
constexpr int T64_BYTES = T64 * ROW_BYTES;      // 8 KiB per operand per stage

template <typename T, bool TR>
__device__ __forceinline__ uint4 load_frag64(const char* tile, int row0, int ks, int lane) {
  const int x = lane & 15, kg = lane >> 4;
  if constexpr (!TR) {
    const int off = ((ks * 4 + kg) ^ (x >> 1)) << 4;
    return *reinterpret_cast<const uint4*>(tile + (row0 + x) * ROW_BYTES + off);
  } else if constexpr (sizeof(T) == 2) {
    const int q = x >> 2, p = x & 3;
    const int kb = ks * 32 + kg * 8;
    const int c = (row0 >> 3) + (p >> 1);
    const int pos = c ^ (((q >> 1) | ((kg & 1) << 1)) << 1);
    const char* a0 = tile + (kb + q) * 128 + pos * 16 + (p & 1) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * 128));
    uint4 r;
    r.x = __builtin_bit_cast(uint2, lo).x; r.y = __builtin_bit_cast(uint2, lo).y;
    r.z = __builtin_bit_cast(uint2, hi).x; r.w = __builtin_bit_cast(uint2, hi).y;
    return r;
  } else {
    const int kb = ks * 16 + kg * 4;
    const int pos = ((row0 + x) >> 2) ^ (kg << 2);
    const char* a0 = tile + kb * 256 + pos * 16 + (x & 3) * 4;
    uint4 r;
    r.x = *reinterpret_cast<const unsigned*>(a0);
    r.y = *reinterpret_cast<const unsigned*>(a0 + 256);
    r.z = *reinterpret_cast<const unsigned*>(a0 + 512);
    r.w = *reinterpret_cast<const unsigned*>(a0 + 768);
    return r;
  }
}

// Wave w issues instructions j = 0..1 into LDS bytes [(w*2+j)*1024, +1024) of the operand tile.
template <typename T, bool TR>
struct Stager64 {
  const char* p[2];
  int kofs[2];
  long step;
  __device__ __forceinline__ void init(const char* base, long rs, long ks_stride, int r0, int rows, int k0, int w, int lane) {
    constexpr int SZ = (int)sizeof(T);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int inst = w * 2 + j;
      if constexpr (!TR) {
        const int row = inst * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((j & 1) << 2) | (lane >> 4));
        const int rr = min(r0 + row, rows - 1);
        p[j] = base + ((long)rr * rs + k0) * SZ + chunk * 16;
        kofs[j] = chunk * (16 / SZ);
      } else if constexpr (SZ == 2) {
        const int k = inst * 8 + (lane >> 3);
        int c = (lane & 7) ^ ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1);
        if (r0 + c * 8 >= rows) c = 0;
        p[j] = base + ((long)(k0 + k) * ks_stride + r0 + c * 8) * SZ;
        kofs[j] = k;
      } else {
        const int k = inst * 4 + (lane >> 4);
        int c = (lane & 15) ^ ((inst & 3) << 2);
        if (r0 + c * 4 >= rows) c = 0;
        p[j] = base + ((long)(k0 + k) * ks_stride + r0 + c * 4) * SZ;
        kofs[j] = k;
      }
    }
    step = TR ? (long)(ROW_BYTES / SZ) * ks_stride * SZ : ROW_BYTES;
  }
  __device__ __forceinline__ void issue(char* lds_tile, int w) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(p[j]), LDS_PTR(lds_tile + (w * 2 + j) * 1024), 16, 0, 0);
      p[j] += step;
    }
  }
  __device__ __forceinline__ void issue_tail(char* lds_tile, int w, int krem) {
    const char* zp = reinterpret_cast<const char*>(&g_zero16);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds(GLB_PTR(kofs[j] < krem ? p[j] : zp), LDS_PTR(lds_tile + (w * 2 + j) * 1024), 16, 0, 0);
  }
};

template <typename T, typename TO, bool ATR, bool BTR, bool PAIR>
__global__ __launch_bounds__(NTHREADS) void gemm_t64_kernel(const GemmK g) {
  __shared__ __attribute__((aligned(16))) char lds0[2 * T64_BYTES];   // [A|B]; also the f32 [64][64] epilogue image
  __shared__ __attribute__((aligned(16))) char lds1[2 * T64_BYTES];
  __shared__ __attribute__((aligned(16))) char lds2[PAIR ? 2 * T64_BYTES : 16];   // PAIR: every barrier hands over two K tiles
  __shared__ __attribute__((aligned(16))) char lds3[PAIR ? 2 * T64_BYTES : 16];
  int z, sp, tm, tn;
  tile_coords(g, z, sp, tm, tn);
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const int m0 = tm * T64, n0 = tn * T64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);
  const int nk = (g.K + BK - 1) / BK;
  const int k_tail = g.K % BK;

  Stager64<T, ATR> sa;
  Stager64<T, BTR> sb;
  sa.init(g.A + (z1 * g.a_bs1 + z2 * g.a_bs2) * (long)sizeof(T), g.a_rs, g.a_ks, m0, g.M, 0, w, lane);
  sb.init(g.B + (z1 * g.b_bs1 + z2 * g.b_bs2) * (long)sizeof(T), g.b_rs, g.b_ks, n0, g.N, 0, w, lane);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto is_tail = [&](int kt) { return k_tail != 0 && kt == nk - 1; };
  auto is_half = [&](int kt) { return is_tail(kt) && k_tail <= BK / 2; };
  auto stage_in = [&](char* st, int kt) {
    if (is_tail(kt)) { sa.issue_tail(st, w, k_tail); sb.issue_tail(st + T64_BYTES, w, k_tail); }
    else { sa.issue(st, w); sb.issue(st + T64_BYTES, w); }
  };
  auto compute = [&](const char* st, bool half) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1 && half) break;
      uint4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = load_frag64<T, ATR>(st, wm * 32 + i * 16, ks, lane);
        bf[i] = load_frag64<T, BTR>(st + T64_BYTES, wn * 32 + i * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<T>::step(bf[j], af[i], acc[i][j]);
    }
  };
  stage_in(lds0, 0);
  if constexpr (!PAIR) {
    for (int kt = 0; kt < nk; kt += 2) {
      __syncthreads();
      if (kt + 1 < nk) stage_in(lds1, kt + 1);
      compute(lds0, is_half(kt));
      if (kt + 1 < nk) {
        __syncthreads();
        if (kt + 2 < nk) stage_in(lds0, kt + 2);
        compute(lds1, is_half(kt + 1));
      }
    }
  } else {
    if (nk > 1) stage_in(lds1, 1);
    for (int kt = 0; kt < nk; kt += 4) {
      __syncthreads();
      if (kt + 2 < nk) stage_in(lds2, kt + 2);
      if (kt + 3 < nk) stage_in(lds3, kt + 3);
      compute(lds0, is_half(kt));
      if (kt + 1 < nk) compute(lds1, is_half(kt + 1));
      if (kt + 2 < nk) {
        __syncthreads();
        if (kt + 4 < nk) stage_in(lds0, kt + 4);
        if (kt + 5 < nk) stage_in(lds1, kt + 5);
        compute(lds2, is_half(kt + 2));
        if (kt + 3 < nk) compute(lds3, is_half(kt + 3));
      }
    }
  }
  frag_out<T, TO, 2, 2>(g, acc, z1, z2, m0 + wm * 32, n0 + wn * 32, lane);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// The fragment reads of load_frag64 as inline asm on 32-bit LDS addresses (same geometry, same swizzles).  issue()
// starts the read(s); the values may only be touched after a manual s_waitcnt lgkmcnt and tie(), which pins the
// use behind that wait.
template <typename T, bool TR> struct FragRd;
template <typename T> struct FragRd<T, false> {           // K-contiguous image [64 rows][128 B]
  static constexpr int NREAD = 1;
  u32x4 v;
  __device__ __forceinline__ void issue(unsigned tile, int row0, int ks, int lane) {
    const int x = lane & 15, kg = lane >> 4;
    const unsigned a = tile + (unsigned)((row0 + x) * ROW_BYTES + (((ks * 4 + kg) ^ (x >> 1)) << 4));
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  }
  __device__ __forceinline__ void tie() { asm volatile("" : "+v"(v)); }
  __device__ __forceinline__ uint4 get() const { return __builtin_bit_cast(uint4, v); }
};
template <> struct FragRd<bf16_t, true> {                  // [64 K-rows][128 B], transposing 64-bit reads
  static constexpr int NREAD = 2;
  u32x2 lo, hi;
  __device__ __forceinline__ void issue(unsigned tile, int row0, int ks, int lane) {
    const int x = lane & 15, kg = lane >> 4, q = x >> 2, p = x & 3;
    const int kb = ks * 32 + kg * 8;
    const int c = (row0 >> 3) + (p >> 1);
    const int pos = c ^ (((q >> 1) | ((kg & 1) << 1)) << 1);
    const unsigned a = tile + (unsigned)((kb + q) * 128 + pos * 16 + (p & 1) * 8);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi) : "v"(a) : "memory");
  }
  __device__ __forceinline__ void tie() { asm volatile("" : "+v"(lo), "+v"(hi)); }
  __device__ __forceinline__ uint4 get() const { uint4 r; r.x = lo.x; r.y = lo.y; r.z = hi.x; r.w = hi.y; return r; }
};
template <> struct FragRd<float, true> {                   // [32 K-rows][256 B], four 32-bit reads one K-row apart
  static constexpr int NREAD = 4;
  unsigned r0, r1, r2, r3;
  __device__ __forceinline__ void issue(unsigned tile, int row0, int ks, int lane) {
    const int x = lane & 15, kg = lane >> 4;
    const int kb = ks * 16 + kg * 4;
    const int pos = ((row0 + x) >> 2) ^ (kg << 2);
    const unsigned a = tile + (unsigned)(kb * 256 + pos * 16 + (x & 3) * 4);
    asm volatile("ds_read_b32 %0, %1" : "=v"(r0) : "v"(a) : "memory");
    asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(r1) : "v"(a) : "memory");
    asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(r2) : "v"(a) : "memory");
    asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(r3) : "v"(a) : "memory");
  }
  __device__ __forceinline__ void tie() { asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); }
  __device__ __forceinline__ uint4 get() const { uint4 r; r.x = r0; r.y = r1; r.z = r2; r.w = r3; return r; }
};

// All-in-flight variant of the 64-tile kernel for a compile-time number of K tiles (NK = 5: K = 320, NK = 8:
// K = 512 in bf16 -- the d_model-wide products of the hot path).  A small product in the middle of a training
// step finds its weight in HBM (the parameters alone exceed the 256 MiB MALL), so a 2-deep pipeline pays the
// DRAM latency on every K tile (scripts/bench_chain.py: 5.3 us hot -> 9.0 us cold).  Here every tile has its
// own LDS stage (NK x 16 KiB), all DMAs are issued up front, and tile t is consumed as soon as each wave's own
// share of it has landed (counted vmcnt, 4 DMA instructions per tile per wave) and the barrier has published
// the other waves' shares: the DRAM latency is paid once.
// LNP (NK = 8, bf16, A rows K-contiguous): the A operand is LayerNorm(A) -- modules.py:28-31: gain * (x - mean) / (std + eps) + offset with
// the unbiased std over the K = 512 channels -- computed IN the LDS stages once all of them have landed (four threads per row, the
// arithmetic of layernorm_vec_kernel), instead of by a launch of its own in front of this one; the workgroups of column tiles 0..7
// each store one 64-channel slice of the normalised rows to ln_out (the weight-gradient product of the backward pass reads them).
template <typename T, typename TO, bool ATR, bool BTR, int NK, bool LNP = false>
__global__ __launch_bounds__(NTHREADS) void gemm_t64_pre_kernel(const GemmK g) {
  __shared__ __attribute__((aligned(16))) char l0[2 * T64_BYTES];
  __shared__ __attribute__((aligned(16))) char l1[2 * T64_BYTES];
  __shared__ __attribute__((aligned(16))) char l2[2 * T64_BYTES];
  __shared__ __attribute__((aligned(16))) char l3[2 * T64_BYTES];
  __shared__ __attribute__((aligned(16))) char l4[2 * T64_BYTES];
  __shared__ __attribute__((aligned(16))) char l5[NK > 5 ? 2 * T64_BYTES : 16];
  __shared__ __attribute__((aligned(16))) char l6[NK > 6 ? 2 * T64_BYTES : 16];
  __shared__ __attribute__((aligned(16))) char l7[NK > 7 ? 2 * T64_BYTES : 16];
  int z, sp, tm, tn;
  tile_coords(g, z, sp, tm, tn);
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const int m0 = tm * T64, n0 = tn * T64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;

  Stager64<T, ATR> sa;
  Stager64<T, BTR> sb;
  sa.init(g.A + (z1 * g.a_bs1 + z2 * g.a_bs2) * (long)sizeof(T), g.a_rs, g.a_ks, m0, g.M, 0, w, lane);
  sb.init(g.B + (z1 * g.b_bs1 + z2 * g.b_bs2) * (long)sizeof(T), g.b_rs, g.b_ks, n0, g.N, 0, w, lane);
#define PRE_IN(L_, t_) if constexpr ((t_) < NK) { sa.issue(L_, w); sb.issue(L_ + T64_BYTES, w); }
  PRE_IN(l0, 0) PRE_IN(l1, 1) PRE_IN(l2, 2) PRE_IN(l3, 3) PRE_IN(l4, 4) PRE_IN(l5, 5) PRE_IN(l6, 6) PRE_IN(l7, 7)
#undef PRE_IN
  if constexpr (LNP) {
    static_assert(NK == 8 && !ATR && sizeof(T) == 2, "LayerNorm prologue: K = 512 bf16, A rows K-contiguous");
    __shared__ __attribute__((aligned(16))) uint4 ln_ab[128];             // gain chunks 0..63, offset chunks 64..127
    uint4 gq = make_uint4(0, 0, 0, 0);
    if (tid < 128) gq = *reinterpret_cast<const uint4*>((tid < 64 ? g.ln_a : g.ln_b) + (tid & 63) * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // every K tile (this wave's share) and the gains
    if (tid < 128) ln_ab[tid] = gq;
    __syncthreads();
    auto stg = [&](int t) -> char* { return t == 0 ? l0 : t == 1 ? l1 : t == 2 ? l2 : t == 3 ? l3 : t == 4 ? l4 : t == 5 ? l5 : t == 6 ? l6 : l7; };
    const int row = tid >> 2, part = tid & 3, sw = (row >> 1) & 7;          // logical chunk pc of a row sits at physical chunk pc ^ sw; the
                                                                            // sums run over LOGICAL chunks: the same order for every row position
    uint4 q[2][8];
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) {
        q[u][pc] = *reinterpret_cast<const uint4*>(stg(part + 4 * u) + row * ROW_BYTES + ((pc ^ sw) << 4));
        const unsigned e4[4] = {q[u][pc].x, q[u][pc].y, q[u][pc].z, q[u][pc].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += __builtin_bit_cast(float, e4[e] << 16) + __builtin_bit_cast(float, e4[e] & 0xffff0000u);
      }
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
    const float mean = sum / 512.f;
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) {
        const unsigned e4[4] = {q[u][pc].x, q[u][pc].y, q[u][pc].z, q[u][pc].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __builtin_bit_cast(float, e4[e] << 16) - mean, hi = __builtin_bit_cast(float, e4[e] & 0xffff0000u) - mean;
          ss += lo * lo; ss += hi * hi;
        }
      }
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64);
    const float inv = 1.f / (sqrtf(ss / 511.f) + g.ln_eps);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) {
        const int lc = (part + 4 * u) * 8 + pc;                             // logical 8-channel chunk of the row
        const uint4 qa = ln_ab[lc], qb = ln_ab[64 + lc];
        const unsigned e4[4] = {q[u][pc].x, q[u][pc].y, q[u][pc].z, q[u][pc].w}, a4[4] = {qa.x, qa.y, qa.z, qa.w}, b4[4] = {qb.x, qb.y, qb.z, qb.w};
        unsigned o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __builtin_bit_cast(float, a4[e] << 16) * (__builtin_bit_cast(float, e4[e] << 16) - mean) * inv + __builtin_bit_cast(float, b4[e] << 16);
          const float hi = __builtin_bit_cast(float, a4[e] & 0xffff0000u) * (__builtin_bit_cast(float, e4[e] & 0xffff0000u) - mean) * inv +
                           __builtin_bit_cast(float, b4[e] & 0xffff0000u);
          const T lo16 = from_f<T>(lo), hi16 = from_f<T>(hi);
          o4[e] = (unsigned)__builtin_bit_cast(unsigned short, lo16) | ((unsigned)__builtin_bit_cast(unsigned short, hi16) << 16);
        }
        *reinterpret_cast<uint4*>(stg(part + 4 * u) + row * ROW_BYTES + ((pc ^ sw) << 4)) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
      }
    __syncthreads();
    if (tn < 8 && g.ln_out) {                                              // this workgroup's 64-channel slice of the normalised rows
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int lcl = part * 2 + u;                                       // logical chunk within the slice
        if (m0 + row < g.M)
          *reinterpret_cast<uint4*>(g.ln_out + ((long)(m0 + row) * g.ln_ld + tn * 64 + lcl * 8) * 2) =
              *reinterpret_cast<const uint4*>(stg(tn) + row * ROW_BYTES + ((lcl ^ sw) << 4));
      }
    }
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Fragment reads as inline asm (FragRd): the compiler's waitcnt pass would make the first LDS read wait for EVERY
  // DMA in flight (vmcnt(0)); with the reads opaque to it, the counted waits below are the only ones.  All reads
  // of a tile (2 K steps x (2 A + 2 B)) are issued together; the second K step's land under the first's MFMAs.
  const unsigned lds_a0 = (unsigned)(size_t)LDS_PTR(l0);
  auto compute = [&](unsigned tile_off) {
    FragRd<T, ATR> af[2][2];
    FragRd<T, BTR> bf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[ks][i].issue(lds_a0 + tile_off, wm * 32 + i * 16, ks, lane);
        bf[ks][i].issue(lds_a0 + tile_off + T64_BYTES, wn * 32 + i * 16, ks, lane);
      }
    constexpr int SECOND = 2 * FragRd<T, ATR>::NREAD + 2 * FragRd<T, BTR>::NREAD;     // reads of K step 1 still in flight
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(SECOND < 15 ? SECOND : 15) : "memory");  // the counter saturates at 15
    af[0][0].tie(); af[0][1].tie(); bf[0][0].tie(); bf[0][1].tie();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) Mma<T>::step(bf[0][j].get(), af[0][i].get(), acc[i][j]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    af[1][0].tie(); af[1][1].tie(); bf[1][0].tie(); bf[1][1].tie();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) Mma<T>::step(bf[1][j].get(), af[1][i].get(), acc[i][j]);
  };
#define PRE_USE(L_, t_)                                                              \
  if constexpr ((t_) < NK) {                                                         \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NK - 1 - (t_))) : "memory");       \
    __builtin_amdgcn_s_barrier();                                                    \
    asm volatile("" ::: "memory");                                                   \
    compute((unsigned)(size_t)LDS_PTR(L_) - (unsigned)(size_t)LDS_PTR(l0));          \
  }
  PRE_USE(l0, 0) PRE_USE(l1, 1) PRE_USE(l2, 2) PRE_USE(l3, 3) PRE_USE(l4, 4) PRE_USE(l5, 5) PRE_USE(l6, 6) PRE_USE(l7, 7)
#undef PRE_USE
  frag_out<T, TO, 2, 2>(g, acc, z1, z2, m0 + wm * 32, n0 + wn * 32, lane);
}

// Ring variant of the all-in-flight kernel for ANY number of K tiles (the K = 2048 products of the feed-forward
// blocks at M = 320: 32 tiles): 8 stages of 16 KiB in ONE LDS array, 6 tiles in flight, two tiles per barrier.  An
// iteration waits for its own share of its tile pair (counted vmcnt), the barrier publishes the others' shares AND
// proves that every wave is done with the previous pair, whose stages are then refilled with tiles t+6, t+7.  Fragment reads are inline asm (FragRd), so the
// compiler sees no LDS read of the ring and adds no waits of its own.
constexpr int RING_D = 8;
template <typename T, typename TO, bool ATR, bool BTR>
__device__ __forceinline__ void t64_ring_body(const GemmK& g, unsigned bid, unsigned nwg, char* ebuf, char* ring) {
  constexpr int D = RING_D;
  int z, sp, tm, tn;
  const int S = g.split_k;
  if (S > 1) {
    // the K slices of one tile are neighbouring workgroups of one XCD (the last one to finish reads the others' slabs)
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    unsigned lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    sp = lid % S; lid /= S;
    tn = lid % g.tiles_n; lid /= g.tiles_n;
    tm = lid % g.tiles_m; z = lid / g.tiles_m;
  } else {
    tile_coords(g, z, sp, tm, tn, bid, nwg);
  }
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const int m0 = tm * T64, n0 = tn * T64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);
  const int nk_all = (g.K + BK - 1) / BK;
  const int k_tail = g.K % BK;
  const int kt_lo = (int)((long)nk_all * sp / S), nk = (int)((long)nk_all * (sp + 1) / S) - kt_lo;      // this slice's K tiles

  Stager64<T, ATR> sa;
  Stager64<T, BTR> sb;
  sa.init(g.A + (z1 * g.a_bs1 + z2 * g.a_bs2) * (long)sizeof(T), g.a_rs, g.a_ks, m0, g.M, kt_lo * BK, w, lane);
  sb.init(g.B + (z1 * g.b_bs1 + z2 * g.b_bs2) * (long)sizeof(T), g.b_rs, g.b_ks, n0, g.N, kt_lo * BK, w, lane);
  auto stage_in = [&](int kt) {
    char* st = ring + (kt & (D - 1)) * 2 * T64_BYTES;
    if (k_tail != 0 && kt_lo + kt == nk_all - 1) { sa.issue_tail(st, w, k_tail); sb.issue_tail(st + T64_BYTES, w, k_tail); }
    else { sa.issue(st, w); sb.issue(st + T64_BYTES, w); }
  };
  for (int kt = 0; kt < D - 2 && kt < nk; ++kt) stage_in(kt);      // tiles 0..5

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned ring0 = (unsigned)(size_t)LDS_PTR(ring);
  // two tiles per barrier: the second tile's fragment reads land under the first tile's MFMAs
  for (int kt = 0; kt < nk; kt += 2) {
    const bool two = kt + 1 < nk;
    const int ahead = min(D - 4, nk - 1 - (two ? kt + 1 : kt));      // tiles issued beyond this pair (4 DMAs each per wave)
    switch (ahead) {
      case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + D - 2 < nk) stage_in(kt + D - 2);          // into the stages of the previous pair: every wave is past them
    if (kt + D - 1 < nk) stage_in(kt + D - 1);
    const unsigned t0 = ring0 + (unsigned)((kt & (D - 1)) * 2 * T64_BYTES);
    const unsigned t1 = ring0 + (unsigned)(((kt + 1) & (D - 1)) * 2 * T64_BYTES);
    FragRd<T, ATR> af[2][2][2];
    FragRd<T, BTR> bf[2][2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[0][ks][i].issue(t0, wm * 32 + i * 16, ks, lane);
        bf[0][ks][i].issue(t0 + T64_BYTES, wn * 32 + i * 16, ks, lane);
      }
    constexpr int PER_TILE = 4 * FragRd<T, ATR>::NREAD + 4 * FragRd<T, BTR>::NREAD;
    if (two) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[1][ks][i].issue(t1, wm * 32 + i * 16, ks, lane);
          bf[1][ks][i].issue(t1 + T64_BYTES, wn * 32 + i * 16, ks, lane);
        }
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PER_TILE < 15 ? PER_TILE : 15) : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      af[0][ks][0].tie(); af[0][ks][1].tie(); bf[0][ks][0].tie(); bf[0][ks][1].tie();
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<T>::step(bf[0][ks][j].get(), af[0][ks][i].get(), acc[i][j]);
    }
    if (two) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        af[1][ks][0].tie(); af[1][ks][1].tie(); bf[1][ks][0].tie(); bf[1][ks][1].tie();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) Mma<T>::step(bf[1][ks][j].get(), af[1][ks][i].get(), acc[i][j]);
      }
    }
  }
  if (S == 1) {
    frag_out<T, TO, 2, 2>(g, acc, z1, z2, m0 + wm * 32, n0 + wn * 32, lane);
    return;
  }
  __syncthreads();
  frag_to_image64(acc, reinterpret_cast<float*>(ebuf), wm, wn, lane);
  __syncthreads();
  {
    // In-launch split-K combine (cdna_hip_programming.md section 5, "Projection GEMM at M = 256", item 2): every slice writes
    // its fp32 tile image to a slab, releases at agent scope and takes a ticket; the slice that draws the last ticket acquires,
    // adds the other slabs to its own image and runs the epilogue.  The counter returns to zero for the next launch.
    const long tile = ((long)z * g.tiles_m + tm) * g.tiles_n + tn;
    int* cnt = reinterpret_cast<int*>(g.ws) + tile;
    float* slabs = g.ws + WS_HEADER + tile * S * (T64 * T64);
    float* img = reinterpret_cast<float*>(ebuf);
    int* flag = reinterpret_cast<int*>(ebuf + 2 * T64_BYTES);
    float4* mine = reinterpret_cast<float4*>(slabs + (long)sp * (T64 * T64));
#pragma unroll
    for (int q = 0; q < 4; ++q) mine[q * NTHREADS + tid] = reinterpret_cast<const float4*>(img)[q * NTHREADS + tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == S - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    float4 sum[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sum[q] = reinterpret_cast<const float4*>(img)[q * NTHREADS + tid];
    for (int s2 = 0; s2 < S; ++s2) {
      if (s2 == sp) continue;
      const float4* other = reinterpret_cast<const float4*>(slabs + (long)s2 * (T64 * T64));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 o = other[q * NTHREADS + tid];
        sum[q].x += o.x; sum[q].y += o.y; sum[q].z += o.z; sum[q].w += o.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(img)[q * NTHREADS + tid] = sum[q];
    __syncthreads();
  }
  stream_out<T, TO, false, 64>(g, ebuf, ebuf, z1, z2, 0, m0, n0, tid);
}

template <typename T, typename TO, bool ATR, bool BTR>
__global__ __launch_bounds__(NTHREADS) void gemm_t64_ring_kernel(const GemmK g) {
  __shared__ __attribute__((aligned(16))) char ebuf[2 * T64_BYTES + 64];
  __shared__ __attribute__((aligned(16))) char ring[RING_D * 2 * T64_BYTES];
  t64_ring_body<T, TO, ATR, BTR>(g, blockIdx.x, gridDim.x, ebuf, ring);
}

// The two backward products of one linear layer in ONE launch: dX = dZ.W (A K-contiguous, B row-contiguous) on the
// first n1 workgroups, dW = dZ^T.X (both row-contiguous) on the rest.  Each alone is a 40-64 workgroup launch on 256
// CUs; together they share the chip and one launch latency (bist_gemm_pair).
// Layout pairs built: PAIR_LIN  (N,T)+(T,T)  linear backward;  PAIR_FOLD (N,N)+(T,T)  per-head fold / un-fold backward;
// PAIR_FWD (N,N)+(N,N)  two forward projections (query and packed key/value of a cross-attention).
enum { PAIR_LIN = 0, PAIR_FOLD = 1, PAIR_FWD = 2 };
template <typename T, typename TO1, typename TO2, int KIND>
__global__ __launch_bounds__(NTHREADS) void gemm_t64_pair_kernel(const GemmK g1, const GemmK g2, unsigned n1) {
  __shared__ __attribute__((aligned(16))) char ebuf[2 * T64_BYTES + 64];
  __shared__ __attribute__((aligned(16))) char ring[RING_D * 2 * T64_BYTES];
  if (blockIdx.x < n1) t64_ring_body<T, TO1, false, KIND == PAIR_LIN>(g1, blockIdx.x, n1, ebuf, ring);
  else t64_ring_body<T, TO2, KIND != PAIR_FWD, KIND != PAIR_FWD>(g2, blockIdx.x - n1, gridDim.x - n1, ebuf, ring);
}

// ---------------------------------------------------------------------------------------------
// 256-column-tile kernel for the big K-contiguous products (P0, the value projection, the output projections:
// M = B*T*S rows).  The 128-tile kernel above sits at the LDS balance point of a 64x64 wave tile (32 FLOP per LDS
// byte = the CU's MFMA:LDS ratio) and stalls on the vmcnt(0) of its barrier every K step.  Here (the deep-pipelined
// structure of cdna_hip_programming.md section 5, re-derived for two phases per K tile):
//   * 8 waves (2 x 4), wave tile (FR*16) x 64 with FR row fragments: tile (FR*32) x 256 (FR = 8 is the one built);
//   * one workgroup per CU, ALL LDS in one 128 KiB array: 2 K tiles x 4 "halves" of [128 rows][128 B]:
//       Aq0 = row fragments 0-3 of both wave rows, Aq1 = fragments 4..FR-1, Bq0 / Bq1 = the first / second 32 rows of
//       each wave column's 64 B rows;
//   * a K tile is 2 phases: A = fragments 0-3 x all 4 column fragments (32 MFMAs; reads Aq0 and both B halves, 16
//     ds_read_b128), B = fragments 4..FR-1 with the B registers kept (reads Aq1);
//   * every phase stages TWO halves by LDS-DMA (4 instructions per wave) in its read segment, i.e. under the OTHER
//     wave row's MFMAs (an LDS-DMA issue stalls its wave for ~60+ cycles):  A(kt): Bq1(kt+1) Aq1(kt+1),
//     B(kt): Aq0(kt+2) Bq0(kt+2) -- each 2+ phases before its first read; the waits are vmcnt(8) / vmcnt(6), so three to
//     four halves (48-64 KiB) stay in flight across the barriers;
//   * the two wave rows run staggered by one barrier (one does MFMAs under s_setprio while the other reads LDS and
//     issues DMA): 4 barriers per K tile.  A read segment ends with lgkmcnt(0) BEFORE its barrier, so a half may be
//     restaged from the next interval on; a half is read one interval after the wait + barrier that retire it.  A
//     halves are staged and read by the same wave row, B halves by both (each wave stages 16 rows of every half);
//   * fragment reads are inline asm (the compiler adds no waits of its own), barriers are raw s_barrier;
//   * the epilogue stores straight from the accumulators (frag_out).
// ---------------------------------------------------------------------------------------------
// LayerNorm EPILOGUE of the 256-tile kernel (N = 512 = two column tiles, bf16 output): C = LayerNorm(act(alpha A.B^T + bias)) with the
// reference's LayerNorm (modules.py:28-31) over the 512 columns of a row, i.e. `self.in_norm(F.relu(self.W(fts)))` of VidEncoder8
// (encoder.py:75-81) in ONE launch instead of a GEMM plus a 2 x M x 512 x 2-byte LayerNorm pass.  A row's statistics need both column
// tiles: the two workgroups of a row block exchange their per-row (sum, sum of squares) of 256 columns through the workspace
// (write-through stores, a flag each, sc1 loads -- placement-independent; each resets the flag it polled, so the header is zero again
// when the launch ends).  One pass: var = (sum v^2 - 512 mean^2) / 511 in fp32 (post-ReLU activations of O(1): no cancellation to
// speak of).  A bounded spin: if the partner never arrives, word WS_HEADER - 1 of the header is set (sticky) and the rows are
// normalised with this tile's statistics alone.
template <int FR>
__device__ __forceinline__ void frag_out_ln(const GemmK& g, f32x4 (&acc)[FR][4], char* lds, int tm, int tn, int m0, int n0, int wr, int wc, int lane, int tid) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  typedef __attribute__((address_space(1))) unsigned gu32;
  const int lr = lane & 15, lg = lane >> 4;
  constexpr int WROWS = FR * 16;
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(g.bias);
  const bool relu = g.act == BIST_ACT_RELU;
  float2* part = reinterpret_cast<float2*>(lds);              // [2 * WROWS rows][4 wave columns]
  float2* stats = part + 2 * WROWS * 4;                       // [2 * WROWS]: (mean, 1 / (std + eps))
  // 1. alpha, bias, activation in place; 2. this wave's per-row partial sums over its 64 columns
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + lg * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
      const uint2 q = *reinterpret_cast<const uint2*>(bias + n);
      bv[0] = __builtin_bit_cast(float, q.x << 16); bv[1] = __builtin_bit_cast(float, q.x & 0xffff0000u);
      bv[2] = __builtin_bit_cast(float, q.y << 16); bv[3] = __builtin_bit_cast(float, q.y & 0xffff0000u);
    }
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] * g.alpha + bv[r];
        acc[i][j][r] = relu ? fmaxf(v, 0.f) : v;
      }
  }
#pragma unroll
  for (int i = 0; i < FR; ++i) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (lg == 0) part[(wr * WROWS + i * 16 + lr) * 4 + wc] = make_float2(s1, s2);
  }
  __syncthreads();
  // 3. this tile's sums to the partner, the partner's to us
  float* ws = g.ws;
  gu32* flag_mine = (gu32*)(ws) + (tm * 2 + tn);
  gu32* flag_other = (gu32*)(ws) + (tm * 2 + (tn ^ 1));
  gu64* x_mine = (gu64*)(ws + WS_HEADER) + (long)(tm * 2 + tn) * (2 * WROWS);
  gu64* x_other = (gu64*)(ws + WS_HEADER) + (long)(tm * 2 + (tn ^ 1)) * (2 * WROWS);
  float S1 = 0.f, S2 = 0.f;
  if (tid < 2 * WROWS) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float2 q = part[tid * 4 + c]; S1 += q.x; S2 += q.y; }
    __hip_atomic_store(x_mine + tid, ((unsigned long long)__builtin_bit_cast(unsigned, S2) << 32) | __builtin_bit_cast(unsigned, S1),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int partner_ok;
  if (tid == 0) {
    __hip_atomic_store(flag_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    bool ok = true;
    while (__hip_atomic_load(flag_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 21)) { ok = false; __hip_atomic_store((gu32*)ws + (WS_HEADER - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    if (ok) __hip_atomic_store(flag_other, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // nobody else reads the flag we polled
    partner_ok = ok;
  }
  __syncthreads();
  if (tid < 2 * WROWS) {
    float n_cols = 256.f;
    if (partner_ok) {
      const unsigned long long q = __hip_atomic_load(x_other + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S1 += __builtin_bit_cast(float, (unsigned)q); S2 += __builtin_bit_cast(float, (unsigned)(q >> 32));
      n_cols = 512.f;
    }
    const float mean = S1 / n_cols;
    const float var = fmaxf(S2 - n_cols * mean * mean, 0.f) / (n_cols - 1.f);
    stats[tid] = make_float2(mean, 1.f / (sqrtf(var) + g.ln_eps));
  }
  __syncthreads();
  // 4. normalise and store (the 16-byte row pieces of frag_rows)
  const bf16_t* ga = reinterpret_cast<const bf16_t*>(g.ln_a);
  const bf16_t* gb = reinterpret_cast<const bf16_t*>(g.ln_b);
  bf16_t* Cz = reinterpret_cast<bf16_t*>(g.C);
  const int cofs = (lg & 1) * 16 + (lg >> 1) * 8;
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const int n = n0 + wc * 64 + jp * 32 + cofs;
    const uint4 qa = *reinterpret_cast<const uint4*>(ga + n), qb = *reinterpret_cast<const uint4*>(gb + n);
    const unsigned a4[4] = {qa.x, qa.y, qa.z, qa.w}, b4[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
    for (int i = 0; i < FR; ++i) {
      const int rl = wr * WROWS + i * 16 + lr;
      const float2 st = stats[rl];
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = acc[i][2 * jp][r], b = acc[i][2 * jp + 1][r];
        swap16(a, b);
        v[r] = a; v[4 + r] = b;
      }
      unsigned o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y0 = __builtin_bit_cast(float, a4[e] << 16) * (v[2 * e] - st.x) * st.y + __builtin_bit_cast(float, b4[e] << 16);
        const float y1 = __builtin_bit_cast(float, a4[e] & 0xffff0000u) * (v[2 * e + 1] - st.x) * st.y + __builtin_bit_cast(float, b4[e] & 0xffff0000u);
        const bf16_t l16 = from_f<bf16_t>(y0), h16 = from_f<bf16_t>(y1);
        o4[e] = (unsigned)__builtin_bit_cast(unsigned short, l16) | ((unsigned)__builtin_bit_cast(unsigned short, h16) << 16);
      }
      *reinterpret_cast<uint4*>(Cz + (long)(m0 + rl) * g.ldc + n) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    }
  }
}

constexpr int BIG = 256;
constexpr int HALF_BYTES = 128 * ROW_BYTES;          // 16 KiB
constexpr int KT_BYTES = 4 * HALF_BYTES;             // one K tile: Aq0 | Bq0 | Bq1 | Aq1
enum { H_AQ0 = 0, H_BQ0 = 1, H_BQ1 = 2, H_AQ1 = 3 };

__device__ __forceinline__ void wait_dma_halves(int newer) {      // all but the `newer` most recent stagings (2 DMAs each) have landed
  switch (newer) {
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <typename TO, int FR, int V>                 // V = 1: in-kernel stamps into the workspace (scripts/stamp_big.py)
__global__ __launch_bounds__(512) void gemm_big_kernel(const GemmK g) {
  using T = bf16_t;
  constexpr int BM = FR * 32, WROWS = FR * 16, NI1 = FR - 4;
  static_assert(FR >= 5 && FR <= 8, "row fragments per wave");
  __shared__ __attribute__((aligned(16))) char lds[2 * KT_BYTES];
  int z, sp, tm, tn;
  tile_coords(g, z, sp, tm, tn);
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const int m0 = tm * BM, n0 = tn * BIG;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 2, wc = w & 3;
  const int nk = g.K / 64;

  // per-lane DMA sources: half hh, instruction j -> image rows (2w+j)*8 + lane/8, chunk swizzled on the source side
  const char* src[4][2];
  {
    const char* Ab = g.A + (z1 * g.a_bs1 + z2 * g.a_bs2) * 2;
    const char* Bb = g.B + (z1 * g.b_bs1 + z2 * g.b_bs2) * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int irow = (w * 2 + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (((j & 1) << 2) | (lane >> 4));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int arow = min(m0 + (irow >> 6) * WROWS + q * 64 + (irow & 63), g.M - 1);
        const int brow = min(n0 + (irow >> 5) * 64 + q * 32 + (irow & 31), g.N - 1);
        src[q ? H_AQ1 : H_AQ0][j] = Ab + (long)arow * g.a_rs * 2 + chunk * 16;
        src[q ? H_BQ1 : H_BQ0][j] = Bb + (long)brow * g.b_rs * 2 + chunk * 16;
      }
    }
  }
  auto stage = [&](int hh, int kt) {                   // half hh of K tile kt (pointers walk K tile by K tile)
    if (kt >= nk) return;
    char* dst = lds + (kt & 1) * KT_BYTES + hh * HALF_BYTES + w * 2048;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[hh][0]), LDS_PTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[hh][1]), LDS_PTR(dst + 1024), 16, 0, 0);
    src[hh][0] += ROW_BYTES; src[hh][1] += ROW_BYTES;
  };
  unsigned long long stamp[6], cyc0 = 0;
  if constexpr (V == 1) stamp[0] = wall_clock64();
  stage(H_AQ0, 0); stage(H_BQ0, 0); stage(H_BQ1, 0); stage(H_AQ1, 0);
  stage(H_AQ0, 1); stage(H_BQ0, 1);
  wait_dma_halves(nk > 1 ? 3 : 1);                      // Aq0(0), Bq0(0), Bq1(0) have landed ...
  __builtin_amdgcn_s_barrier();                         // ... for every wave
  if constexpr (V == 1) { stamp[1] = wall_clock64(); cyc0 = clock64(); }

  f32x4 acc[FR][4];
#pragma unroll
  for (int i = 0; i < FR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  FragRd<T, false> fa[2][4];          // [ks][i]: the A fragments of the current phase
  FragRd<T, false> fb[2][2][2];       // [q][ks][j]: both B halves, read in phase A and kept for phase B
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(lds);

  if (wr == 1) __builtin_amdgcn_s_barrier();            // stagger the two wave rows by one barrier
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned cur = lds0 + (unsigned)((kt & 1) * KT_BYTES);
    const int i1 = kt + 1 < nk, i2 = kt + 2 < nk;
    // ---- phase A: row fragments 0-3 ----
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[q][ks][j].issue(cur + (q ? H_BQ1 : H_BQ0) * HALF_BYTES, wc * 32 + j * 16, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i].issue(cur + H_AQ0 * HALF_BYTES, wr * 64 + i * 16, ks, lane);
    stage(H_BQ1, kt + 1); stage(H_AQ1, kt + 1);
    // stagings in issue order: .. Bq1(kt) Aq1(kt) | Aq0(kt+1) Bq0(kt+1) | Bq1(kt+1) Aq1(kt+1) | Aq0(kt+2) Bq0(kt+2) ..
    wait_dma_halves(4 * i1);                            // Aq1(kt), read in phase B, has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { fb[q][ks][0].tie(); fb[q][ks][1].tie(); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i].tie();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::step(fb[j >> 1][ks][j & 1].get(), fa[ks][i].get(), acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase B: row fragments 4 .. FR-1 ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NI1; ++i) fa[ks][i].issue(cur + H_AQ1 * HALF_BYTES, wr * 64 + i * 16, ks, lane);
    stage(H_AQ0, kt + 2); stage(H_BQ0, kt + 2);
    if (i1) wait_dma_halves(1 + 2 * i2);                // Aq0, Bq0, Bq1 of K tile kt+1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NI1; ++i) fa[ks][i].tie();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NI1; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::step(fb[j >> 1][ks][j & 1].get(), fa[ks][i].get(), acc[4 + i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();            // the extra barrier of the staggered group
  if constexpr (V == 1) { stamp[2] = wall_clock64(); cyc0 = clock64() - cyc0; stamp[4] = stamp[2]; }
  if constexpr (FR == 8 && V == 0 && sizeof(TO) == 2) {
    if (g.ln_mode == 1) {                              // LayerNorm epilogue (the launcher checked the envelope: whole tiles, N = 512, no dropout / residual)
      __syncthreads();                                 // every wave is done with the operand stages
      frag_out_ln<FR>(g, acc, lds, tm, tn, m0, n0, wr, wc, lane, tid);
      return;
    }
  }
  frag_out<T, TO, FR, 4>(g, acc, z1, z2, m0 + wr * WROWS, n0 + wc * 64, lane);
  if constexpr (V == 1) {
    stamp[5] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp[3] = wall_clock64();
    if (tid == 0 && g.ws) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(g.ws) + (size_t)blockIdx.x * 8;
      o[0] = stamp[0]; o[1] = stamp[1]; o[2] = stamp[2]; o[3] = stamp[3]; o[4] = stamp[4]; o[5] = stamp[5]; o[6] = cyc0;
    }
  }
}

// sums the split-K slabs and applies the epilogue: one thread per output element
template <typename T, typename TO>
__global__ void splitk_reduce_kernel(const GemmK g, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long mn = (long)g.M * g.N;
  const long zlin = idx / mn, r = idx % mn;
  const int m = (int)(r / g.N), n = (int)(r % g.N);
  const int z1 = (int)(zlin / g.batch2), z2 = (int)(zlin % g.batch2);
  const float* W = g.ws + WS_HEADER + zlin * g.split_k * mn + r;
  float acc = 0.f;
  for (int s = 0; s < g.split_k; ++s) acc += W[(long)s * mn];
  TO* C = reinterpret_cast<TO*>(g.C) + z1 * g.c_bs1 + z2 * g.c_bs2;
  const T* bias = g.bias ? reinterpret_cast<const T*>(g.bias) + z1 * g.bias_bs1 + z2 * g.bias_bs2 : nullptr;
  const TO* res = g.residual ? reinterpret_cast<const TO*>(g.residual) + z1 * g.r_bs1 + z2 * g.r_bs2 : nullptr;
  const float bv = bias ? to_f(bias[n]) : 0.f;
  C[(long)m * g.ldc + n] = from_f<TO>(epilogue_value<T, TO>(g, acc, bv, res, m, n, (unsigned long long)zlin * mn));
}

// The same for N % 4 == 0 (every product of the path): one thread per 4 consecutive columns, 16-byte slab reads with 8
// slabs in flight per thread, the row-wise epilogue of the tile kernels (RowOut).
template <typename T, typename TO>
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const GemmK g, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const long mn = (long)g.M * g.N;
  const long e0 = idx * 4;
  const long zlin = e0 / mn, r = e0 % mn;
  const int m = (int)(r / g.N), n = (int)(r % g.N);
  const int z1 = (int)(zlin / g.batch2), z2 = (int)(zlin % g.batch2);
  const float* W = g.ws + WS_HEADER + zlin * g.split_k * mn + r;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= g.split_k; s += 8) {
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(W + (long)(s + u) * mn);
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
  }
  for (; s < g.split_k; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(W + (long)s * mn);
    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
  }
  const RowOut<T, TO> out(g, z1, z2, (unsigned long long)zlin * (unsigned long long)mn);
  float bv[4];
  out.load_bias(bv, n);
  out.template row<4>(v, bv, m, n);
}

// ---------------------------------------------------------------------------------------------
// generic kernel: arbitrary element strides, any K; register staged, single buffer
// ---------------------------------------------------------------------------------------------
template <typename T, typename TO>
__global__ __launch_bounds__(NTHREADS) void gemm_gen_kernel(const GemmK g) {
  __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];
  int z, sp, tm, tn;
  tile_coords(g, z, sp, tm, tn);
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);
  constexpr int EPT = BM * BK / NTHREADS;      // elements per thread per operand per K tile

  const T* Az = reinterpret_cast<const T*>(g.A) + z1 * g.a_bs1 + z2 * g.a_bs2;
  const T* Bz = reinterpret_cast<const T*>(g.B) + z1 * g.b_bs1 + z2 * g.b_bs2;
  const bool a_kc = (g.a_ks == 1) || (g.a_rs != 1), b_kc = (g.b_ks == 1) || (g.b_rs != 1);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  T ra[EPT], rb[EPT];
  for (int k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int idx = e * NTHREADS + tid;
      int row, k;
      if (a_kc) { k = idx % BK; row = idx / BK; } else { row = idx % BM; k = idx / BM; }
      const int m = m0 + row, kk = k0 + k;
      ra[e] = (m < g.M && kk < g.K) ? Az[(long)m * g.a_rs + (long)kk * g.a_ks] : from_f<T>(0.f);
      if (b_kc) { k = idx % BK; row = idx / BK; } else { row = idx % BN; k = idx / BN; }
      const int n = n0 + row; const int kb = k0 + k;
      rb[e] = (n < g.N && kb < g.K) ? Bz[(long)n * g.b_rs + (long)kb * g.b_ks] : from_f<T>(0.f);
    }
    __syncthreads();   // previous tile's MFMAs have read the LDS image
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int idx = e * NTHREADS + tid;
      int row, k;
      if (a_kc) { k = idx % BK; row = idx / BK; } else { row = idx % BM; k = idx / BM; }
      int kb = k * (int)sizeof(T);
      *reinterpret_cast<T*>(lds + row * ROW_BYTES + ((((kb >> 4) ^ ((row >> 1) & 7)) << 4) | (kb & 15))) = ra[e];
      if (b_kc) { k = idx % BK; row = idx / BK; } else { row = idx % BN; k = idx / BN; }
      kb = k * (int)sizeof(T);
      *reinterpret_cast<T*>(lds + TILE_BYTES + row * ROW_BYTES + ((((kb >> 4) ^ ((row >> 1) & 7)) << 4) | (kb & 15))) = rb[e];
    }
    __syncthreads();
    compute_tile<T, false, false>(lds, lds + TILE_BYTES, acc, wm, wn, lane);
  }
  epilogue_direct<T, TO>(g, acc, z1, z2, 0, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// skinny products (unbatched): one side of the product is <= 8 wide, so a 128x128 MFMA tile would be >90 %
// padding.  These are the modality-fusion and pointer-switch logits (N = 1..4), their input gradients
// (K = 1..4) and their weight gradients (M = 1..4, K = rows).  All three are bandwidth/latency bound VALU
// kernels that touch each operand element once.
// ---------------------------------------------------------------------------------------------
constexpr int SKINNY = 8;

// N <= 8, both operands contiguous along K: one wave per output row, lanes stride K, wave-reduce per column.
template <typename T, typename TO, bool VEC>
__global__ __launch_bounds__(256) void skinny_n_kernel(const GemmK g) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= g.M) return;
  const T* A = reinterpret_cast<const T*>(g.A) + m * g.a_rs;
  const T* B = reinterpret_cast<const T*>(g.B);
  float acc[SKINNY];
#pragma unroll
  for (int n = 0; n < SKINNY; ++n) acc[n] = 0.f;
  if (VEC) {
    constexpr int E = 16 / (int)sizeof(T);
    for (int k = lane * E; k < g.K; k += 64 * E) {
      const uint4 qa = *reinterpret_cast<const uint4*>(A + k);
      const T* ae = reinterpret_cast<const T*>(&qa);
#pragma unroll
      for (int n = 0; n < SKINNY; ++n) {              // rows >= N re-read row N-1 (unconditional loads; never stored)
        const uint4 qb = *reinterpret_cast<const uint4*>(B + (long)min(n, g.N - 1) * g.b_rs + k);
        const T* be = reinterpret_cast<const T*>(&qb);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[n] += to_f(ae[e]) * to_f(be[e]);
      }
    }
  } else {
    for (int k = lane; k < g.K; k += 64) {
      const float a = to_f(A[k]);
#pragma unroll
      for (int n = 0; n < SKINNY; ++n) acc[n] += a * to_f(B[(long)min(n, g.N - 1) * g.b_rs + k]);
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int n = 0; n < SKINNY; ++n)
    if (n < g.N) { const float v = wave_sum(acc[n]); if (lane == n) mine = v; }
  if (lane < g.N) {
    const T* bias = reinterpret_cast<const T*>(g.bias);
    const TO* res = reinterpret_cast<const TO*>(g.residual);
    reinterpret_cast<TO*>(g.C)[m * g.ldc + lane] =
        from_f<TO>(epilogue_value<T, TO>(g, mine, bias ? to_f(bias[lane]) : 0.f, res, (int)m, lane, 0ULL));
  }
}

// K <= 8, any strides: one thread per output element (n fastest, so stores coalesce).
template <typename T, typename TO>
__global__ __launch_bounds__(256) void skinny_k_kernel(const GemmK g) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)g.M * g.N) return;
  const int m = (int)(idx / g.N), n = (int)(idx % g.N);
  const T* A = reinterpret_cast<const T*>(g.A) + (long)m * g.a_rs;
  const T* B = reinterpret_cast<const T*>(g.B) + (long)n * g.b_rs;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < SKINNY; ++k) {                  // unconditional loads from a clamped k, masked afterwards
    const int kc = min(k, g.K - 1);
    const float pr = to_f(A[kc * g.a_ks]) * to_f(B[kc * g.b_ks]);
    acc += k < g.K ? pr : 0.f;
  }
  const T* bias = reinterpret_cast<const T*>(g.bias);
  const TO* res = reinterpret_cast<const TO*>(g.residual);
  reinterpret_cast<TO*>(g.C)[(long)m * g.ldc + n] = from_f<TO>(epilogue_value<T, TO>(g, acc, bias ? to_f(bias[n]) : 0.f, res, m, n, 0ULL));
}

// One side <= 8 and a long K, the wide operand contiguous along its row index: a workgroup owns 64 wide
// indices x 16 K-lanes; every thread keeps the <= 8 partial sums in registers, LDS combines the K-lanes.
template <typename T, typename TO, bool SMALL_M>
__global__ __launch_bounds__(1024) void skinny_tn_kernel(const GemmK g) {
  __shared__ float red[16][SKINNY][64];
  const int gl = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int NS = SMALL_M ? g.M : g.N, NB = SMALL_M ? g.N : g.M;
  const T* P = reinterpret_cast<const T*>(SMALL_M ? g.B : g.A);     // wide operand
  const T* Q = reinterpret_cast<const T*>(SMALL_M ? g.A : g.B);     // narrow operand
  const long p_ks = SMALL_M ? g.b_ks : g.a_ks;
  const long q_rs = SMALL_M ? g.a_rs : g.b_rs, q_ks = SMALL_M ? g.a_ks : g.b_ks;
  const int gi = blockIdx.x * 64 + gl;
  float acc[SKINNY];
#pragma unroll
  for (int s = 0; s < SKINNY; ++s) acc[s] = 0.f;
  // unconditional loads (rows >= NS re-read row NS-1, wide indices past the end re-read the last one; their sums are
  // never stored): a per-element condition would put a branch and a full wait around every load
  const int gic = min(gi, NB - 1);
  long qoff[SKINNY];
#pragma unroll
  for (int s = 0; s < SKINNY; ++s) qoff[s] = (long)min(s, NS - 1) * q_rs;
#pragma unroll 4
  for (int k = kl; k < g.K; k += 16) {
    const float p = to_f(P[(long)k * p_ks + gic]);
#pragma unroll
    for (int s = 0; s < SKINNY; ++s) acc[s] += p * to_f(Q[qoff[s] + (long)k * q_ks]);
  }
#pragma unroll
  for (int s = 0; s < SKINNY; ++s) red[kl][s][gl] = acc[s];
  __syncthreads();
  const T* bias = reinterpret_cast<const T*>(g.bias);
  const TO* res = reinterpret_cast<const TO*>(g.residual);
  for (int idx = threadIdx.x; idx < 64 * NS; idx += 1024) {
    const int s = idx >> 6, l2 = idx & 63, g2 = blockIdx.x * 64 + l2;
    if (g2 >= NB) continue;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += red[j][s][l2];
    const int m = SMALL_M ? s : g2, n = SMALL_M ? g2 : s;
    reinterpret_cast<TO*>(g.C)[(long)m * g.ldc + n] = from_f<TO>(epilogue_value<T, TO>(g, v, bias ? to_f(bias[n]) : 0.f, res, m, n, 0ULL));
  }
}

// 0: not a skinny shape; 1: K-skinny; 2: N-skinny row reduce; 3: M-skinny; 4: N-skinny with a transposed A
int skinny_kind(const BistGemm* g) {
  if (g->batch1 != 1 || g->batch2 != 1) return 0;
  if (g->K <= SKINNY) return 1;
  if (g->N <= SKINNY && g->a_ks == 1 && g->b_ks == 1) return 2;
  if (g->M <= SKINNY && g->b_rs == 1) return 3;
  if (g->N <= SKINNY && g->a_rs == 1) return 4;
  return 0;
}

template <typename T, typename TO>
int launch_skinny(const BistGemm* g, const GemmK& k, int kind, hipStream_t st) {
  if (kind == 1) {
    const long total = (long)g->M * g->N;
    hipLaunchKernelGGL((skinny_k_kernel<T, TO>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, k);
  } else if (kind == 2) {
    const long sz = sizeof(T), E = 16 / sz;
    const bool vec = ((uintptr_t)g->A % 16 == 0) && ((uintptr_t)g->B % 16 == 0) && (g->a_rs % E == 0) && (g->b_rs % E == 0) && (g->K % E == 0);
    const dim3 grid((unsigned)((g->M + 3) / 4));
    if (vec) hipLaunchKernelGGL((skinny_n_kernel<T, TO, true>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((skinny_n_kernel<T, TO, false>), grid, dim3(256), 0, st, k);
  } else if (kind == 3) {
    hipLaunchKernelGGL((skinny_tn_kernel<T, TO, true>), dim3((unsigned)((g->N + 63) / 64)), dim3(1024), 0, st, k);
  } else {
    hipLaunchKernelGGL((skinny_tn_kernel<T, TO, false>), dim3((unsigned)((g->M + 63) / 64)), dim3(1024), 0, st, k);
  }
  BIST_LAUNCH_CHECK("bist_gemm(skinny)");
  return BIST_OK;
}

// ---- host side ------------------------------------------------------------------------------------
struct Plan { bool fast; bool atr, btr; int split; size_t ws_bytes; int stages; bool t64; };
constexpr size_t WS_HEADER_BYTES = WS_HEADER * sizeof(float);

Plan make_plan(const BistGemm* g) {
  Plan p{false, false, false, 1, 0, 2, false};
  const long sz = g->in_dtype == BIST_BF16 ? 2 : 4;
  const long bk = ROW_BYTES / sz, piece = 16 / sz;
  auto al16 = [&](long elems) { return (elems * sz) % 16 == 0; };
  auto layout_ok = [&](long rs, long ks, long rows, bool& tr) {
    if (ks == 1 && al16(rs)) { tr = false; return true; }                         // K-contiguous
    if (rs == 1 && al16(ks) && rows % piece == 0) { tr = true; return true; }     // row-contiguous
    return false;
  };
  // a K-contiguous operand is staged in 16-byte pieces along K: K must be whole pieces (any K for row-contiguous ones);
  // a trailing partial K tile is zero-filled by the stager
  const bool lay = layout_ok(g->a_rs, g->a_ks, g->M, p.atr) && layout_ok(g->b_rs, g->b_ks, g->N, p.btr);
  const bool ok = lay && ((p.atr && p.btr) || g->K % piece == 0) &&
                  al16(g->a_bs1) && al16(g->a_bs2) && al16(g->b_bs1) && al16(g->b_bs2) &&
                  ((uintptr_t)g->A % 16 == 0) && ((uintptr_t)g->B % 16 == 0);
  p.fast = ok;
  if (!ok) return p;
  const long tiles = (long)((g->M + BM - 1) / BM) * ((g->N + BN - 1) / BN) * g->batch1 * g->batch2;
  const long nk = (g->K + bk - 1) / bk;
  static const int no_t64 = [] { const char* e = getenv("BIST_GEMM_NO_T64"); return e ? atoi(e) : 0; }();      // tuning aid
  if (!no_t64 && tiles < 128 && nk <= 64) {      // cannot fill half the chip: 64x64 tiles
    p.t64 = true;
    // long K on few tiles (the K = 1536 / 2048 products at M = 320: 40 tiles): cut K over up to 4 workgroups per tile, combined
    // inside the launch by the last one to finish; slices of >= 4 K tiles, at most one round of workgroups
    // Measured (round 1): NOT a win at these sizes -- K = 2048 alone 10.9 us either way, inside a pair launch 15.2 vs 12.2 us,
    // training step 16.06 vs 15.61 ms -- the release/acquire episode and the serial slab read cost what the extra
    // workgroups save.  Kept behind BIST_GEMM_T64_SPLIT=1 / hint BIST_GEMM_SPLIT64 (tests) for larger K.
    static const int want_sk = [] { const char* e = getenv("BIST_GEMM_T64_SPLIT"); return e ? atoi(e) : 0; }();
    const bool no_sk = !(want_sk || (g->hint & BIST_GEMM_SPLIT64));
    const long t64s = (long)((g->M + 63) / 64) * ((g->N + 63) / 64) * g->batch1 * g->batch2;
    long s = nk / 4;
    if (s > 4) s = 4;
    if (s * t64s > 256) s = 256 / t64s;
    const size_t need = WS_HEADER_BYTES + (size_t)t64s * s * 64 * 64 * sizeof(float);
    if (!no_sk && nk >= 12 && s >= 2 && t64s <= WS_HEADER && g->workspace && need <= (size_t)g->workspace_bytes) { p.split = (int)s; p.ws_bytes = need; }
    return p;
  }
  if (g->workspace && tiles < 192 && nk >= 16) {
    static const int target = [] { const char* e = getenv("BIST_GEMM_SPLIT_TARGET"); return e ? atoi(e) : 512; }();   // tuning aid
    long s = (target + tiles - 1) / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s > 64) s = 64;
    const size_t need = WS_HEADER_BYTES + (size_t)s * g->M * g->N * sizeof(float) * g->batch1 * g->batch2;
    if (s > 1 && need <= (size_t)g->workspace_bytes) { p.split = (int)s; p.ws_bytes = need; }
  }
  // The 4-stage ring (128 KiB LDS, one workgroup per CU) measured SLOWER than the 2-stage kernel at two
  // workgroups per CU on every shape of this path (scripts/bench_gemm.py, round 1), so it is opt-in only.
  static const int force_stages = [] { const char* e = getenv("BIST_GEMM_STAGES"); return e ? atoi(e) : 0; }();   // tuning aid
  {
    // few workgroups (at most one per CU) and several K tiles: the paired double buffer halves the barrier chain
    const long wgs = tiles * p.split;
    const long nk_per = ((g->K + bk - 1) / bk) / p.split;
    if (wgs <= 256 && nk_per >= 3) p.stages = 22;
  }
  if (force_stages == 2 || force_stages == 4 || force_stages == 22) p.stages = force_stages;
  return p;
}

// scripts/bench_gemm_big.py (SWEEP=1 COLD=1): from ~140 tiles of 256x256 the 256-tile kernel is ahead of the 128-tile
// kernel on every K-contiguous bf16 product of the path (M = 25088: 23.5 vs 29.3 us at K = 512, 53 vs 78 us at K = 2048),
// below that the finer tiles fill the chip better.
// The LayerNorm prologue exists in gemm_t64_pre_kernel<NK = 8> only: bf16, K = 512, A rows K-contiguous and 16-byte aligned, unbatched,
// a launch of at most 256 64x64 tiles (one per CU), N >= 512 so that column tiles 0..7 exist to write ln_out.
bool ln_fusable(const BistGemm* g, const Plan& p) {
  if (!(g->ln_gain && g->ln_offset) || g->ln_mode != 0) return false;
  static const int off = [] { const char* e = getenv("BIST_GEMM_NO_LN"); return e ? atoi(e) : 0; }();      // tuning aid
  const long t64s = (long)((g->M + 63) / 64) * ((g->N + 63) / 64);
  return !off && g->in_dtype == BIST_BF16 && g->K == 512 && g->batch1 == 1 && g->batch2 == 1 && p.fast && p.t64 && !p.atr && p.split == 1 &&
         t64s <= 256 && g->N >= 512 && g->N % 64 == 0 && !skinny_kind(g) && ((uintptr_t)g->ln_gain % 16 == 0) && ((uintptr_t)g->ln_offset % 16 == 0) &&
         (!g->ln_out || (((uintptr_t)g->ln_out % 16 == 0) && g->ln_ld % 8 == 0 && g->ln_ld >= 512));
}

// The LayerNorm EPILOGUE exists in the 256-tile kernel only: bf16 in and out, N = 512 (two column tiles), whole 256-row tiles, unbatched,
// K-contiguous operands, no dropout / residual, a workspace for the exchange of the row statistics (header flags: at most 512 row blocks).
bool ln_epilogue_ok(const BistGemm* g, const Plan& p) {
  if (!(g->ln_gain && g->ln_offset) || g->ln_mode != 1) return false;
  static const int off = [] { const char* e = getenv("BIST_GEMM_NO_LN"); return e ? atoi(e) : 0; }();
  const long blocks = g->M / 256;
  return !off && g->in_dtype == BIST_BF16 && g->out_dtype == BIST_BF16 && g->N == 512 && g->M % 256 == 0 && blocks >= 1 && 2 * blocks < WS_HEADER - 1 &&
         g->batch1 == 1 && g->batch2 == 1 && p.fast && !p.atr && !p.btr && g->K % 64 == 0 && g->K >= 128 && g->drop_p == 0.f && !g->residual &&
         g->ldc % 8 == 0 && ((uintptr_t)g->C % 16 == 0) && ((uintptr_t)g->ln_gain % 16 == 0) && ((uintptr_t)g->ln_offset % 16 == 0) &&
         (!g->bias || (uintptr_t)g->bias % 8 == 0) && g->workspace &&
         (size_t)g->workspace_bytes >= WS_HEADER_BYTES + (size_t)blocks * 2 * 256 * 8;
}

bool use_tile256(const BistGemm* g, const Plan& p) {
  if (g->ln_gain && g->ln_mode == 1) return ln_epilogue_ok(g, p);
  if (g->ln_gain) return false;
  if (g->in_dtype != BIST_BF16) return false;
  const long big_tiles = (long)((g->M + BIG - 1) / BIG) * ((g->N + BIG - 1) / BIG) * g->batch1 * g->batch2;
  const bool legal = p.fast && !p.atr && !p.btr && g->K % 64 == 0 && g->K >= 128;
  static const int no_big = [] { const char* e = getenv("BIST_GEMM_NO_BIG"); return e ? atoi(e) : 0; }();    // tuning aid
  const bool wanted = (g->hint & 15) == BIST_GEMM_TILE256 || (!no_big && big_tiles >= 140 && g->K >= 512 && p.split == 1);
  return legal && wanted;
}

template <typename T, typename TO>
int launch(const BistGemm* g, GemmK& k, hipStream_t st) {
  if (k.ln_a && !(k.ln_mode == 1 ? ln_epilogue_ok(g, make_plan(g)) : ln_fusable(g, make_plan(g)))) {
    bist_set_error("bist_gemm: LayerNorm %s outside its envelope (bist_gemm_ln_ok)", k.ln_mode == 1 ? "epilogue" : "prologue");
    return BIST_EINVAL;
  }
  if (const int sk = skinny_kind(g)) return launch_skinny<T, TO>(g, k, sk, st);
  const Plan p = make_plan(g);
  k.split_k = p.split;
  k.ws = p.split > 1 ? (float*)g->workspace : nullptr;
  const long nwg = (long)k.tiles_m * k.tiles_n * g->batch1 * g->batch2 * p.split;
  if (nwg >= (1L << 31)) { bist_set_error("bist_gemm: grid too large"); return BIST_EINVAL; }
  const dim3 grid((unsigned)nwg), block(NTHREADS);
#define FAST(ATR_, BTR_)                                                                                         \
  do {                                                                                                           \
    if (p.stages == 4) hipLaunchKernelGGL((gemm_fast_kernel<T, TO, ATR_, BTR_, 4>), grid, block, 0, st, k);      \
    else if (p.stages == 22) hipLaunchKernelGGL((gemm_fast_kernel<T, TO, ATR_, BTR_, 22>), grid, block, 0, st, k); \
    else hipLaunchKernelGGL((gemm_fast_kernel<T, TO, ATR_, BTR_, 2>), grid, block, 0, st, k);                    \
  } while (0)
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (use_tile256(g, p)) {
      // Row fragments per wave: tile rows = 32 * FR.  Built: FR = 8 (256 rows) and FR = 5 (160 rows: the batched score
      // products have M = Lq*h = 160 rows per clip, 62 % of a 256-row tile); the one with fewer padded rows is used, FR = 8 on a
      // tie.  (FR = 7 puts M = 25088 on 224 instead of 196 CUs with 1/8 less work each and measured no faster -- 51.9 vs
      // 51.3 us in-kernel at K = 2048: the K loop is bound by the L2 -> LDS stream, which more active CUs only load further.)
      const long pad8 = (long)((g->M + 255) / 256) * 256, pad5 = (long)((g->M + 159) / 160) * 160;
      const bool stamps = (g->hint & 15) == BIST_GEMM_TILE256 && (g->hint >> 4) == 1;       // development aid (FR = 8 only)
      int fr = (!stamps && pad5 < pad8) ? 5 : 8;
      if (k.ln_mode == 1) fr = 8;                      // the LayerNorm epilogue is built for 256-row tiles
      static const int force_fr = [] { const char* e = getenv("BIST_GEMM_BIG_FR"); return e ? atoi(e) : 0; }();      // tuning aid
      if (!stamps && force_fr >= 5 && force_fr <= 8 && k.ln_mode != 1) fr = force_fr;
      k.tiles_m = (g->M + 32 * fr - 1) / (32 * fr); k.tiles_n = (g->N + BIG - 1) / BIG;
      k.split_k = 1; k.ws = k.ln_mode == 1 ? (float*)g->workspace : nullptr;
      // Wave quantisation: one tile per CU per round, so 784 tiles (P0 at B = 64: 392 row blocks x 2) are 3 full rounds plus a
      // fourth with 16 of 256 CUs busy.  When at least two rounds are full and the last one would be less than 1/8 full, its row
      // blocks leave this launch and go through bist_gemm as a product of their own (few rows: finer tiles / split K fill the
      // chip).  Only for products whose epilogue does not depend on the global row index (no dropout, no row-mapped residual).
      static const int no_tail = [] { const char* e = getenv("BIST_GEMM_NO_TAIL"); return e ? atoi(e) : 0; }();      // tuning aid
      const long tiles_all = (long)k.tiles_m * k.tiles_n;
      const long tail_tiles = tiles_all % 256;
      if (!no_tail && !stamps && g->batch1 == 1 && g->batch2 == 1 && tiles_all >= 512 && tail_tiles > 0 && tail_tiles <= 32 &&
          tail_tiles % k.tiles_n == 0 && g->drop_p == 0.f && g->res_outer == 0 && g->M % (32 * fr) == 0) {
        const int tail_blocks = (int)(tail_tiles / k.tiles_n);
        const long rows_main = (long)(k.tiles_m - tail_blocks) * 32 * fr;
        BistGemm tail = *g;
        const long si = 2, so = g->out_dtype == BIST_BF16 ? 2 : 4;
        tail.A = (const char*)g->A + rows_main * g->a_rs * si;
        tail.C = (char*)g->C + rows_main * g->ldc * so;
        if (g->residual) tail.residual = (const char*)g->residual + rows_main * g->ldr * so;
        tail.M = g->M - (int)rows_main;
        tail.hint = 0;
        tail.ln_gain = tail.ln_offset = nullptr; tail.ln_out = nullptr; tail.ln_mode = 0;
        if (const int rc = bist_gemm(&tail, (void*)st)) return rc;
        if (k.ln_mode == 1) {                          // the shed rows get their LayerNorm as a (small) launch of its own, in place
          if (const int rc = bist_layernorm_fwd(tail.C, g->ln_gain, g->ln_offset, tail.C, tail.M, g->N, g->ldc, g->ldc, g->ln_eps, BIST_BF16, (void*)st)) return rc;
        }
        k.M = (int)rows_main;
        k.tiles_m -= tail_blocks;
      }
      const dim3 gb((unsigned)((long)k.tiles_m * k.tiles_n * g->batch1 * g->batch2));
      if (stamps) {                                  // in-kernel stamps into the workspace
        k.ws = (float*)g->workspace;
        hipLaunchKernelGGL((gemm_big_kernel<TO, 8, 1>), gb, dim3(512), 0, st, k);
      } else if (fr == 5) {
        hipLaunchKernelGGL((gemm_big_kernel<TO, 5, 0>), gb, dim3(512), 0, st, k);
      } else if (fr == 6) {
        hipLaunchKernelGGL((gemm_big_kernel<TO, 6, 0>), gb, dim3(512), 0, st, k);
      } else if (fr == 7) {
        hipLaunchKernelGGL((gemm_big_kernel<TO, 7, 0>), gb, dim3(512), 0, st, k);
      } else {
        hipLaunchKernelGGL((gemm_big_kernel<TO, 8, 0>), gb, dim3(512), 0, st, k);
      }
      BIST_LAUNCH_CHECK("bist_gemm(256-tile)");
      return BIST_OK;
    }
  }
  if (p.fast && p.t64) {
    k.tiles_m = (g->M + T64 - 1) / T64; k.tiles_n = (g->N + T64 - 1) / T64;
    const dim3 g64((unsigned)((long)k.tiles_m * k.tiles_n * g->batch1 * g->batch2));
    const long bk64 = ROW_BYTES / (long)sizeof(T);
    const long nk64 = (g->K + bk64 - 1) / bk64;
    const bool pair = nk64 >= 12;                               // long K: halve the barrier chain (64 KiB of LDS)
    static const int no_pre = [] { const char* e = getenv("BIST_GEMM_NO_PRE"); return e ? atoi(e) : 0; }();      // tuning aid
    const bool whole_cu = g64.x <= 256 && !no_pre;                        // one workgroup per CU: 128+ KiB of LDS is free
    const bool whole = g->K % bk64 == 0 && whole_cu;
    const dim3 g64s((unsigned)(g64.x * p.split));                 // in-launch split-K: `split` neighbouring workgroups per tile
    if (k.ln_a) {                                                  // LayerNorm prologue: only the all-in-flight K = 512 kernel has it
      if constexpr (std::is_same<T, bf16_t>::value) {
        if (!(ln_fusable(g, p) && whole && nk64 == 8)) { bist_set_error("bist_gemm: LayerNorm prologue outside its envelope (bist_gemm_ln_ok)"); return BIST_EINVAL; }
        if (p.btr) hipLaunchKernelGGL((gemm_t64_pre_kernel<T, TO, false, true, 8, true>), g64, block, 0, st, k);
        else hipLaunchKernelGGL((gemm_t64_pre_kernel<T, TO, false, false, 8, true>), g64, block, 0, st, k);
        BIST_LAUNCH_CHECK("bist_gemm(64-tile, LayerNorm prologue)");
        return BIST_OK;
      } else { bist_set_error("bist_gemm: LayerNorm prologue needs bf16 operands"); return BIST_EINVAL; }
    }
#define GO64(ATR_, BTR_)                                                                                  \
  do {                                                                                                    \
    if (p.split > 1) hipLaunchKernelGGL((gemm_t64_ring_kernel<T, TO, ATR_, BTR_>), g64s, block, 0, st, k);             \
    else if (whole && nk64 == 8) hipLaunchKernelGGL((gemm_t64_pre_kernel<T, TO, ATR_, BTR_, 8>), g64, block, 0, st, k);      \
    else if (whole && nk64 == 5) hipLaunchKernelGGL((gemm_t64_pre_kernel<T, TO, ATR_, BTR_, 5>), g64, block, 0, st, k); \
    else if (whole_cu && nk64 > 8) hipLaunchKernelGGL((gemm_t64_ring_kernel<T, TO, ATR_, BTR_>), g64, block, 0, st, k); \
    else if (pair) hipLaunchKernelGGL((gemm_t64_kernel<T, TO, ATR_, BTR_, true>), g64, block, 0, st, k);       \
    else hipLaunchKernelGGL((gemm_t64_kernel<T, TO, ATR_, BTR_, false>), g64, block, 0, st, k);           \
  } while (0)
    if (!p.atr && !p.btr) GO64(false, false);
    else if (!p.atr && p.btr) GO64(false, true);
    else if (p.atr && !p.btr) GO64(true, false);
    else GO64(true, true);
#undef GO64
    BIST_LAUNCH_CHECK("bist_gemm(64-tile)");
    return BIST_OK;
  }
  if (!p.fast) {
    static const bool trace = getenv("BIST_GEMM_TRACE") != nullptr;      // development aid: which shapes miss the fast paths
    if (trace) fprintf(stderr, "bist_gemm generic: M=%d N=%d K=%d batch=%dx%d a=(%ld,%ld) b=(%ld,%ld) ldc=%ld in=%d out=%d\n", g->M, g->N, g->K,
                       g->batch1, g->batch2, g->a_rs, g->a_ks, g->b_rs, g->b_ks, g->ldc, g->in_dtype, g->out_dtype);
    hipLaunchKernelGGL((gemm_gen_kernel<T, TO>), grid, block, 0, st, k);
  }
  else if (!p.atr && !p.btr) FAST(false, false);
  else if (!p.atr && p.btr) FAST(false, true);
  else if (p.atr && !p.btr) FAST(true, false);
  else FAST(true, true);
#undef FAST
  BIST_LAUNCH_CHECK("bist_gemm");
  if (p.split > 1) {
    const long total = (long)g->M * g->N * g->batch1 * g->batch2;
    if (g->N % 4 == 0) hipLaunchKernelGGL((splitk_reduce4_kernel<T, TO>), dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, k, total / 4);
    else hipLaunchKernelGGL((splitk_reduce_kernel<T, TO>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, k, total);
    BIST_LAUNCH_CHECK("bist_gemm(split-K reduce)");
  }
  return BIST_OK;
}

}  // namespace

extern "C" int bist_gemm_ln_ok(const BistGemm* g) {
  if (!g || !g->ln_gain) return 0;
  const Plan p = make_plan(g);
  if (g->ln_mode == 1) return ln_epilogue_ok(g, p) ? 1 : 0;
  static const int no_pre = [] { const char* e = getenv("BIST_GEMM_NO_PRE"); return e ? atoi(e) : 0; }();
  return ln_fusable(g, p) && !no_pre ? 1 : 0;
}

extern "C" int bist_gemm_is_fast(const BistGemm* g) {
  if (!g) return 0;
  if (skinny_kind(g)) return 3;
  const Plan p = make_plan(g);
  if (use_tile256(g, p)) return 4;
  return p.fast ? (p.split > 1 ? 2 : 1) : 0;
}

namespace {
int fill_gemmk(const BistGemm* g, GemmK& k) {

  BIST_REQUIRE(g != nullptr, "bist_gemm: null descriptor");
  BIST_REQUIRE(g->A && g->B && g->C, "bist_gemm: null operand");
  BIST_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "bist_gemm: bad shape M=%d N=%d K=%d", g->M, g->N, g->K);
  BIST_REQUIRE(g->batch1 > 0 && g->batch2 > 0, "bist_gemm: bad batch %d x %d", g->batch1, g->batch2);
  BIST_REQUIRE(g->in_dtype == BIST_F32 || g->in_dtype == BIST_BF16, "bist_gemm: bad in_dtype %d", g->in_dtype);
  BIST_REQUIRE(g->out_dtype == BIST_F32 || g->out_dtype == g->in_dtype, "bist_gemm: out_dtype must be f32 or equal in_dtype");
  BIST_REQUIRE(g->drop_p >= 0.f && g->drop_p < 1.f, "bist_gemm: drop_p out of range");
  BIST_REQUIRE((g->res_outer == 0) == (g->res_inner == 0) && g->res_outer >= 0, "bist_gemm: bad residual row map");
  BIST_REQUIRE(g->act == BIST_ACT_NONE || g->act == BIST_ACT_RELU || (g->act == BIST_ACT_GATE && g->residual && g->drop_p == 0.f),
               "bist_gemm: bad act %d (the gate needs a residual operand and no dropout)", (int)g->act);
  BIST_REQUIRE(g->workspace_bytes >= 0 && (g->workspace || g->workspace_bytes == 0), "bist_gemm: bad workspace");
  k.A = (const char*)g->A; k.B = (const char*)g->B; k.C = (char*)g->C;
  k.bias = (const char*)g->bias; k.residual = (const char*)g->residual;
  k.M = g->M; k.N = g->N; k.K = g->K;
  k.a_rs = g->a_rs; k.a_ks = g->a_ks; k.b_rs = g->b_rs; k.b_ks = g->b_ks; k.ldc = g->ldc; k.ldr = g->ldr;
  k.batch2 = g->batch2;
  k.a_bs1 = g->a_bs1; k.a_bs2 = g->a_bs2; k.b_bs1 = g->b_bs1; k.b_bs2 = g->b_bs2;
  k.c_bs1 = g->c_bs1; k.c_bs2 = g->c_bs2; k.r_bs1 = g->r_bs1; k.r_bs2 = g->r_bs2; k.bias_bs1 = g->bias_bs1; k.bias_bs2 = g->bias_bs2;
  k.alpha = g->alpha; k.act = g->act; k.res_outer = g->res_outer; k.res_inner = g->res_inner;
  k.drop_p = g->drop_p; k.drop_seed = g->drop_seed; k.drop_ctr = (const unsigned long long*)g->drop_ctr;
  k.tiles_m = (g->M + BM - 1) / BM; k.tiles_n = (g->N + BN - 1) / BN;
  k.split_k = 1; k.ws = nullptr;
  BIST_REQUIRE((g->ln_gain == nullptr) == (g->ln_offset == nullptr), "bist_gemm: LayerNorm prologue needs gain and offset");
  k.ln_a = (const char*)g->ln_gain; k.ln_b = (const char*)g->ln_offset; k.ln_out = (char*)g->ln_out; k.ln_ld = g->ln_ld; k.ln_eps = g->ln_eps;
  k.ln_mode = g->ln_gain ? g->ln_mode : 0;
  BIST_REQUIRE(k.ln_mode == 0 || k.ln_mode == 1, "bist_gemm: ln_mode must be 0 (prologue) or 1 (epilogue)");
  { static const int dbg = [] { const char* e = getenv("BIST_GEMM_DBG"); return e ? atoi(e) : 0; }(); k.dbg = dbg; }
  {
    const long so = g->out_dtype == BIST_BF16 ? 2 : 4;
    auto al = [&](long elems) { return (elems * so) % 16 == 0; };
    k.vec_c = ((uintptr_t)g->C % 16 == 0) && al(g->ldc) && al(g->c_bs1) && al(g->c_bs2);
    k.vec_r = g->residual && ((uintptr_t)g->residual % 16 == 0) && al(g->ldr) && al(g->r_bs1) && al(g->r_bs2);
  }
  return BIST_OK;
}
}  // namespace

extern "C" int bist_gemm(const BistGemm* g, void* stream) {
  GemmK k;
  if (const int rc = fill_gemmk(g, k)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (g->in_dtype == BIST_BF16) {
    if (g->out_dtype == BIST_BF16) return launch<bf16_t, bf16_t>(g, k, st);
    return launch<bf16_t, float>(g, k, st);
  }
  return launch<float, float>(g, k, st);
}

namespace {
template <typename T, typename TO1, typename TO2>
int launch_pair(const BistGemm* a, const BistGemm* b, GemmK& ka, GemmK& kb, int kind, hipStream_t st) {
  ka.tiles_m = (a->M + T64 - 1) / T64; ka.tiles_n = (a->N + T64 - 1) / T64;        // ka.split_k / ka.ws: set by the caller
  kb.tiles_m = (b->M + T64 - 1) / T64; kb.tiles_n = (b->N + T64 - 1) / T64; kb.split_k = 1; kb.ws = nullptr;
  const long n1 = (long)ka.tiles_m * ka.tiles_n * a->batch1 * a->batch2 * ka.split_k, n2 = (long)kb.tiles_m * kb.tiles_n * b->batch1 * b->batch2;
  const dim3 grid((unsigned)(n1 + n2));
  if (kind == PAIR_LIN) hipLaunchKernelGGL((gemm_t64_pair_kernel<T, TO1, TO2, PAIR_LIN>), grid, dim3(NTHREADS), 0, st, ka, kb, (unsigned)n1);
  else if (kind == PAIR_FOLD) hipLaunchKernelGGL((gemm_t64_pair_kernel<T, TO1, TO2, PAIR_FOLD>), grid, dim3(NTHREADS), 0, st, ka, kb, (unsigned)n1);
  else hipLaunchKernelGGL((gemm_t64_pair_kernel<T, TO1, TO2, PAIR_FWD>), grid, dim3(NTHREADS), 0, st, ka, kb, (unsigned)n1);
  BIST_LAUNCH_CHECK("bist_gemm_pair");
  return BIST_OK;
}
}  // namespace

extern "C" int bist_gemm_pair(const BistGemm* a, const BistGemm* b, void* stream) {
  BIST_REQUIRE(a && b, "bist_gemm_pair: null descriptor");
  BIST_REQUIRE(!a->ln_gain && !b->ln_gain, "bist_gemm_pair: no LayerNorm prologue in the paired launch");
  GemmK ka, kb;
  if (const int rc = fill_gemmk(a, ka)) return rc;
  if (const int rc = fill_gemmk(b, kb)) return rc;
  hipStream_t st = (hipStream_t)stream;
  static const int no_pair = [] { const char* e = getenv("BIST_GEMM_NO_PAIR"); return e ? atoi(e) : 0; }();      // tuning aid
  const Plan pa = make_plan(a), pb = make_plan(b);
  // one launch when both are small 64-tile products in one of the built layout pairs
  int kind = -1;
  if (!pa.atr && pa.btr && pb.atr && pb.btr) kind = PAIR_LIN;
  else if (!pa.atr && !pa.btr && pb.atr && pb.btr) kind = PAIR_FOLD;
  else if (!pa.atr && !pa.btr && !pb.atr && !pb.btr) kind = PAIR_FWD;
  // ... and together fit one round of workgroups (144 KiB of LDS each: one per CU); beyond that two launches are faster
  // the first product may be cut along K (make_plan: long K on few tiles); its slices count as workgroups
  const long wgs = (long)((a->M + T64 - 1) / T64) * ((a->N + T64 - 1) / T64) * a->batch1 * a->batch2 * pa.split +
                   (long)((b->M + T64 - 1) / T64) * ((b->N + T64 - 1) / T64) * b->batch1 * b->batch2;
  ka.split_k = pa.split; ka.ws = pa.split > 1 ? (float*)a->workspace : nullptr;
  const bool fused = !no_pair && kind >= 0 && a->in_dtype == b->in_dtype && !skinny_kind(a) && !skinny_kind(b) && pa.fast && pb.fast &&
                     pa.t64 && pb.t64 && pb.split == 1 && wgs <= 256;
  if (!fused) {
    if (const int rc = bist_gemm(a, stream)) return rc;
    return bist_gemm(b, stream);
  }
  const bool bf = a->in_dtype == BIST_BF16;
  const bool o1 = a->out_dtype == BIST_F32 && bf, o2 = b->out_dtype == BIST_F32 && bf;      // f32 output from bf16 operands
  if (!bf) return launch_pair<float, float, float>(a, b, ka, kb, kind, st);
  if (!o1 && !o2) return launch_pair<bf16_t, bf16_t, bf16_t>(a, b, ka, kb, kind, st);
  if (!o1 && o2) return launch_pair<bf16_t, bf16_t, float>(a, b, ka, kb, kind, st);
  if (o1 && !o2) return launch_pair<bf16_t, float, bf16_t>(a, b, ka, kb, kind, st);
  return launch_pair<bf16_t, float, float>(a, b, ka, kb, kind, st);
}
