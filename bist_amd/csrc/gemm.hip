// GEMM for the BiST hot path on gfx950:  C = epilogue(alpha * A . B^T), fp32 accumulate on MFMA.
//
// One 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA
// fragments of 16x16).  LDS holds the A and B tiles as [128 rows][128 bytes] (64 bf16 or 32 f32
// along K), 16-byte chunks XOR-swizzled by ((row>>1)&7) so that the ds_read_b128 fragment reads
// are bank-conflict free.  The byte geometry is identical for both dtypes: a lane's 16-byte
// fragment read feeds one v_mfma_f32_16x16x32_bf16 (8 bf16) or four v_mfma_f32_16x16x4_f32
// (4 f32, one per instruction; A and B use the same k permutation so the dot product is intact).
//
// Two loaders share the compute and the epilogue:
//   fast: K-contiguous 16-byte-aligned operands, K % (128/sizeof T) == 0 -> LDS-DMA
//         (global_load_lds_dwordx4: 8 rows x 128 B per wave instruction, the swizzle applied on
//         the per-lane SOURCE address, LDS image lane-linear), double-buffered, one barrier per
//         K tile with the next tile's DMA in flight under the MFMAs;
//   gen : any element strides / any K (zero-filled tails), register staged, single buffer.
//
// Workgroup ids are remapped so that each XCD (blocks b, b+8, ... share one) walks a contiguous
// range of tiles: neighbouring tiles share an A row panel, which then stays in that XCD's L2.
#include "common.hpp"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;                 // K extent of a tile in bytes
constexpr int TILE_BYTES = BM * ROW_BYTES;     // 16 KiB per operand per stage
constexpr int NTHREADS = 256;

struct GemmK {   // device-side argument block (by value)
  const char* A; const char* B; char* C; const char* bias; const char* residual;
  int M, N, K;
  long a_rs, a_ks, b_rs, b_ks, ldc, ldr;
  int batch2;
  long a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2, r_bs1, r_bs2, bias_bs2;
  float alpha; int act; int res_outer, res_inner;
  float drop_p; unsigned long long drop_seed;
  int tiles_m, tiles_n;
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

// tile id -> (z, tm, tn) with the XCD-contiguous remap (bijective for any grid size)
__device__ __forceinline__ void tile_coords(const GemmK& g, int& z, int& tm, int& tn) {
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
  const unsigned lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  tn = lid % g.tiles_n;
  const unsigned t2 = lid / g.tiles_n;
  tm = t2 % g.tiles_m;
  z = t2 / g.tiles_m;
}

// MFMAs of one K tile: lds_a / lds_b point at the [128][128 B] swizzled images.
template <typename T>
__device__ __forceinline__ void compute_tile(const char* lds_a, const char* lds_b, f32x4 (&acc)[4][4], int wm, int wn, int lane) {
  const int lr = lane & 15, lg = lane >> 4, sw = lr >> 1;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int off = ((ks * 4 + lg) ^ sw) << 4;
    uint4 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = *reinterpret_cast<const uint4*>(lds_a + (wm * 64 + i * 16 + lr) * ROW_BYTES + off);
      bf[i] = *reinterpret_cast<const uint4*>(lds_b + (wn * 64 + i * 16 + lr) * ROW_BYTES + off);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Mma<T>::step(af[i], bf[j], acc[i][j]);
  }
}

template <typename T, typename TO>
__device__ __forceinline__ void epilogue(const GemmK& g, f32x4 (&acc)[4][4], int z1, int z2, int m0, int n0, int wm, int wn, int lane) {
  const int lr = lane & 15, lg = lane >> 4;
  TO* C = reinterpret_cast<TO*>(g.C) + z1 * g.c_bs1 + z2 * g.c_bs2;
  const T* bias = g.bias ? reinterpret_cast<const T*>(g.bias) + z2 * g.bias_bs2 : nullptr;
  const TO* res = g.residual ? reinterpret_cast<const TO*>(g.residual) + z1 * g.r_bs1 + z2 * g.r_bs2 : nullptr;
  const bool drop = g.drop_p > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - g.drop_p) : 1.f;
  const unsigned long long zoff = (unsigned long long)(z1 * (long)g.batch2 + z2) * (unsigned long long)g.M * (unsigned long long)g.N;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + lr;
    if (n >= g.N) continue;
    const float bv = bias ? to_f(bias[n]) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + lg * 4 + r;
        if (m >= g.M) continue;
        float v = acc[i][j][r] * g.alpha + bv;
        if (g.act == BIST_ACT_RELU) v = fmaxf(v, 0.f);
        if (drop) v = drop_keep(g.drop_seed, zoff + (unsigned long long)m * g.N + n, g.drop_p) ? v * keep_scale : 0.f;
        if (res) {
          const long rr = g.res_outer > 0 ? (long)(m / g.res_outer) * g.res_inner + (m % g.res_inner) : (long)m;
          v += to_f(res[rr * g.ldr + n]);
        }
        C[(long)m * g.ldc + n] = from_f<TO>(v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fast kernel: LDS-DMA staging, double buffer
// ---------------------------------------------------------------------------------------------
template <typename T, typename TO>
__global__ __launch_bounds__(NTHREADS) void gemm_fast_kernel(const GemmK g) {
  // one LDS object per stage: the compiler tracks in-flight LDS-DMA per object, so the
  // fragment reads of stage s need not wait for the DMA that is filling stage s^1
  __shared__ __attribute__((aligned(16))) char lds0[2 * TILE_BYTES];   // [A|B]
  __shared__ __attribute__((aligned(16))) char lds1[2 * TILE_BYTES];
  int z, tm, tn;
  tile_coords(g, z, tm, tn);
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);

  const char* Az = g.A + (z1 * g.a_bs1 + z2 * g.a_bs2) * (long)sizeof(T);
  const char* Bz = g.B + (z1 * g.b_bs1 + z2 * g.b_bs2) * (long)sizeof(T);
  // per-lane source pointers: wave w stages row groups w*4+j (8 rows each) of both operands
  const char* pa[4];
  const char* pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (w * 4 + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (((j & 1) << 2) | (lane >> 4));     // inverse of the read swizzle
    const int ma = min(m0 + row, g.M - 1), nb = min(n0 + row, g.N - 1);  // tails re-read a valid row
    pa[j] = Az + (long)ma * g.a_rs * (long)sizeof(T) + chunk * 16;
    pb[j] = Bz + (long)nb * g.b_rs * (long)sizeof(T) + chunk * 16;
  }
  auto issue = [&](char* stage_base) {
    char* la = stage_base + (w * 4) * 1024;
    char* lb = la + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(pa[j]), LDS_PTR(la + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(pb[j]), LDS_PTR(lb + j * 1024), 16, 0, 0);
      pa[j] += ROW_BYTES;
      pb[j] += ROW_BYTES;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  issue(lds0);
  for (int kt = 0; kt < nk; kt += 2) {
    __syncthreads();                     // waits vmcnt(0): tile kt has landed, tile kt-1 fully consumed
    if (kt + 1 < nk) issue(lds1);
    compute_tile<T>(lds0, lds0 + TILE_BYTES, acc, wm, wn, lane);
    if (kt + 1 < nk) {
      __syncthreads();
      if (kt + 2 < nk) issue(lds0);
      compute_tile<T>(lds1, lds1 + TILE_BYTES, acc, wm, wn, lane);
    }
  }
  epilogue<T, TO>(g, acc, z1, z2, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// generic kernel: arbitrary element strides, any K; register staged, single buffer
// ---------------------------------------------------------------------------------------------
template <typename T, typename TO>
__global__ __launch_bounds__(NTHREADS) void gemm_gen_kernel(const GemmK g) {
  __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];
  int z, tm, tn;
  tile_coords(g, z, tm, tn);
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
    compute_tile<T>(lds, lds + TILE_BYTES, acc, wm, wn, lane);
  }
  epilogue<T, TO>(g, acc, z1, z2, m0, n0, wm, wn, lane);
}

bool fast_ok(const BistGemm* g) {
  const long sz = g->in_dtype == BIST_BF16 ? 2 : 4;
  const long bk = ROW_BYTES / sz;
  auto al16 = [&](long elems) { return (elems * sz) % 16 == 0; };
  return g->a_ks == 1 && g->b_ks == 1 && g->K % bk == 0 && al16(g->a_rs) && al16(g->b_rs) && al16(g->a_bs1) &&
         al16(g->a_bs2) && al16(g->b_bs1) && al16(g->b_bs2) && ((uintptr_t)g->A % 16 == 0) && ((uintptr_t)g->B % 16 == 0);
}

template <typename T, typename TO>
int launch(const BistGemm* g, const GemmK& k, hipStream_t st) {
  const long nwg = (long)k.tiles_m * k.tiles_n * g->batch1 * g->batch2;
  if (fast_ok(g))
    hipLaunchKernelGGL((gemm_fast_kernel<T, TO>), dim3((unsigned)nwg), dim3(NTHREADS), 0, st, k);
  else
    hipLaunchKernelGGL((gemm_gen_kernel<T, TO>), dim3((unsigned)nwg), dim3(NTHREADS), 0, st, k);
  BIST_LAUNCH_CHECK("bist_gemm");
  return BIST_OK;
}

}  // namespace

extern "C" int bist_gemm_is_fast(const BistGemm* g) { return g && fast_ok(g) ? 1 : 0; }

extern "C" int bist_gemm(const BistGemm* g, void* stream) {
  BIST_REQUIRE(g != nullptr, "bist_gemm: null descriptor");
  BIST_REQUIRE(g->A && g->B && g->C, "bist_gemm: null operand");
  BIST_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "bist_gemm: bad shape M=%d N=%d K=%d", g->M, g->N, g->K);
  BIST_REQUIRE(g->batch1 > 0 && g->batch2 > 0, "bist_gemm: bad batch %d x %d", g->batch1, g->batch2);
  BIST_REQUIRE(g->in_dtype == BIST_F32 || g->in_dtype == BIST_BF16, "bist_gemm: bad in_dtype %d", g->in_dtype);
  BIST_REQUIRE(g->out_dtype == BIST_F32 || g->out_dtype == g->in_dtype, "bist_gemm: out_dtype must be f32 or equal in_dtype");
  BIST_REQUIRE(g->drop_p >= 0.f && g->drop_p < 1.f, "bist_gemm: drop_p out of range");
  BIST_REQUIRE((g->res_outer == 0) == (g->res_inner == 0) && g->res_outer >= 0, "bist_gemm: bad residual row map");
  GemmK k;
  k.A = (const char*)g->A; k.B = (const char*)g->B; k.C = (char*)g->C;
  k.bias = (const char*)g->bias; k.residual = (const char*)g->residual;
  k.M = g->M; k.N = g->N; k.K = g->K;
  k.a_rs = g->a_rs; k.a_ks = g->a_ks; k.b_rs = g->b_rs; k.b_ks = g->b_ks; k.ldc = g->ldc; k.ldr = g->ldr;
  k.batch2 = g->batch2;
  k.a_bs1 = g->a_bs1; k.a_bs2 = g->a_bs2; k.b_bs1 = g->b_bs1; k.b_bs2 = g->b_bs2;
  k.c_bs1 = g->c_bs1; k.c_bs2 = g->c_bs2; k.r_bs1 = g->r_bs1; k.r_bs2 = g->r_bs2; k.bias_bs2 = g->bias_bs2;
  k.alpha = g->alpha; k.act = g->act; k.res_outer = g->res_outer; k.res_inner = g->res_inner;
  k.drop_p = g->drop_p; k.drop_seed = g->drop_seed;
  k.tiles_m = (g->M + BM - 1) / BM; k.tiles_n = (g->N + BN - 1) / BN;
  const long nwg = (long)k.tiles_m * k.tiles_n * g->batch1 * g->batch2;
  BIST_REQUIRE(nwg < (1L << 31), "bist_gemm: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (g->in_dtype == BIST_BF16) {
    if (g->out_dtype == BIST_BF16) return launch<bf16_t, bf16_t>(g, k, st);
    return launch<bf16_t, float>(g, k, st);
  }
  return launch<float, float>(g, k, st);
}
