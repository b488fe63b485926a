// Backward kernels of the row-wise / elementwise steps and the optimiser (training path).
// Reductions over rows (bias, LayerNorm gain/offset, embedding rows) accumulate into fp32 buffers
// with global float atomics shaped as contiguous row segments (one dword per lane).
#include "common.hpp"
#include <stdlib.h>

namespace {

inline unsigned blocks_for(long n, int per) { return (unsigned)((n + per - 1) / per); }

// dz = dy * d(epilogue)/dz for the GEMM epilogue  y = drop(act(z)) [+ residual]:
//   relu:            dz = dy * [y > 0]            (y is the stored output, no residual allowed)
//   dropout:         dz = dy * keep(seed, idx)/(1-p)   (relu+dropout: [y>0] already encodes keep)
template <typename T>
__global__ void epilogue_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dz, long M, int N,
                                    long lddy, long ldy, long lddz, int act, float drop_p, unsigned long long seed0,
                                    const unsigned long long* __restrict__ ctr) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  const long m = idx / N; const int n = (int)(idx % N);
  float g = to_f(dy[m * lddy + n]);
  if (drop_p > 0.f) {
    const unsigned long long seed = seed0 + (ctr ? ctr[0] * 0xD1B54A32D192ED03ULL : 0ULL);
    const float sc = 1.f / (1.f - drop_p);
    if (act == BIST_ACT_RELU) g *= sc;
    else g = drop_keep(seed, (unsigned long long)idx, drop_p) ? g * sc : 0.f;
  }
  if (act == BIST_ACT_RELU && !(to_f(y[m * ldy + n]) > 0.f)) g = 0.f;
  dz[m * lddz + n] = from_f<T>(g);
}

// the same, 8 consecutive columns per thread (N % 8 == 0, 16-byte aligned rows): 16-byte loads / stores, two mask hashes
template <typename T>
__global__ __launch_bounds__(256) void epilogue_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dz, long M, int N,
                                                               long lddy, long ldy, long lddz, int act, float drop_p, unsigned long long seed0,
                                                               const unsigned long long* __restrict__ ctr) {
  static_assert(sizeof(T) == 2, "bf16 rows");
  const long idx8 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n8 = N >> 3;
  if (idx8 >= M * n8) return;
  const long m = idx8 / n8; const int n = (int)(idx8 % n8) * 8;
  T g8[8], y8[8];
  *reinterpret_cast<uint4*>(g8) = *reinterpret_cast<const uint4*>(dy + m * lddy + n);
  const bool relu = act == BIST_ACT_RELU;
  if (relu) *reinterpret_cast<uint4*>(y8) = *reinterpret_cast<const uint4*>(y + m * ldy + n);
  float g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = to_f(g8[e]);
  if (drop_p > 0.f) {
    const float sc = 1.f / (1.f - drop_p);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] *= sc;
    } else {
      const unsigned long long seed = seed0 + (ctr ? ctr[0] * 0xD1B54A32D192ED03ULL : 0ULL);
      const uint32_t thr = drop_threshold(drop_p);
      const unsigned long long i4 = (unsigned long long)(m * N + n) >> 2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint64_t bits = drop_bits4(seed, i4 + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[4 * q + e] = drop_keep_of(bits, e, thr) ? g[4 * q + e] * sc : 0.f;
      }
    }
  }
  if (relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (!(to_f(y8[e]) > 0.f)) g[e] = 0.f;
  }
  T o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = from_f<T>(g[e]);
  *reinterpret_cast<uint4*>(dz + m * lddz + n) = *reinterpret_cast<const uint4*>(o);
}

// out[b, i, :] = sum_g x[b, g, i, :]   (gradient of the expanded-query residual of stage 1)
template <typename T>
__global__ void group_sum_kernel(const T* __restrict__ x, T* __restrict__ out, int G, long inner, long total, const T* __restrict__ add) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long b = idx / inner, r = idx % inner;
  const T* p = x + b * G * inner + r;
  float acc = add ? to_f(add[idx]) : 0.f;
  for (int g = 0; g < G; ++g) acc += to_f(p[(long)g * inner]);
  out[idx] = from_f<T>(acc);
}

// the same with 16 bytes per thread (inner % E == 0, aligned): G independent 16-byte loads per thread
template <typename T>
__global__ __launch_bounds__(256) void group_sum_vec_kernel(const T* __restrict__ x, T* __restrict__ out, int G, long inner, long total,
                                                            const T* __restrict__ add) {
  constexpr int E = 16 / (int)sizeof(T);
  const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * E;
  if (idx >= total) return;
  const long b = idx / inner, r = idx % inner;
  const T* p = x + b * G * inner + r;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  if (add) {
    T v[E];
    *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(add + idx);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = to_f(v[e]);
  }
#pragma unroll 4
  for (int g = 0; g < G; ++g) {
    T v[E];
    *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(p + (long)g * inner);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += to_f(v[e]);
  }
  T o[E];
#pragma unroll
  for (int e = 0; e < E; ++e) o[e] = from_f<T>(acc[e]);
  *reinterpret_cast<uint4*>(out + idx) = *reinterpret_cast<const uint4*>(o);
}

// out[n] += sum_m x[m, n]   (bias gradient).  Block = 64 columns x 4 row lanes; grid = (column blocks, row chunks);
// the four row lanes are combined through LDS, then one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void col_sum_kernel(const T* __restrict__ x, float* __restrict__ out, long M, int N, long ldx, int rows_per_block) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const long m0 = (long)blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float acc = 0.f;
  if (n < N)
    for (long m = m0 + rl; m < m1; m += 4) acc += to_f(x[m * ldx + n]);
  part[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && n < N) atomicAdd(out + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

// Several bias gradients in one launch: the trainer queues the (dz, accumulator) pairs of a whole backward pass and
// flushes them in batches of COLSUM_JOBS; block -> job through the prefix table carried in the kernel arguments.
constexpr int COLSUM_JOBS = 48;
struct ColSumBatch {
  const void* x[COLSUM_JOBS]; float* out[COLSUM_JOBS];
  int M[COLSUM_JOBS], N[COLSUM_JOBS], ld[COLSUM_JOBS], rpb[COLSUM_JOBS], cb[COLSUM_JOBS];
  unsigned char vec[COLSUM_JOBS];    // 16-byte loads: N, ld whole pieces and an aligned base
  int first[COLSUM_JOBS + 1];        // first block of every job; first[n] = total
  int n;
};
template <typename T>
__global__ __launch_bounds__(256) void col_sum_multi_kernel(const ColSumBatch b) {
  constexpr int E = 16 / (int)sizeof(T);          // columns per thread on the vector path
  __shared__ float part[4][64 * E];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first[j + 1]) ++j;
  const int lb = (int)blockIdx.x - b.first[j];
  const int bx = lb % b.cb[j], by = lb / b.cb[j];
  const T* x = reinterpret_cast<const T*>(b.x[j]);
  const int M = b.M[j], N = b.N[j], ld = b.ld[j], rpb = b.rpb[j];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int m0 = by * rpb, m1 = min(M, m0 + rpb);
  if (b.vec[j]) {
    // 16 bytes per thread per row: a block covers 64*E columns x 4 row lanes (rows are whole 16-byte pieces here)
    const int n0 = (bx * 64 + c) * E;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (n0 < N) {
#pragma unroll 4
      for (int m = m0 + rl; m < m1; m += 4) {
        const uint4 q = *reinterpret_cast<const uint4*>(x + (long)m * ld + n0);
        const T* qe = reinterpret_cast<const T*>(&q);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += to_f(qe[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) part[rl][c * E + e] = acc[e];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * E; idx += 256) {
      const int n = bx * 64 * E + idx;
      if (n < N) atomicAdd(b.out[j] + n, part[0][idx] + part[1][idx] + part[2][idx] + part[3][idx]);
    }
    return;
  }
  const int n = bx * 64 + c;
  float acc = 0.f;
  if (n < N)
    for (int m = m0 + rl; m < m1; m += 4) acc += to_f(x[(long)m * ld + n]);
  part[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && n < N) atomicAdd(b.out[j] + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

// LayerNorm backward (forward: y = a*(x-mean)/(std+eps)+b, std unbiased):
//   g = dy*a;  dx_i = (g_i - mean(g))/s - xc_i * sum_j(g_j xc_j) / ((d-1) * std * s^2),  s = std+eps
//   da += dy * xc/s ; db += dy          (fp32 atomics, one row per wave, 4 rows per block)
// Several LayerNorm backward passes of one geometry in ONE launch (bist_layernorm_bwd_multi): set = blockIdx.y.  row0 = the set's first
// row in the stacked tensor the dropout mask of dz is indexed by (the mask of the producing GEMM's epilogue runs over the stacked output).
constexpr int LNB_SETS = 8;
struct LnBwdSetsK {
  const void* dy[LNB_SETS]; const void* x[LNB_SETS]; const void* a[LNB_SETS]; void* dx[LNB_SETS]; float* da[LNB_SETS]; float* db[LNB_SETS];
  const void* add[LNB_SETS]; void* dz[LNB_SETS]; unsigned long long row0[LNB_SETS];
};

template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const LnBwdSetsK sets, long rows, int d, long lddy, long ldx, long lddx, float eps,
                                                            int rows_per_wave, long ldadd, const DropArg zdrop) {
  const int zs = blockIdx.y;
  const T* __restrict__ dy = (const T*)sets.dy[zs]; const T* __restrict__ x = (const T*)sets.x[zs]; const T* __restrict__ a = (const T*)sets.a[zs];
  T* __restrict__ dx = (T*)sets.dx[zs]; float* __restrict__ da = sets.da[zs]; float* __restrict__ db = sets.db[zs];
  const T* __restrict__ dx_add = (const T*)sets.add[zs]; T* __restrict__ dz = (T*)sets.dz[zs];
  const unsigned long long zrow0 = sets.row0[zs];
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long r0 = wave * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
  const unsigned long long zkey = dz ? zdrop.key() : 0ULL;
  const float zks = dz ? zdrop.keep_scale() : 1.f;
  // per-lane partial sums of da/db over this wave's rows (columns lane, lane+64, ...), flushed once
  constexpr int MAXC = 32;                       // d <= 2048
  float pa[MAXC], pb[MAXC];
#pragma unroll
  for (int u = 0; u < MAXC; ++u) { pa[u] = 0.f; pb[u] = 0.f; }
  for (long row = r0; row < r1; ++row) {
    const T* xr = x + row * ldx;
    const T* gr = dy + row * lddy;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += to_f(xr[c]);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f, sg = 0.f, sgx = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float xc = to_f(xr[c]) - mean, g = to_f(gr[c]) * to_f(a[c]);
      q += xc * xc; sg += g; sgx += g * xc;
    }
    q = wave_sum(q); sg = wave_sum(sg); sgx = wave_sum(sgx);
    const float stdv = sqrtf(q / (float)(d - 1)), sden = stdv + eps;
    const float inv = 1.f / sden, mg = sg / (float)d;
    const float k2 = sgx / ((float)(d - 1) * stdv * sden * sden);
    T* dxr = dx + row * lddx;
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int c = lane + u * 64;
      if (c < d) {
        const float xc = to_f(xr[c]) - mean, gy = to_f(gr[c]);
        const float add = dx_add ? to_f(dx_add[row * ldadd + c]) : 0.f;
        const T dxv = from_f<T>((gy * to_f(a[c]) - mg) * inv - xc * k2 + add);
        dxr[c] = dxv;
        if (dz) dz[row * (long)d + c] = from_f<T>(to_f(dxv) * drop_mul(zkey, (zrow0 + (unsigned long long)row) * d + c, zdrop.p, zks));
        pa[u] += gy * xc * inv;
        pb[u] += gy;
      }
    }
  }
  if (r0 < r1) {
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int c = lane + u * 64;
      if (c < d) { atomicAdd(da + c, pa[u]); atomicAdd(db + c, pb[u]); }
    }
  }
}

// Vectorised LayerNorm backward for rows that are a whole number of KiB: each lane owns NV 16-byte vectors of
// the row (x and dy read once, kept in registers); the gain/offset gradients are summed over the block's rows in
// registers, combined across the 4 waves through LDS, and leave as ONE atomic per column per block.
// DX: write the input gradient; PARAMS: accumulate the gain/offset gradients.  The trainer runs DX-only kernels on the
// critical path and sums the parameter gradients of all LayerNorms of a backward pass in one batched launch.
template <typename T, int NV, bool DX, bool PARAMS>
__device__ __forceinline__ void layernorm_bwd_vec_body(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ a,
                                                       T* __restrict__ dx, float* __restrict__ da, float* __restrict__ db,
                                                       long rows, int d, long lddy, long ldx, long lddx, float eps, int rows_per_wave,
                                                       const T* __restrict__ dx_add, long ldadd, long blk, float (*red)[4][64 * NV * (16 / (int)sizeof(T))],
                                                       T* __restrict__ dz = nullptr, const DropArg zdrop = DropArg{0.f, 0ULL, nullptr},
                                                       unsigned long long zrow0 = 0ULL) {
  constexpr int E = 16 / (int)sizeof(T), NE = NV * E;
  const unsigned long long zkey = (DX && dz) ? zdrop.key() : 0ULL;
  const float zks = (DX && dz) ? zdrop.keep_scale() : 1.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long wave = blk * 4 + w;
  const long r0 = wave * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
  float av[NE], pa[NE], pb[NE];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const uint4 q = *reinterpret_cast<const uint4*>(a + (j * 64 + lane) * E);
    const T* qe = reinterpret_cast<const T*>(&q);
#pragma unroll
    for (int e = 0; e < E; ++e) { av[j * E + e] = to_f(qe[e]); pa[j * E + e] = 0.f; pb[j * E + e] = 0.f; }
  }
  // two rows per trip: the loads of both rows (x, dy, residual gradient) are issued before either is reduced
  for (long rowb = r0; rowb < r1; rowb += 2) {
    uint4 qx[2][NV], qg[2][NV], qa[2][NV];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const long row = min(rowb + rr, r1 - 1);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        qx[rr][j] = *reinterpret_cast<const uint4*>(x + row * ldx + (j * 64 + lane) * E);
        qg[rr][j] = *reinterpret_cast<const uint4*>(dy + row * lddy + (j * 64 + lane) * E);
        if (DX && dx_add) qa[rr][j] = *reinterpret_cast<const uint4*>(dx_add + row * ldadd + (j * 64 + lane) * E);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const long row = rowb + rr;
      if (row >= r1) break;
      float xv[NE], gv[NE];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const T* xe = reinterpret_cast<const T*>(&qx[rr][j]);
        const T* ge = reinterpret_cast<const T*>(&qg[rr][j]);
#pragma unroll
        for (int e = 0; e < E; ++e) { xv[j * E + e] = to_f(xe[e]); gv[j * E + e] = to_f(ge[e]); s += xv[j * E + e]; }
      }
      const float mean = wave_sum(s) / (float)d;
      float q = 0.f, sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int u = 0; u < NE; ++u) { xv[u] -= mean; const float g = gv[u] * av[u]; q += xv[u] * xv[u]; sg += g; sgx += g * xv[u]; }
      q = wave_sum(q); sg = wave_sum(sg); sgx = wave_sum(sgx);
      const float stdv = sqrtf(q / (float)(d - 1)), sden = stdv + eps;
      const float inv = 1.f / sden, mg = sg / (float)d, k2 = sgx / ((float)(d - 1) * stdv * sden * sden);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        T o[E];
        const T* ad = reinterpret_cast<const T*>(&qa[rr][j]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int u = j * E + e;
          if (DX) o[e] = from_f<T>((gv[u] * av[u] - mg) * inv - xv[u] * k2 + (dx_add ? to_f(ad[e]) : 0.f));
          if (PARAMS) { pa[u] += gv[u] * xv[u] * inv; pb[u] += gv[u]; }
        }
        if (DX) *reinterpret_cast<uint4*>(dx + row * lddx + (j * 64 + lane) * E) = *reinterpret_cast<const uint4*>(o);
        if (DX && dz) {          // the masked copy the producing GEMM's backward needs (its dropout epilogue): dz = mask/(1-p) * dx
          T zo[E];
#pragma unroll
          for (int e = 0; e < E; ++e)
            zo[e] = from_f<T>(to_f(o[e]) * drop_mul(zkey, (zrow0 + (unsigned long long)row) * d + (j * 64 + lane) * E + e, zdrop.p, zks));
          *reinterpret_cast<uint4*>(dz + row * (long)d + (j * 64 + lane) * E) = *reinterpret_cast<const uint4*>(zo);
        }
      }
    }
  }
  if constexpr (PARAMS) {
#pragma unroll
    for (int u = 0; u < NE; ++u) { red[0][w][u * 64 + lane] = pa[u]; red[1][w][u * 64 + lane] = pb[u]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * NE; idx += 256) {
      const int u = idx >> 6, ln = idx & 63;
      const int c = ((u / E) * 64 + ln) * E + (u % E);                    // column owned by (lane ln, register u)
      atomicAdd(da + c, red[0][0][idx] + red[0][1][idx] + red[0][2][idx] + red[0][3][idx]);
      atomicAdd(db + c, red[1][0][idx] + red[1][1][idx] + red[1][2][idx] + red[1][3][idx]);
    }
  }
}

template <typename T, int NV, bool PARAMS>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const LnBwdSetsK sets, long rows, int d, long lddy, long ldx, long lddx, float eps,
                                                                int rows_per_wave, long ldadd, const DropArg zdrop) {
  __shared__ float red[PARAMS ? 2 : 1][4][64 * NV * (16 / (int)sizeof(T))];
  const int zs = blockIdx.y;
  layernorm_bwd_vec_body<T, NV, true, PARAMS>((const T*)sets.dy[zs], (const T*)sets.x[zs], (const T*)sets.a[zs], (T*)sets.dx[zs], sets.da[zs], sets.db[zs],
                                              rows, d, lddy, ldx, lddx, eps, rows_per_wave, (const T*)sets.add[zs], ldadd,
                                              (long)blockIdx.x, red, (T*)sets.dz[zs], zdrop, sets.row0[zs]);
}

// gain/offset gradients of many LayerNorms (one row width) in one launch: block -> job through the prefix table
constexpr int LNGRAD_JOBS = 40;
struct LnGradBatch {
  const void* dy[LNGRAD_JOBS]; const void* x[LNGRAD_JOBS]; const void* a[LNGRAD_JOBS]; float* da[LNGRAD_JOBS]; float* db[LNGRAD_JOBS];
  int rows[LNGRAD_JOBS], lddy[LNGRAD_JOBS], ldx[LNGRAD_JOBS], rpw[LNGRAD_JOBS];
  float eps[LNGRAD_JOBS];
  int first[LNGRAD_JOBS + 1];
  int n, d;
};
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_param_grad_multi_kernel(const LnGradBatch b) {
  __shared__ float red[2][4][64 * NV * (16 / (int)sizeof(T))];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first[j + 1]) ++j;
  layernorm_bwd_vec_body<T, NV, false, true>((const T*)b.dy[j], (const T*)b.x[j], (const T*)b.a[j], nullptr, b.da[j], b.db[j], (long)b.rows[j], b.d,
                                             (long)b.lddy[j], (long)b.ldx[j], 0L, b.eps[j], b.rpw[j], nullptr, 0L,
                                             (long)((int)blockIdx.x - b.first[j]), red);
}

// Embedding backward: dlut[ids[row], c] += dy[row, c] * sqrt(d)   (fp32 atomics)
template <typename T>
__global__ void embed_bwd_kernel(const long* __restrict__ ids, const T* __restrict__ dy, float* __restrict__ dlut, long rows, int d, float scale,
                                 const DropArg drop) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * d) return;
  const long row = idx / d; const int c = (int)(idx % d);
  float g = to_f(dy[idx]) * scale;
  if (drop.p > 0.f) g *= drop_mul(drop.key(), (unsigned long long)idx, drop.p, drop.keep_scale());
  if (g != 0.f) atomicAdd(dlut + ids[row] * d + c, g);
}

// y = x + s[m, head] * bias  and its backward (the value bias of stage 2 under attention dropout, bist_hip.h)
template <typename T>
__global__ void scaled_bias_kernel(const T* __restrict__ x, const float* __restrict__ s, const T* __restrict__ bias, T* __restrict__ y,
                                   long M, int h, int dk, long Mset, long bias_zs) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int d = h * dk;
  if (idx >= M * d) return;
  const long m = idx / d; const int c = (int)(idx % d);
  y[idx] = from_f<T>(to_f(x[idx]) + s[m * h + c / dk] * to_f(bias[(m / Mset) * bias_zs + c]));      // rows [z*Mset, (z+1)*Mset) use bias set z
}
// ds[m, head] = <dy[m, head, :], bias[head, :]>  and  dbias[head, c] += sum_m s[m, head] dy[m, head, c].
// One wave per head per block of RPB rows: the wave walks its rows, keeps the dbias partial sums of its head in
// registers (dk <= 256: up to 4 columns per lane) and issues ONE atomic per column per block (it was one per element).
template <typename T>
__global__ __launch_bounds__(256) void scaled_bias_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ s, const T* __restrict__ bias,
                                                              float* __restrict__ ds, float* __restrict__ dbias, long M, int h, int dk, int rpb,
                                                              long Mset, long bias_zs, long dbias_zs) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long m0 = (long)blockIdx.x * rpb, m1 = min(M, m0 + rpb);
  { const long z = m0 / Mset; bias += z * bias_zs; dbias += z * dbias_zs; }      // Mset % rpb == 0: a block's rows lie in one set
  for (int hh = w; hh < h; hh += 4) {
    float bv[4], part[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; bv[u] = c < dk ? to_f(bias[hh * dk + c]) : 0.f; part[u] = 0.f; }
    for (long m = m0; m < m1; ++m) {
      const T* g = dy + (m * h + hh) * dk;
      const float sv = s[m * h + hh];
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * u;
        const float gv = c < dk ? to_f(g[c]) : 0.f;
        acc += gv * bv[u];
        part[u] += sv * gv;
      }
      acc = wave_sum(acc);
      if (lane == 0) ds[m * h + hh] = acc;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < dk) atomicAdd(dbias + hh * dk + c, part[u]); }
  }
}

// Modality fusion backward: out = sum_j w_j x_j, w = softmax(score)
struct FusePtrs { const void* x[4]; void* dx[4]; };
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_kernel(const T* __restrict__ score, FusePtrs p, const T* __restrict__ dout,
                                                       T* __restrict__ dscore, long rows, int n, int d, int vec) {
  constexpr int E = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float w[4], dw[4], mx = -INFINITY, den = 0.f;
  for (int j = 0; j < n; ++j) { w[j] = to_f(score[row * n + j]); mx = fmaxf(mx, w[j]); }
  for (int j = 0; j < n; ++j) { w[j] = expf(w[j] - mx); den += w[j]; }
  for (int j = 0; j < n; ++j) { w[j] /= den; dw[j] = 0.f; }
  if (vec) {                               // 16-byte pieces: d % E == 0 and every row 16-byte aligned
    for (int c = lane * E; c < d; c += 64 * E) {
      T g8[E];
      *reinterpret_cast<uint4*>(g8) = *reinterpret_cast<const uint4*>(dout + row * d + c);
      for (int j = 0; j < n; ++j) {
        T x8[E], o8[E];
        *reinterpret_cast<uint4*>(x8) = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.x[j]) + row * d + c);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float g = to_f(g8[e]);
          dw[j] += g * to_f(x8[e]);
          o8[e] = from_f<T>(w[j] * g);
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.dx[j]) + row * d + c) = *reinterpret_cast<const uint4*>(o8);
      }
    }
  } else {
    for (int c = lane; c < d; c += 64) {
      const float g = to_f(dout[row * d + c]);
      for (int j = 0; j < n; ++j) {
        dw[j] += g * to_f(reinterpret_cast<const T*>(p.x[j])[row * d + c]);
        reinterpret_cast<T*>(p.dx[j])[row * d + c] = from_f<T>(w[j] * g);
      }
    }
  }
  float dot = 0.f;
  for (int j = 0; j < n; ++j) { dw[j] = wave_sum(dw[j]); dot += w[j] * dw[j]; }
  if (lane == 0)
    for (int j = 0; j < n; ++j) dscore[row * n + j] = from_f<T>(w[j] * (dw[j] - dot));
}

// dst (T) = cast(src f32) -- used to hand fp32-accumulated gradients back in the parameter dtype
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ s, T* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = from_f<T>(s[i]);
}

// dst[i] += src[i] (fp32 accumulator -> gradient buffer in the compute dtype)
template <typename T>
__global__ void add_f32_into_kernel(const float* __restrict__ s, T* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = from_f<T>(to_f(d[i]) + s[i]);
}

// Adam step on fp32 master weights with an optional low-precision working copy:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)
// (torch.optim.Adam semantics, the optimiser the reference wraps in NoamOpt, train.py:129-130)
// hyper (nullable, device): {lr, 1 - beta1^t, 1 - beta2^t, grad_scale} of THIS step -- lets the launch live inside a captured
// hipGraph, whose kernel arguments are frozen at capture time
// one element of torch.optim.Adam (both kernels below call this, so they round alike)
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float lr, float b1, float b2, float eps, float bc1, float bc2) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  p = p - lr * ((m / bc1) / (sqrtf(v / bc2) + eps));
}
template <typename TG, typename TW>
__global__ void adam_kernel(float* __restrict__ p, const TG* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            TW* __restrict__ work, long n, float lr, float b1, float b2, float eps, float bc1, float bc2, float gscale,
                            const float* __restrict__ hyper) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; gscale = hyper[3]; }
  float mi = m[i], vi = v[i], pi = p[i];
  adam_elem(pi, mi, vi, to_f(g[i]) * gscale, lr, b1, b2, eps, bc1, bc2);
  m[i] = mi; v[i] = vi;
  p[i] = pi;
  if (work) work[i] = from_f<TW>(pi);
}

// adam_kernel on 4 elements per thread (16-byte accesses of p / m / v, 8 or 16 of the gradient and the working copy): n4 = n / 4
template <typename TG, typename TW>
__global__ __launch_bounds__(256) void adam4_kernel(float* __restrict__ p, const TG* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    TW* __restrict__ work, long n4, float b1, float b2, float eps, const float* __restrict__ hyper) {
  const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], gscale = hyper[3];
  // grid-stride: one pass for the chip-filling launch; the BACKGROUND form (bist_adam_step_dev_bg) runs a few hundred workgroups that
  // leave the CUs' wave slots to the step's small launches
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
  float gi[4];
  if constexpr (sizeof(TG) == 2) {
    const uint2 q = reinterpret_cast<const uint2*>(g)[i];
    gi[0] = __builtin_bit_cast(float, q.x << 16); gi[1] = __builtin_bit_cast(float, q.x & 0xffff0000u);
    gi[2] = __builtin_bit_cast(float, q.y << 16); gi[3] = __builtin_bit_cast(float, q.y & 0xffff0000u);
  } else {
    const float4 q = reinterpret_cast<const float4*>(g)[i];
    gi[0] = q.x; gi[1] = q.y; gi[2] = q.z; gi[3] = q.w;
  }
  const float4 pm = reinterpret_cast<const float4*>(p)[i], mm = reinterpret_cast<const float4*>(m)[i], vm = reinterpret_cast<const float4*>(v)[i];
  float pv[4] = {pm.x, pm.y, pm.z, pm.w}, mv[4] = {mm.x, mm.y, mm.z, mm.w}, vv[4] = {vm.x, vm.y, vm.z, vm.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) adam_elem(pv[e], mv[e], vv[e], gi[e] * gscale, lr, b1, b2, eps, bc1, bc2);
  reinterpret_cast<float4*>(m)[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
  reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
  reinterpret_cast<float4*>(p)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
  if (work) {
    if constexpr (sizeof(TW) == 2) {
      typedef __attribute__((ext_vector_type(2))) float f32x2_;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
      const bf16x2_ lo = __builtin_convertvector(f32x2_{pv[0], pv[1]}, bf16x2_), hi = __builtin_convertvector(f32x2_{pv[2], pv[3]}, bf16x2_);
      reinterpret_cast<uint2*>(work)[i] = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
    } else {
      reinterpret_cast<float4*>(work)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
    }
  }
  }
}

// Adam on 4 elements per thread with the step's scalars in hyper = {lr, 1 - beta1^t, 1 - beta2^t, grad_scale, apply}; the gradient is
// CLEARED after it is read, so the next backward pass accumulates into zeros without a memset of its own.  apply == 0 (nothing
// pending): only the clear.
template <typename TG, typename TW>
__global__ __launch_bounds__(256) void adam_apply4_kernel(float* __restrict__ p, TG* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          TW* __restrict__ work, long n4, float b1, float b2, float eps,
                                                          const float* __restrict__ hyper) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], gscale = hyper[3];
  const bool apply = hyper[4] != 0.f;
  float gi[4];
  if constexpr (sizeof(TG) == 2) {
    uint2* gp = reinterpret_cast<uint2*>(g) + i;
    if (apply) {
      const uint2 q = *gp;
      gi[0] = __builtin_bit_cast(float, q.x << 16); gi[1] = __builtin_bit_cast(float, q.x & 0xffff0000u);
      gi[2] = __builtin_bit_cast(float, q.y << 16); gi[3] = __builtin_bit_cast(float, q.y & 0xffff0000u);
    }
    *gp = make_uint2(0u, 0u);
  } else {
    float4* gp = reinterpret_cast<float4*>(g) + i;
    if (apply) { const float4 q = *gp; gi[0] = q.x; gi[1] = q.y; gi[2] = q.z; gi[3] = q.w; }
    *gp = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (!apply) return;
  float4 pm = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vm = reinterpret_cast<float4*>(v)[i];
  float pv[4] = {pm.x, pm.y, pm.z, pm.w}, mv[4] = {mm.x, mm.y, mm.z, mm.w}, vv[4] = {vm.x, vm.y, vm.z, vm.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    adam_elem(pv[e], mv[e], vv[e], gi[e] * gscale, lr, b1, b2, eps, bc1, bc2);
  }
  reinterpret_cast<float4*>(m)[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
  reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
  reinterpret_cast<float4*>(p)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
  if (work) {
    if constexpr (sizeof(TW) == 2) {
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      const f32x2 lo = {pv[0], pv[1]}, hi = {pv[2], pv[3]};
      reinterpret_cast<uint2*>(work)[i] = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)),
                                                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2)));
    } else {
      reinterpret_cast<float4*>(work)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
    }
  }
}

// The scalars of the PENDING optimiser step t = pending[0] (0 = none) and pending[0] <- 0: the head of a step applies the previous
// step's update (bist_adam_apply_dev) beside its own forward pass
__global__ void noam_hyper_pending_kernel(long long* __restrict__ pending, float* __restrict__ hyper, double d_model, double factor,
                                          double warmup, double b1, double b2, float gscale) {
  const long long ti = pending[0];
  if (ti <= 0) { hyper[0] = 0.f; hyper[1] = 1.f; hyper[2] = 1.f; hyper[3] = gscale; hyper[4] = 0.f; return; }
  const double t = (double)ti;
  hyper[0] = (float)(factor * (pow(d_model, -0.5) * fmin(pow(t, -0.5), t * pow(warmup, -1.5))));
  hyper[1] = (float)(1.0 - pow(b1, t));
  hyper[2] = (float)(1.0 - pow(b2, t));
  hyper[3] = gscale;
  hyper[4] = 1.f;
  pending[0] = 0;
}

}  // namespace

#define DISPATCH_T(dtype, NAME, ...)                                                  \
  if ((dtype) == BIST_BF16) { NAME(bf16_t, __VA_ARGS__); }                            \
  else if ((dtype) == BIST_F32) { NAME(float, __VA_ARGS__); }                         \
  else { bist_set_error("%s: bad dtype %d", __func__, (int)(dtype)); return BIST_EINVAL; }

extern "C" int bist_epilogue_bwd(const void* dy, const void* y, void* dz, int64_t M, int32_t N, int64_t lddy, int64_t ldy,
                                 int64_t lddz, int32_t act, float drop_p, uint64_t drop_seed, const uint64_t* drop_ctr, int32_t dtype,
                                 void* stream) {
  BIST_REQUIRE(dy && dz && M > 0 && N > 0, "bist_epilogue_bwd: bad argument");
  BIST_REQUIRE(act != BIST_ACT_RELU || y, "bist_epilogue_bwd: relu needs the forward output");
  hipStream_t st = (hipStream_t)stream;
  auto al = [](const void* p, long ld) { return ((uintptr_t)p % 16 == 0) && (ld % 8 == 0); };
  if (dtype == BIST_BF16 && N % 8 == 0 && al(dy, lddy) && al(dz, lddz) && (!y || al(y, ldy))) {
    hipLaunchKernelGGL(epilogue_bwd_vec_kernel<bf16_t>, dim3(blocks_for(M * (N / 8), 256)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y,
                       (bf16_t*)dz, (long)M, N, (long)lddy, (long)ldy, (long)lddz, act, drop_p, (unsigned long long)drop_seed,
                       (const unsigned long long*)drop_ctr);
    BIST_LAUNCH_CHECK("bist_epilogue_bwd");
    return BIST_OK;
  }
#define L(TT, ...) hipLaunchKernelGGL(epilogue_bwd_kernel<TT>, dim3(blocks_for(M * N, 256)), dim3(256), 0, st, (const TT*)dy, (const TT*)y, (TT*)dz, (long)M, N, (long)lddy, (long)ldy, (long)lddz, act, drop_p, (unsigned long long)drop_seed, (const unsigned long long*)drop_ctr)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_epilogue_bwd");
  return BIST_OK;
}

// group sum AND the dropout-masked copy of its input in one pass: dres[b, r] = add[b, r] + sum_g x[b, g, r],  dz[b, g, r] = mask * x[b, g, r] / (1 - p)
// (mask index = the element's flat index, as the GEMM epilogue that produced drop(z) + residual): the two first steps of the backward of
// y = x_query (expanded over the groups) + drop(W_o ctx + b_o) read dy once instead of twice.  16 bytes per thread.
template <typename T>
__global__ __launch_bounds__(256) void group_sum_mask_kernel(const T* __restrict__ x, T* __restrict__ out, T* __restrict__ dz, int G, long inner, long total,
                                                             const T* __restrict__ add, const DropArg drop) {
  // a workgroup = 64 sixteen-byte column pieces x 4 slices of the G groups (one wave per slice: four times the loads in flight of a
  // thread per piece walking all groups, which left 0.3 waves per SIMD at B = 16: 21 us for 32 MB); the slices' partial sums meet in LDS
  constexpr int E = 16 / (int)sizeof(T);
  __shared__ float part[3][64][E];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long idx = ((long)blockIdx.x * 64 + lane) * E;
  const bool live = idx < total;
  const long b = live ? idx / inner : 0, r = live ? idx % inner : 0;
  const long base = b * G * inner + r;
  const unsigned long long key = drop.key();
  const uint32_t thr = drop_threshold(drop.p);
  const float ks = drop.keep_scale();
  const int per = (G + 3) / 4, g0 = sl * per, g1 = min(G, g0 + per);
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  if (live) {
#pragma unroll 4
    for (int g = g0; g < g1; ++g) {
      const long off = base + (long)g * inner;
      T v[E], o[E];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(x + off);
#pragma unroll
      for (int q = 0; q < E / 4; ++q) {
        const uint64_t bits = drop_bits4(key, (unsigned long long)(off >> 2) + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f = to_f(v[4 * q + e]);
          acc[4 * q + e] += f;
          o[4 * q + e] = from_f<T>(drop_keep_of(bits, e, thr) ? f * ks : 0.f);
        }
      }
      *reinterpret_cast<uint4*>(dz + off) = *reinterpret_cast<const uint4*>(o);
    }
  }
  if (sl > 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) part[sl - 1][lane][e] = acc[e];
  }
  __syncthreads();
  if (sl == 0 && live) {
    // (the same order of additions for every launch: slices 0, 1, 2, 3, then the extra addend)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += part[0][lane][e] + part[1][lane][e] + part[2][lane][e];
    if (add) {
      T v[E];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(add + idx);
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += to_f(v[e]);
    }
    T o2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) o2[e] = from_f<T>(acc[e]);
    *reinterpret_cast<uint4*>(out + idx) = *reinterpret_cast<const uint4*>(o2);
  }
}

extern "C" int bist_group_sum_mask(const void* x, const void* add, void* out, void* dz, int64_t B, int32_t G, int64_t inner, const BistDrop* drop,
                                   int32_t dtype, void* stream) {
  BIST_REQUIRE(x && out && dz && drop && B > 0 && G > 0 && inner > 0 && drop->p > 0.f && drop->p < 1.f, "bist_group_sum_mask: bad argument");
  const long gsz = dtype == BIST_BF16 ? 2 : 4, ge = 16 / gsz;
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && inner % ge == 0 &&
               (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dz | (uintptr_t)add) % 16) == 0, "bist_group_sum_mask: 16-byte rows, bf16 / f32");
  hipStream_t st = (hipStream_t)stream;
  const long total = B * inner;
  const DropArg dr = make_drop(drop);
  if (dtype == BIST_BF16) hipLaunchKernelGGL(group_sum_mask_kernel<bf16_t>, dim3(blocks_for(total / ge, 64)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, (bf16_t*)dz, G, (long)inner, total, (const bf16_t*)add, dr);
  else hipLaunchKernelGGL(group_sum_mask_kernel<float>, dim3(blocks_for(total / ge, 64)), dim3(256), 0, st, (const float*)x, (float*)out, (float*)dz, G, (long)inner, total, (const float*)add, dr);
  BIST_LAUNCH_CHECK("bist_group_sum_mask");
  return BIST_OK;
}

extern "C" int bist_group_sum_add(const void* x, const void* add, void* out, int64_t B, int32_t G, int64_t inner, int32_t dtype, void* stream) {
  BIST_REQUIRE(x && out && B > 0 && G > 0 && inner > 0, "bist_group_sum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long total = B * inner;
  const long gsz = dtype == BIST_BF16 ? 2 : 4, ge = 16 / gsz;
  if ((dtype == BIST_BF16 || dtype == BIST_F32) && inner % ge == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)add % 16) == 0) {
    if (dtype == BIST_BF16) hipLaunchKernelGGL(group_sum_vec_kernel<bf16_t>, dim3(blocks_for(total / ge, 256)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, G, (long)inner, total, (const bf16_t*)add);
    else hipLaunchKernelGGL(group_sum_vec_kernel<float>, dim3(blocks_for(total / ge, 256)), dim3(256), 0, st, (const float*)x, (float*)out, G, (long)inner, total, (const float*)add);
    BIST_LAUNCH_CHECK("bist_group_sum");
    return BIST_OK;
  }
#define L(TT, ...) hipLaunchKernelGGL(group_sum_kernel<TT>, dim3(blocks_for(total, 256)), dim3(256), 0, st, (const TT*)x, (TT*)out, G, (long)inner, total, (const TT*)add)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_group_sum");
  return BIST_OK;
}
extern "C" int bist_group_sum(const void* x, void* out, int64_t B, int32_t G, int64_t inner, int32_t dtype, void* stream) {
  return bist_group_sum_add(x, nullptr, out, B, G, inner, dtype, stream);
}

extern "C" int bist_col_sum_acc(const void* x, float* out, int64_t M, int32_t N, int64_t ldx, int32_t dtype, void* stream) {
  BIST_REQUIRE(x && out && M > 0 && N > 0 && ldx >= N, "bist_col_sum_acc: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long cb = blocks_for(N, 64);
  long rpb = (M * cb + 1023) / 1024;                 // aim at ~1024 workgroups
  if (rpb < 16) rpb = 16;
  dim3 grid((unsigned)cb, blocks_for(M, (int)rpb));
#define L(TT, ...) hipLaunchKernelGGL(col_sum_kernel<TT>, grid, dim3(256), 0, st, (const TT*)x, out, (long)M, N, (long)ldx, (int)rpb)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_col_sum_acc");
  return BIST_OK;
}

extern "C" int bist_col_sum_multi(const BistColSum* jobs, int32_t njobs, int32_t dtype, void* stream) {
  BIST_REQUIRE(jobs && njobs > 0, "bist_col_sum_multi: bad argument");
  BIST_REQUIRE(dtype == BIST_BF16 || dtype == BIST_F32, "bist_col_sum_multi: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < njobs; base += COLSUM_JOBS) {
    ColSumBatch b;
    b.n = njobs - base < COLSUM_JOBS ? njobs - base : COLSUM_JOBS;
    int total = 0;
    for (int i = 0; i < b.n; ++i) {
      const BistColSum& q = jobs[base + i];
      BIST_REQUIRE(q.x && q.out && q.M > 0 && q.N > 0 && q.ldx >= q.N && q.M < (1L << 31) && q.ldx < (1L << 31), "bist_col_sum_multi: bad job %d", base + i);
      const long esz = dtype == BIST_BF16 ? 2 : 4, piece = 16 / esz;
      const bool vec = (q.N % piece == 0) && (q.ldx % piece == 0) && ((uintptr_t)q.x % 16 == 0);
      const long cb = vec ? blocks_for(q.N, 64 * piece) : blocks_for(q.N, 64);
      static const long target = [] { const char* e = getenv("BIST_COLSUM_TARGET"); return e ? atol(e) : 128L; }();   // tuning aid
      // ~128 workgroups per job (at least 16 rows each): every workgroup ends in N atomics on the SAME N accumulators, and with
      // 1024 workgroups per [25088 x 512] job the step spent 0.4 ms more in these reductions (12.38 vs 11.97 ms)
      long rpb = (q.M * cb + target - 1) / target;
      if (rpb < 16) rpb = 16;
      b.vec[i] = vec ? 1 : 0;
      b.x[i] = q.x; b.out[i] = q.out; b.M[i] = (int)q.M; b.N[i] = q.N; b.ld[i] = (int)q.ldx; b.rpb[i] = (int)rpb; b.cb[i] = (int)cb;
      b.first[i] = total;
      total += (int)(cb * blocks_for(q.M, (int)rpb));
    }
    b.first[b.n] = total;
    if (dtype == BIST_BF16) hipLaunchKernelGGL(col_sum_multi_kernel<bf16_t>, dim3((unsigned)total), dim3(256), 0, st, b);
    else hipLaunchKernelGGL(col_sum_multi_kernel<float>, dim3((unsigned)total), dim3(256), 0, st, b);
    BIST_LAUNCH_CHECK("bist_col_sum_multi");
  }
  return BIST_OK;
}

extern "C" int bist_layernorm_bwd_multi(const BistLnBwdSet* sets, int32_t nsets, int64_t rows, int32_t d, int64_t lddy, int64_t ldx, int64_t lddx,
                                        float eps, int64_t ldadd, const BistDrop* dz_drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(sets && nsets >= 1 && nsets <= LNB_SETS, "bist_layernorm_bwd_multi: 1..%d sets", LNB_SETS);
  BIST_REQUIRE(rows > 0 && d > 1 && d <= 2048, "bist_layernorm_bwd: bad shape rows=%ld d=%d", (long)rows, d);
  hipStream_t st = (hipStream_t)stream;
  const long sz = dtype == BIST_BF16 ? 2 : 4;
  LnBwdSetsK k{};
  bool any_dz = false, any_add = false, al = (lddy * sz) % 16 == 0 && (ldx * sz) % 16 == 0 && (lddx * sz) % 16 == 0;
  const bool params = sets[0].da != nullptr;
  for (int i = 0; i < nsets; ++i) {
    const BistLnBwdSet& q = sets[i];
    BIST_REQUIRE(q.dy && q.x && q.a && q.dx && ((q.da != nullptr) == (q.db != nullptr)), "bist_layernorm_bwd: null pointer (set %d)", i);
    BIST_REQUIRE((q.da != nullptr) == params, "bist_layernorm_bwd_multi: every set or no set accumulates the parameter gradients");
    k.dy[i] = q.dy; k.x[i] = q.x; k.a[i] = q.a; k.dx[i] = q.dx; k.da[i] = q.da; k.db[i] = q.db; k.add[i] = q.dx_add; k.dz[i] = q.dz; k.row0[i] = q.drop_row0;
    any_dz = any_dz || q.dz; any_add = any_add || q.dx_add;
    al = al && (((uintptr_t)q.dy | (uintptr_t)q.x | (uintptr_t)q.a | (uintptr_t)q.dx | (uintptr_t)q.dx_add | (uintptr_t)q.dz) % 16 == 0);
  }
  BIST_REQUIRE(!any_add || ldadd >= d, "bist_layernorm_bwd: bad dx_add stride");
  BIST_REQUIRE(!any_dz || (dz_drop && dz_drop->p > 0.f && dz_drop->p < 1.f), "bist_layernorm_bwd: dz needs a dropout spec");
  const DropArg zd = make_drop(any_dz ? dz_drop : nullptr);
  al = al && (!any_add || (ldadd * sz) % 16 == 0);
  if (al && (d * sz) == 1024) {                   // d = 512 bf16 / 256 f32: one 16-byte vector per lane
    int rpw2 = (int)((rows * nsets + 1023) / 1024);       // ~1024 waves: enough parallelism for dx, few atomics per column
    if (rpw2 < 2) rpw2 = 2;
    const dim3 g2(blocks_for(blocks_for(rows, rpw2), 4), (unsigned)nsets);
#define LNV(TT, PP) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<TT, 1, PP>), g2, dim3(256), 0, st, k, (long)rows, d, (long)lddy, (long)ldx, (long)lddx, eps, \
                                       rpw2, (long)ldadd, zd)
    if (dtype == BIST_BF16) { if (params) LNV(bf16_t, true); else LNV(bf16_t, false); }
    else { if (params) LNV(float, true); else LNV(float, false); }
#undef LNV
    BIST_LAUNCH_CHECK("bist_layernorm_bwd");
    return BIST_OK;
  }
  BIST_REQUIRE(params, "bist_layernorm_bwd: the dx-only form needs 1 KiB rows (d = 512 bf16 / 256 f32), 16-byte aligned");
  int rpw = (int)((rows * nsets + 4095) / 4096);          // <= 4096 waves flush their partial sums
  if (rpw < 1) rpw = 1;
  const dim3 g(blocks_for(blocks_for(rows, rpw), 4), (unsigned)nsets);
#define L(TT, ...) hipLaunchKernelGGL(layernorm_bwd_kernel<TT>, g, dim3(256), 0, st, k, (long)rows, d, (long)lddy, (long)ldx, (long)lddx, eps, rpw, (long)ldadd, zd)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_layernorm_bwd");
  return BIST_OK;
}

extern "C" int bist_layernorm_bwd(const void* dy, const void* x, const void* a, void* dx, float* da, float* db, int64_t rows,
                                  int32_t d, int64_t lddy, int64_t ldx, int64_t lddx, float eps, const void* dx_add, int64_t ldadd,
                                  void* dz, const BistDrop* dz_drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(dy && x && a && dx && ((da != nullptr) == (db != nullptr)), "bist_layernorm_bwd: null pointer");
  const BistLnBwdSet one{dy, x, a, dx, da, db, dx_add, dz, 0ULL};
  return bist_layernorm_bwd_multi(&one, 1, rows, d, lddy, ldx, lddx, eps, ldadd, dz_drop, dtype, stream);
}

extern "C" int bist_layernorm_param_grad_multi(const BistLnGrad* jobs, int32_t njobs, int32_t d, int32_t dtype, void* stream) {
  BIST_REQUIRE(jobs && njobs > 0, "bist_layernorm_param_grad_multi: bad argument");
  const long sz = dtype == BIST_BF16 ? 2 : 4;
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && d * sz == 1024, "bist_layernorm_param_grad_multi: rows must be 1 KiB (d = 512 bf16 / 256 f32)");
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < njobs; base += LNGRAD_JOBS) {
    LnGradBatch b;
    b.n = njobs - base < LNGRAD_JOBS ? njobs - base : LNGRAD_JOBS;
    b.d = d;
    int total = 0;
    for (int i = 0; i < b.n; ++i) {
      const BistLnGrad& q = jobs[base + i];
      BIST_REQUIRE(q.dy && q.x && q.a && q.da && q.db && q.rows > 0 && q.rows < (1L << 31) && q.lddy >= d && q.ldx >= d,
                   "bist_layernorm_param_grad_multi: bad job %d", base + i);
      BIST_REQUIRE((((uintptr_t)q.dy | (uintptr_t)q.x | (uintptr_t)q.a) % 16 == 0) && (q.lddy * sz) % 16 == 0 && (q.ldx * sz) % 16 == 0,
                   "bist_layernorm_param_grad_multi: job %d is not 16-byte aligned", base + i);
      static const long wtarget = [] { const char* e = getenv("BIST_LNGRAD_TARGET"); return e ? atol(e) : 4096L; }();    // tuning aid
      int rpw = (int)((q.rows + wtarget - 1) / wtarget);           // up to ~4096 waves per job (the [25088 x 512] LayerNorm of P0: 7 rows per wave; 11.80 vs 11.98 ms per step at 256 waves)
      if (rpw < 4) rpw = 4;
      // few-row jobs (the ~120 LayerNorms on B*Lq = 320 rows): every block ends in 2*d atomics on the job's OWN accumulators, so
      // 20 blocks per job were 2.4 M atomics per step; 4 blocks per job instead (12.08 -> 11.89 ms per step; 1 block: 12.16)
      static const long small_blocks = [] { const char* e = getenv("BIST_LNGRAD_SMALL_BLOCKS"); return e ? atol(e) : 4L; }();    // tuning aid
      if (q.rows <= 2048) { const long r2 = (q.rows + 4 * small_blocks - 1) / (4 * small_blocks); if (r2 > rpw) rpw = (int)r2; }
      b.dy[i] = q.dy; b.x[i] = q.x; b.a[i] = q.a; b.da[i] = q.da; b.db[i] = q.db;
      b.rows[i] = (int)q.rows; b.lddy[i] = (int)q.lddy; b.ldx[i] = (int)q.ldx; b.rpw[i] = rpw; b.eps[i] = q.eps;
      b.first[i] = total;
      total += (int)blocks_for(blocks_for(q.rows, rpw), 4);
    }
    b.first[b.n] = total;
    if (dtype == BIST_BF16) hipLaunchKernelGGL((layernorm_param_grad_multi_kernel<bf16_t, 1>), dim3((unsigned)total), dim3(256), 0, st, b);
    else hipLaunchKernelGGL((layernorm_param_grad_multi_kernel<float, 1>), dim3((unsigned)total), dim3(256), 0, st, b);
    BIST_LAUNCH_CHECK("bist_layernorm_param_grad_multi");
  }
  return BIST_OK;
}

extern "C" int bist_scaled_bias_fwd_z(const void* x, const float* s, const void* bias, void* y, int64_t M, int32_t h, int32_t dk,
                                      int32_t nsets, int64_t bias_zs, int32_t dtype, void* stream) {
  BIST_REQUIRE(x && s && bias && y && M > 0 && h > 0 && dk > 0 && nsets >= 1 && M % nsets == 0, "bist_scaled_bias_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long Mset = M / nsets;
#define L(TT, ...) hipLaunchKernelGGL(scaled_bias_kernel<TT>, dim3(blocks_for(M * h * dk, 256)), dim3(256), 0, st, (const TT*)x, s, (const TT*)bias, (TT*)y, (long)M, h, dk, Mset, (long)bias_zs)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_scaled_bias_fwd");
  return BIST_OK;
}
extern "C" int bist_scaled_bias_fwd(const void* x, const float* s, const void* bias, void* y, int64_t M, int32_t h, int32_t dk,
                                    int32_t dtype, void* stream) {
  return bist_scaled_bias_fwd_z(x, s, bias, y, M, h, dk, 1, 0, dtype, stream);
}

extern "C" int bist_scaled_bias_bwd_z(const void* dy, const float* s, const void* bias, float* ds, float* dbias, int64_t M, int32_t h,
                                      int32_t dk, int32_t nsets, int64_t bias_zs, int64_t dbias_zs, int32_t dtype, void* stream) {
  BIST_REQUIRE(dy && s && bias && ds && dbias && M > 0 && h > 0 && dk > 0 && nsets >= 1 && M % nsets == 0, "bist_scaled_bias_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  BIST_REQUIRE(dk <= 256, "bist_scaled_bias_bwd: head width %d > 256", dk);
  const long Mset = M / nsets;
  int rpb = 4;                                   // rows per block: M / 4 blocks, 4 atomics per column per 16 rows ... (M = 320: 80 blocks)
  while (rpb > 1 && Mset % rpb != 0) rpb >>= 1;   // a block's rows must lie in one set
#define L(TT, ...) hipLaunchKernelGGL(scaled_bias_bwd_kernel<TT>, dim3(blocks_for(M, rpb)), dim3(256), 0, st, (const TT*)dy, s, (const TT*)bias, ds, dbias, (long)M, h, dk, rpb, Mset, (long)bias_zs, (long)dbias_zs)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_scaled_bias_bwd");
  return BIST_OK;
}
extern "C" int bist_scaled_bias_bwd(const void* dy, const float* s, const void* bias, float* ds, float* dbias, int64_t M, int32_t h,
                                    int32_t dk, int32_t dtype, void* stream) {
  return bist_scaled_bias_bwd_z(dy, s, bias, ds, dbias, M, h, dk, 1, 0, 0, dtype, stream);
}

extern "C" int bist_embed_bwd(const int64_t* ids, const void* dy, float* dlut, int64_t rows, int32_t d, const BistDrop* drop,
                              int32_t dtype, void* stream) {
  BIST_REQUIRE(ids && dy && dlut && rows > 0 && d > 0, "bist_embed_bwd: bad argument");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_embed_bwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  hipStream_t st = (hipStream_t)stream;
  const float scale = sqrtf((float)d);
#define L(TT, ...) hipLaunchKernelGGL(embed_bwd_kernel<TT>, dim3(blocks_for(rows * d, 256)), dim3(256), 0, st, (const long*)ids, (const TT*)dy, dlut, (long)rows, d, scale, dr)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_embed_bwd");
  return BIST_OK;
}

extern "C" int bist_fuse_modalities_bwd(const void* score, const void* const* xs, const void* dout, void* dscore, void* const* dxs,
                                        int64_t rows, int32_t n, int32_t d, int32_t dtype, void* stream) {
  BIST_REQUIRE(score && xs && dout && dscore && dxs && rows > 0 && d > 0 && n >= 1 && n <= 4, "bist_fuse_modalities_bwd: bad argument");
  FusePtrs p;
  for (int j = 0; j < 4; ++j) { p.x[j] = j < n ? xs[j] : nullptr; p.dx[j] = j < n ? dxs[j] : nullptr; }
  hipStream_t st = (hipStream_t)stream;
  const long esz = dtype == BIST_BF16 ? 2 : 4;
  int vec = (d % (16 / esz)) == 0 && ((uintptr_t)dout % 16) == 0;
  for (int j = 0; j < n; ++j) vec = vec && ((uintptr_t)xs[j] % 16) == 0 && ((uintptr_t)dxs[j] % 16) == 0;
#define L(TT, ...) hipLaunchKernelGGL(fuse_bwd_kernel<TT>, dim3(blocks_for(rows, 4)), dim3(256), 0, st, (const TT*)score, p, (const TT*)dout, (TT*)dscore, (long)rows, n, d, vec)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_fuse_modalities_bwd");
  return BIST_OK;
}

extern "C" int bist_cast_from_f32(const float* src, void* dst, int64_t n, int32_t dtype, void* stream) {
  BIST_REQUIRE(src && dst && n > 0, "bist_cast_from_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
#define L(TT, ...) hipLaunchKernelGGL(cast_from_f32_kernel<TT>, dim3(blocks_for(n, 256)), dim3(256), 0, st, src, (TT*)dst, (long)n)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_cast_from_f32");
  return BIST_OK;
}

extern "C" int bist_add_f32_into(const float* src, void* dst, int64_t n, int32_t dtype, void* stream) {
  BIST_REQUIRE(src && dst && n > 0, "bist_add_f32_into: bad argument");
  hipStream_t st = (hipStream_t)stream;
#define L(TT, ...) hipLaunchKernelGGL(add_f32_into_kernel<TT>, dim3(blocks_for(n, 256)), dim3(256), 0, st, src, (TT*)dst, (long)n)
  DISPATCH_T(dtype, L, 0)
#undef L
  BIST_LAUNCH_CHECK("bist_add_f32_into");
  return BIST_OK;
}

extern "C" int bist_adam_step(float* p, const void* g, float* m, float* v, void* work, int64_t n, float lr, float beta1, float beta2,
                              float eps, int32_t step, float grad_scale, int32_t grad_dtype, int32_t work_dtype, void* stream) {
  BIST_REQUIRE(p && g && m && v && n > 0 && step >= 1, "bist_adam_step: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const unsigned grid = blocks_for(n, 256);
#define ADAM(TG, TW) hipLaunchKernelGGL((adam_kernel<TG, TW>), dim3(grid), dim3(256), 0, st, p, (const TG*)g, m, v, (TW*)work, (long)n, lr, beta1, beta2, eps, bc1, bc2, grad_scale, (const float*)nullptr)
  if (grad_dtype == BIST_F32 && (work == nullptr || work_dtype == BIST_F32)) ADAM(float, float);
  else if (grad_dtype == BIST_F32 && work_dtype == BIST_BF16) ADAM(float, bf16_t);
  else if (grad_dtype == BIST_BF16 && (work == nullptr || work_dtype == BIST_BF16)) ADAM(bf16_t, bf16_t);
  else if (grad_dtype == BIST_BF16 && work_dtype == BIST_F32) ADAM(bf16_t, float);
  else { bist_set_error("bist_adam_step: bad dtype combination"); return BIST_EINVAL; }
#undef ADAM
  BIST_LAUNCH_CHECK("bist_adam_step");
  return BIST_OK;
}

// {lr, 1 - beta1^t, 1 - beta2^t, grad_scale} of optimiser step t = ctr[0], computed on the device (one thread, double precision)
__global__ void noam_hyper_kernel(const long long* __restrict__ ctr, float* __restrict__ hyper, double d_model, double factor, double warmup,
                                  double b1, double b2, float gscale) {
  const double t = (double)ctr[0];
  const double lr = factor * (pow(d_model, -0.5) * fmin(pow(t, -0.5), t * pow(warmup, -1.5)));
  hyper[0] = (float)lr;
  hyper[1] = (float)(1.0 - pow(b1, t));
  hyper[2] = (float)(1.0 - pow(b2, t));
  hyper[3] = gscale;
}

extern "C" int bist_noam_hyper(const int64_t* step_ctr, float* hyper, float d_model, float factor, float warmup, float beta1, float beta2,
                               float grad_scale, void* stream) {
  BIST_REQUIRE(step_ctr && hyper && d_model > 0.f && warmup > 0.f, "bist_noam_hyper: bad argument");
  hipLaunchKernelGGL(noam_hyper_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const long long*)step_ctr, hyper, (double)d_model,
                     (double)factor, (double)warmup, (double)beta1, (double)beta2, grad_scale);
  BIST_LAUNCH_CHECK("bist_noam_hyper");
  return BIST_OK;
}

extern "C" int bist_adam_step_dev_bg(float* p, const void* g, float* m, float* v, void* work, int64_t n, const float* hyper, float beta1,
                                     float beta2, float eps, int32_t grad_dtype, int32_t work_dtype, int32_t max_blocks, void* stream);
extern "C" int bist_adam_step_dev(float* p, const void* g, float* m, float* v, void* work, int64_t n, const float* hyper, float beta1,
                                  float beta2, float eps, int32_t grad_dtype, int32_t work_dtype, void* stream) {
  return bist_adam_step_dev_bg(p, g, m, v, work, n, hyper, beta1, beta2, eps, grad_dtype, work_dtype, 0, stream);
}
// max_blocks > 0: at most that many workgroups walk the range (a background update beside latency-bound launches); 0: one element group per thread
extern "C" int bist_adam_step_dev_bg(float* p, const void* g, float* m, float* v, void* work, int64_t n, const float* hyper, float beta1,
                                     float beta2, float eps, int32_t grad_dtype, int32_t work_dtype, int32_t max_blocks, void* stream) {
  BIST_REQUIRE(p && g && m && v && hyper && n > 0 && max_blocks >= 0, "bist_adam_step_dev: bad argument");
  hipStream_t st = (hipStream_t)stream;
  // whole 16-byte groups: four elements per thread
  const bool vec = n % 4 == 0 && ((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && (uintptr_t)g % (grad_dtype == BIST_BF16 ? 8 : 16) == 0 &&
                   (!work || (uintptr_t)work % (work_dtype == BIST_BF16 ? 8 : 16) == 0);
  unsigned grid = vec ? blocks_for(n / 4, 256) : blocks_for(n, 256);
  if (max_blocks > 0) {
    BIST_REQUIRE(vec, "bist_adam_step_dev_bg: the background form needs whole 16-byte groups (n %% 4 == 0, aligned pointers)");
    if (grid > (unsigned)max_blocks) grid = (unsigned)max_blocks;
  }
#define ADAM(TG, TW)                                                                                                                                  \
  do {                                                                                                                                                \
    if (vec) hipLaunchKernelGGL((adam4_kernel<TG, TW>), dim3(grid), dim3(256), 0, st, p, (const TG*)g, m, v, (TW*)work, (long)(n / 4), beta1, beta2, eps, hyper); \
    else hipLaunchKernelGGL((adam_kernel<TG, TW>), dim3(grid), dim3(256), 0, st, p, (const TG*)g, m, v, (TW*)work, (long)n, 0.f, beta1, beta2, eps, 1.f, 1.f, 1.f, hyper); \
  } while (0)
  if (grad_dtype == BIST_F32 && (work == nullptr || work_dtype == BIST_F32)) ADAM(float, float);
  else if (grad_dtype == BIST_F32 && work_dtype == BIST_BF16) ADAM(float, bf16_t);
  else if (grad_dtype == BIST_BF16 && (work == nullptr || work_dtype == BIST_BF16)) ADAM(bf16_t, bf16_t);
  else if (grad_dtype == BIST_BF16 && work_dtype == BIST_F32) ADAM(bf16_t, float);
  else { bist_set_error("bist_adam_step_dev: bad dtype combination"); return BIST_EINVAL; }
#undef ADAM
  BIST_LAUNCH_CHECK("bist_adam_step_dev");
  return BIST_OK;
}

extern "C" int bist_noam_hyper_pending(int64_t* pending, float* hyper, float d_model, float factor, float warmup, float beta1, float beta2,
                                       float grad_scale, void* stream) {
  BIST_REQUIRE(pending && hyper && d_model > 0.f && warmup > 0.f, "bist_noam_hyper_pending: bad argument");
  hipLaunchKernelGGL(noam_hyper_pending_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)pending, hyper, (double)d_model,
                     (double)factor, (double)warmup, (double)beta1, (double)beta2, grad_scale);
  BIST_LAUNCH_CHECK("bist_noam_hyper_pending");
  return BIST_OK;
}

extern "C" int bist_adam_apply_dev(float* p, void* g, float* m, float* v, void* work, int64_t n, const float* hyper, float beta1,
                                   float beta2, float eps, int32_t grad_dtype, int32_t work_dtype, void* stream) {
  BIST_REQUIRE(p && g && m && v && hyper && n > 0, "bist_adam_apply_dev: bad argument");
  BIST_REQUIRE(n % 4 == 0 && ((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && (uintptr_t)g % (grad_dtype == BIST_BF16 ? 8 : 16) == 0 &&
               (!work || (uintptr_t)work % (work_dtype == BIST_BF16 ? 8 : 16) == 0),
               "bist_adam_apply_dev: ranges must be multiples of 4 elements and 16-byte aligned (fp32) / 8-byte aligned (bf16)");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = blocks_for(n / 4, 256);
#define ADAM(TG, TW) hipLaunchKernelGGL((adam_apply4_kernel<TG, TW>), dim3(grid), dim3(256), 0, st, p, (TG*)g, m, v, (TW*)work, (long)(n / 4), beta1, beta2, eps, hyper)
  if (grad_dtype == BIST_F32 && (work == nullptr || work_dtype == BIST_F32)) ADAM(float, float);
  else if (grad_dtype == BIST_F32 && work_dtype == BIST_BF16) ADAM(float, bf16_t);
  else if (grad_dtype == BIST_BF16 && (work == nullptr || work_dtype == BIST_BF16)) ADAM(bf16_t, bf16_t);
  else if (grad_dtype == BIST_BF16 && work_dtype == BIST_F32) ADAM(bf16_t, float);
  else { bist_set_error("bist_adam_apply_dev: bad dtype combination"); return BIST_EINVAL; }
#undef ADAM
  BIST_LAUNCH_CHECK("bist_adam_apply_dev");
  return BIST_OK;
}
