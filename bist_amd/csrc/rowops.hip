// Row-wise and elementwise kernels of the BiST hot path (memory-bound; one wave per row).
#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// Reference LayerNorm (model/modules.py:28-31): a*(x-mean)/(std_unbiased+eps)+b.
// One 64-lane wave per row; the row is read three times (mean, centred sum of squares, output):
// the second and third reads hit L1/L2.  Two-pass variance, like torch.std.
// ---------------------------------------------------------------------------------------------
// Several LayerNorms of one geometry in ONE launch (bist_layernorm_fwd_multi): set = blockIdx.y picks (rows in, gain, offset, rows out).
// The t2s and s2t instances of a sublayer normalise same-shaped query tensors with different parameters (encoder.py:176,184).
constexpr int LN_SETS = 8;
struct LnSetsK { const void* x[LN_SETS]; const void* a[LN_SETS]; const void* b[LN_SETS]; void* y[LN_SETS]; };

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnSetsK sets, long rows, int d, long ldx, long ldy, float eps) {
  const T* __restrict__ x = (const T*)sets.x[blockIdx.y]; const T* __restrict__ a = (const T*)sets.a[blockIdx.y];
  const T* __restrict__ b = (const T*)sets.b[blockIdx.y]; T* __restrict__ y = (T*)sets.y[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += to_f(xr[c]);
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) { const float t = to_f(xr[c]) - mean; q += t * t; }
  const float stdv = sqrtf(wave_sum(q) / (float)(d - 1));
  const float inv = 1.f / (stdv + eps);
  T* yr = y + row * ldy;
  for (int c = lane; c < d; c += 64) yr[c] = from_f<T>(to_f(a[c]) * (to_f(xr[c]) - mean) * inv + to_f(b[c]));
}

// Vectorised variant for rows that are a whole number of KiB (d = 512 bf16, 256/512 f32, ...): every lane owns
// NV 16-byte vectors of the row, which is read from HBM exactly once and kept in registers for both
// reductions (two-pass variance, like torch.std) and the output.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const LnSetsK sets, long rows, int d, long ldx, long ldy, float eps) {
  const T* __restrict__ x = (const T*)sets.x[blockIdx.y]; const T* __restrict__ a = (const T*)sets.a[blockIdx.y];
  const T* __restrict__ b = (const T*)sets.b[blockIdx.y]; T* __restrict__ y = (T*)sets.y[blockIdx.y];
  constexpr int E = 16 / (int)sizeof(T);             // elements per 16-byte vector
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float v[NV * E];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const uint4 q = *reinterpret_cast<const uint4*>(xr + (j * 64 + lane) * E);
    const T* qe = reinterpret_cast<const T*>(&q);
#pragma unroll
    for (int e = 0; e < E; ++e) { v[j * E + e] = to_f(qe[e]); s += v[j * E + e]; }
  }
  const float mean = wave_sum(s) / (float)d;
  float q2 = 0.f;
#pragma unroll
  for (int u = 0; u < NV * E; ++u) { v[u] -= mean; q2 += v[u] * v[u]; }
  const float inv = 1.f / (sqrtf(wave_sum(q2) / (float)(d - 1)) + eps);
  T* yr = y + row * ldy;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c0 = (j * 64 + lane) * E;
    const uint4 qa = *reinterpret_cast<const uint4*>(a + c0), qb = *reinterpret_cast<const uint4*>(b + c0);
    const T* ae = reinterpret_cast<const T*>(&qa);
    const T* be = reinterpret_cast<const T*>(&qb);
    T o[E];
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = from_f<T>(to_f(ae[e]) * v[j * E + e] * inv + to_f(be[e]));
    *reinterpret_cast<uint4*>(yr + c0) = *reinterpret_cast<const uint4*>(o);
  }
}

// Embeddings*sqrt(d) + PositionalEncoding (modules.py:121-123, 141-144); pe is the f32 table.
template <typename T>
__global__ void embed_pe_kernel(const long* __restrict__ ids, const T* __restrict__ lut, const float* __restrict__ pe,
                                T* __restrict__ y, long rows, int L, int d, float scale, const DropArg drop) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * d) return;
  const long row = idx / d; const int c = (int)(idx % d);
  const long id = ids[row];
  float v = to_f(lut[id * d + c]) * scale + pe[(row % L) * (long)d + c];
  if (drop.p > 0.f) v *= drop_mul(drop.key(), (unsigned long long)idx, drop.p, drop.keep_scale());     // modules.py:144
  y[idx] = from_f<T>(v);
}

// temporal_mask[row] = (sum of the row's S*C features != 0)  (data/dataset.py:79); one 256-thread block per row
template <typename T>
__global__ __launch_bounds__(256) void temporal_mask_kernel(const T* __restrict__ f, unsigned char* __restrict__ m, long rows, long n) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const T* p = f + row * n;
  float s = 0.f;
  constexpr int E = 16 / (int)sizeof(T);
  if ((n % E) == 0 && ((uintptr_t)p % 16) == 0) {
    for (long c = threadIdx.x; c < n / E; c += 256) {
      const uint4 q = reinterpret_cast<const uint4*>(p)[c];
      const T* qe = reinterpret_cast<const T*>(&q);
#pragma unroll
      for (int e = 0; e < E; ++e) s += to_f(qe[e]);
    }
  } else {
    for (long c = threadIdx.x; c < n; c += 256) s += to_f(p[c]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) m[row] = (red[0] + red[1] + red[2] + red[3]) != 0.f ? 1 : 0;
}

// Dynamic modality fusion (decoder.py:155-159): out = sum_j softmax(score[row,:n])_j * x_j[row,:]
struct FusePtrs { const void* x[4]; };
template <typename T>
__global__ void fuse_kernel(const T* __restrict__ score, FusePtrs xs, T* __restrict__ out, long rows, int n, int d) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * d) return;
  const long row = idx / d;
  float sc[4], mx = -INFINITY, den = 0.f;
  for (int j = 0; j < n; ++j) { sc[j] = to_f(score[row * n + j]); mx = fmaxf(mx, sc[j]); }
  for (int j = 0; j < n; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
  float acc = 0.f;
  for (int j = 0; j < n; ++j) acc += (sc[j] / den) * to_f(reinterpret_cast<const T*>(xs.x[j])[idx]);
  out[idx] = from_f<T>(acc);
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = from_f<TD>(to_f(s[i]));
}

// out[i] = a[i] + b[i % nb]  (residual add; nb < n broadcasts b, e.g. the positional table)
template <typename T>
__global__ void add_bcast_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long n, long nb) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = from_f<T>(to_f(a[i]) + to_f(b[i % nb]));
}

// mode 0: out[i] = a[i] + drop(b[i % nb])      x + dropout(sublayer(norm(x)))   (SublayerConnection.forward, modules.py:44)
// mode 1: out[i] = drop(a[i] + b[i % nb])      dropout(x + pe)                  (PositionalEncoding.forward, modules.py:142-144)
// the mask of element i is drop_keep(key, i): bist_epilogue_bwd on the same contiguous tensor regenerates it for the backward
template <typename T>
__global__ void add_dropout_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long n, long nb, int mode,
                                   const DropArg dr) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = drop_mul(dr.key(), (unsigned long long)i, dr.p, dr.keep_scale());
  const float av = to_f(a[i]), bv = to_f(b[i % nb]);
  out[i] = from_f<T>(mode == 0 ? av + m * bv : m * (av + bv));
}

inline unsigned blocks_for(long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int bist_add_dropout_fwd(const void* a, const void* b, void* out, int64_t n, int64_t nb, int32_t mode, const BistDrop* drop,
                                    int32_t dtype, void* stream) {
  BIST_REQUIRE(a && b && out && n > 0 && nb > 0 && (mode == 0 || mode == 1), "bist_add_dropout_fwd: bad argument");
  BIST_REQUIRE(drop && drop->p > 0.f && drop->p < 1.f, "bist_add_dropout_fwd: drop p must be in (0, 1)");
  const DropArg dr = make_drop(drop);
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = blocks_for(n, 256);
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(add_dropout_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long)n, (long)nb, mode, dr);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(add_dropout_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, (long)n, (long)nb, mode, dr);
  else { bist_set_error("bist_add_dropout_fwd: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_add_dropout_fwd");
  return BIST_OK;
}

extern "C" int bist_layernorm_fwd_multi(const BistLnSet* sets, int32_t nsets, int64_t rows, int32_t d, int64_t ldx, int64_t ldy, float eps,
                                        int32_t dtype, void* stream) {
  BIST_REQUIRE(sets && nsets >= 1 && nsets <= LN_SETS, "bist_layernorm_fwd_multi: 1..%d sets", LN_SETS);
  BIST_REQUIRE(rows > 0 && d > 1 && ldx >= d && ldy >= d, "bist_layernorm_fwd: bad shape rows=%ld d=%d", (long)rows, d);
  hipStream_t st = (hipStream_t)stream;
  const long sz = dtype == BIST_BF16 ? 2 : 4;
  LnSetsK k{};
  bool al = (ldx * sz) % 16 == 0 && (ldy * sz) % 16 == 0;
  for (int i = 0; i < nsets; ++i) {
    BIST_REQUIRE(sets[i].x && sets[i].a && sets[i].b && sets[i].y, "bist_layernorm_fwd: null pointer (set %d)", i);
    k.x[i] = sets[i].x; k.a[i] = sets[i].a; k.b[i] = sets[i].b; k.y[i] = sets[i].y;
    al = al && (((uintptr_t)sets[i].x | (uintptr_t)sets[i].y | (uintptr_t)sets[i].a | (uintptr_t)sets[i].b) % 16 == 0);
  }
  const dim3 g(blocks_for(rows, 4), (unsigned)nsets);
  const long nv = (d * sz) % 1024 == 0 ? d * sz / 1024 : 0;     // 16-byte vectors per lane
  if (al && dtype == BIST_BF16 && nv == 1)
    hipLaunchKernelGGL((layernorm_vec_kernel<bf16_t, 1>), g, dim3(256), 0, st, k, (long)rows, d, (long)ldx, (long)ldy, eps);
  else if (al && dtype == BIST_F32 && nv == 1)
    hipLaunchKernelGGL((layernorm_vec_kernel<float, 1>), g, dim3(256), 0, st, k, (long)rows, d, (long)ldx, (long)ldy, eps);
  else if (al && dtype == BIST_F32 && nv == 2)
    hipLaunchKernelGGL((layernorm_vec_kernel<float, 2>), g, dim3(256), 0, st, k, (long)rows, d, (long)ldx, (long)ldy, eps);
  else if (dtype == BIST_BF16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, g, dim3(256), 0, st, k, (long)rows, d, (long)ldx, (long)ldy, eps);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(layernorm_kernel<float>, g, dim3(256), 0, st, k, (long)rows, d, (long)ldx, (long)ldy, eps);
  else { bist_set_error("bist_layernorm_fwd: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_layernorm_fwd");
  return BIST_OK;
}

extern "C" int bist_layernorm_fwd(const void* x, const void* a, const void* b, void* y, int64_t rows, int32_t d,
                                  int64_t ldx, int64_t ldy, float eps, int32_t dtype, void* stream) {
  const BistLnSet one{x, a, b, y};
  return bist_layernorm_fwd_multi(&one, 1, rows, d, ldx, ldy, eps, dtype, stream);
}

extern "C" int bist_embed_pe_fwd(const int64_t* ids, const void* lut, const float* pe, void* y, int64_t rows, int32_t L,
                                 int32_t d, const BistDrop* drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(ids && lut && pe && y, "bist_embed_pe_fwd: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_embed_pe_fwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  BIST_REQUIRE(rows > 0 && L > 0 && d > 0, "bist_embed_pe_fwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const float scale = sqrtf((float)d);
  const unsigned g = blocks_for(rows * d, 256);
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(embed_pe_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const long*)ids, (const bf16_t*)lut, pe, (bf16_t*)y, rows, L, d, scale, dr);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(embed_pe_kernel<float>, dim3(g), dim3(256), 0, st, (const long*)ids, (const float*)lut, pe, (float*)y, rows, L, d, scale, dr);
  else { bist_set_error("bist_embed_pe_fwd: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_embed_pe_fwd");
  return BIST_OK;
}

extern "C" int bist_temporal_mask(const void* fts, uint8_t* mask, int64_t BT, int64_t row_elems, int32_t dtype, void* stream) {
  BIST_REQUIRE(fts && mask && BT > 0 && row_elems > 0, "bist_temporal_mask: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)BT;
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(temporal_mask_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)fts, mask, BT, row_elems);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(temporal_mask_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)fts, mask, BT, row_elems);
  else { bist_set_error("bist_temporal_mask: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_temporal_mask");
  return BIST_OK;
}

extern "C" int bist_fuse_modalities(const void* score, const void* const* xs, void* out, int64_t rows, int32_t n, int32_t d,
                                    int32_t dtype, void* stream) {
  BIST_REQUIRE(score && xs && out && rows > 0 && d > 0 && n >= 1 && n <= 4, "bist_fuse_modalities: bad argument");
  FusePtrs p;
  for (int j = 0; j < 4; ++j) p.x[j] = j < n ? xs[j] : nullptr;
  for (int j = 0; j < n; ++j) BIST_REQUIRE(p.x[j], "bist_fuse_modalities: null input %d", j);
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = blocks_for(rows * d, 256);
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(fuse_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)score, p, (bf16_t*)out, rows, n, d);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(fuse_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)score, p, (float*)out, rows, n, d);
  else { bist_set_error("bist_fuse_modalities: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_fuse_modalities");
  return BIST_OK;
}

namespace {
struct AddNPtrs { const void* p[BIST_ADD_N_MAX]; };
// 8 (bf16) or 4 (f32) elements per thread, 16-byte loads from every source, fp32 sums
template <typename T>
__global__ __launch_bounds__(256) void add_n_kernel(const AddNPtrs src, int n, T* __restrict__ out, long numel) {
  constexpr int VW = 16 / (int)sizeof(T);
  const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VW;
  if (i0 >= numel) return;
  float acc[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) acc[e] = 0.f;
  if (i0 + VW <= numel) {
    for (int j = 0; j < n; ++j) {
      T v[VW];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(src.p[j]) + i0);
#pragma unroll
      for (int e = 0; e < VW; ++e) acc[e] += to_f(v[e]);
    }
    T o[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) o[e] = from_f<T>(acc[e]);
    *reinterpret_cast<uint4*>(out + i0) = *reinterpret_cast<const uint4*>(o);
  } else {
    for (int e = 0; i0 + e < numel; ++e) {
      float a = 0.f;
      for (int j = 0; j < n; ++j) a += to_f(reinterpret_cast<const T*>(src.p[j])[i0 + e]);
      out[i0 + e] = from_f<T>(a);
    }
  }
}
}  // namespace

namespace {
template <typename T>
__global__ __launch_bounds__(256) void permute_ts_kernel(const T* __restrict__ x, T* __restrict__ y, int T_, int S_, int d, long total16) {
  constexpr int E = 16 / (int)sizeof(T);
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total16) return;
  const int cpr = d / E;
  const int c = (int)(idx % cpr); long r = idx / cpr;
  const int s = (int)(r % S_); r /= S_;
  const int t = (int)(r % T_); const long b = r / T_;
  const uint4 v = *reinterpret_cast<const uint4*>(x + idx * E);
  *reinterpret_cast<uint4*>(y + (((b * S_ + s) * T_ + t) * (long)d + (long)c * E)) = v;
}
}  // namespace

extern "C" int bist_permute_ts(const void* x, void* y, int32_t B, int32_t T, int32_t S, int32_t d, int32_t dtype, void* stream) {
  BIST_REQUIRE(x && y && B > 0 && T > 0 && S > 0 && d > 0, "bist_permute_ts: bad argument");
  const long esz = dtype == BIST_BF16 ? 2 : 4;
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && (d * esz) % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
               "bist_permute_ts: rows must be whole 16-byte pieces, 16-byte aligned");
  const long total16 = (long)B * T * S * (d * esz / 16);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BIST_BF16) hipLaunchKernelGGL(permute_ts_kernel<bf16_t>, dim3(blocks_for(total16, 256)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, T, S, d, total16);
  else hipLaunchKernelGGL(permute_ts_kernel<float>, dim3(blocks_for(total16, 256)), dim3(256), 0, st, (const float*)x, (float*)y, T, S, d, total16);
  BIST_LAUNCH_CHECK("bist_permute_ts");
  return BIST_OK;
}

extern "C" int bist_add_n(const void* const* srcs, int32_t n, void* out, int64_t numel, int32_t dtype, void* stream) {
  BIST_REQUIRE(srcs && out && n >= 1 && n <= BIST_ADD_N_MAX && numel > 0, "bist_add_n: bad argument (n = %d)", (int)n);
  AddNPtrs p{};
  for (int j = 0; j < n; ++j) {
    BIST_REQUIRE(srcs[j] && ((uintptr_t)srcs[j] % 16) == 0, "bist_add_n: source %d null or not 16-byte aligned", j);
    p.p[j] = srcs[j];
  }
  BIST_REQUIRE(((uintptr_t)out % 16) == 0, "bist_add_n: out not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(add_n_kernel<bf16_t>, dim3(blocks_for((numel + 7) / 8, 256)), dim3(256), 0, st, p, n, (bf16_t*)out, (long)numel);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(add_n_kernel<float>, dim3(blocks_for((numel + 3) / 4, 256)), dim3(256), 0, st, p, n, (float*)out, (long)numel);
  else { bist_set_error("bist_add_n: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_add_n");
  return BIST_OK;
}

extern "C" int bist_cast(const void* src, void* dst, int64_t n, int32_t sd, int32_t dd, void* stream) {
  BIST_REQUIRE(src && dst && n > 0, "bist_cast: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = blocks_for(n, 256);
  if (sd == BIST_F32 && dd == BIST_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (sd == BIST_BF16 && dd == BIST_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
  else { bist_set_error("bist_cast: unsupported dtype pair %d -> %d", sd, dd); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_cast");
  return BIST_OK;
}

extern "C" int bist_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t nb, int32_t dtype, void* stream) {
  BIST_REQUIRE(a && b && out && n > 0 && nb > 0, "bist_add_bcast: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = blocks_for(n, 256);
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(add_bcast_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n, nb);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(add_bcast_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n, nb);
  else { bist_set_error("bist_add_bcast: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_add_bcast");
  return BIST_OK;
}
