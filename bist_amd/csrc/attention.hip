// Attention cores of the BiST hot path (everything between the projection GEMMs).
//
//  mha_core   : generic small attention (query self-attention, caption/decoder/pointer attention);
//               one wave per (n, head, query row).
//  st1_pv     : stage 1 of t2s / s2t after the folded-query score GEMM: softmax over the key
//               axis of each (clip, spatial) or (clip, temporal) group + P.V, per (clip, head).
//  st2        : stage 2 of both directions with K and V folded out: one workgroup per (clip,
//               query position) reads only the stage-1 outputs.
// All softmaxes use the reference's masking rule: masked scores are REPLACED by -1e9
// (model/modules.py:60), never -inf, so a fully masked row yields the uniform distribution.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr float MASK_FILL = -1e9f;

// ---------------------------------------------------------------------------------------------
// mha_core: block = 4 waves, each wave owns one (n, hh, i) row.
// LDS per wave: q (dk floats) then scores/probabilities (Lk floats).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mha_core_kernel(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V,
                                                       const unsigned char* __restrict__ mask, T* __restrict__ O, float* __restrict__ P,
                                                       int N, int Lq, int Lk, int h, int dk, long ldq, long ldk, long ldv, long ldo,
                                                       long q_bs, long k_bs, long v_bs, long o_bs, long mask_bs, long mask_qs, float scale,
                                                       const DropArg drop, int kvec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * 4 + w;
  const long total = (long)N * h * Lq;
  float* qs = smem + (long)w * (dk + Lk);
  float* ps = qs + dk;
  const bool active = item < total;
  int n = 0, hh = 0, i = 0;
  if (active) { i = (int)(item % Lq); const long t = item / Lq; hh = (int)(t % h); n = (int)(t / h); }
  if (active) {
    const T* q = Q + n * q_bs + (long)i * ldq + hh * dk;
    for (int c = lane; c < dk; c += 64) qs[c] = to_f(q[c]);
  }
  __syncthreads();
  if (!active) return;
  const T* Kn = K + n * k_bs + hh * dk;
  const T* Vn = V + n * v_bs + hh * dk;
  const unsigned char* mrow = mask ? mask + n * mask_bs + (long)i * mask_qs : nullptr;
  float mx = -INFINITY;
  for (int j = lane; j < Lk; j += 64) {
    const T* kr = Kn + (long)j * ldk;
    float s = 0.f;
    if (kvec) {                                           // 16-byte pieces of the key row (was one 2-byte load per element)
      constexpr int E = 16 / (int)sizeof(T);
      for (int c = 0; c < dk; c += E) {
        T k8[E];
        *reinterpret_cast<uint4*>(k8) = *reinterpret_cast<const uint4*>(kr + c);
#pragma unroll
        for (int e = 0; e < E; ++e) s += qs[c + e] * to_f(k8[e]);
      }
    } else {
      for (int c = 0; c < dk; ++c) s += qs[c] * to_f(kr[c]);
    }
    s *= scale;
    if (mrow && mrow[j] == 0) s = MASK_FILL;
    ps[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float den = 0.f;
  for (int j = lane; j < Lk; j += 64) { const float e = expf(ps[j] - mx); ps[j] = e; den += e; }
  den = wave_sum(den);
  const float inv = 1.f / den;
  for (int j = lane; j < Lk; j += 64) ps[j] *= inv;
  __builtin_amdgcn_wave_barrier();
  if (P) {
    float* pr = P + (((long)n * h + hh) * Lq + i) * Lk;
    for (int j = lane; j < Lk; j += 64) pr[j] = ps[j];
  }
  if (drop.p > 0.f) {                                   // p_attn = dropout(p_attn), modules.py:62-63
    const unsigned long long key = drop.key(), base = (unsigned long long)item * Lk;
    const float ks = drop.keep_scale();
    for (int j = lane; j < Lk; j += 64) ps[j] *= drop_mul(key, base + j, drop.p, ks);
    __builtin_amdgcn_wave_barrier();
  }
  T* o = O + n * o_bs + (long)i * ldo + hh * dk;
  for (int c = lane; c < dk; c += 64) {
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j) acc += ps[j] * to_f(Vn[(long)j * ldv + c]);
    o[c] = from_f<T>(acc);
  }
}

// ---------------------------------------------------------------------------------------------
// st1_pv: grid (chunks of groups, h, B).  For a chunk of Gc groups of clip b and head hh:
//   load sc[i][gl][k] = scores[b, i*h+hh, (g,k)] (+mask) into LDS, softmax over k, then
//   O[b,g,i,hh*dk+c] = sum_k P[i][gl][k] * V[b,(g,k),hh*dk+c].
// direction 0 (t2s): g = s, k = t -> score column t*S+s, V row t*S+s, mask tmask[b,t].
// direction 1 (s2t): g = t, k = s -> score column t*S+s (contiguous in k), no mask.
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
__global__ __launch_bounds__(256) void st1_pv_kernel(const TS* __restrict__ scores, const T* __restrict__ V,
                                                     const unsigned char* __restrict__ tmask, T* __restrict__ O,
                                                     int T_, int S_, int Lq, int h, int dk, long ldv, int dir, int Gc, const DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = dir == 0 ? S_ : T_, Kn = dir == 0 ? T_ : S_;
  const int KP = Kn + 1;                                  // padded row: conflict-free column walks
  const int g0 = blockIdx.x * Gc, hh = blockIdx.y, b = blockIdx.z;
  const int gc = min(Gc, G - g0);
  const int tid = threadIdx.x;
  const long TS_ = (long)T_ * S_;
  const int d = h * dk;
  const TS* sc = scores + (long)b * Lq * h * TS_;
  const unsigned char* mk = tmask ? tmask + (long)b * Kn : nullptr;       // key mask [B, K]: frames (t2s) -- or the keys of a permuted call

  // phase 1: gather the score slab (coalesced along the contiguous axis of each direction)
  const int total = Lq * gc * Kn;
  for (int idx = tid; idx < total; idx += 256) {
    int i, gl, k;
    if (dir == 0) { gl = idx % gc; const int t2 = idx / gc; k = t2 % Kn; i = t2 / Kn; }
    else          { k = idx % Kn; const int t2 = idx / Kn; gl = t2 % gc; i = t2 / gc; }
    const int g = g0 + gl;
    const long col = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
    float v = to_f(sc[((long)i * h + hh) * TS_ + col]);
    if (mk && mk[k] == 0) v = MASK_FILL;
    smem[((long)i * gc + gl) * KP + k] = v;
  }
  __syncthreads();
  // phase 2: one thread per (i, gl) row
  for (int r = tid; r < Lq * gc; r += 256) {
    float* p = smem + (long)r * KP;
    float mx = -INFINITY;
    for (int k = 0; k < Kn; ++k) mx = fmaxf(mx, p[k]);
    float den = 0.f;
    for (int k = 0; k < Kn; ++k) { const float e = expf(p[k] - mx); p[k] = e; den += e; }
    const float inv = 1.f / den;
    if (drop.p > 0.f) {
      const int i = r / gc, gl = r - i * gc;
      const unsigned long long key = drop.key();
      const unsigned long long base = ((((unsigned long long)b * G + (g0 + gl)) * h + hh) * Lq + i) * Kn;
      const float ks = drop.keep_scale();
      for (int k = 0; k < Kn; ++k) p[k] *= inv * drop_mul(key, base + k, drop.p, ks);
    } else {
      for (int k = 0; k < Kn; ++k) p[k] *= inv;
    }
  }
  __syncthreads();
  // phase 3: P.V -- work item = (gl, c); 8 query rows at a time in registers
  const T* Vb = V + (long)b * TS_ * ldv + hh * dk;
  for (int item = tid; item < gc * dk; item += 256) {
    const int gl = item / dk, c = item % dk;
    const int g = g0 + gl;
    for (int i0 = 0; i0 < Lq; i0 += 8) {
      float acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = 0.f;
      for (int k = 0; k < Kn; ++k) {
        const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
        const float v = to_f(Vb[row * ldv + c]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i0 + u < Lq) acc[u] += smem[((long)(i0 + u) * gc + gl) * KP + k] * v;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < Lq) O[(((long)b * G + g) * Lq + (i0 + u)) * d + hh * dk + c] = from_f<T>(acc[u]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// st2: one workgroup per (b, i).  LDS: q2f[h][d] floats, sc[h][G] floats.  h <= 16.
// ---------------------------------------------------------------------------------------------
constexpr int ST2_MAXH = 16;
template <typename T>
__global__ __launch_bounds__(256) void st2_kernel(const T* __restrict__ q2f, const T* __restrict__ Y,
                                                  const unsigned char* __restrict__ gmask, T* __restrict__ PY,
                                                  float* __restrict__ rowsum, int G, int Lq, int h, int d, const DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qf = smem;                 // [h][d]
  float* sc = smem + (long)h * d;   // [h][G]
  const int i = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const T* q = q2f + ((long)b * Lq + i) * h * d;
  for (int e = tid; e < h * d; e += 256) qf[e] = to_f(q[e]);
  __syncthreads();
  const T* Yb = Y + ((long)b * G * Lq + i) * d;          // row g at + g*Lq*d
  const long ystride = (long)Lq * d;
  const unsigned char* mk = gmask ? gmask + (long)b * G : nullptr;
  // phase 1: sc[hh][g] = q2f[hh,:] . Y[g,:]   (wave w takes g = w, w+4, ...)
  for (int g = w; g < G; g += 4) {
    const T* yr = Yb + g * ystride;
    float acc[ST2_MAXH];
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh) acc[hh] = 0.f;
    for (int e = lane; e < d; e += 64) {
      const float y = to_f(yr[e]);
#pragma unroll
      for (int hh = 0; hh < ST2_MAXH; ++hh)
        if (hh < h) acc[hh] += qf[hh * d + e] * y;
    }
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh) {
      if (hh < h) {
        float s = wave_sum(acc[hh]);
        if (mk && mk[g] == 0) s = MASK_FILL;
        if (lane == 0) sc[hh * G + g] = s;
      }
    }
  }
  __syncthreads();
  // phase 2: softmax over g (wave per head)
  for (int hh = w; hh < h; hh += 4) {
    float* p = sc + hh * G;
    float mx = -INFINITY;
    for (int g = lane; g < G; g += 64) mx = fmaxf(mx, p[g]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int g = lane; g < G; g += 64) { const float e = expf(p[g] - mx); p[g] = e; den += e; }
    den = wave_sum(den);
    const float inv = 1.f / den;
    float rs = 0.f;
    if (drop.p > 0.f) {
      const unsigned long long key = drop.key(), base = (((unsigned long long)b * Lq + i) * h + hh) * G;
      const float ks = drop.keep_scale();
      for (int g = lane; g < G; g += 64) { p[g] *= inv * drop_mul(key, base + g, drop.p, ks); rs += p[g]; }
    } else {
      for (int g = lane; g < G; g += 64) { p[g] *= inv; rs += p[g]; }
    }
    rs = wave_sum(rs);
    if (rowsum && lane == 0) rowsum[((long)b * Lq + i) * h + hh] = rs;
  }
  __syncthreads();
  // phase 3: PY[hh][e] = sum_g P[hh][g] Y[g][e]
  T* out = PY + ((long)b * Lq + i) * h * d;
  for (int e = tid; e < d; e += 256) {
    float acc[ST2_MAXH];
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh) acc[hh] = 0.f;
    for (int g = 0; g < G; ++g) {
      const float y = to_f(Yb[g * ystride + e]);
#pragma unroll
      for (int hh = 0; hh < ST2_MAXH; ++hh)
        if (hh < h) acc[hh] += sc[hh * G + g] * y;
    }
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh)
      if (hh < h) out[hh * d + e] = from_f<T>(acc[hh]);
  }
}

}  // namespace

extern "C" int bist_mha_core_fwd(const void* Q, const void* K, const void* V, const uint8_t* mask, void* O, float* p_attn,
                                 int32_t N, int32_t Lq, int32_t Lk, int32_t h, int32_t dk, int64_t ldq, int64_t ldk, int64_t ldv,
                                 int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs, int64_t mask_bs,
                                 int64_t mask_qs, float scale, const BistDrop* drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(Q && K && V && O, "bist_mha_core_fwd: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_mha_core_fwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  BIST_REQUIRE(N > 0 && Lq > 0 && Lk > 0 && h > 0 && dk > 0, "bist_mha_core_fwd: bad shape");
  const size_t lds = (size_t)4 * (dk + Lk) * sizeof(float);
  BIST_REQUIRE(lds <= 64 * 1024, "bist_mha_core_fwd: dk+Lk=%d too large for the LDS row buffer", dk + Lk);
  hipStream_t st = (hipStream_t)stream;
  const long items = (long)N * h * Lq;
  const unsigned g = (unsigned)((items + 3) / 4);
  const long esz = dtype == BIST_BF16 ? 2 : 4, piece = 16 / esz;
  const int kvec = ((uintptr_t)K % 16 == 0) && ldk % piece == 0 && k_bs % piece == 0 && dk % piece == 0;
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(mha_core_kernel<bf16_t>, dim3(g), dim3(256), lds, st, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, mask,
                       (bf16_t*)O, p_attn, N, Lq, Lk, h, dk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, dr, kvec);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(mha_core_kernel<float>, dim3(g), dim3(256), lds, st, (const float*)Q, (const float*)K, (const float*)V, mask,
                       (float*)O, p_attn, N, Lq, Lk, h, dk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, dr, kvec);
  else { bist_set_error("bist_mha_core_fwd: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_mha_core_fwd");
  bist_count_launch(BIST_K_MHA_FWD);
  return BIST_OK;
}

extern "C" int bist_st_stage1_pv_fwd(const void* scores, const void* V, const uint8_t* tmask, void* O, int32_t B, int32_t T,
                                     int32_t S, int32_t Lq, int32_t h, int32_t dk, int64_t ldv, int32_t direction,
                                     const BistDrop* drop, int32_t sc_dtype, int32_t dtype, void* stream) {
  BIST_REQUIRE(scores && V && O, "bist_st_stage1_pv_fwd: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_st_stage1_pv_fwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  BIST_REQUIRE(B > 0 && T > 0 && S > 0 && Lq > 0 && h > 0 && dk > 0, "bist_st_stage1_pv_fwd: bad shape");
  BIST_REQUIRE(direction == 0 || direction == 1, "bist_st_stage1_pv_fwd: direction must be 0 (t2s) or 1 (s2t)");
  BIST_REQUIRE(sc_dtype == BIST_F32 || sc_dtype == dtype, "bist_st_stage1_pv_fwd: scores must be f32 or the value dtype");
  BIST_REQUIRE(ldv >= (int64_t)h * dk, "bist_st_stage1_pv_fwd: ldv too small");
  static const bool valu_st1 = getenv("BIST_ST1_VALU") != nullptr;      // tuning aid, read once
  if (dtype == BIST_BF16 && !valu_st1) {          // matrix-core path (attention_mfma.hip)
    const int r = bist_st1_mfma(scores, sc_dtype == BIST_F32, V, tmask, O, nullptr, nullptr, 0, nullptr, B, T, S, Lq, h, dk, ldv, 0,
                                direction, 0, dr, (hipStream_t)stream);
    if (r == 1) return BIST_OK;
    if (r < 0) return BIST_ELAUNCH;      // (the matrix-core launcher left the reason in bist_last_error)
  }
  bist_count_launch(BIST_K_ST1_VALU);
  const int G = direction == 0 ? S : T, Kn = direction == 0 ? T : S;
  // groups per workgroup: as many as fit a 60 KiB slab, but keep >= ~2 workgroups per CU in flight
  int Gc = (int)((60 * 1024) / ((long)Lq * (Kn + 1) * sizeof(float)));
  BIST_REQUIRE(Gc >= 1, "bist_st_stage1_pv_fwd: Lq*K = %d*%d does not fit the LDS slab", Lq, Kn);
  if (Gc > G) Gc = G;
  while (Gc > 1 && (long)((G + Gc - 1) / Gc) * h * B < 512) --Gc;
  const size_t lds = (size_t)Lq * Gc * (Kn + 1) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((G + Gc - 1) / Gc), (unsigned)h, (unsigned)B);
#define ST1_LAUNCH(TT, TSC)                                                                                             \
  hipLaunchKernelGGL((st1_pv_kernel<TT, TSC>), grid, dim3(256), lds, st, (const TSC*)scores, (const TT*)V, tmask, (TT*)O, \
                     T, S, Lq, h, dk, ldv, direction, Gc, dr)
  if (dtype == BIST_BF16 && sc_dtype == BIST_BF16) ST1_LAUNCH(bf16_t, bf16_t);
  else if (dtype == BIST_BF16 && sc_dtype == BIST_F32) ST1_LAUNCH(bf16_t, float);
  else if (dtype == BIST_F32) ST1_LAUNCH(float, float);
  else { bist_set_error("bist_st_stage1_pv_fwd: bad dtype %d", dtype); return BIST_EINVAL; }
#undef ST1_LAUNCH
  BIST_LAUNCH_CHECK("bist_st_stage1_pv_fwd");
  return BIST_OK;
}

extern "C" int bist_st_stage2_fwd(const void* q2f, const void* Y, const uint8_t* gmask, void* PY, float* rowsum, int32_t B, int32_t G,
                                  int32_t Lq, int32_t h, int32_t d, const BistDrop* drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(q2f && Y && PY, "bist_st_stage2_fwd: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_st_stage2_fwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  BIST_REQUIRE(dr.p == 0.f || rowsum, "bist_st_stage2_fwd: dropout needs the rowsum output (the value bias is scaled by it)");
  BIST_REQUIRE(B > 0 && G > 0 && Lq > 0 && h > 0 && d > 0, "bist_st_stage2_fwd: bad shape");
  BIST_REQUIRE(h <= ST2_MAXH, "bist_st_stage2_fwd: at most %d heads", ST2_MAXH);
  static const bool valu_st2 = getenv("BIST_ST2_VALU") != nullptr;      // tuning aid, read once
  if (dtype == BIST_BF16 && !valu_st2) {          // matrix-core path (attention_mfma.hip)
    const int r = bist_st2_mfma(q2f, Y, gmask, PY, nullptr, nullptr, nullptr, rowsum, nullptr, B, G, Lq, h, d, 0, dr, (hipStream_t)stream);
    if (r == 1) return BIST_OK;
    if (r < 0) return BIST_ELAUNCH;      // (the matrix-core launcher left the reason in bist_last_error)
  }
  bist_count_launch(BIST_K_ST2_VALU);
  const size_t lds = ((size_t)h * d + (size_t)h * G) * sizeof(float);
  BIST_REQUIRE(lds <= 64 * 1024, "bist_st_stage2_fwd: h*(d+G) too large for LDS");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)Lq, (unsigned)B);
  if (dtype == BIST_BF16)
    hipLaunchKernelGGL(st2_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)q2f, (const bf16_t*)Y, gmask, (bf16_t*)PY, rowsum, G, Lq, h, d, dr);
  else if (dtype == BIST_F32)
    hipLaunchKernelGGL(st2_kernel<float>, grid, dim3(256), lds, st, (const float*)q2f, (const float*)Y, gmask, (float*)PY, rowsum, G, Lq, h, d, dr);
  else { bist_set_error("bist_st_stage2_fwd: bad dtype %d", dtype); return BIST_EINVAL; }
  BIST_LAUNCH_CHECK("bist_st_stage2_fwd");
  return BIST_OK;
}
