// Backward of the attention cores (training path).  Probabilities are recomputed from the saved
// inputs (scores or Q/K), never stored by the forward pass.  Softmax backward:
//   dS = P * (dP - sum_k P dP), and dS = 0 where the score was replaced by the -1e9 mask fill.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr float MASK_FILL = -1e9f;

// ---------------------------------------------------------------------------------------------
// mha_core backward: one workgroup per (n, head).  LDS: P[Lq][Lk], dS[Lq][Lk] (floats).
//   dP = dO V^T (+ dP_ext, the gradient arriving through the `.attn` side channel)
//   dQ = scale * dS K ; dK = scale * dS^T Q ; dV = P^T dO
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mha_core_bwd_kernel(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V,
                                                           const unsigned char* __restrict__ mask, const T* __restrict__ dO,
                                                           const float* __restrict__ dPext, T* __restrict__ dQ, T* __restrict__ dK,
                                                           T* __restrict__ dV, int Lq, int Lk, int h, int dk, long ldq, long ldk,
                                                           long ldv, long ldo, long q_bs, long k_bs, long v_bs, long o_bs,
                                                           long lddq, long lddk, long lddv, long dq_bs, long dk_bs, long dv_bs,
                                                           long mask_bs, long mask_qs, float scale, const DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* P = smem;
  float* dS = smem + (long)Lq * Lk;
  const unsigned long long dkey = drop.p > 0.f ? drop.key() : 0ULL;
  const float dks = drop.p > 0.f ? drop.keep_scale() : 1.f;
  const int hh = blockIdx.x, n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const T* Qn = Q + n * q_bs + hh * dk;
  const T* Kn = K + n * k_bs + hh * dk;
  const T* Vn = V + n * v_bs + hh * dk;
  const T* dOn = dO ? dO + n * o_bs + hh * dk : nullptr;
  // phase A: wave per query row; the lanes stride the head dimension (coalesced) and every (i, j) dot product is a
  // wave reduction -- with dk = 512 (the single-head pointer attention) a lane-per-key loop would be 512 serial,
  // uncoalesced steps.  Every column-chunk workgroup (blockIdx.z) repeats this phase; it is small.
  for (int i = w; i < Lq; i += 4) {
    const unsigned char* mrow = mask ? mask + n * mask_bs + (long)i * mask_qs : nullptr;
    float* p = P + (long)i * Lk;
    float* ds = dS + (long)i * Lk;
    // the query row and its output gradient live in registers (lane owns columns lane, lane+64, ...: dk <= 512);
    // per key all 2*DKC loads are issued together, so a (i, j) pair costs one memory latency, not dk/64 of them
    constexpr int DKC = 8;
    float qv[DKC], gv[DKC];
#pragma unroll
    for (int t = 0; t < DKC; ++t) {
      const int c = lane + t * 64;
      qv[t] = c < dk ? to_f(Qn[(long)i * ldq + c]) : 0.f;
      gv[t] = (dOn && c < dk) ? to_f(dOn[(long)i * ldo + c]) : 0.f;
    }
    for (int j = 0; j < Lk; ++j) {
      const T* kr = Kn + (long)j * ldk;
      const T* vr = Vn + (long)j * ldv;
      float kv[DKC], vv[DKC];
#pragma unroll
      for (int t = 0; t < DKC; ++t) {
        const int c = lane + t * 64;
        kv[t] = c < dk ? to_f(kr[c]) : 0.f;
        vv[t] = (dOn && c < dk) ? to_f(vr[c]) : 0.f;
      }
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int t = 0; t < DKC; ++t) { s += qv[t] * kv[t]; dp += gv[t] * vv[t]; }
      for (int c = lane + DKC * 64; c < dk; c += 64) {          // dk > 512: the rest the slow way
        s += to_f(Qn[(long)i * ldq + c]) * to_f(kr[c]);
        if (dOn) dp += to_f(dOn[(long)i * ldo + c]) * to_f(vr[c]);
      }
      s = wave_sum(s); dp = wave_sum(dp);
      if (lane == 0) {
        s *= scale;
        if (mrow && mrow[j] == 0) s = MASK_FILL;
        if (drop.p > 0.f) dp *= drop_mul(dkey, (((unsigned long long)n * h + hh) * Lq + i) * Lk + j, drop.p, dks);   // dP = mask/(1-p) dP'
        if (dPext) dp += dPext[(((long)n * h + hh) * Lq + i) * Lk + j];
        p[j] = s; ds[j] = dp;
      }
    }
    float mx = -INFINITY;
    for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, p[j]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int j = lane; j < Lk; j += 64) { const float e = expf(p[j] - mx); p[j] = e; den += e; }
    den = wave_sum(den);
    const float inv = 1.f / den;
    float dot = 0.f;
    for (int j = lane; j < Lk; j += 64) { p[j] *= inv; dot += p[j] * ds[j]; }
    dot = wave_sum(dot);
    for (int j = lane; j < Lk; j += 64) {
      float g = p[j] * (ds[j] - dot);
      if (mrow && mrow[j] == 0) g = 0.f;
      ds[j] = g * scale;
      if (drop.p > 0.f) p[j] *= drop_mul(dkey, (((unsigned long long)n * h + hh) * Lq + i) * Lk + j, drop.p, dks);    // P' feeds dV
    }
  }
  __syncthreads();
  // phase B: this workgroup's chunk of the head columns [c0, c0 + cw)
  const int cw = dk / (int)gridDim.z, c0 = (int)blockIdx.z * cw;
  for (int item = tid; item < Lq * cw; item += 256) {       // dQ[i,c]
    const int i = item / cw, c = c0 + item % cw;
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j) acc += dS[(long)i * Lk + j] * to_f(Kn[(long)j * ldk + c]);
    dQ[n * dq_bs + (long)i * lddq + hh * dk + c] = from_f<T>(acc);
  }
  for (int item = tid; item < Lk * cw; item += 256) {       // dK[j,c], dV[j,c]
    const int j = item / cw, c = c0 + item % cw;
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < Lq; ++i) {
      ak += dS[(long)i * Lk + j] * to_f(Qn[(long)i * ldq + c]);
      if (dOn) av += P[(long)i * Lk + j] * to_f(dOn[(long)i * ldo + c]);
    }
    dK[n * dk_bs + (long)j * lddk + hh * dk + c] = from_f<T>(ak);
    dV[n * dv_bs + (long)j * lddv + hh * dk + c] = from_f<T>(av);
  }
}

// ---------------------------------------------------------------------------------------------
// st1_pv backward: grid (group chunks, h, B), same slab decomposition as the forward.
// LDS: P[i][gl][K+1], D[i][gl][K+1] (dP then dS), dOs[gl][i][dk]   (floats)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void st1_pv_bwd_kernel(const float* __restrict__ scores, const T* __restrict__ V,
                                                         const unsigned char* __restrict__ tmask, const T* __restrict__ dO,
                                                         void* __restrict__ dscores, int dsc_bf16, T* __restrict__ dV, int T_, int S_, int Lq,
                                                         int h, int dk, long ldv, long lddv, int dir, int Gc, const DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = dir == 0 ? S_ : T_, Kn = dir == 0 ? T_ : S_;
  const int KP = Kn + 1;
  const unsigned long long dkey = drop.p > 0.f ? drop.key() : 0ULL;
  const float dks = drop.p > 0.f ? drop.keep_scale() : 1.f;
  const int g0 = blockIdx.x * Gc, hh = blockIdx.y, b = blockIdx.z;
  const int gc = min(Gc, G - g0);
  const int tid = threadIdx.x;
  const long TS_ = (long)T_ * S_;
  const int d = h * dk;
  float* P = smem;
  float* D = smem + (long)Lq * Gc * KP;
  float* dOs = D + (long)Lq * Gc * KP;
  const float* sc = scores + (long)b * Lq * h * TS_;
  float* dsc = reinterpret_cast<float*>(dscores) + (long)b * Lq * h * TS_;
  bf16_t* dsc16 = reinterpret_cast<bf16_t*>(dscores) + (long)b * Lq * h * TS_;
  const unsigned char* mk = tmask ? tmask + (long)b * Kn : nullptr;       // key mask [B, K]: frames (t2s) -- or the keys of a permuted call
  const int total = Lq * gc * Kn;
  for (int idx = tid; idx < total; idx += 256) {
    int i, gl, k;
    if (dir == 0) { gl = idx % gc; const int t2 = idx / gc; k = t2 % Kn; i = t2 / Kn; }
    else          { k = idx % Kn; const int t2 = idx / Kn; gl = t2 % gc; i = t2 / gc; }
    const int g = g0 + gl;
    const long col = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
    float v = sc[((long)i * h + hh) * TS_ + col];
    if (mk && mk[k] == 0) v = MASK_FILL;
    P[((long)i * gc + gl) * KP + k] = v;
  }
  for (int idx = tid; idx < gc * Lq * dk; idx += 256) {     // dO chunk: [gl][i][c]
    const int c = idx % dk; const int t2 = idx / dk; const int i = t2 % Lq, gl = t2 / Lq;
    dOs[idx] = to_f(dO[(((long)b * G + g0 + gl) * Lq + i) * d + hh * dk + c]);
  }
  __syncthreads();
  for (int r = tid; r < Lq * gc; r += 256) {                 // softmax rows
    float* p = P + (long)r * KP;
    float mx = -INFINITY;
    for (int k = 0; k < Kn; ++k) mx = fmaxf(mx, p[k]);
    float den = 0.f;
    for (int k = 0; k < Kn; ++k) { const float e = expf(p[k] - mx); p[k] = e; den += e; }
    const float inv = 1.f / den;
    for (int k = 0; k < Kn; ++k) p[k] *= inv;
  }
  // dP[i][gl][k] = sum_c dO[g,i,c] V[(g,k),c]: thread owns (gl,k), keeps the V row in registers by chunks of 16
  const T* Vb = V + (long)b * TS_ * ldv + hh * dk;
  T* dVb = dV + (long)b * TS_ * lddv + hh * dk;
  for (int idx = tid; idx < Lq * gc * Kn; idx += 256) D[((long)(idx / (gc * Kn)) * gc + (idx / Kn) % gc) * KP + idx % Kn] = 0.f;
  __syncthreads();
  for (int item = tid; item < gc * Kn; item += 256) {
    const int gl = item / Kn, k = item % Kn, g = g0 + gl;
    const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
    for (int c0 = 0; c0 < dk; c0 += 16) {
      float vr[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) vr[u] = (c0 + u < dk) ? to_f(Vb[row * ldv + c0 + u]) : 0.f;
      for (int i = 0; i < Lq; ++i) {
        const float* go = dOs + ((long)gl * Lq + i) * dk + c0;
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (c0 + u < dk) acc += go[u] * vr[u];
        D[((long)i * gc + gl) * KP + k] += acc;
      }
    }
  }
  __syncthreads();
  for (int r = tid; r < Lq * gc; r += 256) {                 // dS = P (dP - sum P dP); masked -> 0
    float* p = P + (long)r * KP;
    float* q = D + (long)r * KP;
    float dot = 0.f;
    const int ri = r / gc, rgl = r - ri * gc;
    const unsigned long long dbase = ((((unsigned long long)b * G + (g0 + rgl)) * h + hh) * Lq + ri) * Kn;
    if (drop.p > 0.f)
      for (int k = 0; k < Kn; ++k) q[k] *= drop_mul(dkey, dbase + k, drop.p, dks);          // dP = mask/(1-p) dP'
    for (int k = 0; k < Kn; ++k) dot += p[k] * q[k];
    for (int k = 0; k < Kn; ++k) {
      float gq = p[k] * (q[k] - dot);
      if (mk && mk[k] == 0) gq = 0.f;
      q[k] = gq;
      if (drop.p > 0.f) p[k] *= drop_mul(dkey, dbase + k, drop.p, dks);                     // P' feeds dV
    }
  }
  __syncthreads();
  for (int idx = tid; idx < total; idx += 256) {             // write dscores with the forward's gather order
    int i, gl, k;
    if (dir == 0) { gl = idx % gc; const int t2 = idx / gc; k = t2 % Kn; i = t2 / Kn; }
    else          { k = idx % Kn; const int t2 = idx / Kn; gl = t2 % gc; i = t2 / gc; }
    const int g = g0 + gl;
    const long col = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
    if (dsc_bf16) dsc16[((long)i * h + hh) * TS_ + col] = (bf16_t)D[((long)i * gc + gl) * KP + k];
    else dsc[((long)i * h + hh) * TS_ + col] = D[((long)i * gc + gl) * KP + k];
  }
  for (int item = tid; item < gc * Kn * dk; item += 256) {   // dV[(g,k), c] = sum_i P[i][gl][k] dO[g,i,c]
    const int c = item % dk; const int t2 = item / dk; const int k = t2 % Kn, gl = t2 / Kn;
    const int g = g0 + gl;
    const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
    float acc = 0.f;
    for (int i = 0; i < Lq; ++i) acc += P[((long)i * gc + gl) * KP + k] * dOs[((long)gl * Lq + i) * dk + c];
    dVb[row * lddv + c] = from_f<T>(acc);
  }
}

// ---------------------------------------------------------------------------------------------
// st2 backward: one workgroup per (b, i).  LDS: q2f[h][d], dPY[h][d], P[h][G], dS[h][G].
// ---------------------------------------------------------------------------------------------
constexpr int ST2_MAXH = 16;
template <typename T>
__global__ __launch_bounds__(256) void st2_bwd_kernel(const T* __restrict__ q2f, const T* __restrict__ Y, const unsigned char* __restrict__ gmask,
                                                      const T* __restrict__ dPY, const float* __restrict__ d_rowsum, T* __restrict__ dq2f,
                                                      T* __restrict__ dY, int G, int Lq, int h, int d, const DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qf = smem;
  float* gp = qf + (long)h * d;
  float* P = gp + (long)h * d;
  float* dS = P + (long)h * G;
  const int i = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long qoff = ((long)b * Lq + i) * h * d;
  for (int e = tid; e < h * d; e += 256) { qf[e] = to_f(q2f[qoff + e]); gp[e] = to_f(dPY[qoff + e]); }
  __syncthreads();
  const long ystride = (long)Lq * d;
  const T* Yb = Y + ((long)b * G * Lq + i) * d;
  T* dYb = dY + ((long)b * G * Lq + i) * d;
  const unsigned char* mk = gmask ? gmask + (long)b * G : nullptr;
  for (int g = w; g < G; g += 4) {                           // sc and dP for every head
    const T* yr = Yb + g * ystride;
    float as[ST2_MAXH], ap[ST2_MAXH];
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh) { as[hh] = 0.f; ap[hh] = 0.f; }
    for (int e = lane; e < d; e += 64) {
      const float y = to_f(yr[e]);
#pragma unroll
      for (int hh = 0; hh < ST2_MAXH; ++hh)
        if (hh < h) { as[hh] += qf[hh * d + e] * y; ap[hh] += gp[hh * d + e] * y; }
    }
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh) {
      if (hh < h) {
        float s = wave_sum(as[hh]);
        const float dp = wave_sum(ap[hh]);
        if (mk && mk[g] == 0) s = MASK_FILL;
        if (lane == 0) { P[hh * G + g] = s; dS[hh * G + g] = dp; }
      }
    }
  }
  __syncthreads();
  for (int hh = w; hh < h; hh += 4) {
    float* p = P + hh * G;
    float* q = dS + hh * G;
    float mx = -INFINITY;
    for (int g = lane; g < G; g += 64) mx = fmaxf(mx, p[g]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int g = lane; g < G; g += 64) { const float e = expf(p[g] - mx); p[g] = e; den += e; }
    den = wave_sum(den);
    const float inv = 1.f / den;
    float dot = 0.f;
    const float drs = d_rowsum ? d_rowsum[((long)b * Lq + i) * h + hh] : 0.f;      // gradient of sum_g P'[g]
    const unsigned long long dkey = drop.p > 0.f ? drop.key() : 0ULL;
    const unsigned long long dbase = (((unsigned long long)b * Lq + i) * h + hh) * G;
    const float dks = drop.p > 0.f ? drop.keep_scale() : 1.f;
    for (int g = lane; g < G; g += 64) {
      p[g] *= inv;
      q[g] += drs;
      if (drop.p > 0.f) q[g] *= drop_mul(dkey, dbase + g, drop.p, dks);             // dP = mask/(1-p) dP'
      dot += p[g] * q[g];
    }
    dot = wave_sum(dot);
    for (int g = lane; g < G; g += 64) {
      float gq = p[g] * (q[g] - dot);
      if (mk && mk[g] == 0) gq = 0.f;
      q[g] = gq;
      if (drop.p > 0.f) p[g] *= drop_mul(dkey, dbase + g, drop.p, dks);             // P' feeds dY
    }
  }
  __syncthreads();
  for (int e = tid; e < d; e += 256) {
    float aq[ST2_MAXH];
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh) aq[hh] = 0.f;
    for (int g = 0; g < G; ++g) {
      const float y = to_f(Yb[g * ystride + e]);
      float dy = 0.f;
#pragma unroll
      for (int hh = 0; hh < ST2_MAXH; ++hh)
        if (hh < h) {
          const float s = dS[hh * G + g];
          aq[hh] += s * y;
          dy += s * qf[hh * d + e] + P[hh * G + g] * gp[hh * d + e];
        }
      dYb[g * ystride + e] = from_f<T>(dy);
    }
#pragma unroll
    for (int hh = 0; hh < ST2_MAXH; ++hh)
      if (hh < h) dq2f[qoff + hh * d + e] = from_f<T>(aq[hh]);
  }
}

}  // namespace

extern "C" int bist_mha_core_bwd(const void* Q, const void* K, const void* V, const uint8_t* mask, const void* dO, const float* dP_ext,
                                 void* dQ, void* dK, void* dV, int32_t N, int32_t Lq, int32_t Lk, int32_t h, int32_t dk,
                                 int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                                 int64_t lddq, int64_t lddk, int64_t lddv, int64_t dq_bs, int64_t dk_bs, int64_t dv_bs,
                                 int64_t mask_bs, int64_t mask_qs, float scale, const BistDrop* drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(Q && K && V && dQ && dK && dV && (dO || dP_ext), "bist_mha_core_bwd: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_mha_core_bwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  BIST_REQUIRE(N > 0 && Lq > 0 && Lk > 0 && h > 0 && dk > 0, "bist_mha_core_bwd: bad shape");
  static const bool valu_mha = getenv("BIST_MHA_VALU") != nullptr;      // tuning aid, read once
  if (dtype == BIST_BF16 && !valu_mha) {          // matrix-core path (attention_mfma.hip)
    const int r = bist_mha_bwd_mfma(Q, K, V, mask, dO, dP_ext, dQ, dK, dV, N, Lq, Lk, h, dk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs,
                                    lddq, lddk, lddv, dq_bs, dk_bs, dv_bs, mask_bs, mask_qs, scale, dr, (hipStream_t)stream);
    if (r == 1) return BIST_OK;
    if (r < 0) return BIST_ELAUNCH;      // (the matrix-core launcher left the reason in bist_last_error)
  }
  bist_count_launch(BIST_K_MHA_BWD_VALU);
  const size_t lds = (size_t)2 * Lq * Lk * sizeof(float);
  BIST_REQUIRE(lds <= 64 * 1024, "bist_mha_core_bwd: Lq*Lk=%d too large for LDS", Lq * Lk);
  hipStream_t st = (hipStream_t)stream;
  const int cs = (dk % 64 == 0 && (long)h * N * (dk / 64) <= 4096) ? dk / 64 : 1;      // column chunks of 64: more workgroups for wide heads
  dim3 grid((unsigned)h, (unsigned)N, (unsigned)cs);
#define L(TT) hipLaunchKernelGGL(mha_core_bwd_kernel<TT>, grid, dim3(256), lds, st, (const TT*)Q, (const TT*)K, (const TT*)V, mask, (const TT*)dO, dP_ext, \
                                 (TT*)dQ, (TT*)dK, (TT*)dV, Lq, Lk, h, dk, (long)ldq, (long)ldk, (long)ldv, (long)ldo, (long)q_bs, (long)k_bs, (long)v_bs, (long)o_bs, \
                                 (long)lddq, (long)lddk, (long)lddv, (long)dq_bs, (long)dk_bs, (long)dv_bs, (long)mask_bs, (long)mask_qs, scale, dr)
  if (dtype == BIST_BF16) L(bf16_t); else if (dtype == BIST_F32) L(float);
  else { bist_set_error("bist_mha_core_bwd: bad dtype %d", dtype); return BIST_EINVAL; }
#undef L
  BIST_LAUNCH_CHECK("bist_mha_core_bwd");
  return BIST_OK;
}

extern "C" int bist_st_stage1_pv_bwd(const float* scores, const void* V, const uint8_t* tmask, const void* dO, void* dscores, int32_t dscores_dtype, void* dV,
                                     int32_t B, int32_t T, int32_t S, int32_t Lq, int32_t h, int32_t dk, int64_t ldv, int64_t lddv,
                                     int32_t direction, const BistDrop* drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(scores && V && dO && dscores && dV, "bist_st_stage1_pv_bwd: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_st_stage1_pv_bwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  BIST_REQUIRE(B > 0 && T > 0 && S > 0 && Lq > 0 && h > 0 && dk > 0, "bist_st_stage1_pv_bwd: bad shape");
  BIST_REQUIRE(direction == 0 || direction == 1, "bist_st_stage1_pv_bwd: bad direction");
  BIST_REQUIRE(dscores_dtype == BIST_F32 || dscores_dtype == BIST_BF16, "bist_st_stage1_pv_bwd: bad dscores dtype %d", (int)dscores_dtype);
  const int dsc_bf16 = dscores_dtype == BIST_BF16;
  static const bool valu_st1 = getenv("BIST_ST1_VALU") != nullptr;      // tuning aid, read once
  if (dtype == BIST_BF16 && !valu_st1) {          // matrix-core path (attention_mfma.hip)
    const int r = bist_st1_mfma(scores, 1, V, tmask, nullptr, dO, dscores, dsc_bf16, dV, B, T, S, Lq, h, dk, ldv, lddv, direction, 1, dr,
                                (hipStream_t)stream);
    if (r == 1) return BIST_OK;
    if (r < 0) return BIST_ELAUNCH;      // (the matrix-core launcher left the reason in bist_last_error)
  }
  bist_count_launch(BIST_K_ST1_VALU);
  const int G = direction == 0 ? S : T, Kn = direction == 0 ? T : S;
  const long per_g = ((long)2 * Lq * (Kn + 1) + (long)Lq * dk) * sizeof(float);
  int Gc = (int)((60 * 1024) / per_g);
  BIST_REQUIRE(Gc >= 1, "bist_st_stage1_pv_bwd: Lq*(2K+dk) does not fit LDS");
  if (Gc > G) Gc = G;
  while (Gc > 1 && (long)((G + Gc - 1) / Gc) * h * B < 512) --Gc;
  const size_t lds = (size_t)per_g * Gc;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((G + Gc - 1) / Gc), (unsigned)h, (unsigned)B);
#define L(TT) hipLaunchKernelGGL(st1_pv_bwd_kernel<TT>, grid, dim3(256), lds, st, scores, (const TT*)V, tmask, (const TT*)dO, dscores, dsc_bf16, (TT*)dV, T, S, Lq, h, dk, (long)ldv, (long)lddv, direction, Gc, dr)
  if (dtype == BIST_BF16) L(bf16_t); else if (dtype == BIST_F32) L(float);
  else { bist_set_error("bist_st_stage1_pv_bwd: bad dtype %d", dtype); return BIST_EINVAL; }
#undef L
  BIST_LAUNCH_CHECK("bist_st_stage1_pv_bwd");
  return BIST_OK;
}

extern "C" int bist_st_stage1_pv_bwd_p(const float* P, int32_t KP, const void* V, const uint8_t* tmask, const void* dO, void* dscores,
                                       int32_t dscores_dtype, void* dV, int32_t B, int32_t T, int32_t S, int32_t Lq, int32_t h, int32_t dk,
                                       int64_t ldv, int64_t lddv, int32_t direction, const BistDrop* drop, int32_t dtype, void* stream) {
  BIST_REQUIRE(P && V && dO && dscores && dV, "bist_st_stage1_pv_bwd_p: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_st_stage1_pv_bwd_p: drop p out of range");
  BIST_REQUIRE(B > 0 && T > 0 && S > 0 && Lq > 0 && h > 0 && dk > 0 && (direction == 0 || direction == 1), "bist_st_stage1_pv_bwd_p: bad shape");
  BIST_REQUIRE(dscores_dtype == BIST_F32 || dscores_dtype == BIST_BF16, "bist_st_stage1_pv_bwd_p: bad dscores dtype %d", (int)dscores_dtype);
  BIST_REQUIRE(dtype == BIST_BF16 && KP >= (direction == 0 ? T : S), "bist_st_stage1_pv_bwd_p: bf16 only, KP >= keys");
  const int r = bist_st1_mfma(P, 1, V, tmask, nullptr, dO, dscores, dscores_dtype == BIST_BF16, dV, B, T, S, Lq, h, dk, ldv, lddv, direction, 1,
                              make_drop(drop), (hipStream_t)stream, KP);
  if (r == 1) return BIST_OK;
  if (r < 0) return BIST_ELAUNCH;
  bist_set_error("bist_st_stage1_pv_bwd_p: shape outside the matrix-core kernel's envelope (dk = 64, Lq <= 32, keys <= 128)");
  return BIST_EINVAL;
}

extern "C" int bist_st_stage2_bwd(const void* q2f, const void* Y, const uint8_t* gmask, const void* dPY, const float* d_rowsum,
                                  void* dq2f, void* dY, int32_t B, int32_t G, int32_t Lq, int32_t h, int32_t d, const BistDrop* drop,
                                  int32_t dtype, void* stream) {
  BIST_REQUIRE(q2f && Y && dPY && dq2f && dY, "bist_st_stage2_bwd: null pointer");
  BIST_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "bist_st_stage2_bwd: drop p out of range");
  const DropArg dr = make_drop(drop);
  BIST_REQUIRE(B > 0 && G > 0 && Lq > 0 && h > 0 && h <= ST2_MAXH && d > 0, "bist_st_stage2_bwd: bad shape");
  static const bool valu_st2 = getenv("BIST_ST2_VALU") != nullptr;      // tuning aid, read once
  if (dtype == BIST_BF16 && !valu_st2) {          // matrix-core path (attention_mfma.hip)
    const int r = bist_st2_mfma(q2f, Y, gmask, nullptr, dPY, dq2f, dY, nullptr, d_rowsum, B, G, Lq, h, d, 1, dr, (hipStream_t)stream);
    if (r == 1) return BIST_OK;
    if (r < 0) return BIST_ELAUNCH;      // (the matrix-core launcher left the reason in bist_last_error)
  }
  bist_count_launch(BIST_K_ST2_VALU);
  const size_t lds = ((size_t)2 * h * d + (size_t)2 * h * G) * sizeof(float);
  BIST_REQUIRE(lds <= 64 * 1024, "bist_st_stage2_bwd: h*(d+G) too large for LDS");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)Lq, (unsigned)B);
#define L(TT) hipLaunchKernelGGL(st2_bwd_kernel<TT>, grid, dim3(256), lds, st, (const TT*)q2f, (const TT*)Y, gmask, (const TT*)dPY, d_rowsum, (TT*)dq2f, (TT*)dY, G, Lq, h, d, dr)
  if (dtype == BIST_BF16) L(bf16_t); else if (dtype == BIST_F32) L(float);
  else { bist_set_error("bist_st_stage2_bwd: bad dtype %d", dtype); return BIST_EINVAL; }
#undef L
  BIST_LAUNCH_CHECK("bist_st_stage2_bwd");
  return BIST_OK;
}
