// Output heads of the path: pointer-generator mixture, log-softmax, label-smoothed KL loss.
// One 256-thread workgroup per output row (a target position); V-wide rows stay in L2/HBM.
#include "common.hpp"

namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();                       // red[] may still be read from a previous reduction
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

struct PtrSrc { const float* p; const long* text; int L; };
struct PtrArgs { PtrSrc s[3]; int n; };

// MultiPointerGenerator / PointerGenerator (model/generator.py:36-75, 84-127):
//   out[row, v] = log( sw[n] * softmax(logits[row])[v] + sum_j sw[j] * sum_{t: text_j[b,t]==v} p_j[row, t] )
// with sw = softmax(switch[row, :n+1]) (n > 1) or sw = (1-sigmoid, sigmoid) (single pointer).
template <bool LDSROW>
__global__ __launch_bounds__(256) void pointer_mix_kernel(const float* __restrict__ logits, const float* __restrict__ sw_logits,
                                                          PtrArgs a, float* __restrict__ out, int V, int Lt, int sigmoid_switch) {
  // LDSROW (V <= 4096): the vocabulary row lives in LDS between its one read and its one write -- otherwise the three passes, the
  // scatter and the log are six dependent global round trips per workgroup
  __shared__ float red[4];
  __shared__ float orow[LDSROW ? 4096 : 1];
  const long row = blockIdx.x;
  const int b = (int)(row / Lt);
  const float* lg = logits + row * V;
  float* og = out + row * V;
  float* o = LDSROW ? orow : og;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) { const float l = lg[v]; if (LDSROW) orow[v] = l; mx = fmaxf(mx, l); }
  mx = block_reduce(mx, red, true);
  if (LDSROW) lg = orow;
  float den = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) den += expf(lg[v] - mx);
  den = block_reduce(den, red, false);
  float sw[4];
  const int ns = a.n + 1;
  if (sigmoid_switch) {                   // generator.py:72-73: switch = sigmoid(W x); ptr gets (1 - switch)
    const float s = 1.f / (1.f + expf(-sw_logits[row]));
    sw[0] = 1.f - s; sw[1] = s;
  } else {
    float m2 = -INFINITY, d2 = 0.f;
    for (int j = 0; j < ns; ++j) m2 = fmaxf(m2, sw_logits[row * ns + j]);
    for (int j = 0; j < ns; ++j) { sw[j] = expf(sw_logits[row * ns + j] - m2); d2 += sw[j]; }
    for (int j = 0; j < ns; ++j) sw[j] /= d2;
  }
  const float vs = sw[a.n] / den;
  for (int v = threadIdx.x; v < V; v += 256) o[v] = vs * expf(lg[v] - mx);
  __syncthreads();
  // scatter_add of the copy distributions, deterministic and parallel: the FIRST position of every distinct id adds the sum of all
  // positions with that id (taken in ascending position order), so no two threads touch the same output and no atomics are needed
  // (the serial loop of one thread this replaces took 15 of the kernel's 20 us at 5 rows)
  for (int j = 0; j < a.n; ++j) {
    const float* p = a.s[j].p + row * a.s[j].L;
    const long* text = a.s[j].text + (long)b * a.s[j].L;
    const int L = a.s[j].L;
    for (int t = threadIdx.x; t < L; t += 256) {
      const long id = text[t];
      bool first = true;
      for (int u = 0; u < t; ++u) first = first && text[u] != id;
      if (first) {
        float acc = 0.f;
        for (int u = t; u < L; ++u) if (text[u] == id) acc += sw[j] * p[u];
        o[id] += acc;
      }
    }
    __syncthreads();                       // the next source may hold the same ids
  }
  for (int v = threadIdx.x; v < V; v += 256) og[v] = logf(o[v]);
}

// The pointer generator's text vector, inference (generator.py:117-118): out[row, c] = sum_t p[row, t] * enc[b, t, c], p f32 [rows, L],
// enc [B, L, d]: one block per row, one 16-byte slice of the d channels per thread (d <= 2048 for bf16).  (Training keeps the batched
// GEMM: it needs the two backward products.)
template <typename T>
__global__ __launch_bounds__(256) void text_vector_kernel(const float* __restrict__ p, const T* __restrict__ enc, T* __restrict__ out, int Lt, int L, int d) {
  constexpr int E = 16 / (int)sizeof(T);
  const long row = blockIdx.x;
  const int b = (int)(row / Lt), c0 = threadIdx.x * E;
  if (c0 >= d) return;
  const float* pr = p + row * L;
  const T* e = enc + (long)b * L * d + c0;
  float acc[E];
#pragma unroll
  for (int i = 0; i < E; ++i) acc[i] = 0.f;
  for (int t = 0; t < L; ++t) {
    const uint4 q = *reinterpret_cast<const uint4*>(e + (long)t * d);
    const T* qe = reinterpret_cast<const T*>(&q);
    const float w = pr[t];
#pragma unroll
    for (int i = 0; i < E; ++i) acc[i] += w * to_f(qe[i]);
  }
  T o[E];
#pragma unroll
  for (int i = 0; i < E; ++i) o[i] = from_f<T>(acc[i]);
  *reinterpret_cast<uint4*>(out + row * d + c0) = *reinterpret_cast<const uint4*>(o);
}

// The switch logits of (Multi)PointerGenerator (generator.py:71, 119-121): pointer_gen_W applied to the concatenation of n_parts <= 4
// [rows, d] tensors WITHOUT the concatenation -- part j multiplies column block j of W [ns <= 4][n_parts d].  One launch forward (a wave
// per row) and two backward (the parts' gradients: a wave per row; the weight / bias gradient: a workgroup per part, four row slices
// per workgroup) where a product per part took 4 launches forward and 12 backward (cast + dX + dW each).
struct SwParts { const void* x[4]; void* dx[4]; const void* res[4]; int n; };

template <typename T, typename TO>
__global__ __launch_bounds__(256) void switch_logits_fwd_kernel(SwParts a, const T* __restrict__ W, long ldw, const T* __restrict__ bias,
                                                                TO* __restrict__ out, long rows, int d, int ns) {
  constexpr int E = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < a.n; ++j) {
    const T* xr = reinterpret_cast<const T*>(a.x[j]) + row * d;
    for (int c = lane * E; c < d; c += 64 * E) {
      T xv[E];
      *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(xr + c);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        if (s_ < ns) {
          T wv[E];
          *reinterpret_cast<uint4*>(wv) = *reinterpret_cast<const uint4*>(W + (long)s_ * ldw + (long)j * d + c);
#pragma unroll
          for (int e = 0; e < E; ++e) acc[s_] += to_f(xv[e]) * to_f(wv[e]);
        }
      }
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    if (s_ < ns) {
      const float v = wave_sum(acc[s_]);
      if (lane == 0) out[row * ns + s_] = from_f<TO>(v + (bias ? to_f(bias[s_]) : 0.f));
    }
  }
}

template <typename T, typename TS>
__global__ __launch_bounds__(256) void switch_logits_bwd_x_kernel(SwParts a, const T* __restrict__ W, long ldw, const TS* __restrict__ dsw,
                                                                  long rows, int d, int ns) {
  constexpr int E = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float g[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) g[s_] = s_ < ns ? to_f(dsw[row * ns + s_]) : 0.f;
  for (int j = 0; j < a.n; ++j) {
    T* dr = reinterpret_cast<T*>(a.dx[j]) + row * d;
    if (a.dx[j] == nullptr) continue;
    const T* rr = a.res[j] ? reinterpret_cast<const T*>(a.res[j]) + row * d : nullptr;      // an addend of the part's gradient from elsewhere
    for (int c = lane * E; c < d; c += 64 * E) {
      float acc[E];
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = 0.f;
      if (rr) {
        T rv[E];
        *reinterpret_cast<uint4*>(rv) = *reinterpret_cast<const uint4*>(rr + c);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = to_f(rv[e]);
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        if (s_ < ns) {
          T wv[E];
          *reinterpret_cast<uint4*>(wv) = *reinterpret_cast<const uint4*>(W + (long)s_ * ldw + (long)j * d + c);
#pragma unroll
          for (int e = 0; e < E; ++e) acc[e] += g[s_] * to_f(wv[e]);
        }
      }
      T o[E];
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = from_f<T>(acc[e]);
      *reinterpret_cast<uint4*>(dr + c) = *reinterpret_cast<const uint4*>(o);
    }
  }
}

// dW[s][j d + c] (+)= sum_rows dsw[row][s] part_j[row][c];  db[s] (+)= sum_rows dsw[row][s].  blockIdx.x = part, blockIdx.y = chunk of 64 E
// columns; thread = (column piece lane, row slice sl): the four slices meet in LDS.
template <typename T, typename TW, typename TS>
__global__ __launch_bounds__(256) void switch_logits_bwd_w_kernel(SwParts a, const TS* __restrict__ dsw, TW* __restrict__ dW, long lddw, int dw_acc,
                                                                  float* __restrict__ db, int db_acc, long rows, int d, int ns) {
  constexpr int E = 16 / (int)sizeof(T);
  __shared__ float part[3][64][4][E];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6, j = blockIdx.x;
  const int c = (blockIdx.y * 64 + lane) * E;
  const bool act = c < d;
  const T* xb = reinterpret_cast<const T*>(a.x[j]);
  float acc[4][E];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[s_][e] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const long per = (rows + 3) / 4, r0 = sl * per, r1 = min(rows, r0 + per);
  // eight rows per step: their loads are in flight together (one row per step made this a chain of ~80 dependent round trips)
  for (long rb = r0; rb < r1; rb += 8) {
    uint4 xq[8];
    float g[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long r = rb + u;
      const bool ok = r < r1;
      xq[u] = (ok && act) ? *reinterpret_cast<const uint4*>(xb + r * d + c) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) g[u][s_] = (ok && s_ < ns) ? to_f(dsw[r * ns + s_]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const T* xv = reinterpret_cast<const T*>(&xq[u]);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[s_][e] += g[u][s_] * to_f(xv[e]);
        bsum[s_] += g[u][s_];
      }
    }
  }
  if (sl > 0) {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int e = 0; e < E; ++e) part[sl - 1][lane][s_][e] = acc[s_][e];
  }
  __shared__ float bpart[4][4];
  if (lane == 0)
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) bpart[sl][s_] = bsum[s_];
  __syncthreads();
  if (sl == 0 && act) {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      if (s_ < ns) {
        TW* dst = dW + (long)s_ * lddw + (long)j * d + c;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float v = acc[s_][e] + part[0][lane][s_][e] + part[1][lane][s_][e] + part[2][lane][s_][e];
          dst[e] = from_f<TW>(dw_acc ? to_f(dst[e]) + v : v);
        }
      }
    }
  }
  if (db && j == 0 && blockIdx.y == 0 && threadIdx.x < ns) {
    const int s_ = threadIdx.x;
    const float v = bpart[0][s_] + bpart[1][s_] + bpart[2][s_] + bpart[3][s_];
    db[s_] = db_acc ? db[s_] + v : v;
  }
}

// The pointer attention of (Multi)PointerGenerator in TRAINING (generator.py:106-118): single head over d channels.
//   forward : p[b,i,:] = softmax_t(live(b,t) ? scale q[b,i].k[b,t] : -1e9),  live = mask[b,t] && (!mask_unk || text[b,t] != unk);
//             tv[b,i,:] = sum_t p[b,i,t] enc[b,t,:]  (the text vector), one wave per (b, i): two launches' worth of the generic attention
//             core (which also multiplied P by a value operand nobody reads), two mask operations and the cast + fallback product.
//   backward: dpt = dp + dtv . enc^T;  dS = p (dpt - sum_t p dpt) scale;  dq = dS k;  dk = dS^T q;  denc = p^T dtv -- one workgroup per b.
// sc[i][t] (+)= a[i] . b[t] over d channels for i < M <= 32, t < N <= 128: bf16 on the matrix cores (16x16x32 tiles, wave-strided over the
// <= 16 output tiles, fragments straight from global memory: lane (x, kg) reads 16 bytes of row x at k0 + 8 kg), f32 with a wave per pair.
template <typename T>
__device__ __forceinline__ void small_nt_product(const T* __restrict__ a, const T* __restrict__ b, int M, int N, int d, float (*sc)[129], bool add,
                                                 int w, int lane) {
  if constexpr (sizeof(T) == 2) {
    const int x = lane & 15, kg = lane >> 4;
    const int NT = (N + 15) / 16, tiles = ((M + 15) / 16) * NT;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    for (int tile = w; tile < tiles; tile += 4) {
      const int mi = tile / NT, ni = tile - mi * NT;
      const int ia = mi * 16 + x, tb = ni * 16 + x;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kb0 = 0; kb0 < d; kb0 += 256) {             // eight k-steps per round: their sixteen loads are in flight together
        uint4 af[8], bf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k0 = kb0 + 32 * u;
          af[u] = (ia < M && k0 < d) ? *reinterpret_cast<const uint4*>(a + (long)ia * d + k0 + kg * 8) : zero;
          bf[u] = (tb < N && k0 < d) ? *reinterpret_cast<const uint4*>(b + (long)tb * d + k0 + kg * 8) : zero;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[u]), __builtin_bit_cast(bf16x8, bf[u]), acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {                      // D[m = 4 kg + r][n = x]
        const int i = mi * 16 + 4 * kg + r, t = ni * 16 + x;
        if (i < M && t < N) sc[i][t] = add ? sc[i][t] + acc[r] : acc[r];
      }
    }
  } else {
    constexpr int E = 4;
    for (int e = w; e < M * N; e += 4) {
      const int i = e / N, t = e - i * N;
      float acc = 0.f;
      for (int c = lane * E; c < d; c += 64 * E) {
        const float4 u = *reinterpret_cast<const float4*>(a + (long)i * d + c), v = *reinterpret_cast<const float4*>(b + (long)t * d + c);
        acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
      }
      acc = wave_sum(acc);
      if (lane == 0) sc[i][t] = add ? sc[i][t] + acc : acc;
    }
  }
}

// out[r][:] = sum_j coef(r, j) mat[j][:] for r < R, j < J, d channels (a multiple of 64): a thread per (row, 64-column group); the
// coefficients sit in LDS (coef(r, j) = cf[r][j], or cf[j][r] when TR), the matrix rows come from global memory 16 bytes at a time (the
// eight lanes of a row read one contiguous run; other rows re-read it out of L1).  Plain loops: the unrolled per-row accumulators of the
// first version were 35 KiB of once-run code (42 us per launch, most of it instruction fetch).
template <typename T, bool TR>
__device__ __forceinline__ void lds_coef_times_rows(const float (*cf)[129], int R, int J, const T* __restrict__ mat, int d, T* __restrict__ out, int tid) {
  constexpr int E = 16 / (int)sizeof(T), NV = 64 / E;
  const int ngroups = d / 64;
  for (int task = tid; task < R * ngroups; task += 256) {
    const int r = task / ngroups, gi = task - r * ngroups;      // piece u of this thread: columns (u ngroups + gi) E .. (the ngroups lanes of a row read one contiguous run)
    float acc[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 < J; j0 += 4) {                    // four matrix rows per round: their loads are in flight together
      uint4 raw[4][NV];
      float cj[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = min(j0 + jj, J - 1);
        cj[jj] = (j0 + jj < J) ? (TR ? cf[j][r] : cf[r][j]) : 0.f;
        const T* mr = mat + (long)j * d + gi * E;
#pragma unroll
        for (int u = 0; u < NV; ++u) raw[jj][u] = *reinterpret_cast<const uint4*>(mr + u * ngroups * E);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int u = 0; u < NV; ++u) {
          const T* v = reinterpret_cast<const T*>(&raw[jj][u]);
#pragma unroll
          for (int e = 0; e < E; ++e) acc[u * E + e] += cj[jj] * to_f(v[e]);
        }
    }
    T* orow = out + (long)r * d + gi * E;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      T v[E];
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = from_f<T>(acc[u * E + e]);
      *reinterpret_cast<uint4*>(orow + u * ngroups * E) = *reinterpret_cast<const uint4*>(v);
    }
  }
}

// forward: one workgroup per sequence b (Lt <= 32 target positions, L <= 128 source positions)
template <typename T>
__global__ __launch_bounds__(256) void pointer_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ enc,
                                                               const unsigned char* __restrict__ mask, long mask_bs, const long* __restrict__ text,
                                                               long unk, float* __restrict__ p, T* __restrict__ tv, int Lt, int L, int d, float scale) {
  __shared__ float sc[32][129];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
  small_nt_product<T>(q + (long)b * Lt * d, k + (long)b * L * d, Lt, L, d, sc, false, w, lane);
  __syncthreads();
  for (int i = w; i < Lt; i += 4) {                     // masked softmax, a wave per target position
    float s0 = -INFINITY, s1 = -INFINITY;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int t = lane + 64 * h2;
      if (t < L) {
        bool live = mask == nullptr || mask[(long)b * mask_bs + t] != 0;
        if (text != nullptr && text[(long)b * L + t] == unk) live = false;
        const float v = live ? sc[i][t] * scale : -1e9f;
        if (h2 == 0) s0 = v; else s1 = v;
      }
    }
    const float mx = wave_max(fmaxf(s0, s1));
    const float e0 = lane < L ? expf(s0 - mx) : 0.f, e1 = lane + 64 < L ? expf(s1 - mx) : 0.f;
    const float inv = 1.f / wave_sum(e0 + e1);
    if (lane < L) { sc[i][lane] = e0 * inv; p[((long)b * Lt + i) * L + lane] = e0 * inv; }
    if (lane + 64 < L) { sc[i][lane + 64] = e1 * inv; p[((long)b * Lt + i) * L + lane + 64] = e1 * inv; }
  }
  if (tv == nullptr) return;
  __syncthreads();
  lds_coef_times_rows<T, false>(sc, Lt, L, enc + (long)b * L * d, d, tv + (long)b * Lt * d, tid);      // text vector = p . enc
}

template <typename T>
__global__ __launch_bounds__(256) void pointer_attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ enc,
                                                               const float* __restrict__ p, const float* __restrict__ dp, const T* __restrict__ dtv,
                                                               T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ denc, int Lt, int L, int d,
                                                               float scale) {
  __shared__ float ps[32][129], ds[32][129];            // Lt <= 32, L <= 128
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
  const T* qb = q + (long)b * Lt * d;
  const T* kb = k + (long)b * L * d;
  const T* eb = enc ? enc + (long)b * L * d : nullptr;
  const T* db = dtv ? dtv + (long)b * Lt * d : nullptr;
  const float* pb = p + (long)b * Lt * L;
  for (int e = tid; e < Lt * L; e += 256) {
    const int i = e / L, t = e - i * L;
    ps[i][t] = pb[e];
    ds[i][t] = dp ? dp[(long)b * Lt * L + e] : 0.f;
  }
  __syncthreads();
  if (db) {                                             // dpt += dtv . enc^T
    small_nt_product<T>(db, eb, Lt, L, d, ds, true, w, lane);
    __syncthreads();
  }
  for (int i = w; i < Lt; i += 4) {                     // softmax backward, a wave per query row
    const float a0 = lane < L ? ps[i][lane] * ds[i][lane] : 0.f, a1 = lane + 64 < L ? ps[i][lane + 64] * ds[i][lane + 64] : 0.f;
    const float part = wave_sum(a0 + a1);
    if (lane < L) ds[i][lane] = ps[i][lane] * (ds[i][lane] - part) * scale;
    if (lane + 64 < L) ds[i][lane + 64] = ps[i][lane + 64] * (ds[i][lane + 64] - part) * scale;
  }
  __syncthreads();
  // the three products on three workgroups per sequence (blockIdx.y; each repeats the small steps above: they are latency, not work)
  if (blockIdx.y == 0) lds_coef_times_rows<T, false>(ds, Lt, L, kb, d, dq + (long)b * Lt * d, tid);                  // dq = dS k
  else if (blockIdx.y == 1) lds_coef_times_rows<T, true>(ds, L, Lt, qb, d, dk + (long)b * L * d, tid);               // dk = dS^T q
  else if (db) lds_coef_times_rows<T, true>(ps, L, Lt, db, d, denc + (long)b * L * d, tid);                          // d enc = p^T dtv
}

// Decode-step form of MultiPointerGenerator.forward (generator.py:84-127) for hypothesis rows that share ONE dialogue (beam search,
// decode.py:59-66).  The pointer attentions' keys do not change during a turn, so the caller folds each source's query projection into
// them once per turn:  scores_j[t] = (W_q x + b_q) . k_j[t] = x . M_j[t] + c_j[t]  with  M_j = K_j W_q  [L, d],  c_j = K_j b_q  [L];  and
// the switch logits' text-vector blocks into  E_j = enc_j W_sw,j^T  [L, n + 1]  (W_sw [x | tgt | tv_0 | ..], generator.py:92,119-121):
//   sw_logits = W_sw,x x + W_sw,tgt tgt + b + sum_j p_j E_j.
// One workgroup per row does the n masked softmaxes (generator.py:106-110, single head, fill -1e9), the switch and the mixture of
// pointer_mix_kernel -- a step's pointer heads are ONE launch after the vocabulary product instead of eighteen (2 x {2 mask ops, 2
// projections, attention core, text vector}, 4 switch products, mixture).  d <= 1024, L_j <= 512.
struct PtrDecSrcK { const float* M; const float* c; const unsigned char* mask; const float* E; const long* text; float* p_out; int L; };
struct PtrDecArgs { PtrDecSrcK s[3]; int n; };

template <typename T, bool LDSROW>
__global__ __launch_bounds__(256) void pointer_decode_mix_kernel(const T* __restrict__ x, const T* __restrict__ tgt, const float* __restrict__ logits,
                                                                 PtrDecArgs a, const T* __restrict__ Wsw, long ldw, const T* __restrict__ bsw,
                                                                 float scale, float* __restrict__ out, int d, int V) {
  __shared__ float xs[1024], ts[1024], pr[3][512], swl[4], red[4];
  __shared__ float orow[LDSROW ? 4096 : 1];
  __shared__ int tx[3][512];                  // the sources' token ids (the scatter walks them L^2 / 2 times: out of LDS, not global memory)
  const long row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int k = tid; k < d; k += 256) { xs[k] = to_f(x[row * d + k]); ts[k] = to_f(tgt[row * d + k]); }
  for (int j = 0; j < a.n; ++j)
    for (int t = tid; t < a.s[j].L; t += 256) tx[j][t] = (int)a.s[j].text[t];
  __syncthreads();
  // scores: a quarter wave (16 lanes) per key, sixteen keys in flight per workgroup; the 16 lanes split the d channels (float4 loads,
  // 256 contiguous bytes per quarter wave and step), so a key costs independent loads and four shuffle steps instead of a serial
  // load -> sum chain (a wave per key measured 38 us per launch at 80 keys: ~20 dependent round trips per wave)
  {
    const int sub = lane >> 4, l16 = lane & 15;
    for (int j = 0; j < a.n; ++j) {
      const PtrDecSrcK& sj = a.s[j];
      for (int t0 = 0; t0 < sj.L; t0 += 16) {
        const int t = t0 + 4 * w + sub;
        float acc = 0.f;
        if (t < sj.L) {
          const float* m = sj.M + (long)t * d;
#pragma unroll 8
          for (int k = l16 * 4; k < d; k += 64) {
            const float4 q = *reinterpret_cast<const float4*>(m + k);
            acc += q.x * xs[k] + q.y * xs[k + 1] + q.z * xs[k + 2] + q.w * xs[k + 3];
          }
        }
        acc += __shfl_xor(acc, 8); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 1);
        if (l16 == 0 && t < sj.L) pr[j][t] = sj.mask[t] ? (acc + sj.c[t]) * scale : -1e9f;
      }
    }
  }
  __syncthreads();
  if (w < a.n) {                              // softmax of source w on wave w
    const int L = a.s[w].L;
    float mx = -INFINITY;
    for (int t = lane; t < L; t += 64) mx = fmaxf(mx, pr[w][t]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int t = lane; t < L; t += 64) den += expf(pr[w][t] - mx);
    den = wave_sum(den);
    for (int t = lane; t < L; t += 64) {
      const float pv = expf(pr[w][t] - mx) / den;
      pr[w][t] = pv;
      if (a.s[w].p_out) a.s[w].p_out[row * L + t] = pv;
    }
  }
  __syncthreads();
  const int ns = a.n + 1;
  if (w < ns) {                               // switch logit w on wave w
    const T* wr = Wsw + (long)w * ldw;
    float acc = 0.f;
    for (int k = lane; k < d; k += 64) acc += to_f(wr[k]) * xs[k] + to_f(wr[d + k]) * ts[k];
    for (int j = 0; j < a.n; ++j)
      for (int t = lane; t < a.s[j].L; t += 64) acc += pr[j][t] * a.s[j].E[t * ns + w];
    acc = wave_sum(acc);
    if (lane == 0) swl[w] = acc + to_f(bsw[w]);
  }
  // the mixture (pointer_mix_kernel with the rows' shared texts).  LDSROW: the vocabulary row lives in LDS between its one read and its one
  // write (V <= 4096) -- three passes, the scatter and the log otherwise cost six dependent global round trips of a 5-workgroup launch
  const float* lg = logits + row * V;
  float* og = out + row * V;
  float* o = LDSROW ? orow : og;
  float mx = -INFINITY;
  for (int v = tid; v < V; v += 256) { const float l = lg[v]; if (LDSROW) orow[v] = l; mx = fmaxf(mx, l); }
  mx = block_reduce(mx, red, true);           // (its barriers also publish swl and orow)
  const float* lsrc = LDSROW ? orow : lg;
  float den = 0.f;
  for (int v = tid; v < V; v += 256) den += expf(lsrc[v] - mx);
  den = block_reduce(den, red, false);
  float sw[4];
  {
    float m2 = -INFINITY, d2 = 0.f;
    for (int j = 0; j < ns; ++j) m2 = fmaxf(m2, swl[j]);
    for (int j = 0; j < ns; ++j) { sw[j] = expf(swl[j] - m2); d2 += sw[j]; }
    for (int j = 0; j < ns; ++j) sw[j] /= d2;
  }
  const float vs = sw[a.n] / den;
  for (int v = tid; v < V; v += 256) o[v] = vs * expf(lsrc[v] - mx);
  __syncthreads();
  for (int j = 0; j < a.n; ++j) {             // deterministic scatter_add, as in pointer_mix_kernel
    const int* text = tx[j];
    const int L = a.s[j].L;
    for (int t = tid; t < L; t += 256) {
      const int id = text[t];
      bool first = true;
      for (int u = 0; u < t; ++u) first = first && text[u] != id;
      if (first) {
        float acc = 0.f;
        for (int u = t; u < L; ++u) if (text[u] == id) acc += sw[j] * pr[j][u];
        o[id] += acc;
      }
    }
    __syncthreads();
  }
  for (int v = tid; v < V; v += 256) og[v] = logf(o[v]);
}

// Generator.forward (generator.py:21-27): log_softmax over the vocabulary.
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int V) {
  __shared__ float red[4];
  const float* xr = x + (long)blockIdx.x * V;
  float* yr = y + (long)blockIdx.x * V;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, xr[v]);
  mx = block_reduce(mx, red, true);
  float den = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) den += expf(xr[v] - mx);
  den = block_reduce(den, red, false);
  const float lse = mx + logf(den);
  for (int v = threadIdx.x; v < V; v += 256) yr[v] = xr[v] - lse;
}

// LabelSmoothing.forward (label_smoothing.py:20-30): KLDiv(sum) of the row against the smoothed
// one-hot: confidence on the target, smoothing/(V-2) elsewhere, 0 on the pad column, whole row 0
// when the target is pad.  row_loss[row] = sum_v td[v] * (log td[v] - logp[v]).
__global__ __launch_bounds__(256) void label_smoothing_kernel(const float* __restrict__ logp, const long* __restrict__ target,
                                                              float* __restrict__ row_loss, int V, float smoothing, int pad) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const long t = target[row];
  if (t == pad) { if (threadIdx.x == 0) row_loss[row] = 0.f; return; }
  const float* lp = logp + row * V;
  const float s = smoothing / (float)(V - 2), conf = 1.f - smoothing;
  const float ls = s > 0.f ? logf(s) : 0.f;
  float acc = 0.f;
  if (s > 0.f)
    for (int v = threadIdx.x; v < V; v += 256)
      if (v != t && v != pad) acc += s * (ls - lp[v]);
  acc = block_reduce(acc, red, false);
  if (threadIdx.x == 0) row_loss[row] = acc + (conf > 0.f ? conf * (logf(conf) - lp[t]) : 0.f);
}

// Generator.forward + LabelSmoothing on the same rows (generator.py:21-27, label_smoothing.py: KL(smoothed one-hot || softmax(logits))) as ONE
// pass over the logits, for G groups of M rows that share the targets (the auto-encoder heads of optimize.py:66-82: the caption / temporal /
// spatial outputs against the same query tokens): row r has target[r % M].
//   row_loss = conf (log conf - logp_t) + sum_{v != t, pad} s (log s - logp_v),  logp = logits - lse,  s = smoothing / (V - 2);  0 at pad targets
//   d row_loss / d logits[v] = softmax[v] - td[v]   (the smoothed target sums to 1)
// so neither the log-probabilities nor their f32 gradient are ever stored; the backward writes the logits' gradient in the dtype of the
// two products that consume it.
__global__ __launch_bounds__(256) void xent_smooth_fwd_kernel(const float* __restrict__ logits, const long* __restrict__ target, long M,
                                                              float* __restrict__ row_loss, float* __restrict__ lse, int V, float smoothing, int pad) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const long t = target[row % M];
  const float* lg = logits + row * V;
  float mx = -INFINITY, sum = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) { const float l = lg[v]; mx = fmaxf(mx, l); if (v != t && v != pad) sum += l; }
  mx = block_reduce(mx, red, true);
  sum = block_reduce(sum, red, false);
  float den = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) den += expf(lg[v] - mx);
  den = block_reduce(den, red, false);
  if (threadIdx.x == 0) {
    const float l = mx + logf(den);
    lse[row] = l;
    const float s = smoothing / (float)(V - 2), conf = 1.f - smoothing;
    float loss = 0.f;
    if (t != pad) {
      if (conf > 0.f) loss += conf * (logf(conf) - (lg[t] - l));
      if (s > 0.f) loss += s * ((float)(V - 2) * (logf(s) + l) - sum);
    }
    row_loss[row] = loss;
  }
}

template <typename TD>
__global__ __launch_bounds__(256) void xent_smooth_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse, const long* __restrict__ target,
                                                              long M, const float* __restrict__ gout, int gout_stride, const long* __restrict__ denom,
                                                              TD* __restrict__ dlogits, int V, float smoothing, int pad) {
  const long row = blockIdx.x;
  const long t = target[row % M];
  const float sc = gout[(row / M) * gout_stride] / (denom ? (float)denom[0] : 1.f);
  const float s = smoothing / (float)(V - 2), conf = 1.f - smoothing, l = lse[row];
  const float* lg = logits + row * V;
  TD* g = dlogits + row * V;
  for (int v = threadIdx.x; v < V; v += 256) {
    float d_ = 0.f;
    if (t != pad) d_ = sc * (expf(lg[v] - l) - ((v == t) ? conf : (v == pad ? 0.f : s)));
    g[v] = from_f<TD>(d_);
  }
}

// out[g] = sum(x[g M .. (g + 1) M)) / denom[0]: one workgroup per group, fixed order
__global__ __launch_bounds__(256) void sum_div_groups_kernel(const float* __restrict__ x, long M, const long* __restrict__ denom, float* __restrict__ out,
                                                             int out_stride) {
  __shared__ float red[4];
  const float* xg = x + (long)blockIdx.x * M;
  float acc = 0.f;
  for (long i = threadIdx.x; i < M; i += 256) acc += xg[i];
  acc = block_reduce(acc, red, false);
  if (threadIdx.x == 0) out[(long)blockIdx.x * out_stride] = acc / (denom ? (float)denom[0] : 1.f);
}

struct StackSrc { const uint4* s[4]; };
__global__ __launch_bounds__(256) void stack_rows_kernel(StackSrc a, uint4* __restrict__ out, long n16) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[(long)blockIdx.y * n16 + i] = a.s[blockIdx.y][i];
}

// out[0] (+)= sum(x[0..n)) / denom[0]   -- single workgroup, fixed order: bitwise reproducible
__global__ __launch_bounds__(256) void sum_div_kernel(const float* __restrict__ x, long n, const long* __restrict__ denom,
                                                      float* __restrict__ out, int accumulate) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) acc += x[i];
  acc = block_reduce(acc, red, false);
  if (threadIdx.x == 0) {
    const float v = acc / (denom ? (float)denom[0] : 1.f);
    out[0] = accumulate ? out[0] + v : v;
  }
}

}  // namespace

extern "C" int bist_pointer_mix_fwd(const float* logits, const float* switch_logits, int32_t n_ptr, const float* const* ptr_p,
                                    const int64_t* const* ptr_text, const int32_t* ptr_len, float* out, int64_t rows, int32_t Lt,
                                    int32_t V, int32_t sigmoid_switch, void* stream) {
  BIST_REQUIRE(logits && switch_logits && out && rows > 0 && Lt > 0 && V > 2, "bist_pointer_mix_fwd: bad argument");
  BIST_REQUIRE(n_ptr >= 1 && n_ptr <= 3, "bist_pointer_mix_fwd: 1..3 pointer sources");
  BIST_REQUIRE(!sigmoid_switch || n_ptr == 1, "bist_pointer_mix_fwd: sigmoid switch needs exactly one pointer source");
  PtrArgs a;
  a.n = n_ptr;
  for (int j = 0; j < 3; ++j) a.s[j] = PtrSrc{nullptr, nullptr, 0};
  for (int j = 0; j < n_ptr; ++j) {
    BIST_REQUIRE(ptr_p[j] && ptr_text[j] && ptr_len[j] > 0, "bist_pointer_mix_fwd: bad pointer source %d", j);
    a.s[j] = PtrSrc{ptr_p[j], (const long*)ptr_text[j], ptr_len[j]};
  }
  if (V <= 4096)
    hipLaunchKernelGGL(pointer_mix_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, switch_logits, a, out, V, Lt,
                       sigmoid_switch);
  else
    hipLaunchKernelGGL(pointer_mix_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, switch_logits, a, out, V, Lt,
                       sigmoid_switch);
  BIST_LAUNCH_CHECK("bist_pointer_mix_fwd");
  return BIST_OK;
}

static int sw_parts(SwParts& a, const void* const* parts, void* const* dparts, const void* const* residuals, int32_t n_parts, const char* who) {
  BIST_REQUIRE(parts && n_parts >= 1 && n_parts <= 4, "%s: 1..4 parts", who);
  a.n = n_parts;
  for (int j = 0; j < 4; ++j) { a.x[j] = nullptr; a.dx[j] = nullptr; a.res[j] = nullptr; }
  for (int j = 0; j < n_parts; ++j) {
    BIST_REQUIRE(parts[j] && ((uintptr_t)parts[j] & 15) == 0 && (!dparts || ((uintptr_t)dparts[j] & 15) == 0) && (!residuals || ((uintptr_t)residuals[j] & 15) == 0),
                 "%s: part %d null or not 16-byte aligned", who, j);
    a.x[j] = parts[j];
    a.dx[j] = dparts ? dparts[j] : nullptr;
    a.res[j] = residuals ? residuals[j] : nullptr;
  }
  return BIST_OK;
}

extern "C" int bist_switch_logits_fwd(const void* const* parts, int32_t n_parts, const void* W, int64_t ldw, const void* bias, void* out,
                                      int32_t out_dtype, int64_t rows, int32_t d, int32_t ns, int32_t dtype, void* stream) {
  BIST_REQUIRE(W && out && rows > 0 && ns >= 1 && ns <= 4 && d > 0, "bist_switch_logits_fwd: bad argument (<= 4 switch logits)");
  const int e = dtype == BIST_BF16 ? 8 : 4;
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && d % e == 0 && ldw % e == 0 && ((uintptr_t)W & 15) == 0, "bist_switch_logits_fwd: bf16 / f32, 16-byte rows");
  BIST_REQUIRE(out_dtype == BIST_F32 || out_dtype == dtype, "bist_switch_logits_fwd: the logits are f32 or of the operand dtype");
  SwParts a;
  if (sw_parts(a, parts, nullptr, nullptr, n_parts, "bist_switch_logits_fwd") != BIST_OK) return BIST_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((rows + 3) / 4);
#define SWF(TT, TOO) hipLaunchKernelGGL((switch_logits_fwd_kernel<TT, TOO>), dim3(grid), dim3(256), 0, st, a, (const TT*)W, (long)ldw, (const TT*)bias, (TOO*)out, (long)rows, d, ns)
  if (dtype == BIST_BF16) { if (out_dtype == BIST_F32) SWF(bf16_t, float); else SWF(bf16_t, bf16_t); }
  else SWF(float, float);
#undef SWF
  BIST_LAUNCH_CHECK("bist_switch_logits_fwd");
  return BIST_OK;
}

extern "C" int bist_switch_logits_bwd(const void* const* parts, int32_t n_parts, const void* W, int64_t ldw, const void* dsw, int32_t dsw_dtype,
                                      void* const* dparts, const void* const* residuals, void* dW, int64_t lddw, int32_t dw_dtype,
                                      int32_t dw_accumulate, float* db, int32_t db_accumulate, int64_t rows, int32_t d, int32_t ns, int32_t dtype,
                                      void* stream) {
  BIST_REQUIRE(W && dsw && rows > 0 && ns >= 1 && ns <= 4 && d > 0, "bist_switch_logits_bwd: bad argument (<= 4 switch logits)");
  const int e = dtype == BIST_BF16 ? 8 : 4;
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && d % e == 0 && ldw % e == 0 && ((uintptr_t)W & 15) == 0, "bist_switch_logits_bwd: bf16 / f32, 16-byte rows");
  BIST_REQUIRE(dsw_dtype == BIST_F32 || dsw_dtype == dtype, "bist_switch_logits_bwd: the logits' gradient is f32 or of the operand dtype");
  BIST_REQUIRE(!dW || dw_dtype == BIST_BF16 || dw_dtype == BIST_F32, "bist_switch_logits_bwd: bad weight-gradient dtype");
  BIST_REQUIRE(!residuals || dparts, "bist_switch_logits_bwd: residuals go with dparts");
  SwParts a;
  if (sw_parts(a, parts, dparts, residuals, n_parts, "bist_switch_logits_bwd") != BIST_OK) return BIST_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool s32 = dsw_dtype == BIST_F32;
  if (dparts) {
    const unsigned grid = (unsigned)((rows + 3) / 4);
#define SWX(TT, TSS) hipLaunchKernelGGL((switch_logits_bwd_x_kernel<TT, TSS>), dim3(grid), dim3(256), 0, st, a, (const TT*)W, (long)ldw, (const TSS*)dsw, (long)rows, d, ns)
    if (dtype == BIST_BF16) { if (s32) SWX(bf16_t, float); else SWX(bf16_t, bf16_t); }
    else SWX(float, float);
#undef SWX
    BIST_LAUNCH_CHECK("bist_switch_logits_bwd (parts)");
  }
  if (dW) {
    dim3 grid((unsigned)n_parts, (unsigned)((d / e + 63) / 64));
#define SWW(TT, TWW, TSS) hipLaunchKernelGGL((switch_logits_bwd_w_kernel<TT, TWW, TSS>), grid, dim3(256), 0, st, a, (const TSS*)dsw, (TWW*)dW, (long)lddw, dw_accumulate, db, db_accumulate, (long)rows, d, ns)
    if (dtype == BIST_BF16) {
      if (dw_dtype == BIST_BF16) { if (s32) SWW(bf16_t, bf16_t, float); else SWW(bf16_t, bf16_t, bf16_t); }
      else { if (s32) SWW(bf16_t, float, float); else SWW(bf16_t, float, bf16_t); }
    } else {
      if (dw_dtype == BIST_BF16) SWW(float, bf16_t, float); else SWW(float, float, float);
    }
#undef SWW
    BIST_LAUNCH_CHECK("bist_switch_logits_bwd (weights)");
  }
  return BIST_OK;
}

extern "C" int bist_pointer_attn_fwd(const void* q, const void* k, const void* enc, const uint8_t* mask, int64_t mask_bs, const int64_t* text,
                                     int64_t unk, float* p, void* tv, int64_t B, int32_t Lt, int32_t L, int32_t d, float scale, int32_t dtype,
                                     void* stream) {
  BIST_REQUIRE(q && k && p && B > 0 && Lt >= 1 && Lt <= 32 && L >= 1 && L <= 128 && d > 0 && (!tv || enc),
               "bist_pointer_attn_fwd: bad argument (<= 32 target positions, 1..128 source positions)");
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && d % 64 == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)enc | (uintptr_t)tv) & 15) == 0,
               "bist_pointer_attn_fwd: bf16 / f32, d a multiple of 64, 16-byte aligned rows");
  hipStream_t st = (hipStream_t)stream;
#define PA(TT) hipLaunchKernelGGL(pointer_attn_fwd_kernel<TT>, dim3((unsigned)B), dim3(256), 0, st, (const TT*)q, (const TT*)k, (const TT*)enc, mask, \
                                  (long)mask_bs, (const long*)text, (long)unk, p, (TT*)tv, Lt, L, d, scale)
  if (dtype == BIST_BF16) PA(bf16_t); else PA(float);
#undef PA
  BIST_LAUNCH_CHECK("bist_pointer_attn_fwd");
  return BIST_OK;
}

extern "C" int bist_pointer_attn_bwd(const void* q, const void* k, const void* enc, const float* p, const float* dp, const void* dtv, void* dq,
                                     void* dk, void* denc, int64_t B, int32_t Lt, int32_t L, int32_t d, float scale, int32_t dtype, void* stream) {
  BIST_REQUIRE(q && k && p && dq && dk && (dp || dtv) && B > 0 && Lt >= 1 && Lt <= 32 && L >= 1 && L <= 128 && d > 0 && d % 64 == 0,
               "bist_pointer_attn_bwd: bad argument (<= 32 query rows, <= 128 positions, d a multiple of 64)");
  BIST_REQUIRE(!dtv || (enc && denc), "bist_pointer_attn_bwd: the text vector's gradient needs enc and denc");
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)enc | (uintptr_t)dtv) & 15) == 0,
               "bist_pointer_attn_bwd: bf16 / f32, 16-byte aligned rows");
  hipStream_t st = (hipStream_t)stream;
#define PB(TT) hipLaunchKernelGGL(pointer_attn_bwd_kernel<TT>, dim3((unsigned)B, dtv ? 3u : 2u), dim3(256), 0, st, (const TT*)q, (const TT*)k, (const TT*)enc, p, dp, \
                                  (const TT*)dtv, (TT*)dq, (TT*)dk, (TT*)denc, Lt, L, d, scale)
  if (dtype == BIST_BF16) PB(bf16_t); else PB(float);
#undef PB
  BIST_LAUNCH_CHECK("bist_pointer_attn_bwd");
  return BIST_OK;
}

extern "C" int bist_pointer_decode_mix_fwd(const void* x, const void* tgt, const float* logits, const BistPtrDecSrc* src, int32_t n_ptr,
                                           const void* Wsw, int64_t ldw, const void* bsw, float scale, float* out, int64_t rows, int32_t d,
                                           int32_t V, int32_t dtype, void* stream) {
  BIST_REQUIRE(x && tgt && logits && src && Wsw && bsw && out && rows > 0 && V > 2, "bist_pointer_decode_mix_fwd: bad argument");
  BIST_REQUIRE(n_ptr >= 1 && n_ptr <= 3 && d >= 4 && d <= 1024 && d % 4 == 0 && ldw >= (int64_t)(n_ptr + 2) * d,
               "bist_pointer_decode_mix_fwd: 1..3 pointer sources, d a multiple of 4 and at most 1024, W_sw rows of (n + 2) d");
  BIST_REQUIRE(dtype == BIST_BF16 || dtype == BIST_F32, "bist_pointer_decode_mix_fwd: bf16 or f32");
  PtrDecArgs a;
  a.n = n_ptr;
  for (int j = 0; j < 3; ++j) a.s[j] = PtrDecSrcK{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  for (int j = 0; j < n_ptr; ++j) {
    const BistPtrDecSrc& sj = src[j];
    BIST_REQUIRE(sj.M && sj.c && sj.mask && sj.E && sj.text && sj.L >= 1 && sj.L <= 512 && ((uintptr_t)sj.M & 15) == 0,
                 "bist_pointer_decode_mix_fwd: bad pointer source %d (1..512 positions, M 16-byte aligned)", j);
    a.s[j] = PtrDecSrcK{sj.M, sj.c, sj.mask, sj.E, (const long*)sj.text, sj.p_out, sj.L};
  }
  hipStream_t st = (hipStream_t)stream;
#define BIST_PDM(T, LR) hipLaunchKernelGGL((pointer_decode_mix_kernel<T, LR>), dim3((unsigned)rows), dim3(256), 0, st, (const T*)x, (const T*)tgt, \
                                          logits, a, (const T*)Wsw, (long)ldw, (const T*)bsw, scale, out, d, V)
  if (dtype == BIST_BF16) { if (V <= 4096) BIST_PDM(bf16_t, true); else BIST_PDM(bf16_t, false); }
  else { if (V <= 4096) BIST_PDM(float, true); else BIST_PDM(float, false); }
#undef BIST_PDM
  BIST_LAUNCH_CHECK("bist_pointer_decode_mix_fwd");
  return BIST_OK;
}

extern "C" int bist_text_vector_fwd(const float* p, const void* enc, void* out, int64_t rows, int32_t Lt, int32_t L, int32_t d, int32_t dtype,
                                    void* stream) {
  BIST_REQUIRE(p && enc && out && rows > 0 && Lt > 0 && L > 0 && d > 0, "bist_text_vector_fwd: bad argument");
  const int e = dtype == BIST_BF16 ? 8 : 4;
  BIST_REQUIRE((dtype == BIST_BF16 || dtype == BIST_F32) && d % e == 0 && d / e <= 256 && ((uintptr_t)enc | (uintptr_t)out) % 16 == 0,
               "bist_text_vector_fwd: d must be a multiple of %d and at most %d, rows 16-byte aligned", e, 256 * e);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BIST_BF16) hipLaunchKernelGGL(text_vector_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, st, p, (const bf16_t*)enc, (bf16_t*)out, Lt, L, d);
  else hipLaunchKernelGGL(text_vector_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, p, (const float*)enc, (float*)out, Lt, L, d);
  BIST_LAUNCH_CHECK("bist_text_vector_fwd");
  return BIST_OK;
}

extern "C" int bist_log_softmax_fwd(const float* x, float* y, int64_t rows, int32_t V, void* stream) {
  BIST_REQUIRE(x && y && rows > 0 && V > 0, "bist_log_softmax_fwd: bad argument");
  hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, V);
  BIST_LAUNCH_CHECK("bist_log_softmax_fwd");
  return BIST_OK;
}

extern "C" int bist_label_smoothing_fwd(const float* logp, const int64_t* target, float* row_loss, int64_t rows, int32_t V,
                                        float smoothing, int32_t pad, void* stream) {
  BIST_REQUIRE(logp && target && row_loss && rows > 0 && V > 2, "bist_label_smoothing_fwd: bad argument");
  BIST_REQUIRE(smoothing >= 0.f && smoothing < 1.f, "bist_label_smoothing_fwd: smoothing out of range");
  hipLaunchKernelGGL(label_smoothing_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logp, (const long*)target,
                     row_loss, V, smoothing, pad);
  BIST_LAUNCH_CHECK("bist_label_smoothing_fwd");
  return BIST_OK;
}

extern "C" int bist_xent_smooth_fwd(const float* logits, const int64_t* target, int64_t M, int64_t rows, int32_t V, float smoothing, int32_t pad,
                                    float* row_loss, float* lse, void* stream) {
  BIST_REQUIRE(logits && target && row_loss && lse && M > 0 && rows > 0 && rows % M == 0 && V > 2, "bist_xent_smooth_fwd: bad argument (rows a multiple of M)");
  BIST_REQUIRE(smoothing >= 0.f && smoothing < 1.f, "bist_xent_smooth_fwd: smoothing out of range");
  hipLaunchKernelGGL(xent_smooth_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (const long*)target, (long)M, row_loss,
                     lse, V, smoothing, pad);
  BIST_LAUNCH_CHECK("bist_xent_smooth_fwd");
  return BIST_OK;
}

extern "C" int bist_xent_smooth_bwd(const float* logits, const float* lse, const int64_t* target, int64_t M, int64_t rows, const float* gout,
                                    int32_t gout_stride, const int64_t* denom, void* dlogits, int32_t dlogits_dtype, int32_t V, float smoothing,
                                    int32_t pad, void* stream) {
  BIST_REQUIRE(logits && lse && target && gout && dlogits && M > 0 && rows > 0 && rows % M == 0 && V > 2 && (gout_stride == 0 || gout_stride == 1),
               "bist_xent_smooth_bwd: bad argument (gout_stride 1 = one upstream gradient per group, 0 = one for all)");
  BIST_REQUIRE(dlogits_dtype == BIST_BF16 || dlogits_dtype == BIST_F32, "bist_xent_smooth_bwd: bad gradient dtype");
  hipStream_t st = (hipStream_t)stream;
  if (dlogits_dtype == BIST_BF16)
    hipLaunchKernelGGL(xent_smooth_bwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, st, logits, lse, (const long*)target, (long)M, gout,
                       gout_stride, (const long*)denom, (bf16_t*)dlogits, V, smoothing, pad);
  else
    hipLaunchKernelGGL(xent_smooth_bwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, logits, lse, (const long*)target, (long)M, gout,
                       gout_stride, (const long*)denom, (float*)dlogits, V, smoothing, pad);
  BIST_LAUNCH_CHECK("bist_xent_smooth_bwd");
  return BIST_OK;
}

extern "C" int bist_sum_div_groups(const float* x, int64_t M, int32_t G, const int64_t* denom, float* out, int32_t out_stride, void* stream) {
  BIST_REQUIRE(x && out && M > 0 && G > 0 && out_stride >= 1, "bist_sum_div_groups: bad argument");
  hipLaunchKernelGGL(sum_div_groups_kernel, dim3((unsigned)G), dim3(256), 0, (hipStream_t)stream, x, (long)M, (const long*)denom, out, out_stride);
  BIST_LAUNCH_CHECK("bist_sum_div_groups");
  return BIST_OK;
}

extern "C" int bist_stack_rows(const void* const* srcs, int32_t n, void* out, int64_t bytes_each, void* stream) {
  BIST_REQUIRE(srcs && out && n >= 1 && n <= 4 && bytes_each > 0 && bytes_each % 16 == 0 && ((uintptr_t)out & 15) == 0, "bist_stack_rows: 1..4 sources of a multiple of 16 bytes");
  StackSrc a;
  for (int j = 0; j < 4; ++j) a.s[j] = nullptr;
  for (int j = 0; j < n; ++j) {
    BIST_REQUIRE(srcs[j] && ((uintptr_t)srcs[j] & 15) == 0, "bist_stack_rows: source %d null or not 16-byte aligned", j);
    a.s[j] = (const uint4*)srcs[j];
  }
  const long n16 = bytes_each / 16;
  hipLaunchKernelGGL(stack_rows_kernel, dim3((unsigned)((n16 + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, a, (uint4*)out, n16);
  BIST_LAUNCH_CHECK("bist_stack_rows");
  return BIST_OK;
}

extern "C" int bist_sum_div(const float* x, int64_t n, const int64_t* denom, float* out, int32_t accumulate, void* stream) {
  BIST_REQUIRE(x && out && n > 0, "bist_sum_div: bad argument");
  hipLaunchKernelGGL(sum_div_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, (long)n, (const long*)denom, out, accumulate);
  BIST_LAUNCH_CHECK("bist_sum_div");
  return BIST_OK;
}

// =============================================================================================
// backward of the output heads
// =============================================================================================
namespace {

// pointer_mix backward.  out = log(mix); given dout:
//   dmix[v] = dout[v] / mix[v];  dsw[n] = sum_v dmix[v] pv[v];  dlogits = pv * sw[n] * (dmix - dsw[n])
//   dsw[j] = sum_t dmix[text_j[t]] p_j[t];  dp_j[t] = sw[j] * dmix[text_j[t]];  dsw_logits = softmax backward of sw
__global__ __launch_bounds__(256) void pointer_mix_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ sw_logits,
                                                              PtrArgs a, const float* __restrict__ out, const float* __restrict__ dout,
                                                              float* __restrict__ dlogits, float* __restrict__ dsw_logits,
                                                              float* dp0, float* dp1, float* dp2, int V, int Lt, int sigmoid_switch) {
  __shared__ float red[4];
  __shared__ float dsw_sh[4];
  const long row = blockIdx.x;
  const int b = (int)(row / Lt);
  const float* lg = logits + row * V;
  const float* o = out + row * V;
  const float* go = dout + row * V;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, lg[v]);
  mx = block_reduce(mx, red, true);
  float den = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) den += expf(lg[v] - mx);
  den = block_reduce(den, red, false);
  float sw[4];
  const int ns = a.n + 1;
  float sig = 0.f;
  if (sigmoid_switch) { sig = 1.f / (1.f + expf(-sw_logits[row])); sw[0] = 1.f - sig; sw[1] = sig; }
  else {
    float m2 = -INFINITY, d2 = 0.f;
    for (int j = 0; j < ns; ++j) m2 = fmaxf(m2, sw_logits[row * ns + j]);
    for (int j = 0; j < ns; ++j) { sw[j] = expf(sw_logits[row * ns + j] - m2); d2 += sw[j]; }
    for (int j = 0; j < ns; ++j) sw[j] /= d2;
  }
  // vocabulary branch
  float acc = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) acc += go[v] * expf(-o[v]) * (expf(lg[v] - mx) / den);
  const float dswn = block_reduce(acc, red, false);            // = sum_v dmix pv
  for (int v = threadIdx.x; v < V; v += 256) {
    const float pv = expf(lg[v] - mx) / den;
    dlogits[row * V + v] = pv * sw[a.n] * (go[v] * expf(-o[v]) - dswn);
  }
  // copy branches (few dozen entries): one wave, lanes over t
  float* dps[3] = {dp0, dp1, dp2};
  if (threadIdx.x < 64) {
    for (int j = 0; j < a.n; ++j) {
      const float* p = a.s[j].p + row * a.s[j].L;
      const long* text = a.s[j].text + (long)b * a.s[j].L;
      float part = 0.f;
      for (int t = threadIdx.x; t < a.s[j].L; t += 64) {
        const long v = text[t];
        const float dm = go[v] * expf(-o[v]);
        part += dm * p[t];
        dps[j][row * a.s[j].L + t] = sw[j] * dm;
      }
      part = wave_sum(part);
      if (threadIdx.x == 0) dsw_sh[j] = part;
    }
    if (threadIdx.x == 0) {
      dsw_sh[a.n] = dswn;
      if (sigmoid_switch) {
        dsw_logits[row] = (dsw_sh[1] - dsw_sh[0]) * sig * (1.f - sig);
      } else {
        float dot = 0.f;
        for (int j = 0; j < ns; ++j) dot += sw[j] * dsw_sh[j];
        for (int j = 0; j < ns; ++j) dsw_logits[row * ns + j] = sw[j] * (dsw_sh[j] - dot);
      }
    }
  }
}

// y = log_softmax(x):  dx = dy - exp(y) * sum(dy)
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int V) {
  __shared__ float red[4];
  const long off = (long)blockIdx.x * V;
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) s += dy[off + v];
  s = block_reduce(s, red, false);
  for (int v = threadIdx.x; v < V; v += 256) dx[off + v] = dy[off + v] - expf(y[off + v]) * s;
}

// loss = sum_rows KL(row) / denom  ->  dlogp[row, v] = -td[row, v] * gout / denom
__global__ __launch_bounds__(256) void label_smoothing_bwd_kernel(const long* __restrict__ target, const float* __restrict__ gout,
                                                                  const long* __restrict__ denom, float* __restrict__ dlogp, int V,
                                                                  float smoothing, int pad) {
  const long row = blockIdx.x;
  const long t = target[row];
  const float sc = -gout[0] / (denom ? (float)denom[0] : 1.f);
  const float s = smoothing / (float)(V - 2), conf = 1.f - smoothing;
  float* g = dlogp + row * V;
  for (int v = threadIdx.x; v < V; v += 256) {
    float td = 0.f;
    if (t != pad) td = (v == t) ? conf : (v == pad ? 0.f : s);
    g[v] = sc * td;
  }
}

}  // namespace

extern "C" int bist_pointer_mix_bwd(const float* logits, const float* switch_logits, int32_t n_ptr, const float* const* ptr_p,
                                    const int64_t* const* ptr_text, const int32_t* ptr_len, const float* out, const float* dout,
                                    float* dlogits, float* dswitch_logits, float* const* dptr_p, int64_t rows, int32_t Lt, int32_t V,
                                    int32_t sigmoid_switch, void* stream) {
  BIST_REQUIRE(logits && switch_logits && out && dout && dlogits && dswitch_logits && dptr_p, "bist_pointer_mix_bwd: null pointer");
  BIST_REQUIRE(n_ptr >= 1 && n_ptr <= 3 && rows > 0 && Lt > 0 && V > 2, "bist_pointer_mix_bwd: bad argument");
  PtrArgs a;
  a.n = n_ptr;
  float* dps[3] = {nullptr, nullptr, nullptr};
  for (int j = 0; j < 3; ++j) a.s[j] = PtrSrc{nullptr, nullptr, 0};
  for (int j = 0; j < n_ptr; ++j) {
    BIST_REQUIRE(ptr_p[j] && ptr_text[j] && ptr_len[j] > 0 && dptr_p[j], "bist_pointer_mix_bwd: bad pointer source %d", j);
    a.s[j] = PtrSrc{ptr_p[j], (const long*)ptr_text[j], ptr_len[j]};
    dps[j] = dptr_p[j];
  }
  hipLaunchKernelGGL(pointer_mix_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, switch_logits, a, out, dout,
                     dlogits, dswitch_logits, dps[0], dps[1], dps[2], V, Lt, sigmoid_switch);
  BIST_LAUNCH_CHECK("bist_pointer_mix_bwd");
  return BIST_OK;
}

extern "C" int bist_log_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int32_t V, void* stream) {
  BIST_REQUIRE(y && dy && dx && rows > 0 && V > 0, "bist_log_softmax_bwd: bad argument");
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, y, dy, dx, V);
  BIST_LAUNCH_CHECK("bist_log_softmax_bwd");
  return BIST_OK;
}

extern "C" int bist_label_smoothing_bwd(const int64_t* target, const float* gout, const int64_t* denom, float* dlogp, int64_t rows,
                                        int32_t V, float smoothing, int32_t pad, void* stream) {
  BIST_REQUIRE(target && gout && dlogp && rows > 0 && V > 2, "bist_label_smoothing_bwd: bad argument");
  hipLaunchKernelGGL(label_smoothing_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const long*)target, gout,
                     (const long*)denom, dlogp, V, smoothing, pad);
  BIST_LAUNCH_CHECK("bist_label_smoothing_bwd");
  return BIST_OK;
}
