// Stage-1 core of t2s / s2t on the matrix cores (bf16 path; dk = 64, Lq <= 32, K <= 128).
//
// Per workgroup = (clip b, head hh, chunk of Gc groups):
//   A. gather the score slab  sc[i][gl][k]  (f32, coalesced along the axis that is contiguous for the
//      direction), apply the temporal mask by REPLACING with -1e9 (modules.py:60);
//   B. row softmax (one thread per (i, gl) row, padded rows -> conflict-free), keep P as f32 in the slab
//      (backward) and as a bf16 MFMA image  pimg[gl][32 query rows][KPAD keys]  (zero padded);
//   C. one wave per group: stage the group's V tile [K keys][64 channels] in LDS as it lies in HBM and run
//        forward :  O[i,c]  = sum_k P[i,k] V[k,c]            A = pimg rows (ds_read_b128), B = V via ds_read_b64_tr_b16
//        backward:  dP[i,k] = sum_c dO[i,c] V[k,c]           A = dO rows,  B = V rows (both ds_read_b128)
//                   dS      = P (dP - rowsum(P dP)), 0 where masked  -> dscores (f32)
//                   dV[k,c] = sum_i P[i,k] dO[i,c]           A = pimg^T, B = dO^T (both ds_read_b64_tr_b16)
// Nothing is permuted or expanded in HBM: group g of direction 0 (t2s) is video column s with keys t
// (rows t*S+s of V), of direction 1 (s2t) it is frame t with keys s (rows t*S+s, contiguous).
#include "common.hpp"

namespace {

constexpr float MASK_FILL = -1e9f;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// fragment of 16 rows x 32 k from a row-major bf16 image [rows][pitch] (k contiguous): lane (x, kg) -> row0+x, k0+8kg..+7
__device__ __forceinline__ uint4 frag_rows(const bf16_t* img, int pitch, int row0, int k0, int lane) {
  return *reinterpret_cast<const uint4*>(img + (row0 + (lane & 15)) * pitch + k0 + (lane >> 4) * 8);
}
// the same fragment from an image stored [k rows][pitch cols] (the logical row index is the COLUMN): transposing read
__device__ __forceinline__ uint4 frag_cols(const bf16_t* img, int pitch, int col0, int k0, int lane) {
  const int x = lane & 15, kg = lane >> 4, q = x >> 2, p = x & 3;
  const bf16_t* a0 = img + (k0 + kg * 8 + q) * pitch + col0 + 4 * p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * pitch));
  uint4 r;
  r.x = __builtin_bit_cast(uint2, lo).x; r.y = __builtin_bit_cast(uint2, lo).y;
  r.z = __builtin_bit_cast(uint2, hi).x; r.w = __builtin_bit_cast(uint2, hi).y;
  return r;
}

struct St1Args {
  const void* scores; const bf16_t* V; const unsigned char* tmask;
  bf16_t* O;                      // forward output
  const bf16_t* dO; float* dscores; bf16_t* dV;   // backward
  int T, S, Lq, h; long ldv, lddv; int dir, Gc;
};

template <typename TS, int KSTEPS, bool BWD>
__global__ __launch_bounds__(256) void st1_mfma_kernel(const St1Args a) {
  constexpr int KPAD = 32 * KSTEPS, DK = 64, NKF = KPAD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T_ = a.T, S_ = a.S, Lq = a.Lq, h = a.h, dir = a.dir, Gc = a.Gc;
  const int G = dir == 0 ? S_ : T_, Kn = dir == 0 ? T_ : S_, KP = Kn + 1;
  const int g0 = blockIdx.x * Gc, hh = blockIdx.y, b = blockIdx.z;
  const int gc = min(Gc, G - g0);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long TS_ = (long)T_ * S_;
  const int d = h * DK;

  float* slab = reinterpret_cast<float*>(smem);                              // [Lq][Gc][KP]
  const int slab_bytes = ((Lq * Gc * KP * 4 + 15) / 16) * 16;
  bf16_t* pimg = reinterpret_cast<bf16_t*>(smem + slab_bytes);               // [Gc][32][KPAD]
  bf16_t* vimg = pimg + Gc * 32 * KPAD;                                      // [4 waves][KPAD][64]
  bf16_t* doimg = vimg + 4 * KPAD * DK;                                      // [4 waves][32][64]   (backward only)
  {
    const int n16 = (Gc * 32 * KPAD + 4 * KPAD * DK + (BWD ? 4 * 32 * DK : 0)) / 8;
    for (int i = tid; i < n16; i += 256) reinterpret_cast<uint4*>(pimg)[i] = make_uint4(0, 0, 0, 0);
  }
  const TS* sc = reinterpret_cast<const TS*>(a.scores) + (long)b * Lq * h * TS_;
  const unsigned char* mk = (dir == 0 && a.tmask) ? a.tmask + (long)b * T_ : nullptr;
  // ---- A: score slab --------------------------------------------------------------------------------
  const int total = Lq * gc * Kn;
  for (int idx = tid; idx < total; idx += 256) {
    int i, gl, k;
    if (dir == 0) { gl = idx % gc; const int t2 = idx / gc; k = t2 % Kn; i = t2 / Kn; }
    else          { k = idx % Kn; const int t2 = idx / Kn; gl = t2 % gc; i = t2 / gc; }
    const int g = g0 + gl;
    const long col = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
    float v = to_f(sc[((long)i * h + hh) * TS_ + col]);
    if (mk && mk[k] == 0) v = MASK_FILL;
    slab[((long)i * gc + gl) * KP + k] = v;
  }
  __syncthreads();
  // ---- B: softmax rows -> slab (f32) and pimg (bf16) -------------------------------------------------
  for (int r = tid; r < Lq * gc; r += 256) {
    const int i = r / gc, gl = r % gc;
    float* p = slab + (long)r * KP;
    float mx = -INFINITY;
    for (int k = 0; k < Kn; ++k) mx = fmaxf(mx, p[k]);
    float den = 0.f;
    for (int k = 0; k < Kn; ++k) { const float e = expf(p[k] - mx); p[k] = e; den += e; }
    const float inv = 1.f / den;
    bf16_t* pi = pimg + ((long)gl * 32 + i) * KPAD;
    for (int k = 0; k < Kn; ++k) { const float q = p[k] * inv; p[k] = q; pi[k] = (bf16_t)q; }
  }
  __syncthreads();
  // ---- C: one wave per group -------------------------------------------------------------------------
  const bf16_t* Vb = a.V + (long)b * TS_ * a.ldv + hh * DK;
  bf16_t* vt = vimg + w * KPAD * DK;
  bf16_t* dt = doimg + w * 32 * DK;
  const int x = lane & 15, lg = lane >> 4;
  for (int gl = w; gl < gc; gl += 4) {
    const int g = g0 + gl;
    for (int k = lane >> 3; k < Kn; k += 8) {                      // V tile: 8 rows x 128 B per wave instruction
      const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
      *reinterpret_cast<uint4*>(vt + k * DK + (lane & 7) * 8) = *reinterpret_cast<const uint4*>(Vb + row * a.ldv + (lane & 7) * 8);
    }
    const bf16_t* pg = pimg + (long)gl * 32 * KPAD;
    if constexpr (!BWD) {
      f32x4 acc[2][4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        uint4 af[2], bfr[4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[mi] = frag_rows(pg, KPAD, mi * 16, ks * 32, lane);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bfr[ni] = frag_cols(vt, DK, ni * 16, ks * 32, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_bf16(af[mi], bfr[ni], acc[mi][ni]);
      }
      bf16_t* Ob = a.O + (((long)b * G + g) * Lq) * d + hh * DK;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = mi * 16 + lg * 4 + r;
          if (i < Lq) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) Ob[(long)i * d + ni * 16 + x] = (bf16_t)acc[mi][ni][r];
          }
        }
    } else {
      const bf16_t* dOb = a.dO + (((long)b * G + g) * Lq) * d + hh * DK;
      for (int i = lane >> 3; i < Lq; i += 8)                       // dO tile [Lq][64]; rows >= Lq stay zero
        *reinterpret_cast<uint4*>(dt + i * DK + (lane & 7) * 8) = *reinterpret_cast<const uint4*>(dOb + (long)i * d + (lane & 7) * 8);
      // dP = dO . V^T   (M = i, N = k, K = c)
      f32x4 dp[2][NKF];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NKF; ++ni) dp[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 af[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[mi] = frag_rows(dt, DK, mi * 16, ks * 32, lane);
#pragma unroll
        for (int ni = 0; ni < NKF; ++ni) {
          const uint4 bfr = frag_rows(vt, DK, ni * 16, ks * 32, lane);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) dp[mi][ni] = mfma_bf16(af[mi], bfr, dp[mi][ni]);
        }
      }
      // dS = P (dP - sum_k P dP), 0 where masked; C layout: k = ni*16 + x, i = mi*16 + lg*4 + r
      float* dsc = a.dscores + (long)b * Lq * h * TS_;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = mi * 16 + lg * 4 + r;
          float pv[NKF], part = 0.f;
#pragma unroll
          for (int ni = 0; ni < NKF; ++ni) {
            const int k = ni * 16 + x;
            pv[ni] = (i < Lq && k < Kn) ? slab[((long)i * gc + gl) * KP + k] : 0.f;
            part += pv[ni] * dp[mi][ni][r];
          }
          part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
          part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
          if (i < Lq) {
#pragma unroll
            for (int ni = 0; ni < NKF; ++ni) {
              const int k = ni * 16 + x;
              if (k < Kn) {
                float gq = pv[ni] * (dp[mi][ni][r] - part);
                if (mk && mk[k] == 0) gq = 0.f;
                const long col = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
                dsc[((long)i * h + hh) * TS_ + col] = gq;
              }
            }
          }
        }
      // dV = P^T . dO   (M = k, N = c, K = i)
      f32x4 dv[NKF][4];
#pragma unroll
      for (int mi = 0; mi < NKF; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) dv[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        uint4 bfr[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bfr[ni] = frag_cols(dt, DK, ni * 16, 0, lane);
#pragma unroll
        for (int mi = 0; mi < NKF; ++mi) {
          const uint4 af = frag_cols(pg, KPAD, mi * 16, 0, lane);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) dv[mi][ni] = mfma_bf16(af, bfr[ni], dv[mi][ni]);
        }
      }
      bf16_t* dVb = a.dV + (long)b * TS_ * a.lddv + hh * DK;
#pragma unroll
      for (int mi = 0; mi < NKF; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = mi * 16 + lg * 4 + r;
          if (k < Kn) {
            const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) dVb[row * a.lddv + ni * 16 + x] = (bf16_t)dv[mi][ni][r];
          }
        }
    }
  }
}

template <typename TS, int KSTEPS, bool BWD>
int launch_one(const St1Args& a, int B, size_t lds, hipStream_t st) {
  static bool attr_set = false;       // allow > 64 KiB of dynamic LDS once per instantiation
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&st1_mfma_kernel<TS, KSTEPS, BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int G = a.dir == 0 ? a.S : a.T;
  dim3 grid((unsigned)((G + a.Gc - 1) / a.Gc), (unsigned)a.h, (unsigned)B);
  hipLaunchKernelGGL((st1_mfma_kernel<TS, KSTEPS, BWD>), grid, dim3(256), lds, st, a);
  return hipGetLastError() == hipSuccess ? 1 : -1;
}

}  // namespace

// returns 1 if launched, 0 if the shape is outside this kernel's envelope (caller falls back), -1 on launch error
int bist_st1_mfma(const void* scores, int sc_is_f32, const void* V, const unsigned char* tmask, void* O, const void* dO,
                  float* dscores, void* dV, int B, int T, int S, int Lq, int h, int dk, long ldv, long lddv, int dir,
                  int bwd, hipStream_t st) {
  const int G = dir == 0 ? S : T, Kn = dir == 0 ? T : S;
  if (dk != 64 || Lq > 32 || Kn > 128 || (ldv % 8) != 0 || ((uintptr_t)V % 16) != 0) return 0;
  if (bwd && ((lddv % 8) != 0 || ((long)h * dk) % 8 != 0 || !sc_is_f32)) return 0;
  const int ksteps = Kn <= 32 ? 1 : (Kn <= 64 ? 2 : 4);
  const int kpad = 32 * ksteps;
  const long per_g = (long)Lq * (Kn + 1) * 4 + 32L * kpad * 2;
  const long fixed = 4L * kpad * 64 * 2 + (bwd ? 4L * 32 * 64 * 2 : 0) + 16;
  int Gc = (int)((150 * 1024 - fixed) / per_g);
  if (Gc < 1) return 0;
  if (Gc > 8) Gc = 8;                       // two groups per wave is enough to amortise phases A/B
  if (Gc > G) Gc = G;
  while (Gc > 4 && (long)((G + Gc - 1) / Gc) * h * B < 1024) --Gc;
  const size_t lds = (size_t)(((long)Lq * Gc * (Kn + 1) * 4 + 15) / 16 * 16) + (size_t)Gc * 32 * kpad * 2 + (size_t)fixed;
  St1Args a{scores, (const bf16_t*)V, tmask, (bf16_t*)O, (const bf16_t*)dO, dscores, (bf16_t*)dV, T, S, Lq, h, ldv, lddv, dir, Gc};
#define GO(TS_, KS_)                                                                   \
  return bwd ? launch_one<TS_, KS_, true>(a, B, lds, st) : launch_one<TS_, KS_, false>(a, B, lds, st)
  if (sc_is_f32) {
    if (ksteps == 1) GO(float, 1); else if (ksteps == 2) GO(float, 2); else GO(float, 4);
  } else {
    if (bwd) return 0;
    if (ksteps == 1) return launch_one<bf16_t, 1, false>(a, B, lds, st);
    if (ksteps == 2) return launch_one<bf16_t, 2, false>(a, B, lds, st);
    return launch_one<bf16_t, 4, false>(a, B, lds, st);
  }
#undef GO
}
