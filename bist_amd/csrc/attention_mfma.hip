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
#include <stdlib.h>

namespace {

// 1 = launched; -1 = the launch failed (reason left in bist_last_error)
inline int launched(const char* name) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 1;
  bist_set_error("%s: launch failed: %s", name, hipGetErrorString(e));
  return -1;
}

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
  const bf16_t* dO; void* dscores; bf16_t* dV;   // backward (dscores f32, or bf16 when dsc_bf16)
  int T, S, Lq, h; long ldv, lddv; int dir, Gc;
  int dsc_bf16;
  int p_kp;     // > 0 (backward): `scores` holds the PROBABILITIES before dropout as [B, G, h, Lq, p_kp] f32 (the fused training forward's
                // side output): phases A and B are one contiguous read per (query row, group) instead of a score gather + softmax
  int dbg;      // timing ablation only (BIST_ST1_DBG): bit0 skip slab gather, bit1 skip softmax, bit2 skip phase C
  DropArg drop; // dropout of the probabilities: the slab keeps P, the bf16 image that feeds the MFMAs holds mask*P/(1-p)
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
  bf16_t* vimg = pimg + Gc * 32 * KPAD;                                      // [Gc waves][KPAD][64]
  bf16_t* doimg = vimg + Gc * KPAD * DK;                                     // [Gc waves][32][64]   (backward only)
  bf16_t* vt = vimg + w * KPAD * DK;
  bf16_t* dt = doimg + w * 32 * DK;
  {   // zero what the phases below do not overwrite: the P images, the key-padding rows of the V tiles, the query-padding
      // rows of the dO tiles (padding must be finite: it meets zero probabilities / zero rows in the MFMAs)
    for (int q = tid; q < Gc * 32 * KPAD / 8; q += 256) reinterpret_cast<uint4*>(pimg)[q] = make_uint4(0, 0, 0, 0);
    if (w < Gc) {
      for (int q = lane; q < (KPAD - Kn) * 8; q += 64) reinterpret_cast<uint4*>(vt + Kn * DK)[q] = make_uint4(0, 0, 0, 0);
      if constexpr (BWD)
        for (int q = lane; q < (32 - Lq) * 8; q += 64) reinterpret_cast<uint4*>(dt + Lq * DK)[q] = make_uint4(0, 0, 0, 0);
    }
  }
  // This wave's group (one per wave: Gc <= 4): start its V tile (and dO tile) now, so the HBM latency hides
  // behind the slab gather and the softmax.  LDS image as it lies in HBM: [k][64 channels].
  const bf16_t* Vb = a.V + (long)b * TS_ * a.ldv + hh * DK;
  if (w < gc && !(a.dbg & 4)) {
    const int g = g0 + w;
    for (int k = lane >> 3; k < Kn; k += 8) {                      // 8 rows x 128 B per wave instruction
      const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
      *reinterpret_cast<uint4*>(vt + k * DK + (lane & 7) * 8) = *reinterpret_cast<const uint4*>(Vb + row * a.ldv + (lane & 7) * 8);
    }
    if constexpr (BWD) {
      const bf16_t* dOb = a.dO + (((long)b * G + g) * Lq) * d + hh * DK;
      for (int i = lane >> 3; i < Lq; i += 8)                       // dO tile [Lq][64]; rows >= Lq stay zero
        *reinterpret_cast<uint4*>(dt + i * DK + (lane & 7) * 8) = *reinterpret_cast<const uint4*>(dOb + (long)i * d + (lane & 7) * 8);
    }
  }
  const TS* sc = reinterpret_cast<const TS*>(a.scores) + (long)b * Lq * h * TS_;
  const unsigned char* mk = a.tmask ? a.tmask + (long)b * Kn : nullptr;      // key mask [B, K]
  // ---- A: score slab ---------------------------------------------------------------------------------------
  if ((a.dbg & 1) || a.p_kp > 0) {
  } else if (dir == 1) {
    // group = frame t, keys = regions: for a fixed query row the gc*S scores are ONE contiguous run -> 16-byte loads
    const int run = gc * Kn;
    const TS* base = sc + (long)hh * TS_ + (long)g0 * S_;
    const bool vec = sizeof(TS) == 4 && (TS_ & 3) == 0 && (((long)g0 * S_) & 3) == 0 && ((uintptr_t)sc & 15) == 0;
    int gl0 = 0, k0 = 4 * lane;                       // (group, key) of this lane's first element, found once
    while (k0 >= Kn) { k0 -= Kn; ++gl0; }
    for (int i = w; i < Lq; i += 4) {
      const TS* src = base + (long)i * h * TS_;
      float* dst = slab + (long)i * gc * KP;
      if (vec && run <= 256) {
        const int e = 4 * lane;
        if (e < run) {
          float v[4];
          if (e + 3 < run) { const float4 q = *reinterpret_cast<const float4*>(src + e); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
          else { for (int j = 0; j < 4; ++j) v[j] = e + j < run ? to_f(src[e + j]) : 0.f; }
          int gl = gl0, k = k0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (e + j < run) dst[gl * KP + k] = (mk && mk[k] == 0) ? MASK_FILL : v[j];
            if (++k == Kn) { k = 0; ++gl; }
          }
        }
      } else {
        int gl = 0, k = lane;
        while (k >= Kn) { k -= Kn; ++gl; }
        for (int e = lane; e < run; e += 64) {
          dst[gl * KP + k] = (mk && mk[k] == 0) ? MASK_FILL : to_f(src[e]);
          k += 64;
          while (k >= Kn) { k -= Kn; ++gl; }
        }
      }
    }
  } else {
    // group = region s, keys = frames: the gc scores of one (query row, key) are adjacent; lane = (key & 15, group)
    const int gl = lane & 3, kk = lane >> 2;
    for (int kb0 = 0; kb0 < Kn; kb0 += 16) {
      const int k = kb0 + kk;
      const bool ok = gl < gc && k < Kn;
      const bool masked = ok && mk && mk[k] == 0;
      const TS* src = sc + (long)hh * TS_ + (long)k * S_ + g0 + gl;
      float v[8];                                    // this wave's rows i = w, w+4, ... (Lq <= 32): all loads in flight together
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = w + 4 * u;
        v[u] = (ok && !masked && i < Lq) ? to_f(src[(long)i * h * TS_]) : MASK_FILL;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = w + 4 * u;
        if (ok && i < Lq) slab[((long)i * gc + gl) * KP + k] = v[u];
      }
    }
  }
  __syncthreads();
  // ---- B: softmax rows -> slab (f32) and pimg (bf16); four lanes share a row, each keeps its slice in registers ----
  const unsigned long long dkey = a.drop.p > 0.f ? a.drop.key() : 0ULL;
  const float dks = a.drop.p > 0.f ? a.drop.keep_scale() : 1.f;
  {
    constexpr int PER = (KPAD + 3) / 4;              // keys per lane (k = part, part+4, ...)
    if (a.p_kp > 0) {
      // the probabilities themselves (before dropout), saved by the fused training forward as [B, G, h, Lq, p_kp] f32: the [Lq][p_kp]
      // block of a (group, head) is contiguous -- 16 bytes per lane, straight into the slab and (with the dropout mask) the bf16 image
      const int kp4 = a.p_kp >> 2;
      for (int e = tid; e < gc * Lq * kp4; e += 256) {
        const int gl = e / (Lq * kp4), rem = e - gl * (Lq * kp4), i = rem / kp4, k0 = (rem - i * kp4) * 4;
        const float4 q4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.scores) +
                                                            ((((long)b * G + (g0 + gl)) * h + hh) * Lq + i) * a.p_kp + k0);
        const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
        float* p = slab + ((long)i * gc + gl) * KP;
        bf16_t* pi = pimg + ((long)gl * 32 + i) * KPAD;
        const unsigned long long dbase = ((((unsigned long long)b * G + (g0 + gl)) * h + hh) * Lq + i) * Kn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + j;
          if (k < Kn) {
            p[k] = qv[j];
            pi[k] = (bf16_t)(a.drop.p > 0.f ? qv[j] * drop_mul(dkey, dbase + k, a.drop.p, dks) : qv[j]);
          }
        }
      }
    }
    const int rows = ((a.dbg & 2) || a.p_kp > 0) ? 0 : Lq * gc;
    for (int r0 = 0; r0 < rows; r0 += 64) {
      const int r = r0 + (tid >> 2), part = tid & 3;
      const bool act = r < rows;
      float* p = slab + (long)(act ? r : 0) * KP;
      float v[PER];
      float mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int k = part + 4 * u;
        v[u] = (act && k < Kn) ? p[k] : -INFINITY;
        mx = fmaxf(mx, v[u]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      float den = 0.f;
#pragma unroll
      for (int u = 0; u < PER; ++u) { v[u] = act ? expf(v[u] - mx) : 0.f; den += v[u]; }
      den += __shfl_xor(den, 1, 64); den += __shfl_xor(den, 2, 64);
      if (act) {
        const float inv = 1.f / den;
        const int i = r / gc, gl = r - i * gc;
        bf16_t* pi = pimg + ((long)gl * 32 + i) * KPAD;
        const unsigned long long dbase = ((((unsigned long long)b * G + (g0 + gl)) * h + hh) * Lq + i) * Kn;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int k = part + 4 * u;
          if (k < Kn) {
            const float q = v[u] * inv;
            p[k] = q;
            pi[k] = (bf16_t)(a.drop.p > 0.f ? q * drop_mul(dkey, dbase + k, a.drop.p, dks) : q);
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- C: one wave per group -------------------------------------------------------------------------
  const int x = lane & 15, lg = lane >> 4;
  for (int gl = w; gl < ((a.dbg & 4) ? 0 : gc); gl += 4) {          // runs at most once (Gc <= 4)
    const int g = g0 + gl;
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
      // the V tile is consumed: reuse it to turn the fragment layout into 128-byte output rows
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) vt[(mi * 16 + lg * 4 + r) * DK + ni * 16 + x] = (bf16_t)acc[mi][ni][r];
      bf16_t* Ob = a.O + (((long)b * G + g) * Lq) * d + hh * DK;
      for (int i = lane >> 3; i < Lq; i += 8)
        *reinterpret_cast<uint4*>(Ob + (long)i * d + (lane & 7) * 8) = *reinterpret_cast<const uint4*>(vt + i * DK + (lane & 7) * 8);
    } else {
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
      float* dsc = reinterpret_cast<float*>(a.dscores) + (long)b * Lq * h * TS_;
      bf16_t* dsc16 = reinterpret_cast<bf16_t*>(a.dscores) + (long)b * Lq * h * TS_;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = mi * 16 + lg * 4 + r;
          float pv[NKF], part = 0.f;
          const unsigned long long dbase = ((((unsigned long long)b * G + g) * h + hh) * Lq + i) * Kn;
#pragma unroll
          for (int ni = 0; ni < NKF; ++ni) {
            const int k = ni * 16 + x;
            pv[ni] = (i < Lq && k < Kn) ? slab[((long)i * gc + gl) * KP + k] : 0.f;
            if (a.drop.p > 0.f) dp[mi][ni][r] *= drop_mul(dkey, dbase + k, a.drop.p, dks);      // dP = mask/(1-p) * dP'
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
                if (a.dsc_bf16) dsc16[((long)i * h + hh) * TS_ + col] = (bf16_t)gq;
                else dsc[((long)i * h + hh) * TS_ + col] = gq;
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
      for (int mi = 0; mi < NKF; ++mi)                  // V tile consumed by dP: stage dV through it
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) vt[(mi * 16 + lg * 4 + r) * DK + ni * 16 + x] = (bf16_t)dv[mi][ni][r];
      for (int k = lane >> 3; k < Kn; k += 8) {
        const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
        *reinterpret_cast<uint4*>(dVb + row * a.lddv + (lane & 7) * 8) = *reinterpret_cast<const uint4*>(vt + k * DK + (lane & 7) * 8);
      }
    }
  }
}

// Backward of the stage-1 core on SAVED probabilities (the fused training forward's side output, St1Args.p_kp > 0): one WAVE per
// (clip, head, group), four independent waves per workgroup, no workgroup barrier and no f32 slab -- 16 / 9 / 6.5 KiB of LDS per wave
// (128 / 64 / 32 keys), so 8-16 waves share a CU where st1_mfma_kernel<.., true> keeps 4 (its 150 KiB of slab + images per workgroup made
// the T = 128 launch run at 0.8 TB/s: 476 us for 366 MB).
//   dP  = dO . V^T        A = dO rows from the wave's LDS tile, B = V rows STRAIGHT from global memory (16 bytes per lane)
//   P, mask               read from global in the accumulator layout (16 lanes = 64 contiguous bytes per query row)
//   dS  = P (dP' - sum_k P dP'), dP' = mask/(1-p) dP, 0 at masked keys  -> dscores, same addresses as st1_mfma_kernel
//   dV  = P'^T . dO       A = P'^T image [k][32 query rows] (pitch 80 B: written 8 bytes at a time from the accumulator layout, read as
//                         one 16-byte row piece per lane), B = dO^T by transposing reads; rows staged through the wave's LDS
template <int KSTEPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void st1_pbwd_kernel(const St1Args a) {
  constexpr int KPAD = 32 * KSTEPS, DK = 64, NKF = KPAD / 16, PP = 40;       // PP: pitch of the P'^T image in elements
  constexpr int WAVE_LDS = KPAD * DK * 2 > KPAD * PP * 2 + 32 * DK * 2 ? KPAD * DK * 2 : KPAD * PP * 2 + 32 * DK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T_ = a.T, S_ = a.S, Lq = a.Lq, h = a.h, dir = a.dir;
  const int G = dir == 0 ? S_ : T_, Kn = dir == 0 ? T_ : S_;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, x = lane & 15, lg = lane >> 4;
  const int g = blockIdx.x * 4 + w, hh = blockIdx.y, b = blockIdx.z;
  if (g >= G) return;                                            // (no workgroup barrier below)
  const long TS_ = (long)T_ * S_;
  const int d = h * DK;
  bf16_t* ptr = reinterpret_cast<bf16_t*>(smem + w * WAVE_LDS);  // [KPAD][PP]
  bf16_t* dt = ptr + KPAD * PP;                                  // [32][DK]
  bf16_t* stage = ptr;                                           // [KPAD][DK], after both are consumed
  // ---- loads: the dO tile into LDS, V fragments and probabilities into registers ----------------------------------------
  const bf16_t* dOb = a.dO + (((long)b * G + g) * Lq) * d + hh * DK;
  for (int i = lane >> 3; i < 32; i += 8) {
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (i < Lq) q = *reinterpret_cast<const uint4*>(dOb + (long)i * d + (lane & 7) * 8);
    *reinterpret_cast<uint4*>(dt + i * DK + (lane & 7) * 8) = q;
  }
  const bf16_t* Vb = a.V + (long)b * TS_ * a.ldv + hh * DK;
  uint4 vf[2][NKF];
#pragma unroll
  for (int ni = 0; ni < NKF; ++ni) {
    const int k = ni * 16 + x;
    const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      vf[ks][ni] = k < Kn ? *reinterpret_cast<const uint4*>(Vb + row * a.ldv + ks * 32 + lg * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
  const float* Pg = reinterpret_cast<const float*>(a.scores) + (((long)b * G + g) * h + hh) * Lq * a.p_kp;
  float pv[2][4][NKF];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = mi * 16 + lg * 4 + r;
#pragma unroll
      for (int ni = 0; ni < NKF; ++ni) {
        const int k = ni * 16 + x;
        pv[mi][r][ni] = (i < Lq && k < Kn) ? Pg[(long)i * a.p_kp + k] : 0.f;
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- dP = dO . V^T ------------------------------------------------------------------------------------------------------
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
    for (int ni = 0; ni < NKF; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) dp[mi][ni] = mfma_bf16(af[mi], vf[ks][ni], dp[mi][ni]);
  }
  // ---- dS, and the P'^T image ---------------------------------------------------------------------------------------------
  const unsigned long long dkey = a.drop.p > 0.f ? a.drop.key() : 0ULL;
  const float dks = a.drop.p > 0.f ? a.drop.keep_scale() : 1.f;
  const unsigned char* mk = a.tmask ? a.tmask + (long)b * Kn : nullptr;
  float* dsc = reinterpret_cast<float*>(a.dscores) + (long)b * Lq * h * TS_;
  bf16_t* dsc16 = reinterpret_cast<bf16_t*>(a.dscores) + (long)b * Lq * h * TS_;
  unsigned keym = 0u;                                            // bit ni: key ni * 16 + x is live (inside the range and not masked)
#pragma unroll
  for (int ni = 0; ni < NKF; ++ni) {
    const int k = ni * 16 + x;
    if (k < Kn && !(mk && mk[k] == 0)) keym |= 1u << ni;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    uint32_t pk[NKF][2];                                         // P' of (r = 0..3) as two bf16 pairs per key fragment
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = mi * 16 + lg * 4 + r;
      const unsigned long long dbase = ((((unsigned long long)b * G + g) * h + hh) * Lq + i) * Kn;
      float part = 0.f, pd[NKF];
#pragma unroll
      for (int ni = 0; ni < NKF; ++ni) {
        const int k = ni * 16 + x;
        const float mul = a.drop.p > 0.f ? drop_mul(dkey, dbase + k, a.drop.p, dks) : 1.f;
        dp[mi][ni][r] *= mul;                                    // dP = mask/(1-p) * dP'
        pd[ni] = pv[mi][r][ni] * mul;                            // P' = mask/(1-p) * P
        part += pv[mi][r][ni] * dp[mi][ni][r];
      }
      part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
#pragma unroll
      for (int ni = 0; ni < NKF; ++ni) {
        const int k = ni * 16 + x;
        if (i < Lq && k < Kn) {
          const float gq = ((keym >> ni) & 1u) ? pv[mi][r][ni] * (dp[mi][ni][r] - part) : 0.f;
          const long col = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
          if (a.dsc_bf16) dsc16[((long)i * h + hh) * TS_ + col] = (bf16_t)gq;
          else dsc[((long)i * h + hh) * TS_ + col] = gq;
        }
        const uint32_t hb = (uint32_t)__builtin_bit_cast(unsigned short, (bf16_t)pd[ni]);
        if (r & 1) pk[ni][r >> 1] |= hb << 16; else pk[ni][r >> 1] = hb;
      }
    }
#pragma unroll
    for (int ni = 0; ni < NKF; ++ni)                             // row k = ni * 16 + x, query rows mi * 16 + lg * 4 .. + 3: 8 bytes
      *reinterpret_cast<uint2*>(ptr + (ni * 16 + x) * PP + mi * 16 + lg * 4) = make_uint2(pk[ni][0], pk[ni][1]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- dV = P'^T . dO   (M = k, N = c, K = i) ---------------------------------------------------------------------------------
  f32x4 dv[NKF][4];
  {
    uint4 bfr[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bfr[ni] = frag_cols(dt, DK, ni * 16, 0, lane);
#pragma unroll
    for (int mi = 0; mi < NKF; ++mi) {
      const uint4 af = *reinterpret_cast<const uint4*>(ptr + (mi * 16 + x) * PP + lg * 8);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) dv[mi][ni] = mfma_bf16(af, bfr[ni], f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();                               // every lane has read its fragments: the images may be overwritten
#pragma unroll
  for (int mi = 0; mi < NKF; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) stage[(mi * 16 + lg * 4 + r) * DK + ni * 16 + x] = (bf16_t)dv[mi][ni][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  bf16_t* dVb = a.dV + (long)b * TS_ * a.lddv + hh * DK;
  for (int k = lane >> 3; k < Kn; k += 8) {
    const long row = dir == 0 ? (long)k * S_ + g : (long)g * S_ + k;
    *reinterpret_cast<uint4*>(dVb + row * a.lddv + (lane & 7) * 8) = *reinterpret_cast<const uint4*>(stage + k * DK + (lane & 7) * 8);
  }
}

template <int KSTEPS>
int launch_pbwd(const St1Args& a, int B, hipStream_t st) {
  constexpr int KPAD = 32 * KSTEPS;
  constexpr int WAVE_LDS = KPAD * 64 * 2 > KPAD * 40 * 2 + 32 * 64 * 2 ? KPAD * 64 * 2 : KPAD * 40 * 2 + 32 * 64 * 2;
  const int G = a.dir == 0 ? a.S : a.T;
  dim3 grid((unsigned)((G + 3) / 4), (unsigned)a.h, (unsigned)B);
  hipLaunchKernelGGL((st1_pbwd_kernel<KSTEPS>), grid, dim3(256), (size_t)4 * WAVE_LDS, st, a);
  bist_count_launch(BIST_K_ST1_PBWD);
  return launched("st1_pbwd_kernel");
}

template <typename TS, int KSTEPS, bool BWD>
int launch_one(const St1Args& a, int B, size_t lds, hipStream_t st) {
  BIST_LDS_OPTIN((&st1_mfma_kernel<TS, KSTEPS, BWD>), 160 * 1024, "bist_st_stage1_pv (matrix-core kernel)", -1);      // > 64 KiB of dynamic LDS
  const int G = a.dir == 0 ? a.S : a.T;
  dim3 grid((unsigned)((G + a.Gc - 1) / a.Gc), (unsigned)a.h, (unsigned)B);
  hipLaunchKernelGGL((st1_mfma_kernel<TS, KSTEPS, BWD>), grid, dim3(256), lds, st, a);
  bist_count_launch(BWD ? (a.p_kp > 0 ? BIST_K_ST1_PBWD : BIST_K_ST1_MFMA_BWD) : BIST_K_ST1_MFMA_FWD);
  return launched("st1_mfma_kernel");
}

// =====================================================================================================
// Stage 2 on the matrix cores (bf16, d = 512-class widths: d % 128 == 0, d <= 512, h <= 8, G <= 128).
// One workgroup per (clip b, query position i).  LDS images (bf16, 16-byte chunks XOR-swizzled by row & 7 so
// that both the row reads and the transposing reads are at most 2-way conflicted):
//   yimg [64 or 128 rows g][d]   the G stage-1 outputs of this position (read ONCE from HBM, rows >= G zero)
//   qg   [8 or 16 rows][d]       rows 0-7 folded query q2f, (backward) rows 8-15 dPY
// forward :  sc[hh,g] = q2f[hh,:].Y[g,:]  -> softmax over g (mask -> -1e9)  -> PY[hh,:] = sum_g P[hh,g] Y[g,:]
// backward:  sc, dP[hh,g] = dPY[hh,:].Y[g,:];  dS = P (dP - sum P dP);  dq2f[hh,:] = sum_g dS[hh,g] Y[g,:]
//            dY[g,:] = sum_hh dS[hh,g] q2f[hh,:] + P[hh,g] dPY[hh,:]      (one K = 16 product against qg)
// =====================================================================================================
__device__ __forceinline__ int swz_off(int row, int e, int pitch) {          // element offset of (row, e)
  return row * pitch + ((((e >> 3) ^ (row & 7)) << 3) | (e & 7));
}
__device__ __forceinline__ uint4 sfrag_rows(const bf16_t* img, int pitch, int row, int k0, int lane) {
  return *reinterpret_cast<const uint4*>(img + swz_off(row, k0 + (lane >> 4) * 8, pitch));
}
__device__ __forceinline__ uint4 sfrag_cols(const bf16_t* img, int pitch, int col0, int k0, int lane) {
  const int x = lane & 15, kg = lane >> 4, q = x >> 2, pp = x & 3;
  const int r0 = k0 + kg * 8 + q;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + swz_off(r0, col0 + 4 * pp, pitch)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + swz_off(r0 + 4, col0 + 4 * pp, pitch)));
  uint4 r;
  r.x = __builtin_bit_cast(uint2, lo).x; r.y = __builtin_bit_cast(uint2, lo).y;
  r.z = __builtin_bit_cast(uint2, hi).x; r.w = __builtin_bit_cast(uint2, hi).y;
  return r;
}

struct St2Args {
  const bf16_t* q2f; const bf16_t* Y; const unsigned char* gmask;
  bf16_t* PY;                                         // forward
  const bf16_t* dPY; bf16_t* dq2f; bf16_t* dY;        // backward
  int G, Lq, h, d;
  float* rowsum; const float* d_rowsum;               // sum_g P'[b,i,hh,g] (forward, nullable) and its gradient (backward, nullable)
  DropArg drop;
};

// zero the K rows >= 16 of a transposing fragment read (lanes with lane >> 4 >= 2): lets the [dS; P] and [q2f; dPY]
// images of the backward keep 16 rows instead of 32 zero-padded ones
__device__ __forceinline__ uint4 k16_only(uint4 v, int lane) {
  const unsigned m = (lane >> 4) < 2 ? 0xffffffffu : 0u;
  v.x &= m; v.y &= m; v.z &= m; v.w &= m;
  return v;
}

// HALF (forward, G <= 32: the s2t direction at T = 32): 32-row images -- 42 KiB of LDS instead of 74, THREE workgroups per CU instead of
// two (the B = 64 launch of 1 280 workgroups then takes two rounds of shorter workgroups instead of three), half the staging.
template <bool BWD, int GB, bool HALF = false>    // GB = 64-key blocks: G <= 64 GB (GB = 2: T = 128, BASELINE configs[3])
__global__ __launch_bounds__(256) void st2_mfma_kernel(const St2Args a) {
  static_assert(!HALF || (!BWD && GB == 1), "the 32-key form is a forward form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int GP = HALF ? 32 : 64 * GB;                          // padded key count = pitch of the score / probability images
  const int G = a.G, Lq = a.Lq, h = a.h, d = a.d;
  const int i = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int x = lane & 15, lg = lane >> 4;
  bf16_t* yimg = reinterpret_cast<bf16_t*>(smem);                  // [GP][d]
  constexpr int QROWS = BWD ? 16 : 8;                              // rows 0-7 q2f, (backward) 8-15 dPY
  bf16_t* qg = yimg + GP * d;                                      // [QROWS][d]
  float* scf = reinterpret_cast<float*>(qg + QROWS * d);           // [8][GP] scores -> probabilities
  float* dpf = scf + 8 * GP;                                       // [8][GP] dP -> dS          (backward)
  bf16_t* pimg = reinterpret_cast<bf16_t*>(dpf + (BWD ? 8 * GP : 0));   // [16][GP]: rows 0-7 A-operand of PY / dq2f (P or dS);
                                                                   //           backward: rows 0-7 dS, 8-15 P
  const int cpr = d >> 3;                                          // 16-byte chunks per row
  const long qoff = ((long)b * Lq + i) * h * d;
  const long ystride = (long)Lq * d;
  const bf16_t* Yb = a.Y + ((long)b * G * Lq + i) * d;
  // ---- stage the images ---------------------------------------------------------------------------------
  for (int idx = tid; idx < GP * cpr; idx += 256) {
    const int row = idx / cpr, c = idx - row * cpr;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < G) v = *reinterpret_cast<const uint4*>(Yb + row * ystride + c * 8);
    *reinterpret_cast<uint4*>(yimg + row * d + ((c ^ (row & 7)) << 3)) = v;
  }
  for (int idx = tid; idx < QROWS * cpr; idx += 256) {
    const int row = idx / cpr, c = idx - row * cpr;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < h) v = *reinterpret_cast<const uint4*>(a.q2f + qoff + (long)row * d + c * 8);
    else if (BWD && row >= 8 && row < 8 + h) v = *reinterpret_cast<const uint4*>(a.dPY + qoff + (long)(row - 8) * d + c * 8);
    *reinterpret_cast<uint4*>(qg + row * d + ((c ^ (row & 7)) << 3)) = v;
  }
  for (int idx = tid; idx < 16 * GP / 8; idx += 256) reinterpret_cast<uint4*>(pimg)[idx] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  // ---- scores (and dP): wave w owns keys g = 64 kb + 16w .. +15 of every key block kb --------------------------
#pragma unroll
  for (int kb = 0; kb < GB; ++kb) {
    if (HALF && w >= 2) break;                                     // (32 keys: waves 0 and 1 hold them all)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acd = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < d / 32; ++ks) {
      const uint4 bfr = sfrag_rows(yimg, d, 64 * kb + 16 * w + x, ks * 32, lane);
      acc = mfma_bf16(sfrag_rows(qg, d, x & 7, ks * 32, lane), bfr, acc);
      if constexpr (BWD) acd = mfma_bf16(sfrag_rows(qg, d, 8 + (x & 7), ks * 32, lane), bfr, acd);
    }
    const int g = 64 * kb + 16 * w + x;
    if (lg < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int hh = lg * 4 + r;
        float s = acc[r];
        if (g >= G) s = -INFINITY;
        else if (a.gmask && a.gmask[(long)b * G + g] == 0) s = MASK_FILL;
        scf[hh * GP + g] = s;
        if constexpr (BWD) dpf[hh * GP + g] = acd[r];
      }
    }
  }
  __syncthreads();
  // ---- softmax over g (wave w: heads 2w, 2w+1; lane = keys lane, lane + 64, ..) ------------------------------
  for (int hh = 2 * w; hh < 2 * w + 2; ++hh) {
    if (hh < h) {
      float s[GB], e[GB], m[GB];
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < GB; ++kb) { s[kb] = (64 * kb + lane < GP) ? scf[hh * GP + 64 * kb + lane] : -INFINITY; mx = fmaxf(mx, s[kb]); }
      mx = wave_max(mx);
      float den = 0.f;
#pragma unroll
      for (int kb = 0; kb < GB; ++kb) { e[kb] = (64 * kb + lane < G) ? expf(s[kb] - mx) : 0.f; den += e[kb]; }
      den = wave_sum(den);
      const float inv = 1.f / den;
      float rs = 0.f, dot = 0.f, dp[GB];
#pragma unroll
      for (int kb = 0; kb < GB; ++kb) {
        const int g = 64 * kb + lane;
        const float p = e[kb] * inv;
        e[kb] = p;
        if (g >= GP) continue;                                // (HALF: lanes 32-63 hold no key)
        scf[hh * GP + g] = p;
        // dropout of the probabilities (modules.py:62-63): P' = mask * P / (1-p) feeds the weighted sums
        m[kb] = 1.f;
        if (a.drop.p > 0.f && g < G)
          m[kb] = drop_mul(a.drop.key(), (((unsigned long long)b * Lq + i) * h + hh) * G + g, a.drop.p, a.drop.keep_scale());
        if constexpr (!BWD) {
          pimg[hh * GP + g] = (bf16_t)(p * m[kb]);
          rs += p * m[kb];
        } else {
          dp[kb] = g < G ? dpf[hh * GP + g] : 0.f;
          if (a.d_rowsum && g < G) dp[kb] += a.d_rowsum[((long)b * Lq + i) * h + hh];
          dp[kb] *= m[kb];                                    // dP = mask/(1-p) * dP'
          dot += p * dp[kb];
        }
      }
      if constexpr (!BWD) {
        if (a.rowsum) {
          rs = wave_sum(rs);
          if (lane == 0) a.rowsum[((long)b * Lq + i) * h + hh] = rs;
        }
      } else {
        dot = wave_sum(dot);
#pragma unroll
        for (int kb = 0; kb < GB; ++kb) {
          const int g = 64 * kb + lane;
          float ds = e[kb] * (dp[kb] - dot);
          if (g >= G || (a.gmask && a.gmask[(long)b * G + g] == 0)) ds = 0.f;
          pimg[hh * GP + g] = (bf16_t)ds;
          pimg[(8 + hh) * GP + g] = (bf16_t)(e[kb] * m[kb]);
        }
      }
    }
  }
  __syncthreads();
  // ---- PY (forward) / dq2f (backward): [8 heads] x [d]; wave w owns columns [w*d/4, (w+1)*d/4) -------------------
  {
    const int nf = d / 64;                           // 16-column fragments per wave
    bf16_t* outp = (BWD ? a.dq2f : a.PY) + qoff;
    for (int f = 0; f < nf; ++f) {
      const int col0 = (w * nf + f) * 16;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < GP / 32; ++ks)
        acc = mfma_bf16(frag_rows(pimg, GP, 0, ks * 32, lane & ~8), sfrag_cols(yimg, d, col0, ks * 32, lane), acc);
      if (lg < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int hh = lg * 4 + r;
          if (hh < h) outp[(long)hh * d + col0 + x] = (bf16_t)acc[r];
        }
      }
    }
  }
  if constexpr (BWD) {
    // ---- dY[g, :] = [dS; P]^T (K = 16 heads') x [q2f; dPY]: wave w owns columns as above, all key fragments.  Both
    //      operands are transposing reads of 16-row images: the K rows 16-31 of the 16x16x32 product are zeroed in
    //      registers (the read itself is clamped into the image) ----
    const int nf = d / 64;
    const int lane16 = lane & 31;                    // same (x, kg & 1): rows kg*8 + q stay below 16
    bf16_t* dYb = a.dY + ((long)b * G * Lq + i) * d;
    for (int f = 0; f < nf; ++f) {
      const int col0 = (w * nf + f) * 16;
      const uint4 bfr = k16_only(sfrag_cols(qg, d, col0, 0, lane16), lane);
#pragma unroll
      for (int mi = 0; mi < 4 * GB; ++mi) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = mfma_bf16(k16_only(frag_cols(pimg, GP, mi * 16, 0, lane16), lane), bfr, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = mi * 16 + lg * 4 + r;
          if (g < G) dYb[g * ystride + col0 + x] = (bf16_t)acc[r];
        }
      }
    }
  }
}

// =====================================================================================================
// Small multi-head attention BACKWARD on the matrix cores (bf16, dk = 64, Lq <= 32, Lk <= 64): one wave per
// (sequence n, head).  Q/K/V/dO tiles are staged in LDS as they lie in HBM ([rows][64]); every product is a
// handful of 16x16x32 MFMAs, the transposed operands (K, Q, dO as "B", dS/P as "A^T") come from
// ds_read_b64_tr_b16:
//   S = Q K^T, dP = dO V^T (+ dP_ext) -> softmax / softmax-backward in the accumulator layout
//   dQ = scale dS K,  dK = scale dS^T Q,  dV = P^T dO
// =====================================================================================================
// v or zeros, component-wise (a ternary on the uint4 struct turns into an address select through scratch memory)
__device__ __forceinline__ uint4 keep4(uint4 v, bool keep) {
  const unsigned m = keep ? 0xffffffffu : 0u;
  v.x &= m; v.y &= m; v.z &= m; v.w &= m;
  return v;
}

struct MhaBwdArgs {
  const bf16_t *Q, *K, *V, *dO; const unsigned char* mask; const float* dPext;
  bf16_t *dQ, *dK, *dV;
  int Lq, Lk, h;
  long ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, lddq, lddk, lddv, dq_bs, dk_bs, dv_bs, mask_bs, mask_qs;
  float scale;
  DropArg drop;
  int nc;          // 64-column chunks per head (dk = 64 nc): the wave walks them, S and dP accumulate across chunks
};

// Four waves per (sequence, head): the one-wave version of this kernel was instruction-bound (~7 k instructions, 20 us).
//   staging: 256 threads, one or two 16-byte rows pieces each;
//   S / dP:  wave w owns the 16 keys [16w, 16w+16)  -> f32 [32][64] images in LDS;
//   softmax and its backward: wave w owns 8 query rows, 8 lanes per row, 8 columns per lane (16-byte LDS traffic);
//   dQ: wave w owns 16 of the chunk's 64 columns;  dK, dV: wave w owns 16 keys;
//   outputs go back through the (dead) operand images and leave as 16-byte rows.
__global__ __launch_bounds__(256) void mha_bwd_mfma_kernel(const MhaBwdArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t qimg[32 * 64], kimg[64 * 64], vimg[64 * 64], gimg[32 * 64], pimg[32 * 64], simg[32 * 64];
  __shared__ __attribute__((aligned(16))) float sbuf[32 * 64], dbuf[32 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char mimg[32 * 64];      // mask bytes of this (n): 0 = masked
  const int hh = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int x = lane & 15, lg = lane >> 4;
  const int Lq = a.Lq, Lk = a.Lk, nc = a.nc;
  const long hoff = (long)hh * nc * 64;
  const bf16_t* Qh = a.Q + n * a.q_bs + hoff;
  const bf16_t* Kh = a.K + n * a.k_bs + hoff;
  const bf16_t* Vh = a.V + n * a.v_bs + hoff;
  const bf16_t* Gh = a.dO ? a.dO + n * a.o_bs + hoff : nullptr;
  const int sr = tid >> 3, sc = (tid & 7) * 8;          // staging: row sr (and sr + 32), columns sc .. sc+7
  auto stage = [&](int cc, bool with_mask, bool with_v) {
    // unconditional loads from clamped rows, zeroed afterwards (a conditional load would become a scratch pointer select)
    const int rq = min(sr, Lq - 1), rk0 = min(sr, Lk - 1), rk1 = min(sr + 32, Lk - 1);
    const uint4 q = *reinterpret_cast<const uint4*>(Qh + cc * 64 + (long)rq * a.ldq + sc);
    const uint4 g = *reinterpret_cast<const uint4*>((Gh ? Gh + cc * 64 : Qh) + (long)rq * (Gh ? a.ldo : a.ldq) + sc);
    const uint4 k0 = *reinterpret_cast<const uint4*>(Kh + cc * 64 + (long)rk0 * a.ldk + sc);
    const uint4 k1 = *reinterpret_cast<const uint4*>(Kh + cc * 64 + (long)rk1 * a.ldk + sc);
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
    if (with_v) {
      v0 = *reinterpret_cast<const uint4*>(Vh + cc * 64 + (long)rk0 * a.ldv + sc);
      v1 = *reinterpret_cast<const uint4*>(Vh + cc * 64 + (long)rk1 * a.ldv + sc);
    }
    unsigned mlo = 0x01010101u, mhi = 0x01010101u;
    if (a.mask && with_mask) {
      const unsigned char* mrow = a.mask + n * a.mask_bs + (long)rq * a.mask_qs;
      unsigned by[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) by[e] = (sc + e < Lk) ? (unsigned)mrow[sc + e] : 1u;
      mlo = by[0] | (by[1] << 8) | (by[2] << 16) | (by[3] << 24);
      mhi = by[4] | (by[5] << 8) | (by[6] << 16) | (by[7] << 24);
    }
    *reinterpret_cast<uint4*>(qimg + sr * 64 + sc) = keep4(q, sr < Lq);
    *reinterpret_cast<uint4*>(gimg + sr * 64 + sc) = keep4(g, sr < Lq && Gh != nullptr);
    *reinterpret_cast<uint4*>(kimg + sr * 64 + sc) = keep4(k0, sr < Lk);
    *reinterpret_cast<uint4*>(kimg + (sr + 32) * 64 + sc) = keep4(k1, sr + 32 < Lk);
    if (with_v) {
      *reinterpret_cast<uint4*>(vimg + sr * 64 + sc) = keep4(v0, sr < Lk);
      *reinterpret_cast<uint4*>(vimg + (sr + 32) * 64 + sc) = keep4(v1, sr + 32 < Lk);
    }
    if (with_mask) *reinterpret_cast<uint2*>(mimg + sr * 64 + sc) = make_uint2(mlo, mhi);
  };
  // ---- S = Q K^T and dP = dO V^T for this wave's 16 keys, summed over the head's 64-column chunks ----
  f32x4 S[2], D[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) { S[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; D[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int cc = 0; cc < nc; ++cc) {
    if (cc) __syncthreads();                              // the previous chunk's fragment reads are done
    stage(cc, cc == 0, true);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 bk = frag_rows(kimg, 64, w * 16, ks * 32, lane), bv = frag_rows(vimg, 64, w * 16, ks * 32, lane);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        S[mi] = mfma_bf16(frag_rows(qimg, 64, mi * 16, ks * 32, lane), bk, S[mi]);
        D[mi] = mfma_bf16(frag_rows(gimg, 64, mi * 16, ks * 32, lane), bv, D[mi]);
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = mi * 16 + lg * 4 + r, j = w * 16 + x;
      sbuf[i * 64 + j] = S[mi][r];
      dbuf[i * 64 + j] = D[mi][r];
    }
  __syncthreads();
  // ---- softmax and its backward: wave w owns rows 8w .. 8w+7, eight lanes per row, eight columns per lane ----
  {
    const int i = w * 8 + (lane >> 3), j0 = (lane & 7) * 8;
    const bool row_ok = i < Lq;
    const unsigned long long dkey = a.drop.p > 0.f ? a.drop.key() : 0ULL;
    const float dks = a.drop.p > 0.f ? a.drop.keep_scale() : 1.f;
    float sv[8], dv[8], dm[8];
    bool msk[8];
    {
      const float4 s0 = *reinterpret_cast<const float4*>(sbuf + i * 64 + j0), s1 = *reinterpret_cast<const float4*>(sbuf + i * 64 + j0 + 4);
      const float4 d0 = *reinterpret_cast<const float4*>(dbuf + i * 64 + j0), d1 = *reinterpret_cast<const float4*>(dbuf + i * 64 + j0 + 4);
      sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w; sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
      dv[0] = d0.x; dv[1] = d0.y; dv[2] = d0.z; dv[3] = d0.w; dv[4] = d1.x; dv[5] = d1.y; dv[6] = d1.z; dv[7] = d1.w;
    }
    const uint2 mb = *reinterpret_cast<const uint2*>(mimg + i * 64 + j0);
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = j0 + e;
      const unsigned byte = ((e < 4 ? mb.x : mb.y) >> (8 * (e & 3))) & 0xffu;
      msk[e] = a.mask && row_ok && j < Lk && byte == 0;
      float sc2 = sv[e] * a.scale;
      if (msk[e]) sc2 = MASK_FILL;
      if (j >= Lk) sc2 = -INFINITY;
      sv[e] = sc2;
      dm[e] = 1.f;
      if (a.drop.p > 0.f && row_ok && j < Lk) {           // dP = mask/(1-p) * dP'   (modules.py:62-63)
        dm[e] = drop_mul(dkey, (((unsigned long long)n * a.h + hh) * Lq + i) * Lk + j, a.drop.p, dks);
        dv[e] *= dm[e];
      }
      mx = fmaxf(mx, sc2);
    }
    if (a.dPext && row_ok) {
      const float* pe = a.dPext + (((long)n * a.h + hh) * Lq + i) * Lk;
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[e] += pe[min(j0 + e, Lk - 1)] * (j0 + e < Lk ? 1.f : 0.f);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    float den = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sv[e] = (j0 + e < Lk) ? expf(sv[e] - mx) : 0.f; den += sv[e]; }
    den += __shfl_xor(den, 1, 64); den += __shfl_xor(den, 2, 64); den += __shfl_xor(den, 4, 64);
    const float inv = 1.f / den;
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sv[e] *= inv; dot += sv[e] * dv[e]; }
    dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64);
    bf16_t pe8[8], se8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float ds = sv[e] * (dv[e] - dot) * a.scale;
      if (msk[e] || !row_ok || j0 + e >= Lk) ds = 0.f;
      pe8[e] = (bf16_t)(row_ok ? sv[e] * dm[e] : 0.f);        // P' = mask * P / (1-p) feeds dV
      se8[e] = (bf16_t)ds;
    }
    *reinterpret_cast<uint4*>(pimg + i * 64 + j0) = *reinterpret_cast<const uint4*>(pe8);
    *reinterpret_cast<uint4*>(simg + i * 64 + j0) = *reinterpret_cast<const uint4*>(se8);
  }
  __syncthreads();
  // ---- dQ = dS K (wave w: 16 of the chunk's columns); dK = dS^T Q, dV = P^T dO (wave w: 16 keys), chunk by chunk ----
  for (int cc = 0; cc < nc; ++cc) {
    if (nc > 1) { stage(cc, false, false); __syncthreads(); }          // one chunk: the tiles of the first phase are still in place
    f32x4 dq[2], dkk[4], dvv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { if (t < 2) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dkk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dvv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 bk = frag_cols(kimg, 64, w * 16, ks * 32, lane);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) dq[mi] = mfma_bf16(frag_rows(simg, 64, mi * 16, ks * 32, lane), bk, dq[mi]);
    }
    {
      const uint4 as = frag_cols(simg, 64, w * 16, 0, lane), ap = frag_cols(pimg, 64, w * 16, 0, lane);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        dkk[ni] = mfma_bf16(as, frag_cols(qimg, 64, ni * 16, 0, lane), dkk[ni]);
        dvv[ni] = mfma_bf16(ap, frag_cols(gimg, 64, ni * 16, 0, lane), dvv[ni]);
      }
    }
    __syncthreads();                                      // every wave is done reading the operand images
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) qimg[(mi * 16 + lg * 4 + r) * 64 + w * 16 + x] = (bf16_t)dq[mi][r];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        kimg[(w * 16 + lg * 4 + r) * 64 + ni * 16 + x] = (bf16_t)dkk[ni][r];
        vimg[(w * 16 + lg * 4 + r) * 64 + ni * 16 + x] = (bf16_t)dvv[ni][r];
      }
    }
    __syncthreads();
    bf16_t* dQn = a.dQ + n * a.dq_bs + hoff + cc * 64;
    bf16_t* dKn = a.dK + n * a.dk_bs + hoff + cc * 64;
    bf16_t* dVn = a.dV + n * a.dv_bs + hoff + cc * 64;
    if (sr < Lq) *reinterpret_cast<uint4*>(dQn + (long)sr * a.lddq + sc) = *reinterpret_cast<const uint4*>(qimg + sr * 64 + sc);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = sr + 32 * t;
      if (r < Lk) {
        *reinterpret_cast<uint4*>(dKn + (long)r * a.lddk + sc) = *reinterpret_cast<const uint4*>(kimg + r * 64 + sc);
        *reinterpret_cast<uint4*>(dVn + (long)r * a.lddv + sc) = *reinterpret_cast<const uint4*>(vimg + r * 64 + sc);
      }
    }
    if (cc + 1 < nc) __syncthreads();                     // the output rows have left before the next chunk is staged
  }
}

}  // namespace

// returns 1 if launched, 0 if outside the envelope, -1 on launch error
int bist_mha_bwd_mfma(const void* Q, const void* K, const void* V, const unsigned char* mask, const void* dO, const float* dPext,
                      void* dQ, void* dK, void* dV, int N, int Lq, int Lk, int h, int dk, long ldq, long ldk, long ldv, long ldo,
                      long q_bs, long k_bs, long v_bs, long o_bs, long lddq, long lddk, long lddv, long dq_bs, long dk_bs, long dv_bs,
                      long mask_bs, long mask_qs, float scale, const DropArg& drop, hipStream_t st) {
  if (dk % 64 != 0 || dk > 512 || Lq > 32 || Lk > 64) return 0;
  auto al8 = [](long v) { return (v % 8) == 0; };
  if (!(al8(ldq) && al8(ldk) && al8(ldv) && al8(lddq) && al8(lddk) && al8(lddv) && al8(q_bs) && al8(k_bs) && al8(v_bs) && al8(dq_bs) &&
        al8(dk_bs) && al8(dv_bs) && (!dO || (al8(ldo) && al8(o_bs))))) return 0;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV | (uintptr_t)dO) % 16) return 0;
  MhaBwdArgs a{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO, mask, dPext, (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV,
               Lq, Lk, h, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, lddq, lddk, lddv, dq_bs, dk_bs, dv_bs, mask_bs, mask_qs, scale, drop, dk / 64};
  hipLaunchKernelGGL(mha_bwd_mfma_kernel, dim3((unsigned)h, (unsigned)N), dim3(256), 0, st, a);
  bist_count_launch(BIST_K_MHA_BWD_MFMA);
  return launched("mha_bwd_mfma_kernel");
}

// returns 1 if launched, 0 if outside the envelope, -1 on launch error
int bist_st2_mfma(const void* q2f, const void* Y, const unsigned char* gmask, void* PY, const void* dPY, void* dq2f, void* dY,
                  float* rowsum, const float* d_rowsum, int B, int G, int Lq, int h, int d, int bwd, const DropArg& drop, hipStream_t st) {
  if (G > 128 || h > 8 || d > 512 || (d % 128) != 0) return 0;
  if (((uintptr_t)q2f | (uintptr_t)Y | (uintptr_t)(bwd ? dPY : q2f)) % 16) return 0;
  static const int no_half = [] { const char* e = getenv("BIST_ST2_NO_HALF"); return e ? atoi(e) : 0; }();      // tuning aid
  const bool half = !bwd && G <= 32 && !no_half;
  const int gb = G > 64 ? 2 : 1, gp = half ? 32 : 64 * gb;
  const size_t lds = (size_t)gp * d * 2 + (size_t)(bwd ? 16 : 8) * d * 2 + (size_t)(bwd ? 2 : 1) * 8 * gp * 4 + 16 * gp * 2;
  St2Args a{(const bf16_t*)q2f, (const bf16_t*)Y, gmask, (bf16_t*)PY, (const bf16_t*)dPY, (bf16_t*)dq2f, (bf16_t*)dY, G, Lq, h, d,
            rowsum, d_rowsum, drop};
  dim3 grid((unsigned)Lq, (unsigned)B);
#define ST2_GO(BWD_, GB_)                                                                                                  \
  do {                                                                                                                     \
    BIST_LDS_OPTIN((&st2_mfma_kernel<BWD_, GB_>), 160 * 1024, "bist_st_stage2 (matrix-core kernel)", -1);                  \
    hipLaunchKernelGGL((st2_mfma_kernel<BWD_, GB_>), grid, dim3(256), lds, st, a);                                         \
  } while (0)
  if (bwd) { if (gb == 2) ST2_GO(true, 2); else ST2_GO(true, 1); }
  else if (half) {
    BIST_LDS_OPTIN((&st2_mfma_kernel<false, 1, true>), 160 * 1024, "bist_st_stage2 (matrix-core kernel)", -1);
    hipLaunchKernelGGL((st2_mfma_kernel<false, 1, true>), grid, dim3(256), lds, st, a);
  } else { if (gb == 2) ST2_GO(false, 2); else ST2_GO(false, 1); }
#undef ST2_GO
  bist_count_launch(bwd ? BIST_K_ST2_MFMA_BWD : BIST_K_ST2_MFMA_FWD);
  return launched("st2_mfma_kernel");
}

// returns 1 if launched, 0 if the shape is outside this kernel's envelope (caller falls back), -1 on launch error
int bist_st1_mfma(const void* scores, int sc_is_f32, const void* V, const unsigned char* tmask, void* O, const void* dO,
                  void* dscores, int dsc_bf16, void* dV, int B, int T, int S, int Lq, int h, int dk, long ldv, long lddv, int dir,
                  int bwd, const DropArg& drop, hipStream_t st, int p_kp) {
  const int G = dir == 0 ? S : T, Kn = dir == 0 ? T : S;
  if (dk != 64 || Lq > 32 || Kn > 128 || (ldv % 8) != 0 || ((uintptr_t)V % 16) != 0) return 0;
  if (bwd && ((lddv % 8) != 0 || ((long)h * dk) % 8 != 0 || !sc_is_f32)) return 0;
  const int ksteps = Kn <= 32 ? 1 : (Kn <= 64 ? 2 : 4);
  const int kpad = 32 * ksteps;
  // per group: slab rows + P image + V tile (+ dO tile); one group per wave, at most 4
  const long per_g = (long)Lq * (Kn + 1) * 4 + 32L * kpad * 2 + (long)kpad * 64 * 2 + (bwd ? 32L * 64 * 2 : 0);
  int Gc = (int)((150 * 1024 - 32) / per_g);
  if (Gc < 1) return 0;
  if (Gc > 4) Gc = 4;
  static const int force_gc = [] { const char* e = getenv("BIST_ST1_GC"); return e ? atoi(e) : 0; }();      // tuning aid
  if (kpad >= 128 && Gc > 2) Gc = 2;        // 128 keys: two groups per workgroup keep two workgroups on a CU (146 vs 168 us at T = 128)
  if (force_gc >= 1 && force_gc < Gc) Gc = force_gc;
  if (Gc > G) Gc = G;
  const size_t lds = (size_t)(((long)Lq * Gc * (Kn + 1) * 4 + 15) / 16 * 16) + (size_t)Gc * 32 * kpad * 2 + (size_t)Gc * kpad * 64 * 2 +
                     (bwd ? (size_t)Gc * 32 * 64 * 2 : 0) + 16;
  static const int dbg = [] { const char* e = getenv("BIST_ST1_DBG"); return e ? atoi(e) : 0; }();
  if (p_kp > 0 && (!bwd || !sc_is_f32 || p_kp < Kn)) return 0;
  static const int old_pbwd = [] { const char* e = getenv("BIST_ST1_PBWD_SLAB"); return e ? atoi(e) : 0; }();      // tuning aid: 1 = the slab kernel on saved probabilities
  if (p_kp > 0 && !old_pbwd) {
    St1Args a{scores, (const bf16_t*)V, tmask, (bf16_t*)O, (const bf16_t*)dO, dscores, (bf16_t*)dV, T, S, Lq, h, ldv, lddv, dir, 4, dsc_bf16, p_kp, dbg, drop};
    return ksteps == 1 ? launch_pbwd<1>(a, B, st) : (ksteps == 2 ? launch_pbwd<2>(a, B, st) : launch_pbwd<4>(a, B, st));
  }
  St1Args a{scores, (const bf16_t*)V, tmask, (bf16_t*)O, (const bf16_t*)dO, dscores, (bf16_t*)dV, T, S, Lq, h, ldv, lddv, dir, Gc, dsc_bf16, p_kp, dbg, drop};
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

