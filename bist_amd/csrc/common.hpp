// Shared device/host helpers for libbist_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/bist_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- host-side error plumbing (defined in api.hip) -------------------------------------------
void bist_set_error(const char* fmt, ...);
void bist_count_launch(int family);       // BIST_K_* launch counters (api.hip)
#define BIST_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      bist_set_error(__VA_ARGS__);         \
      return BIST_EINVAL;                  \
    }                                      \
  } while (0)
#define BIST_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      bist_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return BIST_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

// Per-device, once-per-kernel opt-in to more than 64 KiB of dynamic LDS.  `static` at the expansion site: one flag word per kernel
// instantiation, one bit per device ordinal (a process that drives several GPUs sets the attribute on each).  A refusal surfaces
// HERE as an error return instead of as a launch failure far from its cause.
#define BIST_LDS_OPTIN(kernel_, bytes_, name_, failret_)                                                                     \
  do {                                                                                                                       \
    static unsigned long long done__ = 0ULL;                                                                                 \
    int dev__ = 0;                                                                                                           \
    if (hipGetDevice(&dev__) != hipSuccess) dev__ = 0;                                                                       \
    if (!(dev__ >= 0 && dev__ < 64 && ((done__ >> dev__) & 1ULL))) {                                                        \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes_)) != hipSuccess) { \
        (void)hipGetLastError();                                                                                             \
        bist_set_error("%s: cannot reserve %d bytes of dynamic LDS on device %d", name_, (int)(bytes_), dev__);               \
        return failret_;                                                                                                     \
      }                                                                                                                      \
      if (dev__ >= 0 && dev__ < 64) done__ |= 1ULL << dev__;                                                                 \
    }                                                                                                                        \
  } while (0)

// Development hooks (api.hip): in-kernel stamp buffers are handed over by an explicit call (bist_dev_set_stamps), never parsed
// from the environment; the ablation bits are read from the environment ONCE.
unsigned long long* bist_dev_stamps(int which);      // which: 0 = st1_fused, 1 = decstack; null unless set
int bist_dev_dbg(int which);

// MFMA implementation of the stage-1 core (attention_mfma.hip): 1 = launched, 0 = shape outside its envelope, -1 = error
struct DropArg;
int bist_st1_mfma(const void* scores, int sc_is_f32, const void* V, const unsigned char* tmask, void* O, const void* dO,
                  void* dscores, int dsc_bf16, void* dV, int B, int T, int S, int Lq, int h, int dk, long ldv, long lddv, int dir,
                  int bwd, const DropArg& drop, hipStream_t st, int p_kp = 0);

int bist_st2_mfma(const void* q2f, const void* Y, const unsigned char* gmask, void* PY, const void* dPY, void* dq2f, void* dY,
                  float* rowsum, const float* d_rowsum, int B, int G, int Lq, int h, int d, int bwd, const DropArg& drop, hipStream_t st);

int bist_mha_bwd_mfma(const void* Q, const void* K, const void* V, const unsigned char* mask, const void* dO, const float* dPext,
                      void* dQ, void* dK, void* dV, int N, int Lq, int Lk, int h, int dk, long ldq, long ldk, long ldv, long ldo,
                      long q_bs, long k_bs, long v_bs, long o_bs, long lddq, long lddk, long lddv, long dq_bs, long dk_bs, long dv_bs,
                      long mask_bs, long mask_qs, float scale, const DropArg& drop, hipStream_t st);

// ---- element conversion ----------------------------------------------------------------------
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }  // v_cvt_pk_bf16_f32, RNE

// ---- wave-level reductions (64 lanes) ---------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based dropout mask ----------------------------------------------------------------
// keep(idx) is a pure function of (seed, idx), so the backward pass regenerates the same mask.
// One 64-bit hash serves the 4 consecutive elements 4*idx4 .. 4*idx4+3 (16 bits each; the drop probability is
// floor(p * 65536) / 65536): kernels that walk 4 or 8 consecutive elements per lane hash once per group.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
__device__ __forceinline__ uint32_t drop_threshold(float p) { return (uint32_t)(p * 65536.f); }
__device__ __forceinline__ uint64_t drop_bits4(uint64_t seed, uint64_t idx4) { return mix64(seed + idx4 * 0x9E3779B97F4A7C15ULL); }
__device__ __forceinline__ bool drop_keep_of(uint64_t bits4, int e, uint32_t thr) { return ((uint32_t)(bits4 >> (e * 16)) & 0xffffu) >= thr; }
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, float p) {
  return drop_keep_of(drop_bits4(seed, idx >> 2), (int)(idx & 3), drop_threshold(p));
}
// Dropout of attention probabilities (reference modules.py:62-63) as the kernels carry it: p == 0 means off.
// The mask of probability element `idx` (its linear index in the canonical [.., query, key] order documented at
// each entry point) is drop_keep(seed + step counter, idx), the same in forward and backward.
struct DropArg {
  float p; unsigned long long seed; const unsigned long long* ctr;
  __device__ __forceinline__ unsigned long long key() const { return seed + (ctr ? ctr[0] * 0xD1B54A32D192ED03ULL : 0ULL); }
  __device__ __forceinline__ float keep_scale() const { return 1.f / (1.f - p); }
};
__device__ __forceinline__ float drop_mul(unsigned long long key, unsigned long long idx, float p, float keep_scale) {
  return drop_keep(key, idx, p) ? keep_scale : 0.f;
}
inline DropArg make_drop(const BistDrop* d) {
  DropArg r{0.f, 0ULL, nullptr};
  if (d && d->p > 0.f) { r.p = d->p; r.seed = d->seed; r.ctr = (const unsigned long long*)d->ctr; }
  return r;
}
