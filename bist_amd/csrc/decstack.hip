// The response-decoder stack of one beam-search step as ONE persistent launch (bf16, d = 512, h = 8, <= 64 rows).
//
// Reference semantics: MultimodalDecoderLayer12.forward (model/decoder.py:20-60) with enc_vc_combine != 'none' -- causal self-attention,
// attention to the dialogue history, to the query, to the fused modalities (each x + W_o MHA(LN(x), mem, mem, mask) + b_o,
// modules.py:42-44, 54-64, 81-100), feed-forward (modules.py:112-113) -- for every layer of MultimodalDecoder8's loop
// (decoder.py:114-182) on the R = hypotheses x prefix-length rows of a decode step (decode.py:62-66).  In eval mode the key/value
// projections of the three memories do not depend on the prefix: they arrive precomputed per turn (K [LkP][512], V^T [512][LkP]).
//
// One step of the unfused path is ~120 launches of 5-17 us on these <= 60 rows (pure latency).  Here 32 workgroups of 256 threads
// (one per CU, all resident) walk the 10 phases of each layer and meet at a grid barrier after each:
//     A(s): LN_s(x) for all rows into an LDS image (every workgroup, redundantly), then its 16-column tiles of the projection
//           (QKV: 3 tiles of 1536, cross query: 1 of 512, FFN hidden: 4 of 2048), K split over the 4 waves, partial tiles summed in LDS
//     B(s): the attention core of ALL heads and rows on MFMA (every workgroup, redundantly: 2 heads per wave, scores, masked softmax
//           and P.V in registers as in st1_fused.hip) into the LDS image, then its 16 columns of the output projection + residual;
//           FFN: its 16 columns of W_2 over the 2048-wide hidden rows
// The weight fragments of the NEXT phase are fetched into registers before the barrier, so the HBM / L2 latency of the 12.6 MB of
// weights per layer hides behind the barrier and the previous phase's tail.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int D = 512, H = 8, NWG = 32, NT = 256;
constexpr float MASK_FILL = -1e9f;

struct DecLayerDev {            // one per layer, in device memory (all pointers device pointers)
  const bf16_t* ln_a[5]; const bf16_t* ln_b[5];
  const bf16_t* Wqkv; const bf16_t* bqkv;            // packed [1536][512] / [1536] of the self-attention
  const bf16_t* Wq[3]; const bf16_t* bq[3];          // query projections of the three cross-attentions
  const bf16_t* Wo[4]; const bf16_t* bo[4];          // output projections: self, history, query, fused modalities
  const bf16_t* Kc[3]; const bf16_t* VTc[3];         // per turn: K [LkP][512], V^T [512][LkP] of the three memories
  const unsigned char* cmask[3];                     // key masks [LkP] (1 = attend)
  const bf16_t* W1; const bf16_t* b1; const bf16_t* W2; const bf16_t* b2;
  int Lk[3]; int LkP[3];
};

struct DecArgs {
  const DecLayerDev* layers; int nl;
  const bf16_t* x_in;            // [R][512]
  bf16_t* xbuf[2];               // residual stream, ping-pong [RP][512]
  bf16_t* qbuf;                  // [RP][512]
  bf16_t* kcache; bf16_t* vcache; // self-attention keys / values of the rows seen so far, per layer [64 slots][512] (layer l at + l * 64 * 512):
                                 // this call's row r is slot slot0 + r; self_mask [R][LkS] names the slots a row attends
  int slot0;
  bf16_t* hbuf;                  // [RP][2048]
  const unsigned char* smask;    // [R][LkS] self-attention mask (1 = attend)
  int R, LkS;
  unsigned* sync;                // 8 words, zero before the first call: [0] barrier arrivals, [1] exits (both zero again when a call ends),
                                 // [4] STICKY error flag (a barrier timed out)
  int dbg;                       // BIST_DECSTACK_DBG (development): 1 = agent-scope acquire after every barrier
  int spread;                    // BIST_DECSTACK_XCDS: the 32 workers on 1 (default), 2 or 4 XCDs (8 * spread consecutive blocks hold `spread` workers)
  unsigned long long* stamps;    // BIST_DECSTACK_STAMPS (development): s_memtime stamps of workgroup 0 in layer 0, or null
  float* pbuf;                   // head-local form: per-head partial output projections, f32 [2][8 heads][16 rows][512]
};

__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_max(float v) {      // over lanes x, x+16, x+32, x+48
  float p = v, q = v; swap16(p, q); v = fmaxf(p, q);
  p = v; q = v; swap32(p, q); return fmaxf(p, q);
}
__device__ __forceinline__ float rows_sum(float v) {
  float p = v, q = v; swap16(p, q); v = p + q;
  p = v; q = v; swap32(p, q); return p + q;
}

// Hand-off protocol (MI355X guide, G16 recipe R1): every byte another workgroup reads later in the launch is stored WRITE-THROUGH
// (st8: an 8-byte sc1 store), so the barrier needs no L2 write-back: every storing wave drains its stores, the workgroup meets, lane 0
// arrives on the monotonic counter (agent-scope atomic) and polls it relaxed with a BOUNDED spin, and the workgroup meets again before
// anyone loads.  Every load of handed-off bytes is an sc1 load to registers (ld16 / ld8: served by L2, never by this CU's L1), which
// is the guide's measured form that needs no acquire fence (valid-forms table, row "one lane of each storing workgroup adds"):
// hipMalloc memory, one workgroup per CU, 8-byte sc1 stores, 8- / 16-byte sc1 loads.  Weights, biases, masks and the per-turn
// key / value caches are never written inside the launch and use plain loads.
typedef __attribute__((address_space(1))) unsigned long long gu64;       // shared words and payload: GLOBAL accesses, never flat
typedef __attribute__((address_space(1))) unsigned gu32;
// wt = false (all 32 workgroups verified to sit on ONE XCD, see the kernel): a plain store -- the XCD's L2 is the common point of its
// CUs, the line stays there for the readers' sc1 (L1-bypassing) loads instead of going out to memory and coming back
__device__ __forceinline__ void st8(void* p, uint2 v, bool wt) {
  if (wt) __hip_atomic_store((gu64*)p, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *(gu64*)p = ((unsigned long long)v.y << 32) | v.x;
}
// sc1 loads (L2-served, never from this CU's L1): EVERY load of handed-off bytes is one of these, so no acquire fence either
__device__ __forceinline__ uint4 ld16(const void* base, long byte_off) {          // base wave-uniform
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 16);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld8(const void* p) {
  const unsigned long long v = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void grid_barrier(unsigned* sync_, unsigned target, int dbg = 0) {
  gu32* sync = (gu32*)sync_;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 20)) {                      // ~1 s: a workgroup is missing -- give up (every workgroup will), flag it
        __hip_atomic_store(sync + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // no instruction: keeps the loads below the poll
    if (dbg & 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  __syncthreads();
}

// swizzled [rows][1024 B] image: 16-byte chunk c of row r at chunk c ^ (r & 15)
__device__ __forceinline__ char* img_at(char* img, int row, int chunk) { return img + row * 1024 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ uint4 img_frag(const char* img, int mt, int ks, int x, int kg) {       // rows 16*mt + x, channels 32*ks + 8*kg ..
  return *reinterpret_cast<const uint4*>(img + (mt * 16 + x) * 1024 + (((4 * ks + kg) ^ x) << 4));
}

// LN_s(x) of all rows into the image (rows >= R zero): the RP padded rows are dealt evenly to the four waves (RP / 4 each), sixteen lanes
// per row (32 channels each), four rows per pass; all row loads of a lane are in flight together
__device__ __forceinline__ void layernorm_to_image(const bf16_t* x, const bf16_t* ga, const bf16_t* gb, char* img, int R, int RP, int w, int lane) {
  const int sub = lane & 15, rq = lane >> 4, MTR = RP >> 4, row0 = w * (RP >> 2);
  uint4 q[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (p >= MTR) break;
    const int row = row0 + 4 * p + rq;
#pragma unroll
    for (int c = 0; c < 4; ++c) q[p][c] = ld16(x, (long)min(row, R - 1) * (D * 2) + (sub * 4 + c) * 16);
  }
  uint4 qa[4], qb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { qa[c] = reinterpret_cast<const uint4*>(ga)[sub * 4 + c]; qb[c] = reinterpret_cast<const uint4*>(gb)[sub * 4 + c]; }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (p >= MTR) break;
    const int row = row0 + 4 * p + rq;
    const bool live = row < R;
    float v[32];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t u[4] = {q[p][c].x, q[p][c].y, q[p][c].z, q[p][c].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[c * 8 + 2 * e] = bf_lo(u[e]); v[c * 8 + 2 * e + 1] = bf_hi(u[e]); sum += v[c * 8 + 2 * e] + v[c * 8 + 2 * e + 1]; }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) { const float dlt = v[e] - mean; ss += dlt * dlt; }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.f / (sqrtf(ss * (1.f / (D - 1))) + 1e-6f);          // unbiased std, eps outside the root (modules.py:28-31)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t ua[4] = {qa[c].x, qa[c].y, qa[c].z, qa[c].w}, ub[4] = {qb[c].x, qb[c].y, qb[c].z, qb[c].w};
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y0 = live ? bf_lo(ua[e]) * (v[c * 8 + 2 * e] - mean) * inv + bf_lo(ub[e]) : 0.f;
        const float y1 = live ? bf_hi(ua[e]) * (v[c * 8 + 2 * e + 1] - mean) * inv + bf_hi(ub[e]) : 0.f;
        o[e] = pack2(y0, y1);
      }
      *reinterpret_cast<uint4*>(img_at(img, row, sub * 4 + c)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// weight fragments of NTILES 16-column tiles starting at col0, this wave's KS k-steps (ks = w + 4*i): lane (x, kg) holds
// W[col0 + 16*t + x][32*ks + 8*kg .. +7]  (nn.Linear layout [out][in], row length ldw)
template <int NTILES, int KS>
__device__ __forceinline__ void load_w(uint4 (&wr)[16], const bf16_t* W, int ldw, int col0, int w, int x, int kg) {
#pragma unroll
  for (int t = 0; t < NTILES; ++t)
#pragma unroll
    for (int i = 0; i < KS; ++i)
      wr[t * KS + i] = *reinterpret_cast<const uint4*>(W + (long)(col0 + 16 * t + x) * ldw + 32 * (w + 4 * i) + 8 * kg);
}

// partial products of this wave: acc[t][mt][r] = sum over its k-steps of W[col 4lg+r of tile t][k] * A[row x of tile mt][k]
template <int NTILES, int KS, class AF>
__device__ __forceinline__ void partial_product(f32x4 (&acc)[4][4], const uint4 (&wr)[16], AF afrag, int MTR, int w) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KS; ++i) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt >= MTR) break;
      const uint4 a = afrag(mt, w + 4 * i);
#pragma unroll
      for (int t = 0; t < NTILES; ++t) acc[t][mt] = mfma16(wr[t * KS + i], a, acc[t][mt]);
    }
  }
}

// the same for ONE tile with the A fragments read from global rows: requested in batches of four k-steps, the next batch in flight
// while the current one is multiplied (<= 32 loads per lane outstanding)
template <int KS, bool DB, class AF>
__device__ __forceinline__ void partial_product_glob(f32x4 (&acc)[4][4], const uint4 (&wr)[16], AF afrag, int MTR, int w) {
  constexpr int NB = KS / 4;
  uint4 af[2][4][4];
  auto request = [&](int b) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (mt < MTR) af[b & 1][i][mt] = afrag(mt, w + 4 * (4 * b + i));
  };
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (DB) request(0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (!DB) request(b);
    else if (b + 1 < NB) request(b + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (mt < MTR) acc[0][mt] = mfma16(wr[4 * b + i], af[b & 1][i][mt], acc[0][mt]);
  }
}

// sum the four waves' partial tiles through LDS and hand every (tile, row tile) to `epi(t, mt, v)`: lane (x, lg) gets the 4 consecutive
// columns 4*lg .. 4*lg+3 of row x
template <int NTILES, class EPI>
__device__ __forceinline__ void reduce_tiles(f32x4 (&acc)[4][4], char* part, int MTR, int w, int lane, EPI epi) {
  f32x4* pw = reinterpret_cast<f32x4*>(part);
#pragma unroll
  for (int t = 0; t < NTILES; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (mt < MTR) pw[((w * 16) + t * 4 + mt) * 64 + lane] = acc[t][mt];
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NTILES; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt >= MTR || ((t * 4 + mt) & 3) != w) continue;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const f32x4 u = pw[(o * 16 + t * 4 + mt) * 64 + lane];
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
      }
      epi(t, mt, v);
    }
  __syncthreads();
}

// attention core of ONE (head, 16-row tile) unit, written to the context rows in global memory (bf16 [rows][512], write-through):
// Q [RP][512], K [LkP][512] rows, V^T [512][LkP], mask[row * mrs + key] (mrs = 0: per key only; rows padded to LkP bytes, 4-byte
// aligned), Lk valid keys.  The four waves each compute the scores and the masked softmax of the unit (16 rows x <= 64 keys, in
// registers) and ONE 16-channel tile of P.V (wave w: channels 64*head + 16*w ..).  All loads of the unit are requested together.
// VROW: the values lie row-major [key][512] like the keys (the self-attention cache: rows arrive one decode step at a time) and are
// fetched as 4-byte words (two channels of one key); otherwise transposed V^T [512][LkP] (the per-turn memories), 8-byte loads.
template <bool VROW>
__device__ __forceinline__ void core_unit(const bf16_t* Q, const bf16_t* K, const bf16_t* VT, const unsigned char* mask, int mrs, int Lk, int LkP,
                                          bf16_t* ctx, int R, int head, int mt, int w, int x, int kg, bool wt) {
  const int KT = LkP >> 4;                                  // key tiles (2 or 4)
  const int row = 16 * mt + x, rowc = min(row, R - 1);
  uint4 qf[2], kf[2][4];
  uint32_t mk[4];
  uint2 vv[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = ld16(Q, ((long)rowc * D + head * 64 + 32 * ks + 8 * kg) * 2);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (kt < KT) kf[ks][kt] = ld16(K, ((long)(16 * kt + x) * D + head * 64 + 32 * ks + 8 * kg) * 2);
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
    if (kt < KT) mk[kt] = *reinterpret_cast<const uint32_t*>(mask + (long)rowc * mrs + 16 * kt + 4 * kg);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (2 * j >= KT) break;
    if constexpr (VROW) {
      const int ch = head * 64 + 16 * w + x;
      unsigned e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                          // K slot u: key 32j + 16 (u / 4) + 4 kg + u % 4
        const int key = 32 * j + 16 * (u >> 2) + 4 * kg + (u & 3);
        const unsigned wd = __hip_atomic_load((gu32*)(VT + (long)key * D + (ch & ~1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e[u] = (ch & 1) ? (wd >> 16) : (wd & 0xffffu);
      }
      vv[j][0] = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
      vv[j][1] = make_uint2(e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    } else {
      const bf16_t* vp = VT + (long)(head * 64 + 16 * w + x) * LkP + 32 * j + 4 * kg;
      vv[j][0] = ld8(vp);
      vv[j][1] = ld8(vp + 16);
    }
  }
  f32x4 s[4];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt < KT) {
      s[kt] = mfma16(kf[0][kt], qf[0], s[kt]);              // S^T[key 4lg+r][row x]
      s[kt] = mfma16(kf[1][kt], qf[1], s[kt]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = s[kt][r] * 0.125f;                          // 1 / sqrt(64)
      if (kt < KT && ((mk[kt] >> (8 * r)) & 0xffu) == 0) v = MASK_FILL;
      if (kt >= KT || 16 * kt + 4 * kg + r >= Lk) v = -INFINITY;
      s[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  }
  mx = rows_max(mx);
  float den = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mx); den += s[kt][r]; }
  const float inv = 1.f / rows_sum(den);
  // O^T[c][row] = sum_key V^T[c][key] P[row][key]: the 8 K-slots of an MFMA = 4 keys of tile 2j and 4 of tile 2j+1 per lane
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (2 * j >= KT) break;
    const uint4 vf = make_uint4(vv[j][0].x, vv[j][0].y, vv[j][1].x, vv[j][1].y);
    const uint4 pf = make_uint4(pack2(s[2 * j][0] * inv, s[2 * j][1] * inv), pack2(s[2 * j][2] * inv, s[2 * j][3] * inv),
                                pack2(s[2 * j + 1][0] * inv, s[2 * j + 1][1] * inv), pack2(s[2 * j + 1][2] * inv, s[2 * j + 1][3] * inv));
    o = mfma16(vf, pf, o);                                  // O^T[channel 4lg+r][row x]
  }
  if (row < R) st8(ctx + (long)row * D + head * 64 + 16 * w + 4 * kg, make_uint2(pack2(o[0], o[1]), pack2(o[2], o[3])), wt);
}

// The same unit over a LONG memory (a dialogue history of 65 .. 512 tokens: LkP = 128, 256 or 512) in NC = LkP / 64 chunks of four key tiles
// with a running maximum / denominator (the chunk's probabilities go into the P.V product unnormalised, the context is rescaled when the
// maximum moves and divided once at the end): the register footprint of the short form, a few more dependent round trips -- the short
// form stays as it is for LkP <= 64.  MASK_FILL is finite, so the running maximum is finite after the first chunk and a chunk that lies
// wholly behind Lk contributes exp(-inf) = 0.
__device__ __forceinline__ void core_unit_long(const bf16_t* Q, const bf16_t* K, const bf16_t* VT, const unsigned char* mask, int Lk, int LkP,
                                               bf16_t* ctx, int R, int head, int mt, int w, int x, int kg, bool wt) {
  const int row = 16 * mt + x, rowc = min(row, R - 1);
  uint4 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = ld16(Q, ((long)rowc * D + head * 64 + 32 * ks + 8 * kg) * 2);
  const bf16_t* vrow = VT + (long)(head * 64 + 16 * w + x) * LkP + 4 * kg;
  float m = -INFINITY, den = 0.f;
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
  const int NC = LkP >> 6;
#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    uint4 kf[2][4];
    uint32_t mk[4];
    uint2 vv[2][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int g = 4 * c + kt;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[ks][kt] = ld16(K, ((long)(16 * g + x) * D + head * 64 + 32 * ks + 8 * kg) * 2);
      mk[kt] = *reinterpret_cast<const uint32_t*>(mask + 16 * g + 4 * kg);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { vv[j][0] = ld8(vrow + 64 * c + 32 * j); vv[j][1] = ld8(vrow + 64 * c + 32 * j + 16); }
    f32x4 s[4];
    float cm = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[kt] = mfma16(kf[0][kt], qf[0], s[kt]);              // S^T[key 4lg+r][row x]
      s[kt] = mfma16(kf[1][kt], qf[1], s[kt]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s[kt][r] * 0.125f;                        // 1 / sqrt(64)
        if (((mk[kt] >> (8 * r)) & 0xffu) == 0) v = MASK_FILL;
        if (64 * c + 16 * kt + 4 * kg + r >= Lk) v = -INFINITY;
        s[kt][r] = v;
        cm = fmaxf(cm, v);
      }
    }
    const float mn = fmaxf(m, rows_max(cm));
    const float resc = __expf(m - mn);                      // (first chunk: exp(-inf) = 0 on den = 0, o = 0)
    float cd = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mn); cd += s[kt][r]; }
    den = den * resc + rows_sum(cd);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] *= resc;
    m = mn;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint4 vf = make_uint4(vv[j][0].x, vv[j][0].y, vv[j][1].x, vv[j][1].y);
      const uint4 pf = make_uint4(pack2(s[2 * j][0], s[2 * j][1]), pack2(s[2 * j][2], s[2 * j][3]),
                                  pack2(s[2 * j + 1][0], s[2 * j + 1][1]), pack2(s[2 * j + 1][2], s[2 * j + 1][3]));
      o = mfma16(vf, pf, o);                                // O^T[channel 4lg+r][row x]
    }
  }
  const float inv = 1.f / den;
  if (row < R) st8(ctx + (long)row * D + head * 64 + 16 * w + 4 * kg, make_uint2(pack2(o[0] * inv, o[1] * inv), pack2(o[2] * inv, o[3] * inv)), wt);
}

// one memory's core unit, by the padded length of the memory
// (LONG is a property of the LAUNCH: with the long forms compiled into the same kernel, the short-memory turn of BASELINE configs[4] measured
// 7.33-7.43 ms against 7.12-7.25 ms -- code size and register allocation of the 14-phase loop -- so they live in an instance of their own.)
template <bool LONG>
__device__ __forceinline__ void core_memory(const bf16_t* Q, const bf16_t* K, const bf16_t* VT, const unsigned char* mask, int Lk, int LkP,
                                            bf16_t* ctx, int R, int head, int mt, int w, int x, int kg, bool wt) {
  if constexpr (!LONG) {
    core_unit<false>(Q, K, VT, mask, 0, Lk, LkP, ctx, R, head, mt, w, x, kg, wt);
  } else {
    if (LkP <= 64) core_unit<false>(Q, K, VT, mask, 0, Lk, LkP, ctx, R, head, mt, w, x, kg, wt);
    else core_unit_long(Q, K, VT, mask, Lk, LkP, ctx, R, head, mt, w, x, kg, wt);
  }
}

template <bool LONG>
__global__ __launch_bounds__(NT, 1) void decstack_kernel(const DecArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* img = smem;                       // [64][1024 B]
  char* part = smem + 64 * 1024;          // 4 waves x 16 (tile, row tile) x 64 lanes x 16 B
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15, kg = lane >> 4;
  // The grid is 8 x NWG blocks of which every 8th works: blocks are dealt round-robin over the 8 XCDs (observed, not promised), so
  // the NWG workers land on ONE XCD and can hand data over through its L2.  That is VERIFIED at run time (each worker publishes its
  // HW_REG_XCC_ID before the first barrier); until then, and for good if the ids differ, payload stores are write-through (wt).
  if ((a.dbg & 8) && threadIdx.x == 0)      // development: the XCC ids of ALL 256 blocks, or-ed into sync[6] (8 XCDs -> 0xff)
    __hip_atomic_fetch_or((gu32*)a.sync + 6, 1u << (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // spread s (1 / 2 / 4): of every 8 s consecutive blocks the first s work -- blocks b, b + 1, .. of a group of eight sit on XCDs b % 8, so
  // the workers cover s XCDs (s times the fabric bandwidth for the weight stream; hand-offs then write through, see wt)
  const int sp = a.spread;
  if ((int)(blockIdx.x % (8 * sp)) >= sp) return;
  const int wg = (blockIdx.x / (8 * sp)) * sp + blockIdx.x % (8 * sp), R = a.R, RP = (R + 15) & ~15, MTR = RP >> 4;
  bool wt = true;
  __shared__ unsigned same_xcd;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 15u;          // HW_REG_XCC_ID[3:0]
    __hip_atomic_fetch_or((gu32*)a.sync + 2, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned phase = 0;
  int nst = 0;
#define STAMP() do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && nst < 128) a.stamps[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
  STAMP();
  const bf16_t* xcur = a.x_in;
  int nxt = 0;
  bf16_t* const ctx = a.hbuf;             // context rows [RP][512] of the attention sublayers (the hidden rows are live in the FFN only)
  uint4 wr[16];
  f32x4 acc[4][4];
  auto a_img = [&](int mt, int ks) { return img_frag(img, mt, ks, x, kg); };
  // row-major bf16 store of a reduced tile: lane (x = row, lg = kg) holds columns col0 + 4*kg .. +3
  auto store_rows = [&](bf16_t* out, int ld, int col0, int mt, const float (&v)[4]) {
    const int row = 16 * mt + x;
    if (row < R) st8(out + (long)row * ld + col0 + 4 * kg, make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3])), wt);
  };
  auto bias4 = [&](const bf16_t* b, int col0, float (&bv)[4]) {
    const uint2 q = *reinterpret_cast<const uint2*>(b + col0 + 4 * kg);
    bv[0] = bf_lo(q.x); bv[1] = bf_hi(q.x); bv[2] = bf_lo(q.y); bv[3] = bf_hi(q.y);
  };

  // weights of the first phase
  load_w<3, 4>(wr, a.layers[0].Wqkv, D, wg * 48, w, x, kg);
#pragma unroll 1
  for (int l = 0; l < a.nl; ++l) {
    const DecLayerDev* Lp = a.layers + l;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {          // sublayers: 0 causal self-attention, 1..3 history / query / fused modalities, 4 feed-forward
      // ---------------- A(j): LN_j(x) -> this workgroup's columns of the sublayer's first projection ----------------
      layernorm_to_image(xcur, Lp->ln_a[j], Lp->ln_b[j], img, R, RP, w, lane);
      __syncthreads();
      if (j == 0) {
        partial_product<3, 4>(acc, wr, a_img, MTR, w);
        load_w<1, 4>(wr, Lp->Wo[0], D, wg * 16, w, x, kg);                 // the next product's weights fly over the barriers
        const bf16_t* bqkv = Lp->bqkv;
        bf16_t* const kc = a.kcache + (long)l * (64 * D) + (long)a.slot0 * D;     // this call's rows: slots slot0 ..
        bf16_t* const vc = a.vcache + (long)l * (64 * D) + (long)a.slot0 * D;
        reduce_tiles<3>(acc, part, MTR, w, lane, [&](int t, int mt, const f32x4& v) {
          const int col0 = wg * 48 + 16 * t;                               // column of the packed [q; k; v] output
          float bv[4]; bias4(bqkv, col0, bv);
          const float o[4] = {v[0] + bv[0], v[1] + bv[1], v[2] + bv[2], v[3] + bv[3]};
          if (col0 < 512) store_rows(a.qbuf, D, col0, mt, o);
          else if (col0 < 1024) store_rows(kc, D, col0 - 512, mt, o);
          else store_rows(vc, D, col0 - 1024, mt, o);
        });
      } else if (j < 4) {
        partial_product<1, 4>(acc, wr, a_img, MTR, w);
        load_w<1, 4>(wr, Lp->Wo[j], D, wg * 16, w, x, kg);
        const bf16_t* bq = Lp->bq[j - 1];
        reduce_tiles<1>(acc, part, MTR, w, lane, [&](int t, int mt, const f32x4& v) {
          float bv[4]; bias4(bq, wg * 16, bv);
          const float o[4] = {v[0] + bv[0], v[1] + bv[1], v[2] + bv[2], v[3] + bv[3]};
          store_rows(a.qbuf, D, wg * 16, mt, o);
        });
      } else {
        partial_product<4, 4>(acc, wr, a_img, MTR, w);
        load_w<1, 16>(wr, Lp->W2, 4 * D, wg * 16, w, x, kg);
        const bf16_t* b1 = Lp->b1;
        reduce_tiles<4>(acc, part, MTR, w, lane, [&](int t, int mt, const f32x4& v) {
          const int col0 = wg * 64 + 16 * t;
          float bv[4]; bias4(b1, col0, bv);
          const float o[4] = {fmaxf(v[0] + bv[0], 0.f), fmaxf(v[1] + bv[1], 0.f), fmaxf(v[2] + bv[2], 0.f), fmaxf(v[3] + bv[3], 0.f)};
          store_rows(a.hbuf, 4 * D, col0, mt, o);
        });
      }
      grid_barrier(a.sync, ++phase * NWG, a.dbg);
      if (phase == 1) {                    // every worker's XCC id is in: one bit set = one XCD = plain payload stores from here on
        if (tid == 0) same_xcd = __builtin_popcount(__hip_atomic_load((gu32*)a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1;
        __syncthreads();
        wt = !(same_xcd && !(a.dbg & 4));
        if (tid == 0 && wg == 0) __hip_atomic_store((gu32*)a.sync + 5, wt ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (readable: which mode ran)
      }
      STAMP();
      // ---------------- C(j): the attention core, one (head, 16-row tile) unit per workgroup ----------------
      if (j < 4) {
        const int head = wg & 7, mt = wg >> 3;
        if (mt < MTR) {
          if (j == 0) core_unit<true>(a.qbuf, a.kcache + (long)l * (64 * D), a.vcache + (long)l * (64 * D), a.smask, a.LkS, a.slot0 + R, a.LkS, ctx, R, head, mt, w, x, kg, wt);
          else core_memory<LONG>(a.qbuf, Lp->Kc[j - 1], Lp->VTc[j - 1], Lp->cmask[j - 1], Lp->Lk[j - 1], Lp->LkP[j - 1], ctx, R, head, mt, w, x, kg, wt);
        }
        grid_barrier(a.sync, ++phase * NWG, a.dbg);
        STAMP();
      }
      // ---------------- B(j): this workgroup's 16 columns of the output projection (W_o over the context rows, W_2 over the hidden
      //                  rows, both read as MFMA fragments straight from L2) + bias + residual ----------------
      {
        const bf16_t* src = j < 4 ? ctx : a.hbuf;
        const int ld = j < 4 ? D : 4 * D;
        auto a_glob = [&](int mt, int ks) {  // lane (x, kg) -> row 16*mt + x, channels 32*ks + 8*kg ..
          return ld16(src, ((long)min(16 * mt + x, R - 1) * ld + 32 * ks + 8 * kg) * 2);
        };
        if (j < 4) partial_product_glob<4, true>(acc, wr, a_glob, MTR, w);
        else if (a.dbg & 2) partial_product_glob<16, false>(acc, wr, a_glob, MTR, w);
        else partial_product_glob<16, true>(acc, wr, a_glob, MTR, w);
        if (j == 3) load_w<4, 4>(wr, Lp->W1, D, wg * 64, w, x, kg);
        else if (j < 3) load_w<1, 4>(wr, Lp->Wq[j], D, wg * 16, w, x, kg);
        else if (l + 1 < a.nl) load_w<3, 4>(wr, Lp[1].Wqkv, D, wg * 48, w, x, kg);
        const bf16_t* bo = j < 4 ? Lp->bo[j] : Lp->b2;
        bf16_t* xn = a.xbuf[nxt];
        const bf16_t* xo = xcur;
        reduce_tiles<1>(acc, part, MTR, w, lane, [&](int t, int mt, const f32x4& v) {
          const int col0 = wg * 16, row = min(16 * mt + x, R - 1);
          float bv[4]; bias4(bo, col0, bv);
          const uint2 xr = ld8(xo + (long)row * D + col0 + 4 * kg);
          const float o[4] = {v[0] + bv[0] + bf_lo(xr.x), v[1] + bv[1] + bf_hi(xr.x), v[2] + bv[2] + bf_lo(xr.y), v[3] + bv[3] + bf_hi(xr.y)};
          store_rows(xn, D, col0, mt, o);
        });
        xcur = xn; nxt ^= 1;
      }
      grid_barrier(a.sync, ++phase * NWG, a.dbg);
      STAMP();
    }
  }
  // the counter is zero again for the next call: the workgroup that leaves last (everybody is past the final barrier) resets it
  if (tid == 0) {
    gu32* sync = (gu32*)a.sync;
    if (__hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NWG - 1) {
      __hip_atomic_store(sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// =====================================================================================================================
// Head-local form for R <= 16 rows (one decode step at a time: R = hypotheses): SIX grid barriers per layer instead of fourteen.
// The column-split form above needs three barriers per attention sublayer because every product's columns are dealt to the 32
// workgroups: projection -> (barrier) -> per-head core -> (barrier) -> output projection -> (barrier) -> next LayerNorm.  With one
// 16-row tile there are only 8 core units anyway, so here workgroup hh < 8 OWNS head hh through a whole attention sublayer:
//     x <- x + b_o + sum of the previous sublayer's 8 per-head partial outputs     (every workgroup, redundantly; x lives in LDS)
//     LN(x) -> q_hh (+ k_hh, v_hh of the new rows for the self-attention) -> core of head hh -> ctx_hh [R x 64]
//     partial output projection  Y_hh = ctx_hh . W_o[:, 64 hh .. 64 hh + 63]^T  [R x 512] (K = 64), f32, to pbuf       -> ONE barrier
// and the sum over the heads happens at the head of the next phase.  The feed-forward block keeps its two column-split phases (all
// 32 workgroups: 64 hidden columns each, then 16 output columns each over K = 2048).  Hand-offs inside a workgroup (q, k / v rows,
// context rows) go through L2 as in the column-split form: store, drain, workgroup barrier, sc1 load.
// =====================================================================================================================
__device__ __forceinline__ void st16f(float* p, const f32x4& v, bool wt) {
  const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];       // (a bit_cast of a vector ELEMENT lvalue reads element 0: copy out first)
  st8(p, make_uint2(__float_as_uint(v0), __float_as_uint(v1)), wt);
  st8(p + 2, make_uint2(__float_as_uint(v2), __float_as_uint(v3)), wt);
}

// LN of the R <= 16 rows held in LDS (xs: [16][512] bf16, linear) into the swizzled image rows 0..15 (rows >= R zero): four rows per wave
__device__ __forceinline__ void layernorm_lds_to_image(const char* xs, const bf16_t* ga, const bf16_t* gb, char* img, int R, int w, int lane) {
  const int sub = lane & 15, rq = lane >> 4, row = w * 4 + rq;
  const bool live = row < R;
  uint4 q[4], qa[4], qb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    q[c] = *reinterpret_cast<const uint4*>(xs + min(row, R - 1) * 1024 + (sub * 4 + c) * 16);
    qa[c] = reinterpret_cast<const uint4*>(ga)[sub * 4 + c]; qb[c] = reinterpret_cast<const uint4*>(gb)[sub * 4 + c];
  }
  float v[32];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t u[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[c * 8 + 2 * e] = bf_lo(u[e]); v[c * 8 + 2 * e + 1] = bf_hi(u[e]); sum += v[c * 8 + 2 * e] + v[c * 8 + 2 * e + 1]; }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum * (1.f / D);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 32; ++e) { const float dlt = v[e] - mean; ss += dlt * dlt; }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
  const float inv = 1.f / (sqrtf(ss * (1.f / (D - 1))) + 1e-6f);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t ua[4] = {qa[c].x, qa[c].y, qa[c].z, qa[c].w}, ub[4] = {qb[c].x, qb[c].y, qb[c].z, qb[c].w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y0 = live ? bf_lo(ua[e]) * (v[c * 8 + 2 * e] - mean) * inv + bf_lo(ub[e]) : 0.f;
      const float y1 = live ? bf_hi(ua[e]) * (v[c * 8 + 2 * e + 1] - mean) * inv + bf_hi(ub[e]) : 0.f;
      o[e] = pack2(y0, y1);
    }
    *reinterpret_cast<uint4*>(img_at(img, row, sub * 4 + c)) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

template <bool LONG>
__global__ __launch_bounds__(NT, 1) void decstack_head_kernel(const DecArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* img = smem;                       // image rows 0..15 [16][1024 B]
  char* xs = smem + 16 * 1024;            // the residual stream of the R rows, bf16 [16][512] linear (every workgroup keeps its own copy)
  char* part = smem + 64 * 1024;          // reduce_tiles scratch
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15, kg = lane >> 4;
  if (blockIdx.x & 7) return;
  const int wg = blockIdx.x >> 3, R = a.R;
  const bool headwg = wg < H;
  const int hh = wg & 7;
  bool wt = true;
  __shared__ unsigned same_xcd;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 15u;
    __hip_atomic_fetch_or((gu32*)a.sync + 2, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned phase = 0;
  bf16_t* const ctx = a.hbuf;                                   // context rows [16][512] (attention sublayers); hidden rows in the FFN
  bf16_t* const xout = a.xbuf[(5 * a.nl - 1) & 1];              // the buffer the caller reads the result from; written once per layer
  uint4 wr[16];
  f32x4 acc[4][4];
  auto a_img = [&](int mt, int ks) { return img_frag(img, mt, ks, x, kg); };
  auto bias4 = [&](const bf16_t* b, int col0, float (&bv)[4]) {
    const uint2 q = *reinterpret_cast<const uint2*>(b + col0 + 4 * kg);
    bv[0] = bf_lo(q.x); bv[1] = bf_hi(q.x); bv[2] = bf_lo(q.y); bv[3] = bf_hi(q.y);
  };
  auto store_rows = [&](bf16_t* out, int ld, int col0, const float (&v)[4]) {          // lane (x = row, kg): columns col0 + 4 kg ..
    if (x < R) st8(out + (long)x * ld + col0 + 4 * kg, make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3])), wt);
  };
  // x rows into LDS: from global bf16 rows (mode 0) or x += bias + the 8 per-head partials of the previous sublayer (mode 1).  Thread
  // (half = tid >> 7, c4 = tid & 127) walks rows half, half + 2, ..: 4 consecutive columns each.
  auto load_x = [&](const bf16_t* src) {
    for (int row = tid >> 7; row < R; row += 2) {
      const int c = (tid & 127) * 4;
      const uint2 q = ld8(src + (long)row * D + c);
      *reinterpret_cast<uint2*>(xs + row * 1024 + c * 2) = q;
    }
  };
  auto add_partials = [&](const float* P, const bf16_t* bo) {
    const int c = (tid & 127) * 4;
    const uint2 bq = *reinterpret_cast<const uint2*>(bo + c);
    for (int row = tid >> 7; row < R; row += 2) {
      uint4 pq[H];
#pragma unroll
      for (int p = 0; p < H; ++p) pq[p] = ld16(P, ((long)(p * 16 + row) * D + c) * 4);
      const uint2 xo = *reinterpret_cast<const uint2*>(xs + row * 1024 + c * 2);
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < H; ++p) {
        s4[0] += __builtin_bit_cast(float, pq[p].x); s4[1] += __builtin_bit_cast(float, pq[p].y);
        s4[2] += __builtin_bit_cast(float, pq[p].z); s4[3] += __builtin_bit_cast(float, pq[p].w);
      }
      *reinterpret_cast<uint2*>(xs + row * 1024 + c * 2) =
          make_uint2(pack2(s4[0] + bf_lo(bq.x) + bf_lo(xo.x), s4[1] + bf_hi(bq.x) + bf_hi(xo.x)),
                     pack2(s4[2] + bf_lo(bq.y) + bf_lo(xo.y), s4[3] + bf_hi(bq.y) + bf_hi(xo.y)));
    }
  };
  // this wave's 8 column tiles (16 (8 w + t) ..) x 2 k-steps of W_o's column block of head hh: lane (x, kg) holds W_o[col][64 hh + 32 ks + 8 kg ..]
  auto load_wo = [&](const bf16_t* Wo) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wr[t * 2 + ks] = *reinterpret_cast<const uint4*>(Wo + (long)(16 * (8 * w + t) + x) * D + hh * 64 + 32 * ks + 8 * kg);
  };
  auto head_sync = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };

  if (headwg) load_w<4, 4>(wr, a.layers[0].Wqkv, D, hh * 64, w, x, kg);                  // q rows of head hh, first layer
  load_x(a.x_in);
  __syncthreads();
  int pb = 0;                                                   // partial buffer this phase WRITES
#pragma unroll 1
  for (int l = 0; l < a.nl; ++l) {
    const DecLayerDev* Lp = a.layers + l;
    if (l > 0) { load_x(xout); __syncthreads(); }
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {          // attention sublayers: 0 causal self-attention, 1..3 history / query / fused modalities
      if (j > 0) { add_partials(a.pbuf + (long)(pb ^ 1) * (H * 16 * D), Lp->bo[j - 1]); __syncthreads(); }
      float* const Pw = a.pbuf + (long)pb * (H * 16 * D) + (long)hh * (16 * D);
      if (headwg) {
        layernorm_lds_to_image(xs, Lp->ln_a[j], Lp->ln_b[j], img, R, w, lane);
        __syncthreads();
        if (j == 0) {
          bf16_t* const kc = a.kcache + (long)l * (64 * D) + (long)a.slot0 * D;
          bf16_t* const vc = a.vcache + (long)l * (64 * D) + (long)a.slot0 * D;
#pragma unroll 1
          for (int part3 = 0; part3 < 3; ++part3) {               // q, k, v rows of head hh in the packed [q; k; v] projection
            partial_product<4, 4>(acc, wr, a_img, 1, w);
            if (part3 < 2) load_w<4, 4>(wr, Lp->Wqkv, D, (part3 + 1) * D + hh * 64, w, x, kg);
            bf16_t* dst = part3 == 0 ? a.qbuf : (part3 == 1 ? kc : vc);
            const bf16_t* bqkv = Lp->bqkv + part3 * D;
            reduce_tiles<4>(acc, part, 1, w, lane, [&](int t, int mt, const f32x4& v) {
              const int col0 = hh * 64 + 16 * t;
              float bv[4]; bias4(bqkv, col0, bv);
              const float o[4] = {v[0] + bv[0], v[1] + bv[1], v[2] + bv[2], v[3] + bv[3]};
              store_rows(dst, D, col0, o);
            });
          }
        } else {
          partial_product<4, 4>(acc, wr, a_img, 1, w);
          const bf16_t* bq = Lp->bq[j - 1];
          reduce_tiles<4>(acc, part, 1, w, lane, [&](int t, int mt, const f32x4& v) {
            const int col0 = hh * 64 + 16 * t;
            float bv[4]; bias4(bq, col0, bv);
            const float o[4] = {v[0] + bv[0], v[1] + bv[1], v[2] + bv[2], v[3] + bv[3]};
            store_rows(a.qbuf, D, col0, o);
          });
        }
        load_wo(Lp->Wo[j]);                                        // flies under the core
        head_sync();                                               // q (and this step's k / v rows) are in L2
        if (j == 0) core_unit<true>(a.qbuf, a.kcache + (long)l * (64 * D), a.vcache + (long)l * (64 * D), a.smask, a.LkS, a.slot0 + R, a.LkS, ctx, R, hh, 0, w, x, kg, wt);
        else core_memory<LONG>(a.qbuf, Lp->Kc[j - 1], Lp->VTc[j - 1], Lp->cmask[j - 1], Lp->Lk[j - 1], Lp->LkP[j - 1], ctx, R, hh, 0, w, x, kg, wt);
        head_sync();                                               // the context rows of head hh are in L2
        {
          const int rowc = min(x, R - 1);
          const uint4 c0 = ld16(ctx, ((long)rowc * D + hh * 64 + 8 * kg) * 2), c1 = ld16(ctx, ((long)rowc * D + hh * 64 + 32 + 8 * kg) * 2);
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            o = mfma16(wr[t * 2], c0, o);
            o = mfma16(wr[t * 2 + 1], c1, o);                      // Y_hh^T[col 4kg + r of tile][row x]
            if (x < R) st16f(Pw + (long)x * D + 16 * (8 * w + t) + 4 * kg, o, wt);
          }
        }
        if (j < 3) load_w<4, 4>(wr, Lp->Wq[j], D, hh * 64, w, x, kg);      // the next sublayer's query rows of head hh fly over the barrier
      }
      if (j == 3) load_w<4, 4>(wr, Lp->W1, D, wg * 64, w, x, kg);          // (every workgroup) the feed-forward block's first product
      grid_barrier(a.sync, ++phase * NWG, a.dbg);
      if (phase == 1) {
        if (tid == 0) same_xcd = __builtin_popcount(__hip_atomic_load((gu32*)a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1;
        __syncthreads();
        wt = !(same_xcd && !(a.dbg & 4));
        if (tid == 0 && wg == 0) __hip_atomic_store((gu32*)a.sync + 5, wt ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      pb ^= 1;
    }
    // ---------------- feed-forward block: hidden columns 64 wg .. (barrier), output columns 16 wg .. (barrier) ----------------
    add_partials(a.pbuf + (long)(pb ^ 1) * (H * 16 * D), Lp->bo[3]);
    __syncthreads();
    layernorm_lds_to_image(xs, Lp->ln_a[4], Lp->ln_b[4], img, R, w, lane);
    __syncthreads();
    partial_product<4, 4>(acc, wr, a_img, 1, w);
    load_w<1, 16>(wr, Lp->W2, 4 * D, wg * 16, w, x, kg);
    {
      const bf16_t* b1 = Lp->b1;
      reduce_tiles<4>(acc, part, 1, w, lane, [&](int t, int mt, const f32x4& v) {
        const int col0 = wg * 64 + 16 * t;
        float bv[4]; bias4(b1, col0, bv);
        const float o[4] = {fmaxf(v[0] + bv[0], 0.f), fmaxf(v[1] + bv[1], 0.f), fmaxf(v[2] + bv[2], 0.f), fmaxf(v[3] + bv[3], 0.f)};
        store_rows(a.hbuf, 4 * D, col0, o);
      });
    }
    grid_barrier(a.sync, ++phase * NWG, a.dbg);
    {
      auto a_glob = [&](int mt, int ks) { return ld16(a.hbuf, ((long)min(x, R - 1) * (4 * D) + 32 * ks + 8 * kg) * 2); };
      partial_product_glob<16, true>(acc, wr, a_glob, 1, w);
      if (headwg && l + 1 < a.nl) load_w<4, 4>(wr, Lp[1].Wqkv, D, hh * 64, w, x, kg);
      const bf16_t* b2 = Lp->b2;
      reduce_tiles<1>(acc, part, 1, w, lane, [&](int t, int mt, const f32x4& v) {
        const int col0 = wg * 16, row = min(x, R - 1);
        float bv[4]; bias4(b2, col0, bv);
        const uint2 xr = *reinterpret_cast<const uint2*>(xs + row * 1024 + (col0 + 4 * kg) * 2);
        const float o[4] = {v[0] + bv[0] + bf_lo(xr.x), v[1] + bv[1] + bf_hi(xr.x), v[2] + bv[2] + bf_lo(xr.y), v[3] + bv[3] + bf_hi(xr.y)};
        store_rows(xout, D, col0, o);
      });
    }
    grid_barrier(a.sync, ++phase * NWG, a.dbg);
  }
  if (tid == 0) {
    gu32* sync = (gu32*)a.sync;
    if (__hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NWG - 1) {
      __hip_atomic_store(sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

// The kernel's NWG workers (every 8th block of the grid, i.e. one XCD; each holds 128 KiB of LDS = one per CU) must be resident
// TOGETHER or its grid barriers time out: the current device needs at least NWG CUs per XCD (8 XCDs) and 128 KiB of opt-in LDS per
// workgroup.  Queried once per device ordinal.
static int bist_decoder_stack_device_ok(void) {
  static signed char ok[64] = {0};            // 0 unknown, 1 yes, -1 no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64 && ok[dev] != 0) return ok[dev] > 0;
  hipDeviceProp_t p;
  int good = 0;
  if (hipGetDeviceProperties(&p, dev) == hipSuccess)
    good = p.multiProcessorCount >= 8 * NWG && (long)p.sharedMemPerBlockOptin >= 128L * 1024;
  if (dev >= 0 && dev < 64) ok[dev] = good ? 1 : -1;
  return good;
}

// The per-turn key / value caches of the decoder kernel's three memories out of the packed [k | v] projections (decoder.py:42-55: the
// memories' keys and values do not depend on the prefix): job j copies K rows [Lk][512] and writes V transposed [512][LkP] (the PV
// product's B operand reads 8 consecutive keys per lane), rows >= Lk of VT's columns zero.  One launch for every (layer, memory) of
// a turn instead of two strided copies each.  blockIdx.x = job, blockIdx.y = tile of 16 keys.
struct KvFillK { const bf16_t* src; bf16_t* K; bf16_t* VT; int Lk, LkP; long ld; };
struct KvFillArgs { KvFillK j[32]; };

__global__ __launch_bounds__(256) void kv_cache_fill_kernel(const KvFillArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t vt[16][512 + 8];
  const KvFillK& jb = a.j[blockIdx.x];
  const int t0 = blockIdx.y * 16, tid = threadIdx.x;
  if (t0 >= jb.LkP) return;
  // 16 rows x 1024 columns = 1024 16-byte pieces: four per thread; the k half goes straight out, the v half into LDS
  for (int i = tid; i < 16 * 128; i += 256) {
    const int r = i >> 7, c8 = (i & 127) * 8, t = t0 + r;
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (t < jb.Lk) q = *reinterpret_cast<const uint4*>(jb.src + (long)t * jb.ld + c8);
    if (c8 < 512) { if (t < jb.Lk) *reinterpret_cast<uint4*>(jb.K + (long)t * 512 + c8) = q; }
    else *reinterpret_cast<uint4*>(&vt[r][c8 - 512]) = q;
  }
  __syncthreads();
  for (int c = tid; c < 512; c += 256) {          // column c of V: 16 keys = 32 contiguous bytes of VT's row c
    __attribute__((aligned(16))) bf16_t o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = vt[r][c];
    uint4* dst = reinterpret_cast<uint4*>(jb.VT + (long)c * jb.LkP + t0);
    dst[0] = *reinterpret_cast<const uint4*>(&o[0]);
    dst[1] = *reinterpret_cast<const uint4*>(&o[8]);
  }
}

extern "C" int bist_decoder_cache_fill(const BistKvFill* jobs, int32_t n_jobs, int32_t dtype, void* stream) {
  BIST_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 32 && dtype == BIST_BF16, "bist_decoder_cache_fill: 1..32 jobs, bf16");
  KvFillArgs a;
  int maxp = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const BistKvFill& b = jobs[j];
    BIST_REQUIRE(b.src && b.K && b.VT && b.Lk >= 1 && b.Lk <= b.LkP && (b.LkP == 32 || b.LkP == 64 || b.LkP == 128 || b.LkP == 256 || b.LkP == 512) && b.ld >= 1024 && b.ld % 8 == 0 &&
                 (((uintptr_t)b.src | (uintptr_t)b.K | (uintptr_t)b.VT) & 15) == 0,
                 "bist_decoder_cache_fill: job %d: 1 <= Lk <= LkP in {32, 64, 128, 256, 512}, rows of [k | v] 16-byte aligned", j);
    a.j[j] = KvFillK{(const bf16_t*)b.src, (bf16_t*)b.K, (bf16_t*)b.VT, b.Lk, b.LkP, (long)b.ld};
    maxp = b.LkP > maxp ? b.LkP : maxp;
  }
  hipLaunchKernelGGL(kv_cache_fill_kernel, dim3((unsigned)n_jobs, (unsigned)(maxp / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  BIST_LAUNCH_CHECK("bist_decoder_cache_fill");
  return BIST_OK;
}

extern "C" int bist_decoder_stack_ok(int32_t R, int32_t d, int32_t h, int32_t Lk_max, int32_t dtype) {
  return dtype == BIST_BF16 && d == D && h == H && R >= 1 && R <= 64 && Lk_max >= 1 && Lk_max <= 512 && bist_decoder_stack_device_ok();
}

extern "C" int64_t bist_decoder_layer_desc_bytes(void) { return (int64_t)sizeof(DecLayerDev); }

extern "C" int bist_decoder_stack_fwd(const void* layers_dev, int32_t n_layers, const void* x_in, void* xbuf0, void* xbuf1, void* qbuf,
                                      void* kcache, void* vcache, void* hbuf, const uint8_t* self_mask, int32_t R, int32_t LkS, int32_t slot0,
                                      int32_t lk_pad_max, void* sync, float* pbuf, int32_t dtype, void* stream) {
  BIST_REQUIRE(layers_dev && x_in && xbuf0 && xbuf1 && qbuf && kcache && vcache && hbuf && self_mask && sync, "bist_decoder_stack_fwd: null pointer");
  BIST_REQUIRE(dtype == BIST_BF16 && n_layers >= 1 && R >= 1 && R <= 64 && slot0 >= 0 && slot0 + R <= 64 && LkS >= slot0 + R && LkS <= 64 && LkS % 32 == 0,
               "bist_decoder_stack_fwd: bf16, 1..64 rows, slots slot0 .. slot0 + R - 1 inside the LkS (32 or 64) key slots");
  BIST_REQUIRE(lk_pad_max == 32 || lk_pad_max == 64 || lk_pad_max == 128 || lk_pad_max == 256 || lk_pad_max == 512,
               "bist_decoder_stack_fwd: lk_pad_max = the largest padded memory length of the layer descriptors (32, 64, 128, 256 or 512)");
  const bool lng = lk_pad_max > 64;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  BIST_REQUIRE(bist_decoder_stack_device_ok(), "bist_decoder_stack_fwd: this device cannot keep the kernel's %d workgroups resident on one XCD", NWG);
  DecArgs a{(const DecLayerDev*)layers_dev, n_layers, (const bf16_t*)x_in, {(bf16_t*)xbuf0, (bf16_t*)xbuf1}, (bf16_t*)qbuf, (bf16_t*)kcache,
            (bf16_t*)vcache, slot0, (bf16_t*)hbuf, self_mask, R, LkS, (unsigned*)sync, 0, 1, nullptr, pbuf};
  a.dbg = bist_dev_dbg(1);
  { static const int xcds = [] { const char* e = getenv("BIST_DECSTACK_XCDS"); int v = e ? atoi(e) : 1; return (v == 2 || v == 4) ? v : 1; }(); a.spread = xcds; }
  a.stamps = bist_dev_stamps(1);
  // a partial buffer and one 16-row tile select the head-local form (6 grid barriers per layer instead of 14).  Measured on the beam-5
  // turn of BASELINE configs[4] it is the SLOWER one (431 vs ~380 us per six-layer step; profiles/README.md, round 3): a head's chain of
  // L2 round trips (q rows, context rows, partials) is longer than the barriers it removes, so callers pass pbuf = null by default.
  if (R <= 16 && pbuf) {
    BIST_REQUIRE((reinterpret_cast<uintptr_t>(pbuf) & 15) == 0, "bist_decoder_stack_fwd: pbuf must be 16-byte aligned");
    if (lng) {
      BIST_LDS_OPTIN(&decstack_head_kernel<true>, 128 * 1024, "bist_decoder_stack_fwd", BIST_ELAUNCH);
      hipLaunchKernelGGL(decstack_head_kernel<true>, dim3(8 * NWG), dim3(NT), 128 * 1024, st, a);
    } else {
      BIST_LDS_OPTIN(&decstack_head_kernel<false>, 128 * 1024, "bist_decoder_stack_fwd", BIST_ELAUNCH);
      hipLaunchKernelGGL(decstack_head_kernel<false>, dim3(8 * NWG), dim3(NT), 128 * 1024, st, a);
    }
  } else if (lng) {
    BIST_LDS_OPTIN(&decstack_kernel<true>, 128 * 1024, "bist_decoder_stack_fwd", BIST_ELAUNCH);
    hipLaunchKernelGGL(decstack_kernel<true>, dim3(8 * NWG), dim3(NT), 128 * 1024, st, a);
  } else {
    BIST_LDS_OPTIN(&decstack_kernel<false>, 128 * 1024, "bist_decoder_stack_fwd", BIST_ELAUNCH);
    hipLaunchKernelGGL(decstack_kernel<false>, dim3(8 * NWG), dim3(NT), 128 * 1024, st, a);
  }
  BIST_LAUNCH_CHECK("bist_decoder_stack_fwd");
  bist_count_launch(BIST_K_DECSTACK);
  return BIST_OK;
}
