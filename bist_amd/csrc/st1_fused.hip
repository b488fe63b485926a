// Fused stage 1 of the bi-directional spatio-temporal attention, forward (bf16, d = 512, h = 8, dk = 64).
//
// Reference semantics: VidEncoderLayer4.temporal2spatial stage 1 (model/encoder.py:110-123, direction 0: one attention per
// (clip, region s) over the T frames, frame mask) and spatial2temporal stage 1 (encoder.py:142-150, direction 1: one attention per
// (clip, frame t) over the S regions, no mask), each  y = x + W_o . MHA(LN(x), X_g, X_g) + b_o  with the SublayerConnection residual
// on the EXPANDED query (modules.py:42-44, 54-64, 81-100).  The query side arrives folded through W_k (Qf = (LN(x) W_q^T + b_q) W_k,h
// / sqrt(dk), rows (i, head); the key bias cancels in the softmax), so per group the kernel needs only the raw video rows X_g.
//
// One 512-thread workgroup per (clip b, chunk of NG groups); MT = 8 key tiles of 16 rows, NG = 8 / KT groups of KT tiles each:
//   0. X image: the chunk's video rows [128][512] bf16 in LDS by LDS-DMA, one 1-KiB row per wave instruction, 16-byte chunks
//      XOR-swizzled by (row & 15) on the per-lane SOURCE address; rows of padding keys / missing groups read a zero line.
//   1. wave = head hh:  V_hh = X . W_v,hh^T   [128 keys][64]  (MFMA 16x16x32; A = X rows from LDS by ds_read_b128, B = weight rows
//      straight from L2 into registers, 32 B per lane so that four lanes cover a whole 128-byte line of a weight row);
//      the fp32 result is packed to bf16 IN REGISTERS as the A operand of step 3 (a lane keeps 4 consecutive keys of one channel;
//      two key tiles make the 8 K-slots of one MFMA -- the K order of an MFMA is free as long as both operands agree).
//   2. S_hh^T = X . Qf_hh^T  [128 keys][Lq <= 32]  the same way; masked (-1e9 REPLACES the score, modules.py:60) softmax over the keys
//      of each group in registers (keys of a query row live in 4 lanes x KT x 4 registers), P packed to bf16 as the B operand.
//   3. O_hh^T = V_hh^T . P^T per group on MFMA, + b_v (rows of P sum to one), written as bf16 rows (g, i) of the context image
//      ctx [NG*Lq][512] that replaces the X image in LDS.
//   4. wave = 64 output columns:  Y = ctx . W_o^T + b_o + x[b, i]  (M = NG*Lq rows), 16-byte stores (v_permlane16_swap pairs the
//      column fragments) to Y [B, G, Lq, d].
// Neither K, V, the scores, the probabilities nor the head-concatenated context touch HBM: per group the kernel reads its video
// rows once (K*1 KiB) and writes Lq output rows; the weights (1 MiB) and Qf (160 KiB per clip) stream from L2.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int D = 512, H = 8, MT = 8;
constexpr float MASK_FILL = -1e9f;

struct St1F {
  const bf16_t* qf; const bf16_t* vft; const unsigned char* kmask;
  const bf16_t* Wv; const bf16_t* bv; const bf16_t* Wo; const bf16_t* bo; const bf16_t* xres;
  bf16_t* Y;
  int B, T, S, Lq, dir, cpc;     // cpc = chunks per clip
  unsigned long long* stamps;    // BIST_ST1F_STAMPS (development): per workgroup 8 s_memtime stamps at the phase boundaries, or null
  int dbg;                       // BIST_ST1F_DBG timing ablation (0 in production): bit0/1/2 no weight loads in step 1/2/4, bit3 no X DMA, bit4 skip step 3
  // TRAINING form (bist_st_stage1_fused_train_fwd): what the backward pass needs leaves the kernel as side outputs, and the two dropouts
  // of the sublayer (attention probabilities modules.py:62-63, sublayer output modules.py:44) are applied in place
  bf16_t* Vout;                  // [B,T,S,d]  V = X W_v^T + b_v   (what bist_st_stage1_pv_bwd reads)
  float* Pout;                   // [B,G,h,Lq,KP] probabilities BEFORE dropout, KP = keys rounded up to 4 (bist_st_stage1_pv_bwd_p)
  bf16_t* Oout;                  // [B,G,Lq,d] head-concatenated context = the output projection's input (its weight gradient)
  int KP;
  DropArg adrop;                 // attention-probability dropout, element index ((((b G + g) h + hh) Lq + i) K + key
  DropArg sdrop;                 // sublayer-output dropout, element index (row of Y) * d + column
};

__device__ uint4 g_zero_line;      // 16 zero bytes: DMA source of every padding row

__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void swap16(float& a, float& b) {      // see gemm.hip: lanes 16-31 / 48-63 of a <-> lanes 0-15 / 32-47 of b
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ float bf_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

// Fragment stream over a swizzled LDS image for the 8 k-step pairs of a 512-deep product: NF fragments per pair, read by
// inline-asm ds_read_b128 into a ring of R register slots, R-1 reads in flight ahead of the MFMAs that consume them (the
// compiler's own schedule keeps two reads in flight and waits for each right before its use: the LDS latency was exposed on
// every second fragment).  addr(kp, g) = LDS byte address of fragment g of pair kp; use(kp, f, frag) issues the MFMAs of f.
template <int NF, int R, int ITERS, class AddrF, class UseF>
__device__ __forceinline__ void frag_stream(AddrF addr, UseF use) {
  static_assert(NF % R == 0 && R >= 3 && R <= 9, "ring slots must repeat every step");
  u32x4 ring[R];
  static_for<R - 1>([&ring, &addr](auto g) {
    const unsigned a0 = addr(0, (int)g);
    u32x4& slot = ring[g];
    asm volatile("ds_read_b128 %0, %1" : "=v"(slot) : "v"(a0) : "memory");
  });
#pragma unroll 1
  for (int kp = 0; kp < ITERS; ++kp) {
    const int kn = min(kp + 1, ITERS - 1);  // the reads past the last step re-read it (never consumed)
    static_for<NF>([&ring, &addr, &use, kp, kn](auto fc) {
      constexpr int f = fc, gq = f + R - 1;
      u32x4& cur = ring[f % R];
      u32x4& fut = ring[gq % R];
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(R - 2) : "memory");
      asm volatile("" : "+v"(cur));
      const unsigned nxt = gq < NF ? addr(kp, gq) : addr(kn, gq - NF);
      asm volatile("ds_read_b128 %0, %1" : "=v"(fut) : "v"(nxt) : "memory");
      use(kp, fc, __builtin_bit_cast(uint4, cur));
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  static_for<R>([&ring](auto g) {
    u32x4& slot = ring[g];
    asm volatile("" : "+v"(slot));
  });
}

// Operand fragments that come straight from L2 into registers, as inline asm with counted waits: next to the inline-asm LDS
// reads the compiler waits vmcnt(0) for its own loads, i.e. for the pair just issued instead of the one about to be used.
// Weight fragments are in MFMA-fragment order (bist_pack_frag_rows): tile (16 rows) nt, k-step pair kp, parity e is one 1-KiB
// block [lane][8 bf16], so a wave's dwordx4 load reads 1 KiB contiguously.  base = this wave's first tile + lane*8.
__device__ __forceinline__ void gload(u32x4& dst, const bf16_t* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
template <int N>
__device__ __forceinline__ void tie(u32x4 (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}
__device__ __forceinline__ void load_packed4(u32x4 (&f)[4], const bf16_t* base, int kp, int e) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) gload(f[nt], base + (((nt * 8 + kp) * 2 + e) << 9));
}
__device__ __forceinline__ void load_q2(u32x4 (&f)[2], const bf16_t* q0, const bf16_t* q1, int kp, int e) {
  gload(f[0], q0 + kp * 64 + e * 8);
  gload(f[1], q1 + kp * 64 + e * 8);
}
__device__ __forceinline__ uint4 u4(const u32x4& v) { return __builtin_bit_cast(uint4, v); }

// wave-wide max / sum over the four 16-lane rows of a wave (lanes x, x+16, x+32, x+48) without LDS traffic: v_permlane16_swap and
// v_permlane32_swap of a value with its own copy leave the partner rows' values in the second register
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_max(float v) {
  float p = v, q = v;
  swap16(p, q); v = fmaxf(p, q);
  p = v; q = v;
  swap32(p, q); return fmaxf(p, q);
}
__device__ __forceinline__ float rows_sum(float v) {
  float p = v, q = v;
  swap16(p, q); v = p + q;
  p = v; q = v;
  swap32(p, q); return p + q;
}

struct Chunk { int b, g0, ng, w, lane; };
#define STAMP(i_) do { if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime(); } while (0)

// The work of one chunk with MTA active key tiles (NGA = MTA / KT groups) and MT4A row tiles in the output projection.  The full
// chunk is <KT, 8, MT4>; the last chunk of a clip (G % NG groups) runs the smallest instantiation that holds it.
template <int KT, int MTA, int MT4A, bool TRAIN = false>
__device__ __forceinline__ void chunk_body(const St1F& a, const Chunk c, char* smem) {
  constexpr int NGA = MTA / KT;
  const int w = c.w, lane = c.lane, b = c.b, g0 = c.g0, ng = c.ng;
  const int x = lane & 15, kg = lane >> 4;
  const int T_ = a.T, S_ = a.S, Lq = a.Lq;
  const int G = a.dir == 0 ? S_ : T_, K = a.dir == 0 ? T_ : S_;

  STAMP(0);
  // ---- 0. X image -----------------------------------------------------------------------------------------
  constexpr int NDMA = 2 * MTA;                 // rows per wave
  {
    const bf16_t* vb = a.vft + (long)b * T_ * S_ * D;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const int r = w * NDMA + j, gl = r / (16 * KT), kk = r % (16 * KT);
      const bool valid = gl < ng && kk < K;
      const int t = a.dir == 0 ? kk : g0 + gl, s = a.dir == 0 ? g0 + gl : kk;
      const bf16_t* src = valid ? vb + ((long)t * S_ + s) * D + ((lane ^ (r & 15)) << 3) : reinterpret_cast<const bf16_t*>(&g_zero_line);
      if (!(a.dbg & 8)) __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(smem + r * 1024), 16, 0, 0);
    }
    asm volatile("" ::: "memory");          // keep the loads below behind the DMAs in program order (counted wait in step 1)
  }
  // mask of this lane's keys kk = 16*kt + 4*kg + r (the same for every group and query row): bit kt*4 + r of `repl` set = the score
  // is replaced (by -1e9 where the key mask is 0, modules.py:60; by -inf on padding keys, which then get probability 0)
  unsigned repl = 0, padb = 0;
  {
    const unsigned char* mk = a.kmask ? a.kmask + (long)b * K : nullptr;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 16 * kt + 4 * kg + r;
        if (kk >= K) { repl |= 1u << (kt * 4 + r); padb |= 1u << (kt * 4 + r); }
        else if (mk && mk[kk] == 0) repl |= 1u << (kt * 4 + r);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  // fragment g (parity e = g / NM, tile mt = g % NM) of pair kp in a swizzled [rows][1024 B] image: lane (x, kg) reads row
  // 16*mt + x, logical 16-byte chunk 8*kp + 2*kg + e, stored at chunk ^ x
  const unsigned img0 = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)x * 1024u;
  auto img_addr = [&](int kp, int e, int mt) -> unsigned { return img0 + (unsigned)mt * 16384u + (unsigned)(((8 * kp + 2 * kg + e) ^ x) << 4); };

  // ---- 1. V_hh = X . W_v,hh^T ------------------------------------------------------------------------------
  uint4 vpk[MTA / 2][4];
  {
    f32x4 acc[MTA][4];
#pragma unroll
    for (int mt = 0; mt < MTA; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wv = a.Wv + ((long)(w * 4) << 13) + lane * 8;       // tile = 8 pairs x 2 parities x 512 elements
    u32x4 bw[2][4];
    load_packed4(bw[0], wv, 0, 0);
    load_packed4(bw[1], wv, 0, 1);
    __builtin_amdgcn_s_barrier();            // every wave's share of the X image has landed (vmcnt(0) above); the 8 weight loads fly on
    asm volatile("" ::: "memory");
    STAMP(1);
    frag_stream<2 * MTA, 4, 8>([&](int kp, int g) { return img_addr(kp, g / MTA, g % MTA); },
                            [&](int kp, auto fc, const uint4& ax) {
                              constexpr int f = fc, e = f / MTA, mt = f % MTA;
                              if constexpr (mt == 0) {            // this parity's weights have landed; the other parity's 4 loads stay in flight
                                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                                tie(bw[e]);
                              }
#pragma unroll
                              for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(ax, u4(bw[e][nt]), acc[mt][nt]);
                              if constexpr (mt == MTA - 1) {      // last use of this parity's weights: fetch the next pair's in place
                                if (!(a.dbg & 1)) load_packed4(bw[e], wv, min(kp + 1, 7), e);
                              }
                            });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the re-reads issued by the last pair (never consumed)
    tie(bw[0]); tie(bw[1]);
    // acc[mt][nt][r] = V[key 16*mt + 4*kg + r][channel 16*nt + x]  ->  A fragments of step 3 (K-slots: tile 2p regs 0..3, tile 2p+1 regs 0..3)
#pragma unroll
    for (int p = 0; p < MTA / 2; ++p)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        vpk[p][nt] = make_uint4(pack2(acc[2 * p][nt][0], acc[2 * p][nt][1]), pack2(acc[2 * p][nt][2], acc[2 * p][nt][3]),
                                pack2(acc[2 * p + 1][nt][0], acc[2 * p + 1][nt][1]), pack2(acc[2 * p + 1][nt][2], acc[2 * p + 1][nt][3]));
  }

  STAMP(2);
  // ---- 2. S_hh^T = X . Qf_hh^T -----------------------------------------------------------------------------
  f32x4 sacc[MTA][2];
  {
#pragma unroll
    for (int mt = 0; mt < MTA; ++mt)
#pragma unroll
      for (int it = 0; it < 2; ++it) sacc[mt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* q0 = a.qf + (((long)b * Lq + min(x, Lq - 1)) * H + w) * D + kg * 16;
    const bf16_t* q1 = a.qf + (((long)b * Lq + min(16 + x, Lq - 1)) * H + w) * D + kg * 16;
    u32x4 bq[2][2];
    load_q2(bq[0], q0, q1, 0, 0);
    load_q2(bq[1], q0, q1, 0, 1);
    frag_stream<2 * MTA, 4, 8>([&](int kp, int g) { return img_addr(kp, g / MTA, g % MTA); },
                            [&](int kp, auto fc, const uint4& ax) {
                              constexpr int f = fc, e = f / MTA, mt = f % MTA;
                              if constexpr (mt == 0) {
                                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                                tie(bq[e]);
                              }
#pragma unroll
                              for (int it = 0; it < 2; ++it) sacc[mt][it] = mfma16(ax, u4(bq[e][it]), sacc[mt][it]);
                              if constexpr (mt == MTA - 1) {
                                if (!(a.dbg & 2)) load_q2(bq[e], q0, q1, min(kp + 1, 7), e);
                              }
                            });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie(bq[0]); tie(bq[1]);
  }
  float bvf[4][4];                               // value bias of this lane's output channels hh*64 + 16*nt + 4*kg + r
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const uint2 q = *reinterpret_cast<const uint2*>(a.bv + w * 64 + nt * 16 + kg * 4);
    bvf[nt][0] = bf_lo(q.x); bvf[nt][1] = bf_hi(q.x); bvf[nt][2] = bf_lo(q.y); bvf[nt][3] = bf_hi(q.y);
  }
  // softmax of every (group, query row) over the group's keys, in registers, BEFORE the barrier: the two waves of a SIMD do not
  // finish steps 1-2 together, and this vector work of the first runs under the matrix work of the second
  uint4 pf[NGA][2][KT / 2];
  float rsum[TRAIN ? NGA : 1][2];                // TRAIN: sum of the KEPT (dropped-out, rescaled) probabilities of this lane's query rows
  if (!(a.dbg & 16)) {
    const bool plain = repl == 0;            // per lane; uniform in the common case (no padding keys, nothing masked)
    const unsigned long long akey = (TRAIN && a.adrop.p > 0.f) ? a.adrop.key() : 0ULL;
    const float aks = (TRAIN && a.adrop.p > 0.f) ? a.adrop.keep_scale() : 1.f;
#pragma unroll
    for (int gl = 0; gl < NGA; ++gl)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        float sv[KT][4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = sacc[gl * KT + kt][it][r];
            if (!plain) v = (repl >> (kt * 4 + r) & 1u) ? ((padb >> (kt * 4 + r) & 1u) ? -INFINITY : MASK_FILL) : v;
            sv[kt][r] = v;
            mx = fmaxf(mx, v);
          }
        mx = rows_max(mx);
        float den = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { sv[kt][r] = __expf(sv[kt][r] - mx); den += sv[kt][r]; }
        const float inv = 1.f / rows_sum(den);
        if constexpr (TRAIN) {
          const int i = it * 16 + x;
          const bool live = gl < ng && i < Lq;
          // probabilities before dropout, row (b, g, hh, i) of Pout: this lane's keys 16 kt + 4 kg + r are 4 consecutive floats
          const long prow = ((((long)b * G + g0 + gl) * H + w) * Lq + i);
          float kept = 0.f;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            const int kk0 = 16 * kt + 4 * kg;
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[kt][r] *= inv;
            if (live && kk0 < a.KP) *reinterpret_cast<float4*>(a.Pout + prow * a.KP + kk0) = make_float4(sv[kt][0], sv[kt][1], sv[kt][2], sv[kt][3]);
            if (a.adrop.p > 0.f) {
#pragma unroll
              for (int r = 0; r < 4; ++r) sv[kt][r] *= drop_mul(akey, (unsigned long long)prow * K + kk0 + r, a.adrop.p, aks);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) kept += sv[kt][r];
          }
          rsum[gl][it] = rows_sum(kept);
#pragma unroll
          for (int k2 = 0; k2 < KT / 2; ++k2)
            pf[gl][it][k2] = make_uint4(pack2(sv[2 * k2][0], sv[2 * k2][1]), pack2(sv[2 * k2][2], sv[2 * k2][3]),
                                        pack2(sv[2 * k2 + 1][0], sv[2 * k2 + 1][1]), pack2(sv[2 * k2 + 1][2], sv[2 * k2 + 1][3]));
        } else {
#pragma unroll
          for (int k2 = 0; k2 < KT / 2; ++k2)
            pf[gl][it][k2] = make_uint4(pack2(sv[2 * k2][0] * inv, sv[2 * k2][1] * inv), pack2(sv[2 * k2][2] * inv, sv[2 * k2][3] * inv),
                                        pack2(sv[2 * k2 + 1][0] * inv, sv[2 * k2 + 1][1] * inv), pack2(sv[2 * k2 + 1][2] * inv, sv[2 * k2 + 1][3] * inv));
        }
      }
  }
  STAMP(3);
  __syncthreads();          // every wave is done with the X image: it becomes the context image
  STAMP(4);

  if (TRAIN && a.Vout != nullptr) {
    // V side output (skipped when the caller keeps a value projection of its own).  vpk holds V (without its bias) as packed bf16 in MFMA-fragment order -- lane (x, kg): channel 16 nt + x of keys
    // 16 mt + 4 kg + r.  Each wave turns its [32 keys][64 channels] pieces into 128-byte rows through a 4 KiB staging tile of its own in
    // the part of the dead X image that the context rows (NGA * Lq <= 96 rows) do not use, and stores them 16 bytes per lane.
    char* stg = smem + 96 * 1024 + w * 4096;
    float bvx[4];                                // value bias of this lane's channel 16 nt + x
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bvx[nt] = (float)a.bv[w * 64 + nt * 16 + x];
#pragma unroll
    for (int p = 0; p < MTA / 2; ++p) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const uint4 q = vpk[p][nt];
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {             // j = 4 * (tile parity) + r
          const float val = ((j & 1) ? bf_hi(u[j >> 1]) : bf_lo(u[j >> 1])) + bvx[nt];
          const int kl = (j >> 2) * 16 + 4 * kg + (j & 3);
          *reinterpret_cast<bf16_t*>(stg + kl * 128 + (nt * 16 + x) * 2) = (bf16_t)val;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int kl = q4 * 8 + (lane >> 3), r = p * 32 + kl;          // key row of the chunk image
        const int gl = r / (16 * KT), kk = r % (16 * KT);
        const uint4 row = *reinterpret_cast<const uint4*>(stg + kl * 128 + (lane & 7) * 16);
        if (gl < ng && kk < K) {
          const int t = a.dir == 0 ? kk : g0 + gl, sI = a.dir == 0 ? g0 + gl : kk;
          *reinterpret_cast<uint4*>(a.Vout + (((long)b * T_ + t) * S_ + sI) * D + w * 64 + (lane & 7) * 8) = row;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  // ---- 3. context rows: O^T = V^T . P^T per group, + b_v, as bf16 rows (g, i) of the context image ------------------------------
#pragma unroll
  for (int gl = 0; gl < NGA; ++gl) {
    if (gl >= ng || (a.dbg & 16)) break;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int c0 = w * 64 + nt * 16 + kg * 4;                    // this lane's 4 consecutive output channels
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < KT / 2; ++k2) o = mfma16(vpk[gl * (KT / 2) + k2][nt], pf[gl][it][k2], o);
        const int i = it * 16 + x;
        if (i < Lq) {
          const int row = gl * Lq + i;
          const float bs = TRAIN ? rsum[TRAIN ? gl : 0][it] : 1.f;      // dropped probabilities do not sum to one: P'(V + b) = P'V + rowsum(P') b
          char* dst = smem + row * 1024 + ((((c0 >> 3)) ^ (row & 15)) << 4) + (kg & 1) * 8;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o[0] + bs * bvf[nt][0], o[1] + bs * bvf[nt][1]), pack2(o[2] + bs * bvf[nt][2], o[3] + bs * bvf[nt][3]));
        }
      }
    }
  }
  __syncthreads();
  STAMP(5);
  if constexpr (TRAIN) {
    // the context rows (g, i) as they lie in the image (16-byte chunks swizzled by row & 15) -> Oout [B, G, Lq, d], one 1-KiB row per wave step
    const int rows_o = ng * Lq;
    for (int row = w; row < rows_o; row += 8) {
      const int gl = row / Lq, i = row - gl * Lq;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + row * 1024 + ((lane ^ (row & 15)) << 4));
      *reinterpret_cast<uint4*>(a.Oout + (((long)b * G + g0 + gl) * Lq + i) * D + lane * 8) = v;
    }
  }

  // ---- 4. Y = ctx . W_o^T + b_o + x ----------------------------------------------------------------------------
  {
    f32x4 acc[MT4A][4];
#pragma unroll
    for (int mt = 0; mt < MT4A; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wo = a.Wo + ((long)(w * 4) << 13) + lane * 8;
    // W_o fragments of NP k-step pairs in registers (aw[pair parity][k parity][tile]); each set of four fragments is re-fetched in
    // place, NP pairs ahead, right after its last MFMA.  NP = 2 where the accumulators leave room (a pair of this step is only
    // 8 * MT4A MFMAs per wave: one pair ahead does not cover the L2 latency), NP = 1 with 8 row tiles.
    constexpr int NP = MT4A <= 5 ? 2 : 1;
    u32x4 aw[NP][2][4];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int e = 0; e < 2; ++e) load_packed4(aw[p][e], wo, p, e);
    frag_stream<2 * NP * MT4A, 4, 8 / NP>([&](int it, int g) { return img_addr(NP * it + g / (2 * MT4A), (g / MT4A) & 1, g % MT4A); },
                                          [&](int it, auto fc, const uint4& cx) {
                                            constexpr int f = fc, p = f / (2 * MT4A), e = (f / MT4A) & 1, mt = f % MT4A;
                                            if constexpr (mt == 0) {        // this set has landed; the sets issued after it stay in flight
                                              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (2 * NP - 1)) : "memory");
                                              tie(aw[p][e]);
                                            }
#pragma unroll
                                            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(u4(aw[p][e][nt]), cx, acc[mt][nt]);
                                            if constexpr (mt == MT4A - 1) {
                                              if (!(a.dbg & 4)) load_packed4(aw[p][e], wo, min(NP * it + p + NP, 7), e);
                                            }
                                          });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < NP; ++p) { tie(aw[p][0]); tie(aw[p][1]); }
    // residual rows x[b, i] and output offsets of this lane's rows (row = 16*mt + x)
    const int cofs = (kg & 1) * 16 + (kg >> 1) * 8;
    const int rows = ng * Lq;
    constexpr bool PRE = MT4A <= 5;                       // all residual rows in flight together, ahead of the swaps and stores below
    uint4 xr[PRE ? MT4A : 1][2];
    long yoff[MT4A], xoff[MT4A];
#pragma unroll
    for (int mt = 0; mt < MT4A; ++mt) {
      const int row = min(mt * 16 + x, rows - 1), gl = row / Lq, i = row - gl * Lq;
      yoff[mt] = (((long)b * G + g0 + gl) * Lq + i) * D + w * 64 + cofs;
      xoff[mt] = ((long)b * Lq + i) * D + w * 64 + cofs;
      if constexpr (PRE) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) xr[mt][jp] = *reinterpret_cast<const uint4*>(a.xres + xoff[mt] + jp * 32);
      }
    }
    STAMP(6);
    // acc[mt][nt][r] = Y[row 16*mt + x][column 64*w + 16*nt + 4*kg + r]; after the swap a lane holds 8 consecutive columns
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const uint4 bq = *reinterpret_cast<const uint4*>(a.bo + w * 64 + jp * 32 + cofs);
#pragma unroll
      for (int mt = 0; mt < MT4A; ++mt) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = acc[mt][2 * jp][r], q = acc[mt][2 * jp + 1][r];
          swap16(p, q);
          v[r] = p; v[4 + r] = q;
        }
        uint4 xq;
        if constexpr (PRE) xq = xr[mt][jp]; else xq = *reinterpret_cast<const uint4*>(a.xres + xoff[mt] + jp * 32);
        const float bb[8] = {bf_lo(bq.x), bf_hi(bq.x), bf_lo(bq.y), bf_hi(bq.y), bf_lo(bq.z), bf_hi(bq.z), bf_lo(bq.w), bf_hi(bq.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bb[e];
        if constexpr (TRAIN) {
          if (a.sdrop.p > 0.f) {               // y = x + dropout(W_o ctx + b_o): mask index = (row of Y) * d + column, as in the GEMM epilogue
            const unsigned long long i4 = (unsigned long long)(yoff[mt] + jp * 32) >> 2;
            const uint32_t thr = drop_threshold(a.sdrop.p);
            const float ks = a.sdrop.keep_scale();
            const unsigned long long skey = a.sdrop.key();
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
              const uint64_t bits = drop_bits4(skey, i4 + q4);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * q4 + e] = drop_keep_of(bits, e, thr) ? v[4 * q4 + e] * ks : 0.f;
            }
          }
        }
        const uint4 o = make_uint4(pack2(v[0] + bf_lo(xq.x), v[1] + bf_hi(xq.x)), pack2(v[2] + bf_lo(xq.y), v[3] + bf_hi(xq.y)),
                                   pack2(v[4] + bf_lo(xq.z), v[5] + bf_hi(xq.z)), pack2(v[6] + bf_lo(xq.w), v[7] + bf_hi(xq.w)));
        if (mt * 16 + x < rows) *reinterpret_cast<uint4*>(a.Y + yoff[mt] + jp * 32) = o;
      }
    }
  }
  STAMP(7);
}

template <int KT, int MT4, bool TRAIN = false>
__global__ __launch_bounds__(512, 2) void st1_fused_kernel(const St1F a) {
  constexpr int NG = MT / KT;
  constexpr int LQC = MT4 * 16 / NG;                    // the query-length class of this instantiation (20 or 32 rows per group)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int G = a.dir == 0 ? a.S : a.T;
  // Block order: the B * (G / NG) full chunks first, in XCD-contiguous order (blocks b, b+8, ... share an XCD and its L2: each XCD
  // walks a contiguous range of chunks, so the Qf rows of the few clips it works on stay in that L2), then the B short chunks
  // (G % NG groups) -- they cost less and fill the last, partly empty round of workgroups.
  const int fpc = G / NG;                                // full chunks per clip
  const unsigned nfull = (unsigned)(a.B * fpc), bid = blockIdx.x;
  Chunk c;
  if (bid < nfull) {
    const unsigned xcd = bid & 7u, qq = nfull >> 3, rr = nfull & 7u;
    const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    c.b = lid / fpc; c.g0 = (lid % fpc) * NG; c.ng = NG;
  } else {
    c.b = bid - nfull; c.g0 = fpc * NG; c.ng = G - c.g0;
  }
  c.lane = threadIdx.x & 63; c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (KT == 2) {
    if (c.ng == 1) { chunk_body<2, 2, (LQC + 15) / 16, TRAIN>(a, c, smem); return; }
    if (c.ng == 2) { chunk_body<2, 4, (2 * LQC + 15) / 16, TRAIN>(a, c, smem); return; }
  } else if constexpr (KT == 4) {
    if (c.ng == 1) { chunk_body<4, 4, (LQC + 15) / 16, TRAIN>(a, c, smem); return; }
  }
  chunk_body<KT, MT, MT4, TRAIN>(a, c, smem);
}

// W [rows][cols] row-major -> fragment order [rows/16][cols/64][2][64 lanes][8]: lane (x = lane & 15, kg = lane >> 4) of block
// (tile nt, pair kp, parity e) holds W[16*nt + x][64*kp + 16*kg + 8*e .. +7] -- the operand of one v_mfma_f32_16x16x32_bf16
__global__ void pack_frag_rows_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ out, int rows, int cols) {
  const long piece = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece per thread
  const int kps = cols >> 6;
  if (piece >= (long)rows * cols / 8) return;
  const int lane = piece & 63, e = (piece >> 6) & 1;
  const long blk = piece >> 7;
  const int kp = blk % kps, nt = blk / kps;
  const bf16_t* src = W + (long)(16 * nt + (lane & 15)) * cols + 64 * kp + 16 * (lane >> 4) + 8 * e;
  reinterpret_cast<uint4*>(out)[piece] = *reinterpret_cast<const uint4*>(src);
}

// several same-sized weights in one launch (the training step re-packs the value / output projections of every reasoning layer)
constexpr int PACK_MAX = 32;
struct PackSets { const bf16_t* src[PACK_MAX]; bf16_t* dst[PACK_MAX]; };
__global__ void pack_frag_rows_multi_kernel(const PackSets sets, int rows, int cols) {
  const long piece = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kps = cols >> 6;
  if (piece >= (long)rows * cols / 8) return;
  const bf16_t* W = sets.src[blockIdx.y];
  bf16_t* out = sets.dst[blockIdx.y];
  const int lane = piece & 63, e = (piece >> 6) & 1;
  const long blk = piece >> 7;
  const int kp = blk % kps, nt = blk / kps;
  const bf16_t* src = W + (long)(16 * nt + (lane & 15)) * cols + 64 * kp + 16 * (lane >> 4) + 8 * e;
  reinterpret_cast<uint4*>(out)[piece] = *reinterpret_cast<const uint4*>(src);
}

template <int KT, int MT4, bool TRAIN = false>
int launch(const St1F& a, hipStream_t st) {
  BIST_LDS_OPTIN((&st1_fused_kernel<KT, MT4, TRAIN>), MT * 16 * 1024, "bist_st_stage1_fused_fwd", BIST_ELAUNCH);
  hipLaunchKernelGGL((st1_fused_kernel<KT, MT4, TRAIN>), dim3((unsigned)(a.B * a.cpc)), dim3(512), MT * 16 * 1024, st, a);
  BIST_LAUNCH_CHECK("bist_st_stage1_fused_fwd");
  bist_count_launch(TRAIN ? BIST_K_ST1_FUSED_TRAIN : BIST_K_ST1_FUSED);
  return BIST_OK;
}

}  // namespace

extern "C" int bist_pack_frag_rows(const void* W, void* out, int32_t rows, int32_t cols, int32_t dtype, void* stream) {
  BIST_REQUIRE(W && out && W != out && rows > 0 && cols > 0, "bist_pack_frag_rows: bad argument");
  BIST_REQUIRE(dtype == BIST_BF16 && rows % 16 == 0 && cols % 64 == 0, "bist_pack_frag_rows: bf16 [rows %% 16 == 0][cols %% 64 == 0] only");
  BIST_REQUIRE(((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "bist_pack_frag_rows: 16-byte alignment");
  const long pieces = (long)rows * cols / 8;
  hipLaunchKernelGGL(pack_frag_rows_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const bf16_t*)W, (bf16_t*)out, rows, cols);
  BIST_LAUNCH_CHECK("bist_pack_frag_rows");
  return BIST_OK;
}

extern "C" int bist_pack_frag_rows_multi(const void* const* Ws, void* const* outs, int32_t n, int32_t rows, int32_t cols, int32_t dtype, void* stream) {
  BIST_REQUIRE(Ws && outs && n >= 1 && n <= PACK_MAX && rows > 0 && cols > 0, "bist_pack_frag_rows_multi: 1..%d matrices", PACK_MAX);
  BIST_REQUIRE(dtype == BIST_BF16 && rows % 16 == 0 && cols % 64 == 0, "bist_pack_frag_rows_multi: bf16 [rows %% 16 == 0][cols %% 64 == 0] only");
  PackSets k{};
  for (int i = 0; i < n; ++i) {
    BIST_REQUIRE(Ws[i] && outs[i] && Ws[i] != outs[i] && ((reinterpret_cast<uintptr_t>(Ws[i]) | reinterpret_cast<uintptr_t>(outs[i])) & 15) == 0,
                 "bist_pack_frag_rows_multi: matrix %d null, in place or not 16-byte aligned", i);
    k.src[i] = (const bf16_t*)Ws[i]; k.dst[i] = (bf16_t*)outs[i];
  }
  const long pieces = (long)rows * cols / 8;
  hipLaunchKernelGGL(pack_frag_rows_multi_kernel, dim3((unsigned)((pieces + 255) / 256), (unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), k, rows, cols);
  BIST_LAUNCH_CHECK("bist_pack_frag_rows_multi");
  return BIST_OK;
}

extern "C" int bist_st_stage1_fused_ok(int32_t T, int32_t S, int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype) {
  const int K = direction == 0 ? T : S;
  return dtype == BIST_BF16 && d == D && h == H && Lq >= 1 && Lq <= 32 && K >= 1 && K <= 128 && T >= 1 && S >= 1 &&
         (direction == 0 || direction == 1);
}

namespace {
int fused_common(const void* qf, const void* vft, const uint8_t* kmask, const void* Wv, const void* bv, const void* Wo, const void* bo,
                 const void* xres, void* Y, int32_t B, int32_t T, int32_t S, int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype,
                 void* Vout, float* Pout, void* Oout, const BistDrop* attn_drop, const BistDrop* sub_drop, bool train, void* stream) {
  const char* who = train ? "bist_st_stage1_fused_train_fwd" : "bist_st_stage1_fused_fwd";
  BIST_REQUIRE(qf && vft && Wv && bv && Wo && bo && xres && Y && B > 0, "%s: null pointer or empty batch", who);
  BIST_REQUIRE(bist_st_stage1_fused_ok(T, S, Lq, d, h, direction, dtype), "%s: shape outside the kernel's envelope (bf16, d=512, h=8, Lq<=32, keys<=128)", who);
  const void* ptrs[] = {qf, vft, Wv, bv, Wo, bo, xres, Y, Vout, Pout, Oout};
  for (const void* p : ptrs) BIST_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "%s: operands must be 16-byte aligned", who);
  const int K = direction == 0 ? T : S, G = direction == 0 ? S : T;
  const int KT = K <= 32 ? 2 : K <= 64 ? 4 : 8, NG = MT / KT;
  St1F a{(const bf16_t*)qf, (const bf16_t*)vft, kmask, (const bf16_t*)Wv, (const bf16_t*)bv, (const bf16_t*)Wo, (const bf16_t*)bo,
         (const bf16_t*)xres, (bf16_t*)Y, B, T, S, Lq, direction, (G + NG - 1) / NG, nullptr, 0,
         (bf16_t*)Vout, Pout, (bf16_t*)Oout, (K + 3) / 4 * 4, make_drop(attn_drop), make_drop(sub_drop)};
  a.stamps = bist_dev_stamps(0);
  a.dbg = bist_dev_dbg(0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int need = (NG * Lq + 15) / 16;          // 16-row tiles of the output projection
  if (train) {
    BIST_REQUIRE(Pout && Oout, "%s: null side output", who);
    BIST_REQUIRE(bist_st_stage1_fused_train_ok(T, S, Lq, d, h, direction, dtype), "%s: the staging tiles of the value rows need NG * Lq <= 96 context rows", who);
    BIST_REQUIRE((!attn_drop || (attn_drop->p >= 0.f && attn_drop->p < 1.f)) && (!sub_drop || (sub_drop->p >= 0.f && sub_drop->p < 1.f)), "%s: drop p out of range", who);
    if (KT == 2) return need <= 5 ? launch<2, 5, true>(a, st) : launch<2, 8, true>(a, st);
    if (KT == 4) return need <= 3 ? launch<4, 3, true>(a, st) : launch<4, 4, true>(a, st);
    return launch<8, 2, true>(a, st);
  }
  if (KT == 2) return need <= 5 ? launch<2, 5>(a, st) : launch<2, 8>(a, st);
  if (KT == 4) return need <= 3 ? launch<4, 3>(a, st) : launch<4, 4>(a, st);
  return launch<8, 2>(a, st);
}
}  // namespace

extern "C" int bist_st_stage1_fused_train_ok(int32_t T, int32_t S, int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype) {
  if (!bist_st_stage1_fused_ok(T, S, Lq, d, h, direction, dtype)) return 0;
  const int K = direction == 0 ? T : S;
  const int KT = K <= 32 ? 2 : K <= 64 ? 4 : 8, NG = MT / KT;
  return NG * Lq <= 96;
}

extern "C" int bist_st_stage1_fused_fwd(const void* qf, const void* vft, const uint8_t* kmask, const void* Wv, const void* bv,
                                        const void* Wo, const void* bo, const void* xres, void* Y, int32_t B, int32_t T, int32_t S,
                                        int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype, void* stream) {
  return fused_common(qf, vft, kmask, Wv, bv, Wo, bo, xres, Y, B, T, S, Lq, d, h, direction, dtype, nullptr, nullptr, nullptr, nullptr, nullptr, false, stream);
}

extern "C" int bist_st_stage1_fused_train_fwd(const void* qf, const void* vft, const uint8_t* kmask, const void* Wv, const void* bv,
                                              const void* Wo, const void* bo, const void* xres, void* Y, void* Vout, float* Pout, void* Oout,
                                              const BistDrop* attn_drop, const BistDrop* sub_drop, int32_t B, int32_t T, int32_t S,
                                              int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype, void* stream) {
  return fused_common(qf, vft, kmask, Wv, bv, Wo, bo, xres, Y, B, T, S, Lq, d, h, direction, dtype, Vout, Pout, Oout, attn_drop, sub_drop, true, stream);
}
