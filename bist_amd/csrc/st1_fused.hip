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

namespace {

constexpr int D = 512, H = 8, MT = 8;
constexpr float MASK_FILL = -1e9f;

struct St1F {
  const bf16_t* qf; const bf16_t* vft; const unsigned char* kmask;
  const bf16_t* Wv; const bf16_t* bv; const bf16_t* Wo; const bf16_t* bo; const bf16_t* xres;
  bf16_t* Y;
  int B, T, S, Lq, dir, cpc;     // cpc = chunks per clip
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

// fragment of 16 rows x (K-slots of k-step pair kp, parity e) from a swizzled [rows][1024 B] image: lane (x, kg) -> row row0+x,
// logical 16-byte chunk 8*kp + 2*kg + e
__device__ __forceinline__ uint4 img_frag(const char* img, int row0, int kp, int e, int x, int kg) {
  return *reinterpret_cast<const uint4*>(img + (row0 + x) * 1024 + (((8 * kp + 2 * kg + e) ^ x) << 4));
}

// weight fragments of k-step pair kp for four 16-row tiles: lane (x, kg) holds 32 consecutive bytes of row 16*nt + x (the two
// parities), so that the four kg lanes of a row cover one whole 128-byte line
__device__ __forceinline__ void load_rows4(uint4 (&f)[2][4], const bf16_t* base, int kp) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    f[0][nt] = *reinterpret_cast<const uint4*>(base + nt * 16 * D + kp * 64);
    f[1][nt] = *reinterpret_cast<const uint4*>(base + nt * 16 * D + kp * 64 + 8);
  }
}
__device__ __forceinline__ void load_q(uint4 (&f)[2][2], const bf16_t* q0, const bf16_t* q1, int kp) {
  f[0][0] = *reinterpret_cast<const uint4*>(q0 + kp * 64);
  f[1][0] = *reinterpret_cast<const uint4*>(q0 + kp * 64 + 8);
  f[0][1] = *reinterpret_cast<const uint4*>(q1 + kp * 64);
  f[1][1] = *reinterpret_cast<const uint4*>(q1 + kp * 64 + 8);
}

template <int KT, int MT4>
__global__ __launch_bounds__(512, 2) void st1_fused_kernel(const St1F a) {
  constexpr int NG = MT / KT;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, kg = lane >> 4;
  const int T_ = a.T, S_ = a.S, Lq = a.Lq;
  const int G = a.dir == 0 ? S_ : T_, K = a.dir == 0 ? T_ : S_;
  // XCD-contiguous chunk order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous range of chunks, so
  // that the Qf rows of the few clips it works on stay in that L2
  const unsigned nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7u, qq = nwg >> 3, rr = nwg & 7u;
  const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int b = lid / a.cpc, g0 = (lid % a.cpc) * NG;
  const int ng = min(NG, G - g0);

  // ---- 0. X image -----------------------------------------------------------------------------------------
  {
    const bf16_t* vb = a.vft + (long)b * T_ * S_ * D;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = w * 16 + j, gl = r / (16 * KT), kk = r % (16 * KT);
      const bool valid = gl < ng && kk < K;
      const int t = a.dir == 0 ? kk : g0 + gl, s = a.dir == 0 ? g0 + gl : kk;
      const bf16_t* src = valid ? vb + ((long)t * S_ + s) * D + ((lane ^ j) << 3) : reinterpret_cast<const bf16_t*>(&g_zero_line);
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(smem + r * 1024), 16, 0, 0);
    }
  }
  // key mask bits of this lane's keys kk = 16*kt + 4*kg + r  ->  bit kt*4 + r  (1 = masked out); padding keys in `pad`
  unsigned mbits = 0, pad = 0;
  {
    const unsigned char* mk = a.kmask ? a.kmask + (long)b * K : nullptr;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 16 * kt + 4 * kg + r;
        if (kk >= K) pad |= 1u << (kt * 4 + r);
        else if (mk && mk[kk] == 0) mbits |= 1u << (kt * 4 + r);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 1. V_hh = X . W_v,hh^T ------------------------------------------------------------------------------
  uint4 vpk[MT / 2][4];
  {
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wv = a.Wv + (long)(w * 64 + x) * D + kg * 16;
    uint4 bw[2][4];
    load_rows4(bw, wv, 0);
#pragma unroll 1
    for (int kp = 0; kp < 8; ++kp) {
      uint4 bn[2][4];
      load_rows4(bn, wv, min(kp + 1, 7));          // next k-step pair in flight under this one's MFMAs
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const uint4 ax = img_frag(smem, mt * 16, kp, e, x, kg);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(ax, bw[e][nt], acc[mt][nt]);
        }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bw[e][nt] = bn[e][nt];
    }
    // acc[mt][nt][r] = V[key 16*mt + 4*kg + r][channel 16*nt + x]  ->  A fragments of step 3 (K-slots: tile 2p regs 0..3, tile 2p+1 regs 0..3)
#pragma unroll
    for (int p = 0; p < MT / 2; ++p)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        vpk[p][nt] = make_uint4(pack2(acc[2 * p][nt][0], acc[2 * p][nt][1]), pack2(acc[2 * p][nt][2], acc[2 * p][nt][3]),
                                pack2(acc[2 * p + 1][nt][0], acc[2 * p + 1][nt][1]), pack2(acc[2 * p + 1][nt][2], acc[2 * p + 1][nt][3]));
  }

  // ---- 2. S_hh^T = X . Qf_hh^T -----------------------------------------------------------------------------
  f32x4 sacc[MT][2];
  {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int it = 0; it < 2; ++it) sacc[mt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* q0 = a.qf + (((long)b * Lq + min(x, Lq - 1)) * H + w) * D + kg * 16;
    const bf16_t* q1 = a.qf + (((long)b * Lq + min(16 + x, Lq - 1)) * H + w) * D + kg * 16;
    uint4 bq[2][2];
    load_q(bq, q0, q1, 0);
#pragma unroll 1
    for (int kp = 0; kp < 8; ++kp) {
      uint4 bn[2][2];
      load_q(bn, q0, q1, min(kp + 1, 7));
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const uint4 ax = img_frag(smem, mt * 16, kp, e, x, kg);
#pragma unroll
          for (int it = 0; it < 2; ++it) sacc[mt][it] = mfma16(ax, bq[e][it], sacc[mt][it]);
        }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int it = 0; it < 2; ++it) bq[e][it] = bn[e][it];
    }
  }
  __syncthreads();          // every wave is done with the X image: it becomes the context image

  // ---- 3. softmax, O^T = V^T . P^T, context rows ---------------------------------------------------------------
#pragma unroll
  for (int gl = 0; gl < NG; ++gl) {
    if (gl >= ng) break;
    uint4 pf[2][KT / 2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float sv[KT][4];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sacc[gl * KT + kt][it][r];
          if (pad >> (kt * 4 + r) & 1u) v = -INFINITY;
          else if (mbits >> (kt * 4 + r) & 1u) v = MASK_FILL;
          sv[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float den = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sv[kt][r] = __expf(sv[kt][r] - mx); den += sv[kt][r]; }
      den += __shfl_xor(den, 16, 64);
      den += __shfl_xor(den, 32, 64);
      const float inv = 1.f / den;
#pragma unroll
      for (int k2 = 0; k2 < KT / 2; ++k2)
        pf[it][k2] = make_uint4(pack2(sv[2 * k2][0] * inv, sv[2 * k2][1] * inv), pack2(sv[2 * k2][2] * inv, sv[2 * k2][3] * inv),
                                pack2(sv[2 * k2 + 1][0] * inv, sv[2 * k2 + 1][1] * inv), pack2(sv[2 * k2 + 1][2] * inv, sv[2 * k2 + 1][3] * inv));
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int c0 = w * 64 + nt * 16 + kg * 4;                    // this lane's 4 consecutive output channels
      const uint2 bq = *reinterpret_cast<const uint2*>(a.bv + c0);
      const float bias[4] = {bf_lo(bq.x), bf_hi(bq.x), bf_lo(bq.y), bf_hi(bq.y)};
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < KT / 2; ++k2) o = mfma16(vpk[gl * (KT / 2) + k2][nt], pf[it][k2], o);
        const int i = it * 16 + x;
        if (i < Lq) {
          const int row = gl * Lq + i;
          char* dst = smem + row * 1024 + ((((c0 >> 3)) ^ (row & 15)) << 4) + (kg & 1) * 8;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(o[0] + bias[0], o[1] + bias[1]), pack2(o[2] + bias[2], o[3] + bias[3]));
        }
      }
    }
  }
  __syncthreads();

  // ---- 4. Y = ctx . W_o^T + b_o + x ----------------------------------------------------------------------------
  {
    f32x4 acc[MT4][4];
#pragma unroll
    for (int mt = 0; mt < MT4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wo = a.Wo + (long)(w * 64 + x) * D + kg * 16;
    uint4 aw[2][4];
    load_rows4(aw, wo, 0);
#pragma unroll 1
    for (int kp = 0; kp < 8; ++kp) {
      uint4 an[2][4];
      load_rows4(an, wo, min(kp + 1, 7));
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int mt = 0; mt < MT4; ++mt) {
          const uint4 cx = img_frag(smem, mt * 16, kp, e, x, kg);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(aw[e][nt], cx, acc[mt][nt]);
        }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) aw[e][nt] = an[e][nt];
    }
    // acc[mt][nt][r] = Y[row 16*mt + x][column 64*w + 16*nt + 4*kg + r]; after the swap a lane holds 8 consecutive columns
    const int cofs = (kg & 1) * 16 + (kg >> 1) * 8;
    const int rows = ng * Lq;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const int n0 = w * 64 + jp * 32 + cofs;
      const uint4 bq = *reinterpret_cast<const uint4*>(a.bo + n0);
#pragma unroll
      for (int mt = 0; mt < MT4; ++mt) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = acc[mt][2 * jp][r], q = acc[mt][2 * jp + 1][r];
          swap16(p, q);
          v[r] = p; v[4 + r] = q;
        }
        const int row = mt * 16 + x;
        if (row < rows) {
          const int gl = row / Lq, i = row - gl * Lq;
          const uint4 xr = *reinterpret_cast<const uint4*>(a.xres + ((long)b * Lq + i) * D + n0);
          const uint4 o = make_uint4(pack2(v[0] + bf_lo(bq.x) + bf_lo(xr.x), v[1] + bf_hi(bq.x) + bf_hi(xr.x)),
                                     pack2(v[2] + bf_lo(bq.y) + bf_lo(xr.y), v[3] + bf_hi(bq.y) + bf_hi(xr.y)),
                                     pack2(v[4] + bf_lo(bq.z) + bf_lo(xr.z), v[5] + bf_hi(bq.z) + bf_hi(xr.z)),
                                     pack2(v[6] + bf_lo(bq.w) + bf_lo(xr.w), v[7] + bf_hi(bq.w) + bf_hi(xr.w)));
          *reinterpret_cast<uint4*>(a.Y + (((long)b * G + g0 + gl) * Lq + i) * D + n0) = o;
        }
      }
    }
  }
}

template <int KT, int MT4>
int launch(const St1F& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&st1_fused_kernel<KT, MT4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MT * 16 * 1024) != hipSuccess) {
      bist_set_error("bist_st_stage1_fused_fwd: cannot reserve %d bytes of LDS", MT * 16 * 1024);
      return BIST_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((st1_fused_kernel<KT, MT4>), dim3((unsigned)(a.B * a.cpc)), dim3(512), MT * 16 * 1024, st, a);
  BIST_LAUNCH_CHECK("bist_st_stage1_fused_fwd");
  return BIST_OK;
}

}  // namespace

extern "C" int bist_st_stage1_fused_ok(int32_t T, int32_t S, int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype) {
  const int K = direction == 0 ? T : S;
  return dtype == BIST_BF16 && d == D && h == H && Lq >= 1 && Lq <= 32 && K >= 1 && K <= 128 && T >= 1 && S >= 1 &&
         (direction == 0 || direction == 1);
}

extern "C" int bist_st_stage1_fused_fwd(const void* qf, const void* vft, const uint8_t* kmask, const void* Wv, const void* bv,
                                        const void* Wo, const void* bo, const void* xres, void* Y, int32_t B, int32_t T, int32_t S,
                                        int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype, void* stream) {
  BIST_REQUIRE(qf && vft && Wv && bv && Wo && bo && xres && Y && B > 0, "bist_st_stage1_fused_fwd: null pointer or empty batch");
  BIST_REQUIRE(bist_st_stage1_fused_ok(T, S, Lq, d, h, direction, dtype),
               "bist_st_stage1_fused_fwd: shape outside the kernel's envelope (bf16, d=512, h=8, Lq<=32, keys<=128)");
  const void* ptrs[] = {qf, vft, Wv, bv, Wo, bo, xres, Y};
  for (const void* p : ptrs) BIST_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "bist_st_stage1_fused_fwd: operands must be 16-byte aligned");
  const int K = direction == 0 ? T : S, G = direction == 0 ? S : T;
  const int KT = K <= 32 ? 2 : K <= 64 ? 4 : 8, NG = MT / KT;
  St1F a{(const bf16_t*)qf, (const bf16_t*)vft, kmask, (const bf16_t*)Wv, (const bf16_t*)bv, (const bf16_t*)Wo, (const bf16_t*)bo,
         (const bf16_t*)xres, (bf16_t*)Y, B, T, S, Lq, direction, (G + NG - 1) / NG};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int need = (NG * Lq + 15) / 16;          // 16-row tiles of the output projection
  if (KT == 2) return need <= 5 ? launch<2, 5>(a, st) : launch<2, 8>(a, st);
  if (KT == 4) return need <= 3 ? launch<4, 3>(a, st) : launch<4, 4>(a, st);
  return launch<8, 2>(a, st);
}
