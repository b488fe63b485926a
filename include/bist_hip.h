/*
 * bist_hip.h -- C ABI of libbist_hip.so, the MI355X (gfx950) kernels behind the BiST
 * bi-directional spatio-temporal attention hot path.
 *
 * The reference (salesforce/BiST) has no native/FFI layer at all: its hot path is a chain of
 * stock PyTorch ops issued from Python.  Each entry point below therefore
 * cites the reference *Python* call site whose arithmetic it replaces; the Python host code in
 * bist_amd/ binds these with ctypes (see INTEGRATION.md for the binding a maintainer adds).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted;
 *   - the caller owns every buffer; the library never allocates or frees device memory;
 *   - every launcher is asynchronous on the hipStream_t passed as the last argument (opaque
 *     `void*` here so the header needs no HIP include) and never synchronises;
 *   - return value: 0 on success, a negative BIST_E* code otherwise; nothing throws or exits;
 *     bist_last_error() returns a thread-local message for the last failure;
 *   - dtype codes: BIST_F32 = 0 (exact fp32 path, parity gate), BIST_BF16 = 1 (bf16 storage,
 *     fp32 accumulate -- throughput path).
 */
#ifndef BIST_HIP_H
#define BIST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIST_OK 0
#define BIST_EINVAL (-1)   /* bad argument (shape/stride/alignment/dtype)            */
#define BIST_ELAUNCH (-2)  /* hipLaunchKernel reported an error                       */
#define BIST_ENODEV (-3)   /* no gfx950 device visible                                */

#define BIST_F32 0
#define BIST_BF16 1

#define BIST_ACT_NONE 0
#define BIST_ACT_RELU 1
#define BIST_ACT_GATE 2   /* y = residual[m,n] > 0 ? alpha*acc + bias : 0 -- the residual operand is a GATE, not an addend: the
                           * backward of y1 = dropout(relu(z)) fused into the product that computes dy1 (dh = dy.W2 of the
                           * feed-forward block, modules.py:112-113), with 1/(1-p) folded into alpha                          */

int bist_version(void);
const char* bist_last_error(void);
/* 1 when a HIP device is visible and is gfx950, else 0 (host-side probe, no kernel launch). */
int bist_device_ok(void);
/* Host-side counters of launches per attention-kernel family since the last reset (tests / bench report: WHICH implementation ran). */
#define BIST_K_ST1_MFMA_FWD 0   /* st1_mfma_kernel forward (bf16 stage-1 core on the matrix cores)          */
#define BIST_K_ST1_MFMA_BWD 1
#define BIST_K_ST1_VALU 2       /* st1_pv / st1_pv_bwd VALU kernels (fp32 path, shapes outside the envelope) */
#define BIST_K_ST2_MFMA_FWD 3
#define BIST_K_ST2_MFMA_BWD 4
#define BIST_K_ST2_VALU 5
#define BIST_K_MHA_FWD 6        /* mha_core forward (VALU)                                                   */
#define BIST_K_MHA_BWD_MFMA 7
#define BIST_K_MHA_BWD_VALU 8
#define BIST_K_ST1_FUSED 9      /* bist_st_stage1_fused_fwd                                                  */
#define BIST_K_DECSTACK 10      /* bist_decoder_stack_fwd (persistent decoder-stack kernel)                   */
#define BIST_K_ST1_FUSED_TRAIN 11   /* bist_st_stage1_fused_train_fwd                                        */
#define BIST_K_ST1_PBWD 12      /* bist_st_stage1_pv_bwd_p (backward core fed with the saved probabilities)  */
#define BIST_K_COUNT 13
int64_t bist_launch_count(int32_t family);
void bist_launch_count_reset(void);
/* Development hook: hand the fused stage-1 kernel (which = 0) or the persistent decoder kernel (which = 1) a caller-owned DEVICE
 * buffer for its in-kernel s_memtime stamps (8 / 128 uint64 per workgroup; see scripts/stamp_*.py), or NULL to switch them off.
 * Production never calls it; nothing is read from the environment.                                                           */
int bist_dev_set_stamps(int32_t which, void* device_buffer);
/* Development hook: a one-thread launch on `stream` that writes the device's 100 MHz wall clock to *slot (uint64, device memory).
 * Captured into a hipGraph it timestamps that point of that stream on every replay (bist_amd/stamps.py).  Production never calls it. */
int bist_dev_timestamp(void* slot, void* stream);

/* ------------------------------------------------------------------------------------------
 * GEMM  C[z] = epilogue( alpha * A[z] . B[z]^T )            (fp32 accumulate on the MFMA units)
 *
 * Replaces every nn.Linear / torch.matmul on the path: the Q/K/V/out projections of
 * MultiHeadedAttention.forward (model/modules.py:89-91,100), PositionwiseFeedForward
 * (modules.py:112-113), the video input projection VidEncoder8 (model/encoder.py:75) and the
 * vocabulary projection (model/generator.py:86).  Operands are addressed by element strides,
 *     A(m,k) = A[m*a_rs + k*a_ks],  B(n,k) = B[n*b_rs + k*b_ks],  C(m,n) = C[m*ldc + n]
 * so x.W^T (nn.Linear), x.W and x^T.y (the backward products) are all the same call.
 * Batch index z = z1*batch2 + z2 advances each operand by z1*?_bs1 + z2*?_bs2 elements.
 * Epilogue, in order: *alpha, +bias[n], activation, +residual, store as out_dtype.
 * The residual row of output row m is (m / res_outer)*res_inner + (m % res_inner): with
 * res_outer = G*Lq, res_inner = Lq this adds the un-expanded query row to each of G groups,
 * which is the "x + sublayer(x)" of SublayerConnection (modules.py:44) applied to the
 * expanded query of temporal2spatial/spatial2temporal (encoder.py:114-121,145-148) without
 * materialising the expansion.  res_outer = res_inner = 0 means the plain row m.
 * ------------------------------------------------------------------------------------------ */
typedef struct BistGemm {
  const void* A;
  const void* B;
  void* C;
  const void* bias;      /* [N] in in_dtype precision class (f32 for F32, bf16 for BF16) or NULL */
  const void* residual;  /* out_dtype elements, leading dimension ldr, or NULL                 */
  int32_t M, N, K;
  int64_t a_rs, a_ks, b_rs, b_ks, ldc, ldr;
  int32_t batch1, batch2;
  int64_t a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2, r_bs1, r_bs2, bias_bs2;
  float alpha;
  int32_t act;
  int32_t res_outer, res_inner;
  int32_t in_dtype, out_dtype;
  /* inverted dropout applied after the activation and before the residual add
   * (SublayerConnection's dropout, modules.py:44): keep-probability 1-drop_p, counter-based
   * mask keyed by (drop_seed, element index) so backward can regenerate it. 0 disables.      */
  float drop_p;
  uint64_t drop_seed;
  /* Optional device-side step counter mixed into the seed (seed + ctr[0]*odd constant): lets a
   * captured hipGraph draw a fresh mask on every replay.  NULL = seed only.                     */
  const uint64_t* drop_ctr;
  /* Optional split-K scratch (caller-owned device memory, fp32 partial tiles).  When given and the
   * problem has few output tiles and a long K (weight gradients), K is cut over several workgroups
   * and a second kernel sums the slabs; NULL / 0 disables split-K.  The first 4 KiB hold ticket counters
   * of the in-launch combine: the buffer must be ZERO when first handed over (the library returns the
   * counters to zero after every call) and must not be shared by launches that may run concurrently.   */
  void* workspace;
  int64_t workspace_bytes;
  /* Kernel selection: 0 = automatic.  BIST_GEMM_TILE256 asks for the 256x256-tile deep-pipelined kernel
   * (bf16, both operands K-contiguous, K a multiple of 64 and >= 128, unbatched or batched) wherever it is legal; it is
   * chosen automatically for such products of >= 140 tiles and K >= 512 (the M = B*T*S products of the path from B = 12:
   * P0 56 vs 78 us against the 128-tile kernel), with 160-row tiles when M pads better to 160 than to 256.            */
  int32_t hint;
  int32_t reserved;
  /* Optional LayerNorm PROLOGUE (all NULL / 0 = none): the product is taken of LayerNorm(A) = ln_gain * (a - mean) / (std + ln_eps)
   * + ln_offset per row (unbiased std over the K channels, eps outside the root: LayerNorm.forward, model/modules.py:28-31), i.e. the
   * `sublayer(self.norm(x))` of SublayerConnection (modules.py:44) with the norm folded into the sublayer's first projection instead
   * of a launch of its own.  ln_out (nullable) receives the normalised rows [M][ln_ld] in in_dtype (the weight-gradient product of
   * the backward pass reads them).  Envelope (bist_gemm_ln_ok): bf16, K = 512, A rows K-contiguous, unbatched, N a multiple of 64
   * and >= 512, at most 256 tiles of 64x64; outside it bist_gemm returns BIST_EINVAL (the caller runs bist_layernorm_fwd first). */
  const void* ln_gain;
  const void* ln_offset;
  void* ln_out;
  int64_t ln_ld;
  float ln_eps;
  /* ln_mode = 1: the LayerNorm is applied to the OUTPUT rows instead -- C = LayerNorm(act(alpha A.B^T + bias)) over the N = 512
   * columns, the `self.in_norm(F.relu(self.W(fts)))` of VidEncoder8 (model/encoder.py:75-81) as one launch (256-tile kernel: the
   * two column-tile workgroups of a row block exchange their row statistics through `workspace`).  Envelope (bist_gemm_ln_ok): bf16,
   * N = 512, M a multiple of 256, K-contiguous operands, no dropout / residual, workspace >= 4 KiB + M * 16 bytes; ln_out unused. */
  int32_t ln_mode;
  /* bias stride (elements) of the OUTER batch index z1 (bias_bs2 above is the inner one's): a product batched over two sets of
   * weights -- the t2s and s2t instances of one sublayer (encoder.py:176,184: attn[0] / attn[3], ...), run as one launch -- reads
   * set z1's bias at bias + z1 * bias_bs1.  (Appended in 0.1.1; 0 = one bias for every z1.)                                    */
  int64_t bias_bs1;
} BistGemm;
int bist_gemm_ln_ok(const BistGemm* g);       /* 1 if bist_gemm(g) will run g's LayerNorm prologue */
#define BIST_GEMM_TILE256 2
#define BIST_GEMM_SPLIT64 4   /* hint bit: cut a long K of a small 64-tile product over neighbouring workgroups and combine
                               * them inside the launch (agent-scope release / ticket / acquire); measured slower than the
                               * unsplit ring kernel on this model's shapes, so never chosen automatically              */

int bist_gemm(const BistGemm* g, void* stream);
/* Two independent products in one call.  When they are the backward pair of a linear layer on few rows -- a: dX = dZ.W
 * (A K-contiguous, B row-contiguous), b: dW = dZ^T.X (both row-contiguous), each too small to fill the chip on its own --
 * they share ONE launch (the first workgroups work on a, the rest on b); otherwise this is bist_gemm(a) then
 * bist_gemm(b).  Replaces the two addmm calls of nn.Linear's backward (modules.py:89-91,100 under autograd).      */
int bist_gemm_pair(const BistGemm* a, const BistGemm* b, void* stream);
/* Which kernel bist_gemm would pick for this problem: 1 = LDS-DMA MFMA tile kernel (128- or 64-tile),
 * 2 = the same with split-K, 3 = skinny (one side <= 8) VALU kernel, 4 = the 256x256-tile kernel (K-contiguous bf16
 * products of 140+ such tiles, or hint BIST_GEMM_TILE256), 0 = generic strided kernel
 * (host-side query, used by tests and the bench report).                                          */
int bist_gemm_is_fast(const BistGemm* g);

/* ------------------------------------------------------------------------------------------
 * LayerNorm of the reference:  y = a * (x - mean) / (std_unbiased + eps) + b   per row
 * (model/modules.py:28-31 -- N-1 variance, eps added OUTSIDE the square root).
 * x,y [rows, d] with row strides ldx/ldy; a,b [d] in the same dtype as x.
 * ------------------------------------------------------------------------------------------ */
int bist_layernorm_fwd(const void* x, const void* a, const void* b, void* y, int64_t rows, int32_t d,
                       int64_t ldx, int64_t ldy, float eps, int32_t dtype, void* stream);
/* Up to 8 LayerNorms of ONE geometry (rows, d, strides) in one launch: set i normalises rows x_i with (a_i, b_i) into y_i.  The t2s and
 * s2t instances of a VidEncoderLayer4 sublayer (encoder.py:176 / :184, :121 / :148, ...) normalise same-shaped query tensors with
 * different parameters: run together, the two directions' query-side chains are ONE sequence of launches instead of two.          */
typedef struct BistLnSet { const void* x; const void* a; const void* b; void* y; } BistLnSet;
int bist_layernorm_fwd_multi(const BistLnSet* sets, int32_t nsets, int64_t rows, int32_t d, int64_t ldx, int64_t ldy, float eps,
                             int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dropout of attention probabilities in training (modules.py:62-63: p_attn = dropout(p_attn) after the
 * softmax, before p_attn.V).  NULL or p == 0 turns it off.  The keep decision of probability element
 * idx is a pure function of (seed, *ctr, idx) -- a counter-based mask regenerated by the backward
 * kernels, never stored; ctr (nullable) is a device step counter so that a captured hipGraph draws
 * fresh masks on every replay.  idx is the element's linear index in the order stated per entry point.
 * ------------------------------------------------------------------------------------------ */
typedef struct BistDrop {
  float p;                 /* drop probability, 0 <= p < 1 */
  uint64_t seed;
  const uint64_t* ctr;     /* device pointer or NULL */
} BistDrop;

/* ------------------------------------------------------------------------------------------
 * Small multi-head attention core (after the projections):
 *     O[n,i,hh*dk+c] = sum_j softmax_j( Q[n,i,hh,:].K[n,j,hh,:] * scale  masked -1e9 ) V[n,j,hh,c]
 * = attention() model/modules.py:54-64 for every head of MultiHeadedAttention (modules.py:94).
 * Q [N,Lq,*] K,V [N,Lk,*] with row strides ldq/ldk/ldv (elements) and per-n strides
 * q_bs/k_bs/v_bs, heads side by side in the last dim (h*dk columns).  mask is uint8 (torch.bool)
 * with strides (mask_bs per n, mask_qs per query row; 0 broadcasts) or NULL; masked scores are
 * REPLACED by -1e9 (not -inf) exactly as the reference does, so a fully masked row is uniform.
 * p_attn (nullable) receives the probabilities as f32 [N,h,Lq,Lk] -- the `.attn` side channel
 * the pointer generator reads (model/generator.py:109-110).
 * Used for the query self-attentions A0/A3 (encoder.py:176,184), CapEncoderLayer
 * (encoder.py:214-215), MultimodalDecoderLayer12 (decoder.py:21-58) and the pointer attentions.
 * drop: mask index ((n*h + hh)*Lq + i)*Lk + j; p_attn receives the probabilities BEFORE dropout.
 * ------------------------------------------------------------------------------------------ */
int bist_mha_core_fwd(const void* Q, const void* K, const void* V, const uint8_t* mask, void* O, float* p_attn,
                      int32_t N, int32_t Lq, int32_t Lk, int32_t h, int32_t dk,
                      int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                      int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                      int64_t mask_bs, int64_t mask_qs, float scale, const BistDrop* drop, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage 1 of temporal->spatial (direction 0, encoder.py:110-123) and spatial->temporal
 * (direction 1, encoder.py:142-150), after the score GEMM.
 *
 * scores [B, Lq*h, T*S] (f32 or bf16 = sc_dtype) hold  Qf[b,(i,hh),:] . vft[b,t,s,:]  where Qf is
 * the query folded through W_k and pre-scaled by 1/sqrt(dk) (the key bias cancels in the
 * softmax), row order r = i*h + hh.  V [B,T,S,*] is the value projection (row stride ldv).
 *   direction 0: softmax over t for every (b,s), masked by tmask[b,t] (uint8, nullable);
 *                O[b,s,i,hh*dk+c] = sum_t P[b,i,hh,t,s] V[b,t,s,hh*dk+c]        O: [B,S,Lq,d]
 *   direction 1: softmax over s for every (b,t), masked by tmask[b,s] if given (s2t itself has no mask here; the t2s step
 *                on (s,t)-ordered tensors -- bist_permute_ts -- is this form with T and S exchanged and the frame mask);
 *                O[b,t,i,hh*dk+c] = sum_s P[b,i,hh,t,s] V[b,t,s,hh*dk+c]        O: [B,T,Lq,d]
 * which is attention() (modules.py:54-64) of every (b,s) / (b,t) group without materialising
 * K, the permuted video tensor or the expanded query.
 * drop: mask index ((((b*G + g)*h + hh)*Lq + i)*Kn + k), g the group (s or t), k the key (t or s).
 * ------------------------------------------------------------------------------------------ */
int bist_st_stage1_pv_fwd(const void* scores, const void* V, const uint8_t* tmask, void* O,
                          int32_t B, int32_t T, int32_t S, int32_t Lq, int32_t h, int32_t dk,
                          int64_t ldv, int32_t direction, const BistDrop* drop, int32_t sc_dtype, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage 1 of one direction as ONE launch (forward, inference form: no dropout, nothing saved for a backward pass):
 * replaces, for every group g of every clip, the whole SublayerConnection of encoder.py:121 / :148
 *     Y[b,g,i,:] = x[b,i,:] + W_o . concat_h softmax_k( Qf[b,(i,hh),:] . X_g[k,:] (masked) ) (X_g W_v,hh^T + b_v,hh) + b_o
 * i.e. MultiHeadedAttention.forward (modules.py:81-100) + attention() (modules.py:54-64) + the residual on the expanded query
 * (modules.py:44), with X_g = vft[b,:,g,:] (direction 0: keys = frames, kmask[b,t] uint8 nullable, -1e9 REPLACES a masked score)
 * or vft[b,g,:,:] (direction 1: keys = regions, kmask[b,s] normally NULL).  Qf [B, Lq*h, d] (row i*h + hh) is the query
 * folded through W_k and pre-scaled by 1/sqrt(dk), exactly the operand of the score product feeding bist_st_stage1_pv_fwd;
 * Wv / Wo are the [d,d] nn.Linear weights (row = output channel) of linears[2] / linears[3] IN FRAGMENT ORDER (bist_pack_frag_rows,
 * done once per set of weights), bv / bo their biases; xres [B,Lq,d] is the un-expanded residual; Y [B,G,Lq,d] (G = S for
 * direction 0, T for direction 1).  All operands bf16, 16-byte aligned.
 * The value projection, the scores, the probabilities and the head-concatenated context never reach HBM.
 * bist_st_stage1_fused_ok: 1 when the shape is inside the kernel's envelope (bf16, d = 512, h = 8, Lq <= 32, keys <= 128).
 * ------------------------------------------------------------------------------------------ */
/* W [rows][cols] (row-major bf16, rows % 16 == 0, cols % 64 == 0) -> out, same size, in MFMA-fragment order
 * [rows/16][cols/64][2][64][8]: the 16 bytes lane (x, kg) feeds a v_mfma_f32_16x16x32_bf16 for row tile nt, k-step pair kp, parity
 * e are W[16*nt + x][64*kp + 16*kg + 8*e .. +7], and one wave load of a block reads 1 KiB contiguously.                       */
int bist_pack_frag_rows(const void* W, void* out, int32_t rows, int32_t cols, int32_t dtype, void* stream);
/* n <= 32 matrices of one size in ONE launch (host arrays of device pointers): a training step re-packs the value / output projection
 * weights of every reasoning layer after the optimiser moved them.                                                             */
int bist_pack_frag_rows_multi(const void* const* Ws, void* const* outs, int32_t n, int32_t rows, int32_t cols, int32_t dtype, void* stream);
int bist_st_stage1_fused_ok(int32_t T, int32_t S, int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype);
int bist_st_stage1_fused_fwd(const void* qf, const void* vft, const uint8_t* kmask, const void* Wv, const void* bv,
                             const void* Wo, const void* bo, const void* xres, void* Y, int32_t B, int32_t T, int32_t S,
                             int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype, void* stream);
/* The TRAINING form of the same launch: the sublayer's two dropouts are applied in the kernel -- attn_drop on the probabilities
 * (modules.py:62-63; mask index ((((b*G + g)*h + hh)*Lq + i)*K + key, as bist_st_stage1_pv_fwd; the value bias is then scaled by the
 * kept probabilities' row sum) and sub_drop on W_o ctx + b_o before the residual (modules.py:44; mask index (row of Y)*d + column,
 * as bist_gemm's epilogue) -- and what the backward pass needs leaves as side outputs:
 *   Vout [B,T,S,d]        V = X W_v^T + b_v (the operand of bist_st_stage1_pv_bwd_p and of the value projection's weight gradient);
 *                         NULL when the caller runs the value projection as a product of its own (on another stream, off the
 *                         critical chain of the direction) and only wants the probabilities and the context rows from here,
 *   Pout [B,G,h,Lq,KP]    f32 probabilities BEFORE dropout, KP = K rounded up to a multiple of 4 (padding zero),
 *   Oout [B,G,Lq,d]       the head-concatenated context (the output projection's input).
 * Against the unfused training forward (value GEMM, score GEMM, softmax + P.V core, output projection: four launches, fp32 scores
 * written and re-read) this is one launch and no score tensor.  bist_st_stage1_fused_train_ok: the envelope of the inference form
 * and at most 96 context rows per workgroup (groups per workgroup x Lq; Lq <= 24 at <= 32 keys).                              */
int bist_st_stage1_fused_train_ok(int32_t T, int32_t S, int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype);
int bist_st_stage1_fused_train_fwd(const void* qf, const void* vft, const uint8_t* kmask, const void* Wv, const void* bv,
                                   const void* Wo, const void* bo, const void* xres, void* Y, void* Vout, float* Pout, void* Oout,
                                   const BistDrop* attn_drop, const BistDrop* sub_drop, int32_t B, int32_t T, int32_t S,
                                   int32_t Lq, int32_t d, int32_t h, int32_t direction, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Beam-search bookkeeping of one decode step ON THE DEVICE (model/decode.py:59-99; the reference copies every hypothesis's
 * log-prob row to the host -- decode.py:71, a synchronisation per hypothesis and step -- and runs numpy argsort + the beam update in
 * Python).  Two small launches: per row the beam + 2 largest entries of lp_vec = logp[row] + lp[row] (decode.py:72) and lp_vec[eos];
 * then decode.py:74-97 literally on one thread (completed-hypothesis scores from min_len on; rows in order, candidates descending,
 * replace-the-minimum with the first minimal index, early break), which rewrites IN PLACE the next step's inputs -- tok [beam] int64
 * tokens, lp [beam] running scores, mask64 [beam,64] ancestry masks over the decoder kernel's self-attention slots (hypothesis j of
 * the next step owns slot slot0_next + j), mask_out [beam,LkS_next] the same mask in the width the next launch reads (LkS_next = 0:
 * last step) -- and the step's record rec_*[step*beam + j] (parent row, token, score; rec_comp[step*beam + row] completed score,
 * rec_n[step] rows) from which the host rebuilds the token lists once per turn.  flag (sticky int32): 1 = the beam + 3 largest values
 * of a row are not all distinct (numpy orders ties by its sort's internals), 2 = NaN, 4 = fewer than beam survivors: the caller
 * repeats the turn on the host path, so the n-best lists are the reference's in every case.  n <= beam <= 14, V <= 4096.
 * ------------------------------------------------------------------------------------------ */
int bist_beam_step(const float* logp, float* lp, int64_t* tok, uint8_t* mask64, uint8_t* mask_out, float* cand_val, int32_t* cand_idx,
                   float* eos_val, int32_t* rec_parent, int32_t* rec_token, float* rec_score, float* rec_comp, int32_t* rec_n,
                   int32_t* flag, int32_t n, int32_t V, int32_t beam, int32_t step, int32_t min_len, int32_t unk, int32_t eos,
                   int32_t dec_eos, int32_t slot0_next, int32_t LkS_next, float penalty, void* stream);

/* ------------------------------------------------------------------------------------------
 * The response-decoder stack of one beam-search step as ONE persistent launch (inference; bf16, d = 512, h = 8, R <= 64 rows).
 * Replaces, for all layers of MultimodalDecoder8's loop (model/decoder.py:114-182, reasoning results cached per turn), the
 * MultimodalDecoderLayer12.forward of decoder.py:20-60 with enc_vc_combine != 'none': causal self-attention, attention to the
 * history, to the query and to the fused modalities (each x + W_o MHA(LN(x), mem, mem, mask) + b_o; modules.py:42-44, 54-64,
 * 81-100) and the feed-forward block (modules.py:112-113), on the R = hypotheses x prefix-length rows of a decode step
 * (model/decode.py:62-66).  One unfused step is ~120 launches of 5-17 us on these rows; here 32 resident workgroups walk the
 * 14 phases of a layer (per attention sublayer: LN + projection, one (head, 16-row tile) core unit per workgroup, output
 * projection + residual; two for the feed-forward block) with a grid barrier after each and prefetch the next product's weights
 * across it; hand-offs are write-through stores + sc1 loads, no cache fences.
 *   layers_dev  device array of n_layers BistDecLayer (below): LayerNorm gains/offsets, nn.Linear weights [out][in] and biases,
 *               and per memory c = 0 (history), 1 (query), 2 (fused modalities) the keys K = mem W_k^T + b_k as [LkP][512] and
 *               the values TRANSPOSED, V^T [512][LkP] (LkP = Lk rounded up to 32, 64, 128, 256 or 512, padding zero), computed once per turn
 *               (they do not depend on the prefix), with the key mask [LkP] (uint8, 1 = attend; -1e9 REPLACES a masked score)
 *   x_in [R][512] the embedded rows of this call; xbuf0 / xbuf1 [64][512], qbuf [64][512], hbuf [64][2048]: caller-owned scratch, ZERO
 *   when first handed over.  kcache / vcache [n_layers][64][512]: the self-attention keys / values (row-major) of every row the
 *   stack has seen in this turn, one 64-slot pool per layer; this call's row r is written to slot slot0 + r, and self_mask [R][LkS]
 *   (uint8, 1 = attend) says which slots < slot0 + R a row attends (LkS = 32 or 64 >= slot0 + R; -1e9 REPLACES a masked score).
 *     - all prefix rows in one call (decode.py:62-66 as written): slot0 = 0, R = hypotheses x prefix length, self_mask = the
 *       block-diagonal causal mask of data/dataset.py:101-105;
 *     - one decode step at a time: R = hypotheses (only the NEW position's rows), slot0 = the first free slot, self_mask row j = the
 *       slots of hypothesis j's ancestors and its own -- the keys / values of earlier positions are the ones earlier calls left in
 *       the pool (row-wise the same arithmetic: the decoder is causal).
 *   sync: 32 bytes = 8 words the caller zeroes ONCE: words 0, 1
 *   the barrier's arrival / exit counters (the kernel leaves them zero for the next call -- no memset node per call), word 4 a STICKY
 *   error flag to read after a turn: non-zero = a barrier timed out (the 32 workgroups were not co-resident), the results of that call are invalid.
 *   lk_pad_max: the largest LkP among the descriptors' memories (32, 64, 128, 256 or 512); above 64 the launch is the instance whose attention core
 *   walks a memory in 64-key chunks (dialogue histories of 65 .. 512 tokens).
 * Result: the rows of xbuf[(5 * n_layers - 1) % 2] (the residual stream ping-pongs between the two buffers, five writes per layer).  Returns BIST_EINVAL outside the envelope (bist_decoder_stack_ok).
 * ------------------------------------------------------------------------------------------ */
typedef struct BistDecLayer {
  const void* ln_a[5]; const void* ln_b[5];
  const void* Wqkv; const void* bqkv;
  const void* Wq[3]; const void* bq[3];
  const void* Wo[4]; const void* bo[4];
  const void* Kc[3]; const void* VTc[3];
  const uint8_t* cmask[3];
  const void* W1; const void* b1; const void* W2; const void* b2;
  int32_t Lk[3]; int32_t LkP[3];
} BistDecLayer;
int bist_decoder_stack_ok(int32_t R, int32_t d, int32_t h, int32_t Lk_max, int32_t dtype);
int64_t bist_decoder_layer_desc_bytes(void);
/* The per-turn key / value caches the persistent decoder kernel reads (BistDecLayer.Kc / VTc): job j takes the packed projection of one
 * memory for one layer, rows t < Lk of [k(512) | v(512)] at src + t*ld (MultiHeadedAttention.linears[1], [2] of decoder.py:42-55 on the
 * encoded history / query / fused modalities), copies the keys to K [Lk][512] and writes the values transposed to VT [512][LkP] with
 * columns Lk..LkP-1 zero.  bf16; jobs is a HOST array of 1..32 entries.                                                        */
typedef struct BistKvFill { const void* src; void* K; void* VT; int32_t Lk; int32_t LkP; int64_t ld; } BistKvFill;
int bist_decoder_cache_fill(const BistKvFill* jobs, int32_t n_jobs, int32_t dtype, void* stream);
/* pbuf (nullable): f32 [2][8][16][512] scratch (256 KiB).  With it and R <= 16 rows (one decode step at a time) the launch takes the
 * HEAD-LOCAL form: workgroup hh < 8 owns head hh through a whole attention sublayer (LayerNorm, its 64 query columns -- and the new
 * rows' key / value columns of the self-attention --, the core, and a PARTIAL output projection over its 64 context columns, K = 64,
 * f32 into pbuf); the sum over the heads, the bias and the residual are taken at the head of the next phase by every workgroup.  One
 * grid barrier per attention sublayer instead of three: 6 per layer instead of 14.                                                */
int bist_decoder_stack_fwd(const void* layers_dev, int32_t n_layers, const void* x_in, void* xbuf0, void* xbuf1, void* qbuf,
                           void* kcache, void* vcache, void* hbuf, const uint8_t* self_mask, int32_t R, int32_t LkS, int32_t slot0,
                           int32_t lk_pad_max, void* sync, float* pbuf, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage 2 of both directions (encoder.py:125-134 / 152-165): query position (b,i) attends,
 * alone, over the G stage-1 outputs Y[b,g,i,:] of its own position (G = S for t2s, T for s2t).
 * With the query folded through W_k (q2f [B,Lq,h,d], pre-scaled) and the value projection
 * applied after the weighted sum, the kernel needs only Y:
 *     sc[hh,g] = q2f[b,i,hh,:] . Y[b,g,i,:]   (masked -1e9 by gmask[b,g] if given)
 *     PY[b,i,hh,:] = sum_g softmax_g(sc)[hh,g] * Y[b,g,i,:]                    PY: [B,Lq,h,d]
 * drop: mask index ((b*Lq + i)*h + hh)*G + g.  With dropout the kept probabilities no longer sum to one,
 * and the value bias applied after the weighted sum must be scaled by their sum:
 *     P'(Y W_v^T + b_v) = (P'Y) W_v^T + rowsum(P') b_v
 * rowsum (f32 [B,Lq,h], nullable without dropout) receives sum_g P'[b,i,hh,g] (bist_scaled_bias_fwd).
 * ------------------------------------------------------------------------------------------ */
int bist_st_stage2_fwd(const void* q2f, const void* Y, const uint8_t* gmask, void* PY, float* rowsum,
                       int32_t B, int32_t G, int32_t Lq, int32_t h, int32_t d, const BistDrop* drop, int32_t dtype, void* stream);
/* y[m, hh*dk + c] = x[m, hh*dk + c] + s[m, hh] * bias[hh*dk + c]   (in place allowed: y == x)            */
int bist_scaled_bias_fwd(const void* x, const float* s, const void* bias, void* y, int64_t M, int32_t h, int32_t dk,
                         int32_t dtype, void* stream);
/* ds[m, hh] = sum_c dy[m, hh*dk+c] bias[hh*dk+c];  dbias[hh*dk+c] += sum_m s[m, hh] dy[m, hh*dk+c]  (f32 acc) */
int bist_scaled_bias_bwd(const void* dy, const float* s, const void* bias, float* ds, float* dbias, int64_t M, int32_t h,
                         int32_t dk, int32_t dtype, void* stream);
/* The same over `nsets` stacked row blocks of M / nsets rows: block z reads bias + z * bias_zs (and adds into dbias + z * dbias_zs) --
 * the two directions' stage-2 value biases (attn[2] / attn[5]) in one launch.                                                     */
int bist_scaled_bias_fwd_z(const void* x, const float* s, const void* bias, void* y, int64_t M, int32_t h, int32_t dk,
                           int32_t nsets, int64_t bias_zs, int32_t dtype, void* stream);
int bist_scaled_bias_bwd_z(const void* dy, const float* s, const void* bias, float* ds, float* dbias, int64_t M, int32_t h,
                           int32_t dk, int32_t nsets, int64_t bias_zs, int64_t dbias_zs, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise helpers on the path.
 * ------------------------------------------------------------------------------------------ */
/* y[b,l,:] = lut[ids[b,l],:]*sqrt(d) + pe[l,:]   Embeddings + PositionalEncoding
 * (modules.py:121-123,141-144); y and lut in dtype, ids int64 [rows = B*L], pe the f32
 * sinusoid table [>=L, d] (the reference's registered buffer `pe`, modules.py:131-139).
 * drop: the dropout after the position is added (modules.py:144), mask index row*d + c.        */
int bist_embed_pe_fwd(const int64_t* ids, const void* lut, const float* pe, void* y, int64_t rows, int32_t L,
                      int32_t d, const BistDrop* drop, int32_t dtype, void* stream);

/* temporal_mask[b,t] = any(fts[b,t,:,:] != 0) computed as sum != 0 exactly like
 * data/dataset.py:79 ((fts.sum(2).sum(-1) != 0)); fts [B,T,S*C] in dtype, out uint8 [B,T].     */
int bist_temporal_mask(const void* fts, uint8_t* mask, int64_t BT, int64_t row_elems, int32_t dtype, void* stream);

/* out = sum_j softmax_j(score[row,:n])[perm] * x_j[row,:]  -- the dynamic modality fusion of
 * MultimodalDecoder8.forward (decoder.py:155-159): score [rows,n] f32/bf16, xs = n pointers
 * (host array of device pointers, n <= 4) each [rows,d]; out [rows,d].                         */
int bist_fuse_modalities(const void* score, const void* const* xs, void* out, int64_t rows, int32_t n, int32_t d,
                         int32_t dtype, void* stream);

/* out[i] = a[i] + b[i % nb]: the residual add of SublayerConnection (modules.py:44) when it is
 * not fused into a GEMM epilogue; nb < n broadcasts b.                                          */
/* The generic forms of the reference's two dropout-carrying adds (training mode), for callers that use the modules one by one
 * instead of the fused layer classes:  mode 0: out = a + dropout(b)  (SublayerConnection.forward, modules.py:42-44),
 * mode 1: out = dropout(a + b)  (PositionalEncoding.forward, modules.py:142-144); b is broadcast with period nb.  The mask of
 * element i is the counter-based drop_keep((seed, ctr), i), so bist_epilogue_bwd on the same contiguous [M,N] view is its backward. */
int bist_add_dropout_fwd(const void* a, const void* b, void* out, int64_t n, int64_t nb, int32_t mode, const BistDrop* drop,
                         int32_t dtype, void* stream);
int bist_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t nb, int32_t dtype, void* stream);

/* y[b,s,t,:] = x[b,t,s,:]  (x [B,T,S,d] contiguous): the video tensor in region-major order.  The t2s direction walks, for
 * every region s, its T frames; on [B,T,S,d] those are rows S*d apart and the per-group score gather / gradient scatter are
 * 16-byte pieces, on the permuted copy (made once per step, shared by all layers) they are contiguous runs.              */
int bist_permute_ts(const void* x, void* y, int32_t B, int32_t T, int32_t S, int32_t d, int32_t dtype, void* stream);

/* out[i] = sum_j srcs[j][i] for n <= BIST_ADD_N_MAX contiguous tensors of `numel` elements (fp32 accumulation): the
 * gradient of a tensor with n consumers (the video tensor feeds 3 products in each of the L reasoning layers) in ONE
 * pass, where autograd's AccumulateGrad chain (what train.py:32 `loss.backward()` does under PyTorch) makes n-1
 * pairwise passes.  srcs = host array of device pointers; out may alias srcs[0].                                  */
#define BIST_ADD_N_MAX 24
int bist_add_n(const void* const* srcs, int32_t n, void* out, int64_t numel, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Output heads ("next" row (f)-2 of SURVEY.md section 8; needed for the logits parity gate).
 * ------------------------------------------------------------------------------------------ */
/* MultiPointerGenerator.forward (model/generator.py:84-127) / PointerGenerator (:36-75) after the
 * projection GEMMs:  out[row,v] = log( sw[n]*softmax(logits[row])[v]
 *                                      + sum_j sw[j] * sum_{t: text_j[b,t]==v} p_j[row,t] ),  b = row / Lt,
 * sw = softmax(switch_logits[row,:n+1]) or, for one source with sigmoid_switch, (1-sig, sig).
 * logits [rows,V] f32, p_j [rows,L_j] f32 (the pointer attentions' `.attn`), text_j int64 [B,L_j];
 * ptr_p / ptr_text / ptr_len are HOST arrays of n_ptr entries.                                   */
int bist_pointer_mix_fwd(const float* logits, const float* switch_logits, int32_t n_ptr, const float* const* ptr_p,
                         const int64_t* const* ptr_text, const int32_t* ptr_len, float* out, int64_t rows, int32_t Lt,
                         int32_t V, int32_t sigmoid_switch, void* stream);

/* Generator.forward (model/generator.py:21-27): y = log_softmax(x) over V, f32 [rows,V].        */
int bist_log_softmax_fwd(const float* x, float* y, int64_t rows, int32_t V, void* stream);

/* LabelSmoothing.forward (model/label_smoothing.py:20-30): per-row KL divergence (sum reduction)
 * of logp [rows,V] against the smoothed one-hot of target[rows] (pad rows and the pad column
 * carry no mass).                                                                                */
int bist_label_smoothing_fwd(const float* logp, const int64_t* target, float* row_loss, int64_t rows, int32_t V,
                             float smoothing, int32_t pad, void* stream);

/* out[0] (+)= sum(x[0..n)) / denom[0]  (denom: device int64 count or NULL = 1); one workgroup,
 * fixed summation order.  The `/ norm` of SimpleLossCompute (model/optimize.py:50).               */
int bist_sum_div(const float* x, int64_t n, const int64_t* denom, float* out, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward kernels (training step: loss.backward() of model/optimize.py:85 without autograd's
 * aten kernels).  Probabilities are recomputed from the saved inputs; reductions over rows
 * accumulate into caller-zeroed fp32 buffers.
 * ------------------------------------------------------------------------------------------ */
/* dz = dy * d(epilogue)/dz of bist_gemm's epilogue (relu via the stored output y, dropout via the
 * regenerated mask of (drop_seed, element index)).                                              */
int bist_epilogue_bwd(const void* dy, const void* y, void* dz, int64_t M, int32_t N, int64_t lddy, int64_t ldy,
                      int64_t lddz, int32_t act, float drop_p, uint64_t drop_seed, const uint64_t* drop_ctr, int32_t dtype,
                      void* stream);
/* out[b, r] = sum_g x[b, g, r]  (r < inner): gradient of the un-expanded query of stage 1.       */
int bist_group_sum(const void* x, void* out, int64_t B, int32_t G, int64_t inner, int32_t dtype, void* stream);
/* out[b, r] = add[b, r] + sum_g x[b, g, r]: the same with a second gradient of the un-expanded query (the one that arrives through its
 * next consumer, stage 2's sublayer) folded in, so that autograd launches no accumulation of its own.  add nullable.          */
int bist_group_sum_add(const void* x, const void* add, void* out, int64_t B, int32_t G, int64_t inner, int32_t dtype, void* stream);
/* The same pass also writes dz[b, g, r] = dropout-mask(drop) * x[b, g, r] / (1 - p), mask index = the element's flat index: with x = dY of
 * the stage-1 sublayer (Y = query expanded over the groups + drop(W_o ctx + b_o), encoder.py:121,148 + modules.py:44) these are the
 * gradient of the un-expanded query and the masked gradient the output projection's backward products read -- dY is read once.   */
int bist_group_sum_mask(const void* x, const void* add, void* out, void* dz, int64_t B, int32_t G, int64_t inner, const BistDrop* drop,
                        int32_t dtype, void* stream);
/* out[n] += sum_m x[m, n]  (bias gradient, fp32 accumulator).                                    */
int bist_col_sum_acc(const void* x, float* out, int64_t M, int32_t N, int64_t ldx, int32_t dtype, void* stream);
/* Several bias gradients in one launch (the trainer queues the weight-gradient GEMMs' dz operands of a
 * whole backward pass): out[n] += sum_m x[m, n] per job.  All jobs share one dtype.               */
typedef struct BistColSum {
  const void* x; float* out;
  int64_t M; int32_t N; int64_t ldx;
} BistColSum;
int bist_col_sum_multi(const BistColSum* jobs, int32_t njobs, int32_t dtype, void* stream);
/* LayerNorm backward: dx (dtype) and da/db (+=, fp32 [d]).  dx_add (nullable, [rows, ldadd]) is added
 * to dx: the gradient arriving on the residual branch of x + sublayer(LN(x)) (modules.py:44), so the
 * two gradients of x are never summed by a separate pass.                                          */
int bist_layernorm_bwd(const void* dy, const void* x, const void* a, void* dx, float* da, float* db, int64_t rows,
                       int32_t d, int64_t lddy, int64_t ldx, int64_t lddx, float eps, const void* dx_add, int64_t ldadd,
                       void* dz, const BistDrop* dz_drop, int32_t dtype, void* stream);
/* Up to 8 LayerNorm backward passes of one geometry in one launch (see bist_layernorm_fwd_multi); every set or no set accumulates
 * da / db.  drop_row0: the set's first row in the STACKED tensor its dz mask is indexed over, (drop_row0 + row) * d + col -- the mask
 * index of a z-batched GEMM's dropout epilogue.                                                                                  */
typedef struct BistLnBwdSet {
  const void* dy; const void* x; const void* a; void* dx; float* da; float* db; const void* dx_add; void* dz; uint64_t drop_row0;
} BistLnBwdSet;
int bist_layernorm_bwd_multi(const BistLnBwdSet* sets, int32_t nsets, int64_t rows, int32_t d, int64_t lddy, int64_t ldx, int64_t lddx,
                             float eps, int64_t ldadd, const BistDrop* dz_drop, int32_t dtype, void* stream);
/* dz (nullable, [rows, d] contiguous) additionally receives dropout-mask(dz_drop) * dx with mask index row*d + col:
 * when x is the output of a GEMM with a dropout epilogue (x = drop(z) + res, modules.py:44), this is exactly the
 * gradient of z, and the GEMM's backward needs no separate masking pass.                                        */
/* da = db = NULL in bist_layernorm_bwd computes dx only (1 KiB rows: d = 512 bf16 / 256 f32); the gain / offset
 * gradients of all such LayerNorms of a backward pass are then summed by ONE batched launch per 40 jobs:
 *   da[c] += sum_r dy[r,c] (x[r,c] - mean_r) / (std_r + eps),   db[c] += sum_r dy[r,c]                      */
typedef struct BistLnGrad {
  const void* dy; const void* x; const void* a; float* da; float* db;
  int64_t rows, lddy, ldx; float eps;
} BistLnGrad;
int bist_layernorm_param_grad_multi(const BistLnGrad* jobs, int32_t njobs, int32_t d, int32_t dtype, void* stream);
/* dlut[ids[row], :] += dy[row, :] * sqrt(d)  (fp32 accumulator [V, d]).                          */
int bist_embed_bwd(const int64_t* ids, const void* dy, float* dlut, int64_t rows, int32_t d, const BistDrop* drop,
                   int32_t dtype, void* stream);
int bist_fuse_modalities_bwd(const void* score, const void* const* xs, const void* dout, void* dscore, void* const* dxs,
                             int64_t rows, int32_t n, int32_t d, int32_t dtype, void* stream);
/* dQ/dK/dV of bist_mha_core_fwd; dO and/or dP_ext (gradient w.r.t. the returned probabilities,
 * f32 [N,h,Lq,Lk]) may be NULL (not both).  Gradients are written with their own strides so they
 * can land in column slices of a packed projection gradient.                                     */
int bist_mha_core_bwd(const void* Q, const void* K, const void* V, const uint8_t* mask, const void* dO, const float* dP_ext,
                      void* dQ, void* dK, void* dV, int32_t N, int32_t Lq, int32_t Lk, int32_t h, int32_t dk,
                      int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                      int64_t lddq, int64_t lddk, int64_t lddv, int64_t dq_bs, int64_t dk_bs, int64_t dv_bs,
                      int64_t mask_bs, int64_t mask_qs, float scale, const BistDrop* drop, int32_t dtype, void* stream);
/* dscores ([B,Lq*h,T*S] in dscores_dtype: BIST_F32, or BIST_BF16 when the score products' backward takes bf16 operands --
 * saves the f32 -> bf16 pass over 4*B*Lq*h*T*S bytes) and dV ([B,T,S,*], row stride lddv) of bist_st_stage1_pv_fwd.        */
/* The same backward fed with the probabilities the fused training forward saved (P [B,G,h,Lq,KP] f32, before dropout) instead of the
 * raw scores: no score gather, no softmax recompute.  bf16, matrix-core kernel only (dk = 64, Lq <= 32, keys <= 128).            */
int bist_st_stage1_pv_bwd_p(const float* P, int32_t KP, const void* V, const uint8_t* tmask, const void* dO, void* dscores,
                            int32_t dscores_dtype, void* dV, int32_t B, int32_t T, int32_t S, int32_t Lq, int32_t h, int32_t dk,
                            int64_t ldv, int64_t lddv, int32_t direction, const BistDrop* drop, int32_t dtype, void* stream);
int bist_st_stage1_pv_bwd(const float* scores, const void* V, const uint8_t* tmask, const void* dO, void* dscores, int32_t dscores_dtype, void* dV,
                          int32_t B, int32_t T, int32_t S, int32_t Lq, int32_t h, int32_t dk, int64_t ldv, int64_t lddv,
                          int32_t direction, const BistDrop* drop, int32_t dtype, void* stream);
/* dq2f and dY of bist_st_stage2_fwd; d_rowsum (f32 [B,Lq,h], nullable) is the gradient of `rowsum`.  */
int bist_st_stage2_bwd(const void* q2f, const void* Y, const uint8_t* gmask, const void* dPY, const float* d_rowsum,
                       void* dq2f, void* dY, int32_t B, int32_t G, int32_t Lq, int32_t h, int32_t d, const BistDrop* drop,
                       int32_t dtype, void* stream);
int bist_pointer_mix_bwd(const float* logits, const float* switch_logits, int32_t n_ptr, const float* const* ptr_p,
                         const int64_t* const* ptr_text, const int32_t* ptr_len, const float* out, const float* dout,
                         float* dlogits, float* dswitch_logits, float* const* dptr_p, int64_t rows, int32_t Lt, int32_t V,
                         int32_t sigmoid_switch, void* stream);
/* Generator.forward (model/generator.py:21-27: log_softmax of the vocabulary product) and LabelSmoothing (model/label_smoothing.py)
 * on the same rows as one pass each way, for G = rows / M groups of M rows sharing the targets (the query auto-encoder heads of
 * model/optimize.py:66-82): row r has target[r % M].  Forward: row_loss f32 [rows] (KL of the smoothed one-hot target against
 * softmax(logits); 0 at pad targets) and the rows' log-sum-exp.  Backward: dlogits[r][v] = gout[(r / M) * gout_stride] / denom * (softmax - smoothed
 * target), written in dlogits_dtype (gout_stride 0: one upstream gradient for all groups).  bist_sum_div_groups: out[g * out_stride] = sum of the group's M row losses / denom.
 * bist_stack_rows: n <= 4 equally sized buffers behind one another (one launch; srcs is a HOST array).                              */
int bist_xent_smooth_fwd(const float* logits, const int64_t* target, int64_t M, int64_t rows, int32_t V, float smoothing, int32_t pad,
                         float* row_loss, float* lse, void* stream);
int bist_xent_smooth_bwd(const float* logits, const float* lse, const int64_t* target, int64_t M, int64_t rows, const float* gout,
                         int32_t gout_stride, const int64_t* denom, void* dlogits, int32_t dlogits_dtype, int32_t V, float smoothing,
                         int32_t pad, void* stream);
int bist_sum_div_groups(const float* x, int64_t M, int32_t G, const int64_t* denom, float* out, int32_t out_stride, void* stream);
int bist_stack_rows(const void* const* srcs, int32_t n, void* out, int64_t bytes_each, void* stream);
/* A small linear layer over the CONCATENATION of n_parts <= 4 tensors [rows][d] without the concatenation: part j multiplies column block
 * j of W [ns <= 4][ldw >= n_parts d]; out [rows][ns] = sum_j part_j . W_j^T + bias, f32 or (out_dtype) the operand dtype.  The switch logits
 * of (Multi)PointerGenerator (model/generator.py:69-71, 119-121: pointer_gen_W(torch.cat(parts, -1))) and the fusion logits of
 * MultimodalDecoder8 (model/decoder.py:142-159: vc_combine_W(torch.cat(...))).
 * Backward: dsw [rows][ns] f32 or (dsw_dtype) the operand dtype; dparts[j] = dsw . W_j (+ residuals[j]: an addend of that part's gradient
 * from elsewhere, e.g. the fusion's weighted sum) written (entries / the arrays may be NULL), dW [ns][lddw] and db [ns] written or
 * accumulated (dW NULL = not wanted; db only together with dW).  parts / dparts / residuals are HOST arrays of n_parts pointers.        */
int bist_switch_logits_fwd(const void* const* parts, int32_t n_parts, const void* W, int64_t ldw, const void* bias, void* out,
                           int32_t out_dtype, int64_t rows, int32_t d, int32_t ns, int32_t dtype, void* stream);
int bist_switch_logits_bwd(const void* const* parts, int32_t n_parts, const void* W, int64_t ldw, const void* dsw, int32_t dsw_dtype,
                           void* const* dparts, const void* const* residuals, void* dW, int64_t lddw, int32_t dw_dtype,
                           int32_t dw_accumulate, float* db, int32_t db_accumulate, int64_t rows, int32_t d, int32_t ns, int32_t dtype,
                           void* stream);
/* The pointer attention of (Multi)PointerGenerator with its text vector, training and evaluation (model/generator.py:106-118: a
 * single-head MultiHeadedAttention whose value product and output projection the reference computes and throws away, then
 * `(p.unsqueeze(-1) * enc.unsqueeze(1)).sum(2)`): q [B][Lt][d] and k [B][L][d] the projected queries / keys, mask [B][L] (row stride
 * mask_bs, 0 = one row for all) and, when text != NULL, also text[b][t] != unk (generator.py:106-107);
 * p f32 [B][Lt][L] = softmax_t(scale q.k, masked -1e9); tv [B][Lt][d] = sum_t p enc[b][t] (NULL: not wanted).
 * Backward: dp f32 (the mixture's gradient, nullable), dtv (nullable) -> dq, dk, denc (written, not accumulated).             */
int bist_pointer_attn_fwd(const void* q, const void* k, const void* enc, const uint8_t* mask, int64_t mask_bs, const int64_t* text,
                          int64_t unk, float* p, void* tv, int64_t B, int32_t Lt, int32_t L, int32_t d, float scale, int32_t dtype,
                          void* stream);
int bist_pointer_attn_bwd(const void* q, const void* k, const void* enc, const float* p, const float* dp, const void* dtv, void* dq,
                          void* dk, void* denc, int64_t B, int32_t Lt, int32_t L, int32_t d, float scale, int32_t dtype, void* stream);
/* Decode-step form of MultiPointerGenerator.forward (model/generator.py:84-127) for rows that share one dialogue (the hypotheses of
 * a beam-search turn, model/decode.py:59-66): per source j the caller holds, per TURN, M_j = K_j W_q [L_j][d] and c_j = K_j b_q [L_j]
 * (K_j the projected keys of generator.py:109, so that scores = x.M_j^T + c_j), mask_j [L_j] (generator.py:106-107), E_j = enc_j W_sw,j^T
 * [L_j][n_ptr+1] (the text vector's block of the switch product, generator.py:117-121) and the source's token ids text_j [L_j];
 * x = the decoded rows [rows][d], tgt = their target embeddings, logits f32 [rows][V] the vocabulary product (generator.py:90),
 * Wsw [n_ptr+1][ldw] = pointer_gen_W.weight with blocks [x | tgt | tv_0 | ..], bsw its bias, scale = 1/sqrt(d).  out f32 [rows][V] =
 * the log mixture of bist_pointer_mix_fwd; p_out (optional) receives the pointer probabilities [rows][L_j].  src is a HOST array.   */
typedef struct BistPtrDecSrc {
  const float* M; const float* c; const uint8_t* mask; const float* E; const int64_t* text; float* p_out; int32_t L; int32_t pad_;
} BistPtrDecSrc;
int bist_pointer_decode_mix_fwd(const void* x, const void* tgt, const float* logits, const BistPtrDecSrc* src, int32_t n_ptr,
                                const void* Wsw, int64_t ldw, const void* bsw, float scale, float* out, int64_t rows, int32_t d,
                                int32_t V, int32_t dtype, void* stream);
/* The pointer generator's text vector (generator.py:117-118), inference: out[row, :] = sum_t p[row, t] * enc[row / Lt, t, :]
 * (p f32 [rows][L], enc [rows / Lt][L][d], out [rows][d] in dtype); training uses the batched bist_gemm (it needs the backward products). */
int bist_text_vector_fwd(const float* p, const void* enc, void* out, int64_t rows, int32_t Lt, int32_t L, int32_t d, int32_t dtype,
                         void* stream);
int bist_log_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int32_t V, void* stream);
/* dlogp[row,v] = -smoothed_target[row,v] * gout[0] / denom[0]  (gout, denom: device scalars).     */
int bist_label_smoothing_bwd(const int64_t* target, const float* gout, const int64_t* denom, float* dlogp, int64_t rows,
                             int32_t V, float smoothing, int32_t pad, void* stream);
int bist_cast_from_f32(const float* src, void* dst, int64_t n, int32_t dtype, void* stream);
/* dst[i] += src[i]: folds the fp32 accumulators of the atomically-reduced gradients (biases, LayerNorm
 * gains, embedding rows) into the gradient buffer once per step.                                   */
int bist_add_f32_into(const float* src, void* dst, int64_t n, int32_t dtype, void* stream);
/* One Adam step (torch.optim.Adam semantics, the optimiser NoamOpt wraps: train.py:129-130) on fp32
 * master weights p with moments m, v; g in grad_dtype, scaled by grad_scale first; `work`
 * (nullable) receives the updated weights in work_dtype (the bf16 copy the kernels read).        */
int bist_adam_step(float* p, const void* g, float* m, float* v, void* work, int64_t n, float lr, float beta1, float beta2,
                   float eps, int32_t step, float grad_scale, int32_t grad_dtype, int32_t work_dtype, void* stream);
/* The same step with the per-step scalars read from device memory: hyper = {lr, 1 - beta1^t, 1 - beta2^t, grad_scale} (f32).
 * A launch inside a captured hipGraph freezes its kernel arguments; with this form the optimiser is part of the replayed
 * step (the host refreshes `hyper` before each replay) and can run beside the tail of the backward pass.                */
/* hyper[0..3] = {Noam rate of step t (NoamOpt.rate, model/optimize.py:28-34), 1 - beta1^t, 1 - beta2^t, grad_scale} with t read from
 * the DEVICE counter step_ctr[0]: the scalars of bist_adam_step_dev without a host buffer that a later step could overwrite
 * while an earlier one is still queued.                                                                                     */
int bist_noam_hyper(const int64_t* step_ctr, float* hyper, float d_model, float factor, float warmup, float beta1, float beta2,
                    float grad_scale, void* stream);
int bist_adam_step_dev(float* p, const void* g, float* m, float* v, void* work, int64_t n, const float* hyper, float beta1,
                       float beta2, float eps, int32_t grad_dtype, int32_t work_dtype, void* stream);
/* The same update as a BACKGROUND launch: at most max_blocks workgroups walk the range (0 = the chip-filling form above), so that it can
 * run beside the latency-bound launches of a backward pass without taking the CUs' wave slots from them (n % 4 == 0, aligned pointers). */
int bist_adam_step_dev_bg(float* p, const void* g, float* m, float* v, void* work, int64_t n, const float* hyper, float beta1,
                          float beta2, float eps, int32_t grad_dtype, int32_t work_dtype, int32_t max_blocks, void* stream);
/* Deferred form: the update of optimiser step t is applied at the HEAD of step t+1, beside its forward pass (the tail of a step is
 * then the backward pass alone).  bist_noam_hyper_pending reads t = pending[0] (device int64; 0 = nothing pending), writes
 * hyper[0..4] = {rate(t), 1 - beta1^t, 1 - beta2^t, grad_scale, apply = (t > 0)} and sets pending[0] = 0; the caller sets
 * pending[0] = t when step t's gradients are complete.  bist_adam_apply_dev is bist_adam_step_dev on 4 elements per thread that
 * also CLEARS g (the next backward accumulates into zeros; with apply = 0 it only clears): n a multiple of 4, pointers 16-byte
 * (fp32) / 8-byte (bf16) aligned.  Element for element the same arithmetic as bist_adam_step.                                */
int bist_noam_hyper_pending(int64_t* pending, float* hyper, float d_model, float factor, float warmup, float beta1, float beta2,
                            float grad_scale, void* stream);
int bist_adam_apply_dev(float* p, void* g, float* m, float* v, void* work, int64_t n, const float* hyper, float beta1,
                        float beta2, float eps, int32_t grad_dtype, int32_t work_dtype, void* stream);

/* dst = cast(src) between f32 and bf16 (n elements). */
int bist_cast(const void* src, void* dst, int64_t n, int32_t src_dtype, int32_t dst_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * The inputs of a dialogue turn staged into the fixed buffers the turn's hipGraphs read -- the place of batch.move_to_cuda() + the
 * per-field copies in the reference's generate loop (generate.py:30-60 / data/dataset.py:130-150) -- as ONE launch: job j copies `rows`
 * rows of src_row_bytes to rows of dst_row_bytes >= src_row_bytes and fills each row's tail with the pad_bytes-byte pattern `pad`
 * (little endian: the pad token id of an int64 tensor, 0 = False of a mask).  The tail is how token tensors reach their length bucket
 * (model/decode.py: padded positions are masked everywhere).  jobs is a HOST array of 1..16 entries; device pointers, any alignment.
 * ------------------------------------------------------------------------------------------ */
typedef struct BistStageJob {
  const void* src; void* dst; int64_t rows; int64_t src_row_bytes; int64_t dst_row_bytes; uint64_t pad; int32_t pad_bytes; int32_t reserved_;
} BistStageJob;
int bist_stage_inputs(const BistStageJob* jobs, int32_t n_jobs, void* stream);

/* ------------------------------------------------------------------------------------------
 * Replay of a captured multi-stream hipGraph as ONE LINEAR GRAPH PER STREAM (csrc/graphsplit.hip).
 *
 * The reference runs its step as eager PyTorch launches on one stream (train.py:29-37; there is no graph or executor in the
 * reference to cite); this build captures the step over three streams (bist_amd/train.py) and, instead of handing the multi-branch
 * graph to the runtime's executor -- which serialises independent branches -- splits it: per capture stream a
 * clone that keeps that stream's nodes in capture order, with every cross-stream dependency replaced by a SIGNAL launch after the
 * producer and a WAIT launch before the consumer on a device flag word (step-numbered, so nothing is reset between replays; waits
 * implied by earlier waits are dropped; waits are bounded by a time-out that counts into an error word instead of hanging a queue).
 *
 *   bist_graph_capture_tail   during a capture: the node the capturing `stream` would depend on next (its last node), or NULL --
 *                             the caller records "this node was captured on this stream" after each of its launches
 *   bist_graph_nodes          the node handles of a hipGraph_t in the order the labels are given in
 *   bist_graph_split_plan     the planner alone on index arrays (host only, no device): tests check it against random DAGs
 *   bist_graph_split_create   plan for `graph`: labels[i] = chain (0..n_chains-1) of node i, -1 = unknown (joins a predecessor's chain);
 *                             main_chain = the chain that is launched into the caller's stream
 *   bist_graph_split_sync_words   uint64 words of DEVICE memory the caller provides, zeroed: [n_chains epochs][1 error count][flags][2 stamps per sync launch]
 *   bist_graph_split_build    clones, prunes, inserts the sync launches, instantiates; timeout_ticks of the 100 MHz device clock per wait
 *   bist_graph_split_launch   one hipGraphLaunch per chain; streams[c] must sit on pairwise different hardware queues
 *   bist_graph_queues_distinct    probe for that: scratch = 4 zeroed uint64 (device); after both streams drained scratch[1] == 0 iff
 *                             a launch on stream_b could run while stream_a held a wait for it AND a further launch behind that wait
 * The library allocates no device memory; the BistGraphSplit object is host memory released by bist_graph_split_destroy.
 * ------------------------------------------------------------------------------------------ */
typedef struct BistGraphSplit BistGraphSplit;
int bist_graph_capture_tail(void* stream, void** node_out);
int bist_graph_nodes(void* hip_graph, void** nodes_out, int32_t cap, int32_t* n_out);
int64_t bist_graph_split_plan(int32_t n_nodes, const int32_t* edge_from, const int32_t* edge_to, int32_t n_edges, const int32_t* labels,
                              int32_t n_chains, int32_t main_chain, int32_t* out, int64_t cap);
int bist_graph_edges(void* hip_graph, int32_t* from_out, int32_t* to_out, int32_t cap, int32_t* n_out);      /* analysis aid: edges as index pairs */
int64_t bist_graph_split_dump(const BistGraphSplit* split, int32_t* out, int64_t cap);                        /* analysis aid: the plan, flat form of _plan */
int bist_graph_split_create(void* hip_graph, const int32_t* labels, int32_t n_labels, int32_t n_chains, int32_t main_chain, BistGraphSplit** out);
int64_t bist_graph_split_sync_words(const BistGraphSplit* split);
/* The sync launches in the order of their stamp words (the caller's words end with two uint64 per sync launch: device clock at its begin
 * and end): 8 int32 each -- chain, kind (1 epoch bump, 2 signal, 3 wait), index of the captured node a signal follows / a wait precedes
 * (-1: none), flag ids (signal: one; wait: up to four, -1 unused), 0.  Returns the number of int32 (also when cap is too small). */
int64_t bist_graph_split_sync_items(const BistGraphSplit* split, int32_t* out, int64_t cap);
int bist_graph_split_info(const BistGraphSplit* split, int32_t* out5, int32_t* nodes_per_chain);
int bist_graph_split_build(BistGraphSplit* split, void* hip_graph, void* sync_words, int64_t timeout_ticks);
int bist_graph_split_launch(BistGraphSplit* split, void* const* streams);
int bist_graph_split_launch_chain(BistGraphSplit* split, int32_t chain, void* stream);      /* development aid: one chain's launch alone */
void bist_graph_split_destroy(BistGraphSplit* split);
int bist_graph_queues_distinct(void* stream_a, void* stream_b, void* scratch, int64_t timeout_ticks);
/* Launch-to-launch time (us) of a linear graph of n one-thread launches replayed into `stream`, alone (resident_stream NULL) or while one
 * wave stays resident on `resident_stream` for resident_ticks of the 100 MHz clock: hardware queues that share a dispatch pipe slow each
 * other down threefold (csrc/graphsplit.hip), and the chains of a split graph must not.  word: 2 uint64 of device memory.  Synchronises. */
int bist_graph_queue_pace(void* stream, int32_t n, void* resident_stream, int64_t resident_ticks, void* word, float* us_per_launch_out);
/* Ready flag of the overlapped gradient exchange (the reference's DataParallel exchange, train.py:96-99, as a bucketed all-reduce that
 * starts DURING the backward pass): *flag = *value_dev -- flag a 64-bit word in pinned (host-coherent) memory, value_dev a 64-bit word in
 * device memory (the step counter) -- as a system-scope release store ordered behind the stream's earlier launches.  Capturable: the
 * replayed step signals "this stream's share of bucket j is final", the host polls the word and issues the bucket's all-reduce. */
int bist_flag_signal(void* flag, const void* value_dev, void* stream);
/* Development aid: a one-wave launch that stays resident on `stream` for `ticks` of the 100 MHz clock (mode 0: sleeps and reads the clock;
 * 1: also polls word[0] with relaxed loads; 2: with acquire loads; 3: `ticks` rounds of s_sleep, no memory, no clock) -- measures what a
 * resident wave on another queue costs the launches of a step (scripts/probe_idle_wave.py).  word: 2 uint64 of device memory. */
int bist_dev_idle_wave(void* stream, int64_t ticks, int32_t mode, void* word);

#ifdef __cplusplus
}
#endif
#endif /* BIST_HIP_H */
