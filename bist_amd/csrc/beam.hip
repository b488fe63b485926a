// Beam-search bookkeeping of one decode step on the device (reference: model/decode.py:59-99).
//
// The reference copies every hypothesis's log-prob row [V] to the host (decode.py:71, one synchronisation per hypothesis and step),
// argsorts it with numpy and runs the beam update in Python.  Here a step's update is two small launches and NO host round trip:
//   beam_topk_kernel    one workgroup per hypothesis row: lp_vec = logp[row] + lp[row] (decode.py:72, the same f32 addition), its
//                       KC = beam + 2 largest entries in descending order (value, token id) and lp_vec[<eos>] -- the loop of
//                       decode.py:79-97 takes at most `beam` candidates per row and skips at most two symbols;
//   beam_select_kernel  one wave (lane 0 walks the candidates out of LDS, the masks are rewritten one slot per lane): decode.py:74-97 literally -- the completed-hypothesis score of every row (from min_len on), then
//                       the rows in order, their candidates in descending order, into a list of at most `beam` entries with
//                       replace-the-minimum (first minimal index, strict comparisons) and the early break; it writes the next step's
//                       inputs in place (token per surviving hypothesis, its running score, its ancestry mask over the decoder
//                       kernel's self-attention slots) and the step's record (parent row, token, score) for the host to rebuild the
//                       token lists from at the END of the turn.
// numpy's argsort orders equal values by its sort's internals, so whenever the KC + 1 largest values of a row are not all distinct
// (or a value is NaN, or fewer than `beam` hypotheses survive) a sticky flag is raised and the caller repeats the turn on the host
// path: the n-best lists are the reference's in every case.
#include "common.hpp"

namespace {

constexpr int KC_MAX = 16;

struct BeamArgs {
  const float* logp;         // [n, V] log-probs of this step's rows
  float* lp;                 // [beam] running scores: read for this step's rows, rewritten for the next step's
  long* tok;                 // [beam] next step's input tokens
  unsigned char* mask64;     // [beam, 64] ancestry masks over the decoder kernel's slots (read: this step's, written: next step's)
  unsigned char* mask_out;   // [beam, LkS_next] the next step's mask in the width its launch reads (32 or 64)
  float* cand_val; int* cand_idx; float* eos_val;      // scratch [n, KC], [n, KC], [n]
  int* rec_parent; int* rec_token; float* rec_score; float* rec_comp; int* rec_n;      // records [max_len, beam] ..., rows per step [max_len]
  int* flag;                 // sticky: 1 ties, 2 NaN, 4 fewer than beam survivors
  int n, V, beam, KC, step, min_len, unk, eos, dec_eos, slot0_next, LkS_next;
  float penalty;
};

// order-preserving map of a float onto an unsigned (larger float <-> larger unsigned; -inf smallest of the finite order)
__device__ __forceinline__ unsigned ord_of(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float float_of(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ __launch_bounds__(256) void beam_topk_kernel(const BeamArgs a) {
  // KC + 1 rounds of "largest remaining entry": every thread keeps 16 entries in registers; a round is a 64-bit (value, ~index) maximum
  // per thread, a shuffle reduction per wave and ONE barrier (the four wave results alternate between two LDS slots).  Equal values
  // come out in successive rounds, so "this round's value == the last one's" finds every tie among the KC + 1 largest.
  __shared__ unsigned long long skey[2][4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* lr = a.logp + (long)row * a.V;
  const float base = a.lp[row];
  constexpr int PER = 16;                              // V <= 4096
  float v[PER];
  bool nan = false;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + 256 * u;
    v[u] = i < a.V ? lr[i] + base : -INFINITY;
    nan |= (v[u] != v[u]);
  }
  if (tid == 0) a.eos_val[row] = lr[a.eos] + base;
  float last = INFINITY;
  for (int k = 0; k <= a.KC; ++k) {                    // KC + 1 rounds: the extra one only checks distinctness
    unsigned long long best = 0ull;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const unsigned long long key = ((unsigned long long)ord_of(v[u]) << 32) | (unsigned)(0xffffffffu - (unsigned)(tid + 256 * u));   // smaller index wins a tie
      best = key > best ? key : best;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const unsigned long long o = __shfl_xor(best, s);
      best = o > best ? o : best;
    }
    if (lane == 0) skey[k & 1][w] = best;
    __syncthreads();
    unsigned long long b4 = skey[k & 1][0];
#pragma unroll
    for (int i = 1; i < 4; ++i) b4 = skey[k & 1][i] > b4 ? skey[k & 1][i] : b4;
    const float bv = float_of((unsigned)(b4 >> 32));
    const int bidx = (int)(0xffffffffu - (unsigned)(b4 & 0xffffffffu));
    if (tid == 0) {
      if (bv == last) atomicOr(a.flag, 1);
      if (k < a.KC) { a.cand_val[row * a.KC + k] = bv; a.cand_idx[row * a.KC + k] = bv > -INFINITY ? bidx : -1; }
    }
    last = bv;
#pragma unroll
    for (int u = 0; u < PER; ++u) if (tid + 256 * u == bidx) v[u] = -INFINITY;
  }
  if (nan) atomicOr(a.flag, 2);
}

__global__ __launch_bounds__(64) void beam_select_kernel(const BeamArgs a) {
  // one wave: the candidates are staged in LDS by all lanes, lane 0 walks them (decode.py:74-97 is sequential by construction: at most
  // beam * KC short iterations on LDS operands), and the ancestry masks are rewritten one slot per lane
  __shared__ float s_val[KC_MAX * KC_MAX], s_eos[KC_MAX], s_score[KC_MAX];
  __shared__ int s_idx[KC_MAX * KC_MAX], s_parent[KC_MAX], s_token[KC_MAX], s_cnt;
  const int n = a.n, beam = a.beam, KC = a.KC, l = a.step, lane = threadIdx.x;
  for (int i = lane; i < n * KC; i += 64) { s_val[i] = a.cand_val[i]; s_idx[i] = a.cand_idx[i]; }
  if (lane < n) s_eos[lane] = a.eos_val[lane];
  __syncthreads();
  if (lane == 0) {
    int cnt = 0, argmin = 0;
    int parent[KC_MAX], token[KC_MAX];
    float score[KC_MAX];
    for (int idx = 0; idx < n; ++idx) {
      for (int c = 0; c < KC; ++c) {
        const int o = s_idx[idx * KC + c];
        if (o < 0) break;
        if (o == a.unk || (!a.dec_eos && o == a.eos)) continue;
        const float new_lp = s_val[idx * KC + c];
        if (cnt == beam) {
          if (score[argmin] < new_lp) {
            parent[argmin] = idx; token[argmin] = o; score[argmin] = new_lp;
            argmin = 0;
            for (int j = 1; j < beam; ++j) if (score[j] < score[argmin]) argmin = j;       // first minimal index
          } else {
            break;
          }
        } else {
          parent[cnt] = idx; token[cnt] = o; score[cnt] = new_lp; ++cnt;
          if (cnt == beam) {
            argmin = 0;
            for (int j = 1; j < beam; ++j) if (score[j] < score[argmin]) argmin = j;
          }
        }
      }
    }
    if (cnt < beam) atomicOr(a.flag, 4);
    for (int j = 0; j < cnt; ++j) { s_parent[j] = parent[j]; s_token[j] = token[j]; s_score[j] = score[j]; }
    s_cnt = cnt;
    a.rec_n[l] = n;
  }
  __syncthreads();
  const int cnt = s_cnt;
  // completed hypothesis of each row (decode.py:74-78): len(out) = l tokens so far
  if (lane < n) a.rec_comp[l * beam + lane] = l >= a.min_len ? s_eos[lane] + a.penalty * (float)(l + 1) : -INFINITY;
  // next step's inputs: token, running score (lane j), ancestry mask (lane = slot: the parent's slots and the hypothesis's own new one)
  if (lane < beam) {
    const bool live = lane < cnt;
    a.rec_parent[l * beam + lane] = live ? s_parent[lane] : -1;
    a.rec_token[l * beam + lane] = live ? s_token[lane] : -1;
    a.rec_score[l * beam + lane] = live ? s_score[lane] : -INFINITY;
    a.tok[lane] = live ? (long)s_token[lane] : 0L;
    a.lp[lane] = live ? s_score[lane] : 0.f;
  }
  unsigned char nm[KC_MAX];
#pragma unroll
  for (int j = 0; j < KC_MAX; ++j)
    if (j < beam) nm[j] = j < cnt ? (unsigned char)(a.mask64[s_parent[j] * 64 + lane] | (lane == a.slot0_next + j)) : (unsigned char)(lane == 0);
  __syncthreads();                                       // every lane has read the old masks of its slot before any is rewritten
#pragma unroll
  for (int j = 0; j < KC_MAX; ++j)
    if (j < beam) {
      a.mask64[j * 64 + lane] = nm[j];
      if (a.mask_out && lane < a.LkS_next) a.mask_out[j * a.LkS_next + lane] = nm[j];
    }
}

// The inputs of a dialogue turn into the static buffers its hipGraphs read, one launch for all fields (model/decode.py staged them with a
// torch copy per field after padding each with torch ops: ~30 eager launches, 0.25 ms of host time at the head of every 7 ms turn).
// Job j: rows x src_row_bytes from src, written to rows of dst_row_bytes; the tail of every row is filled with the pad_bytes-byte pattern
// `pad` (the pad token id of an int64 tensor, False of a mask).  16-byte lanes when the job's pointers and row sizes allow, bytes otherwise.
struct StageJobK { const unsigned char* src; unsigned char* dst; long rows, srb, drb; unsigned long long pad; int pad_bytes, vec; };
struct StageArgs { StageJobK j[16]; };

__global__ __launch_bounds__(256) void stage_inputs_kernel(const StageArgs a) {
  const StageJobK& jb = a.j[blockIdx.y];
  const long step = (long)gridDim.x * 256;
  if (jb.vec) {                                                  // whole 16-byte pieces: src_row_bytes == dst_row_bytes, everything aligned
    const long n16 = jb.rows * jb.drb / 16;
    const uint4* s = reinterpret_cast<const uint4*>(jb.src);
    uint4* d = reinterpret_cast<uint4*>(jb.dst);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += step) d[i] = s[i];
    return;
  }
  const long n = jb.rows * jb.drb;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    const long r = i / jb.drb, b = i - r * jb.drb;
    jb.dst[i] = b < jb.srb ? jb.src[r * jb.srb + b] : (unsigned char)(jb.pad >> (8 * (int)((b - jb.srb) % jb.pad_bytes)));
  }
}

}  // namespace

extern "C" int bist_stage_inputs(const BistStageJob* jobs, int32_t n_jobs, void* stream) {
  BIST_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 16, "bist_stage_inputs: 1..16 jobs");
  StageArgs a;
  long most = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const BistStageJob& b = jobs[j];
    BIST_REQUIRE(b.src && b.dst && b.rows >= 1 && b.src_row_bytes >= 1 && b.dst_row_bytes >= b.src_row_bytes &&
                 (b.pad_bytes == 1 || b.pad_bytes == 2 || b.pad_bytes == 4 || b.pad_bytes == 8) && b.src_row_bytes % b.pad_bytes == 0 &&
                 b.dst_row_bytes % b.pad_bytes == 0,
                 "bist_stage_inputs: job %d: rows of src_row_bytes <= dst_row_bytes, both whole elements of pad_bytes (1, 2, 4 or 8)", j);
    const bool vec = b.src_row_bytes == b.dst_row_bytes && (b.rows * b.dst_row_bytes) % 16 == 0 &&
                     (((uintptr_t)b.src | (uintptr_t)b.dst) & 15) == 0;
    a.j[j] = StageJobK{(const unsigned char*)b.src, (unsigned char*)b.dst, (long)b.rows, (long)b.src_row_bytes, (long)b.dst_row_bytes,
                       (unsigned long long)b.pad, b.pad_bytes, vec ? 1 : 0};
    const long units = vec ? b.rows * b.dst_row_bytes / 16 : b.rows * b.dst_row_bytes;
    most = units > most ? units : most;
  }
  for (int j = n_jobs; j < 16; ++j) a.j[j] = a.j[0];
  long blocks = (most + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(stage_inputs_kernel, dim3((unsigned)blocks, (unsigned)n_jobs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  BIST_LAUNCH_CHECK("bist_stage_inputs");
  return BIST_OK;
}

extern "C" int bist_beam_step(const float* logp, float* lp, int64_t* tok, uint8_t* mask64, uint8_t* mask_out, float* cand_val, int32_t* cand_idx,
                              float* eos_val, int32_t* rec_parent, int32_t* rec_token, float* rec_score, float* rec_comp, int32_t* rec_n,
                              int32_t* flag, int32_t n, int32_t V, int32_t beam, int32_t step, int32_t min_len, int32_t unk, int32_t eos,
                              int32_t dec_eos, int32_t slot0_next, int32_t LkS_next, float penalty, void* stream) {
  BIST_REQUIRE(logp && lp && tok && mask64 && cand_val && cand_idx && eos_val && rec_parent && rec_token && rec_score && rec_comp && rec_n && flag,
               "bist_beam_step: null pointer");
  BIST_REQUIRE(n >= 1 && n <= beam && beam >= 1 && beam + 2 <= KC_MAX && V >= 1 && V <= 4096 && step >= 0 && eos >= 0 && eos < V,
               "bist_beam_step: 1 <= n <= beam <= %d, V <= 4096", KC_MAX - 2);
  BIST_REQUIRE(LkS_next == 0 || ((LkS_next == 32 || LkS_next == 64) && mask_out), "bist_beam_step: the next step's mask is 32 or 64 slots wide");
  BeamArgs a{logp, lp, (long*)tok, mask64, LkS_next ? mask_out : nullptr, cand_val, cand_idx, eos_val, rec_parent, rec_token, rec_score, rec_comp, rec_n,
             flag, n, V, beam, beam + 2, step, min_len, unk, eos, dec_eos, slot0_next, LkS_next, penalty};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(beam_topk_kernel, dim3((unsigned)n), dim3(256), 0, st, a);
  BIST_LAUNCH_CHECK("bist_beam_step (top-k)");
  hipLaunchKernelGGL(beam_select_kernel, dim3(1), dim3(64), 0, st, a);
  BIST_LAUNCH_CHECK("bist_beam_step (select)");
  return BIST_OK;
}
