// A captured multi-stream hipGraph replayed as ONE LINEAR GRAPH PER STREAM, tied together by device-side flags.
//
// Why.  A training step is captured over three HIP streams (t2s chain, s2t chain, caption + decoder chain: ~1 200 kernel nodes).  The HIP
// runtime replays a multi-branch graph through its own executor: it re-assigns the nodes to internal queues (first child inherits the
// parent's queue, further children round-robin) and releases cross-queue dependencies in coarse pieces.  Device-side timestamps taken
// INSIDE a replay (bist_amd/stamps.py, profiles/r04_step_stamps_base.txt) show what that costs: the two direction chains of reasoning
// layer l start only when the caption layer l has finished on "its" queue although they depend on nothing it computes -- a layer's
// forward takes 690 us where its longest dependent chain is ~220 us -- and with the whole graph enqueued behind a 30 ms spin kernel the
// device still needs the same 8.9 ms: the step is bound by the executor's serialisation, not by the host and not by the kernels.
// A graph of ONE branch, however, is submitted as a pre-built batch of AQL packets that the queue runs strictly in order, 0.3 us of host
// time per node.
//
// What.  bist_graph_split_* take the captured hipGraph and a label per node (the stream it was captured on), and build one clone per
// label that keeps only that label's nodes, chained linearly in capture order.  Every dependency between two chains becomes a pair of
// one-thread launches: after the producer node a SIGNAL (store-release of the chain's step number -- its "epoch" -- to a flag word), before
// the consumer node a WAIT (load-acquire until the flag has reached the consumer chain's epoch; s_sleep between polls; bounded by a
// time-out that sets an error word instead of hanging the queue).  Vector clocks over the chains drop every wait that an earlier wait
// already implies.  Each chain starts with a launch that increments its epoch; side chains then wait for the main chain's START flag and
// signal an END flag, the main chain's tail waits for all END flags: a step begins after the previous step has finished on every
// chain, and whatever the launching stream runs after the main graph sees the whole step done.  The chains are launched into
// caller-provided streams that must sit on DIFFERENT hardware queues (a wait that shares its queue with the signal it waits for can
// only time out): bist_graph_queues_distinct probes that with the same two kernels.
//
// The planner (labels, chain order, which waits are needed) is plain host code over index arrays -- bist_graph_split_plan -- so that
// tests can check it against random DAGs without a device: every edge of the DAG must be implied by chain order and signal/wait pairs.
#include "common.hpp"

#include <algorithm>
#include <array>
#include <vector>

namespace {

constexpr int MAXC = 8;            // chains (hardware queues) at most
constexpr int MAXW = 4;            // flags per wait launch

// Every sync launch leaves the device clock (100 MHz) in its own two stamp words -- begin, end -- so that one replay can be read as a
// timeline afterwards: where each chain was when, and how long every wait blocked it (bist_amd/graphsplit.py: timeline()).
__global__ void gs_bump_kernel(unsigned long long* epoch, unsigned long long* stamp) {
  if (stamp) stamp[0] = stamp[1] = wall_clock64();
  *epoch = *epoch + 1ULL;
}

__global__ void gs_signal_kernel(unsigned long long* flag, const unsigned long long* epoch, unsigned long long* stamp) {
  if (stamp) stamp[0] = stamp[1] = wall_clock64();
  __hip_atomic_store(flag, *epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// value: flags must reach *epoch (epoch != null) or `fixed` (probe launches)
__global__ void gs_wait_kernel(const unsigned long long* f0, const unsigned long long* f1, const unsigned long long* f2,
                               const unsigned long long* f3, const unsigned long long* epoch, unsigned long long fixed,
                               unsigned long long* err, unsigned long long timeout_ticks, unsigned long long* stamp) {
  const unsigned long long want = epoch ? *epoch : fixed;
  const unsigned long long t0 = wall_clock64();
  if (stamp) stamp[0] = t0;
  const unsigned long long* fs[MAXW] = {f0, f1, f2, f3};
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const unsigned long long* f = fs[i];
    if (!f) continue;
    // RELAXED polls: an acquire here would invalidate this XCD's L2 on every iteration -- measured: one wave spinning with acquire loads
    // slows every kernel on the device (a 9.1 ms step takes 15.7 ms beside it).  The wait reads nothing but the flag; the launch that
    // follows it in the queue begins with its own acquire, which is the one that matters.
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > timeout_ticks) {          // every wave reaches an exit: a lost signal voids the step, it does not hang the queue
        __hip_atomic_fetch_add(err, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamp) stamp[1] = wall_clock64();
        return;
      }
    }
  }
  if (stamp) stamp[1] = wall_clock64();
}

// development aid (scripts/probe_idle_wave.py): one wave that stays resident for `ticks` of the 100 MHz clock.  mode 0: sleeps and reads
// the clock; 1: also polls a word with relaxed agent-scope loads; 2: polls with acquire loads; 3: sleeps a fixed number of rounds
// without touching memory or the clock.
__global__ void gs_idle_kernel(unsigned long long ticks, int mode, const unsigned long long* word, unsigned long long* sink) {
  unsigned long long acc = 0;
  if (mode == 3) {
    for (unsigned long long i = 0; i < ticks; ++i) __builtin_amdgcn_s_sleep(127);
  } else {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
      __builtin_amdgcn_s_sleep(8);
      if (mode == 1) acc += __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mode == 2) acc += __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 0xdeadbeefULL) *sink = acc;
}

struct Item {                      // one launch of a chain's linear graph
  int kind;                        // 0 = captured node, 1 = bump, 2 = signal, 3 = wait
  int node;                        // kind 0: node index; kind 2: flag id
  int flags[MAXW];                 // kind 3: flag ids (-1 = unused)
};

struct Plan {
  int n = 0, C = 0, main_chain = 0, n_flags = 0;       // flags: [0] START, [1 .. C-1] END of side chains (index by chain, main's unused), then one per signalled node
  std::vector<int> label, pos, flag_of;                // per node
  std::vector<std::vector<Item>> seq;                  // per chain
  int n_wait = 0, n_signal = 0;
  int n_sync() const { int k = 0; for (auto& s : seq) for (auto& it : s) k += it.kind != 0; return k; }
};

// ---- the planner: pure index arithmetic ---------------------------------------------------------------------------------------
int make_plan(int n, const int* ef, const int* et, int ne, const int* labels_in, int C, int main_chain, Plan& P) {
  if (n <= 0 || C < 1 || C > MAXC || main_chain < 0 || main_chain >= C) { bist_set_error("bist_graph_split_plan: %d nodes, %d chains (1..%d), main %d", n, C, MAXC, main_chain); return BIST_EINVAL; }
  std::vector<std::vector<int>> pred(n), succ(n);
  for (int e = 0; e < ne; ++e) {
    if (ef[e] < 0 || ef[e] >= n || et[e] < 0 || et[e] >= n || ef[e] == et[e]) { bist_set_error("bist_graph_split_plan: edge %d out of range", e); return BIST_EINVAL; }
    pred[et[e]].push_back(ef[e]); succ[ef[e]].push_back(et[e]);
  }
  // a topological order that keeps the capture (index) order wherever the edges allow: Kahn with the smallest ready index first
  std::vector<int> order; order.reserve(n);
  {
    std::vector<int> indeg(n);
    for (int v = 0; v < n; ++v) indeg[v] = (int)pred[v].size();
    std::vector<int> heap;
    auto cmp = [](int a, int b) { return a > b; };
    for (int v = 0; v < n; ++v) if (!indeg[v]) heap.push_back(v);
    std::make_heap(heap.begin(), heap.end(), cmp);
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      int v = heap.back(); heap.pop_back();
      order.push_back(v);
      for (int w : succ[v]) if (--indeg[w] == 0) { heap.push_back(w); std::push_heap(heap.begin(), heap.end(), cmp); }
    }
    if ((int)order.size() != n) { bist_set_error("bist_graph_split_plan: the graph has a cycle"); return BIST_EINVAL; }
  }
  // labels: given ones are kept; an unlabelled node (a launch the caller did not see: a framework-internal fill or copy) continues the
  // chain of a predecessor that is still the tail of its chain, else joins its first predecessor's chain, else the main chain
  P.n = n; P.C = C; P.main_chain = main_chain;
  P.label.assign(n, -1); P.pos.assign(n, -1); P.flag_of.assign(n, -1);
  std::vector<int> tail(C, -1), count(C, 0);
  std::vector<std::vector<int>> chain(C);
  for (int v : order) {
    int l = labels_in ? labels_in[v] : -1;
    if (l >= C) { bist_set_error("bist_graph_split_plan: label %d of node %d >= %d chains", l, v, C); return BIST_EINVAL; }
    if (l < 0) {
      for (int p : pred[v]) if (tail[P.label[p]] == p) { l = P.label[p]; break; }
      if (l < 0) l = pred[v].empty() ? main_chain : P.label[pred[v][0]];
    }
    P.label[v] = l; P.pos[v] = count[l]++; tail[l] = v; chain[l].push_back(v);
  }
  // vector clocks: vc[v][c] = the last position of chain c known to have finished before v starts
  std::vector<std::array<int, MAXC>> vc(n);
  std::vector<std::vector<int>> needs(n);
  std::vector<char> signalled(n, 0);
  for (int v : order) {
    const int l = P.label[v];
    std::array<int, MAXC> cur; cur.fill(-1);
    if (P.pos[v] > 0) { int pv = chain[l][P.pos[v] - 1]; cur = vc[pv]; cur[l] = P.pos[pv]; }
    // side chains start behind the main chain's START signal (its position -1): nothing to inherit
    std::array<int, MAXC> best; best.fill(-1);          // per foreign chain the latest predecessor
    for (int p : pred[v]) { int a = P.label[p]; if (a != l && (best[a] < 0 || P.pos[p] > P.pos[best[a]])) best[a] = p; }
    std::vector<int> cand;
    for (int a = 0; a < C; ++a) if (best[a] >= 0) cand.push_back(best[a]);
    // the predecessor that knows most first, so that it can make the others redundant
    std::sort(cand.begin(), cand.end(), [&](int x, int y) { int sx = 0, sy = 0; for (int c = 0; c < C; ++c) { sx += vc[x][c]; sy += vc[y][c]; } return sx + P.pos[x] > sy + P.pos[y]; });
    for (int u : cand) {
      int a = P.label[u];
      if (P.pos[u] <= cur[a]) continue;                  // implied by an earlier wait of this chain (or of this node)
      needs[v].push_back(u); signalled[u] = 1;
      for (int c = 0; c < C; ++c) cur[c] = std::max(cur[c], vc[u][c]);
      cur[a] = std::max(cur[a], P.pos[u]);
    }
    vc[v] = cur;
  }
  // flags and sequences
  int nf = C;                                             // 0 = START, c = END of side chain c
  for (int v = 0; v < n; ++v) if (signalled[v]) P.flag_of[v] = nf++;
  P.n_flags = nf;
  P.seq.assign(C, {});
  auto wait_items = [&](std::vector<Item>& s, const std::vector<int>& fl) {
    for (size_t i = 0; i < fl.size(); i += MAXW) {
      Item w{3, -1, {-1, -1, -1, -1}};
      for (size_t j = 0; j < MAXW && i + j < fl.size(); ++j) w.flags[j] = fl[i + j];
      s.push_back(w); ++P.n_wait;
    }
  };
  for (int c = 0; c < C; ++c) {
    auto& s = P.seq[c];
    s.push_back(Item{1, -1, {-1, -1, -1, -1}});
    if (c == main_chain) { s.push_back(Item{2, 0, {-1, -1, -1, -1}}); ++P.n_signal; }
    else wait_items(s, {0});
    for (int v : chain[c]) {
      if (!needs[v].empty()) { std::vector<int> fl; for (int u : needs[v]) fl.push_back(P.flag_of[u]); wait_items(s, fl); }
      s.push_back(Item{0, v, {-1, -1, -1, -1}});
      if (signalled[v]) { s.push_back(Item{2, P.flag_of[v], {-1, -1, -1, -1}}); ++P.n_signal; }
    }
    if (c != main_chain) { s.push_back(Item{2, c == 0 ? main_chain : c, {-1, -1, -1, -1}}); ++P.n_signal; }
  }
  // (END flag of side chain c: word c, except that chain 0 -- when it is a side chain -- uses the main chain's otherwise unused word)
  std::vector<int> ends;
  for (int c = 0; c < C; ++c) if (c != main_chain) ends.push_back(c == 0 ? main_chain : c);
  if (!ends.empty()) wait_items(P.seq[main_chain], ends);
  return BIST_OK;
}

}  // namespace

// Flat form of the plan for tests (no device): out = per chain "-1, chain", then per item "kind, a, f0, f1, f2, f3" (kind 0: a = node;
// 1: bump; 2: a = flag; 3: flags f0..f3), terminated by -2.  Returns the number of ints the plan needs (also when cap is too small).
extern "C" int64_t bist_graph_split_plan(int32_t n_nodes, const int32_t* edge_from, const int32_t* edge_to, int32_t n_edges, const int32_t* labels,
                                         int32_t n_chains, int32_t main_chain, int32_t* out, int64_t cap) {
  Plan P;
  if (make_plan(n_nodes, edge_from, edge_to, n_edges, labels, n_chains, main_chain, P) != BIST_OK) return -1;
  int64_t k = 0;
  auto put = [&](int v) { if (out && k < cap) out[k] = v; ++k; };
  for (int c = 0; c < P.C; ++c) {
    put(-1); put(c);
    for (const Item& it : P.seq[c]) { put(it.kind); put(it.node); for (int j = 0; j < MAXW; ++j) put(it.flags[j]); }
  }
  put(-2); put(P.n_flags);
  return k;
}

// ---- the device side: clones, pruning, sync launches, instantiation --------------------------------------------------------------
struct BistGraphSplit {
  Plan plan;
  std::vector<hipGraph_t> graphs;
  std::vector<hipGraphExec_t> execs;
  unsigned long long* words = nullptr;       // caller-owned device memory: [C epochs][1 error][n_flags flags]
  int n_nodes = 0;
};

#define GS_HIP(call, what)                                                                     \
  do {                                                                                         \
    hipError_t e__ = (call);                                                                   \
    if (e__ != hipSuccess) {                                                                   \
      bist_set_error("bist_graph_split: %s: %s", what, hipGetErrorString(e__));                \
      return BIST_ELAUNCH;                                                                     \
    }                                                                                          \
  } while (0)

extern "C" int bist_graph_capture_tail(void* stream, void** node_out) {
  BIST_REQUIRE(node_out != nullptr, "bist_graph_capture_tail: null output");
  *node_out = nullptr;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t nd = 0;
  GS_HIP(hipStreamGetCaptureInfo_v2(static_cast<hipStream_t>(stream), &st, &id, &g, &deps, &nd), "hipStreamGetCaptureInfo_v2");
  if (st == hipStreamCaptureStatusActive && nd == 1) *node_out = deps[0];
  return BIST_OK;
}

extern "C" int bist_graph_nodes(void* graph, void** nodes_out, int32_t cap, int32_t* n_out) {
  BIST_REQUIRE(graph && n_out, "bist_graph_nodes: null argument");
  size_t n = 0;
  GS_HIP(hipGraphGetNodes(static_cast<hipGraph_t>(graph), nullptr, &n), "hipGraphGetNodes");
  *n_out = (int32_t)n;
  if (nodes_out && cap >= (int32_t)n && n) {
    std::vector<hipGraphNode_t> v(n);
    GS_HIP(hipGraphGetNodes(static_cast<hipGraph_t>(graph), v.data(), &n), "hipGraphGetNodes");
    for (size_t i = 0; i < n; ++i) nodes_out[i] = v[i];
  }
  return BIST_OK;
}

// The edges of a hipGraph as index pairs into bist_graph_nodes' order (analysis aid: scripts/critical_path.py)
extern "C" int bist_graph_edges(void* graph, int32_t* from_out, int32_t* to_out, int32_t cap, int32_t* n_out) {
  BIST_REQUIRE(graph && n_out, "bist_graph_edges: null argument");
  hipGraph_t g = static_cast<hipGraph_t>(graph);
  size_t n = 0, ne = 0;
  GS_HIP(hipGraphGetNodes(g, nullptr, &n), "hipGraphGetNodes");
  std::vector<hipGraphNode_t> nodes(n);
  if (n) GS_HIP(hipGraphGetNodes(g, nodes.data(), &n), "hipGraphGetNodes");
  GS_HIP(hipGraphGetEdges(g, nullptr, nullptr, &ne), "hipGraphGetEdges");
  *n_out = (int32_t)ne;
  if (!from_out || !to_out || cap < (int32_t)ne || !ne) return BIST_OK;
  std::vector<hipGraphNode_t> ef(ne), et(ne);
  GS_HIP(hipGraphGetEdges(g, ef.data(), et.data(), &ne), "hipGraphGetEdges");
  std::vector<std::pair<hipGraphNode_t, int>> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = {nodes[i], (int)i};
  std::sort(idx.begin(), idx.end());
  auto find = [&](hipGraphNode_t p) { auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(p, -1)); return (it != idx.end() && it->first == p) ? it->second : -1; };
  for (size_t e = 0; e < ne; ++e) { from_out[e] = find(ef[e]); to_out[e] = find(et[e]); }
  return BIST_OK;
}

// The whole plan in the flat form of bist_graph_split_plan (analysis aid)
extern "C" int64_t bist_graph_split_dump(const BistGraphSplit* S, int32_t* out, int64_t cap) {
  if (!S) return -1;
  const Plan& P = S->plan;
  int64_t k = 0;
  auto put = [&](int v) { if (out && k < cap) out[k] = v; ++k; };
  for (int c = 0; c < P.C; ++c) {
    put(-1); put(c);
    for (const Item& it : P.seq[c]) { put(it.kind); put(it.node); for (int j = 0; j < MAXW; ++j) put(it.flags[j]); }
  }
  put(-2); put(P.n_flags);
  return k;
}

extern "C" int bist_graph_split_create(void* graph, const int32_t* labels, int32_t n_labels, int32_t n_chains, int32_t main_chain,
                                       BistGraphSplit** out) {
  BIST_REQUIRE(graph && out, "bist_graph_split_create: null argument");
  *out = nullptr;
  hipGraph_t g = static_cast<hipGraph_t>(graph);
  size_t n = 0, ne = 0;
  GS_HIP(hipGraphGetNodes(g, nullptr, &n), "hipGraphGetNodes");
  BIST_REQUIRE((int32_t)n == n_labels && n > 0, "bist_graph_split_create: %d labels for a graph of %zu nodes", n_labels, n);
  std::vector<hipGraphNode_t> nodes(n);
  GS_HIP(hipGraphGetNodes(g, nodes.data(), &n), "hipGraphGetNodes");
  GS_HIP(hipGraphGetEdges(g, nullptr, nullptr, &ne), "hipGraphGetEdges");
  std::vector<hipGraphNode_t> ef(ne), et(ne);
  if (ne) GS_HIP(hipGraphGetEdges(g, ef.data(), et.data(), &ne), "hipGraphGetEdges");
  std::vector<std::pair<hipGraphNode_t, int>> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = {nodes[i], (int)i};
  std::sort(idx.begin(), idx.end());
  auto find = [&](hipGraphNode_t p) { auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(p, -1)); return (it != idx.end() && it->first == p) ? it->second : -1; };
  std::vector<int> f(ne), t(ne);
  for (size_t e = 0; e < ne; ++e) { f[e] = find(ef[e]); t[e] = find(et[e]); BIST_REQUIRE(f[e] >= 0 && t[e] >= 0, "bist_graph_split_create: an edge names a node outside the graph"); }
  BistGraphSplit* S = new BistGraphSplit();
  S->n_nodes = (int)n;
  int rc = make_plan((int)n, f.data(), t.data(), (int)ne, labels, n_chains, main_chain, S->plan);
  if (rc != BIST_OK) { delete S; return rc; }
  // keep the source graph's handle out of the object: build() clones from the caller's graph again
  *out = S;
  return BIST_OK;
}

extern "C" int64_t bist_graph_split_sync_words(const BistGraphSplit* S) { return S ? (int64_t)S->plan.C + 1 + S->plan.n_flags + 2 * (int64_t)S->plan.n_sync() : -1; }

// The sync launches in the order of their stamp words: per launch 8 ints -- chain, kind (1 bump, 2 signal, 3 wait), index of the captured
// node it follows (signal) / precedes (wait) or -1, flag ids f0..f3 (signal: f0) -- so that the stamps can be read as a timeline.
extern "C" int64_t bist_graph_split_sync_items(const BistGraphSplit* S, int32_t* out, int64_t cap) {
  if (!S) return -1;
  int64_t k = 0;
  auto put = [&](int v) { if (out && k < cap) out[k] = v; ++k; };
  for (int c = 0; c < S->plan.C; ++c) {
    const auto& s = S->plan.seq[c];
    for (size_t i = 0; i < s.size(); ++i) {
      const Item& it = s[i];
      if (it.kind == 0) continue;
      int near = -1;
      if (it.kind == 2 && i > 0 && s[i - 1].kind == 0) near = s[i - 1].node;
      if (it.kind == 3) for (size_t j = i + 1; j < s.size(); ++j) if (s[j].kind == 0) { near = s[j].node; break; }
      put(c); put(it.kind); put(near);
      if (it.kind == 2) { put(it.node); put(-1); put(-1); put(-1); } else for (int j = 0; j < MAXW; ++j) put(it.flags[j]);
      put(0);
    }
  }
  return k;
}

// counts: [chains, captured nodes, wait launches, signal launches, flags]
extern "C" int bist_graph_split_info(const BistGraphSplit* S, int32_t* out5, int32_t* nodes_per_chain) {
  BIST_REQUIRE(S && out5, "bist_graph_split_info: null argument");
  out5[0] = S->plan.C; out5[1] = S->plan.n; out5[2] = S->plan.n_wait; out5[3] = S->plan.n_signal; out5[4] = S->plan.n_flags;
  if (nodes_per_chain) for (int c = 0; c < S->plan.C; ++c) { int k = 0; for (const Item& it : S->plan.seq[c]) k += it.kind == 0; nodes_per_chain[c] = k; }
  return BIST_OK;
}

static int add_kernel_node(hipGraph_t g, hipGraphNode_t prev, void* func, void** args, hipGraphNode_t* out) {
  hipKernelNodeParams p{};
  p.blockDim = dim3(1, 1, 1); p.gridDim = dim3(1, 1, 1); p.sharedMemBytes = 0; p.func = func; p.kernelParams = args; p.extra = nullptr;
  GS_HIP(hipGraphAddKernelNode(out, g, prev ? &prev : nullptr, prev ? 1 : 0, &p), "hipGraphAddKernelNode (sync launch)");
  return BIST_OK;
}

extern "C" int bist_graph_split_build(BistGraphSplit* S, void* graph, void* sync_words, int64_t timeout_ticks) {
  BIST_REQUIRE(S && graph && sync_words, "bist_graph_split_build: null argument");
  BIST_REQUIRE(S->execs.empty(), "bist_graph_split_build: already built");
  BIST_REQUIRE(timeout_ticks > 0, "bist_graph_split_build: the waits need a time-out (ticks of the 100 MHz device clock)");
  hipGraph_t g = static_cast<hipGraph_t>(graph);
  const Plan& P = S->plan;
  size_t n = 0;
  GS_HIP(hipGraphGetNodes(g, nullptr, &n), "hipGraphGetNodes");
  BIST_REQUIRE((int)n == P.n, "bist_graph_split_build: the graph changed since the plan (%zu nodes, planned %d)", n, P.n);
  std::vector<hipGraphNode_t> nodes(n);
  GS_HIP(hipGraphGetNodes(g, nodes.data(), &n), "hipGraphGetNodes");
  S->words = static_cast<unsigned long long*>(sync_words);
  unsigned long long* epoch = S->words;
  unsigned long long* err = S->words + P.C;
  unsigned long long* flags = S->words + P.C + 1;
  unsigned long long* stamp = flags + P.n_flags;          // two words per sync launch, in chain-then-sequence order
  for (int c = 0; c < P.C; ++c) {
    hipGraph_t gc = nullptr;
    GS_HIP(hipGraphClone(&gc, g), "hipGraphClone");
    S->graphs.push_back(gc);
    std::vector<hipGraphNode_t> cn(n);
    for (size_t i = 0; i < n; ++i) GS_HIP(hipGraphNodeFindInClone(&cn[i], nodes[i], gc), "hipGraphNodeFindInClone");
    for (size_t i = 0; i < n; ++i) if (P.label[i] != c) GS_HIP(hipGraphDestroyNode(cn[i]), "hipGraphDestroyNode");
    size_t ne = 0;
    GS_HIP(hipGraphGetEdges(gc, nullptr, nullptr, &ne), "hipGraphGetEdges (clone)");
    if (ne) {
      std::vector<hipGraphNode_t> ef(ne), et(ne);
      GS_HIP(hipGraphGetEdges(gc, ef.data(), et.data(), &ne), "hipGraphGetEdges (clone)");
      GS_HIP(hipGraphRemoveDependencies(gc, ef.data(), et.data(), ne), "hipGraphRemoveDependencies");
    }
    hipGraphNode_t prev = nullptr;
    for (const Item& it : P.seq[c]) {
      hipGraphNode_t node = nullptr;
      if (it.kind == 0) {
        node = cn[it.node];
        if (prev) GS_HIP(hipGraphAddDependencies(gc, &prev, &node, 1), "hipGraphAddDependencies");
      } else if (it.kind == 1) {
        unsigned long long* e = epoch + c;
        void* args[] = {&e, &stamp};
        int rc = add_kernel_node(gc, prev, reinterpret_cast<void*>(gs_bump_kernel), args, &node);
        if (rc != BIST_OK) return rc;
        stamp += 2;
      } else if (it.kind == 2) {
        unsigned long long* fl = flags + it.node;
        const unsigned long long* e = epoch + c;
        void* args[] = {&fl, &e, &stamp};
        int rc = add_kernel_node(gc, prev, reinterpret_cast<void*>(gs_signal_kernel), args, &node);
        if (rc != BIST_OK) return rc;
        stamp += 2;
      } else {
        const unsigned long long* f[MAXW];
        for (int j = 0; j < MAXW; ++j) f[j] = it.flags[j] >= 0 ? flags + it.flags[j] : nullptr;
        const unsigned long long* e = epoch + c;
        unsigned long long fixed = 0, tmo = (unsigned long long)timeout_ticks;
        void* args[] = {&f[0], &f[1], &f[2], &f[3], &e, &fixed, &err, &tmo, &stamp};
        int rc = add_kernel_node(gc, prev, reinterpret_cast<void*>(gs_wait_kernel), args, &node);
        if (rc != BIST_OK) return rc;
        stamp += 2;
      }
      prev = node;
    }
    hipGraphExec_t ex = nullptr;
    GS_HIP(hipGraphInstantiate(&ex, gc, nullptr, nullptr, 0), "hipGraphInstantiate");
    S->execs.push_back(ex);
  }
  return BIST_OK;
}

// streams[c]: the stream chain c is launched into; side chains first, the main chain last (its START signal releases them)
extern "C" int bist_graph_split_launch(BistGraphSplit* S, void* const* streams) {
  BIST_REQUIRE(S && streams && (int)S->execs.size() == S->plan.C, "bist_graph_split_launch: not built");
  for (int c = 0; c < S->plan.C; ++c)
    if (c != S->plan.main_chain) GS_HIP(hipGraphLaunch(S->execs[c], static_cast<hipStream_t>(streams[c])), "hipGraphLaunch (side chain)");
  GS_HIP(hipGraphLaunch(S->execs[S->plan.main_chain], static_cast<hipStream_t>(streams[S->plan.main_chain])), "hipGraphLaunch (main chain)");
  return BIST_OK;
}

// development aid: one chain's launch alone
extern "C" int bist_graph_split_launch_chain(BistGraphSplit* S, int32_t chain, void* stream) {
  BIST_REQUIRE(S && chain >= 0 && chain < (int)S->execs.size(), "bist_graph_split_launch_chain: not built / no such chain");
  GS_HIP(hipGraphLaunch(S->execs[chain], static_cast<hipStream_t>(stream)), "hipGraphLaunch");
  return BIST_OK;
}

extern "C" void bist_graph_split_destroy(BistGraphSplit* S) {
  if (!S) return;
  for (auto e : S->execs) (void)hipGraphExecDestroy(e);
  for (auto g : S->graphs) (void)hipGraphDestroy(g);
  delete S;
}

// Do two streams sit on hardware queues that run side by side?  Into `a`: a wait for a flag and, behind it, one more launch (so that
// a's queue holds a packet that cannot start until the wait has ended -- the state every chain of a split graph is in most of the
// time); into `b`, afterwards: the launch that sets the flag.  If b shares a's queue, or a's queue holds up the hardware pipe that also
// serves b's, the signal cannot start before the wait has given up.  scratch: 4 uint64 of device memory, zeroed by the caller; after
// both streams have drained scratch[1] == 0 iff the wait saw the flag in time; timeout_ticks of the 100 MHz clock.
extern "C" int bist_graph_queues_distinct(void* stream_a, void* stream_b, void* scratch, int64_t timeout_ticks) {
  BIST_REQUIRE(scratch && timeout_ticks > 0, "bist_graph_queues_distinct: null scratch / no time-out");
  unsigned long long* w = static_cast<unsigned long long*>(scratch);
  gs_wait_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream_a)>>>(w, nullptr, nullptr, nullptr, nullptr, 1ULL, w + 1, (unsigned long long)timeout_ticks, nullptr);
  BIST_LAUNCH_CHECK("bist_graph_queues_distinct (wait)");
  gs_bump_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream_a)>>>(w + 2, nullptr);
  BIST_LAUNCH_CHECK("bist_graph_queues_distinct (pending)");
  gs_bump_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream_b)>>>(w, nullptr);
  BIST_LAUNCH_CHECK("bist_graph_queues_distinct (signal)");
  return BIST_OK;
}

// ---- ready flags of the overlapped gradient exchange (bist_amd/train.py) ---------------------------------------------------
// The captured step writes the step number into a flag IN PINNED HOST MEMORY, on every stream it runs on, at the point where a bucket
// of gradients (the matrices of the last layers, whose backward runs first) is final on that stream; the host, which has queued the
// whole step, polls the bucket's flags and issues its all-reduce the moment they arrive -- under the rest of the backward pass.  (A wait
// KERNEL on an exchange stream was measured first: a resident wave on a fifth hardware queue slows the chain that shares its dispatch
// pipe threefold per launch, and where RCCL's own stream lands is not ours to choose: 9.7 - 20 ms per step against 8.6.)
__global__ void flag_signal_host_kernel(unsigned long long* flag, const unsigned long long* value) {
  __hip_atomic_store(flag, *value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bist_flag_signal: *flag = *value_dev (64-bit; flag in host-coherent memory, value in device memory) on `stream`, a system-scope release
// store behind the stream's earlier launches; capturable.
extern "C" int bist_flag_signal(void* flag, const void* value_dev, void* stream) {
  BIST_REQUIRE(flag && value_dev, "bist_flag_signal: null flag / value");
  flag_signal_host_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(static_cast<unsigned long long*>(flag), static_cast<const unsigned long long*>(value_dev));
  BIST_LAUNCH_CHECK("bist_flag_signal");
  return BIST_OK;
}

// development aid: a one-wave launch that stays resident (see gs_idle_kernel)
extern "C" int bist_dev_idle_wave(void* stream, int64_t ticks, int32_t mode, void* word) {
  BIST_REQUIRE(word != nullptr && ticks > 0 && mode >= 0 && mode <= 3, "bist_dev_idle_wave: bad argument");
  unsigned long long* w = static_cast<unsigned long long*>(word);
  gs_idle_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>((unsigned long long)ticks, mode, w, w + 1);
  BIST_LAUNCH_CHECK("bist_dev_idle_wave");
  return BIST_OK;
}

// Do two hardware queues get in each other's way?  A linear graph of n one-thread launches replayed into `stream`, timed with events:
// alone (resident_stream null) or while one wave stays resident on `resident_stream` for resident_ticks of the 100 MHz clock.  Measured
// on MI355X (profiles/r04_queue_pairs.txt): with eight hardware queues the queues come in PAIRS that share a dispatch pipe -- a
// resident wave on one triples the launch-to-launch time on its partner (2.1 -> 6.2 us) and leaves the other six alone; a chain of a
// split graph spends much of its life resident in a wait, so chains must sit on different pipes.  word: 2 uint64 of device memory.
// Synchronises both streams.  us_per_launch_out: the measured microseconds per launch.
extern "C" int bist_graph_queue_pace(void* stream, int32_t n, void* resident_stream, int64_t resident_ticks, void* word, float* us_per_launch_out) {
  BIST_REQUIRE(stream && n >= 8 && n <= 4096 && word && us_per_launch_out, "bist_graph_queue_pace: bad argument (a non-null stream, 8..4096 launches)");
  unsigned long long* w = static_cast<unsigned long long*>(word);
  hipGraph_t g = nullptr;
  GS_HIP(hipGraphCreate(&g, 0), "hipGraphCreate");
  hipGraphNode_t prev = nullptr;
  unsigned long long* e = w;
  unsigned long long* nostamp = nullptr;
  for (int i = 0; i < n; ++i) {
    void* args[] = {&e, &nostamp};
    hipGraphNode_t node = nullptr;
    int rc = add_kernel_node(g, prev, reinterpret_cast<void*>(gs_bump_kernel), args, &node);
    if (rc != BIST_OK) { (void)hipGraphDestroy(g); return rc; }
    prev = node;
  }
  hipGraphExec_t ex = nullptr;
  hipError_t er = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (er != hipSuccess) { (void)hipGraphDestroy(g); bist_set_error("bist_graph_queue_pace: hipGraphInstantiate: %s", hipGetErrorString(er)); return BIST_ELAUNCH; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = BIST_OK;
  float ms = 0.f;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = BIST_ELAUNCH; bist_set_error("bist_graph_queue_pace: hipEventCreate failed"); }
  if (rc == BIST_OK && hipGraphLaunch(ex, st) != hipSuccess) rc = BIST_ELAUNCH;          // warm-up
  if (rc == BIST_OK) (void)hipStreamSynchronize(st);
  if (rc == BIST_OK && resident_stream) {
    gs_idle_kernel<<<1, 1, 0, static_cast<hipStream_t>(resident_stream)>>>((unsigned long long)resident_ticks, 0, w, w + 1);
    if (hipGetLastError() != hipSuccess) rc = BIST_ELAUNCH;
  }
  if (rc == BIST_OK && (hipEventRecord(e0, st) != hipSuccess || hipGraphLaunch(ex, st) != hipSuccess || hipEventRecord(e1, st) != hipSuccess)) rc = BIST_ELAUNCH;
  if (rc == BIST_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = BIST_ELAUNCH;
  if (resident_stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(resident_stream));
  if (rc != BIST_OK && !*bist_last_error()) bist_set_error("bist_graph_queue_pace: a runtime call failed");
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(ex);
  (void)hipGraphDestroy(g);
  *us_per_launch_out = ms * 1e3f / (float)n;
  return rc;
}
