"""Critical path of one replayed training step under the split-graph executor (analysis aid, round 4).

inputs: the plan dumped by SplitGraph.dump (chain sequences, the captured graph's edges) and a rocprofv3 kernel trace of the same process
(scripts/bench_step.py): per hardware queue the launches of one steady-state step in order = the chain's sequence, so every item gets
its kernel name and duration.  The longest path through {chain order, signal -> wait pairs} with cost = kernel time + GAP per launch is
the time the step cannot beat on this partition; the listing says which launches it runs through and how much of it each chain carries.
usage: critical_path.py plan.json trace.csv [gap_us]"""
import collections, csv, json, sys
plan = json.load(open(sys.argv[1]))
GAP = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
flat = plan["plan"]
seqs, i = {}, 0
while flat[i] != -2:
    c = flat[i + 1]; i += 2
    seqs[c] = []
    while flat[i] not in (-1, -2):
        seqs[c].append(tuple(flat[i:i + 6])); i += 6
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
C = len(seqs)
byq_all = collections.defaultdict(list)
for r in rows:
    byq_all[r["Queue_Id"]].append(r)
# per queue the launches from its third-last epoch bump on: one steady-state step of that chain
byq = {}
for q, rs in byq_all.items():
    bumps = [k for k, r in enumerate(rs) if "gs_bump" in r["Kernel_Name"]]
    if len(bumps) >= 4:
        byq[q] = rs[bumps[-3]:bumps[-2]]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")
    return n.split("(")[0].split("<")[0][:34]
# match queues to chains by launch count (a chain's queue also carries a few eager launches of the step loop: tolerate extras at the ends)
dur, name = {}, {}
for c, s in seqs.items():
    want = len(s)
    q = min(byq, key=lambda q_: abs(len(byq[q_]) - want))
    rs = byq[q][:want]
    assert len(rs) == want, (c, q, len(rs), want)
    for k, (it, r) in enumerate(zip(s, rs)):
        kind = it[0]
        nm = short(r["Kernel_Name"])
        assert (kind == 1) == ("gs_bump" in nm) and (kind == 2) == ("gs_signal" in nm) and (kind == 3) == ("gs_wait" in nm), (c, k, it, nm)
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        dur[(c, k)] = 1.0 if kind == 3 else d          # a wait costs its launch, not the time it stood blocked
        name[(c, k)] = nm
    del byq[q]
def longest(dur_):
    finish, pred = {}, {}
    pos = {c: 0 for c in seqs}
    progress = True
    while progress:
        progress = False
        for c, s_ in seqs.items():
            while pos[c] < len(s_):
                k = pos[c]; it = s_[k]
                cands = [((c, k - 1), finish[(c, k - 1)])] if k else []
                if it[0] == 3:
                    fl = [f for f in it[2:6] if f >= 0]
                    if not all(f in sig_at and sig_at[f] in finish for f in fl):
                        break
                    cands += [(sig_at[f], finish[sig_at[f]]) for f in fl]
                p, t = max(cands, key=lambda x: x[1]) if cands else (None, 0.0)
                cost = 0.0 if dur_[(c, k)] == 0.0 else GAP + dur_[(c, k)]
                finish[(c, k)] = t + cost; pred[(c, k)] = p
                pos[c] += 1; progress = True
    return finish, pred


sig_at = {}
for c, s_ in seqs.items():
    for k, it in enumerate(s_):
        if it[0] == 2:
            sig_at[it[1]] = (c, k)
finish, pred = longest(dur)
end = max(finish, key=finish.get)
print(f"chains {C}; modelled step {finish[end]:.0f} us with {GAP} us per launch gap; per chain: " +
      ", ".join(f"chain {c}: {len(s)} launches, {sum(dur[(c, k)] for k in range(len(s))):.0f} us of kernels" for c, s in seqs.items()))
path = []
cur = end
while cur is not None:
    path.append(cur); cur = pred[cur]
path.reverse()
share = collections.Counter(); by_name = collections.defaultdict(lambda: [0, 0.0])
for (c, k) in path:
    share[c] += dur[(c, k)] + GAP
    by_name[name[(c, k)]][0] += 1; by_name[name[(c, k)]][1] += dur[(c, k)]
print("critical path: %d launches; time on chain: %s" % (len(path), {c: round(v) for c, v in share.items()}))
print("by kernel on the critical path:")
for nm, (n, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"   {n:4d} x {t / n:7.1f} us = {t:8.1f} us  {nm}")
hops = sum(1 for a, b in zip(path, path[1:]) if a[0] != b[0])
print(f"chain switches on the path: {hops}")
if len(sys.argv) > 4:
    for (c, k) in path:
        print(f"  {finish[(c, k)]:9.1f}  chain {c} #{k:4d}  {dur[(c, k)]:7.1f}  {name[(c, k)]}")

# what-ifs: the modelled step with some launches gone (their work done elsewhere, off every chain) or faster
def what_if(label, f):
    d2 = {key: f(name[key], v) for key, v in dur.items()}
    fin, _ = longest(d2)
    print(f"what if {label}: {max(fin.values()):.0f} us")


what_if("the fusion logits' weight gradient left the chains", lambda n, v: 0.0 if "switch_logits_bwd_w" in n else v)
what_if("the closing reductions (col_sum_multi, layernorm_param_grad_multi) left the chains", lambda n, v: 0.0 if ("col_sum_multi" in n or "layernorm_param_grad" in n) else v)
what_if("both", lambda n, v: 0.0 if ("col_sum_multi" in n or "layernorm_param_grad" in n or "switch_logits_bwd_w" in n) else v)
what_if("both + the video gradients' sums (add_n)", lambda n, v: 0.0 if ("col_sum_multi" in n or "layernorm_param_grad" in n or "switch_logits_bwd_w" in n or "add_n" in n) else v)
what_if("every LayerNorm forward launch fused into its consumer", lambda n, v: 0.0 if n.startswith("layernorm_vec") else v)
what_if("every LayerNorm forward and backward launch fused away", lambda n, v: 0.0 if n.startswith("layernorm_vec") or n.startswith("layernorm_bwd") else v)
what_if("small products (t64 kernels) 30 % faster", lambda n, v: v * 0.7 if "gemm_t64" in n else v)
what_if("mha_core + mha_bwd 40 % faster", lambda n, v: v * 0.6 if "mha_" in n else v)
what_if("frame-grid products (gemm_fast / gemm_big) 25 % faster", lambda n, v: v * 0.75 if ("gemm_fast" in n or "gemm_big" in n) else v)
what_if("no signal launches (free cross-chain edges)", lambda n, v: 0.0 if "gs_signal" in n else v)
GAP_SAVE = GAP
GAP = 1.0
fin, _ = longest(dur); print(f"what if the launch gap were 1 us: {max(fin.values()):.0f} us")
GAP = GAP_SAVE
