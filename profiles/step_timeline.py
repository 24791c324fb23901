#!/usr/bin/env python3
"""Where is the chip idle inside one training step?  From a rocprofv3 kernel trace of bench.py: per queue the busy time of one
steady-state step, the union of all queues, the gaps on the caller's queue and the tail after its last kernel.
   python profiles/step_timeline.py gpurun_out/<tag>/prof/r_kernel_trace.csv [step index from the end, default 3]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for r in rows:
    r["s"], r["e"], r["q"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
rows.sort(key=lambda r: r["s"])
# a step starts with the encoder's first chain launch after the optimizer-less bench step: use the decoder's small wgrad as the marker
marks = [i for i, r in enumerate(rows) if "k_sim_loss" in r["Kernel_Name"] or "k_sim_l" in r["Kernel_Name"]]
a, b = marks[-back - 1], marks[-back]
step = rows[a:b]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
print(f"step window {(rows[b]['s'] - t0) / 1e3:.0f} us between two loss kernels, {len(step)} launches")
byq = collections.defaultdict(list)
for r in step: byq[r["q"]].append(r)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = union([(r["s"], r["e"]) for r in rs])
    print(f"queue {q}: {len(rs):3d} launches, busy {busy / 1e3:7.0f} us, first {(rs[0]['s'] - t0) / 1e3:7.0f}, last end {(max(r['e'] for r in rs) - t0) / 1e3:7.0f}")
print(f"union of all queues busy {union([(r['s'], r['e']) for r in step]) / 1e3:.0f} us")
main = max(byq.values(), key=len)
gaps = sorted(((main[i + 1]["s"] - main[i]["e"]) / 1e3, main[i]["Kernel_Name"][:40], main[i + 1]["Kernel_Name"][:40]) for i in range(len(main) - 1))
print("largest gaps on the caller's queue (us, after, before):")
for g in gaps[-12:][::-1]: print(f"  {g[0]:6.1f}  {g[1]}  ->  {g[2]}")
print(f"sum of gaps {sum(g[0] for g in gaps):.0f} us over {len(gaps)} boundaries; median {sorted(g[0] for g in gaps)[len(gaps) // 2]:.1f}")
# chip occupancy proxy: time during which ONLY side-queue kernels run
mi = [(r["s"], r["e"]) for r in main]
side = [(r["s"], r["e"]) for r in step if r["q"] != main[0]["q"]]
allu, mainu = union(mi + side), union(mi)
print(f"caller's queue busy {mainu / 1e3:.0f} us; time with only side-lane kernels running {(allu - mainu) / 1e3:.0f} us")
