#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv of the pipelined bench: how the dominant kernel's launches follow each other.  The pipeline runs
one seeding kernel at a time (a worker's stream waits for the event behind the previous worker's kernel): the time between the END of one
reads_kernel and the START of the next is GPU time the dominant kernel does not use.  Prints the distribution of those gaps, the share
of the wall clock with a reads_kernel running, and which kernels ran inside the gaps."""
import argparse
import csv

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--kernel", default="reads_kernel")
    ap.add_argument("--last", type=int, default=200, help="launches of the kernel to look at (the end of the trace: the timed region)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
    k = [e for e in ev if a.kernel in e[2]][-a.last:]
    if len(k) < 3:
        print("too few launches"); return
    gaps = np.array([(k[i + 1][0] - k[i][1]) / 1e3 for i in range(len(k) - 1)])          # us; negative = overlap
    dur = np.array([(e[1] - e[0]) / 1e3 for e in k])
    period = np.array([(k[i + 1][0] - k[i][0]) / 1e3 for i in range(len(k) - 1)])
    wall = (k[-1][1] - k[0][0]) / 1e3
    print(f"{len(k)} launches of {a.kernel} over {wall / 1e3:.3f} ms: duration mean {dur.mean():.1f} us (min {dur.min():.1f}), start-to-start {period.mean():.1f} us")
    print(f"end-to-next-start gap: mean {gaps.mean():.1f} us, p10 {np.percentile(gaps, 10):.1f}, p50 {np.percentile(gaps, 50):.1f}, p90 {np.percentile(gaps, 90):.1f}, max {gaps.max():.1f}; "
          f"share of the wall clock with the kernel running {dur.sum() / wall:.3f}")
    print(f"queues of consecutive launches differ in {sum(k[i][3] != k[i + 1][3] for i in range(len(k) - 1))} of {len(k) - 1} cases")
    # what ran inside the gaps
    inside = {}
    for i in range(len(k) - 1):
        g0, g1 = k[i][1], k[i + 1][0]
        if g1 <= g0:
            continue
        for s, e, n, q in ev:
            if e <= g0 or s >= g1 or a.kernel in n:
                continue
            nm = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("sylph::", "").split("(")[0][:40]
            inside[nm] = inside.get(nm, 0.0) + (min(e, g1) - max(s, g0)) / 1e3
    tot_gap = gaps[gaps > 0].sum()
    print(f"kernel time overlapping the gaps ({tot_gap:.0f} us of gaps in all):")
    for nm, t in sorted(inside.items(), key=lambda kv: -kv[1])[:8]:
        print(f"  {nm:40s} {t:9.1f} us")


if __name__ == "__main__":
    main()
