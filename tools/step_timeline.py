#!/usr/bin/env python3
"""Timeline of the last hot-path step in a rocprofv3 kernel_trace.csv: every dispatch from the last large seeds launch to the
next one (or the end of the trace) with its start offset, duration and the idle gap before it; plus a per-kernel summary over
the last `--steps` steps."""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.match(r"(void )?([\w:<>, ]+?)\(", name)
    return (m.group(2) if m else name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--anchor", default="seeds_slots_kernel")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--summary-only", action="store_true", help="steps overlap (pipelined run): only the per-kernel table")
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))
    anchors = [i for i, r in enumerate(rows) if a.anchor in r["Kernel_Name"]]
    gmax = max(int(rows[i]["Grid_Size_X"]) for i in anchors)
    anchors = [i for i in anchors if int(rows[i]["Grid_Size_X"]) == gmax]
    last = anchors[-a.steps:]
    if a.summary_only:
        return summary(rows, last)
    # timeline of the last complete step (second to last anchor .. last anchor)
    lo, hi = (anchors[-2], anchors[-1]) if len(anchors) >= 2 else (anchors[-1], len(rows))
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = t0
    print(f"## one step: dispatches {lo}..{hi - 1}, wall {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us" if hi < len(rows) else "## one step")
    print("| start_us | dur_us | gap_before_us | kernel | grid |")
    print("|---|---|---|---|---|")
    busy = 0.0
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"| {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {max(0, s - prev_end) / 1e3:.1f} | `{short(r['Kernel_Name'])}` | {r['Grid_Size_X']} |")
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    print(f"\nkernel time in the step: {busy:.1f} us")
    summary(rows, last)


def summary(rows, last):
    agg = collections.defaultdict(list)
    for r in rows[last[0]:]:
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"\n## per kernel over the last {len(last)} steps")
    print("| kernel | grid | calls | total_us | avg_us | min_us | max_us |")
    print("|---|---|---|---|---|---|---|")
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print(f"| `{k}` | {g} | {len(v)} | {sum(v):.1f} | {sum(v) / len(v):.2f} | {min(v):.2f} | {max(v):.2f} |")


if __name__ == "__main__":
    main()
