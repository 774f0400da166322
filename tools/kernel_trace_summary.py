#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid size) calls / total / avg / min / max in microseconds,
restricted to the last `--tail-ms` of the trace if given (to cut the benchmark's setup phase off)."""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.match(r"(void )?([\w:<>, ]+?)\(", name)
    n = m.group(2) if m else name
    return n[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail-ms", type=float, default=None)
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    t_end = max(int(r["End_Timestamp"]) for r in rows)
    if a.tail_ms:
        rows = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - a.tail_ms * 1e6]
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(f"{len(rows)} dispatches, {tot / 1e3:.3f} ms of kernel time" + (f" in the last {a.tail_ms} ms" if a.tail_ms else ""))
    print("| kernel | grid (threads) | calls | total_us | avg_us | min_us | max_us | % |")
    print("|---|---|---|---|---|---|---|---|")
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: a.top]:
        print(f"| `{k}` | {g} | {len(v)} | {sum(v):.1f} | {sum(v) / len(v):.2f} | {min(v):.2f} | {max(v):.2f} | {100 * sum(v) / tot:.1f} |")


if __name__ == "__main__":
    main()
