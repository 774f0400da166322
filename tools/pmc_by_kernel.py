#!/usr/bin/env python3
"""Averages of rocprofv3 --pmc counters per (kernel, grid) over the last dispatches of each: `pmc_by_kernel.py DIR [DIR ...]` reads every
*counter_collection.csv below the directories (one counter group per directory) and prints one JSON object {kernel@grid: {counter: mean}}."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(void )?([\w:<>, ]+?)\(", name)
    return (m.group(2) if m else name).replace("sylph::", "")[:60]


res = collections.defaultdict(dict)
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 4
dirs = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--last"]
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, g, c), v in agg.items():
            v = v[-last:]
            res[f"{k}@{g}"][c] = sum(v) / len(v)
            res[f"{k}@{g}"]["dispatches"] = len(agg[(k, g, c)])
print(json.dumps(res, indent=1, sort_keys=True))
