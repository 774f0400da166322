#!/usr/bin/env python3
"""Per-kernel footprint of the library (VGPRs, static LDS, scratch) from hipcc's resource-usage remarks — what decides whether a workgroup of
one kernel gets onto a CU that the next samples' seeding kernels keep full.  Usage: python tools/kernel_footprints.py [file.hip ...]"""
import glob, os, re, subprocess, sys

src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sylph_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(src, "*.hip")))
rows = []
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, cwd=src)
    name = None
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::|sylph::", "", name)
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            cur = {"file": os.path.basename(f), "kernel": name}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r"remark:\s+VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
print(f'{"file":16s} {"kernel":44s} {"VGPRs":>5s} {"LDS B":>7s} {"scratch":>7s} {"waves/SIMD":>10s}')
for r in sorted(rows, key=lambda r: -r.get("lds", 0)):
    print(f'{r["file"]:16s} {r["kernel"][:44]:44s} {r.get("vgpr", 0):5d} {r.get("lds", 0):7d} {r.get("scratch", 0):7d} {r.get("occ", 0):10d}')
