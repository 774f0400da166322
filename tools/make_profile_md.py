#!/usr/bin/env python3
"""Assemble profiles/r02_* from the outputs of tools/r02_profile.sh (gpurun_out/r02_final/): bench lines, the rocprofv3
kernel-trace timeline, the PMC summary of reads_kernel / probe_kernel, the instruction-rate microbenchmark, the feed numbers."""
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r02_final")
dst = os.path.join(ROOT, "profiles")


def load(name):
    with open(os.path.join(src, name)) as f:
        return json.loads(f.read().strip().split("\n")[-1])


# tools/run_r02_profile.sh refuses to start with a dirty tree and records the commit the snapshot was taken from
hl = open(os.path.join(src, "head_local.txt")).read().strip().split("\n")
head, subject = hl[0], hl[1] if len(hl) > 1 else ""
b = {}
for wl in ("c3", "c3r", "c2", "c5", "c4"):
    try:
        b[wl] = load(f"bench_{wl}.json")
        shutil.copy(os.path.join(src, f"bench_{wl}.json"), os.path.join(dst, f"r02_bench_{wl}.json"))
    except Exception as e:
        print("missing", wl, e)
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
timeline = open(os.path.join(src, "step_timeline.md")).read()
c3 = b["c3"]
rf, rp = c3["roofline"], c3["roofline_profile"]
fetch = pmc["pmc_FETCH_SIZE"]
write = pmc["pmc_WRITE_SIZE"]
sq = pmc["pmc_SQ"]
sq2 = pmc.get("pmc_SQ2", {})
sqr = pmc.get("pmc_SQ_c3r", {})
# gfx950 rocprofv3: FETCH_SIZE is reported in KB of 64 B requests while streaming loads are 128 B requests -> x2 for wide coalesced
# streams (MI355X_MICROARCH.md, HBM section); random 64 B line reads (the probe) are counted as they are.
reads_fetch = fetch["reads.FETCH_SIZE"] * 1024 * 2
reads_write = write["reads.WRITE_SIZE"] * 1024
probe_fetch_raw = fetch["probe.FETCH_SIZE"] * 1024
probe_write = write["probe.WRITE_SIZE"] * 1024
kmers = rf["valu_ceiling"]["kmers_per_launch"]
cyc = sq["reads.GRBM_GUI_ACTIVE"] / 8
valu_per_kmer = sq["reads.SQ_INSTS_VALU"] / (kmers / 64)
valu_busy = sq["reads.SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
pcyc = sq["probe.GRBM_GUI_ACTIVE"] / 8
kp = open(os.path.join(src, "kernels_pipelined.md")).read()
seq = c3.get("one_step_at_a_time", {})
out = [
    "# r02 — rocprofv3 of bench.py C3 (1 Gbp of 2x150 bp reads vs 113,104-genome DB), one MI355X", "",
    f"Code state: `{head}` ({subject}) — the GPU box ran a snapshot of exactly this commit (clean working tree, checked by",
    "`tools/run_r02_profile.sh`); recipe `tools/r02_profile.sh`, assembled by `tools/make_profile_md.py`.", "",
    f"Un-profiled default run of the same build (profiles/r02_bench_c3.json, {c3['steps']} steps, {c3['config']['steps_in_flight']} steps in flight on {c3['config']['sketch_workers_per_gpu']} sketch streams",
    f"+ the profile stream): **{c3['ms_per_step']} ms/step = {c3['value']} Gbp/s**; `{rf['kernel']}` {rf['avg_launch_ms']} ms/launch and `probe_kernel` {rp['avg_launch_ms']} ms/launch",
    "by HIP events inside the library (durations include the time shared with the other streams' kernels).",
    f"One step at a time (same run, `one_step_at_a_time`): **{seq.get('ms_per_step')} ms/step = {seq.get('value')} Gbp/s**, `reads_kernel` {seq.get('kernel_ms', {}).get('seeds', [None])[0]} ms,",
    f"`probe_kernel` {seq.get('kernel_ms', {}).get('probe', [None])[0]} ms alone on the GPU: {100 * seq.get('roofline_frac', 0):.1f} % / {100 * seq.get('roofline_profile_frac', 0):.1f} % of the 8 TB/s peak.", "",
    "## (a) the default, pipelined command under the tracer", "",
    "    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_final/stats_p -o c3 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d --no-verify --no-sequential-leg", "",
    "Per kernel over the 10 timed steps (+ the untimed final step); dispatches of different streams overlap, so the durations",
    "are what the bench line's HIP events see:", "",
    kp, "",
    "## (b) one step at a time under the tracer", "",
    "    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_final/stats -o c3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2d --no-verify --pipeline-depth 1", "",
    "First table: the complete dispatch sequence of ONE step (sketch + profile of one sample, everything on one stream) with the idle gap before each",
    "dispatch (`__amd_rocclr_copyBuffer` = the runtime's copy kernels, `fillBufferAligned` = hipMemsetAsync).  Second table: totals over the last 5 steps.", "",
    timeline, "",
    "## PMC passes (each counter group in its own run, counters only — no trace domains)", "",
    "    rocprofv3 --pmc FETCH_SIZE  --kernel-include-regex 'reads_kernel|probe_kernel' ... -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-verify --no-kernel-timers --pipeline-depth 1",
    "    rocprofv3 --pmc WRITE_SIZE  (same)      rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE (same)", "",
    "Means over the last 3 dispatches of each kernel (the timed steps).", "",
    "### `reads_kernel<31,1,0>` (one launch = the whole 1 Gbp batch)", "",
    f"* FETCH_SIZE {fetch['reads.FETCH_SIZE']:.4g} KB raw → read bytes = x 1024 x 2 (gfx950: wide coalesced streaming reads are 128 B requests counted as 64 B,",
    f"  MI355X_MICROARCH.md HBM section) = **{reads_fetch:.4e} B**; WRITE_SIZE {write['reads.WRITE_SIZE']:.4g} KB → **{reads_write:.3e} B** (finished 32 B occurrence records into per-block slots).",
    f"* HBM traffic per launch = **{reads_fetch + reads_write:.4e} B** vs {rf['algorithmic_bytes_per_launch']:.4e} B algorithmic = {(reads_fetch + reads_write) / rf['algorithmic_bytes_per_launch']:.2f}x: no wasted re-reads",
    "  (the surplus: the 2 x 400-base halo per block, 32 B records where 8 B of seed would do).",
    f"* achieved = algorithmic bytes / {rf['avg_launch_ms']} ms (pipelined steps) = **{rf['achieved']} GB/s = {100 * rf['frac']:.1f} % of 8 TB/s**;",
    f"  alone on the GPU {seq.get('kernel_ms', {}).get('seeds', [None])[0]} ms = {100 * seq.get('roofline_frac', 0):.1f} %.",
    f"* SQ_INSTS_VALU {sq['reads.SQ_INSTS_VALU']:.4g} per launch / ({kmers:.3e} hashed k-mers / 64 lanes) = **{valu_per_kmer:.1f} VALU wave-instructions per hashed k-mer** (r01: 44.2, first r02 profile: 41.9);",
    f"  GRBM_GUI_ACTIVE / 8 XCDs = {cyc:.4g} cycles; VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles) = **{100 * valu_busy:.0f} %**.",
    "", "| counter | reads_kernel | probe_kernel |", "|---|---|---|"]
names = sorted({k.split(".", 1)[1] for k in list(sq) + list(sq2)})
allsq = dict(sq)
allsq.update(sq2)
for n in names:
    out.append(f"| {n} | {allsq.get('reads.' + n, float('nan')):.4g} | {allsq.get('probe.' + n, float('nan')):.4g} |")
out += ["",
        "### `probe_kernel` (one launch = one 1.9 M-entry sample table, four distinct samples rotated: nothing is cache-warm)", "",
        f"* FETCH_SIZE {fetch['probe.FETCH_SIZE']:.4g} KB → **{probe_fetch_raw:.4e} B** per launch = **{probe_fetch_raw / rp['probes_per_launch']:.0f} B per probe** as counted",
        f"  (64 B requests: the table in (12 B per probe, streamed) + one 64 B index line + overflow runs; r01: 180 B per probe with a cache-warm sample);",
        f"  WRITE_SIZE {write['probe.WRITE_SIZE']:.4g} KB → {probe_write:.3e} B ({rp['hits_per_launch']} hits x 8 B staged through LDS).",
        f"* algorithmic bytes = probes x (12 + 64) + 8 x hits = {rp['algorithmic_bytes_per_launch']:.4e} B; alone on the GPU {seq.get('kernel_ms', {}).get('probe', [None])[0]} ms = "
        f"**{100 * seq.get('roofline_profile_frac', 0):.1f} % of 8 TB/s** (latency-bound random line reads; {pcyc:.3g} cycles per launch);",
        f"  in the pipelined steps, beside two seeding kernels, {rp['avg_launch_ms']} ms = {rp['achieved']} GB/s = {100 * rp['frac']:.1f} %.",
        (f"* a batch of 8 samples per launch (profiles/r02_bench_c4.json): alone {b['c4'].get('one_step_at_a_time', {}).get('kernel_ms', {}).get('probe', [None])[0]} ms = "
         f"{100 * b['c4'].get('one_step_at_a_time', {}).get('roofline_profile_frac', 0):.1f} % of peak; pipelined {b['c4']['roofline_profile']['avg_launch_ms']} ms = {100 * b['c4']['roofline_profile']['frac']:.1f} %.") if "c4" in b else ""]
if sqr:
    k3 = b["c3r"]["roofline"]["valu_ceiling"]["kmers_per_launch"]
    c3c = sqr["reads.GRBM_GUI_ACTIVE"] / 8
    out += ["", "### ragged input (c3r: 2 x 35-151 bp uniform, 0.1 % N)", "",
            f"SQ_INSTS_VALU {sqr['reads.SQ_INSTS_VALU']:.4g} / ({k3:.3e} hashed k-mers / 64) = **{sqr['reads.SQ_INSTS_VALU'] / (k3 / 64):.1f} VALU wave-instructions per hashed k-mer** —",
            "wave-instructions are issued for the longest read of each wavefront while the shorter lanes idle; VALU busy "
            f"{100 * sqr['reads.SQ_ACTIVE_INST_VALU'] * 4 / (1024 * c3c):.0f} %; {b['c3r'].get('one_step_at_a_time', {}).get('kernel_ms', {}).get('seeds', [None])[0]} ms per 0.62 Gbp launch alone "
            f"({b['c3r']['roofline']['avg_launch_ms']} ms in the pipelined steps)."]
open(os.path.join(dst, "r02_kernel_stats.md"), "w").write("\n".join(out) + "\n")
json.dump({"hbm_bytes_per_launch": int(reads_fetch + reads_write), "kernel": "reads_kernel<31,1,0>", "valu_per_kmer": round(valu_per_kmer, 1),
           "valu_per_kmer_position_kernel": 38, "valu_busy": round(valu_busy, 3), "probe_fetch_bytes_per_probe": round(probe_fetch_raw / rp["probes_per_launch"], 1),
           "probe_hbm_bytes_per_probe": round((probe_fetch_raw + probe_write) / rp["probes_per_launch"], 1),
           "head": head, "source": "profiles/r02_kernel_stats.md PMC section (FETCH_SIZE x2 gfx950 correction for the streaming reads + WRITE_SIZE, separate --pmc passes)"},
          open(os.path.join(dst, "seeds_traffic.json"), "w"))
for name in ("valu_rates.txt", "feed.txt", "atomic_rates.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "r02_" + name))
print("ok: traffic", reads_fetch + reads_write, "valu/kmer", valu_per_kmer, "busy", valu_busy, "probe B/probe", probe_fetch_raw / rp["probes_per_launch"])
