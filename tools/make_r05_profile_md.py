#!/usr/bin/env python3
"""Assemble profiles/r05_* from the outputs of tools/r05_profile.sh (gpurun_out/r05_final/): bench lines, the rocprofv3
kernel-trace timelines, the PMC summary of reads_kernel / probe_kernel / bucket_replay_kernel, the feed and stress numbers; and
profiles/seeds_traffic.json (the PMC traffic figures bench.py quotes — only for the kernel sources they were measured on)."""
import hashlib
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
TRAFFIC_ONLY = "--traffic-only" in sys.argv          # on the GPU box, between the PMC passes and the bench lines: only profiles/seeds_traffic.json
src = os.path.join(ROOT, "gpurun_out", "r05_final")
dst = os.path.join(ROOT, "profiles")
PREFIX = os.environ.get("SYLPH_PROFILE_ROUND", "r05")     # file-name prefix under profiles/ (round 6 runs the same recipe: SYLPH_PROFILE_ROUND=r06)


def load(name):
    with open(os.path.join(src, name)) as f:
        return json.loads([ln for ln in f.read().strip().split("\n") if ln.startswith("{")][-1])


def csrc_fingerprint():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sylph_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


hl = open(os.path.join(ROOT, "profiles", ".head_local")).read().strip().split("\n")
head, subject = hl[0], hl[1] if len(hl) > 1 else ""
b = {}
for wl in (("c3_pre",) if TRAFFIC_ONLY else ("c3", "c3r", "c2", "c5", "c4", "small_2ranks", "small_2ranks_genome", "small_2ranks_shard")):
    try:
        b[wl] = load(f"bench_{wl}.json")
        if not TRAFFIC_ONLY:
            shutil.copy(os.path.join(src, f"bench_{wl}.json"), os.path.join(dst, f"{PREFIX}_bench_{wl}.json"))
    except Exception as e:
        print("missing", wl, e)
if TRAFFIC_ONLY:
    b["c3"] = b["c3_pre"]
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
timeline = open(os.path.join(src, "step_timeline.md")).read() if not TRAFFIC_ONLY else ""
kp = open(os.path.join(src, "kernels_pipelined.md")).read() if not TRAFFIC_ONLY else ""
c3 = b["c3"]
rf, rp = c3["roofline"], c3["roofline_profile"]
seq = c3.get("one_step_at_a_time", {})
pip = c3.get("pipelined", {})
fetch, write, sq, sq2, sqr = pmc["pmc_FETCH_SIZE"], pmc["pmc_WRITE_SIZE"], pmc["pmc_SQ"], pmc.get("pmc_SQ2", {}), pmc.get("pmc_SQ_c3r", {})
# gfx950 rocprofv3: FETCH_SIZE is reported in KB of 64 B requests while streaming loads are 128 B requests -> x2 for wide coalesced
# streams (MI355X_MICROARCH.md, HBM section); random 64 B line reads (the probe, the replay's gather) are counted as they are.
reads_fetch = fetch["reads.FETCH_SIZE"] * 1024 * 2
reads_write = write["reads.WRITE_SIZE"] * 1024
probe_fetch_raw = fetch["probe.FETCH_SIZE"] * 1024
probe_write = write["probe.WRITE_SIZE"] * 1024
n_bases = c3["config"]["reads_per_sample_gbp"] * 1e9
kmers = n_bases - 2 * 3_333_334 * 30                      # hashed k-mers of 2 x 150 bp reads, k = 31 (avx2_compat: 120 per read)
cyc = sq["reads.GRBM_GUI_ACTIVE"] / 8
valu_per_kmer = sq["reads.SQ_INSTS_VALU"] / (kmers / 64)
valu_busy = sq["reads.SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
a_seeds = rf.get("alone_on_gpu", {})
a_probe = rp.get("alone_on_gpu", {})
out = [
    f"# {PREFIX} — rocprofv3 of bench.py C3 (1 Gbp of 2x150 bp reads vs 113,104-genome DB), one MI355X", "",
    f"Code state: `{head}` ({subject}) — the GPU box ran a snapshot of exactly this commit (clean working tree, checked by",
    "`tools/run_r05_profile.sh`); recipe `tools/r05_profile.sh`, assembled by `tools/make_r05_profile_md.py`.", "",
    f"Un-profiled default run of the same build (profiles/{PREFIX}_bench_c3.json): mode **{c3['mode']}**, {c3['steps']} steps of {c3['config']['samples_per_gpu_per_step']} samples, timed region "
    f"{c3['timed_region_s']} s: **{c3['ms_per_sample']} ms per sample = {c3['value']} Gbp/s** (per-sample completion interval p50 {c3['sample_interval_ms']['p50']} / p99 {c3['sample_interval_ms']['p99']} / max {c3['sample_interval_ms']['max']} ms;",
    f"step p50 {c3['step_ms']['p50']} / max {c3['step_ms']['max']} ms).  The other mode in the same run: pipelined {pip.get('ms_per_sample')} ms, one sample at a time {seq.get('ms_per_sample')} ms per sample",
    f"({seq.get('value')} Gbp/s; sketch {seq.get('sketch_ms')} ms + profile {seq.get('profile_ms')} ms wall clock).",
    f"`{rf['kernel']}`: {rf['avg_launch_ms']} ms per launch in the timed ({c3['mode']}) region, **{a_seeds.get('avg_launch_ms')} ms alone on the GPU = {100 * a_seeds.get('frac', 0):.1f} % of the 8 TB/s HBM peak**;",
    f"`probe_kernel`: {rp['avg_launch_ms']} ms ({rp['tables_per_launch']} tables per launch) in the timed region, {a_probe.get('avg_launch_ms')} ms alone = {100 * a_probe.get('frac', 0):.1f} %.", "",
    "## (a) the pipelined mode under the tracer", "",
    "    rocprofv3 --kernel-trace --stats --output-format csv ... -- python bench.py --steps 4 --warmup 1 --min-seconds 0.3 --mode pipelined --no-second-leg --no-cpu-baseline --no-h2d --no-verify", "",
    "Per kernel over the last samples; dispatches of different streams overlap, so the durations are what the bench line's HIP events see:", "",
    kp, "",
    "## (b) one sample at a time under the tracer", "",
    "    rocprofv3 --kernel-trace --stats --output-format csv ... -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg --no-cpu-baseline --no-h2d --no-verify", "",
    "First table: the complete dispatch sequence of ONE sample (sketch + profile, everything on one stream) with the idle gap before each",
    "dispatch (`__amd_rocclr_copyBuffer` = the runtime's copy kernels, `fillBufferAligned` = hipMemsetAsync).  Second table: totals over the last 5 samples.", "",
    timeline, "",
    "## PMC passes (each counter group in its own run, counters only — no trace domains)", "",
    "    rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'reads_kernel|probe_kernel|bucket_replay_kernel' ... -- python bench.py --steps 3 --warmup 1 --min-seconds 0.02 --mode sequential --no-second-leg ... --no-kernel-timers",
    "    rocprofv3 --pmc WRITE_SIZE (same)      rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE (same)", "",
    "Means over the last 6 dispatches of each kernel.", "",
    "### `reads_kernel<31,1,0>` (one launch = the whole 1 Gbp batch)", "",
    f"* FETCH_SIZE {fetch['reads.FETCH_SIZE']:.4g} KB raw -> read bytes = x 1024 x 2 (gfx950: wide coalesced streaming reads are 128 B requests counted as 64 B,",
    f"  MI355X_MICROARCH.md HBM section) = **{reads_fetch:.4e} B**; WRITE_SIZE {write['reads.WRITE_SIZE']:.4g} KB -> **{reads_write:.3e} B** (32 B occurrence records + 4 B bucket keys into per-block slots).",
    f"* HBM traffic per launch = **{reads_fetch + reads_write:.4e} B** vs {rf['algorithmic_bytes_per_launch']:.4e} B algorithmic = {(reads_fetch + reads_write) / rf['algorithmic_bytes_per_launch']:.2f}x: no wasted re-reads.",
    f"* SQ_INSTS_VALU {sq['reads.SQ_INSTS_VALU']:.4g} per launch / ({kmers:.3e} hashed k-mers / 64 lanes) = **{valu_per_kmer:.1f} VALU wave-instructions per hashed k-mer**;",
    f"  GRBM_GUI_ACTIVE / 8 XCDs = {cyc:.4g} cycles; VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles) = **{100 * valu_busy:.0f} %**.",
    "", "| counter | reads_kernel | probe_kernel | bucket_replay_kernel |", "|---|---|---|---|"]
allsq = dict(sq)
allsq.update(sq2)
allsq.update({k: v for k, v in fetch.items()})
allsq.update({k: v for k, v in write.items()})
for n in sorted({k.split(".", 1)[1] for k in allsq}):
    out.append(f"| {n} | {allsq.get('reads.' + n, float('nan')):.4g} | {allsq.get('probe.' + n, float('nan')):.4g} | {allsq.get('replay.' + n, float('nan')):.4g} |")
out += ["",
        "### `probe_kernel` (one launch = one 1.9 M-entry sample table, four distinct samples rotated: nothing is cache-warm)", "",
        f"* FETCH_SIZE {fetch['probe.FETCH_SIZE']:.4g} KB -> **{probe_fetch_raw:.4e} B** per launch = **{probe_fetch_raw / a_probe.get('probes_per_launch', 1):.0f} B per probe** as counted",
        f"  (64 B requests: the table in (12 B per probe, streamed) + one 64 B index line + overflow runs); WRITE_SIZE {write['probe.WRITE_SIZE']:.4g} KB -> {probe_write:.3e} B.",
        f"* algorithmic bytes = probes x (12 + 64) + 8 x hits = {a_probe.get('algorithmic_bytes_per_launch', 0):.4e} B; alone on the GPU {a_probe.get('avg_launch_ms')} ms = "
        f"**{100 * a_probe.get('frac', 0):.1f} % of 8 TB/s** (random 64 B line reads; two lines in flight per lane since round 3: profiles/r03_probe_ilp.txt)."]
if "replay.FETCH_SIZE" in fetch:
    out += ["", "### `bucket_replay_kernel<256,128>` (one launch = the sample's ~39,000 hash-range buckets)", "",
            f"* FETCH_SIZE {fetch['replay.FETCH_SIZE']:.4g} KB -> {fetch['replay.FETCH_SIZE'] * 1024:.3e} B (the 32 B occurrence records gathered straight from the seeding kernel's slots "
            f"through the partition's permutation, + the permutation itself); WRITE_SIZE {write['replay.WRITE_SIZE']:.4g} KB -> {write['replay.WRITE_SIZE'] * 1024:.3e} B ((k-mer, count) rows)."]
if sqr:
    c3r = b.get("c3r", {})
    k3 = c3r["config"]["reads_per_sample_gbp"] * 1e9 if c3r else 0
    out += ["", "### ragged input (c3r: 2 x 35-151 bp uniform, 0.1 % N)", "",
            f"SQ_INSTS_VALU {sqr['reads.SQ_INSTS_VALU']:.4g} per launch over {k3:.3e} bases (k-mers hashed: bases - 30 per record, AVX2 tail rule): wave-instructions are issued for the longest read of each",
            f"wavefront while the shorter lanes idle; VALU busy {100 * sqr['reads.SQ_ACTIVE_INST_VALU'] * 4 / (1024 * sqr['reads.GRBM_GUI_ACTIVE'] / 8):.0f} %; "
            f"{c3r.get('roofline', {}).get('alone_on_gpu', {}).get('avg_launch_ms')} ms per launch alone ({c3r.get('value')} Gbp/s for the whole step, mode {c3r.get('mode')})."]
asm_rows = []
for kname, label in (("count", "hits_count_kernel"), ("scatter", "hits_scatter_kernel"), ("rowsort", "rows_sort_kernel")):
    if f"{kname}.FETCH_SIZE" in fetch:
        asm_rows.append(f"| `{label}` | {fetch[kname + '.FETCH_SIZE'] * 1024:.3e} | {write.get(kname + '.WRITE_SIZE', 0) * 1024:.3e} | {sq.get(kname + '.SQ_INSTS_VALU', float('nan')):.3g} | {sq.get(kname + '.GRBM_GUI_ACTIVE', 0) / 8:.4g} |")
if asm_rows:
    out += ["", "### row assembly of the hits (csrc/hits.hip, round 4: replaces the library's radix sort of the hit list)", "",
            "| kernel | FETCH_SIZE bytes (raw 64 B requests) | WRITE_SIZE bytes | VALU wave-instr | cycles per XCD |", "|---|---|---|---|---|"] + asm_rows
c5f, c5w, c5s = pmc.get("pmc_c5_FETCH_SIZE", {}), pmc.get("pmc_c5_WRITE_SIZE", {}), pmc.get("pmc_c5_SQ", {})
slots_traffic = None
if c5f and c5w:
    c5 = b.get("c5", {})
    rf5 = c5.get("roofline", {})
    slots_traffic = c5f.get("slots.FETCH_SIZE", 0) * 1024 * 2 + c5w.get("slots.WRITE_SIZE", 0) * 1024
    per_launch_kmers = rf5.get("valu_ceiling", {}).get("kmers_per_launch") or (rf5.get("algorithmic_bytes_per_launch", 0) / 1.08)
    out += ["", "### `seeds_slots_kernel<31,1>` (C5: long reads, one launch = one push of <= 3e9 bases)", "",
            f"* FETCH_SIZE {c5f.get('slots.FETCH_SIZE', 0):.4g} KB x 1024 x 2 (streaming reads, gfx950 correction) + WRITE_SIZE {c5w.get('slots.WRITE_SIZE', 0):.4g} KB x 1024 = **{slots_traffic:.4e} B per launch** vs "
            f"{rf5.get('algorithmic_bytes_per_launch', 0):.4e} B algorithmic = {slots_traffic / max(1, rf5.get('algorithmic_bytes_per_launch', 1)):.2f}x.",
            f"* SQ_INSTS_VALU {c5s.get('slots.SQ_INSTS_VALU', 0):.4g} per launch = {c5s.get('slots.SQ_INSTS_VALU', 0) / max(1, per_launch_kmers / 64):.1f} wave-instructions per hashed position; "
            f"VALU busy {100 * c5s.get('slots.SQ_ACTIVE_INST_VALU', 0) * 4 / max(1, 1024 * c5s.get('slots.GRBM_GUI_ACTIVE', 1) / 8):.0f} %; "
            f"{rf5.get('alone_on_gpu', {}).get('avg_launch_ms')} ms per launch alone ({c5.get('value')} Gbp/s for the whole step)."]
# ---- the filter dedup's partitioned pass (round 5): six dispatches per sample, counters per (kernel, grid)
a10_traffic = None
try:
    a10 = json.load(open(os.path.join(src, "pmc_a10.json")))
    rows_a = []
    tot_f = tot_w = tot_valu = 0.0
    # the operations' partition is the pair of part_* dispatches with the LARGER grid (two words per slot); the smaller one is the replay's
    grids = {}
    for key in a10:
        name, grid = key.rsplit("@", 1)
        grids.setdefault(name, []).append(int(grid))
    for key, v in sorted(a10.items()):
        name, grid = key.rsplit("@", 1)
        mine = name.startswith("a10_") or (name.startswith("part_") and (len(grids[name]) == 1 or int(grid) == max(grids[name])))
        if name.startswith("part_fine") and any(k.startswith("a10_range_kernel") for k in a10):
            mine = False                                     # round 6: one partition level — the fine level is the replay's alone
        if name.startswith("part_hist") and any(k.startswith("a10_ops_tile_kernel") for k in a10):
            mine = False                                     # ... and the histogram comes from the operation words' kernel: part_hist is the replay's alone
        if not mine or name == "part_scan_kernel" and len(grids[name]) > 1 and int(grid) != max(grids[name]):
            continue
        f_b, w_b = v.get("FETCH_SIZE", 0) * 1024 * 2, v.get("WRITE_SIZE", 0) * 1024
        # one (kernel, grid) serves BOTH partitions of a sample (the operations' 8.0 M words and the replay's 4.0 M pairs: same number of
        # coarse ranges): the mean over its dispatches is the mean of the two, the operations' share of it 2 x 8 / (8 + 4)
        share = 4.0 / 3.0 if (name.startswith("part_") and len(grids[name]) == 1) else 1.0
        tot_f += f_b * share; tot_w += w_b * share; tot_valu += v.get("SQ_INSTS_VALU", 0) * share
        rows_a.append(f"| `{name}` | {grid} | {f_b * share:.3e} | {w_b * share:.3e} | {v.get('SQ_INSTS_VALU', 0) * share:.3g} | {v.get('SQ_INSTS_SALU', 0) * share:.3g} | {v.get('SQ_LDS_BANK_CONFLICT', 0) * share:.3g} |")
    a10_traffic = tot_f + tot_w
    ra = b["c3"].get("roofline_a10", {})
    out += ["", "### the filter dedup's partitioned pass (csrc/a10.hip: `--main-dedup-fpr 1e-4`, every sample behind sylph's default filter)", "",
            "FETCH_SIZE x 1024 x 2 (coalesced streaming reads: calibrated in round 5 on part_hist_kernel, which read the 8.0 M operation words = 64 MB and reported 32.0 MB), WRITE_SIZE x 1024:", "",
            "| kernel | grid | read bytes | written bytes | VALU wave-instr | SALU | LDS bank conflicts |", "|---|---|---|---|---|---|---|"] + rows_a
    out += ["", f"Per sample: **{a10_traffic:.4e} B of HBM traffic** for {ra.get('algorithmic_bytes_per_launch', 0):.4e} algorithmic bytes (the 32 B record of every occurrence) = "
                f"{a10_traffic / max(1, ra.get('algorithmic_bytes_per_launch', 1)):.1f}x — 8-byte operation words through one partition level (round 5: two) and an LDS bit table per range, all of it coalesced — in {ra.get('avg_launch_ms')} ms alone on the GPU "
                f"= {a10_traffic / max(1e-9, ra.get('avg_launch_ms', 1) * 1e-3) / 1e12:.2f} TB/s; round 4's walk: 16 M device-wide atomics + 8 M agent-scope loads on a 256 MB table, 0.92 ms."]
    if os.path.exists(os.path.join(src, "step_timeline_filter.md")):
        out += ["", "One sample with the filter on, dispatch by dispatch (rocprofv3 --kernel-trace of the same sequential run):", "", open(os.path.join(src, "step_timeline_filter.md")).read()]
except Exception as e:
    print("a10 section:", e)
fk = os.path.join(src, "kernel_stats_filter.csv")
if os.path.exists(fk):
    import csv
    rows = [r for r in csv.DictReader(open(fk)) if "a10_" in r["Name"] or "fillBuffer" in r["Name"] or "compact_occ" in r["Name"]]
    fl = b.get("c3", {}).get("default_pair_dedup", {})
    out += ["", "### the filter dedup (csrc/a10.hip; rocprofv3 --stats of a run whose last leg sketches with `dedup_fpr` 1e-4, averages over its pipelined and its one-at-a-time samples)", "",
            "| kernel | calls | average us |", "|---|---|---|"]
    out += [f"| `{r['Name'].replace('sylph::(anonymous namespace)::', '').split('(')[0][:60]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} |" for r in rows]
    if fl and "pipelined" in fl:
        out += ["", f"Bench line of the same tree: pipelined {fl['pipelined']['value']} Gbp/s ({fl['pipelined']['ms_per_sample']} ms per sample), one at a time "
                    f"{fl['one_step_at_a_time']['ms_per_sample']} ms; the whole table of a 1 Gbp sample equal to the oracle's walk of the filter: {fl.get('verify', {}).get('table_equal')}."]
if not TRAFFIC_ONLY:
    open(os.path.join(dst, PREFIX + "_kernel_stats.md"), "w").write("\n".join(out) + "\n")
json.dump({"hbm_bytes_per_launch": int(reads_fetch + reads_write), "kernel": "reads_kernel<31,1,0>", "valu_per_kmer": round(valu_per_kmer, 1),
           "valu_per_kmer_position_kernel": 38, "valu_busy": round(valu_busy, 3),
           "position_kernel_hbm_bytes_per_launch": int(slots_traffic) if slots_traffic else None,
           "probe_fetch_bytes_per_probe": round(probe_fetch_raw / a_probe.get("probes_per_launch", 1), 1),
           "probe_hbm_bytes_per_probe": round((probe_fetch_raw + probe_write) / a_probe.get("probes_per_launch", 1), 1),
           "a10_hbm_bytes_per_sample": int(a10_traffic) if a10_traffic else None,
           "head": head, "csrc_sha": csrc_fingerprint(),
           "source": "the round's pmc_summary.json + pmc_a10.json under profiles/ (tools/r05_profile.sh / r05_pmc_refresh.sh: FETCH_SIZE x2 gfx950 correction for the streaming reads + WRITE_SIZE, separate --pmc passes)"},
          open(os.path.join(dst, "seeds_traffic.json"), "w"))
if TRAFFIC_ONLY:
    shutil.copy(os.path.join(dst, "seeds_traffic.json"), os.path.join(src, "seeds_traffic.json"))
    print("traffic only: csrc", csrc_fingerprint())
    sys.exit(0)
for name in ("feed.txt", "db_load.txt", "stress_shared_kmers.txt", "stress_deep_coverage.txt", "stress_deep_coverage_filter_dedup.txt", "stress_deep_long_reads.txt", "kernel_stats.csv",
             "pytest_gpu.txt", "multi_pipeline_2replicas_one_gpu.json", "pmc_a10.json", "cli_first_sample_trace.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, PREFIX + "_" + name))
print("ok: traffic", reads_fetch + reads_write, "valu/kmer", valu_per_kmer, "busy", valu_busy, "probe B/probe", probe_fetch_raw / a_probe.get("probes_per_launch", 1), "csrc", csrc_fingerprint())
