#!/usr/bin/env python3
"""Vectors produced by the REAL reference binary (oracle/_ref/sylph, built by oracle/ref_build.sh) -> tests/golden/ref_binary_vectors.npz.

Needs a Rust toolchain, which neither this image nor the GPU boxes of rounds 1-4 have: the file this script writes does not exist
yet, and tests/test_ref_binary.py skips.  The day it can run, the oracle stops being "unpinned": the test compares, bit for bit,
  * genome sketches  — `sylph sketch -g <the three E. coli assemblies of test_files/> -c 200 -k 31` -> genome_kmers +
    pseudotax_tracked_nonused_kmers of every genome (sketch.rs:550-622) vs oracle.sketch_genome on the same files;
  * read sketches    — `sylph sketch -1 k12_R1.fq -2 k12_R2.fq --fpr 0` (exact dedup, sketch.rs:690-731) and
    `sylph sketch -r k12_R1.fq` -> the (k-mer, count) multiset of the .sylsp vs oracle.sketch_reads;
  * the profile TSV  — `sylph profile db.syldb *.sylsp` and `sylph query`: the ANI / coverage / abundance columns at 1e-6
    vs oracle.contain + oracle.stats (and, on a GPU box, vs `sylph-hip`).
The bincode reader below is independent of sylph_amd/host/formats.cpp (struct.unpack only)."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("SYLPH_REFERENCE", "/root/reference")
BIN = os.path.join(ROOT, "oracle", "_ref", "sylph")
OUT = os.path.join(ROOT, "tests", "golden", "ref_binary_vectors.npz")


class Reader:
    def __init__(self, b):
        self.b, self.o = b, 0

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def vec_u64(self):
        n = self.u64()
        a = np.frombuffer(self.b, dtype="<u8", count=n, offset=self.o).copy()
        self.o += 8 * n
        return a

    def string(self):
        n = self.u64()
        s = self.b[self.o:self.o + n].decode()
        self.o += n
        return s

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v


def read_syldb(path):
    """bincode 1.3 default options of Vec<GenomeSketch> (types.rs:163-173)"""
    r = Reader(open(path, "rb").read())
    out = []
    for _ in range(r.u64()):
        g = {"genome_kmers": r.vec_u64()}
        g["tracked"] = r.vec_u64() if r.u8() else None
        g["file_name"], g["first_contig_name"] = r.string(), r.string()
        g["c"], g["k"], g["gn_size"], g["min_spacing"] = r.u64(), r.u64(), r.u64(), r.u64()
        out.append(g)
    assert r.o == len(r.b), "trailing bytes in .syldb"
    return out


def read_sylsp(path):
    """SequencesSketch (types.rs:145-155): seq of (u64, u32), c, k, file_name, Option<String>, bool, f64"""
    r = Reader(open(path, "rb").read())
    n = r.u64()
    rec = np.frombuffer(r.b, dtype=np.dtype([("k", "<u8"), ("c", "<u4")]), count=n, offset=r.o)
    r.o += 12 * n
    order = np.argsort(rec["k"], kind="stable")
    s = {"kmers": rec["k"][order].copy(), "counts": rec["c"][order].copy(), "c": r.u64(), "k": r.u64(), "file_name": r.string()}
    s["sample_name"] = r.string() if r.u8() else None
    s["paired"] = bool(r.u8())
    s["mean_read_length"] = struct.unpack_from("<d", r.b, r.o)[0]
    r.o += 8
    assert r.o == len(r.b), "trailing bytes in .sylsp"
    return s


def run(args, cwd):
    p = subprocess.run([BIN] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"sylph {' '.join(args)} failed ({p.returncode}): {p.stderr[-2000:]}")
    return p.stdout


def copy_crate_sources():
    """The source of the cuckoo filter crate cargo fetched for the build (~/.cargo/registry/src/*/scalable_cuckoo_filter-0.2.4) ->
    oracle/_ref/crate_src/ (git-ignored like the rest of oracle/_ref: reference material for the checker, never product source).
    tests/test_ref_binary.py::test_filter_model_against_crate_source reads it: fingerprint width, bucket count, hash derivation and —
    the one assumption of the model that changes results for samples above ~1.3 Gbp — WHEN a filter is declared full."""
    import glob
    import shutil
    home = os.environ.get("CARGO_HOME", os.path.join(os.path.expanduser("~"), ".cargo"))
    hits = sorted(glob.glob(os.path.join(home, "registry", "src", "*", "scalable_cuckoo_filter-0.2.4")))
    if not hits:
        print("crate sources not found under " + home + " (vendored build?): the model stays unchecked against them", file=sys.stderr)
        return
    dst = os.path.join(ROOT, "oracle", "_ref", "crate_src", "scalable_cuckoo_filter-0.2.4")
    shutil.rmtree(dst, ignore_errors=True)
    shutil.copytree(os.path.join(hits[-1], "src"), dst)
    print(f"copied {hits[-1]}/src -> {dst}")


def main():
    if not os.path.exists(BIN):
        print(f"{BIN} does not exist: run oracle/ref_build.sh on a box with cargo first", file=sys.stderr)
        return 3
    tf = os.path.join(REF, "test_files")
    genomes = [os.path.join(tf, f) for f in ("e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz", "e.coli-o157.fasta.gz")]
    with tempfile.TemporaryDirectory() as d:
        run(["sketch", "-g"] + genomes + ["-o", os.path.join(d, "db"), "-c", "200", "-k", "31", "-t", "2"], d)
        db = read_syldb(os.path.join(d, "db.syldb"))
        run(["sketch", "-1", os.path.join(tf, "k12_R1.fq"), "-2", os.path.join(tf, "k12_R2.fq"), "--fpr", "0", "-d", os.path.join(d, "pe"), "-c", "200", "-k", "31"], d)
        run(["sketch", "-r", os.path.join(tf, "k12_R1.fq"), "-d", os.path.join(d, "se"), "-c", "200", "-k", "31"], d)
        pe = read_sylsp(os.path.join(d, "pe", "k12_R1.fq.paired.sylsp"))
        se = read_sylsp(os.path.join(d, "se", "k12_R1.fq.sylsp"))
        # round 5: the reference's DEFAULT for pairs — dup_removal_lsh_full behind scalable_cuckoo_filter 0.2.4 (sketch.rs:733-769, :796-804) —
        # at its default --fpr and at a leaky one (false positives that show).  The oracle's model of that crate has its own hash bits
        # (the crate is not in the reference tree): these vectors say how far the model is from the real filter, they pin nothing yet.
        pe_def = {}
        for tag, extra in (("fpr_default", []), ("fpr_0.02", ["--fpr", "0.02"])):
            run(["sketch", "-1", os.path.join(tf, "k12_R1.fq"), "-2", os.path.join(tf, "k12_R2.fq"), "-d", os.path.join(d, tag), "-c", "200", "-k", "31"] + extra, d)
            pe_def[tag] = read_sylsp(os.path.join(d, tag, "k12_R1.fq.paired.sylsp"))
        prof = run(["profile", os.path.join(d, "db.syldb"), os.path.join(d, "pe", "k12_R1.fq.paired.sylsp"), os.path.join(d, "se", "k12_R1.fq.sylsp"), "-t", "2"], d)
        query = run(["query", os.path.join(d, "db.syldb"), os.path.join(d, "pe", "k12_R1.fq.paired.sylsp"), "-t", "2"], d)
    arrays = {"version": np.array(subprocess.run([BIN, "--version"], stdout=subprocess.PIPE, text=True).stdout.strip()),
              "profile_tsv": np.array(prof), "query_tsv": np.array(query)}
    for i, g in enumerate(db):
        arrays[f"g{i}_file"] = np.array(os.path.basename(g["file_name"]))
        arrays[f"g{i}_kmers"] = g["genome_kmers"]
        arrays[f"g{i}_tracked"] = g["tracked"] if g["tracked"] is not None else np.zeros(0, np.uint64)
        arrays[f"g{i}_gn_size"] = np.array(g["gn_size"], dtype=np.uint64)
    for name, s in (("pe", pe), ("se", se)):
        arrays[f"{name}_kmers"], arrays[f"{name}_counts"] = s["kmers"], s["counts"]
        arrays[f"{name}_mean_read_length"] = np.array(s["mean_read_length"])
    for tag, s in pe_def.items():
        arrays[f"pe_{tag}_kmers"], arrays[f"pe_{tag}_counts"] = s["kmers"], s["counts"]
    np.savez_compressed(OUT, **arrays)
    copy_crate_sources()
    print(f"wrote {OUT}: {len(db)} genome sketches, {len(pe['kmers'])} / {len(se['kmers'])} read-sketch entries")
    return 0


if __name__ == "__main__":
    sys.exit(main())
