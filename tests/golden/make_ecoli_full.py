#!/usr/bin/env python3
"""Packs the reference's three E. coli assemblies (test_files/*.fasta.gz, BASELINE configs[0]) into one small fixture so that the
FULL genomes can be sketched on the GPU box, where /root/reference does not exist:

    python tests/golden/make_ecoli_full.py        (build container only)

ecoli_full_bases.npz: per genome g0..g2 — `packed` (2 bits per base, 4 bases per byte, first base in the top bits: A=0 C=1 G=2
T=3), `n` (bases), `off` (contig offsets, needletail record boundaries), `exc_pos` / `exc_val` (every byte that is not one of
the upper-case letters ACGT, so that the test rebuilds the exact sequence bytes the reference's parser yields), `md5` of the
concatenated sequence bytes.  The expected answers are NOT in this file: they are SURVEY.md Appendix A.2's (survey_kat.json), an
answer key that was not produced by oracle/."""
import gzip
import hashlib
import os

import numpy as np

T = "/root/reference/test_files/"
GENOMES = ["e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz", "e.coli-o157.fasta.gz"]
HERE = os.path.dirname(os.path.abspath(__file__))


def read_fasta(path):
    seqs, cur = [], None
    with gzip.open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                cur = []
                seqs.append(cur)
            elif cur is not None:
                cur.append(line)
    return [b"".join(s) for s in seqs]


def main():
    out = {}
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    for g, name in enumerate(GENOMES):
        contigs = read_fasta(T + name)
        flat = np.frombuffer(b"".join(contigs), dtype=np.uint8)
        off = np.zeros(len(contigs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(c) for c in contigs])
        c = code[flat]
        exc = np.nonzero(c == 255)[0]
        c = np.where(c == 255, 0, c).astype(np.uint8)
        pad = (-len(c)) % 4
        c4 = np.concatenate([c, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        packed = (c4[:, 0] << 6) | (c4[:, 1] << 4) | (c4[:, 2] << 2) | c4[:, 3]
        out[f"g{g}_packed"] = packed.astype(np.uint8)
        out[f"g{g}_n"] = np.array([len(flat)], dtype=np.uint64)
        out[f"g{g}_off"] = off
        out[f"g{g}_exc_pos"] = exc.astype(np.uint64)
        out[f"g{g}_exc_val"] = flat[exc]
        out[f"g{g}_md5"] = np.frombuffer(hashlib.md5(flat.tobytes()).hexdigest().encode(), dtype=np.uint8)
        print(name, len(contigs), "contigs", len(flat), "bases", len(exc), "non-ACGT bytes", hashlib.md5(flat.tobytes()).hexdigest())
    np.savez_compressed(os.path.join(HERE, "ecoli_full_bases.npz"), **out)


if __name__ == "__main__":
    main()
