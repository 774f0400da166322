"""gzip inflated on the device (csrc/inflate.hip) against zlib: byte-identical text or a clean SYLPH_ERR_FORMAT — never different
bytes.  The checker is Python's zlib (the library the reference's flate2 wraps a port of); nothing under oracle/ is involved.
Inputs: the reference's own three test_files/*.fasta.gz (copied as data into tests/golden/ref_test_files/), synthetic FASTQ at gzip
levels 1 / 6 / 9, stored-only streams, fixed-Huffman streams, concatenated members, BGZF, libdeflate's block splitting where the box
has the library, and damaged / truncated files."""
import ctypes
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import sylph_amd as S
from sylph_amd.binding import ERR_FORMAT, MEM_DEVICE, SylphHipError

from .helpers import bgzf_compress

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILES = os.path.join(HERE, "golden", "ref_test_files")


def fastq_text(rng, n_records, read_len=150, quals="binned"):
    """Four-line FASTQ: Illumina-like headers, random bases with a few N, qualities of a few distinct values in runs."""
    out = []
    qa = np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8) if quals == "binned" else np.arange(33, 74, dtype=np.uint8)
    for i in range(n_records):
        L = read_len if isinstance(read_len, int) else int(rng.integers(read_len[0], read_len[1]))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L, p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002])
        q = np.repeat(rng.choice(qa, size=L // 7 + 1), 7)[:L]
        out.append(b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGT\n" % (1101 + i // 9000, 1000 + (i * 37) % 30000, i))
        out.append(seq.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return b"".join(out)


def gz_level(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    return co.compress(data) + co.flush()


def check(ctx, gz, expect):
    t = S.Inflated(ctx, gz)
    try:
        assert t.n_bytes == len(expect)
        got = t.read().tobytes()
        if got != expect:
            a = np.frombuffer(got, dtype=np.uint8)
            b = np.frombuffer(expect, dtype=np.uint8)
            bad = np.flatnonzero(a != b)
            raise AssertionError(f"{len(bad)} of {len(b)} bytes differ, first at {bad[0]}: {got[bad[0] - 20:bad[0] + 20]!r} vs {expect[bad[0] - 20:bad[0] + 20]!r}")
        return dict(members=t.n_members, blocks=t.n_blocks, candidates=t.n_candidates, host_members=t.n_host_members, again=t.n_decoded_again)
    finally:
        t.close()


@pytest.mark.parametrize("name", ["e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz", "e.coli-o157.fasta.gz"])
def test_reference_fasta_gz(ctx, name):
    gz = open(os.path.join(REF_FILES, name), "rb").read()
    info = check(ctx, gz, gzip.decompress(gz))
    assert info["members"] == 1 and info["blocks"] >= 10 and info["host_members"] == 0
    assert info["candidates"] <= info["blocks"] + 2          # a false block start is a rare thing (one in ~4,000 blocks of a FASTQ file)


@pytest.mark.parametrize("level", [1, 6, 9])
def test_fastq_levels(ctx, level):
    text = fastq_text(np.random.default_rng(level), 60000)
    info = check(ctx, gz_level(text, level), text)
    assert info["blocks"] >= 20


def test_unbinned_qualities_and_ragged_reads(ctx):
    text = fastq_text(np.random.default_rng(11), 30000, read_len=(35, 251), quals="wide")
    check(ctx, gz_level(text, 6), text)


def test_stored_and_fixed_blocks(ctx):
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, size=300000, dtype=np.uint8).tobytes()
    check(ctx, gz_level(noise, 0), noise)                         # level 0: stored blocks only (one wave walks them all)
    check(ctx, gz_level(noise, 6), noise)                         # incompressible: zlib emits stored blocks between dynamic ones
    tiny = b"ACGT" * 5
    check(ctx, gz_level(tiny, 6), tiny)                           # one fixed-Huffman block
    check(ctx, gz_level(b"", 6), b"")
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)  # a long fixed-Huffman stream
    text = fastq_text(rng, 3000)
    check(ctx, co.compress(text) + co.flush(), text)
    # sync flushes: empty stored blocks between dynamic ones, a block boundary every few KB
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for a in range(0, len(text), 7001):
        parts.append(co.compress(text[a:a + 7001]) + co.flush(zlib.Z_SYNC_FLUSH))
    check(ctx, b"".join(parts) + co.flush(), text)


def test_long_matches_and_runs(ctx):
    # constant qualities (what bench.py's FASTQ holds): distance-1 runs of 150; a text of one byte; a period-3 run
    rng = np.random.default_rng(5)
    recs = []
    for i in range(40000):
        recs.append(b"@r%09d\n" % i + rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=150).tobytes() + b"\n+\n" + b"I" * 150 + b"\n")
    text = b"".join(recs)
    for level in (1, 6):
        check(ctx, gz_level(text, level), text)
    # deflates 1000:1 — far beyond the region a block gets at first (16 cells per compressed byte): such a block is decoded to its end
    # for its length alone, then once more into a region of that size
    ones = b"A" * 1000000 + b"CGT" * 300000
    info = check(ctx, gz_level(ones, 6), ones)
    assert info["again"] >= 1


def test_blocks_that_outgrow_their_region_are_decoded_again(ctx, monkeypatch):
    """A header-like bit pattern INSIDE a block (one in ~4,000 blocks of bench.py's FASTQ file has one) cuts the block's region short; so
    does a block that deflates better than 16:1.  Such a block runs dry — decoded to its end for its length alone — and is decoded
    again into a region of exactly that size.  Here: every block, by giving the regions one cell per compressed byte."""
    text = fastq_text(np.random.default_rng(23), 150000)
    gz = gz_level(text, 6)
    monkeypatch.setenv("SYLPH_HIP_INFLATE_REGION_RATIO", "1")
    # (a block that outgrows its region first moves to one of the 96 regions of the spill arena; the blocks that find it empty run dry)
    info = check(ctx, gz, text)
    assert info["blocks"] >= 130 and info["blocks"] - 97 <= info["again"] < info["blocks"]
    info = check(ctx, bgzf_compress(text), text)
    assert info["blocks"] - 97 <= info["again"] < info["blocks"]
    small = fastq_text(np.random.default_rng(24), 20000)
    assert check(ctx, gz_level(small, 6), small)["again"] == 0      # few blocks: the arena takes them all
    monkeypatch.delenv("SYLPH_HIP_INFLATE_REGION_RATIO")
    assert check(ctx, gz, text)["again"] == 0


def test_members_and_bgzf(ctx):
    rng = np.random.default_rng(7)
    a, b, c = fastq_text(rng, 9000), fastq_text(rng, 11), fastq_text(rng, 20000)
    info = check(ctx, gz_level(a, 6) + gz_level(b, 6) + gz_level(b"", 6) + gz_level(c, 1), a + b + c)
    assert info["members"] == 4 and info["host_members"] >= 1    # the empty member is one fixed block: zlib takes it
    text = a + c
    info = check(ctx, bgzf_compress(text), text)
    assert info["members"] == len(range(0, len(text), 65280)) + 1
    info = check(ctx, bgzf_compress(text, level=1), text)
    # a gzip header with every optional field
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\0\xff" + struct.pack("<H", 5) + b"extra" + b"name.fq\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(a) + co.flush()
    check(ctx, hdr + body + struct.pack("<II", zlib.crc32(a), len(a)), a)


def test_several_files_in_one_call(ctx):
    """sylph_inflate_files: the two mates of a pair (any files) taken as one stream of gzip members — one scan, one decode launch — and
    every file's text a piece of the one text, each usable by the FASTQ index where it lies."""
    rng = np.random.default_rng(29)
    a, b, c = fastq_text(rng, 20000), fastq_text(rng, 21000, read_len=(50, 151)), fastq_text(rng, 300)
    gzs = [gz_level(a, 6), gz_level(b, 1), bgzf_compress(c)]
    t = S.Inflated(ctx, gzs)
    assert t.n_bytes == len(a) + len(b) + len(c) and [n for _, n in t.files] == [len(a), len(b), len(c)]
    assert t.read().tobytes() == a + b + c
    assert t.files[1][0] == t.files[0][0] + len(a) and t.files[2][0] == t.files[1][0] + len(b)
    for (ptr, n), text in zip(t.files, (a, b, c)):
        fq, plain = S.FastqText(ctx, ptr, MEM_DEVICE, n), S.FastqText(ctx, text)
        assert fq.n_records == plain.n_records and fq.n_bases == plain.n_bases and np.array_equal(fq.lengths(), plain.lengths())
        fq.close(); plain.close()
    t.close()
    with pytest.raises(SylphHipError) as e:                         # all files or none
        S.Inflated(ctx, [gzs[0], gzs[1][:-9]])
    assert e.value.code == ERR_FORMAT


def test_libdeflate_streams(ctx):
    try:
        L = ctypes.CDLL("libdeflate.so.0")
    except OSError:
        pytest.skip("no libdeflate on this box")
    L.libdeflate_alloc_compressor.restype = ctypes.c_void_p
    L.libdeflate_gzip_compress.restype = ctypes.c_size_t
    L.libdeflate_gzip_compress.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    L.libdeflate_free_compressor.argtypes = [ctypes.c_void_p]
    text = fastq_text(np.random.default_rng(13), 50000)
    for level in (1, 6, 12):
        c = L.libdeflate_alloc_compressor(level)
        out = ctypes.create_string_buffer(len(text) + 1024)
        n = L.libdeflate_gzip_compress(c, text, len(text), out, len(out))
        L.libdeflate_free_compressor(c)
        assert n
        check(ctx, out.raw[:n], text)


def test_damage_is_declined(ctx):
    text = fastq_text(np.random.default_rng(17), 20000)
    gz = gz_level(text, 6)

    def declined(b):
        with pytest.raises(SylphHipError) as e:
            S.Inflated(ctx, b)
        assert e.value.code == ERR_FORMAT, e.value

    declined(text[:100000])                                       # not gzip at all
    declined(gz[: len(gz) // 2])                                  # truncated
    declined(gz[:-1])
    declined(gz + b"trailing garbage")
    bad = bytearray(gz)
    bad[len(gz) // 2] ^= 0x10                                     # one flipped bit in the middle: a broken chain or a CRC that differs
    declined(bytes(bad))
    bad = bytearray(gz)
    bad[-8] ^= 1                                                  # the trailer's CRC
    declined(bytes(bad))
    bad = bytearray(gz)
    bad[-1] ^= 1                                                  # ISIZE
    declined(bytes(bad))
    check(ctx, gz, text)                                          # and the context is fine afterwards


def test_inflated_text_feeds_the_fastq_index(ctx):
    text = fastq_text(np.random.default_rng(19), 25000, read_len=(60, 200))
    t = S.Inflated(ctx, gz_level(text, 6))
    plain = S.FastqText(ctx, text)
    fq = S.FastqText(ctx, t.dev_ptr, MEM_DEVICE, t.n_bytes)
    assert fq.n_records == plain.n_records == 25000 and fq.n_bases == plain.n_bases
    assert np.array_equal(fq.lengths(), plain.lengths())
    sk1, sk2 = S.ReadSketcher(ctx, c=20), S.ReadSketcher(ctx, c=20)
    sk1.push_fastq(fq)
    sk2.push_fastq(plain)
    r1, r2 = sk1.finish(), sk2.finish()
    assert np.array_equal(r1["kmers"], r2["kmers"]) and np.array_equal(r1["counts"], r2["counts"]) and len(r1["kmers"]) > 1000
    for o in (sk1, sk2, fq, plain, t):
        o.close()
