"""GPU tests of sylph_fastq_* (csrc/fastq.hip): the records of plain four-line FASTQ text found on the device must be the records the
host readers find (oracle.read_fastx follows needletail: sketch.rs:775-815, :897-921), the sketches of a text pushed through
sylph_sketch_push_fastq must be the sketches of those records pushed as a batch, and everything that is not exactly four-line FASTQ must
come back as SYLPH_ERR_FORMAT (the caller's cue to use its host reader) — never as a guess."""
import numpy as np
import pytest

import sylph_amd as S
from oracle import oracle as O
from sylph_amd.binding import ERR_FORMAT, MEM_DEVICE, MEM_HOST, MEM_HOST_PINNED

from .helpers import concat, random_seq
from .test_gpu_parity import assert_same_sketch, make_reads

pytestmark = pytest.mark.gpu


def fastq_text(recs, rng, eol=b"\n", last_eol=True, trailing=b"", lower=0.0):
    out = []
    for i, r in enumerate(recs):
        q = bytes(rng.integers(33, 127, size=len(r), dtype=np.uint8))           # any printable: '@' and '+' at a quality line's start too
        if i % 7 == 0 and len(r):
            q = b"@" + q[1:]
        if i % 11 == 0 and len(r):
            q = b"+" + q[1:]
        seq = bytes(r)
        if lower and rng.random() < lower:
            seq = seq.lower()
        out.append(b"@read%d some description" % i + eol + seq + eol + (b"+" if i % 3 else b"+read%d" % i) + eol + q + eol)
    t = b"".join(out)
    if not last_eol and t.endswith(eol):
        t = t[: -len(eol)]
    return t + trailing


@pytest.mark.parametrize("eol,last_eol,trailing", [(b"\n", True, b""), (b"\n", False, b""), (b"\r\n", True, b""), (b"\r\n", False, b""),
                                                    (b"\n", True, b"\n\n\r\n\n"), (b"\r\n", True, b"\r\n\r\n")])
def test_records_of_fastq_text_found_on_the_device(ctx, eol, last_eol, trailing):
    rng = np.random.default_rng(31)
    genome = random_seq(rng, 50000)
    recs = make_reads(rng, genome, 3000, 150, dup_frac=0.1, ragged=True)
    recs[5] = genome[:0].copy()                                                 # an empty sequence (and quality) line
    recs[17] = genome[100:101].copy()
    recs[40] = genome[1000:1700].copy()                                         # a long record among the short ones
    for r in recs[::19]:
        if len(r) > 3:
            r[1] = ord("N")
    text = fastq_text(recs, rng, eol=eol, last_eol=last_eol, trailing=trailing, lower=0.2)
    f = S.FastqText(ctx, text)
    assert f.n_records == len(recs) and f.n_bases == sum(len(r) for r in recs)
    assert np.array_equal(f.lengths(), np.array([len(r) for r in recs], dtype=np.uint32))
    assert np.array_equal(f.lengths(10, 5), np.array([len(r) for r in recs[10:15]], dtype=np.uint32))
    b, off = concat(recs)
    for c in (5, 50):
        e = O.sketch_reads(b, off, c=c, paired=False)
        sk = S.ReadSketcher(ctx, c=c, paired=False)
        sk.push_fastq(f)
        assert_same_sketch(sk.finish(), e, ("whole", c))
        sk.close()
        sk = S.ReadSketcher(ctx, c=c, paired=False)                              # the same in three pushes of whole records
        for lo, hi in ((0, 1000), (1000, 1001), (1001, len(recs))):
            sk.push_fastq(f, first=lo, n_items=hi - lo)
        assert_same_sketch(sk.finish(), e, ("three pushes", c))
        sk.close()
    f.close()


def test_fastq_pairs_from_host_pinned_and_device_text(ctx):
    import torch
    rng = np.random.default_rng(32)
    genome = random_seq(rng, 80000)
    inter = make_reads(rng, genome, 12000, 150, paired=True, dup_frac=0.15, ragged=True)
    r1, r2 = inter[0::2], inter[1::2]
    r2 = r2 + [genome[:77].copy()] * 5                                          # mate file 2 is longer: the extra records are not pushed
    t1, t2 = fastq_text(r1, rng), fastq_text(r2, rng, eol=b"\r\n")
    b, off = concat(inter)
    pin1, pin2 = S.PinnedBuffer(len(t1) + 64), S.PinnedBuffer(len(t2) + 64)
    pin1.array[:len(t1)] = np.frombuffer(t1, dtype=np.uint8)
    pin2.array[:len(t2)] = np.frombuffer(t2, dtype=np.uint8)
    # device text at an odd address inside a larger buffer (the kernels read 16-byte words of the aligned stream around it)
    d1 = torch.zeros(len(t1) + 128, dtype=torch.uint8, device="cuda")
    d2 = torch.zeros(len(t2) + 128, dtype=torch.uint8, device="cuda")
    d1[37:37 + len(t1)] = torch.from_numpy(np.frombuffer(t1, dtype=np.uint8).copy()).cuda()
    d2[5:5 + len(t2)] = torch.from_numpy(np.frombuffer(t2, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    texts = {"host": (lambda: S.FastqText(ctx, t1), lambda: S.FastqText(ctx, t2)),
             "pinned": (lambda: S.FastqText(ctx, pin1.ptr, MEM_HOST_PINNED, len(t1)), lambda: S.FastqText(ctx, pin2.ptr, MEM_HOST_PINNED, len(t2))),
             "device": (lambda: S.FastqText(ctx, d1.data_ptr() + 37, MEM_DEVICE, len(t1)), lambda: S.FastqText(ctx, d2.data_ptr() + 5, MEM_DEVICE, len(t2)))}
    for c in (20, 200):
        e = O.sketch_reads(b, off, c=c, paired=True)
        ef = O.sketch_reads_cuckoo_model(b, off, c=c, fpr=1e-4)
        for name, (ma, mb) in texts.items():
            fa, fb = ma(), mb()
            assert fa.n_records == len(r1) and fb.n_records == len(r2)
            for fpr, want in ((None, e), (1e-4, ef)):
                for borrow in (0, 1):
                    sk = S.ReadSketcher(ctx, c=c, paired=True, **({} if fpr is None else {"dedup_fpr": fpr}))
                    sk.set_option("borrow_until_finish", borrow)                # the batch is the session's own: the verdict may wait for finish
                    sk.push_fastq(fa, fb)
                    assert_same_sketch(sk.finish(), want, (name, c, fpr, borrow))
                    sk.close()
            sk = S.ReadSketcher(ctx, c=c, paired=True)                          # two pushes; the second one after a deferred first
            sk.set_option("borrow_until_finish", 1)
            sk.push_fastq(fa, fb, 0, 9000)
            sk.push_fastq(fa, fb, 9000, len(r1) - 9000)
            assert_same_sketch(sk.finish(), e, (name, c, "two pushes"))
            sk.close()
            sk2 = S.ReadSketcher(ctx, c=c, paired=True)
            with pytest.raises(S.SylphHipError):
                sk2.push_fastq(fa)                                              # a paired session wants both texts
            with pytest.raises(S.SylphHipError):
                sk2.push_fastq(fa, fb, 0, len(r2))                              # more items than mate file 1 holds
            sk2.close()
            fa.close()
            fb.close()
    pin1.close()
    pin2.close()


def test_what_is_not_four_line_fastq_is_refused(ctx):
    rng = np.random.default_rng(33)
    genome = random_seq(rng, 5000)
    recs = [genome[i * 100:i * 100 + 90].copy() for i in range(40)]
    good = fastq_text(recs, rng)
    lines = good.split(b"\n")
    bad = {
        "empty": b"",
        "only blank space": b"\n\n\r\n",
        "fasta": b">a\nACGT\n>b\nGGCC\n",
        "three lines": b"@a\nACGT\n+\n",
        "a record without its '@'": good.replace(b"@read7 ", b"read7 ", 1),
        "a record without its '+'": b"\n".join(lines[:10] + [b"-"] + lines[11:]),
        "quality shorter than the sequence": b"\n".join(lines[:7] + [lines[7][:-1]] + lines[8:]),
        "a blank line between records": b"\n".join(lines[:8] + [b""] + lines[8:]),
        "multi-line sequence": b"@a\nACGT\nACGT\n+\nIIIIIIII\n",
        "five lines": good + b"@x\n",
    }
    for name, text in bad.items():
        with pytest.raises(S.SylphHipError) as ei:
            S.FastqText(ctx, text)
        assert ei.value.code == ERR_FORMAT, (name, str(ei.value))
    f = S.FastqText(ctx, good)                                                  # ... and the context is none the worse for it
    assert f.n_records == len(recs)
    f.close()


def test_fastq_text_larger_than_a_few_tiles_and_many_short_lines(ctx):
    """Line numbering across thousands of 4 KiB tiles: reads of 0-40 bases (hundreds of lines per tile) and reads of 20 kb (lines that span tiles)."""
    rng = np.random.default_rng(34)
    genome = random_seq(rng, 200000)
    tiny = [genome[s:s + int(rng.integers(0, 41))].copy() for s in rng.integers(0, 150000, size=60000)]
    long_ = [genome[s:s + int(rng.integers(15000, 25000))].copy() for s in rng.integers(0, 170000, size=150)]
    for recs, c in ((tiny, 3), (long_, 100), (tiny[:5000] + long_[:20] + tiny[5000:9000], 20)):
        text = fastq_text(recs, rng)
        f = S.FastqText(ctx, text)
        assert f.n_records == len(recs)
        assert np.array_equal(f.lengths(), np.array([len(r) for r in recs], dtype=np.uint32))
        b, off = concat(recs)
        e = O.sketch_reads(b, off, c=c, paired=False)
        sk = S.ReadSketcher(ctx, c=c, paired=False)
        sk.push_fastq(f)
        assert_same_sketch(sk.finish(), e, (len(recs), c))
        sk.close()
        f.close()
