"""CPU tests of the device inflate's bookkeeping (sylph_amd/csrc/inflate_plan.h, the very header csrc/inflate.hip includes): CRC-32 put
together from shifted pieces, gzip member headers, and the chain walk — driven by a CPU model of the device kernels' reports
(tests/inflate_plan_capi.cpp) and checked against zlib.  The GPU kernels themselves are tested in tests/test_gpu_inflate.py."""
import ctypes as C
import gzip
import os
import struct
import subprocess
import tempfile
import zlib

import numpy as np
import pytest

from .helpers import bgzf_compress

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def L():
    out = os.path.join(tempfile.gettempdir(), f"sylph_inflate_plan_{os.getuid()}.so")
    src = os.path.join(HERE, "inflate_plan_capi.cpp")
    hdr = os.path.join(HERE, "..", "sylph_amd", "csrc", "inflate_plan.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        tmp = out + f".{os.getpid()}"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", tmp, src, "-lz"])
        os.replace(tmp, out)
    lib = C.CDLL(out)
    lib.ip_member_body.restype = C.c_uint64
    lib.ip_member_body.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
    lib.ip_crc_members.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ip_crc_shift.restype = C.c_uint32
    lib.ip_crc_shift.argtypes = [C.c_uint32, C.c_uint64]
    lib.ip_model_inflate.restype = C.c_longlong
    lib.ip_model_inflate.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    return lib


def model(L, gz, cap, ratio=16, slack=1024):
    out = C.create_string_buffer(max(cap, 1))
    info = (C.c_uint64 * 5)()
    err = C.create_string_buffer(256)
    n = L.ip_model_inflate(gz, len(gz), out, cap, info, ratio, slack, err, 256)
    if n < 0:
        return None, err.value.decode(), None
    return out.raw[:n], "", dict(members=info[0], blocks=info[1], candidates=info[2], host_members=info[3], redone=info[4])


def gz_level(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    return co.compress(data) + co.flush()


def fastq_text(rng, n):
    out = []
    for i in range(n):
        L_ = int(rng.integers(50, 200))
        out.append(b"@read%d/1\n" % i + rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L_).tobytes() + b"\n+\n" +
                   np.repeat(rng.choice(np.frombuffer(b"F:,#", dtype=np.uint8), size=L_ // 5 + 1), 5)[:L_].tobytes() + b"\n")
    return b"".join(out)


def test_crc_from_shifted_pieces_equals_zlib(L):
    rng = np.random.default_rng(1)
    data = rng.integers(0, 256, size=300001, dtype=np.uint8).tobytes()
    for ends in ([len(data)], [1, 2, 1500, 1500, 70000, len(data)], [0, 0, len(data)], list(range(1000, len(data), 9973)) + [len(data)]):
        for piece in (1024, 4096, 7):
            if piece == 7 and len(ends) > 3:
                continue
            e = (C.c_uint64 * len(ends))(*ends)
            out = (C.c_uint32 * len(ends))()
            L.ip_crc_members(data, len(data), piece, e, len(ends), out)
            begin = 0
            for i, z in enumerate(ends):
                assert out[i] == zlib.crc32(data[begin:z]), (ends[:4], piece, i)
                begin = z
    # the register after n zero bytes
    for n in (0, 1, 5, 1 << 20, (1 << 33) + 12345):
        want = 0x12345678
        if n <= 1 << 20:
            tab = [0] * 256
            for i in range(256):
                c = i
                for _ in range(8):
                    c = (c >> 1) ^ 0xEDB88320 if c & 1 else c >> 1
                tab[i] = c
            for _ in range(n):
                want = tab[want & 0xFF] ^ (want >> 8)
            assert L.ip_crc_shift(0x12345678, n) == want
        else:   # composition
            assert L.ip_crc_shift(0x12345678, n) == L.ip_crc_shift(L.ip_crc_shift(0x12345678, 1 << 33), 12345)


def test_member_headers(L):
    plain = gzip.compress(b"hello")
    assert L.ip_member_body(plain, len(plain), 0) == 10
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\0\xff" + struct.pack("<H", 5) + b"extra" + b"name.fq\0" + b"a comment\0" + b"\x12\x34"
    whole = hdr + plain[10:]
    assert L.ip_member_body(whole, len(whole), 0) == len(hdr)
    assert L.ip_member_body(b"\x1f\x8b\x07" + plain[3:], len(plain), 0) == 0         # not deflate
    assert L.ip_member_body(plain[:12], 12, 0) == 0                                 # too short to hold a trailer
    assert L.ip_member_body(b"@read\nACGT\n+\nIIII\n" * 4, 64, 0) == 0
    b = bgzf_compress(b"ACGT" * 100)
    assert L.ip_member_body(b, len(b), 0) == 18


@pytest.mark.parametrize("level", [1, 6, 9])
def test_model_road_equals_zlib(L, level):
    text = fastq_text(np.random.default_rng(level), 12000)
    got, why, info = model(L, gz_level(text, level), len(text) + 16)
    assert why == "" and got == text
    assert info["members"] == 1 and info["blocks"] >= 4 and info["candidates"] <= info["blocks"] + 1


def test_model_road_members_bgzf_stored_fixed(L):
    rng = np.random.default_rng(4)
    a, b, c = fastq_text(rng, 3000), fastq_text(rng, 5), fastq_text(rng, 4000)
    got, why, info = model(L, gz_level(a, 6) + gz_level(b, 6) + gz_level(b"", 6) + gz_level(c, 1), len(a + b + c) + 16)
    assert why == "" and got == a + b + c and info["members"] == 4 and info["host_members"] >= 1
    text = a + c
    got, why, info = model(L, bgzf_compress(text), len(text) + 16)
    assert why == "" and got == text and info["members"] == len(range(0, len(text), 65280)) + 1
    noise = rng.integers(0, 256, size=100000, dtype=np.uint8).tobytes()
    for gz in (gz_level(noise, 0), gz_level(noise, 6), gz_level(b"ACGT" * 5, 6), gz_level(b"", 6), gz_level(a, 6, zlib.Z_FIXED)):
        want = gzip.decompress(gz)
        got, why, info = model(L, gz, len(want) + 16)
        assert why == "" and got == want
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = [co.compress(a[i:i + 7001]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(a), 7001)]
    got, why, info = model(L, b"".join(parts) + co.flush(), len(a) + 16)
    assert why == "" and got == a


def test_model_road_declines_damage(L):
    text = fastq_text(np.random.default_rng(6), 6000)
    gz = gz_level(text, 6)
    cap = len(text) + 16
    assert model(L, text[:5000], cap)[0] is None
    assert model(L, gz[: len(gz) // 2], cap)[0] is None
    assert model(L, gz[:-1], cap)[0] is None
    assert model(L, gz + b"garbage", cap)[0] is None
    for at, bit in ((len(gz) // 2, 0x10), (len(gz) - 8, 1), (len(gz) - 1, 1)):
        bad = bytearray(gz)
        bad[at] ^= bit
        assert model(L, bytes(bad), cap)[0] is None, at
    # a region too small for its block (here: a ratio no FASTQ block meets) is not damage: the block is decoded again with room for it
    got, why, info = model(L, gz, cap, ratio=1, slack=0)
    assert why == "" and got == text and info["redone"] == info["blocks"] >= 2
    got, why, info = model(L, gz, cap)
    assert got == text and info["redone"] == 0


def test_reference_fasta_gz_through_the_model(L, golden_dir):
    gz = open(os.path.join(golden_dir, "ref_test_files", "e.coli-K12.fasta.gz"), "rb").read()
    want = gzip.decompress(gz)
    # the model tests every bit position one bit at a time: cut the file's text to 600 kB and re-deflate it the way the file was (zlib level 6)
    cut = want[:600000]
    got, why, info = model(L, gz_level(cut, 6), len(cut) + 16)
    assert why == "" and got == cut and info["blocks"] >= 3
