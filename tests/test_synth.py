"""The workload generators (synth.py, beside bench.py): every random number is one word of a splitmix64 stream addressed by
its index, so that a C++ / Rust harness can regenerate the bench inputs."""
import hashlib

import numpy as np
import torch

import synth


def test_splitmix64_words_match_the_published_generator():
    # splitmix64 seeded with 0: the first outputs of the reference implementation (Vigna)
    want = [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F, 0xF88BB8A8724C81EC]
    got_t = synth.sm64(0, torch.arange(4, dtype=torch.int64)).numpy().view(np.uint64).tolist()
    got_n = synth.sm64_np(0, np.arange(4)).tolist()
    py = []
    state = 0
    for _ in range(4):
        state = (state + 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
        py.append(synth.mix_int(state))
    assert got_t == want and got_n == want and py == want
    # arbitrary seeds and far-away indices: torch (wrapping int64) and numpy (uint64) agree with Python integers
    for seed in (1, 20250711, synth.stream(20250711, 3), (1 << 64) - 5):
        idx = np.array([0, 1, 2, 999_999_937, (1 << 40) + 17], dtype=np.int64)
        exp = [synth.mix_int(seed + (int(i) + 1) * 0x9E3779B97F4A7C15) for i in idx]
        assert synth.sm64(seed, torch.from_numpy(idx)).numpy().view(np.uint64).tolist() == exp
        assert synth.sm64_np(seed, idx).tolist() == exp


def test_small_workload_is_reproducible_and_follows_the_documented_rules():
    dev = torch.device("cpu")
    g = synth.random_genomes(5, 20_000, dev, seed=3, mutated_frac=0.2, identity=0.9)
    assert set(np.unique(g.numpy()).tolist()) <= {65, 67, 71, 84}
    # base i of genome 2 = top two bits of word i of stream(3, 2)
    w = synth.sm64_np(synth.stream(3, 2), np.arange(20_000))
    assert np.array_equal(g[2].numpy(), np.frombuffer(b"ACGT", dtype=np.uint8)[(w >> np.uint64(62)).astype(np.int64)])
    ident = float((g[4] == g[0]).float().mean())
    assert 0.88 < ident < 0.95                                            # a 90 %-identity copy (substitutions only)
    b1, o1 = synth.paired_reads(g, 3000, seed=11)
    b2, o2 = synth.paired_reads(g, 3000, seed=11)
    assert torch.equal(b1, b2) and torch.equal(o1, o2) and b1.numel() == 3000 * 300 + 64
    assert hashlib.sha256(b1.numpy().tobytes()).hexdigest() != hashlib.sha256(synth.paired_reads(g, 3000, seed=12)[0].numpy().tobytes()).hexdigest()
    # duplicates: floor(2 %) pairs equal some other pair of the set
    pairs = b1[:3000 * 300].reshape(3000, 300).numpy()
    _, counts = np.unique(pairs, axis=0, return_counts=True)
    assert (counts > 1).sum() >= 30
    rb, ro = synth.ragged_paired_reads(g, 500, seed=5)
    lens = np.diff(ro.numpy())
    assert lens.min() >= 35 and lens.max() <= 151 and rb.numel() == int(ro[-1]) + 64
    dk, doff = synth.decoy_sketches(50, c=200, device=dev, seed=7)
    assert int(doff[-1]) == dk.numel() and int(dk.max()) < (2**64 - 1) // 200 and int(dk.min()) >= 0
    lb, lo = synth.long_reads(g, 300_000, seed=9)
    assert int(lo[-1]) >= 300_000 and lb.numel() == int(lo[-1]) + 64


def test_long_reads_indels_event_by_event():
    """C5's error model (5 % errors, substitution : insertion : deletion = 2 : 1 : 1): a few reads of the generator
    replayed base by base from the words of the error stream — plain Python, no tensors — and the error mix of a whole set."""
    import math
    dev = torch.device("cpu")
    seed = 9
    g = synth.random_genomes(3, 200_000, dev, 3, mutated_frac=0.0)
    glen = g.shape[1]
    b, o = synth.long_reads(g, 300_000, seed=seed)
    b, o = b.numpy(), o.numpy()
    n = len(o) - 1
    flat = g.reshape(-1).numpy()
    cum = np.cumsum(synth.abundance_weights(3, seed, 1.0))
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    mask = (1 << 64) - 1
    word = lambda st, i: synth.mix_int(st + (i + 1) * 0x9E3779B97F4A7C15)   # noqa: E731
    s21, s22, s23 = synth.stream(seed, 21), synth.stream(seed, 22), synth.stream(seed, 23)
    err_thr = int(0.05 * (1 << 32))
    kinds = [0, 0, 0]
    for r in list(range(4)) + [n - 1]:
        L = int(o[r + 1] - o[r])
        gid = min(int(np.searchsorted(cum, (word(s21, r) >> 32) % int(cum[-1]), side="right")), 2)
        ws = word(s22, r)
        win = min(L + (L >> 3) + 8, glen)
        start = ((ws >> 32) * (glen - win)) >> 32
        flip = ws & 1
        p = 0
        for j in range(L):
            we = word(s23, int(o[r]) + j) & mask
            is_err = (we >> 32) < err_thr
            kind = (we >> 32) & 3
            rnd = (65, 67, 71, 84)[(we >> 30) & 3]
            if is_err and kind == 3:
                p += 1
            q = min(p, win - 1)
            src = flat[gid * glen + start + (win - 1 - q if flip else q)]
            src = comp[int(src)] if flip else int(src)
            exp = rnd if (is_err and kind != 3) else src
            assert int(b[int(o[r]) + j]) == exp, (r, j)
            if not (is_err and kind == 2):
                p += 1
            if is_err:
                kinds[0 if kind < 2 else kind - 1] += 1
    tot = sum(kinds)
    assert tot > 100 and abs(kinds[0] / tot - 0.5) < 0.1 and abs(kinds[1] / tot - 0.25) < 0.08 and abs(kinds[2] / tot - 0.25) < 0.08
    assert abs(tot / sum(int(o[r + 1] - o[r]) for r in list(range(4)) + [n - 1]) - 0.05) < 0.01
    assert math.isclose(float(np.mean(np.isin(b[:int(o[-1])], [65, 67, 71, 84]))), 1.0)
