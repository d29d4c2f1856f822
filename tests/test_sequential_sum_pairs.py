"""Host model of the arithmetic behind k_mt_chunk_pairs / k_mt_marginal_walk (csrc/plsa_kernels.hpp): a left-to-right
float64 sum of non-negative multiples of 2^-53 below 1 -- the reference's `marginal[i] += ndarray[i, j]`, utils.py:24-29,
over numpy's random_sample draws -- evaluated from per-chunk (start parity -> increment) pairs must reproduce the
sequential roundings bit for bit, ties (round half to even) included.  The GPU tests pin the kernels themselves to
numpy's accumulation; this one pins the statement the kernels implement, on inputs the generator would hardly ever produce."""
import struct

import numpy as np
import pytest

L = 64


def _bits(d):
    return struct.unpack("<q", struct.pack("<d", d))[0]


def _from_bits(b):
    return struct.unpack("<d", struct.pack("<q", b))[0]


def _pair(X, sh):
    """increment of T = s / 2^(e-52) over one chunk for start parity 0 / 1; X in units of 2^-53, sh = e + 1"""
    half, mask = 1 << (sh - 1), (1 << sh) - 1
    out = []
    for p in (0, 1):
        t = p
        for x in X:
            q, r = x >> sh, x & mask
            t += q + (1 if (r > half or (r == half and ((t + q) & 1))) else 0)
        out.append(t - p)
    return out


def chunked_sum(xs):
    X = [int(x * 2.0 ** 53) for x in xs]
    assert all(float(v) * 2.0 ** -53 == x for v, x in zip(X, xs))           # draws are exact multiples of 2^-53
    nch = (len(xs) + L - 1) // L
    csum = [float(sum(X[c * L:(c + 1) * L])) for c in range(nch)]
    b, pre, slow = 0, 0.0, 0
    for c in range(nch):
        start = pre * 2.0 ** -53                                             # approximate prefix: only a GUESS of the binade
        pre += csum[c]
        e = (_bits(start) >> 52 & 0x7FF) - 1023 if start >= 1.0 else -1
        ok = False
        if e >= 0:
            lo = (e + 1023) << 52
            nb = b + _pair(X[c * L:(c + 1) * L], e + 1)[b & 1]
            ok = b >= lo and nb < lo + (1 << 52)                             # the sum starts AND ends inside binade e
        if ok:
            b = nb
        else:                                                               # binade crossing, s < 1, wrong guess: draw by draw
            slow += 1
            s = _from_bits(b)
            for x in xs[c * L:(c + 1) * L]:
                s = s + x
            b = _bits(s)
    return _from_bits(b), slow


def _cases():
    rs = np.random.RandomState(7)
    top = 1.0 - 2.0 ** -53
    for m in (1, 63, 64, 65, 1000, 20000):
        yield "uniform-%d" % m, rs.rand(m)
        yield "coarse-%d" % m, rs.randint(0, 1 << 20, m) / 2.0 ** 20
        # low bits chosen so that exact ties are the rule, not the exception
        yield "ties-%d" % m, np.minimum(rs.randint(0, 4, m) * 2.0 ** -53 + rs.choice([0.0, 0.5, 0.25, 0.75], m), top)
        yield "largest-%d" % m, np.full(m, top)
        yield "tiny-%d" % m, rs.randint(0, 8, m) * 2.0 ** -53


@pytest.mark.parametrize("name,xs", list(_cases()), ids=[n for n, _ in _cases()])
def test_chunk_pairs_reproduce_the_sequential_float64_sum(name, xs):
    xs = np.asarray(xs, np.float64)
    want = float(np.add.accumulate(xs)[-1])                                  # strictly sequential
    got, slow = chunked_sum(list(xs))
    assert _bits(got) == _bits(want), name
    # the draw-by-draw path stays the exception: one chunk per binade crossing (+ the start) for ordinary data
    if name.startswith("uniform") and len(xs) >= 1000:
        assert slow <= 2 + int(np.log2(len(xs)))
