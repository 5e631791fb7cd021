"""The build's three ways of turning a 64-bit location index into a bit position — the fp64 route for 64 <= m <= 2^19
(kernels.hip.h: mod_f64), the 32-bit Barrett quotient below 2^31 and the 64-bit one beyond — at the geometries and the
index values where each could be off by one: m at both ends of every range and at powers of two, indices that are exact
multiples of m, all-ones / all-zero halves, 2^64 - 1, values one either side of a multiple of m.

The indices are chosen, not hashed: bsg_build_hashed takes the four base hashes of an entry, and (h0, h1, 0, 0) makes
location(h, i) = h[i % 2] for every i (bloom/v3 location(): h[i%2] + i*h[2 + ...], oracle bo_location).  Expected bit
positions come from the oracle's location() and Python's big-integer %.
"""
import numpy as np
import pytest

from bloomsearch_amd._lib import DESC_DTYPE
from oracle import oracle as O

pytestmark = pytest.mark.gpu

U64 = (1 << 64) - 1

# (m, k): the fp64 route's ends and middle, its neighbours on the Barrett routes, powers of two on each, the C3 geometry
GEOMETRIES = [(64, 11), (65, 11), (127, 10), (128, 10), (130, 11), (1000, 7), (4096, 10), (65535, 10), (65536, 10), (281629, 10),
              (287552, 10), ((1 << 19) - 1, 10), (1 << 19, 10), ((1 << 19) + 1, 10), (1 << 20, 10), (63, 11), (15, 11), (1, 1), (2, 3),
              ((1 << 31) - 1, 4), (1 << 31, 4), ((1 << 31) + 11, 4)]


def crafted_indices(m: int, rng) -> list[int]:
    xs = {0, 1, m - 1, m, m + 1, U64, U64 - 1, U64 - m, U64 - m + 1, 1 << 32, (1 << 32) - 1, (1 << 32) + 1, 0xFFFFFFFF00000000,
          0x00000000FFFFFFFF, 1 << 63, (1 << 63) - 1, 1 << 52, (1 << 52) - 1, 1 << 53}
    top = U64 // m
    for q in (1, 2, 3, top, top - 1, top // 2, (1 << 32) // m, (1 << 32) // m + 1, (1 << 33) // max(m, 1)) + tuple(int(v) for v in rng.integers(0, top + 1, size=24, dtype=np.uint64)):
        for d in (-1, 0, 1):
            v = q * m + d
            if 0 <= v <= U64:
                xs.add(v)
    for hi in (0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF):          # every residue class of the low word beside the extreme high words
        for lo in (0, 1, m % (1 << 32), 0xFFFFFFFF, 0xFFFFFFFE):
            xs.add((hi << 32) | lo)
    xs.update(int(v) for v in rng.integers(0, U64, size=256, dtype=np.uint64, endpoint=True))
    return sorted(xs)


@pytest.mark.parametrize("m,k", GEOMETRIES)
def test_chosen_location_indices_land_on_the_oracles_bits(ctx, m, k):
    rng = np.random.default_rng(m * 31 + k)
    xs = crafted_indices(m, rng)
    if len(xs) % 2:
        xs.append(xs[0])
    nw = (m + 63) // 64
    stride = (nw + 15) // 16 * 16
    # one filter per PAIR of indices (h0, h1): a wrong bit cannot hide behind another entry's right one
    n = len(xs) // 2
    if stride * n > (1 << 27):                                                      # the 2^31-bit geometries: a few pairs only (32 MiB per bitset)
        n = 2
        xs = [m - 1, U64, (U64 // m) * m, (U64 // m) * m - 1]
    h = np.zeros((n, 4), dtype=np.uint64)
    h[:, 0] = np.array(xs[0:2 * n:2], dtype=np.uint64)
    h[:, 1] = np.array(xs[1:2 * n:2], dtype=np.uint64)
    desc = np.zeros(n, dtype=DESC_DTYPE)
    for f in range(n):
        desc[f] = (f * stride, m, k, 0)
    fstart = np.arange(n + 1, dtype=np.uint32)
    got = ctx.build_hashed(h, fstart, desc, stride * n)
    for f in range(n):
        want = np.zeros(nw, dtype=np.uint64)
        hh = tuple(int(v) for v in h[f])
        for i in range(k):
            x = O.location(hh, i)
            assert x == hh[i % 2]
            bit = x % m
            want[bit >> 6] |= np.uint64(1 << (bit & 63))
        words = got[f * stride: f * stride + nw]
        assert np.array_equal(words, want), "m=%d k=%d h0=%#x h1=%#x: bits %s, oracle %s" % (
            m, k, hh[0], hh[1], [int(w) * 64 + b for w in np.flatnonzero(words) for b in range(64) if int(words[w]) >> b & 1][:8],
            sorted({hh[0] % m, hh[1] % m}))


@pytest.mark.parametrize("m,k", [(281629, 10), (1 << 19, 10), (524287, 13), (64, 11), (100003, 30)])
def test_full_recurrence_with_wrapping_sums(ctx, m, k):
    """All four hash words in play, chosen so that h[i%2] + i*h[2 + ...] wraps 2^64 at some i and not at others."""
    rng = np.random.default_rng(m + k)
    n = 512
    h = rng.integers(0, U64, size=(n, 4), dtype=np.uint64, endpoint=True)
    h[: n // 4, 2:] |= np.uint64(0xF000000000000000)                               # large multipliers: wraps from i = 1 on
    h[n // 4: n // 2, :2] = np.uint64(U64) - h[n // 4: n // 2, 2:] * np.uint64(3)   # the sum passes 2^64 around i = 3
    nw = (m + 63) // 64
    stride = (nw + 15) // 16 * 16
    desc = np.zeros(2, dtype=DESC_DTYPE)
    desc[0] = (0, m, k, 0)
    desc[1] = (stride, m, k, 0)
    got = ctx.build_hashed(h, np.array([0, n // 2, n], dtype=np.uint32), desc, 2 * stride)
    for f, (a, b) in enumerate(((0, n // 2), (n // 2, n))):
        want = np.zeros(nw, dtype=np.uint64)
        for e in range(a, b):
            hh = tuple(int(v) for v in h[e])
            for i in range(k):
                bit = O.location(hh, i) % m
                want[bit >> 6] |= np.uint64(1 << (bit & 63))
        assert np.array_equal(got[f * stride: f * stride + nw], want)


@pytest.mark.parametrize("n_terms", [40, 300])        # 40: the few-term kernel (Barrett); 300: the many-term kernel (fp64 route below 2^19 bits)
@pytest.mark.parametrize("m,k", [(64, 11), (65, 7), (4096, 10), (281629, 10), ((1 << 19) - 1, 10), (1 << 19, 10), ((1 << 19) + 1, 10), (1 << 20, 4)])
def test_probe_with_chosen_term_hashes(ctx, m, k, n_terms):
    """The probe's side of the same routes: term hashes are an input of bsg_probe, so the location indices can be chosen here too.
    Bitsets are random words (half the bits set): a verdict is right only if every one of the k tested positions is."""
    from bloomsearch_amd import query as Q
    from bloomsearch_amd._lib import TERM_DTYPE
    rng = np.random.default_rng(m * 7 + k + n_terms)
    n_blocks = 3
    nw = (m + 63) // 64
    stride = (nw + 15) // 16 * 16
    desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
    words = rng.integers(0, U64, size=stride * n_blocks * 3, dtype=np.uint64, endpoint=True)
    for f in range(n_blocks * 3):
        desc[f] = (f * stride, m, k, 0)
    if m % 64:
        for f in range(n_blocks * 3):
            words[f * stride + nw - 1] &= np.uint64((1 << (m % 64)) - 1)
    xs = crafted_indices(m, rng)
    picks = [xs[int(i) % len(xs)] for i in rng.permutation(max(len(xs), 2 * n_terms))[: 2 * n_terms]]
    cb = Q.compile_queries([Q.Token("t%d" % i) for i in range(n_terms)])
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=TERM_DTYPE)
    assert len(terms) == n_terms
    terms["kind"] = kinds
    h = np.zeros((n_terms, 4), dtype=np.uint64)
    h[:, 0] = np.array(picks[0::2], dtype=np.uint64)
    h[:, 1] = np.array(picks[1::2], dtype=np.uint64)
    h[n_terms // 2:, 2:] = rng.integers(0, U64, size=(n_terms - n_terms // 2, 2), dtype=np.uint64, endpoint=True)   # half with the full recurrence
    terms["h"] = h
    aid = ctx.arena_load(words, desc)
    try:
        got = ctx.probe(aid, n_blocks, terms, ops, poff)
    finally:
        ctx.arena_free(aid)
    want = O.probe_batch(words, desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    assert np.array_equal(got, want)
    assert want.any() or k > 4                           # (with k = 10 random bits, few verdicts pass: the comparison above is the test)
