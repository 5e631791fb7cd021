"""Exactness where the reference is exact (map keys and matchRowBytes compare BYTES): MurmurHash3_x64_128 has
seed-independent internal-state collisions, so two different entries can share all four bloom/v3 base hashes.  This file
CONSTRUCTS such pairs (differences in three consecutive 8-byte words that cancel inside the block function, for any
prefix and any suffix) and checks that
  * the CPU oracle confirms them (same sum256, different bytes) — CPU test;
  * the device's distinct-entry tables notice the pair through the keyed fingerprint and flag the set (status 2: rebuild on
    the host path) instead of counting one entry where Go's map counts two — GPU test;
  * the device row matcher hands a row containing the partner of a condition string to the host matcher instead of
    accepting it — GPU test.
"""
import json

import numpy as np
import pytest

from oracle import oracle as O

M64 = (1 << 64) - 1
C1, C2 = 0x87C37B91114253D5, 0x4CF5AD432745937F
C1_INV, C2_INV = pow(C1, -1, 1 << 64), pow(C2, -1, 1 << 64)


def rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def rotr(x, r):
    return rotl(x, 64 - r)


def f1(k):       # the k1 mixing of murmur3_x64_128's block function: k1 *= c1; k1 = rotl(k1, 31); k1 *= c2
    return (rotl((k * C1) & M64, 31) * C2) & M64


def f1_inv(v):
    return (rotr((v * C2_INV) & M64, 31) * C1_INV) & M64


def f2(k):       # k2 *= c2; k2 = rotl(k2, 33); k2 *= c1
    return (rotl((k * C2) & M64, 33) * C1) & M64


def f2_inv(v):
    return (rotr((v * C1_INV) & M64, 33) * C2_INV) & M64


def collide(block1: bytes, k1_next: bytes):
    """Given 16 bytes of block i and the first 8 bytes of block i + 1, the three words of a colliding partner.
    h1 ^= f1(k1) with a difference in bit 36 becomes, after rotl 27, a difference in bit 63 = +2^63 through the additions
    and the multiply by 5; a difference in bit 32 of f2(k2) (bit 63 after rotl 31) cancels it in h2; a difference in bit 63
    of the next block's f1(k1) cancels it in h1.  The state after block i + 1 is identical, whatever the state before."""
    k1 = int.from_bytes(block1[:8], "little")
    k2 = int.from_bytes(block1[8:], "little")
    k3 = int.from_bytes(k1_next, "little")
    k1p = f1_inv(f1(k1) ^ (1 << 36))
    k2p = f2_inv(f2(k2) ^ (1 << 32))
    k3p = f1_inv(f1(k3) ^ (1 << 63))
    return k1p.to_bytes(8, "little") + k2p.to_bytes(8, "little"), k3p.to_bytes(8, "little")


def pair(seed: int):
    """(a, b): a is a 32-byte printable lower-case token the device walker emits unchanged; b is its colliding partner —
    arbitrary bytes (for these bit differences the partner of a printable word is never printable: its words differ from
    the original's by fixed multiples of the multiplier inverses), so b reaches the device as a host-walked entry or as a
    condition string, which are byte strings at the C-ABI."""
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789-_.:/", dtype=np.uint8)
    a = alphabet[rng.integers(0, len(alphabet), size=32)].tobytes()
    pb, pn = collide(a[:16], a[16:24])
    return a, pb + pn + a[24:]


def test_constructed_pairs_collide_in_all_four_base_hashes():
    rng = np.random.default_rng(5)
    for prefix_blocks in (0, 1, 3):
        for suffix in (b"", b"x", b"0123456789abcdef", b"a longer suffix that spans blocks"):
            prefix = rng.integers(0, 256, size=16 * prefix_blocks, dtype=np.uint8).tobytes()
            blk = rng.integers(0, 256, size=16, dtype=np.uint8).tobytes()
            nxt = rng.integers(0, 256, size=8, dtype=np.uint8).tobytes()
            rest = rng.integers(0, 256, size=8, dtype=np.uint8).tobytes()
            pb, pn = collide(blk, nxt)
            a = prefix + blk + nxt + rest + suffix
            b = prefix + pb + pn + rest + suffix
            assert a != b and len(a) == len(b)
            assert O.base_hashes(a) == O.base_hashes(b), (prefix_blocks, suffix)
    a, b = pair(1)
    assert a != b and O.base_hashes(a) == O.base_hashes(b)


@pytest.mark.gpu
def test_device_sets_flag_a_collision_pair_instead_of_counting_it_once(ctx):
    from bloomsearch_amd import ingest as I
    a, b = pair(2)
    rows_clean = [json.dumps({"msg": "hello %d" % i}).encode() for i in range(50)]
    row_a = b'{"k":"' + a + b'"}'
    # one partner alone is an ordinary entry; the same one many times is a duplicate (table hit and LDS cache hit)
    res = I.device_ingest(ctx, [rows_clean + [row_a] * 70], 0.001, flags=1)
    assert not res.status.any() and len(res.fallback_rows) == 0
    n_tokens = int(res.counts[0, 1])
    # the partner arrives as a host-walked entry (bsg_ingest_add_entries hashes and fingerprints it from its bytes) into the
    # table the device walker filled: equal hashes, another fingerprint -> the set is flagged, not counted as one entry
    for first in (False, True):
        ing = ctx.ingest_rows(rows_clean + [row_a] * 3, [0, len(rows_clean) + 3], [0], 1, None, 1)
        ctx.ingest_add_entries(ing, [b"plain", b] if first else [b, b"plain"], [0, 0], [1, 1])
        counts, status = ctx.ingest_finish(ing, 2)
        ctx.ingest_free(ing)
        assert list(status) == [2, 0] or list(status) == [2, 2], status
    # both partners as host-walked entries of one set; and a pair split over two sets only collides in their file-level union
    ing = ctx.ingest_rows(rows_clean, [0, 25, len(rows_clean)], [0, 0], 1, None, 1)
    ctx.ingest_add_entries(ing, [a, b], [0, 1], [1, 1])
    counts, status = ctx.ingest_finish(ing, 3)
    ctx.ingest_free(ing)
    assert list(status) == [0, 0, 2], status
    assert n_tokens > 0


@pytest.mark.gpu
def test_device_matcher_sends_a_hash_collision_with_a_condition_to_the_host(ctx):
    from bloomsearch_amd import query as Q
    a, b = pair(3)
    rows = [b'{"k":"' + a + b'"}', b'{"k":"other"}', b'{"' + a + b'":1}']

    def matcher(kinds, fields, tokens, ops):
        m = Q.CompiledMatcher(None)
        m.kinds, m.fields, m.tokens, m.prog_ops = kinds, fields, tokens, ops
        return m
    from bloomsearch_amd._lib import op, OP_TERM, OP_OR, KIND_FIELD, KIND_TOKEN, KIND_FIELD_TOKEN
    hits, handed_back = ctx.match_rows(rows, matcher([KIND_TOKEN], [b""], [a], [op(OP_TERM, 0)]))
    assert list(hits) == [True, False, False] and len(handed_back) == 0
    # the condition is the PARTNER: row 0 emits a token with the condition's hashes but not its bytes -> the host decides
    hits, handed_back = ctx.match_rows(rows, matcher([KIND_TOKEN], [b""], [b], [op(OP_TERM, 0)]))
    assert list(hits) == [False, False, False] and list(handed_back) == [0]
    hits, handed_back = ctx.match_rows(rows, matcher([KIND_FIELD, KIND_FIELD_TOKEN], [b, b"k"], [b"", b], [op(OP_TERM, 0), op(OP_TERM, 1), op(OP_OR, 2)]))
    assert list(hits) == [False, False, False] and sorted(handed_back) == [0, 2]
