"""Crafted filter sections through the device decoder (GPU box):  python tools/fuzz_sections.py [first_seed] [n_seeds]
A section with a wrong checksum is the easy case; this sweep damages a section's header fields, lengths, m, k, bitset
length or tail and then RE-SEALS it with a correct CRC32C, so only the structural checks of k_decode_sections stand between
the bytes and an out-of-bounds read — through bsg_arena_load_sections, through the chunked stream in order, and through the
stream with its chunks scrambled (any order, duplicates, overlaps, chunks never sent), on a single-device and on a sharded context.  Every block must come back either rejected (status != 0, nil filters) exactly when the host codec
rejects it, or decoded to filters that probe like the host-parsed ones.  Exits non-zero on the first difference."""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import helpers as H
from bloomsearch_amd import host as Hst, query as Q
from bloomsearch_amd._lib import DESC_DTYPE
from bloomsearch_amd.gpu import Context
from oracle import oracle as O


def pick(rng, values):
    return values[int(rng.integers(0, len(values)))]


def reseal(body: bytes) -> bytes:
    return body + struct.pack("<I", Hst.crc32c(body))


def damage(rng, sec: bytes) -> bytes:
    body = bytearray(sec[:-4])
    mode = int(rng.integers(0, 9))
    if mode == 0 and body:                                  # flags
        body[0] = int(rng.integers(0, 256))
    elif mode == 1 and len(body) >= 5:                      # a length field: first filter's
        struct.pack_into("<I", body, 1, pick(rng, [0, 1, 23, 24, 25, len(body), 2 ** 31, 2 ** 32 - 1, int(rng.integers(0, 2 ** 32))]))
    elif mode == 2 and len(body) >= 29:                     # m / k / bitset length of the first filter (big-endian u64 each)
        field = int(rng.integers(0, 3))
        struct.pack_into(">Q", body, 5 + 8 * field, pick(rng, [0, 1, 63, 64, 65, 2 ** 31, 2 ** 32, 2 ** 63, 2 ** 64 - 1, int(rng.integers(0, 2 ** 40))]))
    elif mode == 3:                                         # truncate anywhere
        del body[int(rng.integers(0, len(body) + 1)):]
    elif mode == 4:                                         # random tail appended
        body += rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8).tobytes()
    elif mode == 5 and body:                                # a random byte anywhere
        body[int(rng.integers(0, len(body)))] = int(rng.integers(0, 256))
    elif mode == 6:                                         # nothing but a header byte
        body = bytearray([int(rng.integers(0, 8))])
    elif mode == 7:                                         # empty
        body = bytearray()
    # mode 8: left intact
    return reseal(bytes(body))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    ctxs = [Context((0,)), Context((0,) * 3)]
    n_sec = n_bad = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        nb = int(rng.integers(1, 80))
        plan, _, vocab = H.make_random_arena(rng, nb, fpr=float(rng.choice([0.3, 0.01, 0.001])), absent_frac=0.1, max_tokens=int(rng.choice([4, 60, 800])), vocab_size=50)
        ctx = ctxs[seed % 2]
        good = ctx.build_sections(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        secs = [damage(rng, s) if rng.random() < 0.6 else s for s in good]
        # what the host codec makes of every section
        host_ok, words, desc = [], [], np.zeros(nb * 3, dtype=DESC_DTYPE)
        cursor = 0
        for b, s in enumerate(secs):
            try:
                fl = Hst.section_parse(s)
                host_ok.append(True)
                for c, f in enumerate(fl):
                    if f is None:
                        continue
                    m, k, w = f
                    desc[b * 3 + c] = (cursor, m, k, 0)
                    words.append(np.asarray(w, dtype=np.uint64))
                    pad = (-len(w)) % 16
                    if pad:
                        words.append(np.zeros(pad, dtype=np.uint64))
                    cursor += len(w) + pad
            except Exception:                       # noqa: BLE001 - whatever the host rejects, the device must reject
                host_ok.append(False)
        host_words = np.concatenate(words) if words else np.zeros(2, dtype=np.uint64)
        cb = Q.compile_queries([None] + [H.random_expression(rng, vocab, None) for _ in range(40)])
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        want = O.probe_batch(host_words, desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
        for how in ("whole", "stream", "scrambled"):
            unread = set()
            if how == "whole":
                aid, st = ctx.arena_load_sections(secs)
            else:
                blob = b"".join(secs)
                offs = np.zeros(len(secs) + 1, dtype=np.uint64)
                offs[1:] = np.cumsum([len(x) for x in secs])
                sid = ctx.arena_stream_begin(offs[:-1], offs[1:])
                step = int(rng.choice([1, 7, 4096, 1 << 20])) if how == "stream" else int(rng.choice([13, 200, 4096, 50000]))
                chunks = [(o, min(o + step, len(blob))) for o in range(0, len(blob), step)]
                if how == "scrambled":
                    # chunks in any order, some twice, some widened over their neighbours, some never sent: a section is decoded
                    # exactly when every byte of it arrived, whatever the order; the others report -7 and stay nil
                    keep = [c for c in chunks if rng.random() > 0.15] or chunks[:1]
                    sent = keep + [keep[int(i)] for i in rng.integers(0, len(keep), size=len(keep) // 5)]
                    sent = [(max(0, a - int(rng.integers(0, 40))), min(len(blob), b + int(rng.integers(0, 40)))) if rng.random() < 0.2 else (a, b) for a, b in sent]
                    order = rng.permutation(len(sent))
                    chunks = [sent[int(i)] for i in order]
                    covered = np.zeros(len(blob) + 1, dtype=np.int32)
                    for a, b in chunks:
                        covered[a] += 1; covered[b] -= 1
                    have = np.cumsum(covered)[:-1] > 0
                    for b in range(nb):
                        lo, hi = int(offs[b]), int(offs[b + 1])
                        if hi > lo and not have[lo:hi].all():
                            unread.add(b)
                for a, b in chunks:
                    ctx.arena_stream_append(sid, a, blob[a:b])
                aid, st = ctx.arena_stream_finish(sid, len(secs))
            dev_ok = [int(x) == 0 for x in st]
            # (an empty section — no bytes at all — is "the block has no section": status 0 and nil filters on both sides)
            for b in range(nb):
                if len(secs[b]) == 0:
                    continue
                if b in unread:
                    if int(st[b]) != -7:
                        sys.exit("seed %d block %d (%s): section never completely sent, status %d instead of -7" % (seed, b, how, int(st[b])))
                    continue
                if dev_ok[b] != host_ok[b]:
                    sys.exit("seed %d block %d (%s): device status %d, host codec %s the section (%d bytes)"
                             % (seed, b, how, int(st[b]), "accepts" if host_ok[b] else "rejects", len(secs[b])))
            expect = want
            if unread:
                d2 = desc.copy()
                for b in unread:
                    d2["m"][b * 3: b * 3 + 3] = 0
                expect = O.probe_batch(host_words, d2.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
            got = ctx.probe(aid, nb, terms, ops, poff)
            if not np.array_equal(got, expect):
                sys.exit("seed %d (%s): decoded filters probe differently from the host-parsed ones" % (seed, how))
            ctx.arena_free(aid)
        n_sec += nb
        n_bad += sum(1 for x in host_ok if not x)
        if (seed - first) % 20 == 19:
            print("seed %d ok (%d sections so far, %d rejected by both sides)" % (seed, n_sec, n_bad), flush=True)
    for c in ctxs:
        c.close()
    print("done: %d sections, %d rejected by both sides, no difference" % (n_sec, n_bad))


if __name__ == "__main__":
    main()
