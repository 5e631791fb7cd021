"""Concurrent bsg_query callers on the C2 arena (GPU box):  python tools/conc_lab.py [seconds] [inflight ...]
T native threads x one 3-term query per call against 1 / 10 arenas, every call going alone vs combined, with the combined
cycles' phase breakdown (collector's clock, microseconds per cycle)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bloomsearch_amd import conc, query as Q, synth
from bloomsearch_amd.arena import plan_blocks
from bloomsearch_amd.gpu import Context


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    inflights = [int(x) for x in sys.argv[2:]] or [2]
    B, rows = 1000, int(os.environ.get("ROWS", "10000"))
    print("host cpus: %d" % os.cpu_count(), flush=True)
    blocks = [synth.block_entry_sets(b * rows, rows) for b in range(B)]
    plan = plan_blocks(blocks, 0.001)
    with Context((0,)) as ctx:
        words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        arenas = [ctx.arena_load(words, plan.desc) for _ in range(12)]
        exprs = synth.make_queries(256, "c2", seed=1234)
        cb = Q.compile_queries(exprs)
        ops, poff, kinds = cb.arrays()
        from bloomsearch_amd import _lib
        terms = np.zeros(len(cb.term_strings), dtype=_lib.TERM_DTYPE)
        terms["h"] = ctx.hash_strings(cb.term_strings)
        terms["kind"] = kinds
        expected = ctx.probe(arenas[0], B, terms, ops, poff)
        hot = [int(x) for x in os.environ.get("HOT", "8").split(",")]
        inls = [int(x) for x in os.environ.get("INL", "1").split(",")]            # key 21: the short job list in the kernel arguments
        Ts = [int(x) for x in os.environ.get("TS", "1,16,64,256").split(",")]
        cases = ((1, arenas[:1]), (1, arenas), (10, arenas))
        for apc, pool in [cases[int(i)] for i in os.environ.get("CASES", "0,1,2").split(",")]:
            for T in Ts:
                ctx.set_lab(12, 0)
                for ring in [int(x) for x in os.environ.get("RING", "").split(",") if x]:      # key 22: workgroups from which a lone call's doorbell is a dispatch of its own
                    ctx.set_lab(22, ring)
                    a = conc.run(ctx, exprs, pool, B, expected, T, seconds, apc)
                    print("   ring beyond %d workgroups: alone %.3g q/s p50 %.0f p99 %.0f" % (ring, a["queries_per_s"], a["p50_us"], a["p99_us"]), flush=True)
                ctx.set_lab(22, 32)
                a = conc.run(ctx, exprs, pool, B, expected, T, seconds, apc)
                line = "T=%3d x %2d arena(s) of %2d: alone %.3g q/s p50 %.0f p99 %.0f cpu %.1f us/call (%.1f busy) |" % (T, apc, len(pool), a["queries_per_s"], a["p50_us"], a["p99_us"], a["cpu_us_per_call"], a["cpus_busy"])
                ctx.set_lab(12, 1)
                for inf in inflights:
                    for hm, inl in [(h, i) for h in hot for i in inls]:
                        ctx.set_lab(13, inf)
                        ctx.set_lab(16, hm)
                        ctx.set_lab(21, inl)
                        if "SPIN" in os.environ:
                            ctx.set_lab(17, int(os.environ["SPIN"]))
                        ctx.query_stats(reset=True)
                        prof = os.environ.get("PROF") == "1"
                        cpu = np.zeros(5, dtype=np.uint64)
                        if prof:
                            ctx.set_lab(20, 1)
                            ctx._check(ctx.L.bsg_lab_query_cpu(ctx.h, cpu.ctypes.data, 1))
                        r = conc.run(ctx, exprs, pool, B, expected, T, seconds, apc)
                        st = ctx.query_stats()
                        if prof:
                            ctx._check(ctx.L.bsg_lab_query_cpu(ctx.h, cpu.ctypes.data, 1))
                            ctx.set_lab(20, 0)
                            n = max(int(cpu[0]), 1)
                            line += " [callers' cpu per call: %.2f us in the combiner path = wait %.2f + duty %.2f + collecting %.2f (amortised)]" % (
                                cpu[1] / n / 1e3, cpu[2] / n / 1e3, cpu[3] / n / 1e3, cpu[4] / n / 1e3)
                        assert not (a["mismatches"] or a["errors"] or r["mismatches"] or r["errors"]), (a, r)
                        cyc = max(st["cycles"] - st["solo_calls"], 1)
                        line += " inflight %d hot>=%d inline %d: %.3g q/s (%.1fx) p50 %.0f p99 %.0f cpu %.1f us/call (%.1f busy), %.1f calls/cycle, %d solo; per combined cycle: prepare %.1f enqueue %.1f wait %.1f deal %.1f (scatter %.1f free %.1f retire %.1f wake %.1f) us, %.2f dispatches (%.2f hot) |" % (
                            inf, hm, inl, r["queries_per_s"], r["queries_per_s"] / a["queries_per_s"], r["p50_us"], r["p99_us"], r["cpu_us_per_call"], r["cpus_busy"], st["cycle_calls"] / max(st["cycles"], 1), st["solo_calls"],
                            st["ns_prepare"] / cyc / 1e3, st["ns_enqueue"] / cyc / 1e3, st["ns_wait"] / cyc / 1e3, st["ns_deal"] / cyc / 1e3, st["ns_scatter"] / cyc / 1e3, st["ns_free"] / cyc / 1e3, st["ns_retire"] / cyc / 1e3, st["ns_wake"] / cyc / 1e3,
                            st["dispatches"] / cyc, st["hot_arenas"] / cyc)
                print(line, flush=True)


if __name__ == "__main__":
    main()
