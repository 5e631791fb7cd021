"""Which library call leaves a sticky HIP error behind (hipGetLastError after every step of test_c5's scenario)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bloomsearch_amd._lib import DESC_DTYPE
from bloomsearch_amd.gpu import Context, pack_entries
from oracle import oracle as O
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
def last(tag):
    e = hip.hipGetLastError()
    print("%-40s hipGetLastError = %d %s" % (tag, e, hip.hipGetErrorString(e).decode() if e else ""), flush=True)
n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 50
per_block = [["b%dt%d" % (b, i) for i in range(300)] + ["shared%d" % i for i in range(200)] for b in range(n_blocks)]
union = set(t for p in per_block for t in p)
m, k = O.estimate_parameters(len(union), 0.001)
nw = O.words_for(m); stride = (nw + 15) // 16 * 16
desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
fstart, ents = [0], []
for b in range(n_blocks):
    fstart.append(len(ents)); desc[b * 3 + 1] = (b * stride, m, k, 0); ents += per_block[b]; fstart += [len(ents), len(ents)]
blob, off = pack_entries(ents)
print("m =", m, "words", nw)
with Context((0,)) as ctx:
    last("open")
    words = ctx.build(blob, off, np.asarray(fstart, dtype=np.uint32), desc, n_blocks * stride); last("build")
    aid = ctx.arena_load(words, desc); last("arena_load")
    got = ctx.or_reduce(aid, 1, nw); last("or_reduce")
    ctx.arena_free(aid); last("arena_free")
    try:
        ctx.or_allreduce(aid, 1, nw)
    except Exception as e:
        print("expected:", str(e)[:80])
    last("or_allreduce without comm")
    ctx.comm_init(Context.comm_unique_id(), 0, 1); last("comm_init")
    ctx.comm_destroy()
