"""Where the time of a large bsg_match_rows call goes (GPU box): BSG_LAB_TRACE=1 python tools/match_trace.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bloomsearch_amd import synth, query as Q
from bloomsearch_amd.gpu import Context
from benchlib import common as bench
nb=300
parts=[bench._gen_rows((b,10000,1)) for b in range(nb)]
blob=np.frombuffer(b"".join(p[0] for p in parts),dtype=np.uint8)
lens=np.concatenate([p[1] for p in parts]); off=np.zeros(len(lens)+1,dtype=np.uint64); np.cumsum(lens,out=off[1:])
with Context((0,)) as ctx:
    m=Q.CompiledMatcher(Q.FieldToken("level","error"))
    ctx.match_rows((blob,off),m)
    pinned=ctx.pinned_array(len(blob)); pinned[:]=blob
    for name,src in (("pageable",blob),("pinned",pinned),("pinned",pinned)):
        t0=time.time(); ctx.match_rows((src,off),m); print(name, "%.1f ms"%((time.time()-t0)*1e3), "kernel %.2f"%ctx.last_match_ms(), file=sys.stderr)
