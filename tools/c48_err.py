import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from posepipeline_amd import _lib as L
from tests.helpers import hip_conv_op
from tests.test_gpu_split import conv64
ctx = L.Context(0); lib = ctx.lib
for case in [(3,20,34,96,48),(2,9,40,256,40),(1,64,48,16,36),(2,96,72,48,48),(2,16,16,48,48),(3,20,34,96,64),(2,96,72,48,64)]:
    n,h,w,cin,cout = case
    rng = np.random.default_rng(sum(case))
    for kind in ("wide","normal"):
        x = rng.standard_normal((n,h,w,cin))
        if kind=="wide": x = x*np.exp(2*rng.standard_normal((n,h,w,cin)))
        x = x.astype(np.float32)
        wt = (rng.standard_normal((cout,cin,3,3))/np.sqrt(cin*9)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        ref = conv64(x, wt, b, 1)
        out = {}
        for name, ex in (("exact",1),("split",0)):
            lib.pp_conv_exact(ex)
            out[name] = hip_conv_op(ctx, x, wt, b, pad=(1,1))
        sc = np.abs(ref).max()
        e = {k: (np.abs(v-ref).max()/sc, np.sqrt(np.mean((v-ref)**2))/sc) for k,v in out.items()}
        print(case, kind, "exact max %.2e rms %.2e | split max %.2e rms %.2e | ratio max %.2f rms %.2f" % (*e["exact"], *e["split"], e["split"][0]/e["exact"][0], e["split"][1]/e["exact"][1]))
