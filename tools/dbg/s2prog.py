import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from posepipeline_amd import _lib as L
from posepipeline_amd.program import ProgramBuilder, Net

os.environ["POSEPIPE_SPLIT_S2_MIN_CIN"] = "16"
ctx = L.Context(0)
rng = np.random.default_rng(23)
c = 48
w = [(rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32) for co, ci in ((c, c), (96, c), (96, 96))]
b = [rng.standard_normal(co).astype(np.float32) for co in (c, 96, 96)]
xin = rng.standard_normal((5, 25, 37, c)).astype(np.float32)

def prog(depth):
    pb = ProgramBuilder()
    x = pb.buf(25, 37, c, name="input")
    if depth == 1:
        out = pb.buf(25, 37, c, name="output")
        pb.conv(x, w[0], b[0], pad=1, relu=L.PP_RELU_LAST, out=out)
        return pb.build()
    y1 = pb.conv(x, w[0], b[0], pad=1, relu=L.PP_RELU_LAST)
    if depth == 2:
        out = pb.buf(13, 19, 96, name="output")
        pb.conv(y1, w[1], b[1], pad=1, stride=2, relu=L.PP_RELU_LAST, out=out)
        return pb.build()
    y2 = pb.conv(y1, w[1], b[1], pad=1, stride=2, relu=L.PP_RELU_LAST)
    if depth == 3:
        out = pb.buf(13, 19, 96, name="output")
        pb.conv(y2, w[2], b[2], pad=1, relu=L.PP_RELU_LAST, out=out)
        return pb.build()
    y3 = pb.conv(y2, w[2], b[2], pad=1, relu=L.PP_RELU_LAST)
    out = pb.buf(13, 19, 96, name="output")
    pb.conv(y1, w[1], b[1], pad=1, stride=2, relu=L.PP_RELU_LAST, res1=y2, res2=y3, out=out)
    return pb.build()

for d in (1, 2, 3, 4):
    p = prog(d)
    outs = {}
    for mode in ("exact", "split_bf16", "split_f16"):
        net = Net(ctx, p, max_batch=5, numerics=mode)
        outs[mode] = net.forward(xin)
        kinds = net.conv_kinds()
    e = outs["exact"]
    print(d, kinds, "bf16 err", np.abs(outs["split_bf16"] - e).max() / np.abs(e).max(), "f16 err", np.abs(outs["split_f16"] - e).max() / np.abs(e).max(),
          "per-sample f16 err", [float(np.abs(outs["split_f16"][i] - e[i]).max()) for i in range(5)])
