#!/usr/bin/env python
"""Reads the per-workgroup timeline a -DPP_SPLIT_TIMELINE build of conv_split.hip appends to $POSEPIPE_SPLIT_TIMELINE and prints, per
layer class (kernel form, map, channels): where a workgroup's time goes (s_memtime ticks = shader cycles: index set-up, first patch,
rest of the prologue, K loop, epilogue until the loads are consumed, stores issued, stores acknowledged) and how well the resident
workgroups of a CU cover each other (fraction of the launch during which >= 1 / >= 2 of a CU's workgroups are inside their K loop).
s_memtime counters of different XCDs have different bases: times are taken relative to the XCD's first workgroup of the launch.

  bash tools/build_variant.sh tl -DPP_SPLIT_TIMELINE
  POSEPIPE_LIB=posepipeline_amd/libposepipe_hip_tl.so POSEPIPE_SPLIT_TIMELINE=/tmp/tl.bin python tools/profile_net.py w48 128
  python tools/split_timeline.py /tmp/tl.bin
"""
import sys
from collections import defaultdict

import numpy as np

MAGIC = 0x54494D454C494E45


def launches(path):
    raw = np.fromfile(path, dtype=np.int64)
    i = 0
    while i + 16 <= len(raw):
        h = raw[i:i + 16]
        assert h[0] == MAGIC, "bad header"
        nwg = int(h[2])
        rec = raw[i + 16:i + 16 + nwg * 16].reshape(nwg, 16).view(np.uint64)
        name = h[14:16].tobytes().split(b"\0")[0].decode()
        yield dict(id=int(h[1]), nwg=nwg, N=int(h[3]), H=int(h[4]), W=int(h[5]), cin=int(h[6]), cout=int(h[7]), mode=int(h[8]),
                   nchunks=int(h[9]), gx=int(h[10]), gy=int(h[11]), res=int(h[12]) + int(h[13]), kernel=name), rec
        i += 16 + nwg * 16


def analyse(rec, mfma_cycles):
    """s_memtime has a different base on every CU (measured: starts of one launch spread over 1e7 ticks inside one XCD), so every
    time is taken relative to the CU's first workgroup of the launch"""
    rec = rec[rec[:, 5] > 0]
    xcc = (rec[:, 1] >> np.uint64(32)).astype(np.int64)
    cu = (xcc << 16) | ((rec[:, 1].astype(np.int64) >> 8) & 0xFF)
    T = {k: rec[:, c].astype(np.int64) for k, c in dict(t0=2, t1=3, t2=4, t3=5, s4=8, s5=9, e6=10, e7=11).items()}
    seg = dict(setup=T["s4"] - T["t0"], patch0=T["s5"] - T["s4"], pro_rest=T["t1"] - T["s5"], loop=T["t2"] - T["t1"],
               epi_loads=T["e6"] - T["t2"], epi_issue=T["t3"] - T["e6"], epi_ack=T["e7"] - T["t3"], life=T["e7"] - T["t0"])
    f1 = f2 = util = 0.0
    spans, starts, per_cu = [], [], []
    cus = np.unique(cu)
    for c in cus:
        sel = cu == c
        b = T["t0"][sel].min()
        span = int(T["e7"][sel].max() - b)
        spans.append(span)
        starts.append(T["t0"][sel] - b)
        per_cu.append(int(sel.sum()))
        ev = sorted([(int(a - b), 1) for a in T["t1"][sel]] + [(int(e - b), -1) for e in T["t2"][sel]])
        depth, last, c1, c2 = 0, 0, 0, 0
        for t, d in ev:
            if depth >= 1:
                c1 += t - last
            if depth >= 2:
                c2 += t - last
            depth += d
            last = t
        f1 += c1 / span
        f2 += c2 / span
        util += sel.sum() * mfma_cycles / span
    n = len(cus)
    return np.median(spans), max(spans), n, f1 / n, f2 / n, util / n, np.concatenate(starts), seg, (min(per_cu), max(per_cu))


def main():
    groups = defaultdict(list)
    for h, rec in launches(sys.argv[1]):
        groups[(h["kernel"], h["H"], h["W"], h["cin"], h["cout"], h["N"], h["res"])].append((h, rec))
    print("kernel HxW cin->cout xN res | launches WGs chunks grid WGs/CU | CU span p50 / max | medians: setup patch0 pro_rest LOOP epi_loads "
          "epi_issue epi_ack life | mfma/WG | CU-time with >=1 / >=2 WGs in loop | matrix-pipe utilisation (sum of MFMA cycles / CU span)")
    tot = {}
    for key, ls in groups.items():
        h = ls[0][0]
        cob = 1 if (((h["cout"] + 31) // 32) & 1) else 2
        # matrix-pipe cycles one workgroup needs on each SIMD: MFMAs per wave x cycles x waves per SIMD (8-wave forms: 2)
        if key[0] == "c48":
            mf = h["nchunks"] * 5760
        elif key[0].startswith("g") and key[0] != "gemm4":       # product kernel: a wave = 128 pixels x 64 channels per 16-channel stage
            mf = h["nchunks"] * 48 * 32 * (1 if key[0] == "g256x128" else 2)
        elif key[0] == "gemm4":
            mf = h["nchunks"] * cob * 2 * 6 * 32
        else:
            mf = h["nchunks"] * 9 * cob * 2 * 6 * 32 * (2 if key[0] == "nw8" else 1)
        out = [analyse(rec, mf) for _, rec in ls[:6]]
        tot[key] = (np.median([o[0] for o in out]) * len(ls), ls, out, mf)
    for key, (w, ls, out, mf) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        h = ls[0][0]
        med = {k: np.median(np.concatenate([o[7][k] for o in out])) for k in out[0][7]}
        print(f"{key[0]:6s} {key[1]:3d}x{key[2]:<3d} {key[3]:4d}->{key[4]:<4d} x{key[5]:<4d} res{key[6]} | {len(ls):3d} {h['nwg']:5d} {h['nchunks']:3d} "
              f"{h['gx']}x{h['gy']} {out[0][8][0]}-{out[0][8][1]} | {np.median([o[0] for o in out]):8.0f} {np.median([o[1] for o in out]):8.0f} | "
              f"{med['setup']:6.0f} {med['patch0']:6.0f} {med['pro_rest']:6.0f} {med['loop']:8.0f} {med['epi_loads']:6.0f} {med['epi_issue']:6.0f} "
              f"{med['epi_ack']:6.0f} {med['life']:8.0f} | {mf:7d} | {np.mean([o[3] for o in out]):5.3f} {np.mean([o[4] for o in out]):5.3f} | "
              f"{np.mean([o[5] for o in out]):5.3f}")


if __name__ == "__main__":
    main()
