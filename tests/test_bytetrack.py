"""ByteTrack configuration (wrappers/mmtrack.py method "bytetrack") on the CPU: the YOLOX oracle against an independent
torch statement of mmdet's modules, parameter / FLOP counts of the program, and the product ByteTracker (host Python over
the C++ Kalman filter and Hungarian solver) against the oracle tracker on seeded multi-person sequences."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import bytetrack as obt
from oracle import yolox as oyx
from posepipeline_amd.models import synth, yolox
from posepipeline_amd.tracking import ByteTracker


def test_yolox_shapes_and_program():
    shapes = yolox.yolox_param_shapes(num_classes=80)
    n_params = sum(int(np.prod(s)) for k, s in shapes.items() if "running" not in k)
    assert 98.5e6 < n_params < 100.0e6                          # YOLOX-X: 99.1 M parameters
    sd = synth.synth_state_dict(yolox.yolox_param_shapes(), seed=6)
    prog = yolox.build_yolox_program(sd, 64, 96)
    assert prog.bufs[prog.named["input"]] == (64, 96, 4)
    assert [prog.bufs[prog.named[f"reg{l}"]] for l in range(3)] == [(8, 12, 4), (4, 6, 4), (2, 3, 4)]
    full = yolox.build_yolox_program(sd, 640, 640)
    assert 138e9 < full.flops / 2 < 144e9                       # 281.9 GFLOPs = 141 GMAC at 640 x 640 (mmdet model zoo)


def _torch_yolox(sd, x_nhwc):
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    P = "detector."

    def cm(x, name, stride=1):
        w = t[P + name + ".conv.weight"]
        y = F.conv2d(x, w, None, stride, w.shape[2] // 2)
        y = F.batch_norm(y, t[P + name + ".bn.running_mean"], t[P + name + ".bn.running_var"], t[P + name + ".bn.weight"],
                         t[P + name + ".bn.bias"], False, 0.03, 1e-3)
        return y * torch.sigmoid(y)

    def csp(x, name, blocks, ident):
        short, main = cm(x, name + ".short_conv"), cm(x, name + ".main_conv")
        for b in range(blocks):
            y = cm(cm(main, f"{name}.blocks.{b}.conv1"), f"{name}.blocks.{b}.conv2")
            main = y + main if ident else y
        return cm(torch.cat((main, short), 1), name + ".final_conv")

    x = torch.from_numpy(np.ascontiguousarray(np.transpose(x_nhwc, (0, 3, 1, 2))))
    x = torch.cat((x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]), 1)
    x = cm(x, "backbone.stem.conv")
    feats = []
    for i, (blocks, ident, spp) in enumerate(((4, True, False), (12, True, False), (12, True, False), (4, False, True))):
        s = f"backbone.stage{i + 1}"
        x = cm(x, s + ".0", 2)
        j = 1
        if spp:
            y = cm(x, s + ".1.conv1")
            x = cm(torch.cat([y] + [F.max_pool2d(y, k, 1, k // 2) for k in (5, 9, 13)], 1), s + ".1.conv2")
            j = 2
        x = csp(x, f"{s}.{j}", blocks, ident)
        if i >= 1:
            feats.append(x)
    inner = [feats[-1]]
    for k, idx in enumerate((2, 1)):
        high = cm(inner[0], f"neck.reduce_layers.{k}")
        inner[0] = high
        up = F.interpolate(high, scale_factor=2, mode="nearest")
        inner.insert(0, csp(torch.cat([up, feats[idx - 1]], 1), f"neck.top_down_blocks.{k}", 4, False))
    outs = [inner[0]]
    for idx in (0, 1):
        down = cm(outs[-1], f"neck.downsamples.{idx}", 2)
        outs.append(csp(torch.cat([down, inner[idx + 1]], 1), f"neck.bottom_up_blocks.{idx}", 4, False))
    res = ([], [], [])
    for l, o in enumerate(outs):
        f = cm(o, f"neck.out_convs.{l}")
        c = r = f
        for j in range(2):
            c = cm(c, f"bbox_head.multi_level_cls_convs.{l}.{j}")
            r = cm(r, f"bbox_head.multi_level_reg_convs.{l}.{j}")
        for k, (name, src) in enumerate((("cls", c), ("reg", r), ("obj", r))):
            p = f"{P}bbox_head.multi_level_conv_{name}.{l}"
            res[k].append(np.transpose(F.conv2d(src, t[p + ".weight"], t[p + ".bias"]).numpy(), (0, 2, 3, 1)))
    return res


def test_yolox_oracle_vs_torch():
    sd = synth.synth_state_dict(yolox.yolox_param_shapes(), seed=6)
    x = np.random.default_rng(6).uniform(0, 255, (1, 64, 96, 3)).astype(np.float32)
    ref = _torch_yolox(sd, x)
    got = oyx.YOLOXRef(sd).forward(x)
    for g_list, r_list in zip(got, ref):
        for g, r in zip(g_list, r_list):
            assert g.shape == r.shape
            np.testing.assert_allclose(g, r, rtol=3e-3, atol=3e-3 * max(1.0, float(np.abs(r).max())))


def _sequence(seed, n_frames=60, n_people=5):
    """boxes on linear tracks with jitter, score fluctuations across the .1 / .6 / .7 thresholds, dropouts, false positives"""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(100, 900, (n_people, 2))
    vel = rng.uniform(-6, 6, (n_people, 2))
    size = rng.uniform(40, 120, (n_people, 2)) * [1, 2.2]
    frames = []
    for t in range(n_frames):
        rows = []
        for p in range(n_people):
            if rng.uniform() < 0.12:
                continue
            c = pos[p] + vel[p] * t + rng.normal(0, 1.5, 2)
            wh = size[p] * rng.uniform(0.95, 1.05, 2)
            rows.append([c[0] - wh[0] / 2, c[1] - wh[1] / 2, c[0] + wh[0] / 2, c[1] + wh[1] / 2, rng.choice([0.95, 0.8, 0.65, 0.45, 0.2, 0.05])])
        for _ in range(rng.integers(0, 3)):
            c = rng.uniform(0, 1000, 2)
            rows.append([c[0], c[1], c[0] + 50, c[1] + 100, rng.uniform(0.02, 0.9)])
        rows = np.array(rows, np.float32).reshape(-1, 5)
        frames.append(rows[np.argsort(-rows[:, 4], kind="stable")])
    return frames


def test_bytetracker_product_equals_oracle():
    for seed in (1, 2, 3):
        frames = _sequence(seed)
        ref, got = obt.ByteTrackerRef(), ByteTracker()
        seen = set()
        for t, dets in enumerate(frames):
            a, b = ref.step(dets), got.step(dets)
            assert a.shape == b.shape and np.array_equal(a, b), (seed, t)
            seen |= set(int(i) for i in a[:, 0])
        assert len(seen) >= 5 and sorted(ref.tracks) == sorted(got.tracks)
        for i in ref.tracks:                                    # same filter state (C++ Kalman vs the numpy restatement)
            np.testing.assert_allclose(got.tracks[i][0], ref.tracks[i]["mean"], rtol=1e-9, atol=1e-9)


def test_bytetracker_known_answers():
    trk = ByteTracker()
    box = lambda x, s: [x, 100.0, x + 60.0, 260.0, s]
    r0 = trk.step(np.array([box(100, 0.9), box(400, 0.65)], np.float32))
    assert list(r0[:, 0]) == [0.0] and r0[0, 5] == np.float32(0.9)          # only score > init_track_thr (.7) starts a track
    r1 = trk.step(np.array([box(104, 0.9), box(400, 0.65)], np.float32))
    assert list(r1[:, 0]) == [0.0, 1.0]                                       # matched; the unmatched high-score box starts id 1
    r2 = trk.step(np.array([box(108, 0.3), box(404, 0.65)], np.float32))
    assert sorted(r2[:, 0]) == [0.0, 1.0]                                     # id 0 recovered by the LOW-score second association
    r3 = trk.step(np.array([box(112, 0.05)], np.float32))
    assert len(r3) == 0 and 1 not in trk.tracks and 0 in trk.tracks           # score <= .1 dropped; tentative id 1 dies on its first miss
    for _ in range(30):
        trk.step(np.array([box(900, 0.65)], np.float32))
    assert 0 not in trk.tracks                                                # confirmed track forgotten after num_frames_retain
