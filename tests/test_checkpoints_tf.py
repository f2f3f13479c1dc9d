"""Checkpoint readers of tracking_method 0 (posepipeline_amd/checkpoints_tf.py) on files written in the same formats: a
frozen TensorFlow GraphDef (protobuf wire format, encoded by hand below -- TensorFlow is not installed) with the slim
variable names tools/freeze_model.py produces, and the per-layer weight dictionary of a Keras .h5.  The real
mars-small128.pb / yolo4.h5 are not available here: what is tested is the wire decoding, the name mapping and the layout
transposes, by round trip from seeded parameters."""
import struct

import numpy as np
import pytest

from posepipeline_amd import checkpoints_tf as ck
from posepipeline_amd.models import mars, yolov4


# ---- a minimal protobuf encoder (test side only) -------------------------------------------------------------------------
def varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def field(num, wt, payload):
    if wt == 0:
        return varint(num << 3) + varint(payload)
    if wt == 2:
        return varint((num << 3) | 2) + varint(len(payload)) + payload
    return varint((num << 3) | wt) + payload


def tensor_proto(arr, mode="content"):
    arr = np.asarray(arr, np.float32)
    shape = b"".join(field(2, 2, field(1, 0, d)) for d in arr.shape)
    msg = field(1, 0, ck.DT_FLOAT) + field(2, 2, shape)
    if mode == "content":
        msg += field(4, 2, arr.astype("<f4").tobytes())
    elif mode == "packed":
        msg += field(5, 2, arr.astype("<f4").tobytes())
    elif mode == "repeated":
        msg += b"".join(field(5, 5, struct.pack("<f", v)) for v in arr.reshape(-1))
    elif mode == "fill":
        msg += field(5, 5, struct.pack("<f", float(arr.reshape(-1)[0])))
    return msg


def node(name, op, tensor=None, inputs=()):
    msg = field(1, 2, name.encode()) + field(2, 2, op.encode())
    for i in inputs:
        msg += field(3, 2, i.encode())
    msg += field(5, 2, field(1, 2, b"dtype") + field(2, 2, field(6, 0, ck.DT_FLOAT)))       # attr dtype: AttrValue.type = 6
    if tensor is not None:
        msg += field(5, 2, field(1, 2, b"value") + field(2, 2, field(8, 2, tensor)))
    return field(1, 2, msg)


def test_graphdef_wire_decoding():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((3, 3, 4, 8)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    graph = (node("images", "Placeholder") + node("w", "Const", tensor_proto(a)) + node("b_packed", "Const", tensor_proto(b, "packed")) +
             node("b_rep", "Const", tensor_proto(b, "repeated")) + node("fill", "Const", tensor_proto(np.full((2, 5), 0.25), "fill")) +
             node("scalar", "Const", tensor_proto(np.float32(1e-8), "repeated")) +
             node("conv", "Conv2D", inputs=("images", "w")) + field(4, 2, field(1, 0, 27)))       # versions { producer: 27 }
    c = ck.read_graphdef_consts(graph)
    assert sorted(c) == ["b_packed", "b_rep", "fill", "scalar", "w"]
    assert np.array_equal(c["w"], a) and np.array_equal(c["b_packed"], b) and np.array_equal(c["b_rep"], b)
    assert c["fill"].shape == (2, 5) and (c["fill"] == 0.25).all() and c["scalar"].shape == () and c["scalar"] == np.float32(1e-8)


@pytest.mark.parametrize("prefix", ["", "net/"])
def test_mars_graphdef_round_trip(prefix, tmp_path):
    shapes = mars.mars_param_shapes()
    sd = yolov4.synth_params(shapes, seed=5)
    leaf = {"weight": "weights", "bias": "biases", "beta": "beta", "mean": "moving_mean", "var": "moving_variance"}
    nodes = b""
    for name, arr in sd.items():
        tok = name.split(".")
        scope = tok[:-1]
        if "bn" in scope and scope[0] not in ("ball",) and len(scope) >= 2 and scope[-1] == "bn":
            inner = scope[:-1]
            # slim: a batch norm created INSIDE conv2d / fully_connected repeats the layer's scope (conv1_1/conv1_1/bn/beta);
            # the pre-activation batch norm of a residual link is created at top level (conv2_3/bn/beta)
            inside = ".".join(inner + ["weight"]) in sd
            scope = inner + inner + ["bn"] if inside else scope
        if tok[-1] == "weight":
            arr = arr.T if arr.ndim == 2 else np.transpose(arr, (2, 3, 1, 0))             # OIHW -> HWIO, [out][in] -> [in][out]
        nodes += node(prefix + "/".join(scope + [leaf[tok[-1]]]), "Const", tensor_proto(arr))
    nodes += node(prefix + "Const_eps", "Const", tensor_proto(np.float32(1e-8), "repeated"))
    path = tmp_path / "mars-small128.pb"
    path.write_bytes(nodes)
    got = ck.mars_params(str(path), shapes)
    assert got.keys() == sd.keys()
    for k in sd:
        assert np.array_equal(got[k], sd[k]), k
    assert ck._slim_name("net/conv2_1/1/conv2_1/1/bn/moving_mean") == "net.conv2_1.1.bn.mean"
    assert ck._slim_name("conv3_1/projection/weights") == "conv3_1.projection.weight" and ck._slim_name("map/while/Enter") is None


@pytest.mark.parametrize("first_index", [0, 1])
def test_yolo4_keras_mapping(first_index, tmp_path):
    shapes = yolov4.yolov4_param_shapes()
    sd = yolov4.synth_params(shapes, seed=4)
    n_conv = sum(k.endswith(".weight") for k in shapes)

    def kname(base, i):
        return base if (i == 0 and first_index == 0) else f"{base}_{i if first_index == 0 else i + 1}"

    layers, j = {"input_1": {}, "mish_3": {}, "add_1": {}}, 0
    for i in range(n_conv):
        cn = kname("conv2d", i)
        layers[cn] = {f"{cn}/kernel:0": np.transpose(sd[f"l{i}.weight"], (2, 3, 1, 0))}
        if f"l{i}.bias" in sd:
            layers[cn][f"{cn}/bias:0"] = sd[f"l{i}.bias"]
        else:
            bn = kname("batch_normalization", j)
            j += 1
            layers[bn] = {f"{bn}/gamma:0": sd[f"l{i}.bn.gamma"], f"{bn}/beta:0": sd[f"l{i}.bn.beta"],
                          f"{bn}/moving_mean:0": sd[f"l{i}.bn.mean"], f"{bn}/moving_variance:0": sd[f"l{i}.bn.var"]}
    got = ck.yolo_params_from_keras(dict(reversed(list(layers.items()))), shapes)        # file order must not matter
    assert got.keys() == sd.keys() and all(np.array_equal(got[k], sd[k]) for k in sd)
    # the converted .npz next to the .h5 is what the wrapper loads without h5py
    np.savez(tmp_path / "yolo4.npz", **got)
    again = ck.yolo_params(str(tmp_path / "yolo4.h5"), shapes)
    assert all(np.array_equal(again[k], sd[k]) for k in sd)
    del layers[kname("conv2d", 5)]
    with pytest.raises(ValueError, match="convolutions"):
        ck.yolo_params_from_keras(layers, shapes)


def test_wrapper_prefers_installed_checkpoints(tmp_path, monkeypatch):
    from posepipeline_amd.wrappers.deep_sort_yolov4 import parser
    monkeypatch.setenv("PIPELINE_3RDPARTY", str(tmp_path))
    monkeypatch.delenv("POSEPIPE_SYNTHETIC_WEIGHTS", raising=False)
    shapes = mars.mars_param_shapes()
    with pytest.raises(FileNotFoundError):
        parser._params("deep_sort_yolov4/mars-small128.pb", shapes, seed=5)
    (tmp_path / "deep_sort_yolov4").mkdir()
    sd = yolov4.synth_params(shapes, seed=11)
    np.savez(tmp_path / "deep_sort_yolov4" / "mars-small128.npz", **sd)
    got, seeded = parser._params("deep_sort_yolov4/mars-small128.pb", shapes, seed=5)
    assert not seeded and all(np.array_equal(got[k], sd[k]) for k in sd)


def test_npz_only_yolo_checkpoint_is_not_reseeded(tmp_path, monkeypatch):
    """only the converted yolo4.npz is installed (no .h5): it is a REAL checkpoint -- `_models` must not run seed_person_head
    on it (that rewrites the objectness / class-0 biases of the heads and would silence a trained detector)"""
    from posepipeline_amd.wrappers.deep_sort_yolov4 import parser
    monkeypatch.setenv("PIPELINE_3RDPARTY", str(tmp_path))
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")            # even with the synthetic switch on, an installed file wins
    (tmp_path / "deep_sort_yolov4").mkdir()
    yshapes = yolov4.yolov4_param_shapes()
    ysd = yolov4.synth_params(yshapes, seed=21)
    np.savez(tmp_path / "deep_sort_yolov4" / "yolo4.npz", **ysd)
    got, seeded = parser._params("deep_sort_yolov4/yolo4.h5", yshapes, seed=4)
    assert not seeded and all(np.array_equal(got[k], ysd[k]) for k in ysd)
    # what _models does with the flag, without a GPU: the seeded branch is the only one that touches the head
    called = []
    monkeypatch.setattr(yolov4, "seed_person_head", lambda sd: called.append(1))

    class _Stop(Exception):
        pass

    def no_gpu(*a, **k):
        raise _Stop()
    monkeypatch.setattr(parser._lib, "Context", no_gpu)
    parser._cache.clear()
    with pytest.raises(_Stop):
        parser._models(64, 64)
    assert not called
    (tmp_path / "deep_sort_yolov4" / "yolo4.npz").unlink()
    with pytest.raises(_Stop):
        parser._models(64, 64)
    assert called == [1]                                             # no file at all: seeded weights, head seeded
