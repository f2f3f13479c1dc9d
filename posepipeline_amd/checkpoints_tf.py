"""Readers for the two TensorFlow-era checkpoints of tracking_method 0 (`DeepSortYOLOv4`, the reference recipes' default):

  deep_sort_yolov4/mars-small128.pb   frozen TensorFlow GraphDef written by tools/freeze_model.py:250-275 -- the weights are
                                      Const nodes named after the slim variables (`conv1_1/weights`,
                                      `conv1_1/conv1_1/bn/beta`, `conv2_1/2/biases`, `fc1/weights`, `ball/moving_mean` ...);
                                      loaded by the reference at wrappers/deep_sort_yolov4/parser.py:41-42
  deep_sort_yolov4/yolo4.h5           Keras model file (`load_model`, wrappers/deep_sort_yolov4/yolo.py:51-54): one HDF5
                                      group per layer with kernel / bias / gamma / beta / moving_mean / moving_variance

Neither TensorFlow nor h5py is needed at run time:
  * a GraphDef is a protobuf message; `read_graphdef_consts` decodes the wire format directly (GraphDef.node = 1; NodeDef
    name = 1, op = 2, attr = 5 (map entry key = 1, value = 2); AttrValue.tensor = 8; TensorProto dtype = 1, tensor_shape = 2,
    tensor_content = 4, float_val = 5; TensorShapeProto.dim = 2, Dim.size = 1) and returns the float32 Const tensors;
  * the HDF5 file is converted ONCE, on any machine with h5py, by tools/convert_yolo4_h5.py into a flat `.npz` next to it,
    which `yolo_params` loads (h5py is used directly when it happens to be importable).
Both map to the parameter names of models/mars.py and models/yolov4.py.  No checkpoint file exists in the build or GPU
images: the mappings are exercised on synthetic files written in the same formats (tests/test_checkpoints_tf.py); against
the real files they are unverified.
"""
from __future__ import annotations

import os
import re
import struct

import numpy as np

DT_FLOAT = 1


# ---- protobuf wire format ---------------------------------------------------------------------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """yield (field number, wire type, value) of one message; length-delimited values are memoryview slices"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield field, wt, val


def _tensor(buf):
    dtype, dims, content, floats = 0, [], None, []
    for f, wt, v in _fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:                                   # TensorShapeProto
            for f2, _, v2 in _fields(v):
                if f2 == 2:                            # Dim
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3 if v3 < (1 << 63) else v3 - (1 << 64)
                    dims.append(size)
        elif f == 4:
            content = bytes(v)
        elif f == 5:                                   # float_val: packed or repeated fixed32
            floats += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", bytes(v))[0]]
    if dtype != DT_FLOAT:
        return None
    n = int(np.prod(dims)) if dims else 1
    if content is not None:
        a = np.frombuffer(content, "<f4")
    elif len(floats) == 1:
        a = np.full(n, floats[0], np.float32)          # TensorProto's "fill with the single value" convention
    else:
        a = np.asarray(floats, np.float32)
    return a.reshape(dims).astype(np.float32) if a.size == n else None


def read_graphdef_consts(path_or_bytes) -> dict:
    """{node name: float32 array} of every float Const node of a frozen GraphDef"""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    out = {}
    for f, _, node in _fields(memoryview(data)):
        if f != 1:
            continue
        name, op, value = None, None, None
        for f2, _, v2 in _fields(node):
            if f2 == 1:
                name = bytes(v2).decode()
            elif f2 == 2:
                op = bytes(v2).decode()
            elif f2 == 5:                              # attr map entry
                key, val = None, None
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        key = bytes(v3).decode()
                    elif f3 == 2:
                        val = v3
                if key == "value" and val is not None:
                    for f4, _, v4 in _fields(val):
                        if f4 == 8:
                            value = _tensor(v4)
        if op == "Const" and name and value is not None:
            out[name] = value
    return out


# ---- mars-small128 ----------------------------------------------------------------------------------------------------
_TF_LEAF = {"weights": "weight", "biases": "bias", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}


def _slim_name(tf_name: str):
    """`net/conv2_1/1/conv2_1/1/bn/moving_mean` -> `conv2_1.1.bn.mean` (slim repeats the enclosing scope for the batch norm
    created inside conv2d / fully_connected; an import prefix may precede everything)"""
    tok = tf_name.split("/")
    if tok[-1] not in _TF_LEAF:
        return None
    tok[-1] = _TF_LEAF[tok[-1]]
    for k in range(len(tok) // 2, 0, -1):              # drop one copy of a scope that appears twice in a row (longest first)
        hit = next((st for st in range(len(tok) - 2 * k + 1) if tok[st:st + k] == tok[st + k:st + 2 * k]), None)
        if hit is not None:
            tok = tok[:hit] + tok[hit + k:]
            break
    return ".".join(tok)


def mars_params_from_consts(consts: dict, shapes: dict) -> dict:
    """GraphDef constants -> the parameters of models/mars.py (conv HWIO -> OIHW, fc1 [in][out] -> [out][in])"""
    by_name = {}
    for name, arr in consts.items():
        key = _slim_name(name)
        if key is None:
            continue
        for want in shapes:                            # tolerate a graph prefix: match on the suffix
            if key == want or key.endswith("." + want):
                by_name[want] = arr
    sd = {}
    for want, shp in shapes.items():
        if want not in by_name:
            raise KeyError(f"mars-small128 graph has no constant for {want}")
        a = by_name[want]
        if want.endswith(".weight"):
            a = a.T if a.ndim == 2 else np.transpose(a, (3, 2, 0, 1))
        a = np.ascontiguousarray(a, np.float32)
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f"{want}: shape {a.shape} in the graph, expected {shp}")
        sd[want] = a
    return sd


def mars_params(path: str, shapes: dict) -> dict:
    if path.endswith(".npz"):
        z = np.load(path)
        return {k: np.asarray(z[k], np.float32) for k in shapes}
    return mars_params_from_consts(read_graphdef_consts(path), shapes)


# ---- YOLOv4 (Keras) ------------------------------------------------------------------------------------------------------
def _suffix(layer_name: str) -> int:
    m = re.search(r"_(\d+)$", layer_name)
    return int(m.group(1)) if m else 0                 # 'conv2d' (TF2 Keras' first layer) sorts before 'conv2d_1'


def yolo_params_from_keras(layers: dict, shapes: dict) -> dict:
    """layers: {keras layer name: {weight name: array}} as stored in the HDF5 file (weight names like
    'conv2d_7/kernel:0', 'batch_normalization_7/moving_mean:0').  Keras numbers its conv2d_* / batch_normalization_* layers in
    creation order, which is the order models/yolov4._layers lists them (yolo4/model.py:100-190), so the i-th convolution
    is l{i} and the j-th batch norm belongs to the j-th convolution that has one."""
    def leaf(d, *names):
        for k, v in d.items():
            if k.split("/")[-1].split(":")[0] in names:
                return np.asarray(v, np.float32)
        raise KeyError(f"none of {names} in {sorted(d)}")

    convs = sorted((n for n in layers if n.startswith("conv2d") and layers[n]), key=_suffix)
    bns = sorted((n for n in layers if n.startswith("batch_normalization") and layers[n]), key=_suffix)
    n_conv = sum(1 for k in shapes if k.endswith(".weight"))
    n_bn = sum(1 for k in shapes if k.endswith(".bn.gamma"))
    if len(convs) != n_conv or len(bns) != n_bn:
        raise ValueError(f"checkpoint has {len(convs)} convolutions / {len(bns)} batch norms, the program {n_conv} / {n_bn}")
    sd, j = {}, 0
    for i, cname in enumerate(convs):
        sd[f"l{i}.weight"] = np.ascontiguousarray(np.transpose(leaf(layers[cname], "kernel"), (3, 2, 0, 1)))    # HWIO -> OIHW
        if f"l{i}.bias" in shapes:
            sd[f"l{i}.bias"] = leaf(layers[cname], "bias")
        else:
            b = layers[bns[j]]
            j += 1
            sd[f"l{i}.bn.gamma"], sd[f"l{i}.bn.beta"] = leaf(b, "gamma"), leaf(b, "beta")
            sd[f"l{i}.bn.mean"], sd[f"l{i}.bn.var"] = leaf(b, "moving_mean"), leaf(b, "moving_variance")
    for k, shp in shapes.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {sd[k].shape} in the checkpoint, expected {shp}")
    return sd


def read_keras_h5(path: str) -> dict:
    """{layer: {weight name: array}} of a Keras .h5 (model or weights file).  Needs h5py."""
    import h5py
    out = {}
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                out.setdefault(name.split("/")[0], {})["/".join(name.split("/")[1:])] = np.asarray(obj)
        root.visititems(visit)
    return out


def yolo_params(path_h5: str, shapes: dict) -> dict:
    """yolo4.h5 -> parameters of models/yolov4.py: the converted `<path>.npz` when it exists (tools/convert_yolo4_h5.py), else
    through h5py when that is importable."""
    npz = os.path.splitext(path_h5)[0] + ".npz"
    if os.path.exists(npz):
        z = np.load(npz)
        missing = [k for k in shapes if k not in z.files]
        if missing:
            raise KeyError(f"{npz}: missing {missing[:4]}")
        return {k: np.asarray(z[k], np.float32) for k in shapes}
    try:
        layers = read_keras_h5(path_h5)
    except ImportError as e:
        raise FileNotFoundError(f"{npz} not found and h5py is not installed: run `python tools/convert_yolo4_h5.py {path_h5}` "
                                "on a machine with h5py and copy the .npz next to the .h5") from e
    return yolo_params_from_keras(layers, shapes)
