"""DataJoint `longblob` (de)serialisation for the hot-path tables.

`TrackingBbox.tracks` (list of per-frame lists of dicts holding ints, floats and small float arrays),
`PersonBbox.bbox / present`, `TopDownPerson.keypoints`, `LiftingPerson.keypoints_3d / keypoints_valid`
(pose_pipeline/pipeline.py:508-513, 649-654, 1011-1015, 1252-1257) are `longblob` attributes: DataJoint stores
`datajoint.blob.pack(value)` in MySQL and returns `unpack(bytes)` on fetch.  DataJoint is a third-party dependency that
is NOT installed here (requirements.txt:1 `datajoint`, unpinned), so this is a restatement of its published `dj0` /
`mYm` wire format (datajoint-python `blob.py`, 0.13 / 0.14) and is PARITY UNPINNED -- the byte layouts asserted in
tests/test_blob.py are this file's reading of that format, not vectors produced by DataJoint.  With the real package
(POSEPIPE_USE_DATAJOINT=1) its own codec is used and this module is idle; on the in-memory shim it gives the tables
DataJoint's value semantics (a fetch returns a fresh copy, inserts reject values the database could not hold) and it
is the codec a non-Python host would use to read or write the same tables.

Format (little endian):
    [ "ZL123\\0" uint64(len) zlib(...) ]  optional compression wrapper, used when it shrinks a blob > 1000 bytes
    "mYm\\0" | "dj0\\0"                   protocol: dj0 as soon as a native Python type (list, dict, int, ...) occurs
    value:
      "A" uint64 ndim, uint64 shape[ndim], uint32 type_id, uint32 is_complex, data in Fortran order  numpy array / scalar
      "\\x02" list, "\\x01" tuple, "\\x03" set : uint64 n, then n x (uint64 len, value)
      "\\x04" dict                          : uint64 n, then n x (uint64 len, key, uint64 len, value)
      "\\x05" str (uint64 len, utf-8), "\\x06" bytes (uint64 len, raw)
      "\\x0a" int (uint16 nbytes, signed little endian), "\\x0b" bool (1 byte), "\\x0d" float (float64),
      "\\x0c" complex (2 x float64), "\\xff" None,
      "t" date / time / datetime (int32 yyyymmdd or -1, int64 hhmmss * 10^6 + microseconds or -1),
      "d" Decimal (uint64 len, text), "u" UUID (16 bytes)
"""
from __future__ import annotations

import datetime
import decimal
import uuid
import zlib
from collections.abc import Mapping, MutableSequence, Sequence, Set

import numpy as np

# mYm / MATLAB class ids
_DTYPES = {3: np.dtype("bool"), 4: np.dtype("c"), 5: np.dtype("O"), 6: np.dtype("float64"), 7: np.dtype("float32"),
           8: np.dtype("int8"), 9: np.dtype("uint8"), 10: np.dtype("int16"), 11: np.dtype("uint16"),
           12: np.dtype("int32"), 13: np.dtype("uint32"), 14: np.dtype("int64"), 15: np.dtype("uint64")}
_TYPE_ID = {v: k for k, v in _DTYPES.items()}
_COMPRESSED = b"ZL123\0"


class BlobError(ValueError):
    pass


def _u64(n) -> bytes:
    return np.uint64(n).tobytes()


class _Packer:
    def __init__(self):
        self.protocol = b"mYm\0"

    def pack(self, obj) -> bytes:
        if isinstance(obj, (np.ndarray, np.number, np.bool_)):
            return self.array(np.asarray(obj))
        self.protocol = b"dj0\0"
        if obj is None:
            return b"\xff"
        if isinstance(obj, bool):
            return b"\x0b" + np.array(obj, dtype="bool").tobytes()
        if isinstance(obj, int):
            n = obj.bit_length() // 8 + 1
            if n > 0xFFFF:
                raise BlobError("integer too large for a blob")
            return b"\x0a" + np.uint16(n).tobytes() + obj.to_bytes(n, "little", signed=True)
        if isinstance(obj, float):
            return b"\x0d" + np.array(obj, dtype="float64").tobytes()
        if isinstance(obj, complex):
            return b"\x0c" + np.array([obj.real, obj.imag], dtype="float64").tobytes()
        if isinstance(obj, str):
            b = obj.encode()
            return b"\x05" + _u64(len(b)) + b
        if isinstance(obj, (bytes, bytearray)):
            return b"\x06" + _u64(len(obj)) + bytes(obj)
        if isinstance(obj, (datetime.datetime, datetime.date, datetime.time)):     # VideoInfo.timestamps
            if isinstance(obj, datetime.datetime):
                d, t = obj.date(), obj.time()
            elif isinstance(obj, datetime.date):
                d, t = obj, None
            else:
                d, t = None, obj
            di = -1 if d is None else (d.year * 100 + d.month) * 100 + d.day
            ti = -1 if t is None else ((t.hour * 100 + t.minute) * 100 + t.second) * 1_000_000 + t.microsecond
            return b"t" + np.int32(di).tobytes() + np.int64(ti).tobytes()
        if isinstance(obj, decimal.Decimal):
            s = str(obj).encode()
            return b"d" + _u64(len(s)) + s
        if isinstance(obj, uuid.UUID):
            return b"u" + obj.bytes
        if isinstance(obj, Mapping):
            return b"\x04" + _u64(len(obj)) + b"".join(self._sized(k) + self._sized(v) for k, v in obj.items())
        if isinstance(obj, MutableSequence):
            return b"\x02" + _u64(len(obj)) + b"".join(self._sized(v) for v in obj)
        if isinstance(obj, Sequence):
            return b"\x01" + _u64(len(obj)) + b"".join(self._sized(v) for v in obj)
        if isinstance(obj, Set):
            return b"\x03" + _u64(len(obj)) + b"".join(self._sized(v) for v in obj)
        raise BlobError(f"cannot serialise {type(obj).__name__} into a DataJoint blob")

    def _sized(self, obj) -> bytes:
        b = self.pack(obj)
        return _u64(len(b)) + b

    def array(self, a: np.ndarray) -> bytes:
        head = b"A" + np.array((a.ndim,) + a.shape, dtype=np.uint64).tobytes()
        is_complex = np.iscomplexobj(a)
        imag = None
        if is_complex:
            a, imag = np.real(a), np.imag(a)
        dt = np.dtype("O") if a.dtype.char == "U" else a.dtype
        if dt not in _TYPE_ID:
            raise BlobError(f"array dtype {a.dtype} has no blob type id")
        tid = _TYPE_ID[dt]
        head += np.array([tid, is_complex], dtype=np.uint32).tobytes()
        if tid == 5:                                   # object arrays: every element its own sized blob
            self.protocol = b"dj0\0"
            return head + b"".join(self._sized(v) for v in a.flatten(order="F"))
        if tid == 4:                                   # chars travel as 16-bit code units
            return head + a.view(np.uint8).astype(np.uint16).tobytes()
        body = a.tobytes(order="F")
        if is_complex:
            body += imag.tobytes(order="F")
        return head + body


def pack(obj, compress: bool = True) -> bytes:
    p = _Packer()
    body = p.pack(obj)                                 # may switch the protocol; evaluate it afterwards
    blob = p.protocol + body
    if compress and len(blob) > 1000:
        z = _COMPRESSED + _u64(len(blob)) + zlib.compress(blob)
        if len(z) < len(blob):
            blob = z
    return blob


class _Reader:
    def __init__(self, blob: bytes):
        self.b = blob
        self.pos = 0

    def take(self, dtype="uint64", count=1):
        dt = np.dtype(dtype)
        n = dt.itemsize * count
        if self.pos + n > len(self.b):
            raise BlobError("truncated blob")
        v = np.frombuffer(self.b, dtype=dt, count=count, offset=self.pos)
        self.pos += n
        return v[0] if count == 1 else v

    def raw(self, n: int) -> bytes:
        if self.pos + n > len(self.b):
            raise BlobError("truncated blob")
        v = self.b[self.pos:self.pos + n]
        self.pos += n
        return v

    def sized(self):
        n = int(self.take())
        end = self.pos + n
        v = self.value()
        if self.pos != end:
            raise BlobError("element length mismatch")
        return v

    def value(self):
        code = self.raw(1)
        if code == b"A":
            ndim = int(self.take())
            shape = tuple(int(s) for s in np.atleast_1d(self.take(count=ndim))) if ndim else ()
            n = int(np.prod(shape, dtype=np.int64)) if ndim else 1
            tid, is_complex = (int(v) for v in self.take("uint32", 2))
            if tid not in _DTYPES:
                raise BlobError(f"unknown array type id {tid}")
            if tid == 5:
                data = np.empty(n, dtype=object)
                for i in range(n):
                    data[i] = self.sized()
            elif tid == 4:
                data = np.atleast_1d(self.take("uint16", n)).astype(np.uint8).view("c")
            else:
                data = np.atleast_1d(self.take(_DTYPES[tid], n)).copy()
                if is_complex:
                    data = data + 1j * np.atleast_1d(self.take(_DTYPES[tid], n))
            return data.reshape(shape, order="F")
        if code == b"\xff":
            return None
        if code == b"\x0b":
            return bool(self.take("bool"))
        if code == b"\x0a":
            n = int(self.take("uint16"))
            return int.from_bytes(self.raw(n), "little", signed=True)
        if code == b"\x0d":
            return float(self.take("float64"))
        if code == b"\x0c":
            re_, im = self.take("float64", 2)
            return complex(re_, im)
        if code == b"\x05":
            return self.raw(int(self.take())).decode()
        if code == b"\x06":
            return self.raw(int(self.take()))
        if code in (b"\x01", b"\x02", b"\x03"):
            items = [self.sized() for _ in range(int(self.take()))]
            return tuple(items) if code == b"\x01" else items if code == b"\x02" else set(items)
        if code == b"\x04":
            out = {}
            for _ in range(int(self.take())):
                k = self.sized()
                out[k] = self.sized()
            return out
        if code == b"t":
            di, ti = int(self.take("int32")), int(self.take("int64"))
            d = datetime.date(di // 10000, (di // 100) % 100, di % 100) if di >= 0 else None
            t = datetime.time((ti // 10_000_000_000) % 100, (ti // 100_000_000) % 100, (ti // 1_000_000) % 100,
                              ti % 1_000_000) if ti >= 0 else None
            return datetime.datetime.combine(d, t) if d is not None and t is not None else (t if d is None else d)
        if code == b"d":
            return decimal.Decimal(self.raw(int(self.take())).decode())
        if code == b"u":
            return uuid.UUID(bytes=self.raw(16))
        raise BlobError(f"unsupported blob element code {code!r}")


def unpack(blob: bytes):
    blob = bytes(blob)
    if blob.startswith(_COMPRESSED):
        r = _Reader(blob)
        r.pos = len(_COMPRESSED)
        size = int(r.take())
        blob = zlib.decompress(blob[r.pos:])
        if len(blob) != size:
            raise BlobError("compressed blob size mismatch")
    if blob[:4] not in (b"mYm\0", b"dj0\0"):
        raise BlobError("not a DataJoint blob")
    r = _Reader(blob)
    r.pos = 4
    v = r.value()
    if r.pos != len(blob):
        raise BlobError("trailing bytes in blob")
    return v
