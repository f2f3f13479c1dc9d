"""A small in-memory stand-in for the DataJoint surface the hot-path tables use.

The reference's L3 layer is DataJoint on MySQL (`dj.schema`, `dj.Manual / Lookup / Computed`,
`Table & restriction`, `fetch1`, `insert1`, `populate`).  Neither is available here, and the hot path
does not need a database: this shim keeps rows in Python lists so that the table classes in
posepipeline_amd/pipeline.py -- same names, same `definition` strings, same `make()` dispatch as
pose_pipeline/pipeline.py -- run unchanged, and real DataJoint stays pluggable (set
POSEPIPE_USE_DATAJOINT=1 to import the real package instead).

Supported: dependency-free primary keys parsed from `definition` (lines above `---`, `-> Parent`
expanded recursively), `Table & dict`, `Table & other_table`, `Table - other`, `len`, `fetch1(*attrs)`,
`fetch("KEY")`, `fetch(attr...)`, `insert1(row, skip_duplicates=)`, `insert(rows)`, `delete()`,
Lookup `contents`, Computed `populate(*restrictions)` over `key_source` (default: join of parents).
"""
from __future__ import annotations

import os
import re

from . import blob as _blob

config = {"custom": {}}

_REGISTRY: dict = {}


class DuplicateError(Exception):
    pass


def _parse_definition(cls):
    """-> (primary attribute names, parent tables, secondary attribute names); sets cls.blob_attrs"""
    pk, parents, sec = [], [], []
    cls.blob_attrs = set()
    section = pk
    for raw in cls.definition.splitlines():
        line = raw.split("#")[0].strip()
        if not line:
            continue
        if line.startswith("---"):
            section = sec
            continue
        if line.startswith("->"):
            name = line[2:].strip()
            parent = _REGISTRY[name]
            parents.append(parent) if section is pk else None
            for a in parent.primary_key:
                if a not in section and a not in pk:
                    section.append(a)
            continue
        m = re.match(r"([A-Za-z_][A-Za-z0-9_]*)\s*(=[^:]*)?:\s*([A-Za-z]+)?", line)
        if m:
            section.append(m.group(1))
            if (m.group(3) or "").lower() in ("longblob", "blob", "mediumblob", "tinyblob"):
                cls.blob_attrs.add(m.group(1))
    return pk, parents, sec


class _Packed(bytes):
    """a serialised longblob value in a row store"""


BLOBS = os.environ.get("POSEPIPE_SHIM_BLOBS", "1") != "0"     # longblob attributes are stored as DataJoint blobs


class _Relation:
    """A restricted view: rows of `table` that agree with every restriction."""

    def __init__(self, table, restrictions=()):
        self.table = table
        self.restrictions = tuple(restrictions)

    # -- row iteration --------------------------------------------------------------------------
    def _match(self, row):
        for kind, r in self.restrictions:
            if kind == "dict":
                if any(k in row and row[k] != v for k, v in r.items()):
                    return False
            elif kind == "rel":
                keys = r._rows()
                common = [a for a in self.table.heading if keys and a in keys[0]]
                if not any(all(row[a] == k[a] for a in common) for k in keys):
                    return False
            elif kind == "not":
                keys = r._rows()
                common = [a for a in self.table.heading if keys and a in keys[0]]
                if common and any(all(row[a] == k[a] for a in common) for k in keys):
                    return False
            elif kind == "sql":
                ms = re.match(r"""\s*(\w+)\s*=\s*(["'])(.*)\2\s*$""", r)     # attr="text"
                if ms:
                    if row[ms.group(1)] != ms.group(3):
                        return False
                    continue
                m = re.match(r"\s*(\w+)\s*(>=|<=|=|>|<)\s*(-?\d+(?:\.\d+)?)\s*$", r)
                if not m:
                    raise NotImplementedError(f"restriction string {r!r}")
                a, op, v = m.group(1), m.group(2), float(m.group(3))
                x = row[a]
                if not {">=": x >= v, "<=": x <= v, "=": x == v, ">": x > v, "<": x < v}[op]:
                    return False
        return True

    def _rows(self):
        return [r for r in self.table._store if self._match(r)]

    # -- relational operators ---------------------------------------------------------------------
    def __and__(self, other):
        return _Relation(self.table, self.restrictions + (_as_restriction(other),))

    def __sub__(self, other):
        return _Relation(self.table, self.restrictions + (("not", _as_relation(other)),))

    def __len__(self):
        return len(self._rows())

    def __bool__(self):
        return len(self) > 0

    # -- fetch --------------------------------------------------------------------------------------
    def _rows_out(self):
        """rows as a fetch returns them: blob attributes deserialised (a fresh object per fetch, like DataJoint)"""
        return [{k: (_blob.unpack(v) if isinstance(v, _Packed) else v) for k, v in r.items()} for r in self._rows()]

    def fetch1(self, *attrs):
        rows = self._rows_out()
        if len(rows) != 1:
            raise ValueError(f"fetch1 on {self.table.__name__}: expected one row, found {len(rows)}")
        if not attrs:
            return dict(rows[0])
        if attrs == ("KEY",):
            return {a: rows[0][a] for a in self.table.primary_key}
        vals = [rows[0][a] for a in attrs]
        return vals[0] if len(vals) == 1 else tuple(vals)

    def fetch(self, *attrs, as_dict=False):
        rows = self._rows_out()
        if attrs == ("KEY",):
            return [{a: r[a] for a in self.table.primary_key} for r in rows]
        pk = self.table.primary_key
        if not attrs or as_dict:
            # "KEY" among the attributes of an as_dict fetch: the primary-key attributes are merged into each row's dict
            names = [b for a in attrs for b in (pk if a == "KEY" else (a,))]
            return [dict(r) if not attrs else {a: r[a] for a in names} for r in rows]
        cols = [[({b: r[b] for b in pk} if a == "KEY" else r[a]) for r in rows] for a in attrs]
        return cols[0] if len(cols) == 1 else tuple(cols)

    def delete(self):
        doomed = [id(r) for r in self._rows()]
        self.table._store[:] = [r for r in self.table._store if id(r) not in doomed]

    def proj(self):
        return self


def _as_relation(x):
    if isinstance(x, _Relation):
        return x
    if isinstance(x, type) and issubclass(x, Table):
        return _Relation(x)
    if isinstance(x, Table):
        return _Relation(type(x))
    raise TypeError(type(x))


def _as_restriction(x):
    if isinstance(x, dict):
        return ("dict", x)
    if isinstance(x, str):
        return ("sql", x)
    return ("rel", _as_relation(x))


class _TableMeta(type):
    def __and__(cls, other):
        return _Relation(cls) & other

    def __sub__(cls, other):
        return _Relation(cls) - other

    def __len__(cls):
        return len(cls._store)


class _hybrid:
    """method usable on the table class as well as on an instance (DataJoint code calls both
    `TopDownPerson.populate(key)` and `TopDownPerson().populate(key)`)"""

    def __init__(self, f):
        self.f = f
        self.__doc__ = f.__doc__

    def __get__(self, obj, cls):
        import functools
        return functools.partial(self.f, obj if obj is not None else cls())


class Table(metaclass=_TableMeta):
    definition = ""
    primary_key: list = []
    heading: list = []
    parents: list = []
    _store: list = []

    # instances behave like the unrestricted relation (DataJoint style: `Table()`)
    def __and__(self, other):
        return _Relation(type(self)) & other

    def __sub__(self, other):
        return _Relation(type(self)) - other

    def __len__(self):
        return len(type(self)._store)

    @_hybrid
    def fetch1(self, *attrs):
        return _Relation(type(self)).fetch1(*attrs)

    @_hybrid
    def fetch(self, *attrs, **kw):
        return _Relation(type(self)).fetch(*attrs, **kw)

    @classmethod
    def _insert_row(cls, row, skip_duplicates=False):
        row = dict(row)
        missing = [a for a in cls.primary_key if a not in row]
        if missing:
            raise KeyError(f"{cls.__name__}.insert1: missing primary key attribute(s) {missing}")
        row = {a: row[a] for a in cls.heading if a in row}
        for a in getattr(cls, "blob_attrs", ()):
            if a in row and BLOBS:
                row[a] = _Packed(_blob.pack(row[a]))      # what DataJoint would send to MySQL (blob.py)
        for r in cls._store:
            if all(r[a] == row[a] for a in cls.primary_key):
                if skip_duplicates:
                    return
                raise DuplicateError(f"duplicate entry in {cls.__name__}: { {a: row[a] for a in cls.primary_key} }")
        cls._store.append(row)

    @_hybrid
    def insert1(self, row, skip_duplicates=False, **kw):
        type(self)._insert_row(row, skip_duplicates)

    @_hybrid
    def insert(self, rows, skip_duplicates=False, **kw):
        for r in rows:
            type(self)._insert_row(r, skip_duplicates)

    @_hybrid
    def delete(self):
        type(self)._store.clear()


class Manual(Table):
    pass


class Lookup(Table):
    contents: list = []


class Computed(Table):
    @property
    def key_source(self):
        return None

    def make(self, key):
        raise NotImplementedError

    @_hybrid
    def populate(self, *restrictions, reserve_jobs=False, suppress_errors=False, display_progress=False, **kw):
        cls = type(self)
        src = self.key_source
        if src is None:
            # default key source: join of the primary parents' keys
            keys = [{}]
            for p in cls.parents:
                new = []
                for k in keys:
                    for r in p._store:
                        pk = {a: r[a] for a in p.primary_key}
                        if all(k.get(a, pk[a]) == pk[a] for a in pk):
                            new.append({**k, **pk})
                keys = new
        else:
            rel = _as_relation(src)
            keys = [{a: r[a] for a in cls.primary_key if a in r} for r in rel._rows()]
        done = 0
        errors = []
        for key in keys:
            if any(not _Relation(cls, (_as_restriction(r),))._match_key(key) for r in restrictions):
                continue
            # done already?  A key_source may be coarser than the table's primary key (BestDetectedFrames: Video & DetectedFrames,
            # pipeline.py:783-785): like DataJoint's `key_source - target`, compare on the attributes the key has
            if any(all(r[a] == key[a] for a in key) for r in cls._store):
                continue
            try:
                self.make(dict(key))
                done += 1
            except Exception as e:  # noqa: BLE001 -- DataJoint's suppress_errors contract
                if not suppress_errors:
                    raise
                errors.append((key, e))
        return errors if suppress_errors else None


def _match_key(self, key):
    """does a (partial) primary key agree with the restrictions?  Attributes the key lacks are free."""
    for kind, r in self.restrictions:
        if kind == "dict":
            if any(k in key and key[k] != v for k, v in r.items()):
                return False
        elif kind == "rel":
            rows = r._rows()
            common = [a for a in key if rows and a in rows[0]]
            if not any(all(key[a] == row[a] for a in common) for row in rows):
                return False
        else:
            raise NotImplementedError(f"populate restriction of kind {kind}")
    return True


_Relation._match_key = _match_key


def schema(name=None, **kw):
    """`@schema` class decorator: parse the definition, give the class its own row store."""
    def deco(cls):
        _REGISTRY[cls.__name__] = cls
        cls._store = []
        pk, parents, sec = _parse_definition(cls)
        cls.primary_key = pk
        cls.parents = parents
        cls.heading = pk + [a for a in sec if a not in pk]
        if issubclass(cls, Lookup):
            for row in cls.contents:
                cls._insert_row(row, skip_duplicates=True)
        return cls
    return deco


Schema = schema


def reset():
    """Empty every non-lookup table (tests)."""
    for cls in _REGISTRY.values():
        if not issubclass(cls, Lookup):
            cls._store.clear()
