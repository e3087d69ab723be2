"""CPU oracle — TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

A restatement, on host numpy arrays, of the reference's physical plan
(/root/reference/query-engine/src/physical_plan/{mod.rs,physical_expressions.rs},
columnar_value.rs, error.rs): the same struct names, the same four-way
Array/Scalar dispatch per expression node (one materialised array per node), the
same operators as pull iterators.  The arithmetic the reference delegates to
arrow2 0.10.1 (not vendored, not buildable here) is restated in qe_oracle.c.

PARITY STATUS: pinned for filter ordering / all-column compaction / Utf8 == scalar
(tests/mod.rs:138-188) and column pass-through (tests/mod.rs:86-136) by
tests/golden/userdata_*.  Everything else (arithmetic, numeric comparisons,
grouping, min/max, and the SUM/COUNT/AVG/Int64/ordered-comparison/boolean
extensions, which the reference does not implement) is "parity unpinned".
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqe_oracle.so")

INT32, INT64, FLOAT64, BOOLEAN, UTF8 = 1, 2, 3, 4, 5
OP_EQ, OP_NEQ, OP_GT, OP_GTEQ, OP_LT, OP_LTEQ, OP_AND, OP_OR, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD = range(1, 14)
AGG_MAX, AGG_MIN, AGG_SUM, AGG_COUNT, AGG_AVG = 1, 2, 3, 4, 5
_NP = {INT32: np.int32, INT64: np.int64, FLOAT64: np.float64}
_TYPE_NAME = {INT32: "Int32", INT64: "Int64", FLOAT64: "Float64", BOOLEAN: "Boolean", UTF8: "Utf8"}


def build(force=False):
    src = os.path.join(_HERE, "qe_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-shared", "-std=c11",
                               "-o", _LIB, src, "-lm"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_filter_fixed.restype = C.c_int64
        _lib.orc_filter_bitmap.restype = C.c_int64
        _lib.orc_filter_utf8.restype = C.c_int64
        _lib.orc_filter_count.restype = C.c_int64
        _lib.orc_bitmap_count.restype = C.c_int64
        _lib.orc_aggregate_reference_batch.restype = C.c_int64
        _lib.orc_hash_aggregate.restype = C.c_int64
        _lib.orc_ord_inverse_f64.restype = C.c_double
        _lib.orc_ord_inverse_f64.argtypes = [C.c_uint64]
    return _lib


class OracleError(Exception):
    """Mirrors `enum Error` (error.rs:3-35); `.name` is the variant name."""

    def __init__(self, name, msg=""):
        self.name = name
        super().__init__(f"{name}: {msg}")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack(mask):
    """bool array -> LSB-first bitmap, padded."""
    b = np.packbits(np.asarray(mask, dtype=np.bool_), bitorder="little")
    return np.concatenate([b, np.zeros(8, np.uint8)])


def unpack(bits, n):
    return np.unpackbits(bits, bitorder="little")[:n].astype(np.bool_)


class Array:
    """values: numpy array (int32/int64/float64/bool) or (offsets int32[n+1], bytes uint8[]) for utf8;
    validity: bool array or None."""

    def __init__(self, type_, values, validity=None):
        self.type = type_
        self.values = values
        self.validity = None if validity is None else np.asarray(validity, dtype=np.bool_)

    def __len__(self):
        return len(self.values[0]) - 1 if self.type == UTF8 else len(self.values)

    @staticmethod
    def from_python(x, validity=None):
        if isinstance(x, Array):
            return x
        if isinstance(x, tuple):
            x, validity = x
        if isinstance(x, list) and (not x or isinstance(x[0], (str, bytes, type(None)))):
            enc = [b"" if s is None else (s.encode() if isinstance(s, str) else s) for s in x]
            val = np.array([s is not None for s in x], dtype=np.bool_)
            offs = np.zeros(len(x) + 1, np.int32)
            np.cumsum([len(s) for s in enc], out=offs[1:])
            data = np.frombuffer(b"".join(enc), dtype=np.uint8).copy()
            if validity is not None:
                val &= np.asarray(validity, dtype=np.bool_)
            return Array(UTF8, (offs, data), None if val.all() else val)
        a = np.ascontiguousarray(x)
        t = {np.dtype(np.int32): INT32, np.dtype(np.int64): INT64, np.dtype(np.float64): FLOAT64,
             np.dtype(np.bool_): BOOLEAN}[a.dtype]
        return Array(t, a, validity)

    def strings(self):
        offs, data = self.values
        raw = data.tobytes()
        return [raw[offs[i]:offs[i + 1]] for i in range(len(self))]

    def to_pylist(self):
        n = len(self)
        vals = self.strings() if self.type == UTF8 else self.values
        out = []
        for i in range(n):
            if self.validity is not None and not self.validity[i]:
                out.append(None)
            elif self.type == UTF8:
                out.append(vals[i].decode())
            else:
                out.append(vals[i].item())
        return out


class Scalar:
    def __init__(self, type_, value):
        self.type = type_
        self.value = value  # None = null scalar


class ColumnarValue:
    """enum ColumnarValue { Array, Scalar } (columnar_value.rs:12-15)."""

    def __init__(self, array=None, scalar=None):
        self.array = array
        self.scalar = scalar

    @property
    def is_scalar(self):
        return self.scalar is not None

    def to_array(self, length):
        # columnar_value.rs:18-23
        return self.array if self.array is not None else scalar_to_array(self.scalar, length)


def scalar_to_array(s, length):
    """columnar_value.rs:26-59: broadcast for Int32/Float64/Utf8/Boolean (and Int64 here); a null
    scalar is ScalarToArrayError."""
    if s.value is None:
        raise OracleError("ScalarToArrayError", "null scalar")
    if s.type in _NP:
        return Array(s.type, np.full(length, s.value, dtype=_NP[s.type]))
    if s.type == BOOLEAN:
        return Array(BOOLEAN, np.full(length, bool(s.value), dtype=np.bool_))
    if s.type == UTF8:
        return Array.from_python([s.value] * length)
    raise OracleError("ScalarToArrayError", str(s.type))


class Chunk:
    """arrow2 Chunk<Arc<dyn Array>>: equal-length arrays."""

    def __init__(self, arrays):
        self.arrays = [Array.from_python(a) for a in arrays]
        self.len = len(self.arrays[0]) if self.arrays else 0

    def __len__(self):
        return self.len


def _and_validity(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return a & b


# ------------------------------------------------------------- expressions
class PhysicalExpression:
    def evaluate(self, chunk):
        raise NotImplementedError


class ColumnExpression(PhysicalExpression):
    """physical_expressions.rs:26-62."""

    def __init__(self, index):
        self.index = index

    def evaluate(self, chunk):
        if self.index < 0 or self.index >= len(chunk.arrays):
            raise OracleError("PrimitiveTypeNotSuported", "Int32")
        return ColumnarValue(array=chunk.arrays[self.index])


class _Literal(PhysicalExpression):
    TYPE = 0

    def __init__(self, value):
        self.value = value

    def evaluate(self, chunk):
        return ColumnarValue(scalar=Scalar(self.TYPE, self.value))


class LiteralBoolExpression(_Literal):
    TYPE = BOOLEAN


class LiteralStringExpression(_Literal):
    TYPE = UTF8


class LiteralIntegerExpression(_Literal):
    TYPE = INT32


class LiteralInt64Expression(_Literal):
    TYPE = INT64


class LiteralFloatExpression(_Literal):
    TYPE = FLOAT64


def _wrap_int(v, t):
    bits = 32 if t == INT32 else 64
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def _cmp_arrays(op, a, b, b_scalar):
    """arrow2 compute::comparison::{eq,neq}[_scalar] (K1/K2) and the ordered extensions."""
    n = len(a)
    out = np.zeros((n + 7) // 8 + 8, np.uint8)
    L = lib()
    if a.type == UTF8:
        if op not in (OP_EQ, OP_NEQ):
            raise OracleError("PhysicalTypeNotSuported", "Utf8")
        L.orc_cmp_utf8(op, _p(a.values[0]), _p(a.values[1] if len(a.values[1]) else np.zeros(1, np.uint8)),
                       _p(b.values[0]), _p(b.values[1] if len(b.values[1]) else np.zeros(1, np.uint8)),
                       C.c_int64(n), int(b_scalar), _p(out))
    elif a.type == BOOLEAN:
        if op not in (OP_EQ, OP_NEQ):
            raise OracleError("PhysicalTypeNotSuported", "Boolean")
        bv = b.values[0] if b_scalar else b.values
        res = (a.values == bv) if op == OP_EQ else (a.values != bv)
        return Array(BOOLEAN, res, _and_validity(a.validity, None if b_scalar else b.validity))
    else:
        fn = {INT32: L.orc_cmp_i32, INT64: L.orc_cmp_i64, FLOAT64: L.orc_cmp_f64}[a.type]
        fn(op, _p(a.values), _p(b.values), C.c_int64(n), int(b_scalar), _p(out))
    return Array(BOOLEAN, unpack(out, n), _and_validity(a.validity, None if b_scalar else b.validity))


def _scalar_as_array1(s):
    if s.type == UTF8:
        return Array.from_python([s.value])
    if s.type == BOOLEAN:
        return Array(BOOLEAN, np.array([bool(s.value)]))
    return Array(s.type, np.array([s.value], dtype=_NP[s.type]))


class _Comparison(PhysicalExpression):
    """comparisonExpression! (physical_expressions.rs:170-233)."""
    OP = 0

    def __init__(self, left, right):
        self.left, self.right = left, right

    def evaluate(self, chunk):
        l, r = self.left.evaluate(chunk), self.right.evaluate(chunk)
        op = self.OP
        if not l.is_scalar and not r.is_scalar:
            a, b = l.array, r.array
            if len(a) != len(b):
                raise OracleError("DifferentSizes")
            if a.type != b.type:
                raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
            return ColumnarValue(array=_cmp_arrays(op, a, b, False))
        if not l.is_scalar or not r.is_scalar:
            # (Array, Scalar) and (Scalar, Array): the reference passes (array, scalar) in both arms
            # (:194-203), which is right for the symmetric eq / neq; the ordered extensions mirror
            # the operator instead so that `lit < col` means what it says.
            arr, sc = (l.array, r.scalar) if r.is_scalar else (r.array, l.scalar)
            if l.is_scalar:
                op = {OP_GT: OP_LT, OP_GTEQ: OP_LTEQ, OP_LT: OP_GT, OP_LTEQ: OP_GTEQ}.get(op, op)
            if arr.type != sc.type:
                raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[arr.type])
            if sc.value is None:
                # arrow2's *_scalar comparisons: a null scalar gives an all-null result
                return ColumnarValue(array=Array(BOOLEAN, np.zeros(len(arr), np.bool_), np.zeros(len(arr), np.bool_)))
            return ColumnarValue(array=_cmp_arrays(op, arr, _scalar_as_array1(sc), True))
        a, b = l.scalar, r.scalar
        if op in (OP_EQ, OP_NEQ):
            eq = a.type == b.type and a.value == b.value  # dyn Scalar PartialEq (:204-208)
            return ColumnarValue(scalar=Scalar(BOOLEAN, eq if op == OP_EQ else not eq))
        if a.type != b.type or a.type not in _NP:
            raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
        if a.value is None or b.value is None:
            return ColumnarValue(scalar=Scalar(BOOLEAN, None))
        res = {OP_GT: a.value > b.value, OP_GTEQ: a.value >= b.value, OP_LT: a.value < b.value,
               OP_LTEQ: a.value <= b.value}[op]
        return ColumnarValue(scalar=Scalar(BOOLEAN, bool(res)))


def _cmp_class(name, op):
    return type(name, (_Comparison,), {"OP": op})


EqExpression = _cmp_class("EqExpression", OP_EQ)
NeqExpression = _cmp_class("NeqExpression", OP_NEQ)
GtExpression = _cmp_class("GtExpression", OP_GT)
GtEqExpression = _cmp_class("GtEqExpression", OP_GTEQ)
LtExpression = _cmp_class("LtExpression", OP_LT)
LtEqExpression = _cmp_class("LtEqExpression", OP_LTEQ)


class _Boolean(PhysicalExpression):
    """AND / OR: logical-only in the reference (logical_expression.rs:293-294); arrow2's plain
    `and`/`or` semantics: value = bitwise op, validity = AND of the operands'."""
    OP = 0

    def __init__(self, left, right):
        self.left, self.right = left, right

    def evaluate(self, chunk):
        l, r = self.left.evaluate(chunk), self.right.evaluate(chunk)
        if l.is_scalar and r.is_scalar:
            a, b = l.scalar, r.scalar
            if a.type != BOOLEAN or b.type != BOOLEAN:
                raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
            if a.value is None or b.value is None:
                return ColumnarValue(scalar=Scalar(BOOLEAN, None))
            v = (a.value and b.value) if self.OP == OP_AND else (a.value or b.value)
            return ColumnarValue(scalar=Scalar(BOOLEAN, bool(v)))
        a, b = l.to_array(len(chunk)), r.to_array(len(chunk))
        if a.type != BOOLEAN or b.type != BOOLEAN:
            raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
        v = (a.values & b.values) if self.OP == OP_AND else (a.values | b.values)
        return ColumnarValue(array=Array(BOOLEAN, v, _and_validity(a.validity, b.validity)))


AndExpression = type("AndExpression", (_Boolean,), {"OP": OP_AND})
OrExpression = type("OrExpression", (_Boolean,), {"OP": OP_OR})


def _arith_arrays(op, a, b, b_scalar):
    """arrow2 compute::arithmetics::{add,sub,mul,div}[_scalar] (K3/K4)."""
    n = len(a)
    L = lib()
    validity = _and_validity(a.validity, None if b_scalar else b.validity)
    if a.type == FLOAT64:
        if op == OP_MOD:
            raise OracleError("PhysicalTypeNotSuported", "Float64")
        out = np.empty(n, np.float64)
        L.orc_arith_f64(op, _p(a.values), _p(b.values), C.c_int64(n), int(b_scalar), _p(out))
        return Array(FLOAT64, out, validity)
    if a.type not in (INT32, INT64):
        raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
    out = np.empty(n, _NP[a.type])
    vbits = pack(np.ones(n, np.bool_) if validity is None else validity)
    fn = L.orc_arith_i32 if a.type == INT32 else L.orc_arith_i64
    fn(op, _p(a.values), _p(b.values), C.c_int64(n), int(b_scalar), _p(out), _p(vbits))
    v = unpack(vbits, n)
    return Array(a.type, out, None if (validity is None and v.all()) else v)


class _Math(PhysicalExpression):
    """mathExpression! (physical_expressions.rs:235-349)."""
    OP = 0
    swap_scalar_array = True  # reproduce :264-268 (set False for the mathematically ordered variant)

    def __init__(self, left, right):
        self.left, self.right = left, right

    def evaluate(self, chunk):
        l, r = self.left.evaluate(chunk), self.right.evaluate(chunk)
        op = self.OP
        if not l.is_scalar and not r.is_scalar:
            a, b = l.array, r.array
            if len(a) != len(b):
                raise OracleError("DifferentSizes")
            if a.type != b.type:
                raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
            return ColumnarValue(array=_arith_arrays(op, a, b, False))
        if not l.is_scalar and r.is_scalar:
            arr, sc = l.array, r.scalar
        elif l.is_scalar and not r.is_scalar:
            # (Scalar, Array) arm calls `$name2(&*right, &*left)`: array OP scalar, also for sub/div
            arr, sc = r.array, l.scalar
            if not type(self).swap_scalar_array and op in (OP_SUB, OP_DIV, OP_MOD):
                full = scalar_to_array(sc, len(arr))
                return ColumnarValue(array=_arith_arrays(op, full, arr, False))
        else:
            a, b = l.scalar, r.scalar
            if a.type != b.type or a.type not in _NP or (a.type == FLOAT64 and op == OP_MOD):
                raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
            if a.value is None or b.value is None:
                return ColumnarValue(scalar=Scalar(a.type, None))
            x, y = a.value, b.value
            if a.type == FLOAT64:
                with np.errstate(all="ignore"):
                    fx, fy = np.float64(x), np.float64(y)
                    v = {OP_ADD: fx + fy, OP_SUB: fx - fy, OP_MUL: fx * fy, OP_DIV: fx / fy}[op]
                return ColumnarValue(scalar=Scalar(FLOAT64, float(v)))
            if op in (OP_DIV, OP_MOD):
                mn = -(1 << 31) if a.type == INT32 else -(1 << 63)
                if y == 0 or (x == mn and y == -1):
                    return ColumnarValue(scalar=Scalar(a.type, None))  # the reference panics
                q = abs(x) // abs(y) * (1 if (x >= 0) == (y >= 0) else -1)
                v = q if op == OP_DIV else x - q * y
            else:
                v = {OP_ADD: x + y, OP_SUB: x - y, OP_MUL: x * y}[op]
            return ColumnarValue(scalar=Scalar(a.type, _wrap_int(v, a.type)))
        if arr.type != sc.type:
            raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[arr.type])
        if sc.value is None:
            # arrow2: null scalar -> all-null array
            return ColumnarValue(array=Array(arr.type, np.zeros(len(arr), _NP[arr.type]), np.zeros(len(arr), np.bool_)))
        return ColumnarValue(array=_arith_arrays(op, arr, _scalar_as_array1(sc), True))


def _math_class(name, op):
    return type(name, (_Math,), {"OP": op})


AddExpression = _math_class("AddExpression", OP_ADD)
SubExpression = _math_class("SubExpression", OP_SUB)
MulExpression = _math_class("MulExpression", OP_MUL)
DivExpression = _math_class("DivExpression", OP_DIV)
ModExpression = _math_class("ModExpression", OP_MOD)


class _Aggregate(PhysicalExpression):
    """MaxExpression / MinExpression (physical_expressions.rs:470-499) + Sum / Count / Avg."""
    KIND = 0

    def __init__(self, expr=None):
        self.expr = expr

    def evaluate(self, chunk):
        return self.expr.evaluate(chunk)


MaxExpression = type("MaxExpression", (_Aggregate,), {"KIND": AGG_MAX})
MinExpression = type("MinExpression", (_Aggregate,), {"KIND": AGG_MIN})
SumExpression = type("SumExpression", (_Aggregate,), {"KIND": AGG_SUM})
CountExpression = type("CountExpression", (_Aggregate,), {"KIND": AGG_COUNT})
AvgExpression = type("AvgExpression", (_Aggregate,), {"KIND": AGG_AVG})


# ---------------------------------------------------------------- operators
def filter_array(a, mask):
    """compute::filter::filter_chunk for one column (K5)."""
    L = lib()
    n = len(a)
    mv = pack(mask.values)
    mval = None if mask.validity is None else pack(mask.validity)
    m = L.orc_filter_count(_p(mv), _p(mval), C.c_int64(n))
    aval = None if a.validity is None else pack(a.validity)
    oval = None if a.validity is None else np.zeros((m + 7) // 8 + 8, np.uint8)
    if a.type == UTF8:
        offs, data = a.values
        oo = np.zeros(m + 1, np.int32)
        ob = np.zeros(len(data) + 1, np.uint8)
        L.orc_filter_utf8(_p(offs), _p(data if len(data) else np.zeros(1, np.uint8)), _p(aval), C.c_int64(n), _p(mv),
                          _p(mval), _p(oo), _p(ob), _p(oval))
        vals = (oo, ob[:oo[m]].copy())
    elif a.type == BOOLEAN:
        out = np.zeros((m + 7) // 8 + 8, np.uint8)
        L.orc_filter_bitmap(_p(pack(a.values)), _p(aval), C.c_int64(n), _p(mv), _p(mval), _p(out), _p(oval))
        vals = unpack(out, m)
    else:
        out = np.empty(m, a.values.dtype)
        L.orc_filter_fixed(a.values.dtype.itemsize, _p(a.values), _p(aval), C.c_int64(n), _p(mv), _p(mval), _p(out),
                           _p(oval))
        vals = out
    return Array(a.type, vals, None if oval is None else unpack(oval, m))


class PhysicalPlan:
    def execute(self):
        raise NotImplementedError


class ScanExec(PhysicalPlan):
    def __init__(self, chunks):
        self.chunks = [c if isinstance(c, Chunk) else Chunk(c) for c in chunks]

    def execute(self):
        return iter(self.chunks)


class ProjectionExec(PhysicalPlan):
    """physical_plan/mod.rs:112-176."""

    def __init__(self, input, exprs):
        self.input, self.exprs = input, list(exprs)

    def execute(self):
        if not self.input:
            raise OracleError("MissingInputPhysicalPlan", "Projection")
        child = self.input[-1] if isinstance(self.input, list) else self.input
        for chunk in child.execute():
            yield Chunk([e.evaluate(chunk).to_array(len(chunk)) for e in self.exprs])


class SelectionExec(PhysicalPlan):
    """physical_plan/mod.rs:178-249: every input column is compacted by the predicate mask."""

    def __init__(self, input, expr):
        self.input, self.expr = input, expr

    def execute(self):
        if not self.input:
            raise OracleError("MissingInputPhysicalPlan", "Selection")
        child = self.input[-1] if isinstance(self.input, list) else self.input
        for chunk in child.execute():
            mask = self.expr.evaluate(chunk).to_array(len(chunk))
            if mask.type != BOOLEAN:
                raise OracleError("NoBooleanArrayForFilter")
            yield Chunk([filter_array(a, mask) for a in chunk.arrays])


def pack_keys(arrays, n):
    """Group key columns -> one uint64 per row (Int32: 32 bits, Int64: 64, Boolean: 1) + validity.
    No key column: every row gets key 0 (one global group, physical_plan/mod.rs:310-316)."""
    key = np.zeros(n, np.uint64)
    valid = None
    shift = 0
    for a in arrays:
        bits = {INT32: 32, INT64: 64, BOOLEAN: 1}.get(a.type)
        if bits is None:
            raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
        if shift + bits > 64:
            raise OracleError("PhysicalTypeNotSuported", "key wider than 64 bits")
        v = a.values.astype(np.int64).view(np.uint64) if a.type != BOOLEAN else a.values.astype(np.uint64)
        if bits < 64:
            v = v & np.uint64((1 << bits) - 1)
        key |= v << np.uint64(shift)
        valid = _and_validity(valid, a.validity)
        shift += bits
    return key, valid, shift


def unpack_key(keys, arrays_types):
    out, shift = [], 0
    for t in arrays_types:
        bits = {INT32: 32, INT64: 64, BOOLEAN: 1}[t]
        f = (keys >> np.uint64(shift)) & np.uint64((1 << bits) - 1 if bits < 64 else 0xFFFFFFFFFFFFFFFF)
        if t == INT32:
            out.append(Array(INT32, f.astype(np.uint32).view(np.int32)))
        elif t == INT64:
            out.append(Array(INT64, f.view(np.int64)))
        else:
            out.append(Array(BOOLEAN, f.astype(np.bool_)))
        shift += bits
    return out


def hash_aggregate(keys, key_valid, values, value_valid, row_mask=None, g_max=None):
    """Single-pass aggregate of one value column (float64 or int64 array) by packed key.
    Returns dict of first-seen-ordered arrays: keys, rows, nn, sum, min, max."""
    L = lib()
    n = len(keys)
    g_max = g_max or max(16, n)
    is_f = values is not None and values.dtype == np.float64
    if values is not None and not is_f:
        values = np.ascontiguousarray(values.astype(np.int64))
    ok = np.zeros(g_max, np.uint64)
    rows = np.zeros(g_max, np.int64)
    nn = np.zeros(g_max, np.int64)
    sum_f = np.zeros(g_max, np.float64)
    sum_i = np.zeros(g_max, np.int64)
    mn_f, mx_f = np.zeros(g_max, np.float64), np.zeros(g_max, np.float64)
    mn_i, mx_i = np.zeros(g_max, np.int64), np.zeros(g_max, np.int64)
    g = L.orc_hash_aggregate(_p(np.ascontiguousarray(keys)), _p(None if key_valid is None else pack(key_valid)),
                             _p(None if row_mask is None else pack(row_mask)), C.c_int64(n),
                             3 if is_f else 2, _p(values), _p(None if value_valid is None else pack(value_valid)),
                             C.c_int64(g_max), _p(ok), _p(rows), _p(nn), _p(sum_f), _p(sum_i), _p(mn_f), _p(mx_f),
                             _p(mn_i), _p(mx_i))
    if g < 0:
        raise OracleError("Internal", "g_max too small")
    return {"keys": ok[:g], "rows": rows[:g], "nn": nn[:g], "sum_f64": sum_f[:g], "sum_i64": sum_i[:g],
            "min": (mn_f if is_f else mn_i)[:g], "max": (mx_f if is_f else mx_i)[:g], "is_f64": is_f}


class AggregateExec(PhysicalPlan):
    """physical_plan/mod.rs:251-490.  `literal=True` restates the reference's own loop (hash-only
    group identity, per-batch membership masks, max/min only, float64 inputs) and is O(rows x groups);
    `literal=False` is the single-pass form with full key equality that also covers
    SUM / COUNT / AVG.  Output: one chunk, key columns then aggregates, groups in first-seen order
    (the reference's order is HashMap iteration order, i.e. unspecified)."""

    def __init__(self, input, group_exprs, agg_exprs, literal=False):
        self.input, self.group_exprs, self.agg_exprs, self.literal = input, list(group_exprs), list(agg_exprs), literal

    def execute(self):
        if not self.input:
            raise OracleError("MissingInputPhysicalPlan", "Aggregate")
        child = self.input[-1] if isinstance(self.input, list) else self.input
        chunks = list(child.execute())
        yield (self._literal(chunks) if self.literal else self._single_pass(chunks))

    def _inputs(self, chunk):
        n = len(chunk)
        gk = [e.evaluate(chunk).to_array(n) for e in self.group_exprs]
        ai = [(e.evaluate(chunk).to_array(n) if e.expr is not None else None) for e in self.agg_exprs]
        return gk, ai

    def _single_pass(self, chunks):
        all_keys, all_valid, key_types = [], [], None
        per_agg = [[] for _ in self.agg_exprs]
        # A single Utf8 group column is dictionary-encoded (codes in first-seen order over all chunks)
        # and aggregated on the codes; the output key column holds the strings again.  arrow2 hashes
        # utf8 arrays, so the reference groups by strings (its test_max groups by `country`).
        dictionary = {}
        utf8_key = False
        tuple_types = None  # key tuples that do not fit 64 bits: dictionary-encoded as tuples
        for ch in chunks:
            gk, ai = self._inputs(ch)
            bits = sum({INT32: 32, INT64: 64, BOOLEAN: 1, UTF8: 64}.get(a.type, 999) for a in gk)
            if len(gk) > 1 and (bits > 64 or any(a.type == UTF8 for a in gk)):
                for a in gk:
                    if a.type not in (INT32, INT64, BOOLEAN, UTF8):
                        raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
                tuple_types = [a.type for a in gk]
                cols_py = [a.strings() if a.type == UTF8 else a.values.tolist() for a in gk]
                valid = None
                for a in gk:
                    valid = _and_validity(valid, a.validity)
                codes = np.array([dictionary.setdefault(t, len(dictionary)) for t in zip(*cols_py)], dtype=np.int64)
                gk = [Array(INT64, codes, valid)]
            elif any(a.type == UTF8 for a in gk):
                utf8_key = True
                codes = np.array([dictionary.setdefault(sv, len(dictionary)) for sv in gk[0].strings()], dtype=np.int64)
                gk = [Array(INT64, codes, gk[0].validity)]
            k, kv, _ = pack_keys(gk, len(ch))
            key_types = [a.type for a in gk]
            all_keys.append(k)
            all_valid.append(np.ones(len(ch), np.bool_) if kv is None else kv)
            for j, a in enumerate(ai):
                per_agg[j].append(a)
        if not chunks:
            raise OracleError("EmptyHashmapForAggregate")
        keys = np.concatenate(all_keys)
        kvalid = np.concatenate(all_valid)
        base = hash_aggregate(keys, kvalid, None, None)
        if len(base["keys"]) == 0:
            raise OracleError("EmptyHashmapForAggregate")
        cols = unpack_key(base["keys"], key_types)
        if utf8_key:
            words = [None] * len(dictionary)
            for sv, code in dictionary.items():
                words[code] = sv
            cols[0] = Array.from_python([words[int(c)] for c in cols[0].values])
        if tuple_types is not None:
            tuples = [None] * len(dictionary)
            for t, code in dictionary.items():
                tuples[code] = t
            picked = [tuples[int(c)] for c in cols[0].values]
            cols = []
            for i, t in enumerate(tuple_types):
                vals = [p[i] for p in picked]
                cols.append(Array.from_python(vals) if t == UTF8 else Array(t, np.array(vals, dtype={INT32: np.int32, INT64: np.int64, BOOLEAN: np.bool_}[t])))
        for j, e in enumerate(self.agg_exprs):
            arrs = per_agg[j]
            if e.expr is None:
                cols.append(Array(INT64, base["rows"].copy()))
                continue
            t = arrs[0].type
            if t not in _NP:
                raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[t])
            vals = np.concatenate([a.values for a in arrs])
            vval = None
            if any(a.validity is not None for a in arrs):
                vval = np.concatenate([np.ones(len(a), np.bool_) if a.validity is None else a.validity for a in arrs])
            r = hash_aggregate(keys, kvalid, vals, vval)
            assert np.array_equal(r["keys"], base["keys"])
            nn = r["nn"]
            null = nn == 0
            if e.KIND in (AGG_MAX, AGG_MIN):
                v = r["max" if e.KIND == AGG_MAX else "min"].astype(_NP[t])
                isnan_only = np.isnan(v) if t == FLOAT64 else np.zeros(len(v), np.bool_)
                bad = null | isnan_only
                cols.append(Array(t, v, None if not bad.any() else ~bad))
            elif e.KIND == AGG_SUM:
                v = r["sum_f64"] if t == FLOAT64 else r["sum_i64"].astype(np.int64)
                if t == INT32:
                    v = (v & 0xFFFFFFFF).astype(np.uint32).view(np.int32)  # wrapping Int32 sum
                cols.append(Array(t, v, None if not null.any() else ~null))
            elif e.KIND == AGG_COUNT:
                cols.append(Array(INT64, nn.copy()))
            else:
                s = r["sum_f64"] if t == FLOAT64 else None
                if s is None:
                    rf = hash_aggregate(keys, kvalid, vals.astype(np.float64), vval)
                    s = rf["sum_f64"]
                with np.errstate(all="ignore"):
                    v = s / nn
                cols.append(Array(FLOAT64, np.where(null, 0.0, v), None if not null.any() else ~null))
        return Chunk(cols)

    def _literal(self, chunks):
        L = lib()
        g_max = 1 << 16
        states = []
        for e in self.agg_exprs:
            if e.KIND not in (AGG_MAX, AGG_MIN):
                raise OracleError("PhysicalExpressionNotSuported", "only max/min lower physically (query_planner.rs:90-100)")
            states.append({"hash": np.zeros(g_max, np.uint64), "first_row": np.zeros(g_max, np.int64),
                           "first_batch": np.zeros(g_max, np.int64), "state": np.zeros(g_max, np.uint64),
                           "set": np.zeros(g_max, np.uint8), "n": 0})
        key_slices = {}
        key_types = None
        val_type = [None] * len(self.agg_exprs)
        for bi, ch in enumerate(chunks):
            gk, ai = self._inputs(ch)
            key_types = [a.type for a in gk]
            n = len(ch)
            hashes = np.zeros(n, np.uint64)
            L.orc_hash_init(C.c_int64(n), _p(hashes))
            hv = None
            for a in gk:
                if a.type not in (INT32, INT64, BOOLEAN):
                    raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])  # arrow2 cannot hash floats
                L.orc_hash_add_i64(_p(np.ascontiguousarray(a.values.astype(np.int64))), C.c_int64(n), _p(hashes))
                hv = _and_validity(hv, a.validity)
            for j, (e, a) in enumerate(zip(self.agg_exprs, ai)):
                if a.type not in (INT32, FLOAT64, INT64):
                    raise OracleError("PhysicalTypeNotSuported", _TYPE_NAME[a.type])
                val_type[j] = a.type
                st = states[j]
                vals = np.ascontiguousarray(a.values.astype(np.float64))
                g = L.orc_aggregate_reference_batch(_p(hashes), _p(None if hv is None else pack(hv)), C.c_int64(n),
                                                    e.KIND, _p(vals), _p(None if a.validity is None else pack(a.validity)),
                                                    C.c_int64(bi), _p(st["hash"]), _p(st["first_row"]), _p(st["first_batch"]),
                                                    _p(st["state"]), _p(st["set"]), C.c_int64(st["n"]), C.c_int64(g_max))
                if g < 0:
                    raise OracleError("Internal", "too many groups for the literal aggregate")
                for k in range(st["n"], g):
                    key_slices[(int(st["first_batch"][k]), int(st["first_row"][k]))] = [
                        (a2.values[int(st["first_row"][k])], a2.type) for a2 in gk]
                st["n"] = g
        if not chunks or (states and states[0]["n"] == 0) or (not states):
            raise OracleError("EmptyHashmapForAggregate")
        n_groups = states[0]["n"]
        st0 = states[0]
        cols = []
        for ki, t in enumerate(key_types):
            vals = [key_slices[(int(st0["first_batch"][g]), int(st0["first_row"][g]))][ki][0] for g in range(n_groups)]
            cols.append(Array(t, np.array(vals, dtype=np.bool_ if t == BOOLEAN else _NP[t])))
        for j, st in enumerate(states):
            v = np.array([L.orc_ord_inverse_f64(C.c_uint64(int(st["state"][g]))) if st["set"][g] else 0.0
                          for g in range(n_groups)])
            valid = st["set"][:n_groups].astype(np.bool_)
            cols.append(Array(val_type[j], v.astype(_NP[val_type[j]]), None if valid.all() else valid))
        return Chunk(cols)
