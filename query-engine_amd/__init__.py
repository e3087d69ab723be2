"""Host-side mirror of the reference's physical-plan interface over the C ABI of
libqe_gpu.so (include/qe_gpu.h).

Names follow the reference (paths relative to /root/reference/query-engine/src):
  * expressions: ColumnExpression, Literal*Expression, EqExpression, NeqExpression,
    AddExpression .. DivExpression (physical_plan/physical_expressions.rs:26-349),
    plus the comparison / boolean / Mod operators its planner rejects
    (query_planner.rs:82) and MaxExpression / MinExpression (:503-504) with
    Sum / Count / Avg.
  * operators: ScanExec, ProjectionExec, SelectionExec, AggregateExec and
    PhysicalPlan.execute() -> iterator of batches (physical_plan/mod.rs:34-66).
  * ColumnarValue = Array | Scalar (columnar_value.rs:12-15).

Every call goes through ctypes into the HIP library; there is no CPU fallback:
importing this package without the built library raises ImportError and creating
a Context without a gfx950 device raises QeError(DeviceError).

With PyTorch in the same process, import torch first: its wheel carries a HIP
runtime of its own, and of two runtimes in one process the second to initialise
finds no device.  Loaded after torch, this library binds to torch's.  Pass
`torch.cuda.Stream().cuda_stream` (not the default stream, whose handle is 0 =
"make your own") to Context to share one stream with torch.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QE_GPU_LIBRARY") or os.path.join(_HERE, "libqe_gpu.so")  # the override serves A/B timing of two builds

# ---- constants (include/qe_gpu.h)
QE_OK, QE_END = 0, 1
TYPE_INT32, TYPE_INT64, TYPE_FLOAT64, TYPE_BOOLEAN, TYPE_UTF8, TYPE_TIMESTAMP_NS = 1, 2, 3, 4, 5, 6
MEM_HOST, MEM_DEVICE = 0, 1
OP_EQ, OP_NEQ, OP_GT, OP_GTEQ, OP_LT, OP_LTEQ, OP_AND, OP_OR, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD = range(1, 14)
AGG_MAX, AGG_MIN, AGG_SUM, AGG_COUNT, AGG_AVG = 1, 2, 3, 4, 5
AGG_MODE_SINGLE, AGG_MODE_PARTIAL, AGG_MODE_FINAL = 0, 1, 2
FLAG_REFERENCE_SCALAR_ARRAY_SWAP = 1
FLAG_NO_SPECIALIZED_KERNELS = 2
FLAG_JIT_SYNC = 4
FLAG_DEFERRED_GROUP_COUNT = 8
FLAG_DETERMINISTIC_SUM = 16
FLAG_DEFERRED_ROW_COUNT = 32
FLAG_HASH_ONLY_GROUP_IDENTITY = 64
FLAG_DEBUG_WEAK_KEY_HASH = 128

_NP_OF_TYPE = {TYPE_INT32: np.int32, TYPE_INT64: np.int64, TYPE_FLOAT64: np.float64, TYPE_TIMESTAMP_NS: np.int64}
_TYPE_OF_NP = {np.dtype(np.int32): TYPE_INT32, np.dtype(np.int64): TYPE_INT64, np.dtype(np.float64): TYPE_FLOAT64,
               np.dtype(np.bool_): TYPE_BOOLEAN}


class qe_column(C.Structure):
    _fields_ = [("type", C.c_int32), ("memory_space", C.c_int32), ("length", C.c_int64), ("null_count", C.c_int64),
                ("validity", C.c_void_p), ("offsets", C.c_void_p), ("values", C.c_void_p), ("values_bytes", C.c_int64)]


class qe_value(C.Structure):
    _fields_ = [("is_scalar", C.c_int32), ("type", C.c_int32), ("scalar_valid", C.c_int32), ("_pad", C.c_int32),
                ("scalar_i64", C.c_int64), ("scalar_f64", C.c_double), ("scalar_utf8", C.c_void_p),
                ("scalar_utf8_len", C.c_int64), ("array", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/qe_gpu.h declares
ABI = {
    "qe_version": (C.c_char_p, []),
    "qe_last_error_message": (C.c_char_p, []),
    "qe_status_name": (C.c_char_p, [C.c_int32]),
    "qe_device_count": (C.c_int32, []),
    "qe_context_create": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_context_set_flags": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "qe_context_synchronize": (C.c_int32, [C.c_void_p]),
    "qe_context_free": (None, [C.c_void_p]),
    "qe_batch_import": (C.c_int32, [C.c_void_p, C.POINTER(qe_column), C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_batch_num_rows": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]),
    "qe_batch_num_columns": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "qe_batch_column": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(qe_column)]),
    "qe_batch_to_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_batch_to_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_batch_free": (None, [C.c_void_p]),
    "qe_expr_column": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_expr_lit_i32": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_expr_lit_i64": (C.c_int32, [C.c_int64, C.POINTER(C.c_void_p)]),
    "qe_expr_lit_f64": (C.c_int32, [C.c_double, C.POINTER(C.c_void_p)]),
    "qe_expr_lit_bool": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_expr_lit_utf8": (C.c_int32, [C.c_char_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "qe_expr_binary": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_expr_clone": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_expr_to_string": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "qe_expr_evaluate": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(qe_value)]),
    "qe_value_release": (None, [C.POINTER(qe_value)]),
    "qe_expr_free": (None, [C.c_void_p]),
    "qe_agg_expr_new": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_agg_expr_free": (None, [C.c_void_p]),
    "qe_agg_state_new": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int64,
                                     C.POINTER(C.c_void_p)]),
    "qe_agg_update": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "qe_agg_finalize": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_agg_state_free": (None, [C.c_void_p]),
    "qe_plan_scan_memory": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_plan_scan_parquet": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_plan_scan_parquet_row_groups": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32,
                                                    C.POINTER(C.c_void_p)]),
    "qe_parquet_schema": (C.c_int32, [C.c_char_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "qe_plan_projection": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_plan_selection": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_plan_aggregate": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                      C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "qe_plan_to_string": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "qe_plan_num_children": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "qe_plan_free": (None, [C.c_void_p]),
    "qe_plan_execute": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_iter_next": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_iter_free": (None, [C.c_void_p]),
    "qe_comm_unique_id": (C.c_int32, [C.c_char_p]),
    "qe_comm_init": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_void_p)]),
    "qe_comm_init_loopback": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_comm_rccl_version": (C.c_int32, [C.POINTER(C.c_int32)]),
    "qe_comm_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "qe_comm_free": (None, [C.c_void_p]),
    "qe_exchange_all_gather": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "qe_exchange_by_key": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "qe_selection_noalloc": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(qe_column),
                                         C.POINTER(C.c_int64)]),
    "qe_generate_column": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                       C.c_void_p]),
    "qe_context_profile": (C.c_int32, [C.c_void_p, C.c_int32]),
    "qe_context_profile_read": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "qe_col": (C.c_int32, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "qe_lit_string": (C.c_int32, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "qe_lit_int": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_lit_int64": (C.c_int32, [C.c_int64, C.POINTER(C.c_void_p)]),
    "qe_lit_float": (C.c_int32, [C.c_double, C.POINTER(C.c_void_p)]),
    "qe_lit_bool": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_logical_binary": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_logical_aggregate": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_logical_expr_to_string": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "qe_logical_expr_free": (None, [C.c_void_p]),
    "qe_dataframe_parquet": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "qe_dataframe_from_batches": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                              C.POINTER(C.c_void_p)]),
    "qe_dataframe_project": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "qe_dataframe_filter": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_dataframe_aggregate": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                           C.POINTER(C.c_void_p)]),
    "qe_dataframe_optimize": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_dataframe_schema": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "qe_dataframe_logical_plan": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "qe_dataframe_to_physical_plan": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_dataframe_execute": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qe_dataframe_free": (None, [C.c_void_p]),
    "qe_lowering_describe": (C.c_int32, [C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_int32, C.c_uint32, C.c_void_p,
                                         C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                         C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_char_p, C.c_int64,
                                         C.POINTER(C.c_int64)]),
    "qe_lowering_compile": (C.c_int32, [C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_int32, C.c_uint32, C.c_void_p,
                                        C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                        C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python query-engine_amd/build.py` "
            "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class QeError(Exception):
    """A non-OK qe_status; `.name` is the reference's Error variant name (error.rs:3-35)."""

    def __init__(self, status):
        self.status = status
        self.name = lib.qe_status_name(status).decode()
        self.message = lib.qe_last_error_message().decode(errors="replace")
        super().__init__(f"{self.name} ({status}): {self.message}")


def _check(status):
    if status != QE_OK:
        raise QeError(status)


def device_count():
    return lib.qe_device_count()


# ------------------------------------------------------------------ context
class Context:
    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        _check(lib.qe_context_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = device
        self.flags = FLAG_REFERENCE_SCALAR_ARRAY_SWAP  # the library's default

    def set_flags(self, flags):
        _check(lib.qe_context_set_flags(self._h, flags))
        self.flags = flags

    def synchronize(self):
        _check(lib.qe_context_synchronize(self._h))

    def profile(self, enable=True):
        """True / 1: time every launch; 2: only the main kernels of large launches; False: off."""
        _check(lib.qe_context_profile(self._h, int(enable)))

    def profile_read(self):
        import json
        buf = C.create_string_buffer(1 << 16)
        need = C.c_int64()
        _check(lib.qe_context_profile_read(self._h, buf, len(buf), C.byref(need)))
        return json.loads(buf.value.decode())

    def generate(self, kind, seed, row0, n, lo, hi, device_ptr):
        _check(lib.qe_generate_column(self._h, kind, seed, row0, n, float(lo), float(hi), C.c_void_p(device_ptr)))

    def close(self):
        if self._h:
            lib.qe_context_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------- batches
def _pack_bits(mask):
    return np.packbits(np.asarray(mask, dtype=np.bool_), bitorder="little")


def _unpack_bits(buf, n):
    return np.unpackbits(np.frombuffer(buf, dtype=np.uint8), bitorder="little")[:n].astype(np.bool_)


class Column:
    """Host-side view of one Arrow array: `values` (numpy array, or list of bytes for
    utf8) and `validity` (bool numpy array or None)."""

    def __init__(self, type_, values, validity=None):
        self.type = type_
        self.values = values
        self.validity = validity

    def __len__(self):
        return len(self.values)

    def to_pylist(self):
        out = []
        for i in range(len(self.values)):
            if self.validity is not None and not self.validity[i]:
                out.append(None)
            elif self.type == TYPE_UTF8:
                out.append(self.values[i].decode())
            else:
                out.append(self.values[i].item())
        return out


def _column_from_python(x):
    """numpy array | (values, validity) | list of str/None | Column -> Column."""
    if isinstance(x, Column):
        return x
    validity = None
    if isinstance(x, tuple):
        x, validity = x
        if validity is not None:
            validity = np.asarray(validity, dtype=np.bool_)
    if isinstance(x, (list, tuple)) and (len(x) == 0 or isinstance(x[0], (str, bytes, type(None)))) and not isinstance(x, np.ndarray):
        vals, val = [], []
        for s in x:
            val.append(s is not None)
            vals.append(b"" if s is None else (s.encode() if isinstance(s, str) else s))
        v = np.array(val, dtype=np.bool_)
        if validity is not None:
            v &= validity
        return Column(TYPE_UTF8, vals, None if v.all() else v)
    arr = np.ascontiguousarray(x)
    if arr.dtype not in _TYPE_OF_NP:
        raise TypeError(f"unsupported dtype {arr.dtype}")
    return Column(_TYPE_OF_NP[arr.dtype], arr, validity)


class Batch:
    """RecordBatch handle (device or host resident)."""

    def __init__(self, ctx, handle, keepalive=None):
        self.ctx = ctx
        self._h = handle
        self._keep = keepalive

    # -- constructors
    @staticmethod
    def from_host_columns(ctx, columns):
        cols = [_column_from_python(c) for c in columns]
        descs = (qe_column * max(1, len(cols)))()
        keep = []
        for i, c in enumerate(cols):
            d = descs[i]
            d.type = c.type
            d.memory_space = MEM_HOST
            d.length = len(c.values)
            if c.type == TYPE_UTF8:
                offs = np.zeros(len(c.values) + 1, dtype=np.int32)
                np.cumsum([len(s) for s in c.values], out=offs[1:])
                data = np.frombuffer(b"".join(c.values) + b"\0", dtype=np.uint8).copy()
                keep += [offs, data]
                d.offsets = offs.ctypes.data
                d.values = data.ctypes.data
                d.values_bytes = int(offs[-1])
            elif c.type == TYPE_BOOLEAN:
                bits = _pack_bits(c.values)
                bits = np.concatenate([bits, np.zeros(8, np.uint8)])
                keep.append(bits)
                d.values = bits.ctypes.data
                d.values_bytes = (len(c.values) + 7) // 8
            else:
                arr = np.ascontiguousarray(c.values)
                keep.append(arr)
                d.values = arr.ctypes.data if arr.size else None
                d.values_bytes = arr.nbytes
            if c.validity is not None:
                vb = np.concatenate([_pack_bits(c.validity), np.zeros(8, np.uint8)])
                keep.append(vb)
                d.validity = vb.ctypes.data
                d.null_count = int(len(c.validity) - np.count_nonzero(c.validity))
        h = C.c_void_p()
        _check(lib.qe_batch_import(ctx._h, descs, len(cols), C.byref(h)))
        return Batch(ctx, h, keep)

    @staticmethod
    def from_numpy(ctx, columns):
        """Host columns copied to the device (one transfer per buffer)."""
        host = Batch.from_host_columns(ctx, columns)
        return host.to_device()

    @staticmethod
    def from_device_columns(ctx, columns, keepalive=None):
        """columns: list of dicts {type, length, values, validity=0, offsets=0, values_bytes=0, null_count=0}
        with raw device addresses; buffers are borrowed."""
        descs = (qe_column * max(1, len(columns)))()
        for i, c in enumerate(columns):
            d = descs[i]
            d.type = c["type"]
            d.memory_space = MEM_DEVICE
            d.length = c["length"]
            d.values = c["values"]
            d.validity = c.get("validity") or None
            d.offsets = c.get("offsets") or None
            d.values_bytes = c.get("values_bytes", 0)
            d.null_count = c.get("null_count", 0)
        h = C.c_void_p()
        _check(lib.qe_batch_import(ctx._h, descs, len(columns), C.byref(h)))
        return Batch(ctx, h, keepalive)

    @staticmethod
    def from_torch(ctx, tensors):
        """1-D contiguous int32 / int64 / float64 CUDA tensors -> device batch (zero copy)."""
        import torch
        tmap = {torch.int32: TYPE_INT32, torch.int64: TYPE_INT64, torch.float64: TYPE_FLOAT64}
        cols = []
        for t in tensors:
            assert t.is_cuda and t.dim() == 1 and t.is_contiguous() and t.dtype in tmap
            cols.append({"type": tmap[t.dtype], "length": t.numel(), "values": t.data_ptr()})
        return Batch.from_device_columns(ctx, cols, keepalive=list(tensors))

    # -- accessors
    @property
    def num_rows(self):
        n = C.c_int64()
        _check(lib.qe_batch_num_rows(self._h, C.byref(n)))
        return n.value

    @property
    def num_columns(self):
        n = C.c_int32()
        _check(lib.qe_batch_num_columns(self._h, C.byref(n)))
        return n.value

    def __len__(self):
        return self.num_rows

    def column_desc(self, i):
        d = qe_column()
        _check(lib.qe_batch_column(self._h, i, C.byref(d)))
        return d

    def to_device(self):
        h = C.c_void_p()
        _check(lib.qe_batch_to_device(self.ctx._h, self._h, C.byref(h)))
        return Batch(self.ctx, h)

    def to_host(self):
        h = C.c_void_p()
        _check(lib.qe_batch_to_host(self.ctx._h, self._h, C.byref(h)))
        return Batch(self.ctx, h)

    def columns(self):
        """Copies the batch to the host (if needed) and returns a list of Column."""
        d0 = self.column_desc(0) if self.num_columns else None
        host = self if (d0 is None or d0.memory_space == MEM_HOST) else self.to_host()
        out = []
        for i in range(host.num_columns):
            d = host.column_desc(i)
            n = d.length
            validity = None
            if d.validity:
                validity = _unpack_bits(C.string_at(d.validity, (n + 7) // 8), n)
            if d.type == TYPE_UTF8:
                offs = np.frombuffer(C.string_at(d.offsets, (n + 1) * 4), dtype=np.int32)
                data = C.string_at(d.values, int(offs[-1]) if n else 0) if d.values else b""
                vals = [data[offs[j]:offs[j + 1]] for j in range(n)]
            elif d.type == TYPE_BOOLEAN:
                vals = _unpack_bits(C.string_at(d.values, (n + 7) // 8), n)
            else:
                dt = _NP_OF_TYPE[d.type]
                vals = np.frombuffer(C.string_at(d.values, n * np.dtype(dt).itemsize), dtype=dt).copy() if n else np.zeros(0, dt)
            col = Column(d.type, vals, validity)
            col.null_count = d.null_count
            out.append(col)
        return out

    def free(self):
        if self._h:
            lib.qe_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


COMM_ID_BYTES = 128


class Comm:
    """The exchange step of a multi-GPU aggregate over RCCL (qe_comm_*, qe_exchange_*): one per rank, on the context's
    device and stream.  `Comm.unique_id()` on one rank, the bytes handed to every rank by the caller's own means."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(lib.qe_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def library_rccl_version():
        """ncclGetVersion of the RCCL the library binds (QeError DeviceError when none can be loaded)."""
        v = C.c_int32()
        _check(lib.qe_comm_rccl_version(C.byref(v)))
        return v.value

    def __init__(self, ctx, rank, world, unique_id=None, loopback=False):
        h = C.c_void_p()
        if loopback:
            # no peers: the exchanges are laid out as rank `rank` of `world` lays them out, moved by device copies
            _check(lib.qe_comm_init_loopback(ctx._h, rank, world, C.byref(h)))
        else:
            if unique_id is None or len(unique_id) != COMM_ID_BYTES:
                raise ValueError("unique id must be COMM_ID_BYTES long")
            _check(lib.qe_comm_init(ctx._h, rank, world, unique_id, C.byref(h)))
        self._h, self.ctx, self.rank, self.world, self.loopback = h, ctx, rank, world, loopback

    @property
    def rccl_version(self):
        v = C.c_int32()
        _check(lib.qe_comm_info(self._h, None, None, C.byref(v)))
        return v.value

    def all_gather(self, partial, padded_rows):
        """Every rank's partial table in one batch of world x padded_rows rows (unfilled rows zero)."""
        h = C.c_void_p()
        _check(lib.qe_exchange_all_gather(self._h, partial._h, padded_rows, C.byref(h)))
        return Batch(self.ctx, h)

    def exchange_by_key(self, partial, n_key_columns, max_rows_per_rank):
        """The rows whose key this rank owns, from all ranks, in one batch (unfilled rows zero)."""
        h = C.c_void_p()
        _check(lib.qe_exchange_by_key(self._h, partial._h, n_key_columns, max_rows_per_rank, C.byref(h)))
        return Batch(self.ctx, h)

    def close(self):
        if self._h:
            lib.qe_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


NOALLOC_MAX_ROWS = 1 << 29  # QE_NOALLOC_MAX_ROWS


def selection_noalloc(batch, predicate, exprs, outs):
    """qe_selection_noalloc: Selection(predicate) -> Projection(exprs) over one device batch into buffers the caller
    provides, with no device allocation inside the call.  outs: per expression a dict {type, values (device address),
    length (capacity in rows), validity (device address or 0)}.  Returns the number of rows selected."""
    hs = []
    try:
        ph = predicate._build()
        hs.append(ph)
        for e in exprs:
            hs.append(e._build())
        arr = (C.c_void_p * max(1, len(exprs)))(*[h.value for h in hs[1:]])
        descs = (qe_column * max(1, len(outs)))()
        for i, o in enumerate(outs):
            d = descs[i]
            d.type, d.memory_space, d.length = o["type"], MEM_DEVICE, o["length"]
            d.values = o["values"]
            d.validity = o.get("validity") or None
        rows = C.c_int64()
        _check(lib.qe_selection_noalloc(batch.ctx._h, batch._h, ph, arr, len(exprs), descs, C.byref(rows)))
        return rows.value
    finally:
        for h in hs:
            lib.qe_expr_free(h)


# -------------------------------------------------------------- expressions
class ColumnarValue:
    """Array (a one-column Batch) or Scalar (python value / None)."""

    def __init__(self, type_, array=None, scalar=None, is_scalar=False):
        self.type = type_
        self.array = array
        self.scalar = scalar
        self.is_scalar = is_scalar

    def to_array(self, length):
        """scalar_to_array (columnar_value.rs:26-59) is done by ProjectionExec on the device;
        here only the Array arm is exposed."""
        if not self.is_scalar:
            return self.array.columns()[0]
        raise QeError(21)


class PhysicalExpression:
    """trait PhysicalExpression: Display { fn evaluate(&self, input) -> Result<ColumnarValue, Error> }"""

    def _build(self):
        raise NotImplementedError

    def evaluate(self, batch):
        h = self._build()
        v = qe_value()
        try:
            _check(lib.qe_expr_evaluate(batch.ctx._h, h, batch._h, C.byref(v)))
        finally:
            lib.qe_expr_free(h)
        try:
            if v.is_scalar:
                if not v.scalar_valid:
                    val = None
                elif v.type == TYPE_FLOAT64:
                    val = v.scalar_f64
                elif v.type == TYPE_BOOLEAN:
                    val = bool(v.scalar_i64)
                elif v.type == TYPE_UTF8:
                    val = C.string_at(v.scalar_utf8, v.scalar_utf8_len).decode()
                else:
                    val = v.scalar_i64
                return ColumnarValue(v.type, scalar=val, is_scalar=True)
            arr = Batch(batch.ctx, C.c_void_p(v.array))
            v.array = None
            return ColumnarValue(v.type, array=arr)
        finally:
            lib.qe_value_release(C.byref(v))

    def __str__(self):
        h = self._build()
        try:
            buf = C.create_string_buffer(4096)
            need = C.c_int64()
            _check(lib.qe_expr_to_string(h, buf, len(buf), C.byref(need)))
            return buf.value.decode()
        finally:
            lib.qe_expr_free(h)


class ColumnExpression(PhysicalExpression):
    def __init__(self, index):
        self.index = index

    def _build(self):
        h = C.c_void_p()
        _check(lib.qe_expr_column(self.index, C.byref(h)))
        return h


class LiteralIntegerExpression(PhysicalExpression):
    """i32 literal (physical_expressions.rs:120-131)."""

    def __init__(self, value):
        self.value = value

    def _build(self):
        h = C.c_void_p()
        _check(lib.qe_expr_lit_i32(self.value, C.byref(h)))
        return h


class LiteralInt64Expression(PhysicalExpression):
    def __init__(self, value):
        self.value = value

    def _build(self):
        h = C.c_void_p()
        _check(lib.qe_expr_lit_i64(self.value, C.byref(h)))
        return h


class LiteralFloatExpression(PhysicalExpression):
    def __init__(self, value):
        self.value = value

    def _build(self):
        h = C.c_void_p()
        _check(lib.qe_expr_lit_f64(self.value, C.byref(h)))
        return h


class LiteralBoolExpression(PhysicalExpression):
    def __init__(self, value):
        self.value = value

    def _build(self):
        h = C.c_void_p()
        _check(lib.qe_expr_lit_bool(1 if self.value else 0, C.byref(h)))
        return h


class LiteralStringExpression(PhysicalExpression):
    def __init__(self, value):
        self.value = value

    def _build(self):
        h = C.c_void_p()
        b = self.value.encode() if isinstance(self.value, str) else self.value
        _check(lib.qe_expr_lit_utf8(b, len(b), C.byref(h)))
        return h


class _Binary(PhysicalExpression):
    OP = 0

    def __init__(self, left, right):
        self.left = left
        self.right = right

    def _build(self):
        l = self.left._build()
        try:
            r = self.right._build()
        except Exception:
            lib.qe_expr_free(l)
            raise
        h = C.c_void_p()
        _check(lib.qe_expr_binary(self.OP, l, r, C.byref(h)))  # consumes l and r, also on failure
        return h


def _binary(name, op):
    return type(name, (_Binary,), {"OP": op})


EqExpression = _binary("EqExpression", OP_EQ)
NeqExpression = _binary("NeqExpression", OP_NEQ)
GtExpression = _binary("GtExpression", OP_GT)
GtEqExpression = _binary("GtEqExpression", OP_GTEQ)
LtExpression = _binary("LtExpression", OP_LT)
LtEqExpression = _binary("LtEqExpression", OP_LTEQ)
AndExpression = _binary("AndExpression", OP_AND)
OrExpression = _binary("OrExpression", OP_OR)
AddExpression = _binary("AddExpression", OP_ADD)
SubExpression = _binary("SubExpression", OP_SUB)
MulExpression = _binary("MulExpression", OP_MUL)
DivExpression = _binary("DivExpression", OP_DIV)
ModExpression = _binary("ModExpression", OP_MOD)


class PhysicalAggregateExpression:
    KIND = 0

    def __init__(self, expr=None):
        self.expr = expr

    def _build(self):
        inp = self.expr._build() if self.expr is not None else None
        h = C.c_void_p()
        _check(lib.qe_agg_expr_new(self.KIND, inp, C.byref(h)))
        return h


def _agg(name, kind):
    return type(name, (PhysicalAggregateExpression,), {"KIND": kind})


MaxExpression = _agg("MaxExpression", AGG_MAX)
MinExpression = _agg("MinExpression", AGG_MIN)
SumExpression = _agg("SumExpression", AGG_SUM)
CountExpression = _agg("CountExpression", AGG_COUNT)
AvgExpression = _agg("AvgExpression", AGG_AVG)


# -------------------------------------------------------------------- plans
class PhysicalPlan:
    """enum PhysicalPlan { Scan, Projection, Selection, Aggregate } (physical_plan/mod.rs:34-39)."""

    def __init__(self, ctx, children):
        self.ctx = ctx
        self._children = children
        self._consumed = False

    def children(self):
        return self._children or None

    def _build(self):
        raise NotImplementedError

    def describe(self):
        h = self._build()
        try:
            buf = C.create_string_buffer(1 << 14)
            need = C.c_int64()
            _check(lib.qe_plan_to_string(h, buf, len(buf), C.byref(need)))
            return buf.value.decode()
        finally:
            lib.qe_plan_free(h)

    def execute(self):
        """PhysicalPlan::execute(self): consumes the plan, returns an iterator of Batch."""
        if self._consumed:
            raise RuntimeError("a physical plan is single use: execute(self) consumes it (physical_plan/mod.rs:58)")
        self._consumed = True
        h = self._build()
        it = C.c_void_p()
        _check(lib.qe_plan_execute(h, C.byref(it)))  # consumes h
        return _BatchIterator(self.ctx, it)


class _BatchIterator:
    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle

    def __iter__(self):
        return self

    def __next__(self):
        b = C.c_void_p()
        st = lib.qe_iter_next(self._h, C.byref(b))
        if st == QE_END:
            raise StopIteration
        _check(st)
        return Batch(self.ctx, b)

    def __del__(self):
        try:
            if self._h:
                lib.qe_iter_free(self._h)
                self._h = None
        except Exception:
            pass


class ScanExec(PhysicalPlan):
    """In-memory, device-resident scan leaf (stands where ScanExec::new(data_source, projection, schema) does)."""

    def __init__(self, ctx, batches):
        super().__init__(ctx, [])
        self.batches = list(batches)

    def _build(self):
        arr = (C.c_void_p * max(1, len(self.batches)))(*[b._h for b in self.batches])
        h = C.c_void_p()
        _check(lib.qe_plan_scan_memory(self.ctx._h, arr, len(self.batches), C.byref(h)))
        return h


class ParquetScanExec(PhysicalPlan):
    """ScanExec over DataSource::Parquet (physical_plan/mod.rs:68-110, data_source.rs:24-72): one
    device-resident batch per row group; `projection` = column names (None = all), file order."""

    def __init__(self, ctx, path, projection=None, row_groups=None):
        """row_groups: (first, count) to scan a contiguous run of the file's row groups (one rank's share)."""
        super().__init__(ctx, [])
        self.path, self.projection = path, None if projection is None else list(projection)
        self.first, self.count = (0, -1) if row_groups is None else row_groups

    def _build(self):
        h = C.c_void_p()
        names, n = None, -1
        if self.projection is not None:
            names = (C.c_char_p * max(1, len(self.projection)))(*[x.encode() for x in self.projection])
            n = len(self.projection)
        _check(lib.qe_plan_scan_parquet_row_groups(self.ctx._h, self.path.encode(), names, n, self.first, self.count, C.byref(h)))
        return h


def parquet_schema(path):
    """ParquetDataSource::schema (data_source.rs:45-47): {"fields": [...], "rows": N, "row_groups": G}.  Needs no GPU."""
    import json
    need = C.c_int64()
    _check(lib.qe_parquet_schema(path.encode(), None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value + 1)
    _check(lib.qe_parquet_schema(path.encode(), buf, len(buf), None))
    return json.loads(buf.value.decode())


def _build_list(items):
    handles = []
    try:
        for it in items:
            handles.append(it._build())
    except Exception:
        for h in handles:
            lib.qe_expr_free(h)
        raise
    return handles


class ProjectionExec(PhysicalPlan):
    def __init__(self, input, exprs):
        super().__init__(input[0].ctx if isinstance(input, list) else input.ctx, input if isinstance(input, list) else [input])
        self.exprs = list(exprs)

    def _build(self):
        child = self._children[-1]._build()
        try:
            hs = _build_list(self.exprs)
        except Exception:
            lib.qe_plan_free(child)
            raise
        arr = (C.c_void_p * max(1, len(hs)))(*hs)
        h = C.c_void_p()
        _check(lib.qe_plan_projection(child, arr, len(hs), C.byref(h)))
        return h


class SelectionExec(PhysicalPlan):
    def __init__(self, input, expr):
        super().__init__(input[0].ctx if isinstance(input, list) else input.ctx, input if isinstance(input, list) else [input])
        self.expr = expr

    def _build(self):
        child = self._children[-1]._build()
        try:
            e = self.expr._build()
        except Exception:
            lib.qe_plan_free(child)
            raise
        h = C.c_void_p()
        _check(lib.qe_plan_selection(child, e, C.byref(h)))
        return h


class AggregateExec(PhysicalPlan):
    def __init__(self, input, group_exprs, agg_exprs, mode=AGG_MODE_SINGLE, group_capacity_hint=0):
        super().__init__(input[0].ctx if isinstance(input, list) else input.ctx, input if isinstance(input, list) else [input])
        self.group_exprs = list(group_exprs)
        self.agg_exprs = list(agg_exprs)
        self.mode = mode
        self.group_capacity_hint = group_capacity_hint

    def _build(self):
        child = self._children[-1]._build()
        try:
            gs = _build_list(self.group_exprs)
        except Exception:
            lib.qe_plan_free(child)
            raise
        ags = []
        try:
            for a in self.agg_exprs:
                ags.append(a._build())
        except Exception:
            lib.qe_plan_free(child)
            for g in gs:
                lib.qe_expr_free(g)
            for a in ags:
                lib.qe_agg_expr_free(a)
            raise
        garr = (C.c_void_p * max(1, len(gs)))(*gs)
        aarr = (C.c_void_p * max(1, len(ags)))(*ags)
        h = C.c_void_p()
        _check(lib.qe_plan_aggregate(child, garr, len(gs), aarr, len(ags), self.mode, self.group_capacity_hint, C.byref(h)))
        return h


class AggregationState:
    """The Accumulator protocol (physical_expressions.rs:351-362) over the device group table:
    `accumulate(batch)` folds one device-resident batch in, `final_value()` returns the output Batch."""

    def __init__(self, ctx, group_exprs, agg_exprs, mode=AGG_MODE_SINGLE, group_capacity_hint=0):
        self.ctx = ctx
        self._batches = []  # the state borrows their buffers until it is freed
        gs = _build_list(group_exprs)
        ags = []
        try:
            for a in agg_exprs:
                ags.append(a._build())
        except Exception:
            for g in gs:
                lib.qe_expr_free(g)
            for a in ags:
                lib.qe_agg_expr_free(a)
            raise
        garr = (C.c_void_p * max(1, len(gs)))(*gs)
        aarr = (C.c_void_p * max(1, len(ags)))(*ags)
        self._h = C.c_void_p()
        _check(lib.qe_agg_state_new(ctx._h, garr, len(gs), aarr, len(ags), mode, group_capacity_hint, C.byref(self._h)))

    def accumulate(self, batch):
        _check(lib.qe_agg_update(self._h, batch._h))
        self._batches.append(batch)

    def final_value(self):
        b = C.c_void_p()
        _check(lib.qe_agg_finalize(self._h, C.byref(b)))
        return Batch(self.ctx, b)

    def close(self):
        if self._h:
            lib.qe_agg_state_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partial_layout(n_group_exprs, agg_exprs):
    """Column index of each aggregate's first partial column in a QE_AGG_MODE_PARTIAL batch
    (keys first, then per aggregate 1 column, AVG 2: sum and count)."""
    pos, out = n_group_exprs, []
    for a in agg_exprs:
        out.append(pos)
        pos += 2 if a.KIND == AGG_AVG else 1
    return out, pos


def final_plan(scan_of_partials, n_group_exprs, agg_exprs, group_capacity_hint=0):
    """AggregateExec in FINAL mode over batches produced by PARTIAL aggregates with the same
    group / aggregate lists."""
    first, _ = partial_layout(n_group_exprs, agg_exprs)
    groups = [ColumnExpression(i) for i in range(n_group_exprs)]
    aggs = [type(a)(ColumnExpression(first[i])) for i, a in enumerate(agg_exprs)]
    return AggregateExec(scan_of_partials, groups, aggs, mode=AGG_MODE_FINAL, group_capacity_hint=group_capacity_hint)


def _lowering_call(fn, col_types, col_nullable, predicate, exprs, group_exprs, agg_exprs, mode, flags, tail):
    n = len(col_types)
    types = (C.c_int32 * max(1, n))(*col_types)
    nullable = (C.c_uint8 * max(1, n))(*[1 if x else 0 for x in (col_nullable or [False] * n)])
    pred = predicate._build() if predicate is not None else None
    es, gs, ags = _build_list(exprs), _build_list(group_exprs), [a._build() for a in agg_exprs]
    try:
        earr = (C.c_void_p * max(1, len(es)))(*es)
        garr = (C.c_void_p * max(1, len(gs)))(*gs)
        aarr = (C.c_void_p * max(1, len(ags)))(*ags)
        _check(fn(types, nullable, n, flags, pred, earr, len(es), garr, len(gs), aarr, len(ags), mode, *tail))
    finally:
        if pred is not None:
            lib.qe_expr_free(pred)
        for h in es + gs:
            lib.qe_expr_free(h)
        for h in ags:
            lib.qe_agg_expr_free(h)


def describe_lowering(col_types, col_nullable=None, predicate=None, exprs=(), group_exprs=(), agg_exprs=(),
                      mode=AGG_MODE_SINGLE, flags=FLAG_REFERENCE_SCALAR_ARRAY_SWAP):
    """What Scan(schema) -> [Selection] -> [Projection] -> [Aggregate] compiles to: a dict with one entry per
    kernel launch (interpreter code, cache layout, sink descriptors, launch signature).  Needs no GPU."""
    import json
    buf = C.create_string_buffer(1 << 18)
    need = C.c_int64()
    _lowering_call(lib.qe_lowering_describe, col_types, col_nullable, predicate, exprs, group_exprs, agg_exprs, mode, flags,
                   (buf, len(buf), C.byref(need)))
    return json.loads(buf.value.decode())


def compile_lowering(col_types, col_nullable=None, predicate=None, exprs=(), group_exprs=(), agg_exprs=(),
                     mode=AGG_MODE_SINGLE, flags=FLAG_REFERENCE_SCALAR_ARRAY_SWAP):
    """Run-time compiles the shape-specialised kernel of every launch of that pipeline (no GPU needed).
    Returns (kernels built, total code bytes)."""
    nk, nb = C.c_int32(), C.c_int64()
    _lowering_call(lib.qe_lowering_compile, col_types, col_nullable, predicate, exprs, group_exprs, agg_exprs, mode, flags,
                   (C.byref(nk), C.byref(nb)))
    return nk.value, nb.value


from .dataframe import DataFrame, LogicalExpression  # noqa: E402,F401
from . import prelude  # noqa: E402,F401
