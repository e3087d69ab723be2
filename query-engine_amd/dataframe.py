"""DataFrame builder and logical expressions over the C ABI (qe_dataframe_*, qe_col, qe_lit_*, ...):
the reference's dataframe.rs / logical_expression.rs seen from Python.  A DataFrame is immutable here
(every builder returns a new one), handles are built when a result is asked for."""
import ctypes as C
import json


def _qe():
    import importlib
    return importlib.import_module(__package__)


class LogicalExpression:
    """enum LogicalExpression (logical_expression.rs:12-36) with the LogicalExpressionMethods
    (:425-472: eq, neq, gt, gteq, lt, lteq, and, or) and arithmetic operators."""

    def __init__(self, kind, value=None, op=None, left=None, right=None):
        self.kind, self.value, self.op, self.left, self.right = kind, value, op, left, right

    def _build(self):
        qe = _qe()
        h = C.c_void_p()
        if self.kind == "col":
            qe._check(qe.lib.qe_col(self.value.encode(), C.byref(h)))
        elif self.kind == "string":
            qe._check(qe.lib.qe_lit_string(self.value.encode(), C.byref(h)))
        elif self.kind == "int":
            qe._check(qe.lib.qe_lit_int(self.value, C.byref(h)))
        elif self.kind == "int64":
            qe._check(qe.lib.qe_lit_int64(self.value, C.byref(h)))
        elif self.kind == "float":
            qe._check(qe.lib.qe_lit_float(self.value, C.byref(h)))
        elif self.kind == "bool":
            qe._check(qe.lib.qe_lit_bool(1 if self.value else 0, C.byref(h)))
        elif self.kind == "binary":
            l = self.left._build()
            try:
                r = self.right._build()
            except Exception:
                qe.lib.qe_logical_expr_free(l)
                raise
            qe._check(qe.lib.qe_logical_binary(self.op, l, r, C.byref(h)))  # consumes l, r
        else:
            qe._check(qe.lib.qe_logical_aggregate(self.op, self.left._build(), C.byref(h)))
        return h

    def __str__(self):
        qe = _qe()
        h = self._build()
        try:
            buf = C.create_string_buffer(1 << 14)
            qe._check(qe.lib.qe_logical_expr_to_string(h, buf, len(buf), None))
            return buf.value.decode()
        finally:
            qe.lib.qe_logical_expr_free(h)

    def _bin(self, op, other):
        return LogicalExpression("binary", op=op, left=self, right=other)

    def eq(self, other): return self._bin(_qe().OP_EQ, other)
    def neq(self, other): return self._bin(_qe().OP_NEQ, other)
    def gt(self, other): return self._bin(_qe().OP_GT, other)
    def gteq(self, other): return self._bin(_qe().OP_GTEQ, other)
    def lt(self, other): return self._bin(_qe().OP_LT, other)
    def lteq(self, other): return self._bin(_qe().OP_LTEQ, other)
    def and_(self, other): return self._bin(_qe().OP_AND, other)
    def or_(self, other): return self._bin(_qe().OP_OR, other)
    def __add__(self, other): return self._bin(_qe().OP_ADD, other)
    def __sub__(self, other): return self._bin(_qe().OP_SUB, other)
    def __mul__(self, other): return self._bin(_qe().OP_MUL, other)
    def __truediv__(self, other): return self._bin(_qe().OP_DIV, other)
    def __mod__(self, other): return self._bin(_qe().OP_MOD, other)


def _handles(exprs):
    qe = _qe()
    hs = []
    try:
        for e in exprs:
            hs.append(e._build())
    except Exception:
        for h in hs:
            qe.lib.qe_logical_expr_free(h)
        raise
    return hs


class DataFrame:
    """struct DataFrame { plan } (dataframe.rs:29-84)."""

    def __init__(self, ctx, names, batches, label="memory", steps=(), path=None):
        self.ctx, self.names, self.batches, self.label, self.steps = ctx, list(names), list(batches), label, tuple(steps)
        self.path = path  # a Parquet file scanned on the device instead of `batches`

    def _with(self, step):
        return DataFrame(self.ctx, self.names, self.batches, self.label, self.steps + (step,), self.path)

    # -- sources ------------------------------------------------------------
    @staticmethod
    def from_batches(ctx, names, batches, label="memory"):
        return DataFrame(ctx, names, batches, label)

    @staticmethod
    def from_arrow(ctx, table, label="arrow"):
        """A pyarrow Table (or RecordBatch) uploaded once, one device batch per Arrow chunk.
        Timestamps travel as Int64 (their raw unit), as in the golden fixture."""
        import numpy as np
        import pyarrow as pa
        qe = _qe()
        if isinstance(table, pa.RecordBatch):
            table = pa.Table.from_batches([table])
        batches = []
        for rb in table.combine_chunks().to_batches() or [None]:
            if rb is None:
                break
            cols = []
            for arr in rb.columns:
                t = arr.type
                valid = None if arr.null_count == 0 else np.asarray(arr.is_valid())
                if pa.types.is_string(t) or pa.types.is_large_string(t):
                    cols.append(qe.Column(qe.TYPE_UTF8, [b"" if s is None else s.encode() for s in arr.to_pylist()], valid))
                    continue
                if pa.types.is_timestamp(t):
                    arr = arr.cast(pa.int64())
                elif pa.types.is_boolean(t):
                    cols.append(qe.Column(qe.TYPE_BOOLEAN, np.asarray(arr.fill_null(False)), valid))
                    continue
                fill = arr.fill_null(0) if arr.null_count else arr
                np_arr = np.ascontiguousarray(fill.to_numpy(zero_copy_only=False))
                if np_arr.dtype not in qe._TYPE_OF_NP:
                    raise TypeError(f"Arrow type {t} has no device representation")
                cols.append(qe.Column(qe._TYPE_OF_NP[np_arr.dtype], np_arr, valid))
            batches.append(qe.Batch.from_numpy(ctx, cols))
        return DataFrame(ctx, table.schema.names, batches, label)

    @staticmethod
    def parquet(ctx, path):
        """DataFrame::parquet (dataframe.rs:38-41): a scan of the file, one batch per row group, decoded
        on the device (qe_dataframe_parquet).  Opening the file is deferred to the first use."""
        return DataFrame(ctx, [], [], label=path, path=path)

    # -- builders (DataFrameTrait, dataframe.rs:44-69) ------------------------
    def project(self, exprs):
        return self._with(("project", list(exprs)))

    def filter(self, expr):
        return self._with(("filter", expr))

    def aggregate(self, group_by, aggregate_expr):
        return self._with(("aggregate", list(group_by), list(aggregate_expr)))

    def optimize(self):
        """LogicalPlan::optimize: projection push-down (logical_plan/optimizer.rs)."""
        return self._with(("optimize",))

    def _build(self):
        qe = _qe()
        h = C.c_void_p()
        if self.path is not None:
            qe._check(qe.lib.qe_dataframe_parquet(self.ctx._h, self.path.encode(), C.byref(h)))
        else:
            names = (C.c_char_p * max(1, len(self.names)))(*[n.encode() for n in self.names])
            arr = (C.c_void_p * max(1, len(self.batches)))(*[b._h for b in self.batches])
            qe._check(qe.lib.qe_dataframe_from_batches(self.ctx._h, self.label.encode(), names, len(self.names), arr, len(self.batches), C.byref(h)))
        for step in self.steps:
            out = C.c_void_p()
            try:
                if step[0] == "project":
                    hs = _handles(step[1])
                    st = qe.lib.qe_dataframe_project(h, (C.c_void_p * max(1, len(hs)))(*hs), len(hs), C.byref(out))
                elif step[0] == "filter":
                    st = qe.lib.qe_dataframe_filter(h, step[1]._build(), C.byref(out))
                elif step[0] == "aggregate":
                    gs = _handles(step[1])
                    try:
                        ags = _handles(step[2])
                    except Exception:
                        for g in gs:
                            qe.lib.qe_logical_expr_free(g)
                        raise
                    st = qe.lib.qe_dataframe_aggregate(h, (C.c_void_p * max(1, len(gs)))(*gs), len(gs),
                                                       (C.c_void_p * max(1, len(ags)))(*ags), len(ags), C.byref(out))
                else:
                    st = qe.lib.qe_dataframe_optimize(h, C.byref(out))
            except Exception:
                qe.lib.qe_dataframe_free(h)
                raise
            qe._check(st)  # the builder consumed `h` either way
            h = out
        return h

    def _text(self, fn):
        qe = _qe()
        h = self._build()
        try:
            buf = C.create_string_buffer(1 << 16)
            qe._check(fn(h, buf, len(buf), None))
            return buf.value.decode()
        finally:
            qe.lib.qe_dataframe_free(h)

    def schema(self):
        """DataFrameTrait::schema: list of {"name", "data_type", "type", "is_nullable"}."""
        return json.loads(self._text(_qe().lib.qe_dataframe_schema))

    def logical_plan(self):
        """format_logical_plan of the plan built so far (logical_plan/mod.rs:48-58)."""
        return self._text(_qe().lib.qe_dataframe_logical_plan)

    def execute(self):
        """DataFrame::execute: plan -> physical plan -> all result batches."""
        qe = _qe()
        it = C.c_void_p()
        qe._check(qe.lib.qe_dataframe_execute(self._build(), C.byref(it)))  # consumes the handle
        return list(qe._BatchIterator(self.ctx, it))
