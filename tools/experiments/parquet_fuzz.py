"""Seeded random Parquet files (pyarrow writes them: column types, null patterns, value runs, compression,
dictionary use, page version and size, row-group size) scanned on the device against the oracle's decode;
also a heavier run of random damage (every scan fails cleanly or returns the file's row count)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import importlib

import pyarrow as pa
import pyarrow.parquet as pq

qe = importlib.import_module("query-engine_amd")
from test_gpu_parquet import scan_and_compare

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
damage = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = qe.Context(0)
r = np.random.default_rng(seed0)
bad = 0
tmp = tempfile.mkdtemp(prefix="pqfuzz")


def mask(n):
    k = r.integers(0, 5)
    if k == 0:
        return None
    if k == 1:
        return r.random(n) < 0.5
    if k == 2:
        return r.random(n) < 0.02
    if k == 3:
        m = np.zeros(n, bool)
        m[: n // 2] = True  # a long run of nulls
        return m
    return np.ones(n, bool) if n < 50 else r.random(n) < 0.97  # nearly all null


def runs(values, n):
    """Values with runs of random lengths (RLE in the dictionary indices)."""
    out = np.empty(n, dtype=values.dtype)
    i = 0
    while i < n:
        ln = int(r.choice([1, 1, 2, 9, 100, 1000]))
        out[i:i + ln] = values[r.integers(0, len(values))]
        i += ln
    return out


def column(kind, n):
    if kind == "i32":
        v = runs(r.integers(-2 ** 31, 2 ** 31, 50).astype(np.int32), n) if r.random() < 0.5 else r.integers(-2 ** 31, 2 ** 31, n).astype(np.int32)
        return pa.array(v, mask=mask(n))
    if kind == "i64":
        v = runs(r.integers(-2 ** 62, 2 ** 62, 20), n) if r.random() < 0.5 else r.integers(-2 ** 62, 2 ** 62, n)
        return pa.array(v, mask=mask(n))
    if kind == "f64":
        v = runs(r.standard_normal(30), n) if r.random() < 0.3 else r.standard_normal(n)
        return pa.array(v, mask=mask(n))
    if kind == "f32":
        return pa.array(r.standard_normal(n).astype(np.float32), mask=mask(n))
    if kind == "bool":
        v = runs(np.array([True, False]), n) if r.random() < 0.5 else r.random(n) < 0.5
        return pa.array(v, mask=mask(n))
    words = ["", "a", "bc", "Ünï", "w" * int(r.integers(1, 400))] + ["k%d" % i for i in range(int(r.integers(1, 3000)))]
    idx = runs(np.arange(len(words)), n) if r.random() < 0.5 else r.integers(0, len(words), n)
    m = mask(n)
    return pa.array([None if (m is not None and m[i]) else words[idx[i]] for i in range(n)], type=pa.string())


for it in range(rounds):
    n = int(r.choice([1, 2, 63, 1000, 20_000, 70_000]))
    kinds = [["i32", "i64", "f64", "f32", "bool", "str"][k] for k in r.integers(0, 6, int(r.integers(1, 8)))]
    tbl = pa.table({f"c{i}_{k}": column(k, n) for i, k in enumerate(kinds)})
    names = tbl.column_names
    opts = dict(compression=str(r.choice(["none", "snappy", "gzip", "zstd", "lz4"])),
                data_page_version=str(r.choice(["1.0", "2.0"])),
                data_page_size=int(r.choice([256, 4096, 65536, 1 << 20])),
                row_group_size=int(r.choice([50, 3000, 1 << 20])) if n > 50 else 1 << 20,
                use_dictionary=[True, False, [c for c in names if r.random() < 0.5]][r.integers(0, 3)],
                dictionary_pagesize_limit=int(r.choice([64, 4096, 1 << 20])))
    path = os.path.join(tmp, f"f{it}.parquet")
    what = f"it={it} n={n} kinds={kinds} opts={opts}"
    try:
        pq.write_table(tbl, path, **opts)
        proj = None if r.random() < 0.6 else [names[i] for i in r.permutation(len(names))[: int(r.integers(1, len(names) + 1))]]
        if opts["compression"] in ("zstd", "lz4"):  # codecs the oracle lacks: pyarrow's own read is the check
            got = list(qe.ParquetScanExec(ctx, path, proj).execute())
        else:
            got = scan_and_compare(qe, ctx, path, proj)
        # the reference's scan keeps the file's column order whatever order the projection names them in (data_source.rs:49-62)
        for j, name in enumerate([c for c in names if proj is None or c in proj]):
            vals = []
            for b in got:
                vals += b.columns()[j].to_pylist()
            want = tbl.column(name).to_pylist()
            assert len(vals) == len(want), (name, len(vals), len(want))
            for i, (a, b) in enumerate(zip(vals, want)):
                if isinstance(a, bytes):
                    a = a.decode()
                same = (a is None) == (b is None) and (a is None or a == b or (a != a and b != b))  # NaN never equals itself
                assert same, (name, i, a, b, proj)
        if damage:
            data = bytearray(open(path, "rb").read())
            rows = qe.parquet_schema(path)["rows"]
            for trial in range(damage):
                b2 = bytearray(data)
                for _ in range(int(r.integers(1, 4))):
                    pos = int(r.integers(4, max(5, len(b2) - 8)))
                    b2[pos] = int(r.integers(0, 256)) if trial % 2 else b2[pos] ^ (1 << int(r.integers(0, 8)))
                p2 = os.path.join(tmp, "bad.parquet")
                open(p2, "wb").write(bytes(b2))
                try:
                    got = list(qe.ParquetScanExec(ctx, p2).execute())
                    n_got = sum(b.num_rows for b in got)
                    # (damage to the footer's own row counts makes a different, equally consistent file)
                    assert n_got == rows or n_got == qe.parquet_schema(p2)["rows"], ("row count", n_got, rows, qe.parquet_schema(p2)["rows"])
                    for b in got:
                        b.columns()
                except qe.QeError as e:
                    assert e.name in ("ArrowError", "IoError", "PhysicalTypeNotSuported", "ExceedingBoundsError", "OutOfMemory"), e.name
    except BaseException as e:  # noqa: BLE001
        bad += 1
        print("FAIL", repr(e)[:500], what, flush=True)
        if bad > 5:
            sys.exit(1)
    finally:
        if os.path.exists(path):
            os.unlink(path)
    if it % 10 == 9:
        print("iteration", it + 1, "failures", bad, flush=True)
print("done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
