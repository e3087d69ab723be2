"""Shared helpers for the parity tests: the same host data goes to the oracle (numpy, CPU) and to
the HIP backend (device batch), and results are compared bit for bit unless a tolerance is given."""
import json
import os

import numpy as np

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

USERDATA_COLUMNS = ["registration_dttm", "id", "first_name", "last_name", "email", "gender", "ip_address", "cc",
                    "country", "birthdate", "salary", "title", "comments"]


def load_userdata():
    """Returns (list of oracle Arrays in file order, expected-json).  registration_dttm
    (Timestamp) is carried as Int64 nanoseconds."""
    z = np.load(os.path.join(GOLDEN, "userdata.npz"), allow_pickle=False)
    with open(os.path.join(GOLDEN, "userdata_expected.json")) as f:
        exp = json.load(f)
    arrays = []
    for name in USERDATA_COLUMNS:
        valid = z[name + ".valid"]
        validity = None if valid.all() else valid
        if name + ".offsets" in z:
            arrays.append(orc.Array(orc.UTF8, (z[name + ".offsets"], z[name + ".bytes"]), validity))
        else:
            v = z[name + ".values"]
            t = {np.dtype(np.int32): orc.INT32, np.dtype(np.int64): orc.INT64, np.dtype(np.float64): orc.FLOAT64}[v.dtype]
            arrays.append(orc.Array(t, v, validity))
    return arrays, exp


def to_gpu_column(qe, a):
    """oracle Array -> product Column description."""
    if a.type == orc.UTF8:
        return qe.Column(qe.TYPE_UTF8, a.strings(), a.validity)
    return qe.Column(a.type, a.values, a.validity)


def gpu_batch(qe, ctx, arrays):
    return qe.Batch.from_numpy(ctx, [to_gpu_column(qe, orc.Array.from_python(a)) for a in arrays])


def assert_column_equal(got, want, rtol=None, what=""):
    """got: product Column (host copy); want: oracle Array.  Null slots are compared by validity only."""
    assert got.type == want.type, f"{what}: type {got.type} != {want.type}"
    n = len(want)
    assert len(got.values) == n, f"{what}: length {len(got.values)} != {n}"
    wv = np.ones(n, np.bool_) if want.validity is None else want.validity
    gv = np.ones(n, np.bool_) if got.validity is None else got.validity
    assert np.array_equal(gv, wv), f"{what}: validity differs at {np.nonzero(gv != wv)[0][:10]}"
    if want.type == orc.UTF8:
        ws = want.strings()
        for i in range(n):
            if wv[i]:
                assert got.values[i] == ws[i], f"{what}: row {i}: {got.values[i]!r} != {ws[i]!r}"
        return
    g = np.asarray(got.values)[wv]
    w = np.asarray(want.values)[wv]
    if rtol is None:
        if want.type == orc.FLOAT64:
            ok = g.view(np.uint64) == w.view(np.uint64)
            # NaN payloads are not part of the contract
            ok |= np.isnan(g) & np.isnan(w)
            assert ok.all(), f"{what}: float64 bits differ at {np.nonzero(~ok)[0][:10]}: {g[~ok][:5]} vs {w[~ok][:5]}"
        else:
            assert np.array_equal(g, w), f"{what}: values differ at {np.nonzero(g != w)[0][:10]}"
    else:
        assert np.allclose(g, w, rtol=rtol, atol=0), f"{what}: beyond rtol {rtol}"


def assert_batch_equal(got_batch, want_chunk, rtol=None, what=""):
    cols = got_batch.columns()
    assert len(cols) == len(want_chunk.arrays), f"{what}: {len(cols)} columns != {len(want_chunk.arrays)}"
    assert got_batch.num_rows == len(want_chunk), f"{what}: rows {got_batch.num_rows} != {len(want_chunk)}"
    for i, (g, w) in enumerate(zip(cols, want_chunk.arrays)):
        assert_column_equal(g, w, rtol=rtol, what=f"{what} column {i}")


def sort_by_first_columns(arrays_or_cols, n_keys):
    """Group output order is unspecified (HashMap iteration in the reference): sort rows by the key
    columns before comparing.  Accepts oracle Arrays or product Columns; returns row permutation."""
    keys = []
    for k in reversed(range(n_keys)):
        c = arrays_or_cols[k]
        keys.append(np.asarray(c.values).astype(np.int64))
    if not keys:
        return np.arange(len(arrays_or_cols[0].values) if arrays_or_cols else 0)
    return np.lexsort(keys)
