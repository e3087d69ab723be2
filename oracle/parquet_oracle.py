"""CPU oracle of the Parquet scan leaf — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py).

What is restated: the reference's scan, `ParquetDataSource::{new, schema, scan}`
(/root/reference/query-engine/src/data_source.rs:24-72) driven by `ScanExec`
(physical_plan/mod.rs:68-110): one chunk per row group, the projected columns in file order.
The decoding itself lives in the reference's dependency arrow2 0.10.1 (`io::parquet::read`,
on top of parquet2; neither is vendored or buildable here), so this file follows the published
Apache Parquet format (thrift compact footer, data pages v1/v2, PLAIN / dictionary / RLE-hybrid
encodings) for flat schemas, in plain Python + numpy.

PARITY STATUS: pinned by the reference's own fixture tests/userdata.parquet — its schema
(tests/mod.rs:31) and the cell values its tests assert (tests/mod.rs:34-188) — and cross-checked
against pyarrow's decode of the same file (tests/golden/userdata.npz).  Encodings and codecs the
fixture does not use (data page v2, snappy, gzip, required columns, booleans) are checked against
pyarrow-written files only: "parity unpinned" with respect to the reference.
"""
import struct
import zlib

import numpy as np

from . import oracle as O

TIMESTAMP_NS = 6
TYPE_NAME = dict(O._TYPE_NAME)
TYPE_NAME[TIMESTAMP_NS] = "Timestamp(Nanosecond, None)"

# parquet.thrift enums
BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY = range(8)
PLAIN, PLAIN_DICTIONARY, RLE, BIT_PACKED, RLE_DICTIONARY = 0, 2, 3, 4, 8
DATA_PAGE, DICTIONARY_PAGE, DATA_PAGE_V2 = 0, 2, 3
UNCOMPRESSED, SNAPPY, GZIP, ZSTD = 0, 1, 2, 6


class ParquetError(O.OracleError):
    pass


# ------------------------------------------------------------------ thrift compact protocol
class _Thrift:
    def __init__(self, buf, pos=0):
        self.b = buf
        self.p = pos

    def varint(self):
        r = s = 0
        while True:
            c = self.b[self.p]
            self.p += 1
            r |= (c & 0x7F) << s
            if not c & 0x80:
                return r
            s += 7

    def zigzag(self):
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def binary(self):
        n = self.varint()
        v = bytes(self.b[self.p:self.p + n])
        self.p += n
        return v

    def value(self, t):
        if t in (1, 2):  # bool true / false carried by the field header
            return t == 1
        if t == 3:
            v = self.b[self.p]
            self.p += 1
            return v
        if t in (4, 5, 6):
            return self.zigzag()
        if t == 7:
            v = struct.unpack_from("<d", self.b, self.p)[0]
            self.p += 8
            return v
        if t == 8:
            return self.binary()
        if t in (9, 10):
            h = self.b[self.p]
            self.p += 1
            n = h >> 4
            if n == 15:
                n = self.varint()
            et = h & 15
            if et in (1, 2):  # list<bool>: one byte each
                out = [self.b[self.p + i] == 1 for i in range(n)]
                self.p += n
                return out
            return [self.value(et) for _ in range(n)]
        if t == 12:
            return self.struct()
        raise ParquetError("ArrowError", f"thrift type {t} is not supported")

    def struct(self):
        """-> {field id: value}"""
        out = {}
        fid = 0
        while True:
            h = self.b[self.p]
            self.p += 1
            if h == 0:
                return out
            delta, t = h >> 4, h & 15
            fid = fid + delta if delta else self.zigzag()
            out[fid] = self.value(t)


# ------------------------------------------------------------------ block codecs
def snappy_decompress(src):
    t = _Thrift(src)
    n = t.varint()
    p = t.p
    out = bytearray()
    while p < len(src):
        tag = src[p]
        p += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[p:p + nb], "little")
                p += nb
            ln += 1
            out += src[p:p + ln]
            p += ln
            continue
        if kind == 1:
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | src[p]
            p += 1
        elif kind == 2:
            ln = 1 + (tag >> 2)
            off = src[p] | (src[p + 1] << 8)
            p += 2
        else:
            ln = 1 + (tag >> 2)
            off = int.from_bytes(src[p:p + 4], "little")
            p += 4
        if off == 0 or off > len(out):
            raise ParquetError("ArrowError", "corrupt snappy stream")
        for _ in range(ln):  # may overlap itself
            out.append(out[-off])
    if len(out) != n:
        raise ParquetError("ArrowError", "snappy length mismatch")
    return bytes(out)


def _decompress(codec, data, size):
    if codec == UNCOMPRESSED:
        return bytes(data)
    if codec == SNAPPY:
        return snappy_decompress(bytes(data))
    if codec == GZIP:
        return zlib.decompress(bytes(data), 15 + 32)
    raise ParquetError("ArrowError", f"compression codec {codec} is not supported by the oracle")


# ------------------------------------------------------------------ RLE / bit-packed hybrid
def rle_hybrid(buf, bit_width, count):
    """First `count` values of an RLE/bit-packing hybrid stream (LSB-first packing)."""
    out = np.zeros(count, np.int64)
    t = _Thrift(buf)
    n = 0
    byte_w = (bit_width + 7) // 8
    while n < count and t.p < len(buf):
        h = t.varint()
        if h & 1:
            groups = h >> 1
            nbytes = groups * bit_width
            bits = np.unpackbits(np.frombuffer(buf[t.p:t.p + nbytes], np.uint8), bitorder="little")
            t.p += nbytes
            vals = bits[:len(bits) // bit_width * bit_width].reshape(-1, bit_width).astype(np.int64) if bit_width else np.zeros((groups * 8, 0), np.int64)
            vals = (vals << np.arange(bit_width, dtype=np.int64)).sum(axis=1)
            take = min(count - n, len(vals))
            out[n:n + take] = vals[:take]
            n += take
        else:
            run = h >> 1
            v = int.from_bytes(buf[t.p:t.p + byte_w], "little")
            t.p += byte_w
            take = min(count - n, run)
            out[n:n + take] = v
            n += take
    if n < count:
        raise ParquetError("ArrowError", "RLE stream ends early")
    return out


# ------------------------------------------------------------------ metadata
class Column:
    def __init__(self, name, physical, converted, logical, optional):
        self.name, self.physical, self.optional = name, physical, optional
        self.type = None  # engine type, None = cannot be scanned
        if physical == BOOLEAN:
            self.type = O.BOOLEAN
        elif physical == INT32 and converted in (None, 17):  # none or INT_32
            self.type = O.INT32
        elif physical == INT64:
            nanos = logical is not None and 8 in logical and 3 in logical[8].get(2, {})
            self.type = TIMESTAMP_NS if nanos else O.INT64
        elif physical == INT96:
            self.type = TIMESTAMP_NS
        elif physical in (DOUBLE, FLOAT):  # float32 is widened: the engine's floating type is float64
            self.type = O.FLOAT64
        elif physical == BYTE_ARRAY and (converted == 0 or (logical is not None and 1 in logical)):
            self.type = O.UTF8


class ParquetFile:
    """ParquetDataSource::new (data_source.rs:30-41): open + read_metadata."""

    def __init__(self, path):
        try:
            with open(path, "rb") as f:
                self.data = f.read()
        except OSError as e:
            raise ParquetError("IoError", str(e))
        d = self.data
        if len(d) < 12 or d[:4] != b"PAR1" or d[-4:] != b"PAR1":
            raise ParquetError("ArrowError", "not a parquet file")
        n = struct.unpack("<I", d[-8:-4])[0]
        meta = _Thrift(d, len(d) - 8 - n).struct()
        schema = meta[2]
        self.num_rows = meta[3]
        root = schema[0]
        if root.get(5, 0) != len(schema) - 1:
            raise ParquetError("ArrowError", "nested schemas are not supported")
        self.columns = [Column(e[4].decode(), e.get(1), e.get(6), e.get(10), e.get(3, 0) == 1) for e in schema[1:]]
        for e in schema[1:]:
            if e.get(3, 0) == 2:
                raise ParquetError("ArrowError", "repeated fields are not supported")
        self.row_groups = meta[4]

    def schema(self):
        """ParquetDataSource::schema (data_source.rs:45-47): [(name, type name, nullable)]"""
        return [(c.name, TYPE_NAME.get(c.type, "Unsupported"), c.optional) for c in self.columns]

    # ---------------------------------------------------------------- scan
    def scan(self, projection=None):
        """ParquetDataSource::scan (data_source.rs:48-71): the columns whose names the projection
        holds, in file order; one Chunk per row group."""
        idx = [i for i, c in enumerate(self.columns) if projection is None or c.name in projection]
        for rg in self.row_groups:
            yield O.Chunk([self._column_chunk(rg[1][i][3], self.columns[i], rg[3]) for i in idx])

    def _column_chunk(self, md, col, n_rows):
        if col.type is None:
            raise ParquetError("PhysicalTypeNotSuported", f"column `{col.name}` has a type the engine cannot scan")
        codec = md[4]
        start = md.get(11) or md[9]
        if md[9] < start:
            start = md[9]
        pos, end = start, start + md[7]
        dictionary = None
        values, valid = [], []
        seen = 0
        while pos < end and seen < md[5]:
            t = _Thrift(self.data, pos)
            h = t.struct()
            body = self.data[t.p:t.p + h[3]]
            pos = t.p + h[3]
            if h[1] == DICTIONARY_PAGE:
                raw = _decompress(codec, body, h[2])
                dictionary = self._plain(col, raw, h[7][1])
                continue
            if h[1] == DATA_PAGE:
                raw = _decompress(codec, body, h[2])
                n, enc = h[5][1], h[5][2]
                p = 0
                if col.optional:
                    if h[5][3] != RLE:
                        raise ParquetError("ArrowError", "definition levels must be RLE encoded")
                    ln = struct.unpack_from("<I", raw, 0)[0]
                    levels = rle_hybrid(raw[4:4 + ln], 1, n)
                    p = 4 + ln
                else:
                    levels = np.ones(n, np.int64)
                body = raw[p:]
            elif h[1] == DATA_PAGE_V2:
                v2 = h[8]
                n, enc = v2[1], v2[4]
                dl, rl = v2[5], v2[6]
                if rl:
                    raise ParquetError("ArrowError", "repetition levels are not supported")
                levels = rle_hybrid(body[:dl], 1, n) if col.optional else np.ones(n, np.int64)
                rest = body[rl + dl:]
                body = _decompress(codec, rest, h[2] - rl - dl) if v2.get(7, True) else bytes(rest)
            else:
                continue  # index pages
            nn = int(levels.sum())
            if enc == PLAIN:
                vals = self._plain(col, body, nn)
            elif enc in (PLAIN_DICTIONARY, RLE_DICTIONARY):
                ids = rle_hybrid(body[1:], body[0], nn) if nn else np.zeros(0, np.int64)
                vals = [dictionary[i] for i in ids] if isinstance(dictionary, list) else dictionary[ids]
            elif enc == RLE and col.physical == BOOLEAN:  # length-prefixed hybrid stream of 1-bit values
                vals = rle_hybrid(body[4:], 1, nn).astype(np.bool_)
            else:
                raise ParquetError("ArrowError", f"encoding {enc} is not supported")
            values.append(vals)
            valid.append(levels.astype(np.bool_))
            seen += n
        valid = np.concatenate(valid) if valid else np.zeros(0, np.bool_)
        if len(valid) != n_rows:
            raise ParquetError("ArrowError", "row group and column chunk disagree about the number of rows")
        validity = valid if col.optional else None
        if col.type == O.UTF8:
            flat = [s for page in values for s in page]
            out, k = [], 0
            for ok in valid:
                out.append(flat[k] if ok else b"")
                k += int(ok)
            offs = np.zeros(len(out) + 1, np.int32)
            np.cumsum([len(s) for s in out], out=offs[1:])
            return O.Array(O.UTF8, (offs, np.frombuffer(b"".join(out), np.uint8).copy()), validity)
        dtype = {O.BOOLEAN: np.bool_, O.INT32: np.int32, O.INT64: np.int64, TIMESTAMP_NS: np.int64, O.FLOAT64: np.float64}[col.type]
        dense = np.concatenate(values) if values else np.zeros(0, dtype)
        full = np.zeros(n_rows, dtype)
        full[valid] = dense
        return O.Array(col.type, full, validity)

    @staticmethod
    def _plain(col, raw, n):
        if col.physical == BOOLEAN:
            return np.unpackbits(np.frombuffer(raw, np.uint8), bitorder="little")[:n].astype(np.bool_)
        if col.physical == INT32:
            return np.frombuffer(raw, "<i4", n).copy()
        if col.physical == INT64:
            return np.frombuffer(raw, "<i8", n).copy()
        if col.physical == DOUBLE:
            return np.frombuffer(raw, "<f8", n).copy()
        if col.physical == FLOAT:
            return np.frombuffer(raw, "<f4", n).astype(np.float64)
        if col.physical == INT96:  # nanoseconds of the day (u64) + Julian day (u32) -> ns since the epoch
            a = np.frombuffer(raw, np.uint8, n * 12).reshape(n, 12)
            nanos = a[:, :8].copy().view("<u8").reshape(n).astype(np.int64)
            days = a[:, 8:].copy().view("<u4").reshape(n).astype(np.int64)
            return (days - 2440588) * 86400000000000 + nanos
        out, p = [], 0
        for _ in range(n):
            ln = struct.unpack_from("<I", raw, p)[0]
            out.append(raw[p + 4:p + 4 + ln])
            p += 4 + ln
        return out
