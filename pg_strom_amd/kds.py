"""
Host-side helpers around the chunk formats of include/strom_kds.h.

These wrap the C builders of libstrom_hip.so (strom_datastore.h) for numpy
callers and decode the wire structs the operators exchange
(kern_data_store head, kern_gpuscan = {kern_parambuf, kern_resultbuf}).
Reference: datastore.c:312-529 (builders), opencl_common.h:335-486 (structs),
opencl_gpuscan.h:63-90 (kern_gpuscan packing).
"""
import ctypes

import numpy as np

from ._lib import lib, strom_column_input

KDS_FORMAT_ROW = 1
KDS_FORMAT_ROW_FLAT = 2
KDS_FORMAT_TUPSLOT = 3
KDS_FORMAT_COLUMN = 4
FORMAT_NAMES = {"row": 1, "row_flat": 2, "tupslot": 3, "column": 4}

STROMALIGN_LEN = 16
KDS_HEAD_FIXED = 48            # offsetof(kern_data_store, colmeta)
RESULTBUF_HEAD = 20            # offsetof(kern_resultbuf, results)

# name -> (pg_type oid, attlen, numpy dtype)
SQL_TYPES = {
    "bool": (16, 1, np.int8),
    "int2": (21, 2, np.int16),
    "int4": (23, 4, np.int32),
    "int8": (20, 8, np.int64),
    "float4": (700, 4, np.float32),
    "float8": (701, 8, np.float64),
    "date": (1082, 4, np.int32),
    "time": (1083, 8, np.int64),
    "timestamp": (1114, 8, np.int64),
    "numeric": (1700, 8, np.uint64),   # 64-bit device form
    # numeric(p,s) as int8 at 10^-s ("decimal64"; COLUMN chunks; IR: (var N decimal S)):
    # 'values' are the scaled integers
    "decimal": (0x10000 | 1700, 8, np.int64),
    # the same values laid out as PostgreSQL's varlena numeric inside heap
    # tuples (ROW / ROW_FLAT only); 'values' are the 64-bit images
    "numeric_varlena": (1700, -1, np.uint64),
    "char1": (1042, 1, np.int8),
    # heap formats only: complete varlena datums (bytes objects), copied verbatim -- numerics of
    # any magnitude, also those the 64-bit device form cannot hold
    "numeric_raw": (1700, -1, object),
    # text / character(n) as PostgreSQL stores them (heap formats only): 'values' are the
    # payloads (bytes or str); a short 1-byte header up to 126 bytes, a 4-byte header beyond
    "text": (25, -1, object),
    "character": (1042, -1, object),
    # complete varlena datums copied verbatim (e.g. a compressed or external one)
    "text_raw": (25, -1, object),
}

VARLENA_RAW_TYPES = ("numeric_raw", "text", "character", "text_raw")


def varlena_datum(payload):
    """payload bytes -> the varlena datum PostgreSQL would store (short header when it fits)"""
    if isinstance(payload, str):
        payload = payload.encode()
    payload = bytes(payload)
    if len(payload) + 1 <= 127:
        return bytes([((len(payload) + 1) << 1) | 1]) + payload
    return np.array([(len(payload) + 4) << 2], dtype="<u4").tobytes() + payload


def stromalign(n):
    return (n + STROMALIGN_LEN - 1) & ~(STROMALIGN_LEN - 1)


def aligned_buffer(nbytes, align=256):
    """uint8 array of nbytes whose data pointer is 'align'-aligned"""
    raw = np.empty(nbytes + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    buf = raw[off:off + nbytes]
    assert buf.ctypes.data % align == 0
    return buf


class Column(object):
    """one input column: SQL type name, values, optional null mask"""

    def __init__(self, sqltype, values, isnull=None):
        oid, attlen, dtype = SQL_TYPES[sqltype]
        self.sqltype = sqltype
        self.type_oid = oid
        self.attlen = attlen
        if sqltype in VARLENA_RAW_TYPES:
            # keep the datums and an array of their addresses alive with the column
            if sqltype in ("text", "character"):
                values = [varlena_datum(v if v is not None else b"") for v in values]
            self._datums = [ctypes.create_string_buffer(bytes(v) + b"\0" * 8) for v in values]
            self.values = np.array([ctypes.addressof(b) for b in self._datums], dtype=np.uint64)
        else:
            self.values = np.ascontiguousarray(values, dtype=dtype)
        self.isnull = None
        if isnull is not None:
            self.isnull = np.ascontiguousarray(isnull, dtype=np.uint8)
            assert self.isnull.shape == self.values.shape

    def __len__(self):
        return len(self.values)


def _column_inputs(columns):
    arr = (strom_column_input * len(columns))()
    for i, c in enumerate(columns):
        arr[i].type_oid = c.type_oid
        arr[i].attlen = c.attlen
        arr[i].attalign = c.attlen if c.attlen > 0 else (-1 if c.sqltype in VARLENA_RAW_TYPES else 4)
        arr[i].attbyval = 1 if c.attlen > 0 else 0
        arr[i].values = c.values.ctypes.data
        arr[i].isnull = c.isnull.ctypes.data if c.isnull is not None else None
    return arr


def build_kds(fmt, columns):
    """lay 'columns' out as one chunk of format 'fmt'; returns uint8 array"""
    if isinstance(fmt, str):
        fmt = FORMAT_NAMES[fmt]
    nrows = len(columns[0])
    assert all(len(c) == nrows for c in columns)
    inputs = _column_inputs(columns)
    need = lib.strom_kds_required_length(fmt, len(columns), inputs, nrows)
    if need == 0:
        raise ValueError("strom_kds_required_length: bad column description")
    buf = aligned_buffer(need)
    rc = lib.strom_kds_build(fmt, len(columns), inputs, nrows, buf.ctypes.data, need)
    if rc != 0:
        raise RuntimeError("strom_kds_build failed: %s" % lib.strom_strerror(rc).decode())
    return buf


def column_head(sqltypes, nrows, minmax=None):
    """head of a NULL-free KDS_FORMAT_COLUMN chunk whose column arrays are filled
    elsewhere (strom_kds_column_head): returns (head uint8 array, chunk length,
    values_off per column).  minmax: [(min, max)] per column -- ints, or floats for
    float4/float8 columns; None for a column without one -- or None for "no zone maps"."""
    ncols = len(sqltypes)
    arr = (strom_column_input * ncols)()
    for i, t in enumerate(sqltypes):
        oid, attlen, _ = SQL_TYPES[t]
        arr[i].type_oid, arr[i].attlen, arr[i].attalign, arr[i].attbyval = oid, attlen, attlen, 1
    mm = None
    if minmax is not None:
        mm = np.zeros(2 * ncols, dtype=np.int64)
        for i, (t, lohi) in enumerate(zip(sqltypes, minmax)):
            if lohi is None:
                mm[2 * i:2 * i + 2] = [1, 0]            # min > max: no zone map for this column
                continue
            lo, hi = lohi
            if SQL_TYPES[t][0] in (700, 701):
                mm[2 * i:2 * i + 2] = np.array([lo, hi], dtype=np.float64).view(np.int64)
            else:
                mm[2 * i:2 * i + 2] = [int(lo), int(hi)]
    head = aligned_buffer(4096 + 64 * ncols)
    voff = np.zeros(ncols, dtype=np.uint32)
    total = lib.strom_kds_column_head(ncols, arr, nrows, mm.ctypes.data if mm is not None else None,
                                      head.ctypes.data, len(head), voff.ctypes.data)
    if total == 0:
        raise ValueError("strom_kds_column_head: bad column description or chunk over 4 GB")
    return head[:int(voff[0])], int(total), [int(v) for v in voff]


def kds_to_column(kds_buf):
    need = lib.strom_kds_to_column(kds_buf.ctypes.data, None, 0)
    if need == 0:
        raise ValueError("chunk cannot be converted to COLUMN format")
    out = aligned_buffer(need)
    got = lib.strom_kds_to_column(kds_buf.ctypes.data, out.ctypes.data, need)
    assert got == need
    return out


class KdsHead(object):
    """decoded fixed head of a kern_data_store image"""

    def __init__(self, buf):
        u32 = np.frombuffer(buf[:KDS_HEAD_FIXED].tobytes(), dtype=np.uint32)
        self.length = int(u32[2])
        self.usage = int(u32[3])
        self.ncols = int(u32[4])
        self.nitems = int(u32[5])
        self.nrooms = int(u32[6])
        self.nblocks = int(u32[7])
        self.maxblocks = int(u32[8])
        self.format = int(np.frombuffer(buf[36:37].tobytes(), dtype=np.int8)[0])
        meta = np.frombuffer(buf[KDS_HEAD_FIXED:KDS_HEAD_FIXED + 8 * self.ncols].tobytes(),
                             dtype=np.dtype([("attbyval", "i1"), ("attalign", "i1"),
                                             ("attlen", "<i2"), ("attnum", "<i2"),
                                             ("attcacheoff", "<i2")]))
        self.colmeta = meta


def kds_fetch(kds_buf, row, col):
    """(isnull, raw 64-bit datum image) of one cell, host side"""
    v = ctypes.c_uint64(0)
    isnull = lib.strom_kds_fetch(kds_buf.ctypes.data, row, col, ctypes.byref(v))
    return bool(isnull), v.value


# ---------------------------------------------------------------------
# kern_gpuscan = kern_parambuf followed by kern_resultbuf
# ---------------------------------------------------------------------
def make_kern_gpuscan(parambuf_bytes, nrooms, nrels=1, host_results=True):
    """host image of kern_gpuscan.  host_results=False: results[] stay on the
    device (STROM_RESULTS_ON_DEVICE), only the head is backed by host memory"""
    plen = stromalign(len(parambuf_bytes))
    rlen = stromalign(RESULTBUF_HEAD + 4 * nrels * (nrooms if host_results else 0))
    buf = aligned_buffer(plen + rlen, 64)
    buf[:plen] = 0
    buf[:len(parambuf_bytes)] = np.frombuffer(parambuf_bytes, dtype=np.uint8)
    head = np.zeros(5, dtype=np.uint32)
    head[0] = nrels
    head[1] = nrooms
    buf[plen:plen + RESULTBUF_HEAD] = head.view(np.uint8)
    return buf, plen


def read_resultbuf(buf, res_offset):
    head = np.frombuffer(buf[res_offset:res_offset + RESULTBUF_HEAD].tobytes(), dtype=np.int32)
    nrels, nrooms, nitems, errcode = int(head[0]), int(head[1]), int(head[2]), int(head[3])
    n = nitems * nrels
    start = res_offset + RESULTBUF_HEAD
    results = np.frombuffer(buf[start:start + 4 * n].tobytes(), dtype=np.int32)
    return nitems, errcode, results


# ---------------------------------------------------------------------
# 64-bit device NUMERIC (opencl_numeric.h:122-162): exponent 63..58 (signed,
# base 10), sign 57, mantissa 56..0; normalised (no trailing decimal zero)
# ---------------------------------------------------------------------
def numeric_encode(value):
    """Decimal / str / int -> uint64 image, or None when it does not fit"""
    from decimal import Decimal
    d = Decimal(value)
    sign, digits, exp = d.as_tuple()
    mant = int("".join(map(str, digits))) if digits else 0
    if mant == 0:
        return 0
    while mant % 10 == 0:
        mant //= 10
        exp += 1
    while exp > 31 and mant < (1 << 57) // 10:
        mant *= 10
        exp -= 1
    if mant >= (1 << 57) or exp < -32 or exp > 31:
        return None
    return ((exp & 0x3f) << 58) | (sign << 57) | mant


def decimal_type(scale):
    """type tag for strom_dstore_to_column(): turn this numeric column into a decimal column
    (int8 at 10^-scale) -- STROM_DECIMAL_TYPE(scale) of strom_kds.h"""
    return (0x10000 | 1700) | (int(scale) << 20)


def numeric_decode(image):
    """uint64 image -> Decimal"""
    from decimal import Decimal
    image = int(image)
    exp = image >> 58
    if exp >= 32:
        exp -= 64
    mant = image & ((1 << 57) - 1)
    d = Decimal(mant).scaleb(exp)
    return -d if (image >> 57) & 1 else d


def numeric_column(strings, isnull=None):
    """decimal strings -> Column('numeric'); values that do not fit the
    64-bit form raise (a real ingest would leave such rows to the CPU)"""
    vals = np.zeros(len(strings), dtype=np.uint64)
    for i, s in enumerate(strings):
        if isnull is not None and isnull[i]:
            continue
        img = numeric_encode(s)
        if img is None:
            raise ValueError("numeric %r does not fit the 64-bit device form" % (s,))
        vals[i] = img
    return Column("numeric", vals, isnull)


def decode_column_chunk(buf):
    """KDS_FORMAT_COLUMN image -> list of dicts(values, notnull, stat_flags,
    minval, maxval) -- raw little-endian integers of attlen bytes"""
    head = KdsHead(buf)
    assert head.format == FORMAT_NAMES["column"]
    coldir_off = stromalign(KDS_HEAD_FIXED + 8 * head.ncols)     # KDS_HEAD_LENGTH
    cd = np.frombuffer(buf[coldir_off:coldir_off + 32 * head.ncols].tobytes(),
                       dtype=np.dtype([("values_off", "<u4"), ("nulls_off", "<u4"),
                                       ("extra_off", "<u4"), ("stat_flags", "<u4"),
                                       ("minval", "<i8"), ("maxval", "<i8")]))
    out = []
    n = head.nitems
    for c in range(head.ncols):
        attlen = int(head.colmeta[c]["attlen"])
        if attlen < 0:
            attlen = 8              # a varlena column: 8-byte offsets of the datums (strom_kds.h)
        voff = int(cd[c]["values_off"])
        vals = np.frombuffer(buf[voff:voff + attlen * n].tobytes(), dtype="<i%d" % attlen)
        notnull = None
        if cd[c]["nulls_off"]:
            noff = int(cd[c]["nulls_off"])
            words = np.frombuffer(buf[noff:noff + 4 * ((n + 31) // 32)].tobytes(), dtype="<u4")
            bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
            notnull = bits.astype(bool)
        out.append(dict(values=vals, notnull=notnull, stat_flags=int(cd[c]["stat_flags"]),
                        minval=int(cd[c]["minval"]), maxval=int(cd[c]["maxval"]),
                        extra_off=int(cd[c]["extra_off"])))
    return out


def numeric_from_scaled(values, scale, isnull=None):
    """vectorised numeric column: integer array 'values' at 10^-scale
    (value = values * 10**-scale) -> Column('numeric') in the normalised
    64-bit form (trailing decimal zeros moved into the exponent)"""
    v = np.asarray(values, dtype=np.int64)
    sign = (v < 0).astype(np.uint64)
    mant = np.abs(v).astype(np.uint64)
    exp = np.full(v.shape, -int(scale), dtype=np.int64)
    for _ in range(19):
        strip = (mant != 0) & (mant % np.uint64(10) == 0)
        if not strip.any():
            break
        mant = np.where(strip, mant // np.uint64(10), mant)
        exp = exp + strip.astype(np.int64)
    if (mant >= np.uint64(1 << 57)).any() or (exp < -32).any() or (exp > 31).any():
        raise ValueError("numeric value does not fit the 64-bit device form")
    img = ((exp & 0x3f).astype(np.uint64) << np.uint64(58)) | (sign << np.uint64(57)) | mant
    img = np.where(mant == 0, np.uint64(0), img)
    return Column("numeric", img, isnull)
