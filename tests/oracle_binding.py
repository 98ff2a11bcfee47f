"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# (STROM_ORACLE_LIBRARY: the sanitizer build of the same sources, scripts/cpu_suite_sanitized.sh)
LIB_PATH = os.environ.get("STROM_ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "liboracle.so")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
    return LIB_PATH


class oracle_layout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "sizeof_kern_data_store_head", "sizeof_kern_colmeta", "sizeof_kern_rowitem",
        "sizeof_kern_blkitem", "offsetof_resultbuf_results", "sizeof_kern_parambuf_head",
        "sizeof_kern_hashentry", "offsetof_hashentry_htup", "offsetof_htup_t_bits",
        "sizeof_kern_multihash_head", "offsetof_gpupreagg_kparams", "sizeof_kern_coldir")]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_oracle()
        lib = ctypes.CDLL(LIB_PATH)
        lib.oracle_gpuscan.restype = ctypes.c_int32
        lib.oracle_gpuscan.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.c_char_p, ctypes.c_size_t]
        lib.oracle_get_layout.argtypes = [ctypes.POINTER(oracle_layout)]
        lib.oracle_gpupreagg.restype = ctypes.c_int32
        lib.oracle_gpupreagg.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_uint32), ctypes.c_char_p,
                                         ctypes.c_size_t]
        _lib = lib
    return _lib


def datum_image(v):
    if v is None:
        return 0
    if isinstance(v, np.float32):
        return int(np.array([v], dtype=np.float32).view(np.uint32)[0])
    if isinstance(v, (float, np.floating)):
        return int(np.array([v], dtype=np.float64).view(np.uint64)[0])
    return int(v) & 0xFFFFFFFFFFFFFFFF


def ext_arrays(ext_params):
    n = len(ext_params)
    vals = np.zeros(max(n, 1), dtype=np.uint64)
    nulls = np.zeros(max(n, 1), dtype=np.uint8)
    keep = []
    for i, v in enumerate(ext_params):
        if isinstance(v, (bytes, str)):
            # text / character(n): the address of a varlena datum (kept alive with 'vals')
            import ctypes as _ct
            from pg_strom_amd.kds import varlena_datum
            keep.append(_ct.create_string_buffer(varlena_datum(v) + b"\0" * 8))
            vals[i] = _ct.addressof(keep[-1])
            continue
        vals[i] = datum_image(v)
        nulls[i] = 1 if v is None else 0
    _KEEPALIVE.append(keep)
    del _KEEPALIVE[:-64]
    return vals, nulls, n


_KEEPALIVE = []


def gpuscan(qual, kds_buf, ext_params=(), row_map=None, nitems=None):
    """returns (errcode, results int32 array) in ascending row order"""
    lib = load()
    vals, nulls, n = ext_arrays(ext_params)
    if nitems is None:
        nitems = int(np.frombuffer(kds_buf[20:24].tobytes(), dtype=np.uint32)[0])
    rm = None
    room = nitems
    if row_map is not None:
        rm = np.concatenate([np.array([len(row_map)], dtype=np.int32),
                             np.ascontiguousarray(row_map, dtype=np.int32)])
        room = len(row_map)
    results = np.zeros(max(room, 1), dtype=np.int32)
    cnt = ctypes.c_uint32(0)
    err = ctypes.create_string_buffer(256)
    rc = lib.oracle_gpuscan(qual.encode(), vals.ctypes.data, nulls.ctypes.data, n,
                            kds_buf.ctypes.data, rm.ctypes.data if rm is not None else None,
                            results.ctypes.data, ctypes.byref(cnt), err, 256)
    if rc == 101 and err.value:
        raise ValueError("oracle: " + err.value.decode())
    return rc, results[:cnt.value].copy()


def gpupreagg(spec, kds_buf, ntargets, ext_params=(), row_map=None, max_groups=1 << 16):
    """one chunk -> (status, values uint64 [ngroups, ntargets], isnull bool [...])"""
    lib = load()
    vals, nulls, n = ext_arrays(ext_params)
    rm = None
    if row_map is not None:
        rm = np.concatenate([np.array([len(row_map)], dtype=np.int32),
                             np.ascontiguousarray(row_map, dtype=np.int32)])
    out_v = np.zeros((max_groups, ntargets), dtype=np.uint64)
    out_n = np.zeros((max_groups, ntargets), dtype=np.uint8)
    cnt = ctypes.c_uint32(0)
    err = ctypes.create_string_buffer(256)
    rc = lib.oracle_gpupreagg(spec.encode(), vals.ctypes.data, nulls.ctypes.data, n,
                              kds_buf.ctypes.data, rm.ctypes.data if rm is not None else None,
                              max_groups, out_v.ctypes.data, out_n.ctypes.data,
                              ctypes.byref(cnt), err, 256)
    if rc == 101 and err.value:
        raise ValueError("oracle: " + err.value.decode())
    return rc, out_v[:cnt.value].copy(), out_n[:cnt.value].astype(bool)


def gpuhashjoin(spec, outer_buf, inner_bufs, ext_params=(), row_map=None, nrooms=None):
    """returns (errcode, nitems, records int32 [n, 1+ninner]) with the inner side
    identified by ROW INDEX in its inner chunk"""
    lib = load()
    lib.oracle_gpuhashjoin.restype = ctypes.c_int32
    lib.oracle_gpuhashjoin.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.c_char_p, ctypes.c_size_t]
    vals, nulls, n = ext_arrays(ext_params)
    rm = None
    if row_map is not None:
        rm = np.concatenate([np.array([len(row_map)], dtype=np.int32),
                             np.ascontiguousarray(row_map, dtype=np.int32)])
    ninner = len(inner_bufs)
    ptrs = (ctypes.c_void_p * ninner)(*[b.ctypes.data for b in inner_bufs])
    if nrooms is None:
        nrooms = 1 << 22
    results = np.zeros((nrooms, ninner + 1), dtype=np.int32)
    cnt = ctypes.c_uint32(0)
    err = ctypes.create_string_buffer(256)
    rc = lib.oracle_gpuhashjoin(spec.encode(), vals.ctypes.data, nulls.ctypes.data, n,
                                outer_buf.ctypes.data, rm.ctypes.data if rm is not None else None,
                                ptrs, ninner, results.ctypes.data, nrooms, ctypes.byref(cnt), err, 256)
    if rc == 101 and err.value:
        raise ValueError("oracle: " + err.value.decode())
    return rc, cnt.value, results[:min(cnt.value, nrooms)].copy()


def check_hashtable(kmhash_buf, depth, inner_buf, key_attnos, key_lens):
    lib = load()
    lib.oracle_check_hashtable.restype = ctypes.c_long
    lib.oracle_check_hashtable.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    a = np.array(key_attnos, dtype=np.int32)
    l = np.array(key_lens, dtype=np.int32)
    return lib.oracle_check_hashtable(kmhash_buf.ctypes.data, depth, inner_buf.ctypes.data,
                                      a.ctypes.data, l.ctypes.data, len(a))


def pg_crc32(data):
    lib = load()
    lib.oracle_pg_crc32.restype = ctypes.c_uint32
    lib.oracle_pg_crc32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
    return lib.oracle_pg_crc32(0xFFFFFFFF, data, len(data)) ^ 0xFFFFFFFF


def eval_rows(expr, kds_buf, ext_params=()):
    """(type oid, values uint64[n], isnull bool[n], errcode int32[n])"""
    lib = load()
    lib.oracle_eval_rows.restype = ctypes.c_int32
    lib.oracle_eval_rows.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_int32), ctypes.c_char_p, ctypes.c_size_t]
    vals, nulls, n = ext_arrays(ext_params)
    nitems = int(np.frombuffer(kds_buf[20:24].tobytes(), dtype=np.uint32)[0])
    ov = np.zeros(max(nitems, 1), dtype=np.uint64)
    on = np.zeros(max(nitems, 1), dtype=np.uint8)
    oe = np.zeros(max(nitems, 1), dtype=np.int32)
    oid = ctypes.c_int32(0)
    err = ctypes.create_string_buffer(256)
    rc = lib.oracle_eval_rows(expr.encode(), vals.ctypes.data, nulls.ctypes.data, n, kds_buf.ctypes.data,
                              ov.ctypes.data, on.ctypes.data, oe.ctypes.data, ctypes.byref(oid), err, 256)
    if rc != 0:
        raise ValueError("oracle: " + err.value.decode())
    return oid.value, ov[:nitems], on[:nitems].astype(bool), oe[:nitems]


def layout():
    out = oracle_layout()
    load().oracle_get_layout(ctypes.byref(out))
    return {n: getattr(out, n) for n, _ in oracle_layout._fields_}


def numeric_from_varlena(raw):
    """PostgreSQL varlena numeric bytes -> 64-bit image, or None when it
    cannot be carried (oracle_numeric_from_varlena)"""
    lib = load()
    lib.oracle_numeric_from_varlena.restype = ctypes.c_int
    lib.oracle_numeric_from_varlena.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    out = ctypes.c_uint64(0)
    ok = lib.oracle_numeric_from_varlena(bytes(raw) + b"\0" * 8, ctypes.byref(out))
    return int(out.value) if ok else None
