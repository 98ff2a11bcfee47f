"""
Thin object layer over the C ABI of the device runtime (strom_hip.h).
Reference roles: pgstrom_get_devprog_key / lookup (opencl_devprog.c), the
pgstrom_data_store upload (datastore.c:837-973), device info SRF
(pg_strom--1.0.sql:9-82).
"""
import ctypes
import json

import numpy as np

from ._lib import lib, libc, strom_codegen_result, strom_perfmon

_initialized = False


class StromError(RuntimeError):
    def __init__(self, errcode, what=""):
        self.errcode = errcode
        RuntimeError.__init__(self, "%s: %s (%d)" % (
            what, lib.strom_strerror(errcode).decode(), errcode))


def init(device_ids=None):
    """strom_init(); idempotent.  device_ids=None -> HIP's current device"""
    global _initialized
    if _initialized:
        return
    if device_ids is None:
        rc = lib.strom_init(None, 0)
    else:
        arr = (ctypes.c_int * len(device_ids))(*device_ids)
        rc = lib.strom_init(arr, len(device_ids))
    if rc != 0:
        raise StromError(rc, "strom_init")
    _initialized = True


def shutdown():
    global _initialized
    if _initialized:
        lib.strom_shutdown()
        _initialized = False


def device_info(dindex=0):
    buf = ctypes.create_string_buffer(1024)
    if lib.strom_device_info(dindex, buf, 1024) < 0:
        raise StromError(100, "strom_device_info")
    return json.loads(buf.value.decode())


class Codegen(object):
    """result of one strom_codegen_* call"""

    def __init__(self, res):
        self._res = res
        self.source = ctypes.string_at(res.source).decode()
        self.extra_flags = res.extra_flags
        self.params = [(res.params[i].type_oid, bool(res.params[i].is_const),
                        res.params[i].param_id, bool(res.params[i].isnull))
                       for i in range(res.nparams)]
        self.vars = [(res.vars[i].attno, res.vars[i].type_oid) for i in range(res.nvars)]

    def parambuf(self, ext_values=(), ext_isnull=None):
        """kern_parambuf image (bytes) for these external Param datums"""
        n = len(ext_values)
        vals = (ctypes.c_uint64 * max(n, 1))()
        nulls = (ctypes.c_uint8 * max(n, 1))()
        keep = []
        for i, v in enumerate(ext_values):
            if isinstance(v, (bytes, str)):
                # text / character(n): the datum image is the ADDRESS of a varlena datum
                from .kds import varlena_datum
                keep.append(ctypes.create_string_buffer(varlena_datum(v) + b"\0" * 8))
                vals[i] = ctypes.addressof(keep[-1])
                continue
            vals[i] = datum_image(v)
            nulls[i] = 1 if (ext_isnull is not None and ext_isnull[i]) or v is None else 0
        p = lib.strom_create_kern_parambuf(ctypes.byref(self._res), vals, nulls, n)
        if not p:
            raise MemoryError("strom_create_kern_parambuf")
        length = ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32))[0]
        data = ctypes.string_at(p, length)
        libc.free(p)
        return data

    def __del__(self):
        try:
            lib.strom_codegen_release(ctypes.byref(self._res))
        except Exception:
            pass


def datum_image(v):
    """python scalar -> little-endian 64-bit datum image"""
    if v is None:
        return 0
    if isinstance(v, (float, np.floating)):
        if isinstance(v, np.float32):
            return int(np.array([v], dtype=np.float32).view(np.uint32)[0])
        return int(np.array([v], dtype=np.float64).view(np.uint64)[0])
    return int(v) & 0xFFFFFFFFFFFFFFFF


def codegen_gpuscan(qual):
    res = strom_codegen_result()
    rc = lib.strom_codegen_gpuscan(qual.encode(), ctypes.byref(res))
    if rc != 0:
        msg = ctypes.string_at(res.errmsg).decode() if res.errmsg else "?"
        lib.strom_codegen_release(ctypes.byref(res))
        raise ValueError("codegen: " + msg)
    return Codegen(res)


def expression_available(expr):
    err = ctypes.c_void_p()
    ok = lib.strom_codegen_available_expression(expr.encode(), ctypes.byref(err))
    msg = None
    if err.value:
        msg = ctypes.string_at(err.value).decode()
        libc.free(err)
    return bool(ok), msg


class DevProgram(object):
    """a device program handle (the reference's Datum dprog_key)"""

    def __init__(self, source, extra_flags):
        self.key = lib.strom_get_devprog_key(source.encode(), extra_flags)

    def wait(self):
        state = lib.strom_lookup_device_program(self.key, 1)
        if state != 1:
            raise StromError(-11, "device program build failed:\n" + self.errmsg())
        return self

    def state(self):
        return lib.strom_lookup_device_program(self.key, 0)

    def errmsg(self):
        return lib.strom_get_devprog_errmsg(self.key).decode()

    def release(self):
        lib.strom_put_devprog_key(self.key)


class DeviceStore(object):
    """a chunk resident in HBM (strom_dstore)"""

    def __init__(self, handle, nitems=None):
        if not handle:
            raise StromError(106, "strom_dstore")
        self.handle = handle
        self.nitems = nitems

    @classmethod
    def upload(cls, kds_buf, dindex=0):
        from .kds import KdsHead
        return cls(lib.strom_dstore_upload(kds_buf.ctypes.data, dindex), KdsHead(kds_buf).nitems)

    @classmethod
    def wrap(cls, devptr, length, nitems=None, dindex=0):
        return cls(lib.strom_dstore_wrap(devptr, length, dindex), nitems)

    @classmethod
    def from_torch_columns(cls, sqltypes, tensors, minmax=None, dindex=0):
        """a resident COLUMN chunk assembled in device memory from torch tensors that
        already live on the GPU (one per column, NULL-free): the head comes from
        strom_kds_column_head, the column arrays are device-to-device copies, the
        buffer is adopted with strom_dstore_wrap (kept alive by this object)"""
        import torch
        from .kds import column_head
        nrows = int(tensors[0].numel())
        head, total, voff = column_head(sqltypes, nrows, minmax)
        buf = torch.zeros(total, dtype=torch.uint8, device=tensors[0].device)
        buf[:len(head)] = torch.from_numpy(head.copy()).to(buf.device)
        for t, off in zip(tensors, voff):
            raw = t.contiguous().view(torch.uint8).reshape(-1)
            buf[off:off + raw.numel()] = raw
        torch.cuda.synchronize()
        ds = cls(lib.strom_dstore_wrap(buf.data_ptr(), total, dindex), nrows)
        ds._keepalive = buf
        return ds

    @property
    def devptr(self):
        return lib.strom_dstore_devptr(self.handle)

    @property
    def length(self):
        return lib.strom_dstore_length(self.handle)

    def to_column(self, type_oids=None):
        """transpose a resident ROW / ROW_FLAT / TUPSLOT chunk on the device
        (strom_dstore_to_column); returns (DeviceStore, kernel_ns)"""
        err = ctypes.c_int(0)
        ns = ctypes.c_uint64(0)
        oids, n = None, 0
        if type_oids is not None:
            n = len(type_oids)
            oids = (ctypes.c_int32 * n)(*[int(t) for t in type_oids])
        h = lib.strom_dstore_to_column(self.handle, oids, n, ctypes.byref(ns), ctypes.byref(err))
        if not h:
            raise StromError(err.value, "strom_dstore_to_column")
        return DeviceStore(h, self.nitems), ns.value

    def download(self):
        """copy the chunk image back to the host (tests, debugging)"""
        n = self.length
        out = np.zeros((n + 7) // 8, dtype=np.uint64).view(np.uint8)[:n]
        rc = lib.strom_dstore_download(self.handle, out.ctypes.data, n)
        if rc != 0:
            raise StromError(rc, "strom_dstore_download")
        return out

    def release(self):
        if self.handle:
            lib.strom_dstore_release(self.handle)
            self.handle = None


class DeviceRowMap(object):
    """row ids selected by a GpuScan, left in HBM as a kern_row_map
    (strom_rowmap): the hand-over between chained operators"""

    def __init__(self, handle):
        if not handle:
            raise StromError(106, "strom_rowmap")
        self.handle = handle

    @property
    def nvalids(self):
        return lib.strom_rowmap_nvalids(self.handle)

    def release(self):
        if self.handle:
            lib.strom_rowmap_release(self.handle)
            self.handle = None


def perfmon_dict(pfm):
    return {name: getattr(pfm, name) for name, _ in strom_perfmon._fields_}
