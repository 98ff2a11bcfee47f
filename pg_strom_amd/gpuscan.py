"""
GpuScan operator, host side -- mirrors the executor half of gpuscan.c:
    gpuscan_begin      (754-853)   codegen + program key + parambuf
    pgstrom_load/create_gpuscan (883-997)  one request per chunk
    pgstrom_fetch_gpuscan (1065-1163)      async window of chunks in flight
    gpuscan_next_tuple (999-1056)  positive ids pass, negative ids are
                                   re-checked on the CPU by the caller
    gpuscan_end        (1448-1500)
The planner half (cost model, path hooks) is PostgreSQL glue and is not
mirrored.  All device work goes through libstrom_hip.so.
"""
import collections
import ctypes

import numpy as np

from ._lib import lib, strom_perfmon
from . import runtime
from .kds import KdsHead, make_kern_gpuscan, read_resultbuf

STROM_RESULTS_ON_DEVICE = 0x0001


class GpuScanResult(object):
    def __init__(self, nitems, errcode, results, pfm):
        self.nitems = nitems
        self.errcode = errcode
        self.results = results          # int32: +(row+1) pass, -(row+1) recheck
        self.perfmon = pfm

    def passed_rows(self):
        """0-based ids of rows whose qual is TRUE on the device"""
        r = self.results
        return np.sort(r[r > 0] - 1)

    def recheck_rows(self):
        """0-based ids of rows the host must re-evaluate (CpuReCheck)"""
        r = self.results
        return np.sort(-r[r < 0] - 1)


class GpuScan(object):
    """
    scan = GpuScan("(and (int4lt (var 1 int4) (param 0 int4)) ...)")
    scan.begin(ext_params=[k, c])
    for res in scan.scan_chunks(chunks): ...
    scan.end()
    """

    def __init__(self, qual, max_async_chunks=3):
        self.qual = qual
        self.max_async_chunks = max_async_chunks   # pg_strom.max_async_chunks
        self.codegen = runtime.codegen_gpuscan(qual)
        self.program = None
        self.parambuf = None

    def begin(self, ext_params=(), ext_isnull=None):
        runtime.init()
        self.program = runtime.DevProgram(self.codegen.source, self.codegen.extra_flags)
        self.parambuf = self.codegen.parambuf(ext_params, ext_isnull)
        return self

    def kernel_source(self):
        return self.codegen.source

    # one request --------------------------------------------------------
    def submit(self, chunk, nitems=None, row_map=None, flags=0):
        """chunk: uint8 kds image (host) or runtime.DeviceStore (resident)"""
        if isinstance(chunk, runtime.DeviceStore):
            nrows = chunk.nitems if nitems is None else nitems
            kds_host, kds_dev = None, chunk.handle
        else:
            nrows = KdsHead(chunk).nitems
            kds_host, kds_dev = chunk.ctypes.data, None
        rowmap_buf = None
        nrooms = nrows
        if isinstance(row_map, runtime.DeviceRowMap):
            nrooms = row_map.nvalids
            kgs, res_off = make_kern_gpuscan(self.parambuf, max(nrooms, 1),
                                             host_results=not (flags & STROM_RESULTS_ON_DEVICE))
            err = ctypes.c_int(0)
            task = lib.strom_submit_gpuscan_mapped(self.program.key, kgs.ctypes.data, kds_dev,
                                                   row_map.handle, flags, None, None, ctypes.byref(err))
            if not task:
                raise runtime.StromError(err.value, "strom_submit_gpuscan_mapped")
            return (task, kgs, res_off, chunk, row_map, bool(flags & STROM_RESULTS_ON_DEVICE))
        if row_map is not None:
            rm = np.ascontiguousarray(row_map, dtype=np.int32)
            rowmap_buf = np.concatenate([np.array([len(rm)], dtype=np.int32), rm])
            nrooms = len(rm)
        kgs, res_off = make_kern_gpuscan(self.parambuf, max(nrooms, 1),
                                         host_results=not (flags & STROM_RESULTS_ON_DEVICE))
        err = ctypes.c_int(0)
        task = lib.strom_submit_gpuscan(self.program.key, kgs.ctypes.data, kds_host, kds_dev,
                                        rowmap_buf.ctypes.data if rowmap_buf is not None else None,
                                        flags, None, None, ctypes.byref(err))
        if not task:
            raise runtime.StromError(err.value, "strom_submit_gpuscan")
        return (task, kgs, res_off, chunk, rowmap_buf, bool(flags & STROM_RESULTS_ON_DEVICE))

    def collect(self, pending):
        task, kgs, res_off, _chunk, _rm, on_device = pending
        pfm = strom_perfmon()
        rc = lib.strom_task_wait(task, ctypes.byref(pfm))
        if rc != 0:
            if rc == -11:
                raise runtime.StromError(rc, "GpuScan kernel build:\n" + self.program.errmsg())
            raise runtime.StromError(rc, "GpuScan")
        if on_device:                    # head only (results stayed on the device)
            head = np.frombuffer(kgs[res_off:res_off + 20].tobytes(), dtype=np.int32)
            nitems, errcode, results = int(head[2]), int(head[3]), np.zeros(0, dtype=np.int32)
        else:
            nitems, errcode, results = read_resultbuf(kgs, res_off)
        return GpuScanResult(nitems, errcode, results, runtime.perfmon_dict(pfm))

    def scan_chunk(self, chunk, **kw):
        return self.collect(self.submit(chunk, **kw))

    def scan_to_rowmap(self, chunk, row_map=None):
        """scan a resident chunk and keep the selected row ids in HBM as a
        kern_row_map for the next operator (strom_rowmap_from_task).
        Returns (DeviceRowMap, GpuScanResult); raises StromError(2) when the
        chunk has rows to re-check -- the caller then takes the host path."""
        pending = self.submit(chunk, row_map=row_map, flags=STROM_RESULTS_ON_DEVICE)
        err = ctypes.c_int(0)
        handle = lib.strom_rowmap_from_task(pending[0], ctypes.byref(err))
        res = self.collect(pending)              # the scan task is still waited for
        if not handle:
            raise runtime.StromError(err.value, "strom_rowmap_from_task")
        return runtime.DeviceRowMap(handle), res

    def scan_chunks(self, chunks, **kw):
        """keeps up to max_async_chunks requests in flight, yields in order"""
        window = collections.deque()
        for chunk in chunks:
            window.append(self.submit(chunk, **kw))
            if len(window) >= self.max_async_chunks:
                yield self.collect(window.popleft())
        while window:
            yield self.collect(window.popleft())

    def end(self):
        if self.program is not None:
            self.program.release()
            self.program = None
