"""
GpuHashJoin operator, host side -- mirrors the executor half of gpuhashjoin.c:
    multihash_preload_khashtable (3614-3816)  build the kern_multihash
    gpuhashjoin_begin / load_next_chunk (2167-2704)
    pgstrom_create_gpuhashjoin (2461-2567)    result room = nitems x ratio x 1.1
    clserv_respond_hashjoin (4330-4425)       DataStoreNoSpace -> resize, retry
    gpuhashjoin_next_tuple (2706-2771)        (outer row, inner tuples)
"""
import ctypes

import numpy as np

from ._lib import lib, strom_perfmon, strom_codegen_result, strom_hashtable_input
from . import runtime
from .kds import KdsHead, aligned_buffer, stromalign, RESULTBUF_HEAD

STROM_RESULTS_ON_DEVICE = 0x0001
ERR_NOSPACE = 301


def codegen_gpuhashjoin(spec):
    res = strom_codegen_result()
    nrels = ctypes.c_int(0)
    rc = lib.strom_codegen_gpuhashjoin(spec.encode(), ctypes.byref(res), ctypes.byref(nrels))
    if rc != 0:
        msg = ctypes.string_at(res.errmsg).decode() if res.errmsg else "?"
        lib.strom_codegen_release(ctypes.byref(res))
        raise ValueError("codegen: " + msg)
    cg = runtime.Codegen(res)
    cg.nrels = nrels.value
    return cg


def build_multihash(inners):
    """inners: [(kds image, [key attnos])] -> kern_multihash image (uint8)"""
    arr = (strom_hashtable_input * len(inners))()
    for i, (buf, keys) in enumerate(inners):
        arr[i].inner = buf.ctypes.data
        arr[i].nkeys = len(keys)
        for k, a in enumerate(keys):
            arr[i].key_attnos[k] = a
    need = lib.strom_multihash_required_length(len(inners), arr)
    if need == 0:
        raise ValueError("strom_multihash_required_length: bad input")
    out = aligned_buffer(need, 64)
    rc = lib.strom_multihash_build(len(inners), arr, out.ctypes.data, need)
    if rc != 0:
        raise runtime.StromError(rc, "strom_multihash_build")
    return out


def entry_rowids(kmhash_buf, depth, offsets):
    """kern_hashentry byte offsets (inside the depth-th kern_hashtable) -> rowid"""
    ntables = int(np.frombuffer(kmhash_buf[1032:1036].tobytes(), dtype=np.uint32)[0])
    assert 1 <= depth <= ntables
    toff = int(np.frombuffer(kmhash_buf[1036 + 4 * (depth - 1):1040 + 4 * (depth - 1)].tobytes(),
                             dtype=np.uint32)[0])
    u32 = kmhash_buf.view(np.uint8)
    offs = np.asarray(offsets, dtype=np.int64) + toff + 8      # rowid at +8 of the entry
    idx = offs[:, None] + np.arange(4)[None, :]
    return u32[idx].copy().view(np.uint32).reshape(-1)


class HashJoinResult(object):
    def __init__(self, nitems, errcode, records, pfm):
        self.nitems = nitems
        self.errcode = errcode
        self.records = records        # int32 [nitems, nrels]: outer_row+1, entry offsets
        self.perfmon = pfm


class GpuHashJoin(object):
    def __init__(self, spec, row_population_ratio=1.0):
        self.spec = spec
        self.codegen = codegen_gpuhashjoin(spec)
        self.nrels = self.codegen.nrels + 1
        self.ratio = row_population_ratio
        self.program = None
        self.table = None
        self.kmhash = None

    def begin(self, kmhash_buf, ext_params=(), ext_isnull=None, dindex=0):
        runtime.init()
        self.program = runtime.DevProgram(self.codegen.source, self.codegen.extra_flags)
        self.parambuf = self.codegen.parambuf(ext_params, ext_isnull)
        err = ctypes.c_int(0)
        self.kmhash = kmhash_buf
        self.table = lib.strom_hashjoin_table_create(self.program.key, kmhash_buf.ctypes.data,
                                                     len(kmhash_buf), dindex, ctypes.byref(err))
        if not self.table:
            if err.value == -11:
                raise runtime.StromError(err.value, "GpuHashJoin kernel build:\n" + self.program.errmsg())
            raise runtime.StromError(err.value, "strom_hashjoin_table_create")
        return self

    def table_info(self, depth=1):
        mode, nslots, uniq, nent = ctypes.c_int(0), ctypes.c_uint32(0), ctypes.c_int(0), ctypes.c_uint32(0)
        lib.strom_hashjoin_table_info(self.table, depth, ctypes.byref(mode), ctypes.byref(nslots),
                                      ctypes.byref(uniq), ctypes.byref(nent))
        return {"mode": {1: "direct", 2: "keyed"}.get(mode.value, "hash"), "nslots": nslots.value,
                "unique": bool(uniq.value), "nentries": nent.value}

    def device_kmhash(self):
        """the re-linked device copy (rowid / tuples unchanged)"""
        out = aligned_buffer(len(self.kmhash), 64)
        rc = lib.strom_hashjoin_table_download(self.table, out.ctypes.data, len(out))
        if rc != 0:
            raise runtime.StromError(rc, "strom_hashjoin_table_download")
        return out

    def _make_khj(self, nrooms, host_results=True):
        plen = stromalign(len(self.parambuf))
        rlen = stromalign(RESULTBUF_HEAD + 4 * self.nrels * (nrooms if host_results else 0))
        buf = aligned_buffer(plen + rlen, 64)
        buf[:plen] = 0
        buf[:len(self.parambuf)] = np.frombuffer(self.parambuf, dtype=np.uint8)
        head = np.zeros(5, dtype=np.uint32)
        head[0] = self.nrels
        head[1] = nrooms
        buf[plen:plen + RESULTBUF_HEAD] = head.view(np.uint8)
        return buf, plen

    def submit(self, chunk, nrooms=None, row_map=None, flags=0):
        if isinstance(chunk, runtime.DeviceStore):
            nrows = chunk.nitems
            kds_host, kds_dev = None, chunk.handle
        else:
            nrows = KdsHead(chunk).nitems
            kds_host, kds_dev = chunk.ctypes.data, None
        rm = None
        if isinstance(row_map, runtime.DeviceRowMap):
            nrows = row_map.nvalids
            if nrooms is None:
                nrooms = int(nrows * self.ratio * 1.1) + 1
            khj, res_off = self._make_khj(nrooms, not (flags & STROM_RESULTS_ON_DEVICE))
            err = ctypes.c_int(0)
            task = lib.strom_submit_gpuhashjoin_mapped(self.table, khj.ctypes.data, kds_dev,
                                                       row_map.handle, flags, None, None,
                                                       ctypes.byref(err))
            if not task:
                raise runtime.StromError(err.value, "strom_submit_gpuhashjoin_mapped")
            return (task, khj, res_off, chunk, row_map, flags)
        if row_map is not None:
            r = np.ascontiguousarray(row_map, dtype=np.int32)
            rm = np.concatenate([np.array([len(r)], dtype=np.int32), r])
            nrows = len(r)
        if nrooms is None:
            nrooms = int(nrows * self.ratio * 1.1) + 1       # gpuhashjoin.c:2513-2514
        khj, res_off = self._make_khj(nrooms, not (flags & STROM_RESULTS_ON_DEVICE))
        err = ctypes.c_int(0)
        task = lib.strom_submit_gpuhashjoin(self.table, khj.ctypes.data, kds_host, kds_dev,
                                            rm.ctypes.data if rm is not None else None,
                                            flags, None, None, ctypes.byref(err))
        if not task:
            raise runtime.StromError(err.value, "strom_submit_gpuhashjoin")
        return (task, khj, res_off, chunk, rm, flags)

    def collect(self, pending):
        task, khj, res_off, chunk, rm, flags = pending
        pfm = strom_perfmon()
        rc = lib.strom_task_wait(task, ctypes.byref(pfm))
        head = np.frombuffer(khj[res_off:res_off + 20].tobytes(), dtype=np.int32)
        nitems = int(np.uint32(head[2]))
        if rc == ERR_NOSPACE:
            return HashJoinResult(nitems, rc, None, runtime.perfmon_dict(pfm))
        if rc != 0:
            raise runtime.StromError(rc, "GpuHashJoin")
        recs = np.zeros((0, self.nrels), dtype=np.int32)
        if not (flags & STROM_RESULTS_ON_DEVICE):
            start = res_off + RESULTBUF_HEAD
            recs = np.frombuffer(khj[start:start + 4 * nitems * self.nrels].tobytes(),
                                 dtype=np.int32).reshape(nitems, self.nrels)
        return HashJoinResult(nitems, 0, recs, runtime.perfmon_dict(pfm))

    def join_chunk(self, chunk, row_map=None, nrooms=None, flags=0):
        """one chunk, with the reference's resize-and-retry on DataStoreNoSpace"""
        res = self.collect(self.submit(chunk, nrooms=nrooms, row_map=row_map, flags=flags))
        if res.errcode == ERR_NOSPACE:
            res = self.collect(self.submit(chunk, nrooms=res.nitems, row_map=row_map, flags=flags))
            res.retried = True
        return res

    def join_chunk_project(self, chunk, dest_columns, nrooms=None):
        """join + kern_gpuhashjoin_projection_slot: dest_columns is a list of
        (depth, attno, sqltype) -- depth 0 is the outer chunk, d the d-th inner
        relation.  Returns (nitems, [(values, isnull)] per destination column)"""
        from .kds import SQL_TYPES, KDS_HEAD_FIXED
        nrows = chunk.nitems if isinstance(chunk, runtime.DeviceStore) else KdsHead(chunk).nitems
        if nrooms is None:
            nrooms = int(nrows * self.ratio * 1.1) + 1
        ncols = len(dest_columns)
        for attempt in range(2):
            head_len = (KDS_HEAD_FIXED + 8 * ncols + 15) & ~15
            stride = (9 * ncols + 7) & ~7
            dest = aligned_buffer(head_len + stride * nrooms, 64)
            dest[:head_len] = 0
            u32 = dest[:KDS_HEAD_FIXED].view(np.uint32)
            u32[2] = len(dest)
            u32[4] = ncols
            u32[6] = nrooms
            dest[36] = 3                                     # KDS_FORMAT_TUPSLOT
            for i, (_, _, typ) in enumerate(dest_columns):
                attlen = SQL_TYPES[typ][1]
                meta = dest[KDS_HEAD_FIXED + 8 * i: KDS_HEAD_FIXED + 8 * i + 8]
                meta[0] = 1
                meta[1] = attlen
                meta[2:4] = np.array([attlen], dtype=np.int16).view(np.uint8)
                meta[4:6] = np.array([i + 1], dtype=np.int16).view(np.uint8)
                meta[6:8] = np.array([-1], dtype=np.int16).view(np.uint8)
            depth = np.array([d for d, _, _ in dest_columns], dtype=np.int32)
            colidx = np.array([a - 1 for _, a, _ in dest_columns], dtype=np.int32)
            khj, res_off = self._make_khj(nrooms, True)
            if isinstance(chunk, runtime.DeviceStore):
                kds_host, kds_dev = None, chunk.handle
            else:
                kds_host, kds_dev = chunk.ctypes.data, None
            err = ctypes.c_int(0)
            task = lib.strom_submit_gpuhashjoin_projection(
                self.table, khj.ctypes.data, kds_host, kds_dev, None, dest.ctypes.data,
                depth.ctypes.data, colidx.ctypes.data, 0, None, None, ctypes.byref(err))
            if not task:
                raise runtime.StromError(err.value, "strom_submit_gpuhashjoin_projection")
            rc = lib.strom_task_wait(task, None)
            nitems = int(np.frombuffer(khj[res_off + 8:res_off + 12].tobytes(), dtype=np.uint32)[0])
            if rc == ERR_NOSPACE and attempt == 0:
                nrooms = nitems
                continue
            if rc != 0:
                raise runtime.StromError(rc, "GpuHashJoin projection")
            body = dest[head_len:head_len + stride * nitems].reshape(nitems, stride)
            values = body[:, :8 * ncols].copy().view(np.uint64).reshape(nitems, ncols)
            isnull = body[:, 8 * ncols:9 * ncols] != 0
            out = []
            for i, (_, _, typ) in enumerate(dest_columns):
                dt = np.dtype(SQL_TYPES[typ][2])
                raw = np.ascontiguousarray(values[:, i])
                v = raw.view(dt) if dt.itemsize == 8 else \
                    raw.view(np.uint8).reshape(-1, 8)[:, :dt.itemsize].copy().view(dt).reshape(-1)
                out.append((v, isnull[:, i]))
            return nitems, out

    def join_chunk_project_rows(self, chunk, dest_columns, nrooms=None, data_bytes=None):
        """join + kern_gpuhashjoin_projection_row (opencl_hashjoin.h:437-689): the joined rows
        as heap tuples in a ROW_FLAT kern_data_store (text / character(n) columns included).
        dest_columns as in join_chunk_project.  Returns (nitems, dest chunk (uint8 array),
        result records); DataStoreNoSpace is answered by a larger store, like
        gpuhashjoin.c:4330-4425."""
        from .kds import SQL_TYPES, KDS_HEAD_FIXED
        nrows = chunk.nitems if isinstance(chunk, runtime.DeviceStore) else KdsHead(chunk).nitems
        if nrooms is None:
            nrooms = int(nrows * self.ratio * 1.1) + 1
        ncols = len(dest_columns)
        if data_bytes is None:
            data_bytes = 64 * nrooms
        for attempt in range(8):
            head_len = (KDS_HEAD_FIXED + 8 * ncols + 15) & ~15
            items_len = (4 * nrooms + 15) & ~15
            dest = aligned_buffer(head_len + items_len + data_bytes, 64)
            dest[:head_len + items_len] = 0
            u32 = dest[:KDS_HEAD_FIXED].view(np.uint32)
            u32[2] = len(dest)
            u32[4] = ncols
            u32[6] = nrooms
            dest[36] = 2                                     # KDS_FORMAT_ROW_FLAT
            dest[40:44] = np.array([2249], dtype=np.uint32).view(np.uint8)     # tdtypeid RECORDOID
            dest[44:48] = np.array([-1], dtype=np.int32).view(np.uint8)        # tdtypmod
            for i, (_, _, typ) in enumerate(dest_columns):
                attlen = SQL_TYPES[typ][1]
                meta = dest[KDS_HEAD_FIXED + 8 * i: KDS_HEAD_FIXED + 8 * i + 8]
                meta[0] = 1 if attlen > 0 else 0
                meta[1] = attlen if attlen > 0 else 4
                meta[2:4] = np.array([attlen], dtype=np.int16).view(np.uint8)
                meta[4:6] = np.array([i + 1], dtype=np.int16).view(np.uint8)
                meta[6:8] = np.array([-1], dtype=np.int16).view(np.uint8)
            depth = np.array([d for d, _, _ in dest_columns], dtype=np.int32)
            colidx = np.array([a - 1 for _, a, _ in dest_columns], dtype=np.int32)
            khj, res_off = self._make_khj(nrooms, True)
            if isinstance(chunk, runtime.DeviceStore):
                kds_host, kds_dev = None, chunk.handle
            else:
                kds_host, kds_dev = chunk.ctypes.data, None
            err = ctypes.c_int(0)
            task = lib.strom_submit_gpuhashjoin_projection(
                self.table, khj.ctypes.data, kds_host, kds_dev, None, dest.ctypes.data,
                depth.ctypes.data, colidx.ctypes.data, 0, None, None, ctypes.byref(err))
            if not task:
                raise runtime.StromError(err.value, "strom_submit_gpuhashjoin_projection")
            rc = lib.strom_task_wait(task, None)
            nitems = int(np.frombuffer(khj[res_off + 8:res_off + 12].tobytes(), dtype=np.uint32)[0])
            if rc == ERR_NOSPACE and attempt < 7:
                if nitems > nrooms:
                    nrooms = nitems                          # more records than rooms
                else:
                    data_bytes *= 8                          # the tuples did not fit
                continue
            if rc != 0:
                raise runtime.StromError(rc, "GpuHashJoin projection (rows)")
            start = res_off + RESULTBUF_HEAD
            recs = np.frombuffer(khj[start:start + 4 * nitems * self.nrels].tobytes(),
                                 dtype=np.int32).reshape(nitems, self.nrels)
            return nitems, dest, recs

    def join_to_column(self, chunk, dest_columns, row_map=None, nrooms=None, zone_maps=True):
        """join a RESIDENT chunk and leave the joined rows in HBM as a COLUMN chunk
        for the next operator (strom_hashjoin_project_column): dest_columns as in
        join_chunk_project.  Returns (DeviceStore, nitems); the result pairs never
        cross PCIe.  zone_maps=False when the consumer brings its own key domain."""
        from .kds import SQL_TYPES
        ncols = len(dest_columns)
        depth = np.array([d for d, _, _ in dest_columns], dtype=np.int32)
        colidx = np.array([a - 1 for _, a, _ in dest_columns], dtype=np.int32)
        oids = np.array([SQL_TYPES[t][0] for _, _, t in dest_columns], dtype=np.int32)
        if not zone_maps:
            oids = -oids                                     # widths only: no min/max pass
        for attempt in range(2):
            pending = self.submit(chunk, nrooms=nrooms, row_map=row_map, flags=STROM_RESULTS_ON_DEVICE)
            err = ctypes.c_int(0)
            handle = lib.strom_hashjoin_project_column(pending[0], self.table, chunk.handle, ncols,
                                                       depth.ctypes.data, colidx.ctypes.data,
                                                       oids.ctypes.data, ctypes.byref(err))
            res = self.collect(pending)
            if not handle and err.value == ERR_NOSPACE and attempt == 0:
                nrooms = res.nitems                          # resize and retry (gpuhashjoin.c:4330-4425)
                continue
            if not handle:
                raise runtime.StromError(err.value, "strom_hashjoin_project_column")
            return runtime.DeviceStore(handle, res.nitems), res.nitems

    def end(self):
        if self.table:
            lib.strom_hashjoin_table_release(self.table)
            self.table = None
        if self.program is not None:
            self.program.release()
            self.program = None
