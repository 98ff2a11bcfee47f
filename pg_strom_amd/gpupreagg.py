"""
GpuPreAgg operator, host side -- mirrors the executor half of gpupreagg.c:
    gpupreagg_begin (2189-2327)      codegen, program key, parambuf
    gpupreagg_load_next_outer / pgstrom_create_gpupreagg (2329-2498)
                                     one request per chunk
    gpupreagg_next_tuple (2609-2663) partial rows out (TUPSLOT)
and the partial -> final merge the reference leaves to PostgreSQL's Agg node
with the pgstrom.* final aggregates (gpupreagg.c:4430-4773,
pg_strom--1.0.sql:247-401): see finalize().

Aggregates are rewritten to partial functions exactly as
gpupreagg_rewrite_expr does (gpupreagg.c:134-333, 729-1166):
    count(*)      -> sum(nrows())
    count(x)      -> sum(nrows(x is not null))
    sum(int2/4)   -> sum(psum(x::int8))            (int8 result)
    avg(int2/4/8) -> psum(x::int8) / nrows(x is not null)
    avg/sum(float)-> psum(x::float8) [/ nrows]
    min/max       -> min/max(pmin/pmax(x))
    stddev/var    -> (nrows, psum(x), psum_x2(x))  float only
"""
import ctypes

import numpy as np

from ._lib import lib, strom_perfmon, strom_codegen_result, strom_preagg_target, \
    strom_preagg_domain
from . import runtime
from .kds import KdsHead, aligned_buffer, KDS_HEAD_FIXED

KIND_KEY, KIND_NROWS, KIND_PSUM, KIND_PMIN, KIND_PMAX = 1, 2, 3, 4, 5
FLOAT_OIDS = (700, 701)
TYPE_DTYPES = {16: np.int8, 21: np.int16, 23: np.int32, 20: np.int64, 700: np.float32,
               701: np.float64, 1082: np.int32, 1083: np.int64, 1114: np.int64,
               1042: np.int8, 1700: np.uint64}


def codegen_gpupreagg(spec):
    res = strom_codegen_result()
    targets = (strom_preagg_target * 64)()
    n = ctypes.c_int(0)
    rc = lib.strom_codegen_gpupreagg(spec.encode(), ctypes.byref(res), targets, 64, ctypes.byref(n))
    if rc != 0:
        msg = ctypes.string_at(res.errmsg).decode() if res.errmsg else "?"
        lib.strom_codegen_release(ctypes.byref(res))
        raise ValueError("codegen: " + msg)
    cg = runtime.Codegen(res)
    cg.targets = [(targets[i].kind, targets[i].type_oid) for i in range(n.value)]
    cg._targets_c = targets
    return cg


def domain_of(chunks, key_columns):
    """dense group domain (min, range per key) from the zone maps of COLUMN
    chunks; chunks are uint8 kds images, key_columns 0-based column numbers"""
    mins, maxs = [None] * len(key_columns), [None] * len(key_columns)
    for buf in chunks:
        head = KdsHead(buf)
        assert head.format == 4, "zone maps exist in KDS_FORMAT_COLUMN only"
        off = (KDS_HEAD_FIXED + 8 * head.ncols + 15) & ~15
        for i, col in enumerate(key_columns):
            cd = buf[off + 32 * col: off + 32 * col + 32]
            flags = int(np.frombuffer(cd[12:16].tobytes(), dtype=np.uint32)[0])
            if not flags & 1:
                continue                      # all NULL in this chunk
            mn, mx = np.frombuffer(cd[16:32].tobytes(), dtype=np.int64)
            mins[i] = mn if mins[i] is None else min(mins[i], mn)
            maxs[i] = mx if maxs[i] is None else max(maxs[i], mx)
    dom = []
    for mn, mx in zip(mins, maxs):
        dom.append((0, 0) if mn is None else (int(mn), int(mx - mn + 1)))
    return dom


class PartialRows(object):
    """decoded TUPSLOT result: one partial row per group"""

    def __init__(self, targets, values, isnull):
        self.targets = targets
        self.values = values        # uint64 [nrows, ncols] raw datum images
        self.isnull = isnull        # bool   [nrows, ncols]

    def __len__(self):
        return self.values.shape[0]

    def column(self, resno):
        kind, oid = self.targets[resno]
        raw = np.ascontiguousarray(self.values[:, resno])
        if kind == KIND_NROWS:
            return raw.view(np.int64), self.isnull[:, resno]
        dt = np.dtype(TYPE_DTYPES[oid])
        if dt.itemsize == 8:
            return raw.view(dt), self.isnull[:, resno]
        return raw.view(np.uint8).reshape(-1, 8)[:, :dt.itemsize].copy().view(dt).reshape(-1), \
            self.isnull[:, resno]


class GpuPreAgg(object):
    def __init__(self, spec):
        self.spec = spec
        self.codegen = codegen_gpupreagg(spec)
        self.targets = self.codegen.targets
        self.program = None
        self.session = None

    def begin(self, domain, ext_params=(), ext_isnull=None, dindex=0):
        """domain: [(min, range)] per key in target order (see domain_of)"""
        runtime.init()
        self.program = runtime.DevProgram(self.codegen.source, self.codegen.extra_flags)
        self.parambuf = self.codegen.parambuf(ext_params, ext_isnull)
        dom = strom_preagg_domain()
        dom.nkeys = len(domain)
        for i, (mn, rng) in enumerate(domain):
            dom.key_min[i] = mn
            dom.key_range[i] = rng
        err = ctypes.c_int(0)
        pb = ctypes.create_string_buffer(self.parambuf, len(self.parambuf))
        self.session = lib.strom_gpupreagg_create(self.program.key, self.codegen._targets_c,
                                                  len(self.targets), pb, ctypes.byref(dom),
                                                  dindex, ctypes.byref(err))
        if not self.session:
            raise runtime.StromError(err.value, "strom_gpupreagg_create")
        return self

    def begin_hashed(self, ext_params=(), ext_isnull=None, ngroups_hint=0, dindex=0):
        """GROUP BY over keys of any type / spread: groups live in a hash table in
        HBM (strom_gpupreagg_create_hashed); same submit / fetch calls afterwards"""
        runtime.init()
        # the hashed kernels are a program of their own (strom_gpupreagg.h builds either
        # family); asking for it directly spares the build of the dense one
        self.program = runtime.DevProgram("#define GPUPREAGG_HASHED 1\n" + self.codegen.source,
                                          self.codegen.extra_flags)
        self.parambuf = self.codegen.parambuf(ext_params, ext_isnull)
        err = ctypes.c_int(0)
        pb = ctypes.create_string_buffer(self.parambuf, len(self.parambuf))
        self.session = lib.strom_gpupreagg_create_hashed(self.program.key, self.codegen._targets_c,
                                                         len(self.targets), pb, int(ngroups_hint),
                                                         dindex, ctypes.byref(err))
        if not self.session:
            raise runtime.StromError(err.value, "strom_gpupreagg_create_hashed")
        return self

    def num_groups(self):
        return lib.strom_gpupreagg_num_groups(self.session)

    def checked_folds(self):
        """requests whose integer sums the range proof did not cover: folded again by the
        GPUPREAGG_CHECKED program (strom_gpupreagg_checked_folds)"""
        return lib.strom_gpupreagg_checked_folds(self.session)

    # group-slot agreement ---------------------------------------------------
    def census(self, chunk, row_map=None):
        """mark the dense ids that occur in 'chunk' (after the qual); returns
        the accumulated bitmap (uint32 words, one bit per dense id)"""
        if isinstance(chunk, runtime.DeviceStore):
            kds_host, kds_dev = None, chunk.handle
        else:
            kds_host, kds_dev = chunk.ctypes.data, None
        rm = None
        if row_map is not None:
            r = np.ascontiguousarray(row_map, dtype=np.int32)
            rm = np.concatenate([np.array([len(r)], dtype=np.int32), r])
        nwords = (lib.strom_gpupreagg_dense_groups(self.session) + 31) // 32
        bitmap = np.zeros(nwords, dtype=np.uint32)
        rc = lib.strom_gpupreagg_census(self.session, kds_host, kds_dev,
                                        rm.ctypes.data if rm is not None else None,
                                        bitmap.ctypes.data, nwords)
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_census")
        return bitmap

    def compact(self, bitmap=None):
        """map the ids marked in 'bitmap' (default: this session's census) to
        consecutive table slots; before the first fold"""
        if bitmap is not None:
            bitmap = np.ascontiguousarray(bitmap, dtype=np.uint32)
        rc = lib.strom_gpupreagg_compact(self.session,
                                         bitmap.ctypes.data if bitmap is not None else None,
                                         len(bitmap) if bitmap is not None else 0)
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_compact")
        return lib.strom_gpupreagg_num_groups(self.session)

    # requests -------------------------------------------------------------
    def submit(self, chunk, row_map=None):
        if isinstance(chunk, runtime.DeviceStore):
            kds_host, kds_dev = None, chunk.handle
        else:
            kds_host, kds_dev = chunk.ctypes.data, None
        rm = None
        if isinstance(row_map, runtime.DeviceRowMap):
            err = ctypes.c_int(0)
            task = lib.strom_submit_gpupreagg_mapped(self.session, kds_dev, row_map.handle,
                                                     None, None, ctypes.byref(err))
            if not task:
                raise runtime.StromError(err.value, "strom_submit_gpupreagg_mapped")
            return (task, chunk, row_map)
        if row_map is not None:
            r = np.ascontiguousarray(row_map, dtype=np.int32)
            rm = np.concatenate([np.array([len(r)], dtype=np.int32), r])
        err = ctypes.c_int(0)
        task = lib.strom_submit_gpupreagg(self.session, kds_host, kds_dev,
                                          rm.ctypes.data if rm is not None else None,
                                          None, None, ctypes.byref(err))
        if not task:
            raise runtime.StromError(err.value, "strom_submit_gpupreagg")
        return (task, chunk, rm)

    def submit_joined(self, join, join_pending, chunk, columns):
        """fold the rows of a join straight from its result pairs
        (strom_submit_gpupreagg_joined): join_pending = join.submit(chunk, flags=
        STROM_RESULTS_ON_DEVICE) not yet collected; columns as in join_to_column.
        The join's device results pass to this request."""
        from .kds import SQL_TYPES
        depth = np.array([d for d, _, _ in columns], dtype=np.int32)
        colidx = np.array([a - 1 for _, a, _ in columns], dtype=np.int32)
        oids = np.array([SQL_TYPES[t][0] for _, _, t in columns], dtype=np.int32)
        err = ctypes.c_int(0)
        task = lib.strom_submit_gpupreagg_joined(self.session, join_pending[0], join.table, chunk.handle,
                                                 len(columns), depth.ctypes.data, colidx.ctypes.data,
                                                 oids.ctypes.data, None, None, ctypes.byref(err))
        if not task:
            raise runtime.StromError(err.value, "strom_submit_gpupreagg_joined")
        return (task, chunk, (depth, colidx, oids))

    def submit_lookup(self, join, chunk, columns):
        """fact JOIN dim GROUP BY in one pass (strom_submit_gpupreagg_lookup): the join
        is a lookup in the aggregate's own pass over the resident COLUMN chunk; 'join' only
        lends its hash table.  columns as in join_to_column."""
        from .kds import SQL_TYPES
        depth = np.array([d for d, _, _ in columns], dtype=np.int32)
        colidx = np.array([a - 1 for _, a, _ in columns], dtype=np.int32)
        oids = np.array([SQL_TYPES[t][0] for _, _, t in columns], dtype=np.int32)
        err = ctypes.c_int(0)
        task = lib.strom_submit_gpupreagg_lookup(self.session, join.table, chunk.handle,
                                                 len(columns), depth.ctypes.data, colidx.ctypes.data,
                                                 oids.ctypes.data, None, None, ctypes.byref(err))
        if not task:
            raise runtime.StromError(err.value, "strom_submit_gpupreagg_lookup")
        return (task, chunk, (depth, colidx, oids))

    def collect(self, pending):
        """returns (status, perfmon): status 0 folded, 2 CpuReCheck (not folded)"""
        pfm = strom_perfmon()
        rc = lib.strom_task_wait(pending[0], ctypes.byref(pfm))
        if rc == -11:
            raise runtime.StromError(rc, "GpuPreAgg kernel build:\n" + self.program.errmsg())
        if rc not in (0, 2):
            raise runtime.StromError(rc, "GpuPreAgg")
        return rc, runtime.perfmon_dict(pfm)

    def fold(self, chunk, row_map=None):
        return self.collect(self.submit(chunk, row_map))

    # results ---------------------------------------------------------------
    def _decode_tupslot(self, buf, n):
        ncols = len(self.targets)
        head = (KDS_HEAD_FIXED + 8 * ncols + 15) & ~15
        stride = (9 * ncols + 7) & ~7
        body = np.frombuffer(buf[head:head + stride * n].tobytes(), dtype=np.uint8).reshape(n, stride)
        values = body[:, :8 * ncols].copy().view(np.uint64).reshape(n, ncols)
        isnull = body[:, 8 * ncols:9 * ncols] != 0
        return PartialRows(self.targets, values, isnull)

    def fetch(self):
        need = lib.strom_gpupreagg_fetch(self.session, None, 0)
        if need < 0:
            raise runtime.StromError(-need, "strom_gpupreagg_fetch")
        buf = aligned_buffer(need, 64)
        n = lib.strom_gpupreagg_fetch(self.session, buf.ctypes.data, need)
        if n < 0:
            raise runtime.StromError(-n, "strom_gpupreagg_fetch")
        return self._decode_tupslot(buf, n)

    # the reference's per-chunk message ---------------------------------------
    def _begin_program(self, ext_params, ext_isnull):
        runtime.init()
        if self.program is None:
            self.program = runtime.DevProgram(self.codegen.source, self.codegen.extra_flags)
        return self.codegen.parambuf(ext_params, ext_isnull)

    def chunk_domain(self, chunk, row_map=None, ext_params=(), ext_isnull=None, dindex=0):
        """[(min, range)] per key of ONE chunk, after the qual (strom_gpupreagg_chunk_domain)"""
        parambuf = self._begin_program(ext_params, ext_isnull)
        pb = ctypes.create_string_buffer(parambuf, len(parambuf))
        if isinstance(chunk, runtime.DeviceStore):
            kds_host, kds_dev = None, chunk.handle
        else:
            kds_host, kds_dev = chunk.ctypes.data, None
        rm = None
        if row_map is not None:
            r = np.ascontiguousarray(row_map, dtype=np.int32)
            rm = np.concatenate([np.array([len(r)], dtype=np.int32), r])
        dom = strom_preagg_domain()
        rc = lib.strom_gpupreagg_chunk_domain(self.program.key, self.codegen._targets_c, len(self.targets),
                                              pb, kds_host, kds_dev,
                                              rm.ctypes.data if rm is not None else None,
                                              dindex, ctypes.byref(dom))
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_chunk_domain")
        return [(int(dom.key_min[i]), int(dom.key_range[i])) for i in range(dom.nkeys)]

    def submit_chunk(self, chunk, row_map=None, ext_params=(), ext_isnull=None, dest_rooms=None,
                     num_groups=0.0, done=None, dindex=0):
        """One pgstrom_gpupreagg message (opencl_gpupreagg.h:994-1003; built by
        pgstrom_create_gpupreagg, gpupreagg.c:2329-2498): the kern_gpupreagg image
        {status, sortbuf_len, kparams, krowmap} and a TUPSLOT kds_dest the caller sizes
        (dest_rooms rows; default: the chunk's row count, what the reference allocates).
        Returns a pending request for collect_chunk()."""
        parambuf = self._begin_program(ext_params, ext_isnull)
        if isinstance(chunk, runtime.DeviceStore):
            kds_host, kds_dev = None, chunk.handle
            nitems = chunk.nitems if hasattr(chunk, "nitems") else None
        else:
            kds_host, kds_dev = chunk.ctypes.data, None
            nitems = KdsHead(chunk).nitems
        r = (np.ascontiguousarray(row_map, dtype=np.int32) if row_map is not None
             else np.zeros(0, dtype=np.int32))
        nvalids = len(r) if row_map is not None else -1
        # KERN_GPUPREAGG_KROWMAP: STROMALIGN(offsetof(kparams) + kparams.length)
        map_off = (16 + len(parambuf) + 15) & ~15
        kg = aligned_buffer(map_off + 4 + 4 * len(r) + 16, 64)
        kg[:] = 0
        kg[16:16 + len(parambuf)] = np.frombuffer(parambuf, dtype=np.uint8)
        kg[map_off:map_off + 4] = np.array([nvalids], dtype=np.int32).view(np.uint8)
        if len(r):
            kg[map_off + 4:map_off + 4 + 4 * len(r)] = r.view(np.uint8)
        ncols = len(self.targets)
        if dest_rooms is None:
            dest_rooms = max(1, nvalids if nvalids >= 0 else (nitems or 1))
        head = (KDS_HEAD_FIXED + 8 * ncols + 15) & ~15
        stride = (9 * ncols + 7) & ~7
        dest = aligned_buffer(head + stride * dest_rooms, 64)
        dest[:head] = 0
        err = ctypes.c_int(0)
        nkeys = sum(1 for k, _ in self.targets if k == KIND_KEY)
        task = lib.strom_submit_gpupreagg_chunk(self.program.key, self.codegen._targets_c, ncols,
                                                kg.ctypes.data, kds_host, kds_dev,
                                                dest.ctypes.data, len(dest),
                                                1 if nkeys else 0, float(num_groups), dindex,
                                                done, None, ctypes.byref(err))
        if not task:
            raise runtime.StromError(err.value, "strom_submit_gpupreagg_chunk")
        return (task, chunk, kg, dest, r)

    def collect_chunk(self, pending):
        """(status, PartialRows or None): status 0 -> the chunk's partial rows, 2 CpuReCheck"""
        task, _, kg, dest, _ = pending
        pfm = strom_perfmon()
        rc = lib.strom_task_wait(task, ctypes.byref(pfm))
        status = int(kg[0:4].view(np.int32)[0])
        assert status == rc, (status, rc)
        if rc == -11:
            raise runtime.StromError(rc, "GpuPreAgg kernel build:\n" + self.program.errmsg())
        if rc == 2:
            return rc, None
        if rc != 0:
            raise runtime.StromError(rc, "GpuPreAgg (chunk message)")
        n = KdsHead(dest).nitems
        return rc, self._decode_tupslot(dest, n)

    def table_tensor_info(self):
        """(device pointer, nbytes, ngroups) of the resident table -- what an
        RCCL all-reduce operates on"""
        return (lib.strom_gpupreagg_table_devptr(self.session),
                lib.strom_gpupreagg_table_length(self.session),
                lib.strom_gpupreagg_num_groups(self.session))

    def table_layout(self, resno):
        b, v = ctypes.c_size_t(0), ctypes.c_size_t(0)
        rc = lib.strom_gpupreagg_table_layout(self.session, resno, ctypes.byref(b), ctypes.byref(v))
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_table_layout")
        return b.value, v.value

    # multi-GPU ------------------------------------------------------------
    def bind_torch_table(self):
        """move the resident table into a torch uint8 tensor (same device) so
        that torch.distributed can reduce it in place; returns the tensor"""
        import torch
        nbytes = lib.strom_gpupreagg_table_length(self.session)
        t = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        rc = lib.strom_gpupreagg_bind_table(self.session, t.data_ptr())
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_bind_table")
        self._table_tensor = t
        return t

    def agree_group_slots(self, chunks, group=None, device=None):
        """multi-GPU planning step: census of the local chunks, union over
        the ranks, compact -- all ranks end with identical table slots"""
        from . import parallel
        bitmap = None
        for c in chunks:
            bitmap = self.census(c)
        if bitmap is None:
            nwords = (lib.strom_gpupreagg_dense_groups(self.session) + 31) // 32
            bitmap = np.zeros(nwords, dtype=np.uint32)
        merged = parallel.allreduce_census(bitmap, group, device)
        return self.compact(merged)

    def census_allreduce(self, comm):
        """union of the ranks' census bitmaps over RCCL inside the C library
        (strom_gpupreagg_census_allreduce); then compact() with no argument"""
        rc = lib.strom_gpupreagg_census_allreduce(self.session, comm.handle, None)
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_census_allreduce")

    def allreduce_rccl(self, comm):
        """merge the per-GPU tables: strom_gpupreagg_allreduce (csrc/parallel.cpp), ordered
        behind the folds on the session's own stream; returns when the merge is done"""
        rc = lib.strom_gpupreagg_allreduce(self.session, comm.handle, None)
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_allreduce")

    def merge_from(self, other):
        """hashed sessions of one device: other's groups are added to this session's table
        (strom_gpupreagg_merge); other is left as it is"""
        rc = lib.strom_gpupreagg_merge(self.session, other.session)
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_merge")

    def reduce_scatter_rccl(self, comm):
        """hashed sessions: the hash-partitioned exchange without the final all-gather -- this rank
        ends up with the merged groups it owns (strom_gpupreagg_reduce_scatter)"""
        rc = lib.strom_gpupreagg_reduce_scatter(self.session, comm.handle, None)
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_reduce_scatter")

    @staticmethod
    def exchange_local(sessions, gather_after=False):
        """the multi-GPU exchange of hashed sessions among sessions of ONE device, session i as
        rank i (strom_gpupreagg_exchange_local)"""
        arr = (ctypes.c_void_p * len(sessions))(*[s.session for s in sessions])
        rc = lib.strom_gpupreagg_exchange_local(arr, len(sessions), 1 if gather_after else 0)
        if rc != 0:
            raise runtime.StromError(rc, "strom_gpupreagg_exchange_local")

    def allreduce(self, group=None):
        """the same merge stated with torch.distributed collectives (pg_strom_amd.parallel):
        the gloo rehearsal's path and the GPU cross-check of allreduce_rccl()"""
        import torch
        from . import parallel
        if getattr(self, "_table_tensor", None) is None:
            self.bind_torch_table()
        lib.strom_synchronize()                 # every fold has landed in the table
        layout = parallel.TableLayout(self.targets, lib.strom_gpupreagg_num_groups(self.session))
        assert layout.nbytes == self._table_tensor.numel()
        parallel.allreduce_table(self._table_tensor, layout, group)
        torch.cuda.synchronize()

    def reset(self):
        lib.strom_gpupreagg_reset(self.session)

    def end(self):
        if self.session:
            lib.strom_gpupreagg_release(self.session)
            self.session = None
        if self.program is not None:
            self.program.release()
            self.program = None
