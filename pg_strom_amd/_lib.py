"""
ctypes binding of libstrom_hip.so (include/strom_hip.h, strom_codegen.h,
strom_datastore.h).  The library is the product; this module only loads it
and declares prototypes.  There is no Python or CPU fallback: if the
library is missing or a symbol cannot be resolved the import fails.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (STROM_HIP_LIBRARY: another build of the same library -- the sanitizer build of
# scripts/cpu_suite_sanitized.sh; there is still no fallback: the named file must load)
LIB_PATH = os.environ.get("STROM_HIP_LIBRARY") or os.path.join(_HERE, "libstrom_hip.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int32 = ctypes.c_int32
c_uint32 = ctypes.c_uint32
c_uint64 = ctypes.c_uint64
c_size_t = ctypes.c_size_t
c_char_p = ctypes.c_char_p


class strom_perfmon(ctypes.Structure):
    _fields_ = [
        ("enabled", ctypes.c_int8),
        ("num_samples", c_uint32),
        ("time_inner_load", c_uint64),
        ("time_outer_load", c_uint64),
        ("time_materialize", c_uint64),
        ("time_in_sendq", c_uint64),
        ("time_in_recvq", c_uint64),
        ("time_kern_build", c_uint64),
        ("num_dma_send", c_uint32),
        ("num_dma_recv", c_uint32),
        ("bytes_dma_send", c_uint64),
        ("bytes_dma_recv", c_uint64),
        ("time_dma_send", c_uint64),
        ("time_dma_recv", c_uint64),
        ("num_kern_exec", c_uint32),
        ("time_kern_exec", c_uint64),
        ("num_kern_proj", c_uint32),
        ("time_kern_proj", c_uint64),
        ("num_kern_prep", c_uint32),
        ("num_kern_sort", c_uint32),
        ("time_kern_prep", c_uint64),
        ("time_kern_sort", c_uint64),
        ("time_kern_exec_ns", c_uint64),
        ("time_kern_prep_ns", c_uint64),
        ("time_kern_proj_ns", c_uint64),
    ]


class strom_kparam_desc(ctypes.Structure):
    _fields_ = [
        ("type_oid", c_int32),
        ("is_const", c_int32),
        ("param_id", c_int32),
        ("isnull", c_int32),
        ("length", c_int32),
        ("value", ctypes.c_uint8 * 16),
    ]


class strom_kvar_desc(ctypes.Structure):
    _fields_ = [("attno", c_int32), ("type_oid", c_int32)]


class strom_codegen_result(ctypes.Structure):
    _fields_ = [
        ("source", c_void_p),
        ("extra_flags", c_int32),
        ("nparams", c_int32),
        ("params", ctypes.POINTER(strom_kparam_desc)),
        ("nvars", c_int32),
        ("vars", ctypes.POINTER(strom_kvar_desc)),
        ("errmsg", c_void_p),
    ]


class strom_preagg_target(ctypes.Structure):
    _fields_ = [("kind", c_int32), ("type_oid", c_int32), ("scale", c_int32)]


class strom_preagg_domain(ctypes.Structure):
    _fields_ = [("nkeys", c_int32), ("key_min", ctypes.c_int64 * 8),
                ("key_range", c_uint32 * 8)]


class strom_hashtable_input(ctypes.Structure):
    _fields_ = [("inner", c_void_p), ("nkeys", c_int32), ("key_attnos", c_int32 * 8)]


class strom_column_input(ctypes.Structure):
    _fields_ = [
        ("type_oid", c_int32),
        ("attlen", ctypes.c_int16),
        ("attalign", ctypes.c_int8),
        ("attbyval", ctypes.c_int8),
        ("values", c_void_p),
        ("isnull", c_void_p),
    ]


DONE_CB = ctypes.CFUNCTYPE(None, c_void_p, c_int, ctypes.POINTER(strom_perfmon))

# every symbol include/*.h declares; tests check that each one resolves
PROTOTYPES = {
    # strom_hip.h
    "strom_init": (c_int, [ctypes.POINTER(c_int), c_int]),
    "strom_shutdown": (None, []),
    "strom_num_devices": (c_int, []),
    "strom_device_schedule": (c_int, []),
    "strom_strerror": (c_char_p, [c_int]),
    "strom_pin_host_range": (c_int, [c_void_p, c_size_t]),
    "strom_unpin_host_range": (c_int, [c_void_p]),
    "strom_device_info": (c_int, [c_int, c_char_p, c_size_t]),
    "strom_set_perfmon": (None, [c_int]),
    "strom_get_devprog_key": (c_uint64, [c_char_p, c_int32]),
    "strom_retain_devprog_key": (None, [c_uint64]),
    "strom_put_devprog_key": (None, [c_uint64]),
    "strom_get_devprog_errmsg": (c_char_p, [c_uint64]),
    "strom_lookup_device_program": (c_int, [c_uint64, c_int]),
    "strom_get_devprog_source": (c_char_p, [c_uint64]),
    "strom_dstore_upload": (c_void_p, [c_void_p, c_int]),
    "strom_dstore_wrap": (c_void_p, [c_void_p, c_size_t, c_int]),
    "strom_dstore_devptr": (c_void_p, [c_void_p]),
    "strom_dstore_length": (c_size_t, [c_void_p]),
    "strom_dstore_release": (None, [c_void_p]),
    "strom_dstore_download": (c_int, [c_void_p, c_void_p, c_size_t]),
    "strom_dstore_to_column": (c_void_p, [c_void_p, ctypes.POINTER(ctypes.c_int32), c_int,
                                          ctypes.POINTER(c_uint64), ctypes.POINTER(c_int)]),
    "strom_submit_gpuscan": (c_void_p, [c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_uint32, c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_gpupreagg_create": (c_void_p, [c_uint64, ctypes.POINTER(strom_preagg_target), c_int,
                                          c_void_p, ctypes.POINTER(strom_preagg_domain), c_int,
                                          ctypes.POINTER(c_int)]),
    "strom_hashjoin_project_column": (c_void_p, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                 c_void_p, ctypes.POINTER(c_int)]),
    "strom_submit_gpupreagg_joined": (c_void_p, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_submit_gpupreagg_lookup": (c_void_p, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_gpupreagg_create_hashed": (c_void_p, [c_uint64, ctypes.POINTER(strom_preagg_target), c_int,
                                                 c_void_p, c_uint32, c_int, ctypes.POINTER(c_int)]),
    "strom_gpupreagg_table_length": (c_size_t, [c_void_p]),
    "strom_gpupreagg_bind_table": (c_int, [c_void_p, c_void_p]),
    "strom_gpupreagg_table_devptr": (c_void_p, [c_void_p]),
    "strom_kernel_numeric_cstring": (c_int, [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]),
    "strom_fixup_kernel_numeric": (c_int, [ctypes.c_uint64, c_void_p, ctypes.c_size_t]),
    "strom_gpupreagg_num_groups": (c_uint32, [c_void_p]),
    "strom_gpupreagg_checked_folds": (c_uint32, [c_void_p]),
    "strom_gpupreagg_table_layout": (c_int, [c_void_p, c_int, ctypes.POINTER(c_size_t),
                                             ctypes.POINTER(c_size_t)]),
    "strom_submit_gpupreagg": (c_void_p, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, ctypes.POINTER(c_int)]),
    "strom_gpupreagg_fetch": (ctypes.c_long, [c_void_p, c_void_p, c_size_t]),
    "strom_gpupreagg_dense_groups": (c_uint32, [c_void_p]),
    "strom_gpupreagg_census": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "strom_gpupreagg_compact": (c_int, [c_void_p, c_void_p, c_size_t]),
    "strom_gpupreagg_reset": (None, [c_void_p]),
    "strom_gpupreagg_release": (None, [c_void_p]),
    "strom_gpupreagg_chunk_domain": (c_int, [c_uint64, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int, ctypes.POINTER(strom_preagg_domain)]),
    "strom_submit_gpupreagg_chunk": (c_void_p, [c_uint64, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_size_t, c_int, ctypes.c_double, c_int,
                                                c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_hashjoin_table_create": (c_void_p, [c_uint64, c_void_p, c_size_t, c_int,
                                               ctypes.POINTER(c_int)]),
    "strom_hashjoin_table_release": (None, [c_void_p]),
    "strom_hashjoin_table_info": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int),
                                          ctypes.POINTER(c_uint32), ctypes.POINTER(c_int),
                                          ctypes.POINTER(c_uint32)]),
    "strom_hashjoin_table_download": (c_int, [c_void_p, c_void_p, c_size_t]),
    "strom_submit_gpuhashjoin": (c_void_p, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_uint32, c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_submit_gpuhashjoin_projection": (c_void_p, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_uint32,
                                                       c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_rowmap_from_task": (c_void_p, [c_void_p, ctypes.POINTER(c_int)]),
    "strom_rowmap_nvalids": (c_uint32, [c_void_p]),
    "strom_rowmap_devptr": (c_void_p, [c_void_p]),
    "strom_rowmap_release": (None, [c_void_p]),
    "strom_submit_gpuscan_mapped": (c_void_p, [c_uint64, c_void_p, c_void_p, c_void_p, c_uint32,
                                               c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_submit_gpuhashjoin_mapped": (c_void_p, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32,
                                                   c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_submit_gpupreagg_mapped": (c_void_p, [c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "strom_task_wait": (c_int, [c_void_p, ctypes.POINTER(strom_perfmon)]),
    "strom_task_release": (None, [c_void_p]),
    "strom_task_devptr": (c_void_p, [c_void_p]),
    "strom_gpupreagg_allreduce": (c_int, [c_void_p, c_void_p, c_void_p]),
    "strom_gpupreagg_sum_bound_bits": (c_int, [c_char_p, c_void_p]),
    "strom_gpupreagg_merge": (c_int, [c_void_p, c_void_p]),
    "strom_gpupreagg_reduce_scatter": (c_int, [c_void_p, c_void_p, c_void_p]),
    "strom_gpupreagg_exchange_local": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int]),
    "strom_gpupreagg_census_allreduce": (c_int, [c_void_p, c_void_p, c_void_p]),
    "strom_rccl_unique_id_bytes": (c_size_t, []),
    "strom_rccl_get_unique_id": (c_int, [c_void_p, c_size_t]),
    "strom_rccl_comm_init_rank": (c_int, [ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t, c_int, c_int]),
    "strom_rccl_comm_destroy": (c_int, [c_void_p]),
    "strom_membw_probe": (c_int, [c_int, c_size_t, c_int, ctypes.POINTER(ctypes.c_double)]),
    "strom_synchronize": (None, []),
    # strom_codegen.h
    "strom_codegen_gpuscan": (c_int, [c_char_p, ctypes.POINTER(strom_codegen_result)]),
    "strom_codegen_gpupreagg": (c_int, [c_char_p, ctypes.POINTER(strom_codegen_result),
                                        ctypes.POINTER(strom_preagg_target), c_int,
                                        ctypes.POINTER(c_int)]),
    "strom_codegen_gpuhashjoin": (c_int, [c_char_p, ctypes.POINTER(strom_codegen_result),
                                          ctypes.POINTER(c_int)]),
    "strom_codegen_available_expression": (c_int, [c_char_p, ctypes.POINTER(c_void_p)]),
    "strom_codegen_release": (None, [ctypes.POINTER(strom_codegen_result)]),
    "strom_create_kern_parambuf": (c_void_p, [ctypes.POINTER(strom_codegen_result),
                                              ctypes.POINTER(c_uint64),
                                              ctypes.POINTER(ctypes.c_uint8), c_int]),
    # strom_datastore.h
    "strom_kds_required_length": (c_size_t, [c_int, c_int, ctypes.POINTER(strom_column_input),
                                             c_uint32]),
    "strom_kds_build": (c_int, [c_int, c_int, ctypes.POINTER(strom_column_input), c_uint32,
                                c_void_p, c_size_t]),
    "strom_kds_column_head": (c_size_t, [c_int, ctypes.POINTER(strom_column_input), c_uint32, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "strom_multihash_required_length": (c_size_t, [c_int, ctypes.POINTER(strom_hashtable_input)]),
    "strom_multihash_build": (c_int, [c_int, ctypes.POINTER(strom_hashtable_input), c_void_p,
                                      c_size_t]),
    "strom_kds_to_column": (c_size_t, [c_void_p, c_void_p, c_size_t]),
    "strom_kds_fetch": (c_int, [c_void_p, c_uint32, c_uint32, ctypes.POINTER(c_uint64)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pg_strom_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or python pg_strom_amd/build.py) first; there is no fallback path"
            % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is absent
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()
libc = ctypes.CDLL(None)
libc.free.argtypes = [c_void_p]
libc.free.restype = None
