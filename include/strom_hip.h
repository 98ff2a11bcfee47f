/*
 * strom_hip.h -- C ABI of libstrom_hip.so, the HIP device runtime that
 * stands where the reference's OpenCL server stands.
 *
 * Every entry point names the reference interface it replaces.  The
 * reference's boundary is a message protocol (pgstrom_message,
 * pg_strom.h:238-248; mqueue.c): a backend fills a request object, the
 * server thread runs msg->cb_process() which must only *enqueue* device
 * work, and a runtime thread later stores msg->errcode and replies.  The
 * same contract is kept here as plain functions:
 *
 *     strom_submit_*()   ==  pgstrom_enqueue_message() + clserv_process_*()
 *     strom_done_cb      ==  clserv_respond_*() + pgstrom_reply_message()
 *
 * No torch / C++ types cross this boundary; pointers are host pointers to
 * the wire structs of strom_kds.h unless named strom_dstore (a chunk kept
 * resident in HBM).
 */
#ifndef STROM_HIP_H
#define STROM_HIP_H

#include "strom_kds.h"
#include "strom_codegen.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ *
 * extra_flags of a device program (pg_strom.h:263-274)
 * ------------------------------------------------------------------ */
#define DEVINFO_IS_NEGATIVE			0x0001
#define DEVTYPE_IS_VARLENA			0x0002
#define DEVTYPE_IS_BUILTIN			0x0004
#define DEVFUNC_NEEDS_TIMELIB		0x0008
#define DEVFUNC_NEEDS_TEXTLIB		0x0010
#define DEVFUNC_NEEDS_NUMERIC		0x0020
#define DEVFUNC_NEEDS_MATHLIB		0x0040
#define DEVFUNC_INCL_FLAGS			0x0078
#define DEVKERNEL_DISABLE_OPTIMIZE	0x0100
#define DEVKERNEL_NEEDS_GPUSCAN		0x0200
#define DEVKERNEL_NEEDS_HASHJOIN	0x0400
#define DEVKERNEL_NEEDS_GPUPREAGG	0x0800

/* ------------------------------------------------------------------ *
 * perfmon: same fields as pgstrom_perfmon (pg_strom.h:177-213); times in
 * microseconds like the reference (gettimeofday / CL profiling -> usec)
 * ------------------------------------------------------------------ */
typedef struct {
	cl_bool		enabled;
	cl_uint		num_samples;
	cl_ulong	time_inner_load;
	cl_ulong	time_outer_load;
	cl_ulong	time_materialize;
	cl_ulong	time_in_sendq;
	cl_ulong	time_in_recvq;
	cl_ulong	time_kern_build;
	cl_uint		num_dma_send;
	cl_uint		num_dma_recv;
	cl_ulong	bytes_dma_send;
	cl_ulong	bytes_dma_recv;
	cl_ulong	time_dma_send;
	cl_ulong	time_dma_recv;
	cl_uint		num_kern_exec;
	cl_ulong	time_kern_exec;
	cl_uint		num_kern_proj;
	cl_ulong	time_kern_proj;
	cl_uint		num_kern_prep;
	cl_uint		num_kern_sort;
	cl_ulong	time_kern_prep;
	cl_ulong	time_kern_sort;
	/* extension: kernel time in nanoseconds (HIP events resolve < 1us) */
	cl_ulong	time_kern_exec_ns;
	cl_ulong	time_kern_prep_ns;
	cl_ulong	time_kern_proj_ns;
} strom_perfmon;

/* ------------------------------------------------------------------ *
 * life cycle  (opencl_serv.c:156-215 init_opencl_context_and_shmem,
 *              224-305 pgstrom_opencl_main; shutdown 412-430)
 * device_ids == NULL: use the device HIP reports as current (one process
 * per GPU under torch.distributed sets it through LOCAL_RANK).
 * ------------------------------------------------------------------ */
int			strom_init(const int *device_ids, int ndevices);
void		strom_shutdown(void);
int			strom_num_devices(void);
/* round-robin device pick (opencl_serv.c:100-106) */
int			strom_device_schedule(void);
/* pgstrom_strerror (main.c:288-328) */
const char *strom_strerror(int errcode);
/* pin a host range so DMA is direct (opencl_serv.c:115-137) */
int			strom_pin_host_range(void *ptr, size_t length);
int			strom_unpin_host_range(void *ptr);
/* device properties the reference prints via pgstrom_opencl_device_info */
int			strom_device_info(int dindex, char *buf, size_t buflen);
/* GUC-like switch: pg_strom.perfmon (main.c:116-123) */
void		strom_set_perfmon(int enabled);

/* ------------------------------------------------------------------ *
 * device program cache  (opencl_devprog.c)
 *   strom_get_devprog_key      <- pgstrom_get_devprog_key      (580-659)
 *   strom_retain/put           <- pgstrom_retain/put_devprog_key (669-697)
 *   strom_get_devprog_errmsg   <- pgstrom_get_devprog_errmsg   (708-720)
 *   strom_lookup_device_program<- clserv_lookup_device_program (270-569)
 * A key is the handle the reference stores as 'Datum dprog_key'.  The
 * build (hiprtc, --offload-arch=gfx950) starts asynchronously on first
 * reference; requests submitted while it runs are parked and re-issued
 * when it ends, as the reference parks messages on the program's wait
 * queue.
 * ------------------------------------------------------------------ */
typedef uint64_t strom_devprog_key;

#define STROM_DEVPROG_READY		1
#define STROM_DEVPROG_PENDING	0
#define STROM_DEVPROG_BAD		(-1)	/* BAD_OPENCL_PROGRAM, pg_strom.h:587 */

strom_devprog_key strom_get_devprog_key(const char *source, int32_t extra_flags);
void		strom_retain_devprog_key(strom_devprog_key key);
void		strom_put_devprog_key(strom_devprog_key key);
const char *strom_get_devprog_errmsg(strom_devprog_key key);
int			strom_lookup_device_program(strom_devprog_key key, int wait);
/* the full text handed to the compiler: pg_strom.show_device_kernel */
const char *strom_get_devprog_source(strom_devprog_key key);

/* ------------------------------------------------------------------ *
 * device-resident chunks.  The reference re-sends a chunk for every
 * request (clserv_dmasend_data_store, datastore.c:837-973); a strom_dstore
 * is the same bytes kept in HBM so that a table can be scanned again (and
 * chained operators can share it) without crossing PCIe.
 * ------------------------------------------------------------------ */
typedef struct strom_dstore strom_dstore;

strom_dstore *strom_dstore_upload(const kern_data_store *kds, int dindex);
/* adopt device memory somebody else filled (e.g. a torch tensor): no copy,
 * not freed on release */
strom_dstore *strom_dstore_wrap(void *devptr, size_t length, int dindex);
void	   *strom_dstore_devptr(strom_dstore *ds);
size_t		strom_dstore_length(strom_dstore *ds);
void		strom_dstore_release(strom_dstore *ds);
/* copy the first 'length' bytes of the chunk image back to the host */
int			strom_dstore_download(strom_dstore *ds, void *host, size_t length);
/*
 * Transpose a resident ROW / ROW_FLAT / TUPSLOT chunk (fixed-width columns)
 * into a new resident KDS_FORMAT_COLUMN chunk on the device -- the step the
 * reference has no need for because its kernels walk heap tuples
 * (kern_get_datum_rs, opencl_common.h:886-907).  'type_oids' (ncols
 * entries, may be NULL) enables the zone maps GpuPreAgg / GpuHashJoin use.
 * A numeric column tagged STROM_NUMERICOID becomes the 8-byte device form;
 * tagged STROM_DECIMAL_TYPE(scale) -- a typmod-scaled numeric(p,s) -- it
 * becomes a DECIMAL column, int8 at 10^-scale, which programs read as
 * (var N decimal S) without any per-row decode (exact, or the whole
 * conversion answers StromError_CpuReCheck and the chunk stays as it is).
 * Blocks until the chunk is ready; *p_kern_ns receives the device time.
 */
strom_dstore *strom_dstore_to_column(strom_dstore *src, const int32_t *type_oids, int ntypes,
									 uint64_t *p_kern_ns, int *p_errcode);

/* ------------------------------------------------------------------ *
 * requests
 * ------------------------------------------------------------------ */
/*
 * Completion protocol (pgstrom_message, pg_strom.h:238-248; mqueue.c:140-183,
 * 533-554).  'done', when not NULL, is invoked exactly once per accepted
 * request, always on a runtime thread (never on the submitter's stack, also
 * when the request fails before any device work), after errcode, perfmon and
 * the host result image (kern_resultbuf / kds_dest) are final -- the
 * reference's clserv_respond_*() + pgstrom_reply_message().  The returned
 * handle is the message reference the backend keeps: it stays valid until
 * strom_task_wait() / strom_task_release() (pgstrom_put_message()), with or
 * without a callback, so results left in HBM can be chained after the
 * callback fired.  strom_task_wait() may be called from inside done().
 * A submit call that returns NULL (with *p_errcode set) never calls done.
 */
typedef void (*strom_done_cb)(void *arg, int errcode, const strom_perfmon *pfm);
typedef struct strom_task strom_task;

/* request flags */
#define STROM_RESULTS_ON_DEVICE		0x0001	/* leave results[] in HBM; copy
											 * back only the resultbuf head */

/*
 * GpuScan on one chunk.
 *   <- clserv_process_gpuscan (gpuscan.c:1895-2182) + clserv_respond_gpuscan
 *      (1760-1888).  'kgpuscan' is the host image {kern_parambuf,
 *      kern_resultbuf}; the head up to results[] is sent, and
 *      kern_resultbuf{nitems, errcode, results[0..nitems)} is written back
 *      into the same host memory before done() runs.
 * Exactly one of kds (host chunk, uploaded for this request) or kds_dev
 * (resident chunk) is non-NULL.  Returns NULL and sets *p_errcode when the
 * request cannot be queued.  'done' may be NULL; either way the handle is
 * given back with strom_task_wait() / strom_task_release().
 */
strom_task *strom_submit_gpuscan(strom_devprog_key key,
								 kern_gpuscan *kgpuscan,
								 const kern_data_store *kds,
								 strom_dstore *kds_dev,
								 const kern_row_map *krowmap,
								 uint32_t flags,
								 strom_done_cb done, void *arg,
								 int *p_errcode);

/* ------------------------------------------------------------------ *
 * GpuPreAgg
 *
 * The reference reduces one chunk per request and ships one partial row
 * per (work-group, group) back to the backend, whose Agg node finishes
 * with the pgstrom.* final aggregates (gpupreagg.c:2665-2776, 3849-4240;
 * pg_strom--1.0.sql:247-401).  Here a strom_gpupreagg is a reduction in
 * progress on one GPU: chunks are folded into a table that stays in HBM
 * and strom_gpupreagg_fetch() returns ONE partial row per group in the
 * reference's result format (TUPSLOT kern_data_store, group keys and
 * partial values in target-list order).  Those rows feed the same final
 * aggregates, so the final result is the reference's.  With several GPUs
 * the per-GPU tables have identical layout and are merged with an RCCL
 * all-reduce on the table itself (strom_gpupreagg_table_devptr).
 *
 * nrows() partials are int8 in the fetched rows (the reference's are
 * int4 per chunk; a whole-table count does not fit int4).
 * ------------------------------------------------------------------ */
typedef struct strom_gpupreagg strom_gpupreagg;

#define STROM_PREAGG_MAXKEYS	8
typedef struct {
	int32_t		nkeys;
	int64_t		key_min[STROM_PREAGG_MAXKEYS];
	uint32_t	key_range[STROM_PREAGG_MAXKEYS];	/* max - min + 1 */
} strom_preagg_domain;

/* domain == NULL: taken from the first chunk's zone maps (COLUMN) or a
 * min/max pre-pass (other formats).  table_devptr == NULL: the runtime
 * allocates the resident table; otherwise the caller's buffer of
 * strom_gpupreagg_table_length() bytes is used (e.g. a torch tensor that
 * RCCL reduces in place). */
strom_gpupreagg *strom_gpupreagg_create(strom_devprog_key key,
										const strom_preagg_target *targets, int ntargets,
										const kern_parambuf *kparams,
										const strom_preagg_domain *domain,
										int dindex, int *p_errcode);
size_t		strom_gpupreagg_table_length(strom_gpupreagg *sess);
int			strom_gpupreagg_bind_table(strom_gpupreagg *sess, void *table_devptr);
void	   *strom_gpupreagg_table_devptr(strom_gpupreagg *sess);
uint32_t	strom_gpupreagg_num_groups(strom_gpupreagg *sess);
/*
 * Integer partial sums never wrap silently -- CHECK_OVERFLOW_INT of the reference's
 * GPUPREAGG_AGGCALC_PSUM_TEMPLATE (opencl_gpupreagg.h:142-143, 933-948) answers CpuReCheck
 * when an accumulate leaves int8.  Here a range proof (rows x largest input magnitude
 * < 2^63) lets unchecked kernels run; a chunk it cannot cover is folded a second time,
 * add by add, by a program built with GPUPREAGG_CHECKED, inside the same request
 * (devlib/strom_gpupreagg.h, "integer sums never wrap"; hashed sessions fold such a chunk
 * into a scratch table that way and let its groups join the table under a per-group check:
 * exact there too, and a CpuReCheck leaves the table as it was).  How many requests of this session
 * took that second fold: a statistic (EXPLAIN ANALYZE material, like the reference's perfmon).
 */
uint32_t	strom_gpupreagg_checked_folds(strom_gpupreagg *sess);
/*
 * The range proof's input for a sum over an EXPRESSION: the code generator emits a bound formula over
 * the columns' zone maps (#define GPUPREAGG_SUMBOUND_<a> "..." in the generated source: reverse Polish
 * over magnitudes -- cN / nN a column's zone map / integer-part bounds, kV a constant, + and *, eK a
 * rescale by 10^K); the launch path evaluates it per COLUMN chunk instead of measuring every row.
 * This is that evaluation, for a host-resident chunk: bits of the largest magnitude, or -1.
 */
int			strom_gpupreagg_sum_bound_bits(const char *formula, const kern_data_store *kds);
/* byte offset / element kind of target 'resno' inside the table:
 * *p_bits_off  offset of its has-value bitmap (seen bitmap for keys)
 * *p_vals_off  offset of its 8-byte value array (0 for keys)          */
int			strom_gpupreagg_table_layout(strom_gpupreagg *sess, int resno,
										 size_t *p_bits_off, size_t *p_vals_off);
/*
 * fold one chunk  <- clserv_process_gpupreagg (gpupreagg.c:3849-4240).
 * The task's errcode is 0, StromError_CpuReCheck (the chunk was NOT folded:
 * the caller re-does it on the CPU like gpupreagg_next_tuple_fallback,
 * gpupreagg.c:2507-2607) or a significant error.
 */
strom_task *strom_submit_gpupreagg(strom_gpupreagg *sess,
								   const kern_data_store *kds,
								   strom_dstore *kds_dev,
								   const kern_row_map *krowmap,
								   strom_done_cb done, void *arg,
								   int *p_errcode);
/* partial rows, one per group, as a TUPSLOT kern_data_store.  Returns the
 * bytes needed when dest == NULL; the number of groups (>= 0) otherwise;
 * a negative StromError on failure. */
long		strom_gpupreagg_fetch(strom_gpupreagg *sess, kern_data_store *dest, size_t destlen);
/*
 * Group-slot agreement ("agree on dense group slots").
 * Zone maps bound every key on its own; the product of the ranges can be far
 * larger than the combinations that occur.  strom_gpupreagg_census() marks,
 * one bit per dense id, the ids of the rows of a chunk that pass the qual
 * (accumulating over calls; a copy is returned in bitmap_out when not NULL,
 * strom_gpupreagg_dense_groups() ids -> (n+31)/32 words).  The caller may
 * OR the bitmaps of several ranks.  strom_gpupreagg_compact() then maps the
 * marked ids (bitmap == NULL: the session's own census) to consecutive table
 * slots; it must precede the first fold.  A later row whose combination was
 * not marked fails its chunk with StromError_DataStoreOutOfRange.
 */
uint32_t	strom_gpupreagg_dense_groups(strom_gpupreagg *sess);
int			strom_gpupreagg_census(strom_gpupreagg *sess,
								   const kern_data_store *kds, strom_dstore *kds_dev,
								   const kern_row_map *krowmap,
								   uint32_t *bitmap_out, size_t nwords);
int			strom_gpupreagg_compact(strom_gpupreagg *sess, const uint32_t *bitmap, size_t nwords);
/*
 * hashed GROUP BY: keys of ANY device type (float4/float8/numeric as well)
 * and any spread.  The reference sorts row indexes by gpupreagg_keycomp and
 * reduces runs of equal keys (opencl_gpupreagg.h:620-856, bitonic steps
 * driven from gpupreagg.c:3955-4120); here the groups live in an
 * open-addressing table in HBM keyed by the keys' canonical 64-bit images
 * (-0 = +0, one NaN, stripped numerics), grown by the library as the group
 * count needs.  A session made by this call takes the same
 * strom_submit_gpupreagg*() / strom_gpupreagg_fetch() / _reset() / _release()
 * calls; the dense-table calls (table_length, bind_table, table_devptr,
 * table_layout, census, compact) answer BadRequest / 0 for it, and
 * strom_gpupreagg_num_groups() returns the groups seen so far.  Sessions of
 * several ranks are merged by concatenating their partial rows: the final
 * aggregate adds them up (pg_strom--1.0.sql:247-401).
 * ngroups_hint sizes the first table (0 = default) and picks the number of
 * hash roles of the first fold.  The hashed kernels are a device program of
 * their own, derived from 'key' by the library (its source behind
 * "#define GPUPREAGG_HASHED 1"); it builds in the background and the first
 * fold waits for it.  A caller may also pass the key of that derived program.
 */
strom_gpupreagg *strom_gpupreagg_create_hashed(strom_devprog_key key,
											   const strom_preagg_target *targets, int ntargets,
											   const kern_parambuf *kparams,
											   uint32_t ngroups_hint,
											   int dindex, int *p_errcode);
void		strom_gpupreagg_reset(strom_gpupreagg *sess);
void		strom_gpupreagg_release(strom_gpupreagg *sess);
/*
 * The dense domain of ONE chunk: per group key the min and the range of the
 * values its rows carry after the qual (kernel gpupreagg_keyrange; any chunk
 * format, key expressions included).  Blocking, planning time.  Answers
 * StromError_DataStoreOutOfRange when the keys have no dense ids (float /
 * numeric keys, a spread beyond 32 bits): such a query takes
 * strom_gpupreagg_create_hashed().
 */
int			strom_gpupreagg_chunk_domain(strom_devprog_key key,
										 const strom_preagg_target *targets, int ntargets,
										 const kern_parambuf *kparams,
										 const kern_data_store *kds, strom_dstore *kds_dev,
										 const kern_row_map *krowmap,
										 int dindex, strom_preagg_domain *domain_out);
/*
 * The reference's own per-chunk message, field by field:
 *   pgstrom_gpupreagg {msg, dprog_key, needs_grouping, num_groups, pds,
 *                      pds_dest, kern_gpupreagg}   (opencl_gpupreagg.h:994-1003)
 *   <- clserv_process_gpupreagg (gpupreagg.c:3849-4240) + clserv_respond_gpupreagg
 * One chunk in, that chunk's partial rows out; nothing is kept between
 * requests (a backend that wants the table to stay in HBM across chunks uses
 * the session calls above).  'kgpreagg' is the host image {status,
 * sortbuf_len, kern_parambuf, kern_row_map} as KERN_GPUPREAGG_* lay it out
 * (nvalids < 0: every row; sortbuf_len is not used -- nothing is sorted);
 * 'kds_dest' is the caller's TUPSLOT buffer of dest_length bytes.  Before
 * done() runs: kgpreagg->status is the errcode (KERN_GPUPREAGG_DMARECV_*),
 * and on success kds_dest holds one row per group of this chunk, nitems
 * set, keys and partial values in target-list order exactly as
 * strom_gpupreagg_fetch() writes them (the backend reads rows 0..nitems-1,
 * gpupreagg.c:2610-2664).  StromError_CpuReCheck: kds_dest is untouched, the
 * backend aggregates the chunk itself (gpupreagg_next_tuple_fallback,
 * gpupreagg.c:2507-2607).  StromError_DataStoreNoSpace: kds_dest is too small
 * for the chunk's groups.  needs_grouping must agree with the program (it
 * has group keys or not); num_groups is the planner's estimate and sizes the
 * first hash table when the keys have no dense ids.  The request runs on a
 * runtime thread (key range -> table geometry -> fold -> partial rows); this
 * call only queues it.
 */
strom_task *strom_submit_gpupreagg_chunk(strom_devprog_key key,
										 const strom_preagg_target *targets, int ntargets,
										 kern_gpupreagg *kgpreagg,
										 const kern_data_store *kds, strom_dstore *kds_dev,
										 kern_data_store *kds_dest, size_t dest_length,
										 int needs_grouping, double num_groups,
										 int dindex,
										 strom_done_cb done, void *arg,
										 int *p_errcode);

/* ------------------------------------------------------------------ *
 * GpuHashJoin
 *
 * strom_hashjoin_table  <- the kern_multihash the reference uploads once
 *   per device and shares between the chunks of a join by a count under
 *   mhtables->lock (gpuhashjoin.c:4498-4557, released 4311-4320).  Create
 *   uploads a private copy and builds the probe index with kernels of the
 *   join's own program (so the key must be ready or becomes ready here).
 * strom_submit_gpuhashjoin <- clserv_process_gpuhashjoin (4430-5073).
 *   'khashjoin' is the host image {kern_parambuf, kern_resultbuf}; result
 *   records are nrels ints: outer_row + 1, then per inner relation the byte
 *   offset of the matched kern_hashentry inside its kern_hashtable.
 *   errcode StromError_DataStoreNoSpace: kern_resultbuf.nitems holds the
 *   number of records a retry needs room for (the reference re-enqueues
 *   with an exactly-sized buffer, 4330-4425).  A row-level CpuReCheck is
 *   reported as the chunk errcode: the reference has no CPU path for join
 *   rows ("CPU Recheck not implemented yet", 2948-2952).
 * ------------------------------------------------------------------ */
typedef struct strom_hashjoin_table strom_hashjoin_table;

strom_hashjoin_table *strom_hashjoin_table_create(strom_devprog_key key,
												  const kern_multihash *kmhash, size_t length,
												  int dindex, int *p_errcode);
void		strom_hashjoin_table_release(strom_hashjoin_table *tbl);
/* index form chosen for inner relation 'depth' (1-based): mode 1 = direct (one dense
 * integer-like key), 2 = keyed (one key of any type: 16-byte slots that carry the key
 * image), 0 = hashed (several keys) */
int			strom_hashjoin_table_info(strom_hashjoin_table *tbl, int depth,
									  int *p_mode, uint32_t *p_nslots,
									  int *p_unique, uint32_t *p_nentries);
/* the device copy of the kern_multihash (entry offsets in result records
 * index into it; rowid / htup of an entry are unchanged) */
int			strom_hashjoin_table_download(strom_hashjoin_table *tbl, void *buffer, size_t buflen);
strom_task *strom_submit_gpuhashjoin(strom_hashjoin_table *tbl,
									 kern_hashjoin *khashjoin,
									 const kern_data_store *kds,
									 strom_dstore *kds_dev,
									 const kern_row_map *krowmap,
									 uint32_t flags,
									 strom_done_cb done, void *arg,
									 int *p_errcode);

/*
 * Same, followed by kern_gpuhashjoin_projection_slot (opencl_hashjoin.h:
 * 691-839): the joined rows are materialised into 'kds_dest', a TUPSLOT
 * kern_data_store whose head (ncols, colmeta, nrooms) the caller filled;
 * destination column r takes column src_colidx[r] (0-based) of relation
 * src_depth[r] (0 = outer chunk, d = d-th inner relation).  On completion
 * kds_dest holds nitems rows.  Fixed-width by-value columns.
 * A kds_dest of KDS_FORMAT_ROW_FLAT takes kern_gpuhashjoin_projection_row
 * instead (opencl_hashjoin.h:437-689): the joined rows as heap tuples that
 * grow from the tail of the caller's buffer ('length' bytes in all; head with
 * ncols, colmeta {attlen, attalign}, nrooms, tdtypeid / tdtypmod filled by the
 * caller), row items behind the head, 'usage' = bytes of tuples.  Varlena
 * columns (attlen -1: text, character(n), numeric in its heap form) are
 * copied verbatim.  StromError_DataStoreNoSpace when the records outnumber
 * nrooms (kern_resultbuf.nitems says how many) or the tuples do not fit
 * 'length'; DataStoreCorruption when a destination column's attlen is not
 * its source's.
 */
strom_task *strom_submit_gpuhashjoin_projection(strom_hashjoin_table *tbl,
												kern_hashjoin *khashjoin,
												const kern_data_store *kds,
												strom_dstore *kds_dev,
												const kern_row_map *krowmap,
												kern_data_store *kds_dest,
												const int32_t *src_depth,
												const int32_t *src_colidx,
												uint32_t flags,
												strom_done_cb done, void *arg,
												int *p_errcode);

/*
 * Joined rows for the NEXT operator, without leaving HBM (SURVEY.md section
 * 8 f2; the reference hands a projected TUPSLOT store to the next node
 * through the host, gpuhashjoin.c:2686-2689, 4883-4968): after a
 * strom_submit_gpuhashjoin*() with STROM_RESULTS_ON_DEVICE over the resident
 * chunk 'outer', and BEFORE strom_task_wait() on it, this call waits for the
 * join and materialises its result records as a KDS_FORMAT_COLUMN chunk
 * (zone maps and not-null bitmaps included) that GpuScan / GpuPreAgg /
 * another GpuHashJoin take as a strom_dstore.  Column r takes column
 * src_colidx[r] (0-based) of relation src_depth[r]; type_oids[r] gives its
 * width, which must equal the source's (StromError_DataStoreCorruption
 * otherwise); a NEGATIVE oid means "that type, no zone map needed" and
 * spares the min/max pass over the column.  Fixed-width by-value columns.
 * Inner columns of a single-relation table with a DIRECT index and unique
 * keys are served from slot-indexed arrays the table builds on first use.
 * A join that ended with
 * StromError_DataStoreNoSpace is reported as such: resize, join again.
 */
strom_dstore *strom_hashjoin_project_column(strom_task *join_task, strom_hashjoin_table *tbl,
											strom_dstore *outer, int ncols,
											const int32_t *src_depth, const int32_t *src_colidx,
											const int32_t *type_oids, int *p_errcode);

/*
 * GpuPreAgg straight over a join's result pairs -- the projection fused into
 * its consumer.  Row i of the virtual joined
 * relation is result pair i of 'join_task' (a GpuHashJoin submitted with
 * STROM_RESULTS_ON_DEVICE over the resident COLUMN chunk 'outer', finished
 * without error, and not yet given to strom_task_wait; its device results
 * pass to the new task, so the two can then be waited for in any order);
 * its column r is column src_colidx[r] of relation src_depth[r], of
 * type type_oids[r]; the session's program reads it as (var r+1 ...).
 * Needs one inner relation with a DIRECT index and unique keys, joined on a
 * plain outer column: inner columns then come from slot-indexed arrays the
 * table builds on first use.  Anything else answers BadRequest and the
 * caller goes through strom_hashjoin_project_column().
 */
strom_task *strom_submit_gpupreagg_joined(strom_gpupreagg *sess, strom_task *join_task,
										  strom_hashjoin_table *tbl, strom_dstore *outer,
										  int ncols, const int32_t *src_depth, const int32_t *src_colidx,
										  const int32_t *type_oids,
										  strom_done_cb done, void *arg, int *p_errcode);

/*
 * ... and without any join request: the join becomes a LOOKUP inside the
 * aggregate's own pass over the resident COLUMN chunk 'outer'
 * (gpupreagg_dense_lookup): a row's slot is outer key - key_min, a row
 * without a partner is dropped, virtual columns as above, a WHERE over
 * outer columns is the aggregate program's (qual ...).  Same requirements on
 * the table (one relation, DIRECT index, unique keys, key = a plain outer
 * column of int4 / int8 / date width); anything else answers BadRequest.
 * The JOIN program's own qual and parameters are NOT evaluated on this path
 * (only the aggregate program runs): a table whose program carries a
 * pulled-up outer qual or a join qual is refused with BadRequest -- put the
 * WHERE into the aggregate program, or go through _joined / the plain join.
 */
strom_task *strom_submit_gpupreagg_lookup(strom_gpupreagg *sess,
										  strom_hashjoin_table *tbl, strom_dstore *outer,
										  int ncols, const int32_t *src_depth, const int32_t *src_colidx,
										  const int32_t *type_oids,
										  strom_done_cb done, void *arg, int *p_errcode);

/* ------------------------------------------------------------------ *
 * chained operators: device-resident row maps
 *
 * The reference hands the rows an operator selected to the next one as
 * pgstrom_bulkslot {pds, nvalids, rindex[]} built on the host from
 * kern_resultbuf (pg_strom.h:323-329, gpuscan.c:1318-1446).  Here the ids
 * stay in HBM: strom_rowmap_from_task() waits for a GpuScan submitted with
 * STROM_RESULTS_ON_DEVICE, turns its results[] into a kern_row_map in place
 * (one tiny kernel) and takes the buffer over; the *_mapped submit calls
 * pass it to the next operator on the same resident chunk.  A chunk with
 * rows to re-check (negative ids) cannot be chained: StromError_CpuReCheck
 * is returned and the caller takes the host path for that chunk.  The map
 * must outlive the requests that use it; strom_task_wait() is still due for
 * the scan task.
 * ------------------------------------------------------------------ */
typedef struct strom_rowmap strom_rowmap;
strom_rowmap *strom_rowmap_from_task(strom_task *gpuscan_task, int *p_errcode);
uint32_t	strom_rowmap_nvalids(strom_rowmap *map);
void	   *strom_rowmap_devptr(strom_rowmap *map);
void		strom_rowmap_release(strom_rowmap *map);
strom_task *strom_submit_gpuscan_mapped(strom_devprog_key key, kern_gpuscan *kgpuscan,
										strom_dstore *kds_dev, strom_rowmap *rowmap,
										uint32_t flags, strom_done_cb done, void *arg, int *p_errcode);
strom_task *strom_submit_gpuhashjoin_mapped(strom_hashjoin_table *table, kern_hashjoin *khashjoin,
											strom_dstore *kds_dev, strom_rowmap *rowmap,
											uint32_t flags, strom_done_cb done, void *arg, int *p_errcode);
strom_task *strom_submit_gpupreagg_mapped(strom_gpupreagg *sess, strom_dstore *kds_dev, strom_rowmap *rowmap,
										  strom_done_cb done, void *arg, int *p_errcode);

/* ------------------------------------------------------------------ *
 * multi-GPU: merge of the per-GPU GpuPreAgg tables over RCCL (xGMI)
 *
 * The reference has no collective (SURVEY.md section 2.3 / section 5
 * "Distributed communication backend: none"): one backend's Agg node adds up
 * the partial rows of every chunk (gpupreagg.c:4430-4773, pg_strom--1.0.sql:
 * 247-401).  With one process per GPU that addition happens between the
 * GPUs: sessions created with the SAME program, targets and key domain have
 * tables of identical layout, and strom_gpupreagg_allreduce() adds them up in
 * place -- SUM on int64 (nrows, integer and numeric psum: exact), SUM on
 * double (float psum: tolerance = summation order), MIN / MAX for pmin /
 * pmax, OR for the has-value flags -- one collective per table section,
 * grouped into one RCCL launch.  Afterwards every rank's
 * strom_gpupreagg_fetch() returns the partial rows of the WHOLE table.
 * 'comm' is an ncclComm_t (from the host's own RCCL, or from
 * strom_rccl_comm_init_rank below); 'stream' a hipStream_t or NULL = the
 * session's own stream.  Both calls return when the merge is done.
 * strom_gpupreagg_census_allreduce() is the planning-time twin: the union of
 * the ranks' census bitmaps, to be followed by strom_gpupreagg_compact(sess,
 * NULL, 0) on every rank.
 * RCCL is loaded on first use (librccl.so.1); no RCCL type crosses the ABI.
 * ------------------------------------------------------------------ */
int			strom_gpupreagg_allreduce(strom_gpupreagg *sess, void *comm, void *stream);
/*
 * Hashed GROUP BY sessions (strom_gpupreagg_create_hashed) have no common table layout: their
 * groups travel, HASH-PARTITIONED -- a group belongs to the rank its key
 * hashes to; every rank packs its groups by owner, the ranks exchange them pairwise
 * (ncclSend / ncclRecv) and each merges its own partition:
 *   strom_gpupreagg_reduce_scatter()  ends there: afterwards this rank holds the merged groups
 *                                     it owns and nothing else -- the ranks' fetches are disjoint,
 *                                     their union is the result (what parallel backends under a
 *                                     Gather node want);
 *   strom_gpupreagg_allreduce()       goes on to all-gather the final groups: afterwards every
 *                                     rank holds every merged group, as above.
 * Integer sums never wrap here either: the ranks first all-gather the largest |sum| each holds,
 * and when the sum of those is 2^63 or more every rank returns StromError_CpuReCheck with its
 * table untouched (the chunks are then aggregated on the CPU, the reference's answer to
 * CHECK_OVERFLOW_INT, opencl_gpupreagg.h:142-143).  A rank-local failure (out of memory, a full
 * table) makes EVERY rank return an error: no rank is left waiting in a collective.
 * strom_gpupreagg_exchange_local() runs the same exchange among n (<= 64) hashed sessions of ONE
 * device, session i as rank i, with device copies where the collectives are -- the harness that
 * shows a wrong owner, a lost partition or a dropped range check on one GPU.
 * strom_gpupreagg_merge() is the merge between two sessions of ONE device: src's groups are
 * added to dst's table, src is left as it is.  Hashed sessions: export, a read-only pass that
 * checks every group that exists on both sides (an integer sum that would leave int8:
 * StromError_CpuReCheck, dst untouched), then import.  Dense
 * sessions (same program, same domain, same compaction): the tables are added lane by lane with the
 * all-reduce merge's own prepare / operator / finish steps -- identities for entries without a value,
 * sign flips for the float min / max keys, flags as bytes under MAX, integer sums as carry-free limbs
 * (they are 128 bits wide in the table and must not wrap in the merge either).  Sessions that do not
 * match are refused with StromError_BadRequestMessage.
 */
int			strom_gpupreagg_merge(strom_gpupreagg *dst, strom_gpupreagg *src);
int			strom_gpupreagg_reduce_scatter(strom_gpupreagg *sess, void *comm, void *stream);
int			strom_gpupreagg_exchange_local(strom_gpupreagg **sessions, int n, int gather_after);
int			strom_gpupreagg_census_allreduce(strom_gpupreagg *sess, void *comm, void *stream);
/* communicator bootstrap for a host without its own: rank 0 makes the id
 * (strom_rccl_unique_id_bytes() bytes), hands it to the others by whatever
 * channel the host has, every rank calls comm_init_rank with its device */
size_t		strom_rccl_unique_id_bytes(void);
int			strom_rccl_get_unique_id(void *id_out, size_t len);
int			strom_rccl_comm_init_rank(void **p_comm, int nranks, const void *id, size_t len,
									  int rank, int dindex);
int			strom_rccl_comm_destroy(void *comm);

/*
 * Achievable HBM read rate of this box: a read-only streaming kernel over
 * 'nbytes' of device memory, best of 'nreps' launches, in GB/s (SURVEY.md
 * section 8d asks for the measured denominator next to the nominal 8 TB/s).
 */
int			strom_membw_probe(int dindex, size_t nbytes, int nreps, double *p_gbs);

/* block until the request finished; returns its errcode.  Frees the task. */
int			strom_task_wait(strom_task *task, strom_perfmon *pfm_out);
/* the same without the answer: pgstrom_put_message (mqueue.c:533-554) */
void		strom_task_release(strom_task *task);
/* device address of the kern_gpuscan / kern_hashjoin image of a task that
 * asked for STROM_RESULTS_ON_DEVICE (valid until strom_task_wait) */
void	   *strom_task_devptr(strom_task *task);
/* drain everything queued on every device */
void		strom_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif	/* STROM_HIP_H */
