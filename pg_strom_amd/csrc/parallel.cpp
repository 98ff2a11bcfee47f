/*
 * parallel.cpp -- multi-GPU merge of GpuPreAgg partial tables over RCCL,
 * behind the C ABI.
 *
 * The reference has no collective anywhere (SURVEY.md section 2.3, section 5
 * "Distributed communication backend: none"): one backend's Agg node adds up
 * the partial rows of all chunks with the pgstrom.* final aggregates
 * (gpupreagg.c:4430-4773, pg_strom--1.0.sql:247-401).  With one process per
 * GPU the same addition happens between the GPUs: every rank folds its row
 * range into a resident table of identical dense layout and the tables are
 * all-reduced in place, one collective per table section, grouped into ONE
 * RCCL launch (a section of 1e4 groups is 80 KB: latency bound over xGMI, so
 * few and whole-section).  devlib/strom_merge.h explains the per-section
 * operators.
 *
 * RCCL is bound at first use with dlopen("librccl.so.1"): a process that
 * already loaded an RCCL (torch does) gets that very library, a C host gets
 * the one of /opt/rocm, and a single-GPU user never loads it.  No RCCL type
 * crosses the ABI: a communicator travels as void *, the unique id as bytes.
 */
#include <dlfcn.h>
#include <cstring>
#include <cstdio>
#include <mutex>
#include <vector>
#include <algorithm>

#include <rccl/rccl.h>

#include "runtime.h"

using namespace strom;

namespace {

struct rccl_api {
	void		   *handle = nullptr;
	ncclResult_t  (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t  (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t  (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t  (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
							   ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t  (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t  (*CommCount)(const ncclComm_t, int *) = nullptr;
	ncclResult_t  (*CommUserRank)(const ncclComm_t, int *) = nullptr;
	ncclResult_t  (*GroupStart)(void) = nullptr;
	ncclResult_t  (*GroupEnd)(void) = nullptr;
	const char   *(*GetErrorString)(ncclResult_t) = nullptr;
	bool			ok = false;
};

rccl_api &
rccl(void)
{
	static rccl_api	api;
	static std::once_flag once;
	std::call_once(once, []{
		const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
		for (const char *n : names)
		{
			api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
			if (api.handle)
				break;
		}
		if (!api.handle)
		{
			fprintf(stderr, "strom_hip: cannot load librccl: %s\n", dlerror());
			return;
		}
#define BIND(field, sym)	api.field = (decltype(api.field))dlsym(api.handle, sym)
		BIND(GetUniqueId, "ncclGetUniqueId");
		BIND(CommInitRank, "ncclCommInitRank");
		BIND(CommDestroy, "ncclCommDestroy");
		BIND(AllReduce, "ncclAllReduce");
		BIND(AllGather, "ncclAllGather");
		BIND(CommCount, "ncclCommCount");
		BIND(CommUserRank, "ncclCommUserRank");
		BIND(GroupStart, "ncclGroupStart");
		BIND(GroupEnd, "ncclGroupEnd");
		BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
		api.ok = (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce &&
				  api.AllGather && api.CommCount && api.CommUserRank &&
				  api.GroupStart && api.GroupEnd && api.GetErrorString);
	});
	return api;
}

int
rccl_errcode(ncclResult_t rc, const char *what)
{
	if (rc == ncclSuccess)
		return 0;
	fprintf(stderr, "strom_hip: %s failed: %s\n", what, rccl().GetErrorString(rc));
	return StromError_HipInternal;
}

const char *merge_source =
	"#include \"strom_kds.h\"\n"
	"#include \"strom_common.h\"\n"
	"#include \"strom_merge.h\"\n";

strom_devprog_key
merge_program_key(void)
{
	/* one reference is kept for the life of the process */
	static strom_devprog_key key = strom_get_devprog_key(merge_source, 0);
	return key;
}

hipFunction_t
merge_function(Device *dev, const char *name, int *p_errcode)
{
	strom_devprog_key key = merge_program_key();
	if (strom_lookup_device_program(key, 1) != STROM_DEVPROG_READY)
	{
		*p_errcode = StromError_ProgramBuildFailure;
		return nullptr;
	}
	return lookup_program(key)->get_function(dev, name, p_errcode);
}

/* order 'stream' behind everything queued for the session's table: the folds on
 * streams[0] and the slab merges that follow them on the merge stream */
int
stream_follows(Device *dev, hipStream_t stream)
{
	for (hipStream_t src : { dev->streams[0], dev->merge_stream })
	{
		if (!src || src == stream)
			continue;
		hipEvent_t ev;
		if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
			return StromError_HipInternal;
		hipError_t rc = hipEventRecord(ev, src);
		if (rc == hipSuccess)
			rc = hipStreamWaitEvent(stream, ev, 0);
		(void)hipEventDestroy(ev);
		if (rc != hipSuccess)
			return StromError_HipInternal;
	}
	return 0;
}

}	/* namespace */

/* ------------------------------------------------------------------ *
 * communicator bootstrap for hosts without torch.distributed
 * ------------------------------------------------------------------ */
extern "C" size_t
strom_rccl_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int
strom_rccl_get_unique_id(void *id_out, size_t len)
{
	if (!id_out || len < NCCL_UNIQUE_ID_BYTES)
		return StromError_BadRequestMessage;
	if (!rccl().ok)
		return StromError_ServerNotReady;
	ncclUniqueId id;
	int rc = rccl_errcode(rccl().GetUniqueId(&id), "ncclGetUniqueId");
	if (rc == 0)
		memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
	return rc;
}

extern "C" int
strom_rccl_comm_init_rank(void **p_comm, int nranks, const void *id_bytes, size_t len, int rank, int dindex)
{
	Device *dev = get_device(dindex);
	if (!p_comm || !id_bytes || len < NCCL_UNIQUE_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks)
		return StromError_BadRequestMessage;
	if (!dev || !rccl().ok)
		return StromError_ServerNotReady;
	ncclUniqueId id;
	memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
	(void)hipSetDevice(dev->hip_id);
	ncclComm_t comm = nullptr;
	int rc = rccl_errcode(rccl().CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
	if (rc == 0)
		*p_comm = (void *)comm;
	return rc;
}

extern "C" int
strom_rccl_comm_destroy(void *comm)
{
	if (!comm)
		return 0;
	if (!rccl().ok)
		return StromError_ServerNotReady;
	return rccl_errcode(rccl().CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}

/* ------------------------------------------------------------------ *
 * merge of the resident tables
 * ------------------------------------------------------------------ */
/*
 * hashed GROUP BY sessions: the tables of the ranks have no common layout, so the groups
 * travel -- every rank packs its groups (gpupreagg_hash_export), the counts and then the
 * records are all-gathered (padded to the largest rank), and every rank merges the others'
 * records into its table (gpupreagg_hash_import).  Afterwards every rank holds every group.
 */
static int
hashed_allreduce(strom_gpupreagg *sess, int dindex, ncclComm_t comm, hipStream_t stream_or_null)
{
	Device	   *dev = get_device(dindex);
	hipStream_t	stream = (stream_or_null ? stream_or_null : dev->streams[0]);
	int			world = 0, rank = 0;
	char	   *d_mine = nullptr, *d_all = nullptr;
	cl_uint	   *d_counts = nullptr;
	cl_uint		mine = 0;
	size_t		reclen = 0;
	int			rc;

	(void)hipSetDevice(dev->hip_id);
	if ((rc = rccl_errcode(rccl().CommCount(comm, &world), "ncclCommCount")) != 0 ||
		(rc = rccl_errcode(rccl().CommUserRank(comm, &rank), "ncclCommUserRank")) != 0)
		return rc;
	if ((rc = stream_follows(dev, stream)) != 0)
		return rc;
	if ((rc = gpupreagg_hash_export_device(sess, &d_mine, &mine, &reclen)) != 0)
		return rc;
	std::vector<cl_uint> counts((size_t)world, 0);
	do {
		d_counts = (cl_uint *)dev->pool.alloc(sizeof(cl_uint) * (size_t)(world + 1));
		if (!d_counts)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		/* counts: this rank's at d_counts[world], all of them in d_counts[0 .. world) */
		if (hipMemcpyAsync(d_counts + world, &mine, sizeof(cl_uint), hipMemcpyHostToDevice, stream) != hipSuccess ||
			(rc = rccl_errcode(rccl().AllGather(d_counts + world, d_counts, 1, ncclUint32, comm, stream),
							   "ncclAllGather (group counts)")) != 0 ||
			hipMemcpyAsync(counts.data(), d_counts, sizeof(cl_uint) * (size_t)world, hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipStreamSynchronize(stream) != hipSuccess)
		{
			if (rc == 0)
				rc = StromError_HipInternal;
			break;
		}
		cl_uint		seg_len = 0;
		for (cl_uint c : counts)
			seg_len = std::max(seg_len, c);
		if (seg_len == 0 || world == 1)
			break;							/* nothing to merge */
		d_all = (char *)dev->pool.alloc(reclen * (size_t)seg_len * (size_t)(world + 1));
		if (!d_all)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		/* the send buffer is the padded copy of this rank's records behind the gather area */
		char	   *d_send = d_all + reclen * (size_t)seg_len * (size_t)world;
		if ((mine > 0 && hipMemcpyAsync(d_send, d_mine, reclen * mine, hipMemcpyDeviceToDevice, stream) != hipSuccess) ||
			(rc = rccl_errcode(rccl().AllGather(d_send, d_all, reclen * (size_t)seg_len, ncclUint8, comm, stream),
							   "ncclAllGather (groups)")) != 0 ||
			hipStreamSynchronize(stream) != hipSuccess)
		{
			if (rc == 0)
				rc = StromError_HipInternal;
			break;
		}
		rc = gpupreagg_hash_import_device(sess, d_all, seg_len, (cl_uint)world, counts.data(), (cl_uint)rank);
	} while (0);
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	gpupreagg_hash_release(sess, d_mine);
	if (d_all) dev->pool.release(d_all);
	if (d_counts) dev->pool.release(d_counts);
	return rc;
}

/* one hashed session's groups merged into another's (same program, same device): per-stream or
 * per-range sessions of one GPU added up without leaving HBM */
extern "C" int
strom_gpupreagg_merge(strom_gpupreagg *dst, strom_gpupreagg *src)
{
	int		d1 = -1, d2 = -1;
	if (!gpupreagg_is_hashed(dst, &d1) || !gpupreagg_is_hashed(src, &d2) || d1 != d2 || dst == src)
		return StromError_BadRequestMessage;
	char	   *d_recs = nullptr;
	cl_uint		count = 0;
	size_t		reclen = 0;
	int			rc = gpupreagg_hash_export_device(src, &d_recs, &count, &reclen);
	if (rc == 0 && count > 0)
		rc = gpupreagg_hash_import_device(dst, d_recs, count, 1, &count, ~0u);
	gpupreagg_hash_release(src, d_recs);
	return rc;
}

extern "C" int
strom_gpupreagg_allreduce(strom_gpupreagg *sess, void *comm_handle, void *stream_handle)
{
	int		hashed_dindex = -1;
	if (gpupreagg_is_hashed(sess, &hashed_dindex))
	{
		if (!comm_handle)
			return StromError_BadRequestMessage;
		if (!rccl().ok)
			return StromError_ServerNotReady;
		return hashed_allreduce(sess, hashed_dindex, (ncclComm_t)comm_handle, (hipStream_t)stream_handle);
	}
	gpupreagg_merge_plan plan;
	int		rc = gpupreagg_get_merge_plan(sess, &plan);

	if (rc != 0)
		return rc;
	if (!comm_handle)
		return StromError_BadRequestMessage;
	if (!rccl().ok)
		return StromError_ServerNotReady;
	Device	   *dev = get_device(plan.dindex);
	ncclComm_t	comm = (ncclComm_t)comm_handle;
	hipStream_t	stream = (stream_handle ? (hipStream_t)stream_handle : dev->streams[0]);
	int			errcode = 0;

	(void)hipSetDevice(dev->hip_id);
	hipFunction_t fn_prep = merge_function(dev, "preagg_merge_prepare", &errcode);
	hipFunction_t fn_fin = fn_prep ? merge_function(dev, "preagg_merge_finish", &errcode) : nullptr;
	if (!fn_prep || !fn_fin)
		return errcode;
	cl_uint		nbits = plan.spec.naggs + 1;
	size_t		bits_len = (size_t)nbits * plan.spec.ngroups;
	char	   *d_spec = (char *)dev->pool.alloc(sizeof(plan.spec));
	cl_uchar   *d_bits = (cl_uchar *)dev->pool.alloc(bits_len);
	do {
		if (!d_spec || !d_bits)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		if ((rc = stream_follows(dev, stream)) != 0)
			break;
		/* (the spec is a few hundred bytes of pageable memory: synchronous copy) */
		if (hipMemcpy(d_spec, &plan.spec, sizeof(plan.spec), hipMemcpyHostToDevice) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		void	   *a_table = plan.table;
		const void *a_spec = d_spec;
		void	   *a_bits = d_bits;
		void	   *args[] = { &a_table, &a_spec, &a_bits };
		unsigned	grid = std::max(1u, std::min<unsigned>((plan.spec.ngroups + 255) / 256,
														   (unsigned)dev->prop.multiProcessorCount * 4));
		if (hipModuleLaunchKernel(fn_prep, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		/* one collective per section, fused into one RCCL launch */
		ncclResult_t nrc = rccl().GroupStart();
		for (cl_uint a = 0; nrc == ncclSuccess && a < plan.spec.naggs; a++)
		{
			void		   *vals = plan.table + plan.spec.vals_off[a];
			cl_uint			op = plan.spec.op[a];
			ncclDataType_t	dt = (op == 2 ? ncclFloat64 : ncclInt64);
			ncclRedOp_t		red = (op <= 2 ? ncclSum : (op == 3 || op == 5) ? ncclMin : ncclMax);
			nrc = rccl().AllReduce(vals, vals, plan.spec.ngroups, dt, red, comm, stream);
		}
		if (nrc == ncclSuccess)
			nrc = rccl().AllReduce(d_bits, d_bits, bits_len, ncclUint8, ncclMax, comm, stream);
		ncclResult_t erc = rccl().GroupEnd();
		if (nrc == ncclSuccess)
			nrc = erc;
		if ((rc = rccl_errcode(nrc, "ncclAllReduce (GpuPreAgg table)")) != 0)
			break;
		if (hipModuleLaunchKernel(fn_fin, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
			rc = StromError_HipInternal;
	} while (0);
	/* the scratch buffers go back to the pool: the stream must be through with them */
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	if (d_spec) dev->pool.release(d_spec);
	if (d_bits) dev->pool.release(d_bits);
	return rc;
}

/*
 * union of the ranks' census bitmaps (strom_gpupreagg_census): afterwards
 * strom_gpupreagg_compact(sess, NULL, 0) gives every rank the SAME table
 * slots, which is what makes the table merge an element-wise collective
 * (SURVEY.md section 8e "agree on dense group slots")
 */
extern "C" int
strom_gpupreagg_census_allreduce(strom_gpupreagg *sess, void *comm_handle, void *stream_handle)
{
	void	   *d_census = nullptr;
	cl_uint		nbits = 0;
	int			dindex = -1;
	int			rc = gpupreagg_get_census(sess, &d_census, &nbits, &dindex);

	if (rc != 0)
		return rc;
	if (!comm_handle)
		return StromError_BadRequestMessage;
	if (!rccl().ok)
		return StromError_ServerNotReady;
	Device	   *dev = get_device(dindex);
	ncclComm_t	comm = (ncclComm_t)comm_handle;
	hipStream_t	stream = (stream_handle ? (hipStream_t)stream_handle : dev->streams[0]);
	int			errcode = 0;

	(void)hipSetDevice(dev->hip_id);
	hipFunction_t fn_unpack = merge_function(dev, "preagg_census_unpack", &errcode);
	hipFunction_t fn_pack = fn_unpack ? merge_function(dev, "preagg_census_pack", &errcode) : nullptr;
	if (!fn_unpack || !fn_pack)
		return errcode;
	cl_uchar   *d_bytes = (cl_uchar *)dev->pool.alloc(nbits);
	if (!d_bytes)
		return StromError_OutOfMemory;
	do {
		if ((rc = stream_follows(dev, stream)) != 0)
			break;
		void	   *a_bitmap = d_census;
		void	   *a_bytes = d_bytes;
		void	   *args_unpack[] = { &a_bitmap, &nbits, &a_bytes };
		unsigned	grid = std::max(1u, std::min<unsigned>((nbits + 255) / 256,
														   (unsigned)dev->prop.multiProcessorCount * 8));
		if (hipModuleLaunchKernel(fn_unpack, grid, 1, 1, 256, 1, 1, 0, stream, args_unpack, nullptr) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		if ((rc = rccl_errcode(rccl().AllReduce(d_bytes, d_bytes, nbits, ncclUint8, ncclMax, comm, stream),
							   "ncclAllReduce (census)")) != 0)
			break;
		if (hipModuleLaunchKernel(fn_pack, grid, 1, 1, 256, 1, 1, 0, stream, args_unpack, nullptr) != hipSuccess)
			rc = StromError_HipInternal;
	} while (0);
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	dev->pool.release(d_bytes);
	return rc;
}

/* ------------------------------------------------------------------ *
 * streaming-read probe: the measured HBM ceiling of this box
 * ------------------------------------------------------------------ */
extern "C" int
strom_membw_probe(int dindex, size_t nbytes, int nreps, double *p_gbs)
{
	Device *dev = get_device(dindex);
	if (!dev || !p_gbs || nbytes < (1UL << 20) || nreps < 1)
		return StromError_BadRequestMessage;
	int		errcode = 0;
	(void)hipSetDevice(dev->hip_id);
	hipFunction_t fn = merge_function(dev, "membw_stream_read", &errcode);
	if (!fn)
		return errcode;
	nbytes &= ~(size_t)15;
	unsigned	grid = (unsigned)dev->prop.multiProcessorCount * 8;
	char	   *d_src = (char *)dev->pool.alloc(nbytes);
	cl_uint	   *d_sink = (cl_uint *)dev->pool.alloc(sizeof(cl_uint) * grid);
	hipEvent_t	e0 = nullptr, e1 = nullptr;
	int			rc = 0;
	double		best = 0.0;
	do {
		if (!d_src || !d_sink)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		hipStream_t stream = dev->streams[0];
		if (hipMemsetAsync(d_src, 0x5a, nbytes, stream) != hipSuccess ||
			hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		const void *a_src = d_src;
		cl_ulong	a_nvec = nbytes / 16;
		void	   *a_sink = d_sink;
		void	   *args[] = { &a_src, &a_nvec, &a_sink };
		for (int i = 0; i < nreps + 1 && rc == 0; i++)
		{
			float	ms = 0.0f;
			if (hipEventRecord(e0, stream) != hipSuccess ||
				hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess ||
				hipEventRecord(e1, stream) != hipSuccess ||
				hipEventSynchronize(e1) != hipSuccess ||
				hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
				rc = StromError_HipInternal;
			else if (i > 0 && ms > 0.0f)			/* the first launch loads the code object */
				best = std::max(best, (double)nbytes / ((double)ms * 1e-3) / 1e9);
		}
	} while (0);
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	if (d_src) dev->pool.release(d_src);
	if (d_sink) dev->pool.release(d_sink);
	*p_gbs = best;
	return rc;
}
