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

}	/* namespace */

hipFunction_t
strom::fixed_function(Device *dev, const char *name, int *p_errcode)
{
	return merge_function(dev, name, p_errcode);
}

namespace {

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
	const cl_uint FAILED = 0xffffffffu;

	(void)hipSetDevice(dev->hip_id);
	if ((rc = rccl_errcode(rccl().CommCount(comm, &world), "ncclCommCount")) != 0 ||
		(rc = rccl_errcode(rccl().CommUserRank(comm, &rank), "ncclCommUserRank")) != 0)
		return rc;
	/*
	 * A failure of ONE rank must fail all of them: a rank that left early would leave the others
	 * inside a collective for good.  So every rank takes part in every collective up to the point
	 * where all have agreed to go on: a rank whose export failed sends FAILED instead of its group
	 * count, a rank that cannot allocate the gather area says so in a second, one-word round, and
	 * the records are gathered only when every rank is ready.  (What is left are failures of RCCL
	 * itself, which are not a rank's own.)
	 */
	int			local_rc = stream_follows(dev, stream);
	if (local_rc == 0)
		local_rc = gpupreagg_hash_export_device(sess, &d_mine, &mine, &reclen);
	std::vector<cl_uint> counts((size_t)world, 0);
	/* gather words: [0 .. world) the answer, [world] this rank's contribution */
	d_counts = (cl_uint *)dev->pool.alloc(sizeof(cl_uint) * (size_t)(world + 1));
	auto gather_word = [&](cl_uint word, const char *what) -> int
	{
		if (!d_counts)
			return StromError_OutOfMemory;		/* (cannot even say so: the one failure that is not agreed on) */
		if (hipMemcpyAsync(d_counts + world, &word, sizeof(cl_uint), hipMemcpyHostToDevice, stream) != hipSuccess)
			return StromError_HipInternal;
		int r = rccl_errcode(rccl().AllGather(d_counts + world, d_counts, 1, ncclUint32, comm, stream), what);
		if (r != 0)
			return r;
		if (hipMemcpyAsync(counts.data(), d_counts, sizeof(cl_uint) * (size_t)world, hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipStreamSynchronize(stream) != hipSuccess)
			return StromError_HipInternal;
		return 0;
	};
	do {
		if ((rc = gather_word(local_rc == 0 ? mine : FAILED, "ncclAllGather (group counts)")) != 0)
			break;
		cl_uint		seg_len = 0;
		bool		somebody_failed = false;
		for (cl_uint c : counts)
		{
			somebody_failed = somebody_failed || (c == FAILED);
			if (c != FAILED)
				seg_len = std::max(seg_len, c);
		}
		if (somebody_failed)
		{
			rc = (local_rc != 0 ? local_rc : StromError_HipInternal);	/* every rank returns an error */
			break;
		}
		if (seg_len == 0 || world == 1)
			break;							/* nothing to merge */
		std::vector<cl_uint> group_counts = counts;
		d_all = (char *)dev->pool.alloc(reclen * (size_t)seg_len * (size_t)(world + 1));
		if ((rc = gather_word(d_all ? 1u : FAILED, "ncclAllGather (ready)")) != 0)
			break;
		for (cl_uint c : counts)
			somebody_failed = somebody_failed || (c == FAILED);
		if (somebody_failed)
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
		rc = gpupreagg_hash_import_device(sess, d_all, seg_len, (cl_uint)world, group_counts.data(), (cl_uint)rank);
	} while (0);
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	gpupreagg_hash_release(sess, d_mine);
	if (d_all) dev->pool.release(d_all);
	if (d_counts) dev->pool.release(d_counts);
	return rc;
}

/*
 * The resident table of a dense session as the merge sees it: LANES of ngroups elements, each
 * merged by one operator -- what is handed to RCCL as one all-reduce each, or applied between two
 * tables of one GPU by preagg_merge_apply (devlib/strom_merge.h).  An integer sum is three lanes
 * (low 32 bits in the sum's own section, the next 32 bits in the scratch area 'mid', the high
 * word in its section); the flags travel as bytes in the scratch area 'bits'.
 */
namespace {

enum lane_kind { LANE_SUM_I64 = 0, LANE_SUM_F64 = 1, LANE_MIN_I64 = 2, LANE_MAX_I64 = 3, LANE_MAX_U8 = 4 };

struct merge_lane {
	int			where;			/* 0 table, 1 mid, 2 bits */
	size_t		offset;			/* bytes */
	size_t		count;			/* elements */
	lane_kind	kind;
};

std::vector<merge_lane>
merge_lanes(const gpupreagg_merge_plan &plan)
{
	std::vector<merge_lane> lanes;
	size_t		n = plan.spec.ngroups;
	for (cl_uint a = 0; a < plan.spec.naggs; a++)
	{
		cl_uint		op = plan.spec.op[a];
		lane_kind	kind = (op == 2 ? LANE_SUM_F64 : op <= 1 ? LANE_SUM_I64 :
							(op == 3 || op == 5) ? LANE_MIN_I64 : LANE_MAX_I64);
		lanes.push_back(merge_lane{0, (size_t)plan.spec.vals_off[a], n, kind});
		if (op == 1)
		{
			lanes.push_back(merge_lane{1, (size_t)plan.spec.mid_idx[a] * n * sizeof(cl_ulong), n, LANE_SUM_I64});
			lanes.push_back(merge_lane{0, (size_t)plan.spec.hi_off[a], n, LANE_SUM_I64});
		}
	}
	lanes.push_back(merge_lane{2, 0, (size_t)(plan.spec.naggs + 1) * n, LANE_MAX_U8});
	return lanes;
}

/* the scratch areas of one table taking part in a merge */
struct merge_scratch {
	char	   *d_spec = nullptr;
	cl_uchar   *d_bits = nullptr;
	cl_ulong   *d_mid = nullptr;

	int alloc(Device *dev, const gpupreagg_merge_plan &plan, bool with_spec)
	{
		size_t	n = plan.spec.ngroups;
		if (with_spec)
			d_spec = (char *)dev->pool.alloc(sizeof(plan.spec));
		d_bits = (cl_uchar *)dev->pool.alloc((size_t)(plan.spec.naggs + 1) * n);
		d_mid = (cl_ulong *)dev->pool.alloc(std::max<size_t>(1, plan.nmid) * n * sizeof(cl_ulong));
		return ((with_spec && !d_spec) || !d_bits || !d_mid) ? StromError_OutOfMemory : 0;
	}
	void release(Device *dev)
	{
		if (d_spec) dev->pool.release(d_spec);
		if (d_bits) dev->pool.release(d_bits);
		if (d_mid) dev->pool.release(d_mid);
	}
	void *lane_ptr(char *table, const merge_lane &l) const
	{
		return (l.where == 0 ? (void *)(table + l.offset) : l.where == 1 ? (void *)((char *)d_mid + l.offset)
				: (void *)(d_bits + l.offset));
	}
};

int
launch_prepare_or_finish(Device *dev, hipStream_t stream, hipFunction_t fn, char *table, const char *d_spec,
						 const merge_scratch &sc, cl_uint ngroups)
{
	void	   *a_table = table;
	const void *a_spec = d_spec;
	void	   *a_bits = sc.d_bits;
	void	   *a_mid = sc.d_mid;
	void	   *args[] = { &a_table, &a_spec, &a_bits, &a_mid };
	unsigned	grid = std::max(1u, std::min<unsigned>((ngroups + 255) / 256, (unsigned)dev->prop.multiProcessorCount * 4));
	return hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess
		? 0 : StromError_HipInternal;
}

/*
 * dense sessions of one GPU: dst += src through the multi-GPU merge's own steps -- both tables
 * prepared (src as a copy: its session keeps its table), every lane combined by the operator RCCL
 * would be given, dst finished.  Order: behind everything queued for either table.
 */
int
dense_merge_local(strom_gpupreagg *dst, strom_gpupreagg *src)
{
	gpupreagg_merge_plan pd, ps;
	int		rc;
	if ((rc = gpupreagg_get_merge_plan(dst, &pd)) != 0 || (rc = gpupreagg_get_merge_plan(src, &ps)) != 0)
		return rc;
	if (pd.dindex != ps.dindex || pd.table_bytes != ps.table_bytes ||
		memcmp(&pd.spec, &ps.spec, sizeof(pd.spec)) != 0)
		return StromError_BadRequestMessage;
	Device	   *dev = get_device(pd.dindex);
	hipStream_t	stream = dev->streams[0];
	int			errcode = 0;

	(void)hipSetDevice(dev->hip_id);
	hipFunction_t fn_prep = merge_function(dev, "preagg_merge_prepare", &errcode);
	hipFunction_t fn_fin = fn_prep ? merge_function(dev, "preagg_merge_finish", &errcode) : nullptr;
	hipFunction_t fn_apply = fn_fin ? merge_function(dev, "preagg_merge_apply", &errcode) : nullptr;
	if (!fn_apply)
		return errcode;
	merge_scratch sd, ss;
	char	   *d_copy = (char *)dev->pool.alloc(ps.table_bytes);
	do {
		if (!d_copy || (rc = sd.alloc(dev, pd, true)) != 0 || (rc = ss.alloc(dev, ps, false)) != 0)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		if ((rc = stream_follows(dev, stream)) != 0)
			break;
		if (hipMemcpy(sd.d_spec, &pd.spec, sizeof(pd.spec), hipMemcpyHostToDevice) != hipSuccess ||
			hipMemcpyAsync(d_copy, ps.table, ps.table_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		if ((rc = launch_prepare_or_finish(dev, stream, fn_prep, pd.table, sd.d_spec, sd, pd.spec.ngroups)) != 0 ||
			(rc = launch_prepare_or_finish(dev, stream, fn_prep, d_copy, sd.d_spec, ss, pd.spec.ngroups)) != 0)
			break;
		for (const merge_lane &l : merge_lanes(pd))
		{
			void	   *a_dst = sd.lane_ptr(pd.table, l);
			const void *a_src = ss.lane_ptr(d_copy, l);
			cl_ulong	a_count = l.count;
			cl_uint		a_kind = (cl_uint)l.kind;
			void	   *args[] = { &a_dst, &a_src, &a_count, &a_kind };
			unsigned	grid = (unsigned)std::max<size_t>(1, std::min<size_t>((l.count + 255) / 256,
																			  (size_t)dev->prop.multiProcessorCount * 4));
			if (hipModuleLaunchKernel(fn_apply, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
			{
				rc = StromError_HipInternal;
				break;
			}
		}
		if (rc == 0)
			rc = launch_prepare_or_finish(dev, stream, fn_fin, pd.table, sd.d_spec, sd, pd.spec.ngroups);
	} while (0);
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	sd.release(dev);
	ss.release(dev);
	if (d_copy) dev->pool.release(d_copy);
	return rc;
}

}	/* namespace */

/*
 * one session's groups merged into another's (same program, same device): per-stream or
 * per-range sessions of one GPU added up without leaving HBM.  Hashed sessions: src's groups are
 * packed and imported (gpupreagg_hash_export / _import, the steps of the all-gather merge);
 * dense sessions: the tables are added lane by lane with the all-reduce merge's own prepare /
 * operator / finish steps (dense_merge_local).
 */
extern "C" int
strom_gpupreagg_merge(strom_gpupreagg *dst, strom_gpupreagg *src)
{
	if (!gpupreagg_sessions_mergeable(dst, src))
		return StromError_BadRequestMessage;
	if (!gpupreagg_is_hashed(dst, nullptr))
		return dense_merge_local(dst, src);
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
	merge_scratch sc;
	do {
		if ((rc = sc.alloc(dev, plan, true)) != 0)
			break;
		if ((rc = stream_follows(dev, stream)) != 0)
			break;
		/* (the spec is a few hundred bytes of pageable memory: synchronous copy) */
		if (hipMemcpy(sc.d_spec, &plan.spec, sizeof(plan.spec), hipMemcpyHostToDevice) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		if ((rc = launch_prepare_or_finish(dev, stream, fn_prep, plan.table, sc.d_spec, sc, plan.spec.ngroups)) != 0)
			break;
		/* one collective per lane, fused into one RCCL launch; no lane's sum can wrap */
		ncclResult_t nrc = rccl().GroupStart();
		for (const merge_lane &l : merge_lanes(plan))
		{
			if (nrc != ncclSuccess)
				break;
			void		   *ptr = sc.lane_ptr(plan.table, l);
			ncclDataType_t	dt = (l.kind == LANE_SUM_F64 ? ncclFloat64 : l.kind == LANE_MAX_U8 ? ncclUint8 : ncclInt64);
			ncclRedOp_t		red = (l.kind <= LANE_SUM_F64 ? ncclSum : l.kind == LANE_MIN_I64 ? ncclMin : ncclMax);
			nrc = rccl().AllReduce(ptr, ptr, l.count, dt, red, comm, stream);
		}
		ncclResult_t erc = rccl().GroupEnd();
		if (nrc == ncclSuccess)
			nrc = erc;
		if ((rc = rccl_errcode(nrc, "ncclAllReduce (GpuPreAgg table)")) != 0)
			break;
		rc = launch_prepare_or_finish(dev, stream, fn_fin, plan.table, sc.d_spec, sc, plan.spec.ngroups);
	} while (0);
	/* the scratch buffers go back to the pool: the stream must be through with them */
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	sc.release(dev);
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
