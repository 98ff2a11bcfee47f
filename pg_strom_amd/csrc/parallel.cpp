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
	ncclResult_t  (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t  (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
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
		BIND(Send, "ncclSend");
		BIND(Recv, "ncclRecv");
		BIND(CommCount, "ncclCommCount");
		BIND(CommUserRank, "ncclCommUserRank");
		BIND(GroupStart, "ncclGroupStart");
		BIND(GroupEnd, "ncclGroupEnd");
		BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
		api.ok = (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce &&
				  api.AllGather && api.Send && api.Recv && api.CommCount && api.CommUserRank &&
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
 * hashed GROUP BY sessions: the tables of the ranks have no common layout, so the groups travel.
 * HASH-PARTITIONED: group g belongs to rank owner(g), a function of its key
 * (devlib: gpupreagg_hash_owner).
 *   1. every rank packs its groups by owner (gpupreagg_hash_export_parts) and measures the largest
 *      |integer sum| it holds;
 *   2. the ranks all-gather their counts-per-owner and that bound: every rank now knows the whole
 *      traffic matrix, and all of them take the same decision about the integer sums' range
 *      (the sum of the ranks' bounds below 2^63: no group's total can leave int8; otherwise every
 *      rank answers CpuReCheck and nothing has moved -- integer sums never wrap);
 *   3. one grouped send / receive per pair: rank r sends owner p's records to rank p
 *      ((world-1)/world of its groups leave, against world-1 copies of ALL of them in an all-gather);
 *   4. the table is emptied and takes the received records: every group of this rank's partition,
 *      each merged from up to 'world' partials.  A reduce-scatter ends here
 *      (strom_gpupreagg_reduce_scatter: the ranks' fetches are disjoint and their union is the result);
 *   5. the all-reduce goes on: the now final groups are packed and all-gathered (padded to the largest
 *      rank), and every rank inserts the others' -- new keys all of them, nothing is added up.
 * Per rank that is about 2 G (world-1)/world records moved and 2 G merged for G groups per rank and
 * overlapping key sets, where the all-gather of the partials moved and merged (world-1) G.
 *
 * A failure of ONE rank must fail all of them: a rank that left early would leave the others inside a
 * collective for good.  So every rank takes part in every collective up to the point where all have
 * agreed to go on: the first all-gather carries a status word next to the counts, a rank that cannot
 * allocate its receive area says so in a second, one-word round, and the records move only when every
 * rank is ready.  (What is left are failures of RCCL itself, which are not a rank's own.)
 *
 * 'peers' stands where the communicator stands when the "ranks" are sessions of ONE GPU
 * (strom_gpupreagg_exchange_local): the same packing, decisions and imports, with device copies and
 * host arrays where the collectives are -- what lets a one-GPU test see a wrong owner, a lost
 * partition or a dropped bound.
 */
namespace {

struct exchange_side {
	strom_gpupreagg *sess = nullptr;
	char	   *d_send = nullptr;			/* this side's groups, packed by owner */
	std::vector<cl_uint> counts;			/* [world] records per owner */
	std::vector<cl_uint> send_off;			/* [world] first record of owner p in d_send */
	cl_ulong	bound = 0;
	size_t		reclen = 0;
	int			rc = 0;
	char	   *d_recv = nullptr;			/* [world] segments of seg_len records */
	cl_uint		seg_len = 0;
};

/* step 1 */
void
exchange_pack(exchange_side &side, cl_uint world)
{
	side.counts.assign(world, 0);
	side.send_off.assign(world, 0);
	side.rc = gpupreagg_hash_export_parts_device(side.sess, world, &side.d_send, side.counts.data(),
												 &side.reclen, &side.bound);
	cl_uint	off = 0;
	for (cl_uint p = 0; p < world; p++)
	{
		side.send_off[p] = off;
		off += side.counts[p];
	}
}

/* step 2's decision, the same on every rank: 0, or the error all of them return */
int
exchange_verdict(const std::vector<cl_uint> &status, const std::vector<cl_ulong> &bounds, int my_rc)
{
	unsigned __int128 total = 0;
	for (size_t r = 0; r < status.size(); r++)
	{
		if (status[r] != 0)
			return (my_rc != 0 ? my_rc : StromError_HipInternal);
		total += bounds[r];
	}
	/* (one rank alone holding a sum at the edge is fine: nothing is added to it -- unless another
	 * rank has groups at all, which a zero bound excludes only for tables without integer sums) */
	size_t	nonzero = 0;
	for (cl_ulong b : bounds)
		nonzero += (b != 0);
	if (nonzero > 1 && total >= ((unsigned __int128)1 << 63))
		return StromError_CpuReCheck;
	return 0;
}

/* step 4 (and the insert half of step 5) */
int
exchange_take(exchange_side &side, cl_uint world, const cl_uint *seg_counts, bool reset_first, cl_uint skip_seg)
{
	if (reset_first)
		strom_gpupreagg_reset(side.sess);
	/* (the range of the integer sums was settled by exchange_verdict; new keys add nothing) */
	return gpupreagg_hash_import_device(side.sess, side.d_recv, side.seg_len, world, seg_counts, skip_seg, 0);
}

void
exchange_release(exchange_side &side, Device *dev)
{
	gpupreagg_hash_release(side.sess, side.d_send);
	side.d_send = nullptr;
	if (side.d_recv)
		dev->pool.release(side.d_recv);
	side.d_recv = nullptr;
}

}	/* namespace */

static int
hashed_exchange(strom_gpupreagg *sess, int dindex, ncclComm_t comm, hipStream_t stream_or_null, bool gather_after)
{
	Device	   *dev = get_device(dindex);
	hipStream_t	stream = (stream_or_null ? stream_or_null : dev->streams[0]);
	int			world = 0, rank = 0;
	cl_uint	   *d_words = nullptr;
	int			rc;
	const cl_uint FAILED = 0xffffffffu;

	(void)hipSetDevice(dev->hip_id);
	if ((rc = rccl_errcode(rccl().CommCount(comm, &world), "ncclCommCount")) != 0 ||
		(rc = rccl_errcode(rccl().CommUserRank(comm, &rank), "ncclCommUserRank")) != 0)
		return rc;
	if (world > 64)
		return StromError_BadRequestMessage;		/* (owners are counted in a 64-entry LDS histogram) */
	exchange_side side;
	side.sess = sess;
	int			local_rc = stream_follows(dev, stream);
	if (local_rc == 0)
	{
		exchange_pack(side, (cl_uint)world);
		local_rc = side.rc;
	}
	if (world == 1)
	{
		exchange_release(side, dev);
		return local_rc;							/* one rank owns every group: nothing moves */
	}
	/* the words every rank contributes: counts per owner, its bound (two words), its status */
	const size_t nw = (size_t)world + 3;
	std::vector<cl_uint> mine(nw, 0), all(nw * (size_t)world, 0);
	for (int p = 0; p < world; p++)
		mine[p] = side.counts[p];
	mine[world] = (cl_uint)side.bound;
	mine[world + 1] = (cl_uint)(side.bound >> 32);
	mine[world + 2] = (local_rc == 0 ? 0u : FAILED);
	d_words = (cl_uint *)dev->pool.alloc(sizeof(cl_uint) * nw * (size_t)(world + 1));
	auto gather_words = [&](size_t n, const char *what) -> int
	{
		if (!d_words)
			return StromError_OutOfMemory;			/* (cannot even say so: the one failure that is not agreed on) */
		cl_uint	   *d_mine = d_words + nw * (size_t)world;
		if (hipMemcpyAsync(d_mine, mine.data(), sizeof(cl_uint) * n, hipMemcpyHostToDevice, stream) != hipSuccess)
			return StromError_HipInternal;
		int r = rccl_errcode(rccl().AllGather(d_mine, d_words, n, ncclUint32, comm, stream), what);
		if (r != 0)
			return r;
		if (hipMemcpyAsync(all.data(), d_words, sizeof(cl_uint) * n * (size_t)world, hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipStreamSynchronize(stream) != hipSuccess)
			return StromError_HipInternal;
		return 0;
	};
	do {
		if ((rc = gather_words(nw, "ncclAllGather (counts per owner, bounds)")) != 0)
			break;
		std::vector<cl_uint> status((size_t)world);
		std::vector<cl_ulong> bounds((size_t)world);
		std::vector<cl_uint> from((size_t)world);			/* records rank r holds for this rank */
		for (int r = 0; r < world; r++)
		{
			const cl_uint *w = all.data() + nw * (size_t)r;
			status[r] = w[world + 2];
			bounds[r] = ((cl_ulong)w[world + 1] << 32) | w[world];
			from[r] = w[rank];
			side.seg_len = std::max(side.seg_len, w[rank]);
		}
		if ((rc = exchange_verdict(status, bounds, local_rc)) != 0)
			break;
		if (side.seg_len > 0)
			side.d_recv = (char *)dev->pool.alloc(side.reclen * (size_t)side.seg_len * (size_t)world);
		mine[0] = (side.seg_len == 0 || side.d_recv ? 1u : FAILED);
		if ((rc = gather_words(1, "ncclAllGather (ready)")) != 0)
			break;
		bool	somebody_failed = false;
		for (int r = 0; r < world; r++)
			somebody_failed = somebody_failed || (all[(size_t)r] == FAILED);
		if (somebody_failed)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		/* step 3: this rank's own partition by a device copy, the others pairwise */
		ncclResult_t nrc = rccl().GroupStart();
		for (int r = 0; r < world && nrc == ncclSuccess; r++)
		{
			if (r == rank)
				continue;
			if (side.counts[r] > 0)
				nrc = rccl().Send(side.d_send + side.reclen * (size_t)side.send_off[r],
								  side.reclen * (size_t)side.counts[r], ncclUint8, r, comm, stream);
			if (nrc == ncclSuccess && from[r] > 0)
				nrc = rccl().Recv(side.d_recv + side.reclen * (size_t)side.seg_len * (size_t)r,
								  side.reclen * (size_t)from[r], ncclUint8, r, comm, stream);
		}
		ncclResult_t erc = rccl().GroupEnd();
		if (nrc == ncclSuccess)
			nrc = erc;
		if ((rc = rccl_errcode(nrc, "ncclSend / ncclRecv (groups by owner)")) != 0)
			break;
		if (side.counts[rank] > 0 &&
			hipMemcpyAsync(side.d_recv + side.reclen * (size_t)side.seg_len * (size_t)rank,
						   side.d_send + side.reclen * (size_t)side.send_off[rank],
						   side.reclen * (size_t)side.counts[rank], hipMemcpyDeviceToDevice, stream) != hipSuccess)
			rc = StromError_HipInternal;
		if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
			rc = StromError_HipInternal;
		if (rc != 0)
			break;
		/* step 4 */
		rc = exchange_take(side, (cl_uint)world, from.data(), true, ~0u);
		if (!gather_after)
			break;
		/* step 5: the final groups of every partition to every rank.  (A rank whose import failed --
		 * no room to grow its table -- still takes part in the count round, and says so there.) */
		exchange_release(side, dev);
		char	   *d_mine = nullptr;
		cl_uint		n_mine = 0;
		local_rc = rc;
		rc = 0;
		if (local_rc == 0)
			local_rc = gpupreagg_hash_export_device(sess, &d_mine, &n_mine, &side.reclen);
		side.d_send = d_mine;
		mine[0] = (local_rc == 0 ? n_mine : FAILED);
		if ((rc = gather_words(1, "ncclAllGather (final group counts)")) != 0)
			break;
		std::vector<cl_uint> finals((size_t)world);
		side.seg_len = 0;
		for (int r = 0; r < world; r++)
		{
			finals[r] = all[(size_t)r];
			somebody_failed = somebody_failed || (finals[r] == FAILED);
			if (finals[r] != FAILED)
				side.seg_len = std::max(side.seg_len, finals[r]);
		}
		if (somebody_failed)
		{
			rc = (local_rc != 0 ? local_rc : StromError_HipInternal);
			break;
		}
		if (side.seg_len == 0)
			break;
		/* (the send buffer is the padded copy of this rank's records behind the gather area) */
		side.d_recv = (char *)dev->pool.alloc(side.reclen * (size_t)side.seg_len * (size_t)(world + 1));
		mine[0] = (side.d_recv ? 1u : FAILED);
		if ((rc = gather_words(1, "ncclAllGather (ready)")) != 0)
			break;
		for (int r = 0; r < world; r++)
			somebody_failed = somebody_failed || (all[(size_t)r] == FAILED);
		if (somebody_failed)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		char	   *d_pad = side.d_recv + side.reclen * (size_t)side.seg_len * (size_t)world;
		if ((n_mine > 0 && hipMemcpyAsync(d_pad, d_mine, side.reclen * n_mine, hipMemcpyDeviceToDevice, stream) != hipSuccess) ||
			(rc = rccl_errcode(rccl().AllGather(d_pad, side.d_recv, side.reclen * (size_t)side.seg_len, ncclUint8, comm, stream),
							   "ncclAllGather (final groups)")) != 0 ||
			hipStreamSynchronize(stream) != hipSuccess)
		{
			if (rc == 0)
				rc = StromError_HipInternal;
			break;
		}
		rc = exchange_take(side, (cl_uint)world, finals.data(), false, (cl_uint)rank);
	} while (0);
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	exchange_release(side, dev);
	if (d_words) dev->pool.release(d_words);
	return rc;
}

/*
 * The same exchange among n hashed sessions of ONE GPU: session i plays rank i.  gather_after = 0:
 * afterwards session i holds exactly the groups it owns (their union is the merged result, no group
 * twice); 1: every session holds every group.
 */
extern "C" int
strom_gpupreagg_exchange_local(strom_gpupreagg **sessions, int n, int gather_after)
{
	if (!sessions || n < 1 || n > 64)
		return StromError_BadRequestMessage;
	int		dindex = -1;
	for (int i = 0; i < n; i++)
	{
		int		di = -1;
		if (!sessions[i] || !gpupreagg_is_hashed(sessions[i], &di) || (i > 0 && di != dindex) ||
			(i > 0 && !gpupreagg_sessions_mergeable(sessions[0], sessions[i])))
			return StromError_BadRequestMessage;
		for (int j = 0; j < i; j++)
			if (sessions[j] == sessions[i])
				return StromError_BadRequestMessage;
		dindex = di;
	}
	Device	   *dev = get_device(dindex);
	cl_uint		world = (cl_uint)n;
	std::vector<exchange_side> sides((size_t)n);
	std::vector<cl_uint> status((size_t)n);
	std::vector<cl_ulong> bounds((size_t)n);
	int			rc = 0;

	(void)hipSetDevice(dev->hip_id);
	hipStream_t	stream = dev->streams[0];
	for (int i = 0; i < n; i++)
	{
		sides[i].sess = sessions[i];
		exchange_pack(sides[i], world);
		status[i] = (sides[i].rc == 0 ? 0u : 1u);
		bounds[i] = sides[i].bound;
		if (sides[i].rc != 0 && rc == 0)
			rc = sides[i].rc;
	}
	do {
		if ((rc = exchange_verdict(status, bounds, rc)) != 0)
			break;
		/* step 3 with device copies */
		std::vector<std::vector<cl_uint>> from((size_t)n, std::vector<cl_uint>((size_t)n, 0));
		for (int me = 0; me < n && rc == 0; me++)
		{
			exchange_side &dst = sides[me];
			for (int r = 0; r < n; r++)
			{
				from[me][r] = sides[r].counts[me];
				dst.seg_len = std::max(dst.seg_len, from[me][r]);
			}
			if (dst.seg_len == 0)
				continue;
			dst.d_recv = (char *)dev->pool.alloc(dst.reclen * (size_t)dst.seg_len * (size_t)n);
			if (!dst.d_recv)
			{
				rc = StromError_OutOfMemory;
				break;
			}
			for (int r = 0; r < n; r++)
				if (from[me][r] > 0 &&
					hipMemcpyAsync(dst.d_recv + dst.reclen * (size_t)dst.seg_len * (size_t)r,
								   sides[r].d_send + sides[r].reclen * (size_t)sides[r].send_off[me],
								   dst.reclen * (size_t)from[me][r], hipMemcpyDeviceToDevice, stream) != hipSuccess)
					rc = StromError_HipInternal;
		}
		if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess)
			rc = StromError_HipInternal;
		/* step 4 */
		for (int me = 0; me < n && rc == 0; me++)
			rc = exchange_take(sides[me], world, from[me].data(), true, ~0u);
		if (rc != 0 || !gather_after)
			break;
		/* step 5 */
		for (int i = 0; i < n; i++)
			exchange_release(sides[i], dev);
		std::vector<cl_uint> finals((size_t)n, 0);
		cl_uint		seg_len = 0;
		for (int i = 0; i < n && rc == 0; i++)
		{
			rc = gpupreagg_hash_export_device(sessions[i], &sides[i].d_send, &finals[i], &sides[i].reclen);
			seg_len = std::max(seg_len, finals[i]);
		}
		if (rc != 0 || seg_len == 0)
			break;
		size_t		reclen = sides[0].reclen;
		char	   *d_all = (char *)dev->pool.alloc(reclen * (size_t)seg_len * (size_t)n);
		if (!d_all)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		for (int i = 0; i < n; i++)
			if (finals[i] > 0 &&
				hipMemcpyAsync(d_all + reclen * (size_t)seg_len * (size_t)i, sides[i].d_send, reclen * (size_t)finals[i],
							   hipMemcpyDeviceToDevice, stream) != hipSuccess)
				rc = StromError_HipInternal;
		if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess)
			rc = StromError_HipInternal;
		for (int me = 0; me < n && rc == 0; me++)
			rc = gpupreagg_hash_import_device(sessions[me], d_all, seg_len, world, finals.data(), (cl_uint)me, 0);
		dev->pool.release(d_all);
	} while (0);
	(void)hipStreamSynchronize(stream);
	for (int i = 0; i < n; i++)
		exchange_release(sides[i], dev);
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
	/* (one source: every group that exists on both sides is checked by itself -- exact) */
	if (rc == 0 && count > 0)
		rc = gpupreagg_hash_import_device(dst, d_recs, count, 1, &count, ~0u, GPUPREAGG_IMPORT_EXACT);
	gpupreagg_hash_release(src, d_recs);
	return rc;
}

/* hashed sessions only: afterwards this rank holds the merged groups it owns, and nothing else */
extern "C" int
strom_gpupreagg_reduce_scatter(strom_gpupreagg *sess, void *comm_handle, void *stream_handle)
{
	int		hashed_dindex = -1;
	if (!gpupreagg_is_hashed(sess, &hashed_dindex) || !comm_handle)
		return StromError_BadRequestMessage;
	if (!rccl().ok)
		return StromError_ServerNotReady;
	return hashed_exchange(sess, hashed_dindex, (ncclComm_t)comm_handle, (hipStream_t)stream_handle, false);
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
		return hashed_exchange(sess, hashed_dindex, (ncclComm_t)comm_handle, (hipStream_t)stream_handle, true);
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
	/*
	 * A failure of ONE rank must fail all of them (as for the hashed sessions above): a rank that
	 * cannot allocate its scratch areas, or whose prepare launch fails, would otherwise leave while
	 * the others wait inside the lanes' all-reduces.  So every rank first contributes one word --
	 * "my side is ready" -- to a MAX all-reduce; the lanes move only when all are.  A rank that had
	 * prepared its table puts it back (finish is prepare's inverse) before it returns the error.
	 */
	cl_uint	   *d_agree = (cl_uint *)dev->pool.alloc(256);
	if (!d_agree)
		return StromError_OutOfMemory;		/* (cannot even say so: the one failure that is not agreed on) */
	do {
		int			local_rc = 0;
		bool		prepared = false;
		if ((local_rc = sc.alloc(dev, plan, true)) == 0 &&
			(local_rc = stream_follows(dev, stream)) == 0)
		{
			/* (the spec is a few hundred bytes of pageable memory: synchronous copy) */
			if (hipMemcpy(sc.d_spec, &plan.spec, sizeof(plan.spec), hipMemcpyHostToDevice) != hipSuccess)
				local_rc = StromError_HipInternal;
			else if ((local_rc = launch_prepare_or_finish(dev, stream, fn_prep, plan.table, sc.d_spec, sc,
														   plan.spec.ngroups)) == 0)
				prepared = true;
		}
		cl_uint		word = (local_rc == 0 ? 0u : 1u), agreed = 1u;
		if (hipMemcpyAsync(d_agree, &word, sizeof(word), hipMemcpyHostToDevice, stream) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		if ((rc = rccl_errcode(rccl().AllReduce(d_agree, d_agree, 1, ncclUint32, ncclMax, comm, stream),
							   "ncclAllReduce (ready)")) != 0)
			break;
		if (hipMemcpyAsync(&agreed, d_agree, sizeof(agreed), hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipStreamSynchronize(stream) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		if (agreed != 0)
		{
			if (prepared)
				(void)launch_prepare_or_finish(dev, stream, fn_fin, plan.table, sc.d_spec, sc, plan.spec.ngroups);
			rc = (local_rc != 0 ? local_rc : StromError_HipInternal);		/* every rank returns an error */
			break;
		}
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
	dev->pool.release(d_agree);
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
		void	   *a_bitmap = d_census;
		void	   *a_bytes = d_bytes;
		void	   *args_unpack[] = { &a_bitmap, &nbits, &a_bytes };
		unsigned	grid = std::max(1u, std::min<unsigned>((nbits + 255) / 256,
														   (unsigned)dev->prop.multiProcessorCount * 8));
		/* (a rank whose own side fails still joins the collective, with bytes of 2 -- "failed" beats
		 * "seen" under MAX -- so that every rank learns of it and none is left waiting) */
		int			local_rc = stream_follows(dev, stream);
		if (local_rc == 0 &&
			hipModuleLaunchKernel(fn_unpack, grid, 1, 1, 256, 1, 1, 0, stream, args_unpack, nullptr) != hipSuccess)
			local_rc = StromError_HipInternal;
		if (local_rc != 0 && hipMemsetAsync(d_bytes, 2, nbits, stream) != hipSuccess)
		{
			rc = local_rc;
			break;
		}
		if ((rc = rccl_errcode(rccl().AllReduce(d_bytes, d_bytes, nbits, ncclUint8, ncclMax, comm, stream),
							   "ncclAllReduce (census)")) != 0)
			break;
		cl_uchar	first = 0;
		if (nbits > 0 &&
			(hipMemcpyAsync(&first, d_bytes, 1, hipMemcpyDeviceToHost, stream) != hipSuccess ||
			 hipStreamSynchronize(stream) != hipSuccess))
		{
			rc = StromError_HipInternal;
			break;
		}
		if (first >= 2)
		{
			rc = (local_rc != 0 ? local_rc : StromError_HipInternal);
			break;
		}
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
