/*
 * gpuhashjoin.cpp -- host side of GpuHashJoin
 *
 * Role in the reference: clserv_process_gpuhashjoin (gpuhashjoin.c:4430-5073)
 * and clserv_respond_hashjoin (4132-4428).  The reference uploads the
 * kern_multihash once per device and shares it between the chunks of a
 * join by a count under a spinlock (4498-4557); here that shared object
 * is a strom_hashjoin_table: a private copy of the kern_multihash in HBM
 * plus the probe index built from it (strom_hashjoin.h).
 */
#include <cstring>
#include <map>
#include <cstdio>
#include <algorithm>

#include "runtime.h"

using namespace strom;

namespace {

/* mirror strom_hashjoin.h */
struct index_rel {
	cl_uint		mode;
	cl_uint		nslots;
	cl_long		key_min;
	cl_uint		unique;
	cl_uint		slots_off;
	cl_uint		nentries;
	cl_uint		slots3_off;		/* DIRECT + unique: the 3-byte slot array, 0 = none */
};
struct index_head {
	cl_uint		nrels;
	cl_uint		__pad[3];
	index_rel	rel[8];
};
struct build_stats {
	cl_long		key_min;
	cl_long		key_max;
	cl_uint		nentries;
	cl_uint		intlike;
};

}	/* namespace */

struct strom_hashjoin_table {
	strom_devprog_key	key = 0;
	Program			   *prog = nullptr;
	Device			   *dev = nullptr;
	char			   *d_kmhash = nullptr;
	size_t				kmhash_len = 0;
	char			   *d_index = nullptr;
	size_t				index_len = 0;
	index_head			head;
	int					ntables = 0;
	cl_uint				rel_ncols[8] = {};	/* columns of each inner relation (host-side request validation) */
	std::atomic<int>	refcnt{1};
	/* inner columns by slot for the COLUMN projection (DIRECT index, unique
	 * keys): (0-based column, attlen) -> {values, isnull}, built on first use */
	std::mutex			dim_lock;
	std::map<std::pair<int, int>, std::pair<char *, char *>> dimcols;
	/* the same PACKED, one record per slot, per set of (column, attlen)
	 * (hashjoin_build_dimrec_kernel): key = the set as text */
	std::map<std::string, std::pair<char *, unsigned>> dimrecs;
	/* narrow form of a record set (same key), built on the first lookup that can take it;
	 * reclen 0 = tried, does not fit */
	std::map<std::string, dimrec_narrow> dimrecs_narrow;
};

extern "C" strom_hashjoin_table *
strom_hashjoin_table_create(strom_devprog_key key, const kern_multihash *kmhash, size_t length,
							int dindex, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	Program *prog = lookup_program(key);
	Device *dev = get_device(dindex);
	if (!prog || !dev || !kmhash || kmhash->ntables < 1 || kmhash->ntables > 8 ||
		length < sizeof(kern_multihash))
	{
		*p_errcode = (!dev ? StromError_ServerNotReady : StromError_BadRequestMessage);
		return nullptr;
	}
	/* the image is about to be walked by kernels: every offset in it must
	 * stay inside 'length' (a faulting kernel takes the host process down) */
	cl_uint		rel_ncols[8] = {};
	{
		size_t	mh_head = offsetof(kern_multihash, htable_offset) + sizeof(cl_uint) * (size_t)kmhash->ntables;
		bool	sane = (mh_head <= length);
		for (cl_uint t = 0; sane && t < kmhash->ntables; t++)
		{
			size_t	off = kmhash->htable_offset[t];
			if (off < mh_head || (off & 7) != 0 || off + offsetof(kern_hashtable, colmeta) > length)
			{
				sane = false;
				break;
			}
			const kern_hashtable *kht = KERN_HASHTABLE(kmhash, t);
			size_t	kht_head = STROM_LONGALIGN(offsetof(kern_hashtable, colmeta) +
											   sizeof(kern_colmeta) * (size_t)kht->ncols) +
				sizeof(cl_uint) * (size_t)kht->nslots;
			if (kht->ncols < 1 || kht->ncols > 1600 || kht->nslots < 1 ||
				kht_head > kht->length || off + (size_t)kht->length > length)
				sane = false;
			rel_ncols[t] = kht->ncols;
		}
		if (!sane)
		{
			*p_errcode = StromError_DataStoreCorruption;
			return nullptr;
		}
	}
	/* the index is built by kernels of the join's own program */
	{
		std::unique_lock<std::mutex> g(prog->lock);
		prog->cond.wait(g, [&]{ return prog->state != STROM_DEVPROG_PENDING; });
		if (prog->state != STROM_DEVPROG_READY)
		{
			*p_errcode = StromError_ProgramBuildFailure;
			return nullptr;
		}
	}
	(void)hipSetDevice(dev->hip_id);
	int		errcode = 0;
	hipFunction_t fn_stats = prog->get_function(dev, "hashjoin_build_stats_kernel", &errcode);
	hipFunction_t fn_index = fn_stats ? prog->get_function(dev, "hashjoin_build_index_kernel", &errcode) : nullptr;
	if (!fn_stats || !fn_index)
	{
		*p_errcode = errcode;
		return nullptr;
	}
	std::unique_ptr<strom_hashjoin_table> tbl(new strom_hashjoin_table());
	tbl->key = key;
	tbl->prog = prog;
	tbl->dev = dev;
	tbl->ntables = (int)kmhash->ntables;
	memcpy(tbl->rel_ncols, rel_ncols, sizeof(rel_ncols));
	tbl->kmhash_len = length;
	hipStream_t stream = dev->streams[0];
	auto fail = [&](int code) -> strom_hashjoin_table * {
		if (tbl->d_kmhash) dev->pool.release(tbl->d_kmhash);
		if (tbl->d_index) dev->pool.release(tbl->d_index);
		*p_errcode = code;
		return nullptr;
	};
	tbl->d_kmhash = (char *)dev->pool.alloc(length);
	if (!tbl->d_kmhash)
		return fail(StromError_OutOfMemory);
	if (hipMemcpyAsync(tbl->d_kmhash, kmhash, length, hipMemcpyHostToDevice, stream) != hipSuccess)
		return fail(StromError_HipInternal);
	/* step 1: per relation entry count and key range */
	build_stats *d_stats = (build_stats *)dev->pool.alloc(sizeof(build_stats) * 8);
	if (!d_stats)
		return fail(StromError_OutOfMemory);
	build_stats h_stats[8];
	for (int t = 0; t < 8; t++)
	{
		h_stats[t].key_min = 0x7fffffffffffffffL;
		h_stats[t].key_max = -0x7fffffffffffffffL - 1;
		h_stats[t].nentries = 0;
		h_stats[t].intlike = 0;
	}
	bool	ok = (hipMemcpyAsync(d_stats, h_stats, sizeof(h_stats), hipMemcpyHostToDevice, stream) == hipSuccess);
	unsigned grid = (unsigned)dev->prop.multiProcessorCount * 8;
	for (int t = 0; ok && t < tbl->ntables; t++)
	{
		void   *a_km = tbl->d_kmhash;
		int		a_depth = t + 1;
		void   *a_st = d_stats + t;
		void   *args[] = { &a_km, &a_depth, &a_st };
		ok = (hipModuleLaunchKernel(fn_stats, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess);
	}
	ok = ok && hipMemcpyAsync(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, stream) == hipSuccess
		&& hipStreamSynchronize(stream) == hipSuccess;
	dev->pool.release(d_stats);
	if (!ok)
		return fail(StromError_HipInternal);
	/* step 2: choose the index form per relation, lay the slots out */
	memset(&tbl->head, 0, sizeof(tbl->head));
	tbl->head.nrels = tbl->ntables;
	size_t	off = STROM_TYPEALIGN(256, sizeof(index_head));
	bool	force_hash = (getenv("STROM_HASHJOIN_FORCE_HASH") != nullptr);
	for (int t = 0; t < tbl->ntables; t++)
	{
		index_rel  &ir = tbl->head.rel[t];
		cl_ulong	n = h_stats[t].nentries;
		bool		direct = false;
		if (!force_hash && h_stats[t].intlike && n > 0 && h_stats[t].key_max >= h_stats[t].key_min)
		{
			/* dense enough?  4 bytes per key value against ~8 bytes per
			 * entry of an open hash at load 0.5 */
			unsigned __int128 range = (unsigned __int128)((__int128)h_stats[t].key_max - h_stats[t].key_min) + 1;
			if (range <= (1UL << 27) && range <= std::max<cl_ulong>(8 * n, 1UL << 16))
			{
				direct = true;
				ir.mode = 1;
				ir.nslots = (cl_uint)range;
				ir.key_min = h_stats[t].key_min;
			}
		}
		size_t	slot_bytes = sizeof(cl_uint);
		if (!direct)
		{
			cl_ulong slots = 1024;
			while (slots < 2 * n)
				slots <<= 1;
			/* one key whose image is its value: a KEYED index (16-byte slots that carry the
			 * key image, one per distinct key); several keys, or a text / character(n) key
			 * (its image is a hash of the bytes): the HASH index over the entries' chains,
			 * probed with the types' equality functions */
			char	def[48];
			snprintf(def, sizeof(def), "#define HASHJOIN_KEYED_OK_%d 1\n", t + 1);
			bool	keyed = (strstr(prog->source.c_str(), def) != nullptr && !getenv("STROM_HASHJOIN_NO_KEYED"));
			ir.mode = (keyed ? 2 : 0);
			ir.nslots = (cl_uint)slots;
			ir.key_min = 0;
			if (keyed)
				slot_bytes = 16;
		}
		ir.unique = 1;
		ir.nentries = (cl_uint)n;
		ir.slots_off = (cl_uint)off;
		off += STROM_TYPEALIGN(256, slot_bytes * (size_t)ir.nslots);
		/* room for the 3-byte form of a DIRECT slot array (strom_hashjoin.h: slots3_off): worth it
		 * when the 4-byte array is larger than a fraction of an XCD's L2 and the table's entry
		 * offsets fit 27 bits; filled below if the keys turn out unique */
		ir.slots3_off = 0;
		if (direct && KERN_HASHTABLE(kmhash, t)->length < (1u << 27) &&
			sizeof(cl_uint) * (size_t)ir.nslots > (1u << 20) && !getenv("STROM_HASHJOIN_NO_NARROW_SLOTS"))
		{
			ir.slots3_off = (cl_uint)off;
			off += STROM_TYPEALIGN(256, 3 * (size_t)ir.nslots + 16);
		}
	}
	tbl->index_len = off;
	tbl->d_index = (char *)dev->pool.alloc(off);
	if (!tbl->d_index)
		return fail(StromError_OutOfMemory);
	ok = (hipMemsetAsync(tbl->d_index, 0, off, stream) == hipSuccess &&
		  hipMemcpyAsync(tbl->d_index, &tbl->head, sizeof(index_head), hipMemcpyHostToDevice, stream) == hipSuccess);
	for (int t = 0; ok && t < tbl->ntables; t++)
	{
		void   *a_km = tbl->d_kmhash;
		int		a_depth = t + 1;
		void   *a_ix = tbl->d_index;
		void   *args[] = { &a_km, &a_depth, &a_ix };
		ok = (hipModuleLaunchKernel(fn_index, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess);
	}
	ok = ok && hipMemcpyAsync(&tbl->head, tbl->d_index, sizeof(index_head), hipMemcpyDeviceToHost, stream) == hipSuccess
		&& hipStreamSynchronize(stream) == hipSuccess;
	if (!ok)
		return fail(StromError_HipInternal);
	/* step 3: the 3-byte slot arrays of the DIRECT relations whose keys are unique */
	{
		bool	head_changed = false;
		hipFunction_t fn_narrow = nullptr;
		for (int t = 0; ok && t < tbl->ntables; t++)
		{
			index_rel  &ir = tbl->head.rel[t];
			if (ir.slots3_off == 0)
				continue;
			if (!ir.unique || (!fn_narrow && !(fn_narrow = prog->get_function(dev, "hashjoin_narrow_slots_kernel", &errcode))))
			{
				ir.slots3_off = 0;			/* chains: the slot is the chain's head, the 4-byte form stays */
				head_changed = true;
				continue;
			}
			void   *a_ix = tbl->d_index;
			int		a_depth = t + 1;
			void   *args[] = { &a_ix, &a_depth };
			ok = (hipModuleLaunchKernel(fn_narrow, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess);
		}
		if (ok && head_changed)
			ok = (hipMemcpyAsync(tbl->d_index, &tbl->head, sizeof(index_head), hipMemcpyHostToDevice, stream) == hipSuccess);
		if (!ok || hipStreamSynchronize(stream) != hipSuccess)
			return fail(StromError_HipInternal);
	}
	strom_retain_devprog_key(key);
	return tbl.release();
	STROM_ABI_CATCH(nullptr, p_errcode)
}

extern "C" void
strom_hashjoin_table_release(strom_hashjoin_table *tbl)
{
	if (!tbl || --tbl->refcnt > 0)
		return;
	Device *dev = tbl->dev;
	(void)hipSetDevice(dev->hip_id);
	for (auto st : dev->streams)
		(void)hipStreamSynchronize(st);
	dev->pool.release(tbl->d_kmhash);
	dev->pool.release(tbl->d_index);
	for (auto &kv : tbl->dimcols)
	{
		dev->pool.release(kv.second.first);
		dev->pool.release(kv.second.second);
	}
	for (auto &kv : tbl->dimrecs)
		dev->pool.release(kv.second.first);
	for (auto &kv : tbl->dimrecs_narrow)
		if (kv.second.recs)
			dev->pool.release(kv.second.recs);
	strom_put_devprog_key(tbl->key);
	delete tbl;
}

extern "C" int
strom_hashjoin_table_info(strom_hashjoin_table *tbl, int depth,
						  int *p_mode, uint32_t *p_nslots, int *p_unique, uint32_t *p_nentries)
{
	if (!tbl || depth < 1 || depth > tbl->ntables)
		return StromError_BadRequestMessage;
	const index_rel &ir = tbl->head.rel[depth - 1];
	if (p_mode) *p_mode = (int)ir.mode;
	if (p_nslots) *p_nslots = ir.nslots;
	if (p_unique) *p_unique = (int)ir.unique;
	if (p_nentries) *p_nentries = ir.nentries;
	return 0;
}

/* copy the (re-linked) table back, for callers that resolve entry offsets */
extern "C" int
strom_hashjoin_table_download(strom_hashjoin_table *tbl, void *buffer, size_t buflen)
{
	if (!tbl || !buffer || buflen < tbl->kmhash_len)
		return StromError_BadRequestMessage;
	(void)hipSetDevice(tbl->dev->hip_id);
	return hip_errcode(hipMemcpy(buffer, tbl->d_kmhash, tbl->kmhash_len, hipMemcpyDeviceToHost),
					   "download kern_multihash");
}

namespace {

#define REQ_CHECK(call, what)												\
	do {																	\
		hipError_t __rc = (call);											\
		if (__rc != hipSuccess)												\
		{																	\
			task_fail(task, hip_errcode(__rc, what));						\
			return;															\
		}																	\
	} while (0)

struct hashjoin_request {
	/* projection (optional): TUPSLOT destination + column mapping */
	kern_data_store	   *kds_dest = nullptr;
	std::vector<cl_int>	map_depth, map_colidx;
	strom_hashjoin_table *tbl;
	kern_hashjoin	   *khj;
	const kern_data_store *kds;
	strom_dstore	   *kds_dev;
	const kern_row_map *krowmap;
	strom_rowmap	   *rowmap_dev = nullptr;	/* device-resident row map (chained operators) */
	uint32_t			flags;
	uint32_t			format;
	uint32_t			nrows;
};

void
gpuhashjoin_launch(strom_task_impl *task, hashjoin_request req)
{
	strom_hashjoin_table *tbl = req.tbl;
	Device	   *dev = task->dev;
	Program	   *prog = tbl->prog;
	kern_hashjoin *khj = req.khj;
	kern_resultbuf *kres_host = KERN_HASHJOIN_RESULTBUF(khj);
	size_t		res_offset = KERN_HASHJOIN_PARAMBUF_LENGTH(khj);
	size_t		head_len = res_offset + offsetof(kern_resultbuf, results);
	size_t		total_len = res_offset + KERN_RESULTBUF_LENGTH(kres_host->nrels, kres_host->nrooms);
	int			errcode = 0;

	(void)hipSetDevice(dev->hip_id);
	task->pfm.time_kern_build = (cl_ulong)prog->build_usec;
	if (req.kds_dev || dev->streams.size() < 2)
		task->stream = dev->streams[0];
	else
		task->stream = dev->streams[1 + dev->next_stream++ % (dev->streams.size() - 1)];
	bool	fast = (tbl->ntables == 1 && (tbl->head.rel[0].mode == 1 || tbl->head.rel[0].mode == 2) &&
					tbl->head.rel[0].unique &&
					req.format == KDS_FORMAT_COLUMN && req.krowmap == nullptr && req.rowmap_dev == nullptr &&
					!getenv("STROM_HASHJOIN_NO_FAST"));
	bool	fast_keyed = (fast && tbl->head.rel[0].mode == 2);	/* sparse integer keys: KEYED index */
	hipFunction_t fn = nullptr;
	if (fast)
	{
		/* only programs whose single clause is "int key = inner key" have it */
		int e2 = 0;
		fn = prog->get_function(dev, fast_keyed ? "gpuhashjoin_main_fast_keyed"
								: tbl->head.rel[0].slots3_off != 0 ? "gpuhashjoin_main_fast_narrow"
								: "gpuhashjoin_main_fast", &e2);
	}
	/* a program that is not fast-eligible still exports the symbol; the
	 * eligibility is baked into hashjoin_fast_outer_key() returning false,
	 * so ask the generated code */
	if (fn && !strstr(prog->source.c_str(), "#define HASHJOIN_FAST_ELIGIBLE 1"))
		fn = nullptr;
	fast = (fn != nullptr);
	/*
	 * "inner hash staged in LDS" (gpuhashjoin_main_fast_lds): a DIRECT slot array
	 * copied into the work-group's LDS and probed with ds_reads.  Measured
	 * (profiles/r02_hashjoin_lds_probe.txt, 1e8 rows, 80 % match): 2000 slots
	 * 314 us staged vs 329 us through the caches; 25000 slots 362 vs 340 us --
	 * a small slot array already lives in the CU's L1, and an LDS image that
	 * takes the whole CU leaves ONE work-group per CU.  Both sit near the floor
	 * the 0.64 GB of result pairs set (~330 us), so the staged form is used
	 * where it wins: arrays up to 32 KB (two work-groups per CU).
	 */
	size_t		lds_slot_bytes = 0;
	if (fast && !fast_keyed && !getenv("STROM_HASHJOIN_NO_LDS_SLOTS"))
	{
		int			e2 = 0, static_lds = 0;
		hipFunction_t fn_lds = prog->get_function(dev, "gpuhashjoin_main_fast_lds", &e2);
		size_t		need = STROM_TYPEALIGN(256, sizeof(cl_uint) * (size_t)tbl->head.rel[0].nslots);
		size_t		lds_max = std::min<size_t>(dev->prop.sharedMemPerBlock, 160 * 1024);
		size_t		lds_slot_limit = 32 * 1024;
		if (const char *v = getenv("STROM_HASHJOIN_LDS_SLOT_LIMIT"))
			lds_slot_limit = (size_t)atol(v);
		if (fn_lds && need <= lds_slot_limit &&
			hipFuncGetAttribute(&static_lds, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fn_lds) == hipSuccess &&
			static_lds >= 0 && (size_t)static_lds + need + 512 <= lds_max)
		{
			fn = fn_lds;
			lds_slot_bytes = need;
		}
	}
	if (!fn)
		fn = prog->get_function(dev, "gpuhashjoin_main", &errcode);
	if (!fn)
	{
		task_fail(task, errcode);
		return;
	}
	char	   *d_khj = (char *)dev->pool.alloc(total_len);
	if (!d_khj)
	{
		task_fail(task, StromError_OutOfMemory);
		return;
	}
	task->main_devptr = d_khj;
	task->keep_main = ((req.flags & STROM_RESULTS_ON_DEVICE) != 0);
	char	   *stage = (head_len + 64 <= PinnedPool::BLOCK ? dev->pinned.alloc() : nullptr);
	if (stage)
	{
		task->pinned_blocks.push_back(stage);
		memcpy(stage, khj, head_len);
	}
	task_event(task);									/* ev[0] */
	REQ_CHECK(hipMemcpyAsync(d_khj, stage ? (void *)stage : (void *)khj, head_len,
							 hipMemcpyHostToDevice, task->stream), "send kern_hashjoin");
	task->pfm.num_dma_send++;
	task->pfm.bytes_dma_send += head_len;
	const void *d_kds;
	if (req.kds_dev)
		d_kds = req.kds_dev->devptr;
	else
	{
		size_t	kds_len = req.kds->length;
		if (req.kds->format == KDS_FORMAT_ROW)
			kds_len = KERN_DATA_STORE_ROWBLOCK_OFFSET(req.kds) + (size_t)BLCKSZ * req.kds->nblocks;
		void   *p = dev->pool.alloc(kds_len);
		if (!p)
		{
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		task->devbufs.push_back(p);
		REQ_CHECK(hipMemcpyAsync(p, req.kds, kds_len, hipMemcpyHostToDevice, task->stream),
				  "send kern_data_store");
		task->pfm.num_dma_send++;
		task->pfm.bytes_dma_send += kds_len;
		d_kds = p;
	}
	const void *d_rowmap = (req.rowmap_dev ? req.rowmap_dev->devptr : nullptr);
	if (!req.rowmap_dev && req.krowmap)
	{
		size_t	len = offsetof(kern_row_map, rindex) + sizeof(cl_int) * (size_t)req.krowmap->nvalids;
		void   *p = dev->pool.alloc(len);
		if (!p)
		{
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		task->devbufs.push_back(p);
		REQ_CHECK(hipMemcpyAsync(p, req.krowmap, len, hipMemcpyHostToDevice, task->stream),
				  "send kern_row_map");
		d_rowmap = p;
	}
	task_event(task);									/* ev[1] */
	{
		int		block = 256;
		int		generic_rows = 64;								/* HASHJOIN_GENERIC_ROWS */
		if (const char *v = getenv("STROM_HASHJOIN_GENERIC_ROWS"))
			generic_rows = std::max(1, atoi(v));
		int		lds_quads = 8;									/* HASHJOIN_LDS_QUADS */
		if (const char *v = getenv("STROM_HASHJOIN_LDS_QUADS"))
			lds_quads = std::max(1, atoi(v));
		/* (the general kernel sizes its tiles from the grid: one row per thread is the smallest) */
		size_t	tile_rows = lds_slot_bytes ? (size_t)block * 4 * lds_quads
			: fast ? (size_t)block * 4 * 2 : (size_t)block;
		size_t	ntiles = (req.nrows + tile_rows - 1) / tile_rows;
		(void)generic_rows;
		int		per_cu = 0;
		if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds_slot_bytes) != hipSuccess ||
			per_cu < 1)
			per_cu = 1;
		if (const char *v = getenv("STROM_HASHJOIN_BLOCKS_PER_CU"))
			per_cu = std::max(1, atoi(v));
		size_t	grid = std::max<size_t>(1, std::min<size_t>(ntiles, (size_t)dev->prop.multiProcessorCount * per_cu));
		void	   *a_khj = d_khj;
		const void *a_km = tbl->d_kmhash;
		const void *a_ix = tbl->d_index;
		const void *a_kds = d_kds;
		const void *a_toast = nullptr;
		const void *a_map = d_rowmap;
		/*
		 * the general kernel's scratch: one record of inner offsets per row position, where its count
		 * pass leaves a row's only match for the emit pass (strom_hashjoin.h: single_mask).  Without it
		 * (no memory to spare) the emit pass probes again, as before.
		 */
		void	   *a_first = nullptr;
		if (!fast && !getenv("STROM_HASHJOIN_NO_FIRST_MATCH"))
		{
			size_t	len = (size_t)req.nrows * (size_t)tbl->ntables * sizeof(cl_int);
			if (len > 0 && len <= ((size_t)8 << 30))
			{
				a_first = dev->pool.alloc(len);
				if (a_first)
					task->devbufs.push_back(a_first);
			}
		}
		void	   *args_fast[] = { &a_khj, &a_ix, &a_kds };
		void	   *args_gen[] = { &a_khj, &a_km, &a_ix, &a_kds, &a_toast, &a_map, &a_first };
		REQ_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, block, 1, 1, (unsigned)lds_slot_bytes,
										task->stream, fast ? args_fast : args_gen, nullptr),
				  "launch gpuhashjoin kernel");
		task->pfm.num_kern_exec++;
		if (lds_slot_bytes)
			task->pfm.num_kern_prep++;			/* (reported: the slot array was staged in LDS) */
	}
	task_event(task);									/* ev[2] */
	char	   *d_dest = nullptr;
	size_t		dest_head = 0, dest_stride = 0;
	if (req.kds_dest)
	{
		/* kern_gpuhashjoin_projection_slot (gpuhashjoin.c:4585-4588, 4883-4968) */
		kern_data_store *kd = req.kds_dest;
		int			e2 = 0;
		/* TUPSLOT: one fixed-stride slot per record; ROW_FLAT: heap tuples growing from the
		 * tail of the caller's buffer (kern_gpuhashjoin_projection_row, 437-689) */
		bool		dest_rows = (kd->format == KDS_FORMAT_ROW_FLAT);
		hipFunction_t fn_proj = prog->get_function(dev, dest_rows ? "gpuhashjoin_projection_row"
												   : "gpuhashjoin_projection_slot", &e2);
		if (!fn_proj)
		{
			task_fail(task, e2);
			return;
		}
		dest_head = KDS_HEAD_LENGTH(kd->ncols);
		dest_stride = (dest_rows ? 0 : KDS_TUPSLOT_STRIDE(kd->ncols));
		size_t	dest_len = (dest_rows ? (size_t)kd->length : dest_head + dest_stride * (size_t)kd->nrooms);
		if (dest_rows)
			kd->usage = 0;
		size_t	map_len = sizeof(cl_int) * kd->ncols;
		d_dest = (char *)dev->pool.alloc(dest_len);
		cl_int *d_map = (cl_int *)dev->pool.alloc(2 * map_len);
		if (!d_dest || !d_map)
		{
			if (d_dest) dev->pool.release(d_dest);
			if (d_map) dev->pool.release(d_map);
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		task->devbufs.push_back(d_dest);
		task->devbufs.push_back(d_map);
		/* small, rare: plain synchronous-staging copies are fine here */
		REQ_CHECK(hipMemcpyAsync(d_dest, kd, dest_head, hipMemcpyHostToDevice, task->stream),
				  "send kds_dest head");
		REQ_CHECK(hipMemcpyAsync(d_map, req.map_depth.data(), map_len, hipMemcpyHostToDevice, task->stream),
				  "send projection map");
		REQ_CHECK(hipMemcpyAsync(d_map + kd->ncols, req.map_colidx.data(), map_len,
								 hipMemcpyHostToDevice, task->stream), "send projection map");
		void	   *a_khj = d_khj;
		const void *a_km = tbl->d_kmhash;
		const void *a_kds = d_kds;
		const void *a_toast = nullptr;
		void	   *a_dest = d_dest;
		const void *a_md = d_map;
		const void *a_mc = d_map + kd->ncols;
		void	   *args[] = { &a_khj, &a_km, &a_kds, &a_toast, &a_dest, &a_md, &a_mc };
		unsigned	pgrid = (unsigned)dev->prop.multiProcessorCount * 8;
		REQ_CHECK(hipModuleLaunchKernel(fn_proj, pgrid, 1, 1, 256, 1, 1, 0, task->stream, args, nullptr),
				  "launch gpuhashjoin projection");
		task->pfm.num_kern_proj++;
	}
	char	   *stage_res = (stage ? stage + STROMALIGN(head_len) : nullptr);
	REQ_CHECK(hipMemcpyAsync(stage_res ? (void *)stage_res : (void *)kres_host,
							 d_khj + res_offset, offsetof(kern_resultbuf, results),
							 hipMemcpyDeviceToHost, task->stream),
			  "recv kern_resultbuf head");
	task->pfm.num_dma_recv++;
	task->pfm.bytes_dma_recv += offsetof(kern_resultbuf, results);
	task_event(task);									/* ev[3] */
	bool	results_on_device = (req.flags & STROM_RESULTS_ON_DEVICE) != 0;
	kern_data_store *kds_dest = req.kds_dest;
	task->res_offset = res_offset;
	task->res_is_join = true;
	task->finish = [kres_host, d_khj, res_offset, results_on_device, stage_res,
					kds_dest, d_dest, dest_head, dest_stride](strom_task_impl *t)
	{
		if (stage_res)
			memcpy(kres_host, stage_res, offsetof(kern_resultbuf, results));
		if (kres_host->errcode == StromError_Success && kds_dest &&
			kres_host->nitems > kds_dest->nrooms)
			kres_host->errcode = StromError_DataStoreNoSpace;
		if (kres_host->errcode != StromError_Success)
		{
			/* DataStoreNoSpace: nitems holds the room a retry needs
			 * (gpuhashjoin.c:4330-4425); CpuReCheck: no CPU path for joins */
			t->errcode = kres_host->errcode;
			return;
		}
		if (kds_dest && kds_dest->format == KDS_FORMAT_ROW_FLAT)
		{
			/* the head (usage, nitems) and the row items, then the tuples at the tail */
			size_t	items_end = dest_head + STROMALIGN(sizeof(kern_blkitem) * (size_t)kds_dest->maxblocks) +
				STROMALIGN(sizeof(kern_rowitem) * (size_t)kres_host->nitems);
			size_t	total = kds_dest->length;
			hostptr_t hostptr = kds_dest->hostptr;
			hipError_t rc = hipMemcpyAsync(kds_dest, d_dest, std::min(items_end, total),
										   hipMemcpyDeviceToHost, t->stream);
			if (rc == hipSuccess)
				rc = hipStreamSynchronize(t->stream);
			kds_dest->hostptr = hostptr;
			size_t	usage = kds_dest->usage;
			if (rc == hipSuccess && (items_end + usage > total || kds_dest->nitems != kres_host->nitems))
			{
				/* (cannot happen when the kernel reported success) */
				t->errcode = StromError_DataStoreCorruption;
				return;
			}
			if (rc == hipSuccess && usage > 0)
			{
				rc = hipMemcpyAsync((char *)kds_dest + total - usage, d_dest + total - usage, usage,
									hipMemcpyDeviceToHost, t->stream);
				if (rc == hipSuccess)
					rc = hipStreamSynchronize(t->stream);
			}
			if (rc != hipSuccess)
				t->errcode = hip_errcode(rc, "recv kds_dest");
			t->pfm.num_dma_recv += 2;
			t->pfm.bytes_dma_recv += items_end + usage;
		}
		else if (kds_dest)
		{
			size_t	len = dest_stride * (size_t)kres_host->nitems;
			hipError_t rc = hipSuccess;
			if (len > 0)
				rc = hipMemcpyAsync((char *)kds_dest + dest_head, d_dest + dest_head, len,
									hipMemcpyDeviceToHost, t->stream);
			if (rc == hipSuccess)
				rc = hipStreamSynchronize(t->stream);
			if (rc != hipSuccess)
				t->errcode = hip_errcode(rc, "recv kds_dest");
			kds_dest->nitems = kres_host->nitems;
			t->pfm.num_dma_recv++;
			t->pfm.bytes_dma_recv += len;
		}
		t->res_nitems = kres_host->nitems;
		if (results_on_device || kres_host->nitems == 0)
			return;
		size_t	len = sizeof(cl_int) * (size_t)kres_host->nitems * kres_host->nrels;
		hipError_t rc = hipMemcpyAsync(kres_host->results,
									   d_khj + res_offset + offsetof(kern_resultbuf, results),
									   len, hipMemcpyDeviceToHost, t->stream);
		if (rc == hipSuccess)
			rc = hipStreamSynchronize(t->stream);
		if (rc != hipSuccess)
			t->errcode = hip_errcode(rc, "recv results[]");
		t->pfm.num_dma_recv++;
		t->pfm.bytes_dma_recv += len;
	};
	task_enqueue(task);
}

}	/* namespace */

static strom_task *
submit_hashjoin_common(strom_hashjoin_table *tbl, kern_hashjoin *khashjoin,
					   const kern_data_store *kds, strom_dstore *kds_dev,
					   const kern_row_map *krowmap, strom_rowmap *rowmap_dev,
					   kern_data_store *kds_dest, const int32_t *src_depth, const int32_t *src_colidx,
					   uint32_t flags, strom_done_cb done, void *arg, int *p_errcode);

/*
 * a projection mapping names (relation, column) pairs the kernels then
 * dereference: relation 0 = the outer chunk, d = the d-th inner relation
 */
static bool
projection_mapping_is_sane(const strom_hashjoin_table *tbl, cl_uint outer_ncols,
						   int ncols, const int32_t *src_depth, const int32_t *src_colidx)
{
	for (int i = 0; i < ncols; i++)
	{
		int		d = src_depth[i], c = src_colidx[i];
		if (d < 0 || d > tbl->ntables || c < 0)
			return false;
		if ((cl_uint)c >= (d == 0 ? outer_ncols : tbl->rel_ncols[d - 1]))
			return false;
	}
	return true;
}

extern "C" strom_task *
strom_submit_gpuhashjoin(strom_hashjoin_table *tbl,
						 kern_hashjoin *khashjoin,
						 const kern_data_store *kds, strom_dstore *kds_dev,
						 const kern_row_map *krowmap,
						 uint32_t flags,
						 strom_done_cb done, void *arg, int *p_errcode)
{
	return submit_hashjoin_common(tbl, khashjoin, kds, kds_dev, krowmap, nullptr, nullptr, nullptr, nullptr,
								  flags, done, arg, p_errcode);
}

extern "C" strom_task *
strom_submit_gpuhashjoin_projection(strom_hashjoin_table *tbl,
									kern_hashjoin *khashjoin,
									const kern_data_store *kds, strom_dstore *kds_dev,
									const kern_row_map *krowmap,
									kern_data_store *kds_dest,
									const int32_t *src_depth, const int32_t *src_colidx,
									uint32_t flags,
									strom_done_cb done, void *arg, int *p_errcode)
{
	if (!kds_dest || !src_depth || !src_colidx ||
		(kds_dest->format != KDS_FORMAT_TUPSLOT && kds_dest->format != KDS_FORMAT_ROW_FLAT) ||
		(kds_dest->format == KDS_FORMAT_ROW_FLAT &&
		 (size_t)kds_dest->length < KDS_HEAD_LENGTH(kds_dest->ncols) +
		 STROMALIGN(sizeof(kern_blkitem) * (size_t)kds_dest->maxblocks) +
		 STROMALIGN(sizeof(kern_rowitem) * (size_t)kds_dest->nrooms)))
	{
		if (p_errcode)
			*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	return submit_hashjoin_common(tbl, khashjoin, kds, kds_dev, krowmap, nullptr, kds_dest, src_depth, src_colidx,
								  flags, done, arg, p_errcode);
}

static strom_task *
submit_hashjoin_common(strom_hashjoin_table *tbl, kern_hashjoin *khashjoin,
					   const kern_data_store *kds, strom_dstore *kds_dev,
					   const kern_row_map *krowmap, strom_rowmap *rowmap_dev,
					   kern_data_store *kds_dest, const int32_t *src_depth, const int32_t *src_colidx,
					   uint32_t flags, strom_done_cb done, void *arg, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	if (!tbl || !khashjoin || (!kds) == (!kds_dev) ||
		(kds_dev && kds_dev->dindex != tbl->dev->dindex) ||
		(rowmap_dev && (!kds_dev || krowmap || rowmap_dev->dindex != kds_dev->dindex)))
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	kern_resultbuf *kres = KERN_HASHJOIN_RESULTBUF(khashjoin);
	if ((int)kres->nrels != tbl->ntables + 1 ||
		!program_accepts_format(tbl->prog, kds ? kds->format : kds_dev->head.format))
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	kres->nitems = 0;
	kres->errcode = StromError_Success;
	hashjoin_request req;
	req.tbl = tbl;
	req.khj = khashjoin;
	req.kds = kds;
	req.kds_dev = kds_dev;
	req.krowmap = (krowmap && krowmap->nvalids >= 0) ? krowmap : nullptr;
	req.rowmap_dev = rowmap_dev;
	req.flags = flags;
	if (kds_dest)
	{
		cl_uint	outer_ncols = (kds ? kds->ncols : kds_dev->head.ncols);
		if (kds_dest->ncols < 1 || kds_dest->ncols > 1600 ||
			!projection_mapping_is_sane(tbl, outer_ncols, (int)kds_dest->ncols, src_depth, src_colidx))
		{
			*p_errcode = StromError_BadRequestMessage;
			return nullptr;
		}
		req.kds_dest = kds_dest;
		req.map_depth.assign(src_depth, src_depth + kds_dest->ncols);
		req.map_colidx.assign(src_colidx, src_colidx + kds_dest->ncols);
	}
	kern_data_store head;
	if (kds)
		memcpy(&head, kds, offsetof(kern_data_store, colmeta));
	else
		head = kds_dev->head;
	req.format = head.format;
	req.nrows = (rowmap_dev ? rowmap_dev->nvalids
				 : req.krowmap ? (uint32_t)req.krowmap->nvalids : head.nitems);
	strom_task_impl *task = task_create(tbl->dev, done, arg);
	gpuhashjoin_launch(task, req);		/* the program is ready: the table needed it */
	return task;
	STROM_ABI_CATCH(nullptr, p_errcode)
}

extern "C" strom_task *
strom_submit_gpuhashjoin_mapped(strom_hashjoin_table *tbl, kern_hashjoin *khashjoin,
								strom_dstore *kds_dev, strom_rowmap *rowmap,
								uint32_t flags, strom_done_cb done, void *arg, int *p_errcode)
{
	if (!rowmap)
	{
		if (p_errcode)
			*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	return submit_hashjoin_common(tbl, khashjoin, nullptr, kds_dev, nullptr, rowmap,
								  nullptr, nullptr, nullptr, flags, done, arg, p_errcode);
}


/*
 * inner column 'col' (0-based, attlen bytes wide) of a single-relation table
 * with a DIRECT index and unique keys, as arrays by slot; built on first use
 * (hashjoin_build_dimcol_kernel) and kept with the table
 */
int
strom::hashjoin_table_dimcol(strom_hashjoin_table *tbl, int col, int attlen, void **p_values, void **p_isnull)
{
	Device *dev = tbl->dev;
	if (tbl->ntables != 1 || tbl->head.rel[0].mode != 1 || !tbl->head.rel[0].unique ||
		!(attlen == 1 || attlen == 2 || attlen == 4 || attlen == 8))
		return StromError_BadRequestMessage;
	std::lock_guard<std::mutex> g(tbl->dim_lock);
	auto	it = tbl->dimcols.find({col, attlen});
	if (it == tbl->dimcols.end())
	{
		int		errcode = 0;
		cl_uint	nslots = tbl->head.rel[0].nslots;
		(void)hipSetDevice(dev->hip_id);
		hipFunction_t fn_dim = tbl->prog->get_function(dev, "hashjoin_build_dimcol_kernel", &errcode);
		char   *d_vals = (char *)dev->pool.alloc((size_t)attlen * nslots + 16);
		char   *d_null = (char *)dev->pool.alloc((size_t)nslots + 16);
		cl_uint	failed = 1;
		if (fn_dim && d_vals && d_null)
		{
			const void *a_km = tbl->d_kmhash;
			const void *a_idx = tbl->d_index;
			cl_int		a_col = col, a_len = attlen;
			void	   *a_vals = d_vals, *a_null = d_null;
			void	   *a_failed = d_null + (((size_t)nslots + 3) & ~(size_t)3);	/* flag behind the array */
			void	   *args[] = { &a_km, &a_idx, &a_col, &a_len, &a_vals, &a_null, &a_failed };
			unsigned	grid = std::max(1u, std::min<unsigned>((nslots + 255) / 256,
															   (unsigned)dev->prop.multiProcessorCount * 8));
			if (!(hipMemsetAsync(a_failed, 0, sizeof(cl_uint), dev->streams[0]) == hipSuccess &&
				  hipModuleLaunchKernel(fn_dim, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) == hipSuccess &&
				  hipMemcpyAsync(&failed, a_failed, sizeof(cl_uint), hipMemcpyDeviceToHost, dev->streams[0]) == hipSuccess &&
				  hipStreamSynchronize(dev->streams[0]) == hipSuccess))
				failed = 1;
		}
		if (failed)
		{
			if (d_vals) dev->pool.release(d_vals);
			if (d_null) dev->pool.release(d_null);
			return StromError_DataStoreCorruption;
		}
		it = tbl->dimcols.insert({{col, attlen}, {d_vals, d_null}}).first;
	}
	*p_values = it->second.first;
	*p_isnull = it->second.second;
	return 0;
}

/*
 * packed slot records of the inner columns cols[] (0-based, attlens[] wide):
 * a u32 flags word (bit 0: row present, bit 1+i: column i NULL) followed by
 * the values at offsets[] (out); record length in *p_reclen.  Built on first
 * use per column set and kept with the table.
 */
int
strom::hashjoin_table_dimrecs(strom_hashjoin_table *tbl, int n, const int *cols, const int *attlens,
							  unsigned *offsets, void **p_recs, unsigned *p_reclen,
							  const int *narrowable, dimrec_narrow *narrow)
{
	struct spec_image {
		cl_uint		ncols, reclen;
		struct { cl_int col, attlen; cl_uint offset, pad; } c[16];
	} spec;
	Device *dev = tbl->dev;

	if (tbl->ntables != 1 || tbl->head.rel[0].mode != 1 || !tbl->head.rel[0].unique || n < 0 || n > 16)
		return StromError_BadRequestMessage;
	memset(&spec, 0, sizeof(spec));
	unsigned	off = 4;
	std::string	key;
	for (int i = 0; i < n; i++)
	{
		if (!(attlens[i] == 1 || attlens[i] == 2 || attlens[i] == 4 || attlens[i] == 8))
			return StromError_BadRequestMessage;
		off = (off + attlens[i] - 1) & ~(unsigned)(attlens[i] - 1);
		spec.c[i].col = cols[i];
		spec.c[i].attlen = attlens[i];
		spec.c[i].offset = offsets[i] = off;
		off += attlens[i];
		key += std::to_string(cols[i]) + ":" + std::to_string(attlens[i]) + ",";
	}
	unsigned	reclen = (off <= 8 ? 8 : off <= 16 ? 16 : off <= 32 ? 32 : (off + 63) & ~63u);
	spec.ncols = n;
	spec.reclen = reclen;
	std::lock_guard<std::mutex> g(tbl->dim_lock);
	auto	it = tbl->dimrecs.find(key);
	if (it == tbl->dimrecs.end())
	{
		int		errcode = 0;
		cl_uint	nslots = tbl->head.rel[0].nslots;
		(void)hipSetDevice(dev->hip_id);
		hipFunction_t fn = tbl->prog->get_function(dev, "hashjoin_build_dimrec_kernel", &errcode);
		char   *d_recs = (char *)dev->pool.alloc((size_t)reclen * nslots + 64);
		char   *d_spec = (char *)dev->pool.alloc(sizeof(spec) + 16);
		cl_uint	failed = 1;
		if (fn && d_recs && d_spec)
		{
			const void *a_km = tbl->d_kmhash;
			const void *a_idx = tbl->d_index;
			const void *a_spec = d_spec;
			void	   *a_recs = d_recs;
			void	   *a_failed = d_spec + sizeof(spec);
			void	   *args[] = { &a_km, &a_idx, &a_spec, &a_recs, &a_failed };
			unsigned	grid = std::max(1u, std::min<unsigned>((nslots + 255) / 256,
															   (unsigned)dev->prop.multiProcessorCount * 8));
			if (!(hipMemcpy(d_spec, &spec, sizeof(spec), hipMemcpyHostToDevice) == hipSuccess &&
				  hipMemsetAsync(a_failed, 0, sizeof(cl_uint), dev->streams[0]) == hipSuccess &&
				  hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) == hipSuccess &&
				  hipMemcpyAsync(&failed, a_failed, sizeof(cl_uint), hipMemcpyDeviceToHost, dev->streams[0]) == hipSuccess &&
				  hipStreamSynchronize(dev->streams[0]) == hipSuccess))
				failed = 1;
		}
		if (d_spec) dev->pool.release(d_spec);
		if (failed)
		{
			if (d_recs) dev->pool.release(d_recs);
			return StromError_DataStoreCorruption;
		}
		it = tbl->dimrecs.insert({key, {d_recs, reclen}}).first;
	}
	*p_recs = it->second.first;
	*p_reclen = it->second.second;
	if (narrow)
	{
		narrow->reclen = 0;
		bool	candidate = (narrowable != nullptr && n >= 1 && !getenv("STROM_HASHJOIN_NO_NARROW_RECS"));
		for (int i = 0; candidate && i < n; i++)
			candidate = (narrowable[i] != 0);
		auto	nit = tbl->dimrecs_narrow.find(key);
		if (candidate && nit == tbl->dimrecs_narrow.end())
		{
			/*
			 * ranges of the wanted columns over the present, non-NULL values -> field widths;
			 * 1 + n flag bits + the fields must fit 16 or 32 bits
			 */
			struct range_image { cl_long vmin[16]; cl_long vmax[16]; cl_uint nvalues[16]; } rg;
			struct nspec_image { cl_uint ncols, reclen; cl_uint shift[16]; cl_uint mask[16]; cl_long vmin[16]; } ns;
			dimrec_narrow	nw;
			int			errcode = 0;
			cl_uint		nslots = tbl->head.rel[0].nslots;
			hipStream_t	stream = dev->streams[0];
			(void)hipSetDevice(dev->hip_id);
			hipFunction_t fn_mm = tbl->prog->get_function(dev, "hashjoin_dimrec_minmax_kernel", &errcode);
			hipFunction_t fn_nw = tbl->prog->get_function(dev, "hashjoin_dimrec_narrow_kernel", &errcode);
			char   *d_spec = (char *)dev->pool.alloc(sizeof(spec));
			char   *d_rg = (char *)dev->pool.alloc(sizeof(rg));
			char   *d_ns = (char *)dev->pool.alloc(sizeof(ns));
			unsigned	grid = std::max(1u, std::min<unsigned>((nslots + 255) / 256,
														   (unsigned)dev->prop.multiProcessorCount * 8));
			memset(&rg, 0, sizeof(rg));
			for (int i = 0; i < 16; i++)
			{
				rg.vmin[i] = INT64_MAX;
				rg.vmax[i] = INT64_MIN;
			}
			bool	ok = (fn_mm && fn_nw && d_spec && d_rg && d_ns);
			if (ok)
			{
				const void *a_spec = d_spec;
				const void *a_recs = it->second.first;
				void	   *a_rg = d_rg;
				void	   *args[] = { &a_spec, &a_recs, &nslots, &a_rg };
				ok = (hipMemcpyAsync(d_spec, &spec, sizeof(spec), hipMemcpyHostToDevice, stream) == hipSuccess &&
					  hipMemcpyAsync(d_rg, &rg, sizeof(rg), hipMemcpyHostToDevice, stream) == hipSuccess &&
					  hipModuleLaunchKernel(fn_mm, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess &&
					  hipMemcpyAsync(&rg, d_rg, sizeof(rg), hipMemcpyDeviceToHost, stream) == hipSuccess &&
					  hipStreamSynchronize(stream) == hipSuccess);
			}
			unsigned	bits = 1 + (unsigned)n;
			memset(&ns, 0, sizeof(ns));
			for (int i = 0; ok && i < n; i++)
			{
				cl_ulong span = (rg.nvalues[i] ? (cl_ulong)rg.vmax[i] - (cl_ulong)rg.vmin[i] : 0);
				unsigned w = 0;
				while (w < 64 && (span >> w) != 0)
					w++;
				if (w > 31 || bits + w > 32)
				{
					ok = false;
					break;
				}
				ns.shift[i] = nw.shift[i] = bits;
				ns.mask[i] = nw.mask[i] = (w == 0 ? 0u : (cl_uint)((1UL << w) - 1));
				ns.vmin[i] = nw.vmin[i] = (rg.nvalues[i] ? rg.vmin[i] : 0);
				bits += w;
			}
			if (ok)
			{
				ns.ncols = (cl_uint)n;
				ns.reclen = nw.reclen = (bits <= 16 ? 2 : 4);
				char   *d_out = (char *)dev->pool.alloc((size_t)nw.reclen * nslots + 64);
				const void *a_spec = d_spec;
				const void *a_recs = it->second.first;
				const void *a_ns = d_ns;
				void	   *a_out = d_out;
				void	   *args[] = { &a_spec, &a_recs, &nslots, &a_ns, &a_out };
				ok = (d_out &&
					  hipMemcpyAsync(d_ns, &ns, sizeof(ns), hipMemcpyHostToDevice, stream) == hipSuccess &&
					  hipModuleLaunchKernel(fn_nw, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess &&
					  hipStreamSynchronize(stream) == hipSuccess);
				if (ok)
					nw.recs = d_out;
				else if (d_out)
					dev->pool.release(d_out);
			}
			if (!ok)
				nw.reclen = 0;
			if (d_spec) dev->pool.release(d_spec);
			if (d_rg) dev->pool.release(d_rg);
			if (d_ns) dev->pool.release(d_ns);
			nit = tbl->dimrecs_narrow.insert({key, nw}).first;
		}
		if (candidate && nit != tbl->dimrecs_narrow.end())
			*narrow = nit->second;
	}
	return 0;
}

/* key_min / slot count of that table, and which outer column its program joins on
 * (0: the key is an expression) */
int
strom::hashjoin_table_direct_info(strom_hashjoin_table *tbl, cl_long *p_key_min, cl_uint *p_nslots,
								  int *p_outer_key_attno, int *p_dindex, int *p_has_outer_qual)
{
	if (!tbl || tbl->ntables != 1 || tbl->head.rel[0].mode != 1 || !tbl->head.rel[0].unique ||
		!strstr(tbl->prog->source.c_str(), "#define HASHJOIN_FAST_ELIGIBLE 1"))
		return StromError_BadRequestMessage;
	const char *p = strstr(tbl->prog->source.c_str(), "#define HASHJOIN_FAST_OUTER_KEY_ATTNO ");
	*p_outer_key_attno = (p ? atoi(p + strlen("#define HASHJOIN_FAST_OUTER_KEY_ATTNO ")) : 0);
	*p_key_min = tbl->head.rel[0].key_min;
	*p_nslots = tbl->head.rel[0].nslots;
	*p_dindex = tbl->dev->dindex;
	/* a WHERE pulled up into the join program: only the join kernels evaluate it */
	if (p_has_outer_qual)
		*p_has_outer_qual = (strstr(tbl->prog->source.c_str(), "#define HASHJOIN_FAST_OUTER_QUAL 1") != nullptr);
	return 0;
}

/*
 * joined rows -> a COLUMN chunk resident in HBM (gpuhashjoin_projection_column
 * in strom_hashjoin.h); zone maps and NULL bookkeeping by the ingest
 * program's kernels, so the result is what strom_dstore_to_column() or a
 * host-built COLUMN chunk would be
 */
extern "C" strom_dstore *
strom_hashjoin_project_column(strom_task *handle, strom_hashjoin_table *tbl, strom_dstore *outer,
							  int ncols, const int32_t *src_depth, const int32_t *src_colidx,
							  const int32_t *type_oids, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	strom_task_impl *task = static_cast<strom_task_impl *>(handle);
	if (!task || !tbl || !outer || ncols < 1 || ncols > 64 || !src_depth || !src_colidx || !type_oids)
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	task_wait_completed(task);
	if (!task->res_is_join || !task->keep_main || task->errcode != 0 || !task->main_devptr ||
		outer->dindex != tbl->dev->dindex)
	{
		/* not a finished GpuHashJoin with STROM_RESULTS_ON_DEVICE */
		*p_errcode = (task->errcode ? task->errcode : StromError_BadRequestMessage);
		return nullptr;
	}
	if (!projection_mapping_is_sane(tbl, outer->head.ncols, ncols, src_depth, src_colidx))
	{
		/* the kernel indexes colmeta[] / the column directory with these */
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	Device *dev = tbl->dev;
	static const char *ingest_source =
		"#include \"strom_kds.h\"\n#include \"strom_common.h\"\n#include \"strom_mathlib.h\"\n"
		"#include \"strom_numeric.h\"\n#include \"strom_ingest.h\"\n";
	static strom_devprog_key ingest_key = strom_get_devprog_key(ingest_source, 0);
	if (strom_lookup_device_program(ingest_key, 1) != STROM_DEVPROG_READY)
	{
		*p_errcode = StromError_ProgramBuildFailure;
		return nullptr;
	}
	int		errcode = 0;
	(void)hipSetDevice(dev->hip_id);
	Program *iprog = lookup_program(ingest_key);
	hipFunction_t fn_proj = tbl->prog->get_function(dev, "gpuhashjoin_projection_column", &errcode);
	hipFunction_t fn_mm = fn_proj ? iprog->get_function(dev, "ingest_minmax", &errcode) : nullptr;
	hipFunction_t fn_fin = fn_mm ? iprog->get_function(dev, "ingest_finish", &errcode) : nullptr;
	if (!fn_proj || !fn_mm || !fn_fin)
	{
		*p_errcode = errcode;
		return nullptr;
	}
	/* destination head: COLUMN, widths from the type oids */
	cl_uint	nitems = task->res_nitems;
	std::vector<char> hbuf(KDS_COLUMN_HEAD_LENGTH(ncols), 0);
	kern_data_store *head = (kern_data_store *)hbuf.data();
	head->ncols = ncols;						/* (the column directory sits behind ncols colmeta) */
	kern_coldir *cd = KERN_DATA_STORE_COLDIR(head);
	size_t	off = KDS_COLUMN_HEAD_LENGTH(ncols);
	if ((char *)(cd + ncols) > hbuf.data() + hbuf.size())
	{
		*p_errcode = StromError_DataStoreCorruption;
		return nullptr;
	}
	for (int i = 0; i < ncols; i++)
	{
		int		attlen;
		switch (type_oids[i] < 0 ? -type_oids[i] : type_oids[i])
		{
			case STROM_TEXTOID: case STROM_BPCHARNOID:
				/* joined rows with text columns leave as heap tuples (the ROW_FLAT projection) */
				*p_errcode = StromError_BadRequestMessage;
				return nullptr;
			case STROM_BOOLOID: case STROM_BPCHAROID:	attlen = 1; break;
			case STROM_INT2OID:							attlen = 2; break;
			case STROM_INT4OID: case STROM_FLOAT4OID: case STROM_DATEOID:	attlen = 4; break;
			default:									attlen = 8; break;
		}
		head->colmeta[i].attbyval = 1;
		head->colmeta[i].attalign = (cl_char)attlen;
		head->colmeta[i].attlen = (cl_short)attlen;
		head->colmeta[i].attnum = (cl_short)(i + 1);
		head->colmeta[i].attcacheoff = -1;
		cd[i].values_off = (cl_uint)off;
		off += KDS_COLUMN_VALUES_LENGTH(attlen, nitems);
		cd[i].nulls_off = (cl_uint)off;			/* dropped by ingest_finish if unused */
		off += KDS_COLUMN_NULLS_LENGTH(nitems);
		cd[i].minval = (cl_long)~0UL;			/* unsigned-ordered seeds */
		cd[i].maxval = 0;
		if (off > 0xffffffffUL)
		{
			*p_errcode = StromError_DataStoreOutOfRange;
			return nullptr;
		}
	}
	head->length = (cl_uint)off;
	head->nitems = nitems;
	head->nrooms = nitems;
	head->format = KDS_FORMAT_COLUMN;
	head->tdtypeid = 2249;
	head->tdtypmod = -1;

	/*
	 * inner columns of a DIRECT, unique-key, single-relation table come from
	 * slot-indexed arrays (hashjoin_build_dimcol_kernel) instead of the entries
	 */
	std::vector<cl_ulong> dimptr(2 * (size_t)ncols, 0);
	bool	use_dim = (tbl->ntables == 1 && tbl->head.rel[0].mode == 1 && tbl->head.rel[0].unique &&
					   outer->head.format == KDS_FORMAT_COLUMN &&
					   strstr(tbl->prog->source.c_str(), "#define HASHJOIN_FAST_ELIGIBLE 1") != nullptr &&
					   !getenv("STROM_HASHJOIN_NO_DIMCOLS"));
	for (int i = 0; use_dim && i < ncols; i++)
	{
		void   *vals = nullptr, *nulls = nullptr;
		if (src_depth[i] == 1 &&
			hashjoin_table_dimcol(tbl, src_colidx[i], head->colmeta[i].attlen, &vals, &nulls) == 0)
		{
			dimptr[2 * i] = (cl_ulong)(uintptr_t)vals;
			dimptr[2 * i + 1] = (cl_ulong)(uintptr_t)nulls;
		}
		/* (no such column / another width: the entry path reports it) */
	}

	size_t	aux_ints = 4 * (size_t)ncols + 1;		/* depth map, column map, type oids, NULL flags, failure */
	char   *d_dst = (char *)dev->pool.alloc(off);
	cl_int *d_aux = (cl_int *)dev->pool.alloc(sizeof(cl_int) * aux_ints);
	cl_ulong *d_dimptr = (cl_ulong *)dev->pool.alloc(sizeof(cl_ulong) * dimptr.size());
	std::vector<cl_int> aux(aux_ints, 0);
	memcpy(aux.data(), src_depth, sizeof(cl_int) * ncols);
	memcpy(aux.data() + ncols, src_colidx, sizeof(cl_int) * ncols);
	/* a negative oid: that type, but no zone map wanted (ingest_minmax skips oid 0) */
	bool	any_zone_map = false;
	for (int i = 0; i < ncols; i++)
	{
		aux[2 * ncols + i] = (type_oids[i] > 0 ? type_oids[i] : 0);
		any_zone_map = any_zone_map || (type_oids[i] > 0);
	}
	hipStream_t stream = dev->streams[0];
	strom_dstore *result = nullptr;
	do {
		if (!d_dst || !d_aux || !d_dimptr)
		{
			*p_errcode = StromError_OutOfMemory;
			break;
		}
		if (hipMemcpyAsync(d_dst, hbuf.data(), hbuf.size(), hipMemcpyHostToDevice, stream) != hipSuccess ||
			hipMemcpyAsync(d_dimptr, dimptr.data(), sizeof(cl_ulong) * dimptr.size(),
						   hipMemcpyHostToDevice, stream) != hipSuccess ||
			hipMemcpyAsync(d_aux, aux.data(), sizeof(cl_int) * aux_ints, hipMemcpyHostToDevice, stream) != hipSuccess)
		{
			*p_errcode = StromError_HipInternal;
			break;
		}
		void	   *a_khj = task->main_devptr;
		const void *a_km = tbl->d_kmhash;
		const void *a_kds = outer->devptr;
		const void *a_toast = nullptr;
		void	   *a_dst = d_dst;
		const void *a_md = d_aux;
		const void *a_mc = d_aux + ncols;
		const void *a_oids = d_aux + 2 * ncols;
		void	   *a_flags = d_aux + 3 * ncols;
		const void *a_hjidx = tbl->d_index;
		const void *a_dimptr = d_dimptr;
		void	   *args[] = { &a_khj, &a_km, &a_kds, &a_toast, &a_dst, &a_md, &a_mc, &a_flags, &a_hjidx, &a_dimptr };
		void	   *args_mm[] = { &a_dst, &a_oids };
		void	   *args_fin[] = { &a_dst, &a_oids, &a_flags };
		unsigned	grid = (unsigned)std::min<size_t>(((size_t)nitems + 255) / 256,
													  (size_t)dev->prop.multiProcessorCount * 8);
		/* (the projection takes 4 records per thread and turn: HASHJOIN_PROJ_ROWS) */
		unsigned	pgrid = (unsigned)std::min<size_t>(((size_t)nitems + 1023) / 1024,
													   (size_t)dev->prop.multiProcessorCount * 8);
		if (grid > 0 &&
			(hipModuleLaunchKernel(fn_proj, pgrid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess ||
			 (any_zone_map &&
			  hipModuleLaunchKernel(fn_mm, std::min(grid, (unsigned)dev->prop.multiProcessorCount),
									(unsigned)ncols, 1, 256, 1, 1, 0, stream, args_mm, nullptr) != hipSuccess)))
		{
			*p_errcode = StromError_HipInternal;
			break;
		}
		if (hipModuleLaunchKernel(fn_fin, 1, 1, 1, 64, 1, 1, 0, stream, args_fin, nullptr) != hipSuccess)
		{
			*p_errcode = StromError_HipInternal;
			break;
		}
		cl_int	failed = 0;
		result = new strom_dstore{d_dst, off, outer->dindex, true, {}};
		if (hipMemcpyAsync(&result->head, d_dst, offsetof(kern_data_store, colmeta),
						   hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipMemcpyAsync(&failed, d_aux + 4 * (size_t)ncols, sizeof(cl_int),
						   hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipStreamSynchronize(stream) != hipSuccess)
		{
			delete result;
			result = nullptr;
			*p_errcode = StromError_HipInternal;
			break;
		}
		if (failed)
		{
			/* a mapped source column is not as wide as its destination type */
			delete result;
			result = nullptr;
			*p_errcode = StromError_DataStoreCorruption;
			break;
		}
	} while (0);
	if (d_aux) dev->pool.release(d_aux);
	if (d_dimptr) dev->pool.release(d_dimptr);
	if (!result && d_dst) dev->pool.release(d_dst);
	return result;
	STROM_ABI_CATCH(nullptr, p_errcode)
}
