/*
 * gpuscan.cpp -- host side of one GpuScan request
 *
 * Role in the reference: clserv_process_gpuscan (gpuscan.c:1895-2182) and
 * clserv_respond_gpuscan (1760-1888): look the program up (park if the
 * build is in flight), pick a device, send {kern_parambuf, kern_resultbuf
 * head} and the chunk, run the qual kernel, read the result buffer back,
 * fill errcode + perfmon, reply.
 *
 * Differences that matter for speed: buffers come from a per-device pool
 * instead of clCreateBuffer/clReleaseMemObject per chunk; a chunk can
 * already be resident (strom_dstore); only results[0..nitems) travels
 * back, not nrooms entries.
 */
#include <cstring>
#include <cstdio>

#include "runtime.h"

using namespace strom;

namespace {

struct gpuscan_request {
	strom_devprog_key	key;
	kern_gpuscan	   *kgpuscan;		/* host image */
	const kern_data_store *kds;			/* host chunk or NULL */
	strom_dstore	   *kds_dev;
	const kern_row_map *krowmap;
	strom_rowmap	   *rowmap_dev;		/* device-resident row map (chained operators) */
	uint32_t			flags;
	uint32_t			nrows;			/* rows the kernel walks */
	uint32_t			format;
};

#define REQ_CHECK(call, what)												\
	do {																	\
		hipError_t __rc = (call);											\
		if (__rc != hipSuccess)												\
		{																	\
			task_fail(task, hip_errcode(__rc, what));						\
			return;															\
		}																	\
	} while (0)

void
gpuscan_launch(strom_task_impl *task, Program *prog, gpuscan_request req)
{
	Device	   *dev = task->dev;
	kern_gpuscan *kgs = req.kgpuscan;
	kern_resultbuf *kres_host = KERN_GPUSCAN_RESULTBUF(kgs);
	size_t		head_len = KERN_GPUSCAN_DMASEND_LENGTH(kgs);
	size_t		total_len = KERN_GPUSCAN_LENGTH(kgs);
	size_t		res_offset = KERN_GPUSCAN_PARAMBUF_LENGTH(kgs);
	int			errcode = 0;

	(void)hipSetDevice(dev->hip_id);
	if (prog->state != STROM_DEVPROG_READY)
	{
		task_fail(task, StromError_ProgramBuildFailure);
		return;
	}
	task->pfm.time_kern_build = (cl_ulong)prog->build_usec;
	/*
	 * A resident chunk needs no bulk DMA, so nothing is gained by letting
	 * two bandwidth-bound kernels share the chip: such requests run in
	 * order on stream 0 (their event pairs then time the kernel alone).
	 * Requests that upload their chunk rotate over the other streams so
	 * that the next chunk's DMA overlaps this chunk's kernel.
	 */
	if (req.kds_dev || dev->streams.size() < 2)
		task->stream = dev->streams[0];
	else
		task->stream = dev->streams[1 + dev->next_stream++ % (dev->streams.size() - 1)];

	bool		use_column = (req.format == KDS_FORMAT_COLUMN && req.krowmap == nullptr && req.rowmap_dev == nullptr);
	hipFunction_t fn = prog->get_function(dev, use_column ? "gpuscan_qual_column"
										  : "gpuscan_qual_generic", &errcode);
	if (!fn)
	{
		task_fail(task, errcode);
		return;
	}
	char	   *d_kgs = (char *)dev->pool.alloc(total_len);
	if (!d_kgs)
	{
		task_fail(task, StromError_OutOfMemory);
		return;
	}
	task->main_devptr = d_kgs;
	task->keep_main = ((req.flags & STROM_RESULTS_ON_DEVICE) != 0);

	/* request head goes down (and the result head comes back) through a
	 * pinned staging block so that both copies are truly asynchronous */
	char	   *stage = (head_len + 64 <= PinnedPool::BLOCK ? dev->pinned.alloc() : nullptr);
	if (stage)
	{
		task->pinned_blocks.push_back(stage);
		memcpy(stage, kgs, head_len);
	}
	/*
	 * resident chunk: three streams.  The small copies would otherwise sit
	 * between two kernels of consecutive requests on the same stream (two
	 * DMA round trips, ~20 us per 200 us kernel).
	 */
	bool		piped = (req.kds_dev != nullptr && stage != nullptr && dev->copy_in && dev->copy_out);
	hipStream_t	s_in = (piped ? dev->copy_in : task->stream);
	hipStream_t	s_out = (piped ? dev->copy_out : task->stream);

	task_event(task, s_in);								/* ev[0] */
	REQ_CHECK(hipMemcpyAsync(d_kgs, stage ? (void *)stage : (void *)kgs, head_len,
							 hipMemcpyHostToDevice, s_in),
			  "send kern_gpuscan");
	task->pfm.num_dma_send++;
	task->pfm.bytes_dma_send += head_len;

	const void *d_kds;
	if (req.kds_dev)
		d_kds = req.kds_dev->devptr;
	else
	{
		size_t	kds_len = req.kds->length;
		if (req.kds->format == KDS_FORMAT_ROW)
			kds_len = KERN_DATA_STORE_ROWBLOCK_OFFSET(req.kds) +
				(size_t)BLCKSZ * req.kds->nblocks;
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
	if (!req.rowmap_dev && req.krowmap && req.krowmap->nvalids >= 0)
	{
		size_t	len = offsetof(kern_row_map, rindex) + sizeof(cl_int) * (size_t)req.krowmap->nvalids;
		void   *p = dev->pool.alloc(len);
		if (!p)
		{
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		task->devbufs.push_back(p);
		REQ_CHECK(hipMemcpyAsync(p, req.krowmap, len, hipMemcpyHostToDevice, s_in),
				  "send kern_row_map");
		task->pfm.num_dma_send++;
		task->pfm.bytes_dma_send += len;
		d_rowmap = p;
	}
	/* piped requests with perfmon: the kernel carries its own start / stop events
	 * (hipExtModuleLaunchKernel) instead of two recorded ones around it */
	bool		ext_launch = (piped && task->pfm.enabled && use_ext_launch());
	hipEvent_t	ev_begin = nullptr, ev_done = nullptr;
	if (piped)
	{
		hipEvent_t sent = task_event(task, s_in);		/* ev[1]: head is down */
		REQ_CHECK(hipStreamWaitEvent(task->stream, sent, 0), "wait for the request head");
		/* the kernel is timed from the moment the main stream gets to it */
		ev_begin = (ext_launch ? task_event_slot(task) : task_event(task));	/* ev[2] */
		if (ext_launch)
			ev_done = task_event_slot(task);			/* ev[3] */
		if (ext_launch && (!ev_begin || !ev_done))
		{
			task_fail(task, StromError_HipInternal);
			return;
		}
	}
	else
		task_event(task);								/* ev[1] */

	/*
	 * grid: enough persistent work-groups to fill every CU at the
	 * occupancy the kernel's LDS stage allows, never more than tiles
	 * (clserv_compute_workgroup_size's job, opencl_devinfo.c:1126-1231)
	 */
	{
		int		block = 256;
		if (const char *v = getenv("STROM_GPUSCAN_BLOCK"))
			block = atoi(v);
		int		quads = 1;
		if (const char *v = getenv("STROM_GPUSCAN_QUADS"))
			quads = atoi(v);
		size_t	tile_rows = (size_t)block * 4 * quads;
		size_t	ntiles = (req.nrows + tile_rows - 1) / tile_rows;
		int		per_cu = 0;
		if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, 0) != hipSuccess
			|| per_cu < 1)
			per_cu = 1;
		if (const char *v = getenv("STROM_GPUSCAN_BLOCKS_PER_CU"))
			per_cu = std::max(1, atoi(v));
		size_t	grid = (size_t)dev->prop.multiProcessorCount * per_cu;
		if (grid > ntiles)
			grid = ntiles;
		if (grid < 1)
			grid = 1;
		void   *a_kgs = d_kgs;
		const void *a_kds = d_kds;
		const void *a_toast = nullptr;
		const void *a_map = d_rowmap;
		void   *args_col[] = { &a_kgs, &a_kds };
		void   *args_gen[] = { &a_kgs, &a_kds, &a_toast, &a_map };
		if (ext_launch)
			REQ_CHECK(hipExtModuleLaunchKernel(fn, (uint32_t)(grid * block), 1, 1, block, 1, 1, 0, task->stream,
											   use_column ? args_col : args_gen, nullptr, ev_begin, ev_done, 0),
					  "launch gpuscan kernel");
		else
			REQ_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, block, 1, 1, 0, task->stream,
											use_column ? args_col : args_gen, nullptr),
					  "launch gpuscan kernel");
		task->pfm.num_kern_exec++;
	}
	hipEvent_t kernel_done = (ext_launch ? ev_done : task_event(task));	/* ev[2] (piped: ev[3]) */
	if (piped)
		REQ_CHECK(hipStreamWaitEvent(s_out, kernel_done, 0), "wait for the kernel");
	/* result head first; results[0..nitems) follow once nitems is known */
	char	   *stage_res = (stage ? stage + STROMALIGN(head_len) : nullptr);
	REQ_CHECK(hipMemcpyAsync(stage_res ? (void *)stage_res : (void *)kres_host,
							 d_kgs + res_offset, offsetof(kern_resultbuf, results),
							 hipMemcpyDeviceToHost, s_out),
			  "recv kern_resultbuf head");
	task->pfm.num_dma_recv++;
	task->pfm.bytes_dma_recv += offsetof(kern_resultbuf, results);
	task_event(task, s_out);							/* ev[3] (piped: ev[4]) */
	task->has_ev_prep = piped;		/* tells the completer about the extra event */

	bool	results_on_device = (req.flags & STROM_RESULTS_ON_DEVICE) != 0;
	task->res_offset = res_offset;
	task->res_is_scan = true;
	task->finish = [kres_host, d_kgs, res_offset, results_on_device, stage_res](strom_task_impl *t)
	{
		if (stage_res)
			memcpy(kres_host, stage_res, offsetof(kern_resultbuf, results));
		if (StromErrorIsSignificant(kres_host->errcode))
			t->errcode = kres_host->errcode;
		t->res_nitems = kres_host->nitems;
		if (results_on_device || kres_host->nitems == 0)
			return;
		size_t	len = sizeof(cl_int) * (size_t)kres_host->nitems;
		auto	t0 = std::chrono::steady_clock::now();
		hipStream_t	st = (t->has_ev_prep ? t->dev->copy_out : t->stream);
		hipError_t rc = hipMemcpyAsync(kres_host->results,
									   d_kgs + res_offset + offsetof(kern_resultbuf, results),
									   len, hipMemcpyDeviceToHost, st);
		if (rc == hipSuccess)
			rc = hipStreamSynchronize(st);
		if (rc != hipSuccess)
			t->errcode = hip_errcode(rc, "recv results[]");
		t->pfm.num_dma_recv++;
		t->pfm.bytes_dma_recv += len;
		t->pfm.time_dma_recv += (cl_ulong)std::chrono::duration<double, std::micro>
			(std::chrono::steady_clock::now() - t0).count();
	};
	task_enqueue(task);
}

}	/* namespace */

static strom_task *
submit_gpuscan_common(strom_devprog_key key,
					  kern_gpuscan *kgpuscan,
					  const kern_data_store *kds,
					  strom_dstore *kds_dev,
					  const kern_row_map *krowmap,
					  strom_rowmap *rowmap_dev,
					  uint32_t flags,
					  strom_done_cb done, void *arg,
					  int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	Program *prog = lookup_program(key);
	if (!prog || !kgpuscan || (!kds) == (!kds_dev) ||
		(rowmap_dev && (!kds_dev || krowmap || rowmap_dev->dindex != kds_dev->dindex)))
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	int		dindex = kds_dev ? kds_dev->dindex : strom_device_schedule();
	Device *dev = get_device(dindex);
	if (!dev)
	{
		*p_errcode = StromError_ServerNotReady;
		return nullptr;
	}
	gpuscan_request req;
	req.key = key;
	req.kgpuscan = kgpuscan;
	req.kds = kds;
	req.kds_dev = kds_dev;
	req.krowmap = (krowmap && krowmap->nvalids >= 0) ? krowmap : nullptr;
	req.rowmap_dev = rowmap_dev;
	req.flags = flags;
	/*
	 * the kernel needs nitems / format to size its grid; a resident chunk
	 * keeps a host snapshot of its head
	 */
	kern_data_store head;
	if (kds)
		memcpy(&head, kds, offsetof(kern_data_store, colmeta));
	else
		head = kds_dev->head;
	req.format = head.format;
	if (!program_accepts_format(prog, head.format))
	{
		*p_errcode = StromError_BadRequestMessage;		/* text columns live in heap tuples */
		return nullptr;
	}
	req.nrows = (rowmap_dev ? rowmap_dev->nvalids
				 : req.krowmap ? (uint32_t)req.krowmap->nvalids : head.nitems);
	kern_resultbuf *kres = KERN_GPUSCAN_RESULTBUF(kgpuscan);
	if (kres->nrels != 1 || kres->nrooms < req.nrows)
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	/* "kernel code assumes all the fields shall be initialized to zero" */
	kres->nitems = 0;
	kres->errcode = StromError_Success;

	strom_task_impl *task = task_create(dev, done, arg);
	program_run_or_park(prog, [task, prog, req]() { gpuscan_launch(task, prog, req); });
	return task;
	STROM_ABI_CATCH(nullptr, p_errcode)
}

extern "C" strom_task *
strom_submit_gpuscan(strom_devprog_key key,
					 kern_gpuscan *kgpuscan,
					 const kern_data_store *kds,
					 strom_dstore *kds_dev,
					 const kern_row_map *krowmap,
					 uint32_t flags,
					 strom_done_cb done, void *arg,
					 int *p_errcode)
{
	return submit_gpuscan_common(key, kgpuscan, kds, kds_dev, krowmap, nullptr, flags, done, arg, p_errcode);
}

extern "C" strom_task *
strom_submit_gpuscan_mapped(strom_devprog_key key,
							kern_gpuscan *kgpuscan,
							strom_dstore *kds_dev,
							strom_rowmap *rowmap,
							uint32_t flags,
							strom_done_cb done, void *arg,
							int *p_errcode)
{
	if (!rowmap)
	{
		if (p_errcode)
			*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	return submit_gpuscan_common(key, kgpuscan, nullptr, kds_dev, nullptr, rowmap, flags, done, arg, p_errcode);
}

/* ------------------------------------------------------------------ *
 * device-resident row maps: the hand-over between chained operators
 * ------------------------------------------------------------------ */
namespace {
const char *rowmap_source =
	"#include \"strom_kds.h\"\n"
	"#include \"strom_common.h\"\n"
	"#include \"strom_rowmap.h\"\n";
}

extern "C" strom_rowmap *
strom_rowmap_from_task(strom_task *handle, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	strom_task_impl *task = static_cast<strom_task_impl *>(handle);
	if (!task)
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	task_wait_completed(task);
	if (!task->res_is_scan || !task->keep_main)
	{
		*p_errcode = (task->errcode ? task->errcode
					  : StromError_BadRequestMessage);	/* not a GpuScan with STROM_RESULTS_ON_DEVICE */
		return nullptr;
	}
	if (task->errcode != 0 || !task->main_devptr)
	{
		*p_errcode = (task->errcode ? task->errcode : StromError_BadRequestMessage);
		return nullptr;
	}
	Device *dev = task->dev;
	static strom_devprog_key key = strom_get_devprog_key(rowmap_source, 0);
	if (strom_lookup_device_program(key, 1) != STROM_DEVPROG_READY)
	{
		*p_errcode = StromError_ProgramBuildFailure;
		return nullptr;
	}
	int		errcode = 0;
	(void)hipSetDevice(dev->hip_id);
	hipFunction_t fn = lookup_program(key)->get_function(dev, "rowmap_from_results", &errcode);
	if (!fn)
	{
		*p_errcode = errcode;
		return nullptr;
	}
	hipStream_t stream = dev->streams[0];
	cl_int	   *d_status = (cl_int *)dev->pool.alloc(sizeof(cl_int));
	cl_int		h_status = 0;
	if (!d_status)
	{
		*p_errcode = StromError_OutOfMemory;
		return nullptr;
	}
	void	   *a_kres = (char *)task->main_devptr + task->res_offset;
	cl_uint		a_nitems = task->res_nitems;
	void	   *a_status = d_status;
	void	   *args[] = { &a_kres, &a_nitems, &a_status };
	unsigned	grid = (unsigned)std::max<size_t>(1, std::min<size_t>(((size_t)a_nitems + 255) / 256,
																	  (size_t)dev->prop.multiProcessorCount * 8));
	bool		ok = (hipMemsetAsync(d_status, 0, sizeof(cl_int), stream) == hipSuccess &&
					  hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess &&
					  hipMemcpyAsync(&h_status, d_status, sizeof(cl_int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
					  hipStreamSynchronize(stream) == hipSuccess);
	dev->pool.release(d_status);
	if (!ok)
	{
		*p_errcode = StromError_HipInternal;
		return nullptr;
	}
	if (h_status != 0)
	{
		/* rows to re-check on the CPU: the ids were partly rewritten, the
		 * buffer is of no further use -- the caller redoes this chunk on the
		 * host path */
		*p_errcode = h_status;
		return nullptr;
	}
	strom_rowmap *map = new strom_rowmap();
	map->buffer = task->main_devptr;
	map->devptr = (char *)a_kres + offsetof(kern_resultbuf, results) - sizeof(cl_int);
	map->nvalids = a_nitems;
	map->dindex = dev->dindex;
	task->main_devptr = nullptr;		/* ownership moved; strom_task_wait still frees the task */
	return map;
	STROM_ABI_CATCH(nullptr, p_errcode)
}

extern "C" uint32_t
strom_rowmap_nvalids(strom_rowmap *map) { return map ? map->nvalids : 0; }

extern "C" void *
strom_rowmap_devptr(strom_rowmap *map) { return map ? map->devptr : nullptr; }

extern "C" void
strom_rowmap_release(strom_rowmap *map)
{
	if (!map)
		return;
	Device *dev = get_device(map->dindex);
	if (dev && map->buffer)
		dev->pool.release(map->buffer);
	delete map;
}
