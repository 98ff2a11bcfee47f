/*
 * runtime.h -- internals of the HIP device runtime (libstrom_hip.so)
 *
 * Stands where opencl_serv.c / opencl_devprog.c / opencl_devinfo.c /
 * opencl_entry.c and the clserv_* halves of the three operators stand in
 * the reference.
 */
#ifndef STROM_RUNTIME_H
#define STROM_RUNTIME_H

#include <hip/hip_runtime.h>
/* from <hip/hip_ext.h> (whose C++ half needs hipcc; this library is built with g++) */
extern "C" hipError_t hipExtModuleLaunchKernel(hipFunction_t f, uint32_t globalWorkSizeX,
											   uint32_t globalWorkSizeY, uint32_t globalWorkSizeZ,
											   uint32_t localWorkSizeX, uint32_t localWorkSizeY,
											   uint32_t localWorkSizeZ, size_t sharedMemBytes,
											   hipStream_t hStream, void **kernelParams, void **extra,
											   hipEvent_t startEvent, hipEvent_t stopEvent, uint32_t flags);
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "strom_hip.h"

namespace strom {

struct Device;

/* ---- pooled device memory: no hipMalloc/hipFree per chunk (the reference
 * creates and releases 2-6 cl_mem objects per request) ------------------ */
class BufferPool {
public:
	void   *alloc(size_t nbytes);
	void	release(void *ptr);
	void	drain();
	int		hip_id = 0;
private:
	std::mutex						lock_;
	std::map<size_t, std::vector<void *>>	free_;	/* by size class */
	std::map<void *, size_t>		live_;
	static size_t size_class(size_t nbytes);
};

/* ---- pinned host staging blocks for the small per-request DMA (request
 * head down, result head up): a copy from/to pageable memory would make
 * hipMemcpyAsync synchronous ------------------------------------------- */
class PinnedPool {
public:
	static const size_t BLOCK = 16384;
	char   *alloc();
	void	release(char *blk);
private:
	std::mutex				lock_;
	std::vector<char *>		free_;
};

struct Program {
	strom_devprog_key	key = 0;
	std::string			source;
	int32_t				extra_flags = 0;
	std::mutex			lock;
	std::condition_variable cond;
	int					state = STROM_DEVPROG_PENDING;
	int					refcnt = 1;
	std::string			errmsg;
	std::vector<char>	code;			/* gfx950 code object */
	double				build_usec = 0;
	std::map<int, hipModule_t> modules;	/* per device index */
	std::vector<std::function<void()>> parked;	/* requests waiting for the build */

	hipFunction_t	get_function(Device *dev, const char *name, int *p_errcode);
};

struct strom_task_impl;

struct Device {
	int					dindex = 0;
	int					hip_id = 0;
	hipDeviceProp_t		prop;
	std::vector<hipStream_t> streams;
	/* resident-chunk requests: the request head goes down on copy_in and
	 * the result head comes back on copy_out, so that on streams[0] one
	 * chunk's kernel follows the previous one's without a DMA in between */
	hipStream_t			copy_in = nullptr, copy_out = nullptr;
	/* GpuPreAgg over resident chunks: the slab merge (and the status read-back)
	 * of chunk k runs here while chunk k+1 is folded on streams[0] */
	hipStream_t			merge_stream = nullptr;
	std::atomic<unsigned> next_stream{0};
	BufferPool			pool;
	PinnedPool			pinned;
	std::mutex			ev_lock;
	std::vector<hipEvent_t> ev_free_timing, ev_free_plain;
	/* completion side: one thread per device plays the role of the OpenCL
	 * runtime's callback thread (clSetEventCallback -> clserv_respond_*) */
	std::thread			completer;
	std::mutex			cq_lock;
	std::condition_variable cq_cond;
	std::deque<strom_task_impl *> cq;
	bool				shutting_down = false;
	size_t				inflight = 0;
	/* requests that are a SEQUENCE of device steps with host decisions in between
	 * (the per-chunk GpuPreAgg message: key range -> table geometry -> fold -> partial
	 * rows) run on these threads, started on first use -- the reference's server threads
	 * (opencl_serv.c:76-90, one per CPU there); the submitter only queues.  Requests are
	 * independent of each other (every message has a session of its own), so several
	 * are in their host-side steps at once: one thread answered ~1500 messages a second
	 * whatever was in flight (profiles/r02_chunk_message_probe.txt) */
	std::vector<std::thread> workers;		/* STROM_HIP_NUM_WORKERS, default 4 */
	std::mutex			wq_lock;
	std::condition_variable wq_cond;
	std::deque<std::function<void()>> wq;
	bool				worker_stop = false;

	hipStream_t pick_stream() { return streams[next_stream++ % streams.size()]; }
};

}	/* namespace strom */

struct strom_dstore {
	void	   *devptr;
	size_t		length;
	int			dindex;
	bool		owned;
	kern_data_store head;		/* host snapshot of the fixed head (no colmeta) */
	/* KDS_FORMAT_COLUMN: host snapshot of the column directory (NULL bitmaps
	 * present?  zone maps), fetched on first use by strom::dstore_coldir() */
	std::shared_ptr<std::vector<kern_coldir>> coldir;
};

/* device-resident kern_row_map (a finished GpuScan's results, in place) */
struct strom_rowmap {
	void	   *buffer;			/* pool allocation that holds the map */
	void	   *devptr;			/* the kern_row_map inside it */
	uint32_t	nvalids;
	int			dindex;
};

struct strom_task {
	/* the public handle is the impl itself */
};

namespace strom {

struct strom_task_impl : public strom_task {
	Device	   *dev = nullptr;
	hipStream_t	stream = nullptr;
	int			errcode = 0;
	strom_perfmon pfm;
	strom_done_cb done = nullptr;
	void	   *done_arg = nullptr;
	/* events: [0] start, [1] after send, [2] after main kernel(s),
	 * [3] after recv; extra pairs for prep/proj kernels */
	hipEvent_t	ev[8] = {};
	int			nev = 0;
	bool		has_ev_prep = false, has_ev_proj = false;
	/* GpuPreAgg, piped: [0] start [1] head sent (copy-in stream), [2] fold begins [3] fold done
	 * (streams[0]), [4] merge done [5] status received (merge stream) */
	bool		ev_preagg_piped = false;
	/* device buffers to hand back to the pool at completion */
	std::vector<void *> devbufs;
	std::vector<char *> pinned_blocks;
	void	   *main_devptr = nullptr;	/* kern_gpuscan / kern_hashjoin image */
	bool		keep_main = false;		/* released by strom_task_wait */
	/* GpuScan with STROM_RESULTS_ON_DEVICE: where the kern_resultbuf sits in
	 * main_devptr and how many rows it holds (strom_rowmap_from_task) */
	size_t		res_offset = 0;
	uint32_t	res_nitems = 0;
	bool		res_is_scan = false;
	bool		res_is_join = false;		/* GpuHashJoin likewise (strom_hashjoin_project_column) */
	/* operator-specific second half, runs on the completer thread after the
	 * first event fired; may issue further copies on 'stream' and must
	 * leave the stream idle when it returns */
	std::function<void(strom_task_impl *)> finish;
	/* set by finish(): the request is not over -- its device steps are issued once more
	 * (GpuPreAgg: the checked fold of a chunk whose integer sums were not proven to stay
	 * in range).  The completer hands it to a worker thread; whatever it queues ends in
	 * task_enqueue() / task_fail() again, and done() still runs exactly once. */
	std::function<void()> retry;
	/* run once when the request completes, however it ends, before done(): what a request
	 * made for itself on the way (GpuPreAgg: the scratch session of an exact hashed fold) */
	std::vector<std::function<void()>> at_complete;
	/* waiter side */
	std::mutex	lock;
	std::condition_variable cond;
	bool		completed = false;
	/* a request that could not be started (task_fail): the completer thread
	 * drains its streams, skips the timing events and completes it */
	bool		failed = false;
	/* done() is running on the completer thread; strom_task_wait() from inside
	 * it must not wait for 'completed' (set after done() returns) */
	bool		cb_running = false;
	bool		released_in_cb = false;
	std::thread::id cb_thread;
	std::chrono::steady_clock::time_point t_enqueue;
};

/* runtime.cpp */
/* column directory of a resident COLUMN chunk (host snapshot, cached); NULL for other formats */
std::shared_ptr<std::vector<kern_coldir>> dstore_coldir(strom_dstore *ds);
/* block until the request is over (also true inside its own done() callback) */
void		task_wait_completed(strom_task_impl *task);
Device	   *get_device(int dindex);
/* gpuhashjoin.cpp, for consumers of join results (gpupreagg.cpp) */
int			hashjoin_table_dimcol(strom_hashjoin_table *tbl, int col, int attlen, void **p_values, void **p_isnull);
/* narrow form of the slot records (strom_hashjoin.h): reclen 2 or 4, or 0 = not available */
struct dimrec_narrow {
	unsigned	reclen = 0;
	void	   *recs = nullptr;
	cl_uint		shift[16] = {};
	cl_uint		mask[16] = {};
	cl_long		vmin[16] = {};
};
/* narrowable (n flags, may be NULL): column i holds integers (not float bits); narrow (may be
 * NULL) receives the narrow records when every column is narrowable and the fields fit 32 bits */
int			hashjoin_table_dimrecs(strom_hashjoin_table *tbl, int n, const int *cols, const int *attlens,
								   unsigned *offsets, void **p_recs, unsigned *p_reclen,
								   const int *narrowable = nullptr, dimrec_narrow *narrow = nullptr);
int			hashjoin_table_direct_info(strom_hashjoin_table *tbl, cl_long *p_key_min, cl_uint *p_nslots,
									   int *p_outer_key_attno, int *p_dindex, int *p_has_outer_qual = nullptr);
/* gpupreagg.cpp, for the RCCL merge (parallel.cpp): mirrors preagg_merge_spec of
 * devlib/strom_merge.h */
struct gpupreagg_merge_plan {
	struct {
		cl_uint		ngroups;
		cl_uint		naggs;
		cl_uint		op[31];
		cl_uint		__pad;
		cl_ulong	vals_off[31];
		cl_ulong	hi_off[31];
		cl_uint		mid_idx[31];
		cl_uint		__pad2;
	} spec;
	char	   *table;
	size_t		table_bytes;
	cl_uint		nmid;				/* integer sums: lanes of ngroups words the merge needs as scratch */
	int			dindex;
};
/* may 'src' be merged into 'dst' (strom_gpupreagg_merge)?  Same program, same targets, same device,
 * both hashed or both dense with the same group slots */
bool		gpupreagg_sessions_mergeable(strom_gpupreagg *dst, strom_gpupreagg *src);
int			gpupreagg_get_merge_plan(strom_gpupreagg *sess, gpupreagg_merge_plan *plan);
int			gpupreagg_get_census(strom_gpupreagg *sess, void **p_bitmap, cl_uint *p_nbits, int *p_dindex);
/* hashed sessions (gpupreagg.cpp): the groups packed on the device (records of *p_reclen bytes in a
 * pool buffer the caller releases with gpupreagg_hash_release), and packed groups merged into the table */
bool		gpupreagg_is_hashed(strom_gpupreagg *sess, int *p_dindex);
int			gpupreagg_hash_export_device(strom_gpupreagg *sess, char **p_recs, cl_uint *p_count, size_t *p_reclen,
										 cl_ulong *p_sum_bound = nullptr);
/* ... packed by owner (h_counts[nparts], nparts <= 64) for the hash-partitioned exchange */
int			gpupreagg_hash_export_parts_device(strom_gpupreagg *sess, cl_uint nparts, char **p_recs, cl_uint *h_counts,
											   size_t *p_reclen, cl_ulong *p_sum_bound);
void		gpupreagg_hash_release(strom_gpupreagg *sess, char *recs);
/* incoming_sum_bound: see gpupreagg.cpp -- the integer sums' range proof of the merge; this value:
 * ONE segment of pairwise different keys (another session's export), proven group by group */
const cl_ulong GPUPREAGG_IMPORT_EXACT = ~0UL;
int			gpupreagg_hash_import_device(strom_gpupreagg *sess, const char *d_recs, cl_uint seg_len, cl_uint nsegs,
										 const cl_uint *h_counts, cl_uint skip_seg, cl_ulong incoming_sum_bound);
/* parallel.cpp: a kernel of the fixed-function program (devlib/strom_merge.h: the merge's prepare /
 * finish / apply steps, the dense partial-row export, the streaming-read probe) */
hipFunction_t fixed_function(Device *dev, const char *name, int *p_errcode);
int			num_devices();
Program	   *lookup_program(strom_devprog_key key);
/* text / character(n) values are addresses of varlena datums (strom_textlib.h): a program that
 * uses them reads the chunks that hold such datums -- heap tuples (ROW / ROW_FLAT) and the heap
 * area of a COLUMN chunk (strom_kds.h); a TUPSLOT chunk holds by-value datums */
inline bool
program_accepts_format(const Program *prog, cl_int format)
{
	return !(prog->extra_flags & DEVTYPE_IS_VARLENA) ||
		format == KDS_FORMAT_ROW || format == KDS_FORMAT_ROW_FLAT || format == KDS_FORMAT_COLUMN;
}
bool		perfmon_enabled();
strom_task_impl *task_create(Device *dev, strom_done_cb done, void *arg);
void		task_enqueue(strom_task_impl *task);
void		task_fail(strom_task_impl *task, int errcode);
/* queue 'fn' for the device's worker thread (see Device::worker) */
void		device_run_async(Device *dev, std::function<void()> fn);
hipEvent_t	task_event(strom_task_impl *task, hipStream_t stream = nullptr);
/* an event of the task that is NOT recorded here: hipExtModuleLaunchKernel attaches a
 * start / stop pair to the kernel's own dispatch packet (its timestamps are the kernel's),
 * which spares the two barrier packets hipEventRecord would put around the launch */
hipEvent_t	task_event_slot(strom_task_impl *task);
/* launch with the kernel's own start / stop events when the knob allows it */
bool		use_ext_launch(void);
int			hip_errcode(hipError_t rc, const char *what);
/* run 'fn' now if the program is ready, park it if the build is in
 * flight; returns the program state */
int			program_run_or_park(Program *prog, std::function<void()> fn);

/*
 * No C++ exception crosses the C ABI (a PostgreSQL backend links this library:
 * an escaping exception is std::terminate -> abort()).  The host code throws
 * nothing itself; what can still surface is std::bad_alloc from the containers.
 * Entry points whose allocations scale with their input run inside this guard.
 */
#define STROM_ABI_TRY		try {
#define STROM_ABI_CATCH(retval, p_errcode)									\
	} catch (const std::bad_alloc &) {										\
		int *__pe = (p_errcode);											\
		if (__pe) *__pe = StromError_OutOfMemory;							\
		return retval;														\
	} catch (...) {															\
		int *__pe = (p_errcode);											\
		if (__pe) *__pe = StromError_HipInternal;							\
		return retval;														\
	}

#define STROM_HIP_CHECK(call, task)										\
	do {																\
		hipError_t __rc = (call);										\
		if (__rc != hipSuccess)											\
		{																\
			(task)->errcode = hip_errcode(__rc, #call);					\
			goto hip_error;												\
		}																\
	} while (0)

}	/* namespace strom */
#endif
