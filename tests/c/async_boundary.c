/*
 * async_boundary.c -- the callback half of the C ABI, driven from plain C.
 *
 * What a PostgreSQL backend does with the reference (pgstrom_message protocol,
 * pg_strom.h:238-248): fill a request, enqueue it (pgstrom_enqueue_message,
 * mqueue.c:140-183), get the reply on a runtime thread (clserv_respond_gpuscan
 * -> pgstrom_reply_message, gpuscan.c:1760-1888), release the message
 * (pgstrom_put_message, mqueue.c:533-554).  The device program may still be
 * building when the first requests arrive: they are parked on the program and
 * re-issued when the build ends (opencl_devprog.c:291-527).
 *
 * This program links libstrom_hip.so, submits GpuScan / GpuHashJoin /
 * GpuPreAgg requests with a real strom_done_cb from several threads, several
 * in flight, with a COLD program cache (the first requests park behind the
 * hiprtc build), and asserts for every request:
 *   - the callback ran exactly once, on a thread that is not a submitter,
 *   - errcode / kern_resultbuf / results[] were final when it ran,
 *   - a request that cannot start (broken program) still answers through
 *     the callback, on a runtime thread, with the build failure code,
 *   - StromError_DataStoreNoSpace carries the room a retry needs, and the
 *     retry with exactly that room succeeds (gpuhashjoin.c:4330-4425),
 *   - strom_task_wait() from inside a callback and from another thread both
 *     release the handle.
 * Expected values are computed right here in C with the same predicate.
 * Exit code 0 = all assertions held; prints one "ok: ..." line per section.
 * Test infrastructure (tests/test_async_c_gpu.py builds nothing: build() does).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "strom_hip.h"
#include "strom_datastore.h"

#define CHECK(cond)															\
	do {																	\
		if (!(cond))														\
		{																	\
			fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);	\
			exit(1);														\
		}																	\
	} while (0)

#define NROWS		200000
#define NTHREADS	3
#define NREQ_PER_THREAD	4

static int32_t	col_a[NROWS];
static double	col_b[NROWS];
static int32_t	col_fk[NROWS];
static uint8_t	null_a[NROWS];

static uint64_t
splitmix(uint64_t *s)
{
	uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}

/* ---- one request's bookkeeping ------------------------------------- */
typedef struct {
	atomic_int		calls;			/* times the callback ran */
	pthread_t		cb_thread;
	int				errcode;
	uint32_t		nitems_seen;	/* kern_resultbuf.nitems as the callback saw it */
	int64_t			result_sum;		/* sum of results[] as the callback saw it */
	kern_resultbuf *kres;
	strom_task	   *task;
	int				wait_inside;	/* call strom_task_wait() from inside the callback */
	atomic_int		waited_inside;
	atomic_int	   *pending;		/* requests of this section still to answer */
} request;

static pthread_mutex_t	done_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t	done_cond = PTHREAD_COND_INITIALIZER;

static void
on_done(void *arg, int errcode, const strom_perfmon *pfm)
{
	request *rq = (request *)arg;

	(void)pfm;
	rq->cb_thread = pthread_self();
	rq->errcode = errcode;
	if (rq->kres)
	{
		int64_t sum = 0;
		rq->nitems_seen = rq->kres->nitems;
		if (errcode == 0)
			for (uint32_t i = 0; i < rq->kres->nitems * rq->kres->nrels; i++)
				sum += rq->kres->results[i];
		rq->result_sum = sum;
	}
	if (rq->wait_inside)
	{
		/* the handle may not have reached rq->task yet (the callback can beat
		 * the submitter's store): only release it here when it has */
		strom_task *t = rq->task;
		if (t)
		{
			int rc = strom_task_wait(t, NULL);
			CHECK(rc == errcode);
			atomic_store(&rq->waited_inside, 1);
		}
	}
	atomic_fetch_add(&rq->calls, 1);
	pthread_mutex_lock(&done_lock);
	atomic_fetch_sub(rq->pending, 1);
	pthread_cond_broadcast(&done_cond);
	pthread_mutex_unlock(&done_lock);
}

static void
wait_all(atomic_int *pending)
{
	pthread_mutex_lock(&done_lock);
	while (atomic_load(pending) > 0)
		pthread_cond_wait(&done_cond, &done_lock);
	pthread_mutex_unlock(&done_lock);
}

/* ---- helpers --------------------------------------------------------- */
static void *
xaligned(size_t len)
{
	void *p = NULL;
	CHECK(posix_memalign(&p, 256, len + 256) == 0);
	memset(p, 0, len + 256);
	return p;
}

static kern_data_store *
build_chunk(int format, int with_nulls)
{
	strom_column_input cols[3];
	memset(cols, 0, sizeof(cols));
	cols[0].type_oid = STROM_INT4OID; cols[0].attlen = 4; cols[0].attalign = 4; cols[0].attbyval = 1;
	cols[0].values = col_a; cols[0].isnull = with_nulls ? null_a : NULL;
	cols[1].type_oid = STROM_FLOAT8OID; cols[1].attlen = 8; cols[1].attalign = 8; cols[1].attbyval = 1;
	cols[1].values = col_b;
	cols[2].type_oid = STROM_INT4OID; cols[2].attlen = 4; cols[2].attalign = 4; cols[2].attbyval = 1;
	cols[2].values = col_fk;
	size_t len = strom_kds_required_length(format, 3, cols, NROWS);
	CHECK(len > 0);
	kern_data_store *kds = (kern_data_store *)xaligned(len);
	CHECK(strom_kds_build(format, 3, cols, NROWS, kds, len) == 0);
	return kds;
}

/* {kern_parambuf, kern_resultbuf} image with room for nrooms records of nrels ints */
static void *
make_request_image(const kern_parambuf *kparams, uint32_t nrels, uint32_t nrooms, kern_resultbuf **p_kres)
{
	size_t	plen = STROMALIGN(kparams->length);
	size_t	len = plen + KERN_RESULTBUF_LENGTH(nrels, nrooms);
	char   *img = (char *)xaligned(len);
	memcpy(img, kparams, kparams->length);
	kern_resultbuf *kres = (kern_resultbuf *)(img + plen);
	kres->nrels = nrels;
	kres->nrooms = nrooms;
	*p_kres = kres;
	return img;
}

/* ---- section 1: GpuScan, cold program, several threads --------------- */
typedef struct {
	strom_devprog_key	key;
	const kern_parambuf *kparams;
	const kern_data_store *kds_host[2];		/* COLUMN, ROW */
	strom_dstore	   *kds_dev;
	request			   *reqs;				/* NREQ_PER_THREAD of them */
	pthread_t			self;
	int					saw_pending;
} scan_thread_arg;

static void *
scan_submitter(void *p)
{
	scan_thread_arg *ta = (scan_thread_arg *)p;

	ta->self = pthread_self();
	for (int i = 0; i < NREQ_PER_THREAD; i++)
	{
		request *rq = &ta->reqs[i];
		int		errcode = -1;
		void   *img = make_request_image(ta->kparams, 1, NROWS, &rq->kres);
		if (strom_lookup_device_program(ta->key, 0) == STROM_DEVPROG_PENDING)
			ta->saw_pending++;
		/* alternate: resident chunk / host COLUMN chunk / host ROW chunk */
		strom_task *t;
		if (i % 3 == 0)
			t = strom_submit_gpuscan(ta->key, (kern_gpuscan *)img, NULL, ta->kds_dev, NULL, 0,
									 on_done, rq, &errcode);
		else
			t = strom_submit_gpuscan(ta->key, (kern_gpuscan *)img, ta->kds_host[i % 3 - 1], NULL, NULL, 0,
									 on_done, rq, &errcode);
		CHECK(t != NULL && errcode == 0);
		rq->task = t;
	}
	return NULL;
}

int
main(int argc, char **argv)
{
	uint64_t seed = 0x5eed0010;
	int32_t	 k = (int32_t)(0.5 * 2147483648.0);
	double	 c = 0.6;
	int64_t	 expect_sum = 0;
	uint32_t expect_n = 0;

	(void)argc; (void)argv;
	/* cold cache: nothing may come from disk, the programs must be built now */
	setenv("STROM_HIP_NO_DISK_CACHE", "1", 1);
	for (int i = 0; i < NROWS; i++)
	{
		col_a[i] = (int32_t)(splitmix(&seed) >> 33);
		col_b[i] = (double)(splitmix(&seed) >> 11) / 9007199254740992.0;
		col_fk[i] = (int32_t)(splitmix(&seed) % 2500);
		null_a[i] = (splitmix(&seed) % 20 == 0);
		if (!null_a[i] && col_a[i] < k && col_b[i] > c)
		{
			expect_n++;
			expect_sum += i + 1;
		}
	}
	CHECK(strom_init(NULL, 0) == 0);

	/* ------------------------------------------------------------- *
	 * 1. GpuScan: 3 submitter threads x 4 requests, program cold
	 * ------------------------------------------------------------- */
	{
		strom_codegen_result cg;
		char	qual[256];
		/* a constant nobody else uses makes the program text -- and its key -- new */
		snprintf(qual, sizeof(qual),
				 "(and (int4lt (var 1 int4) (param 0 int4)) (float8gt (var 2 float8) (param 1 float8))"
				 " (int4ne (var 3 int4) (const int4 -%d)))", 1000000 + (int)(getpid() % 1000000));
		CHECK(strom_codegen_gpuscan(qual, &cg) == 0);
		uint64_t ext[2];
		ext[0] = (uint64_t)(uint32_t)k;
		memcpy(&ext[1], &c, 8);
		kern_parambuf *kparams = strom_create_kern_parambuf(&cg, ext, NULL, 2);
		CHECK(kparams != NULL);
		kern_data_store *kds_col = build_chunk(KDS_FORMAT_COLUMN, 1);
		kern_data_store *kds_row = build_chunk(KDS_FORMAT_ROW, 1);
		strom_dstore *resident = strom_dstore_upload(kds_col, 0);
		CHECK(resident != NULL);

		strom_devprog_key key = strom_get_devprog_key(cg.source, cg.extra_flags);
		CHECK(strom_lookup_device_program(key, 0) == STROM_DEVPROG_PENDING);	/* the build has just begun */

		atomic_int	pending = NTHREADS * NREQ_PER_THREAD;
		request	   *reqs = (request *)calloc(NTHREADS * NREQ_PER_THREAD, sizeof(request));
		scan_thread_arg ta[NTHREADS];
		pthread_t	th[NTHREADS];
		for (int t = 0; t < NTHREADS; t++)
		{
			memset(&ta[t], 0, sizeof(ta[t]));
			ta[t].key = key;
			ta[t].kparams = kparams;
			ta[t].kds_host[0] = kds_col;
			ta[t].kds_host[1] = kds_row;
			ta[t].kds_dev = resident;
			ta[t].reqs = reqs + t * NREQ_PER_THREAD;
			for (int i = 0; i < NREQ_PER_THREAD; i++)
			{
				ta[t].reqs[i].pending = &pending;
				ta[t].reqs[i].wait_inside = ((t + i) % 2 == 0);
			}
			CHECK(pthread_create(&th[t], NULL, scan_submitter, &ta[t]) == 0);
		}
		int parked = 0;
		for (int t = 0; t < NTHREADS; t++)
		{
			pthread_join(th[t], NULL);
			parked += ta[t].saw_pending;
		}
		/* submission returned while the program was still building: the requests are parked */
		CHECK(parked > 0);
		CHECK(strom_lookup_device_program(key, 1) == STROM_DEVPROG_READY);
		wait_all(&pending);
		strom_synchronize();
		for (int t = 0; t < NTHREADS; t++)
			for (int i = 0; i < NREQ_PER_THREAD; i++)
			{
				request *rq = &ta[t].reqs[i];
				CHECK(atomic_load(&rq->calls) == 1);
				CHECK(rq->errcode == 0);
				CHECK(rq->nitems_seen == expect_n);			/* final BEFORE the callback ran */
				CHECK(rq->result_sum == expect_sum);
				CHECK(!pthread_equal(rq->cb_thread, pthread_self()));
				for (int u = 0; u < NTHREADS; u++)
					CHECK(!pthread_equal(rq->cb_thread, ta[u].self));
				if (!atomic_load(&rq->waited_inside))
					CHECK(strom_task_wait(rq->task, NULL) == 0);	/* release from another thread */
			}
		/* exactly once: give a stray second call a moment to show */
		usleep(20000);
		for (int r = 0; r < NTHREADS * NREQ_PER_THREAD; r++)
			CHECK(atomic_load(&reqs[r].calls) == 1);
		printf("ok: gpuscan %d requests from %d threads, %d submitted while the program was building, "
			   "%u rows selected each\n", NTHREADS * NREQ_PER_THREAD, NTHREADS, parked, expect_n);

		/* ---- 1b. a request that cannot start still answers through the callback ---- */
		{
			const char *broken = "#include \"strom_kds.h\"\n#include \"strom_common.h\"\nthis is not HIP;\n";
			strom_devprog_key bad = strom_get_devprog_key(broken, DEVKERNEL_NEEDS_GPUSCAN);
			atomic_int	pend = 1;
			request		rq;
			int			errcode = -1;
			memset(&rq, 0, sizeof(rq));
			rq.pending = &pend;
			void *img = make_request_image(kparams, 1, NROWS, &rq.kres);
			strom_task *t = strom_submit_gpuscan(bad, (kern_gpuscan *)img, NULL, resident, NULL, 0,
												 on_done, &rq, &errcode);
			CHECK(t != NULL && errcode == 0);
			rq.task = t;
			wait_all(&pend);
			CHECK(atomic_load(&rq.calls) == 1);
			CHECK(rq.errcode == StromError_ProgramBuildFailure);
			CHECK(!pthread_equal(rq.cb_thread, pthread_self()));
			CHECK(strom_lookup_device_program(bad, 1) == STROM_DEVPROG_BAD);
			CHECK(strlen(strom_get_devprog_errmsg(bad)) > 0);
			CHECK(strom_task_wait(t, NULL) == StromError_ProgramBuildFailure);
			/* ... and when the program is already known to be bad: same path, not the caller's stack */
			atomic_store(&pend, 1);
			memset(&rq, 0, sizeof(rq));
			rq.pending = &pend;
			img = make_request_image(kparams, 1, NROWS, &rq.kres);
			t = strom_submit_gpuscan(bad, (kern_gpuscan *)img, NULL, resident, NULL, 0, on_done, &rq, &errcode);
			CHECK(t != NULL);
			rq.task = t;
			wait_all(&pend);
			CHECK(atomic_load(&rq.calls) == 1 && rq.errcode == StromError_ProgramBuildFailure);
			CHECK(!pthread_equal(rq.cb_thread, pthread_self()));
			strom_task_release(t);
			/* a request the library refuses outright returns NULL and never calls back */
			memset(&rq, 0, sizeof(rq));
			rq.pending = &pend;
			img = make_request_image(kparams, 1, 10 /* too few rooms */, &rq.kres);
			t = strom_submit_gpuscan(key, (kern_gpuscan *)img, NULL, resident, NULL, 0, on_done, &rq, &errcode);
			CHECK(t == NULL && errcode == StromError_BadRequestMessage);
			usleep(20000);
			CHECK(atomic_load(&rq.calls) == 0);
			printf("ok: build failure and refused requests answer as specified\n");
		}
		strom_dstore_release(resident);
		strom_put_devprog_key(key);
	}

	/* ------------------------------------------------------------- *
	 * 2. GpuHashJoin: DataStoreNoSpace -> retry with the room asked for
	 * ------------------------------------------------------------- */
	{
		enum { NDIM = 2000 };
		static int32_t dkey[NDIM], dval[NDIM];
		for (int i = 0; i < NDIM; i++)
		{
			dkey[i] = (i * 7) % NDIM;		/* a permutation of 0..1999 (7 and 2000 coprime) */
			dval[i] = dkey[i] % 13;
		}
		uint32_t expect_matches = 0;
		for (int i = 0; i < NROWS; i++)
			expect_matches += (col_fk[i] < NDIM);
		strom_column_input dcols[2];
		memset(dcols, 0, sizeof(dcols));
		for (int j = 0; j < 2; j++)
		{
			dcols[j].type_oid = STROM_INT4OID; dcols[j].attlen = 4; dcols[j].attalign = 4; dcols[j].attbyval = 1;
		}
		dcols[0].values = dkey;
		dcols[1].values = dval;
		size_t dlen = strom_kds_required_length(KDS_FORMAT_ROW_FLAT, 2, dcols, NDIM);
		kern_data_store *dim = (kern_data_store *)xaligned(dlen);
		CHECK(strom_kds_build(KDS_FORMAT_ROW_FLAT, 2, dcols, NDIM, dim, dlen) == 0);
		strom_hashtable_input hin;
		memset(&hin, 0, sizeof(hin));
		hin.inner = dim; hin.nkeys = 1; hin.key_attnos[0] = 1;
		size_t mlen = strom_multihash_required_length(1, &hin);
		kern_multihash *km = (kern_multihash *)xaligned(mlen);
		CHECK(strom_multihash_build(1, &hin, km, mlen) == 0);

		strom_codegen_result cg;
		int		nrels = 0;
		char	spec[256];
		snprintf(spec, sizeof(spec),
				 "(gpuhashjoin (rel (hashkey (var 3 int4) 1 int4) (qual (int4ne (var 3 int4) (const int4 -%d)))))",
				 2000000 + (int)(getpid() % 1000000));
		CHECK(strom_codegen_gpuhashjoin(spec, &cg, &nrels) == 0 && nrels == 1);
		kern_parambuf *kparams = strom_create_kern_parambuf(&cg, NULL, NULL, 0);
		strom_devprog_key key = strom_get_devprog_key(cg.source, cg.extra_flags);
		int		errcode = -1;
		strom_hashjoin_table *tbl = strom_hashjoin_table_create(key, km, mlen, 0, &errcode);
		CHECK(tbl != NULL && errcode == 0);
		kern_data_store *kds_col = build_chunk(KDS_FORMAT_COLUMN, 0);

		atomic_int	pend = 1;
		request		rq;
		memset(&rq, 0, sizeof(rq));
		rq.pending = &pend;
		void *img = make_request_image(kparams, 2, 1000 /* far too small */, &rq.kres);
		strom_task *t = strom_submit_gpuhashjoin(tbl, (kern_hashjoin *)img, kds_col, NULL, NULL, 0,
												 on_done, &rq, &errcode);
		CHECK(t != NULL);
		rq.task = t;
		wait_all(&pend);
		CHECK(atomic_load(&rq.calls) == 1);
		CHECK(rq.errcode == StromError_DataStoreNoSpace);
		CHECK(rq.nitems_seen == expect_matches);		/* the room a retry needs */
		CHECK(!pthread_equal(rq.cb_thread, pthread_self()));
		strom_task_release(t);
		/* retry, exactly sized */
		atomic_store(&pend, 1);
		uint32_t need = rq.nitems_seen;
		memset(&rq, 0, sizeof(rq));
		rq.pending = &pend;
		img = make_request_image(kparams, 2, need, &rq.kres);
		t = strom_submit_gpuhashjoin(tbl, (kern_hashjoin *)img, kds_col, NULL, NULL, 0, on_done, &rq, &errcode);
		CHECK(t != NULL);
		rq.task = t;
		wait_all(&pend);
		CHECK(atomic_load(&rq.calls) == 1 && rq.errcode == 0 && rq.nitems_seen == expect_matches);
		/* every pair: outer row id + 1, then an entry offset whose tuple carries the key */
		{
			int64_t osum = 0, esum = 0;
			for (uint32_t i = 0; i < rq.kres->nitems; i++)
			{
				int32_t orow = rq.kres->results[2 * i] - 1;
				CHECK(orow >= 0 && orow < NROWS && col_fk[orow] < NDIM);
				osum += orow;
			}
			for (int i = 0; i < NROWS; i++)
				if (col_fk[i] < NDIM)
					esum += i;
			CHECK(osum == esum);
		}
		CHECK(strom_task_wait(t, NULL) == 0);
		printf("ok: gpuhashjoin DataStoreNoSpace asked for %u records, the retry returned them\n", expect_matches);
		strom_hashjoin_table_release(tbl);
		strom_put_devprog_key(key);
	}

	/* ------------------------------------------------------------- *
	 * 3. GpuPreAgg: folds with callbacks from 2 threads' worth of chunks
	 * ------------------------------------------------------------- */
	{
		strom_codegen_result cg;
		strom_preagg_target targets[8];
		int		ntargets = 0;
		char	spec[256];
		snprintf(spec, sizeof(spec),
				 "(gpupreagg (qual (int4ne (var 3 int4) (const int4 -%d))) (key (var 3 int4)) (nrows)"
				 " (psum (int8 (var 1 int4))) (psum (var 2 float8)))", 3000000 + (int)(getpid() % 1000000));
		CHECK(strom_codegen_gpupreagg(spec, &cg, targets, 8, &ntargets) == 0 && ntargets == 4);
		kern_parambuf *kparams = strom_create_kern_parambuf(&cg, NULL, NULL, 0);
		strom_devprog_key key = strom_get_devprog_key(cg.source, cg.extra_flags);
		strom_preagg_domain dom;
		memset(&dom, 0, sizeof(dom));
		dom.nkeys = 1; dom.key_min[0] = 0; dom.key_range[0] = 2500;
		int		errcode = -1;
		strom_gpupreagg *sess = strom_gpupreagg_create(key, targets, ntargets, kparams, &dom, 0, &errcode);
		CHECK(sess != NULL && errcode == 0);
		kern_data_store *kds_col = build_chunk(KDS_FORMAT_COLUMN, 0);
		enum { NFOLD = 5 };
		atomic_int	pend = NFOLD;
		request		rq[NFOLD];
		memset(rq, 0, sizeof(rq));
		for (int i = 0; i < NFOLD; i++)
		{
			rq[i].pending = &pend;
			strom_task *t = strom_submit_gpupreagg(sess, kds_col, NULL, NULL, on_done, &rq[i], &errcode);
			CHECK(t != NULL);
			rq[i].task = t;
		}
		wait_all(&pend);
		for (int i = 0; i < NFOLD; i++)
		{
			CHECK(atomic_load(&rq[i].calls) == 1 && rq[i].errcode == 0);
			CHECK(!pthread_equal(rq[i].cb_thread, pthread_self()));
			strom_task_release(rq[i].task);
		}
		long need = strom_gpupreagg_fetch(sess, NULL, 0);
		CHECK(need > 0);
		kern_data_store *dest = (kern_data_store *)xaligned((size_t)need);
		long ngroups = strom_gpupreagg_fetch(sess, dest, (size_t)need);
		static int64_t cnt[2500], sx[2500];
		memset(cnt, 0, sizeof(cnt)); memset(sx, 0, sizeof(sx));
		long expect_groups = 0;
		for (int i = 0; i < NROWS; i++)
		{
			if (cnt[col_fk[i]]++ == 0)
				expect_groups++;
			sx[col_fk[i]] += col_a[i];
		}
		CHECK(ngroups == expect_groups);
		for (long g = 0; g < ngroups; g++)
		{
			uint64_t v[4];
			for (int cidx = 0; cidx < 4; cidx++)
				CHECK(strom_kds_fetch(dest, (uint32_t)g, (uint32_t)cidx, &v[cidx]) == 0);
			int32_t gk = (int32_t)v[0];
			CHECK(gk >= 0 && gk < 2500);
			CHECK((int64_t)v[1] == NFOLD * cnt[gk]);			/* exact: integers */
			CHECK((int64_t)v[2] == NFOLD * sx[gk]);
		}
		printf("ok: gpupreagg %d folds with callbacks, %ld groups, counts and integer sums exact\n", NFOLD, ngroups);
		strom_gpupreagg_release(sess);

		/* --------------------------------------------------------- *
		 * 4. the reference's own per-chunk message (pgstrom_gpupreagg,
		 *    opencl_gpupreagg.h:994-1003): kern_gpupreagg image + pds +
		 *    pds_dest, no session, no domain; three in flight, callbacks
		 * --------------------------------------------------------- */
		enum { NMSG = 3 };
		size_t	map_off = STROMALIGN(offsetof(kern_gpupreagg, kparams) + kparams->length);
		size_t	kg_len = map_off + sizeof(kern_row_map);
		size_t	dest_len = KDS_HEAD_LENGTH(ntargets) + KDS_TUPSLOT_STRIDE(ntargets) * (size_t)2500;
		kern_gpupreagg *kg[NMSG];
		kern_data_store *pds_dest[NMSG];
		request		mq[NMSG];
		atomic_int	mpend = NMSG;
		memset(mq, 0, sizeof(mq));
		for (int i = 0; i < NMSG; i++)
		{
			kg[i] = (kern_gpupreagg *)xaligned(kg_len);
			memset(kg[i], 0, kg_len);
			memcpy(&kg[i]->kparams, kparams, kparams->length);
			KERN_GPUPREAGG_KROWMAP(kg[i])->nvalids = -1;		/* every row */
			kg[i]->status = 12345;								/* must be overwritten */
			pds_dest[i] = (kern_data_store *)xaligned(dest_len);
			memset(pds_dest[i], 0, KDS_HEAD_LENGTH(ntargets));
			mq[i].pending = &mpend;
			strom_task *t = strom_submit_gpupreagg_chunk(key, targets, ntargets, kg[i], kds_col, NULL,
														 pds_dest[i], dest_len, 1, 2500.0, 0,
														 on_done, &mq[i], &errcode);
			CHECK(t != NULL && errcode == 0);
			mq[i].task = t;
		}
		wait_all(&mpend);
		for (int i = 0; i < NMSG; i++)
		{
			CHECK(atomic_load(&mq[i].calls) == 1 && mq[i].errcode == 0 && kg[i]->status == 0);
			CHECK(!pthread_equal(mq[i].cb_thread, pthread_self()));
			CHECK(strom_task_wait(mq[i].task, NULL) == 0);
			CHECK((long)pds_dest[i]->nitems == expect_groups && pds_dest[i]->format == KDS_FORMAT_TUPSLOT);
			for (long g = 0; g < expect_groups; g++)
			{
				uint64_t v[4];
				for (int cidx = 0; cidx < 4; cidx++)
					CHECK(strom_kds_fetch(pds_dest[i], (uint32_t)g, (uint32_t)cidx, &v[cidx]) == 0);
				int32_t gk = (int32_t)v[0];
				CHECK(gk >= 0 && gk < 2500);
				CHECK((int64_t)v[1] == cnt[gk] && (int64_t)v[2] == sx[gk]);
			}
		}
		/* a store that is too small for the chunk's groups: DataStoreNoSpace, status says so */
		{
			request		one;
			atomic_int	p1 = 1;
			memset(&one, 0, sizeof(one));
			one.pending = &p1;
			kg[0]->status = 12345;
			strom_task *t = strom_submit_gpupreagg_chunk(key, targets, ntargets, kg[0], kds_col, NULL,
														 pds_dest[0], KDS_HEAD_LENGTH(ntargets) + 64, 1, 0.0, 0,
														 on_done, &one, &errcode);
			CHECK(t != NULL);
			wait_all(&p1);
			CHECK(one.errcode == StromError_DataStoreNoSpace && kg[0]->status == StromError_DataStoreNoSpace);
			CHECK(strom_task_wait(t, NULL) == StromError_DataStoreNoSpace);
			/* needs_grouping that contradicts the program is refused at once, no callback */
			CHECK(strom_submit_gpupreagg_chunk(key, targets, ntargets, kg[0], kds_col, NULL, pds_dest[0], dest_len,
											   0, 0.0, 0, on_done, &one, &errcode) == NULL &&
				  errcode == StromError_BadRequestMessage);
		}
		printf("ok: %d per-chunk gpupreagg messages in flight, %ld groups each, status word and pds_dest filled "
			   "before the callback\n", NMSG, expect_groups);
		strom_put_devprog_key(key);
	}
	strom_shutdown();
	printf("ALL OK\n");
	return 0;
}
