/*
 * strom_gpuscan.h -- GpuScan kernels (HIP, gfx950)
 *
 * Role in the reference: opencl_gpuscan.h:98-177 (gpuscan_qual +
 * gpuscan_writeback_row_error).  Per row i of the chunk:
 *     errcode = Success; rc = gpuscan_qual_eval(&errcode, ...);
 *     STROM_SET_ERROR(&errcode, rc is TRUE ? Success : RowFiltered);
 *     Success    -> results[] gets  +(i+1)
 *     CpuReCheck -> results[] gets  -(i+1)   (host re-evaluates the row)
 *     significant-> chunk errcode, first one wins
 * and kresults->nitems counts the entries written.  The reference leaves
 * the order of results[] undefined across work-groups; so does this.
 *
 * What is different is how a chunk is walked.  The generated code supplies
 *     STROM_KVAR_LIST(X)    X(attno, colidx, NAME)  one per referenced Var
 *     STROM_KPARAM_LIST(X)  X(index, NAME)          one per Const/Param
 *     gpuscan_qual_eval(errcode, KP, KV)
 * and this file supplies two kernels around it:
 *
 *   gpuscan_qual_column  KDS_FORMAT_COLUMN, no row map.  Persistent blocks
 *       walk tiles of BLOCK*4*QUADS rows; each thread fetches its quads with
 *       16-byte vector loads straight from the column arrays (no LDS on the
 *       read side: every byte is used exactly once), evaluates 4*QUADS rows
 *       in registers, and the survivors are compacted in row order with
 *       ballot+mbcnt (no LDS scan) into an LDS stage of STAGE entries.  The
 *       stage is flushed with one global atomic per flush -- a few thousand
 *       per 1e8-row chunk instead of one per work-group per 256 rows --
 *       and the flush itself is a contiguous coalesced store.
 *
 *   gpuscan_qual_generic  any format, optional kern_row_map.  Same
 *       compaction, rows fetched one datum at a time through kern_get_datum
 *       (this is the heap-tuple walk the reference does for every row).
 */
#ifndef STROM_GPUSCAN_DEVICE_H
#define STROM_GPUSCAN_DEVICE_H

#ifndef GPUSCAN_BLOCK
#define GPUSCAN_BLOCK		256
#endif
#ifndef GPUSCAN_QUADS
#define GPUSCAN_QUADS		1			/* 4 rows per thread and tile (swept: profiles/r01_gpuscan_tune.txt) */
#endif
#ifndef GPUSCAN_STAGE
#define GPUSCAN_STAGE		8192		/* LDS entries (32 KB) */
#endif
#define GPUSCAN_NWAVES		(GPUSCAN_BLOCK / STROM_WAVE)
#define GPUSCAN_TILE_ROWS	(GPUSCAN_BLOCK * 4 * GPUSCAN_QUADS)

/* ---- what the generated code declares -------------------------------- */
struct strom_kparams {
#define X(idx,NAME)	pg_##NAME##_t KPARAM_##idx;
	STROM_KPARAM_LIST(X)
#undef X
	int __dummy;
};
struct strom_kvars {
#define X(attno,colidx,NAME)	pg_##NAME##_t KVAR_##attno;
	STROM_KVAR_LIST(X)
#undef X
	int __dummy;
};
/* text / character(n) variables of a row taken from COLUMN arrays: offset -> address (strom_common.h) */
#ifndef STROM_KVARLENA_LIST
#define STROM_KVARLENA_LIST(X)
#endif
STROM_DEFINE_KVARS_FROM_COLUMN

STROM_DEVICE pg_bool_t
gpuscan_qual_eval(cl_int *errcode,
				  const strom_kparams &KP,
				  const strom_kvars &KV);

STROM_DEVICE void
gpuscan_load_kparams(strom_kparams &KP, const kern_parambuf *kparams, cl_int *errcode)
{
#define X(idx,NAME)	KP.KPARAM_##idx = pg_##NAME##_param(kparams, errcode, idx);
	STROM_KPARAM_LIST(X)
#undef X
	KP.__dummy = 0;
}

/* row status -> +1 pass, -1 recheck, 0 drop; significant errors accumulate */
STROM_DEVICE int
gpuscan_row_status(pg_bool_t rc, cl_int errcode, cl_int *chunk_error)
{
	STROM_SET_ERROR(&errcode, (!rc.isnull && rc.value != 0)
					? StromError_Success : StromError_RowFiltered);
	if (errcode == StromError_Success)
		return 1;
	if (errcode == StromError_CpuReCheck)
		return -1;
	if (StromErrorIsSignificant(errcode))
		STROM_SET_ERROR(chunk_error, errcode);
	return 0;
}

/* ---- LDS stage shared by both kernels ---------------------------------- */
template <int NENTRIES>
struct gpuscan_stage_t {
	cl_int		entries[NENTRIES];
	cl_uint		wave_total[GPUSCAN_QUADS][GPUSCAN_NWAVES];
	cl_uint		flush_base;
};
typedef gpuscan_stage_t<GPUSCAN_STAGE> gpuscan_stage;
/* the row-at-a-time kernel's stage: measured, a shallow stage (2048 entries,
 * twice the resident waves) is SLOWER than the deep one (630 vs 390 us per
 * 1e7 rows) -- more flushes mean more reservations on the one result cursor */
#ifndef GPUSCAN_GENERIC_STAGE
#define GPUSCAN_GENERIC_STAGE	GPUSCAN_STAGE
#endif
typedef gpuscan_stage_t<GPUSCAN_GENERIC_STAGE> gpuscan_generic_stage;

template <typename STAGE>
STROM_DEVICE void
gpuscan_stage_flush(STAGE &stage, kern_resultbuf *kresults, cl_uint fill)
{
	/* caller guarantees a barrier since the last write into the stage */
	if (fill == 0)
		return;
#if (defined(GPUSCAN_ABLATE) && GPUSCAN_ABLATE != 0) && !defined(STROM_DIAGNOSTIC_BUILD)
#error "GPUSCAN_ABLATE builds leave work out and give wrong results: measurement only (set STROM_DIAGNOSTIC_BUILD=1, as scripts/gpu_*_ablate* do)"
#endif
#if defined(GPUSCAN_ABLATE) && GPUSCAN_ABLATE == 2
	/* diagnostic build (wrong results): no reservation atomic */
	if (threadIdx.x == 0)
		stage.flush_base = (blockIdx.x * 7919u) % (kresults->nrooms - GPUSCAN_STAGE);
#else
	if (threadIdx.x == 0)
		stage.flush_base = atomicAdd(&kresults->nitems, fill);
#endif
	__syncthreads();
	cl_uint		base = stage.flush_base;
	cl_int	   *dest = kresults->results + base;
#if !defined(GPUSCAN_ABLATE) || GPUSCAN_ABLATE != 1
#if !defined(GPUSCAN_STORE_NT) || GPUSCAN_STORE_NT
	/* results are written once and read elsewhere: non-temporal stores keep
	 * them out of the L2 write-allocate path of the read stream (+5 % at 10 %
	 * selectivity, +8 % at 49 %: profiles/r01_gpuscan_tune_v2.txt) */
	for (cl_uint i = threadIdx.x; i < fill; i += GPUSCAN_BLOCK)
		__builtin_nontemporal_store(stage.entries[i], &dest[i]);
#else
	for (cl_uint i = threadIdx.x; i < fill; i += GPUSCAN_BLOCK)
		dest[i] = stage.entries[i];
#endif
#else
	/* diagnostic build (wrong results): no result stores */
	if (fill == 0xffffffffu)
		dest[0] = stage.entries[0];
#endif
	__syncthreads();
}

/*
 * Compaction of one tile.  st[k][j] is the status of row
 * (tile_base + (k*BLOCK + tid)*4 + j).  Entries land in the stage in row
 * order: (k, wave, lane, j).  Returns the number of entries appended; the
 * value is identical in every thread.
 */
/* LANE_ROWS: st[k][j] belongs to row tile_base + (k*4 + j)*BLOCK + tid instead -- consecutive
 * lanes, consecutive rows (the row-at-a-time kernel: neighbouring lanes then walk neighbouring
 * heap tuples) */
template <bool LANE_ROWS = false, typename STAGE>
STROM_DEVICE cl_uint
gpuscan_stage_append(STAGE &stage, cl_uint fill,
					 cl_uint tile_base, const int (&st)[GPUSCAN_QUADS][4])
{
	cl_uint		lane = threadIdx.x & (STROM_WAVE - 1);
	cl_uint		wave = threadIdx.x / STROM_WAVE;
	cl_uint		my_prefix[GPUSCAN_QUADS];

#pragma unroll
	for (int k = 0; k < GPUSCAN_QUADS; k++)
	{
		cl_uint	prefix = 0, total = 0;
#pragma unroll
		for (int j = 0; j < 4; j++)
		{
			strom_lanemask_t m = __ballot(st[k][j] != 0);
			prefix += strom_mbcnt(m);
			total += __popcll(m);
		}
		my_prefix[k] = prefix;
		if (lane == 0)
			stage.wave_total[k][wave] = total;
	}
	__syncthreads();

	cl_uint		appended = 0;
#pragma unroll
	for (int k = 0; k < GPUSCAN_QUADS; k++)
	{
		cl_uint	before = 0, all = 0;
#pragma unroll
		for (int w = 0; w < GPUSCAN_NWAVES; w++)
		{
			cl_uint	t = stage.wave_total[k][w];
			before += (w < (int)wave ? t : 0);
			all += t;
		}
		cl_uint	pos = fill + appended + before + my_prefix[k];
		cl_uint	row0 = tile_base + (k * GPUSCAN_BLOCK + threadIdx.x) * 4;
#pragma unroll
		for (int j = 0; j < 4; j++)
		{
			if (st[k][j] != 0)
			{
				cl_int	rowid = (LANE_ROWS
								 ? (cl_int)(tile_base + (k * 4 + j) * GPUSCAN_BLOCK + threadIdx.x + 1)
								 : (cl_int)(row0 + j + 1));
				stage.entries[pos++] = (st[k][j] > 0 ? rowid : -rowid);
			}
		}
		appended += all;
	}
	/* the next append rewrites wave_total[]; entries[] are read by flush */
	__syncthreads();
	return appended;
}

/* ====================================================================== *
 * COLUMN format, streaming
 * ====================================================================== */
struct gpuscan_column_tile {
#define X(attno,colidx,NAME)											\
	pg_##NAME##_base_t	v_##attno[GPUSCAN_QUADS][4];						\
	cl_uint				nn_##attno[GPUSCAN_QUADS];
	STROM_KVAR_LIST(X)
#undef X
	int __dummy;
};

extern "C" __global__ void
__launch_bounds__(GPUSCAN_BLOCK)
gpuscan_qual_column(kern_gpuscan *kgpuscan, const kern_data_store *kds)
{
	__shared__ gpuscan_stage stage;
	const kern_parambuf *kparams = KERN_GPUSCAN_PARAMBUF(kgpuscan);
	kern_resultbuf *kresults = KERN_GPUSCAN_RESULTBUF(kgpuscan);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
	cl_uint		nitems = kds->nitems;
	cl_uint		ntiles = (nitems + GPUSCAN_TILE_ROWS - 1) / GPUSCAN_TILE_ROWS;
	cl_int		chunk_error = StromError_Success;
	cl_int		param_error = StromError_Success;
	cl_uint		fill = 0;
	strom_kparams KP;

	gpuscan_load_kparams(KP, kparams, &param_error);

	/* column base pointers, hoisted */
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (const char *)kds + coldir[colidx].values_off;	\
	const cl_uint *nul_##attno = (coldir[colidx].nulls_off != 0					\
		? (const cl_uint *)((const char *)kds + coldir[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	bool		any_nulls = false;		/* wave-uniform: picks the bitmap-free loader */
#define X(attno,colidx,NAME)	any_nulls = any_nulls || (nul_##attno != NULL);
	STROM_KVAR_LIST(X)
#undef X

	for (cl_uint tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
	{
		cl_uint		tile_base = tile * GPUSCAN_TILE_ROWS;
		bool		full_tile = (tile_base + GPUSCAN_TILE_ROWS <= nitems);
		gpuscan_column_tile T;
		int			st[GPUSCAN_QUADS][4];

		/* issue every load of the tile before the first use */
		if (full_tile && !any_nulls)
		{
#pragma unroll
			for (int k = 0; k < GPUSCAN_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUSCAN_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, true, true>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
		else if (full_tile)
		{
#pragma unroll
			for (int k = 0; k < GPUSCAN_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUSCAN_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, true>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
		else
		{
#pragma unroll
			for (int k = 0; k < GPUSCAN_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUSCAN_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, false>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
		if (fill + GPUSCAN_TILE_ROWS > GPUSCAN_STAGE)
		{
			gpuscan_stage_flush(stage, kresults, fill);
			fill = 0;
		}
#pragma unroll
		for (int k = 0; k < GPUSCAN_QUADS; k++)
		{
			cl_uint	row0 = tile_base + (k * GPUSCAN_BLOCK + threadIdx.x) * 4;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				strom_kvars	KV;
				cl_int		errcode = param_error;
#define X(attno,colidx,NAME)													\
				KV.KVAR_##attno = pg_##NAME##_make(T.v_##attno[k][j],				\
												   !((T.nn_##attno[k] >> j) & 1));
				STROM_KVAR_LIST(X)
#undef X
				KV.__dummy = 0;
				strom_kvars_from_column(KV, kds, &errcode);
				pg_bool_t rc = gpuscan_qual_eval(&errcode, KP, KV);
				st[k][j] = (row0 + j < nitems
							? gpuscan_row_status(rc, errcode, &chunk_error) : 0);
			}
		}
		fill += gpuscan_stage_append(stage, fill, tile_base, st);
	}
	gpuscan_stage_flush(stage, kresults, fill);
	kern_writeback_error_status(&kresults->errcode, chunk_error);
}

/* ====================================================================== *
 * any format, optional row map: one datum at a time
 * ====================================================================== */
template <bool IS_COLUMN>
__device__ __forceinline__ void
gpuscan_qual_generic_body(kern_gpuscan *kgpuscan,
					 const kern_data_store *kds,
					 const kern_data_store *ktoast,
					 const kern_row_map *krowmap,
					 gpuscan_generic_stage &stage)
{
	const kern_parambuf *kparams = KERN_GPUSCAN_PARAMBUF(kgpuscan);
	kern_resultbuf *kresults = KERN_GPUSCAN_RESULTBUF(kgpuscan);
	bool		use_map = (krowmap != NULL && krowmap->nvalids >= 0);
	cl_uint		nrows = (use_map ? (cl_uint)krowmap->nvalids : kds->nitems);
	cl_uint		ntiles = (nrows + GPUSCAN_TILE_ROWS - 1) / GPUSCAN_TILE_ROWS;
	cl_int		chunk_error = StromError_Success;
	cl_int		param_error = StromError_Success;
	cl_uint		fill = 0;
	strom_kparams KP;

	gpuscan_load_kparams(KP, kparams, &param_error);

	/* COLUMN chunk behind a row map: column pointers hoisted, no chunk
	 * header field is read per row */
	const bool	is_column = IS_COLUMN;		/* fixed per launch: the other accessor is not compiled in */
	const cl_int chunk_format = kds->format;
	const bool	row_family = (chunk_format == KDS_FORMAT_ROW || chunk_format == KDS_FORMAT_ROW_FLAT);
	const kern_coldir *coldir_g = KERN_DATA_STORE_COLDIR(kds);
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (is_column ? (const char *)kds + coldir_g[colidx].values_off : NULL);	\
	const cl_uint *nul_##attno = ((is_column && coldir_g[colidx].nulls_off != 0)	\
		? (const cl_uint *)((const char *)kds + coldir_g[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	for (cl_uint tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
	{
		cl_uint		tile_base = tile * GPUSCAN_TILE_ROWS;
		int			st[GPUSCAN_QUADS][4];

		if (fill + GPUSCAN_TILE_ROWS > GPUSCAN_GENERIC_STAGE)
		{
			gpuscan_stage_flush(stage, kresults, fill);
			fill = 0;
		}
#pragma unroll
		for (int k = 0; k < GPUSCAN_QUADS; k++)
		{
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				/*
				 * consecutive lanes, consecutive rows: with a quad of rows per lane (the
				 * streaming kernel's layout) a wave's 64 lanes reach for heap tuples 4 rows
				 * apart -- 80 cache lines per load of a 40-byte-tuple table instead of 20
				 */
				cl_uint		r = tile_base + (k * 4 + j) * GPUSCAN_BLOCK + threadIdx.x;

				st[k][j] = 0;
				if (r < nrows)
				{
					cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : r);
					strom_kvars	KV;
					cl_int		errcode = param_error;
					const HeapTupleHeaderData *htup = NULL;
					if (!is_column && row_family)
						htup = strom_locate_tuple(kds, chunk_format, kds_index);
#define X(attno,colidx,NAME)													\
					KV.KVAR_##attno = (is_column											\
						? STROM_COLUMN_REF(NAME, col_##attno, nul_##attno, kds_index)		\
						: row_family ? STROM_TUPLE_REF(NAME, kds, htup, colidx)				\
						: pg_##NAME##_vref(kds, ktoast, &errcode, colidx, kds_index));
					STROM_KVAR_LIST(X)
#undef X
					KV.__dummy = 0;
					if (is_column)
						strom_kvars_from_column(KV, kds, &errcode);
					pg_bool_t rc = gpuscan_qual_eval(&errcode, KP, KV);
					st[k][j] = gpuscan_row_status(rc, errcode, &chunk_error);
				}
			}
		}
		/*
		 * with a row map the reported id is the position in the map's
		 * input order translated back to the chunk's row: patch below
		 */
		if (!use_map)
			fill += gpuscan_stage_append<true>(stage, fill, tile_base, st);
		else
		{
			cl_uint	before = fill;
			cl_uint	n = gpuscan_stage_append<true>(stage, fill, tile_base, st);
			/* translate map positions -> chunk rows, in place */
			for (cl_uint i = before + threadIdx.x; i < before + n; i += GPUSCAN_BLOCK)
			{
				cl_int	e = stage.entries[i];
				cl_int	pos = (e > 0 ? e : -e) - 1;
				cl_int	rowid = krowmap->rindex[pos] + 1;
				stage.entries[i] = (e > 0 ? rowid : -rowid);
			}
			__syncthreads();
			fill += n;
		}
	}
	gpuscan_stage_flush(stage, kresults, fill);
	kern_writeback_error_status(&kresults->errcode, chunk_error);
}

extern "C" __global__ void
__launch_bounds__(GPUSCAN_BLOCK)
gpuscan_qual_generic(kern_gpuscan *kgpuscan,
					 const kern_data_store *kds,
					 const kern_data_store *ktoast,
					 const kern_row_map *krowmap)
{
	__shared__ gpuscan_generic_stage stage;

	/* the chunk format is decided once per launch, not once per datum */
	if (kds->format == KDS_FORMAT_COLUMN)
		gpuscan_qual_generic_body<true>(kgpuscan, kds, ktoast, krowmap, stage);
	else
		gpuscan_qual_generic_body<false>(kgpuscan, kds, ktoast, krowmap, stage);
}

#endif	/* STROM_GPUSCAN_DEVICE_H */
