/*
 * strom_kds.h -- chunk / wire formats of the GpuScan / GpuHashJoin / GpuPreAgg path
 *
 * One header, three consumers: host C++ (the HIP runtime), plain C (the
 * CPU oracle under oracle/) and HIP device code (prepended to every program
 * handed to hiprtc).  Every struct that also exists in the reference keeps
 * the reference's field order, widths and alignment so that a chunk built
 * by the reference's datastore.c can be handed to this library unchanged:
 *
 *   kern_colmeta / kern_rowitem / kern_blkitem / kern_data_store
 *                           <-> opencl_common.h:335-389 (accessors 392-434)
 *   kern_parambuf           <-> opencl_common.h:443-457
 *   kern_resultbuf          <-> opencl_common.h:466-475
 *   kern_row_map            <-> opencl_common.h:483-486
 *   kern_gpuscan            <-> opencl_gpuscan.h:63-90
 *   kern_hashentry/_hashtable/_multihash/kern_hashjoin
 *                           <-> opencl_hashjoin.h:102-165, 222-252
 *   kern_gpupreagg          <-> opencl_gpupreagg.h:67-106
 *   StromError_* codes      <-> opencl_common.h:108-123
 *
 * NEW in this build (not in the reference): KDS_FORMAT_COLUMN, a
 * column-major chunk layout that the gfx950 kernels stream at HBM rate.
 * ROW / ROW_FLAT / TUPSLOT stay bit-compatible because they are what the
 * reference's host side produces and consumes.
 */
#ifndef STROM_KDS_H
#define STROM_KDS_H

#if defined(__HIPCC_RTC__)
typedef signed char			int8_t;
typedef unsigned char		uint8_t;
typedef short				int16_t;
typedef unsigned short		uint16_t;
typedef int					int32_t;
typedef unsigned int		uint32_t;
typedef long				int64_t;
typedef unsigned long		uint64_t;
typedef unsigned long		uintptr_t;
typedef unsigned long		size_t;
#ifndef offsetof
#define offsetof(T,F)		__builtin_offsetof(T,F)
#endif
#ifndef NULL
#define NULL				0
#endif
#else
#include <stdint.h>
#include <stddef.h>
#endif

#ifdef __cplusplus
#define STROM_INLINE	static inline
#else
#define STROM_INLINE	static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
#define STROM_HD	__host__ __device__
#else
#define STROM_HD
#endif

/* the cl_* spellings are part of the contract's vocabulary */
typedef int8_t		cl_char;
typedef uint8_t		cl_uchar;
typedef int16_t		cl_short;
typedef uint16_t	cl_ushort;
typedef int32_t		cl_int;
typedef uint32_t	cl_uint;
typedef int64_t		cl_long;
typedef uint64_t	cl_ulong;
typedef float		cl_float;
typedef double		cl_double;
typedef cl_char		cl_bool;
typedef uint64_t	hostptr_t;		/* HOSTPTRLEN == 8 */
typedef uint64_t	Datum;

#ifndef BLCKSZ
#define BLCKSZ		8192
#endif
#define STROM_MAXIMUM_ALIGNOF	8

#define STROM_TYPEALIGN(ALIGNVAL,LEN)	\
	(((uintptr_t)(LEN) + ((ALIGNVAL) - 1)) & ~((uintptr_t)((ALIGNVAL) - 1)))
#define STROM_TYPEALIGN_DOWN(ALIGNVAL,LEN)	\
	(((uintptr_t)(LEN)) & ~((uintptr_t)((ALIGNVAL) - 1)))
#define STROMALIGN_LEN		16
#define STROMALIGN(LEN)		STROM_TYPEALIGN(STROMALIGN_LEN,(LEN))
#define STROM_LONGALIGN(LEN)	STROM_TYPEALIGN(8,(LEN))
#define STROM_INTALIGN(LEN)	STROM_TYPEALIGN(4,(LEN))

/* ----------------------------------------------------------------
 * error codes (opencl_common.h:108-123)
 * ---------------------------------------------------------------- */
#define StromError_Success				0
#define StromError_RowFiltered			1
#define StromError_CpuReCheck			2
/* internal to GpuPreAgg (never returned to a caller): the device could not prove that the
 * integer sums of a chunk stay inside int8, nothing was merged, the runtime folds the chunk
 * again with checked additions -- the value of StromError_RowFiltered, which GpuPreAgg has no
 * use for: below CpuReCheck, above Success ("worst wins" is a max) */
#define StromError_SumRangeUnproven		1
#define StromError_ServerNotReady		100
#define StromError_BadRequestMessage	101
#define StromError_OpenCLInternal		102		/* kept for value compat */
#define StromError_HipInternal			102
#define StromError_OutOfSharedMemory	105
#define StromError_OutOfMemory			106
#define StromError_DataStoreCorruption	300
#define StromError_DataStoreNoSpace		301
#define StromError_DataStoreOutOfRange	302
#define StromError_DataStoreReCheck		303
#define StromError_SanityCheckViolation	999
/* build failure keeps the numeric value the reference's backend tests
 * for (CL_BUILD_PROGRAM_FAILURE, gpuscan.c:1140) */
#define StromError_ProgramBuildFailure	(-11)

#define StromErrorIsSignificant(errcode)	((errcode) >= 100 || (errcode) < 0)

/* ----------------------------------------------------------------
 * kern_data_store
 * ---------------------------------------------------------------- */
typedef struct {
	cl_char			attbyval;		/* pass-by-value? */
	cl_char			attalign;		/* 1,2,4 or 8 (bytes, not pg_attribute chars) */
	cl_short		attlen;			/* >0 fixed width, -1 varlena */
	cl_short		attnum;
	cl_short		attcacheoff;	/* fixed offset inside a null-free tuple, or -1 */
} kern_colmeta;

typedef union {
	struct {
		cl_ushort	blk_index;		/* ROW: which page of this chunk */
		cl_ushort	item_offset;	/* ROW: 1-based line pointer number */
	};
	cl_uint			htup_offset;	/* ROW_FLAT: byte offset of the tuple */
} kern_rowitem;

typedef struct {
	cl_int			buffer;			/* PostgreSQL Buffer id (opaque here) */
	hostptr_t		page;			/* host address of the page (opaque here) */
} kern_blkitem;

#define KDS_FORMAT_ROW			1
#define KDS_FORMAT_ROW_FLAT		2
#define KDS_FORMAT_TUPSLOT		3
#define KDS_FORMAT_COLUMN		4	/* new: column-major, see below */

typedef struct {
	hostptr_t		hostptr;		/* host address of this kds (for pointer fix-ups) */
	cl_uint			length;
	cl_uint			usage;
	cl_uint			ncols;
	cl_uint			nitems;
	cl_uint			nrooms;
	cl_uint			nblocks;
	cl_uint			maxblocks;
	cl_char			format;
	cl_char			tdhasoid;
	cl_uint			tdtypeid;
	cl_int			tdtypmod;
	kern_colmeta	colmeta[1];		/* really [ncols] */
} kern_data_store;

#define KDS_HEAD_LENGTH(ncols)	\
	STROMALIGN(offsetof(kern_data_store, colmeta) + sizeof(kern_colmeta) * (ncols))

/* ROW / ROW_FLAT */
#define KERN_DATA_STORE_BLKITEM(kds,blk_index)	\
	(((kern_blkitem *)((char *)(kds) + KDS_HEAD_LENGTH((kds)->ncols))) + (blk_index))
#define KERN_DATA_STORE_ROWITEM(kds,row_index)	\
	(((kern_rowitem *)((char *)(kds) + KDS_HEAD_LENGTH((kds)->ncols) + \
					   STROMALIGN(sizeof(kern_blkitem) * (kds)->maxblocks))) + (row_index))
#define KERN_DATA_STORE_ROWBLOCK_OFFSET(kds)	\
	STROM_TYPEALIGN(BLCKSZ, KDS_HEAD_LENGTH((kds)->ncols) + \
					STROMALIGN(sizeof(kern_blkitem) * (kds)->maxblocks) + \
					STROMALIGN(sizeof(kern_rowitem) * (kds)->nitems))
#define KERN_DATA_STORE_ROWBLOCK(kds,blk_index)	\
	((char *)(kds) + KERN_DATA_STORE_ROWBLOCK_OFFSET(kds) + (size_t)BLCKSZ * (blk_index))

/* TUPSLOT: per row  Datum values[ncols]; char isnull[ncols]; LONGALIGNed */
#define KDS_TUPSLOT_STRIDE(ncols)	STROM_LONGALIGN((sizeof(Datum) + sizeof(cl_char)) * (ncols))
#define KERN_DATA_STORE_VALUES(kds,row_index)	\
	((Datum *)((char *)(kds) + KDS_HEAD_LENGTH((kds)->ncols) + \
			   KDS_TUPSLOT_STRIDE((kds)->ncols) * (size_t)(row_index)))
#define KERN_DATA_STORE_ISNULL(kds,row_index)	\
	((cl_char *)(KERN_DATA_STORE_VALUES((kds),(row_index)) + (kds)->ncols))

/*
 * KDS_FORMAT_COLUMN (this build)
 *
 *   kern_data_store head + colmeta[ncols]              (as every format)
 *   kern_coldir coldir[ncols]                          (STROMALIGNed)
 *   for each column, KDS_COLUMN_ALIGN-aligned:
 *       values[nrooms]   attlen bytes each, row i at values + i*attlen
 *       notnull bitmap   (nrooms+31)/32 words, bit (i&31) of word i>>5 is
 *                        1 when row i is NOT NULL -- the polarity of a heap
 *                        tuple's t_bits; absent (nulls_off == 0) when the
 *                        column has no NULL in this chunk
 *
 * All offsets are bytes from the kds head.  'length' covers everything, so
 * a chunk is one contiguous upload like the other formats.  Rows are
 * already visibility-filtered at ingest (the ROW format's kern_rowitem
 * step); row i of a COLUMN chunk is simply index i of every array.
 * Fixed-width by-value columns (attlen 1,2,4,8) are carried column-major as
 * they are.  A VARLENA column (attlen -1: text, character(n)) carries, per row,
 * the 8-byte OFFSET of the row's datum from the kds head (0 == NULL; the
 * bitmap says so too), and the datums themselves -- complete varlenas, header
 * included, exactly the bytes a heap tuple holds (opencl_common.h:1126-1154
 * pg_varlena_t points at such bytes) -- lie in a heap area of the chunk,
 * anywhere between the column arrays and 'length' (extra_off = where the
 * column's datums start, informational; 'usage' = end of the last datum).  8
 * bytes per row so that the streaming kernels load the column like any int8
 * column and turn offset into address with one add (strom_kvars_from_column).
 * A datum with a 4-byte header is 4-byte aligned, as in a heap tuple.
 *
 * zone map: minval/maxval hold the chunk-wide minimum / maximum of the
 * column (as int64 for integer-like, as IEEE double bits for float4/8)
 * when KDS_COLSTAT_MINMAX is set in stat_flags.  GpuPreAgg uses it to
 * pick a direct-indexed LDS table when (max-min) is small.
 */
#define KDS_COLUMN_ALIGN		256
#define KDS_COLSTAT_MINMAX		0x0001
#define KDS_COLSTAT_ISFLOAT		0x0002
/* a column of 64-bit NUMERIC images (their bit patterns do not order like their values): minval /
 * maxval bound the VALUES, rounded outward to integers -- floor(min), ceil(max) -- so that
 * |value| <= max(|minval|, |maxval|).  What GpuPreAgg needs to bound a sum over the column without
 * looking at the rows.  Set instead of KDS_COLSTAT_MINMAX, never with it. */
#define KDS_COLSTAT_INTPART		0x0004

typedef struct {
	cl_uint			values_off;
	cl_uint			nulls_off;		/* 0 = no NULLs */
	cl_uint			extra_off;		/* 0 = none */
	cl_uint			stat_flags;
	cl_long			minval;
	cl_long			maxval;
} kern_coldir;

#define KERN_DATA_STORE_COLDIR(kds)	\
	((kern_coldir *)((char *)(kds) + KDS_HEAD_LENGTH((kds)->ncols)))
#define KDS_COLUMN_HEAD_LENGTH(ncols)	\
	STROM_TYPEALIGN(KDS_COLUMN_ALIGN, KDS_HEAD_LENGTH(ncols) + sizeof(kern_coldir) * (ncols))
/* bytes per row in a column's value array: the datum, or the offset of a varlena */
#define KDS_COLUMN_ATTWIDTH(attlen)		((attlen) > 0 ? (size_t)(attlen) : sizeof(cl_ulong))
#define KDS_COLUMN_VALUES_LENGTH(attlen,nrooms)	\
	STROM_TYPEALIGN(KDS_COLUMN_ALIGN, KDS_COLUMN_ATTWIDTH(attlen) * (nrooms))
#define KDS_COLUMN_NULLS_LENGTH(nrooms)	\
	STROM_TYPEALIGN(KDS_COLUMN_ALIGN, sizeof(cl_uint) * (((size_t)(nrooms) + 31) / 32))

/* ----------------------------------------------------------------
 * kern_parambuf / kern_resultbuf / kern_row_map
 * ---------------------------------------------------------------- */
typedef struct {
	cl_uint		length;			/* total bytes incl. this head */
	cl_uint		nparams;
	cl_uint		poffset[1];		/* really [nparams]; 0 == NULL parameter */
} kern_parambuf;

typedef struct {
	cl_uint		nrels;
	cl_uint		nrooms;
	cl_uint		nitems;
	cl_int		errcode;
	cl_char		has_rechecks;
	cl_char		all_visible;
	cl_char		__padding__[2];
	cl_int		results[1];		/* really [nrels * nrooms] */
} kern_resultbuf;

typedef struct {
	cl_int		nvalids;		/* -1: every row of the chunk is valid */
	cl_int		rindex[1];
} kern_row_map;

/* ----------------------------------------------------------------
 * kern_gpuscan : { kern_parambuf ; kern_resultbuf } back to back
 * ---------------------------------------------------------------- */
typedef struct {
	kern_parambuf	kparams;
} kern_gpuscan;

#define KERN_GPUSCAN_PARAMBUF(kgs)			((kern_parambuf *)(&(kgs)->kparams))
#define KERN_GPUSCAN_PARAMBUF_LENGTH(kgs)	STROMALIGN((kgs)->kparams.length)
#define KERN_GPUSCAN_RESULTBUF(kgs)			\
	((kern_resultbuf *)((char *)&(kgs)->kparams + STROMALIGN((kgs)->kparams.length)))
#define KERN_RESULTBUF_LENGTH(nrels,nrooms)	\
	STROMALIGN(offsetof(kern_resultbuf, results) + sizeof(cl_int) * (size_t)(nrels) * (size_t)(nrooms))
#define KERN_GPUSCAN_RESULTBUF_LENGTH(kgs)	\
	KERN_RESULTBUF_LENGTH(KERN_GPUSCAN_RESULTBUF(kgs)->nrels, KERN_GPUSCAN_RESULTBUF(kgs)->nrooms)
#define KERN_GPUSCAN_LENGTH(kgs)			\
	(KERN_GPUSCAN_PARAMBUF_LENGTH(kgs) + KERN_GPUSCAN_RESULTBUF_LENGTH(kgs))
#define KERN_GPUSCAN_DMASEND_LENGTH(kgs)	\
	(KERN_GPUSCAN_PARAMBUF_LENGTH(kgs) + offsetof(kern_resultbuf, results))

/* ----------------------------------------------------------------
 * hash join: chained table that owns whole inner heap tuples
 * ---------------------------------------------------------------- */
typedef struct {
	cl_uint		t_xmin, t_xmax, t_field3;		/* t_choice (12 bytes) */
	cl_ushort	bi_hi, bi_lo, ip_posid;			/* t_ctid */
	cl_ushort	t_infomask2;
	cl_ushort	t_infomask;
	cl_uchar	t_hoff;
	cl_uchar	t_bits[1];
} HeapTupleHeaderData;

#define HEAP_HASNULL		0x0001
#define HEAP_HASVARWIDTH	0x0002
#define HEAP_HASEXTERNAL	0x0004
#define HEAP_HASOID			0x0008
#define HEAP_NATTS_MASK		0x07FF
#define HEAPTUPLE_HEADER_FIXED	23		/* offsetof(HeapTupleHeaderData, t_bits) */

typedef struct {
	cl_uint		next;		/* byte offset of next entry in chain, 0 = end */
	cl_uint		hash;
	cl_uint		rowid;
	cl_uint		t_len;
	HeapTupleHeaderData htup;
} kern_hashentry;

typedef struct {
	cl_uint		length;
	cl_uint		ncols;
	cl_uint		nslots;
	cl_char		is_outer;
	cl_char		__padding__[3];
	kern_colmeta colmeta[1];
} kern_hashtable;

typedef struct {
	hostptr_t	hostptr;
	cl_uint		pg_crc32_table[256];
	cl_uint		ntables;
	cl_uint		htable_offset[1];
} kern_multihash;

#define KERN_HASHTABLE(kmhash,depth)	\
	((kern_hashtable *)((char *)(kmhash) + (kmhash)->htable_offset[(depth)]))
#define KERN_HASHTABLE_SLOT(khtable)	\
	((cl_uint *)((char *)(khtable) + \
				 STROM_LONGALIGN(offsetof(kern_hashtable, colmeta) + \
								 sizeof(kern_colmeta) * (khtable)->ncols)))
#define KERN_HASHENTRY_SIZE_BY_TLEN(t_len)	\
	STROM_LONGALIGN(offsetof(kern_hashentry, htup) + (t_len))

typedef struct {
	kern_parambuf	kparams;
} kern_hashjoin;

#define KERN_HASHJOIN_PARAMBUF(khj)			((kern_parambuf *)(&(khj)->kparams))
#define KERN_HASHJOIN_PARAMBUF_LENGTH(khj)	STROMALIGN((khj)->kparams.length)
#define KERN_HASHJOIN_RESULTBUF(khj)		\
	((kern_resultbuf *)((char *)&(khj)->kparams + KERN_HASHJOIN_PARAMBUF_LENGTH(khj)))
#define KERN_HASHJOIN_RESULTBUF_LENGTH(khj)	STROMALIGN(offsetof(kern_resultbuf, results))
#define KERN_HASHJOIN_ROWMAP(khj)			\
	((kern_row_map *)((char *)KERN_HASHJOIN_RESULTBUF(khj) + KERN_HASHJOIN_RESULTBUF_LENGTH(khj)))

/* ----------------------------------------------------------------
 * preagg control block
 * ---------------------------------------------------------------- */
typedef struct {
	cl_int			status;
	cl_int			sortbuf_len;
	char			__padding[8];
	kern_parambuf	kparams;
	/* kern_row_map, then sort rindex[] in the reference; unused by the
	 * hash-based reduction of this build but kept addressable */
} kern_gpupreagg;

#define KERN_GPUPREAGG_PARAMBUF(kgp)		((kern_parambuf *)(&(kgp)->kparams))
#define KERN_GPUPREAGG_KROWMAP(kgp)			\
	((kern_row_map *)((char *)(kgp) + \
					  STROMALIGN(offsetof(kern_gpupreagg, kparams) + (kgp)->kparams.length)))

/* per-output-column role flags carried in KPARAM_0 (opencl_gpupreagg.h:129-137) */
#define GPUPREAGG_FIELD_IS_NULL			0
#define GPUPREAGG_FIELD_IS_GROUPKEY		1
#define GPUPREAGG_FIELD_IS_AGGFUNC		2

/* ----------------------------------------------------------------
 * SQL type tags used by the expression IR and the builders
 * (numeric values are PostgreSQL's pg_type OIDs)
 * ---------------------------------------------------------------- */
#define STROM_BOOLOID		16
#define STROM_BYTEAOID		17
#define STROM_INT8OID		20
#define STROM_INT2OID		21
#define STROM_INT4OID		23
#define STROM_TEXTOID		25
#define STROM_FLOAT4OID		700
#define STROM_FLOAT8OID		701
#define STROM_BPCHAROID		1042
/* character(n) as the varlena PostgreSQL stores (pg_type oid 1042 as well; the
 * expression IR tells it from the by-value char(1) above by its own tag) */
#define STROM_BPCHARNOID		(0x10000 | 1042)
#define STROM_DATEOID		1082
#define STROM_TIMEOID		1083
#define STROM_TIMESTAMPOID	1114
#define STROM_NUMERICOID	1700
/* numeric(p,s) held as a scaled int8 -- "decimal64": value * 10^s -- in a KDS_FORMAT_COLUMN
 * chunk (pg_type oid 1700 as well; the expression IR names such a column (var N decimal S)).
 * A storage choice of the ingest step for typmod-scaled columns: the kernels then do integer
 * arithmetic from the first instruction instead of decoding the 64-bit float-decimal image
 * per row (TPC-H Q1 shape: 1.44 -> 0.83 ms per 1e8 rows). */
#define STROM_DECIMALOID	(0x10000 | 1700)
/* strom_dstore_to_column(): "turn this numeric column into a decimal column at SCALE" */
#define STROM_DECIMAL_TYPE(scale)		(STROM_DECIMALOID | ((scale) << 20))
#define STROM_TYPE_IS_DECIMAL(oid)		(((oid) & 0xfffff) == STROM_DECIMALOID)
#define STROM_DECIMAL_TYPE_SCALE(oid)	(((oid) >> 20) & 0x3f)

#endif	/* STROM_KDS_H */
