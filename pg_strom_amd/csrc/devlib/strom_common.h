/*
 * strom_common.h -- device-side common library (HIP, gfx950 only)
 *
 * Prepended (after strom_kds.h) to every program the runtime hands to
 * hiprtc; the generated expression function and one kernel skeleton
 * (strom_gpuscan.h / strom_hashjoin.h / strom_gpupreagg.h) follow it.
 * Role in the reference: the device half of opencl_common.h
 *   - pg_<type>_t {value,isnull} and vref/param/isnull accessors
 *                                   (opencl_common.h:530-670)
 *   - kern_get_datum for ROW / ROW_FLAT / TUPSLOT (opencl_common.h:817-981)
 *   - STROM_SET_ERROR priority rule (opencl_common.h:132-144)
 *   - work-group prefix sum (arithmetic_stairlike_add, 1446-1470) and
 *     chunk status write-back (kern_writeback_error_status, 1481-1527)
 *   - 3-valued bool helpers (1539-1622)
 * None of the OpenCL text is reused: the collectives below are wave64
 * ballot/mbcnt based, and a COLUMN-format vector tile loader is added.
 */
#ifndef STROM_COMMON_DEVICE_H
#define STROM_COMMON_DEVICE_H

#define STROM_WAVE			64
#define STROM_DEVICE		static __device__ __forceinline__

typedef unsigned long long	strom_lanemask_t;

/* ---------------------------------------------------------------- *
 * publishing a record other work-groups (other XCDs) probe
 *
 * The open-addressing tables in HBM -- the hashed GROUP BY's records
 * (strom_gpupreagg.h) and the KEYED join index (strom_hashjoin.h) -- are
 * claimed with a compare-and-swap on a state word, filled, and marked ready.
 * Formally that is: payload stores, then a RELEASE store of the state;
 * probes: an ACQUIRE load of the state, then the payload.  At agent scope on
 * a chip whose eight L2s are kept coherent by write-back / invalidate those
 * two are an L2 write-back per claim and a cache invalidate per probe
 * (measured: 28 of 42 ms).  Neither is needed when EVERY
 * access to the record -- state and payload, reads and writes -- is an
 * agent-scope atomic: those are performed at the coherence point itself.  What
 * is left is ordering the claimer's own stores: the payload stores must have
 * been acknowledged before the state store is issued.
 *
 *   STROM_PUBLISH_STATE(ptr, val)   store 'val' to the state word once all
 *                                   earlier stores of this wave are acknowledged
 *   STROM_PROBE_STATE(ptr)          load the state word
 *
 * Fast form: s_waitcnt vmcnt(0), in the gfx9 encoding (vmcnt bits 3:0 and
 * 15:14, expcnt 6:4, lgkmcnt 11:8 -> 0x0f70 leaves the other counters alone;
 * stores count in vmcnt on gfx9, gfx10+ moved them to vscnt): gfx9 family
 * ONLY, asserted below.  -DSTROM_FORMAL_PUBLISH=1 (runtime knob
 * STROM_FORMAL_PUBLISH) builds the RELEASE / ACQUIRE form instead -- the
 * reference semantics to test the fast form against, and the fallback for
 * any other target.  The payload of a record lies in the 64-byte line of its
 * state word (16-byte KEYED slots; GROUP BY records are aligned to their
 * power-of-two stride and keep state, NULL bits and up to 6 keys in the first
 * 64 bytes); a probe that matched re-reads the state word after the payload
 * and goes round again if it changed (it cannot: states only move
 * empty -> busy -> ready; the re-read costs an L2 hit and closes the door on
 * a payload that was read before the state it belongs to).
 * ---------------------------------------------------------------- */
#if !defined(STROM_FORMAL_PUBLISH)
#define STROM_FORMAL_PUBLISH	0
#endif
#if !STROM_FORMAL_PUBLISH && !defined(__GFX9__)
#error "STROM_PUBLISH_STATE's s_waitcnt encoding is gfx9's: build with -DSTROM_FORMAL_PUBLISH=1 for this target"
#endif
#if STROM_FORMAL_PUBLISH
#define STROM_PUBLISH_STATE(ptr, val)	\
	__hip_atomic_store((ptr), (val), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define STROM_PROBE_STATE(ptr)			\
	__hip_atomic_load((ptr), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#else
#define STROM_PUBLISH_STATE(ptr, val)	\
	do {								\
		__builtin_amdgcn_s_waitcnt(0x0f70);		/* vmcnt(0): the payload stores are acknowledged */	\
		__hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);	\
	} while (0)
#define STROM_PROBE_STATE(ptr)			\
	__hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

/* ---------------------------------------------------------------- *
 * error priority: significant (>=100 or <0) sticks, first one wins;
 * among minor codes the larger wins (CpuReCheck=2 > RowFiltered=1 > 0)
 * ---------------------------------------------------------------- */
STROM_DEVICE void
STROM_SET_ERROR(cl_int *p_error, cl_int errcode)
{
	cl_int	oldcode = *p_error;

	if (StromErrorIsSignificant(errcode))
	{
		if (!StromErrorIsSignificant(oldcode))
			*p_error = errcode;
	}
	else if (!StromErrorIsSignificant(oldcode) && errcode > oldcode)
		*p_error = errcode;
}

/* the common case in arithmetic, without a branch: raise CpuReCheck when
 * 'cond' holds (a select; per-row "if"s cost a taken skip-branch each in
 * the usual no-error case) */
STROM_DEVICE void
STROM_SET_RECHECK_IF(cl_int *p_error, bool cond)
{
	cl_int	oldcode = *p_error;

	*p_error = ((cond & (oldcode >= 0) & (oldcode < StromError_CpuReCheck))
				? StromError_CpuReCheck : oldcode);
}

/* ---------------------------------------------------------------- *
 * SQL value = { BASE value; bool isnull; }
 * ---------------------------------------------------------------- */
#define STROM_DECLARE_SIMPLE_TYPE(NAME,BASE)				\
	typedef struct {										\
		BASE	value;										\
		bool	isnull;										\
	} pg_##NAME##_t;										\
	typedef BASE pg_##NAME##_base_t;

STROM_DECLARE_SIMPLE_TYPE(bool,   cl_bool)
STROM_DECLARE_SIMPLE_TYPE(int2,   cl_short)
STROM_DECLARE_SIMPLE_TYPE(int4,   cl_int)
STROM_DECLARE_SIMPLE_TYPE(int8,   cl_long)
STROM_DECLARE_SIMPLE_TYPE(float4, cl_float)
STROM_DECLARE_SIMPLE_TYPE(float8, cl_double)
/* date = days since 2000-01-01, time/timestamp = microseconds (int64) */
STROM_DECLARE_SIMPLE_TYPE(date,      cl_int)
STROM_DECLARE_SIMPLE_TYPE(time,      cl_long)
STROM_DECLARE_SIMPLE_TYPE(timestamp, cl_long)
/* bpchar(1) carried by value in COLUMN chunks ("char" column) */
STROM_DECLARE_SIMPLE_TYPE(char1,  cl_char)

/* ---------------------------------------------------------------- *
 * varlena header helpers (only what the tuple walker needs)
 * ---------------------------------------------------------------- */
STROM_DEVICE bool strom_varatt_is_1b(const char *p)
{ return (((const cl_uchar *)p)[0] & 0x01) == 0x01; }
STROM_DEVICE bool strom_varatt_is_1b_e(const char *p)
{ return ((const cl_uchar *)p)[0] == 0x01; }
STROM_DEVICE bool strom_varatt_not_pad_byte(const char *p)
{ return ((const cl_uchar *)p)[0] != 0; }
STROM_DEVICE cl_uint strom_varsize_any(const char *p)
{
	cl_uchar	b0 = ((const cl_uchar *)p)[0];

	if (b0 == 0x01)
	{
		/* external TOAST pointer: header(2) + payload by tag */
		cl_uchar tag = ((const cl_uchar *)p)[1];
		return 2 + (tag == 18 ? 16 : 8);
	}
	if (b0 & 0x01)
		return (b0 >> 1) & 0x7f;
	/* 4-byte header, little endian, length in the upper 30 bits */
	cl_uint w = ((cl_uint)((const cl_uchar *)p)[0])
		| ((cl_uint)((const cl_uchar *)p)[1] << 8)
		| ((cl_uint)((const cl_uchar *)p)[2] << 16)
		| ((cl_uint)((const cl_uchar *)p)[3] << 24);
	return (w >> 2) & 0x3fffffff;
}

/* ---------------------------------------------------------------- *
 * heap page / line pointer (x86-64 little-endian bit-field order, the
 * values the reference probes at run time: opencl_devprog.c:949-974)
 * ---------------------------------------------------------------- */
#define STROM_ITEMID_OFFSET(itemid)		((itemid) & 0x7fff)
#define STROM_ITEMID_FLAGS(itemid)		(((itemid) >> 15) & 0x0003)
#define STROM_ITEMID_LENGTH(itemid)		(((itemid) >> 17) & 0x7fff)
#define STROM_PAGE_HEADER_SIZE			24		/* offsetof(PageHeaderData, pd_linp) */
#define STROM_PAGE_PD_LOWER_OFF			12

/*
 * Locate attribute 'colidx' inside one heap tuple.  NULL when the
 * attribute is SQL NULL or colidx is past the tuple's natts.
 */
STROM_DEVICE const char *
kern_get_datum_tuple(const kern_colmeta *colmeta,
					 const HeapTupleHeaderData *htup,
					 cl_uint colidx)
{
	bool	hasnull = ((htup->t_infomask & HEAP_HASNULL) != 0);
	cl_uint	offset = htup->t_hoff;
	cl_uint	natts = (htup->t_infomask2 & HEAP_NATTS_MASK);

	if (colidx >= natts)
		return NULL;
	if (!hasnull)
	{
		cl_short	cacheoff = colmeta[colidx].attcacheoff;

		if (cacheoff >= 0)
			return (const char *)htup + cacheoff;
	}
	for (cl_uint i = 0; i < natts; i++)
	{
		if (hasnull && !(htup->t_bits[i >> 3] & (1 << (i & 7))))
		{
			if (i == colidx)
				return NULL;
			continue;
		}
		kern_colmeta	cm = colmeta[i];
		const char	   *base = (const char *)htup;

		if (cm.attlen > 0)
			offset = STROM_TYPEALIGN(cm.attalign, offset);
		else if (!strom_varatt_not_pad_byte(base + offset))
			offset = STROM_TYPEALIGN(cm.attalign, offset);
		if (i == colidx)
			return base + offset;
		offset += (cm.attlen > 0 ? (cl_uint)cm.attlen
				   : strom_varsize_any(base + offset));
	}
	return NULL;
}

STROM_DEVICE const HeapTupleHeaderData *
kern_get_tuple_rs(const kern_data_store *kds, cl_uint rowidx)
{
	if (rowidx >= kds->nitems)
		return NULL;
	const kern_rowitem *ritem = KERN_DATA_STORE_ROWITEM(kds, rowidx);
	cl_ushort	blk_index = ritem->blk_index;
	cl_ushort	item_offset = ritem->item_offset;

	if (blk_index >= kds->nblocks)
		return NULL;
	const char *page = KERN_DATA_STORE_ROWBLOCK(kds, blk_index);
	cl_ushort	pd_lower = *(const cl_ushort *)(page + STROM_PAGE_PD_LOWER_OFF);
	cl_uint		item_max = (pd_lower <= STROM_PAGE_HEADER_SIZE ? 0 :
							(pd_lower - STROM_PAGE_HEADER_SIZE) / sizeof(cl_uint));
	/* phantom-page paranoia, as the reference: never chase a wild offset */
	if (STROM_PAGE_HEADER_SIZE + sizeof(cl_uint) * (item_max + 1) >= BLCKSZ ||
		item_offset == 0 || item_offset > item_max)
		return NULL;
	cl_uint		itemid = ((const cl_uint *)(page + STROM_PAGE_HEADER_SIZE))[item_offset - 1];
	if (STROM_ITEMID_OFFSET(itemid) + HEAPTUPLE_HEADER_FIXED >= BLCKSZ)
		return NULL;
	return (const HeapTupleHeaderData *)(page + STROM_ITEMID_OFFSET(itemid));
}

STROM_DEVICE const HeapTupleHeaderData *
kern_get_tuple_rsflat(const kern_data_store *kds, cl_uint rowidx)
{
	if (rowidx >= kds->nitems)
		return NULL;
	cl_uint	off = KERN_DATA_STORE_ROWITEM(kds, rowidx)->htup_offset;
	if (off >= kds->length)
		return NULL;
	return (const HeapTupleHeaderData *)((const char *)kds + off);
}

/*
 * Generic accessor, any format.  Returns the address of the datum or NULL
 * for SQL NULL / out of range.
 */
STROM_DEVICE const void *
kern_get_datum(const kern_data_store *kds,
			   const kern_data_store *ktoast,
			   cl_uint colidx, cl_uint rowidx)
{
	if (colidx >= kds->ncols || rowidx >= kds->nitems)
		return NULL;
	switch (kds->format)
	{
		case KDS_FORMAT_COLUMN:
			{
				const kern_coldir *cd = KERN_DATA_STORE_COLDIR(kds) + colidx;
				if (cd->nulls_off != 0)
				{
					const cl_uint *nn = (const cl_uint *)((const char *)kds + cd->nulls_off);
					if (!((nn[rowidx >> 5] >> (rowidx & 31)) & 1))
						return NULL;
				}
				cl_short attlen = kds->colmeta[colidx].attlen;
				if (attlen > 0)
					return (const char *)kds + cd->values_off + (size_t)attlen * rowidx;
				/* a varlena column: the offset of the row's datum from the kds head */
				cl_ulong voff = ((const cl_ulong *)((const char *)kds + cd->values_off))[rowidx];
				if (voff == 0 || voff >= kds->length)
					return NULL;
				return (const char *)kds + voff;
			}
		case KDS_FORMAT_ROW:
			{
				const HeapTupleHeaderData *htup = kern_get_tuple_rs(kds, rowidx);
				return htup ? kern_get_datum_tuple(kds->colmeta, htup, colidx) : NULL;
			}
		case KDS_FORMAT_ROW_FLAT:
			{
				const HeapTupleHeaderData *htup = kern_get_tuple_rsflat(kds, rowidx);
				return htup ? kern_get_datum_tuple(kds->colmeta, htup, colidx) : NULL;
			}
		case KDS_FORMAT_TUPSLOT:
			{
				const Datum	   *values = KERN_DATA_STORE_VALUES(kds, rowidx);
				const cl_char  *isnull = KERN_DATA_STORE_ISNULL(kds, rowidx);
				if (isnull[colidx])
					return NULL;
				if (kds->colmeta[colidx].attlen > 0)
					return values + colidx;
				return (const char *)ktoast + values[colidx];
			}
		default:
			return NULL;
	}
}

/*
 * A value every lane of the wave holds alike (launch geometry, column base
 * addresses, directory entries): say so.  Read through a pointer the compiler
 * cannot prove invariant, such a value lands in a VECTOR register -- one copy
 * per lane, vector ALU for every use, spills in the row loops; through
 * readfirstlane it lives in scalar registers.
 */
template <typename T>
STROM_DEVICE T strom_uniform(T v)
{
	if (sizeof(T) == 8)
	{
		unsigned long long bits;
		__builtin_memcpy(&bits, &v, 8);
		unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)bits);
		unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(bits >> 32));
		bits = ((unsigned long long)hi << 32) | lo;
		__builtin_memcpy(&v, &bits, 8);
		return v;
	}
	else
	{
		unsigned int bits = 0;
		__builtin_memcpy(&bits, &v, sizeof(T));
		bits = (unsigned int)__builtin_amdgcn_readfirstlane((int)bits);
		__builtin_memcpy(&v, &bits, sizeof(T));
		return v;
	}
}

/* unaligned-safe scalar fetch (heap tuples only guarantee attalign) */
template <typename BASE>
STROM_DEVICE BASE strom_fetch(const void *addr)
{
	BASE v;
	__builtin_memcpy(&v, addr, sizeof(BASE));
	return v;
}

/*
 * FROM_ADDR(errcode, addr, attlen) turns a located datum into the SQL value;
 * fixed-width types copy attlen bytes, NUMERIC also decodes PostgreSQL's
 * varlena form (strom_numeric.h).
 */
#define STROM_DECLARE_VARREF_CORE(NAME, FROM_ADDR)							\
	STROM_DEVICE pg_##NAME##_t												\
	pg_##NAME##_vref(const kern_data_store *kds,							\
					 const kern_data_store *ktoast,							\
					 cl_int *errcode, cl_uint colidx, cl_uint rowidx)		\
	{																		\
		pg_##NAME##_t	result;												\
		const void	   *addr = kern_get_datum(kds, ktoast, colidx, rowidx);	\
		if (!addr)															\
		{																	\
			result.isnull = true;											\
			result.value = 0;												\
		}																	\
		else																\
			result = FROM_ADDR(errcode, (const char *)addr,					\
							   kds->colmeta[colidx].attlen);				\
		return result;														\
	}																		\
	/* attribute of a bare heap tuple (row formats; inner tuple of a hash	\
	 * entry) */															\
	STROM_DEVICE pg_##NAME##_t												\
	pg_##NAME##_tupref(cl_int *errcode, const kern_colmeta *colmeta,		\
					   const HeapTupleHeaderData *htup, cl_uint colidx)		\
	{																		\
		pg_##NAME##_t	result;												\
		const void	   *addr = kern_get_datum_tuple(colmeta, htup, colidx);	\
		if (!addr)															\
		{																	\
			result.isnull = true;											\
			result.value = 0;												\
		}																	\
		else																\
			result = FROM_ADDR(errcode, (const char *)addr,					\
							   colmeta[colidx].attlen);						\
		return result;														\
	}																		\
	STROM_DEVICE pg_##NAME##_t												\
	pg_##NAME##_make(pg_##NAME##_base_t value, bool isnull)					\
	{																		\
		pg_##NAME##_t	result;												\
		result.value = value;												\
		result.isnull = isnull;												\
		return result;														\
	}																		\
	STROM_DEVICE pg_bool_t													\
	pgfn_##NAME##_isnull(cl_int *errcode, pg_##NAME##_t arg)				\
	{																		\
		pg_bool_t r; r.isnull = false; r.value = arg.isnull; return r;		\
	}																		\
	STROM_DEVICE pg_bool_t													\
	pgfn_##NAME##_isnotnull(cl_int *errcode, pg_##NAME##_t arg)				\
	{																		\
		pg_bool_t r; r.isnull = false; r.value = !arg.isnull; return r;		\
	}

/* a by-value parameter: its image sits in the kern_parambuf at poffset[] */
#define STROM_DECLARE_PARAMREF_BYVAL(NAME)									\
	STROM_DEVICE pg_##NAME##_t												\
	pg_##NAME##_param(const kern_parambuf *kparams,							\
					  cl_int *errcode, cl_uint param_id)					\
	{																		\
		pg_##NAME##_t	result;												\
		if (param_id < kparams->nparams &&									\
			kparams->poffset[param_id] > 0)									\
		{																	\
			result.value = strom_fetch<pg_##NAME##_base_t>					\
				((const char *)kparams + kparams->poffset[param_id]);		\
			result.isnull = false;											\
		}																	\
		else																\
		{																	\
			result.isnull = true;											\
			result.value = 0;												\
		}																	\
		return result;														\
	}

#define STROM_DECLARE_VARREF_EX(NAME, FROM_ADDR)							\
	STROM_DECLARE_VARREF_CORE(NAME, FROM_ADDR)								\
	STROM_DECLARE_PARAMREF_BYVAL(NAME)

#define STROM_DECLARE_VARREF(NAME)											\
	STROM_DEVICE pg_##NAME##_t												\
	pg_##NAME##_from_addr(cl_int *errcode, const char *addr, cl_short attlen)	\
	{																		\
		pg_##NAME##_t	result;												\
		result.isnull = false;												\
		result.value = strom_fetch<pg_##NAME##_base_t>(addr);				\
		return result;														\
	}																		\
	STROM_DECLARE_VARREF_EX(NAME, pg_##NAME##_from_addr)

STROM_DECLARE_VARREF(bool)
STROM_DECLARE_VARREF(int2)
STROM_DECLARE_VARREF(int4)
STROM_DECLARE_VARREF(int8)
STROM_DECLARE_VARREF(float4)
STROM_DECLARE_VARREF(float8)
STROM_DECLARE_VARREF(date)
STROM_DECLARE_VARREF(time)
STROM_DECLARE_VARREF(timestamp)
STROM_DECLARE_VARREF(char1)

/* ---------------------------------------------------------------- *
 * 3-valued logic
 * ---------------------------------------------------------------- */
STROM_DEVICE bool EVAL(pg_bool_t arg)
{ return !arg.isnull && arg.value != 0; }

STROM_DEVICE pg_bool_t pgfn_bool_is_true(cl_int *e, pg_bool_t a)
{ pg_bool_t r; r.value = (!a.isnull && a.value); r.isnull = false; return r; }
STROM_DEVICE pg_bool_t pgfn_bool_is_not_true(cl_int *e, pg_bool_t a)
{ pg_bool_t r; r.value = (a.isnull || !a.value); r.isnull = false; return r; }
STROM_DEVICE pg_bool_t pgfn_bool_is_false(cl_int *e, pg_bool_t a)
{ pg_bool_t r; r.value = (!a.isnull && !a.value); r.isnull = false; return r; }
STROM_DEVICE pg_bool_t pgfn_bool_is_not_false(cl_int *e, pg_bool_t a)
{ pg_bool_t r; r.value = (a.isnull || a.value); r.isnull = false; return r; }
STROM_DEVICE pg_bool_t pgfn_bool_is_unknown(cl_int *e, pg_bool_t a)
{ pg_bool_t r; r.value = a.isnull; r.isnull = false; return r; }
STROM_DEVICE pg_bool_t pgfn_bool_is_not_unknown(cl_int *e, pg_bool_t a)
{ pg_bool_t r; r.value = !a.isnull; r.isnull = false; return r; }
STROM_DEVICE pg_bool_t pgfn_boolop_not(cl_int *e, pg_bool_t a)
{ a.value = !a.value; return a; }
/*
 * AND / OR follow SQL (Kleene) semantics: FALSE AND NULL = FALSE,
 * TRUE OR NULL = TRUE.  The reference's generated boolop_and_N returns NULL
 * whenever any argument is NULL (codegen.c:816-861); as a WHERE clause both
 * drop the row, so qualifying row sets are identical.
 */
STROM_DEVICE pg_bool_t pgfn_boolop_and2(pg_bool_t a, pg_bool_t b)
{
	pg_bool_t r;
	bool	a_false = (!a.isnull && !a.value);
	bool	b_false = (!b.isnull && !b.value);
	if (a_false || b_false)		{ r.isnull = false; r.value = false; }
	else if (a.isnull || b.isnull)	{ r.isnull = true;  r.value = false; }
	else							{ r.isnull = false; r.value = true; }
	return r;
}
STROM_DEVICE pg_bool_t pgfn_boolop_or2(pg_bool_t a, pg_bool_t b)
{
	pg_bool_t r;
	bool	a_true = (!a.isnull && a.value);
	bool	b_true = (!b.isnull && b.value);
	if (a_true || b_true)			{ r.isnull = false; r.value = true; }
	else if (a.isnull || b.isnull)	{ r.isnull = true;  r.value = false; }
	else							{ r.isnull = false; r.value = false; }
	return r;
}

#define devfunc_int_comp(x,y)	((x) < (y) ? -1 : ((x) > (y) ? 1 : 0))
#define devfunc_float_comp(x,y)											\
	(__builtin_isnan(x) ? (__builtin_isnan(y) ? 0 : 1)					\
	 : (__builtin_isnan(y) ? -1 : devfunc_int_comp((x),(y))))

/* ---------------------------------------------------------------- *
 * wave64 / work-group collectives
 * ---------------------------------------------------------------- */
STROM_DEVICE cl_uint strom_lane_id(void)
{
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

/* number of set bits of 'mask' in lanes below the caller */
STROM_DEVICE cl_uint strom_mbcnt(strom_lanemask_t mask)
{
	return __builtin_amdgcn_mbcnt_hi((cl_uint)(mask >> 32),
									 __builtin_amdgcn_mbcnt_lo((cl_uint)mask, 0u));
}

STROM_DEVICE cl_int strom_wave_max_i32(cl_int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		cl_int o = __shfl_xor(v, off, STROM_WAVE);
		v = (o > v ? o : v);
	}
	return v;
}

/*
 * First-significant-error-wins chunk status (role of
 * kern_writeback_error_status).  Minor codes must have been cleared by the
 * caller.  One compare-and-swap per wave that saw an error; the common
 * all-zero case costs one ballot.
 */
STROM_DEVICE void
kern_writeback_error_status(cl_int *error_status, cl_int own_errcode)
{
	strom_lanemask_t bad = __ballot(own_errcode != StromError_Success);

	if (bad != 0)
	{
		int		leader = __ffsll((long long)bad) - 1;
		cl_int	code = __shfl(own_errcode, leader, STROM_WAVE);

		if ((int)strom_lane_id() == leader)
			atomicCAS(error_status, StromError_Success, code);
	}
}

/* ---------------------------------------------------------------- *
 * COLUMN-format tile loader
 *
 * A thread owns "quads": 4 consecutive rows starting at a multiple of 4,
 * fetched with one vector load per column (16 B for 4-byte types, 2x16 B
 * for 8-byte types), so a wave's loads of a 4-byte column are one fully
 * coalesced 1 KiB request.  values_off is 256-B aligned by construction.
 * ---------------------------------------------------------------- */
__device__ const cl_uint strom_all_ones[2] = { 0xffffffffu, 0xffffffffu };
/* explicit global address space: a select between two generic pointers
 * would make the load a flat_load */
typedef const __attribute__((address_space(1))) cl_uint *strom_global_uint_p;

template <typename BASE> struct strom_quad {
	typedef BASE vec_t __attribute__((ext_vector_type(4)));
};

/* FULL: the caller knows (wave-uniformly) that the whole tile is in range;
 * the loads then sit in straight-line code and all of a tile's requests
 * are in flight before the first use -- with the per-lane range check
 * every load ends in its own branch join and the waits serialise */
template <typename BASE, bool FULL, bool NONULL = false, bool CACHED = false>
STROM_DEVICE void
strom_column_load_quad(const char *values, const cl_uint *notnull,
					   cl_uint row0, cl_uint nitems,
					   BASE (&v)[4], cl_uint &nnbits)
{
	typedef typename strom_quad<BASE>::vec_t vec_t;

	if (FULL || row0 + 4 <= nitems)
	{
		/* the bitmap word is fetched unconditionally (from an all-ones
		 * word when the column has no NULL): a branch around it would
		 * put a full vmcnt(0) wait between the value loads */
		strom_global_uint_p nnword = (notnull ? (strom_global_uint_p)(notnull + (row0 >> 5))
									  : (strom_global_uint_p)strom_all_ones);
		vec_t	q;
#if !defined(COLUMN_LOAD_NT) || COLUMN_LOAD_NT
		/* streamed once: do not keep the lines in L2 / Infinity Cache --
		 * unless a sibling work-group reads the same tile (CACHED) */
		if (!CACHED)
			q = __builtin_nontemporal_load((const vec_t *)(values + (size_t)row0 * sizeof(BASE)));
		else
#endif
			q = *(const vec_t *)(values + (size_t)row0 * sizeof(BASE));
		cl_uint	w = (NONULL ? 0xffffffffu : *nnword);	/* NONULL: no chunk column has a bitmap */
		v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
		nnbits = (w >> (row0 & 31)) & 0xf;
	}
	else
	{
		nnbits = 0;
#pragma unroll
		for (int j = 0; j < 4; j++)
		{
			cl_uint	r = row0 + j;
			if (r < nitems)
			{
				v[j] = ((const BASE *)values)[r];
				if (!notnull || ((notnull[r >> 5] >> (r & 31)) & 1))
					nnbits |= (1u << j);
			}
			else
				v[j] = 0;
		}
	}
}

/*
 * one datum of a COLUMN chunk from hoisted column pointers: what the
 * row-at-a-time kernels use instead of kern_get_datum() when the chunk is
 * KDS_FORMAT_COLUMN (row maps, several inner relations), so that no chunk
 * header field is read per row.  Non-temporal: the column streams through
 * once and must not evict what the kernel probes at random (hash slots).
 */
#define STROM_COLUMN_REF(NAME, values, notnull, rowidx)								\
	pg_##NAME##_make(__builtin_nontemporal_load(&((const pg_##NAME##_base_t *)(values))[rowidx]),	\
					 (notnull) != NULL &&												\
					 !((((const cl_uint *)(notnull))[(rowidx) >> 5] >> ((rowidx) & 31)) & 1))

/*
 * Varlena columns of a COLUMN chunk (strom_kds.h): the column array holds the 8-byte OFFSET of
 * each row's datum from the kds head, which the loaders above bring in like any int8 value.  A
 * kernel that has assembled a row's strom_kvars from COLUMN arrays calls
 * strom_kvars_from_column(KV, kds, errcode) -- defined next to strom_kvars in each operator's
 * header over STROM_KVARLENA_LIST, the generated list of the program's text / character(n)
 * variables, empty (and the call compiled away) for every program without one -- to turn the
 * offsets into what a pg_text_t carries everywhere else, the datum's address, with the same
 * "can the device read it in place" check the row formats apply (pg_<type>_from_addr).
 */
#define STROM_KVARLENA_FIX(attno,NAME)	\
	KV.KVAR_##attno = pg_##NAME##_from_column(errcode, kds, (attno) - 1, KV.KVAR_##attno);
#define STROM_DEFINE_KVARS_FROM_COLUMN											\
	STROM_DEVICE void															\
	strom_kvars_from_column(strom_kvars &KV, const kern_data_store *kds, cl_int *errcode)	\
	{																			\
		STROM_KVARLENA_LIST(STROM_KVARLENA_FIX)									\
	}

/* the same through the caches: for rows that several work-groups of an XCD
 * read one after the other (hash roles of the hashed GROUP BY) */
#define STROM_COLUMN_REF_CACHED(NAME, values, notnull, rowidx)						\
	pg_##NAME##_make(((const pg_##NAME##_base_t *)(values))[rowidx],					\
					 (notnull) != NULL &&												\
					 !((((const cl_uint *)(notnull))[(rowidx) >> 5] >> ((rowidx) & 31)) & 1))

/*
 * row formats in the row-at-a-time kernels: locate the heap tuple ONCE per
 * row (row item -> page -> line pointer is a chain of dependent loads) and
 * take every referenced attribute from it; the per-datum accessor repeats
 * the chain for each attribute
 */
STROM_DEVICE const HeapTupleHeaderData *
strom_locate_tuple(const kern_data_store *kds, cl_int format, cl_uint rowidx)
{
	if (format == KDS_FORMAT_ROW)
		return kern_get_tuple_rs(kds, rowidx);
	if (format == KDS_FORMAT_ROW_FLAT)
		return kern_get_tuple_rsflat(kds, rowidx);
	return NULL;
}

/* needs a 'cl_int errcode' in scope, like the pg_<T>_vref() calls it replaces */
#define STROM_TUPLE_REF(NAME, kds, htup, colidx)									\
	(((htup) != NULL && (cl_uint)(colidx) < (kds)->ncols)							\
	 ? pg_##NAME##_tupref(&errcode, (kds)->colmeta, htup, colidx)					\
	 : pg_##NAME##_make(0, true))

#endif	/* STROM_COMMON_DEVICE_H */
