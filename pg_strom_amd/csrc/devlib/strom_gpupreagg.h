/*
 * strom_gpupreagg.h -- GpuPreAgg kernels (HIP, gfx950)
 *
 * Role in the reference: opencl_gpupreagg.h -- gpupreagg_preparation
 * (380-447: qual + projection of per-row partial inputs),
 * gpupreagg_set_rindex / bitonic_local / bitonic_step / bitonic_merge
 * (620-856: order rows by group key) and gpupreagg_reduction (459-608:
 * per-work-group segmented tree reduction, one partial row per
 * (work-group, group)), with the PMIN/PMAX/PSUM rules of 862-987.
 *
 * What this build does instead (every partial
 * function is an associative merge, so any reduction order that ends in
 * the same partial->final merge is a legal replacement):
 *
 *   no sort, no scratch TUPSLOT store.  A row's partial inputs stay in
 *   registers; its group id is computed directly from the key values
 *   (dense id = sum_k (key_k - min_k) * stride_k, NULL gets its own slot;
 *   min/range come from the chunk zone maps of KDS_FORMAT_COLUMN) and the
 *   inputs are folded into typed accumulators in LDS with LDS atomics
 *   (ds_add_u32 / ds_add_u64 / ds_add_f64 / ds_min/max).  Rows are read
 *   once, coalesced, 16 bytes per lane per column.
 *
 *   small group counts replicate the table NREP times inside LDS (lane ->
 *   replica) so that a wave does not serialise on one hot address; large
 *   group counts (state > LDS budget) split the dense id range over
 *   NSPLIT work-group roles, each role keeping its slice.
 *
 *   at the end a work-group folds its replicas and writes ONE slab
 *   (plain coalesced stores); gpupreagg_dense_merge then adds the slabs
 *   into the resident per-GPU table in a fixed order -- float sums are
 *   reproducible run to run, and nothing is merged when the chunk
 *   raised CpuReCheck (the reference re-does such a chunk on the CPU,
 *   gpupreagg.c:2746-2750).
 *
 * Generated code supplies STROM_KPARAM_LIST / STROM_KVAR_LIST and
 *   GPUPREAGG_KEY_LIST(X)  X(kidx, resno, NAME)
 *   GPUPREAGG_AGG_LIST(X)  X(aidx, resno, OP, NAME)   OP in NROWS PSUM PMIN PMAX
 *   gpupreagg_qual_eval / gpupreagg_key_<k> / gpupreagg_agg_<a>
 */
#ifndef STROM_GPUPREAGG_DEVICE_H
#define STROM_GPUPREAGG_DEVICE_H

#ifndef GPUPREAGG_BLOCK
#define GPUPREAGG_BLOCK		1024
#endif
#ifndef GPUPREAGG_QUADS
#define GPUPREAGG_QUADS		2
#endif
#define GPUPREAGG_TILE_ROWS	(GPUPREAGG_BLOCK * 4 * GPUPREAGG_QUADS)
#define GPUPREAGG_MAXKEYS	8

struct strom_kparams {
#define X(idx,NAME)	pg_##NAME##_t KPARAM_##idx;
	STROM_KPARAM_LIST(X)
#undef X
	int __dummy;
};
#ifndef STROM_KFIXED_LIST
#define STROM_KFIXED_LIST(X)
#endif
struct strom_kvars {
#define X(attno,colidx,NAME)	pg_##NAME##_t KVAR_##attno;
	STROM_KVAR_LIST(X)
#undef X
	/* (var N numeric SCALE) as fixed point, converted once per row: STROM_KVARS_FINISH */
#define X(attno,scale)			pg_fixed_cache_t KFIX_##attno##_##scale;
	STROM_KFIXED_LIST(X)
#undef X
	int __dummy;
};
/*
 * text / character(n) variables of a row taken from COLUMN arrays: offset -> address
 * (strom_common.h).  Called by the row-at-a-time kernels and by the streaming dense kernels
 * (dense / packed / reg1 / priv _column); the hash roles' scans and the lookup / joined variants
 * do not, and the host keeps a program with such variables away from them (gpupreagg.cpp:
 * column_streams, strom_submit_gpupreagg_joined / _lookup).
 */
#ifndef STROM_KVARLENA_LIST
#define STROM_KVARLENA_LIST(X)
#define GPUPREAGG_HAS_VARLENA_VARS	0
#else
#define GPUPREAGG_HAS_VARLENA_VARS	1
#endif
STROM_DEFINE_KVARS_FROM_COLUMN
/*
 * the row's variables are assembled: convert the numeric columns the program reads as fixed point
 * (the conversion raises nothing here -- a row the qual drops must not send the chunk back; whoever
 * USES the value sees its flag: pg_fixed_cached, strom_numeric.h)
 */
#define STROM_KFIXED_FILL_ONE(attno,scale)	\
	(KV_).KFIX_##attno##_##scale = pg_fixed_cache_fill((KV_).KVAR_##attno, scale);
#define STROM_KVARS_FINISH(KV)				\
	do {									\
		strom_kvars &KV_ = (KV);			\
		KV_.__dummy = 0;					\
		STROM_KFIXED_LIST(STROM_KFIXED_FILL_ONE)	\
	} while (0)

STROM_DEVICE pg_bool_t
gpupreagg_qual_eval(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV);
#define X(kidx,resno,NAME)													\
	STROM_DEVICE pg_##NAME##_t												\
	gpupreagg_key_##kidx(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV);
GPUPREAGG_KEY_LIST(X)
#undef X
#define X(aidx,resno,OP,NAME)												\
	STROM_DEVICE pg_##NAME##_t												\
	gpupreagg_agg_##aidx(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV);
GPUPREAGG_AGG_LIST(X)
#undef X

/* control block written by the host (gpupreagg.cpp mirrors this struct) */
struct gpupreagg_dense_ctl {
	cl_uint		ngroups;			/* dense ids in the whole domain */
	cl_uint		nsplits;			/* work-group roles over the id range */
	cl_uint		groups_per_split;
	cl_uint		nrep;				/* LDS replicas, power of two */
	cl_uint		nslabs;				/* == gridDim.x */
	cl_uint		nkeys;
	cl_ulong	slab_bytes;
	cl_long		key_min[GPUPREAGG_MAXKEYS];
	cl_uint		key_range[GPUPREAGG_MAXKEYS];	/* max-min+1; NULL slot == key_range */
	cl_uint		key_stride[GPUPREAGG_MAXKEYS];
	/* compaction (strom_gpupreagg_compact): dense id -> slot of the ids that
	 * actually occur, ~0 = absent; 0 = ids are used as they are */
	cl_ulong	remap;				/* device address of cl_uint[dense_ngroups] */
	cl_uint		dense_ngroups;		/* product of (key_range + 1) */
	cl_uint		merge_ws;			/* stripes of the slab merge (power of two <= 64), 0 = derive */
};

/*
 * dense id -> table slot; false when the combination is not in the table.
 * A small map is staged in LDS by gpupreagg_remap_init(): the lookup sits on
 * every row's dependent path (keys -> dense id -> slot -> accumulator
 * address) and an L1/L2 round trip there costs more than the fold.
 */
#define GPUPREAGG_REMAP_LDS		1024
__shared__ cl_uint	gpupreagg_remap_staged[GPUPREAGG_REMAP_LDS];

STROM_DEVICE void
gpupreagg_remap_init(const gpupreagg_dense_ctl *ctl)
{
	if (ctl->remap != 0 && ctl->dense_ngroups <= GPUPREAGG_REMAP_LDS)
	{
		for (cl_uint i = threadIdx.x; i < ctl->dense_ngroups; i += blockDim.x)
			gpupreagg_remap_staged[i] = ((const cl_uint *)ctl->remap)[i];
	}
	__syncthreads();
}

STROM_DEVICE bool
gpupreagg_remap_gid(const gpupreagg_dense_ctl *ctl, cl_uint &gid)
{
	if (ctl->remap != 0)
	{
		gid = (ctl->dense_ngroups <= GPUPREAGG_REMAP_LDS
			   ? gpupreagg_remap_staged[gid]
			   : ((const cl_uint *)ctl->remap)[gid]);
		if (gid == 0xffffffffu)
			return false;
	}
	return true;
}

/* ---- accumulator encodings ------------------------------------------ *
 * NROWS   : u32 in LDS / slab, i64 in the resident table
 * others  : 8 bytes.  int-like values as i64; float8 PSUM as f64; float8
 *           PMIN/PMAX as an order-preserving u64 (NaN above +inf, the
 *           PostgreSQL float ordering)
 */
STROM_DEVICE cl_ulong
gpupreagg_f64_ordered(cl_double v)
{
	cl_ulong bits = __builtin_isnan(v) ? 0x7ff8000000000000UL : (cl_ulong)__double_as_longlong(v);
	return (bits & 0x8000000000000000UL) ? ~bits : (bits | 0x8000000000000000UL);
}
STROM_DEVICE cl_double
gpupreagg_f64_unordered(cl_ulong key)
{
	cl_ulong bits = (key & 0x8000000000000000UL) ? (key & 0x7fffffffffffffffUL) : ~key;
	return __longlong_as_double((long long)bits);
}

#define GPUPREAGG_OP_NROWS	0
#define GPUPREAGG_OP_PSUM	1
#define GPUPREAGG_OP_PMIN	2
#define GPUPREAGG_OP_PMAX	3

template <typename T> struct gpupreagg_is_float { static const bool value = false; };
template <> struct gpupreagg_is_float<cl_double> { static const bool value = true; };
template <> struct gpupreagg_is_float<cl_float> { static const bool value = true; };
/*
 * NUMERIC partials without a scale -- "(psum EXPR)" over a numeric -- are
 * accumulated in the reference's own 64-bit form (6-bit exponent, sign, 57-bit
 * mantissa) with pgfn_numeric_add / strom_numeric_cmp, exact or CpuReCheck, as
 * GPUPREAGG_AGGCALC_PSUM_NUMERIC / _PMINMAX_NUMERIC do (opencl_gpupreagg.h:
 * 882-900, 965-987).  There is no LDS atomic for that: a compare-and-swap
 * loop.  (The fixed-point form "(psum EXPR SCALE)" is an int8 accumulator and
 * takes the integer atomics.)  pg_numeric_t is the only type whose base is
 * cl_ulong.
 */
template <typename T> struct gpupreagg_is_numeric { static const bool value = false; };
#ifdef STROM_NUMERIC_DEVICE_H
template <> struct gpupreagg_is_numeric<cl_ulong> { static const bool value = true; };
/* "no value yet" in a numeric PMIN / PMAX accumulator: minus zero, which no
 * arithmetic produces (strom_numeric_pack returns plain 0 for a zero mantissa) */
#define GPUPREAGG_NUMERIC_EMPTY		PG_NUMERIC_SIGN_MASK

template <int OP>
STROM_DEVICE cl_ulong
gpupreagg_numeric_combine(cl_ulong acc, cl_ulong v, cl_int *errcode)
{
	pg_numeric_t	a, b;
	a.isnull = b.isnull = false;
	a.value = acc;
	b.value = v;
	if (OP == GPUPREAGG_OP_PSUM)
	{
		pg_numeric_t r = pgfn_numeric_add(errcode, a, b);
		return r.isnull ? acc : r.value;
	}
	if (acc == GPUPREAGG_NUMERIC_EMPTY)
		return v;
	if (v == GPUPREAGG_NUMERIC_EMPTY)
		return acc;
	int		c = strom_numeric_cmp(b, a);
	return ((OP == GPUPREAGG_OP_PMIN) ? (c < 0) : (c > 0)) ? v : acc;
}
#endif

STROM_DEVICE cl_uint gpupreagg_align16(cl_uint v) { return (v + 15u) & ~15u; }

/*
 * Per-group flags: bit 0 = a row of this group passed the qual ("seen"),
 * bit 1+a = aggregate a received a non-NULL input.  One word per (group,
 * replica) so that a row updates all its flags with ONE LDS operation, and
 * after the first row of a group only a read remains.
 */
#if GPUPREAGG_NAGGS <= 7
typedef cl_uchar	gpupreagg_flags_t;
#elif GPUPREAGG_NAGGS <= 15
typedef cl_ushort	gpupreagg_flags_t;
#else
typedef cl_uint		gpupreagg_flags_t;
#endif
#define GPUPREAGG_FLAG_SEEN		1u

/*
 * LDS / slab image for G groups and REP replicas:
 *   section 0        flags[G*REP]
 *   section 1+a      values of aggregate a: u32[G*REP] (NROWS) or 8 bytes[G*REP]
 *   section 1+NAGGS  total size
 * 32-bit offsets: the image lives in LDS (<= 160 KB) or in a slab of the
 * same shape.
 */
STROM_DEVICE cl_uint
gpupreagg_image_offset(int sec, cl_uint G, cl_uint REP)
{
	cl_uint	off = 0;
	int		cur = 0;

	if (sec == cur) return off;
	off += gpupreagg_align16((cl_uint)sizeof(gpupreagg_flags_t) * G * REP); cur++;
#define X(aidx,resno,OP,NAME)																\
	if (sec == cur) return off;																\
	off += gpupreagg_align16((GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS ? 4u : 8u) * G * REP);	\
	cur++;
	GPUPREAGG_AGG_LIST(X)
#undef X
	return off;
}

/*
 * resident table for N groups, 256-byte aligned sections:
 *   section 0          flags as u32[N]
 *   section 1+a        8-byte values[N] (NROWS widened to i64)
 *   section 1+NAGGS+j  the j-th INTEGER sum's high word, i64[N]: such a sum is
 *                      128 bits wide in the table (low word in its section 1+a,
 *                      two's complement), so a total over any number of chunks
 *                      cannot wrap ("integer sums never wrap", below)
 */
STROM_DEVICE size_t
gpupreagg_table_offset(int sec, cl_uint N)
{
	size_t	off = 0;
	size_t	flags = STROM_TYPEALIGN(256, sizeof(cl_uint) * (size_t)N);
	size_t	vals = STROM_TYPEALIGN(256, 8 * (size_t)N);

	if (sec == 0) return off;
	off += flags;
	return off + vals * (size_t)(sec - 1);
}

/*
 * fold one row's partial value into the LDS accumulator at 'slot'; returns
 * the flag bit to raise (0 when the input was NULL).  No branch: a NULL input
 * folds the operation's identity (0, -0.0 which leaves even the sign of a
 * zero sum alone, the min/max sentinels), so the atomic is issued for every
 * row that reaches this point -- a per-row "if" costs an exec-mask save and a
 * taken skip-branch in the common case, the extra lanes of an atomic nothing.
 */
template <int OP, int AIDX, typename PGT>
STROM_DEVICE cl_uint
gpupreagg_lds_accum(char *lds, cl_uint vals_off, cl_uint slot, PGT v, cl_int *chunk_status = NULL)
{
	typedef decltype(v.value) base_t;
	bool	has = !v.isnull;

#ifdef STROM_NUMERIC_DEVICE_H
	if (OP != GPUPREAGG_OP_NROWS && gpupreagg_is_numeric<base_t>::value)
	{
		/* one retry loop, no inner wait: a lane that lost the race goes round
		 * again with what the winner stored */
		cl_ulong   *addr = (cl_ulong *)(lds + vals_off) + slot;
		if (has)
		{
			cl_ulong	cur = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			for (;;)
			{
				cl_int		e2 = StromError_Success;
				cl_ulong	next = gpupreagg_numeric_combine<OP>(cur, (cl_ulong)v.value, &e2);
				if (e2 != StromError_Success)
				{
					/* the sum left the 64-bit form: the chunk goes back to the CPU */
					if (chunk_status)
						STROM_SET_ERROR(chunk_status, e2);
					break;
				}
				if (next == cur ||
					__hip_atomic_compare_exchange_strong(addr, &cur, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
														 __HIP_MEMORY_SCOPE_WORKGROUP))
					break;
			}
		}
		return has ? (2u << AIDX) : 0u;
	}
#endif
	if (OP == GPUPREAGG_OP_NROWS)
	{
		__hip_atomic_fetch_add((cl_uint *)(lds + vals_off) + slot, has ? (cl_uint)v.value : 0u,
							   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		return 0;
	}
	if (gpupreagg_is_float<base_t>::value)
	{
		if (OP == GPUPREAGG_OP_PSUM)
			__hip_atomic_fetch_add((cl_double *)(lds + vals_off) + slot,
								   has ? (cl_double)v.value : -0.0,
								   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else if (OP == GPUPREAGG_OP_PMIN)
			__hip_atomic_fetch_min((cl_ulong *)(lds + vals_off) + slot,
								   has ? gpupreagg_f64_ordered((cl_double)v.value) : 0xffffffffffffffffUL,
								   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else
			__hip_atomic_fetch_max((cl_ulong *)(lds + vals_off) + slot,
								   has ? gpupreagg_f64_ordered((cl_double)v.value) : 0UL,
								   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	else
	{
		if (OP == GPUPREAGG_OP_PSUM)
		{
#if defined(GPUPREAGG_CHECKED) && GPUPREAGG_CHECKED
			/* the returning form: the add is checked against what it was added to
			 * (CHECK_OVERFLOW_INT of the reference's PSUM template).  The atomics of a
			 * slot are one sequence of additions; each link of it is checked here. */
			cl_long		x = (has ? (cl_long)v.value : 0L), sum;
			cl_long		old = __hip_atomic_fetch_add((cl_long *)(lds + vals_off) + slot, x,
													 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (chunk_status)
				STROM_SET_RECHECK_IF(chunk_status, __builtin_add_overflow(old, x, &sum));
#else
			__hip_atomic_fetch_add((cl_long *)(lds + vals_off) + slot, has ? (cl_long)v.value : 0L,
								   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
		}
		else if (OP == GPUPREAGG_OP_PMIN)
			__hip_atomic_fetch_min((cl_long *)(lds + vals_off) + slot,
								   has ? (cl_long)v.value : 0x7fffffffffffffffL,
								   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else
			__hip_atomic_fetch_max((cl_long *)(lds + vals_off) + slot,
								   has ? (cl_long)v.value : (-0x7fffffffffffffffL - 1),
								   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	return has ? (2u << AIDX) : 0u;
}

/* identity element of an 8-byte accumulator */
template <int OP, typename BASE>
STROM_DEVICE cl_ulong
gpupreagg_identity(void)
{
#ifdef STROM_NUMERIC_DEVICE_H
	if (gpupreagg_is_numeric<BASE>::value)
		return (OP == GPUPREAGG_OP_PSUM ? 0UL : GPUPREAGG_NUMERIC_EMPTY);
#endif
	if (OP == GPUPREAGG_OP_PMIN)
		return gpupreagg_is_float<BASE>::value ? 0xffffffffffffffffUL : 0x7fffffffffffffffUL;
	if (OP == GPUPREAGG_OP_PMAX)
		return gpupreagg_is_float<BASE>::value ? 0UL : 0x8000000000000000UL;
	return 0UL;		/* PSUM: +0 / 0.0 */
}

/* merge two 8-byte accumulators (b into a) */
template <int OP, typename BASE>
STROM_DEVICE cl_ulong
gpupreagg_merge8(cl_ulong a, cl_ulong b)
{
#ifdef STROM_NUMERIC_DEVICE_H
	if (gpupreagg_is_numeric<BASE>::value)
	{
		/* (callers that must see a lost sum use gpupreagg_merge8e) */
		cl_int	ignored = StromError_Success;
		return gpupreagg_numeric_combine<OP == GPUPREAGG_OP_NROWS ? GPUPREAGG_OP_PSUM : OP>(a, b, &ignored);
	}
#endif
	if (OP == GPUPREAGG_OP_PSUM)
	{
		if (gpupreagg_is_float<BASE>::value)
			return (cl_ulong)__double_as_longlong(__longlong_as_double((long long)a) +
												  __longlong_as_double((long long)b));
		return a + b;
	}
	if (gpupreagg_is_float<BASE>::value)
		return (OP == GPUPREAGG_OP_PMIN) ? (a < b ? a : b) : (a > b ? a : b);
	return (OP == GPUPREAGG_OP_PMIN) ? ((cl_long)a < (cl_long)b ? a : b)
									 : ((cl_long)a > (cl_long)b ? a : b);
}

/* the same, reporting a numeric sum that left the 64-bit form */
template <int OP, typename BASE>
STROM_DEVICE cl_ulong
gpupreagg_merge8e(cl_ulong a, cl_ulong b, cl_int *errcode)
{
#ifdef STROM_NUMERIC_DEVICE_H
	if (gpupreagg_is_numeric<BASE>::value)
		return gpupreagg_numeric_combine<OP == GPUPREAGG_OP_NROWS ? GPUPREAGG_OP_PSUM : OP>(a, b, errcode);
#endif
#if defined(GPUPREAGG_CHECKED) && GPUPREAGG_CHECKED
	if (OP == GPUPREAGG_OP_PSUM && !gpupreagg_is_float<BASE>::value)
	{
		cl_long	sum;
		if (__builtin_add_overflow((cl_long)a, (cl_long)b, &sum))
			STROM_SET_ERROR(errcode, StromError_CpuReCheck);
		return (cl_ulong)sum;
	}
#endif
	return gpupreagg_merge8<OP, BASE>(a, b);
}

/* ---------------------------------------------------------------------- *
 * integer sums never wrap silently
 *
 * The reference checks every accumulate of its reduction (CHECK_OVERFLOW_INT,
 * opencl_gpupreagg.h:142-143, used by GPUPREAGG_AGGCALC_PSUM_TEMPLATE 933-948)
 * and sends the chunk back to the CPU when an int8 partial sum leaves its
 * type.  A check per LDS atomic needs the returning form of every ds_add_u64;
 * instead the common case is PROVEN not to wrap and only the rest is checked:
 *
 *   Every summed input of the request has magnitude below 2^B -- B from the
 *   type (a sum over int2 / int4 values cast to int8: GPUPREAGG_SUMBITS_<a>, by
 *   the code generator), from the zone map (packed accumulators), or measured:
 *   the fold ORs the magnitudes of the inputs it adds and leaves the bit count
 *   in kern_gpupreagg (KERN_GPUPREAGG_SUM_MAGBITS).  N rows then give
 *   |any partial sum| <= N * 2^B.
 *
 *   tier 1  rows of the CHUNK x 2^B < 2^63: nothing in the chunk -- LDS
 *           accumulators, replicas, slabs, the chunk's total per group -- can
 *           have wrapped.  gpupreagg_dense_merge takes the slabs; nothing else runs.
 *   tier 2  only rows of ONE WORK-GROUP x 2^B < 2^63 (KERN_GPUPREAGG_WG_ROWS, from
 *           the launch geometry): every slab is exact, their sum may not be.
 *           gpupreagg_dense_merge_check adds the slabs up in 128 bits first and
 *           answers CpuReCheck when a group's chunk total leaves int8 -- the
 *           reference's answer for such a chunk -- before the table takes any of it.
 *   tier 3  not even that: the merge leaves the table alone and answers
 *           StromError_SumRangeUnproven, an internal status that never reaches
 *           the caller: the host folds the chunk again with the program built with
 *           GPUPREAGG_CHECKED, whose LDS adds return the old value and are checked
 *           one by one, replicas likewise, then tier 2's slab check.
 *
 *   The resident table itself keeps such a sum 128 bits wide (a "hi" section
 *   per integer sum): the total over any number of chunks cannot wrap, and the
 *   fetch hands a total beyond int8 out as several partial rows.
 * ---------------------------------------------------------------------- */
/* (StromError_SumRangeUnproven: strom_kds.h) */

template <int OP, typename BASE> struct gpupreagg_is_intsum {
	static const bool value = (OP == GPUPREAGG_OP_PSUM && !gpupreagg_is_float<BASE>::value &&
							   !gpupreagg_is_numeric<BASE>::value);
};

/* v >= 0: v; v < 0: -v - 1 -- below 2^B means |v| <= 2^B, and an OR of magnitudes
 * has the bit count of their maximum */
STROM_DEVICE cl_ulong
gpupreagg_sum_magnitude(cl_long v)
{
	return (cl_ulong)(v ^ (v >> 63));
}

/*
 * what the range proof reads, in the 8 padding bytes of kern_gpupreagg (written by
 * the host per request): the bit count B of the largest input magnitude -- preset with
 * what is known statically, raised by the folds (atomic max) -- and the rows ONE
 * work-group folds at most; sortbuf_len -- the reference's sort buffer length, no use
 * here -- carries the rows of the whole request
 */
#define KERN_GPUPREAGG_SUM_MAGBITS(kgp)		((cl_uint *)((kgp)->__padding))
#define KERN_GPUPREAGG_WG_ROWS(kgp)			(*(const cl_uint *)((kgp)->__padding + 4) & 0x7fffffffu)
/* top bit of that word: the host has bounded the sums of PLAIN columns (GPUPREAGG_SUMBITS_<a> 65) and
 * of expressions over decimal columns (66: GPUPREAGG_SUMBOUND_<a>, a formula over the columns' zone
 * maps from the code generator) by the chunk's zone maps -- the fold need not measure them */
#define KERN_GPUPREAGG_ZONE_BOUNDED(kgp)	((*(const cl_uint *)((kgp)->__padding + 4) >> 31) != 0)
/*
 * per-launch facts the row functions take as one word:
 *   ROWFLAG_ZONE_BOUNDED   see above
 *   ROWFLAG_ALL_NOTNULL    no column of the chunk has a NULL bitmap: every nrows() whose arguments
 *                          only say "column X is not NULL" (GPUPREAGG_COUNTALL_<a>) counts exactly
 *                          what count(*) counts -- ONE of them is accumulated (GPUPREAGG_COUNTALL_FIRST),
 *                          gpupreagg_store_slab copies it to the others (Q1: four counters, one atomic)
 */
#define ROWFLAG_ZONE_BOUNDED	1u
#define ROWFLAG_ALL_NOTNULL		2u
#ifndef GPUPREAGG_COUNTALL_FIRST
#define GPUPREAGG_COUNTALL_FIRST	(-1)
#endif
#define GPUPREAGG_MEASURE_SUM(aidx, rowflags)	\
	(GPUPREAGG_SUMBITS_##aidx == 64 ||		\
	 ((GPUPREAGG_SUMBITS_##aidx == 65 || GPUPREAGG_SUMBITS_##aidx == 66) && !((rowflags) & ROWFLAG_ZONE_BOUNDED)))
#define GPUPREAGG_COUNT_IS_ALIASED(aidx, rowflags)	\
	(GPUPREAGG_COUNTALL_##aidx && (aidx) != GPUPREAGG_COUNTALL_FIRST && ((rowflags) & ROWFLAG_ALL_NOTNULL))
#define KERN_GPUPREAGG_FOLD_NROWS(kgp)		((cl_uint)(kgp)->sortbuf_len)

STROM_DEVICE void
gpupreagg_writeback_summag(kern_gpupreagg *kgpreagg, cl_ulong summag)
{
#pragma unroll
	for (int m = 32; m > 0; m >>= 1)
	{
		cl_uint lo = __shfl_xor((cl_uint)summag, m, STROM_WAVE);
		cl_uint hi = __shfl_xor((cl_uint)(summag >> 32), m, STROM_WAVE);
		summag |= ((cl_ulong)hi << 32) | lo;
	}
	if (strom_lane_id() == 0 && summag != 0)
		__hip_atomic_fetch_max(KERN_GPUPREAGG_SUM_MAGBITS(kgpreagg), (cl_uint)(64 - __builtin_clzl(summag)),
							   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* the position of aggregate aidx among the integer sums (its high-word section in the table) */
STROM_DEVICE constexpr int
gpupreagg_intsum_index(int aidx)
{
	int		n = 0;
#define X(a,resno,OP,NAME)																\
	n += ((a) < aidx && gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value ? 1 : 0);
	GPUPREAGG_AGG_LIST(X)
#undef X
	return n;
}

/* nrows inputs below 2^magbits in magnitude add up to at most this (saturating at 2^63) */
STROM_DEVICE cl_ulong
gpupreagg_sum_bound(cl_uint nrows, cl_uint magbits)
{
	if (nrows == 0)
		return 0;
	if (magbits >= 63 || ((cl_ulong)nrows >> (63 - magbits)) != 0)
		return (1UL << 63);
	return (cl_ulong)nrows << magbits;
}

/* can a sum of nrows such inputs leave int8? */
STROM_DEVICE bool
gpupreagg_sum_range_proven(cl_uint nrows, cl_uint magbits)
{
	return gpupreagg_sum_bound(nrows, magbits) < (1UL << 63);
}

/* LDS section offsets, computed once per kernel */
struct gpupreagg_lds_layout {
	cl_uint		vals_off[GPUPREAGG_NAGGS + 1];
	cl_uint		total;
};

/*
 * A wave-uniform value the row loops add to a per-lane LDS index (a section's offset): kept in a
 * VECTOR register on purpose.  As a scalar it competes with the column pointers, the key domain
 * and the parameters for ~100 SGPRs, loses, and is spilled into a lane of a VGPR -- every row
 * then pays a v_readlane_b32 plus the wait states before the add that uses it, per aggregate
 * (Q1's kernels: 7 of them per row with the lane-private accumulators, 33 with the LDS atomics).
 * The asm's output constraint is all the compiler needs to treat the value as per-lane.
 */
STROM_DEVICE cl_uint
strom_keep_in_vgpr(cl_uint v)
{
#if defined(GPUPREAGG_LDS_OFFSETS_SCALAR) && GPUPREAGG_LDS_OFFSETS_SCALAR
	return v;
#else
	cl_uint		r;
	asm("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
	return r;
#endif
}

STROM_DEVICE void
gpupreagg_lds_layout_init(gpupreagg_lds_layout &L, cl_uint G, cl_uint NREP)
{
#define X(aidx,resno,OP,NAME)	L.vals_off[aidx] = strom_keep_in_vgpr(gpupreagg_image_offset(1 + aidx, G, NREP));
	GPUPREAGG_AGG_LIST(X)
#undef X
	L.total = gpupreagg_image_offset(1 + GPUPREAGG_NAGGS, G, NREP);
}

/* ---------------------------------------------------------------------- *
 * one row: qual, group id, fold
 * ---------------------------------------------------------------------- */
STROM_DEVICE void
gpupreagg_dense_row(char *lds, const gpupreagg_dense_ctl *ctl, const gpupreagg_lds_layout &L,
					const strom_kparams &KP, const strom_kvars &KV,
					cl_uint gid_lo, cl_uint G, cl_uint NREP, cl_uint rep,
					cl_int param_error, cl_int *chunk_status, cl_ulong &summag, cl_uint rowflags,
					bool qual_done = false)
{
	cl_int		errcode = param_error;
	cl_uint		gid = 0;
	bool		out_of_domain = false;

	if (!qual_done)				/* (the caller may have seen the qual pass, without an error) */
	{
		pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KV);
		if (errcode == StromError_Success && !EVAL(rc))
			return;
	}
	/* group id: dense, NULL key in its own slot */
#define X(kidx,resno,NAME)															\
	{																				\
		pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);					\
		cl_long		off64 = (cl_long)kv.value - ctl->key_min[kidx];					\
		cl_uint		range = ctl->key_range[kidx];									\
		cl_uint		off = (kv.isnull ? range : (cl_uint)off64);						\
		if (!kv.isnull && (off64 < 0 || off64 >= (cl_long)range))					\
			out_of_domain = true;													\
		gid += (kidx == 0 ? off : off * ctl->key_stride[kidx]);	/* (stride 0 is 1) */											\
	}
	GPUPREAGG_KEY_LIST(X)
#undef X
	if (!out_of_domain && !gpupreagg_remap_gid(ctl, gid))
		out_of_domain = true;
	/*
	 * a row of another role's slice of the slot range: that role evaluates
	 * (and error-checks) its partial inputs -- leaving here halves the
	 * per-row work of a split table, which is bound by instruction issue
	 */
	if (errcode == StromError_Success && !out_of_domain && gid - gid_lo >= G)
		return;
	/* partial inputs (evaluated for every surviving row so that arithmetic
	 * errors are seen before anything is folded) */
#define X(aidx,resno,OP,NAME)														\
	pg_##NAME##_t av_##aidx = gpupreagg_agg_##aidx(&errcode, KP, KV);
	GPUPREAGG_AGG_LIST(X)
#undef X
	if (errcode != StromError_Success)
	{
		/* CpuReCheck (or worse) anywhere in the row: the chunk goes back */
		STROM_SET_ERROR(chunk_status, errcode);
		return;
	}
	if (out_of_domain)
	{
		STROM_SET_ERROR(chunk_status, StromError_DataStoreOutOfRange);
		return;
	}
	cl_uint		lgid = gid - gid_lo;
	if (lgid >= G)				/* another role's slice of the id range */
		return;
	cl_uint		slot = lgid * NREP + rep;
	cl_uint		need = GPUPREAGG_FLAG_SEEN;

#define X(aidx,resno,OP,NAME)														\
	/* an integer sum without a static bound: measure (see "integer sums never wrap") */	\
	if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value && GPUPREAGG_MEASURE_SUM(aidx, rowflags))	\
		summag |= (av_##aidx.isnull ? 0UL : gpupreagg_sum_magnitude((cl_long)av_##aidx.value));	\
	if (!(GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS && GPUPREAGG_COUNT_IS_ALIASED(aidx, rowflags)))	\
		need |= gpupreagg_lds_accum<GPUPREAGG_OP_##OP, aidx>(lds, L.vals_off[aidx], slot, av_##aidx, chunk_status);
	GPUPREAGG_AGG_LIST(X)
#undef X
	/* flags: read, and only touch the word when something is missing */
	gpupreagg_flags_t *flags = (gpupreagg_flags_t *)lds;
	if ((flags[slot] & need) != need)
	{
		cl_uint *word = (cl_uint *)(lds + ((slot * (cl_uint)sizeof(gpupreagg_flags_t)) & ~3u));
		cl_uint	 shift = ((slot * (cl_uint)sizeof(gpupreagg_flags_t)) & 3u) * 8u;
		__hip_atomic_fetch_or(word, need << shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

/* initialise the LDS image */
STROM_DEVICE void
gpupreagg_lds_init(char *lds, const gpupreagg_lds_layout &L, cl_uint G, cl_uint NREP)
{
	/* (blockDim.x: the hashed partition plan's fold runs 1024 threads, everything else GPUPREAGG_BLOCK) */
	for (cl_uint i = threadIdx.x * 16; i < L.total; i += blockDim.x * 16)
		*(uint4 *)(lds + i) = make_uint4(0, 0, 0, 0);
	__syncthreads();
#define X(aidx,resno,OP,NAME)															\
	if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN || GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMAX)	\
	{																					\
		cl_ulong   *vals = (cl_ulong *)(lds + L.vals_off[aidx]);						\
		cl_ulong	ident = gpupreagg_identity<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>();	\
		for (cl_uint i = threadIdx.x; i < G * NREP; i += blockDim.x)					\
			vals[i] = ident;															\
	}
	GPUPREAGG_AGG_LIST(X)
#undef X
	__syncthreads();
}

/* fold replicas and store the work-group's slab (REP = 1 image) */
STROM_DEVICE void
gpupreagg_store_slab(const char *lds, const gpupreagg_lds_layout &L, char *slab,
					 cl_uint G, cl_uint NREP, cl_int *chunk_status, cl_uint rowflags = 0)
{
	const gpupreagg_flags_t *lflags = (const gpupreagg_flags_t *)lds;

	__syncthreads();
	for (cl_uint g = threadIdx.x; g < G; g += GPUPREAGG_BLOCK)
	{
		cl_uint f = 0;
		for (cl_uint r = 0; r < NREP; r++)
			f |= lflags[g * NREP + r];
		((gpupreagg_flags_t *)slab)[g] = (gpupreagg_flags_t)f;
	}
#define X(aidx,resno,OP,NAME)																\
	{																						\
		/* (an aliased count was never added to: it is the first such counter's, see ROWFLAG_ALL_NOTNULL) */	\
		const char *lvals = lds + L.vals_off[(GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS &&		\
											  GPUPREAGG_COUNT_IS_ALIASED(aidx, rowflags))		\
											 ? GPUPREAGG_COUNTALL_FIRST : aidx];				\
		char	   *svals = slab + gpupreagg_image_offset(1 + aidx, G, 1);					\
		if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)											\
		{																					\
			for (cl_uint g = threadIdx.x; g < G; g += GPUPREAGG_BLOCK)						\
			{																				\
				cl_uint sum = 0;															\
				for (cl_uint r = 0; r < NREP; r++)											\
					sum += ((const cl_uint *)lvals)[g * NREP + r];							\
				((cl_uint *)svals)[g] = sum;												\
			}																				\
		}																					\
		else																				\
		{																					\
			for (cl_uint g = threadIdx.x; g < G; g += GPUPREAGG_BLOCK)						\
			{																				\
				cl_ulong acc = ((const cl_ulong *)lvals)[g * NREP];							\
				for (cl_uint r = 1; r < NREP; r++)											\
					acc = gpupreagg_merge8e<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>			\
						(acc, ((const cl_ulong *)lvals)[g * NREP + r], chunk_status);		\
				((cl_ulong *)svals)[g] = acc;												\
			}																				\
		}																					\
	}
	GPUPREAGG_AGG_LIST(X)
#undef X
}

/*
 * chunk status.  Device code only raises positive codes, so "worst wins"
 * is a max: significant (>=100) > CpuReCheck (2) > Success (0).
 */
STROM_DEVICE void
gpupreagg_writeback_status(cl_int *status, cl_int chunk_status)
{
	cl_int worst = strom_wave_max_i32(chunk_status);
	if (strom_lane_id() == 0 && worst != StromError_Success)
		atomicMax(status, worst);
}

/* ---------------------------------------------------------------------- *
 * packed accumulators (gpupreagg_packed_column)
 *
 * The standard LDS image spends 1 (flags) + 4 (nrows) + 8 per other aggregate
 * bytes per group; BASELINE configs[3] -- GROUP BY int4 with 1e4 groups,
 * COUNT / SUM(int4) / AVG(float8) -- needs 21 B x 1e4 = 210 KB and therefore
 * two id-range roles that EACH visit every row (measured 400 us per 1e8 rows
 * against 258 us for a table that fits one work-group's LDS).  Integer
 * aggregates rarely need their full width inside ONE work-group and ONE
 * chunk: the zone map bounds a summed column (max - min < 2^vbits) and the
 * launch geometry bounds the rows a work-group folds (< 2^cbits), so
 * count(*) and the integer sums fit bit fields of ONE 64-bit word
 *
 *     [ count : cbits ][ sum_k (x - min) : cbits + vbits_k ] ...
 *
 * updated by ONE ds_add_u64 per row; a float8 sum keeps a word of its own and
 * the flags are implied (seen = count > 0; "has a value" = seen, because this
 * path is taken only when no input column has a NULL in the chunk).  C4 is
 * then 16 B per group -- 160 KB, one role, every row visited once.  The
 * fields are unpacked (sum = field + count x min) when the work-group writes
 * its slab, which has the standard shape: gpupreagg_dense_merge and the
 * resident table do not know the difference.
 *
 * The host takes this path per launch (gpupreagg.cpp: packed_plan) when the
 * generated code says every aggregate is count(*), or psum of a plain column
 * (GPUPREAGG_PACK_LIST: kind 1 / 2 integer / 3 float8), the chunk's zone maps
 * and NULL-free columns allow it, the fields fit 64 bits and the packed image
 * needs fewer roles than the standard one.
 * ---------------------------------------------------------------------- */
#define GPUPREAGG_PACK_MAXAGGS	32
struct gpupreagg_pack_ctl {
	cl_uint		count_shift;					/* count field: bits count_shift .. 63 */
	cl_uint		nwords;							/* 1 (the packed word) + float8 sums */
	cl_uint		spill_at;						/* 0, or: a group whose count field reaches this moves to the slab */
	cl_uint		count_limit;					/* spill_at != 0: the count field's largest value */
	cl_uint		shift[GPUPREAGG_PACK_MAXAGGS];	/* kind 2: position of the field in word 0 */
	cl_uint		word[GPUPREAGG_PACK_MAXAGGS];	/* kind 3: the aggregate's own word */
	cl_ulong	mask[GPUPREAGG_PACK_MAXAGGS];	/* kind 2: field mask (after the shift) */
	cl_ulong	vmax[GPUPREAGG_PACK_MAXAGGS];	/* kind 2: max - min of the column (zone map) */
	cl_long		bias[GPUPREAGG_PACK_MAXAGGS];	/* kind 2: min of the column */
};

#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
STROM_DEVICE cl_uint
gpupreagg_pack_word_offset(cl_uint w, cl_uint G)
{
	return w * gpupreagg_align16(8u * G);
}

/*
 * a group's packed word V (the value right after the add that brought its count to spill_at)
 * leaves LDS: V is subtracted from the word -- its fields only grow, so every field of the word
 * is at least V's and nothing borrows -- and its count and sums are added to the work-group's
 * own slab, which the kernel zeroed at its start (atomics: another thread may spill the same
 * group a quarter field later).  Exactly one thread sees a count BECOME spill_at.
 */
STROM_DEVICE void
gpupreagg_packed_spill(char *lds, const gpupreagg_pack_ctl *pk, char *slab, cl_uint slot, cl_ulong V, cl_uint G)
{
	__hip_atomic_fetch_add((cl_ulong *)lds + slot, (cl_ulong)0 - V, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	cl_ulong	n = V >> pk->count_shift;
#define X(aidx,kind,attno)																\
	{																					\
		char   *svals = slab + gpupreagg_image_offset(1 + aidx, G, 1);					\
		if (kind == 1)																	\
			__hip_atomic_fetch_add((cl_uint *)svals + slot, (cl_uint)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);	\
		else if (kind == 2)																\
			__hip_atomic_fetch_add((cl_ulong *)svals + slot,							\
								   ((V >> pk->shift[aidx]) & pk->mask[aidx]) + n * (cl_ulong)pk->bias[aidx],	\
								   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);			\
	}
	GPUPREAGG_PACK_LIST(X)
#undef X
	/* "rows of this group were seen": all the final store needs when no row follows */
	__hip_atomic_store((gpupreagg_flags_t *)slab + slot, (gpupreagg_flags_t)GPUPREAGG_FLAG_SEEN,
					   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

STROM_DEVICE void
gpupreagg_packed_row(char *lds, const gpupreagg_dense_ctl *ctl, const gpupreagg_pack_ctl *pk,
					 const strom_kparams &KP, const strom_kvars &KV,
					 cl_uint gid_lo, cl_uint G, cl_int param_error, cl_int *chunk_status, char *slab,
					 bool qual_done = false)
{
	cl_int		errcode = param_error;
	cl_uint		gid = 0;
	bool		out_of_domain = false;

	if (!qual_done)
	{
		pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KV);
		if (errcode == StromError_Success && !EVAL(rc))
			return;
	}
#define X(kidx,resno,NAME)															\
	{																				\
		pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);					\
		cl_long		off64 = (cl_long)kv.value - ctl->key_min[kidx];					\
		cl_uint		range = ctl->key_range[kidx];									\
		cl_uint		off = (kv.isnull ? range : (cl_uint)off64);						\
		if (!kv.isnull && (off64 < 0 || off64 >= (cl_long)range))					\
			out_of_domain = true;													\
		gid += (kidx == 0 ? off : off * ctl->key_stride[kidx]);	/* (stride 0 is 1) */											\
	}
	GPUPREAGG_KEY_LIST(X)
#undef X
	if (errcode == StromError_Success && !out_of_domain && gid - gid_lo >= G)
		return;							/* another role's slice of the id range */
	if (errcode != StromError_Success)
	{
		STROM_SET_ERROR(chunk_status, errcode);
		return;
	}
	if (out_of_domain)
	{
		STROM_SET_ERROR(chunk_status, StromError_DataStoreOutOfRange);
		return;
	}
	cl_uint		slot = gid - gid_lo;
	cl_ulong	addend = 1UL << pk->count_shift;
	bool		bad = false;
	/* a plain column of a NULL-free chunk: the value is there, the only thing
	 * that can be wrong is a zone map that does not bound it */
#define X(aidx,kind,attno)															\
	if (kind == 2)																	\
	{																				\
		cl_ulong	f = (cl_ulong)((cl_long)gpupreagg_agg_##aidx(&errcode, KP, KV).value - pk->bias[aidx]);	\
		bad |= (f > pk->vmax[aidx]);												\
		addend += f << pk->shift[aidx];												\
	}
	GPUPREAGG_PACK_LIST(X)
#undef X
	if (bad)
	{
		STROM_SET_ERROR(chunk_status, StromError_DataStoreCorruption);
		return;
	}
	if (pk->spill_at == 0)			/* (uniform) the fields hold every row the work-group folds */
		__hip_atomic_fetch_add((cl_ulong *)lds + slot, addend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	else
	{
		/*
		 * narrow fields (count + sums of the work-group's rows would need more than 64 bits):
		 * the add returns what it was added to.  The ONE add that takes a group's count to
		 * spill_at -- a quarter of the field -- moves the group's word to the slab
		 * (gpupreagg_packed_spill); three quarters of the field are the margin for the adds
		 * that land in between.  Should the field fill up all the same, the chunk goes back
		 * (never a wrong sum).
		 */
		cl_ulong	old = __hip_atomic_fetch_add((cl_ulong *)lds + slot, addend, __ATOMIC_RELAXED,
												 __HIP_MEMORY_SCOPE_WORKGROUP);
		cl_uint		n = (cl_uint)(old >> pk->count_shift) + 1u;
		STROM_SET_RECHECK_IF(chunk_status, n >= pk->count_limit);
		if (n == pk->spill_at)
			gpupreagg_packed_spill(lds, pk, slab, slot, old + addend, G);
	}
#define X(aidx,kind,attno)															\
	if (kind == 3)																	\
		__hip_atomic_fetch_add((cl_double *)(lds + gpupreagg_pack_word_offset(pk->word[aidx], G)) + slot,	\
							   (cl_double)gpupreagg_agg_##aidx(&errcode, KP, KV).value,	\
							   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	GPUPREAGG_PACK_LIST(X)
#undef X
}

/*
 * Unpack the work-group's LDS image into its slab of the standard shape.
 *
 * spilled: the fields were too narrow for all the rows a work-group folds in the chunk
 * (pk->spill_at != 0) -- the slab's count / integer-sum sections and flags were zeroed by
 * gpupreagg_packed_slab_zero at the start of the kernel and hold what gpupreagg_packed_spill
 * moved out of LDS; what is left in LDS is added to it.  (Read with agent-scope atomic loads:
 * the spills were atomics at the L2, this CU's L1 may still hold the zeroes.)
 */
STROM_DEVICE void
gpupreagg_packed_slab_zero(const gpupreagg_pack_ctl *pk, char *slab, cl_uint G)
{
	for (cl_uint g = threadIdx.x; g < G; g += GPUPREAGG_BLOCK)
	{
		((gpupreagg_flags_t *)slab)[g] = (gpupreagg_flags_t)0u;
#define X(aidx,kind,attno)																\
		if (kind == 1)																	\
			((cl_uint *)(slab + gpupreagg_image_offset(1 + aidx, G, 1)))[g] = 0u;		\
		else if (kind == 2)																\
			((cl_ulong *)(slab + gpupreagg_image_offset(1 + aidx, G, 1)))[g] = 0UL;
		GPUPREAGG_PACK_LIST(X)
#undef X
	}
}

STROM_DEVICE void
gpupreagg_store_slab_packed(char *lds, const gpupreagg_pack_ctl *pk, char *slab, cl_uint G, bool spilled)
{
	__syncthreads();
	for (cl_uint g = threadIdx.x; g < G; g += GPUPREAGG_BLOCK)
	{
		cl_ulong	w0 = ((const cl_ulong *)lds)[g];
		cl_ulong	n = w0 >> pk->count_shift;
		cl_uint		flags = 0;
		bool		seen = (n != 0);
		if (spilled)
			seen = seen || (__hip_atomic_load((gpupreagg_flags_t *)slab + g, __ATOMIC_RELAXED,
											  __HIP_MEMORY_SCOPE_AGENT) != 0);
#define X(aidx,kind,attno)																\
		{																				\
			char   *svals = slab + gpupreagg_image_offset(1 + aidx, G, 1);				\
			if (kind == 1)																\
				((cl_uint *)svals)[g] = (cl_uint)n +									\
					(spilled ? __hip_atomic_load((cl_uint *)svals + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);	\
			else if (kind == 2)															\
			{																			\
				cl_ulong f = (w0 >> pk->shift[aidx]) & pk->mask[aidx];					\
				((cl_long *)svals)[g] = (cl_long)f + (cl_long)n * pk->bias[aidx]		\
					+ (spilled ? (cl_long)__hip_atomic_load((cl_ulong *)svals + g, __ATOMIC_RELAXED,	\
															__HIP_MEMORY_SCOPE_AGENT) : 0L);	\
				flags |= (2u << aidx);													\
			}																			\
			else																		\
			{																			\
				((cl_ulong *)svals)[g] =												\
					((const cl_ulong *)(lds + gpupreagg_pack_word_offset(pk->word[aidx], G)))[g];	\
				flags |= (2u << aidx);													\
			}																			\
		}
		GPUPREAGG_PACK_LIST(X)
#undef X
		((gpupreagg_flags_t *)slab)[g] = (gpupreagg_flags_t)(seen ? (flags | GPUPREAGG_FLAG_SEEN) : 0u);
	}
}
#endif	/* GPUPREAGG_PACKABLE */

struct gpupreagg_column_tile {
#define X(attno,colidx,NAME)											\
	pg_##NAME##_base_t	v_##attno[GPUPREAGG_QUADS][4];					\
	cl_uint				nn_##attno[GPUPREAGG_QUADS];
	STROM_KVAR_LIST(X)
#undef X
	int __dummy;
};

STROM_DEVICE void
gpupreagg_load_kparams(strom_kparams &KP, const kern_parambuf *kparams, cl_int *errcode)
{
#define X(idx,NAME)	KP.KPARAM_##idx = pg_##NAME##_param(kparams, errcode, idx);
	STROM_KPARAM_LIST(X)
#undef X
	KP.__dummy = 0;
}

/*
 * A program is built either with the dense-id kernels or, for a session made
 * by strom_gpupreagg_create_hashed (which derives it from the caller's program
 * by defining GPUPREAGG_HASHED), with the hashed GROUP BY kernels: a query uses
 * one of the two, and hiprtc time is paid per kernel.
 */
#ifndef GPUPREAGG_HASHED
/* ====================================================================== *
 * dense-id reduction, COLUMN format
 * ====================================================================== */
template <bool PACKED>
__device__ __forceinline__ void
gpupreagg_dense_column_body(kern_gpupreagg *kgpreagg,
							const kern_data_store *kds,
							const gpupreagg_dense_ctl *ctl_in_memory,
							const gpupreagg_pack_ctl *pack_in_memory,
							char *slabs)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	/*
	 * the control block by value: read through the pointer, its fields
	 * (key_min / key_range / key_stride / remap) were scalar loads in EVERY row
	 * body, and the s_waitcnt lgkmcnt(0) behind each of them also waited for
	 * the previous row's LDS atomics -- lgkmcnt counts both
	 */
	const gpupreagg_dense_ctl ctl_by_value = *ctl_in_memory;
	const gpupreagg_dense_ctl *ctl = &ctl_by_value;
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
	cl_uint		nitems = kds->nitems;
	cl_uint		ntiles = (nitems + GPUPREAGG_TILE_ROWS - 1) / GPUPREAGG_TILE_ROWS;
	cl_uint		nsplits = ctl->nsplits;
	cl_uint		G = ctl->groups_per_split;
	cl_uint		NREP = ctl->nrep;
	cl_uint		split = blockIdx.x % nsplits;
	cl_uint		wg_in_split = blockIdx.x / nsplits;
	cl_uint		wgs_per_split = gridDim.x / nsplits;
	/*
	 * Several roles read every tile (state > LDS): put the nsplits
	 * work-groups of one tile stream on the SAME XCD (work-groups are
	 * dealt round-robin over the 8 XCDs) and read with cacheable loads,
	 * so that only the first reader goes to HBM and the siblings hit
	 * that XCD's L2.
	 */
	bool		shared_tiles = (nsplits > 1 && gridDim.x % (8 * nsplits) == 0);
	if (shared_tiles)
	{
		cl_uint	xcd = blockIdx.x & 7;
		cl_uint	slot = blockIdx.x >> 3;
		split = slot % nsplits;
		wg_in_split = (slot / nsplits) * 8 + xcd;
	}
	cl_uint		gid_lo = split * G;
	cl_uint		rep = threadIdx.x & (NREP - 1);
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	strom_kparams KP;
	gpupreagg_lds_layout L;

	gpupreagg_load_kparams(KP, kparams, &param_error);
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
	/* (by value, like the control block: its fields are read in every row body) */
	gpupreagg_pack_ctl pack_by_value;
	const gpupreagg_pack_ctl *pk = &pack_by_value;
	if (PACKED)
	{
		pack_by_value = *pack_in_memory;
		cl_uint	total = pk->nwords * gpupreagg_align16(8u * G);
		for (cl_uint i = threadIdx.x * 16; i < total; i += GPUPREAGG_BLOCK * 16)
			*(uint4 *)(lds + i) = make_uint4(0, 0, 0, 0);
		__syncthreads();
	}
	else
#endif
	{
		gpupreagg_remap_init(ctl);
		gpupreagg_lds_layout_init(L, G, NREP);
		gpupreagg_lds_init(lds, L, G, NREP);
	}

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
	const cl_uint rowflags = (KERN_GPUPREAGG_ZONE_BOUNDED(kgpreagg) ? ROWFLAG_ZONE_BOUNDED : 0u) |
		(any_nulls ? 0u : ROWFLAG_ALL_NOTNULL);

	char	   *my_slab = slabs + (size_t)(wg_in_split * nsplits + split) * ctl->slab_bytes;
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
	if (PACKED && pk->spill_at != 0)
	{
		/* narrow fields: groups move to the slab as their counts fill up (gpupreagg_packed_spill) */
		gpupreagg_packed_slab_zero(pk, my_slab, G);
		__syncthreads();
	}
#endif
	for (cl_uint tile = wg_in_split; tile < ntiles; tile += wgs_per_split)
	{
		cl_uint		tile_base = tile * GPUPREAGG_TILE_ROWS;
		bool		full_tile = (tile_base + GPUPREAGG_TILE_ROWS <= nitems);
		gpupreagg_column_tile T;


		if (full_tile && !any_nulls && shared_tiles)
		{
#pragma unroll
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, true, true, true>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
		else if (full_tile && !any_nulls)
		{
#pragma unroll
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, false>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
			cl_uint	row0 = tile_base + (k * GPUPREAGG_BLOCK + threadIdx.x) * 4;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if (full_tile || row0 + j < nitems)
				{
					strom_kvars	KV;
#define X(attno,colidx,NAME)													\
					KV.KVAR_##attno = pg_##NAME##_make(T.v_##attno[k][j],			\
													   !((T.nn_##attno[k] >> j) & 1));
					STROM_KVAR_LIST(X)
#undef X
					/* (the row's error slot starts as the parameters' and takes what turning a text
					 * column's offsets into addresses raises; nothing there for other programs) */
					cl_int		row_error = param_error;
					strom_kvars_from_column(KV, kds, &row_error);
					STROM_KVARS_FINISH(KV);
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
					if (PACKED)
						gpupreagg_packed_row(lds, ctl, pk, KP, KV, gid_lo, G, row_error, &chunk_status, my_slab);
					else
#endif
						gpupreagg_dense_row(lds, ctl, L, KP, KV, gid_lo, G, NREP, rep,
											row_error, &chunk_status, summag, rowflags);
				}
			}
		}
	}
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
	if (PACKED)
		gpupreagg_store_slab_packed(lds, pk, my_slab, G, pk->spill_at != 0);
	else
#endif
		gpupreagg_store_slab(lds, L, my_slab, G, NREP, &chunk_status, rowflags);
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
	gpupreagg_writeback_summag(kgpreagg, summag);
}

#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_dense_column(kern_gpupreagg *kgpreagg,
					   const kern_data_store *kds,
					   const gpupreagg_dense_ctl *ctl_in_memory,
					   char *slabs)
{
	gpupreagg_dense_column_body<false>(kgpreagg, kds, ctl_in_memory, NULL, slabs);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */

#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_packed_column(kern_gpupreagg *kgpreagg,
						const kern_data_store *kds,
						const gpupreagg_dense_ctl *ctl_in_memory,
						const gpupreagg_pack_ctl *pack_in_memory,
						char *slabs)
{
	gpupreagg_dense_column_body<true>(kgpreagg, kds, ctl_in_memory, pack_in_memory, slabs);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */
#endif

/* ====================================================================== *
 * dense-id reduction, any format / row map (one datum at a time)
 * ====================================================================== */
template <bool IS_COLUMN>
__device__ __forceinline__ void
gpupreagg_dense_generic_body(kern_gpupreagg *kgpreagg,
						const kern_data_store *kds,
						const kern_data_store *ktoast,
						const kern_row_map *krowmap,
						const gpupreagg_dense_ctl *ctl,
						char *slabs,
	char *lds)
{
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	bool		use_map = (krowmap != NULL && krowmap->nvalids >= 0);
	cl_uint		nrows = (use_map ? (cl_uint)krowmap->nvalids : kds->nitems);
	cl_uint		nsplits = ctl->nsplits;
	cl_uint		G = ctl->groups_per_split;
	cl_uint		NREP = ctl->nrep;
	cl_uint		split = blockIdx.x % nsplits;
	cl_uint		gid_lo = split * G;
	cl_uint		wg_in_split = blockIdx.x / nsplits;
	cl_uint		wgs_per_split = gridDim.x / nsplits;
	cl_uint		rep = threadIdx.x & (NREP - 1);
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	strom_kparams KP;
	gpupreagg_lds_layout L;

	gpupreagg_load_kparams(KP, kparams, &param_error);
	gpupreagg_remap_init(ctl);
	gpupreagg_lds_layout_init(L, G, NREP);
	gpupreagg_lds_init(lds, L, G, NREP);
	/* COLUMN chunk (row map, census): column pointers hoisted, no chunk
	 * header field is read per row */
	const bool	is_column = IS_COLUMN;		/* fixed per launch */
	const cl_int chunk_format = kds->format;
	const bool	row_family = (chunk_format == KDS_FORMAT_ROW || chunk_format == KDS_FORMAT_ROW_FLAT);
	const kern_coldir *coldir_g = KERN_DATA_STORE_COLDIR(kds);
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (is_column ? (const char *)kds + coldir_g[colidx].values_off : NULL);	\
	const cl_uint *nul_##attno = ((is_column && coldir_g[colidx].nulls_off != 0)	\
		? (const cl_uint *)((const char *)kds + coldir_g[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	for (size_t r = (size_t)wg_in_split * GPUPREAGG_BLOCK + threadIdx.x;
		 r < nrows;
		 r += (size_t)wgs_per_split * GPUPREAGG_BLOCK)
	{
		cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : (cl_uint)r);
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
		if (is_column)
			strom_kvars_from_column(KV, kds, &errcode);
		STROM_KVARS_FINISH(KV);
		gpupreagg_dense_row(lds, ctl, L, KP, KV, gid_lo, G, NREP, rep, errcode, &chunk_status, summag, 0u);
	}
	gpupreagg_store_slab(lds, L, slabs + (size_t)blockIdx.x * ctl->slab_bytes, G, NREP, &chunk_status);
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
	gpupreagg_writeback_summag(kgpreagg, summag);
}

#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_dense_generic(kern_gpupreagg *kgpreagg,
						const kern_data_store *kds,
						const kern_data_store *ktoast,
						const kern_row_map *krowmap,
						const gpupreagg_dense_ctl *ctl_in_memory,
						char *slabs)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	/* the control block by value (see gpupreagg_dense_column) */
	const gpupreagg_dense_ctl ctl_by_value = *ctl_in_memory;
	const gpupreagg_dense_ctl *ctl = &ctl_by_value;
	/* the chunk format is decided once per launch, not once per datum */
	if (kds->format == KDS_FORMAT_COLUMN)
		gpupreagg_dense_generic_body<true>(kgpreagg, kds, ktoast, krowmap, ctl, slabs, lds);
	else
		gpupreagg_dense_generic_body<false>(kgpreagg, kds, ktoast, krowmap, ctl, slabs, lds);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */

/* ====================================================================== *
 * dense-id reduction straight over a GpuHashJoin's result pairs
 *
 * The consumer the join's projection was made for, without the projection
 * ("fuse into consumer"): row i of the virtual
 * joined relation is result pair i = {outer row + 1, inner entry}; its
 * column n is column jmap->c[n].col of the outer COLUMN chunk (depth 0) or
 * of the inner relation (depth 1), the latter read from the slot-indexed
 * dimension arrays of a DIRECT, unique-key table
 * (hashjoin_build_dimcol_kernel) at slot = outer key - key_min.  The
 * program's (var N ...) are the virtual relation's columns.
 * ====================================================================== */
struct gpupreagg_joined_map {
	cl_uint		ncols;
	cl_int		key_col;			/* outer column (0-based) that is the join key */
	cl_int		key_attlen;
	cl_uint		nslots;
	cl_long		key_min;
	struct {
		cl_int		depth;
		cl_int		col;
		cl_ulong	dimvalues;		/* depth 1: device arrays by slot */
		cl_ulong	dimisnull;
	} c[64];
	/* gpupreagg_dense_lookup: packed slot records (hashjoin_build_dimrec_kernel); an inner
	 * column i then has c[i].dimvalues = byte offset of its value in the record and
	 * c[i].dimisnull = its bit in the record's flags word */
	cl_ulong	recs;
	cl_uint		reclen;
	/* NARROW records (hashjoin_dimrec_narrow_kernel, lookup only): reclen 2 or 4, the word is
	 * presence | NULL bits | (value - nmin) fields; inner column i = nmin[i] + field */
	cl_uint		narrow;
	cl_uint		nshift[64];
	cl_uint		nmask[64];
	cl_long		nmin[64];
};

#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_dense_joined(kern_gpupreagg *kgpreagg,
					   const kern_resultbuf *kresults,
					   const kern_data_store *kds,
					   const gpupreagg_joined_map *jmap,
					   const gpupreagg_dense_ctl *ctl_in_memory,
					   char *slabs)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	/* the control block by value (see gpupreagg_dense_column) */
	const gpupreagg_dense_ctl ctl_by_value = *ctl_in_memory;
	const gpupreagg_dense_ctl *ctl = &ctl_by_value;

	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
	cl_uint		nrows = kresults->nitems;
	cl_uint		nrels = kresults->nrels;
	cl_uint		nsplits = ctl->nsplits;
	cl_uint		G = ctl->groups_per_split;
	cl_uint		NREP = ctl->nrep;
	cl_uint		split = blockIdx.x % nsplits;
	cl_uint		gid_lo = split * G;
	cl_uint		wg_in_split = blockIdx.x / nsplits;
	cl_uint		wgs_per_split = gridDim.x / nsplits;
	cl_uint		rep = threadIdx.x & (NREP - 1);
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	strom_kparams KP;
	gpupreagg_lds_layout L;

	gpupreagg_load_kparams(KP, kparams, &param_error);
	gpupreagg_remap_init(ctl);
	gpupreagg_lds_layout_init(L, G, NREP);
	gpupreagg_lds_init(lds, L, G, NREP);
	/* per virtual column: where it comes from (uniform, hoisted) */
	/* inner columns come from the table's packed slot records (one record, one
	 * L2 request per pair: hashjoin_build_dimrec_kernel), see gpupreagg_dense_lookup */
#define X(attno,colidx,NAME)													\
	const bool	inner_##attno = (jmap->c[colidx].depth != 0);					\
	const cl_uint recoff_##attno = (cl_uint)jmap->c[colidx].dimvalues;			\
	const cl_uint recbit_##attno = (cl_uint)jmap->c[colidx].dimisnull;			\
	const char *val_##attno = (inner_##attno ? NULL								\
							   : (const char *)kds + coldir[jmap->c[colidx].col].values_off);	\
	const char *nul_##attno = (inner_##attno ? NULL								\
							   : (coldir[jmap->c[colidx].col].nulls_off != 0		\
								  ? (const char *)kds + coldir[jmap->c[colidx].col].nulls_off : NULL));
	STROM_KVAR_LIST(X)
#undef X
	const char *recs = (const char *)jmap->recs;
	cl_uint		reclen = jmap->reclen;
	const char *keyvals = (const char *)kds + coldir[jmap->key_col].values_off;
	cl_int		key_attlen = jmap->key_attlen;
	cl_long		key_min = strom_uniform((cl_long)jmap->key_min);
	cl_uint		nslots = strom_uniform((cl_uint)jmap->nslots);

	for (size_t r = (size_t)wg_in_split * GPUPREAGG_BLOCK + threadIdx.x;
		 r < nrows;
		 r += (size_t)wgs_per_split * GPUPREAGG_BLOCK)
	{
		cl_uint		outer_row = (cl_uint)(kresults->results[(size_t)nrels * r] - 1);
		cl_long		key = (key_attlen == 4 ? (cl_long)((const cl_int *)keyvals)[outer_row]
						   : key_attlen == 2 ? (cl_long)((const cl_short *)keyvals)[outer_row]
						   : key_attlen == 1 ? (cl_long)((const cl_char *)keyvals)[outer_row]
						   : ((const cl_long *)keyvals)[outer_row]);
		cl_ulong	slot = (cl_ulong)(key - key_min);
		strom_kvars	KV;
		cl_int		errcode = param_error;
		if (slot >= nslots)
		{
			/* not a row this table can have matched */
			STROM_SET_ERROR(&chunk_status, StromError_DataStoreCorruption);
			continue;
		}
		/* the pair's inner record, whole (8 / 16 bytes in one load) */
		cl_uint		words[4];
		{
			const char *rec = recs + (size_t)reclen * slot;
			if (reclen == 8)
			{
				cl_ulong w = *(const cl_ulong *)rec;
				words[0] = (cl_uint)w; words[1] = (cl_uint)(w >> 32); words[2] = words[3] = 0;
			}
			else if (reclen == 16)
			{
				uint4 q = *(const uint4 *)rec;
				words[0] = q.x; words[1] = q.y; words[2] = q.z; words[3] = q.w;
			}
			else
			{
				words[0] = *(const cl_uint *)rec;
				words[1] = words[2] = words[3] = 0;
			}
		}
#define X(attno,colidx,NAME)													\
		if (inner_##attno)														\
		{																		\
			pg_##NAME##_base_t val;												\
			if (reclen <= 16)													\
			{																	\
				cl_uint	wi = recoff_##attno >> 2;								\
				cl_uint	lo = (wi == 1 ? words[1] : wi == 2 ? words[2] : words[3]);	\
				cl_uint	hi = (wi == 2 ? words[3] : 0u);							\
				cl_ulong bits = (((cl_ulong)hi << 32) | lo) >> ((recoff_##attno & 3u) * 8u);	\
				__builtin_memcpy(&val, &bits, sizeof(val));						\
			}																	\
			else																\
				val = *(const pg_##NAME##_base_t *)(recs + (size_t)reclen * slot + recoff_##attno);	\
			KV.KVAR_##attno = pg_##NAME##_make(val, ((words[0] >> recbit_##attno) & 1u) != 0);	\
		}																		\
		else																	\
			KV.KVAR_##attno = STROM_COLUMN_REF(NAME, val_##attno, nul_##attno, outer_row);
		STROM_KVAR_LIST_GROUPING(X)
		STROM_KVARS_FINISH(KV);
		if (nsplits > 1)
		{
			/* several id-range roles read every pair: whose row this is follows
			 * from the columns the keys read -- gather the others only for the
			 * role's own rows (gpupreagg_dense_row decides again, with errors) */
			cl_int		e2 = errcode;
			cl_uint		gid = 0;
			bool		out_of_domain = false;
#define Y(kidx,resno,NAME)														\
			{																	\
				pg_##NAME##_t kv = gpupreagg_key_##kidx(&e2, KP, KV);			\
				cl_long		off64 = (cl_long)kv.value - ctl->key_min[kidx];		\
				cl_uint		range = ctl->key_range[kidx];						\
				cl_uint		off = (kv.isnull ? range : (cl_uint)off64);			\
				if (!kv.isnull && (off64 < 0 || off64 >= (cl_long)range))		\
					out_of_domain = true;										\
				gid += (kidx == 0 ? off : off * ctl->key_stride[kidx]);	/* (stride 0 is 1) */								\
			}
			GPUPREAGG_KEY_LIST(Y)
#undef Y
			if (e2 == StromError_Success && !out_of_domain &&
				gpupreagg_remap_gid(ctl, gid) && gid - gid_lo >= G)
				continue;
		}
		STROM_KVAR_LIST_REST(X)
#undef X
		gpupreagg_dense_row(lds, ctl, L, KP, KV, gid_lo, G, NREP, rep, errcode, &chunk_status, summag, 0u);
	}
	gpupreagg_store_slab(lds, L, slabs + (size_t)blockIdx.x * ctl->slab_bytes, G, NREP, &chunk_status);
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
	gpupreagg_writeback_summag(kgpreagg, summag);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */

/* ====================================================================== *
 * join lookup + dense-id reduction in ONE pass over the outer chunk
 *
 * The whole query shape "fact JOIN dim GROUP BY ..." for a dimension with a
 * DIRECT index and unique keys: the outer COLUMN chunk streams through the
 * tile loader exactly as in gpupreagg_dense_column, a row's slot is
 * outer key - key_min, a row without a partner (NULL key, key outside the
 * table, empty slot) is dropped, and a virtual column of depth 1 comes from
 * the table's slot-indexed arrays (a few MB, L2-resident).  No result pairs
 * are written or read at all.  The program's (var N ...) and its qual see
 * the same virtual relation as gpupreagg_dense_joined; int4 / int8 keys.
 * ====================================================================== */
#ifndef GPUPREAGG_ABLATE
#define GPUPREAGG_ABLATE 0
#endif
#if (GPUPREAGG_ABLATE != 0) && !defined(STROM_DIAGNOSTIC_BUILD)
#error "GPUPREAGG_ABLATE builds leave work out and give wrong results: measurement only (set STROM_DIAGNOSTIC_BUILD=1, as scripts/gpu_*_ablate* do)"
#endif
#if GPUPREAGG_ABLATE & 1
STROM_DEVICE cl_uint
gpupreagg_ablate_sink(const strom_kvars &KV)
{
	cl_uint		acc = 0;
#define X(attno,colidx,NAME)	acc ^= (cl_uint)KV.KVAR_##attno.value + (cl_uint)KV.KVAR_##attno.isnull;
	STROM_KVAR_LIST(X)
#undef X
	return acc;
}
#endif

template <typename KEY_T, bool PACKED>
__device__ __forceinline__ void
gpupreagg_dense_lookup_body(kern_gpupreagg *kgpreagg,
							const kern_data_store *kds,
							const gpupreagg_joined_map *jmap,
							const gpupreagg_dense_ctl *ctl,
							const gpupreagg_pack_ctl *pack_in_memory,
							char *slabs, char *lds)
{
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
	cl_uint		nitems = strom_uniform((cl_uint)kds->nitems);
	cl_uint		ntiles = (nitems + GPUPREAGG_TILE_ROWS - 1) / GPUPREAGG_TILE_ROWS;
	cl_uint		nsplits = ctl->nsplits;
	cl_uint		G = ctl->groups_per_split;
	cl_uint		NREP = ctl->nrep;
	cl_uint		split = blockIdx.x % nsplits;
	cl_uint		wg_in_split = blockIdx.x / nsplits;
	cl_uint		wgs_per_split = gridDim.x / nsplits;
	cl_uint		gid_lo = split * G;
	cl_uint		rep = threadIdx.x & (NREP - 1);
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	strom_kparams KP;
	gpupreagg_lds_layout L;

	gpupreagg_load_kparams(KP, kparams, &param_error);
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
	gpupreagg_pack_ctl pack_by_value;
	const gpupreagg_pack_ctl *pk = &pack_by_value;
	if (PACKED)
	{
		pack_by_value = *pack_in_memory;
		cl_uint	total = pk->nwords * gpupreagg_align16(8u * G);
		for (cl_uint i = threadIdx.x * 16; i < total; i += GPUPREAGG_BLOCK * 16)
			*(uint4 *)(lds + i) = make_uint4(0, 0, 0, 0);
		__syncthreads();
	}
	else
#endif
	{
		gpupreagg_remap_init(ctl);
		gpupreagg_lds_layout_init(L, G, NREP);
		gpupreagg_lds_init(lds, L, G, NREP);
	}
	/*
	 * Which virtual column is an inner one, how long a slot record is and whether it is the
	 * narrow form are facts of the request (the column mapping, the table) -- as run-time values
	 * they were ~40 scalar registers per kernel (spilled into VGPR lanes), a maze of scalar
	 * branches around the record decode with a full s_waitcnt at every join, and flat loads.
	 * The host therefore builds the program FOR the mapping (GPUPREAGG_LOOKUP_INNER_MASK / _RECLEN
	 * / _NARROW / _KEYLEN: compile-time constants, gpupreagg.cpp: lookup_program) and the compiler
	 * keeps one straight path; without those defines the same code reads the values from the
	 * map (any mapping, no build).
	 */
#if defined(GPUPREAGG_LOOKUP_INNER_MASK)
#define LOOKUP_INNER(colidx)	((((GPUPREAGG_LOOKUP_INNER_MASK) >> (colidx)) & 1UL) != 0)
	/* (pointers computed from kernel arguments and scalar loads ARE uniform, and keep their
	 * address space: through strom_uniform's readfirstlane they come back as flat pointers) */
#define LOOKUP_PTR(x)			(x)
#else
#define LOOKUP_INNER(colidx)	(strom_uniform((cl_int)jmap->c[colidx].depth) != 0)
#define LOOKUP_PTR(x)			strom_uniform(x)
#endif
	/* (the slot records are global memory: said so, the loads are global_load, not flat_load) */
#ifndef GPUPREAGG_LOOKUP_PROBE
#define GPUPREAGG_LOOKUP_PROBE	0
#endif
#if GPUPREAGG_LOOKUP_PROBE == 2
	/* (experiment: agent-scope loads are served by the L2 without a line fill in the CU's L1) */
#define LOOKUP_GLOBAL(T, p)		__hip_atomic_load((const __attribute__((address_space(1))) T *)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#elif GPUPREAGG_LOOKUP_PROBE == 1
#define LOOKUP_GLOBAL(T, p)		__builtin_nontemporal_load((const __attribute__((address_space(1))) T *)(p))
#else
#define LOOKUP_GLOBAL(T, p)		(*(const __attribute__((address_space(1))) T *)(p))
#endif
#define X(attno,colidx,NAME)													\
	const bool	inner_##attno = LOOKUP_INNER(colidx);								\
	const cl_uint recoff_##attno = strom_uniform((cl_uint)jmap->c[colidx].dimvalues);	\
	const cl_uint recbit_##attno = strom_uniform((cl_uint)jmap->c[colidx].dimisnull);	\
	const char *val_##attno = LOOKUP_PTR(inner_##attno ? (const char *)NULL		\
							   : (const char *)kds + coldir[jmap->c[colidx].col].values_off);	\
	const char *nul_##attno = LOOKUP_PTR(inner_##attno ? (const char *)NULL		\
							   : (coldir[jmap->c[colidx].col].nulls_off != 0		\
								  ? (const char *)kds + coldir[jmap->c[colidx].col].nulls_off : (const char *)NULL));
	STROM_KVAR_LIST(X)
#undef X
	const char *keyvals = LOOKUP_PTR((const char *)kds + coldir[jmap->key_col].values_off);
	const cl_uint *keynulls = LOOKUP_PTR(coldir[jmap->key_col].nulls_off != 0
							   ? (const cl_uint *)((const char *)kds + coldir[jmap->key_col].nulls_off)
							   : (const cl_uint *)NULL);
	const char *recs = LOOKUP_PTR((const char *)jmap->recs);
#if defined(GPUPREAGG_LOOKUP_INNER_MASK)
	const cl_uint reclen = GPUPREAGG_LOOKUP_RECLEN;
	const bool	narrow = (GPUPREAGG_LOOKUP_NARROW != 0);
#else
	cl_uint		reclen = strom_uniform((cl_uint)jmap->reclen);
	const bool	narrow = (strom_uniform((cl_uint)jmap->narrow) != 0);
#endif
#define X(attno,colidx,NAME)													\
	const cl_uint nshift_##attno = strom_uniform((cl_uint)jmap->nshift[colidx]);	\
	const cl_uint nmask_##attno = strom_uniform((cl_uint)jmap->nmask[colidx]);	\
	const cl_long nmin_##attno = strom_uniform((cl_long)jmap->nmin[colidx]);
	STROM_KVAR_LIST(X)
#undef X
	/* a qual that reads outer columns only is evaluated BEFORE the probe: a row it
	 * rejects (without an error) costs no L2 request; gpupreagg_dense_row evaluates
	 * it again for the rows that do have a partner, errors included */
	cl_ulong	inner_mask = 0;
#define X(attno,colidx,NAME)	inner_mask |= (inner_##attno ? (1UL << colidx) : 0UL);
	STROM_KVAR_LIST(X)
#undef X
	const bool	qual_first = ((GPUPREAGG_QUAL_VARMASK & inner_mask) == 0 && GPUPREAGG_QUAL_VARMASK != 0);
	bool		any_nulls = (keynulls != NULL);		/* wave-uniform: picks the bitmap-free loader */
#define X(attno,colidx,NAME)	any_nulls = any_nulls || (nul_##attno != NULL);
	STROM_KVAR_LIST(X)
#undef X
	/* (an inner column may be NULL whatever the chunk's bitmaps say: no count aliasing when the
	 * program reads one; its columns are virtual, so the host bounds no sum by zone maps) */
	const cl_uint rowflags = ((any_nulls || inner_mask != 0) ? 0u : ROWFLAG_ALL_NOTNULL);
	cl_long		key_min = jmap->key_min;
	cl_uint		nslots = jmap->nslots;

	char	   *my_slab = slabs + (size_t)blockIdx.x * ctl->slab_bytes;
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
	if (PACKED && pk->spill_at != 0)
	{
		/* narrow fields: groups move to the slab as their counts fill up (gpupreagg_packed_spill) */
		gpupreagg_packed_slab_zero(pk, my_slab, G);
		__syncthreads();
	}
#endif
	/*
	 * A software pipeline over the tiles: the column loads of tile t+1 are issued right behind
	 * the probes of tile t, before anything of tile t is waited for -- loads return in order,
	 * so the probes complete under the next tile's stream instead of after a drained queue
	 * (one work-group per CU: four waves per SIMD do not hide two exposed round trips per
	 * tile by themselves).
	 */
	auto load_tile = [&](cl_uint tile, gpupreagg_column_tile &T, KEY_T (&keyq)[GPUPREAGG_QUADS][4],
						 cl_uint (&keynn)[GPUPREAGG_QUADS])
	{
		cl_uint		tile_base = tile * GPUPREAGG_TILE_ROWS;
		bool		full_tile = (tile_base + GPUPREAGG_TILE_ROWS <= nitems);
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
			cl_uint	row0 = tile_base + (k * GPUPREAGG_BLOCK + threadIdx.x) * 4;
			if (full_tile && !any_nulls)
			{
				/* no column of the chunk has a NULL bitmap: no bitmap words are fetched (one
				 * request per quad and column otherwise, in a kernel bound by requests) */
				strom_column_load_quad<KEY_T, true, true>(keyvals, keynulls, row0, nitems, keyq[k], keynn[k]);
#define X(attno,colidx,NAME)													\
				if (!inner_##attno)												\
					strom_column_load_quad<pg_##NAME##_base_t, true, true>(val_##attno, (const cl_uint *)nul_##attno,	\
															   row0, nitems, T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
			else if (full_tile)
			{
				strom_column_load_quad<KEY_T, true>(keyvals, keynulls, row0, nitems, keyq[k], keynn[k]);
#define X(attno,colidx,NAME)													\
				if (!inner_##attno)												\
					strom_column_load_quad<pg_##NAME##_base_t, true>(val_##attno, (const cl_uint *)nul_##attno,	\
															   row0, nitems, T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
			else
			{
				strom_column_load_quad<KEY_T, false>(keyvals, keynulls, row0, nitems, keyq[k], keynn[k]);
#define X(attno,colidx,NAME)													\
				if (!inner_##attno)												\
					strom_column_load_quad<pg_##NAME##_base_t, false>(val_##attno, (const cl_uint *)nul_##attno,	\
															   row0, nitems, T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
	};
	gpupreagg_column_tile T, Tnext;
	KEY_T		keyq[GPUPREAGG_QUADS][4], keyq_next[GPUPREAGG_QUADS][4];
	cl_uint		keynn[GPUPREAGG_QUADS], keynn_next[GPUPREAGG_QUADS];

	if (wg_in_split < ntiles)
		load_tile(wg_in_split, T, keyq, keynn);
	for (cl_uint tile = wg_in_split; tile < ntiles; tile += wgs_per_split)
	{
		cl_uint		tile_base = tile * GPUPREAGG_TILE_ROWS;
		bool		full_tile = (tile_base + GPUPREAGG_TILE_ROWS <= nitems);
		bool		more = (tile + wgs_per_split < ntiles);		/* (uniform) */

		/*
		 * every lookup of the tile is issued before the first is used (a
		 * row-by-row "is there a partner? then fetch its columns" is a chain
		 * of dependent loads per row: 1.5 ms per 1e8 rows that way).  Rows
		 * without a partner look at slot 0, harmlessly.
		 */
		cl_uint		slot[GPUPREAGG_QUADS][4];
		cl_uint		gone[GPUPREAGG_QUADS];			/* bit j: no partner */
		cl_uint		qual_ok[GPUPREAGG_QUADS];		/* bit j: the qual was seen to pass, without an error */
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
			cl_uint	row0 = tile_base + (k * GPUPREAGG_BLOCK + threadIdx.x) * 4;
			gone[k] = 0;
			qual_ok[k] = 0;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				cl_ulong	s64 = (cl_ulong)((cl_long)keyq[k][j] - key_min);
				bool		live = ((full_tile || row0 + j < nitems) && ((keynn[k] >> j) & 1) && s64 < nslots);
				if (qual_first && live)
				{
					strom_kvars	KV;
					cl_int		qerr = param_error;
#define X(attno,colidx,NAME)													\
					KV.KVAR_##attno = (inner_##attno								\
						? pg_##NAME##_make((pg_##NAME##_base_t)0, true)			\
						: pg_##NAME##_make(T.v_##attno[k][j], !((T.nn_##attno[k] >> j) & 1)));
					STROM_KVAR_LIST(X)
#undef X
					STROM_KVARS_FINISH(KV);
					pg_bool_t	rc = gpupreagg_qual_eval(&qerr, KP, KV);
					live = !(qerr == StromError_Success && !EVAL(rc));
					qual_ok[k] |= ((qerr == StromError_Success && EVAL(rc)) ? (1u << j) : 0u);
				}
				slot[k][j] = (live ? (cl_uint)s64 : 0u);
				gone[k] |= (live ? 0u : (1u << j));
			}
		}
		cl_uint	words[GPUPREAGG_QUADS][4][4];	/* records of 8 / 16 bytes: the whole record, fetched in ONE load */
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				const char *rec = recs + (size_t)reclen * slot[k][j];
				/* a second load of the same line is a second L2 request, and this
				 * kernel is bound by the L2 request rate */
#if GPUPREAGG_ABLATE & 2
				if (reclen != 0)			/* (measurement only: no record is read) */
				{
					words[k][j][0] = 1;
					words[k][j][1] = slot[k][j] % 10000u;
					words[k][j][2] = words[k][j][3] = 0;
				}
				else
#endif
				if (narrow)
				{
					/* 2- or 4-byte record: flags and value fields in one word */
					words[k][j][0] = (reclen == 2 ? (cl_uint)LOOKUP_GLOBAL(cl_ushort, rec) : LOOKUP_GLOBAL(cl_uint, rec));
					words[k][j][1] = words[k][j][2] = words[k][j][3] = 0;
				}
				else if (reclen == 8)
				{
					cl_ulong w = LOOKUP_GLOBAL(cl_ulong, rec);
					words[k][j][0] = (cl_uint)w;
					words[k][j][1] = (cl_uint)(w >> 32);
					words[k][j][2] = words[k][j][3] = 0;
				}
				else if (reclen == 16)
				{
					uint4 q = *(const __attribute__((address_space(1))) uint4 *)rec;
					words[k][j][0] = q.x; words[k][j][1] = q.y; words[k][j][2] = q.z; words[k][j][3] = q.w;
				}
				else
				{
					words[k][j][0] = LOOKUP_GLOBAL(cl_uint, rec);
					words[k][j][1] = words[k][j][2] = words[k][j][3] = 0;
				}
			}
		}
		/* the next tile's columns: behind this tile's probes, ahead of their use */
		if (more)
			load_tile(tile + wgs_per_split, Tnext, keyq_next, keynn_next);
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
			cl_uint	flags[4];
			cl_uint	ab = 0;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				flags[j] = words[k][j][0];
				ab |= ((flags[j] & 1u) ? 0u : (1u << j));
			}
			gone[k] |= ab;
#define X(attno,colidx,NAME)													\
			if (inner_##attno)													\
			{																	\
				cl_uint nn = 0;													\
				_Pragma("unroll")												\
				for (int j = 0; j < 4; j++)										\
				{																\
					pg_##NAME##_base_t val;										\
					if (narrow)													\
						val = (pg_##NAME##_base_t)(nmin_##attno +				\
							(cl_long)((flags[j] >> nshift_##attno) & nmask_##attno));	\
					else if (reclen <= 16)										\
					{															\
						/* (offsets are uniform: the selects are scalar) */		\
						cl_uint	wi = recoff_##attno >> 2;						\
						cl_uint	lo = (wi == 1 ? words[k][j][1] : wi == 2 ? words[k][j][2] : words[k][j][3]);	\
						cl_uint	hi = (wi == 2 ? words[k][j][3] : 0u);				\
						cl_ulong bits = (((cl_ulong)hi << 32) | lo) >> ((recoff_##attno & 3u) * 8u);	\
						__builtin_memcpy(&val, &bits, sizeof(val));				\
					}															\
					else														\
						val = *(const pg_##NAME##_base_t *)						\
							(recs + (size_t)reclen * slot[k][j] + recoff_##attno);	\
					T.v_##attno[k][j] = val;									\
					nn |= (((flags[j] >> recbit_##attno) & 1u) ? 0u : (1u << j));	\
				}																\
				T.nn_##attno[k] = nn;											\
			}
			STROM_KVAR_LIST(X)
#undef X
		}
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if (!((gone[k] >> j) & 1))
				{
					strom_kvars	KV;
#define X(attno,colidx,NAME)													\
					KV.KVAR_##attno = pg_##NAME##_make(T.v_##attno[k][j],			\
													   !((T.nn_##attno[k] >> j) & 1));
					STROM_KVAR_LIST(X)
#undef X
					STROM_KVARS_FINISH(KV);
#if GPUPREAGG_ABLATE & 1
					/* (measurement only: nothing is accumulated) */
					if (gpupreagg_ablate_sink(KV) == 0x12345677)
						chunk_status = StromError_CpuReCheck;
					else if (false)
#endif
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
					if (PACKED)
						gpupreagg_packed_row(lds, ctl, pk, KP, KV, gid_lo, G, param_error, &chunk_status, my_slab,
											 (qual_ok[k] >> j) & 1);
					else
#endif
						gpupreagg_dense_row(lds, ctl, L, KP, KV, gid_lo, G, NREP, rep,
											param_error, &chunk_status, summag, rowflags, (qual_ok[k] >> j) & 1);
				}
			}
		}
		/* the next tile becomes the current one (register moves; the loads may still be in flight) */
		if (more)
		{
			T = Tnext;
#pragma unroll
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				keynn[k] = keynn_next[k];
#pragma unroll
				for (int j = 0; j < 4; j++)
					keyq[k][j] = keyq_next[k][j];
			}
		}
	}
#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
	if (PACKED)
		gpupreagg_store_slab_packed(lds, pk, my_slab, G, pk->spill_at != 0);
	else
#endif
		gpupreagg_store_slab(lds, L, my_slab, G, NREP, &chunk_status, rowflags);
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
	gpupreagg_writeback_summag(kgpreagg, summag);
}

extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_dense_lookup(kern_gpupreagg *kgpreagg,
					   const kern_data_store *kds,
					   const gpupreagg_joined_map *jmap,
					   const gpupreagg_dense_ctl *ctl_in_memory,
					   char *slabs)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	/* the control block by value (see gpupreagg_dense_column) */
	const gpupreagg_dense_ctl ctl_by_value = *ctl_in_memory;
	const gpupreagg_dense_ctl *ctl = &ctl_by_value;
#if defined(GPUPREAGG_LOOKUP_KEYLEN) && GPUPREAGG_LOOKUP_KEYLEN == 8
	gpupreagg_dense_lookup_body<cl_long, false>(kgpreagg, kds, jmap, ctl, NULL, slabs, lds);
#elif defined(GPUPREAGG_LOOKUP_KEYLEN)
	gpupreagg_dense_lookup_body<cl_int, false>(kgpreagg, kds, jmap, ctl, NULL, slabs, lds);
#else
	if (jmap->key_attlen == 8)
		gpupreagg_dense_lookup_body<cl_long, false>(kgpreagg, kds, jmap, ctl, NULL, slabs, lds);
	else
		gpupreagg_dense_lookup_body<cl_int, false>(kgpreagg, kds, jmap, ctl, NULL, slabs, lds);
#endif
}

#if defined(GPUPREAGG_PACKABLE) && GPUPREAGG_PACKABLE
/* the same with packed accumulators (see gpupreagg_packed_column): the summed
 * columns must be OUTER columns without NULLs -- their zone maps bound the fields */
extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_packed_lookup(kern_gpupreagg *kgpreagg,
						const kern_data_store *kds,
						const gpupreagg_joined_map *jmap,
						const gpupreagg_dense_ctl *ctl_in_memory,
						const gpupreagg_pack_ctl *pack_in_memory,
						char *slabs)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	const gpupreagg_dense_ctl ctl_by_value = *ctl_in_memory;
	const gpupreagg_dense_ctl *ctl = &ctl_by_value;
#if defined(GPUPREAGG_LOOKUP_KEYLEN) && GPUPREAGG_LOOKUP_KEYLEN == 8
	gpupreagg_dense_lookup_body<cl_long, true>(kgpreagg, kds, jmap, ctl, pack_in_memory, slabs, lds);
#elif defined(GPUPREAGG_LOOKUP_KEYLEN)
	gpupreagg_dense_lookup_body<cl_int, true>(kgpreagg, kds, jmap, ctl, pack_in_memory, slabs, lds);
#else
	if (jmap->key_attlen == 8)
		gpupreagg_dense_lookup_body<cl_long, true>(kgpreagg, kds, jmap, ctl, pack_in_memory, slabs, lds);
	else
		gpupreagg_dense_lookup_body<cl_int, true>(kgpreagg, kds, jmap, ctl, pack_in_memory, slabs, lds);
#endif
}
#endif

/* ====================================================================== *
 * register accumulators: at most GPUPREAGG_REG_GROUPS dense ids
 *
 * LDS atomics cost ~24 (u32) / ~53 (u64) / ~110 (f64) cycles per wave
 * instruction on gfx950 (profiles/r01_preagg_atomics.txt), so a
 * low-cardinality GROUP BY (TPC-H Q1: 6 groups, 9 partial columns; any
 * aggregate without GROUP BY) is atomic-bound in the LDS path whatever
 * the replication.  Here every thread keeps one accumulator per
 * (aggregate, group) in registers and folds a row with predicated selects
 * (no memory traffic at all); at the end a wave tree-reduces with
 * shuffles and its lane 0 merges into the work-group's LDS image, which
 * then leaves through the same slab / merge path as the LDS kernels.
 * ====================================================================== */
#ifndef GPUPREAGG_REG_GROUPS
#define GPUPREAGG_REG_GROUPS	8
#endif
#ifndef GPUPREAGG_REG_BLOCK
#define GPUPREAGG_REG_BLOCK		256
#endif
#define GPUPREAGG_REG_TILE_ROWS	(GPUPREAGG_REG_BLOCK * 4 * GPUPREAGG_QUADS)

template <int NG>
struct gpupreagg_reg_state {
#define X(aidx,resno,OP,NAME)	cl_ulong v_##aidx[NG];
	GPUPREAGG_AGG_LIST(X)
#undef X
	cl_uint		flags[NG];
};

template <int NG>
STROM_DEVICE void
gpupreagg_reg_row(gpupreagg_reg_state<NG> &S, const gpupreagg_dense_ctl *ctl,
				  const strom_kparams &KP, const strom_kvars &KV,
				  cl_int param_error, cl_int *chunk_status, cl_ulong &summag, cl_uint rowflags)
{
	cl_int		errcode = param_error;
	pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KV);
	cl_uint		gid = 0;
	bool		out_of_domain = false;

	if (errcode == StromError_Success && !EVAL(rc))
		return;
#define X(kidx,resno,NAME)															\
	{																				\
		pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);					\
		cl_long		off64 = (cl_long)kv.value - ctl->key_min[kidx];					\
		cl_uint		range = ctl->key_range[kidx];									\
		cl_uint		off = (kv.isnull ? range : (cl_uint)off64);						\
		if (!kv.isnull && (off64 < 0 || off64 >= (cl_long)range))					\
			out_of_domain = true;													\
		gid += (kidx == 0 ? off : off * ctl->key_stride[kidx]);	/* (stride 0 is 1) */											\
	}
	GPUPREAGG_KEY_LIST(X)
#undef X
	if (!out_of_domain && !gpupreagg_remap_gid(ctl, gid))
		out_of_domain = true;
#define X(aidx,resno,OP,NAME)														\
	pg_##NAME##_t av_##aidx = gpupreagg_agg_##aidx(&errcode, KP, KV);
	GPUPREAGG_AGG_LIST(X)
#undef X
	if (errcode != StromError_Success)
	{
		STROM_SET_ERROR(chunk_status, errcode);
		return;
	}
	if (out_of_domain || gid >= (cl_uint)NG)
	{
		STROM_SET_ERROR(chunk_status, StromError_DataStoreOutOfRange);
		return;
	}
	cl_uint		need = GPUPREAGG_FLAG_SEEN;
#define X(aidx,resno,OP,NAME)														\
	{																				\
		typedef pg_##NAME##_base_t base_t;											\
		bool		has = !av_##aidx.isnull;										\
		cl_ulong	x;																\
		if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)								\
			x = (has ? (cl_ulong)(cl_uint)av_##aidx.value : 0);						\
		else if (gpupreagg_is_float<base_t>::value)									\
			x = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PSUM								\
				 ? (cl_ulong)__double_as_longlong((cl_double)av_##aidx.value)		\
				 : gpupreagg_f64_ordered((cl_double)av_##aidx.value));				\
		else																		\
			x = (cl_ulong)(cl_long)av_##aidx.value;									\
		if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, base_t>::value && GPUPREAGG_MEASURE_SUM(aidx, rowflags))	\
			summag |= (has ? gpupreagg_sum_magnitude((cl_long)x) : 0UL);			\
		if (GPUPREAGG_OP_##OP != GPUPREAGG_OP_NROWS && has)							\
			need |= (2u << aidx);													\
		_Pragma("unroll")															\
		for (int g = 0; g < NG; g++)												\
		{																			\
			bool hit = (has && gid == (cl_uint)g);									\
			cl_ulong cur = S.v_##aidx[g];											\
			cl_ulong nxt = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS ? cur + x		\
							: gpupreagg_merge8<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS	\
											   ? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP, base_t>(cur, x));	\
			S.v_##aidx[g] = (hit ? nxt : cur);										\
		}																			\
	}
	GPUPREAGG_AGG_LIST(X)
#undef X
#pragma unroll
	for (int g = 0; g < NG; g++)
		S.flags[g] |= (gid == (cl_uint)g ? need : 0u);
}

STROM_DEVICE cl_ulong
gpupreagg_shfl_xor_u64(cl_ulong v, int mask)
{
	cl_uint lo = __shfl_xor((cl_uint)v, mask, STROM_WAVE);
	cl_uint hi = __shfl_xor((cl_uint)(v >> 32), mask, STROM_WAVE);
	return ((cl_ulong)hi << 32) | lo;
}

template <int NG>
__device__ __forceinline__ void
gpupreagg_reg_kernel_body(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
						  const gpupreagg_dense_ctl *ctl, char *slabs, char *lds)
{
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
	cl_uint		nitems = kds->nitems;
	cl_uint		ntiles = (nitems + GPUPREAGG_REG_TILE_ROWS - 1) / GPUPREAGG_REG_TILE_ROWS;
	cl_uint		G = ctl->groups_per_split;
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	strom_kparams KP;
	gpupreagg_lds_layout L;
	gpupreagg_reg_state<NG> S;

	gpupreagg_load_kparams(KP, kparams, &param_error);
	gpupreagg_remap_init(ctl);
	gpupreagg_lds_layout_init(L, G, 1);
	/* LDS image (one replica) is only touched at the very end */
	for (cl_uint i = threadIdx.x * 16; i < L.total; i += GPUPREAGG_REG_BLOCK * 16)
		*(uint4 *)(lds + i) = make_uint4(0, 0, 0, 0);
	__syncthreads();
#define X(aidx,resno,OP,NAME)															\
	if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN || GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMAX)	\
		for (cl_uint i = threadIdx.x; i < G; i += GPUPREAGG_REG_BLOCK)					\
			((cl_ulong *)(lds + L.vals_off[aidx]))[i] =									\
				gpupreagg_identity<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>();
	GPUPREAGG_AGG_LIST(X)
#undef X
	__syncthreads();
#pragma unroll
	for (int g = 0; g < NG; g++)
	{
		S.flags[g] = 0;
#define X(aidx,resno,OP,NAME)															\
		S.v_##aidx[g] = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS ? 0UL :				\
						 gpupreagg_identity<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS		\
											? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,	\
											pg_##NAME##_base_t>());
		GPUPREAGG_AGG_LIST(X)
#undef X
	}
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
	const cl_uint rowflags = (KERN_GPUPREAGG_ZONE_BOUNDED(kgpreagg) ? ROWFLAG_ZONE_BOUNDED : 0u);
	for (cl_uint tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
	{
		cl_uint		tile_base = tile * GPUPREAGG_REG_TILE_ROWS;
		bool		full_tile = (tile_base + GPUPREAGG_REG_TILE_ROWS <= nitems);
		gpupreagg_column_tile T;

		if (full_tile && !any_nulls)
		{
#pragma unroll
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, false>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
			cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if (full_tile || row0 + j < nitems)
				{
					strom_kvars	KV;
#define X(attno,colidx,NAME)													\
					KV.KVAR_##attno = pg_##NAME##_make(T.v_##attno[k][j],			\
													   !((T.nn_##attno[k] >> j) & 1));
					STROM_KVAR_LIST(X)
#undef X
					cl_int		row_error = param_error;
					strom_kvars_from_column(KV, kds, &row_error);
					STROM_KVARS_FINISH(KV);
					gpupreagg_reg_row<NG>(S, ctl, KP, KV, row_error, &chunk_status, summag, rowflags);
				}
			}
		}
	}
	/* wave tree reduction, then lane 0 folds into the LDS image */
#pragma unroll
	for (int g = 0; g < NG; g++)
	{
		cl_uint	flags = S.flags[g];
#pragma unroll
		for (int m = 32; m > 0; m >>= 1)
			flags |= __shfl_xor(flags, m, STROM_WAVE);
#define X(aidx,resno,OP,NAME)															\
		{																				\
			cl_ulong v = S.v_##aidx[g];													\
			_Pragma("unroll")															\
			for (int m = 32; m > 0; m >>= 1)											\
			{																			\
				cl_ulong o = gpupreagg_shfl_xor_u64(v, m);								\
				v = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS ? v + o					\
					 : gpupreagg_merge8<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS			\
										? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,		\
										pg_##NAME##_base_t>(v, o));						\
			}																			\
			if (strom_lane_id() == 0 && (cl_uint)g < G)									\
			{																			\
				char *slot = lds + L.vals_off[aidx];									\
				if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)							\
					__hip_atomic_fetch_add((cl_uint *)slot + g, (cl_uint)v,				\
										   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
				else if (flags & (2u << aidx))											\
				{																		\
					if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PSUM)							\
					{																	\
						if (gpupreagg_is_float<pg_##NAME##_base_t>::value)				\
							__hip_atomic_fetch_add((cl_double *)slot + g,				\
												   __longlong_as_double((long long)v),	\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
						else															\
							__hip_atomic_fetch_add((cl_long *)slot + g, (cl_long)v,		\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
					}																	\
					else if (gpupreagg_is_float<pg_##NAME##_base_t>::value)				\
					{																	\
						if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN)						\
							__hip_atomic_fetch_min((cl_ulong *)slot + g, v,				\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
						else															\
							__hip_atomic_fetch_max((cl_ulong *)slot + g, v,				\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
					}																	\
					else																\
					{																	\
						if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN)						\
							__hip_atomic_fetch_min((cl_long *)slot + g, (cl_long)v,		\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
						else															\
							__hip_atomic_fetch_max((cl_long *)slot + g, (cl_long)v,		\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
					}																	\
				}																		\
			}																			\
		}
		GPUPREAGG_AGG_LIST(X)
#undef X
		if (strom_lane_id() == 0 && (cl_uint)g < G && flags != 0)
		{
			cl_uint *word = (cl_uint *)(lds + ((g * (cl_uint)sizeof(gpupreagg_flags_t)) & ~3u));
			cl_uint	 shift = ((g * (cl_uint)sizeof(gpupreagg_flags_t)) & 3u) * 8u;
			__hip_atomic_fetch_or(word, flags << shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	/* same slab format as the LDS kernels (block size differs: plain loops) */
	__syncthreads();
	{
		char *slab = slabs + (size_t)blockIdx.x * ctl->slab_bytes;
		for (cl_uint i = threadIdx.x * 4; i < L.total; i += GPUPREAGG_REG_BLOCK * 4)
			*(cl_uint *)(slab + i) = *(const cl_uint *)(lds + i);
	}
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
	gpupreagg_writeback_summag(kgpreagg, summag);
}

/* ====================================================================== *
 * lane-private LDS accumulators: 2 .. 32 dense ids
 *
 * Every thread owns one accumulator per (aggregate, group) in LDS at
 * [group][thread] -- a wave's lanes always hit 64 consecutive words
 * whatever their groups are (group stride = block size * width, a
 * multiple of the bank row), so a fold is a plain conflict-free
 * ds_read / op / ds_write: 2-8 LDS cycles per wave instruction against
 * the 24 / 53 / 110 of the u32 / u64 / f64 LDS atomics
 * (profiles/r01_preagg_atomics.txt).  "Seen" / "has a value" flags are
 * bit masks in registers.  At the end a wave tree-reduces its 64 entries
 * per (aggregate, group) with shuffles and lane 0 folds into the
 * work-group image, which leaves through the same slab / merge path.
 * ====================================================================== */
struct gpupreagg_priv_state {
	cl_uint		poff[GPUPREAGG_NAGGS + 1];
	cl_uint		has[GPUPREAGG_NAGGS + 1];
	cl_uint		seen;
};

STROM_DEVICE void
gpupreagg_priv_row(char *lds, gpupreagg_priv_state &S, const gpupreagg_dense_ctl *ctl,
				   const strom_kparams &KP, const strom_kvars &KV, cl_uint G,
				   cl_int param_error, cl_int *chunk_status, cl_ulong &summag, cl_uint rowflags)
{
	cl_int		errcode = param_error;
	pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KV);
	cl_uint		gid = 0;
	bool		out_of_domain = false;

	if (errcode == StromError_Success && !EVAL(rc))
		return;
#define X(kidx,resno,NAME)															\
	{																				\
		pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);					\
		cl_long		off64 = (cl_long)kv.value - ctl->key_min[kidx];					\
		cl_uint		range = ctl->key_range[kidx];									\
		cl_uint		off = (kv.isnull ? range : (cl_uint)off64);						\
		if (!kv.isnull && (off64 < 0 || off64 >= (cl_long)range))					\
			out_of_domain = true;													\
		gid += (kidx == 0 ? off : off * ctl->key_stride[kidx]);	/* (stride 0 is 1) */											\
	}
	GPUPREAGG_KEY_LIST(X)
#undef X
	if (!out_of_domain && !gpupreagg_remap_gid(ctl, gid))
		out_of_domain = true;
#define X(aidx,resno,OP,NAME)														\
	pg_##NAME##_t av_##aidx = gpupreagg_agg_##aidx(&errcode, KP, KV);
	GPUPREAGG_AGG_LIST(X)
#undef X
	if (errcode != StromError_Success)
	{
		STROM_SET_ERROR(chunk_status, errcode);
		return;
	}
	if (out_of_domain || gid >= G)
	{
		STROM_SET_ERROR(chunk_status, StromError_DataStoreOutOfRange);
		return;
	}
	cl_uint		bit = 1u << gid;
	cl_uint		idx = gid * GPUPREAGG_REG_BLOCK + threadIdx.x;

	S.seen |= bit;
	/* the slots are this thread's own: read them all, fold, write them all
	 * back -- three LDS reads in flight instead of three round trips */
#define X(aidx,resno,OP,NAME)														\
	cl_ulong	cur_##aidx = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS					\
		? (cl_ulong)((const cl_uint *)(lds + S.poff[aidx]))[idx]						\
		: ((const cl_ulong *)(lds + S.poff[aidx]))[idx]);
	GPUPREAGG_AGG_LIST(X)
#undef X
#define X(aidx,resno,OP,NAME)														\
	if (!av_##aidx.isnull)															\
	{																				\
		typedef pg_##NAME##_base_t base_t;											\
		if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)								\
			cur_##aidx += (cl_uint)av_##aidx.value;									\
		else																		\
		{																			\
			cl_ulong  x;															\
			if (gpupreagg_is_float<base_t>::value)									\
				x = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PSUM							\
					 ? (cl_ulong)__double_as_longlong((cl_double)av_##aidx.value)	\
					 : gpupreagg_f64_ordered((cl_double)av_##aidx.value));			\
			else																	\
			{																		\
				x = (cl_ulong)(cl_long)av_##aidx.value;								\
				if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, base_t>::value && GPUPREAGG_MEASURE_SUM(aidx, rowflags))	\
					summag |= gpupreagg_sum_magnitude((cl_long)x);					\
			}																		\
			cur_##aidx = gpupreagg_merge8<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS	\
										  ? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,	\
										  base_t>(cur_##aidx, x);					\
			S.has[aidx] |= bit;														\
		}																			\
	}
	GPUPREAGG_AGG_LIST(X)
#undef X
#define X(aidx,resno,OP,NAME)														\
	if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)									\
		((cl_uint *)(lds + S.poff[aidx]))[idx] = (cl_uint)cur_##aidx;				\
	else																			\
		((cl_ulong *)(lds + S.poff[aidx]))[idx] = cur_##aidx;
	GPUPREAGG_AGG_LIST(X)
#undef X
}

#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(GPUPREAGG_REG_BLOCK)
gpupreagg_priv_column(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
					  const gpupreagg_dense_ctl *ctl_in_memory, char *slabs)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	/* the control block by value (see gpupreagg_dense_column) */
	const gpupreagg_dense_ctl ctl_by_value = *ctl_in_memory;
	const gpupreagg_dense_ctl *ctl = &ctl_by_value;

	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
	cl_uint		nitems = kds->nitems;
	cl_uint		ntiles = (nitems + GPUPREAGG_REG_TILE_ROWS - 1) / GPUPREAGG_REG_TILE_ROWS;
	cl_uint		G = ctl->groups_per_split;		/* 2 .. 32, one split */
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	strom_kparams KP;
	gpupreagg_lds_layout L;
	gpupreagg_priv_state S;

	gpupreagg_load_kparams(KP, kparams, &param_error);
	gpupreagg_remap_init(ctl);
	gpupreagg_lds_layout_init(L, G, 1);
	/* work-group image, touched again only at the very end */
	for (cl_uint i = threadIdx.x * 16; i < L.total; i += GPUPREAGG_REG_BLOCK * 16)
		*(uint4 *)(lds + i) = make_uint4(0, 0, 0, 0);
	__syncthreads();
	S.seen = 0;
	{
		cl_uint	off = L.total;
#define X(aidx,resno,OP,NAME)															\
		S.poff[aidx] = strom_keep_in_vgpr(off);											\
		S.has[aidx] = 0;																\
		off += G * GPUPREAGG_REG_BLOCK * (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS ? 4u : 8u);	\
		if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN || GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMAX)	\
			for (cl_uint i = threadIdx.x; i < G; i += GPUPREAGG_REG_BLOCK)				\
				((cl_ulong *)(lds + L.vals_off[aidx]))[i] =								\
					gpupreagg_identity<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>();		\
		for (cl_uint g = 0; g < G; g++)													\
		{																				\
			if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)								\
				((cl_uint *)(lds + S.poff[aidx]))[g * GPUPREAGG_REG_BLOCK + threadIdx.x] = 0;	\
			else																		\
				((cl_ulong *)(lds + S.poff[aidx]))[g * GPUPREAGG_REG_BLOCK + threadIdx.x] =	\
					gpupreagg_identity<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS			\
									   ? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,			\
									   pg_##NAME##_base_t>();							\
		}
		GPUPREAGG_AGG_LIST(X)
#undef X
	}
	__syncthreads();
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
	const cl_uint rowflags = (KERN_GPUPREAGG_ZONE_BOUNDED(kgpreagg) ? ROWFLAG_ZONE_BOUNDED : 0u);
	for (cl_uint tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
	{
		cl_uint		tile_base = tile * GPUPREAGG_REG_TILE_ROWS;
		bool		full_tile = (tile_base + GPUPREAGG_REG_TILE_ROWS <= nitems);
		gpupreagg_column_tile T;

		if (full_tile && !any_nulls)
		{
#pragma unroll
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < GPUPREAGG_QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, false>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
#pragma unroll
		for (int k = 0; k < GPUPREAGG_QUADS; k++)
		{
			cl_uint	row0 = tile_base + (k * GPUPREAGG_REG_BLOCK + threadIdx.x) * 4;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if (full_tile || row0 + j < nitems)
				{
					strom_kvars	KV;
#define X(attno,colidx,NAME)													\
					KV.KVAR_##attno = pg_##NAME##_make(T.v_##attno[k][j],			\
													   !((T.nn_##attno[k] >> j) & 1));
					STROM_KVAR_LIST(X)
#undef X
					cl_int		row_error = param_error;
					strom_kvars_from_column(KV, kds, &row_error);
					STROM_KVARS_FINISH(KV);
					gpupreagg_priv_row(lds, S, ctl, KP, KV, G, row_error, &chunk_status, summag, rowflags);
				}
			}
		}
	}
	/* wave tree reduction of the lane-private entries, lane 0 folds into
	 * the work-group image */
	cl_uint		wave_seen = S.seen;
#pragma unroll
	for (int m = 32; m > 0; m >>= 1)
		wave_seen |= __shfl_xor(wave_seen, m, STROM_WAVE);
#define X(aidx,resno,OP,NAME)															\
	{																					\
		_Pragma("unroll")																\
		for (int m = 32; m > 0; m >>= 1)												\
			S.has[aidx] |= __shfl_xor(S.has[aidx], m, STROM_WAVE);						\
	}
	GPUPREAGG_AGG_LIST(X)
#undef X
	for (cl_uint g = 0; g < G; g++)
	{
		if (!((wave_seen >> g) & 1))
			continue;					/* wave-uniform */
		cl_uint		idx = g * GPUPREAGG_REG_BLOCK + threadIdx.x;
		cl_uint		flags = GPUPREAGG_FLAG_SEEN;
#define X(aidx,resno,OP,NAME)															\
		{																				\
			cl_ulong v = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS						\
						  ? (cl_ulong)((const cl_uint *)(lds + S.poff[aidx]))[idx]		\
						  : ((const cl_ulong *)(lds + S.poff[aidx]))[idx]);				\
			_Pragma("unroll")															\
			for (int m = 32; m > 0; m >>= 1)											\
			{																			\
				cl_ulong o = gpupreagg_shfl_xor_u64(v, m);								\
				v = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS ? v + o					\
					 : gpupreagg_merge8<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS			\
										? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,		\
										pg_##NAME##_base_t>(v, o));						\
			}																			\
			bool has = (GPUPREAGG_OP_##OP != GPUPREAGG_OP_NROWS && ((S.has[aidx] >> g) & 1));	\
			if (has)																	\
				flags |= (2u << aidx);													\
			if (strom_lane_id() == 0)													\
			{																			\
				char *slot = lds + L.vals_off[aidx];									\
				if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)							\
					__hip_atomic_fetch_add((cl_uint *)slot + g, (cl_uint)v,				\
										   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
				else if (has)															\
				{																		\
					if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PSUM)							\
					{																	\
						if (gpupreagg_is_float<pg_##NAME##_base_t>::value)				\
							__hip_atomic_fetch_add((cl_double *)slot + g,				\
												   __longlong_as_double((long long)v),	\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
						else															\
							__hip_atomic_fetch_add((cl_long *)slot + g, (cl_long)v,		\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
					}																	\
					else if (gpupreagg_is_float<pg_##NAME##_base_t>::value)				\
					{																	\
						if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN)						\
							__hip_atomic_fetch_min((cl_ulong *)slot + g, v,				\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
						else															\
							__hip_atomic_fetch_max((cl_ulong *)slot + g, v,				\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
					}																	\
					else																\
					{																	\
						if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN)						\
							__hip_atomic_fetch_min((cl_long *)slot + g, (cl_long)v,		\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
						else															\
							__hip_atomic_fetch_max((cl_long *)slot + g, (cl_long)v,		\
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);	\
					}																	\
				}																		\
			}																			\
		}
		GPUPREAGG_AGG_LIST(X)
#undef X
		if (strom_lane_id() == 0)
		{
			cl_uint *word = (cl_uint *)(lds + ((g * (cl_uint)sizeof(gpupreagg_flags_t)) & ~3u));
			cl_uint	 shift = ((g * (cl_uint)sizeof(gpupreagg_flags_t)) & 3u) * 8u;
			__hip_atomic_fetch_or(word, flags << shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	__syncthreads();
	{
		char *slab = slabs + (size_t)blockIdx.x * ctl->slab_bytes;
		for (cl_uint i = threadIdx.x * 4; i < L.total; i += GPUPREAGG_REG_BLOCK * 4)
			*(cl_uint *)(slab + i) = *(const cl_uint *)(lds + i);
	}
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
	gpupreagg_writeback_summag(kgpreagg, summag);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */

#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(GPUPREAGG_REG_BLOCK)
gpupreagg_reg1_column(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
					  const gpupreagg_dense_ctl *ctl_in_memory, char *slabs)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	/* the control block by value (see gpupreagg_dense_column) */
	const gpupreagg_dense_ctl ctl_by_value = *ctl_in_memory;
	const gpupreagg_dense_ctl *ctl = &ctl_by_value;
	gpupreagg_reg_kernel_body<1>(kgpreagg, kds, ctl, slabs, lds);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */

#endif	/* !GPUPREAGG_HASHED */

#ifdef GPUPREAGG_HASHED
/* ====================================================================== *
 * hashed GROUP BY: any key type, any key range
 *
 * The dense-id kernels need integer-like keys whose ranges multiply to at
 * most 2^26 ids.  Everything else -- float8 / numeric keys, sparse int8
 * keys -- goes through an open-addressing table in HBM keyed by the keys'
 * canonical 64-bit images (the reference sorts row indexes by
 * gpupreagg_keycomp instead, opencl_gpupreagg.h:620-856).  One slot is one
 * record, so a probe touches one cache line:
 *
 *   +0   state  u32     0 empty, 1 being claimed, 2 ready
 *   +4   knull  u32     bit k: key k is NULL
 *   +8   flags  u32     bit 0 seen, bit 1+a aggregate a has a value
 *   +16  keys[NKEYS]    u64 images
 *        vals[NAGGS]    8 bytes each (NROWS widened to i64, float min/max as
 *                       order-preserving keys, like the dense table)
 *
 * records start GPUPREAGG_HASH_HEAD bytes into the table, after the head.
 *
 * A chunk is folded in two launches: gpupreagg_hash_check evaluates every
 * row for errors only; gpupreagg_hash_fold runs if the chunk is clean, so a
 * CpuReCheck still sends the WHOLE chunk back untouched (gpupreagg.c:2746-2750).
 * The fold keeps a small table of the same shape in LDS in front of the
 * global one: rows of keys that found room there cost LDS atomics only and
 * reach HBM once per work-group, the others go to the global table row by
 * row.
 * ====================================================================== */
#define GPUPREAGG_HASH_HEAD		256
#define GPUPREAGG_HASH_RECLEN	(16 + 8 * (GPUPREAGG_NKEYS + GPUPREAGG_NAGGS))
#define GPUPREAGG_HASH_STRIDE	(GPUPREAGG_HASH_RECLEN <= 32 ? 32 :					\
								 GPUPREAGG_HASH_RECLEN <= 64 ? 64 :					\
								 ((GPUPREAGG_HASH_RECLEN + 127) / 128 * 128))
#ifndef GPUPREAGG_HASH_UNROLL
#define GPUPREAGG_HASH_UNROLL	2		/* 4 needs 66 VGPRs: the second work-group of a CU no longer fits */
#endif
/* queued row numbers per wave (roles): a power of two, >= 64 * (UNROLL + 1) */
#define GPUPREAGG_HASH_QUEUE	256
/* role map (one byte per row, written by the check pass): low 6 bits = the row's role among
 * up to 64; this value = no role folds the row (the qual dropped it, or padding) */
#define GPUPREAGG_ROLE_NONE		0xffu
#ifndef GPUPREAGG_HASH_LDS_PROBES
#define GPUPREAGG_HASH_LDS_PROBES	64	/* a key that finds no room in LDS sends ALL its rows to one
										 * global record: same-address atomics, to be avoided */
#endif

static_assert(GPUPREAGG_HASH_QUEUE >= 64 * (GPUPREAGG_HASH_UNROLL + 1) &&
			  (GPUPREAGG_HASH_QUEUE & (GPUPREAGG_HASH_QUEUE - 1)) == 0, "role queue too small for the tile");

struct gpupreagg_hash_head {
	cl_uint		capacity;			/* power of two */
	cl_uint		nkeys;
	cl_uint		ngroups;			/* slots claimed so far */
	cl_uint		overflow;			/* set when a probe found no free slot */
	cl_uint		stride;				/* GPUPREAGG_HASH_STRIDE, checked by the host */
	cl_uint		naggs;
	cl_uint		__pad[2];
	/*
	 * integer sums never wrap ("integer sums never wrap", above -- here for a table whose
	 * accumulators are 64 bits wide and are updated by atomics all over the chip): an upper
	 * bound of |any partial sum in this table|, the sum over the folded chunks of
	 * rows x 2^(bits of the largest input magnitude).  While it stays below 2^63 nothing can have
	 * wrapped and nothing is checked.  Two slots: the fold of chunk k reads slot k & 1 and
	 * (its first work-group) writes the other, which the fold of chunk k + 1 reads -- no
	 * work-group of a launch reads what another one of it writes.
	 */
	cl_ulong	sum_bound[2];
};

/*
 * a fold is about to add the request's rows to the table: is the bound still below
 * 2^63 afterwards?  No: nothing is folded; the status becomes SumRangeUnproven (the host
 * measures the table's true largest |sum| -- gpupreagg_hash_sum_refresh -- and sends the
 * chunk once more), or CpuReCheck on that second attempt (turn bit 2).  turn bit 1: a
 * relaunch for rows the table had no room for -- the chunk is accounted already.
 */
STROM_DEVICE bool
gpupreagg_hash_sum_account(kern_gpupreagg *kgpreagg, gpupreagg_hash_head *head, cl_uint sum_turn)
{
	if (gpupreagg_intsum_index(GPUPREAGG_NAGGS) == 0 || (sum_turn & 2u) != 0)
		return true;
#if defined(GPUPREAGG_CHECKED) && GPUPREAGG_CHECKED
	return true;				/* every addition is checked one by one: no proof needed */
#endif
	cl_ulong	prev = head->sum_bound[sum_turn & 1u];
	cl_ulong	add = gpupreagg_sum_bound(KERN_GPUPREAGG_FOLD_NROWS(kgpreagg), *KERN_GPUPREAGG_SUM_MAGBITS(kgpreagg));
	bool		ok = (prev < (1UL << 63) && add < (1UL << 63) && prev + add < (1UL << 63));
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		head->sum_bound[(sum_turn & 1u) ^ 1u] = (ok ? prev + add : prev);
		if (!ok)
			atomicMax(&kgpreagg->status, (sum_turn & 4u) ? StromError_CpuReCheck : StromError_SumRangeUnproven);
	}
	return ok;
}

STROM_DEVICE char *gpupreagg_hash_rec(char *htab, cl_uint slot)
{ return htab + GPUPREAGG_HASH_HEAD + (size_t)slot * GPUPREAGG_HASH_STRIDE; }
STROM_DEVICE const char *gpupreagg_hash_rec(const char *htab, cl_uint slot)
{ return htab + GPUPREAGG_HASH_HEAD + (size_t)slot * GPUPREAGG_HASH_STRIDE; }
#define HASH_REC_STATE(rec)		((cl_uint *)(rec))
#define HASH_REC_KNULL(rec)		((cl_uint *)((rec) + 4))
#define HASH_REC_FLAGS(rec)		((cl_uint *)((rec) + 8))
#define HASH_REC_KEYS(rec)		((cl_ulong *)((rec) + 16))
#define HASH_REC_VALS(rec)		((cl_ulong *)((rec) + 16) + GPUPREAGG_NKEYS)

STROM_DEVICE cl_ulong strom_key_image(cl_char v)	{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong strom_key_image(cl_short v)	{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong strom_key_image(cl_int v)		{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong strom_key_image(cl_long v)	{ return (cl_ulong)v; }
STROM_DEVICE cl_ulong strom_key_image(cl_ulong v)	{ return v; }	/* numeric: canonical image */
STROM_DEVICE cl_ulong strom_key_image(cl_double v)
{
	if (__builtin_isnan(v))
		return 0x7ff8000000000000UL;		/* all NaNs are one group, as in PostgreSQL */
	if (v == 0.0)
		v = 0.0;							/* -0 = +0 */
	return (cl_ulong)__double_as_longlong(v);
}
STROM_DEVICE cl_ulong strom_key_image(cl_float v)	{ return strom_key_image((cl_double)v); }

/* 32-bit multiplies only: a 64-bit product is four quarter-rate VALU operations
 * on gfx950, and with hash roles every row is hashed once per role */
STROM_DEVICE cl_uint
gpupreagg_hash_of(const cl_ulong *kimg, cl_uint knull)
{
	cl_uint		h = 0x9e3779b9u ^ knull;
	for (int k = 0; k < GPUPREAGG_NKEYS; k++)
	{
		h = (h ^ (cl_uint)kimg[k]) * 0x85ebca6bu;
		h ^= h >> 15;
		h = (h ^ (cl_uint)(kimg[k] >> 32)) * 0xc2b2ae35u;
		h ^= h >> 13;
	}
	h *= 0x27d4eb2fu;
	h ^= h >> 16;
	return h;
}

/*
 * find the slot of a key, or claim an empty one.  One loop, no inner wait:
 * a lane that meets a slot somebody is still filling (state 1) just goes
 * round again, so the claimer -- possibly a lane of the SAME wave, whose
 * divergent block runs before or after ours but within this iteration --
 * always gets to publish.  Returns the slot, GPUPREAGG_HASH_FULL, or -- for a
 * key that is not in the table while claim_limit groups are -- _DEFER: the
 * host then grows the table and folds the deferred rows again.
 */
#define GPUPREAGG_HASH_FULL		(~0u)
#define GPUPREAGG_HASH_DEFER	(~1u)		/* a new group, but claim_limit groups exist: not now */

template <bool MATCH>
STROM_DEVICE cl_uint
gpupreagg_hash_slot(char *htab, cl_uint hash, const cl_ulong *kimg, cl_uint knull, cl_uint claim_limit)
{
	gpupreagg_hash_head *head = (gpupreagg_hash_head *)htab;
	cl_uint		mask = head->capacity - 1;
	cl_uint		slot = hash & mask;
	cl_uint		probes = 0;
	cl_uint		result = ~0u;
	bool		done = false;

	for (cl_uint turns = 0; !done && turns < (1u << 28); turns++)
	{
		char   *rec = gpupreagg_hash_rec(htab, slot);
		/*
		 * No acquire here: at agent scope that is an invalidate of the caches in front of
		 * the table for every probe (the flush of a work-group's LDS table spent 2.9 of
		 * its 3.6 ms per 1e8 rows and 1e6 groups there).  The record's state, NULL bits
		 * and keys are read with agent-scope atomic loads instead, served at the coherence
		 * point -- where the claimer's key stores had been acknowledged before it stored
		 * state 2 (STROM_PUBLISH_STATE / STROM_PROBE_STATE, strom_common.h).
		 */
		cl_uint	st = STROM_PROBE_STATE(HASH_REC_STATE(rec));

		if (st == 0 && claim_limit != ~0u &&
			__hip_atomic_load(&head->ngroups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= claim_limit)
		{
			result = GPUPREAGG_HASH_DEFER;
			done = true;
		}
		else if (st == 0)
		{
			cl_uint	expect = 0;
			if (__hip_atomic_compare_exchange_strong(HASH_REC_STATE(rec), &expect, 1u,
													 __ATOMIC_RELAXED, __ATOMIC_RELAXED,
													 __HIP_MEMORY_SCOPE_AGENT))
			{
				/*
				 * publish: the keys and NULL bits go out as agent-scope atomic stores (written
				 * through to where the probes' agent-scope loads read), the wave waits for
				 * them to be acknowledged, then the state follows.  An agent-scope RELEASE
				 * here is a write-back of the XCD's whole L2 per claim: a first chunk with 1e6
				 * new groups spent 5 of its 7 ms in it (profiles/r02_hashed_fetch.txt).
				 */
				for (int k = 0; k < GPUPREAGG_NKEYS; k++)
					__hip_atomic_store(HASH_REC_KEYS(rec) + k, kimg[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(HASH_REC_KNULL(rec), knull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				STROM_PUBLISH_STATE(HASH_REC_STATE(rec), 2u);	/* behind the stores above; no cache maintenance */
				atomicAdd(&head->ngroups, 1u);
				result = slot;
				done = true;
			}
			/* lost the race: look at the same slot again next turn */
		}
		else if (st == 2)
		{
			bool	same = (MATCH && __hip_atomic_load(HASH_REC_KNULL(rec), __ATOMIC_RELAXED,
													   __HIP_MEMORY_SCOPE_AGENT) == knull);
			if (MATCH)
			{
				for (int k = 0; k < GPUPREAGG_NKEYS; k++)
					same = same && (__hip_atomic_load(HASH_REC_KEYS(rec) + k, __ATOMIC_RELAXED,
													  __HIP_MEMORY_SCOPE_AGENT) == kimg[k]);
			}
			/* the keys were read after the state they belong to?  (see strom_common.h: a
			 * ready record never changes, so this re-read only ever confirms) */
			if (MATCH && STROM_PROBE_STATE(HASH_REC_STATE(rec)) != 2u)
				continue;
			if (same)
			{
				result = slot;
				done = true;
			}
			else if (++probes > mask)
				done = true;			/* every slot taken by other keys */
			else
				slot = (slot + 1) & mask;
		}
		else
			__builtin_amdgcn_s_sleep(1);	/* being filled: next turn */
	}
	return result;
}

/* the same search in the work-group's LDS table, a few probes only */
struct gpupreagg_hash_lds {
	cl_uint	   *state;
	cl_uint	   *knull;
	cl_ulong   *keys;				/* [slots * NKEYS] */
	cl_uint		mask;
	cl_uint		shift;				/* 32 - log2(slots) */
};

STROM_DEVICE cl_uint
gpupreagg_hash_lds_slot(const gpupreagg_hash_lds &T, cl_uint hash, const cl_ulong *kimg, cl_uint knull)
{
	cl_uint		slot = (hash * 0x9e3779b1u) >> T.shift;
	cl_uint		probes = 0;
	cl_uint		result = ~0u;
	bool		done = false;

	for (cl_uint turns = 0; !done && turns < (1u << 24); turns++)
	{
		cl_uint	st = __hip_atomic_load(&T.state[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);

		if (st == 0)
		{
			cl_uint	expect = 0;
			if (__hip_atomic_compare_exchange_strong(&T.state[slot], &expect, 1u,
													 __ATOMIC_ACQUIRE, __ATOMIC_ACQUIRE,
													 __HIP_MEMORY_SCOPE_WORKGROUP))
			{
				for (int k = 0; k < GPUPREAGG_NKEYS; k++)
					T.keys[slot * GPUPREAGG_NKEYS + k] = kimg[k];
				T.knull[slot] = knull;
				__hip_atomic_store(&T.state[slot], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
				result = slot;
				done = true;
			}
		}
		else if (st == 2)
		{
			bool	same = (T.knull[slot] == knull);
			for (int k = 0; k < GPUPREAGG_NKEYS; k++)
				same = same && (T.keys[slot * GPUPREAGG_NKEYS + k] == kimg[k]);
			if (same)
			{
				result = slot;
				done = true;
			}
			else if (++probes >= GPUPREAGG_HASH_LDS_PROBES)
				done = true;			/* no room nearby: this row goes to the global table */
			else
				slot = (slot + 1) & T.mask;
		}
	}
	return result;
}

/* merge one 8-byte value (in its stored form) into a global accumulator */
template <int OP, typename BASE>
STROM_DEVICE void
gpupreagg_hash_merge8(cl_ulong *addr, cl_ulong x, cl_int *chunk_status = NULL)
{
#if defined(GPUPREAGG_CHECKED) && GPUPREAGG_CHECKED
	/* the exact fold of a chunk whose integer sums' range could not be proven (gpupreagg.cpp:
	 * gpupreagg_hashed_exact): every link of a group's chain of additions is checked against
	 * what it was added to, as in gpupreagg_lds_accum */
	if (OP == GPUPREAGG_OP_PSUM && !gpupreagg_is_float<BASE>::value && chunk_status != NULL)
	{
		cl_long		sum;
		cl_long		old = __hip_atomic_fetch_add((cl_long *)addr, (cl_long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		STROM_SET_RECHECK_IF(chunk_status, __builtin_add_overflow(old, (cl_long)x, &sum));
		return;
	}
#endif
	if (OP == GPUPREAGG_OP_NROWS || (OP == GPUPREAGG_OP_PSUM && !gpupreagg_is_float<BASE>::value))
		__hip_atomic_fetch_add(addr, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (OP == GPUPREAGG_OP_PSUM)
		__hip_atomic_fetch_add((cl_double *)addr, __longlong_as_double((long long)x),
							   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (gpupreagg_is_float<BASE>::value)
	{
		if (OP == GPUPREAGG_OP_PMIN)
			__hip_atomic_fetch_min(addr, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else
			__hip_atomic_fetch_max(addr, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	else
	{
		if (OP == GPUPREAGG_OP_PMIN)
			__hip_atomic_fetch_min((cl_long *)addr, (cl_long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else
			__hip_atomic_fetch_max((cl_long *)addr, (cl_long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

template <bool IS_COLUMN, bool FOLD, bool ROLES>
__device__ __forceinline__ void
gpupreagg_hash_body(kern_gpupreagg *kgpreagg,
					const kern_data_store *kds,
					const kern_data_store *ktoast,
					const kern_row_map *krowmap,
					char *htab, cl_uint claim_limit, kern_row_map *deferred,
					cl_uint lds_slots, cl_uint nroles, char *lds, cl_uchar *rolemap, cl_uint sum_turn)
{
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	gpupreagg_hash_head *head = (gpupreagg_hash_head *)htab;
	bool		use_map = (krowmap != NULL && krowmap->nvalids >= 0);
	cl_uint		nrows = (use_map ? (cl_uint)krowmap->nvalids : kds->nitems);
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	strom_kparams KP;
	gpupreagg_lds_layout L;
	gpupreagg_hash_lds T;

	if (FOLD && kgpreagg->status != StromError_Success)
		return;							/* the check pass found a reason to send the chunk back */
	if (FOLD && !gpupreagg_hash_sum_account(kgpreagg, head, sum_turn))
		return;							/* an integer sum could leave int8: nothing is folded */
	gpupreagg_load_kparams(KP, kparams, &param_error);
	if (FOLD)
	{
		/* LDS: the dense kernels' image for lds_slots slots, then the hash part */
		gpupreagg_lds_layout_init(L, lds_slots, 1);
		T.state = (cl_uint *)(lds + L.total);
		T.knull = T.state + lds_slots;
		T.keys = (cl_ulong *)(T.knull + lds_slots);
		T.mask = lds_slots - 1;
		T.shift = 32 - (31 - __clz((int)lds_slots));
		gpupreagg_lds_init(lds, L, lds_slots, 1);
		for (cl_uint i = threadIdx.x; i < lds_slots; i += blockDim.x)
			T.state[i] = 0;
		__syncthreads();
	}
	const bool	is_column = IS_COLUMN;
	const cl_int chunk_format = kds->format;
	const bool	row_family = (chunk_format == KDS_FORMAT_ROW || chunk_format == KDS_FORMAT_ROW_FLAT);
	const kern_coldir *coldir_g = KERN_DATA_STORE_COLDIR(kds);
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (is_column ? (const char *)kds + coldir_g[colidx].values_off : NULL);	\
	const cl_uint *nul_##attno = ((is_column && coldir_g[colidx].nulls_off != 0)	\
		? (const cl_uint *)((const char *)kds + coldir_g[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	/*
	 * one row, its columns loaded: qual, keys, partial inputs, then into the
	 * work-group's LDS table or, without room there, the global one
	 */
	auto fold_loaded = [&](const strom_kvars &KV, cl_int errcode, cl_uint kds_index, size_t pos)
	{
		cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
		cl_uint		knull = 0;
		pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KV);
		if (errcode == StromError_Success && !EVAL(rc))
		{
			if (!FOLD && rolemap != NULL)
				rolemap[pos] = GPUPREAGG_ROLE_NONE;		/* filtered: no role folds it */
			return;
		}
#define X(kidx,resno,NAME)															\
		{																			\
			pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);				\
			kimg[kidx] = (kv.isnull ? 0UL : strom_key_image(kv.value));				\
			knull |= (kv.isnull ? (1u << kidx) : 0u);								\
		}
		GPUPREAGG_KEY_LIST(X)
#undef X
#define X(aidx,resno,OP,NAME)														\
		pg_##NAME##_t av_##aidx = gpupreagg_agg_##aidx(&errcode, KP, KV);
		GPUPREAGG_AGG_LIST(X)
#undef X
		if (errcode != StromError_Success)
		{
			STROM_SET_ERROR(&chunk_status, errcode);
			return;
		}
		if (!FOLD)
		{
			/* integer sums without a static bound: the check pass measures their inputs */
#define X(aidx,resno,OP,NAME)														\
			if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value && GPUPREAGG_SUMBITS_##aidx >= 64)	\
				summag |= (av_##aidx.isnull ? 0UL : gpupreagg_sum_magnitude((cl_long)av_##aidx.value));
			GPUPREAGG_AGG_LIST(X)
#undef X
			/* the check pass has the row's hash at hand: leave its role (6 bits: up to 64
			 * roles) for the fold's role scans, which then read one byte per row instead
			 * of the qual's and the keys' columns */
			if (rolemap != NULL)
				rolemap[pos] = (cl_uchar)((gpupreagg_hash_of(kimg, knull) >> 7) & 63u);
			return;
		}
		cl_uint		hash = gpupreagg_hash_of(kimg, knull);
		cl_uint		lslot = gpupreagg_hash_lds_slot(T, hash, kimg, knull);
		cl_uint		need = GPUPREAGG_FLAG_SEEN;
		if (lslot != ~0u)
		{
			/* the work-group's own table: LDS atomics, as in gpupreagg_dense_row */
#define X(aidx,resno,OP,NAME)														\
			need |= gpupreagg_lds_accum<GPUPREAGG_OP_##OP, aidx>(lds, L.vals_off[aidx], lslot, av_##aidx, &chunk_status);
			GPUPREAGG_AGG_LIST(X)
#undef X
			gpupreagg_flags_t *flags = (gpupreagg_flags_t *)lds;
			if ((flags[lslot] & need) != need)
			{
				cl_uint *word = (cl_uint *)(lds + ((lslot * (cl_uint)sizeof(gpupreagg_flags_t)) & ~3u));
				cl_uint	 shift = ((lslot * (cl_uint)sizeof(gpupreagg_flags_t)) & 3u) * 8u;
				__hip_atomic_fetch_or(word, need << shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			return;
		}
		/* no room in LDS: straight to the global table */
		cl_uint		slot = gpupreagg_hash_slot<true>(htab, hash, kimg, knull, claim_limit);
		if (slot == GPUPREAGG_HASH_DEFER)
		{
			/* a new group and the table is at its fill limit: the host grows it
			 * and sends this row again */
			cl_uint	idx = atomicAdd((cl_uint *)&deferred->nvalids, 1u);
			deferred->rindex[idx] = (cl_int)kds_index;
			return;
		}
		if (slot == GPUPREAGG_HASH_FULL)
		{
			head->overflow = 1;
			return;
		}
		char	   *rec = gpupreagg_hash_rec(htab, slot);
#define X(aidx,resno,OP,NAME)														\
		{																			\
			typedef pg_##NAME##_base_t base_t;										\
			bool		has = !av_##aidx.isnull;									\
			cl_ulong	x;															\
			if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)							\
				x = (has ? (cl_ulong)(cl_uint)av_##aidx.value : 0);					\
			else if (gpupreagg_is_float<base_t>::value)								\
				x = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PSUM							\
					 ? (cl_ulong)__double_as_longlong((cl_double)av_##aidx.value)	\
					 : gpupreagg_f64_ordered((cl_double)av_##aidx.value));			\
			else																	\
				x = (cl_ulong)(cl_long)av_##aidx.value;								\
			if (GPUPREAGG_OP_##OP != GPUPREAGG_OP_NROWS && has)						\
				need |= (2u << aidx);												\
			if (has && !(GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS && x == 0))		\
				gpupreagg_hash_merge8<GPUPREAGG_OP_##OP, base_t>(HASH_REC_VALS(rec) + aidx, x, &chunk_status);	\
		}
		GPUPREAGG_AGG_LIST(X)
#undef X
		if ((*HASH_REC_FLAGS(rec) & need) != need)
			atomicOr(HASH_REC_FLAGS(rec), need);
	};

	/*
	 * who folds what.  More groups than one LDS table holds: the work-groups
	 * take ROLES (COLUMN chunks), role j folds the keys whose hash says j, so
	 * each role's groups fit its LDS table again (the dense kernels split
	 * their id range the same way).  Work-group i runs on XCD i % 8: the tiles
	 * are dealt to the XCDs, and inside an XCD all roles walk the same tiles,
	 * so the repeated reads meet in that XCD's L2.  gridDim.x is a multiple of
	 * 8 * nroles (nroles a power of two; 1 without ROLES).
	 */
	cl_uint		xcd = blockIdx.x & 7;
	cl_uint		local = blockIdx.x >> 3;
	cl_uint		role = local & (nroles - 1);
	cl_uint		member = local / nroles;
	cl_uint		nmembers = (gridDim.x >> 3) / nroles;
	/*
	 * with roles a wave SCANS its tile -- the columns the qual and the keys
	 * read, hash, whose row is it -- and only queues its own rows' numbers in
	 * LDS; whenever 64 are queued the whole wave folds them.  Folding straight
	 * from the scan cost a full pass per role (0.3 ms per 1e8 rows, whatever
	 * was loaded): the wave waited for the few lanes that had a row of its
	 * role to get through their LDS atomics.
	 */
	cl_uint	   *queue = NULL;
	cl_uint		qhead = 0, qtail = 0;		/* wave-uniform */
	if (ROLES)
		queue = (cl_uint *)(T.keys + (size_t)lds_slots * GPUPREAGG_NKEYS)
			+ (threadIdx.x / STROM_WAVE) * GPUPREAGG_HASH_QUEUE;
	auto drain = [&](cl_uint nready)
	{
		/* the first nready (<= 64) queued rows, one per lane */
		bool		active = (strom_lane_id() < nready);
		cl_uint		kds_index = queue[(qhead + (active ? strom_lane_id() : 0)) & (GPUPREAGG_HASH_QUEUE - 1)];
		strom_kvars	KV;
#define X(attno,colidx,NAME)													\
		KV.KVAR_##attno = STROM_COLUMN_REF_CACHED(NAME, col_##attno, nul_##attno, kds_index);
		STROM_KVAR_LIST_GROUPING(X)
#undef X
#define X(attno,colidx,NAME)													\
		KV.KVAR_##attno = STROM_COLUMN_REF(NAME, col_##attno, nul_##attno, kds_index);
		STROM_KVAR_LIST_REST(X)
#undef X
		STROM_KVARS_FINISH(KV);
		if (active)
			fold_loaded(KV, param_error, kds_index, 0);
		qhead += nready;
	};
	/*
	 * ROLES with a role map (the check pass wrote one byte per row: the row's role, or
	 * GPUPREAGG_ROLE_NONE for a row the qual drops): the scan is 16 bytes per lane and
	 * load and a compare per row -- 0.1 GB per 1e8 rows and role instead of the qual's and
	 * keys' columns plus a hash per row.  The host pads the map with ROLE_NONE to whole
	 * tiles; row maps and deferred rows take the scan below.
	 */
	if (ROLES && rolemap != NULL)
	{
		/*
		 * a tile is 16 x blockDim rows; in turn j a wave looks at 64 CONSECUTIVE rows (one
		 * byte per lane), so the rows it queues are neighbours and the loads that fetch
		 * their columns in drain() share cache lines.  (16 bytes per lane in one load put
		 * rows 16 apart next to each other in the queue: every queued row then cost a
		 * line of its own per column -- 3000 groups, two roles: 1.8 -> 2.8 ms.)
		 */
		size_t		tile_rows = (size_t)16 * blockDim.x;
		for (size_t tile = xcd + 8 * (size_t)member;
			 tile * tile_rows < nrows;
			 tile += 8 * (size_t)nmembers)
		{
			size_t		r0 = tile * tile_rows + threadIdx.x;
			cl_uint		b[16];
			/* (the map is padded to whole tiles: no bounds test on the loads) */
#pragma unroll
			for (int j = 0; j < 16; j++)
				b[j] = rolemap[r0 + (size_t)j * blockDim.x];
#pragma unroll
			for (int j = 0; j < 16; j++)
			{
				bool		own = (b[j] != GPUPREAGG_ROLE_NONE && (b[j] & (nroles - 1)) == role);
				cl_ulong	mask = __ballot(own);
				if (own)
					queue[(qtail + (cl_uint)__popcll(mask & ((1UL << strom_lane_id()) - 1)))
						  & (GPUPREAGG_HASH_QUEUE - 1)] = (cl_uint)(r0 + (size_t)j * blockDim.x);
				qtail += (cl_uint)__popcll(mask);
				__builtin_amdgcn_wave_barrier();
				while (qtail - qhead >= STROM_WAVE)
					drain(STROM_WAVE);
			}
		}
	}
	else
	/*
	 * a tile is GPUPREAGG_HASH_UNROLL x blockDim rows: a thread first loads
	 * its rows of the tile, then works on them one by one (with one row per
	 * turn the loop is bound by the latency of that one load)
	 */
	for (size_t tile = xcd + 8 * (size_t)member;
		 tile * GPUPREAGG_HASH_UNROLL * blockDim.x < nrows;
		 tile += 8 * (size_t)nmembers)
	{
		strom_kvars	KVs[GPUPREAGG_HASH_UNROLL] = {};	/* (with roles the scan fills the grouping columns only) */
		cl_int		errs[GPUPREAGG_HASH_UNROLL];
		cl_uint		kidx[GPUPREAGG_HASH_UNROLL];
		bool		live[GPUPREAGG_HASH_UNROLL];
#pragma unroll
		for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
		{
			size_t		r = (tile * GPUPREAGG_HASH_UNROLL + j) * blockDim.x + threadIdx.x;
			live[j] = (r < nrows);
			if (!live[j])
				r = 0;					/* nrows > 0 here: a harmless row, ignored below */
			cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : (cl_uint)r);
			cl_int		errcode = param_error;
			const HeapTupleHeaderData *htup = NULL;
			if (!is_column && row_family)
				htup = strom_locate_tuple(kds, chunk_format, kds_index);
#define X(attno,colidx,NAME)													\
			KVs[j].KVAR_##attno = (is_column									\
				? (ROLES ? STROM_COLUMN_REF_CACHED(NAME, col_##attno, nul_##attno, kds_index)	\
						 : STROM_COLUMN_REF(NAME, col_##attno, nul_##attno, kds_index))	\
				: row_family ? STROM_TUPLE_REF(NAME, kds, htup, colidx)			\
				: pg_##NAME##_vref(kds, ktoast, &errcode, colidx, kds_index));
			STROM_KVAR_LIST_GROUPING(X)
			if (!ROLES)
			{
				STROM_KVAR_LIST_REST(X)
			}
#undef X
			if (is_column && !ROLES)
				strom_kvars_from_column(KVs[j], kds, &errcode);
			STROM_KVARS_FINISH(KVs[j]);
			errs[j] = errcode;
			kidx[j] = kds_index;
		}
#pragma unroll
		for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
		{
			if (!ROLES)
			{
				if (live[j])
					fold_loaded(KVs[j], errs[j], kidx[j],
								(tile * GPUPREAGG_HASH_UNROLL + j) * blockDim.x + threadIdx.x);
				continue;
			}
			/* scan: is this row the role's?  (every lane stays for the ballot) */
			bool		own = false;
			if (live[j])
			{
				cl_int		errcode = errs[j];
				pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KVs[j]);
				if (!(errcode == StromError_Success && !EVAL(rc)))
				{
					cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
					cl_uint		knull = 0;
#define X(kidx,resno,NAME)															\
					{																\
						pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KVs[j]);	\
						kimg[kidx] = (kv.isnull ? 0UL : strom_key_image(kv.value));	\
						knull |= (kv.isnull ? (1u << kidx) : 0u);					\
					}
					GPUPREAGG_KEY_LIST(X)
#undef X
					own = (errcode != StromError_Success ||
						   ((gpupreagg_hash_of(kimg, knull) >> 7) & (nroles - 1)) == role);
				}
			}
			cl_ulong	mask = __ballot(own);
			if (own)
				queue[(qtail + (cl_uint)__popcll(mask & ((1UL << strom_lane_id()) - 1)))
					  & (GPUPREAGG_HASH_QUEUE - 1)] = kidx[j];
			qtail += (cl_uint)__popcll(mask);
			__builtin_amdgcn_wave_barrier();
		}
		if (ROLES)
		{
			while (qtail - qhead >= STROM_WAVE)
				drain(STROM_WAVE);
		}
	}
	if (ROLES && qtail != qhead)
		drain(qtail - qhead);
	if (FOLD)
	{
		/* the work-group's groups reach the global table once */
		const gpupreagg_flags_t *lflags = (const gpupreagg_flags_t *)lds;
		__syncthreads();
		for (cl_uint s = threadIdx.x; s < lds_slots; s += blockDim.x)
		{
			if (T.state[s] != 2)
				continue;
			cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
			cl_uint		knull = T.knull[s];
			for (int k = 0; k < GPUPREAGG_NKEYS; k++)
				kimg[k] = T.keys[s * GPUPREAGG_NKEYS + k];
			/* (no limit here: these rows are folded already; the host keeps
			 * gridDim.x * lds_slots slots of headroom for this) */
			cl_uint		slot = gpupreagg_hash_slot<true>(htab, gpupreagg_hash_of(kimg, knull), kimg, knull, ~0u);
			if (slot == GPUPREAGG_HASH_FULL)
			{
				head->overflow = 1;
				continue;
			}
			char	   *rec = gpupreagg_hash_rec(htab, slot);
			cl_uint		lf = lflags[s];
#define X(aidx,resno,OP,NAME)														\
			if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)							\
			{																		\
				cl_uint c = ((const cl_uint *)(lds + L.vals_off[aidx]))[s];			\
				if (c != 0)															\
					gpupreagg_hash_merge8<GPUPREAGG_OP_NROWS, cl_long>(HASH_REC_VALS(rec) + aidx, (cl_ulong)c);	\
			}																		\
			else if (lf & (2u << aidx))												\
				gpupreagg_hash_merge8<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>		\
					(HASH_REC_VALS(rec) + aidx, ((const cl_ulong *)(lds + L.vals_off[aidx]))[s], &chunk_status);
			GPUPREAGG_AGG_LIST(X)
#undef X
			if ((*HASH_REC_FLAGS(rec) & lf) != lf)
				atomicOr(HASH_REC_FLAGS(rec), lf);
		}
	}
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
	gpupreagg_writeback_summag(kgpreagg, summag);
}

extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_check(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
					 const kern_data_store *ktoast, const kern_row_map *krowmap, char *htab,
					 cl_uint claim_limit, kern_row_map *deferred, cl_uint lds_slots, cl_uint nroles,
					 cl_uchar *rolemap)
{
	if (kds->format == KDS_FORMAT_COLUMN)
		gpupreagg_hash_body<true, false, false>(kgpreagg, kds, ktoast, krowmap, htab, 0, NULL, 0, 1, NULL, rolemap, 0);
	else
		gpupreagg_hash_body<false, false, false>(kgpreagg, kds, ktoast, krowmap, htab, 0, NULL, 0, 1, NULL, rolemap, 0);
}

extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_hash_fold(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
					const kern_data_store *ktoast, const kern_row_map *krowmap, char *htab,
					cl_uint claim_limit, kern_row_map *deferred, cl_uint lds_slots, cl_uint nroles,
					cl_uchar *rolemap, cl_uint sum_turn)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	/* (decided once per launch; with roles a row is read by several
	 * work-groups of the XCD, so its lines should stay in L2) */
	if (kds->format != KDS_FORMAT_COLUMN)
		gpupreagg_hash_body<false, true, false>(kgpreagg, kds, ktoast, krowmap, htab, claim_limit, deferred,
												lds_slots, nroles, lds, NULL, sum_turn);
	else if (nroles > 1)
		gpupreagg_hash_body<true, true, true>(kgpreagg, kds, ktoast, krowmap, htab, claim_limit, deferred,
											  lds_slots, nroles, lds, rolemap, sum_turn);
	else
		gpupreagg_hash_body<true, true, false>(kgpreagg, kds, ktoast, krowmap, htab, claim_limit, deferred,
											   lds_slots, nroles, lds, NULL, sum_turn);
}

/* ---------------------------------------------------------------------- *
 * hashed GROUP BY over hash PARTITIONS (more groups than the hash roles' LDS
 * tables take together)
 *
 * Past ~1e5 groups the rows of a chunk meet the global table one by one: three
 * atomics on a random 64-byte record per row, 23-47 ms per 1e8 rows.  Here the
 * chunk is first ordered by a slice of the hash, so that the groups of a run
 * of rows fit a work-group's LDS table again:
 *
 *   gpupreagg_hash_check_parts  the check pass; also leaves every row's
 *                               partition (2 bytes; NONE = dropped by the
 *                               qual) and the partitions' row counts
 *   gpupreagg_hash_part_plan    counts -> partition offsets and the list of
 *                               UNITS (a partition, cut at unit_rows so that
 *                               one heavy key is not one work-group's job)
 *   gpupreagg_hash_scatter      rows -> RECORDS { knull, value bits, key
 *                               images, raw values } in partition order: a
 *                               tile of 32 x blockDim rows reserves its share
 *                               of every partition with one atomic each
 *   gpupreagg_hash_fold_parts   a work-group takes a unit: LDS table, then
 *                               one find-or-claim and one merge per group
 *                               and unit.  Run first as CLAIM pass (keys
 *                               only, under the table's fill limit: a new
 *                               group that does not fit raises 'deferred',
 *                               the host grows the table and claims again
 *                               -- claiming is idempotent), then as the
 *                               fold, where every group is found.
 *
 * The partition is the top of the row's slot index in the global table
 * (hash >> pshift), so a unit's groups also sit in one slice of the table.
 * ---------------------------------------------------------------------- */
#define GPUPREAGG_PART_NONE			0xffffu
#ifndef GPUPREAGG_ABLATE
#define GPUPREAGG_ABLATE 0		/* probes: 16 no record stores, 32 no row loads in the scatter, 64 no flush, 128 no accumulate */
#endif
#define GPUPREAGG_PART_MAX			4096
#define GPUPREAGG_SCATTER_ROWS		32		/* rows per thread and tile of the scatter */

struct gpupreagg_part_ctl {
	cl_uint		nparts;				/* power of two, <= GPUPREAGG_PART_MAX */
	cl_uint		pshift;
	cl_uint		unit_rows;
	cl_uint		nunits;				/* by gpupreagg_hash_part_plan */
	cl_uint		nrecords;
	cl_uint		deferred;			/* by the claim pass */
	cl_uint		max_units;
	cl_uint		reclen;				/* the host's record length (it sized the buffer): checked by the kernels */
};

/* the scatter and the fold: do host and device mean the same record?  (a mismatch would write
 * past the buffer) */
#define GPUPREAGG_PART_RECLEN_OK(kgpreagg, ctl)												\
	((ctl)->reclen == 8 * GPUPREAGG_REC_WORDS ||												\
	 (atomicMax(&(kgpreagg)->status, (cl_int)StromError_DataStoreCorruption), false))

/* words of a record: { knull | value bits << 32, key images, raw values of the
 * aggregates that are not NROWS (those are a bit: this row counts) }.
 * (Tried in round 3 and removed: C4's int4 input packed into word 0 -- 24-byte records instead of
 * 32.  A quarter less traffic, and 4-8 % SLOWER: records of an odd number of words leave in 8-byte
 * pieces instead of 16-byte ones, and a scattered 8-byte store is a request of its own at the L2.) */
STROM_DEVICE constexpr int gpupreagg_rec_nvals()
{
	int		n = 0;
#define X(aidx,resno,OP,NAME)	n += (GPUPREAGG_OP_##OP != GPUPREAGG_OP_NROWS ? 1 : 0);
	GPUPREAGG_AGG_LIST(X)
#undef X
	return n;
}
#define GPUPREAGG_REC_WORDS		(1 + GPUPREAGG_NKEYS + gpupreagg_rec_nvals())

/* a record moves in 16-byte pieces where its length allows (records are 16-byte aligned
 * then): a scattered 8-byte store is a request of its own at the L2 */
typedef cl_ulong gpupreagg_rec2_t __attribute__((ext_vector_type(2)));
template <int NWORDS>
STROM_DEVICE void gpupreagg_rec_copy(cl_ulong *dst, const cl_ulong *src)
{
	if ((NWORDS & 1) == 0)
	{
#pragma unroll
		for (int q = 0; q < NWORDS; q += 2)
		{
			gpupreagg_rec2_t v = { src[q], src[q + 1] };
			*(gpupreagg_rec2_t *)(dst + q) = v;
		}
	}
	else
	{
#pragma unroll
		for (int q = 0; q < NWORDS; q++)
			dst[q] = src[q];
	}
}
template <int NWORDS>
STROM_DEVICE void gpupreagg_rec_load(cl_ulong *dst, const cl_ulong *src)
{
	if ((NWORDS & 1) == 0)
	{
#pragma unroll
		for (int q = 0; q < NWORDS; q += 2)
		{
			gpupreagg_rec2_t v = *(const gpupreagg_rec2_t *)(src + q);
			dst[q] = v.x;
			dst[q + 1] = v.y;
		}
	}
	else
	{
#pragma unroll
		for (int q = 0; q < NWORDS; q++)
			dst[q] = src[q];
	}
}

STROM_DEVICE cl_ulong gpupreagg_raw_image(cl_char v)	{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong gpupreagg_raw_image(cl_short v)	{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong gpupreagg_raw_image(cl_int v)		{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong gpupreagg_raw_image(cl_long v)	{ return (cl_ulong)v; }
STROM_DEVICE cl_ulong gpupreagg_raw_image(cl_ulong v)	{ return v; }
STROM_DEVICE cl_ulong gpupreagg_raw_image(cl_float v)	{ return (cl_ulong)__float_as_uint(v); }
STROM_DEVICE cl_ulong gpupreagg_raw_image(cl_double v)	{ return (cl_ulong)__double_as_longlong(v); }
STROM_DEVICE void gpupreagg_raw_value(cl_ulong x, cl_char *v)	{ *v = (cl_char)x; }
STROM_DEVICE void gpupreagg_raw_value(cl_ulong x, cl_short *v)	{ *v = (cl_short)x; }
STROM_DEVICE void gpupreagg_raw_value(cl_ulong x, cl_int *v)	{ *v = (cl_int)x; }
STROM_DEVICE void gpupreagg_raw_value(cl_ulong x, cl_long *v)	{ *v = (cl_long)x; }
STROM_DEVICE void gpupreagg_raw_value(cl_ulong x, cl_ulong *v)	{ *v = x; }
STROM_DEVICE void gpupreagg_raw_value(cl_ulong x, cl_float *v)	{ *v = __uint_as_float((cl_uint)x); }
STROM_DEVICE void gpupreagg_raw_value(cl_ulong x, cl_double *v)	{ *v = __longlong_as_double((long long)x); }

/*
 * check pass and scatter pass: both walk the chunk (any format, row map or
 * not) and evaluate the keys and partial inputs of a row
 */
template <bool IS_COLUMN, int MODE>		/* 0 check, 1 scatter, 2 scatter through LDS */
__device__ __forceinline__ void
gpupreagg_hash_parts_body(kern_gpupreagg *kgpreagg,
						  const kern_data_store *kds,
						  const kern_data_store *ktoast,
						  const kern_row_map *krowmap,
						  cl_ushort *partmap, cl_uint *hist, cl_uint *cursor, cl_ulong *records,
						  const gpupreagg_part_ctl *ctl, cl_uint *lds_words, cl_uint lds_rows = 0)
{
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	bool		use_map = (krowmap != NULL && krowmap->nvalids >= 0);
	cl_uint		nrows = (use_map ? (cl_uint)krowmap->nvalids : kds->nitems);
	cl_int		chunk_status = StromError_Success;
	cl_ulong	summag = 0;			/* OR of the integer sums' input magnitudes */
	cl_int		param_error = StromError_Success;
	const cl_uint nparts = ctl->nparts;
	const cl_uint pshift = ctl->pshift;
	strom_kparams KP;

	if (MODE != 0 && kgpreagg->status != StromError_Success)
		return;							/* the check pass sends the chunk back */
	gpupreagg_load_kparams(KP, kparams, &param_error);
	const bool	is_column = IS_COLUMN;
	const cl_int chunk_format = kds->format;
	const bool	row_family = (chunk_format == KDS_FORMAT_ROW || chunk_format == KDS_FORMAT_ROW_FLAT);
	const kern_coldir *coldir_g = KERN_DATA_STORE_COLDIR(kds);
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (is_column ? (const char *)kds + coldir_g[colidx].values_off : NULL);	\
	const cl_uint *nul_##attno = ((is_column && coldir_g[colidx].nulls_off != 0)	\
		? (const cl_uint *)((const char *)kds + coldir_g[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	auto load_row = [&](size_t r, strom_kvars &KV, cl_int &errcode)
	{
		cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : (cl_uint)r);
		const HeapTupleHeaderData *htup = NULL;
		errcode = param_error;
		if (!is_column && row_family)
			htup = strom_locate_tuple(kds, chunk_format, kds_index);
#define X(attno,colidx,NAME)													\
		KV.KVAR_##attno = (is_column											\
			? STROM_COLUMN_REF(NAME, col_##attno, nul_##attno, kds_index)		\
			: row_family ? STROM_TUPLE_REF(NAME, kds, htup, colidx)				\
			: pg_##NAME##_vref(kds, ktoast, &errcode, colidx, kds_index));
		STROM_KVAR_LIST(X)
#undef X
		if (is_column)
			strom_kvars_from_column(KV, kds, &errcode);
		STROM_KVARS_FINISH(KV);
	};
	/* keys of a row -> images; false when an expression failed */
	auto eval_keys = [&](const strom_kvars &KV, cl_int &errcode, cl_ulong *kimg, cl_uint &knull)
	{
		knull = 0;
#define X(kidx,resno,NAME)															\
		{																			\
			pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);				\
			kimg[kidx] = (kv.isnull ? 0UL : strom_key_image(kv.value));				\
			knull |= (kv.isnull ? (1u << kidx) : 0u);								\
		}
		GPUPREAGG_KEY_LIST(X)
#undef X
	};

	if (MODE == 0)
	{
		/*
		 * check: errors as gpupreagg_hash_check finds them, the row's partition,
		 * the partitions' row counts (LDS, merged once per work-group)
		 */
		for (cl_uint i = threadIdx.x; i < nparts; i += blockDim.x)
			lds_words[i] = 0;
		__syncthreads();
		for (size_t tile = blockIdx.x;
			 tile * GPUPREAGG_HASH_UNROLL * blockDim.x < nrows;
			 tile += gridDim.x)
		{
			strom_kvars	KVs[GPUPREAGG_HASH_UNROLL];
			cl_int		errs[GPUPREAGG_HASH_UNROLL];
			size_t		pos[GPUPREAGG_HASH_UNROLL];
#pragma unroll
			for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
			{
				pos[j] = (tile * GPUPREAGG_HASH_UNROLL + j) * blockDim.x + threadIdx.x;
				load_row(pos[j] < nrows ? pos[j] : 0, KVs[j], errs[j]);
			}
#pragma unroll
			for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
			{
				if (pos[j] >= nrows)
					continue;
				cl_int		errcode = errs[j];
				cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
				cl_uint		knull;
				cl_uint		part = GPUPREAGG_PART_NONE;
				pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KVs[j]);
				if (errcode != StromError_Success || EVAL(rc))
				{
					eval_keys(KVs[j], errcode, kimg, knull);
#define X(aidx,resno,OP,NAME)														\
					pg_##NAME##_t av_##aidx = gpupreagg_agg_##aidx(&errcode, KP, KVs[j]);
					GPUPREAGG_AGG_LIST(X)
#undef X
					if (errcode != StromError_Success)
						STROM_SET_ERROR(&chunk_status, errcode);
					else
					{
#define X(aidx,resno,OP,NAME)														\
						if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value && GPUPREAGG_SUMBITS_##aidx >= 64)	\
							summag |= (av_##aidx.isnull ? 0UL : gpupreagg_sum_magnitude((cl_long)av_##aidx.value));
						GPUPREAGG_AGG_LIST(X)
#undef X
						part = (gpupreagg_hash_of(kimg, knull) >> pshift) & (nparts - 1);
						__hip_atomic_fetch_add(&lds_words[part], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
				partmap[pos[j]] = (cl_ushort)part;
			}
		}
		__syncthreads();
		for (cl_uint i = threadIdx.x; i < nparts; i += blockDim.x)
		{
			cl_uint		c = lds_words[i];
			if (c != 0)
				__hip_atomic_fetch_add(&hist[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
		gpupreagg_writeback_summag(kgpreagg, summag);
		return;
	}
	if (!GPUPREAGG_PART_RECLEN_OK(kgpreagg, ctl))
		return;
	if (MODE == 2)
	{
		/*
		 * scatter through LDS.  A scattered 16-byte store is one request at the L2, and
		 * the chip serves ~1e11 of those per second: records written where they belong
		 * one by one cost 2.7 ms per 1e8 rows, everything else in the pass 0.14 ms.  So
		 * a tile (lds_rows rows per thread) is put in partition order in LDS first and
		 * leaves in runs: consecutive lanes write consecutive 16 bytes.
		 *   LDS: lcount[P] lstart[P] lbase[P] | wave sums | records[T] | lpart[T] (u16)
		 */
		const cl_uint T = lds_rows * blockDim.x;
		cl_uint	   *lcount = lds_words;
		cl_uint	   *lstart = lcount + nparts;
		cl_uint	   *lbase = lstart + nparts;
		cl_uint	   *wsum = lbase + nparts;					/* 32 words */
		cl_ulong   *lrec = (cl_ulong *)(wsum + 32);
		cl_ushort  *lpart = (cl_ushort *)(lrec + (size_t)T * GPUPREAGG_REC_WORDS);
		const cl_uint per = (nparts + blockDim.x - 1) / blockDim.x;		/* counters per thread in the scan */

		for (cl_uint i = threadIdx.x; i < nparts; i += blockDim.x)
			lcount[i] = 0;
		__syncthreads();
		for (size_t tile = blockIdx.x; tile * T < nrows; tile += gridDim.x)
		{
			size_t		r0 = tile * T + threadIdx.x;
			cl_uint		part[4], rank[4];
			/* (A) partition and rank inside the tile's share of it */
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				size_t		r = r0 + (size_t)j * blockDim.x;
				part[j] = ((cl_uint)j < lds_rows && r < nrows ? (cl_uint)partmap[r] : GPUPREAGG_PART_NONE);
			}
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				rank[j] = 0;
				if (part[j] != GPUPREAGG_PART_NONE)
					rank[j] = __hip_atomic_fetch_add(&lcount[part[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			/* the rows' columns are on their way while the counters are scanned */
			strom_kvars	KVs[4];
			cl_int		errs[4];
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if ((cl_uint)j < lds_rows)
					load_row(part[j] != GPUPREAGG_PART_NONE ? r0 + (size_t)j * blockDim.x : 0, KVs[j], errs[j]);
			}
			__syncthreads();
			/* (B) exclusive scan of the counters; one reservation per partition that occurs */
			cl_uint		c[4], mine = 0;
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				cl_uint		i = threadIdx.x * per + q;
				c[q] = ((cl_uint)q < per && i < nparts ? lcount[i] : 0);
				mine += c[q];
			}
			cl_uint		incl = mine;
			for (int d = 1; d < STROM_WAVE; d <<= 1)
			{
				cl_uint	up = __shfl_up(incl, d, STROM_WAVE);
				if ((int)strom_lane_id() >= d)
					incl += up;
			}
			if (strom_lane_id() == STROM_WAVE - 1)
				wsum[threadIdx.x / STROM_WAVE] = incl;
			__syncthreads();
			cl_uint		before = 0, total = 0;
			for (cl_uint w = 0; w < blockDim.x / STROM_WAVE; w++)
			{
				cl_uint	v = wsum[w];
				before += (w < threadIdx.x / STROM_WAVE ? v : 0);
				total += v;
			}
			cl_uint		off = before + incl - mine;
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				cl_uint		i = threadIdx.x * per + q;
				if ((cl_uint)q < per && i < nparts)
				{
					lstart[i] = off;
					if (c[q] != 0)
						lbase[i] = __hip_atomic_fetch_add(&cursor[i], c[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					lcount[i] = 0;
					off += c[q];
				}
			}
			__syncthreads();
			/* (C) records into LDS at their place in partition order */
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if (part[j] == GPUPREAGG_PART_NONE)
					continue;
				cl_int		errcode = errs[j];
				cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
				cl_uint		knull;
				cl_uint		abits = 0;
				cl_ulong	rec[GPUPREAGG_REC_WORDS];
				int			vp = 1 + GPUPREAGG_NKEYS;
				eval_keys(KVs[j], errcode, kimg, knull);
				for (int k = 0; k < GPUPREAGG_NKEYS; k++)
					rec[1 + k] = kimg[k];
#define X(aidx,resno,OP,NAME)														\
				{																	\
					pg_##NAME##_t av = gpupreagg_agg_##aidx(&errcode, KP, KVs[j]);	\
					if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)					\
						abits |= ((!av.isnull && av.value != 0) ? (1u << aidx) : 0u);	\
					else															\
					{																\
						abits |= (!av.isnull ? (1u << aidx) : 0u);					\
						rec[vp++] = (av.isnull ? 0UL : gpupreagg_raw_image(av.value));	\
					}																\
				}
				GPUPREAGG_AGG_LIST(X)
#undef X
				rec[0] = (cl_ulong)knull | ((cl_ulong)abits << 32);
				cl_uint		pos = lstart[part[j]] + rank[j];
				gpupreagg_rec_copy<GPUPREAGG_REC_WORDS>(lrec + (size_t)pos * GPUPREAGG_REC_WORDS, rec);
				lpart[pos] = (cl_ushort)part[j];
			}
			__syncthreads();
			/* (D) out in runs: piece q of the tile's records, 16 bytes where the record length allows */
			if ((GPUPREAGG_REC_WORDS & 1) == 0)
			{
				const cl_uint ppr = GPUPREAGG_REC_WORDS / 2;
				for (cl_uint q = threadIdx.x; q < total * ppr; q += blockDim.x)
				{
					cl_uint		i = q / ppr, piece = q - i * ppr;
					cl_uint		p = lpart[i];
					size_t		g = (size_t)lbase[p] + (i - lstart[p]);
					*(gpupreagg_rec2_t *)(records + g * GPUPREAGG_REC_WORDS + 2 * piece) =
						*(const gpupreagg_rec2_t *)(lrec + (size_t)i * GPUPREAGG_REC_WORDS + 2 * piece);
				}
			}
			else
			{
				for (cl_uint q = threadIdx.x; q < total * GPUPREAGG_REC_WORDS; q += blockDim.x)
				{
					cl_uint		i = q / GPUPREAGG_REC_WORDS, piece = q - i * GPUPREAGG_REC_WORDS;
					cl_uint		p = lpart[i];
					size_t		g = (size_t)lbase[p] + (i - lstart[p]);
					records[g * GPUPREAGG_REC_WORDS + piece] = lrec[(size_t)i * GPUPREAGG_REC_WORDS + piece];
				}
			}
			__syncthreads();
		}
		return;
	}
	/*
	 * scatter.  A tile is GPUPREAGG_SCATTER_ROWS x blockDim rows: (A) its rows per
	 * partition, from the partition map; (B) one reservation per partition that
	 * occurs; (C) the rows' records go to their partition's reserved run.
	 */
	cl_uint	   *lcount = lds_words;
	cl_uint	   *lbase = lds_words + nparts;
	size_t		tile_rows = (size_t)GPUPREAGG_SCATTER_ROWS * blockDim.x;
	for (cl_uint i = threadIdx.x; i < nparts; i += blockDim.x)
		lcount[i] = 0;
	__syncthreads();
	for (size_t tile = blockIdx.x; tile * tile_rows < nrows; tile += gridDim.x)
	{
		size_t		r0 = tile * tile_rows + threadIdx.x;
#pragma unroll 1
		for (int j0 = 0; j0 < GPUPREAGG_SCATTER_ROWS; j0 += 8)
		{
			cl_uint		parts[8];
#pragma unroll
			for (int j = 0; j < 8; j++)
			{
				size_t		r = r0 + (size_t)(j0 + j) * blockDim.x;
				parts[j] = (r < nrows ? (cl_uint)partmap[r] : GPUPREAGG_PART_NONE);
			}
#pragma unroll
			for (int j = 0; j < 8; j++)
			{
				if (parts[j] != GPUPREAGG_PART_NONE)
					__hip_atomic_fetch_add(&lcount[parts[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		/* (phase C reads the map again -- two bytes per row out of L2 -- instead of keeping
		 * the partition numbers alive across the row loads) */
		__syncthreads();
		for (cl_uint i = threadIdx.x; i < nparts; i += blockDim.x)
		{
			cl_uint		c = lcount[i];
			if (c != 0)
				lbase[i] = __hip_atomic_fetch_add(&cursor[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			lcount[i] = 0;
		}
		__syncthreads();
#pragma unroll 1
		for (int j0 = 0; j0 < GPUPREAGG_SCATTER_ROWS; j0 += GPUPREAGG_HASH_UNROLL)
		{
			strom_kvars	KVs[GPUPREAGG_HASH_UNROLL];
			cl_int		errs[GPUPREAGG_HASH_UNROLL];
			cl_uint		part[GPUPREAGG_HASH_UNROLL];
#pragma unroll
			for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
			{
				size_t		r = r0 + (size_t)(j0 + j) * blockDim.x;
				part[j] = (r < nrows ? (cl_uint)partmap[r] : GPUPREAGG_PART_NONE);
				load_row((GPUPREAGG_ABLATE & 32) ? (size_t)threadIdx.x : part[j] != GPUPREAGG_PART_NONE ? r : 0, KVs[j], errs[j]);
			}
#pragma unroll
			for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
			{
				if (part[j] == GPUPREAGG_PART_NONE)
					continue;
				cl_int		errcode = errs[j];
				cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
				cl_uint		knull;
				cl_uint		abits = 0;
				eval_keys(KVs[j], errcode, kimg, knull);
				cl_uint		rank = __hip_atomic_fetch_add(&lcount[part[j]], 1u, __ATOMIC_RELAXED,
														  __HIP_MEMORY_SCOPE_WORKGROUP);
				cl_ulong   *dst = records + (size_t)(lbase[part[j]] + rank) * GPUPREAGG_REC_WORDS;
				cl_ulong	rec[GPUPREAGG_REC_WORDS];
				int			vp = 1 + GPUPREAGG_NKEYS;
				for (int k = 0; k < GPUPREAGG_NKEYS; k++)
					rec[1 + k] = kimg[k];
#define X(aidx,resno,OP,NAME)														\
				{																	\
					pg_##NAME##_t av = gpupreagg_agg_##aidx(&errcode, KP, KVs[j]);	\
					if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)					\
						abits |= ((!av.isnull && av.value != 0) ? (1u << aidx) : 0u);	\
					else															\
					{																\
						abits |= (!av.isnull ? (1u << aidx) : 0u);					\
						rec[vp++] = (av.isnull ? 0UL : gpupreagg_raw_image(av.value));	\
					}																\
				}
				GPUPREAGG_AGG_LIST(X)
#undef X
				rec[0] = (cl_ulong)knull | ((cl_ulong)abits << 32);
				if (!(GPUPREAGG_ABLATE & 16) || rec[0] == 0x123456789abcdefUL)
					gpupreagg_rec_copy<GPUPREAGG_REC_WORDS>(dst, rec);
			}
		}
		__syncthreads();
		for (cl_uint i = threadIdx.x; i < nparts; i += blockDim.x)
			lcount[i] = 0;
		__syncthreads();
	}
}

extern "C" __global__ void
__launch_bounds__(1024)
gpupreagg_hash_check_parts(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
						   const kern_data_store *ktoast, const kern_row_map *krowmap,
						   cl_ushort *partmap, cl_uint *hist, const gpupreagg_part_ctl *ctl)
{
	__shared__ cl_uint lds_words[GPUPREAGG_PART_MAX];
	if (kds->format == KDS_FORMAT_COLUMN)
		gpupreagg_hash_parts_body<true, 0>(kgpreagg, kds, ktoast, krowmap, partmap, hist, NULL, NULL, ctl, lds_words);
	else
		gpupreagg_hash_parts_body<false, 0>(kgpreagg, kds, ktoast, krowmap, partmap, hist, NULL, NULL, ctl, lds_words);
}

extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_hash_scatter(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
					   const kern_data_store *ktoast, const kern_row_map *krowmap,
					   cl_ushort *partmap, cl_uint *cursor, cl_ulong *records, const gpupreagg_part_ctl *ctl)
{
	__shared__ cl_uint lds_words[2 * GPUPREAGG_PART_MAX];
	if (kds->format == KDS_FORMAT_COLUMN)
		gpupreagg_hash_parts_body<true, 1>(kgpreagg, kds, ktoast, krowmap, partmap, NULL, cursor, records, ctl, lds_words);
	else
		gpupreagg_hash_parts_body<false, 1>(kgpreagg, kds, ktoast, krowmap, partmap, NULL, cursor, records, ctl, lds_words);
}

extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_hash_scatter_lds(kern_gpupreagg *kgpreagg, const kern_data_store *kds,
						   const kern_data_store *ktoast, const kern_row_map *krowmap,
						   cl_ushort *partmap, cl_uint *cursor, cl_ulong *records, const gpupreagg_part_ctl *ctl,
						   cl_uint lds_rows)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	if (kds->format == KDS_FORMAT_COLUMN)
		gpupreagg_hash_parts_body<true, 2>(kgpreagg, kds, ktoast, krowmap, partmap, NULL, cursor, records, ctl,
										   (cl_uint *)lds, lds_rows);
	else
		gpupreagg_hash_parts_body<false, 2>(kgpreagg, kds, ktoast, krowmap, partmap, NULL, cursor, records, ctl,
											(cl_uint *)lds, lds_rows);
}

/*
 * counts -> offsets (the scatter's cursors) and units { first record, records }.
 * One work-group; a thread owns nparts / blockDim consecutive partitions.
 */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_part_plan(const cl_uint *hist, cl_uint *cursor, cl_uint *units, gpupreagg_part_ctl *ctl)
{
	__shared__ cl_uint s_rows[256], s_units[256];
	cl_uint		nparts = ctl->nparts;
	cl_uint		unit_rows = ctl->unit_rows;
	cl_uint		per = (nparts + 255) / 256;
	cl_uint		p0 = threadIdx.x * per;
	cl_uint		rows = 0, nunits = 0;

	for (cl_uint p = p0; p < p0 + per && p < nparts; p++)
	{
		rows += hist[p];
		nunits += (hist[p] + unit_rows - 1) / unit_rows;
	}
	s_rows[threadIdx.x] = rows;
	s_units[threadIdx.x] = nunits;
	__syncthreads();
	/* exclusive scans over the 256 threads */
	for (cl_uint d = 1; d < 256; d <<= 1)
	{
		cl_uint	a = (threadIdx.x >= d ? s_rows[threadIdx.x - d] : 0);
		cl_uint	b = (threadIdx.x >= d ? s_units[threadIdx.x - d] : 0);
		__syncthreads();
		s_rows[threadIdx.x] += a;
		s_units[threadIdx.x] += b;
		__syncthreads();
	}
	cl_uint		row_off = s_rows[threadIdx.x] - rows;
	cl_uint		unit_off = s_units[threadIdx.x] - nunits;
	for (cl_uint p = p0; p < p0 + per && p < nparts; p++)
	{
		cl_uint		c = hist[p];
		cursor[p] = row_off;
		for (cl_uint done = 0; done < c; done += unit_rows)
		{
			if (unit_off < ctl->max_units)
			{
				units[2 * unit_off] = row_off + done;
				units[2 * unit_off + 1] = (c - done < unit_rows ? c - done : unit_rows);
			}
			unit_off++;
		}
		row_off += c;
	}
	if (threadIdx.x == 255)
	{
		ctl->nunits = (s_units[255] < ctl->max_units ? s_units[255] : ctl->max_units);
		ctl->nrecords = s_rows[255];
		/* (ctl->reclen is the HOST's idea of a record, which sized the buffer: the scatter and
		 * the fold compare it with theirs before they touch the records) */
	}
}

/* the slot of a key in the LDS table, or ~0u: a look, no claim (states >= 2 are taken slots) */
STROM_DEVICE cl_uint
gpupreagg_hash_lds_find(const gpupreagg_hash_lds &T, cl_uint hash, const cl_ulong *kimg, cl_uint knull)
{
	cl_uint		slot = (hash * 0x9e3779b1u) >> T.shift;

	for (cl_uint probes = 0; probes <= GPUPREAGG_HASH_LDS_PROBES; probes++)
	{
		if (T.state[slot] == 0)
			return ~0u;
		bool	same = (T.knull[slot] == knull);
		for (int k = 0; k < GPUPREAGG_NKEYS; k++)
			same = same && (T.keys[slot * GPUPREAGG_NKEYS + k] == kimg[k]);
		if (same)
			return slot;
		slot = (slot + 1) & T.mask;
	}
	return ~0u;
}

/* one record's partial inputs into the global record of its group */
#define GPUPREAGG_HASH_MERGE_ROW(grec)													\
	do {																				\
		cl_uint		need_ = GPUPREAGG_FLAG_SEEN;										\
		GPUPREAGG_AGG_LIST(X)															\
		if ((*HASH_REC_FLAGS(grec) & need_) != need_)									\
			atomicOr(HASH_REC_FLAGS(grec), need_);										\
	} while (0)

/*
 * A work-group takes units.  A unit is all or nothing:
 *   accumulate  the unit's records into a fresh LDS table (a record that finds no room
 *               there only makes sure its group has a slot in the global table)
 *   resolve     every group of the LDS table finds or claims its global slot, under
 *               claim_limit.  A new group that does not fit: the unit goes on the redo
 *               list with nothing merged -- the host grows the table and runs the list
 *   merge       the LDS table's accumulators to the resolved slots; if some records had
 *               no room in LDS, the unit's records are read again and those (the ones
 *               the LDS table does not know) are merged one by one
 * so no claim pass over the whole chunk is needed to keep the fill limit.
 */
__device__ __forceinline__ void
gpupreagg_hash_fold_units(kern_gpupreagg *kgpreagg, char *htab, cl_uint claim_limit,
						  gpupreagg_part_ctl *ctl, const cl_uint *units, const cl_ulong *records,
						  cl_uint lds_slots, const cl_uint *todo, cl_uint ntodo, cl_uint *redo, char *lds,
						  cl_uint sum_turn)
{
	gpupreagg_hash_head *head = (gpupreagg_hash_head *)htab;
	gpupreagg_lds_layout L;
	gpupreagg_hash_lds T;
	cl_uint		nunits = (todo ? ntodo : ctl->nunits);
	cl_int		chunk_status = StromError_Success;	/* (raised by the checked program's additions only) */

	if (kgpreagg->status != StromError_Success)
		return;
	if (!GPUPREAGG_PART_RECLEN_OK(kgpreagg, ctl))
		return;
	if (!gpupreagg_hash_sum_account(kgpreagg, head, sum_turn))
		return;							/* an integer sum could leave int8: nothing is folded */
	gpupreagg_lds_layout_init(L, lds_slots, 1);
	T.state = (cl_uint *)(lds + L.total);
	T.knull = T.state + lds_slots;
	T.keys = (cl_ulong *)(T.knull + lds_slots);
	T.mask = lds_slots - 1;
	T.shift = 32 - (31 - __clz((int)lds_slots));
	cl_uint	   *uflags = (cl_uint *)(T.keys + (size_t)lds_slots * GPUPREAGG_NKEYS);	/* deferred, spilled */
	for (cl_uint u = blockIdx.x; u < nunits; u += gridDim.x)
	{
		cl_uint		unit = (todo ? todo[u] : u);
		const cl_ulong *base = records + (size_t)units[2 * unit] * GPUPREAGG_REC_WORDS;
		cl_uint		count = units[2 * unit + 1];

		gpupreagg_lds_init(lds, L, lds_slots, 1);
		for (cl_uint i = threadIdx.x; i < lds_slots; i += blockDim.x)
			T.state[i] = 0;
		if (threadIdx.x < 2)
			uflags[threadIdx.x] = 0;
		__syncthreads();
		/* pass 0: accumulate.  pass 1 (only after a merge with spilled records): those records */
		for (int pass = 0; pass < 2; pass++)
		{
			for (cl_uint i0 = 0; i0 < count; i0 += GPUPREAGG_HASH_UNROLL * blockDim.x)
			{
				cl_ulong	w[GPUPREAGG_HASH_UNROLL][GPUPREAGG_REC_WORDS];
#pragma unroll
				for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
				{
					cl_uint		i = i0 + j * blockDim.x + threadIdx.x;
					gpupreagg_rec_load<GPUPREAGG_REC_WORDS>(w[j], base + (size_t)(i < count ? i : 0) * GPUPREAGG_REC_WORDS);
				}
#pragma unroll
				for (int j = 0; j < GPUPREAGG_HASH_UNROLL; j++)
				{
					if (i0 + j * blockDim.x + threadIdx.x >= count)
						continue;
					cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
					cl_uint		knull = (cl_uint)w[j][0];
					cl_uint		abits = (cl_uint)(w[j][0] >> 32);
					for (int k = 0; k < GPUPREAGG_NKEYS; k++)
						kimg[k] = w[j][1 + k];
					cl_uint		hash = gpupreagg_hash_of(kimg, knull);
					if ((GPUPREAGG_ABLATE & 128) && hash != 12345u)
						continue;
					cl_uint		lslot = (pass == 0 ? gpupreagg_hash_lds_slot(T, hash, kimg, knull)
											   : gpupreagg_hash_lds_find(T, hash, kimg, knull));
					if (pass == 1 && lslot != ~0u)
						continue;			/* merged with its group's LDS accumulators */
					/* the row's partial inputs, typed again */
					int			vp = 1 + GPUPREAGG_NKEYS;
#define X(aidx,resno,OP,NAME)														\
					pg_##NAME##_t av_##aidx;										\
					av_##aidx.isnull = false;										\
					av_##aidx.value = 0;											\
					if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)					\
						av_##aidx.value = (pg_##NAME##_base_t)((abits >> aidx) & 1u);	\
					else															\
					{																\
						av_##aidx.isnull = !((abits >> aidx) & 1u);					\
						gpupreagg_raw_value(w[j][vp++], &av_##aidx.value);			\
					}
					GPUPREAGG_AGG_LIST(X)
#undef X
					(void)vp;
					if (lslot != ~0u)
					{
						cl_uint		need = GPUPREAGG_FLAG_SEEN;
#define X(aidx,resno,OP,NAME)														\
						need |= gpupreagg_lds_accum<GPUPREAGG_OP_##OP, aidx>(lds, L.vals_off[aidx], lslot, av_##aidx, &chunk_status);
						GPUPREAGG_AGG_LIST(X)
#undef X
						gpupreagg_flags_t *flags = (gpupreagg_flags_t *)lds;
						if ((flags[lslot] & need) != need)
						{
							cl_uint *word = (cl_uint *)(lds + ((lslot * (cl_uint)sizeof(gpupreagg_flags_t)) & ~3u));
							cl_uint	 shift = ((lslot * (cl_uint)sizeof(gpupreagg_flags_t)) & 3u) * 8u;
							__hip_atomic_fetch_or(word, need << shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						}
						continue;
					}
					/* no room in the LDS table.  pass 0: a slot for the group; pass 1: merge */
					cl_uint		slot = gpupreagg_hash_slot<true>(htab, hash, kimg, knull, pass == 0 ? claim_limit : ~0u);
					if (slot == GPUPREAGG_HASH_DEFER)
					{
						uflags[0] = 1;
						continue;
					}
					if (slot == GPUPREAGG_HASH_FULL)
					{
						head->overflow = 1;
						continue;
					}
					if (pass == 0)
					{
						uflags[1] = 1;
						continue;
					}
					char	   *grec = gpupreagg_hash_rec(htab, slot);
#define X(aidx,resno,OP,NAME)														\
					{																\
						typedef pg_##NAME##_base_t base_t;							\
						bool		has = !av_##aidx.isnull;						\
						cl_ulong	x;												\
						if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)				\
							x = (has ? (cl_ulong)(cl_uint)av_##aidx.value : 0);		\
						else if (gpupreagg_is_float<base_t>::value)					\
							x = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PSUM				\
								 ? (cl_ulong)__double_as_longlong((cl_double)av_##aidx.value)	\
								 : gpupreagg_f64_ordered((cl_double)av_##aidx.value));	\
						else														\
							x = (cl_ulong)(cl_long)av_##aidx.value;					\
						if (GPUPREAGG_OP_##OP != GPUPREAGG_OP_NROWS && has)			\
							need_ |= (2u << aidx);									\
						if (has && !(GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS && x == 0))	\
							gpupreagg_hash_merge8<GPUPREAGG_OP_##OP, base_t>(HASH_REC_VALS(grec) + aidx, x, &chunk_status);	\
					}
					GPUPREAGG_HASH_MERGE_ROW(grec);
#undef X
				}
			}
			if (pass == 1)
				break;
			/* resolve: the LDS table's groups find or claim their global slots */
			const gpupreagg_flags_t *lflags = (const gpupreagg_flags_t *)lds;
			__syncthreads();
			for (cl_uint s = threadIdx.x; s < lds_slots; s += blockDim.x)
			{
				if (T.state[s] != 2 || ((GPUPREAGG_ABLATE & 64) && unit > 0))
					continue;
				cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
				cl_uint		knull = T.knull[s];
				for (int k = 0; k < GPUPREAGG_NKEYS; k++)
					kimg[k] = T.keys[s * GPUPREAGG_NKEYS + k];
				cl_uint		slot = gpupreagg_hash_slot<true>(htab, gpupreagg_hash_of(kimg, knull), kimg, knull, claim_limit);
				if (slot == GPUPREAGG_HASH_DEFER)
					uflags[0] = 1;
				else if (slot == GPUPREAGG_HASH_FULL)
					head->overflow = 1;
				else
					T.state[s] = 3 + slot;
			}
			__syncthreads();
			if (uflags[0] != 0)
			{
				/* a new group did not fit under the limit: nothing of this unit is merged */
				if (threadIdx.x == 0)
					redo[__hip_atomic_fetch_add(&ctl->deferred, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = unit;
				break;
			}
			for (cl_uint s = threadIdx.x; s < lds_slots; s += blockDim.x)
			{
				cl_uint		st = T.state[s];
				if (st < 3)
					continue;
				char	   *grec = gpupreagg_hash_rec(htab, st - 3);
				cl_uint		lf = lflags[s];
#define X(aidx,resno,OP,NAME)														\
				if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)						\
				{																	\
					cl_uint c = ((const cl_uint *)(lds + L.vals_off[aidx]))[s];		\
					if (c != 0)														\
						gpupreagg_hash_merge8<GPUPREAGG_OP_NROWS, cl_long>(HASH_REC_VALS(grec) + aidx, (cl_ulong)c);	\
				}																	\
				else if (lf & (2u << aidx))											\
					gpupreagg_hash_merge8<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>	\
						(HASH_REC_VALS(grec) + aidx, ((const cl_ulong *)(lds + L.vals_off[aidx]))[s], &chunk_status);
				GPUPREAGG_AGG_LIST(X)
#undef X
				if ((*HASH_REC_FLAGS(grec) & lf) != lf)
					atomicOr(HASH_REC_FLAGS(grec), lf);
			}
			if (uflags[1] == 0)
				break;						/* (uniform: read after the barrier above) */
		}
		__syncthreads();
	}
#if defined(GPUPREAGG_CHECKED) && GPUPREAGG_CHECKED
	gpupreagg_writeback_status(&kgpreagg->status, chunk_status);
#else
	(void)chunk_status;
#endif
}

extern "C" __global__ void
__launch_bounds__(GPUPREAGG_BLOCK)
gpupreagg_hash_fold_parts(kern_gpupreagg *kgpreagg, char *htab, cl_uint claim_limit,
						  gpupreagg_part_ctl *ctl, const cl_uint *units, const cl_ulong *records,
						  cl_uint lds_slots, const cl_uint *todo, cl_uint ntodo, cl_uint *redo, cl_uint sum_turn)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	gpupreagg_hash_fold_units(kgpreagg, htab, claim_limit, ctl, units, records, lds_slots, todo, ntodo, redo, lds,
							  sum_turn);
}

/* min / max accumulators start from their identities (sums from the zeroed table) */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_init(char *htab)
{
	gpupreagg_hash_head *head = (gpupreagg_hash_head *)htab;
	cl_uint		C = head->capacity;
	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x)
	{
		cl_ulong   *vals = HASH_REC_VALS(gpupreagg_hash_rec(htab, i));
#define X(aidx,resno,OP,NAME)															\
		if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMIN || GPUPREAGG_OP_##OP == GPUPREAGG_OP_PMAX)	\
			vals[aidx] = gpupreagg_identity<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>();
		GPUPREAGG_AGG_LIST(X)
#undef X
		(void)vals;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		head->stride = GPUPREAGG_HASH_STRIDE;
		head->naggs = GPUPREAGG_NAGGS;
	}
}

/*
 * the groups, packed for the host: records of
 * { knull u32, flags u32, keys[NKEYS] u64, vals[NAGGS] u64 } in out[], their
 * number in *counter (the order is arbitrary: partial rows are a set)
 */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_export(const char *htab, char *out, cl_uint *counter)
{
	const gpupreagg_hash_head *head = (const gpupreagg_hash_head *)htab;
	const size_t	reclen = 8 + 8 * (GPUPREAGG_NKEYS + GPUPREAGG_NAGGS);
	cl_uint		C = head->capacity;

	for (cl_uint base = blockIdx.x * blockDim.x; base < C; base += gridDim.x * blockDim.x)
	{
		cl_uint		i = base + threadIdx.x;
		const char *src = gpupreagg_hash_rec(htab, i < C ? i : 0);
		bool		ready = (i < C && *HASH_REC_STATE(src) == 2);
		cl_ulong	mask = __ballot(ready);
		cl_uint		first = 0;
		/* one reservation per wave */
		if (mask == 0)
			continue;
		if (strom_lane_id() == 0)
			first = atomicAdd(counter, (cl_uint)__popcll(mask));
		first = __shfl(first, 0, STROM_WAVE);
		if (!ready)
			continue;
		cl_uint		idx = first + (cl_uint)__popcll(mask & ((1UL << strom_lane_id()) - 1));
		char	   *rec = out + reclen * idx;
		((cl_uint *)rec)[0] = *HASH_REC_KNULL(src);
		((cl_uint *)rec)[1] = *HASH_REC_FLAGS(src);
		cl_ulong   *body = (cl_ulong *)(rec + 8);
		for (int k = 0; k < GPUPREAGG_NKEYS + GPUPREAGG_NAGGS; k++)
			body[k] = HASH_REC_KEYS(src)[k];
	}
}

/*
 * The groups packed BY OWNER for the hash-partitioned exchange between ranks
 * (csrc/parallel.cpp: hashed_exchange): group g belongs to rank owner(g) of 'nparts', a function
 * of the key alone, so every rank sends a group's partial to the same place.  First the owners are
 * counted (one LDS histogram per work-group, one atomic per owner and work-group), the host turns
 * counts into offsets, then the records leave as in gpupreagg_hash_export with one cursor per owner
 * -- a wave reserves once per owner it holds records of.
 */
STROM_DEVICE cl_uint
gpupreagg_hash_owner(cl_uint hash, cl_uint nparts)
{
	/* (the table's slot is the hash's low bits, the partition plan's unit its high ones: a
	 * multiplicative remix keeps the owner independent of both) */
	return (cl_uint)(((cl_ulong)(hash * 0x9e3779b1u) * nparts) >> 32);
}

STROM_DEVICE cl_uint
gpupreagg_hash_rec_owner(const char *src, cl_uint nparts)
{
	cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
	for (int k = 0; k < GPUPREAGG_NKEYS; k++)
		kimg[k] = HASH_REC_KEYS(src)[k];
	return gpupreagg_hash_owner(gpupreagg_hash_of(kimg, *HASH_REC_KNULL(src)), nparts);
}

#define GPUPREAGG_HASH_MAXOWNERS	64

extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_owner_count(const char *htab, cl_uint nparts, cl_uint *counts)
{
	__shared__ cl_uint	s_counts[GPUPREAGG_HASH_MAXOWNERS];
	const gpupreagg_hash_head *head = (const gpupreagg_hash_head *)htab;
	cl_uint		C = head->capacity;

	if (threadIdx.x < GPUPREAGG_HASH_MAXOWNERS)
		s_counts[threadIdx.x] = 0;
	__syncthreads();
	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x)
	{
		const char *src = gpupreagg_hash_rec(htab, i);
		if (*HASH_REC_STATE(src) == 2)
			atomicAdd(&s_counts[gpupreagg_hash_rec_owner(src, nparts)], 1u);
	}
	__syncthreads();
	if (threadIdx.x < nparts && s_counts[threadIdx.x] != 0)
		atomicAdd(&counts[threadIdx.x], s_counts[threadIdx.x]);
}

/* offsets[p] = first record of owner p in out[]; cursors[p] starts at 0 */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_export_parts(const char *htab, char *out, cl_uint nparts,
							const cl_uint *offsets, cl_uint *cursors)
{
	const gpupreagg_hash_head *head = (const gpupreagg_hash_head *)htab;
	const size_t	reclen = 8 + 8 * (GPUPREAGG_NKEYS + GPUPREAGG_NAGGS);
	cl_uint		C = head->capacity;

	for (cl_uint base = blockIdx.x * blockDim.x; base < C; base += gridDim.x * blockDim.x)
	{
		cl_uint		i = base + threadIdx.x;
		const char *src = gpupreagg_hash_rec(htab, i < C ? i : 0);
		bool		ready = (i < C && *HASH_REC_STATE(src) == 2);
		cl_uint		owner = (ready ? gpupreagg_hash_rec_owner(src, nparts) : ~0u);
		cl_uint		idx = 0;
		cl_ulong	todo = __ballot(ready);
		/* one reservation per wave and owner */
		while (todo != 0)
		{
			cl_uint		p = (cl_uint)__shfl((int)owner, (int)__builtin_ctzll(todo), STROM_WAVE);
			cl_ulong	mask = __ballot(ready && owner == p);
			cl_uint		first = 0;
			if (strom_lane_id() == 0)
				first = atomicAdd(&cursors[p], (cl_uint)__popcll(mask));
			first = __shfl(first, 0, STROM_WAVE);
			if (ready && owner == p)
				idx = offsets[p] + first + (cl_uint)__popcll(mask & ((1UL << strom_lane_id()) - 1));
			todo &= ~mask;
		}
		if (!ready)
			continue;
		char	   *rec = out + reclen * idx;
		((cl_uint *)rec)[0] = *HASH_REC_KNULL(src);
		((cl_uint *)rec)[1] = *HASH_REC_FLAGS(src);
		cl_ulong   *body = (cl_ulong *)(rec + 8);
		for (int k = 0; k < GPUPREAGG_NKEYS + GPUPREAGG_NAGGS; k++)
			body[k] = HASH_REC_KEYS(src)[k];
	}
}

/*
 * the groups as TUPSLOT rows, written where the host will copy them from: what a fetch needs
 * when no partial is a 64-bit numeric (those may split into two rows: host).  rows = the
 * first row of the destination image, stride = KDS_TUPSLOT_STRIDE(ncols); tmeta[i] describes
 * target i: bits 0-7 datum length, 8 key, 9 float, 10 float4 output, 11 NROWS, 12 stored as
 * order-preserving key (float pmin / pmax), 16-23 its index among the keys / the aggregates.
 * Without codegen macros: the stored forms are 8-byte images whatever the type.
 */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_export_rows(const char *htab, char *rows, cl_uint stride, cl_uint ncols,
						   const cl_uint *tmeta, cl_uint *counter)
{
	const gpupreagg_hash_head *head = (const gpupreagg_hash_head *)htab;
	cl_uint		C = head->capacity;

	for (cl_uint base = blockIdx.x * blockDim.x; base < C; base += gridDim.x * blockDim.x)
	{
		cl_uint		i = base + threadIdx.x;
		const char *src = gpupreagg_hash_rec(htab, i < C ? i : 0);
		bool		ready = (i < C && *HASH_REC_STATE(src) == 2);
		cl_ulong	mask = __ballot(ready);
		cl_uint		first = 0;
		if (mask == 0)
			continue;
		if (strom_lane_id() == 0)
			first = atomicAdd(counter, (cl_uint)__popcll(mask));
		first = __shfl(first, 0, STROM_WAVE);
		if (!ready)
			continue;
		cl_uint		idx = first + (cl_uint)__popcll(mask & ((1UL << strom_lane_id()) - 1));
		cl_ulong   *values = (cl_ulong *)(rows + (size_t)stride * idx);
		cl_char	   *isnull = (cl_char *)(values + ncols);
		cl_uint		knull = *HASH_REC_KNULL(src);
		cl_uint		flags = *HASH_REC_FLAGS(src);
		for (cl_uint w = ncols; w < stride / 8; w++)
			values[w] = 0;						/* the NULL flags and the padding behind them */
		for (cl_uint c = 0; c < ncols; c++)
		{
			cl_uint		m = tmeta[c];
			cl_uint		len = m & 0xffu, which = (m >> 16) & 0xffu;
			cl_ulong	raw;
			bool		null;
			if (m & 0x100u)
			{
				raw = HASH_REC_KEYS(src)[which];
				null = ((knull >> which) & 1u) != 0;
			}
			else
			{
				raw = HASH_REC_KEYS(src)[GPUPREAGG_NKEYS + which];
				null = (!(m & 0x800u) && !(flags & (2u << which)));
			}
			if (!null && (m & 0x200u))
			{
				if (m & 0x1000u)				/* order-preserving key -> IEEE bits */
					raw = (raw & 0x8000000000000000UL) ? (raw & 0x7fffffffffffffffUL) : ~raw;
				if (m & 0x400u)
					raw = (cl_ulong)__float_as_uint((cl_float)__longlong_as_double((long long)raw));
			}
			else if (!null && len < 8)
				raw &= (1UL << (8 * len)) - 1;
			values[c] = (null ? 0UL : raw);
			if (null)
				isnull[c] = 1;
		}
	}
}

/*
 * the other way round: packed groups (another session's, another GPU's -- all-gathered by
 * strom_gpupreagg_allreduce) merged into this table.  recs[] is nsegs segments of seg_len
 * records of which the first counts[seg] are set; segment skip_seg (this rank's own) is left
 * out.  The host has made room for every incoming group.
 */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_import(char *htab, const char *recs, cl_uint seg_len, cl_uint nsegs,
					  const cl_uint *counts, cl_uint skip_seg)
{
	gpupreagg_hash_head *head = (gpupreagg_hash_head *)htab;
	const size_t	reclen = 8 + 8 * (GPUPREAGG_NKEYS + GPUPREAGG_NAGGS);
	size_t		total = (size_t)seg_len * nsegs;

	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
	{
		cl_uint		seg = (cl_uint)(i / seg_len);
		cl_uint		j = (cl_uint)(i - (size_t)seg * seg_len);
		if (seg == skip_seg || j >= counts[seg])
			continue;
		const char *rec = recs + reclen * i;
		cl_uint		knull = ((const cl_uint *)rec)[0];
		cl_uint		flags = ((const cl_uint *)rec)[1];
		const cl_ulong *body = (const cl_ulong *)(rec + 8);
		cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
		for (int k = 0; k < GPUPREAGG_NKEYS; k++)
			kimg[k] = body[k];
		cl_uint		slot = gpupreagg_hash_slot<true>(htab, gpupreagg_hash_of(kimg, knull), kimg, knull, ~0u);
		if (slot == GPUPREAGG_HASH_FULL)
		{
			head->overflow = 1;
			continue;
		}
		char	   *grec = gpupreagg_hash_rec(htab, slot);
		const cl_ulong *vals = body + GPUPREAGG_NKEYS;
#define X(aidx,resno,OP,NAME)														\
		if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)								\
		{																			\
			if (vals[aidx] != 0)													\
				gpupreagg_hash_merge8<GPUPREAGG_OP_NROWS, cl_long>(HASH_REC_VALS(grec) + aidx, vals[aidx]);	\
		}																			\
		else if (flags & (2u << aidx))												\
			gpupreagg_hash_merge8<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>(HASH_REC_VALS(grec) + aidx, vals[aidx]);
		GPUPREAGG_AGG_LIST(X)
#undef X
		(void)vals;
		if ((*HASH_REC_FLAGS(grec) & flags) != flags)
			atomicOr(HASH_REC_FLAGS(grec), flags);
	}
}

/*
 * Would gpupreagg_hash_import take an integer sum out of int8?  Read-only pass over ONE segment of
 * records with pairwise different keys (another session's export): a record either meets its
 * group in this table -- then the two partial sums are added in range, or *overflowed is set --
 * or is new here.  The host imports only when nothing was set ("integer sums never wrap").
 */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_import_verify(char *htab, const char *recs, cl_uint count, cl_uint *overflowed)
{
	const size_t	reclen = 8 + 8 * (GPUPREAGG_NKEYS + GPUPREAGG_NAGGS);

	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
	{
		const char *rec = recs + reclen * i;
		cl_uint		knull = ((const cl_uint *)rec)[0];
		cl_uint		flags = ((const cl_uint *)rec)[1];
		const cl_ulong *body = (const cl_ulong *)(rec + 8);
		cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
		for (int k = 0; k < GPUPREAGG_NKEYS; k++)
			kimg[k] = body[k];
		/* (claim limit 0: an empty slot ends the search, nothing is claimed) */
		cl_uint		slot = gpupreagg_hash_slot<true>(htab, gpupreagg_hash_of(kimg, knull), kimg, knull, 0u);
		if (slot == GPUPREAGG_HASH_FULL || slot == GPUPREAGG_HASH_DEFER)
			continue;
		const char *grec = gpupreagg_hash_rec(htab, slot);
		cl_uint		gflags = *HASH_REC_FLAGS(grec);
		const cl_ulong *vals = body + GPUPREAGG_NKEYS;
		bool		bad = false;
#define X(aidx,resno,OP,NAME)															\
		if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value && (flags & gflags & (2u << aidx)))	\
		{																				\
			cl_long		sum;															\
			bad = bad || __builtin_add_overflow((cl_long)HASH_REC_VALS(grec)[aidx], (cl_long)vals[aidx], &sum);	\
		}
		GPUPREAGG_AGG_LIST(X)
#undef X
		(void)vals;
		(void)gflags;
		if (bad)
			*overflowed = 1u;
	}
}

/* growth: every group of the old table moves to a (zeroed, initialised) larger one */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_rehash(const char *otab, char *ntab)
{
	const gpupreagg_hash_head *ohead = (const gpupreagg_hash_head *)otab;
	gpupreagg_hash_head *nhead = (gpupreagg_hash_head *)ntab;
	cl_uint		C = ohead->capacity;

	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		nhead->sum_bound[0] = ohead->sum_bound[0];
		nhead->sum_bound[1] = ohead->sum_bound[1];
	}
	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x)
	{
		const char *src = gpupreagg_hash_rec(otab, i);
		if (*HASH_REC_STATE(src) != 2)
			continue;
		cl_ulong	kimg[GPUPREAGG_NKEYS + 1];
		cl_uint		knull = *HASH_REC_KNULL(src);
		for (int k = 0; k < GPUPREAGG_NKEYS; k++)
			kimg[k] = HASH_REC_KEYS(src)[k];
		cl_uint		slot = gpupreagg_hash_slot<false>(ntab, gpupreagg_hash_of(kimg, knull), kimg, knull, ~0u);
		if (slot == GPUPREAGG_HASH_FULL)
		{
			nhead->overflow = 1;
			continue;
		}
		char	   *dst = gpupreagg_hash_rec(ntab, slot);
		*HASH_REC_FLAGS(dst) = *HASH_REC_FLAGS(src);
		for (int a = 0; a < GPUPREAGG_NAGGS; a++)
			HASH_REC_VALS(dst)[a] = HASH_REC_VALS(src)[a];
	}
}

/*
 * the bound of the integer sums, measured: the largest |sum| the table holds now.  The
 * host zeroes BOTH slots first and runs this after groups arrived otherwise than by a fold
 * (merge / import), or when a fold found the running bound at 2^63 (see
 * gpupreagg_hash_sum_account): the bound restarts from what is really there.
 */
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_hash_sum_refresh(char *htab)
{
	gpupreagg_hash_head *head = (gpupreagg_hash_head *)htab;
	cl_uint		C = head->capacity;
	cl_ulong	most = 0;

	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x)
	{
		const char *rec = gpupreagg_hash_rec(htab, i);
		if (*HASH_REC_STATE(rec) != 2)
			continue;
		cl_uint		flags = *HASH_REC_FLAGS(rec);
#define X(aidx,resno,OP,NAME)															\
		if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value && (flags & (2u << aidx)))	\
		{																				\
			cl_ulong m = gpupreagg_sum_magnitude((cl_long)HASH_REC_VALS(rec)[aidx]);	\
			most = (m >= most ? m : most);												\
		}
		GPUPREAGG_AGG_LIST(X)
#undef X
	}
#pragma unroll
	for (int m = 32; m > 0; m >>= 1)
	{
		cl_ulong o = ((cl_ulong)__shfl_xor((cl_uint)(most >> 32), m, STROM_WAVE) << 32) |
			__shfl_xor((cl_uint)most, m, STROM_WAVE);
		most = (o > most ? o : most);
	}
	if (strom_lane_id() == 0 && most != 0)
	{
		/* |sum| <= magnitude + 1 */
		cl_ulong	bound = (most >= (1UL << 63) - 1 ? (1UL << 63) : most + 1);
		__hip_atomic_fetch_max(&head->sum_bound[0], bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_fetch_max(&head->sum_bound[1], bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

#endif	/* GPUPREAGG_HASHED */

#ifndef GPUPREAGG_HASHED
/* ====================================================================== *
 * census: which dense ids occur in this chunk (after the qual)?
 *
 * Zone maps bound each key separately; the product of the ranges can be
 * far larger than the number of combinations present (TPC-H Q1: 19 x 11
 * dense ids, 4-6 groups).  The host compacts the ids that occur into table
 * slots (remap), which restores full LDS replication / the lane-private
 * kernel for such queries, and is the step where the ranks of a multi-GPU
 * run agree on the slots.  One bit per dense id.
 * ====================================================================== */
template <bool IS_COLUMN>
__device__ __forceinline__ void
gpupreagg_census_body(const kern_gpupreagg *kgpreagg, const kern_data_store *kds,
				 const kern_data_store *ktoast, const kern_row_map *krowmap,
				 const gpupreagg_dense_ctl *ctl, cl_uint *bitmap)
{
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	bool		use_map = (krowmap != NULL && krowmap->nvalids >= 0);
	size_t		nrows = (use_map ? (size_t)krowmap->nvalids : (size_t)kds->nitems);
	cl_int		param_error = StromError_Success;
	strom_kparams KP;

	gpupreagg_load_kparams(KP, kparams, &param_error);
	/* COLUMN chunk (row map, census): column pointers hoisted, no chunk
	 * header field is read per row */
	const bool	is_column = IS_COLUMN;		/* fixed per launch */
	const cl_int chunk_format = kds->format;
	const bool	row_family = (chunk_format == KDS_FORMAT_ROW || chunk_format == KDS_FORMAT_ROW_FLAT);
	const kern_coldir *coldir_g = KERN_DATA_STORE_COLDIR(kds);
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (is_column ? (const char *)kds + coldir_g[colidx].values_off : NULL);	\
	const cl_uint *nul_##attno = ((is_column && coldir_g[colidx].nulls_off != 0)	\
		? (const cl_uint *)((const char *)kds + coldir_g[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
		 r < nrows;
		 r += (size_t)gridDim.x * blockDim.x)
	{
		cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : (cl_uint)r);
		strom_kvars	KV;
		cl_int		errcode = param_error;
		cl_uint		gid = 0;
		bool		out_of_domain = false;
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
		if (is_column)
			strom_kvars_from_column(KV, kds, &errcode);
		STROM_KVARS_FINISH(KV);
		pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KV);
		if (errcode == StromError_Success && !EVAL(rc))
			continue;
#define X(kidx,resno,NAME)															\
		{																			\
			pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);				\
			cl_long		off64 = (cl_long)kv.value - ctl->key_min[kidx];				\
			cl_uint		range = ctl->key_range[kidx];								\
			cl_uint		off = (kv.isnull ? range : (cl_uint)off64);					\
			if (!kv.isnull && (off64 < 0 || off64 >= (cl_long)range))				\
				out_of_domain = true;												\
			gid += (kidx == 0 ? off : off * ctl->key_stride[kidx]);	/* (stride 0 is 1) */										\
		}
		GPUPREAGG_KEY_LIST(X)
#undef X
		if (out_of_domain)
			continue;				/* the fold reports it */
		cl_uint		bit = 1u << (gid & 31);
		if (!(bitmap[gid >> 5] & bit))
			atomicOr(&bitmap[gid >> 5], bit);
	}
}

#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_census(const kern_gpupreagg *kgpreagg, const kern_data_store *kds,
				 const kern_data_store *ktoast, const kern_row_map *krowmap,
				 const gpupreagg_dense_ctl *ctl, cl_uint *bitmap)
{
	/* the chunk format is decided once per launch, not once per datum */
	if (kds->format == KDS_FORMAT_COLUMN)
		gpupreagg_census_body<true>(kgpreagg, kds, ktoast, krowmap, ctl, bitmap);
	else
		gpupreagg_census_body<false>(kgpreagg, kds, ktoast, krowmap, ctl, bitmap);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */

/* ====================================================================== *
 * key range of ONE chunk (after the qual): the dense domain of a request
 * that arrives without one -- the reference's per-chunk message
 * (pgstrom_gpupreagg, opencl_gpupreagg.h:994-1003) knows nothing about the
 * other chunks of its query; its kernels sort whatever keys the chunk holds
 * (opencl_gpupreagg.h:620-856).  Here the host sizes the dense table from
 * this pass (strom_gpupreagg_chunk_domain / strom_submit_gpupreagg_chunk),
 * or goes to the hashed GROUP BY when the ranges are too sparse.
 * Per key: min, max over the rows that pass the qual and have a non-NULL
 * key; a lane keeps its own pair, a wave combines by shuffles, one atomic
 * pair per wave and key.
 * ====================================================================== */
struct gpupreagg_keyrange_t {
	cl_long		kmin[GPUPREAGG_MAXKEYS];
	cl_long		kmax[GPUPREAGG_MAXKEYS];
	cl_uint		nvalues[GPUPREAGG_MAXKEYS];		/* != 0: the key had a non-NULL value */
	cl_uint		nrows;							/* != 0: a row passed the qual */
	cl_uint		__pad;
};

template <bool IS_COLUMN>
__device__ __forceinline__ void
gpupreagg_keyrange_body(const kern_gpupreagg *kgpreagg, const kern_data_store *kds,
						const kern_data_store *ktoast, const kern_row_map *krowmap,
						gpupreagg_keyrange_t *out)
{
	const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
	bool		use_map = (krowmap != NULL && krowmap->nvalids >= 0);
	size_t		nrows = (use_map ? (size_t)krowmap->nvalids : (size_t)kds->nitems);
	cl_int		param_error = StromError_Success;
	strom_kparams KP;

	gpupreagg_load_kparams(KP, kparams, &param_error);
	const bool	is_column = IS_COLUMN;
	const cl_int chunk_format = kds->format;
	const bool	row_family = (chunk_format == KDS_FORMAT_ROW || chunk_format == KDS_FORMAT_ROW_FLAT);
	const kern_coldir *coldir_g = KERN_DATA_STORE_COLDIR(kds);
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (is_column ? (const char *)kds + coldir_g[colidx].values_off : NULL);	\
	const cl_uint *nul_##attno = ((is_column && coldir_g[colidx].nulls_off != 0)	\
		? (const cl_uint *)((const char *)kds + coldir_g[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	cl_long		my_min[GPUPREAGG_NKEYS > 0 ? GPUPREAGG_NKEYS : 1];
	cl_long		my_max[GPUPREAGG_NKEYS > 0 ? GPUPREAGG_NKEYS : 1];
	cl_uint		my_seen = 0;			/* bit k: key k had a value; bit 31: a row passed */
	for (int k = 0; k < (GPUPREAGG_NKEYS > 0 ? GPUPREAGG_NKEYS : 1); k++)
	{
		my_min[k] = 0x7fffffffffffffffL;
		my_max[k] = -0x7fffffffffffffffL - 1;
	}
	for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
		 r < nrows;
		 r += (size_t)gridDim.x * blockDim.x)
	{
		cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : (cl_uint)r);
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
		if (is_column)
			strom_kvars_from_column(KV, kds, &errcode);
		STROM_KVARS_FINISH(KV);
		pg_bool_t	rc = gpupreagg_qual_eval(&errcode, KP, KV);
		if (errcode == StromError_Success && !EVAL(rc))
			continue;
		my_seen |= 0x80000000u;
#define X(kidx,resno,NAME)															\
		{																			\
			pg_##NAME##_t kv = gpupreagg_key_##kidx(&errcode, KP, KV);				\
			if (!kv.isnull)															\
			{																		\
				cl_long	v = (cl_long)kv.value;										\
				my_min[kidx] = (v < my_min[kidx] ? v : my_min[kidx]);				\
				my_max[kidx] = (v > my_max[kidx] ? v : my_max[kidx]);				\
				my_seen |= (1u << kidx);											\
			}																		\
		}
		GPUPREAGG_KEY_LIST(X)
#undef X
	}
	/* wave combine, then one atomic pair per wave and key */
	for (int off = STROM_WAVE / 2; off > 0; off >>= 1)
	{
		my_seen |= (cl_uint)__shfl_xor((int)my_seen, off, STROM_WAVE);
		for (int k = 0; k < GPUPREAGG_NKEYS; k++)
		{
			cl_long	omin = __shfl_xor(my_min[k], off, STROM_WAVE);
			cl_long	omax = __shfl_xor(my_max[k], off, STROM_WAVE);
			my_min[k] = (omin < my_min[k] ? omin : my_min[k]);
			my_max[k] = (omax > my_max[k] ? omax : my_max[k]);
		}
	}
	if ((threadIdx.x & (STROM_WAVE - 1)) == 0 && my_seen != 0)
	{
		atomicOr(&out->nrows, 1u);
		for (int k = 0; k < GPUPREAGG_NKEYS; k++)
		{
			if (!(my_seen & (1u << k)))
				continue;
			atomicOr(&out->nvalues[k], 1u);
			atomicMin((long long *)&out->kmin[k], (long long)my_min[k]);
			atomicMax((long long *)&out->kmax[k], (long long)my_max[k]);
		}
	}
}

#if !defined(GPUPREAGG_LOOKUP_ONLY)
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_keyrange(const kern_gpupreagg *kgpreagg, const kern_data_store *kds,
				   const kern_data_store *ktoast, const kern_row_map *krowmap,
				   gpupreagg_keyrange_t *out)
{
	if (kds->format == KDS_FORMAT_COLUMN)
		gpupreagg_keyrange_body<true>(kgpreagg, kds, ktoast, krowmap, out);
	else
		gpupreagg_keyrange_body<false>(kgpreagg, kds, ktoast, krowmap, out);
}
#endif	/* !GPUPREAGG_LOOKUP_ONLY */

/* ====================================================================== *
 * slabs -> resident table, fixed order; skipped when the chunk failed
 * ====================================================================== */
/*
 * CHECK: compute everything, write nothing, raise the chunk status when a
 * NUMERIC sum leaves the 64-bit form while slabs are added to each other or to
 * the table, or when an INTEGER sum's chunk total leaves int8 (tier 2 of
 * "integer sums never wrap": the slabs are exact, they are added up in 128 bits
 * here) -- the reference discards such a chunk's device result and re-does it
 * on the CPU (gpupreagg.c:2746-2750), so the table must not have taken part of
 * it.  Launched in front of the real merge for programs that can need it; the
 * real merge computes the same and cannot fail.
 */
template <bool CHECK>
__device__ __forceinline__ void
gpupreagg_dense_merge_body(kern_gpupreagg *__restrict__ kgpreagg,
						   const gpupreagg_dense_ctl *__restrict__ ctl,
						   const char *__restrict__ slabs,
						   char *__restrict__ table)
{
	/*
	 * 256 threads = GL lanes along consecutive groups (coalesced slab
	 * reads) x WS stripes over the slabs.  A thread folds slabs stripe,
	 * stripe+WS, ... in that order, the stripes are then combined by a
	 * fixed-shape tree in LDS: the result depends on the launch geometry
	 * only, never on timing -- float sums are reproducible.
	 */
	__shared__ cl_ulong	red_val[256];
	__shared__ cl_long	red_hi[256];
	__shared__ cl_uint	red_flags[256];
	cl_uint		N = ctl->ngroups;
	cl_uint		G = ctl->groups_per_split;
	cl_uint		nsplits = ctl->nsplits;
	cl_uint		wgs_per_split = ctl->nslabs / nsplits;
	size_t		slab_bytes = ctl->slab_bytes;
	cl_uint	   *t_flags = (cl_uint *)table;
	cl_uint		WS = 1;
	cl_int		merr = StromError_Success;

	if (kgpreagg->status != StromError_Success)
		return;
	/* integer sums: which tier of the range proof holds (every work-group decides the same
	 * from the same words) */
	const bool	has_intsums = (gpupreagg_intsum_index(GPUPREAGG_NAGGS) > 0);
	bool		chunk_proven = true;
	if (has_intsums)
	{
		cl_uint		magbits = *KERN_GPUPREAGG_SUM_MAGBITS(kgpreagg);
		chunk_proven = gpupreagg_sum_range_proven(KERN_GPUPREAGG_FOLD_NROWS(kgpreagg), magbits);
#if !(defined(GPUPREAGG_CHECKED) && GPUPREAGG_CHECKED)
		if (!gpupreagg_sum_range_proven(KERN_GPUPREAGG_WG_ROWS(kgpreagg), magbits))
		{
			/* tier 3: a work-group's own sums may have wrapped -- nothing is merged, the
			 * host folds the chunk again with the checked program */
			if (blockIdx.x == 0 && threadIdx.x == 0)
				atomicMax(&kgpreagg->status, StromError_SumRangeUnproven);
			return;
		}
#endif
	}
#if !(defined(GPUPREAGG_NUMERIC_AGGS) && GPUPREAGG_NUMERIC_AGGS)
	if (CHECK && chunk_proven)
		return;							/* tier 1: nothing to check */
#endif
	/* stripes: about 8 slabs per thread (the loop is latency bound: the
	 * slabs are small), at least 4 group lanes for some coalescing */
	while (WS < 64 && WS * 8 < wgs_per_split)
		WS <<= 1;
	if (ctl->merge_ws != 0)
		WS = ctl->merge_ws;
	cl_uint		GL = 256 / WS;
	cl_uint		lane = threadIdx.x % GL;
	cl_uint		stripe = threadIdx.x / GL;

	for (cl_uint gbase = blockIdx.x * GL; gbase < N; gbase += gridDim.x * GL)
	{
		cl_uint		gid = gbase + lane;
		bool		valid = (gid < N);
		cl_uint		split = (valid ? gid / G : 0);
		cl_uint		lgid = (valid ? gid - split * G : 0);
		const char *slab0 = slabs + (size_t)split * slab_bytes;
		size_t		slab_step = (size_t)nsplits * slab_bytes;
		cl_uint		flags = 0;

		if (valid)
			for (cl_uint w = stripe; w < wgs_per_split; w += WS)
				flags |= ((const gpupreagg_flags_t *)(slab0 + w * slab_step))[lgid];
		red_flags[threadIdx.x] = flags;
		__syncthreads();
		for (cl_uint s = WS / 2; s > 0; s >>= 1)
		{
			if (stripe < s)
				red_flags[threadIdx.x] |= red_flags[threadIdx.x + s * GL];
			__syncthreads();
		}
		flags = red_flags[lane];
		__syncthreads();
		cl_uint		had = (valid && stripe == 0 ? t_flags[gid] : 0);
#define X(aidx,resno,OP,NAME)																	\
		{																						\
			cl_uint		s_vals = gpupreagg_image_offset(1 + aidx, G, 1);						\
			cl_ulong   *t_vals = (cl_ulong *)(table + gpupreagg_table_offset(1 + aidx, N));	\
			cl_ulong	acc = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS ? 0UL :					\
							   gpupreagg_identity<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS		\
												  ? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,		\
												  pg_##NAME##_base_t>());						\
			bool		live = (valid && (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS				\
										  ? flags != 0 : (flags & (2u << aidx)) != 0));			\
			if (CHECK && !chunk_proven &&														\
				gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value)				\
			{																					\
				/* tier 2: the slabs are exact; their sum, in 128 bits, must fit int8 */		\
				cl_ulong	lo = 0;																\
				cl_long		hi = 0;																\
				if (live)																		\
					for (cl_uint w = stripe; w < wgs_per_split; w += WS)						\
					{																			\
						const char *slab = slab0 + w * slab_step;								\
						if (((const gpupreagg_flags_t *)slab)[lgid] & (2u << aidx))				\
						{																		\
							cl_ulong x = ((const cl_ulong *)(slab + s_vals))[lgid];				\
							cl_ulong nlo = lo + x;												\
							hi += ((cl_long)x >> 63) + (nlo < lo ? 1L : 0L);					\
							lo = nlo;															\
						}																		\
					}																			\
				red_val[threadIdx.x] = lo;														\
				red_hi[threadIdx.x] = hi;														\
				__syncthreads();																\
				for (cl_uint s = WS / 2; s > 0; s >>= 1)										\
				{																				\
					if (stripe < s)																\
					{																			\
						cl_ulong a0 = red_val[threadIdx.x], b0 = red_val[threadIdx.x + s * GL];	\
						cl_ulong n0 = a0 + b0;													\
						red_val[threadIdx.x] = n0;												\
						red_hi[threadIdx.x] += red_hi[threadIdx.x + s * GL] + (n0 < a0 ? 1L : 0L);	\
					}																			\
					__syncthreads();															\
				}																				\
				if (live && stripe == 0 &&														\
					red_hi[threadIdx.x] != ((cl_long)red_val[threadIdx.x] >> 63))				\
					merr = StromError_CpuReCheck;												\
				__syncthreads();																\
			}																					\
			else																				\
			{																					\
			if (live)																			\
			{																					\
				for (cl_uint w = stripe; w < wgs_per_split; w += WS)							\
				{																				\
					const char *slab = slab0 + w * slab_step;									\
					if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)								\
						acc += ((const cl_uint *)(slab + s_vals))[lgid];						\
					else if (((const gpupreagg_flags_t *)slab)[lgid] & (2u << aidx))			\
						acc = gpupreagg_merge8e<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS			\
											   ? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,			\
											   pg_##NAME##_base_t>								\
							(acc, ((const cl_ulong *)(slab + s_vals))[lgid], &merr);			\
				}																				\
			}																					\
			red_val[threadIdx.x] = acc;															\
			__syncthreads();																	\
			for (cl_uint s = WS / 2; s > 0; s >>= 1)											\
			{																					\
				if (stripe < s)																	\
				{																				\
					cl_ulong o = red_val[threadIdx.x + s * GL];									\
					red_val[threadIdx.x] = (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS				\
						? red_val[threadIdx.x] + o												\
						: gpupreagg_merge8e<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS				\
										   ? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,				\
										   pg_##NAME##_base_t>(red_val[threadIdx.x], o, &merr));	\
				}																				\
				__syncthreads();																\
			}																					\
			if (live && stripe == 0)															\
			{																					\
				acc = red_val[threadIdx.x];														\
				if (GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS)									\
				{																				\
					if (!CHECK)																	\
						t_vals[gid] += acc;														\
				}																				\
				else if (gpupreagg_is_intsum<GPUPREAGG_OP_##OP, pg_##NAME##_base_t>::value)		\
				{																				\
					/* 128-bit total: the chunk's sum (exact in 64 bits, see above) is added		\
					 * with carry into {low word, high word} */									\
					if (!CHECK)																	\
					{																			\
						cl_long	   *t_hi = (cl_long *)(table + gpupreagg_table_offset			\
							(1 + GPUPREAGG_NAGGS + gpupreagg_intsum_index(aidx), N));			\
						cl_ulong	lo = ((had & (2u << aidx)) ? t_vals[gid] : 0UL);				\
						cl_long		hi = ((had & (2u << aidx)) ? t_hi[gid] : 0L);					\
						cl_ulong	nlo = lo + acc;													\
						t_vals[gid] = nlo;														\
						t_hi[gid] = hi + ((cl_long)acc >> 63) + (nlo < lo ? 1L : 0L);			\
					}																			\
				}																				\
				else																			\
				{																				\
					cl_ulong merged = (had & (2u << aidx))										\
						? gpupreagg_merge8e<GPUPREAGG_OP_##OP == GPUPREAGG_OP_NROWS				\
										   ? GPUPREAGG_OP_PSUM : GPUPREAGG_OP_##OP,				\
										   pg_##NAME##_base_t>(t_vals[gid], acc, &merr)			\
						: acc;																	\
					if (!CHECK)																	\
						t_vals[gid] = merged;													\
				}																				\
			}																					\
			__syncthreads();																	\
			}																					\
		}
		GPUPREAGG_AGG_LIST(X)
#undef X
		if (!CHECK && valid && stripe == 0 && flags != 0)
			t_flags[gid] = had | flags;
	}
	if (CHECK && __ballot(merr != StromError_Success) != 0 && (threadIdx.x & 63) == 0)
		atomicMax(&kgpreagg->status, StromError_CpuReCheck);
}

extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_dense_merge(kern_gpupreagg *kgpreagg, const gpupreagg_dense_ctl *ctl,
					  const char *slabs, char *table)
{
	gpupreagg_dense_merge_body<false>(kgpreagg, ctl, slabs, table);
}

#if (defined(GPUPREAGG_NUMERIC_AGGS) && GPUPREAGG_NUMERIC_AGGS) || (defined(GPUPREAGG_CHECKED) && GPUPREAGG_CHECKED) || \
	(defined(GPUPREAGG_HAS_INTSUMS) && GPUPREAGG_HAS_INTSUMS)
extern "C" __global__ void
__launch_bounds__(256)
gpupreagg_dense_merge_check(kern_gpupreagg *kgpreagg, const gpupreagg_dense_ctl *ctl,
							const char *slabs, char *table)
{
	gpupreagg_dense_merge_body<true>(kgpreagg, ctl, slabs, table);
}
#endif


#endif	/* !GPUPREAGG_HASHED */

#endif	/* STROM_GPUPREAGG_DEVICE_H */
