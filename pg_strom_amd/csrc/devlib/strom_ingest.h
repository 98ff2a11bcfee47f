/*
 * strom_ingest.h -- ROW / ROW_FLAT / TUPSLOT -> KDS_FORMAT_COLUMN on the device
 *
 * The step in front of the hot path: PostgreSQL hands
 * over heap pages (KDS_FORMAT_ROW, datastore.c:556-710); the streaming
 * kernels want column arrays.  One thread per row walks its tuple ONCE
 * (the single pass of kern_get_datum_tuple, opencl_common.h:817-864) and
 * stores every fixed-width attribute into the destination column at the
 * row's index -- consecutive lanes write consecutive addresses; the
 * not-null bitmap words come from one __ballot per column.  Zone-map
 * min/max of integer-like columns are folded with wave reductions.
 *
 * A varlena column that is not a numeric (text, character(n)) keeps its datums
 * as they are: the bytes go to the destination's heap area -- a wave sums its
 * lanes' (4-byte rounded) sizes, takes that much from the chunk's 'usage'
 * cursor with one atomic, and every lane copies its datum to its share -- and
 * the column array receives the datum's offset from the chunk head (strom_kds.h).
 * The datums of a chunk thus lie in no particular order; nothing reads them
 * other than through their row's offset.
 */
#ifndef STROM_INGEST_DEVICE_H
#define STROM_INGEST_DEVICE_H

#define INGEST_MAXCOLS	64

STROM_DEVICE const HeapTupleHeaderData *
ingest_get_tuple(const kern_data_store *src, cl_uint row)
{
	if (src->format == KDS_FORMAT_ROW)
		return kern_get_tuple_rs(src, row);
	if (src->format == KDS_FORMAT_ROW_FLAT)
		return kern_get_tuple_rsflat(src, row);
	return NULL;
}

/* VARNUM: some column is a varlena NUMERIC to decode (fixed per launch, so the
 * common all-fixed-width chunk does not carry the decoder) */
template <bool VARNUM, bool VARLENA = false>
__device__ __forceinline__ void
ingest_to_column_body(const kern_data_store *__restrict__ src, kern_data_store *__restrict__ dst,
					  const cl_int *__restrict__ type_oids, cl_uint *__restrict__ col_has_null)
{
	/* per work-group NULL flags (global stores to one address serialise) */
	__shared__ cl_uint	s_hasnull[INGEST_MAXCOLS];
	__shared__ cl_uint	s_failed;
	/* column descriptors staged once: read per (row, column) otherwise, each
	 * a dependent scalar load the compiler may not hoist over the stores */
	__shared__ kern_colmeta	s_colmeta[INGEST_MAXCOLS];
	__shared__ cl_uint		s_values_off[INGEST_MAXCOLS];
	__shared__ cl_uint		s_nulls_off[INGEST_MAXCOLS];
	cl_uint		nitems = src->nitems;
	cl_uint		ncols = src->ncols;
	cl_int		format = src->format;
	cl_uint		lane = threadIdx.x & 63;

	if (threadIdx.x == 0)
		s_failed = 0;
	for (cl_uint c = threadIdx.x; c < INGEST_MAXCOLS; c += blockDim.x)
	{
		s_hasnull[c] = 0;
		if (c < ncols)
		{
			const kern_coldir *cd = KERN_DATA_STORE_COLDIR(dst) + c;
			s_colmeta[c] = src->colmeta[c];
			s_values_off[c] = cd->values_off;
			s_nulls_off[c] = cd->nulls_off;
		}
	}
	__syncthreads();

	for (cl_uint base = (blockIdx.x * blockDim.x) & ~63u;
		 base < nitems;
		 base += (gridDim.x * blockDim.x))
	{
		cl_uint		row = base + threadIdx.x;		/* blockDim is a multiple of 64 */
		bool		valid = (row < nitems);
		const HeapTupleHeaderData *htup = NULL;
		cl_uint		offset = 0, natts = 0;
		bool		hasnull = false;

		if (valid && format != KDS_FORMAT_TUPSLOT)
		{
			htup = ingest_get_tuple(src, row);
			if (htup)
			{
				offset = htup->t_hoff;
				natts = htup->t_infomask2 & HEAP_NATTS_MASK;
				hasnull = (htup->t_infomask & HEAP_HASNULL) != 0;
			}
		}
		for (cl_uint c = 0; c < ncols; c++)
		{
			kern_colmeta cm = s_colmeta[c];
			const char *addr = NULL;

			if (valid)
			{
				if (format == KDS_FORMAT_TUPSLOT)
				{
					if (!KERN_DATA_STORE_ISNULL(src, row)[c])
						addr = (const char *)(KERN_DATA_STORE_VALUES(src, row) + c);
				}
				else if (htup && c < natts &&
						 !(hasnull && !(htup->t_bits[c >> 3] & (1 << (c & 7)))))
				{
					if (cm.attlen > 0)
						offset = STROM_TYPEALIGN(cm.attalign, offset);
					else if (!strom_varatt_not_pad_byte((const char *)htup + offset))
						offset = STROM_TYPEALIGN(cm.attalign, offset);
					addr = (const char *)htup + offset;
					offset += (cm.attlen > 0 ? (cl_uint)cm.attlen : strom_varsize_any(addr));
				}
			}
			/* a varlena NUMERIC becomes the 8-byte device form (canonical
			 * image); one that does not fit fails the whole conversion */
			const cl_int oid_c = ((VARNUM && type_oids != NULL) ? type_oids[c] : 0);
			const bool	to_decimal = (VARNUM && STROM_TYPE_IS_DECIMAL(oid_c));
			if (VARNUM && ((cm.attlen < 0 && oid_c == STROM_NUMERICOID) || to_decimal))
			{
				if (valid)
				{
					cl_ulong   *out = (cl_ulong *)((char *)dst + s_values_off[c]) + row;
					cl_ulong	image = 0;
					if (addr)
					{
						cl_int		e = StromError_Success;
						pg_numeric_t nv;
						if (cm.attlen < 0)
							nv = strom_numeric_from_varlena(&e, addr);
						else
						{
							nv.isnull = false;
							nv.value = strom_fetch<cl_ulong>(addr);		/* the 64-bit device form */
						}
						if (to_decimal)
						{
							/* numeric(p,s) -> int8 at 10^-s, exactly; a value that is finer than
							 * the scale or beyond 57 bits fails the whole conversion */
							pg_int8_t fx = strom_numeric_to_fixed(&e, nv, STROM_DECIMAL_TYPE_SCALE(oid_c));
							cl_long	mag = (fx.value < 0 ? -fx.value : fx.value);
							if (nv.isnull || fx.isnull || e != StromError_Success || mag >= (1L << 57))
								s_failed = 1;
							else
								image = (cl_ulong)fx.value;
						}
						else
						{
							nv = pgfn_numeric_normalize(&e, nv);
							if (nv.isnull)
								s_failed = 1;
							else
								image = nv.value;
						}
					}
					*out = image;
				}
			}
			/* any other varlena: the datum moves to the heap area, its offset to the column */
			if (VARLENA && cm.attlen < 0 && !(VARNUM && (oid_c == STROM_NUMERICOID || to_decimal)))
			{
				cl_uint		sz = (addr ? strom_varsize_any(addr) : 0);
				cl_uint		room = (sz + 3u) & ~3u;			/* every datum starts on a 4-byte boundary */
				cl_uint		before = room;
				/* inclusive prefix sum over the wave's lanes */
#pragma unroll
				for (int d = 1; d < 64; d <<= 1)
				{
					cl_uint	o = (cl_uint)__shfl_up((int)before, d, 64);
					if (lane >= (cl_uint)d)
						before += o;
				}
				cl_uint		total = (cl_uint)__shfl((int)before, 63, 64);
				cl_uint		base_off = 0;
				if (total > 0)
				{
					if (lane == 0)
						base_off = atomicAdd(&dst->usage, total);
					base_off = (cl_uint)__shfl((int)base_off, 0, 64);
				}
				cl_ulong	at = (cl_ulong)base_off + (before - room);
				if (addr && (sz < 1 || at + sz > (cl_ulong)dst->length || at < base_off))
				{
					s_failed = 1;				/* a datum length no chunk holds: corrupt tuple */
					at = 0;
				}
				else if (addr)
				{
					char   *out = (char *)dst + at;
					for (cl_uint b = 0; b < sz; b++)
						out[b] = addr[b];
					for (cl_uint b = sz; b < room; b++)
						out[b] = 0;
				}
				if (valid)
					((cl_ulong *)((char *)dst + s_values_off[c]))[row] = (addr ? at : 0UL);
			}
			/* value */
			cl_long		v = 0;
			if (to_decimal)
				;						/* stored above */
			else if (addr && cm.attlen > 0 && cm.attlen <= 8)
			{
				char *out = (char *)dst + s_values_off[c] + (size_t)cm.attlen * row;
				switch (cm.attlen)
				{
					case 1: { cl_char x = *(const cl_char *)addr; *(cl_char *)out = x; v = x; } break;
					case 2: { cl_short x = strom_fetch<cl_short>(addr); *(cl_short *)out = x; v = x; } break;
					case 4: { cl_int x = strom_fetch<cl_int>(addr); *(cl_int *)out = x; v = x; } break;
					default:{ cl_long x = strom_fetch<cl_long>(addr); *(cl_long *)out = x; v = x; } break;
				}
			}
			else if (valid && cm.attlen > 0 && cm.attlen <= 8)
			{
				/* NULL slot holds zero: reads of it are deterministic */
				char *out = (char *)dst + s_values_off[c] + (size_t)cm.attlen * row;
				for (int b = 0; b < cm.attlen; b++)
					out[b] = 0;
			}
			/* not-null bitmap: one ballot -> two 32-bit words per wave */
			strom_lanemask_t nn = __ballot(addr != NULL);
			strom_lanemask_t vv = __ballot(valid);
			if (s_nulls_off[c] != 0 && vv != 0)
			{
				cl_uint *words = (cl_uint *)((char *)dst + s_nulls_off[c]);
				cl_uint	 w0 = (base + (threadIdx.x & ~63u)) >> 5;
				if (lane == 0)
					words[w0] = (cl_uint)nn;
				if (lane == 32 && (vv >> 32) != 0)
					words[w0 + 1] = (cl_uint)(nn >> 32);
			}
			if (nn != vv && lane == 0)
				s_hasnull[c] = 1;
		}
	}
	__syncthreads();
	for (cl_uint c = threadIdx.x; c < ncols; c += blockDim.x)
	{
		if (s_hasnull[c])
			col_has_null[c] = 1;
	}
	if (threadIdx.x == 0 && s_failed)
		col_has_null[ncols] = 1;			/* slot after the flags: conversion failed */
}

extern "C" __global__ void
__launch_bounds__(256)
ingest_to_column(const kern_data_store *__restrict__ src, kern_data_store *__restrict__ dst,
				 const cl_int *__restrict__ type_oids, cl_uint *__restrict__ col_has_null)
{
	ingest_to_column_body<false>(src, dst, type_oids, col_has_null);
}

extern "C" __global__ void
__launch_bounds__(256)
ingest_to_column_varlena(const kern_data_store *__restrict__ src, kern_data_store *__restrict__ dst,
						 const cl_int *__restrict__ type_oids, cl_uint *__restrict__ col_has_null)
{
	ingest_to_column_body<true, true>(src, dst, type_oids, col_has_null);
}

extern "C" __global__ void
__launch_bounds__(256)
ingest_to_column_varnum(const kern_data_store *__restrict__ src, kern_data_store *__restrict__ dst,
						const cl_int *__restrict__ type_oids, cl_uint *__restrict__ col_has_null)
{
	ingest_to_column_body<true>(src, dst, type_oids, col_has_null);
}

/*
 * (Tried and removed in round 2: heap pages staged in LDS -- 4 pages per 256-row tile,
 * coalesced 16-byte loads, tuples walked with ds_read.  185 us per 1e7 rows against 140 us
 * for the walk through global memory above: 34 KB of LDS per work-group halve the waves a
 * CU holds, and it was those waves that hid the walk's dependent loads.
 * profiles/r02_ingest_staged_probe.txt)
 */

/*
 * zone maps in a pass of their own over the transposed columns (coalesced
 * 4-12 B/row, mostly still in L2 / Infinity Cache): blockIdx.y = column.  A
 * thread keeps a running min / max of the order-preserving u64 image of
 * its rows (sign bit flipped for integers; the usual IEEE trick for floats,
 * NaN left out); one wave reduction and one atomic pair per work-group at
 * the end.  Doing this inside the transpose cost a 64-lane reduction per
 * column and wave.
 */
extern "C" __global__ void
__launch_bounds__(256)
ingest_minmax(kern_data_store *dst, const cl_int *type_oids)
{
	__shared__ cl_ulong	s_min, s_max;
	kern_coldir *coldir = KERN_DATA_STORE_COLDIR(dst);
	cl_uint		c = blockIdx.y;
	cl_uint		nitems = dst->nitems;
	cl_int		oid = (type_oids ? type_oids[c] : 0);
	int			attlen = dst->colmeta[c].attlen;

	if (oid == 0 || !(attlen == 1 || attlen == 2 || attlen == 4 || attlen == 8))
		return;
	bool		isflt = (oid == STROM_FLOAT4OID || oid == STROM_FLOAT8OID);
	const char *values = (const char *)dst + coldir[c].values_off;
	const cl_uint *notnull = (coldir[c].nulls_off != 0
							  ? (const cl_uint *)((const char *)dst + coldir[c].nulls_off) : NULL);
	cl_ulong	mn = ~0UL, mx = 0UL;

	if (threadIdx.x == 0)
	{
		s_min = ~0UL;
		s_max = 0UL;
	}
	__syncthreads();
	if (oid == STROM_NUMERICOID)
	{
		/*
		 * 64-bit numeric images: the bounds are of the VALUES' integer parts, rounded outward
		 * (KDS_COLSTAT_INTPART, strom_kds.h) -- the images' own bit patterns do not order like the
		 * values.  A value beyond int64 spoils the column's bounds (the full range: dropped by
		 * ingest_finish).
		 */
		if (attlen != 8)
			return;
		for (cl_uint row = blockIdx.x * blockDim.x + threadIdx.x; row < nitems; row += gridDim.x * blockDim.x)
		{
			if (notnull && !((notnull[row >> 5] >> (row & 31)) & 1))
				continue;
			cl_ulong	image = ((const cl_ulong *)values)[row];
			cl_int		expo = (cl_int)((cl_long)image >> 58);
			bool		sign = ((image >> 57) & 1) != 0;
			cl_ulong	m = image & ((1UL << 57) - 1);
			bool		fits = true;
			if (expo >= 0)
			{
				for (cl_int i = 0; i < expo && fits; i++)
				{
					fits = (m <= 0x7fffffffffffffffUL / 10);
					m *= 10;
				}
			}
			else
			{
				cl_ulong	d = 1;
				cl_int		i = 0;
				for (; i < -expo && d <= 0xffffffffffffffffUL / 10; i++)
					d *= 10;
				m = (i < -expo ? (m != 0 ? 1UL : 0UL) : (m / d + (m % d != 0 ? 1UL : 0UL)));
			}
			if (!fits || m > 0x7fffffffffffffffUL)
			{
				mn = 0UL;
				mx = ~0UL;
				continue;
			}
			cl_ulong	key = (cl_ulong)(sign ? -(cl_long)m : (cl_long)m) ^ 0x8000000000000000UL;
			mn = (key < mn ? key : mn);
			mx = (key > mx ? key : mx);
		}
	}
	else
	{
	/*
	 * 16 bytes per lane and load (the value arrays are 256-byte aligned), four loads in flight,
	 * ONE work-group per CU and column: the pass ends in an atomic min / max pair per work-group
	 * on the column's directory entry, and same-address atomics are served at tens of millions
	 * per second -- with 1024 work-groups per column they, not the 120 MB read, were the 74 us
	 * this pass took (profiles/r02_ingest_staged_probe.txt)
	 */
	typedef cl_uint ingest_vec4 __attribute__((ext_vector_type(4)));
	const cl_uint	per_vec = 16u / (cl_uint)attlen;			/* rows per 16 bytes */
	const cl_uint	nvec = nitems / per_vec;
	auto fold = [&](cl_long v, cl_uint row)
	{
		bool		ok = (!notnull || ((notnull[row >> 5] >> (row & 31)) & 1));
		cl_ulong	key;
		if (isflt)
		{
			cl_double d = (attlen == 4 ? (cl_double)__int_as_float((cl_int)v)
						   : __longlong_as_double((long long)v));
			cl_ulong bits = (cl_ulong)__double_as_longlong(d);
			ok = ok && !__builtin_isnan(d);
			key = (bits & 0x8000000000000000UL) ? ~bits : (bits | 0x8000000000000000UL);
		}
		else
			key = (cl_ulong)v ^ 0x8000000000000000UL;
		if (ok)
		{
			mn = (key < mn ? key : mn);
			mx = (key > mx ? key : mx);
		}
	};
	auto fold_vec = [&](ingest_vec4 q, cl_uint i)
	{
		cl_uint		row0 = i * per_vec;
		cl_uint		w[4] = { q.x, q.y, q.z, q.w };
		switch (attlen)
		{
			case 1:
				_Pragma("unroll")
				for (int k = 0; k < 16; k++)
					fold((cl_long)(cl_char)(w[k >> 2] >> ((k & 3) * 8)), row0 + k);
				break;
			case 2:
				_Pragma("unroll")
				for (int k = 0; k < 8; k++)
					fold((cl_long)(cl_short)(w[k >> 1] >> ((k & 1) * 16)), row0 + k);
				break;
			case 4:
				_Pragma("unroll")
				for (int k = 0; k < 4; k++)
					fold((cl_long)(cl_int)w[k], row0 + k);
				break;
			default:
				_Pragma("unroll")
				for (int k = 0; k < 2; k++)
					fold((cl_long)(((cl_ulong)w[2 * k + 1] << 32) | w[2 * k]), row0 + k);
				break;
		}
	};
	const ingest_vec4 *vecs = (const ingest_vec4 *)values;
	cl_uint		stride = gridDim.x * blockDim.x;
	cl_uint		i = blockIdx.x * blockDim.x + threadIdx.x;
	for (; i + 3 * stride < nvec; i += 4 * stride)
	{
		ingest_vec4 q0 = vecs[i];
		ingest_vec4 q1 = vecs[i + stride];
		ingest_vec4 q2 = vecs[i + 2 * stride];
		ingest_vec4 q3 = vecs[i + 3 * stride];
		fold_vec(q0, i);
		fold_vec(q1, i + stride);
		fold_vec(q2, i + 2 * stride);
		fold_vec(q3, i + 3 * stride);
	}
	for (; i < nvec; i += stride)
		fold_vec(vecs[i], i);
	/* the rows behind the last whole vector */
	for (cl_uint row = nvec * per_vec + blockIdx.x * blockDim.x + threadIdx.x; row < nitems; row += stride)
	{
		cl_long		v;
		switch (attlen)
		{
			case 1: v = ((const cl_char *)values)[row]; break;
			case 2: v = ((const cl_short *)values)[row]; break;
			case 4: v = ((const cl_int *)values)[row]; break;
			default: v = ((const cl_long *)values)[row]; break;
		}
		fold(v, row);
	}
	}	/* (integer-like and float columns) */
#pragma unroll
	for (int m = 32; m > 0; m >>= 1)
	{
		cl_ulong o1 = ((cl_ulong)(cl_uint)__shfl_xor((cl_int)(mn >> 32), m, 64) << 32) |
			(cl_uint)__shfl_xor((cl_int)mn, m, 64);
		cl_ulong o2 = ((cl_ulong)(cl_uint)__shfl_xor((cl_int)(mx >> 32), m, 64) << 32) |
			(cl_uint)__shfl_xor((cl_int)mx, m, 64);
		mn = (o1 < mn ? o1 : mn);
		mx = (o2 > mx ? o2 : mx);
	}
	if ((threadIdx.x & 63) == 0 && mn <= mx)
	{
		atomicMin((unsigned long long *)&s_min, (unsigned long long)mn);
		atomicMax((unsigned long long *)&s_max, (unsigned long long)mx);
	}
	__syncthreads();
	if (threadIdx.x == 0 && s_min <= s_max)
	{
		atomicMin((unsigned long long *)&coldir[c].minval, (unsigned long long)s_min);
		atomicMax((unsigned long long *)&coldir[c].maxval, (unsigned long long)s_max);
	}
}

/* after the pass: drop the bitmap of columns that turned out NULL-free and
 * decode the zone map (the host seeded minval = ~0, maxval = 0) */
extern "C" __global__ void
ingest_finish(kern_data_store *dst, const cl_int *type_oids, const cl_uint *col_has_null)
{
	kern_coldir *coldir = KERN_DATA_STORE_COLDIR(dst);
	/* a heap area was sized for the worst case: the chunk ends where its last datum does */
	if (threadIdx.x == 0 && dst->usage != 0)
	{
		cl_uint	end = (cl_uint)STROM_TYPEALIGN(KDS_COLUMN_ALIGN, dst->usage);
		if (end >= dst->usage && end < dst->length)
			dst->length = end;
	}
	for (cl_uint c = threadIdx.x; c < dst->ncols; c += blockDim.x)
	{
		if (!col_has_null[c])
			coldir[c].nulls_off = 0;
		cl_int		oid = (type_oids ? type_oids[c] : 0);
		bool		isflt = (oid == STROM_FLOAT4OID || oid == STROM_FLOAT8OID);
		cl_ulong	mn = (cl_ulong)coldir[c].minval;
		cl_ulong	mx = (cl_ulong)coldir[c].maxval;

		if (oid == 0 || mn > mx || (oid == STROM_NUMERICOID && mn == 0UL && mx == ~0UL))
		{
			coldir[c].stat_flags = 0;
			coldir[c].minval = coldir[c].maxval = 0;
		}
		else if (oid == STROM_NUMERICOID)
		{
			coldir[c].stat_flags = KDS_COLSTAT_INTPART;
			coldir[c].minval = (cl_long)(mn ^ 0x8000000000000000UL);
			coldir[c].maxval = (cl_long)(mx ^ 0x8000000000000000UL);
		}
		else if (isflt)
		{
			coldir[c].stat_flags = KDS_COLSTAT_MINMAX | KDS_COLSTAT_ISFLOAT;
			coldir[c].minval = (cl_long)((mn & 0x8000000000000000UL) ? (mn & 0x7fffffffffffffffUL) : ~mn);
			coldir[c].maxval = (cl_long)((mx & 0x8000000000000000UL) ? (mx & 0x7fffffffffffffffUL) : ~mx);
		}
		else
		{
			coldir[c].stat_flags = KDS_COLSTAT_MINMAX;
			coldir[c].minval = (cl_long)(mn ^ 0x8000000000000000UL);
			coldir[c].maxval = (cl_long)(mx ^ 0x8000000000000000UL);
		}
	}
}

#endif	/* STROM_INGEST_DEVICE_H */
