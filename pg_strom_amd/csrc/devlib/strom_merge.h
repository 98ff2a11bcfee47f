/*
 * strom_merge.h -- device side of the multi-GPU GpuPreAgg merge, plus the
 * streaming-read probe the bench uses as its measured HBM ceiling.
 *
 * The reference has no collective: one backend's Agg
 * node adds up the partial rows of all chunks (pg_strom--1.0.sql:247-401).
 * Here every rank folds its row range into a resident table of identical
 * dense layout -- section 0: one u32 flags word per group (bit 0 "seen",
 * bit 1+a "aggregate a has a value"), section 1+a: one 8-byte value per group
 * -- and the tables are merged in place by RCCL all-reduces, one per section
 * (csrc/parallel.cpp).  RCCL knows SUM / MIN / MAX on int64 and double; the
 * kernels below bring every section into a form those operators merge
 * correctly and back:
 *
 *   op 0  nrows                SUM int64 as is
 *   op 1  psum, integer        the table keeps such a sum 128 bits wide {low word in section
 *                              1+a, high word in a section of its own: hi_off} so that it
 *                              never wraps (strom_gpupreagg.h, "integer sums never wrap") --
 *                              and neither may the merge: the value travels as THREE
 *                              carry-free limbs, low 32 bits | next 32 bits | high word, each
 *                              summed as int64 (2^32 x 2^31 ranks < 2^63), recombined with
 *                              carries afterwards.  Entries without a value -> 0.
 *   op 2  psum, float8         SUM double, entries without a value -> 0.0
 *   op 3/4 pmin/pmax, integer  MIN/MAX int64, entries without a value -> identity
 *   op 5/6 pmin/pmax, float    the table keeps floats as order-preserving
 *                              UNSIGNED keys (strom_gpupreagg.h); flipping the
 *                              sign bit makes signed MIN/MAX order them
 *   flags                      bitwise OR = MAX over the bits unpacked to bytes
 *
 * Nothing here depends on generated code: a fixed-function program.
 */
#ifndef STROM_MERGE_DEVICE_H
#define STROM_MERGE_DEVICE_H

#define PREAGG_MERGE_MAXAGGS	31

typedef struct {
	cl_uint		ngroups;
	cl_uint		naggs;
	cl_uint		op[PREAGG_MERGE_MAXAGGS];
	cl_uint		__pad;
	cl_ulong	vals_off[PREAGG_MERGE_MAXAGGS];		/* byte offset of section 1+a in the table */
	cl_ulong	hi_off[PREAGG_MERGE_MAXAGGS];		/* op 1: byte offset of the sum's high-word section */
	cl_uint		mid_idx[PREAGG_MERGE_MAXAGGS];		/* op 1: which ngroups-long lane of 'mid' takes bits 32..63 */
	cl_uint		__pad2;
} preagg_merge_spec;

#define PREAGG_MERGE_SIGN	0x8000000000000000UL

extern "C" __global__ void
__launch_bounds__(256)
preagg_merge_prepare(char *table, const preagg_merge_spec *spec, cl_uchar *bits, cl_ulong *mid)
{
	cl_uint		ngroups = spec->ngroups;
	cl_uint		naggs = spec->naggs;
	cl_uint		nbits = naggs + 1;
	const cl_uint *flags = (const cl_uint *)table;

	for (cl_uint g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x)
	{
		cl_uint		f = flags[g];
		for (cl_uint b = 0; b < nbits; b++)
			bits[(size_t)b * ngroups + g] = (cl_uchar)((f >> b) & 1);
		for (cl_uint a = 0; a < naggs; a++)
		{
			cl_ulong   *vals = (cl_ulong *)(table + spec->vals_off[a]);
			bool		has = ((f >> (1 + a)) & 1) != 0;
			cl_ulong	v = vals[g];
			switch (spec->op[a])
			{
				case 0:		break;
				case 1:
				{
					/* {lo, hi} -> limbs: lo & 2^32-1 (stays here) | lo >> 32 (mid) | hi (its section) */
					cl_long	   *hi = (cl_long *)(table + spec->hi_off[a]);
					if (!has)
					{
						v = 0;
						hi[g] = 0;
					}
					mid[(size_t)spec->mid_idx[a] * ngroups + g] = v >> 32;
					v &= 0xffffffffUL;
					break;
				}
				case 2:		if (!has) v = 0; break;
				case 3:		if (!has) v = 0x7fffffffffffffffUL; break;
				case 4:		if (!has) v = PREAGG_MERGE_SIGN; break;
				case 5:		v = (has ? (v ^ PREAGG_MERGE_SIGN) : 0x7fffffffffffffffUL); break;
				default:	v = (has ? (v ^ PREAGG_MERGE_SIGN) : PREAGG_MERGE_SIGN); break;
			}
			vals[g] = v;
		}
	}
}

extern "C" __global__ void
__launch_bounds__(256)
preagg_merge_finish(char *table, const preagg_merge_spec *spec, const cl_uchar *bits, const cl_ulong *mid)
{
	cl_uint		ngroups = spec->ngroups;
	cl_uint		naggs = spec->naggs;
	cl_uint		nbits = naggs + 1;
	cl_uint	   *flags = (cl_uint *)table;

	for (cl_uint g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x)
	{
		cl_uint		f = 0;
		for (cl_uint b = 0; b < nbits; b++)
			f |= (bits[(size_t)b * ngroups + g] != 0 ? (1u << b) : 0u);
		flags[g] = f;
		for (cl_uint a = 0; a < naggs; a++)
		{
			cl_ulong   *vals = (cl_ulong *)(table + spec->vals_off[a]);
			bool		has = ((f >> (1 + a)) & 1) != 0;
			cl_uint		op = spec->op[a];
			if (op == 1)
			{
				/* limbs (each a sum over the ranks, none of them wrapped) -> {lo, hi} */
				cl_long	   *hi = (cl_long *)(table + spec->hi_off[a]);
				cl_ulong	l0 = vals[g];
				cl_ulong	t = mid[(size_t)spec->mid_idx[a] * ngroups + g] + (l0 >> 32);
				vals[g] = (has ? ((l0 & 0xffffffffUL) | (t << 32)) : 0UL);
				hi[g] = (has ? hi[g] + (cl_long)(t >> 32) : 0L);
			}
			if (op >= 3)
			{
				/* no rank had a value: back to the table's "untouched = 0" */
				cl_ulong v = vals[g];
				if (!has)
					v = 0;
				else if (op >= 5)
					v ^= PREAGG_MERGE_SIGN;
				vals[g] = v;
			}
		}
	}
}

/*
 * dst = dst (op) src, element by element: what ONE RCCL all-reduce of a lane does between two
 * ranks.  strom_gpupreagg_merge() adds the tables of two sessions of one GPU with exactly the
 * lanes, operators, prepare and finish steps strom_gpupreagg_allreduce() hands to RCCL -- which is
 * also how that path is tested on a single GPU.  kind: 0 SUM int64, 1 SUM float64, 2 MIN int64,
 * 3 MAX int64, 4 MAX uint8 (count bytes).
 */
extern "C" __global__ void
__launch_bounds__(256)
preagg_merge_apply(void *dst, const void *src, cl_ulong count, cl_uint kind)
{
	for (cl_ulong i = (cl_ulong)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (cl_ulong)gridDim.x * blockDim.x)
	{
		if (kind == 4)
		{
			cl_uchar a = ((cl_uchar *)dst)[i], b = ((const cl_uchar *)src)[i];
			((cl_uchar *)dst)[i] = (a > b ? a : b);
		}
		else if (kind == 1)
			((cl_double *)dst)[i] += ((const cl_double *)src)[i];
		else
		{
			cl_long a = ((cl_long *)dst)[i], b = ((const cl_long *)src)[i];
			((cl_long *)dst)[i] = (kind == 0 ? (cl_long)((cl_ulong)a + (cl_ulong)b) : kind == 2 ? (a < b ? a : b) : (a > b ? a : b));
		}
	}
}

/*
 * partial rows of a dense session, formatted on the device: one TUPSLOT row (8-byte datums, then
 * one NULL byte per column) per group that was seen, packed in no particular order -- what
 * strom_gpupreagg_fetch hands to the caller, without the host loop over the table (190-290 us for
 * 1e4 groups: more than the fold of a reference-sized chunk, profiles/r02_chunk_message_probe.txt).
 * Group keys come back from the dense id (key_min + offset; the NULL slot is offset == range).
 * counter[0] counts the rows (written only below max_rows: max_rows = 0 just counts), counter[1] is
 * set when a 128-bit integer sum does not fit its int8 datum -- the host then splits that table
 * into several rows itself (gpupreagg.cpp: sum_pieces).  Column kinds: 0 key, 1 nrows, 2 integer
 * value of len bytes, 3 float8 sum (float4: narrowed), 4 float min / max (order-preserving key),
 * 5 integer sum {vals_off: low word, hi_off: high word}.
 */
typedef struct {
	cl_uint		ngroups;
	cl_uint		ncols;
	cl_uint		stride;
	cl_uint		nkeys;
	cl_long		key_min[8];
	cl_uint		key_range[8];
	cl_uint		key_stride[8];
	struct {
		cl_uint		kind;
		cl_uint		len;			/* bytes of the datum */
		cl_uint		which;			/* key number, or the aggregate's has-value bit (1 + a) */
		cl_uint		float4;
		cl_ulong	vals_off;
		cl_ulong	hi_off;
	} col[64];
} preagg_export_spec;

extern "C" __global__ void
__launch_bounds__(256)
preagg_dense_export_rows(const char *table, const preagg_export_spec *spec, char *rows, cl_uint max_rows,
						 cl_uint *counter)
{
	cl_uint		N = spec->ngroups;
	cl_uint		ncols = spec->ncols;
	cl_uint		stride = spec->stride;
	const cl_uint *t_flags = (const cl_uint *)table;

	for (cl_uint base = blockIdx.x * blockDim.x; base < N; base += gridDim.x * blockDim.x)
	{
		cl_uint		g = base + threadIdx.x;
		cl_uint		flags = (g < N ? t_flags[g] : 0u);
		bool		seen = (flags & 1u) != 0;
		cl_ulong	mask = __ballot(seen);
		cl_uint		first = 0;
		if (mask == 0)
			continue;
		if ((threadIdx.x & 63) == 0)
			first = atomicAdd(&counter[0], (cl_uint)__popcll(mask));
		first = __shfl(first, 0, 64);
		cl_uint		idx = first + (cl_uint)__popcll(mask & ((1UL << (threadIdx.x & 63)) - 1));
		if (!seen)
			continue;
		/* a sum beyond one int8 datum is reported whether or not the row is written (the size
		 * query counts with max_rows = 0 and must send the caller the host's way just the same) */
		for (cl_uint c = 0; c < ncols; c++)
		{
			if (spec->col[c].kind == 5 && ((flags >> spec->col[c].which) & 1u))
			{
				cl_long lo = ((const cl_long *)(table + spec->col[c].vals_off))[g];
				cl_long hi = ((const cl_long *)(table + spec->col[c].hi_off))[g];
				if (hi != (lo >> 63))
					counter[1] = 1;
			}
		}
		if (idx >= max_rows)
			continue;
		cl_ulong   *values = (cl_ulong *)(rows + (size_t)stride * idx);
		cl_char	   *isnull = (cl_char *)(values + ncols);
		for (cl_uint w = ncols; w < stride / 8; w++)
			values[w] = 0;						/* the NULL flags and the padding behind them */
		for (cl_uint c = 0; c < ncols; c++)
		{
			cl_uint		kind = spec->col[c].kind, len = spec->col[c].len, which = spec->col[c].which;
			cl_ulong	raw = 0;
			bool		null = false;
			if (kind == 0)
			{
				cl_uint	off = (g / spec->key_stride[which]) % (spec->key_range[which] + 1);
				null = (off == spec->key_range[which]);
				raw = (cl_ulong)(spec->key_min[which] + (cl_long)off);
			}
			else
			{
				raw = ((const cl_ulong *)(table + spec->col[c].vals_off))[g];
				null = (kind != 1 && !((flags >> which) & 1u));
				if (kind == 4 && !null)
					raw = (raw & PREAGG_MERGE_SIGN) ? (raw & 0x7fffffffffffffffUL) : ~raw;
				if ((kind == 3 || kind == 4) && spec->col[c].float4 && !null)
				{
					cl_double d;
					cl_float f;
					__builtin_memcpy(&d, &raw, 8);
					f = (cl_float)d;
					raw = 0;
					__builtin_memcpy(&raw, &f, 4);
				}
			}
			if (!null && len < 8)
				raw &= (1UL << (8 * len)) - 1;
			values[c] = (null ? 0UL : raw);
			if (null)
				isnull[c] = 1;
		}
	}
}

/* census bitmaps (one bit per dense id, gpupreagg_census): OR over the ranks
 * = MAX over the bits unpacked to bytes */
extern "C" __global__ void
__launch_bounds__(256)
preagg_census_unpack(const cl_uint *bitmap, cl_uint nbits, cl_uchar *bytes)
{
	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x; i < nbits; i += gridDim.x * blockDim.x)
		bytes[i] = (cl_uchar)((bitmap[i >> 5] >> (i & 31)) & 1);
}

extern "C" __global__ void
__launch_bounds__(256)
preagg_census_pack(cl_uint *bitmap, cl_uint nbits, const cl_uchar *bytes)
{
	cl_uint		nwords = (nbits + 31) / 32;
	for (cl_uint w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x)
	{
		cl_uint		word = 0;
		for (cl_uint b = 0; b < 32 && w * 32 + b < nbits; b++)
			word |= (bytes[w * 32 + b] != 0 ? (1u << b) : 0u);
		bitmap[w] = word;
	}
}

/*
 * Streaming-read probe ("measure the achievable peak on
 * the box with a read-only streaming kernel and report both nominal and
 * measured denominators").  Every work-group walks 16 KB tiles (256 threads x
 * four 16-byte non-temporal loads, all issued before the first use), XORs
 * what it read and writes ONE word per work-group so that the loads cannot be
 * dropped.  nbytes is a multiple of 16.
 */
typedef cl_uint membw_vec_t __attribute__((ext_vector_type(4)));

extern "C" __global__ void
__launch_bounds__(256)
membw_stream_read(const membw_vec_t *src, cl_ulong nvec, cl_uint *sink)
{
	cl_ulong	tile = 1024;						/* vectors per work-group and turn */
	membw_vec_t	acc = {0, 0, 0, 0};

	for (cl_ulong base = (cl_ulong)blockIdx.x * tile; base < nvec; base += (cl_ulong)gridDim.x * tile)
	{
		if (base + tile <= nvec)
		{
			membw_vec_t	v0 = __builtin_nontemporal_load(src + base + threadIdx.x);
			membw_vec_t	v1 = __builtin_nontemporal_load(src + base + 256 + threadIdx.x);
			membw_vec_t	v2 = __builtin_nontemporal_load(src + base + 512 + threadIdx.x);
			membw_vec_t	v3 = __builtin_nontemporal_load(src + base + 768 + threadIdx.x);
			acc ^= v0 ^ v1 ^ v2 ^ v3;
		}
		else
		{
			for (cl_ulong i = base + threadIdx.x; i < nvec; i += 256)
			{
				acc ^= __builtin_nontemporal_load(src + i);
			}
		}
	}
	cl_uint		x = acc.x ^ acc.y ^ acc.z ^ acc.w;
	/* one store per work-group, and only when the data says so: keeps the
	 * loads alive without a write stream */
	for (int off = 32; off > 0; off >>= 1)
		x ^= __shfl_xor(x, off);
	if ((threadIdx.x & 63) == 0 && x == 0x9e3779b9u)
		sink[blockIdx.x] = x;
}

#endif	/* STROM_MERGE_DEVICE_H */
