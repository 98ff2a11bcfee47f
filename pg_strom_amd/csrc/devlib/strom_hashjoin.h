/*
 * strom_hashjoin.h -- GpuHashJoin kernels (HIP, gfx950)
 *
 * Role in the reference: opencl_hashjoin.h -- kern_hashentry /
 * kern_hashtable / kern_multihash accessors (102-192), kern_gpuhashjoin_main
 * (284-416: count pass, reserve result slots with one atomic per work-group,
 * emit pass, StromError_DataStoreNoSpace when the buffer is short) and the
 * CRC32 hash-key templates (844-953).
 *
 * What is different:
 *   - the inner side arrives as the reference's kern_multihash (whole
 *     inner heap tuples in kern_hashentry chains).  The runtime keeps a
 *     private copy in HBM and RE-LINKS it once per join
 *     (hashjoin_build_index): entries are chained from a probe index of its
 *     own instead of slot = crc32 % nslots.  Two index forms:
 *       DIRECT  one integer-like key whose value range is dense:
 *               slots[key - min] -> first entry.  4 bytes per key value:
 *               a 1e6-key dimension is a 4 MB table that lives in L2; no
 *               hashing, no key compare, no touch of the 48-byte entries.
 *       KEYED   one key of any type and spread (sparse int8, numeric, float):
 *               open addressing over 16-byte slots {first entry, tag, key
 *               image}, ONE slot per distinct key -- a probe is one 16-byte
 *               load that carries the key to compare with, so a lookup of a
 *               unique key touches no 48-byte entry at all (the HASH form read
 *               one per candidate: 7.8 ms per 1e8 rows for C3; this one ~1 ms);
 *               the chain hanging off a slot holds exactly that key's entries.
 *       HASH    several keys: slots[mix(key images) & mask], entry->hash
 *               rewritten to the same mix, chain walked with key compare.
 *   - result records keep the reference's meaning: {outer_row + 1, byte
 *     offset of the matched kern_hashentry inside its kern_hashtable}.
 *   - gpuhashjoin_main_fast: DIRECT + no duplicate keys + COLUMN outer: one
 *     pass (a row matches at most once), rows streamed with 16-byte loads,
 *     all slot reads of a tile issued before the first use, ballot/mbcnt
 *     compaction into an LDS stage flushed with one atomic -- the GpuScan
 *     structure with 8-byte records.
 *   - gpuhashjoin_main: any format / row map / duplicates / several inner
 *     relations: two passes per tile like the reference, but one result
 *     reservation per 2048 rows instead of one per work-group of rows.
 */
#ifndef STROM_HASHJOIN_DEVICE_H
#define STROM_HASHJOIN_DEVICE_H

#ifndef HASHJOIN_BLOCK
#define HASHJOIN_BLOCK		256
#endif
#ifndef HASHJOIN_QUADS
#define HASHJOIN_QUADS		2
#endif
#ifndef HASHJOIN_STAGE
#define HASHJOIN_STAGE		4096		/* records (8 bytes each) */
#endif
/* the one-pass kernels over the L2-resident slot arrays: 4096-row tiles (16 slot reads in flight
 * per thread): 429 against 440 us per 1e8 rows for BASELINE configs[2] with 2048-row tiles, 446
 * with 8192 (profiles/r03_c3_sweep.txt) */
#ifndef HASHJOIN_FAST_QUADS
#define HASHJOIN_FAST_QUADS	4
#endif
#define HASHJOIN_NWAVES		(HASHJOIN_BLOCK / STROM_WAVE)
#define HASHJOIN_TILE_ROWS	(HASHJOIN_BLOCK * 4 * HASHJOIN_QUADS)
#define HASHJOIN_MAXRELS	8

#define HASHJOIN_MODE_HASH		0
#define HASHJOIN_MODE_DIRECT	1
#define HASHJOIN_MODE_KEYED		2

/* KEYED slot: { x = offset of the key's first entry, y = tag (0 empty, 1 being
 * written by the index build, 2 ready), z|w = the key's canonical 64-bit image } */
typedef cl_uint hashjoin_keyed_slot __attribute__((ext_vector_type(4)));
#define HASHJOIN_KEYED_EMPTY	0u
#define HASHJOIN_KEYED_BUSY		1u
#define HASHJOIN_KEYED_READY	2u

/* probe index; gpuhashjoin.cpp mirrors these structs */
struct hashjoin_index_rel {
	cl_uint		mode;
	cl_uint		nslots;			/* HASH / KEYED: power of two; DIRECT: key range */
	cl_long		key_min;
	cl_uint		unique;			/* no chain longer than one entry */
	cl_uint		slots_off;		/* bytes from the index base to cl_uint slots[] */
	cl_uint		nentries;
	/*
	 * DIRECT + unique keys: the same slots in THREE bytes each -- (entry offset >> 3; entries are
	 * LONGALIGNed, KERN_HASHENTRY_SIZE_BY_TLEN) -- when the table is below 2^27 bytes; 0 = none.
	 * A slot array is probed at random by every CU of an XCD: what counts is whether it fits that
	 * XCD's 4 MB L2 next to the stream.  1.25e6 key values (BASELINE configs[2]) are 5.0 MB as
	 * cl_uint -- one probe in three went to HBM for a 64-byte line -- and 3.75 MB like this.
	 * Made by hashjoin_narrow_slots_kernel, read by gpuhashjoin_main_fast_narrow.
	 */
	cl_uint		slots3_off;
};
struct hashjoin_index {
	cl_uint		nrels;
	cl_uint		__pad[3];
	hashjoin_index_rel rel[HASHJOIN_MAXRELS];
};
struct hashjoin_build_stats {
	cl_long		key_min;
	cl_long		key_max;
	cl_uint		nentries;
	cl_uint		intlike;
};

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

/* canonical 64-bit image of a key value: equal values <=> equal images */
STROM_DEVICE cl_ulong hashjoin_key_image(cl_bool v)		{ return (cl_ulong)(v != 0); }
STROM_DEVICE cl_ulong hashjoin_key_image(cl_short v)	{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong hashjoin_key_image(cl_int v)		{ return (cl_ulong)(cl_long)v; }
STROM_DEVICE cl_ulong hashjoin_key_image(cl_long v)		{ return (cl_ulong)v; }
STROM_DEVICE cl_ulong hashjoin_key_image(cl_ulong v)	{ return v; }
STROM_DEVICE cl_ulong hashjoin_key_image(cl_double v)
{
	if (__builtin_isnan(v))
		return 0x7ff8000000000000UL;
	if (v == 0.0)
		v = 0.0;				/* -0 == +0 */
	return (cl_ulong)__double_as_longlong(v);
}
STROM_DEVICE cl_ulong hashjoin_key_image(cl_float v)	{ return hashjoin_key_image((cl_double)v); }

#ifdef STROM_TEXTLIB_DEVICE_H
/*
 * text / character(n) hash keys (STROMCL_VARLENA_HASHKEY_TEMPLATE, opencl_hashjoin.h:935-953: the
 * hash runs over VARDATA_ANY / VARSIZE_ANY_EXHDR): the value is the address of the datum, the image
 * a 64-bit mix of its payload, eight bytes a step -- equal strings have equal images, NOT the other way round, so a
 * relation with such a key gets the HASH index and every candidate is compared with texteq /
 * bpchareq (codegen_hashjoin.cpp: image_is_exact).  character(n): trailing blanks do not count
 * (bpchar_truelen, opencl_textlib.h:154-166), on either side.
 */
STROM_DEVICE cl_ulong
hashjoin_varlena_image(cl_ulong datum, bool blank_padded)
{
	cl_int		len;
	const cl_uchar *p = strom_varlena_payload(datum, &len);
	cl_ulong	h = 0xcbf29ce484222325UL ^ (cl_ulong)0;
	cl_int		i = 0;

	if (blank_padded)
		while (len > 0 && p[len - 1] == ' ')
			len--;
	/* eight bytes a step (strom_load_u64), the tail gathered into one last word */
	for (; i + 8 <= len; i += 8)
	{
		h = (h ^ strom_load_u64(p + i)) * 0x9e3779b97f4a7c15UL;
		h ^= h >> 29;
	}
	cl_ulong	tail = (cl_ulong)(cl_uint)len << 56;
	if (i < len && len >= 8)
	{
		/* the last (len - i) bytes out of one overlapping load of the string's last eight */
		tail ^= strom_load_u64(p + len - 8) >> (8 * (8 - (len - i)));
		i = len;
	}
	for (int sh = 0; i < len; i++, sh += 8)
		tail ^= (cl_ulong)p[i] << sh;
	h = (h ^ tail) * 0xbf58476d1ce4e5b9UL;
	h ^= h >> 31;
	return h;
}
#endif

STROM_DEVICE cl_uint
hashjoin_hash_images(const cl_ulong *images, int nkeys)
{
	cl_ulong h = 0x9e3779b97f4a7c15UL;
	for (int k = 0; k < nkeys; k++)
	{
		h ^= images[k];
		h *= 0xff51afd7ed558ccdUL;
		h ^= h >> 33;
	}
	h *= 0xc4ceb9fe1a85ec53UL;
	h ^= h >> 29;
	return (cl_uint)h;
}

STROM_DEVICE cl_uint
hashjoin_first(const hashjoin_index *hjidx, int d0, const cl_ulong *images, int nkeys, cl_uint *p_hash)
{
	const hashjoin_index_rel *ir = &hjidx->rel[d0];
	const cl_uint *slots = (const cl_uint *)((const char *)hjidx + ir->slots_off);

	if (ir->mode == HASHJOIN_MODE_DIRECT)
	{
		cl_long	idx = (cl_long)images[0] - ir->key_min;
		*p_hash = 0;
		if (idx < 0 || idx >= (cl_long)ir->nslots)
			return 0;
		return slots[idx];
	}
	cl_uint h = hashjoin_hash_images(images, nkeys);
	*p_hash = h;
	if (ir->mode == HASHJOIN_MODE_KEYED)
	{
		const hashjoin_keyed_slot *kslots = (const hashjoin_keyed_slot *)slots;
		cl_uint		mask = ir->nslots - 1;
		cl_uint		lo = (cl_uint)images[0], hi = (cl_uint)(images[0] >> 32);
		/* load factor <= 1/2: 1.5 slots on average; an empty slot ends the search */
		for (cl_uint p = h & mask, n = 0; n <= mask; p = (p + 1) & mask, n++)
		{
			hashjoin_keyed_slot s = kslots[p];			/* one 16-byte load: one L2 request */
			if (s.y == HASHJOIN_KEYED_EMPTY)
				return 0;
			if (s.z == lo && s.w == hi)
				return s.x;
		}
		return 0;
	}
	return slots[h & (ir->nslots - 1)];
}

STROM_DEVICE bool
hashjoin_candidate(const hashjoin_index *hjidx, int d0, const kern_hashentry *ent, cl_uint hash)
{
	return hjidx->rel[d0].mode != HASHJOIN_MODE_HASH || ent->hash == hash;
}

/* generated */
STROM_DEVICE bool
hashjoin_inner_key_images(int depth, const kern_hashtable *kht, const kern_hashentry *ent,
						  cl_ulong *images);
STROM_DEVICE bool
hashjoin_fast_outer_key(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV,
						cl_long *p_key);
#if HASHJOIN_FAST_OUTER_QUAL
STROM_DEVICE bool
hashjoin_fast_outer_qual(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV);
#endif
template <bool ALL_SINGLE>
STROM_DEVICE cl_uint
gpuhashjoin_execute(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV,
					const kern_multihash *__restrict__ kmhash, const hashjoin_index *__restrict__ hjidx,
					cl_uint kds_index, cl_int *__restrict__ rbuffer, cl_int *first_match = NULL);

STROM_DEVICE int
hashjoin_nkeys_of(int depth)
{
	switch (depth)
	{
#ifdef HASHJOIN_NKEYS_1
		case 1: return HASHJOIN_NKEYS_1;
#endif
#ifdef HASHJOIN_NKEYS_2
		case 2: return HASHJOIN_NKEYS_2;
#endif
#ifdef HASHJOIN_NKEYS_3
		case 3: return HASHJOIN_NKEYS_3;
#endif
#ifdef HASHJOIN_NKEYS_4
		case 4: return HASHJOIN_NKEYS_4;
#endif
#ifdef HASHJOIN_NKEYS_5
		case 5: return HASHJOIN_NKEYS_5;
#endif
#ifdef HASHJOIN_NKEYS_6
		case 6: return HASHJOIN_NKEYS_6;
#endif
#ifdef HASHJOIN_NKEYS_7
		case 7: return HASHJOIN_NKEYS_7;
#endif
#ifdef HASHJOIN_NKEYS_8
		case 8: return HASHJOIN_NKEYS_8;
#endif
	}
	return 0;
}

STROM_DEVICE int
hashjoin_key0_intlike(int depth)
{
	switch (depth)
	{
#ifdef HASHJOIN_KEY0_INTLIKE_1
		case 1: return HASHJOIN_KEY0_INTLIKE_1;
#endif
#ifdef HASHJOIN_KEY0_INTLIKE_2
		case 2: return HASHJOIN_KEY0_INTLIKE_2;
#endif
#ifdef HASHJOIN_KEY0_INTLIKE_3
		case 3: return HASHJOIN_KEY0_INTLIKE_3;
#endif
#ifdef HASHJOIN_KEY0_INTLIKE_4
		case 4: return HASHJOIN_KEY0_INTLIKE_4;
#endif
#ifdef HASHJOIN_KEY0_INTLIKE_5
		case 5: return HASHJOIN_KEY0_INTLIKE_5;
#endif
#ifdef HASHJOIN_KEY0_INTLIKE_6
		case 6: return HASHJOIN_KEY0_INTLIKE_6;
#endif
#ifdef HASHJOIN_KEY0_INTLIKE_7
		case 7: return HASHJOIN_KEY0_INTLIKE_7;
#endif
#ifdef HASHJOIN_KEY0_INTLIKE_8
		case 8: return HASHJOIN_KEY0_INTLIKE_8;
#endif
	}
	return 0;
}

/* ====================================================================== *
 * index build, step 1: entry count and key range of one inner relation.
 * One thread per slot of the HOST-built table walks that slot's chain.
 * ====================================================================== */
extern "C" __global__ void
hashjoin_build_stats_kernel(const kern_multihash *kmhash, int depth,
							hashjoin_build_stats *stats)
{
	const kern_hashtable *kht = KERN_HASHTABLE(kmhash, depth - 1);
	const cl_uint *slots = KERN_HASHTABLE_SLOT(kht);
	cl_long		kmin = 0x7fffffffffffffffL, kmax = -0x7fffffffffffffffL - 1;
	cl_uint		count = 0;

	for (cl_uint s = blockIdx.x * blockDim.x + threadIdx.x;
		 s < kht->nslots;
		 s += gridDim.x * blockDim.x)
	{
		for (cl_uint off = slots[s]; off != 0; )
		{
			const kern_hashentry *ent = (const kern_hashentry *)((const char *)kht + off);
			cl_ulong	images[8];

			if (off >= kht->length)
				break;			/* corrupt chain: stop rather than fault */
			count++;
			if (hashjoin_inner_key_images(depth, kht, ent, images))
			{
				cl_long v = (cl_long)images[0];
				kmin = (v < kmin ? v : kmin);
				kmax = (v > kmax ? v : kmax);
			}
			off = ent->next;
		}
	}
	if (count > 0)
	{
		atomicAdd(&stats->nentries, count);
		atomicMin((long long *)&stats->key_min, (long long)kmin);
		atomicMax((long long *)&stats->key_max, (long long)kmax);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
		stats->intlike = hashjoin_key0_intlike(depth);
}

/* ====================================================================== *
 * index build, step 2: re-link the entries of the private table copy
 * ====================================================================== */
extern "C" __global__ void
hashjoin_build_index_kernel(kern_multihash *kmhash, int depth, hashjoin_index *hjidx)
{
	kern_hashtable *kht = KERN_HASHTABLE(kmhash, depth - 1);
	const cl_uint *old_slots = KERN_HASHTABLE_SLOT(kht);
	hashjoin_index_rel *ir = &hjidx->rel[depth - 1];
	cl_uint	   *slots = (cl_uint *)((char *)hjidx + ir->slots_off);
	int			nkeys = hashjoin_nkeys_of(depth);

	for (cl_uint s = blockIdx.x * blockDim.x + threadIdx.x;
		 s < kht->nslots;
		 s += gridDim.x * blockDim.x)
	{
		for (cl_uint off = old_slots[s]; off != 0; )
		{
			kern_hashentry *ent = (kern_hashentry *)((char *)kht + off);
			cl_ulong	images[8];
			cl_uint		next_old;

			if (off >= kht->length)
				break;
			next_old = ent->next;		/* read before the entry is re-linked */
			if (!hashjoin_inner_key_images(depth, kht, ent, images))
			{
				ent->next = 0;			/* a NULL key can never join */
				ent->hash = 0;
			}
			else
			{
				cl_uint	h = 0, idx;
				cl_uint *head;				/* where this key's chain starts */
				if (ir->mode == HASHJOIN_MODE_DIRECT)
				{
					idx = (cl_uint)((cl_long)images[0] - ir->key_min);
					head = &slots[idx];
				}
				else if (ir->mode == HASHJOIN_MODE_KEYED)
				{
					/*
					 * find or claim THE slot of this key.  One loop without an inner
					 * wait: a thread that meets a slot another one is filling goes
					 * round again (the filler may be a lane of its own wave, which
					 * publishes in the same pass through the loop body)
					 */
					cl_uint	   *words = slots;			/* 4 words per slot */
					cl_uint		mask = ir->nslots - 1;
					cl_uint		lo = (cl_uint)images[0], hi = (cl_uint)(images[0] >> 32);
					h = hashjoin_hash_images(images, nkeys);
					idx = h & mask;
					for (;;)
					{
						cl_uint	   *slot = words + 4 * (size_t)idx;
						/* (no agent-scope acquire / release: those are a cache invalidate and an L2
						 * write-back per probe and per claim, DESIGN section 9.27.  Tag and key words
						 * are read and written with agent-scope atomics, served at the coherence
						 * point; the claimer waits for its key stores before it stores READY) */
						cl_uint		tag = STROM_PROBE_STATE(&slot[1]);
						if (tag == HASHJOIN_KEYED_EMPTY)
						{
							cl_uint expect = HASHJOIN_KEYED_EMPTY;
							if (__hip_atomic_compare_exchange_strong(&slot[1], &expect, HASHJOIN_KEYED_BUSY,
																	 __ATOMIC_RELAXED, __ATOMIC_RELAXED,
																	 __HIP_MEMORY_SCOPE_AGENT))
							{
								__hip_atomic_store(&slot[2], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
								__hip_atomic_store(&slot[3], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
								STROM_PUBLISH_STATE(&slot[1], HASHJOIN_KEYED_READY);	/* strom_common.h */
								break;
							}
							continue;				/* lost the race: look at the slot again */
						}
						if (tag == HASHJOIN_KEYED_BUSY)
							continue;
						if (__hip_atomic_load(&slot[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lo &&
							__hip_atomic_load(&slot[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == hi)
							break;					/* this key's slot */
						idx = (idx + 1) & mask;		/* another key lives here */
					}
					head = words + 4 * (size_t)idx;
				}
				else
				{
					h = hashjoin_hash_images(images, nkeys);
					idx = h & (ir->nslots - 1);
					head = &slots[idx];
				}
				ent->hash = h;
				cl_uint prev = atomicExch(head, off);
				ent->next = prev;
				if (prev != 0)
					ir->unique = 0;		/* benign race: every writer stores 0 */
			}
			off = next_old;
		}
	}
}

STROM_DEVICE void
hashjoin_load_kparams(strom_kparams &KP, const kern_parambuf *kparams, cl_int *errcode)
{
#define X(idx,NAME)	KP.KPARAM_##idx = pg_##NAME##_param(kparams, errcode, idx);
	STROM_KPARAM_LIST(X)
#undef X
	KP.__dummy = 0;
}

/* exclusive prefix of v over the work-group; *p_total = sum */
STROM_DEVICE cl_uint
hashjoin_block_scan(cl_uint v, cl_uint *lds_wave_totals, cl_uint *p_total)
{
	cl_uint	lane = threadIdx.x & (STROM_WAVE - 1);
	cl_uint	wave = threadIdx.x / STROM_WAVE;
	cl_uint	incl = v;

#pragma unroll
	for (int off = 1; off < STROM_WAVE; off <<= 1)
	{
		cl_uint o = __shfl_up(incl, off, STROM_WAVE);
		if ((int)lane >= off)
			incl += o;
	}
	if (lane == STROM_WAVE - 1)
		lds_wave_totals[wave] = incl;
	__syncthreads();
	cl_uint before = 0, total = 0;
#pragma unroll
	for (int w = 0; w < HASHJOIN_NWAVES; w++)
	{
		cl_uint t = lds_wave_totals[w];
		before += (w < (int)wave ? t : 0);
		total += t;
	}
	__syncthreads();
	*p_total = total;
	return before + incl - v;
}

/* ====================================================================== *
 * general probe: any format, row map, duplicates, several relations
 * ====================================================================== */
/* tile geometry of the general probe (see gpuhashjoin_main_body) */
#ifndef HASHJOIN_GENERIC_ROWS
#define HASHJOIN_GENERIC_ROWS	64		/* rows per thread and tile */
#endif
#define HASHJOIN_SLICE_ROWS		16		/* rows per thread and emit slice */
#define HASHJOIN_NSLICES		(HASHJOIN_GENERIC_ROWS / HASHJOIN_SLICE_ROWS)
#define HASHJOIN_EMIT_STAGE_BYTES	32768
static_assert(HASHJOIN_GENERIC_ROWS <= 64 && HASHJOIN_GENERIC_ROWS % HASHJOIN_SLICE_ROWS == 0,
			  "a tile is at most 64 rows per thread (emit bit mask), in whole slices");

template <bool IS_COLUMN, bool ALL_SINGLE>
__device__ __forceinline__ void
gpuhashjoin_main_body(kern_hashjoin *__restrict__ khashjoin,
				 const kern_multihash *__restrict__ kmhash,
				 const hashjoin_index *__restrict__ hjidx,
				 const kern_data_store *__restrict__ kds,
				 const kern_data_store *__restrict__ ktoast,
				 const kern_row_map *__restrict__ krowmap,
				 cl_uint *wave_totals, cl_uint &tile_base_slot, cl_int *emit_stage,
				 cl_int *__restrict__ first_buf)
{
	const kern_parambuf *kparams = KERN_HASHJOIN_PARAMBUF(khashjoin);
	kern_resultbuf *kresults = KERN_HASHJOIN_RESULTBUF(khashjoin);
	bool		use_map = (krowmap != NULL && krowmap->nvalids >= 0);
	cl_uint		nrows = (use_map ? (cl_uint)krowmap->nvalids : kds->nitems);
	/*
	 * rows per thread and tile: up to HASHJOIN_GENERIC_ROWS, but no more than it takes to give every
	 * work-group of the launch a tile -- a thread walks its rows one after the other through chains
	 * of dependent loads (row -> slot -> entry -> tuple), so a small chunk cut into few large tiles
	 * is a few work-groups waiting out memory latency while the rest of the chip is idle (a 325 k-row
	 * chunk is 20 tiles of 16384 rows; a 4e6-row text-key join ran at 1.4 ms in 244 tiles).  The host
	 * launches min(rows / 256, what the chip holds) work-groups; this is the inverse.
	 */
	cl_uint		rows_this = (cl_uint)(((cl_ulong)nrows + (cl_ulong)gridDim.x * HASHJOIN_BLOCK - 1) /
									  ((cl_ulong)gridDim.x * HASHJOIN_BLOCK));
	rows_this = (rows_this < 1 ? 1 : rows_this > HASHJOIN_GENERIC_ROWS ? HASHJOIN_GENERIC_ROWS : rows_this);
	cl_uint		tile_rows = HASHJOIN_BLOCK * rows_this;
	cl_uint		ntiles = (nrows + tile_rows - 1) / tile_rows;
	cl_uint		nrels = kresults->nrels;
	cl_int		chunk_error = StromError_Success;
	cl_int		param_error = StromError_Success;
	strom_kparams KP;

	hashjoin_load_kparams(KP, kparams, &param_error);
	/* COLUMN chunk: column pointers hoisted, no header reads per row */
	const bool	is_column = IS_COLUMN;	/* compile-time: the other accessor is not even compiled in */
	const cl_int chunk_format = kds->format;
	const bool	row_family = (chunk_format == KDS_FORMAT_ROW || chunk_format == KDS_FORMAT_ROW_FLAT);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
#define X(attno,colidx,NAME)													\
	const char *col_##attno = (is_column ? (const char *)kds + coldir[colidx].values_off : NULL);	\
	const cl_uint *nul_##attno = ((is_column && coldir[colidx].nulls_off != 0)	\
		? (const cl_uint *)((const char *)kds + coldir[colidx].nulls_off) : NULL);
	STROM_KVAR_LIST(X)
#undef X
	if (nrels != kmhash->ntables + 1 || nrels != HASHJOIN_NRELS + 1)
	{
		/* uniform: every thread leaves (opencl_hashjoin.h:305-309) */
		if (threadIdx.x == 0)
			atomicMax(&kresults->errcode, StromError_DataStoreCorruption);
		return;
	}
	/*
	 * A tile is HASHJOIN_GENERIC_ROWS rows per thread in slices of
	 * HASHJOIN_SLICE_ROWS.  Counting covers the whole tile and ends in ONE
	 * reservation on kresults->nitems: a returning atomic on one address is
	 * served at ~27 M/s chip-wide (48.8 k of them -- 2048-row tiles -- cost
	 * 1.8 ms per 1e8 rows before any probe).  Emission goes slice by slice
	 * through an LDS stage and leaves with coalesced stores: with large
	 * tiles a thread's own output range is hundreds of bytes from its
	 * neighbour's, and writing it directly costs more than the atomics did
	 * (profiles/r01_hashjoin_ablation.txt).
	 */
	cl_uint		stage_cap = (HASHJOIN_EMIT_STAGE_BYTES / (cl_uint)sizeof(cl_int)) / nrels;	/* records */

	/* chunk_error is per thread: the tile loop must not end divergently */
	for (cl_uint tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
	{
		cl_ulong	emit_mask = 0;		/* bit j: row j of this thread has matches to emit */
		/*
		 * bit j: ... exactly ONE, and the count pass left it in first_buf (HASHJOIN_NRELS offsets
		 * per row position): the emit pass copies it instead of probing again -- the probe of a
		 * KEYED / HASH index is a chain of dependent loads into a table far larger than any cache
		 * (C3 through the HASH index: 7.8 ms per 1e8 rows with both passes probing, the second one
		 * for four rows in five), 8 bytes of streamed scratch per row are not.
		 */
		cl_ulong	single_mask = 0;
		cl_uint		cnt[HASHJOIN_NSLICES];
		cl_uint		off[HASHJOIN_NSLICES];
		cl_uint		tot[HASHJOIN_NSLICES];
		cl_uint		total = 0;

		/* pass 1: count */
#pragma unroll
		for (int s = 0; s < HASHJOIN_NSLICES; s++)
		{
			cnt[s] = 0;
#pragma unroll 1
			for (int jj = 0; jj < HASHJOIN_SLICE_ROWS; jj++)
			{
				int		j = s * HASHJOIN_SLICE_ROWS + jj;
				if ((cl_uint)j >= rows_this)
					break;					/* (uniform) */
				cl_uint	r = tile * tile_rows + j * HASHJOIN_BLOCK + threadIdx.x;
				if (r < nrows)
				{
					cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : r);
					cl_int		errcode = param_error;
					cl_uint		n;
					strom_kvars	KV;
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
#if (defined(HASHJOIN_ABLATE) && HASHJOIN_ABLATE != 0) && !defined(STROM_DIAGNOSTIC_BUILD)
#error "HASHJOIN_ABLATE builds leave work out and give wrong results: measurement only (set STROM_DIAGNOSTIC_BUILD=1, as scripts/gpu_*_ablate* do)"
#endif
#if defined(HASHJOIN_ABLATE) && HASHJOIN_ABLATE == 2
					n = 1;				/* diagnostic build (wrong results): no probe */
#else
					cl_int		fm[HASHJOIN_NRELS];
					n = gpuhashjoin_execute<ALL_SINGLE>(&errcode, KP, KV, kmhash, hjidx, kds_index, NULL,
														first_buf ? fm : NULL);
					if (first_buf && n == 1 && errcode == StromError_Success)
					{
#pragma unroll
						for (int d = 0; d < HASHJOIN_NRELS; d++)
							first_buf[(size_t)r * HASHJOIN_NRELS + d] = fm[d];
						single_mask |= (1UL << j);
					}
#endif
					if (errcode != StromError_Success)
					{
						/* the reference cannot re-check a join row on the CPU
						 * (gpuhashjoin.c:2948-2952): surface it as the chunk status */
						STROM_SET_ERROR(&chunk_error, errcode);
						n = 0;
					}
					if (n > 0)
						emit_mask |= (1UL << j);
					cnt[s] += n;
				}
			}
		}
		/* per slice: this thread's offset inside the slice, the slice total */
#pragma unroll
		for (int s = 0; s < HASHJOIN_NSLICES; s++)
		{
			off[s] = hashjoin_block_scan(cnt[s], wave_totals, &tot[s]);
			total += tot[s];
		}
		if (threadIdx.x == 0)
			tile_base_slot = (total > 0 ? atomicAdd(&kresults->nitems, total) : 0);
		__syncthreads();
		cl_uint		base = tile_base_slot;
		__syncthreads();
		if ((cl_ulong)base + total > (cl_ulong)kresults->nrooms)
		{
			/* keep counting so that nitems ends up as the room required */
			if (threadIdx.x == 0)
				atomicMax(&kresults->errcode, StromError_DataStoreNoSpace);
			continue;
		}
#if defined(HASHJOIN_ABLATE) && HASHJOIN_ABLATE >= 1
		continue;					/* diagnostic build (wrong results): count only */
#endif
		/* pass 2: emit (the probe is repeated; what it finds is what was counted) */
#pragma unroll
		for (int s = 0; s < HASHJOIN_NSLICES; s++)
		{
			bool		staged = (tot[s] <= stage_cap);		/* uniform */
			cl_int	   *out = (staged ? emit_stage + (size_t)nrels * off[s]
								   : kresults->results + (size_t)nrels * (base + off[s]));
			if (tot[s] == 0)
				continue;
#pragma unroll 1
			for (int jj = 0; jj < HASHJOIN_SLICE_ROWS; jj++)
			{
				int		j = s * HASHJOIN_SLICE_ROWS + jj;
				if ((single_mask >> j) & 1)
				{
					/* the count pass's one match, from the scratch */
					cl_uint		r = tile * tile_rows + j * HASHJOIN_BLOCK + threadIdx.x;
					cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : r);
					out[0] = (cl_int)(kds_index + 1);
#pragma unroll
					for (int d = 0; d < HASHJOIN_NRELS; d++)
						out[1 + d] = first_buf[(size_t)r * HASHJOIN_NRELS + d];
					out += nrels;
				}
				else if ((emit_mask >> j) & 1)
				{
					cl_uint		r = tile * tile_rows + j * HASHJOIN_BLOCK + threadIdx.x;
					cl_uint		kds_index = (use_map ? (cl_uint)krowmap->rindex[r] : r);
					cl_int		errcode = param_error;
					strom_kvars	KV;
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
					out += (size_t)nrels *
						gpuhashjoin_execute<ALL_SINGLE>(&errcode, KP, KV, kmhash, hjidx, kds_index, out);
				}
			}
			if (staged)
			{
				/* stage -> results[], coalesced */
				__syncthreads();
				cl_int	   *dest = kresults->results + (size_t)nrels * base;
				cl_uint		nwords = nrels * tot[s];
				for (cl_uint i = threadIdx.x; i < nwords; i += HASHJOIN_BLOCK)
					__builtin_nontemporal_store(emit_stage[i], &dest[i]);
				__syncthreads();
			}
			base += tot[s];
		}
	}
	/* significant errors and CpuReCheck alike end up in kresults->errcode */
	{
		cl_int worst = strom_wave_max_i32(chunk_error);
		if (strom_lane_id() == 0 && worst != StromError_Success)
			atomicMax(&kresults->errcode, worst);
	}
}

extern "C" __global__ void
__launch_bounds__(HASHJOIN_BLOCK)
gpuhashjoin_main(kern_hashjoin *__restrict__ khashjoin,
				 const kern_multihash *__restrict__ kmhash,
				 const hashjoin_index *__restrict__ hjidx,
				 const kern_data_store *__restrict__ kds,
				 const kern_data_store *__restrict__ ktoast,
				 const kern_row_map *__restrict__ krowmap,
				 cl_int *__restrict__ first_buf)		/* rows x HASHJOIN_NRELS ints of scratch, or NULL */
{
	__shared__ cl_uint	wave_totals[HASHJOIN_NWAVES];
	__shared__ cl_uint	tile_base_slot;
	__shared__ cl_int	emit_stage[HASHJOIN_EMIT_STAGE_BYTES / sizeof(cl_int)];

	/*
	 * Uniform decisions made once per launch, not per datum / per probe: the
	 * chunk format, and whether every relation is a DIRECT index with unique
	 * keys.  Left as run-time values inside the row loop they cost 4.4 ms
	 * per 1e8 rows against 0.95 ms (profiles/r01_hashjoin_ablation.txt): the
	 * hash-entry loads behind them are issued whether they are needed or not.
	 */
	bool	all_single = true;
	for (cl_uint d = 0; d < hjidx->nrels; d++)
		all_single = all_single && (hjidx->rel[d].mode != HASHJOIN_MODE_HASH && hjidx->rel[d].unique != 0);
	if (kds->format == KDS_FORMAT_COLUMN)
	{
		if (all_single)
			gpuhashjoin_main_body<true, true>(khashjoin, kmhash, hjidx, kds, ktoast, krowmap, wave_totals, tile_base_slot, emit_stage, first_buf);
		else
			gpuhashjoin_main_body<true, false>(khashjoin, kmhash, hjidx, kds, ktoast, krowmap, wave_totals, tile_base_slot, emit_stage, first_buf);
	}
	else
	{
		/* row formats are bound by the tuple walk: one instantiation (every
		 * copy of the generated probe costs JIT time at query start) */
		gpuhashjoin_main_body<false, false>(khashjoin, kmhash, hjidx, kds, ktoast, krowmap, wave_totals, tile_base_slot, emit_stage, first_buf);
	}
}

/* ====================================================================== *
 * fast probe: DIRECT index, unique keys, COLUMN outer, no row map
 * ====================================================================== */
/* quads per thread and tile of the LDS-slots variant: with the slot array in
 * LDS only ONE 256-thread work-group fits a CU, so a thread keeps 32 rows of
 * column data in flight instead of 8 to cover the HBM latency */
#ifndef HASHJOIN_LDS_QUADS
#define HASHJOIN_LDS_QUADS		8
#endif
#define HASHJOIN_MAX_QUADS		(HASHJOIN_LDS_QUADS > HASHJOIN_QUADS ? HASHJOIN_LDS_QUADS : HASHJOIN_QUADS)
/* matches are appended to the stage in groups of quads that fit it */
#define HASHJOIN_STAGE_ENTRIES	HASHJOIN_STAGE
#define HASHJOIN_APPEND_QUADS(Q)	((Q) * HASHJOIN_BLOCK * 4 <= HASHJOIN_STAGE ? (Q)		\
									 : HASHJOIN_STAGE / (HASHJOIN_BLOCK * 4))

struct hashjoin_stage {
	cl_int		entries[HASHJOIN_STAGE_ENTRIES][2];
	cl_uint		wave_total[HASHJOIN_MAX_QUADS][HASHJOIN_NWAVES];
	cl_uint		flush_base;
};

/* slots[] (entry offsets, 4 bytes) -> slots3[] (offset >> 3, 3 bytes); four slots per thread:
 * twelve bytes, three aligned words */
extern "C" __global__ void
__launch_bounds__(256)
hashjoin_narrow_slots_kernel(hashjoin_index *hjidx, cl_int depth)
{
	const hashjoin_index_rel *ir = &hjidx->rel[depth - 1];
	const cl_uint *slots = (const cl_uint *)((const char *)hjidx + ir->slots_off);
	cl_uint	   *out = (cl_uint *)((char *)hjidx + ir->slots3_off);
	cl_uint		n = ir->nslots;
	cl_uint		nquads = (n + 3) / 4;

	for (cl_uint q = blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += gridDim.x * blockDim.x)
	{
		cl_uint		v[4];
#pragma unroll
		for (int j = 0; j < 4; j++)
			v[j] = (4 * q + j < n ? slots[4 * q + j] >> 3 : 0u);
		out[3 * q + 0] = v[0] | (v[1] << 24);
		out[3 * q + 1] = (v[1] >> 8) | (v[2] << 16);
		out[3 * q + 2] = (v[2] >> 16) | (v[3] << 8);
	}
}

/* one probe of the 3-byte slot array: ONE (unaligned) 4-byte load */
struct __attribute__((packed)) hashjoin_unaligned_u32 { cl_uint v; };
STROM_DEVICE cl_uint
hashjoin_slot3(const cl_uchar *slots3, cl_ulong idx)
{
	return (((const __attribute__((address_space(1))) hashjoin_unaligned_u32 *)(slots3 + 3 * idx))->v & 0xffffffu) << 3;
}

template <int QUADS>
struct hashjoin_column_tile {
#define X(attno,colidx,NAME)											\
	pg_##NAME##_base_t	v_##attno[QUADS][4];							\
	cl_uint				nn_##attno[QUADS];
	STROM_KVAR_LIST(X)
#undef X
	int __dummy;
};

/*
 * LDS_SLOTS: "inner hash staged in LDS" (BASELINE configs[2]; the reference
 * stages its CRC table there, opencl_hashjoin.h:284-416).  A DIRECT index of
 * up to ~30 k key values -- date, nation, category dimensions -- is copied
 * into the work-group's LDS once and every probe is a ds_read: no L2 request
 * at all, the kernel streams at the rate of its column reads and result
 * writes.  The L2-resident form below it is bound by the L2 request rate
 * (~2e11 random 4-byte reads per second chip-wide) whatever the table size.
 */
/* LDS_SLOTS: 0 = DIRECT slots read through the caches, 1 = DIRECT slots staged in LDS,
 * 2 = KEYED index (sparse integer keys): the probe is the 16-byte slot search of
 * hashjoin_first(), one pass instead of the general kernel's count + emit,
 * 3 = DIRECT slots in their 3-byte form (hashjoin_index_rel.slots3_off) */
template <int LDS_SLOTS, int QUADS>
__device__ __forceinline__ void
gpuhashjoin_main_fast_body(kern_hashjoin *khashjoin,
						   const hashjoin_index *hjidx,
						   const kern_data_store *kds)
{
	__shared__ hashjoin_stage stage;
	extern __shared__ __attribute__((aligned(16))) cl_uint lds_slots[];
	const cl_uint TILE_ROWS = HASHJOIN_BLOCK * 4 * QUADS;
	const kern_parambuf *kparams = KERN_HASHJOIN_PARAMBUF(khashjoin);
	kern_resultbuf *kresults = KERN_HASHJOIN_RESULTBUF(khashjoin);
	const kern_coldir *coldir = KERN_DATA_STORE_COLDIR(kds);
	const hashjoin_index_rel *ir = &hjidx->rel[0];
	const cl_uint *slots = (const cl_uint *)((const char *)hjidx + ir->slots_off);
	const cl_uchar *slots3 = (const cl_uchar *)hjidx + ir->slots3_off;
	cl_long		key_min = ir->key_min;
	cl_uint		key_range = ir->nslots;
	cl_uint		nitems = kds->nitems;
	cl_uint		nrooms = kresults->nrooms;
	cl_uint		ntiles = (nitems + TILE_ROWS - 1) / TILE_ROWS;
	cl_uint		lane = threadIdx.x & (STROM_WAVE - 1);
	cl_uint		wave = threadIdx.x / STROM_WAVE;
	cl_int		chunk_error = StromError_Success;
	cl_int		param_error = StromError_Success;
	cl_uint		fill = 0;
	strom_kparams KP;

	hashjoin_load_kparams(KP, kparams, &param_error);
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
	if (LDS_SLOTS == 1)
	{
		/* the whole slot array into LDS, 16 bytes per thread and turn (the
		 * host launches this variant only when it fits: gpuhashjoin.cpp) */
		typedef cl_uint v4_t __attribute__((ext_vector_type(4)));
		cl_uint		nvec = (key_range + 3) / 4;		/* the array is padded to 256 bytes */
		for (cl_uint i = threadIdx.x; i < nvec; i += HASHJOIN_BLOCK)
			((v4_t *)lds_slots)[i] = ((const v4_t *)slots)[i];
		__syncthreads();
	}

	for (cl_uint tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
	{
		cl_uint		tile_base = tile * TILE_ROWS;
		bool		full_tile = (tile_base + TILE_ROWS <= nitems);
		hashjoin_column_tile<QUADS> T;
		cl_uint		match[QUADS][4];
		cl_uint		my_prefix[QUADS];
#if HASHJOIN_FAST_OUTER_QUAL
		cl_int		qual_error[QUADS][4];
#endif

		if (full_tile && !any_nulls)
		{
#pragma unroll
			for (int k = 0; k < QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * HASHJOIN_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * HASHJOIN_BLOCK + threadIdx.x) * 4;
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
			for (int k = 0; k < QUADS; k++)
			{
				cl_uint	row0 = tile_base + (k * HASHJOIN_BLOCK + threadIdx.x) * 4;
#define X(attno,colidx,NAME)													\
				strom_column_load_quad<pg_##NAME##_base_t, false>(col_##attno, nul_##attno,	\
														   row0, nitems,				\
														   T.v_##attno[k], T.nn_##attno[k]);
				STROM_KVAR_LIST(X)
#undef X
			}
		}
		/* every slot read of the tile is issued before the first is used */
#pragma unroll
		for (int k = 0; k < QUADS; k++)
		{
			cl_uint	row0 = tile_base + (k * HASHJOIN_BLOCK + threadIdx.x) * 4;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				strom_kvars	KV;
				cl_int		errcode = param_error;
				cl_long		key;
#define X(attno,colidx,NAME)													\
				KV.KVAR_##attno = pg_##NAME##_make(T.v_##attno[k][j],				\
												   !((T.nn_##attno[k] >> j) & 1));
				STROM_KVAR_LIST(X)
#undef X
				KV.__dummy = 0;
				strom_kvars_from_column(KV, kds, &errcode);
				match[k][j] = 0;
#if HASHJOIN_FAST_OUTER_QUAL
				/*
				 * a qual over outer columns only (a scan's WHERE pulled up into
				 * the join, gpuhashjoin.c:2047-2050).  The general kernel
				 * evaluates it for rows that found their entry, so its errors
				 * count for those rows only; here it is evaluated FIRST, on
				 * the registers the tile already holds, and a row it rejects
				 * without an error is not probed at all -- the kernel is bound
				 * by the L2 request rate of the slot reads.  An error is kept
				 * and raised below if the row turns out to have a partner.
				 */
				cl_int		qerr = StromError_Success;
				bool		qpass = hashjoin_fast_outer_qual(&qerr, KP, KV);
				qual_error[k][j] = qerr;
				if (qpass || qerr != StromError_Success)
#endif
				if (row0 + j < nitems &&
					hashjoin_fast_outer_key(&errcode, KP, KV, &key))
				{
					if (LDS_SLOTS == 2)
					{
						cl_ulong	image = hashjoin_key_image(key);
						cl_uint		h;
						match[k][j] = hashjoin_first(hjidx, 0, &image, 1, &h);
					}
					else
					{
						cl_ulong idx = (cl_ulong)(key - key_min);
						if (idx < key_range)
							match[k][j] = (LDS_SLOTS == 1 ? lds_slots[idx] : LDS_SLOTS == 3 ? hashjoin_slot3(slots3, idx) : slots[idx]);
					}
				}
				if (errcode != StromError_Success)
				{
					STROM_SET_ERROR(&chunk_error, errcode);
					match[k][j] = 0;
				}
			}
		}
#if HASHJOIN_FAST_OUTER_QUAL
#pragma unroll
		for (int k = 0; k < QUADS; k++)
		{
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if (match[k][j] != 0 && qual_error[k][j] != StromError_Success)
				{
					STROM_SET_ERROR(&chunk_error, qual_error[k][j]);
					match[k][j] = 0;
				}
			}
		}
#endif
		const int	GQ = HASHJOIN_APPEND_QUADS(QUADS);
		static_assert(HASHJOIN_APPEND_QUADS(QUADS) >= 1 && QUADS % HASHJOIN_APPEND_QUADS(QUADS) == 0,
					  "the LDS stage takes whole groups of quads");
#pragma unroll
		for (int g0 = 0; g0 < QUADS; g0 += GQ)
		{
		if (fill + GQ * HASHJOIN_BLOCK * 4 > HASHJOIN_STAGE_ENTRIES)
		{
			/* flush: one reservation, contiguous store */
			if (fill > 0)
			{
				if (threadIdx.x == 0)
					stage.flush_base = atomicAdd(&kresults->nitems, fill);
				__syncthreads();
				cl_uint	base = stage.flush_base;
				if ((cl_ulong)base + fill <= nrooms)
				{
					cl_long *dest = (cl_long *)(kresults->results + 2 * (size_t)base);
					for (cl_uint i = threadIdx.x; i < fill; i += HASHJOIN_BLOCK)
						{
							/* non-temporal: 640 MB of result pairs must not evict
							 * the slot array the probes hit in L2 */
							cl_long	pair;
							__builtin_memcpy(&pair, stage.entries[i], 8);
							__builtin_nontemporal_store(pair, &dest[i]);
						}
				}
				else
					STROM_SET_ERROR(&chunk_error, StromError_DataStoreNoSpace);
				__syncthreads();
			}
			fill = 0;
		}
#pragma unroll
		for (int k = g0; k < g0 + GQ; k++)
		{
			cl_uint	prefix = 0, total = 0;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				strom_lanemask_t m = __ballot(match[k][j] != 0);
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
		for (int k = g0; k < g0 + GQ; k++)
		{
			cl_uint	before = 0, all = 0;
#pragma unroll
			for (int w = 0; w < HASHJOIN_NWAVES; w++)
			{
				cl_uint	t = stage.wave_total[k][w];
				before += (w < (int)wave ? t : 0);
				all += t;
			}
			cl_uint	pos = fill + appended + before + my_prefix[k];
			cl_uint	row0 = tile_base + (k * HASHJOIN_BLOCK + threadIdx.x) * 4;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				if (match[k][j] != 0)
				{
					stage.entries[pos][0] = (cl_int)(row0 + j + 1);
					stage.entries[pos][1] = (cl_int)match[k][j];
					pos++;
				}
			}
			appended += all;
		}
		__syncthreads();
		fill += appended;
		}
	}
	if (fill > 0)
	{
		if (threadIdx.x == 0)
			stage.flush_base = atomicAdd(&kresults->nitems, fill);
		__syncthreads();
		cl_uint	base = stage.flush_base;
		if ((cl_ulong)base + fill <= nrooms)
		{
			cl_long *dest = (cl_long *)(kresults->results + 2 * (size_t)base);
			for (cl_uint i = threadIdx.x; i < fill; i += HASHJOIN_BLOCK)
				{
							/* non-temporal: 640 MB of result pairs must not evict
							 * the slot array the probes hit in L2 */
							cl_long	pair;
							__builtin_memcpy(&pair, stage.entries[i], 8);
							__builtin_nontemporal_store(pair, &dest[i]);
						}
		}
		else
			STROM_SET_ERROR(&chunk_error, StromError_DataStoreNoSpace);
	}
	{
		cl_int worst = strom_wave_max_i32(chunk_error);
		if (strom_lane_id() == 0 && worst != StromError_Success)
			atomicMax(&kresults->errcode, worst);
	}
}

extern "C" __global__ void
__launch_bounds__(HASHJOIN_BLOCK)
gpuhashjoin_main_fast(kern_hashjoin *khashjoin,
					  const hashjoin_index *hjidx,
					  const kern_data_store *kds)
{
	gpuhashjoin_main_fast_body<0, HASHJOIN_FAST_QUADS>(khashjoin, hjidx, kds);
}

extern "C" __global__ void
__launch_bounds__(HASHJOIN_BLOCK)
gpuhashjoin_main_fast_lds(kern_hashjoin *khashjoin,
						  const hashjoin_index *hjidx,
						  const kern_data_store *kds)
{
	gpuhashjoin_main_fast_body<1, HASHJOIN_LDS_QUADS>(khashjoin, hjidx, kds);
}

extern "C" __global__ void
__launch_bounds__(HASHJOIN_BLOCK)
gpuhashjoin_main_fast_narrow(kern_hashjoin *khashjoin,
							 const hashjoin_index *hjidx,
							 const kern_data_store *kds)
{
	gpuhashjoin_main_fast_body<3, HASHJOIN_FAST_QUADS>(khashjoin, hjidx, kds);
}

extern "C" __global__ void
__launch_bounds__(HASHJOIN_BLOCK)
gpuhashjoin_main_fast_keyed(kern_hashjoin *khashjoin,
							const hashjoin_index *hjidx,
							const kern_data_store *kds)
{
	gpuhashjoin_main_fast_body<2, HASHJOIN_QUADS>(khashjoin, hjidx, kds);
}

/* ====================================================================== *
 * projection into a TUPSLOT kern_data_store
 * (kern_gpuhashjoin_projection_slot, opencl_hashjoin.h:691-839).  One
 * thread per result record; destination column r takes column
 * src_colidx[r] of relation src_depth[r] (0 = the outer chunk, d = the d-th
 * inner relation) -- the mapping the reference bakes into the generated
 * gpuhashjoin_projection_mapping / _datum (gpuhashjoin.c:1026-1181) arrives
 * here as two small arrays.  Fixed-width by-value columns only.
 * ====================================================================== */
extern "C" __global__ void
__launch_bounds__(256)
gpuhashjoin_projection_slot(kern_hashjoin *khashjoin,
							const kern_multihash *kmhash,
							const kern_data_store *kds,
							const kern_data_store *ktoast,
							kern_data_store *kds_dest,
							const cl_int *src_depth,
							const cl_int *src_colidx)
{
	kern_resultbuf *kresults = KERN_HASHJOIN_RESULTBUF(khashjoin);
	cl_uint		nrels = kresults->nrels;
	cl_uint		nitems = kresults->nitems;
	cl_uint		ncols = kds_dest->ncols;

	/* overflow: the main kernel already said DataStoreNoSpace, nothing to do */
	if (kresults->errcode != StromError_Success ||
		nitems > kresults->nrooms || nitems > kds_dest->nrooms)
	{
		if (blockIdx.x == 0 && threadIdx.x == 0 && nitems > kds_dest->nrooms)
			atomicMax(&kresults->errcode, StromError_DataStoreNoSpace);
		return;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
		kds_dest->nitems = nitems;
	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x;
		 i < nitems;
		 i += gridDim.x * blockDim.x)
	{
		const cl_int *rbuffer = kresults->results + (size_t)nrels * i;
		Datum	   *values = KERN_DATA_STORE_VALUES(kds_dest, i);
		cl_char	   *isnull = KERN_DATA_STORE_ISNULL(kds_dest, i);

		for (cl_uint r = 0; r < ncols; r++)
		{
			cl_int		depth = src_depth[r];
			cl_int		col = src_colidx[r];
			const void *addr = NULL;
			cl_int		attlen = 0;

			if (depth == 0)
			{
				addr = kern_get_datum(kds, ktoast, col, (cl_uint)(rbuffer[0] - 1));
				attlen = (col < (cl_int)kds->ncols ? kds->colmeta[col].attlen : 0);
			}
			else if (depth > 0 && depth < (cl_int)nrels)
			{
				const kern_hashtable *kht = KERN_HASHTABLE(kmhash, depth - 1);
				const kern_hashentry *ent = (const kern_hashentry *)((const char *)kht + rbuffer[depth]);
				addr = kern_get_datum_tuple(kht->colmeta, &ent->htup, col);
				attlen = (col < (cl_int)kht->ncols ? kht->colmeta[col].attlen : 0);
			}
			Datum	d = 0;
			if (addr && attlen > 0 && attlen <= 8)
				__builtin_memcpy(&d, addr, attlen);
			values[r] = d;
			isnull[r] = (addr == NULL || attlen <= 0);
		}
	}
}

/* ====================================================================== *
 * projection into a ROW_FLAT kern_data_store: joined rows as heap tuples
 * (kern_gpuhashjoin_projection_row, opencl_hashjoin.h:437-689), what the
 * reference hands to a parent node that reads tuples (varlena columns
 * included: a text datum is copied verbatim, short header or not).
 *
 * Per record: (1) the tuple's length from the located source datums --
 * att_align_datum / att_addlength_datum, NULL bitmap only if a column is
 * NULL, header and data MAXALIGNed; (2) room: lengths are prefix-summed over
 * the wave by shuffles, wave totals over the work-group through LDS, ONE
 * atomic on kds_dest->usage per work-group and tile (the reference's
 * arithmetic_stairlike_add + atomic, 543-552); tuples grow from the tail of
 * the buffer, row items from the head; (3) the tuple is built in place.
 * A tile that does not fit raises DataStoreNoSpace (the host retries with a
 * larger store); nothing is torn: nitems is the record count either way.
 * ====================================================================== */
struct hashjoin_proj_src {
	const char *addr;
	cl_int		attlen;
};

STROM_DEVICE hashjoin_proj_src
hashjoin_projection_locate(const kern_multihash *kmhash, const kern_data_store *kds,
						   const kern_data_store *ktoast, const cl_int *rbuffer, cl_uint nrels,
						   cl_int depth, cl_int col)
{
	hashjoin_proj_src s;
	s.addr = NULL;
	s.attlen = 0;
	if (depth == 0)
	{
		if (col >= 0 && col < (cl_int)kds->ncols)
		{
			s.addr = (const char *)kern_get_datum(kds, ktoast, col, (cl_uint)(rbuffer[0] - 1));
			s.attlen = kds->colmeta[col].attlen;
		}
	}
	else if (depth > 0 && depth < (cl_int)nrels)
	{
		const kern_hashtable *kht = KERN_HASHTABLE(kmhash, depth - 1);
		const kern_hashentry *ent = (const kern_hashentry *)((const char *)kht + rbuffer[depth]);
		if (col >= 0 && col < (cl_int)kht->ncols)
		{
			s.addr = kern_get_datum_tuple(kht->colmeta, &ent->htup, col);
			s.attlen = kht->colmeta[col].attlen;
		}
	}
	return s;
}

extern "C" __global__ void
__launch_bounds__(256)
gpuhashjoin_projection_row(kern_hashjoin *khashjoin,
						   const kern_multihash *kmhash,
						   const kern_data_store *kds,
						   const kern_data_store *ktoast,
						   kern_data_store *kds_dest,
						   const cl_int *src_depth,
						   const cl_int *src_colidx)
{
	__shared__ cl_uint	wave_total[256 / STROM_WAVE];
	__shared__ cl_uint	tile_base;
	__shared__ cl_int	tile_error;
	kern_resultbuf *kresults = KERN_HASHJOIN_RESULTBUF(khashjoin);
	const cl_uint	nrels = kresults->nrels;
	const cl_uint	nitems = kresults->nitems;
	const cl_uint	ncols = kds_dest->ncols;
	const cl_uint	dest_length = kds_dest->length;
	const cl_uint	lane = threadIdx.x & (STROM_WAVE - 1);
	const cl_uint	wave = threadIdx.x / STROM_WAVE;

	if (kresults->errcode != StromError_Success ||
		nitems > kresults->nrooms || nitems > kds_dest->nrooms)
	{
		if (blockIdx.x == 0 && threadIdx.x == 0 && nitems > kds_dest->nrooms)
			atomicMax(&kresults->errcode, StromError_DataStoreNoSpace);
		return;
	}
	if (kds_dest->format != KDS_FORMAT_ROW_FLAT)
	{
		if (blockIdx.x == 0 && threadIdx.x == 0)
			atomicMax(&kresults->errcode, StromError_DataStoreCorruption);
		return;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
		kds_dest->nitems = nitems;
	/* row items end here; tuples must stay above */
	const cl_uint	usage_head = (cl_uint)(KDS_HEAD_LENGTH(ncols) +
										   STROMALIGN(sizeof(kern_blkitem) * kds_dest->maxblocks) +
										   STROMALIGN(sizeof(kern_rowitem) * (size_t)nitems));
	const cl_uint	ntiles = (nitems + blockDim.x - 1) / blockDim.x;

	for (cl_uint tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
	{
		cl_uint		i = tile * blockDim.x + threadIdx.x;
		bool		valid = (i < nitems);
		const cl_int *rbuffer = kresults->results + (size_t)nrels * (valid ? i : 0);
		cl_uint		datalen = 0, t_hoff = 0, required = 0;
		bool		hasnull = false, bad = false;

		/* step 1: length of the joined tuple */
		if (valid)
		{
			for (cl_uint r = 0; r < ncols; r++)
			{
				kern_colmeta	cm = kds_dest->colmeta[r];
				hashjoin_proj_src src = hashjoin_projection_locate(kmhash, kds, ktoast, rbuffer, nrels,
																   src_depth[r], src_colidx[r]);
				if (!src.addr)
				{
					hasnull = true;
					continue;
				}
				if ((cm.attlen > 0) != (src.attlen > 0) || (cm.attlen > 0 && cm.attlen != src.attlen))
				{
					bad = true;					/* the destination column is not the source's shape */
					continue;
				}
				if (cm.attlen > 0)
					datalen = STROM_TYPEALIGN(cm.attalign, datalen) + cm.attlen;
				else
				{
					if (!strom_varatt_is_1b(src.addr))
						datalen = STROM_TYPEALIGN(cm.attalign, datalen);
					datalen += strom_varsize_any(src.addr);
				}
			}
			t_hoff = HEAPTUPLE_HEADER_FIXED + (hasnull ? (ncols + 7) / 8 : 0);
			t_hoff = STROM_LONGALIGN(t_hoff);
			required = t_hoff + STROM_LONGALIGN(datalen);
			if (t_hoff > 255)
				bad = true;						/* t_hoff is one byte */
		}
		/* step 2: room -- inclusive prefix over the wave, wave totals through LDS */
		cl_uint		incl = required;
		for (int off = 1; off < STROM_WAVE; off <<= 1)
		{
			cl_uint o = (cl_uint)__shfl_up((int)incl, off, STROM_WAVE);
			if (lane >= (cl_uint)off)
				incl += o;
		}
		if (lane == STROM_WAVE - 1)
			wave_total[wave] = incl;
		if (threadIdx.x == 0)
			tile_error = StromError_Success;
		__syncthreads();
		cl_uint		before = 0, total = 0;
		for (cl_uint w = 0; w < blockDim.x / STROM_WAVE; w++)
		{
			if (w < wave)
				before += wave_total[w];
			total += wave_total[w];
		}
		if (bad)
			atomicMax(&tile_error, StromError_DataStoreCorruption);
		if (threadIdx.x == 0)
		{
			cl_uint	prev = (total > 0 ? atomicAdd(&kds_dest->usage, total) : 0);
			/* (64-bit sum: usage may have run past 4 GB on a hopeless request) */
			if ((cl_ulong)usage_head + (cl_ulong)prev + (cl_ulong)total > (cl_ulong)dest_length)
				atomicMax(&tile_error, StromError_DataStoreNoSpace);
			tile_base = prev;
		}
		__syncthreads();
		cl_int		terr = tile_error;
		cl_uint		base = tile_base;
		__syncthreads();						/* tile_base / wave_total are rewritten by the next tile */
		if (terr != StromError_Success)
		{
			if (threadIdx.x == 0)
				atomicMax(&kresults->errcode, terr);
			continue;
		}
		if (!valid)
			continue;
		/* step 3: the tuple, in place */
		cl_uint		htup_offset = dest_length - (base + before + incl);
		char	   *tup = (char *)kds_dest + htup_offset;
		HeapTupleHeaderData *htup = (HeapTupleHeaderData *)tup;

		KERN_DATA_STORE_ROWITEM(kds_dest, i)->htup_offset = htup_offset;
		for (cl_uint b = 0; b < t_hoff; b++)
			tup[b] = 0;
		htup->t_xmin = required << 2;			/* t_choice.t_datum: SET_VARSIZE, typmod, type id */
		htup->t_xmax = (cl_uint)kds_dest->tdtypmod;
		htup->t_field3 = kds_dest->tdtypeid;
		htup->t_infomask2 = (cl_ushort)(ncols & HEAP_NATTS_MASK);
		htup->t_hoff = (cl_uchar)t_hoff;
		cl_uint		curr = t_hoff;
		bool		hasvarwidth = false;
		for (cl_uint r = 0; r < ncols; r++)
		{
			kern_colmeta	cm = kds_dest->colmeta[r];
			hashjoin_proj_src src = hashjoin_projection_locate(kmhash, kds, ktoast, rbuffer, nrels,
															   src_depth[r], src_colidx[r]);
			if (!src.addr)
				continue;						/* bit stays 0 */
			if (hasnull)
				htup->t_bits[r >> 3] |= (cl_uchar)(1 << (r & 7));
			cl_uint	len;
			if (cm.attlen > 0)
			{
				while (STROM_TYPEALIGN(cm.attalign, curr) != curr)
					tup[curr++] = 0;
				len = (cl_uint)cm.attlen;
			}
			else
			{
				if (!strom_varatt_is_1b(src.addr))
					while (STROM_TYPEALIGN(cm.attalign, curr) != curr)
						tup[curr++] = 0;
				len = strom_varsize_any(src.addr);
				hasvarwidth = true;
			}
			for (cl_uint b = 0; b < len; b++)
				tup[curr + b] = src.addr[b];
			curr += len;
		}
		while (curr < required)
			tup[curr++] = 0;
		htup->t_infomask = (cl_ushort)((hasnull ? HEAP_HASNULL : 0) | (hasvarwidth ? HEAP_HASVARWIDTH : 0));
	}
}

/* ====================================================================== *
 * a dimension column by SLOT (DIRECT index, unique keys): values[slot] /
 * isnull[slot] of inner column 'col'.  Built once per table and column, on
 * the first projection that asks for it: reading a joined row's inner column
 * from the entry is one scattered 64-byte line out of the whole table per
 * (record, column) -- 86 % L2 misses, 3.2 ms per 8e7 records and column --
 * while slot-indexed arrays of a 1e6-row dimension are 4 MB and stay in L2.
 * ====================================================================== */
extern "C" __global__ void
__launch_bounds__(256)
hashjoin_build_dimcol_kernel(const kern_multihash *kmhash, const hashjoin_index *hjidx,
							 cl_int col, cl_int attlen, char *values, cl_uchar *isnull, cl_uint *p_failed)
{
	const kern_hashtable *kht = KERN_HASHTABLE(kmhash, 0);
	const hashjoin_index_rel *ir = &hjidx->rel[0];
	const cl_uint *slots = (const cl_uint *)((const char *)hjidx + ir->slots_off);

	if (col == -1)
	{
		/* not a column: which slots hold no inner row at all */
		for (cl_uint s = blockIdx.x * blockDim.x + threadIdx.x; s < ir->nslots; s += gridDim.x * blockDim.x)
			isnull[s] = (slots[s] == 0);
		return;
	}
	if (col < 0 || col >= (cl_int)kht->ncols || kht->colmeta[col].attlen != attlen)
	{
		if (blockIdx.x == 0 && threadIdx.x == 0)
			*p_failed = 1;
		return;
	}
	for (cl_uint s = blockIdx.x * blockDim.x + threadIdx.x; s < ir->nslots; s += gridDim.x * blockDim.x)
	{
		cl_uint		off = slots[s];
		const char *addr = NULL;
		if (off != 0)
		{
			const kern_hashentry *ent = (const kern_hashentry *)((const char *)kht + off);
			addr = kern_get_datum_tuple(kht->colmeta, &ent->htup, col);
		}
		char	   *out = values + (size_t)attlen * s;
		switch (attlen)
		{
			case 1: *(cl_char *)out = (addr ? *(const cl_char *)addr : 0); break;
			case 2: *(cl_short *)out = (addr ? strom_fetch<cl_short>(addr) : 0); break;
			case 4: *(cl_int *)out = (addr ? strom_fetch<cl_int>(addr) : 0); break;
			default: *(cl_long *)out = (addr ? strom_fetch<cl_long>(addr) : 0); break;
		}
		isnull[s] = (addr == NULL);
	}
}

/* ====================================================================== *
 * the same by slot, but PACKED: one record per slot holding a flags word
 * (bit 0: an inner row is there; bit 1+i: its i-th wanted column is NULL)
 * and the wanted columns' values, for the join-as-a-lookup aggregate
 * (gpupreagg_dense_lookup): with separate arrays every row costs one L2
 * request per array, and that kernel is bound by the L2 request rate.
 * ====================================================================== */
struct hashjoin_dimrec_spec {
	cl_uint		ncols;
	cl_uint		reclen;
	struct {
		cl_int		col;			/* inner column, 0-based */
		cl_int		attlen;
		cl_uint		offset;			/* of the value inside the record */
		cl_uint		__pad;
	} c[16];
};

extern "C" __global__ void
__launch_bounds__(256)
hashjoin_build_dimrec_kernel(const kern_multihash *kmhash, const hashjoin_index *hjidx,
							 const hashjoin_dimrec_spec *spec, char *recs, cl_uint *p_failed)
{
	const kern_hashtable *kht = KERN_HASHTABLE(kmhash, 0);
	const hashjoin_index_rel *ir = &hjidx->rel[0];
	const cl_uint *slots = (const cl_uint *)((const char *)hjidx + ir->slots_off);
	cl_uint		ncols = spec->ncols;
	cl_uint		reclen = spec->reclen;

	for (cl_uint i = 0; i < ncols; i++)
	{
		cl_int	col = spec->c[i].col;
		if (col < 0 || col >= (cl_int)kht->ncols || kht->colmeta[col].attlen != spec->c[i].attlen)
		{
			if (blockIdx.x == 0 && threadIdx.x == 0)
				*p_failed = 1;
			return;
		}
	}
	for (cl_uint s = blockIdx.x * blockDim.x + threadIdx.x; s < ir->nslots; s += gridDim.x * blockDim.x)
	{
		char	   *rec = recs + (size_t)reclen * s;
		cl_uint		off = slots[s];
		cl_uint		flags = (off != 0 ? 1u : 0u);

		for (cl_uint i = 0; i < ncols; i++)
		{
			const char *addr = NULL;
			cl_int		attlen = spec->c[i].attlen;
			char	   *out = rec + spec->c[i].offset;
			if (off != 0)
			{
				const kern_hashentry *ent = (const kern_hashentry *)((const char *)kht + off);
				addr = kern_get_datum_tuple(kht->colmeta, &ent->htup, spec->c[i].col);
			}
			switch (attlen)
			{
				case 1: *(cl_char *)out = (addr ? *(const cl_char *)addr : 0); break;
				case 2: *(cl_short *)out = (addr ? strom_fetch<cl_short>(addr) : 0); break;
				case 4: *(cl_int *)out = (addr ? strom_fetch<cl_int>(addr) : 0); break;
				default: *(cl_long *)out = (addr ? strom_fetch<cl_long>(addr) : 0); break;
			}
			if (addr == NULL)
				flags |= (2u << i);
		}
		*(cl_uint *)rec = flags;
	}
}

/* ====================================================================== *
 * NARROW slot records: the same information in 2 or 4 bytes per slot.
 *
 * A probe into the slot records is one random read; what it costs depends on
 * whether the table stays in an XCD's 4 MB L2 while the fact columns stream
 * past.  8-byte records of a 1e6-row dimension are 10 MB: every fifth probe
 * misses L2 and fetches a 64-byte line from HBM for 8 useful bytes (PMC:
 * 1e7 misses per 1e8 fact rows, more HBM traffic than one of the streamed
 * columns).  Integer columns rarely need their declared width: with the
 * range of each wanted column known (hashjoin_dimrec_minmax_kernel over the
 * standard records), a record is  bit 0 presence | bit 1+i NULL of column i |
 * (value_i - min_i) in just enough bits  -- a GROUP BY column with 1e4
 * distinct values and its two flag bits fit 16 bits, 2.5 MB for 1.25e6 slots.
 * The presence / NULL bits sit where the standard record's flags word has
 * them, so a consumer tests them the same way.
 * ====================================================================== */
struct hashjoin_dimrec_range {
	cl_long		vmin[16];
	cl_long		vmax[16];
	cl_uint		nvalues[16];
};

struct hashjoin_dimrec_narrow_spec {
	cl_uint		ncols;
	cl_uint		reclen;				/* 2 or 4 */
	cl_uint		shift[16];
	cl_uint		mask[16];
	cl_long		vmin[16];
};

STROM_DEVICE cl_long
hashjoin_dimrec_value(const char *rec, cl_uint offset, cl_int attlen)
{
	switch (attlen)
	{
		case 1:  return *(const cl_char *)(rec + offset);
		case 2:  return *(const cl_short *)(rec + offset);
		case 4:  return *(const cl_int *)(rec + offset);
		default: return *(const cl_long *)(rec + offset);
	}
}

extern "C" __global__ void
__launch_bounds__(256)
hashjoin_dimrec_minmax_kernel(const hashjoin_dimrec_spec *spec, const char *recs, cl_uint nslots,
							  hashjoin_dimrec_range *out)
{
	cl_uint		ncols = spec->ncols;
	cl_uint		reclen = spec->reclen;
	cl_long		my_min[16], my_max[16];
	cl_uint		seen = 0;

	for (int i = 0; i < 16; i++)
	{
		my_min[i] = 0x7fffffffffffffffL;
		my_max[i] = -0x7fffffffffffffffL - 1;
	}
	for (cl_uint s = blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += gridDim.x * blockDim.x)
	{
		const char *rec = recs + (size_t)reclen * s;
		cl_uint		flags = *(const cl_uint *)rec;
		if (!(flags & 1u))
			continue;
#pragma unroll
		for (int i = 0; i < 16; i++)
		{
			if (i < (int)ncols && !(flags & (2u << i)))
			{
				cl_long v = hashjoin_dimrec_value(rec, spec->c[i].offset, spec->c[i].attlen);
				my_min[i] = (v < my_min[i] ? v : my_min[i]);
				my_max[i] = (v > my_max[i] ? v : my_max[i]);
				seen |= (1u << i);
			}
		}
	}
	for (int off = STROM_WAVE / 2; off > 0; off >>= 1)
	{
		seen |= (cl_uint)__shfl_xor((int)seen, off, STROM_WAVE);
#pragma unroll
		for (int i = 0; i < 16; i++)
		{
			cl_long	omin = __shfl_xor(my_min[i], off, STROM_WAVE);
			cl_long	omax = __shfl_xor(my_max[i], off, STROM_WAVE);
			my_min[i] = (omin < my_min[i] ? omin : my_min[i]);
			my_max[i] = (omax > my_max[i] ? omax : my_max[i]);
		}
	}
	if ((threadIdx.x & (STROM_WAVE - 1)) == 0)
	{
#pragma unroll
		for (int i = 0; i < 16; i++)
		{
			if (!(seen & (1u << i)))
				continue;
			atomicOr(&out->nvalues[i], 1u);
			atomicMin((long long *)&out->vmin[i], (long long)my_min[i]);
			atomicMax((long long *)&out->vmax[i], (long long)my_max[i]);
		}
	}
}

extern "C" __global__ void
__launch_bounds__(256)
hashjoin_dimrec_narrow_kernel(const hashjoin_dimrec_spec *spec, const char *recs, cl_uint nslots,
							  const hashjoin_dimrec_narrow_spec *nspec, char *out)
{
	cl_uint		ncols = spec->ncols;
	cl_uint		reclen = spec->reclen;
	cl_uint		flagmask = (2u << ncols) - 1u;

	for (cl_uint s = blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += gridDim.x * blockDim.x)
	{
		const char *rec = recs + (size_t)reclen * s;
		cl_uint		flags = *(const cl_uint *)rec;
		cl_uint		w = flags & flagmask;

		if (flags & 1u)
		{
			for (cl_uint i = 0; i < ncols; i++)
			{
				if (flags & (2u << i))
					continue;
				cl_long v = hashjoin_dimrec_value(rec, spec->c[i].offset, spec->c[i].attlen);
				w |= ((cl_uint)(cl_ulong)(v - nspec->vmin[i]) & nspec->mask[i]) << nspec->shift[i];
			}
		}
		if (nspec->reclen == 2)
			((cl_ushort *)out)[s] = (cl_ushort)w;
		else
			((cl_uint *)out)[s] = w;
	}
}

/* ====================================================================== *
 * projection into a COLUMN chunk that stays in HBM
 *
 * What the next operator of a device-resident chain reads (SURVEY.md
 * section 8 f2): the reference materialises joined rows as TUPSLOT and ships
 * them to the host for the next node (gpuhashjoin.c:4883-4968, bulk-slot
 * hand-over gpuhashjoin.c:2686-2689); here they become column arrays, so
 * that GpuPreAgg / GpuScan run their streaming kernels on them.  One
 * thread per result record, same mapping arrays as the TUPSLOT kernel;
 * values are stored at the record's index (consecutive lanes, consecutive
 * addresses), the not-null words come from one ballot per column.  The
 * destination's column widths must equal the sources'.
 * ====================================================================== */
#define HASHJOIN_PROJ_MAXCOLS	64
#define HASHJOIN_PROJ_ROWS		4

extern "C" __global__ void
__launch_bounds__(256)
gpuhashjoin_projection_column(kern_hashjoin *khashjoin,
							  const kern_multihash *kmhash,
							  const kern_data_store *kds,
							  const kern_data_store *ktoast,
							  kern_data_store *dst,
							  const cl_int *src_depth,
							  const cl_int *src_colidx,
							  cl_uint *col_has_null,		/* [ncols] flags, then one failure flag */
							  const hashjoin_index *hjidx,
							  const cl_ulong *dimptr)		/* [2 * ncols] slot-indexed values / isnull arrays, or 0 */
{
	/* the mapping and both column directories, staged once: read per (record,
	 * column) they are dependent scalar loads in front of every gather */
	__shared__ cl_int	s_depth[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_int	s_col[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_int	s_dstlen[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_uint	s_values_off[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_uint	s_nulls_off[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_int	s_srclen[HASHJOIN_PROJ_MAXCOLS];		/* outer COLUMN chunk: the source column */
	__shared__ cl_uint	s_src_values[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_uint	s_src_nulls[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_uint	s_hasnull[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_int	s_cacheoff[HASHJOIN_PROJ_MAXCOLS];		/* inner column: attcacheoff, or -1 */
	__shared__ cl_ulong	s_dimvalues[HASHJOIN_PROJ_MAXCOLS];		/* inner column by slot (hashjoin_build_dimcol_kernel) */
	__shared__ cl_ulong	s_dimisnull[HASHJOIN_PROJ_MAXCOLS];
	__shared__ cl_uint	s_any_dim;
	kern_resultbuf *kresults = KERN_HASHJOIN_RESULTBUF(khashjoin);
	cl_uint		nrels = kresults->nrels;
	cl_uint		nitems = dst->nitems;				/* set by the host from the finished join */
	cl_uint		ncols = dst->ncols;
	bool		outer_is_column = (kds->format == KDS_FORMAT_COLUMN);
	cl_uint		lane = threadIdx.x & 63;

	if (threadIdx.x == 0)
		s_any_dim = 0;
	__syncthreads();
	for (cl_uint r = threadIdx.x; r < HASHJOIN_PROJ_MAXCOLS; r += blockDim.x)
	{
		s_hasnull[r] = 0;
		s_dimvalues[r] = 0;
		s_dimisnull[r] = 0;
		if (r < ncols)
		{
			const kern_coldir *cd = KERN_DATA_STORE_COLDIR(dst) + r;
			if (HASHJOIN_FAST_ELIGIBLE && dimptr != NULL && outer_is_column && dimptr[2 * r] != 0)
			{
				s_dimvalues[r] = dimptr[2 * r];
				s_dimisnull[r] = dimptr[2 * r + 1];
				s_any_dim = 1;
			}
			cl_int	col = src_colidx[r];
			s_depth[r] = src_depth[r];
			s_col[r] = col;
			s_dstlen[r] = dst->colmeta[r].attlen;
			s_values_off[r] = cd->values_off;
			s_nulls_off[r] = cd->nulls_off;
			s_srclen[r] = 0;
			s_src_values[r] = 0;
			s_src_nulls[r] = 0;
			if (src_depth[r] == 0 && outer_is_column && col >= 0 && col < (cl_int)kds->ncols)
			{
				const kern_coldir *sd = KERN_DATA_STORE_COLDIR(kds) + col;
				s_srclen[r] = kds->colmeta[col].attlen;
				s_src_values[r] = sd->values_off;
				s_src_nulls[r] = sd->nulls_off;
			}
			s_cacheoff[r] = -1;
			if (src_depth[r] > 0 && src_depth[r] < (cl_int)nrels)
			{
				const kern_hashtable *kht = KERN_HASHTABLE(kmhash, src_depth[r] - 1);
				if (col >= 0 && col < (cl_int)kht->ncols)
				{
					s_srclen[r] = kht->colmeta[col].attlen;
					s_cacheoff[r] = kht->colmeta[col].attcacheoff;
				}
			}
		}
	}
	__syncthreads();

	/*
	 * HASHJOIN_PROJ_ROWS records per thread and turn: the way to an inner
	 * datum is a chain of dependent loads (result pair -> entry -> tuple
	 * header -> attribute), so a thread first resolves the addresses of all
	 * its records, then fetches, then stores.  Measured, 4e7 records x 3
	 * columns (two from a COLUMN outer chunk, one from 1e6 inner tuples):
	 * 2.9 ms through kern_get_datum, 1.8 ms with the directories staged in
	 * LDS and direct column reads -- and still 1.8 ms with 4 records in
	 * flight per thread: what is left is one scattered 64-byte line per
	 * record and inner column, not the latency of the chain.
	 */
	const kern_hashtable *kht_of[8];
	for (cl_uint d = 1; d < nrels && d <= 8; d++)
		kht_of[d - 1] = KERN_HASHTABLE(kmhash, d - 1);
	strom_kparams KP;
	cl_int		perr = StromError_Success;
	if (HASHJOIN_FAST_ELIGIBLE && s_any_dim)
		hashjoin_load_kparams(KP, KERN_HASHJOIN_PARAMBUF(khashjoin), &perr);
	else
		KP.__dummy = 0;
	for (cl_uint base = blockIdx.x * blockDim.x * HASHJOIN_PROJ_ROWS;
		 base < nitems;
		 base += gridDim.x * blockDim.x * HASHJOIN_PROJ_ROWS)
	{
		cl_uint		idx[HASHJOIN_PROJ_ROWS];
		bool		valid[HASHJOIN_PROJ_ROWS];
		const cl_int *rbuf[HASHJOIN_PROJ_ROWS];
		cl_uint		outer_row[HASHJOIN_PROJ_ROWS];
#pragma unroll
		for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
		{
			idx[k] = base + k * blockDim.x + threadIdx.x;		/* blockDim is a multiple of 64 */
			valid[k] = (idx[k] < nitems);
			rbuf[k] = kresults->results + (size_t)nrels * (valid[k] ? idx[k] : 0);
			outer_row[k] = (cl_uint)(rbuf[k][0] - 1);
		}
		/* the slot of a record's inner row, from the outer key (the generated
		 * hashjoin_fast_outer_key of the one-pass join kernel) */
		cl_uint		slot_idx[HASHJOIN_PROJ_ROWS];
		if (HASHJOIN_FAST_ELIGIBLE && s_any_dim)
		{
			const kern_coldir *ocd = KERN_DATA_STORE_COLDIR(kds);
#pragma unroll
			for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
			{
				strom_kvars	KV;
				cl_int		errcode = perr;
				cl_long		key = 0;
#define X(attno,colidx,NAME)													\
				KV.KVAR_##attno = STROM_COLUMN_REF_CACHED(NAME, (const char *)kds + ocd[colidx].values_off,	\
					(ocd[colidx].nulls_off != 0 ? (const char *)kds + ocd[colidx].nulls_off : NULL),	\
					(valid[k] ? outer_row[k] : 0));
				STROM_KVAR_LIST(X)
#undef X
				KV.__dummy = 0;
				strom_kvars_from_column(KV, kds, &errcode);
				slot_idx[k] = ~0u;
				if (valid[k] && hashjoin_fast_outer_key(&errcode, KP, KV, &key))
				{
					cl_ulong idx = (cl_ulong)(key - hjidx->rel[0].key_min);
					if (idx < hjidx->rel[0].nslots)
						slot_idx[k] = (cl_uint)idx;
				}
			}
		}
		for (cl_uint r = 0; r < ncols; r++)
		{
			cl_int		depth = s_depth[r];
			cl_int		col = s_col[r];
			cl_int		dstlen = s_dstlen[r];
			const void *addr[HASHJOIN_PROJ_ROWS];
			cl_long		val[HASHJOIN_PROJ_ROWS];
			bool		mismatch = false;

			if (depth == 0 && s_src_values[r] != 0)
			{
				/* COLUMN outer chunk: straight from the column array */
				cl_int	attlen = s_srclen[r];
				const cl_uint *nn = (s_src_nulls[r] != 0
									 ? (const cl_uint *)((const char *)kds + s_src_nulls[r]) : NULL);
				mismatch = (attlen != dstlen);
#pragma unroll
				for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
				{
					bool	isnull = (nn != NULL && !((nn[outer_row[k] >> 5] >> (outer_row[k] & 31)) & 1));
					addr[k] = ((valid[k] && !isnull && !mismatch)
							   ? (const char *)kds + s_src_values[r] + (size_t)attlen * outer_row[k] : NULL);
				}
			}
			else if (depth == 0)
			{
				mismatch = ((col < (cl_int)kds->ncols ? kds->colmeta[col].attlen : 0) != dstlen);
#pragma unroll
				for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
					addr[k] = ((valid[k] && !mismatch) ? kern_get_datum(kds, ktoast, col, outer_row[k]) : NULL);
			}
			else if (HASHJOIN_FAST_ELIGIBLE && depth == 1 && s_dimvalues[r] != 0)
			{
				/* the dimension column by slot: a few MB, L2-resident */
				const char	   *dvalues = (const char *)s_dimvalues[r];
				const cl_uchar *disnull = (const cl_uchar *)s_dimisnull[r];
				mismatch = (s_srclen[r] != dstlen);
#pragma unroll
				for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
				{
					addr[k] = NULL;
					if (valid[k] && !mismatch && slot_idx[k] != ~0u && !disnull[slot_idx[k]])
						addr[k] = dvalues + (size_t)dstlen * slot_idx[k];
				}
			}
			else if (depth > 0 && depth < (cl_int)nrels && depth <= 8)
			{
				/*
				 * an inner tuple without NULLs keeps a fixed-width column at
				 * its attcacheoff: ONE load of t_infomask2 / t_infomask (adjacent) decides
				 * that, instead of the accessor's three loads and a colmeta read
				 * per record and column (3.3 ms per 8e7 records and column before)
				 */
				const kern_hashtable *kht = kht_of[depth - 1];
				cl_int	cacheoff = s_cacheoff[r];
				mismatch = (s_srclen[r] != dstlen);
#pragma unroll
				for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
				{
					const kern_hashentry *ent = (const kern_hashentry *)((const char *)kht + rbuf[k][depth]);
					const HeapTupleHeaderData *htup = &ent->htup;
					addr[k] = NULL;
					if (valid[k] && !mismatch)
					{
						cl_uint		hw = strom_fetch<cl_uint>((const char *)htup
															  + offsetof(HeapTupleHeaderData, t_infomask2));
						cl_uint		natts = (hw & 0xffffu) & HEAP_NATTS_MASK;
						cl_uint		infomask = (hw >> 16);
						if (!(infomask & HEAP_HASNULL) && cacheoff >= 0 && (cl_uint)col < natts)
							addr[k] = (const char *)htup + cacheoff;
						else
							addr[k] = kern_get_datum_tuple(kht->colmeta, htup, col);
					}
				}
			}
			else
			{
				mismatch = true;
#pragma unroll
				for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
					addr[k] = NULL;
			}
			if (mismatch && threadIdx.x == 0)
				col_has_null[ncols] = 1;			/* the mapping does not fit the destination */
#pragma unroll
			for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
			{
				val[k] = 0;
				if (addr[k])
				{
					switch (dstlen)
					{
						case 1: val[k] = *(const cl_char *)addr[k]; break;
						case 2: val[k] = strom_fetch<cl_short>(addr[k]); break;
						case 4: val[k] = strom_fetch<cl_int>(addr[k]); break;
						default: val[k] = strom_fetch<cl_long>(addr[k]); break;
					}
				}
			}
#pragma unroll
			for (int k = 0; k < HASHJOIN_PROJ_ROWS; k++)
			{
				if (valid[k])
				{
					char   *out = (char *)dst + s_values_off[r] + (size_t)dstlen * idx[k];
					switch (dstlen)
					{
						case 1: *(cl_char *)out = (cl_char)val[k]; break;
						case 2: *(cl_short *)out = (cl_short)val[k]; break;
						case 4: *(cl_int *)out = (cl_int)val[k]; break;
						default: *(cl_long *)out = val[k]; break;
					}
				}
				strom_lanemask_t nn = __ballot(addr[k] != NULL);
				strom_lanemask_t vv = __ballot(valid[k]);
				if (vv != 0)
				{
					cl_uint *words = (cl_uint *)((char *)dst + s_nulls_off[r]);
					cl_uint	 w0 = (base + k * blockDim.x + (threadIdx.x & ~63u)) >> 5;
					if (lane == 0)
						words[w0] = (cl_uint)nn;
					if (lane == 32 && (vv >> 32) != 0)
						words[w0 + 1] = (cl_uint)(nn >> 32);
					if (nn != vv && lane == 0)
						s_hasnull[r] = 1;
				}
			}
		}
	}
	__syncthreads();
	for (cl_uint r = threadIdx.x; r < ncols; r += blockDim.x)
	{
		if (s_hasnull[r])
			col_has_null[r] = 1;
	}
}

#endif	/* STROM_HASHJOIN_DEVICE_H */
