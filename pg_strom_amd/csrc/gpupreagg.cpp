/*
 * gpupreagg.cpp -- host side of GpuPreAgg
 *
 * Role in the reference: clserv_process_gpupreagg (gpupreagg.c:3849-4240)
 * and clserv_respond_gpupreagg (3009-3194): send the chunk, run
 * preparation -> (set_rindex | bitonic sort) -> reduction, read back the
 * partial rows; plus pgstrom_create_gpupreagg (2329-2416) which sizes the
 * buffers.  Here a request folds its chunk into a per-GPU table that stays
 * in HBM (see strom_hip.h) using strom_gpupreagg.h's dense-id kernels; the
 * geometry decisions -- how many LDS replicas, how many id-range roles,
 * how many work-groups -- are made below from the group domain, which is
 * the job clserv_compute_workgroup_size + the sort-size heuristics
 * (gpupreagg.c:4105-4141) do in the reference.
 */
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <cstdio>
#include <algorithm>

#include "runtime.h"

using namespace strom;

/* a scratch session of a hashed one, its program built with GPUPREAGG_CHECKED (defined below) */
static strom_gpupreagg *gpupreagg_exact_child(strom_gpupreagg *sess, int *p_errcode);

namespace {

/* mirrors struct gpupreagg_dense_ctl of strom_gpupreagg.h */
struct dense_ctl {
	cl_uint		ngroups;
	cl_uint		nsplits;
	cl_uint		groups_per_split;
	cl_uint		nrep;
	cl_uint		nslabs;
	cl_uint		nkeys;
	cl_ulong	slab_bytes;
	cl_long		key_min[STROM_PREAGG_MAXKEYS];
	cl_uint		key_range[STROM_PREAGG_MAXKEYS];
	cl_uint		key_stride[STROM_PREAGG_MAXKEYS];
	cl_ulong	remap;				/* device cl_uint[dense_ngroups], 0 = none */
	cl_uint		dense_ngroups;
	cl_uint		merge_ws;
};

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}	/* namespace */

struct strom_gpupreagg {
	strom_devprog_key	key = 0;
	Program			   *prog = nullptr;
	Device			   *dev = nullptr;
	std::vector<strom_preagg_target> targets;
	std::vector<int>	agg_resno, key_resno;
	std::vector<char>	kparams;			/* kern_parambuf image */
	bool				has_domain = false;
	dense_ctl			ctl;
	size_t				lds_bytes = 0;
	size_t				table_bytes = 0;
	char			   *table = nullptr;	/* resident table (device) */
	bool				table_owned = false;
	char			   *d_ctl = nullptr;	/* device copy of ctl */
	char			   *d_slabs = nullptr;
	int					block = 1024, quads = 2;
	/* compaction of the dense ids that occur (census -> compact) */
	char			   *d_census = nullptr;	/* bitmap over dense ids (device) */
	char			   *d_remap = nullptr;
	std::vector<cl_uint> present;			/* table slot -> dense id */
	size_t				nfolds = 0;
	/* hashed GROUP BY (strom_gpupreagg_create_hashed) */
	bool				hashed = false;
	char			   *htab = nullptr;
	size_t				htab_bytes = 0;
	cl_uint				hash_capacity = 0;
	cl_ulong			groups_upper = 0;	/* groups known at the last read-back + rows folded since */
	std::atomic<cl_uint> groups_known{0};	/* groups at the last read-back (or the caller's hint) */
	int					reg_groups = 0;		/* 1: register accumulators, 2: lane-private LDS, 0: LDS atomics */
	/* packed accumulators (gpupreagg_packed_column): what the generated code says
	 * about every aggregate, and the launch geometry per role count */
	/* a partial accumulated in the 64-bit numeric form (compare-and-swap in LDS,
	 * checked merge): LDS-atomics kernel only, one replica, no RCCL SUM */
	bool				numeric_aggs = false;
	bool				packable = false;
	std::vector<int>	pack_kind, pack_attno;
	struct packed_geom {
		dense_ctl	ctl;
		char	   *d_ctl = nullptr;
		char	   *d_slabs = nullptr;		/* two buffers of nslabs * slab_bytes (see slab_turn) */
		size_t		lds_bytes = 0;
	};
	/*
	 * Fold k+1 runs while merge k adds its slabs to the table (resident chunks: the
	 * merge has a stream of its own): slabs are double-buffered, slab_turn picks the
	 * buffer, merge_ev[b] is recorded behind the merge that last read buffer b --
	 * the next fold into b waits for it, and so does the next merge (the table is
	 * read-modify-written in request order)
	 */
	unsigned			slab_turn = 0;
	hipEvent_t			merge_ev[2] = {nullptr, nullptr};
	bool				merge_ev_used[2] = {false, false};
	std::mutex			launch_lock;
	std::map<cl_uint, packed_geom> packed;		/* by number of roles */
	size_t				packed_static_lds = ~(size_t)0;
	std::mutex			lock;
	/*
	 * integer sums never wrap (strom_gpupreagg.h): per aggregate the static magnitude bound
	 * the code generator found (GPUPREAGG_SUMBITS_<a>: 0 not an integer sum, 1..63 bits, 64
	 * none), the OR of the static bounds (preset in every request's kern_gpupreagg), the
	 * aggregates whose sums are 128 bits wide in the resident table, and the program built
	 * with GPUPREAGG_CHECKED that folds a chunk whose range proof failed
	 */
	std::vector<int>	sumbits;
	std::vector<std::string> sumbound;	/* sumbits 66: the code generator's bound formula (eval_sum_bound) */
	cl_uint				static_magbits = 0;		/* the largest static bound, in bits */
	bool				sums_measured = false;	/* an integer sum without a static bound */
	std::vector<int>	intsum_of;			/* aggregate -> index among the integer sums, or -1 */
	int					nintsums = 0;
	strom_devprog_key	key_checked = 0;
	Program			   *prog_checked = nullptr;
	std::atomic<cl_uint> checked_folds{0};	/* (reported: chunks that took the checked program) */
	cl_uint				sum_turn = 0;		/* hashed: parity of the next fold (gpupreagg_hash_sum_account) */
	/* join-as-a-lookup: the program built FOR a column mapping (lookup_program), by its defines */
	std::map<std::string, std::pair<strom_devprog_key, Program *>> lookup_programs;	/* (lookup_mapping_program) */
	char			   *d_export_spec = nullptr;	/* preagg_export_spec of this table (fetch on the device) */

	/* mirrors gpupreagg_image_offset / gpupreagg_table_offset of
	 * strom_gpupreagg.h: section 0 = flags, 1+a = values of aggregate a,
	 * 1+naggs = total */
	size_t flag_width() const
	{
		size_t n = agg_resno.size();
		return n <= 7 ? 1 : (n <= 15 ? 2 : 4);
	}
	size_t image_offset(int sec, cl_uint G, cl_uint REP) const
	{
		size_t	off = 0;
		int		cur = 0;
		if (sec == cur) return off;
		off += align16(flag_width() * (size_t)G * REP); cur++;
		for (int resno : agg_resno)
		{
			bool nrows = (targets[resno].kind == STROM_PREAGG_NROWS);
			if (sec == cur) return off;
			off += align16((nrows ? 4 : 8) * (size_t)G * REP);
			cur++;
		}
		return off;
	}
	size_t table_offset(int sec, cl_uint N) const
	{
		size_t	flags = STROM_TYPEALIGN(256, sizeof(cl_uint) * (size_t)N);
		size_t	vals = STROM_TYPEALIGN(256, 8 * (size_t)N);
		if (sec == 0) return 0;
		return flags + vals * (size_t)(sec - 1);
	}
	int nsections() const { return 1 + (int)agg_resno.size(); }
	/* the table has one more section per integer sum: its high word (section 1 + naggs + j) */
	int table_sections() const { return nsections() + nintsums; }
	size_t table_hi_offset(int a, cl_uint N) const { return table_offset(nsections() + intsum_of[a], N); }
	bool is_intsum(int a) const { return intsum_of[a] >= 0; }
};

namespace {

/* everything queued for the session's table has landed: folds (streams[0]) and
 * the merges that run behind them on the merge stream */
hipError_t
session_quiesce(strom_gpupreagg *sess)
{
	hipError_t rc = hipStreamSynchronize(sess->dev->streams[0]);
	if (rc == hipSuccess && sess->dev->merge_stream)
		rc = hipStreamSynchronize(sess->dev->merge_stream);
	return rc;
}

int
type_length(int type_oid)
{
	switch (type_oid)
	{
		case STROM_BOOLOID: case STROM_BPCHAROID:	return 1;
		case STROM_INT2OID:							return 2;
		case STROM_INT4OID: case STROM_FLOAT4OID: case STROM_DATEOID:	return 4;
		default:									return 8;
	}
}

bool
type_is_float(int type_oid)
{
	return type_oid == STROM_FLOAT4OID || type_oid == STROM_FLOAT8OID;
}

/*
 * geometry for a domain: how the dense ids are laid over LDS
 */
int	setup_layout(strom_gpupreagg *sess);

int
setup_geometry(strom_gpupreagg *sess, const strom_preagg_domain *dom)
{
	dense_ctl  &ctl = sess->ctl;

	if (dom->nkeys != (int)sess->key_resno.size() || dom->nkeys > STROM_PREAGG_MAXKEYS)
		return StromError_BadRequestMessage;
	memset(&ctl, 0, sizeof(ctl));
	ctl.nkeys = dom->nkeys;
	cl_ulong	ngroups = 1;
	for (int k = 0; k < dom->nkeys; k++)
	{
		ctl.key_min[k] = dom->key_min[k];
		ctl.key_range[k] = dom->key_range[k];
		ctl.key_stride[k] = (cl_uint)ngroups;
		ngroups *= (cl_ulong)dom->key_range[k] + 1;		/* + NULL slot */
		if (ngroups > (1UL << 26))
			return StromError_DataStoreOutOfRange;		/* too sparse for dense ids */
	}
	ctl.ngroups = (cl_uint)ngroups;
	ctl.dense_ngroups = (cl_uint)ngroups;
	ctl.remap = 0;
	sess->present.clear();
	return setup_layout(sess);
}

/*
 * how ctl.ngroups table slots are laid over LDS
 */
int
setup_layout(strom_gpupreagg *sess)
{
	Device	   *dev = sess->dev;
	dense_ctl  &ctl = sess->ctl;
	size_t		lds_budget = std::min<size_t>(dev->prop.sharedMemPerBlock, 160 * 1024) - 8192;	/* static LDS: remap stage, scan scratch */

	if (const char *v = getenv("STROM_GPUPREAGG_LDS_BUDGET"))
		lds_budget = (size_t)atol(v);
	size_t	one = sess->image_offset(sess->nsections(), ctl.ngroups, 1);
	if (one <= lds_budget)
	{
		ctl.nsplits = 1;
		ctl.groups_per_split = ctl.ngroups;
		/* replicas: as many as fit, power of two, at most one per lane of
		 * a wave -- a wave's 64 lanes then never share an address */
		cl_uint rep = 1;
		while (rep < 64 &&
			   sess->image_offset(sess->nsections(), ctl.ngroups, rep * 2) <= lds_budget / 2)
			rep *= 2;
		ctl.nrep = rep;
	}
	else
	{
		cl_uint nsplits = (cl_uint)((one + lds_budget - 1) / lds_budget);
		for (;;)
		{
			cl_uint G = (ctl.ngroups + nsplits - 1) / nsplits;
			if (sess->image_offset(sess->nsections(), G, 1) <= lds_budget)
			{
				ctl.groups_per_split = G;
				break;
			}
			nsplits++;
		}
		ctl.nsplits = nsplits;
		ctl.nrep = 1;
		if (nsplits > 64)
			return StromError_DataStoreOutOfRange;
	}
	if (const char *v = getenv("STROM_GPUPREAGG_NREP"))
		ctl.nrep = std::max(1, atoi(v));
	if (sess->numeric_aggs)
		ctl.nrep = 1;			/* replicas are folded with plain merges: no error path there */
	/*
	 * a handful of groups: no LDS atomics at all.  One group (no GROUP BY):
	 * register accumulators (gpupreagg_reg1_column); up to 32: lane-private
	 * LDS accumulators (gpupreagg_priv_column) when [group][thread] arrays
	 * for a 256-thread work-group fit the budget.
	 */
	sess->reg_groups = 0;
	size_t	priv_bytes = 0;
	if (ctl.nsplits == 1 && ctl.ngroups <= 32 && !sess->numeric_aggs && !getenv("STROM_GPUPREAGG_NO_REG"))
	{
		size_t	priv_budget = 64 * 1024;
		if (const char *v = getenv("STROM_GPUPREAGG_PRIV_LDS"))
			priv_budget = (size_t)atol(v);
		size_t	per_entry = 0;
		for (int resno : sess->agg_resno)
			per_entry += (sess->targets[resno].kind == STROM_PREAGG_NROWS ? 4 : 8);
		priv_bytes = sess->image_offset(sess->nsections(), ctl.ngroups, 1) +
			(size_t)ctl.ngroups * 256 * per_entry;
		/* (lane-private accumulators cost a ds_read + ds_write per aggregate and row, LDS atomics
		 * one ds_add: with Q1's nine aggregates over three groups the private form is the slower
		 * one -- 765 against 749 us per 1e8 rows, profiles/r03_q1_lds_offsets.txt -- so it is
		 * kept for the narrow aggregates it was built for) */
		size_t	priv_entry_max = 32;
		if (const char *v = getenv("STROM_GPUPREAGG_PRIV_ENTRY"))
			priv_entry_max = (size_t)atol(v);
		if (ctl.ngroups == 1)
			sess->reg_groups = 1;
		else if (priv_bytes <= priv_budget && per_entry <= priv_entry_max)
			sess->reg_groups = 2;
		if (sess->reg_groups)
			ctl.nrep = 1;
	}
	sess->lds_bytes = sess->image_offset(sess->nsections(), ctl.groups_per_split, ctl.nrep);
	if (sess->reg_groups == 2)
		sess->lds_bytes = priv_bytes;
	ctl.slab_bytes = STROM_TYPEALIGN(256, sess->image_offset(sess->nsections(), ctl.groups_per_split, 1));
	sess->table_bytes = sess->table_offset(sess->table_sections(), ctl.ngroups);
	/* work-groups: fill the CUs at the occupancy LDS allows */
	size_t	per_cu = std::max<size_t>(1, std::min<size_t>((size_t)dev->prop.sharedMemPerBlock / (sess->lds_bytes + 4608),	/* + static LDS */
														  2048 / sess->block));
	if (const char *v = getenv("STROM_GPUPREAGG_BLOCKS_PER_CU"))
		per_cu = std::max(1, atoi(v));
	size_t	wgs = (size_t)dev->prop.multiProcessorCount * per_cu;
	wgs = std::max<size_t>(ctl.nsplits, wgs - wgs % ctl.nsplits);
	/* several roles per tile: a multiple of 8 XCDs x nsplits lets the kernel
	 * keep the roles of one tile stream on one XCD (shared L2) */
	if (ctl.nsplits > 1 && wgs >= 8 * (size_t)ctl.nsplits)
		wgs -= wgs % (8 * (size_t)ctl.nsplits);
	if (sess->reg_groups)								/* 256-thread work-groups */
	{
		size_t	fit = std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) /
															(sess->lds_bytes + 4608)));
		if (const char *v = getenv("STROM_GPUPREAGG_BLOCKS_PER_CU"))
			fit = std::max(1, atoi(v));
		wgs = (size_t)dev->prop.multiProcessorCount * fit;
	}
	ctl.nslabs = (cl_uint)wgs;
	ctl.merge_ws = 0;
	if (const char *v = getenv("STROM_GPUPREAGG_MERGE_WS"))
	{
		int		w = atoi(v);
		if (w >= 1 && w <= 64 && (w & (w - 1)) == 0)
			ctl.merge_ws = (cl_uint)w;
	}
	sess->has_domain = true;
	return 0;
}

int
alloc_session_buffers(strom_gpupreagg *sess)
{
	Device *dev = sess->dev;
	(void)hipSetDevice(dev->hip_id);
	if (!sess->table)
	{
		sess->table = (char *)dev->pool.alloc(sess->table_bytes);
		if (!sess->table)
			return StromError_OutOfMemory;
		sess->table_owned = true;
		if (hipMemset(sess->table, 0, sess->table_bytes) != hipSuccess)
			return StromError_HipInternal;
	}
	sess->d_ctl = (char *)dev->pool.alloc(sizeof(dense_ctl));
	sess->d_slabs = (char *)dev->pool.alloc(2 * (size_t)sess->ctl.nslabs * sess->ctl.slab_bytes);
	if (!sess->d_ctl || !sess->d_slabs)
		return StromError_OutOfMemory;
	if (hipMemcpy(sess->d_ctl, &sess->ctl, sizeof(dense_ctl), hipMemcpyHostToDevice) != hipSuccess)
		return StromError_HipInternal;
	return 0;
}

/* host image of struct gpupreagg_joined_map (strom_gpupreagg.h) */
struct joined_map_image {
	cl_uint		ncols;
	cl_int		key_col;
	cl_int		key_attlen;
	cl_uint		nslots;
	cl_long		key_min;
	struct {
		cl_int		depth;
		cl_int		col;
		cl_ulong	dimvalues;
		cl_ulong	dimisnull;
	} c[64];
	cl_ulong	recs;
	cl_uint		reclen;
	cl_uint		narrow;
	cl_uint		nshift[64];
	cl_uint		nmask[64];
	cl_long		nmin[64];
};

/* mirrors struct gpupreagg_pack_ctl of strom_gpupreagg.h */
struct pack_ctl {
	cl_uint		count_shift;
	cl_uint		nwords;
	cl_uint		spill_at;
	cl_uint		count_limit;
	cl_uint		shift[32];
	cl_uint		word[32];
	cl_ulong	mask[32];
	cl_ulong	vmax[32];
	cl_long		bias[32];
};

int
bits_for(cl_ulong v)			/* bits needed to hold values 0..v */
{
	int		n = 0;
	while (v) { n++; v >>= 1; }
	return n;
}

/*
 * Packed accumulators for THIS chunk?  (strom_gpupreagg.h explains the
 * layout.)  Yes when the generated code says every aggregate is count(*) or
 * psum of a plain column, no input column has a NULL bitmap in the chunk,
 * the integer columns' zone maps and the rows one work-group folds bound
 * all fields to one 64-bit word, and the packed image needs FEWER id-range
 * roles than the standard one (each role visits every row).
 * Returns the geometry to launch with, or NULL.
 */
strom_gpupreagg::packed_geom *
packed_plan(strom_gpupreagg *sess, hipFunction_t fn_packed, const kern_coldir *coldir, cl_uint ncols,
			cl_uint nitems, pack_ctl *pk)
{
	Device	   *dev = sess->dev;
	dense_ctl  &std_ctl = sess->ctl;

	if (!sess->packable || !fn_packed || !coldir || std_ctl.remap != 0 || std_ctl.nsplits < 2 ||
		getenv("STROM_GPUPREAGG_NO_PACKED"))
		return nullptr;
	if (sess->packed_static_lds == ~(size_t)0)
	{
		int		v = 0;
		if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fn_packed) != hipSuccess || v < 0)
			v = 8192;
		sess->packed_static_lds = (size_t)v;
	}
	size_t		lds_max = std::min<size_t>(dev->prop.sharedMemPerBlock, 160 * 1024);
	if (sess->packed_static_lds + 1024 >= lds_max)
		return nullptr;
	size_t		lds_budget = lds_max - sess->packed_static_lds - 256;
	cl_uint		naggs = (cl_uint)sess->agg_resno.size();
	cl_uint		nwords = 1;
	memset(pk, 0, sizeof(*pk));
	for (cl_uint a = 0; a < naggs; a++)
	{
		int		kind = sess->pack_kind[a];
		if (kind == 1)
			continue;
		int		col = sess->pack_attno[a] - 1;
		if (col < 0 || col >= (int)ncols || coldir[col].nulls_off != 0)
			return nullptr;					/* a NULL input needs the has-value flags */
		if (kind == 3)
			pk->word[a] = nwords++;
		else if (!(coldir[col].stat_flags & KDS_COLSTAT_MINMAX) || (coldir[col].stat_flags & KDS_COLSTAT_ISFLOAT) ||
				 coldir[col].maxval < coldir[col].minval)
			return nullptr;					/* no zone map to bound the sum with */
	}
	/* roles and work-groups of the packed image */
	size_t		per_group = 8 * (size_t)nwords;
	cl_uint		nsplits = 1;
	cl_uint		G = std_ctl.ngroups;
	while ((size_t)nwords * align16(8 * (size_t)G) > lds_budget)
	{
		nsplits++;
		G = (std_ctl.ngroups + nsplits - 1) / nsplits;
	}
	(void)per_group;
	if (nsplits >= std_ctl.nsplits)
		return nullptr;						/* nothing gained */
	size_t		wgs = (size_t)dev->prop.multiProcessorCount;		/* the image takes a CU's LDS */
	wgs = std::max<size_t>(nsplits, wgs - wgs % nsplits);
	if (nsplits > 1 && wgs >= 8 * (size_t)nsplits)
		wgs -= wgs % (8 * (size_t)nsplits);
	/* field widths: rows one work-group can fold in this chunk, zone-map ranges */
	size_t		tile_rows = (size_t)sess->block * 4 * sess->quads;
	size_t		ntiles = ((size_t)nitems + tile_rows - 1) / tile_rows;
	size_t		wgs_per_split = wgs / nsplits;
	cl_ulong	rows_per_wg = (cl_ulong)((ntiles + wgs_per_split - 1) / wgs_per_split) * tile_rows;
	/*
	 * count field: wide enough for every row the work-group may fold -- or, when the sums do not
	 * leave that much room, NARROW: the fold then adds with the returning atomic and moves a group
	 * to the slab when its count reaches a quarter of the field (strom_gpupreagg.h:
	 * gpupreagg_packed_spill; three quarters of the field are the margin for the adds in flight, so
	 * the field must hold a few tiles' worth of rows).  Uniform keys never get there: 1e4 groups
	 * share the 4e5 rows of a work-group.  (Round 2 flushed the whole table every few tiles
	 * instead -- "epochs": 0.7 GB of extra slab traffic per 1e8 rows and two barriers in the tile
	 * loop, whatever the key distribution.)
	 */
	int			vbits = 0, nsums = 0;
	for (cl_uint a = 0; a < naggs; a++)
	{
		if (sess->pack_kind[a] != 2)
			continue;
		int			col = sess->pack_attno[a] - 1;
		vbits += bits_for((cl_ulong)coldir[col].maxval - (cl_ulong)coldir[col].minval);
		nsums++;
	}
	if (vbits >= 64)
		return nullptr;
	int			cbits = bits_for(rows_per_wg);
	cl_uint		spill_at = 0, count_limit = 0;
	const char *cap = getenv("STROM_GPUPREAGG_PACK_COUNT_BITS");	/* tests: narrow fields on small inputs */
	int			cap_bits = (cap ? atoi(cap) : 0);
	if (vbits + (nsums + 1) * cbits > 64 || (cap_bits > 0 && cap_bits < cbits))
	{
		cbits = (64 - vbits) / (nsums + 1);
		if (cap_bits > 0)
			cbits = std::min(cbits, cap_bits);
		/* 2^cbits rows: at least 4 tiles (a quarter to get there, three for the adds in flight) */
		if (cbits > 31 || (1UL << cbits) < 4 * tile_rows || getenv("STROM_GPUPREAGG_NO_SPILL"))
			return nullptr;
		spill_at = 1u << (cbits - 2);
		count_limit = (1u << cbits) - 1;
	}
	pk->spill_at = spill_at;
	pk->count_limit = count_limit;
	int			pos = 0;
	for (cl_uint a = 0; a < naggs; a++)
	{
		if (sess->pack_kind[a] != 2)
			continue;
		int			col = sess->pack_attno[a] - 1;
		cl_ulong	vrange = (cl_ulong)coldir[col].maxval - (cl_ulong)coldir[col].minval;
		int			width = cbits + bits_for(vrange);
		if (pos + width > 64)
			return nullptr;
		pk->shift[a] = (cl_uint)pos;
		pk->mask[a] = (width >= 64 ? ~0UL : ((1UL << width) - 1));
		pk->vmax[a] = vrange;
		pk->bias[a] = coldir[col].minval;
		pos += width;
	}
	if (pos + cbits > 64)
		return nullptr;
	pk->count_shift = (cl_uint)pos;
	pk->nwords = nwords;
	/* geometry of this role count: control block + slabs, made once */
	std::lock_guard<std::mutex> g(sess->lock);
	auto	it = sess->packed.find(nsplits);
	if (it == sess->packed.end())
	{
		strom_gpupreagg::packed_geom geom;
		geom.ctl = std_ctl;
		geom.ctl.nsplits = nsplits;
		geom.ctl.groups_per_split = G;
		geom.ctl.nrep = 1;
		geom.ctl.nslabs = (cl_uint)wgs;
		geom.ctl.slab_bytes = STROM_TYPEALIGN(256, sess->image_offset(sess->nsections(), G, 1));
		geom.lds_bytes = (size_t)nwords * align16(8 * (size_t)G);
		(void)hipSetDevice(dev->hip_id);
		geom.d_ctl = (char *)dev->pool.alloc(sizeof(dense_ctl));
		geom.d_slabs = (char *)dev->pool.alloc(2 * (size_t)geom.ctl.nslabs * geom.ctl.slab_bytes);
		if (!geom.d_ctl || !geom.d_slabs ||
			hipMemcpy(geom.d_ctl, &geom.ctl, sizeof(dense_ctl), hipMemcpyHostToDevice) != hipSuccess)
		{
			if (geom.d_ctl) dev->pool.release(geom.d_ctl);
			if (geom.d_slabs) dev->pool.release(geom.d_slabs);
			return nullptr;
		}
		it = sess->packed.insert({nsplits, geom}).first;
	}
	return &it->second;
}

#define REQ_CHECK(call, what)												\
	do {																	\
		hipError_t __rc = (call);											\
		if (__rc != hipSuccess)												\
		{																	\
			task_fail(task, hip_errcode(__rc, what));						\
			return;															\
		}																	\
	} while (0)

struct preagg_request {
	strom_gpupreagg	   *sess;
	const kern_data_store *kds;
	strom_dstore	   *kds_dev;
	const kern_row_map *krowmap;
	strom_rowmap	   *rowmap_dev;		/* device-resident row map (chained operators) */
	uint32_t			format;
	uint32_t			nrows;
	/* rows = a finished GpuHashJoin's result pairs (strom_submit_gpupreagg_joined) */
	const void		   *joined_results = nullptr;	/* device kern_resultbuf */
	void			   *joined_buffer = nullptr;	/* the join's device image, owned by this request now */
	bool				lookup = false;				/* no result pairs: the join is a lookup in the aggregate's pass */
	Program			   *prog = nullptr;				/* lookup: the session's program built for this column mapping */
	std::shared_ptr<std::vector<char>> joined_map;	/* host image of gpupreagg_joined_map */
	/* hashed, exact fold (gpupreagg_hashed_exact): 'sess' is a scratch session of this one */
	strom_gpupreagg	   *exact_parent = nullptr;
};

/* bits of the largest magnitude a zone map allows (strom_gpupreagg.h: gpupreagg_sum_magnitude) */
cl_uint
zone_magbits(const kern_coldir &cd)
{
	cl_long		lo = cd.minval, hi = cd.maxval;
	return (cl_uint)bits_for((cl_ulong)(lo ^ (lo >> 63)) | (cl_ulong)(hi ^ (hi >> 63)));
}

/*
 * bits of the largest magnitude a summed EXPRESSION can take in this chunk: the code generator's
 * formula (codegen_preagg.cpp: sum_bound_formula -- reverse Polish over magnitudes: cN column N's
 * zone map, nN a numeric image column's integer-part bounds, kV a constant, + and *, eK a rescale by
 * 10^K) over the chunk's zone maps.  -1: a column
 * without a zone map, or a malformed formula -- the fold measures the rows then.
 */
int
eval_sum_bound(const std::string &formula, const kern_coldir *cd, cl_uint ncd)
{
	std::vector<unsigned __int128> st;
	const unsigned __int128 CAP = (unsigned __int128)1 << 100;		/* saturate well above 2^64 */
	const char *p = formula.c_str();
	while (*p)
	{
		while (*p == ' ')
			p++;
		if (!*p)
			break;
		char	op = *p++;
		if (op == 'c' || op == 'k' || op == 'e' || op == 'n')
		{
			char   *end;
			unsigned long long v = strtoull(p, &end, 10);
			if (end == p)
				return -1;
			p = end;
			if (op == 'c' || op == 'n')
			{
				if (v < 1 || v > ncd)
					return -1;
				const kern_coldir &c = cd[v - 1];
				/* (cN: an integer-like column's zone map; nN: a numeric image column's integer-part bounds) */
				if (op == 'c' ? (!(c.stat_flags & KDS_COLSTAT_MINMAX) || (c.stat_flags & KDS_COLSTAT_ISFLOAT))
					: !(c.stat_flags & KDS_COLSTAT_INTPART))
					return -1;
				if (c.maxval < c.minval)
					return -1;
				cl_ulong lo = (cl_ulong)(c.minval < 0 ? -(unsigned __int128)c.minval : (unsigned __int128)c.minval);
				cl_ulong hi = (cl_ulong)(c.maxval < 0 ? -(unsigned __int128)c.maxval : (unsigned __int128)c.maxval);
				st.push_back(std::max(lo, hi));
			}
			else if (op == 'k')
				st.push_back(v);
			else
			{
				if (st.empty() || v > 38)
					return -1;
				for (unsigned long long i = 0; i < v && st.back() < CAP; i++)
					st.back() *= 10;
			}
		}
		else if (op == '+' || op == '*')
		{
			if (st.size() < 2)
				return -1;
			unsigned __int128 b = st.back();
			st.pop_back();
			unsigned __int128 a = st.back();
			a = std::min(a, CAP);
			b = std::min(b, CAP);
			st.back() = (op == '+' ? a + b : (a == 0 || b == 0) ? 0 : (a > CAP / b ? CAP : a * b));
		}
		else
			return -1;
	}
	if (st.size() != 1)
		return -1;
	unsigned __int128 v = st.back();
	int		bits = 0;
	while (v != 0 && bits < 127)
	{
		v >>= 1;
		bits++;
	}
	return bits;
}

/*
 * join-as-a-lookup: the session's source built FOR a column mapping -- which virtual column is an
 * inner one, the slot records' length and form, the key's width become compile-time constants
 * (strom_gpupreagg.h: gpupreagg_dense_lookup_body says what that is worth), and only the lookup
 * and merge kernels are built.  One program per distinct mapping, kept with the session; the first
 * request of a mapping parks behind its build like any request behind a cold program.
 */
Program *
lookup_mapping_program(strom_gpupreagg *sess, const joined_map_image *jm)
{
	if (getenv("STROM_GPUPREAGG_LOOKUP_GENERIC"))
		return nullptr;						/* the run-time form: any mapping, no build */
	cl_ulong	inner_mask = 0;
	for (cl_uint i = 0; i < jm->ncols && i < 64; i++)
		if (jm->c[i].depth != 0)
			inner_mask |= (1UL << i);
	char		defs[512];
	snprintf(defs, sizeof(defs),
			 "#define GPUPREAGG_LOOKUP_ONLY 1\n#define GPUPREAGG_LOOKUP_INNER_MASK 0x%lxUL\n"
			 "#define GPUPREAGG_LOOKUP_RECLEN %u\n#define GPUPREAGG_LOOKUP_NARROW %u\n"
			 "#define GPUPREAGG_LOOKUP_KEYLEN %d\n",
			 (unsigned long)inner_mask, jm->reclen, jm->narrow ? 1u : 0u, jm->key_attlen);
	std::lock_guard<std::mutex> g(sess->lock);
	auto	it = sess->lookup_programs.find(defs);
	if (it != sess->lookup_programs.end())
		return it->second.second;
	std::string	source = std::string(defs) + sess->prog->source;
	strom_devprog_key key = strom_get_devprog_key(source.c_str(), sess->prog->extra_flags);
	Program	   *prog = (key ? lookup_program(key) : nullptr);
	if (!prog)
		return nullptr;
	sess->lookup_programs[defs] = std::make_pair(key, prog);
	return prog;
}

/*
 * the program that folds a chunk whose integer sums could not be proven to stay in
 * int8 (strom_gpupreagg.h, "integer sums never wrap"): the session's own source built with
 * GPUPREAGG_CHECKED.  Made on first use -- such chunks are rare -- and kept with the session.
 */
int
ensure_checked_program(strom_gpupreagg *sess)
{
	std::lock_guard<std::mutex> g(sess->lock);
	if (sess->prog_checked)
		return 0;
	std::string	source = "#define GPUPREAGG_CHECKED 1\n" + sess->prog->source;
	strom_devprog_key key = strom_get_devprog_key(source.c_str(), sess->prog->extra_flags);
	Program	   *prog = (key ? lookup_program(key) : nullptr);
	if (!prog)
		return StromError_OutOfMemory;
	sess->key_checked = key;
	sess->prog_checked = prog;
	return 0;
}

void
gpupreagg_launch(strom_task_impl *task, preagg_request req, bool checked = false)
{
	strom_gpupreagg *sess = req.sess;
	Device	   *dev = task->dev;
	Program	   *prog = (checked ? sess->prog_checked : req.prog ? req.prog : sess->prog);
	int			errcode = 0;

	(void)hipSetDevice(dev->hip_id);
	if (req.joined_buffer && !checked)
		task->devbufs.push_back(req.joined_buffer);		/* released with this task, whatever happens */
	if (prog->state != STROM_DEVPROG_READY)
	{
		task_fail(task, StromError_ProgramBuildFailure);
		return;
	}
	task->pfm.time_kern_build = (cl_ulong)prog->build_usec;
	/* chunks fold into one table: keep them in order on one stream */
	task->stream = dev->streams[0];
	/*
	 * resident chunk: no bulk DMA, so the small copies and the slab merge leave
	 * the fold stream -- the request head goes down on copy_in, the merge of
	 * chunk k (and its status read-back) runs on merge_stream while chunk k+1 is
	 * folded.  Measured for C4 (1e9 rows in 10 chunks): 317 us of wall clock per
	 * chunk with everything on one stream, of which 250 us are the fold kernel.
	 */
	bool		piped = (req.kds_dev != nullptr && dev->copy_in && dev->merge_stream &&
						 !getenv("STROM_GPUPREAGG_NO_PIPE"));
	hipStream_t	s_in = (piped ? dev->copy_in : task->stream);
	hipStream_t	s_mrg = (piped ? dev->merge_stream : task->stream);
	/* one launch at a time per session: the slab turn and the event chain are ordered */
	std::lock_guard<std::mutex> launch_guard(sess->launch_lock);

	bool	use_lookup = req.lookup;
	bool	use_joined = (req.joined_results != nullptr);
	bool	use_column = (!use_joined && !use_lookup && req.format == KDS_FORMAT_COLUMN &&
						  req.krowmap == nullptr && req.rowmap_dev == nullptr);
	/* (the checked program adds in LDS with returning atomics: the LDS-atomics kernels only) */
	bool	use_reg = (use_column && sess->reg_groups != 0 && !checked);
	hipFunction_t fn = prog->get_function(dev, use_lookup ? "gpupreagg_dense_lookup"
										  : use_joined ? "gpupreagg_dense_joined"
										  : use_reg ? (sess->reg_groups == 1 ? "gpupreagg_reg1_column"
																			  : "gpupreagg_priv_column")
										  : use_column ? "gpupreagg_dense_column"
										  : "gpupreagg_dense_generic", &errcode);
	hipFunction_t fn_merge = fn ? prog->get_function(dev, "gpupreagg_dense_merge", &errcode) : nullptr;
	if (!fn || !fn_merge)
	{
		task_fail(task, errcode);
		return;
	}
	/* what is known about the integer sums' inputs before the fold measures the rest */
	cl_uint		magbits = sess->static_magbits;
	/* packed accumulators for this chunk?  (fewer id-range roles: see packed_plan) */
	pack_ctl	pk;
	strom_gpupreagg::packed_geom *packed = nullptr;
	if ((use_lookup || (use_column && !use_reg)) && sess->packable && sess->ctl.nsplits > 1 && !checked)
	{
		int		e2 = 0;
		hipFunction_t fn_packed = prog->get_function(dev, use_lookup ? "gpupreagg_packed_lookup"
													 : "gpupreagg_packed_column", &e2);
		std::shared_ptr<std::vector<kern_coldir>> snap;
		std::vector<kern_coldir> virt;
		const kern_coldir *coldir = nullptr;
		cl_uint		ncols = 0;
		if (req.kds)
		{
			coldir = KERN_DATA_STORE_COLDIR(req.kds);
			ncols = req.kds->ncols;
		}
		else if ((snap = dstore_coldir(req.kds_dev)) != nullptr)
		{
			coldir = snap->data();
			ncols = (cl_uint)snap->size();
		}
		if (use_lookup && coldir)
		{
			/* the program's columns are virtual: an outer column brings its own
			 * directory entry, an inner column counts as "may be NULL, no zone map" */
			const joined_map_image *jm = (const joined_map_image *)req.joined_map->data();
			virt.resize(jm->ncols);
			for (cl_uint i = 0; i < jm->ncols; i++)
			{
				memset(&virt[i], 0, sizeof(kern_coldir));
				if (jm->c[i].depth == 0 && jm->c[i].col >= 0 && (cl_uint)jm->c[i].col < ncols)
					virt[i] = coldir[jm->c[i].col];
				else
					virt[i].nulls_off = 1;
			}
			coldir = virt.data();
			ncols = jm->ncols;
		}
		packed = packed_plan(sess, fn_packed, coldir, ncols, req.nrows, &pk);
		if (packed)
		{
			fn = fn_packed;
			/* the packed fold measures nothing: the zone maps that bound its fields bound
			 * the integer sums' inputs too */
			for (size_t a = 0; a < sess->agg_resno.size(); a++)
				if (sess->pack_kind[a] == 2 && sess->sumbits[a] >= 64)
					magbits = std::max(magbits, zone_magbits(coldir[sess->pack_attno[a] - 1]));
		}
	}
	/* kern_gpupreagg image: {status, sortbuf_len, pad, kern_parambuf} [+ the pack control block] */
	size_t	kg_len = STROMALIGN(offsetof(kern_gpupreagg, kparams) + sess->kparams.size());
	size_t	send_len = kg_len + (packed ? STROMALIGN(sizeof(pack_ctl)) : 0);
	char   *stage = dev->pinned.alloc();
	char   *d_kg = (char *)dev->pool.alloc(send_len);
	if (!stage || !d_kg || send_len + 64 > PinnedPool::BLOCK)
	{
		if (stage) dev->pinned.release(stage);
		if (d_kg) dev->pool.release(d_kg);
		task_fail(task, StromError_OutOfMemory);
		return;
	}
	task->pinned_blocks.push_back(stage);
	task->main_devptr = d_kg;
	memset(stage, 0, kg_len);
	memcpy(stage + offsetof(kern_gpupreagg, kparams), sess->kparams.data(), sess->kparams.size());
	{
		/*
		 * the range proof's inputs (strom_gpupreagg.h, "integer sums never wrap"): rows of the
		 * request, bits of the largest input magnitude known so far, and the rows ONE work-group
		 * folds at most -- tiles (or, row by row, blocks) are dealt round-robin to the work-groups
		 * of a role; one unit of slack
		 */
		const dense_ctl &g = (packed ? packed->ctl : sess->ctl);
		size_t		unit = (use_reg ? (size_t)256 * 4 * sess->quads
							: (use_column || use_lookup) ? (size_t)sess->block * 4 * sess->quads
							: (size_t)sess->block);
		size_t		units = ((size_t)req.nrows + unit - 1) / unit;
		size_t		wgs_per_role = std::max<size_t>(1, g.nslabs / std::max<cl_uint>(1, g.nsplits));
		size_t		wg_rows = std::min<size_t>(((units + wgs_per_role - 1) / wgs_per_role + 1) * unit, 0xffffffffUL);
		/*
		 * sums of PLAIN columns (GPUPREAGG_SUMBITS_<a> 65) over a COLUMN chunk with zone maps: bounded
		 * here, the fold does not measure them (top bit of the second word; the streaming kernels
		 * read it, the row-at-a-time ones measure as always)
		 */
		cl_uint		zone_bounded = 0;
		if (!packed && !checked && (use_column || use_reg))
		{
			std::shared_ptr<std::vector<kern_coldir>> snap;
			const kern_coldir *cd = nullptr;
			cl_uint		ncd = 0;
			if (req.kds)
			{
				cd = KERN_DATA_STORE_COLDIR(req.kds);
				ncd = req.kds->ncols;
			}
			else if ((snap = dstore_coldir(req.kds_dev)) != nullptr)
			{
				cd = snap->data();
				ncd = (cl_uint)snap->size();
			}
			bool		all = (cd != nullptr), any = false;
			cl_uint		zbits = 0;
			for (size_t a = 0; all && a < sess->agg_resno.size(); a++)
			{
				if (sess->sumbits[a] == 66)
				{
					/* an expression over decimal columns: the code generator's formula */
					int		bits = (cd ? eval_sum_bound(sess->sumbound[a], cd, ncd) : -1);
					if (bits < 0 || bits > 63)
						all = false;			/* (no zone map -- or a bound beyond int8: measured, then checked) */
					else
					{
						zbits = std::max(zbits, (cl_uint)bits);
						any = true;
					}
					continue;
				}
				if (sess->sumbits[a] != 65)
					continue;
				int		col = (a < sess->pack_attno.size() ? sess->pack_attno[a] - 1 : -1);
				if (col < 0 || col >= (int)ncd || !(cd[col].stat_flags & KDS_COLSTAT_MINMAX) ||
					(cd[col].stat_flags & KDS_COLSTAT_ISFLOAT) || cd[col].maxval < cd[col].minval)
					all = false;
				else
				{
					zbits = std::max(zbits, zone_magbits(cd[col]));
					any = true;
				}
			}
			if (all && any)
			{
				zone_bounded = 0x80000000u;
				magbits = std::max(magbits, zbits);
			}
		}
		cl_uint		words[2] = { magbits, (cl_uint)std::min<size_t>(wg_rows, 0x7fffffffUL) | zone_bounded };
		((kern_gpupreagg *)stage)->sortbuf_len = (cl_int)req.nrows;
		memcpy(((kern_gpupreagg *)stage)->__padding, words, sizeof(words));
	}
	if (packed)
		memcpy(stage + kg_len, &pk, sizeof(pk));

	task_event(task, s_in);								/* ev[0] */
	REQ_CHECK(hipMemcpyAsync(d_kg, stage, send_len, hipMemcpyHostToDevice, s_in),
			  "send kern_gpupreagg");
	task->pfm.num_dma_send++;
	task->pfm.bytes_dma_send += send_len;
	const void *d_kds;
	if (req.kds_dev)
		d_kds = req.kds_dev->devptr;
	else
	{
		size_t	kds_len = req.kds->length;
		if (req.kds->format == KDS_FORMAT_ROW)
			kds_len = KERN_DATA_STORE_ROWBLOCK_OFFSET(req.kds) + (size_t)BLCKSZ * req.kds->nblocks;
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
	if (!req.rowmap_dev && req.krowmap)
	{
		size_t	len = offsetof(kern_row_map, rindex) + sizeof(cl_int) * (size_t)req.krowmap->nvalids;
		void   *p = dev->pool.alloc(len);
		if (!p)
		{
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		task->devbufs.push_back(p);
		REQ_CHECK(hipMemcpyAsync(p, req.krowmap, len, hipMemcpyHostToDevice, task->stream),
				  "send kern_row_map");
		d_rowmap = p;
	}
	void   *d_jmap = nullptr;
	if (use_joined || use_lookup)
	{
		d_jmap = dev->pool.alloc(req.joined_map->size());
		if (!d_jmap)
		{
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		task->devbufs.push_back(d_jmap);
		/* small and pageable: a synchronous copy keeps the source alive */
		REQ_CHECK(hipMemcpy(d_jmap, req.joined_map->data(), req.joined_map->size(), hipMemcpyHostToDevice),
				  "send joined map");
	}
	/* this request's slab buffer; whoever read it last must be through */
	unsigned	turn = (sess->slab_turn++ & 1);
	for (int b = 0; b < 2; b++)
		if (!sess->merge_ev[b] &&
			hipEventCreateWithFlags(&sess->merge_ev[b], hipEventDisableTiming) != hipSuccess)
		{
			task_fail(task, StromError_HipInternal);
			return;
		}
	if (piped)
	{
		hipEvent_t sent = task_event(task, s_in);			/* ev[1]: head is down */
		REQ_CHECK(hipStreamWaitEvent(task->stream, sent, 0), "wait for the request head");
	}
	else
		task_event(task);									/* ev[1] */
	if (sess->merge_ev_used[turn])
		REQ_CHECK(hipStreamWaitEvent(task->stream, sess->merge_ev[turn], 0), "wait for the slab buffer");
	/* piped: the fold kernel carries its own start / stop events (see task_event_slot) */
	bool		ext_launch = (piped && use_ext_launch());
	hipEvent_t	ev_fold_begin = nullptr, ev_fold_done = nullptr;
	if (piped)
	{
		ev_fold_begin = (ext_launch ? task_event_slot(task) : task_event(task));	/* ev[2]: the fold begins */
		if (ext_launch)
			ev_fold_done = task_event_slot(task);			/* ev[3] */
		if (ext_launch && (!ev_fold_begin || !ev_fold_done))
		{
			task_fail(task, StromError_HipInternal);
			return;
		}
	}
	{
		void	   *a_kg = d_kg;
		const void *a_kds = d_kds;
		const void *a_toast = nullptr;
		const void *a_map = d_rowmap;
		const dense_ctl &lctl = (packed ? packed->ctl : sess->ctl);		/* this launch's geometry */
		void	   *a_ctl = (packed ? packed->d_ctl : sess->d_ctl);
		void	   *a_slabs = (packed ? packed->d_slabs : sess->d_slabs) +
			(size_t)turn * lctl.nslabs * lctl.slab_bytes;
		void	   *a_table = sess->table;
		const void *a_res = req.joined_results;
		const void *a_jmap = d_jmap;
		const void *a_pack = d_kg + kg_len;
		void	   *args_col[] = { &a_kg, &a_kds, &a_ctl, &a_slabs };
		void	   *args_pack[] = { &a_kg, &a_kds, &a_ctl, &a_pack, &a_slabs };
		void	   *args_plook[] = { &a_kg, &a_kds, &a_jmap, &a_ctl, &a_pack, &a_slabs };
		void	   *args_gen[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_ctl, &a_slabs };
		void	   *args_join[] = { &a_kg, &a_res, &a_kds, &a_jmap, &a_ctl, &a_slabs };
		void	   *args_look[] = { &a_kg, &a_kds, &a_jmap, &a_ctl, &a_slabs };
		void	   *args_mrg[] = { &a_kg, &a_ctl, &a_slabs, &a_table };
		void	  **fold_args = (packed ? (use_lookup ? args_plook : args_pack)
								 : use_lookup ? args_look : use_joined ? args_join
								 : use_column ? args_col : args_gen);
		unsigned	fold_block = (use_reg ? 256 : (unsigned)sess->block);
		unsigned	fold_lds = (unsigned)(packed ? packed->lds_bytes : sess->lds_bytes);
		if (ext_launch)
			REQ_CHECK(hipExtModuleLaunchKernel(fn, lctl.nslabs * fold_block, 1, 1, fold_block, 1, 1, fold_lds,
											   task->stream, fold_args, nullptr, ev_fold_begin, ev_fold_done, 0),
					  "launch gpupreagg reduction");
		else
			REQ_CHECK(hipModuleLaunchKernel(fn, lctl.nslabs, 1, 1, fold_block, 1, 1, fold_lds, task->stream,
											fold_args, nullptr),
					  "launch gpupreagg reduction");
		if (packed)
			task->pfm.num_kern_prep++;			/* (reported: this request took the packed path) */
		hipEvent_t fold_done = ev_fold_done;
		if (!ext_launch && (task->pfm.enabled || piped))
		{
			fold_done = task_event(task);				/* ev[2] (piped: ev[3]): main kernel done */
			task->has_ev_proj = !piped;
		}
		if (piped)
		{
			/* the merge stream takes over: behind this fold and behind the previous merge */
			if (!fold_done)
			{
				task_fail(task, StromError_HipInternal);
				return;
			}
			REQ_CHECK(hipStreamWaitEvent(s_mrg, fold_done, 0), "merge waits for the fold");
		}
		if (sess->merge_ev_used[turn ^ 1])
			REQ_CHECK(hipStreamWaitEvent(s_mrg, sess->merge_ev[turn ^ 1], 0), "merge waits for the previous merge");
		/* the merge kernel lays 256 threads out as GL group lanes x stripes
		 * over the slabs (strom_gpupreagg.h) */
		unsigned ws = 1, per_split = lctl.nslabs / lctl.nsplits;
		while (ws < 64 && ws * 8 < per_split)
			ws <<= 1;
		if (lctl.merge_ws != 0)
			ws = lctl.merge_ws;
		unsigned gl = 256 / ws;
		unsigned mgrid = std::min<unsigned>((lctl.ngroups + gl - 1) / gl,
											(unsigned)dev->prop.multiProcessorCount * 8);
		/* the slab check: numeric sums; integer sums whose chunk totals the range proof may
		 * not cover (tier 2: the kernel leaves at once when it does) */
		if (sess->numeric_aggs || checked ||
			(sess->nintsums > 0 && (sess->sums_measured || sess->static_magbits > 31)))
		{
			/* numeric sums may leave the 64-bit form while slabs are added up: find
			 * out BEFORE the table takes any of it (gpupreagg_dense_merge_body<CHECK>) */
			int		e3 = 0;
			hipFunction_t fn_check = prog->get_function(dev, "gpupreagg_dense_merge_check", &e3);
			if (!fn_check)
			{
				task_fail(task, e3);
				return;
			}
			REQ_CHECK(hipModuleLaunchKernel(fn_check, std::max(1u, mgrid), 1, 1, 256, 1, 1, 0,
											s_mrg, args_mrg, nullptr),
					  "launch gpupreagg merge check");
			task->pfm.num_kern_exec++;
		}
		REQ_CHECK(hipModuleLaunchKernel(fn_merge, std::max(1u, mgrid), 1, 1, 256, 1, 1, 0,
										s_mrg, args_mrg, nullptr),
				  "launch gpupreagg merge");
		task->pfm.num_kern_exec += 2;
		REQ_CHECK(hipEventRecord(sess->merge_ev[turn], s_mrg), "mark the merge");
		sess->merge_ev_used[turn] = true;
	}
	task_event(task, s_mrg);							/* ev[2] (+1 with the fold event; piped: ev[4]) */
	char   *stage_status = stage + send_len;
	REQ_CHECK(hipMemcpyAsync(stage_status, d_kg + offsetof(kern_gpupreagg, status), sizeof(cl_int),
							 hipMemcpyDeviceToHost, s_mrg),
			  "recv status");
	task->pfm.num_dma_recv++;
	task->pfm.bytes_dma_recv += sizeof(cl_int);
	task_event(task, s_mrg);							/* ev[3] (+1; piped: ev[5]) */
	task->ev_preagg_piped = piped;
	if (checked)
		sess->checked_folds++;
	task->finish = [stage_status, req, checked](strom_task_impl *t)
	{
		cl_int status;
		memcpy(&status, stage_status, sizeof(status));
		if (status == StromError_SumRangeUnproven && !checked)
		{
			/*
			 * the merge could not prove that the chunk's integer sums stay in int8 and left
			 * the table alone: the chunk is folded again by the checked program (on a worker
			 * thread: the program may have to be built first), and this request completes
			 * when that is through
			 */
			t->retry = [t, req]()
			{
				int		rc = ensure_checked_program(req.sess);
				if (rc != 0)
				{
					task_fail(t, rc);
					return;
				}
				program_run_or_park(req.sess->prog_checked, [t, req]() { gpupreagg_launch(t, req, true); });
			};
			return;
		}
		if (status == StromError_SumRangeUnproven)
			status = StromError_CpuReCheck;		/* (not reached: the checked merge has no proof to fail) */
		t->errcode = status;		/* 0, CpuReCheck, or significant */
	};
	task_enqueue(task);
}


/* ------------------------------------------------------------------ *
 * hashed GROUP BY (gpupreagg_hash_* of strom_gpupreagg.h)
 * ------------------------------------------------------------------ */

/* host image of the table: head (GPUPREAGG_HASH_HEAD bytes), then one record
 * per slot { state, knull, flags, pad, keys[nkeys], vals[naggs] } */
struct hash_layout {
	std::vector<char> head;
	size_t		stride;
	size_t		total;
};

const size_t HASH_HEAD_LEN = 256;

hash_layout
hash_table_layout(const strom_gpupreagg *sess, cl_uint capacity)
{
	hash_layout	L;
	size_t		nkeys = sess->key_resno.size(), naggs = sess->agg_resno.size();
	size_t		reclen = 16 + 8 * (nkeys + naggs);

	L.stride = (reclen <= 32 ? 32 : reclen <= 64 ? 64 : (reclen + 127) / 128 * 128);
	L.total = HASH_HEAD_LEN + L.stride * (size_t)capacity;
	L.head.assign(HASH_HEAD_LEN, 0);
	cl_uint	words[6] = { capacity, (cl_uint)nkeys, 0, 0, (cl_uint)L.stride, (cl_uint)naggs };
	memcpy(L.head.data(), words, sizeof(words));
	return L;
}

/* slots of the work-group's LDS table in front of the global one, and its bytes */
cl_uint
hash_lds_slots(const strom_gpupreagg *sess, size_t *p_bytes, bool for_units = false)
{
	size_t		nkeys = sess->key_resno.size();
	/* after the table: 256 queued row numbers per wave (GPUPREAGG_HASH_QUEUE, used with roles) */
	size_t		queue = (for_units ? 16 /* the unit's flags */ : ((size_t)sess->block / 64) * 256 * sizeof(cl_uint));
	auto fit = [&](size_t budget, size_t *p_b) -> cl_uint {
		cl_uint	slots = 16384;
		for (;;)
		{
			*p_b = sess->image_offset(sess->nsections(), slots, 1) + (size_t)slots * (8 + 8 * nkeys) + queue;
			if (*p_b <= budget || slots == 64)
				return slots;
			slots >>= 1;
		}
	};
	/*
	 * LDS per work-group: ONE work-group per CU with a table of up to 136 KB, four times the
	 * slots of the 64 KB table two work-groups per CU had.  Every doubling of the table halves
	 * the hash roles, and each role is a scan of every row's key columns: 1e4 groups 5.1 ->
	 * 3.4 ms per 1e8 rows, 3e4 groups 13.2 -> 7.1 ms; a fuller small table also probes longer,
	 * so the big one wins from 100 groups on (698 -> 656 us; 600 groups 1463 -> 932 us).
	 * profiles/r02_hashed_lds_tables.txt
	 */
	if (for_units)
	{
		/* the partition plan's units: no role queues, and as much of the CU's 160 KB as a
		 * power of two of slots takes */
		size_t	unit_budget = 159 * 1024;
		if (const char *v = getenv("STROM_GPUPREAGG_HASH_UNIT_LDS_KB"))
			unit_budget = (size_t)std::max(8L, std::min(159L, atol(v))) * 1024;
		return fit(unit_budget, p_bytes);
	}
	if (const char *v = getenv("STROM_GPUPREAGG_HASH_LDS_SLOTS"))
	{
		/* tests: a tiny table in front of the global one sends (nearly) every row to the
		 * global table's claim / probe protocol */
		cl_uint		want = (cl_uint)atoi(v);
		if (want >= 64 && want <= 16384 && (want & (want - 1)) == 0)
		{
			*p_bytes = sess->image_offset(sess->nsections(), want, 1) + (size_t)want * (8 + 8 * nkeys) + queue;
			return want;
		}
	}
	size_t		small_bytes, big_bytes;
	cl_uint		small_slots = fit(64 * 1024, &small_bytes);
	cl_uint		big_slots = fit(136 * 1024, &big_bytes);
	bool		big = true;
	if (const char *v = getenv("STROM_GPUPREAGG_HASH_LDS_KB"))
		big = (atol(v) > 64);
	*p_bytes = (big ? big_bytes : small_bytes);
	return (big ? big_slots : small_slots);
}

/* a zeroed, initialised table of 'capacity' slots on the session's stream */
int
hash_table_new(strom_gpupreagg *sess, cl_uint capacity, char **p_tab, size_t *p_bytes, char *reuse)
{
	Device	   *dev = sess->dev;
	hipStream_t	stream = dev->streams[0];
	hash_layout	L = hash_table_layout(sess, capacity);
	int			errcode = 0;
	hipFunction_t fn_init = sess->prog->get_function(dev, "gpupreagg_hash_init", &errcode);
	if (!fn_init)
		return errcode;
	char   *tab = (reuse ? reuse : (char *)dev->pool.alloc(L.total));
	if (!tab)
		return StromError_OutOfMemory;
	/* the head is tiny: a synchronous copy from pageable memory is fine and keeps
	 * the source alive */
	if (hipMemsetAsync(tab, 0, L.total, stream) != hipSuccess ||
		hipStreamSynchronize(stream) != hipSuccess ||
		hipMemcpy(tab, L.head.data(), L.head.size(), hipMemcpyHostToDevice) != hipSuccess)
	{
		if (!reuse) dev->pool.release(tab);
		return StromError_HipInternal;
	}
	void   *a_tab = tab;
	void   *args[] = { &a_tab };
	unsigned grid = std::max(1u, std::min<unsigned>((capacity + 255) / 256,
													(unsigned)dev->prop.multiProcessorCount * 8));
	if (hipModuleLaunchKernel(fn_init, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
	{
		if (!reuse) dev->pool.release(tab);
		return StromError_HipInternal;
	}
	*p_tab = tab;
	*p_bytes = L.total;
	return 0;
}

/* groups the table takes before it must grow: 7/8 of the slots minus what racing threads and
 * the work-groups' final LDS flushes can still claim (never negative: a table made by an
 * import may be smaller than a fold's headroom -- the fold grows it first, see the callers) */
cl_ulong
hash_fill_limit(const strom_gpupreagg *sess, cl_ulong headroom)
{
	cl_ulong	room = (cl_ulong)sess->hash_capacity / 8 * 7;
	return room > headroom ? room - headroom : 0;
}

/* groups claimed so far (drains the session's stream) */
int
hash_table_ngroups(strom_gpupreagg *sess, cl_uint *p_ngroups, cl_uint *p_overflow)
{
	cl_uint	words[4];
	if (hipStreamSynchronize(sess->dev->streams[0]) != hipSuccess ||
		hipMemcpy(words, sess->htab, sizeof(words), hipMemcpyDeviceToHost) != hipSuccess)
		return StromError_HipInternal;
	*p_ngroups = words[2];
	sess->groups_known = std::max<cl_uint>(sess->groups_known, words[2]);
	if (p_overflow)
		*p_overflow = words[3];
	return 0;
}

/*
 * a table of at least min_capacity slots, the groups of the current one
 * re-inserted by gpupreagg_hash_rehash (drains the stream)
 */
int
hash_table_grow(strom_gpupreagg *sess, cl_ulong min_capacity)
{
	Device	   *dev = sess->dev;
	cl_ulong	capacity = sess->hash_capacity;
	cl_uint		ngroups = 0;
	int			rc = hash_table_ngroups(sess, &ngroups, nullptr);

	if (rc)
		return rc;
	while (capacity < min_capacity)
		capacity <<= 1;
	if (capacity > (1UL << 31))
		return StromError_DataStoreNoSpace;
	if (capacity == sess->hash_capacity)
		return 0;
	char	   *ntab = nullptr;
	size_t		nbytes = 0;
	rc = hash_table_new(sess, (cl_uint)capacity, &ntab, &nbytes, nullptr);
	if (rc)
		return rc;
	if (ngroups > 0)
	{
		int		errcode = 0;
		hipFunction_t fn = sess->prog->get_function(dev, "gpupreagg_hash_rehash", &errcode);
		const void *a_old = sess->htab;
		void	   *a_new = ntab;
		void	   *args[] = { &a_old, &a_new };
		unsigned	grid = std::min<unsigned>((sess->hash_capacity + 255) / 256,
											  (unsigned)dev->prop.multiProcessorCount * 8);
		if (!fn ||
			hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) != hipSuccess ||
			hipStreamSynchronize(dev->streams[0]) != hipSuccess)
		{
			dev->pool.release(ntab);
			return fn ? StromError_HipInternal : errcode;
		}
	}
	dev->pool.release(sess->htab);
	sess->htab = ntab;
	sess->htab_bytes = nbytes;
	sess->hash_capacity = (cl_uint)capacity;
	sess->groups_upper = ngroups;
	return 0;
}

/*
 * the bound of the table's integer sums restarts from what the table really holds
 * (gpupreagg_hash_sum_refresh; queued on the session's stream, sess->lock held)
 */
int
hash_sum_refresh(strom_gpupreagg *sess)
{
	Device	   *dev = sess->dev;
	int			errcode = 0;
	if (sess->nintsums == 0 || !sess->htab)
		return 0;
	hipFunction_t fn = sess->prog->get_function(dev, "gpupreagg_hash_sum_refresh", &errcode);
	if (!fn)
		return errcode;
	void	   *a_tab = sess->htab;
	void	   *args[] = { &a_tab };
	unsigned	grid = std::max(1u, std::min<unsigned>((sess->hash_capacity + 255) / 256,
													(unsigned)dev->prop.multiProcessorCount * 8));
	/* sum_bound[2] sits 32 bytes into the head (struct gpupreagg_hash_head) */
	if (hipMemsetAsync(sess->htab + 32, 0, 16, dev->streams[0]) != hipSuccess ||
		hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) != hipSuccess)
		return StromError_HipInternal;
	return 0;
}

/*
 * the largest |integer sum| the table holds, plus one (0: no integer sums, or an empty table):
 * measured by hash_sum_refresh, read back.  sess->lock held.
 */
int
hash_sum_measured_bound(strom_gpupreagg *sess, cl_ulong *p_bound)
{
	*p_bound = 0;
	if (sess->nintsums == 0 || !sess->htab)
		return 0;
	int		rc = hash_sum_refresh(sess);
	if (rc)
		return rc;
	if (hipStreamSynchronize(sess->dev->streams[0]) != hipSuccess ||
		hipMemcpy(p_bound, sess->htab + 32, sizeof(cl_ulong), hipMemcpyDeviceToHost) != hipSuccess)
		return StromError_HipInternal;
	return 0;
}

void gpupreagg_launch_hashed(strom_task_impl *task, preagg_request req, bool second = false);
/*
 * The exact fold of a hashed session ("integer sums never wrap", strom_gpupreagg.h).  The running
 * bound of a hashed table -- rows x largest input magnitude, over all groups at once -- can fail
 * although no group comes near the edge (2e6 rows of 1e13 in a thousand groups).  Such a chunk is
 * folded into a SCRATCH session of the same targets whose program is built with GPUPREAGG_CHECKED
 * (every LDS and table addition returns what it was added to and is checked, the reference's
 * CHECK_OVERFLOW_INT on every accumulate, opencl_gpupreagg.h:142-143): an addition that leaves int8
 * inside the chunk is CpuReCheck and the session's table has not been touched.  Then the scratch
 * groups join the table: a read-only pass checks every group that exists on both sides
 * (gpupreagg_hash_import_verify), and only when all fit are they imported.  Exact, or CpuReCheck
 * with the table as it was.  Rare and slower (a second table, one more pass): the price is paid by
 * the chunks that need it.  The scratch session lives for the request (strom_task_impl::at_complete).
 */
void
gpupreagg_hashed_exact(strom_task_impl *task, preagg_request req)
{
	int		errcode = 0;
	strom_gpupreagg *child = gpupreagg_exact_child(req.sess, &errcode);
	if (!child)
	{
		task_fail(task, errcode ? errcode : StromError_OutOfMemory);
		return;
	}
	task->at_complete.push_back([child]() { strom_gpupreagg_release(child); });
	req.sess->checked_folds++;
	preagg_request creq = req;
	creq.exact_parent = req.sess;
	creq.sess = child;
	program_run_or_park(child->prog, [task, creq]() { gpupreagg_launch_hashed(task, creq, false); });
}

void
gpupreagg_hashed_exact_merge(strom_task_impl *task, preagg_request req)
{
	strom_gpupreagg *child = req.sess, *parent = req.exact_parent;
	char	   *d_recs = nullptr;
	cl_uint		count = 0;
	size_t		reclen = 0;
	int			rc = gpupreagg_hash_export_device(child, &d_recs, &count, &reclen);
	if (rc == 0 && count > 0)
		rc = gpupreagg_hash_import_device(parent, d_recs, count, 1, &count, ~0u, GPUPREAGG_IMPORT_EXACT);
	gpupreagg_hash_release(child, d_recs);
	if (rc != 0 && rc != StromError_CpuReCheck)
	{
		task_fail(task, rc);
		return;
	}
	task->finish = [rc](strom_task_impl *t) { t->errcode = rc; };
	task_enqueue(task);
}

void
gpupreagg_launch_hashed(strom_task_impl *task, preagg_request req, bool second)
{
	strom_gpupreagg *sess = req.sess;
	Device	   *dev = task->dev;
	Program	   *prog = sess->prog;
	int			errcode = 0;
	std::lock_guard<std::mutex> g(sess->lock);

	(void)hipSetDevice(dev->hip_id);
	if (prog->state != STROM_DEVPROG_READY)
	{
		task_fail(task, StromError_ProgramBuildFailure);
		return;
	}
	task->pfm.time_kern_build = (cl_ulong)prog->build_usec;
	task->stream = dev->streams[0];
	hipFunction_t fn_check = prog->get_function(dev, "gpupreagg_hash_check", &errcode);
	hipFunction_t fn_fold = fn_check ? prog->get_function(dev, "gpupreagg_hash_fold", &errcode) : nullptr;
	if (!fn_check || !fn_fold)
	{
		task_fail(task, errcode);
		return;
	}
	/* hash roles, the role map and the partition plan scan a COLUMN chunk's arrays as they are: not
	 * for a program with text / character(n) variables (their column holds offsets that each row
	 * turns into an address -- strom_kvars_from_column, the one-role walk does it) */
	const bool	column_streams = (req.format == KDS_FORMAT_COLUMN && !(prog->extra_flags & DEVTYPE_IS_VARLENA));
	/*
	 * geometry: two work-groups per CU (64 KB of LDS each).  Slots can be
	 * claimed past the fill limit by threads that raced through the limit test
	 * (one each at most) and by the work-groups' final LDS flushes: that much
	 * headroom, and an eighth of the table on top, always stays free.
	 */
	size_t		lds_bytes = 0;
	cl_uint		lds_slots = hash_lds_slots(sess, &lds_bytes);
	unsigned	block = (unsigned)sess->block;
	unsigned	fold_grid = (unsigned)dev->prop.multiProcessorCount * (lds_bytes <= 72 * 1024 ? 2 : 1);
	cl_ulong	headroom = (cl_ulong)fold_grid * block + (cl_ulong)fold_grid * lds_slots;
	if (!sess->htab)
	{
		cl_ulong	capacity = sess->hash_capacity;
		while (capacity < 4 * headroom)
			capacity <<= 1;
		sess->hash_capacity = (cl_uint)capacity;
		int rc = hash_table_new(sess, sess->hash_capacity, &sess->htab, &sess->htab_bytes, nullptr);
		if (rc)
		{
			task_fail(task, rc);
			return;
		}
	}
	else if ((cl_ulong)sess->hash_capacity < 4 * headroom)
	{
		/* a table made by a merge / import before the first fold (at the session's initial
		 * size): a fold needs its headroom of free slots whatever made the table */
		int rc = hash_table_grow(sess, 4 * headroom);
		if (rc)
		{
			task_fail(task, rc);
			return;
		}
	}
	size_t	kg_len = STROMALIGN(offsetof(kern_gpupreagg, kparams) + sess->kparams.size());
	char   *stage = dev->pinned.alloc();
	char   *d_kg = (char *)dev->pool.alloc(kg_len);
	if (!stage || !d_kg || kg_len + 64 > PinnedPool::BLOCK)
	{
		if (stage) dev->pinned.release(stage);
		if (d_kg) dev->pool.release(d_kg);
		task_fail(task, StromError_OutOfMemory);
		return;
	}
	task->pinned_blocks.push_back(stage);
	task->main_devptr = d_kg;
	memset(stage, 0, kg_len);
	memcpy(stage + offsetof(kern_gpupreagg, kparams), sess->kparams.data(), sess->kparams.size());
	/* integer sums never wrap (strom_gpupreagg.h): rows of the request, what is known about
	 * the inputs before the check pass measures the rest */
	((kern_gpupreagg *)stage)->sortbuf_len = (cl_int)req.nrows;
	memcpy(((kern_gpupreagg *)stage)->__padding, &sess->static_magbits, sizeof(cl_uint));
	/* the fold's turn (gpupreagg_hash_sum_account): parity; 4 = second attempt, after the
	 * bound was measured; relaunches for deferred rows add 2 */
	/* (bit 2 -- "a failed proof is final: CpuReCheck" -- is no longer asked for: the host sends an
	 * unproven chunk to the exact fold instead, gpupreagg_hashed_exact) */
	cl_uint		sum_turn = (sess->sum_turn++ & 1u) | ((second && getenv("STROM_GPUPREAGG_HASH_NO_EXACT")) ? 4u : 0u);
	if (second)
	{
		int rc = hash_sum_refresh(sess);
		if (rc)
		{
			task_fail(task, rc);
			return;
		}
	}

	task_event(task);									/* ev[0] */
	REQ_CHECK(hipMemcpyAsync(d_kg, stage, kg_len, hipMemcpyHostToDevice, task->stream),
			  "send kern_gpupreagg");
	task->pfm.num_dma_send++;
	task->pfm.bytes_dma_send += kg_len;
	const void *d_kds;
	if (req.kds_dev)
		d_kds = req.kds_dev->devptr;
	else
	{
		size_t	kds_len = req.kds->length;
		if (req.kds->format == KDS_FORMAT_ROW)
			kds_len = KERN_DATA_STORE_ROWBLOCK_OFFSET(req.kds) + (size_t)BLCKSZ * req.kds->nblocks;
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
	if (!req.rowmap_dev && req.krowmap)
	{
		size_t	len = offsetof(kern_row_map, rindex) + sizeof(cl_int) * (size_t)req.krowmap->nvalids;
		void   *p = dev->pool.alloc(len);
		if (!p)
		{
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		task->devbufs.push_back(p);
		REQ_CHECK(hipMemcpyAsync(p, req.krowmap, len, hipMemcpyHostToDevice, task->stream),
				  "send kern_row_map");
		d_rowmap = p;
	}
	task_event(task);									/* ev[1] */
	/*
	 * More groups than the hash roles' LDS tables take together (or than a few roles, each a
	 * scan of the chunk, are worth): the chunk is ordered by hash partition first and folded
	 * unit by unit (strom_gpupreagg.h: gpupreagg_hash_check_parts ... _fold_parts).  1e8 rows,
	 * 1e5 / 1e6 groups: 23 / 42 ms through the global table; profiles/r02_hashed_partitions.txt
	 */
	cl_ulong	parts_min = 10000;
	if (const char *v = getenv("STROM_GPUPREAGG_HASH_PARTS_MIN"))
		parts_min = strtoul(v, nullptr, 10);
	/* nothing known about the group count yet (no hint, first chunk; every chunk of the stateless
	 * per-chunk message): a large chunk takes the plan whose cost does not depend on it -- 2.6 ms
	 * per 1e8 rows whatever the count, against 0.7 ... 14 ms through the LDS table and the global one */
	bool		unknown = (sess->groups_known == 0 && req.nrows >= (4u << 20));
	bool		use_parts = (req.nrows > 0 && ((cl_ulong)sess->groups_known >= parts_min || (unknown && parts_min > 0)) &&
							 !getenv("STROM_GPUPREAGG_HASH_NO_PARTS"));
	if (use_parts)
	{
		hipFunction_t fn_pcheck = prog->get_function(dev, "gpupreagg_hash_check_parts", &errcode);
		hipFunction_t fn_plan = fn_pcheck ? prog->get_function(dev, "gpupreagg_hash_part_plan", &errcode) : nullptr;
		hipFunction_t fn_scatter = fn_plan ? prog->get_function(dev, "gpupreagg_hash_scatter", &errcode) : nullptr;
		hipFunction_t fn_scatter_lds = fn_scatter ? prog->get_function(dev, "gpupreagg_hash_scatter_lds", &errcode) : nullptr;
		hipFunction_t fn_units = fn_scatter_lds ? prog->get_function(dev, "gpupreagg_hash_fold_parts", &errcode) : nullptr;
		if (!fn_units)
		{
			task_fail(task, errcode);
			return;
		}
		cl_uint		nrows = req.nrows;
		/*
		 * partitions: as few as keep a partition's groups within 5/8 of a unit's LDS table --
		 * the fewer, the longer the runs the scatter writes (a tile of ~4096 rows leaves in
		 * runs of 4096 / nparts records)
		 */
		cl_uint		nparts = 256;
		{
			size_t	b;
			cl_ulong per_unit = (cl_ulong)hash_lds_slots(sess, &b, true) * 5 / 8;
			while (nparts < 4096 && (cl_ulong)sess->groups_known > (cl_ulong)nparts * per_unit)
				nparts <<= 1;
		}
		if (const char *v = getenv("STROM_GPUPREAGG_HASH_PARTS"))
		{
			int want = atoi(v);
			if (want >= 64 && want <= 4096 && (want & (want - 1)) == 0)
				nparts = (cl_uint)want;
		}
		/* a unit is a whole partition unless the partition is four times the average (one
		 * heavy key must not be one work-group's job): every unit ends in a flush of its
		 * groups to the global table, the fewer the better */
		cl_uint		unit_rows = std::max<cl_uint>(32768, (cl_uint)std::min<cl_ulong>(1u << 30, 4 * (cl_ulong)nrows / nparts));
		if (const char *v = getenv("STROM_GPUPREAGG_HASH_UNIT_ROWS"))
			unit_rows = std::max(1024, atoi(v));
		cl_uint		log2cap = 0, log2parts = 0;
		while ((1UL << (log2cap + 1)) <= sess->hash_capacity)
			log2cap++;
		while ((1u << (log2parts + 1)) <= nparts)
			log2parts++;
		size_t		nvals = 0;
		for (int resno : sess->agg_resno)
			nvals += (sess->targets[resno].kind != STROM_PREAGG_NROWS ? 1 : 0);
		size_t		reclen = 8 * (1 + sess->key_resno.size() + nvals);
		cl_uint		max_units = nparts + nrows / unit_rows + 1;
		struct part_ctl { cl_uint nparts, pshift, unit_rows, nunits, nrecords, deferred, max_units, reclen; } ctl_img;
		ctl_img = part_ctl{ nparts, (log2cap > log2parts ? log2cap - log2parts : 0), unit_rows, 0, 0, 0, max_units, (cl_uint)reclen };
		/* ctl | hist[P] | cursor[P] | units[2 * max_units] | two redo lists of max_units */
		size_t		ctl_len = sizeof(part_ctl) + sizeof(cl_uint) * ((size_t)2 * nparts + (size_t)4 * max_units);
		char	   *d_ctl = (char *)dev->pool.alloc(ctl_len);
		char	   *d_partmap = (char *)dev->pool.alloc(STROMALIGN((size_t)nrows * sizeof(cl_ushort)));
		char	   *d_records = (char *)dev->pool.alloc((size_t)nrows * reclen);
		if (d_ctl) task->devbufs.push_back(d_ctl);
		if (d_partmap) task->devbufs.push_back(d_partmap);
		if (d_records) task->devbufs.push_back(d_records);
		if (!d_ctl || !d_partmap || !d_records || kg_len + 128 + sizeof(part_ctl) > PinnedPool::BLOCK)
		{
			task_fail(task, StromError_OutOfMemory);
			return;
		}
		char	   *stage_ctl = stage + kg_len + 64;
		memcpy(stage_ctl, &ctl_img, sizeof(ctl_img));
		REQ_CHECK(hipMemsetAsync(d_ctl, 0, sizeof(part_ctl) + sizeof(cl_uint) * nparts, task->stream),
				  "reset partition counts");
		REQ_CHECK(hipMemcpyAsync(d_ctl, stage_ctl, sizeof(part_ctl), hipMemcpyHostToDevice, task->stream),
				  "send partition plan");
		void	   *a_kg = d_kg;
		const void *a_kds = d_kds;
		const void *a_toast = nullptr;
		const void *a_map = d_rowmap;
		void	   *a_ctl = d_ctl;
		void	   *a_hist = d_ctl + sizeof(part_ctl);
		void	   *a_cursor = d_ctl + sizeof(part_ctl) + sizeof(cl_uint) * nparts;
		void	   *a_units = d_ctl + sizeof(part_ctl) + sizeof(cl_uint) * 2 * nparts;
		void	   *a_partmap = d_partmap;
		void	   *a_records = d_records;
		unsigned	ncus = (unsigned)dev->prop.multiProcessorCount;
		{
			void	   *args[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_partmap, &a_hist, &a_ctl };
			/* (1024 threads, two work-groups per CU: 2.37 against 2.48 ms for the whole plan with 256 x 4,
			 * the other combinations in between -- scripts/hashed_check_sweep.sh) */
			unsigned	cblock = 1024, cper = 2;
			if (const char *v = getenv("STROM_GPUPREAGG_HASH_CHECK_BLOCK"))
			{
				int want = atoi(v);
				if (want == 256 || want == 512 || want == 1024)
					cblock = (unsigned)want;
			}
			if (const char *v = getenv("STROM_GPUPREAGG_HASH_CHECK_PER_CU"))
				cper = (unsigned)std::max(1, atoi(v));
			unsigned	grid = std::max(1u, std::min<unsigned>((nrows + 2 * cblock - 1) / (2 * cblock), ncus * cper));
			REQ_CHECK(hipModuleLaunchKernel(fn_pcheck, grid, 1, 1, cblock, 1, 1, 0, task->stream, args, nullptr),
					  "launch gpupreagg hash check (partitions)");
		}
		{
			void	   *args[] = { &a_hist, &a_cursor, &a_units, &a_ctl };
			REQ_CHECK(hipModuleLaunchKernel(fn_plan, 1, 1, 1, 256, 1, 1, 0, task->stream, args, nullptr),
					  "launch gpupreagg partition plan");
		}
		/* the scatter goes through LDS (records leave in runs) when a tile of at least one row
		 * per thread fits next to the partitions' counters */
		size_t		stage_fixed = 12 * (size_t)nparts + 128;
		size_t		stage_budget = 159 * 1024;
		cl_uint		lds_rows = 0;
		unsigned	sblock = block;
		if (const char *v = getenv("STROM_GPUPREAGG_HASH_SCATTER_BLOCK"))
		{
			int want = atoi(v);
			if ((want == 256 || want == 512 || want == 1024) && (unsigned)want <= block)
				sblock = (unsigned)want;
		}
		if (nparts <= 4 * sblock && stage_fixed < stage_budget && !getenv("STROM_GPUPREAGG_HASH_NO_LDS_SCATTER"))
			lds_rows = (cl_uint)std::min<size_t>(4, (stage_budget - stage_fixed) / ((reclen + 2) * sblock));
		if (const char *v = getenv("STROM_GPUPREAGG_HASH_SCATTER_ROWS"))
			lds_rows = std::min<cl_uint>(lds_rows, (cl_uint)std::max(1, atoi(v)));
		if (lds_rows > 0)
		{
			size_t		stage_bytes = stage_fixed + (size_t)lds_rows * sblock * (reclen + 2);
			void	   *args[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_partmap, &a_cursor, &a_records, &a_ctl, &lds_rows };
			size_t		tile = (size_t)sblock * lds_rows;
			unsigned	grid = (unsigned)std::max<size_t>(1, std::min<size_t>((nrows + tile - 1) / tile,
																			   (size_t)ncus * (stage_bytes <= 72 * 1024 ? 2 : 1)));
			REQ_CHECK(hipModuleLaunchKernel(fn_scatter_lds, grid, 1, 1, sblock, 1, 1, (unsigned)stage_bytes,
											task->stream, args, nullptr),
					  "launch gpupreagg hash scatter (LDS)");
		}
		else
		{
			void	   *args[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_partmap, &a_cursor, &a_records, &a_ctl };
			size_t		tile = (size_t)block * 32;
			unsigned	grid = (unsigned)std::max<size_t>(1, std::min<size_t>((nrows + tile - 1) / tile, (size_t)ncus * 2));
			REQ_CHECK(hipModuleLaunchKernel(fn_scatter, grid, 1, 1, block, 1, 1, 0, task->stream, args, nullptr),
					  "launch gpupreagg hash scatter");
		}
		task->pfm.num_kern_exec += 3;
		task->pfm.num_kern_prep++;				/* (reported: this request took the partition plan) */
		/*
		 * fold, unit by unit.  While even nrows new groups fit under the fill limit the
		 * task stays asynchronous; otherwise a unit with a new group beyond the limit comes
		 * back on the redo list with nothing merged: the table grows and the list is run.
		 */
		size_t		lds_bytes = 0;				/* (the units' table, not the roles') */
		cl_uint		lds_slots = hash_lds_slots(sess, &lds_bytes, true);
		unsigned	fold_grid = ncus * (lds_bytes <= 79 * 1024 ? 2 : 1);
		/*
		 * the unit's LDS table leaves room for one work-group per CU: its size in threads is the
		 * CU's whole occupancy, so the full GPUPREAGG_BLOCK (1024: four waves per SIMD).  Smaller
		 * work-groups are slower all the way: 256 / 512 / 1024 threads fold 1e8 rows at 1e6 groups
		 * in 7.0 / 4.1 / 3.2 ms, and the LDS scatter likewise (profiles/r03_hashed_block_sweep.txt).
		 */
		unsigned	fold_block = block;
		if (const char *v = getenv("STROM_GPUPREAGG_HASH_FOLD_BLOCK"))
		{
			int want = atoi(v);
			if ((want == 256 || want == 512 || want == 1024) && (unsigned)want <= block)
				fold_block = (unsigned)want;
		}
		char	   *d_lists = d_ctl + sizeof(part_ctl) + sizeof(cl_uint) * ((size_t)2 * nparts + (size_t)2 * max_units);
		void	   *a_todo = nullptr;
		cl_uint		ntodo = 0;
		for (int turn = 0;; turn++)
		{
			cl_ulong	fill_limit = hash_fill_limit(sess, headroom);
			if (sess->groups_upper + nrows > fill_limit)
			{
				cl_uint	ngroups = 0;
				int		rc = hash_table_ngroups(sess, &ngroups, nullptr);
				if (rc == 0)
				{
					sess->groups_upper = ngroups;
					if (turn > 0 || (cl_ulong)ngroups * 2 > fill_limit)
						rc = hash_table_grow(sess, (cl_ulong)sess->hash_capacity * 4);
				}
				if (rc)
				{
					task_fail(task, rc);
					return;
				}
				fill_limit = hash_fill_limit(sess, headroom);
			}
			bool		may_defer = (sess->groups_upper + nrows > fill_limit);
			cl_uint		claim_limit = (may_defer ? (cl_uint)fill_limit : ~0u);
			void	   *a_tab = sess->htab;
			void	   *a_redo = d_lists + sizeof(cl_uint) * (size_t)max_units * (turn & 1);
			if (may_defer)
				REQ_CHECK(hipMemsetAsync(d_ctl + offsetof(part_ctl, deferred), 0, sizeof(cl_uint), task->stream),
						  "reset the redo list");
			cl_uint		a_turn = sum_turn | (turn > 0 ? 2u : 0u);
			void	   *args[] = { &a_kg, &a_tab, &claim_limit, &a_ctl, &a_units, &a_records, &lds_slots,
								   &a_todo, &ntodo, &a_redo, &a_turn };
			REQ_CHECK(hipModuleLaunchKernel(fn_units, fold_grid, 1, 1, fold_block, 1, 1, (unsigned)lds_bytes,
											task->stream, args, nullptr),
					  "launch gpupreagg hash fold (partitions)");
			task->pfm.num_kern_exec++;
			if (!may_defer)
			{
				sess->groups_upper += nrows;
				break;
			}
			cl_uint		nredo = 0;
			REQ_CHECK(hipMemcpyAsync(&nredo, d_ctl + offsetof(part_ctl, deferred), sizeof(cl_uint),
									 hipMemcpyDeviceToHost, task->stream),
					  "recv redo list");
			REQ_CHECK(hipStreamSynchronize(task->stream), "fold (partitions)");
			sess->groups_upper = std::min<cl_ulong>(sess->groups_upper + nrows, sess->hash_capacity);
			if (nredo == 0)
				break;
			a_todo = a_redo;
			ntodo = nredo;
		}
	}
	else
	{
		void	   *a_kg = d_kg;
		const void *a_kds = d_kds;
		const void *a_toast = nullptr;
		const void *a_map = d_rowmap;
		cl_uint		nrows = req.nrows;
		unsigned	maxgrid = (unsigned)dev->prop.multiProcessorCount * 8;
		cl_uint		no_limit = ~0u;
		void	   *a_nodefer = nullptr;
		/*
		 * role map: when the fold will run with hash roles (COLUMN chunk, no row map, more
		 * groups than one LDS table takes) the check pass leaves one byte per row -- the
		 * row's role -- and the roles' scans read that instead of the qual's and the keys'
		 * columns.  Padded with ROLE_NONE to whole scan tiles (16 rows per lane of a
		 * fold work-group).
		 */
		void	   *a_rolemap = nullptr;
		{
			cl_ulong	per_role0 = (cl_ulong)lds_slots * 5 / 8;
			if (column_streams && !d_rowmap && nrows > 0 &&
				(cl_ulong)sess->groups_known > per_role0 && !getenv("STROM_GPUPREAGG_HASH_NO_ROLEMAP"))
			{
				size_t	tile = (size_t)block * 16;
				size_t	len = (((size_t)nrows + tile - 1) / tile) * tile;
				a_rolemap = dev->pool.alloc(len);
				if (a_rolemap)
				{
					task->devbufs.push_back(a_rolemap);
					REQ_CHECK(hipMemsetAsync(a_rolemap, 0xff, len, task->stream), "pad the role map");
				}
			}
		}
		/* pass 1: errors only -- a chunk with a CpuReCheck row is not folded at all */
		{
			void	   *a_tab = sess->htab;
			cl_uint		one_role = 1;
			void	   *args[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_tab, &no_limit, &a_nodefer, &lds_slots,
								   &one_role, &a_rolemap };
			/* (the kernels deal tiles to 8 XCDs: grids are multiples of 8) */
			unsigned	grid = std::max(8u, std::min<unsigned>((nrows + 255) / 256 + 7, maxgrid) & ~7u);
			REQ_CHECK(hipModuleLaunchKernel(fn_check, grid, 1, 1, 256, 1, 1, 0, task->stream, args, nullptr),
					  "launch gpupreagg hash check");
			task->pfm.num_kern_exec++;
		}
		/*
		 * pass 2: fold.  While even nrows new groups fit under the fill limit
		 * nothing can be deferred and the task stays asynchronous; otherwise
		 * rows that need a new group beyond the limit come back as a row map,
		 * the table grows, and they are folded again.
		 */
		cl_uint		todo = nrows;
		void	   *d_defer[2] = { nullptr, nullptr };
		int			turn = 0;
		while (todo > 0)
		{
			cl_ulong	fill_limit = hash_fill_limit(sess, headroom);
			if (sess->groups_upper + todo > fill_limit)
			{
				/* the bound is stale or the table is small: look */
				cl_uint	ngroups = 0;
				int		rc = hash_table_ngroups(sess, &ngroups, nullptr);
				if (rc == 0)
				{
					sess->groups_upper = ngroups;
					if (turn > 0 || (cl_ulong)ngroups * 2 > fill_limit)
						rc = hash_table_grow(sess, (cl_ulong)sess->hash_capacity * 4);	/* geometric: the deferred
																						 * rows say little about
																						 * the groups among them */
				}
				if (rc)
				{
					task_fail(task, rc);
					return;
				}
				fill_limit = hash_fill_limit(sess, headroom);
			}
			bool		may_defer = (sess->groups_upper + todo > fill_limit);
			cl_uint		claim_limit = (may_defer ? (cl_uint)fill_limit : no_limit);
			void	   *a_tab = sess->htab;
			void	   *a_defer = nullptr;
			if (may_defer)
			{
				size_t	len = offsetof(kern_row_map, rindex) + sizeof(cl_int) * (size_t)todo;
				if (!d_defer[turn & 1])
				{
					d_defer[turn & 1] = dev->pool.alloc(len);
					if (!d_defer[turn & 1])
					{
						task_fail(task, StromError_OutOfMemory);
						return;
					}
					task->devbufs.push_back(d_defer[turn & 1]);
				}
				a_defer = d_defer[turn & 1];
				REQ_CHECK(hipMemsetAsync(a_defer, 0, offsetof(kern_row_map, rindex), task->stream),
						  "reset deferred rows");
			}
			/*
			 * roles: as many as it takes for a role's share of the groups seen so
			 * far to fill 5/8 of its LDS table (64 probes find room at that fill).
			 * Every role scans every row's key columns: ~0.2 ms per 1e8 rows and
			 * role against 23-47 ms through the global table, so up to 64 roles
			 * pay (measured, 1e8 rows: 1000 groups 11.5 -> 1.6 ms with 2 roles, 1e4
			 * groups 26.7 -> 5.1 ms with 16, 3e4 groups 13.2 ms with 64)
			 */
			cl_uint		nroles = 1;
			cl_ulong	per_role = (cl_ulong)lds_slots * 5 / 8;
			cl_ulong	known = sess->groups_known;
			if (const char *v = getenv("STROM_GPUPREAGG_HASH_FILL"))
				per_role = std::max<cl_ulong>(1, (cl_ulong)lds_slots * (cl_ulong)atoi(v) / 100);
			if (column_streams && known > per_role && known <= per_role * 64)
			{
				while (nroles < 64 && (cl_ulong)nroles * per_role < known)
					nroles <<= 1;
			}
			if (const char *v = getenv("STROM_GPUPREAGG_HASH_ROLES"))
			{
				/* 1, 2, 4 ... 64; anything else would break the tile walk */
				int want = atoi(v);
				if (column_streams && want >= 1 && want <= 64 && (want & (want - 1)) == 0)
					nroles = (cl_uint)want;
			}
			/* (deferred rows of a later turn are a row map: the scan over the columns) */
			void	   *a_rm = (turn == 0 && nroles > 1 ? a_rolemap : nullptr);
			cl_uint		a_turn = sum_turn | (turn > 0 ? 2u : 0u);
			void	   *args[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_tab, &claim_limit, &a_defer, &lds_slots,
								   &nroles, &a_rm, &a_turn };
			unsigned	unit = 8 * nroles;
			unsigned	grid = std::min<unsigned>(((todo + block - 1) / block + unit - 1) / unit * unit, fold_grid);
			grid = std::max(unit, grid / unit * unit);
			REQ_CHECK(hipModuleLaunchKernel(fn_fold, grid, 1, 1, block, 1, 1, (unsigned)lds_bytes,
											task->stream, args, nullptr),
					  "launch gpupreagg hash fold");
			task->pfm.num_kern_exec++;
			if (!may_defer)
			{
				sess->groups_upper += todo;
				break;
			}
			cl_int		ndeferred = 0;
			REQ_CHECK(hipMemcpyAsync(&ndeferred, a_defer, sizeof(cl_int), hipMemcpyDeviceToHost, task->stream),
					  "recv deferred rows");
			REQ_CHECK(hipStreamSynchronize(task->stream), "fold");
			sess->groups_upper = std::min<cl_ulong>(sess->groups_upper + todo, sess->hash_capacity);
			todo = (cl_uint)ndeferred;
			a_map = a_defer;			/* the deferred rows are the next launch's row map */
			turn++;
		}
	}
	task_event(task);									/* ev[2] */
	char   *stage_status = stage + kg_len;
	REQ_CHECK(hipMemcpyAsync(stage_status, d_kg + offsetof(kern_gpupreagg, status), sizeof(cl_int),
							 hipMemcpyDeviceToHost, task->stream),
			  "recv status");
	REQ_CHECK(hipMemcpyAsync(stage_status + 16, sess->htab, 16, hipMemcpyDeviceToHost, task->stream),
			  "recv table head");
	task->pfm.num_dma_recv += 2;
	task->pfm.bytes_dma_recv += sizeof(cl_int) + 16;
	task_event(task);									/* ev[3] */
	task->finish = [stage_status, sess, req, second](strom_task_impl *t)
	{
		cl_int	status;
		cl_uint	words[4];
		memcpy(&status, stage_status, sizeof(status));
		memcpy(words, stage_status + 16, sizeof(words));
		if (status == StromError_SumRangeUnproven && !second)
		{
			/* the running bound of the integer sums reached 2^63 and nothing was folded:
			 * once more, with the bound measured from the table (hash_sum_refresh) */
			t->retry = [t, req]() { gpupreagg_launch_hashed(t, req, true); };
			return;
		}
		if (status == StromError_SumRangeUnproven && !req.exact_parent && !getenv("STROM_GPUPREAGG_HASH_NO_EXACT"))
		{
			/* rows x largest magnitude still reaches 2^63 -- a bound over ALL groups at once,
			 * which says little about any one of them: the chunk is folded exactly instead */
			t->retry = [t, req]() { gpupreagg_hashed_exact(t, req); };
			return;
		}
		if (status == StromError_Success && req.exact_parent)
		{
			/* the scratch session holds the chunk's groups, every addition checked: they
			 * join the parent's under a per-group range check */
			t->retry = [t, req]() { gpupreagg_hashed_exact_merge(t, req); };
			return;
		}
		if (status == StromError_SumRangeUnproven)
			status = StromError_CpuReCheck;
		if (status == StromError_Success && words[3] != 0)
			status = StromError_DataStoreNoSpace;	/* cannot happen: headroom is kept for every claim */
		sess->groups_known = std::max<cl_uint>(sess->groups_known, words[2]);
		t->errcode = status;
	};
	task_enqueue(task);
}

}	/* namespace */

/* ================================================================== *
 * C ABI
 * ================================================================== */
static strom_gpupreagg *
gpupreagg_session_new(strom_devprog_key key,
					  const strom_preagg_target *targets, int ntargets,
					  const kern_parambuf *kparams,
					  const strom_preagg_domain *domain, bool hashed,
					  int dindex, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	Program *prog = lookup_program(key);
	Device *dev = get_device(dindex);
	if (!prog || !dev || !targets || ntargets < 1 || !kparams)
	{
		*p_errcode = (!dev ? StromError_ServerNotReady : StromError_BadRequestMessage);
		return nullptr;
	}
	strom_gpupreagg *sess = new strom_gpupreagg();
	sess->key = key;
	sess->prog = prog;
	sess->dev = dev;
	sess->targets.assign(targets, targets + ntargets);
	for (int i = 0; i < ntargets; i++)
	{
		if (targets[i].kind == STROM_PREAGG_KEY)
		{
			if (!hashed && (type_is_float(targets[i].type_oid) || targets[i].type_oid == STROM_NUMERICOID))
			{
				/* dense ids need integer-like keys */
				*p_errcode = StromError_DataStoreOutOfRange;
				delete sess;
				return nullptr;
			}
			sess->key_resno.push_back(i);
		}
		else
			sess->agg_resno.push_back(i);
	}
	sess->kparams.assign((const char *)kparams, (const char *)kparams + kparams->length);
	{
		/* what the code generator says about packed accumulators (codegen_preagg.cpp) */
		const char *src = prog->source.c_str();
		const char *lst = strstr(src, "#define GPUPREAGG_PACK_LIST(X)");
		sess->numeric_aggs = (strstr(src, "#define GPUPREAGG_NUMERIC_AGGS 1") != nullptr);
		sess->packable = (strstr(src, "#define GPUPREAGG_PACKABLE 1") != nullptr && lst != nullptr);
		if (lst)
		{
			/* (read for every program: the source column of a plain-column sum also bounds it,
			 * see gpupreagg_launch) */
			bool		whole = true;
			const char *eol = strchr(lst, '\n');
			for (const char *p = strstr(lst, " X("); p && (!eol || p < eol); p = strstr(p + 1, " X("))
			{
				int		aidx = -1, kind = 0, attno = 0;
				if (sscanf(p, " X(%d,%d,%d)", &aidx, &kind, &attno) != 3 || aidx != (int)sess->pack_kind.size())
				{
					whole = false;
					break;
				}
				sess->pack_kind.push_back(kind);
				sess->pack_attno.push_back(attno);
			}
			if (!whole || sess->pack_kind.size() != sess->agg_resno.size())
			{
				sess->pack_kind.assign(sess->agg_resno.size(), 0);
				sess->pack_attno.assign(sess->agg_resno.size(), 0);
				sess->packable = false;
			}
			if (sess->agg_resno.size() > 32)
				sess->packable = false;
		}
		else
		{
			sess->pack_kind.assign(sess->agg_resno.size(), 0);
			sess->pack_attno.assign(sess->agg_resno.size(), 0);
		}
		/* integer sums: which aggregates, and what the code generator knows about their inputs */
		for (size_t a = 0; a < sess->agg_resno.size(); a++)
		{
			const strom_preagg_target &t = sess->targets[sess->agg_resno[a]];
			bool	intsum = (t.kind == STROM_PREAGG_PSUM &&
							  (t.type_oid == STROM_INT8OID || (t.type_oid == STROM_NUMERICOID && t.scale >= 0)));
			char	name[64];
			int		bits = 64;
			snprintf(name, sizeof(name), "#define GPUPREAGG_SUMBITS_%zu ", a);
			const char *def = strstr(src, name);
			if (def)
				bits = atoi(def + strlen(name));
			if (intsum != (bits != 0))
			{
				/* the caller's targets do not describe this program */
				*p_errcode = StromError_BadRequestMessage;
				delete sess;
				return nullptr;
			}
			sess->sumbits.push_back(bits);
			{
				std::string	formula;
				snprintf(name, sizeof(name), "#define GPUPREAGG_SUMBOUND_%zu \"", a);
				const char *fdef = strstr(src, name);
				if (fdef)
				{
					const char *q = strchr(fdef + strlen(name), '"');
					if (q)
						formula.assign(fdef + strlen(name), q);
				}
				sess->sumbound.push_back(formula);
			}
			sess->intsum_of.push_back(intsum ? sess->nintsums++ : -1);
			if (bits >= 1 && bits <= 63)
				sess->static_magbits = std::max(sess->static_magbits, (cl_uint)bits);
			if (intsum && bits >= 64)
				sess->sums_measured = true;
		}
	}
	if (const char *v = getenv("STROM_GPUPREAGG_BLOCK"))
		sess->block = atoi(v);
	if (const char *v = getenv("STROM_GPUPREAGG_QUADS"))
		sess->quads = atoi(v);
	strom_retain_devprog_key(key);
	sess->hashed = hashed;
	if (hashed && (sess->key_resno.size() > 31 || sess->agg_resno.size() > 30 || sess->numeric_aggs))
	{
		*p_errcode = StromError_BadRequestMessage;
		strom_gpupreagg_release(sess);
		return nullptr;
	}
	if (domain)
	{
		int rc = setup_geometry(sess, domain);
		if (rc == 0)
			rc = alloc_session_buffers(sess);
		if (rc != 0)
		{
			*p_errcode = rc;
			strom_gpupreagg_release(sess);
			return nullptr;
		}
	}
	return sess;
	STROM_ABI_CATCH(nullptr, p_errcode)
}

extern "C" strom_gpupreagg *
strom_gpupreagg_create(strom_devprog_key key,
					   const strom_preagg_target *targets, int ntargets,
					   const kern_parambuf *kparams,
					   const strom_preagg_domain *domain,
					   int dindex, int *p_errcode)
{
	return gpupreagg_session_new(key, targets, ntargets, kparams, domain, false, dindex, p_errcode);
}

extern "C" strom_gpupreagg *
strom_gpupreagg_create_hashed(strom_devprog_key key,
							  const strom_preagg_target *targets, int ntargets,
							  const kern_parambuf *kparams,
							  uint32_t ngroups_hint,
							  int dindex, int *p_errcode)
{
	/* the hashed kernels are a program of their own, derived from the caller's
	 * (strom_gpupreagg.h builds one family or the other); it builds in the
	 * background, the first fold parks behind it */
	Program *base = lookup_program(key);
	if (!base)
	{
		if (p_errcode)
			*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	const char *hashed_define = "#define GPUPREAGG_HASHED 1\n";
	std::string	hsource = base->source;
	if (hsource.compare(0, strlen(hashed_define), hashed_define) != 0)
		hsource = hashed_define + hsource;	/* (a caller may also hand over the derived program itself) */
	strom_devprog_key hkey = strom_get_devprog_key(hsource.c_str(), base->extra_flags);
	strom_gpupreagg *sess = gpupreagg_session_new(hkey, targets, ntargets, kparams, nullptr, true,
												  dindex, p_errcode);
	strom_put_devprog_key(hkey);			/* the session holds its own reference */
	if (!sess)
		return nullptr;
	/* the table itself is made by the first fold (the program may still be building) */
	cl_ulong	capacity = 1UL << 16;
	while (capacity < (cl_ulong)ngroups_hint * 2 && capacity < (1UL << 31))
		capacity <<= 1;
	sess->hash_capacity = (cl_uint)capacity;
	sess->groups_known = ngroups_hint;
	return sess;
}

static strom_gpupreagg *
gpupreagg_exact_child(strom_gpupreagg *sess, int *p_errcode)
{
	std::string	source = "#define GPUPREAGG_CHECKED 1\n" + sess->prog->source;
	strom_devprog_key ckey = strom_get_devprog_key(source.c_str(), sess->prog->extra_flags);
	if (!ckey)
	{
		*p_errcode = StromError_OutOfMemory;
		return nullptr;
	}
	strom_gpupreagg *child = gpupreagg_session_new(ckey, sess->targets.data(), (int)sess->targets.size(),
												   (const kern_parambuf *)sess->kparams.data(), nullptr, true,
												   sess->dev->dindex, p_errcode);
	strom_put_devprog_key(ckey);			/* the session holds its own reference */
	if (!child)
		return nullptr;
	child->hash_capacity = 1u << 16;		/* grows with the chunk's groups like any hashed table */
	return child;
}

/* the bound formula of a summed expression (GPUPREAGG_SUMBOUND_<a> of a generated program) over
 * a COLUMN chunk's zone maps: bits of the largest magnitude, or -1 (a column without a zone map,
 * another format, a malformed formula).  No device involved: what the launch path computes. */
extern "C" int
strom_gpupreagg_sum_bound_bits(const char *formula, const kern_data_store *kds)
{
	if (!formula || !kds || kds->format != KDS_FORMAT_COLUMN)
		return -1;
	return eval_sum_bound(formula, KERN_DATA_STORE_COLDIR(kds), kds->ncols);
}

extern "C" size_t
strom_gpupreagg_table_length(strom_gpupreagg *sess)
{
	return (sess && sess->has_domain) ? sess->table_bytes : 0;
}

extern "C" int
strom_gpupreagg_bind_table(strom_gpupreagg *sess, void *table_devptr)
{
	if (!sess || !sess->has_domain || !table_devptr)
		return StromError_BadRequestMessage;
	std::lock_guard<std::mutex> g(sess->lock);
	(void)hipSetDevice(sess->dev->hip_id);
	if (session_quiesce(sess) != hipSuccess ||
		hipMemcpy(table_devptr, sess->table, sess->table_bytes, hipMemcpyDeviceToDevice) != hipSuccess)
		return StromError_HipInternal;
	if (sess->table_owned)
		sess->dev->pool.release(sess->table);
	sess->table = (char *)table_devptr;
	sess->table_owned = false;
	return 0;
}

extern "C" void *
strom_gpupreagg_table_devptr(strom_gpupreagg *sess) { return sess ? sess->table : nullptr; }

extern "C" uint32_t
strom_gpupreagg_num_groups(strom_gpupreagg *sess)
{
	if (sess && sess->hashed)
	{
		cl_uint	ngroups = 0;
		std::lock_guard<std::mutex> g(sess->lock);
		(void)hipSetDevice(sess->dev->hip_id);
		if (!sess->htab || hash_table_ngroups(sess, &ngroups, nullptr) != 0)
			return 0;
		return ngroups;
	}
	return (sess && sess->has_domain) ? sess->ctl.ngroups : 0;
}

extern "C" uint32_t
strom_gpupreagg_checked_folds(strom_gpupreagg *sess) { return sess ? sess->checked_folds.load() : 0; }

extern "C" uint32_t
strom_gpupreagg_dense_groups(strom_gpupreagg *sess) { return (sess && sess->has_domain) ? sess->ctl.dense_ngroups : 0; }

extern "C" int
strom_gpupreagg_table_layout(strom_gpupreagg *sess, int resno, size_t *p_bits_off, size_t *p_vals_off)
{
	if (!sess || !sess->has_domain || resno < 0 || resno >= (int)sess->targets.size())
		return StromError_BadRequestMessage;
	if (sess->targets[resno].kind == STROM_PREAGG_KEY)
	{
		*p_bits_off = 0;
		*p_vals_off = 0;
		return 0;
	}
	int a = (int)(std::find(sess->agg_resno.begin(), sess->agg_resno.end(), resno) - sess->agg_resno.begin());
	*p_bits_off = 0;			/* flags word: bit 0 seen, bit 1+a has-value */
	*p_vals_off = sess->table_offset(1 + a, sess->ctl.ngroups);
	return 0;
}

static strom_task *
submit_gpupreagg_common(strom_gpupreagg *sess,
						const kern_data_store *kds, strom_dstore *kds_dev,
						const kern_row_map *krowmap, strom_rowmap *rowmap_dev,
						strom_done_cb done, void *arg, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	if (!sess || (!kds) == (!kds_dev))
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	if ((kds_dev && kds_dev->dindex != sess->dev->dindex) ||
		(rowmap_dev && (!kds_dev || krowmap || rowmap_dev->dindex != kds_dev->dindex)))
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	if (!sess->hashed && !sess->has_domain)
	{
		/* first chunk fixes the dense domain (see strom_hip.h) */
		*p_errcode = StromError_DataStoreOutOfRange;
		return nullptr;
	}
	preagg_request req;
	req.sess = sess;
	req.kds = kds;
	req.kds_dev = kds_dev;
	req.krowmap = (krowmap && krowmap->nvalids >= 0) ? krowmap : nullptr;
	req.rowmap_dev = rowmap_dev;
	kern_data_store head;
	if (kds)
		memcpy(&head, kds, offsetof(kern_data_store, colmeta));
	else
		head = kds_dev->head;
	req.format = head.format;
	if (!program_accepts_format(sess->prog, head.format))
	{
		*p_errcode = StromError_BadRequestMessage;		/* text columns live in heap tuples */
		return nullptr;
	}
	req.nrows = (rowmap_dev ? rowmap_dev->nvalids
				 : req.krowmap ? (uint32_t)req.krowmap->nvalids : head.nitems);
	sess->nfolds++;
	strom_task_impl *task = task_create(sess->dev, done, arg);
	if (sess->hashed)
		program_run_or_park(sess->prog, [task, req]() { gpupreagg_launch_hashed(task, req); });
	else
		program_run_or_park(sess->prog, [task, req]() { gpupreagg_launch(task, req); });
	return task;
	STROM_ABI_CATCH(nullptr, p_errcode)
}

extern "C" strom_task *
strom_submit_gpupreagg(strom_gpupreagg *sess,
					   const kern_data_store *kds, strom_dstore *kds_dev,
					   const kern_row_map *krowmap,
					   strom_done_cb done, void *arg, int *p_errcode)
{
	return submit_gpupreagg_common(sess, kds, kds_dev, krowmap, nullptr, done, arg, p_errcode);
}

extern "C" strom_task *
strom_submit_gpupreagg_mapped(strom_gpupreagg *sess, strom_dstore *kds_dev, strom_rowmap *rowmap,
							  strom_done_cb done, void *arg, int *p_errcode)
{
	if (!rowmap)
	{
		if (p_errcode)
			*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	return submit_gpupreagg_common(sess, nullptr, kds_dev, nullptr, rowmap, done, arg, p_errcode);
}


/*
 * fold the rows a finished GpuHashJoin produced, straight from its result
 * pairs (gpupreagg_dense_joined): see strom_hip.h
 */
static strom_task *
submit_gpupreagg_over_join(strom_gpupreagg *sess, strom_task *join_handle, bool lookup,
						   strom_hashjoin_table *tbl, strom_dstore *outer,
						   int ncols, const int32_t *src_depth, const int32_t *src_colidx,
						   const int32_t *type_oids,
						   strom_done_cb done, void *arg, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	strom_task_impl *jtask = static_cast<strom_task_impl *>(join_handle);
	if (!sess || sess->hashed || !sess->has_domain || (!lookup && !jtask) || !tbl || !outer ||
		ncols < 1 || ncols > 64 || !src_depth || !src_colidx || !type_oids)
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	if (jtask)
		task_wait_completed(jtask);
	cl_long	key_min = 0;
	cl_uint	nslots = 0;
	int		key_attno = 0, tbl_dindex = -1, has_outer_qual = 0;
	if ((jtask && (!jtask->res_is_join || !jtask->keep_main || jtask->errcode != 0 || !jtask->main_devptr)) ||
		hashjoin_table_direct_info(tbl, &key_min, &nslots, &key_attno, &tbl_dindex, &has_outer_qual) != 0 ||
		/* lookup mode runs the aggregate program only: a WHERE that lives in the
		 * join program would silently not be applied */
		(lookup && has_outer_qual) ||
		key_attno < 1 || outer->head.format != KDS_FORMAT_COLUMN ||
		/* (these kernels stream the outer columns; text variables take the plain fold) */
		(sess->prog->extra_flags & DEVTYPE_IS_VARLENA) != 0 ||
		outer->dindex != sess->dev->dindex || tbl_dindex != sess->dev->dindex)
	{
		/* needs: a finished join with STROM_RESULTS_ON_DEVICE, one inner relation
		 * with a DIRECT index and unique keys, joined on a plain outer column,
		 * over a resident COLUMN chunk */
		*p_errcode = ((jtask && jtask->errcode) ? jtask->errcode : StromError_BadRequestMessage);
		return nullptr;
	}
	(void)hipSetDevice(sess->dev->hip_id);
	/* the outer chunk's column widths */
	int		outer_ncols = (int)outer->head.ncols;
	std::vector<char> ohead(KDS_HEAD_LENGTH(outer_ncols));
	if (key_attno > outer_ncols ||
		hipMemcpy(ohead.data(), outer->devptr, ohead.size(), hipMemcpyDeviceToHost) != hipSuccess)
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	const kern_data_store *oh = (const kern_data_store *)ohead.data();
	auto	img = std::make_shared<std::vector<char>>(sizeof(joined_map_image), 0);
	joined_map_image *jm = (joined_map_image *)img->data();
	jm->ncols = ncols;
	jm->key_col = key_attno - 1;
	jm->key_attlen = oh->colmeta[key_attno - 1].attlen;
	jm->nslots = nslots;
	jm->key_min = key_min;
	for (int i = 0; i < ncols; i++)
	{
		int		attlen;
		switch (type_oids[i] < 0 ? -type_oids[i] : type_oids[i])
		{
			case STROM_BOOLOID: case STROM_BPCHAROID:	attlen = 1; break;
			case STROM_INT2OID:							attlen = 2; break;
			case STROM_INT4OID: case STROM_FLOAT4OID: case STROM_DATEOID:	attlen = 4; break;
			default:									attlen = 8; break;
		}
		jm->c[i].depth = src_depth[i];
		jm->c[i].col = src_colidx[i];
		if (src_depth[i] == 0)
		{
			if (src_colidx[i] < 0 || src_colidx[i] >= outer_ncols ||
				oh->colmeta[src_colidx[i]].attlen != attlen)
			{
				*p_errcode = StromError_DataStoreCorruption;
				return nullptr;
			}
		}
		else if (src_depth[i] == 1 && src_colidx[i] >= 0)
			;						/* packed slot records, below (the build kernel checks the column) */
		else
		{
			*p_errcode = StromError_BadRequestMessage;
			return nullptr;
		}
	}
	if (!(jm->key_attlen == 1 || jm->key_attlen == 2 || jm->key_attlen == 4 || jm->key_attlen == 8) ||
		(lookup && !(jm->key_attlen == 4 || jm->key_attlen == 8)))
	{
		*p_errcode = (lookup ? StromError_BadRequestMessage : StromError_DataStoreCorruption);
		return nullptr;
	}
	{
		/* one packed record per slot: presence, NULL bits and the wanted inner columns */
		int		cols[16], lens[16], which[16], ints[16], n = 0;
		unsigned offs[16], reclen = 0;
		void   *recs = nullptr;
		dimrec_narrow nw;
		for (int i = 0; i < ncols; i++)
		{
			if (src_depth[i] != 1)
				continue;
			if (n == 16)
			{
				*p_errcode = StromError_BadRequestMessage;
				return nullptr;
			}
			int		oid = (type_oids[i] < 0 ? -type_oids[i] : type_oids[i]);
			cols[n] = src_colidx[i];
			lens[n] = ((oid == STROM_BOOLOID || oid == STROM_BPCHAROID) ? 1 : oid == STROM_INT2OID ? 2
					   : (oid == STROM_INT4OID || oid == STROM_FLOAT4OID || oid == STROM_DATEOID) ? 4 : 8);
			ints[n] = !(oid == STROM_FLOAT4OID || oid == STROM_FLOAT8OID || oid == STROM_NUMERICOID);
			which[n++] = i;
		}
		/* the lookup kernel also reads the narrow form (2 / 4 bytes per slot: more of the table
		 * stays in L2 while the fact columns stream past) */
		int		rc = hashjoin_table_dimrecs(tbl, n, cols, lens, offs, &recs, &reclen,
											ints, lookup ? &nw : nullptr);
		if (rc != 0)
		{
			*p_errcode = rc;
			return nullptr;
		}
		for (int k = 0; k < n; k++)
		{
			jm->c[which[k]].dimvalues = offs[k];		/* byte offset in the record */
			jm->c[which[k]].dimisnull = 1 + k;			/* bit in the flags word */
		}
		jm->recs = (cl_ulong)(uintptr_t)recs;
		jm->reclen = reclen;
		if (lookup && nw.reclen != 0)
		{
			jm->recs = (cl_ulong)(uintptr_t)nw.recs;
			jm->reclen = nw.reclen;
			jm->narrow = 1;
			for (int k = 0; k < n; k++)
			{
				jm->c[which[k]].dimvalues = 0;
				jm->nshift[which[k]] = nw.shift[k];
				jm->nmask[which[k]] = nw.mask[k];
				jm->nmin[which[k]] = nw.vmin[k];
			}
		}
	}
	preagg_request req;
	req.sess = sess;
	req.kds = nullptr;
	req.kds_dev = outer;
	req.krowmap = nullptr;
	req.rowmap_dev = nullptr;
	req.format = KDS_FORMAT_COLUMN;
	req.joined_map = img;
	req.lookup = lookup;
	if (lookup)
		req.nrows = outer->head.nitems;
	else
	{
		req.nrows = jtask->res_nitems;
		req.joined_results = (const char *)jtask->main_devptr + jtask->res_offset;
		/* the result pairs now belong to this request: the join task can be
		 * waited for and released in any order */
		req.joined_buffer = jtask->main_devptr;
		jtask->main_devptr = nullptr;
		jtask->keep_main = false;
	}
	if (lookup)
		req.prog = lookup_mapping_program(sess, jm);
	sess->nfolds++;
	strom_task_impl *task = task_create(sess->dev, done, arg);
	program_run_or_park(req.prog ? req.prog : sess->prog, [task, req]() { gpupreagg_launch(task, req); });
	return task;
	STROM_ABI_CATCH(nullptr, p_errcode)
}

extern "C" strom_task *
strom_submit_gpupreagg_joined(strom_gpupreagg *sess, strom_task *join_handle,
							  strom_hashjoin_table *tbl, strom_dstore *outer,
							  int ncols, const int32_t *src_depth, const int32_t *src_colidx,
							  const int32_t *type_oids,
							  strom_done_cb done, void *arg, int *p_errcode)
{
	return submit_gpupreagg_over_join(sess, join_handle, false, tbl, outer, ncols, src_depth, src_colidx,
									  type_oids, done, arg, p_errcode);
}

/*
 * the join as a lookup inside the aggregate's own pass over the outer chunk
 * (gpupreagg_dense_lookup): no join request at all
 */
extern "C" strom_task *
strom_submit_gpupreagg_lookup(strom_gpupreagg *sess,
							  strom_hashjoin_table *tbl, strom_dstore *outer,
							  int ncols, const int32_t *src_depth, const int32_t *src_colidx,
							  const int32_t *type_oids,
							  strom_done_cb done, void *arg, int *p_errcode)
{
	return submit_gpupreagg_over_join(sess, nullptr, true, tbl, outer, ncols, src_depth, src_colidx,
									  type_oids, done, arg, p_errcode);
}

/*
 * census / compact: see gpupreagg_census in strom_gpupreagg.h.  Both are
 * planning-time calls (blocking); they must precede the first fold.
 */
extern "C" int
strom_gpupreagg_census(strom_gpupreagg *sess,
					   const kern_data_store *kds, strom_dstore *kds_dev,
					   const kern_row_map *krowmap,
					   uint32_t *bitmap_out, size_t nwords)
{
	if (!sess || !sess->has_domain || (!kds) == (!kds_dev))
		return StromError_BadRequestMessage;
	if (sess->ctl.remap != 0)
		return StromError_BadRequestMessage;		/* already compacted */
	Device *dev = sess->dev;
	if (kds_dev && kds_dev->dindex != dev->dindex)
		return StromError_BadRequestMessage;
	size_t	words = ((size_t)sess->ctl.dense_ngroups + 31) / 32;
	if (bitmap_out && nwords < words)
		return StromError_DataStoreNoSpace;
	if (!program_accepts_format(sess->prog, kds ? kds->format : kds_dev->head.format))
		return StromError_BadRequestMessage;
	if (strom_lookup_device_program(sess->key, 1) != STROM_DEVPROG_READY)
		return StromError_ProgramBuildFailure;
	std::lock_guard<std::mutex> g(sess->lock);
	(void)hipSetDevice(dev->hip_id);
	hipStream_t stream = dev->streams[0];
	int		errcode = 0;
	hipFunction_t fn = sess->prog->get_function(dev, "gpupreagg_census", &errcode);
	if (!fn)
		return errcode;
	if (!sess->d_census)
	{
		sess->d_census = (char *)dev->pool.alloc(words * sizeof(cl_uint));
		if (!sess->d_census)
			return StromError_OutOfMemory;
		if (hipMemsetAsync(sess->d_census, 0, words * sizeof(cl_uint), stream) != hipSuccess)
			return StromError_HipInternal;
	}
	size_t	kg_len = STROMALIGN(offsetof(kern_gpupreagg, kparams) + sess->kparams.size());
	std::vector<char> kg(kg_len, 0);
	memcpy(kg.data() + offsetof(kern_gpupreagg, kparams), sess->kparams.data(), sess->kparams.size());
	char   *d_kg = (char *)dev->pool.alloc(kg_len);
	void   *d_chunk = nullptr, *d_map = nullptr;
	int		rc = 0;
	do {
		if (!d_kg || hipMemcpyAsync(d_kg, kg.data(), kg_len, hipMemcpyHostToDevice, stream) != hipSuccess)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		const void *a_kds = (kds_dev ? kds_dev->devptr : nullptr);
		cl_uint	nrows;
		if (kds)
		{
			size_t	len = kds->length;
			if (kds->format == KDS_FORMAT_ROW)
				len = KERN_DATA_STORE_ROWBLOCK_OFFSET(kds) + (size_t)BLCKSZ * kds->nblocks;
			d_chunk = dev->pool.alloc(len);
			if (!d_chunk || hipMemcpyAsync(d_chunk, kds, len, hipMemcpyHostToDevice, stream) != hipSuccess)
			{
				rc = StromError_OutOfMemory;
				break;
			}
			a_kds = d_chunk;
			nrows = kds->nitems;
		}
		else
			nrows = kds_dev->head.nitems;
		const void *a_map = nullptr;
		if (krowmap && krowmap->nvalids >= 0)
		{
			size_t	len = offsetof(kern_row_map, rindex) + sizeof(cl_int) * (size_t)krowmap->nvalids;
			d_map = dev->pool.alloc(len);
			if (!d_map || hipMemcpyAsync(d_map, krowmap, len, hipMemcpyHostToDevice, stream) != hipSuccess)
			{
				rc = StromError_OutOfMemory;
				break;
			}
			a_map = d_map;
			nrows = (cl_uint)krowmap->nvalids;
		}
		const void *a_kg = d_kg;
		const void *a_toast = nullptr;
		const void *a_ctl = sess->d_ctl;
		void	   *a_bitmap = sess->d_census;
		void	   *args[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_ctl, &a_bitmap };
		unsigned	grid = (unsigned)std::min<size_t>(((size_t)nrows + 255) / 256,
													  (size_t)dev->prop.multiProcessorCount * 8);
		if (grid > 0 &&
			hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		if (bitmap_out &&
			hipMemcpyAsync(bitmap_out, sess->d_census, words * sizeof(cl_uint),
						   hipMemcpyDeviceToHost, stream) != hipSuccess)
			rc = StromError_HipInternal;
	} while (0);
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	if (d_kg) dev->pool.release(d_kg);
	if (d_chunk) dev->pool.release(d_chunk);
	if (d_map) dev->pool.release(d_map);
	return rc;
}

extern "C" int
strom_gpupreagg_compact(strom_gpupreagg *sess, const uint32_t *bitmap, size_t nwords)
{
	if (!sess || !sess->has_domain || sess->nfolds != 0 || !sess->table_owned || sess->ctl.remap != 0)
		return StromError_BadRequestMessage;
	Device *dev = sess->dev;
	dense_ctl &ctl = sess->ctl;
	size_t	words = ((size_t)ctl.dense_ngroups + 31) / 32;
	std::vector<cl_uint> bits(words, 0);
	std::lock_guard<std::mutex> g(sess->lock);

	(void)hipSetDevice(dev->hip_id);
	if (bitmap)
	{
		if (nwords < words)
			return StromError_BadRequestMessage;
		memcpy(bits.data(), bitmap, words * sizeof(cl_uint));
	}
	else if (sess->d_census)
	{
		if (session_quiesce(sess) != hipSuccess ||
			hipMemcpy(bits.data(), sess->d_census, words * sizeof(cl_uint), hipMemcpyDeviceToHost) != hipSuccess)
			return StromError_HipInternal;
	}
	std::vector<cl_uint> remap(ctl.dense_ngroups, 0xffffffffu);
	sess->present.clear();
	for (cl_uint d = 0; d < ctl.dense_ngroups; d++)
		if (bits[d >> 5] & (1u << (d & 31)))
		{
			remap[d] = (cl_uint)sess->present.size();
			sess->present.push_back(d);
		}
	if (sess->present.empty())
	{
		/* nothing passes the qual: keep one (never used) slot */
		sess->present.push_back(0);
	}
	sess->d_remap = (char *)dev->pool.alloc(remap.size() * sizeof(cl_uint));
	if (!sess->d_remap)
		return StromError_OutOfMemory;
	if (hipMemcpy(sess->d_remap, remap.data(), remap.size() * sizeof(cl_uint), hipMemcpyHostToDevice) != hipSuccess)
		return StromError_HipInternal;
	/* new table geometry over the compact slots */
	if (sess->table) dev->pool.release(sess->table);
	if (sess->d_ctl) dev->pool.release(sess->d_ctl);
	if (sess->d_slabs) dev->pool.release(sess->d_slabs);
	sess->table = sess->d_ctl = sess->d_slabs = nullptr;
	if (sess->d_export_spec)
	{
		dev->pool.release(sess->d_export_spec);
		sess->d_export_spec = nullptr;
	}
	ctl.ngroups = (cl_uint)sess->present.size();
	ctl.remap = (cl_ulong)(uintptr_t)sess->d_remap;
	int rc = setup_layout(sess);
	if (rc == 0)
		rc = alloc_session_buffers(sess);
	return rc;
}

/* what parallel.cpp needs to all-reduce the resident table in place */
bool
strom::gpupreagg_is_hashed(strom_gpupreagg *sess, int *p_dindex)
{
	if (!sess || !sess->hashed)
		return false;
	if (p_dindex)
		*p_dindex = sess->dev->dindex;
	return true;
}

/* the groups of a hashed session packed on the device: { knull, flags, keys[], vals[] } each */
int
strom::gpupreagg_hash_export_device(strom_gpupreagg *sess, char **p_recs, cl_uint *p_count, size_t *p_reclen,
									cl_ulong *p_sum_bound)
{
	Device *dev = sess->dev;
	size_t	reclen = 8 + 8 * (sess->key_resno.size() + sess->agg_resno.size());
	cl_uint	ngroups = 0, overflow = 0;
	std::lock_guard<std::mutex> g(sess->lock);

	*p_recs = nullptr;
	*p_count = 0;
	*p_reclen = reclen;
	if (p_sum_bound)
		*p_sum_bound = 0;
	(void)hipSetDevice(dev->hip_id);
	if (!sess->htab)
		return 0;
	int rc = hash_table_ngroups(sess, &ngroups, &overflow);
	if (rc == 0 && p_sum_bound)
		rc = hash_sum_measured_bound(sess, p_sum_bound);
	if (rc)
		return rc;
	if (overflow)
		return StromError_DataStoreNoSpace;
	if (ngroups == 0)
		return 0;
	int		errcode = 0;
	hipFunction_t fn = sess->prog->get_function(dev, "gpupreagg_hash_export", &errcode);
	if (!fn)
		return errcode;
	char   *d_out = (char *)dev->pool.alloc(reclen * ngroups + 16);
	if (!d_out)
		return StromError_OutOfMemory;
	char   *d_counter = d_out + reclen * ngroups;
	const void *a_tab = sess->htab;
	void	   *a_out = d_out, *a_cnt = d_counter;
	void	   *args[] = { &a_tab, &a_out, &a_cnt };
	unsigned	grid = std::min<unsigned>((sess->hash_capacity + 255) / 256, (unsigned)dev->prop.multiProcessorCount * 8);
	cl_uint		count = 0;
	bool ok = (hipMemsetAsync(d_counter, 0, 16, dev->streams[0]) == hipSuccess &&
			   hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) == hipSuccess &&
			   hipStreamSynchronize(dev->streams[0]) == hipSuccess &&
			   hipMemcpy(&count, d_counter, sizeof(count), hipMemcpyDeviceToHost) == hipSuccess &&
			   count == ngroups);
	if (!ok)
	{
		dev->pool.release(d_out);
		return StromError_HipInternal;
	}
	*p_recs = d_out;
	*p_count = ngroups;
	return 0;
}

/*
 * ... packed by owner for the hash-partitioned exchange (devlib: gpupreagg_hash_export_parts): the
 * records of owner p are h_counts[p] records starting at record sum(h_counts[0 .. p)).
 */
int
strom::gpupreagg_hash_export_parts_device(strom_gpupreagg *sess, cl_uint nparts, char **p_recs, cl_uint *h_counts,
										  size_t *p_reclen, cl_ulong *p_sum_bound)
{
	Device *dev = sess->dev;
	size_t	reclen = 8 + 8 * (sess->key_resno.size() + sess->agg_resno.size());
	cl_uint	ngroups = 0, overflow = 0;
	std::lock_guard<std::mutex> g(sess->lock);

	*p_recs = nullptr;
	*p_reclen = reclen;
	*p_sum_bound = 0;
	for (cl_uint p = 0; p < nparts; p++)
		h_counts[p] = 0;
	if (nparts < 1 || nparts > 64)
		return StromError_BadRequestMessage;
	(void)hipSetDevice(dev->hip_id);
	if (!sess->htab)
		return 0;
	int rc = hash_table_ngroups(sess, &ngroups, &overflow);
	if (rc == 0)
		rc = hash_sum_measured_bound(sess, p_sum_bound);
	if (rc)
		return rc;
	if (overflow)
		return StromError_DataStoreNoSpace;
	if (ngroups == 0)
		return 0;
	int		errcode = 0;
	hipFunction_t fn_count = sess->prog->get_function(dev, "gpupreagg_hash_owner_count", &errcode);
	hipFunction_t fn_parts = fn_count ? sess->prog->get_function(dev, "gpupreagg_hash_export_parts", &errcode) : nullptr;
	if (!fn_parts)
		return errcode;
	/* behind the records: counts, offsets, cursors (nparts words each) */
	size_t	words_at = STROM_TYPEALIGN(16, reclen * ngroups);
	char   *d_out = (char *)dev->pool.alloc(words_at + 3 * sizeof(cl_uint) * 64);
	if (!d_out)
		return StromError_OutOfMemory;
	cl_uint	   *d_counts = (cl_uint *)(d_out + words_at);
	cl_uint	   *d_offsets = d_counts + 64, *d_cursors = d_counts + 128;
	hipStream_t	stream = dev->streams[0];
	const void *a_tab = sess->htab;
	void	   *a_out = d_out, *a_cnt = d_counts, *a_off = d_offsets, *a_cur = d_cursors;
	void	   *args_count[] = { &a_tab, &nparts, &a_cnt };
	void	   *args_parts[] = { &a_tab, &a_out, &nparts, &a_off, &a_cur };
	unsigned	grid = std::min<unsigned>((sess->hash_capacity + 255) / 256, (unsigned)dev->prop.multiProcessorCount * 8);
	cl_uint		offsets[64], total = 0;
	bool ok = (hipMemsetAsync(d_counts, 0, 3 * sizeof(cl_uint) * 64, stream) == hipSuccess &&
			   hipModuleLaunchKernel(fn_count, grid, 1, 1, 256, 1, 1, 0, stream, args_count, nullptr) == hipSuccess &&
			   hipMemcpyAsync(h_counts, d_counts, sizeof(cl_uint) * nparts, hipMemcpyDeviceToHost, stream) == hipSuccess &&
			   hipStreamSynchronize(stream) == hipSuccess);
	for (cl_uint p = 0; ok && p < nparts; p++)
	{
		offsets[p] = total;
		total += h_counts[p];
	}
	ok = ok && total == ngroups &&
		hipMemcpyAsync(d_offsets, offsets, sizeof(cl_uint) * nparts, hipMemcpyHostToDevice, stream) == hipSuccess &&
		hipModuleLaunchKernel(fn_parts, grid, 1, 1, 256, 1, 1, 0, stream, args_parts, nullptr) == hipSuccess &&
		hipStreamSynchronize(stream) == hipSuccess;
	if (!ok)
	{
		dev->pool.release(d_out);
		return StromError_HipInternal;
	}
	*p_recs = d_out;
	return 0;
}

void
strom::gpupreagg_hash_release(strom_gpupreagg *sess, char *recs)
{
	if (recs)
		sess->dev->pool.release(recs);
}

/* packed groups (nsegs segments of seg_len records, h_counts[seg] of them set) into the table */
/*
 * incoming_sum_bound: the sum over the segments of "largest |integer sum| + 1" of the tables the
 * records come from (gpupreagg_hash_export*_device), or 0 when no incoming group can meet a group
 * of this table or another segment (disjoint keys).  The import adds 64-bit sums with plain
 * atomics: it runs only when this table's own bound plus the incoming one stays below 2^63 --
 * no group's sum can then leave int8 -- and answers CpuReCheck, before anything is touched,
 * otherwise (integer sums never wrap: strom_gpupreagg.h).
 */
int
strom::gpupreagg_hash_import_device(strom_gpupreagg *sess, const char *d_recs, cl_uint seg_len, cl_uint nsegs,
									const cl_uint *h_counts, cl_uint skip_seg, cl_ulong incoming_sum_bound)
{
	Device *dev = sess->dev;
	std::lock_guard<std::mutex> g(sess->lock);
	cl_ulong	incoming = 0;

	for (cl_uint s = 0; s < nsegs; s++)
		incoming += (s == skip_seg ? 0 : h_counts[s]);
	if (incoming == 0)
		return 0;
	(void)hipSetDevice(dev->hip_id);
	if (incoming_sum_bound == GPUPREAGG_IMPORT_EXACT && sess->nintsums != 0 && sess->htab)
	{
		/* one segment of pairwise different keys: group by group (gpupreagg_hash_import_verify) */
		int		e2 = 0;
		hipFunction_t fn_verify = sess->prog->get_function(dev, "gpupreagg_hash_import_verify", &e2);
		if (!fn_verify)
			return e2;
		if (nsegs != 1 || skip_seg != ~0u)
			return StromError_BadRequestMessage;
		cl_uint	   *d_flag = (cl_uint *)dev->pool.alloc(sizeof(cl_uint));
		if (!d_flag)
			return StromError_OutOfMemory;
		void	   *a_tab = sess->htab;
		const void *a_recs = d_recs;
		cl_uint		a_count = h_counts[0], flag = 0;
		void	   *a_flag = d_flag;
		void	   *vargs[] = { &a_tab, &a_recs, &a_count, &a_flag };
		unsigned	vgrid = (unsigned)std::max<size_t>(1, std::min<size_t>(((size_t)a_count + 255) / 256,
																			(size_t)dev->prop.multiProcessorCount * 8));
		bool	vok = (hipMemsetAsync(d_flag, 0, sizeof(cl_uint), dev->streams[0]) == hipSuccess &&
					   hipModuleLaunchKernel(fn_verify, vgrid, 1, 1, 256, 1, 1, 0, dev->streams[0], vargs, nullptr) == hipSuccess &&
					   hipMemcpyAsync(&flag, d_flag, sizeof(cl_uint), hipMemcpyDeviceToHost, dev->streams[0]) == hipSuccess &&
					   hipStreamSynchronize(dev->streams[0]) == hipSuccess);
		dev->pool.release(d_flag);
		if (!vok)
			return StromError_HipInternal;
		if (flag)
			return StromError_CpuReCheck;
	}
	else if (incoming_sum_bound != 0 && incoming_sum_bound != GPUPREAGG_IMPORT_EXACT && sess->nintsums != 0)
	{
		cl_ulong	mine = 0;
		int			brc = hash_sum_measured_bound(sess, &mine);
		if (brc)
			return brc;
		if (incoming_sum_bound >= (1UL << 63) || mine >= (1UL << 63) || mine + incoming_sum_bound >= (1UL << 63))
			return StromError_CpuReCheck;
	}
	int		errcode = 0;
	hipFunction_t fn = sess->prog->get_function(dev, "gpupreagg_hash_import", &errcode);
	if (!fn)
		return errcode;
	/* room for every incoming group, should all of them be new */
	cl_uint		ngroups = 0;
	int			rc = 0;
	if (!sess->htab)
		rc = hash_table_new(sess, sess->hash_capacity, &sess->htab, &sess->htab_bytes, nullptr);
	if (rc == 0)
		rc = hash_table_ngroups(sess, &ngroups, nullptr);
	if (rc == 0 && ((cl_ulong)ngroups + incoming) * 8 / 7 + 4096 > sess->hash_capacity)
		rc = hash_table_grow(sess, ((cl_ulong)ngroups + incoming) * 2 + 4096);
	if (rc)
		return rc;
	cl_uint	   *d_counts = (cl_uint *)dev->pool.alloc(sizeof(cl_uint) * nsegs);
	if (!d_counts)
		return StromError_OutOfMemory;
	void	   *a_tab = sess->htab;
	const void *a_recs = d_recs;
	const void *a_counts = d_counts;
	void	   *args[] = { &a_tab, &a_recs, &seg_len, &nsegs, &a_counts, &skip_seg };
	size_t		total = (size_t)seg_len * nsegs;
	unsigned	grid = (unsigned)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, (size_t)dev->prop.multiProcessorCount * 8));
	bool ok = (hipMemcpy(d_counts, h_counts, sizeof(cl_uint) * nsegs, hipMemcpyHostToDevice) == hipSuccess &&
			   hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) == hipSuccess &&
			   hipStreamSynchronize(dev->streams[0]) == hipSuccess);
	dev->pool.release(d_counts);
	if (!ok)
		return StromError_HipInternal;
	/* groups arrived otherwise than by a fold: the integer sums' bound is measured anew */
	rc = hash_sum_refresh(sess);
	if (rc)
		return rc;
	cl_uint		overflow = 0;
	rc = hash_table_ngroups(sess, &ngroups, &overflow);
	if (rc == 0)
	{
		sess->groups_upper = ngroups;
		sess->groups_known = std::max<cl_uint>(sess->groups_known, ngroups);
		if (overflow)
			rc = StromError_DataStoreNoSpace;
	}
	return rc;
}

int
strom::gpupreagg_get_merge_plan(strom_gpupreagg *sess, gpupreagg_merge_plan *plan)
{
	if (!sess || sess->hashed || !sess->has_domain || !sess->table || sess->agg_resno.size() > 31 ||
		sess->numeric_aggs)				/* (64-bit numerics do not add up with ncclSum: gather the partial rows) */
		return StromError_BadRequestMessage;
	memset(&plan->spec, 0, sizeof(plan->spec));
	plan->spec.ngroups = sess->ctl.ngroups;
	plan->spec.naggs = (cl_uint)sess->agg_resno.size();
	for (size_t a = 0; a < sess->agg_resno.size(); a++)
	{
		const strom_preagg_target &t = sess->targets[sess->agg_resno[a]];
		bool	isfloat = type_is_float(t.type_oid);
		cl_uint	op;
		switch (t.kind)
		{
			case STROM_PREAGG_NROWS:	op = 0; break;
			case STROM_PREAGG_PSUM:		op = (isfloat ? 2 : 1); break;
			case STROM_PREAGG_PMIN:		op = (isfloat ? 5 : 3); break;
			case STROM_PREAGG_PMAX:		op = (isfloat ? 6 : 4); break;
			default:					return StromError_BadRequestMessage;
		}
		plan->spec.op[a] = op;
		plan->spec.vals_off[a] = sess->table_offset(1 + (int)a, sess->ctl.ngroups);
		if (sess->is_intsum((int)a))
		{
			/* 128 bits wide in the table: travels as three carry-free limbs (strom_merge.h) */
			plan->spec.hi_off[a] = sess->table_hi_offset((int)a, sess->ctl.ngroups);
			plan->spec.mid_idx[a] = (cl_uint)sess->intsum_of[a];
		}
	}
	plan->table = sess->table;
	plan->table_bytes = sess->table_bytes;
	plan->nmid = (cl_uint)sess->nintsums;
	plan->dindex = sess->dev->dindex;
	return 0;
}

bool
strom::gpupreagg_sessions_mergeable(strom_gpupreagg *dst, strom_gpupreagg *src)
{
	if (!dst || !src || dst == src || dst->dev != src->dev || dst->prog != src->prog ||
		dst->hashed != src->hashed || dst->targets.size() != src->targets.size())
		return false;
	for (size_t i = 0; i < dst->targets.size(); i++)
		if (dst->targets[i].kind != src->targets[i].kind || dst->targets[i].type_oid != src->targets[i].type_oid ||
			dst->targets[i].scale != src->targets[i].scale)
			return false;
	if (dst->hashed)
		return true;
	/* dense tables: slot i must be the same group in both */
	const dense_ctl &a = dst->ctl, &b = src->ctl;
	if (!dst->has_domain || !src->has_domain || a.ngroups != b.ngroups || a.nkeys != b.nkeys ||
		a.dense_ngroups != b.dense_ngroups || dst->present != src->present)
		return false;
	for (cl_uint k = 0; k < a.nkeys; k++)
		if (a.key_min[k] != b.key_min[k] || a.key_range[k] != b.key_range[k] || a.key_stride[k] != b.key_stride[k])
			return false;
	return true;
}

int
strom::gpupreagg_get_census(strom_gpupreagg *sess, void **p_bitmap, cl_uint *p_nbits, int *p_dindex)
{
	if (!sess || sess->hashed || !sess->has_domain || sess->ctl.remap != 0)
		return StromError_BadRequestMessage;
	if (!sess->d_census)
	{
		/* a rank without rows still takes part in the union */
		size_t	words = ((size_t)sess->ctl.dense_ngroups + 31) / 32;
		std::lock_guard<std::mutex> g(sess->lock);
		(void)hipSetDevice(sess->dev->hip_id);
		sess->d_census = (char *)sess->dev->pool.alloc(words * sizeof(cl_uint));
		if (!sess->d_census)
			return StromError_OutOfMemory;
		if (hipMemset(sess->d_census, 0, words * sizeof(cl_uint)) != hipSuccess)
			return StromError_HipInternal;
	}
	*p_bitmap = sess->d_census;
	*p_nbits = sess->ctl.dense_ngroups;
	*p_dindex = sess->dev->dindex;
	return 0;
}

extern "C" void
strom_gpupreagg_reset(strom_gpupreagg *sess)
{
	if (sess && sess->hashed)
	{
		std::lock_guard<std::mutex> g(sess->lock);
		(void)hipSetDevice(sess->dev->hip_id);
		if (sess->htab)
			(void)hash_table_new(sess, sess->hash_capacity, &sess->htab, &sess->htab_bytes, sess->htab);
		sess->groups_upper = 0;
		sess->groups_known = 0;
		return;
	}
	if (!sess || !sess->table)
		return;
	(void)hipSetDevice(sess->dev->hip_id);
	(void)session_quiesce(sess);
	(void)hipMemset(sess->table, 0, sess->table_bytes);
}

extern "C" void
strom_gpupreagg_release(strom_gpupreagg *sess)
{
	if (!sess)
		return;
	Device *dev = sess->dev;
	(void)hipSetDevice(dev->hip_id);
	(void)session_quiesce(sess);
	if (sess->table_owned && sess->table)
		dev->pool.release(sess->table);
	if (sess->d_ctl)
		dev->pool.release(sess->d_ctl);
	if (sess->d_slabs)
		dev->pool.release(sess->d_slabs);
	if (sess->d_census)
		dev->pool.release(sess->d_census);
	if (sess->d_remap)
		dev->pool.release(sess->d_remap);
	if (sess->d_export_spec)
		dev->pool.release(sess->d_export_spec);
	for (auto &kv : sess->packed)
	{
		dev->pool.release(kv.second.d_ctl);
		dev->pool.release(kv.second.d_slabs);
	}
	for (int b = 0; b < 2; b++)
		if (sess->merge_ev[b])
			(void)hipEventDestroy(sess->merge_ev[b]);
	if (sess->htab)
		dev->pool.release(sess->htab);
	if (sess->key_checked)
		strom_put_devprog_key(sess->key_checked);
	for (auto &kv : sess->lookup_programs)
		strom_put_devprog_key(kv.second.first);
	strom_put_devprog_key(sess->key);
	delete sess;
}

namespace {

/* int64 fixed point at 10^-scale -> 64-bit device numeric; false when the
 * mantissa does not fit 57 bits */
bool
fixed_to_numeric(cl_long v, int scale, cl_ulong *out)
{
	bool		sign = (v < 0);
	cl_ulong	mant = (sign ? (cl_ulong)0 - (cl_ulong)v : (cl_ulong)v);
	int			expo = -scale;

	if (mant == 0)
	{
		*out = 0;
		return true;
	}
	while (mant % 10 == 0 && expo < 31)
	{
		mant /= 10;
		expo++;
	}
	if (mant >= (1UL << 57) || expo < -32 || expo > 31)
		return false;
	*out = (((cl_ulong)(cl_long)expo) << 58) | (sign ? (1UL << 57) : 0) | mant;
	return true;
}

struct numeric_spill {
	cl_uint		gid;
	int			resno;
	cl_ulong	image;
};

/*
 * An integer sum of the resident table is 128 bits wide {lo, hi} (strom_gpupreagg.h:
 * "integer sums never wrap"); a partial row carries an int8, or a 64-bit numeric with a
 * 57-bit mantissa.  A total that does not fit leaves as SEVERAL partial rows of its group --
 * partial rows add up, so the final aggregate is unchanged: datum images of the pieces, the
 * first for the group's own row.  int8: pieces of +-(2^63 - 1); numeric at 10^-scale: the
 * total's base-10^17 digits (each a 57-bit mantissa with its own exponent).
 */
int
sum_pieces(const strom_preagg_target &t, cl_ulong lo, cl_long hi, std::vector<cl_ulong> &out)
{
	__int128	total = ((__int128)hi << 64) | (__int128)(unsigned __int128)lo;
	out.clear();
	if (t.type_oid != STROM_NUMERICOID)
	{
		const __int128 most = (__int128)INT64_MAX;
		for (int n = 0; ; n++)
		{
			if (n == 4096)
				return StromError_DataStoreOutOfRange;	/* (a total beyond 2^75: not a sum of int8 rows) */
			if (total >= -most - 1 && total <= most)
			{
				out.push_back((cl_ulong)(cl_long)total);
				return 0;
			}
			cl_long piece = (total > 0 ? INT64_MAX : -INT64_MAX);
			out.push_back((cl_ulong)piece);
			total -= piece;
		}
	}
	bool		neg = (total < 0);
	unsigned __int128 mag = (neg ? (unsigned __int128)0 - (unsigned __int128)total : (unsigned __int128)total);
	cl_ulong	img;
	if (mag < ((unsigned __int128)1 << 63) &&
		fixed_to_numeric(neg ? -(cl_long)(cl_ulong)mag : (cl_long)(cl_ulong)mag, t.scale, &img))
	{
		out.push_back(img);
		return 0;
	}
	const cl_ulong P17 = 100000000000000000UL;
	for (int k = 0; mag != 0; k++)
	{
		cl_long		digit = (cl_long)(cl_ulong)(mag % P17);
		mag /= P17;
		if (digit == 0 && !(mag == 0 && out.empty()))
			continue;
		/* digit x 10^(17k) at 10^-scale = digit at 10^-(scale - 17k) */
		if (!fixed_to_numeric(neg ? -digit : digit, t.scale - 17 * k, &img))
			return StromError_DataStoreOutOfRange;
		out.push_back(img);
	}
	if (out.empty())
		out.push_back(0);
	return 0;
}

}	/* namespace */

namespace {

void
fetch_init_head(strom_gpupreagg *sess, kern_data_store *dest, size_t need, size_t nrooms)
{
	int		ncols = (int)sess->targets.size();
	memset(dest, 0, KDS_HEAD_LENGTH(ncols));
	dest->hostptr = (hostptr_t)(uintptr_t)&dest->hostptr;
	dest->length = (cl_uint)need;
	dest->ncols = ncols;
	dest->nrooms = (cl_uint)nrooms;
	dest->format = KDS_FORMAT_TUPSLOT;
	dest->tdtypeid = 2249;
	dest->tdtypmod = -1;
	for (int i = 0; i < ncols; i++)
	{
		const strom_preagg_target &t = sess->targets[i];
		int len = (t.kind == STROM_PREAGG_NROWS ? 8 : type_length(t.type_oid));
		dest->colmeta[i].attbyval = 1;
		dest->colmeta[i].attalign = (cl_char)len;
		dest->colmeta[i].attlen = (cl_short)len;
		dest->colmeta[i].attnum = (cl_short)(i + 1);
		dest->colmeta[i].attcacheoff = -1;
	}
}

/* a float accumulator image -> the column's datum */
cl_ulong
float_image_to_datum(const strom_preagg_target &t, cl_ulong raw, bool ordered)
{
	if (ordered)		/* order-preserving key -> IEEE bits */
		raw = (raw & 0x8000000000000000UL) ? (raw & 0x7fffffffffffffffUL) : ~raw;
	if (t.type_oid == STROM_FLOAT4OID)
	{
		double d; float f;
		memcpy(&d, &raw, 8);
		f = (float)d;
		raw = 0;
		memcpy(&raw, &f, 4);
	}
	return raw;
}

/*
 * hashed sessions: gpupreagg_hash_export packs the groups on the device,
 * only ngroups records cross PCIe
 */
long
gpupreagg_fetch_hashed(strom_gpupreagg *sess, kern_data_store *dest, size_t destlen)
{
	Device *dev = sess->dev;
	int		ncols = (int)sess->targets.size();
	size_t	nkeys = sess->key_resno.size(), naggs = sess->agg_resno.size();
	size_t	reclen = 8 + 8 * (nkeys + naggs);
	cl_uint	ngroups = 0, overflow = 0;
	std::vector<char> recs;
	std::lock_guard<std::mutex> g(sess->lock);

	(void)hipSetDevice(dev->hip_id);
	if (sess->htab)
	{
		int rc = hash_table_ngroups(sess, &ngroups, &overflow);
		if (rc)
			return -rc;
		if (overflow)
			return -StromError_DataStoreNoSpace;
	}
	/*
	 * No 64-bit numeric partial (those may leave as two rows: the host decides): the rows are
	 * formatted on the device and cross PCIe once, into the caller's buffer.  The host loop
	 * below took 56 ms for 1e6 groups, and asking for the size ran all of it too.
	 */
	bool	any_numeric = false;
	for (size_t a = 0; a < naggs; a++)
	{
		const strom_preagg_target &t = sess->targets[sess->agg_resno[a]];
		any_numeric = any_numeric || (t.type_oid == STROM_NUMERICOID && t.kind != STROM_PREAGG_NROWS);
	}
	if (!any_numeric && ncols <= 64 && !getenv("STROM_GPUPREAGG_FETCH_ON_HOST"))
	{
		size_t	stride = KDS_TUPSLOT_STRIDE(ncols);
		size_t	need = STROMALIGN(KDS_HEAD_LENGTH(ncols) + stride * (size_t)ngroups);
		if (!dest)
			return (long)need;
		if (destlen < need)
			return -StromError_DataStoreNoSpace;
		fetch_init_head(sess, dest, need, ngroups);
		if (ngroups > 0)
		{
			int		errcode = 0;
			hipFunction_t fn = sess->prog->get_function(dev, "gpupreagg_hash_export_rows", &errcode);
			if (!fn)
				return -errcode;
			cl_uint	tmeta[64];
			for (size_t k = 0; k < nkeys; k++)
			{
				const strom_preagg_target &t = sess->targets[sess->key_resno[k]];
				tmeta[sess->key_resno[k]] = (cl_uint)type_length(t.type_oid) | 0x100u | ((cl_uint)k << 16) |
					(type_is_float(t.type_oid) ? 0x200u : 0u) | (t.type_oid == STROM_FLOAT4OID ? 0x400u : 0u);
			}
			for (size_t a = 0; a < naggs; a++)
			{
				const strom_preagg_target &t = sess->targets[sess->agg_resno[a]];
				bool	nrows = (t.kind == STROM_PREAGG_NROWS);
				tmeta[sess->agg_resno[a]] = (cl_uint)(nrows ? 8 : type_length(t.type_oid)) | ((cl_uint)a << 16) |
					(nrows ? 0x800u : 0u) |
					(!nrows && type_is_float(t.type_oid) ? 0x200u : 0u) |
					(!nrows && t.type_oid == STROM_FLOAT4OID ? 0x400u : 0u) |
					(!nrows && type_is_float(t.type_oid) && t.kind != STROM_PREAGG_PSUM ? 0x1000u : 0u);
			}
			size_t	rows_len = stride * (size_t)ngroups;
			char   *d_rows = (char *)dev->pool.alloc(rows_len + 16 + sizeof(tmeta));
			if (!d_rows)
				return -StromError_OutOfMemory;
			char   *d_counter = d_rows + rows_len;
			char   *d_meta = d_counter + 16;
			const void *a_tab = sess->htab;
			void	   *a_rows = d_rows, *a_cnt = d_counter;
			const void *a_meta = d_meta;
			cl_uint		a_stride = (cl_uint)stride, a_ncols = (cl_uint)ncols;
			void	   *args[] = { &a_tab, &a_rows, &a_stride, &a_ncols, &a_meta, &a_cnt };
			unsigned	grid = std::min<unsigned>((sess->hash_capacity + 255) / 256,
												  (unsigned)dev->prop.multiProcessorCount * 8);
			cl_uint		count = 0;
			bool ok = (hipMemsetAsync(d_counter, 0, 16, dev->streams[0]) == hipSuccess &&
					   hipMemcpyAsync(d_meta, tmeta, sizeof(cl_uint) * ncols, hipMemcpyHostToDevice, dev->streams[0]) == hipSuccess &&
					   hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) == hipSuccess &&
					   hipStreamSynchronize(dev->streams[0]) == hipSuccess &&
					   hipMemcpy(&count, d_counter, sizeof(count), hipMemcpyDeviceToHost) == hipSuccess &&
					   count == ngroups &&
					   hipMemcpy((char *)dest + KDS_HEAD_LENGTH(ncols), d_rows, rows_len, hipMemcpyDeviceToHost) == hipSuccess);
			dev->pool.release(d_rows);
			if (!ok)
				return -StromError_HipInternal;
		}
		dest->nitems = ngroups;
		return (long)ngroups;
	}
	if (ngroups > 0)
	{
		int		errcode = 0;
		hipFunction_t fn = sess->prog->get_function(dev, "gpupreagg_hash_export", &errcode);
		if (!fn)
			return -errcode;
		char   *d_out = (char *)dev->pool.alloc(reclen * ngroups + 16);
		if (!d_out)
			return -StromError_OutOfMemory;
		char   *d_counter = d_out + reclen * ngroups;
		const void *a_tab = sess->htab;
		void	   *a_out = d_out, *a_cnt = d_counter;
		void	   *args[] = { &a_tab, &a_out, &a_cnt };
		unsigned	grid = std::min<unsigned>((sess->hash_capacity + 255) / 256,
											  (unsigned)dev->prop.multiProcessorCount * 8);
		cl_uint		count = 0;
		recs.resize(reclen * ngroups);
		bool ok = (hipMemsetAsync(d_counter, 0, 16, dev->streams[0]) == hipSuccess &&
				   hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, dev->streams[0], args, nullptr) == hipSuccess &&
				   hipStreamSynchronize(dev->streams[0]) == hipSuccess &&
				   hipMemcpy(&count, d_counter, sizeof(count), hipMemcpyDeviceToHost) == hipSuccess &&
				   count == ngroups &&
				   hipMemcpy(recs.data(), d_out, recs.size(), hipMemcpyDeviceToHost) == hipSuccess);
		dev->pool.release(d_out);
		if (!ok)
			return -StromError_HipInternal;
	}
	/* numeric sums too wide for the 64-bit form leave as two partial rows */
	const cl_long P17 = 100000000000000000L;
	struct spill { cl_uint rec; int resno; cl_ulong image; };
	std::vector<spill> spills;
	for (cl_uint r = 0; r < ngroups; r++)
	{
		const char *rec = recs.data() + reclen * r;
		cl_uint		flags = ((const cl_uint *)rec)[1];
		const cl_ulong *vals = (const cl_ulong *)(rec + 8) + nkeys;
		for (size_t a = 0; a < naggs; a++)
		{
			const strom_preagg_target &t = sess->targets[sess->agg_resno[a]];
			cl_ulong	img;
			if (t.type_oid != STROM_NUMERICOID || t.kind == STROM_PREAGG_NROWS || !(flags & (2u << a)))
				continue;
			cl_long v = (cl_long)vals[a];
			if (!fixed_to_numeric(v, t.scale, &img))
			{
				if (!fixed_to_numeric((v / P17) * P17, t.scale, &img))
					return -StromError_DataStoreOutOfRange;
				spills.push_back(spill{r, sess->agg_resno[a], img});
			}
		}
	}
	size_t	nrows_out = (size_t)ngroups + spills.size();
	size_t	need = STROMALIGN(KDS_HEAD_LENGTH(ncols) + KDS_TUPSLOT_STRIDE(ncols) * nrows_out);
	if (!dest)
		return (long)need;
	if (destlen < need)
		return -StromError_DataStoreNoSpace;
	fetch_init_head(sess, dest, need, nrows_out);
	auto put_keys = [&](const char *rec, Datum *values, cl_char *isnull)
	{
		cl_uint		knull = ((const cl_uint *)rec)[0];
		const cl_ulong *kimg = (const cl_ulong *)(rec + 8);
		for (size_t k = 0; k < nkeys; k++)
		{
			int		resno = sess->key_resno[k];
			const strom_preagg_target &t = sess->targets[resno];
			if (knull & (1u << k))
			{
				isnull[resno] = 1;
				continue;
			}
			isnull[resno] = 0;
			cl_ulong raw = kimg[k];
			if (type_is_float(t.type_oid))
				raw = float_image_to_datum(t, raw, false);
			values[resno] = 0;
			memcpy(&values[resno], &raw, type_length(t.type_oid));
		}
	};
	cl_uint	row = 0;
	for (cl_uint r = 0; r < ngroups; r++, row++)
	{
		const char *rec = recs.data() + reclen * r;
		cl_uint		flags = ((const cl_uint *)rec)[1];
		const cl_ulong *vals = (const cl_ulong *)(rec + 8) + nkeys;
		Datum	   *values = KERN_DATA_STORE_VALUES(dest, row);
		cl_char	   *isnull = KERN_DATA_STORE_ISNULL(dest, row);
		memset(values, 0, KDS_TUPSLOT_STRIDE(ncols));
		put_keys(rec, values, isnull);
		for (size_t a = 0; a < naggs; a++)
		{
			int		resno = sess->agg_resno[a];
			const strom_preagg_target &t = sess->targets[resno];
			cl_ulong raw = vals[a];
			if (t.kind == STROM_PREAGG_NROWS)
				values[resno] = raw;
			else if (!(flags & (2u << a)))
				isnull[resno] = 1;
			else if (t.type_oid == STROM_NUMERICOID)
			{
				cl_long v = (cl_long)raw;
				if (!fixed_to_numeric(v, t.scale, &raw))
					(void)fixed_to_numeric(v - (v / P17) * P17, t.scale, &raw);	/* high part: spill row */
				values[resno] = raw;
			}
			else if (type_is_float(t.type_oid))
				values[resno] = float_image_to_datum(t, raw, t.kind != STROM_PREAGG_PSUM);
			else
				memcpy(&values[resno], &raw, type_length(t.type_oid));
		}
	}
	for (const spill &sp : spills)
	{
		Datum	   *values = KERN_DATA_STORE_VALUES(dest, row);
		cl_char	   *isnull = KERN_DATA_STORE_ISNULL(dest, row);
		memset(values, 0, KDS_TUPSLOT_STRIDE(ncols));
		for (int i = 0; i < ncols; i++)
			isnull[i] = (sess->targets[i].kind != STROM_PREAGG_NROWS);	/* nrows = 0 */
		put_keys(recs.data() + reclen * sp.rec, values, isnull);
		isnull[sp.resno] = 0;
		values[sp.resno] = sp.image;
		row++;
	}
	dest->nitems = row;
	return (long)row;
}

}	/* namespace */

/* mirrors preagg_export_spec of devlib/strom_merge.h */
struct export_spec {
	cl_uint		ngroups, ncols, stride, nkeys;
	cl_long		key_min[8];
	cl_uint		key_range[8];
	cl_uint		key_stride[8];
	struct {
		cl_uint		kind, len, which, float4;
		cl_ulong	vals_off, hi_off;
	} col[64];
};

/*
 * dense sessions: the partial rows formatted on the device (preagg_dense_export_rows) and copied
 * into the caller's buffer once.  *p_fallback: not this way (compacted slots, numeric partials --
 * their sums may leave as several rows --, an integer sum beyond int8): the host loop below.
 */
static long
gpupreagg_fetch_dense_device(strom_gpupreagg *sess, kern_data_store *dest, size_t destlen, bool *p_fallback)
{
	Device	   *dev = sess->dev;
	int			ncols = (int)sess->targets.size();
	cl_uint		N = sess->ctl.ngroups;

	*p_fallback = true;
	if (!sess->present.empty() || ncols > 64 || sess->key_resno.size() > 8 || getenv("STROM_GPUPREAGG_FETCH_ON_HOST"))
		return 0;
	for (const strom_preagg_target &t : sess->targets)
		if (t.type_oid == STROM_NUMERICOID && t.kind != STROM_PREAGG_NROWS)
			return 0;
	int			errcode = 0;
	hipFunction_t fn = fixed_function(dev, "preagg_dense_export_rows", &errcode);
	if (!fn)
		return 0;
	size_t		stride = KDS_TUPSLOT_STRIDE(ncols);
	size_t		head_len = KDS_HEAD_LENGTH(ncols);
	std::lock_guard<std::mutex> g(sess->lock);
	(void)hipSetDevice(dev->hip_id);
	if (!sess->d_export_spec)
	{
		export_spec	spec;
		memset(&spec, 0, sizeof(spec));
		spec.ngroups = N;
		spec.ncols = (cl_uint)ncols;
		spec.stride = (cl_uint)stride;
		spec.nkeys = (cl_uint)sess->key_resno.size();
		for (size_t k = 0; k < sess->key_resno.size(); k++)
		{
			spec.key_min[k] = sess->ctl.key_min[k];
			spec.key_range[k] = sess->ctl.key_range[k];
			spec.key_stride[k] = std::max<cl_uint>(1, sess->ctl.key_stride[k]);
			auto &c = spec.col[sess->key_resno[k]];
			c.kind = 0;
			c.len = (cl_uint)type_length(sess->targets[sess->key_resno[k]].type_oid);
			c.which = (cl_uint)k;
		}
		for (size_t a = 0; a < sess->agg_resno.size(); a++)
		{
			const strom_preagg_target &t = sess->targets[sess->agg_resno[a]];
			auto &c = spec.col[sess->agg_resno[a]];
			bool	isfloat = type_is_float(t.type_oid);
			c.which = 1 + (cl_uint)a;
			c.vals_off = sess->table_offset(1 + (int)a, N);
			c.len = (cl_uint)(t.kind == STROM_PREAGG_NROWS ? 8 : type_length(t.type_oid));
			c.float4 = (t.type_oid == STROM_FLOAT4OID);
			if (t.kind == STROM_PREAGG_NROWS)
				c.kind = 1;
			else if (sess->is_intsum((int)a))
			{
				c.kind = 5;
				c.hi_off = sess->table_hi_offset((int)a, N);
			}
			else if (isfloat)
				c.kind = (t.kind == STROM_PREAGG_PSUM ? 3 : 4);
			else
				c.kind = 2;
		}
		sess->d_export_spec = (char *)dev->pool.alloc(sizeof(spec));
		if (!sess->d_export_spec ||
			hipMemcpy(sess->d_export_spec, &spec, sizeof(spec), hipMemcpyHostToDevice) != hipSuccess)
			return -StromError_OutOfMemory;
	}
	if (session_quiesce(sess) != hipSuccess)
		return -StromError_HipInternal;
	size_t		max_rows = (dest && destlen > head_len ? std::min<size_t>(N, (destlen - head_len) / stride) : 0);
	char	   *d_rows = (char *)dev->pool.alloc(stride * max_rows + 16);
	if (!d_rows)
		return -StromError_OutOfMemory;
	char	   *d_counter = d_rows + stride * max_rows;
	hipStream_t	stream = dev->streams[0];
	const void *a_table = sess->table;
	const void *a_spec = sess->d_export_spec;
	void	   *a_rows = d_rows, *a_cnt = d_counter;
	cl_uint		a_max = (cl_uint)max_rows;
	void	   *args[] = { &a_table, &a_spec, &a_rows, &a_max, &a_cnt };
	unsigned	grid = std::max(1u, std::min<unsigned>((N + 255) / 256, (unsigned)dev->prop.multiProcessorCount * 8));
	cl_uint		counter[2] = { 0, 0 };
	bool ok = (hipMemsetAsync(d_counter, 0, 16, stream) == hipSuccess &&
			   hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) == hipSuccess &&
			   hipMemcpyAsync(counter, d_counter, sizeof(counter), hipMemcpyDeviceToHost, stream) == hipSuccess &&
			   hipStreamSynchronize(stream) == hipSuccess);
	long		result;
	size_t		need = STROMALIGN(head_len + stride * (size_t)counter[0]);
	if (!ok)
		result = -StromError_HipInternal;
	else if (counter[1] != 0)
		result = 0;								/* a sum beyond int8: several rows, made on the host */
	else
	{
		*p_fallback = false;
		if (!dest)
			result = (long)need;
		else if (destlen < need)
			result = -StromError_DataStoreNoSpace;
		else
		{
			fetch_init_head(sess, dest, need, counter[0]);
			if (counter[0] > 0 &&
				hipMemcpy((char *)dest + head_len, d_rows, stride * (size_t)counter[0], hipMemcpyDeviceToHost) != hipSuccess)
				result = -StromError_HipInternal;
			else
			{
				dest->nitems = counter[0];
				result = (long)counter[0];
			}
		}
	}
	if (result < 0)
		*p_fallback = false;
	dev->pool.release(d_rows);
	return result;
}

/*
 * partial rows out: TUPSLOT, one row per group seen so far (plus, for a
 * numeric sum too wide for the 64-bit numeric form, one extra partial row
 * of the same group carrying the high part: partial rows add up, so the
 * final aggregate is unchanged)
 */
extern "C" long
strom_gpupreagg_fetch(strom_gpupreagg *sess, kern_data_store *dest, size_t destlen)
{
	if (sess && sess->hashed)
		return gpupreagg_fetch_hashed(sess, dest, destlen);
	if (!sess || !sess->has_domain)
		return -StromError_BadRequestMessage;
	{
		bool	fallback = false;
		long	r = gpupreagg_fetch_dense_device(sess, dest, destlen, &fallback);
		if (!fallback)
			return r;
	}
	Device *dev = sess->dev;
	int		ncols = (int)sess->targets.size();
	cl_uint	N = sess->ctl.ngroups;
	std::vector<char> host(sess->table_bytes);

	(void)hipSetDevice(dev->hip_id);
	if (session_quiesce(sess) != hipSuccess ||
		hipMemcpy(host.data(), sess->table, sess->table_bytes, hipMemcpyDeviceToHost) != hipSuccess)
		return -StromError_HipInternal;
	const cl_uint *gflags = (const cl_uint *)host.data();
	size_t	ngroups = 0;
	std::vector<numeric_spill> spills;
	std::vector<cl_ulong> pieces;
	for (cl_uint g = 0; g < N; g++)
	{
		if (!(gflags[g] & 1))
			continue;
		ngroups++;
		for (size_t a = 0; a < sess->agg_resno.size(); a++)
		{
			/* an integer sum (128 bits in the table) that does not fit its datum leaves as
			 * several partial rows: see sum_pieces */
			const strom_preagg_target &t = sess->targets[sess->agg_resno[a]];
			if (!sess->is_intsum((int)a) || !(gflags[g] & (2u << a)))
				continue;
			cl_ulong lo = ((const cl_ulong *)(host.data() + sess->table_offset(1 + (int)a, N)))[g];
			cl_long	 hi = ((const cl_long *)(host.data() + sess->table_hi_offset((int)a, N)))[g];
			int		rc = sum_pieces(t, lo, hi, pieces);
			if (rc != 0)
				return -rc;
			for (size_t k = 1; k < pieces.size(); k++)
				spills.push_back(numeric_spill{g, sess->agg_resno[a], pieces[k]});
		}
	}
	ngroups += spills.size();
	size_t	need = STROMALIGN(KDS_HEAD_LENGTH(ncols) + KDS_TUPSLOT_STRIDE(ncols) * ngroups);
	if (!dest)
		return (long)need;
	if (destlen < need)
		return -StromError_DataStoreNoSpace;
	memset(dest, 0, KDS_HEAD_LENGTH(ncols));
	dest->hostptr = (hostptr_t)(uintptr_t)&dest->hostptr;
	dest->length = (cl_uint)need;
	dest->ncols = ncols;
	dest->nrooms = (cl_uint)ngroups;
	dest->format = KDS_FORMAT_TUPSLOT;
	dest->tdtypeid = 2249;
	dest->tdtypmod = -1;
	for (int i = 0; i < ncols; i++)
	{
		const strom_preagg_target &t = sess->targets[i];
		int len = (t.kind == STROM_PREAGG_NROWS ? 8 : type_length(t.type_oid));
		dest->colmeta[i].attbyval = 1;
		dest->colmeta[i].attalign = (cl_char)len;
		dest->colmeta[i].attlen = (cl_short)len;
		dest->colmeta[i].attnum = (cl_short)(i + 1);
		dest->colmeta[i].attcacheoff = -1;
	}
	cl_uint	row = 0;
	for (cl_uint g = 0; g < N; g++)
	{
		if (!(gflags[g] & 1))
			continue;
		Datum	   *values = KERN_DATA_STORE_VALUES(dest, row);
		cl_char	   *isnull = KERN_DATA_STORE_ISNULL(dest, row);
		memset(values, 0, KDS_TUPSLOT_STRIDE(ncols));
		for (size_t k = 0; k < sess->key_resno.size(); k++)
		{
			int		resno = sess->key_resno[k];
			cl_uint	dense = (sess->present.empty() ? g : sess->present[g]);
			cl_uint	off = (dense / sess->ctl.key_stride[k]) % (sess->ctl.key_range[k] + 1);
			if (off == sess->ctl.key_range[k])
				isnull[resno] = 1;
			else
			{
				cl_long v = sess->ctl.key_min[k] + off;
				memcpy(&values[resno], &v, type_length(sess->targets[resno].type_oid));
			}
		}
		for (size_t a = 0; a < sess->agg_resno.size(); a++)
		{
			int		resno = sess->agg_resno[a];
			const strom_preagg_target &t = sess->targets[resno];
			const cl_ulong *vals = (const cl_ulong *)(host.data() + sess->table_offset(1 + (int)a, N));
			if (t.kind == STROM_PREAGG_NROWS)
				values[resno] = vals[g];
			else if (!(gflags[g] & (2u << a)))
				isnull[resno] = 1;
			else
			{
				cl_ulong raw = vals[g];
				if (t.type_oid == STROM_NUMERICOID && t.scale < 0)
					values[resno] = raw;			/* accumulated in the 64-bit form itself */
				else if (sess->is_intsum((int)a))
				{
					/* the first piece; the others went to spill rows above */
					cl_long	 hi = ((const cl_long *)(host.data() + sess->table_hi_offset((int)a, N)))[g];
					(void)sum_pieces(t, raw, hi, pieces);
					values[resno] = pieces[0];
				}
				else if (t.type_oid == STROM_NUMERICOID)
				{
					/* pmin / pmax of a fixed-point numeric: one int8 value */
					if (!fixed_to_numeric((cl_long)raw, t.scale, &raw))
						return -StromError_DataStoreOutOfRange;
					values[resno] = raw;
				}
				else if (type_is_float(t.type_oid))
				{
					if (t.kind != STROM_PREAGG_PSUM)
					{
						/* order-preserving key -> IEEE bits */
						raw = (raw & 0x8000000000000000UL) ? (raw & 0x7fffffffffffffffUL) : ~raw;
					}
					if (t.type_oid == STROM_FLOAT4OID)
					{
						double d; float f;
						memcpy(&d, &raw, 8);
						f = (float)d;
						raw = 0;
						memcpy(&raw, &f, 4);
					}
					values[resno] = raw;
				}
				else
					memcpy(&values[resno], &raw, type_length(t.type_oid));
			}
		}
		row++;
	}
	for (const numeric_spill &sp : spills)
	{
		Datum	   *values = KERN_DATA_STORE_VALUES(dest, row);
		cl_char	   *isnull = KERN_DATA_STORE_ISNULL(dest, row);
		memset(values, 0, KDS_TUPSLOT_STRIDE(ncols));
		for (int i = 0; i < ncols; i++)
			isnull[i] = (sess->targets[i].kind != STROM_PREAGG_NROWS);	/* nrows = 0 */
		for (size_t k = 0; k < sess->key_resno.size(); k++)
		{
			int		resno = sess->key_resno[k];
			cl_uint	dense = (sess->present.empty() ? sp.gid : sess->present[sp.gid]);
			cl_uint	off = (dense / sess->ctl.key_stride[k]) % (sess->ctl.key_range[k] + 1);
			if (off != sess->ctl.key_range[k])
			{
				cl_long v = sess->ctl.key_min[k] + off;
				isnull[resno] = 0;
				memcpy(&values[resno], &v, type_length(sess->targets[resno].type_oid));
			}
		}
		isnull[sp.resno] = 0;
		values[sp.resno] = sp.image;
		row++;
	}
	dest->nitems = row;
	return (long)row;
}

/* ================================================================== *
 * The reference's per-chunk message: pgstrom_gpupreagg {msg, dprog_key,
 * needs_grouping, num_groups, pds, pds_dest, kern_gpupreagg}
 * (opencl_gpupreagg.h:994-1003), served by clserv_process_gpupreagg
 * (gpupreagg.c:3849-4240).  One chunk in, the chunk's partial rows out in
 * kds_dest -- nothing is kept between requests.  Built from the session
 * machinery above: key range of the chunk -> dense table geometry (or the
 * hashed GROUP BY when the keys do not map to dense ids) -> fold -> fetch.
 * The steps have host decisions in between, so the request runs on the
 * device's worker thread; the submitter only queues.
 * ================================================================== */
namespace {

/* host image of gpupreagg_keyrange_t (strom_gpupreagg.h) */
struct keyrange_image {
	cl_long		kmin[STROM_PREAGG_MAXKEYS];
	cl_long		kmax[STROM_PREAGG_MAXKEYS];
	cl_uint		nvalues[STROM_PREAGG_MAXKEYS];
	cl_uint		nrows;
	cl_uint		pad;
};

int
chunk_domain(strom_devprog_key key, Program *prog, Device *dev,
			 const strom_preagg_target *targets, int ntargets,
			 const kern_parambuf *kparams, strom_dstore *kds_dev,
			 const kern_row_map *krowmap, strom_preagg_domain *dom)
{
	int		nkeys = 0;
	memset(dom, 0, sizeof(*dom));
	for (int i = 0; i < ntargets; i++)
	{
		if (targets[i].kind != STROM_PREAGG_KEY)
			continue;
		if (type_is_float(targets[i].type_oid) || targets[i].type_oid == STROM_NUMERICOID)
			return StromError_DataStoreOutOfRange;		/* no dense ids for these */
		nkeys++;
	}
	if (nkeys > STROM_PREAGG_MAXKEYS)
		return StromError_DataStoreOutOfRange;
	dom->nkeys = nkeys;
	if (nkeys == 0)
		return 0;
	if (!program_accepts_format(prog, kds_dev->head.format))
		return StromError_BadRequestMessage;
	/*
	 * a COLUMN chunk whose group keys are plain columns: the zone maps bound the keys (of all
	 * rows -- a superset of the rows the qual keeps, which is all a dense domain needs): no pass
	 * over the chunk, no round trip (the key-range kernel and its wait were 165 us of a 325k-row
	 * message, profiles/r02_chunk_message_probe.txt).  No row map: its rows may be few.
	 */
	if (kds_dev->head.format == KDS_FORMAT_COLUMN && !krowmap && !getenv("STROM_GPUPREAGG_NO_ZONE_DOMAIN"))
	{
		const char *lst = strstr(prog->source.c_str(), "#define GPUPREAGG_KEYCOLS_LIST(X)");
		std::shared_ptr<std::vector<kern_coldir>> cd = dstore_coldir(kds_dev);
		bool		ok = (lst != nullptr && cd != nullptr);
		const char *eol = (lst ? strchr(lst, '\n') : nullptr);
		int			found = 0;
		for (const char *p = (ok ? strstr(lst, " X(") : nullptr); ok && p && (!eol || p < eol); p = strstr(p + 1, " X("))
		{
			int		kidx = -1, attno = 0;
			if (sscanf(p, " X(%d,%d)", &kidx, &attno) != 2 || kidx != found || kidx >= nkeys ||
				attno < 1 || attno > (int)cd->size())
			{
				ok = false;
				break;
			}
			const kern_coldir &c = (*cd)[attno - 1];
			if (c.stat_flags & KDS_COLSTAT_ISFLOAT)
				ok = false;
			else if (!(c.stat_flags & KDS_COLSTAT_MINMAX))
			{
				dom->key_min[kidx] = 0;			/* NULL keys only: the NULL slot alone */
				dom->key_range[kidx] = 0;
			}
			else
			{
				cl_ulong span = (cl_ulong)c.maxval - (cl_ulong)c.minval;
				if (c.maxval < c.minval || span >= 0xfffffffeUL)
					ok = false;
				dom->key_min[kidx] = c.minval;
				dom->key_range[kidx] = (cl_uint)span + 1;
			}
			found++;
		}
		if (ok && found == nkeys)
			return 0;
		memset(dom, 0, sizeof(*dom));
		dom->nkeys = nkeys;
	}
	if (strom_lookup_device_program(key, 1) != STROM_DEVPROG_READY)
		return StromError_ProgramBuildFailure;
	(void)hipSetDevice(dev->hip_id);
	hipStream_t stream = dev->pick_stream();
	int		errcode = 0;
	hipFunction_t fn = prog->get_function(dev, "gpupreagg_keyrange", &errcode);
	if (!fn)
		return errcode;
	keyrange_image img;
	memset(&img, 0, sizeof(img));
	for (int k = 0; k < STROM_PREAGG_MAXKEYS; k++)
	{
		img.kmin[k] = INT64_MAX;
		img.kmax[k] = INT64_MIN;
	}
	size_t	kg_len = STROMALIGN(offsetof(kern_gpupreagg, kparams) + kparams->length);
	std::vector<char> kg(kg_len, 0);
	memcpy(kg.data() + offsetof(kern_gpupreagg, kparams), kparams, kparams->length);
	char   *d_kg = (char *)dev->pool.alloc(kg_len);
	char   *d_out = (char *)dev->pool.alloc(sizeof(img));
	void   *d_map = nullptr;
	int		rc = 0;
	do {
		if (!d_kg || !d_out)
		{
			rc = StromError_OutOfMemory;
			break;
		}
		if (hipMemcpyAsync(d_kg, kg.data(), kg_len, hipMemcpyHostToDevice, stream) != hipSuccess ||
			hipMemcpyAsync(d_out, &img, sizeof(img), hipMemcpyHostToDevice, stream) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		cl_uint	nrows = kds_dev->head.nitems;
		const void *a_map = nullptr;
		if (krowmap && krowmap->nvalids >= 0)
		{
			size_t	len = offsetof(kern_row_map, rindex) + sizeof(cl_int) * (size_t)krowmap->nvalids;
			d_map = dev->pool.alloc(len);
			if (!d_map || hipMemcpyAsync(d_map, krowmap, len, hipMemcpyHostToDevice, stream) != hipSuccess)
			{
				rc = StromError_OutOfMemory;
				break;
			}
			a_map = d_map;
			nrows = (cl_uint)krowmap->nvalids;
		}
		const void *a_kg = d_kg;
		const void *a_kds = kds_dev->devptr;
		const void *a_toast = nullptr;
		void	   *a_out = d_out;
		void	   *args[] = { &a_kg, &a_kds, &a_toast, &a_map, &a_out };
		unsigned	grid = (unsigned)std::min<size_t>(((size_t)nrows + 255) / 256,
													  (size_t)dev->prop.multiProcessorCount * 8);
		if (grid > 0 &&
			hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, stream, args, nullptr) != hipSuccess)
		{
			rc = StromError_HipInternal;
			break;
		}
		if (hipMemcpyAsync(&img, d_out, sizeof(img), hipMemcpyDeviceToHost, stream) != hipSuccess)
			rc = StromError_HipInternal;
	} while (0);
	if (hipStreamSynchronize(stream) != hipSuccess && rc == 0)
		rc = StromError_HipInternal;
	if (d_kg) dev->pool.release(d_kg);
	if (d_out) dev->pool.release(d_out);
	if (d_map) dev->pool.release(d_map);
	if (rc != 0)
		return rc;
	for (int k = 0; k < nkeys; k++)
	{
		if (img.nvalues[k] == 0)
		{
			dom->key_min[k] = 0;		/* NULL keys only (or no row at all): the NULL slot alone */
			dom->key_range[k] = 0;
			continue;
		}
		/* (unsigned arithmetic: max - min of two int64 may not fit int64) */
		cl_ulong span = (cl_ulong)img.kmax[k] - (cl_ulong)img.kmin[k];
		if (span >= 0xfffffffeUL)
			return StromError_DataStoreOutOfRange;
		dom->key_min[k] = img.kmin[k];
		dom->key_range[k] = (cl_uint)span + 1;
	}
	return 0;
}

}	/* namespace */

/* the key domain the last per-chunk message of a program measured, widened by every later one
 * (strom_submit_gpupreagg_chunk: heap-page chunks have no zone maps) */
static std::mutex domain_hint_lock;
static std::map<strom_devprog_key, strom_preagg_domain> domain_hints;

static bool
domain_hint_get(strom_devprog_key key, strom_preagg_domain *dom)
{
	std::lock_guard<std::mutex> g(domain_hint_lock);
	auto it = domain_hints.find(key);
	if (it == domain_hints.end())
		return false;
	*dom = it->second;
	return true;
}

static void
domain_hint_put(strom_devprog_key key, const strom_preagg_domain &dom)
{
	std::lock_guard<std::mutex> g(domain_hint_lock);
	auto it = domain_hints.find(key);
	if (it == domain_hints.end() || it->second.nkeys != dom.nkeys)
	{
		if (domain_hints.size() > 4096)
			domain_hints.clear();					/* (a hint: forgetting costs one key-range pass) */
		domain_hints[key] = dom;
		return;
	}
	strom_preagg_domain &old = it->second;
	for (int k = 0; k < dom.nkeys && k < STROM_PREAGG_MAXKEYS; k++)
	{
		if (dom.key_range[k] == 0)
			continue;								/* NULL keys only: nothing to widen by */
		if (old.key_range[k] == 0)
		{
			old.key_min[k] = dom.key_min[k];
			old.key_range[k] = dom.key_range[k];
			continue;
		}
		__int128	lo = std::min<__int128>(old.key_min[k], dom.key_min[k]);
		__int128	hi = std::max<__int128>((__int128)old.key_min[k] + old.key_range[k], (__int128)dom.key_min[k] + dom.key_range[k]);
		if (hi - lo >= 0xfffffffeLL)
		{
			old = dom;								/* too wide to be a dense domain: start over */
			return;
		}
		old.key_min[k] = (int64_t)lo;
		old.key_range[k] = (uint32_t)(hi - lo);
	}
}

extern "C" int
strom_gpupreagg_chunk_domain(strom_devprog_key key,
							 const strom_preagg_target *targets, int ntargets,
							 const kern_parambuf *kparams,
							 const kern_data_store *kds, strom_dstore *kds_dev,
							 const kern_row_map *krowmap,
							 int dindex, strom_preagg_domain *domain_out)
{
	STROM_ABI_TRY
	Program *prog = lookup_program(key);
	Device *dev = get_device(dindex);
	if (!prog || !dev || !targets || ntargets < 1 || !kparams || !domain_out || (!kds) == (!kds_dev) ||
		(kds_dev && kds_dev->dindex != dindex))
		return (!dev ? StromError_ServerNotReady : StromError_BadRequestMessage);
	strom_dstore *tmp = nullptr;
	if (!kds_dev)
	{
		tmp = strom_dstore_upload(kds, dindex);
		if (!tmp)
			return StromError_OutOfMemory;
		kds_dev = tmp;
	}
	int rc = chunk_domain(key, prog, dev, targets, ntargets, kparams, kds_dev, krowmap, domain_out);
	if (tmp)
		strom_dstore_release(tmp);
	return rc;
	STROM_ABI_CATCH(StromError_OutOfMemory, (int *)nullptr)
}

extern "C" strom_task *
strom_submit_gpupreagg_chunk(strom_devprog_key key,
							 const strom_preagg_target *targets, int ntargets,
							 kern_gpupreagg *kgpreagg,
							 const kern_data_store *kds, strom_dstore *kds_dev,
							 kern_data_store *kds_dest, size_t dest_length,
							 int needs_grouping, double num_groups,
							 int dindex,
							 strom_done_cb done, void *arg, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	Program *prog = lookup_program(key);
	Device *dev = get_device(dindex);
	if (!prog || !dev || !targets || ntargets < 1 || !kgpreagg || !kds_dest ||
		(!kds) == (!kds_dev) || (kds_dev && kds_dev->dindex != dindex) ||
		kgpreagg->kparams.length < offsetof(kern_parambuf, poffset) ||
		dest_length < (size_t)KDS_HEAD_LENGTH(ntargets))
	{
		*p_errcode = (!dev ? StromError_ServerNotReady : StromError_BadRequestMessage);
		return nullptr;
	}
	int		nkeys = 0;
	for (int i = 0; i < ntargets; i++)
		nkeys += (targets[i].kind == STROM_PREAGG_KEY);
	if ((needs_grouping != 0) != (nkeys > 0))
	{
		/* the message says GROUP BY, the program has no key (or the reverse) */
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	std::vector<strom_preagg_target> tg(targets, targets + ntargets);
	strom_task_impl *task = task_create(dev, done, arg);
	device_run_async(dev, [=]() {
		const kern_parambuf *kparams = KERN_GPUPREAGG_PARAMBUF(kgpreagg);
		const kern_row_map *krowmap = KERN_GPUPREAGG_KROWMAP(kgpreagg);
		strom_dstore	   *tmp = nullptr, *src = kds_dev;
		strom_gpupreagg	   *sess = nullptr;
		strom_perfmon		pfm;
		int					rc = 0;
		memset(&pfm, 0, sizeof(pfm));
		if (krowmap->nvalids < 0)
			krowmap = nullptr;
		do {
			if (strom_lookup_device_program(key, 1) != STROM_DEVPROG_READY)
			{
				rc = StromError_ProgramBuildFailure;
				break;
			}
			if (!src)
			{
				src = tmp = strom_dstore_upload(kds, dindex);
				if (!tmp)
				{
					rc = StromError_OutOfMemory;
					break;
				}
			}
			strom_preagg_domain dom;
			/*
			 * The key domain.  A COLUMN chunk's zone maps give it for nothing (chunk_domain); heap
			 * pages have none, and the key-range pass over them -- a kernel and a wait -- is a third
			 * of such a message.  The chunks of one scan look alike, so the domain the PREVIOUS
			 * message of this program measured is tried first: a key outside it makes the fold answer
			 * DataStoreOutOfRange, the range is measured after all (and remembered, widened) and the
			 * chunk folded again.  A hint, not state: the answer never depends on it.
			 */
			bool		from_hint = false;
			if (src->head.format != KDS_FORMAT_COLUMN && !getenv("STROM_GPUPREAGG_NO_DOMAIN_HINT"))
				from_hint = domain_hint_get(key, &dom);
			if (from_hint)
				rc = 0;
			else
			{
				rc = chunk_domain(key, prog, dev, tg.data(), ntargets, kparams, src, krowmap, &dom);
				if (rc == 0 && src->head.format != KDS_FORMAT_COLUMN)
					domain_hint_put(key, dom);
			}
		fold_again:
			uint32_t hint = (num_groups > 0 && num_groups < 4e9 ? (uint32_t)num_groups : 0);
			if (rc == 0)
			{
				/*
				 * dense ids past ~1.6e5: the dense kernels split them over id-range roles that
				 * each read every row (1e5 ids: 1.4 ms per 1e8 rows, 64 roles at most), the
				 * hashed GROUP BY's partition plan takes 2.3 ms whatever the count
				 * (profiles/r02_dense_vs_hashed.txt)
				 */
				double	ids = 1.0;
				int		nk = 0;
				for (int i = 0; i < ntargets; i++)
				{
					if (tg[i].kind == STROM_PREAGG_KEY && nk < STROM_PREAGG_MAXKEYS)
						ids *= (double)dom.key_range[nk++];
				}
				if (ids > 160000.0 && !getenv("STROM_GPUPREAGG_CHUNK_DENSE_ONLY"))
				{
					double	bound = std::min(ids, std::max(1.0, (double)src->head.nitems / 8));
					int		rc2 = 0;
					sess = strom_gpupreagg_create_hashed(key, tg.data(), ntargets, kparams,
														 hint ? hint : (uint32_t)std::min(bound, 4.0e6), dindex, &rc2);
					/* (what the hashed kernels do not take -- 64-bit numeric partials -- stays dense) */
				}
			}
			if (rc == 0 && !sess)
				sess = strom_gpupreagg_create(key, tg.data(), ntargets, kparams, &dom, dindex, &rc);
			if (!sess && rc == StromError_DataStoreOutOfRange)
			{
				/* keys without dense ids (float / numeric / sparse): hashed GROUP BY */
				rc = 0;
				sess = strom_gpupreagg_create_hashed(key, tg.data(), ntargets, kparams, hint, dindex, &rc);
			}
			if (!sess)
			{
				if (rc == 0)
					rc = StromError_HipInternal;
				break;
			}
			strom_task *fold = strom_submit_gpupreagg(sess, nullptr, src, krowmap, nullptr, nullptr, &rc);
			if (!fold)
				break;
			rc = strom_task_wait(fold, &pfm);
			if (rc == StromError_DataStoreOutOfRange && from_hint)
			{
				/* a key outside the remembered domain: measure this chunk's, once */
				from_hint = false;
				strom_gpupreagg_release(sess);
				sess = nullptr;
				rc = chunk_domain(key, prog, dev, tg.data(), ntargets, kparams, src, krowmap, &dom);
				if (rc != 0)
					break;
				domain_hint_put(key, dom);
				goto fold_again;
			}
			if (rc != 0)
				break;					/* CpuReCheck: the chunk goes back whole (gpupreagg.c:2746-2750) */
			/* (one call: a kds_dest that is too small answers DataStoreNoSpace by itself) */
			long	n = strom_gpupreagg_fetch(sess, kds_dest, dest_length);
			if (n < 0)
				rc = (int)-n;
		} while (0);
		if (sess)
			strom_gpupreagg_release(sess);
		if (tmp)
			strom_dstore_release(tmp);
		kgpreagg->status = rc;			/* KERN_GPUPREAGG_DMARECV: the status word */
		task->pfm = pfm;
		task->pfm.enabled = perfmon_enabled();
		/* nothing of its own on a stream (every step above was waited for): it completes on
		 * the completer thread with this status, without the drain of all streams a request
		 * that failed in mid-launch needs (task_fail) */
		task->errcode = rc;
		task_enqueue(task);
	});
	return task;
	STROM_ABI_CATCH(nullptr, p_errcode)
}
