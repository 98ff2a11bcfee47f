/*
 * datastore.cpp -- host-side chunk builders and converters
 *
 * Role in the reference: datastore.c (see strom_datastore.h).  Byte layout
 * of heap pages / tuples follows PostgreSQL 9.4's bufpage.h / htup_details.h
 * as re-declared for the device in opencl_common.h:156-264.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "strom_datastore.h"

namespace {

const size_t PAGE_HEADER = 24;			/* offsetof(PageHeaderData, pd_linp) */
const uint32_t LP_NORMAL = 1;

inline size_t maxalign(size_t v) { return STROM_TYPEALIGN(STROM_MAXIMUM_ALIGNOF, v); }

void
init_kds_head(kern_data_store *kds, int format, int ncols,
			  const strom_column_input *cols, uint32_t nrooms,
			  uint32_t maxblocks, size_t length)
{
	memset(kds, 0, KDS_HEAD_LENGTH(ncols));
	kds->hostptr = (hostptr_t)(uintptr_t)&kds->hostptr;
	kds->length = (cl_uint)length;
	kds->usage = 0;
	kds->ncols = ncols;
	kds->nitems = 0;
	kds->nrooms = nrooms;
	kds->nblocks = 0;
	kds->maxblocks = maxblocks;
	kds->format = (cl_char)format;
	kds->tdhasoid = 0;
	kds->tdtypeid = 2249;		/* RECORDOID */
	kds->tdtypmod = -1;

	/* attcacheoff: init_kern_data_store (datastore.c:336-379) */
	int		attcacheoff = (int)maxalign(HEAPTUPLE_HEADER_FIXED);
	for (int i = 0; i < ncols; i++)
	{
		kern_colmeta *cm = &kds->colmeta[i];
		if (attcacheoff > 0)
		{
			if (cols[i].attlen > 0)
				attcacheoff = (int)STROM_TYPEALIGN(cols[i].attalign, attcacheoff);
			else
				attcacheoff = -1;
		}
		cm->attbyval = cols[i].attbyval;
		/* datums given verbatim (attalign -1 on input) are int-aligned varlenas like text / numeric */
		cm->attalign = (cl_char)((cols[i].attlen == -1 && cols[i].attalign == -1) ? 4 : cols[i].attalign);
		cm->attlen = cols[i].attlen;
		cm->attnum = (cl_short)(i + 1);
		cm->attcacheoff = (cl_short)attcacheoff;
		if (attcacheoff >= 0)
			attcacheoff += cols[i].attlen;
	}
}

inline bool
is_null(const strom_column_input &c, uint32_t row)
{
	return c.isnull != nullptr && c.isnull[row] != 0;
}

/*
 * NUMERIC as PostgreSQL stores it in a heap tuple (utils/adt/numeric.c of
 * 9.4; decoded on the device by strom_numeric_from_varlena): a varlena with
 * a 1-byte header (these values are short), then the "short" layout
 *   uint16 n_header = 0x8000 | sign 0x2000 | dscale << 7 | weight & 0x7F
 * or, when weight / dscale do not fit, the "long" one
 *   uint16 n_sign_dscale (0x4000 = negative) ; int16 n_weight
 * and base-10000 digits, most significant first.  Input is the 64-bit
 * device image (6-bit exponent, sign, 57-bit mantissa).
 */
inline bool
is_varlena_numeric(const strom_column_input &c)
{
	return c.attlen == -1 && c.type_oid == STROM_NUMERICOID;
}

/*
 * ... or given as the datum itself: attalign == -1 says that 'values' is an
 * array of pointers to complete varlena datums (1- or 4-byte header, length in
 * the header) which go into the heap tuple verbatim -- numerics of any
 * magnitude, also those the 64-bit device form cannot hold (the kernels then
 * answer CpuReCheck for the row, opencl_numeric.h:166-307).
 */
inline bool
is_varlena_raw(const strom_column_input &c)
{
	return c.attlen == -1 && c.attalign == -1;
}

inline size_t
varlena_raw_size(const unsigned char *p)
{
	if (p[0] == 0x01)
		return 2 + (size_t)(p[1] == 18 ? 16 : 8);	/* external TOAST pointer: tag + payload by tag */
	if (p[0] & 0x01)
		return (size_t)(p[0] >> 1);					/* 1-byte header: total length */
	uint32_t	h;
	memcpy(&h, p, 4);
	return (size_t)(h >> 2);						/* 4-byte header, uncompressed */
}

size_t
numeric_image_to_varlena(uint64_t image, unsigned char *out)
{
	int			expo = (int)((int64_t)image >> 58);
	bool		sign = (image & (1UL << 57)) != 0;
	unsigned __int128 mant = image & ((1UL << 57) - 1);
	std::vector<uint16_t> digits;			/* least significant first */
	int			weight = 0, dscale = 0;

	if (mant != 0)
	{
		int		e4 = (expo >= 0 ? expo / 4 : -((-expo + 3) / 4));	/* floor(expo / 4) */
		int		r = expo - 4 * e4;									/* 0..3 */
		for (int i = 0; i < r; i++)
			mant *= 10;
		while (mant != 0)
		{
			digits.push_back((uint16_t)(mant % 10000));
			mant /= 10000;
		}
		size_t	strip = 0;
		while (strip < digits.size() && digits[strip] == 0)
			strip++;						/* PostgreSQL strips trailing zero digits */
		digits.erase(digits.begin(), digits.begin() + strip);
		e4 += (int)strip;
		weight = (int)digits.size() - 1 + e4;
		dscale = (expo < 0 ? -expo : 0);
	}
	else
		sign = false;
	bool		is_short = (weight >= -64 && weight <= 63 && dscale <= 63);
	size_t		body = (is_short ? 2 : 4) + 2 * digits.size();
	size_t		total = 1 + body;
	unsigned char *p = out;

	*p++ = (unsigned char)((total << 1) | 0x01);				/* 1-byte varlena header */
	if (is_short)
	{
		uint16_t h = (uint16_t)(0x8000 | (sign ? 0x2000 : 0) | (dscale << 7) |
								(weight < 0 ? 0x0040 : 0) | (weight & 0x003F));
		memcpy(p, &h, 2); p += 2;
	}
	else
	{
		uint16_t h = (uint16_t)((sign ? 0x4000 : 0) | (dscale & 0x3FFF));
		int16_t	 w = (int16_t)weight;
		memcpy(p, &h, 2); p += 2;
		memcpy(p, &w, 2); p += 2;
	}
	for (size_t i = digits.size(); i-- > 0; )
	{
		memcpy(p, &digits[i], 2);
		p += 2;
	}
	return total;
}

/* size of the heap tuple for 'row': header, optional bitmap, aligned data */
size_t
heap_tuple_size(int ncols, const strom_column_input *cols, uint32_t row, size_t *p_hoff)
{
	bool	hasnull = false;
	for (int i = 0; i < ncols; i++)
		if (is_null(cols[i], row))
			hasnull = true;
	size_t	hoff = HEAPTUPLE_HEADER_FIXED;
	if (hasnull)
		hoff += (ncols + 7) / 8;
	hoff = maxalign(hoff);
	size_t	off = hoff;
	for (int i = 0; i < ncols; i++)
	{
		if (is_null(cols[i], row))
			continue;
		if (is_varlena_raw(cols[i]))
		{
			const unsigned char *datum = ((const unsigned char *const *)cols[i].values)[row];
			if (!(datum[0] & 0x01))
				off = STROM_TYPEALIGN(4, off);		/* a 4-byte header is int-aligned */
			off += varlena_raw_size(datum);
			continue;
		}
		if (is_varlena_numeric(cols[i]))
		{
			/* short varlena header: no alignment padding */
			unsigned char tmp[64];
			off += numeric_image_to_varlena(((const uint64_t *)cols[i].values)[row], tmp);
			continue;
		}
		off = STROM_TYPEALIGN(cols[i].attalign, off);
		off += cols[i].attlen;
	}
	*p_hoff = hoff;
	return off;
}

void
heap_tuple_form(char *dest, int ncols, const strom_column_input *cols, uint32_t row,
				size_t t_len, size_t hoff, uint32_t blkno, uint16_t posid)
{
	HeapTupleHeaderData *htup = (HeapTupleHeaderData *)dest;
	bool	hasnull = false;

	memset(dest, 0, t_len);
	for (int i = 0; i < ncols; i++)
		if (is_null(cols[i], row))
			hasnull = true;
	htup->t_xmin = 2;							/* FrozenTransactionId */
	htup->bi_hi = (cl_ushort)(blkno >> 16);
	htup->bi_lo = (cl_ushort)(blkno & 0xffff);
	htup->ip_posid = posid;
	htup->t_infomask2 = (cl_ushort)(ncols & HEAP_NATTS_MASK);
	htup->t_infomask = (cl_ushort)((hasnull ? HEAP_HASNULL : 0) | 0x0800 /* XMAX_INVALID */
								   | 0x0100 /* XMIN_COMMITTED */);
	htup->t_hoff = (cl_uchar)hoff;
	size_t	off = hoff;
	bool	hasvarwidth = false;
	for (int i = 0; i < ncols; i++)
	{
		if (is_null(cols[i], row))
			continue;
		if (hasnull)
			htup->t_bits[i >> 3] |= (cl_uchar)(1 << (i & 7));
		if (is_varlena_raw(cols[i]))
		{
			const unsigned char *datum = ((const unsigned char *const *)cols[i].values)[row];
			if (!(datum[0] & 0x01))
				off = STROM_TYPEALIGN(4, off);
			memcpy(dest + off, datum, varlena_raw_size(datum));
			off += varlena_raw_size(datum);
			hasvarwidth = true;
			continue;
		}
		if (is_varlena_numeric(cols[i]))
		{
			off += numeric_image_to_varlena(((const uint64_t *)cols[i].values)[row],
											(unsigned char *)dest + off);
			hasvarwidth = true;
			continue;
		}
		off = STROM_TYPEALIGN(cols[i].attalign, off);
		memcpy(dest + off, (const char *)cols[i].values + (size_t)cols[i].attlen * row,
			   cols[i].attlen);
		off += cols[i].attlen;
	}
	if (hasvarwidth)
		htup->t_infomask |= 0x0002;				/* HEAP_HASVARWIDTH */
}

/* ---- ROW: how many pages do nrows rows need ------------------------- */
uint32_t
row_count_pages(int ncols, const strom_column_input *cols, uint32_t nrows)
{
	uint32_t npages = 0;
	size_t	lower = PAGE_HEADER, upper = BLCKSZ;
	bool	open = false;
	for (uint32_t r = 0; r < nrows; r++)
	{
		size_t hoff, t_len = heap_tuple_size(ncols, cols, r, &hoff);
		size_t need = maxalign(t_len) + sizeof(cl_uint);
		if (!open || upper - lower < need)
		{
			npages++;
			lower = PAGE_HEADER;
			upper = BLCKSZ;
			open = true;
		}
		lower += sizeof(cl_uint);
		upper -= maxalign(t_len);
	}
	return npages;
}

bool
cols_valid(int ncols, const strom_column_input *cols, int format = 0)
{
	if (ncols < 1 || !cols)
		return false;
	for (int i = 0; i < ncols; i++)
	{
		int l = cols[i].attlen;
		if (!cols[i].values)
			return false;
		if (is_varlena_numeric(cols[i]) || is_varlena_raw(cols[i]))
		{
			/* heap tuples; a COLUMN chunk carries datums given verbatim in its heap area
			 * (strom_kds.h) and numerics in the 8-byte form; TUPSLOT by-value datums only */
			if (format == KDS_FORMAT_COLUMN && is_varlena_raw(cols[i]))
				continue;
			if (format != KDS_FORMAT_ROW && format != KDS_FORMAT_ROW_FLAT)
				return false;
			continue;
		}
		if (!(l == 1 || l == 2 || l == 4 || l == 8))
			return false;
	}
	return true;
}

/* where a datum of a COLUMN chunk's heap area goes: 4-byte headers on 4-byte boundaries */
inline size_t
column_heap_place(size_t off, const unsigned char *datum)
{
	return (datum[0] & 0x01) ? off : STROM_TYPEALIGN(4, off);
}

size_t
column_layout(int ncols, const strom_column_input *cols, uint32_t nrooms,
			  std::vector<size_t> *values_off, std::vector<size_t> *nulls_off,
			  std::vector<size_t> *extra_off = nullptr)
{
	size_t	off = KDS_COLUMN_HEAD_LENGTH(ncols);
	for (int i = 0; i < ncols; i++)
	{
		if (values_off)
			(*values_off)[i] = off;
		off += KDS_COLUMN_VALUES_LENGTH(cols[i].attlen, nrooms);
		bool anynull = false;
		if (cols[i].isnull)
			for (uint32_t r = 0; r < nrooms && !anynull; r++)
				anynull = (cols[i].isnull[r] != 0);
		if (nulls_off)
			(*nulls_off)[i] = anynull ? off : 0;
		if (anynull)
			off += KDS_COLUMN_NULLS_LENGTH(nrooms);
	}
	/* the heap area: the varlena columns' datums, column by column, row by row */
	for (int i = 0; i < ncols; i++)
	{
		if (extra_off)
			(*extra_off)[i] = 0;
		if (!is_varlena_raw(cols[i]))
			continue;
		if (extra_off)
			(*extra_off)[i] = off;
		for (uint32_t r = 0; r < nrooms; r++)
		{
			if (is_null(cols[i], r))
				continue;
			const unsigned char *d = ((const unsigned char *const *)cols[i].values)[r];
			off = column_heap_place(off, d) + varlena_raw_size(d);
		}
		off = STROM_TYPEALIGN(KDS_COLUMN_ALIGN, off);
	}
	return off;
}

/* a 64-bit numeric image's value rounded away from zero to an integer (KDS_COLSTAT_INTPART);
 * false when that does not fit int64 */
bool
numeric_image_outward(uint64_t image, int64_t *p_out)
{
	int			expo = (int)((int64_t)image >> 58);
	bool		sign = ((image >> 57) & 1) != 0;
	unsigned __int128 m = image & ((1ULL << 57) - 1);

	if (expo >= 0)
	{
		for (int i = 0; i < expo && m < ((unsigned __int128)1 << 64); i++)
			m *= 10;
	}
	else
	{
		unsigned __int128 d = 1;
		for (int i = 0; i < -expo && d < ((unsigned __int128)1 << 64); i++)
			d *= 10;
		m = (m + d - 1) / d;
	}
	if (m > (unsigned __int128)INT64_MAX)
		return false;
	*p_out = (sign ? -(int64_t)m : (int64_t)m);
	return true;
}

void
column_minmax(const strom_column_input &c, uint32_t nrows, kern_coldir *cd)
{
	bool	isfloat = (c.type_oid == STROM_FLOAT4OID || c.type_oid == STROM_FLOAT8OID);
	bool	any = false;

	cd->stat_flags = 0;
	if (c.type_oid == 0)
		return;				/* type unknown (converted chunk): no zone map */
	if (c.type_oid == STROM_NUMERICOID)
	{
		/* 64-bit images: bounds of the values' integer parts, outward (strom_kds.h) */
		int64_t	lo = 0, hi = 0;
		if (c.attlen != 8)
			return;
		for (uint32_t r = 0; r < nrows; r++)
		{
			if (is_null(c, r))
				continue;
			uint64_t	image;
			int64_t		v;
			memcpy(&image, (const char *)c.values + 8 * (size_t)r, 8);
			if (!numeric_image_outward(image, &v))
				return;			/* a value beyond int64: no bound to give */
			if (!any || v < lo) lo = v;
			if (!any || v > hi) hi = v;
			any = true;
		}
		if (any)
		{
			cd->stat_flags = KDS_COLSTAT_INTPART;
			cd->minval = lo;
			cd->maxval = hi;
		}
		return;
	}
	int64_t	imin = 0, imax = 0;
	double	fmin = 0, fmax = 0;

	for (uint32_t r = 0; r < nrows; r++)
	{
		if (is_null(c, r))
			continue;
		const char *p = (const char *)c.values + (size_t)c.attlen * r;
		if (isfloat)
		{
			double v;
			if (c.attlen == 4) { float f; memcpy(&f, p, 4); v = f; }
			else memcpy(&v, p, 8);
			if (std::isnan(v))
				continue;
			if (!any || v < fmin) fmin = v;
			if (!any || v > fmax) fmax = v;
		}
		else
		{
			int64_t v;
			switch (c.attlen)
			{
				case 1: { int8_t x; memcpy(&x, p, 1); v = x; } break;
				case 2: { int16_t x; memcpy(&x, p, 2); v = x; } break;
				case 4: { int32_t x; memcpy(&x, p, 4); v = x; } break;
				default: memcpy(&v, p, 8); break;
			}
			if (!any || v < imin) imin = v;
			if (!any || v > imax) imax = v;
		}
		any = true;
	}
	cd->stat_flags = 0;
	if (any)
	{
		cd->stat_flags = KDS_COLSTAT_MINMAX | (isfloat ? KDS_COLSTAT_ISFLOAT : 0);
		if (isfloat)
		{
			memcpy(&cd->minval, &fmin, 8);
			memcpy(&cd->maxval, &fmax, 8);
		}
		else
		{
			cd->minval = imin;
			cd->maxval = imax;
		}
	}
}

}	/* namespace */

extern "C" size_t
strom_kds_required_length(int format, int ncols, const strom_column_input *cols, uint32_t nrows)
{
	if (!cols_valid(ncols, cols, format))
		return 0;
	switch (format)
	{
		case KDS_FORMAT_ROW:
			{
				uint32_t npages = row_count_pages(ncols, cols, nrows);
				size_t	head = KDS_HEAD_LENGTH(ncols)
					+ STROMALIGN(sizeof(kern_blkitem) * (size_t)npages)
					+ STROMALIGN(sizeof(kern_rowitem) * (size_t)nrows);
				return STROM_TYPEALIGN(BLCKSZ, head) + (size_t)BLCKSZ * npages;
			}
		case KDS_FORMAT_ROW_FLAT:
			{
				size_t	len = KDS_HEAD_LENGTH(ncols) + STROMALIGN(sizeof(kern_rowitem) * (size_t)nrows);
				for (uint32_t r = 0; r < nrows; r++)
				{
					size_t hoff;
					len += STROM_LONGALIGN(heap_tuple_size(ncols, cols, r, &hoff));
				}
				return STROMALIGN(len);
			}
		case KDS_FORMAT_TUPSLOT:
			return STROMALIGN(KDS_HEAD_LENGTH(ncols) + KDS_TUPSLOT_STRIDE(ncols) * (size_t)nrows);
		case KDS_FORMAT_COLUMN:
			return column_layout(ncols, cols, nrows, nullptr, nullptr);
	}
	return 0;
}

extern "C" int
strom_kds_build(int format, int ncols, const strom_column_input *cols,
				uint32_t nrows, void *buffer, size_t buflen)
{
	size_t	required = strom_kds_required_length(format, ncols, cols, nrows);
	if (required == 0 || !buffer || ((uintptr_t)buffer & 15) != 0)
		return StromError_BadRequestMessage;
	if (buflen < required)
		return StromError_DataStoreNoSpace;
	if (required > 0xffffffffUL)
		return StromError_DataStoreOutOfRange;		/* 'length' is 32 bit */
	kern_data_store *kds = (kern_data_store *)buffer;
	char   *base = (char *)buffer;

	if (format == KDS_FORMAT_ROW)
	{
		uint32_t npages = row_count_pages(ncols, cols, nrows);
		if (npages > 0xffff)
			return StromError_DataStoreOutOfRange;	/* blk_index is 16 bit */
		init_kds_head(kds, format, ncols, cols, nrows, npages, required);
		kds->nitems = nrows;
		kds->nblocks = npages;
		memset(base + KDS_HEAD_LENGTH(ncols), 0,
			   KERN_DATA_STORE_ROWBLOCK_OFFSET(kds) - KDS_HEAD_LENGTH(ncols));
		int32_t	blk = -1;
		char   *page = nullptr;
		size_t	lower = 0, upper = 0;
		for (uint32_t r = 0; r < nrows; r++)
		{
			size_t hoff, t_len = heap_tuple_size(ncols, cols, r, &hoff);
			size_t need = maxalign(t_len) + sizeof(cl_uint);
			if (blk < 0 || upper - lower < need)
			{
				if (page)
				{
					*(cl_ushort *)(page + 12) = (cl_ushort)lower;
					*(cl_ushort *)(page + 14) = (cl_ushort)upper;
				}
				blk++;
				page = KERN_DATA_STORE_ROWBLOCK(kds, blk);
				memset(page, 0, BLCKSZ);
				*(cl_ushort *)(page + 16) = BLCKSZ;			/* pd_special */
				*(cl_ushort *)(page + 18) = BLCKSZ | 4;		/* pagesize | layout version */
				lower = PAGE_HEADER;
				upper = BLCKSZ;
				kern_blkitem *bitem = KERN_DATA_STORE_BLKITEM(kds, blk);
				bitem->buffer = 0;							/* InvalidBuffer */
				bitem->page = (hostptr_t)(uintptr_t)page;
			}
			upper -= maxalign(t_len);
			uint32_t linenum = (uint32_t)((lower - PAGE_HEADER) / sizeof(cl_uint)) + 1;
			heap_tuple_form(page + upper, ncols, cols, r, t_len, hoff, blk, (uint16_t)linenum);
			cl_uint itemid = (cl_uint)upper | (LP_NORMAL << 15) | ((cl_uint)t_len << 17);
			memcpy(page + lower, &itemid, sizeof(itemid));
			lower += sizeof(cl_uint);
			kern_rowitem *ritem = KERN_DATA_STORE_ROWITEM(kds, r);
			ritem->blk_index = (cl_ushort)blk;
			ritem->item_offset = (cl_ushort)linenum;
		}
		if (page)
		{
			*(cl_ushort *)(page + 12) = (cl_ushort)lower;
			*(cl_ushort *)(page + 14) = (cl_ushort)upper;
		}
		return 0;
	}
	if (format == KDS_FORMAT_ROW_FLAT)
	{
		init_kds_head(kds, format, ncols, cols, nrows, 0, required);
		size_t	tail = required;
		for (uint32_t r = 0; r < nrows; r++)
		{
			size_t hoff, t_len = heap_tuple_size(ncols, cols, r, &hoff);
			tail -= STROM_LONGALIGN(t_len);
			heap_tuple_form(base + tail, ncols, cols, r, t_len, hoff, 0, (uint16_t)(r & 0xffff));
			KERN_DATA_STORE_ROWITEM(kds, r)->htup_offset = (cl_uint)tail;
		}
		kds->nitems = nrows;
		kds->usage = (cl_uint)(required - tail);
		return 0;
	}
	if (format == KDS_FORMAT_TUPSLOT)
	{
		init_kds_head(kds, format, ncols, cols, nrows, 0, required);
		for (uint32_t r = 0; r < nrows; r++)
		{
			Datum	   *values = KERN_DATA_STORE_VALUES(kds, r);
			cl_char	   *isnull = KERN_DATA_STORE_ISNULL(kds, r);
			memset(values, 0, KDS_TUPSLOT_STRIDE(ncols));
			for (int i = 0; i < ncols; i++)
			{
				if (is_null(cols[i], r))
					isnull[i] = 1;
				else
					memcpy(&values[i], (const char *)cols[i].values + (size_t)cols[i].attlen * r,
						   cols[i].attlen);
			}
		}
		kds->nitems = nrows;
		return 0;
	}
	if (format == KDS_FORMAT_COLUMN)
	{
		std::vector<size_t> voff(ncols), noff(ncols), xoff(ncols);
		column_layout(ncols, cols, nrows, &voff, &noff, &xoff);
		init_kds_head(kds, format, ncols, cols, nrows, 0, required);
		kern_coldir *cd = KERN_DATA_STORE_COLDIR(kds);
		memset(cd, 0, KDS_COLUMN_HEAD_LENGTH(ncols) - KDS_HEAD_LENGTH(ncols));
		for (int i = 0; i < ncols; i++)
		{
			size_t	vbytes = (size_t)cols[i].attlen * nrows;
			cd[i].values_off = (cl_uint)voff[i];
			cd[i].nulls_off = (cl_uint)noff[i];
			cd[i].extra_off = (cl_uint)xoff[i];
			if (is_varlena_raw(cols[i]))
			{
				/* offsets of the datums (0 = NULL), the datums behind the column arrays */
				cl_ulong   *offs = (cl_ulong *)(base + voff[i]);
				size_t		off = xoff[i];
				memset(offs, 0, KDS_COLUMN_VALUES_LENGTH(-1, nrows));
				if (noff[i])
					memset(base + noff[i], 0, KDS_COLUMN_NULLS_LENGTH(nrows));
				for (uint32_t r = 0; r < nrows; r++)
				{
					if (is_null(cols[i], r))
						continue;
					const unsigned char *d = ((const unsigned char *const *)cols[i].values)[r];
					size_t	len = varlena_raw_size(d);
					size_t	at = column_heap_place(off, d);
					memset(base + off, 0, at - off);
					memcpy(base + at, d, len);
					offs[r] = at;
					off = at + len;
					if (noff[i])
						((cl_uint *)(base + noff[i]))[r >> 5] |= (1u << (r & 31));
				}
				memset(base + off, 0, STROM_TYPEALIGN(KDS_COLUMN_ALIGN, off) - off);
				kds->usage = (cl_uint)off;
				continue;
			}
			memcpy(base + voff[i], cols[i].values, vbytes);
			memset(base + voff[i] + vbytes, 0,
				   KDS_COLUMN_VALUES_LENGTH(cols[i].attlen, nrows) - vbytes);
			if (noff[i])
			{
				cl_uint *nn = (cl_uint *)(base + noff[i]);
				memset(nn, 0, KDS_COLUMN_NULLS_LENGTH(nrows));
				for (uint32_t r = 0; r < nrows; r++)
					if (!cols[i].isnull[r])
						nn[r >> 5] |= (1u << (r & 31));
				/* a NULL slot holds zero so that reads are deterministic */
				for (uint32_t r = 0; r < nrows; r++)
					if (cols[i].isnull[r])
						memset(base + voff[i] + (size_t)cols[i].attlen * r, 0, cols[i].attlen);
			}
			column_minmax(cols[i], nrows, &cd[i]);
		}
		kds->nitems = nrows;
		return 0;
	}
	return StromError_BadRequestMessage;
}

/*
 * Head only of a NULL-free COLUMN chunk: the caller fills the column arrays
 * itself, typically on the device (values of column i start at the returned
 * chunk offset values_off[i]), and hands the buffer to strom_dstore_wrap().
 */
extern "C" size_t
strom_kds_column_head(int ncols, const strom_column_input *cols, uint32_t nrows,
					  const int64_t *minmax, void *head, size_t headlen, uint32_t *values_off)
{
	if (ncols < 1 || ncols > 1600 || !cols || !head || ((uintptr_t)head & 15) != 0 ||
		headlen < KDS_COLUMN_HEAD_LENGTH(ncols))
		return 0;
	for (int i = 0; i < ncols; i++)
		if (!(cols[i].attlen == 1 || cols[i].attlen == 2 || cols[i].attlen == 4 || cols[i].attlen == 8))
			return 0;
	std::vector<strom_column_input> nonull(cols, cols + ncols);
	for (auto &c : nonull)
		c.isnull = nullptr;
	std::vector<size_t> voff(ncols), noff(ncols);
	size_t	required = column_layout(ncols, nonull.data(), nrows, &voff, &noff);
	if (required > 0xffffffffUL)
		return 0;								/* 'length' is 32 bit */
	kern_data_store *kds = (kern_data_store *)head;
	init_kds_head(kds, KDS_FORMAT_COLUMN, ncols, nonull.data(), nrows, 0, required);
	kern_coldir *cd = KERN_DATA_STORE_COLDIR(kds);
	memset(cd, 0, KDS_COLUMN_HEAD_LENGTH(ncols) - KDS_HEAD_LENGTH(ncols));
	for (int i = 0; i < ncols; i++)
	{
		bool	isfloat = (cols[i].type_oid == STROM_FLOAT4OID || cols[i].type_oid == STROM_FLOAT8OID);
		cd[i].values_off = (cl_uint)voff[i];
		if (values_off)
			values_off[i] = (uint32_t)voff[i];
		/* (min > max, as integers: "no zone map for this column" -- e.g. 64-bit numeric images,
		 * whose bit patterns do not order like their values) */
		if (minmax && cols[i].type_oid != 0 && nrows > 0 && (isfloat || minmax[2 * i] <= minmax[2 * i + 1]))
		{
			/* (a numeric column: the caller's bounds are of the values' integer parts, outward) */
			cd[i].stat_flags = (cols[i].type_oid == STROM_NUMERICOID ? KDS_COLSTAT_INTPART
								: KDS_COLSTAT_MINMAX | (isfloat ? KDS_COLSTAT_ISFLOAT : 0));
			cd[i].minval = minmax[2 * i];
			cd[i].maxval = minmax[2 * i + 1];
		}
	}
	kds->nitems = nrows;
	return required;
}

/* ---- host-side datum fetch -------------------------------------------- */
namespace {

const char *
host_get_datum_tuple(const kern_colmeta *colmeta, const HeapTupleHeaderData *htup, uint32_t colidx)
{
	bool	hasnull = (htup->t_infomask & HEAP_HASNULL) != 0;
	uint32_t natts = htup->t_infomask2 & HEAP_NATTS_MASK;
	size_t	off = htup->t_hoff;

	if (colidx >= natts)
		return nullptr;
	for (uint32_t i = 0; i < natts; i++)
	{
		if (hasnull && !(htup->t_bits[i >> 3] & (1 << (i & 7))))
		{
			if (i == colidx)
				return nullptr;
			continue;
		}
		if (colmeta[i].attlen <= 0)
		{
			/* varlena: padding only in front of a 4-byte header */
			const unsigned char *p = (const unsigned char *)htup + off;
			if (p[0] == 0)
			{
				off = STROM_TYPEALIGN(colmeta[i].attalign, off);
				p = (const unsigned char *)htup + off;
			}
			if (i == colidx)
				return (const char *)p;
			if (p[0] == 0x01)
				return nullptr;				/* external TOAST pointer: not on this path */
			off += ((p[0] & 0x01) ? (size_t)((p[0] >> 1) & 0x7f)
					: (size_t)((((uint32_t)p[0]) | ((uint32_t)p[1] << 8) |
								((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)) >> 2));
			continue;
		}
		off = STROM_TYPEALIGN(colmeta[i].attalign, off);
		if (i == colidx)
			return (const char *)htup + off;
		off += colmeta[i].attlen;
	}
	return nullptr;
}

const char *
host_get_datum(const kern_data_store *kds, uint32_t rowidx, uint32_t colidx)
{
	if (colidx >= kds->ncols || rowidx >= kds->nitems)
		return nullptr;
	switch (kds->format)
	{
		case KDS_FORMAT_ROW:
			{
				const kern_rowitem *ri = KERN_DATA_STORE_ROWITEM(kds, rowidx);
				if (ri->blk_index >= kds->nblocks)
					return nullptr;
				const char *page = KERN_DATA_STORE_ROWBLOCK(kds, ri->blk_index);
				cl_uint itemid;
				memcpy(&itemid, page + PAGE_HEADER + sizeof(cl_uint) * (ri->item_offset - 1), 4);
				return host_get_datum_tuple(kds->colmeta,
											(const HeapTupleHeaderData *)(page + (itemid & 0x7fff)),
											colidx);
			}
		case KDS_FORMAT_ROW_FLAT:
			return host_get_datum_tuple(kds->colmeta,
										(const HeapTupleHeaderData *)
										((const char *)kds + KERN_DATA_STORE_ROWITEM(kds, rowidx)->htup_offset),
										colidx);
		case KDS_FORMAT_TUPSLOT:
			if (KERN_DATA_STORE_ISNULL(kds, rowidx)[colidx])
				return nullptr;
			return (const char *)(KERN_DATA_STORE_VALUES(kds, rowidx) + colidx);
		case KDS_FORMAT_COLUMN:
			{
				const kern_coldir *cd = KERN_DATA_STORE_COLDIR(kds) + colidx;
				if (cd->nulls_off)
				{
					const cl_uint *nn = (const cl_uint *)((const char *)kds + cd->nulls_off);
					if (!((nn[rowidx >> 5] >> (rowidx & 31)) & 1))
						return nullptr;
				}
				if (kds->colmeta[colidx].attlen < 0)
				{
					/* a varlena column: the offset of the row's datum (strom_kds.h) */
					cl_ulong off = ((const cl_ulong *)((const char *)kds + cd->values_off))[rowidx];
					if (off == 0 || off >= kds->length)
						return nullptr;
					return (const char *)kds + off;
				}
				return (const char *)kds + cd->values_off +
					(size_t)kds->colmeta[colidx].attlen * rowidx;
			}
	}
	return nullptr;
}

}	/* namespace */

extern "C" int
strom_kds_fetch(const kern_data_store *kds, uint32_t rowidx, uint32_t colidx, uint64_t *value)
{
	const char *p = host_get_datum(kds, rowidx, colidx);
	if (!p)
		return 1;
	*value = 0;
	int		attlen = kds->colmeta[colidx].attlen;
	if (attlen > 0)
		memcpy(value, p, attlen);
	else
	{
		/* varlena: the first bytes of the datum (header included), at most 8 */
		const unsigned char *v = (const unsigned char *)p;
		size_t	len = ((v[0] & 0x01) ? (size_t)((v[0] >> 1) & 0x7f) : 4);
		memcpy(value, p, len < 8 ? len : 8);
	}
	return 0;
}

extern "C" size_t
strom_kds_to_column(const kern_data_store *src, void *dst, size_t dstlen)
{
	uint32_t nrows = src->nitems;
	int		ncols = (int)src->ncols;
	std::vector<std::vector<char>> values(ncols);
	std::vector<std::vector<uint8_t>> nulls(ncols);
	std::vector<strom_column_input> cols(ncols);

	for (int i = 0; i < ncols; i++)
	{
		int	attlen = src->colmeta[i].attlen;
		/* varlena columns (heap formats hold their datums): the datums go into the COLUMN
		 * chunk's heap area verbatim */
		bool	varlena = (attlen == -1 && (src->format == KDS_FORMAT_ROW || src->format == KDS_FORMAT_ROW_FLAT ||
										   src->format == KDS_FORMAT_COLUMN));
		if (!(attlen == 1 || attlen == 2 || attlen == 4 || attlen == 8 || varlena))
			return 0;
		values[i].assign(KDS_COLUMN_ATTWIDTH(attlen) * nrows + 8, 0);
		nulls[i].assign(nrows + 1, 0);
	}
	for (uint32_t r = 0; r < nrows; r++)
		for (int i = 0; i < ncols; i++)
		{
			const char *p = host_get_datum(src, r, i);
			if (!p)
				nulls[i][r] = 1;
			else if (src->colmeta[i].attlen < 0)
				((const char **)values[i].data())[r] = p;
			else
				memcpy(&values[i][(size_t)src->colmeta[i].attlen * r], p, src->colmeta[i].attlen);
		}
	for (int i = 0; i < ncols; i++)
	{
		cols[i].type_oid = 0;
		cols[i].attlen = src->colmeta[i].attlen;
		cols[i].attalign = (src->colmeta[i].attlen < 0 ? -1 : src->colmeta[i].attalign);
		cols[i].attbyval = src->colmeta[i].attbyval;
		cols[i].values = values[i].data();
		cols[i].isnull = nulls[i].data();
	}
	size_t	required = strom_kds_required_length(KDS_FORMAT_COLUMN, ncols, cols.data(), nrows);
	if (!dst)
		return required;
	if (strom_kds_build(KDS_FORMAT_COLUMN, ncols, cols.data(), nrows, dst, dstlen) != 0)
		return 0;
	return required;
}

/* ====================================================================== *
 * kern_multihash builder: multihash_preload_khashtable
 * (gpuhashjoin.c:3614-3816).  Every inner row becomes a kern_hashentry that
 * owns the whole heap tuple; hash = PostgreSQL 9.4 pg_crc32 over the datum
 * images of the non-NULL key columns; entries are pushed on
 * hash_slots[hash % nslots].
 * ====================================================================== */
namespace {

uint32_t	crc_table[256];
bool		crc_ready = false;

void
crc_init(void)
{
	if (crc_ready)
		return;
	for (uint32_t i = 0; i < 256; i++)
	{
		uint32_t c = i;
		for (int j = 0; j < 8; j++)
			c = (c & 1) ? (0xEDB88320U ^ (c >> 1)) : (c >> 1);
		crc_table[i] = c;
	}
	crc_ready = true;
}

/* COMP_CRC32 of PostgreSQL 9.4: reflected table, MSB-first update */
uint32_t
legacy_crc32(uint32_t crc, const void *data, size_t len)
{
	const unsigned char *p = (const unsigned char *)data;
	while (len-- > 0)
		crc = crc_table[((crc >> 24) ^ *p++) & 0xFF] ^ (crc << 8);
	return crc;
}

/* columns of one row of any kds, as strom_column_input over 1-row buffers */
struct row_image {
	std::vector<uint64_t>	values;
	std::vector<uint8_t>	nulls;
	std::vector<strom_column_input> cols;
};

void
row_image_init(row_image &ri, const kern_data_store *kds)
{
	int ncols = (int)kds->ncols;
	ri.values.assign(ncols, 0);
	ri.nulls.assign(ncols, 0);
	ri.cols.resize(ncols);
	for (int i = 0; i < ncols; i++)
	{
		ri.cols[i].type_oid = 0;
		ri.cols[i].attlen = kds->colmeta[i].attlen;
		ri.cols[i].attalign = kds->colmeta[i].attalign;
		ri.cols[i].attbyval = kds->colmeta[i].attbyval;
		ri.cols[i].values = &ri.values[i];
		ri.cols[i].isnull = &ri.nulls[i];
		/* a varlena column (text, character(n), heap-form numeric): values[i] holds the
		 * ADDRESS of the row's datum, which goes into the entry's tuple verbatim -- the
		 * reference copies whole inner heap tuples (gpuhashjoin.c:3717-3805) */
		if (kds->colmeta[i].attlen < 0)
			ri.cols[i].attalign = -1;
	}
}

void
row_image_load(row_image &ri, const kern_data_store *kds, uint32_t row)
{
	for (uint32_t i = 0; i < kds->ncols; i++)
	{
		const char *p = host_get_datum(kds, row, i);
		ri.nulls[i] = (p == nullptr);
		ri.values[i] = 0;
		if (p && kds->colmeta[i].attlen < 0)
			ri.values[i] = (uint64_t)(uintptr_t)p;
		else if (p)
			memcpy(&ri.values[i], p, kds->colmeta[i].attlen);
	}
}

size_t
hashtable_head_length(uint32_t ncols, uint32_t nslots)
{
	return STROM_LONGALIGN(STROM_LONGALIGN(offsetof(kern_hashtable, colmeta) +
										   sizeof(kern_colmeta) * ncols) +
						   sizeof(cl_uint) * (size_t)nslots);
}

uint32_t
hashtable_nslots(uint32_t ntuples)
{
	/* the planner sizes nslots from its row estimate (gpuhashjoin.c:366,958) */
	return (uint32_t)((double)ntuples * 1.15) + 1;
}

}	/* namespace */

extern "C" size_t
strom_multihash_required_length(int ntables, const strom_hashtable_input *tables)
{
	if (ntables < 1 || !tables)
		return 0;
	size_t	len = STROM_LONGALIGN(offsetof(kern_multihash, htable_offset) + sizeof(cl_uint) * ntables);
	for (int t = 0; t < ntables; t++)
	{
		const kern_data_store *kds = tables[t].inner;
		if (!kds || tables[t].nkeys < 1 || tables[t].nkeys > 8)
			return 0;
		for (uint32_t c = 0; c < kds->ncols; c++)
		{
			int l = kds->colmeta[c].attlen;
			/* varlena columns: the formats that hold their datums (heap tuples, a COLUMN
			 * chunk's heap area) */
			if (l == -1 && (kds->format == KDS_FORMAT_ROW || kds->format == KDS_FORMAT_ROW_FLAT ||
							kds->format == KDS_FORMAT_COLUMN))
				continue;
			if (!(l == 1 || l == 2 || l == 4 || l == 8))
				return 0;
		}
		len += hashtable_head_length(kds->ncols, hashtable_nslots(kds->nitems));
		row_image ri;
		row_image_init(ri, kds);
		for (uint32_t r = 0; r < kds->nitems; r++)
		{
			size_t hoff;
			row_image_load(ri, kds, r);
			len += KERN_HASHENTRY_SIZE_BY_TLEN(heap_tuple_size((int)kds->ncols, ri.cols.data(), 0, &hoff));
		}
	}
	return STROMALIGN(len);
}

extern "C" int
strom_multihash_build(int ntables, const strom_hashtable_input *tables, void *buffer, size_t buflen)
{
	size_t	required = strom_multihash_required_length(ntables, tables);
	if (required == 0 || !buffer || ((uintptr_t)buffer & 7) != 0)
		return StromError_BadRequestMessage;
	if (buflen < required)
		return StromError_DataStoreNoSpace;
	if (required > 0xffffffffUL)
		return StromError_DataStoreOutOfRange;
	crc_init();
	memset(buffer, 0, required);
	kern_multihash *kmhash = (kern_multihash *)buffer;
	kmhash->hostptr = (hostptr_t)(uintptr_t)&kmhash->hostptr;
	memcpy(kmhash->pg_crc32_table, crc_table, sizeof(crc_table));
	kmhash->ntables = ntables;
	size_t	usage = STROM_LONGALIGN(offsetof(kern_multihash, htable_offset) + sizeof(cl_uint) * ntables);
	for (int t = 0; t < ntables; t++)
	{
		const kern_data_store *kds = tables[t].inner;
		kern_hashtable *kht = (kern_hashtable *)((char *)buffer + usage);
		uint32_t nslots = hashtable_nslots(kds->nitems);

		kmhash->htable_offset[t] = (cl_uint)usage;
		kht->ncols = kds->ncols;
		kht->nslots = nslots;
		kht->is_outer = 0;
		memcpy(kht->colmeta, kds->colmeta, sizeof(kern_colmeta) * kds->ncols);
		cl_uint *slots = KERN_HASHTABLE_SLOT(kht);
		size_t	consumed = hashtable_head_length(kds->ncols, nslots);
		row_image ri;
		row_image_init(ri, kds);
		for (uint32_t r = 0; r < kds->nitems; r++)
		{
			size_t	hoff, t_len;
			row_image_load(ri, kds, r);
			t_len = heap_tuple_size((int)kds->ncols, ri.cols.data(), 0, &hoff);
			kern_hashentry *he = (kern_hashentry *)((char *)kht + consumed);
			uint32_t crc = 0xFFFFFFFFU;
			for (int k = 0; k < tables[t].nkeys; k++)
			{
				int col = tables[t].key_attnos[k] - 1;
				if (col < 0 || col >= (int)kds->ncols)
					return StromError_BadRequestMessage;
				if (ri.nulls[col])
					continue;			/* NULL keys are not hashed */
				if (kds->colmeta[col].attlen < 0)
				{
					/* text / character(n): COMP_CRC32 over VARDATA_ANY / VARSIZE_ANY_EXHDR
					 * (gpuhashjoin.c:3775-3779).  A compressed or external key datum cannot
					 * be compared on the device: this relation joins on the CPU. */
					const unsigned char *p = (const unsigned char *)(uintptr_t)ri.values[col];
					if (p[0] == 0x01 || (p[0] & 0x03) == 0x02)
						return StromError_CpuReCheck;
					if (p[0] & 0x01)
						crc = legacy_crc32(crc, p + 1, (size_t)((p[0] >> 1) & 0x7f) - 1);
					else
					{
						uint32_t w;
						memcpy(&w, p, 4);
						crc = legacy_crc32(crc, p + 4, (size_t)((w >> 2) & 0x3fffffff) - 4);
					}
					continue;
				}
				crc = legacy_crc32(crc, &ri.values[col], kds->colmeta[col].attlen);
			}
			crc ^= 0xFFFFFFFFU;
			he->hash = crc;
			he->rowid = r;
			he->t_len = (cl_uint)t_len;
			heap_tuple_form((char *)&he->htup, (int)kds->ncols, ri.cols.data(), 0, t_len, hoff, 0, 0);
			he->next = slots[crc % nslots];
			slots[crc % nslots] = (cl_uint)consumed;
			consumed += KERN_HASHENTRY_SIZE_BY_TLEN(t_len);
		}
		kht->length = (cl_uint)consumed;
		usage += consumed;
	}
	return 0;
}

/* ------------------------------------------------------------------ *
 * 64-bit device numeric -> PostgreSQL's numeric  (pgstrom_fixup_kernel_numeric,
 * datastore.c:150-167)
 * ------------------------------------------------------------------ */
extern "C" int
strom_kernel_numeric_cstring(uint64_t image, char *buf, size_t room)
{
	int			expo = (int)((int64_t)image >> 58);
	bool		sign = ((image >> 57) & 1) != 0;
	uint64_t	mant = image & ((1ULL << 57) - 1);
	char		temp[64];
	/* (the reference's own format string) */
	int			n = snprintf(temp, sizeof(temp), "%c%llue%d", sign ? '-' : '+', (unsigned long long)mant, expo);
	if (n < 0 || !buf || (size_t)n + 1 > room)
		return -StromError_DataStoreNoSpace;
	memcpy(buf, temp, (size_t)n + 1);
	return n;
}

extern "C" int
strom_fixup_kernel_numeric(uint64_t image, void *varlena_out, size_t room)
{
	int			expo = (int)((int64_t)image >> 58);
	bool		sign = ((image >> 57) & 1) != 0;
	uint64_t	mant = image & ((1ULL << 57) - 1);
	/* decimal digits of the value, the decimal point after 'npoint' of them */
	char		digits[128];
	int			ndigits = 0, npoint, dscale;
	{
		char	m[32];
		int		nm = snprintf(m, sizeof(m), "%llu", (unsigned long long)mant);
		if (expo >= 0)
		{
			memcpy(digits, m, nm);
			memset(digits + nm, '0', expo);
			ndigits = nm + expo;
			npoint = ndigits;
			dscale = 0;
		}
		else
		{
			int		pad = (-expo + 1 > nm ? -expo + 1 - nm : 0);	/* at least one digit in front */
			memset(digits, '0', pad);
			memcpy(digits + pad, m, nm);
			ndigits = pad + nm;
			npoint = ndigits + expo;
			dscale = -expo;
		}
	}
	/* base-10000 groups aligned on the decimal point */
	int16_t		groups[48];
	int			ngroups = 0, weight;
	{
		int		lead = (4 - npoint % 4) % 4;			/* zeros in front of the integer part */
		int		ipos = -lead;
		int		nint = (npoint + lead) / 4;
		int		nfrac = (ndigits - npoint + 3) / 4;
		for (int g = 0; g < nint + nfrac; g++)
		{
			int		v = 0;
			for (int j = 0; j < 4; j++, ipos++)
				v = v * 10 + ((ipos >= 0 && ipos < ndigits) ? digits[ipos] - '0' : 0);
			groups[ngroups++] = (int16_t)v;
		}
		weight = nint - 1;
	}
	int			first = 0;
	while (first < ngroups && groups[first] == 0)
	{
		first++;
		weight--;
	}
	while (ngroups > first && groups[ngroups - 1] == 0)
		ngroups--;
	if (mant == 0)
	{
		first = ngroups = 0;
		weight = 0;
		sign = false;
	}
	int			nd = ngroups - first;
	bool		shortform = (weight >= -64 && weight <= 63 && dscale <= 63);
	size_t		len = 4 + (shortform ? 2 : 4) + 2 * (size_t)nd;
	if (!varlena_out || len > room)
		return -StromError_DataStoreNoSpace;
	unsigned char *out = (unsigned char *)varlena_out;
	uint32_t	vl = (uint32_t)len << 2;				/* SET_VARSIZE, 4-byte header */
	memcpy(out, &vl, 4);
	out += 4;
	if (shortform)
	{
		uint16_t h = (uint16_t)(0x8000 | (sign ? 0x2000 : 0) | (dscale << 7) |
								(weight < 0 ? 0x0040 : 0) | (weight & 0x3f));
		memcpy(out, &h, 2);
		out += 2;
	}
	else
	{
		uint16_t h = (uint16_t)((sign ? 0x4000 : 0) | dscale);
		int16_t	 w = (int16_t)weight;
		memcpy(out, &h, 2);
		memcpy(out + 2, &w, 2);
		out += 4;
	}
	for (int g = first; g < ngroups; g++, out += 2)
		memcpy(out, &groups[g], 2);
	return (int)len;
}

