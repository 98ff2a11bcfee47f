/*
 * strom_oracle.c -- CPU restatement of the per-chunk algorithms
 *
 * TEST INFRASTRUCTURE ONLY (see strom_oracle.h).  Single thread, one tuple
 * at a time, interpreting the expression tree per row -- the shape of
 * PostgreSQL's SeqScan + ExecQual, which is what the reference's own tests
 * compare against.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "strom_oracle.h"

/* ====================================================================== *
 * tuple access: restates opencl_common.h:817-981
 * ====================================================================== */
#define PAGE_HEADER_SIZE	24
#define ITEMID_OFFSET(x)	((x) & 0x7fff)			/* shift 0  */
#define ITEMID_LENGTH(x)	(((x) >> 17) & 0x7fff)	/* shift 17 */

static unsigned
varsize_any(const unsigned char *p)
{
	if (p[0] == 0x01)
		return 2 + (p[1] == 18 ? 16 : 8);			/* external TOAST pointer */
	if (p[0] & 0x01)
		return (p[0] >> 1) & 0x7f;					/* 1-byte header */
	return ((unsigned)p[0] | ((unsigned)p[1] << 8) |
			((unsigned)p[2] << 16) | ((unsigned)p[3] << 24)) >> 2;
}

/* kern_get_datum_tuple (opencl_common.h:817-864) */
static const void *
get_datum_tuple(const kern_colmeta *colmeta, const HeapTupleHeaderData *htup, uint32_t colidx)
{
	int			hasnull = (htup->t_infomask & HEAP_HASNULL) != 0;
	uint32_t	offset = htup->t_hoff;
	uint32_t	natts = htup->t_infomask2 & HEAP_NATTS_MASK;
	uint32_t	i;

	if (colidx >= natts)
		return NULL;
	if (!hasnull && colmeta[colidx].attcacheoff >= 0)
		return (const char *)htup + colmeta[colidx].attcacheoff;
	for (i = 0; i < natts; i++)
	{
		if (hasnull && !(htup->t_bits[i >> 3] & (1 << (i & 7))))
		{
			if (i == colidx)
				return NULL;
		}
		else
		{
			const char *addr;

			if (colmeta[i].attlen > 0)
				offset = STROM_TYPEALIGN(colmeta[i].attalign, offset);
			else if (((const unsigned char *)htup)[offset] == 0)
				offset = STROM_TYPEALIGN(colmeta[i].attalign, offset);
			addr = (const char *)htup + offset;
			if (i == colidx)
				return addr;
			offset += (colmeta[i].attlen > 0 ? (uint32_t)colmeta[i].attlen
					   : varsize_any((const unsigned char *)addr));
		}
	}
	return NULL;
}

const void *
oracle_get_datum(const kern_data_store *kds, uint32_t colidx, uint32_t rowidx)
{
	if (colidx >= kds->ncols || rowidx >= kds->nitems)
		return NULL;
	switch (kds->format)
	{
		case KDS_FORMAT_ROW:		/* kern_get_tuple_rs (866-903) */
			{
				const kern_rowitem *ri = KERN_DATA_STORE_ROWITEM(kds, rowidx);
				const char *page;
				uint16_t	pd_lower;
				uint32_t	item_max, itemid;

				if (ri->blk_index >= kds->nblocks)
					return NULL;
				page = KERN_DATA_STORE_ROWBLOCK(kds, ri->blk_index);
				memcpy(&pd_lower, page + 12, 2);
				item_max = (pd_lower <= PAGE_HEADER_SIZE ? 0
							: (pd_lower - PAGE_HEADER_SIZE) / 4);
				if (PAGE_HEADER_SIZE + 4 * (item_max + 1) >= BLCKSZ ||
					ri->item_offset == 0 || ri->item_offset > item_max)
					return NULL;
				memcpy(&itemid, page + PAGE_HEADER_SIZE + 4 * (ri->item_offset - 1), 4);
				if (ITEMID_OFFSET(itemid) + HEAPTUPLE_HEADER_FIXED >= BLCKSZ)
					return NULL;
				return get_datum_tuple(kds->colmeta,
									   (const HeapTupleHeaderData *)(page + ITEMID_OFFSET(itemid)),
									   colidx);
			}
		case KDS_FORMAT_ROW_FLAT:	/* kern_get_tuple_rsflat (919-933) */
			{
				uint32_t off = KERN_DATA_STORE_ROWITEM(kds, rowidx)->htup_offset;
				if (off >= kds->length)
					return NULL;
				return get_datum_tuple(kds->colmeta,
									   (const HeapTupleHeaderData *)((const char *)kds + off),
									   colidx);
			}
		case KDS_FORMAT_TUPSLOT:	/* kern_get_datum_tupslot (949-963) */
			if (KERN_DATA_STORE_ISNULL(kds, rowidx)[colidx])
				return NULL;
			return KERN_DATA_STORE_VALUES(kds, rowidx) + colidx;
		case KDS_FORMAT_COLUMN:		/* this build's layout, strom_kds.h */
			{
				const kern_coldir *cd = KERN_DATA_STORE_COLDIR(kds) + colidx;
				if (cd->nulls_off)
				{
					const uint32_t *nn = (const uint32_t *)((const char *)kds + cd->nulls_off);
					if (!((nn[rowidx >> 5] >> (rowidx & 31)) & 1))
						return NULL;
				}
				if (kds->colmeta[colidx].attlen < 0)
				{
					/* a varlena column: 8-byte offset of the row's datum from the chunk head,
					 * 0 = NULL (strom_kds.h) */
					uint64_t off;
					memcpy(&off, (const char *)kds + cd->values_off + 8 * (size_t)rowidx, 8);
					if (off == 0 || off >= kds->length)
						return NULL;
					return (const char *)kds + off;
				}
				return (const char *)kds + cd->values_off +
					(size_t)kds->colmeta[colidx].attlen * rowidx;
			}
	}
	return NULL;
}

/* ====================================================================== *
 * expression tree
 * ====================================================================== */
enum {
	N_CONST, N_PARAM, N_VAR, N_FUNC, N_AND, N_OR, N_NOT, N_ISNULL, N_ISNOTNULL,
	N_BOOLTEST, N_CASE, N_RELABEL, N_IVAR
};
/* operator classes of N_FUNC */
enum {
	OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD,
	OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_CMP,
	OP_UMINUS, OP_UPLUS, OP_ABS, OP_BITNOT, OP_BITAND, OP_BITOR, OP_BITXOR,
	OP_SHL, OP_SHR, OP_CAST,
	OP_CEIL, OP_FLOOR, OP_ROUND, OP_TRUNC, OP_SIGN, OP_SQRT, OP_PI,
	OP_CBRT, OP_EXP, OP_LN, OP_LOG10, OP_POW, OP_DEGREES, OP_RADIANS,
	OP_ACOS, OP_ASIN, OP_ATAN, OP_ATAN2, OP_COS, OP_SIN, OP_TAN,
	OP_DATE_PLI, OP_DATE_MII, OP_DATE_MI, OP_INT_PL_DATE,
	OP_DATE_TO_TS, OP_TS_TO_DATE, OP_TS_TO_TIME, OP_DATETIME_PL, OP_TIMEDATE_PL,
	OP_NUM_ADD, OP_NUM_SUB, OP_NUM_MUL, OP_NUM_UMINUS, OP_NUM_UPLUS, OP_NUM_ABS,
	OP_NUM_FROM_INT, OP_NUM_TO_INT, OP_NUM_TO_FLOAT, OP_NUM_FROM_FLOAT, OP_IDENTITY
};
enum { BT_TRUE, BT_NOT_TRUE, BT_FALSE, BT_NOT_FALSE, BT_UNKNOWN, BT_NOT_UNKNOWN };

struct oracle_expr {
	int			kind;
	int			type_oid;		/* result type */
	int			op;				/* OP_* / BT_* */
	int			attno;			/* N_VAR (1-based) / N_PARAM id */
	int			depth;			/* N_IVAR: which inner relation (1-based) */
	oracle_value cval;			/* N_CONST */
	int			nargs;
	struct oracle_expr **args;	/* N_CASE: cond0,res0,cond1,res1,...,[else] */
	int			has_else;
	struct oracle_expr *case_arg;	/* simple CASE subject */
	/* N_VAR of a column stored as int8 at 10^-dscale ((var N decimal S), STROM_DECIMALOID):
	 * its SQL value is the numeric, which is what the checker computes with */
	int			is_decimal;
	int			dscale;
};

typedef struct { const char *p; char *err; size_t errlen; int failed; } parser;

static void
perr(parser *ps, const char *fmt, ...)
{
	va_list ap;
	if (ps->failed)
		return;
	ps->failed = 1;
	if (ps->err && ps->errlen)
	{
		va_start(ap, fmt);
		vsnprintf(ps->err, ps->errlen, fmt, ap);
		va_end(ap);
	}
}

static void skip_ws(parser *ps) { while (isspace((unsigned char)*ps->p)) ps->p++; }

static int
read_atom(parser *ps, char *buf, size_t buflen)
{
	size_t n = 0;
	skip_ws(ps);
	if (*ps->p == '\'')
	{
		ps->p++;
		while (*ps->p && *ps->p != '\'' && n + 1 < buflen)
			buf[n++] = *ps->p++;
		if (*ps->p != '\'')
			return 0;
		ps->p++;
		buf[n] = 0;
		return 1;
	}
	while (*ps->p && !isspace((unsigned char)*ps->p) && *ps->p != '(' && *ps->p != ')' && n + 1 < buflen)
		buf[n++] = *ps->p++;
	buf[n] = 0;
	return n > 0;
}

static int
type_by_name(const char *s)
{
	static const struct { const char *n; int oid; } t[] = {
		{"bool", STROM_BOOLOID}, {"int2", STROM_INT2OID}, {"int4", STROM_INT4OID},
		{"int8", STROM_INT8OID}, {"float4", STROM_FLOAT4OID}, {"float8", STROM_FLOAT8OID},
		{"date", STROM_DATEOID}, {"time", STROM_TIMEOID}, {"timestamp", STROM_TIMESTAMPOID},
		{"numeric", STROM_NUMERICOID}, {"char1", STROM_BPCHAROID},
		{"integer", STROM_INT4OID}, {"int", STROM_INT4OID}, {"smallint", STROM_INT2OID},
		{"bigint", STROM_INT8OID}, {"real", STROM_FLOAT4OID}, {"double", STROM_FLOAT8OID},
		{"float", STROM_FLOAT8OID}, {"bpchar", STROM_BPCHAROID},
		{"text", STROM_TEXTOID}, {"character", STROM_BPCHARNOID},
	};
	size_t i;
	for (i = 0; i < sizeof(t) / sizeof(t[0]); i++)
		if (strcmp(t[i].n, s) == 0)
			return t[i].oid;
	return 0;
}

static int type_is_varlena(int t)
{ return t == STROM_TEXTOID || t == STROM_BPCHARNOID; }

/*
 * text / character(n) values are the address of their varlena datum
 * (opencl_common.h:1126-1154 pg_varlena_t).  Readable in place: a 1-byte
 * header or an uncompressed 4-byte header; compressed and external datums go
 * back to the CPU.
 */
static int
varlena_readable(const void *addr)
{
	uint8_t b0 = *(const uint8_t *)addr;
	return !(b0 == 0x01 || (b0 & 0x03) == 0x02);
}

static const uint8_t *
varlena_payload(const void *addr, int *len)
{
	const uint8_t *p = (const uint8_t *)addr;
	uint32_t w;
	if (p[0] & 0x01)
	{
		*len = (int)((p[0] >> 1) & 0x7f) - 1;
		return p + 1;
	}
	memcpy(&w, p, 4);
	*len = (int)((w >> 2) & 0x3fffffff) - 4;
	return p + 4;
}

/* text_compare / bpchar_compare (opencl_textlib.h:150-193, 285-312): bytewise
 * on unsigned bytes (PostgreSQL's "C" collation = memcmp), the shorter one
 * first; character(n) ignores trailing blanks */
static int
varlena_compare(const void *a, const void *b, int blank_padded)
{
	int la, lb, n, c;
	const uint8_t *pa = varlena_payload(a, &la);
	const uint8_t *pb = varlena_payload(b, &lb);
	if (blank_padded)
	{
		while (la > 0 && pa[la - 1] == ' ') la--;
		while (lb > 0 && pb[lb - 1] == ' ') lb--;
	}
	n = la < lb ? la : lb;
	c = n > 0 ? memcmp(pa, pb, (size_t)n) : 0;
	if (c != 0)
		return c < 0 ? -1 : 1;
	return la == lb ? 0 : (la > lb ? 1 : -1);
}

static int type_is_int(int t)
{ return t == STROM_INT2OID || t == STROM_INT4OID || t == STROM_INT8OID; }
static int type_is_float(int t)
{ return t == STROM_FLOAT4OID || t == STROM_FLOAT8OID; }

static int
type_len(int t)
{
	switch (t)
	{
		case STROM_BOOLOID: case STROM_BPCHAROID: return 1;
		case STROM_INT2OID: return 2;
		case STROM_INT4OID: case STROM_FLOAT4OID: case STROM_DATEOID: return 4;
		default: return 8;
	}
}

/* days from civil date, PostgreSQL date2j (timelib date2j, opencl_timelib.h:125-160) */
static int
oracle_date2j(int y, int m, int d)
{
	int julian, century;
	if (m > 2) { m += 1; y += 4800; } else { m += 13; y += 4799; }
	century = y / 100;
	julian = y * 365 - 32167;
	julian += y / 4 - century + century / 4;
	julian += 7834 * m / 256 + d;
	return julian;
}
#define POSTGRES_EPOCH_JDATE 2451545


/* ---------------------------------------------------------------------- *
 * 64-bit NUMERIC (opencl_numeric.h:122-162): exponent 63..58 (signed, base
 * 10), sign 57, mantissa 56..0.  The operations below restate
 * opencl_numeric.h:816-1234 with its overflow rules: an intermediate that
 * leaves 64 bits, a mantissa beyond 57 bits or an exponent outside
 * [-32, 31] is a CpuReCheck.
 * ---------------------------------------------------------------------- */
#define NUM_EXPO(u)		((int)((int64_t)(u) >> 58))
#define NUM_SIGN(u)		((int)(((u) >> 57) & 1))
#define NUM_MANT(u)		((u) & ((1ULL << 57) - 1))

static int
num_pack(int expo, int sign, uint64_t mant, uint64_t *out)
{
	if (mant == 0) { *out = 0; return 1; }
	while (mant % 10 == 0 && expo < 31) { mant /= 10; expo++; }
	while (expo > 31)
	{
		if (mant > ((1ULL << 57) - 1) / 10) return 0;
		mant *= 10; expo--;
	}
	if (expo < -32 || mant >= (1ULL << 57)) return 0;
	*out = (((uint64_t)(int64_t)expo) << 58) | ((uint64_t)(sign != 0) << 57) | mant;
	return 1;
}

/* ---- float -> numeric: exact scaling (see OP_NUM_FROM_FLOAT) ---- */
/* 192-bit unsigned integers, six 32-bit limbs, little endian: all the arithmetic float -> numeric needs */
static void
oracle_w192_mul_small(uint32_t *w, uint32_t m)
{
	uint64_t	carry = 0;
	for (int i = 0; i < 6; i++)
	{
		uint64_t t = (uint64_t)w[i] * m + carry;
		w[i] = (uint32_t)t;
		carry = t >> 32;
	}
}
static uint32_t
oracle_w192_div_small(uint32_t *w, uint32_t d)
{
	uint64_t	rem = 0;
	for (int i = 5; i >= 0; i--)
	{
		uint64_t t = (rem << 32) | w[i];
		w[i] = (uint32_t)(t / d);
		rem = t % d;
	}
	return (uint32_t)rem;
}
static void
oracle_w192_shl(uint32_t *w, int n)
{
	int		limbs = n / 32, bits = n % 32;
	for (int i = 5; i >= 0; i--)
	{
		uint64_t lo = (i - limbs >= 0 ? w[i - limbs] : 0), lo2 = (i - limbs - 1 >= 0 ? w[i - limbs - 1] : 0);
		w[i] = (uint32_t)(bits ? ((lo << bits) | (lo2 >> (32 - bits))) : lo);
	}
}
/* >> n; returns whether a 1 bit was shifted out */
static int
oracle_w192_shr(uint32_t *w, int n)
{
	int		limbs = n / 32, bits = n % 32, sticky = 0;
	for (int i = 0; i < 6; i++)
	{
		if (i < limbs)
			sticky |= (w[i] != 0);
		else if (i == limbs && bits)
			sticky |= ((w[i] & ((1u << bits) - 1)) != 0);
	}
	for (int i = 0; i < 6; i++)
	{
		uint64_t lo = (i + limbs < 6 ? w[i + limbs] : 0), hi = (i + limbs + 1 < 6 ? w[i + limbs + 1] : 0);
		w[i] = (uint32_t)(bits ? ((lo >> bits) | (hi << (32 - bits))) : lo);
	}
	return sticky;
}

/*
 * round_half_even(M x 2^E x 10^k), exactly: multiplications and left shifts first, then one more
 * bit (the half), then divisions and right shifts with a sticky flag.  0 when it does not fit 64 bits.
 */
static int
oracle_scaled_mantissa(uint64_t M, int E, int k, uint64_t *p_mant)
{
	uint32_t	w[6] = { (uint32_t)M, (uint32_t)(M >> 32), 0, 0, 0, 0 };
	int		a = E + k, b = k, sticky = 0;		/* x 2^a x 5^b */

	if (b > 47 || b < -36 || a > 130 || a < -185)
		return 0;
	for (; b >= 13; b -= 13)
		oracle_w192_mul_small(w, 1220703125u);		/* 5^13 */
	for (; b > 0; b--)
		oracle_w192_mul_small(w, 5u);
	if (a > 0)
		oracle_w192_shl(w, a);
	oracle_w192_shl(w, 1);
	for (; b <= -13; b += 13)
		sticky |= (oracle_w192_div_small(w, 1220703125u) != 0);
	for (; b < 0; b++)
		sticky |= (oracle_w192_div_small(w, 5u) != 0);
	if (a < 0)
		sticky |= oracle_w192_shr(w, -a);
	if (w[2] | w[3] | w[4] | w[5])
		return 0;
	uint64_t	q = ((uint64_t)w[1] << 32) | w[0];
	uint64_t	mant = q >> 1;
	if ((q & 1) && (sticky || (mant & 1)))
		mant++;									/* above the half, or the tie to even */
	*p_mant = mant;
	return 1;
}

static uint64_t
num_pow10(int n)
{
	uint64_t m = 1;
	int i;
	if (n < 0 || n > 19) return 0;
	for (i = 0; i < n; i++) m *= 10;
	return m;
}

/*
 * PostgreSQL's varlena numeric as it sits in a heap tuple -> 64-bit form
 * (opencl_numeric.h:166-307 does this per row on the device; layout:
 * utils/adt/numeric.c of PostgreSQL 9.4).  Returns 0 when the value cannot
 * be carried (NaN, compressed / external datum, mantissa beyond 57 bits).
 */
int
oracle_numeric_from_varlena(const void *addr, uint64_t *out)
{
	const unsigned char *p = (const unsigned char *)addr;
	uint32_t	len, n_header, ndigits, i;
	int			sign, weight;
	uint64_t	mant = 0;

	if (p[0] == 0x01)
		return 0;
	if (p[0] & 0x01)
	{
		len = (p[0] >> 1) & 0x7f;
		if (len < 3) return 0;
		len -= 1;
		p += 1;
	}
	else
	{
		uint32_t w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
		if ((w & 3) != 0 || (w >> 2) < 6) return 0;
		len = (w >> 2) - 4;
		p += 4;
	}
	n_header = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
	if ((n_header & 0xC000) == 0xC000)
		return 0;							/* NaN */
	if ((n_header & 0xC000) == 0x8000)
	{
		sign = (n_header & 0x2000) != 0;
		weight = (int)(n_header & 0x3F);
		if (n_header & 0x40) weight |= ~0x3F;
		p += 2;
		ndigits = (len - 2) / 2;
	}
	else
	{
		if (len < 4) return 0;
		sign = (n_header & 0xC000) == 0x4000;
		weight = (int)(int16_t)((uint32_t)p[2] | ((uint32_t)p[3] << 8));
		p += 4;
		ndigits = (len - 4) / 2;
	}
	for (i = 0; i < ndigits; i++)
	{
		uint32_t d = (uint32_t)p[2 * i] | ((uint32_t)p[2 * i + 1] << 8);
		if (d > 9999 || mant > (UINT64_MAX - 9999) / 10000)
			return 0;
		mant = mant * 10000 + d;
	}
	/* num_pack strips trailing decimal zeros before it checks the 57 bits */
	return num_pack((weight - (int)ndigits + 1) * 4, sign, mant, out);
}

int
oracle_numeric_from_text(const char *lit, uint64_t *out)
{
	const char *p = lit;
	int		neg = 0, expo = 0, seen = 0, dot = 0;
	unsigned __int128 mant = 0;

	if (*p == '+' || *p == '-') neg = (*p++ == '-');
	for (; *p; p++)
	{
		if (*p >= '0' && *p <= '9')
		{
			mant = mant * 10 + (unsigned)(*p - '0');
			if (mant >> 100) return 0;
			if (dot) expo--;
			seen = 1;
		}
		else if (*p == '.' && !dot) dot = 1;
		else break;
	}
	if (!seen) return 0;
	if (*p == 'e' || *p == 'E')
	{
		char *end;
		long e = strtol(p + 1, &end, 10);
		if (end == p + 1 || *end) return 0;
		expo += (int)e;
	}
	else if (*p) return 0;
	if (mant == 0) { *out = 0; return 1; }
	while (mant % 10 == 0) { mant /= 10; expo++; }
	while (expo > 31 && mant < ((unsigned __int128)1 << 57) / 10) { mant *= 10; expo--; }
	if (mant >= ((unsigned __int128)1 << 57) || expo < -32 || expo > 31) return 0;
	*out = (((uint64_t)(int64_t)expo) << 58) | ((uint64_t)neg << 57) | (uint64_t)mant;
	return 1;
}

static int
num_add(uint64_t a, uint64_t b, uint64_t *out)
{
	int			e1 = NUM_EXPO(a), e2 = NUM_EXPO(b), s1 = NUM_SIGN(a), s2 = NUM_SIGN(b);
	uint64_t	m1 = NUM_MANT(a), m2 = NUM_MANT(b);

	if (m1 == 0) { *out = b; return 1; }
	if (m2 == 0) { *out = a; return 1; }
	if (e1 != e2)
	{
		int diff = e1 > e2 ? e1 - e2 : e2 - e1;
		uint64_t mag = num_pow10(diff);
		uint64_t *big = e1 > e2 ? &m1 : &m2;
		if (mag == 0 || __builtin_mul_overflow(*big, mag, big)) return 0;
		e1 = e2 = (e1 < e2 ? e1 : e2);
	}
	if (s1 != s2)
	{
		if (m1 < m2) { s1 = s2; m1 = m2 - m1; }
		else m1 -= m2;
	}
	else if (__builtin_add_overflow(m1, m2, &m1)) return 0;
	return num_pack(e1, s1, m1, out);
}

static int
num_mul(uint64_t a, uint64_t b, uint64_t *out)
{
	uint64_t m1 = NUM_MANT(a), m2 = NUM_MANT(b), prod;
	if (m1 == 0 || m2 == 0) { *out = 0; return 1; }
	if (__builtin_mul_overflow(m1, m2, &prod)) return 0;
	return num_pack(NUM_EXPO(a) + NUM_EXPO(b), NUM_SIGN(a) != NUM_SIGN(b), prod, out);
}

static int
num_cmp(uint64_t a, uint64_t b)
{
	int			e1 = NUM_EXPO(a), e2 = NUM_EXPO(b), s1 = NUM_SIGN(a), s2 = NUM_SIGN(b), ret;
	uint64_t	m1 = NUM_MANT(a), m2 = NUM_MANT(b);
	/* exact: compare sign * mant * 10^expo with 128-bit headroom where it fits */
	if (m1 == 0 && m2 == 0) return 0;
	if (m1 == 0) return s2 ? 1 : -1;
	if (m2 == 0) return s1 ? -1 : 1;
	if (s1 != s2) return s1 ? -1 : 1;
	if (e1 == e2) ret = m1 < m2 ? -1 : m1 > m2 ? 1 : 0;
	else
	{
		int first_big = e1 > e2, diff = first_big ? e1 - e2 : e2 - e1, c;
		uint64_t mag = num_pow10(diff), big = first_big ? m1 : m2, small = first_big ? m2 : m1, scaled;
		if (mag == 0 || __builtin_mul_overflow(big, mag, &scaled)) c = 1;
		else c = scaled < small ? -1 : scaled > small ? 1 : 0;
		ret = first_big ? c : -c;
	}
	return s1 ? -ret : ret;
}

/* numeric -> fixed point int64 at 10^-scale, exact or fail */
static int
num_to_fixed(uint64_t a, int scale, int64_t *out)
{
	int			shift = NUM_EXPO(a) + scale;
	uint64_t	mant = NUM_MANT(a), mag = num_pow10(shift);
	if (mant == 0) { *out = 0; return 1; }
	if (shift < 0 || mag == 0 || __builtin_mul_overflow(mant, mag, &mant) ||
		mant > 9223372036854775807ULL)
		return 0;
	*out = NUM_SIGN(a) ? -(int64_t)mant : (int64_t)mant;
	return 1;
}

static int
parse_literal(parser *ps, int type, const char *lit, oracle_value *out)
{
	char *end = NULL;
	memset(out, 0, sizeof(*out));
	out->type_oid = type;
	if (strcmp(lit, "null") == 0 || strcmp(lit, "NULL") == 0)
	{
		out->isnull = 1;
		return 1;
	}
	if (type_is_varlena(type))
	{
		/* a varlena image with a 4-byte header; lives as long as the process
		 * (test infrastructure) */
		size_t		len = strlen(lit);
		uint32_t	hdr = (uint32_t)((len + 4) << 2);
		char	   *img = malloc(len + 4);
		if (!img)
			return 0;
		memcpy(img, &hdr, 4);
		memcpy(img + 4, lit, len);
		out->v.i = (int64_t)(intptr_t)img;
		return 1;
	}
	switch (type)
	{
		case STROM_BOOLOID:
			if (!strcmp(lit, "t") || !strcmp(lit, "true") || !strcmp(lit, "1")) out->v.i = 1;
			else if (!strcmp(lit, "f") || !strcmp(lit, "false") || !strcmp(lit, "0")) out->v.i = 0;
			else { perr(ps, "bad bool literal %s", lit); return 0; }
			return 1;
		case STROM_INT2OID: case STROM_INT4OID: case STROM_INT8OID: case STROM_TIMEOID:
			out->v.i = strtoll(lit, &end, 10);
			if (end == lit || *end) { perr(ps, "bad integer literal %s", lit); return 0; }
			return 1;
		case STROM_FLOAT4OID:
			out->v.f = strtof(lit, &end);
			if (end == lit || *end) { perr(ps, "bad float literal %s", lit); return 0; }
			return 1;
		case STROM_FLOAT8OID:
			out->v.d = strtod(lit, &end);
			if (end == lit || *end) { perr(ps, "bad float literal %s", lit); return 0; }
			return 1;
		case STROM_DATEOID:
			{
				int y, m, d;
				if (sscanf(lit, "%d-%d-%d", &y, &m, &d) == 3 && strchr(lit + 1, '-'))
					out->v.i = oracle_date2j(y, m, d) - POSTGRES_EPOCH_JDATE;
				else
				{
					out->v.i = strtoll(lit, &end, 10);
					if (end == lit || *end) { perr(ps, "bad date literal %s", lit); return 0; }
				}
				return 1;
			}
		case STROM_TIMESTAMPOID:
			{
				int y, m, d, hh = 0, mi = 0; double ss = 0;
				int n = sscanf(lit, "%d-%d-%d %d:%d:%lf", &y, &m, &d, &hh, &mi, &ss);
				if (n >= 3 && strchr(lit + 1, '-'))
					out->v.i = (int64_t)(oracle_date2j(y, m, d) - POSTGRES_EPOCH_JDATE) * 86400000000LL
						+ ((int64_t)hh * 3600 + mi * 60) * 1000000LL + (int64_t)llround(ss * 1e6);
				else
				{
					out->v.i = strtoll(lit, &end, 10);
					if (end == lit || *end) { perr(ps, "bad timestamp literal %s", lit); return 0; }
				}
				return 1;
			}
		case STROM_NUMERICOID:
			if (!oracle_numeric_from_text(lit, &out->v.u))
			{ perr(ps, "numeric literal %s does not fit the 64-bit device form", lit); return 0; }
			return 1;
		case STROM_BPCHAROID:
			if (strlen(lit) != 1) { perr(ps, "char1 literal must be one byte"); return 0; }
			out->v.i = (signed char)lit[0];
			return 1;
	}
	perr(ps, "no literal syntax for type %d", type);
	return 0;
}

static oracle_expr *
new_node(int kind)
{
	oracle_expr *e = calloc(1, sizeof(*e));
	e->kind = kind;
	return e;
}

static void
add_arg(oracle_expr *e, oracle_expr *a)
{
	e->args = realloc(e->args, sizeof(oracle_expr *) * (e->nargs + 1));
	e->args[e->nargs++] = a;
}

void
oracle_expr_free(oracle_expr *e)
{
	int i;
	if (!e)
		return;
	for (i = 0; i < e->nargs; i++)
		oracle_expr_free(e->args[i]);
	oracle_expr_free(e->case_arg);
	free(e->args);
	free(e);
}

/* ---- function resolution: pg_proc name + argument types ------------- */
static int
family_types(const char *sfx, const char *pfx, int *x, int *y, const char **rest)
{
	/* sfx examples: "4pl", "24lt", "82mi" after the "int"/"float" prefix */
	static const struct { const char *s; int x, y; } ints[] = {
		{"24", STROM_INT2OID, STROM_INT4OID}, {"28", STROM_INT2OID, STROM_INT8OID},
		{"42", STROM_INT4OID, STROM_INT2OID}, {"48", STROM_INT4OID, STROM_INT8OID},
		{"82", STROM_INT8OID, STROM_INT2OID}, {"84", STROM_INT8OID, STROM_INT4OID},
		{"2", STROM_INT2OID, STROM_INT2OID}, {"4", STROM_INT4OID, STROM_INT4OID},
		{"8", STROM_INT8OID, STROM_INT8OID},
	};
	static const struct { const char *s; int x, y; } flts[] = {
		{"48", STROM_FLOAT4OID, STROM_FLOAT8OID}, {"84", STROM_FLOAT8OID, STROM_FLOAT4OID},
		{"4", STROM_FLOAT4OID, STROM_FLOAT4OID}, {"8", STROM_FLOAT8OID, STROM_FLOAT8OID},
	};
	size_t i;
	if (strcmp(pfx, "int") == 0)
	{
		for (i = 0; i < sizeof(ints) / sizeof(ints[0]); i++)
			if (strncmp(sfx, ints[i].s, strlen(ints[i].s)) == 0)
			{
				*x = ints[i].x; *y = ints[i].y; *rest = sfx + strlen(ints[i].s);
				return 1;
			}
	}
	else
	{
		for (i = 0; i < sizeof(flts) / sizeof(flts[0]); i++)
			if (strncmp(sfx, flts[i].s, strlen(flts[i].s)) == 0)
			{
				*x = flts[i].x; *y = flts[i].y; *rest = sfx + strlen(flts[i].s);
				return 1;
			}
	}
	return 0;
}

static int
wider_type(int x, int y)
{
	int rank_x = type_len(x), rank_y = type_len(y);
	if (type_is_float(x) || type_is_float(y))
		return (rank_x == 8 || rank_y == 8) ? STROM_FLOAT8OID : STROM_FLOAT4OID;
	return rank_x >= rank_y ? x : y;
}

static int
binop_by_suffix(const char *s)
{
	static const struct { const char *s; int op; } t[] = {
		{"pl", OP_ADD}, {"mi", OP_SUB}, {"mul", OP_MUL}, {"div", OP_DIV}, {"mod", OP_MOD},
		{"eq", OP_EQ}, {"ne", OP_NE}, {"lt", OP_LT}, {"le", OP_LE}, {"gt", OP_GT}, {"ge", OP_GE},
		{"and", OP_BITAND}, {"or", OP_BITOR}, {"xor", OP_BITXOR}, {"shl", OP_SHL}, {"shr", OP_SHR},
	};
	size_t i;
	for (i = 0; i < sizeof(t) / sizeof(t[0]); i++)
		if (strcmp(t[i].s, s) == 0)
			return t[i].op;
	return -1;
}

/* fills e->op / e->type_oid for a function call; 0 when unknown */
static int
resolve_func(oracle_expr *e, const char *name)
{
	int		nargs = e->nargs;
	int		a0 = nargs > 0 ? e->args[0]->type_oid : 0;
	int		a1 = nargs > 1 ? e->args[1]->type_oid : 0;
	int		x, y, op;
	const char *rest;

	/* casts */
	if (nargs == 1)
	{
		int target = 0;
		if (!strcmp(name, "int2")) target = STROM_INT2OID;
		else if (!strcmp(name, "int4")) target = STROM_INT4OID;
		else if (!strcmp(name, "int8")) target = STROM_INT8OID;
		else if (!strcmp(name, "float4")) target = STROM_FLOAT4OID;
		else if (!strcmp(name, "float8")) target = STROM_FLOAT8OID;
		if (target && target != a0 &&
			(type_is_int(a0) || type_is_float(a0) || (a0 == STROM_BOOLOID && target == STROM_INT4OID)))
		{
			e->op = OP_CAST;
			e->type_oid = target;
			return 1;
		}
	}
	/* character(n) / text (codegen.c:616-629) */
	if (nargs == 2 && a0 == STROM_BPCHARNOID && a1 == STROM_BPCHARNOID && !strncmp(name, "bpchar", 6))
	{
		if (!strcmp(name + 6, "cmp")) { e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
		op = binop_by_suffix(name + 6);
		if (op >= OP_EQ && op <= OP_GE) { e->op = op; e->type_oid = STROM_BOOLOID; return 1; }
	}
	if (nargs == 2 && a0 == STROM_TEXTOID && a1 == STROM_TEXTOID)
	{
		if (!strcmp(name, "bttextcmp")) { e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
		op = -1;
		if (!strcmp(name, "texteq") || !strcmp(name, "textne"))
			op = binop_by_suffix(name + 4);
		else if (!strncmp(name, "text_", 5))
			op = binop_by_suffix(name + 5);
		if (op >= OP_EQ && op <= OP_GE) { e->op = op; e->type_oid = STROM_BOOLOID; return 1; }
	}
	/* numeric */
	if (nargs == 2 && a0 == STROM_NUMERICOID && a1 == STROM_NUMERICOID && !strncmp(name, "numeric_", 8))
	{
		const char *sfx = name + 8;
		if (!strcmp(sfx, "add")) { e->op = OP_NUM_ADD; e->type_oid = STROM_NUMERICOID; return 1; }
		if (!strcmp(sfx, "sub")) { e->op = OP_NUM_SUB; e->type_oid = STROM_NUMERICOID; return 1; }
		if (!strcmp(sfx, "mul")) { e->op = OP_NUM_MUL; e->type_oid = STROM_NUMERICOID; return 1; }
		if (!strcmp(sfx, "cmp")) { e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
		op = binop_by_suffix(sfx);
		if (op >= OP_EQ && op <= OP_GE) { e->op = op; e->type_oid = STROM_BOOLOID; return 1; }
	}
	if (nargs == 1 && a0 == STROM_NUMERICOID)
	{
		if (!strcmp(name, "numeric_uminus")) { e->op = OP_NUM_UMINUS; e->type_oid = a0; return 1; }
		if (!strcmp(name, "numeric_uplus"))  { e->op = OP_NUM_UPLUS;  e->type_oid = a0; return 1; }
		if (!strcmp(name, "numeric_abs") || !strcmp(name, "abs")) { e->op = OP_NUM_ABS; e->type_oid = a0; return 1; }
		if (!strcmp(name, "int2")) { e->op = OP_NUM_TO_INT; e->type_oid = STROM_INT2OID; return 1; }
		if (!strcmp(name, "int4")) { e->op = OP_NUM_TO_INT; e->type_oid = STROM_INT4OID; return 1; }
		if (!strcmp(name, "int8")) { e->op = OP_NUM_TO_INT; e->type_oid = STROM_INT8OID; return 1; }
		if (!strcmp(name, "float4")) { e->op = OP_NUM_TO_FLOAT; e->type_oid = STROM_FLOAT4OID; return 1; }
		if (!strcmp(name, "float8")) { e->op = OP_NUM_TO_FLOAT; e->type_oid = STROM_FLOAT8OID; return 1; }
	}
	if (nargs == 1 && !strcmp(name, "numeric") && type_is_int(a0))
	{ e->op = OP_NUM_FROM_INT; e->type_oid = STROM_NUMERICOID; return 1; }
	/* float4_numeric / float8_numeric (codegen.c:519-520, opencl_numeric.h:625-779) */
	if (nargs == 1 && !strcmp(name, "numeric") && type_is_float(a0))
	{ e->op = OP_NUM_FROM_FLOAT; e->type_oid = STROM_NUMERICOID; return 1; }
	/* the alias casts date(date), time(time), timestamp(timestamp) (codegen.c:543-548) */
	if (nargs == 1 && ((!strcmp(name, "date") && a0 == STROM_DATEOID) ||
					   (!strcmp(name, "time") && a0 == STROM_TIMEOID) ||
					   (!strcmp(name, "timestamp") && a0 == STROM_TIMESTAMPOID)))
	{ e->op = OP_IDENTITY; e->type_oid = a0; return 1; }
	/* bt<family>cmp */
	if (strncmp(name, "bt", 2) == 0 && nargs == 2)
	{
		const char *n = name + 2;
		if (!strcmp(n, "boolcmp") && a0 == STROM_BOOLOID && a1 == STROM_BOOLOID)
		{ e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
		if (!strncmp(n, "int", 3) && family_types(n + 3, "int", &x, &y, &rest) &&
			!strcmp(rest, "cmp") && a0 == x && a1 == y)
		{ e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
		if (!strncmp(n, "float", 5) && family_types(n + 5, "float", &x, &y, &rest) &&
			!strcmp(rest, "cmp") && a0 == x && a1 == y)
		{ e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
	}
	/* int / float families */
	if ((!strncmp(name, "int", 3) && family_types(name + 3, "int", &x, &y, &rest)) ||
		(!strncmp(name, "float", 5) && family_types(name + 5, "float", &x, &y, &rest)))
	{
		int isint = (name[0] == 'i');
		if (nargs == 2 && (op = binop_by_suffix(rest)) >= 0)
		{
			if (op == OP_SHL || op == OP_SHR)
			{
				if (isint && x == y && a0 == x && a1 == STROM_INT4OID)
				{ e->op = op; e->type_oid = x; return 1; }
				return 0;
			}
			if (a0 != x || a1 != y)
				return 0;
			if (op == OP_MOD && (!isint || x != y))
				return 0;
			if ((op == OP_BITAND || op == OP_BITOR || op == OP_BITXOR) && (!isint || x != y))
				return 0;
			e->op = op;
			e->type_oid = (op >= OP_EQ && op <= OP_GE) ? STROM_BOOLOID : wider_type(x, y);
			return 1;
		}
		if (nargs == 1 && x == y && a0 == x)
		{
			if (!strcmp(rest, "um"))  { e->op = OP_UMINUS; e->type_oid = x; return 1; }
			if (!strcmp(rest, "up"))  { e->op = OP_UPLUS;  e->type_oid = x; return 1; }
			if (!strcmp(rest, "abs")) { e->op = OP_ABS;    e->type_oid = x; return 1; }
			if (!strcmp(rest, "not") && isint) { e->op = OP_BITNOT; e->type_oid = x; return 1; }
		}
		return 0;
	}
	if (!strcmp(name, "abs") && nargs == 1 && (type_is_int(a0) || type_is_float(a0)))
	{ e->op = OP_ABS; e->type_oid = a0; return 1; }
	if ((!strcmp(name, "booleq") || !strcmp(name, "boolne")) && nargs == 2 &&
		a0 == STROM_BOOLOID && a1 == STROM_BOOLOID)
	{ e->op = name[4] == 'e' ? OP_EQ : OP_NE; e->type_oid = STROM_BOOLOID; return 1; }
	if (nargs == 1 && a0 == STROM_FLOAT8OID)
	{
		static const struct { const char *n; int op; } f[] = {
			{"ceil", OP_CEIL}, {"ceiling", OP_CEIL}, {"floor", OP_FLOOR},
			{"round", OP_ROUND}, {"dround", OP_ROUND}, {"trunc", OP_TRUNC}, {"dtrunc", OP_TRUNC},
			{"sign", OP_SIGN}, {"sqrt", OP_SQRT}, {"dsqrt", OP_SQRT},
			/* codegen.c:467-503 */
			{"cbrt", OP_CBRT}, {"dcbrt", OP_CBRT}, {"exp", OP_EXP}, {"dexp", OP_EXP},
			{"ln", OP_LN}, {"dlog1", OP_LN}, {"log", OP_LOG10}, {"dlog10", OP_LOG10},
			{"degrees", OP_DEGREES}, {"radians", OP_RADIANS}, {"acos", OP_ACOS}, {"asin", OP_ASIN},
			{"atan", OP_ATAN}, {"cos", OP_COS}, {"sin", OP_SIN}, {"tan", OP_TAN},
		};
		size_t i;
		for (i = 0; i < sizeof(f) / sizeof(f[0]); i++)
			if (!strcmp(f[i].n, name))
			{ e->op = f[i].op; e->type_oid = STROM_FLOAT8OID; return 1; }
	}
	if (!strcmp(name, "pi") && nargs == 0)
	{ e->op = OP_PI; e->type_oid = STROM_FLOAT8OID; return 1; }
	if (nargs == 2 && a0 == STROM_FLOAT8OID && a1 == STROM_FLOAT8OID)
	{
		if (!strcmp(name, "power") || !strcmp(name, "pow") || !strcmp(name, "dpow"))
		{ e->op = OP_POW; e->type_oid = STROM_FLOAT8OID; return 1; }
		if (!strcmp(name, "atan2"))
		{ e->op = OP_ATAN2; e->type_oid = STROM_FLOAT8OID; return 1; }
	}
	/* date / time / timestamp and bpchar(1): comparisons on the raw value */
	{
		static const struct { const char *pfx; int x, y; } cmpfam[] = {
			{"date_", STROM_DATEOID, STROM_DATEOID}, {"time_", STROM_TIMEOID, STROM_TIMEOID},
			{"timestamp_", STROM_TIMESTAMPOID, STROM_TIMESTAMPOID},
			{"bpchar", STROM_BPCHAROID, STROM_BPCHAROID},
		};
		size_t i;
		for (i = 0; i < sizeof(cmpfam) / sizeof(cmpfam[0]); i++)
		{
			size_t l = strlen(cmpfam[i].pfx);
			if (nargs == 2 && !strncmp(name, cmpfam[i].pfx, l) &&
				a0 == cmpfam[i].x && a1 == cmpfam[i].y)
			{
				const char *s = name + l;
				if (!strcmp(s, "cmp")) { e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
				op = binop_by_suffix(s);
				if (op >= OP_EQ && op <= OP_GE) { e->op = op; e->type_oid = STROM_BOOLOID; return 1; }
			}
		}
		/* date <op> timestamp and the mirror image: the date is promoted */
		if (nargs == 2 && !strncmp(name, "date_", 5) && strstr(name, "_timestamp") &&
			a0 == STROM_DATEOID && a1 == STROM_TIMESTAMPOID)
		{
			char opn[8] = {0};
			size_t ol = strcspn(name + 5, "_");
			memcpy(opn, name + 5, ol < 7 ? ol : 7);
			if (!strcmp(opn, "cmp")) { e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
			op = binop_by_suffix(opn);
			if (op >= OP_EQ && op <= OP_GE) { e->op = op; e->type_oid = STROM_BOOLOID; return 1; }
		}
		if (nargs == 2 && !strncmp(name, "timestamp_", 10) && strstr(name, "_date") &&
			a0 == STROM_TIMESTAMPOID && a1 == STROM_DATEOID)
		{
			char opn[8] = {0};
			size_t ol = strcspn(name + 10, "_");
			memcpy(opn, name + 10, ol < 7 ? ol : 7);
			if (!strcmp(opn, "cmp")) { e->op = OP_CMP; e->type_oid = STROM_INT4OID; return 1; }
			op = binop_by_suffix(opn);
			if (op >= OP_EQ && op <= OP_GE) { e->op = op; e->type_oid = STROM_BOOLOID; return 1; }
		}
	}
	if (nargs == 2 && !strcmp(name, "date_pli") && a0 == STROM_DATEOID && a1 == STROM_INT4OID)
	{ e->op = OP_DATE_PLI; e->type_oid = STROM_DATEOID; return 1; }
	if (nargs == 2 && !strcmp(name, "date_mii") && a0 == STROM_DATEOID && a1 == STROM_INT4OID)
	{ e->op = OP_DATE_MII; e->type_oid = STROM_DATEOID; return 1; }
	if (nargs == 2 && !strcmp(name, "date_mi") && a0 == STROM_DATEOID && a1 == STROM_DATEOID)
	{ e->op = OP_DATE_MI; e->type_oid = STROM_INT4OID; return 1; }
	if (nargs == 2 && !strcmp(name, "integer_pl_date") && a0 == STROM_INT4OID && a1 == STROM_DATEOID)
	{ e->op = OP_INT_PL_DATE; e->type_oid = STROM_DATEOID; return 1; }
	if (nargs == 1 && !strcmp(name, "timestamp") && a0 == STROM_DATEOID)
	{ e->op = OP_DATE_TO_TS; e->type_oid = STROM_TIMESTAMPOID; return 1; }
	if (nargs == 1 && !strcmp(name, "date") && a0 == STROM_TIMESTAMPOID)
	{ e->op = OP_TS_TO_DATE; e->type_oid = STROM_DATEOID; return 1; }
	if (nargs == 1 && !strcmp(name, "time") && a0 == STROM_TIMESTAMPOID)
	{ e->op = OP_TS_TO_TIME; e->type_oid = STROM_TIMEOID; return 1; }
	if (nargs == 2 && !strcmp(name, "datetime_pl") && a0 == STROM_DATEOID && a1 == STROM_TIMEOID)
	{ e->op = OP_DATETIME_PL; e->type_oid = STROM_TIMESTAMPOID; return 1; }
	if (nargs == 2 && !strcmp(name, "timedate_pl") && a0 == STROM_TIMEOID && a1 == STROM_DATEOID)
	{ e->op = OP_TIMEDATE_PL; e->type_oid = STROM_TIMESTAMPOID; return 1; }
	return 0;
}

static oracle_expr *parse_expr(parser *ps);

static int
expect_char(parser *ps, char c)
{
	skip_ws(ps);
	if (*ps->p != c)
	{
		perr(ps, "expected '%c' near \"%.16s\"", c, ps->p);
		return 0;
	}
	ps->p++;
	return 1;
}

static int
peek_close(parser *ps)
{
	skip_ws(ps);
	return *ps->p == ')';
}

static oracle_expr *
parse_expr(parser *ps)
{
	char		head[64], buf[128];
	oracle_expr *e = NULL;

	if (!expect_char(ps, '(') || !read_atom(ps, head, sizeof(head)))
	{
		perr(ps, "expression expected");
		return NULL;
	}
	if (!strcmp(head, "const"))
	{
		int t;
		e = new_node(N_CONST);
		if (!read_atom(ps, buf, sizeof(buf)) || !(t = type_by_name(buf)))
		{ perr(ps, "bad type in const"); goto fail; }
		if (!read_atom(ps, buf, sizeof(buf)) || !parse_literal(ps, t, buf, &e->cval))
		{ perr(ps, "bad literal in const"); goto fail; }
		e->type_oid = t;
	}
	else if (!strcmp(head, "param") || !strcmp(head, "var"))
	{
		e = new_node(head[0] == 'p' ? N_PARAM : N_VAR);
		if (!read_atom(ps, buf, sizeof(buf)))
		{ perr(ps, "number expected"); goto fail; }
		e->attno = atoi(buf);
		if (!read_atom(ps, buf, sizeof(buf)))
		{ perr(ps, "bad type"); goto fail; }
		if (head[0] == 'v' && !strcmp(buf, "decimal"))
		{
			e->type_oid = STROM_NUMERICOID;
			e->is_decimal = 1;
		}
		else if (!(e->type_oid = type_by_name(buf)))
		{ perr(ps, "bad type"); goto fail; }
		/* (var ATTNO numeric SCALE): the column's typmod scale -- a promise
		 * the device emitter uses to pick fixed-point code; the value is the
		 * same, so the checker ignores it.  (var ATTNO decimal SCALE): the scale
		 * the stored integers are at */
		if (head[0] == 'v' && !peek_close(ps))
		{
			if (!read_atom(ps, buf, sizeof(buf))) { perr(ps, "scale expected"); goto fail; }
			e->dscale = atoi(buf);
		}
		if (e->is_decimal && (e->dscale < 0 || e->dscale > 18))
		{ perr(ps, "(var N decimal SCALE): scale 0..18 expected"); goto fail; }
	}
	else if (!strcmp(head, "ivar"))
	{
		/* (ivar DEPTH ATTNO TYPE): column of the matched inner tuple */
		e = new_node(N_IVAR);
		if (!read_atom(ps, buf, sizeof(buf))) { perr(ps, "depth expected"); goto fail; }
		e->depth = atoi(buf);
		if (!read_atom(ps, buf, sizeof(buf))) { perr(ps, "attno expected"); goto fail; }
		e->attno = atoi(buf);
		if (!read_atom(ps, buf, sizeof(buf)) || !(e->type_oid = type_by_name(buf)))
		{ perr(ps, "bad type"); goto fail; }
	}
	else if (!strcmp(head, "and") || !strcmp(head, "or"))
	{
		e = new_node(head[0] == 'a' ? N_AND : N_OR);
		e->type_oid = STROM_BOOLOID;
		while (!peek_close(ps) && !ps->failed)
		{
			oracle_expr *a = parse_expr(ps);
			if (!a) goto fail;
			add_arg(e, a);
		}
		if (e->nargs < 1) { perr(ps, "empty and/or"); goto fail; }
	}
	else if (!strcmp(head, "not") || !strcmp(head, "isnull") || !strcmp(head, "isnotnull") ||
			 !strncmp(head, "is_", 3))
	{
		oracle_expr *a;
		if (!strcmp(head, "not")) e = new_node(N_NOT);
		else if (!strcmp(head, "isnull")) e = new_node(N_ISNULL);
		else if (!strcmp(head, "isnotnull")) e = new_node(N_ISNOTNULL);
		else
		{
			e = new_node(N_BOOLTEST);
			if (!strcmp(head, "is_true")) e->op = BT_TRUE;
			else if (!strcmp(head, "is_not_true")) e->op = BT_NOT_TRUE;
			else if (!strcmp(head, "is_false")) e->op = BT_FALSE;
			else if (!strcmp(head, "is_not_false")) e->op = BT_NOT_FALSE;
			else if (!strcmp(head, "is_unknown")) e->op = BT_UNKNOWN;
			else if (!strcmp(head, "is_not_unknown")) e->op = BT_NOT_UNKNOWN;
			else { perr(ps, "unknown test %s", head); goto fail; }
		}
		e->type_oid = STROM_BOOLOID;
		a = parse_expr(ps);
		if (!a) goto fail;
		add_arg(e, a);
	}
	else if (!strcmp(head, "relabel"))
	{
		oracle_expr *a;
		e = new_node(N_RELABEL);
		if (!read_atom(ps, buf, sizeof(buf)) || !type_by_name(buf))
		{ perr(ps, "bad type in relabel"); goto fail; }
		a = parse_expr(ps);
		if (!a) goto fail;
		add_arg(e, a);
		e->type_oid = a->type_oid;
	}
	else if (!strcmp(head, "case") || !strcmp(head, "case_eq"))
	{
		e = new_node(N_CASE);
		if (!strcmp(head, "case_eq"))
		{
			e->case_arg = parse_expr(ps);
			if (!e->case_arg) goto fail;
		}
		while (!peek_close(ps) && !ps->failed)
		{
			char arm[16];
			if (!expect_char(ps, '(') || !read_atom(ps, arm, sizeof(arm))) goto fail;
			if (!strcmp(arm, "when"))
			{
				oracle_expr *c = parse_expr(ps), *r;
				if (!c) goto fail;
				add_arg(e, c);
				r = parse_expr(ps);
				if (!r) goto fail;
				add_arg(e, r);
				e->type_oid = r->type_oid;
			}
			else if (!strcmp(arm, "else"))
			{
				oracle_expr *r = parse_expr(ps);
				if (!r) goto fail;
				add_arg(e, r);
				e->has_else = 1;
				e->type_oid = r->type_oid;
			}
			else { perr(ps, "bad case arm %s", arm); goto fail; }
			if (!expect_char(ps, ')')) goto fail;
		}
	}
	else
	{
		e = new_node(N_FUNC);
		while (!peek_close(ps) && !ps->failed)
		{
			oracle_expr *a = parse_expr(ps);
			if (!a) goto fail;
			add_arg(e, a);
		}
		if (!resolve_func(e, head))
		{ perr(ps, "function %s is not known to the oracle for these argument types", head); goto fail; }
	}
	if (!expect_char(ps, ')'))
		goto fail;
	return e;
fail:
	oracle_expr_free(e);
	return NULL;
}

oracle_expr *
oracle_expr_parse(const char *text, char *errbuf, size_t errlen)
{
	parser ps = { text, errbuf, errlen, 0 };
	oracle_expr *e = parse_expr(&ps);
	if (e)
	{
		skip_ws(&ps);
		if (*ps.p)
		{
			perr(&ps, "trailing characters");
			oracle_expr_free(e);
			return NULL;
		}
	}
	return e;
}

/* ====================================================================== *
 * evaluation
 * ====================================================================== */
static void
set_error(int32_t *p_error, int32_t errcode)
{
	int32_t oldcode = *p_error;
	if (StromErrorIsSignificant(errcode))
	{
		if (!StromErrorIsSignificant(oldcode))
			*p_error = errcode;
	}
	else if (errcode > oldcode)
		*p_error = errcode;
}

static oracle_value
make_null(int type)
{
	oracle_value r;
	memset(&r, 0, sizeof(r));
	r.type_oid = type;
	r.isnull = 1;
	return r;
}

static oracle_value
recheck(int type, int32_t *errcode)
{
	set_error(errcode, StromError_CpuReCheck);
	return make_null(type);
}

static oracle_value
load_datum(int type, const void *addr)
{
	oracle_value r;
	memset(&r, 0, sizeof(r));
	r.type_oid = type;
	if (!addr)
	{
		r.isnull = 1;
		return r;
	}
	switch (type)
	{
		case STROM_BOOLOID: case STROM_BPCHAROID:
			{ int8_t x; memcpy(&x, addr, 1); r.v.i = x; } break;
		case STROM_INT2OID:
			{ int16_t x; memcpy(&x, addr, 2); r.v.i = x; } break;
		case STROM_INT4OID: case STROM_DATEOID:
			{ int32_t x; memcpy(&x, addr, 4); r.v.i = x; } break;
		case STROM_FLOAT4OID:
			memcpy(&r.v.f, addr, 4); break;
		case STROM_FLOAT8OID:
			memcpy(&r.v.d, addr, 8); break;
		default:
			memcpy(&r.v.i, addr, 8); break;
	}
	return r;
}

static double
as_double(oracle_value v)
{
	if (v.type_oid == STROM_FLOAT4OID) return (double)v.v.f;
	if (v.type_oid == STROM_FLOAT8OID) return v.v.d;
	return (double)v.v.i;
}

static void
int_range(int type, __int128 *lo, __int128 *hi)
{
	switch (type)
	{
		case STROM_INT2OID: *lo = -32768; *hi = 32767; break;
		case STROM_INT4OID: case STROM_DATEOID: *lo = -2147483648LL; *hi = 2147483647LL; break;
		default: *lo = -(__int128)9223372036854775807LL - 1; *hi = 9223372036854775807LL; break;
	}
}

/* PostgreSQL float ordering: NaN sorts above everything, equals itself */
static int
float_cmp(double x, double y)
{
	if (isnan(x))
		return isnan(y) ? 0 : 1;
	if (isnan(y))
		return -1;
	return x < y ? -1 : (x > y ? 1 : 0);
}

static oracle_value
store_float(int type, double v)
{
	oracle_value r;
	memset(&r, 0, sizeof(r));
	r.type_oid = type;
	if (type == STROM_FLOAT4OID) r.v.f = (float)v; else r.v.d = v;
	return r;
}

/* CHECKFLOATVAL (opencl_mathlib.h:24-28) */
static int
float_bad(double val, int inf_is_valid, int zero_is_valid)
{
	return (isinf(val) && !inf_is_valid) || (val == 0.0 && !zero_is_valid);
}

static oracle_value
eval_func(const oracle_expr *e, oracle_value *a, int32_t *errcode)
{
	int		rt = e->type_oid;
	int		i;
	oracle_value r;

	memset(&r, 0, sizeof(r));
	r.type_oid = rt;
	/* strict functions: any NULL argument -> NULL */
	for (i = 0; i < e->nargs; i++)
		if (a[i].isnull)
			return make_null(rt);

	switch (e->op)
	{
		case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV:
			if (type_is_float(rt))
			{
				/* computed in the result type, like the device code */
				double x = as_double(a[0]), y = as_double(a[1]), z;
				if (rt == STROM_FLOAT4OID)
				{
					float fx = (float)x, fy = (float)y, fz;
					if (e->op == OP_DIV && fy == 0.0f) return recheck(rt, errcode);
					fz = e->op == OP_ADD ? fx + fy : e->op == OP_SUB ? fx - fy :
						e->op == OP_MUL ? fx * fy : fx / fy;
					z = fz;
				}
				else
				{
					if (e->op == OP_DIV && y == 0.0) return recheck(rt, errcode);
					z = e->op == OP_ADD ? x + y : e->op == OP_SUB ? x - y :
						e->op == OP_MUL ? x * y : x / y;
				}
				if (float_bad(z, isinf(x) || isinf(y),
							  e->op == OP_MUL ? (x == 0.0 || y == 0.0) :
							  e->op == OP_DIV ? (x == 0.0) : 1))
					return recheck(rt, errcode);
				return store_float(rt, z);
			}
			else
			{
				__int128 x = a[0].v.i, y = a[1].v.i, z, lo, hi;
				int_range(rt, &lo, &hi);
				if (e->op == OP_DIV)
				{
					if (y == 0) return recheck(rt, errcode);
					z = x / y;
				}
				else
					z = e->op == OP_ADD ? x + y : e->op == OP_SUB ? x - y : x * y;
				if (z < lo || z > hi)
					return recheck(rt, errcode);
				r.v.i = (int64_t)z;
				return r;
			}
		case OP_MOD:
			if (a[1].v.i == 0) return recheck(rt, errcode);
			r.v.i = (a[1].v.i == -1 ? 0 : a[0].v.i % a[1].v.i);
			return r;
		case OP_EQ: case OP_NE: case OP_LT: case OP_LE: case OP_GT: case OP_GE: case OP_CMP:
			{
				int c;
				int t0 = a[0].type_oid, t1 = a[1].type_oid;
				if (type_is_varlena(t0))
					c = varlena_compare((const void *)(intptr_t)a[0].v.i, (const void *)(intptr_t)a[1].v.i,
										t0 == STROM_BPCHARNOID);
				else if (t0 == STROM_NUMERICOID)
					c = num_cmp(a[0].v.u, a[1].v.u);
				else if (type_is_float(t0) || type_is_float(t1))
					c = float_cmp(as_double(a[0]), as_double(a[1]));
				else if ((t0 == STROM_DATEOID && t1 == STROM_TIMESTAMPOID) ||
						 (t0 == STROM_TIMESTAMPOID && t1 == STROM_DATEOID))
				{
					/* the date is promoted (pgfn_date_timestamp,
					 * opencl_timelib.h:289-320): +-infinity kept, a
					 * product outside int64 goes back to the CPU */
					int64_t l, rr, ts;
					int		di = (t0 == STROM_DATEOID ? 0 : 1);
					int64_t	dv = a[di].v.i;
					if (dv == INT32_MIN) ts = INT64_MIN;
					else if (dv == INT32_MAX) ts = INT64_MAX;
					else if (__builtin_mul_overflow(dv, (int64_t)86400000000LL, &ts))
						return recheck(rt, errcode);
					l = (di == 0 ? ts : a[0].v.i);
					rr = (di == 0 ? a[1].v.i : ts);
					c = l < rr ? -1 : l > rr ? 1 : 0;
				}
				else if (t0 == STROM_BOOLOID)
					c = ((a[0].v.i != 0) > (a[1].v.i != 0)) - ((a[0].v.i != 0) < (a[1].v.i != 0));
				else if (t0 == STROM_BPCHAROID)
					c = ((uint8_t)a[0].v.i > (uint8_t)a[1].v.i) - ((uint8_t)a[0].v.i < (uint8_t)a[1].v.i);
				else
					c = (a[0].v.i > a[1].v.i) - (a[0].v.i < a[1].v.i);
				switch (e->op)
				{
					case OP_EQ: r.v.i = (c == 0); break;
					case OP_NE: r.v.i = (c != 0); break;
					case OP_LT: r.v.i = (c < 0); break;
					case OP_LE: r.v.i = (c <= 0); break;
					case OP_GT: r.v.i = (c > 0); break;
					case OP_GE: r.v.i = (c >= 0); break;
					default:    r.v.i = c; break;
				}
				return r;
			}
		case OP_UPLUS:
			r = a[0];
			return r;
		case OP_UMINUS: case OP_ABS:
			if (type_is_float(rt))
			{
				double x = as_double(a[0]);
				return store_float(rt, e->op == OP_UMINUS ? -x : fabs(x));
			}
			else
			{
				__int128 x = a[0].v.i, lo, hi;
				int_range(rt, &lo, &hi);
				if (e->op == OP_UMINUS || x < 0)
					x = -x;
				if (x < lo || x > hi)
					return recheck(rt, errcode);
				r.v.i = (int64_t)x;
				return r;
			}
		case OP_BITNOT:
			r.v.i = (rt == STROM_INT2OID ? (int16_t)~a[0].v.i :
					 rt == STROM_INT4OID ? (int32_t)~a[0].v.i : ~a[0].v.i);
			return r;
		case OP_BITAND: r.v.i = a[0].v.i & a[1].v.i; return r;
		case OP_BITOR:  r.v.i = a[0].v.i | a[1].v.i; return r;
		case OP_BITXOR: r.v.i = a[0].v.i ^ a[1].v.i; return r;
		case OP_SHL:
			if (rt == STROM_INT2OID) r.v.i = (int16_t)((int32_t)a[0].v.i << (a[1].v.i & 31));
			else if (rt == STROM_INT4OID) r.v.i = (int32_t)((uint32_t)a[0].v.i << (a[1].v.i & 31));
			else r.v.i = (int64_t)((uint64_t)a[0].v.i << (a[1].v.i & 63));
			return r;
		case OP_SHR:
			if (rt == STROM_INT8OID) r.v.i = a[0].v.i >> (a[1].v.i & 63);
			else r.v.i = a[0].v.i >> (a[1].v.i & 31);
			if (rt == STROM_INT2OID) r.v.i = (int16_t)r.v.i;
			return r;
		case OP_CAST:
			{
				int st = a[0].type_oid;
				if (type_is_float(rt))
				{
					if (st == STROM_FLOAT8OID && rt == STROM_FLOAT4OID)
					{
						float f = (float)a[0].v.d;
						if (float_bad(f, isinf(a[0].v.d), a[0].v.d == 0.0))
							return recheck(rt, errcode);
						r.v.f = f;
						return r;
					}
					return store_float(rt, as_double(a[0]));
				}
				else
				{
					__int128 lo, hi;
					int_range(rt, &lo, &hi);
					if (type_is_float(st))
					{
						/* dtoi4 / dtoi8: rint() then range check */
						double x = rint(as_double(a[0]));
						if (isnan(x) || x < (double)lo || x >= (double)hi + 1.0)
							return recheck(rt, errcode);
						r.v.i = (int64_t)x;
						return r;
					}
					if (a[0].v.i < lo || a[0].v.i > hi)
						return recheck(rt, errcode);
					r.v.i = (st == STROM_BOOLOID ? (a[0].v.i != 0) : a[0].v.i);
					return r;
				}
			}
		case OP_CEIL:  r.v.d = ceil(a[0].v.d); return r;
		case OP_FLOOR: r.v.d = floor(a[0].v.d); return r;
		case OP_ROUND: r.v.d = rint(a[0].v.d); return r;
		case OP_TRUNC: r.v.d = trunc(a[0].v.d); return r;
		case OP_SIGN:  r.v.d = (a[0].v.d > 0) - (a[0].v.d < 0); return r;
		case OP_SQRT:
			if (a[0].v.d < 0) return recheck(rt, errcode);
			r.v.d = sqrt(a[0].v.d);
			return r;
		case OP_PI:    r.v.d = 3.14159265358979323846; return r;
		/* PostgreSQL 9.4 float.c: a domain error, an infinite result of finite arguments or
		 * an underflow to zero is an ERROR there -- CpuReCheck here (CHECKFLOATVAL) */
#define ORACLE_MATH1(OPC, BAD, EXPR, INF_OK, ZERO_OK)							\
		case OPC:																\
			{																	\
				double x = a[0].v.d;											\
				if (BAD) return recheck(rt, errcode);							\
				r.v.d = (EXPR);													\
				if ((isinf(r.v.d) && !(INF_OK)) || (r.v.d == 0.0 && !(ZERO_OK)))	\
					return recheck(rt, errcode);								\
				return r;														\
			}
		ORACLE_MATH1(OP_CBRT, 0, cbrt(x), isinf(x), x == 0.0)
		ORACLE_MATH1(OP_EXP, 0, exp(x), isinf(x), 0)
		ORACLE_MATH1(OP_LN, x <= 0.0, log(x), isinf(x), x == 1.0)
		ORACLE_MATH1(OP_LOG10, x <= 0.0, log10(x), isinf(x), x == 1.0)
		ORACLE_MATH1(OP_DEGREES, 0, x * (180.0 / 3.14159265358979323846), isinf(x), x == 0.0)
		ORACLE_MATH1(OP_RADIANS, 0, x * (3.14159265358979323846 / 180.0), isinf(x), x == 0.0)
		ORACLE_MATH1(OP_ACOS, (x < -1.0 || x > 1.0), acos(x), 0, 1)
		ORACLE_MATH1(OP_ASIN, (x < -1.0 || x > 1.0), asin(x), 0, 1)
		ORACLE_MATH1(OP_ATAN, 0, atan(x), 0, 1)
		ORACLE_MATH1(OP_COS, isinf(x), cos(x), 0, 1)
		ORACLE_MATH1(OP_SIN, isinf(x), sin(x), 0, 1)
		ORACLE_MATH1(OP_TAN, isinf(x), tan(x), 0, 1)
#undef ORACLE_MATH1
		case OP_POW:
			{
				double x = a[0].v.d, y = a[1].v.d;
				if ((x == 0.0 && y < 0.0) || (x < 0.0 && floor(y) != y))
					return recheck(rt, errcode);
				r.v.d = pow(x, y);
				if ((isinf(r.v.d) && !(isinf(x) || isinf(y))) || (r.v.d == 0.0 && x != 0.0))
					return recheck(rt, errcode);
				return r;
			}
		case OP_ATAN2: r.v.d = atan2(a[0].v.d, a[1].v.d); return r;
		case OP_DATE_PLI: case OP_DATE_MII: case OP_INT_PL_DATE: case OP_DATE_MI:
			{
				__int128 x = a[0].v.i, y = a[1].v.i, z, lo, hi;
				int_range(STROM_INT4OID, &lo, &hi);
				z = (e->op == OP_DATE_MII || e->op == OP_DATE_MI) ? x - y : x + y;
				if (z < lo || z > hi)
					return recheck(rt, errcode);
				r.v.i = (int64_t)z;
				return r;
			}
		case OP_DATE_TO_TS:
			if (a[0].v.i == INT32_MIN) r.v.i = INT64_MIN;
			else if (a[0].v.i == INT32_MAX) r.v.i = INT64_MAX;
			else if (__builtin_mul_overflow(a[0].v.i, (int64_t)86400000000LL, &r.v.i))
				return recheck(rt, errcode);
			return r;
		case OP_TS_TO_DATE:
			if (a[0].v.i == INT64_MIN) { r.v.i = INT32_MIN; return r; }
			if (a[0].v.i == INT64_MAX) { r.v.i = INT32_MAX; return r; }
			{
				int64_t t = a[0].v.i, d = t / 86400000000LL;
				if (t % 86400000000LL < 0) d--;
				r.v.i = d;
				return r;
			}
		case OP_TS_TO_TIME:
			if (a[0].v.i == INT64_MIN || a[0].v.i == INT64_MAX)
				return make_null(rt);
			{
				int64_t t = a[0].v.i % 86400000000LL;
				if (t < 0) t += 86400000000LL;
				r.v.i = t;
				return r;
			}
		case OP_NUM_ADD: case OP_NUM_SUB:
			{
				uint64_t b = a[1].v.u;
				if (e->op == OP_NUM_SUB && NUM_MANT(b) != 0) b ^= (1ULL << 57);
				if (!num_add(a[0].v.u, b, &r.v.u)) return recheck(rt, errcode);
				return r;
			}
		case OP_NUM_MUL:
			if (!num_mul(a[0].v.u, a[1].v.u, &r.v.u)) return recheck(rt, errcode);
			return r;
		case OP_NUM_UPLUS:
			r.v.u = a[0].v.u;
			return r;
		case OP_NUM_UMINUS:
			r.v.u = a[0].v.u;
			if (NUM_MANT(r.v.u) != 0) r.v.u ^= (1ULL << 57);
			return r;
		case OP_NUM_ABS:
			r.v.u = a[0].v.u & ~(1ULL << 57);
			return r;
		case OP_IDENTITY:
			r.v = a[0].v;
			return r;
		case OP_NUM_FROM_FLOAT:
			{
				/*
				 * PostgreSQL: the value printed with FLT_DIG / DBL_DIG significant digits and read
				 * back (float4_numeric / float8_numeric, utils/adt/numeric.c); the reference:
				 * float_to_numeric (opencl_numeric.h:625-738) with the same digit counts, through
				 * log10 / exp10 in floating point (a few per cent of all doubles come out one off in
				 * the last digit that way).  Stated here as PostgreSQL's definition: the binary value
				 * M x 2^E scaled by 10^k in exact 192-bit integer arithmetic and rounded half to even
				 * once; NaN, infinities and what the 64-bit form cannot hold are CpuReCheck.
				 */
				int		dig = (e->args[0]->type_oid == STROM_FLOAT4OID ? 6 : 15);
				double	value = as_double(a[0]);
				uint64_t bits, M, lim = 1, mant = 0;
				int		sign, be, E, e2, e10, k, i, turn, ok = 0;
				if (isnan(value) || isinf(value)) return recheck(rt, errcode);
				if (value == 0.0) { r.v.u = 0; return r; }
				memcpy(&bits, &value, 8);
				sign = (int)(bits >> 63);
				be = (int)((bits >> 52) & 0x7ff);
				M = (bits & 0xfffffffffffffULL) | (be ? (1ULL << 52) : 0);
				E = (be ? be : 1) - 1075;
				e2 = 63 - __builtin_clzll(M) + E;
				e10 = (e2 * 1233) >> 12;
				k = dig - 1 - e10;
				for (i = 0; i < dig; i++) lim *= 10;
				for (turn = 0; turn < 4; turn++)
				{
					if (!oracle_scaled_mantissa(M, E, k, &mant)) break;
					if (mant > lim) k--;
					else if (mant < lim / 10) k++;
					else { ok = 1; break; }
				}
				if (!ok || !num_pack(-k, sign, mant, &r.v.u)) return recheck(rt, errcode);
				return r;
			}
		case OP_NUM_FROM_INT:
			{
				int64_t v = a[0].v.i;
				uint64_t mant = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
				if (!num_pack(0, v < 0, mant, &r.v.u)) return recheck(rt, errcode);
				return r;
			}
		case OP_NUM_TO_INT:
			{
				/* round half away from zero, then range check
				 * (numeric_to_integer, opencl_numeric.h:400-460) */
				uint64_t u = a[0].v.u, mant = NUM_MANT(u);
				int expo = NUM_EXPO(u);
				__int128 lo, hi, val;
				int_range(rt, &lo, &hi);
				if (expo < 0)
				{
					uint64_t mag = num_pow10(-expo);
					mant = mag ? (mant + mag / 2) / mag : 0;
				}
				else if (expo > 0)
				{
					uint64_t mag = num_pow10(expo);
					if (mag == 0 || __builtin_mul_overflow(mant, mag, &mant)) return recheck(rt, errcode);
				}
				val = NUM_SIGN(u) ? -(__int128)mant : (__int128)mant;
				if (val < lo || val > hi) return recheck(rt, errcode);
				r.v.i = (int64_t)val;
				return r;
			}
		case OP_NUM_TO_FLOAT:
			{
				uint64_t u = a[0].v.u;
				int expo = NUM_EXPO(u), k;
				double m = (double)NUM_MANT(u), p = 1.0, d;
				for (k = 0; k < (expo < 0 ? -expo : expo); k++) p *= 10.0;
				d = expo < 0 ? m / p : m * p;
				if (NUM_SIGN(u)) d = -d;
				if (rt == STROM_FLOAT4OID)
				{
					float f = (float)d;
					if (float_bad(f, 0, d == 0.0)) return recheck(rt, errcode);
					r.v.f = f;
				}
				else
					r.v.d = d;
				return r;
			}
		case OP_DATETIME_PL: case OP_TIMEDATE_PL:
			{
				int64_t dv = (e->op == OP_DATETIME_PL ? a[0].v.i : a[1].v.i);
				int64_t tv = (e->op == OP_DATETIME_PL ? a[1].v.i : a[0].v.i);
				if (dv == INT32_MIN) { r.v.i = INT64_MIN; return r; }
				if (dv == INT32_MAX) { r.v.i = INT64_MAX; return r; }
				if (__builtin_mul_overflow(dv, (int64_t)86400000000LL, &r.v.i) ||
					__builtin_add_overflow(r.v.i, tv, &r.v.i))
					return recheck(rt, errcode);
				return r;
			}
	}
	return make_null(rt);
}

typedef struct {
	const kern_data_store *kds;
	uint32_t		rowidx;
	const uint64_t *ext_values;
	const uint8_t  *ext_isnull;
	int				n_ext;
	/* hash join: the inner relations and the inner row matched so far */
	const kern_data_store *const *inner_kds;
	const uint32_t *inner_row;
} eval_ctx;

static oracle_value
eval_node(const oracle_expr *e, const eval_ctx *cx, int32_t *errcode)
{
	oracle_value r, a[4];
	int		i;

	switch (e->kind)
	{
		case N_CONST:
			return e->cval;
		case N_PARAM:
			if (e->attno < 0 || e->attno >= cx->n_ext ||
				(cx->ext_isnull && cx->ext_isnull[e->attno]))
				return make_null(e->type_oid);
			if (type_is_varlena(e->type_oid))
			{
				/* the external value is the address of a varlena datum */
				const void *vl = (const void *)(uintptr_t)cx->ext_values[e->attno];
				if (!vl)
					return make_null(e->type_oid);
				if (!varlena_readable(vl))
					return recheck(e->type_oid, errcode);
				r = make_null(e->type_oid);
				r.isnull = 0;
				r.v.i = (int64_t)(intptr_t)vl;
				return r;
			}
			return load_datum(e->type_oid, &cx->ext_values[e->attno]);
		case N_VAR:
		case N_IVAR:
			{
				const kern_data_store *k = (e->kind == N_VAR ? cx->kds : cx->inner_kds[e->depth - 1]);
				uint32_t	row = (e->kind == N_VAR ? cx->rowidx : cx->inner_row[e->depth - 1]);
				const void *addr = oracle_get_datum(k, e->attno - 1, row);

				/* a numeric inside a heap tuple is PostgreSQL's varlena form */
				if (addr && e->type_oid == STROM_NUMERICOID &&
					(uint32_t)(e->attno - 1) < k->ncols && k->colmeta[e->attno - 1].attlen < 0)
				{
					uint64_t	image;
					if (!oracle_numeric_from_varlena(addr, &image))
						return recheck(e->type_oid, errcode);
					return load_datum(e->type_oid, &image);
				}
				if (e->kind == N_VAR && e->is_decimal)
				{
					int64_t		v;
					uint64_t	image;
					if (!addr)
						return make_null(STROM_NUMERICOID);
					memcpy(&v, addr, 8);
					if (!num_pack(-e->dscale, v < 0, v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v, &image))
						return recheck(STROM_NUMERICOID, errcode);
					return load_datum(STROM_NUMERICOID, &image);
				}
				if (type_is_varlena(e->type_oid))
				{
					if (!addr)
						return make_null(e->type_oid);
					if ((uint32_t)(e->attno - 1) >= k->ncols || k->colmeta[e->attno - 1].attlen >= 0)
					{
						set_error(errcode, StromError_DataStoreCorruption);
						return make_null(e->type_oid);
					}
					if (!varlena_readable(addr))
						return recheck(e->type_oid, errcode);
					r = make_null(e->type_oid);
					r.isnull = 0;
					r.v.i = (int64_t)(intptr_t)addr;
					return r;
				}
				return load_datum(e->type_oid, addr);
			}
		case N_RELABEL:
			return eval_node(e->args[0], cx, errcode);
		case N_FUNC:
			for (i = 0; i < e->nargs && i < 4; i++)
				a[i] = eval_node(e->args[i], cx, errcode);
			return eval_func(e, a, errcode);
		case N_AND: case N_OR:
			{
				/* SQL three-valued logic; every argument is evaluated (the
				 * device code does not short-circuit either, so error side
				 * effects of later arguments are kept) */
				int anynull = 0, decided = 0;
				for (i = 0; i < e->nargs; i++)
				{
					oracle_value x = eval_node(e->args[i], cx, errcode);
					if (x.isnull) anynull = 1;
					else if (e->kind == N_AND ? !x.v.i : x.v.i != 0) decided = 1;
				}
				memset(&r, 0, sizeof(r));
				r.type_oid = STROM_BOOLOID;
				if (decided) r.v.i = (e->kind == N_OR);
				else if (anynull) r.isnull = 1;
				else r.v.i = (e->kind == N_AND);
				return r;
			}
		case N_NOT:
			r = eval_node(e->args[0], cx, errcode);
			if (!r.isnull) r.v.i = !r.v.i;
			return r;
		case N_ISNULL: case N_ISNOTNULL:
			{
				oracle_value x = eval_node(e->args[0], cx, errcode);
				memset(&r, 0, sizeof(r));
				r.type_oid = STROM_BOOLOID;
				r.v.i = (e->kind == N_ISNULL ? x.isnull : !x.isnull);
				return r;
			}
		case N_BOOLTEST:
			{
				oracle_value x = eval_node(e->args[0], cx, errcode);
				int t = (!x.isnull && x.v.i), f = (!x.isnull && !x.v.i);
				memset(&r, 0, sizeof(r));
				r.type_oid = STROM_BOOLOID;
				switch (e->op)
				{
					case BT_TRUE: r.v.i = t; break;
					case BT_NOT_TRUE: r.v.i = !t; break;
					case BT_FALSE: r.v.i = f; break;
					case BT_NOT_FALSE: r.v.i = !f; break;
					case BT_UNKNOWN: r.v.i = x.isnull; break;
					default: r.v.i = !x.isnull; break;
				}
				return r;
			}
		case N_CASE:
			{
				int npairs = (e->nargs - e->has_else) / 2;
				oracle_value subj;
				memset(&subj, 0, sizeof(subj));
				if (e->case_arg)
					subj = eval_node(e->case_arg, cx, errcode);
				for (i = 0; i < npairs; i++)
				{
					oracle_value c = eval_node(e->args[2 * i], cx, errcode);
					int hit;
					if (e->case_arg)
					{
						if (subj.isnull || c.isnull) hit = 0;
						else if (type_is_float(subj.type_oid) || type_is_float(c.type_oid))
							hit = float_cmp(as_double(subj), as_double(c)) == 0;
						else hit = (subj.v.i == c.v.i);
					}
					else
						hit = (!c.isnull && c.v.i);
					if (hit)
						return eval_node(e->args[2 * i + 1], cx, errcode);
				}
				if (e->has_else)
					return eval_node(e->args[e->nargs - 1], cx, errcode);
				return make_null(e->type_oid);
			}
	}
	return make_null(e->type_oid);
}

oracle_value
oracle_expr_eval(const oracle_expr *expr, const kern_data_store *kds, uint32_t rowidx,
				 const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
				 int32_t *errcode)
{
	eval_ctx cx = { kds, rowidx, ext_values, ext_isnull, n_ext, NULL, NULL };
	return eval_node(expr, &cx, errcode);
}

/* ====================================================================== *
 * GpuScan: gpuscan_qual + gpuscan_writeback_row_error
 *          (opencl_gpuscan.h:98-177)
 * ====================================================================== */
int32_t
oracle_gpuscan(const char *qual,
			   const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
			   const kern_data_store *kds, const kern_row_map *krowmap,
			   int32_t *results, uint32_t *p_nitems,
			   char *errbuf, size_t errlen)
{
	oracle_expr *expr = oracle_expr_parse(qual, errbuf, errlen);
	int32_t		chunk_error = StromError_Success;
	uint32_t	nitems = 0, i, nrows;
	int			use_map = (krowmap && krowmap->nvalids >= 0);

	if (!expr)
		return StromError_BadRequestMessage;
	if (expr->type_oid != STROM_BOOLOID)
	{
		snprintf(errbuf, errlen, "qualifier is not boolean");
		oracle_expr_free(expr);
		return StromError_BadRequestMessage;
	}
	nrows = use_map ? (uint32_t)krowmap->nvalids : kds->nitems;
	for (i = 0; i < nrows; i++)
	{
		uint32_t	row = use_map ? (uint32_t)krowmap->rindex[i] : i;
		int32_t		errcode = StromError_Success;
		oracle_value rc = oracle_expr_eval(expr, kds, row, ext_values, ext_isnull, n_ext, &errcode);

		set_error(&errcode, (!rc.isnull && rc.v.i != 0)
				  ? StromError_Success : StromError_RowFiltered);
		if (errcode == StromError_Success)
			results[nitems++] = (int32_t)(row + 1);
		else if (errcode == StromError_CpuReCheck)
			results[nitems++] = -(int32_t)(row + 1);
		else if (StromErrorIsSignificant(errcode) && chunk_error == StromError_Success)
			chunk_error = errcode;
	}
	*p_nitems = nitems;
	oracle_expr_free(expr);
	return chunk_error;
}

void
oracle_get_layout(oracle_layout *out)
{
	out->sizeof_kern_data_store_head = offsetof(kern_data_store, colmeta);
	out->sizeof_kern_colmeta = sizeof(kern_colmeta);
	out->sizeof_kern_rowitem = sizeof(kern_rowitem);
	out->sizeof_kern_blkitem = sizeof(kern_blkitem);
	out->offsetof_resultbuf_results = offsetof(kern_resultbuf, results);
	out->sizeof_kern_parambuf_head = offsetof(kern_parambuf, poffset);
	out->sizeof_kern_hashentry = sizeof(kern_hashentry);
	out->offsetof_hashentry_htup = offsetof(kern_hashentry, htup);
	out->offsetof_htup_t_bits = offsetof(HeapTupleHeaderData, t_bits);
	out->sizeof_kern_multihash_head = offsetof(kern_multihash, htable_offset);
	out->offsetof_gpupreagg_kparams = offsetof(kern_gpupreagg, kparams);
	out->sizeof_kern_coldir = sizeof(kern_coldir);
}

/* ====================================================================== *
 * GpuPreAgg: one chunk -> one partial row per group
 *
 * Restates the net effect of gpupreagg_preparation + sort + reduction
 * (opencl_gpupreagg.h:380-608) for one chunk: rows failing the pulled-up
 * qual are dropped; per surviving row the partial inputs are
 *     nrows(args..)  1 when every argument is TRUE else 0   -> summed
 *     psum(x)        x, NULLs ignored, NULL when no input    -> summed
 *     pmin/pmax(x)   likewise                                -> min / max
 * (gpupreagg.c:1495-1748; PSUM/PMIN/PMAX rules opencl_gpupreagg.h:862-987);
 * groups are formed on the key values with NULL keys grouped together
 * (keycomp, gpupreagg.c:1230-1249).  Any CpuReCheck raised by a row makes
 * the whole chunk CpuReCheck (gpupreagg.c:2746-2750): status 2, no rows.
 * Sums are accumulated sequentially in row order: int8 exactly (with the
 * reference's overflow -> CpuReCheck rule), floats in double.
 * ====================================================================== */
enum { T_KEY = 1, T_NROWS, T_PSUM, T_PMIN, T_PMAX };

typedef struct {
	int			kind;
	int			type_oid;
	int			nexprs;
	oracle_expr *exprs[4];
	int			pcov;			/* 0 none, 1 x, 2 y, 3 x2, 4 y2, 5 xy */
	int			x2;				/* psum_x2 */
	int			numeric_scale;	/* >= 0: numeric partial kept as int8 at 10^-scale */
} preagg_target;

typedef struct {
	oracle_expr *qual;
	int			ntargets;
	preagg_target targets[64];
} preagg_spec;

static void
preagg_spec_free(preagg_spec *sp)
{
	int i, j;
	oracle_expr_free(sp->qual);
	for (i = 0; i < sp->ntargets; i++)
		for (j = 0; j < sp->targets[i].nexprs; j++)
			oracle_expr_free(sp->targets[i].exprs[j]);
}

static int
preagg_spec_parse(const char *text, preagg_spec *sp, char *errbuf, size_t errlen)
{
	parser ps = { text, errbuf, errlen, 0 };
	char	head[32];

	memset(sp, 0, sizeof(*sp));
	if (!expect_char(&ps, '(') || !read_atom(&ps, head, sizeof(head)) || strcmp(head, "gpupreagg"))
	{
		perr(&ps, "(gpupreagg ...) expected");
		return 0;
	}
	while (!peek_close(&ps) && !ps.failed)
	{
		preagg_target *t;
		if (!expect_char(&ps, '(') || !read_atom(&ps, head, sizeof(head)))
			break;
		if (!strcmp(head, "qual"))
		{
			sp->qual = parse_expr(&ps);
			if (!sp->qual || !expect_char(&ps, ')'))
				break;
			continue;
		}
		if (sp->ntargets >= 64)
		{
			perr(&ps, "too many targets");
			break;
		}
		t = &sp->targets[sp->ntargets++];
		if (!strcmp(head, "key")) t->kind = T_KEY;
		else if (!strcmp(head, "nrows")) t->kind = T_NROWS;
		else if (!strcmp(head, "psum")) t->kind = T_PSUM;
		else if (!strcmp(head, "pmin")) t->kind = T_PMIN;
		else if (!strcmp(head, "pmax")) t->kind = T_PMAX;
		else if (!strcmp(head, "psum_x2")) { t->kind = T_PSUM; t->x2 = 1; }
		else if (!strncmp(head, "pcov_", 5))
		{
			t->kind = T_PSUM;
			t->pcov = !strcmp(head + 5, "x") ? 1 : !strcmp(head + 5, "y") ? 2 :
				!strcmp(head + 5, "x2") ? 3 : !strcmp(head + 5, "y2") ? 4 :
				!strcmp(head + 5, "xy") ? 5 : -1;
			if (t->pcov < 0) { perr(&ps, "unknown target %s", head); break; }
		}
		else { perr(&ps, "unknown target %s", head); break; }
		t->numeric_scale = -1;
		while (!peek_close(&ps) && !ps.failed && t->nexprs < 4)
		{
			skip_ws(&ps);
			if (*ps.p != '(')
			{
				/* trailing atom: the scale of a numeric partial */
				char sb[16];
				if (!read_atom(&ps, sb, sizeof(sb))) { perr(&ps, "scale expected"); break; }
				t->numeric_scale = atoi(sb);
				continue;
			}
			t->exprs[t->nexprs] = parse_expr(&ps);
			if (!t->exprs[t->nexprs])
				break;
			t->nexprs++;
		}
		if (ps.failed || !expect_char(&ps, ')'))
			break;
		if (t->kind == T_NROWS) t->type_oid = STROM_INT4OID;
		else if (t->x2 && t->nexprs == 1 && t->exprs[0]->type_oid == STROM_NUMERICOID)
			t->type_oid = STROM_NUMERICOID;		/* sum of squares in the 64-bit numeric form */
		else if (t->x2 || t->pcov) t->type_oid = STROM_FLOAT8OID;
		else if (t->nexprs == 1) t->type_oid = t->exprs[0]->type_oid;
		else { perr(&ps, "%s takes one argument", head); break; }
		if (t->type_oid == STROM_NUMERICOID && t->kind != T_KEY)
		{
			if (t->numeric_scale > 32)
			{ perr(&ps, "numeric scale out of range in %s", head); break; }
			/* with a scale: accumulated as fixed point int8; without: in the 64-bit
			 * numeric form itself (GPUPREAGG_AGGCALC_PSUM_NUMERIC, opencl_gpupreagg.h:965-987) */
			if (t->numeric_scale >= 0)
				t->type_oid = STROM_INT8OID;
		}
		else
			t->numeric_scale = -1;
		if (t->pcov && t->nexprs != 3) { perr(&ps, "pcov takes (filter x y)"); break; }
	}
	if (!ps.failed)
		expect_char(&ps, ')');
	if (ps.failed)
	{
		preagg_spec_free(sp);
		return 0;
	}
	return 1;
}

typedef struct {
	int64_t	   *keyval;		/* [nkeys] */
	uint8_t	   *keynull;
	oracle_value *acc;		/* [naggs] */
} preagg_group;

static double
float_of(oracle_value v)
{
	return v.type_oid == STROM_FLOAT4OID ? (double)v.v.f : v.v.d;
}

int32_t
oracle_gpupreagg(const char *spec_text,
				 const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
				 const kern_data_store *kds, const kern_row_map *krowmap,
				 uint32_t max_groups,
				 uint64_t *out_values,	/* [max_groups * ntargets] raw 8-byte images */
				 uint8_t *out_isnull,	/* [max_groups * ntargets] */
				 uint32_t *p_ngroups,
				 char *errbuf, size_t errlen)
{
	preagg_spec	sp;
	int			nkeys = 0, naggs = 0, key_of[64], agg_of[64];
	preagg_group *groups = NULL;
	uint32_t	ngroups = 0, cap = 0, i, nrows;
	uint32_t   *hidx = NULL, hcap = 0, gcap = 0;	/* hash index: slot -> group number */
	uint64_t   *hval = NULL, *ghash = NULL;
	int			t, use_map = (krowmap && krowmap->nvalids >= 0);
	int32_t		status = StromError_Success;

	*p_ngroups = 0;
	if (!preagg_spec_parse(spec_text, &sp, errbuf, errlen))
		return StromError_BadRequestMessage;
	for (t = 0; t < sp.ntargets; t++)
	{
		if (sp.targets[t].kind == T_KEY) key_of[nkeys++] = t;
		else agg_of[naggs++] = t;
	}
	nrows = use_map ? (uint32_t)krowmap->nvalids : kds->nitems;
	for (i = 0; i < nrows && status == StromError_Success; i++)
	{
		uint32_t	row = use_map ? (uint32_t)krowmap->rindex[i] : i;
		int32_t		errcode = StromError_Success;
		int64_t		kv[64];
		uint8_t		kn[64];
		oracle_value av[64];
		uint32_t	g;
		int			k, a;

		if (sp.qual)
		{
			oracle_value rc = oracle_expr_eval(sp.qual, kds, row, ext_values, ext_isnull, n_ext, &errcode);
			if (errcode == StromError_Success && (rc.isnull || !rc.v.i))
				continue;
		}
		for (k = 0; k < nkeys; k++)
		{
			oracle_value v = oracle_expr_eval(sp.targets[key_of[k]].exprs[0], kds, row,
											  ext_values, ext_isnull, n_ext, &errcode);
			kn[k] = (uint8_t)v.isnull;
			/* group identity is the key's CANONICAL image (what the type's own
			 * comparator calls equal, gpupreagg_keycomp opencl_gpupreagg.h:236):
			 * float keys as float8 with -0 = +0 and one NaN, numerics stripped */
			if (v.isnull)
				kv[k] = 0;
			else if (type_is_float(v.type_oid))
			{
				double d = float_of(v);
				if (isnan(d)) kv[k] = 0x7ff8000000000000LL;
				else { if (d == 0.0) d = 0.0; memcpy(&kv[k], &d, 8); }
			}
			else if (v.type_oid == STROM_NUMERICOID)
			{
				uint64_t c = v.v.u;
				(void)num_pack(NUM_EXPO(c), NUM_SIGN(c), NUM_MANT(c), &c);
				kv[k] = (int64_t)c;
			}
			else
				kv[k] = v.v.i;
		}
		for (a = 0; a < naggs; a++)
		{
			preagg_target *tg = &sp.targets[agg_of[a]];
			oracle_value r;
			memset(&r, 0, sizeof(r));
			r.type_oid = tg->type_oid;
			if (tg->kind == T_NROWS)
			{
				int ok = 1, j;
				for (j = 0; j < tg->nexprs; j++)
				{
					oracle_value x = oracle_expr_eval(tg->exprs[j], kds, row, ext_values, ext_isnull, n_ext, &errcode);
					ok &= (!x.isnull && x.v.i != 0);
				}
				r.v.i = ok;
			}
			else if (tg->pcov)
			{
				oracle_value f = oracle_expr_eval(tg->exprs[0], kds, row, ext_values, ext_isnull, n_ext, &errcode);
				oracle_value x = oracle_expr_eval(tg->exprs[1], kds, row, ext_values, ext_isnull, n_ext, &errcode);
				oracle_value y = oracle_expr_eval(tg->exprs[2], kds, row, ext_values, ext_isnull, n_ext, &errcode);
				if (f.isnull || !f.v.i || x.isnull || y.isnull)
					r.isnull = 1;
				else
				{
					double p = (tg->pcov == 1 ? x.v.d : tg->pcov == 2 ? y.v.d :
								tg->pcov == 3 ? x.v.d * x.v.d : tg->pcov == 4 ? y.v.d * y.v.d :
								x.v.d * y.v.d);
					if (tg->pcov >= 3 && float_bad(p, isinf(x.v.d) || isinf(y.v.d),
												   x.v.d == 0.0 || y.v.d == 0.0))
					{
						set_error(&errcode, StromError_CpuReCheck);
						r.isnull = 1;
					}
					r.v.d = p;
				}
			}
			else
			{
				r = oracle_expr_eval(tg->exprs[0], kds, row, ext_values, ext_isnull, n_ext, &errcode);
				if (tg->numeric_scale >= 0)
				{
					/* strom_numeric_to_fixed: exact or CpuReCheck */
					int64_t fx = 0;
					if (!r.isnull && !num_to_fixed(r.v.u, tg->numeric_scale, &fx))
					{
						set_error(&errcode, StromError_CpuReCheck);
						r.isnull = 1;
					}
					r.type_oid = STROM_INT8OID;
					r.v.i = fx;
				}
				if (tg->type_oid == STROM_NUMERICOID && !r.isnull)
				{
					/* kept normalised; a square that leaves the form sends the chunk back */
					uint64_t c = r.v.u;
					if (tg->x2 && !num_mul(c, c, &c))
					{
						set_error(&errcode, StromError_CpuReCheck);
						r.isnull = 1;
					}
					else
						(void)num_pack(NUM_EXPO(c), NUM_SIGN(c), NUM_MANT(c), &c);
					r.type_oid = STROM_NUMERICOID;
					r.v.u = c;
				}
				else if (tg->x2 && !r.isnull)
				{
					double x = r.v.d, p = x * x;
					if (float_bad(p, isinf(x), x == 0.0))
					{
						set_error(&errcode, StromError_CpuReCheck);
						r.isnull = 1;
					}
					r.v.d = p;
				}
			}
			av[a] = r;
		}
		if (errcode != StromError_Success)
		{
			set_error(&status, errcode);
			break;
		}
		/* find or create the group: a hash index over the groups made so far
		 * (what PostgreSQL's hash aggregate does); groups keep creation order */
		{
			uint64_t	h = 0x9e3779b97f4a7c15ULL;
			uint32_t	slot;
			for (k = 0; k < nkeys; k++)
			{
				h ^= (kn[k] ? 0x1234567ULL : (uint64_t)kv[k]) + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
				h *= 0xff51afd7ed558ccdULL;
				h ^= h >> 33;
			}
			if ((ngroups + 1) * 2 > hcap)
			{
				uint32_t	j;
				hcap = hcap ? hcap * 4 : 256;
				free(hidx);
				free(hval);
				hidx = malloc(sizeof(uint32_t) * hcap);
				hval = malloc(sizeof(uint64_t) * hcap);
				memset(hidx, 0xff, sizeof(uint32_t) * hcap);
				for (j = 0; j < ngroups; j++)
				{
					for (slot = (uint32_t)(ghash[j] & (hcap - 1)); hidx[slot] != 0xffffffffu;
						 slot = (slot + 1) & (hcap - 1))
						;
					hidx[slot] = j;
					hval[slot] = ghash[j];
				}
			}
			g = ngroups;
			for (slot = (uint32_t)(h & (hcap - 1)); hidx[slot] != 0xffffffffu; slot = (slot + 1) & (hcap - 1))
			{
				uint32_t	cand = hidx[slot];
				int			same = (hval[slot] == h);
				for (k = 0; k < nkeys && same; k++)
					same = (groups[cand].keynull[k] == kn[k] && (kn[k] || groups[cand].keyval[k] == kv[k]));
				if (same)
				{
					g = cand;
					break;
				}
			}
			if (g == ngroups)
			{
				hidx[slot] = g;
				hval[slot] = h;
				if (ngroups == gcap)
				{
					gcap = gcap ? gcap * 2 : 64;
					ghash = realloc(ghash, sizeof(uint64_t) * gcap);
				}
				ghash[g] = h;
			}
		}
		if (g == ngroups)
		{
			if (ngroups == cap)
			{
				cap = cap ? cap * 2 : 64;
				groups = realloc(groups, sizeof(preagg_group) * cap);
			}
			groups[g].keyval = malloc(sizeof(int64_t) * (nkeys + 1));
			groups[g].keynull = malloc(nkeys + 1);
			groups[g].acc = calloc(naggs + 1, sizeof(oracle_value));
			memcpy(groups[g].keyval, kv, sizeof(int64_t) * nkeys);
			memcpy(groups[g].keynull, kn, nkeys);
			for (a = 0; a < naggs; a++)
			{
				groups[g].acc[a].type_oid = sp.targets[agg_of[a]].type_oid;
				groups[g].acc[a].isnull = (sp.targets[agg_of[a]].kind != T_NROWS);
			}
			ngroups++;
		}
		for (a = 0; a < naggs; a++)
		{
			preagg_target *tg = &sp.targets[agg_of[a]];
			oracle_value *acc = &groups[g].acc[a];
			if (tg->kind == T_NROWS)
			{
				acc->v.i += av[a].v.i;
				continue;
			}
			if (av[a].isnull)
				continue;
			if (tg->type_oid == STROM_NUMERICOID)
			{
				uint64_t x = av[a].v.u;
				if (acc->isnull) acc->v.u = x;
				else if (tg->kind == T_PSUM)
				{
					if (!num_add(acc->v.u, x, &acc->v.u))
						set_error(&status, StromError_CpuReCheck);
				}
				else if (tg->kind == T_PMIN) { if (num_cmp(x, acc->v.u) < 0) acc->v.u = x; }
				else { if (num_cmp(x, acc->v.u) > 0) acc->v.u = x; }
			}
			else if (type_is_float(tg->type_oid))
			{
				double x = float_of(av[a]);
				if (acc->isnull) acc->v.d = x;
				else if (tg->kind == T_PSUM) acc->v.d += x;
				else if (tg->kind == T_PMIN) { if (float_cmp(x, acc->v.d) < 0) acc->v.d = x; }
				else { if (float_cmp(x, acc->v.d) > 0) acc->v.d = x; }
			}
			else
			{
				int64_t x = av[a].v.i;
				if (acc->isnull) acc->v.i = x;
				else if (tg->kind == T_PSUM)
				{
					if (__builtin_add_overflow(acc->v.i, x, &acc->v.i))
						set_error(&status, StromError_CpuReCheck);
				}
				else if (tg->kind == T_PMIN) { if (x < acc->v.i) acc->v.i = x; }
				else { if (x > acc->v.i) acc->v.i = x; }
			}
			acc->isnull = 0;
		}
	}
	if (status == StromError_Success)
	{
		if (ngroups > max_groups)
			status = StromError_DataStoreNoSpace;
		else
		{
			uint32_t g;
			for (g = 0; g < ngroups; g++)
			{
				int k, a;
				for (k = 0; k < nkeys; k++)
				{
					size_t o = (size_t)g * sp.ntargets + key_of[k];
					out_isnull[o] = groups[g].keynull[k];
					out_values[o] = (uint64_t)groups[g].keyval[k];	/* float keys: float8 image, like float partials */
				}
				for (a = 0; a < naggs; a++)
				{
					size_t o = (size_t)g * sp.ntargets + agg_of[a];
					oracle_value *acc = &groups[g].acc[a];
					out_isnull[o] = (uint8_t)acc->isnull;
					if (acc->isnull) out_values[o] = 0;
					else if (type_is_float(acc->type_oid)) memcpy(&out_values[o], &acc->v.d, 8);
					else out_values[o] = (uint64_t)acc->v.i;
				}
			}
			*p_ngroups = ngroups;
		}
	}
	for (i = 0; i < ngroups; i++)
	{
		free(groups[i].keyval);
		free(groups[i].keynull);
		free(groups[i].acc);
	}
	free(groups);
	free(hidx);
	free(hval);
	free(ghash);
	preagg_spec_free(&sp);
	return status;
}

/* ====================================================================== *
 * GpuHashJoin
 *
 * Restates kern_gpuhashjoin_main + the generated gpuhashjoin_execute
 * (opencl_hashjoin.h:284-416; template gpuhashjoin.c:1184-1317): for every
 * outer row (optionally through a kern_row_map), depth by depth, every
 * inner tuple whose hash keys are equal (NULL never matches) and whose
 * join qual is TRUE; one result record {outer_row+1, inner_1, .., inner_n}
 * per complete match.  The inner side is identified here by its ROW INDEX
 * in the inner chunk (the device reports the byte offset of the
 * kern_hashentry, whose rowid field is that index).
 *
 * Deliberately independent of the hash table the product builds: the
 * inner rows are indexed with a private bucket array on the key images.
 * A row-level CpuReCheck in a join expression is a hard error in the
 * reference ("CPU Recheck not implemented yet", gpuhashjoin.c:2948-2952):
 * reported as chunk errcode CpuReCheck.
 * ====================================================================== */
#define HJ_MAXDEPTH	8
#define HJ_MAXKEYS	8

typedef struct {
	int			nkeys;
	oracle_expr *outer_key[HJ_MAXKEYS];
	int			inner_attno[HJ_MAXKEYS];
	int			key_type[HJ_MAXKEYS];
	oracle_expr *qual;
	/* private index over the inner rows */
	uint32_t	nbuckets;
	int32_t	   *bucket;		/* head row or -1 */
	int32_t	   *next;
	uint64_t   *keyhash;
} hj_rel;

static uint64_t
hj_mix(uint64_t h, uint64_t v)
{
	h ^= v + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
	return h;
}

/* canonical 64-bit image of a key value for hashing/equality */
static uint64_t
hj_key_image(oracle_value v)
{
	if (v.type_oid == STROM_FLOAT4OID)
	{
		double d = v.v.f;
		if (d == 0.0) d = 0.0;			/* -0 == +0 */
		if (isnan(d)) return 0x7ff8000000000000ULL;
		{ uint64_t u; memcpy(&u, &d, 8); return u; }
	}
	if (v.type_oid == STROM_FLOAT8OID)
	{
		double d = v.v.d;
		if (d == 0.0) d = 0.0;
		if (isnan(d)) return 0x7ff8000000000000ULL;
		{ uint64_t u; memcpy(&u, &d, 8); return u; }
	}
	/* text / character(n): the value is the address of the datum; equal strings must land in
	 * the same bucket, the comparison itself is hj_keys_equal's (character(n): without the
	 * trailing blanks).  STROMCL_VARLENA_HASHKEY_TEMPLATE, opencl_hashjoin.h:935-953. */
	if (type_is_varlena(v.type_oid))
	{
		int		len, i;
		const uint8_t *p = varlena_payload((const void *)(intptr_t)v.v.i, &len);
		uint64_t h = 1469598103934665603ULL;
		if (v.type_oid == STROM_BPCHARNOID)
			while (len > 0 && p[len - 1] == ' ') len--;
		for (i = 0; i < len; i++)
			h = (h ^ p[i]) * 1099511628211ULL;
		return h;
	}
	return (uint64_t)v.v.i;
}

/* an inner key column: by-value types through load_datum; text / character(n) as the address of
 * the datum (pg_varlena_hashref, opencl_hashjoin.h:860-890: a datum the device cannot read in
 * place is NULL and raises CpuReCheck) */
static oracle_value
hj_inner_key(int type, const void *addr, int32_t *errcode)
{
	oracle_value r;
	if (!type_is_varlena(type) || !addr)
		return load_datum(type, addr);
	memset(&r, 0, sizeof(r));
	r.type_oid = type;
	if (!varlena_readable(addr))
	{
		r.isnull = 1;
		if (errcode)
			set_error(errcode, StromError_CpuReCheck);
		return r;
	}
	r.v.i = (int64_t)(intptr_t)addr;
	return r;
}

/* hash clause "outer key = inner column" by the type's equality operator */
static int
hj_keys_equal(oracle_value a, oracle_value b)
{
	if (type_is_varlena(a.type_oid))
		return varlena_compare((const void *)(intptr_t)a.v.i, (const void *)(intptr_t)b.v.i,
							   a.type_oid == STROM_BPCHARNOID) == 0;
	return hj_key_image(a) == hj_key_image(b);
}

typedef struct {
	int			ndepth;
	hj_rel		rel[HJ_MAXDEPTH];
} hj_spec;

static void
hj_spec_free(hj_spec *sp)
{
	int d, k;
	for (d = 0; d < sp->ndepth; d++)
	{
		for (k = 0; k < sp->rel[d].nkeys; k++)
			oracle_expr_free(sp->rel[d].outer_key[k]);
		oracle_expr_free(sp->rel[d].qual);
		free(sp->rel[d].bucket);
		free(sp->rel[d].next);
		free(sp->rel[d].keyhash);
	}
}

static int
hj_spec_parse(const char *text, hj_spec *sp, char *errbuf, size_t errlen)
{
	parser ps = { text, errbuf, errlen, 0 };
	char	head[32], buf[64];

	memset(sp, 0, sizeof(*sp));
	if (!expect_char(&ps, '(') || !read_atom(&ps, head, sizeof(head)) || strcmp(head, "gpuhashjoin"))
	{
		perr(&ps, "(gpuhashjoin ...) expected");
		return 0;
	}
	while (!peek_close(&ps) && !ps.failed)
	{
		hj_rel *rel;
		if (!expect_char(&ps, '(') || !read_atom(&ps, head, sizeof(head)) || strcmp(head, "rel"))
		{ perr(&ps, "(rel ...) expected"); break; }
		if (sp->ndepth >= HJ_MAXDEPTH) { perr(&ps, "too many inner relations"); break; }
		rel = &sp->rel[sp->ndepth++];
		while (!peek_close(&ps) && !ps.failed)
		{
			if (!expect_char(&ps, '(') || !read_atom(&ps, head, sizeof(head)))
				break;
			if (!strcmp(head, "hashkey"))
			{
				int k = rel->nkeys;
				if (k >= HJ_MAXKEYS) { perr(&ps, "too many hash keys"); break; }
				rel->outer_key[k] = parse_expr(&ps);
				if (!rel->outer_key[k]) break;
				if (!read_atom(&ps, buf, sizeof(buf))) { perr(&ps, "inner attno expected"); break; }
				rel->inner_attno[k] = atoi(buf);
				if (!read_atom(&ps, buf, sizeof(buf)) || !(rel->key_type[k] = type_by_name(buf)))
				{ perr(&ps, "bad hashkey type"); break; }
				rel->nkeys++;
			}
			else if (!strcmp(head, "qual"))
			{
				rel->qual = parse_expr(&ps);
				if (!rel->qual) break;
			}
			else { perr(&ps, "unknown item %s in rel", head); break; }
			if (!expect_char(&ps, ')')) break;
		}
		if (ps.failed || !expect_char(&ps, ')')) break;
		if (rel->nkeys < 1) { perr(&ps, "a rel needs at least one hashkey"); break; }
	}
	if (!ps.failed)
		expect_char(&ps, ')');
	if (ps.failed || sp->ndepth < 1)
	{
		if (!ps.failed) perr(&ps, "no inner relation");
		hj_spec_free(sp);
		return 0;
	}
	return 1;
}

typedef struct {
	hj_spec	   *sp;
	eval_ctx	cx;
	uint32_t	inner_row[HJ_MAXDEPTH];
	int32_t	   *results;
	uint32_t	nrooms;
	uint32_t	nitems;
	int32_t		errcode;
	uint32_t	outer_row;
} hj_state;

static void
hj_probe(hj_state *st, int d)
{
	hj_spec	   *sp = st->sp;
	hj_rel	   *rel = &sp->rel[d];
	const kern_data_store *ikds = st->cx.inner_kds[d];
	oracle_value kv[HJ_MAXKEYS];
	uint64_t	h = 0;
	int			k;
	int32_t		r;

	for (k = 0; k < rel->nkeys; k++)
	{
		kv[k] = eval_node(rel->outer_key[k], &st->cx, &st->errcode);
		if (kv[k].isnull)
			return;					/* NULL = anything is never TRUE */
		h = hj_mix(h, hj_key_image(kv[k]));
	}
	for (r = rel->bucket[h % rel->nbuckets]; r >= 0; r = rel->next[r])
	{
		int same = (rel->keyhash[r] == h);
		for (k = 0; k < rel->nkeys && same; k++)
		{
			oracle_value iv = hj_inner_key(rel->key_type[k],
										   oracle_get_datum(ikds, rel->inner_attno[k] - 1, (uint32_t)r),
										   &st->errcode);
			same = (!iv.isnull && hj_keys_equal(iv, kv[k]));
		}
		if (!same)
			continue;
		st->inner_row[d] = (uint32_t)r;
		if (rel->qual)
		{
			oracle_value q = eval_node(rel->qual, &st->cx, &st->errcode);
			if (q.isnull || !q.v.i)
				continue;
		}
		if (d + 1 < sp->ndepth)
			hj_probe(st, d + 1);
		else
		{
			if (st->nitems < st->nrooms)
			{
				int32_t *rb = st->results + (size_t)st->nitems * (sp->ndepth + 1);
				int		i;
				rb[0] = (int32_t)(st->outer_row + 1);
				for (i = 0; i < sp->ndepth; i++)
					rb[i + 1] = (int32_t)st->inner_row[i];
			}
			st->nitems++;
		}
	}
}

int32_t
oracle_gpuhashjoin(const char *spec_text,
				   const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
				   const kern_data_store *outer, const kern_row_map *krowmap,
				   const kern_data_store *const *inner, int ninner,
				   int32_t *results, uint32_t nrooms, uint32_t *p_nitems,
				   char *errbuf, size_t errlen)
{
	hj_spec		sp;
	hj_state	st;
	int			d, use_map = (krowmap && krowmap->nvalids >= 0);
	uint32_t	i, nrows;

	*p_nitems = 0;
	if (!hj_spec_parse(spec_text, &sp, errbuf, errlen))
		return StromError_BadRequestMessage;
	if (sp.ndepth != ninner)
	{
		snprintf(errbuf, errlen, "spec has %d inner relations, %d given", sp.ndepth, ninner);
		hj_spec_free(&sp);
		return StromError_BadRequestMessage;
	}
	/* private index per inner relation */
	for (d = 0; d < sp.ndepth; d++)
	{
		hj_rel *rel = &sp.rel[d];
		uint32_t n = inner[d]->nitems, r;
		int		k;
		rel->nbuckets = n * 2 + 1;
		rel->bucket = malloc(sizeof(int32_t) * rel->nbuckets);
		rel->next = malloc(sizeof(int32_t) * (n + 1));
		rel->keyhash = malloc(sizeof(uint64_t) * (n + 1));
		for (r = 0; r < rel->nbuckets; r++)
			rel->bucket[r] = -1;
		for (r = 0; r < n; r++)
		{
			uint64_t h = 0;
			int		 isnull = 0;
			for (k = 0; k < rel->nkeys; k++)
			{
				oracle_value iv = hj_inner_key(rel->key_type[k],
											   oracle_get_datum(inner[d], rel->inner_attno[k] - 1, r), NULL);
				if (iv.isnull) { isnull = 1; continue; }
				h = hj_mix(h, hj_key_image(iv));
			}
			rel->keyhash[r] = h;
			rel->next[r] = -1;
			if (isnull)
				continue;			/* a NULL key never joins */
			rel->next[r] = rel->bucket[h % rel->nbuckets];
			rel->bucket[h % rel->nbuckets] = (int32_t)r;
		}
	}
	memset(&st, 0, sizeof(st));
	st.sp = &sp;
	st.cx.kds = outer;
	st.cx.ext_values = ext_values;
	st.cx.ext_isnull = ext_isnull;
	st.cx.n_ext = n_ext;
	st.cx.inner_kds = inner;
	st.cx.inner_row = st.inner_row;
	st.results = results;
	st.nrooms = nrooms;
	nrows = use_map ? (uint32_t)krowmap->nvalids : outer->nitems;
	for (i = 0; i < nrows; i++)
	{
		st.outer_row = use_map ? (uint32_t)krowmap->rindex[i] : i;
		st.cx.rowidx = st.outer_row;
		hj_probe(&st, 0);
	}
	*p_nitems = st.nitems;
	d = st.errcode;
	hj_spec_free(&sp);
	if (StromErrorIsSignificant(d) || d == StromError_CpuReCheck)
		return d;
	return st.nitems > nrooms ? StromError_DataStoreNoSpace : StromError_Success;
}

/*
 * PostgreSQL 9.4's pg_crc32 as the reference uses it for hash keys
 * (gpuhashjoin.c:3758-3788; opencl_hashjoin.h:907-953): the reflected
 * 0xEDB88320 table driven MSB-first.
 */
static uint32_t oracle_crc_table[256];
static void
oracle_crc_init(void)
{
	uint32_t i, j, c;
	if (oracle_crc_table[1])
		return;
	for (i = 0; i < 256; i++)
	{
		c = i;
		for (j = 0; j < 8; j++)
			c = (c & 1) ? (0xEDB88320U ^ (c >> 1)) : (c >> 1);
		oracle_crc_table[i] = c;
	}
}

uint32_t
oracle_pg_crc32(uint32_t crc, const void *data, size_t len)
{
	const unsigned char *p = data;
	oracle_crc_init();
	while (len-- > 0)
		crc = oracle_crc_table[((crc >> 24) ^ *p++) & 0xFF] ^ (crc << 8);
	return crc;
}

/*
 * Walk a kern_multihash the way the reference's device code does
 * (KERN_HASH_FIRST_ENTRY / NEXT_ENTRY, opencl_hashjoin.h:167-192) and check
 * it against the inner chunk it was built from: every row reachable exactly
 * once from slot (hash % nslots), hash == pg_crc32 of its non-NULL key
 * images, tuple bytes identical.  Returns the number of entries, or a
 * negative number naming the first defect.
 */
long
oracle_check_hashtable(const kern_multihash *kmhash, int depth,
					   const kern_data_store *inner,
					   const int *key_attnos, const int *key_lens, int nkeys)
{
	const kern_hashtable *kht;
	const uint32_t *slots;
	uint8_t	   *seen;
	uint32_t	s;
	long		count = 0;

	oracle_crc_init();
	if (depth < 1 || (uint32_t)depth > kmhash->ntables)
		return -1;
	if (memcmp(kmhash->pg_crc32_table, oracle_crc_table, sizeof(oracle_crc_table)) != 0)
		return -2;
	kht = KERN_HASHTABLE(kmhash, depth - 1);
	if (kht->ncols != inner->ncols || kht->nslots == 0)
		return -3;
	slots = KERN_HASHTABLE_SLOT(kht);
	seen = calloc(inner->nitems + 1, 1);
	for (s = 0; s < kht->nslots; s++)
	{
		uint32_t off = slots[s];
		while (off != 0)
		{
			const kern_hashentry *he = (const kern_hashentry *)((const char *)kht + off);
			uint32_t crc = 0xFFFFFFFFU;
			int		k;
			if (off >= kht->length || he->rowid >= inner->nitems || seen[he->rowid])
			{ free(seen); return -4; }
			seen[he->rowid] = 1;
			for (k = 0; k < nkeys; k++)
			{
				const void *p = oracle_get_datum(inner, key_attnos[k] - 1, he->rowid);
				const void *q = get_datum_tuple(kht->colmeta, &he->htup, key_attnos[k] - 1);
				if ((p == NULL) != (q == NULL))
				{ free(seen); return -5; }
				if (p && key_lens[k] < 0)
				{
					/* varlena key: same datum bytes, hashed over the payload
					 * (gpuhashjoin.c:3775-3779) */
					int		lp, lq;
					const uint8_t *pp = varlena_payload(p, &lp);
					const uint8_t *qq = varlena_payload(q, &lq);
					if (lp != lq || memcmp(pp, qq, (size_t)lp) != 0)
					{ free(seen); return -5; }
					crc = oracle_pg_crc32(crc, pp, (size_t)lp);
					continue;
				}
				if (p && memcmp(p, q, key_lens[k]) != 0)
				{ free(seen); return -5; }
				if (p)
				{
					uint64_t datum = 0;
					memcpy(&datum, p, key_lens[k]);
					crc = oracle_pg_crc32(crc, &datum, key_lens[k]);
				}
			}
			crc ^= 0xFFFFFFFFU;
			if (he->hash != crc || crc % kht->nslots != s)
			{ free(seen); return -6; }
			count++;
			off = he->next;
		}
	}
	for (s = 0; s < inner->nitems; s++)
		if (!seen[s])
		{ free(seen); return -7; }
	free(seen);
	return count;
}

/*
 * Evaluate one expression on every row: raw 8-byte value image, isnull and
 * the row's errcode (0 or CpuReCheck).  Lets tests look at scalar results
 * the device only ever consumes inside quals and aggregates.
 */
int32_t
oracle_eval_rows(const char *expr_text,
				 const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
				 const kern_data_store *kds,
				 uint64_t *out_values, uint8_t *out_isnull, int32_t *out_errcode,
				 int32_t *p_type_oid, char *errbuf, size_t errlen)
{
	oracle_expr *expr = oracle_expr_parse(expr_text, errbuf, errlen);
	uint32_t	i;

	if (!expr)
		return StromError_BadRequestMessage;
	*p_type_oid = expr->type_oid;
	for (i = 0; i < kds->nitems; i++)
	{
		int32_t		errcode = StromError_Success;
		oracle_value v = oracle_expr_eval(expr, kds, i, ext_values, ext_isnull, n_ext, &errcode);
		out_isnull[i] = (uint8_t)v.isnull;
		out_errcode[i] = errcode;
		out_values[i] = 0;
		if (!v.isnull)
		{
			if (v.type_oid == STROM_FLOAT4OID) memcpy(&out_values[i], &v.v.f, 4);
			else out_values[i] = v.v.u;
		}
	}
	oracle_expr_free(expr);
	return StromError_Success;
}
