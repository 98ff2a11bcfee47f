/*
 * strom_textlib.h -- text / character(n) values on the device
 *
 * Stands where opencl_textlib.h stands (bpchar* 150-283, text* 285-399) plus
 * the varlena reference templates of opencl_common.h:1126-1263.
 *
 * A value is the ADDRESS of its varlena datum (inside a heap tuple of a ROW /
 * ROW_FLAT chunk, inside an inner tuple of a hash table, or inside the
 * kern_parambuf), exactly what the reference's pg_varlena_t carries.  Only
 * datums the device can read in place are accepted -- a short (1-byte) or an
 * uncompressed 4-byte header (VARATT_IS_1B / VARATT_IS_4B_U,
 * opencl_common.h:1142); a compressed datum or an external TOAST pointer
 * makes the value NULL and raises CpuReCheck, the row goes back to the CPU.
 * (The reference's VARATT_IS_1B test also lets the 1-byte EXTERNAL tag
 * through and would compare the bytes of the TOAST pointer; not reproduced.)
 *
 * Comparison is bytewise on UNSIGNED bytes -- what PostgreSQL's "C" collation
 * does (memcmp), the only collation the reference lets onto the device
 * (devtype_runnable_collation, codegen.c:148-186).  The reference compares
 * signed cl_char (opencl_textlib.h:170-193): same answer for ASCII, the wrong
 * one for bytes >= 0x80; not reproduced.  character(n) ignores trailing
 * blanks (bpchar_truelen, 154-166); text does not.
 *
 * In a COLUMN chunk a text column is an array of 8-byte offsets into the chunk's heap area
 * (strom_kds.h); the kernels load it like an int8 column and pg_<type>_from_column turns offset
 * into address.
 */
#ifndef STROM_TEXTLIB_DEVICE_H
#define STROM_TEXTLIB_DEVICE_H

STROM_DECLARE_SIMPLE_TYPE(text,    cl_ulong)
STROM_DECLARE_SIMPLE_TYPE(bpcharn, cl_ulong)

/* payload of a varlena the device may read: start and length */
STROM_DEVICE const cl_uchar *
strom_varlena_payload(cl_ulong datum, cl_int *p_len)
{
	const cl_uchar *p = (const cl_uchar *)datum;
	cl_uchar	b0 = p[0];

	if (b0 & 0x01)
	{
		*p_len = (cl_int)((b0 >> 1) & 0x7f) - 1;
		return p + 1;
	}
	cl_uint w = (cl_uint)p[0] | ((cl_uint)p[1] << 8) | ((cl_uint)p[2] << 16) | ((cl_uint)p[3] << 24);
	*p_len = (cl_int)((w >> 2) & 0x3fffffff) - 4;
	return p + 4;
}

#define STROM_DECLARE_VARLENA_TYPE(NAME)										\
	STROM_DEVICE pg_##NAME##_t													\
	pg_##NAME##_from_addr(cl_int *errcode, const char *addr, cl_short attlen)	\
	{																			\
		pg_##NAME##_t	result;													\
		cl_uchar		b0 = ((const cl_uchar *)addr)[0];						\
		/* 1-byte external tag, or a 4-byte header with the "compressed" bit */	\
		bool			unreadable = (b0 == 0x01) || ((b0 & 0x03) == 0x02);		\
		if (attlen >= 0)														\
		{																		\
			/* a by-value column declared as text: the chunk lies */			\
			STROM_SET_ERROR(errcode, StromError_DataStoreCorruption);			\
			unreadable = true;													\
		}																		\
		else if (unreadable)													\
			STROM_SET_ERROR(errcode, StromError_CpuReCheck);					\
		result.isnull = unreadable;												\
		result.value = (unreadable ? 0UL : (cl_ulong)addr);						\
		return result;															\
	}																			\
	STROM_DECLARE_VARREF_CORE(NAME, pg_##NAME##_from_addr)						\
	/* a value loaded from a COLUMN chunk's offset array (strom_common.h:		\
	 * strom_kvars_from_column): offset from the kds head -> address */			\
	STROM_DEVICE pg_##NAME##_t													\
	pg_##NAME##_from_column(cl_int *errcode, const kern_data_store *kds,		\
							cl_uint colidx, pg_##NAME##_t v)					\
	{																			\
		if (v.isnull || v.value == 0)											\
			return pg_##NAME##_make(0UL, true);									\
		/* a by-value column declared as text (the chunk lies), or an offset	\
		 * that leaves the chunk */												\
		if (kds->colmeta[colidx].attlen >= 0 || v.value >= kds->length)			\
		{																		\
			STROM_SET_ERROR(errcode, StromError_DataStoreCorruption);			\
			return pg_##NAME##_make(0UL, true);									\
		}																		\
		/* ... or a datum whose own length leaves it (its bytes are about to		\
		 * be read by length) */												\
		if (v.value + strom_varsize_any((const char *)kds + v.value) > kds->length)	\
		{																		\
			STROM_SET_ERROR(errcode, StromError_DataStoreCorruption);			\
			return pg_##NAME##_make(0UL, true);									\
		}																		\
		return pg_##NAME##_from_addr(errcode, (const char *)kds + v.value, -1);	\
	}																			\
	STROM_DEVICE pg_##NAME##_t													\
	pg_##NAME##_param(const kern_parambuf *kparams,								\
					  cl_int *errcode, cl_uint param_id)						\
	{																			\
		if (param_id < kparams->nparams && kparams->poffset[param_id] > 0)		\
			return pg_##NAME##_from_addr(errcode,								\
				(const char *)kparams + kparams->poffset[param_id], -1);		\
		return pg_##NAME##_make(0UL, true);										\
	}

STROM_DECLARE_VARLENA_TYPE(text)
STROM_DECLARE_VARLENA_TYPE(bpcharn)

/*
 * Eight payload bytes in ONE load, whatever their alignment (a datum's payload starts one byte
 * behind a short header): datums live in global memory -- chunks, the kern_parambuf, hash table
 * entries -- where gfx950 serves unaligned accesses; said through a packed struct in address space
 * 1, the compiler issues one global_load_dwordx2 instead of eight byte loads.  Never reads past the
 * bytes asked for: callers take whole words only while 8 bytes are left.
 */
struct __attribute__((packed)) strom_unaligned_u64 { cl_ulong v; };
STROM_DEVICE cl_ulong
strom_load_u64(const cl_uchar *p)
{
	return ((const __attribute__((address_space(1))) strom_unaligned_u64 *)p)->v;
}

/* memcmp order, then the shorter one first (text_compare, 288-312); eight bytes at a time: the
 * first differing word decides, compared most significant byte first */
STROM_DEVICE cl_int
strom_bytes_compare(const cl_uchar *s1, cl_int len1, const cl_uchar *s2, cl_int len2)
{
	cl_int		len = (len1 < len2 ? len1 : len2);
	cl_int		i = 0;

	for (; i + 8 <= len; i += 8)
	{
		cl_ulong	w1 = strom_load_u64(s1 + i), w2 = strom_load_u64(s2 + i);
		if (w1 != w2)
			return (__builtin_bswap64(w1) < __builtin_bswap64(w2) ? -1 : 1);
	}
	if (i < len && len >= 8)
	{
		/* the tail as ONE more word: the last eight bytes of the common prefix, overlapping what
		 * was compared (and found equal) already -- the first differing byte still decides */
		cl_ulong	w1 = strom_load_u64(s1 + len - 8), w2 = strom_load_u64(s2 + len - 8);
		if (w1 != w2)
			return (__builtin_bswap64(w1) < __builtin_bswap64(w2) ? -1 : 1);
		i = len;
	}
	for (; i < len; i++)
	{
		cl_uchar c1 = s1[i], c2 = s2[i];
		if (c1 != c2)
			return (c1 < c2 ? -1 : 1);
	}
	return (len1 == len2 ? 0 : (len1 > len2 ? 1 : -1));
}

/* equality alone: strings of different length are different before a byte is read (texteq) */
STROM_DEVICE bool
strom_bytes_equal(const cl_uchar *s1, cl_int len1, const cl_uchar *s2, cl_int len2)
{
	if (len1 != len2)
		return false;
	cl_int		i = 0;
	for (; i + 8 <= len1; i += 8)
		if (strom_load_u64(s1 + i) != strom_load_u64(s2 + i))
			return false;
	if (i < len1 && len1 >= 8)
		return strom_load_u64(s1 + len1 - 8) == strom_load_u64(s2 + len1 - 8);	/* the tail, overlapping */
	for (; i < len1; i++)
		if (s1[i] != s2[i])
			return false;
	return true;
}

STROM_DEVICE cl_int
strom_text_compare(cl_ulong a, cl_ulong b)
{
	cl_int		len1, len2;
	const cl_uchar *s1 = strom_varlena_payload(a, &len1);
	const cl_uchar *s2 = strom_varlena_payload(b, &len2);
	return strom_bytes_compare(s1, len1, s2, len2);
}

/* character(n): trailing blanks do not count (bpchar_truelen, 154-166) */
STROM_DEVICE cl_int
strom_bpchar_compare(cl_ulong a, cl_ulong b)
{
	cl_int		len1, len2;
	const cl_uchar *s1 = strom_varlena_payload(a, &len1);
	const cl_uchar *s2 = strom_varlena_payload(b, &len2);
	while (len1 > 0 && s1[len1 - 1] == ' ')
		len1--;
	while (len2 > 0 && s2[len2 - 1] == ' ')
		len2--;
	return strom_bytes_compare(s1, len1, s2, len2);
}

#define STROM_DECLARE_TEXT_COMPARE(FNAME, NAME, CMPFN, OP)						\
	STROM_DEVICE pg_bool_t														\
	pgfn_##FNAME(cl_int *errcode, pg_##NAME##_t arg1, pg_##NAME##_t arg2)		\
	{																			\
		pg_bool_t	result;														\
		result.isnull = (arg1.isnull | arg2.isnull);							\
		result.value = false;													\
		if (!result.isnull)														\
			result.value = (CMPFN(arg1.value, arg2.value) OP 0);				\
		return result;															\
	}

/* = / <> through the length-first equality */
STROM_DEVICE cl_int
strom_text_differs(cl_ulong a, cl_ulong b)
{
	cl_int		len1, len2;
	const cl_uchar *s1 = strom_varlena_payload(a, &len1);
	const cl_uchar *s2 = strom_varlena_payload(b, &len2);
	return strom_bytes_equal(s1, len1, s2, len2) ? 0 : 1;
}
STROM_DEVICE cl_int
strom_bpchar_differs(cl_ulong a, cl_ulong b)
{
	cl_int		len1, len2;
	const cl_uchar *s1 = strom_varlena_payload(a, &len1);
	const cl_uchar *s2 = strom_varlena_payload(b, &len2);
	while (len1 > 0 && s1[len1 - 1] == ' ')
		len1--;
	while (len2 > 0 && s2[len2 - 1] == ' ')
		len2--;
	return strom_bytes_equal(s1, len1, s2, len2) ? 0 : 1;
}

STROM_DECLARE_TEXT_COMPARE(bpchareq, bpcharn, strom_bpchar_differs, ==)
STROM_DECLARE_TEXT_COMPARE(bpcharne, bpcharn, strom_bpchar_differs, !=)
STROM_DECLARE_TEXT_COMPARE(bpcharlt, bpcharn, strom_bpchar_compare, <)
STROM_DECLARE_TEXT_COMPARE(bpcharle, bpcharn, strom_bpchar_compare, <=)
STROM_DECLARE_TEXT_COMPARE(bpchargt, bpcharn, strom_bpchar_compare, >)
STROM_DECLARE_TEXT_COMPARE(bpcharge, bpcharn, strom_bpchar_compare, >=)
STROM_DECLARE_TEXT_COMPARE(texteq,   text, strom_text_differs, ==)
STROM_DECLARE_TEXT_COMPARE(textne,   text, strom_text_differs, !=)
STROM_DECLARE_TEXT_COMPARE(text_lt,  text, strom_text_compare, <)
STROM_DECLARE_TEXT_COMPARE(text_le,  text, strom_text_compare, <=)
STROM_DECLARE_TEXT_COMPARE(text_gt,  text, strom_text_compare, >)
STROM_DECLARE_TEXT_COMPARE(text_ge,  text, strom_text_compare, >=)

STROM_DEVICE pg_int4_t
pgfn_bpcharcmp(cl_int *errcode, pg_bpcharn_t arg1, pg_bpcharn_t arg2)
{
	pg_int4_t	result;
	result.isnull = (arg1.isnull | arg2.isnull);
	result.value = (result.isnull ? 0 : strom_bpchar_compare(arg1.value, arg2.value));
	return result;
}

STROM_DEVICE pg_int4_t
pgfn_text_cmp(cl_int *errcode, pg_text_t arg1, pg_text_t arg2)
{
	pg_int4_t	result;
	result.isnull = (arg1.isnull | arg2.isnull);
	result.value = (result.isnull ? 0 : strom_text_compare(arg1.value, arg2.value));
	return result;
}

#endif	/* STROM_TEXTLIB_DEVICE_H */
