/*
 * strom_numeric.h -- 64-bit in-kernel NUMERIC (device)
 *
 * Role in the reference: opencl_numeric.h.  Same value format
 * (122-162): bits 63..58 signed base-10 exponent, bit 57 sign, bits 56..0
 * mantissa; value = (-1)^sign * mantissa * 10^exponent; zero is all-zero.
 * Chunks carry the normalised form (no trailing decimal zero in the mantissa
 * while the exponent can still grow).  INSIDE an expression results are
 * normalised lazily: stripping zeros costs 64-bit divisions, so a result
 * that fits as it is stays as it is, and whenever an operation would fail on
 * such operands it normalises them and tries again -- the outcome (value or
 * CpuReCheck) is exactly that of always-normalised arithmetic, every
 * consumer (compare, casts, fixed-point partial sums) works on the value.
 * Same contract: whatever does not fit -- an
 * exponent outside [-32, 31], a mantissa beyond 57 bits, an intermediate
 * product beyond 64 bits -- yields NULL + StromError_CpuReCheck and
 * PostgreSQL's arbitrary-precision numeric finishes the row
 * (casts 399-779, add/sub/mul 816-1027, compare 1035-1234).
 *
 * Chunks carry either the 8-byte form (COLUMN, TUPSLOT, hash-join images:
 * the reference's "internal_format", datastore.c:355-363) or, inside heap
 * tuples, PostgreSQL's varlena numeric, decoded per row by
 * strom_numeric_from_varlena (166-307).  Not here: float -> numeric casts.
 */
#ifndef STROM_NUMERIC_DEVICE_H
#define STROM_NUMERIC_DEVICE_H

#define PG_NUMERIC_EXPONENT_MAX		31
#define PG_NUMERIC_EXPONENT_MIN		(-32)
#define PG_NUMERIC_SIGN_MASK		(1UL << 57)
#define PG_NUMERIC_MANTISSA_MASK	((1UL << 57) - 1)
#define PG_NUMERIC_EXPONENT(num)	((cl_int)((cl_long)(num) >> 58))
#define PG_NUMERIC_SIGN(num)		(((num) & PG_NUMERIC_SIGN_MASK) != 0)
#define PG_NUMERIC_MANTISSA(num)	((num) & PG_NUMERIC_MANTISSA_MASK)
#define PG_NUMERIC_SET(expo,sign,mant)							\
	((((cl_ulong)(cl_long)(expo)) << 58) |						\
	 ((sign) ? PG_NUMERIC_SIGN_MASK : 0UL) |					\
	 ((mant) & PG_NUMERIC_MANTISSA_MASK))

STROM_DECLARE_SIMPLE_TYPE(numeric, cl_ulong)

STROM_DEVICE pg_numeric_t
strom_numeric_recheck(cl_int *errcode)
{
	pg_numeric_t v;
	v.isnull = true;
	v.value = 0;
	STROM_SET_ERROR(errcode, StromError_CpuReCheck);
	return v;
}

/* 10^n for 0 <= n <= 19, 0 when it does not fit 64 bits */
/* 10^n for 0 <= n <= 19, 0 when it does not fit 64 bits.  Arithmetic, not a
 * table: the power sits on every numeric operation's dependent path and a
 * table read is a global-memory round trip there (measured: each numeric ->
 * fixed conversion cost ~3 ns/row with the table); exponents below 8 -- all
 * of money arithmetic -- need two 32-bit multiplies */
STROM_DEVICE cl_ulong
strom_pow10_u64(int n)
{
	if (n < 0 || n > 19)
		return 0;
	cl_uint		lo = ((n & 1) ? 10u : 1u) * ((n & 2) ? 100u : 1u) * ((n & 4) ? 10000u : 1u);
	cl_ulong	r = lo;
	if (n & 8)
		r *= 100000000UL;
	if (n & 16)
		r *= 10000000000000000UL;
	return r;
}

/* a * b with overflow report; operands below 2^32 (the common case: money
 * amounts, rates) take one 32x32->64 multiply instead of the 128-bit check */
STROM_DEVICE bool
strom_mul_overflow_u64(cl_ulong a, cl_ulong b, cl_ulong *out)
{
	if (((a | b) >> 32) == 0)
	{
		*out = (cl_ulong)(cl_uint)a * (cl_ulong)(cl_uint)b;
		return false;
	}
	return __builtin_mul_overflow(a, b, out);
}

/* move trailing decimal zeros of the mantissa into the exponent */
STROM_DEVICE void
strom_numeric_strip(cl_ulong &mant, int &expo)
{
	while (mant != 0 && mant % 10 == 0 && expo < PG_NUMERIC_EXPONENT_MAX)
	{
		mant /= 10;
		expo++;
	}
}

/* normalise and pack; anything out of range goes back to the CPU */
STROM_DEVICE pg_numeric_t
strom_numeric_pack(cl_int *errcode, int expo, bool sign, cl_ulong mant)
{
	pg_numeric_t v;

	if (mant == 0)
	{
		v.isnull = false;
		v.value = 0;
		return v;
	}
	/* fits as it is: no normalisation (see the header comment) */
	if (!(mant & ~PG_NUMERIC_MANTISSA_MASK) &&
		expo >= PG_NUMERIC_EXPONENT_MIN && expo <= PG_NUMERIC_EXPONENT_MAX)
	{
		v.isnull = false;
		v.value = PG_NUMERIC_SET(expo, sign, mant);
		return v;
	}
	strom_numeric_strip(mant, expo);
	/* an exponent above the field can be traded for mantissa digits */
	while (expo > PG_NUMERIC_EXPONENT_MAX)
	{
		if (mant > PG_NUMERIC_MANTISSA_MASK / 10)
			return strom_numeric_recheck(errcode);
		mant *= 10;
		expo--;
	}
	if (expo < PG_NUMERIC_EXPONENT_MIN || (mant & ~PG_NUMERIC_MANTISSA_MASK))
		return strom_numeric_recheck(errcode);
	v.isnull = false;
	v.value = PG_NUMERIC_SET(expo, sign, mant);
	return v;
}

/*
 * PostgreSQL's on-disk numeric (what a heap page holds; the reference decodes
 * it per row, opencl_numeric.h:166-307): a varlena whose payload is either
 *   short:  uint16 n_header                    ((n_header & 0xC000) == 0x8000)
 *           sign 0x2000, dscale (0x1F80 >> 7), weight sign 0x0040, weight 0x003F
 *   long :  uint16 n_sign_dscale (sign 0xC000: 0 +, 0x4000 -, 0xC000 NaN),
 *           int16 n_weight
 * followed by base-10000 digits, most significant first:
 *   value = sum digit[i] * 10000^(weight - i).
 * Compressed / external varlenas, NaN and values beyond the 64-bit form go
 * back to the CPU.
 */
STROM_DEVICE pg_numeric_t
strom_numeric_from_varlena(cl_int *errcode, const char *addr)
{
	const cl_uchar *p = (const cl_uchar *)addr;
	cl_uint		len;

	if (p[0] == 0x01)
		return strom_numeric_recheck(errcode);			/* external TOAST pointer */
	if (p[0] & 0x01)
	{
		len = (p[0] >> 1) & 0x7f;						/* 1-byte header, total size */
		if (len < 1 + 2)
			return strom_numeric_recheck(errcode);
		len -= 1;
		p += 1;
	}
	else
	{
		cl_uint	w = (cl_uint)p[0] | ((cl_uint)p[1] << 8) | ((cl_uint)p[2] << 16) | ((cl_uint)p[3] << 24);
		if ((w & 0x03) != 0 || (w >> 2) < 4 + 2)
			return strom_numeric_recheck(errcode);		/* compressed, or not a numeric */
		len = (w >> 2) - 4;
		p += 4;
	}
	cl_uint		n_header = (cl_uint)p[0] | ((cl_uint)p[1] << 8);
	bool		sign;
	int			weight;
	cl_uint		ndigits;

	if ((n_header & 0xC000) == 0xC000)
		return strom_numeric_recheck(errcode);			/* NaN */
	if ((n_header & 0xC000) == 0x8000)
	{
		sign = (n_header & 0x2000) != 0;
		weight = (int)(n_header & 0x003F);
		if (n_header & 0x0040)
			weight |= ~0x003F;
		p += 2;
		ndigits = (len - 2) / 2;
	}
	else
	{
		if (len < 4)
			return strom_numeric_recheck(errcode);
		sign = (n_header & 0xC000) == 0x4000;
		weight = (int)(cl_short)((cl_uint)p[2] | ((cl_uint)p[3] << 8));
		p += 4;
		ndigits = (len - 4) / 2;
	}
	cl_ulong	mant = 0;
	for (cl_uint i = 0; i < ndigits; i++)
	{
		cl_uint	d = (cl_uint)p[2 * i] | ((cl_uint)p[2 * i + 1] << 8);
		if (d > 9999 || mant > (0xffffffffffffffffUL - 9999UL) / 10000UL)
			return strom_numeric_recheck(errcode);
		mant = mant * 10000UL + d;
	}
	return strom_numeric_pack(errcode, (weight - (int)ndigits + 1) * 4, sign, mant);
}

/* a located numeric datum: the 8-byte device form (attlen 8, what COLUMN /
 * TUPSLOT chunks and hash-join images carry) or the heap's varlena form */
STROM_DEVICE pg_numeric_t
pg_numeric_from_addr(cl_int *errcode, const char *addr, cl_short attlen)
{
	if (attlen > 0)
	{
		pg_numeric_t v;
		v.isnull = false;
		v.value = strom_fetch<cl_ulong>(addr);
		return v;
	}
	return strom_numeric_from_varlena(errcode, addr);
}
STROM_DECLARE_VARREF_EX(numeric, pg_numeric_from_addr)

/* canonical image (what chunks carry): needed where the datum image itself
 * is consumed -- hashing, stores */
STROM_DEVICE pg_numeric_t
pgfn_numeric_normalize(cl_int *errcode, pg_numeric_t arg)
{
	if (!arg.isnull)
	{
		int			expo = PG_NUMERIC_EXPONENT(arg.value);
		cl_ulong	mant = PG_NUMERIC_MANTISSA(arg.value);

		strom_numeric_strip(mant, expo);
		arg.value = (mant == 0 ? 0UL : PG_NUMERIC_SET(expo, PG_NUMERIC_SIGN(arg.value), mant));
	}
	return arg;
}

STROM_DEVICE pg_numeric_t pgfn_numeric_uplus(cl_int *errcode, pg_numeric_t arg)
{ return arg; }
STROM_DEVICE pg_numeric_t pgfn_numeric_uminus(cl_int *errcode, pg_numeric_t arg)
{
	if (!arg.isnull && PG_NUMERIC_MANTISSA(arg.value) != 0)
		arg.value ^= PG_NUMERIC_SIGN_MASK;
	return arg;
}
STROM_DEVICE pg_numeric_t pgfn_numeric_abs(cl_int *errcode, pg_numeric_t arg)
{
	arg.value &= ~PG_NUMERIC_SIGN_MASK;
	return arg;
}

STROM_DEVICE pg_numeric_t
pgfn_numeric_add(cl_int *errcode, pg_numeric_t arg1, pg_numeric_t arg2)
{
	pg_numeric_t v;

	if (arg1.isnull || arg2.isnull)
	{
		v.isnull = true;
		v.value = 0;
		return v;
	}
	int			expo1 = PG_NUMERIC_EXPONENT(arg1.value), expo2 = PG_NUMERIC_EXPONENT(arg2.value);
	bool		sign1 = PG_NUMERIC_SIGN(arg1.value), sign2 = PG_NUMERIC_SIGN(arg2.value);
	cl_ulong	mant1 = PG_NUMERIC_MANTISSA(arg1.value), mant2 = PG_NUMERIC_MANTISSA(arg2.value);

	if (mant1 == 0)
		return arg2;
	if (mant2 == 0)
		return arg1;
	/* bring both to the smaller exponent; operands may be un-normalised:
	 * on failure normalise them and try once more */
	for (int attempt = 0; ; attempt++)
	{
		cl_ulong	m1 = mant1, m2 = mant2;
		int			e = (expo1 < expo2 ? expo1 : expo2);
		bool		ok = true;

		if (expo1 != expo2)
		{
			bool		first_big = (expo1 > expo2);
			int			diff = (first_big ? expo1 - expo2 : expo2 - expo1);
			cl_ulong	mag = strom_pow10_u64(diff);
			cl_ulong	big = (first_big ? m1 : m2);	/* no pointer to a local: that is scratch */

			if (mag == 0 || strom_mul_overflow_u64(big, mag, &big))
				ok = false;
			m1 = (first_big ? big : m1);
			m2 = (first_big ? m2 : big);
		}
		if (ok)
		{
			if (sign1 != sign2)
			{
				if (m1 < m2)
				{
					sign1 = sign2;
					m1 = m2 - m1;
				}
				else
					m1 -= m2;
			}
			else if (__builtin_add_overflow(m1, m2, &m1))
				ok = false;
		}
		if (ok)
			return strom_numeric_pack(errcode, e, sign1, m1);
		if (attempt != 0)
			return strom_numeric_recheck(errcode);
		strom_numeric_strip(mant1, expo1);
		strom_numeric_strip(mant2, expo2);
	}
}

STROM_DEVICE pg_numeric_t
pgfn_numeric_sub(cl_int *errcode, pg_numeric_t arg1, pg_numeric_t arg2)
{
	return pgfn_numeric_add(errcode, arg1, pgfn_numeric_uminus(errcode, arg2));
}

STROM_DEVICE pg_numeric_t
pgfn_numeric_mul(cl_int *errcode, pg_numeric_t arg1, pg_numeric_t arg2)
{
	pg_numeric_t v;

	if (arg1.isnull || arg2.isnull)
	{
		v.isnull = true;
		v.value = 0;
		return v;
	}
	cl_ulong	mant1 = PG_NUMERIC_MANTISSA(arg1.value), mant2 = PG_NUMERIC_MANTISSA(arg2.value);
	cl_ulong	prod;

	if (mant1 == 0 || mant2 == 0)
	{
		v.isnull = false;
		v.value = 0;
		return v;
	}
	int			expo1 = PG_NUMERIC_EXPONENT(arg1.value), expo2 = PG_NUMERIC_EXPONENT(arg2.value);

	if (strom_mul_overflow_u64(mant1, mant2, &prod))
	{
		/* un-normalised operands?  strip them and try once more */
		strom_numeric_strip(mant1, expo1);
		strom_numeric_strip(mant2, expo2);
		if (strom_mul_overflow_u64(mant1, mant2, &prod))
			return strom_numeric_recheck(errcode);
	}
	return strom_numeric_pack(errcode, expo1 + expo2,
							  PG_NUMERIC_SIGN(arg1.value) != PG_NUMERIC_SIGN(arg2.value), prod);
}

/* three-way compare of two non-NULL values */
STROM_DEVICE int
strom_numeric_cmp(pg_numeric_t arg1, pg_numeric_t arg2)
{
	int			expo1 = PG_NUMERIC_EXPONENT(arg1.value), expo2 = PG_NUMERIC_EXPONENT(arg2.value);
	bool		sign1 = PG_NUMERIC_SIGN(arg1.value), sign2 = PG_NUMERIC_SIGN(arg2.value);
	cl_ulong	mant1 = PG_NUMERIC_MANTISSA(arg1.value), mant2 = PG_NUMERIC_MANTISSA(arg2.value);
	int			ret;

	if (mant1 == 0 && mant2 == 0)
		return 0;
	if (mant1 == 0)
		return sign2 ? 1 : -1;
	if (mant2 == 0)
		return sign1 ? -1 : 1;
	if (sign1 != sign2)
		return sign1 ? -1 : 1;
	/* same sign: compare magnitudes at the smaller exponent */
	if (expo1 == expo2)
		ret = (mant1 < mant2 ? -1 : (mant1 > mant2 ? 1 : 0));
	else
	{
		bool		first_big = (expo1 > expo2);
		int			diff = (first_big ? expo1 - expo2 : expo2 - expo1);
		cl_ulong	mag = strom_pow10_u64(diff);
		cl_ulong	big = (first_big ? mant1 : mant2), small = (first_big ? mant2 : mant1);
		cl_ulong	scaled;
		int			c;		/* compare(big side, small side) */

		if (mag == 0 || strom_mul_overflow_u64(big, mag, &scaled))
			c = 1;			/* does not even fit 64 bits: it is the larger */
		else
			c = (scaled < small ? -1 : (scaled > small ? 1 : 0));
		ret = (first_big ? c : -c);
	}
	return sign1 ? -ret : ret;
}

#define STROM_NUMERIC_COMPARE(name,EXPR)										\
	STROM_DEVICE pg_bool_t														\
	pgfn_numeric_##name(cl_int *errcode, pg_numeric_t arg1, pg_numeric_t arg2)	\
	{																			\
		pg_bool_t r;															\
		r.isnull = arg1.isnull | arg2.isnull;									\
		r.value = false;														\
		if (!r.isnull)															\
		{																		\
			int c = strom_numeric_cmp(arg1, arg2);								\
			r.value = (EXPR);													\
		}																		\
		return r;																\
	}
STROM_NUMERIC_COMPARE(eq, c == 0)
STROM_NUMERIC_COMPARE(ne, c != 0)
STROM_NUMERIC_COMPARE(lt, c <  0)
STROM_NUMERIC_COMPARE(le, c <= 0)
STROM_NUMERIC_COMPARE(gt, c >  0)
STROM_NUMERIC_COMPARE(ge, c >= 0)
STROM_DEVICE pg_int4_t
pgfn_numeric_cmp(cl_int *errcode, pg_numeric_t arg1, pg_numeric_t arg2)
{
	pg_int4_t r;
	r.isnull = arg1.isnull | arg2.isnull;
	r.value = (r.isnull ? 0 : strom_numeric_cmp(arg1, arg2));
	return r;
}

/* ---- casts -------------------------------------------------------------- */
STROM_DEVICE pg_numeric_t
strom_integer_to_numeric(cl_int *errcode, cl_long value, bool isnull)
{
	pg_numeric_t v;
	if (isnull)
	{
		v.isnull = true;
		v.value = 0;
		return v;
	}
	bool		sign = (value < 0);
	cl_ulong	mant = (sign ? (cl_ulong)0 - (cl_ulong)value : (cl_ulong)value);
	return strom_numeric_pack(errcode, 0, sign, mant);
}
STROM_DEVICE pg_numeric_t pgfn_int2_numeric(cl_int *e, pg_int2_t a)
{ return strom_integer_to_numeric(e, a.value, a.isnull); }
STROM_DEVICE pg_numeric_t pgfn_int4_numeric(cl_int *e, pg_int4_t a)
{ return strom_integer_to_numeric(e, a.value, a.isnull); }
STROM_DEVICE pg_numeric_t pgfn_int8_numeric(cl_int *e, pg_int8_t a)
{ return strom_integer_to_numeric(e, a.value, a.isnull); }

/*
 * float4 / float8 -> numeric (float_to_numeric, opencl_numeric.h:625-738; codegen.c:519-520).
 * PostgreSQL prints the value with FLT_DIG / DBL_DIG significant digits ("%.*g", correctly rounded
 * by the C library) and reads that back (float4_numeric / float8_numeric, utils/adt/numeric.c): the
 * result IS the binary value rounded half-to-even to 6 / 15 significant decimal digits.  The
 * reference gets near that with log10 / exp10 in floating point -- the product value x 10^k is
 * rounded before the digit is: a few per cent of all doubles come out one off in the 15th digit.
 * Here the scaling is exact integer arithmetic (value = M x 2^E; M x 2^(E+k) x 5^k in 192 bits,
 * one extra bit for the half, a sticky flag for what the divisions and shifts drop), so the answer
 * is PostgreSQL's, and the CPU oracle's, bit for bit.  NaN, infinities and values the 64-bit form
 * cannot hold are CpuReCheck, as in the reference.
 */
/* 192-bit unsigned integers, six 32-bit limbs, little endian: all the arithmetic float -> numeric needs */
STROM_DEVICE void
strom_w192_mul_small(cl_uint *w, cl_uint m)
{
	cl_ulong	carry = 0;
	for (int i = 0; i < 6; i++)
	{
		cl_ulong t = (cl_ulong)w[i] * m + carry;
		w[i] = (cl_uint)t;
		carry = t >> 32;
	}
}
STROM_DEVICE cl_uint
strom_w192_div_small(cl_uint *w, cl_uint d)
{
	cl_ulong	rem = 0;
	for (int i = 5; i >= 0; i--)
	{
		cl_ulong t = (rem << 32) | w[i];
		w[i] = (cl_uint)(t / d);
		rem = t % d;
	}
	return (cl_uint)rem;
}
STROM_DEVICE void
strom_w192_shl(cl_uint *w, int n)
{
	int		limbs = n / 32, bits = n % 32;
	for (int i = 5; i >= 0; i--)
	{
		cl_ulong lo = (i - limbs >= 0 ? w[i - limbs] : 0), lo2 = (i - limbs - 1 >= 0 ? w[i - limbs - 1] : 0);
		w[i] = (cl_uint)(bits ? ((lo << bits) | (lo2 >> (32 - bits))) : lo);
	}
}
/* >> n; returns whether a 1 bit was shifted out */
STROM_DEVICE int
strom_w192_shr(cl_uint *w, int n)
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
		cl_ulong lo = (i + limbs < 6 ? w[i + limbs] : 0), hi = (i + limbs + 1 < 6 ? w[i + limbs + 1] : 0);
		w[i] = (cl_uint)(bits ? ((lo >> bits) | (hi << (32 - bits))) : lo);
	}
	return sticky;
}

/*
 * round_half_even(M x 2^E x 10^k), exactly: multiplications and left shifts first, then one more
 * bit (the half), then divisions and right shifts with a sticky flag.  0 when it does not fit 64 bits.
 */
STROM_DEVICE int
strom_scaled_mantissa(cl_ulong M, int E, int k, cl_ulong *p_mant)
{
	cl_uint	w[6] = { (cl_uint)M, (cl_uint)(M >> 32), 0, 0, 0, 0 };
	int		a = E + k, b = k, sticky = 0;		/* x 2^a x 5^b */

	if (b > 47 || b < -36 || a > 130 || a < -185)
		return 0;
	for (; b >= 13; b -= 13)
		strom_w192_mul_small(w, 1220703125u);		/* 5^13 */
	for (; b > 0; b--)
		strom_w192_mul_small(w, 5u);
	if (a > 0)
		strom_w192_shl(w, a);
	strom_w192_shl(w, 1);
	for (; b <= -13; b += 13)
		sticky |= (strom_w192_div_small(w, 1220703125u) != 0);
	for (; b < 0; b++)
		sticky |= (strom_w192_div_small(w, 5u) != 0);
	if (a < 0)
		sticky |= strom_w192_shr(w, -a);
	if (w[2] | w[3] | w[4] | w[5])
		return 0;
	cl_ulong	q = ((cl_ulong)w[1] << 32) | w[0];
	cl_ulong	mant = q >> 1;
	if ((q & 1) && (sticky || (mant & 1)))
		mant++;									/* above the half, or the tie to even */
	*p_mant = mant;
	return 1;
}

STROM_DEVICE pg_numeric_t
strom_float_to_numeric(cl_int *errcode, cl_double value, bool isnull, int dig)
{
	pg_numeric_t	v;

	v.isnull = true;
	v.value = 0;
	if (isnull)
		return v;
	if (__builtin_isnan(value) || __builtin_isinf(value))
	{
		STROM_SET_ERROR(errcode, StromError_CpuReCheck);
		return v;
	}
	if (value == 0.0)
	{
		v.isnull = false;
		return v;
	}
	cl_ulong	bits = (cl_ulong)__double_as_longlong(value);
	bool		sign = (bits >> 63) != 0;
	int			be = (int)((bits >> 52) & 0x7ff);
	cl_ulong	M = (bits & 0xfffffffffffffUL) | (be ? (1UL << 52) : 0);
	int			E = (be ? be : 1) - 1075;		/* value = M x 2^E */
	int			e2 = 63 - __builtin_clzl(M) + E;	/* floor(log2(value)) */
	int			e10 = (e2 * 1233) >> 12;			/* floor(e2 x log10(2)), give or take one */
	int			k = dig - 1 - e10;					/* value x 10^k: 'dig' digits in front of the point */
	cl_ulong	lim = 1, mant = 0;
	for (int i = 0; i < dig; i++)
		lim *= 10;
	bool		ok = false;
	for (int turn = 0; turn < 4; turn++)
	{
		if (!strom_scaled_mantissa(M, E, k, &mant))
			break;
		if (mant > lim)
			k--;
		else if (mant < lim / 10)
			k++;
		else
		{
			ok = true;
			break;
		}
	}
	if (!ok)
	{
		STROM_SET_ERROR(errcode, StromError_CpuReCheck);
		return v;
	}
	return strom_numeric_pack(errcode, -k, sign, mant);		/* (mant == 10^dig: normalised there) */
}
STROM_DEVICE pg_numeric_t pgfn_float4_numeric(cl_int *e, pg_float4_t a)
{ return strom_float_to_numeric(e, (cl_double)a.value, a.isnull, 6); }
STROM_DEVICE pg_numeric_t pgfn_float8_numeric(cl_int *e, pg_float8_t a)
{ return strom_float_to_numeric(e, a.value, a.isnull, 15); }

/* numeric -> integer: round half away from zero, then range check */
STROM_DEVICE bool
strom_numeric_to_int64(pg_numeric_t arg, cl_long lo, cl_long hi, cl_long *p_value)
{
	int			expo = PG_NUMERIC_EXPONENT(arg.value);
	bool		sign = PG_NUMERIC_SIGN(arg.value);
	cl_ulong	mant = PG_NUMERIC_MANTISSA(arg.value);

	if (expo < 0)
	{
		cl_ulong mag = strom_pow10_u64(-expo);
		if (mag == 0)
			mant = 0;			/* |value| < 1e-19 ... rounds to zero */
		else
			mant = (mant + mag / 2) / mag;
	}
	else if (expo > 0)
	{
		cl_ulong mag = strom_pow10_u64(expo);
		if (mag == 0 || strom_mul_overflow_u64(mant, mag, &mant))
			return false;
	}
	if (!sign)
	{
		if (mant > (cl_ulong)hi)
			return false;
		*p_value = (cl_long)mant;
	}
	else
	{
		if (mant > (cl_ulong)0 - (cl_ulong)lo)
			return false;
		*p_value = (cl_long)((cl_ulong)0 - mant);
	}
	return true;
}
#define STROM_NUMERIC_TO_INT(name,r_type,LO,HI)									\
	STROM_DEVICE pg_##r_type##_t												\
	pgfn_numeric_##name(cl_int *errcode, pg_numeric_t arg)						\
	{																			\
		pg_##r_type##_t r;														\
		cl_long		v = 0;														\
		r.isnull = arg.isnull;													\
		r.value = 0;															\
		if (!r.isnull)															\
		{																		\
			if (!strom_numeric_to_int64(arg, (LO), (HI), &v))					\
			{																	\
				r.isnull = true;												\
				STROM_SET_ERROR(errcode, StromError_CpuReCheck);				\
			}																	\
			else																\
				r.value = (pg_##r_type##_base_t)v;								\
		}																		\
		return r;																\
	}
STROM_NUMERIC_TO_INT(int2, int2, -32768L, 32767L)
STROM_NUMERIC_TO_INT(int4, int4, -2147483648L, 2147483647L)
STROM_NUMERIC_TO_INT(int8, int8, (-9223372036854775807L - 1), 9223372036854775807L)

STROM_DEVICE pg_float8_t
pgfn_numeric_float8(cl_int *errcode, pg_numeric_t arg)
{
	pg_float8_t r;
	r.isnull = arg.isnull;
	r.value = 0.0;
	if (!r.isnull)
	{
		int			expo = PG_NUMERIC_EXPONENT(arg.value);
		cl_ulong	mant = PG_NUMERIC_MANTISSA(arg.value);
		double		m = (double)mant, p = 1.0;
		/* 10^|expo| is exact in double up to 1e22; one correctly rounded
		 * multiply or divide follows */
		for (int i = 0; i < (expo < 0 ? -expo : expo); i++)
			p *= 10.0;
		r.value = (expo < 0 ? m / p : m * p);
		if (PG_NUMERIC_SIGN(arg.value))
			r.value = -r.value;
	}
	return r;
}
STROM_DEVICE pg_float4_t
pgfn_numeric_float4(cl_int *errcode, pg_numeric_t arg)
{
	pg_float8_t d = pgfn_numeric_float8(errcode, arg);
	pg_float4_t r;
	r.isnull = d.isnull;
	r.value = (cl_float)d.value;
	if (!r.isnull && STROM_CHECKFLOATVAL(r.value, false, d.value == 0.0))
	{
		r.isnull = true;
		STROM_SET_ERROR(errcode, StromError_CpuReCheck);
	}
	return r;
}

/*
 * numeric -> fixed-point int8 at 10^-scale, exact or CpuReCheck.  GpuPreAgg
 * accumulates numeric partials in this form (integer LDS atomics); the
 * scale comes from the target (column typmod / expression scale).
 */
STROM_DEVICE pg_int8_t
strom_numeric_to_fixed(cl_int *errcode, pg_numeric_t arg, int scale)
{
	pg_int8_t r;

	r.isnull = arg.isnull;
	r.value = 0;
	if (!r.isnull)
	{
		int			expo = PG_NUMERIC_EXPONENT(arg.value);
		cl_ulong	mant = PG_NUMERIC_MANTISSA(arg.value);

		if (mant == 0)
			return r;
		if (expo + scale < 0)
			strom_numeric_strip(mant, expo);	/* un-normalised: zeros may cover it */
		int			shift = expo + scale;
		cl_ulong	mag = strom_pow10_u64(shift);

		if (shift < 0 || mag == 0 || strom_mul_overflow_u64(mant, mag, &mant) ||
			mant > 9223372036854775807UL)
		{
			/* finer than the accumulator's scale, or too large */
			r.isnull = true;
			STROM_SET_ERROR(errcode, StromError_CpuReCheck);
			return r;
		}
		r.value = (PG_NUMERIC_SIGN(arg.value) ? -(cl_long)mant : (cl_long)mant);
	}
	return r;
}

/* ====================================================================== *
 * fixed-scale numerics: pg_fixed_t
 *
 * When the plan knows a numeric column's typmod scale -- (var N numeric
 * SCALE) -- the emitter keeps the value as an int64 at 10^-SCALE for the
 * whole expression: the column datum is converted once (exactly, or the
 * row is re-checked), add / sub / mul / compare are single checked integer
 * operations with scales resolved at code-generation time, and a partial
 * sum needs no conversion at all.  The 64-bit float-decimal form stays the
 * chunk format and the fallback for anything mixed with scale-less values.
 * Results are exact; what overflows int64 goes back to the CPU like any
 * other numeric overflow.
 * ====================================================================== */
struct pg_fixed_t {
	cl_long		value;
	cl_bool		isnull;
};

STROM_DEVICE pg_fixed_t
pg_fixed_make(cl_long value, bool isnull)
{
	pg_fixed_t r;
	r.value = value;
	r.isnull = isnull;
	return r;
}

/* a literal: 'asnumeric' is the same constant as a kern_parambuf entry, used
 * by the emitter when the literal meets a scale-less operand */
STROM_DEVICE pg_fixed_t
pg_fixed_lit(cl_long value, pg_numeric_t asnumeric)
{
	return pg_fixed_make(value, false);
}

/* a numeric(p,s) column stored as int8 at 10^-s (STROM_DECIMALOID, (var N decimal S)): the
 * datum is the fixed-point value */
STROM_DECLARE_SIMPLE_TYPE(decimal, cl_long)
STROM_DECLARE_VARREF(decimal)

STROM_DEVICE pg_fixed_t
pg_fixed_from_decimal(pg_decimal_t arg)
{
	return pg_fixed_make(arg.value, arg.isnull);
}

STROM_DEVICE pg_fixed_t
pg_fixed_recheck(cl_int *errcode)
{
	STROM_SET_ERROR(errcode, StromError_CpuReCheck);
	return pg_fixed_make(0, true);
}

STROM_DEVICE pg_fixed_t
pgfn_numeric_as_fixed(cl_int *errcode, pg_numeric_t arg, int scale)
{
	/*
	 * column datum -> int64 at 10^-scale.  The usual datum (mantissa below
	 * 2^32, 0 <= exponent + scale <= 9) takes straight-line code: 10^shift
	 * from three selects and two 32-bit multiplies, one 32x32->64 multiply.
	 * Everything else goes through the general conversion.
	 */
	int			expo = PG_NUMERIC_EXPONENT(arg.value);
	cl_ulong	mant = PG_NUMERIC_MANTISSA(arg.value);
	cl_uint		shift = (cl_uint)(expo + scale);
	bool		fast = ((mant >> 32) == 0) & (shift <= 9u);

	/* (wave-uniform: every lane's datum is the usual kind, or the whole wave converts the long way) */
	if (__builtin_amdgcn_ballot_w64(!fast) == 0)
	{
		cl_uint		p = ((shift & 1) ? 10u : 1u) * ((shift & 2) ? 100u : 1u) * ((shift & 4) ? 10000u : 1u);
		cl_ulong	v = (cl_ulong)(cl_uint)mant * (cl_ulong)p;

		v = ((shift & 8) ? v * 100000000UL : v);		/* < 2^32 * 10^9 < 2^62 */
		return pg_fixed_make(PG_NUMERIC_SIGN(arg.value) ? -(cl_long)v : (cl_long)v, arg.isnull);
	}
	pg_int8_t r = strom_numeric_to_fixed(errcode, arg, scale);
	return pg_fixed_make(r.value, r.isnull);
}

/*
 * the same conversion made once per row, where the row's variables are assembled (the GpuPreAgg
 * kernels: STROM_KVARS_FINISH), for every expression that reads the column: value, NULL flag and
 * "this image does not convert" -- raised as CpuReCheck by the expression that USES the value, so a
 * row the qual drops raises nothing, exactly as when every use converted for itself
 */
struct pg_fixed_cache_t {
	cl_long		value;
	cl_bool		isnull;
	cl_bool		recheck;
};

STROM_DEVICE pg_fixed_cache_t
pg_fixed_cache_fill(pg_numeric_t arg, int scale)
{
	cl_int		e = StromError_Success;
	pg_fixed_t	f = pgfn_numeric_as_fixed(&e, arg, scale);
	pg_fixed_cache_t c;
	c.value = f.value;
	c.isnull = f.isnull;
	c.recheck = (e != StromError_Success);
	return c;
}

STROM_DEVICE pg_fixed_t
pg_fixed_cached(cl_int *errcode, pg_fixed_cache_t c)
{
	STROM_SET_RECHECK_IF(errcode, c.recheck);
	return pg_fixed_make(c.value, c.isnull);
}

STROM_DEVICE pg_numeric_t
pgfn_fixed_to_numeric(cl_int *errcode, pg_fixed_t arg, int scale)
{
	if (arg.isnull)
	{
		pg_numeric_t v;
		v.isnull = true;
		v.value = 0;
		return v;
	}
	bool		sign = (arg.value < 0);
	cl_ulong	mant = (sign ? (cl_ulong)0 - (cl_ulong)arg.value : (cl_ulong)arg.value);
	return strom_numeric_pack(errcode, -scale, sign, mant);
}

STROM_DEVICE pg_int8_t
pgfn_fixed_to_int8(cl_int *errcode, pg_fixed_t arg)
{
	pg_int8_t r;
	r.value = arg.value;
	r.isnull = arg.isnull;
	return r;
}

/*
 * The checked operations below are written without branches: NULL-ness and
 * overflow are selects, the error code is raised by STROM_SET_RECHECK_IF.
 */
/* value * factor, factor = 10^k chosen by the emitter */
STROM_DEVICE pg_fixed_t
pgfn_fixed_scaleup(cl_int *errcode, pg_fixed_t arg, cl_long factor)
{
	cl_long		v;
	bool		ovf = __builtin_mul_overflow(arg.value, factor, &v) & !arg.isnull;

	STROM_SET_RECHECK_IF(errcode, ovf);
	return pg_fixed_make(ovf ? 0 : v, arg.isnull | ovf);
}

STROM_DEVICE pg_fixed_t
pgfn_fixed_add(cl_int *errcode, pg_fixed_t a, pg_fixed_t b)
{
	cl_long		v;
	bool		isnull = a.isnull | b.isnull;
	bool		ovf = __builtin_add_overflow(a.value, b.value, &v) & !isnull;

	STROM_SET_RECHECK_IF(errcode, ovf);
	return pg_fixed_make((isnull | ovf) ? 0 : v, isnull | ovf);
}

STROM_DEVICE pg_fixed_t
pgfn_fixed_sub(cl_int *errcode, pg_fixed_t a, pg_fixed_t b)
{
	cl_long		v;
	bool		isnull = a.isnull | b.isnull;
	bool		ovf = __builtin_sub_overflow(a.value, b.value, &v) & !isnull;

	STROM_SET_RECHECK_IF(errcode, ovf);
	return pg_fixed_make((isnull | ovf) ? 0 : v, isnull | ovf);
}

STROM_DEVICE pg_fixed_t
pgfn_fixed_mul(cl_int *errcode, pg_fixed_t a, pg_fixed_t b)
{
	bool		isnull = a.isnull | b.isnull;

	/* both within int32 (amounts x rates) in EVERY lane: one 32x32->64 multiply, and the wave
	 * takes ONE scalar branch (a per-lane branch costs exec-mask bookkeeping on both sides and a
	 * taken skip in the common case) */
	bool		narrow = (a.value == (cl_long)(cl_int)a.value) & (b.value == (cl_long)(cl_int)b.value);
	if (__builtin_amdgcn_ballot_w64(!narrow) == 0)
		return pg_fixed_make(isnull ? 0 : (cl_long)(cl_int)a.value * (cl_long)(cl_int)b.value, isnull);
	cl_long		v;
	bool		ovf = __builtin_mul_overflow(a.value, b.value, &v) & !isnull;

	STROM_SET_RECHECK_IF(errcode, ovf);
	return pg_fixed_make((isnull | ovf) ? 0 : v, isnull | ovf);
}

STROM_DEVICE pg_fixed_t
pgfn_fixed_uminus(cl_int *errcode, pg_fixed_t a)
{
	if (!a.isnull)
	{
		if (a.value == (-0x7fffffffffffffffL - 1))
			return pg_fixed_recheck(errcode);
		a.value = -a.value;
	}
	return a;
}

STROM_DEVICE pg_fixed_t
pgfn_fixed_abs(cl_int *errcode, pg_fixed_t a)
{
	return (!a.isnull && a.value < 0) ? pgfn_fixed_uminus(errcode, a) : a;
}

STROM_DEVICE pg_bool_t
pgfn_fixed_isnull(cl_int *errcode, pg_fixed_t a)
{
	pg_bool_t r;
	r.isnull = false;
	r.value = a.isnull;
	return r;
}

STROM_DEVICE pg_bool_t
pgfn_fixed_isnotnull(cl_int *errcode, pg_fixed_t a)
{
	pg_bool_t r;
	r.isnull = false;
	r.value = !a.isnull;
	return r;
}

#define STROM_FIXED_COMPARE(name,OP)											\
	STROM_DEVICE pg_bool_t														\
	pgfn_fixed_##name(cl_int *errcode, pg_fixed_t a, pg_fixed_t b)				\
	{																			\
		pg_bool_t r;															\
		r.isnull = a.isnull | b.isnull;											\
		r.value = (!r.isnull && (a.value OP b.value));							\
		return r;																\
	}
STROM_FIXED_COMPARE(eq, ==)
STROM_FIXED_COMPARE(ne, !=)
STROM_FIXED_COMPARE(lt, <)
STROM_FIXED_COMPARE(le, <=)
STROM_FIXED_COMPARE(gt, >)
STROM_FIXED_COMPARE(ge, >=)

#endif	/* STROM_NUMERIC_DEVICE_H */
