/*
 * strom_mathlib.h -- checked scalar arithmetic / comparison / casts (device)
 *
 * Role in the reference: opencl_mathlib.h (int/float + - * / % with
 * PostgreSQL's overflow rules, 34-812; dpow/dpi 819-845) plus the operator
 * and cast functions codegen.c synthesises from its catalog templates
 * (codegen.c:632-814).  Contract kept: arguments NULL => result NULL
 * (strict); overflow, division by zero or a non-finite float result
 * => result NULL and StromError_CpuReCheck on the row, so the host
 * re-evaluates that row (opencl_mathlib.h:34-53).
 *
 * Deliberate differences, both towards PostgreSQL (whose output is the
 * reference's own test oracle, input/make_expected.sh:22-28):
 *   - float comparisons order NaN above every number and equal to itself
 *     (float8_cmp_internal); the reference emits a bare IEEE '<'.
 *   - narrowing casts range-check (out of range => CpuReCheck) and
 *     float->int rounds to nearest-even like dtoi4/dtoi8; the reference
 *     emits a bare C cast (codegen.c:641-660).
 * Result types follow PostgreSQL's signatures, not the mis-instantiated
 * templates listed in SURVEY.md Appendix C.
 */
#ifndef STROM_MATHLIB_DEVICE_H
#define STROM_MATHLIB_DEVICE_H

#define STROM_SAMESIGN(a,b)		(((a) < 0) == ((b) < 0))

#define STROM_STRICT2(R)										\
	R result;													\
	result.isnull = arg1.isnull | arg2.isnull;					\
	result.value = 0;

#define STROM_RECHECK()											\
	do { result.isnull = true;									\
		 STROM_SET_ERROR(errcode, StromError_CpuReCheck); } while (0)

/* ---- integer add / sub / mul / div / mod ----------------------------- */
#define STROM_INT_ADDFUNC(name,r_type,x_type,y_type)						\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		/* branch-free: overflow and NULL-ness are selects */				\
		pg_##r_type##_base_t a = arg1.value, b = arg2.value, c;				\
		bool	ovf = __builtin_add_overflow(a, b, &c) & !result.isnull;		\
		STROM_SET_RECHECK_IF(errcode, ovf);									\
		result.value = ((result.isnull | ovf) ? (pg_##r_type##_base_t)0 : c);	\
		result.isnull |= ovf;												\
		return result;														\
	}
#define STROM_INT_SUBFUNC(name,r_type,x_type,y_type)						\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		/* branch-free: overflow and NULL-ness are selects */				\
		pg_##r_type##_base_t a = arg1.value, b = arg2.value, c;				\
		bool	ovf = __builtin_sub_overflow(a, b, &c) & !result.isnull;		\
		STROM_SET_RECHECK_IF(errcode, ovf);									\
		result.value = ((result.isnull | ovf) ? (pg_##r_type##_base_t)0 : c);	\
		result.isnull |= ovf;												\
		return result;														\
	}
#define STROM_INT_MULFUNC(name,r_type,x_type,y_type)						\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		/* branch-free: overflow and NULL-ness are selects */				\
		pg_##r_type##_base_t a = arg1.value, b = arg2.value, c;				\
		bool	ovf = __builtin_mul_overflow(a, b, &c) & !result.isnull;		\
		STROM_SET_RECHECK_IF(errcode, ovf);									\
		result.value = ((result.isnull | ovf) ? (pg_##r_type##_base_t)0 : c);	\
		result.isnull |= ovf;												\
		return result;														\
	}
/* x / 0 and MIN / -1 both go back to the CPU */
#define STROM_INT_DIVFUNC(name,r_type,x_type,y_type)						\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		if (!result.isnull)													\
		{																	\
			pg_##r_type##_base_t a = arg1.value, b = arg2.value, c;			\
			if (b == 0)														\
				STROM_RECHECK();											\
			else if (b == -1)												\
			{																\
				if (__builtin_sub_overflow((pg_##r_type##_base_t)0, a, &c))	\
					STROM_RECHECK();										\
				else														\
					result.value = c;										\
			}																\
			else															\
				result.value = a / b;										\
		}																	\
		return result;														\
	}
#define STROM_INT_MODFUNC(name,d_type)										\
	STROM_DEVICE pg_##d_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##d_type##_t arg1, pg_##d_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##d_type##_t)										\
		if (!result.isnull)													\
		{																	\
			if (arg2.value == 0)											\
				STROM_RECHECK();											\
			else if (arg2.value == -1)										\
				result.value = 0;											\
			else															\
				result.value = arg1.value % arg2.value;						\
		}																	\
		return result;														\
	}

#define STROM_INT_ARITH_FAMILY(OPFUNC, sfx)			\
	OPFUNC(int2##sfx,  int2, int2, int2)			\
	OPFUNC(int24##sfx, int4, int2, int4)			\
	OPFUNC(int28##sfx, int8, int2, int8)			\
	OPFUNC(int42##sfx, int4, int4, int2)			\
	OPFUNC(int4##sfx,  int4, int4, int4)			\
	OPFUNC(int48##sfx, int8, int4, int8)			\
	OPFUNC(int82##sfx, int8, int8, int2)			\
	OPFUNC(int84##sfx, int8, int8, int4)			\
	OPFUNC(int8##sfx,  int8, int8, int8)

STROM_INT_ARITH_FAMILY(STROM_INT_ADDFUNC, pl)
STROM_INT_ARITH_FAMILY(STROM_INT_SUBFUNC, mi)
STROM_INT_ARITH_FAMILY(STROM_INT_MULFUNC, mul)
STROM_INT_ARITH_FAMILY(STROM_INT_DIVFUNC, div)
STROM_INT_MODFUNC(int2mod, int2)
STROM_INT_MODFUNC(int4mod, int4)
STROM_INT_MODFUNC(int8mod, int8)

/* ---- float add / sub / mul / div -------------------------------------- *
 * CHECKFLOATVAL: a result that is inf although no input was, or zero
 * although (for mul/div) no input allowed it, is an overflow/underflow.
 */
#define STROM_CHECKFLOATVAL(val, inf_is_valid, zero_is_valid)	\
	((__builtin_isinf(val) && !(inf_is_valid)) ||				\
	 ((val) == 0.0 && !(zero_is_valid)))

#define STROM_FLOAT_ADDSUB(name,r_type,x_type,y_type,OP)					\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		if (!result.isnull)													\
		{																	\
			pg_##r_type##_base_t a = arg1.value, b = arg2.value;			\
			result.value = a OP b;											\
			if (STROM_CHECKFLOATVAL(result.value,							\
									__builtin_isinf(a) ||					\
									__builtin_isinf(b), true))				\
				STROM_RECHECK();											\
		}																	\
		return result;														\
	}
#define STROM_FLOAT_MULFUNC(name,r_type,x_type,y_type)						\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		if (!result.isnull)													\
		{																	\
			pg_##r_type##_base_t a = arg1.value, b = arg2.value;			\
			result.value = a * b;											\
			if (STROM_CHECKFLOATVAL(result.value,							\
									__builtin_isinf(a) ||					\
									__builtin_isinf(b),						\
									a == 0.0 || b == 0.0))					\
				STROM_RECHECK();											\
		}																	\
		return result;														\
	}
#define STROM_FLOAT_DIVFUNC(name,r_type,x_type,y_type)						\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		if (!result.isnull)													\
		{																	\
			pg_##r_type##_base_t a = arg1.value, b = arg2.value;			\
			if (b == 0.0)													\
				STROM_RECHECK();											\
			else															\
			{																\
				result.value = a / b;										\
				if (STROM_CHECKFLOATVAL(result.value,						\
										__builtin_isinf(a) ||				\
										__builtin_isinf(b), a == 0.0))		\
					STROM_RECHECK();										\
			}																\
		}																	\
		return result;														\
	}

STROM_FLOAT_ADDSUB(float4pl,  float4, float4, float4, +)
STROM_FLOAT_ADDSUB(float48pl, float8, float4, float8, +)
STROM_FLOAT_ADDSUB(float84pl, float8, float8, float4, +)
STROM_FLOAT_ADDSUB(float8pl,  float8, float8, float8, +)
STROM_FLOAT_ADDSUB(float4mi,  float4, float4, float4, -)
STROM_FLOAT_ADDSUB(float48mi, float8, float4, float8, -)
STROM_FLOAT_ADDSUB(float84mi, float8, float8, float4, -)
STROM_FLOAT_ADDSUB(float8mi,  float8, float8, float8, -)
STROM_FLOAT_MULFUNC(float4mul,  float4, float4, float4)
STROM_FLOAT_MULFUNC(float48mul, float8, float4, float8)
STROM_FLOAT_MULFUNC(float84mul, float8, float8, float4)
STROM_FLOAT_MULFUNC(float8mul,  float8, float8, float8)
STROM_FLOAT_DIVFUNC(float4div,  float4, float4, float4)
STROM_FLOAT_DIVFUNC(float48div, float8, float4, float8)
STROM_FLOAT_DIVFUNC(float84div, float8, float8, float4)
STROM_FLOAT_DIVFUNC(float8div,  float8, float8, float8)

/* ---- unary ------------------------------------------------------------ */
#define STROM_INT_UNARY_MINUS(name,d_type)									\
	STROM_DEVICE pg_##d_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##d_type##_t arg1)						\
	{																		\
		pg_##d_type##_t result = arg1;										\
		if (!result.isnull)													\
		{																	\
			pg_##d_type##_base_t c;											\
			if (__builtin_sub_overflow((pg_##d_type##_base_t)0, arg1.value, &c))	\
				STROM_RECHECK();											\
			else															\
				result.value = c;											\
		}																	\
		return result;														\
	}
#define STROM_INT_ABS(name,d_type)											\
	STROM_DEVICE pg_##d_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##d_type##_t arg1)						\
	{																		\
		pg_##d_type##_t result = arg1;										\
		if (!result.isnull && arg1.value < 0)								\
		{																	\
			pg_##d_type##_base_t c;											\
			if (__builtin_sub_overflow((pg_##d_type##_base_t)0, arg1.value, &c))	\
				STROM_RECHECK();											\
			else															\
				result.value = c;											\
		}																	\
		return result;														\
	}
STROM_INT_UNARY_MINUS(int2um, int2)
STROM_INT_UNARY_MINUS(int4um, int4)
STROM_INT_UNARY_MINUS(int8um, int8)
STROM_INT_ABS(int2abs, int2)
STROM_INT_ABS(int4abs, int4)
STROM_INT_ABS(int8abs, int8)
#define STROM_SIMPLE_UNARY(name,d_type,EXPR)								\
	STROM_DEVICE pg_##d_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##d_type##_t arg1)						\
	{																		\
		pg_##d_type##_t result = arg1;										\
		if (!result.isnull)													\
		{																	\
			pg_##d_type##_base_t x = arg1.value;							\
			result.value = (EXPR);											\
		}																	\
		return result;														\
	}
STROM_SIMPLE_UNARY(int2up, int2, x)
STROM_SIMPLE_UNARY(int4up, int4, x)
STROM_SIMPLE_UNARY(int8up, int8, x)
STROM_SIMPLE_UNARY(float4up, float4, x)
STROM_SIMPLE_UNARY(float8up, float8, x)
STROM_SIMPLE_UNARY(float4um, float4, -x)
STROM_SIMPLE_UNARY(float8um, float8, -x)
STROM_SIMPLE_UNARY(float4abs, float4, __builtin_fabsf(x))
STROM_SIMPLE_UNARY(float8abs, float8, __builtin_fabs(x))
STROM_SIMPLE_UNARY(int2not, int2, (cl_short)~x)
STROM_SIMPLE_UNARY(int4not, int4, ~x)
STROM_SIMPLE_UNARY(int8not, int8, ~x)

/* ---- bitwise / shift --------------------------------------------------- */
#define STROM_SIMPLE_BINARY(name,r_type,x_type,y_type,EXPR)					\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)	\
	{																		\
		STROM_STRICT2(pg_##r_type##_t)										\
		if (!result.isnull)													\
		{																	\
			pg_##x_type##_base_t x = arg1.value;							\
			pg_##y_type##_base_t y = arg2.value;							\
			result.value = (pg_##r_type##_base_t)(EXPR);					\
		}																	\
		return result;														\
	}
STROM_SIMPLE_BINARY(int2and, int2, int2, int2, x & y)
STROM_SIMPLE_BINARY(int4and, int4, int4, int4, x & y)
STROM_SIMPLE_BINARY(int8and, int8, int8, int8, x & y)
STROM_SIMPLE_BINARY(int2or,  int2, int2, int2, x | y)
STROM_SIMPLE_BINARY(int4or,  int4, int4, int4, x | y)
STROM_SIMPLE_BINARY(int8or,  int8, int8, int8, x | y)
STROM_SIMPLE_BINARY(int2xor, int2, int2, int2, x ^ y)
STROM_SIMPLE_BINARY(int4xor, int4, int4, int4, x ^ y)
STROM_SIMPLE_BINARY(int8xor, int8, int8, int8, x ^ y)
STROM_SIMPLE_BINARY(int2shl, int2, int2, int4, x << (y & 31))
STROM_SIMPLE_BINARY(int4shl, int4, int4, int4, (cl_uint)x << (y & 31))
STROM_SIMPLE_BINARY(int8shl, int8, int8, int4, (cl_ulong)x << (y & 63))
STROM_SIMPLE_BINARY(int2shr, int2, int2, int4, x >> (y & 31))
STROM_SIMPLE_BINARY(int4shr, int4, int4, int4, x >> (y & 31))
STROM_SIMPLE_BINARY(int8shr, int8, int8, int4, x >> (y & 63))

/* ---- comparison -------------------------------------------------------- */
#define STROM_INT_COMPARE_FAMILY(pfx,x_type,y_type)									\
	STROM_SIMPLE_BINARY(pfx##eq, bool, x_type, y_type, (cl_long)x == (cl_long)y)	\
	STROM_SIMPLE_BINARY(pfx##ne, bool, x_type, y_type, (cl_long)x != (cl_long)y)	\
	STROM_SIMPLE_BINARY(pfx##lt, bool, x_type, y_type, (cl_long)x <  (cl_long)y)	\
	STROM_SIMPLE_BINARY(pfx##le, bool, x_type, y_type, (cl_long)x <= (cl_long)y)	\
	STROM_SIMPLE_BINARY(pfx##gt, bool, x_type, y_type, (cl_long)x >  (cl_long)y)	\
	STROM_SIMPLE_BINARY(pfx##ge, bool, x_type, y_type, (cl_long)x >= (cl_long)y)
STROM_INT_COMPARE_FAMILY(int2,  int2, int2)
STROM_INT_COMPARE_FAMILY(int24, int2, int4)
STROM_INT_COMPARE_FAMILY(int28, int2, int8)
STROM_INT_COMPARE_FAMILY(int42, int4, int2)
STROM_INT_COMPARE_FAMILY(int4,  int4, int4)
STROM_INT_COMPARE_FAMILY(int48, int4, int8)
STROM_INT_COMPARE_FAMILY(int82, int8, int2)
STROM_INT_COMPARE_FAMILY(int84, int8, int4)
STROM_INT_COMPARE_FAMILY(int8,  int8, int8)
STROM_SIMPLE_BINARY(booleq, bool, bool, bool, (x != 0) == (y != 0))
STROM_SIMPLE_BINARY(boolne, bool, bool, bool, (x != 0) != (y != 0))

/* bpchar(1) carried by value: bytewise (unsigned) comparison, the role of
 * opencl_textlib.h's bpchar* functions for one-byte keys */
STROM_SIMPLE_BINARY(char1eq, bool, char1, char1, (cl_uchar)x == (cl_uchar)y)
STROM_SIMPLE_BINARY(char1ne, bool, char1, char1, (cl_uchar)x != (cl_uchar)y)
STROM_SIMPLE_BINARY(char1lt, bool, char1, char1, (cl_uchar)x <  (cl_uchar)y)
STROM_SIMPLE_BINARY(char1le, bool, char1, char1, (cl_uchar)x <= (cl_uchar)y)
STROM_SIMPLE_BINARY(char1gt, bool, char1, char1, (cl_uchar)x >  (cl_uchar)y)
STROM_SIMPLE_BINARY(char1ge, bool, char1, char1, (cl_uchar)x >= (cl_uchar)y)
STROM_SIMPLE_BINARY(char1cmp, int4, char1, char1, devfunc_int_comp((cl_uchar)x, (cl_uchar)y))

/* float comparison through a 3-way compare that places NaN last */
STROM_DEVICE int strom_float_cmp(double x, double y)
{
	return devfunc_float_comp(x, y);
}
/*
 * The six comparisons written out for the same order (NaN above everything, NaN = NaN):
 * two IEEE compares each -- an unordered-aware one and an "is it a number" test -- instead
 * of the three-way compare's four and its selects.  x > y: x is NaN and y is not, or both are
 * numbers and x > y  ==  !(x <= y) && y == y.
 */
STROM_DEVICE bool strom_float_gt(double x, double y) { return !(x <= y) & !__builtin_isnan(y); }
STROM_DEVICE bool strom_float_lt(double x, double y) { return !(x >= y) & !__builtin_isnan(x); }
STROM_DEVICE bool strom_float_ge(double x, double y) { return !strom_float_lt(x, y); }
STROM_DEVICE bool strom_float_le(double x, double y) { return !strom_float_gt(x, y); }
STROM_DEVICE bool strom_float_eq(double x, double y) { return (x == y) | (__builtin_isnan(x) & __builtin_isnan(y)); }
STROM_DEVICE bool strom_float_ne(double x, double y) { return !strom_float_eq(x, y); }
#define STROM_FLOAT_COMPARE_FAMILY(pfx,x_type,y_type)										\
	STROM_SIMPLE_BINARY(pfx##eq, bool, x_type, y_type, strom_float_eq(x, y))				\
	STROM_SIMPLE_BINARY(pfx##ne, bool, x_type, y_type, strom_float_ne(x, y))				\
	STROM_SIMPLE_BINARY(pfx##lt, bool, x_type, y_type, strom_float_lt(x, y))				\
	STROM_SIMPLE_BINARY(pfx##le, bool, x_type, y_type, strom_float_le(x, y))				\
	STROM_SIMPLE_BINARY(pfx##gt, bool, x_type, y_type, strom_float_gt(x, y))				\
	STROM_SIMPLE_BINARY(pfx##ge, bool, x_type, y_type, strom_float_ge(x, y))
STROM_FLOAT_COMPARE_FAMILY(float4,  float4, float4)
STROM_FLOAT_COMPARE_FAMILY(float48, float4, float8)
STROM_FLOAT_COMPARE_FAMILY(float84, float8, float4)
STROM_FLOAT_COMPARE_FAMILY(float8,  float8, float8)

/* 3-way compare functions (used as GROUP BY / sort keys) */
STROM_SIMPLE_BINARY(btboolcmp,  int4, bool, bool, devfunc_int_comp((x != 0), (y != 0)))
STROM_SIMPLE_BINARY(btint2cmp,  int4, int2, int2, devfunc_int_comp(x, y))
STROM_SIMPLE_BINARY(btint24cmp, int4, int2, int4, devfunc_int_comp((cl_long)x, (cl_long)y))
STROM_SIMPLE_BINARY(btint28cmp, int4, int2, int8, devfunc_int_comp((cl_long)x, (cl_long)y))
STROM_SIMPLE_BINARY(btint42cmp, int4, int4, int2, devfunc_int_comp((cl_long)x, (cl_long)y))
STROM_SIMPLE_BINARY(btint4cmp,  int4, int4, int4, devfunc_int_comp(x, y))
STROM_SIMPLE_BINARY(btint48cmp, int4, int4, int8, devfunc_int_comp((cl_long)x, (cl_long)y))
STROM_SIMPLE_BINARY(btint82cmp, int4, int8, int2, devfunc_int_comp((cl_long)x, (cl_long)y))
STROM_SIMPLE_BINARY(btint84cmp, int4, int8, int4, devfunc_int_comp((cl_long)x, (cl_long)y))
STROM_SIMPLE_BINARY(btint8cmp,  int4, int8, int8, devfunc_int_comp(x, y))
STROM_SIMPLE_BINARY(btfloat4cmp,  int4, float4, float4, strom_float_cmp(x, y))
STROM_SIMPLE_BINARY(btfloat48cmp, int4, float4, float8, strom_float_cmp(x, y))
STROM_SIMPLE_BINARY(btfloat84cmp, int4, float8, float4, strom_float_cmp(x, y))
STROM_SIMPLE_BINARY(btfloat8cmp,  int4, float8, float8, strom_float_cmp(x, y))

/* ---- casts -------------------------------------------------------------- */
#define STROM_CAST_WIDEN(name,r_type,x_type)								\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1)						\
	{																		\
		pg_##r_type##_t result;												\
		result.isnull = arg1.isnull;										\
		result.value = (pg_##r_type##_base_t)arg1.value;					\
		return result;														\
	}
#define STROM_CAST_INT_NARROW(name,r_type,x_type,RMIN,RMAX)					\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1)						\
	{																		\
		pg_##r_type##_t result;												\
		result.isnull = arg1.isnull;										\
		result.value = 0;													\
		if (!result.isnull)													\
		{																	\
			if (arg1.value < (RMIN) || arg1.value > (RMAX))					\
				STROM_RECHECK();											\
			else															\
				result.value = (pg_##r_type##_base_t)arg1.value;			\
		}																	\
		return result;														\
	}
/* float -> int: round half to even, then range check (dtoi2/4/8) */
#define STROM_CAST_FLOAT_INT(name,r_type,x_type,RMIN,RMAX_EXCL)				\
	STROM_DEVICE pg_##r_type##_t											\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1)						\
	{																		\
		pg_##r_type##_t result;												\
		result.isnull = arg1.isnull;										\
		result.value = 0;													\
		if (!result.isnull)													\
		{																	\
			double r = __builtin_rint((double)arg1.value);					\
			if (__builtin_isnan(r) || r < (double)(RMIN) ||					\
				r >= (double)(RMAX_EXCL))									\
				STROM_RECHECK();											\
			else															\
				result.value = (pg_##r_type##_base_t)r;						\
		}																	\
		return result;														\
	}
/* float8 -> float4: overflow to inf / underflow to 0 go to the CPU */
STROM_DEVICE pg_float4_t
pgfn_float8_float4(cl_int *errcode, pg_float8_t arg1)
{
	pg_float4_t result;
	result.isnull = arg1.isnull;
	result.value = 0;
	if (!result.isnull)
	{
		result.value = (cl_float)arg1.value;
		if (STROM_CHECKFLOATVAL(result.value, __builtin_isinf(arg1.value),
								arg1.value == 0.0))
			STROM_RECHECK();
	}
	return result;
}
STROM_CAST_WIDEN(bool_int4,   int4,   bool)
STROM_CAST_WIDEN(int2_int4,   int4,   int2)
STROM_CAST_WIDEN(int2_int8,   int8,   int2)
STROM_CAST_WIDEN(int4_int8,   int8,   int4)
STROM_CAST_WIDEN(int2_float4, float4, int2)
STROM_CAST_WIDEN(int4_float4, float4, int4)
STROM_CAST_WIDEN(int8_float4, float4, int8)
STROM_CAST_WIDEN(int2_float8, float8, int2)
STROM_CAST_WIDEN(int4_float8, float8, int4)
STROM_CAST_WIDEN(int8_float8, float8, int8)
STROM_CAST_WIDEN(float4_float8, float8, float4)
STROM_CAST_INT_NARROW(int4_int2, int2, int4, -32768, 32767)
STROM_CAST_INT_NARROW(int8_int2, int2, int8, -32768, 32767)
STROM_CAST_INT_NARROW(int8_int4, int4, int8, -2147483648L, 2147483647L)
STROM_CAST_FLOAT_INT(float4_int2, int2, float4, -32768.0, 32768.0)
STROM_CAST_FLOAT_INT(float8_int2, int2, float8, -32768.0, 32768.0)
STROM_CAST_FLOAT_INT(float4_int4, int4, float4, -2147483648.0, 2147483648.0)
STROM_CAST_FLOAT_INT(float8_int4, int4, float8, -2147483648.0, 2147483648.0)
STROM_CAST_FLOAT_INT(float4_int8, int8, float4, -9223372036854775808.0, 9223372036854775808.0)
STROM_CAST_FLOAT_INT(float8_int8, int8, float8, -9223372036854775808.0, 9223372036854775808.0)

/* ---- misc math ---------------------------------------------------------- */
#define STROM_FLOAT8_FUNC1(name,EXPR)										\
	STROM_DEVICE pg_float8_t												\
	pgfn_##name(cl_int *errcode, pg_float8_t arg1)							\
	{																		\
		pg_float8_t result = arg1;											\
		if (!result.isnull)													\
		{																	\
			double x = arg1.value;											\
			result.value = (EXPR);											\
		}																	\
		return result;														\
	}
STROM_FLOAT8_FUNC1(ceil,   __builtin_ceil(x))
STROM_FLOAT8_FUNC1(floor,  __builtin_floor(x))
STROM_FLOAT8_FUNC1(round,  __builtin_rint(x))
STROM_FLOAT8_FUNC1(trunc,  __builtin_trunc(x))
STROM_FLOAT8_FUNC1(sign,   (x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0)))
STROM_DEVICE pg_float8_t
pgfn_dsqrt(cl_int *errcode, pg_float8_t arg1)
{
	pg_float8_t result = arg1;
	if (!result.isnull)
	{
		if (arg1.value < 0.0)
			STROM_RECHECK();
		else
			result.value = __builtin_sqrt(arg1.value);
	}
	return result;
}
STROM_DEVICE pg_float8_t
pgfn_dpi(cl_int *errcode)
{
	pg_float8_t result;
	result.isnull = false;
	result.value = 3.14159265358979323846;
	return result;
}


/* ---- transcendental functions (codegen.c:467-503; semantics of PostgreSQL 9.4's
 * float.c: a domain error, an infinite result of finite arguments or an
 * underflow to zero is an ERROR there -- here the row goes back to the CPU,
 * which then raises it).  ocml's double-precision functions are within 1-2 ulp
 * of glibc's; parity tests compare at 1e-14 relative. --------------------- */
#define STROM_FLOAT8_MATH1(name, DOMAIN_BAD, EXPR, INF_OK, ZERO_OK)			\
	STROM_DEVICE pg_float8_t												\
	pgfn_##name(cl_int *errcode, pg_float8_t arg1)							\
	{																		\
		pg_float8_t result = arg1;											\
		if (!result.isnull)													\
		{																	\
			double x = arg1.value;											\
			if (DOMAIN_BAD)													\
				STROM_RECHECK();											\
			else															\
			{																\
				result.value = (EXPR);										\
				if (STROM_CHECKFLOATVAL(result.value, INF_OK, ZERO_OK))		\
					STROM_RECHECK();										\
			}																\
		}																	\
		return result;														\
	}
STROM_FLOAT8_MATH1(dcbrt,   false,               cbrt(x),  __builtin_isinf(x), x == 0.0)
STROM_FLOAT8_MATH1(dexp,    false,               exp(x),   __builtin_isinf(x), false)
STROM_FLOAT8_MATH1(dlog1,   (x <= 0.0),          log(x),   __builtin_isinf(x), x == 1.0)
STROM_FLOAT8_MATH1(dlog10,  (x <= 0.0),          log10(x), __builtin_isinf(x), x == 1.0)
STROM_FLOAT8_MATH1(degrees, false,               x * (180.0 / 3.14159265358979323846), __builtin_isinf(x), x == 0.0)
STROM_FLOAT8_MATH1(radians, false,               x * (3.14159265358979323846 / 180.0), __builtin_isinf(x), x == 0.0)
STROM_FLOAT8_MATH1(dacos,   (x < -1.0 || x > 1.0), acos(x), false, true)
STROM_FLOAT8_MATH1(dasin,   (x < -1.0 || x > 1.0), asin(x), false, true)
STROM_FLOAT8_MATH1(datan,   false,               atan(x),  false, true)
STROM_FLOAT8_MATH1(dcos,    __builtin_isinf(x),  cos(x),   false, true)
STROM_FLOAT8_MATH1(dsin,    __builtin_isinf(x),  sin(x),   false, true)
STROM_FLOAT8_MATH1(dtan,    __builtin_isinf(x),  tan(x),   false, true)

STROM_DEVICE pg_float8_t
pgfn_dpow(cl_int *errcode, pg_float8_t arg1, pg_float8_t arg2)
{
	pg_float8_t result;
	result.isnull = arg1.isnull | arg2.isnull;
	result.value = 0.0;
	if (!result.isnull)
	{
		double x = arg1.value, y = arg2.value;
		/* "zero raised to a negative power is undefined", "a negative number
		 * raised to a non-integer power yields a complex result" (dpow) */
		if ((x == 0.0 && y < 0.0) || (x < 0.0 && __builtin_floor(y) != y))
			STROM_RECHECK();
		else
		{
			result.value = pow(x, y);
			if (STROM_CHECKFLOATVAL(result.value, __builtin_isinf(x) || __builtin_isinf(y), x == 0.0))
				STROM_RECHECK();
		}
	}
	return result;
}

STROM_DEVICE pg_float8_t
pgfn_datan2(cl_int *errcode, pg_float8_t arg1, pg_float8_t arg2)
{
	pg_float8_t result;
	result.isnull = arg1.isnull | arg2.isnull;
	result.value = (result.isnull ? 0.0 : atan2(arg1.value, arg2.value));
	return result;
}

#endif	/* STROM_MATHLIB_DEVICE_H */
