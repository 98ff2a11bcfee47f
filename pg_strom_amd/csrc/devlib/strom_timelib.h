/*
 * strom_timelib.h -- date / time / timestamp functions (device)
 *
 * Role in the reference: opencl_timelib.h (types 102-120, casts 227-320,
 * date +/- int 326-410, date<->timestamp comparisons 413-660).  date is
 * days since 2000-01-01 (int32), time and timestamp are microseconds
 * (int64, HAVE_INT64_TIMESTAMP).  Anything that would overflow goes back
 * to the CPU (CpuReCheck) exactly as the arithmetic library does.
 */
#ifndef STROM_TIMELIB_DEVICE_H
#define STROM_TIMELIB_DEVICE_H

#define STROM_USECS_PER_DAY		86400000000L
#define STROM_DATE_NOBEGIN		(-2147483647 - 1)
#define STROM_DATE_NOEND		2147483647
#define STROM_TS_NOBEGIN		(-9223372036854775807L - 1)
#define STROM_TS_NOEND			9223372036854775807L

#define STROM_TIME_COMPARE_FAMILY(pfx,TYPE)												\
	STROM_DEVICE pg_bool_t pgfn_##pfx##eq(cl_int *e, pg_##TYPE##_t a, pg_##TYPE##_t b)	\
	{ pg_bool_t r; r.isnull = a.isnull | b.isnull; r.value = (a.value == b.value); return r; }	\
	STROM_DEVICE pg_bool_t pgfn_##pfx##ne(cl_int *e, pg_##TYPE##_t a, pg_##TYPE##_t b)	\
	{ pg_bool_t r; r.isnull = a.isnull | b.isnull; r.value = (a.value != b.value); return r; }	\
	STROM_DEVICE pg_bool_t pgfn_##pfx##lt(cl_int *e, pg_##TYPE##_t a, pg_##TYPE##_t b)	\
	{ pg_bool_t r; r.isnull = a.isnull | b.isnull; r.value = (a.value <  b.value); return r; }	\
	STROM_DEVICE pg_bool_t pgfn_##pfx##le(cl_int *e, pg_##TYPE##_t a, pg_##TYPE##_t b)	\
	{ pg_bool_t r; r.isnull = a.isnull | b.isnull; r.value = (a.value <= b.value); return r; }	\
	STROM_DEVICE pg_bool_t pgfn_##pfx##gt(cl_int *e, pg_##TYPE##_t a, pg_##TYPE##_t b)	\
	{ pg_bool_t r; r.isnull = a.isnull | b.isnull; r.value = (a.value >  b.value); return r; }	\
	STROM_DEVICE pg_bool_t pgfn_##pfx##ge(cl_int *e, pg_##TYPE##_t a, pg_##TYPE##_t b)	\
	{ pg_bool_t r; r.isnull = a.isnull | b.isnull; r.value = (a.value >= b.value); return r; }	\
	STROM_DEVICE pg_int4_t pgfn_##pfx##cmp(cl_int *e, pg_##TYPE##_t a, pg_##TYPE##_t b)	\
	{ pg_int4_t r; r.isnull = a.isnull | b.isnull;										\
	  r.value = devfunc_int_comp(a.value, b.value); return r; }

STROM_TIME_COMPARE_FAMILY(date_, date)
STROM_TIME_COMPARE_FAMILY(time_, time)
STROM_TIME_COMPARE_FAMILY(timestamp_, timestamp)

/* date -> timestamp, +-infinity preserved, overflow -> CpuReCheck */
STROM_DEVICE pg_timestamp_t
pgfn_date_timestamp(cl_int *errcode, pg_date_t arg1)
{
	pg_timestamp_t result;
	result.isnull = arg1.isnull;
	result.value = 0;
	if (!result.isnull)
	{
		if (arg1.value == STROM_DATE_NOBEGIN)		result.value = STROM_TS_NOBEGIN;
		else if (arg1.value == STROM_DATE_NOEND)	result.value = STROM_TS_NOEND;
		else if (__builtin_mul_overflow((cl_long)arg1.value, STROM_USECS_PER_DAY, &result.value))
		{
			result.isnull = true;
			STROM_SET_ERROR(errcode, StromError_CpuReCheck);
		}
	}
	return result;
}

/* date(date), time(time), timestamp(timestamp): the catalog's alias casts (codegen.c:543-548, "ta/c:") */
STROM_DEVICE pg_date_t pgfn_date_date(cl_int *errcode, pg_date_t arg1) { return arg1; }
STROM_DEVICE pg_time_t pgfn_time_time(cl_int *errcode, pg_time_t arg1) { return arg1; }
STROM_DEVICE pg_timestamp_t pgfn_timestamp_timestamp(cl_int *errcode, pg_timestamp_t arg1) { return arg1; }

STROM_DEVICE pg_date_t
pgfn_timestamp_date(cl_int *errcode, pg_timestamp_t arg1)
{
	pg_date_t result;
	result.isnull = arg1.isnull;
	result.value = 0;
	if (!result.isnull)
	{
		if (arg1.value == STROM_TS_NOBEGIN)			result.value = STROM_DATE_NOBEGIN;
		else if (arg1.value == STROM_TS_NOEND)		result.value = STROM_DATE_NOEND;
		else
		{
			cl_long d = arg1.value / STROM_USECS_PER_DAY;
			if (arg1.value % STROM_USECS_PER_DAY < 0)
				d--;
			result.value = (cl_int)d;
		}
	}
	return result;
}

STROM_DEVICE pg_time_t
pgfn_timestamp_time(cl_int *errcode, pg_timestamp_t arg1)
{
	pg_time_t result;
	result.isnull = arg1.isnull;
	result.value = 0;
	if (!result.isnull)
	{
		if (arg1.value == STROM_TS_NOBEGIN || arg1.value == STROM_TS_NOEND)
			result.isnull = true;
		else
		{
			cl_long t = arg1.value % STROM_USECS_PER_DAY;
			result.value = (t < 0 ? t + STROM_USECS_PER_DAY : t);
		}
	}
	return result;
}

#define STROM_DATE_INT_OP(name,x_type,y_type,r_type,OVF)								\
	STROM_DEVICE pg_##r_type##_t														\
	pgfn_##name(cl_int *errcode, pg_##x_type##_t arg1, pg_##y_type##_t arg2)			\
	{																					\
		pg_##r_type##_t result;															\
		result.isnull = arg1.isnull | arg2.isnull;										\
		result.value = 0;																\
		if (!result.isnull && OVF(arg1.value, arg2.value, &result.value))				\
		{																				\
			result.isnull = true;														\
			STROM_SET_ERROR(errcode, StromError_CpuReCheck);							\
		}																				\
		return result;																	\
	}
STROM_DATE_INT_OP(date_pli, date, int4, date, __builtin_add_overflow)
STROM_DATE_INT_OP(date_mii, date, int4, date, __builtin_sub_overflow)
STROM_DATE_INT_OP(date_mi,  date, date, int4, __builtin_sub_overflow)
STROM_DATE_INT_OP(integer_pl_date, int4, date, date, __builtin_add_overflow)

STROM_DEVICE pg_timestamp_t
pgfn_datetime_pl(cl_int *errcode, pg_date_t arg1, pg_time_t arg2)
{
	pg_timestamp_t result = pgfn_date_timestamp(errcode, arg1);
	result.isnull |= arg2.isnull;
	if (!result.isnull && result.value != STROM_TS_NOBEGIN && result.value != STROM_TS_NOEND &&
		__builtin_add_overflow(result.value, arg2.value, &result.value))
	{
		result.isnull = true;
		STROM_SET_ERROR(errcode, StromError_CpuReCheck);
	}
	return result;
}
STROM_DEVICE pg_timestamp_t
pgfn_timedate_pl(cl_int *errcode, pg_time_t arg1, pg_date_t arg2)
{
	return pgfn_datetime_pl(errcode, arg2, arg1);
}

/* date <op> timestamp: promote the date, then compare */
#define STROM_DATE_TS_COMPARE(op)																\
	STROM_DEVICE pg_bool_t																		\
	pgfn_date_##op##_timestamp(cl_int *errcode, pg_date_t arg1, pg_timestamp_t arg2)			\
	{ return pgfn_timestamp_##op(errcode, pgfn_date_timestamp(errcode, arg1), arg2); }			\
	STROM_DEVICE pg_bool_t																		\
	pgfn_timestamp_##op##_date(cl_int *errcode, pg_timestamp_t arg1, pg_date_t arg2)			\
	{ return pgfn_timestamp_##op(errcode, arg1, pgfn_date_timestamp(errcode, arg2)); }
STROM_DATE_TS_COMPARE(eq)
STROM_DATE_TS_COMPARE(ne)
STROM_DATE_TS_COMPARE(lt)
STROM_DATE_TS_COMPARE(le)
STROM_DATE_TS_COMPARE(gt)
STROM_DATE_TS_COMPARE(ge)
STROM_DEVICE pg_int4_t
pgfn_date_cmp_timestamp(cl_int *errcode, pg_date_t arg1, pg_timestamp_t arg2)
{ return pgfn_timestamp_cmp(errcode, pgfn_date_timestamp(errcode, arg1), arg2); }
STROM_DEVICE pg_int4_t
pgfn_timestamp_cmp_date(cl_int *errcode, pg_timestamp_t arg1, pg_date_t arg2)
{ return pgfn_timestamp_cmp(errcode, arg1, pgfn_date_timestamp(errcode, arg2)); }

#endif	/* STROM_TIMELIB_DEVICE_H */
