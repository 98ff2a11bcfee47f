/*
 * codegen.cpp -- expression IR -> HIP __device__ text
 *
 * Role in the reference: codegen.c (type catalog 46-78, function catalog
 * 211-630, template expanders 632-861, expression walker 1065-1392,
 * declaration emitters 1435-1623, availability check 1631-1759).  The
 * PostgreSQL-facing half (Node trees, syscache) does not exist here; the
 * tree arrives as the S-expression IR described in strom_codegen.h.
 *
 * Output convention (differs from the OpenCL text on purpose): the
 * kernel skeletons load referenced columns with vector loads before any
 * expression runs, so a generated function never touches the chunk.  It
 * receives every KPARAM_i / KVAR_n already materialised:
 *
 *     #define STROM_KPARAM_LIST(X)  X(0,int4) X(1,float8)
 *     #define STROM_KVAR_LIST(X)    X(1,0,int4) X(2,1,float8)
 *     #include "strom_gpuscan.h"
 *     STROM_DEVICE pg_bool_t gpuscan_qual_eval(cl_int *errcode,
 *             const strom_kparams &KP, const strom_kvars &KV)
 *     { return <expr over KP.KPARAM_i, KV.KVAR_n>; }
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <stdexcept>

#include "strom_codegen.h"
#include "strom_hip.h"
#include "codegen_internal.h"

namespace strom {

/* ------------------------------------------------------------------ *
 * type catalog
 * ------------------------------------------------------------------ */
static const devtype_info devtype_catalog[] = {
	/* oid, sql name, device name, length, flags */
	{ STROM_BOOLOID,      "bool",      "bool",      1, 0 },
	{ STROM_INT2OID,      "int2",      "int2",      2, 0 },
	{ STROM_INT4OID,      "int4",      "int4",      4, 0 },
	{ STROM_INT8OID,      "int8",      "int8",      8, 0 },
	{ STROM_FLOAT4OID,    "float4",    "float4",    4, 0 },
	{ STROM_FLOAT8OID,    "float8",    "float8",    8, 0 },
	{ STROM_DATEOID,      "date",      "date",      4, DEVFUNC_NEEDS_TIMELIB },
	{ STROM_TIMEOID,      "time",      "time",      8, DEVFUNC_NEEDS_TIMELIB },
	{ STROM_TIMESTAMPOID, "timestamp", "timestamp", 8, DEVFUNC_NEEDS_TIMELIB },
	{ STROM_NUMERICOID,   "numeric",   "numeric",   8, DEVFUNC_NEEDS_NUMERIC },
	{ STROM_BPCHAROID,    "char1",     "char1",     1, 0 },
	/* numeric(p,s) stored as int8 at 10^-s ("decimal64", COLUMN chunks): (var N decimal S) */
	{ STROM_DECIMALOID,   "decimal",   "decimal",   8, DEVFUNC_NEEDS_NUMERIC },
	/* varlena: a value is the address of its datum (strom_textlib.h); row formats only */
	{ STROM_TEXTOID,      "text",      "text",     -1, DEVFUNC_NEEDS_TEXTLIB | DEVTYPE_IS_VARLENA },
	{ STROM_BPCHARNOID,   "character", "bpcharn",  -1, DEVFUNC_NEEDS_TEXTLIB | DEVTYPE_IS_VARLENA },
};

const devtype_info *
devtype_lookup(int oid)
{
	for (const auto &t : devtype_catalog)
		if (t.type_oid == oid)
			return &t;
	return nullptr;
}

const devtype_info *
devtype_lookup_by_name(const std::string &name)
{
	for (const auto &t : devtype_catalog)
		if (name == t.sql_name)
			return &t;
	/* a few spellings SQL users expect */
	if (name == "integer" || name == "int")		return devtype_lookup(STROM_INT4OID);
	if (name == "smallint")						return devtype_lookup(STROM_INT2OID);
	if (name == "bigint")						return devtype_lookup(STROM_INT8OID);
	if (name == "real")							return devtype_lookup(STROM_FLOAT4OID);
	if (name == "double" || name == "float")	return devtype_lookup(STROM_FLOAT8OID);
	if (name == "bpchar")						return devtype_lookup(STROM_BPCHAROID);
	return nullptr;
}

/* ------------------------------------------------------------------ *
 * function catalog: pg_proc name + argument types -> device function
 * ------------------------------------------------------------------ */
struct devfunc_info {
	std::string			name;
	std::vector<int>	argtypes;
	int					rettype;
	std::string			devname;	/* pgfn_<devname> */
	int					flags;
};

static std::vector<devfunc_info> devfunc_catalog;

static void
add_func(const std::string &name, std::vector<int> args, int ret,
		 const std::string &devname, int flags)
{
	devfunc_catalog.push_back(devfunc_info{name, std::move(args), ret, devname, flags});
}

static void
build_catalog(void)
{
	const int B = STROM_BOOLOID, I2 = STROM_INT2OID, I4 = STROM_INT4OID,
		I8 = STROM_INT8OID, F4 = STROM_FLOAT4OID, F8 = STROM_FLOAT8OID,
		DT = STROM_DATEOID, TM = STROM_TIMEOID, TS = STROM_TIMESTAMPOID,
		NU = STROM_NUMERICOID, C1 = STROM_BPCHAROID;
	const int M = DEVFUNC_NEEDS_MATHLIB, T = DEVFUNC_NEEDS_TIMELIB,
		N = DEVFUNC_NEEDS_NUMERIC;

	/* casts: SQL function named after the target type; device name is
	 * <source>_<target> (the alias rule of devfunc_setup_cast) */
	struct { const char *fn; int ret; const char *rname; } targets[] = {
		{"int2", I2, "int2"}, {"int4", I4, "int4"}, {"int8", I8, "int8"},
		{"float4", F4, "float4"}, {"float8", F8, "float8"},
	};
	struct { int oid; const char *name; } sources[] = {
		{I2,"int2"},{I4,"int4"},{I8,"int8"},{F4,"float4"},{F8,"float8"},
	};
	for (auto &t : targets)
		for (auto &s : sources)
			if (t.ret != s.oid)
				add_func(t.fn, {s.oid}, t.ret, std::string(s.name) + "_" + t.rname, M);
	add_func("int4", {B}, I4, "bool_int4", M);

	/* integer arithmetic families; result = wider operand (PostgreSQL) */
	struct { const char *pfx; int x, y, r; } ifam[] = {
		{"int2", I2,I2,I2}, {"int24",I2,I4,I4}, {"int28",I2,I8,I8},
		{"int42",I4,I2,I4}, {"int4", I4,I4,I4}, {"int48",I4,I8,I8},
		{"int82",I8,I2,I8}, {"int84",I8,I4,I8}, {"int8", I8,I8,I8},
	};
	for (auto &f : ifam)
	{
		for (const char *op : {"pl","mi","mul","div"})
			add_func(std::string(f.pfx) + op, {f.x,f.y}, f.r, std::string(f.pfx) + op, M);
		for (const char *op : {"eq","ne","lt","le","gt","ge"})
			add_func(std::string(f.pfx) + op, {f.x,f.y}, B, std::string(f.pfx) + op, M);
		add_func(std::string("bt") + f.pfx + "cmp", {f.x,f.y}, I4,
				 std::string("bt") + f.pfx + "cmp", M);
	}
	struct { const char *pfx; int x, y, r; } ffam[] = {
		{"float4", F4,F4,F4}, {"float48",F4,F8,F8},
		{"float84",F8,F4,F8}, {"float8", F8,F8,F8},
	};
	for (auto &f : ffam)
	{
		for (const char *op : {"pl","mi","mul","div"})
			add_func(std::string(f.pfx) + op, {f.x,f.y}, f.r, std::string(f.pfx) + op, M);
		for (const char *op : {"eq","ne","lt","le","gt","ge"})
			add_func(std::string(f.pfx) + op, {f.x,f.y}, B, std::string(f.pfx) + op, M);
		add_func(std::string("bt") + f.pfx + "cmp", {f.x,f.y}, I4,
				 std::string("bt") + f.pfx + "cmp", M);
	}
	struct { const char *n; int t; } ints[] = {{"int2",I2},{"int4",I4},{"int8",I8}};
	for (auto &i : ints)
	{
		std::string n = i.n;
		add_func(n + "mod", {i.t,i.t}, i.t, n + "mod", M);
		add_func(n + "um",  {i.t}, i.t, n + "um", M);
		add_func(n + "up",  {i.t}, i.t, n + "up", M);
		add_func(n + "abs", {i.t}, i.t, n + "abs", M);
		add_func("abs",     {i.t}, i.t, n + "abs", M);
		add_func(n + "not", {i.t}, i.t, n + "not", M);
		add_func(n + "and", {i.t,i.t}, i.t, n + "and", M);
		add_func(n + "or",  {i.t,i.t}, i.t, n + "or", M);
		add_func(n + "xor", {i.t,i.t}, i.t, n + "xor", M);
		add_func(n + "shl", {i.t,I4}, i.t, n + "shl", M);
		add_func(n + "shr", {i.t,I4}, i.t, n + "shr", M);
	}
	struct { const char *n; int t; } flts[] = {{"float4",F4},{"float8",F8}};
	for (auto &f : flts)
	{
		std::string n = f.n;
		add_func(n + "um",  {f.t}, f.t, n + "um", M);
		add_func(n + "up",  {f.t}, f.t, n + "up", M);
		add_func(n + "abs", {f.t}, f.t, n + "abs", M);
		add_func("abs",     {f.t}, f.t, n + "abs", M);
	}
	add_func("booleq", {B,B}, B, "booleq", M);
	add_func("boolne", {B,B}, B, "boolne", M);
	add_func("btboolcmp", {B,B}, I4, "btboolcmp", M);
	for (const char *n : {"ceil","ceiling"})	add_func(n, {F8}, F8, "ceil", M);
	add_func("floor", {F8}, F8, "floor", M);
	for (const char *n : {"round","dround"})	add_func(n, {F8}, F8, "round", M);
	for (const char *n : {"trunc","dtrunc"})	add_func(n, {F8}, F8, "trunc", M);
	add_func("sign", {F8}, F8, "sign", M);
	for (const char *n : {"sqrt","dsqrt"})		add_func(n, {F8}, F8, "dsqrt", M);
	add_func("pi", {}, F8, "dpi", M);
	/* transcendental functions (codegen.c:467-503) */
	for (const char *n : {"cbrt","dcbrt"})			add_func(n, {F8}, F8, "dcbrt", M);
	for (const char *n : {"exp","dexp"})			add_func(n, {F8}, F8, "dexp", M);
	for (const char *n : {"ln","dlog1"})			add_func(n, {F8}, F8, "dlog1", M);
	for (const char *n : {"log","dlog10"})			add_func(n, {F8}, F8, "dlog10", M);
	for (const char *n : {"power","pow","dpow"})	add_func(n, {F8,F8}, F8, "dpow", M);
	add_func("degrees", {F8}, F8, "degrees", M);
	add_func("radians", {F8}, F8, "radians", M);
	add_func("acos", {F8}, F8, "dacos", M);
	add_func("asin", {F8}, F8, "dasin", M);
	add_func("atan", {F8}, F8, "datan", M);
	add_func("atan2", {F8,F8}, F8, "datan2", M);
	add_func("cos", {F8}, F8, "dcos", M);
	add_func("sin", {F8}, F8, "dsin", M);
	add_func("tan", {F8}, F8, "dtan", M);

	/* date / time / timestamp (timelib) */
	for (const char *op : {"eq","ne","lt","le","gt","ge"})
	{
		add_func(std::string("date_") + op, {DT,DT}, B, std::string("date_") + op, T);
		add_func(std::string("time_") + op, {TM,TM}, B, std::string("time_") + op, T);
		add_func(std::string("timestamp_") + op, {TS,TS}, B, std::string("timestamp_") + op, T);
		add_func(std::string("date_") + op + "_timestamp", {DT,TS}, B,
				 std::string("date_") + op + "_timestamp", T);
		add_func(std::string("timestamp_") + op + "_date", {TS,DT}, B,
				 std::string("timestamp_") + op + "_date", T);
	}
	add_func("date_cmp", {DT,DT}, I4, "date_cmp", T);
	add_func("time_cmp", {TM,TM}, I4, "time_cmp", T);
	add_func("timestamp_cmp", {TS,TS}, I4, "timestamp_cmp", T);
	add_func("date_cmp_timestamp", {DT,TS}, I4, "date_cmp_timestamp", T);
	add_func("timestamp_cmp_date", {TS,DT}, I4, "timestamp_cmp_date", T);
	add_func("date_pli", {DT,I4}, DT, "date_pli", T);
	add_func("date_mii", {DT,I4}, DT, "date_mii", T);
	add_func("date_mi",  {DT,DT}, I4, "date_mi", T);
	add_func("integer_pl_date", {I4,DT}, DT, "integer_pl_date", T);
	add_func("datetime_pl", {DT,TM}, TS, "datetime_pl", T);
	add_func("timedate_pl", {TM,DT}, TS, "timedate_pl", T);
	add_func("date", {DT}, DT, "date_date", T);					/* alias casts (codegen.c:543-548) */
	add_func("time", {TM}, TM, "time_time", T);
	add_func("timestamp", {TS}, TS, "timestamp_timestamp", T);
	add_func("date", {TS}, DT, "timestamp_date", T);
	add_func("time", {TS}, TM, "timestamp_time", T);
	add_func("timestamp", {DT}, TS, "date_timestamp", T);

	/* numeric (64-bit in-kernel form) */
	struct { const char *fn; int ret; const char *dev; } ncast_out[] = {
		{"int2", I2, "numeric_int2"}, {"int4", I4, "numeric_int4"},
		{"int8", I8, "numeric_int8"}, {"float4", F4, "numeric_float4"},
		{"float8", F8, "numeric_float8"},
	};
	for (auto &c : ncast_out)
		add_func(c.fn, {NU}, c.ret, c.dev, N);
	add_func("numeric", {I2}, NU, "int2_numeric", N);
	add_func("numeric", {I4}, NU, "int4_numeric", N);
	add_func("numeric", {I8}, NU, "int8_numeric", N);
	/* float -> numeric: the value at FLT_DIG / DBL_DIG significant digits (strom_numeric.h) */
	add_func("numeric", {F4}, NU, "float4_numeric", N);
	add_func("numeric", {F8}, NU, "float8_numeric", N);
	for (const char *op : {"add","sub","mul"})
		add_func(std::string("numeric_") + op, {NU,NU}, NU, std::string("numeric_") + op, N);
	add_func("numeric_uplus",  {NU}, NU, "numeric_uplus", N);
	add_func("numeric_uminus", {NU}, NU, "numeric_uminus", N);
	add_func("numeric_abs",    {NU}, NU, "numeric_abs", N);
	add_func("abs",            {NU}, NU, "numeric_abs", N);
	for (const char *op : {"eq","ne","lt","le","gt","ge"})
		add_func(std::string("numeric_") + op, {NU,NU}, B, std::string("numeric_") + op, N);
	add_func("numeric_cmp", {NU,NU}, I4, "numeric_cmp", N);

	/* character(n) and text as PostgreSQL stores them (codegen.c:616-629, opencl_textlib.h) */
	{
		const int CN = STROM_BPCHARNOID, TX = STROM_TEXTOID, S = DEVFUNC_NEEDS_TEXTLIB;
		for (const char *op : {"eq", "ne", "lt", "le", "gt", "ge"})
		{
			add_func(std::string("bpchar") + op, {CN,CN}, B, std::string("bpchar") + op, S);
			bool	eqne = (op[0] == 'e' || op[0] == 'n');
			add_func(std::string(eqne ? "text" : "text_") + op, {TX,TX}, B,
					 std::string(eqne ? "text" : "text_") + op, S);
		}
		add_func("bpcharcmp", {CN,CN}, I4, "bpcharcmp", S);
		add_func("bttextcmp", {TX,TX}, I4, "text_cmp", S);
	}
	/* bpchar(1) by value: bytewise comparison (textlib's role for Q1 keys) */
	for (const char *op : {"eq","ne","lt","le","gt","ge"})
		add_func(std::string("bpchar") + op, {C1,C1}, B, std::string("char1") + op, M);
	add_func("bpcharcmp", {C1,C1}, I4, "char1cmp", M);
}

static const devfunc_info *
devfunc_lookup(const std::string &name, const std::vector<int> &argtypes)
{
	if (devfunc_catalog.empty())
		build_catalog();
	for (const auto &f : devfunc_catalog)
		if (f.name == name && f.argtypes == argtypes)
			return &f;
	return nullptr;
}

/* equality / comparison function per type (devtype_info.type_eqfunc / cmpfunc) */
const char *
devtype_eqfunc(int oid)
{
	switch (oid)
	{
		case STROM_BOOLOID:		return "booleq";
		case STROM_INT2OID:		return "int2eq";
		case STROM_INT4OID:		return "int4eq";
		case STROM_INT8OID:		return "int8eq";
		case STROM_FLOAT4OID:	return "float4eq";
		case STROM_FLOAT8OID:	return "float8eq";
		case STROM_DATEOID:		return "date_eq";
		case STROM_TIMEOID:		return "time_eq";
		case STROM_TIMESTAMPOID:return "timestamp_eq";
		case STROM_NUMERICOID:	return "numeric_eq";
		case STROM_BPCHAROID:	return "bpchareq";
		case STROM_BPCHARNOID:	return "bpchareq";
		case STROM_TEXTOID:		return "texteq";
	}
	return nullptr;
}
const char *
devtype_cmpfunc(int oid)
{
	switch (oid)
	{
		case STROM_BOOLOID:		return "btboolcmp";
		case STROM_INT2OID:		return "btint2cmp";
		case STROM_INT4OID:		return "btint4cmp";
		case STROM_INT8OID:		return "btint8cmp";
		case STROM_FLOAT4OID:	return "btfloat4cmp";
		case STROM_FLOAT8OID:	return "btfloat8cmp";
		case STROM_DATEOID:		return "date_cmp";
		case STROM_TIMEOID:		return "time_cmp";
		case STROM_TIMESTAMPOID:return "timestamp_cmp";
		case STROM_NUMERICOID:	return "numeric_cmp";
		case STROM_BPCHAROID:	return "bpcharcmp";
	}
	return nullptr;
}

/* ------------------------------------------------------------------ *
 * S-expression reader
 * ------------------------------------------------------------------ */
[[noreturn]] void
codegen_error(const char *fmt, ...)
{
	char	buf[1024];
	va_list	ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	throw std::runtime_error(buf);
}

static void
skip_ws(const char *&p)
{
	while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')
		p++;
}

static sexpr
parse_sexpr(const char *&p)
{
	sexpr	node;

	skip_ws(p);
	if (*p == '\0')
		codegen_error("unexpected end of expression");
	if (*p == '(')
	{
		p++;
		node.is_list = true;
		for (;;)
		{
			skip_ws(p);
			if (*p == '\0')
				codegen_error("missing ')'");
			if (*p == ')')
			{
				p++;
				break;
			}
			node.items.push_back(parse_sexpr(p));
		}
		return node;
	}
	if (*p == ')')
		codegen_error("unexpected ')'");
	node.is_list = false;
	if (*p == '\'')
	{
		/* quoted literal */
		p++;
		while (*p && *p != '\'')
			node.atom.push_back(*p++);
		if (*p != '\'')
			codegen_error("unterminated quoted literal");
		p++;
		return node;
	}
	while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r' &&
		   *p != '(' && *p != ')')
		node.atom.push_back(*p++);
	return node;
}

sexpr
sexpr_parse(const char *text)
{
	const char *p = text;
	sexpr	node = parse_sexpr(p);
	skip_ws(p);
	if (*p != '\0')
		codegen_error("trailing characters after expression: \"%.20s\"", p);
	return node;
}

/* ------------------------------------------------------------------ *
 * literal -> datum image
 * ------------------------------------------------------------------ */
static int
date2j(int y, int m, int d)
{
	int		julian, century;

	if (m > 2)	{ m += 1; y += 4800; }
	else		{ m += 13; y += 4799; }
	century = y / 100;
	julian = y * 365 - 32167;
	julian += y / 4 - century + century / 4;
	julian += 7834 * m / 256 + d;
	return julian;
}
#define POSTGRES_EPOCH_JDATE	2451545

/*
 * decimal literal -> 64-bit device numeric: 6-bit base-10 exponent (63..58),
 * sign (57), 57-bit mantissa (opencl_numeric.h:122-162).  Trailing zeros of
 * the mantissa are folded into the exponent so equal values get one image.
 */
bool
numeric_literal_to_kernel(const std::string &lit, uint64_t *out)
{
	const char *p = lit.c_str();
	bool		neg = false;
	unsigned __int128 mant = 0;
	int			expo = 0;
	bool		seen_digit = false, seen_dot = false;

	if (*p == '+' || *p == '-')
		neg = (*p++ == '-');
	for (; *p; p++)
	{
		if (*p >= '0' && *p <= '9')
		{
			mant = mant * 10 + (unsigned)(*p - '0');
			if (mant >> 100)
				return false;
			if (seen_dot)
				expo--;
			seen_digit = true;
		}
		else if (*p == '.' && !seen_dot)
			seen_dot = true;
		else
			break;
	}
	if (!seen_digit)
		return false;
	if (*p == 'e' || *p == 'E')
	{
		char *end;
		long e = strtol(p + 1, &end, 10);
		if (end == p + 1 || *end != '\0')
			return false;
		expo += (int)e;
	}
	else if (*p != '\0')
		return false;
	if (mant == 0)
	{
		*out = 0;
		return true;
	}
	while (mant % 10 == 0)
	{
		mant /= 10;
		expo++;
	}
	/* an exponent above the field's range can be traded for mantissa digits */
	while (expo > 31 && mant < ((unsigned __int128)1 << 57) / 10)
	{
		mant *= 10;
		expo--;
	}
	if (mant >= ((unsigned __int128)1 << 57) || expo < -32 || expo > 31)
		return false;
	*out = (((uint64_t)(int64_t)expo) << 58) | (neg ? (1ULL << 57) : 0) |
		((uint64_t)mant & ((1ULL << 57) - 1));
	return true;
}

static void
literal_to_datum(const devtype_info *dtype, const std::string &lit,
				 strom_kparam_desc *desc)
{
	const char *s = lit.c_str();
	char	   *end = nullptr;

	memset(desc->value, 0, sizeof(desc->value));
	desc->length = dtype->type_length;
	switch (dtype->type_oid)
	{
		case STROM_BOOLOID:
			{
				bool v;
				if (lit == "t" || lit == "true" || lit == "1")	v = true;
				else if (lit == "f" || lit == "false" || lit == "0")	v = false;
				else codegen_error("invalid bool literal \"%s\"", s);
				desc->value[0] = v;
			}
			break;
		case STROM_INT2OID: case STROM_INT4OID: case STROM_INT8OID:
		case STROM_TIMEOID:
			{
				long long v = strtoll(s, &end, 10);
				if (end == s || *end != '\0')
					codegen_error("invalid integer literal \"%s\"", s);
				if ((dtype->type_oid == STROM_INT2OID && (v < -32768 || v > 32767)) ||
					(dtype->type_oid == STROM_INT4OID && (v < -2147483648LL || v > 2147483647LL)))
					codegen_error("integer literal \"%s\" out of range for %s", s, dtype->sql_name);
				memcpy(desc->value, &v, dtype->type_length);
			}
			break;
		case STROM_FLOAT4OID:
			{
				float v = strtof(s, &end);
				if (end == s || *end != '\0')
					codegen_error("invalid float literal \"%s\"", s);
				memcpy(desc->value, &v, 4);
			}
			break;
		case STROM_FLOAT8OID:
			{
				double v = strtod(s, &end);
				if (end == s || *end != '\0')
					codegen_error("invalid float literal \"%s\"", s);
				memcpy(desc->value, &v, 8);
			}
			break;
		case STROM_DATEOID:
			{
				int y, m, d;
				int32_t v;
				if (sscanf(s, "%d-%d-%d", &y, &m, &d) == 3 && strchr(s + 1, '-'))
					v = date2j(y, m, d) - POSTGRES_EPOCH_JDATE;
				else
				{
					long long t = strtoll(s, &end, 10);
					if (end == s || *end != '\0')
						codegen_error("invalid date literal \"%s\"", s);
					v = (int32_t)t;
				}
				memcpy(desc->value, &v, 4);
			}
			break;
		case STROM_TIMESTAMPOID:
			{
				int y, m, d, hh = 0, mi = 0;
				double ss = 0.0;
				int64_t v;
				int n = sscanf(s, "%d-%d-%d %d:%d:%lf", &y, &m, &d, &hh, &mi, &ss);
				if (n >= 3 && strchr(s + 1, '-'))
				{
					int64_t days = date2j(y, m, d) - POSTGRES_EPOCH_JDATE;
					v = days * 86400000000LL +
						((int64_t)hh * 3600 + (int64_t)mi * 60) * 1000000LL +
						(int64_t)llround(ss * 1000000.0);
				}
				else
				{
					v = strtoll(s, &end, 10);
					if (end == s || *end != '\0')
						codegen_error("invalid timestamp literal \"%s\"", s);
				}
				memcpy(desc->value, &v, 8);
			}
			break;
		case STROM_NUMERICOID:
			{
				uint64_t v;
				if (!numeric_literal_to_kernel(lit, &v))
					codegen_error("numeric literal \"%s\" does not fit the 64-bit device form", s);
				memcpy(desc->value, &v, 8);
			}
			break;
		case STROM_BPCHAROID:
			if (lit.size() != 1)
				codegen_error("char1 literal must be one byte: \"%s\"", s);
			desc->value[0] = (uint8_t)lit[0];
			break;
		case STROM_TEXTOID: case STROM_BPCHARNOID:
			{
				/* the varlena image (4-byte header) is kept on the heap, its address in
				 * value[0..7]; strom_codegen_release() frees it */
				uint32_t	hdr = (uint32_t)((lit.size() + 4) << 2);
				char	   *img = (char *)malloc(lit.size() + 4);
				if (!img)
					codegen_error("out of memory");
				memcpy(img, &hdr, 4);
				memcpy(img + 4, lit.data(), lit.size());
				memcpy(desc->value, &img, sizeof(img));
				desc->length = (int32_t)lit.size() + 4;
			}
			break;
		default:
			codegen_error("no literal syntax for type %s", dtype->sql_name);
	}
}

/* ------------------------------------------------------------------ *
 * the walker
 * ------------------------------------------------------------------ */
int
codegen_context::track_param(const strom_kparam_desc &d)
{
	for (size_t i = 0; i < used_params.size(); i++)
	{
		const strom_kparam_desc &o = used_params[i];
		if ((d.type_oid == STROM_TEXTOID || d.type_oid == STROM_BPCHARNOID) && d.is_const && !d.isnull)
		{
			/* value[] holds the address of the image: compare what it points to */
			const char *po, *pd;
			memcpy(&po, o.value, sizeof(po));
			memcpy(&pd, d.value, sizeof(pd));
			if (o.type_oid == d.type_oid && o.is_const && !o.isnull && o.length == d.length &&
				memcmp(po, pd, d.length) == 0)
			{
				free((void *)pd);
				return (int)i;
			}
			continue;
		}
		if (o.type_oid == d.type_oid && o.is_const == d.is_const &&
			o.param_id == d.param_id && o.isnull == d.isnull &&
			o.length == d.length && memcmp(o.value, d.value, sizeof(d.value)) == 0)
			return (int)i;
	}
	used_params.push_back(d);
	return (int)used_params.size() - 1;
}

void
codegen_context::track_var(int attno, int type_oid)
{
	for (auto &v : used_vars)
	{
		if (v.attno == attno)
		{
			if (v.type_oid != type_oid)
				codegen_error("attribute %d referenced as two different types", attno);
			return;
		}
	}
	used_vars.push_back(strom_kvar_desc{attno, type_oid});
}

static const devtype_info *
type_atom(const sexpr &n)
{
	if (n.is_list)
		codegen_error("type name expected");
	const devtype_info *t = devtype_lookup_by_name(n.atom);
	if (!t)
		codegen_error("type \"%s\" is not supported on the device", n.atom.c_str());
	return t;
}

static int emit_expr(const sexpr &n, codegen_context &ctx, std::string &out);

/*
 * fixed-scale numerics (strom_numeric.h, pg_fixed_t): pseudo type ids that
 * exist only inside the emitter.  STROM_FIXED_BASE + s = int64 at 10^-s.
 */
bool	codegen_type_is_fixed(int t) { return t >= STROM_FIXED_BASE && t <= STROM_FIXED_BASE + 18; }
int		codegen_fixed_scale(int t) { return t - STROM_FIXED_BASE; }

static std::string
pow10_literal(int k)
{
	std::string s = "1";
	for (int i = 0; i < k; i++)
		s += "0";
	return s + "L";
}

/* fixed<scale> text -> pg_numeric_t text */
std::string
codegen_fixed_as_numeric(const std::string &text, int scale)
{
	/* a literal carries its kern_parambuf twin: pg_fixed_lit(VALUE, KP.KPARAM_n) */
	if (text.compare(0, 13, "pg_fixed_lit(") == 0 && text.back() == ')')
	{
		size_t comma = text.find(", ");
		if (comma != std::string::npos)
			return text.substr(comma + 2, text.size() - comma - 3);
	}
	return "pgfn_fixed_to_numeric(errcode, " + text + ", " + std::to_string(scale) + ")";
}

/* fixed<from> text -> fixed<to> text, to >= from */
std::string
codegen_fixed_rescale(const std::string &text, int from, int to)
{
	if (to == from)
		return text;
	return "pgfn_fixed_scaleup(errcode, " + text + ", " + pow10_literal(to - from) + ")";
}

/* like emit_expr, but a fixed-scale result is turned into a plain numeric */
static int
emit_expr_plain(const sexpr &n, codegen_context &ctx, std::string &out)
{
	std::string t;
	int		type = emit_expr(n, ctx, t);
	if (codegen_type_is_fixed(type))
	{
		out += codegen_fixed_as_numeric(t, codegen_fixed_scale(type));
		return STROM_NUMERICOID;
	}
	out += t;
	return type;
}

static void
emit_bool_chain(const sexpr &n, codegen_context &ctx, std::string &out, const char *fn)
{
	/* (and a b c) -> and2(and2(a,b),c) */
	size_t	nargs = n.items.size() - 1;
	if (nargs < 1)
		codegen_error("(%s) needs at least one argument", n.items[0].atom.c_str());
	std::string acc;
	for (size_t i = 1; i <= nargs; i++)
	{
		std::string arg;
		if (emit_expr(n.items[i], ctx, arg) != STROM_BOOLOID)
			codegen_error("argument %zu of %s is not bool", i, n.items[0].atom.c_str());
		if (i == 1)
			acc = arg;
		else
			acc = std::string(fn) + "(" + acc + ", " + arg + ")";
	}
	out += acc;
}

static int
emit_expr(const sexpr &n, codegen_context &ctx, std::string &out)
{
	if (!n.is_list || n.items.empty() || n.items[0].is_list)
		codegen_error("expression node must be a list starting with a symbol");
	const std::string &head = n.items[0].atom;
	size_t	nargs = n.items.size() - 1;
	char	tmp[64];

	if (head == "const")
	{
		if (nargs != 2 || n.items[2].is_list)
			codegen_error("(const TYPE LITERAL) expected");
		const devtype_info *t = type_atom(n.items[1]);
		strom_kparam_desc d;
		memset(&d, 0, sizeof(d));
		d.type_oid = t->type_oid;
		d.is_const = 1;
		d.param_id = -1;
		if (n.items[2].atom == "null" || n.items[2].atom == "NULL")
		{
			d.isnull = 1;
			d.length = (t->type_length > 0 ? t->type_length : 0);
		}
		else
			literal_to_datum(t, n.items[2].atom, &d);
		ctx.extra_flags |= (t->type_flags & ~DEVTYPE_IS_VARLENA);	/* (only a chunk COLUMN of that type binds the chunk format) */
		snprintf(tmp, sizeof(tmp), "KP.KPARAM_%d", ctx.track_param(d));
		if (t->type_oid == STROM_NUMERICOID && !d.isnull)
		{
			/* plain decimal literal: also usable as a fixed-scale value */
			const std::string &lit = n.items[2].atom;
			size_t	i = 0, ndigits = 0, ndec = 0;
			bool	neg = false, seen_dot = false, ok = !lit.empty();
			std::string digits;
			if (i < lit.size() && (lit[i] == '+' || lit[i] == '-'))
				neg = (lit[i++] == '-');
			for (; ok && i < lit.size(); i++)
			{
				if (lit[i] >= '0' && lit[i] <= '9')
				{
					digits += lit[i];
					ndigits++;
					if (seen_dot)
						ndec++;
				}
				else if (lit[i] == '.' && !seen_dot)
					seen_dot = true;
				else
					ok = false;
			}
			if (ok && ndigits >= 1 && ndigits <= 18 && ndec <= 18)
			{
				out += std::string("pg_fixed_lit(") + (neg ? "-" : "") + digits + "L, " + tmp + ")";
				return STROM_FIXED_BASE + (int)ndec;
			}
		}
		out += tmp;
		return t->type_oid;
	}
	if (head == "param")
	{
		if (nargs != 2 || n.items[1].is_list)
			codegen_error("(param INDEX TYPE) expected");
		const devtype_info *t = type_atom(n.items[2]);
		strom_kparam_desc d;
		memset(&d, 0, sizeof(d));
		d.type_oid = t->type_oid;
		d.is_const = 0;
		d.param_id = atoi(n.items[1].atom.c_str());
		d.length = t->type_length;
		if (d.param_id < 0)
			codegen_error("negative parameter number");
		ctx.extra_flags |= (t->type_flags & ~DEVTYPE_IS_VARLENA);	/* (only a chunk COLUMN of that type binds the chunk format) */
		snprintf(tmp, sizeof(tmp), "KP.KPARAM_%d", ctx.track_param(d));
		out += tmp;
		return t->type_oid;
	}
	if (head == "var")
	{
		if ((nargs != 2 && nargs != 3) || n.items[1].is_list)
			codegen_error("(var ATTNO TYPE [SCALE]) expected");
		const devtype_info *t = type_atom(n.items[2]);
		int		attno = atoi(n.items[1].atom.c_str());
		if (attno < 1)
			codegen_error("attribute numbers start at 1");
		ctx.track_var(attno, t->type_oid);
		ctx.extra_flags |= t->type_flags;
		snprintf(tmp, sizeof(tmp), "%s.%s_%d", ctx.var_struct.c_str(),
				 ctx.var_label.c_str(), attno);
		if (t->type_oid == STROM_DECIMALOID)
		{
			/* the column IS the fixed-point value: nothing to convert */
			int		scale = ((nargs == 3 && !n.items[3].is_list) ? atoi(n.items[3].atom.c_str()) : -1);
			if (scale < 0 || scale > 18)
				codegen_error("(var ATTNO decimal SCALE): the scale 0..18 the column is stored at expected");
			out += std::string("pg_fixed_from_decimal(") + tmp + ")";
			return STROM_FIXED_BASE + scale;
		}
		if (nargs == 3)
		{
			/* typmod scale of a numeric column: fixed-point from here on */
			int		scale = (n.items[3].is_list ? -1 : atoi(n.items[3].atom.c_str()));
			if (t->type_oid != STROM_NUMERICOID || scale < 0 || scale > 18)
				codegen_error("(var ATTNO numeric SCALE): scale 0..18 on a numeric column expected");
			if (ctx.fixed_cache)
			{
				bool	known = false;
				for (auto &f : ctx.used_fixed)
					known = known || (f.attno == attno && f.scale == scale);
				if (!known)
					ctx.used_fixed.push_back({attno, scale});
				out += std::string("pg_fixed_cached(errcode, ") + ctx.var_struct + ".KFIX_" +
					std::to_string(attno) + "_" + std::to_string(scale) + ")";
			}
			else
				out += std::string("pgfn_numeric_as_fixed(errcode, ") + tmp + ", " + std::to_string(scale) + ")";
			return STROM_FIXED_BASE + scale;
		}
		out += tmp;
		return t->type_oid;
	}
	if (head == "ivar")
	{
		if (nargs != 3 || n.items[1].is_list || n.items[2].is_list)
			codegen_error("(ivar DEPTH ATTNO TYPE) expected");
		const devtype_info *t = type_atom(n.items[3]);
		int		depth = atoi(n.items[1].atom.c_str());
		int		attno = atoi(n.items[2].atom.c_str());
		if (depth < 1 || depth > ctx.ivar_max_depth)
			codegen_error("(ivar %d ..) is not visible here", depth);
		if (attno < 1)
			codegen_error("attribute numbers start at 1");
		bool known = false;
		for (auto &iv : ctx.used_ivars)
			if (iv.depth == depth && iv.attno == attno)
			{
				if (iv.type_oid != t->type_oid)
					codegen_error("inner attribute referenced as two different types");
				known = true;
			}
		if (!known)
			ctx.used_ivars.push_back({depth, attno, t->type_oid});
		ctx.extra_flags |= (t->type_flags & ~DEVTYPE_IS_VARLENA);	/* (only a chunk COLUMN of that type binds the chunk format) */
		snprintf(tmp, sizeof(tmp), "IVAR_%d_%d", depth, attno);
		out += tmp;
		return t->type_oid;
	}
	if (head == "and" || head == "or")
	{
		emit_bool_chain(n, ctx, out, head == "and" ? "pgfn_boolop_and2" : "pgfn_boolop_or2");
		return STROM_BOOLOID;
	}
	if (head == "not")
	{
		if (nargs != 1)
			codegen_error("(not e) expected");
		out += "pgfn_boolop_not(errcode, ";
		if (emit_expr(n.items[1], ctx, out) != STROM_BOOLOID)
			codegen_error("argument of not is not bool");
		out += ")";
		return STROM_BOOLOID;
	}
	if (head == "isnull" || head == "isnotnull")
	{
		if (nargs != 1)
			codegen_error("(%s e) expected", head.c_str());
		std::string arg;
		int		t = emit_expr(n.items[1], ctx, arg);
		if (codegen_type_is_fixed(t))
			out += "pgfn_fixed_" + head + "(errcode, " + arg + ")";
		else
			out += "pgfn_" + std::string(devtype_lookup(t)->dev_name) + "_" + head +
				"(errcode, " + arg + ")";
		return STROM_BOOLOID;
	}
	if (head == "is_true" || head == "is_not_true" || head == "is_false" ||
		head == "is_not_false" || head == "is_unknown" || head == "is_not_unknown")
	{
		if (nargs != 1)
			codegen_error("(%s e) expected", head.c_str());
		out += "pgfn_bool_" + head + "(errcode, ";
		if (emit_expr(n.items[1], ctx, out) != STROM_BOOLOID)
			codegen_error("argument of BooleanTest is not bool");
		out += ")";
		return STROM_BOOLOID;
	}
	if (head == "relabel")
	{
		if (nargs != 2)
			codegen_error("(relabel TYPE e) expected");
		const devtype_info *t = type_atom(n.items[1]);
		int		at = emit_expr_plain(n.items[2], ctx, out);
		if (devtype_lookup(at)->type_length != t->type_length)
			codegen_error("relabel between types of different width");
		return at;		/* same binary form */
	}
	if (head == "case" || head == "case_eq")
	{
		/*
		 * EVAL(cond) ? (result) : (... default)   as codegen.c:1340-1389;
		 * every branch must have one type.
		 */
		size_t		first = 1;
		std::string	argtext;
		int			argtype = 0;
		if (head == "case_eq")
		{
			if (nargs < 2)
				codegen_error("(case_eq arg (when v r) ... (else d)) expected");
			argtype = emit_expr_plain(n.items[1], ctx, argtext);
			first = 2;
		}
		int			restype = 0;
		std::string	text;
		int			depth = 0;
		bool		has_else = false;
		for (size_t i = first; i < n.items.size(); i++)
		{
			const sexpr &c = n.items[i];
			if (!c.is_list || c.items.empty() || c.items[0].is_list)
				codegen_error("case arm must be (when ...) or (else ...)");
			if (c.items[0].atom == "when")
			{
				if (c.items.size() != 3 || has_else)
					codegen_error("(when cond result) expected before (else ...)");
				std::string cond, res;
				int ct = emit_expr_plain(c.items[1], ctx, cond);
				if (head == "case_eq")
				{
					const char *eq = devtype_eqfunc(argtype);
					const devfunc_info *f = eq ? devfunc_lookup(eq, {argtype, ct}) : nullptr;
					if (!f)
						codegen_error("no equality function for simple CASE");
					ctx.extra_flags |= f->flags;
					cond = "pgfn_" + f->devname + "(errcode, " + argtext + ", " + cond + ")";
				}
				else if (ct != STROM_BOOLOID)
					codegen_error("WHEN condition is not bool");
				int rt = emit_expr_plain(c.items[2], ctx, res);
				if (restype == 0)	restype = rt;
				else if (restype != rt)
					codegen_error("CASE branches have different types");
				text += "(EVAL(" + cond + ") ? (" + res + ") : ";
				depth++;
			}
			else if (c.items[0].atom == "else")
			{
				if (c.items.size() != 2)
					codegen_error("(else result) expected");
				std::string res;
				int rt = emit_expr_plain(c.items[1], ctx, res);
				if (restype == 0)	restype = rt;
				else if (restype != rt)
					codegen_error("CASE branches have different types");
				text += "(" + res + ")";
				has_else = true;
			}
			else
				codegen_error("case arm must be (when ...) or (else ...)");
		}
		if (restype == 0)
			codegen_error("empty CASE");
		if (!has_else)
		{
			/* SQL: missing ELSE means NULL */
			const devtype_info *rt = devtype_lookup(restype);
			text += "(pg_" + std::string(rt->dev_name) + "_make(0, true))";
		}
		for (int i = 0; i < depth; i++)
			text += ")";
		out += text;
		return restype;
	}

	/* FuncExpr / OpExpr */
	std::vector<int>			argtypes;
	std::vector<std::string>	argtexts;
	for (size_t i = 1; i <= nargs; i++)
	{
		std::string t;
		argtypes.push_back(emit_expr(n.items[i], ctx, t));
		argtexts.push_back(t);
	}
	/* fixed-scale numerics: scales are resolved here, at code-generation time */
	bool	any_fixed = false, all_fixed = (nargs > 0);
	for (int t : argtypes)
	{
		any_fixed = any_fixed || codegen_type_is_fixed(t);
		all_fixed = all_fixed && codegen_type_is_fixed(t);
	}
	if (all_fixed)
	{
		ctx.extra_flags |= DEVFUNC_NEEDS_NUMERIC;
		if (nargs == 2 && (head == "numeric_add" || head == "numeric_sub"))
		{
			int s1 = codegen_fixed_scale(argtypes[0]), s2 = codegen_fixed_scale(argtypes[1]);
			int sc = std::max(s1, s2);
			out += "pgfn_fixed_" + head.substr(8) + "(errcode, " +
				codegen_fixed_rescale(argtexts[0], s1, sc) + ", " +
				codegen_fixed_rescale(argtexts[1], s2, sc) + ")";
			return STROM_FIXED_BASE + sc;
		}
		if (nargs == 2 && head == "numeric_mul" &&
			codegen_fixed_scale(argtypes[0]) + codegen_fixed_scale(argtypes[1]) <= 18)
		{
			out += "pgfn_fixed_mul(errcode, " + argtexts[0] + ", " + argtexts[1] + ")";
			return STROM_FIXED_BASE + codegen_fixed_scale(argtypes[0]) + codegen_fixed_scale(argtypes[1]);
		}
		if (nargs == 2 && (head == "numeric_eq" || head == "numeric_ne" || head == "numeric_lt" ||
						   head == "numeric_le" || head == "numeric_gt" || head == "numeric_ge"))
		{
			int s1 = codegen_fixed_scale(argtypes[0]), s2 = codegen_fixed_scale(argtypes[1]);
			int sc = std::max(s1, s2);
			out += "pgfn_fixed_" + head.substr(8) + "(errcode, " +
				codegen_fixed_rescale(argtexts[0], s1, sc) + ", " +
				codegen_fixed_rescale(argtexts[1], s2, sc) + ")";
			return STROM_BOOLOID;
		}
		if (nargs == 1 && (head == "numeric_uminus" || head == "numeric_abs" || head == "numeric_uplus"))
		{
			if (head == "numeric_uplus")
				out += argtexts[0];
			else
				out += "pgfn_fixed_" + head.substr(8) + "(errcode, " + argtexts[0] + ")";
			return argtypes[0];
		}
	}
	if (any_fixed)
	{
		/* mixed with a scale-less value or not a fixed-point operation:
		 * back to the 64-bit numeric form */
		for (size_t i = 0; i < argtypes.size(); i++)
			if (codegen_type_is_fixed(argtypes[i]))
			{
				argtexts[i] = codegen_fixed_as_numeric(argtexts[i], codegen_fixed_scale(argtypes[i]));
				argtypes[i] = STROM_NUMERICOID;
			}
	}
	const devfunc_info *f = devfunc_lookup(head, argtypes);
	if (!f)
	{
		std::string sig;
		for (size_t i = 0; i < argtypes.size(); i++)
			sig += std::string(i ? "," : "") + devtype_lookup(argtypes[i])->sql_name;
		codegen_error("function %s(%s) is not supported on the device", head.c_str(), sig.c_str());
	}
	ctx.extra_flags |= f->flags;
	out += "pgfn_" + f->devname + "(errcode";
	for (auto &t : argtexts)
		out += ", " + t;
	out += ")";
	return f->rettype;
}

int
codegen_expression(const sexpr &n, codegen_context &ctx, std::string &out)
{
	return emit_expr_plain(n, ctx, out);
}

/* may return a fixed-scale pseudo type (codegen_type_is_fixed) */
int
codegen_expression_raw(const sexpr &n, codegen_context &ctx, std::string &out)
{
	return emit_expr(n, ctx, out);
}

std::string
devfunc_devname(const std::string &name, const std::vector<int> &argtypes,
				int *rettype, int *flags)
{
	const devfunc_info *f = devfunc_lookup(name, argtypes);
	if (!f)
		return std::string();
	if (rettype)	*rettype = f->rettype;
	if (flags)		*flags = f->flags;
	return f->devname;
}

/* ------------------------------------------------------------------ *
 * declaration emitters
 * ------------------------------------------------------------------ */
std::string
codegen_param_list(const codegen_context &ctx)
{
	std::string s = "#define STROM_KPARAM_LIST(X)";
	char	tmp[96];
	for (size_t i = 0; i < ctx.used_params.size(); i++)
	{
		snprintf(tmp, sizeof(tmp), " X(%zu,%s)", i,
				 devtype_lookup(ctx.used_params[i].type_oid)->dev_name);
		s += tmp;
	}
	return s + "\n";
}

std::string
codegen_var_list(const codegen_context &ctx, const char *macro_name)
{
	std::string s = std::string("#define ") + macro_name + "(X)";
	char	tmp[96];
	for (auto &v : ctx.used_vars)
	{
		snprintf(tmp, sizeof(tmp), " X(%d,%d,%s)", v.attno, v.attno - 1,
				 devtype_lookup(v.type_oid)->dev_name);
		s += tmp;
	}
	s += "\n";
	/* the text / character(n) variables: what strom_kvars_from_column (strom_common.h) turns from
	 * offset to address when the row came from a COLUMN chunk */
	std::string vl;
	for (auto &v : ctx.used_vars)
		if (devtype_lookup(v.type_oid)->type_flags & DEVTYPE_IS_VARLENA)
		{
			snprintf(tmp, sizeof(tmp), " X(%d,%s)", v.attno, devtype_lookup(v.type_oid)->dev_name);
			vl += tmp;
		}
	if (!vl.empty() && !strcmp(macro_name, "STROM_KVAR_LIST"))
		s += "#define STROM_KVARLENA_LIST(X)" + vl + "\n";
	return s;
}

std::string
codegen_includes(int extra_flags)
{
	std::string s = "#include \"strom_kds.h\"\n#include \"strom_common.h\"\n";
	if (extra_flags & (DEVFUNC_NEEDS_MATHLIB | DEVFUNC_NEEDS_NUMERIC))
		s += "#include \"strom_mathlib.h\"\n";
	if (extra_flags & DEVFUNC_NEEDS_TIMELIB)	s += "#include \"strom_timelib.h\"\n";
	if (extra_flags & DEVFUNC_NEEDS_NUMERIC)	s += "#include \"strom_numeric.h\"\n";
	if (extra_flags & DEVFUNC_NEEDS_TEXTLIB)	s += "#include \"strom_textlib.h\"\n";
	return s;
}

void
codegen_fill_result(const codegen_context &ctx, const std::string &source,
					strom_codegen_result *out)
{
	out->source = strdup(source.c_str());
	out->extra_flags = ctx.extra_flags;
	out->nparams = (int)ctx.used_params.size();
	out->params = (strom_kparam_desc *)calloc(out->nparams + 1, sizeof(strom_kparam_desc));
	for (int i = 0; i < out->nparams; i++)
		out->params[i] = ctx.used_params[i];
	out->nvars = (int)ctx.used_vars.size();
	out->vars = (strom_kvar_desc *)calloc(out->nvars + 1, sizeof(strom_kvar_desc));
	for (int i = 0; i < out->nvars; i++)
		out->vars[i] = ctx.used_vars[i];
	out->errmsg = nullptr;
}

}	/* namespace strom */

using namespace strom;

extern "C" int
strom_codegen_gpuscan(const char *qual, strom_codegen_result *out)
{
	memset(out, 0, sizeof(*out));
	try {
		codegen_context ctx;
		ctx.var_label = "KVAR";
		ctx.var_struct = "KV";
		ctx.extra_flags = DEVKERNEL_NEEDS_GPUSCAN | DEVFUNC_NEEDS_MATHLIB;
		sexpr tree = sexpr_parse(qual);
		std::string body;
		if (codegen_expression(tree, ctx, body) != STROM_BOOLOID)
			codegen_error("GpuScan qualifier is not a boolean expression");
		std::string src = "/* generated by strom_codegen_gpuscan */\n";
		src += codegen_includes(ctx.extra_flags);
		src += codegen_param_list(ctx);
		src += codegen_var_list(ctx, "STROM_KVAR_LIST");
		src += "#include \"strom_gpuscan.h\"\n";
		src += "STROM_DEVICE pg_bool_t\n"
			"gpuscan_qual_eval(cl_int *errcode,\n"
			"                  const strom_kparams &KP,\n"
			"                  const strom_kvars &KV)\n"
			"{\n"
			"  return " + body + ";\n"
			"}\n";
		codegen_fill_result(ctx, src, out);
		return 0;
	} catch (const std::exception &e) {
		out->errmsg = strdup(e.what());
		return -1;
	}
}

extern "C" int
strom_codegen_available_expression(const char *expr, char **errmsg)
{
	try {
		codegen_context ctx;
		ctx.var_label = "KVAR";
		ctx.var_struct = "KV";
		sexpr tree = sexpr_parse(expr);
		std::string body;
		codegen_expression(tree, ctx, body);
		if (errmsg)
			*errmsg = nullptr;
		return 1;
	} catch (const std::exception &e) {
		if (errmsg)
			*errmsg = strdup(e.what());
		return 0;
	}
}

extern "C" void
strom_codegen_release(strom_codegen_result *res)
{
	free(res->source);
	for (int i = 0; res->params && i < res->nparams; i++)
	{
		const strom_kparam_desc *d = &res->params[i];
		if ((d->type_oid == STROM_TEXTOID || d->type_oid == STROM_BPCHARNOID) && d->is_const && !d->isnull)
		{
			char *img;
			memcpy(&img, d->value, sizeof(img));
			free(img);
		}
	}
	free(res->params);
	free(res->vars);
	free(res->errmsg);
	memset(res, 0, sizeof(*res));
}

/* bytes of a varlena datum a parameter points to (VARSIZE_ANY) */
static size_t
varlena_size_any(const void *datum)
{
	const uint8_t *p = (const uint8_t *)datum;
	if (p[0] == 0x01)
		return 2 + (p[1] == 18 ? 16 : 8);		/* external TOAST pointer: the device re-checks */
	if (p[0] & 0x01)
		return (p[0] >> 1) & 0x7f;
	uint32_t w;
	memcpy(&w, p, 4);
	return (w >> 2) & 0x3fffffff;
}

extern "C" kern_parambuf *
strom_create_kern_parambuf(const strom_codegen_result *res,
						   const uint64_t *ext_values,
						   const uint8_t *ext_isnull,
						   int n_ext)
{
	int		nparams = res->nparams;
	size_t	offset = STROMALIGN(offsetof(kern_parambuf, poffset) + sizeof(cl_uint) * nparams);
	std::vector<const void *> srcs(nparams, nullptr);
	std::vector<size_t>	lens(nparams, 0);
	size_t	length = offset + STROMALIGN_LEN;

	/*
	 * by-value parameters: the image itself (d->value / ext_values[id]);
	 * text / character(n): the varlena datum the image POINTS to (a constant's
	 * lives in the codegen result, an external one is the caller's; any header
	 * form, copied verbatim like datastore.c:100-127)
	 */
	for (int i = 0; i < nparams; i++)
	{
		const strom_kparam_desc *d = &res->params[i];
		bool	varlena = (d->type_oid == STROM_TEXTOID || d->type_oid == STROM_BPCHARNOID);

		if (d->is_const)
		{
			if (!d->isnull)
				srcs[i] = d->value;
		}
		else if (d->param_id < n_ext && !(ext_isnull && ext_isnull[d->param_id]))
			srcs[i] = &ext_values[d->param_id];
		lens[i] = (size_t)(d->length > 0 ? d->length : 0);
		if (varlena && srcs[i])
		{
			const void *datum;
			memcpy(&datum, srcs[i], sizeof(datum));
			srcs[i] = datum;
			lens[i] = (datum ? varlena_size_any(datum) : 0);
		}
		if (srcs[i])
			length += STROMALIGN(lens[i]) + STROMALIGN_LEN;
	}
	char   *buf = (char *)calloc(1, length);
	kern_parambuf *kpbuf = (kern_parambuf *)buf;

	if (!buf)
		return nullptr;
	for (int i = 0; i < nparams; i++)
	{
		if (!srcs[i])
			kpbuf->poffset[i] = 0;		/* NULL */
		else
		{
			kpbuf->poffset[i] = (cl_uint)offset;
			memcpy(buf + offset, srcs[i], lens[i]);
			offset = STROMALIGN(offset + lens[i]);
		}
	}
	kpbuf->length = (cl_uint)offset;
	kpbuf->nparams = nparams;
	return kpbuf;
}
