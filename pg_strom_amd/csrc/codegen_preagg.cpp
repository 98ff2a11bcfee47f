/*
 * codegen_preagg.cpp -- GpuPreAgg program text
 *
 * Role in the reference: gpupreagg_codegen and friends (gpupreagg.c:
 * 1180-1943).  The reference emits qual_eval, keycomp (for its bitonic
 * sort), aggcalc (a switch over GPUPREAGG_AGGCALC_* macros) and a
 * projection that stores partial inputs into a TUPSLOT scratch store.
 * This build reduces by group id in LDS instead of sorting, so the
 * generated part is smaller: one function per target that yields the
 * per-row partial input in registers, plus X-macro catalogues of keys and
 * aggregates that strom_gpupreagg.h expands into typed LDS accumulators.
 */
#include <cstring>
#include <cstdio>
#include <string>
#include <vector>

#include "strom_codegen.h"
#include "strom_hip.h"
#include "codegen_internal.h"

using namespace strom;

namespace {

struct target {
	int			pack_kind = 0;	/* packed accumulators (strom_gpupreagg.h): 1 count(*), 2 psum of a plain
								 * integer column, 3 psum of a plain float8 column, 0 none of these */
	int			pack_attno = 0;
	/*
	 * integer sums (strom_gpupreagg.h, "integer sums never wrap"): 0 = not one; 1..63 = every
	 * input is below 2^sumbits in magnitude whatever the data (a sum over int2 / int4 values
	 * cast to int8); 64 = no static bound, the kernels measure the inputs
	 */
	int			sumbits = 0;
	/* sumbits 66: how large the summed expression can get, as a function of its columns' zone maps
	 * (sum_bound_formula below); evaluated by the host per chunk (gpupreagg.cpp: eval_sum_bound) */
	std::string	sumbound;
	int			key_attno = 0;	/* a group key that is a plain column (var N T): N, else 0 */
	bool		countall = false;	/* nrows() whose arguments only say "column X is not NULL": in a chunk
									 * without NULL bitmaps it counts what count(*) counts */
	int			kind;
	int			type_oid;	/* type of the partial value as the caller sees it */
	int			acc_oid;	/* type of the device accumulator */
	int			scale;		/* numeric partials: fixed point at 10^-scale, else -1 */
	std::string	body;		/* function body text */
};

/*
 * (var N T), or a widening cast of one -- (int8 (var N int2|int4|int8)),
 * (float8 (var N float4|float8)): the value is the column's, it is NULL only
 * where the column is, and it cannot raise an error.  Returns the attno or 0.
 */
/*
 * (int8 X) with X of type int2 / int4 (or narrower): |value| <= 2^15 / 2^31, so
 * its magnitude (v >= 0 ? v : -v - 1) is below 2^15 / 2^31 -- returns the bits, or 64
 */
int
static_sum_bits(const sexpr &e, const codegen_context &ctx)
{
	if (!e.is_list || e.items.size() != 2 || e.items[0].is_list || e.items[0].atom != "int8")
		return 64;
	codegen_context scratch = ctx;			/* (the argument is generated again for real by the caller) */
	std::string		text;
	int		inner = 0;
	try {
		inner = codegen_expression(e.items[1], scratch, text);
	} catch (...) {
		return 64;
	}
	if (inner == STROM_INT4OID || inner == STROM_DATEOID)
		return 31;
	if (inner == STROM_INT2OID)
		return 15;
	if (inner == STROM_BOOLOID)
		return 1;
	return 64;
}

int
plain_column_attno(const sexpr &e, int *p_var_oid)
{
	const sexpr *v = &e;
	std::string cast;
	if (e.is_list && e.items.size() == 2 && !e.items[0].is_list &&
		(e.items[0].atom == "int8" || e.items[0].atom == "float8") && e.items[1].is_list)
	{
		cast = e.items[0].atom;
		v = &e.items[1];
	}
	if (!v->is_list || v->items.size() != 3 || v->items[0].is_list || v->items[0].atom != "var" ||
		v->items[1].is_list || v->items[2].is_list)
		return 0;
	const std::string &t = v->items[2].atom;
	bool	is_int = (t == "int2" || t == "int4" || t == "int8");
	bool	is_flt = (t == "float4" || t == "float8");
	if ((cast == "int8" && !is_int) || (cast == "float8" && !is_flt) || (cast.empty() && !(t == "int8" || t == "float8")))
		return 0;
	int		attno = atoi(v->items[1].atom.c_str());
	*p_var_oid = (t == "int2" ? STROM_INT2OID : t == "int4" ? STROM_INT4OID : t == "int8" ? STROM_INT8OID :
				  t == "float4" ? STROM_FLOAT4OID : STROM_FLOAT8OID);
	return attno >= 1 ? attno : 0;
}

std::string
fn_header(const char *rettype, const char *name, int idx)
{
	char buf[256];
	snprintf(buf, sizeof(buf),
			 "STROM_DEVICE pg_%s_t\n%s_%d(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV)\n",
			 rettype, name, idx);
	return buf;
}

}	/* namespace */

/*
 * An upper bound of |EXPR| for a fixed-point expression over decimal columns, as a formula the host
 * evaluates with the chunk's zone maps (so that the fold need not measure every row's magnitude --
 * 7 VALU instructions per row and sum in Q1's kernels): reverse Polish over magnitudes,
 *   cN   largest |value| of column N's stored integers (zone map)
 *   nN   the same for a column of 64-bit numeric images: its integer-part bounds (KDS_COLSTAT_INTPART)
 *   kV   the constant V
 *   +    |a +- b| <= |a| + |b|        *    |a b| <= |a| |b|        eK   times 10^K (a rescale)
 * Returns the expression's scale, or -1 when something in it has no such bound (a parameter, a
 * function, a column that is not a decimal one): the sum is measured row by row then, as before.
 */
static int
sum_bound_formula(const sexpr &x, std::string &rpn)
{
	if (!x.is_list || x.items.empty() || x.items[0].is_list)
		return -1;
	const std::string &head = x.items[0].atom;
	char	tmp[64];
	if (head == "var" && x.items.size() == 4 && !x.items[1].is_list && !x.items[2].is_list &&
		x.items[2].atom == "decimal" && !x.items[3].is_list)
	{
		int attno = atoi(x.items[1].atom.c_str()), scale = atoi(x.items[3].atom.c_str());
		if (attno < 1 || scale < 0 || scale > 18)
			return -1;
		snprintf(tmp, sizeof(tmp), " c%d", attno);
		rpn += tmp;
		return scale;
	}
	if (head == "var" && x.items.size() == 4 && !x.items[1].is_list && !x.items[2].is_list &&
		x.items[2].atom == "numeric" && !x.items[3].is_list)
	{
		/* a 64-bit numeric image read as fixed point at its typmod scale: its zone map bounds the
		 * value's integer part, outward (KDS_COLSTAT_INTPART) */
		int attno = atoi(x.items[1].atom.c_str()), scale = atoi(x.items[3].atom.c_str());
		if (attno < 1 || scale < 0 || scale > 18)
			return -1;
		snprintf(tmp, sizeof(tmp), " n%d", attno);
		rpn += tmp;
		if (scale > 0)
		{
			snprintf(tmp, sizeof(tmp), " e%d", scale);
			rpn += tmp;
		}
		return scale;
	}
	if (head == "const" && x.items.size() == 3 && !x.items[1].is_list && x.items[1].atom == "numeric" &&
		!x.items[2].is_list)
	{
		/* a plain decimal literal: digits without the point, scale = digits behind it */
		const std::string &lit = x.items[2].atom;
		std::string digits;
		int		scale = 0;
		bool	point = false;
		for (size_t i = 0; i < lit.size(); i++)
		{
			char c = lit[i];
			if ((c == '-' || c == '+') && i == 0)
				continue;
			if (c == '.' && !point)
				point = true;
			else if (c >= '0' && c <= '9')
			{
				digits += c;
				scale += (point ? 1 : 0);
			}
			else
				return -1;
		}
		if (digits.empty() || digits.size() > 18 || scale > 18)
			return -1;
		rpn += " k" + digits;
		return scale;
	}
	if ((head == "numeric_add" || head == "numeric_sub") && x.items.size() == 3)
	{
		std::string a, b;
		int		sa = sum_bound_formula(x.items[1], a), sb = sum_bound_formula(x.items[2], b);
		if (sa < 0 || sb < 0)
			return -1;
		int		sc = (sa > sb ? sa : sb);
		rpn += a;
		if (sc > sa) { snprintf(tmp, sizeof(tmp), " e%d", sc - sa); rpn += tmp; }
		rpn += b;
		if (sc > sb) { snprintf(tmp, sizeof(tmp), " e%d", sc - sb); rpn += tmp; }
		rpn += " +";
		return sc;
	}
	if (head == "numeric_mul" && x.items.size() == 3)
	{
		std::string a, b;
		int		sa = sum_bound_formula(x.items[1], a), sb = sum_bound_formula(x.items[2], b);
		if (sa < 0 || sb < 0 || sa + sb > 18)
			return -1;
		rpn += a + b + " *";
		return sa + sb;
	}
	if ((head == "numeric_uminus" || head == "numeric_uplus" || head == "numeric_abs") && x.items.size() == 2)
		return sum_bound_formula(x.items[1], rpn);
	return -1;
}

extern "C" int
strom_codegen_gpupreagg(const char *spec, strom_codegen_result *out,
						strom_preagg_target *targets_out, int max_targets, int *p_ntargets)
{
	memset(out, 0, sizeof(*out));
	try {
		codegen_context ctx;
		ctx.var_label = "KVAR";
		ctx.var_struct = "KV";
		ctx.extra_flags = DEVKERNEL_NEEDS_GPUPREAGG | DEVFUNC_NEEDS_MATHLIB;
		ctx.fixed_cache = true;				/* (var N numeric SCALE): converted once per row */
		sexpr tree = sexpr_parse(spec);
		if (!tree.is_list || tree.items.empty() || tree.items[0].is_list ||
			tree.items[0].atom != "gpupreagg")
			codegen_error("(gpupreagg ...) expected");
		std::string qual_body = "  pg_bool_t r; r.isnull = false; r.value = true; return r;\n";
		std::vector<target> targets;
		bool	numeric_aggs = false;		/* a partial accumulated in the 64-bit numeric form */
		for (size_t i = 1; i < tree.items.size(); i++)
		{
			const sexpr &t = tree.items[i];
			if (!t.is_list || t.items.empty() || t.items[0].is_list)
				codegen_error("target must be a list");
			const std::string &head = t.items[0].atom;
			size_t	nargs = t.items.size() - 1;
			if (head == "qual")
			{
				if (nargs != 1)
					codegen_error("(qual EXPR) expected");
				std::string e;
				if (codegen_expression(t.items[1], ctx, e) != STROM_BOOLOID)
					codegen_error("qual is not boolean");
				qual_body = "  return " + e + ";\n";
				continue;
			}
			target tg;
			tg.scale = -1;
			tg.acc_oid = 0;
			if (head == "key")
			{
				if (nargs != 1)
					codegen_error("(key EXPR) expected");
				std::string e;
				tg.kind = STROM_PREAGG_KEY;
				tg.type_oid = codegen_expression(t.items[1], ctx, e);
				{
					const sexpr &x = t.items[1];
					if (x.is_list && x.items.size() == 3 && !x.items[0].is_list && x.items[0].atom == "var" &&
						!x.items[1].is_list)
						tg.key_attno = atoi(x.items[1].atom.c_str());
				}
				/* a numeric key groups by its canonical image (hashed GROUP BY) */
				if (tg.type_oid == STROM_NUMERICOID)
					e = "pgfn_numeric_normalize(errcode, " + e + ")";
				tg.body = "  return " + e + ";\n";
			}
			else if (head == "nrows")
			{
				tg.kind = STROM_PREAGG_NROWS;
				tg.type_oid = STROM_INT4OID;
				std::string cond;
				tg.countall = true;
				for (size_t a = 1; a <= nargs; a++)
				{
					std::string e;
					const sexpr &x = t.items[a];
					/* (isnotnull (var N ...)) and nothing else? */
					if (!(x.is_list && x.items.size() == 2 && !x.items[0].is_list && x.items[0].atom == "isnotnull" &&
						  x.items[1].is_list && !x.items[1].items.empty() && !x.items[1].items[0].is_list &&
						  x.items[1].items[0].atom == "var"))
						tg.countall = false;
					if (codegen_expression(t.items[a], ctx, e) != STROM_BOOLOID)
						codegen_error("nrows() argument is not boolean");
					/* every argument is evaluated: no short circuit, so a
					 * CpuReCheck raised by a later argument is not lost */
					cond += "  ok &= EVAL(" + e + ");\n";
				}
				tg.body = "  pg_int4_t r; bool ok = true;\n" + cond +
					"  r.isnull = false; r.value = ok ? 1 : 0; return r;\n";
				if (nargs == 0)
					tg.pack_kind = 1;
			}
			else if (head == "psum" || head == "pmin" || head == "pmax")
			{
				if (nargs != 1 && nargs != 2)
					codegen_error("(%s EXPR [SCALE]) expected", head.c_str());
				std::string e;
				tg.kind = (head == "psum" ? STROM_PREAGG_PSUM :
						   head == "pmin" ? STROM_PREAGG_PMIN : STROM_PREAGG_PMAX);
				int		raw_type = codegen_expression_raw(t.items[1], ctx, e);
				int		fixed_scale = -1;
				if (codegen_type_is_fixed(raw_type))
				{
					/* typmod-scaled numerics stay int64 all the way into the sum */
					fixed_scale = codegen_fixed_scale(raw_type);
					int want = (nargs == 2 && !t.items[2].is_list) ? atoi(t.items[2].atom.c_str()) : -1;
					if (want < fixed_scale || want > 18)
					{
						e = codegen_fixed_as_numeric(e, fixed_scale);
						fixed_scale = -1;
					}
					raw_type = STROM_NUMERICOID;
				}
				tg.type_oid = raw_type;
				if (tg.kind == STROM_PREAGG_PSUM &&
					!(tg.type_oid == STROM_INT8OID || tg.type_oid == STROM_FLOAT8OID ||
					  tg.type_oid == STROM_FLOAT4OID || tg.type_oid == STROM_NUMERICOID))
					codegen_error("psum() takes int8, float4, float8 or numeric (cast the argument)");
				if (tg.type_oid == STROM_NUMERICOID)
				{
					/*
					 * numeric partials are folded as fixed-point int8 at
					 * 10^-SCALE (integer LDS atomics; exact or CpuReCheck).
					 * The reference adds 64-bit numerics pairwise in its
					 * reduction tree and re-checks on overflow
					 * (opencl_gpupreagg.h:933-948).
					 */
					if (nargs == 1)
					{
						/*
						 * no scale: the partial stays a 64-bit numeric and is folded with
						 * pgfn_numeric_add / strom_numeric_cmp in a compare-and-swap loop,
						 * exact or CpuReCheck -- GPUPREAGG_AGGCALC_PSUM_NUMERIC /
						 * _PMINMAX_NUMERIC of the reference (opencl_gpupreagg.h:882-900,
						 * 965-987).  Stored normalised.
						 */
						if (fixed_scale >= 0)
							e = codegen_fixed_as_numeric(e, fixed_scale);
						tg.body = "  return pgfn_numeric_normalize(errcode, " + e + ");\n";
						numeric_aggs = true;
						goto target_done;
					}
					if (t.items[2].is_list)
						codegen_error("numeric partial needs a scale: (%s EXPR SCALE)", head.c_str());
					tg.scale = atoi(t.items[2].atom.c_str());
					if (tg.scale < 0 || tg.scale > 32)
						codegen_error("numeric scale out of range");
					tg.acc_oid = STROM_INT8OID;
					if (tg.kind == STROM_PREAGG_PSUM)
						tg.sumbits = 64;			/* fixed point in an int8 accumulator */
					char sb[16];
					snprintf(sb, sizeof(sb), "%d", tg.scale);
					if (fixed_scale >= 0)
					{
						tg.body = "  return pgfn_fixed_to_int8(errcode, " +
							codegen_fixed_rescale(e, fixed_scale, tg.scale) + ");\n";
						/*
						 * sum of a DECIMAL column at the column's own scale: the accumulator adds
						 * the stored integers as they are, and the column's zone map bounds them
						 * -- a packed-accumulator input like an int8 column (count and bounded
						 * sums in one LDS word, strom_gpupreagg.h)
						 */
						const sexpr &x = t.items[1];
						if (tg.kind == STROM_PREAGG_PSUM && fixed_scale == tg.scale &&
							x.is_list && x.items.size() == 4 && !x.items[0].is_list && x.items[0].atom == "var" &&
							!x.items[1].is_list && !x.items[2].is_list && x.items[2].atom == "decimal")
						{
							int attno = atoi(x.items[1].atom.c_str());
							if (attno >= 1)
							{
								tg.pack_kind = 2;
								tg.pack_attno = attno;
								tg.sumbits = 65;		/* a plain column: its zone map bounds the inputs */
							}
						}
						else if (tg.kind == STROM_PREAGG_PSUM)
						{
							/* an expression over decimal columns: bounded from their zone maps */
							std::string rpn;
							int		bs = sum_bound_formula(x, rpn);
							if (bs >= 0 && bs <= tg.scale)
							{
								if (tg.scale > bs)
									rpn += " e" + std::to_string(tg.scale - bs);
								tg.sumbound = rpn;
								tg.sumbits = 66;
							}
						}
					}
					else
						tg.body = "  return strom_numeric_to_fixed(errcode, " + e + ", " + sb + ");\n";
				}
				else
				{
					if (nargs != 1)
						codegen_error("only numeric partials take a scale");
					tg.body = "  return " + e + ";\n";
					if (tg.kind == STROM_PREAGG_PSUM && tg.type_oid == STROM_INT8OID)
						tg.sumbits = static_sum_bits(t.items[1], ctx);
					int		var_oid = 0;
					int		attno = plain_column_attno(t.items[1], &var_oid);
					if (tg.kind == STROM_PREAGG_PSUM && attno > 0 &&
						(tg.type_oid == STROM_INT8OID || tg.type_oid == STROM_FLOAT8OID))
					{
						tg.pack_kind = (tg.type_oid == STROM_INT8OID ? 2 : 3);
						tg.pack_attno = attno;
						if (tg.sumbits == 64 && tg.pack_kind == 2)
							tg.sumbits = 65;			/* a plain int8 column: its zone map bounds the inputs */
					}
				}
			}
			else if (head == "psum_x2")
			{
				if (nargs != 1)
					codegen_error("(psum_x2 EXPR) expected");
				std::string e;
				tg.kind = STROM_PREAGG_PSUM;
				int		xtype = codegen_expression(t.items[1], ctx, e);
				if (xtype == STROM_NUMERICOID)
				{
					/* stddev / variance over numeric: sum of squares in the 64-bit form
					 * (gpupreagg.c:218-313 ALTFUNC_EXPR_PSUM_X2 with NUMERICOID) */
					tg.type_oid = STROM_NUMERICOID;
					tg.body = "  pg_numeric_t x = " + e + ";\n"
						"  return pgfn_numeric_normalize(errcode, pgfn_numeric_mul(errcode, x, x));\n";
					numeric_aggs = true;
				}
				else
				{
					if (xtype != STROM_FLOAT8OID)
						codegen_error("psum_x2() takes float8 or numeric");
					tg.type_oid = STROM_FLOAT8OID;
					tg.body = "  pg_float8_t x = " + e + ";\n  return pgfn_float8mul(errcode, x, x);\n";
				}
			}
			else if (head == "pcov_x" || head == "pcov_y" || head == "pcov_x2" ||
					 head == "pcov_y2" || head == "pcov_xy")
			{
				if (nargs != 3)
					codegen_error("(%s FILTER X Y) expected", head.c_str());
				std::string f, x, y;
				if (codegen_expression(t.items[1], ctx, f) != STROM_BOOLOID ||
					codegen_expression(t.items[2], ctx, x) != STROM_FLOAT8OID ||
					codegen_expression(t.items[3], ctx, y) != STROM_FLOAT8OID)
					codegen_error("%s(bool, float8, float8) expected", head.c_str());
				tg.kind = STROM_PREAGG_PSUM;
				tg.type_oid = STROM_FLOAT8OID;
				std::string val = (head == "pcov_x" ? "x" : head == "pcov_y" ? "y" :
								   head == "pcov_x2" ? "pgfn_float8mul(errcode, x, x)" :
								   head == "pcov_y2" ? "pgfn_float8mul(errcode, y, y)" :
								   "pgfn_float8mul(errcode, x, y)");
				tg.body = "  pg_bool_t f = " + f + ";\n  pg_float8_t x = " + x +
					";\n  pg_float8_t y = " + y + ";\n"
					"  if (!EVAL(f) || x.isnull || y.isnull) return pg_float8_make(0.0, true);\n"
					"  return " + val + ";\n";
			}
			else
				codegen_error("unknown GpuPreAgg target \"%s\"", head.c_str());
		target_done:
			const devtype_info *dt = devtype_lookup(tg.type_oid);
			/* a partial row carries by-value datums only (TUPSLOT, 8 bytes per column):
			 * text / character(n) may be compared in quals and arguments, not grouped or
			 * aggregated (the reference's keycomp / aggcalc catalogues have no varlena
			 * entry either, gpupreagg.c:1181-1440) */
			if (dt->type_flags & DEVTYPE_IS_VARLENA)
				codegen_error("GpuPreAgg target of type %s: group keys and partial aggregates "
							  "are fixed-width", dt->sql_name);
			ctx.extra_flags |= dt->type_flags;
			if (tg.acc_oid == 0)
				tg.acc_oid = tg.type_oid;
			targets.push_back(tg);
		}
		if (targets.empty())
			codegen_error("GpuPreAgg needs at least one target");
		if ((int)targets.size() > max_targets)
			codegen_error("too many targets");

		std::string key_list = "#define GPUPREAGG_KEY_LIST(X)";
		std::string agg_list = "#define GPUPREAGG_AGG_LIST(X)";
		std::string pack_list = "#define GPUPREAGG_PACK_LIST(X)";
		std::string sumbits_defs;
		int		countall_first = -1;
		/* group keys that are plain columns: X(kidx, attno) -- their zone maps bound the dense ids
		 * of a COLUMN chunk without a pass over it (gpupreagg.cpp: chunk_domain) */
		std::string keycols = "#define GPUPREAGG_KEYCOLS_LIST(X)";
		bool	packable = true;
		std::string funcs;
		int		nkeys = 0, naggs = 0;
		char	tmp[160];
		for (size_t i = 0; i < targets.size(); i++)
		{
			const target &tg = targets[i];
			const char *tname = devtype_lookup(tg.acc_oid)->dev_name;
			if (tg.kind == STROM_PREAGG_KEY)
			{
				snprintf(tmp, sizeof(tmp), " X(%d,%zu,%s)", nkeys, i, tname);
				key_list += tmp;
				snprintf(tmp, sizeof(tmp), " X(%d,%d)", nkeys, tg.key_attno);
				keycols += tmp;
				funcs += fn_header(tname, "gpupreagg_key", nkeys) + "{\n" + tg.body + "}\n";
				nkeys++;
			}
			else
			{
				const char *op = (tg.kind == STROM_PREAGG_NROWS ? "NROWS" :
								  tg.kind == STROM_PREAGG_PSUM ? "PSUM" :
								  tg.kind == STROM_PREAGG_PMIN ? "PMIN" : "PMAX");
				snprintf(tmp, sizeof(tmp), " X(%d,%zu,%s,%s)", naggs, i, op, tname);
				agg_list += tmp;
				snprintf(tmp, sizeof(tmp), " X(%d,%d,%d)", naggs, tg.pack_kind, tg.pack_attno);
				pack_list += tmp;
				packable = packable && (tg.pack_kind != 0);
				snprintf(tmp, sizeof(tmp), "#define GPUPREAGG_SUMBITS_%d %d\n", naggs, tg.sumbits);
				sumbits_defs += tmp;
				if (tg.sumbits == 66)
					sumbits_defs += "#define GPUPREAGG_SUMBOUND_" + std::to_string(naggs) + " \"" + tg.sumbound + " \"\n";
				snprintf(tmp, sizeof(tmp), "#define GPUPREAGG_COUNTALL_%d %d\n", naggs,
						 (tg.kind == STROM_PREAGG_NROWS && tg.countall) ? 1 : 0);
				sumbits_defs += tmp;
				if (tg.kind == STROM_PREAGG_NROWS && tg.countall && countall_first < 0)
					countall_first = naggs;
				funcs += fn_header(tname, "gpupreagg_agg", naggs) + "{\n" + tg.body + "}\n";
				naggs++;
			}
			targets_out[i].kind = tg.kind;
			targets_out[i].type_oid = tg.type_oid;
			targets_out[i].scale = tg.scale;
		}
		*p_ntargets = (int)targets.size();

		std::string src = "/* generated by strom_codegen_gpupreagg */\n";
		src += codegen_includes(ctx.extra_flags);
		src += codegen_param_list(ctx);
		src += codegen_var_list(ctx, "STROM_KVAR_LIST");
		{
			/*
			 * the columns the qual and the keys read, and the rest: the hashed
			 * fold with roles (strom_gpupreagg.h) decides from the former
			 * whether a row is its own and loads the latter only then
			 */
			codegen_context kctx, qctx;
			kctx.var_label = qctx.var_label = "KVAR";
			kctx.var_struct = qctx.var_struct = "KV";
			for (size_t i = 1; i < tree.items.size(); i++)
			{
				const sexpr &t = tree.items[i];
				if (t.is_list && t.items.size() == 2 && !t.items[0].is_list &&
					(t.items[0].atom == "qual" || t.items[0].atom == "key"))
				{
					std::string e;
					codegen_expression(t.items[1], kctx, e);
					if (t.items[0].atom == "qual")
						codegen_expression(t.items[1], qctx, e);
				}
			}
			{
				/* which columns the qual reads (bit attno-1): the join-as-a-lookup kernel
				 * evaluates a qual over outer columns before it probes the table */
				unsigned long long qmask = 0;
				for (auto &v : qctx.used_vars)
					if (v.attno >= 1 && v.attno <= 64)
						qmask |= (1ULL << (v.attno - 1));
				char qb[64];
				snprintf(qb, sizeof(qb), "#define GPUPREAGG_QUAL_VARMASK 0x%llxUL\n", qmask);
				src += qb;
			}
			codegen_context rctx;
			for (auto &v : ctx.used_vars)
			{
				bool grouping = false;
				for (auto &k : kctx.used_vars)
					grouping = grouping || (k.attno == v.attno);
				if (!grouping)
					rctx.used_vars.push_back(v);
			}
			codegen_context gctx;
			for (auto &v : ctx.used_vars)
			{
				bool rest = false;
				for (auto &r : rctx.used_vars)
					rest = rest || (r.attno == v.attno);
				if (!rest)
					gctx.used_vars.push_back(v);
			}
			src += codegen_var_list(gctx, "STROM_KVAR_LIST_GROUPING");
			src += codegen_var_list(rctx, "STROM_KVAR_LIST_REST");
		}
		snprintf(tmp, sizeof(tmp),
				 "#define GPUPREAGG_NTARGETS %zu\n#define GPUPREAGG_NKEYS %d\n#define GPUPREAGG_NAGGS %d\n",
				 targets.size(), nkeys, naggs);
		src += tmp;
		src += key_list + "\n" + agg_list + "\n" + keycols + "\n";
		{
			/* the per-row fixed-point cache: X(attno, scale) (codegen_internal.h: fixed_cache) */
			std::string fl = "#define STROM_KFIXED_LIST(X)";
			for (auto &f : ctx.used_fixed)
			{
				snprintf(tmp, sizeof(tmp), " X(%d,%d)", f.attno, f.scale);
				fl += tmp;
			}
			src += fl + "\n";
		}
		/* packed accumulators: X(aidx, kind, attno of the source column) -- see strom_gpupreagg.h */
		snprintf(tmp, sizeof(tmp), "#define GPUPREAGG_NUMERIC_AGGS %d\n", numeric_aggs ? 1 : 0);
		src += tmp;
		snprintf(tmp, sizeof(tmp), "#define GPUPREAGG_PACKABLE %d\n", (packable && naggs > 0 && naggs <= 32) ? 1 : 0);
		src += tmp;
		src += pack_list + "\n";
		/* integer sums: static magnitude bound per aggregate (see struct target) */
		src += sumbits_defs;
		snprintf(tmp, sizeof(tmp), "#define GPUPREAGG_COUNTALL_FIRST %d\n", countall_first);
		src += tmp;
		{
			bool	any = false;
			for (auto &tg : targets)
				any = any || (tg.kind != STROM_PREAGG_KEY && tg.sumbits != 0);
			src += (any ? "#define GPUPREAGG_HAS_INTSUMS 1\n" : "#define GPUPREAGG_HAS_INTSUMS 0\n");
		}
		{
			/* wide rows: one quad per thread and tile, two would not leave
			 * registers for the row body (a -D tunable still wins) */
			int		row_bytes = 0;
			for (auto &v : ctx.used_vars)
			{
				const devtype_info *dt = devtype_lookup(v.type_oid);
				row_bytes += (dt && dt->type_length > 0 ? dt->type_length : 8);
			}
			if (row_bytes > 24)
				src += "#ifndef GPUPREAGG_QUADS\n#define GPUPREAGG_QUADS 1\n#endif\n";
		}
		src += "#include \"strom_gpupreagg.h\"\n";
		src += "STROM_DEVICE pg_bool_t\n"
			"gpupreagg_qual_eval(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV)\n{\n" +
			qual_body + "}\n";
		src += funcs;
		codegen_fill_result(ctx, src, out);
		return 0;
	} catch (const std::exception &e) {
		out->errmsg = strdup(e.what());
		return -1;
	}
}
