/*
 * codegen_hashjoin.cpp -- GpuHashJoin program text
 *
 * Role in the reference: gpuhashjoin_codegen (gpuhashjoin.c:1353-1460) with
 * gpuhashjoin_codegen_recurse (1184-1317), which emits gpuhashjoin_execute:
 * depth by depth, hash the outer-side key expressions, walk the chain of
 * the slot, load the inner columns the clauses reference, test
 * "hash equal && EVAL(hash clauses) && EVAL(quals)", recurse or emit
 * {outer_row+1, offset of inner entry per depth}.  The same function is
 * emitted here, in HIP, against this build's probe index
 * (strom_hashjoin.h) instead of CRC32 + the host-built slot array.
 *
 * IR:  (gpuhashjoin (rel (hashkey OUTER-EXPR INNER-ATTNO TYPE) ...
 *                        [(qual BOOL-EXPR)]) ...)
 * OUTER-EXPR / quals may use (var attno type) of the outer chunk and
 * (ivar depth attno type) of an inner relation already matched (for quals
 * also the current depth).
 */
#include <cstring>
#include <cstdio>
#include <string>
#include <vector>
#include <set>

#include "strom_codegen.h"
#include "strom_hip.h"
#include "codegen_internal.h"

using namespace strom;

namespace {

struct hashkey {
	std::string	outer_text;
	int			inner_attno;
	int			type_oid;
	int			outer_attno = 0;	/* > 0: the outer side is that plain column */
};

struct rel {
	std::vector<hashkey> keys;
	std::string	qual_text;		/* empty: none */
	std::set<std::pair<int,int>> ivars;	/* (attno, type) of this depth used anywhere */
	std::set<std::pair<int,int>> expr_ivars;	/* ... used by an expression (not only as a key column) */
};

/*
 * The 64-bit image of a key value the probe index hashes (and, in its DIRECT / KEYED forms,
 * compares instead of the value): the value itself for the by-value types; for text and
 * character(n) -- where the value is the ADDRESS of a datum -- a hash of the payload bytes, the
 * reference's pg_<type>_hashkey over VARDATA_ANY / VARSIZE_ANY_EXHDR (opencl_hashjoin.h:935-953).
 * Such an image says "maybe equal" only: image_is_exact() is false and the relation gets the HASH
 * index, whose probe compares the keys with the type's equality function.
 */
std::string
key_image_call(int type_oid, const std::string &value)
{
	if (type_oid == STROM_TEXTOID)
		return "hashjoin_varlena_image(" + value + ", false)";
	if (type_oid == STROM_BPCHARNOID)
		return "hashjoin_varlena_image(" + value + ", true)";	/* trailing blanks do not count */
	return "hashjoin_key_image(" + value + ")";
}
bool
image_is_exact(int type_oid)
{
	return type_oid != STROM_TEXTOID && type_oid != STROM_BPCHARNOID;
}

bool
type_is_intlike(int oid)
{
	return oid == STROM_INT2OID || oid == STROM_INT4OID || oid == STROM_INT8OID ||
		oid == STROM_DATEOID || oid == STROM_BOOLOID || oid == STROM_BPCHAROID ||
		oid == STROM_TIMEOID || oid == STROM_TIMESTAMPOID;
}

}	/* namespace */

extern "C" int
strom_codegen_gpuhashjoin(const char *spec, strom_codegen_result *out, int *p_nrels)
{
	memset(out, 0, sizeof(*out));
	try {
		codegen_context ctx;
		ctx.var_label = "KVAR";
		ctx.var_struct = "KV";
		ctx.extra_flags = DEVKERNEL_NEEDS_HASHJOIN | DEVFUNC_NEEDS_MATHLIB;
		sexpr tree = sexpr_parse(spec);
		if (!tree.is_list || tree.items.empty() || tree.items[0].is_list ||
			tree.items[0].atom != "gpuhashjoin")
			codegen_error("(gpuhashjoin ...) expected");
		std::vector<rel> rels;
		for (size_t i = 1; i < tree.items.size(); i++)
		{
			const sexpr &r = tree.items[i];
			if (!r.is_list || r.items.empty() || r.items[0].is_list || r.items[0].atom != "rel")
				codegen_error("(rel ...) expected");
			rel R;
			int depth = (int)rels.size() + 1;
			ctx.ivar_max_depth = depth - 1;		/* keys may look at earlier depths only */
			for (size_t j = 1; j < r.items.size(); j++)
			{
				const sexpr &it = r.items[j];
				if (!it.is_list || it.items.empty() || it.items[0].is_list)
					codegen_error("rel item must be a list");
				if (it.items[0].atom == "hashkey")
				{
					if (it.items.size() != 4 || it.items[2].is_list || it.items[3].is_list)
						codegen_error("(hashkey OUTER-EXPR INNER-ATTNO TYPE) expected");
					hashkey k;
					const devtype_info *t = devtype_lookup_by_name(it.items[3].atom);
					if (!t)
						codegen_error("unknown hashkey type %s", it.items[3].atom.c_str());
					ctx.ivar_max_depth = depth - 1;
					int ot = codegen_expression(it.items[1], ctx, k.outer_text);
					if (ot != t->type_oid)
						codegen_error("hashkey: outer expression is %s but inner column is %s "
									  "(cast the outer side)", devtype_lookup(ot)->sql_name, t->sql_name);
					if (it.items[1].is_list && it.items[1].items.size() >= 3 && !it.items[1].items[0].is_list &&
						it.items[1].items[0].atom == "var" && !it.items[1].items[1].is_list)
						k.outer_attno = atoi(it.items[1].items[1].atom.c_str());
					k.inner_attno = atoi(it.items[2].atom.c_str());
					if (k.inner_attno < 1)
						codegen_error("inner attribute numbers start at 1");
					k.type_oid = t->type_oid;
					/* expressions normalise numerics lazily (strom_numeric.h);
					 * the hash runs over the datum image */
					if (k.type_oid == STROM_NUMERICOID)
						k.outer_text = "pgfn_numeric_normalize(errcode, " + k.outer_text + ")";
					if (!devtype_eqfunc(k.type_oid))
						codegen_error("type %s has no equality function", t->sql_name);
					ctx.extra_flags |= t->type_flags;
					R.keys.push_back(k);
				}
				else if (it.items[0].atom == "qual")
				{
					if (it.items.size() != 2)
						codegen_error("(qual EXPR) expected");
					ctx.ivar_max_depth = depth;
					if (codegen_expression(it.items[1], ctx, R.qual_text) != STROM_BOOLOID)
						codegen_error("join qual is not boolean");
				}
				else
					codegen_error("unknown rel item %s", it.items[0].atom.c_str());
			}
			if (R.keys.empty())
				codegen_error("a rel needs at least one hashkey");
			if (R.keys.size() > 8)
				codegen_error("too many hash keys");
			rels.push_back(R);
		}
		if (rels.empty() || rels.size() > 8)
			codegen_error("1..8 inner relations expected");
		/* inner columns referenced through (ivar ..), plus every key column */
		for (auto &iv : ctx.used_ivars)
		{
			rels[iv.depth - 1].ivars.insert({iv.attno, iv.type_oid});
			rels[iv.depth - 1].expr_ivars.insert({iv.attno, iv.type_oid});
		}
		for (auto &R : rels)
			for (auto &k : R.keys)
				R.ivars.insert({k.inner_attno, k.type_oid});

		int		nrels = (int)rels.size();
		char	tmp[1024];
		std::string src = "/* generated by strom_codegen_gpuhashjoin */\n";
		src += codegen_includes(ctx.extra_flags);
		src += codegen_param_list(ctx);
		src += codegen_var_list(ctx, "STROM_KVAR_LIST");
		snprintf(tmp, sizeof(tmp), "#define HASHJOIN_NRELS %d\n", nrels);
		src += tmp;
		/* fast path: one relation, one integer-like key, nothing else to test */
		/* (a qual that reads outer columns only -- a scan's WHERE pulled up into
		 * the join -- rides along: HASHJOIN_FAST_OUTER_QUAL) */
		bool	fast = (nrels == 1 && rels[0].keys.size() == 1 &&
						type_is_intlike(rels[0].keys[0].type_oid) && ctx.used_ivars.empty());
		bool	fast_qual = (fast && !rels[0].qual_text.empty());
		snprintf(tmp, sizeof(tmp), "#define HASHJOIN_FAST_ELIGIBLE %d\n#define HASHJOIN_FAST_OUTER_QUAL %d\n",
				 fast ? 1 : 0, fast_qual ? 1 : 0);
		src += tmp;
		/* the outer key is a plain column: a consumer can find a joined row's slot
		 * from the outer chunk alone (strom_submit_gpupreagg_joined) */
		snprintf(tmp, sizeof(tmp), "#define HASHJOIN_FAST_OUTER_KEY_ATTNO %d\n",
				 fast ? rels[0].keys[0].outer_attno : 0);
		src += tmp;
		for (int d = 1; d <= nrels; d++)
		{
			snprintf(tmp, sizeof(tmp), "#define HASHJOIN_NKEYS_%d %zu\n", d, rels[d - 1].keys.size());
			src += tmp;
			snprintf(tmp, sizeof(tmp), "#define HASHJOIN_KEY0_INTLIKE_%d %d\n", d,
					 (rels[d - 1].keys.size() == 1 && type_is_intlike(rels[d - 1].keys[0].type_oid)) ? 1 : 0);
			src += tmp;
			/* a KEYED index finds a key by its image alone: one key, whose image is its value */
			snprintf(tmp, sizeof(tmp), "#define HASHJOIN_KEYED_OK_%d %d\n", d,
					 (rels[d - 1].keys.size() == 1 && image_is_exact(rels[d - 1].keys[0].type_oid)) ? 1 : 0);
			src += tmp;
		}
		src += "#include \"strom_hashjoin.h\"\n";

		/* ---- inner key images (used by the index build kernels) -------- */
		src += "STROM_DEVICE bool\n"
			"hashjoin_inner_key_images(int depth, const kern_hashtable *kht, const kern_hashentry *ent,\n"
			"                          cl_ulong *images)\n{\n  cl_int errcode_ = 0; cl_int *errcode = &errcode_;\n"
			"  switch (depth)\n  {\n";
		for (int d = 1; d <= nrels; d++)
		{
			snprintf(tmp, sizeof(tmp), "    case %d:\n", d);
			src += tmp;
			for (size_t k = 0; k < rels[d - 1].keys.size(); k++)
			{
				const hashkey &hk = rels[d - 1].keys[k];
				const char *tn = devtype_lookup(hk.type_oid)->dev_name;
				/* numeric: the index hashes the datum image, which must be the
				 * canonical one (a varlena numeric decodes un-normalised) */
				snprintf(tmp, sizeof(tmp),
						 "      { pg_%s_t v = %spg_%s_tupref(errcode, kht->colmeta, &ent->htup, %d));\n"
						 "        if (v.isnull) return false;\n"
						 "        images[%zu] = %s; }\n",
						 tn, hk.type_oid == STROM_NUMERICOID ? "pgfn_numeric_normalize(errcode, " : "(",
						 tn, hk.inner_attno - 1, k, key_image_call(hk.type_oid, "v.value").c_str());
				src += tmp;
			}
			src += "      return true;\n";
		}
		src += "  }\n  (void)errcode;\n  return false;\n}\n";

		/* ---- outer key of the fast path -------------------------------- */
		if (fast)
		{
			const char *tn = devtype_lookup(rels[0].keys[0].type_oid)->dev_name;
			src += std::string("STROM_DEVICE bool\n"
				"hashjoin_fast_outer_key(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV,\n"
				"                        cl_long *p_key)\n{\n  pg_") + tn + "_t k = " +
				rels[0].keys[0].outer_text + ";\n  *p_key = (cl_long)k.value;\n  return !k.isnull;\n}\n";
		}
		else
			src += "STROM_DEVICE bool\n"
				"hashjoin_fast_outer_key(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV,\n"
				"                        cl_long *p_key)\n{\n  return false;\n}\n";
		if (fast_qual)
			src += "STROM_DEVICE bool\n"
				"hashjoin_fast_outer_qual(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV)\n"
				"{\n  pg_bool_t q = " + rels[0].qual_text + ";\n  return EVAL(q);\n}\n";

		/* ---- gpuhashjoin_execute: nested probe loops --------------------- */
		/* ALL_SINGLE: every relation has a DIRECT index with unique keys (the
		 * kernel checks once per launch); the mode tests then fold away and
		 * no hash entry is read at all */
		src += "template <bool ALL_SINGLE>\nSTROM_DEVICE cl_uint\n"
			"gpuhashjoin_execute(cl_int *errcode, const strom_kparams &KP, const strom_kvars &KV,\n"
			"                    const kern_multihash *__restrict__ kmhash, const hashjoin_index *__restrict__ hjidx,\n"
			"                    cl_uint kds_index, cl_int *__restrict__ rbuffer, cl_int *first_match)\n{\n"
			"  cl_uint n_matches = 0;\n";
		std::string indent = "  ";
		std::string closing;
		for (int d = 1; d <= nrels; d++)
		{
			const rel &R = rels[d - 1];
			snprintf(tmp, sizeof(tmp), "%sconst kern_hashtable *kht_%d = KERN_HASHTABLE(kmhash, %d);\n",
					 indent.c_str(), d, d - 1);
			src += tmp;
			std::string allnotnull;
			for (size_t k = 0; k < R.keys.size(); k++)
			{
				const char *tn = devtype_lookup(R.keys[k].type_oid)->dev_name;
				snprintf(tmp, sizeof(tmp), "%spg_%s_t okey_%d_%zu = ", indent.c_str(), tn, d, k);
				src += tmp + R.keys[k].outer_text + ";\n";
				snprintf(tmp, sizeof(tmp), "%s!okey_%d_%zu.isnull", k ? " && " : "", d, k);
				allnotnull += tmp;
			}
			src += indent + "if (" + allnotnull + ")\n" + indent + "{\n";
			indent += "  ";
			snprintf(tmp, sizeof(tmp), "%scl_ulong kimg_%d[%zu];\n", indent.c_str(), d, R.keys.size());
			src += tmp;
			for (size_t k = 0; k < R.keys.size(); k++)
			{
				char	oval[48];
				snprintf(oval, sizeof(oval), "okey_%d_%zu.value", d, k);
				snprintf(tmp, sizeof(tmp), "%skimg_%d[%zu] = %s;\n",
						 indent.c_str(), d, k, key_image_call(R.keys[k].type_oid, oval).c_str());
				src += tmp;
			}
			/*
			 * A DIRECT index (one integer-like key, slots[key - min]) chains
			 * exactly the entries of that key: the key comparison is implied
			 * and, when the keys are unique too, the entry is not read at all
			 * -- the probe then costs one 4-byte slot read per outer row.
			 */
			/* ... and so does a KEYED index (one key of any type: a slot per distinct
			 * key image, found by comparing images inside the 16-byte slot) */
			bool	direct_ok = (R.keys.size() == 1 && image_is_exact(R.keys[0].type_oid));
			snprintf(tmp, sizeof(tmp),
					 "%scl_uint hash_%d;\n"
					 "%sconst bool direct_%d = %s;\n"
					 "%sconst bool single_%d = ALL_SINGLE || (direct_%d && hjidx->rel[%d].unique != 0);\n"
					 "%sfor (cl_uint off_%d = hashjoin_first(hjidx, %d, kimg_%d, %zu, &hash_%d), next_%d = 0;\n"
					 "%s     off_%d != 0;\n"
					 "%s     off_%d = next_%d)\n%s{\n",
					 indent.c_str(), d,
					 indent.c_str(), d,
					 direct_ok ? ("(ALL_SINGLE || hjidx->rel[" + std::to_string(d - 1) + "].mode != HASHJOIN_MODE_HASH)").c_str() : "false",
					 indent.c_str(), d, d, d - 1,
					 indent.c_str(), d, d - 1, d, R.keys.size(), d, d,
					 indent.c_str(), d,
					 indent.c_str(), d, d, indent.c_str());
			src += tmp;
			indent += "  ";
			snprintf(tmp, sizeof(tmp),
					 "%sconst kern_hashentry *ent_%d = (const kern_hashentry *)((const char *)kht_%d + off_%d);\n"
					 "%snext_%d = (single_%d ? 0 : ent_%d->next);\n"
					 "%sif (!direct_%d && ent_%d->hash != hash_%d)\n%s  continue;\n",
					 indent.c_str(), d, d, d,
					 indent.c_str(), d, d, d,
					 indent.c_str(), d, d, d, indent.c_str());
			src += tmp;
			/* inner columns an expression refers to are fetched here; columns
			 * that only serve the key comparison inside its branch */
			for (auto &iv : R.expr_ivars)
			{
				const char *tn = devtype_lookup(iv.second)->dev_name;
				snprintf(tmp, sizeof(tmp),
						 "%spg_%s_t IVAR_%d_%d = pg_%s_tupref(errcode, kht_%d->colmeta, &ent_%d->htup, %d);\n",
						 indent.c_str(), tn, d, iv.first, tn, d, d, iv.first - 1);
				src += tmp;
			}
			snprintf(tmp, sizeof(tmp), "%sbool keys_equal_%d = true;\n%sif (!direct_%d)\n%s{\n",
					 indent.c_str(), d, indent.c_str(), d, indent.c_str());
			src += tmp;
			/* hash clauses: outer key = inner column, by the type's equality function */
			for (size_t k = 0; k < R.keys.size(); k++)
			{
				int		rettype, flags;
				std::string eq = devfunc_devname(devtype_eqfunc(R.keys[k].type_oid),
												 {R.keys[k].type_oid, R.keys[k].type_oid},
												 &rettype, &flags);
				if (eq.empty())
					codegen_error("no device equality function for hash key");
				ctx.extra_flags |= flags;
				const char *tn = devtype_lookup(R.keys[k].type_oid)->dev_name;
				std::string ivname;
				if (R.expr_ivars.count({R.keys[k].inner_attno, R.keys[k].type_oid}))
				{
					snprintf(tmp, sizeof(tmp), "IVAR_%d_%d", d, R.keys[k].inner_attno);
					ivname = tmp;
				}
				else
				{
					snprintf(tmp, sizeof(tmp), "ikey_%d_%zu", d, k);
					ivname = tmp;
					snprintf(tmp, sizeof(tmp),
							 "%s  pg_%s_t %s = pg_%s_tupref(errcode, kht_%d->colmeta, &ent_%d->htup, %d);\n",
							 indent.c_str(), tn, ivname.c_str(), tn, d, d, R.keys[k].inner_attno - 1);
					src += tmp;
				}
				snprintf(tmp, sizeof(tmp),
						 "%s  keys_equal_%d = keys_equal_%d && EVAL(pgfn_%s(errcode, okey_%d_%zu, %s));\n",
						 indent.c_str(), d, d, eq.c_str(), d, k, ivname.c_str());
				src += tmp;
			}
			src += indent + "}\n";
			std::string cond = "keys_equal_" + std::to_string(d);
			if (!R.qual_text.empty())
				cond += " && EVAL(" + R.qual_text + ")";
			src += indent + "if (" + cond + ")\n" + indent + "{\n";
			indent += "  ";
			closing += "}\n}\n}\n";		/* if (cond), for, if (keys not null) */
		}
		/* innermost: emit one record */
		src += indent + "if (rbuffer)\n" + indent + "{\n" +
			indent + "  __builtin_nontemporal_store((cl_int)(kds_index + 1), &rbuffer[0]);\n";
		for (int d = 1; d <= nrels; d++)
		{
			snprintf(tmp, sizeof(tmp), "%s  __builtin_nontemporal_store((cl_int)off_%d, &rbuffer[%d]);\n",
					 indent.c_str(), d, d);
			src += tmp;
		}
		snprintf(tmp, sizeof(tmp), "%s  rbuffer += %d;\n%s}\n", indent.c_str(), nrels + 1, indent.c_str());
		src += tmp;
		/* (the count pass keeps the first match: a row with exactly one is emitted from it, not probed again) */
		src += indent + "else if (first_match && n_matches == 0)\n" + indent + "{\n";
		for (int d = 1; d <= nrels; d++)
		{
			snprintf(tmp, sizeof(tmp), "%s  first_match[%d] = (cl_int)off_%d;\n", indent.c_str(), d - 1, d);
			src += tmp;
		}
		src += indent + "}\n" + indent + "n_matches++;\n";
		src += closing;
		src += "  return n_matches;\n}\n";

		codegen_fill_result(ctx, src, out);
		out->extra_flags = ctx.extra_flags;
		if (p_nrels)
			*p_nrels = nrels;
		return 0;
	} catch (const std::exception &e) {
		out->errmsg = strdup(e.what());
		return -1;
	}
}
