/*
 * codegen_internal.h -- shared between codegen.cpp and the per-operator
 * emitters (codegen_hashjoin.cpp, codegen_preagg.cpp)
 */
#ifndef STROM_CODEGEN_INTERNAL_H
#define STROM_CODEGEN_INTERNAL_H

#include <string>
#include <vector>
#include "strom_codegen.h"

namespace strom {

struct devtype_info {
	int			type_oid;
	const char *sql_name;
	const char *dev_name;		/* pg_<dev_name>_t */
	int			type_length;
	int			type_flags;		/* DEVFUNC_NEEDS_* of the declaring library */
};

struct sexpr {
	bool				is_list = false;
	std::string			atom;
	std::vector<sexpr>	items;
};

/* codegen_context (pg_strom.h:406-419) */
struct codegen_context {
	std::string						var_label;		/* "KVAR" */
	std::string						var_struct;		/* "KV" */
	std::vector<strom_kparam_desc>	used_params;
	std::vector<strom_kvar_desc>	used_vars;
	int								extra_flags = 0;
	/* hash join: (ivar depth attno type) refers to a matched inner tuple */
	struct ivar_ref { int depth, attno, type_oid; };
	int								ivar_max_depth = 0;	/* 0: ivar not allowed here */
	std::vector<ivar_ref>			used_ivars;
	/*
	 * (var N numeric SCALE) read through a per-row cache: the 64-bit image is converted to fixed
	 * point ONCE per row, where the row's variables are assembled (strom_kvars.KFIX_<N>_<SCALE>,
	 * filled by STROM_KVARS_FINISH), instead of once per use in every expression that reads the
	 * column -- the conversion is 40 % of a Q1-shaped row and the compiler does not merge the
	 * copies across the generated functions.  GpuPreAgg programs only (the emitter that owns the
	 * row assembly sets it).
	 */
	bool							fixed_cache = false;
	struct fixed_ref { int attno, scale; };
	std::vector<fixed_ref>			used_fixed;

	int		track_param(const strom_kparam_desc &d);
	void	track_var(int attno, int type_oid);
};

/* fixed-scale numerics: pseudo type ids inside the emitter (codegen.cpp) */
#define STROM_FIXED_BASE	0x7F000000
bool		codegen_type_is_fixed(int type);
int			codegen_fixed_scale(int type);
std::string	codegen_fixed_as_numeric(const std::string &text, int scale);
std::string	codegen_fixed_rescale(const std::string &text, int from, int to);
int			codegen_expression_raw(const sexpr &n, codegen_context &ctx, std::string &out);

[[noreturn]] void codegen_error(const char *fmt, ...);
sexpr		sexpr_parse(const char *text);
const devtype_info *devtype_lookup(int oid);
const devtype_info *devtype_lookup_by_name(const std::string &name);
const char *devtype_eqfunc(int oid);
const char *devtype_cmpfunc(int oid);
/* emits the expression text, returns its type oid */
int			codegen_expression(const sexpr &n, codegen_context &ctx, std::string &out);
std::string	codegen_param_list(const codegen_context &ctx);
std::string	codegen_var_list(const codegen_context &ctx, const char *macro_name);
std::string	codegen_includes(int extra_flags);
void		codegen_fill_result(const codegen_context &ctx, const std::string &source,
								strom_codegen_result *out);
/* device function name (without pgfn_) for name+argtypes, or empty */
std::string	devfunc_devname(const std::string &name, const std::vector<int> &argtypes,
							int *rettype, int *flags);

}	/* namespace strom */
#endif
