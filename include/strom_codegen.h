/*
 * strom_codegen.h -- C ABI of the expression emitter
 *
 * Replaces the back half of the reference's codegen.c: it turns an
 * expression tree into the text of HIP __device__ functions
 * (pgstrom_codegen_expression, codegen.c:1394-1430, and the per-operator
 * wrappers gpuscan_codegen_quals gpuscan.c:522-564, gpuhashjoin_codegen
 * gpuhashjoin.c:1353-1460, gpupreagg_codegen gpupreagg.c:1902-1943).
 *
 * The front half of codegen.c walks PostgreSQL Node trees and system
 * catalogs; there is no PostgreSQL here, so the tree arrives as text in a
 * small S-expression IR with the same node kinds the reference's walker
 * accepts (codegen.c:1065-1392):
 *
 *   (const TYPE LITERAL) | (const TYPE null)        Const
 *   (param INDEX TYPE)                              Param (PARAM_EXTERN)
 *   (var ATTNO TYPE)                                Var, ATTNO is 1-based
 *   (FUNCNAME arg ...)                              FuncExpr / OpExpr, by
 *                                                   pg_proc name, resolved
 *                                                   on argument types
 *   (and e ...) (or e ...) (not e)                  BoolExpr
 *   (isnull e) (isnotnull e)                        NullTest
 *   (is_true e) (is_not_true e) (is_false e) (is_not_false e)
 *   (is_unknown e) (is_not_unknown e)               BooleanTest
 *   (case (when c r) ... (else d))                  CaseExpr (searched)
 *   (case_eq arg (when v r) ... (else d))           CaseExpr (simple)
 *   (relabel TYPE e)                                RelabelType
 *
 * TYPE is one of: bool int2 int4 int8 float4 float8 date time timestamp
 * numeric char1.  Consts and Params both become KPARAM_<i>, de-duplicated
 * (codegen.c:1075-1130); Vars become KVAR_<attno> (1131-1145).
 */
#ifndef STROM_CODEGEN_H
#define STROM_CODEGEN_H

#include "strom_kds.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int32_t		type_oid;		/* STROM_*OID */
	int32_t		is_const;		/* 1: value[] is the constant; 0: external Param */
	int32_t		param_id;		/* external parameter number when !is_const */
	int32_t		isnull;
	int32_t		length;			/* bytes used in value[] */
	uint8_t		value[16];
} strom_kparam_desc;

typedef struct {
	int32_t		attno;			/* 1-based attribute number */
	int32_t		type_oid;
} strom_kvar_desc;

typedef struct {
	char	   *source;			/* malloc'ed program text */
	int32_t		extra_flags;	/* DEVFUNC_NEEDS_* | DEVKERNEL_NEEDS_* */
	int32_t		nparams;
	strom_kparam_desc *params;
	int32_t		nvars;
	strom_kvar_desc *vars;
	char	   *errmsg;			/* malloc'ed, NULL on success */
} strom_codegen_result;

/* GpuScan: qual is one boolean expression (an implicit-AND list is
 * written (and q1 q2 ...)), gpuscan.c:522-564 */
int		strom_codegen_gpuscan(const char *qual, strom_codegen_result *out);

/*
 * GpuPreAgg: gpupreagg_codegen (gpupreagg.c:1902-1943) -- qual_eval
 * (1851-1900), the per-target projection of partial-aggregate inputs
 * (1450-1837) and the key / aggregate catalogue the kernels switch on
 * (keycomp 1181-1284, aggcalc 1320-1440).  The target list is written
 *
 *   (gpupreagg [(qual BOOL-EXPR)] TARGET ...)
 *   TARGET := (key EXPR)                     grouping key (a Var in the reference)
 *           | (nrows BOOL-EXPR ...)          1 when every argument is TRUE, else 0
 *           | (psum EXPR) | (pmin EXPR) | (pmax EXPR)   numeric: (psum EXPR SCALE)
 *           | (psum_x2 EXPR)                 EXPR*EXPR as float8
 *           | (pcov_x F X Y) (pcov_y ..) (pcov_x2 ..) (pcov_y2 ..) (pcov_xy ..)
 *
 * in output-column order; these are the pgstrom.* partial functions of
 * pg_strom--1.0.sql:232-246.  targets[] describes each output column.
 */
#define STROM_PREAGG_KEY		1
#define STROM_PREAGG_NROWS		2
#define STROM_PREAGG_PSUM		3
#define STROM_PREAGG_PMIN		4
#define STROM_PREAGG_PMAX		5

typedef struct {
	int32_t		kind;			/* STROM_PREAGG_* (psum_x2 / pcov_* are PSUM) */
	int32_t		type_oid;		/* type of the partial value */
	int32_t		scale;			/* numeric partials: accumulated as fixed-point int8
								 * at 10^-scale; -1 otherwise.  Written (psum EXPR SCALE) */
} strom_preagg_target;

int		strom_codegen_gpupreagg(const char *spec, strom_codegen_result *out,
								strom_preagg_target *targets, int max_targets,
								int *p_ntargets);

/*
 * GpuHashJoin: gpuhashjoin_codegen (gpuhashjoin.c:1353-1460).
 *   (gpuhashjoin (rel (hashkey OUTER-EXPR INNER-ATTNO TYPE) ... [(qual BOOL-EXPR)]) ...)
 * One (rel ..) per inner relation, in join order.  OUTER-EXPR may use
 * (var ..) of the outer chunk and (ivar DEPTH ATTNO TYPE) of a relation
 * matched earlier; a qual may also use the current depth.
 */
int		strom_codegen_gpuhashjoin(const char *spec, strom_codegen_result *out, int *p_nrels);

/*
 * Can this expression run on the device?  (pgstrom_codegen_available_
 * expression, codegen.c:1631-1759.)  1 yes, 0 no; errmsg (if not NULL)
 * receives a malloc'ed reason.
 */
int		strom_codegen_available_expression(const char *expr, char **errmsg);

void	strom_codegen_release(strom_codegen_result *res);

/*
 * Build the kern_parambuf for a generated program
 * (pgstrom_create_kern_parambuf, datastore.c:41-148).  ext_values[i] /
 * ext_isnull[i] supply external Param i as a 64-bit datum image.
 * Result is malloc'ed; its 'length' field is the byte size.
 */
kern_parambuf *strom_create_kern_parambuf(const strom_codegen_result *res,
										  const uint64_t *ext_values,
										  const uint8_t *ext_isnull,
										  int n_ext);

#ifdef __cplusplus
}
#endif
#endif	/* STROM_CODEGEN_H */
