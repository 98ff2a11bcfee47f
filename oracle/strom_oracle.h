/*
 * strom_oracle.h -- CPU oracle for the GpuScan / GpuHashJoin / GpuPreAgg path
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, loaded
 * by or called from the product (pg_strom_amd/, libstrom_hip.so).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * as the checker / the timed CPU baseline.
 *
 * It restates, in plain single-threaded C, what the reference computes for
 * one chunk -- tuple-at-a-time over the reference's own chunk formats,
 * the way PostgreSQL's executor (the reference's test oracle,
 * input/make_expected.sh:22-28) and the reference's device code do:
 *
 *   oracle_get_datum      kern_get_datum*        opencl_common.h:817-981
 *   expression evaluator  generated device code  codegen.c:1065-1392 with
 *                         the arithmetic rules of opencl_mathlib.h:34-812
 *   oracle_gpuscan        gpuscan_qual +         opencl_gpuscan.h:98-177
 *                         gpuscan_writeback_row_error
 *
 * Pinning: see oracle/README.md -- struct sizes/offsets against SURVEY.md
 * Appendix A (compiled from the reference headers), aggregate results
 * against the reference's expected/<suite>.out files (tests/golden/).
 */
#ifndef STROM_ORACLE_H
#define STROM_ORACLE_H

#include "strom_kds.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one SQL value */
typedef struct {
	int32_t		type_oid;
	int32_t		isnull;
	union {
		int64_t		i;
		double		d;
		float		f;
		uint64_t	u;
	} v;
} oracle_value;

typedef struct oracle_expr oracle_expr;

/* parse the S-expression IR of include/strom_codegen.h; NULL + message on error */
oracle_expr *oracle_expr_parse(const char *text, char *errbuf, size_t errlen);
void		 oracle_expr_free(oracle_expr *expr);
/* evaluate on row 'rowidx' of kds; *errcode accumulates with the
 * STROM_SET_ERROR priority rule */
oracle_value oracle_expr_eval(const oracle_expr *expr,
							  const kern_data_store *kds, uint32_t rowidx,
							  const uint64_t *ext_values, const uint8_t *ext_isnull,
							  int n_ext, int32_t *errcode);

/* address of a datum or NULL (SQL NULL / out of range), any format */
const void  *oracle_get_datum(const kern_data_store *kds, uint32_t colidx, uint32_t rowidx);

/*
 * GpuScan over one chunk.  results[] must have room for nitems (or
 * nvalids) entries; it receives +(row+1) for passing rows and -(row+1)
 * for rows that need a CPU recheck, in ascending row order.  Returns the
 * chunk errcode (0 or the first significant error).
 */
int32_t		oracle_gpuscan(const char *qual,
						   const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
						   const kern_data_store *kds, const kern_row_map *krowmap,
						   int32_t *results, uint32_t *p_nitems,
						   char *errbuf, size_t errlen);

/*
 * GpuPreAgg over one chunk: one partial row per group.  out_values /
 * out_isnull are [max_groups x ntargets] in target-list order; values are
 * raw 8-byte images (int64, or the bits of a double for float partials;
 * float4 partials are carried as double).  Returns 0, StromError_CpuReCheck
 * (nothing produced) or another error.
 */
int32_t		oracle_gpupreagg(const char *spec,
							 const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
							 const kern_data_store *kds, const kern_row_map *krowmap,
							 uint32_t max_groups,
							 uint64_t *out_values, uint8_t *out_isnull,
							 uint32_t *p_ngroups,
							 char *errbuf, size_t errlen);

/*
 * GpuHashJoin over one outer chunk.  results[] receives (ninner+1) ints per
 * match: outer_row+1, then the ROW INDEX of the matched tuple of each inner
 * relation.  Returns 0, StromError_DataStoreNoSpace (*p_nitems = required),
 * or an error.
 */
int32_t		oracle_gpuhashjoin(const char *spec,
							   const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
							   const kern_data_store *outer, const kern_row_map *krowmap,
							   const kern_data_store *const *inner, int ninner,
							   int32_t *results, uint32_t nrooms, uint32_t *p_nitems,
							   char *errbuf, size_t errlen);
uint32_t	oracle_pg_crc32(uint32_t crc, const void *data, size_t len);
long		oracle_check_hashtable(const kern_multihash *kmhash, int depth,
								   const kern_data_store *inner,
								   const int *key_attnos, const int *key_lens, int nkeys);

int32_t		oracle_eval_rows(const char *expr,
							 const uint64_t *ext_values, const uint8_t *ext_isnull, int n_ext,
							 const kern_data_store *kds,
							 uint64_t *out_values, uint8_t *out_isnull, int32_t *out_errcode,
							 int32_t *p_type_oid, char *errbuf, size_t errlen);
int			oracle_numeric_from_text(const char *lit, uint64_t *out);
/* PostgreSQL's varlena numeric (heap tuple datum) -> 64-bit form; 0 = not representable */
int			oracle_numeric_from_varlena(const void *addr, uint64_t *out);

/* sizes / offsets of the wire structs, for the layout tests */
typedef struct {
	uint32_t	sizeof_kern_data_store_head;
	uint32_t	sizeof_kern_colmeta;
	uint32_t	sizeof_kern_rowitem;
	uint32_t	sizeof_kern_blkitem;
	uint32_t	offsetof_resultbuf_results;
	uint32_t	sizeof_kern_parambuf_head;
	uint32_t	sizeof_kern_hashentry;
	uint32_t	offsetof_hashentry_htup;
	uint32_t	offsetof_htup_t_bits;
	uint32_t	sizeof_kern_multihash_head;
	uint32_t	offsetof_gpupreagg_kparams;
	uint32_t	sizeof_kern_coldir;
} oracle_layout;
void		oracle_get_layout(oracle_layout *out);

#ifdef __cplusplus
}
#endif
#endif
