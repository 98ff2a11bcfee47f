/*
 * strom_datastore.h -- C ABI for building / converting chunks on the host
 *
 * Role in the reference: datastore.c -- init_kern_data_store (312-380),
 * pgstrom_create_data_store_row / _row_flat / _tupslot (382-529),
 * pgstrom_data_store_insert_block (556-710) and _insert_tuple (718-828).
 * Those read PostgreSQL heap pages and TupleTableSlots; here the rows
 * arrive as plain column arrays and the functions lay them out in the
 * same bytes: real heap pages with line pointers for ROW, packed heap
 * tuples growing from the tail for ROW_FLAT, Datum/isnull pairs for
 * TUPSLOT, and the column-major arrays of KDS_FORMAT_COLUMN.
 */
#ifndef STROM_DATASTORE_H
#define STROM_DATASTORE_H

#include "strom_kds.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int32_t			type_oid;	/* STROM_*OID */
	int16_t			attlen;		/* 1,2,4,8 */
	int8_t			attalign;	/* 1,2,4,8 */
	int8_t			attbyval;
	const void	   *values;		/* nrows * attlen bytes; attlen -1 (heap formats only): NUMERIC as
								 * nrows 64-bit device images, encoded as PostgreSQL varlena -- or,
								 * with attalign -1, nrows pointers to complete varlena datums that
								 * go into the tuples verbatim */
	const uint8_t  *isnull;		/* nrows bytes, 1 = NULL; may be NULL */
} strom_column_input;

/* bytes a chunk of 'nrows' rows needs in 'format' (0 on bad input) */
size_t	strom_kds_required_length(int format, int ncols,
								  const strom_column_input *cols,
								  uint32_t nrows);
/*
 * Lay the rows out.  'buffer' must hold strom_kds_required_length() bytes
 * and be 16-byte aligned.  The image is contiguous -- for ROW the heap
 * pages follow the row items at the BLCKSZ-aligned offset the device
 * expects (KERN_DATA_STORE_ROWBLOCK), i.e. what clserv_dmasend_data_store
 * assembles on the device.  Returns 0 or a StromError code.
 */
int		strom_kds_build(int format, int ncols,
						const strom_column_input *cols,
						uint32_t nrows, void *buffer, size_t buflen);

/*
 * Head only of a KDS_FORMAT_COLUMN chunk of 'nrows' rows without NULLs
 * (init_kern_data_store, datastore.c:312-380, for a chunk whose payload is
 * produced elsewhere): writes KDS_COLUMN_HEAD_LENGTH(ncols) bytes to 'head'
 * (16-byte aligned; cols[i].values / isnull are not read) and returns the
 * length of the whole chunk, 0 on bad input.  values_off[i] (if not NULL)
 * receives the chunk offset where column i's nrows * attlen value bytes go.
 * minmax (if not NULL) = zone map {min, max} per column: int64 for
 * integer-like types, IEEE double bits for float4/float8.  A buffer laid out
 * this way in device memory becomes a resident chunk with strom_dstore_wrap().
 */
size_t	strom_kds_column_head(int ncols, const strom_column_input *cols, uint32_t nrows,
							  const int64_t *minmax, void *head, size_t headlen,
							  uint32_t *values_off);

/*
 * ROW / ROW_FLAT / TUPSLOT -> COLUMN on the host (the ingest step).
 * Returns required length when dst == NULL.
 */
size_t	strom_kds_to_column(const kern_data_store *src, void *dst, size_t dstlen);

/* fetch one datum on the host: pgstrom_fetch_data_store (datastore.c:169-242).
 * Returns 1 when NULL, 0 otherwise with up to 8 bytes stored in *value. */
int		strom_kds_fetch(const kern_data_store *kds, uint32_t rowidx, uint32_t colidx,
						uint64_t *value);

/*
 * A 64-bit device numeric (exponent 63..58, sign 57, mantissa 56..0: opencl_numeric.h:122-160) as
 * PostgreSQL's own numeric datum  <- pgstrom_fixup_kernel_numeric (datastore.c:150-167), which the
 * backend applies to every NUMERIC column of a row that comes back from the device
 * (pgstrom_fetch_data_store, datastore.c:169-242).  The reference prints "%c%lue%d" and calls
 * numeric_in(); both halves are here:
 *   strom_kernel_numeric_cstring   that very text (sign, mantissa, 'e', exponent) -- what a backend
 *                                  hands to numeric_in() if it wants PostgreSQL to build the datum;
 *   strom_fixup_kernel_numeric     the datum itself as PostgreSQL 9.4 stores it (utils/adt/numeric.c:
 *                                  4-byte varlena header, short numeric header when weight and
 *                                  display scale fit, base-10000 digits), display scale = the
 *                                  digits behind the decimal point, as numeric_in gives it.
 * Both return the bytes written (the text without its NUL), or -StromError_DataStoreNoSpace.
 */
int		strom_kernel_numeric_cstring(uint64_t image, char *buf, size_t room);
int		strom_fixup_kernel_numeric(uint64_t image, void *varlena_out, size_t room);

/*
 * Inner side of a hash join: multihash_preload_khashtable
 * (gpuhashjoin.c:3614-3816).  One kern_hashtable per inner relation inside
 * one kern_multihash; each entry owns the inner row as a heap tuple,
 * rowid = row number in 'inner', hash = pg_crc32 of the key datums.
 */
typedef struct {
	const kern_data_store *inner;	/* any format */
	int32_t			nkeys;
	int32_t			key_attnos[8];	/* 1-based */
} strom_hashtable_input;

size_t	strom_multihash_required_length(int ntables, const strom_hashtable_input *tables);
int		strom_multihash_build(int ntables, const strom_hashtable_input *tables,
							  void *buffer, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif	/* STROM_DATASTORE_H */
