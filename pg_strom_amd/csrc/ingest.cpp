/*
 * ingest.cpp -- ROW / ROW_FLAT / TUPSLOT chunk -> KDS_FORMAT_COLUMN in HBM
 *
 * The caller side of the hot path: the reference
 * ships heap pages to the device for every request
 * (clserv_dmasend_data_store, datastore.c:837-973) and every kernel walks
 * tuples.  Here a resident chunk is transposed ONCE by ingest_to_column
 * (devlib/strom_ingest.h) and the streaming kernels then read exactly the
 * referenced bytes.  The layout equals what strom_kds_build() produces
 * for KDS_FORMAT_COLUMN on the host, so both can be compared byte by byte.
 */
#include <cstring>
#include <vector>

#include "runtime.h"

using namespace strom;

namespace {

const char *ingest_source =
	"#include \"strom_kds.h\"\n"
	"#include \"strom_common.h\"\n"
	"#include \"strom_mathlib.h\"\n"
	"#include \"strom_numeric.h\"\n"
	"#include \"strom_ingest.h\"\n";

}	/* namespace */

extern "C" strom_dstore *
strom_dstore_to_column(strom_dstore *src, const int32_t *type_oids, int ntypes,
					   uint64_t *p_kern_ns, int *p_errcode)
{
	STROM_ABI_TRY
	int		dummy;
	if (!p_errcode)
		p_errcode = &dummy;
	*p_errcode = 0;
	if (p_kern_ns)
		*p_kern_ns = 0;
	if (!src)
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	Device *dev = get_device(src->dindex);
	if (!dev)
	{
		*p_errcode = StromError_ServerNotReady;
		return nullptr;
	}
	int		ncols = (int)src->head.ncols;
	cl_uint	nitems = src->head.nitems;
	int		format = src->head.format;
	if (ncols < 1 || ncols > 64 || (type_oids && ntypes != ncols) ||
		!(format == KDS_FORMAT_ROW || format == KDS_FORMAT_ROW_FLAT || format == KDS_FORMAT_TUPSLOT))
	{
		*p_errcode = StromError_BadRequestMessage;
		return nullptr;
	}
	/* one reference is kept for the life of the process */
	static strom_devprog_key key = strom_get_devprog_key(ingest_source, 0);
	if (strom_lookup_device_program(key, 1) != STROM_DEVPROG_READY)
	{
		*p_errcode = StromError_ProgramBuildFailure;
		return nullptr;
	}
	Program *prog = lookup_program(key);
	int		errcode = 0;
	(void)hipSetDevice(dev->hip_id);
	/* the decoder for varlena NUMERIC columns rides only when one is there */
	bool	varnum = false;
	for (int i = 0; type_oids && i < ncols; i++)
		varnum = varnum || (type_oids[i] == STROM_NUMERICOID) || STROM_TYPE_IS_DECIMAL(type_oids[i]);
	/* ... and the heap-area writer only for text / character(n) columns (named by their type: a
	 * varlena column of unknown type is not guessed at) */
	auto is_text_type = [](int32_t oid) {
		/* (1042 is pg_type's bpchar of any length; a by-value char(1) column has attlen 1) */
		return oid == STROM_TEXTOID || oid == STROM_BPCHARNOID || oid == STROM_BPCHAROID;
	};
	hipFunction_t fn_main = prog->get_function(dev, varnum ? "ingest_to_column_varnum" : "ingest_to_column",
											   &errcode);
	hipFunction_t fn_heap = fn_main ? prog->get_function(dev, "ingest_to_column_varlena", &errcode) : nullptr;
	if (!fn_heap)
		fn_main = nullptr;
	hipFunction_t fn_fin = fn_main ? prog->get_function(dev, "ingest_finish", &errcode) : nullptr;
	hipFunction_t fn_mm = fn_fin ? prog->get_function(dev, "ingest_minmax", &errcode) : nullptr;
	if (!fn_main || !fn_fin || !fn_mm)
	{
		*p_errcode = errcode;
		return nullptr;
	}
	hipStream_t stream = dev->streams[0];
	/* source column metadata */
	size_t	head_len = KDS_HEAD_LENGTH(ncols);
	std::vector<char> hbuf(KDS_COLUMN_HEAD_LENGTH(ncols), 0);
	if (hipMemcpy(hbuf.data(), src->devptr, head_len, hipMemcpyDeviceToHost) != hipSuccess)
	{
		*p_errcode = StromError_HipInternal;
		return nullptr;
	}
	kern_data_store *head = (kern_data_store *)hbuf.data();
	kern_coldir *cd = KERN_DATA_STORE_COLDIR(head);
	size_t	off = KDS_COLUMN_HEAD_LENGTH(ncols);
	int		nheapcols = 0;
	for (int i = 0; i < ncols; i++)
	{
		int attlen = head->colmeta[i].attlen;
		if (type_oids && STROM_TYPE_IS_DECIMAL(type_oids[i]) && !(attlen == -1 || attlen == 8))
		{
			*p_errcode = StromError_BadRequestMessage;		/* not a numeric column */
			return nullptr;
		}
		if (attlen == -1 && type_oids &&
			(type_oids[i] == STROM_NUMERICOID || STROM_TYPE_IS_DECIMAL(type_oids[i])))
		{
			/* PostgreSQL's varlena numeric -> the 8-byte by-value device form
			 * (what the reference does to colmeta, datastore.c:355-363) */
			attlen = 8;
			head->colmeta[i].attlen = 8;
			head->colmeta[i].attalign = 8;
			head->colmeta[i].attbyval = 1;
		}
		bool	heap_col = (attlen == -1 && type_oids && is_text_type(type_oids[i]));
		if (!(attlen == 1 || attlen == 2 || attlen == 4 || attlen == 8 || heap_col))
		{
			*p_errcode = StromError_DataStoreCorruption;	/* varlena columns of other types: host path */
			return nullptr;
		}
		nheapcols += (heap_col ? 1 : 0);
		cd[i].values_off = (cl_uint)off;
		off += KDS_COLUMN_VALUES_LENGTH(attlen, nitems);
		cd[i].nulls_off = (cl_uint)off;			/* dropped by ingest_finish if unused */
		off += KDS_COLUMN_NULLS_LENGTH(nitems);
		cd[i].extra_off = 0;
		cd[i].stat_flags = 0;
		cd[i].minval = (cl_long)~0UL;			/* unsigned-ordered seeds */
		cd[i].maxval = 0;
		if (off > 0xffffffffUL)
		{
			*p_errcode = StromError_DataStoreOutOfRange;
			return nullptr;
		}
	}
	/*
	 * the heap area behind the column arrays: the source's datums fit the source (its 'length'
	 * bounds their sum), plus up to 3 bytes per datum for the 4-byte boundaries they start on.
	 * 'usage' is the writer's cursor; ingest_finish leaves 'length' at the bytes really used.
	 */
	size_t	heap_off = off;
	if (nheapcols > 0)
	{
		if (format == KDS_FORMAT_TUPSLOT)
		{
			*p_errcode = StromError_BadRequestMessage;		/* a TUPSLOT chunk holds no datum bytes */
			return nullptr;
		}
		fn_main = fn_heap;
		off += STROM_TYPEALIGN(KDS_COLUMN_ALIGN, (size_t)head->length + 4 * (size_t)nitems * nheapcols + 4);
		if (off > 0xffffffffUL)
		{
			*p_errcode = StromError_DataStoreOutOfRange;
			return nullptr;
		}
		for (int i = 0; i < ncols; i++)
			if (head->colmeta[i].attlen == -1)
				cd[i].extra_off = (cl_uint)heap_off;
	}
	head->hostptr = 0;
	head->length = (cl_uint)off;
	head->usage = (cl_uint)(nheapcols > 0 ? heap_off : 0);
	head->nitems = nitems;
	head->nrooms = nitems;
	head->nblocks = 0;
	head->maxblocks = 0;
	head->format = KDS_FORMAT_COLUMN;

	size_t	aux_len = sizeof(cl_int) * (2 * (size_t)ncols + 1);	/* type oids, NULL flags, failure flag */
	char   *d_dst = (char *)dev->pool.alloc(off);
	char   *d_aux = (char *)dev->pool.alloc(aux_len);
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	strom_dstore *result = nullptr;
	std::vector<cl_int> aux(2 * (size_t)ncols + 1, 0);
	if (type_oids)
		memcpy(aux.data(), type_oids, sizeof(cl_int) * ncols);
	do {
		if (!d_dst || !d_aux)
		{
			*p_errcode = StromError_OutOfMemory;
			break;
		}
		if (hipMemcpyAsync(d_dst, hbuf.data(), hbuf.size(), hipMemcpyHostToDevice, stream) != hipSuccess ||
			hipMemcpyAsync(d_aux, aux.data(), aux_len, hipMemcpyHostToDevice, stream) != hipSuccess ||
			hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess)
		{
			*p_errcode = StromError_HipInternal;
			break;
		}
		const void *a_src = src->devptr;
		void	   *a_dst = d_dst;
		const void *a_oids = (type_oids ? d_aux : nullptr);
		void	   *a_flags = d_aux + sizeof(cl_int) * ncols;
		void	   *args_main[] = { &a_src, &a_dst, &a_oids, &a_flags };
		void	   *args_fin[] = { &a_dst, &a_oids, &a_flags };
		unsigned	nwg = (unsigned)std::min<size_t>(((size_t)nitems + 255) / 256,
													 (size_t)dev->prop.multiProcessorCount * 8);
		(void)hipEventRecord(ev0, stream);
		if (nwg > 0 &&
			hipModuleLaunchKernel(fn_main, nwg, 1, 1, 256, 1, 1, 0, stream, args_main, nullptr) != hipSuccess)
		{
			*p_errcode = StromError_HipInternal;
			break;
		}
		/* zone maps from the transposed columns, one grid row per column */
		void	   *args_mm[] = { &a_dst, &a_oids };
		unsigned	mmgrid = (unsigned)std::min<size_t>(((size_t)nitems + 255) / 256,
														(size_t)dev->prop.multiProcessorCount);	/* one work-group per CU and column: see ingest_minmax */
		if (type_oids && mmgrid > 0 &&
			hipModuleLaunchKernel(fn_mm, mmgrid, (unsigned)ncols, 1, 256, 1, 1, 0, stream,
								  args_mm, nullptr) != hipSuccess)
		{
			*p_errcode = StromError_HipInternal;
			break;
		}
		if (hipModuleLaunchKernel(fn_fin, 1, 1, 1, 64, 1, 1, 0, stream, args_fin, nullptr) != hipSuccess)
		{
			*p_errcode = StromError_HipInternal;
			break;
		}
		(void)hipEventRecord(ev1, stream);
		result = new strom_dstore{d_dst, off, src->dindex, true, {}};
		cl_int	failed = 0;
		if (hipMemcpyAsync(&result->head, d_dst, offsetof(kern_data_store, colmeta),
						   hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipMemcpyAsync(&failed, d_aux + sizeof(cl_int) * 2 * (size_t)ncols, sizeof(cl_int),
						   hipMemcpyDeviceToHost, stream) != hipSuccess ||
			hipStreamSynchronize(stream) != hipSuccess)
		{
			delete result;
			result = nullptr;
			*p_errcode = StromError_HipInternal;
			break;
		}
		if (failed)
		{
			/* a numeric beyond the 64-bit device form: this chunk stays in
			 * its row format (the kernels re-check such rows one by one) */
			delete result;
			result = nullptr;
			*p_errcode = StromError_CpuReCheck;
			break;
		}
		if (p_kern_ns)
		{
			float ms = 0;
			if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess)
				*p_kern_ns = (uint64_t)((double)ms * 1e6);
		}
	} while (0);
	if (ev0) (void)hipEventDestroy(ev0);
	if (ev1) (void)hipEventDestroy(ev1);
	if (d_aux) dev->pool.release(d_aux);
	if (!result && d_dst) dev->pool.release(d_dst);
	return result;
	STROM_ABI_CATCH(nullptr, p_errcode)
}
