/*
 * strom_rowmap.h -- GpuScan results -> kern_row_map, in place, on the device
 *
 * The reference chains operators through pgstrom_bulkslot {pds, nvalids,
 * rindex[]} (pg_strom.h:323-329; gpuscan.c:1318-1446 builds it on the host
 * from kern_resultbuf).  Here the selected row ids never leave HBM: the
 * +(i+1) entries of results[] become the 0-based rindex[] of a kern_row_map
 * whose nvalids word is the 4 bytes in front of results[] (the flag bytes of
 * kern_resultbuf), so the next operator's kernel reads the very same buffer.
 * A negative entry (row to re-check on the CPU) cannot be chained: the
 * status word reports it and the caller falls back to the host path.
 */
#ifndef STROM_ROWMAP_DEVICE_H
#define STROM_ROWMAP_DEVICE_H

extern "C" __global__ void
__launch_bounds__(256)
rowmap_from_results(kern_resultbuf *kresults, cl_uint nitems, cl_int *status)
{
	cl_int	   *rindex = kresults->results;
	bool		bad = false;

	for (cl_uint i = blockIdx.x * blockDim.x + threadIdx.x;
		 i < nitems;
		 i += gridDim.x * blockDim.x)
	{
		cl_int	v = rindex[i];
		if (v <= 0)
			bad = true;
		else
			rindex[i] = v - 1;
	}
	if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0)
		atomicMax(status, StromError_CpuReCheck);
	if (blockIdx.x == 0 && threadIdx.x == 0)
		*(cl_int *)((char *)kresults + offsetof(kern_resultbuf, results) - sizeof(cl_int)) = (cl_int)nitems;
}

#endif	/* STROM_ROWMAP_DEVICE_H */
