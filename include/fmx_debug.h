/* fmx_debug.h -- the two diagnostic exports of libfmx.  Not part of the drop-in boundary (include/fmx.h): nothing in the reference corresponds
 * to them; bench.py and tools/ use them. */
#ifndef FMX_DEBUG_H
#define FMX_DEBUG_H
#include "fmx.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Streaming bandwidth of GPU `device` in GB/s over `bytes` (>= 1 MiB) of float2 data, the mean of `iters` launches timed with HIP events.
 * mode 0: copy (bytes read + written are counted); mode 1: read 12, write 1 -- the input-filter stage's traffic shape; mode 2: the same reads, nothing written (SURVEY 8d asks for the
 * measured device bandwidth beside the nominal 8 TB/s: bench.py `roofline.measured_stream_bandwidth`). */
int fmx_debug_stream_bandwidth(int32_t device, int32_t mode, int64_t bytes, int32_t iters, double *gbps);
/* Per-phase shader-cycle counters of front_kernel (and, in -DSB_FINE_TICKS builds, of stage B), summed over the handle's channels into out[96]
 * (may be null); enable != 0 allocates and clears the counters, 0 frees them.  tools/front_phases.py, tools/stageb_fine.py. */
int fmx_debug_phase_cycles(fmx_handle h, int32_t enable, unsigned long long *out);
#ifdef __cplusplus
}
#endif
#endif
