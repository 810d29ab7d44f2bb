/*
 * ffq_probe.h -- entry points that exist ONLY in the instrumented build of the library,
 * libffq_probe.so (the same sources compiled with -DFFQ_PROBES; fastq-and-furious_amd/build.py
 * build_probe()).  Nothing here is part of the drop-in boundary (include/ffq.h) and nothing in the
 * product loads that library: it serves the scripts under tools/ and the `hbm_read_probe` figure of bench.py.
 * The instrumented build also honours the ablation switches FFQ_ABLATE / FFQ_K1_ABLATE /
 * FFQ_DQ_ABLATE / FFQ_PROF / FFQ_DEBUG (environment), which the product build does not compile.
 */
#ifndef FFQ_PROBE_H
#define FFQ_PROBE_H

#include "ffq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measured streaming-read ceiling of the device in the scan kernel's launch geometry
 * (mode 0) or as a grid-stride loop (mode 1): average ms over `reps` launches of a
 * kernel that only reads n_bytes (rounded down to 16 KiB); mode 6: mode 0 with non-temporal
 * loads.  Modes 2 / 3 / 4: the line-index kernel itself without the rest of a scan, launched
 * back to back / between its own pair of events / the same on a buffer with a ragged last
 * tile.  Mode 7 and 100+: the index kernel with a decoupled look-back riding along; 200+: the
 * persistent streaming loop with a lagged two-level prefix (tools/lookback_probe.py,
 * tools/pipe_probe.py).                                                                    */
int ffq_read_probe(ffq_ctx *ctx, const uint8_t *d_buf, int64_t n_bytes, int mode, int reps,
                   float *ms_avg);

#ifdef __cplusplus
}
#endif
#endif /* FFQ_PROBE_H */
