/* ffq_count.c -- the C ABI of libffq_hip.so from plain C: count the records of a FASTQ file (and the bytes of their
 * sequences) through the stream front end, the way the reference's benchmark loop does with readfastq_iter
 * (/root/reference/src/demo/benchmark.py:10-23: one pass, entries and bytes counted).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/ffq_count.c -o ffq_count -Lfastq-and-furious_amd/csrc -lffq_hip \
 *       -Wl,-rpath,$PWD/fastq-and-furious_amd/csrc
 *   ./ffq_count reads.fq            (needs a gfx950 device: there is no CPU fallback)
 *
 * tests/test_abi.py compiles and links it (CPU), tests/test_stream.py runs it against the oracle's count (GPU).
 */
#include <fcntl.h>
#include <inttypes.h>
#include <stdio.h>
#include <unistd.h>

#include "ffq.h"

int main(int argc, char **argv)
{
    if (argc == 2 && argv[1][0] == '-' && argv[1][1] == 'v') {      /* no device needed */
        printf("ffq abi %d build %s\n", ffq_abi_version(), ffq_build_id());
        return ffq_abi_version() == FFQ_ABI_VERSION ? 0 : 1;
    }
    if (argc != 2) { fprintf(stderr, "usage: %s file.fq | -v\n", argv[0]); return 2; }
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror(argv[1]); return 2; }
    ffq_ctx *ctx = NULL;
    ffq_stream *st = NULL;
    if (ffq_ctx_create(0, &ctx) != FFQ_OK || ffq_stream_open(ctx, fd, (int64_t)16 << 20, &st) != FFQ_OK) {
        fprintf(stderr, "ffq: %s\n", ffq_last_error());
        return 1;
    }
    int64_t records = 0, bases = 0;
    int end_state = FFQ_END_REFILL, rc = FFQ_OK;
    while (end_state == FFQ_END_REFILL) {
        const int64_t *rows = NULL;
        int64_t n = 0, err_offset = -1;
        rc = ffq_stream_next(st, &rows, &n, &end_state, &err_offset, NULL, NULL, NULL);
        if (rc != FFQ_OK) { fprintf(stderr, "ffq: %s\n", ffq_last_error()); break; }
        for (int64_t i = 0; i < n; i++) bases += rows[6 * i + 3] - rows[6 * i + 2];     /* pos3 - pos2 */
        records += n;
        if (end_state != FFQ_END_REFILL && end_state != FFQ_END_OK)
            fprintf(stderr, "stream error %d at byte %" PRId64 " (the reference raises ValueError here)\n", end_state, err_offset);
    }
    printf("%" PRId64 " records, %" PRId64 " bases\n", records, bases);
    ffq_stream_close(st);
    ffq_ctx_destroy(ctx);
    close(fd);
    return rc == FFQ_OK && end_state == FFQ_END_OK ? 0 : 1;
}
