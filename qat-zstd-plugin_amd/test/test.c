/*
 * test.c — round-trip demo/test of the drop-in sequence producer (BASELINE config #1).
 *
 * Counterpart of the reference's test/test.c (/root/reference/test/test.c:53-146): compress one
 * file through ZSTD_compress2 with qatSequenceProducer registered and
 * ZSTD_c_enableSeqProducerFallback = 1, decompress, compare.  Written from scratch with the
 * two fixes SURVEY.md §4 calls for: parameter results are checked with ZSTD_isError (the
 * reference relies on setParameter returning the value set), and the exit code is non-zero on
 * any failure (the reference always returns 0).  Without a GPU the device start fails, the
 * producer reports an error per block and libzstd falls back to its own match-finder — the
 * test still passes, exactly like the reference without QAT hardware.
 *
 *   usage: test <file> [level]       prints sizes, "plugin blocks"/"fallback" and PASS/FAIL
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qatseqprod.h"

static unsigned char *readFile(const char *name, size_t *size)
{
    FILE *f = fopen(name, "rb");
    unsigned char *buf = NULL;
    long n;
    if (!f) return NULL;
    if (fseek(f, 0, SEEK_END) == 0 && (n = ftell(f)) >= 0 && fseek(f, 0, SEEK_SET) == 0) {
        buf = (unsigned char *)malloc((size_t)n + 1);
        if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
        *size = (size_t)n;
    }
    fclose(f);
    return buf;
}

int main(int argc, char *argv[])
{
    size_t srcSize = 0, dstCap, cSize, dSize, r;
    unsigned char *src = NULL, *dst = NULL, *back = NULL;
    ZSTD_CCtx *zc = NULL;
    void *state = NULL;
    int level = 1, status, rc = 1;

    if (argc < 2) { fprintf(stderr, "usage: %s <file> [level 1-12]\n", argv[0]); return 2; }
    if (argc > 2) level = atoi(argv[2]);
    src = readFile(argv[1], &srcSize);
    if (!src) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }

    zc = ZSTD_createCCtx();
    status = QZSTD_startQatDevice(); /* QZSTD_OK with a GPU, QZSTD_FAIL without: both are fine here */
    state = QZSTD_createSeqProdState();
    dstCap = ZSTD_compressBound(srcSize);
    dst = (unsigned char *)malloc(dstCap ? dstCap : 1);
    back = (unsigned char *)malloc(srcSize ? srcSize : 1);
    if (!zc || !state || !dst || !back) { fprintf(stderr, "allocation failed\n"); goto done; }

    ZSTD_registerSequenceProducer(zc, state, qatSequenceProducer);
    r = ZSTD_CCtx_setParameter(zc, ZSTD_c_enableSeqProducerFallback, 1);
    if (ZSTD_isError(r)) { fprintf(stderr, "enableSeqProducerFallback: %s\n", ZSTD_getErrorName(r)); goto done; }
    r = ZSTD_CCtx_setParameter(zc, ZSTD_c_compressionLevel, level);
    if (ZSTD_isError(r)) { fprintf(stderr, "compressionLevel: %s\n", ZSTD_getErrorName(r)); goto done; }

    cSize = ZSTD_compress2(zc, dst, dstCap, src, srcSize);
    if (ZSTD_isError(cSize)) { fprintf(stderr, "ZSTD_compress2: %s\n", ZSTD_getErrorName(cSize)); goto done; }
    dSize = ZSTD_decompress(back, srcSize, dst, cSize);
    if (ZSTD_isError(dSize) || dSize != srcSize) { fprintf(stderr, "decompressed size differs\n"); goto done; }
    if (memcmp(back, src, srcSize) != 0) { fprintf(stderr, "ERROR: input and round-trip buffers differ\n"); goto done; }

    printf("plugin %s, device status %d (%s)\n", QZSTD_version(), status,
           status == QZSTD_OK ? "GPU offload" : "no device: libzstd software fallback");
    printf("Source size: %zu\nCompressed size: %zu\nPASS\n", srcSize, cSize);
    rc = 0;
done:
    if (rc) printf("FAIL\n");
    ZSTD_freeCCtx(zc);
    QZSTD_freeSeqProdState(state);
    QZSTD_stopQatDevice();
    free(src);
    free(dst);
    free(back);
    return rc;
}
