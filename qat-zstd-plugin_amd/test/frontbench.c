/*
 * frontbench.c — wall-clock throughput of the batch front-end (include/qzstd_frontend.h) on one file.
 *
 * The caller shape is the reference's benchmark (/root/reference/test/benchmark.c:222-382: chunks, one frame per
 * chunk, whole-buffer decompress + memcmp as the PASS criterion), but ONE buffer is shared by all workers and the figure
 * is bytes / wall clock of the whole call — what a service that hands a batch of buffers to the plugin sees.
 *
 *   frontbench [-t threads] [-l loops] [-c chunk] [-L level] [-E extRepcodes] [-s segmentMiB] [-m 0|1] file
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qzstd_frontend.h"
#include "qatseqprod.h"

static double nowS(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + (double)t.tv_nsec / 1e9;
}

static size_t parseSize(const char *s)
{
    char *end;
    unsigned long v = strtoul(s, &end, 10);
    if (*end == 'K' || *end == 'k') v <<= 10;
    else if (*end == 'M' || *end == 'm') v <<= 20;
    return (size_t)v;
}

int main(int argc, char **argv)
{
    QZSTD_FrontParams p = { 16, 1, 131072, (size_t)2 << 20, 0, 1 };
    unsigned loops = 3;
    const char *file = NULL;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') { file = a; continue; }
        const char *v = a[2] ? a + 2 : (i + 1 < argc ? argv[++i] : "");
        switch (a[1]) {
        case 't': p.nThreads = atoi(v); break;
        case 'l': loops = (unsigned)atoi(v); break;
        case 'c': p.chunkSize = parseSize(v); break;
        case 'L': p.level = atoi(v); break;
        case 'E': p.extRepcodes = atoi(v); break;
        case 's': p.segmentBytes = (size_t)atoi(v) << 20; break;
        case 'm': p.useProducer = atoi(v); break;
        default: fprintf(stderr, "usage: %s [-t threads] [-l loops] [-c chunk] [-L level] [-E 0|1|2] [-s segMiB] [-m 0|1] file\n", argv[0]); return 1;
        }
    }
    if (!file || loops < 1) { fprintf(stderr, "no input file\n"); return 1; }
    FILE *fp = fopen(file, "rb");
    if (!fp) { fprintf(stderr, "cannot open %s\n", file); return 1; }
    fseek(fp, 0, SEEK_END);
    const size_t n = (size_t)ftell(fp);
    fseek(fp, 0, SEEK_SET);
    unsigned char *src = (unsigned char *)malloc(n ? n : 1);
    if (!src || fread(src, 1, n, fp) != n) { fprintf(stderr, "cannot read %s\n", file); return 1; }
    fclose(fp);

    QZSTD_Front *f = QZSTD_createFront(&p);
    if (!f) { fprintf(stderr, "cannot create the front-end\n"); return 1; }
    const size_t nChunks = (n + p.chunkSize - 1) / p.chunkSize, stride = QZSTD_frontFrameStride(f);
    unsigned char *dst = (unsigned char *)malloc(nChunks * stride + 1);
    size_t *sizes = (size_t *)calloc(nChunks + 1, sizeof(size_t));
    unsigned char *back = (unsigned char *)malloc(n ? n : 1);
    if (!dst || !sizes || !back) { fprintf(stderr, "out of memory\n"); return 1; }
    memset(dst, 0, nChunks * stride); /* touch the pages before timing */

    double best = 0, sum = 0, rate[256];
    unsigned nr = 0;
    int ok = 1;
    for (unsigned l = 0; l < loops + 1 && ok; l++) { /* the first pass warms buffers, streams and pinned memory */
        const double t0 = nowS();
        const size_t r = QZSTD_frontCompress(f, src, n, dst, nChunks * stride, sizes);
        const double dt = nowS() - t0;
        if (r != nChunks) { ok = 0; break; }
        if (l == 0) continue;
        sum += dt;
        if (best == 0 || dt < best) best = dt;
        if (nr < 256 && dt > 0) rate[nr++] = (double)n / 1e6 / dt;
    }
    size_t csize = 0;
    ZSTD_DCtx *zd = ZSTD_createDCtx();
    for (size_t c = 0; c < nChunks && ok; c++) {
        const size_t off = c * p.chunkSize, len = n - off < p.chunkSize ? n - off : p.chunkSize;
        const size_t r = ZSTD_decompressDCtx(zd, back + off, len, dst + c * stride, sizes[c]);
        if (ZSTD_isError(r) || r != len) ok = 0;
        csize += sizes[c];
    }
    if (ok && memcmp(back, src, n) != 0) ok = 0;
    ZSTD_freeDCtx(zd);
    unsigned long st[2], fl[8];
    QZSTD_frontStats(f, st);
    QZSTD_frontFailStats(f, fl);
    for (unsigned i = 1; i < nr; i++) /* insertion sort: median / min / max of the passes */
        for (unsigned j = i; j > 0 && rate[j - 1] > rate[j]; j--) { const double x = rate[j]; rate[j] = rate[j - 1]; rate[j - 1] = x; }
    QZSTD_freeFront(f);
    char perGpu[1024] = "";
    if (p.useProducer) { /* who did the work: blocks per GPU (announcement ranges / batches / resident service) */
        const int nd = QZSTD_deviceStats(0, NULL);
        size_t o = 0;
        for (int d = 0; d < nd && o + 64 < sizeof perGpu; d++) {
            unsigned long ds[4];
            (void)QZSTD_deviceStats(d, ds);
            o += (size_t)snprintf(perGpu + o, sizeof perGpu - o, "%sgpu%d %lu/%lu/%lu", d ? ", " : "", d, ds[0], ds[1], ds[2]);
        }
    }
    if (p.useProducer) QZSTD_stopQatDevice();
    printf("frontbench libzstd %s mode %d level %d chunk %zu threads %d segment %zu: %zu -> %zu bytes, wall-clock %.1f MB/s "
           "(mean of %u passes; best %.1f MB/s), %lu block(s) from announcements, %lu per block, %s\n",
           ZSTD_versionString(), p.useProducer, p.level, p.chunkSize, p.nThreads, p.segmentBytes, n, csize,
           sum > 0 ? (double)n * loops / 1e6 / sum : 0.0, loops, best > 0 ? (double)n / 1e6 / best : 0.0, st[0], st[1],
           ok ? "PASS" : "FAIL");
    if (nr) printf("passes MB/s: median %.1f min %.1f max %.1f\n", nr & 1 ? rate[nr / 2] : 0.5 * (rate[nr / 2 - 1] + rate[nr / 2]), rate[0], rate[nr - 1]);
    if (p.useProducer) printf("blocks per GPU (announced/batched/service): %s\n", perGpu);
    printf("producer errors: %lu (guards %lu, device down %lu, time-outs %lu, capacity %lu, runtime %lu) - blocks compressed by libzstd's "
           "own match-finder instead; dense blocks redone alone: %lu\n", fl[0], fl[1], fl[2], fl[3], fl[4], fl[5], fl[6]);
    free(src); free(dst); free(sizes); free(back);
    return ok ? 0 : 1;
}
