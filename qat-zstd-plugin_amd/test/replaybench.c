/*
 * replaybench.c — the ceiling of ANY external sequence producer on this host with this libzstd.
 *
 * Behind ZSTD_registerSequenceProducer() everything but the match-finder stays on the calling thread.  How fast can
 * ZSTD_compress2 get when the producer costs nothing?  This tool records the sequences the real plugin returns for
 * every block of a file once, then times ZSTD_compress2 with a producer that only copies the recorded sequences out
 * (a memcpy per block), from T threads over one shared buffer — chunks claimed from a shared counter, one frame per
 * chunk, the caller shape of the reference's benchmark (/root/reference/test/benchmark.c:300-326).  The figure is the
 * Amdahl ceiling the GPU producer is measured against in bench.py (`e2e_ceiling_replay`).
 *
 *   replaybench [-t threads] [-l loops] [-c chunk] [-L level] [-E extRepcodes] file
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qatseqprod.h"

#define BLOCK_MAX 131072u

typedef struct {
    ZSTD_Sequence *seqs; /* NULL: the plugin answered with an error for this block (libzstd's fallback took it) */
    size_t n;
} Rec;

typedef struct {
    const unsigned char *base;
    size_t grain; /* bytes between recorded block starts: min(chunk, 128 KiB) */
    size_t nRec;
    Rec *rec;
    void *plugin; /* recording pass only: the real producer's state */
} Table;

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

static Rec *slotOf(Table *t, const void *src)
{
    const size_t off = (size_t)((const unsigned char *)src - t->base);
    if ((const unsigned char *)src < t->base || off % t->grain != 0 || off / t->grain >= t->nRec) return NULL;
    return &t->rec[off / t->grain];
}

/* pass 1: the real plugin, its answer kept */
static size_t recordProducer(void *state, ZSTD_Sequence *out, size_t cap, const void *src, size_t srcSize,
                             const void *dict, size_t dictSize, int level, size_t windowSize)
{
    Table *t = (Table *)state;
    const size_t r = qatSequenceProducer(t->plugin, out, cap, src, srcSize, dict, dictSize, level, windowSize);
    Rec *rec = slotOf(t, src);
    if (rec && r != ZSTD_SEQUENCE_PRODUCER_ERROR && !rec->seqs) {
        rec->seqs = (ZSTD_Sequence *)malloc((r ? r : 1) * sizeof(ZSTD_Sequence));
        if (rec->seqs) { memcpy(rec->seqs, out, r * sizeof(ZSTD_Sequence)); rec->n = r; }
    }
    return r;
}

/* pass 2: a producer that costs one memcpy */
static size_t replayProducer(void *state, ZSTD_Sequence *out, size_t cap, const void *src, size_t srcSize,
                             const void *dict, size_t dictSize, int level, size_t windowSize)
{
    Table *t = (Table *)state;
    const Rec *rec = slotOf(t, src);
    (void)srcSize; (void)dict; (void)dictSize; (void)level; (void)windowSize;
    if (!rec || !rec->seqs || rec->n > cap) return ZSTD_SEQUENCE_PRODUCER_ERROR;
    memcpy(out, rec->seqs, rec->n * sizeof(ZSTD_Sequence));
    return rec->n;
}

typedef struct {
    Table *tab;
    const unsigned char *src;
    size_t n, chunk, nChunks, stride;
    unsigned char *dst;
    size_t *sizes;
    int level, extRep;
    unsigned loops;
    volatile size_t *next; /* per pass: the shared chunk counter */
    pthread_barrier_t *bar;
    int failed;
} Job;

static ZSTD_CCtx *makeCCtx(int level, int extRep, ZSTD_sequenceProducer_F f, void *state)
{
    ZSTD_CCtx *c = ZSTD_createCCtx();
    if (!c) return NULL;
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_enableSeqProducerFallback, 1);
    if (extRep) ZSTD_CCtx_setParameter(c, ZSTD_c_searchForExternalRepcodes, extRep);
    ZSTD_registerSequenceProducer(c, state, f);
    return c;
}

static void *worker(void *arg)
{
    Job *j = (Job *)arg;
    ZSTD_CCtx *c = makeCCtx(j->level, j->extRep, replayProducer, j->tab);
    if (!c) j->failed = 1;
    for (unsigned l = 0; l < j->loops + 1; l++) {
        pthread_barrier_wait(j->bar); /* start of pass l */
        for (;;) {
            const size_t k = __sync_fetch_and_add(&j->next[l], 1);
            if (k >= j->nChunks || !c) break;
            const size_t o = k * j->chunk, len = j->n - o < j->chunk ? j->n - o : j->chunk;
            const size_t r = ZSTD_compress2(c, j->dst + k * j->stride, j->stride, j->src + o, len);
            if (ZSTD_isError(r)) { j->failed = 1; break; }
            j->sizes[k] = r;
        }
        pthread_barrier_wait(j->bar); /* end of pass l */
    }
    if (c) ZSTD_freeCCtx(c);
    return NULL;
}

int main(int argc, char **argv)
{
    int threads = 16, level = 1, extRep = 0;
    unsigned loops = 3;
    size_t chunk = 131072;
    const char *file = NULL;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') { file = a; continue; }
        const char *v = a[2] ? a + 2 : (i + 1 < argc ? argv[++i] : "");
        switch (a[1]) {
        case 't': threads = atoi(v); break;
        case 'l': loops = (unsigned)atoi(v); break;
        case 'c': chunk = parseSize(v); break;
        case 'L': level = atoi(v); break;
        case 'E': extRep = atoi(v); break;
        default: fprintf(stderr, "usage: %s [-t threads] [-l loops] [-c chunk] [-L level] [-E 0|1|2] file\n", argv[0]); return 1;
        }
    }
    if (!file || loops < 1 || threads < 1 || threads > 1024 || chunk == 0 || (chunk > BLOCK_MAX && chunk % BLOCK_MAX)) {
        fprintf(stderr, "need a file, loops >= 1, 1..1024 threads, and a chunk <= 128K or a multiple of it\n");
        return 1;
    }
    FILE *fp = fopen(file, "rb");
    if (!fp) { fprintf(stderr, "cannot open %s\n", file); return 1; }
    fseek(fp, 0, SEEK_END);
    const size_t n = (size_t)ftell(fp);
    fseek(fp, 0, SEEK_SET);
    unsigned char *src = (unsigned char *)malloc(n ? n : 1);
    if (!src || fread(src, 1, n, fp) != n) { fprintf(stderr, "cannot read %s\n", file); return 1; }
    fclose(fp);

    const size_t nChunks = (n + chunk - 1) / chunk, stride = ZSTD_compressBound(chunk);
    Table tab;
    tab.base = src;
    tab.grain = chunk < BLOCK_MAX ? chunk : BLOCK_MAX;
    tab.nRec = (n + tab.grain - 1) / tab.grain;
    tab.rec = (Rec *)calloc(tab.nRec + 1, sizeof(Rec));
    unsigned char *dst = (unsigned char *)malloc(nChunks * stride + 1);
    size_t *sizes = (size_t *)calloc(nChunks + 1, sizeof(size_t));
    unsigned char *back = (unsigned char *)malloc(chunk);
    if (!tab.rec || !dst || !sizes || !back) { fprintf(stderr, "out of memory\n"); return 1; }
    memset(dst, 0, nChunks * stride);

    /* ---- pass 1: record what the real producer returns ---- */
    QZSTD_startQatDevice();
    tab.plugin = QZSTD_createSeqProdState();
    ZSTD_CCtx *rc = makeCCtx(level, extRep, recordProducer, &tab);
    if (!tab.plugin || !rc) { fprintf(stderr, "cannot set up the plugin\n"); return 1; }
    for (size_t k = 0; k < nChunks; k++) {
        const size_t o = k * chunk, len = n - o < chunk ? n - o : chunk;
        if (ZSTD_isError(ZSTD_compress2(rc, dst + k * stride, stride, src + o, len))) { fprintf(stderr, "recording pass failed\n"); return 1; }
    }
    ZSTD_freeCCtx(rc);
    QZSTD_freeSeqProdState(tab.plugin);
    QZSTD_stopQatDevice();
    size_t recorded = 0;
    for (size_t i = 0; i < tab.nRec; i++) recorded += tab.rec[i].seqs != NULL;

    /* ---- pass 2: replay from T threads ---- */
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads + 1u);
    volatile size_t *next = (volatile size_t *)calloc(loops + 1, sizeof(size_t));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    Job *jobs = (Job *)calloc((size_t)threads, sizeof(Job));
    if (!next || !th || !jobs) { fprintf(stderr, "out of memory\n"); return 1; }
    for (int t = 0; t < threads; t++) {
        Job j = { &tab, src, n, chunk, nChunks, stride, dst, sizes, level, extRep, loops, next, &bar, 0 };
        jobs[t] = j;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    double best = 0, sum = 0, rate[256];
    unsigned nr = 0;
    for (unsigned l = 0; l < loops + 1; l++) { /* the first pass warms the contexts */
        pthread_barrier_wait(&bar);
        const double t0 = nowS();
        pthread_barrier_wait(&bar);
        const double mbps = (double)n / 1e6 / (nowS() - t0);
        if (l == 0) continue;
        sum += mbps;
        if (mbps > best) best = mbps;
        if (nr < 256) rate[nr++] = mbps;
    }
    for (unsigned i = 1; i < nr; i++) /* insertion sort: median / min / max of the passes */
        for (unsigned j = i; j > 0 && rate[j - 1] > rate[j]; j--) { const double x = rate[j]; rate[j] = rate[j - 1]; rate[j - 1] = x; }
    int ok = 1;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (jobs[t].failed) ok = 0; }

    /* ---- verify: every frame decompresses to its chunk ---- */
    size_t csize = 0;
    for (size_t k = 0; k < nChunks && ok; k++) {
        const size_t o = k * chunk, len = n - o < chunk ? n - o : chunk;
        const size_t r = ZSTD_decompress(back, chunk, dst + k * stride, sizes[k]);
        if (ZSTD_isError(r) || r != len || memcmp(back, src + o, len) != 0) ok = 0;
        csize += sizes[k];
    }
    printf("replay: %d threads, level %d, chunk %zu, %zu bytes, %zu of %zu blocks recorded, csize %zu, "
           "%.1f MB/s wall (best pass %.1f), round trip %s\n",
           threads, level, chunk, n, recorded, tab.nRec, csize, sum / loops, best, ok ? "PASS" : "FAIL");
    if (nr) printf("passes MB/s: median %.1f min %.1f max %.1f\n", nr & 1 ? rate[nr / 2] : 0.5 * (rate[nr / 2 - 1] + rate[nr / 2]), rate[0], rate[nr - 1]);
    return ok ? 0 : 1;
}
