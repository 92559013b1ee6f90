/*
 * qzstd_frontend.c — batch front-end for the host entropy stage (include/qzstd_frontend.h).
 *
 * A persistent pool of workers, each with its own ZSTD_CCtx + producer state (the reference's threading model:
 * one CCtx per thread, /root/reference/test/benchmark.c:241, :514-516), fed from one shared segment counter.  A worker
 * keeps two segments claimed: the one it is entropy-coding and the next one, already announced to the GPUs
 * (QZSTD_hintSource), so match-finding always runs one segment ahead of the thread that will consume it.
 */
#include "qzstd_frontend.h"

#include "qatseqprod.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define QF_NONE ((size_t)-1)
#define QF_HINT_MAX ((size_t)16 << 20)

typedef struct {
    QZSTD_Front *front;
    pthread_t th;
    ZSTD_CCtx *zc;
    void *state;
    int ok;
} QF_Worker;

struct QZSTD_Front_s {
    QZSTD_FrontParams p;
    size_t stride, segChunks;
    QF_Worker *w;
    pthread_mutex_t mu;
    pthread_cond_t cvWork, cvDone;
    /* the job in progress */
    unsigned long gen; /* bumped per job */
    int quit, running, failed;
    const unsigned char *src;
    size_t srcSize, nChunks, nSegs;
    unsigned char *dst;
    size_t *sizes;
    size_t nextSeg; /* shared claim counter (atomic) */
    unsigned long served[2];
};

static size_t qfClaim(QZSTD_Front *f)
{
    const size_t s = __atomic_fetch_add(&f->nextSeg, 1, __ATOMIC_RELAXED);
    return s < f->nSegs ? s : QF_NONE;
}

static void qfAnnounce(QZSTD_Front *f, QF_Worker *w, size_t seg)
{
    const size_t off = seg * f->segChunks * f->p.chunkSize;
    size_t len = f->segChunks * f->p.chunkSize, grid = f->p.chunkSize;
    if (!f->p.useProducer || seg == QF_NONE) return;
    if (len > f->srcSize - off) len = f->srcSize - off;
    /* the block grid of the announcement: the chunk when a chunk is one block; 128 KiB blocks inside bigger frames */
    if (grid > 131072) grid = 131072;
    if ((grid & 15) || len > QF_HINT_MAX) return; /* not announceable: the callbacks take the per-block path */
    /* every chunk is its own frame whose blocks start at the chunk's start: the announcement's grid (anchored at the
     * segment's start) only names those blocks if the chunks are whole grid cells — otherwise every callback would miss
     * and the GPU would match-find everything twice (round-2 ADVICE) */
    if (f->p.chunkSize > grid && f->p.chunkSize % grid != 0) return;
    (void)QZSTD_hintSource(w->state, f->src + off, len, grid, f->p.level);
}

static int qfCompressSegment(QZSTD_Front *f, QF_Worker *w, size_t seg)
{
    size_t c = seg * f->segChunks;
    const size_t cEnd = c + f->segChunks < f->nChunks ? c + f->segChunks : f->nChunks;
    for (; c < cEnd; c++) {
        const size_t off = c * f->p.chunkSize;
        const size_t n = f->srcSize - off < f->p.chunkSize ? f->srcSize - off : f->p.chunkSize;
        const size_t r = ZSTD_compress2(w->zc, f->dst + c * f->stride, f->stride, f->src + off, n);
        if (ZSTD_isError(r)) return -1;
        f->sizes[c] = r;
    }
    return 0;
}

static void *qfWorker(void *arg)
{
    QF_Worker *w = (QF_Worker *)arg;
    QZSTD_Front *f = w->front;
    unsigned long seen = 0;
    for (;;) {
        size_t cur, nxt;
        int bad = 0;
        pthread_mutex_lock(&f->mu);
        while (!f->quit && f->gen == seen) pthread_cond_wait(&f->cvWork, &f->mu);
        if (f->quit) { pthread_mutex_unlock(&f->mu); break; }
        seen = f->gen;
        pthread_mutex_unlock(&f->mu);

        cur = qfClaim(f);
        qfAnnounce(f, w, cur);
        while (cur != QF_NONE) {
            nxt = qfClaim(f);
            qfAnnounce(f, w, nxt); /* the GPUs work on the next segment while this thread entropy-codes the current one */
            if (!bad && qfCompressSegment(f, w, cur) != 0) bad = 1;
            cur = nxt;
        }
        pthread_mutex_lock(&f->mu);
        if (bad) f->failed = 1;
        if (--f->running == 0) pthread_cond_signal(&f->cvDone);
        pthread_mutex_unlock(&f->mu);
    }
    return NULL;
}

QZSTD_Front *QZSTD_createFront(const QZSTD_FrontParams *p)
{
    QZSTD_Front *f;
    int t, made = 0;
    size_t seg;
    if (!p || p->nThreads < 1 || p->nThreads > 1024 || p->level < 1 || p->level > 12 || p->chunkSize == 0) return NULL;
    f = (QZSTD_Front *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->p = *p;
    seg = p->segmentBytes ? p->segmentBytes : ((size_t)2 << 20); /* measured best on MI355X + 16 cores: 2 MiB */
    if (seg > QF_HINT_MAX) seg = QF_HINT_MAX;
    f->segChunks = seg / p->chunkSize ? seg / p->chunkSize : 1;
    f->stride = ZSTD_compressBound(p->chunkSize);
    f->w = (QF_Worker *)calloc((size_t)p->nThreads, sizeof(QF_Worker));
    pthread_mutex_init(&f->mu, NULL);
    pthread_cond_init(&f->cvWork, NULL);
    pthread_cond_init(&f->cvDone, NULL);
    if (!f->w) { QZSTD_freeFront(f); return NULL; }
    if (p->useProducer) (void)QZSTD_startQatDevice(); /* return value ignored, as the reference's callers do: fallback below */
    for (t = 0; t < p->nThreads; t++) {
        QF_Worker *w = &f->w[t];
        const int e = p->extRepcodes == 1 ? ZSTD_ps_enable : (p->extRepcodes == 2 ? ZSTD_ps_disable : ZSTD_ps_auto);
        w->front = f;
        w->zc = ZSTD_createCCtx();
        if (!w->zc) break;
        if (p->useProducer) {
            w->state = QZSTD_createSeqProdState();
            if (!w->state) break;
            ZSTD_registerSequenceProducer(w->zc, w->state, qatSequenceProducer);
            /* any producer error (device down, time-out, dense block) falls back to libzstd's own match-finder */
            (void)ZSTD_CCtx_setParameter(w->zc, ZSTD_c_enableSeqProducerFallback, 1);
        }
        if (ZSTD_isError(ZSTD_CCtx_setParameter(w->zc, ZSTD_c_compressionLevel, p->level)) ||
            ZSTD_isError(ZSTD_CCtx_setParameter(w->zc, ZSTD_c_searchForExternalRepcodes, e)))
            break;
        if (pthread_create(&w->th, NULL, qfWorker, w) != 0) break;
        w->ok = 1;
        made++;
    }
    if (made != p->nThreads) { QZSTD_freeFront(f); return NULL; }
    return f;
}

size_t QZSTD_frontFrameStride(const QZSTD_Front *f) { return f ? f->stride : 0; }

size_t QZSTD_frontCompress(QZSTD_Front *f, const void *src, size_t srcSize, void *dst, size_t dstCapacity, size_t *frameSizes)
{
    size_t nChunks;
    int failed;
    if (!f || (!src && srcSize) || !dst || !frameSizes) return (size_t)-1;
    nChunks = (srcSize + f->p.chunkSize - 1) / f->p.chunkSize;
    if (nChunks == 0) return 0;
    if (dstCapacity / f->stride < nChunks) return (size_t)-1;
    pthread_mutex_lock(&f->mu);
    f->src = (const unsigned char *)src;
    f->srcSize = srcSize;
    f->dst = (unsigned char *)dst;
    f->sizes = frameSizes;
    f->nChunks = nChunks;
    f->nSegs = (nChunks + f->segChunks - 1) / f->segChunks;
    f->nextSeg = 0;
    f->failed = 0;
    f->running = f->p.nThreads;
    f->gen++;
    pthread_cond_broadcast(&f->cvWork);
    while (f->running) pthread_cond_wait(&f->cvDone, &f->mu);
    failed = f->failed;
    pthread_mutex_unlock(&f->mu);
    return failed ? (size_t)-1 : nChunks;
}

size_t QZSTD_frontCompact(const QZSTD_Front *f, void *dst, const size_t *frameSizes, size_t nFrames)
{
    unsigned char *d = (unsigned char *)dst;
    size_t c, pos = 0;
    if (!f || !dst || !frameSizes) return 0;
    for (c = 0; c < nFrames; c++) {
        if (pos != c * f->stride) memmove(d + pos, d + c * f->stride, frameSizes[c]);
        pos += frameSizes[c];
    }
    return pos;
}

void QZSTD_frontStats(QZSTD_Front *f, unsigned long stats[2])
{
    int t;
    if (!stats) return;
    stats[0] = stats[1] = 0;
    if (!f || !f->p.useProducer) return;
    for (t = 0; t < f->p.nThreads; t++) {
        unsigned long s[4] = { 0, 0, 0, 0 };
        if (f->w[t].state) QZSTD_hintStats(f->w[t].state, s);
        stats[0] += s[0];
        stats[1] += s[1];
    }
}

void QZSTD_frontFailStats(QZSTD_Front *f, unsigned long stats[8])
{
    int t, k;
    if (!stats) return;
    for (k = 0; k < 8; k++) stats[k] = 0;
    if (!f || !f->p.useProducer) return;
    for (t = 0; t < f->p.nThreads; t++) {
        unsigned long s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (f->w[t].state) QZSTD_failStats(f->w[t].state, s);
        for (k = 0; k < 8; k++) stats[k] += s[k];
    }
}

void QZSTD_freeFront(QZSTD_Front *f)
{
    int t;
    if (!f) return;
    pthread_mutex_lock(&f->mu);
    f->quit = 1;
    pthread_cond_broadcast(&f->cvWork);
    pthread_mutex_unlock(&f->mu);
    for (t = 0; f->w && t < f->p.nThreads; t++) {
        QF_Worker *w = &f->w[t];
        if (w->ok) pthread_join(w->th, NULL);
        if (w->zc) ZSTD_freeCCtx(w->zc);
        if (w->state) QZSTD_freeSeqProdState(w->state);
    }
    free(f->w);
    pthread_mutex_destroy(&f->mu);
    pthread_cond_destroy(&f->cvWork);
    pthread_cond_destroy(&f->cvDone);
    free(f);
}
