/*
 * qzstd_frontend.c — batch front-end for the host entropy stage (include/qzstd_frontend.h).
 *
 * A persistent pool of workers, each with its own ZSTD_CCtx + producer state (the reference's threading model:
 * one CCtx per thread, /root/reference/test/benchmark.c:241, :514-516), fed from one shared chunk cursor.  A worker
 * keeps several claims: the one it is entropy-coding and the next two (QZSTD_FRONT_AHEAD), already announced to the GPUs
 * (QZSTD_hintSource), so match-finding always runs ahead of the thread that will consume it.
 */
#include "qzstd_frontend.h"

#include "qatseqprod.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

extern int zstdshim_ok(void) __attribute__((weak)); /* tools/zstdshim, when that is the libzstd in the process; else absent */

#define QF_NONE ((size_t)-1)
#define QF_HINT_MAX ((size_t)16 << 20)
#define QF_AHEAD_MAX 3u

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
    size_t srcSize, nChunks;
    unsigned char *dst;
    size_t *sizes;
    size_t nextChunk; /* shared claim cursor (atomic) */
    int uniform;      /* every claim a whole segment: the levels whose match-finding, not the entropy stage, sets the pace (see qfClaim) */
    unsigned ahead;   /* claims a worker keeps announced beyond the one it is entropy-coding: $QZSTD_FRONT_AHEAD, 1..3, default 2 */
    unsigned long served[2];
};

/* A claim: chunks [c0, c1) of the job.  Where the entropy stage sets the pace (levels 1-4: with a libzstd that entropy-codes 1.6 GB/s per core
 * a 512 MiB job is over in 25 ms and its first and last milliseconds count) claims are not all the same size:
 *   - a worker's first claims are small (an eighth, then a quarter, then half of a segment): the GPU's first results are back after
 *     the time one block takes, not after a whole segment was staged and queued, and the pipeline below fills while they are consumed;
 *   - towards the end of the job a claim is at most a (2 x threads)-th of what is left, so the workers finish together;
 *   - never less than QF_MIN_CHUNKS chunks (a launch of fewer than four blocks is not worth its queueing).
 * Where the match-finder sets the pace (the chain levels, 5-12: a 128 KiB block takes a workgroup 4-5 ms at level 6) every claim is a whole
 * segment and three are kept announced: what counts there is how many blocks are on the GPU at any time, and a launch of four blocks
 * takes as long as one of thirty-two (level 6, 18 threads, 4 MiB: 5.8 GB/s with the varying claims, 8.3 with uniform ones; the kernel
 * alone, input resident: 10.5). */
#define QF_MIN_CHUNKS 4u
typedef struct { size_t c0, c1; } QF_Seg;

static int qfClaim(QZSTD_Front *f, unsigned nth, QF_Seg *out)
{
    size_t want = f->segChunks, have, left;
    have = __atomic_load_n(&f->nextChunk, __ATOMIC_RELAXED);
    if (have >= f->nChunks) return 0;
    if (!f->uniform) {
        if (nth < 3 && (f->segChunks >> (3 - nth)) >= QF_MIN_CHUNKS) want = f->segChunks >> (3 - nth);
        left = (f->nChunks - have) / (2u * (size_t)f->p.nThreads);
        if (want > left) want = left;
    }
    if (want < QF_MIN_CHUNKS) want = QF_MIN_CHUNKS;
    if (want > f->segChunks) want = f->segChunks;
    have = __atomic_fetch_add(&f->nextChunk, want, __ATOMIC_RELAXED);
    if (have >= f->nChunks) return 0;
    out->c0 = have;
    out->c1 = have + want < f->nChunks ? have + want : f->nChunks;
    return 1;
}

static void qfAnnounce(QZSTD_Front *f, QF_Worker *w, const QF_Seg *sg)
{
    const size_t off = sg->c0 * f->p.chunkSize;
    size_t len = (sg->c1 - sg->c0) * f->p.chunkSize, grid = f->p.chunkSize;
    if (!f->p.useProducer) return;
    if (len > f->srcSize - off) len = f->srcSize - off;
    /* the block grid of the announcement: the chunk when a chunk is one block; 128 KiB blocks inside bigger frames */
    if (grid > 131072) grid = 131072;
    if ((grid & 15) || len > QF_HINT_MAX) return; /* not announceable: the callbacks take the per-block path */
    /* every chunk is its own frame whose blocks start at the chunk's start: the announcement's grid (anchored at the
     * claim's start) only names those blocks if the chunks are whole grid cells — otherwise every callback would miss
     * and the GPU would match-find everything twice (round-2 ADVICE) */
    if (f->p.chunkSize > grid && f->p.chunkSize % grid != 0) return;
    /* (stable: the source is the const argument of the QZSTD_frontCompress call in progress) */
    (void)QZSTD_hintSourceEx(w->state, f->src + off, len, grid, f->p.level, QZSTD_HINT_STABLE);
}

static int qfCompressSegment(QZSTD_Front *f, QF_Worker *w, const QF_Seg *sg)
{
    size_t c;
    for (c = sg->c0; c < sg->c1; c++) {
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
        QF_Seg q[QF_AHEAD_MAX + 1]; /* claimed and announced, oldest first: q[0] is the one being entropy-coded */
        unsigned n = 0, claims = 0;
        int bad = 0, more = 1;
        pthread_mutex_lock(&f->mu);
        while (!f->quit && f->gen == seen) pthread_cond_wait(&f->cvWork, &f->mu);
        if (f->quit) { pthread_mutex_unlock(&f->mu); break; }
        seen = f->gen;
        pthread_mutex_unlock(&f->mu);

        for (;;) {
            /* keep `ahead` claims announced beyond the current one: the GPUs match-find them while this thread entropy-codes q[0]
             * (a state holds four announcements: three ahead at most) */
            while (more && n < f->ahead + 1u) {
                more = qfClaim(f, claims, &q[n]);
                if (!more) break;
                claims++;
                if (!bad) qfAnnounce(f, w, &q[n]);
                n++;
            }
            if (n == 0) break;
            if (!bad && qfCompressSegment(f, w, &q[0]) != 0) bad = 1;
            memmove(&q[0], &q[1], (n - 1) * sizeof(q[0]));
            n--;
        }
        /* the job is over for this worker: whatever it announced ends here.  An announcement otherwise lives until the callback of its
         * last block — which never comes for a last block below 7 bytes (libzstd does not ask the producer), after a failed part, or
         * after `bad` — and a STABLE one would serve the NEXT job by address if the caller used the same buffer again (round-4 ADVICE) */
        if (f->p.useProducer && w->state) QZSTD_dropHints(w->state);
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
    /* the libzstd this was linked against must be one that works: when it is tools/zstdshim (tests and bench only) and its look-up of
     * libarrow.so's copy failed, every ZSTD_* call would abort() the process — refuse instead; and the producer API needs >= 1.5.4 */
    if (zstdshim_ok && zstdshim_ok() != 1) return NULL;
    if (ZSTD_versionNumber() < 10504u) return NULL;
    f = (QZSTD_Front *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->p = *p;
    /* measured on MI355X + 16 cores: 2 MiB where the entropy stage sets the pace (levels 1-4), 4 MiB claims of one size where the match-finder does */
    seg = p->segmentBytes ? p->segmentBytes : ((size_t)(p->level >= 5 ? 4 : 2) << 20);
    /* ... and at the chain levels at most 64 chunks per claim (round 5): a claim is one launch, its blocks one workgroup each — level 12 on 32 KiB
     * chunks: 2 MiB claims 8.7 GB/s, 4 MiB 7.7, 8 MiB 6.3 (the kernel alone: 9.8); level 6 on 128 KiB chunks keeps its 4 MiB (32 blocks): 10.3-11.0 */
    if (!p->segmentBytes && p->level >= 5 && seg / p->chunkSize > 64u) seg = 64u * p->chunkSize;
    if (seg > QF_HINT_MAX) seg = QF_HINT_MAX;
    f->segChunks = seg / p->chunkSize ? seg / p->chunkSize : 1;
    f->stride = ZSTD_compressBound(p->chunkSize);
    {
        const char *a = getenv("QZSTD_FRONT_AHEAD");
        const int v = a && *a ? atoi(a) : 2;
        f->ahead = v < 1 ? 1u : (v > (int)QF_AHEAD_MAX ? QF_AHEAD_MAX : (unsigned)v);
        a = getenv("QZSTD_FRONT_UNIFORM");
        f->uniform = a && *a ? atoi(a) != 0 : p->level >= 5;
        if (f->uniform && !(getenv("QZSTD_FRONT_AHEAD") && *getenv("QZSTD_FRONT_AHEAD"))) f->ahead = QF_AHEAD_MAX;
    }
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
    f->nextChunk = 0;
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
