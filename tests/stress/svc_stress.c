/*
 * svc_stress.c — TEST INFRASTRUCTURE.  Real threads (not Python's, which take turns under the interpreter lock) against the
 * resident service and the callback path of lib/libqatseqprod.so, every result compared with the ORACLE (oracle/qzstd_oracle.c is
 * linked into this test binary; the product never sees it).  Round-3 verdict, weak 2: the chain-level shared scratch (cross-workgroup
 * flags, write-through entries) and the self-certifying results only show their bug classes under load.
 *
 *   svc_stress items  <corpus> <level> <threads> <reps> [blockBytes]
 *       every thread owns the buffers of one service slot (what a slot of the plugin holds) and submits `reps` requests through
 *       qzstd_hip_service_submit, blocks rotating; EVERY work item of every request is compared, sequence for sequence, with
 *       qzo_find_sequences_from over the block up to the item's end.  Reference shape: many DC instances polled by their own
 *       threads, /root/reference/src/qatseqprod.c:905-928, :1243-1272.
 *   svc_stress frames <corpus> <levels, e.g. 6,12> <threads> <reps> [chunkBytes]
 *       the drop-in path: every thread has a ZSTD_CCtx + producer state, thread t compresses at levels[t % n]: callers of SEVERAL
 *       levels share one GPU (one level's workers resident, the others through the batches); every frame must equal the frame
 *       libzstd builds from the oracle's sequences; prints who served (QZSTD_failStats / QZSTD_deviceStats).
 * Output: one line "svc_stress ok: ..." and exit code 0, or the first differences and exit code 1.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qatseqprod.h"
#include "qzstd_hip.h"
#include "qzstd_oracle.h"

#define MAX_ITEMS QZSTD_HIP_SVC_MAX_ITEMS
#define ITEM_CAP 1371u
#define MAX_BLOCKS 16
#define MAX_THREADS 64

static unsigned char *gData;
static size_t gLen, gBlock;
static int gNBlocks;
static int gReps, gThreads;
static pthread_barrier_t gBar;
static volatile int gBad;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static size_t block_len(int b)
{
    const size_t o = (size_t)b * gBlock;
    size_t n = gLen - o < gBlock ? gLen - o : gBlock;
    if (b == 1 && n > 30000) n -= 27071; /* one ragged block: its last item is short and its size no multiple of anything */
    return n;
}

/* ---------------------------------------------------------------- items ---------------- */
typedef struct {
    uint32_t n;
    qzo_seq_t *s;
} ItemWant;
static ItemWant gWant[MAX_BLOCKS][MAX_ITEMS];
static uint32_t gItems[MAX_BLOCKS], gItemBytes[MAX_BLOCKS];
static int gLevel;

typedef struct {
    int t;
    unsigned long requests, items, refused, lateEntries;
} ItemThread;

static void *items_thread(void *arg)
{
    ItemThread *th = (ItemThread *)arg;
    unsigned char *hSrc = (unsigned char *)qzstd_hip_host_alloc_coherent(QZSTD_HIP_BLOCK_MAX + 64);
    qzo_seq_t *hSeqs = (qzo_seq_t *)qzstd_hip_host_alloc_coherent((size_t)MAX_ITEMS * ITEM_CAP * 16u);
    uint32_t *hCount = (uint32_t *)qzstd_hip_host_alloc_coherent(MAX_ITEMS * 4u);
    void *dSrc = qzstd_hip_malloc(0, QZSTD_HIP_BLOCK_MAX + 64);
    void *dWork = qzstd_hip_workspace_bytes(gLevel, 1, QZSTD_HIP_BLOCK_MAX) ? qzstd_hip_malloc(0, QZSTD_HIP_SVC_WORK_BYTES) : NULL;
    uint32_t epoch = 0;
    int rep;
    if (!hSrc || !hSeqs || !hCount || !dSrc) { fprintf(stderr, "thread %d: buffers: %s\n", th->t, qzstd_hip_last_error()); gBad = 1; }
    pthread_barrier_wait(&gBar); /* (allocations and frees stop a resident service: all of them before the first request) */
    for (rep = 0; rep < gReps && !gBad; rep++) {
        const int b = (th->t + rep) % gNBlocks;
        const size_t n = block_len(b);
        const uint32_t nIt = gItems[b], cap = MAX_ITEMS * ITEM_CAP / nIt;
        qzstd_hip_svc_req_t rq;
        uint32_t k;
        int rc;
        double t0;
        memcpy(hSrc, gData + (size_t)b * gBlock, n);
        memset(hSrc + n, 0, 16);
        for (k = 0; k < nIt; k++) __atomic_store_n(&hCount[k], 0u, __ATOMIC_RELAXED);
        epoch = epoch % 0xFFFFFFu + 1u;
        rq.hSrc = hSrc; rq.dSrc = dSrc; rq.hSeqs = hSeqs; rq.hCount = hCount;
        rq.srcLen = (uint32_t)n; rq.itemBytes = gItemBytes[b]; rq.nItems = nIt; rq.seqCapPerItem = cap;
        rq.slot = (uint32_t)(200 + th->t); rq.epoch = epoch; rq.dWork = dWork;
        rc = qzstd_hip_service_submit(0, gLevel, &rq);
        if (rc == 1) { th->refused++; rep--; { const struct timespec nap = { 0, 200000 }; nanosleep(&nap, NULL); } if (th->refused > 100000) gBad = 1; continue; }
        if (rc != 0) { fprintf(stderr, "thread %d: submit: %s\n", th->t, qzstd_hip_last_error()); gBad = 1; break; }
        th->requests++;
        t0 = now_s();
        for (k = 0; k < nIt && !gBad; k++) {
            uint32_t cnt, j, polls = 0;
            const ItemWant *w = &gWant[b][k];
            const qzo_seq_t *q = hSeqs + (size_t)k * cap;
            while ((cnt = __atomic_load_n(&hCount[k], __ATOMIC_ACQUIRE)) == 0u) {
                if ((++polls & 1023u) == 0u) {
                    if (now_s() - t0 > 10.0) { fprintf(stderr, "thread %d rep %d block %d item %u: no count after 10 s\n", th->t, rep, b, k); gBad = 1; break; }
                    (void)qzstd_hip_service_poke(0, gLevel);
                }
            }
            if (gBad) break;
            if (cnt != w->n) {
                fprintf(stderr, "thread %d rep %d block %d item %u of %u: count %u, oracle %u\n", th->t, rep, b, k, nIt, cnt, w->n);
                gBad = 1;
                break;
            }
            for (j = 0; j < cnt; j++) { /* an entry is there when it shows the request's epoch (include/qzstd_hip.h) */
                uint32_t spins = 0;
                while (__atomic_load_n(&q[j].rep, __ATOMIC_ACQUIRE) != epoch) {
                    if (spins++ == 0) th->lateEntries++;
                    if ((spins & 0xFFFFFu) == 0u && now_s() - t0 > 10.0) { fprintf(stderr, "thread %d: entry %u of item %u never arrived\n", th->t, j, k); gBad = 1; break; }
                }
                if (gBad) break;
                if (q[j].offset != w->s[j].offset || q[j].litLength != w->s[j].litLength || q[j].matchLength != w->s[j].matchLength) {
                    fprintf(stderr, "thread %d rep %d block %d item %u sequence %u of %u: {%u,%u,%u}, oracle {%u,%u,%u}\n", th->t, rep, b, k, j, cnt,
                            q[j].offset, q[j].litLength, q[j].matchLength, w->s[j].offset, w->s[j].litLength, w->s[j].matchLength);
                    gBad = 1;
                    break;
                }
            }
            th->items++;
        }
    }
    pthread_barrier_wait(&gBar); /* (frees stop the service: only after everybody's last request) */
    qzstd_hip_host_free(hSrc); qzstd_hip_host_free(hSeqs); qzstd_hip_host_free(hCount);
    qzstd_hip_free(0, dSrc);
    if (dWork) qzstd_hip_free(0, dWork);
    return NULL;
}

static int run_items(int level)
{
    pthread_t tid[MAX_THREADS];
    ItemThread th[MAX_THREADS];
    qzo_profile_t pf;
    unsigned long requests = 0, items = 0, refused = 0, late = 0, info[8];
    int b, t;
    double t0;
    gLevel = level;
    if (qzo_profile_for_level(level, gBlock, &pf) != 0) return 2;
    for (b = 0; b < gNBlocks; b++) { /* what every item has to look like: the oracle, once */
        const size_t n = block_len(b);
        size_t item = 4096;
        uint32_t k;
        while ((n + item - 1) / item > MAX_ITEMS) item *= 2;
        gItemBytes[b] = (uint32_t)item;
        gItems[b] = (uint32_t)((n + item - 1) / item);
        for (k = 0; k < gItems[b]; k++) {
            const size_t upTo = (k + 1) * item < n ? (k + 1) * item : n;
            const size_t cap = MAX_ITEMS * ITEM_CAP / gItems[b];
            qzo_seq_t *s = (qzo_seq_t *)malloc(cap * sizeof(qzo_seq_t));
            const size_t cnt = qzo_find_sequences_from(&pf, gData + (size_t)b * gBlock, upTo, k * item, s, cap);
            if (!s || cnt == QZO_ERROR) { fprintf(stderr, "oracle: block %d item %u does not fit\n", b, k); return 2; }
            gWant[b][k].n = (uint32_t)cnt;
            gWant[b][k].s = s;
        }
    }
    pthread_barrier_init(&gBar, NULL, (unsigned)gThreads);
    t0 = now_s();
    for (t = 0; t < gThreads; t++) { memset(&th[t], 0, sizeof(th[t])); th[t].t = t; pthread_create(&tid[t], NULL, items_thread, &th[t]); }
    for (t = 0; t < gThreads; t++) { pthread_join(tid[t], NULL); requests += th[t].requests; items += th[t].items; refused += th[t].refused; late += th[t].lateEntries; }
    (void)qzstd_hip_service_info(0, info);
    (void)qzstd_hip_service_stop(0);
    if (gBad) return 1;
    printf("svc_stress ok: items, level %#x, %d threads x %d requests = %lu requests, %lu work items bit-exact vs the oracle, %lu entries arrived after "
           "their count, %lu submits refused; service: %lu launch(es), %lu item(s) gave up on a slice, broken %lu; %.2f s\n",
           (unsigned)level, gThreads, gReps, requests, items, late, refused, info[0], info[6], info[3], now_s() - t0);
    return info[3] != 0 || info[6] != 0;
}

/* ---------------------------------------------------------------- frames ---------------- */
static int gLevels[8], gNLevels;
static unsigned char *gFrame[8][MAX_BLOCKS];
static size_t gFrameLen[8][MAX_BLOCKS];

typedef struct {
    int t;
    unsigned long frames, fs[8], hs[4];
} FrameThread;

static size_t frame_of(ZSTD_CCtx *zc, int b, unsigned char *dst, size_t cap)
{
    return ZSTD_compress2(zc, dst, cap, gData + (size_t)b * gBlock, block_len(b));
}

static void *frames_thread(void *arg)
{
    FrameThread *th = (FrameThread *)arg;
    const int li = th->t % gNLevels, level = gLevels[li];
    void *st = QZSTD_createSeqProdState();
    ZSTD_CCtx *zc = ZSTD_createCCtx();
    const size_t cap = ZSTD_compressBound(gBlock);
    unsigned char *dst = (unsigned char *)malloc(cap);
    int rep;
    if (!st || !zc || !dst) { gBad = 1; }
    else {
        ZSTD_registerSequenceProducer(zc, st, qatSequenceProducer);
        (void)ZSTD_CCtx_setParameter(zc, ZSTD_c_compressionLevel, level);
        (void)ZSTD_CCtx_setParameter(zc, ZSTD_c_validateSequences, 1);
        (void)ZSTD_CCtx_setParameter(zc, ZSTD_c_enableSeqProducerFallback, 0);
    }
    pthread_barrier_wait(&gBar);
    for (rep = 0; rep < gReps && !gBad; rep++) {
        const int b = (th->t + rep) % gNBlocks;
        const size_t r = frame_of(zc, b, dst, cap);
        if (ZSTD_isError(r)) { fprintf(stderr, "thread %d level %d rep %d: %s\n", th->t, level, rep, ZSTD_getErrorName(r)); gBad = 1; break; }
        if (r != gFrameLen[li][b] || memcmp(dst, gFrame[li][b], r) != 0) {
            fprintf(stderr, "thread %d level %d rep %d block %d: frame of %zu bytes differs from libzstd + oracle (%zu bytes)\n", th->t, level, rep, b, r, gFrameLen[li][b]);
            gBad = 1;
            break;
        }
        th->frames++;
    }
    if (st) { QZSTD_failStats(st, th->fs); QZSTD_hintStats(st, th->hs); }
    ZSTD_freeCCtx(zc);
    QZSTD_freeSeqProdState(st);
    free(dst);
    return NULL;
}

static int run_frames(void)
{
    pthread_t tid[MAX_THREADS];
    FrameThread th[MAX_THREADS];
    unsigned long frames = 0, errors = 0, service = 0, redone = 0, dev[4] = { 0, 0, 0, 0 };
    int li, b, t;
    double t0;
    const size_t cap = ZSTD_compressBound(gBlock);
    for (li = 0; li < gNLevels; li++) { /* the frames libzstd builds from the oracle's sequences */
        ZSTD_CCtx *zo = ZSTD_createCCtx();
        ZSTD_registerSequenceProducer(zo, NULL, (ZSTD_sequenceProducer_F)qzo_sequence_producer);
        (void)ZSTD_CCtx_setParameter(zo, ZSTD_c_compressionLevel, gLevels[li]);
        (void)ZSTD_CCtx_setParameter(zo, ZSTD_c_validateSequences, 1);
        (void)ZSTD_CCtx_setParameter(zo, ZSTD_c_enableSeqProducerFallback, 0);
        for (b = 0; b < gNBlocks; b++) {
            gFrame[li][b] = (unsigned char *)malloc(cap);
            gFrameLen[li][b] = frame_of(zo, b, gFrame[li][b], cap);
            if (ZSTD_isError(gFrameLen[li][b])) { fprintf(stderr, "oracle frame: %s\n", ZSTD_getErrorName(gFrameLen[li][b])); return 2; }
        }
        ZSTD_freeCCtx(zo);
    }
    if (QZSTD_startQatDevice() != QZSTD_OK) { fprintf(stderr, "QZSTD_startQatDevice failed\n"); return 3; }
    pthread_barrier_init(&gBar, NULL, (unsigned)gThreads);
    t0 = now_s();
    for (t = 0; t < gThreads; t++) { memset(&th[t], 0, sizeof(th[t])); th[t].t = t; pthread_create(&tid[t], NULL, frames_thread, &th[t]); }
    for (t = 0; t < gThreads; t++) {
        pthread_join(tid[t], NULL);
        frames += th[t].frames; errors += th[t].fs[0]; service += th[t].fs[7]; redone += th[t].fs[6];
    }
    (void)QZSTD_deviceStats(0, dev);
    QZSTD_stopQatDevice();
    if (gBad) return 1;
    printf("svc_stress ok: frames, %d level(s) on one GPU, %d threads x %d frames = %lu frames identical to libzstd + oracle; producer errors %lu, "
           "served by the resident service %lu, redone %lu; device 0: announced %lu, batches %lu, service %lu; %.2f s\n",
           gNLevels, gThreads, gReps, frames, errors, service, redone, dev[0], dev[1], dev[2], now_s() - t0);
    return errors != 0;
}

int main(int argc, char **argv)
{
    FILE *f;
    if (argc < 6) { fprintf(stderr, "usage: svc_stress items|frames <corpus> <level[,level..]> <threads> <reps> [block bytes]\n"); return 2; }
    gThreads = atoi(argv[4]);
    gReps = atoi(argv[5]);
    gBlock = argc > 6 ? (size_t)atol(argv[6]) : 131072;
    if (gThreads < 1 || gThreads > MAX_THREADS || gReps < 1 || gBlock < 4096 || gBlock > QZSTD_HIP_BLOCK_MAX) return 2;
    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    gData = (unsigned char *)malloc((size_t)MAX_BLOCKS * gBlock + 64);
    gLen = fread(gData, 1, (size_t)MAX_BLOCKS * gBlock, f);
    fclose(f);
    gNBlocks = (int)((gLen + gBlock - 1) / gBlock);
    if (gNBlocks < 2) { fprintf(stderr, "corpus too small\n"); return 2; }
    memset(gData + gLen, 0, 64);
    if (strcmp(argv[1], "items") == 0) {
        if (qzstd_hip_device_count() <= 0) { fprintf(stderr, "no device: %s\n", qzstd_hip_last_error()); return 3; }
        return run_items((int)strtol(argv[3], NULL, 0));
    }
    {
        char *p = argv[3];
        while (*p && gNLevels < 8) { gLevels[gNLevels++] = (int)strtol(p, &p, 0); if (*p == ',') p++; }
    }
    return run_frames();
}
