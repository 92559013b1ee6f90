/*
 * qatseqprod.c — host side of the MI355X-native sequence producer, plain C.
 *
 * Mirrors the operator interface of intel/QAT-ZSTD-Plugin for its hot path
 * (/root/reference/src/qatseqprod.c) — same entry points, same guards, same error
 * behaviour — over the thin HIP C ABI of include/qzstd_hip.h:
 *
 *   reference                                      here
 *   ---------------------------------------------  -----------------------------------------
 *   QZSTD_startQatDevice   :948-964                runtime probe + slot table, under a mutex
 *   instance discovery + round-robin shuffle       slots interleaved across GPUs so that
 *     :529-663                                       consecutive slots sit on different devices
 *   QZSTD_grabInstance / releaseInstance :905-933  test-and-set sweep starting at the hint (hint path,
 *                                                    QZSTD_HIP_COALESCE=0); by default callers are
 *                                                    merged into one launch per tick per GPU (coalescer)
 *   QZSTD_allocInstMem (lazy)  :685-822            pinned + device buffers, created on first use
 *   input staging memcpy       :1222-1227          memcpy into the pinned staging buffer
 *   cpaDcCompressData2 + poll  :1243-1272          H2D, kernel launch, D2H on the slot's stream,
 *                                                    stream sync
 *   QZSTD_decLz4s              :1013-1091          (none: the kernel emits ZSTD_Sequence)
 *   result / capacity checks   :1293-1322          count == NSEQ_ERROR or >= cap-1 -> ERROR
 *   device-down counter, retry every 1000 blocks   same (failOffloadCnt)
 *     :88, :1140-1152
 *
 * No QAT / icp_sal / cpa symbol is used or emulated.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* process_vm_readv: the fault-safe read behind the transparent look-ahead */
#endif
#include "qatseqprod.h"
#include "qzstd_hip.h"

#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/prctl.h>
#include <sys/types.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>

#ifndef DEBUGLEVEL
#define DEBUGLEVEL 0
#endif

#define QZ_LEVEL_MIN 1
#define QZ_LEVEL_MAX 12
#define QZ_RETRY_INTERVAL_BLOCKS 1000 /* re-probe a dead device every N failed blocks */
#define QZ_GRAB_SWEEPS 40000 /* 64 yields, then 50 us naps: about two seconds */
#define QZ_MAX_DEVICES 64
#define QZ_MAX_SLOTS 1024
#define QZ_DEFAULT_SLOTS_PER_DEVICE 128
#define QZ_FIRST_COPY_SEQS 16384u /* sequences fetched together with the count */

static int qzLogLevel = DEBUGLEVEL; /* 0 silent, 1 errors, 2 events, 3 every sequence */
#define QZ_LOG(l, ...)                                       \
    do {                                                     \
        if ((l) <= qzLogLevel) {                             \
            fprintf(stderr, "qatseqprod(hip): " __VA_ARGS__); \
        }                                                    \
    } while (0)

/* One slot = one in-flight block on one GPU (the analogue of a QAT DC instance). */
typedef struct {
    int device;
    volatile int lock;
    int ready; /* buffers + stream exist */
    void *stream;
    unsigned char *hSrc; /* pinned staging, QZSTD_HIP_BLOCK_MAX + pad */
    unsigned char *dSrc;
    ZSTD_Sequence *hSeqs; /* pinned, seqCap entries */
    ZSTD_Sequence *dSeqs;
    qzstd_hip_block_t *hDesc; /* pinned */
    qzstd_hip_block_t *dDesc;
    unsigned int *hCount; /* pinned */
    unsigned int *dCount;
    size_t seqCap;
    /* grow-only buffers of the batched (hinted) path */
    unsigned char *dBatchSrc; size_t dBatchSrcCap;
    /* launch scratch (hash chains of levels >= 6), grow-only: one block / a hinted batch */
    void *dWork; size_t dWorkCap;
    void *dBatchWork; size_t dBatchWorkCap;
} QZSTD_Slot_T;

/*
 * Cross-thread request coalescing (one per GPU).  The producer API hands over ONE block per
 * call and waits, and a single block keeps one of 256 CUs busy for ~0.7 ms; many callers
 * (one CCtx per thread, the reference's own scaling model: README.md:138) are therefore merged
 * into one launch per tick: the first caller to find the device idle becomes the leader of the
 * open batch and runs it; callers arriving while it runs pile up in the other batch, whose first
 * member leads it when the device frees up ("group commit": no timers, no added latency for a
 * lone caller).  Every caller copies its own block into the batch's pinned staging area and its
 * own result out of it, so those copies run in parallel on the callers' threads.
 */
#define QZ_BATCH_MAX 64
typedef struct {
    const void *src;
    size_t srcSize, cap, rc;
} QZSTD_Req_T;

typedef struct {
    int state; /* 0 open (collecting), 1 running, 2 done (results being copied out) */
    int n, copied, consumed, level;
    QZSTD_Req_T req[QZ_BATCH_MAX];
    unsigned char *hSrc;      /* pinned, QZ_BATCH_MAX x QZ_SRC_STRIDE */
    ZSTD_Sequence *hSeqs;     /* pinned, QZ_BATCH_MAX x seqStride */
    qzstd_hip_block_t *hDesc; /* pinned */
    unsigned int *hCount;     /* pinned */
    void *dvSeqs, *dvDesc, *dvCount; /* device-side addresses of hSeqs / hDesc / hCount */
    pthread_cond_t cvLead;    /* the batch's leader (its first member) waits here: device idle / members staged */
    pthread_cond_t cvDone;    /* the other members wait here for the results */
} QZSTD_Batch_T;

typedef struct {
    int device, ready, running, open; /* open = index of the batch that accepts requests */
    pthread_mutex_t mu;
    pthread_cond_t cvOpen; /* callers that found no batch to join wait here */
    QZSTD_Batch_T batch[2];
    void *stream;
    unsigned char *dSrc;
    void *dWork; /* launch scratch, grow-only */
    size_t dWorkCap;
    size_t seqStride;
    unsigned long launches, blocks;
} QZSTD_Coalescer_T;
#define QZ_SRC_STRIDE ((size_t)QZSTD_HIP_BLOCK_MAX + 64)

typedef struct {
    int status; /* QZSTD_Status_e */
    int numDevices;
    int numSlots;
    QZSTD_Slot_T *slots;
    QZSTD_Coalescer_T *coal; /* one per device */
    int coalesce;            /* QZSTD_HIP_COALESCE (default 1) */
    int levelFlags;          /* QZSTD_HIP_LEVEL_REPCODES when QZSTD_HIP_EXT_REPCODES=1 */
    int lookahead;           /* transparent look-ahead: 0 off, 1 fault-safe read by process_vm_readv, 2 through a pipe */
    pthread_mutex_t mutex;
} QZSTD_Process_T;

static QZSTD_Process_T gProc = { QZSTD_FAIL, 0, 0, NULL, NULL, 1, 0, 0, PTHREAD_MUTEX_INITIALIZER };

/* One announced buffer: staged in pinned memory, match-found asynchronously on a slot's stream,
 * results (count + the first QZ_HINT_PITCH sequences of every block) copied back asynchronously. */
#define QZ_HINT_MAX_BYTES ((size_t)16 << 20)
#define QZ_HINT_PITCH ((size_t)16384) /* blocks with more sequences take the per-block path */
typedef struct {
    int st;   /* 0 empty, 1 in flight on the GPU (slot held), 2 ready */
    int slot; /* index of the slot held while in flight */
    const unsigned char *base;
    size_t size, block, nb;
    int level;
    unsigned char *hSrc;      /* pinned staging copy of the buffer */
    ZSTD_Sequence *hSeqs;     /* pinned, nb x QZ_HINT_PITCH */
    unsigned int *hCount;     /* pinned */
    qzstd_hip_block_t *hDesc; /* pinned */
    void *dvSeqs, *dvCount, *dvDesc; /* device-side addresses of the three: the kernel uses them directly */
    size_t hSrcCap, hSeqsCap, hCountCap, hDescCap; /* bytes */
} QZSTD_Hint_T;

/* Per-CCtx state (opaque to the caller). */
typedef struct {
    int slotHint;
    unsigned int failOffloadCnt;
    /* look-ahead batches served to later callbacks (QZSTD_hintSource): two, so that the GPU can
     * work on the next buffer while libzstd entropy-codes the current one on this thread */
    QZSTD_Hint_T hint[4]; /* [0..1] announced by the caller, [2..3] speculative (transparent look-ahead) */
    int hintNext, autoNext;
    unsigned autoDepth, autoBackoff, autoFails; /* blocks to speculate on, callbacks to sit out, misses in a row */
    int autoOutstanding;                        /* a guess was launched and nothing has been served from it yet */
    int pipeFd[2];                              /* the fault-safe read's pipe (mode 2), -1 = not opened */
    unsigned long autoLaunched, autoServed;
    unsigned long servedFromBatch, servedSync;
    unsigned long hintCalls, hintStageNs, hintQueueNs, hintWaitNs; /* event log only */
} QZSTD_Session_T;

#define QZ_AUTO_DEPTH_MIN 2u  /* transparent look-ahead: blocks guessed ahead, doubling while guesses are consumed */
#define QZ_AUTO_DEPTH_MAX 32u
static size_t qzSafeRead(void *dst, const void *src, size_t len, size_t block);
static void qzSpeculate(QZSTD_Session_T *s, const unsigned char *next, size_t blockSize, int compressionLevel);


const char *QZSTD_version(void)
{
    return QZSTD_VERSION;
}

/* ---------------------------------------------------------------- slots ---------- */

static void qzFreeSlot(QZSTD_Slot_T *s)
{
    if (s->stream) (void)qzstd_hip_stream_sync(s->device, s->stream);
    qzstd_hip_host_free(s->hSrc);
    qzstd_hip_host_free(s->hSeqs);
    qzstd_hip_host_free(s->hDesc);
    qzstd_hip_host_free(s->hCount);
    qzstd_hip_free(s->device, s->dSrc);
    qzstd_hip_free(s->device, s->dSeqs);
    qzstd_hip_free(s->device, s->dDesc);
    qzstd_hip_free(s->device, s->dCount);
    qzstd_hip_free(s->device, s->dBatchSrc);
    qzstd_hip_free(s->device, s->dWork);
    qzstd_hip_free(s->device, s->dBatchWork);
    if (s->stream) qzstd_hip_stream_destroy(s->device, s->stream);
    {
        const int dev = s->device;
        memset(s, 0, sizeof(*s));
        s->device = dev;
    }
}

/* lazy per-slot setup, first use only (reference: QZSTD_allocInstMem, :685-822) */
static int qzSetupSlot(QZSTD_Slot_T *s)
{
    if (s->ready) return QZSTD_OK;
    s->seqCap = qzstd_hip_sequence_bound(QZSTD_HIP_BLOCK_MAX);
    s->stream = qzstd_hip_stream_create(s->device);
    s->hSrc = (unsigned char *)qzstd_hip_host_alloc(QZSTD_HIP_BLOCK_MAX + 64);
    s->hSeqs = (ZSTD_Sequence *)qzstd_hip_host_alloc(s->seqCap * sizeof(ZSTD_Sequence));
    s->hDesc = (qzstd_hip_block_t *)qzstd_hip_host_alloc(sizeof(qzstd_hip_block_t));
    s->hCount = (unsigned int *)qzstd_hip_host_alloc(64);
    s->dSrc = (unsigned char *)qzstd_hip_malloc(s->device, QZSTD_HIP_BLOCK_MAX + 64);
    s->dSeqs = (ZSTD_Sequence *)qzstd_hip_malloc(s->device, s->seqCap * sizeof(ZSTD_Sequence));
    s->dDesc = (qzstd_hip_block_t *)qzstd_hip_malloc(s->device, sizeof(qzstd_hip_block_t));
    s->dCount = (unsigned int *)qzstd_hip_malloc(s->device, 64);
    if (!s->stream || !s->hSrc || !s->hSeqs || !s->hDesc || !s->hCount || !s->dSrc || !s->dSeqs ||
        !s->dDesc || !s->dCount) {
        QZ_LOG(1, "slot setup failed on device %d: %s\n", s->device, qzstd_hip_last_error());
        qzFreeSlot(s);
        return QZSTD_FAIL;
    }
    s->ready = 1;
    return QZSTD_OK;
}

/* test-and-set sweep over the slots, starting at the caller's sticky hint
 * (reference: QZSTD_grabInstance, :905-928) */
static int qzGrabSlot(int hint)
{
    int sweep, k;
    const int n = gProc.numSlots;
    if (n <= 0) return -1;
    if (hint < 0 || hint >= n) hint = 0;
    /* The reference sweeps its instances 10 times and then fails the block (src/qatseqprod.c:905-928, :915);
     * here a caller WAITS for a slot: short spins first, then 50 us naps, giving up only after about two seconds
     * (the reference's own time-out for a stuck request, :1261-1285) */
    for (sweep = 0; sweep < QZ_GRAB_SWEEPS; sweep++) {
        for (k = 0; k < n; k++) {
            const int i = (hint + k) % n;
            if (__sync_lock_test_and_set(&gProc.slots[i].lock, 1) == 0) return i;
        }
        if (sweep < 64) {
            sched_yield(); /* every slot busy: more threads than slots; let the holders finish */
        } else {
            const struct timespec nap = { 0, 50000 };
            nanosleep(&nap, NULL);
        }
    }
    return -1;
}

static void qzReleaseSlot(int i)
{
    __sync_lock_release(&gProc.slots[i].lock);
}

/* ---------------------------------------------------------------- coalescer ------ */

static void qzFreeCoalescer(QZSTD_Coalescer_T *c)
{
    int b;
    if (c->stream) (void)qzstd_hip_stream_sync(c->device, c->stream);
    for (b = 0; b < 2; b++) {
        qzstd_hip_host_free(c->batch[b].hSrc);
        qzstd_hip_host_free(c->batch[b].hSeqs);
        qzstd_hip_host_free(c->batch[b].hDesc);
        qzstd_hip_host_free(c->batch[b].hCount);
    }
    qzstd_hip_free(c->device, c->dSrc);
    qzstd_hip_free(c->device, c->dWork);
    if (c->stream) qzstd_hip_stream_destroy(c->device, c->stream);
    QZ_LOG(2, "device %d: %lu block(s) in %lu coalesced launch(es)\n", c->device, c->blocks, c->launches);
    pthread_mutex_destroy(&c->mu);
    pthread_cond_destroy(&c->cvOpen);
    for (b = 0; b < 2; b++) {
        pthread_cond_destroy(&c->batch[b].cvLead);
        pthread_cond_destroy(&c->batch[b].cvDone);
    }
}

/* lazy, under c->mu */
static int qzSetupCoalescer(QZSTD_Coalescer_T *c)
{
    int b, ok = 1;
    if (c->ready) return QZSTD_OK;
    c->seqStride = qzstd_hip_sequence_bound(QZSTD_HIP_BLOCK_MAX);
    c->stream = qzstd_hip_stream_create(c->device);
    c->dSrc = (unsigned char *)qzstd_hip_malloc(c->device, QZ_BATCH_MAX * QZ_SRC_STRIDE);
    ok = c->stream && c->dSrc;
    for (b = 0; b < 2 && ok; b++) {
        QZSTD_Batch_T *bt = &c->batch[b];
        bt->hSrc = (unsigned char *)qzstd_hip_host_alloc(QZ_BATCH_MAX * QZ_SRC_STRIDE);
        bt->hSeqs = (ZSTD_Sequence *)qzstd_hip_host_alloc(QZ_BATCH_MAX * c->seqStride * sizeof(ZSTD_Sequence));
        bt->hDesc = (qzstd_hip_block_t *)qzstd_hip_host_alloc(QZ_BATCH_MAX * sizeof(qzstd_hip_block_t));
        bt->hCount = (unsigned int *)qzstd_hip_host_alloc(QZ_BATCH_MAX * sizeof(unsigned int));
        bt->dvSeqs = qzstd_hip_host_device_ptr(bt->hSeqs);
        bt->dvDesc = qzstd_hip_host_device_ptr(bt->hDesc);
        bt->dvCount = qzstd_hip_host_device_ptr(bt->hCount);
        ok = bt->hSrc && bt->hSeqs && bt->hDesc && bt->hCount && bt->dvSeqs && bt->dvDesc && bt->dvCount;
    }
    if (!ok) {
        QZ_LOG(1, "coalescer setup failed on device %d: %s\n", c->device, qzstd_hip_last_error());
        return QZSTD_FAIL;
    }
    c->ready = 1;
    return QZSTD_OK;
}

static void *qzGrowDev(int dev, void *old, size_t *cap, size_t need);

/* the leader's job: one launch for the whole batch (called WITHOUT c->mu held) */
static void qzRunBatch(QZSTD_Coalescer_T *c, QZSTD_Batch_T *bt)
{
    const int n = bt->n, dev = c->device;
    unsigned int maxLen = 0;
    int i, failed = 0;
    for (i = 0; i < n; i++) {
        bt->hDesc[i].srcOff = (size_t)i * QZ_SRC_STRIDE;
        bt->hDesc[i].seqOff = (size_t)i * c->seqStride;
        bt->hDesc[i].srcLen = (unsigned int)bt->req[i].srcSize;
        bt->hDesc[i].seqCap = (unsigned int)(bt->req[i].cap < c->seqStride ? bt->req[i].cap : c->seqStride);
        if (bt->hDesc[i].srcLen > maxLen) maxLen = bt->hDesc[i].srcLen;
    }
    {
        const size_t work = qzstd_hip_workspace_bytes(bt->level, (unsigned int)n, maxLen);
        if (work) c->dWork = qzGrowDev(dev, c->dWork, &c->dWorkCap, work);
        failed = work && !c->dWork;
    }
    /* one copy in, one launch, one wait: the kernel reads the descriptors from and writes the sequences and
     * counts to this batch's pinned host buffers directly (posted PCIe writes while it runs), which takes two
     * copies and one synchronisation off the latency of a request */
    failed = failed || qzstd_hip_memcpy_h2d(dev, c->stream, c->dSrc, bt->hSrc, (size_t)n * QZ_SRC_STRIDE) ||
             qzstd_hip_find_sequences(dev, c->stream, bt->level, c->dSrc, (const qzstd_hip_block_t *)bt->dvDesc,
                                      (unsigned int)n, maxLen, bt->dvSeqs, (unsigned int *)bt->dvCount, c->dWork,
                                      c->dWorkCap) ||
             qzstd_hip_stream_sync(dev, c->stream);
    for (i = 0; i < n; i++) {
        const size_t cnt = failed ? QZSTD_HIP_NSEQ_ERROR : bt->hCount[i];
        /* capacity rule, reference :1318-1322 */
        bt->req[i].rc = (cnt == QZSTD_HIP_NSEQ_ERROR || cnt == 0 || cnt >= bt->req[i].cap - 1) ? ZSTD_SEQUENCE_PRODUCER_ERROR : cnt;
    }
    if (failed) QZ_LOG(1, "device request failed: %s\n", qzstd_hip_last_error());
    c->launches++;
    c->blocks += (unsigned long)n;
}

/* one block through the coalescer of device `dev`; returns the sequence count or the error code */
static size_t qzCoalescedBlock(int dev, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src,
                               size_t srcSize, int level)
{
    QZSTD_Coalescer_T *c = &gProc.coal[dev];
    QZSTD_Batch_T *bt;
    size_t rc;
    int i;

    pthread_mutex_lock(&c->mu);
    if (qzSetupCoalescer(c) != QZSTD_OK) {
        pthread_mutex_unlock(&c->mu);
        return ZSTD_SEQUENCE_PRODUCER_ERROR;
    }
    for (;;) { /* join the open batch (same level only) */
        bt = &c->batch[c->open];
        if (bt->state == 0 && bt->n < QZ_BATCH_MAX && (bt->n == 0 || bt->level == level)) break;
        pthread_cond_wait(&c->cvOpen, &c->mu);
    }
    i = bt->n++;
    bt->level = level;
    bt->req[i].src = src;
    bt->req[i].srcSize = srcSize;
    bt->req[i].cap = outSeqsCapacity;
    bt->req[i].rc = ZSTD_SEQUENCE_PRODUCER_ERROR;
    pthread_mutex_unlock(&c->mu);

    memcpy(bt->hSrc + (size_t)i * QZ_SRC_STRIDE, src, srcSize); /* staging copy (reference :1223), on the caller's thread */

    pthread_mutex_lock(&c->mu);
    bt->copied++;
    if (i == 0) {
        /* the first member leads its batch: it keeps collecting while the device works on the other batch,
         * then closes it and launches.  Every waiter has its own condition variable (no thundering herd
         * when more threads than cores wait here). */
        while (c->running) pthread_cond_wait(&bt->cvLead, &c->mu);
        c->running = 1;
        bt->state = 1;
        c->open ^= 1; /* newcomers now collect in the other batch (once its results are handed out) */
        pthread_cond_broadcast(&c->cvOpen);
        while (bt->copied < bt->n) pthread_cond_wait(&bt->cvLead, &c->mu); /* members still staging */
        pthread_mutex_unlock(&c->mu);
        qzRunBatch(c, bt);
        pthread_mutex_lock(&c->mu);
        bt->state = 2;
        c->running = 0;
        pthread_cond_broadcast(&bt->cvDone);
        pthread_cond_signal(&c->batch[c->open].cvLead); /* the other batch's leader may go now */
    } else {
        if (bt->state == 1 && bt->copied == bt->n) pthread_cond_signal(&bt->cvLead);
        while (bt->state != 2) pthread_cond_wait(&bt->cvDone, &c->mu);
    }
    pthread_mutex_unlock(&c->mu);

    rc = bt->req[i].rc;
    if (rc != ZSTD_SEQUENCE_PRODUCER_ERROR)
        memcpy(outSeqs, bt->hSeqs + (size_t)i * c->seqStride, rc * sizeof(ZSTD_Sequence));

    pthread_mutex_lock(&c->mu);
    if (++bt->consumed == bt->n) { /* last one out re-opens the batch */
        bt->n = bt->copied = bt->consumed = 0;
        bt->state = 0;
        pthread_cond_broadcast(&c->cvOpen);
    }
    pthread_mutex_unlock(&c->mu);
    return rc;
}

/* ---------------------------------------------------------------- lifecycle ------ */

static int qzEnvInt(const char *name, int dflt, int lo, int hi)
{
    const char *v = getenv(name);
    long x;
    if (!v || !*v) return dflt;
    x = strtol(v, NULL, 10);
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return (int)x;
}

/* Enumerate GPUs and lay the slots out round-robin across them, so that threads whose
 * hints are consecutive land on different devices (reference: the instance shuffle of
 * QZSTD_getAndShuffleInstance, :601-630). */
static int qzBuildSlots(void)
{
    int nDev = qzstd_hip_device_count();
    int perDev, i, maxDev;
    if (nDev <= 0) return QZSTD_FAIL;
    maxDev = qzEnvInt("QZSTD_HIP_MAX_DEVICES", QZ_MAX_DEVICES, 1, QZ_MAX_DEVICES);
    if (nDev > maxDev) nDev = maxDev;
    perDev = qzEnvInt("QZSTD_HIP_SLOTS", QZ_DEFAULT_SLOTS_PER_DEVICE, 1, QZ_MAX_SLOTS / nDev);
    gProc.slots = (QZSTD_Slot_T *)calloc((size_t)nDev * perDev, sizeof(QZSTD_Slot_T));
    if (!gProc.slots) return QZSTD_FAIL;
    gProc.numDevices = nDev;
    gProc.numSlots = nDev * perDev;
    for (i = 0; i < gProc.numSlots; i++) gProc.slots[i].device = i % nDev;
    gProc.coalesce = qzEnvInt("QZSTD_HIP_COALESCE", 1, 0, 1);
    gProc.coal = (QZSTD_Coalescer_T *)calloc((size_t)nDev, sizeof(QZSTD_Coalescer_T));
    if (!gProc.coal) return QZSTD_FAIL;
    for (i = 0; i < nDev; i++) {
        gProc.coal[i].device = i;
        pthread_mutex_init(&gProc.coal[i].mu, NULL);
        pthread_cond_init(&gProc.coal[i].cvOpen, NULL);
        pthread_cond_init(&gProc.coal[i].batch[0].cvLead, NULL);
        pthread_cond_init(&gProc.coal[i].batch[0].cvDone, NULL);
        pthread_cond_init(&gProc.coal[i].batch[1].cvLead, NULL);
        pthread_cond_init(&gProc.coal[i].batch[1].cvDone, NULL);
    }
    return QZSTD_OK;
}

int QZSTD_startQatDevice(void)
{
    int status;
    pthread_mutex_lock(&gProc.mutex);
    {
        const char *dbg = getenv("QZSTD_HIP_DEBUG");
        const char *rep = getenv("QZSTD_HIP_EXT_REPCODES");
        if (dbg && *dbg) qzLogLevel = atoi(dbg);
        /* the caller promises ZSTD_c_searchForExternalRepcodes = enable on its CCtx (libzstd's default only
         * from level 10): repeat-offset aware sequences at every level */
        gProc.levelFlags = (rep && atoi(rep) > 0) ? QZSTD_HIP_LEVEL_REPCODES : 0;
        {
            /* transparent look-ahead needs a fault-safe read.  QZSTD_HIP_LOOKAHEAD: 0 off, 1 (default) on, 2 on and
             * always through a pipe.  process_vm_readv is only tried where no seccomp filter could make an unusual
             * system call fatal, and only kept if a probe on ourselves works */
            char probe[16] = "qzstd", back[16];
            const int want = qzEnvInt("QZSTD_HIP_LOOKAHEAD", 1, 0, 2);
            gProc.lookahead = want;
            if (want == 1 && (prctl(PR_GET_SECCOMP, 0, 0, 0, 0) != 0 || qzSafeRead(back, probe, 16, 16) != 16 ||
                              memcmp(back, probe, 16) != 0))
                gProc.lookahead = 2;
        }
    }
    if (gProc.status == QZSTD_FAIL) {
        /* runtime up? (reference: QZSTD_salUserStart, :498-527) */
        gProc.status = qzstd_hip_device_count() > 0 ? QZSTD_STARTED : QZSTD_FAIL;
        if (gProc.status == QZSTD_FAIL) QZ_LOG(2, "no HIP device: %s\n", qzstd_hip_last_error());
    }
    if (gProc.status == QZSTD_STARTED) {
        gProc.status = qzBuildSlots() == QZSTD_OK ? QZSTD_OK : QZSTD_STARTED;
    }
    QZ_LOG(2, "start: status %d, %d device(s), %d slot(s)\n", gProc.status, gProc.numDevices, gProc.numSlots);
    status = gProc.status;
    pthread_mutex_unlock(&gProc.mutex);
    return status;
}

void QZSTD_stopQatDevice(void)
{
    int i;
    pthread_mutex_lock(&gProc.mutex);
    if (gProc.slots) {
        for (i = 0; i < gProc.numSlots; i++) qzFreeSlot(&gProc.slots[i]);
        free(gProc.slots);
    }
    if (gProc.coal) {
        for (i = 0; i < gProc.numDevices; i++) qzFreeCoalescer(&gProc.coal[i]);
        free(gProc.coal);
    }
    gProc.coal = NULL;
    gProc.slots = NULL;
    gProc.numSlots = 0;
    gProc.numDevices = 0;
    gProc.status = QZSTD_FAIL;
    pthread_mutex_unlock(&gProc.mutex);
}

void *QZSTD_createSeqProdState(void)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)calloc(1, sizeof(QZSTD_Session_T));
    if (!s) return NULL;
    s->slotHint = -1;
    s->pipeFd[0] = s->pipeFd[1] = -1;
    return s;
}

static void qzReleaseSlot(int i);

static unsigned long qzNowNs(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long)ts.tv_sec * 1000000000ul + (unsigned long)ts.tv_nsec;
}

/* wait for an in-flight hint and give its slot back; the hint becomes ready (or empty on failure) */
static void qzHintFinish(QZSTD_Hint_T *h)
{
    if (h->st != 1) return;
    if (gProc.slots && h->slot >= 0 && h->slot < gProc.numSlots) {
        QZSTD_Slot_T *sl = &gProc.slots[h->slot];
        const int bad = qzstd_hip_stream_sync(sl->device, sl->stream);
        qzReleaseSlot(h->slot);
        h->st = bad ? 0 : 2;
        if (bad) QZ_LOG(1, "look-ahead batch failed: %s\n", qzstd_hip_last_error());
    } else {
        h->st = 0;
    }
}

void QZSTD_freeSeqProdState(void *sequenceProducerState)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)sequenceProducerState;
    int k;
    if (!s) return;
    QZ_LOG(2, "state %p: %lu block(s) served from a look-ahead batch (%lu of them speculative, %lu speculation(s)), %lu per "
              "block; %lu hint(s): staging %.2f ms, queueing %.2f ms, waited %.2f ms for the GPU\n", (void *)s,
           s->servedFromBatch, s->autoServed, s->autoLaunched, s->servedSync, s->hintCalls, s->hintStageNs / 1e6,
           s->hintQueueNs / 1e6, s->hintWaitNs / 1e6);
    for (k = 0; k < 4; k++) {
        qzHintFinish(&s->hint[k]);
        qzstd_hip_host_free(s->hint[k].hSrc);
        qzstd_hip_host_free(s->hint[k].hSeqs);
        qzstd_hip_host_free(s->hint[k].hCount);
        qzstd_hip_host_free(s->hint[k].hDesc);
    }
    if (s->pipeFd[0] >= 0) close(s->pipeFd[0]);
    if (s->pipeFd[1] >= 0) close(s->pipeFd[1]);
    free(s);
}

/* ---------------------------------------------------------------- hot path ------- */

/* shared by the producer and the hint: is the device usable?  Counts failures and
 * re-probes every QZ_RETRY_INTERVAL_BLOCKS-th block (reference :1140-1152). */
static int qzDeviceUsable(QZSTD_Session_T *s)
{
    if (gProc.status == QZSTD_OK) return 1;
    s->failOffloadCnt++;
    if (s->failOffloadCnt >= QZ_RETRY_INTERVAL_BLOCKS) {
        s->failOffloadCnt = 0;
        if (QZSTD_startQatDevice() == QZSTD_OK) return 1;
        QZ_LOG(1, "tried to restart the device, but failed\n");
        return 0;
    }
    QZ_LOG(1, "the device was not successfully started\n");
    return 0;
}

/* one block, synchronously, on slot i */
static size_t qzRunBlock(QZSTD_Slot_T *sl, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src,
                         size_t srcSize, int level)
{
    const size_t cap = outSeqsCapacity < sl->seqCap ? outSeqsCapacity : sl->seqCap;
    size_t first, count;
    memcpy(sl->hSrc, src, srcSize); /* staging copy, reference :1223 */
    sl->hDesc->srcOff = 0;
    sl->hDesc->seqOff = 0;
    sl->hDesc->srcLen = (unsigned int)srcSize;
    sl->hDesc->seqCap = (unsigned int)cap;
    {
        const size_t work = qzstd_hip_workspace_bytes(level, 1, (unsigned int)srcSize);
        if (work) sl->dWork = qzGrowDev(sl->device, sl->dWork, &sl->dWorkCap, work);
        if (work && !sl->dWork) goto fail;
    }
    if (qzstd_hip_memcpy_h2d(sl->device, sl->stream, sl->dSrc, sl->hSrc, (srcSize + 15) & ~(size_t)15) ||
        qzstd_hip_memcpy_h2d(sl->device, sl->stream, sl->dDesc, sl->hDesc, sizeof(*sl->hDesc)) ||
        qzstd_hip_find_sequences(sl->device, sl->stream, level, sl->dSrc, sl->dDesc, 1, (unsigned int)srcSize,
                                 sl->dSeqs, sl->dCount, sl->dWork, sl->dWorkCap))
        goto fail;
    first = cap < QZ_FIRST_COPY_SEQS ? cap : QZ_FIRST_COPY_SEQS;
    if (qzstd_hip_memcpy_d2h(sl->device, sl->stream, sl->hCount, sl->dCount, sizeof(unsigned int)) ||
        qzstd_hip_memcpy_d2h(sl->device, sl->stream, sl->hSeqs, sl->dSeqs, first * sizeof(ZSTD_Sequence)) ||
        qzstd_hip_stream_sync(sl->device, sl->stream))
        goto fail;
    count = *sl->hCount;
    if (count == QZSTD_HIP_NSEQ_ERROR || count == 0 || count >= outSeqsCapacity - 1) {
        QZ_LOG(1, "sequence count %zu does not fit capacity %zu\n", count, outSeqsCapacity);
        return ZSTD_SEQUENCE_PRODUCER_ERROR; /* reference :1318-1322 */
    }
    if (count > first) {
        if (qzstd_hip_memcpy_d2h(sl->device, sl->stream, sl->hSeqs + first, sl->dSeqs + first,
                                 (count - first) * sizeof(ZSTD_Sequence)) ||
            qzstd_hip_stream_sync(sl->device, sl->stream))
            goto fail;
    }
    memcpy(outSeqs, sl->hSeqs, count * sizeof(ZSTD_Sequence));
    return count;
fail:
    QZ_LOG(1, "device request failed: %s\n", qzstd_hip_last_error());
    return ZSTD_SEQUENCE_PRODUCER_ERROR;
}

size_t qatSequenceProducer(void *sequenceProducerState, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity,
                           const void *src, size_t srcSize, const void *dict, size_t dictSize,
                           int compressionLevel, size_t windowSize)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)sequenceProducerState;
    size_t rc = ZSTD_SEQUENCE_PRODUCER_ERROR;
    int i;

    /* guards, reference :1123-1137 */
    if (windowSize < (srcSize < 32 * 1024 ? srcSize : 32 * 1024) || dictSize > 0 || dict) {
        QZ_LOG(2, "window %zu too small for block %zu, or dictionary given (%zu)\n", windowSize, srcSize, dictSize);
        return ZSTD_SEQUENCE_PRODUCER_ERROR;
    }
    if (compressionLevel < QZ_LEVEL_MIN || compressionLevel > QZ_LEVEL_MAX) {
        QZ_LOG(1, "only levels 1-12 can be offloaded, got %d\n", compressionLevel);
        return ZSTD_SEQUENCE_PRODUCER_ERROR;
    }
    if (!s || !outSeqs || !src || srcSize > QZSTD_HIP_BLOCK_MAX || outSeqsCapacity < 3) return ZSTD_SEQUENCE_PRODUCER_ERROR;
    if (!qzDeviceUsable(s)) return ZSTD_SEQUENCE_PRODUCER_ERROR;

    /* look-ahead batch hit?  (src, srcSize) must start on an announced (k < 2) or guessed (k >= 2) block grid and
     * cover one or more whole blocks of it: libzstd 1.5.7 cuts multi-block frames into blocks of 32..128 KiB at 32 KiB
     * steps, so a finer grid serves several sizes — independently parsed neighbours are simply concatenated, the trailing literals of
     * one block flowing into the first sequence of the next */
    {
        int k, guessMissed = 0;
        for (k = 0; k < 4; k++) {
            QZSTD_Hint_T *h = &s->hint[k];
            const unsigned char *p = (const unsigned char *)src;
            size_t rel, b, e, covered = 0;
            if (h->st == 0 || h->level != compressionLevel || p < h->base || p + srcSize > h->base + h->size) continue;
            rel = (size_t)(p - h->base);
            b = rel / h->block;
            if (rel % h->block != 0 || b >= h->nb) continue;
            for (e = b; e < h->nb && covered < srcSize; e++) covered += h->hDesc[e].srcLen;
            /* a GUESS must not change what the caller gets: it serves a callback only block for block (joining
             * independently parsed grid blocks costs ratio; for announcements that is the announcer's choice) */
            if (covered != srcSize || e - b > 8 || (k >= 2 && e - b != 1)) {
                QZ_LOG(3, "look-ahead %d: block %zu+%zu does not fit the grid (%zu)\n", k, rel, srcSize, h->block);
                continue;
            }
            if (k >= 2 && memcmp(h->hSrc + rel, src, srcSize) != 0) { /* the guess was read before these bytes were final */
                guessMissed = 1;
                continue;
            }
            if (h->st == 1) { /* first use: wait for the GPU (usually long done) */
                const unsigned long w0 = qzNowNs();
                qzHintFinish(h);
                s->hintWaitNs += qzNowNs() - w0;
            }
            if (h->st == 2) {
                const int last = rel + srcSize >= h->size;
                size_t total = 1, carry = 0, out = 0, bi;
                int usable = 1;
                for (bi = b; bi < e; bi++) {
                    const size_t count = h->hCount[bi];
                    if (count == QZSTD_HIP_NSEQ_ERROR || count == 0 || count > QZ_HINT_PITCH) usable = 0;
                    total += count - 1;
                }
                if (usable && total < outSeqsCapacity - 1) {
                    for (bi = b; bi < e; bi++) {
                        const ZSTD_Sequence *q = h->hSeqs + bi * QZ_HINT_PITCH;
                        const size_t count = h->hCount[bi];
                        if (count > 1) {
                            memcpy(outSeqs + out, q, (count - 1) * sizeof(ZSTD_Sequence));
                            outSeqs[out].litLength += (unsigned int)carry;
                            out += count - 1;
                            carry = 0;
                        }
                        carry += q[count - 1].litLength; /* the block's delimiter: its trailing literals */
                    }
                    outSeqs[out].offset = 0;
                    outSeqs[out].litLength = (unsigned int)carry;
                    outSeqs[out].matchLength = 0;
                    outSeqs[out].rep = 0;
                    out++;
                    s->servedFromBatch++;
                    if (k >= 2) {
                        s->autoServed++;
                        s->autoFails = 0;
                        s->autoOutstanding = 0;
                        /* keep the pipeline full: once past the middle of a guess, guess what follows it */
                        if ((b < (h->nb + 1) / 2 && e >= (h->nb + 1) / 2) || h->nb == 1) {
                            const QZSTD_Hint_T *o = &s->hint[2 + ((k - 2) ^ 1)];
                            const unsigned char *nxt = h->base + h->size;
                            if (!(o->st != 0 && o->base == nxt)) {
                                if (s->autoDepth < QZ_AUTO_DEPTH_MAX) s->autoDepth *= 2;
                                s->autoNext = (k - 2) ^ 1;
                                qzSpeculate(s, nxt, h->block, compressionLevel);
                            }
                        }
                    }
                    if (last) h->st = 0; /* last block consumed */
                    return out;
                }
                if (last) h->st = 0;
            }
            break; /* announced but unusable (too many sequences, failed launch): per-block path */
        }
        QZ_LOG(3, "miss: %p + %zu (guesses: %d %p+%zu, %d %p+%zu) outstanding %d backoff %u depth %u\n", src, srcSize, s->hint[2].st,
               (const void *)s->hint[2].base, s->hint[2].size, s->hint[3].st, (const void *)s->hint[3].base, s->hint[3].size,
               s->autoOutstanding, s->autoBackoff, s->autoDepth);
        /* nothing to serve from.  Unannounced caller: guess that the bytes after this block come next */
        if (s->hint[0].st == 0 && s->hint[1].st == 0) {
            if (guessMissed || s->autoOutstanding) { /* the last guess was wrong: back off exponentially, start small again */
                s->autoOutstanding = 0;
                s->autoFails++;
                s->autoBackoff = s->autoFails < 8 ? (1u << (s->autoFails - 1)) - 1u : 255u;
                s->autoDepth = QZ_AUTO_DEPTH_MIN;
                for (k = 2; k < 4; k++)
                    if (s->hint[k].st == 2) s->hint[k].st = 0;
            }
            qzSpeculate(s, (const unsigned char *)src + srcSize, srcSize, compressionLevel);
        }
    }

    if (gProc.coalesce) {
        /* sticky device per state, states spread round-robin over the GPUs */
        static volatile unsigned int nextDev = 0;
        if (s->slotHint < 0) s->slotHint = (int)(__sync_fetch_and_add(&nextDev, 1u) & 0x3FFFFFFFu);
        rc = qzCoalescedBlock(s->slotHint % gProc.numDevices, outSeqs, outSeqsCapacity, src, srcSize,
                              compressionLevel | gProc.levelFlags);
        if (rc != ZSTD_SEQUENCE_PRODUCER_ERROR) s->servedSync++;
        QZ_LOG(2, "block %zu B level %d -> %zu sequences (coalesced, device %d)\n", srcSize, compressionLevel, rc,
               s->slotHint % gProc.numDevices);
        return rc;
    }
    i = qzGrabSlot(s->slotHint);
    if (i < 0) {
        QZ_LOG(1, "failed to grab a slot\n");
        return ZSTD_SEQUENCE_PRODUCER_ERROR;
    }
    s->slotHint = i;
    if (qzSetupSlot(&gProc.slots[i]) == QZSTD_OK) {
        rc = qzRunBlock(&gProc.slots[i], outSeqs, outSeqsCapacity, src, srcSize, compressionLevel | gProc.levelFlags);
        if (rc != ZSTD_SEQUENCE_PRODUCER_ERROR) s->servedSync++;
    }
    QZ_LOG(2, "block %zu B level %d -> %zu sequences (slot %d, device %d)\n", srcSize, compressionLevel, rc, i,
           gProc.slots[i].device);
    qzReleaseSlot(i);
    return rc;
}

/* ---------------------------------------------------------------- look-ahead ----- */

/* grow-only buffers: returns the (possibly new) pointer, NULL on failure */
static void *qzGrowHost(void *old, size_t *cap, size_t need)
{
    void *p;
    if (old && *cap >= need) return old;
    qzstd_hip_host_free(old);
    p = qzstd_hip_host_alloc(need);
    *cap = p ? need : 0;
    return p;
}

static void *qzGrowDev(int dev, void *old, size_t *cap, size_t need)
{
    void *p;
    if (old && *cap >= need) return old;
    qzstd_hip_free(dev, old);
    p = qzstd_hip_malloc(dev, need);
    *cap = p ? need : 0;
    return p;
}

/* one quick sweep over the slots (no waiting) */
static int qzTryGrabSlot(int hint)
{
    int k;
    const int n = gProc.numSlots;
    if (n <= 0) return -1;
    if (hint < 0 || hint >= n) hint = 0;
    for (k = 0; k < n; k++) {
        const int i = (hint + k) % n;
        if (__sync_lock_test_and_set(&gProc.slots[i].lock, 1) == 0) return i;
    }
    return -1;
}

void QZSTD_hintStats(void *sequenceProducerState, unsigned long stats[4])
{
    const QZSTD_Session_T *s = (const QZSTD_Session_T *)sequenceProducerState;
    if (!stats) return;
    stats[0] = s ? s->servedFromBatch : 0;
    stats[1] = s ? s->servedSync : 0;
    stats[2] = s ? s->hintCalls : 0;
    stats[3] = s ? s->hintWaitNs / 1000 : 0;
}

/* Copy [src, src + len) into dst without ever faulting: whole blocks of `block` bytes as long as they are
 * readable.  process_vm_readv on ourselves is the kernel's copy_from_user: an unmapped or PROT_NONE page ends the
 * transfer (at iovec granularity) instead of raising SIGSEGV.  Returns the bytes copied (a multiple of block). */
static size_t qzSafeRead(void *dst, const void *src, size_t len, size_t block)
{
    struct iovec rem[QZ_HINT_MAX_BYTES / 4096 > 1024 ? 1024 : 64], loc;
    const size_t nb = len / block;
    size_t b, done = 0;
    ssize_t n;
    if (nb == 0 || nb > sizeof(rem) / sizeof(rem[0])) return 0;
    for (b = 0; b < nb; b++) {
        rem[b].iov_base = (void *)((uintptr_t)src + b * block);
        rem[b].iov_len = block;
    }
    loc.iov_base = dst;
    loc.iov_len = nb * block;
    n = process_vm_readv(getpid(), &loc, 1, rem, (unsigned long)nb, 0);
    if (n > 0) done = ((size_t)n / block) * block;
    return done;
}

/* The same through a pipe, with nothing but pipe/write/read (for processes under a seccomp filter, where an unusual
 * system call may be fatal): write() copies from user memory inside the kernel and stops with EFAULT at an
 * unreadable page; what went in is read back out into dst. */
static size_t qzSafeReadPipe(int fd[2], void *dst, const void *src, size_t len, size_t block)
{
    size_t done = 0;
    if (fd[0] < 0) {
        if (pipe(fd) != 0) { fd[0] = fd[1] = -1; return 0; }
        (void)fcntl(fd[1], F_SETFL, O_NONBLOCK);
    }
    while (done < len) {
        const size_t want = len - done < 65536 ? len - done : 65536; /* the default capacity of a pipe */
        const ssize_t w = write(fd[1], (const char *)src + done, want);
        size_t got = 0;
        if (w <= 0) break;
        while (got < (size_t)w) {
            const ssize_t r = read(fd[0], (char *)dst + done + got, (size_t)w - got);
            if (r <= 0) return (done / block) * block;
            got += (size_t)r;
        }
        done += (size_t)w;
        if ((size_t)w < want) break;
    }
    return (done / block) * block;
}

/* Stage a buffer, queue its match-finding on a slot's stream and remember it in *h (asynchronous, see
 * QZSTD_hintSource).  speculative: the buffer is a GUESS (what follows the block of the current callback): read it
 * fault-safely, take only whole readable blocks, never wait for a slot.  Returns the bytes announced, 0 if none. */
static size_t qzAnnounce(QZSTD_Session_T *s, QZSTD_Hint_T *h, const void *src, size_t srcSize, size_t blockSize,
                         int compressionLevel, int speculative)
{
    QZSTD_Slot_T *sl;
    size_t nb, b, stride, blocksBytes, srcBytes;
    unsigned long tq;
    int i;

    qzHintFinish(h); /* an old announcement that was never consumed */
    h->st = 0;
    nb = (srcSize + blockSize - 1) / blockSize;
    stride = qzstd_hip_sequence_bound(blockSize);
    blocksBytes = nb * sizeof(qzstd_hip_block_t);
    srcBytes = (srcSize + 63) & ~(size_t)63;

    h->hSrc = (unsigned char *)qzGrowHost(h->hSrc, &h->hSrcCap, srcBytes);
    h->hDesc = (qzstd_hip_block_t *)qzGrowHost(h->hDesc, &h->hDescCap, blocksBytes);
    h->hCount = (unsigned int *)qzGrowHost(h->hCount, &h->hCountCap, nb * sizeof(unsigned int));
    h->hSeqs = (ZSTD_Sequence *)qzGrowHost(h->hSeqs, &h->hSeqsCap, nb * QZ_HINT_PITCH * sizeof(ZSTD_Sequence));
    h->dvDesc = qzstd_hip_host_device_ptr(h->hDesc);
    h->dvCount = qzstd_hip_host_device_ptr(h->hCount);
    h->dvSeqs = qzstd_hip_host_device_ptr(h->hSeqs);
    if (!h->hSrc || !h->hDesc || !h->hCount || !h->hSeqs || !h->dvDesc || !h->dvCount || !h->dvSeqs) return 0;

    tq = qzNowNs();
    if (speculative) {
        srcSize = gProc.lookahead == 2 ? qzSafeReadPipe(s->pipeFd, h->hSrc, src, srcSize, blockSize)
                                       : qzSafeRead(h->hSrc, src, srcSize, blockSize);
        if (srcSize == 0) return 0;
        nb = srcSize / blockSize;
        srcBytes = (srcSize + 63) & ~(size_t)63;
    }
    i = qzTryGrabSlot(s->slotHint);
    if (i < 0) {
        if (speculative) return 0;
        /* every slot is busy: give back what this state still holds, then wait for one */
        for (b = 0; b < 4; b++) qzHintFinish(&s->hint[b]);
        i = qzGrabSlot(s->slotHint);
        if (i < 0) return 0;
    }
    s->slotHint = i;
    sl = &gProc.slots[i];
    if (qzSetupSlot(sl) != QZSTD_OK) goto fail;
    sl->dBatchSrc = (unsigned char *)qzGrowDev(sl->device, sl->dBatchSrc, &sl->dBatchSrcCap, srcBytes);
    if (!sl->dBatchSrc) goto fail;
    {
        const size_t work = qzstd_hip_workspace_bytes(compressionLevel | gProc.levelFlags, (unsigned int)nb, (unsigned int)blockSize);
        if (work) sl->dBatchWork = qzGrowDev(sl->device, sl->dBatchWork, &sl->dBatchWorkCap, work);
        if (work && !sl->dBatchWork) goto fail;
    }
    if (!speculative) memcpy(h->hSrc, src, srcSize); /* pinned staging: the H2D below is then truly asynchronous */
    for (b = 0; b < nb; b++) {
        const size_t o = b * blockSize;
        h->hDesc[b].srcOff = o;
        /* results go straight into the pinned buffer, QZ_HINT_PITCH entries per block: a block with more
         * sequences reports an error and is redone by the per-block path when its callback comes */
        h->hDesc[b].seqOff = b * QZ_HINT_PITCH;
        h->hDesc[b].srcLen = (unsigned int)(srcSize - o < blockSize ? srcSize - o : blockSize);
        h->hDesc[b].seqCap = (unsigned int)(stride < QZ_HINT_PITCH ? stride : QZ_HINT_PITCH);
    }
    s->hintStageNs += qzNowNs() - tq;
    tq = qzNowNs();
    /* everything below is queued on the slot's stream and returns immediately */
    if (qzstd_hip_memcpy_h2d(sl->device, sl->stream, sl->dBatchSrc, h->hSrc, srcBytes) ||
        qzstd_hip_find_sequences(sl->device, sl->stream, compressionLevel | gProc.levelFlags, sl->dBatchSrc,
                                 (const qzstd_hip_block_t *)h->dvDesc, (unsigned int)nb, (unsigned int)blockSize, h->dvSeqs,
                                 (unsigned int *)h->dvCount, sl->dBatchWork, sl->dBatchWorkCap)) {
        (void)qzstd_hip_stream_sync(sl->device, sl->stream);
        goto fail;
    }
    h->base = (const unsigned char *)src;
    h->size = srcSize;
    h->block = blockSize;
    h->level = compressionLevel;
    h->nb = nb;
    h->slot = i;
    h->st = 1; /* in flight; the slot stays ours until qzHintFinish() */
    s->hintQueueNs += qzNowNs() - tq;
    return srcSize;
fail:
    QZ_LOG(1, "look-ahead not taken: %s\n", qzstd_hip_last_error());
    qzReleaseSlot(i);
    return 0;
}

int QZSTD_hintSource(void *sequenceProducerState, const void *src, size_t srcSize, size_t blockSize,
                     int compressionLevel)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)sequenceProducerState;
    QZSTD_Hint_T *h;

    if (!s || !src || srcSize == 0 || srcSize > QZ_HINT_MAX_BYTES || blockSize == 0 || blockSize > QZSTD_HIP_BLOCK_MAX ||
        (blockSize & 15))
        return -1;
    if (compressionLevel < QZ_LEVEL_MIN || compressionLevel > QZ_LEVEL_MAX) return -1;
    if (!qzDeviceUsable(s)) return -1;
    h = &s->hint[s->hintNext];
    s->hintNext ^= 1;
    if (qzAnnounce(s, h, src, srcSize, blockSize, compressionLevel, 0) == 0) return -1;
    s->hintCalls++;
    return 0;
}

/* Transparent look-ahead for callers that announce nothing.  libzstd hands over one block per callback and waits,
 * but most callers walk a contiguous buffer (a file in chunks, a multi-block frame), so the bytes that FOLLOW the
 * current block are very likely the next blocks.  On a callback that had to take the per-block path, guess: read
 * the following blocks fault-safely, and let the GPU match-find them while this block is being served and its
 * frame entropy-coded.  A later callback is served from a guess only if its (src, srcSize) sits on the guessed
 * grid AND its bytes still equal the staged copy (memcmp), so a wrong guess costs GPU time, never correctness.
 * The depth doubles while guesses are consumed (2 .. 32 blocks); misses back off exponentially. */
static void qzSpeculate(QZSTD_Session_T *s, const unsigned char *next, size_t blockSize, int compressionLevel)
{
    QZSTD_Hint_T *h;
    /* only with the coalescer: there the per-block path needs no slot, so guesses that hold slots cannot starve it */
    if (!gProc.lookahead || !gProc.coalesce || (blockSize & 15) || blockSize < 4096) return;
    if (s->autoBackoff) { s->autoBackoff--; return; }
    if (s->autoDepth < QZ_AUTO_DEPTH_MIN) s->autoDepth = QZ_AUTO_DEPTH_MIN;
    h = &s->hint[2 + s->autoNext];
    if (h->st == 1 && qzstd_hip_stream_query(gProc.slots[h->slot].device, gProc.slots[h->slot].stream) == 1)
        return; /* the buffer we would reuse is still on the GPU: do not wait for a guess */
    if (qzAnnounce(s, h, next, (size_t)s->autoDepth * blockSize, blockSize, compressionLevel, 1) != 0) {
        s->autoNext ^= 1;
        s->autoLaunched++;
        s->autoOutstanding = 1;
    } else {
        s->autoBackoff = 16; /* unreadable, or no slot free: try again later */
    }
}
