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
 *     :529-663                                       consecutive slots sit on different devices;
 *                                                    an announced buffer is split across the GPUs
 *   QZSTD_grabInstance / releaseInstance :905-933  test-and-set sweep starting at the hint (hint path,
 *                                                    QZSTD_HIP_COALESCE=0); by default callers are
 *                                                    merged into batches, several in flight per GPU
 *   QZSTD_allocInstMem (lazy)  :685-822            pinned + device buffers, created on first use
 *   input staging memcpy       :1222-1227          memcpy into the pinned staging buffer
 *   cpaDcCompressData2 + poll  :1243-1272          H2D, kernel launch on a stream, bounded wait
 *   poll time-out              :1099-1104,:1261-72 QZSTD_HIP_TIMEOUT_MS (default 2000): error -> libzstd's
 *                                                    fallback; the stream is quarantined until it drains
 *   QZSTD_decLz4s              :1013-1091          (none: the kernel emits ZSTD_Sequence)
 *   result / capacity checks   :1293-1322          count == NSEQ_ERROR or >= cap-1 -> ERROR
 *   device-down counter, retry every 1000 blocks   same (failOffloadCnt)
 *     :88, :1140-1152
 *
 * No QAT / icp_sal / cpa symbol is used or emulated.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* SYS_getcpu, sched_getaffinity */
#endif
#include "qatseqprod.h"
#include "qzstd_hip.h"

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <emmintrin.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <sys/types.h>
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
#define QZ_DEFAULT_TIMEOUT_MS 2000 /* reference: 2 s of polling, src/qatseqprod.c:1099-1104 */

static int qzLogLevel = DEBUGLEVEL; /* 0 silent, 1 errors, 2 events, 3 every sequence */

/* Why a callback returned ZSTD_SEQUENCE_PRODUCER_ERROR (counted per state, QZSTD_failStats; the reference counts only
 * the device-down case: failOffloadCnt, src/qatseqprod.c:122,:1141).  The failing site names the cause in a thread-local;
 * a batch's leader leaves it in the request for the member that waits. */
enum { QZ_CAUSE_NONE = 0, QZ_CAUSE_GUARD, QZ_CAUSE_DEVICE_DOWN, QZ_CAUSE_TIMEOUT, QZ_CAUSE_CAPACITY, QZ_CAUSE_RUNTIME, QZ_CAUSE_N };
static __thread int qzCause = QZ_CAUSE_NONE;
#define QZ_LOG(l, ...)                                       \
    do {                                                     \
        if ((l) <= qzLogLevel) {                             \
            fprintf(stderr, "qatseqprod(hip): " __VA_ARGS__); \
        }                                                    \
    } while (0)

/* One slot = one stream with its staging and device buffers on one GPU (the analogue of a QAT DC instance). */
typedef struct {
    int device;
    volatile int lock;
    int ready; /* buffers + stream exist */
    int stuck; /* a wait on the stream timed out: not reusable until the stream drains */
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
    /* grow-only buffers of the batched (announced) path */
    unsigned char *dBatchSrc; size_t dBatchSrcCap;
    /* launch scratch (hash chains of levels >= 5), grow-only: one block / an announced batch */
    void *dWork; size_t dWorkCap;
    void *dBatchWork; size_t dBatchWorkCap;
    /* buffers of the resident service's requests (qzstd_hip_service_submit), created on first use: the staged block, its
     * device twin, QZ_SVC_ITEMS_MAX result regions and count words — the words are what the caller polls */
    unsigned char *vSrc;   /* pinned */
    unsigned char *vdSrc;  /* device */
    ZSTD_Sequence *vSeqs;  /* pinned, QZ_SVC_ITEMS_MAX x QZ_SVC_ITEM_CAP */
    unsigned int *vCount;  /* pinned, QZ_SVC_ITEMS_MAX */
    void *vdWork;          /* device: chain scratch of a request (levels >= 5: shared by its items), QZSTD_HIP_SVC_WORK_BYTES, on first use */
    unsigned int vEpoch, vItems; /* the last request: its epoch, its item count */
    int vStuck;            /* that request timed out: the slot serves no request before all its count words have arrived */
} QZSTD_Slot_T;
#define QZ_SVC_ITEMS_MAX QZSTD_HIP_SVC_MAX_ITEMS
#define QZ_SVC_ITEM_CAP ((size_t)1371) /* ZSTD_sequenceBound(4096): the smallest item is one 4 KiB segment; 32 of them hold any block's worst case */
#define QZ_NOT_SERVED ((size_t)-2)     /* qzServiceBlock: the resident service does not take this request, use the launch path */

/*
 * Cross-thread request coalescing (one per GPU).  The producer API hands over ONE block per call and waits, and a
 * single block keeps one of 256 CUs busy; many callers (one CCtx per thread, the reference's own scaling model:
 * README.md:138, many DC instances per device: src/qatseqprod.c:905-928) are therefore merged into batches.  A GPU has
 * QZ_BATCHES batches, each with its own stream and staging, so several launches are in flight at once: a caller that
 * finds no batch collecting opens an idle one and leads it — it stages its block, closes the batch and launches, so a
 * lone caller never waits for anybody; callers that arrive while a leader is staging join its batch; when every batch
 * is busy, newcomers wait for the first one to come back and then pile into it together ("group commit": no timers).
 * Every caller copies its own block into the batch's pinned staging area and its own result out of it, so those
 * copies run in parallel on the callers' threads.  Levels may be mixed: one launch per level present in a batch.
 */
#define QZ_BATCH_MAX 128
#define QZ_BATCHES 4
#define QZ_BATCH_PITCH ((size_t)16384) /* sequences per block in the batch's result area; denser blocks are redone alone */
#define QZ_SEGS_MAX 32 /* a batched block goes to the GPU as up to 32 work items, each a run of whole segments (profile.segLog): as many
                        * as keep the launch within one workgroup per CU — 32 per block while at most 8 callers share a batch, 8 up to 32
                        * callers, whole blocks beyond */
#define QZ_DESC_MAX (QZ_BATCH_MAX + QZ_SPLIT_ITEMS_MAX) /* descriptors of a batch: unsplit, or split within QZ_SPLIT_ITEMS_MAX */
#define QZ_SPLIT_ITEMS_MAX 256 /* batches are only cut into segment items while the launch stays within one workgroup per CU */
typedef struct {
    const void *src;
    size_t srcSize, cap, rc;
    int level;
    int nSeg, dense;                /* segments submitted; the batch's result area was too small: redo alone */
    int cause;                      /* QZ_CAUSE_* when rc is the error code (set by the batch's leader) */
    unsigned int segPitch;          /* result entries per segment: QZ_BATCH_PITCH / nSeg */
    unsigned int segCnt[QZ_SEGS_MAX]; /* sequences per segment, each including its delimiter */
} QZSTD_Req_T;

typedef struct {
    int state; /* 0 idle or collecting, 1 closed (running), 2 done (results being copied out) */
    int n, copied, consumed;
    int ready, stuck;
    QZSTD_Req_T req[QZ_BATCH_MAX];
    unsigned char *hSrc;      /* pinned, QZ_BATCH_MAX x QZ_SRC_STRIDE */
    unsigned char *dSrc;      /* device, same size */
    ZSTD_Sequence *hSeqs;     /* pinned, QZ_BATCH_MAX x QZ_BATCH_PITCH */
    qzstd_hip_block_t *hDesc; /* pinned, QZ_BATCH_MAX x QZ_SEGS_MAX */
    unsigned int *hCount;     /* pinned, QZ_BATCH_MAX x QZ_SEGS_MAX */
    void *dvSeqs, *dvDesc, *dvCount; /* device-side addresses of hSeqs / hDesc / hCount */
    void *stream;
    void *dWork; /* launch scratch, grow-only */
    size_t dWorkCap;
    pthread_cond_t cvLead;    /* the batch's leader (its first member) waits here for the members to finish staging */
    pthread_cond_t cvDone;    /* the other members wait here for the results */
} QZSTD_Batch_T;

typedef struct {
    int device;
    int open; /* index of the batch that is collecting, -1 = none */
    pthread_mutex_t mu;
    pthread_cond_t cvOpen; /* callers that found every batch busy wait here */
    QZSTD_Batch_T batch[QZ_BATCHES];
    unsigned long launches, blocks, batches;
} QZSTD_Coalescer_T;
#define QZ_SRC_STRIDE ((size_t)QZSTD_HIP_BLOCK_MAX + 64)
#define QZ_NUMA_NODES_MAX 16

typedef struct {
    int status; /* QZSTD_Status_e */
    int numDevices;
    int numSlots;
    QZSTD_Slot_T *slots;
    QZSTD_Coalescer_T *coal; /* one per device */
    int coalesce;            /* QZSTD_HIP_COALESCE (default 1) */
    int levelFlags;          /* QZSTD_HIP_LEVEL_REPCODES when QZSTD_HIP_EXT_REPCODES=1 */
    int timeoutMs;           /* QZSTD_HIP_TIMEOUT_MS */
    int split;               /* QZSTD_HIP_SPLIT: announced buffers are split across this many GPUs (default: all) */
    int splitBlocks;         /* QZSTD_HIP_SPLIT_BLOCKS (default 1): per-block requests of segmentable levels go as segments */
    int service;             /* QZSTD_HIP_SERVICE (default 1): per-block requests go to the resident service where it serves the level */
    int svcItemBytes;        /* QZSTD_HIP_SERVICE_ITEM (default 4096): bytes per work item of a service request (whole segments) */
    int svcSpinUs;           /* QZSTD_HIP_SERVICE_SPIN_US (default 400; 10 when there are more states than usable cores): busy polling of the count words before napping */
    unsigned long devBlocks[QZ_MAX_DEVICES][3]; /* per GPU: blocks queued from announcements, blocks through batches, blocks through the service */
    pthread_mutex_t mutex;
    /* NUMA (reference: qaeMemAllocNUMA(size, node, 64) for every DMA buffer, src/qatseqprod.c:216-246): the host node every GPU hangs
     * off, so that a slot's / batch's / announcement's pinned memory sits next to its GPU and a state lands on a GPU of the socket
     * its thread runs on */
    int numa;                      /* QZSTD_HIP_NUMA (default 1) */
    int numaThreadNode;            /* QZSTD_HIP_NUMA_NODE: treat every calling thread as running on this node (-1: ask the kernel) */
    int devNode[QZ_MAX_DEVICES];   /* -1 = unknown */
    unsigned int nodeNext[QZ_NUMA_NODES_MAX], anyNext; /* round-robin counters: per node, and the fallback over all GPUs */
    int svcSpinSet;                /* QZSTD_HIP_SERVICE_SPIN_US was given: no adapting */
    int liveStates;                /* producer states alive (one per CCtx, i.e. per calling thread): more of them than usable cores = the callers
                                    * oversubscribe the cores, and a caller that waits for the GPU should give its core away at once */
    int hintFlags;                 /* QZSTD_HIP_HINT_FLAGS (default 1): an announcement's launch is complete when its blocks' count words are in (0: when
                                    * the runtime says its stream is idle) */
    int hintCompact;               /* QZSTD_HIP_HINT_COMPACT (default 1): announcements' result entries are packed — 8 bytes with a 12-bit tag */
    int stageNt;                   /* QZSTD_HIP_STAGE_NT (default 1): announcements are staged with streaming stores (qzStageCopy) */
    int hintDirect;                /* QZSTD_HIP_HINT_DIRECT: 0 (default) an announcement's staging copy goes to device memory by a copy kernel on the
                                    * launch's stream; 1 the match-finder reads the pinned staging copy itself; 2 that at the levels without
                                    * chains only; 3 the copy by hipMemcpyAsync (rounds 1-3).  Batch front-end, 16 threads, 2 MiB claims, GB/s:
                                    * level 1 21.4 / 18.5 / 18.5 / 12-14, level 3 18.6 / 15.7 (the kernel takes 0.72 ms per launch out of device
                                    * memory, 0.96 ms over the bus; hipMemcpyAsync holds the caller 0.8-1.1 ms per 4 MiB) */
} QZSTD_Process_T;

static QZSTD_Process_T gProc = { .status = QZSTD_FAIL, .coalesce = 1, .timeoutMs = QZ_DEFAULT_TIMEOUT_MS, .splitBlocks = 1, .service = 1,
                                 .svcItemBytes = 4096, .svcSpinUs = 400, .mutex = PTHREAD_MUTEX_INITIALIZER, .numa = 1, .numaThreadNode = -1,
                                 .hintFlags = 1 };

/* One announced buffer: staged in pinned memory, match-found asynchronously — split into contiguous block ranges, one
 * per GPU, each on a slot's stream — results (count + the first QZ_HINT_PITCH sequences of every block) written by
 * the kernels straight into the announcement's pinned host buffers. */
#define QZ_HINT_MAX_BYTES ((size_t)16 << 20)
#define QZ_HINT_PITCH ((size_t)16384) /* blocks with more sequences take the per-block path */
#define QZ_HINT_PARTS 8
#define QZ_COUNT_PENDING 0xFFFFFFFDu /* an announced block's count word until its workgroup publishes it (gProc.hintFlags) */
#define QZ_HINTS 4 /* announcements a state keeps: a ring (QZSTD_hintSource) */
#define QZ_CONTENT_LOOKUP_BLOCKS 256u /* announcements with more grid blocks are matched by address only */
typedef struct {
    int st;   /* 0 none, 1 in flight on the GPU (slot held), 2 ready, 3 failed */
    int slot; /* index of the slot held while in flight */
    size_t b0, b1; /* block range [b0, b1) of the announcement */
    size_t seen;   /* flags: every block below this one has published its count */
    int flags;     /* completion by the blocks' count words (gProc.hintFlags), not by the stream */
} QZSTD_Part_T;

typedef struct {
    int st;   /* 0 empty, 1 announced (parts in flight or ready) */
    int touched; /* a callback was served from it */
    unsigned int epoch; /* count-word completion: the mark (24 bits, never 0) every entry of THIS announcement carries in its fourth word */
    int stable;  /* QZSTD_HINT_STABLE: the announcer holds the bytes still until their callbacks have come (no memcmp per callback) */
    unsigned int seq;   /* the state's announcement number (QZSTD_Session_T.hintSeq): callbacks look at the newest announcement first */
    size_t servedUpTo;  /* grid blocks below this one have been served: a STABLE announcement serves every block once, going forward */
    unsigned misses; /* callbacks that found nothing to serve since the announcement was last used */
    int nParts;
    QZSTD_Part_T part[QZ_HINT_PARTS];
    const unsigned char *base;
    size_t size, block, nb;
    int level;
    unsigned char *hSrc;      /* pinned staging copy of the buffer */
    size_t pitch;             /* result entries per block: min(QZ_HINT_PITCH, ZSTD_sequenceBound(block)) */
    unsigned long long *keys; /* per grid block: size + first / last 8 bytes folded (content look-up for streaming callers) */
    size_t keysCap;
    ZSTD_Sequence *hSeqs;     /* pinned, nb x pitch */
    unsigned int *hCount;     /* pinned */
    qzstd_hip_block_t *hDesc; /* pinned */
    void *dvSeqs, *dvCount, *dvDesc; /* device-side addresses of the three: the kernels use them directly */
    void *dvSrc;                     /* ... and of hSrc (QZSTD_HIP_HINT_DIRECT: the kernel reads the staging copy itself) */
    const void *dvOf[4];             /* the host buffers the four were asked for: asked again only when a buffer was replaced (the
                                      * runtime's look-up takes its global lock: ~0.5 ms per call with 16 announcing threads) */
    size_t hSrcCap, hSeqsCap, hCountCap, hDescCap; /* bytes */
    int nStuck, stuckSlot[QZ_HINT_PARTS]; /* slots whose wait timed out: a kernel may still read and write the buffers above */
    int noAddr; /* a newer announcement names (some of) these addresses: this one no longer serves by address, only by verified content */
    int packedLast; /* the form of the entries the result area held last (1 packed, 0 sixteen-byte): a change of form (the device layer restarted with
                     * another QZSTD_HIP_HINT_COMPACT) wipes it, as a lap of the epochs does */
} QZSTD_Hint_T;

/* Pinned buffers of an announcement that a timed-out kernel may still read (hSrc, hDesc) and write (hSeqs, hCount): the
 * announcement gets fresh ones, these are parked here — never reused, scrubbed or freed — until the streams of the slots
 * involved have drained (round-2 ADVICE: a late kernel must not write into the next announcement's results). */
typedef struct QZSTD_Orphan_S {
    struct QZSTD_Orphan_S *next;
    void *buf[4];
    size_t srcCap;
    int nSlots, slot[QZ_HINT_PARTS];
} QZSTD_Orphan_T;
static QZSTD_Orphan_T *qzOrphans;
static unsigned long qzOrphanCount; /* test hook / log: announcements whose buffers were parked */
static pthread_mutex_t qzOrphanMu = PTHREAD_MUTEX_INITIALIZER;

/* Per-CCtx state (opaque to the caller). */
typedef struct {
    int slotHint;
    unsigned int failOffloadCnt;
    /* announced batches served to later callbacks (QZSTD_hintSource): so that the GPU can
     * work on the next buffer while libzstd entropy-codes the current one on this thread */
    QZSTD_Hint_T hint[QZ_HINTS]; /* announced by the caller, a ring of four */
    int hintNext;                /* the ring slot the next announcement tries first */
    unsigned int hintSeq;        /* announcements made so far: every one gets the next number (the look-up goes newest first) */
    unsigned int stableSampler;  /* blocks looked up in STABLE announcements: every 16th is compared with the staged copy all the same */
    unsigned long stableBroken;  /* ... and differed (QZSTD_hintBroken) */
    unsigned long servedFromBatch, servedSync, servedService;
    unsigned long fail[QZ_CAUSE_N]; /* callbacks that returned the error code, by cause ([0] = all of them) */
    unsigned long redoneAlone;      /* blocks too dense for a batch's result area, redone on a slot of their own (not errors) */
    unsigned long hintCalls, hintStageNs, hintQueueNs, hintWaitNs, hintCopyCallNs, hintLaunchCallNs, hintPrepNs, hintDropNs; /* event log only */
} QZSTD_Session_T;

#define QZ_HINT_STALE_MISSES 16u /* an announcement that was used and then missed this often is dropped */
static void qzReapOrphans(int force);
static size_t qzFailed(QZSTD_Session_T *s, int cause);
static void *qzGrowDev(int dev, void *old, size_t *cap, size_t need);
static int qzUsableCores(void);

/* cheap fingerprint of a block: its size and its first and last 8 bytes (a candidate is always verified with memcmp) */
static unsigned long long qzBlockKey(const unsigned char *p, size_t n)
{
    unsigned long long a, b;
    memcpy(&a, p, 8);
    memcpy(&b, p + n - 8, 8);
    return a ^ (b * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)n << 40);
}

const char *QZSTD_version(void)
{
    return QZSTD_VERSION;
}

static unsigned long qzNowNs(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long)ts.tv_sec * 1000000000ul + (unsigned long)ts.tv_nsec;
}

/* Bounded wait for a stream (reference: the polling loop with its 2 s limit, src/qatseqprod.c:1261-1285).
 * 0 = everything queued on the stream is done, 1 = timed out (the work is still running: the caller must not touch
 * the buffers involved and marks their owner as stuck), -1 = the runtime reported an error. */
static int qzWait(int dev, void *stream)
{
    const int r = qzstd_hip_stream_wait(dev, stream, (unsigned)gProc.timeoutMs);
    if (r == 1) { qzCause = QZ_CAUSE_TIMEOUT; QZ_LOG(1, "device %d: request timed out after %d ms\n", dev, gProc.timeoutMs); }
    if (r < 0) { qzCause = QZ_CAUSE_RUNTIME; QZ_LOG(1, "device %d: %s\n", dev, qzstd_hip_last_error()); }
    return r;
}

/* A stream that timed out is quarantined; it becomes usable again once a non-blocking query finds it drained. */
static int qzStillStuck(int dev, void *stream, int *stuck)
{
    if (!*stuck) return 0;
    if (qzstd_hip_stream_query(dev, stream) == 0) {
        *stuck = 0;
        QZ_LOG(1, "device %d: a timed-out stream has drained, back in service\n", dev);
        return 0;
    }
    return 1;
}

/* ---------------------------------------------------------------- slots ---------- */

static void qzFreeSlot(QZSTD_Slot_T *s)
{
    if (s->stream) (void)qzstd_hip_stream_wait(s->device, s->stream, (unsigned)gProc.timeoutMs);
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
    if (s->vSrc) memset(s->vSrc, 0, QZSTD_HIP_BLOCK_MAX + 64); /* staged caller data: scrubbed before the pages go back */
    qzstd_hip_host_free(s->vSrc);
    qzstd_hip_host_free(s->vSeqs);
    qzstd_hip_host_free(s->vCount);
    qzstd_hip_free(s->device, s->vdSrc);
    qzstd_hip_free(s->device, s->vdWork);
    if (s->stream) qzstd_hip_stream_destroy(s->device, s->stream);
    {
        const int dev = s->device;
        memset(s, 0, sizeof(*s));
        s->device = dev;
    }
}

/* ---- NUMA placement ---- */
/* pinned host memory that device `dev` reads or writes: on the NUMA node the GPU is attached to (reference: qaeMemAllocNUMA on the
 * instance's node, src/qatseqprod.c:216-246).  QZSTD_HIP_NUMA=0 or an unknown node: wherever the calling thread's policy puts it */
static void *qzHostAlloc(size_t bytes, int dev, int coherent)
{
    const int node = (gProc.numa && dev >= 0 && dev < QZ_MAX_DEVICES) ? gProc.devNode[dev] : -1;
    return qzstd_hip_host_alloc_on_node(bytes, node, coherent);
}

/* the NUMA node the calling thread runs on right now (-1 = unknown) */
static int qzThreadNode(void)
{
    unsigned int cpu = 0, node = 0;
    if (gProc.numaThreadNode >= 0) return gProc.numaThreadNode;
#ifdef SYS_getcpu
    if (syscall(SYS_getcpu, &cpu, &node, NULL) == 0) return (int)node;
#endif
    (void)cpu;
    return -1;
}

/* The sticky hint of a state at its first use: hint % numDevices is its GPU, hint / numDevices where its slot sweeps start.  A GPU of
 * the socket the calling thread runs on if there is one (round-robin among those), else round-robin over all GPUs — the reference
 * interleaves its instances across devices (src/qatseqprod.c:601-630) and leaves locality to the instance's node. */
static int qzPickHint(void)
{
    const int nd = gProc.numDevices, node = gProc.numa ? qzThreadNode() : -1;
    if (nd <= 0) return 0;
    if (node >= 0) {
        int local[QZ_MAX_DEVICES], n = 0, d;
        for (d = 0; d < nd && d < QZ_MAX_DEVICES; d++) if (gProc.devNode[d] == node) local[n++] = d;
        if (n > 0 && n < nd) { /* (every GPU on this node = plain round-robin below) */
            const unsigned int k = __sync_fetch_and_add(&gProc.nodeNext[(unsigned)node % QZ_NUMA_NODES_MAX], 1u) & 0xFFFFFu;
            return local[k % (unsigned)n] + nd * (int)(k / (unsigned)n);
        }
    }
    return (int)(__sync_fetch_and_add(&gProc.anyNext, 1u) & 0x3FFFFFFFu);
}

/* lazy per-slot setup, first use only (reference: QZSTD_allocInstMem, :685-822); `full` also creates the buffers of the
 * one-block path (an announced batch only needs the stream and the grow-only batch buffers) */
static int qzSetupSlot(QZSTD_Slot_T *s, int full)
{
    if (!s->stream) {
        s->stream = qzstd_hip_stream_create(s->device);
        if (!s->stream) {
            QZ_LOG(1, "slot setup failed on device %d: %s\n", s->device, qzstd_hip_last_error());
            return QZSTD_FAIL;
        }
    }
    if (!full || s->ready) return QZSTD_OK;
    s->seqCap = qzstd_hip_sequence_bound(QZSTD_HIP_BLOCK_MAX);
    s->hSrc = (unsigned char *)qzHostAlloc(QZSTD_HIP_BLOCK_MAX + 64, s->device, 0);
    s->hSeqs = (ZSTD_Sequence *)qzHostAlloc(s->seqCap * sizeof(ZSTD_Sequence), s->device, 0);
    s->hDesc = (qzstd_hip_block_t *)qzHostAlloc(sizeof(qzstd_hip_block_t), s->device, 0);
    s->hCount = (unsigned int *)qzHostAlloc(64, s->device, 0);
    s->dSrc = (unsigned char *)qzstd_hip_malloc(s->device, QZSTD_HIP_BLOCK_MAX + 64);
    s->dSeqs = (ZSTD_Sequence *)qzstd_hip_malloc(s->device, s->seqCap * sizeof(ZSTD_Sequence));
    s->dDesc = (qzstd_hip_block_t *)qzstd_hip_malloc(s->device, sizeof(qzstd_hip_block_t));
    s->dCount = (unsigned int *)qzstd_hip_malloc(s->device, 64);
    if (!s->hSrc || !s->hSeqs || !s->hDesc || !s->hCount || !s->dSrc || !s->dSeqs || !s->dDesc || !s->dCount) {
        QZ_LOG(1, "slot setup failed on device %d: %s\n", s->device, qzstd_hip_last_error());
        qzFreeSlot(s);
        return QZSTD_FAIL;
    }
    s->ready = 1;
    return QZSTD_OK;
}

/* one quick sweep over the slots (no waiting), starting at the caller's sticky hint; dev >= 0 confines it to the slots of
 * that GPU (slot i sits on device i % numDevices) */
static int qzTryGrabSlot(int hint, int dev)
{
    int k;
    const int n = gProc.numSlots, nd = gProc.numDevices;
    if (n <= 0) return -1;
    if (hint < 0 || hint >= n) hint = 0;
    if (dev >= 0) hint = hint - hint % nd + dev; /* same row, that device's column */
    for (k = 0; k < n; k += dev >= 0 ? nd : 1) {
        const int i = (hint + k) % n;
        if (__sync_lock_test_and_set(&gProc.slots[i].lock, 1) == 0) return i;
    }
    return -1;
}

/* test-and-set sweep over the slots, starting at the caller's sticky hint
 * (reference: QZSTD_grabInstance, :905-928) */
static int qzGrabSlot(int hint, int dev)
{
    int sweep;
    /* The reference sweeps its instances 10 times and then fails the block (src/qatseqprod.c:905-928, :915);
     * here a caller WAITS for a slot: short spins first, then 50 us naps, giving up only after about two seconds
     * (the reference's own time-out for a stuck request, :1261-1285) */
    for (sweep = 0; sweep < QZ_GRAB_SWEEPS; sweep++) {
        const int i = qzTryGrabSlot(hint, dev);
        if (i >= 0 || gProc.numSlots <= 0) return i;
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

static void qzFreeBatch(QZSTD_Coalescer_T *c, QZSTD_Batch_T *bt)
{
    if (bt->stream) (void)qzstd_hip_stream_wait(c->device, bt->stream, (unsigned)gProc.timeoutMs);
    qzstd_hip_host_free(bt->hSrc);
    qzstd_hip_host_free(bt->hSeqs);
    qzstd_hip_host_free(bt->hDesc);
    qzstd_hip_host_free(bt->hCount);
    qzstd_hip_free(c->device, bt->dSrc);
    qzstd_hip_free(c->device, bt->dWork);
    if (bt->stream) qzstd_hip_stream_destroy(c->device, bt->stream);
    bt->hSrc = NULL; bt->hSeqs = NULL; bt->hDesc = NULL; bt->hCount = NULL; bt->dSrc = NULL; bt->dWork = NULL;
    bt->dWorkCap = 0; bt->stream = NULL; bt->dvSeqs = bt->dvDesc = bt->dvCount = NULL;
    bt->ready = 0;
}

static void qzFreeCoalescer(QZSTD_Coalescer_T *c)
{
    int b;
    for (b = 0; b < QZ_BATCHES; b++) qzFreeBatch(c, &c->batch[b]);
    QZ_LOG(2, "device %d: %lu block(s) in %lu batch(es), %lu launch(es)\n", c->device, c->blocks, c->batches, c->launches);
    pthread_mutex_destroy(&c->mu);
    pthread_cond_destroy(&c->cvOpen);
    for (b = 0; b < QZ_BATCHES; b++) {
        pthread_cond_destroy(&c->batch[b].cvLead);
        pthread_cond_destroy(&c->batch[b].cvDone);
    }
}

/* lazy, per batch, under c->mu; a failure frees what was created (nothing leaks when pinned memory is short) */
static int qzSetupBatch(QZSTD_Coalescer_T *c, QZSTD_Batch_T *bt)
{
    if (bt->ready) return QZSTD_OK;
    bt->stream = qzstd_hip_stream_create(c->device);
    bt->dSrc = (unsigned char *)qzstd_hip_malloc(c->device, QZ_BATCH_MAX * QZ_SRC_STRIDE);
    bt->hSrc = (unsigned char *)qzHostAlloc(QZ_BATCH_MAX * QZ_SRC_STRIDE, c->device, 0);
    bt->hSeqs = (ZSTD_Sequence *)qzHostAlloc(QZ_BATCH_MAX * QZ_BATCH_PITCH * sizeof(ZSTD_Sequence), c->device, 0);
    bt->hDesc = (qzstd_hip_block_t *)qzHostAlloc(QZ_DESC_MAX * sizeof(qzstd_hip_block_t), c->device, 0);
    bt->hCount = (unsigned int *)qzHostAlloc(QZ_DESC_MAX * sizeof(unsigned int), c->device, 0);
    bt->dvSeqs = qzstd_hip_host_device_ptr(bt->hSeqs);
    bt->dvDesc = qzstd_hip_host_device_ptr(bt->hDesc);
    bt->dvCount = qzstd_hip_host_device_ptr(bt->hCount);
    if (!(bt->stream && bt->dSrc && bt->hSrc && bt->hSeqs && bt->hDesc && bt->hCount && bt->dvSeqs && bt->dvDesc && bt->dvCount)) {
        QZ_LOG(1, "batch setup failed on device %d: %s\n", c->device, qzstd_hip_last_error());
        qzFreeBatch(c, bt);
        return QZSTD_FAIL;
    }
    bt->ready = 1;
    return QZSTD_OK;
}

/* the leader's job: one launch per level present in the batch (called WITHOUT c->mu held).
 * Lone-request latency: at the levels whose profile allows it (segLog: no match crosses a 32 KiB boundary) a block goes
 * to the GPU as up to four SEGMENT work items — each workgroup inserts the block before its segment into its tables and
 * parses its segment only — which finish in about 40 % of the time one workgroup needs for the whole block; the
 * caller joins the segments' sequence lists (identical to the whole block's list by construction). */
static void qzRunBatch(QZSTD_Coalescer_T *c, QZSTD_Batch_T *bt)
{
    const int n = bt->n, dev = c->device;
    int order[QZ_BATCH_MAX], first[QZ_BATCH_MAX];
    int i, j, k = 0, g0, failed = 0, launches = 0;
    /* requests grouped by level (insertion sort, stable: the usual batch has one level) */
    for (i = 0; i < n; i++) {
        for (j = i; j > 0 && bt->req[order[j - 1]].level > bt->req[i].level; j--) order[j] = order[j - 1];
        order[j] = i;
    }
    for (j = 0; j < n; j++) {
        QZSTD_Req_T *r = &bt->req[order[j]];
        qzstd_hip_profile_t pf;
        size_t seg = 0;
        int sg;
        r->nSeg = 1;
        r->dense = 0;
        /* a full batch fills the GPU as it is: segment items then only repeat the insert work of the block before them
         * (and, at the chain levels, multiply the scratch); they pay for batches that leave CUs idle */
        size_t segsMax = !gProc.splitBlocks ? 1 : (n * 32 <= QZ_SPLIT_ITEMS_MAX ? 32 : (n * 8 <= QZ_SPLIT_ITEMS_MAX ? 8 : 1));
        if (segsMax > 1 && qzstd_hip_profile_for_level(r->level, r->srcSize, &pf) == 0 && pf.segLog) {
            /* chain levels: every item of a LAUNCH links the block before it in a scratch of its own (20 B per position of the whole
             * block: the workgroups of a launch do not start in order, so they cannot share one as the service's items do) — 32 items
             * per block would grow the batch's grow-only scratch to 32 x 2.6 MB per caller; eight keep most of the latency gain */
            if (pf.chainDepth && segsMax > 8) segsMax = 8;
            seg = (size_t)1 << pf.segLog;
            while ((r->srcSize + seg - 1) / seg > segsMax) seg *= 2; /* whole segments per item */
            if (r->srcSize > seg) r->nSeg = (int)((r->srcSize + seg - 1) / seg);
        }
        r->segPitch = (unsigned int)(QZ_BATCH_PITCH / (size_t)r->nSeg);
        first[j] = k;
        for (sg = 0; sg < r->nSeg; sg++, k++) {
            qzstd_hip_block_t *d = &bt->hDesc[k];
            d->srcOff = (size_t)order[j] * QZ_SRC_STRIDE;
            d->mark = 0;
            if (r->nSeg == 1) {
                d->seqOff = (size_t)order[j] * QZ_BATCH_PITCH;
                d->srcLen = (unsigned int)r->srcSize;
                d->seqCap = (unsigned int)(r->cap < QZ_BATCH_PITCH ? r->cap : QZ_BATCH_PITCH);
                d->parseFrom = 0;
            } else { /* segment sg: the block up to the segment's end, parsed from the segment's start */
                const size_t end = (size_t)(sg + 1) * seg;
                d->seqOff = (size_t)order[j] * QZ_BATCH_PITCH + (size_t)sg * r->segPitch;
                d->srcLen = (unsigned int)(end < r->srcSize ? end : r->srcSize);
                d->seqCap = r->segPitch;
                d->parseFrom = (unsigned int)((size_t)sg * seg);
            }
        }
    }
    /* one copy in, one launch per level, one wait: the kernel reads the descriptors from and writes the sequences and
     * counts to this batch's pinned host buffers directly (posted PCIe writes while it runs), which takes two
     * copies and one synchronisation off the latency of a request */
    failed = qzstd_hip_memcpy_h2d(dev, bt->stream, bt->dSrc, bt->hSrc, (size_t)n * QZ_SRC_STRIDE);
    for (g0 = 0; g0 < n && !failed; ) {
        const int level = bt->req[order[g0]].level;
        unsigned int maxLen = 0;
        int g1 = g0, k0 = first[g0], k1;
        size_t work;
        while (g1 < n && bt->req[order[g1]].level == level) {
            if (bt->req[order[g1]].srcSize > maxLen) maxLen = (unsigned int)bt->req[order[g1]].srcSize;
            g1++;
        }
        k1 = g1 < n ? first[g1] : k;
        work = qzstd_hip_workspace_bytes(level, (unsigned int)(k1 - k0), maxLen);
        if (work > bt->dWorkCap && launches) { /* the scratch is about to be replaced: what runs on it has to finish first */
            const int w = qzWait(dev, bt->stream);
            if (w == 1) bt->stuck = 1;
            failed = w != 0;
        }
        if (work && !failed) bt->dWork = qzGrowDev(dev, bt->dWork, &bt->dWorkCap, work);
        failed = failed || (work && !bt->dWork) ||
                 qzstd_hip_find_sequences(dev, bt->stream, level, bt->dSrc, (const qzstd_hip_block_t *)bt->dvDesc + k0,
                                          (unsigned int)(k1 - k0), maxLen, bt->dvSeqs, (unsigned int *)bt->dvCount + k0,
                                          bt->dWork, bt->dWorkCap);
        launches++;
        g0 = g1;
    }
    /* whatever was queued — the copy, the launches of earlier level groups — has to be off the stream before the batch's
     * buffers are handed back, also when a later step failed (round-2 ADVICE) */
    if (!bt->stuck) {
        const int w = qzWait(dev, bt->stream);
        if (w == 1) bt->stuck = 1; /* the kernel may still write into this batch's buffers: quarantined */
        failed = failed || w != 0;
    }
    {
        const int cause = bt->stuck ? QZ_CAUSE_TIMEOUT : QZ_CAUSE_RUNTIME;
        for (j = 0; j < n; j++) bt->req[j].cause = cause;
    }
    for (j = 0; j < n; j++) {
        QZSTD_Req_T *r = &bt->req[order[j]];
        size_t total = 1;
        int sg, bad = failed;
        for (sg = 0; sg < r->nSeg && !bad; sg++) {
            const unsigned int cnt = bt->hCount[first[j] + sg];
            r->segCnt[sg] = cnt;
            if (cnt == QZSTD_HIP_NSEQ_ERROR || cnt == 0) bad = 1;
            else total += cnt - 1;
        }
        /* capacity rule, reference :1318-1322; a result area of the batch that was too small (not the caller's
         * capacity) means: redo this block alone */
        if (bad) r->dense = !failed && (r->nSeg > 1 || r->cap > QZ_BATCH_PITCH);
        if (!failed) r->cause = QZ_CAUSE_CAPACITY;
        r->rc = (bad || total >= r->cap - 1) ? ZSTD_SEQUENCE_PRODUCER_ERROR : total;
    }
    if (failed) QZ_LOG(1, "device request failed: %s\n", qzstd_hip_last_error());
    c->launches += (unsigned long)launches;
    c->batches++;
    c->blocks += (unsigned long)n;
    __atomic_fetch_add(&gProc.devBlocks[dev][1], (unsigned long)n, __ATOMIC_RELAXED);
}

static size_t qzSlotBlock(QZSTD_Session_T *s, int dev, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src,
                          size_t srcSize, int level);

/* one block through the coalescer of device `dev`; returns the sequence count or the error code */
static size_t qzCoalescedBlock(QZSTD_Session_T *s, int dev, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src,
                               size_t srcSize, int level)
{
    QZSTD_Coalescer_T *c = &gProc.coal[dev];
    QZSTD_Batch_T *bt = NULL;
    size_t rc;
    int i, b, dense;

    pthread_mutex_lock(&c->mu);
    for (;;) {
        int usable = 0;
        if (c->open >= 0 && c->batch[c->open].state == 0 && c->batch[c->open].n < QZ_BATCH_MAX) { /* join the collecting batch */
            bt = &c->batch[c->open];
            break;
        }
        if (c->open < 0) { /* nobody is collecting: open an idle batch and lead it */
            for (b = 0; b < QZ_BATCHES && !bt; b++) {
                QZSTD_Batch_T *cand = &c->batch[b];
                if (cand->state != 0 || cand->n != 0) { usable++; continue; } /* busy, but it will come back */
                if (qzStillStuck(dev, cand->stream, &cand->stuck)) continue;
                if (qzSetupBatch(c, cand) != QZSTD_OK) continue;
                bt = cand;
                c->open = b;
            }
            if (bt) break;
            if (!usable) { /* every batch is stuck or cannot be set up: fail the block (libzstd's fallback takes over) */
                pthread_mutex_unlock(&c->mu);
                qzCause = QZ_CAUSE_RUNTIME;
                return ZSTD_SEQUENCE_PRODUCER_ERROR;
            }
        }
        pthread_cond_wait(&c->cvOpen, &c->mu);
    }
    i = bt->n++;
    bt->req[i].src = src;
    bt->req[i].srcSize = srcSize;
    bt->req[i].cap = outSeqsCapacity;
    bt->req[i].rc = ZSTD_SEQUENCE_PRODUCER_ERROR;
    bt->req[i].level = level;
    pthread_mutex_unlock(&c->mu);

    memcpy(bt->hSrc + (size_t)i * QZ_SRC_STRIDE, src, srcSize); /* staging copy (reference :1223), on the caller's thread */

    pthread_mutex_lock(&c->mu);
    bt->copied++;
    if (i == 0) {
        /* the first member leads its batch: whoever arrived while it was staging has joined; close and launch.
         * Every waiter has its own condition variable (no thundering herd when more threads than cores wait here). */
        bt->state = 1;
        c->open = -1; /* newcomers open another batch, or wait for one to come back */
        pthread_cond_broadcast(&c->cvOpen);
        while (bt->copied < bt->n) pthread_cond_wait(&bt->cvLead, &c->mu); /* members still staging */
        pthread_mutex_unlock(&c->mu);
        qzRunBatch(c, bt);
        pthread_mutex_lock(&c->mu);
        bt->state = 2;
        pthread_cond_broadcast(&bt->cvDone);
    } else {
        if (bt->state == 1 && bt->copied == bt->n) pthread_cond_signal(&bt->cvLead);
        while (bt->state != 2) pthread_cond_wait(&bt->cvDone, &c->mu);
    }
    pthread_mutex_unlock(&c->mu);

    rc = bt->req[i].rc;
    if (rc == ZSTD_SEQUENCE_PRODUCER_ERROR) qzCause = bt->req[i].cause;
    if (rc != ZSTD_SEQUENCE_PRODUCER_ERROR) {
        const QZSTD_Req_T *r = &bt->req[i];
        if (r->nSeg == 1) {
            memcpy(outSeqs, bt->hSeqs + (size_t)i * QZ_BATCH_PITCH, rc * sizeof(ZSTD_Sequence));
        } else { /* join the segments: the trailing literals of one flow into the first sequence of the next */
            size_t out = 0, carry = 0;
            int sg;
            for (sg = 0; sg < r->nSeg; sg++) {
                const ZSTD_Sequence *q = bt->hSeqs + (size_t)i * QZ_BATCH_PITCH + (size_t)sg * r->segPitch;
                const size_t count = r->segCnt[sg];
                if (count > 1) {
                    memcpy(outSeqs + out, q, (count - 1) * sizeof(ZSTD_Sequence));
                    outSeqs[out].litLength += (unsigned int)carry;
                    out += count - 1;
                    carry = 0;
                }
                carry += q[count - 1].litLength;
            }
            outSeqs[out].offset = 0;
            outSeqs[out].litLength = (unsigned int)carry;
            outSeqs[out].matchLength = 0;
            outSeqs[out].rep = 0;
        }
    }
    dense = rc == ZSTD_SEQUENCE_PRODUCER_ERROR && bt->req[i].dense && !bt->stuck;

    pthread_mutex_lock(&c->mu);
    if (++bt->consumed == bt->n) { /* last one out hands the batch back */
        bt->n = bt->copied = bt->consumed = 0;
        bt->state = 0;
        pthread_cond_broadcast(&c->cvOpen);
    }
    pthread_mutex_unlock(&c->mu);
    /* a block with more sequences than the batch's result pitch holds (incompressible-looking data with many
     * short matches) is redone alone with the caller's full capacity */
    if (dense) {
        s->redoneAlone++;
        rc = qzSlotBlock(s, dev, outSeqs, outSeqsCapacity, src, srcSize, level);
    }
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
    int perDev, i, b, maxDev;
    if (nDev <= 0) return QZSTD_FAIL;
    maxDev = qzEnvInt("QZSTD_HIP_MAX_DEVICES", QZ_MAX_DEVICES, 1, QZ_MAX_DEVICES);
    if (nDev > maxDev) nDev = maxDev;
    perDev = qzEnvInt("QZSTD_HIP_SLOTS", QZ_DEFAULT_SLOTS_PER_DEVICE, 1, QZ_MAX_SLOTS / nDev);
    gProc.slots = (QZSTD_Slot_T *)calloc((size_t)nDev * perDev, sizeof(QZSTD_Slot_T));
    if (!gProc.slots) return QZSTD_FAIL;
    gProc.numDevices = nDev;
    gProc.numSlots = nDev * perDev;
    memset(gProc.devBlocks, 0, sizeof(gProc.devBlocks));
    for (i = 0; i < gProc.numSlots; i++) gProc.slots[i].device = i % nDev;
    gProc.numa = qzEnvInt("QZSTD_HIP_NUMA", 1, 0, 1);
    gProc.numaThreadNode = qzEnvInt("QZSTD_HIP_NUMA_NODE", -1, -1, 1023);
    memset(gProc.nodeNext, 0, sizeof(gProc.nodeNext));
    gProc.anyNext = 0;
    for (i = 0; i < nDev && i < QZ_MAX_DEVICES; i++) {
        gProc.devNode[i] = qzstd_hip_device_numa_node(i);
        QZ_LOG(2, "device %d: host NUMA node %d\n", i, gProc.devNode[i]);
    }
    gProc.coalesce = qzEnvInt("QZSTD_HIP_COALESCE", 1, 0, 1);
    gProc.split = qzEnvInt("QZSTD_HIP_SPLIT", nDev < QZ_HINT_PARTS ? nDev : QZ_HINT_PARTS, 1, nDev < QZ_HINT_PARTS ? nDev : QZ_HINT_PARTS);
    gProc.coal = (QZSTD_Coalescer_T *)calloc((size_t)nDev, sizeof(QZSTD_Coalescer_T));
    if (!gProc.coal) return QZSTD_FAIL;
    for (i = 0; i < nDev; i++) {
        gProc.coal[i].device = i;
        gProc.coal[i].open = -1;
        pthread_mutex_init(&gProc.coal[i].mu, NULL);
        pthread_cond_init(&gProc.coal[i].cvOpen, NULL);
        for (b = 0; b < QZ_BATCHES; b++) {
            pthread_cond_init(&gProc.coal[i].batch[b].cvLead, NULL);
            pthread_cond_init(&gProc.coal[i].batch[b].cvDone, NULL);
        }
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
        gProc.timeoutMs = qzEnvInt("QZSTD_HIP_TIMEOUT_MS", QZ_DEFAULT_TIMEOUT_MS, 1, 600000);
        gProc.splitBlocks = qzEnvInt("QZSTD_HIP_SPLIT_BLOCKS", 1, 0, 1);
        gProc.service = qzEnvInt("QZSTD_HIP_SERVICE", 1, 0, 1);
        gProc.svcItemBytes = qzEnvInt("QZSTD_HIP_SERVICE_ITEM", 4096, 4096, (int)QZSTD_HIP_BLOCK_MAX) & ~4095;
        gProc.svcSpinUs = qzEnvInt("QZSTD_HIP_SERVICE_SPIN_US", 400, 0, 1000000);
        gProc.svcSpinSet = getenv("QZSTD_HIP_SERVICE_SPIN_US") != NULL;
        gProc.hintFlags = qzEnvInt("QZSTD_HIP_HINT_FLAGS", 1, 0, 1);
        /* announcements: PACKED result entries (qzstd_hip.h: QZSTD_HIP_MARK_COMPACT) — 8 bytes per sequence over PCIe instead of 16 (round 6: the
         * level-1 kernel ran at the bus's write rate); needs the count-word completion (every entry certifies itself).  0 = 16-byte entries (A/B) */
        gProc.hintCompact = gProc.hintFlags ? qzEnvInt("QZSTD_HIP_HINT_COMPACT", 1, 0, 1) : 0;
        gProc.stageNt = qzEnvInt("QZSTD_HIP_STAGE_NT", 1, 0, 1);
        gProc.hintDirect = qzEnvInt("QZSTD_HIP_HINT_DIRECT", 0, 0, 3);
        (void)qzUsableCores(); /* (read once here, under the process mutex: the waits only load it) */
    }
    if (gProc.status == QZSTD_FAIL) {
        /* runtime up? (reference: QZSTD_salUserStart, :498-527) */
        gProc.status = qzstd_hip_device_count() > 0 ? QZSTD_STARTED : QZSTD_FAIL;
        if (gProc.status == QZSTD_FAIL) QZ_LOG(2, "no usable HIP device: %s\n", qzstd_hip_last_error());
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
    for (i = 0; i < gProc.numDevices; i++) (void)qzstd_hip_service_stop(i); /* the resident kernels leave before anything is freed */
    if (gProc.slots) {
        for (i = 0; i < gProc.numSlots; i++) qzFreeSlot(&gProc.slots[i]); /* waits (bounded) for each stream */
        qzReapOrphans(1);
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

/* the cores this process may use: its affinity mask, capped by a cgroup CPU quota (cpu.max) */
static int qzUsableCores(void)
{
    static int cached; /* written once with the same value by whoever gets here first (QZSTD_startQatDevice does, under the process mutex) */
    int n = 0;
    cpu_set_t set;
    FILE *f;
    if ((n = __atomic_load_n(&cached, __ATOMIC_RELAXED)) != 0) return n;
    n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)sysconf(_SC_NPROCESSORS_ONLN);
    f = fopen("/sys/fs/cgroup/cpu.max", "re");
    if (f) {
        char q[32] = "";
        long per = 0;
        if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
            const long quota = (atol(q) + per - 1) / per;
            if (quota > 0 && quota < n) n = (int)quota;
        }
        fclose(f);
    }
    n = n > 0 ? n : 1;
    __atomic_store_n(&cached, n, __ATOMIC_RELAXED);
    return n;
}

void *QZSTD_createSeqProdState(void)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)calloc(1, sizeof(QZSTD_Session_T));
    if (!s) return NULL;
    s->slotHint = -1;
    __atomic_fetch_add(&gProc.liveStates, 1, __ATOMIC_RELAXED);
    return s;
}

/* frees (force) or tries to free the parked buffers whose streams have drained; scrubs the staged caller data first */
static void qzReapOrphans(int force)
{
    QZSTD_Orphan_T **pp;
    pthread_mutex_lock(&qzOrphanMu);
    for (pp = &qzOrphans; *pp; ) {
        QZSTD_Orphan_T *o = *pp;
        int k, busy = 0;
        for (k = 0; k < o->nSlots && !force; k++) {
            const int i = o->slot[k];
            if (gProc.slots && i >= 0 && i < gProc.numSlots && gProc.slots[i].stream &&
                qzstd_hip_stream_query(gProc.slots[i].device, gProc.slots[i].stream) != 0)
                busy = 1;
        }
        if (busy) { pp = &o->next; continue; }
        if (o->buf[0]) memset(o->buf[0], 0, o->srcCap);
        for (k = 0; k < 4; k++) qzstd_hip_host_free(o->buf[k]);
        *pp = o->next;
        free(o);
    }
    pthread_mutex_unlock(&qzOrphanMu);
}

/* the announcement's buffers may still be in use by a kernel that timed out: park them, the announcement starts afresh */
static void qzOrphanHint(QZSTD_Hint_T *h)
{
    QZSTD_Orphan_T *o = (QZSTD_Orphan_T *)calloc(1, sizeof(*o));
    int k;
    if (o) { /* (no memory for the list node: the buffers are leaked rather than freed under a running kernel) */
        o->buf[0] = h->hSrc; o->buf[1] = h->hSeqs; o->buf[2] = h->hCount; o->buf[3] = h->hDesc;
        o->srcCap = h->hSrcCap;
        o->nSlots = h->nStuck;
        for (k = 0; k < h->nStuck; k++) o->slot[k] = h->stuckSlot[k];
        pthread_mutex_lock(&qzOrphanMu);
        o->next = qzOrphans;
        qzOrphans = o;
        qzOrphanCount++;
        pthread_mutex_unlock(&qzOrphanMu);
    }
    h->hSrc = NULL; h->hSeqs = NULL; h->hCount = NULL; h->hDesc = NULL;
    h->dvSeqs = h->dvCount = h->dvDesc = h->dvSrc = NULL;
    h->hSrcCap = h->hSeqsCap = h->hCountCap = h->hDescCap = 0;
    h->nStuck = 0;
    QZ_LOG(1, "an announcement's buffers are parked until a timed-out stream drains\n");
}
#ifdef QZ_TEST_HOOKS /* the mock build of tests/test_host_mock.py only: never in the release library */
unsigned long qzstd_test_orphans(void) { return qzOrphanCount; }
void qzstd_test_set_hint_epochs(void *state, unsigned int e) /* the state's next announcements get epoch e + 1 (the 24-bit wrap) */
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)state;
    int k;
    for (k = 0; s && k < QZ_HINTS; k++) s->hint[k].epoch = e & 0xFFFFFFu;
}
void qzstd_test_set_service_epochs(unsigned int e) /* every slot's next service request gets epoch e + 1 (the 24-bit wrap) */
{
    int i;
    for (i = 0; gProc.slots && i < gProc.numSlots; i++) gProc.slots[i].vEpoch = e & 0xFFFFFFu;
}
#endif

/* Takes n entries of a result area that is completed by count words: the count says how many entries there are, not that they have all
 * arrived — they are stored by other waves than the count and nothing orders them on their ways to host memory (measured in round 3 on
 * the resident service: under load an item's last entries land up to microseconds after its count).  Every entry is ONE 16-byte store
 * that carries the announcement's epoch in its fourth word (qzstd_hip_block_t.mark): an entry is taken when it shows it; the mark is
 * masked out of the copy.  dst may be NULL (the entries are only waited for).  0 = taken, 1 = not within the time-out. */
static int qzTakeMarked(ZSTD_Sequence *dst, const ZSTD_Sequence *q, size_t n, unsigned int epoch)
{
    const __m128i keep = _mm_set_epi32(0, -1, -1, -1);
    unsigned long t0 = 0;
    size_t j;
    for (j = 0; j < n; j++) {
        __m128i v = _mm_load_si128((const __m128i *)(const void *)(q + j));
        if ((unsigned int)_mm_cvtsi128_si32(_mm_srli_si128(v, 12)) != epoch) { /* not there yet: rare */
            unsigned spins = 0;
            if (!t0) t0 = qzNowNs();
            do {
                __builtin_ia32_pause();
                __asm__ volatile("" ::: "memory"); /* (a fresh load every time round) */
                if ((++spins & 1023u) == 0u && qzNowNs() - t0 > (unsigned long)gProc.timeoutMs * 1000000ul) return 1;
                v = _mm_load_si128((const __m128i *)(const void *)(q + j));
            } while ((unsigned int)_mm_cvtsi128_si32(_mm_srli_si128(v, 12)) != epoch);
        }
        if (dst) _mm_storeu_si128((__m128i *)(void *)(dst + j), _mm_and_si128(v, keep));
    }
    return 0;
}

/* The same for PACKED entries (qzstd_hip.h: QZSTD_HIP_MARK_COMPACT): one 8-byte load per entry (it arrived as one store), taken when its top 12
 * bits show the tag, unpacked into the caller's ZSTD_Sequence array.  *covered += literal + match lengths of the entries taken (dst != NULL). */
static int qzTakePacked(ZSTD_Sequence *dst, const unsigned long long *q, size_t n, unsigned int tag)
{
    const __m128i m17 = _mm_set_epi32(0, 0x1FFFF, 0, 0x1FFFF), m18 = _mm_set_epi32(0, 0x3FFFF, 0, 0x3FFFF);
    unsigned long t0 = 0;
    size_t j = 0;
    while (j < n) {
        if (dst && j + 2 <= n && !((uintptr_t)(q + j) & 15u)) { /* two entries per step: the usual case, both there (an ALIGNED 16-byte load: each
                                                                  * 8-byte half is one of the kernel's stores, never torn) */
            const __m128i v = _mm_load_si128((const __m128i *)(const void *)(q + j));
            const __m128i t = _mm_srli_epi64(v, 52);
            if ((unsigned int)_mm_cvtsi128_si32(t) == tag && (unsigned int)_mm_cvtsi128_si32(_mm_srli_si128(t, 8)) == tag) {
                const __m128i ol = _mm_or_si128(_mm_and_si128(v, m17), _mm_slli_epi64(_mm_and_si128(_mm_srli_epi64(v, 17), m18), 32)); /* off0 lit0 off1 lit1 */
                const __m128i ml = _mm_and_si128(_mm_srli_epi64(v, 35), m17);                                                          /* ml0  0    ml1  0    */
                _mm_storeu_si128((__m128i *)(void *)(dst + j), _mm_unpacklo_epi64(ol, ml));
                _mm_storeu_si128((__m128i *)(void *)(dst + j + 1), _mm_unpackhi_epi64(ol, ml));
                j += 2;
                continue;
            }
        }
        {
            unsigned long long v = __atomic_load_n(&q[j], __ATOMIC_RELAXED);
            if (QZSTD_HIP_PACKED_TAG(v) != tag) { /* not there yet: rare */
                unsigned spins = 0;
                if (!t0) t0 = qzNowNs();
                do {
                    __builtin_ia32_pause();
                    if ((++spins & 1023u) == 0u && qzNowNs() - t0 > (unsigned long)gProc.timeoutMs * 1000000ul) return 1;
                    v = __atomic_load_n(&q[j], __ATOMIC_RELAXED);
                } while (QZSTD_HIP_PACKED_TAG(v) != tag);
            }
            if (dst) {
                dst[j].offset = QZSTD_HIP_PACKED_OFF(v);
                dst[j].litLength = QZSTD_HIP_PACKED_LIT(v);
                dst[j].matchLength = QZSTD_HIP_PACKED_ML(v);
                dst[j].rep = 0;
            }
            j++;
        }
    }
    return 0;
}

/* Completion of an announcement's launch WITHOUT the runtime (round 4): every workgroup publishes its block's count word — in the
 * announcement's pinned, coherent result area — with a system-scope release after all of its result stores (csrc/qzstd_kernels.hip,
 * qzstd_find_sequences_kernel), and the words start out as QZ_COUNT_PENDING.  A callback waits for ITS blocks only, by reading memory.
 * (hipStreamQuery on one of 48 streams folded onto 16 hardware queues waits for whatever was queued behind the launch it asks about:
 * with the stream waits a worker of the batch front-end spent 30 % of its time waiting for launches that had long finished.)
 * Blocks [b0, b1): 0 = all published, 1 = not within the time-out. */
static int qzBlocksWait(const QZSTD_Hint_T *h, size_t b0, size_t b1)
{
    const unsigned long t0 = qzNowNs(), limit = (unsigned long)gProc.timeoutMs * 1000000ul;
    /* 50 us of polling, then naps: a waiting caller does not burn a core others could entropy-code on; with far more callers than
     * cores (states alive > 1.5 x usable cores) no polling at all (front-end, 64 workers on 16 cores: 4.4 GB/s polling).
     * Decided once, at the first miss (round-4 ADVICE: not by how long the first scan happened to take) */
    unsigned long spinNs = 50000ul;
    int decided = 0;
    size_t b = b0;
    for (;;) {
        unsigned long el;
        while (b < b1 && __atomic_load_n(&h->hCount[b], __ATOMIC_ACQUIRE) != QZ_COUNT_PENDING) b++;
        if (b >= b1) return 0;
        if (!decided) { /* at the first miss, whenever that is */
            decided = 1;
            if (2 * __atomic_load_n(&gProc.liveStates, __ATOMIC_RELAXED) > 3 * qzUsableCores()) spinNs = 0ul;
        }
        el = qzNowNs() - t0;
        if (el > limit) {
            qzCause = QZ_CAUSE_TIMEOUT;
            QZ_LOG(1, "announcement: block %zu still not published after %d ms\n", b, gProc.timeoutMs);
            return 1;
        }
        if (el > spinNs) {
            const struct timespec nap = { 0, el < 20000000ul ? 20000l : 200000l };
            nanosleep(&nap, NULL);
        }
    }
}

/* wait for one part of an announcement and give its slot back; the part becomes ready (2) or failed (3) */
static void qzPartFinish(QZSTD_Hint_T *h, QZSTD_Part_T *pt)
{
    if (pt->st != 1) return;
    if (gProc.slots && pt->slot >= 0 && pt->slot < gProc.numSlots) {
        QZSTD_Slot_T *sl = &gProc.slots[pt->slot];
        const int w = pt->flags ? qzBlocksWait(h, pt->seen > pt->b0 ? pt->seen : pt->b0, pt->b1) : qzWait(sl->device, sl->stream);
        if (w == 0) pt->seen = pt->b1;
        if (w == 1) sl->stuck = 1; /* the slot is given back, but nobody uses it before its stream has drained */
        if (w != 0 && h->nStuck < QZ_HINT_PARTS) h->stuckSlot[h->nStuck++] = pt->slot; /* ... and the kernel may still use h's buffers */
        qzReleaseSlot(pt->slot);
        pt->st = w == 0 ? 2 : 3;
        if (w != 0) QZ_LOG(1, "announced batch failed: %s\n", qzstd_hip_last_error());
    } else {
        pt->st = 3;
    }
}

/* drop an announcement: wait for what is still in flight (its buffers are about to be reused or freed) */
static void qzHintDrop(QZSTD_Hint_T *h)
{
    int k;
    for (k = 0; k < h->nParts; k++) qzPartFinish(h, &h->part[k]);
    if (h->nStuck) qzOrphanHint(h);
    h->nParts = 0;
    h->st = 0;
    h->touched = 0;
    h->misses = 0;
}

void QZSTD_freeSeqProdState(void *sequenceProducerState)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)sequenceProducerState;
    int k;
    if (!s) return;
    __atomic_fetch_sub(&gProc.liveStates, 1, __ATOMIC_RELAXED);
    QZ_LOG(2, "state %p: %lu block(s) served from announcements, %lu per "
              "block; %lu hint(s), timers from the 9th on: drop %.2f ms, buffers %.2f ms, staging %.2f ms, queueing %.2f ms (%.2f ms of it in the copy call, %.2f in the launch call), waited %.2f ms for the GPU\n", (void *)s,
           s->servedFromBatch, s->servedSync, s->hintCalls, s->hintDropNs / 1e6, s->hintPrepNs / 1e6, s->hintStageNs / 1e6,
           s->hintQueueNs / 1e6, s->hintCopyCallNs / 1e6, s->hintLaunchCallNs / 1e6, s->hintWaitNs / 1e6);
    for (k = 0; k < QZ_HINTS; k++) {
        qzHintDrop(&s->hint[k]);
        /* the staged copies are the caller's data (for a guess: bytes it never handed over): scrubbed before the
         * pinned pages go back to the allocator */
        if (s->hint[k].hSrc) memset(s->hint[k].hSrc, 0, s->hint[k].hSrcCap);
        qzstd_hip_host_free(s->hint[k].hSrc);
        qzstd_hip_host_free(s->hint[k].hSeqs);
        qzstd_hip_host_free(s->hint[k].hCount);
        qzstd_hip_host_free(s->hint[k].hDesc);
        free(s->hint[k].keys);
    }
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

/* a callback is about to return the error code: count it by cause (QZSTD_failStats) */
static size_t qzFailed(QZSTD_Session_T *s, int cause)
{
    if (s) {
        s->fail[0]++;
        s->fail[cause > 0 && cause < QZ_CAUSE_N ? cause : QZ_CAUSE_RUNTIME]++;
    }
    return ZSTD_SEQUENCE_PRODUCER_ERROR;
}

/* one block, synchronously, on a slot */
static size_t qzRunBlock(QZSTD_Slot_T *sl, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src,
                         size_t srcSize, int level)
{
    const size_t cap = outSeqsCapacity < sl->seqCap ? outSeqsCapacity : sl->seqCap;
    size_t first, count;
    int w;
    memcpy(sl->hSrc, src, srcSize); /* staging copy, reference :1223 */
    sl->hDesc->srcOff = 0;
    sl->hDesc->seqOff = 0;
    sl->hDesc->srcLen = (unsigned int)srcSize;
    sl->hDesc->seqCap = (unsigned int)cap;
    sl->hDesc->parseFrom = 0;
    sl->hDesc->mark = 0;
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
        qzstd_hip_memcpy_d2h(sl->device, sl->stream, sl->hSeqs, sl->dSeqs, first * sizeof(ZSTD_Sequence)))
        goto fail;
    w = qzWait(sl->device, sl->stream);
    if (w == 1) sl->stuck = 1;
    if (w != 0) goto fail;
    count = *sl->hCount;
    if (count == QZSTD_HIP_NSEQ_ERROR || count == 0 || count >= outSeqsCapacity - 1) {
        QZ_LOG(1, "sequence count %zu does not fit capacity %zu\n", count, outSeqsCapacity);
        qzCause = QZ_CAUSE_CAPACITY;
        return ZSTD_SEQUENCE_PRODUCER_ERROR; /* reference :1318-1322 */
    }
    if (count > first) {
        if (qzstd_hip_memcpy_d2h(sl->device, sl->stream, sl->hSeqs + first, sl->dSeqs + first,
                                 (count - first) * sizeof(ZSTD_Sequence)))
            goto fail;
        w = qzWait(sl->device, sl->stream);
        if (w == 1) sl->stuck = 1;
        if (w != 0) goto fail;
    }
    memcpy(outSeqs, sl->hSeqs, count * sizeof(ZSTD_Sequence));
    return count;
fail:
    QZ_LOG(1, "device request failed: %s\n", qzstd_hip_last_error());
    return ZSTD_SEQUENCE_PRODUCER_ERROR;
}

/* buffers of a slot's service requests, first use only; 0 on success (a failure frees what was created) */
static int qzSetupSlotService(QZSTD_Slot_T *sl)
{
    if (sl->vSrc) return 0;
    sl->vSrc = (unsigned char *)qzHostAlloc(QZSTD_HIP_BLOCK_MAX + 64, sl->device, 1);
    sl->vSeqs = (ZSTD_Sequence *)qzHostAlloc(QZ_SVC_ITEMS_MAX * QZ_SVC_ITEM_CAP * sizeof(ZSTD_Sequence), sl->device, 1);
    if (sl->vSeqs) memset(sl->vSeqs, 0, QZ_SVC_ITEMS_MAX * QZ_SVC_ITEM_CAP * sizeof(ZSTD_Sequence)); /* (fresh pinned pages may hold marked entries of an earlier life) */
    sl->vCount = (unsigned int *)qzHostAlloc(QZ_SVC_ITEMS_MAX * sizeof(unsigned int), sl->device, 1);
    sl->vdSrc = (unsigned char *)qzstd_hip_malloc(sl->device, QZSTD_HIP_BLOCK_MAX + 64);
    if (!sl->vSrc || !sl->vSeqs || !sl->vCount || !sl->vdSrc) {
        QZ_LOG(1, "service buffers of a slot on device %d: %s\n", sl->device, qzstd_hip_last_error());
        qzstd_hip_host_free(sl->vSrc); qzstd_hip_host_free(sl->vSeqs); qzstd_hip_host_free(sl->vCount);
        qzstd_hip_free(sl->device, sl->vdSrc);
        sl->vSrc = NULL; sl->vSeqs = NULL; sl->vCount = NULL; sl->vdSrc = NULL;
        return -1;
    }
    memset(sl->vCount, 0xFF, QZ_SVC_ITEMS_MAX * sizeof(unsigned int)); /* "arrived": nothing is outstanding */
    sl->vItems = 0;
    return 0;
}

/*
 * One block through the resident service (qzstd_hip_service_submit): no launch, no stream — the block is staged in the slot's
 * pinned buffer, one 64-byte request goes into the service's ring, workgroups that are already resident parse the block as
 * up to 32 work items (runs of whole segments) and write sequences and, last, the counts into this slot's pinned memory;
 * the caller polls those count words and joins the items' lists (the same join as the batch path's segments).
 * Reference shape: synchronous submit + poll of one request on one DC instance, src/qatseqprod.c:1243-1272.
 * Returns the count, the error code, or QZ_NOT_SERVED (no slot free right now, the level is not served, the service is
 * down or busy with another level): the caller then takes the launch path.
 */
/* levels >= 5 (exact hash chains): their service requests carry a chain scratch (a launch's scratch is sized by qzstd_hip_workspace_bytes at every level) */
static int qzIsChainLevel(int level)
{
    qzstd_hip_profile_t p;
    return qzstd_hip_profile_for_level(level, QZSTD_HIP_BLOCK_MAX, &p) == 0 && p.chainDepth != 0;
}

static size_t qzServiceBlock(QZSTD_Session_T *s, int dev, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src,
                             size_t srcSize, int level)
{
    QZSTD_Slot_T *sl;
    qzstd_hip_svc_req_t rq;
    size_t itemBytes, nItems, k, out = 0, carry = 0, covered = 0;
    unsigned long t0, spinNs, limitNs;
    int i, rc, rejected = 0, bad = 0, wrong = 0, progressive;

    if (!gProc.service || srcSize == 0) return QZ_NOT_SERVED;
    itemBytes = (size_t)gProc.svcItemBytes;
    while ((srcSize + itemBytes - 1) / itemBytes > QZ_SVC_ITEMS_MAX) itemBytes *= 2;
    nItems = (srcSize + itemBytes - 1) / itemBytes;
    i = qzTryGrabSlot(s->slotHint, dev);
    if (i < 0) return QZ_NOT_SERVED;
    sl = &gProc.slots[i];
    if (sl->vStuck) { /* a request of this slot timed out: usable again once every count word of it has arrived */
        for (k = 0; k < sl->vItems; k++)
            if (__atomic_load_n(&sl->vCount[k], __ATOMIC_ACQUIRE) == 0u) { qzReleaseSlot(i); return QZ_NOT_SERVED; }
        sl->vStuck = 0;
    }
    if (qzSetupSlotService(sl) != 0) { qzReleaseSlot(i); return QZ_NOT_SERVED; }

    /* progressive staging (qzstd_hip.h, round 5): where the workers look at a count word before they read its slice, the request is queued
     * FIRST and the block is copied into the pinned buffer behind it, slice by slice — the staging copy (13 us for 128 KiB) then overlaps
     * the request's way to the first worker (8 us) instead of standing in front of it */
    progressive = qzstd_hip_service_progressive(sl->device) == 1;
    if (!progressive) {
        memcpy(sl->vSrc, src, srcSize); /* staging copy, reference :1223 */
        memset(sl->vSrc + srcSize, 0, ((srcSize + 15) & ~(size_t)15) - srcSize);
    }
    for (k = 0; k < nItems; k++) sl->vCount[k] = progressive ? QZSTD_HIP_NSEQ_STAGING : 0u;
    sl->vItems = (unsigned int)nItems;
    sl->vEpoch = (sl->vEpoch + 1u) & 0xFFFFFFu;
    if (sl->vEpoch == 0u) {
        /* the 24-bit epoch starts over: an entry that no request of the last 2^24 overwrote would show a mark that is valid again.
         * The slot is ours and its previous request is complete (every count word in): wipe the result area once per lap. */
        memset(sl->vSeqs, 0, QZ_SVC_ITEMS_MAX * QZ_SVC_ITEM_CAP * sizeof(ZSTD_Sequence));
        sl->vEpoch = 1u;
    }
    rq.hSrc = sl->vSrc; rq.dSrc = sl->vdSrc; rq.hSeqs = sl->vSeqs; rq.hCount = sl->vCount;
    rq.srcLen = (uint32_t)srcSize; rq.itemBytes = (uint32_t)itemBytes; rq.nItems = (uint32_t)nItems;
    rq.seqCapPerItem = (uint32_t)(QZ_SVC_ITEMS_MAX * QZ_SVC_ITEM_CAP / nItems); /* the slot's whole result area, shared out */
    rq.slot = (uint32_t)i; rq.epoch = sl->vEpoch;
    rq.dWork = NULL;
    if (qzIsChainLevel(level)) { /* every item links the block before it in its own scratch */
        if (!sl->vdWork) sl->vdWork = qzstd_hip_malloc(sl->device, QZSTD_HIP_SVC_WORK_BYTES);
        if (!sl->vdWork) {
            memset(sl->vCount, 0xFF, nItems * sizeof(unsigned int));
            qzReleaseSlot(i);
            return QZ_NOT_SERVED;
        }
        rq.dWork = sl->vdWork;
    }
    rc = qzstd_hip_service_submit(sl->device, level, &rq);
    if (rc != 0) {
        memset(sl->vCount, 0xFF, nItems * sizeof(unsigned int)); /* nothing outstanding */
        qzReleaseSlot(i);
        if (rc > 0) return QZ_NOT_SERVED;
        QZ_LOG(1, "service request not queued: %s\n", qzstd_hip_last_error());
        return QZ_NOT_SERVED; /* the launch path may still work */
    }
    if (progressive) {
        for (k = 0; k < nItems; k++) {
            const size_t from = k * itemBytes, upTo = (k + 1) * itemBytes < srcSize ? (k + 1) * itemBytes : srcSize;
            unsigned int expect = QZSTD_HIP_NSEQ_STAGING;
            memcpy(sl->vSrc + from, (const unsigned char *)src + from, upTo - from);
            if (k + 1 == nItems) memset(sl->vSrc + srcSize, 0, ((srcSize + 15) & ~(size_t)15) - srcSize);
            /* "slice k is in": only a word that still says STAGING (a request the dispatcher handed back has REJECTED there) */
            (void)__atomic_compare_exchange_n(&sl->vCount[k], &expect, 0u, 0, __ATOMIC_RELEASE, __ATOMIC_RELAXED);
        }
    }
    /* poll the count words in item order — the items finish roughly in that order (the later, the more history in front of it) — and
     * JOIN every item's list as it arrives (the trailing literals of one flow into the first sequence of the next): by the time the
     * last item is in, the rest of the block's list stands.  Busy for svcSpinUs, then naps. */
    t0 = qzNowNs();
    /* how long a waiting caller polls before it naps: with a core per caller polling is free and a nap costs its wake-up (level 1, 16 threads on
     * 16 cores: 12.7 GB/s polling 400 us, 10.5 napping at once); with more callers than cores a poller keeps another caller's entropy stage
     * off the core (32 threads: 9.5 GB/s polling 400 us, 14.2-14.6 polling 10 us or not at all; 48 threads: 5.6 against 15.1-15.5) */
    spinNs = (unsigned long)gProc.svcSpinUs * 1000ul;
    if (!gProc.svcSpinSet && __atomic_load_n(&gProc.liveStates, __ATOMIC_RELAXED) > qzUsableCores()) spinNs = 10000ul;
    limitNs = (unsigned long)gProc.timeoutMs * 1000000ul;
    for (k = 0; k < nItems; k++) {
        unsigned polls = 0;
        while (__atomic_load_n(&sl->vCount[k], __ATOMIC_ACQUIRE) == 0u) {
            __builtin_ia32_pause();
            if ((++polls & 63u) == 0u) {
                const unsigned long dt = qzNowNs() - t0;
                if (dt > limitNs) { bad = 1; break; }
                /* the service may have left with this request in its ring: it is launched again; or for good (another request timed out) */
                if ((polls & 1023u) == 0u && qzstd_hip_service_poke(sl->device, level) == 2 && __atomic_load_n(&sl->vCount[k], __ATOMIC_ACQUIRE) == 0u) { bad = 1; break; }
                if (dt > spinNs) { const struct timespec nap = { 0, dt < 20000000ul ? 5000 : 200000 }; nanosleep(&nap, NULL); }
            }
        }
        if (bad) break;
        if (!wrong) { /* (after a wrong item the rest is only waited for: the slot's buffers are in use until every count is in) */
            const unsigned int cnt = sl->vCount[k];
            const ZSTD_Sequence *q = sl->vSeqs + k * rq.seqCapPerItem;
            if (cnt == QZSTD_HIP_NSEQ_REJECTED) {
                /* handed back whole: nothing of it was queued.  The dispatcher writes all nItems words with one wave store, but stores
                 * to pinned host memory can land microseconds apart: the slot is given back only when every word is in — a word
                 * landing later would be read by the slot's NEXT request as its own (round-3 ADVICE) */
                size_t m;
                rejected = 1;
                for (m = k + 1; m < nItems && !bad; m++) {
                    unsigned spins = 0;
                    while (__atomic_load_n(&sl->vCount[m], __ATOMIC_ACQUIRE) == 0u) {
                        __builtin_ia32_pause();
                        if ((++spins & 1023u) == 0u && qzNowNs() - t0 > limitNs) { bad = 1; break; }
                    }
                }
                break;
            }
            if (cnt == QZSTD_HIP_NSEQ_ERROR || cnt > rq.seqCapPerItem || out + cnt >= outSeqsCapacity - 1) { wrong = 1; continue; } /* capacity rule, reference :1318-1322 */
            /* The count says how many entries there are, not that they are all there: the entries are stored by eight waves, the
             * count by a ninth, and on their ways to host memory nothing orders the one behind the others (measured: under load an
             * item's last entries arrive up to microseconds after its count).  Every entry is ONE 16-byte store that carries the
             * request's epoch in its fourth word (qzstd_hip_block_t.mark): an entry is taken when it shows it. */
            {   /* one aligned 16-byte load per entry (it arrived as one store), the mark checked here and masked out of the copy (the pinned
                 * area keeps it: marks of an earlier lap of the 24-bit epoch are wiped when the epoch starts over, above) */
                const __m128i keep = _mm_set_epi32(0, -1, -1, -1);
                __m128i acc = _mm_setzero_si128();
                size_t j;
                for (j = 0; j < cnt; j++) {
                    __m128i v = _mm_load_si128((const __m128i *)(const void *)(q + j));
                    if ((unsigned int)_mm_cvtsi128_si32(_mm_srli_si128(v, 12)) != rq.epoch) { /* not there yet: rare */
                        unsigned spins = 0;
                        do {
                            __builtin_ia32_pause();
                            __asm__ volatile("" ::: "memory"); /* (a fresh load every time round) */
                            if ((++spins & 1023u) == 0u && qzNowNs() - t0 > limitNs) { bad = 1; break; }
                            v = _mm_load_si128((const __m128i *)(const void *)(q + j));
                        } while ((unsigned int)_mm_cvtsi128_si32(_mm_srli_si128(v, 12)) != rq.epoch);
                        if (bad) break;
                    }
                    if (j + 1 < cnt) _mm_storeu_si128((__m128i *)(void *)(outSeqs + out + j), _mm_and_si128(v, keep));
                    acc = _mm_add_epi32(acc, v); /* literal and match lengths add up in lanes 1 and 2 (a block is 128 KiB at most) */
                }
                if (bad) break;
                covered += (size_t)(unsigned int)_mm_cvtsi128_si32(_mm_srli_si128(acc, 4)) + (unsigned int)_mm_cvtsi128_si32(_mm_srli_si128(acc, 8));
                if (cnt > 1) outSeqs[out].litLength += (unsigned int)carry;
            }
            if (cnt > 1) {
                out += cnt - 1;
                carry = 0;
            }
            carry += q[cnt - 1].litLength;
        }
    }
    if (bad) { /* reference: the 2 s poll limit, :1261-1285 */
        {
            unsigned long in[8] = { 0 }, dg[8] = { 0 };
            (void)qzstd_hip_service_info(sl->device, in);
            (void)qzstd_hip_service_debug(sl->device, dg);
            QZ_LOG(1, "device %d: service request not answered after %lu ms (limit %d; service: %lu launch(es), %lu request(s), state %lu; dispatcher: %lu poll(s), "
                      "%lu request(s) taken, %lu item(s) queued; workers: %lu started, %lu item(s) picked up, %lu finished, %lu gave up on a slice; ring: %lu "
                      "consumed, %lu reserved, quit %lu, frozen %lu; this request: %zu item(s), level %#x)\n",
                   sl->device, (qzNowNs() - t0) / 1000000ul, gProc.timeoutMs, in[0], in[1], in[4], dg[0], dg[1], dg[2], dg[3], dg[4], dg[5], in[6], dg[6],
                   dg[7] & 0x1FFFFFFFFFFFFFFFul, dg[7] >> 62, (dg[7] >> 61) & 1ul, nItems, (unsigned)level);
        }
        sl->vStuck = 1;
        qzstd_hip_service_mark_broken(sl->device);
        qzReleaseSlot(i);
        /* the resident kernels are asked to leave and are not used again; THIS block is not lost: it goes through the batches
         * (whose own wait has the same limit: a device that is really wedged then returns the error, reference :1261-1285) */
        s->redoneAlone++;
        return QZ_NOT_SERVED;
    }
    if (rejected) { qzReleaseSlot(i); return QZ_NOT_SERVED; }
    if (wrong) {
        qzReleaseSlot(i);
        qzCause = QZ_CAUSE_CAPACITY;
        return ZSTD_SEQUENCE_PRODUCER_ERROR;
    }
    outSeqs[out].offset = 0;
    outSeqs[out].litLength = (unsigned int)carry;
    outSeqs[out].matchLength = 0;
    outSeqs[out].rep = 0;
    out++;
    qzReleaseSlot(i);
    {   /* what arrived has to add up to the block (a wrong result must never reach libzstd, which does not validate sequences by
         * default): the lengths were added up on the way */
        const size_t sum = covered;
        if (sum != srcSize) {
            QZ_LOG(1, "service result does not add up: %zu of %zu bytes\n", sum, srcSize);
            for (k = 0; k < nItems; k++) { /* which item: every item's own list adds up to its range */
                const ZSTD_Sequence *q = sl->vSeqs + k * rq.seqCapPerItem;
                const size_t from = k * itemBytes, upTo = (k + 1) * itemBytes < srcSize ? (k + 1) * itemBytes : srcSize;
                size_t isum = 0, j;
                for (j = 0; j < sl->vCount[k] && j < rq.seqCapPerItem; j++) isum += (size_t)q[j].litLength + q[j].matchLength;
                if (isum != upTo - from) QZ_LOG(1, "  item %zu of %zu: %u sequences cover %zu of %zu bytes\n", k, nItems, sl->vCount[k], isum, upTo - from);
            }
            qzstd_hip_service_mark_broken(sl->device);
            qzCause = QZ_CAUSE_RUNTIME;
            return ZSTD_SEQUENCE_PRODUCER_ERROR;
        }
    }
    s->servedService++;
    __atomic_fetch_add(&gProc.devBlocks[sl->device][2], 1ul, __ATOMIC_RELAXED);
    return out;
}

/* one block on a slot of its own (QZSTD_HIP_COALESCE=0, and blocks too dense for a batch) */
static size_t qzSlotBlock(QZSTD_Session_T *s, int dev, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src,
                          size_t srcSize, int level)
{
    size_t rc = ZSTD_SEQUENCE_PRODUCER_ERROR;
    int tries;
    for (tries = 0; tries < 4; tries++) {
        const int i = qzGrabSlot(s->slotHint + tries * gProc.numDevices, dev);
        QZSTD_Slot_T *sl;
        if (i < 0) {
            QZ_LOG(1, "failed to grab a slot\n");
            return ZSTD_SEQUENCE_PRODUCER_ERROR;
        }
        sl = &gProc.slots[i];
        if (sl->stream && qzStillStuck(sl->device, sl->stream, &sl->stuck)) { /* quarantined: try the next one */
            qzReleaseSlot(i);
            continue;
        }
        if (dev < 0) s->slotHint = i;
        if (qzSetupSlot(sl, 1) == QZSTD_OK) rc = qzRunBlock(sl, outSeqs, outSeqsCapacity, src, srcSize, level);
        QZ_LOG(2, "block %zu B level %d -> %zu sequences (slot %d, device %d)\n", srcSize, level & 0xFF, rc, i, sl->device);
        qzReleaseSlot(i);
        return rc;
    }
    return rc;
}

size_t qatSequenceProducer(void *sequenceProducerState, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity,
                           const void *src, size_t srcSize, const void *dict, size_t dictSize,
                           int compressionLevel, size_t windowSize)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)sequenceProducerState;
    size_t rc = ZSTD_SEQUENCE_PRODUCER_ERROR;

    /* guards, reference :1123-1137 */
    if (windowSize < (srcSize < 32 * 1024 ? srcSize : 32 * 1024) || dictSize > 0 || dict) {
        QZ_LOG(2, "window %zu too small for block %zu, or dictionary given (%zu)\n", windowSize, srcSize, dictSize);
        return qzFailed(s, QZ_CAUSE_GUARD);
    }
    if (compressionLevel < QZ_LEVEL_MIN || compressionLevel > QZ_LEVEL_MAX) {
        QZ_LOG(1, "only levels 1-12 can be offloaded, got %d\n", compressionLevel);
        return qzFailed(s, QZ_CAUSE_GUARD);
    }
    if (!s || !outSeqs || !src || srcSize > QZSTD_HIP_BLOCK_MAX || outSeqsCapacity < 3) return qzFailed(s, QZ_CAUSE_GUARD);
    if (!qzDeviceUsable(s)) return qzFailed(s, QZ_CAUSE_DEVICE_DOWN);
    qzCause = QZ_CAUSE_RUNTIME; /* until a failing site below says otherwise */

    /* Served from an announcement?  (src, srcSize) must start on an announced block grid and
     * cover one or more whole blocks of it: libzstd 1.5.7 cuts multi-block frames into blocks of 32..128 KiB at 32 KiB
     * steps, so a finer grid serves several sizes — independently parsed neighbours are simply concatenated, the trailing literals of
     * one block flowing into the first sequence of the next.  The bytes of the callback must still
     * equal the staged copy the sequences were computed from (the caller may have reused or changed the buffer
     * since the announcement): one memcmp per callback, a mismatch drops the announcement — unless the announcer promised to hold
     * the bytes still (QZSTD_HINT_STABLE); such an announcement serves every block ONCE, going forward: a block asked for a second time
     * ends it (round-4 ADVICE).  The newest announcement is looked at first: an older
     * one that names the same addresses never shadows it. */
    {
        int order[QZ_HINTS], n = 0, oi, k;
        for (k = 0; k < QZ_HINTS; k++) { /* live announcements, newest first (insertion sort of at most four) */
            int j = n++;
            if (s->hint[k].st == 0) { n--; continue; }
            while (j > 0 && (int)(s->hint[order[j - 1]].seq - s->hint[k].seq) < 0) { order[j] = order[j - 1]; j--; }
            order[j] = k;
        }
        for (oi = 0; oi < n; oi++) {
            QZSTD_Hint_T *h = &s->hint[order[oi]];
            const unsigned char *p = (const unsigned char *)src;
            size_t rel, b, e, covered = 0;
            int pi, ok = 1, byAddr = 0;
            k = order[oi];
            if (h->st == 0) continue; /* (dropped further up in this loop) */
            if (h->level != compressionLevel) continue;
            if (!h->noAddr && p >= h->base && p + srcSize <= h->base + h->size) { /* by address: the callback names announced memory */
                rel = (size_t)(p - h->base);
                b = rel / h->block;
                byAddr = 1;
                if (rel % h->block != 0 || b >= h->nb) continue; /* off the grid (libzstd 1.5.7 pre-splits multi-block frames at 32 KiB steps): not served from here */
                for (e = b; e < h->nb && covered < srcSize; e++) covered += h->hDesc[e].srcLen;
                if (covered != srcSize || e - b > 8) {
                    QZ_LOG(3, "announcement %d: block %zu+%zu does not fit the grid (%zu)\n", k, rel, srcSize, h->block);
                    continue;
                }
                if (h->stable && b < h->servedUpTo) {
                    /* a block of a STABLE announcement asked for a second time: the buffer is being used again (a new job over the
                     * same memory whose last callbacks never came) — nothing vouches for its bytes any more */
                    QZ_LOG(2, "announcement %d: block %zu asked for again; dropped\n", k, b);
                    qzHintDrop(h);
                    continue;
                }
                /* verified: every callback; STABLE: every 16th block served, a sampled check of the announcer's promise (QZSTD_hintBroken) */
                if ((!h->stable || (++s->stableSampler & 15u) == 0u) && memcmp(h->hSrc + rel, src, srcSize) != 0) {
                    QZ_LOG(h->stable ? 1 : 2, "announcement %d: the buffer changed after it was announced%s; dropped\n", k,
                           h->stable ? " although it was announced with QZSTD_HINT_STABLE" : "");
                    if (h->stable) s->stableBroken++;
                    qzHintDrop(h);
                    continue;
                }
            } else {
                /* by content: a streaming caller (ZSTD_compressStream2 with small feeds, the zstd CLI) announces the buffer it
                 * read into, but libzstd hands the producer blocks out of its OWN window buffer.  An announced grid block with
                 * the same size, the same first and last 8 bytes and — verified — the same bytes serves such a callback
                 * just as well (the sequences depend on nothing but the block's bytes). */
                unsigned long long key;
                if (srcSize < 16 || h->nb > QZ_CONTENT_LOOKUP_BLOCKS || !h->keys) continue;
                key = qzBlockKey((const unsigned char *)src, srcSize);
                for (b = 0; b < h->nb; b++)
                    if (h->keys[b] == key && h->hDesc[b].srcLen == srcSize && memcmp(h->hSrc + b * h->block, src, srcSize) == 0) break;
                if (b >= h->nb) continue;
                rel = b * h->block;
                e = b + 1;
            }
            /* the parts that hold these blocks: wait for them (usually long done) */
            for (pi = 0; pi < h->nParts; pi++) {
                QZSTD_Part_T *pt = &h->part[pi];
                if (pt->b1 <= b || pt->b0 >= e) continue;
                if (pt->st == 1 && pt->flags) {
                    /* this callback's blocks only; the slot goes back once the part's last block is in */
                    const size_t w0b = b > pt->b0 ? b : pt->b0, w1b = e < pt->b1 ? e : pt->b1;
                    const unsigned long w0 = qzNowNs();
                    if (qzBlocksWait(h, w0b, w1b) != 0) {
                        qzPartFinish(h, pt); /* (times out again at once: the same block is still pending) -> failed, slot quarantined */
                    } else {
                        if (pt->seen < pt->b0) pt->seen = pt->b0;
                        while (pt->seen < pt->b1 && __atomic_load_n(&h->hCount[pt->seen], __ATOMIC_ACQUIRE) != QZ_COUNT_PENDING) pt->seen++;
                        if (pt->seen >= pt->b1) qzPartFinish(h, pt);
                    }
                    s->hintWaitNs += qzNowNs() - w0;
                    if (pt->st == 3) ok = 0;
                    continue;
                }
                if (pt->st == 1) {
                    const unsigned long w0 = qzNowNs();
                    qzPartFinish(h, pt);
                    s->hintWaitNs += qzNowNs() - w0;
                }
                if (pt->st != 2) ok = 0;
            }
            if (ok) {
                const int last = rel + srcSize >= h->size;
                size_t total = 1, carry = 0, out = 0, bi;
                int usable = 1;
                for (bi = b; bi < e; bi++) {
                    const size_t count = h->hCount[bi];
                    if (count == QZSTD_HIP_NSEQ_ERROR || count == 0 || count > h->pitch) { usable = 0; break; }
                    total += count - 1;
                }
                if (usable && total < outSeqsCapacity - 1) {
                    for (bi = b; bi < e; bi++) {
                        const ZSTD_Sequence *q;
                        const size_t count = h->hCount[bi];
                        if (h->hDesc[bi].mark & QZSTD_HIP_MARK_COMPACT) {
                            /* packed entries (8 bytes, 12-bit tag): unpacked into the caller's array as they are taken */
                            const unsigned long long *q8 = (const unsigned long long *)(const void *)((const unsigned char *)h->hSeqs + bi * h->pitch * 8u);
                            ZSTD_Sequence dl;
                            if (qzTakePacked(count > 1 ? outSeqs + out : NULL, q8, count - 1, h->hDesc[bi].mark & 0xFFFu) != 0 ||
                                qzTakePacked(&dl, q8 + count - 1, 1, h->hDesc[bi].mark & 0xFFFu) != 0) {
                                QZ_LOG(1, "announcement: entries of block %zu did not arrive within %d ms of their count\n", bi, gProc.timeoutMs);
                                usable = 0;
                                break;
                            }
                            if (count > 1) {
                                outSeqs[out].litLength += (unsigned int)carry;
                                out += count - 1;
                                carry = 0;
                            }
                            carry += dl.litLength; /* the block's delimiter: its trailing literals */
                            continue;
                        }
                        q = h->hSeqs + bi * h->pitch;
                        /* completed by its count word: the entries certify themselves one by one (the last one, the delimiter, too) */
                        if (h->hDesc[bi].mark != 0u && (qzTakeMarked(count > 1 ? outSeqs + out : NULL, q, count - 1, h->hDesc[bi].mark) != 0 ||
                                                        qzTakeMarked(NULL, q + count - 1, 1, h->hDesc[bi].mark) != 0)) {
                            QZ_LOG(1, "announcement: entries of block %zu did not arrive within %d ms of their count\n", bi, gProc.timeoutMs);
                            usable = 0;
                            break;
                        }
                        if (count > 1) {
                            if (h->hDesc[bi].mark == 0u) memcpy(outSeqs + out, q, (count - 1) * sizeof(ZSTD_Sequence));
                            outSeqs[out].litLength += (unsigned int)carry;
                            out += count - 1;
                            carry = 0;
                        }
                        carry += q[count - 1].litLength; /* the block's delimiter: its trailing literals */
                    }
                }
                if (usable && total < outSeqsCapacity - 1) {
                    outSeqs[out].offset = 0;
                    outSeqs[out].litLength = (unsigned int)carry;
                    outSeqs[out].matchLength = 0;
                    outSeqs[out].rep = 0;
                    out++;
                    s->servedFromBatch++;
                    h->touched = 1;
                    h->misses = 0;
                    if (byAddr && e > h->servedUpTo) h->servedUpTo = e; /* (a block served by content says nothing about where the announcer is) */
                    if (last && byAddr) qzHintDrop(h); /* last block consumed */
                    return out;
                }
            }
            /* announced but not served (a failed or timed-out part, a block with too many sequences): the per-block path takes THIS
             * block; the announcement lives on for its other blocks unless this was the last one.  (Round 5 ended a STABLE announcement
             * here, waiting for every launch still in flight: one dense block — more sequences than the result pitch holds — then cost
             * the rest of a 2-4 MiB claim its announcement, a cliff on short-match data (round-5 ADVICE).  Its life is bounded without
             * that: QZSTD_dropHints at the end of the announcer's job, the asked-twice rule above, the overlap rule of QZSTD_hintSourceEx.) */
            if (byAddr && rel + srcSize >= h->size) qzHintDrop(h);
            else if (byAddr && e > h->servedUpTo) h->servedUpTo = e;
            break;
        }
        /* nothing to serve from.  Announcements the caller has walked away from (used, then missed again and again)
         * are dropped, so that they do not serve stale positions */
        for (k = 0; k < QZ_HINTS; k++) {
            QZSTD_Hint_T *h = &s->hint[k];
            if (h->st != 0 && (h->touched || h->noAddr) && ++h->misses > QZ_HINT_STALE_MISSES) {
                QZ_LOG(2, "announcement %d: abandoned by the caller; dropped\n", k);
                qzHintDrop(h);
            }
        }
    }

    /* sticky device per state: a GPU of the calling thread's socket if there is one, states spread round-robin (qzPickHint) */
    if (s->slotHint < 0) s->slotHint = qzPickHint();
    rc = qzServiceBlock(s, s->slotHint % gProc.numDevices, outSeqs, outSeqsCapacity, src, srcSize, compressionLevel | gProc.levelFlags);
    if (rc != QZ_NOT_SERVED) {
        QZ_LOG(2, "block %zu B level %d -> %zu sequences (service, device %d)\n", srcSize, compressionLevel, rc, s->slotHint % gProc.numDevices);
    } else if (gProc.coalesce) {
        rc = qzCoalescedBlock(s, s->slotHint % gProc.numDevices, outSeqs, outSeqsCapacity, src, srcSize,
                              compressionLevel | gProc.levelFlags);
        QZ_LOG(2, "block %zu B level %d -> %zu sequences (coalesced, device %d)\n", srcSize, compressionLevel, rc,
               s->slotHint % gProc.numDevices);
    } else {
        rc = qzSlotBlock(s, -1, outSeqs, outSeqsCapacity, src, srcSize, compressionLevel | gProc.levelFlags);
    }
    if (rc != ZSTD_SEQUENCE_PRODUCER_ERROR) s->servedSync++;
    else (void)qzFailed(s, qzCause);
    return rc;
}

/* ---------------------------------------------------------------- announcements -- */

/* The staging copy of an announcement (reference: the memcpy into DMA memory, :1222-1227): caller's bytes -> pinned buffer the GPU reads over the
 * bus.  For a STABLE announcement nobody on the host reads the copy again soon (a sampled memcmp at most), so it goes out with STREAMING stores: no
 * read-for-ownership of the destination lines and no 2 MiB of a thread's cache spent on bytes only the GPU wants — with 16 threads staging at
 * once the copies share the socket's memory bandwidth, a third of which the ordinary stores' line fills took.  QZSTD_HIP_STAGE_NT=0: memcpy. */
static void qzStageCopy(unsigned char *dst, const unsigned char *src, size_t n)
{
    size_t i = 0;
    if (!gProc.stageNt || n < 4096 || ((uintptr_t)dst & 15u)) { memcpy(dst, src, n); return; }
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(const void *)(src + i)), b = _mm_loadu_si128((const __m128i *)(const void *)(src + i + 16)),
                      c = _mm_loadu_si128((const __m128i *)(const void *)(src + i + 32)), d = _mm_loadu_si128((const __m128i *)(const void *)(src + i + 48));
        _mm_stream_si128((__m128i *)(void *)(dst + i), a);
        _mm_stream_si128((__m128i *)(void *)(dst + i + 16), b);
        _mm_stream_si128((__m128i *)(void *)(dst + i + 32), c);
        _mm_stream_si128((__m128i *)(void *)(dst + i + 48), d);
    }
    _mm_sfence(); /* streaming stores are weakly ordered: before anything that tells the GPU the bytes are there */
    if (i < n) memcpy(dst + i, src + i, n - i);
}

/* grow-only buffers: returns the (possibly new) pointer, NULL on failure */
static void *qzGrowHostC(void *old, size_t *cap, size_t need, int dev, int coherent)
{
    void *p;
    if (old && *cap >= need) return old;
    if (old) memset(old, 0, *cap); /* staged caller data: scrubbed before the pages go back */
    qzstd_hip_host_free(old);
    p = qzHostAlloc(need, dev, coherent); /* next to the state's own GPU (the first of the GPUs an announcement is split across) */
    /* a result area completed by count words: its entries certify themselves by the announcement's epoch, and fresh pinned pages may
     * hold anything — entries of another announcement's earlier life included, whose epochs count from 1 like everybody's.  Mark 0 is
     * never valid: the area starts out wiped */
    if (p && coherent) memset(p, 0, need);
    *cap = p ? need : 0;
    return p;
}

static void *qzGrowHost(void *old, size_t *cap, size_t need, int dev) { return qzGrowHostC(old, cap, need, dev, 0); }

static void *qzGrowDev(int dev, void *old, size_t *cap, size_t need)
{
    void *p;
    if (old && *cap >= need) return old;
    qzstd_hip_free(dev, old);
    p = qzstd_hip_malloc(dev, need);
    *cap = p ? need : 0;
    return p;
}

int QZSTD_deviceStats(int device, unsigned long stats[4])
{
    int k;
    if (stats) for (k = 0; k < 4; k++) stats[k] = 0;
    if (stats && device >= 0 && device < gProc.numDevices && device < QZ_MAX_DEVICES)
    {
        for (k = 0; k < 3; k++) stats[k] = __atomic_load_n(&gProc.devBlocks[device][k], __ATOMIC_RELAXED);
        stats[3] = (unsigned long)(gProc.devNode[device] + 1); /* host NUMA node of the GPU + 1 (0 = unknown) */
    }
    return gProc.numDevices;
}

void QZSTD_failStats(void *sequenceProducerState, unsigned long stats[8])
{
    const QZSTD_Session_T *s = (const QZSTD_Session_T *)sequenceProducerState;
    int k;
    if (!stats) return;
    for (k = 0; k < 8; k++) stats[k] = 0;
    if (!s) return;
    stats[0] = s->fail[0];
    stats[1] = s->fail[QZ_CAUSE_GUARD];
    stats[2] = s->fail[QZ_CAUSE_DEVICE_DOWN];
    stats[3] = s->fail[QZ_CAUSE_TIMEOUT];
    stats[4] = s->fail[QZ_CAUSE_CAPACITY];
    stats[5] = s->fail[QZ_CAUSE_RUNTIME];
    stats[6] = s->redoneAlone;
    stats[7] = s->servedService;
}

unsigned long QZSTD_hintBroken(void *sequenceProducerState)
{
    return sequenceProducerState ? ((const QZSTD_Session_T *)sequenceProducerState)->stableBroken : 0ul;
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

/* queue blocks [b0, b1) of announcement h on a slot of device `dev` (any device when dev < 0); 0 on success */
static int qzLaunchPart(QZSTD_Session_T *s, QZSTD_Hint_T *h, QZSTD_Part_T *pt, size_t b0, size_t b1, int dev, int level,
                        int mayWait)
{
    QZSTD_Slot_T *sl;
    const size_t o0 = b0 * h->block;
    const size_t bytes = ((b1 * h->block < h->size ? b1 * h->block : h->size) - o0 + 63) & ~(size_t)63;
    size_t b;
    int i, tries, direct = 0;
    pt->st = 0;
    for (tries = 0; ; tries++) {
        /* a state's announcements start their sweep four rows apart from the next state's (a state has up to four in flight): each state
         * keeps to its own few slots, whose streams then exist after its first announcements (a stream is created at a slot's first use:
         * with sweeps that all start in the same corner the slots in use kept drifting and streams were still being created ten passes in) */
        const int nd = gProc.numDevices, rows = gProc.numSlots / (nd > 0 ? nd : 1);
        const int row = rows > 0 ? ((s->slotHint / nd) * QZ_HINTS) % rows : 0;
        i = qzTryGrabSlot(row * nd + s->slotHint % nd + tries * nd, dev);
        if (i < 0 && mayWait) {
            /* every slot is busy: give back what this state still holds, then wait for one */
            int k, j;
            for (k = 0; k < QZ_HINTS; k++)
                for (j = 0; j < s->hint[k].nParts; j++)
                    if (&s->hint[k] != h) qzPartFinish(&s->hint[k], &s->hint[k].part[j]);
            i = qzGrabSlot(s->slotHint, dev);
        }
        if (i < 0) return -1;
        sl = &gProc.slots[i];
        if (!(sl->stream && qzStillStuck(sl->device, sl->stream, &sl->stuck))) break;
        qzReleaseSlot(i); /* quarantined after a time-out: try another one */
        if (tries >= 8) return -1;
    }
    if (dev < 0) s->slotHint = i;
    if (qzSetupSlot(sl, 0) != QZSTD_OK) goto fail;
    {
        const size_t work = qzstd_hip_workspace_bytes(level, (unsigned int)(b1 - b0), (unsigned int)h->block);
        if (work) sl->dBatchWork = qzGrowDev(sl->device, sl->dBatchWork, &sl->dBatchWorkCap, work);
        if (work && !sl->dBatchWork) goto fail;
        /* (A/B switch, see hintDirect) */
        direct = h->dvSrc != NULL && (gProc.hintDirect == 1 || (gProc.hintDirect == 2 && work == 0));
    }
    if (!direct) {
        sl->dBatchSrc = (unsigned char *)qzGrowDev(sl->device, sl->dBatchSrc, &sl->dBatchSrcCap, bytes);
        if (!sl->dBatchSrc) goto fail;
    }
    for (b = b0; b < b1; b++) h->hDesc[b].srcOff = b * h->block - o0; /* relative to this part's first byte */
    /* everything below is queued on the slot's stream */
    if (!direct) {
        const unsigned long t0 = qzNowNs();
        /* the staging copy goes to device memory by a copy kernel on the stream (qzstd_hip_copy_in says why not by the runtime's copy) */
        const int rc = h->dvSrc && gProc.hintDirect != 3 ? qzstd_hip_copy_in(sl->device, sl->stream, sl->dBatchSrc, (const unsigned char *)h->dvSrc + o0, bytes)
                                                         : qzstd_hip_memcpy_h2d(sl->device, sl->stream, sl->dBatchSrc, h->hSrc + o0, bytes);
        s->hintCopyCallNs += qzNowNs() - t0;
        if (rc) { if (qzWait(sl->device, sl->stream) == 1) sl->stuck = 1; goto fail; }
    }
    {
        const unsigned char *dsrc = direct ? (const unsigned char *)h->dvSrc + o0 : sl->dBatchSrc;
        const unsigned long t0 = qzNowNs();
        const int rc = qzstd_hip_find_sequences(sl->device, sl->stream, level, dsrc, (const qzstd_hip_block_t *)h->dvDesc + b0,
                                                (unsigned int)(b1 - b0), (unsigned int)h->block, h->dvSeqs, (unsigned int *)h->dvCount + b0,
                                                sl->dBatchWork, sl->dBatchWorkCap);
        s->hintLaunchCallNs += qzNowNs() - t0;
        if (rc) {
            if (qzWait(sl->device, sl->stream) == 1) sl->stuck = 1;
            goto fail;
        }
    }
    __atomic_fetch_add(&gProc.devBlocks[sl->device][0], (unsigned long)(b1 - b0), __ATOMIC_RELAXED);
    pt->st = 1; /* in flight; the slot stays ours until qzPartFinish() */
    pt->slot = i;
    pt->b0 = b0;
    pt->b1 = b1;
    pt->seen = b0;
    pt->flags = gProc.hintFlags;
    return 0;
fail:
    for (b = b0; b < b1; b++) h->hCount[b] = QZSTD_HIP_NSEQ_ERROR; /* never launched: nothing will publish these */
    QZ_LOG(1, "announcement not queued: %s\n", qzstd_hip_last_error());
    qzReleaseSlot(i);
    return -1;
}

/* Stage a buffer, queue its match-finding and remember it in *h (asynchronous, see QZSTD_hintSource).  The blocks
 * are split into contiguous ranges, one per GPU (reference analogue: instances interleaved across devices,
 * src/qatseqprod.c:601-630), each on its own slot and stream; the results land in the announcement's pinned buffers.
 * Returns the bytes announced, 0 if none. */
static size_t qzAnnounce(QZSTD_Session_T *s, QZSTD_Hint_T *h, const void *src, size_t srcSize, size_t blockSize,
                         int compressionLevel, int stable)
{
    size_t nb, blocksBytes, srcBytes;
    unsigned long tq, tp;
    int parts, k, firstDev;

    tp = qzNowNs();
    qzHintDrop(h); /* an old announcement that was never consumed */
    s->hintDropNs += qzNowNs() - tp;
    if (qzOrphans) qzReapOrphans(0);
    nb = (srcSize + blockSize - 1) / blockSize;
    /* a fine grid means many blocks: the result area is sized by what a block of that size can produce at most */
    h->pitch = qzstd_hip_sequence_bound(blockSize) < QZ_HINT_PITCH ? qzstd_hip_sequence_bound(blockSize) : QZ_HINT_PITCH;
    h->pitch = (h->pitch + 1u) & ~(size_t)1; /* (even: a block's region of packed entries is pitch / 2 sixteen-byte units long) */
    blocksBytes = nb * sizeof(qzstd_hip_block_t);
    srcBytes = (srcSize + 63) & ~(size_t)63;

    if (s->slotHint < 0) s->slotHint = qzPickHint(); /* the state's own GPU: sticky from its first use */
    firstDev = s->slotHint % gProc.numDevices;
    tp = qzNowNs();
    h->hSrc = (unsigned char *)qzGrowHost(h->hSrc, &h->hSrcCap, srcBytes, firstDev);
    h->hDesc = (qzstd_hip_block_t *)qzGrowHost(h->hDesc, &h->hDescCap, blocksBytes, firstDev);
    h->hCount = (unsigned int *)qzGrowHostC(h->hCount, &h->hCountCap, nb * sizeof(unsigned int), firstDev, gProc.hintFlags);
    h->hSeqs = (ZSTD_Sequence *)qzGrowHostC(h->hSeqs, &h->hSeqsCap, nb * h->pitch * (gProc.hintCompact ? 8u : sizeof(ZSTD_Sequence)), firstDev, gProc.hintFlags); /* (packed entries: 8 bytes) */
    if (!h->hSrc || !h->hDesc || !h->hCount || !h->hSeqs) return 0;
    if (h->dvOf[0] != h->hDesc || !h->dvDesc) { h->dvDesc = qzstd_hip_host_device_ptr(h->hDesc); h->dvOf[0] = h->hDesc; }
    if (h->dvOf[1] != h->hCount || !h->dvCount) { h->dvCount = qzstd_hip_host_device_ptr(h->hCount); h->dvOf[1] = h->hCount; }
    if (h->dvOf[2] != h->hSeqs || !h->dvSeqs) { h->dvSeqs = qzstd_hip_host_device_ptr(h->hSeqs); h->dvOf[2] = h->hSeqs; }
    if (h->dvOf[3] != h->hSrc || !h->dvSrc) { h->dvSrc = qzstd_hip_host_device_ptr(h->hSrc); h->dvOf[3] = h->hSrc; }
    if (!h->dvDesc || !h->dvCount || !h->dvSeqs) return 0;
    s->hintPrepNs += qzNowNs() - tp;

    tq = qzNowNs();
    /* pinned staging (reference: the staging memcpy, :1222-1227).  Streaming stores where the host will not read the copy back: a verified
     * announcement compares every callback's block with it (measured with the copy streamed out: -2 ... -4 %), a STABLE one only samples (+1.5 %) */
    if (stable) qzStageCopy(h->hSrc, (const unsigned char *)src, srcSize);
    else memcpy(h->hSrc, src, srcSize);
    s->hintStageNs += qzNowNs() - tq;
    tq = qzNowNs();
    h->base = (const unsigned char *)src;
    h->size = srcSize;
    h->block = blockSize;
    h->level = compressionLevel;
    h->nb = nb;
    h->epoch = (h->epoch + 1u) & (gProc.hintCompact ? 0xFFFu : 0xFFFFFFu);
    if (h->packedLast != gProc.hintCompact) { h->epoch = 0u; h->packedLast = gProc.hintCompact; } /* (entries of the other form mean nothing as marks) */
    if (h->epoch == 0u) {
        /* the epoch (24 bits; 12 as the tag of packed entries) starts over: an entry that no announcement of the last lap overwrote would show a
         * mark that is valid again.  Nothing of this announcement's buffers is in flight here (qzHintDrop above): wipe the result area once per
         * lap (packed: 2 MiB every 4 095 announcements) */
        memset(h->hSeqs, 0, h->hSeqsCap);
        h->epoch = 1u;
    }
    if (nb <= QZ_CONTENT_LOOKUP_BLOCKS) {
        if (h->keysCap < nb) {
            free(h->keys);
            h->keys = (unsigned long long *)malloc(nb * sizeof(*h->keys));
            h->keysCap = h->keys ? nb : 0;
        }
    }
    {
        /* descriptors of every block (also of ranges that cannot be queued: the callbacks' grid arithmetic reads them).
         * Results go straight into the pinned buffer, QZ_HINT_PITCH entries per block: a block with more sequences
         * reports an error and is redone by the per-block path when its callback comes */
        size_t b;
        for (b = 0; b < nb; b++) {
            const size_t o = b * blockSize;
            h->hDesc[b].srcOff = o;
            h->hDesc[b].seqOff = gProc.hintCompact ? b * (h->pitch / 2u) : b * h->pitch; /* (sixteen-byte units) */
            h->hDesc[b].srcLen = (unsigned int)(srcSize - o < blockSize ? srcSize - o : blockSize);
            h->hDesc[b].seqCap = (unsigned int)h->pitch;
            h->hDesc[b].parseFrom = 0;
            h->hDesc[b].mark = gProc.hintFlags ? (gProc.hintCompact ? (h->epoch | QZSTD_HIP_MARK_COMPACT) : h->epoch) : 0u; /* (count-word completion: every entry certifies itself, see qzTakeMarked) */
            h->hCount[b] = gProc.hintFlags ? QZ_COUNT_PENDING : QZSTD_HIP_NSEQ_ERROR; /* until a kernel says otherwise */
            if (h->keys && nb <= h->keysCap && nb <= QZ_CONTENT_LOOKUP_BLOCKS)
                h->keys[b] = h->hDesc[b].srcLen >= 16 ? qzBlockKey(h->hSrc + o, h->hDesc[b].srcLen) : 0ull;
        }
    }
    /* contiguous block ranges, one per GPU, starting at this state's own GPU; a range is worth a launch from 4 blocks */
    parts = gProc.split;
    if ((size_t)parts > nb / 4) parts = nb / 4 ? (int)(nb / 4) : 1;
    h->nParts = 0;
    for (k = 0; k < parts; k++) {
        const size_t b0 = nb * (size_t)k / (size_t)parts, b1 = nb * (size_t)(k + 1) / (size_t)parts;
        const int dev = parts > 1 ? (firstDev + k) % gProc.numDevices : firstDev;
        if (qzLaunchPart(s, h, &h->part[h->nParts], b0, b1, dev, compressionLevel | gProc.levelFlags, 1) != 0) {
            /* that range could not be queued: its callbacks take the per-block path */
            h->part[h->nParts].st = 3;
            h->part[h->nParts].b0 = b0;
            h->part[h->nParts].b1 = b1;
        }
        h->nParts++;
    }
    h->st = 1;
    h->touched = 0;
    h->misses = 0;
    h->stable = 0;
    h->servedUpTo = 0;
    h->seq = ++s->hintSeq;
    s->hintQueueNs += qzNowNs() - tq;
    return srcSize;
}

/* Lifetime of an announcement (round-4 ADVICE): it ends when the callback of its last block has come, when a callback finds its bytes
 * changed (verified announcements) or asks for a block a second time (STABLE ones), when its LAST block could not be served,
 * when a new announcement names addresses it covers (STABLE: at once; verified: it stops serving by address and lives on for
 * by-content look-ups), after 16 callbacks in a row that it could not serve, at QZSTD_dropHints(), and with its state.  At most
 * four are alive per state. */
int QZSTD_hintSourceEx(void *sequenceProducerState, const void *src, size_t srcSize, size_t blockSize,
                       int compressionLevel, unsigned int flags)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)sequenceProducerState;
    QZSTD_Hint_T *h = NULL;
    int k, inflight = 0;

    /* the grid: 1 KiB (libzstd's smallest ZSTD_c_maxBlockSize) .. 128 KiB, a multiple of 16 */
    if (!s || !src || srcSize == 0 || srcSize > QZ_HINT_MAX_BYTES || blockSize < 1024 || blockSize > QZSTD_HIP_BLOCK_MAX ||
        (blockSize & 15))
        return -1;
    if (compressionLevel < QZ_LEVEL_MIN || compressionLevel > QZ_LEVEL_MAX || (flags & ~(unsigned int)QZSTD_HINT_STABLE)) return -1;
    if (!qzDeviceUsable(s)) return -1;
    /* an older announcement over (some of) the same addresses is void from here on: whatever it says about them is not what the
     * caller is announcing now */
    for (k = 0; k < QZ_HINTS; k++) {
        QZSTD_Hint_T *o = &s->hint[k];
        if (o->st == 0 || !((const unsigned char *)src < o->base + o->size && o->base < (const unsigned char *)src + srcSize)) continue;
        /* A STABLE one goes (nothing but its announcer's word vouches for its bytes).  A verified one only stops serving BY ADDRESS: a
         * streaming caller that refills and announces the same input buffer again (ZSTD_compressStream2, the zstd CLI) is served by
         * content out of libzstd's window — every such block is compared with the staged copy — and still needs the tail blocks of the
         * older announcement; dropping it also meant a synchronous wait for its launches on every announcement (round-5 ADVICE).  It
         * ends as every announcement does: consumed, stale after 16 misses, pushed out by the fourth newer one, QZSTD_dropHints. */
        if (o->stable || o->nb > QZ_CONTENT_LOOKUP_BLOCKS || !o->keys) qzHintDrop(o);
        else o->noAddr = 1;
    }
    /* an empty place of the ring first (starting where the last announcement left off); only when all four are alive the oldest goes */
    for (k = 0; k < QZ_HINTS && !h; k++)
        if (s->hint[(s->hintNext + k) % QZ_HINTS].st == 0) h = &s->hint[(s->hintNext + k) % QZ_HINTS];
    if (!h) {
        h = &s->hint[0];
        for (k = 1; k < QZ_HINTS; k++)
            if ((int)(s->hint[k].seq - h->seq) < 0) h = &s->hint[k];
    }
    s->hintNext = (int)((h - s->hint) + 1) % QZ_HINTS;
    if (qzAnnounce(s, h, src, srcSize, blockSize, compressionLevel, (flags & QZSTD_HINT_STABLE) != 0) == 0) return -1;
    for (k = 0; k < h->nParts; k++) inflight += h->part[k].st == 1;
    if (!inflight) { /* nothing could be queued */
        qzHintDrop(h);
        return -1;
    }
    h->stable = (flags & QZSTD_HINT_STABLE) != 0;
    h->noAddr = 0;
    s->hintCalls++;
    if (s->hintCalls == 8) /* the event log's timers: steady state only (the first announcements allocate pinned buffers and create streams) */
        s->hintStageNs = s->hintQueueNs = s->hintWaitNs = s->hintCopyCallNs = s->hintLaunchCallNs = s->hintPrepNs = s->hintDropNs = 0;
    return 0;
}

int QZSTD_hintSource(void *sequenceProducerState, const void *src, size_t srcSize, size_t blockSize,
                     int compressionLevel)
{
    return QZSTD_hintSourceEx(sequenceProducerState, src, srcSize, blockSize, compressionLevel, 0u);
}

/* the announcer is done with what it announced (end of a job; before it frees or rewrites a buffer it promised to hold still):
 * every announcement of the state ends here — launches in flight are waited for, nothing is served from them afterwards */
void QZSTD_dropHints(void *sequenceProducerState)
{
    QZSTD_Session_T *s = (QZSTD_Session_T *)sequenceProducerState;
    int k;
    if (!s) return;
    for (k = 0; k < QZ_HINTS; k++)
        if (s->hint[k].st != 0) qzHintDrop(&s->hint[k]);
}
