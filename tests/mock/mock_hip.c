/*
 * mock_hip.c — TEST INFRASTRUCTURE ONLY.  A CPU stand-in for the device layer of include/qzstd_hip.h so that the
 * host logic of qat-zstd-plugin_amd/host/qatseqprod.c (guards, slots, cross-thread coalescer, announced and guessed
 * look-ahead, result checks) can be exercised by `pytest -m "not gpu"` in a container without a GPU.  "Device" memory
 * is plain malloc, streams are synchronous, and the launch entry point runs the oracle (oracle/qzstd_oracle.c) per
 * block.  It is linked ONLY into tests/mock/libqatseqprod_mock.so (built by tests/test_host_mock.py); the product
 * library never contains it and keeps failing loudly without a GPU.
 */
#include "qzstd_hip.h"
#include "qzstd_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include <time.h>

static _Thread_local char gErr[128] = "";
static int gLaunches;
static int gLaunchDev[64];
static volatile long long gStallUntilNs; /* test hook: every stream looks busy until then (a wedged device) */

static long long nowNs(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
/* test hooks */
void qzstd_mock_stall_ms(int ms) { gStallUntilNs = ms > 0 ? nowNs() + (long long)ms * 1000000ll : 0; }
int qzstd_mock_launches_on(int device) { return device >= 0 && device < 64 ? gLaunchDev[device] : -1; }

const char *qzstd_hip_last_error(void) { return gErr; }
int qzstd_mock_launches(void) { return gLaunches; }
int qzstd_hip_device_count(void)
{
    const char *v = getenv("QZSTD_MOCK_DEVICES"); /* several "GPUs": the split of announcements and the device round-robin */
    const int n = v ? atoi(v) : 1;
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}
int qzstd_hip_device_name(int device, char *buf, size_t bufLen)
{
    (void)device;
    if (buf && bufLen) snprintf(buf, bufLen, "mock device (CPU oracle)");
    return 0;
}
void *qzstd_hip_malloc(int device, size_t bytes) { (void)device; return malloc(bytes ? bytes : 1); }
void qzstd_hip_free(int device, void *p) { (void)device; free(p); }
void *qzstd_hip_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
/* NUMA, mocked: QZSTD_MOCK_NODES = k puts mock device d on node d % k (default: every device on node 0); allocations "on a node"
 * are plain malloc, counted per node (test hook qzstd_mock_node_allocs) */
static unsigned long gNodeAllocs[16], gUnplacedAllocs;
int qzstd_hip_device_numa_node(int device)
{
    const char *v = getenv("QZSTD_MOCK_NODES");
    const int k = v ? atoi(v) : 1;
    if (k < 0) return -1; /* "unknown" */
    return k > 1 ? device % k : 0;
}
void *qzstd_hip_host_alloc_on_node(size_t bytes, int node, int coherent)
{
    (void)coherent;
    if (node >= 0 && node < 16) __atomic_fetch_add(&gNodeAllocs[node], 1ul, __ATOMIC_RELAXED);
    else __atomic_fetch_add(&gUnplacedAllocs, 1ul, __ATOMIC_RELAXED);
    return malloc(bytes ? bytes : 1);
}
int qzstd_hip_host_node_of(const void *h) { (void)h; return -1; }
int qzstd_hip_occupancy(int device, int level) { (void)device; (void)level; return 1; }
unsigned long qzstd_mock_node_allocs(int node) { return node >= 0 && node < 16 ? gNodeAllocs[node] : gUnplacedAllocs; }
void qzstd_mock_node_allocs_reset(void) { memset(gNodeAllocs, 0, sizeof gNodeAllocs); gUnplacedAllocs = 0; }
void *qzstd_hip_host_device_ptr(void *h) { return h; }
void qzstd_hip_host_free(void *h) { free(h); }
void *qzstd_hip_stream_create(int device) { (void)device; return malloc(1); }
void qzstd_hip_stream_destroy(int device, void *s) { (void)device; free(s); }
int qzstd_hip_stream_sync(int device, void *s) { (void)device; (void)s; return 0; }
int qzstd_hip_stream_query(int device, void *s) { (void)device; (void)s; return nowNs() < gStallUntilNs ? 1 : 0; }
int qzstd_hip_stream_wait(int device, void *s, unsigned timeoutMs)
{
    const long long deadline = nowNs() + (long long)timeoutMs * 1000000ll;
    (void)device; (void)s;
    while (nowNs() < gStallUntilNs) {
        const struct timespec nap = { 0, 200000 };
        if (nowNs() >= deadline) { snprintf(gErr, sizeof gErr, "mock: stream still busy after %u ms", timeoutMs); return 1; }
        nanosleep(&nap, NULL);
    }
    return 0;
}
int qzstd_hip_memcpy_h2d(int device, void *s, void *dst, const void *src, size_t n) { (void)device; (void)s; memcpy(dst, src, n); return 0; }
int qzstd_hip_copy_in(int device, void *s, void *dst, const void *src, size_t n) { (void)device; (void)s; if (n & 15) return -1; memcpy(dst, src, n); return 0; }
int qzstd_hip_memcpy_d2h(int device, void *s, void *dst, const void *src, size_t n) { (void)device; (void)s; memcpy(dst, src, n); return 0; }
int qzstd_hip_memset(int device, void *s, void *dst, int v, size_t n) { (void)device; (void)s; memset(dst, v, n); return 0; }
int qzstd_hip_memcpy2d_d2h(int device, void *s, void *dst, size_t dp, const void *src, size_t sp, size_t w, size_t h)
{
    size_t r;
    (void)device; (void)s;
    for (r = 0; r < h; r++) memcpy((char *)dst + r * dp, (const char *)src + r * sp, w);
    return 0;
}

/* ---- the resident service, mocked: a request is served synchronously by the oracle, item by item, counts written last.
 * Test hooks: QZSTD_MOCK_SERVICE=0 -> "not served" (the callers take the launch path); a stall (qzstd_mock_stall_ms) ->
 * the counts never arrive (the caller's poll times out); qzstd_mock_service_level(l) -> requests of other levels are
 * handed back as rejected, as the dispatcher does when another level is resident. */
static int gSvcRequests, gSvcBroken, gSvcStops, gSvcLevel;
void qzstd_mock_service_level(int level) { gSvcLevel = level; }
int qzstd_mock_service_requests(void) { return gSvcRequests; }
void *qzstd_hip_host_alloc_coherent(size_t bytes) { return malloc(bytes ? bytes : 1); }
static volatile int gLateMarks; /* test hook (qzstd_mock_late_marks) */
void qzstd_mock_late_marks(int on) { gLateMarks = on; }
typedef struct { uint32_t *w, *real; size_t n; uint32_t epoch; int us; } late_t;
static void *late_marks(void *p)
{
    late_t *lt = (late_t *)p;
    size_t j;
    struct timespec nap = { 0, 0 };
    nap.tv_nsec = (long)lt->us * 1000l;
    nanosleep(&nap, NULL);
    if (lt->epoch & QZSTD_HIP_MARK_COMPACT) { /* packed entries: one 8-byte store each, last entries first */
        uint64_t *w8 = (uint64_t *)lt->w;
        for (j = lt->n; j-- > 0; )
            __atomic_store_n(&w8[j], QZSTD_HIP_PACK(lt->real[j * 4u], lt->real[j * 4u + 1u], lt->real[j * 4u + 2u], lt->epoch & 0xFFFu), __ATOMIC_RELEASE);
        free(lt->real);
        free(lt);
        return NULL;
    }
    for (j = lt->n; j-- > 0; ) { /* last entries first; an entry's mark with (here: after) its three other words */
        lt->w[j * 4u] = lt->real[j * 4u]; lt->w[j * 4u + 1u] = lt->real[j * 4u + 1u]; lt->w[j * 4u + 2u] = lt->real[j * 4u + 2u];
        __atomic_store_n(&lt->w[j * 4u + 3u], lt->epoch, __ATOMIC_RELEASE);
    }
    free(lt->real);
    free(lt);
    return NULL;
}
/* QZSTD_MOCK_PROGRESSIVE=1: the mock serves requests on a thread of its own and its "workers" wait for a count word to leave
 * QZSTD_HIP_NSEQ_STAGING before they read its slice — the protocol of the resident kernels' progressive staging (qzstd_hip.h), so that the
 * host's queue-first-stage-behind path runs in the CPU suite */
static int progressiveOn(void)
{
    const char *v = getenv("QZSTD_MOCK_PROGRESSIVE");
    return v && atoi(v) > 0;
}
int qzstd_hip_service_progressive(int device) { (void)device; return progressiveOn(); }
static int serve_request(int level, const qzstd_hip_svc_req_t *r);
typedef struct { int level; qzstd_hip_svc_req_t r; } async_req_t;
static void *serve_async(void *p)
{
    async_req_t *a = (async_req_t *)p;
    (void)serve_request(a->level, &a->r);
    free(a);
    return NULL;
}

int qzstd_hip_service_submit(int device, int level, const qzstd_hip_svc_req_t *r)
{
    const char *v = getenv("QZSTD_MOCK_SERVICE");
    qzo_profile_t pf;
    (void)device;
    if ((v && atoi(v) == 0) || gSvcBroken) return 1;
    if (qzo_profile_for_level(level, r->srcLen, &pf)) return 1;
    if (pf.chainDepth && !r->dWork) { snprintf(gErr, sizeof gErr, "mock: the chain levels need dWork"); return -1; }
    if (r->nItems < 1 || r->nItems > QZSTD_HIP_SVC_MAX_ITEMS || r->slot >= QZSTD_HIP_SVC_MAX_SLOTS || (r->itemBytes & ((1u << pf.segLog) - 1u)) ||
        (size_t)(r->nItems - 1) * r->itemBytes >= r->srcLen) {
        snprintf(gErr, sizeof gErr, "mock: bad service request");
        return -1;
    }
    __sync_fetch_and_add(&gSvcRequests, 1);
    if (nowNs() < gStallUntilNs) return 0; /* queued, never served */
    if (progressiveOn()) {
        async_req_t *a = (async_req_t *)malloc(sizeof *a);
        pthread_t th;
        if (!a) return -1;
        a->level = level;
        a->r = *r;
        if (pthread_create(&th, NULL, serve_async, a) != 0) { free(a); return -1; }
        pthread_detach(th);
        return 0;
    }
    return serve_request(level, r);
}

static int serve_request(int level, const qzstd_hip_svc_req_t *r)
{
    qzo_profile_t pf;
    uint32_t k;
    if (qzo_profile_for_level(level, r->srcLen, &pf)) return 1;
    for (k = 0; k < r->nItems; k++) {
        const uint32_t from = k * r->itemBytes, upTo = from + r->itemBytes < r->srcLen ? from + r->itemBytes : r->srcLen;
        size_t n;
        if (gSvcLevel && gSvcLevel != level) { __atomic_store_n(&r->hCount[k], QZSTD_HIP_NSEQ_REJECTED, __ATOMIC_RELEASE); continue; }
        while (__atomic_load_n(&r->hCount[k], __ATOMIC_ACQUIRE) == QZSTD_HIP_NSEQ_STAGING) { /* the caller is still copying slice k in */
            const struct timespec nap = { 0, 1000 };
            nanosleep(&nap, NULL);
        }
        memcpy((uint8_t *)r->dSrc + from, (const uint8_t *)r->hSrc + from, upTo - from); /* the item's slice */
        n = qzo_find_sequences_from(&pf, (const uint8_t *)r->dSrc, upTo, from, (qzo_seq_t *)r->hSeqs + (size_t)k * r->seqCapPerItem, r->seqCapPerItem);
        if (n != QZO_ERROR) { /* every entry carries the request's epoch in its fourth word, as the real workers' do */
            size_t j;
            if (gLateMarks) { /* test hook: the counts first, the entries' marks a while later (the order host memory may see them in) */
                late_t *lt = (late_t *)malloc(sizeof *lt);
                pthread_t th;
                lt->w = (uint32_t *)r->hSeqs + (size_t)k * r->seqCapPerItem * 4u; lt->n = n; lt->epoch = r->epoch; lt->us = 300 + 50 * (int)k;
                lt->real = (uint32_t *)malloc(n * 16u);
                memcpy(lt->real, lt->w, n * 16u);
                memset(lt->w, 0xEE, n * 16u); /* what is there before the entries arrive: anything */
                if (pthread_create(&th, NULL, late_marks, lt) == 0) pthread_detach(th);
                else { late_marks(lt); }
            } else {
                for (j = 0; j < n; j++) ((uint32_t *)r->hSeqs)[((size_t)k * r->seqCapPerItem + j) * 4u + 3u] = r->epoch;
            }
        }
        __atomic_store_n(&r->hCount[k], n == QZO_ERROR ? QZSTD_HIP_NSEQ_ERROR : (uint32_t)n, __ATOMIC_RELEASE);
    }
    return 0;
}
int qzstd_hip_service_poke(int device, int level) { (void)device; (void)level; return 0; }
int qzstd_hip_service_stop(int device) { (void)device; __sync_fetch_and_add(&gSvcStops, 1); return 0; }
void qzstd_hip_service_mark_broken(int device) { (void)device; gSvcBroken = 1; }
void qzstd_mock_service_repair(void) { gSvcBroken = 0; }
int qzstd_hip_service_info(int device, unsigned long out[8])
{
    int k;
    (void)device;
    for (k = 0; k < 8; k++) out[k] = 0;
    out[1] = (unsigned long)gSvcRequests;
    out[3] = (unsigned long)gSvcBroken;
    return 0;
}

int qzstd_hip_service_debug(int device, unsigned long out[8]) { int k; (void)device; for (k = 0; k < 8; k++) out[k] = 0; return 0; }

/* ---- QZSTD_MOCK_REPLAY=1: a "device" that costs (almost) nothing — tests/stress/hostpath_bench.c measures the HOST side of the
 * announcement path with it.  A block's sequences are computed by the oracle once, remembered under a key of (level, length,
 * parseFrom, 16 sampled words of the block) and copied out from then on.  The key does not cover every byte: for benchmarks over a
 * buffer that does not change, never for a correctness test. */
typedef struct { uint64_t key; uint32_t n; qzo_seq_t *seqs; } replay_t;
#define REPLAY_SLOTS 16384u
static replay_t gReplay[REPLAY_SLOTS];
static pthread_mutex_t gReplayMu = PTHREAD_MUTEX_INITIALIZER;
static unsigned long long gReplayNs, gReplayHits;
unsigned long long qzstd_mock_replay_ns(void) { return __atomic_load_n(&gReplayNs, __ATOMIC_RELAXED); }
unsigned long long qzstd_mock_replay_hits(void) { return __atomic_load_n(&gReplayHits, __ATOMIC_RELAXED); }
static int replayOn(void)
{
    static int on = -1;
    if (on < 0) { const char *v = getenv("QZSTD_MOCK_REPLAY"); on = v && atoi(v) > 0; }
    return on;
}
static uint64_t replayKey(int level, const uint8_t *p, uint32_t n, uint32_t parseFrom)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)(unsigned)level << 48) ^ ((uint64_t)n << 20) ^ parseFrom;
    uint32_t i;
    for (i = 0; i < 16u && n >= 8u; i++) {
        uint64_t w;
        memcpy(&w, p + (size_t)((uint64_t)(n - 8u) * i / 15u), 8);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 29;
    }
    return h | 1ull; /* never 0: 0 = empty slot */
}
static const replay_t *replayFind(uint64_t key)
{
    uint32_t i = (uint32_t)(key >> 17) & (REPLAY_SLOTS - 1u), tries;
    for (tries = 0; tries < 64u; tries++, i = (i + 1u) & (REPLAY_SLOTS - 1u)) {
        const uint64_t k = __atomic_load_n(&gReplay[i].key, __ATOMIC_ACQUIRE);
        if (k == key) return &gReplay[i];
        if (k == 0) return NULL;
    }
    return NULL;
}
static void replayKeep(uint64_t key, const qzo_seq_t *seqs, size_t n)
{
    uint32_t i = (uint32_t)(key >> 17) & (REPLAY_SLOTS - 1u), tries;
    pthread_mutex_lock(&gReplayMu);
    for (tries = 0; tries < 64u; tries++, i = (i + 1u) & (REPLAY_SLOTS - 1u)) {
        if (gReplay[i].key == key) break;
        if (gReplay[i].key == 0) {
            gReplay[i].seqs = (qzo_seq_t *)malloc((n ? n : 1) * sizeof(qzo_seq_t));
            if (gReplay[i].seqs) {
                memcpy(gReplay[i].seqs, seqs, n * sizeof(qzo_seq_t));
                gReplay[i].n = (uint32_t)n;
                __atomic_store_n(&gReplay[i].key, key, __ATOMIC_RELEASE);
            }
            break;
        }
    }
    pthread_mutex_unlock(&gReplayMu);
}

int qzstd_hip_find_sequences(int device, void *stream, int level, const void *d_src, const qzstd_hip_block_t *d_blocks,
                             uint32_t nBlocks, uint32_t maxBlockLen, void *d_seqs, uint32_t *d_nseq, void *d_work,
                             size_t workBytes)
{
    uint32_t b;
    qzo_profile_t pf;
    (void)stream;
    if (qzo_profile_for_level(level, maxBlockLen, &pf)) { snprintf(gErr, sizeof gErr, "mock: bad level"); return -1; }
    if (qzstd_hip_workspace_bytes(level, nBlocks, maxBlockLen) > workBytes || (workBytes && !d_work)) {
        snprintf(gErr, sizeof gErr, "mock: workspace missing or too small");
        return -1;
    }
    __sync_fetch_and_add(&gLaunches, 1);
    if (device >= 0 && device < 64) __sync_fetch_and_add(&gLaunchDev[device], 1);
    for (b = 0; b < nBlocks; b++) {
        const qzstd_hip_block_t *k = &d_blocks[b];
        size_t n;
        if (k->mark & QZSTD_HIP_MARK_COMPACT) {
            /* packed entries (qzstd_hip.h): the oracle's sequences, 8 bytes each with the 12-bit tag, in the item's region (seqOff in 16-byte units) */
            qzo_seq_t *tmp = (qzo_seq_t *)malloc(((size_t)k->seqCap + 1u) * sizeof(qzo_seq_t));
            uint64_t *w8 = (uint64_t *)d_seqs + (size_t)k->seqOff * 2u;
            size_t j;
            n = QZO_ERROR;
            if (tmp && replayOn()) { /* (QZSTD_MOCK_REPLAY: a "device" that costs one look-up and the packing) */
                const long long t0 = nowNs();
                const uint64_t key = replayKey(level, (const uint8_t *)d_src + k->srcOff, k->srcLen, k->parseFrom);
                const replay_t *r = replayFind(key);
                if (r && r->n <= k->seqCap) {
                    for (j = 0; j < r->n; j++)
                        __atomic_store_n(&w8[j], QZSTD_HIP_PACK(r->seqs[j].offset, r->seqs[j].litLength, r->seqs[j].matchLength, k->mark & 0xFFFu), __ATOMIC_RELAXED);
                    __atomic_store_n(&d_nseq[b], r->n, __ATOMIC_RELEASE);
                    __atomic_fetch_add(&gReplayNs, (unsigned long long)(nowNs() - t0), __ATOMIC_RELAXED);
                    __atomic_fetch_add(&gReplayHits, 1ull, __ATOMIC_RELAXED);
                    free(tmp);
                    continue;
                }
                n = qzo_find_sequences_from(&pf, (const uint8_t *)d_src + k->srcOff, k->srcLen, k->parseFrom, tmp, k->seqCap);
                if (n != QZO_ERROR) replayKeep(key, tmp, n);
            } else if (tmp) {
                n = qzo_find_sequences_from(&pf, (const uint8_t *)d_src + k->srcOff, k->srcLen, k->parseFrom, tmp, k->seqCap);
            }
            if (nowNs() < gStallUntilNs) { free(tmp); continue; }
            if (n != QZO_ERROR) {
                if (gLateMarks) { /* test hook: the count first, the entries a while later, last entries first */
                    late_t *lt = (late_t *)malloc(sizeof *lt);
                    pthread_t th;
                    lt->w = (uint32_t *)w8; lt->n = n; lt->epoch = k->mark; lt->us = 300 + 20 * (int)(b & 15u);
                    lt->real = (uint32_t *)malloc(n * 16u);
                    memcpy(lt->real, tmp, n * 16u);
                    memset(w8, 0, n * 8u);
                    if (pthread_create(&th, NULL, late_marks, lt) == 0) pthread_detach(th);
                    else late_marks(lt);
                } else {
                    for (j = 0; j < n; j++)
                        __atomic_store_n(&w8[j], QZSTD_HIP_PACK(tmp[j].offset, tmp[j].litLength, tmp[j].matchLength, k->mark & 0xFFFu), __ATOMIC_RELAXED);
                }
            }
            free(tmp);
            __atomic_store_n(&d_nseq[b], n == QZO_ERROR ? QZSTD_HIP_NSEQ_ERROR : (uint32_t)n, __ATOMIC_RELEASE);
            continue;
        }
        if (replayOn()) {
            const long long t0 = nowNs();
            const uint64_t key = replayKey(level, (const uint8_t *)d_src + k->srcOff, k->srcLen, k->parseFrom);
            const replay_t *r = replayFind(key);
            if (r && r->n <= k->seqCap) {
                uint32_t *w = (uint32_t *)d_seqs + (size_t)k->seqOff * 4u;
                size_t j;
                memcpy(w, r->seqs, (size_t)r->n * sizeof(qzo_seq_t));
                for (j = 0; j < r->n; j++) w[j * 4u + 3u] = k->mark;
                __atomic_store_n(&d_nseq[b], r->n, __ATOMIC_RELEASE);
                __atomic_fetch_add(&gReplayNs, (unsigned long long)(nowNs() - t0), __ATOMIC_RELAXED);
                __atomic_fetch_add(&gReplayHits, 1ull, __ATOMIC_RELAXED);
                continue;
            }
            n = qzo_find_sequences_from(&pf, (const uint8_t *)d_src + k->srcOff, k->srcLen, k->parseFrom, (qzo_seq_t *)d_seqs + k->seqOff, k->seqCap);
            if (n != QZO_ERROR) replayKeep(key, (const qzo_seq_t *)d_seqs + k->seqOff, n);
        } else {
            n = qzo_find_sequences_from(&pf, (const uint8_t *)d_src + k->srcOff, k->srcLen, k->parseFrom,
                                        (qzo_seq_t *)d_seqs + k->seqOff, k->seqCap);
        }
        /* a stalled "GPU" (qzstd_mock_stall_ms) never publishes: the count words keep what the host put there (announcements poll them) */
        if (nowNs() < gStallUntilNs) continue;
        if (n != QZO_ERROR) { /* every entry carries the block's mark in its fourth word, as the kernel's do (qzstd_hip_block_t.mark) */
            uint32_t *w = (uint32_t *)d_seqs + (size_t)k->seqOff * 4u;
            size_t j;
            if (gLateMarks && k->mark) { /* test hook: the count first, the entries a while later, last entries first */
                late_t *lt = (late_t *)malloc(sizeof *lt);
                pthread_t th;
                lt->w = w; lt->n = n; lt->epoch = k->mark; lt->us = 300 + 20 * (int)(b & 15u);
                lt->real = (uint32_t *)malloc(n * 16u);
                memcpy(lt->real, w, n * 16u);
                memset(w, 0xEE, n * 16u);
                if (pthread_create(&th, NULL, late_marks, lt) == 0) pthread_detach(th);
                else late_marks(lt);
            } else {
                for (j = 0; j < n; j++) w[j * 4u + 3u] = k->mark;
            }
        }
        __atomic_store_n(&d_nseq[b], n == QZO_ERROR ? QZSTD_HIP_NSEQ_ERROR : (uint32_t)n, __ATOMIC_RELEASE);
    }
    return 0;
}
