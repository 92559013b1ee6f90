/*
 * fuzz_roundtrip.c — TEST INFRASTRUCTURE.  Randomised round-trip driver for the plugin's C surface, in the spirit of the
 * upstream zstd fuzz targets the reference runs through its adapter (/root/reference/test/fuzzing/README.md:9-28:
 * simple_round_trip, stream_round_trip, block_round_trip ...), which need a zstd source tree that this image does not
 * have.  Every iteration draws a buffer (random kind, size, repeats, runs), a level, a block-size limit, one-shot or
 * streaming compression, optional announcements over random sub-ranges, optional rewriting of the buffer between the
 * announcement and the compression, compresses through libzstd with qatSequenceProducer registered
 * (ZSTD_c_validateSequences = 1, no fallback unless the iteration also injects producer errors), decompresses and
 * compares.  It is linked twice by tests/test_fuzz.py:
 *   - CPU: host/qatseqprod.c + tests/mock/mock_hip.c + the oracle, all built with -fsanitize=address,undefined;
 *   - GPU box (-m gpu): against lib/libqatseqprod.so (the real kernels).
 * Also calls the five FUZZ_* adapter symbols of test/fuzzing/qatseqprodfuzzer.c the way zstd's fuzzers do.
 *
 * PARITY, not only a property (round-3 verdict, f3): every k-th iteration the same input goes through libzstd a second time with
 * the ORACLE's producer registered (qzo_sequence_producer, oracle/qzstd_oracle.c — linked into this test binary, never into the
 * product) under the same parameters and the same feed pattern, and the two frames must be byte-identical: whatever path served
 * the blocks (service, batches, announcements, rewritten announcements, the adapter), the sequences were the oracle's.
 * usage: fuzz_roundtrip <seed> <iterations> [max buffer KiB, default 3072] [compare with the oracle every k-th iteration, default 1; 0 = never]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qatseqprod.h"
#include "qzstd_oracle.h"

/* the adapter under test (qat-zstd-plugin_amd/test/fuzzing/qatseqprodfuzzer.c) */
size_t FUZZ_seqProdSetup(void);
size_t FUZZ_seqProdTearDown(void);
void *FUZZ_createSeqProdState(void);
size_t FUZZ_freeSeqProdState(void *state);
size_t FUZZ_thirdPartySeqProd(void *state, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity, const void *src, size_t srcSize,
                              const void *dict, size_t dictSize, int compressionLevel, size_t windowSize);

typedef ZSTD_outBuffer OutB;
typedef ZSTD_inBuffer InB;

static uint64_t gRng;
static uint32_t rnd(void)
{
    gRng ^= gRng << 13; gRng ^= gRng >> 7; gRng ^= gRng << 17;
    return (uint32_t)(gRng >> 16);
}
static uint32_t below(uint32_t n) { return n ? rnd() % n : 0; }

static void fill(unsigned char *b, size_t n)
{
    const uint32_t kind = below(6);
    size_t i;
    if (kind == 0) { for (i = 0; i < n; i++) b[i] = (unsigned char)rnd(); }                      /* incompressible */
    else if (kind == 1) { memset(b, (int)below(256), n); }                                       /* one run */
    else if (kind == 2) { const uint32_t a = 2 + below(6); for (i = 0; i < n; i++) b[i] = (unsigned char)('a' + below(a)); }
    else if (kind == 3) {                                                                        /* words from a small vocabulary */
        unsigned char voc[64][12];
        uint32_t w, l;
        for (w = 0; w < 64; w++) for (l = 0; l < 12; l++) voc[w][l] = (unsigned char)('a' + below(26));
        for (i = 0; i < n;) { const unsigned char *v = voc[below(64)]; const uint32_t len = 3 + below(9); for (l = 0; l < len && i < n; l++) b[i++] = v[l]; if (i < n) b[i++] = ' '; }
    } else if (kind == 4) {                                                                      /* records with zero padding */
        for (i = 0; i < n; i++) b[i] = (i % 61) < 9 ? (unsigned char)('0' + (i / 61) % 10) : 0;
    } else { for (i = 0; i < n; i++) b[i] = (unsigned char)(i * 7 + (i >> 8)); }
    /* plant copies: near, far, long */
    for (i = 0; i < 6 && n > 64; i++) {
        const size_t len = 4 + below((uint32_t)(n / 4 > 20000 ? 20000 : n / 4));
        const size_t s = below((uint32_t)(n - len)), d = below((uint32_t)(n - len));
        memmove(b + d, b + s, len);
    }
}

/* On the iterations that are compared with the oracle the plugin's producer is wrapped: every callback's result is compared with
 * qzo_sequence_producer's on the spot, so a difference names the BLOCK (index, size, level, window), not only the frame */
typedef struct {
    void *state;                     /* the plugin's (or the adapter's) state */
    ZSTD_sequenceProducer_F inner;
    void *ostate;                    /* the oracle's: NULL or a profile */
    qzo_seq_t *tmp;
    size_t tmpCap;
    unsigned calls, bad, joined;
    /* the announcement of this iteration, if any: a callback that covers SEVERAL whole blocks of an announced grid is served by joining
     * the independently parsed grid blocks (host/qatseqprod.c, "the announcer's choice": include/qatseqprod.h says what the grid should
     * be) — then the expected list is the join of the oracle's lists of those grid blocks */
    const unsigned char *hBase;
    size_t hLen, hGrid;
} Wrap;

/* the oracle's lists of the grid blocks [p, p + n) covers, joined as the plugin joins them; (size_t)-1 when the callback does not cover whole grid blocks */
static size_t joined_expectation(Wrap *w, const unsigned char *p, size_t n, size_t cap, int level, size_t window)
{
    size_t rel, pos = 0, out = 0, carry = 0, k = 0;
    if (!w->hBase || p < w->hBase || p + n > w->hBase + w->hLen) return (size_t)-1;
    rel = (size_t)(p - w->hBase);
    if (rel % w->hGrid != 0) return (size_t)-1;
    while (pos < n) {
        const size_t left = w->hLen - (rel + pos), len = left < w->hGrid ? left : w->hGrid;
        qzo_seq_t *q = w->tmp + out;
        size_t cnt;
        if (pos + len > n || ++k > 8 || out + 4 >= cap) return (size_t)-1;
        cnt = qzo_sequence_producer(w->ostate, q, cap - out, p + pos, len, NULL, 0, level, window < len ? len : window);
        if (cnt == (size_t)-1 || cnt == 0) return (size_t)-1;
        if (cnt > 1) { q[0].litLength += (uint32_t)carry; out += cnt - 1; carry = 0; }
        carry += q[cnt - 1].litLength;
        pos += len;
    }
    w->tmp[out].offset = 0; w->tmp[out].litLength = (uint32_t)carry; w->tmp[out].matchLength = 0; w->tmp[out].rep = 0;
    return k > 1 ? out + 1 : (size_t)-1;
}

static int same_lists(const ZSTD_Sequence *a, const qzo_seq_t *b, size_t n, size_t *where)
{
    size_t i;
    for (i = 0; i < n; i++)
        if (a[i].offset != b[i].offset || a[i].litLength != b[i].litLength || a[i].matchLength != b[i].matchLength) { *where = i; return 0; }
    return 1;
}

static size_t wrap_producer(void *st, ZSTD_Sequence *out, size_t cap, const void *src, size_t n, const void *dict, size_t dictSize, int level, size_t window)
{
    Wrap *w = (Wrap *)st;
    const size_t got = w->inner(w->state, out, cap, src, n, dict, dictSize, level, window);
    size_t want, i;
    if (cap > w->tmpCap) { free(w->tmp); w->tmp = (qzo_seq_t *)malloc(cap * sizeof(qzo_seq_t)); w->tmpCap = cap; }
    want = qzo_sequence_producer(w->ostate, w->tmp, cap, src, n, dict, dictSize, level, window);
    if (got == want && (got == (size_t)-1 || same_lists(out, w->tmp, got, &i))) { w->calls++; return got; }
    {   /* not the whole block's list: then it has to be the join of an announced finer grid's lists */
        const size_t wantJ = got == (size_t)-1 ? (size_t)-1 : joined_expectation(w, (const unsigned char *)src, n, cap, level, window);
        if (wantJ != (size_t)-1 && wantJ == got && same_lists(out, w->tmp, got, &i)) { w->joined++; w->calls++; return got; }
        want = qzo_sequence_producer(w->ostate, w->tmp, cap, src, n, dict, dictSize, level, window);
        if (!w->bad++) {
            if (got != want) fprintf(stderr, "  callback %u (block of %zu bytes, level %d, window %zu, capacity %zu): %zu sequences, oracle %zu (joined grid blocks: %zu)\n", w->calls, n, level, window, cap, got, want, wantJ);
            else if (!same_lists(out, w->tmp, got, &i))
                fprintf(stderr, "  callback %u (block of %zu bytes, level %d, window %zu): sequence %zu of %zu is {%u,%u,%u}, oracle {%u,%u,%u}\n", w->calls, n, level, window, i, got,
                        out[i].offset, out[i].litLength, out[i].matchLength, w->tmp[i].offset, w->tmp[i].litLength, w->tmp[i].matchLength);
        }
    }
    w->calls++;
    return got;
}

/* what an iteration drew: everything the compression depends on besides the bytes, so that it can be repeated with another producer */
typedef struct {
    int level, stream, maxBlock, extRep, splitter;
    uint64_t feedSeed; /* streaming: the feed sizes come from a generator of their own */
} Case;

/* one compression of src[0, n) as the case says, `producer` registered with `state`; returns the frame size or (size_t)-1 */
static size_t compress_case(ZSTD_CCtx *zc, void *state, ZSTD_sequenceProducer_F producer, const Case *cs, const unsigned char *src, size_t n,
                            unsigned char *dst, size_t dstCap, unsigned it)
{
    size_t r;
    ZSTD_registerSequenceProducer(zc, state, producer);
    if (ZSTD_isError(ZSTD_CCtx_setParameter(zc, ZSTD_c_compressionLevel, cs->level)) ||
        ZSTD_isError(ZSTD_CCtx_setParameter(zc, ZSTD_c_validateSequences, 1)) ||
        ZSTD_isError(ZSTD_CCtx_setParameter(zc, ZSTD_c_enableSeqProducerFallback, 0))) return (size_t)-1;
    if (cs->maxBlock) (void)ZSTD_CCtx_setParameter(zc, ZSTD_c_maxBlockSize, cs->maxBlock);
    if (cs->extRep >= 0) (void)ZSTD_CCtx_setParameter(zc, ZSTD_c_searchForExternalRepcodes, cs->extRep);
    if (cs->splitter >= 0) (void)ZSTD_CCtx_setParameter(zc, ZSTD_c_blockSplitterLevel, cs->splitter);
    if (cs->stream) {
        OutB o = { dst, dstCap, 0 };
        InB in = { src, 0, 0 };
        uint64_t g = cs->feedSeed;
        while (in.pos < n) {
            size_t feed;
            g ^= g << 13; g ^= g >> 7; g ^= g << 17;
            feed = 1 + (size_t)((g >> 16) % 400000u);
            in.size = in.pos + feed < n ? in.pos + feed : n;
            r = ZSTD_compressStream2(zc, &o, &in, ZSTD_e_continue);
            if (ZSTD_isError(r)) { fprintf(stderr, "it %u: compressStream2: %s\n", it, ZSTD_getErrorName(r)); return (size_t)-1; }
        }
        do { r = ZSTD_compressStream2(zc, &o, &in, ZSTD_e_end); } while (r != 0 && !ZSTD_isError(r));
        if (ZSTD_isError(r)) { fprintf(stderr, "it %u: compressStream2(end): %s\n", it, ZSTD_getErrorName(r)); return (size_t)-1; }
        return o.pos;
    }
    r = ZSTD_compress2(zc, dst, dstCap, src, n);
    if (ZSTD_isError(r)) { fprintf(stderr, "it %u (n %zu level %d): compress2: %s\n", it, n, cs->level, ZSTD_getErrorName(r)); return (size_t)-1; }
    return r;
}

int main(int argc, char **argv)
{
    const uint64_t seed = argc > 1 ? strtoull(argv[1], NULL, 0) : 1;
    const unsigned iters = argc > 2 ? (unsigned)atoi(argv[2]) : 100;
    const size_t maxN = (size_t)(argc > 3 ? atoi(argv[3]) : 3072) << 10;
    unsigned char *src = (unsigned char *)malloc(maxN), *back = (unsigned char *)malloc(maxN);
    const size_t dstCap = ZSTD_compressBound(maxN);
    const unsigned every = argc > 4 ? (unsigned)atoi(argv[4]) : 1u;
    const char *extE = getenv("QZSTD_HIP_EXT_REPCODES");
    const int extEnv = extE && atoi(extE) == 1;
    unsigned char *dst = (unsigned char *)malloc(dstCap), *dst2 = (unsigned char *)malloc(dstCap);
    unsigned it, viaAdapter = 0, streamed = 0, hinted = 0, compared = 0, joinedNow = 0, joinedCalls = 0;
    const unsigned char *hBase = NULL;
    size_t hLen = 0, hGrid = 0;
    gRng = seed * 0x9E3779B97F4A7C15ull + 0x1234567;
    if (!src || !back || !dst || !dst2) return 2;
    if (FUZZ_seqProdSetup() != 0) { fprintf(stderr, "FUZZ_seqProdSetup failed (no device?)\n"); return 3; }
    for (it = 0; it < iters; it++) {
        const uint32_t cls = below(10);
        const size_t n = cls < 3 ? below(700) : (cls < 7 ? below(300000) : below((uint32_t)maxN));
        Case cs;
        joinedNow = 0;
        const int useAdapter = below(3) == 0;
        void *state = useAdapter ? FUZZ_createSeqProdState() : QZSTD_createSeqProdState();
        ZSTD_CCtx *zc = ZSTD_createCCtx();
        size_t csize = 0, r;
        if (!state || !zc) return 2;
        cs.level = 1 + (int)below(12);
        cs.stream = below(3) == 0;
        cs.maxBlock = below(3) == 0 ? 1024 << below(8) : 0;
        cs.extRep = below(4) == 0 ? (int)below(3) : -1;
        cs.splitter = below(4) == 0 ? (int)below(3) : -1;
        cs.feedSeed = ((uint64_t)rnd() << 20) | 1u;
        fill(src, n);
        hBase = NULL; hLen = hGrid = 0;
        if (!useAdapter && n >= 4096 && below(2) == 0) { /* announce a random sub-range on a random grid, sometimes rewrite it afterwards */
            const size_t grid = (size_t)16 << (4 + below(10)); /* 256 B .. 128 KiB */
            const size_t off = below(2) ? 0 : (below((uint32_t)(n / 2)) & ~(size_t)15);
            size_t len = n - off;
            if (len > ((size_t)16 << 20)) len = (size_t)16 << 20;
            if (grid <= 131072 && QZSTD_hintSource(state, src + off, len, grid, cs.level) == 0) { hinted++; hBase = src + off; hLen = len; hGrid = grid; }
            if (below(4) == 0) fill(src + off, len < 5000 ? len : 5000); /* the caller breaks the immutability contract */
        }
        if (cs.stream) streamed++;
        {
            const int cmp = every && it % every == 0;
            Wrap w = { state, useAdapter ? FUZZ_thirdPartySeqProd : qatSequenceProducer, NULL, NULL, 0, 0, 0, 0, hBase, hLen, hGrid };
            qzo_profile_t wprof;
            if (cmp && extEnv) { if (qzo_profile_for_level(cs.level | QZO_LEVEL_REPCODES, 0, &wprof) != 0) return 2; w.ostate = &wprof; }
            csize = cmp ? compress_case(zc, &w, wrap_producer, &cs, src, n, dst, dstCap, it) : compress_case(zc, state, w.inner, &cs, src, n, dst, dstCap, it);
            free(w.tmp);
            if (csize == (size_t)-1) return 1;
            joinedNow = w.joined;
            joinedCalls += w.joined;
            if (w.bad) {
                unsigned long fs[8] = { 0 }, hs[4] = { 0 };
                if (!useAdapter) { QZSTD_failStats(state, fs); QZSTD_hintStats(state, hs); }
                fprintf(stderr, "it %u (seed %llu, n %zu, level %d, stream %d, maxBlock %d, adapter %d): %u of %u CALLBACKS DIFFER FROM THE ORACLE "
                                "(this state: %lu blocks from announcements, %lu per block of which %lu by the service, %lu redone)\n",
                        it, (unsigned long long)seed, n, cs.level, cs.stream, cs.maxBlock, useAdapter, w.bad, w.calls, hs[0], hs[1], fs[7], fs[6]);
                return 1;
            }
        }
        r = ZSTD_decompress(back, maxN, dst, csize);
        if (ZSTD_isError(r) || r != n || memcmp(back, src, n) != 0) {
            fprintf(stderr, "it %u (seed %llu, n %zu, level %d, stream %d): ROUND TRIP MISMATCH\n", it, (unsigned long long)seed, n, cs.level, cs.stream);
            return 1;
        }
        if (every && it % every == 0 && joinedNow == 0) { /* the same case through libzstd + the oracle's producer: the frames must be the same bytes
                                                          * (callbacks served as joined blocks of a finer announced grid were checked list by list above) */
            ZSTD_CCtx *zo = ZSTD_createCCtx();
            qzo_profile_t prof;
            size_t osize;
            void *ostate = NULL; /* NULL: the profile of the callback's level */
            if (!zo) return 2;
            if (extEnv) { /* QZSTD_HIP_EXT_REPCODES=1: the plugin serves every level in its repeat-aware form */
                if (qzo_profile_for_level(cs.level | QZO_LEVEL_REPCODES, 0, &prof) != 0) return 2;
                ostate = &prof;
            }
            osize = compress_case(zo, ostate, (ZSTD_sequenceProducer_F)qzo_sequence_producer, &cs, src, n, dst2, dstCap, it);
            if (osize == (size_t)-1) return 1;
            if (osize != csize || memcmp(dst, dst2, csize) != 0) {
                fprintf(stderr, "it %u (seed %llu, n %zu, level %d, stream %d, maxBlock %d, adapter %d): FRAME DIFFERS FROM THE ORACLE'S (%zu vs %zu bytes)\n",
                        it, (unsigned long long)seed, n, cs.level, cs.stream, cs.maxBlock, useAdapter, csize, osize);
                return 1;
            }
            ZSTD_freeCCtx(zo);
            compared++;
        }
        ZSTD_freeCCtx(zc);
        if (useAdapter) { (void)FUZZ_freeSeqProdState(state); viaAdapter++; }
        else QZSTD_freeSeqProdState(state);
    }
    (void)FUZZ_seqProdTearDown(); /* does not stop the device (reference adapter :46-49) */
    QZSTD_stopQatDevice();
    printf("fuzz ok: seed %llu, %u iterations (%u through the FUZZ_* adapter, %u streamed, %u with announcements, %u frames identical to the oracle's, "
           "%u callbacks served as joined blocks of a finer announced grid = the oracle's lists of those blocks joined)\n",
           (unsigned long long)seed, iters, viaAdapter, streamed, hinted, compared, joinedCalls);
    free(src); free(back); free(dst); free(dst2);
    return 0;
}
