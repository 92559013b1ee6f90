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
    unsigned it, viaAdapter = 0, streamed = 0, hinted = 0, compared = 0;
    gRng = seed * 0x9E3779B97F4A7C15ull + 0x1234567;
    if (!src || !back || !dst || !dst2) return 2;
    if (FUZZ_seqProdSetup() != 0) { fprintf(stderr, "FUZZ_seqProdSetup failed (no device?)\n"); return 3; }
    for (it = 0; it < iters; it++) {
        const uint32_t cls = below(10);
        const size_t n = cls < 3 ? below(700) : (cls < 7 ? below(300000) : below((uint32_t)maxN));
        Case cs;
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
        if (!useAdapter && n >= 4096 && below(2) == 0) { /* announce a random sub-range on a random grid, sometimes rewrite it afterwards */
            const size_t grid = (size_t)16 << (4 + below(10)); /* 256 B .. 128 KiB */
            const size_t off = below(2) ? 0 : (below((uint32_t)(n / 2)) & ~(size_t)15);
            size_t len = n - off;
            if (len > ((size_t)16 << 20)) len = (size_t)16 << 20;
            if (grid <= 131072 && QZSTD_hintSource(state, src + off, len, grid, cs.level) == 0) hinted++;
            if (below(4) == 0) fill(src + off, len < 5000 ? len : 5000); /* the caller breaks the immutability contract */
        }
        if (cs.stream) streamed++;
        csize = compress_case(zc, state, useAdapter ? FUZZ_thirdPartySeqProd : qatSequenceProducer, &cs, src, n, dst, dstCap, it);
        if (csize == (size_t)-1) return 1;
        r = ZSTD_decompress(back, maxN, dst, csize);
        if (ZSTD_isError(r) || r != n || memcmp(back, src, n) != 0) {
            fprintf(stderr, "it %u (seed %llu, n %zu, level %d, stream %d): ROUND TRIP MISMATCH\n", it, (unsigned long long)seed, n, cs.level, cs.stream);
            return 1;
        }
        if (every && it % every == 0) { /* the same case through libzstd + the oracle's producer: the frames must be the same bytes */
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
    printf("fuzz ok: seed %llu, %u iterations (%u through the FUZZ_* adapter, %u streamed, %u with announcements, %u frames identical to the oracle's)\n",
           (unsigned long long)seed, iters, viaAdapter, streamed, hinted, compared);
    free(src); free(back); free(dst); free(dst2);
    return 0;
}
