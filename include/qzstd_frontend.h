/*
 * qzstd_frontend.h — batch front-end for the host entropy stage.
 *
 * Not part of the reference's surface: the reference stops at the per-block producer and leaves the threading to the
 * caller ("one thread per DC instance", /root/reference/README.md:138; the caller shape is its benchmark,
 * /root/reference/test/benchmark.c:222-382, :514-516).  Everything behind ZSTD_compress2 except the match-finder stays
 * on the calling thread, so the end-to-end rate of a GPU-backed producer is set by how many host threads run libzstd's
 * entropy stage and by whether the GPU works AHEAD of them.  This front-end packages exactly that:
 *
 *   - a persistent pool of worker threads, each with its own ZSTD_CCtx and its own QZSTD_createSeqProdState();
 *   - the input is cut into chunks (each chunk its own frame — the reference benchmark's framing, :300-321); workers claim
 *     runs of chunks from a shared cursor and keep two claims (three at the chain levels; QZSTD_FRONT_AHEAD=1..3) announced beyond
 *     the one they are entropy-coding, with QZSTD_hintSourceEx(..., QZSTD_HINT_STABLE) — the source is the const argument of the
 *     call in progress — so the GPUs match-find ahead of every worker (a state holds four announcements; on a multi-GPU node an
 *     announcement is split across the GPUs); then ZSTD_compress2 per chunk.  Claims are at most segmentBytes; where the entropy
 *     stage sets the pace (levels 1-4) a worker's first claims and the job's last ones are smaller, so that the first results
 *     come back early and the workers finish together (QZSTD_FRONT_UNIFORM=0|1 overrides the choice by level);
 *   - frames land at fixed strides in the destination (frame c at dst + c * QZSTD_frontFrameStride()), sizes in
 *     frameSizes[c]; QZSTD_frontCompact() packs them back to back.
 *
 * Plain C, links against libqatseqprod and a libzstd >= 1.5.4.
 */
#ifndef QZSTD_FRONTEND_H
#define QZSTD_FRONTEND_H

#include <stddef.h>

#if defined(__cplusplus)
extern "C" {
#endif

typedef struct QZSTD_Front_s QZSTD_Front;

typedef struct {
    int nThreads;        /* worker threads (>= 1) */
    int level;           /* 1..12 */
    size_t chunkSize;    /* bytes per frame, > 0 */
    size_t segmentBytes; /* bytes announced at a time at most, rounded to whole chunks, <= 16 MiB (0 = 2 MiB at levels 1-4; 4 MiB but at most 64 chunks at levels 5-12) */
    int extRepcodes;     /* ZSTD_c_searchForExternalRepcodes: 0 auto, 1 enable, 2 disable (the reference's -E) */
    int useProducer;     /* 1 = register the GPU sequence producer (with software fallback), 0 = software zstd (baseline) */
} QZSTD_FrontParams;

/* NULL on bad parameters or when a worker cannot be set up.  Starts the device layer (QZSTD_startQatDevice). */
QZSTD_Front *QZSTD_createFront(const QZSTD_FrontParams *params);

/* bytes between the starts of consecutive frames in the destination: ZSTD_compressBound(chunkSize) */
size_t QZSTD_frontFrameStride(const QZSTD_Front *f);

/* Compresses src as ceil(srcSize / chunkSize) independent frames.  dst must hold that many strides; frameSizes that many
 * entries.  Blocks until done.  Returns the number of frames, or (size_t)-1 on error.  One call at a time per front. */
size_t QZSTD_frontCompress(QZSTD_Front *f, const void *src, size_t srcSize, void *dst, size_t dstCapacity, size_t *frameSizes);

/* packs the frames back to back at the start of dst; returns the total compressed size */
size_t QZSTD_frontCompact(const QZSTD_Front *f, void *dst, const size_t *frameSizes, size_t nFrames);

/* blocks served from an announcement / per block, summed over the workers' states, since creation */
void QZSTD_frontStats(QZSTD_Front *f, unsigned long stats[2]);

/* producer callbacks that returned the error code — those blocks were compressed by libzstd's own match-finder (the
 * front-end switches ZSTD_c_enableSeqProducerFallback on) — summed over the workers' states, by cause: the layout of
 * QZSTD_failStats() in qatseqprod.h.  A run whose stats[0] is not 0 was not served by the GPU alone. */
void QZSTD_frontFailStats(QZSTD_Front *f, unsigned long stats[8]);

void QZSTD_freeFront(QZSTD_Front *f);

#if defined(__cplusplus)
}
#endif
#endif /* QZSTD_FRONTEND_H */
