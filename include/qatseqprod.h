/*
 * qatseqprod.h — public C surface of the MI355X-native ZSTD block-level sequence
 * producer.  Drop-in for the header of intel/QAT-ZSTD-Plugin: same file name, same
 * function names, signatures, status values and version macros
 * (/root/reference/src/qatseqprod.h:50-55 version, :60-65 status enum, :72 version(),
 * :110-116 producer, :130 start, :137 stop, :145 create state, :151 free state), so
 * a program that already does
 *
 *     QZSTD_startQatDevice();
 *     void *st = QZSTD_createSeqProdState();
 *     ZSTD_registerSequenceProducer(cctx, st, qatSequenceProducer);
 *     ZSTD_compress2(...) / ZSTD_compressStream2(...)
 *
 * links against this libqatseqprod.{so,a} unchanged.  "QatDevice" in the names is
 * kept for source compatibility; the device is an AMD Instinct MI355X (gfx950) and
 * the match search is a HIP kernel (csrc/qzstd_kernels.hip), not QAT hardware.
 */
#ifndef QATSEQPROD_H
#define QATSEQPROD_H

#if defined(__cplusplus)
extern "C" {
#endif

#ifndef ZSTD_STATIC_LINKING_ONLY
#define ZSTD_STATIC_LINKING_ONLY
#endif
/* zstd >= 1.5.4 is required for ZSTD_registerSequenceProducer.  When no such zstd.h is
 * installed (the ROCm build image only ships the 1.5.7 shared object), the
 * declarations this library needs come from qzstd_zstd_abi.h instead. */
#if defined(QZSTD_USE_SYSTEM_ZSTD_H)
#include "zstd.h"
#else
#include "qzstd_zstd_abi.h"
#endif

#define QZSTD_VERSION "0.2.0"
#define QZSTD_VERSION_MAJOR 0
#define QZSTD_VERSION_MINOR 2
#define QZSTD_VERSION_RELEASE 0
#define QZSTD_VERSION_NUMBER \
    (QZSTD_VERSION_MAJOR * 100 * 100 + QZSTD_VERSION_MINOR * 100 + QZSTD_VERSION_RELEASE)

/* Status codes of QZSTD_startQatDevice (values as in the reference). */
typedef enum {
    QZSTD_OK = 0,          /* device(s) up and usable */
    QZSTD_STARTED = 1,     /* runtime initialised but no usable device/slot */
    QZSTD_FAIL = -1,       /* could not initialise */
    QZSTD_UNSUPPORTED = -2 /* declared for compatibility; never returned */
} QZSTD_Status_e;

/* Version string of the plugin API this library implements ("0.2.0"). */
const char *QZSTD_version(void);

/*
 * Block-level sequence producer (a ZSTD_sequenceProducer_F).  libzstd calls it once
 * per <=128 KiB block; it returns the number of ZSTD_Sequence entries written to
 * outSeqs (the last one is the trailing-literals delimiter) or
 * ZSTD_SEQUENCE_PRODUCER_ERROR.
 *
 * Same limits as the reference: levels 1..12; no dictionary (dict must be NULL);
 * every block is parsed without history; windowSize must cover min(srcSize, 32 KiB);
 * ZSTD_c_nbWorkers > 0 and long-distance matching are rejected by libzstd itself;
 * one CCtx + one state per thread.  With ZSTD_c_enableSeqProducerFallback = 1 any
 * error makes libzstd fall back to its own match-finder for that block.
 */
size_t qatSequenceProducer(void *sequenceProducerState, ZSTD_Sequence *outSeqs,
                           size_t outSeqsCapacity, const void *src, size_t srcSize,
                           const void *dict, size_t dictSize, int compressionLevel,
                           size_t windowSize);

/* Process-wide, idempotent, thread-safe initialisation of the GPU runtime and slots. */
int QZSTD_startQatDevice(void);

/* Releases every device resource; call after all states are freed.  Safe if never started. */
void QZSTD_stopQatDevice(void);

/* One state per CCtx/thread, reusable across compressions.  NULL on allocation failure. */
void *QZSTD_createSeqProdState(void);

/* NULL-safe. */
void QZSTD_freeSeqProdState(void *sequenceProducerState);

/* ------------------------------------------------------------------------------------
 * Additive extension (not in the reference): look-ahead hint.
 *
 * The producer API is synchronous per block, which would leave 255 of the 256 CUs idle.
 * When the caller is about to run ZSTD_compress2 over a contiguous buffer it may tell
 * the state first; the plugin then match-finds the whole buffer in one batched launch
 * and serves the following qatSequenceProducer() callbacks whose (src, srcSize) lie on
 * the announced block grid from that result.  Purely an optimisation: callbacks that do
 * not match the hint take the normal single-block path.  `blockSize` is the block grid
 * (131072 for plain ZSTD_compress2; the frame/chunk size when each chunk is its own
 * frame; 1024 .. 131072, a multiple of 16; at most 16 MiB per call).  Returns 0 when the hint was accepted, -1 otherwise.
 *
 * The call is asynchronous: it copies at most 16 MiB into pinned memory, queues the
 * transfers and the launches, and returns.  A state holds four announcements (two until round 4), so a caller
 * announces segments k+1 .. k+3 at most and compresses segment k; a fifth replaces the oldest.  On a node with several GPUs the
 * blocks of one announcement are split into contiguous ranges, one per GPU, each on its own
 * stream; every kernel writes its results into the announcement's pinned host buffers
 * (QZSTD_HIP_SPLIT=n limits the split to n GPUs, 1 keeps it on the state's own GPU).
 *
 * Streaming callers (ZSTD_compressStream2 with small feeds, the zstd CLI): libzstd hands the producer blocks out
 * of its own window buffer, so the callback never names announced memory.  Announcements of up to 256 grid blocks
 * are therefore also matched BY CONTENT: a callback whose size, first and last 8 bytes and — verified by memcmp —
 * bytes equal an announced grid block is served from it.  Announce what you read, on the 128 KiB grid of the stream.
 *
 * Contract: the announced bytes should not change until their callbacks have come.  The
 * plugin does not rely on it — every callback served from an announcement is compared with
 * the staged copy first (memcmp); if the buffer was rewritten the announcement is dropped
 * and the block is match-found afresh — so breaking the contract costs time, never
 * correctness.  An announcement the caller walks away from is dropped after 16 misses.
 * ------------------------------------------------------------------------------------ */
int QZSTD_hintSource(void *sequenceProducerState, const void *src, size_t srcSize,
                     size_t blockSize, int compressionLevel);

/* The same with flags (additive, round 4).  QZSTD_HINT_STABLE — NOT VERIFIED PER BLOCK, ONLY SAMPLED (1 block in 16): a caller that
 * breaks the promise below may get up to 15 blocks of stale sequences before the library notices.  The caller holds [src, src + srcSize) unchanged until the
 * callbacks of these blocks have come — what a compress call's const source promises anyway, here promised from the
 * announcement on — and the per-callback memcmp against the staged copy is skipped (6-8 us of every 128 KiB callback; the
 * batch front-end, include/qzstd_frontend.h, announces this way: its source is the const argument of one call).  Blocks are
 * then matched by ADDRESS only, every block is served ONCE, going forward: a block asked for a second time ends the announcement
 * (see "Lifetime" below).  Breaking the promise while the announcement is alive produces
 * frames that do not decode to the input: use it only for memory nobody else writes, and end it with QZSTD_dropHints() when the
 * job is over.  Every 16th block served from a STABLE announcement is compared with the staged copy anyway (0.5 us per block on
 * average): a mismatch ends the announcement, the block is match-found afresh and QZSTD_hintBroken() counts it — a sampled check
 * that reveals a caller who does not keep the promise, not a guarantee.  Unknown flag bits are refused (-1). */
#define QZSTD_HINT_STABLE 1u
int QZSTD_hintSourceEx(void *sequenceProducerState, const void *src, size_t srcSize,
                       size_t blockSize, int compressionLevel, unsigned int flags);

/* Lifetime of an announcement — bounded, whatever the caller does: it ends when the callback of its last block has come; when a
 * callback finds its bytes changed (verified announcements); when a callback asks a STABLE announcement for a block a second time
 * (a buffer that is being used again); when its LAST block could not be served (any other block that cannot be served — too many
 * sequences for the result area, a failed launch — just takes the per-block path); when a newer announcement names addresses it covers
 * (a STABLE one ends at once, a verified one stops serving by address and lives on for by-content look-ups: streaming callers that refill
 * and re-announce one buffer); after 16 callbacks in a row it could not serve; at QZSTD_dropHints(); with its state.  A state keeps at
 * most four; callbacks look at the newest first.
 *
 * QZSTD_dropHints(state): every announcement of the state ends now (launches in flight are waited for).  Call it when the job
 * the announcements belonged to is over — in particular before a buffer announced with QZSTD_HINT_STABLE is rewritten or freed
 * while callbacks for some of its blocks never came (libzstd does not call the producer for blocks below 7 bytes, nor after an
 * error).  The batch front-end does, at the end of every QZSTD_frontCompress. */
void QZSTD_dropHints(void *sequenceProducerState);

/* The library reads nothing but [src, src + srcSize) of a callback or an announcement: the opt-in "transparent look-ahead" of
 * rounds 1-4 (QZSTD_HIP_LOOKAHEAD: guessed reads behind a callback's block) is gone — it lost to the resident service at every
 * level but one and read memory the caller never handed over; the variable is ignored. */

/* Time-out (reference: 2 s of polling, src/qatseqprod.c:1099-1104): a request that is still running after
 * QZSTD_HIP_TIMEOUT_MS (default 2000) returns ZSTD_SEQUENCE_PRODUCER_ERROR, so that ZSTD_c_enableSeqProducerFallback
 * takes over; the stream involved is not used again before it has drained. */

/* Diagnostics for the above: stats[0] = blocks served from an announcement, [1] = blocks that took
 * the per-block path, [2] = announcements accepted, [3] = microseconds spent waiting for the GPU. */
void QZSTD_hintStats(void *sequenceProducerState, unsigned long stats[4]);
/* blocks of QZSTD_HINT_STABLE announcements whose sampled comparison with the staged copy failed (a broken promise: see above) */
unsigned long QZSTD_hintBroken(void *sequenceProducerState);

/* Callbacks of this state that returned ZSTD_SEQUENCE_PRODUCER_ERROR, by cause — with ZSTD_c_enableSeqProducerFallback = 1
 * libzstd compresses such a block with its own match-finder and says nothing, so this is the only place they show (the
 * reference counts one cause only: failOffloadCnt, src/qatseqprod.c:122,:1141):
 *   stats[0] all of them = [1] argument guards (window, dictionary, level; reference :1123-1137) + [2] device not started
 *   + [3] time-outs (QZSTD_HIP_TIMEOUT_MS) + [4] capacity rule (count >= capacity - 1, reference :1318) + [5] runtime errors
 *   (allocation, launch, no free slot);  stats[6] = blocks served at the second attempt (too dense for a batch's result area and
 *   redone alone, or redone through the batches after a request to the resident service timed out: served, not errors);  stats[7] = blocks served by the resident service (a subset of QZSTD_hintStats' [1]). */
void QZSTD_failStats(void *sequenceProducerState, unsigned long stats[8]);

/* Process-wide, per GPU (the blocks shard across the GPUs of a node with no exchange step; reference analogue: instances
 * interleaved across devices, src/qatseqprod.c:601-630): stats[0] = blocks queued on that GPU from announcements (an
 * announcement is cut into contiguous ranges, one per GPU), [1] = blocks that went through its batches, [2] = blocks served by
 * its resident service, since QZSTD_startQatDevice(); [3] = the host NUMA node the GPU is attached to + 1 (0 = unknown): pinned staging
 * buffers, result areas and request rings of a GPU are allocated on that node, and a state's GPU is picked among the GPUs of the socket
 * its thread runs on when it is first used (QZSTD_HIP_NUMA=0 turns both off; reference: qaeMemAllocNUMA, src/qatseqprod.c:216-246).  Returns the number of GPUs in use (stats may be NULL). */
int QZSTD_deviceStats(int device, unsigned long stats[4]);

#if defined(__cplusplus)
}
#endif
#endif /* QATSEQPROD_H */
