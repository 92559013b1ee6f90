/*
 * qzstd_zstd_abi.h — the slice of libzstd's public ABI (zstd >= 1.5.4) that a
 * block-level sequence producer and its callers touch.
 *
 * Why this file exists: qatseqprod.h (like the reference's src/qatseqprod.h:42-45)
 * wants `zstd.h` with ZSTD_STATIC_LINKING_ONLY.  A zstd >= 1.5.4 *header* is not
 * present on the build image (only the 1.5.7 shared object is), so when no usable
 * zstd.h is found these hand-written declarations are used instead.  Every type,
 * prototype and enum value below was checked against libzstd 1.5.7 at run time
 * (tests/test_zstd_abi.py does it again on every run: parameter bounds via
 * ZSTD_cParam_getBounds, sizeof(ZSTD_Sequence) via ZSTD_sequenceBound round trips).
 *
 * It is NOT a copy of zstd.h: it declares only what this repository calls.
 * When a real zstd.h >= 1.5.4 is on the include path, define QZSTD_USE_SYSTEM_ZSTD_H
 * (the Makefile does it automatically) and this file is skipped.
 */
#ifndef QZSTD_ZSTD_ABI_H
#define QZSTD_ZSTD_ABI_H

#include <stddef.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define ZSTD_BLOCKSIZELOG_MAX 17
#define ZSTD_BLOCKSIZE_MAX (1 << ZSTD_BLOCKSIZELOG_MAX)

typedef struct ZSTD_CCtx_s ZSTD_CCtx;
typedef struct ZSTD_DCtx_s ZSTD_DCtx;

/* 16 bytes; `rep` is ignored by libzstd on the sequence-producer path */
typedef struct {
    unsigned int offset;
    unsigned int litLength;
    unsigned int matchLength;
    unsigned int rep;
} ZSTD_Sequence;

typedef struct { size_t error; int lowerBound; int upperBound; } ZSTD_bounds;

typedef enum { ZSTD_ps_auto = 0, ZSTD_ps_enable = 1, ZSTD_ps_disable = 2 } ZSTD_paramSwitch_e;
typedef enum { ZSTD_e_continue = 0, ZSTD_e_flush = 1, ZSTD_e_end = 2 } ZSTD_EndDirective;
typedef enum {
    ZSTD_reset_session_only = 1,
    ZSTD_reset_parameters = 2,
    ZSTD_reset_session_and_parameters = 3
} ZSTD_ResetDirective;

/* only the parameter ids this repository sets; values verified against 1.5.7 */
typedef enum {
    ZSTD_c_compressionLevel = 100,
    ZSTD_c_windowLog = 101,
    ZSTD_c_enableLongDistanceMatching = 160,
    ZSTD_c_contentSizeFlag = 200,
    ZSTD_c_checksumFlag = 201,
    ZSTD_c_nbWorkers = 400,
    ZSTD_c_experimentalParam9 = 1006,  /* ZSTD_c_stableInBuffer */
    ZSTD_c_experimentalParam12 = 1009, /* ZSTD_c_validateSequences */
    ZSTD_c_experimentalParam17 = 1014, /* ZSTD_c_enableSeqProducerFallback */
    ZSTD_c_experimentalParam18 = 1015, /* ZSTD_c_maxBlockSize */
    ZSTD_c_experimentalParam19 = 1016, /* ZSTD_c_searchForExternalRepcodes */
    ZSTD_c_experimentalParam20 = 1017  /* ZSTD_c_blockSplitterLevel */
} ZSTD_cParameter;
#define ZSTD_c_stableInBuffer ZSTD_c_experimentalParam9
#define ZSTD_c_validateSequences ZSTD_c_experimentalParam12
#define ZSTD_c_enableSeqProducerFallback ZSTD_c_experimentalParam17
#define ZSTD_c_maxBlockSize ZSTD_c_experimentalParam18
#define ZSTD_c_searchForExternalRepcodes ZSTD_c_experimentalParam19
#define ZSTD_c_blockSplitterLevel ZSTD_c_experimentalParam20

typedef struct { const void *src; size_t size; size_t pos; } ZSTD_inBuffer;
typedef struct { void *dst; size_t size; size_t pos; } ZSTD_outBuffer;

#define ZSTD_SEQUENCE_PRODUCER_ERROR ((size_t)(-1))

typedef size_t (*ZSTD_sequenceProducer_F)(
    void *sequenceProducerState,
    ZSTD_Sequence *outSeqs, size_t outSeqsCapacity,
    const void *src, size_t srcSize,
    const void *dict, size_t dictSize,
    int compressionLevel,
    size_t windowSize);

unsigned ZSTD_versionNumber(void);
const char *ZSTD_versionString(void);
unsigned ZSTD_isError(size_t code);
const char *ZSTD_getErrorName(size_t code);
size_t ZSTD_compressBound(size_t srcSize);
size_t ZSTD_sequenceBound(size_t srcSize);

ZSTD_CCtx *ZSTD_createCCtx(void);
size_t ZSTD_freeCCtx(ZSTD_CCtx *cctx);
size_t ZSTD_CCtx_setParameter(ZSTD_CCtx *cctx, ZSTD_cParameter param, int value);
size_t ZSTD_CCtx_reset(ZSTD_CCtx *cctx, ZSTD_ResetDirective reset);
ZSTD_bounds ZSTD_cParam_getBounds(ZSTD_cParameter cParam);
size_t ZSTD_compress2(ZSTD_CCtx *cctx, void *dst, size_t dstCapacity,
                      const void *src, size_t srcSize);
size_t ZSTD_compressStream2(ZSTD_CCtx *cctx, ZSTD_outBuffer *output,
                            ZSTD_inBuffer *input, ZSTD_EndDirective endOp);
void ZSTD_registerSequenceProducer(ZSTD_CCtx *cctx, void *sequenceProducerState,
                                   ZSTD_sequenceProducer_F sequenceProducer);

ZSTD_DCtx *ZSTD_createDCtx(void);
size_t ZSTD_freeDCtx(ZSTD_DCtx *dctx);
size_t ZSTD_decompress(void *dst, size_t dstCapacity, const void *src, size_t compressedSize);
size_t ZSTD_decompressDCtx(ZSTD_DCtx *dctx, void *dst, size_t dstCapacity,
                           const void *src, size_t srcSize);

#if defined(__cplusplus)
}
#endif
#endif /* QZSTD_ZSTD_ABI_H */
