/*
 * qatseqprodfuzzer.c — adapter between upstream zstd's third-party sequence-producer fuzzing
 * hook (zstd/tests/fuzz/fuzz_third_party_seq_prod.h) and this plugin, so that zstd's own fuzz
 * targets (simple_round_trip, stream_round_trip, sequence_compression_api, ...) can drive the
 * GPU producer.  Same five symbols as the reference's adapter
 * (/root/reference/test/fuzzing/qatseqprodfuzzer.c:41-74); like there, TearDown does not stop
 * the device (the fuzzers call Setup/TearDown around every input).
 *
 * Build (needs a zstd source tree, which this image does not have):
 *   make -C qat-zstd-plugin_amd/test/fuzzing          # -> qatseqprodfuzzer.o
 *   cd zstd/tests/fuzz && ./fuzz.py build all --custom-seq-prod=<...>/qatseqprodfuzzer.o \
 *        --ldflags "<...>/lib/libqatseqprod.a -L/opt/rocm/lib -lamdhip64 -lstdc++"
 */
#include "qatseqprod.h"

size_t FUZZ_seqProdSetup(void)
{
    (void)QZSTD_startQatDevice(); /* a failed start is not fatal: the producer then reports an error per block */
    return 0;
}

size_t FUZZ_seqProdTearDown(void)
{
    return 0; /* keep the device up between inputs */
}

void *FUZZ_createSeqProdState(void)
{
    return QZSTD_createSeqProdState();
}

size_t FUZZ_freeSeqProdState(void *state)
{
    QZSTD_freeSeqProdState(state);
    return 0;
}

size_t FUZZ_thirdPartySeqProd(void *sequenceProducerState, ZSTD_Sequence *outSeqs, size_t outSeqsCapacity,
                              const void *src, size_t srcSize, const void *dict, size_t dictSize,
                              int compressionLevel, size_t windowSize)
{
    return qatSequenceProducer(sequenceProducerState, outSeqs, outSeqsCapacity, src, srcSize, dict, dictSize,
                               compressionLevel, windowSize);
}
