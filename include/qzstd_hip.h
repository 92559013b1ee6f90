/*
 * qzstd_hip.h — thin C ABI between the plain-C plugin (host/qatseqprod.c) and the
 * CDNA4 / gfx950 HIP side (csrc/qzstd_kernels.hip).  extern "C", plain pointers and
 * sizes, no C++ / torch types.
 *
 * Which reference interface each entry point replaces (the reference reaches its
 * accelerator through the Intel QAT driver API; none of that API is reproduced):
 *
 *   qzstd_hip_device_count          icp_adf_get_numDevices / icp_sal_userIsQatAvailable
 *                                   (/root/reference/src/qatseqprod.c:503,514)
 *   qzstd_hip_device_name           cpaDcInstanceGetInfo2 (:569)
 *   qzstd_hip_malloc / _free        qaeMemAllocNUMA / qaeMemFreeNUMA device-visible
 *   qzstd_hip_host_alloc / _free    buffers (:221,:235; QZSTD_calloc :216-246)
 *   qzstd_hip_stream_create/…       cpaDcStartInstance / cpaDcStopInstance (:835,:316)
 *   qzstd_hip_memcpy_h2d/_d2h       the DMA the QAT ring performs on submit/response
 *   qzstd_hip_find_sequences        cpaDcCompressData2 (:1245) + QZSTD_decLz4s
 *                                   (:1013-1091): the kernel emits ZSTD_Sequence
 *                                   arrays directly, no LZ4s intermediate
 *   qzstd_hip_stream_sync / _query  icp_sal_DcPollInstance (:1265)
 *   qzstd_hip_profile_for_level     CpaDcSessionSetupData.compLevel (:1154, :935-946)
 *
 * Every function returns 0 on success and a negative value on failure unless noted;
 * qzstd_hip_last_error() gives a printable reason (thread-local).
 */
#ifndef QZSTD_HIP_H
#define QZSTD_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#define QZSTD_HIP_BLOCK_MAX (1u << 17) /* ZSTD_BLOCKSIZE_MAX */
#define QZSTD_HIP_NSEQ_ERROR 0xFFFFFFFFu
#define QZSTD_HIP_SRC_ALIGN 16u        /* every block's srcOff must be a multiple of this */

/* Search profile (what a zstd level means to the match-finder).  Same field layout
 * as the oracle's qzo_profile_t; tests compare the two level tables. */
typedef struct {
    uint32_t tableSize; /* entries of the LDS hash table                                   */
    uint32_t tileLog;   /* look-up / insert granularity: tiles of 1<<tileLog positions     */
    uint32_t capLen;    /* candidate-phase match length cap                                */
    uint32_t minMatch;  /* minimum match length for near offsets                           */
    uint32_t farLog1;   /* offset >= 1<<farLog1 needs minMatch+1                           */
    uint32_t farLog2;   /* offset >= 1<<farLog2 needs minMatch+2                           */
    uint32_t lazy;      /* 0 greedy; 1..3 = lazy, looks that many positions ahead            */
    uint32_t backExt;   /* max backward extension of a chosen match                        */
    uint32_t nearTab;   /* 1 = tile-local "earliest occurrence" probe                      */
    uint32_t window;    /* max offset, 0 = whole block                                     */
    uint32_t hashBytes; /* bytes hashed per position (4..8)                                */
    uint32_t extLog;    /* a match never extends past the end of the next 1<<extLog cell       */
    uint32_t longSize;  /* entries of the second table keyed by 8 bytes (0 = none)               */
    uint32_t repWin;    /* 0 = plain parse; n = repeat-offset aware parse: after every match the next n positions
                         * are also tried with the last two offsets (for callers with
                         * ZSTD_c_searchForExternalRepcodes on, which is libzstd's default from level 10) */
    uint32_t chainDepth; /* 0 = table probes only; n = also walk the main table's predecessor chain (a per-block
                         * array in device memory: chain[p] = the slot's content before p's tile), n candidates deep */
    uint32_t subTileLog; /* 0 = the tables are updated once per tile; n = per 1<<n positions, in position order
                         * (GPU: the matcher waves take turns), so a position also sees the earlier sub-tiles of its tile */
    uint32_t segLog;     /* 0 = none; n = no match crosses a multiple of 1<<n (12 at every level): one block may then be submitted as
                         * several work items that each parse a run of whole segments (qzstd_hip_block_t.parseFrom) — what the
                         * per-block paths do to cut the latency of a lone request */
} qzstd_hip_profile_t;

/* One work item = one <=128 KiB block, parsed with no history
 * (reference contract: src/qatseqprod.h:103-105) — or a run of whole SEGMENTS of such a block: parseFrom != 0 (a multiple of
 * 1 << profile.segLog) makes the workgroup insert [0, parseFrom) into its
 * tables without parsing it and emit the sequences of [parseFrom, srcLen) only; srcLen is then the block up to the
 * end of the item's last segment.  The items' sequence lists, concatenated with the trailing literals carried over, equal the
 * sequences of the whole block submitted as one item. */
typedef struct {
    uint64_t srcOff;  /* byte offset of the block inside d_src, multiple of QZSTD_HIP_SRC_ALIGN */
    uint64_t seqOff;  /* index (in ZSTD_Sequence units) of the item's output region inside d_seqs */
    uint32_t srcLen;  /* <= QZSTD_HIP_BLOCK_MAX                                                  */
    uint32_t seqCap;  /* capacity of the output region, >= ZSTD_sequenceBound(srcLen - parseFrom) */
    uint32_t parseFrom; /* 0 = the whole block                                                    */
    uint32_t mark;      /* written into the fourth word (ZSTD_Sequence.rep) of every entry the item produces, the delimiter included; 0 on
                         * the launch paths (there the end of the kernel orders results and completion); the resident service puts the
                         * request's epoch there.  With QZSTD_HIP_MARK_COMPACT set the item's entries are PACKED (below) and the low 12 bits
                         * are their tag */
} qzstd_hip_block_t;

/* PACKED entries (round 6) — for result areas in pinned HOST memory.  The product paths let the kernel write its results straight over
 * PCIe, and at level 1 a launch writes 0.83 bytes of ZSTD_Sequence entries per input byte: measured, the level-1 kernel takes 19.7 ms per
 * GiB with its results in pinned host memory against 12.1 with them in device memory — it runs at the bus's write rate (43 GB/s of the
 * 57 GB/s the link carries one way), not at its own.  An entry has 52 bits of payload (offset < 2^17, litLength <= 2^17, matchLength <
 * 2^17 inside a 128 KiB block): packed into ONE 8-byte store together with a 12-bit tag it halves that traffic and still certifies itself
 * — an entry is taken when it shows the tag; the host wipes a result area whenever its tags start over (every 4 095 uses), so a tag that is
 * valid again cannot be found.  qzstd_hip_block_t.mark = QZSTD_HIP_MARK_COMPACT | tag (1 .. 4095); seqOff stays in 16-byte units (the
 * region of an item with seqCap packed entries is seqCap / 2 of them long); seqCap counts packed entries.
 *     bits  0..16 offset (0 = the delimiter)   17..34 litLength   35..51 matchLength   52..63 tag */
#define QZSTD_HIP_MARK_COMPACT 0x80000000u
#define QZSTD_HIP_PACK(off, lit, ml, tag) ((uint64_t)(off) | ((uint64_t)(lit) << 17) | ((uint64_t)(ml) << 35) | ((uint64_t)(tag) << 52))
#define QZSTD_HIP_PACKED_OFF(v) ((uint32_t)((v) & 0x1FFFFu))
#define QZSTD_HIP_PACKED_LIT(v) ((uint32_t)(((v) >> 17) & 0x3FFFFu))
#define QZSTD_HIP_PACKED_ML(v) ((uint32_t)(((v) >> 35) & 0x1FFFFu))
#define QZSTD_HIP_PACKED_TAG(v) ((uint32_t)((v) >> 52))

const char *qzstd_hip_last_error(void);

/* OR-ed into a `level` argument: the caller compresses with ZSTD_c_searchForExternalRepcodes enabled (libzstd
 * only does that by itself from level 10), so repeat-offset aware sequences pay off at every level */
#define QZSTD_HIP_LEVEL_REPCODES 0x100

/* ---- level -> profile (pure host function, no GPU needed) ---- */
int qzstd_hip_profile_for_level(int level, size_t blockSize, qzstd_hip_profile_t *out);
/* ZSTD_sequenceBound restated (zstd 1.5.x): srcSize/3 + 1 + srcSize/1024 + 1 */
size_t qzstd_hip_sequence_bound(size_t srcSize);
/* dynamic LDS bytes the kernel needs for a launch whose largest block is maxBlockLen */
size_t qzstd_hip_lds_bytes(int level, uint32_t maxBlockLen);

/* ---- device / memory / stream plumbing ---- */
int qzstd_hip_device_count(void); /* number of usable devices (gfx950: the only code object in the library), <0 on error;
                                   * devices are indexed 0..count-1 in that filtered order */
int qzstd_hip_device_name(int device, char *buf, size_t bufLen);
void *qzstd_hip_malloc(int device, size_t bytes);
void qzstd_hip_free(int device, void *dptr);
void *qzstd_hip_host_alloc(size_t bytes); /* pinned, portable across devices */
/* device-side address of a qzstd_hip_host_alloc() buffer: kernels may read descriptors from and write results
 * to pinned host memory directly (small latency-bound requests skip the D2H copies that way) */
void *qzstd_hip_host_device_ptr(void *hptr);
void qzstd_hip_host_free(void *hptr);
void *qzstd_hip_stream_create(int device);
void qzstd_hip_stream_destroy(int device, void *stream);
int qzstd_hip_stream_sync(int device, void *stream);
int qzstd_hip_stream_query(int device, void *stream); /* 0 done, 1 still running, <0 error */
/* bounded wait (the reference polls icp_sal_DcPollInstance for at most 2 s, src/qatseqprod.c:1099-1104,:1261-1272):
 * 0 = everything queued on the stream is done, 1 = still running after timeoutMs, <0 error */
int qzstd_hip_stream_wait(int device, void *stream, unsigned timeoutMs);
int qzstd_hip_memcpy_h2d(int device, void *stream, void *dst, const void *src, size_t bytes);
/* the same for a PINNED source, done by a kernel on the stream instead of the runtime's copy path (round 4: an announcing thread spent 1 ms
 * per 4 MiB inside hipMemcpyAsync): src_dev = qzstd_hip_host_device_ptr(pinned source); 16-byte aligned, bytes a multiple of 16 */
int qzstd_hip_copy_in(int device, void *stream, void *dst, const void *src_dev, size_t bytes);
int qzstd_hip_memcpy_d2h(int device, void *stream, void *dst, const void *src, size_t bytes);
int qzstd_hip_memset(int device, void *stream, void *dst, int value, size_t bytes);
/* strided device->host copy: `height` rows of `width` bytes, row r at src + r*spitch -> dst + r*dpitch */
int qzstd_hip_memcpy2d_d2h(int device, void *stream, void *dst, size_t dpitch, const void *src, size_t spitch,
                           size_t width, size_t height);

/*
 * The hot path.  Asynchronously launches the match-finder on `stream` (NULL = the
 * device's default stream) over nBlocks independent blocks.  All pointers are DEVICE
 * pointers (or pinned host pointers mapped into the device).
 *
 *   d_src     input bytes; block i occupies [srcOff_i, srcOff_i + srcLen_i); the buffer
 *             must stay readable up to the next multiple of 16 past each block end
 *   d_blocks  nBlocks descriptors
 *   d_seqs    ZSTD_Sequence array; block i writes entries [seqOff_i, seqOff_i + count_i)
 *   d_nseq    per block: number of sequences INCLUDING the trailing-literals delimiter
 *             (what qatSequenceProducer returns, src/qatseqprod.c:1090,:1323), or
 *             QZSTD_HIP_NSEQ_ERROR when count >= seqCap-1 (src/qatseqprod.c:1318)
 *   d_work    device scratch of at least qzstd_hip_workspace_bytes(level, nBlocks, maxBlockLen)
 *             bytes, private to this launch until it completes: REQUIRED AT EVERY LEVEL since
 *             round 6 (levels >= 5 keep their hash chains there; below them a launch leaves one
 *             parse word per position there and parses after its tile loop, eight 4 KiB segments
 *             at a time: 4 bytes per position)
 *
 * One workgroup per block; block bytes, hash table and parse scratch live in LDS.
 */
size_t qzstd_hip_workspace_bytes(int level, uint32_t nBlocks, uint32_t maxBlockLen);
int qzstd_hip_find_sequences(int device, void *stream, int level, const void *d_src,
                             const qzstd_hip_block_t *d_blocks, uint32_t nBlocks,
                             uint32_t maxBlockLen, void *d_seqs, uint32_t *d_nseq,
                             void *d_work, size_t workBytes);

/* diagnostics: workgroups of the level's launch kernel that fit one CU according to the runtime (registers, LDS, wave slots) */
int qzstd_hip_occupancy(int device, int level);

/*
 * The resident service: one block per request WITHOUT a launch (reference: the synchronous submit + poll of one
 * request on a DC instance, src/qatseqprod.c:1243-1272; many instances per device :905-928).  A request names the caller's
 * own buffers: the block staged in pinned host memory, a device staging buffer of the same size, a pinned result area and
 * one pinned count word per work item.  The block is cut into nItems work items of itemBytes (a multiple of the level's
 * segment size, 1 << profile.segLog; the last item takes the rest); item k parses [k * itemBytes, min(srcLen, (k + 1) *
 * itemBytes)), writes its sequences to hSeqs + k * seqCapPerItem and then — last, with a system-scope release — its count
 * (sequences including the item's delimiter, QZSTD_HIP_NSEQ_ERROR, or QZSTD_HIP_NSEQ_REJECTED) to hCount[k], which the
 * caller has zeroed and polls: no stream, no query.  The count says how many entries the item has; every ENTRY is one 16-byte
 * store that carries the request's epoch in its fourth word (ZSTD_Sequence.rep): the caller takes an entry when it shows the
 * epoch — nothing orders the count behind the entries of other waves on their ways to host memory (measured: up to microseconds
 * apart under load).  The items' lists, joined with the trailing literals carried over, are the sequences of the block.
 *
 * Resident kernels (one worker workgroup per CU + a one-wave dispatcher) are launched by the first request and leave
 * after QZSTD_HIP_SERVICE_IDLE_US (default 20000) without work, when memory is freed, or on qzstd_hip_service_stop().
 * Served: every level.  ONE multi-level worker serves levels 1-2 and 5-12 (with and without the repeat-aware parse) at the same time —
 * the request carries its level —, levels 3-4 (136 KB of LDS per workgroup) have a worker of their own: a request of a level the resident
 * workers do not serve is handed back while they are resident.  The workers leave when a launch needs the LDS they hold: a launch and a service whose workgroups do not fit a CU's 160 KB
 * together take turns (batch launches of levels 3-4 against any service, any batch launch against a service of levels 3-4).
 *
 *   qzstd_hip_service_submit   0 = queued;  1 = not served (QZSTD_HIP_SERVICE=0, a level the resident workers do not serve, the
 *                              service is down): the caller takes the launch path;  < 0 = error
 */
#define QZSTD_HIP_NSEQ_REJECTED 0xFFFFFFFEu
/* Progressive staging (round 5): a count word the caller sets to QZSTD_HIP_NSEQ_STAGING instead of 0 says "slice k of hSrc is not staged
 * yet": the worker that gets item k waits for the word to leave that value before it reads the slice.  The caller may therefore queue
 * the request FIRST and copy the block into hSrc behind it, slice by slice, turning every word to 0 (compare-and-swap: a request that was
 * handed back has REJECTED there) as its slice is in — the staging copy of a 128 KiB block (13 us) then overlaps the request's way to
 * the first worker (8 us).  qzstd_hip_service_progressive(): 1 = this device layer's workers look (the resident kernels), 0 = they do
 * not: stage everything before qzstd_hip_service_submit (a synchronous stand-in, e.g. the tests' mock). */
#define QZSTD_HIP_NSEQ_STAGING 0xFFFFFFFCu
int qzstd_hip_service_progressive(int device);
#define QZSTD_HIP_SVC_MAX_ITEMS 32u
#define QZSTD_HIP_SVC_MAX_SLOTS 1024u
typedef struct {
    const void *hSrc;   /* pinned host memory: the block, readable up to the next multiple of 16 */
    void *dSrc;         /* device memory, same size: the items copy their slices there */
    void *hSeqs;        /* pinned: nItems x seqCapPerItem ZSTD_Sequence entries */
    uint32_t *hCount;   /* pinned: nItems words, zeroed by the caller before the call */
    uint32_t srcLen, itemBytes, nItems, seqCapPerItem;
    uint32_t slot;      /* < QZSTD_HIP_SVC_MAX_SLOTS: one request in flight per slot (its slices' flags) */
    uint32_t epoch;     /* 1 .. 0xFFFFFF, different from the slot's previous request */
    void *dWork;        /* chain levels (>= 5): device scratch of the request, QZSTD_HIP_SVC_WORK_BYTES, shared by its items (every item
                         * links the block before it, and leaves the chain entries of one item's range to the items after it);
                         * NULL at the other levels */
} qzstd_hip_svc_req_t;
#ifndef QZSTD_HIP_CHAIN_ENTRY_LINKS
#define QZSTD_HIP_CHAIN_ENTRY_LINKS 4u /* links per chain entry in device memory (levels >= 5), 4 or 8: 4 B each, one dependent gather per entry in the walk.
                                        * Eight were measured in round 4 (bit-exact): level 5 61 -> 118, level 6 102 -> 194, level 12 269 -> 347 ms per GiB — the entries of
                                        * positions whose predecessors lie in the same tile are hopped together link by link from LDS, seven dependent reads instead of three */
#endif
#define QZSTD_HIP_SVC_WORK_BYTES ((size_t)QZSTD_HIP_BLOCK_MAX * (8u * QZSTD_HIP_CHAIN_ENTRY_LINKS + 4u) + (size_t)QZSTD_HIP_SVC_MAX_ITEMS * 5888u * 4u) /* chain entries of the history and of the
                                       * items' own positions, first links, and one published head table per item */
int qzstd_hip_service_submit(int device, int level, const qzstd_hip_svc_req_t *req);
int qzstd_hip_service_stop(int device);          /* asks the resident kernels to leave and waits for them; 0 = stopped */
/* called by a caller that waits for its request: if the service has left meanwhile (idle exit, a free, a launch that needed the LDS)
 * and requests wait in the ring, it is launched again (0); 1 = not now (memory is being freed, a launch that fills the LDS is in
 * flight): keep waiting, poke again; 2 = the service has been taken out of use (a request timed out) and has left: requests it
 * has not answered will not be answered */
int qzstd_hip_service_poke(int device, int level);
void qzstd_hip_service_mark_broken(int device);  /* a request timed out: stop and do not use the service again */
/* out[0] launches of the service, [1] requests queued, [2] requests refused (a level the resident workers do not serve), [3] broken,
 * [4] state (0 stopped, 1 running), [5] items finished, [6] items that gave up waiting for a slice, [7] workers */
int qzstd_hip_service_info(int device, unsigned long out[8]);
/* diagnostics the dispatcher refreshes while it runs: [0] its polls of the ring, [1] requests taken, [2] items queued, [3] worker
 * workgroups that have started, [4] items picked up, [5] items finished, [6] requests consumed over all launches, [7] 0 */
int qzstd_hip_service_debug(int device, unsigned long out[8]);
void *qzstd_hip_host_alloc_coherent(size_t bytes); /* pinned + mapped + fine-grained: what the host polls while a kernel writes it */

/* ---- NUMA placement (reference: every DMA buffer is allocated on a NUMA node, qaeMemAllocNUMA(size, node, 64),
 * /root/reference/src/qatseqprod.c:216-246): on a two-socket node half the GPUs hang off each socket, and pinned staging buffers,
 * result areas and request rings that sit on the other socket cross the inter-socket link on every access. */
/* host NUMA node the GPU is attached to (hipDeviceAttributeHostNumaId, else sysfs numa_node of its PCI function); -1 = unknown */
int qzstd_hip_device_numa_node(int device);
/* qzstd_hip_host_alloc() / _coherent() with the pages on `node` (node < 0: wherever the calling thread's policy puts them).
 * Best effort: where the kernel does not let the process set a memory policy, the allocation still succeeds, unplaced. */
void *qzstd_hip_host_alloc_on_node(size_t bytes, int node, int coherent);
/* NUMA node that holds the first page of a host buffer (-1 = cannot tell): diagnostics, tests */
int qzstd_hip_host_node_of(const void *hptr);

#if defined(__cplusplus)
}
#endif
#endif /* QZSTD_HIP_H */
