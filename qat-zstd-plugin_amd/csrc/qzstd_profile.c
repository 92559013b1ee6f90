/*
 * qzstd_profile.c — zstd level -> search profile, and the LDS budget derived from it.
 *
 * The reference passes the zstd level straight to the accelerator as the QAT
 * compression level (/root/reference/src/qatseqprod.c:1154, session params
 * :935-946) and only accepts 1..12 (:86-87, :1132-1137).  Here the level selects
 * the parameters of the LDS match-finder.  Plain C, no GPU needed.
 *
 * LDS budget per workgroup (gfx950: 160 KiB = 163840 B per CU, target: TWO workgroups per CU):
 *     32 KiB ring of recent block bytes (+128 B wrap mirror) + 4*tableSize + near table 4<<tileLog
 *     + 2 tiles of per-position parse words + per-window emission records + 64 B control
 *     (+ 96 B of item words for the resident service) = 72 560 B with 8192 table entries at tileLog 9 (levels 1-2).
 */
#include "qzstd_hip.h"

#include <string.h>

#define QZ_LDS_MAX 163840u
#define QZ_LDS_CTRL (64u + 16u) /* control words + the 16 bytes below the kernel's first LDS address (kLdsBase) */
#define QZ_LDS_SVC 96u          /* the resident service's item words, at the end of the allocation (csrc/qzstd_kernels.hip) */


int qzstd_hip_profile_for_level(int level, size_t blockSize, qzstd_hip_profile_t *out)
{
    const int repcodes = (level & QZSTD_HIP_LEVEL_REPCODES) != 0;
    level &= ~QZSTD_HIP_LEVEL_REPCODES;
    if (level < 1 || level > 12 || !out) return -1;
    memset(out, 0, sizeof(*out));
    (void)blockSize; /* the profile does not depend on the block size: the LDS footprint is fixed (ring + tables) */
    /* levels 1-2: 8192 entries, no long table = 72.6 KB of LDS -> two blocks per CU;
     * levels 3-4: 16384 entries + a second table keyed by 8 bytes (the double-fast idea of zstd's
     * levels 3-4) = 138.1 KB -> one block per CU;
     * levels >= 5: exact hash chains over a 4-byte hash (zstd: greedy / lazy / lazy2 / btlazy2); the size of the
     * head table hardly matters there (a collision costs one chain step): 5888 entries -> two blocks per CU */
    {
        const int chains = level >= 5;
#ifndef QZ_CHAIN_TABLE
#define QZ_CHAIN_TABLE 5888u /* head-table entries of the chain levels (A/B builds: make variant XFLAGS=-DQZ_CHAIN_TABLE=n) */
#endif
        out->tableSize = chains ? QZ_CHAIN_TABLE : (level >= 3 ? 16384u : 8192u); /* below the chain levels powers of two: the slot is a shift (round 5; 16000 / 6400 and a multiply-high before) */
        out->longSize = (!chains && level >= 3) ? 8192u : 0u;
        out->tileLog = 9;
#ifndef QZ_CAP_HI
#define QZ_CAP_HI 48u /* candidate cap of levels 9-12 (A/B builds).  Round 5: ONE cap at every level — the 16-byte head and one step of 32 bytes; a capped
                       * match is extended to its true end when the parse takes it, the cap only blunts the lazy comparison between two long candidates.
                       * Levels 9-12 128 -> 64 -> 48: kernel time -11 % and another -4 %, compressed size +0.06 % / +0.05 % over eleven corpora */
#endif
#ifndef QZ_CAP_MID
#define QZ_CAP_MID 48u /* ... of levels 5-8: 64 -> 48, kernel time -7 % (level 6: 82.4 -> 76.3 ms per GiB), compressed size +0.08 % */
#endif
#ifndef QZ_DEPTH_HI
#define QZ_DEPTH_HI 40u /* links walked at levels 10-12 (round 5: 48 -> 40 = ten entries of four links, another -9 % of kernel time for +0.05 % of compressed
                         * size; round 4: 64 -> 48; the repeat-aware parse of these levels leaves room in the 2 % bound:
                         * compressed size +0.1 % over eleven corpora (web-log 32 KiB blocks 0.996 -> 0.994 of software), kernel time -14 % (config 4's
                         * shape) to -18 % (128 KiB blocks): the walk's cost is linear in the links) */
#endif
        out->capLen = level >= 9 ? QZ_CAP_HI : (level >= 5 ? QZ_CAP_MID : 48u); /* one 32-byte step after the 16-byte head; capped matches are extended when taken */
        out->minMatch = 4;
        out->farLog1 = 12;
        out->farLog2 = 16;
        out->lazy = chains ? 4u : 3u; /* 4 = the lazy rules compare gains (length and offset cost), not lengths */
        out->backExt = 4;
        out->nearTab = chains ? 0u : 1u;
        out->window = 0;
        out->hashBytes = chains ? 4u : 5u;
        out->extLog = 11;
        /* libzstd turns repeat offsets of external sequences into repcodes only from level 10 (or when the
         * caller sets ZSTD_c_searchForExternalRepcodes); without that, short repeat matches cost a full offset */
        out->repWin = (repcodes || level >= 10) ? 16u : 0u;
        /* levels >= 5: links walked per position (software zstd: 2^searchLog = 4..128 attempts plus repcodes; the
         * producer API gives no repcodes below level 10, which deeper chains make up for) */
#ifndef QZ_DEPTH_6
#define QZ_DEPTH_6 12u /* links walked at level 6 (round 5: 16 -> 12 = three entries of four links: kernel time -19 %, compressed size +0.28 % over eleven corpora,
                        * 0.999 -> 0.997 of software zstd level 6 — whose own chain search makes 8 attempts; the walk's cost is linear in the links) */
#endif
        out->chainDepth = level >= 10 ? QZ_DEPTH_HI : (level >= 9 ? 64u : (level >= 7 ? 32u : (level >= 6 ? QZ_DEPTH_6 : (level >= 5 ? 8u : 0u))));
        /* the tables are updated per 64 positions, in position order, at the chain levels (there: exactly) and at
         * level 2, which buys its better ratio with them */
        out->subTileLog = (chains || level == 2) ? 6u : 0u;
        /* no match crosses a 4 KiB boundary (and the repeat-aware parse forgets its offsets there), so that a lone block
         * can be parsed as up to 32 work items in parallel with the same result (qzstd_hip_block_t.parseFrom: an item
         * parses any run of whole segments).  Costs 0.10-0.15 % of compressed size and takes the GPU time
         * of a lone 128 KiB level-1 block from 566 us (one item) over 164 us (four) to 46 us (32 items) */
        out->segLog = 12u;
    }
    return 0;
}

size_t qzstd_hip_sequence_bound(size_t srcSize)
{
    return srcSize / 3 + 1 + srcSize / 1024 + 1;
}

/* device scratch of one launch: per position of every work item the chain entry of levels >= 5 (16 B, four links) and — for the
 * history pass of segment items — a dense array of 4 B per position.  Below the chain levels only that array: the parse words of a launch,
 * which parses after its tile loop (csrc/qzstd_kernels.hip: DEFER). */
size_t qzstd_hip_workspace_bytes(int level, uint32_t nBlocks, uint32_t maxBlockLen)
{
    qzstd_hip_profile_t p;
    if (maxBlockLen > QZSTD_HIP_BLOCK_MAX || qzstd_hip_profile_for_level(level, maxBlockLen, &p)) return 0;
#ifndef QZ_PLAIN_DEFER
#define QZ_PLAIN_DEFER 1 /* csrc/qzstd_kernels.hip must agree */
#endif
#ifndef QZ_REP_DEFER
#define QZ_REP_DEFER 1
#endif
    if (!p.chainDepth) /* one word per position (the plain parse only uses the front of every 64: its starts) + the windows' start masks, 8 B per 64 positions */
        return (p.repWin ? QZ_REP_DEFER != 0 : QZ_PLAIN_DEFER != 0) ? (size_t)nBlocks * ((((size_t)maxBlockLen + 511u) & ~(size_t)511u) / 8u * 33u) : 0;
    return (size_t)nBlocks * (((size_t)maxBlockLen + 511u) & ~(size_t)511u) * (4u * QZSTD_HIP_CHAIN_ENTRY_LINKS + 4u); /* a chain entry of QZSTD_HIP_CHAIN_ENTRY_LINKS links + the first link again, dense */
}

#ifndef QZ_RING
#define QZ_RING 32768u
#endif
#ifndef QZ_PARSE_LAG
#define QZ_PARSE_LAG 2u /* csrc/qzstd_kernels.hip must agree (make variant passes XFLAGS to both); 3 / 4: the decoupled parse wave experiment of round 6 */
#endif
#define QZ_RING_BYTES (QZ_RING + 128u) /* ring of recent block bytes + wrap mirror (csrc/qzstd_kernels.hip: kRing) */

/* LDS per workgroup: independent of the block size — 72 560 B at levels 1-2 (two workgroups per CU) */
size_t qzstd_hip_lds_bytes(int level, uint32_t maxBlockLen)
{
    qzstd_hip_profile_t p;
    size_t need, lag;
    if (maxBlockLen > QZSTD_HIP_BLOCK_MAX || qzstd_hip_profile_for_level(level, maxBlockLen, &p)) return 0;
    /* tiles between matching and emission (csrc/qzstd_kernels.hip: kLagT): the decoupled parse wave of levels 1-4 keeps QZ_PARSE_LAG tiles of
     * parse words and emission records; the chain levels and the repeat-aware parse run it in lock-step with the matchers: 2 */
    lag = (p.chainDepth || p.repWin) ? 2u : QZ_PARSE_LAG;
    need = (size_t)QZ_RING_BYTES
           + 4u * p.tableSize        /* hash table                                    */
           + 4u * p.longSize         /* 8-byte-key table (levels >= 3)                */
           + (4u << p.tileLog)     /* tile-local near table / the current tile's chain links (levels >= 5) */
           + lag * ((4u << p.tileLog) + 32u) /* per-position parse words (+ override spill), `lag` tiles in flight */
           + lag * ((1u << p.tileLog) >> 6) * 32u /* per-window emission records, as many tiles */
           + (p.chainDepth ? (4u << p.tileLog) : 0u) /* chain levels: slot | tag of the tile's positions, for the insert wave */
           + QZ_LDS_CTRL + QZ_LDS_SVC;
    return need <= QZ_LDS_MAX ? need : 0;
}
