/*
 * qzstd_oracle.c — CPU oracle (plain C).  TEST INFRASTRUCTURE ONLY; see the header
 * for the scope and parity status ("parity unpinned" at sequence level against the
 * reference, pinned through libzstd round trips / LZ4s vectors / software sizes).
 *
 * Sequential definition of the match-finder the HIP kernel implements
 * (qat-zstd-plugin_amd/csrc/qzstd_kernels.hip).  The definition is written so a
 * workgroup can evaluate it in parallel and still get the same bits:
 *
 *   0. Levels >= 5 (profile.chainDepth != 0): EXACT hash chains, the sequential semantics of zstd's own
 *      lazy levels — position p sees every earlier position with its 4-byte hash slot, newest first,
 *      chainDepth links deep, and keeps the candidate with the highest gain (qzo_candidates_chain).
 *      On the GPU the matcher waves take turns in position order and resolve same-slot positions inside
 *      a wave with ballots, so the result does not depend on scheduling either.
 *   1. CANDIDATES of levels 1-4, tile by tile (tile = 1<<tileLog consecutive positions):
 *      every position p of the tile reads the hash-table entry of its 4-byte
 *      hash as it was BEFORE the tile (=> newest earlier-tile position with that
 *      hash), then all positions of the tile are inserted; concurrent inserts to
 *      one slot resolve to the largest position (GPU: ds_max_u32).  Optionally a
 *      second, tile-local "earliest occurrence" table finds matches whose source
 *      lies inside the current tile (GPU: ds_min_u32 on a scratch table).
 *   2. LENGTHS: common-prefix length of candidate and position, capped at capLen.
 *   3. PARSE: greedy (or one-step lazy) left-to-right selection over the
 *      per-position (length, offset) array; a chosen match that hit the cap is
 *      extended to its true end (bounded by extLog); optional short backward extension into the
 *      pending literals.
 *   4. EMISSION: {offset, litLength, matchLength}, then the trailing-literals
 *      delimiter — the output contract of QZSTD_decLz4s,
 *      /root/reference/src/qatseqprod.c:1013-1091 (and :1309-1313 for a block
 *      with no match at all: exactly one {lit=srcSize,0,0}).
 */
#include "qzstd_oracle.h"

#include <stdlib.h>
#include <string.h>

#define QZO_TAG_BITS 14u
#define QZO_TAG_MASK ((1u << QZO_TAG_BITS) - 1u)
#define QZO_HASH_PRIME 2654435761u
#define QZO_NEAR_EMPTY 0xFFFFFFFFu

static inline uint32_t qzo_rd32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v; /* little endian hosts only (x86-64 / the GPU) */
}

/* ---- level table ------------------------------------------------------------
 * Search effort tiers.  The reference forwards the zstd level 1..12 unchanged as
 * the QAT "compLevel" (src/qatseqprod.c:1154) and rejects anything else
 * (:1132-1137); here the level picks the search profile.  Table sizes follow the
 * kernel's LDS budget: block bytes + table + 8 KiB <= 160 KiB.
 * MUST match qzstd_hip_profile_for_level() in qat-zstd-plugin_amd/csrc/qzstd_profile.c
 * (tests/test_host_cpu.py: test_profile_tables_agree compares the two tables field by field).
 */
int qzo_profile_for_level(int level, size_t blockSize, qzo_profile_t *out)
{
    const int repcodes = (level & QZO_LEVEL_REPCODES) != 0;
    level &= ~QZO_LEVEL_REPCODES;
    if (level < 1 || level > 12 || !out) return -1;
    memset(out, 0, sizeof(*out));
    (void)blockSize; /* the profile does not depend on the block size (the kernel's LDS footprint is fixed) */
    /* levels 1-2: 8192 entries, no long table = 72.6 KB of LDS, two blocks per CU; levels 3-4: a bigger
     * table plus a second table keyed by 8 bytes (the double-fast idea of zstd's levels 3-4), one block per CU;
     * levels >= 5: exact hash chains (zstd: greedy / lazy / lazy2 / btlazy2 over a 4-byte hash), where the size
     * of the head table hardly matters (a collision costs one chain step): 5888 entries, two blocks per CU */
    const int chains = level >= 5;
    out->tableSize = chains ? 5888u : (level >= 3 ? 16384u : 8192u); /* (round 5: powers of two below the chain levels — the slot is a shift, not a multiply-high; 16000 / 6400 before) */
    out->longSize = (!chains && level >= 3) ? 8192u : 0u;
    out->tileLog = 9;
    out->capLen = 48u; /* every level (round 5; levels 5-8 had 64, levels 9-12 128): the 16-byte head + one step of 32 bytes, capped matches are extended when taken */
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
    out->chainDepth = level >= 10 ? 40u : (level >= 9 ? 64u : (level >= 7 ? 32u : (level >= 6 ? 12u : (level >= 5 ? 8u : 0u)))); /* (round 5: level 6 16 -> 12, levels 10-12 48 -> 40) */
    /* the tables are updated per 64 positions, in position order, at the chain levels (there: exactly) and at
     * level 2, which buys its better ratio with them */
    out->subTileLog = (chains || level == 2) ? 6u : 0u;
    /* no match crosses a 4 KiB boundary (and the repeat-aware parse forgets its offsets there), so that a lone block can be
     * parsed as up to 32 work items in parallel (each workgroup inserts the block before its item, parses its item: any
     * run of whole segments) with the same result.  Measured cost in compressed size against no boundaries at all, system +
     * synthetic corpus: 0.15 % at level 1, 0.10 % at level 3, 0.11 % at level 6, 0.10-0.12 % at level 12 (32 KiB
     * boundaries: 0.01-0.02 %) — see DESIGN.md §4.6 */
    out->segLog = 12u;
    return 0;
}

/* ---- candidate phase -------------------------------------------------------- */

#define QZO_HASH_PRIME2 0x85EBCA77u
/* 32-bit mix of the first hashBytes (4..7) bytes at a position: ONE 32-bit multiply, the bytes behind the fourth (at most three: < 2^24) come
 * in through a 24 x 24-bit product (round 5: the kernel's v_mul_u32_u24 is a full-rate instruction, its 32-bit multiplies quarter-rate — the
 * insert pass in front of a service item is bound by them).  The 8-byte key of the second table (qzo_mix8) keeps two full multiplies. */
static inline uint32_t qzo_mix(const uint8_t *p, uint32_t hashBytes)
{
    uint32_t lo = qzo_rd32(p), hi = 0;
    if (hashBytes > 4) hi = qzo_rd32(p + hashBytes - 4) >> (8u * (8u - hashBytes)); /* bytes 4.. (hashBytes <= 7: below 2^24) */
    return (lo * QZO_HASH_PRIME) ^ (hi * (QZO_HASH_PRIME2 & 0xFFFFFFu));
}

/* main table slot: multiply-high range reduction (any table size) */
static inline uint32_t qzo_slot(uint32_t m, uint32_t tableSize)
{
    return (uint32_t)(((uint64_t)m * tableSize) >> 32);
}
/* tile-local table slot: top tileLog bits */
static inline uint32_t qzo_near_slot(uint32_t m, uint32_t tileLog)
{
    return tileLog ? m >> (32u - tileLog) : 0u;
}
/* 14-bit check tag from bits the slot functions barely use */
static inline uint32_t qzo_tag(uint32_t m)
{
    return (m >> 3) & QZO_TAG_MASK;
}

/* end of p's segment (profile.segLog), or n */
static inline uint32_t qzo_seg_end(const qzo_profile_t *pf, uint32_t p, uint32_t n)
{
    uint32_t e;
    if (!pf->segLog) return n;
    e = ((p >> pf->segLog) + 1u) << pf->segLog;
    return e < n ? e : n;
}

/* a position takes part (is looked up and inserted) only if the bytes it hashes lie inside its segment: a workgroup
 * that parses one segment sees the block up to the segment's end and must come to the same tables */
static inline int qzo_hashable(const qzo_profile_t *pf, uint32_t p, uint32_t n, uint32_t bytes)
{
    return p + bytes <= qzo_seg_end(pf, p, n);
}

static inline uint32_t qzo_prefix_len(const uint8_t *src, uint32_t q, uint32_t p, uint32_t maxLen)
{
    uint32_t l = 0;
    while (l < maxLen && src[q + l] == src[p + l]) l++;
    return l;
}

typedef struct {
    uint32_t len; /* 0 = no match */
    uint32_t off;
} qzo_cand_t;

/* 32-bit mix of the 8 bytes at a position (the key of the optional "long" table) */
static inline uint32_t qzo_mix8(const uint8_t *p)
{
    return (qzo_rd32(p) * QZO_HASH_PRIME) ^ (qzo_rd32(p + 4) * QZO_HASH_PRIME2);
}

static void qzo_candidates(const qzo_profile_t *pf, const uint8_t *src, uint32_t n,
                           qzo_cand_t *cand, uint32_t *tbl, uint32_t *near, uint32_t *tblL)
{
    const uint32_t nl = pf->longSize && n >= 8u ? n - 7u : 0u; /* positions that have 8 bytes for the long table */
    const uint32_t nh = n >= pf->hashBytes ? n - pf->hashBytes + 1 : 0; /* hashable positions */
    const uint32_t T = 1u << pf->tileLog;
    const uint32_t S = pf->subTileLog ? 1u << pf->subTileLog : T;
    uint32_t t0, s0, p;

    memset(tbl, 0, sizeof(uint32_t) * pf->tableSize);
    if (pf->longSize) memset(tblL, 0, sizeof(uint32_t) * pf->longSize);
    for (p = 0; p < n; p++) cand[p].len = 0, cand[p].off = 0;

    for (t0 = 0; t0 < nh; t0 += T) {
        const uint32_t t1 = t0 + T < nh ? t0 + T : nh;
        if (pf->nearTab) {
            /* tile-local table of the EARLIEST position per slot (GPU: ds_min_u32) */
            memset(near, 0xFF, sizeof(uint32_t) << pf->tileLog);
            for (p = t0; p < t1; p++) {
                const uint32_t m = qzo_mix(src + p, pf->hashBytes);
                const uint32_t hn = qzo_near_slot(m, pf->tileLog);
                if (!qzo_hashable(pf, p, n, pf->hashBytes)) continue;
                const uint32_t e = ((p - t0) << QZO_TAG_BITS) | qzo_tag(m);
                if (e < near[hn]) near[hn] = e;
            }
        }
        /* look-up and insertion go sub-tile by sub-tile (= the whole tile when subTileLog is 0) */
        for (s0 = t0; s0 < t1; s0 += S) {
            const uint32_t s1 = s0 + S < t1 ? s0 + S : t1;
            for (p = s0; p < s1; p++) {
                const uint32_t v = qzo_rd32(src + p);
                const uint32_t m = qzo_mix(src + p, pf->hashBytes);
                const uint32_t room = qzo_seg_end(pf, p, n) - p; /* a match never leaves its segment */
                const uint32_t cap = pf->capLen < room ? pf->capLen : room;
                uint32_t bestLen = 0, bestOff = 0;
                if (!qzo_hashable(pf, p, n, pf->hashBytes)) continue;
                /* probe 1: newest position of EARLIER (sub-)tiles in this slot (table as it was before the sub-tile) */
                const uint32_t e = tbl[qzo_slot(m, pf->tableSize)];
                if (e != 0 && (e & QZO_TAG_MASK) == qzo_tag(m)) {
                    const uint32_t q = (e >> QZO_TAG_BITS) - 1u;
                    const uint32_t off = p - q;
                    if ((pf->window == 0 || off <= pf->window) && qzo_rd32(src + q) == v) {
                        bestLen = qzo_prefix_len(src, q, p, cap);
                        bestOff = off;
                    }
                }
                /* probe 3 (levels >= 3): newest position of EARLIER tiles whose first 8 bytes hash alike */
                if (p < nl && qzo_hashable(pf, p, n, 8u)) {
                    const uint32_t m8 = qzo_mix8(src + p);
                    const uint32_t eL = tblL[qzo_slot(m8, pf->longSize)];
                    if (eL != 0 && (eL & QZO_TAG_MASK) == qzo_tag(m8)) {
                        const uint32_t q = (eL >> QZO_TAG_BITS) - 1u;
                        if (qzo_rd32(src + q) == v) {
                            const uint32_t l = qzo_prefix_len(src, q, p, cap);
                            if (l > bestLen) { bestLen = l; bestOff = p - q; } /* must be strictly longer */
                        }
                    }
                }
                /* probe 2: earliest position of THIS tile in the near slot, if it lies before p */
                if (pf->nearTab) {
                    const uint32_t en = near[qzo_near_slot(m, pf->tileLog)];
                    if (en != QZO_NEAR_EMPTY && (en & QZO_TAG_MASK) == qzo_tag(m)) {
                        const uint32_t q = t0 + (en >> QZO_TAG_BITS);
                        if (q < p && qzo_rd32(src + q) == v) {
                            const uint32_t l = qzo_prefix_len(src, q, p, cap);
                            /* longer wins; on a tie the nearer source (this one) wins */
                            if (l >= bestLen) { bestLen = l; bestOff = p - q; }
                        }
                    }
                }
                cand[p].len = bestLen;
                cand[p].off = bestOff;
            }
            /* insert the sub-tile: ascending order == "largest position wins" (GPU: ds_max_u32) */
            for (p = s0; p < s1; p++) {
                const uint32_t m = qzo_mix(src + p, pf->hashBytes);
                if (!qzo_hashable(pf, p, n, pf->hashBytes)) continue;
                tbl[qzo_slot(m, pf->tableSize)] = ((p + 1u) << QZO_TAG_BITS) | qzo_tag(m);
                if (p < nl && qzo_hashable(pf, p, n, 8u)) {
                    const uint32_t m8 = qzo_mix8(src + p);
                    tblL[qzo_slot(m8, pf->longSize)] = ((p + 1u) << QZO_TAG_BITS) | qzo_tag(m8);
                }
            }
        }
    }
}

/* gain of a match in quarter bytes: 4 per matched byte minus the bit length of the offset */
static inline int qzo_gain(uint32_t len, uint32_t off)
{
    return (int)(4u * len) - (int)(31u - (uint32_t)__builtin_clz(off + 1u));
}

/*
 * Candidates of the chain levels (>= 5): exact hash chains.  tbl[slot] = newest position + 1 with that slot of
 * the first hashBytes (4) bytes, chain[p] = what the slot held when p was inserted.  Position p walks
 * chainDepth links from the slot's content (collisions count as links), measures every link whose first
 * four bytes equal its own (length capped at capLen) and keeps the one with the highest gain; on equal gain
 * the nearer (earlier visited) one stays.  Then p is inserted.  Same semantics as the hash-chain match-finder of
 * zstd's lazy levels.  On the GPU the head table lives in LDS, chain[] in device memory; the matcher waves take
 * turns in position order, and positions of one wave that share a slot are ordered with ballots.
 */
static void qzo_candidates_chain(const qzo_profile_t *pf, const uint8_t *src, uint32_t n,
                                 qzo_cand_t *cand, uint32_t *tbl, uint32_t *chain)
{
    const uint32_t nh = n >= pf->hashBytes ? n - pf->hashBytes + 1 : 0;
    uint32_t p;
    memset(tbl, 0, sizeof(uint32_t) * pf->tableSize);
    for (p = 0; p < n; p++) cand[p].len = 0, cand[p].off = 0;
    for (p = 0; p < nh; p++) {
        const uint32_t v = qzo_rd32(src + p);
        const uint32_t sl = qzo_slot(qzo_mix(src + p, pf->hashBytes), pf->tableSize);
        const uint32_t room = qzo_seg_end(pf, p, n) - p; /* a match never leaves its segment */
        const uint32_t cap = pf->capLen < room ? pf->capLen : room;
        uint32_t link, d, bestLen = 0, bestOff = 0;
        int bg = 0;
        if (!qzo_hashable(pf, p, n, pf->hashBytes)) continue; /* hashes bytes of the next segment: takes no part */
        link = tbl[sl];
        chain[p] = link;
        tbl[sl] = p + 1u;
        for (d = 0; d < pf->chainDepth && link != 0u; d++) {
            const uint32_t q = link - 1u;
            if ((pf->window == 0 || p - q <= pf->window) && qzo_rd32(src + q) == v) {
                const uint32_t l = qzo_prefix_len(src, q, p, cap);
                const int g = qzo_gain(l, p - q);
                if (l >= 4u && (bestLen == 0u || g > bg)) { bestLen = l; bestOff = p - q; bg = g; }
            }
            link = chain[q];
        }
        cand[p].len = bestLen;
        cand[p].off = bestOff;
    }
}

/* ---- parse + emission ------------------------------------------------------- */

static inline uint32_t qzo_min_len(const qzo_profile_t *pf, uint32_t off)
{
    return pf->minMatch + (off >> pf->farLog1 ? 1u : 0u) + (off >> pf->farLog2 ? 1u : 0u);
}
static inline int qzo_take(const qzo_profile_t *pf, const qzo_cand_t *c)
{
    return c->len != 0 && c->len >= qzo_min_len(pf, c->off);
}

/* start flag of the plain parse: a usable candidate that none of the lazy rules defers.  The rules never look
 * across a 64-position window edge (one wave decides a window).  lazy 1..3: by length; lazy 4 (chain levels): by
 * gain, the thresholds of zstd's lazy2 (a match one position on must gain more than 4 quarter bytes, two on more
 * than 7) */
#define QZO_LAZY_T1 4
#define QZO_LAZY_T2 7
static inline int qzo_is_start(const qzo_profile_t *pf, const qzo_cand_t *cand, uint32_t nh, uint32_t p)
{
    if (!qzo_take(pf, &cand[p])) return 0;
    if (pf->lazy >= 4) {
        const int g = qzo_gain(cand[p].len, cand[p].off);
        if (p + 1 < nh && (p & 63u) < 63u && qzo_take(pf, &cand[p + 1]) && qzo_gain(cand[p + 1].len, cand[p + 1].off) > g + QZO_LAZY_T1) return 0;
        if (p + 2 < nh && (p & 63u) < 62u && qzo_take(pf, &cand[p + 2]) && qzo_gain(cand[p + 2].len, cand[p + 2].off) > g + QZO_LAZY_T2) return 0;
        return 1;
    }
    if (pf->lazy && p + 1 < nh && (p & 63u) < 63u && qzo_take(pf, &cand[p + 1]) && cand[p + 1].len > cand[p].len) return 0;
    if (pf->lazy >= 2 && p + 2 < nh && (p & 63u) < 62u && qzo_take(pf, &cand[p + 2]) && cand[p + 2].len > cand[p].len) return 0;
    if (pf->lazy >= 3 && p + 3 < nh && (p & 63u) < 61u && qzo_take(pf, &cand[p + 3]) && cand[p + 3].len > cand[p].len + 2u)
        return 0;
    return 1;
}

/* a match found with its length capped at `from` bytes is extended to its true end, but never past the
 * end of the NEXT 1<<extLog cell: bounds the parallel extension work (a longer repeat simply continues
 * as another sequence) */
static inline uint32_t qzo_extend(const qzo_profile_t *pf, const uint8_t *src, uint32_t n, uint32_t p, uint32_t off,
                                  uint32_t from)
{
    const uint32_t l0 = ((p >> pf->extLog) + 2u) << pf->extLog;
    const uint32_t se = qzo_seg_end(pf, p, n);
    const uint32_t lim = l0 < se ? l0 : se;
    const uint32_t q = p - off;
    uint32_t L = from;
    while (p + L < lim && src[q + L] == src[p + L]) L++;
    return L;
}

/* "price" of taking a match, in quarter bytes saved: 4 per matched byte minus the bits of the offset;
 * a repeated offset is nearly free.  0 = no match.  Always > 0 for a usable candidate. */
#define QZO_REP_CAP 32u   /* repeat-offset probes compare this many bytes; a full hit always wins */
#define QZO_REP_MIN 3u
static inline uint32_t qzo_hash_gain(const qzo_profile_t *pf, const qzo_cand_t *c)
{
    if (!qzo_take(pf, c)) return 0;
    return 4u * c->len + 32u - (31u - (uint32_t)__builtin_clz(c->off + 1u));
}
static inline uint32_t qzo_rep_gain(uint32_t l, uint32_t r)
{
    if (l < QZO_REP_MIN) return 0;
    return l >= QZO_REP_CAP ? 1000u - r : 4u * l + 36u - r;
}

/*
 * Parse, repeat-offset aware variant (profile.repWin != 0: levels >= 10, or any level | QZO_LEVEL_REPCODES).
 * Same candidates; the parse walks window by window: the repWin positions from the cursor (never across a tile
 * edge) are offered {candidate, repeat 1, repeat 2} — the repeats being the last two distinct offsets, probed over
 * QZO_REP_CAP bytes — and the first position whose best option is not beaten by the next position (by more than
 * 4 quarter bytes of gain) or the one after (by more than 11) is taken; a window without any option is skipped.
 * The deferral looks at most two positions past the window.  libzstd encodes such offsets as repcodes when
 * ZSTD_c_searchForExternalRepcodes is on.  Shaped for the kernel's parse wave: one lane per window position, the
 * repeat probes as two byte-equality ballots.
 */
static size_t qzo_parse_rep(const qzo_profile_t *pf, const uint8_t *src, uint32_t n, uint32_t nh,
                            const qzo_cand_t *cand, qzo_seq_t *out, size_t cap, uint32_t parseFrom)
{
    uint32_t cur = parseFrom, anchor = parseFrom, rep[2] = { 0u, 0u };
    uint32_t repSeg = pf->segLog ? parseFrom >> pf->segLog : 0u; /* the segment the repeat offsets were collected in */
    size_t ns = 0;
    while (cur < nh) {
        const uint32_t tileEnd = ((cur >> pf->tileLog) + 1u) << pf->tileLog;
        /* no match — a repeat neither — starts in the last hashBytes - 1 positions of a segment (the positions that are
         * not hashable, qzo_hashable): what holds at the end of the block holds at every boundary, so that the parse of an
         * item that ends at a boundary and the parse of the whole block agree there */
        const uint32_t segEnd = qzo_seg_end(pf, cur, n);
        const uint32_t startEnd = segEnd - pf->hashBytes + 1u; /* segEnd >= hashBytes: cur < nh lies in it */
        const uint32_t lim = tileEnd < startEnd ? tileEnd : startEnd;
        uint32_t W, V;
        uint32_t G[34], opt[34], k, r, q = 0, off = 0, L = 0, b = 0, floor;
        int found = 0;
        if (cur >= startEnd) { cur = segEnd; continue; } /* on to the next segment (or the end) */
        W = pf->repWin < lim - cur ? pf->repWin : lim - cur; /* positions on offer */
        V = W + 2u < lim - cur ? W + 2u : lim - cur;         /* ... + look-ahead for the deferral */
        if (pf->segLog && (cur >> pf->segLog) != repSeg) { /* a new segment starts without repeat offsets */
            rep[0] = rep[1] = 0u;
            repSeg = cur >> pf->segLog;
        }
        for (k = 0; k < V; k++) {
            const uint32_t p = cur + k;
            G[k] = qzo_hash_gain(pf, &cand[p]);
            opt[k] = 0;
            for (r = 0; r < 2u; r++) {
                if (rep[r] != 0u) {
                    const uint32_t room = qzo_seg_end(pf, p, n) - p;
                    const uint32_t mx = room < QZO_REP_CAP ? room : QZO_REP_CAP;
                    const uint32_t g = qzo_rep_gain(qzo_prefix_len(src, p - rep[r], p, mx), r);
                    if (g > G[k]) { G[k] = g; opt[k] = 1u + r; }
                }
            }
        }
        for (k = 0; k < W && !found; k++) {
            if (G[k] == 0u) continue;
            if (k + 1u < V && G[k + 1u] > G[k] + 4u) continue;
            if (k + 2u < V && G[k + 2u] > G[k] + 11u) continue;
            found = 1;
            q = cur + k;
            if (opt[k]) {
                const uint32_t room = qzo_seg_end(pf, q, n) - q;
                const uint32_t mx = room < QZO_REP_CAP ? room : QZO_REP_CAP;
                off = rep[opt[k] - 1u];
                L = qzo_prefix_len(src, q - off, q, mx);
                if (L == QZO_REP_CAP) L = qzo_extend(pf, src, n, q, off, L);
            } else {
                off = cand[q].off;
                L = cand[q].len;
                if (L == pf->capLen) L = qzo_extend(pf, src, n, q, off, L);
            }
        }
        if (!found) { cur += W; continue; }
        floor = pf->segLog ? (q >> pf->segLog) << pf->segLog : 0u; /* never backwards across the start of q's segment */
        if (floor < anchor) floor = anchor;
        while (b < pf->backExt && q - b > floor && q - off - b > 0 && src[q - b - 1] == src[q - off - b - 1]) b++;
        if (ns + 1 >= cap - 1) return QZO_ERROR; /* src/qatseqprod.c:1073-1076 */
        out[ns].offset = off;
        out[ns].litLength = q - b - anchor;
        out[ns].matchLength = L + b;
        out[ns].rep = 0;
        ns++;
        if (off != rep[0]) { rep[1] = rep[0]; rep[0] = off; }
        cur = anchor = q + L;
    }
    out[ns].offset = 0; /* trailing literals delimiter, src/qatseqprod.c:1037-1045 */
    out[ns].litLength = n - anchor;
    out[ns].matchLength = 0;
    out[ns].rep = 0;
    ns++;
    if (ns >= cap - 1) ns = QZO_ERROR; /* src/qatseqprod.c:1318 */
    return ns;
}

size_t qzo_find_sequences(const qzo_profile_t *pf, const uint8_t *src, size_t srcSize,
                          qzo_seq_t *out, size_t cap)
{
    return qzo_find_sequences_from(pf, src, srcSize, 0, out, cap);
}

size_t qzo_find_sequences_from(const qzo_profile_t *pf, const uint8_t *src, size_t srcSize, size_t parseFrom,
                               qzo_seq_t *out, size_t cap)
{
    const uint32_t n = (uint32_t)srcSize;
    uint32_t nh;
    qzo_cand_t *cand;
    uint32_t *tbl, *near, *tblL, *chain;
    uint32_t p = (uint32_t)parseFrom, anchor = (uint32_t)parseFrom;
    size_t ns = 0;

    if (!pf || !out || cap < 2 || srcSize > QZO_BLOCK_MAX || (srcSize && !src)) return QZO_ERROR;
    if (pf->tableSize < 256 || pf->tableSize > (1u << 18) || pf->tileLog > 10 || pf->minMatch < 3 ||
        pf->capLen < pf->minMatch + 2 || pf->repWin > 32 || pf->lazy > 4 || (pf->chainDepth && (pf->nearTab || pf->longSize)) || pf->chainDepth > 64 || pf->subTileLog > pf->tileLog || (pf->subTileLog && pf->subTileLog < 4) || pf->hashBytes < 4 || pf->hashBytes > 7 || pf->extLog < 8 || pf->extLog > 17 || pf->longSize > (1u << 18))
        return QZO_ERROR;
    if (pf->segLog && (pf->segLog < pf->tileLog || pf->segLog > 17)) return QZO_ERROR;
    if (parseFrom && (!pf->segLog || (parseFrom & ((1u << pf->segLog) - 1u)) || parseFrom >= srcSize)) return QZO_ERROR;
    nh = n >= pf->hashBytes ? n - pf->hashBytes + 1 : 0;

    cand = (qzo_cand_t *)malloc(sizeof(qzo_cand_t) * (n + 1));
    tbl = (uint32_t *)malloc(sizeof(uint32_t) * pf->tableSize);
    near = (uint32_t *)malloc(sizeof(uint32_t) << pf->tileLog);
    tblL = (uint32_t *)malloc(sizeof(uint32_t) * (pf->longSize ? pf->longSize : 1u));
    chain = (uint32_t *)calloc((size_t)n + 1u, sizeof(uint32_t));
    if (!cand || !tbl || !near || !tblL || !chain) { free(cand); free(tbl); free(near); free(tblL); free(chain); return QZO_ERROR; }

    if (pf->chainDepth) qzo_candidates_chain(pf, src, n, cand, tbl, chain);
    else qzo_candidates(pf, src, n, cand, tbl, near, tblL);

    if (pf->repWin) {
        ns = qzo_parse_rep(pf, src, n, nh, cand, out, cap, (uint32_t)parseFrom);
        goto done;
    }
    while (p < nh) {
        uint32_t L, off, q, b = 0, floor;
        if (!qzo_is_start(pf, cand, nh, p)) { p++; continue; } /* no usable candidate, or deferred by a lazy rule */
        L = cand[p].len;
        off = cand[p].off;
        q = p - off;
        if (L == pf->capLen) L = qzo_extend(pf, src, n, p, off, L); /* hit the candidate-phase cap: extend to the true (bounded) end */
        /* backward extension into the pending literals, never across the start of p's segment */
        floor = pf->segLog ? (p >> pf->segLog) << pf->segLog : 0u;
        if (floor < anchor) floor = anchor;
        while (b < pf->backExt && p - b > floor && q - b > 0 && src[p - b - 1] == src[q - b - 1]) b++;
        if (ns + 1 >= cap - 1) { ns = QZO_ERROR; goto done; } /* src/qatseqprod.c:1073-1076 */
        out[ns].offset = off;
        out[ns].litLength = p - b - anchor;
        out[ns].matchLength = L + b;
        out[ns].rep = 0;
        ns++;
        p += L;
        anchor = p;
    }
    /* trailing literals delimiter, src/qatseqprod.c:1037-1045 */
    out[ns].offset = 0;
    out[ns].litLength = n - anchor;
    out[ns].matchLength = 0;
    out[ns].rep = 0;
    ns++;
    if (ns >= cap - 1) ns = QZO_ERROR; /* src/qatseqprod.c:1318 */
done:
    free(cand); free(tbl); free(near); free(tblL); free(chain);
    return ns;
}

size_t qzo_sequence_producer(void *state, qzo_seq_t *outSeqs, size_t outSeqsCapacity,
                             const void *src, size_t srcSize, const void *dict,
                             size_t dictSize, int compressionLevel, size_t windowSize)
{
    qzo_profile_t local;
    const qzo_profile_t *pf = (const qzo_profile_t *)state;
    /* guards of src/qatseqprod.c:1123-1137 */
    if (windowSize < (srcSize < 32 * 1024 ? srcSize : 32 * 1024) || dictSize > 0 || dict) return QZO_ERROR;
    if (compressionLevel < 1 || compressionLevel > 12) return QZO_ERROR;
    if (!pf) {
        if (qzo_profile_for_level(compressionLevel, srcSize, &local)) return QZO_ERROR;
        pf = &local;
    }
    return qzo_find_sequences(pf, (const uint8_t *)src, srcSize, outSeqs, outSeqsCapacity);
}

/* ---- validator / reconstructor ---------------------------------------------- */

int qzo_validate(const qzo_seq_t *s, size_t nb, size_t srcSize, size_t windowSize)
{
    size_t i, pos = 0;
    if (nb == 0) return -4;
    for (i = 0; i < nb; i++) {
        pos += s[i].litLength;
        if (i + 1 == nb) {
            if (s[i].matchLength != 0 || s[i].offset != 0) return -4;
        } else {
            if (s[i].matchLength < 3) return -3;
            if (s[i].offset == 0 || s[i].offset > pos) return -2;
            if (windowSize && s[i].offset > windowSize) return -5;
        }
        pos += s[i].matchLength;
        if (pos > srcSize) return -1;
    }
    return pos == srcSize ? 0 : -1;
}

size_t qzo_reconstruct_check(const qzo_seq_t *s, size_t nb, const uint8_t *src, size_t srcSize)
{
    uint8_t *dst = (uint8_t *)malloc(srcSize + 1);
    size_t i, pos = 0, bad = 0;
    if (!dst) return 1;
    for (i = 0; i < nb && !bad; i++) {
        uint32_t k;
        if (pos + s[i].litLength > srcSize) { bad = i + 1; break; }
        memcpy(dst + pos, src + pos, s[i].litLength); /* literals come from the block */
        pos += s[i].litLength;
        if (s[i].matchLength) {
            if (s[i].offset == 0 || s[i].offset > pos || pos + s[i].matchLength > srcSize) { bad = i + 1; break; }
            for (k = 0; k < s[i].matchLength; k++) dst[pos + k] = dst[pos + k - s[i].offset];
            pos += s[i].matchLength;
        }
    }
    if (!bad && (pos != srcSize || memcmp(dst, src, srcSize) != 0)) {
        /* locate the first sequence whose output differs */
        size_t j, q = 0;
        bad = nb + 1;
        for (j = 0; j < nb; j++) {
            size_t e = q + s[j].litLength + s[j].matchLength;
            if (e > srcSize || memcmp(dst + q, src + q, e - q) != 0) { bad = j + 1; break; }
            q = e;
        }
    }
    free(dst);
    return bad;
}

void qzo_seq_stats(const qzo_seq_t *s, size_t nb, uint64_t *sumMatch, uint64_t *sumLit, uint64_t *fnv1a)
{
    uint64_t m = 0, l = 0, h = 1469598103934665603ull;
    size_t i;
    int k;
    for (i = 0; i < nb; i++) {
        const uint32_t f[3] = { s[i].offset, s[i].litLength, s[i].matchLength };
        m += s[i].matchLength;
        l += s[i].litLength;
        for (k = 0; k < 3; k++) {
            int b;
            for (b = 0; b < 4; b++) { h ^= (f[k] >> (8 * b)) & 0xFF; h *= 1099511628211ull; }
        }
    }
    if (sumMatch) *sumMatch = m;
    if (sumLit) *sumLit = l;
    if (fnv1a) *fnv1a = h;
}

/* ---- LZ4s restatement (QZSTD_decLz4s, src/qatseqprod.c:1013-1091) ------------ */

size_t qzo_lz4s_decode(qzo_seq_t *outSeqs, size_t cap, const uint8_t *buf, size_t size)
{
    const uint8_t *ip = buf, *const end = buf + size;
    uint32_t pendingLit = 0; /* literals of "no match" tokens, merged forward (:1077-1084) */
    size_t n = 0;
    while (ip < end && size > 0) {
        const unsigned token = *ip++;
        size_t lit = token >> 4, ml;
        uint32_t off;
        if (lit == 15) { unsigned s; do { s = *ip++; lit += s; } while (s == 255); }
        ip += lit;
        if (ip == end) { /* last token: literals only (:1037-1045) */
            outSeqs[n].litLength = (uint32_t)lit + pendingLit;
            outSeqs[n].offset = 0;
            outSeqs[n].matchLength = 0;
            outSeqs[n].rep = 0;
            break;
        }
        off = (uint32_t)ip[0] | ((uint32_t)ip[1] << 8);
        ip += 2;
        ml = token & 15;
        if (ml == 15) { unsigned s; do { s = *ip++; ml += s; } while (s == 255); }
        if (ml != 0) {
            ml += 2;                      /* LZ4MINMATCH (:104, :1061) */
            outSeqs[n].offset = off;
            outSeqs[n].litLength = (uint32_t)lit + pendingLit;
            outSeqs[n].matchLength = (uint16_t)ml; /* 16-bit truncation (:1062) */
            outSeqs[n].rep = 0;
            pendingLit = 0;
            n++;
            if (n >= cap - 1) return QZO_ERROR; /* :1073-1076 */
        } else if (lit > 0) {
            pendingLit += (uint32_t)lit;
        }
    }
    if (ip != end) return QZO_ERROR; /* :1086-1089 */
    return n + 1;
}

static size_t qzo_put_run(uint8_t *dst, size_t pos, size_t cap, size_t v)
{
    /* v = length - 15 to spread over 255-run bytes */
    while (v >= 255) { if (pos >= cap) return QZO_ERROR; dst[pos++] = 255; v -= 255; }
    if (pos >= cap) return QZO_ERROR;
    dst[pos++] = (uint8_t)v;
    return pos;
}

size_t qzo_lz4s_encode(uint8_t *dst, size_t cap, const qzo_seq_t *s, size_t nb,
                       const uint8_t *src, size_t srcSize)
{
    size_t i, o = 0, pos = 0;
    for (i = 0; i < nb; i++) {
        const size_t lit = s[i].litLength;
        const int last = (i + 1 == nb);
        size_t mlCode = 0;
        if (!last) {
            if (s[i].matchLength < 3 || s[i].matchLength > 65535 || s[i].offset > 65535) return QZO_ERROR;
            mlCode = s[i].matchLength - 2;
        }
        if (o >= cap) return QZO_ERROR;
        dst[o++] = (uint8_t)(((lit >= 15 ? 15 : lit) << 4) | (mlCode >= 15 ? 15 : mlCode));
        if (lit >= 15 && (o = qzo_put_run(dst, o, cap, lit - 15)) == QZO_ERROR) return QZO_ERROR;
        if (pos + lit > srcSize || o + lit > cap) return QZO_ERROR;
        memcpy(dst + o, src + pos, lit);
        o += lit;
        pos += lit;
        if (last) break;
        if (o + 2 > cap) return QZO_ERROR;
        dst[o++] = (uint8_t)(s[i].offset & 0xFF);
        dst[o++] = (uint8_t)(s[i].offset >> 8);
        if (mlCode >= 15 && (o = qzo_put_run(dst, o, cap, mlCode - 15)) == QZO_ERROR) return QZO_ERROR;
        pos += s[i].matchLength;
    }
    return o;
}
