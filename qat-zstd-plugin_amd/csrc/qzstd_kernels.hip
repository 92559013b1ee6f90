/*
 * qzstd_kernels.hip — the block-level ZSTD sequence producer's match-finder for
 * AMD Instinct MI355X (CDNA4, gfx950), plus the thin C ABI of include/qzstd_hip.h.
 *
 * What it replaces: the reference hands each <=128 KiB block to QAT silicon
 * (cpaDcCompressData2, /root/reference/src/qatseqprod.c:1245) and then parses the
 * LZ4s stream it gets back into ZSTD_Sequence entries (QZSTD_decLz4s, :1013-1091).
 * Here one workgroup does both jobs for one block and writes ZSTD_Sequence entries
 * straight to HBM:
 *
 *   - the block streams from HBM into a 32 KiB LDS RING of its most recent bytes (kRing; 16-byte
 *     coalesced loads, 4.5 KiB ahead of the tile being matched); sources more than kNear (~25 KiB)
 *     back (a few % of the candidates) are compared from HBM/L2 instead, which keeps the LDS
 *     footprint at 65 KB: two blocks are resident per CU at levels 1-2 and 5-12;
 *   - a 4-byte-entry hash table ((position+1)<<14 | 14-bit tag) lives in LDS next to it
 *     (levels >= 3 add a second table keyed by 8 bytes);
 *   - positions are processed in tiles of 512: every position of a tile reads its slot
 *     (newest position of EARLIER tiles), then all insert with ds_max_u32, so the result does
 *     not depend on wave scheduling; a tile-local ds_min_u32 table finds sources inside the
 *     current tile;
 *   - the 8 matcher waves measure the candidate lengths (16 bytes, then 32 per step), apply the
 *     lazy start rules with three DPP shifts, and reduce every position to one packed word;
 *   - the serial greedy parse.  In the resident service's work items (one 4 KiB segment each) a dedicated 9th wave
 *     runs it in lock-step with the matchers as a 4-instruction scalar pointer chase
 *     (bitset / readlane / compare / select), extends capped matches cooperatively when it takes them, and
 *     publishes per-window records; the matcher waves then emit their chosen {offset, litLength, matchLength}
 *     entries ranked by a popcount prefix.  In the LAUNCH kernels (round 6; qz_item: DEFER) the
 *     parse and the emission run AFTER the tile loop, and the 9th wave ENDS before it (at the chain levels matcher
 *     wave 0 takes its ordered inserts over): the parse of a 4 KiB segment depends on nothing before
 *     the segment and the candidates do not depend on the parse, so the loop only matches (one parse word per
 *     position to the launch's scratch) and then the waves parse the block's segments side by side
 *     (parse_plain_windows / parse_rep_span<true>) and emit one lane per sequence.
 *
 * Kernel variants (template parameters of qzstd_find_sequences_kernel):
 *   HAS_LONG  levels >= 3: the second table;
 *   CHAIN     levels >= 5: hash chains, four links (16 B) per position in DEVICE memory (the workspace argument of
 *             qzstd_hip_find_sequences; below the chain levels the workspace holds the deferred parse's words),
 *             walked after the table probes;
 *   TURNS     level 2 and levels >= 5: the tables are updated per 64 positions, the matcher waves taking
 *             turns in position order;
 *   REP       levels >= 10, or any level | QZSTD_HIP_LEVEL_REPCODES: the repeat-offset aware parse
 *             (byte-wise ballot probe of the last two offsets on arrival at every match end).
 *
 * Integer byte matching: no MFMA.  The roofline that bounds it is HBM (block read once,
 * 16 B per sequence written); in practice it is bound by instruction issue at 18 waves per CU
 * (profiles/).
 *
 * The sequential definition of exactly this computation is oracle/qzstd_oracle.c
 * (test infrastructure); tests compare the two sequence-for-sequence.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <sched.h>
#include <time.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "qzstd_hip.h"

#ifndef QZ_REP_DEFER
#define QZ_REP_DEFER 1 /* the launch kernels of the repeat-aware parse BELOW THE CHAIN LEVELS (level 1-4 | REPCODES) parse AFTER their tile loop, eight segments at a
                        * time (qz_item: DEFER): 65.0 -> 23.5 ms per GiB at level 1 | REPCODES, 125.9 -> 40.4 at level 3 | REPCODES, bit-exact.  A/B: 0 = in the loop,
                        * one wave, as the resident service's items do; 2 = deferred at the chain levels too — measured SLOWER there (level 12: 185.8 -> 192.1 ms
                        * per GiB, config 4's shape 105.7 -> 107.8): the chain walk takes longer than the parse wave's serial chain, which the lock-step loop
                        * hides completely, so deferring it only adds the parse's own time */
#endif
#ifndef QZ_PLAIN_DEFER
#define QZ_PLAIN_DEFER 1 /* the launch kernels of levels 1-4 defer their plain parse and their emission in the same way (csrc/qzstd_profile.c must agree: the workspace) */
#endif
#ifndef QZ_CHAIN_SHIFT
#define QZ_CHAIN_SHIFT 1 /* chain levels: a tile's start flags are written in the next iteration's first interval (A/B: 0) */
#endif
#ifndef QZ_CHAIN_DEFER_EXT
#define QZ_CHAIN_DEFER_EXT 0 /* chain walk (round 6 A/B, 1): the 32-byte extensions of a step's links in a loop of their own, every lane taking ITS next one — bit-exact, +2 ... +7 % time (profiles/r06_ab_chain_walk.txt); 0 = inside
                              * every link's own block, as in rounds 3-5) */
#endif
/* Round 6: "progress-inverse priority".  The issue arbiter serves the OLDER waves of a SIMD first, so waves 4-7 of a workgroup reach every barrier last
 * and run the end of every interval alone, with nobody to hide their latencies behind (profiles/r06_level1_wave_timing_before.txt).  A matcher wave of the
 * tile levels therefore starts an interval at priority 2 and lowers it as it gets on (1 after the first part, 0 after the second): whoever is behind
 * wins the arbitration; the parse wave stays above them at 3.  Measured (profiles/r06_ab_small_steps.txt): level 1 11.91 -> 11.78 ms per GiB (-1.1 %),
 * level 3 10.47 -> 10.31 (-1.6 %); 2 = the same with priorities 1, 0, 0: half of that; a fixed raised priority for waves 4-7: nothing.  0 = off (A/B). */
#ifndef QZ_PROGRESS_PRIO
#define QZ_PROGRESS_PRIO 1
#endif
#define QZ_PRIO(n) do { if (QZ_PROGRESS_PRIO && !CHAIN) __builtin_amdgcn_s_setprio((n) >= QZ_PROGRESS_PRIO ? (n) + 1 - QZ_PROGRESS_PRIO : 0); } while (0) /* 1: 2,1,0   2: 1,0,0 */
/* A/B (round 6): the emission's common path trimmed — the "a capped match was extended" flag rides in bit 31 of the record's sequence base (no second LDS
 * request + scalar test per window), the own chosen bit is tested without a 64-bit shift by the lane (a quarter-rate instruction).  Fewer instructions,
 * bit-exact — and +2.8 % time at level 1 (12.25 vs 11.91 ms per GiB), +1.4 % at level 3: NOT the product (profiles/r06_ab_small_steps.txt). */
#ifndef QZ_EMIT_TRIM
#define QZ_EMIT_TRIM 0
#endif
constexpr uint32_t kExtFlag = 0x80000000u; /* QZ_EMIT_TRIM: set in ParseRecs.r3 (the window's first sequence index) when r4 / r5 hold extended lengths */
#ifndef QZ_CHAIN_HOIST_P
#define QZ_CHAIN_HOIST_P 0 /* A/B: 1 = the position's own 32 bytes behind its head requested and byte-aligned ONCE per tile, kept in registers over the whole walk */
#endif

namespace {

constexpr int kMatchWaves = 8;
constexpr int kMatchThreads = kMatchWaves * 64; /* one position per matcher thread per tile */
constexpr int kThreads = kMatchThreads + 64;     /* + 1 parse wave */
#ifndef QZ_RING
#define QZ_RING 32768u /* bytes of the LDS ring (csrc/qzstd_profile.c: QZ_RING_BYTES must agree).  32 KiB: two workgroups per CU at
                        * levels 1-2 and 5-12.  Measured A/B (bit-exact either way): 16 KiB = three per CU buys nothing at level 1 (12.32 vs 11.96 ms per
                        * GiB: the CU is VALU-bound, not latency-bound) and costs 13 % at the chain levels (more sources beyond the ring's reach) */
#endif
constexpr uint32_t kTileLog = 9;                 /* tile = 512 positions = kMatchThreads */
constexpr uint32_t kTile = 1u << kTileLog;
constexpr uint32_t kWin = kTile >> 6;            /* 64-position windows per tile (one per matcher wave) */
constexpr uint32_t kTagBits = 14;
constexpr uint32_t kTagMask = (1u << kTagBits) - 1u;
constexpr uint32_t kPrime1 = 2654435761u;
constexpr uint32_t kPrime2 = 0x85EBCA77u;
constexpr uint32_t kNone = 0xFFFFFFFFu;
/* Only a RING of the most recent block bytes lives in LDS (so that two workgroups fit on a CU):
 * position x sits at ring offset x mod kRing; the first kMirror bytes are mirrored behind the ring
 * so that a 36-byte read never has to wrap.  At iteration `it` the ring holds
 * [it*512 + 512 + kLook - kRing, it*512 + 512 + kLook); sources farther back than kNear bytes
 * ("far" candidates, a few %) are compared straight from HBM/L2 instead. */
constexpr uint32_t kRing = QZ_RING;  /* a power of two: x mod kRing is one AND */
constexpr uint32_t kRingMask = kRing - 1u;
constexpr uint32_t kMirror = 128u;
constexpr uint32_t kLook = 4608u;  /* bytes staged ahead of the current tile (covers the bounded extension) */
constexpr uint32_t kNear = kRing - kLook - 3u * kTile - 1024u; /* kRing - kLook - 3 tiles of pipeline lag - slack */
static_assert((kRing & kRingMask) == 0u && kRing >= 16384u, "the ring is a power of two of at least 16 KiB");
constexpr size_t kLdsPerCu = 163840u; /* 160 KB */
#ifdef QZ_EXP_LINKS4_SPACED /* experiment: four links per entry at the SPACING of eight (same gathers and hops as four, the lines of eight) */
constexpr uint32_t kEL = 4u;
#else
constexpr uint32_t kEL = QZSTD_HIP_CHAIN_ENTRY_LINKS; /* links per chain entry: the walk needs one dependent gather per kEL links */
#endif
constexpr uint32_t kEQ = QZSTD_HIP_CHAIN_ENTRY_LINKS / 4u; /* 16-byte words between consecutive entries (= per entry, outside the experiment) */
static_assert(kEL == 4u || kEL == 8u, "chain entries hold four or eight links");
constexpr uint32_t kLdsBase = 16u; /* first LDS byte the kernel uses (csrc/qzstd_profile.c: QZ_LDS_CTRL covers it) */
/* Tiles between the matchers and the emission of a tile = tiles of parse words and emission records kept in LDS (csrc/qzstd_profile.c: QZ_PARSE_LAG
 * must agree).  2 = the parse wave in lock-step with the matchers: THE PRODUCT at every level.  QZ_PARSE_LAG = 3 or 4 builds the round-6 experiment
 * for levels 1-4 (the round-5 verdict's "take the serial parse off the barrier-critical path"): a DECOUPLED parse wave that parses whatever tile is
 * ready, window by window, and joins the matchers' barriers when they are all waiting at one — bit-exact, and 14 % SLOWER (13.6 vs 11.9 ms per GiB
 * at level 1; profiles/r06_ab_decoupled_parse_wave.txt says why: running without pauses, the parse wave's serial chain takes 5 200 cycles per tile
 * and IS the tile's time, where the lock-step one took 4 600 and idled a quarter of the time). */
#ifndef QZ_PARSE_LAG
#define QZ_PARSE_LAG 2
#endif
constexpr uint32_t kParseLag = QZ_PARSE_LAG;
static_assert(kParseLag >= 2u && kParseLag <= 4u, "2 = the lock-step parse wave of rounds 1-5 at every level");
constexpr uint32_t kCtlArrive = 4u; /* control word: barriers the matcher waves have reached, summed over the eight waves (a HINT for the parse wave) */

struct LaunchArgs {
    const uint8_t *src;
    const qzstd_hip_block_t *blocks;
    uint4 *seqs; /* ZSTD_Sequence = 4 x u32 */
    uint32_t *nseq;
    qzstd_hip_profile_t prof; /* the level's search profile (block-size independent) */
    uint4 *chain;             /* levels >= 5: per-block chain entries (four links each), chainStride entries per block */
    uint32_t chainStride;     /* uint4 units between the scratch regions of consecutive work items */
    uint32_t chainEntries;    /* 16-byte words of entries per region (positions x kEQ); the region's dense array of first links (4 B per position) follows them */
    uint32_t pwWords;         /* below the chain levels (the deferred parse): parse words per block region; the windows' start masks (8 B per 64 positions) follow them */
    uint32_t orderedLds;      /* this device's LDS returns from ds_max_rtn, to lanes of one instruction that hit the same address, the
                               * values in lane order (probed once per device, probe_lds_order) */
#ifdef QZ_DEBUG_DUMP
    uint32_t dbg; /* profiling build only: ablation switches (QZSTD_HIP_ABLATE) */
#endif
};

typedef unsigned long long u64;

/* profiling build (-DQZ_DEBUG_DUMP): phases can be switched off and per-wave cycle counts are dumped */
#ifdef QZ_DEBUG_DUMP
#define QZ_ABLATED(bit) (args.dbg & (bit))
#define QZ_DBG args.dbg
#else
#define QZ_ABLATED(bit) false
#define QZ_DBG 0u
#endif

/* A workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global loads and stores
 * (s_waitcnt vmcnt(0)): every barrier of the tile loop then waits for the round trips of the ring refill's load and of the result
 * stores — to PINNED HOST memory in the product paths — although no wave reads what another wave wrote to global memory (below the
 * chain levels; there the barrier at the end of a tile stays a full one: chain entries).  Loads the compiler issued are still
 * waited for where their registers are used. */
#define QZ_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define QZ_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define QZ_RLX_SYSTEM __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

/* v_readlane_b32 with an unsigned result (the builtin returns int: a set bit 31 would sign-extend) */
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t rdfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 below(uint32_t c) { return c >= 64u ? ~0ull : ((1ull << c) - 1ull); }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return umax(umax(a, b), c); }

/* where the block's bytes can be read from: the LDS ring (recent bytes) or HBM (anything) */
/* explicit address spaces: through generic pointers the compiler falls back to FLAT loads for the ring */
typedef const __attribute__((address_space(3))) uint32_t *LdsWords;
typedef const __attribute__((address_space(1))) uint32_t *HbmWords;
struct Src {
    LdsWords ring; /* LDS, kRing + kMirror bytes */
    HbmWords g;    /* the block in HBM; 16-byte aligned, so aligned dword loads work */
    uint32_t nearLimit; /* sources farther back than this are compared from device memory: kNear, or 0xFFFFFFFF in the NEAR kernels (every
                         * block of the launch fits the ring: "offset > nearLimit" folds to false and the device-memory side of every
                         * compare is compiled out) */
};

/* dword index inside the ring of byte position a (a < 3 * kRing) */
__device__ __forceinline__ uint32_t ring_dw(uint32_t a)
{
    return (a & kRingMask) >> 2;
}

/* ring offsets kept incrementally (the matchers' hot path): x mod kRing moved by d < kRing, one add + one min */
__device__ __forceinline__ uint32_t ring_fwd(uint32_t r, uint32_t d) { return (r + d) & kRingMask; }
__device__ __forceinline__ uint32_t ring_back(uint32_t r, uint32_t d) { return (r - d) & kRingMask; }

/* the same with the ring offset r of position a already known */
template <int N>
__device__ __forceinline__ void load_dw_r(const Src &s, uint32_t a, uint32_t r, bool far, uint32_t (&D)[N])
{
    if (far) {
        const uint32_t d = a >> 2;
#pragma unroll
        for (int i = 0; i < N; i++) D[i] = s.g[d + i];
    } else {
        const uint32_t d = r >> 2;
#pragma unroll
        for (int i = 0; i < N; i++) D[i] = s.ring[d + i];
    }
}

/* N consecutive aligned dwords covering byte position a: from the ring, or from HBM when `far` */
template <int N>
__device__ __forceinline__ void load_dw(const Src &s, uint32_t a, bool far, uint32_t (&D)[N])
{
    if (far) {
        const uint32_t d = a >> 2;
#pragma unroll
        for (int i = 0; i < N; i++) D[i] = s.g[d + i];
    } else {
        const uint32_t d = ring_dw(a);
#pragma unroll
        for (int i = 0; i < N; i++) D[i] = s.ring[d + i];
    }
}

/* bit index of the lowest set bit of x, or 0xFFFFFFFF when x == 0 (raw v_ffbl_b32: exactly the "no
 * difference" value an unsigned min chain wants; OR-ing 32*i into it cannot wrap) */
__device__ __forceinline__ uint32_t first_diff_bit(uint32_t x)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

/* 4 bytes at an arbitrary position: two aligned dwords + v_alignbyte_b32 */
__device__ __forceinline__ uint32_t rd32u(const Src &s, uint32_t a, bool far)
{
    uint32_t D[2];
    load_dw<2>(s, a, far, D);
    return __builtin_amdgcn_alignbyte(D[1], D[0], a & 3u);
}

/* the same with the ring offset r of position a already known */
__device__ __forceinline__ uint32_t rd32_r(const Src &s, uint32_t a, uint32_t r, bool far)
{
    uint32_t D[2];
    load_dw_r<2>(s, a, r, far, D);
    return __builtin_amdgcn_alignbyte(D[1], D[0], a & 3u);
}

/* first mismatching byte (0..32) of the 32 bytes at p (ring) and at q (ring or HBM): 9 aligned
 * dwords per side are fetched with independent loads (ONE round trip), then compared in registers */
__device__ __forceinline__ uint32_t chunk_len(const Src &s, uint32_t p, uint32_t rp, uint32_t off, bool far)
{
    const uint32_t q = p - off, ps = p & 3u, qs = q & 3u; /* rp = ring offset of p; the source sits `off` before */
    uint32_t P[9], Q[9];
    load_dw_r<9>(s, p, rp, false, P);
    load_dw_r<9>(s, q, ring_back(rp, off), far, Q);
    /* first differing bit of dword i, | 32 i; -1 = "no difference" drops out of the unsigned min chain:
     * xor / ffbl / or per dword plus a few v_min3 and one shift, no compare-select chain */
    uint32_t B = 256u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t x = __builtin_amdgcn_alignbyte(P[i + 1], P[i], ps) ^ __builtin_amdgcn_alignbyte(Q[i + 1], Q[i], qs);
        B = umin(B, first_diff_bit(x) | (32u * (uint32_t)i));
    }
    return B >> 3;
}

/* first mismatching byte (0..16) of the 16 bytes at p (already byte-aligned in `oa`) and at q */
__device__ __forceinline__ uint32_t head_len(const Src &s, const uint32_t (&oa)[4], uint32_t q, uint32_t rq, bool far)
{
    const uint32_t qs = q & 3u;
    uint32_t Q[5];
    load_dw_r<5>(s, q, rq, far, Q);
    uint32_t B = 128u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t x = oa[i] ^ __builtin_amdgcn_alignbyte(Q[i + 1], Q[i], qs);
        B = umin(B, first_diff_bit(x) | (32u * (uint32_t)i));
    }
    return B >> 3;
}

/* the same, with the candidate's five dwords already requested (two candidates' loads in flight together) */
__device__ __forceinline__ uint32_t head_cmp(const uint32_t (&oa)[4], const uint32_t (&Q)[5], uint32_t qs)
{
    uint32_t B = 128u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t x = oa[i] ^ __builtin_amdgcn_alignbyte(Q[i + 1], Q[i], qs);
        B = umin(B, first_diff_bit(x) | (32u * (uint32_t)i));
    }
    return B >> 3;
}

/* first mismatching byte (0..32) of 32 bytes already byte-aligned in `pa` and the candidate's nine dwords, already requested */
__device__ __forceinline__ __attribute__((unused)) uint32_t tail_cmp(const uint32_t (&pa)[8], const uint32_t (&Q)[9], uint32_t qs)
{
    uint32_t B = 256u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t x = pa[i] ^ __builtin_amdgcn_alignbyte(Q[i + 1], Q[i], qs);
        B = umin(B, first_diff_bit(x) | (32u * (uint32_t)i));
    }
    return B >> 3;
}

/* first mismatching byte (0..32) of 32 bytes already byte-aligned in `pa` and the 32 bytes at q */
__device__ __forceinline__ uint32_t tail_len(const Src &s, const uint32_t (&pa)[8], uint32_t q, uint32_t rq, bool far)
{
    const uint32_t qs = q & 3u;
    uint32_t Q[9];
    load_dw_r<9>(s, q, rq, far, Q);
    uint32_t B = 256u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t x = pa[i] ^ __builtin_amdgcn_alignbyte(Q[i + 1], Q[i], qs);
        B = umin(B, first_diff_bit(x) | (32u * (uint32_t)i));
    }
    return B >> 3;
}

/* end of p's segment (profile.segLog: no match crosses a multiple of 1 << segLog), or n */
__device__ __forceinline__ uint32_t seg_end(const qzstd_hip_profile_t &pf, uint32_t p, uint32_t n)
{
    return pf.segLog ? umin(n, ((p >> pf.segLog) + 1u) << pf.segLog) : n;
}

__device__ __forceinline__ uint32_t min_len(const qzstd_hip_profile_t &pf, uint32_t off)
{
    return pf.minMatch + ((off >> pf.farLog1) ? 1u : 0u) + ((off >> pf.farLog2) ? 1u : 0u);
}

/* cooperative forward extension of a match that hit the candidate cap: 64 lanes x 16 bytes
 * (1 KiB) per step, never past `lim` */
template <bool AFAR = false> /* AFAR: the ring does not hold the match itself either (the deferred plain parse): both sides from device memory */
__device__ __forceinline__ uint32_t extend_match_from(const Src &s, uint32_t p, uint32_t off, uint32_t L, uint32_t lim,
                                                      uint32_t lane, const bool far /* uniform: the whole match has one offset */,
                                                      const uint32_t lastDw = 0xFFFFFFFFu /* AFAR: the last dword of the block's buffer that may be read */)
{
    for (;;) {
        const uint32_t a = p + L + 16u * lane;
        uint32_t ok = 0; /* bytes of this lane's 16 that match and lie below lim */
        if (a < lim) {
            const uint32_t b = a - off, as = a & 3u, bs = b & 3u;
            uint32_t A[5], B[5];
            if constexpr (AFAR) { /* (a dword behind the buffer's end only holds bytes at or behind lim — on the source's side: bytes that face such bytes —
                                   * and is never counted: clamped.  The ring-based callers only go to device memory for sources far before the block's end) */
#pragma unroll
                for (int i = 0; i < 5; i++) { A[i] = s.g[umin((a >> 2) + (uint32_t)i, lastDw)]; B[i] = s.g[umin((b >> 2) + (uint32_t)i, lastDw)]; }
            } else {
                load_dw<5>(s, a, false, A);
                if (far) {
#pragma unroll
                    for (int i = 0; i < 5; i++) B[i] = s.g[umin((b >> 2) + (uint32_t)i, lastDw)];
                } else load_dw<5>(s, b, false, B);
            }
            ok = 16u;
#pragma unroll
            for (int i = 3; i >= 0; i--) {
                const uint32_t x = __builtin_amdgcn_alignbyte(A[i + 1], A[i], as) ^ __builtin_amdgcn_alignbyte(B[i + 1], B[i], bs);
                if (x) ok = 4u * (uint32_t)i + ((uint32_t)__builtin_ctz(x) >> 3);
            }
            ok = umin(ok, lim - a);
        }
        const u64 bad = __ballot(ok < 16u);
        if (bad) {
            const uint32_t f = (uint32_t)__builtin_ctzll(bad);
            return L + 16u * f + rdlane(ok, f);
        }
        L += 1024u;
    }
}

__device__ __forceinline__ uint32_t extend_match(const Src &s, uint32_t p, uint32_t off, uint32_t L, uint32_t lim, uint32_t lane)
{
    return extend_match_from(s, p, off, L, lim, lane, off > s.nearLimit);
}

/* parse-wave state, uniform across the wave */
struct ParseState {
    uint32_t cur;    /* next position the parse stands on */
    uint32_t anchor; /* end of the last chosen match = start of pending literals */
    uint32_t nseq;   /* matches chosen so far */
};

/*
 * The (lazy) greedy parse of one tile, by the parse wave: a lean scalar pointer chase.  The
 * matcher waves have reduced every position to ONE word (kept in LDS for two tiles):
 *     bits  0..7   nx   where the parse stands after taking the match that starts here: the next
 *                       start at/after its end (< 64), the exit cursor (64..191), or 255 = the
 *                       candidate hit the length cap and must be extended first
 *     bits  8..14  ns   the next start at/after this position (64 = none left in the window)
 *     bits 15..31       the match length, or for a capped candidate its offset
 * so one parse step is bitset / readlane / compare / select with a single taken branch.  A
 * capped match is extended here, cooperatively, only when the parse actually takes it.  All of
 * the tile's windows are fetched up front (one LDS wait).  Per window it publishes {chosen mask,
 * literal anchor at entry, index of the first sequence, up to two (lane, extended length)}.
 */
constexpr uint32_t kSrecWords = 8;
constexpr uint32_t kPvStride = kTile + 8u; /* parse words of one tile (+ room for the override of its last position) */
constexpr uint32_t kNxCapped = 255u;

__device__ __forceinline__ uint32_t pack_pos(uint32_t nx, uint32_t ns, uint32_t payload)
{
    return nx | (ns << 8) | (payload << 15);
}

/* vec with lane L replaced by the (uniform) val: ONE v_writelane_b32 (as `lane == L ? val : vec` the compiler emits a
 * compare + select, and the parse wave's VALU instructions weigh on the SIMD it shares with two matcher waves).  The lane
 * is an immediate: a second SGPR operand would break the constant-bus rule. */
template <uint32_t L>
__device__ __forceinline__ uint32_t wrlane(uint32_t vec, uint32_t val)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(val), "n"(L));
    return vec;
}

/* From cursor c (< 64) of a window: j = first start at/after it (>= 64: none, then jn = 64), the chase along the
 * in-window successors (4 scalar instructions + 1 taken branch per sequence; every start passed is set in `chosen`),
 * j = the last start taken, jn = where it leaves the window (64..191 = exit cursor, kNxCapped = extend first),
 * pl = the payload of j (length, or the offset of a capped match) */
#define QZ_CHASE(c)                                                                   \
    asm volatile("v_readlane_b32 %[j], %[wd], %[cc]\n"                               \
                 "s_movk_i32 %[jn], 64\n"                                            \
                 "s_bfe_u32 %[j], %[j], 0x70008\n"                                   \
                 "s_cmp_gt_u32 %[j], 63\n"                                           \
                 "s_cbranch_scc1 2f\n"                                               \
                 "1:\n"                                                              \
                 "s_bitset1_b64 %[ch], %[j]\n"                                       \
                 "v_readlane_b32 %[jn], %[nx], %[j]\n"                               \
                 "s_cmp_lt_u32 %[jn], 64\n"                                          \
                 "s_cselect_b32 %[j], %[jn], %[j]\n"                                 \
                 "s_cbranch_scc1 1b\n"                                               \
                 "v_readlane_b32 %[pl], %[wd], %[j]\n"                               \
                 "s_lshr_b32 %[pl], %[pl], 15\n"                                     \
                 "2:\n"                                                              \
                 : [ch] "+s"(chosen), [j] "=&s"(j), [jn] "=&s"(jn), [pl] "=&s"(pl)    \
                 : [nx] "v"(nx), [wd] "v"(wordW), [cc] "s"(c)                         \
                 : "scc");

/* the records of the windows a parse_tile call covers: lane w collects window w's */
struct ParseRecs {
    uint32_t r0, r1, r2, r3, r4, r5;
};

/* one window of the parse; W is a template parameter so that the record lanes are immediates */
template <uint32_t W>
__device__ __forceinline__ void parse_window(const qzstd_hip_profile_t &pf, const Src &src, const uint32_t wordW,
                                             uint32_t base, uint32_t n, uint32_t lane, ParseState &st, ParseRecs &r)
{
    constexpr uint32_t w = W;
    const uint32_t w0 = base + 64u * w;
    const uint32_t anchorIn = st.anchor, seqBase = st.nseq;
    u64 chosen = 0;
    uint32_t ext0 = 0, ext1 = 0; /* (lane << 24 | extended length) of up to two taken capped matches */
    const uint32_t c0 = st.cur - w0; /* the cursor never lies before the window */
    if (c0 < 64u) {
        const uint32_t nx = wordW & 0xFFu;
        uint32_t j, jn, pl;
        QZ_CHASE(c0)
        uint32_t cEnd = jn, eEnd = j < 64u ? j + pl : st.anchor - w0;
        if (__builtin_expect(jn == kNxCapped, 0)) {
            for (;;) {
                /* the match just taken hit the candidate cap: extend it to its true (bounded) end */
                const uint32_t pj = w0 + j;
                const uint32_t lim = umin(seg_end(pf, pj, n), ((pj >> pf.extLog) + 2u) << pf.extLog);
                const uint32_t xl = extend_match(src, pj, pl, pf.capLen, lim, lane);
                if (!ext0) ext0 = (j << 24) | xl; else ext1 = (j << 24) | xl;
                cEnd = eEnd = j + xl;
                if (cEnd >= 64u) break;
                QZ_CHASE(cEnd)
                if (j >= 64u) { cEnd = 64u; break; }
                cEnd = jn; eEnd = j + pl;
                if (jn != kNxCapped) break;
            }
        }
        st.nseq += (uint32_t)__popcll(chosen);
        st.cur = w0 + cEnd;
        st.anchor = w0 + eEnd;
    }
    r.r0 = wrlane<W>(r.r0, (uint32_t)chosen);
    r.r1 = wrlane<W>(r.r1, (uint32_t)(chosen >> 32));
    r.r2 = wrlane<W>(r.r2, anchorIn);
    r.r3 = wrlane<W>(r.r3, (QZ_EMIT_TRIM && ext0) ? (seqBase | kExtFlag) : seqBase);
    if (ext0) { /* rare */
        r.r4 = wrlane<W>(r.r4, ext0);
        r.r5 = wrlane<W>(r.r5, ext1);
    }
}

template <uint32_t W, uint32_t W_END>
__device__ __forceinline__ void parse_windows(const qzstd_hip_profile_t &pf, const Src &src, const uint32_t (&word)[kWin],
                                              uint32_t base, uint32_t n, uint32_t lane, ParseState &st, ParseRecs &r)
{
    if constexpr (W < W_END) {
        parse_window<W>(pf, src, word[W], base, n, lane, st, r);
        parse_windows<W + 1u, W_END>(pf, src, word, base, n, lane, st, r);
    }
}

template <uint32_t W_BEGIN, uint32_t W_END>
__device__ __forceinline__ void parse_tile(const qzstd_hip_profile_t &pf, const Src &src, const uint32_t *pv,
                                           uint32_t *srecOut, uint32_t base, uint32_t n, uint32_t lane, ParseState &st)
{
    uint32_t word[kWin];
#pragma unroll
    for (uint32_t w = W_BEGIN; w < W_END; w++) word[w] = pv[64u * w + lane];
    /* pin the loads here: otherwise the compiler sinks each one into its window and the serial chain pays the LDS
     * latency once per window instead of once per call */
#pragma unroll
    for (uint32_t w = W_BEGIN; w < W_END; w++) asm volatile("" : "+v"(word[w]));
    ParseRecs r = { 0u, 0u, 0u, 0u, 0u, 0u };
    parse_windows<W_BEGIN, W_END>(pf, src, word, base, n, lane, st, r);
    if (lane >= W_BEGIN && lane < W_END) {
        *reinterpret_cast<uint4 *>(srecOut + lane * kSrecWords) = make_uint4(r.r0, r.r1, r.r2, r.r3);
        srecOut[lane * kSrecWords + 4u] = r.r4;
        srecOut[lane * kSrecWords + 5u] = r.r5;
    }
}

/*
 * Repeat-offset aware parse (profile.repWin != 0; oracle: qzo_parse_rep).  The parse wave walks window by
 * window: lane k stands for position cursor+k (repWin = 16 positions on offer + 2 of look-ahead, never across the
 * tile edge) and weighs {candidate, repeat 1, repeat 2}; the repeats (the last two distinct offsets) are probed
 * byte-wise across the wave — lane b compares byte cursor+b with the byte one offset back, one ballot per offset =
 * the equality bitmap of the next 64 bytes, and the match length at position cursor+k is the run of ones from bit k,
 * capped at kRepCap = 32.  The first position whose best option is not beaten by the next one (by more than 4
 * quarter bytes of gain) or the one after (by more than 11) is taken; a window without any option is skipped.
 * Per position the matchers leave   hash gain (bits 0-9, 0 = no usable candidate) | offset (bits 10-26); bit 31 stays clear
 * (kChosenBit).
 * Every chosen match is written back over the parse words of its first three positions (behind the
 * cursor: dead) as {offset, length, index, literal anchor} for the emitting wave: no masks, no ranks.
 */
struct RepState {
    uint32_t cur, anchor, nseq;
    uint32_t rep1, rep2; /* the last two distinct offsets */
    uint32_t tileSeq;    /* nseq when the parse entered the tile being parsed */
    uint32_t seg;        /* the segment (profile.segLog) the repeat offsets were collected in */
};
constexpr uint32_t kRepCap = 32u, kRepMin = 3u;
constexpr uint32_t kLenCapped = 127u; /* length field of a deferred plain parse word whose candidate hit the cap: >= 64, so the in-window chase ends on it */
constexpr uint32_t kChosenBit = 0x80000000u; /* marks a parse-word slot rewritten into a chosen-match record */

/* one byte of the block at position x: from the ring, or from HBM when `far` */
__device__ __forceinline__ uint32_t ring_byte(const Src &s, uint32_t x, bool far)
{
    if (far) return reinterpret_cast<const __attribute__((address_space(1))) uint8_t *>(s.g)[x];
    return reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(s.ring)[x & kRingMask];
}

/* equality bitmap of the 64 bytes from `cur` against the bytes `rp` back (0 when rp == 0): bit b = byte cur+b matches.
 * General form (any offset, sources beyond the ring's reach come from HBM); the parse loop below only uses it when one of
 * the repeats is farther back than the ring holds */
__device__ __forceinline__ u64 rep_bitmap(const Src &src, uint32_t cur, uint32_t rp, uint32_t n, uint32_t lane, uint32_t ringFrom = 0u)
{
    const uint32_t bpos = cur + lane;
    bool eq = false;
    /* ringFrom: the deferred parse (below) reads a ring that holds [ringFrom, ringFrom + kRing) — a source is far when it lies before that */
    if (rp != 0u && bpos < n) eq = ring_byte(src, bpos, false) == ring_byte(src, bpos - rp, rp > src.nearLimit || bpos - rp < ringFrom);
    return __ballot(eq);
}

/* The deferred repeat-aware parse (qz_item<..., DEFER>): one record per chosen match, 60 bits — position (17) | offset (17) << 17 | length (13: a match
 * never leaves its 4 KiB segment) << 34 | literals since the previous match of the SEGMENT (13) << 47 — written over the segment's own parse words in
 * device memory: record k lands on the words of the segment's positions 2k and 2k + 1, behind the cursor (a match is at least three bytes long). */
__device__ __forceinline__ u64 rep_record(uint32_t q, uint32_t off, uint32_t L, uint32_t lit)
{
    return (u64)q | ((u64)off << 17) | ((u64)L << 34) | ((u64)lit << 47);
}

/* scalar x mod kRing for x < 3 * kRing (kept in SGPRs by the parse wave) */
__device__ __forceinline__ uint32_t ring_off_s(uint32_t a)
{
    return a & kRingMask;
}

/* parse from the cursor up to `limit` (a window boundary inside the tile that starts at `base`): one window
 * evaluation per iteration, state in SGPRs.  Everything an iteration reads from LDS — the window's parse words, the 64 bytes
 * from the cursor and the bytes one repeat-1 / repeat-2 offset before them — is fetched up front in ONE round trip
 * (addresses depend on the scalars only), the run lengths come from the two ballots with a funnel shift + ffbl, the hash
 * gains were computed by the matcher waves (G in the low 10 bits of the word). */
template <bool DEF>
__device__ __forceinline__ void parse_rep_span(const qzstd_hip_profile_t &pf, const Src &src, uint32_t *pvT,
                                               uint32_t base, uint32_t limit, uint32_t n, uint32_t nh, uint32_t lane,
                                               RepState &st, const uint32_t ringFrom = 0u, u64 *recG = nullptr)
{
    /* DEF: the deferred parse of one segment by one wave (qz_item, after the tile loop): the ring holds [ringFrom, ringFrom + kRing) — the 32 KiB
     * "quarter" of the block the segment lies in — so a repeat source is near when it lies at or behind ringFrom, whatever its offset; the chosen
     * matches go out as 8-byte records (rep_record) to recG[st.nseq], st.nseq counting the matches of the segment */
    /* the span lies inside one tile, a tile inside one segment: both are fixed for the call.  No match — a repeat neither —
     * starts in the last hashBytes - 1 positions of a segment (oracle: qzo_parse_rep, startEnd): the cursor a segment leaves
     * there moves on to the next segment's first position */
    const uint32_t segEnd = seg_end(pf, base, n);
    const uint32_t startEnd = segEnd - pf.hashBytes + 1u; /* <= nh */
    const uint32_t tileLim = umin(base + kTile, startEnd), stop = umin(limit, startEnd);
    if (st.cur < base) st.cur = base;
    if (st.cur >= stop) return;
    if (pf.segLog && (base >> pf.segLog) != st.seg) { /* a new segment starts without repeat offsets */
        st.rep1 = st.rep2 = 0u;
        st.seg = base >> pf.segLog;
    }
    uint32_t rc = rdfirst(ring_off_s(st.cur)); /* ring offset of the cursor, kept incrementally */
    const __attribute__((address_space(3))) uint8_t *rb = reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(src.ring);
    while (st.cur < stop) {
        const uint32_t rem = tileLim - st.cur;
        const uint32_t W = umin(pf.repWin, rem), V = umin(W + 2u, rem);
        const uint32_t curIn = st.cur;
        uint32_t wd = 0;
        if (lane < V) wd = pvT[st.cur + lane - base]; /* the candidate of the window position: gain | offset << 10 */
        u64 M1, M2;
        if (DEF ? (st.cur - st.rep1 >= ringFrom && st.cur - st.rep2 >= ringFrom)
                : (st.rep1 <= src.nearLimit && st.rep2 <= src.nearLimit)) { /* the usual case: both sources inside the ring */
            /* 64 bytes from each of the three ring offsets: what runs over the ring's end is in the mirror (kMirror >= 64) */
            const uint32_t t1 = rc - st.rep1, t2 = rc - st.rep2; /* offsets never reach before the block */
            const uint32_t r1 = t1 & kRingMask, r2 = t2 & kRingMask;
            const uint32_t A = rb[rc + lane], B1 = rb[r1 + lane], B2 = rb[r2 + lane]; /* three byte loads, one wait */
            const u64 in = below(segEnd - st.cur); /* a repeat match never leaves its segment either */
            M1 = st.rep1 ? __ballot(A == B1) & in : 0ull;
            M2 = st.rep2 ? __ballot(A == B2) & in : 0ull;
        } else {
            M1 = rep_bitmap(src, st.cur, st.rep1, segEnd, lane, ringFrom);
            M2 = rep_bitmap(src, st.cur, st.rep2, segEnd, lane, ringFrom);
        }
        /* run of ones from bit `lane`, counted up to the cap: lanes < 18 always have 32 bits of look-ahead in the low
         * word of the shifted bitmap (funnel shift), ffbl of an all-ones word gives -1 -> the cap */
        const uint32_t x1 = __builtin_amdgcn_alignbit((uint32_t)(M1 >> 32), (uint32_t)M1, lane);
        const uint32_t x2 = __builtin_amdgcn_alignbit((uint32_t)(M2 >> 32), (uint32_t)M2, lane);
        const uint32_t rl1 = umin(first_diff_bit(~x1), kRepCap), rl2 = umin(first_diff_bit(~x2), kRepCap);
        const uint32_t rg1 = rl1 < kRepMin ? 0u : (rl1 >= kRepCap ? 1000u : 4u * rl1 + 36u);
        const uint32_t rg2 = rl2 < kRepMin ? 0u : (rl2 >= kRepCap ? 999u : 4u * rl2 + 35u);
        const uint32_t Gh = wd & 0x3FFu;
        const uint32_t G = lane < V ? umax3(Gh, rg1, rg2) : 0u;
        /* the gains one and two positions on: whole-wave DPP shifts (lane i reads lane i+1); lanes >= V hold 0 */
        const uint32_t G1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)G, 0x130, 0xF, 0xF, true);
        const uint32_t G2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)G1, 0x130, 0xF, 0xF, true);
        const bool ok = lane < W && G != 0u && !(G1 > G + 4u) && !(G2 > G + 11u);
        const u64 m = __ballot(ok);
        if (!m) { /* nothing on offer in this window */
            st.cur += W;
            rc = ring_fwd(rc, W);
            continue;
        }
        const uint32_t ks = (uint32_t)__builtin_ctzll(m);
        const uint32_t gh = rdlane(Gh, ks), g1 = rdlane(rg1, ks), g2 = rdlane(rg2, ks);
        const uint32_t q = st.cur + ks;
        uint32_t off, L, from;
        if (g1 <= gh && g2 <= gh) { /* the candidate (a repeat has to gain strictly more) */
            off = rdlane(wd, ks) >> 10;
            L = (gh - 32u + (31u - (uint32_t)__builtin_clz(off + 1u))) >> 2; /* the gain holds the length: 4 len + 32 - bits(offset) */
            from = pf.capLen;
        } else if (g2 > umax(gh, g1)) {
            off = st.rep2;
            L = rdlane(rl2, ks);
            from = kRepCap;
        } else {
            off = st.rep1;
            L = rdlane(rl1, ks);
            from = kRepCap;
        }
        if (L == from) L = extend_match_from(src, q, off, from, umin(segEnd, ((q >> pf.extLog) + 2u) << pf.extLog), lane,
                                             DEF ? q - off < ringFrom : off > src.nearLimit,
                                             DEF ? (((n + 15u) & ~15u) >> 2) - 1u : 0xFFFFFFFFu); /* (DEF: a "far" source may lie just before the match, at the buffer's end) */
        if (DEF) {
            if (lane == 0u) recG[st.nseq] = rep_record(q, off, L, q - st.anchor);
        } else if (lane == 0u) {
            /* record, branch-free: the three parse-word slots at the start of the match (all behind the new cursor,
             * L >= 3) become {chosen | offset, length | index in tile, literal anchor} for the emitting wave */
            uint32_t *r = pvT + (q - base);
            r[0] = kChosenBit | off;
            r[1] = L | ((st.nseq - st.tileSeq) << 17);
            r[2] = st.anchor;
        }
        st.nseq++;
        if (off != st.rep1) {
            st.rep2 = st.rep1;
            st.rep1 = off;
        }
        st.cur = st.anchor = q + L;
        rc = rdfirst(ring_off_s(st.cur)); /* a match may be long: recompute */
        (void)curIn;
    }
}

/* The plain (lazy) greedy parse of the deferring kernels (qz_item<..., DEFER>, levels 1-4; oracle: the loop of qzo_find_sequences_from), windows
 * [W0, W1) of the tile at `base`: lane = position, wds[w] = the parse words of window w (offset 17 | capped length 7, kLenCapped = hit the cap | lane 6 |
 * start flag in bit 31).  Three steps, so that only the chase itself is a serial chain:
 *   (a) the windows' start masks (one compare into an SGPR pair each) and length fields                                — independent vector work
 *   (b) the chase through the windows, hand-written: the starts at / behind the cursor c, the first of them, its length, the cursor behind it — NINE
 *       scalar instructions per sequence (the compiler's version of the same loop: sixteen) — until the window is left (c >= 64: by a match, or because
 *       no start is left: c = 64, L = 0); a candidate that hit the cap carries kLenCapped, ends the chase and is extended to its true, bounded end
 *       (cooperatively, both sides from device memory); a cursor behind the window (a long match) skips the chase inside the asm
 *   (c) the lanes of the chosen starts store their records {position, offset, length} at [count + rank among the chosen]   — independent again
 * st.cur = the cursor, st.cnt = the segment's matches so far, st.endA = where its last match ends. */
struct PlainParse {
    uint32_t cur, cnt, endA;
};
template <uint32_t W0, uint32_t W1>
__device__ __forceinline__ void parse_plain_windows(const qzstd_hip_profile_t &pf, const Src &src, u64 *recG, const uint32_t base, const uint32_t n,
                                                    const uint32_t lastDw, const uint32_t lane, const uint32_t (&wds)[kWin], PlainParse &st)
{
    u64 smA[kWin], chA[kWin];
    uint32_t lenA[kWin];
#pragma unroll
    for (uint32_t w = W0; w < W1; w++) {
        smA[w] = __ballot((wds[w] & kChosenBit) != 0u);
        lenA[w] = (wds[w] >> 17) & 127u;
    }
#pragma unroll
    for (uint32_t w = W0; w < W1; w++) {
        const uint32_t w0 = base + 64u * w;
        const u64 sm = smA[w];
        uint32_t c = rdfirst(st.cur - w0); /* the cursor never lies before the window (uniform: said so, for the scalar chase) */
        u64 chosen = 0ull;
        uint32_t lenF = lenA[w];
        uint32_t e = 0u, j = 0u, L = 0u; /* e: where the last match taken ends (relative to the window) */
        auto chase = [&]() {
            u64 m;
            uint32_t t;
            asm volatile("s_cmp_lt_u32 %[c], 64\n"
                         "s_cbranch_scc0 3f\n"
                         "1:\n"
                         "s_lshr_b64 %[m], %[sm], %[c]\n"
                         "s_cbranch_scc0 2f\n"
                         "s_ff1_i32_b64 %[t], %[m]\n"
                         "s_add_u32 %[j], %[c], %[t]\n"
                         "s_bitset1_b64 %[ch], %[j]\n"
                         "v_readlane_b32 %[L], %[len], %[j]\n"
                         "s_add_u32 %[c], %[j], %[L]\n"
                         "s_cmp_lt_u32 %[c], 64\n"
                         "s_cbranch_scc1 1b\n"
                         "s_mov_b32 %[e], %[c]\n"
                         "s_branch 3f\n"
                         "2:\n"
                         "s_mov_b32 %[e], %[c]\n"
                         "s_movk_i32 %[c], 64\n"
                         "s_mov_b32 %[L], 0\n"
                         "3:\n"
                         : [m] "=&s"(m), [t] "=&s"(t), [j] "+s"(j), [ch] "+s"(chosen), [L] "+s"(L), [c] "+s"(c), [e] "+s"(e)
                         : [sm] "s"(sm), [len] "v"(lenF)
                         : "scc");
        };
        chase();
        while (__builtin_expect(L == kLenCapped, 0)) {
            /* the match just taken hit the candidate cap: extend it to its true (bounded) end, then go on from there */
            const uint32_t pj = w0 + j, offj = rdlane(wds[w], j) & 0x1FFFFu;
            L = extend_match_from<true>(src, pj, offj, pf.capLen, umin(seg_end(pf, pj, n), ((pj >> pf.extLog) + 2u) << pf.extLog), lane, true, lastDw);
            if (lane == j) lenF = L;
            c = e = j + L;
            L = 0u;
            chase();
        }
        lenA[w] = lenF;
        st.cur = w0 + c;
        if (chosen) st.endA = w0 + e;
        chA[w] = chosen;
    }
#pragma unroll
    for (uint32_t w = W0; w < W1; w++) {
        const u64 chosen = chA[w];
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(chosen >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)chosen, 0u));
        if ((chosen >> lane) & 1ull) recG[st.cnt + rank] = rep_record(base + 64u * w + lane, wds[w] & 0x1FFFFu, lenA[w], 0u);
        st.cnt += (uint32_t)__popcll(chosen);
    }
}

/* One result entry: a ZSTD_Sequence with the item's mark in its fourth word — or, for result areas in pinned host memory (qzstd_hip.h:
 * QZSTD_HIP_MARK_COMPACT), the same three numbers and a 12-bit tag packed into ONE 8-byte store: half the bytes the kernel pushes over PCIe */
__device__ __forceinline__ void store_entry(uint4 *out, uint32_t idx, uint32_t off, uint32_t lit, uint32_t ml, uint32_t mark)
{
    if (mark & QZSTD_HIP_MARK_COMPACT) /* uniform: one flag per work item */
        reinterpret_cast<u64 *>(out)[idx] = QZSTD_HIP_PACK(off, lit, ml, mark & 0xFFFu);
    else
        out[idx] = make_uint4(off, lit, ml, mark);
}

/* emission of one window's chosen matches by the wave that owns the window */
template <bool REP>
__device__ __forceinline__ void emit_window(const qzstd_hip_profile_t &pf, const Src &src, const uint32_t *srec,
                                            const uint32_t *pvW, uint32_t off, uint32_t len, uint32_t w0, uint32_t rpE,
                                            uint32_t lane, uint4 *out, uint32_t seqCap, uint32_t tileSeq, uint32_t mark)
{
    bool ch;
    uint32_t prevEnd, idx;
    if (REP) { /* the parse wave left a complete record in the slots of every chosen match (parse_rep_span) */
        const uint32_t r0 = pvW[lane];
        ch = (r0 & kChosenBit) != 0u;
        if (!__ballot(ch)) return;
        const uint32_t r1 = pvW[lane + 1u];
        prevEnd = pvW[lane + 2u];
        off = r0 & 0x1FFFFu;
        len = r1 & 0x1FFFFu;
        idx = tileSeq + (r1 >> 17);
    } else {
        const uint4 rec = *reinterpret_cast<const uint4 *>(srec);
        const u64 chosen = (u64)rec.x | ((u64)rec.y << 32);
        if (!chosen) return;
        const uint32_t anchorIn = rec.z, seqBase = QZ_EMIT_TRIM ? rec.w & ~kExtFlag : rec.w;
        const bool hasExt = QZ_EMIT_TRIM ? (rdfirst(rec.w) & kExtFlag) != 0u : true; /* uniform: a scalar branch skips the (rare) extended matches */
        if (hasExt) {
            const uint32_t ext0 = rdfirst(srec[4]);
            if (ext0) {
                const uint32_t ext1 = srec[5];
                if ((ext0 >> 24) == lane) len = ext0 & 0xFFFFFFu; /* extended by the parse wave */
                if (ext1 && (ext1 >> 24) == lane) len = ext1 & 0xFFFFFFu;
            }
        }
        if (QZ_EMIT_TRIM) ch = (((lane & 32u) ? rec.y : rec.x) >> (lane & 31u)) & 1u; /* (a 64-bit shift by the lane is a quarter-rate instruction) */
        else ch = (chosen >> lane) & 1ull;
        const u64 lower = chosen & below(lane);
        const uint32_t myEnd = w0 + lane + len;
        const int jprev = lower ? 63 - __builtin_clzll(lower) : 0;
        prevEnd = __shfl(myEnd, jprev);
        if (!lower) prevEnd = anchorIn;
        idx = seqBase + __builtin_amdgcn_mbcnt_hi((uint32_t)(chosen >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)chosen, 0u)); /* chosen lanes below this one */
    }
    if (ch) {
        const uint32_t p = w0 + lane, q = p - off;
        const uint32_t lit = p - prevEnd;
        const uint32_t maxb = umin(umin(umin(pf.backExt, lit), q), pf.segLog ? (p & ((1u << pf.segLog) - 1u)) : p);
        uint32_t b = 0;
        if (maxb) {
            /* the 4 bytes before p and before q, top byte = nearest; count equal bytes from the top.  Through the ring the
             * reads simply wrap below position 0: what they find there is never counted (maxb <= q) */
            const bool far = off > src.nearLimit;
            const uint32_t rp4 = ring_back(rpE, 4u);
            const uint32_t pb = rd32_r(src, p - 4u, rp4, false);
            uint32_t qb;
            if (far) qb = q >= 4u ? rd32u(src, q - 4u, true) : rd32u(src, 0u, true) << (8u * (4u - q));
            else qb = rd32_r(src, q - 4u, ring_back(rp4, off), false);
            const uint32_t x = pb ^ qb;
            b = umin(x ? (uint32_t)__builtin_clz(x) >> 3 : 4u, maxb);
        }
        if (idx < seqCap) store_entry(out, idx, off, lit - b, len + b, mark); /* ONE store: entry and mark arrive together */
    }
}

/*
 * One workgroup = one block.  8 matcher waves (one position per thread per 512-position tile)
 * + 1 parse wave, each role with its own loop and the same cadence of two barriers per iteration:
 *
 *   interval 1 of iteration it   matchers: own bytes + ring refill loads, emit(it-2),
 *                                          phase A(it): hash, table look-up(s), near-table ds_min
 *                                parse wave: windows 0-1 of tile it-1
 *   barrier
 *   interval 2                   matchers: ring refill store, phase B(it): ds_max insert(s), near read,
 *                                          candidate lengths, lazy start flags, packed parse words
 *                                parse wave: windows 2-7 of tile it-1
 *   barrier
 *
 * HAS_LONG selects the level >= 3 variant with the second (8-byte-key) table, REP the repeat-offset aware
 * parse, CHAIN (levels >= 5) the walk along the main table's predecessor chain in device memory, TURNS (level 2 and
 * levels >= 5) the per-wave ordered table updates.
 */
/* One window (64 positions, lane = position) of ORDERED inserts into the head table of the chain levels: returns every lane's
 * predecessor entry in its slot — the nearest lower lane of the same slot in this window, else what the slot held before (0 =
 * none).  `ordered`: this device's LDS serves same-address lanes of one returning ds_max in lane order (probed); otherwise the
 * same-slot lanes are ordered with ballots.  st = slot | tag << 16, or kNone for a position that takes no part. */
__device__ __forceinline__ uint32_t chain_insert_window(uint32_t *tbl, uint32_t st, uint32_t mine, uint32_t lane, bool ordered)
{
    const bool vk = st != kNone;
    const uint32_t slotK = st & 0xFFFFu;
    if (ordered) return vk ? atomicMax(&tbl[slotK], mine) : 0u;
    uint32_t prd = 0u, fin = mine;
    if (vk) {
        uint32_t *e = &tbl[slotK];
        prd = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        atomicMax(e, mine);
        fin = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    u64 rem = __ballot(fin != mine); /* lanes that are not the newest of their slot in this window have a same-slot lane above them */
    uint32_t predLane = lane;
    while (rem) {
        const uint32_t s0 = rdlane(slotK, (uint32_t)__builtin_ctzll(rem));
        const bool in = vk && slotK == s0;
        const u64 grp = __ballot(in);
        const u64 lower = grp & below(lane);
        if (in && lower) predLane = 63u - (uint32_t)__builtin_clzll(lower);
        rem &= ~grp;
    }
    const uint32_t fromLane = (uint32_t)__shfl((int)mine, (int)predLane);
    return vk ? (predLane != lane ? fromLane : prd) : 0u;
}

/* Chain levels: the INSERTS of one tile (512 positions), by ONE wave — the parse wave, which has next to nothing to do at these
 * levels, one tile ahead of the matchers: it hashes the tile's positions itself (the bytes are in the ring long before the matchers get
 * there), updates the head table window by window in position order and leaves every position's exact predecessor in its slot
 * ((position + 1) << 14 | tag, 0 = none) in P1[0..512).  LDS operations of one wave execute in order, so the eight windows are
 * pipelined back to back and still see each other exactly as sequential inserts would.  `ordered`: this device's LDS serves the
 * lanes of one returning ds_max that hit the same address in lane order (probed, probe_lds_order) — what a lane gets back then IS its
 * predecessor; otherwise the positions of one window that share a slot are ordered with ballots.  The head table belongs to this
 * wave alone (the matchers only read P1). */
__device__ __forceinline__ void chain_insert_tile(const qzstd_hip_profile_t &pf, const Src &src, uint32_t *tbl, uint32_t *P1, uint32_t t0,
                                                  uint32_t n, uint32_t nh, uint32_t lane, bool ordered)
{
    const uint32_t segE = rdfirst(seg_end(pf, t0, n)); /* a tile lies inside one segment */
    uint32_t stv[kWin];
#pragma unroll
    for (uint32_t k = 0; k < kWin; k++) {
        const uint32_t p = t0 + 64u * k + lane;
        const uint32_t mixH = rd32u(src, p, false) * kPrime1; /* the chain levels hash four bytes */
        stv[k] = (p < nh && p + 4u <= segE) ? (__umulhi(mixH, pf.tableSize) | (((mixH >> 3) & kTagMask) << 16)) : kNone;
    }
    if (ordered) {
#pragma unroll
        for (uint32_t k = 0; k < kWin; k++) {
            const uint32_t mineK = ((t0 + 64u * k + lane + 1u) << kTagBits) | (stv[k] >> 16);
            uint32_t pred = 0u;
            if (stv[k] != kNone) pred = atomicMax(&tbl[stv[k] & 0xFFFFu], mineK);
            P1[64u * k + lane] = pred;
        }
    } else {
        /* the portable path, one window at a time (it is there for devices whose LDS does not pass the probe, not for speed: kept out
         * of the kernel's register budget) */
#pragma unroll 1
        for (uint32_t k = 0; k < kWin; k++) {
            const uint32_t st = stv[k];
            const bool vk = st != kNone;
            const uint32_t slotK = st & 0xFFFFu;
            const uint32_t mineK = ((t0 + 64u * k + lane + 1u) << kTagBits) | (st >> 16);
            uint32_t prd = 0u, fin = mineK;
            if (vk) {
                uint32_t *e = &tbl[slotK];
                prd = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                atomicMax(e, mineK);
                fin = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            /* lanes that are not the newest of their slot in this window have a same-slot lane above them */
            u64 rem = __ballot(fin != mineK);
            uint32_t predLane = lane;
            while (rem) {
                const uint32_t s0 = rdlane(slotK, (uint32_t)__builtin_ctzll(rem));
                const bool in = vk && slotK == s0;
                const u64 grp = __ballot(in);
                const u64 lower = grp & below(lane);
                if (in && lower) predLane = 63u - (uint32_t)__builtin_clzll(lower); /* the nearest lower lane of the group */
                rem &= ~grp;
            }
            const uint32_t fromLane = (uint32_t)__shfl((int)mineK, (int)predLane);
            P1[64u * k + lane] = vk ? (predLane != lane ? fromLane : prd) : 0u;
        }
    }
}

/* Chain levels, items of ONE request of the resident service: they share one device scratch.  Every item still inserts and links the
 * whole block before it (the head table is its own, in LDS; the dense first links it writes are the same values whoever writes them),
 * but it completes the four-link ENTRIES only for the item's range before its own — 4 KiB instead of up to 124 — and picks up the
 * entries of the earlier ranges from the items before it: they were handed out earlier, have less history in front of them and have
 * long said so (flags[j] = epoch: item j has stored the entries of range j - 1, written through) by the time this item is done
 * inserting.  flags == nullptr: a scratch of its own, all entries built here (the launch paths: workgroups of a launch do not start
 * in order). */
struct HistShare {
    uint32_t *flags;    /* [kSvcMaxItems] of the request's slot: item j has published the chain entries of ITS range */
    uint32_t item;      /* this item's index in the request */
    uint32_t epoch;     /* of the request */
    uint32_t spinLimit; /* bound of the waits for the items before this one */
    /* round 4: the items of a request also share their HEAD TABLES (qz_item: "the shared history") */
    uint32_t nItems;    /* items of the request */
    uint32_t *tabFlags; /* [kSvcMaxItems]: item j has published the head table of its range */
    uint32_t *linkFlags;/* [kSvcMaxItems]: item j has published the first links of its range */
    uint32_t *tabs;     /* [kSvcMaxItems][kSvcTabStride] in the request's scratch */
};
constexpr uint32_t kSvcTabStride = 5888u; /* words per published head table (the chain levels' tableSize; QZSTD_HIP_SVC_WORK_BYTES counts 32 of them) */

/* One work item (a block, or a run of whole segments of one): `blk` describes it, gsrc = the block's bytes in device memory,
 * out = the item's result region, chainB = its chain entries (CHAIN), p1B = its array of first links (CHAIN, segment items).  Returns, in the parse wave, the item's sequence
 * count including the delimiter or QZSTD_HIP_NSEQ_ERROR (every thread returns that for an item that is refused); the matcher
 * waves return 0.  Both kernels below are thin shells around it: one launch = one item per workgroup
 * (qzstd_find_sequences_kernel), or a resident worker that takes items from a queue (qzstd_service_worker). */
template <bool HAS_LONG, bool REP, bool CHAIN, bool TURNS, bool NEAR, bool DEFER = false>
__device__ __forceinline__ uint32_t qz_item(const LaunchArgs &args, const qzstd_hip_block_t &blk, const uint8_t *gsrc, uint4 *out,
                                            uint4 *chainB, uint32_t *p1B, const HistShare hsh)
{
    /* DEFER (round 6, the launch kernels of the repeat-aware parse): THE PARSE IS TAKEN OUT OF THE TILE LOOP.  The repeat-aware parse is a serial
     * state machine (cursor, two repeat offsets) — in the loop, one wave walks it while eight matcher waves wait 80 % of the time (65 ms per GiB at
     * level 1 | repcodes against 12 without).  But by definition the parse of a 4 KiB SEGMENT depends on nothing before the segment (no match
     * crosses a boundary, the repeat offsets are forgotten there: what lets the resident service cut a block into 32 items), and the candidates
     * do not depend on the parse at all.  So: the tile loop only matches — every position's parse word goes to device memory (p1B: 4 B per
     * position; the chain levels' dense first-link array, which a launch only uses before the loop) — and AFTERWARDS the block is parsed 32 KiB
     * "quarter" by quarter: the quarter's bytes are staged in the ring, EIGHT waves parse its eight segments at the same time (parse_rep_span<true>,
     * a tile of parse words at a time through a private LDS window, 8-byte records back over the words), the segments' counts are summed, and every
     * wave emits its own segment's records, one lane per sequence.  Same definition (oracle: qzo_parse_rep), same sequences. */
    /* DEFER at the chain levels (A/B, QZ_CHAIN8): there the ninth wave also INSERTS for the matchers (chain_insert_tile, one tile ahead); with the parse
     * deferred, matcher wave 0 — the wave that reaches every barrier of a chain level first — takes the inserts over, and the ninth wave ends */
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    /* the wave index is wave-uniform: say so (readfirstlane), so that the role branch below is a
     * scalar branch and the parse wave's chain state lives in SGPRs instead of exec-masked VGPRs */
    const uint32_t wave = rdfirst(tid >> 6);
    const bool matcher = wave < (uint32_t)kMatchWaves;
    const uint32_t n = blk.srcLen;
    const qzstd_hip_profile_t pf = args.prof;
    const uint32_t nh = n >= pf.hashBytes ? n - pf.hashBytes + 1u : 0u; /* hashable positions */
    const uint32_t nTiles = (nh + kTile - 1u) >> kTileLog;
    /* tiles in flight between matching and emission (kParseLag): the decoupled parse wave of levels 1-4; lock-step (2) where the parse wave also
     * inserts for the matchers (CHAIN) or walks repeat offsets (REP) */
    constexpr uint32_t kLagT = (CHAIN || REP) ? 2u : kParseLag;
    constexpr bool kDecoupled = kLagT > 2u;
    static_assert(!(DEFER && kDecoupled), "the deferred parse replaces the decoupled parse wave experiment");
    /* iterations of the tile loop: the tiles, + the kLagT in which the last tiles are parsed and emitted — which a deferring kernel below the chain
     * levels does after the loop (at the chain levels a tile's flags are written one iteration later: QZ_CHAIN_SHIFT) */
    const uint32_t itEnd = (DEFER && !CHAIN) ? nTiles : nTiles + kLagT;
#ifndef QZ_TILE_SHIFT
#define QZ_TILE_SHIFT 1 /* A/B builds: 0 = lengths, start flags and parse words of a tile all inside its own second interval, as in rounds 1-5 */
#endif
    /* Round 6: with the parse wave decoupled, a tile's 32-byte extensions ("tails"), lazy start flags and parse words move to the FIRST interval of
     * the NEXT iteration: what a matcher wave does between two barriers is a chain of dependent LDS round trips (own bytes -> hash -> table read;
     * near-table read -> candidate heads -> tails -> flag permute -> parse words: ~4 + ~6.5 per iteration before), and the wave's time between
     * barriers IS that chain (r06_level1_wave_timing_before.txt: interval 2 takes 2 300 cycles on an otherwise idle SIMD for ~230 vector
     * instructions).  Shifted, the tails' source bytes are requested together with the next tile's own bytes (one round trip instead of three), the
     * position's own side of the tails comes from that same request one iteration earlier (13 dwords instead of 5: no request at all), and the flag
     * permute overlaps the table read.  Same values, same order of table operations: bit-exact. */
    constexpr bool kShift = kDecoupled && QZ_TILE_SHIFT != 0;
    /* (offset, length) of the own position in earlier tiles, kept for the emission: [i] = of tile it - 1 - i once the iteration has shifted them — at the
     * end of interval 2, or (kShift) in interval 1 after the emission, which then finds its tile one place earlier */
    constexpr uint32_t kEmitIdx = kShift ? kLagT - 2u : kLagT - 1u;
    /* what the parse wave and the late emission read from the ring has to be there still: sources up to kNear back of a tile kLagT tiles behind the matchers */
    static_assert(kNear + kLagT * kTile + 4u <= kRing - kLook, "the ring no longer holds a near source when its tile is parsed / emitted");
    /* below the chain levels the tables are powers of two (csrc/qzstd_profile.c; the launcher refuses anything else): slot = mix >> shift */
    const uint32_t tabShift = (uint32_t)__builtin_clz(pf.tableSize) + 1u, longShift = pf.longSize ? (uint32_t)__builtin_clz(pf.longSize) + 1u : 31u;
    /* segment mode (qzstd_hip_block_t.parseFrom): tiles before the segment are only inserted into the tables */
    const uint32_t firstTile = blk.parseFrom >> kTileLog;
    if (blk.parseFrom != 0u && (pf.segLog == 0u || (blk.parseFrom & ((1u << pf.segLog) - 1u)) != 0u || blk.parseFrom >= n)) {
        return QZSTD_HIP_NSEQ_ERROR; /* not a segment boundary of this level: refused (uniform: before the first barrier) */
    }
    if (NEAR && n > kRing) return QZSTD_HIP_NSEQ_ERROR; /* a descriptor longer than the launch's maxBlockLen: refused, never compared from a ring that lost its bytes */
    if (DEFER && args.pwWords != 0u && ((n + kTile - 1u) & ~(kTile - 1u)) > args.pwWords) return QZSTD_HIP_NSEQ_ERROR; /* ... nor parsed out of a scratch region it does not fit (uniform: before the first barrier) */

    /* ---- LDS layout (qzstd_hip_lds_bytes(): 72 560 B at levels 1-2, 65 392 B at levels 5-12 = two workgroups per CU; 138 096 B at levels 3-4) ---- */
    /* The workgroup's LDS is addressed from an integer constant, not from the `smem` symbol: the dynamic allocation starts at
     * LDS address 0 (the kernel has no static LDS), but the compiler resolves the symbol too late to fold it and every LDS
     * address would carry a dead `v_add 0`.  kLdsBase (16: never the null pointer) is part of qzstd_hip_lds_bytes(). */
    uint8_t *smemI = (uint8_t *)(__attribute__((address_space(3))) uint8_t *)kLdsBase;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem != 0u) {
        /* the dynamic allocation does not start at LDS address 0 after all: refuse loudly (every block an error, the host
         * falls back) rather than touch memory that is not ours */
        return QZSTD_HIP_NSEQ_ERROR; /* uniform: before the first barrier */
    }
    uint32_t *ring32 = reinterpret_cast<uint32_t *>(smemI);
    uint4 *ring128 = reinterpret_cast<uint4 *>(smemI);
    uint32_t *tbl = reinterpret_cast<uint32_t *>(smemI + kRing + kMirror);
    uint32_t *tblL = tbl + pf.tableSize;               /* [longSize]    8-byte-key table (levels >= 3)    */
    uint32_t *nearTab = tblL + pf.longSize;
    uint32_t *srec = nearTab + kTile;                  /* [kLagT][kWin][8]  emission records              */
    uint32_t *pv = srec + kLagT * kWin * kSrecWords;   /* [kLagT][kPvStride] per-position parse words     */
    uint32_t *turnCtr = pv + kLagT * kPvStride;        /* control: [0] whose turn it is to update the tables (TURNS), [2] a verdict, [kCtlArrive] */
    uint32_t *P1odd = turnCtr + 16u;                   /* [kTile] (CHAIN) predecessors of the odd tiles' positions (the even tiles': nearTab's words) */
    const uint4 *g128 = reinterpret_cast<const uint4 *>(gsrc);
    Src src;
    src.ring = (LdsWords)ring32;
    src.g = (HbmWords)reinterpret_cast<const uint32_t *>(gsrc);
#ifdef QZ_EXP_ALLNEAR /* timing experiment only (wrong lengths for far candidates): what the device-memory side of the compares costs a kernel */
    src.nearLimit = 0xFFFFFFFFu;
#else
    src.nearLimit = NEAR ? 0xFFFFFFFFu : kNear;
#endif
    const uint32_t nPad = (n + 15u) & ~15u; /* the caller keeps the buffer readable up to here */

    /* ---- clear the tables; segment mode below the chain levels: fast-forward over the tiles before the segment ---- */
    /* A segment item only INSERTS the positions before its segment.  Where the tables hold "the newest position of a slot"
     * and nothing else (levels 1-4: ds_max, no chains to link), the order of those inserts does not matter: all 576 threads
     * hash them straight from HBM, no ring, no barriers, and the tile loop starts at the segment's first tile — the state it
     * finds (tables, ring) is exactly what iterating over the history tiles would have left.  itBegin = that first tile. */
    const uint32_t itBegin = firstTile;
    /* One pass over [0, hi) does both jobs, 16 bytes per thread and step, the loads running two steps ahead of their use (a
     * segment item spends most of its start-up here: 120 KiB of history in front of the last item of a block):
     *   - chunks below histEnd (= parseFrom: a segment boundary, so every position before it hashes bytes before it) are
     *     HASHED and inserted — 16 positions from two coalesced 16-byte loads (a chunk never straddles a segment boundary;
     *     the bytes behind it exist: parseFrom < n);
     *   - chunks from `lo` on are what the tile loop expects in the RING at its first iteration: everything up to
     *     itBegin * kTile + kLook (one tile more when the loop starts at tile 0, whose iteration stages nothing), at most
     *     kRing bytes back. */
    const uint32_t histEnd = itBegin << kTileLog;
    const uint32_t hi = umin(nPad, histEnd + kLook + (itBegin ? 0u : kTile));
    const uint32_t lo = hi > kRing ? (hi - kRing + 15u) & ~15u : 0u;
    constexpr uint32_t kStep = (uint32_t)kThreads * 16u;
    const uint32_t hashEnd = CHAIN ? 0u : histEnd; /* the chain levels link their history in a pass of its own (below) */
    uint32_t fo = (hashEnd ? 0u : lo) + tid * 16u; /* (nothing to hash here: only what the ring needs) */
    uint4 fa0 = make_uint4(0u, 0u, 0u, 0u), fb0 = fa0, fa1 = fa0, fb1 = fa0;
    if (fo < hi) { fa0 = g128[fo >> 4]; if (fo < hashEnd) fb0 = g128[(fo >> 4) + 1u]; }
    if (fo + kStep < hi) { fa1 = g128[(fo + kStep) >> 4]; if (fo + kStep < hashEnd) fb1 = g128[((fo + kStep) >> 4) + 1u]; }
    for (uint32_t i = tid; i < pf.tableSize + pf.longSize; i += kThreads) tbl[i] = 0u; /* both tables */
    if (!(CHAIN && itBegin != 0u)) { /* (the chain levels' history pass borrows these words first) */
        for (uint32_t i = tid; i < kTile; i += kThreads) nearTab[i] = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < kLagT * kWin * kSrecWords + kLagT * kPvStride + 16u; i += kThreads) srec[i] = 0u; /* srec, pv, control */
    }
    const bool sharedHist = CHAIN && hsh.flags != nullptr && hsh.nItems > 1u && pf.tableSize <= kSvcTabStride;
    if (itBegin != 0u || sharedHist) __syncthreads(); /* the cleared tables, before the first insert */
    if (sharedHist) {
        /* Chain levels, the items of ONE request of the resident service: THE SHARED HISTORY (round 4).  Until now every item
         * inserted and linked the whole block before its own range by itself — 131 us of ordered LDS inserts in front of a 128 KiB
         * block's last item, the largest part of what an unchanged caller waits for at the chain levels.  But the head table is a
         * MAXIMUM per slot (the newest position): the table after [0, a) is the element-wise maximum of the tables of the ranges
         * before a.  So every item k handles ITS OWN range R_k = [parseFrom, n) only, all items at the same time:
         *   1  hash + ordered insert of R_k into the (empty) table: T_k, and the first links of R_k as far as they stay inside R_k;
         *   2  publish T_k (written through) and say so; wait for T_j of the items before it (they do the same at the same time:
         *      nobody waits for a later item, the service hands the items of a request out in order);
         *   3  table := max over j < k of T_j — the state the tile loop starts from — and the first links of R_k that found no
         *      predecessor inside R_k become that table's entries; publish them;
         *   4  complete the four-link entries of R_k by chasing first links (its own and the earlier ranges'); publish; wait for
         *      the entries of the ranges before it; one agent-scope acquire; the tile loop starts at the item's first tile.
         * The work in front of an item no longer grows with its position in the block: 4 KiB of inserts + k table reads instead of
         * up to 124 KiB of inserts.  Bit-exact with the tile loop's own result (tests/test_gpu_service.py, tests/stress). */
        constexpr uint32_t kGroup = 4096u;
        uint32_t *hist = ring32;          /* [kGroup] slot | tag */
        uint32_t *link = ring32 + kGroup; /* [kGroup] first links inside the range */
        const uint32_t a0 = blk.parseFrom, b0 = n, k = hsh.item;
        auto wait_for = [&](uint32_t *fl) -> bool { /* items 0 .. k - 1 have raised fl[j]; called by every thread, decided by wave 0 */
            if (wave == 0u) {
                uint32_t spins = 0u, ok = 1u;
                for (;;) {
                    const bool there = lane >= k || __hip_atomic_load(&fl[lane], QZ_RLX_AGENT) == hsh.epoch;
                    if (__all(there)) break;
                    if (++spins > hsh.spinLimit) { ok = 0u; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (lane == 0u) turnCtr[2] = ok;
            }
            __syncthreads();
            const bool ok = rdfirst(turnCtr[2]) != 0u;
            __syncthreads();
            return ok;
        };
        auto hash_group = [&](uint32_t g0) { /* slot | tag of the group's positions -> hist[] (kNone: takes no part) */
            if (tid < 256u) {
                const uint32_t c = g0 + tid * 16u;
                uint4 ha = make_uint4(0u, 0u, 0u, 0u), hb = ha;
                if (c < b0) { ha = g128[c >> 4]; hb = g128[(c >> 4) + 1u]; } /* (the staging buffer is readable 64 bytes past the block) */
                const uint32_t W[5] = { ha.x, ha.y, ha.z, ha.w, hb.x };
                const uint32_t segEc = seg_end(pf, c, n);
                uint32_t st16[16];
#pragma unroll
                for (uint32_t j = 0; j < 16u; j++) {
                    const uint32_t v = (j & 3u) ? __builtin_amdgcn_alignbyte(W[(j >> 2) + 1u], W[j >> 2], j & 3u) : W[j >> 2];
                    const uint32_t mixH = v * kPrime1;
                    const bool okp = c + j < nh && c + j + 4u <= segEc;
                    st16[j] = okp ? (__umulhi(mixH, pf.tableSize) | (((mixH >> 3) & kTagMask) << 16)) : kNone;
                }
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++)
                    reinterpret_cast<uint4 *>(hist)[tid * 4u + j] = make_uint4(st16[4u * j], st16[4u * j + 1u], st16[4u * j + 2u], st16[4u * j + 3u]);
            }
        };
        /* 1: the range's own table and the links that stay inside the range */
        for (uint32_t g0 = a0; g0 < b0; g0 += kGroup) {
            hash_group(g0);
            QZ_BARRIER_LDS();
            if (wave == 2u || wave == 3u) {
                const uint32_t nW = (umin(kGroup, b0 - g0) + 63u) >> 6;
                const uint32_t par = wave & 1u;
                for (uint32_t w0 = 0; w0 < nW; w0 += 16u) {
                    uint32_t st[16], pred[16];
#pragma unroll
                    for (uint32_t j = 0; j < 16u; j++) st[j] = w0 + j < nW ? hist[64u * (w0 + j) + lane] : kNone;
#pragma unroll
                    for (uint32_t j = 0; j < 16u; j++) {
                        const uint32_t pos = g0 + 64u * (w0 + j) + lane;
                        const bool mine = st[j] != kNone && (st[j] & 1u) == par;
                        pred[j] = 0u;
                        if (mine) pred[j] = args.orderedLds != 0u ? atomicMax(&tbl[st[j] & 0xFFFFu], ((pos + 1u) << kTagBits) | (st[j] >> 16))
                                                                   : chain_insert_window(tbl, st[j], ((pos + 1u) << kTagBits) | (st[j] >> 16), lane, false);
                        else if (args.orderedLds == 0u) (void)chain_insert_window(tbl, kNone, 0u, lane, false); /* (the ballots of the portable path need every lane) */
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 16u; j++)
                        if (w0 + j < nW && (st[j] == kNone || (st[j] & 1u) == par)) link[64u * (w0 + j) + lane] = pred[j];
                }
            }
            QZ_BARRIER_LDS();
            if (tid < 512u) { /* the group's links: plain stores (read back by this workgroup in step 3) */
#pragma unroll
                for (uint32_t j = 0; j < 2u; j++)
                    if (g0 + (j * 512u + tid) * 4u < b0) reinterpret_cast<uint4 *>(p1B + g0)[j * 512u + tid] = reinterpret_cast<const uint4 *>(link)[j * 512u + tid];
            }
            QZ_BARRIER_LDS();
        }
        /* 2: publish the table */
        {
            uint32_t *mine = hsh.tabs + (size_t)k * kSvcTabStride;
            for (uint32_t i = tid; i < pf.tableSize; i += kThreads) __hip_atomic_store(&mine[i], tbl[i], QZ_RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0u) __hip_atomic_store(&hsh.tabFlags[k], hsh.epoch, QZ_RLX_AGENT);
        }
        if (!wait_for(hsh.tabFlags)) return QZSTD_HIP_NSEQ_ERROR;
        /* 3: the table before the range, and the links that leave the range */
        for (uint32_t i = tid; i < pf.tableSize; i += kThreads) {
            uint32_t m = 0u;
            for (uint32_t j = 0; j < k; j++) m = umax(m, hsh.tabs[(size_t)j * kSvcTabStride + i]);
            tbl[i] = m;
        }
        __syncthreads();
        for (uint32_t g0 = a0; g0 < b0; g0 += kGroup) {
            hash_group(g0);
            QZ_BARRIER_LDS();
            if (tid < 512u) {
#pragma unroll
                for (uint32_t j = 0; j < 8u; j++) {
                    const uint32_t i = j * 512u + tid, pos = g0 + i;
                    if (pos < b0) {
                        uint32_t l1 = p1B[pos];
                        const uint32_t st = hist[i];
                        if (l1 == 0u && st != kNone) l1 = tbl[st & 0xFFFFu];
                        __hip_atomic_store(&p1B[pos], l1, QZ_RLX_AGENT);
                    }
                }
            }
            QZ_BARRIER_LDS();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0u) __hip_atomic_store(&hsh.linkFlags[k], hsh.epoch, QZ_RLX_AGENT);
        if (!wait_for(hsh.linkFlags)) return QZSTD_HIP_NSEQ_ERROR;
        /* 4: the entries of the range */
        constexpr uint32_t kCh = 16u / kEL; /* chases in flight per thread */
        for (uint32_t p0 = a0 + tid; p0 < b0; p0 += kCh * (uint32_t)kThreads) {
            uint32_t e[kEL][kCh];
#pragma unroll
            for (uint32_t i = 0; i < kCh; i++) { const uint32_t pp = p0 + i * (uint32_t)kThreads; e[0][i] = pp < b0 ? __hip_atomic_load(&p1B[pp], QZ_RLX_AGENT) : 0u; }
#pragma unroll
            for (uint32_t d = 1; d < kEL; d++)
#pragma unroll
                for (uint32_t i = 0; i < kCh; i++) e[d][i] = e[d - 1u][i] ? __hip_atomic_load(&p1B[(e[d - 1u][i] >> kTagBits) - 1u], QZ_RLX_AGENT) : 0u;
#pragma unroll
            for (uint32_t i = 0; i < kCh; i++) {
                const uint32_t pp = p0 + i * (uint32_t)kThreads;
                if (pp < b0) {
                    u64 *w = reinterpret_cast<u64 *>(chainB + (size_t)pp * kEQ);
#pragma unroll
                    for (uint32_t d = 0; d < kEL; d += 2u) __hip_atomic_store(w + d / 2u, (u64)e[d][i] | ((u64)e[d + 1u][i] << 32), QZ_RLX_AGENT);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0u) __hip_atomic_store(&hsh.flags[k], hsh.epoch, QZ_RLX_AGENT);
        if (!wait_for(hsh.flags)) return QZSTD_HIP_NSEQ_ERROR;
        for (uint32_t i = tid; i < kTile; i += kThreads) nearTab[i] = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < kLagT * kWin * kSrecWords + kLagT * kPvStride + 16u; i += kThreads) srec[i] = 0u; /* srec, pv, control */
    } else if (CHAIN && itBegin != 0u) {
        /* Chain levels, segment item: the history [0, parseFrom) has to be INSERTED AND LINKED, exactly (every position's
         * predecessor in its slot), but not walked.  Going through the tile loop for that costs one exposed HBM round trip per tile
         * (the predecessor's entry): 5.5 us x 240 tiles in front of a block's last item.  Instead:
         *   pass 1  groups of 2048 positions: 512 threads hash four positions each straight from device memory (slot | tag into
         *           LDS), ONE wave inserts the group window by window, in position order, and stores every position's
         *           predecessor (the first link of its chain entry);
         *   pass 2  all threads complete the entries by chasing those first links three more steps (independent gathers, L2).
         * The result in the head table and in the chain scratch is what the tile loop would have left; the loop starts at the
         * item's first tile.  (Sibling items of the same block may share one scratch: they write identical values.) */
        /* pass 1, groups of 4096 positions (the ring's 32 KiB of LDS, unused until the prefill below, hold two words per position:
         * slot | tag in, first link out):
         *   hash     256 threads, 16 positions each from two coalesced 16-byte loads (the next group's are already in flight);
         *   insert   TWO waves at once, one for the even slots, one for the odd ones: the order only matters among the positions
         *            of one slot, so each walks all 64 windows in position order and inserts its own lanes (returning ds_max,
         *            SIXTEEN windows in flight: one wave with eight in flight is bound by the atomics' latency, 130 us per
         *            120 KiB measured; eight waves each owning an eighth of the slots are bound by the LDS's instruction rate,
         *            230 us: every wave reads every window);
         *   copy     the 4096 first links go out to the dense array p1B, 16 bytes per thread and store. */
        constexpr uint32_t kGroup = 4096u;
        static_assert(kGroup * 8u <= kRing, "two words per position of a group in the ring's LDS");
        uint32_t *hist = ring32;          /* [kGroup] slot | tag */
        uint32_t *link = ring32 + kGroup; /* [kGroup] first links */
        const uint32_t nGroups = (histEnd + kGroup - 1u) / kGroup;
        uint4 ha = make_uint4(0u, 0u, 0u, 0u), hb = ha;
        if (tid < 256u && tid * 16u < histEnd) { ha = g128[tid]; hb = g128[tid + 1u]; }
        for (uint32_t g = 0; g < nGroups && !QZ_ABLATED(2048u); g++) {
            const uint32_t g0 = g * kGroup;
            if (tid < 256u) { /* hash */
                const uint32_t c = g0 + tid * 16u;
                const uint32_t W[5] = { ha.x, ha.y, ha.z, ha.w, hb.x };
                const uint32_t segEc = seg_end(pf, c, n); /* a 16-byte chunk never straddles a segment boundary */
                uint32_t st16[16];
#pragma unroll
                for (uint32_t k = 0; k < 16u; k++) {
                    const uint32_t v = (k & 3u) ? __builtin_amdgcn_alignbyte(W[(k >> 2) + 1u], W[k >> 2], k & 3u) : W[k >> 2];
                    const uint32_t mixH = v * kPrime1; /* the chain levels hash four bytes */
                    const bool ok = c + k < histEnd && c + k + 4u <= segEc;
                    st16[k] = ok ? (__umulhi(mixH, pf.tableSize) | (((mixH >> 3) & kTagMask) << 16)) : kNone;
                }
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++)
                    reinterpret_cast<uint4 *>(hist)[tid * 4u + k] = make_uint4(st16[4u * k], st16[4u * k + 1u], st16[4u * k + 2u], st16[4u * k + 3u]);
                const uint32_t cn = c + kGroup; /* the next group's bytes: in flight during the inserts */
                ha = hb = make_uint4(0u, 0u, 0u, 0u);
                if (cn < histEnd) { ha = g128[cn >> 4]; hb = g128[(cn >> 4) + 1u]; } /* parseFrom < n: the bytes behind exist */
            }
            QZ_BARRIER_LDS();
            if (wave == 2u || wave == 3u) { /* insert: this wave's slots (neither wave shares its SIMD with the parse wave) */
                const uint32_t nW = umin(kGroup, histEnd - g0) >> 6; /* histEnd is a multiple of 4096: nW of 64 */
                const uint32_t par = wave & 1u;
                if (args.orderedLds != 0u) {
                    /* the returning ds_max of sixteen windows issued back to back (LDS operations of one wave execute in order) */
                    for (uint32_t w0 = 0; w0 < nW; w0 += 16u) {
                        uint32_t st[16], pred[16];
#pragma unroll
                        for (uint32_t k = 0; k < 16u; k++) st[k] = hist[64u * (w0 + k) + lane];
#pragma unroll
                        for (uint32_t k = 0; k < 16u; k++) {
                            const uint32_t pos = g0 + 64u * (w0 + k) + lane;
                            pred[k] = 0u;
                            if (st[k] != kNone && (st[k] & 1u) == par) pred[k] = atomicMax(&tbl[st[k] & 0xFFFFu], ((pos + 1u) << kTagBits) | (st[k] >> 16));
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 16u; k++)
                            if (st[k] == kNone || (st[k] & 1u) == par) link[64u * (w0 + k) + lane] = pred[k]; /* (no part: 0, from both waves) */
                    }
                } else {
                    for (uint32_t w = 0; w < nW; w++) { /* the portable path: same-slot lanes ordered with ballots, window by window */
                        const uint32_t st = hist[64u * w + lane];
                        const uint32_t pos = g0 + 64u * w + lane;
                        const bool mine = st != kNone && (st & 1u) == par;
                        const uint32_t pred = chain_insert_window(tbl, mine ? st : kNone, ((pos + 1u) << kTagBits) | (st >> 16), lane, false);
                        if (mine || st == kNone) link[64u * w + lane] = pred;
                    }
                }
            }
            QZ_BARRIER_LDS();
            if (tid < 512u && !QZ_ABLATED(1024u)) { /* copy out: coalesced 16-byte stores */
#pragma unroll
                for (uint32_t k = 0; k < 2u; k++)
                    if (g0 + (k * 512u + tid) * 4u < histEnd) reinterpret_cast<uint4 *>(p1B + g0)[k * 512u + tid] = reinterpret_cast<const uint4 *>(link)[k * 512u + tid];
            }
            QZ_BARRIER_LDS(); /* the words are rewritten by the next group's hashes */
        }
        /* pass 2: every history position's entry = its first link and the three behind it, chased through the dense array (L2:
         * about 1.8 us per dependent gather under load); eight independent chases per thread in flight (sixteen push the kernel past
         * 104 VGPRs: one workgroup per CU, level 6 26 -> 37 ms per 256 MiB) */
        constexpr uint32_t kChase = 32u / kEL; /* chases in flight per thread: kEL words each */
        /* (shared scratch: only the range of the item before this one, the rest comes from the items before it) */
        const uint32_t chaseFrom = hsh.flags ? histEnd - histEnd / hsh.item : 0u; /* (a history implies item >= 1) */
        for (uint32_t p0 = chaseFrom + tid; p0 < histEnd && !QZ_ABLATED(512u); p0 += kChase * (uint32_t)kThreads) {
            uint32_t e[kEL][kChase];
#pragma unroll
            for (uint32_t i = 0; i < kChase; i++) { const uint32_t pp = p0 + i * (uint32_t)kThreads; e[0][i] = pp < histEnd ? p1B[pp] : 0u; }
#pragma unroll
            for (uint32_t d = 1; d < kEL; d++)
#pragma unroll
                for (uint32_t i = 0; i < kChase; i++) e[d][i] = e[d - 1u][i] ? p1B[(e[d - 1u][i] >> kTagBits) - 1u] : 0u;
#pragma unroll
            for (uint32_t i = 0; i < kChase; i++) {
                const uint32_t pp = p0 + i * (uint32_t)kThreads;
                if (pp < histEnd) {
                    if (hsh.flags) { /* written through: other items (other XCDs) read these */
                        u64 *w = reinterpret_cast<u64 *>(chainB + (size_t)pp * kEQ);
#pragma unroll
                        for (uint32_t d = 0; d < kEL; d += 2u) __hip_atomic_store(w + d / 2u, (u64)e[d][i] | ((u64)e[d + 1u][i] << 32), QZ_RLX_AGENT);
                    } else {
#pragma unroll
                        for (uint32_t j = 0; j < kEL / 4u; j++) chainB[(size_t)pp * kEQ + j] = make_uint4(e[4u * j][i], e[4u * j + 1u][i], e[4u * j + 2u][i], e[4u * j + 3u][i]);
                    }
                }
            }
        }
        if (hsh.flags) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* every storing wave drains before the flag */
            __syncthreads();
            if (wave == 0u) {
                if (lane == 0u) __hip_atomic_store(&hsh.flags[hsh.item], hsh.epoch, QZ_RLX_AGENT);
                uint32_t spins = 0u, ok = 1u;
                for (;;) { /* items 1 .. item - 1 have stored the ranges 0 .. item - 2 */
                    const bool there = lane == 0u || lane >= hsh.item || __hip_atomic_load(&hsh.flags[lane], QZ_RLX_AGENT) == hsh.epoch;
                    if (__all(there)) break;
                    if (++spins > hsh.spinLimit) { ok = 0u; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (lane == 0u) turnCtr[2] = ok;
            }
            __syncthreads();
            if (rdfirst(turnCtr[2]) == 0u) return QZSTD_HIP_NSEQ_ERROR; /* (uniform; the worker counts it with the slices given up on) */
        }
        __syncthreads();
        for (uint32_t i = tid; i < kTile; i += kThreads) nearTab[i] = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < kLagT * kWin * kSrecWords + kLagT * kPvStride + 16u; i += kThreads) srec[i] = 0u; /* srec, pv, control */
    }
    {
        const uint32_t hiMaskH = pf.hashBytes >= 8 ? 0xFFFFFFFFu : ((1u << (8u * (pf.hashBytes - 4u))) - 1u);
        for (; fo < hi; fo += kStep) {
            uint4 fa2 = make_uint4(0u, 0u, 0u, 0u), fb2 = fa2;
            const uint32_t o2 = fo + 2u * kStep;
            if (o2 < hi) { fa2 = g128[o2 >> 4]; if (o2 < hashEnd) fb2 = g128[(o2 >> 4) + 1u]; }
            if (fo >= lo) {
                const uint32_t r = ring_dw(fo) << 2;
                ring128[r >> 4] = fa0;
                if (r < kMirror) ring128[(kRing + r) >> 4] = fa0;
            }
            if (fo < hashEnd) {
                /* segment mode below the chain levels: where the tables hold "the newest position of a slot" and nothing else
                 * (ds_max, no chains to link), the order of the inserts does not matter, so the history is inserted here — no
                 * ring, no barriers — and the tile loop starts at the segment's first tile: the state it finds (tables, ring)
                 * is exactly what iterating over the history tiles would have left */
                const uint32_t W[6] = { fa0.x, fa0.y, fa0.z, fa0.w, fb0.x, fb0.y };
                const uint32_t segEc = seg_end(pf, fo, n);
#pragma unroll
                for (uint32_t k = 0; k < 16u; k++) {
                    const uint32_t p = fo + k;
                    if (p + pf.hashBytes <= segEc) { /* oracle: qzo_hashable */
                        const uint32_t v = __builtin_amdgcn_alignbyte(W[(k >> 2) + 1u], W[k >> 2], k & 3u);
                        const uint32_t w = __builtin_amdgcn_alignbyte(W[(k >> 2) + 2u], W[(k >> 2) + 1u], k & 3u);
                        const uint32_t mixH = (v * kPrime1) ^ __umul24(pf.hashBytes > 4 ? w & hiMaskH : 0u, kPrime2 & 0xFFFFFFu);
                        atomicMax(&tbl[mixH >> tabShift], ((p + 1u) << kTagBits) | ((mixH >> 3) & kTagMask));
                        if (HAS_LONG && p + 8u <= segEc) {
                            const uint32_t m8 = (v * kPrime1) ^ (w * kPrime2);
                            atomicMax(&tblL[m8 >> longShift], ((p + 1u) << kTagBits) | ((m8 >> 3) & kTagMask));
                        }
                    }
                }
            }
            fa0 = fa1; fb0 = fb1; fa1 = fa2; fb1 = fb2;
        }
        if (itBegin != 0u && TURNS && !CHAIN && tid == 0u) *turnCtr = itBegin * (uint32_t)kMatchWaves; /* the turn the first tile's wave 0 waits for */
        if (itBegin == 0u) /* short blocks: zeros behind the end, as before */
            for (uint32_t o = hi + tid * 16u; o < kTile + kLook; o += kThreads * 16u) {
                ring128[o >> 4] = make_uint4(0u, 0u, 0u, 0u);
                if (o < kMirror) ring128[(kRing + o) >> 4] = make_uint4(0u, 0u, 0u, 0u);
            }
    }
    __syncthreads();

    /* DEFER, plain parse, A/B (QZ_DEFER_INLOOP): what the NINTH wave — which only keeps the barriers' count in a deferring loop — could parse of the block
     * WHILE the matchers match: a quarter tile (two windows) per iteration, between the loop's full barrier and its LDS-only one, of tiles whose words
     * are complete.  That is a quarter of the matchers' pace — eight of a 128 KiB block's 32 segments — and takes the parse after the loop from four
     * rounds to three.  Measured slower (below): the state stays here for the A/B build. */
    PlainParse nwSt;                /* the segment in progress */
    uint32_t nwSeg, nwBase, nwQ;    /* ... its index, the tile in progress, quarters of that tile done */
    bool nwBegun = false, nwHave = false; /* inside a segment; nwNxt holds the words of the tile after this one */
    uint32_t nwWds[kWin], nwNxt[kWin];
    nwSeg = blk.parseFrom >> 12;
    nwBase = nwSeg << 12;
    nwQ = 0u;
    nwSt = PlainParse{ nwBase, 0u, 0xFFFFFFFFu };
#pragma unroll
    for (uint32_t j = 0; j < kWin; j++) nwWds[j] = nwNxt[j] = 0u;
#ifndef QZ_DEFER_INLOOP
#define QZ_DEFER_INLOOP 0 /* A/B, 1 = the ninth wave parses a quarter tile per iteration during the loop (below).  Built bit-exact and measured SLOWER: level 1 10.81 -> 11.55 ms
                           * per GiB, level 3 19.3 -> 20.8, level 4 19.8 -> 21.2 (only level 2, whose matchers wait for their turns anyway, gains: 15.0 -> 14.5) —
                           * even two windows per iteration on the ninth wave delay the two matcher waves of its SIMD, and with them every barrier: what rounds
                           * 1-5 knew as the co-critical parse wave.  Not kept: 0 = the ninth wave only keeps the barriers' count, the whole parse runs after
                           * the loop on all nine waves.  profiles/r06_ab_deferred_parse.txt */
#endif
#ifndef QZ_NINTH_EXIT
#define QZ_NINTH_EXIT 1 /* in a kernel that defers its parse below the chain levels the ninth wave has nothing to do in the tile loop: it ENDS before the loop (s_endpgm: a
                         * barrier only waits for the waves that are left — ISA, S_BARRIER) instead of keeping the barriers' count; eight waves parse and emit after
                         * the loop, wave 0 closes the block.  Level 1 10.82 -> 9.63 ms per GiB (-11 %), level 2 14.9 -> 12.8, 32 KiB blocks 12.2 -> 10.5, level 3 (one
                         * workgroup per CU) unchanged: with two workgroups per CU the idle wave was the 17th and 18th of the CU — sixteen matcher waves sit four to a
                         * SIMD.  A/B: 0 = it stays and keeps the count.  profiles/r06_ab_deferred_parse.txt */
#endif
    constexpr bool kNinthExit = DEFER && QZ_NINTH_EXIT != 0 && (REP || CHAIN || QZ_DEFER_INLOOP == 0);
    constexpr bool kWave0Inserts = CHAIN && kNinthExit;
    if (kNinthExit && !matcher) __builtin_amdgcn_endpgm(); /* (after the start-up's barrier: the wave has cleared and prefilled its share) */
    if (!matcher) {
        /* ---------------- the parse wave: its own scalar loop, same barrier cadence ---------------- */
#ifndef QZ_PARSE_PRIO
#define QZ_PARSE_PRIO 3 /* the priority of the DECOUPLED parse wave (A/B builds); the lock-step one is the serial critical path: 3 */
#endif
        if (!QZ_ABLATED(32u)) __builtin_amdgcn_s_setprio(kDecoupled ? QZ_PARSE_PRIO : 3); /* win issue arbitration on its SIMD */
#ifndef QZ_PARSE_SPLIT
#define QZ_PARSE_SPLIT 3 /* windows parsed in interval 1 (the short one), the rest in interval 2 (A/B builds) */
#endif
        constexpr uint32_t kSplit = QZ_PARSE_SPLIT;
#ifdef QZ_DEBUG_DUMP
        u64 pI1 = 0, pW1 = 0, pI2 = 0, pW2 = 0, tQ = __builtin_amdgcn_s_memtime();
#define QZ_PLAP(acc) { const u64 tN = __builtin_amdgcn_s_memtime(); acc += tN - tQ; tQ = tN; }
#else
#define QZ_PLAP(acc)
#endif
        uint32_t nseqEnd, anchorEnd;
        if constexpr (kDecoupled) {
            /* Levels 1-4 (round 6): THE DECOUPLED PARSE WAVE.  Rounds 1-5 ran the parse wave in lock-step with the matchers — windows 0-2 of tile
             * it-1 in interval 1, windows 3-7 in interval 2 — so every interval took max(slowest matcher wave, the parse wave's serial chain) and the
             * round-5 measurements named exactly that as what a tile waits for.  s_barrier knows no "arrive without waiting" on gfx950, so the parse
             * wave has to execute every barrier of the matchers' loop — but WHEN is its own business as long as
             *   (a) it parses tile k only after B2 of iteration k (the tile's parse words are complete),
             *   (b) it lets B2 of iteration j go only once tile j + 1 - kLagT is parsed (iteration j + 1 emits that tile and, in its second
             *       interval, overwrites the parse words of tile j + 1 - kLagT's ring slot).
             * Between those bounds it works through the tiles window by window and, after every window, looks at a counter the matcher waves bump
             * before each of their barriers: when all eight are waiting and (b) allows, it joins the barrier at once (a hint, not a
             * synchronisation: a stale value only costs time).  With nothing to parse it simply waits at the next barrier.  The matchers thus
             * wait for the parse only when it is kLagT - 1 tiles behind, not twice per tile. */
            ParseState st = { blk.parseFrom, blk.parseFrom, 0u };
            uint32_t bar = 0u; /* barriers of the matchers' loop executed so far: B1 of iteration j = 2 (j - itBegin), B2 = that + 1 */
            const uint32_t nBar = 2u * (nTiles + kLagT - itBegin);
            const uint32_t *arriveP = turnCtr + kCtlArrive;
#ifdef QZ_DEBUG_DUMP
            u64 pYield = 0, pHeld = 0;
#endif
            for (uint32_t k = itBegin; k < nTiles && !QZ_ABLATED(1u); k++) {
                QZ_PLAP(pI1)
                /* (a): the tile's parse words are complete after B2 of its own iteration — after B1 of the next one when the flags are written there (kShift).
                 * Ahead of the matchers there is nothing to do but wait with them */
                while (bar < (kShift ? 2u * (k + 1u - itBegin) + 1u : 2u * (k - itBegin) + 2u)) { QZ_BARRIER_LDS(); bar++; }
                QZ_PLAP(pW1)
                const uint32_t *pvT = pv + (k % kLagT) * kPvStride;
                uint32_t word[kWin];
#pragma unroll
                for (uint32_t w = 0; w < kWin; w++) word[w] = pvT[64u * w + lane];
#pragma unroll
                for (uint32_t w = 0; w < kWin; w++) asm volatile("" : "+v"(word[w])); /* all eight requested here: ONE LDS wait per tile */
                ParseRecs r = { 0u, 0u, 0u, 0u, 0u, 0u };
                const uint32_t base = k << kTileLog;
                auto yield = [&](uint32_t seenV) {
                    const uint32_t seen = rdfirst(seenV);
                    if (seen >= (uint32_t)kMatchWaves * (bar + 1u)) { /* every matcher wave is waiting at barrier `bar` */
                        if (!(bar & 1u) || k + kLagT >= itBegin + (bar >> 1) + 2u) { QZ_BARRIER_LDS(); bar++;
#ifdef QZ_DEBUG_DUMP
                            pYield++;
#endif
                        }
#ifdef QZ_DEBUG_DUMP
                        else pHeld++;
#endif
                    }
                };
#ifndef QZ_HINT_EVERY
#define QZ_HINT_EVERY 1 /* windows between two looks at the matchers' counter (A/B builds) */
#endif
#define QZ_PW(W) { if (((W) + 1u) % QZ_HINT_EVERY == 0u) { \
                       const uint32_t seenV = __hip_atomic_load(arriveP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                       parse_window<W>(pf, src, word[W], base, n, lane, st, r); yield(seenV); \
                   } else parse_window<W>(pf, src, word[W], base, n, lane, st, r); }
                QZ_PW(0) QZ_PW(1) QZ_PW(2) QZ_PW(3) QZ_PW(4) QZ_PW(5) QZ_PW(6)
                parse_window<7>(pf, src, word[7], base, n, lane, st, r); /* (the records first, then the barrier — next iteration's loop head, or below) */
#undef QZ_PW
                if (lane < kWin) {
                    uint32_t *so = srec + ((k % kLagT) * kWin + lane) * kSrecWords;
                    *reinterpret_cast<uint4 *>(so) = make_uint4(r.r0, r.r1, r.r2, r.r3);
                    so[4] = r.r4;
                    so[5] = r.r5;
                }
            }
            QZ_PLAP(pI1)
            while (bar < nBar) { QZ_BARRIER_LDS(); bar++; }
            QZ_PLAP(pW1)
#ifdef QZ_DEBUG_DUMP
            pI2 = pYield; pW2 = pHeld;
#endif
            nseqEnd = st.nseq;
            anchorEnd = st.anchor;
        } else if (REP) {
            RepState st = { blk.parseFrom, blk.parseFrom, 0u, 0u, 0u, 0u, pf.segLog ? blk.parseFrom >> pf.segLog : 0u };
            for (uint32_t it = itBegin; it < itEnd; it++) {
                const bool work = it >= 1u && it - 1u < nTiles && it - 1u >= firstTile && !QZ_ABLATED(1u);
                const uint32_t k = it - 1u, base = k << kTileLog;
                uint32_t *pvT = pv + (k % kLagT) * kPvStride, *srecT = srec + (k % kLagT) * kWin * kSrecWords;
                if (work && !DEFER) {
                    st.tileSeq = st.nseq;
                    if (lane == 0u) srecT[0] = st.nseq; /* the records carry indices relative to this */
                    if (!(CHAIN && QZ_CHAIN_SHIFT)) parse_rep_span<false>(pf, src, pvT, base, base + 64u * kSplit, n, nh, lane, st);
                }
                if (CHAIN && it == itBegin && it < nTiles) chain_insert_tile(pf, src, tbl, (it & 1u) ? P1odd : nearTab, it << kTileLog, n, nh, lane, args.orderedLds != 0u);
                QZ_PLAP(pI1)
                __syncthreads(); /* B1 */
                QZ_PLAP(pW1)
                if (work && !DEFER) parse_rep_span<false>(pf, src, pvT, base, base + kTile, n, nh, lane, st);
                if (CHAIN && it + 1u < nTiles) chain_insert_tile(pf, src, tbl, ((it + 1u) & 1u) ? P1odd : nearTab, (it + 1u) << kTileLog, n, nh, lane, args.orderedLds != 0u);
                QZ_PLAP(pI2)
                if (DEFER && !CHAIN) QZ_BARRIER_LDS(); else __syncthreads(); /* B2 */
                QZ_PLAP(pW2)
            }
            nseqEnd = st.nseq;
            anchorEnd = st.anchor;
        } else {
            ParseState st = { blk.parseFrom, blk.parseFrom, 0u };
            for (uint32_t it = itBegin; it < itEnd; it++) {
                const bool work = it >= 1u && it - 1u < nTiles && it - 1u >= firstTile && !QZ_ABLATED(1u);
                const uint32_t k = it - 1u;
                if (work && !DEFER && !(CHAIN && QZ_CHAIN_SHIFT))
                    parse_tile<0, kSplit>(pf, src, pv + (k % kLagT) * kPvStride, srec + (k % kLagT) * kWin * kSrecWords,
                                          k << kTileLog, n, lane, st);
                if (CHAIN && it == itBegin && it < nTiles) chain_insert_tile(pf, src, tbl, (it & 1u) ? P1odd : nearTab, it << kTileLog, n, nh, lane, args.orderedLds != 0u);
                QZ_PLAP(pI1)
                __syncthreads(); /* B1 */
                QZ_PLAP(pW1)
                if constexpr (DEFER && !REP && !CHAIN && QZ_DEFER_INLOOP != 0) {
                    /* B1 was a full barrier: the words the matchers stored in iterations < it are complete */
                    const uint32_t nSegsP = (nh + 4095u) >> 12;
                    if (nwSeg < nSegsP && (nwBase >> kTileLog) + 1u <= it) {
                        const uint32_t tEnd = umin((nwSeg << 12) + 4096u, nTiles << kTileLog);
                        u64 *recG = reinterpret_cast<u64 *>(p1B + (nwSeg << 12));
                        const uint32_t lastDw = (nPad >> 2) - 1u;
                        if (nwQ == 0u) { /* a new tile: its words (requested a tile ago if they were complete then), and the next tile's */
                            if (nwHave) {
#pragma unroll
                                for (uint32_t j = 0; j < kWin; j++) nwWds[j] = nwNxt[j];
                            } else {
#pragma unroll
                                for (uint32_t j = 0; j < kWin; j++) nwWds[j] = p1B[nwBase + 64u * j + lane];
                            }
                            nwBegun = true;
                            const bool more = nwBase + kTile < tEnd || nwSeg + 1u < nSegsP;
                            const uint32_t nb = nwBase + kTile < tEnd ? nwBase + kTile : (nwSeg + 1u) << 12;
                            nwHave = more && (nb >> kTileLog) + 1u <= it;
                            if (nwHave) {
#pragma unroll
                                for (uint32_t j = 0; j < kWin; j++) nwNxt[j] = p1B[nb + 64u * j + lane];
                            }
                        }
                        switch (nwQ) {
                        case 0u: parse_plain_windows<0, 2>(pf, src, recG, nwBase, n, lastDw, lane, nwWds, nwSt); break;
                        case 1u: parse_plain_windows<2, 4>(pf, src, recG, nwBase, n, lastDw, lane, nwWds, nwSt); break;
                        case 2u: parse_plain_windows<4, 6>(pf, src, recG, nwBase, n, lastDw, lane, nwWds, nwSt); break;
                        default: parse_plain_windows<6, 8>(pf, src, recG, nwBase, n, lastDw, lane, nwWds, nwSt); break;
                        }
                        nwQ = (nwQ + 1u) & 3u;
                        if (nwQ == 0u) {
                            nwBase += kTile;
                            if (nwBase >= tEnd) { /* the segment is done: its count and its last match end (LDS: srec, as the parse after the loop leaves them) */
                                if (lane == 0u) { srec[nwSeg] = nwSt.cnt; srec[32u + nwSeg] = nwSt.endA; }
                                nwSeg++;
                                nwBase = nwSeg << 12;
                                nwSt = PlainParse{ nwBase, 0u, 0xFFFFFFFFu };
                                nwBegun = false;
                            }
                        }
                    }
                }
                if (work && !DEFER) {
                    if (CHAIN && QZ_CHAIN_SHIFT)
                        parse_tile<0, kWin>(pf, src, pv + (k % kLagT) * kPvStride, srec + (k % kLagT) * kWin * kSrecWords, k << kTileLog, n, lane, st);
                    else
                        parse_tile<kSplit, kWin>(pf, src, pv + (k % kLagT) * kPvStride, srec + (k % kLagT) * kWin * kSrecWords,
                                                 k << kTileLog, n, lane, st);
                }
                if (CHAIN && it + 1u < nTiles) chain_insert_tile(pf, src, tbl, ((it + 1u) & 1u) ? P1odd : nearTab, (it + 1u) << kTileLog, n, nh, lane, args.orderedLds != 0u);
                QZ_PLAP(pI2)
                if (DEFER && !CHAIN) QZ_BARRIER_LDS(); else __syncthreads(); /* B2 */
                QZ_PLAP(pW2)
            }
            nseqEnd = st.nseq;
            anchorEnd = st.anchor;
        }
#ifdef QZ_DEBUG_DUMP
        if (lane == 0 && !(blk.mark & QZSTD_HIP_MARK_COMPACT)) out[blk.seqCap - 2u] = make_uint4((uint32_t)pI1, (uint32_t)pW1, (uint32_t)pI2, (uint32_t)pW2);
        if (lane == 0 && !(blk.mark & QZSTD_HIP_MARK_COMPACT)) out[blk.seqCap - 12u - wave] = make_uint4(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)), 0u, 0u, 0u); /* HW_ID: where the wave runs */
#endif
        if constexpr (!DEFER) {
        /* delimiter {lit = tail, 0, 0}: QZSTD_decLz4s, src/qatseqprod.c:1037-1045 */
        uint32_t count = nseqEnd + 1u;
        if (lane == 0 && nseqEnd < blk.seqCap) store_entry(out, nseqEnd, 0u, n - anchorEnd, 0u, blk.mark);
        if (count >= blk.seqCap - 1u) count = QZSTD_HIP_NSEQ_ERROR; /* src/qatseqprod.c:1318 */
        return count;
        }
    }

    if (matcher) {
    /* ---------------- the 8 matcher waves ---------------- */
#ifdef QZ_MATCH_PRIO_HI /* A/B: the issue arbiter prefers the OLDER waves of a SIMD — waves 4-7 of a workgroup reach every barrier last (r06_level1_wave_timing_before.txt);
                         * a raised priority for them evens that out */
    if (wave >= 4u) __builtin_amdgcn_s_setprio(QZ_MATCH_PRIO_HI);
#endif
    /* Chain entries: of the positions before the item (its history) in chainB; of the item's own positions in ownB — the same array
     * on the launch paths, an array of its own where the items of a request share chainB (HistShare): there the entries of an
     * item's range are stored by the item AFTER it, written through, and nobody else may leave half-written lines of them in an
     * L2 that a reader on the same XCD would hit. */
    uint4 *const ownB = (CHAIN && hsh.flags) ? chainB + (size_t)QZSTD_HIP_BLOCK_MAX * kEQ : chainB;
    const uint32_t ownFrom = blk.parseFrom;
    auto entryOf = [&](uint32_t q, uint32_t (&D)[kEL]) { /* the kEL links of position q's entry */
        const uint4 *b = (q >= ownFrom ? ownB : chainB) + (size_t)q * kEQ;
#pragma unroll
        for (uint32_t j = 0; j < kEL / 4u; j++) { const uint4 v = b[j]; D[4u * j] = v.x; D[4u * j + 1u] = v.y; D[4u * j + 2u] = v.z; D[4u * j + 3u] = v.w; }
    };
    const uint32_t hiMask = pf.hashBytes >= 8 ? 0xFFFFFFFFu : ((1u << (8u * (pf.hashBytes - 4u))) - 1u);
    const uint32_t nearShift = 32u - kTileLog;
    const uint32_t stampShift = kTileLog + kTagBits;
    const uint32_t nTilesMax = QZSTD_HIP_BLOCK_MAX >> kTileLog;

    /* (offset, jump length) of this thread's position in tiles it-1 and it-2 */
    uint32_t offH[kLagT], lenH[kLagT]; /* [i] = of tile it - 1 - i */
#pragma unroll
    for (uint32_t i = 0; i < kLagT; i++) offH[i] = lenH[i] = 0u;
    uint32_t rp = ring_dw((itBegin << kTileLog) + tid) << 2 | (tid & 3u); /* ring offset of the own position, advanced by one tile per iteration */
#ifdef QZ_DEBUG_DUMP
    u64 dI1 = 0, dW1 = 0, dI2 = 0, dW2 = 0, tP = __builtin_amdgcn_s_memtime();
#define QZ_LAP(acc) { const u64 tN = __builtin_amdgcn_s_memtime(); acc += tN - tP; tP = tN; }
    /* chain walk, by phase: 0 entry build, 1 next-entry fetch issued, 2 four-byte tests, 3 heads + extensions, 4 wait for the next entry, 5 steps */
    u64 dC[6] = { 0, 0, 0, 0, 0, 0 }, tC = 0;
#define QZ_CLAP(k) { const u64 tN = __builtin_amdgcn_s_memtime(); dC[k] += tN - tC; tC = tN; }
#else
#define QZ_LAP(acc)
#define QZ_CLAP(k)
#endif

    /* Start flags and parse words of one tile from every position's candidate (cl = capped length, off = offset).  Below the chain levels this
     * runs at the end of the tile's own interval 2.  At the chain levels (QZ_CHAIN_SHIFT, round 4) it runs in interval 1 of the NEXT iteration, and
     * the parse wave — which has next to nothing to do there — parses the whole tile in that iteration's interval 2: the tile's last words no
     * longer sit between the walk and the barrier. */
    auto write_flags = [&](uint32_t tileIdx, uint32_t cl, uint32_t off) {
            /* start flags: the lazy rules compare capped lengths and never look across the window edge */
            const bool take = cl != 0u && cl >= min_len(pf, off);
            bool defer1, defer2, defer3 = false;
            if (pf.lazy >= 4u) {
                /* chain levels: by gain (4 per matched byte minus the offset's bit length, biased to stay positive);
                 * one position on must gain more than 4, two on more than 7 (oracle: qzo_is_start) */
                const uint32_t G = take ? 4u * cl + 32u - (31u - (uint32_t)__builtin_clz(off + 1u)) : 0u;
                const uint32_t G1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)G, 0x130, 0xF, 0xF, true);
                const uint32_t G2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)G1, 0x130, 0xF, 0xF, true);
                defer1 = lane < 63u && G1 > G + 4u;
                defer2 = lane < 62u && G2 > G + 7u;
            } else {
                const uint32_t tl = take ? cl : 0u; /* length if this position could start a match, else 0 */
                /* the next three positions' values: whole-wave DPP shifts (wave_shl:1 = lane i reads lane i+1), three
                 * VALU moves instead of three LDS permutes; what lane 63/62/61 read is masked by the edge rule below */
                const uint32_t tl1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl, 0x130, 0xF, 0xF, true);
                const uint32_t tl2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl1, 0x130, 0xF, 0xF, true);
                const uint32_t tl3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl2, 0x130, 0xF, 0xF, true);
                defer1 = pf.lazy >= 1u && lane < 63u && tl1 > cl;      /* next position: strictly longer */
                defer2 = pf.lazy >= 2u && lane < 62u && tl2 > cl;      /* two on: strictly longer */
                defer3 = pf.lazy >= 3u && lane < 61u && tl3 > cl + 2u; /* three on: longer by more than two */
            }
            const bool start = take && !defer1 && !defer2 && !defer3;
            if constexpr (DEFER && !REP) {
                /* the deferred plain parse (after the tile loop) works from the start flag, the capped length and the offset of every position */
                /* offset 17 | length 7 (kLenCapped: the candidate hit the cap) | start flag.  (A/B, commit 72302cb: only a start can be chosen — a window's
                 * start mask + the words of its starts only: device-memory traffic 6.97 -> 5.71 x algorithmic, kernel time +7 %: not kept) */
                p1B[(tileIdx << kTileLog) + tid] = off | ((cl == pf.capLen ? kLenCapped : cl) << 17) | (start ? kChosenBit : 0u);
                return;
            }
            const u64 startMask = __ballot(start);
            /* what the parse wave needs, one word per position (see parse_tile) */
            /* ns = the first start at/after this lane: the starts below the lane masked off word by word (a 64-bit shift by the
             * lane is a quarter-rate instruction), v_ffbl's -1 for "none" drops out of the unsigned min */
            const uint32_t smLo = (uint32_t)startMask, smHi = (uint32_t)(startMask >> 32);
            const uint32_t keepLo = lane < 32u ? smLo & (0xFFFFFFFFu << (lane & 31u)) : 0u;
            const uint32_t keepHi = lane < 32u ? smHi : smHi & (0xFFFFFFFFu << (lane & 31u));
            const uint32_t ns = umin(umin(first_diff_bit(keepLo), first_diff_bit(keepHi) | 32u), 64u);
            const bool capped = cl == pf.capLen;
            const uint32_t endj = lane + cl;
            /* the next start at/after the match end is that position's `ns`: one ds_bpermute instead of a second 64-bit shift + count */
            const uint32_t nsEnd = (uint32_t)__shfl((int)ns, (int)(endj & 63u));
            uint32_t nx = endj >= 64u ? endj : nsEnd;
            nx = capped ? kNxCapped : nx;
            /* REP: the hash gain (0 = no usable candidate; 4 len + 32 - bits(offset) < 1024) | offset << 10; bit 31 stays clear */
            const uint32_t gRep = take ? 4u * cl + 32u - (31u - (uint32_t)__builtin_clz(off + 1u)) : 0u;
            if (DEFER) p1B[(tileIdx << kTileLog) + tid] = gRep | (off << 10);
            else pv[(tileIdx % kLagT) * kPvStride + tid] = REP ? (gRep | (off << 10)) : pack_pos(nx, ns, capped ? off : cl);
    };
    /* kShift: what interval 2 of an iteration leaves for interval 1 of the next — the candidates' offsets and head lengths (0 = none), which of
     * them matched all 16 bytes of the head and go on ("need"), the cap, and the 36 bytes behind the position's head (dwords 4-12 of `own`) */
    uint32_t cOff1 = 0, cOff2 = 0, cOff3 = 0, cL1 = 0, cL2 = 0, cL3 = 0, cCap = 0;
    bool cNeed1 = false, cNeed2 = false, cNeed3 = false, cFlags = false;
    uint32_t cP[9];
#pragma unroll
    for (int i = 0; i < 9; i++) cP[i] = 0u;
    for (uint32_t it = itBegin; it < itEnd; it++) {
        const uint32_t t0 = it << kTileLog;
        const uint32_t p = t0 + tid; /* own position in tile it */
        const uint32_t stamp = (nTilesMax - 1u - (it & (nTilesMax - 1u))) << stampShift;
        uint32_t mix = 0, old = 0;
        /* a position takes part only if the bytes it hashes lie inside its segment (oracle: qzo_hashable) */
        const uint32_t segE = rdfirst(seg_end(pf, t0, n)); /* a tile lies inside one segment: uniform, kept in an SGPR */
        const bool valid = it < nTiles && p < nh && p + pf.hashBytes <= segE;
        const bool history = it < firstTile; /* uniform: a tile before the segment (segment mode): inserted, not matched */

        /* ================= interval 1 ================= */
        QZ_PRIO(2);
        if (kWave0Inserts && wave == 0u && it == itBegin && it < nTiles) /* (the ninth wave's job where it is alive) */
            chain_insert_tile(pf, src, tbl, (it & 1u) ? P1odd : nearTab, it << kTileLog, n, nh, lane, args.orderedLds != 0u);
        /* the position's own first 20 bytes (5 aligned dwords): issued first so that their LDS latency
         * hides behind the emission below; used by the hash now and by the candidate compare later */
#ifndef QZ_SHIFT_CARRY_P
#define QZ_SHIFT_CARRY_P 0 /* A/B: 1 = the position's own side of its tails is requested WITH its head (13 dwords instead of 5) and carried in registers to the next
                            * iteration (no request there; 88 VGPRs: the second workgroup no longer fits next to the first on every SIMD); 0 = requested again
                            * with the candidates' tails */
#endif
        constexpr bool kCarryP = kShift && QZ_SHIFT_CARRY_P != 0;
        constexpr int kOwn = kCarryP ? 13 : 5;
        uint32_t own[kOwn];
        load_dw_r<kOwn>(src, p, rp, false, own);
        /* kShift: the tails of tile it-1 — the 32 bytes behind the head of every candidate that matched its whole head — requested NOW, together with
         * the own bytes of tile it: one round trip for both (the candidates' ring offsets follow from the carried offsets) */
        uint32_t TQ1[9], TQ2[9], TQ3[9], TP[9];
        if constexpr (kShift) {
#pragma unroll
            for (int i = 0; i < 9; i++) asm volatile("" : "=v"(TQ1[i]), "=v"(TQ2[i]), "=v"(TQ3[i]), "=v"(TP[i])); /* only the requesting lanes ever read them: "written" without an instruction */
            const uint32_t rpP16 = ring_back(rp, kTile - 16u); /* ring offset of (the position in tile it-1) + 16 */
            if constexpr (kCarryP) {
#pragma unroll
                for (int i = 0; i < 9; i++) TP[i] = cP[i];
            } else if (cNeed1 || cNeed2 || (HAS_LONG && cNeed3)) load_dw_r<9>(src, p - kTile + 16u, rpP16, false, TP);
            if (cNeed1) load_dw_r<9>(src, p - kTile + 16u - cOff1, ring_back(rpP16, cOff1), cOff1 > src.nearLimit, TQ1);
            if (HAS_LONG && cNeed3) load_dw_r<9>(src, p - kTile + 16u - cOff3, ring_back(rpP16, cOff3), cOff3 > src.nearLimit, TQ3);
            if (cNeed2) load_dw_r<9>(src, p - kTile + 16u - cOff2, ring_back(rpP16, cOff2), false, TQ2);
        }
        /* refill: the 512 bytes that enter the look-ahead window this iteration (HBM -> registers now,
         * registers -> ring after the barrier; the ring slots they replace left everyone's reach
         * three tiles ago) */
        uint4 fresh; /* only the refilling lanes ever read it: declared "written" without an instruction (four v_mov per wave and tile otherwise) */
        asm volatile("" : "=v"(fresh.x), "=v"(fresh.y), "=v"(fresh.z), "=v"(fresh.w));
        /* done by half of wave 1: waves 0 and 4 share their SIMD with the parse wave and carry no extra chores */
        const uint32_t fpos = t0 + kLook + (tid - 64u) * 16u; /* iteration it stages [t0 + kLook, t0 + kLook + kTile) */
        const bool refill = wave == 1u && it >= 1u && lane < kTile / 16u && fpos < nPad;
        if (refill) fresh = g128[fpos >> 4];
        if (!DEFER && it >= kLagT + firstTile && !QZ_ABLATED(8u)) /* emit(it - kLagT): needs the parse of that tile (lock-step: done in interval 2 of it-1; decoupled: before the parse wave let B2 of it-1 go) */
            emit_window<REP>(pf, src, srec + ((it % kLagT) * kWin + wave) * kSrecWords, pv + (it % kLagT) * kPvStride + 64u * wave,
                             offH[kEmitIdx], lenH[kEmitIdx], t0 - kLagT * kTile + 64u * wave, ring_back(rp, kLagT * kTile), lane, out, blk.seqCap,
                             REP ? srec[(it % kLagT) * kWin * kSrecWords] : 0u, blk.mark);
        QZ_PRIO(1); /* (interval 1: the emission is behind) */
        uint32_t slot = 0, nslot = 0, slotL = 0, oldL = 0, tagL = 0;
        const bool validL = HAS_LONG && valid && p + 8u <= segE;
        uint32_t oa[4]; /* the position's first 16 bytes, byte-aligned: hashed now, compared against every candidate later */
#pragma unroll
        for (int i = 0; i < 4; i++) oa[i] = __builtin_amdgcn_alignbyte(own[i + 1], own[i], p & 3u);
        if (valid) { /* phase A(it) */
            const uint32_t v = oa[0];
            uint32_t hi = 0;
            if (pf.hashBytes > 4) hi = oa[1] & hiMask;
            /* ONE quarter-rate multiply per position below the chain levels (round 5; three before): the bytes behind the fourth come in through a
             * full-rate 24-bit product (hashBytes <= 7), the tables are powers of two (the slot is a shift) */
            mix = (v * kPrime1) ^ __umul24(hi, kPrime2 & 0xFFFFFFu);
            slot = CHAIN ? __umulhi(mix, pf.tableSize) : mix >> tabShift;
            nslot = mix >> nearShift;
            if (!TURNS) old = tbl[slot]; /* with turns the slot is read when the wave's turn comes */
            if (pf.nearTab && !history) atomicMin(&nearTab[nslot], stamp | (tid << kTagBits) | ((mix >> 3) & kTagMask));
            if (validL) { /* second table, keyed by the first 8 bytes */
                const uint32_t m8 = (v * kPrime1) ^ (oa[1] * kPrime2);
                slotL = m8 >> longShift;
                tagL = (m8 >> 3) & kTagMask;
                if (!TURNS) oldL = tblL[slotL];
            }
        }
        if constexpr (kShift) {
            /* tile it-1: lengths beyond the head, the choice between the candidates, start flags, parse words — while the table read of tile it is in flight */
            uint32_t clP = 0u, offP = 0u;
            if (cFlags) { /* uniform */
                uint32_t l1 = cL1, l2 = cL2, l3 = cL3;
                if (cNeed1 || cNeed2 || cNeed3) {
                    uint32_t pa[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) pa[i] = __builtin_amdgcn_alignbyte(TP[i + 1], TP[i], p & 3u); /* (tile it-1's position has the same alignment) */
                    if (cNeed1) l1 = 16u + tail_cmp(pa, TQ1, (p - cOff1) & 3u);
                    if (HAS_LONG && cNeed3) l3 = 16u + tail_cmp(pa, TQ3, (p - cOff3) & 3u);
                    if (cNeed2) l2 = 16u + tail_cmp(pa, TQ2, (p - cOff2) & 3u);
                }
                l1 = umin(l1, cCap); l2 = umin(l2, cCap); l3 = umin(l3, cCap);
                if (l1 >= 4u) { clP = l1; offP = cOff1; }
                if (l3 >= 4u && l3 > clP) { clP = l3; offP = cOff3; }   /* 8-byte table: only if strictly longer */
                if (l2 >= 4u && l2 >= clP) { clP = l2; offP = cOff2; }  /* same tile: ties go to the nearer source */
                if (!QZ_ABLATED(4u)) write_flags(it - 1u, clP, offP);
            }
#pragma unroll
            for (uint32_t i = kLagT - 1u; i > 0u; i--) { offH[i] = offH[i - 1u]; lenH[i] = lenH[i - 1u]; }
            offH[0] = offP;
            lenH[0] = clP;
        }
        uint32_t pre[kEL];
#pragma unroll
        for (uint32_t j = 0; j < kEL; j++) pre[j] = 0u;
        if (CHAIN && it != itBegin) {
            /* the chain entry of the predecessor (left by the parse wave during the previous tile), fetched now if that is a position of an
             * earlier tile: its entry was stored at least one barrier ago */
            old = valid ? ((it & 1u) ? P1odd : nearTab)[tid] : 0u;
            if (old != 0u && (old >> kTagBits) - 1u < t0) entryOf((old >> kTagBits) - 1u, pre);
        }
        if (CHAIN && QZ_CHAIN_SHIFT && it > itBegin && it - 1u < nTiles && it - 1u >= firstTile && !QZ_ABLATED(4u)) write_flags(it - 1u, lenH[0], offH[0]);
        if (kDecoupled && lane == 0u) (void)__hip_atomic_fetch_add(turnCtr + kCtlArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); /* "at B1": the parse wave's hint */
        QZ_LAP(dI1)
#ifdef QZ_B1_LDS_ONLY /* A/B: the barrier between the intervals orders LDS traffic only — nothing in global memory crosses it (the chain entries of the
                       * previous tile were drained at B2, the refill's load and the predecessor's entry are waited for where they are used) */
        QZ_BARRIER_LDS(); /* B1 */
#else
        __syncthreads(); /* B1 */
#endif
        QZ_LAP(dW1)

        /* ================= interval 2 ================= */
        QZ_PRIO(2);
        if (kWave0Inserts && wave == 0u && it + 1u < nTiles)
            chain_insert_tile(pf, src, tbl, ((it + 1u) & 1u) ? P1odd : nearTab, (it + 1u) << kTileLog, n, nh, lane, args.orderedLds != 0u);
#if defined(QZ_PAD_VALU) || defined(QZ_PAD_SALU) || defined(QZ_PAD_LDS)
        /* calibration builds only (make variant XFLAGS=-DQZ_PAD_VALU=64 ...): what ONE more instruction of a kind costs per matcher wave
         * and tile — the slope says which issue resource binds the kernel */
        {
            uint32_t padv = lane, pads = 1u;
#ifdef QZ_PAD_VALU
#pragma unroll
            for (int i = 0; i < QZ_PAD_VALU; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(padv) : "v"(lane));
#endif
#ifdef QZ_PAD_SALU
#pragma unroll
            for (int i = 0; i < QZ_PAD_SALU; i++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(pads) : : "scc");
#endif
#ifdef QZ_PAD_LDS
#pragma unroll
            for (int i = 0; i < QZ_PAD_LDS; i++) { uint32_t t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((rp & ~3u) + 16u)); padv ^= t; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            if (padv == 0xDEADBEEFu && pads == 0u) pv[0] = padv; /* keeps the chain alive; never true in practice */
        }
#endif
        if (refill) {
            const uint32_t o = ring_dw(fpos) << 2;
            ring128[o >> 4] = fresh;
            if (o < kMirror) ring128[(kRing + o) >> 4] = fresh;
        }
        uint32_t cl = 0, off = 0; /* capped candidate length, offset */
        if constexpr (kShift) { /* what this interval leaves for interval 1 of the next iteration (set below where the position has candidates) */
            cOff1 = cOff2 = cOff3 = cL1 = cL2 = cL3 = cCap = 0u;
            cNeed1 = cNeed2 = cNeed3 = false;
            cFlags = it < nTiles && !history;
            if constexpr (kCarryP) {
#pragma unroll
                for (int i = 0; i < 9; i++) cP[i] = own[kOwn - 9 + i];
            }
        }
        if (CHAIN) {
            /* Levels >= 5: exact hash chains (oracle: qzo_candidates_chain): every position gets its exact predecessor in
             * its slot, and walks chainDepth links from there.
             *  - INSERT.  Done by the parse wave one tile ahead (chain_insert_tile): P1T[i] = predecessor entry
             *    ((position + 1) << 14 | tag, 0 = none) of position t0 + i is there when the matchers arrive; no barrier, no wave
             *    of theirs spent on it (it used to be a serial section of wave 2 with the other seven waiting: 4 % of a tile).
             *  - CHAIN ENTRIES hold up to FOUR links (predecessor, its predecessor, ...), so a walk needs a dependent
             *    load only every fourth link.  The entry of p = {P1} + the first three links of P1's entry: prefetched in
             *    interval 1 when P1 lies in an earlier tile; hopped together from P1T when it lies in this tile (an
             *    entry may then be shorter than four — the walk simply continues from its last link).  Entries go to
             *    device memory (args.chain, 16 B per position); later tiles find them there. */
            const uint32_t *P1T = (it & 1u) ? P1odd : nearTab; /* [kTile] this tile's predecessors */
            const uint32_t tag = (mix >> 3) & kTagMask;
#ifdef QZ_DEBUG_DUMP
            tC = __builtin_amdgcn_s_memtime();
#endif
            /* the entry of the own position */
            uint32_t E[kEL];
#pragma unroll
            for (uint32_t j = 0; j < kEL; j++) E[j] = 0u;
            if (valid) {
                E[0] = P1T[tid];
                if (E[0] != 0u && (E[0] >> kTagBits) - 1u < t0) { /* predecessor in an earlier tile: its entry is here (the item's first tile: fetched now) */
                    if (it == itBegin) entryOf((E[0] >> kTagBits) - 1u, pre);
#pragma unroll
                    for (uint32_t i = 1; i < kEL; i++) E[i] = pre[i - 1u];
                } else {
#pragma unroll
                    for (uint32_t i = 1; i < kEL; i++) { /* hop inside the tile */
                        const uint32_t q = (E[i - 1u] >> kTagBits) - 1u;
                        if (E[i - 1u] == 0u || q < t0) break;
                        E[i] = P1T[q - t0];
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < kEL / 4u; j++) ownB[(size_t)p * kEQ + j] = make_uint4(E[4u * j], E[4u * j + 1u], E[4u * j + 2u], E[4u * j + 3u]);
            }
            /* the walk: chainDepth links, newest first; a link whose tag differs is a slot collision (skipped without
             * touching its bytes); the candidate with the highest gain stays, the nearer one on a tie */
            const uint32_t cap = valid ? umin(umin(pf.capLen, 48u), segE - p) : 0u; /* a match never leaves its segment; candidates are measured up to 48 bytes */
            uint32_t walked = 0;
            if (history) E[0] = 0u; /* a tile before the segment (segment mode): inserted and linked, not matched */
#if QZ_CHAIN_DEFER_EXT && QZ_CHAIN_HOIST_P
            uint32_t pa[8];
            {
                uint32_t P[9];
                load_dw_r<9>(src, p + 16u, ring_fwd(rp, 16u), false, P);
#pragma unroll
                for (int i = 0; i < 8; i++) pa[i] = __builtin_amdgcn_alignbyte(P[i + 1], P[i], p & 3u);
            }
#endif
            int bg = 0;
            QZ_CLAP(0)
#ifndef QZ_CHAIN_PROGRESS_PRIO
#define QZ_CHAIN_PROGRESS_PRIO 0 /* A/B: the walking waves' priority falls with the steps they have done (2 for the first two, 1 for the next two, then 0) */
#endif
            uint32_t stepsDone = 0u;
            if (QZ_CHAIN_PROGRESS_PRIO) __builtin_amdgcn_s_setprio(2);
            while (__ballot(E[0] != 0u)) {
                if (QZ_CHAIN_PROGRESS_PRIO) {
                    if (stepsDone == 2u * QZ_CHAIN_PROGRESS_PRIO) __builtin_amdgcn_s_setprio(1);
                    else if (stepsDone == 4u * QZ_CHAIN_PROGRESS_PRIO) __builtin_amdgcn_s_setprio(0);
                    stepsDone++;
                }
                uint32_t N[kEL];
#pragma unroll
                for (uint32_t j = 0; j < kEL; j++) N[j] = 0u;
                if (E[0] != 0u) {
                    /* the entry behind the last link of this one: in flight during the compares */
                    uint32_t last = E[0], cnt = 1u;
#pragma unroll
                    for (uint32_t j = 1; j < kEL; j++) { if (E[j] != 0u) { last = E[j]; cnt = j + 1u; } } /* (an entry's links are a prefix: zeros only behind the last) */
                    const uint32_t ql = (last >> kTagBits) - 1u;
                    if (walked + cnt < pf.chainDepth) {
                        if (ql < t0) {
                            entryOf(ql, N);
                        } else {
                            N[0] = P1T[ql - t0];
#pragma unroll
                            for (uint32_t i = 1; i < kEL; i++) {
                                const uint32_t q = (N[i - 1u] >> kTagBits) - 1u;
                                if (N[i - 1u] == 0u || q < t0) break;
                                N[i] = P1T[q - t0];
                            }
                        }
                    }
                    /* The four links of an entry TOGETHER (QZ_LINKS_PER_STEP; first two at a time: -2 %, then all four): a wave's time is the chain of its LDS (and, for far candidates,
                     * device-memory) round trips, one after the other — test, first 16 bytes, the next 32 ... — and the SIMDs are
                     * half idle while 4.5 waves each wait for theirs.  The later links of a step are tested against the best
                     * BEFORE the earlier ones (a weaker test, still a necessary condition: they survive a little more often), all
                     * tests are one round trip, all heads another; the updates follow in link order, so the result is
                     * the sequential one. */
#ifndef QZ_LINKS_PER_STEP
#define QZ_LINKS_PER_STEP 4 /* measured: 2 -> 4 another -1 to -2.6 % at levels 5-12 (76 VGPRs: still two workgroups per CU) */
#endif
                    constexpr int kG = QZ_LINKS_PER_STEP;
                    QZ_CLAP(1)
#pragma unroll
                    for (int kk = 0; kk < (int)kEL; kk += kG) {
                        uint32_t q[kG], rq[kG];
                        bool far[kG], m[kG];
#pragma unroll
                        for (int g = 0; g < kG; g++) {
                            const uint32_t l = E[kk + g];
                            q[g] = (l >> kTagBits) - 1u;
                            far[g] = p - q[g] > src.nearLimit;
                            m[g] = l != 0u && walked + (uint32_t)(kk + g) < pf.chainDepth && (l & kTagMask) == tag && (pf.window == 0u || p - q[g] <= pf.window) &&
                                   !QZ_ABLATED(2u) && !(QZ_ABLATED(64u) && far[g]); /* profiling: 64 = what the HBM-side candidates cost */
                            rq[g] = ring_back(rp, p - q[g]);
                        }
                        bool any = false;
#pragma unroll
                        for (int g = 0; g < kG; g++) any = any || m[g];
#ifdef QZ_HEADS_WITH_TESTS /* A/B: the candidates' 16-byte heads requested TOGETHER with the four-byte tests (one LDS round trip instead of two per step;
                            * the heads of candidates that fail the test are fetched for nothing) */
                        uint32_t Q[kG][5];
#pragma unroll
                        for (int g = 0; g < kG; g++) {
#pragma unroll
                            for (int i = 0; i < 5; i++) Q[g][i] = 0u;
                            if (m[g] && cl < cap) load_dw_r<5>(src, q[g], rq[g], far[g], Q[g]);
                        }
#endif
                        if (any && cl != 0u) {
                            /* links come nearest first, so a later one can only win with MORE matching bytes than the best so
                             * far (its offset costs at least as much): it has to match at byte cl, in particular.  Four bytes
                             * ending there are compared before anything else (what zstd's chain search does too); a best
                             * that already fills the cap cannot be beaten at all.  Skips most of the full compares. */
                            uint32_t v[kG];
#pragma unroll
                            for (int g = 0; g < kG; g++) {
                                m[g] = m[g] && cl < cap;
                                v[g] = 0u;
                                if (m[g]) v[g] = rd32_r(src, q[g] + cl - 3u, ring_fwd(rq[g], cl - 3u), far[g]);
                            }
                            const uint32_t pw = rd32_r(src, p + cl - 3u, ring_fwd(rp, cl - 3u), false);
#pragma unroll
                            for (int g = 0; g < kG; g++) m[g] = m[g] && v[g] == pw;
                        }
                        QZ_CLAP(2)
#ifndef QZ_HEADS_WITH_TESTS
                        uint32_t Q[kG][5];
#pragma unroll
                        for (int g = 0; g < kG; g++) {
#pragma unroll
                            for (int i = 0; i < 5; i++) Q[g][i] = 0u;
                            if (m[g]) load_dw_r<5>(src, q[g], rq[g], far[g], Q[g]);
                        }
#endif
#if QZ_CHAIN_DEFER_EXT
                        /* Round 6.  A link's LENGTH depends on nothing but (p, q): only the choice between the links is sequential.  So the step first
                         * measures the 16-byte heads of all its surviving links, then extends the heads that matched whole — in a loop in which every
                         * lane takes ITS next such link: as many passes as the busiest lane has (one or two) instead of one 60-instruction block per
                         * link position that runs whenever ANY lane of the wave needs it (four per step, practically always) —, then chooses in link
                         * order exactly as before.  A wave's time is its instruction count (profiles/r06_ab_decoupled_parse_wave.txt). */
                        uint32_t lg[kG], needX = 0u;
#pragma unroll
                        for (int g = 0; g < kG; g++) {
                            lg[g] = 0u;
                            if (m[g] && cl < cap) {
                                lg[g] = head_cmp(oa, Q[g], q[g] & 3u);
                                if (lg[g] == 16u && cap > 16u) needX |= 1u << g;
                            }
                        }
                        if (__ballot(needX != 0u)) {
#if !QZ_CHAIN_HOIST_P
                            /* the position's own 32 bytes behind the head: requested and byte-aligned once per step (they were per link) */
                            uint32_t P[9], pa[8];
                            load_dw_r<9>(src, p + 16u, ring_fwd(rp, 16u), false, P);
#pragma unroll
                            for (int i = 0; i < 8; i++) pa[i] = __builtin_amdgcn_alignbyte(P[i + 1], P[i], p & 3u);
#endif
                            do {
                                if (needX != 0u) {
                                    const uint32_t gx = first_diff_bit(needX); /* this lane's next link to extend */
                                    needX &= needX - 1u;
                                    uint32_t qx = q[0];
#pragma unroll
                                    for (int g = 1; g < kG; g++) qx = gx == (uint32_t)g ? q[g] : qx;
                                    const uint32_t t = tail_len(src, pa, qx + 16u, ring_back(ring_fwd(rp, 16u), p - qx), p - qx > src.nearLimit);
#pragma unroll
                                    for (int g = 0; g < kG; g++) lg[g] = gx == (uint32_t)g ? 16u + t : lg[g];
                                }
                            } while (__ballot(needX != 0u));
                        }
#pragma unroll
                        for (int g = 0; g < kG; g++) {
                            if (m[g] && cl < cap) { /* (a best that fills the cap cannot be beaten: the sequential walk would not have looked) */
                                const uint32_t l = umin(lg[g], cap);
                                const int gn = (int)(4u * l) - (int)(31u - (uint32_t)__builtin_clz(p - q[g] + 1u));
                                if (l >= 4u && (cl == 0u || gn > bg)) { cl = l; off = p - q[g]; bg = gn; }
                            }
                        }
#else
#pragma unroll
                        for (int g = 0; g < kG; g++) {
                            if (m[g] && cl < cap) { /* (a best that fills the cap cannot be beaten: the sequential walk would not have looked) */
                                uint32_t l = head_cmp(oa, Q[g], q[g] & 3u);
                                if (l == 16u && cap > 16u) l += chunk_len(src, p + 16u, ring_fwd(rp, 16u), p - q[g], far[g]); /* ONE step of 32: the cap is 48 at every level (round 5) */
                                l = umin(l, cap);
                                const int gn = (int)(4u * l) - (int)(31u - (uint32_t)__builtin_clz(p - q[g] + 1u));
                                if (l >= 4u && (cl == 0u || gn > bg)) { cl = l; off = p - q[g]; bg = gn; }
                            }
                        }
#endif
                    }
                    walked += cnt;
                }
                QZ_CLAP(3)
#ifdef QZ_DEBUG_DUMP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* (the next step starts with these values anyway) */
                dC[5] += 1;
#endif
                /* a best that fills the cap cannot be beaten: that position's walk ends here (round 5) — it used to ride along with nothing to do
                 * until the wave's last lane had run out of links, keeping the loop alive and its next entry in flight */
                const bool full = cl != 0u && cl >= cap;
#pragma unroll
                for (uint32_t j = 0; j < kEL; j++) E[j] = full ? 0u : N[j];
                QZ_CLAP(4)
            }
            if (QZ_CHAIN_PROGRESS_PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
        if (TURNS) {
            /* level 2 and levels >= 5 update the tables per 64 positions, in position order: the matcher waves take turns
             * (LDS counter, acquire/release at workgroup scope), each reading its slots before inserting its own
             * positions, so a position also sees the earlier waves of its tile (profile.subTileLog = 6).  The spin is
             * bounded: a lost turn would give wrong candidates, never a hung GPU. */
            const uint32_t turn = it * (uint32_t)kMatchWaves + wave;
            uint32_t spins = 0;
            while (__hip_atomic_load(turnCtr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != turn && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(1);
            if (valid) {
                old = tbl[slot];
                if (validL) oldL = tblL[slotL];
                atomicMax(&tbl[slot], ((p + 1u) << kTagBits) | ((mix >> 3) & kTagMask));
                if (validL) atomicMax(&tblL[slotL], ((p + 1u) << kTagBits) | tagL);
            }
            if (lane == 0u) __hip_atomic_store(turnCtr, turn + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (valid) {
            const uint32_t tag = (mix >> 3) & kTagMask;
            const uint32_t en = pf.nearTab ? nearTab[nslot] : 0xFFFFFFFFu;
            if (!TURNS) {
                atomicMax(&tbl[slot], ((p + 1u) << kTagBits) | tag);
                if (validL) atomicMax(&tblL[slotL], ((p + 1u) << kTagBits) | tagL);
            }
            if (!history) {
            const uint32_t cap = umin(pf.capLen, segE - p); /* a match never leaves its segment */
            /* candidate 1: newest position of earlier tiles (known since interval 1: its bytes are fetched
             * while the near-table read is still in flight); candidate 2: earliest of this tile */
            uint32_t q1 = kNone, q2 = kNone;
            if (old != 0u && (old & kTagMask) == tag) {
                const uint32_t q = (old >> kTagBits) - 1u;
                if (pf.window == 0u || p - q <= pf.window) q1 = q;
            }
            if (QZ_ABLATED(2u)) q1 = kNone;
            if (QZ_ABLATED(64u) && q1 != kNone && p - q1 > src.nearLimit) q1 = kNone; /* profiling: what the HBM-side candidates cost */
            uint32_t l1 = 0, l2 = 0, l3 = 0;
            const bool far1 = q1 != kNone && p - q1 > src.nearLimit;
            if (q1 != kNone) l1 = head_len(src, oa, q1, ring_back(rp, p - q1), far1);
            /* candidate 3 (levels >= 3): newest earlier-tile position whose first 8 bytes hash alike */
            uint32_t q3 = kNone;
            if (HAS_LONG && validL && oldL != 0u && (oldL & kTagMask) == tagL && !QZ_ABLATED(2u)) q3 = (oldL >> kTagBits) - 1u;
            const bool far3 = q3 != kNone && p - q3 > src.nearLimit;
            if (q3 != kNone) l3 = head_len(src, oa, q3, ring_back(rp, p - q3), far3);
            if (pf.nearTab && (en >> stampShift) == (stamp >> stampShift) && (en & kTagMask) == tag) {
                const uint32_t q = t0 + ((en >> kTagBits) & (kTile - 1u));
                if (q < p && !QZ_ABLATED(2u | 256u)) q2 = q;
            }
            if (q2 != kNone) l2 = head_len(src, oa, q2, ring_back(rp, p - q2), false); /* same tile: always near */
            /* survivors of the 16-byte head: 32 more bytes per step.  The first step (the only one at the tile levels'
             * cap of 48) shares the position's own side — fetched and byte-aligned once — between the candidates */
            bool need1 = l1 == 16u && cap > 16u, need2 = l2 == 16u && cap > 16u, need3 = HAS_LONG && l3 == 16u && cap > 16u;
            if (QZ_ABLATED(128u)) need1 = need2 = need3 = false; /* profiling: what the extension past 16 bytes costs */
            if (kShift) { /* the rest of this tile's lengths in interval 1 of the next iteration */
                cOff1 = q1 != kNone ? p - q1 : 0u; cOff2 = q2 != kNone ? p - q2 : 0u; cOff3 = q3 != kNone ? p - q3 : 0u;
                cL1 = l1; cL2 = l2; cL3 = l3;
                cNeed1 = need1; cNeed2 = need2; cNeed3 = need3;
                cCap = cap;
                need1 = need2 = need3 = false;
            }
            QZ_PRIO(1); /* (the heads are behind) */
            if (need1 || need2 || need3) {
                const uint32_t rp16 = ring_fwd(rp, 16u);
                uint32_t P[9], pa[8];
                load_dw_r<9>(src, p + 16u, rp16, false, P);
#pragma unroll
                for (int i = 0; i < 8; i++) pa[i] = __builtin_amdgcn_alignbyte(P[i + 1], P[i], p & 3u);
                if (need1) { const uint32_t l = tail_len(src, pa, q1 + 16u, ring_back(rp16, p - q1), far1); l1 = 16u + l; need1 = l == 32u && 48u < cap; }
                if (HAS_LONG && need3) { const uint32_t l = tail_len(src, pa, q3 + 16u, ring_back(rp16, p - q3), far3); l3 = 16u + l; need3 = l == 32u && 48u < cap; }
                if (need2) { const uint32_t l = tail_len(src, pa, q2 + 16u, ring_back(rp16, p - q2), false); l2 = 16u + l; need2 = l == 32u && 48u < cap; }
            }
            while (need1 || need2 || need3) { /* caps beyond 48: all candidates in one loop */
                const int which = need1 ? 1 : (need3 ? 3 : 2);
                const uint32_t q = which == 1 ? q1 : (which == 3 ? q3 : q2);
                const uint32_t L = which == 1 ? l1 : (which == 3 ? l3 : l2);
                const uint32_t l = chunk_len(src, p + L, ring_fwd(rp, L), p - q, which == 1 ? far1 : (which == 3 ? far3 : false));
                const bool more = l == 32u && L + 32u < cap;
                if (which == 1) { l1 = L + l; need1 = more; }
                else if (which == 3) { l3 = L + l; need3 = more; }
                else { l2 = L + l; need2 = more; }
            }
            QZ_PRIO(0); /* (the tails are behind) */
            if (!kShift) {
            l1 = umin(l1, cap); l2 = umin(l2, cap); l3 = umin(l3, cap);
            if (l1 >= 4u) { cl = l1; off = p - q1; }
            if (l3 >= 4u && l3 > cl) { cl = l3; off = p - q3; }   /* 8-byte table: only if strictly longer */
            if (l2 >= 4u && l2 >= cl) { cl = l2; off = p - q2; }  /* same tile: ties go to the nearer source */
            }
            }
        }
        }
        if (!kShift) {
        if (!(CHAIN && QZ_CHAIN_SHIFT) && it < nTiles && !history && !QZ_ABLATED(4u)) write_flags(it, cl, off);
#pragma unroll
        for (uint32_t i = kLagT - 1u; i > 0u; i--) { offH[i] = offH[i - 1u]; lenH[i] = lenH[i - 1u]; }
        offH[0] = off;
        lenH[0] = cl;
        }
        rp = ring_fwd(rp, kTile);
        if (kDecoupled && lane == 0u) (void)__hip_atomic_fetch_add(turnCtr + kCtlArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); /* "at B2" */
        QZ_LAP(dI2)
#ifdef QZ_B2_LDS_ONLY /* A/B: below the chain levels no wave reads what another wave wrote to global memory — the barrier need not wait for the result stores
                       * (to PINNED HOST memory in the product paths) either; the chain levels keep the full barrier: chain entries */
        if (CHAIN) __syncthreads(); else QZ_BARRIER_LDS(); /* B2 */
#else
        /* (DEFER below the chain levels: the parse words just stored are read after the loop — the barrier need not wait for them) */
        if (DEFER && !CHAIN) QZ_BARRIER_LDS(); else __syncthreads(); /* B2 */
#endif
        QZ_LAP(dW2)
    }
#ifdef QZ_DEBUG_DUMP
    if (!(blk.mark & QZSTD_HIP_MARK_COMPACT)) { /* (the cycle counts are dumped behind 16-byte entries only) */
    if (lane == 0) out[blk.seqCap - 3u - wave] = make_uint4((uint32_t)dI1, (uint32_t)dW1, (uint32_t)dI2, (uint32_t)dW2);
    if (lane == 0) out[blk.seqCap - 24u - 2u * wave] = make_uint4((uint32_t)(dC[0] >> 4), (uint32_t)(dC[1] >> 4), (uint32_t)(dC[2] >> 4), (uint32_t)(dC[3] >> 4));
    if (lane == 0) out[blk.seqCap - 25u - 2u * wave] = make_uint4((uint32_t)(dC[4] >> 4), (uint32_t)dC[5], 0u, 0u);
    if (lane == 0) out[blk.seqCap - 12u - wave] = make_uint4(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)), 0u, 0u, 0u); /* HW_ID */
    }
#endif
    if constexpr (!DEFER) return 0u;
    } /* matcher */

    /* ---------------- DEFER, the plain parse (levels 1-4): every matcher wave parses every eighth segment, then emits their records ---------------- */
    if constexpr (DEFER && !REP) {
        constexpr uint32_t kSegLog = 12u, kSeg = 1u << kSegLog, kMaxSegs = QZSTD_HIP_BLOCK_MAX >> kSegLog; /* the launcher only starts these kernels with profile.segLog == 12 */
        static_assert(kMaxSegs <= 64u, "one lane per segment in the prefix step");
        constexpr uint32_t kNoAnchor = 0xFFFFFFFFu;
        uint32_t *segCnt = srec, *segEndA = srec + kMaxSegs; /* [32] matches of a segment; the end of its last match (kNoAnchor: none) — srec + pv: 2 x 128 + 2 x 520 words */
        const uint32_t firstSeg = blk.parseFrom >> kSegLog, nSegs = (nh + kSeg - 1u) >> kSegLog; /* segments that hold a hashable position */
#ifdef QZ_DEBUG_DUMP
        const u64 tD0 = __builtin_amdgcn_s_memtime();
#endif
        if ((kNinthExit ? wave == 0u : !matcher) && lane == 0u) segCnt[2u * kMaxSegs] = nwSeg + (nwBegun ? 1u : 0u); /* the first segment nobody has begun (the ninth wave's own progress) */
        __syncthreads(); /* every wave's parse words are stored (the loop's last barriers may have ordered LDS only) */
#ifdef QZ_DEBUG_DUMP
        const u64 tD1 = __builtin_amdgcn_s_memtime();
#endif
        /* PASS 1.  The plain (lazy) greedy parse of a segment (oracle: the loop of qzo_find_sequences_from) depends on nothing before the segment: no match
         * crosses a boundary, so the cursor enters every segment at its first position.  Window by window — lane = position — the start flags are one
         * ballot, the chase is scalar (first start at / behind the cursor, jump by its length; a candidate that hit the cap carries kLenCapped, leaves
         * the chase and is extended to its true, bounded end), and the lanes of the chosen starts store their records {position, offset, length} at
         * once, ranked by the chosen lanes below them: record k over the words of the segment's positions 2k, 2k + 1 — behind the cursor. */
        /* (measured and not kept, bit-exact both: the segments handed out dynamically to all nine waves — commit d4651d4: level 1 the same, short blocks
         * +2 %; the words of the starts only + the windows' start masks — commit 72302cb: traffic -18 %, time +7 %) */
        {
            /* which segments are left: the ninth wave has parsed segments [firstSeg, nwSeg) during the tile loop and may stand inside segment nwSeg,
             * which it finishes itself.  S0 = the first segment nobody has begun: wave w (ALL NINE) takes S0 + w, S0 + w + 9, ... */
            const uint32_t S0 = rdfirst(segCnt[2u * kMaxSegs]);
            const uint32_t lastDw = (nPad >> 2) - 1u;
            auto load_tile = [&](uint32_t tb, uint32_t (&dst)[kWin]) {
#pragma unroll
                for (uint32_t j = 0; j < kWin; j++) dst[j] = p1B[tb + 64u * j + lane];
            };
            uint32_t nxt[kWin];
            if (!matcher && nwBegun) {
                /* the ninth wave's segment in progress: the quarter tiles left of its tile, then its other tiles */
                const uint32_t tEnd = umin((nwSeg << kSegLog) + kSeg, nTiles << kTileLog);
                u64 *recG = reinterpret_cast<u64 *>(p1B + (nwSeg << kSegLog));
                if (nwQ != 0u) {
                    if (nwQ <= 1u) parse_plain_windows<2, 4>(pf, src, recG, nwBase, n, lastDw, lane, nwWds, nwSt);
                    if (nwQ <= 2u) parse_plain_windows<4, 6>(pf, src, recG, nwBase, n, lastDw, lane, nwWds, nwSt);
                    parse_plain_windows<6, 8>(pf, src, recG, nwBase, n, lastDw, lane, nwWds, nwSt);
                    nwBase += kTile;
                }
                for (uint32_t base = nwBase; base < tEnd; base += kTile) {
                    uint32_t wds[kWin];
                    load_tile(base, wds);
                    parse_plain_windows<0, kWin>(pf, src, recG, base, n, lastDw, lane, wds, nwSt);
                }
                if (lane == 0u) { segCnt[nwSeg] = nwSt.cnt; segEndA[nwSeg] = nwSt.endA; }
            }
            constexpr uint32_t kParsers = kNinthExit ? (uint32_t)kMatchWaves : (uint32_t)kMatchWaves + 1u; /* waves that are left */
            uint32_t sgN = S0 + wave;
            if (sgN < nSegs) load_tile(sgN << kSegLog, nxt);
            while (sgN < nSegs) {
                const uint32_t sg = sgN;
                sgN = sg + kParsers;
                const uint32_t segStart = sg << kSegLog;
                u64 *recG = reinterpret_cast<u64 *>(p1B + segStart);
                const uint32_t tEnd = umin(segStart + kSeg, nTiles << kTileLog);
                PlainParse st = { segStart, 0u, kNoAnchor };
#ifndef QZ_EXP_NOPARSE /* (timing experiment only, no sequences: what the tile loop of a deferring kernel takes without its parse) */
                for (uint32_t base = segStart; base < tEnd; base += kTile) {
                    uint32_t wds[kWin];
#pragma unroll
                    for (uint32_t j = 0; j < kWin; j++) wds[j] = nxt[j];
                    /* the next tile's words — of this segment, or the first of this wave's next one: in flight while this tile is parsed */
                    const uint32_t nb = base + kTile < tEnd ? base + kTile : sgN << kSegLog;
                    if (base + kTile < tEnd || sgN < nSegs) load_tile(nb, nxt);
                    parse_plain_windows<0, kWin>(pf, src, recG, base, n, lastDw, lane, wds, st);
                }
#endif
                if (lane == 0u) { segCnt[sg] = st.cnt; segEndA[sg] = st.endA; }
            }
        }
#ifdef QZ_DEBUG_DUMP
        const u64 tD2 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads(); /* the segments' counts; every wave's records are stored */
#ifdef QZ_DEBUG_DUMP
        const u64 tD3 = __builtin_amdgcn_s_memtime();
#endif
        /* the segments' first indices and the literal anchors they start from: one lane per segment, two scans */
        const bool mine = lane >= firstSeg && lane < nSegs;
        const uint32_t cv = mine ? segCnt[lane] : 0u, ev = mine ? segEndA[lane] : kNoAnchor;
        uint32_t incl = cv, last = ev; /* inclusive: matches up to and including the lane's segment; the end of the last match up to and including it */
#pragma unroll
        for (uint32_t d = 1; d < kMaxSegs; d <<= 1) {
            const uint32_t ci = (uint32_t)__shfl_up((int)incl, d), li = (uint32_t)__shfl_up((int)last, d);
            if (lane >= d) { incl += ci; if (last == kNoAnchor) last = li; }
        }
        const uint32_t total = rdlane(incl, kMaxSegs - 1u), lastAll = rdlane(last, kMaxSegs - 1u);
        const uint32_t anchorEndAll = lastAll == kNoAnchor ? blk.parseFrom : lastAll;
        /* PASS 2: the records of a segment, one lane per sequence (the four bytes before a match and before its source come from device memory) */
        {
            for (uint32_t sg = firstSeg + wave; sg < nSegs; sg += kNinthExit ? (uint32_t)kMatchWaves : (uint32_t)kMatchWaves + 1u) { /* (any wave may emit any segment: the records were stored before the barrier) */
                const u64 *recG = reinterpret_cast<const u64 *>(p1B + (sg << kSegLog));
                const uint32_t cnt = rdlane(cv, sg), first = rdlane(incl, sg) - cnt;
                uint32_t anchorIn = blk.parseFrom; /* literals pending when the segment starts: behind the last match of any segment before it */
                if (sg > 0u) { const uint32_t a = rdlane(last, sg - 1u); if (a != kNoAnchor) anchorIn = a; }
                /* four steps of 64 records at a time, every step's loads issued before the first is used: a step is two dependent round trips to
                 * device memory (the records, then the bytes before the match and before its source) */
                constexpr uint32_t kU = 4u;
                for (uint32_t k0 = 0; k0 < cnt; k0 += 64u * kU) {
                    u64 r[kU], rp1[kU];
#pragma unroll
                    for (uint32_t u = 0; u < kU; u++) {
                        const uint32_t k = k0 + 64u * u + lane;
                        r[u] = k < cnt ? recG[k] : 0ull;
                        rp1[u] = (k < cnt && k) ? recG[k - 1u] : 0ull;
                    }
                    uint32_t P0[kU], P1[kU], Q0[kU], Q1[kU], lit[kU], maxb[kU];
#pragma unroll
                    for (uint32_t u = 0; u < kU; u++) {
                        const uint32_t k = k0 + 64u * u + lane;
                        const uint32_t pm = (uint32_t)r[u] & 0x1FFFFu, off = (uint32_t)(r[u] >> 17) & 0x1FFFFu;
                        lit[u] = k ? pm - (((uint32_t)rp1[u] & 0x1FFFFu) + ((uint32_t)(rp1[u] >> 34) & 0x1FFFu)) /* behind the end of the match before */
                                   : pm - anchorIn;
                        const uint32_t q = pm - off;
                        maxb[u] = k < cnt ? umin(umin(umin(pf.backExt, lit[u]), q), pm & (kSeg - 1u)) : 0u;
                        P0[u] = P1[u] = Q0[u] = Q1[u] = 0u;
                        if (maxb[u]) { /* as emit_window: the 4 bytes before the match and before its source, top byte = nearest; never counted beyond maxb <= q < pm */
                            const uint32_t pa = pm >= 4u ? pm - 4u : 0u, qa = q >= 4u ? q - 4u : 0u;
                            P0[u] = src.g[pa >> 2]; P1[u] = src.g[(pa >> 2) + 1u];
                            Q0[u] = src.g[qa >> 2]; Q1[u] = src.g[(qa >> 2) + 1u];
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kU; u++) {
                        const uint32_t k = k0 + 64u * u + lane;
                        if (k < cnt) {
                            const uint32_t pm = (uint32_t)r[u] & 0x1FFFFu, off = (uint32_t)(r[u] >> 17) & 0x1FFFFu, len = (uint32_t)(r[u] >> 34) & 0x1FFFu;
                            const uint32_t q = pm - off;
                            uint32_t b = 0;
                            if (maxb[u]) {
                                const uint32_t pa = pm >= 4u ? pm - 4u : 0u, qa = q >= 4u ? q - 4u : 0u;
                                uint32_t pb = __builtin_amdgcn_alignbyte(P1[u], P0[u], pa & 3u), qb = __builtin_amdgcn_alignbyte(Q1[u], Q0[u], qa & 3u);
                                if (pm < 4u) pb <<= 8u * (4u - pm);
                                if (q < 4u) qb <<= 8u * (4u - q);
                                const uint32_t x = pb ^ qb;
                                b = umin(x ? (uint32_t)__builtin_clz(x) >> 3 : 4u, maxb[u]);
                            }
                            const uint32_t idx = first + k;
                            if (idx < blk.seqCap) store_entry(out, idx, off, lit[u] - b, len + b, blk.mark);
                        }
                    }
                }
            }
#ifdef QZ_DEBUG_DUMP /* cycles: the wait for the loop's last wave, pass 1, the wait for pass 1's last wave, pass 2 */
            if (lane == 0 && !(blk.mark & QZSTD_HIP_MARK_COMPACT))
                out[blk.seqCap - 44u - wave] = make_uint4((uint32_t)(tD1 - tD0), (uint32_t)(tD2 - tD1), (uint32_t)(tD3 - tD2), (uint32_t)(__builtin_amdgcn_s_memtime() - tD3));
#endif
        }
        if (kNinthExit ? wave != 0u : matcher) return 0u;
        /* delimiter {lit = tail, 0, 0}: QZSTD_decLz4s, src/qatseqprod.c:1037-1045 */
        uint32_t count = total + 1u;
        if (lane == 0 && total < blk.seqCap) store_entry(out, total, 0u, n - anchorEndAll, 0u, blk.mark);
        if (count >= blk.seqCap - 1u) count = QZSTD_HIP_NSEQ_ERROR; /* src/qatseqprod.c:1318 */
        return count;
    }

    /* ---------------- DEFER, the repeat-aware parse: the parse and the emission, quarter by quarter (all nine waves arrive here) ---------------- */
    if constexpr (DEFER && REP) {
        constexpr uint32_t kSegLog = 12u, kSeg = 1u << kSegLog, kSegsPerQ = kRing >> kSegLog; /* the launcher only starts these kernels with profile.segLog == 12 */
        static_assert(kSegsPerQ == (uint32_t)kMatchWaves, "one wave per segment of a quarter");
        constexpr uint32_t kNoAnchor = 0xFFFFFFFFu;
        __syncthreads(); /* every wave's parse words are stored (the loop's last barriers may have ordered LDS only); ring and tables are free */
        uint32_t *pvW = tbl + wave * kPvStride; /* this wave's window of parse words: one tile (the head table's LDS: 8 x 520 words <= 5888) */
        uint32_t *ctl = srec;                   /* [0, 8) the segments' counts, [8, 16) the end of their last match (kNoAnchor: none) */
        const uint32_t firstSeg = blk.parseFrom >> kSegLog, nSegs = (nh + kSeg - 1u) >> kSegLog; /* segments that hold a hashable position */
        uint32_t total = 0u, anchorCarry = blk.parseFrom; /* sequences emitted so far; where the pending literals start */
        for (uint32_t Q = firstSeg / kSegsPerQ; Q * kSegsPerQ < nSegs; Q++) {
            const uint32_t qs = Q * kRing;
            /* the quarter's bytes into the ring: position x at x mod kRing, as in the tile loop (what a 64-byte read finds behind the ring's end
             * belongs to the next segment and is masked) */
            for (uint32_t o = qs + tid * 16u; o < umin(qs + kRing, nPad); o += (kNinthExit ? (uint32_t)kMatchThreads : (uint32_t)kThreads) * 16u) { /* (by the threads that are left) */
                const uint4 v = g128[o >> 4];
                ring128[(o & kRingMask) >> 4] = v;
                if ((o & kRingMask) < kMirror) ring128[(kRing + (o & kRingMask)) >> 4] = v; /* (the emission's four bytes before a position may wrap) */
            }
            __syncthreads();
            const uint32_t sg = Q * kSegsPerQ + wave; /* this wave's segment */
            const uint32_t segStart = sg << kSegLog;
            u64 *recG = reinterpret_cast<u64 *>(p1B + segStart);
            if (matcher) {
                uint32_t cnt = 0u, endA = kNoAnchor;
#ifndef QZ_DEFER_PRIO
#define QZ_DEFER_PRIO 0 /* A/B: the priority of the parsing waves (a serial chain each) against the other workgroup's matcher waves on their SIMDs */
#endif
                if (QZ_DEFER_PRIO) __builtin_amdgcn_s_setprio(QZ_DEFER_PRIO);
#ifdef QZ_EXP_NOPARSE /* timing experiment only (no sequences): what the tile loop of a deferring kernel takes without its parse */
                if (false) {
#else
                if (sg >= firstSeg && sg < nSegs) {
#endif
                    const uint32_t tEnd = umin(segStart + kSeg, nTiles << kTileLog);
                    uint32_t nxt[kWin];
#pragma unroll
                    for (uint32_t j = 0; j < kWin; j++) nxt[j] = p1B[segStart + 64u * j + lane];
                    {
                        RepState st = { segStart, segStart, 0u, 0u, 0u, 0u, sg };
                        for (uint32_t base = segStart; base < tEnd; base += kTile) {
#pragma unroll
                            for (uint32_t j = 0; j < kWin; j++) pvW[64u * j + lane] = nxt[j];
                            if (base + kTile < tEnd) { /* the next tile's words: in flight while this one is parsed */
#pragma unroll
                                for (uint32_t j = 0; j < kWin; j++) nxt[j] = p1B[base + kTile + 64u * j + lane];
                            }
                            parse_rep_span<true>(pf, src, pvW, base, base + kTile, n, nh, lane, st, qs, recG);
                        }
                        cnt = st.nseq;
                        if (cnt) endA = st.anchor;
                    }
                }
                if (lane == 0u) { ctl[wave] = cnt; ctl[kSegsPerQ + wave] = endA; }
                if (QZ_DEFER_PRIO) __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads(); /* counts in LDS; every wave's records are stored (it reads them back itself) */
            {
                const uint32_t cv = lane < 2u * kSegsPerQ ? ctl[lane] : 0u;
                uint32_t before = 0u, all = 0u, anchorIn = anchorCarry, anchorOut = anchorCarry;
#pragma unroll
                for (uint32_t j = 0; j < kSegsPerQ; j++) {
                    const uint32_t c = rdlane(cv, j), a = rdlane(cv, kSegsPerQ + j);
                    if (j < wave) { before += c; if (a != kNoAnchor) anchorIn = a; }
                    all += c;
                    if (a != kNoAnchor) anchorOut = a;
                }
                if (matcher) {
                    const uint32_t cnt = rdlane(cv, wave);
                    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
                        const uint32_t k = k0 + lane;
                        if (k < cnt) {
                            const u64 r = recG[k];
                            const uint32_t pm = (uint32_t)r & 0x1FFFFu, off = (uint32_t)(r >> 17) & 0x1FFFFu, len = (uint32_t)(r >> 34) & 0x1FFFu;
                            uint32_t lit = pm - anchorIn; /* the segment's first match: literals since the last match of any segment before */
                            if (k) lit = (uint32_t)(r >> 47) & 0x1FFFu;
                            const uint32_t q = pm - off;
                            const uint32_t maxb = umin(umin(umin(pf.backExt, lit), q), pm & (kSeg - 1u));
                            uint32_t b = 0;
                            if (maxb) { /* as emit_window: the 4 bytes before the match and before its source, top byte = nearest */
                                const uint32_t pb = rd32u(src, pm - 4u, false);
                                uint32_t qb;
                                if (q < qs + 4u) qb = q >= 4u ? rd32u(src, q - 4u, true) : rd32u(src, 0u, true) << (8u * (4u - q)); /* the source's bytes lie before the quarter */
                                else qb = rd32u(src, q - 4u, false);
                                const uint32_t x = pb ^ qb;
                                b = umin(x ? (uint32_t)__builtin_clz(x) >> 3 : 4u, maxb);
                            }
                            const uint32_t idx = total + before + k;
                            if (idx < blk.seqCap) store_entry(out, idx, off, lit - b, len + b, blk.mark);
                        }
                    }
                }
                total += all;
                anchorCarry = anchorOut;
            }
            __syncthreads(); /* the ring and the control words are rewritten by the next quarter */
        }
        if (kNinthExit ? wave != 0u : matcher) return 0u; /* (where the ninth wave has ended, wave 0 closes the block) */
        /* delimiter {lit = tail, 0, 0}: QZSTD_decLz4s, src/qatseqprod.c:1037-1045 */
        uint32_t count = total + 1u;
        if (lane == 0 && total < blk.seqCap) store_entry(out, total, 0u, n - anchorCarry, 0u, blk.mark);
        if (count >= blk.seqCap - 1u) count = QZSTD_HIP_NSEQ_ERROR; /* src/qatseqprod.c:1318 */
        return count;
    }
    return 0u; /* (not reached) */
}

/* one launch, one work item per workgroup (the batch paths) */
/* Two workgroups of nine waves per CU put five waves on two of the four SIMDs: the kernel must fit five waves' registers into a
 * SIMD's 512.  (A/B builds: QZ_WAVES_PER_EU=7 caps them for three workgroups per CU — measured in round 4 with a smaller head table:
 * the runtime reports three resident, 768 blocks take 1.56 x the time of 512: the third workgroup buys nothing.) */
#ifndef QZ_WAVES_PER_EU
#define QZ_WAVES_PER_EU 5
#endif
#define QZ_OCCUPANCY __attribute__((amdgpu_waves_per_eu(QZ_WAVES_PER_EU)))
/* NEAR: every block of the launch fits the ring (maxBlockLen <= kRing: BASELINE config 4's 32 KiB blocks) — nothing the ring held is
 * ever overwritten, every source is compared from LDS, and the device-memory side of every compare is compiled out */
template <bool HAS_LONG, bool REP, bool CHAIN, bool TURNS, bool NEAR>
__global__ __launch_bounds__(kThreads) QZ_OCCUPANCY void qzstd_find_sequences_kernel(LaunchArgs args)
{
    const qzstd_hip_block_t blk = args.blocks[blockIdx.x];
    /* the launch kernels parse repeat-aware levels AFTER the tile loop (qz_item: DEFER): the parse words go to the dense 4-byte array of the block's
     * scratch region — behind the chain entries at the chain levels (chainEntries), the whole region below them */
#ifndef QZ_CHAIN8
#define QZ_CHAIN8 1 /* the chain levels defer their parse too, matcher wave 0 (the wave that reaches their barriers first) inserts for the others, the ninth wave ends: eight
                     * waves per workgroup.  Bit-exact; level 6 (config 3's shape) 75.2 -> 66.8 ms per GiB, level 12 on 32 KiB web-log blocks (config 4's) 106.2 -> 90.0,
                     * level 12 on 128 KiB 186.0 -> 169.6, level 9 290.7 -> 243.2, level 5 56.1 -> 50.1.  (With the ninth wave alive the deferred parse was SLOWER at
                     * these levels: 185.8 -> 192.1 — the gain is the wave's end.)  A/B: 0 = nine waves, the parse wave inserts and parses in lock-step */
#endif
    constexpr bool kDefer = CHAIN ? (QZ_CHAIN8 != 0 || (REP && QZ_REP_DEFER > 1)) : (REP ? QZ_REP_DEFER != 0 : QZ_PLAIN_DEFER != 0);
    const uint32_t count = qz_item<HAS_LONG, REP, CHAIN, TURNS, NEAR, kDefer>(args, blk, args.src + blk.srcOff, args.seqs + blk.seqOff,
                                                                CHAIN ? args.chain + (size_t)blockIdx.x * args.chainStride : nullptr,
                                                                (CHAIN || kDefer) ? reinterpret_cast<uint32_t *>(args.chain + (size_t)blockIdx.x * args.chainStride + args.chainEntries) : nullptr,
                                                                HistShare{ nullptr, 0u, 0u, 0u, 0u, nullptr, nullptr, nullptr });
    /* The count is the host's flag when the result area is pinned host memory (announcements: host/qatseqprod.c polls the count words instead
     * of asking the runtime about the stream — a stream query waits for whatever else shares the stream's hardware queue): every wave's result
     * stores are performed, then the count with a system-scope release.  The resident service publishes its items the same way. */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr bool kNinthGone = kDefer && QZ_NINTH_EXIT != 0 && (REP || CHAIN || QZ_DEFER_INLOOP == 0);
    if (threadIdx.x == (kNinthGone ? 0u : (uint32_t)kMatchThreads)) /* lane 0 of the parse wave (of wave 0 where the ninth wave has ended: QZ_NINTH_EXIT) */
        __hip_atomic_store(args.nseq + blockIdx.x, count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* ======================================================================================================================
 * The resident service: per-block requests without a launch.
 *
 * A caller that hands over ONE block and waits (libzstd's producer contract; reference: synchronous submit + poll per block,
 * /root/reference/src/qatseqprod.c:1243-1272, many instances per device :905-928) pays, on the launch path, a copy, one or more
 * kernel launches, a stream poll — and the walk of one workgroup over the whole block.  Here the block is cut into up to 32
 * work items (runs of whole segments, profile.segLog), and the items go to workgroups that are ALREADY RESIDENT:
 *
 *   host thread     writes one 64-byte request into a ring in pinned host memory (8 self-certifying granules:
 *                   tag << 56 | payload, so the reader needs no second look), then polls the request's count words in ITS
 *                   OWN pinned memory;
 *   dispatcher      one wave (qzstd_service_dispatcher) polls the ring over PCIe — one reader, not one per CU — and expands
 *                   a request into work items in a DEVICE-memory queue (granules again, written through);
 *   workers         qzstd_service_worker: one workgroup per CU, each holds a ticket (one returning atomic) and polls ITS
 *                   OWN queue entry; an item = copy the item's slice of the block from pinned host memory into the
 *                   request's device staging buffer (write-through + flag), wait for the slices before it (they are copied
 *                   by the items before it, which were handed out earlier and wait for nothing: no deadlock), one agent-scope
 *                   acquire, then qz_item() as on the launch path; results go straight to pinned host memory, the item's
 *                   count is stored LAST with a system-scope release: the count is the completion flag.
 *
 * The kernels leave when the host asks (QZSTD_stopQatDevice, a free, a launch that needs the CUs' whole LDS) or after
 * idleUs without work (the next request launches them again); every spin is bounded.
 * ====================================================================================================================== */
constexpr uint32_t kSvcQueue = 4096u;  /* entries of the device work queue */
constexpr uint32_t kSvcRing = 256u;    /* entries of the host request ring */
constexpr uint32_t kSvcMaxItems = 32u; /* work items per request */
constexpr uint32_t kSvcSlots = 1024u;  /* request slots (one per caller in flight): slice flags */
constexpr uint32_t kSvcRejected = 0xFFFFFFFEu; /* count word: the service does not serve this request (other level): launch path */

struct SvcDev { /* device memory, zeroed before every launch of the service */
    u64 items[kSvcQueue][8];
    uint32_t head;      /* tickets handed out */
    uint32_t done;      /* items finished */
    uint32_t quit;      /* set by the dispatcher */
    uint32_t spinFails; /* items that gave up waiting for a slice */
    uint32_t started;   /* worker workgroups that have begun (diagnostics) */
    uint32_t taken;     /* items a worker has picked up (diagnostics) */
    uint32_t pad[2];
    uint32_t sliceFlag[kSvcSlots][kSvcMaxItems]; /* epoch of the request whose slice k is in the slot's staging buffer */
    uint32_t histFlag[kSvcSlots][kSvcMaxItems];  /* chain levels: epoch of the request whose item k has stored the entries of its range */
    uint32_t tabFlag[kSvcSlots][kSvcMaxItems];   /* ... has published the head table of its range (qz_item: the shared history) */
    uint32_t linkFlag[kSvcSlots][kSvcMaxItems];  /* ... has published the first links of its range */
};

struct SvcHost { /* pinned host memory */
    u64 ring[kSvcRing][8];
    uint32_t state;    /* 0 stopped, 1 running, 2 the dispatcher is deciding whether to stop */
    uint32_t quitReq;  /* the host asks the service to leave */
    u64 consumed;      /* requests taken from the ring, over all launches */
    u64 itemsDone;     /* statistics, written when the dispatcher leaves */
    uint32_t spinFails;
    uint32_t pad;
    u64 dbg[8];        /* diagnostics, refreshed by the dispatcher: polls, requests seen, items queued, workers started, items taken, items done */
};

typedef __attribute__((address_space(1))) u64 *gu64p;
typedef __attribute__((address_space(1))) uint32_t *gu32p;

__device__ __forceinline__ u64 svc_tag(u64 n, uint32_t entries) { return (n / entries) % 255ull + 1ull; }
__device__ __forceinline__ u64 svc_payload(u64 g) { return g & 0x00FFFFFFFFFFFFFFull; }

/* request granules (host -> dispatcher):
 *   0 hSrc   1 dSrc   2 hSeqs   3 hCount   (pointers, 56 bits)
 *   4 srcLen (18) | itemBytes (18) << 18 | nItems (6) << 36 | slot (10) << 42
 *   5 seqCapPerItem (24)        6 epoch (24) | level (8) << 24        7 dWork: the request's chain scratch, QZSTD_HIP_SVC_WORK_BYTES, shared by its items (chain levels)
 *   (one field per word where a word is multiplied: hipcc 7.2 folded the mask of a packed seqCap away in the dispatcher's
 *   64-bit multiply and the items' result regions landed 256 MiB apart)
 * item granules (dispatcher -> worker):
 *   0 hSrc   1 dSrc   2 the item's result region   3 the item's count word
 *   4 srcLen (18: the block up to the item's end) | parseFrom (18) << 18 | item index (6) << 36 | slot (10) << 42
 *   5 seqCap (24) | items of the request (6) << 24      6 epoch (24) | level (8) << 24      7 the request's chain scratch (chain levels; else 0) */

/* MULTI (round 4): ONE resident worker for every level whose workgroup has the 65 KB layout (levels 1-2 and 5-12, with and without the
 * repeat-aware parse): the item's granule 6 carries the request's level, the profile comes from a table in device memory and the item
 * goes to the variant of qz_item() its profile asks for — callers of several levels on one GPU are all served without a launch (before:
 * one level resident, the others through the batches, about 100 us more per call).  Levels 3-4 (136 KB per workgroup) keep a worker of
 * their own and take turns with everything else.  The four variant flags are ignored when MULTI is set. */
constexpr uint32_t kSvcProfiles = 24u; /* (level - 1) + 12 * repeat-aware */
__device__ __forceinline__ uint32_t svc_profile_index(uint32_t levelByte) { return ((levelByte & 0x7Fu) - 1u) + ((levelByte & 0x80u) ? 12u : 0u); }

template <bool HAS_LONG, bool REP, bool CHAIN, bool TURNS, bool MULTI>
__global__ __launch_bounds__(kThreads) void qzstd_service_worker(LaunchArgs args, SvcDev *sv, uint32_t ctlOff, uint32_t spinLimit, const qzstd_hip_profile_t *profs)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = rdfirst(tid >> 6);
    /* 24 words at the end of the workgroup's LDS (qzstd_hip_lds_bytes counts them): the item's granules + got / ticket */
    uint32_t *ctl = (uint32_t *)(__attribute__((address_space(3))) uint32_t *)(size_t)(kLdsBase + ctlOff);
    if (tid == 0u) (void)__hip_atomic_fetch_add(&sv->started, 1u, QZ_RLX_AGENT);
    for (;;) {
        if (wave == 0u) {
            uint32_t t = 0u;
            if (lane == 0u) t = __hip_atomic_fetch_add(&sv->head, 1u, QZ_RLX_AGENT);
            t = rdfirst(t);
            const u64 want = svc_tag(t, kSvcQueue);
            const u64 *slot = sv->items[t & (kSvcQueue - 1u)];
            u64 g = 0;
            uint32_t got = 0u;
            for (;;) { /* ends with the item or with the dispatcher's quit: the dispatcher is resident and bounded itself */
                if (lane < 8u) g = __hip_atomic_load(slot + lane, QZ_RLX_AGENT);
                else if (lane == 8u) g = (u64)__hip_atomic_load(&sv->quit, QZ_RLX_AGENT);
                const bool mine = lane >= 8u || (g >> 56) == want;
                if (__all(mine)) { got = 1u; break; }
                if (rdlane((uint32_t)g, 8u) != 0u) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane < 8u) { ctl[2u * lane] = (uint32_t)g; ctl[2u * lane + 1u] = (uint32_t)(svc_payload(g) >> 32); }
            if (lane == 8u) ctl[16] = got;
            if (lane == 0u && got) (void)__hip_atomic_fetch_add(&sv->taken, 1u, QZ_RLX_AGENT);
        }
        __syncthreads();
        if (rdfirst(ctl[16]) == 0u) return;
        /* the item's words are the same in every lane: say so (v_readfirstlane), as the launch kernel's descriptor — a kernel
         * argument load — is for the compiler: the loops and branches of qz_item() stay scalar, its parse state in SGPRs */
        const u64 q0 = rdfirst(ctl[0]) | ((u64)rdfirst(ctl[1]) << 32), q1 = rdfirst(ctl[2]) | ((u64)rdfirst(ctl[3]) << 32),
                  q2 = rdfirst(ctl[4]) | ((u64)rdfirst(ctl[5]) << 32), q3 = rdfirst(ctl[6]) | ((u64)rdfirst(ctl[7]) << 32),
                  q4 = rdfirst(ctl[8]) | ((u64)rdfirst(ctl[9]) << 32), q5 = rdfirst(ctl[10]), q6 = rdfirst(ctl[12]),
                  q7 = rdfirst(ctl[14]) | ((u64)rdfirst(ctl[15]) << 32);
        const uint8_t *hSrc = (const uint8_t *)q0;
        uint8_t *dSrc = (uint8_t *)q1;
        uint4 *out = (uint4 *)q2;
        uint32_t *countWord = (uint32_t *)q3;
        qzstd_hip_block_t blk;
        blk.srcOff = 0; blk.seqOff = 0;
        blk.srcLen = (uint32_t)q4 & 0x3FFFFu;
        blk.parseFrom = (uint32_t)(q4 >> 18) & 0x3FFFFu;
        blk.seqCap = (uint32_t)q5 & 0xFFFFFFu;
        const uint32_t nItemsReq = (uint32_t)(q5 >> 24) & 63u;
        const uint32_t k = (uint32_t)(q4 >> 36) & 63u, slotIdx = (uint32_t)(q4 >> 42) & (kSvcSlots - 1u);
        const uint32_t epoch = (uint32_t)q6 & 0xFFFFFFu;
        blk.mark = epoch; /* every entry of the item's result carries the request's epoch: see qzstd_hip_svc_req_t */
        /* ---- the item's slice: pinned host memory -> the request's device staging buffer, written through ---- */
        {
            const uint32_t end = (blk.srcLen + 15u) & ~15u;
            /* progressive staging (qzstd_hip.h): the caller queued the request before it copied the block into hSrc; the item's count word
             * says QZSTD_HIP_NSEQ_STAGING until slice k is there (usually long gone: the request took 8 us to get here, a slice 0.4) */
            if (wave == 0u) {
                uint32_t spins = 0u, okS = 1u;
                while (__hip_atomic_load(countWord, QZ_RLX_SYSTEM) == QZSTD_HIP_NSEQ_STAGING) {
                    if (++spins > spinLimit) { okS = 0u; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                /* the host stored the slice, THEN released the count word (memcpy + release-CAS, host/qatseqprod.c: qzServiceBlock): the
                 * acquire that pairs with it — the slice loads below must not be satisfied from anything older (round-5 ADVICE: until now
                 * this rested on the loads being uncached system-scope loads issued behind the barrier, not on the memory model) */
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                if (lane == 0u) ctl[18] = okS;
            }
            __syncthreads();
            const bool staged = rdfirst(ctl[18]) != 0u;
            for (uint32_t o = blk.parseFrom + tid * 16u; o < end && staged; o += (uint32_t)kThreads * 16u) {
                const u64 a = __hip_atomic_load((const u64 *)(hSrc + o), QZ_RLX_SYSTEM);
                const u64 b = __hip_atomic_load((const u64 *)(hSrc + o + 8u), QZ_RLX_SYSTEM);
                __hip_atomic_store((u64 *)(dSrc + o), a, QZ_RLX_AGENT);
                __hip_atomic_store((u64 *)(dSrc + o + 8u), b, QZ_RLX_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* every storing wave drains before the flag */
            __syncthreads();
            if (wave == 0u) {
                if (lane == 0u && staged) __hip_atomic_store(&sv->sliceFlag[slotIdx][k], epoch, QZ_RLX_AGENT); /* (never staged: the items behind this one give up as well) */
                /* the slices before this one: copied by the items before it (handed out earlier, waiting for nothing) */
                uint32_t spins = 0u, ok = 1u;
                for (;;) {
                    const bool there = lane >= k || __hip_atomic_load(&sv->sliceFlag[slotIdx][lane], QZ_RLX_AGENT) == epoch;
                    if (__all(there)) break;
                    if (++spins > spinLimit) { ok = 0u; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* ONE acquire after the match; plain loads from here */
                if (lane == 0u) ctl[17] = staged ? ok : 0u;
            }
            __syncthreads();
        }
        uint32_t count = QZSTD_HIP_NSEQ_ERROR;
        /* the request's scratch: entries of the history [BLOCK_MAX], entries of the items' own positions [BLOCK_MAX], first links [BLOCK_MAX],
         * one head table per item [kSvcMaxItems][kSvcTabStride] */
#define QZ_SVC_ITEM(A, L, R, C, T)                                                                                                            \
    qz_item<L, R, C, T, false>(A, blk, dSrc, out, (C) ? (uint4 *)q7 : nullptr, (C) ? (uint32_t *)(q7 + (u64)QZSTD_HIP_BLOCK_MAX * (8ull * QZSTD_HIP_CHAIN_ENTRY_LINKS)) : nullptr, \
                               HistShare{ (C) ? &sv->histFlag[slotIdx][0] : nullptr, k, epoch, spinLimit, nItemsReq, &sv->tabFlag[slotIdx][0],   \
                                          &sv->linkFlag[slotIdx][0], (C) ? (uint32_t *)(q7 + (u64)QZSTD_HIP_BLOCK_MAX * (8ull * QZSTD_HIP_CHAIN_ENTRY_LINKS + 4ull)) : nullptr })
        if (rdfirst(ctl[17]) != 0u) {
            if (MULTI) {
                LaunchArgs a2 = args;
                const uint32_t pi = rdfirst(svc_profile_index((uint32_t)(q6 >> 24) & 0xFFu));
                a2.prof = profs[pi < kSvcProfiles ? pi : 0u]; /* (uniform: the dispatcher only queues levels of the table) */
                const bool chain = a2.prof.chainDepth != 0u, rpt = a2.prof.repWin != 0u, turns = a2.prof.subTileLog != 0u;
                if (chain) count = rpt ? QZ_SVC_ITEM(a2, false, true, true, true) : QZ_SVC_ITEM(a2, false, false, true, true);
                else if (turns) count = rpt ? QZ_SVC_ITEM(a2, false, true, false, true) : QZ_SVC_ITEM(a2, false, false, false, true);
                else count = rpt ? QZ_SVC_ITEM(a2, false, true, false, false) : QZ_SVC_ITEM(a2, false, false, false, false);
            } else {
                count = QZ_SVC_ITEM(args, HAS_LONG, REP, CHAIN, TURNS);
            }
        } else if (tid == 0u) (void)__hip_atomic_fetch_add(&sv->spinFails, 1u, QZ_RLX_AGENT);
#undef QZ_SVC_ITEM
        /* ---- completion: every wave's result stores are performed, then the count — the host's flag — with a system-scope release ---- */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == (uint32_t)kMatchThreads) { /* lane 0 of the parse wave: it holds the count */
            __hip_atomic_store(countWord, count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            (void)__hip_atomic_fetch_add(&sv->done, 1u, QZ_RLX_AGENT);
        }
    }
}

/* one wave: host ring -> device queue, and the service's life cycle */
__global__ __launch_bounds__(64) void qzstd_service_dispatcher(SvcHost *hs, SvcDev *sv, uint32_t served, uint32_t idleUs, uint32_t drainUs)
{
    const uint32_t lane = threadIdx.x;
    u64 head = __hip_atomic_load(&hs->consumed, QZ_RLX_SYSTEM); /* where the previous launch stopped */
    uint32_t tail = 0u;                                          /* items queued in this launch */
    u64 lastWork = __builtin_amdgcn_s_memrealtime();             /* 100 MHz */
    u64 polls = 0, seen = 0;
    for (;;) {
        u64 g = 0;
        if ((++polls & 255u) == 0u && lane < 6u) { /* diagnostics for qzstd_hip_service_info */
            const u64 v = lane == 0u ? polls : lane == 1u ? seen : lane == 2u ? (u64)tail : lane == 3u ? (u64)__hip_atomic_load(&sv->started, QZ_RLX_AGENT)
                        : lane == 4u ? (u64)__hip_atomic_load(&sv->taken, QZ_RLX_AGENT) : (u64)__hip_atomic_load(&sv->done, QZ_RLX_AGENT);
            __hip_atomic_store(&hs->dbg[lane], v, QZ_RLX_SYSTEM);
        }
        if (lane < 8u) g = __hip_atomic_load(&hs->ring[head & (kSvcRing - 1u)][lane], QZ_RLX_SYSTEM);
        else if (lane == 8u) g = (u64)__hip_atomic_load(&hs->quitReq, QZ_RLX_SYSTEM);
        const bool mine = lane >= 8u || (g >> 56) == svc_tag(head, kSvcRing);
        const bool quitReq = rdlane((uint32_t)g, 8u) != 0u;
        if (__all(mine) && !quitReq) {
            const u64 r0 = svc_payload(__shfl(g, 0)), r1 = svc_payload(__shfl(g, 1)), r2 = svc_payload(__shfl(g, 2)),
                      r3 = svc_payload(__shfl(g, 3)), r4 = svc_payload(__shfl(g, 4)), r5 = svc_payload(__shfl(g, 5)),
                      r6 = svc_payload(__shfl(g, 6)), r7 = svc_payload(__shfl(g, 7));
            const uint32_t srcLen = (uint32_t)r4 & 0x3FFFFu, itemBytes = (uint32_t)(r4 >> 18) & 0x3FFFFu;
            const uint32_t nItems = (uint32_t)(r4 >> 36) & 63u, slotIdx = (uint32_t)(r4 >> 42) & (kSvcSlots - 1u);
            const uint32_t cap = (uint32_t)r5 & 0xFFFFFFu, epoch = (uint32_t)r6 & 0xFFFFFFu, lv = (uint32_t)(r6 >> 24) & 0xFFu;
            const bool sane = nItems >= 1u && nItems <= kSvcMaxItems && itemBytes != 0u && srcLen != 0u && srcLen <= QZSTD_HIP_BLOCK_MAX &&
                              (u64)(nItems - 1u) * itemBytes < srcLen && (lv & 0x7Fu) >= 1u && (lv & 0x7Fu) <= 12u &&
                              ((served >> svc_profile_index(lv)) & 1u) != 0u; /* served: one bit per (level, repeat-aware) the resident workers take */
            if (!sane) { /* a level the resident workers do not serve, or nonsense: handed back, the caller takes the launch path */
                if (lane < umin(nItems, kSvcMaxItems)) __hip_atomic_store((uint32_t *)r3 + lane, kSvcRejected, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                /* room in the queue: items still running keep nothing in it (a worker copies its entry first), so the bound is
                 * generous — wait while more than a queue's worth is unfinished */
                uint32_t spins = 0u;
                while (tail + nItems - __hip_atomic_load(&sv->done, QZ_RLX_AGENT) > kSvcQueue && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(8);
                if (lane < nItems) {
                    const u64 t = (u64)tail + lane, tg = svc_tag(t, kSvcQueue) << 56;
                    u64 *e = sv->items[t & (kSvcQueue - 1u)];
                    const uint32_t from = lane * itemBytes;
                    const uint32_t upTo = umin(srcLen, from + itemBytes);
                    __hip_atomic_store(e + 0, tg | r0, QZ_RLX_AGENT);
                    __hip_atomic_store(e + 1, tg | r1, QZ_RLX_AGENT);
                    __hip_atomic_store(e + 2, tg | (r2 + (u64)lane * cap * 16ull), QZ_RLX_AGENT);
                    __hip_atomic_store(e + 3, tg | (r3 + 4ull * lane), QZ_RLX_AGENT);
                    __hip_atomic_store(e + 4, tg | upTo | ((u64)from << 18) | ((u64)lane << 36) | ((u64)slotIdx << 42), QZ_RLX_AGENT);
                    __hip_atomic_store(e + 5, tg | cap | ((u64)nItems << 24), QZ_RLX_AGENT);
                    __hip_atomic_store(e + 6, tg | epoch | ((u64)lv << 24), QZ_RLX_AGENT);
                    __hip_atomic_store(e + 7, tg | r7, QZ_RLX_AGENT); /* the items of a request share its scratch (HistShare) */
                }
                tail += nItems;
            }
            head++;
            seen++;
            if (lane == 0u) __hip_atomic_store(&hs->consumed, head, QZ_RLX_SYSTEM);
            lastWork = __builtin_amdgcn_s_memrealtime();
            continue;
        }
        const u64 now = __builtin_amdgcn_s_memrealtime();
        const bool drained = __hip_atomic_load(&sv->done, QZ_RLX_AGENT) == tail;
        if (quitReq || (drained && now - lastWork > (u64)idleUs * 100ull)) {
            if (!quitReq) {
                /* idle: say so, then look at the ring once more — a request written after this look finds `state` 2 or 0 and
                 * its writer launches the service again (PCIe: the read does not pass the write) */
                if (lane == 0u) __hip_atomic_store(&hs->state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                u64 g2 = 0;
                if (lane < 8u) g2 = __hip_atomic_load(&hs->ring[head & (kSvcRing - 1u)][lane], QZ_RLX_SYSTEM);
                if (__all(lane >= 8u || (g2 >> 56) == svc_tag(head, kSvcRing))) {
                    if (lane == 0u) __hip_atomic_store(&hs->state, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    lastWork = now;
                    continue;
                }
            }
            /* leave: let the items in flight finish (bounded), tell the workers, report */
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(&sv->done, QZ_RLX_AGENT) != tail && __builtin_amdgcn_s_memrealtime() - t0 < (u64)drainUs * 100ull)
                __builtin_amdgcn_s_sleep(16);
            if (lane == 0u) {
                __hip_atomic_store(&sv->quit, 1u, QZ_RLX_AGENT);
                __hip_atomic_store(&hs->itemsDone, hs->itemsDone + __hip_atomic_load(&sv->done, QZ_RLX_AGENT), QZ_RLX_SYSTEM);
                __hip_atomic_store(&hs->spinFails, hs->spinFails + __hip_atomic_load(&sv->spinFails, QZ_RLX_AGENT), QZ_RLX_SYSTEM);
                __hip_atomic_store(&hs->state, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

/* Does the LDS hand the lanes of ONE ds_max_rtn instruction that hit the same address their values in lane order?  Every
 * lane of wave 0 inserts an increasing value into the slot a pattern gives it and must get back the value of the nearest lower
 * lane with the same slot (0 if none); several patterns, among them "all lanes one slot" and runs.  The probe runs under the
 * contention the kernel has (round-2 verdict): a workgroup of nine waves, the other eight hammering the same LDS — returning
 * and plain atomics, reads and writes on neighbouring words, bank conflicts included — while wave 0 is measured. */
__global__ __launch_bounds__(kThreads) void qzstd_probe_lds_order(uint32_t *bad)
{
    __shared__ uint32_t slots[64];
    __shared__ uint32_t noise[4096];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t wrong = 0;
    for (uint32_t i = tid; i < 4096u; i += kThreads) noise[i] = 0u;
    for (uint32_t round = 0; round < 64u; round++) {
        if (wave == 0u) slots[lane] = 0u;
        __syncthreads();
        if (wave == 0u) {
            uint32_t slot;
            switch (round & 7u) {
            case 0: slot = 0u; break;
            case 1: slot = lane & 1u; break;
            case 2: slot = lane >> 3; break;
            case 3: slot = lane % 7u; break;
            case 4: slot = (lane * 2654435761u + round) >> 28; break;
            case 5: slot = (lane * 0x85EBCA77u + round) >> 26; break;
            case 6: slot = lane < 32u ? 5u : lane & 3u; break;
            default: slot = (lane ^ (lane >> 2)) & 31u; break;
            }
            const uint32_t mine = (round << 8) + lane + 1u;
            const uint32_t got = atomicMax(&slots[slot], mine);
            uint32_t want = 0u;
            for (uint32_t l = 0; l < 64u; l++) {
                const uint32_t sl = (uint32_t)__shfl((int)slot, (int)l);
                if (l < lane && sl == slot) want = (round << 8) + l + 1u;
            }
            if (got != want) wrong++;
        } else {
            /* the other eight waves: what the matcher waves do to the LDS while the insert wave works */
            uint32_t acc = 0u;
            for (uint32_t k = 0; k < 24u; k++) {
                const uint32_t a = (tid * 2654435761u + k * 40503u + round * 977u) >> 20; /* 0 .. 4095 */
                acc += atomicMax(&noise[a], tid + k);
                atomicMin(&noise[(a * 33u + 7u) & 4095u], acc);
                acc ^= noise[(a + 64u * k) & 4095u];
                noise[(a ^ 1u) & 4095u] = acc;
            }
            if (acc == 0xDEADBEEFu) wrong += 0u * acc; /* keep the traffic */
        }
        __syncthreads();
    }
    if (wrong) atomicAdd(bad, wrong);
}

thread_local char g_err[256] = "";

int fail(const char *what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -1;
}
int fail_msg(const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s", what);
    return -1;
}

#define QZ_CHECK(call, what)                       \
    do {                                           \
        hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) return fail(what, e_); \
    } while (0)

/* ---- the resident service: host-side state (the kernels: qzstd_service_worker / qzstd_service_dispatcher above) ---- */
struct Service {
    std::mutex mu; /* set-up, launches, stops */
    SvcHost *hs = nullptr;
    SvcDev *dv = nullptr;
    hipStream_t sWork = nullptr, sDisp = nullptr;
    hipEvent_t ev = nullptr;
    int level = 0;   /* the level the resident workers were launched for */
    uint32_t served = 0; /* one bit per (level, repeat-aware) they serve: svc_profile_index (the multi-level worker: levels 1-2 and 5-12) */
    qzstd_hip_profile_t *dProfs = nullptr; /* device: the profiles of every level, for the multi-level worker */
    int workers = 0;
    int broken = 0;  /* a request timed out or a launch failed: the service is not used again */
    std::atomic<unsigned long long> reserve{0}; /* request numbers handed to submitters */
    unsigned long launches = 0, requests = 0, refused = 0;
    /* launches of the batch paths whose workgroups cannot share a CU with a worker (levels 3-4 fill a CU's LDS): while one of
     * them is in flight the service stays down — resident workers would hold the LDS its remaining workgroups are waiting for,
     * for as long as requests keep coming (found by the fuzz driver: an announced level-3 kernel, the service launched again
     * behind it, its dispatcher queued behind that kernel in a shared hardware queue: nobody could move) */
    static constexpr int kBig = 64;
    hipEvent_t bigEv[kBig] = {};
    bool bigUsed[kBig] = {};
    size_t bigLds[kBig] = {};   /* LDS per workgroup of the launch the event stands for */
    int bigNext = 0;
    size_t lds = 0;             /* LDS of one worker of the service that runs (or ran last) */
    long long launchNs = 0;     /* when it was launched last (CLOCK_MONOTONIC) */
    bool sawWorkers = false;    /* its dispatcher has reported workers that began */
    bool wide = false;          /* a service whose workers leave no room for ANY batch workgroup on their CU (levels 3-4) has run on this
                                 * device: from then on every launch is remembered by an event */
};
Service g_svc[64];
std::atomic<int> g_svcFreeze{0}; /* > 0: memory is being freed (hipFree / hipHostFree wait for every stream of the device) */

struct SvcConfig { int enabled, workers, idleUs, spinLimit; };
const SvcConfig &svc_config()
{
    static const SvcConfig c = [] {
        SvcConfig k;
        const char *e = getenv("QZSTD_HIP_SERVICE"), *w = getenv("QZSTD_HIP_SERVICE_WORKERS"), *i = getenv("QZSTD_HIP_SERVICE_IDLE_US");
        k.enabled = e ? atoi(e) : 1;
        k.workers = w ? atoi(w) : 0; /* 0 = one per CU */
        k.idleUs = i ? atoi(i) : 20000;
        if (k.idleUs < 100) k.idleUs = 100;
        k.spinLimit = 1 << 20; /* polls of a slice flag before an item gives up (seconds) */
        return k;
    }();
    return c;
}

/* asks the resident kernels of one device to leave and waits (bounded) until they have; 0 = stopped (or never running) */
int svc_stop_locked(Service &s, unsigned waitMs)
{
    if (!s.hs) return 0;
    if (__atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE) != 0u) {
        __atomic_store_n(&s.hs->quitReq, 1u, __ATOMIC_RELEASE);
        struct timespec t0, t;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (__atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE) != 0u) {
            clock_gettime(CLOCK_MONOTONIC, &t);
            if ((t.tv_sec - t0.tv_sec) * 1000ll + (t.tv_nsec - t0.tv_nsec) / 1000000ll > (long long)waitMs) { s.broken = 1; return 1; }
            sched_yield();
        }
    }
    return 0;
}

/* before memory is freed: hipFree / hipHostFree wait for every stream, which a resident kernel never lets finish */
struct SvcFreeze {
    SvcFreeze()
    {
        g_svcFreeze.fetch_add(1);
        for (int d = 0; d < 64; d++) {
            Service &s = g_svc[d];
            if (!s.hs) continue;
            std::lock_guard<std::mutex> g(s.mu);
            (void)svc_stop_locked(s, 2000);
        }
    }
    ~SvcFreeze() { g_svcFreeze.fetch_sub(1); }
};

} // namespace

extern "C" {

const char *qzstd_hip_last_error(void) { return g_err; }

/* The library carries one code object (gfx950).  Only devices that can run it are counted, and the `device` argument
 * of every entry point indexes that filtered list (reference: instance discovery keeps only usable DC instances,
 * /root/reference/src/qatseqprod.c:529-600). */
static std::once_flag g_devOnce;
static int g_devCount = -1;
static int g_devMap[64];
static int g_devReplicas = 1; /* QZSTD_HIP_REPLICATE_DEVICES (test only): logical devices per physical one */
static int g_ldsOrdered[64]; /* per device: -1 not probed yet, 0 no, 1 yes (probe_lds_order) */
static std::mutex g_probeMu;

/* runs qzstd_probe_lds_order once per device; any failure of the probe itself counts as "no" (the ballot path needs nothing).
 * Called for every device when the devices are enumerated (probe_devices, under g_devOnce) — BEFORE any resident service can
 * exist: the probe allocates and frees device memory, and hipFree waits for every stream of the device, so running it lazily
 * from a launch (under the device's service mutex, as rounds 2-3 did) could wait for as long as per-block requests kept a
 * service of another level alive, with the mutex held (round-3 ADVICE, medium).  Later calls only read the verdict. */
static int probe_lds_order(int device, int physDev)
{
    {
        const int known = __atomic_load_n(&g_ldsOrdered[device], __ATOMIC_ACQUIRE);
        if (known >= 0) return known;
    }
    std::lock_guard<std::mutex> g(g_probeMu);
    if (g_ldsOrdered[device] >= 0) return g_ldsOrdered[device];
    int verdict = 0;
    const char *force = getenv("QZSTD_HIP_ORDERED_LDS"); /* 0 = never use the single-instruction insert */
    uint32_t *dBad = nullptr, hBad = 1u;
    if (!(force && atoi(force) == 0) && hipSetDevice(physDev) == hipSuccess && hipMalloc(&dBad, sizeof(uint32_t)) == hipSuccess) {
        if (hipMemset(dBad, 0, sizeof(uint32_t)) == hipSuccess) {
            hipLaunchKernelGGL(qzstd_probe_lds_order, dim3(512), dim3(kThreads), 0, 0, dBad); /* two workgroups per CU, nine waves each */
            if (hipGetLastError() == hipSuccess && hipMemcpy(&hBad, dBad, sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess)
                verdict = hBad == 0u;
        }
        (void)hipFree(dBad);
    }
    (void)hipGetLastError();
    __atomic_store_n(&g_ldsOrdered[device], verdict, __ATOMIC_RELEASE);
    return verdict;
}

static void probe_devices()
{
    int n = 0;
    /* Announcements run on one stream per slot; the runtime folds its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and launches
     * of different streams that share a queue run one after the other — four 32-block launches at a time leave the GPU mostly idle and the
     * announcing threads waiting (batch front-end, 16 threads, level 1: 12.5 GB/s with 4 queues, 14.6 with 8, 18.3 with 16, 17.1 with 24).
     * The variable is read when the runtime starts, so it only helps when this is the process's first HIP call; a value the
     * environment already carries is left alone.  QZSTD_HIP_HW_QUEUES=0 leaves the runtime's default. */
    {
        const char *q = getenv("QZSTD_HIP_HW_QUEUES");
        const int want = q && *q ? atoi(q) : 16;
        if (want > 0 && want <= 64) {
            char buf[16];
            snprintf(buf, sizeof(buf), "%d", want);
            (void)setenv("GPU_MAX_HW_QUEUES", buf, 0);
        }
    }
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail("hipGetDeviceCount", e); (void)hipGetLastError(); g_devCount = -1; return; }
    g_devCount = 0;
    for (int d = 0; d < 64; d++) g_ldsOrdered[d] = -1;
    for (int d = 0; d < n && g_devCount < 64; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) continue; /* no code object for it */
        g_devMap[g_devCount++] = d;
    }
    if (g_devCount == 0) fail_msg("no gfx950 device among the visible HIP devices");
    /* TEST ONLY — QZSTD_HIP_REPLICATE_DEVICES=k lists every physical device k times: the library then sees k x n LOGICAL devices, each with
     * its own streams, batches, pinned buffers and resident service, and the host's split of an announcement into per-GPU ranges, its state
     * placement and QZSTD_deviceStats run over several devices on a box that has one GPU (round-5 verdict: the N > 1 path had only ever run
     * against tests/mock/mock_hip.c).  Replicas share the physical GPU's CUs and LDS: no speed to be had, and resident services of two replicas
     * compete for the same CUs (the tests give each service half the CUs: QZSTD_HIP_SERVICE_WORKERS). */
    {
        const char *r = getenv("QZSTD_HIP_REPLICATE_DEVICES");
        const int k = r && *r ? atoi(r) : 1;
        const int n0 = g_devCount;
        for (int c = 1; c < k && c < 64; c++)
            for (int d = 0; d < n0 && g_devCount < 64; d++) g_devMap[g_devCount++] = g_devMap[d];
        if (n0 > 0 && g_devCount > n0) g_devReplicas = g_devCount / n0;
    }
    for (int d = 0; d < g_devCount; d++) (void)probe_lds_order(d, g_devMap[d]); /* now: nothing is resident yet (see probe_lds_order) */
}

static inline int phys(int device)
{
    std::call_once(g_devOnce, probe_devices);
    return device >= 0 && device < g_devCount ? g_devMap[device] : -1;
}

#define QZ_SET_DEVICE(device)                                            \
    do {                                                                 \
        const int pd_ = phys(device);                                    \
        if (pd_ < 0) return fail_msg("device index out of range");       \
        QZ_CHECK(hipSetDevice(pd_), "hipSetDevice");                     \
    } while (0)

int qzstd_hip_device_count(void)
{
    std::call_once(g_devOnce, probe_devices);
    return g_devCount;
}

int qzstd_hip_device_name(int device, char *buf, size_t bufLen)
{
    hipDeviceProp_t prop;
    if (phys(device) < 0) return fail_msg("device index out of range");
    QZ_CHECK(hipGetDeviceProperties(&prop, phys(device)), "hipGetDeviceProperties");
    if (buf && bufLen) snprintf(buf, bufLen, "%s (%s, %d CUs, %zu KiB LDS/WG)", prop.name, prop.gcnArchName,
                                prop.multiProcessorCount, prop.sharedMemPerBlock >> 10);
    return 0;
}

void *qzstd_hip_malloc(int device, size_t bytes)
{
    void *p = nullptr;
    if (phys(device) < 0 || hipSetDevice(phys(device)) != hipSuccess) return nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { fail("hipMalloc", e); return nullptr; }
    return p;
}

void qzstd_hip_free(int device, void *dptr)
{
    if (!dptr) return;
    SvcFreeze frozen; /* hipFree waits for every stream of the device: the resident service has to leave first */
    if (phys(device) >= 0 && hipSetDevice(phys(device)) == hipSuccess) (void)hipFree(dptr);
}

void *qzstd_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { fail("hipHostMalloc", e); return nullptr; }
    return p;
}

void *qzstd_hip_host_device_ptr(void *hptr)
{
    void *d = nullptr;
    if (!hptr) return nullptr;
    hipError_t e = hipHostGetDevicePointer(&d, hptr, 0);
    if (e != hipSuccess) { fail("hipHostGetDevicePointer", e); return nullptr; }
    return d;
}

void *qzstd_hip_host_alloc_coherent(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { fail("hipHostMalloc(coherent)", e); return nullptr; }
    return p;
}

/* The GPU's host NUMA node: the runtime's attribute first, the PCI function's sysfs entry second */
int qzstd_hip_device_numa_node(int device)
{
    const int pd = phys(device);
    if (pd < 0) return -1;
    int node = -1;
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, pd) == hipSuccess && node >= 0) return node;
    (void)hipGetLastError();
    char bus[32] = "", path[96];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), pd) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a'); /* sysfs spells the address in lower case */
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

/* Pinned host memory on a NUMA node: hipHostMallocNumaUser makes the runtime allocate under the CALLING THREAD's memory policy, so
 * the policy is set to "prefer `node`" around the call (raw system calls: no libnuma in the image) and put back afterwards.  A
 * process that may not set a policy (seccomp, containers without CAP_SYS_NICE for other nodes) gets the memory anyway, unplaced. */
void *qzstd_hip_host_alloc_on_node(size_t bytes, int node, int coherent)
{
    const unsigned flags = hipHostMallocPortable | hipHostMallocMapped | (coherent ? hipHostMallocCoherent : 0u);
    void *p = nullptr;
    if (node >= 0 && node < 1024) {
        unsigned long want[16] = { 0 }, old[16] = { 0 };
        int oldMode = 0;
        want[node / (8 * sizeof(unsigned long))] = 1ul << (node % (8 * sizeof(unsigned long)));
        const bool got = syscall(SYS_get_mempolicy, &oldMode, old, (unsigned long)(8 * sizeof(old)), nullptr, 0ul) == 0;
        if (got && syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, want, (unsigned long)(8 * sizeof(want))) == 0) {
            const hipError_t e = hipHostMalloc(&p, bytes, flags | hipHostMallocNumaUser);
            (void)syscall(SYS_set_mempolicy, oldMode, oldMode == 0 /* MPOL_DEFAULT takes no mask */ ? nullptr : old,
                          oldMode == 0 ? 0ul : (unsigned long)(8 * sizeof(old)));
            if (e == hipSuccess) return p;
            (void)hipGetLastError();
            p = nullptr;
        }
    }
    const hipError_t e = hipHostMalloc(&p, bytes, flags);
    if (e != hipSuccess) { fail(coherent ? "hipHostMalloc(coherent)" : "hipHostMalloc", e); return nullptr; }
    return p;
}

int qzstd_hip_host_node_of(const void *hptr)
{
    int node = -1;
    if (!hptr) return -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, hptr, 3ul /* MPOL_F_NODE | MPOL_F_ADDR */) != 0) return -1;
    return node;
}

void qzstd_hip_host_free(void *hptr)
{
    if (!hptr) return;
    SvcFreeze frozen; /* hipHostFree waits for the device's streams too */
    (void)hipHostFree(hptr);
}

void *qzstd_hip_stream_create(int device)
{
    hipStream_t s = nullptr;
    if (phys(device) < 0 || hipSetDevice(phys(device)) != hipSuccess) return nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { fail("hipStreamCreate", e); return nullptr; }
    return (void *)s;
}

void qzstd_hip_stream_destroy(int device, void *stream)
{
    if (stream && phys(device) >= 0 && hipSetDevice(phys(device)) == hipSuccess) (void)hipStreamDestroy((hipStream_t)stream);
}

int qzstd_hip_stream_sync(int device, void *stream)
{
    QZ_SET_DEVICE(device);
    QZ_CHECK(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
    return 0;
}

int qzstd_hip_stream_query(int device, void *stream)
{
    QZ_SET_DEVICE(device);
    hipError_t e = hipStreamQuery((hipStream_t)stream);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) return 1;
    return fail("hipStreamQuery", e);
}

int qzstd_hip_stream_wait(int device, void *stream, unsigned timeoutMs)
{
    QZ_SET_DEVICE(device);
    /* poll instead of hipStreamSynchronize: a wedged kernel must not take the calling thread with it.  Busy polls for
     * the first QZSTD_HIP_SPIN_US microseconds (default 50), then naps between polls so
     * that a waiting caller does not burn a core other callers could entropy-code on (QZSTD_HIP_NAP_US, default 20) */
    static const long long spinUs = [] { const char *v = getenv("QZSTD_HIP_SPIN_US"); return v ? atoll(v) : 50ll; }();
    static const long napNs = [] { const char *v = getenv("QZSTD_HIP_NAP_US"); return (v ? atol(v) : 20l) * 1000l; }();
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const hipError_t e = hipStreamQuery((hipStream_t)stream);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail("hipStreamQuery", e);
        clock_gettime(CLOCK_MONOTONIC, &t);
        const long long us = (long long)(t.tv_sec - t0.tv_sec) * 1000000ll + (t.tv_nsec - t0.tv_nsec) / 1000;
        if (us >= (long long)timeoutMs * 1000ll) {
            snprintf(g_err, sizeof(g_err), "stream still busy after %u ms", timeoutMs);
            return 1;
        }
        if (us < spinUs) continue;
        if (napNs <= 0) sched_yield();
        else { const struct timespec nap = { 0, us < 20000 ? napNs : 200000l }; nanosleep(&nap, nullptr); }
    }
}

int qzstd_hip_memcpy_h2d(int device, void *stream, void *dst, const void *src, size_t bytes)
{
    QZ_SET_DEVICE(device);
    QZ_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync H2D");
    return 0;
}

/* pinned host memory -> device memory by a KERNEL on the stream: 16 bytes per lane and step, every wave streaming its own 16 KiB stripes.
 * For the staging copies of announcements (host/qatseqprod.c, qzLaunchPart): hipMemcpyAsync holds the calling thread for 0.8-1.1 ms per 4 MiB
 * when 16 threads announce (the runtime's copy path and its locks); a launch costs 10-30 us, and the copy then runs at the bus's rate in front
 * of the match-finder on the same stream.  src_dev = the DEVICE address of the pinned buffer (qzstd_hip_host_device_ptr); bytes a multiple
 * of 16, both addresses 16-byte aligned. */
}
namespace {
__global__ __launch_bounds__(256) void qzstd_copy_in_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
}
extern "C" {
int qzstd_hip_copy_in(int device, void *stream, void *dst, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return 0;
    if (!dst || !src_dev || (bytes & 15u) || ((uintptr_t)dst & 15u) || ((uintptr_t)src_dev & 15u) || bytes > ((size_t)1 << 34))
        return fail_msg("qzstd_hip_copy_in: null, unaligned or oversized");
    QZ_SET_DEVICE(device);
    const uint32_t n16 = (uint32_t)(bytes >> 4);
    uint32_t groups = (n16 + 1023u) / 1024u; /* four steps per lane */
    if (groups > 1024u) groups = 1024u;
    hipLaunchKernelGGL(qzstd_copy_in_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, static_cast<uint4 *>(dst), static_cast<const uint4 *>(src_dev), n16);
    QZ_CHECK(hipGetLastError(), "launch qzstd_copy_in_kernel");
    return 0;
}

int qzstd_hip_memcpy_d2h(int device, void *stream, void *dst, const void *src, size_t bytes)
{
    QZ_SET_DEVICE(device);
    QZ_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync D2H");
    return 0;
}

int qzstd_hip_memset(int device, void *stream, void *dst, int value, size_t bytes)
{
    QZ_SET_DEVICE(device);
    QZ_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream), "hipMemsetAsync");
    return 0;
}

int qzstd_hip_memcpy2d_d2h(int device, void *stream, void *dst, size_t dpitch, const void *src, size_t spitch,
                           size_t width, size_t height)
{
    QZ_SET_DEVICE(device);
    QZ_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, (hipStream_t)stream),
             "hipMemcpy2DAsync D2H");
    return 0;
}

int qzstd_hip_find_sequences(int device, void *stream, int level, const void *d_src,
                             const qzstd_hip_block_t *d_blocks, uint32_t nBlocks, uint32_t maxBlockLen,
                             void *d_seqs, uint32_t *d_nseq, void *d_work, size_t workBytes)
{
    LaunchArgs a;
    if (nBlocks == 0) return 0;
    if (!d_src || !d_blocks || !d_seqs || !d_nseq) return fail_msg("qzstd_hip_find_sequences: null pointer");
    if (maxBlockLen > QZSTD_HIP_BLOCK_MAX) return fail_msg("qzstd_hip_find_sequences: block larger than 128 KiB");
    if (qzstd_hip_profile_for_level(level, maxBlockLen, &a.prof))
        return fail_msg("qzstd_hip_find_sequences: level outside 1..12 (optionally | QZSTD_HIP_LEVEL_REPCODES)");
    if (a.prof.tileLog != kTileLog || a.prof.extLog < 8 || a.prof.extLog > 15 || a.prof.capLen > 48 || a.prof.capLen < 32 || /* (the kernels measure candidates up to 48 bytes: the 16-byte head and one step of 32) */
        a.prof.minMatch < 4 || a.prof.hashBytes < 4 || a.prof.hashBytes > 8 || a.prof.repWin > 16 || a.prof.chainDepth > 64 ||
        a.prof.lazy > 4 || (a.prof.subTileLog != 0u && a.prof.subTileLog != 6u) || (a.prof.segLog != 0u && (a.prof.segLog < kTileLog || a.prof.segLog > 17u)) ||
        (a.prof.chainDepth && (a.prof.subTileLog != 6u || a.prof.longSize || a.prof.nearTab)) || (a.prof.longSize && a.prof.subTileLog))
        return fail_msg("qzstd_hip_find_sequences: unsupported profile");
    const size_t lds = qzstd_hip_lds_bytes(level, maxBlockLen);
    if (lds == 0) return fail_msg("qzstd_hip_find_sequences: LDS budget exceeded");
    QZ_SET_DEVICE(device);
    /* A resident service holds one worker's LDS (65 KB) on every CU.  A launch whose workgroups cannot share a CU with a worker
     * (levels 3-4 fill a CU) would wait for as long as requests keep the service alive — and the service must not come back
     * while such a launch is in flight: it is asked to leave first, and the launch is remembered by an event that the
     * service's next launch checks.  One mutex per device covers "stop, launch, record". */
    /* Generally: a launch and a running service clash when one workgroup of each do not fit a CU's LDS together.  Levels 3-4
     * clash with every service; a service of levels 3-4 clashes with every launch.  "Look at the service, stop it if it
     * clashes, launch, remember" is one critical section per device. */
    bool big = lds + qzstd_hip_lds_bytes(1, QZSTD_HIP_BLOCK_MAX) > kLdsPerCu;
    std::unique_lock<std::mutex> bigLock;
    if (device >= 0 && device < 64) {
        Service &sv = g_svc[device];
        bigLock = std::unique_lock<std::mutex>(sv.mu);
        big = big || sv.wide;
        if (sv.hs && __atomic_load_n(&sv.hs->state, __ATOMIC_ACQUIRE) != 0u && lds + sv.lds > kLdsPerCu) (void)svc_stop_locked(sv, 2000);
    }
    /* [near][long][rep][chain/turns: 0 none, 1 turns, 2 chain + turns] */
#define QZ_K(L, R, C, T, N) reinterpret_cast<const void *>(qzstd_find_sequences_kernel<L, R, C, T, N>)
    static const void *const variants[2][2][2][3] = {
        { { { QZ_K(false, false, false, false, false), QZ_K(false, false, false, true, false), QZ_K(false, false, true, true, false) },
            { QZ_K(false, true, false, false, false), QZ_K(false, true, false, true, false), QZ_K(false, true, true, true, false) } },
          { { QZ_K(true, false, false, false, false), nullptr, nullptr }, { QZ_K(true, true, false, false, false), nullptr, nullptr } } },
        { { { QZ_K(false, false, false, false, true), QZ_K(false, false, false, true, true), QZ_K(false, false, true, true, true) },
            { QZ_K(false, true, false, false, true), QZ_K(false, true, false, true, true), QZ_K(false, true, true, true, true) } },
          { { QZ_K(true, false, false, false, true), nullptr, nullptr }, { QZ_K(true, true, false, false, true), nullptr, nullptr } } } };
#undef QZ_K
    /* the LDS a variant may ask for is a per-function, per-device attribute (process-wide, not per thread): raise it
     * once per device to the most any level needs and never lower it */
    {
        static std::mutex attrMu;
        static bool attrDone[64];
        std::lock_guard<std::mutex> g(attrMu);
        if (phys(device) < 0) return fail_msg("qzstd_hip_find_sequences: device index out of range");
        if (!attrDone[device]) {
            size_t most = 0;
            for (int l = 1; l <= 12; l++) {
                const size_t b = qzstd_hip_lds_bytes(l, QZSTD_HIP_BLOCK_MAX);
                if (b > most) most = b;
            }
            for (int v = 0; v < 24; v++) {
                const void *f = variants[v / 12][(v / 6) % 2][(v / 3) % 2][v % 3];
                if (f) QZ_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)most),
                                "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
            }
            attrDone[device] = true;
        }
    }
    a.chain = nullptr;
    a.pwWords = 0;
    a.chainStride = 0;
    a.chainEntries = 0;
    a.orderedLds = 0;
    if (a.prof.chainDepth) {
        a.orderedLds = (uint32_t)probe_lds_order(device, phys(device));
        const size_t need = qzstd_hip_workspace_bytes(level, nBlocks, maxBlockLen);
        if (!d_work || workBytes < need) return fail_msg("qzstd_hip_find_sequences: workspace missing or too small (qzstd_hip_workspace_bytes)");
        a.chain = static_cast<uint4 *>(d_work);
        a.chainStride = (uint32_t)(need / nBlocks / sizeof(uint4));
        a.chainEntries = (uint32_t)(need / nBlocks / (4u * QZSTD_HIP_CHAIN_ENTRY_LINKS + 4u)) * kEQ;
    } else if (a.prof.repWin ? QZ_REP_DEFER != 0 : QZ_PLAIN_DEFER != 0) { /* the deferred parse below the chain levels: one parse word (4 B) per position */
        const size_t need = qzstd_hip_workspace_bytes(level, nBlocks, maxBlockLen);
        if (!d_work || workBytes < need || need == 0) return fail_msg("qzstd_hip_find_sequences: workspace missing or too small (qzstd_hip_workspace_bytes)");
        a.chain = static_cast<uint4 *>(d_work);
        a.chainStride = (uint32_t)(need / nBlocks / sizeof(uint4));
        a.pwWords = (uint32_t)(need / nBlocks / 33u * 8u); /* a region = 4 B per position + 8 B per 64 positions (the windows' start masks) */
    }
    if (a.prof.segLog != 12u) return fail_msg("qzstd_hip_find_sequences: unsupported profile (the deferred parse of a launch works on 4 KiB segments)");
    a.src = static_cast<const uint8_t *>(d_src);
    a.blocks = d_blocks;
    a.seqs = static_cast<uint4 *>(d_seqs);
    a.nseq = d_nseq;
#ifdef QZ_DEBUG_DUMP
    { const char *ab = getenv("QZSTD_HIP_ABLATE"); a.dbg = ab ? (uint32_t)atoi(ab) : 0u; }
#endif
#ifdef QZ_DEBUG_DUMP
    size_t ldsLaunch = lds; /* profiling: QZSTD_HIP_LDS_PAD = extra bytes per workgroup (fewer workgroups per CU: what occupancy is worth) */
    { const char *pad = getenv("QZSTD_HIP_LDS_PAD"); if (pad) ldsLaunch += (size_t)atoi(pad); }
#else
    const size_t ldsLaunch = lds;
#endif
    const dim3 grid(nBlocks), wg(kThreads);
    /* every block of the launch fits the ring: the kernels without a device-memory side of the compares (NEAR) */
    const int nearK = maxBlockLen <= kRing ? 1 : 0;
    const void *kernel = variants[nearK][a.prof.longSize ? 1 : 0][a.prof.repWin ? 1 : 0][a.prof.chainDepth ? 2 : (a.prof.subTileLog ? 1 : 0)];
    void *kargs[1] = { &a };
    if (!kernel) return fail_msg("qzstd_hip_find_sequences: unsupported profile (no kernel variant)");
    /* below the chain levels the kernels take the slot as a shift of the hash and mix at most three bytes behind the fourth through a 24-bit product */
    if (!a.prof.chainDepth && ((a.prof.tableSize & (a.prof.tableSize - 1u)) != 0u || (a.prof.longSize & (a.prof.longSize - 1u)) != 0u || a.prof.hashBytes > 7u))
        return fail_msg("qzstd_hip_find_sequences: unsupported profile (tables below the chain levels are powers of two, hashBytes <= 7)");
    QZ_CHECK(hipLaunchKernel(kernel, grid, wg, kargs, ldsLaunch, (hipStream_t)stream), "launch qzstd_find_sequences_kernel");
    QZ_CHECK(hipGetLastError(), "launch qzstd_find_sequences_kernel");
    if (big && bigLock.owns_lock()) {
        Service &sv = g_svc[device];
        int k = sv.bigNext;
        for (int tries = 0; tries < Service::kBig && sv.bigUsed[k]; tries++) { /* an event whose launch has finished can be used again */
            if (hipEventQuery(sv.bigEv[k]) != hipErrorNotReady) { (void)hipGetLastError(); sv.bigUsed[k] = false; break; }
            k = (k + 1) % Service::kBig;
        }
        if (sv.bigUsed[k]) { (void)hipEventSynchronize(sv.bigEv[k]); (void)hipGetLastError(); } /* 64 launches in flight: wait for one */
        sv.bigNext = (k + 1) % Service::kBig;
        if (!sv.bigEv[k] && hipEventCreateWithFlags(&sv.bigEv[k], hipEventDisableTiming) != hipSuccess) { sv.bigEv[k] = nullptr; (void)hipGetLastError(); }
        if (sv.bigEv[k] && hipEventRecord(sv.bigEv[k], (hipStream_t)stream) == hipSuccess) { sv.bigUsed[k] = true; sv.bigLds[k] = lds; }
        else (void)hipGetLastError();
    }
    return 0;
}

/* diagnostics: how many workgroups of the level's kernel the runtime says fit one CU (registers, LDS, wave slots); < 0 on error */
int qzstd_hip_occupancy(int device, int level)
{
    qzstd_hip_profile_t p;
    if (qzstd_hip_profile_for_level(level, QZSTD_HIP_BLOCK_MAX, &p)) return fail_msg("qzstd_hip_occupancy: bad level");
    const size_t lds = qzstd_hip_lds_bytes(level, QZSTD_HIP_BLOCK_MAX);
    QZ_SET_DEVICE(device);
    const void *k;
    if (p.chainDepth) k = p.repWin ? reinterpret_cast<const void *>(qzstd_find_sequences_kernel<false, true, true, true, false>) : reinterpret_cast<const void *>(qzstd_find_sequences_kernel<false, false, true, true, false>);
    else if (p.longSize) k = p.repWin ? reinterpret_cast<const void *>(qzstd_find_sequences_kernel<true, true, false, false, false>) : reinterpret_cast<const void *>(qzstd_find_sequences_kernel<true, false, false, false, false>);
    else if (p.subTileLog) k = p.repWin ? reinterpret_cast<const void *>(qzstd_find_sequences_kernel<false, true, false, true, false>) : reinterpret_cast<const void *>(qzstd_find_sequences_kernel<false, false, false, true, false>);
    else k = p.repWin ? reinterpret_cast<const void *>(qzstd_find_sequences_kernel<false, true, false, false, false>) : reinterpret_cast<const void *>(qzstd_find_sequences_kernel<false, false, false, false, false>);
    (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    QZ_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, kThreads, lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
    return n;
}

} /* extern "C" */

/* ---------------------------------------------------------------- the resident service (host side) ---------- */
namespace {

const void *svc_worker_variant(const qzstd_hip_profile_t &p)
{
#define QZ_W(L, R, C, T) reinterpret_cast<const void *>(qzstd_service_worker<L, R, C, T, false>)
    if (p.chainDepth) return p.repWin ? QZ_W(false, true, true, true) : QZ_W(false, false, true, true);
    if (p.longSize) return p.repWin ? QZ_W(true, true, false, false) : QZ_W(true, false, false, false); /* levels 3-4: a worker fills its CU's LDS
                                                                                                         * (see Service::wide) */
    if (p.repWin) return p.subTileLog ? QZ_W(false, true, false, true) : QZ_W(false, true, false, false);
    return p.subTileLog ? QZ_W(false, false, false, true) : QZ_W(false, false, false, false);
#undef QZ_W
}

/* index of a level (optionally | QZSTD_HIP_LEVEL_REPCODES) in the table of profiles: the host side of svc_profile_index */
int svc_level_index(int level)
{
    const int l = level & ~QZSTD_HIP_LEVEL_REPCODES;
    return (l >= 1 && l <= 12) ? (l - 1) + ((level & QZSTD_HIP_LEVEL_REPCODES) ? 12 : 0) : -1;
}

int svc_launch_locked(int device, Service &s, int level)
{
    const SvcConfig &cfg = svc_config();
    LaunchArgs a;
    memset(&a, 0, sizeof(a));
    if (qzstd_hip_profile_for_level(level, QZSTD_HIP_BLOCK_MAX, &a.prof)) return fail_msg("service: bad level");
    /* levels 1-2 and 5-12 share the 65 KB layout: ONE worker serves them all (QZSTD_HIP_SERVICE_MULTI=0: a worker per level, as before
     * round 4); levels 3-4 (136 KB) keep a worker of their own */
    static const bool multiOn = [] { const char *v = getenv("QZSTD_HIP_SERVICE_MULTI"); return !(v && atoi(v) == 0); }();
    const bool multi = multiOn && a.prof.longSize == 0u;
    const void *worker = multi ? reinterpret_cast<const void *>(qzstd_service_worker<false, false, false, false, true>) : svc_worker_variant(a.prof);
    size_t lds = qzstd_hip_lds_bytes(level, QZSTD_HIP_BLOCK_MAX);
    uint32_t served = 0u;
    qzstd_hip_profile_t profs[kSvcProfiles];
    memset(profs, 0, sizeof(profs));
    for (int l = 1; l <= 12; l++)
        for (int r = 0; r < 2; r++) {
            const int lv = l | (r ? QZSTD_HIP_LEVEL_REPCODES : 0), ix = svc_level_index(lv);
            qzstd_hip_profile_t p;
            if (ix < 0 || qzstd_hip_profile_for_level(lv, QZSTD_HIP_BLOCK_MAX, &p)) continue;
            profs[ix] = p;
            if (multi ? p.longSize == 0u : lv == level) {
                served |= 1u << ix;
                if (multi && qzstd_hip_lds_bytes(lv, QZSTD_HIP_BLOCK_MAX) > lds) lds = qzstd_hip_lds_bytes(lv, QZSTD_HIP_BLOCK_MAX);
            }
        }
    if (!worker || lds == 0) return fail_msg("service: level not served");
    if (a.prof.chainDepth || multi) a.orderedLds = (uint32_t)probe_lds_order(device, phys(device));
    QZ_SET_DEVICE(device);
    for (int k = 0; k < Service::kBig; k++) {
        if (!s.bigUsed[k]) continue;
        const hipError_t q = hipEventQuery(s.bigEv[k]);
        if (q == hipErrorNotReady) {
            if (s.bigLds[k] + lds > kLdsPerCu) return fail_msg("service: a launch that cannot share a CU with its workers is in flight");
            continue;
        }
        (void)hipGetLastError();
        s.bigUsed[k] = false;
    }
    if (lds + qzstd_hip_lds_bytes(1, QZSTD_HIP_BLOCK_MAX) > kLdsPerCu && !s.wide) {
        /* the first service of levels 3-4 on this device: launches up to now were not remembered (all of them go out under this
         * mutex, so none can slip in): wait for them once */
        s.wide = true;
        QZ_CHECK(hipDeviceSynchronize(), "hipDeviceSynchronize(before the first service of levels 3-4)");
    }
    if (!s.hs) {
        hipDeviceProp_t prop;
        QZ_CHECK(hipGetDeviceProperties(&prop, phys(device)), "hipGetDeviceProperties");
        /* one worker per CU; replicas of one physical device (QZSTD_HIP_REPLICATE_DEVICES, test only) share its CUs between their services */
        s.workers = cfg.workers > 0 ? cfg.workers : (prop.multiProcessorCount / g_devReplicas > 0 ? prop.multiProcessorCount / g_devReplicas : 1);
        if (s.workers > 1024) s.workers = 1024;
        void *h = nullptr, *d = nullptr;
        /* the request ring the dispatcher polls and the callers write: on the GPU's own NUMA node (QZSTD_HIP_NUMA=0: wherever) */
        {
            const char *nm = getenv("QZSTD_HIP_NUMA");
            h = qzstd_hip_host_alloc_on_node(sizeof(SvcHost), (nm && atoi(nm) == 0) ? -1 : qzstd_hip_device_numa_node(device), 1);
            if (!h) return fail_msg("service: no pinned memory for the request ring");
            QZ_SET_DEVICE(device); /* (the node query may have looked at other devices) */
        }
        memset(h, 0, sizeof(SvcHost));
        /* streams of the highest priority: hardware queues of their own, so that neither kernel is serialised behind a batch
         * kernel that happens to share a queue with it */
        int prLeast = 0, prGreatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest);
        if (hipMalloc(&d, sizeof(SvcDev)) != hipSuccess || hipStreamCreateWithPriority(&s.sWork, hipStreamNonBlocking, prGreatest) != hipSuccess ||
            hipStreamCreateWithPriority(&s.sDisp, hipStreamNonBlocking, prGreatest) != hipSuccess || hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (d) (void)hipFree(d);
            (void)hipHostFree(h);
            s.broken = 1;
            return fail_msg("service: set-up failed");
        }
        s.hs = static_cast<SvcHost *>(h);
        s.dv = static_cast<SvcDev *>(d);
        /* a process that ends without QZSTD_stopQatDevice() must not leave kernels polling memory that is about to go:
         * registered after the HIP runtime's own handlers, so it runs before them */
        static std::once_flag once;
        std::call_once(once, [] {
            atexit([] {
                for (int dvc = 0; dvc < 64; dvc++) {
                    Service &sv = g_svc[dvc];
                    if (!sv.hs) continue;
                    std::lock_guard<std::mutex> g(sv.mu);
                    (void)svc_stop_locked(sv, 500);
                }
            });
        });
    }
    /* per-function attribute, once per variant and device would do; cheap enough to repeat on the (rare) launches */
    QZ_CHECK(hipFuncSetAttribute(worker, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(service worker)");
    /* the queue is cleared, and the previous service's kernels (which have said good-bye) are off their streams, before the new
     * ones go out — by waiting here, not by an event between the streams: a cross-stream dependency that resolves against
     * "the last command of the other stream" would tie the dispatcher to the workers, which never finish */
    QZ_CHECK(hipStreamSynchronize(s.sDisp), "hipStreamSynchronize(service dispatcher stream)");
    QZ_CHECK(hipMemsetAsync(s.dv, 0, sizeof(SvcDev), s.sWork), "hipMemsetAsync(service queue)");
    QZ_CHECK(hipStreamSynchronize(s.sWork), "hipStreamSynchronize(service worker stream)");
    __atomic_store_n(&s.hs->quitReq, 0u, __ATOMIC_RELEASE);
    __atomic_store_n(&s.hs->dbg[3], 0ull, __ATOMIC_RELAXED); /* "workers started", refreshed by the dispatcher */
    s.sawWorkers = false;
    {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        __atomic_store_n(&s.launchNs, (long long)t.tv_sec * 1000000000ll + t.tv_nsec, __ATOMIC_RELAXED); /* (before `state` says "running") */
    }
    if (!s.dProfs && hipMalloc(reinterpret_cast<void **>(&s.dProfs), sizeof(profs)) != hipSuccess) { (void)hipGetLastError(); s.dProfs = nullptr; return fail_msg("service: no memory for the profile table"); }
    QZ_CHECK(hipMemcpyAsync(s.dProfs, profs, sizeof(profs), hipMemcpyHostToDevice, s.sWork), "hipMemcpyAsync(service profiles)");
    QZ_CHECK(hipStreamSynchronize(s.sWork), "hipStreamSynchronize(service worker stream)");
    __atomic_store_n(&s.hs->state, 1u, __ATOMIC_RELEASE);
    s.level = level;
    s.served = served;
    s.lds = lds;
    SvcHost *hsDev = nullptr;
    QZ_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&hsDev), s.hs, 0), "hipHostGetDevicePointer");
    uint32_t ctlOff = (uint32_t)(lds - 96u - kLdsBase), spin = (uint32_t)cfg.spinLimit;
    void *wargs[5] = { &a, &s.dv, &ctlOff, &spin, &s.dProfs };
    uint32_t idle = (uint32_t)cfg.idleUs, drain = 1000000u;
    void *dargs[5] = { &hsDev, &s.dv, &served, &idle, &drain };
    /* the dispatcher first: workers without one would never be told to leave */
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(qzstd_service_dispatcher), dim3(1), dim3(64), dargs, 0, s.sDisp);
    if (e != hipSuccess) {
        __atomic_store_n(&s.hs->state, 0u, __ATOMIC_RELEASE);
        s.broken = 1;
        return fail("launch of the service dispatcher", e);
    }
    e = hipLaunchKernel(worker, dim3((unsigned)s.workers), dim3(kThreads), wargs, lds, s.sWork);
    if (e != hipSuccess) {
        __atomic_store_n(&s.hs->quitReq, 1u, __ATOMIC_RELEASE); /* the dispatcher leaves by itself and says so */
        s.broken = 1;
        return fail("launch of the service workers", e);
    }
    s.launches++;
    return 0;
}

} // namespace

extern "C" {

int qzstd_hip_service_submit(int device, int level, const qzstd_hip_svc_req_t *r)
{
    const SvcConfig &cfg = svc_config();
    if (!cfg.enabled || g_svcFreeze.load() > 0 || device < 0 || device >= 64 || phys(device) < 0 || !r) return 1;
    Service &s = g_svc[device];
    if (s.broken) return 1;
    {
        qzstd_hip_profile_t p;
        if (qzstd_hip_profile_for_level(level, QZSTD_HIP_BLOCK_MAX, &p) || !svc_worker_variant(p)) return 1;
        if (p.chainDepth && !r->dWork) return fail_msg("qzstd_hip_service_submit: the chain levels need dWork");
    }
    if (r->nItems < 1u || r->nItems > kSvcMaxItems || r->slot >= kSvcSlots || r->srcLen == 0u || r->srcLen > QZSTD_HIP_BLOCK_MAX ||
        r->itemBytes == 0u || (size_t)(r->nItems - 1u) * r->itemBytes >= r->srcLen || (r->itemBytes & 15u) || !r->hSrc || !r->dSrc ||
        !r->hSeqs || !r->hCount || r->seqCapPerItem < 4u || r->seqCapPerItem > 0xFFFFFFu)
        return fail_msg("qzstd_hip_service_submit: bad request");
    /* the service of this device: running for this level, or stopped (then it is launched for it) */
    if (!s.hs || __atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE) == 0u) {
        std::lock_guard<std::mutex> g(s.mu);
        if (s.broken || g_svcFreeze.load() > 0) return 1;
        if (!s.hs || __atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE) == 0u) {
            if (svc_launch_locked(device, s, level) != 0) return 1;
        }
    }
    {   /* a level the resident workers do not serve (levels 3-4 beside the multi-level worker, or the other way round): launch path */
        const int ix = svc_level_index(level);
        if (ix < 0 || !((s.served >> ix) & 1u)) { s.refused++; return 1; }
    }
    void *dvSrc = nullptr, *dvSeqs = nullptr, *dvCount = nullptr;
    if (hipHostGetDevicePointer(&dvSrc, const_cast<void *>(r->hSrc), 0) != hipSuccess || hipHostGetDevicePointer(&dvSeqs, r->hSeqs, 0) != hipSuccess ||
        hipHostGetDevicePointer(&dvCount, r->hCount, 0) != hipSuccess) {
        (void)hipGetLastError();
        return fail_msg("qzstd_hip_service_submit: the buffers are not pinned host memory");
    }
    const unsigned long long n = s.reserve.fetch_add(1);
    {   /* room in the ring: the request kSvcRing before this one has been taken */
        struct timespec t0, t;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        unsigned spins = 0;
        while (n - __atomic_load_n(&s.hs->consumed, __ATOMIC_ACQUIRE) >= kSvcRing) {
            if ((++spins & 1023u) == 0u) {
                clock_gettime(CLOCK_MONOTONIC, &t);
                if (t.tv_sec - t0.tv_sec >= 2) { s.broken = 1; return fail_msg("qzstd_hip_service_submit: the request ring stays full"); }
                sched_yield();
            }
        }
    }
    const u64 tg = ((n / kSvcRing) % 255ull + 1ull) << 56;
    const u64 lv = (u64)((unsigned)level & 0xFFu) | ((level & QZSTD_HIP_LEVEL_REPCODES) ? 0x80ull : 0ull);
    volatile u64 *e = s.hs->ring[n & (kSvcRing - 1u)];
    e[0] = tg | (u64)(uintptr_t)dvSrc;
    e[1] = tg | (u64)(uintptr_t)r->dSrc;
    e[2] = tg | (u64)(uintptr_t)dvSeqs;
    e[3] = tg | (u64)(uintptr_t)dvCount;
    e[4] = tg | (u64)r->srcLen | ((u64)r->itemBytes << 18) | ((u64)r->nItems << 36) | ((u64)r->slot << 42);
    e[5] = tg | (u64)r->seqCapPerItem;
    e[6] = tg | (u64)(r->epoch & 0xFFFFFFu) | (lv << 24);
    e[7] = tg | (u64)(uintptr_t)r->dWork;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    /* is anybody there to take it?  2 = the dispatcher is deciding whether to leave: wait for its verdict (microseconds) */
    uint32_t st;
    unsigned spins = 0;
    while ((st = __atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE)) == 2u && ++spins < (1u << 26)) {}
    if (st != 1u) {
        /* the service left between the first look and the write (idle exit, a free, a launch that needs the LDS): it takes the
         * ring up where it left it as soon as it is launched again — now if possible, else by the waiting caller's pokes
         * (qzstd_hip_service_poke): a request that is in the ring is never abandoned */
        std::lock_guard<std::mutex> g(s.mu);
        if (__atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE) == 0u && !s.broken && g_svcFreeze.load() == 0) (void)svc_launch_locked(device, s, level);
    }
    __atomic_fetch_add(&s.requests, 1ul, __ATOMIC_RELAXED);
    return 0;
}

int qzstd_hip_service_progressive(int device)
{
    /* the resident workers wait for a count word to leave QZSTD_HIP_NSEQ_STAGING before they read its slice (qzstd_service_worker);
     * QZSTD_HIP_SERVICE_EARLY=0: the callers stage first, as in rounds 3-4 (A/B) */
    static int on = -1;
    (void)device;
    if (on < 0) { const char *v = getenv("QZSTD_HIP_SERVICE_EARLY"); on = !(v && *v && atoi(v) == 0); }
    return on;
}

int qzstd_hip_service_poke(int device, int level)
{
    if (device < 0 || device >= 64) return -1;
    Service &s = g_svc[device];
    if (!s.hs) return 1;
    if (__atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE) != 0u) {
        /* resident — but do its workers run?  Another PROCESS's resident kernels may hold the LDS of every CU (nothing in this one
         * can ask them to leave): a service whose dispatcher has seen no worker begin 200 ms after the launch is taken out of use at
         * once, instead of after the callers' full time-out */
        if (!s.sawWorkers) {
            if (__atomic_load_n(&s.hs->dbg[3], __ATOMIC_RELAXED) != 0ull) s.sawWorkers = true;
            else {
                struct timespec t;
                clock_gettime(CLOCK_MONOTONIC, &t);
                if ((long long)t.tv_sec * 1000000000ll + t.tv_nsec - s.launchNs > 200000000ll) {
                    s.broken = 1;
                    __atomic_store_n(&s.hs->quitReq, 1u, __ATOMIC_RELEASE);
                    return 2;
                }
            }
        }
        return 0;
    }
    if (s.broken) return 2; /* out of use and gone: what it has not answered by now it never will */
    if (g_svcFreeze.load() > 0) return 1;
    std::lock_guard<std::mutex> g(s.mu);
    if (__atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE) != 0u) return 0;
    if (__atomic_load_n(&s.hs->consumed, __ATOMIC_ACQUIRE) >= s.reserve.load()) return 0; /* nothing waits in the ring */
    return svc_launch_locked(device, s, level) == 0 ? 0 : 1;
}

int qzstd_hip_service_stop(int device)
{
    if (device < 0 || device >= 64) return -1;
    Service &s = g_svc[device];
    std::lock_guard<std::mutex> g(s.mu);
    if (!s.hs) return 0;
    const int r = svc_stop_locked(s, 2000);
    if (r == 0 && phys(device) >= 0 && hipSetDevice(phys(device)) == hipSuccess) {
        /* the kernels have said good-bye; their streams follow within microseconds */
        (void)hipStreamSynchronize(s.sWork);
        (void)hipStreamSynchronize(s.sDisp);
    }
    /* the device layer is being shut down: whoever wrote a request that was never taken has given up on it */
    __atomic_store_n(&s.hs->consumed, (u64)s.reserve.load(), __ATOMIC_RELEASE);
    return r;
}

void qzstd_hip_service_mark_broken(int device)
{
    if (device < 0 || device >= 64) return;
    Service &s = g_svc[device];
    s.broken = 1;
    if (s.hs) __atomic_store_n(&s.hs->quitReq, 1u, __ATOMIC_RELEASE);
}

int qzstd_hip_service_info(int device, unsigned long out[8])
{
    if (device < 0 || device >= 64 || !out) return -1;
    Service &s = g_svc[device];
    for (int k = 0; k < 8; k++) out[k] = 0;
    out[0] = s.launches;
    out[1] = s.requests;
    out[2] = s.refused;
    out[3] = (unsigned long)s.broken;
    if (s.hs) {
        out[4] = (unsigned long)__atomic_load_n(&s.hs->state, __ATOMIC_ACQUIRE);
        out[5] = (unsigned long)s.hs->itemsDone;
        out[6] = (unsigned long)s.hs->spinFails;
        out[7] = (unsigned long)s.workers;
    }
    return 0;
}

int qzstd_hip_service_debug(int device, unsigned long out[8])
{
    if (device < 0 || device >= 64 || !out) return -1;
    Service &s = g_svc[device];
    for (int k = 0; k < 8; k++) out[k] = 0;
    if (!s.hs) return 0;
    for (int k = 0; k < 6; k++) out[k] = (unsigned long)__atomic_load_n(&s.hs->dbg[k], __ATOMIC_RELAXED);
    out[6] = (unsigned long)__atomic_load_n(&s.hs->consumed, __ATOMIC_RELAXED);
    out[7] = (unsigned long)s.reserve.load() | ((unsigned long)__atomic_load_n(&s.hs->quitReq, __ATOMIC_RELAXED) << 62) |
             ((unsigned long)(g_svcFreeze.load() > 0) << 61);
    return 0;
}

} /* extern "C" */
