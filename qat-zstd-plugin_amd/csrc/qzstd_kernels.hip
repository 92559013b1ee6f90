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
 *   - the block's bytes are staged once from HBM into LDS with 16-byte coalesced loads;
 *   - a 4-byte-entry hash table ((position+1)<<14 | 14-bit tag) lives in LDS next to it;
 *   - positions are processed in tiles of 1<<tileLog: every position of a tile reads its
 *     slot (newest position of EARLIER tiles), then all insert with ds_max_u32, so the
 *     result does not depend on wave scheduling; a tile-local ds_min_u32 table finds
 *     sources inside the current tile;
 *   - candidate lengths are measured from LDS (capped), packed per position, and a
 *     dedicated wave runs the (lazy) greedy parse over 64-position windows with
 *     ballot / readlane, extends long matches cooperatively, and the chosen lanes emit
 *     their {offset, litLength, matchLength} entries with a popcount prefix rank.
 *
 * Integer byte matching: no MFMA.  The roofline that bounds it is HBM (block read once,
 * 16 B per sequence written); in practice it is LDS-latency / occupancy bound
 * (one 128 KiB block + table = the CU's whole 160 KiB LDS).
 *
 * The sequential definition of exactly this computation is oracle/qzstd_oracle.c
 * (test infrastructure); tests compare the two sequence-for-sequence.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qzstd_hip.h"

namespace {

constexpr int kMatchWaves = 8;
constexpr int kMatchThreads = kMatchWaves * 64; /* one position per matcher thread per tile */
constexpr int kThreads = kMatchThreads + 64;     /* + 1 parse wave */
constexpr uint32_t kTileLog = 9;                 /* tile = 512 positions = kMatchThreads */
constexpr uint32_t kTile = 1u << kTileLog;
constexpr uint32_t kWin = kTile >> 6;            /* 64-position windows per tile (one per matcher wave) */
constexpr uint32_t kGroups = kTile >> 4;         /* 16-position groups per tile: one parse chain each */
constexpr uint32_t kSlots = 3;                   /* tiles in flight between matchers and the parse wave */
constexpr uint32_t kTagBits = 14;
constexpr uint32_t kTagMask = (1u << kTagBits) - 1u;
constexpr uint32_t kPrime1 = 2654435761u;
constexpr uint32_t kPrime2 = 0x85EBCA77u;
constexpr uint32_t kNone = 0xFFFFFFFFu;
/* per-window parse record (8 words): start mask, stop mask, chosen mask, last chosen end, max reach */
enum { W_START = 0, W_STOP = 2, W_CHOSEN = 4, W_LASTEND = 6, W_REACH = 7, W_WORDS = 8 };

struct LaunchArgs {
    const uint8_t *src;
    const qzstd_hip_block_t *blocks;
    uint4 *seqs; /* ZSTD_Sequence = 4 x u32 */
    uint32_t *nseq;
    qzstd_hip_profile_t prof[3]; /* by block size class: >64 KiB, >32 KiB, <=32 KiB */
};

typedef unsigned long long u64;

/* v_readlane_b32 with an unsigned result (the builtin returns int: a set bit 31 would sign-extend) */
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u64 below(uint32_t c) { return c >= 64u ? ~0ull : ((1ull << c) - 1ull); }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

/* 4 bytes at an arbitrary LDS byte address: two aligned dword reads + v_alignbyte_b32 */
__device__ __forceinline__ uint32_t lds_rd32u(const uint32_t *lds32, uint32_t a)
{
    const uint32_t d = a >> 2;
    return __builtin_amdgcn_alignbyte(lds32[d + 1], lds32[d], a & 3u);
}

/*
 * Common prefix length of [p..) and [q..), capped at cap (<= 128).  Works in 32-byte
 * chunks: 9 aligned dwords per side are fetched with independent ds_reads (one LDS
 * round trip per chunk instead of one per dword), then compared in registers.
 * `own` holds the 9 dwords of the position's own first chunk (shared by both candidates).
 */
__device__ __forceinline__ uint32_t match_len(const uint32_t *lds32, const uint32_t (&own)[9], uint32_t p,
                                              uint32_t q, uint32_t cap)
{
    const uint32_t ps = p & 3u, qs = q & 3u;
    uint32_t L = cap;
    {
        uint32_t Q[9];
        const uint32_t qd = q >> 2;
#pragma unroll
        for (int i = 0; i < 9; i++) Q[i] = lds32[qd + i];
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            const uint32_t x = __builtin_amdgcn_alignbyte(own[i + 1], own[i], ps) ^
                               __builtin_amdgcn_alignbyte(Q[i + 1], Q[i], qs);
            if (x) L = 4u * (uint32_t)i + ((uint32_t)__builtin_ctz(x) >> 3);
        }
    }
    if (L >= 32u && cap > 32u) { /* rare: long candidate, level >= 6 caps */
        L = 32u;
        while (L < cap) {
            const uint32_t x = lds_rd32u(lds32, p + L) ^ lds_rd32u(lds32, q + L);
            if (x) { L += (uint32_t)__builtin_ctz(x) >> 3; break; }
            L += 4u;
        }
    }
    return L < cap ? L : cap;
}

__device__ __forceinline__ uint32_t min_len(const qzstd_hip_profile_t &pf, uint32_t off)
{
    return pf.minMatch + ((off >> pf.farLog1) ? 1u : 0u) + ((off >> pf.farLog2) ? 1u : 0u);
}

/* cooperative forward extension of a match that hit the candidate cap: 64 lanes x 4 bytes
 * per step, never past `lim` */
__device__ __forceinline__ uint32_t extend_match(const uint32_t *lds32, uint32_t p, uint32_t off, uint32_t L,
                                                 uint32_t lim, uint32_t lane)
{
    for (;;) {
        const uint32_t a = p + L + 4u * lane;
        uint32_t ok = 0; /* bytes of this lane's dword that match and lie below lim */
        if (a < lim) {
            const uint32_t x = lds_rd32u(lds32, a) ^ lds_rd32u(lds32, a - off);
            ok = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
            ok = umin(ok, lim - a);
        }
        const u64 bad = __ballot(ok < 4u);
        if (bad) {
            const uint32_t f = (uint32_t)__builtin_ctzll(bad);
            return L + 4u * f + rdlane(ok, f);
        }
        L += 256u;
    }
}

/* inclusive max-scan across the 64 lanes of a wave */
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x, uint32_t lane)
{
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t t = __shfl_up(x, d);
        if (lane >= d) x = umax(x, t);
    }
    return x;
}

/* parse-wave state, uniform across the wave */
struct ParseState {
    uint32_t anchor;  /* end of the last chosen match = start of pending literals */
    uint32_t nseq;    /* matches chosen so far */
    uint32_t pending; /* standing position a chain reached beyond the data available to its pass */
};

/*
 * The parse of one tile, by the parse wave, all chains of the tile in parallel.
 *
 * A position is a SYNC point when no potential match start before it reaches beyond it; the
 * (lazy) greedy parse provably stands on every sync point, so the stretches between sync
 * points can be parsed independently.  One lane per 16-position group starts at the group's
 * first sync point ("stop point") and follows next-start / jump-by-length until it stands on
 * another stop point; lane 32 continues a chain that an earlier pass had to suspend.  Chosen
 * starts are OR-ed into the per-window masks (ds_or), match ends MAX-ed (ds_max).  Then the
 * per-window sequence index bases and literal anchors of tile k are prefix-summed and
 * published for the emitting waves.
 */
__device__ void parse_pass(uint32_t *wrec, const uint16_t *lens, uint32_t *srecOut, uint32_t k, uint32_t nTiles,
                           uint32_t lane, ParseState &st)
{
    const uint32_t base = k << kTileLog;
    const uint32_t limit = umin(k + 2u, nTiles) << kTileLog; /* tiles k and k+1 are complete */
    const uint32_t slotK = k % kSlots;
    uint32_t c = 0;
    bool active = false, first = true;
    if (lane < kGroups) {
        const uint32_t w = lane >> 2, sub = lane & 3u;
        const uint32_t *r = wrec + (slotK * kWin + w) * W_WORDS;
        const uint32_t bits = ((sub & 2u ? r[W_STOP + 1] : r[W_STOP]) >> (16u * (sub & 1u))) & 0xFFFFu;
        if (bits) { active = true; c = base + 64u * w + 16u * sub + (uint32_t)__builtin_ctz(bits); }
    } else if (lane == kGroups && st.pending != kNone && (st.pending >> kTileLog) == k) {
        active = true; first = false; c = st.pending;
    }
    if (st.pending != kNone && (st.pending >> kTileLog) == k) st.pending = kNone;
    bool suspended = false;
    while (__any(active)) {
        if (active) {
            const uint32_t tile = c >> kTileLog, w = (c >> 6) & (kWin - 1u), rel = c & 63u;
            uint32_t *r = wrec + ((tile % kSlots) * kWin + w) * W_WORDS;
            const u64 start = (u64)r[W_START] | ((u64)r[W_START + 1] << 32);
            const u64 stop = (u64)r[W_STOP] | ((u64)r[W_STOP + 1] << 32);
            const u64 ms = start >> rel, ss = stop >> rel;
            if (!first && (ss & 1ull)) {
                active = false; /* standing on another chain's starting point */
            } else {
                const u64 ss1 = ss & ~1ull;
                const uint32_t jrel = ms ? (uint32_t)__builtin_ctzll(ms) : 64u;
                const uint32_t srel = ss1 ? (uint32_t)__builtin_ctzll(ss1) : 64u;
                if (srel <= jrel && srel != 64u) {
                    active = false; /* literals up to the next stop point */
                } else if (jrel == 64u) {
                    c = (c | 63u) + 1u; /* no start left in this window: walk into the next one */
                    if (c >= limit) { active = false; suspended = true; }
                } else {
                    const uint32_t j = c + jrel, jpos = j & 63u;
                    const uint32_t L = lens[(tile % kSlots) * kTile + (j & (kTile - 1u))];
                    atomicOr(&r[W_CHOSEN + (jpos >> 5)], 1u << (jpos & 31u));
                    atomicMax(&r[W_LASTEND], j + L);
                    c = j + L;
                    if (c >= limit) { active = false; suspended = true; }
                }
                first = false;
            }
        }
    }
    {
        const u64 sm = __ballot(suspended);
        if (sm) st.pending = rdlane(c, 63u - (uint32_t)__builtin_clzll(sm));
    }
    /* finalize tile k: lanes 0..kWin-1 = windows */
    {
        const uint32_t w = lane < kWin ? lane : 0u;
        const uint32_t *r = wrec + (slotK * kWin + w) * W_WORDS;
        const uint32_t cLo = lane < kWin ? r[W_CHOSEN] : 0u, cHi = lane < kWin ? r[W_CHOSEN + 1] : 0u;
        const uint32_t le = lane < kWin ? r[W_LASTEND] : 0u;
        const uint32_t cnt = (uint32_t)__popc(cLo) + (uint32_t)__popc(cHi);
        uint32_t ps = cnt, pm = le; /* inclusive prefix sum / max over the kWin window lanes */
#pragma unroll
        for (uint32_t d = 1; d < kWin; d <<= 1) {
            const uint32_t a = __shfl_up(ps, d), b = __shfl_up(pm, d);
            if (lane >= d) { ps += a; pm = umax(pm, b); }
        }
        const uint32_t exm = __shfl_up(pm, 1);
        if (lane < kWin) {
            uint4 o;
            o.x = cLo; o.y = cHi;
            o.z = umax(st.anchor, lane ? exm : 0u); /* anchor when the parse enters the window */
            o.w = st.nseq + ps - cnt;               /* index of the window's first sequence */
            reinterpret_cast<uint4 *>(srecOut)[lane] = o;
        }
        st.nseq += rdlane(ps, kWin - 1u);
        st.anchor = umax(st.anchor, rdlane(pm, kWin - 1u));
    }
}

/* emission of one window's chosen matches by the wave that owns the window */
__device__ __forceinline__ void emit_window(const qzstd_hip_profile_t &pf, const uint32_t *lds32, const uint32_t *srec,
                                            uint32_t off, uint32_t len, uint32_t w0, uint32_t lane, uint4 *out,
                                            uint32_t seqCap)
{
    const uint4 rec = *reinterpret_cast<const uint4 *>(srec);
    const u64 chosen = (u64)rec.x | ((u64)rec.y << 32);
    if (!chosen) return;
    const uint32_t anchorIn = rec.z, seqBase = rec.w;
    const bool ch = (chosen >> lane) & 1ull;
    const u64 lower = chosen & below(lane);
    const uint32_t rank = (uint32_t)__popcll(lower);
    const uint32_t myEnd = w0 + lane + len;
    const int jprev = lower ? 63 - __builtin_clzll(lower) : 0;
    uint32_t prevEnd = __shfl(myEnd, jprev);
    if (!lower) prevEnd = anchorIn;
    if (ch) {
        const uint32_t p = w0 + lane, q = p - off;
        const uint32_t lit = p - prevEnd;
        uint32_t maxb = umin(umin(pf.backExt, lit), q);
        uint32_t b = 0;
        if (maxb) {
            /* the 4 bytes before p and before q, top byte = nearest; count equal bytes from the top */
            const uint32_t pb = p >= 4u ? lds_rd32u(lds32, p - 4u) : lds32[0] << (8u * (4u - p));
            const uint32_t qb = q >= 4u ? lds_rd32u(lds32, q - 4u) : lds32[0] << (8u * (4u - q));
            const uint32_t x = pb ^ qb;
            b = umin(x ? (uint32_t)__builtin_clz(x) >> 3 : 4u, maxb);
        }
        const uint32_t idx = seqBase + rank;
        if (idx < seqCap) out[idx] = make_uint4(off, lit - b, len + b, 0u);
    }
}

/*
 * One workgroup = one block.  8 matcher waves (one position per thread per 512-position tile)
 * + 1 parse wave, software-pipelined over tiles with two barriers per iteration:
 *
 *   interval 1 of iteration it   matchers: sync/stop masks of tile it-1, phase A(it) (table look-up)
 *   barrier
 *   interval 2                   matchers: emit(it-3), phase B(it) (insert), candidate lengths,
 *                                          run extension, start flags, reach of tile it
 *                                parse wave: chains + prefix sums of tile it-2
 *   barrier
 */
__global__ __launch_bounds__(kThreads) void qzstd_find_sequences_kernel(LaunchArgs args)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const bool matcher = tid < (uint32_t)kMatchThreads;
    const qzstd_hip_block_t blk = args.blocks[blockIdx.x];
    const uint32_t n = blk.srcLen;
    const qzstd_hip_profile_t pf = args.prof[n > (64u << 10) ? 0 : (n > (32u << 10) ? 1 : 2)];
    const uint32_t nh = n >= pf.hashBytes ? n - pf.hashBytes + 1u : 0u; /* hashable positions */
    const uint32_t nTiles = (nh + kTile - 1u) >> kTileLog;

    /* ---- LDS layout for THIS block ---- */
    const uint32_t region = ((n + 15u) & ~15u) + 16u;
    uint8_t *lds8 = smem;
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *tbl = reinterpret_cast<uint32_t *>(smem + region);
    uint32_t *nearTab = tbl + pf.tableSize;
    uint32_t *wrec = nearTab + kTile;                                  /* [kSlots][kWin][8]          */
    uint32_t *srec = wrec + kSlots * kWin * W_WORDS;                   /* [2][kWin][4]               */
    uint16_t *lens = reinterpret_cast<uint16_t *>(srec + 2u * kWin * 4u); /* [kSlots][kTile] u16     */

    /* ---- stage the block: HBM -> LDS, 16 B per lane, coalesced ---- */
    {
        const uint8_t *g = args.src + blk.srcOff;
        const uint32_t nvec = n >> 4;
        const uint4 *g4 = reinterpret_cast<const uint4 *>(g);
        uint4 *l4 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = tid; i < nvec; i += kThreads) l4[i] = g4[i];
        for (uint32_t i = (nvec << 4) + tid; i < region; i += kThreads) lds8[i] = i < n ? g[i] : (uint8_t)0;
        for (uint32_t i = tid; i < pf.tableSize; i += kThreads) tbl[i] = 0u;
        for (uint32_t i = tid; i < kTile; i += kThreads) nearTab[i] = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < kSlots * kWin * W_WORDS + 2u * kWin * 4u; i += kThreads) wrec[i] = 0u;
        for (uint32_t i = tid; i < kSlots * kTile / 2u; i += kThreads) reinterpret_cast<uint32_t *>(lens)[i] = 0u;
    }
    __syncthreads();

    ParseState st = { 0u, 0u, kNone };
    uint4 *out = args.seqs + blk.seqOff;
    const uint32_t hiMask = pf.hashBytes >= 8 ? 0xFFFFFFFFu : ((1u << (8u * (pf.hashBytes - 4u))) - 1u);
    const uint32_t nearShift = 32u - kTileLog;
    const uint32_t stampShift = kTileLog + kTagBits;
    const uint32_t nTilesMax = QZSTD_HIP_BLOCK_MAX >> kTileLog;

    /* matcher registers: (offset, full length) of the four most recent tiles of this position slot */
    uint32_t offG0 = 0, offG1 = 0, offG2 = 0, offG3 = 0; /* G0 = tile it (after interval 2), G3 = tile it-3 */
    uint32_t lenG0 = 0, lenG1 = 0, lenG2 = 0, lenG3 = 0;
    uint32_t exReach = 0;  /* exclusive in-window prefix max of reach, tile it-1 (for the sync phase) */
    u64 startMask = 0;     /* start flags of the own window, tile it-1 */
    uint32_t tileCarry = 0; /* max reach of all tiles before it-1 */

    for (uint32_t it = 0; it < nTiles + 3u; it++) {
        const uint32_t t0 = it << kTileLog;
        const uint32_t p = t0 + tid; /* matcher: own position in tile it */
        const uint32_t stamp = (nTilesMax - 1u - (it & (nTilesMax - 1u))) << stampShift;
        uint32_t v = 0, mix = 0, old = 0;
        const bool valid = matcher && it < nTiles && p < nh;

        /* ================= interval 1 ================= */
        if (matcher) {
            if (it >= 1u && it - 1u < nTiles) {
                /* sync / stop masks of tile it-1: a position is a sync point when no start before it
                 * (in the whole block) reaches beyond it */
                const uint32_t slot = (it - 1u) % kSlots;
                uint32_t *rw = wrec + (slot * kWin) * W_WORDS;
                const uint32_t wm = lane < kWin ? rw[lane * W_WORDS + W_REACH] : 0u;
                uint32_t carry = tileCarry, all = tileCarry;
#pragma unroll
                for (uint32_t w = 0; w < kWin; w++) {
                    const uint32_t x = rdlane(wm, w);
                    if (w < wave) carry = umax(carry, x);
                    all = umax(all, x);
                }
                tileCarry = all;
                const uint32_t pos = t0 - kTile + tid;
                const bool sync = umax(carry, exReach) <= pos;
                const u64 sm = __ballot(sync);
                /* stop point = first sync point of each 16-position group */
                const u64 grp = 0xFFFFull << (lane & 48u);
                const bool stopb = sync && (sm & grp & below(lane)) == 0ull;
                const u64 stm = __ballot(stopb);
                uint32_t val = 0;
                val = lane == 0 ? (uint32_t)startMask : val;
                val = lane == 1 ? (uint32_t)(startMask >> 32) : val;
                val = lane == 2 ? (uint32_t)stm : val;
                val = lane == 3 ? (uint32_t)(stm >> 32) : val;
                if (lane < 7u) rw[wave * W_WORDS + lane] = val; /* words 4..6 (chosen, lastEnd) := 0 */
            }
            if (valid) { /* phase A(it) */
                const uint32_t d = p >> 2, s = p & 3u;
                const uint32_t w0 = lds32[d], w1 = lds32[d + 1];
                v = __builtin_amdgcn_alignbyte(w1, w0, s);
                uint32_t hi = 0;
                if (pf.hashBytes > 4) hi = __builtin_amdgcn_alignbyte(lds32[d + 2], w1, s) & hiMask;
                mix = (v * kPrime1) ^ (hi * kPrime2);
                old = tbl[__umulhi(mix, pf.tableSize)];
                if (pf.nearTab)
                    atomicMin(&nearTab[mix >> nearShift], stamp | (tid << kTagBits) | ((mix >> 3) & kTagMask));
            }
        }
        __syncthreads(); /* B1 */

        /* ================= interval 2 ================= */
        if (matcher) {
            if (it >= 3u && it - 3u < nTiles) /* emit(it-3) */
                emit_window(pf, lds32, srec + (((it - 3u) & 1u) * kWin + wave) * 4u, offG2, lenG2,
                            t0 - 3u * kTile + 64u * wave, lane, out, blk.seqCap);
            offG3 = offG2; offG2 = offG1; offG1 = offG0; lenG3 = lenG2; lenG2 = lenG1; lenG1 = lenG0;
            (void)offG3; (void)lenG3;
            uint32_t cl = 0, off = 0; /* capped candidate length, offset */
            if (valid) {
                const uint32_t tag = (mix >> 3) & kTagMask;
                const uint32_t en = pf.nearTab ? nearTab[mix >> nearShift] : 0xFFFFFFFFu;
                atomicMax(&tbl[__umulhi(mix, pf.tableSize)], ((p + 1u) << kTagBits) | tag);
                const uint32_t cap = umin(pf.capLen, n - p);
                /* candidate 1: newest position of earlier tiles; candidate 2: earliest of this tile */
                uint32_t q1 = kNone, q2 = kNone;
                if (old != 0u && (old & kTagMask) == tag) {
                    const uint32_t q = (old >> kTagBits) - 1u;
                    if (pf.window == 0u || p - q <= pf.window) q1 = q;
                }
                if (pf.nearTab && (en >> stampShift) == (stamp >> stampShift) && (en & kTagMask) == tag) {
                    const uint32_t q = t0 + ((en >> kTagBits) & (kTile - 1u));
                    if (q < p) q2 = q;
                }
                if (q1 != kNone || q2 != kNone) {
                    uint32_t own[9];
                    const uint32_t pd = p >> 2;
#pragma unroll
                    for (int i = 0; i < 9; i++) own[i] = lds32[pd + i];
                    if (q1 != kNone) {
                        const uint32_t l = match_len(lds32, own, p, q1, cap);
                        if (l >= 4u) { cl = l; off = p - q1; }
                    }
                    if (q2 != kNone) {
                        const uint32_t l = match_len(lds32, own, p, q2, cap);
                        if (l >= 4u && l >= cl) { cl = l; off = p - q2; }
                    }
                }
            }
            uint32_t full = cl;
            if (it < nTiles) {
                /* runs of capped candidates with one offset are one long match: its end comes from the
                 * run's tail (visible in the window, or found by a bounded cooperative extension) */
                const uint32_t clN = __shfl_down(cl, 1), offN = __shfl_down(off, 1);
                const bool capped = cl == pf.capLen;
                const bool cont = capped && lane != 63u && clN == pf.capLen && offN == off;
                const bool tail = capped && !cont;
                const u64 tailMask = __ballot(tail);
                const bool visible = tail && lane != 63u && clN != 0u && offN == off;
                uint32_t E = visible ? p + 1u + clN : 0u;
                u64 ext = __ballot(tail && !visible);
                while (ext) {
                    const uint32_t t = (uint32_t)__builtin_ctzll(ext);
                    ext &= ext - 1ull;
                    const uint32_t pt = t0 + 64u * wave + t, ot = rdlane(off, t);
                    const uint32_t lim = umin(n, ((pt >> pf.extLog) + 2u) << pf.extLog);
                    const uint32_t Lt = extend_match(lds32, pt, ot, pf.capLen, lim, lane);
                    if (lane == t) E = pt + Lt;
                }
                if (tailMask) {
                    const u64 mine = tailMask & ~below(lane);
                    const uint32_t tl = mine ? (uint32_t)__builtin_ctzll(mine) : 0u;
                    const uint32_t Et = __shfl(E, (int)tl);
                    if (capped) full = umin(Et, ((p >> pf.extLog) + 2u) << pf.extLog) - p;
                }
                /* start flags: the lazy rule compares capped lengths and never looks across the window edge */
                const bool take = cl != 0u && cl >= min_len(pf, off);
                const bool take1 = clN != 0u && clN >= min_len(pf, offN);
                const bool start = take && !(pf.lazy && lane != 63u && take1 && clN > cl);
                startMask = __ballot(start);
                const uint32_t reach = start ? p + full : 0u;
                const uint32_t inc = wave_scan_max(reach, lane);
                exReach = __shfl_up(inc, 1);
                if (lane == 0u) exReach = 0u;
                const uint32_t slot = it % kSlots;
                lens[slot * kTile + tid] = (uint16_t)full;
                if (lane == 63u) wrec[(slot * kWin + wave) * W_WORDS + W_REACH] = inc;
            }
            offG0 = off;
            lenG0 = full;
        } else if (it >= 2u && it - 2u < nTiles) {
            parse_pass(wrec, lens, srec + ((it - 2u) & 1u) * kWin * 4u, it - 2u, nTiles, lane, st);
        }
        __syncthreads(); /* B2 */
    }

    if (!matcher && lane == 0) {
        /* delimiter {lit = tail, 0, 0}: QZSTD_decLz4s, src/qatseqprod.c:1037-1045 */
        uint32_t count = st.nseq + 1u;
        if (st.nseq < blk.seqCap) out[st.nseq] = make_uint4(0u, n - st.anchor, 0u, 0u);
        if (count >= blk.seqCap - 1u) count = QZSTD_HIP_NSEQ_ERROR; /* src/qatseqprod.c:1318 */
        args.nseq[blockIdx.x] = count;
    }
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

} // namespace

extern "C" {

const char *qzstd_hip_last_error(void) { return g_err; }

int qzstd_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail("hipGetDeviceCount", e); (void)hipGetLastError(); return -1; }
    return n;
}

int qzstd_hip_device_name(int device, char *buf, size_t bufLen)
{
    hipDeviceProp_t prop;
    QZ_CHECK(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (buf && bufLen) snprintf(buf, bufLen, "%s (%s, %d CUs, %zu KiB LDS/WG)", prop.name, prop.gcnArchName,
                                prop.multiProcessorCount, prop.sharedMemPerBlock >> 10);
    return 0;
}

void *qzstd_hip_malloc(int device, size_t bytes)
{
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { fail("hipMalloc", e); return nullptr; }
    return p;
}

void qzstd_hip_free(int device, void *dptr)
{
    if (!dptr) return;
    if (hipSetDevice(device) == hipSuccess) (void)hipFree(dptr);
}

void *qzstd_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) { fail("hipHostMalloc", e); return nullptr; }
    return p;
}

void qzstd_hip_host_free(void *hptr)
{
    if (hptr) (void)hipHostFree(hptr);
}

void *qzstd_hip_stream_create(int device)
{
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { fail("hipStreamCreate", e); return nullptr; }
    return (void *)s;
}

void qzstd_hip_stream_destroy(int device, void *stream)
{
    if (stream && hipSetDevice(device) == hipSuccess) (void)hipStreamDestroy((hipStream_t)stream);
}

int qzstd_hip_stream_sync(int device, void *stream)
{
    QZ_CHECK(hipSetDevice(device), "hipSetDevice");
    QZ_CHECK(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
    return 0;
}

int qzstd_hip_stream_query(int device, void *stream)
{
    QZ_CHECK(hipSetDevice(device), "hipSetDevice");
    hipError_t e = hipStreamQuery((hipStream_t)stream);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) return 1;
    return fail("hipStreamQuery", e);
}

int qzstd_hip_memcpy_h2d(int device, void *stream, void *dst, const void *src, size_t bytes)
{
    QZ_CHECK(hipSetDevice(device), "hipSetDevice");
    QZ_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync H2D");
    return 0;
}

int qzstd_hip_memcpy_d2h(int device, void *stream, void *dst, const void *src, size_t bytes)
{
    QZ_CHECK(hipSetDevice(device), "hipSetDevice");
    QZ_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync D2H");
    return 0;
}

int qzstd_hip_memset(int device, void *stream, void *dst, int value, size_t bytes)
{
    QZ_CHECK(hipSetDevice(device), "hipSetDevice");
    QZ_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream), "hipMemsetAsync");
    return 0;
}

int qzstd_hip_find_sequences(int device, void *stream, int level, const void *d_src,
                             const qzstd_hip_block_t *d_blocks, uint32_t nBlocks, uint32_t maxBlockLen,
                             void *d_seqs, uint32_t *d_nseq)
{
    static thread_local int attrDevice = -1;
    static thread_local size_t attrBytes = 0;
    LaunchArgs a;
    if (nBlocks == 0) return 0;
    if (!d_src || !d_blocks || !d_seqs || !d_nseq) return fail_msg("qzstd_hip_find_sequences: null pointer");
    if (maxBlockLen > QZSTD_HIP_BLOCK_MAX) return fail_msg("qzstd_hip_find_sequences: block larger than 128 KiB");
    if (qzstd_hip_profile_for_level(level, 128u << 10, &a.prof[0]) ||
        qzstd_hip_profile_for_level(level, 64u << 10, &a.prof[1]) ||
        qzstd_hip_profile_for_level(level, 32u << 10, &a.prof[2]))
        return fail_msg("qzstd_hip_find_sequences: level outside 1..12");
    for (int c = 0; c < 3; c++)
        if (a.prof[c].tileLog != kTileLog || a.prof[c].extLog < 8 || a.prof[c].extLog > 15 || a.prof[c].capLen > 128 || a.prof[c].capLen < 32 || a.prof[c].minMatch < 4 || a.prof[c].hashBytes < 4 ||
            a.prof[c].hashBytes > 8)
            return fail_msg("qzstd_hip_find_sequences: unsupported profile");
    const size_t lds = qzstd_hip_lds_bytes(level, maxBlockLen);
    if (lds == 0) return fail_msg("qzstd_hip_find_sequences: LDS budget exceeded");
    QZ_CHECK(hipSetDevice(device), "hipSetDevice");
    if (attrDevice != device || attrBytes < lds) {
        QZ_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(qzstd_find_sequences_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                 "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        attrDevice = device;
        attrBytes = lds;
    }
    a.src = static_cast<const uint8_t *>(d_src);
    a.blocks = d_blocks;
    a.seqs = static_cast<uint4 *>(d_seqs);
    a.nseq = d_nseq;
    hipLaunchKernelGGL(qzstd_find_sequences_kernel, dim3(nBlocks), dim3(kThreads), lds, (hipStream_t)stream, a);
    QZ_CHECK(hipGetLastError(), "launch qzstd_find_sequences_kernel");
    return 0;
}

} /* extern "C" */
