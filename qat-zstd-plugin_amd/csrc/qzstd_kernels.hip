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
constexpr int kMatchThreads = kMatchWaves * 64;
constexpr int kThreads = kMatchThreads + 64; /* + 1 parse wave */
constexpr int kMaxPosPerThread = 2;          /* tileLog <= 10 -> <= 1024 / 512 */
constexpr int kMaxWindows = 16;              /* 64-position windows per tile */
constexpr uint32_t kTagBits = 14;
constexpr uint32_t kTagMask = (1u << kTagBits) - 1u;
constexpr uint32_t kPrime1 = 2654435761u;
constexpr uint32_t kPrime2 = 0x85EBCA77u;
constexpr uint32_t kNone = 0xFFFFFFFFu;
/* LDS words behind the per-position scratch: see layout in the kernel */
constexpr uint32_t kSaveWords = 128;              /* last window of a tile, double buffered */
constexpr uint32_t kRecWords = kMaxWindows * 8;   /* per-window records, two kinds */

struct LaunchArgs {
    const uint8_t *src;
    const qzstd_hip_block_t *blocks;
    uint4 *seqs; /* ZSTD_Sequence = 4 x u32 */
    uint32_t *nseq;
    qzstd_hip_profile_t prof[3]; /* by block size class: >64 KiB, >32 KiB, <=32 KiB */
    uint32_t dbg; /* profiling ablation switches (QZSTD_HIP_ABLATE); 0 in production */
};

typedef unsigned long long u64;

/* v_readlane_b32 with an unsigned result (the builtin returns int: a set bit 31 would sign-extend) */
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

__device__ __forceinline__ u64 below(uint32_t c) { return c >= 64u ? ~0ull : ((1ull << c) - 1ull); }

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

/* which positions of a 64-position window may start a match (the lazy rule never looks
 * across the window edge), and which of those hit the candidate cap */
struct WinFlags { u64 start, capped; };
__device__ __forceinline__ WinFlags window_flags(const qzstd_hip_profile_t &pf, uint32_t r, uint32_t lane)
{
    const uint32_t len = r & 0xFFu, off = r >> 8;
    const uint32_t r1 = __shfl_down(r, 1);
    const uint32_t len1 = r1 & 0xFFu, off1 = r1 >> 8;
    const bool take = len != 0u && len >= min_len(pf, off);
    const bool take1 = len1 != 0u && len1 >= min_len(pf, off1);
    const bool start = take && !(pf.lazy && lane != 63u && take1 && len1 > len);
    WinFlags w;
    w.start = __ballot(start);
    w.capped = __ballot(start && len == pf.capLen);
    return w;
}

/* cooperative forward extension of a chosen match that hit the candidate cap:
 * 64 lanes x 4 bytes per step */
__device__ __forceinline__ uint32_t extend_match(const uint32_t *lds32, uint32_t p, uint32_t off, uint32_t L,
                                                 uint32_t n, uint32_t lane)
{
    for (;;) {
        const uint32_t a = p + L + 4u * lane;
        uint32_t ok = 0; /* bytes of this lane's dword that match and lie inside the block */
        if (a < n) {
            const uint32_t x = lds_rd32u(lds32, a) ^ lds_rd32u(lds32, a - off);
            ok = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
            const uint32_t room = n - a;
            ok = ok < room ? ok : room;
        }
        const u64 bad = __ballot(ok < 4u);
        if (bad) {
            const uint32_t f = (uint32_t)__builtin_ctzll(bad);
            return L + 4u * f + rdlane(ok, f);
        }
        L += 256u;
    }
}

/*
 * Speculative chain of one window (run by the wave that owns the window, all windows
 * of a tile in parallel).  The chain is warmed up over the previous window so that it
 * has (almost always) merged with the true parse before it enters this one.  Lengths
 * are the capped candidate lengths; capped starts on the chain are flagged so that the
 * serial pass can extend them.  Record (8 words): visited mask, chain-start mask,
 * capped-on-chain mask, exit cursor (window relative, >= 64), end of the last chain match.
 */
__device__ __forceinline__ void spec_chain(const qzstd_hip_profile_t &pf, uint32_t rPrev, bool havePrev,
                                           uint32_t r, uint32_t lane, uint32_t *rec)
{
    uint32_t c = 0;
    if (havePrev) {
        const WinFlags fp = window_flags(pf, rPrev, lane);
        const uint32_t lenP = rPrev & 0xFFu;
        for (;;) {
            const u64 m = c < 64u ? (fp.start >> c) << c : 0ull;
            if (!m) { c = 64u; break; }
            const uint32_t j = (uint32_t)__builtin_ctzll(m);
            c = j + rdlane(lenP, j);
            if (c >= 64u) break;
        }
        c -= 64u;
    }
    const WinFlags f = window_flags(pf, r, lane);
    const uint32_t len = r & 0xFFu;
    u64 visited = 0, chain = 0;
    uint32_t lastEnd = kNone, exitC = c;
    while (c < 64u) {
        const u64 m = (f.start >> c) << c;
        if (!m) { visited |= ~below(c); exitC = 64u; break; }
        const uint32_t j = (uint32_t)__builtin_ctzll(m);
        visited |= below(j + 1u) & ~below(c);
        chain |= 1ull << j;
        c = j + rdlane(len, j);
        lastEnd = c;
        exitC = c;
    }
    const u64 cap = chain & f.capped;
    uint32_t v = 0;
    v = lane == 0 ? (uint32_t)visited : v;
    v = lane == 1 ? (uint32_t)(visited >> 32) : v;
    v = lane == 2 ? (uint32_t)chain : v;
    v = lane == 3 ? (uint32_t)(chain >> 32) : v;
    v = lane == 4 ? (uint32_t)cap : v;
    v = lane == 5 ? (uint32_t)(cap >> 32) : v;
    v = lane == 6 ? exitC : v;
    v = lane == 7 ? lastEnd : v;
    if (lane < 8u) rec[lane] = v;
}

/* parse-wave state, uniform across the wave */
struct ParseState {
    uint32_t cur;    /* next position the parse looks at */
    uint32_t anchor; /* end of the last chosen match = start of pending literals */
    uint32_t nseq;   /* matches chosen so far */
};

/*
 * Serial pass over the windows of one tile (parse wave).  O(1) per window when the true
 * cursor lies on the window's speculative chain; otherwise it steps manually (readlane
 * over the preloaded candidates) until it merges.  Capped matches are extended here.
 * Output record per window (8 words): chosen mask, anchor at entry, sequence index base,
 * up to two (lane, extended length) pairs.
 */
__device__ void serial_pass(const qzstd_hip_profile_t &pf, const uint32_t *lds32, const uint32_t (&R)[kMaxWindows],
                            const uint32_t *crec, uint32_t *srec, uint32_t t0, uint32_t nWin, uint32_t n,
                            uint32_t lane, ParseState &st, uint4 *dbg)
{
    const uint32_t c0 = crec[lane], c1 = crec[64u + lane]; /* 16 windows x 8 words */
#pragma unroll
    for (int w = 0; w < kMaxWindows; w++) {
        if ((uint32_t)w >= nWin) break;
        const uint32_t w0 = t0 + 64u * (uint32_t)w;
        u64 chosen = 0;
        uint32_t ext0 = kNone, ext1 = kNone;
        const uint32_t anchorIn = st.anchor, seqBase = st.nseq;
#ifdef QZ_DEBUG_DUMP
        const uint32_t curIn = st.cur;
#endif
        if (st.cur < w0 + 64u) {
            const uint32_t cv = w < 8 ? c0 : c1;
            const int b = (w & 7) * 8;
            const u64 vis = (u64)rdlane(cv, (uint32_t)b) | ((u64)rdlane(cv, (uint32_t)b + 1u) << 32);
            const u64 chain = (u64)rdlane(cv, (uint32_t)b + 2u) | ((u64)rdlane(cv, (uint32_t)b + 3u) << 32);
            const u64 capc = (u64)rdlane(cv, (uint32_t)b + 4u) | ((u64)rdlane(cv, (uint32_t)b + 5u) << 32);
            const uint32_t exitC = rdlane(cv, (uint32_t)b + 6u);
            const uint32_t lastEnd = rdlane(cv, (uint32_t)b + 7u);
            uint32_t c = st.cur - w0;
            bool haveFlags = false;
            WinFlags f = { 0ull, 0ull };
            const uint32_t rw = R[w];
            for (;;) {
                uint32_t j, L;
                if ((vis >> c) & 1ull) {
                    /* on the speculative chain: everything from c on is already known */
                    const u64 cc = capc & ~below(c);
                    if (!cc) {
                        const u64 sel = chain & ~below(c);
                        chosen |= sel;
                        if (sel) st.anchor = w0 + lastEnd;
                        st.cur = w0 + exitC;
                        break;
                    }
                    j = (uint32_t)__builtin_ctzll(cc);
                    chosen |= chain & ~below(c) & below(j);
                    L = pf.capLen;
                } else {
                    if (!haveFlags) { f = window_flags(pf, rw, lane); haveFlags = true; }
                    const u64 m = (f.start >> c) << c;
                    if (!m) { st.cur = w0 + 64u; break; }
                    j = (uint32_t)__builtin_ctzll(m);
                    L = rdlane(rw, j) & 0xFFu;
                }
                chosen |= 1ull << j;
                if (L == pf.capLen) {
                    L = extend_match(lds32, w0 + j, rdlane(rw, j) >> 8, L, n, lane);
                    const uint32_t e = (j << 24) | L;
                    if (ext0 == kNone) ext0 = e; else ext1 = e;
                }
                c = j + L;
                st.anchor = w0 + c;
                if (c >= 64u) { st.cur = w0 + c; break; }
            }
            st.nseq += (uint32_t)__popcll(chosen);
        }
        uint32_t v = 0;
        v = lane == 0 ? (uint32_t)chosen : v;
        v = lane == 1 ? (uint32_t)(chosen >> 32) : v;
        v = lane == 2 ? anchorIn : v;
        v = lane == 3 ? seqBase : v;
        v = lane == 4 ? ext0 : v;
        v = lane == 5 ? ext1 : v;
        if (lane < 8u) srec[w * 8 + (int)lane] = v;
#ifdef QZ_DEBUG_DUMP
        if (lane == 0) {
            const uint32_t wi = (w0 >> 6);
            dbg[-(int)(2 * wi) - 1] = make_uint4((uint32_t)chosen, (uint32_t)(chosen >> 32), curIn, anchorIn);
            dbg[-(int)(2 * wi) - 2] = make_uint4(crec[w * 8 + 0], crec[w * 8 + 1], crec[w * 8 + 2], crec[w * 8 + 6]);
        }
#endif
    }
}

/* emission of one window's chosen matches by the wave that owns the window */
__device__ __forceinline__ void emit_window(const qzstd_hip_profile_t &pf, const uint32_t *lds32, const uint32_t *srec,
                                            uint32_t r, uint32_t w0, uint32_t lane, uint4 *out, uint32_t seqCap)
{
    const u64 chosen = (u64)srec[0] | ((u64)srec[1] << 32);
    if (!chosen) return;
    const uint32_t anchorIn = srec[2], seqBase = srec[3], ext0 = srec[4], ext1 = srec[5];
    const uint32_t off = r >> 8;
    uint32_t Lfin = r & 0xFFu;
    if (ext0 != kNone && (ext0 >> 24) == lane) Lfin = ext0 & 0xFFFFFFu;
    if (ext1 != kNone && (ext1 >> 24) == lane) Lfin = ext1 & 0xFFFFFFu;
    const bool ch = (chosen >> lane) & 1ull;
    const u64 lower = chosen & below(lane);
    const uint32_t rank = (uint32_t)__popcll(lower);
    const uint32_t myEnd = w0 + lane + Lfin;
    const int jprev = lower ? 63 - __builtin_clzll(lower) : 0;
    uint32_t prevEnd = __shfl(myEnd, jprev);
    if (!lower) prevEnd = anchorIn;
    if (ch) {
        const uint32_t p = w0 + lane, q = p - off;
        const uint32_t lit = p - prevEnd;
        uint32_t maxb = pf.backExt < lit ? pf.backExt : lit;
        maxb = maxb < q ? maxb : q;
        uint32_t b = 0;
        if (maxb) {
            /* the 4 bytes before p and before q, top byte = nearest; count equal bytes from the top */
            const uint32_t pb = p >= 4u ? lds_rd32u(lds32, p - 4u) : lds32[0] << (8u * (4u - p));
            const uint32_t qb = q >= 4u ? lds_rd32u(lds32, q - 4u) : lds32[0] << (8u * (4u - q));
            const uint32_t x = pb ^ qb;
            b = x ? (uint32_t)__builtin_clz(x) >> 3 : 4u;
            b = b < maxb ? b : maxb;
        }
        const uint32_t idx = seqBase + rank;
        if (idx < seqCap) out[idx] = make_uint4(off, lit - b, Lfin + b, 0u);
    }
}

/*
 * One workgroup = one block.  8 matcher waves + 1 parse wave, software-pipelined over
 * tiles with two barriers per tile:
 *
 *   interval 1 of iteration it      matchers: emit(it-2), speculative chains(it-1), phase A(it)
 *                                   parse wave: preload candidates(it-1) into registers
 *   barrier
 *   interval 2                      matchers: phase B(it) + candidate lengths(it)
 *                                   parse wave: serial pass(it-1)
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
    const qzstd_hip_profile_t &pf = args.prof[n > (64u << 10) ? 0 : (n > (32u << 10) ? 1 : 2)];
    const uint32_t T = 1u << pf.tileLog;
    const uint32_t nWin = T >> 6;
    const uint32_t nh = n >= pf.hashBytes ? n - pf.hashBytes + 1u : 0u; /* hashable positions */
    const uint32_t nTiles = (nh + T - 1u) >> pf.tileLog;

    /* ---- LDS layout for THIS block ---- */
    const uint32_t region = ((n + 15u) & ~15u) + 16u;
    uint8_t *lds8 = smem;
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *tbl = reinterpret_cast<uint32_t *>(smem + region);
    uint32_t *nearTab = tbl + pf.tableSize;
    uint32_t *results = nearTab + T;     /* packed candidates of the tile being parsed */
    uint32_t *save = results + T;        /* last window of the previous tile (x2) */
    uint32_t *crec = save + kSaveWords;  /* speculative-chain records */
    uint32_t *srec = crec + kRecWords;   /* serial-pass records */

    /* ---- stage the block: HBM -> LDS, 16 B per lane, coalesced ---- */
    {
        const uint8_t *g = args.src + blk.srcOff;
        const uint32_t nvec = n >> 4;
        const uint4 *g4 = reinterpret_cast<const uint4 *>(g);
        uint4 *l4 = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = tid; i < nvec; i += kThreads) l4[i] = g4[i];
        for (uint32_t i = (nvec << 4) + tid; i < region; i += kThreads) lds8[i] = i < n ? g[i] : (uint8_t)0;
        for (uint32_t i = tid; i < pf.tableSize; i += kThreads) tbl[i] = 0u;
        for (uint32_t i = tid; i < T; i += kThreads) { nearTab[i] = 0xFFFFFFFFu; results[i] = 0u; }
    }
    __syncthreads();

    ParseState st = { 0u, 0u, 0u };
    uint4 *out = args.seqs + blk.seqOff;
    const uint32_t hiMask = pf.hashBytes >= 8 ? 0xFFFFFFFFu : ((1u << (8u * (pf.hashBytes - 4u))) - 1u);
    const uint32_t nearShift = 32u - pf.tileLog;
    const uint32_t stampShift = pf.tileLog + kTagBits;
    const uint32_t nTilesMax = (QZSTD_HIP_BLOCK_MAX >> pf.tileLog);

    /* matcher registers: packed candidates of the two most recent tiles this thread matched */
    uint32_t resNew[kMaxPosPerThread] = { 0u, 0u };  /* tile it-1 after interval 2 */
    uint32_t resOld[kMaxPosPerThread] = { 0u, 0u };  /* tile it-2 */
    uint32_t R[kMaxWindows];                         /* parse wave: candidates of tile it-1 */
#pragma unroll
    for (int w = 0; w < kMaxWindows; w++) R[w] = 0u;

    for (uint32_t it = 0; it < nTiles + 2u; it++) {
        const uint32_t t0 = it << pf.tileLog;
        const uint32_t stamp = (nTilesMax - 1u - (it & (nTilesMax - 1u))) << stampShift;
        uint32_t v[kMaxPosPerThread], mix[kMaxPosPerThread], old[kMaxPosPerThread];
        bool valid[kMaxPosPerThread];

        /* ================= interval 1 ================= */
        if (matcher) {
            if (it >= 2u && it - 2u < nTiles && !(args.dbg & 8u)) { /* emit(it-2) */
#pragma unroll
                for (int j = 0; j < kMaxPosPerThread; j++) {
                    const uint32_t w = wave + (uint32_t)j * kMatchWaves;
                    if (w < nWin) emit_window(pf, lds32, srec + w * 8u, resOld[j], t0 - 2u * T + 64u * w, lane, out, blk.seqCap);
                }
            }
            if (it >= 1u && it - 1u < nTiles && !(args.dbg & 4u)) { /* speculative chains(it-1) */
#pragma unroll
                for (int j = 0; j < kMaxPosPerThread; j++) {
                    const uint32_t w = wave + (uint32_t)j * kMatchWaves;
                    if (w < nWin) {
                        const bool havePrev = w != 0u || it >= 2u;
                        uint32_t rPrev = 0u;
                        if (havePrev) rPrev = w != 0u ? results[64u * (w - 1u) + lane] : save[((it - 2u) & 1u) * 64u + lane];
                        spec_chain(pf, rPrev, havePrev, resNew[j], lane, crec + w * 8u);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kMaxPosPerThread; j++) { /* phase A(it) */
                const uint32_t pin = tid + (uint32_t)j * kMatchThreads;
                const uint32_t p = t0 + pin;
                valid[j] = it < nTiles && pin < T && p < nh;
                v[j] = 0; mix[j] = 0; old[j] = 0;
                if (valid[j]) {
                    const uint32_t d = p >> 2, s = p & 3u;
                    const uint32_t w0 = lds32[d], w1 = lds32[d + 1];
                    v[j] = __builtin_amdgcn_alignbyte(w1, w0, s);
                    uint32_t hi = 0;
                    if (pf.hashBytes > 4) hi = __builtin_amdgcn_alignbyte(lds32[d + 2], w1, s) & hiMask;
                    mix[j] = (v[j] * kPrime1) ^ (hi * kPrime2);
                    old[j] = tbl[__umulhi(mix[j], pf.tableSize)];
                    if (pf.nearTab)
                        atomicMin(&nearTab[mix[j] >> nearShift], stamp | (pin << kTagBits) | ((mix[j] >> 3) & kTagMask));
                }
            }
        } else if (it >= 1u && it - 1u < nTiles) {
#pragma unroll
            for (int w = 0; w < kMaxWindows; w++) R[w] = (uint32_t)w < nWin ? results[64 * w + (int)lane] : 0u;
        }
        __syncthreads(); /* B1 */

        /* ================= interval 2 ================= */
        if (matcher) {
#pragma unroll
            for (int j = 0; j < kMaxPosPerThread; j++) { resOld[j] = resNew[j]; resNew[j] = 0u; }
#pragma unroll
            for (int j = 0; j < kMaxPosPerThread; j++) {
                const uint32_t pin = tid + (uint32_t)j * kMatchThreads;
                if (valid[j]) {
                    const uint32_t p = t0 + pin;
                    const uint32_t tag = (mix[j] >> 3) & kTagMask;
                    const uint32_t en = pf.nearTab ? nearTab[mix[j] >> nearShift] : 0xFFFFFFFFu;
                    atomicMax(&tbl[__umulhi(mix[j], pf.tableSize)], ((p + 1u) << kTagBits) | tag);
                    const uint32_t cap = pf.capLen < n - p ? pf.capLen : n - p;
                    /* candidate 1: newest position of earlier tiles; candidate 2: earliest of this tile */
                    uint32_t q1 = kNone, q2 = kNone;
                    const uint32_t e = old[j];
                    if (e != 0u && (e & kTagMask) == tag) {
                        const uint32_t q = (e >> kTagBits) - 1u;
                        if (pf.window == 0u || p - q <= pf.window) q1 = q;
                    }
                    if (pf.nearTab && (en >> stampShift) == (stamp >> stampShift) && (en & kTagMask) == tag) {
                        const uint32_t q = t0 + ((en >> kTagBits) & (T - 1u));
                        if (q < p) q2 = q;
                    }
                    uint32_t bestLen = 0, bestOff = 0;
                    if ((q1 != kNone || q2 != kNone) && !(args.dbg & 2u)) {
                        uint32_t own[9];
                        const uint32_t pd = p >> 2;
#pragma unroll
                        for (int i = 0; i < 9; i++) own[i] = lds32[pd + i];
                        if (q1 != kNone) {
                            const uint32_t l = match_len(lds32, own, p, q1, cap);
                            if (l >= 4u) { bestLen = l; bestOff = p - q1; }
                        }
                        if (q2 != kNone) {
                            const uint32_t l = match_len(lds32, own, p, q2, cap);
                            if (l >= 4u && l >= bestLen) { bestLen = l; bestOff = p - q2; }
                        }
                    }
                    resNew[j] = bestLen ? ((bestOff << 8) | bestLen) : 0u;
                }
                if (it < nTiles && pin < T) {
                    results[pin] = resNew[j];
                    if (pin >= T - 64u) save[(it & 1u) * 64u + (pin - (T - 64u))] = resNew[j];
                }
            }
        } else if (it >= 1u && it - 1u < nTiles && !(args.dbg & 1u)) {
            serial_pass(pf, lds32, R, crec, srec, t0 - T, nWin, n, lane, st, out + blk.seqCap);
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
        if (a.prof[c].tileLog > 10 || a.prof[c].tileLog < 6 || a.prof[c].capLen > 128 || a.prof[c].capLen < 32 || a.prof[c].minMatch < 4 || a.prof[c].hashBytes < 4 ||
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
    { const char *ab = getenv("QZSTD_HIP_ABLATE"); a.dbg = ab ? (uint32_t)atoi(ab) : 0u; }
    hipLaunchKernelGGL(qzstd_find_sequences_kernel, dim3(nBlocks), dim3(kThreads), lds, (hipStream_t)stream, a);
    QZ_CHECK(hipGetLastError(), "launch qzstd_find_sequences_kernel");
    return 0;
}

} /* extern "C" */
