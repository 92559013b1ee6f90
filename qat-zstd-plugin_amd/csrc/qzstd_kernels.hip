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
constexpr uint32_t kTagBits = 14;
constexpr uint32_t kTagMask = (1u << kTagBits) - 1u;
constexpr uint32_t kPrime1 = 2654435761u;
constexpr uint32_t kPrime2 = 0x85EBCA77u;

struct LaunchArgs {
    const uint8_t *src;
    const qzstd_hip_block_t *blocks;
    uint4 *seqs; /* ZSTD_Sequence = 4 x u32 */
    uint32_t *nseq;
    qzstd_hip_profile_t prof[3]; /* by block size class: >64 KiB, >32 KiB, <=32 KiB */
    uint32_t dbg; /* ablation switches for profiling (QZSTD_HIP_ABLATE), 0 in production */
};

/* 4 bytes at an arbitrary LDS byte address: two aligned dword reads + v_alignbyte_b32 */
__device__ __forceinline__ uint32_t lds_rd32u(const uint32_t *lds32, uint32_t a)
{
    const uint32_t d = a >> 2;
    return __builtin_amdgcn_alignbyte(lds32[d + 1], lds32[d], a & 3u);
}

/* common prefix length of [p..) and [q..), first 4 bytes already known equal */
__device__ __forceinline__ uint32_t match_len(const uint32_t *lds32, uint32_t p, uint32_t q, uint32_t cap)
{
    uint32_t L = 4;
    while (L < cap) {
        const uint32_t x = lds_rd32u(lds32, p + L) ^ lds_rd32u(lds32, q + L);
        if (x) { L += (uint32_t)__builtin_ctz(x) >> 3; break; }
        L += 4;
    }
    return L < cap ? L : cap;
}

__device__ __forceinline__ uint32_t min_len(const qzstd_hip_profile_t &pf, uint32_t off)
{
    return pf.minMatch + ((off >> pf.farLog1) ? 1u : 0u) + ((off >> pf.farLog2) ? 1u : 0u);
}

/* parse-wave state, uniform across the wave */
struct ParseState {
    uint32_t cur;    /* next position the parse looks at */
    uint32_t anchor; /* end of the last emitted match = start of pending literals */
    uint32_t nseq;   /* matches emitted so far */
};

/* cooperative forward extension of a chosen match that hit the candidate cap */
__device__ __forceinline__ uint32_t extend_match(const uint8_t *lds8, uint32_t p, uint32_t off, uint32_t L,
                                                 uint32_t n, uint32_t lane)
{
    for (;;) {
        const uint32_t a = p + L + lane;
        const bool ok = a < n && lds8[a] == lds8[a - off];
        const unsigned long long bad = __ballot(!ok);
        if (bad) return L + (uint32_t)__builtin_ctzll(bad);
        L += 64;
    }
}

/* the (lazy) greedy parse of one tile's packed candidates, 64 positions per step */
__device__ void parse_tile(const qzstd_hip_profile_t &pf, const uint8_t *lds8, const uint32_t *results,
                           uint32_t t0, uint32_t nh, uint32_t n, uint32_t lane, ParseState &st,
                           uint4 *out, uint32_t seqCap)
{
    const uint32_t T = 1u << pf.tileLog;
    const uint32_t tEnd = t0 + T < nh ? t0 + T : nh;
    for (uint32_t w0 = t0; w0 < tEnd; w0 += 64) {
        if (st.cur >= w0 + 64) continue; /* window lies inside an already emitted match */
        const uint32_t pin = w0 - t0 + lane;
        const uint32_t r = results[pin];
        const uint32_t r1 = lane != 63u ? results[pin + 1] : 0u; /* no lazy deferral across a window edge */
        const uint32_t len = r & 0xFFu, off = r >> 8;
        const uint32_t len1 = r1 & 0xFFu, off1 = r1 >> 8;
        const bool take = len != 0 && len >= min_len(pf, off);
        const bool take1 = len1 != 0 && len1 >= min_len(pf, off1);
        const bool start = take && !(pf.lazy && take1 && len1 > len);
        const unsigned long long mask = __ballot(start);
        unsigned long long chosen = 0;
        uint32_t Lfin = len;
        uint32_t c = st.cur > w0 ? st.cur - w0 : 0u;
        for (;;) {
            const unsigned long long m = (mask >> c) << c;
            if (!m) { st.cur = w0 + 64; break; }
            const uint32_t j = (uint32_t)__builtin_ctzll(m);
            uint32_t L = __builtin_amdgcn_readlane(len, j);
            if (L == pf.capLen) {
                const uint32_t o = __builtin_amdgcn_readlane(off, j);
                L = extend_match(lds8, w0 + j, o, L, n, lane);
                if (lane == j) Lfin = L;
            }
            chosen |= 1ull << j;
            c = j + L;
            if (c >= 64) { st.cur = w0 + c; break; }
        }
        if (chosen) {
            const bool ch = (chosen >> lane) & 1ull;
            const unsigned long long below = chosen & ((1ull << lane) - 1ull);
            const uint32_t rank = (uint32_t)__popcll(below);
            const uint32_t myEnd = w0 + lane + Lfin;
            const int jprev = below ? 63 - __builtin_clzll(below) : 0;
            uint32_t prevEnd = __shfl(myEnd, jprev);
            if (!below) prevEnd = st.anchor;
            if (ch) {
                const uint32_t p = w0 + lane, q = p - off;
                const uint32_t lit = p - prevEnd;
                uint32_t maxb = pf.backExt < lit ? pf.backExt : lit;
                maxb = maxb < q ? maxb : q;
                uint32_t b = 0;
                while (b < maxb && lds8[p - b - 1] == lds8[q - b - 1]) b++;
                const uint32_t idx = st.nseq + rank;
                if (idx < seqCap) out[idx] = make_uint4(off, lit - b, Lfin + b, 0u);
            }
            const uint32_t jl = 63u - (uint32_t)__builtin_clzll(chosen);
            st.anchor = w0 + jl + __builtin_amdgcn_readlane(Lfin, jl);
            st.nseq += (uint32_t)__popcll(chosen);
        }
    }
}

__global__ __launch_bounds__(kThreads) void qzstd_find_sequences_kernel(LaunchArgs args)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const bool matcher = tid < (uint32_t)kMatchThreads;
    const qzstd_hip_block_t blk = args.blocks[blockIdx.x];
    const uint32_t n = blk.srcLen;
    const qzstd_hip_profile_t &pf = args.prof[n > (64u << 10) ? 0 : (n > (32u << 10) ? 1 : 2)];
    const uint32_t T = 1u << pf.tileLog;
    const uint32_t nh = n >= pf.hashBytes ? n - pf.hashBytes + 1u : 0u; /* hashable positions */

    /* ---- LDS layout for THIS block ---- */
    const uint32_t region = ((n + 15u) & ~15u) + 16u;
    uint8_t *lds8 = smem;
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *tbl = reinterpret_cast<uint32_t *>(smem + region);
    uint32_t *nearTab = tbl + pf.tableSize;
    uint32_t *results = nearTab + T;

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

    /* per-thread candidate results of the tile just matched (written to LDS one interval later) */
    uint32_t res[kMaxPosPerThread] = { 0u, 0u };
    bool havePrev = false;

    for (uint32_t t0 = 0; t0 < nh; t0 += T) {
        const uint32_t stamp = (nTilesMax - 1u - (t0 >> pf.tileLog)) << stampShift;
        uint32_t v[kMaxPosPerThread], mix[kMaxPosPerThread], old[kMaxPosPerThread];
        bool valid[kMaxPosPerThread];

        if (matcher) {
            /* interval 1: publish the previous tile's candidates, then phase A of this tile */
            if (havePrev) {
#pragma unroll
                for (int j = 0; j < kMaxPosPerThread; j++) {
                    const uint32_t pin = tid + (uint32_t)j * kMatchThreads;
                    if (pin < T) results[pin] = res[j];
                }
            }
#pragma unroll
            for (int j = 0; j < kMaxPosPerThread; j++) {
                const uint32_t pin = tid + (uint32_t)j * kMatchThreads;
                const uint32_t p = t0 + pin;
                valid[j] = pin < T && p < nh;
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
        }
        __syncthreads(); /* B1: all table reads of the tile done, near table complete */

        if (matcher) {
            /* interval 2: insert (largest position wins), then measure both candidates */
#pragma unroll
            for (int j = 0; j < kMaxPosPerThread; j++) {
                res[j] = 0;
                if (!valid[j]) continue;
                const uint32_t pin = tid + (uint32_t)j * kMatchThreads;
                const uint32_t p = t0 + pin;
                const uint32_t tag = (mix[j] >> 3) & kTagMask;
                const uint32_t en = pf.nearTab ? nearTab[mix[j] >> nearShift] : 0xFFFFFFFFu;
                atomicMax(&tbl[__umulhi(mix[j], pf.tableSize)], ((p + 1u) << kTagBits) | tag);
                const uint32_t cap = pf.capLen < n - p ? pf.capLen : n - p;
                uint32_t bestLen = 0, bestOff = 0;
                if (args.dbg & 2u) continue;
                const uint32_t e = old[j];
                if (e != 0u && (e & kTagMask) == tag) {
                    const uint32_t q = (e >> kTagBits) - 1u;
                    const uint32_t off = p - q;
                    if ((pf.window == 0u || off <= pf.window) && lds_rd32u(lds32, q) == v[j]) {
                        bestLen = match_len(lds32, p, q, cap);
                        bestOff = off;
                    }
                }
                if (pf.nearTab && (en >> stampShift) == (stamp >> stampShift) && (en & kTagMask) == tag) {
                    const uint32_t q = t0 + ((en >> kTagBits) & (T - 1u));
                    if (q < p && lds_rd32u(lds32, q) == v[j]) {
                        const uint32_t l = match_len(lds32, p, q, cap);
                        if (l >= bestLen) { bestLen = l; bestOff = p - q; }
                    }
                }
                res[j] = bestLen ? ((bestOff << 8) | bestLen) : 0u;
            }
            havePrev = true;
        } else if (t0 != 0u && !(args.dbg & 1u)) {
            parse_tile(pf, lds8, results, t0 - T, nh, n, lane, st, out, blk.seqCap);
        }
        __syncthreads(); /* B2: inserts done; parse wave finished reading the scratch */
    }

    /* ---- drain: publish and parse the last tile, then the trailing-literals delimiter ---- */
    if (nh != 0u) {
        if (matcher) {
#pragma unroll
            for (int j = 0; j < kMaxPosPerThread; j++) {
                const uint32_t pin = tid + (uint32_t)j * kMatchThreads;
                if (pin < T) results[pin] = res[j];
            }
        }
        __syncthreads();
        if (!matcher) parse_tile(pf, lds8, results, ((nh - 1u) >> pf.tileLog) << pf.tileLog, nh, n, lane, st, out, blk.seqCap);
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
        if (a.prof[c].tileLog > 10 || a.prof[c].tileLog < 6 || a.prof[c].capLen > 128 || a.prof[c].hashBytes < 4 ||
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
