/*
 * qzstd_profile.c — zstd level -> search profile, and the LDS budget derived from it.
 *
 * The reference passes the zstd level straight to the accelerator as the QAT
 * compression level (/root/reference/src/qatseqprod.c:1154, session params
 * :935-946) and only accepts 1..12 (:86-87, :1132-1137).  Here the level selects
 * the parameters of the LDS match-finder.  Plain C, no GPU needed.
 *
 * LDS budget per workgroup (gfx950: 160 KiB = 163840 B per CU):
 *     block bytes (<=128 KiB, +16 pad) + 4*tableSize + near table 4<<tileLog
 *     + 2 tiles of u16 jump lengths + per-window start masks / emission records + 64 B control
 * so a full 128 KiB block leaves room for 6400 table entries at tileLog 9; smaller
 * blocks get bigger tables and more workgroups per CU.
 */
#include "qzstd_hip.h"

#include <string.h>

#define QZ_LDS_MAX 163840u
#define QZ_LDS_CTRL 64u


int qzstd_hip_profile_for_level(int level, size_t blockSize, qzstd_hip_profile_t *out)
{
    if (level < 1 || level > 12 || !out) return -1;
    memset(out, 0, sizeof(*out));
    if (blockSize > (64u << 10)) out->tableSize = 6400u;
    else if (blockSize > (32u << 10)) out->tableSize = 16384u;
    else out->tableSize = 8192u;
    out->tileLog = 9;
    out->capLen = 64;
    out->minMatch = 4;
    out->farLog1 = 12;
    out->farLog2 = 16;
    out->lazy = 3;
    out->backExt = 4;
    out->nearTab = 1;
    out->window = 0;
    out->hashBytes = 5;
    out->extLog = 11;
    return 0;
}

size_t qzstd_hip_sequence_bound(size_t srcSize)
{
    return srcSize / 3 + 1 + srcSize / 1024 + 1;
}

static size_t qz_need(int level, uint32_t len)
{
    qzstd_hip_profile_t p;
    if (qzstd_hip_profile_for_level(level, len, &p)) return 0;
    /* block bytes (+16 B pad for dword over-reads) + table + near table + parse scratch + control */
    return (size_t)(((len + 15u) & ~15u) + 16u) + 4u * p.tableSize        /* hash table                                    */
           + (4u << p.tileLog)     /* tile-local near table                         */
           + 2u * (4u << p.tileLog) /* per-position parse words, 2 tiles in flight */
           + 2u * ((1u << p.tileLog) >> 6) * 32u /* per-window emission records, x2 */
           + QZ_LDS_CTRL;
}

/* A launch may mix block sizes; each workgroup lays out LDS for ITS block, so the
 * launch needs the largest footprint among the size classes that can occur. */
size_t qzstd_hip_lds_bytes(int level, uint32_t maxBlockLen)
{
    size_t need, n2;
    if (maxBlockLen > QZSTD_HIP_BLOCK_MAX || level < 1 || level > 12) return 0;
    need = qz_need(level, maxBlockLen);
    if (maxBlockLen > (64u << 10) && (n2 = qz_need(level, 64u << 10)) > need) need = n2;
    if (maxBlockLen > (32u << 10) && (n2 = qz_need(level, 32u << 10)) > need) need = n2;
    return need <= QZ_LDS_MAX ? need : 0;
}
