// zsa_common.h -- types, format constants and the error contract shared by the HIP kernels and the host
// side of libzstd_amd.  MI355X (gfx950) only.
//
// Error codes and return convention follow the reference boundary: zstd_errors.h:65-102 and
// error_private.h:57-61 (a size_t that is either a byte count or (size_t)-code).
#pragma once
#include <stdint.h>
#include <stddef.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef int16_t  s16;
typedef uint32_t u32;
typedef int32_t  s32;
typedef uint64_t u64;
typedef int64_t  s64;

#ifndef ZSA_WAVE_HEADER
#define ZSA_WAVE_HEADER "zsa_wave_hip.h"
#endif
#include ZSA_WAVE_HEADER   // defines ZSA_DEV, ZSA_SHARED and the wv_* wave primitives

enum ZsaError : u32 {
    ZSA_no_error = 0, ZSA_GENERIC = 1, ZSA_prefix_unknown = 10, ZSA_version_unsupported = 12,
    ZSA_frameParameter_unsupported = 14, ZSA_frameParameter_windowTooLarge = 16,
    ZSA_corruption_detected = 20, ZSA_checksum_wrong = 22, ZSA_literals_headerWrong = 24,
    ZSA_dictionary_corrupted = 30, ZSA_dictionary_wrong = 32, ZSA_dictionaryCreation_failed = 34,
    ZSA_parameter_unsupported = 40, ZSA_parameter_combination_unsupported = 41, ZSA_parameter_outOfBound = 42,
    ZSA_tableLog_tooLarge = 44, ZSA_maxSymbolValue_tooLarge = 46, ZSA_maxSymbolValue_tooSmall = 48,
    ZSA_stabilityCondition_notRespected = 50, ZSA_stage_wrong = 60, ZSA_init_missing = 62,
    ZSA_memory_allocation = 64, ZSA_workSpace_tooSmall = 66, ZSA_dstSize_tooSmall = 70,
    ZSA_srcSize_wrong = 72, ZSA_dstBuffer_null = 74, ZSA_noForwardProgress_destFull = 80,
    ZSA_noForwardProgress_inputEmpty = 82, ZSA_frameIndex_tooLarge = 100, ZSA_seekableIO = 102,
    ZSA_dstBuffer_wrong = 104, ZSA_srcBuffer_wrong = 105, ZSA_sequenceProducer_failed = 106,
    ZSA_externalSequences_invalid = 107, ZSA_maxCode = 120
};
#define ZSA_ERR(code) ((u64)0 - (u64)(code))

// ---- format constants (zstd_internal.h:83-169, zstd.h:143-144) ----
#define ZSA_MAGIC            0xFD2FB528u
#define ZSA_MAGIC_SKIP_START 0x184D2A50u
#define ZSA_MAGIC_SKIP_MASK  0xFFFFFFF0u
#define ZSA_MAGIC_DICT       0xEC30A437u
#define ZSA_BLOCKSIZE_MAX    (128u * 1024u)
#define ZSA_WLOG_ABSMIN      10
#define ZSA_WLOG_MAX         31
#define ZSA_MINMATCH         3
#define ZSA_MaxLL   35
#define ZSA_MaxML   52
#define ZSA_MaxOff  31
#define ZSA_DefaultMaxOff 28
#define ZSA_MaxSeq  52
#define ZSA_LLFSELog 9
#define ZSA_MLFSELog 9
#define ZSA_OffFSELog 8
#define ZSA_LONGNBSEQ 0x7F00
#define ZSA_HUF_TABLELOG_MAX 12
#define ZSA_FSE_MIN_TABLELOG 5
#define ZSA_FSE_MAX_TABLELOG 12
#define ZSA_FSE_TABLESTEP(ts) (((ts) >> 1) + ((ts) >> 3) + 3)
#define ZSA_LL_DEFAULTNORMLOG 6
#define ZSA_ML_DEFAULTNORMLOG 6
#define ZSA_OF_DEFAULTNORMLOG 5
enum { zsa_set_basic = 0, zsa_set_rle = 1, zsa_set_compressed = 2, zsa_set_repeat = 3 };
enum { zsa_bt_raw = 0, zsa_bt_rle = 1, zsa_bt_compressed = 2, zsa_bt_reserved = 3 };
enum { zsa_rep_none = 0, zsa_rep_check = 1, zsa_rep_valid = 2 };

// Tables of the format, indexable from device code without constant-memory indirection.
ZSA_DEV u32 zsa_LL_bits(u32 c)
{   // zstd_internal.h:124-130
    if (c < 16) return 0;
    if (c < 24) return (u32)((0x33221111u >> ((c - 16) * 4)) & 0xF);
    return c == 24 ? 4 : c - 19;   // 25->6, 26->7, ... 35->16
}
ZSA_DEV u32 zsa_ML_bits(u32 c)
{   // zstd_internal.h:141-149
    if (c < 32) return 0;
    if (c < 40) return (u32)((0x33221111u >> ((c - 32) * 4)) & 0xF);
    if (c < 42) return 4;
    if (c == 42) return 5;
    return c - 36;                 // 43->7 ... 52->16
}
ZSA_DEV s32 zsa_LL_defaultNorm(u32 s)
{   // zstd_internal.h:131-137
    if (s == 0) return 4;
    if (s == 1 || s == 25) return 3;
    if (s >= 32) return -1;
    if ((s >= 13 && s <= 15) || (s >= 27)) return 1;
    return 2;
}
ZSA_DEV s32 zsa_ML_defaultNorm(u32 s)
{   // zstd_internal.h:150-158
    if (s == 0) return 1;
    if (s == 1) return 4;
    if (s == 2) return 3;
    if (s <= 8) return 2;
    if (s >= 46) return -1;
    return 1;
}
ZSA_DEV s32 zsa_OF_defaultNorm(u32 s)
{   // zstd_internal.h:162-167
    if (s >= 24) return -1;
    if (s >= 6 && s <= 8) return 2;
    return 1;
}
// decoder base values (zstd_decompress_internal.h:31-57)
ZSA_DEV u32 zsa_LL_base(u32 c)
{
    if (c < 16) return c;
    if (c < 20) return 16 + (c - 16) * 2;          // 16,18,20,22
    if (c < 22) return 24 + (c - 20) * 4;          // 24,28
    if (c < 24) return 32 + (c - 22) * 8;          // 32,40
    if (c == 24) return 48;
    return 64u << (c - 25);                        // 64,128,...,65536
}
ZSA_DEV u32 zsa_ML_base(u32 c)
{
    if (c < 32) return c + 3;
    if (c < 36) return 35 + (c - 32) * 2;          // 35,37,39,41
    if (c < 38) return 43 + (c - 36) * 4;          // 43,47
    if (c < 40) return 51 + (c - 38) * 8;          // 51,59
    if (c < 42) return 67 + (c - 40) * 16;         // 67,83
    if (c == 42) return 99;
    return (128u << (c - 43)) + 3;                 // 131,259,...,65539
}
ZSA_DEV u32 zsa_OF_base(u32 c) { return c == 0 ? 0 : (c == 1 ? 1 : (1u << c) - 3); }

ZSA_DEV u32 zsa_highbit32(u32 v) { return 31u - (u32)__builtin_clz(v); }

// ---- FSE table layout by the whole wave (shared by the encoder's and the decoder's table builders) --------------------------
// Both FSE_buildCTable_wksp (fse_compress.c:69-215) and ZSTD_buildFSETable_body (zstd_decompress_block.c:455-573) lay the
// symbols out the same way: low-probability symbols (norm == -1) take the top cells in symbol order; every other symbol s
// takes norm[s] of the remaining cells, which are visited in the order 0, step, 2 step, ... (mod tableSize); and each cell
// then needs the RANK of its position among the cells of the same symbol, in position order (it selects the state the cell
// stands for).  The reference does this with two serial loops over the table.  Here:
//   * lane s owns symbol s (at most 53 sequence codes, 13 Huffman weights): a wave prefix sum of the frequencies gives the
//     first visit of every symbol;
//   * the visits are taken 64 at a time: the j-th visit lands on cell (j * step) mod tableSize, a ballot drops the visits that
//     land in the low-probability zone, and the lane finds the symbol of its visit by a 6-step search in the prefix sums;
//   * the cells are then taken 64 at a time in position order: six ballots over the bits of the symbol tell a lane which
//     other lanes hold the same symbol, the ones below it plus a per-symbol running count give its rank.
// emit(cell, symbol, rank) is called once per cell by the lane that owns it.  norm, and the three scratch arrays, are LDS.
struct ZsaSpreadScratch { u16* first; u16* seen; u8* sym; };   // first: 64 entries, seen: 64 entries, sym: tableSize entries
template <class EMIT>
ZSA_DEV void zsa_fse_spread_wave(const ZsaSpreadScratch& X, const s16* norm, u32 maxSV, u32 tableLog, int lane, EMIT emit)
{
    u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1, step = ZSA_FSE_TABLESTEP(tableSize);
    u64 const below = ((u64)1 << lane) - 1;
    int const n = (u32)lane <= maxSV ? (int)norm[lane] : 0;
    u32 const freq = n > 0 ? (u32)n : 0;
    u32 const firstVisit = wv_scan_incl(freq) - freq;
    u64 const lowMask = wv_ballot(n == -1);
    u32 const high = tableSize - 1 - (u32)__builtin_popcountll(lowMask);        // highThreshold
    X.first[lane] = (u16)firstVisit;
    X.seen[lane] = 0;
    if (n == -1) X.sym[tableSize - 1 - (u32)__builtin_popcountll(lowMask & below)] = (u8)lane;
    wv_lds_fence();
    u32 taken = 0;
    for (u32 j0 = 0; j0 < tableSize; j0 += ZSA_WAVE) {
        u32 const j = j0 + (u32)lane, cell = (j * step) & tableMask;
        bool const lands = j < tableSize && cell <= high;
        u64 const lm = wv_ballot(lands);
        u32 const k = taken + (u32)__builtin_popcountll(lm & below);             // the lane's visit is the k-th that counts
        taken += (u32)__builtin_popcountll(lm);
        if (lands) {   // the last symbol whose first visit is at or before k
            u32 s = 0;
            for (u32 b = 32; b; b >>= 1) { u32 const t = s + b; if (t <= maxSV && X.first[t] <= k) s = t; }
            X.sym[cell] = (u8)s;
        }
    }
    wv_lds_fence();
    for (u32 c0 = 0; c0 < tableSize; c0 += ZSA_WAVE) {
        u32 const cell = c0 + (u32)lane;
        bool const on = cell < tableSize;
        u32 const s = on ? X.sym[cell] : 0;
        u64 same = wv_ballot(on);
        for (u32 b = 0; b < 6; b++) { u64 const bb = wv_ballot((s >> b) & 1); same &= ((s >> b) & 1) ? bb : ~bb; }
        u32 const rank = on ? X.seen[s] + (u32)__builtin_popcountll(same & below) : 0;
        wv_lds_fence();
        if (on && (same >> lane) == 1) X.seen[s] += (u16)__builtin_popcountll(same);   // the highest lane of the group
        wv_lds_fence();
        if (on) emit(cell, s, rank);
    }
}

// ---- unaligned little-endian memory access ----
ZSA_DEV u32 zsa_ld16(const u8* p) { u16 v; __builtin_memcpy(&v, p, 2); return v; }
ZSA_DEV u32 zsa_ld24(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }
ZSA_DEV u32 zsa_ld32(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
ZSA_DEV u64 zsa_ld64(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
ZSA_DEV void zsa_st16(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); }
ZSA_DEV void zsa_st24(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); }
ZSA_DEV void zsa_st32(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); p[3] = (u8)(v >> 24); }

// phase-cycle accounting (lane 0, shader clock); slots are named in zsa_lib.hip
#define ZSA_NPROF 12
#define ZSA_PROF(S, lane, slot) do { if ((lane) == 0) { u64 const t_ = wv_clock(); (S)->prof[slot] += t_ - (S)->profLast; (S)->profLast = t_; } } while (0)

// One frame of a batch, as the kernels see it (all pointers are device pointers).
struct ZsaFrameDesc {
    const u8* src; u8* dst;
    u64 srcSize; u64 dstCapacity;
};

// Copies len <= 32 bytes from s to d (global memory, not overlapping), each lane its own run: up to four 8-byte moves -- at
// 0, 8, 16 and at len - 8, which covers the end whatever len is -- or two 4-byte moves, or single bytes below 4.  All loads are
// issued before the first store, so a lane's moves share one memory round trip.
ZSA_DEV void zsa_copy_run32(u8* d, const u8* s, u32 len)
{
    if (len >= 8) {
        u64 const a = wv_ldg64(s), t = wv_ldg64(s + len - 8);
        u64 b = 0, c = 0;
        if (len > 16) b = wv_ldg64(s + 8);
        if (len > 24) c = wv_ldg64(s + 16);
        wv_stg64u(d, a);
        if (len > 16) wv_stg64u(d + 8, b);
        if (len > 24) wv_stg64u(d + 16, c);
        wv_stg64u(d + len - 8, t);
    } else if (len >= 4) {
        u32 const a = wv_ldg32(s), t = wv_ldg32(s + len - 4);
        wv_stg32(d, a); wv_stg32(d + len - 4, t);
    } else if (len) {
        u32 const a = wv_ldg8(s), b = len > 1 ? wv_ldg8(s + 1) : 0, c = len > 2 ? wv_ldg8(s + 2) : 0;
        wv_stg8(d, a);
        if (len > 1) wv_stg8(d + 1, b);
        if (len > 2) wv_stg8(d + 2, c);
    }
}
