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
