/* zo_common.h -- shared definitions of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or executed from the
 * product library (zstd_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and only as the checker.
 *
 * The oracle is a plain-C restatement of the one-shot path of the reference (DataDog/zstd, vendored libzstd 1.5.5):
 * compression at the ZSTD_fast, ZSTD_dfast and ZSTD_greedy/ZSTD_lazy/ZSTD_lazy2 levels (dictionaries at ZSTD_fast),
 * decompression of any frame.  Every function cites the reference file:line it follows.  Parity is PINNED:
 * tests/test_oracle.py and tests/test_oracle_dict.py compare it byte-for-byte with the reference's own C compiled into
 * oracle/_ref/libzstd_ref.so, and against the committed reference-generated fixtures in tests/golden/.
 */
#ifndef ZO_COMMON_H
#define ZO_COMMON_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef int16_t  s16;
typedef uint32_t u32;
typedef uint64_t u64;

/* ---- error contract: zstd_errors.h:65-102, error_private.h:57-61 ---- */
enum {
    ZE_no_error = 0, ZE_GENERIC = 1, ZE_prefix_unknown = 10, ZE_version_unsupported = 12,
    ZE_frameParameter_unsupported = 14, ZE_frameParameter_windowTooLarge = 16,
    ZE_corruption_detected = 20, ZE_checksum_wrong = 22, ZE_literals_headerWrong = 24,
    ZE_dictionary_corrupted = 30, ZE_dictionary_wrong = 32, ZE_dictionaryCreation_failed = 34,
    ZE_parameter_unsupported = 40, ZE_parameter_combination_unsupported = 41,
    ZE_parameter_outOfBound = 42, ZE_tableLog_tooLarge = 44, ZE_maxSymbolValue_tooLarge = 46,
    ZE_maxSymbolValue_tooSmall = 48, ZE_stabilityCondition_notRespected = 50, ZE_stage_wrong = 60,
    ZE_init_missing = 62, ZE_memory_allocation = 64, ZE_workSpace_tooSmall = 66,
    ZE_dstSize_tooSmall = 70, ZE_srcSize_wrong = 72, ZE_dstBuffer_null = 74,
    ZE_noForwardProgress_destFull = 80, ZE_noForwardProgress_inputEmpty = 82,
    ZE_frameIndex_tooLarge = 100, ZE_seekableIO = 102, ZE_dstBuffer_wrong = 104,
    ZE_srcBuffer_wrong = 105, ZE_sequenceProducer_failed = 106, ZE_externalSequences_invalid = 107,
    ZE_maxCode = 120
};
#define ZO_ERR(name) ((size_t)0 - (size_t)ZE_##name)
static inline int zo_isErr(size_t c) { return c > (size_t)0 - (size_t)ZE_maxCode; }

/* ---- format constants: zstd_internal.h:83-169, zstd.h:143-144 ---- */
#define ZO_MAGIC            0xFD2FB528u
#define ZO_MAGIC_SKIP_START 0x184D2A50u
#define ZO_MAGIC_SKIP_MASK  0xFFFFFFF0u
#define ZO_MAGIC_DICT       0xEC30A437u
#define ZO_BLOCKSIZE_MAX    (128 * 1024)
#define ZO_WLOG_ABSMIN      10
#define ZO_WLOG_MAX         31
#define ZO_MINMATCH         3
#define ZO_MaxLL  35
#define ZO_MaxML  52
#define ZO_MaxOff 31
#define ZO_DefaultMaxOff 28
#define ZO_MaxSeq 52
#define ZO_LLFSELog 9
#define ZO_MLFSELog 9
#define ZO_OffFSELog 8
#define ZO_LitHufLog 11
#define ZO_LONGNBSEQ 0x7F00
#define ZO_HUF_TABLELOG_MAX 12
#define ZO_FSE_MIN_TABLELOG 5
#define ZO_FSE_MAX_TABLELOG 12
#define ZO_FSE_TABLESTEP(ts) (((ts) >> 1) + ((ts) >> 3) + 3)
enum { zo_set_basic = 0, zo_set_rle = 1, zo_set_compressed = 2, zo_set_repeat = 3 };
enum { zo_bt_raw = 0, zo_bt_rle = 1, zo_bt_compressed = 2, zo_bt_reserved = 3 };
enum { zo_rep_none = 0, zo_rep_check = 1, zo_rep_valid = 2 };   /* FSE_repeat / HUF_repeat */

extern const u8  zo_LL_bits[ZO_MaxLL + 1];
extern const u8  zo_ML_bits[ZO_MaxML + 1];
extern const s16 zo_LL_defaultNorm[ZO_MaxLL + 1];
extern const s16 zo_ML_defaultNorm[ZO_MaxML + 1];
extern const s16 zo_OF_defaultNorm[ZO_DefaultMaxOff + 1];
#define ZO_LL_DEFAULTNORMLOG 6
#define ZO_ML_DEFAULTNORMLOG 6
#define ZO_OF_DEFAULTNORMLOG 5

/* ---- little helpers (mem.h, bits.h) ---- */
static inline u32 zo_rd16(const void* p) { const u8* b = (const u8*)p; return (u32)b[0] | ((u32)b[1] << 8); }
static inline u32 zo_rd24(const void* p) { const u8* b = (const u8*)p; return zo_rd16(b) | ((u32)b[2] << 16); }
static inline u32 zo_rd32(const void* p) { u32 v; memcpy(&v, p, 4); return v; }   /* host is little-endian */
static inline u64 zo_rd64(const void* p) { u64 v; memcpy(&v, p, 8); return v; }
static inline void zo_wr16(void* p, u32 v) { u8* b = (u8*)p; b[0] = (u8)v; b[1] = (u8)(v >> 8); }
static inline void zo_wr24(void* p, u32 v) { u8* b = (u8*)p; b[0] = (u8)v; b[1] = (u8)(v >> 8); b[2] = (u8)(v >> 16); }
static inline void zo_wr32(void* p, u32 v) { memcpy(p, &v, 4); }
static inline void zo_wr64(void* p, u64 v) { memcpy(p, &v, 8); }
static inline u32 zo_highbit32(u32 v) { return 31u - (u32)__builtin_clz(v); }   /* bits.h ZSTD_highbit32; v != 0 */

/* ---- FSE compression table (fse_compress.c:69-215 lays the same data out in one U32 array) ---- */
typedef struct {
    u16 tableLog;
    u16 maxSV;
    u16 nextState[1 << ZO_FSE_MAX_TABLELOG];      /* "tableU16" */
    int deltaFindState[256];
    u32 deltaNbBits[256];
} zo_fse_ctable;

/* compact variant for the three sequence tables (<= 2^9 states, <= 53 symbols) */
typedef struct {
    u16 tableLog;
    u16 maxSV;
    u16 nextState[512];
    int deltaFindState[ZO_MaxSeq + 1];
    u32 deltaNbBits[ZO_MaxSeq + 1];
} zo_seq_ctable;

/* ---- Huffman compression table: nbBits + canonical value per symbol (huf_compress.c:189-222) ---- */
typedef struct {
    u8  nbBits[256];
    u16 val[256];
    u32 tableLog;      /* CTable[0] */
} zo_huf_ctable;

/* ---- entropy state carried from block to block (zstd_compress_internal.h ZSTD_entropyCTables_t) ---- */
typedef struct {
    zo_huf_ctable huf;
    int           huf_repeat;
    zo_seq_ctable ll, of, ml;
    int           ll_repeat, of_repeat, ml_repeat;
} zo_entropy;

typedef struct { zo_entropy entropy; u32 rep[3]; } zo_cblockstate;

/* ---- LSB-first bit writer (bitstream.h:154-241; flush points never change the bytes) ----
 * `end` bounds the output like the reference's endPtr does (bitstream.h:203-212, huf_compress.c:885-950): bytes beyond it are
 * dropped and the stream closes as 0, "does not fit" -- what the callers turn into raw literals / a raw block. */
typedef struct { u8* start; u8* p; u8* end; u64 acc; u32 nb; int over; } zo_bitw;
static inline void zo_bw_init(zo_bitw* w, void* dst, void* end) { w->start = w->p = (u8*)dst; w->end = (u8*)end; w->acc = 0; w->nb = 0; w->over = 0; }
static inline void zo_bw_put(zo_bitw* w, u8 b) { if (w->end && w->p >= w->end) w->over = 1; else *w->p++ = b; }
static inline void zo_bw_add(zo_bitw* w, u64 value, u32 nbBits)
{   /* BIT_addBits masks the value (bitstream.h:176-187) */
    if (nbBits == 0) return;
    value &= (nbBits >= 64) ? ~(u64)0 : (((u64)1 << nbBits) - 1);
    w->acc |= value << w->nb;
    w->nb += nbBits;
    while (w->nb >= 8) { zo_bw_put(w, (u8)w->acc); w->acc >>= 8; w->nb -= 8; }
}
static inline size_t zo_bw_close(zo_bitw* w)
{   /* BIT_closeCStream: end mark '1' then zero padding (bitstream.h:229-241); 0 when the stream did not fit */
    zo_bw_add(w, 1, 1);
    if (w->nb) { zo_bw_put(w, (u8)w->acc); w->acc = 0; w->nb = 0; }
    return w->over ? 0 : (size_t)(w->p - w->start);
}

/* ---- compression parameters ---- */
typedef struct { u32 wlog, clog, hlog, slog, mml, tlen, strat; } zo_cparams;
enum { zo_cpm_noAttachDict = 0, zo_cpm_attachDict = 1, zo_cpm_createCDict = 2, zo_cpm_unknown = 3 };
#define ZO_CONTENTSIZE_UNKNOWN ((u64)0 - 1)
#define ZO_CONTENTSIZE_ERROR   ((u64)0 - 2)

/* shared entropy-encoder entry points (zo_entropy_enc.c) */
u32    zo_hist(u32* count, u32* maxSVPtr, const u8* src, size_t n);
u32    zo_fse_optimalTableLog(u32 maxTableLog, size_t srcSize, u32 maxSV, u32 minus);
size_t zo_fse_normalizeCount(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSV, u32 useLowProb);
size_t zo_fse_writeNCount(u8* dst, size_t cap, const s16* norm, u32 maxSV, u32 tableLog);
void   zo_fse_buildCTable(u16* nextState, int* dfs, u32* dnb, const s16* norm, u32 maxSV, u32 tableLog);
size_t zo_fse_readNCount(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const u8* src, size_t n);
size_t zo_huf_readStats(u8* weights, size_t hwSize, u32* rankStats, u32* nbSymbolsPtr, u32* tableLogPtr,
                        const u8* src, size_t n);

#endif
