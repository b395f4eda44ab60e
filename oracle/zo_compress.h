/* zo_compress.h -- oracle: compressor-side shared types.  TEST INFRASTRUCTURE ONLY (see zo_common.h). */
#ifndef ZO_COMPRESS_H
#define ZO_COMPRESS_H
#include "zo_common.h"

/* zstd_internal.h seqDef / seqStore_t */
typedef struct { u32 offBase; u16 litLength; u16 mlBase; } zo_seqdef;
typedef struct {
    zo_seqdef* seq; u32 nbSeq; u32 maxNbSeq;
    u8* lit; size_t litSize;
    u32 longLengthType;   /* 0 none, 1 literalLength, 2 matchLength */
    u32 longLengthPos;
} zo_seqstore;

/* digested dictionary for compression (zstd_compress.c ZSTD_CDict_s, the parts this path reads) */
typedef struct zo_cdict_s {
    u8*  dictBuf; size_t dictSize;        /* private copy of the whole dictionary (cdict->dictContent / dictContentSize) */
    const u8* content; size_t contentSize;/* the part that entered the window: after the entropy header, if any */
    int  isLoaded;                        /* 0 when the dictionary was ignored (NULL or < 8 bytes) */
    u32  dictID;
    zo_cparams cp;                        /* cpm_createCDict parameters */
    u32* hashTable;                       /* ZSTD_fast / ZSTD_dfast: tagged, (index << 8) | tag (zstd_fast.c:15-48, zstd_double_fast.c:15-49);
                                           * greedy .. lazy2: plain indices (hash chain heads, or the row finder's cells) */
    u32* chainTable;                      /* ZSTD_dfast: the short-hash table (tagged); hash-chain finder: the chain */
    u8*  tagTable;                        /* row finder: one tag byte per cell, byte 0 of a row its head */
    int  useRow;                          /* cdict->useRowMatchFinder (zstd_compress.c:5420-5468): greedy .. lazy2 with windowLog > 14 */
    u32  endIndex;                        /* window.nextSrc - base == 2 + contentSize */
    zo_cblockstate cbs;                   /* entropy tables + reps loaded from the dictionary */
    int  level;                           /* cdict->compressionLevel */
} zo_cdict;

size_t zo_entropyCompressSeqStore(const zo_seqstore* ss, const zo_entropy* prev, zo_entropy* next,
                                  int literalsDisabled, u8* dst, size_t dstCapacity, size_t srcSize, u8* codeWksp, int strategy);

#endif
