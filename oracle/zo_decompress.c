/* zo_decompress.c -- oracle: frame, block, Huffman and FSE-sequence decoder.
 * TEST INFRASTRUCTURE ONLY (see zo_common.h).  Restates zstd_decompress.c, zstd_decompress_block.c,
 * huf_decompress.c (X1 semantics), fse_decompress.c, entropy_common.c and bitstream.h of the reference.
 * Legacy (v0.4-0.7) frames are out of scope and reported as prefix_unknown.
 */
#include "zo_common.h"
#include "zstd_oracle.h"

/* ------------------------------------------------------------------ backward bit reader */
/* bitstream.h:253-432, restated with the same 64-bit container so that reads past the start of a
 * (corrupted) stream behave like the reference's. */
typedef struct { u64 container; u32 bitsConsumed; const u8* ptr; const u8* start; const u8* limitPtr; } zo_bitr;
enum { ZO_BIT_unfinished = 0, ZO_BIT_endOfBuffer = 1, ZO_BIT_completed = 2, ZO_BIT_overflow = 3 };

static size_t zo_br_init(zo_bitr* b, const u8* src, size_t srcSize)
{
    if (srcSize < 1) { memset(b, 0, sizeof(*b)); return ZO_ERR(srcSize_wrong); }
    b->start = src;
    b->limitPtr = src + 8;
    if (srcSize >= 8) {
        u8 const lastByte = src[srcSize - 1];
        b->ptr = src + srcSize - 8;
        b->container = zo_rd64(b->ptr);
        b->bitsConsumed = lastByte ? 8 - zo_highbit32(lastByte) : 0;
        if (lastByte == 0) return ZO_ERR(GENERIC);
    } else {
        u8 const lastByte = src[srcSize - 1];
        size_t i;
        b->ptr = src;
        b->container = 0;
        for (i = 0; i < srcSize; i++) b->container |= (u64)src[i] << (8 * i);
        b->bitsConsumed = lastByte ? 8 - zo_highbit32(lastByte) : 0;
        if (lastByte == 0) return ZO_ERR(corruption_detected);
        b->bitsConsumed += (u32)(8 - srcSize) * 8;
    }
    return srcSize;
}
static u64 zo_br_look(const zo_bitr* b, u32 nbBits)
{   /* BIT_lookBits, x86-64 form of BIT_getMiddleBits (bitstream.h:306-341) */
    u32 const start = 64 - b->bitsConsumed - nbBits;
    return (b->container >> (start & 63)) & ((((u64)1) << nbBits) - 1);
}
static u64 zo_br_lookFast(const zo_bitr* b, u32 nbBits)
{   /* bitstream.h:345-350, nbBits >= 1 */
    return (b->container << (b->bitsConsumed & 63)) >> (((63 + 1) - nbBits) & 63);
}
static u64 zo_br_read(zo_bitr* b, u32 nbBits) { u64 const v = zo_br_look(b, nbBits); b->bitsConsumed += nbBits; return v; }
static u64 zo_br_readFast(zo_bitr* b, u32 nbBits) { u64 const v = zo_br_lookFast(b, nbBits); b->bitsConsumed += nbBits; return v; }
static int zo_br_reload(zo_bitr* b)
{
    if (b->bitsConsumed > 64) return ZO_BIT_overflow;
    if (b->ptr >= b->limitPtr) {
        b->ptr -= b->bitsConsumed >> 3;
        b->bitsConsumed &= 7;
        b->container = zo_rd64(b->ptr);
        return ZO_BIT_unfinished;
    }
    if (b->ptr == b->start) return (b->bitsConsumed < 64) ? ZO_BIT_endOfBuffer : ZO_BIT_completed;
    {   u32 nbBytes = b->bitsConsumed >> 3;
        int result = ZO_BIT_unfinished;
        if (b->ptr - nbBytes < b->start) { nbBytes = (u32)(b->ptr - b->start); result = ZO_BIT_endOfBuffer; }
        b->ptr -= nbBytes;
        b->bitsConsumed -= nbBytes * 8;
        b->container = zo_rd64(b->ptr);
        return result;
    }
}
static int zo_br_end(const zo_bitr* b) { return (b->ptr == b->start) && (b->bitsConsumed == 64); }

/* ------------------------------------------------------------------ FSE headers */
/* entropy_common.c:43-188 FSE_readNCount_body, restated as a plain LSB-first reader over a zero-padded copy */
size_t zo_fse_readNCount(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const u8* src, size_t hbSize)
{
    u32 const maxSV1 = *maxSVPtr + 1;
    u32 charnum = 0;
    int nbBits, remaining, threshold, previous0 = 0;
    size_t bitPos = 0;                 /* bits consumed from src */
    size_t const bitEnd = hbSize * 8;
    if (hbSize == 0) return ZO_ERR(srcSize_wrong);
    memset(norm, 0, maxSV1 * sizeof(norm[0]));
    {   u32 first = 0; size_t i; for (i = 0; i < 4 && i < hbSize; i++) first |= (u32)src[i] << (8 * i);
        nbBits = (int)(first & 0xF) + ZO_FSE_MIN_TABLELOG;
    }
    if (nbBits > 15) return ZO_ERR(tableLog_tooLarge);
    bitPos = 4;
    *tableLogPtr = (u32)nbBits;
    remaining = (1 << nbBits) + 1;
    threshold = 1 << nbBits;
    nbBits++;
    for (;;) {
        u32 bitStream;
        if (previous0) {
            for (;;) {   /* 2-bit repeat codes: 3 means "3 more zeros and continue" */
                u32 two;
                { size_t by = bitPos >> 3; u32 sh = (u32)(bitPos & 7); u32 v = 0; size_t i;
                  for (i = 0; i < 2; i++) if (by + i < hbSize) v |= (u32)src[by + i] << (8 * i);
                  two = (v >> sh) & 3; }
                bitPos += 2;
                charnum += two;
                if (two != 3) break;
                if (bitPos > bitEnd + 32) return ZO_ERR(corruption_detected);
            }
            if (charnum >= maxSV1) break;
        }
        {   size_t by = bitPos >> 3; u32 sh = (u32)(bitPos & 7); u64 v = 0; size_t i;
            for (i = 0; i < 5; i++) if (by + i < hbSize) v |= (u64)src[by + i] << (8 * i);
            bitStream = (u32)(v >> sh);
        }
        {   int const max = (2 * threshold - 1) - remaining;
            int count;
            if ((bitStream & (u32)(threshold - 1)) < (u32)max) {
                count = (int)(bitStream & (u32)(threshold - 1));
                bitPos += (size_t)(nbBits - 1);
            } else {
                count = (int)(bitStream & (u32)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                bitPos += (size_t)nbBits;
            }
            count--;
            if (count >= 0) remaining -= count; else remaining += count;
            norm[charnum++] = (s16)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (int)zo_highbit32((u32)remaining) + 1;
                threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
        }
    }
    if (remaining != 1) return ZO_ERR(corruption_detected);
    if (charnum > maxSV1) return ZO_ERR(maxSymbolValue_tooSmall);
    if (bitPos > bitEnd) return ZO_ERR(corruption_detected);
    *maxSVPtr = charnum - 1;
    return (bitPos + 7) >> 3;
}

/* FSE decoding table for byte symbols (fse_decompress.c:60-163) */
typedef struct { u16 newState; u8 symbol; u8 nbBits; } zo_fse_dcell;
static size_t zo_fse_buildDTable(zo_fse_dcell* dt, const s16* norm, u32 maxSV, u32 tableLog)
{
    u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1, step = ZO_FSE_TABLESTEP(tableSize);
    u16 symbolNext[256];
    u32 highThreshold = tableSize - 1, s, position = 0, u;
    for (s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { dt[highThreshold--].symbol = (u8)s; symbolNext[s] = 1; }
        else symbolNext[s] = (u16)norm[s];
    }
    for (s = 0; s <= maxSV; s++) {
        int i;
        for (i = 0; i < norm[s]; i++) {
            dt[position].symbol = (u8)s;
            position = (position + step) & tableMask;
            while (position > highThreshold) position = (position + step) & tableMask;
        }
    }
    if (position != 0) return ZO_ERR(GENERIC);
    for (u = 0; u < tableSize; u++) {
        u8 const symbol = dt[u].symbol;
        u32 const nextState = symbolNext[symbol]++;
        dt[u].nbBits = (u8)(tableLog - zo_highbit32(nextState));
        dt[u].newState = (u16)((nextState << dt[u].nbBits) - tableSize);
    }
    return 0;
}
/* fse_decompress.c:177-286 FSE_decompress_wksp (maxLog = 6: Huffman weights) */
static size_t zo_fse_decompressWeights(u8* dst, size_t dstCapacity, const u8* cSrc, size_t cSrcSize)
{
    s16 norm[256];
    u32 tableLog, maxSV = 255;
    zo_fse_dcell dt[1 << 6];
    u8* op = dst;
    u8* const omax = dst + dstCapacity;
    zo_bitr bitD;
    u32 state1, state2;
    {   size_t const n = zo_fse_readNCount(norm, &maxSV, &tableLog, cSrc, cSrcSize);
        if (zo_isErr(n)) return n;
        if (tableLog > 6) return ZO_ERR(tableLog_tooLarge);
        cSrc += n; cSrcSize -= n;
    }
    { size_t const e = zo_fse_buildDTable(dt, norm, maxSV, tableLog); if (zo_isErr(e)) return e; }
    { size_t const e = zo_br_init(&bitD, cSrc, cSrcSize); if (zo_isErr(e)) return e; }
    state1 = (u32)zo_br_read(&bitD, tableLog); zo_br_reload(&bitD);
    state2 = (u32)zo_br_read(&bitD, tableLog); zo_br_reload(&bitD);
#define ZO_FSE_GET(st) do { zo_fse_dcell const c = dt[st]; *op++ = c.symbol; st = c.newState + (u32)zo_br_read(&bitD, c.nbBits); } while (0)
    /* the 4-symbols-per-loop body at :200-217 decodes s1,s2,s1,s2 with reloads that never change values;
     * the tail at :221-234 defines termination */
    for (; (zo_br_reload(&bitD) == ZO_BIT_unfinished) & (op < omax - 3); ) {
        ZO_FSE_GET(state1); ZO_FSE_GET(state2); ZO_FSE_GET(state1); ZO_FSE_GET(state2);
    }
    for (;;) {
        if (op > (omax - 2)) return ZO_ERR(dstSize_tooSmall);
        ZO_FSE_GET(state1);
        if (zo_br_reload(&bitD) == ZO_BIT_overflow) { ZO_FSE_GET(state2); break; }
        if (op > (omax - 2)) return ZO_ERR(dstSize_tooSmall);
        ZO_FSE_GET(state2);
        if (zo_br_reload(&bitD) == ZO_BIT_overflow) { ZO_FSE_GET(state1); break; }
    }
#undef ZO_FSE_GET
    return (size_t)(op - dst);
}

/* entropy_common.c:244-307 HUF_readStats_body */
size_t zo_huf_readStats(u8* huffWeight, size_t hwSize, u32* rankStats, u32* nbSymbolsPtr, u32* tableLogPtr,
                        const u8* src, size_t srcSize)
{
    u32 weightTotal = 0;
    size_t iSize, oSize;
    if (!srcSize) return ZO_ERR(srcSize_wrong);
    iSize = src[0];
    if (iSize >= 128) {
        u32 n;
        oSize = iSize - 127;
        iSize = ((oSize + 1) / 2);
        if (iSize + 1 > srcSize) return ZO_ERR(srcSize_wrong);
        if (oSize >= hwSize) return ZO_ERR(corruption_detected);
        for (n = 0; n < oSize; n += 2) { huffWeight[n] = src[1 + n / 2] >> 4; huffWeight[n + 1] = src[1 + n / 2] & 15; }
    } else {
        if (iSize + 1 > srcSize) return ZO_ERR(srcSize_wrong);
        oSize = zo_fse_decompressWeights(huffWeight, hwSize - 1, src + 1, iSize);
        if (zo_isErr(oSize)) return oSize;
    }
    memset(rankStats, 0, (ZO_HUF_TABLELOG_MAX + 1) * sizeof(u32));
    {   u32 n;
        for (n = 0; n < oSize; n++) {
            if (huffWeight[n] > ZO_HUF_TABLELOG_MAX) return ZO_ERR(corruption_detected);
            rankStats[huffWeight[n]]++;
            weightTotal += (1u << huffWeight[n]) >> 1;
        }
    }
    if (weightTotal == 0) return ZO_ERR(corruption_detected);
    {   u32 const tableLog = zo_highbit32(weightTotal) + 1;
        if (tableLog > ZO_HUF_TABLELOG_MAX) return ZO_ERR(corruption_detected);
        *tableLogPtr = tableLog;
        {   u32 const total = 1u << tableLog;
            u32 const rest = total - weightTotal;
            u32 const verif = 1u << zo_highbit32(rest);
            u32 const lastWeight = zo_highbit32(rest) + 1;
            if (verif != rest) return ZO_ERR(corruption_detected);
            huffWeight[oSize] = (u8)lastWeight;
            rankStats[lastWeight]++;
        }
    }
    if ((rankStats[1] < 2) || (rankStats[1] & 1)) return ZO_ERR(corruption_detected);
    *nbSymbolsPtr = (u32)(oSize + 1);
    return iSize + 1;
}

/* ------------------------------------------------------------------ Huffman decoding (single-symbol table) */
typedef struct { u8 nbBits; u8 byte; } zo_huf_dcell;
typedef struct { u32 tableLog; zo_huf_dcell cell[1 << ZO_HUF_TABLELOG_MAX]; } zo_huf_dtable;

/* huf_decompress.c:358-492 HUF_readDTableX1_wksp.  The reference additionally rescales to an 11-bit table
 * (:325-348, a speed choice); decoded symbols are the same. */
static size_t zo_huf_readDTable(zo_huf_dtable* dt, const u8* src, size_t srcSize)
{
    u8 weights[256];
    u32 rankVal[ZO_HUF_TABLELOG_MAX + 1], rankStart[ZO_HUF_TABLELOG_MAX + 2];
    u32 tableLog = 0, nbSymbols = 0, n, w;
    size_t const iSize = zo_huf_readStats(weights, 256, rankVal, &nbSymbols, &tableLog, src, srcSize);
    if (zo_isErr(iSize)) return iSize;
    if (tableLog > ZO_HUF_TABLELOG_MAX) return ZO_ERR(tableLog_tooLarge);
    dt->tableLog = tableLog;
    {   u32 next = 0;
        for (w = 1; w <= tableLog; w++) { rankStart[w] = next; next += rankVal[w] << (w - 1); }
    }
    for (n = 0; n < nbSymbols; n++) {
        u32 const wt = weights[n];
        if (wt) {
            u32 const length = (1u << wt) >> 1, u0 = rankStart[wt];
            u32 u;
            for (u = u0; u < u0 + length; u++) { dt->cell[u].byte = (u8)n; dt->cell[u].nbBits = (u8)(tableLog + 1 - wt); }
            rankStart[wt] += length;
        }
    }
    return iSize;
}
/* huf_decompress.c:514-541 HUF_decodeStreamX1 */
static void zo_huf_decodeStream(u8* p, zo_bitr* b, u8* const pEnd, const zo_huf_dtable* dt)
{
    u32 const dtLog = dt->tableLog;
#define ZO_HUF_SYM() do { size_t const v = (size_t)zo_br_lookFast(b, dtLog); *p++ = dt->cell[v].byte; b->bitsConsumed += dt->cell[v].nbBits; } while (0)
    if ((pEnd - p) > 3) {
        while ((zo_br_reload(b) == ZO_BIT_unfinished) & (p < pEnd - 3)) { ZO_HUF_SYM(); ZO_HUF_SYM(); ZO_HUF_SYM(); ZO_HUF_SYM(); }
    } else zo_br_reload(b);
    while (p < pEnd) ZO_HUF_SYM();
#undef ZO_HUF_SYM
}
/* huf_decompress.c:543-564 */
static size_t zo_huf_decompress1X(u8* dst, size_t dstSize, const u8* cSrc, size_t cSrcSize, const zo_huf_dtable* dt)
{
    zo_bitr b;
    { size_t const e = zo_br_init(&b, cSrc, cSrcSize); if (zo_isErr(e)) return e; }
    zo_huf_decodeStream(dst, &b, dst + dstSize, dt);
    if (!zo_br_end(&b)) return ZO_ERR(corruption_detected);
    return dstSize;
}
/* huf_decompress.c:570-666: each of the four streams regenerates its own segment */
static size_t zo_huf_decompress4X(u8* dst, size_t dstSize, const u8* cSrc, size_t cSrcSize, const zo_huf_dtable* dt)
{
    if (cSrcSize < 10) return ZO_ERR(corruption_detected);
    {   size_t const length1 = zo_rd16(cSrc), length2 = zo_rd16(cSrc + 2), length3 = zo_rd16(cSrc + 4);
        size_t const length4 = cSrcSize - (length1 + length2 + length3 + 6);
        const u8* const istart1 = cSrc + 6;
        const u8* const istart2 = istart1 + length1;
        const u8* const istart3 = istart2 + length2;
        const u8* const istart4 = istart3 + length3;
        size_t const segmentSize = (dstSize + 3) / 4;
        u8* const op2 = dst + segmentSize;
        u8* const op3 = op2 + segmentSize;
        u8* const op4 = op3 + segmentSize;
        u8* const oend = dst + dstSize;
        zo_bitr b1, b2, b3, b4;
        if (length4 > cSrcSize) return ZO_ERR(corruption_detected);
        if (op4 > oend) return ZO_ERR(corruption_detected);
        if (dstSize < 6) return ZO_ERR(corruption_detected);
        { size_t e = zo_br_init(&b1, istart1, length1); if (zo_isErr(e)) return e; }
        { size_t e = zo_br_init(&b2, istart2, length2); if (zo_isErr(e)) return e; }
        { size_t e = zo_br_init(&b3, istart3, length3); if (zo_isErr(e)) return e; }
        { size_t e = zo_br_init(&b4, istart4, length4); if (zo_isErr(e)) return e; }
        zo_huf_decodeStream(dst, &b1, op2, dt);
        zo_huf_decodeStream(op2, &b2, op3, dt);
        zo_huf_decodeStream(op3, &b3, op4, dt);
        zo_huf_decodeStream(op4, &b4, oend, dt);
        if (!(zo_br_end(&b1) & zo_br_end(&b2) & zo_br_end(&b3) & zo_br_end(&b4))) return ZO_ERR(corruption_detected);
        return dstSize;
    }
}

/* ------------------------------------------------------------------ sequence decoding tables */
typedef struct { u16 nextState; u8 nbAdditionalBits; u8 nbBits; u32 baseValue; } zo_seqsym;
typedef struct { u32 tableLog; zo_seqsym cell[512]; } zo_seqtable;

static u32 zo_LL_base[ZO_MaxLL + 1], zo_ML_base[ZO_MaxML + 1], zo_OF_base[ZO_MaxOff + 1];
static u8  zo_OF_bits[ZO_MaxOff + 1];
static int zo_bases_ready = 0;
static void zo_initBases(void)
{   /* zstd_decompress_internal.h:31-57: bases are cumulative over the extra-bit widths */
    u32 i, v;
    if (zo_bases_ready) return;
    for (i = 0, v = 0; i <= ZO_MaxLL; i++) { zo_LL_base[i] = v; v += 1u << zo_LL_bits[i]; }
    for (i = 0, v = 3; i <= ZO_MaxML; i++) { zo_ML_base[i] = v; v += 1u << zo_ML_bits[i]; }
    zo_OF_base[0] = 0; zo_OF_base[1] = 1;
    for (i = 2; i <= ZO_MaxOff; i++) zo_OF_base[i] = (1u << i) - 3;
    for (i = 0; i <= ZO_MaxOff; i++) zo_OF_bits[i] = (u8)i;
    zo_bases_ready = 1;
}
/* zstd_decompress_block.c:455-573 ZSTD_buildFSETable_body */
static void zo_buildSeqTable(zo_seqtable* t, const s16* norm, u32 maxSV, const u32* baseValue, const u8* nbAddBits, u32 tableLog)
{
    u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1, step = ZO_FSE_TABLESTEP(tableSize);
    u16 symbolNext[ZO_MaxSeq + 1];
    u32 highThreshold = tableSize - 1, s, position = 0, u;
    t->tableLog = tableLog;
    for (s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { t->cell[highThreshold--].baseValue = s; symbolNext[s] = 1; }
        else symbolNext[s] = (u16)norm[s];
    }
    for (s = 0; s <= maxSV; s++) {
        int i;
        for (i = 0; i < norm[s]; i++) {
            t->cell[position].baseValue = s;
            position = (position + step) & tableMask;
            while (position > highThreshold) position = (position + step) & tableMask;
        }
    }
    for (u = 0; u < tableSize; u++) {
        u32 const symbol = t->cell[u].baseValue;
        u32 const nextState = symbolNext[symbol]++;
        t->cell[u].nbBits = (u8)(tableLog - zo_highbit32(nextState));
        t->cell[u].nextState = (u16)((nextState << t->cell[u].nbBits) - tableSize);
        t->cell[u].nbAdditionalBits = nbAddBits[symbol];
        t->cell[u].baseValue = baseValue[symbol];
    }
}

/* ------------------------------------------------------------------ decoder context */
struct zo_ddict_s {
    u8* raw; size_t rawSize;          /* private copy of the whole dictionary */
    const u8* content; size_t contentSize;
    u32 dictID; int entropyPresent;
    zo_huf_dtable huf; zo_seqtable ll, of, ml; u32 rep[3];
};
typedef struct {
    zo_huf_dtable hufTable; const zo_huf_dtable* hufPtr;
    zo_seqtable LLTable, OFTable, MLTable;
    const zo_seqtable *LLTptr, *OFTptr, *MLTptr;
    zo_seqtable defLL, defOF, defML;
    u32 rep[3];
    int litEntropy, fseEntropy;
    u32 dictID;
    const u8 *prefixStart, *virtualStart, *dictEnd, *previousDstEnd;
    u8* litBuf; const u8* litPtr; size_t litSize;
    u32 blockSizeMax;
} zo_dctx;

/* zstd_decompress_block.c:120-322 ZSTD_decodeLiteralsBlock (literal bytes land in a private buffer) */
static size_t zo_decodeLiterals(zo_dctx* d, const u8* istart, size_t srcSize, const u8* dst, size_t dstCapacity)
{
    size_t const expectedWriteSize = dstCapacity < ZO_BLOCKSIZE_MAX ? dstCapacity : ZO_BLOCKSIZE_MAX;
    if (srcSize < 2) return ZO_ERR(corruption_detected);
    switch (istart[0] & 3) {
    case zo_set_repeat:
        if (d->litEntropy == 0) return ZO_ERR(dictionary_corrupted);
        /* fallthrough */
    case zo_set_compressed: {
        size_t lhSize, litSize, litCSize, r;
        u32 singleStream = 0;
        u32 const lhlCode = (istart[0] >> 2) & 3;
        u32 lhc;
        int const isRepeat = (istart[0] & 3) == zo_set_repeat;
        if (srcSize < 5) return ZO_ERR(corruption_detected);
        lhc = zo_rd32(istart);
        switch (lhlCode) {
        case 0: case 1: default: singleStream = !lhlCode; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; break;
        case 2: lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; break;
        case 3: lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((size_t)istart[4] << 10); break;
        }
        if (litSize > 0 && dst == NULL) return ZO_ERR(dstSize_tooSmall);
        if (litSize > ZO_BLOCKSIZE_MAX) return ZO_ERR(corruption_detected);
        if (!singleStream && litSize < 6) return ZO_ERR(literals_headerWrong);
        if (litCSize + lhSize > srcSize) return ZO_ERR(corruption_detected);
        if (expectedWriteSize < litSize) return ZO_ERR(dstSize_tooSmall);
        if (isRepeat) {
            r = singleStream ? zo_huf_decompress1X(d->litBuf, litSize, istart + lhSize, litCSize, d->hufPtr)
                             : zo_huf_decompress4X(d->litBuf, litSize, istart + lhSize, litCSize, d->hufPtr);
        } else {
            /* huf_decompress.c HUF_decompress1X1_DCtx_wksp / HUF_decompress4X_hufOnly_wksp: table then streams */
            const u8* ip = istart + lhSize;
            size_t cSrcSize = litCSize;
            size_t hSize;
            if (!singleStream) { if (litSize == 0) return ZO_ERR(corruption_detected); if (cSrcSize == 0) return ZO_ERR(corruption_detected); }
            hSize = zo_huf_readDTable(&d->hufTable, ip, cSrcSize);
            if (zo_isErr(hSize)) r = hSize;
            else if (hSize >= cSrcSize) r = ZO_ERR(srcSize_wrong);
            else {
                ip += hSize; cSrcSize -= hSize;
                r = singleStream ? zo_huf_decompress1X(d->litBuf, litSize, ip, cSrcSize, &d->hufTable)
                                 : zo_huf_decompress4X(d->litBuf, litSize, ip, cSrcSize, &d->hufTable);
            }
        }
        if (zo_isErr(r)) return ZO_ERR(corruption_detected);
        d->litPtr = d->litBuf; d->litSize = litSize; d->litEntropy = 1;
        if (!isRepeat) d->hufPtr = &d->hufTable;
        return litCSize + lhSize;
    }
    case zo_set_basic: {
        size_t litSize, lhSize;
        u32 const lhlCode = (istart[0] >> 2) & 3;
        switch (lhlCode) {
        case 0: case 2: default: lhSize = 1; litSize = istart[0] >> 3; break;
        case 1: lhSize = 2; litSize = zo_rd16(istart) >> 4; break;
        case 3: lhSize = 3; if (srcSize < 3) return ZO_ERR(corruption_detected); litSize = zo_rd24(istart) >> 4; break;
        }
        if (litSize > 0 && dst == NULL) return ZO_ERR(dstSize_tooSmall);
        if (expectedWriteSize < litSize) return ZO_ERR(dstSize_tooSmall);
        if (litSize + lhSize > srcSize) return ZO_ERR(corruption_detected);
        d->litPtr = istart + lhSize; d->litSize = litSize;
        return lhSize + litSize;
    }
    default: {   /* set_rle */
        u32 const lhlCode = (istart[0] >> 2) & 3;
        size_t litSize, lhSize;
        switch (lhlCode) {
        case 0: case 2: default: lhSize = 1; litSize = istart[0] >> 3; break;
        case 1: lhSize = 2; if (srcSize < 3) return ZO_ERR(corruption_detected); litSize = zo_rd16(istart) >> 4; break;
        case 3: lhSize = 3; if (srcSize < 4) return ZO_ERR(corruption_detected); litSize = zo_rd24(istart) >> 4; break;
        }
        if (litSize > 0 && dst == NULL) return ZO_ERR(dstSize_tooSmall);
        if (litSize > ZO_BLOCKSIZE_MAX) return ZO_ERR(corruption_detected);
        if (expectedWriteSize < litSize) return ZO_ERR(dstSize_tooSmall);
        memset(d->litBuf, istart[lhSize], litSize);
        d->litPtr = d->litBuf; d->litSize = litSize;
        return lhSize + 1;
    } }
}

/* zstd_decompress_block.c:617-663 ZSTD_buildSeqTable */
static size_t zo_buildSeqTableFromHeader(zo_seqtable* space, const zo_seqtable** ptr, int type, u32 max, u32 maxLog,
                                         const u8* src, size_t srcSize, const u32* baseValue, const u8* nbAddBits,
                                         const zo_seqtable* defaultTable, int flagRepeatTable)
{
    switch (type) {
    case zo_set_rle:
        if (!srcSize) return ZO_ERR(srcSize_wrong);
        if (src[0] > max) return ZO_ERR(corruption_detected);
        space->tableLog = 0;
        space->cell[0].nbBits = 0; space->cell[0].nextState = 0;
        space->cell[0].nbAdditionalBits = nbAddBits[src[0]]; space->cell[0].baseValue = baseValue[src[0]];
        *ptr = space;
        return 1;
    case zo_set_basic: *ptr = defaultTable; return 0;
    case zo_set_repeat: if (!flagRepeatTable) return ZO_ERR(corruption_detected); return 0;
    default: {
        u32 tableLog; s16 norm[ZO_MaxSeq + 1];
        size_t const headerSize = zo_fse_readNCount(norm, &max, &tableLog, src, srcSize);
        if (zo_isErr(headerSize)) return ZO_ERR(corruption_detected);
        if (tableLog > maxLog) return ZO_ERR(corruption_detected);
        zo_buildSeqTable(space, norm, max, baseValue, nbAddBits, tableLog);
        *ptr = space;
        return headerSize;
    } }
}
/* zstd_decompress_block.c:665-742 ZSTD_decodeSeqHeaders */
static size_t zo_decodeSeqHeaders(zo_dctx* d, int* nbSeqPtr, const u8* istart, size_t srcSize)
{
    const u8* const iend = istart + srcSize;
    const u8* ip = istart;
    int nbSeq;
    if (srcSize < 1) return ZO_ERR(srcSize_wrong);
    nbSeq = *ip++;
    if (!nbSeq) { *nbSeqPtr = 0; if (srcSize != 1) return ZO_ERR(srcSize_wrong); return 1; }
    if (nbSeq > 0x7F) {
        if (nbSeq == 0xFF) { if (ip + 2 > iend) return ZO_ERR(srcSize_wrong); nbSeq = (int)zo_rd16(ip) + ZO_LONGNBSEQ; ip += 2; }
        else { if (ip >= iend) return ZO_ERR(srcSize_wrong); nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
    }
    *nbSeqPtr = nbSeq;
    if (ip + 1 > iend) return ZO_ERR(srcSize_wrong);
    {   int const LLtype = *ip >> 6, OFtype = (*ip >> 4) & 3, MLtype = (*ip >> 2) & 3;
        ip++;
        {   size_t const h = zo_buildSeqTableFromHeader(&d->LLTable, &d->LLTptr, LLtype, ZO_MaxLL, ZO_LLFSELog, ip, (size_t)(iend - ip),
                                                        zo_LL_base, zo_LL_bits, &d->defLL, d->fseEntropy);
            if (zo_isErr(h)) return ZO_ERR(corruption_detected);
            ip += h; }
        {   size_t const h = zo_buildSeqTableFromHeader(&d->OFTable, &d->OFTptr, OFtype, ZO_MaxOff, ZO_OffFSELog, ip, (size_t)(iend - ip),
                                                        zo_OF_base, zo_OF_bits, &d->defOF, d->fseEntropy);
            if (zo_isErr(h)) return ZO_ERR(corruption_detected);
            ip += h; }
        {   size_t const h = zo_buildSeqTableFromHeader(&d->MLTable, &d->MLTptr, MLtype, ZO_MaxML, ZO_MLFSELog, ip, (size_t)(iend - ip),
                                                        zo_ML_base, zo_ML_bits, &d->defML, d->fseEntropy);
            if (zo_isErr(h)) return ZO_ERR(corruption_detected);
            ip += h; }
    }
    return (size_t)(ip - istart);
}

/* zstd_decompress_block.c:965-1060 + :873-914 ZSTD_execSequence(End): byte-exact semantics, no wild copies */
static size_t zo_execSequence(u8* op, u8* const oend, size_t litLength, size_t matchLength, size_t offset,
                              const u8** litPtr, const u8* const litLimit,
                              const u8* const prefixStart, const u8* const virtualStart, const u8* const dictEnd)
{
    u8* const oLitEnd = op + litLength;
    size_t const sequenceLength = litLength + matchLength;
    const u8* match = oLitEnd - offset;
    if (sequenceLength > (size_t)(oend - op)) return ZO_ERR(dstSize_tooSmall);
    if (litLength > (size_t)(litLimit - *litPtr)) return ZO_ERR(corruption_detected);
    memmove(op, *litPtr, litLength);
    op = oLitEnd;
    *litPtr += litLength;
    if (offset > (size_t)(oLitEnd - prefixStart)) {
        if (offset > (size_t)(oLitEnd - virtualStart)) return ZO_ERR(corruption_detected);
        match = dictEnd - (prefixStart - match);
        if (match + matchLength <= dictEnd) { memmove(oLitEnd, match, matchLength); return sequenceLength; }
        {   size_t const length1 = (size_t)(dictEnd - match);
            memmove(oLitEnd, match, length1);
            op = oLitEnd + length1;
            matchLength -= length1;
            match = prefixStart;
        }
    }
    while (matchLength--) *op++ = *match++;   /* overlap => pattern replication */
    return sequenceLength;
}

/* zstd_decompress_block.c:1569-1654 ZSTD_decompressSequences_body + :1185-1299 ZSTD_decodeSequence */
static size_t zo_decompressSequences(zo_dctx* d, u8* dst, size_t maxDstSize, const u8* seqStart, size_t seqSize, int nbSeq)
{
    u8* const ostart = dst;
    u8* const oend = ostart + maxDstSize;
    u8* op = ostart;
    const u8* litPtr = d->litPtr;
    const u8* const litEnd = litPtr + d->litSize;
    if (nbSeq) {
        zo_bitr b;
        u32 stLL, stOF, stML;
        size_t prevOffset[3];
        const zo_seqsym* const tLL = d->LLTptr->cell; const zo_seqsym* const tOF = d->OFTptr->cell; const zo_seqsym* const tML = d->MLTptr->cell;
        d->fseEntropy = 1;
        prevOffset[0] = d->rep[0]; prevOffset[1] = d->rep[1]; prevOffset[2] = d->rep[2];
        if (zo_isErr(zo_br_init(&b, seqStart, seqSize))) return ZO_ERR(corruption_detected);
        stLL = (u32)zo_br_read(&b, d->LLTptr->tableLog); zo_br_reload(&b);
        stOF = (u32)zo_br_read(&b, d->OFTptr->tableLog); zo_br_reload(&b);
        stML = (u32)zo_br_read(&b, d->MLTptr->tableLog); zo_br_reload(&b);
        for (;;) {
            zo_seqsym const ll = tLL[stLL], ml = tML[stML], of = tOF[stOF];
            size_t matchLength = ml.baseValue, litLength = ll.baseValue, offset;
            u8 const llBits = ll.nbAdditionalBits, mlBits = ml.nbAdditionalBits, ofBits = of.nbAdditionalBits;
            u8 const totalBits = (u8)(llBits + mlBits + ofBits);
            if (ofBits > 1) {
                offset = of.baseValue + (size_t)zo_br_readFast(&b, ofBits);
                prevOffset[2] = prevOffset[1]; prevOffset[1] = prevOffset[0]; prevOffset[0] = offset;
            } else {
                u32 const ll0 = (ll.baseValue == 0);
                if (ofBits == 0) {
                    offset = prevOffset[ll0];
                    prevOffset[1] = prevOffset[!ll0];
                    prevOffset[0] = offset;
                } else {
                    offset = of.baseValue + ll0 + (size_t)zo_br_readFast(&b, 1);
                    {   size_t temp = (offset == 3) ? prevOffset[0] - 1 : prevOffset[offset];
                        temp += !temp;
                        if (offset != 1) prevOffset[2] = prevOffset[1];
                        prevOffset[1] = prevOffset[0];
                        prevOffset[0] = offset = temp;
                    }
                }
            }
            if (mlBits > 0) matchLength += (size_t)zo_br_readFast(&b, mlBits);
            if (totalBits >= 57 - (ZO_LLFSELog + ZO_MLFSELog + ZO_OffFSELog)) zo_br_reload(&b);
            if (llBits > 0) litLength += (size_t)zo_br_readFast(&b, llBits);
            stLL = ll.nextState + (u32)zo_br_read(&b, ll.nbBits);
            stML = ml.nextState + (u32)zo_br_read(&b, ml.nbBits);
            stOF = of.nextState + (u32)zo_br_read(&b, of.nbBits);
            {   size_t const oneSeqSize = zo_execSequence(op, oend, litLength, matchLength, offset, &litPtr, litEnd,
                                                          d->prefixStart, d->virtualStart, d->dictEnd);
                if (zo_isErr(oneSeqSize)) return oneSeqSize;
                op += oneSeqSize;
            }
            if (!--nbSeq) break;
            zo_br_reload(&b);
        }
        if (zo_br_reload(&b) < ZO_BIT_completed) return ZO_ERR(corruption_detected);
        d->rep[0] = (u32)prevOffset[0]; d->rep[1] = (u32)prevOffset[1]; d->rep[2] = (u32)prevOffset[2];
    }
    {   size_t const lastLLSize = (size_t)(litEnd - litPtr);
        if (lastLLSize > (size_t)(oend - op)) return ZO_ERR(dstSize_tooSmall);
        if (op != NULL) { memcpy(op, litPtr, lastLLSize); op += lastLLSize; }
    }
    return (size_t)(op - ostart);
}

/* zstd_decompress_block.c:2053-2161 ZSTD_decompressBlock_internal */
static size_t zo_decompressBlock(zo_dctx* d, u8* dst, size_t dstCapacity, const u8* src, size_t srcSize)
{
    const u8* ip = src;
    int nbSeq;
    if (srcSize > ZO_BLOCKSIZE_MAX) return ZO_ERR(srcSize_wrong);
    {   size_t const litCSize = zo_decodeLiterals(d, src, srcSize, dst, dstCapacity);
        if (zo_isErr(litCSize)) return litCSize;
        ip += litCSize; srcSize -= litCSize;
    }
    {   size_t const seqHSize = zo_decodeSeqHeaders(d, &nbSeq, ip, srcSize);
        if (zo_isErr(seqHSize)) return seqHSize;
        ip += seqHSize; srcSize -= seqHSize;
    }
    if ((dst == NULL || dstCapacity == 0) && nbSeq > 0) return ZO_ERR(dstSize_tooSmall);
    return zo_decompressSequences(d, dst, dstCapacity, ip, srcSize, nbSeq);
}

/* ------------------------------------------------------------------ XXH64 (xxhash.h; public algorithm) for checksummed frames */
static u64 zo_rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
#define ZO_P1 0x9E3779B185EBCA87ULL
#define ZO_P2 0xC2B2AE3D27D4EB4FULL
#define ZO_P3 0x165667B19E3779F9ULL
#define ZO_P4 0x85EBCA77C2B2AE63ULL
#define ZO_P5 0x27D4EB2F165667C5ULL
static u64 zo_xxh_round(u64 acc, u64 in) { acc += in * ZO_P2; acc = zo_rotl64(acc, 31); return acc * ZO_P1; }
static u64 zo_xxh_merge(u64 acc, u64 v) { v = zo_xxh_round(0, v); acc ^= v; return acc * ZO_P1 + ZO_P4; }
static u64 zo_xxh64(const u8* p, size_t len, u64 seed)
{
    const u8* const end = p + len;
    u64 h;
    if (len >= 32) {
        const u8* const limit = end - 32;
        u64 v1 = seed + ZO_P1 + ZO_P2, v2 = seed + ZO_P2, v3 = seed, v4 = seed - ZO_P1;
        do { v1 = zo_xxh_round(v1, zo_rd64(p)); v2 = zo_xxh_round(v2, zo_rd64(p + 8));
             v3 = zo_xxh_round(v3, zo_rd64(p + 16)); v4 = zo_xxh_round(v4, zo_rd64(p + 24)); p += 32; } while (p <= limit);
        h = zo_rotl64(v1, 1) + zo_rotl64(v2, 7) + zo_rotl64(v3, 12) + zo_rotl64(v4, 18);
        h = zo_xxh_merge(h, v1); h = zo_xxh_merge(h, v2); h = zo_xxh_merge(h, v3); h = zo_xxh_merge(h, v4);
    } else h = seed + ZO_P5;
    h += (u64)len;
    while (p + 8 <= end) { h ^= zo_xxh_round(0, zo_rd64(p)); h = zo_rotl64(h, 27) * ZO_P1 + ZO_P4; p += 8; }
    if (p + 4 <= end) { h ^= (u64)zo_rd32(p) * ZO_P1; h = zo_rotl64(h, 23) * ZO_P2 + ZO_P3; p += 4; }
    while (p < end) { h ^= (*p) * ZO_P5; h = zo_rotl64(h, 11) * ZO_P1; p++; }
    h ^= h >> 33; h *= ZO_P2; h ^= h >> 29; h *= ZO_P3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------ frames */
typedef struct { u64 frameContentSize; u64 windowSize; u32 blockSizeMax; u32 isSkippable; u32 headerSize; u32 dictID; u32 checksumFlag; } zo_frameHeader;

/* zstd_decompress.c:414-427 */
static size_t zo_frameHeaderSize(const u8* src, size_t srcSize)
{
    static const size_t did[4] = { 0, 1, 2, 4 }, fcs[4] = { 0, 2, 4, 8 };
    if (srcSize < 5) return ZO_ERR(srcSize_wrong);
    {   u8 const fhd = src[4];
        u32 const dictID = fhd & 3, singleSegment = (fhd >> 5) & 1, fcsId = fhd >> 6;
        return 5 + !singleSegment + did[dictID] + fcs[fcsId] + (singleSegment && !fcsId);
    }
}
/* zstd_decompress.c:445-547 ZSTD_getFrameHeader_advanced (format zstd1): 0 ok, >0 wanted size, or error */
static size_t zo_getFrameHeader(zo_frameHeader* z, const u8* src, size_t srcSize)
{
    size_t const minInputSize = 5;
    if (srcSize > 0 && src == NULL) return ZO_ERR(GENERIC);
    if (srcSize < minInputSize) {
        if (srcSize > 0) {
            size_t const toCopy = srcSize < 4 ? srcSize : 4;
            u8 hbuf[4]; zo_wr32(hbuf, ZO_MAGIC);
            memcpy(hbuf, src, toCopy);
            if (zo_rd32(hbuf) != ZO_MAGIC) {
                zo_wr32(hbuf, ZO_MAGIC_SKIP_START);
                memcpy(hbuf, src, toCopy);
                if ((zo_rd32(hbuf) & ZO_MAGIC_SKIP_MASK) != ZO_MAGIC_SKIP_START) return ZO_ERR(prefix_unknown);
            }
        }
        return minInputSize;
    }
    memset(z, 0, sizeof(*z));
    if (zo_rd32(src) != ZO_MAGIC) {
        if ((zo_rd32(src) & ZO_MAGIC_SKIP_MASK) == ZO_MAGIC_SKIP_START) {
            if (srcSize < 8) return 8;
            z->frameContentSize = zo_rd32(src + 4);
            z->isSkippable = 1;
            return 0;
        }
        return ZO_ERR(prefix_unknown);
    }
    {   size_t const fhsize = zo_frameHeaderSize(src, srcSize);
        if (srcSize < fhsize) return fhsize;
        z->headerSize = (u32)fhsize;
    }
    {   u8 const fhdByte = src[4];
        size_t pos = 5;
        u32 const dictIDSizeCode = fhdByte & 3, checksumFlag = (fhdByte >> 2) & 1, singleSegment = (fhdByte >> 5) & 1, fcsID = fhdByte >> 6;
        u64 windowSize = 0, frameContentSize = ZO_CONTENTSIZE_UNKNOWN;
        u32 dictID = 0;
        if ((fhdByte & 0x08) != 0) return ZO_ERR(frameParameter_unsupported);
        if (!singleSegment) {
            u8 const wlByte = src[pos++];
            u32 const windowLog = (wlByte >> 3) + ZO_WLOG_ABSMIN;
            if (windowLog > ZO_WLOG_MAX) return ZO_ERR(frameParameter_windowTooLarge);
            windowSize = ((u64)1 << windowLog);
            windowSize += (windowSize >> 3) * (wlByte & 7);
        }
        switch (dictIDSizeCode) {
        case 1: dictID = src[pos]; pos++; break;
        case 2: dictID = zo_rd16(src + pos); pos += 2; break;
        case 3: dictID = zo_rd32(src + pos); pos += 4; break;
        default: break;
        }
        switch (fcsID) {
        case 0: if (singleSegment) frameContentSize = src[pos]; break;
        case 1: frameContentSize = zo_rd16(src + pos) + 256; break;
        case 2: frameContentSize = zo_rd32(src + pos); break;
        default: frameContentSize = zo_rd64(src + pos); break;
        }
        if (singleSegment) windowSize = frameContentSize;
        z->frameContentSize = frameContentSize;
        z->windowSize = windowSize;
        z->blockSizeMax = (u32)(windowSize < ZO_BLOCKSIZE_MAX ? windowSize : ZO_BLOCKSIZE_MAX);
        z->dictID = dictID;
        z->checksumFlag = checksumFlag;
    }
    return 0;
}
/* zstd_decompress.c:565-581 (legacy frames not handled: they report ERROR like any unknown magic) */
unsigned long long zo_getFrameContentSize(const void* src, size_t srcSize)
{
    zo_frameHeader z;
    if (zo_getFrameHeader(&z, (const u8*)src, srcSize) != 0) return ZO_CONTENTSIZE_ERROR;
    if (z.isSkippable) return 0;
    return z.frameContentSize;
}

/* zstd_decompress.c:1534-1559 ZSTD_decompressBegin (+ zstd_ddict.c:59-87 ZSTD_copyDDictParameters) */
static void zo_decompressBegin(zo_dctx* d, const zo_ddict_t* dd)
{
    d->previousDstEnd = d->prefixStart = d->virtualStart = d->dictEnd = NULL;
    d->litEntropy = d->fseEntropy = 0;
    d->dictID = 0;
    d->rep[0] = 1; d->rep[1] = 4; d->rep[2] = 8;
    d->LLTptr = &d->LLTable; d->MLTptr = &d->MLTable; d->OFTptr = &d->OFTable; d->hufPtr = &d->hufTable;
    if (dd) {
        d->dictID = dd->dictID;
        d->prefixStart = dd->content;
        d->virtualStart = dd->content;
        d->dictEnd = dd->content + dd->contentSize;
        d->previousDstEnd = d->dictEnd;
        if (dd->entropyPresent) {
            d->litEntropy = d->fseEntropy = 1;
            d->LLTptr = &dd->ll; d->MLTptr = &dd->ml; d->OFTptr = &dd->of; d->hufPtr = &dd->huf;
            d->rep[0] = dd->rep[0]; d->rep[1] = dd->rep[1]; d->rep[2] = dd->rep[2];
        }
    }
}
/* zstd_decompress_block.c:2164-2172 ZSTD_checkContinuity */
static void zo_checkContinuity(zo_dctx* d, const u8* dst, size_t dstSize)
{
    if (dst != d->previousDstEnd && dstSize > 0) {
        d->dictEnd = d->previousDstEnd;
        d->virtualStart = dst - (d->previousDstEnd - d->prefixStart);
        d->prefixStart = dst;
        d->previousDstEnd = dst;
    }
}

/* zstd_decompress.c:947-1053 ZSTD_decompressFrame */
static size_t zo_decompressFrame(zo_dctx* d, u8* dst, size_t dstCapacity, const u8** srcPtr, size_t* srcSizePtr)
{
    const u8* ip = *srcPtr;
    u8* const ostart = dst;
    u8* const oend = dstCapacity != 0 ? ostart + dstCapacity : ostart;
    u8* op = ostart;
    size_t remaining = *srcSizePtr;
    zo_frameHeader fh;
    if (remaining < 6 + 3) return ZO_ERR(srcSize_wrong);   /* ZSTD_FRAMEHEADERSIZE_MIN(zstd1) + blockHeader */
    {   size_t const fhs = zo_frameHeaderSize(ip, 5);
        if (zo_isErr(fhs)) return fhs;
        if (remaining < fhs + 3) return ZO_ERR(srcSize_wrong);
        {   size_t const r = zo_getFrameHeader(&fh, ip, fhs);   /* :698 ZSTD_decodeFrameHeader */
            if (zo_isErr(r)) return r;
            if (r > 0) return ZO_ERR(srcSize_wrong);
            if (fh.dictID && (d->dictID != fh.dictID)) return ZO_ERR(dictionary_wrong);
        }
        ip += fhs; remaining -= fhs;
    }
    d->blockSizeMax = fh.blockSizeMax;
    for (;;) {
        size_t decodedSize, cBlockSize;
        u32 hdr, lastBlock, blockType, origSize;
        if (remaining < 3) return ZO_ERR(srcSize_wrong);
        hdr = zo_rd24(ip);
        lastBlock = hdr & 1; blockType = (hdr >> 1) & 3; origSize = hdr >> 3;
        if (blockType == zo_bt_reserved) return ZO_ERR(corruption_detected);
        cBlockSize = (blockType == zo_bt_rle) ? 1 : origSize;
        ip += 3; remaining -= 3;
        if (cBlockSize > remaining) return ZO_ERR(srcSize_wrong);
        switch (blockType) {
        case zo_bt_compressed:
            decodedSize = zo_decompressBlock(d, op, (size_t)(oend - op), ip, cBlockSize);
            break;
        case zo_bt_raw:
            if (cBlockSize > (size_t)(oend - op)) { decodedSize = ZO_ERR(dstSize_tooSmall); break; }
            if (op == NULL) { decodedSize = cBlockSize == 0 ? 0 : ZO_ERR(dstBuffer_null); break; }
            memmove(op, ip, cBlockSize); decodedSize = cBlockSize;
            break;
        default:
            if (origSize > (size_t)(oend - op)) { decodedSize = ZO_ERR(dstSize_tooSmall); break; }
            if (op == NULL) { decodedSize = origSize == 0 ? 0 : ZO_ERR(dstBuffer_null); break; }
            memset(op, *ip, origSize); decodedSize = origSize;
            break;
        }
        if (zo_isErr(decodedSize)) return decodedSize;
        if (decodedSize != 0) op += decodedSize;
        d->previousDstEnd = op;
        ip += cBlockSize; remaining -= cBlockSize;
        if (lastBlock) break;
    }
    if (fh.frameContentSize != ZO_CONTENTSIZE_UNKNOWN) {
        if ((u64)(op - ostart) != fh.frameContentSize) return ZO_ERR(corruption_detected);
    }
    if (fh.checksumFlag) {
        if (remaining < 4) return ZO_ERR(checksum_wrong);
        if (zo_rd32(ip) != (u32)zo_xxh64(ostart, (size_t)(op - ostart), 0)) return ZO_ERR(checksum_wrong);
        ip += 4; remaining -= 4;
    }
    *srcPtr = ip; *srcSizePtr = remaining;
    return (size_t)(op - ostart);
}

/* zstd_decompress.c:1055-1145 ZSTD_decompressMultiFrame */
size_t zo_decompress_usingDDict(void* dstv, size_t dstCapacity, const void* srcv, size_t srcSize, const zo_ddict_t* dd)
{
    u8* dst = (u8*)dstv;
    u8* const dststart = dst;
    const u8* src = (const u8*)srcv;
    int moreThan1Frame = 0;
    size_t result;
    zo_dctx* d = (zo_dctx*)malloc(sizeof(zo_dctx));
    if (!d) return ZO_ERR(memory_allocation);
    d->litBuf = (u8*)malloc(ZO_BLOCKSIZE_MAX + 64);
    if (!d->litBuf) { free(d); return ZO_ERR(memory_allocation); }
    zo_initBases();
    zo_buildSeqTable(&d->defLL, zo_LL_defaultNorm, ZO_MaxLL, zo_LL_base, zo_LL_bits, ZO_LL_DEFAULTNORMLOG);
    zo_buildSeqTable(&d->defOF, zo_OF_defaultNorm, ZO_DefaultMaxOff, zo_OF_base, zo_OF_bits, ZO_OF_DEFAULTNORMLOG);
    zo_buildSeqTable(&d->defML, zo_ML_defaultNorm, ZO_MaxML, zo_ML_base, zo_ML_bits, ZO_ML_DEFAULTNORMLOG);

    while (srcSize >= 5) {
        if (srcSize >= 4) {
            u32 const magic = zo_rd32(src);
            if ((magic & ZO_MAGIC_SKIP_MASK) == ZO_MAGIC_SKIP_START) {   /* :583-600 readSkippableFrameSize */
                u32 sizeU32; size_t skippableSize;
                if (srcSize < 8) { result = ZO_ERR(srcSize_wrong); goto done; }
                sizeU32 = zo_rd32(src + 4);
                if ((u32)(sizeU32 + 8) < sizeU32) { result = ZO_ERR(frameParameter_unsupported); goto done; }
                skippableSize = 8 + (size_t)sizeU32;
                if (skippableSize > srcSize) { result = ZO_ERR(srcSize_wrong); goto done; }
                src += skippableSize; srcSize -= skippableSize;
                continue;
            }
        }
        zo_decompressBegin(d, dd);
        zo_checkContinuity(d, dst, dstCapacity);
        {   size_t const res = zo_decompressFrame(d, dst, dstCapacity, &src, &srcSize);
            if (zo_isErr(res) && ((size_t)0 - res) == ZE_prefix_unknown && moreThan1Frame == 1) { result = ZO_ERR(srcSize_wrong); goto done; }
            if (zo_isErr(res)) { result = res; goto done; }
            if (res != 0) dst += res;
            dstCapacity -= res;
        }
        moreThan1Frame = 1;
    }
    if (srcSize) { result = ZO_ERR(srcSize_wrong); goto done; }
    result = (size_t)(dst - dststart);
done:
    free(d->litBuf); free(d);
    return result;
}
size_t zo_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    return zo_decompress_usingDDict(dst, dstCapacity, src, srcSize, NULL);
}

/* zstd_ddict.c:90-175 ZSTD_createDDict / ZSTD_loadEntropy_intoDDict; zstd_decompress.c:1425-1511 ZSTD_loadDEntropy */
zo_ddict_t* zo_createDDict(const void* dict, size_t dictSize)
{
    zo_ddict_t* dd = (zo_ddict_t*)calloc(1, sizeof(zo_ddict_t));
    if (!dd) return NULL;
    zo_initBases();
    dd->raw = (u8*)malloc(dictSize ? dictSize : 1);
    if (!dd->raw) { free(dd); return NULL; }
    if (dictSize) memcpy(dd->raw, dict, dictSize);
    dd->rawSize = dictSize;
    dd->content = dd->raw; dd->contentSize = dictSize;
    dd->dictID = 0; dd->entropyPresent = 0;
    if (dictSize < 8) return dd;
    if (zo_rd32(dd->raw) != ZO_MAGIC_DICT) return dd;   /* pure content mode */
    dd->dictID = zo_rd32(dd->raw + 4);
    {   const u8* p = dd->raw + 8;
        const u8* const end = dd->raw + dictSize;
        s16 norm[ZO_MaxSeq + 1]; u32 maxV, log;
        size_t h;
        if (dictSize <= 8) goto bad;
        h = zo_huf_readDTable(&dd->huf, p, (size_t)(end - p)); if (zo_isErr(h)) goto bad; p += h;
        maxV = ZO_MaxOff; h = zo_fse_readNCount(norm, &maxV, &log, p, (size_t)(end - p));
        if (zo_isErr(h) || maxV > ZO_MaxOff || log > ZO_OffFSELog) goto bad;
        zo_buildSeqTable(&dd->of, norm, maxV, zo_OF_base, zo_OF_bits, log); p += h;
        maxV = ZO_MaxML; h = zo_fse_readNCount(norm, &maxV, &log, p, (size_t)(end - p));
        if (zo_isErr(h) || maxV > ZO_MaxML || log > ZO_MLFSELog) goto bad;
        zo_buildSeqTable(&dd->ml, norm, maxV, zo_ML_base, zo_ML_bits, log); p += h;
        maxV = ZO_MaxLL; h = zo_fse_readNCount(norm, &maxV, &log, p, (size_t)(end - p));
        if (zo_isErr(h) || maxV > ZO_MaxLL || log > ZO_LLFSELog) goto bad;
        zo_buildSeqTable(&dd->ll, norm, maxV, zo_LL_base, zo_LL_bits, log); p += h;
        if (p + 12 > end) goto bad;
        {   size_t const dictContentSize = (size_t)(end - (p + 12)); int i;
            for (i = 0; i < 3; i++) { u32 const rep = zo_rd32(p); p += 4; if (rep == 0 || rep > dictContentSize) goto bad; dd->rep[i] = rep; }
        }
        dd->content = p; dd->contentSize = (size_t)(end - p);
        dd->entropyPresent = 1;
    }
    return dd;
bad:
    free(dd->raw); free(dd);
    return NULL;
}
void zo_freeDDict(zo_ddict_t* dd) { if (dd) { free(dd->raw); free(dd); } }

/* ------------------------------------------------------------------ error names (error_private.c:16-66) */
unsigned zo_isError(size_t code) { return zo_isErr(code); }
const char* zo_getErrorName(size_t code)
{
    if (!zo_isErr(code)) return "No error detected";
    switch ((int)((size_t)0 - code)) {
    case ZE_GENERIC: return "Error (generic)";
    case ZE_prefix_unknown: return "Unknown frame descriptor";
    case ZE_version_unsupported: return "Version not supported";
    case ZE_frameParameter_unsupported: return "Unsupported frame parameter";
    case ZE_frameParameter_windowTooLarge: return "Frame requires too much memory for decoding";
    case ZE_corruption_detected: return "Data corruption detected";
    case ZE_checksum_wrong: return "Restored data doesn't match checksum";
    case ZE_literals_headerWrong: return "Header of Literals' block doesn't respect format specification";
    case ZE_parameter_unsupported: return "Unsupported parameter";
    case ZE_parameter_combination_unsupported: return "Unsupported combination of parameters";
    case ZE_parameter_outOfBound: return "Parameter is out of bound";
    case ZE_init_missing: return "Context should be init first";
    case ZE_memory_allocation: return "Allocation error : not enough memory";
    case ZE_workSpace_tooSmall: return "workSpace buffer is not large enough";
    case ZE_stage_wrong: return "Operation not authorized at current processing stage";
    case ZE_tableLog_tooLarge: return "tableLog requires too much memory : unsupported";
    case ZE_maxSymbolValue_tooLarge: return "Unsupported max Symbol Value : too large";
    case ZE_maxSymbolValue_tooSmall: return "Specified maxSymbolValue is too small";
    case ZE_stabilityCondition_notRespected: return "pledged buffer stability condition is not respected";
    case ZE_dictionary_corrupted: return "Dictionary is corrupted";
    case ZE_dictionary_wrong: return "Dictionary mismatch";
    case ZE_dictionaryCreation_failed: return "Cannot create Dictionary from provided samples";
    case ZE_dstSize_tooSmall: return "Destination buffer is too small";
    case ZE_srcSize_wrong: return "Src size is incorrect";
    case ZE_dstBuffer_null: return "Operation on NULL destination buffer";
    case ZE_noForwardProgress_destFull: return "Operation made no progress over multiple calls, due to output buffer being full";
    case ZE_noForwardProgress_inputEmpty: return "Operation made no progress over multiple calls, due to input being empty";
    case ZE_frameIndex_tooLarge: return "Frame index is too large";
    case ZE_seekableIO: return "An I/O error occurred when reading/seeking";
    case ZE_dstBuffer_wrong: return "Destination buffer is wrong";
    case ZE_srcBuffer_wrong: return "Source buffer is wrong";
    case ZE_sequenceProducer_failed: return "Block-level external sequence producer returned an error code";
    case ZE_externalSequences_invalid: return "External sequences are not valid";
    default: return "Unspecified error code";
    }
}
