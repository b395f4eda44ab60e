// zsa_decode.h -- device-side Zstandard frame decoder: one 64-lane wave per frame.
//
// Replaces, for independent frames decoded in one launch, what the reference reaches through
// ZSTD_decompress / ZSTD_decompress_usingDDict (zstd.go:168, zstd_bulk.go:125):
//   frame walk        zstd_decompress.c:947-1145      block header  zstd_decompress_block.c:57-71
//   literals          zstd_decompress_block.c:120-322 Huffman       huf_decompress.c:358-666, entropy_common.c:244-307
//   sequence tables   zstd_decompress_block.c:455-742 FSE headers   entropy_common.c:43-188, fse_decompress.c:60-286
//   sequences         zstd_decompress_block.c:1185-1299, 1569-1654  execution :873-1060
//
// Mapping onto the wave:
//   * decode tables live in LDS (Huffman 2^tableLog x u16, three sequence tables x u64 cells) and are built by the
//     whole wave (zsa_huf_buildTable; zsa_buildSeqTable over zsa_common.h's zsa_fse_spread_wave);
//   * the four Huffman streams of a literal section run on four lanes;
//   * the sequence bitstream is one serial chain (lane 0), decoded 64 sequences at a time into LDS;
//   * the 64 lanes then execute that batch: a wave prefix-sum gives every sequence its output position,
//     literals are copied by all lanes at once, independent matches likewise, dependent matches in rounds (a match is
//     ready when no unfinished match lies in the range its source comes from), with a fence between rounds.
// Serial sections run under `lane == 0` and publish through LDS; every wv_* call is reached by all lanes.
#pragma once
#include "zsa_common.h"

#ifndef ZSA_DEP_FENCE
#define ZSA_DEP_FENCE() wv_wave_fence()
#endif

// ---- packed sequence-table cell: baseValue | nbBits<<32 | nbAdditionalBits<<40 | nextState<<48.  The high word is what the
// one-lane state walk works on: nbBits (<= 9) lies in its low five bits, which is where v_bfe_u32 reads a field's width and
// offset from -- the word itself serves as that operand, no extraction.
typedef u64 ZsaSeqCell;
ZSA_DEV ZsaSeqCell zsa_seqcell(u32 base, u32 nextState, u32 addBits, u32 nbBits)
{ return (u64)base | ((u64)nbBits << 32) | ((u64)addBits << 40) | ((u64)nextState << 48); }
ZSA_DEV u32 zsa_cell_base(ZsaSeqCell c) { return (u32)c; }
ZSA_DEV u32 zsa_cell_nbBits(ZsaSeqCell c) { return (u32)(c >> 32) & 0x1Fu; }
ZSA_DEV u32 zsa_cell_addBits(ZsaSeqCell c) { return (u32)(c >> 40) & 0xFFu; }
ZSA_DEV u32 zsa_cell_next(ZsaSeqCell c) { return (u32)(c >> 48); }

struct ZsaDDictDev {            // digested dictionary, device resident (zstd_ddict.c:37-45)
    const u8* content; u32 contentSize; u32 dictID; u32 entropyPresent;
    u32 hufLog, llLog, ofLog, mlLog;
    u32 rep[3];
    const u16* huf;             // 2^hufLog cells
    const ZsaSeqCell *ll, *of, *ml;
};

#define ZSA_BITSTAGE 768u    // >= 64 sequences x 89 bits + 8
// LDS of one decoding workgroup, sized for 12 workgroups per CU: the literal phase of a block (Huffman table) and its
// sequence phase (three FSE tables) share the same bytes, and so do table construction and batch execution.  Tables a
// later block may name again ("repeat" modes) are parked in the workgroup's global workspace (ZsaDecBackup).
struct __attribute__((aligned(16))) ZsaDecShared {
    union {
        u16 huf[1 << ZSA_HUF_TABLELOG_MAX];                     // literal phase
        struct { ZsaSeqCell ll[512], ml[512], of[256]; };        // sequence phase
    };
    union {
        struct {   // table-building scratch
            u8  weights[256];
            u16 symStart[256];
            u16 symbolNext[256];
            s16 norm[256];
            u32 fseCell[64];            // weights FSE table: newState | symbol<<16 | nbBits<<24
            u32 rankVal[ZSA_HUF_TABLELOG_MAX + 2];
        };
        struct {   // one batch of decoded sequences
            u32 seqLL[64], seqML[64], seqOF[64], outStart[64], litStart[64], mStart[64];
            // the part of the sequence bitstream the next batch reads (it is consumed downwards), staged by all lanes
            u8  bits[ZSA_BITSTAGE + 16];
        };
    };
    u32 hufLog, llLog, ofLog, mlLog;
    // uniform state published by lane 0
    u64 result;                 // final return value of the frame walk
    u32 err;                    // first error raised inside the current block (0 = none)
    u32 u[16];
    u64 prof[ZSA_NPROF], profLast;
};
// While a block's Huffman streams are decoded nothing else of a block lives in LDS: the single-symbol table lies in the first
// 4 KiB and a table of symbol PAIRS (zsa_huf_buildPairs), 4 bytes per cell, takes the 8 KiB behind it -- the rest of the first
// union and the head of the second.
// (tables of up to 2048 cells: what libzstd's encoder produces; a 4096-cell table, which the format allows, is decoded without pairs)
static_assert(offsetof(ZsaDecShared, weights) == 10240 && offsetof(ZsaDecShared, hufLog) >= 4096 + 4 * 2048,
              "the pair table overlays the sequence tables and the batch arrays");
ZSA_DEV u32* zsa_huf_pairTable(ZsaDecShared* S) { return (u32*)((u8*)S + 4096); }
// The table descriptions of a block (Huffman weights, the three FSE distributions) are read by one lane, a few bits per symbol,
// every step waiting for the bytes of the step before: read from the frame in global memory that is one memory round trip per
// symbol or two.  So all lanes first copy the description's bytes into LDS that no stage of table construction uses -- the
// tail of the batch bit stage, behind the table-building scratch -- and the reader runs there.  A valid description is at
// most 128 bytes (Huffman weights: the size byte says so) or 85 (53 symbols of at most 10 + 2 bits); a reader that leaves the
// staged bytes fails on them exactly as it would fail on the frame's (zsa_readNCount pads with zeros and reports the overrun).
#define ZSA_HDRSTAGE 192u
#define ZSA_HDRSTAGE_AT 576u
static_assert(offsetof(ZsaDecShared, bits) + ZSA_HDRSTAGE_AT >= offsetof(ZsaDecShared, rankVal) + sizeof(((ZsaDecShared*)0)->rankVal)
              && ZSA_HDRSTAGE_AT + ZSA_HDRSTAGE <= sizeof(((ZsaDecShared*)0)->bits), "the header stage lies behind the table-building scratch");
ZSA_DEV u8* zsa_hdr_stage(ZsaDecShared* S) { return S->bits + ZSA_HDRSTAGE_AT; }
ZSA_DEV u32 zsa_stage_header(ZsaDecShared* S, const u8* p, u32 n, int lane)
{   // returns the number of bytes staged (uniform)
    u32 const m = n < ZSA_HDRSTAGE ? n : ZSA_HDRSTAGE;
    u8* const st = zsa_hdr_stage(S);
    u32 const i0 = (u32)lane, i1 = i0 + ZSA_WAVE, i2 = i1 + ZSA_WAVE;
    u32 const b0 = i0 < m ? wv_ldg8(p + i0) : 0u, b1 = i1 < m ? wv_ldg8(p + i1) : 0u, b2 = i2 < m ? wv_ldg8(p + i2) : 0u;
    st[i0] = (u8)b0; st[i1] = (u8)b1; st[i2] = (u8)b2;
    wv_lds_fence();
    return m;
}
// per-workgroup global workspace: regenerated literals of the block at hand, then the parked tables
struct __attribute__((aligned(16))) ZsaDecBackup { u16 huf[1 << ZSA_HUF_TABLELOG_MAX]; ZsaSeqCell ll[512], ml[512], of[256]; };
#define ZSA_DEC_LITBUF_BYTES (ZSA_BLOCKSIZE_MAX + 128u)
#define ZSA_DEC_WKSP_BYTES   (ZSA_DEC_LITBUF_BYTES + sizeof(ZsaDecBackup))
enum { ZU_blockType, ZU_lastBlock, ZU_cSize, ZU_origSize, ZU_litMode, ZU_litSize, ZU_litOff, ZU_litCSize,
       ZU_nbSeq, ZU_seqOff, ZU_seqSize, ZU_batchN, ZU_streams, ZU_hufOff, ZU_hufCSize, ZU_tmp };

// ------------------------------------------------------------------ backward bit reader (bitstream.h:253-432)
struct ZsaBitR { u64 container; u32 bitsConsumed; const u8* ptr; const u8* start; };
enum { ZB_unfinished = 0, ZB_endOfBuffer = 1, ZB_completed = 2, ZB_overflow = 3 };

ZSA_DEV u32 zsa_br_init(ZsaBitR& b, const u8* src, u32 srcSize)
{   // returns 0 or an error code
    if (srcSize < 1) { b.container = 0; b.bitsConsumed = 0; b.ptr = b.start = src; return ZSA_srcSize_wrong; }
    b.start = src;
    u32 const lastByte = src[srcSize - 1];
    if (srcSize >= 8) {
        b.ptr = src + srcSize - 8;
        b.container = zsa_ld64(b.ptr);
        b.bitsConsumed = lastByte ? 8 - zsa_highbit32(lastByte) : 0;
        if (lastByte == 0) return ZSA_GENERIC;
    } else {
        b.ptr = src;
        u64 c = 0;
        for (u32 i = 0; i < srcSize; i++) c |= (u64)src[i] << (8 * i);
        b.container = c;
        b.bitsConsumed = lastByte ? 8 - zsa_highbit32(lastByte) : 0;
        if (lastByte == 0) return ZSA_corruption_detected;
        b.bitsConsumed += (8 - srcSize) * 8;
    }
    return 0;
}
ZSA_DEV u32 zsa_br_look(const ZsaBitR& b, u32 nbBits)
{   // BIT_lookBits (x86-64 form of BIT_getMiddleBits): nbBits <= 31 here
    u32 const start = 64u - b.bitsConsumed - nbBits;
    return (u32)(b.container >> (start & 63)) & ((1u << nbBits) - 1u);
}
ZSA_DEV u32 zsa_br_lookFast(const ZsaBitR& b, u32 nbBits)
{   // BIT_lookBitsFast: nbBits >= 1
    return (u32)((b.container << (b.bitsConsumed & 63)) >> ((64u - nbBits) & 63));
}
ZSA_DEV u32 zsa_br_read(ZsaBitR& b, u32 nbBits) { u32 v = zsa_br_look(b, nbBits); b.bitsConsumed += nbBits; return v; }
ZSA_DEV u32 zsa_br_readFast(ZsaBitR& b, u32 nbBits) { u32 v = zsa_br_lookFast(b, nbBits); b.bitsConsumed += nbBits; return v; }
ZSA_DEV int zsa_br_reload(ZsaBitR& b)
{
    if (b.bitsConsumed > 64) return ZB_overflow;
    if (b.ptr >= b.start + 8) {
        b.ptr -= b.bitsConsumed >> 3;
        b.bitsConsumed &= 7;
        b.container = zsa_ld64(b.ptr);
        return ZB_unfinished;
    }
    if (b.ptr == b.start) return (b.bitsConsumed < 64) ? ZB_endOfBuffer : ZB_completed;
    u32 nbBytes = b.bitsConsumed >> 3;
    int result = ZB_unfinished;
    if (b.ptr - nbBytes < b.start) { nbBytes = (u32)(b.ptr - b.start); result = ZB_endOfBuffer; }
    b.ptr -= nbBytes;
    b.bitsConsumed -= nbBytes * 8;
    b.container = zsa_ld64(b.ptr);
    return result;
}
// BIT_reloadDStream with the refill served from the LDS copy of [stageLo, stageHi) when it covers the 8 bytes
ZSA_DEV u64 zsa_br_fetch(const u8* p, const u8* stage, const u8* stageLo, const u8* stageHi)
{
    if (p >= stageLo && p + 8 <= stageHi) { struct __attribute__((packed)) Pk { u64 v; }; return ((const Pk*)(stage + (p - stageLo)))->v; }
    return zsa_ld64(p);
}
ZSA_DEV int zsa_br_reload_staged(ZsaBitR& b, const u8* stage, const u8* stageLo, const u8* stageHi)
{
    if (b.bitsConsumed > 64) return ZB_overflow;
    if (b.ptr >= b.start + 8) {
        b.ptr -= b.bitsConsumed >> 3;
        b.bitsConsumed &= 7;
        b.container = zsa_br_fetch(b.ptr, stage, stageLo, stageHi);
        return ZB_unfinished;
    }
    if (b.ptr == b.start) return (b.bitsConsumed < 64) ? ZB_endOfBuffer : ZB_completed;
    u32 nbBytes = b.bitsConsumed >> 3;
    int result = ZB_unfinished;
    if (b.ptr - nbBytes < b.start) { nbBytes = (u32)(b.ptr - b.start); result = ZB_endOfBuffer; }
    b.ptr -= nbBytes;
    b.bitsConsumed -= nbBytes * 8;
    b.container = zsa_br_fetch(b.ptr, stage, stageLo, stageHi);
    return result;
}
ZSA_DEV bool zsa_br_end(const ZsaBitR& b) { return (b.ptr == b.start) && (b.bitsConsumed == 64); }

// ------------------------------------------------------------------ FSE table description (entropy_common.c:43-188)
// Serial (call from one lane).  Reads LSB-first with zero padding past the buffer, as the reference does
// through its 8-byte zero-padded copy / iend-4 clamping.  Returns header size, or 0 with *err set.
ZSA_DEV u32 zsa_peekbits(const u8* src, u32 n, u32 bitPos)
{   // 25+ bits starting at bitPos
    u32 const by = bitPos >> 3, sh = bitPos & 7;
    u64 v = 0;
    if (by + 8 <= n) v = zsa_ld64(src + by);
    else for (u32 i = 0; i < 5; i++) if (by + i < n) v |= (u64)src[by + i] << (8 * i);
    return (u32)(v >> sh);
}
ZSA_DEV u32 zsa_readNCount(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const u8* src, u32 hbSize, u32* err)
{
    u32 const maxSV1 = *maxSVPtr + 1;
    u32 charnum = 0, bitPos = 0;
    u32 const bitEnd = hbSize * 8;
    int nbBits, remaining, threshold, previous0 = 0;
    if (hbSize == 0) { *err = ZSA_srcSize_wrong; return 0; }
    for (u32 i = 0; i < maxSV1; i++) norm[i] = 0;
    nbBits = (int)(zsa_peekbits(src, hbSize, 0) & 0xF) + ZSA_FSE_MIN_TABLELOG;
    if (nbBits > 15) { *err = ZSA_tableLog_tooLarge; return 0; }
    bitPos = 4;
    *tableLogPtr = (u32)nbBits;
    remaining = (1 << nbBits) + 1;
    threshold = 1 << nbBits;
    nbBits++;
    for (;;) {
        if (previous0) {
            for (;;) {
                u32 const two = zsa_peekbits(src, hbSize, bitPos) & 3;
                bitPos += 2;
                charnum += two;
                if (two != 3) break;
                if (bitPos > bitEnd + 32) { *err = ZSA_corruption_detected; return 0; }
            }
            if (charnum >= maxSV1) break;
        }
        {   u32 const bitStream = zsa_peekbits(src, hbSize, bitPos);
            int const max = (2 * threshold - 1) - remaining;
            int count;
            if ((bitStream & (u32)(threshold - 1)) < (u32)max) {
                count = (int)(bitStream & (u32)(threshold - 1));
                bitPos += (u32)(nbBits - 1);
            } else {
                count = (int)(bitStream & (u32)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                bitPos += (u32)nbBits;
            }
            count--;
            if (count >= 0) remaining -= count; else remaining += count;
            norm[charnum++] = (s16)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (int)zsa_highbit32((u32)remaining) + 1;
                threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
        }
    }
    if (remaining != 1) { *err = ZSA_corruption_detected; return 0; }
    if (charnum > maxSV1) { *err = ZSA_maxSymbolValue_tooSmall; return 0; }
    if (bitPos > bitEnd) { *err = ZSA_corruption_detected; return 0; }
    *maxSVPtr = charnum - 1;
    return (bitPos + 7) >> 3;
}

// The weights' FSE stream (fse_decompress.c:177-237 FSE_decompress_usingDTable_generic, at most 255 weights): two interleaved
// states, one chain -- lane 0.  Table: S->fseCell.  Returns the number of weights, 0 on error.
ZSA_DEV u32 zsa_huf_decodeWeights(ZsaDecShared* S, const u8* cSrc, u32 cSrcSize, u32 tableLog)
{
    u8* const huffWeight = S->weights;
    ZsaBitR bitD;
    if (zsa_br_init(bitD, cSrc, cSrcSize)) return 0;
    u32 state1 = zsa_br_read(bitD, tableLog); zsa_br_reload(bitD);
    u32 state2 = zsa_br_read(bitD, tableLog); zsa_br_reload(bitD);
    u32 op = 0;
    u32 const omax = 255;
#define ZSA_FSE_GET(st) do { u32 const c = S->fseCell[st]; huffWeight[op++] = (u8)(c >> 16); st = (c & 0xFFFF) + zsa_br_read(bitD, c >> 24); } while (0)
    while ((zsa_br_reload(bitD) == ZB_unfinished) & (op + 3 < omax)) {
        ZSA_FSE_GET(state1); ZSA_FSE_GET(state2); ZSA_FSE_GET(state1); ZSA_FSE_GET(state2);
    }
    for (;;) {
        if (op + 2 > omax) return 0;
        ZSA_FSE_GET(state1);
        if (zsa_br_reload(bitD) == ZB_overflow) { ZSA_FSE_GET(state2); break; }
        if (op + 2 > omax) return 0;
        ZSA_FSE_GET(state2);
        if (zsa_br_reload(bitD) == ZB_overflow) { ZSA_FSE_GET(state1); break; }
    }
#undef ZSA_FSE_GET
    return op;
}

// ------------------------------------------------------------------ Huffman table description (entropy_common.c:244-307)
// Serial (one lane): the dictionary digest's reader, and the whole-wave table builder's way out for descriptions that name more
// than 64 weight values.  Fills S->weights, S->rankVal; returns bytes consumed, or 0 with *err set.
ZSA_DEV u32 zsa_huf_readStats(ZsaDecShared* S, u32* nbSymbolsPtr, u32* tableLogPtr, const u8* src, u32 srcSize, u32* err)
{
    u32 weightTotal = 0, iSize, oSize;
    u8* const huffWeight = S->weights;
    if (!srcSize) { *err = ZSA_srcSize_wrong; return 0; }
    iSize = src[0];
    if (iSize >= 128) {
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) { *err = ZSA_srcSize_wrong; return 0; }
        if (oSize >= 256) { *err = ZSA_corruption_detected; return 0; }
        for (u32 n = 0; n < oSize; n += 2) { huffWeight[n] = src[1 + n / 2] >> 4; if (n + 1 < 256) huffWeight[n + 1] = src[1 + n / 2] & 15; }
    } else {
        // fse_decompress.c:177-286 FSE_decompress_wksp, maxLog 6, at most 255 weights
        if (iSize + 1 > srcSize) { *err = ZSA_srcSize_wrong; return 0; }
        const u8* cSrc = src + 1;
        u32 cSrcSize = iSize, tableLog = 0, maxSV = 255;
        u32 const n = zsa_readNCount(S->norm, &maxSV, &tableLog, cSrc, cSrcSize, err);
        if (*err) return 0;
        if (tableLog > 6) { *err = ZSA_tableLog_tooLarge; return 0; }
        cSrc += n; cSrcSize -= n;
        {   // fse_decompress.c:60-163 FSE_buildDTable_internal
            u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1, step = ZSA_FSE_TABLESTEP(tableSize);
            u32 highThreshold = tableSize - 1, position = 0;
            for (u32 s = 0; s <= maxSV; s++) {
                if (S->norm[s] == -1) { S->fseCell[highThreshold--] = s << 16; S->symbolNext[s] = 1; }
                else S->symbolNext[s] = (u16)S->norm[s];
            }
            for (u32 s = 0; s <= maxSV; s++) {
                for (int i = 0; i < S->norm[s]; i++) {
                    S->fseCell[position] = s << 16;
                    position = (position + step) & tableMask;
                    while (position > highThreshold) position = (position + step) & tableMask;
                }
            }
            if (position != 0) { *err = ZSA_GENERIC; return 0; }
            for (u32 u = 0; u < tableSize; u++) {
                u32 const symbol = S->fseCell[u] >> 16;
                u32 const nextState = S->symbolNext[symbol]++;
                u32 const nbBits = tableLog - zsa_highbit32(nextState);
                S->fseCell[u] = (((nextState << nbBits) - tableSize) & 0xFFFF) | (symbol << 16) | (nbBits << 24);
            }
        }
        oSize = zsa_huf_decodeWeights(S, cSrc, cSrcSize, tableLog);   // (every failure of the stream reads as corruption to the callers)
        if (!oSize) { *err = ZSA_corruption_detected; return 0; }
    }
    for (u32 i = 0; i <= ZSA_HUF_TABLELOG_MAX; i++) S->rankVal[i] = 0;
    for (u32 n = 0; n < oSize; n++) {
        if (huffWeight[n] > ZSA_HUF_TABLELOG_MAX) { *err = ZSA_corruption_detected; return 0; }
        S->rankVal[huffWeight[n]]++;
        weightTotal += (1u << huffWeight[n]) >> 1;
    }
    if (weightTotal == 0) { *err = ZSA_corruption_detected; return 0; }
    {   u32 const tableLog = zsa_highbit32(weightTotal) + 1;
        if (tableLog > ZSA_HUF_TABLELOG_MAX) { *err = ZSA_corruption_detected; return 0; }
        *tableLogPtr = tableLog;
        u32 const rest = (1u << tableLog) - weightTotal;
        u32 const verif = 1u << zsa_highbit32(rest);
        u32 const lastWeight = zsa_highbit32(rest) + 1;
        if (verif != rest) { *err = ZSA_corruption_detected; return 0; }
        huffWeight[oSize] = (u8)lastWeight;
        S->rankVal[lastWeight]++;
    }
    if ((S->rankVal[1] < 2) || (S->rankVal[1] & 1)) { *err = ZSA_corruption_detected; return 0; }
    *nbSymbolsPtr = oSize + 1;
    return iSize + 1;
}

// HUF_readStats + HUF_readDTableX1_wksp (entropy_common.c:244-307, huf_decompress.c:358-492) by the whole wave.  Lane 0 reads the
// first byte and, for FSE-compressed weights, the table description and -- once all lanes have laid out that table
// (zsa_fse_spread_wave) -- the weight stream, which is one chain.  Everything after the weights is data-parallel, lane l
// owning symbols l, l + 64, l + 128, l + 192: the number of symbols per weight is a ballot per weight, a symbol's first cell is
// its weight's first cell plus its rank among the symbols of that weight (symbol order, again ballots), and the table is filled by
// each lane for its own short symbols and by all lanes together for the long ones (64 cells and more).
// Returns header size (uniform), 0 on error (S->err set; every failure reads as "corruption detected", zstd_decompress_block.c:224).
ZSA_DEV u32 zsa_huf_buildTable(ZsaDecShared* S, const u8* src, u32 srcSize, int lane)
{
    u8* const huffWeight = S->weights;
    u32 mode = 0, iSize = 0, oSize = 0, wLog = 0, wMax = 0, nc = 0;   // mode 0: error, 1: 4-bit weights, 2: FSE, 3: done by lane 0 alone
    const u8* const hdr = zsa_hdr_stage(S);                              // the description (at most 1 + 127 bytes) in LDS
    zsa_stage_header(S, src, srcSize, lane);
    if (lane == 0 && srcSize) {
        iSize = hdr[0];
        if (iSize >= 128) {
            oSize = iSize - 127; iSize = (oSize + 1) / 2;
            if (iSize + 1 <= srcSize) mode = 1;                  // (oSize <= 128)
        } else if (iSize + 1 <= srcSize) {
            u32 e = 0; wMax = 255;
            nc = zsa_readNCount(S->norm, &wMax, &wLog, hdr + 1, iSize, &e);
            if (!e && wLog <= 6) mode = 2;
        }
    }
    mode = wv_readlane(mode, 0); iSize = wv_readlane(iSize, 0); oSize = wv_readlane(oSize, 0);
    wLog = wv_readlane(wLog, 0); wMax = wv_readlane(wMax, 0); nc = wv_readlane(nc, 0);
    wv_sync();
    if (mode == 1) {
        if (2u * (u32)lane < oSize) { u32 const b = hdr[1 + lane]; huffWeight[2 * lane] = (u8)(b >> 4); huffWeight[2 * lane + 1] = (u8)(b & 15); }
    } else if (mode == 2) {
        if (wMax > 63) {   // more weight values than a valid description can hold: the serial routine decides (and fails later)
            u32 n = 0, tl = 0, h = 0;
            if (lane == 0) { u32 e = 0; h = zsa_huf_readStats(S, &n, &tl, src, srcSize, &e); if (e) n = 0; (void)h; }
            oSize = wv_readlane(n, 0);
            mode = oSize ? 3 : 0;
            if (oSize) oSize -= 1;                               // (readStats counts the implied last symbol)
        } else {
            u32 const tableSize = 1u << wLog;
            ZsaSpreadScratch X; X.first = S->symbolNext; X.seen = S->symbolNext + 64; X.sym = (u8*)S->symStart;
            zsa_fse_spread_wave(X, S->norm, wMax, wLog, lane, [&](u32 u, u32 symbol, u32 rank) {
                int const n = S->norm[symbol];
                u32 const nextState = (n == -1 ? 1u : (u32)n) + rank;
                u32 const nbBits = wLog - zsa_highbit32(nextState);
                S->fseCell[u] = (((nextState << nbBits) - tableSize) & 0xFFFF) | (symbol << 16) | (nbBits << 24);
            });
            wv_sync();
            u32 n = 0;
            if (lane == 0) n = zsa_huf_decodeWeights(S, hdr + 1 + nc, iSize - nc, wLog);
            oSize = wv_readlane(n, 0);
            if (!oSize) mode = 0;
        }
    }
    wv_sync();
    // weights -> symbols per weight, table log, the implied last weight (entropy_common.c:285-305)
    u32 w[4];
    bool tooLarge = false;
    for (u32 k = 0; k < 4; k++) { u32 const n = (u32)lane + 64u * k; w[k] = (mode && n < oSize) ? huffWeight[n] : 0u; tooLarge |= w[k] > ZSA_HUF_TABLELOG_MAX; }
    if (wv_ballot(tooLarge)) mode = 0;
    u32 perWeight[ZSA_HUF_TABLELOG_MAX + 1], weightTotal = 0;
    for (u32 t = 1; t <= ZSA_HUF_TABLELOG_MAX; t++) {
        u32 c = 0;
        for (u32 k = 0; k < 4; k++) c += (u32)__builtin_popcountll(wv_ballot(w[k] == t));
        perWeight[t] = c; weightTotal += c << (t - 1);
    }
    u32 tableLog = 0, lastWeight = 0;
    if (mode && weightTotal) {
        tableLog = zsa_highbit32(weightTotal) + 1;
        u32 const rest = (1u << tableLog) - weightTotal;
        lastWeight = zsa_highbit32(rest) + 1;
        if (tableLog > ZSA_HUF_TABLELOG_MAX || (1u << zsa_highbit32(rest)) != rest) mode = 0;
    } else mode = 0;
    for (u32 t = 1; t <= ZSA_HUF_TABLELOG_MAX; t++) perWeight[t] += (t == lastWeight) ? 1u : 0u;
    if (mode && ((perWeight[1] < 2) || (perWeight[1] & 1))) mode = 0;
    if (!mode) {
        if (lane == 0) S->err = ZSA_corruption_detected;
        wv_sync();
        return 0;
    }
    u32 const nbSymbols = oSize + 1;
    for (u32 k = 0; k < 4; k++) if ((u32)lane + 64u * k == oSize) w[k] = lastWeight;
    if (lane == 0) S->hufLog = tableLog;
    // first cell of every symbol (huf_decompress.c:398-436: symbols sorted by weight, symbol order kept inside a weight)
    u32 first[4] = { 0, 0, 0, 0 };
    {   u32 next = 0, rankStart[ZSA_HUF_TABLELOG_MAX + 1];
        for (u32 t = 1; t <= ZSA_HUF_TABLELOG_MAX; t++) { rankStart[t] = next; next += perWeight[t] << (t - 1); }
        for (u32 k = 0; k < 4; k++)
            for (u32 t = 1; t <= ZSA_HUF_TABLELOG_MAX; t++) {
                u64 const m = wv_ballot(w[k] == t);
                if (w[k] == t) first[k] = rankStart[t] + ((u32)__builtin_popcountll(m & ((1ull << lane) - 1ull)) << (t - 1));
                rankStart[t] += (u32)__builtin_popcountll(m) << (t - 1);
            }
    }
    // fill: symbols of fewer than 64 cells by their lane, the others by all lanes
    for (u32 k = 0; k < 4; k++) {
        u32 const n = (u32)lane + 64u * k;
        u32 const len = w[k] ? (1u << w[k]) >> 1 : 0u;
        u32 const cell = ((tableLog + 1 - w[k]) << 8) | n;
        if (len && len < ZSA_WAVE) for (u32 u = 0; u < len; u++) S->huf[first[k] + u] = (u16)cell;
        for (u64 big = wv_ballot(len >= ZSA_WAVE); big; big &= big - 1) {
            int const j = __builtin_ctzll(big);
            u32 const l2 = wv_readlane(len, j), f2 = wv_readlane(first[k], j), c2 = wv_readlane(cell, j);
            for (u32 u = (u32)lane; u < l2; u += ZSA_WAVE) S->huf[f2 + u] = (u16)c2;
        }
    }
    wv_sync();
    (void)nbSymbols;
    return iSize + 1;
}

// up to 8 bytes of a match: dst[o .. o + len) = history[p .. p + len), where history is the frame's output so far (p >= 0) with the
// dictionary's content in front of it (p < 0).  A full unit that does not straddle the seam is one 8-byte move.
ZSA_DEV void zsa_copy_unit(u8* dst, const u8* dictContent, u32 dictSize, u32 o, s64 p, u32 len)
{
    if (len == 8 && (p >= 0 || p + 8 <= 0)) {
        const u8* const from = p >= 0 ? dst + p : dictContent + ((s64)dictSize + p);
        wv_stg64u(dst + o, wv_ldg64(from));
    } else {
        for (u32 j = 0; j < len; j++) {
            s64 const q = p + j;
            wv_stg8(dst + o + j, q < 0 ? wv_ldg8(dictContent + ((s64)dictSize + q)) : wv_ldg8(dst + q));
        }
    }
}

// Symbol pairs: cell i of the pair table holds the symbol the single table has there and, where the bits that are left of the
// index behind that symbol's code determine a second code completely, that second symbol too (what libzstd's X2 tables do
// for the decoder that uses them; the symbols that come out are the same whichever table decodes them).
// Cell: symbol 1 | symbol 2 << 8 | bits of both << 16 | (second symbol present) << 24.
ZSA_DEV void zsa_huf_buildPairs(ZsaDecShared* S, int lane)
{
    u32 const log = S->hufLog, size = 1u << log;
    u32* const pair = zsa_huf_pairTable(S);
    for (u32 i = (u32)lane; i < size; i += ZSA_WAVE) {
        u32 const c1 = S->huf[i], l1 = c1 >> 8;
        u32 const c2 = S->huf[(i << l1) & (size - 1u)], l2 = c2 >> 8;
        bool const two = l1 + l2 <= log;
        pair[i] = (c1 & 0xFFu) | (two ? (c2 & 0xFFu) << 8 : 0u) | ((two ? l1 + l2 : l1) << 16) | ((two ? 1u : 0u) << 24);
    }
    wv_sync();
}

// huf_decompress.c:514-564: one stream, run by one lane.  Returns 0 ok / 1 corrupted.  `pair`: the pair table, or null.
ZSA_DEV u32 zsa_huf_decodeStream(const ZsaDecShared* S, u8* p, u8* const pEnd, const u8* cSrc, u32 cSrcSize, const u32* pair)
{
    ZsaBitR b;
    if (zsa_br_init(b, cSrc, cSrcSize)) return 1;
    u32 const dtLog = S->hufLog;
#define ZSA_HUF_SYM() do { u32 const c = S->huf[zsa_br_lookFast(b, dtLog)]; *p++ = (u8)c; b.bitsConsumed += c >> 8; } while (0)
    if ((pEnd - p) > 3) {
        // Interior of the stream: the same four-symbols-then-refill steps as the loop below (huf_decompress.c:570-666), with the
        // refill taken out of the dependency chain.  The 8 bytes BELOW the window are requested one step ahead, so a refill is
        // a funnel shift of two registers instead of a wait for memory, and the four symbols leave as one 4-byte store.
        if (zsa_br_reload(b) == ZB_unfinished && b.ptr >= b.start + 16) {
            const u8* ptr = b.ptr; u32 bc = b.bitsConsumed; u64 W = b.container;
            u64 below = wv_ldg64(ptr - 8);
            if (pair) {
                // five cells, then a refill -- a cell yields one or two symbols: both bytes are stored and the
                // output moves on by as many as count (a byte stored in excess is overwritten by the next store; ten bytes
                // of room keep the last one inside this stream's part of the output)
                while (ptr >= b.start + 16 && (size_t)(pEnd - p) >= 10) {
                    u64 X = W << bc;   // (57 bits at least: five cells of up to 11)
#pragma unroll
                    for (u32 k = 0; k < 5; k++) {
                        u32 const c = pair[(u32)(X >> (64u - dtLog))];
                        u32 const nbits = (c >> 16) & 0xFu;
                        X <<= nbits; bc += nbits;
                        wv_stg16(p, c & 0xFFFFu);
                        p += 1u + (c >> 24);
                    }
                    u32 const nb = bc >> 3;
                    bc &= 7;
                    if (nb) W = (W << (8u * nb)) | (below >> (64u - 8u * nb));
                    ptr -= nb;
                    below = wv_ldg64(ptr - 8);
                }
            }
            while (ptr >= b.start + 16 && p < pEnd - 3) {
                u64 X = W << bc;
                u32 out = 0;
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    u32 const c = S->huf[(u32)(X >> (64u - dtLog))];
                    u32 const nbits = c >> 8;
                    X <<= nbits; bc += nbits;
                    out |= (c & 0xFFu) << (8u * k);
                }
                wv_stg32(p, out); p += 4;
                u32 const nb = bc >> 3;            // <= 6: the step stays inside [start + 8, ...), the plain case of the refill
                bc &= 7;
                if (nb) W = (W << (8u * nb)) | (below >> (64u - 8u * nb));
                ptr -= nb;
                below = wv_ldg64(ptr - 8);
            }
            b.ptr = ptr; b.bitsConsumed = bc; b.container = W;
        }
        while ((zsa_br_reload(b) == ZB_unfinished) & (p < pEnd - 3)) { ZSA_HUF_SYM(); ZSA_HUF_SYM(); ZSA_HUF_SYM(); ZSA_HUF_SYM(); }
    } else zsa_br_reload(b);
    while (p < pEnd) ZSA_HUF_SYM();
#undef ZSA_HUF_SYM
    return zsa_br_end(b) ? 0 : 1;
}

// ------------------------------------------------------------------ sequence tables
// zstd_decompress_block.c:455-573 ZSTD_buildFSETable_body by the whole wave: the layout comes from zsa_fse_spread_wave
// (zsa_common.h); a cell's state number is its symbol's cell count plus the cell's rank.  kind: 0 LL, 1 OF, 2 ML.  norm: LDS.
ZSA_DEV u32 zsa_seq_base(int kind, u32 s) { return kind == 0 ? zsa_LL_base(s) : (kind == 1 ? zsa_OF_base(s) : zsa_ML_base(s)); }
ZSA_DEV u32 zsa_seq_bits(int kind, u32 s) { return kind == 0 ? zsa_LL_bits(s) : (kind == 1 ? s : zsa_ML_bits(s)); }
ZSA_DEV void zsa_buildSeqTable(ZsaDecShared* S, ZsaSeqCell* cell, const s16* norm, u32 maxSV, int kind, u32 tableLog, int lane)
{
    u32 const tableSize = 1u << tableLog;
    ZsaSpreadScratch X; X.first = S->symbolNext; X.seen = S->symbolNext + 64; X.sym = (u8*)S->symStart;
    zsa_fse_spread_wave(X, norm, maxSV, tableLog, lane, [&](u32 u, u32 symbol, u32 rank) {
        int const n = norm[symbol];
        u32 const nextState = (n == -1 ? 1u : (u32)n) + rank;
        u32 const nbBits = tableLog - zsa_highbit32(nextState);
        cell[u] = zsa_seqcell(zsa_seq_base(kind, symbol), ((nextState << nbBits) - tableSize) & 0xFFFF, zsa_seq_bits(kind, symbol), nbBits);
    });
    wv_lds_fence();
}
// zstd_decompress_block.c:617-663 ZSTD_buildSeqTable: the table description is read by lane 0, the table is built by all lanes.
// Returns bytes read (wave-uniform); raises S->err.
ZSA_DEV u32 zsa_buildSeqTableFromHeader(ZsaDecShared* S, ZsaSeqCell* cell, u32* logPtr, int type, int kind,
                                        const u8* src, u32 srcSize, int flagRepeatTable, int lane)
{
    u32 const max = kind == 0 ? ZSA_MaxLL : (kind == 1 ? ZSA_MaxOff : ZSA_MaxML);
    u32 const maxLog = kind == 1 ? ZSA_OffFSELog : 9;
    switch (type) {
    case zsa_set_rle:
        if (lane == 0) {
            if (!srcSize) S->err = ZSA_srcSize_wrong;
            else if (src[0] > max) S->err = ZSA_corruption_detected;
            else { cell[0] = zsa_seqcell(zsa_seq_base(kind, src[0]), 0, zsa_seq_bits(kind, src[0]), 0); *logPtr = 0; }
        }
        wv_lds_fence();
        return 1;
    case zsa_set_basic: {
        u32 const dmax = kind == 0 ? ZSA_MaxLL : (kind == 1 ? ZSA_DefaultMaxOff : ZSA_MaxML);
        u32 const dlog = kind == 1 ? ZSA_OF_DEFAULTNORMLOG : 6;
        if ((u32)lane <= dmax)
            S->norm[lane] = (s16)(kind == 0 ? zsa_LL_defaultNorm((u32)lane) : (kind == 1 ? zsa_OF_defaultNorm((u32)lane) : zsa_ML_defaultNorm((u32)lane)));
        if (lane == 0) *logPtr = dlog;
        wv_lds_fence();
        zsa_buildSeqTable(S, cell, S->norm, dmax, kind, dlog, lane);
        return 0; }
    case zsa_set_repeat:
        if (!flagRepeatTable && lane == 0) S->err = ZSA_corruption_detected;
        wv_lds_fence();
        return 0;
    default: {
        u32 tableLog = 0, maxSV = max, h = 0;
        u32 const staged = zsa_stage_header(S, src, srcSize, lane);
        if (lane == 0) {
            u32 err = 0;
            h = zsa_readNCount(S->norm, &maxSV, &tableLog, zsa_hdr_stage(S), staged, &err);
            if (err || tableLog > maxLog) { S->err = ZSA_corruption_detected; tableLog = 0; }
            else *logPtr = tableLog;
        }
        wv_lds_fence();
        tableLog = wv_shfl(tableLog, 0); maxSV = wv_shfl(maxSV, 0); h = wv_shfl(h, 0);
        if (tableLog == 0) return 0;                  // (a valid description has tableLog >= 5)
        zsa_buildSeqTable(S, cell, S->norm, maxSV, kind, tableLog, lane);
        return h; }
    }
}

// ------------------------------------------------------------------ frame context (registers; uniform across lanes)
struct ZsaDecCtx {
    u8* dstBase;          // prefixStart of the current frame
    const u8* dictContent; u32 dictSize;   // bytes logically in front of dstBase
    u32 rep[3];
    int litEntropy, fseEntropy;
    u32 dictID;
    // where the tables a later block may name again ("repeat" / treeless) currently live: the dictionary's digest until
    // a block of this frame builds its own, then the workgroup's workspace (ZsaDecBackup)
    const u16* hufSrc; const ZsaSeqCell* llSrc; const ZsaSeqCell* ofSrc; const ZsaSeqCell* mlSrc;
};

// table moves between LDS and global memory, 16 bytes per lane and step (all tables are 16-byte aligned and padded)
struct __attribute__((aligned(16))) ZsaV16 { u32 a, b, c, d; };
ZSA_DEV void zsa_copy16(void* dst, const void* src, u32 bytes, int lane)
{
    u32 const n = (bytes + 15u) >> 4;
    for (u32 i = (u32)lane; i < n; i += ZSA_WAVE) ((ZsaV16*)dst)[i] = ((const ZsaV16*)src)[i];
}

ZSA_DEV void zsa_raise(ZsaDecShared* S, u32 code) { if (S->err == 0) S->err = code; }

// One compressed block.  op = output offset of the block inside the frame (relative to dstBase),
// cap = remaining capacity.  Returns decoded size (uniform); on error S->err is set.
// lastBlock: no later block of this frame can name this block's tables again, so they are not parked.
ZSA_DEV u32 zsa_decode_block(ZsaDecShared* S, ZsaDecCtx& C, const u8* src, u32 srcSize, u32 op, u64 cap,
                             u8* litBuf, bool lastBlock, int lane)
{
    ZsaDecBackup* const bk = (ZsaDecBackup*)(litBuf + ZSA_DEC_LITBUF_BYTES);
    u32 const expectedWriteSize = cap < ZSA_BLOCKSIZE_MAX ? (u32)cap : ZSA_BLOCKSIZE_MAX;
    ZSA_PROF(S, lane, 0);
    // ---- literals header (zstd_decompress_block.c:120-322), lane 0
    if (lane == 0) {
        u32 err = 0, mode = 0, litSize = 0, litOff = 0, litCSize = 0, streams = 0, consumed = 0;
        if (srcSize > ZSA_BLOCKSIZE_MAX) err = ZSA_srcSize_wrong;
        else if (srcSize < 2) err = ZSA_corruption_detected;
        else {
            mode = src[0] & 3;
            u32 const lhlCode = (src[0] >> 2) & 3;
            if (mode == zsa_set_repeat && C.litEntropy == 0) err = ZSA_dictionary_corrupted;
            else if (mode == zsa_set_compressed || mode == zsa_set_repeat) {
                if (srcSize < 5) err = ZSA_corruption_detected;
                else {
                    u32 const lhc = zsa_ld32(src);
                    u32 lhSize;
                    switch (lhlCode) {
                    case 0: case 1: default: streams = lhlCode ? 4 : 1; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; break;
                    case 2: streams = 4; lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; break;
                    case 3: streams = 4; lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((u32)src[4] << 10); break;
                    }
                    if (litSize > ZSA_BLOCKSIZE_MAX) err = ZSA_corruption_detected;
                    else if (streams == 4 && litSize < 6) err = ZSA_literals_headerWrong;
                    else if (litCSize + lhSize > srcSize) err = ZSA_corruption_detected;
                    else if (expectedWriteSize < litSize) err = ZSA_dstSize_tooSmall;
                    litOff = lhSize; consumed = lhSize + litCSize;
                }
            } else {
                u32 lhSize;
                switch (lhlCode) {
                case 0: case 2: default: lhSize = 1; litSize = src[0] >> 3; break;
                case 1: lhSize = 2; if (mode == zsa_set_rle && srcSize < 3) err = ZSA_corruption_detected; litSize = zsa_ld16(src) >> 4; break;
                case 3: lhSize = 3; if (srcSize < (mode == zsa_set_rle ? 4u : 3u)) err = ZSA_corruption_detected; else litSize = zsa_ld24(src) >> 4; break;
                }
                if (!err) {
                    if (mode == zsa_set_rle && litSize > ZSA_BLOCKSIZE_MAX) err = ZSA_corruption_detected;
                    else if (expectedWriteSize < litSize) err = ZSA_dstSize_tooSmall;
                    else if (mode == zsa_set_basic && litSize + lhSize > srcSize) err = ZSA_corruption_detected;
                }
                litOff = lhSize; consumed = (mode == zsa_set_rle) ? lhSize + 1 : lhSize + litSize;
            }
        }
        S->err = err;
        S->u[ZU_litMode] = mode; S->u[ZU_litSize] = litSize; S->u[ZU_litOff] = litOff; S->u[ZU_litCSize] = litCSize;
        S->u[ZU_streams] = streams; S->u[ZU_seqOff] = consumed;
    }
    wv_sync();
    if (S->err) return 0;
    u32 const litMode = S->u[ZU_litMode], litSize = S->u[ZU_litSize], litOff = S->u[ZU_litOff];
    const u8* litPtr = litBuf;

    if (litMode == zsa_set_basic) {
        litPtr = src + litOff;
    } else if (litMode == zsa_set_rle) {
        u8 const b = src[litOff];
        for (u32 i = (u32)lane; i < litSize; i += ZSA_WAVE) litBuf[i] = b;
    } else {
        u32 const litCSize = S->u[ZU_litCSize], streams = S->u[ZU_streams];
        const u8* cSrc = src + litOff;
        u32 cSrcSize = litCSize;
        if (litMode == zsa_set_compressed) {
            // huf_decompress.c HUF_decompress1X1_DCtx_wksp / HUF_decompress4X_hufOnly_wksp
            if (streams == 4 && (litSize == 0 || cSrcSize == 0)) { if (lane == 0) zsa_raise(S, ZSA_corruption_detected); wv_sync(); return 0; }
            u32 const hSize = zsa_huf_buildTable(S, cSrc, cSrcSize, lane);
            ZSA_PROF(S, lane, 1);
            if (S->err) return 0;
            if (hSize >= cSrcSize) { if (lane == 0) zsa_raise(S, ZSA_corruption_detected); wv_sync(); return 0; }
            cSrc += hSize; cSrcSize -= hSize;
            if (!lastBlock) { zsa_copy16(bk->huf, S->huf, 2u << S->hufLog, lane); C.hufSrc = bk->huf; }   // a later block may repeat it
        } else {   // treeless: the table of an earlier block (or of the dictionary)
            zsa_copy16(S->huf, C.hufSrc, 2u << S->hufLog, lane);
            wv_sync();
        }
        // streams: huf_decompress.c:543-666
        u32 bad = 0;
        const u32* pairs = nullptr;
        if (litSize >= 2048 && S->hufLog <= 11) { zsa_huf_buildPairs(S, lane); pairs = zsa_huf_pairTable(S); }   // (pays from a few hundred symbols per stream on)
        if (streams == 1) {
            if (lane == 0) bad = zsa_huf_decodeStream(S, litBuf, litBuf + litSize, cSrc, cSrcSize, pairs);
        } else {
            if (cSrcSize < 10) bad = 1;
            else {
                u32 const l1 = zsa_ld16(cSrc), l2 = zsa_ld16(cSrc + 2), l3 = zsa_ld16(cSrc + 4);
                u32 const used = l1 + l2 + l3 + 6;
                u32 const segmentSize = (litSize + 3) / 4;
                if (used > cSrcSize) bad = 1;
                else if (3 * segmentSize > litSize) bad = 1;
                else if (lane < 4) {
                    u32 const l4 = cSrcSize - used;
                    u32 const off = 6 + (lane > 0 ? l1 : 0) + (lane > 1 ? l2 : 0) + (lane > 2 ? l3 : 0);
                    u32 const len = lane == 0 ? l1 : (lane == 1 ? l2 : (lane == 2 ? l3 : l4));
                    u8* const o0 = litBuf + (u32)lane * segmentSize;
                    u8* const o1 = lane == 3 ? litBuf + litSize : o0 + segmentSize;
                    bad = zsa_huf_decodeStream(S, o0, o1, cSrc + off, len, pairs);
                }
            }
        }
        if (wv_ballot(bad != 0)) { if (lane == 0) zsa_raise(S, ZSA_corruption_detected); wv_sync(); return 0; }
        C.litEntropy = 1;
    }
    wv_sync();   // literal bytes written by some lanes are read by all below
    ZSA_PROF(S, lane, 2);

    // ---- sequences header (zstd_decompress_block.c:665-742): lane 0 reads the count and the three modes, then the three tables
    // are built one after the other by all lanes
    u32 const seqOff = S->u[ZU_seqOff];
    if (lane == 0) {
        const u8* const istart = src + seqOff;
        u32 const n = srcSize - seqOff;
        const u8* ip = istart;
        const u8* const iend = istart + n;
        u32 err = 0, nbSeq = 0, haveModes = 0;
        if (n < 1) err = ZSA_srcSize_wrong;
        else {
            nbSeq = *ip++;
            if (!nbSeq) { if (n != 1) err = ZSA_srcSize_wrong; }
            else {
                if (nbSeq > 0x7F) {
                    if (nbSeq == 0xFF) { if (ip + 2 > iend) err = ZSA_srcSize_wrong; else { nbSeq = zsa_ld16(ip) + ZSA_LONGNBSEQ; ip += 2; } }
                    else { if (ip >= iend) err = ZSA_srcSize_wrong; else nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
                }
                if (!err && ip + 1 > iend) err = ZSA_srcSize_wrong;
                if (!err) {
                    S->u[ZU_streams] = (u32)(*ip >> 6); S->u[ZU_hufOff] = (u32)((*ip >> 4) & 3); S->u[ZU_hufCSize] = (u32)((*ip >> 2) & 3);
                    ip++;
                    haveModes = 1;
                }
            }
        }
        S->err = err;
        S->u[ZU_nbSeq] = nbSeq;
        S->u[ZU_batchN] = haveModes;
        S->u[ZU_tmp] = (u32)(ip - src);
    }
    wv_sync();
    if (S->err) return 0;
    if (S->u[ZU_batchN]) {
        u32 at = S->u[ZU_tmp];
        int const types[3] = { (int)S->u[ZU_streams], (int)S->u[ZU_hufOff], (int)S->u[ZU_hufCSize] };
        for (int k = 0; k < 3; k++) {   // LL, OF, ML
            ZsaSeqCell* const cell = k == 0 ? S->ll : (k == 1 ? S->of : S->ml);
            u32* const logPtr = k == 0 ? &S->llLog : (k == 1 ? &S->ofLog : &S->mlLog);
            at += zsa_buildSeqTableFromHeader(S, cell, logPtr, types[k], k, src + at, srcSize - at, C.fseEntropy, lane);
            wv_sync();
            if (S->err) { if (lane == 0) S->err = ZSA_corruption_detected; wv_sync(); return 0; }
        }
        if (lane == 0) S->u[ZU_tmp] = at;
    }
    if (lane == 0) {
        if (S->u[ZU_nbSeq] > 0 && cap == 0) S->err = ZSA_dstSize_tooSmall;
        S->u[ZU_seqSize] = srcSize - S->u[ZU_tmp];
    }
    wv_sync();
    if (S->err) return 0;
    if (S->u[ZU_nbSeq]) {   // tables built by this block are parked for later "repeat"s; repeated ones come back from there
        u32 const tLL = S->u[ZU_streams], tOF = S->u[ZU_hufOff], tML = S->u[ZU_hufCSize];
        if (tLL == zsa_set_repeat) zsa_copy16(S->ll, C.llSrc, 8u << S->llLog, lane);
        else if (!lastBlock) { zsa_copy16(bk->ll, S->ll, 8u << S->llLog, lane); C.llSrc = bk->ll; }
        if (tOF == zsa_set_repeat) zsa_copy16(S->of, C.ofSrc, 8u << S->ofLog, lane);
        else if (!lastBlock) { zsa_copy16(bk->of, S->of, 8u << S->ofLog, lane); C.ofSrc = bk->of; }
        if (tML == zsa_set_repeat) zsa_copy16(S->ml, C.mlSrc, 8u << S->mlLog, lane);
        else if (!lastBlock) { zsa_copy16(bk->ml, S->ml, 8u << S->mlLog, lane); C.mlSrc = bk->ml; }
        wv_sync();
    }

    ZSA_PROF(S, lane, 3);
    u32 nbSeq = S->u[ZU_nbSeq];
    u32 outPos = op;                       // relative to C.dstBase
    u64 const oendRel = (u64)op + cap;     // relative end of the writable area
    u32 litPos = 0;
    u8* const dst = C.dstBase;

    if (nbSeq) {
        // lane-0 decoder state lives in registers across batches
        ZsaBitR b; b.container = 0; b.bitsConsumed = 0; b.ptr = b.start = src;
        u32 stLL = 0, stOF = 0, stML = 0, r0 = C.rep[0], r1 = C.rep[1], r2 = C.rep[2];
        C.fseEntropy = 1;
        if (lane == 0) {
            if (zsa_br_init(b, src + S->u[ZU_tmp], S->u[ZU_seqSize])) S->err = ZSA_corruption_detected;
            else {
                stLL = zsa_br_read(b, S->llLog); zsa_br_reload(b);
                stOF = zsa_br_read(b, S->ofLog); zsa_br_reload(b);
                stML = zsa_br_read(b, S->mlLog); zsa_br_reload(b);
            }
        }
        wv_sync();
        if (S->err) return 0;
        while (nbSeq) {
            u32 const n = nbSeq < ZSA_WAVE ? nbSeq : ZSA_WAVE;
            // -- stage the bytes this batch can consume (<= 64 x 89 bits below ptr + 8) into LDS, 16 per lane
            const u8* stageHi; const u8* stageLo;
            {   u64 const p64 = (u64)(size_t)b.ptr;
                u32 const plo = wv_readlane((u32)p64, 0), phi = wv_readlane((u32)(p64 >> 32), 0);
                const u8* const ptr0 = (const u8*)(size_t)(((u64)phi << 32) | plo);
                const u8* const secEnd = src + S->u[ZU_tmp] + S->u[ZU_seqSize];
                stageHi = (ptr0 + 8 <= secEnd) ? ptr0 + 8 : secEnd;
                stageLo = ((size_t)(stageHi - (src + S->u[ZU_tmp])) > ZSA_BITSTAGE) ? stageHi - ZSA_BITSTAGE : src + S->u[ZU_tmp];
                const u8* const c = stageLo + 16u * (u32)lane;
                if (c + 16 <= stageHi) { u64 const a0 = zsa_ld64(c), a1 = zsa_ld64(c + 8); __builtin_memcpy(S->bits + 16u * (u32)lane, &a0, 8); __builtin_memcpy(S->bits + 16u * (u32)lane + 8, &a1, 8); }
                else if (c < stageHi) { for (u32 k = 0; c + k < stageHi; k++) S->bits[16u * (u32)lane + k] = c[k]; }
                wv_lds_fence();   // (LDS only: one wave, LDS operations complete in order)
            }
#define ZSA_RELOAD(b) zsa_br_reload_staged(b, S->bits, stageLo, stageHi)
            // -- zstd_decompress_block.c:1185-1299 ZSTD_decodeSequence, n times on lane 0.  (Measured, round 2: the same loop run by
            // the whole wave on scalar registers -- one LDS round trip and ~70 scalar instructions per sequence -- is no faster:
            // 12 resident waves share the CU's one scalar unit, whereas lane-0 vector code spreads over four SIMDs.)
            // A sequence that starts 16 bytes or more above the staged slice's first byte -- it consumes at most 89 bits, in at most
            // two refills of less than 8 bytes each -- meets none of the reader's end-of-buffer cases, and takes a reader of its own:
            // one bit cursor into the staged bytes, the six fields of a sequence cut from one 64-bit window, and the next
            // sequence's window requested as soon as the three cells tell how many bits this one consumes (the refill then rides
            // on the same LDS round trip as the next cells).  The state it leaves is the one BIT_reloadDStream leaves after every
            // sequence.  The first k sequences of the batch go this way -- all 64 of an interior batch, all but the last few of a
            // section's last batch (which is the only batch of a small frame) -- and the reference's reader below takes the rest.
            // What is serial in a sequence is the walk from one triple of FSE states to the next: three cells, the number of bits
            // they consume, the three state fields.  So lane 0 does just that per sequence and leaves a RECORD -- the 64-bit window
            // at the sequence's first bit and the three states it was decoded in (four words, in the batch arrays the execution
            // stage fills later) -- and the values are cut from the records by all
            // lanes afterwards, one sequence per lane: extra bits, lengths, the offset or the repeat-offset code.  The repeat-offset
            // history is the other chain; it is replayed over the repeat codes only (new offsets between two of them are applied in
            // one step), in uniform control flow.
            u32 k = 0;
#define ZSA_TAKE(v, nb) do { v = (u32)((X >> 1) >> (63u - (nb))); X <<= (nb); } while (0)
            if (lane == 0 && b.bitsConsumed < 8 && b.ptr + 8 <= stageHi && (size_t)(b.ptr - stageLo) >= 16) {
                struct __attribute__((packed)) Pk { u64 v; };
                u32 pOff = (u32)(b.ptr - stageLo), bc = b.bitsConsumed;
                u64 W = ((const Pk*)(S->bits + pOff))->v;
                u32 i = 0;
                // (a sequence moves the cursor down by 12 bytes at most: `safe` sequences need no look at it)
                while (wv_single_lane_b(i < n && pOff >= 16u)) {
                    u32 const safe = (pOff - 16u) / 12u + 1u, iEnd = wv_single_lane_u32((n - i) < safe ? n : i + safe);   // (a scalar: the loop counter stays one)
                    for (; i < iEnd; i++) {
                        ZsaSeqCell const ll = S->ll[stLL], ml = S->ml[stML], of = S->of[stOF];
                        u32 const llH = (u32)(ll >> 32), mlH = (u32)(ml >> 32), ofH = (u32)(of >> 32);
                        // the three high words added as they are: byte 0 = state bits of the sequence (<= 27: no carry), byte 1 = its extra bits
                        u32 const sum = llH + mlH + ofH;
                        u32 const E = (sum >> 8) & 0xFF, T = sum & 0xFF;
                        u64 const X0 = W << bc;
                        u32 const pk = stLL | (stML << 16);
                        u32 pk2 = stOF;
                        u32 sl, sm, so;
                        u32 const used = bc + E + T;
                        if (wv_single_lane_b(E + T <= 32)) {   // the usual case: all six fields lie in the top 32 bits of the window
                            // (v_bfe_u32 takes a field's offset and width from the low five bits of its operands: the high words serve as
                            // widths as they are, and the offsets need to be right modulo 32 only)
                            u32 const top = (u32)(X0 >> 32);
                            u32 const at4 = 0u - E - llH, at5 = at4 - mlH, at6 = at5 - ofH;
                            pOff -= used >> 3; bc = used & 7;
                            W = ((const Pk*)(S->bits + pOff))->v;
                            sl = wv_bfe(top, at4, llH); sm = wv_bfe(top, at5, mlH); so = wv_bfe(top, at6, ofH);
                        } else if (wv_single_lane_b(used <= 64)) {
                            u32 const llS = llH & 31u, mlS = mlH & 31u, ofS = ofH & 31u;
                            u64 X = X0 << E;
                            pOff -= used >> 3; bc = used & 7;
                            W = ((const Pk*)(S->bits + pOff))->v;
                            ZSA_TAKE(sl, llS); ZSA_TAKE(sm, mlS); ZSA_TAKE(so, ofS);
                        } else {   // more than a window holds: refill after the offset and match-length bits (<= 7 + 31 + 16); the
                                   // literal-length bits are then cut here and travel beside the record
                            u32 const ofBits = (ofH >> 8) & 0xFF, mlBits = (mlH >> 8) & 0xFF, llBits = (llH >> 8) & 0xFF;
                            u32 const llS = llH & 31u, mlS = mlH & 31u, ofS = ofH & 31u;
                            u32 llx;
                            {   u32 const u1 = bc + ofBits + mlBits;
                                pOff -= u1 >> 3; bc = u1 & 7;
                                W = ((const Pk*)(S->bits + pOff))->v;
                            }
                            u64 X = W << bc;
                            ZSA_TAKE(llx, llBits); ZSA_TAKE(sl, llS); ZSA_TAKE(sm, mlS); ZSA_TAKE(so, ofS);
                            {   u32 const u2 = bc + llBits + llS + mlS + ofS;
                                pOff -= u2 >> 3; bc = u2 & 7;
                                W = ((const Pk*)(S->bits + pOff))->v;
                            }
                            S->mStart[i] = llx; pk2 |= 0x80000000u;
                        }
                        S->seqLL[i] = (u32)X0; S->seqML[i] = (u32)(X0 >> 32); S->seqOF[i] = pk; S->outStart[i] = pk2;
                        stLL = (llH >> 16) + sl; stML = (mlH >> 16) + sm; stOF = (ofH >> 16) + so;
                    }
                }
                k = i;
                b.ptr = stageLo + pOff; b.bitsConsumed = bc; b.container = W;
                if (k == n) S->u[ZU_batchN] = (nbSeq == n) && (ZSA_RELOAD(b) < ZB_completed);
            }
            k = wv_readlane(k, 0);
            if (k) {
                wv_lds_fence();
                bool const act = (u32)lane < k;
                u32 vLL = 0, vML = 0, vOF = 1, rc = 0;
                bool isN = false;
                if (act) {
                    u32 const pk = S->seqOF[lane], pk2 = S->outStart[lane];
                    u64 X = ((u64)S->seqML[lane] << 32) | S->seqLL[lane];
                    ZsaSeqCell const ll = S->ll[pk & 0xFFFFu], ml = S->ml[pk >> 16], of = S->of[pk2 & 0xFFFFu];
                    u32 const ofBits = zsa_cell_addBits(of), mlBits = zsa_cell_addBits(ml), llBits = zsa_cell_addBits(ll);
                    u32 ofx, mlx, llx;
                    ZSA_TAKE(ofx, ofBits); ZSA_TAKE(mlx, mlBits); ZSA_TAKE(llx, llBits);
                    if (pk2 >> 31) llx = S->mStart[lane];
                    vLL = (u32)ll + llx; vML = (u32)ml + mlx;
                    // zstd_decompress_block.c:1236-1269: a new offset, or a repeat code 0 .. 3 (0: the last offset again, the history
                    // stays; 1 / 2: the one before / two before moves to the front; 3: the last offset minus one)
                    u32 const ll0 = ((u32)ll == 0);
                    if (ofBits > 1) { isN = true; vOF = (u32)of + ofx; }
                    else rc = ofBits == 0 ? ll0 : (u32)of + ll0 + ofx;
                }
#undef ZSA_TAKE
                u64 const newMask = wv_ballot(act && isN), repMask = wv_ballot(act && !isN && rc != 0);
                u32 q0 = wv_readlane(r0, 0), q1 = wv_readlane(r1, 0), q2 = wv_readlane(r2, 0);
                u32 const first0 = q0;
                u64 rm = repMask;
                u32 pos = 0;
                for (;;) {
                    u32 const j = rm ? (u32)__builtin_ctzll(rm) : k;
                    // the new offsets of [pos, j): the last three of them are what is left in the history
                    u64 ns = pos < 64 ? (newMask >> pos) << pos : 0;
                    if (j < 64) ns &= ((u64)1 << j) - 1u;
                    if (ns) {
                        u32 const i1 = 63u - (u32)__builtin_clzll(ns); ns &= ~((u64)1 << i1);
                        u32 const a1 = wv_readlane(vOF, (int)i1);
                        if (ns) {
                            u32 const i2 = 63u - (u32)__builtin_clzll(ns); ns &= ~((u64)1 << i2);
                            u32 const a2 = wv_readlane(vOF, (int)i2);
                            if (ns) { u32 const i3 = 63u - (u32)__builtin_clzll(ns); q2 = wv_readlane(vOF, (int)i3); }
                            else q2 = q0;
                            q1 = a2;
                        } else { q2 = q1; q1 = q0; }
                        q0 = a1;
                    }
                    if (!rm) break;
                    u32 const cj = wv_readlane(rc, (int)j);
                    u32 temp = cj == 3 ? q0 - 1u : (cj == 1 ? q1 : q2);
                    temp += !temp;
                    if (cj != 1) q2 = q1;
                    q1 = q0; q0 = temp;
                    if ((u32)lane == j) vOF = temp;
                    pos = j + 1; rm &= rm - 1;
                }
                {   // repeat code 0: the offset of the nearest earlier sequence that set one, else the one the batch began with
                    u64 const setters = newMask | repMask;
                    u64 const below = setters & (((u64)1 << lane) - 1u);
                    u32 const from = below ? 63u - (u32)__builtin_clzll(below) : 0u;
                    u32 const prev = wv_shfl(vOF, (int)from);
                    if (act && !isN && rc == 0) vOF = below ? prev : first0;
                }
                if (act) { S->seqLL[lane] = vLL; S->seqML[lane] = vML; S->seqOF[lane] = vOF; }
                r0 = q0; r1 = q1; r2 = q2;   // (every lane's copy: lane 0's is the one the serial reader below carries on with)
                wv_lds_fence();
            }
            if (lane == 0 && k < n) {   // the reference's reader, end-of-buffer cases and all, for what the walk above left
                for (u32 i = k; i < n; i++) {
                    ZsaSeqCell const ll = S->ll[stLL], ml = S->ml[stML], of = S->of[stOF];
                    u32 matchLength = (u32)ml, litLength = (u32)ll, offset;
                    u32 const llBits = zsa_cell_addBits(ll), mlBits = zsa_cell_addBits(ml), ofBits = zsa_cell_addBits(of);
                    u32 const totalBits = (llBits + mlBits + ofBits) & 0xFF;
                    if (ofBits > 1) {
                        offset = (u32)of + zsa_br_readFast(b, ofBits);
                        r2 = r1; r1 = r0; r0 = offset;
                    } else {
                        u32 const ll0 = ((u32)ll == 0);
                        if (ofBits == 0) {
                            offset = ll0 ? r1 : r0;
                            r1 = ll0 ? r0 : r1;
                            r0 = offset;
                        } else {
                            u32 const code = (u32)of + ll0 + zsa_br_readFast(b, 1);
                            u32 temp = (code == 3) ? r0 - 1 : (code == 1 ? r1 : (code == 2 ? r2 : r0));
                            temp += !temp;
                            if (code != 1) r2 = r1;
                            r1 = r0;
                            r0 = offset = temp;
                        }
                    }
                    if (mlBits > 0) matchLength += zsa_br_readFast(b, mlBits);
                    if (totalBits >= 57 - (ZSA_LLFSELog + ZSA_MLFSELog + ZSA_OffFSELog)) ZSA_RELOAD(b);
                    if (llBits > 0) litLength += zsa_br_readFast(b, llBits);
                    stLL = zsa_cell_next(ll) + zsa_br_read(b, zsa_cell_nbBits(ll));
                    stML = zsa_cell_next(ml) + zsa_br_read(b, zsa_cell_nbBits(ml));
                    stOF = zsa_cell_next(of) + zsa_br_read(b, zsa_cell_nbBits(of));
                    S->seqLL[i] = litLength; S->seqML[i] = matchLength; S->seqOF[i] = offset;
                    if (i + 1 < n || nbSeq > n) ZSA_RELOAD(b);
                }
                // last batch: exact-end check (zstd_decompress_block.c:1639), reported after the batch has executed
                S->u[ZU_batchN] = (nbSeq == n) && (ZSA_RELOAD(b) < ZB_completed);
            }
#undef ZSA_RELOAD
            ZSA_PROF(S, lane, 4);
            // The one point of a batch that waits for global memory: the output of earlier batches (which this batch's matches
            // read) was stored before the ~64 sequences above were decoded, so the wait is short.  Every other hand-over
            // between lanes below goes through LDS and needs only the in-order guarantee of a single wave.
            wv_sync();
            // -- execute the batch (zstd_decompress_block.c:873-1060 semantics)
            u32 const myLL = (u32)lane < n ? S->seqLL[lane] : 0;
            u32 const myML = (u32)lane < n ? S->seqML[lane] : 0;
            u32 const myOF = (u32)lane < n ? S->seqOF[lane] : 1;
            u32 const endBad = S->u[ZU_batchN];
            u32 const litIncl = wv_scan_incl(myLL);
            u32 const totIncl = wv_scan_incl(myLL + myML);
            u32 const myLitStart = litIncl - myLL, myOutStart = totIncl - (myLL + myML);
            S->litStart[lane] = myLitStart; S->outStart[lane] = myOutStart;
            // per-sequence validity, in the reference's order of checks
            u32 code = 0;
            if ((u64)outPos + totIncl > oendRel) code = ZSA_dstSize_tooSmall;
            else if (litPos + litIncl > litSize) code = ZSA_corruption_detected;
            else if ((u64)myOF > (u64)outPos + myOutStart + myLL + C.dictSize) code = ZSA_corruption_detected;
            unsigned long long const badMask = wv_ballot((u32)lane < n && code != 0);
            if (badMask) {
                int const first = __builtin_ctzll(badMask);
                u32 const c = wv_shfl(code, first);
                if (lane == 0) zsa_raise(S, c);
            }
            wv_lds_fence();
            if (S->err) return 0;
            u32 const totalLit = wv_shfl(litIncl, 63), totalOut = wv_shfl(totIncl, 63);
            // literals of the batch.  Runs of up to 32 bytes (nearly all of them on text-like data) are moved by the lane that owns
            // the sequence, 8 bytes per instruction; the longer ones are flattened over the lanes in units of 8 bytes (the last
            // unit of a run is moved back to end with the run: one move as well).
            if ((u32)lane < n && myLL <= 32) zsa_copy_run32(dst + outPos + myOutStart, litPtr + litPos + myLitStart, myLL);
            if (wv_ballot((u32)lane < n && myLL > 32)) {
                u32 const units = ((u32)lane < n && myLL > 32) ? (myLL + 7) >> 3 : 0;
                u32 const uIncl = wv_scan_incl(units);
                S->mStart[lane] = uIncl - units;
                u32 const totalUnits = wv_shfl(uIncl, 63);
                wv_lds_fence();
                for (u32 t = (u32)lane; t < totalUnits; t += ZSA_WAVE) {
                    u32 lo = 0, hi = n - 1;
                    while (lo < hi) { u32 const mid = (lo + hi + 1) >> 1; if (S->mStart[mid] <= t) lo = mid; else hi = mid - 1; }
                    u32 const ll = S->seqLL[lo];
                    u32 b = 8u * (t - S->mStart[lo]);
                    if (ll - b < 8u) b = ll - 8u;   // (ll > 32)
                    wv_stg64u(dst + outPos + S->outStart[lo] + b, wv_ldg64(litPtr + litPos + S->litStart[lo] + b));
                }
                wv_lds_fence();   // (mStart is written again by the match stage)
            }
            wv_lds_fence();
            ZSA_PROF(S, lane, 5);
            // matches.  Phase I: every match whose source lies entirely before this batch's output (already visible) is
            // independent of the batch -- all of them are copied at once, flattened over the lanes.
            u32 const myO = outPos + myOutStart + myLL;
            u32 const mySpan = myOF < myML ? myOF : myML;
            bool const indep = (u32)lane < n && ((u64)myO + mySpan <= (u64)myOF + outPos);
            // ... of which the short ones that do not overlap their own output and lie inside this frame's output are moved
            // by the lane that owns the sequence, 8 bytes per instruction
            bool const owned = indep && myML <= 32 && myOF >= myML && myO >= myOF;
            if (owned) zsa_copy_run32(dst + myO, dst + myO - myOF, myML);
            // The others are flattened over the lanes in UNITS: 8 bytes of a match that does not overlap its own output (all of
            // its source is history), one byte of a match that does (offset < length: every byte is read from the first period).
            bool const flat = indep && !owned;
            bool const wide = flat && myOF >= myML;
            u32 const units = flat ? (wide ? (myML + 7) >> 3 : myML) : 0;
            u32 const mIncl = wv_ballot(flat) ? wv_scan_incl(units) : 0;
            S->mStart[lane] = mIncl - units;
            S->litStart[lane] = wide ? 1u : 0u;     // (the literal stage is done with this array)
            u64 const depMask = wv_ballot((u32)lane < n && !indep && myML > 0);
            u32 const totalInd = wv_shfl(mIncl, 63);
            wv_lds_fence();
            for (u32 t = (u32)lane; t < totalInd; t += ZSA_WAVE) {
                u32 lo = 0, hi = n - 1;
                while (lo < hi) { u32 const mid = (lo + hi + 1) >> 1; if (S->mStart[mid] <= t) lo = mid; else hi = mid - 1; }
                // lo = last sequence whose units start at or before t (sequences that contribute none share a start
                // with their successor and are skipped by taking the last)
                u32 const k = t - S->mStart[lo];
                u32 const off = S->seqOF[lo], ml = S->seqML[lo];
                u32 const o = outPos + S->outStart[lo] + S->seqLL[lo];
                if (S->litStart[lo]) {
                    // (the last unit of a match of 8 bytes or more is moved back to end with the match: it then overlaps its
                    // predecessor with the same bytes -- source and destination of such a match are disjoint -- and stays one move)
                    u32 b = 8u * k, len = (ml - b) < 8u ? (ml - b) : 8u;
                    if (len < 8u && ml >= 8u) { b = ml - 8u; len = 8u; }
                    zsa_copy_unit(dst, C.dictContent, C.dictSize, o + b, (s64)o - (s64)off + b, len);
                } else {
                    zsa_copy_unit(dst, C.dictContent, C.dictSize, o + k, (s64)o - (s64)off + (k % off), 1);
                }
            }
            // Phase II: the others.  A match may run as soon as every dependent match that produces part of its source has run; the
            // sequences its in-batch source bytes come from are a range of lanes (two searches in the output prefix sums),
            // and any not-yet-done dependent match in that range holds it back.  All matches that are ready run in the same
            // ROUND -- the short ones each on its own lane, so their loads share one memory round trip -- and a round ends with
            // a wave-scope fence: the rounds of a batch are as many as its longest dependency chain, not as many as its
            // dependent matches.
            if (depMask) {
                ZSA_DEP_FENCE();   // phase II may read the literals and the phase-I matches of this batch
                bool const dep = ((depMask >> lane) & 1) != 0;
                u64 needMask = 0;
                if (dep) {
                    s64 const a = (s64)myO - (s64)myOF;
                    u32 const aRel = a <= (s64)outPos ? 0u : (u32)(a - (s64)outPos);
                    u32 const bRel = (u32)(a + (s64)mySpan - (s64)outPos) - 1u;          // (dependent: its source ends inside the batch)
                    u32 sA = 0, sB = 0;
                    for (u32 b = 32; b; b >>= 1) {
                        u32 const tA = sA + b, tB = sB + b;
                        if (tA < n && S->outStart[tA] <= aRel) sA = tA;
                        if (tB < n && S->outStart[tB] <= bRel) sB = tB;
                    }
                    u64 const upTo = sB >= 63 ? ~0ull : ((2ull << sB) - 1ull);
                    needMask = upTo & ~((1ull << sA) - 1ull) & (((u64)1 << lane) - 1ull) & depMask;
                }
                u64 notDone = depMask;
                while (notDone) {
                    bool const ready = dep && ((notDone >> lane) & 1) != 0 && (needMask & notDone) == 0;
                    u64 const R = wv_ballot(ready);      // (never empty: the lowest match not yet done waits for nothing)
                    bool const own = ready && myML <= 32 && myOF >= myML && myO >= myOF;
                    if (own) zsa_copy_run32(dst + myO, dst + myO - myOF, myML);
                    for (u64 m = wv_ballot(ready && !own); m; m &= m - 1) {
                        u32 const i = (u32)__builtin_ctzll(m);
                        u32 const ml = S->seqML[i], off = S->seqOF[i];
                        u32 const o = outPos + S->outStart[i] + S->seqLL[i];
                        if (off >= ml) {   // 8 bytes per lane and step
                            for (u32 b0 = 8u * (u32)lane; b0 < ml; b0 += 8u * ZSA_WAVE) {
                                u32 b = b0, len = (ml - b) < 8u ? (ml - b) : 8u;
                                if (len < 8u && ml >= 8u) { b = ml - 8u; len = 8u; }
                                zsa_copy_unit(dst, C.dictContent, C.dictSize, o + b, (s64)o - (s64)off + b, len);
                            }
                        } else {
                            for (u32 t = (u32)lane; t < ml; t += ZSA_WAVE)
                                zsa_copy_unit(dst, C.dictContent, C.dictSize, o + t, (s64)o - (s64)off + (t % off), 1);
                        }
                    }
                    notDone &= ~R;
                    if (notDone) ZSA_DEP_FENCE();
                }
            }
            outPos += totalOut; litPos += totalLit;
            nbSeq -= n;
            ZSA_PROF(S, lane, 6);
            wv_lds_fence();   // the batch arrays are rewritten by lane 0 next
            if (endBad) { if (lane == 0) zsa_raise(S, ZSA_corruption_detected); wv_sync(); return 0; }
        }
        if (lane == 0) { S->u[ZU_tmp] = r0; S->u[ZU_hufOff] = r1; S->u[ZU_hufCSize] = r2; }
        wv_sync();
        if (S->err) return 0;
        C.rep[0] = S->u[ZU_tmp]; C.rep[1] = S->u[ZU_hufOff]; C.rep[2] = S->u[ZU_hufCSize];
    } else {
        if (S->u[ZU_seqSize] != 0) { /* nbSeq==0 requires the section to be exactly 1 byte: checked above */ }
    }
    // last literals (zstd_decompress_block.c:1644-1651)
    {   u32 const lastLL = litSize - litPos;
        if ((u64)outPos + lastLL > oendRel) { if (lane == 0) zsa_raise(S, ZSA_dstSize_tooSmall); wv_sync(); return 0; }
        for (u32 t = (u32)lane; t < lastLL; t += ZSA_WAVE) wv_stg8(dst + outPos + t, wv_ldg8(litPtr + litPos + t));
        outPos += lastLL;
    }
    wv_sync();
    ZSA_PROF(S, lane, 7);
    return outPos - op;
}

// xxhash.h XXH64 (public algorithm), serial; used only for frames that carry a content checksum
ZSA_DEV u64 zsa_rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
ZSA_DEV u64 zsa_xxh_round(u64 acc, u64 in) { acc += in * 0xC2B2AE3D27D4EB4FULL; acc = zsa_rotl64(acc, 31); return acc * 0x9E3779B185EBCA87ULL; }
ZSA_DEV u64 zsa_xxh_merge(u64 acc, u64 v) { v = zsa_xxh_round(0, v); acc ^= v; return acc * 0x9E3779B185EBCA87ULL + 0x85EBCA77C2B2AE63ULL; }
// XXH64 of a decoded frame (xxhash.c XXH64_digest with seed 0; zstd_decompress.c:1040-1046), by the wave.  The bulk of the
// hash is four independent accumulator chains over interleaved 8-byte words -- one lane each.  64 lanes fetch 512 bytes per
// step with one coalesced load and park them in LDS, from where lanes 0-3 feed their chains, so no chain ever waits for
// global memory; the (< 32 byte) tail and the avalanche are scalar work on lane 0.  Returns the hash in every lane.
ZSA_DEV u64 zsa_xxh64(ZsaDecShared* S, const u8* p, u64 len, int lane)
{
    const u64 P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    const u8* const end = p + len;
    u64* const stage = (u64*)S->seqLL;                 // 64 words (seqLL + seqML), free between blocks
    u64 acc = lane == 0 ? P1 + P2 : (lane == 1 ? P2 : (lane == 2 ? 0 : 0 - P1));
    u64 const nStripes = len >> 5;
    for (u64 s0 = 0; s0 < nStripes; s0 += 16) {
        u32 const cnt = (u32)((nStripes - s0) < 16 ? (nStripes - s0) : 16);
        if ((u32)lane < 4 * cnt) stage[lane] = wv_ldg64(p + 32 * s0 + 8u * (u32)lane);
        wv_lds_fence();
        if (lane < 4) for (u32 k = 0; k < cnt; k++) acc = zsa_xxh_round(acc, stage[4 * k + (u32)lane]);
        wv_lds_fence();
    }
    u32 const lo0 = wv_shfl((u32)acc, 0), hi0 = wv_shfl((u32)(acc >> 32), 0), lo1 = wv_shfl((u32)acc, 1), hi1 = wv_shfl((u32)(acc >> 32), 1);
    u32 const lo2 = wv_shfl((u32)acc, 2), hi2 = wv_shfl((u32)(acc >> 32), 2), lo3 = wv_shfl((u32)acc, 3), hi3 = wv_shfl((u32)(acc >> 32), 3);
    u64 const v1 = ((u64)hi0 << 32) | lo0, v2 = ((u64)hi1 << 32) | lo1, v3 = ((u64)hi2 << 32) | lo2, v4 = ((u64)hi3 << 32) | lo3;
    u64 h;
    if (len >= 32) {
        h = zsa_rotl64(v1, 1) + zsa_rotl64(v2, 7) + zsa_rotl64(v3, 12) + zsa_rotl64(v4, 18);
        h = zsa_xxh_merge(h, v1); h = zsa_xxh_merge(h, v2); h = zsa_xxh_merge(h, v3); h = zsa_xxh_merge(h, v4);
    } else h = P5;
    h += len;
    p += nStripes << 5;
    while (p + 8 <= end) { h ^= zsa_xxh_round(0, zsa_ld64(p)); h = zsa_rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (u64)zsa_ld32(p) * P1; h = zsa_rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = zsa_rotl64(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// point the frame at the dictionary's tables (zstd_ddict.c:59-87 ZSTD_copyDDictParameters)
ZSA_DEV void zsa_load_ddict(ZsaDecShared* S, ZsaDecCtx& C, const ZsaDDictDev* dd, ZsaDecBackup* bk, int lane)
{
    C.dictContent = dd->content; C.dictSize = dd->contentSize; C.dictID = dd->dictID;
    if (dd->entropyPresent) {
        C.litEntropy = C.fseEntropy = 1;
        C.rep[0] = dd->rep[0]; C.rep[1] = dd->rep[1]; C.rep[2] = dd->rep[2];
        C.hufSrc = dd->huf; C.llSrc = dd->ll; C.ofSrc = dd->of; C.mlSrc = dd->ml;   // shared, read-only: nothing is copied per frame
        if (lane == 0) { S->hufLog = dd->hufLog; S->llLog = dd->llLog; S->ofLog = dd->ofLog; S->mlLog = dd->mlLog; }
    }
    wv_sync();
}

// ------------------------------------------------------------------ the frame walk
// zstd_decompress.c:1055-1145 ZSTD_decompressMultiFrame + :947-1053 ZSTD_decompressFrame.
// Returns bytes written or (u64)-error, identical on every lane.
ZSA_DEV u64 zsa_decode_frames(ZsaDecShared* S, const u8* src, u64 srcSize, u8* dst, u64 dstCapacity,
                              u8* litBuf, const ZsaDDictDev* dd)
{
    int const lane = wv_lane();
    u64 written = 0;
    int moreThan1Frame = 0;
    for (;;) {
        if (srcSize < 5) break;
        // ---- lane 0: skippable frame or frame header (zstd_decompress.c:414-547, 583-600)
        if (lane == 0) {
            u64 result = 0; u32 act = 0;   // act: 0 decode, 1 skip, 2 fail
            u32 const magic = zsa_ld32(src);
            S->u[0] = 0;
            if ((magic & ZSA_MAGIC_SKIP_MASK) == ZSA_MAGIC_SKIP_START) {
                act = 1;
                if (srcSize < 8) { act = 2; result = ZSA_ERR(ZSA_srcSize_wrong); }
                else {
                    u32 const sizeU32 = zsa_ld32(src + 4);
                    if ((u32)(sizeU32 + 8) < sizeU32) { act = 2; result = ZSA_ERR(ZSA_frameParameter_unsupported); }
                    else if ((u64)sizeU32 + 8 > srcSize) { act = 2; result = ZSA_ERR(ZSA_srcSize_wrong); }
                    else S->u[1] = sizeU32 + 8;
                }
            } else if (srcSize < 6 + 3) { act = 2; result = ZSA_ERR(ZSA_srcSize_wrong); }
            else {
                u32 const fhd = src[4];
                u32 const dictIDCode = fhd & 3, checksumFlag = (fhd >> 2) & 1, singleSegment = (fhd >> 5) & 1, fcsID = fhd >> 6;
                u32 const didSize = dictIDCode == 3 ? 4 : dictIDCode;
                u32 const fcsSize = fcsID == 0 ? singleSegment : (1u << fcsID);
                u32 const fhs = 5 + !singleSegment + didSize + fcsSize;
                if (srcSize < (u64)fhs + 3) { act = 2; result = ZSA_ERR(ZSA_srcSize_wrong); }
                else if (magic != ZSA_MAGIC) { act = 2; result = moreThan1Frame ? ZSA_ERR(ZSA_srcSize_wrong) : ZSA_ERR(ZSA_prefix_unknown); }
                else if (fhd & 0x08) { act = 2; result = ZSA_ERR(ZSA_frameParameter_unsupported); }
                else {
                    u32 pos = 5, dictID = 0;
                    u64 fcs = ~(u64)0;
                    if (!singleSegment) {
                        u32 const wlByte = src[pos++];
                        if ((wlByte >> 3) + ZSA_WLOG_ABSMIN > ZSA_WLOG_MAX) { act = 2; result = ZSA_ERR(ZSA_frameParameter_windowTooLarge); }
                    }
                    if (didSize == 1) dictID = src[pos]; else if (didSize == 2) dictID = zsa_ld16(src + pos); else if (didSize == 4) dictID = zsa_ld32(src + pos);
                    pos += didSize;
                    if (fcsID == 0) { if (singleSegment) fcs = src[pos]; }
                    else if (fcsID == 1) fcs = zsa_ld16(src + pos) + 256;
                    else if (fcsID == 2) fcs = zsa_ld32(src + pos);
                    else fcs = zsa_ld64(src + pos);
                    u32 const haveDictID = dd ? dd->dictID : 0;
                    if (act != 2 && dictID && haveDictID != dictID) { act = 2; result = ZSA_ERR(ZSA_dictionary_wrong); }   // zstd_decompress.c:698-730
                    S->u[1] = fhs; S->u[2] = checksumFlag; S->u[3] = (u32)fcs; S->u[4] = (u32)(fcs >> 32);
                }
            }
            S->u[0] = act; S->result = result; S->err = 0;
        }
        wv_sync();
        u32 const act = S->u[0];
        if (act == 2) return S->result;
        if (act == 1) { u32 const sk = S->u[1]; src += sk; srcSize -= sk; wv_sync(); continue; }

        u32 const fhs = S->u[1], checksumFlag = S->u[2];
        u64 const fcs = (u64)S->u[3] | ((u64)S->u[4] << 32);
        wv_sync();   // header words are reused below
        ZsaDecCtx C;
        C.dstBase = dst; C.dictContent = dst; C.dictSize = 0;
        C.rep[0] = 1; C.rep[1] = 4; C.rep[2] = 8; C.litEntropy = C.fseEntropy = 0; C.dictID = 0;
        {   ZsaDecBackup* const bk0 = (ZsaDecBackup*)(litBuf + ZSA_DEC_LITBUF_BYTES);
            C.hufSrc = bk0->huf; C.llSrc = bk0->ll; C.ofSrc = bk0->of; C.mlSrc = bk0->ml;
        }
        if (dd) zsa_load_ddict(S, C, dd, (ZsaDecBackup*)(litBuf + ZSA_DEC_LITBUF_BYTES), lane);

        const u8* ip = src + fhs;
        u64 remaining = srcSize - fhs;
        u64 op = 0;
        for (;;) {
            if (remaining < 3) return ZSA_ERR(ZSA_srcSize_wrong);
            u32 const hdr = zsa_ld24(ip);
            u32 const lastBlock = hdr & 1, blockType = (hdr >> 1) & 3, origSize = hdr >> 3;
            if (blockType == zsa_bt_reserved) return ZSA_ERR(ZSA_corruption_detected);
            u32 const cBlockSize = (blockType == zsa_bt_rle) ? 1 : origSize;
            ip += 3; remaining -= 3;
            if (cBlockSize > remaining) return ZSA_ERR(ZSA_srcSize_wrong);
            u64 const cap = dstCapacity - op;
            u32 decoded;
            if (blockType == zsa_bt_compressed) {
                if (op + ZSA_BLOCKSIZE_MAX > 0xFFFFFFFFull) return ZSA_ERR(ZSA_frameParameter_unsupported);   // frames >= 4 GiB: not this path
                decoded = zsa_decode_block(S, C, ip, cBlockSize, (u32)op, cap, litBuf, lastBlock != 0, lane);
                if (S->err) return ZSA_ERR(S->err);
            } else if (blockType == zsa_bt_raw) {
                if (cBlockSize > cap) return ZSA_ERR(ZSA_dstSize_tooSmall);
                for (u32 t = (u32)lane; t < cBlockSize; t += ZSA_WAVE) dst[op + t] = ip[t];
                decoded = cBlockSize;
                wv_sync();
            } else {
                if (origSize > cap) return ZSA_ERR(ZSA_dstSize_tooSmall);
                u8 const bv = ip[0];
                for (u32 t = (u32)lane; t < origSize; t += ZSA_WAVE) dst[op + t] = bv;
                decoded = origSize;
                wv_sync();
            }
            op += decoded;
            ip += cBlockSize; remaining -= cBlockSize;
            if (lastBlock) break;
        }
        if (fcs != ~(u64)0 && op != fcs) return ZSA_ERR(ZSA_corruption_detected);
        if (checksumFlag) {
            if (remaining < 4) return ZSA_ERR(ZSA_checksum_wrong);
            wv_sync();   // (the frame's bytes were stored by many lanes)
            bool const wrong = zsa_ld32(ip) != (u32)zsa_xxh64(S, dst, op, lane);
            wv_sync();
            if (wrong) return ZSA_ERR(ZSA_checksum_wrong);
            ip += 4; remaining -= 4;
        }
        src = ip; srcSize = remaining;
        dst += op; dstCapacity -= op; written += op;
        moreThan1Frame = 1;
    }
    if (srcSize) return ZSA_ERR(ZSA_srcSize_wrong);
    return written;
}
