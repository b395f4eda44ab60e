// zsa_dict.h -- device-side digestion of Zstandard dictionaries, one 64-lane wave per dictionary.
//
// Replaces what the reference does inside ZSTD_createCDict / ZSTD_createDDict (zstd_bulk.go:41-52):
//   DDict   zstd_ddict.c:90-144, zstd_decompress.c:1425-1511 (ZSTD_loadDEntropy)
//   CDict   zstd_compress.c:5314-5365 (ZSTD_initCDict_internal), :4823-4952 (ZSTD_loadCEntropy, ZSTD_loadZstdDictionary),
//           :4682-4802 (ZSTD_loadDictionaryContent), zstd_fast.c:15-48 (ZSTD_fillHashTableForCDict), huf_compress.c:271-321
// The digests stay in device memory; frames compressed/decoded with them read them from there.
// The table builders are the ones the block codecs use (zsa_decode.h / zsa_encode.h).
#pragma once
#include "zsa_decode.h"
#include "zsa_encode.h"

struct __attribute__((aligned(16))) ZsaDDictTables { u16 huf[1 << ZSA_HUF_TABLELOG_MAX]; ZsaSeqCell ll[512], of[256], ml[512]; };

// Returns 0, or an error code (uniform).  `dict` is device memory that outlives the digest.
ZSA_DEV u32 zsa_digest_ddict(ZsaDecShared* S, const u8* dict, u32 dictSize, ZsaDDictDev* out, ZsaDDictTables* T)
{
    int const lane = wv_lane();
    if (lane == 0) {
        // zstd_ddict.c:59-87: the whole dictionary buffer is the addressable history of a frame
        out->content = dict; out->contentSize = dictSize; out->dictID = 0; out->entropyPresent = 0;
        out->hufLog = out->llLog = out->ofLog = out->mlLog = 0;
        out->rep[0] = 1; out->rep[1] = 4; out->rep[2] = 8;
        out->huf = T->huf; out->ll = T->ll; out->of = T->of; out->ml = T->ml;
        S->err = 0;
    }
    wv_sync();
    if (dictSize < 8) return 0;                               // pure content (zstd_ddict.c:98-102)
    if (zsa_ld32(dict) != ZSA_MAGIC_DICT) return 0;           // pure content (:103-109)
    if (dictSize <= 8) return ZSA_dictionary_corrupted;       // zstd_decompress.c:1432
    u32 pos = 8;
    {   u32 const h = zsa_huf_buildTable(S, dict + pos, dictSize - pos, lane);
        if (S->err) return ZSA_dictionary_corrupted;
        pos += h;
    }
    for (u32 i = (u32)lane; i < (1u << S->hufLog); i += ZSA_WAVE) T->huf[i] = S->huf[i];
    wv_sync();   // the sequence tables are built over the same LDS bytes
    // OF, ML, LL in this order (zstd_decompress.c:1455-1498)
    for (int k = 0; k < 3; k++) {   // each description read by lane 0, each table built by all lanes
        int const kind = k == 0 ? 1 : (k == 1 ? 2 : 0);
        ZsaSeqCell* const cell = kind == 0 ? S->ll : (kind == 1 ? S->of : S->ml);
        u32* const logPtr = kind == 0 ? &S->llLog : (kind == 1 ? &S->ofLog : &S->mlLog);
        pos += zsa_buildSeqTableFromHeader(S, cell, logPtr, zsa_set_compressed, kind, dict + pos, dictSize - pos, 0, lane);
        wv_sync();
        if (S->err) return ZSA_dictionary_corrupted;
    }
    if (lane == 0) {
        u32 err = 0, p = pos;
        if (p + 12 > dictSize) err = ZSA_dictionary_corrupted;
        if (!err) {
            u32 const dictContentSize = dictSize - (p + 12);
            for (int i = 0; i < 3; i++) {
                u32 const rep = zsa_ld32(dict + p); p += 4;
                if (rep == 0 || rep > dictContentSize) err = ZSA_dictionary_corrupted;
                out->rep[i] = rep;
            }
        }
        S->err = err ? ZSA_dictionary_corrupted : 0;
    }
    wv_sync();
    if (S->err) return ZSA_dictionary_corrupted;
    for (u32 i = (u32)lane; i < (1u << S->llLog); i += ZSA_WAVE) T->ll[i] = S->ll[i];
    for (u32 i = (u32)lane; i < (1u << S->ofLog); i += ZSA_WAVE) T->of[i] = S->of[i];
    for (u32 i = (u32)lane; i < (1u << S->mlLog); i += ZSA_WAVE) T->ml[i] = S->ml[i];
    if (lane == 0) {
        out->hufLog = S->hufLog; out->llLog = S->llLog; out->ofLog = S->ofLog; out->mlLog = S->mlLog;
        out->dictID = zsa_ld32(dict + 4);
        out->entropyPresent = 1;
    }
    wv_sync();
    return 0;
}

// zstd_compress.c:4809-4821 ZSTD_dictNCountRepeat
ZSA_DEV u32 zsa_dictNCountRepeat(const s16* norm, u32 dictMaxSV, u32 maxSV)
{
    if (dictMaxSV < maxSV) return zsa_rep_check;
    for (u32 s = 0; s <= maxSV; ++s) if (norm[s] == 0) return zsa_rep_check;
    return zsa_rep_valid;
}

// Returns 0, or an error code (uniform).  `hashTable`: device memory, 1 << cp.hlog words (row finder: twice that), `chainTable`:
// 1 << cp.clog words (ZSTD_dfast and the hash-chain finder), both zero-filled by the caller.
ZSA_DEV u32 zsa_digest_cdict(ZsaEncShared* SE, ZsaDecShared* SD, const u8* dict, u32 dictSize, ZsaCParams cp, int level,
                             ZsaCDictDev* out, u32* hashTable, u32* chainTable)
{
    int const lane = wv_lane();
    if (lane == 0) {
        out->content = dict; out->contentSize = 0; out->dictSize = dictSize; out->isLoaded = 0; out->dictID = 0;
        out->level = (level == 0) ? 3 : level;      // zstd_compress.c:5477
        out->cp = cp; out->hashTable = hashTable; out->chainTable = chainTable;
        out->useRow = zsa_useRowFinder(cp.strat, cp.wlog) ? 1u : 0u;
        out->rep[0] = 1; out->rep[1] = 4; out->rep[2] = 8;   // ZSTD_reset_compressedBlockState :1863-1872
        out->hufRepeat = out->llRepeat = out->ofRepeat = out->mlRepeat = zsa_rep_none;
        SD->err = 0;
    }
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { out->huf.nbBits[i] = 0; out->huf.val[i] = 0; }
    wv_sync();
    if (dictSize < 8) return 0;                     // :4969-4972 nothing is loaded
    u32 contentOff = 0;
    if (zsa_ld32(dict) == ZSA_MAGIC_DICT) {
        // ---- ZSTD_loadCEntropy: the descriptions are read by lane 0 (bit-serial), the tables are built by all lanes in LDS
        // (SE->ll/of/ml) and copied out by all lanes.  SD->u[ZU_tmp]: read position; SD->err: first error
        if (lane == 0) {
            u32 err = 0, p = 8;
            out->hufRepeat = zsa_rep_check;
            {   // huf_compress.c:271-321 HUF_readCTable
                u32 nbSymbols = 0, tableLog = 0;
                u32 const h = zsa_huf_readStats(SD, &nbSymbols, &tableLog, dict + p, dictSize - p, &err);
                if (!err) {
                    if (SD->rankVal[0] == 0) out->hufRepeat = zsa_rep_valid;
                    if (nbSymbols < 256) err = ZSA_dictionary_corrupted;     // maxSymbolValue < 255, :4844
                }
                if (!err) {
                    u16 nbPerRank[ZSA_HUF_TABLELOG_MAX + 2], valPerRank[ZSA_HUF_TABLELOG_MAX + 2];
                    for (u32 n = 0; n < ZSA_HUF_TABLELOG_MAX + 2; n++) { nbPerRank[n] = 0; valPerRank[n] = 0; }
                    for (u32 n = 0; n < nbSymbols; n++) {
                        u32 const w = SD->weights[n];
                        u32 const nb = w ? tableLog + 1 - w : 0;
                        SE->hufNext.nbBits[n] = (u8)nb;
                        nbPerRank[nb]++;
                    }
                    {   u16 min = 0;
                        for (u32 n = tableLog; n > 0; n--) { valPerRank[n] = min; min = (u16)(min + nbPerRank[n]); min >>= 1; }
                    }
                    for (u32 n = 0; n < nbSymbols; n++) { u32 const nb = SE->hufNext.nbBits[n]; u16 const v = valPerRank[nb]++; SE->hufNext.val[n] = nb ? v : 0; }
                    p += h;
                }
            }
            SD->err = err ? ZSA_dictionary_corrupted : 0;
            SD->u[ZU_tmp] = p;
        }
        wv_sync();
        for (int k = 0; k < 3; k++) {   // OF, ML, LL (:4848-4888)
            if (SD->err) break;
            int const kind = k == 0 ? 1 : (k == 1 ? 2 : 0);
            ZsaFseCT* const ct = kind == 0 ? &SE->ll : (kind == 1 ? &SE->of : &SE->ml);
            if (lane == 0) {
                u32 err = 0, maxSV = kind == 0 ? ZSA_MaxLL : (kind == 1 ? ZSA_MaxOff : ZSA_MaxML), tableLog = 0;
                u32 const maxLog = kind == 1 ? ZSA_OffFSELog : 9;
                u32 const p = SD->u[ZU_tmp];
                u32 const h = zsa_readNCount(SE->norm, &maxSV, &tableLog, dict + p, dictSize - p, &err);
                if (!err && tableLog > maxLog) err = ZSA_dictionary_corrupted;
                if (err) SD->err = ZSA_dictionary_corrupted;
                else {
                    ct->tableLog = tableLog;
                    if (kind == 2) out->mlRepeat = zsa_dictNCountRepeat(SE->norm, maxSV, ZSA_MaxML);
                    else if (kind == 0) out->llRepeat = zsa_dictNCountRepeat(SE->norm, maxSV, ZSA_MaxLL);
                    else { SD->u[ZU_seqOff] = maxSV; for (u32 s = 0; s <= ZSA_MaxOff; s++) SD->norm[s] = SE->norm[s]; }   // decided below (:4896-4903)
                    SD->u[ZU_tmp] = p + h;
                    // the offset table is built over all MaxOff+1 symbols (:4852-4857); readNCount zero-filled the tail
                    SE->u[EU_s0] = kind == 1 ? ZSA_MaxOff : maxSV;
                    ct->maxSV = SE->u[EU_s0];   // what ZSTD_fseBitCost checks a block's codes against (zstd_compress_sequences.c:115)
                }
            }
            wv_sync();
            if (SD->err) break;
            zsa_fse_buildCTable_wave(SE, ct->nextState, ct->dfs, ct->dnb, SE->norm, SE->u[EU_s0], ct->tableLog, lane);
            wv_sync();
        }
        if (lane == 0 && !SD->err) {
            u32 err = 0, p = SD->u[ZU_tmp];
            u32 const ofMaxSV = SD->u[ZU_seqOff];
            if (p + 12 > dictSize) err = ZSA_dictionary_corrupted;
            if (!err) {
                out->rep[0] = zsa_ld32(dict + p); out->rep[1] = zsa_ld32(dict + p + 4); out->rep[2] = zsa_ld32(dict + p + 8);
                p += 12;
                u32 const dictContentSize = dictSize - p;
                u32 offcodeMax = ZSA_MaxOff;
                if (dictContentSize <= 0xFFFFFFFFu - 128 * 1024) offcodeMax = zsa_highbit32(dictContentSize + 128 * 1024);
                u32 const ofRepeatMax = offcodeMax < ZSA_MaxOff ? offcodeMax : ZSA_MaxOff;
                out->ofRepeat = zsa_dictNCountRepeat(SD->norm, ofMaxSV, ofRepeatMax);
                for (int u = 0; u < 3; u++) if (out->rep[u] == 0 || out->rep[u] > dictContentSize) err = ZSA_dictionary_corrupted;
                out->dictID = zsa_ld32(dict + 4);
            }
            SD->err = err ? ZSA_dictionary_corrupted : 0;
            SD->u[ZU_tmp] = p;
        }
        wv_sync();
        if (SD->err) return ZSA_dictionary_corrupted;
        contentOff = SD->u[ZU_tmp];
        for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { out->huf.nbBits[i] = SE->hufNext.nbBits[i]; out->huf.val[i] = SE->hufNext.val[i]; }
        for (int kind = 0; kind < 3; kind++) {
            const ZsaFseCT* const ct = kind == 0 ? &SE->ll : (kind == 1 ? &SE->of : &SE->ml);
            ZsaFseCT* const o = kind == 0 ? &out->ll : (kind == 1 ? &out->of : &out->ml);
            for (u32 i = (u32)lane; i < 512; i += ZSA_WAVE) o->nextState[i] = ct->nextState[i];
            for (u32 i = (u32)lane; i <= ZSA_MaxSeq; i += ZSA_WAVE) { o->dfs[i] = ct->dfs[i]; o->dnb[i] = ct->dnb[i]; }
            if (lane == 0) { o->tableLog = ct->tableLog; o->maxSV = ct->maxSV; }
        }
        wv_sync();
    }
    // ---- ZSTD_loadDictionaryContent (tfp_forCDict, dtlm_full)
    const u8* content = dict + contentOff;
    u32 contentSize = dictSize - contentOff;
    if (cp.strat <= 2) {   // ZSTD_CDictIndicesAreTagged
        u32 const shortCacheMax = (1u << 24) - 2;    // tagged indices keep 24 bits (:4705-4714)
        if (contentSize > shortCacheMax) { content += contentSize - shortCacheMax; contentSize = shortCacheMax; }
    }
    u32 const endIndex = 2 + contentSize;
    if (lane == 0) { out->content = content; out->contentSize = contentSize; out->isLoaded = 1; }
    {   u32 const m = cp.hlog > cp.clog ? cp.hlog : cp.clog;
        u32 const maxDictSize = 8u << (m < 28 ? m : 28);
        u32 const from = contentSize > maxDictSize ? endIndex - maxDictSize : 2;
        if (endIndex - from > 8 && cp.strat == 2) {
            // zstd_double_fast.c:15-49: every third position enters both tables, the two in between the long table only, and
            // only where its cell is still empty -- order-dependent, so lane 0 applies what all lanes have hashed
            const u8* const dbase = content - 2;
            u32 const hBitsL = cp.hlog + 8, hBitsS = cp.clog + 8, mls = cp.mml;
            for (u32 b = from; b + 10 <= endIndex; b += 3 * ZSA_WAVE) {   // ip + 2 <= end - 8  <=>  index + 10 <= endIndex
                u32 const idx = b + 3 * (u32)lane;
                if (idx + 10 <= endIndex) {
                    u64 const v0 = zsa_ld64(dbase + idx);
                    SD->seqLL[lane] = zsa_hash(v0, hBitsL, 8);
                    SD->seqML[lane] = zsa_hash(zsa_ld64(dbase + idx + 1), hBitsL, 8);
                    SD->seqOF[lane] = zsa_hash(zsa_ld64(dbase + idx + 2), hBitsL, 8);
                    SD->outStart[lane] = zsa_hash(v0, hBitsS, mls);
                }
                wv_sync();
                if (lane == 0) {
                    for (u32 l = 0; l < ZSA_WAVE; l++) {
                        u32 const i0 = b + 3 * l;
                        if (!(i0 + 10 <= endIndex)) break;
                        u32 const h0 = SD->seqLL[l], h1 = SD->seqML[l], h2 = SD->seqOF[l], hs = SD->outStart[l];
                        chainTable[hs >> 8] = (i0 << 8) | (hs & 0xFF);
                        hashTable[h0 >> 8] = (i0 << 8) | (h0 & 0xFF);
                        if (hashTable[h1 >> 8] == 0) hashTable[h1 >> 8] = ((i0 + 1) << 8) | (h1 & 0xFF);
                        if (hashTable[h2 >> 8] == 0) hashTable[h2 >> 8] = ((i0 + 2) << 8) | (h2 & 0xFF);
                    }
                }
                wv_sync();
            }
        } else if (endIndex - from > 8 && cp.strat >= 3) {
            // :4767-4786: every position below end - 8 enters the hash chain (zstd_lazy.c:621-646, mls = minMatch) or the rows
            // (:929-936: hash salt 0, mls capped at 6, no gap rule), 64 positions per step as in the parsers (zsa_lazy.h)
            ZsaLazyCtx C;
            C.S = SE; C.hashTable = hashTable; C.chainTable = chainTable; C.base = content - 2;
            C.hlog = cp.hlog; C.clog = cp.clog; C.wlog = cp.wlog; C.slog = cp.slog;
            C.rowLog = cp.slog < 4 ? 4 : (cp.slog > 6 ? 6 : cp.slog);
            C.rowHashLog = cp.hlog - C.rowLog;
            C.useRow = zsa_useRowFinder(cp.strat, cp.wlog);
            C.mls = C.useRow ? (cp.mml < 6 ? cp.mml : 6) : cp.mml;
            C.lowLimit = 2; C.nextToUpdate = from; C.lazySkipping = 0; C.cN = 0; C.cBase = 0; C.cTag = C.cCell = C.cRel = 0;
            if (C.useRow) zsa_row_insert_range(C, from, endIndex - 8, (1u << C.rowLog) - 1u, lane, 0);
            else (void)zsa_hc_insertAndFind(C, content + contentSize - 8, 0, lane);
        } else if (endIndex - from > 8) {
            // zstd_fast.c:15-48.  Loop bound: ip + 3 < (end - 8) + 2  <=>  index + 9 < endIndex.  All lanes hash a
            // batch of 64 triples (coalesced reads of the content); lane 0 then applies them in the reference's order,
            // because "insert only into an empty cell" makes the result order-dependent.
            const u8* const dbase = content - 2;
            u32 const hBits = cp.hlog + 8, mls = cp.mml;
            for (u32 b = from; b + 9 < endIndex; b += 3 * ZSA_WAVE) {
                u32 const idx = b + 3 * (u32)lane;
                if (idx + 9 < endIndex) {
                    SD->seqLL[lane] = zsa_hash(zsa_ld64(dbase + idx), hBits, mls);
                    SD->seqML[lane] = zsa_hash(zsa_ld64(dbase + idx + 1), hBits, mls);
                    SD->seqOF[lane] = zsa_hash(zsa_ld64(dbase + idx + 2), hBits, mls);
                }
                wv_sync();
                if (lane == 0) {
                    for (u32 l = 0; l < ZSA_WAVE; l++) {
                        u32 const i0 = b + 3 * l;
                        if (!(i0 + 9 < endIndex)) break;
                        u32 const h0 = SD->seqLL[l], h1 = SD->seqML[l], h2 = SD->seqOF[l];
                        hashTable[h0 >> 8] = (i0 << 8) | (h0 & 0xFF);   // ZSTD_writeTaggedIndex
                        if (hashTable[h1 >> 8] == 0) hashTable[h1 >> 8] = ((i0 + 1) << 8) | (h1 & 0xFF);
                        if (hashTable[h2 >> 8] == 0) hashTable[h2 >> 8] = ((i0 + 2) << 8) | (h2 & 0xFF);
                    }
                }
                wv_sync();
            }
        }
    }
    wv_sync();
    return 0;
}
