/* zo_compress.c -- oracle: parameter resolution, ZSTD_fast / ZSTD_dfast parses, block + frame writer.
 * TEST INFRASTRUCTURE ONLY (see zo_common.h).
 */
#include "zo_common.h"
#include "zo_compress.h"
#include "zstd_oracle.h"

/* ------------------------------------------------------------------ parameters */
/* clevels.h:26-131: rows {base for negative levels, 1 .. 12} of the four size classes.  Strategies: 1 fast, 2 dfast,
 * 3 greedy, 4 lazy, 5 lazy2, 6 btlazy2 / 7 btopt (outside this oracle's scope, like the levels above 12: callers reject them). */
#define ZO_MAX_LEVEL 12
static const zo_cparams zo_levelTable[4][ZO_MAX_LEVEL + 1] = {
    {   /* > 256 KB */
        {19,12,13,1,6,1,1}, {19,13,14,1,7,0,1}, {20,15,16,1,6,0,1}, {21,16,17,1,5,0,2}, {21,18,18,1,5,0,2},
        {21,18,19,3,5,2,3}, {21,18,19,3,5,4,4}, {21,19,20,4,5,8,4}, {21,19,20,4,5,16,5}, {22,20,21,4,5,16,5},
        {22,21,22,5,5,16,5}, {22,21,22,6,5,16,5}, {22,22,23,6,5,32,5} },
    {   /* <= 256 KB */
        {18,12,13,1,5,1,1}, {18,13,14,1,6,0,1}, {18,14,14,1,5,0,2}, {18,16,16,1,4,0,2}, {18,16,17,3,5,2,3},
        {18,17,18,5,5,2,3}, {18,18,19,3,5,4,4}, {18,18,19,4,4,4,4}, {18,18,19,4,4,8,5}, {18,18,19,5,4,8,5},
        {18,18,19,6,4,8,5}, {18,18,19,5,4,12,6}, {18,19,19,7,4,12,6} },
    {   /* <= 128 KB */
        {17,12,12,1,5,1,1}, {17,12,13,1,6,0,1}, {17,13,15,1,5,0,1}, {17,15,16,2,5,0,2}, {17,17,17,2,4,0,2},
        {17,16,17,3,4,2,3}, {17,16,17,3,4,4,4}, {17,16,17,3,4,8,5}, {17,16,17,4,4,8,5}, {17,16,17,5,4,8,5},
        {17,16,17,6,4,8,5}, {17,17,17,5,4,8,6}, {17,18,17,7,4,12,6} },
    {   /* <= 16 KB */
        {14,12,13,1,5,1,1}, {14,14,15,1,5,0,1}, {14,14,15,1,4,0,1}, {14,14,15,2,4,0,2}, {14,14,14,4,4,2,3},
        {14,14,14,3,4,4,4}, {14,14,14,4,4,8,5}, {14,14,14,6,4,8,5}, {14,14,14,8,4,8,5}, {14,15,14,5,4,8,6},
        {14,15,14,9,4,8,6}, {14,15,14,3,4,12,7}, {14,15,14,4,3,24,7} },
};

/* zstd_compress.c:1427-1451 ZSTD_dictAndWindowLog */
static u32 zo_dictAndWindowLog(u32 windowLog, u64 srcSize, u64 dictSize)
{
    u64 const maxWindowSize = (u64)1 << ZO_WLOG_MAX;
    if (dictSize == 0) return windowLog;
    {   u64 const windowSize = (u64)1 << windowLog;
        u64 const dictAndWindowSize = dictSize + windowSize;
        if (windowSize >= dictSize + srcSize) return windowLog;
        else if (dictAndWindowSize >= maxWindowSize) return ZO_WLOG_MAX;
        else return zo_highbit32((u32)dictAndWindowSize - 1) + 1;
    }
}
/* zstd_compress.c:1461-1555 ZSTD_adjustCParams_internal (strategy fast/dfast: no row matchfinder, cycleLog = chainLog) */
static zo_cparams zo_adjustCParams(zo_cparams cp, u64 srcSize, size_t dictSize, int mode)
{
    u64 const minSrcSize = 513;
    u64 const maxWindowResize = (u64)1 << (ZO_WLOG_MAX - 1);
    switch (mode) {
    case zo_cpm_createCDict: if (dictSize && srcSize == ZO_CONTENTSIZE_UNKNOWN) srcSize = minSrcSize; break;
    case zo_cpm_attachDict:  dictSize = 0; break;
    default: break;
    }
    if ((srcSize <= maxWindowResize) && (dictSize <= maxWindowResize)) {
        u32 const tSize = (u32)(srcSize + dictSize);
        u32 const hashSizeMin = 1 << 6;
        u32 const srcLog = (tSize < hashSizeMin) ? 6 : zo_highbit32(tSize - 1) + 1;
        if (cp.wlog > srcLog) cp.wlog = srcLog;
    }
    if (srcSize != ZO_CONTENTSIZE_UNKNOWN) {
        u32 const dictAndWindowLog = zo_dictAndWindowLog(cp.wlog, srcSize, (u64)dictSize);
        u32 const cycleLog = cp.clog;   /* ZSTD_cycleLog: chainLog unless btlazy2+ */
        if (cp.hlog > dictAndWindowLog + 1) cp.hlog = dictAndWindowLog + 1;
        if (cycleLog > dictAndWindowLog) cp.clog -= (cycleLog - dictAndWindowLog);
    }
    if (cp.wlog < ZO_WLOG_ABSMIN) cp.wlog = ZO_WLOG_ABSMIN;
    if (mode == zo_cpm_createCDict && (cp.strat == 1 || cp.strat == 2)) {   /* ZSTD_CDictIndicesAreTagged */
        u32 const maxShortCacheHashLog = 32 - 8;
        if (cp.hlog > maxShortCacheHashLog) cp.hlog = maxShortCacheHashLog;
        if (cp.clog > maxShortCacheHashLog) cp.clog = maxShortCacheHashLog;
    }
    return cp;
}
/* zstd_compress.c:6935-6983 ZSTD_getCParamRowSize + ZSTD_getCParams_internal.
 * Returns 0 and fills *out, or 1 when the level is outside the table kept here (> 12, or a btlazy2 row). */
int zo_getCParams(zo_cparams* out, int level, u64 srcSizeHint, size_t dictSize, int mode)
{
    u64 rSize;
    u32 tableID;
    int row;
    {   size_t ds = (mode == zo_cpm_attachDict) ? 0 : dictSize;
        int const unknown = srcSizeHint == ZO_CONTENTSIZE_UNKNOWN;
        size_t const addedSize = unknown && ds > 0 ? 500 : 0;
        rSize = unknown && ds == 0 ? ZO_CONTENTSIZE_UNKNOWN : srcSizeHint + ds + addedSize;
    }
    tableID = (rSize <= 256 * 1024) + (rSize <= 128 * 1024) + (rSize <= 16 * 1024);
    if (level == 0) row = 3;          /* ZSTD_CLEVEL_DEFAULT */
    else if (level < 0) row = 0;
    else if (level > ZO_MAX_LEVEL) return 1;
    else row = level;
    {   zo_cparams cp = zo_levelTable[tableID][row];
        if (cp.strat > 5) return 1;
        if (level < 0) {
            int const minLevel = -(1 << 17);   /* ZSTD_minCLevel() == -ZSTD_TARGETLENGTH_MAX */
            int const clamped = level < minLevel ? minLevel : level;
            cp.tlen = (u32)(-clamped);
        }
        *out = zo_adjustCParams(cp, srcSizeHint, dictSize, mode);
    }
    return 0;
}

/* ------------------------------------------------------------------ match state */
typedef struct {
    const u8* base;        /* window.base: index i <-> base + i */
    const u8* dictBase;    /* window.dictBase: indices in [lowLimit, dictLimit) live at dictBase + i (extDict) */
    u32 dictLimit, lowLimit;
    u32 loadedDictEnd;
    u32* hashTable;
    u32* chainTable;       /* ZSTD_dfast: the short-hash table (zstd_double_fast.c:106) */
    zo_cparams cp;
    const zo_cdict* dms;   /* attached dictionary (dictMatchState), or NULL */
    /* greedy / lazy / lazy2 (zo_lazy.inc) */
    u32 nextToUpdate;      /* first index not yet entered into the tables */
    u32 lazySkipping;
    int useRow;            /* row-hash match finder (windowLog > 14) instead of the hash chain */
    u8* tagTable;          /* row finder: one tag byte per hash-table entry, byte 0 of a row is its head */
    u32 rowHashLog;
    u64 hashSalt;
} zo_matchstate;

/* zstd_compress_internal.h:809-846 ZSTD_hashPtr for mls 4..8 */
static u32 zo_hashPtr(const u8* p, u32 hBits, u32 mls)
{
    switch (mls) {
    default:
    case 4: return (zo_rd32(p) * 2654435761U) >> (32 - hBits);
    case 5: return (u32)(((zo_rd64(p) << (64 - 40)) * 889523592379ULL) >> (64 - hBits));
    case 6: return (u32)(((zo_rd64(p) << (64 - 48)) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (u32)(((zo_rd64(p) << (64 - 56)) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (u32)((zo_rd64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}
/* zstd_compress_internal.h:755-774 ZSTD_count == length of the common prefix, capped at pInLimit */
static size_t zo_count(const u8* pIn, const u8* pMatch, const u8* pInLimit)
{
    const u8* const pStart = pIn;
    while (pIn < pInLimit && *pIn == *pMatch) { pIn++; pMatch++; }
    return (size_t)(pIn - pStart);
}
/* zstd_compress_internal.h:780-793 */
static size_t zo_count_2segments(const u8* ip, const u8* match, const u8* iEnd, const u8* mEnd, const u8* iStart)
{
    const u8* const vEnd = (ip + (mEnd - match) < iEnd) ? ip + (mEnd - match) : iEnd;
    size_t const matchLength = zo_count(ip, match, vEnd);
    if (match + matchLength != mEnd) return matchLength;
    return matchLength + zo_count(ip + matchLength, iStart, iEnd);
}
/* zstd_compress_internal.h:1307-1318 ZSTD_getLowestPrefixIndex */
static u32 zo_lowestPrefixIndex(const zo_matchstate* ms, u32 curr, u32 windowLog)
{
    u32 const maxDistance = 1U << windowLog;
    u32 const lowestValid = ms->dictLimit;
    u32 const withinWindow = (curr - lowestValid > maxDistance) ? curr - maxDistance : lowestValid;
    return (ms->loadedDictEnd != 0) ? lowestValid : withinWindow;
}
/* zstd_compress_internal.h:1290-1302 ZSTD_getLowestMatchIndex */
static u32 zo_lowestMatchIndex(const zo_matchstate* ms, u32 curr, u32 windowLog)
{
    u32 const maxDistance = 1U << windowLog;
    u32 const lowestValid = ms->lowLimit;
    u32 const withinWindow = (curr - lowestValid > maxDistance) ? curr - maxDistance : lowestValid;
    return (ms->loadedDictEnd != 0) ? lowestValid : withinWindow;
}
/* zstd_compress_internal.h:654-712 ZSTD_storeSeq */
static void zo_storeSeq(zo_seqstore* ss, size_t litLength, const u8* literals, u32 offBase, size_t matchLength)
{
    memcpy(ss->lit + ss->litSize, literals, litLength);
    ss->litSize += litLength;
    if (litLength > 0xFFFF) { ss->longLengthType = 1; ss->longLengthPos = ss->nbSeq; }
    ss->seq[ss->nbSeq].litLength = (u16)litLength;
    ss->seq[ss->nbSeq].offBase = offBase;
    {   size_t const mlBase = matchLength - ZO_MINMATCH;
        if (mlBase > 0xFFFF) { ss->longLengthType = 2; ss->longLengthPos = ss->nbSeq; }
        ss->seq[ss->nbSeq].mlBase = (u16)mlBase;
    }
    ss->nbSeq++;
}

/* ------------------------------------------------------------------ the level-1 parse, no dictionary */
/* zstd_fast.c:143-403 ZSTD_compressBlock_fast_noDict_generic.  Returns the size of the last literals run. */
static size_t zo_fast_noDict(zo_matchstate* ms, zo_seqstore* ss, u32 rep[3], const u8* src, size_t srcSize)
{
    u32* const hashTable = ms->hashTable;
    u32 const hlog = ms->cp.hlog, mls = ms->cp.mml;
    int const hasStep = ms->cp.tlen > 1;   /* zstd_fast.c:429 */
    size_t const stepSize = hasStep ? (ms->cp.tlen + !(ms->cp.tlen) + 1) : 2;
    const u8* const base = ms->base;
    const u8* const istart = src;
    u32 const endIndex = (u32)((size_t)(istart - base) + srcSize);
    u32 const prefixStartIndex = zo_lowestPrefixIndex(ms, endIndex, ms->cp.wlog);
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    const u8* anchor = istart;
    const u8* ip0 = istart;
    u32 rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    size_t const kStepIncr = 1 << (8 - 1);

    ip0 += (ip0 == prefixStart);
    {   u32 const curr = (u32)(ip0 - base);
        u32 const windowLow = zo_lowestPrefixIndex(ms, curr, ms->cp.wlog);
        u32 const maxRep = curr - windowLow;
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }

    for (;;) {   /* one search episode per iteration (label _start) */
        size_t step = stepSize;
        const u8* nextStep = ip0 + kStepIncr;
        const u8* ip1 = ip0 + 1;
        const u8* ip2 = ip0 + step;
        const u8* ip3 = ip2 + 1;
        u32 h0, h1, idx, current0 = 0, offcode = 0;
        const u8* match0 = NULL;
        size_t mLength = 0;
        int found = 0;   /* 1: repcode at ip2, 2: hash match at ip0 */

        if (ip3 >= ilimit) break;
        h0 = zo_hashPtr(ip0, hlog, mls);
        h1 = zo_hashPtr(ip1, hlog, mls);
        idx = hashTable[h0];
        do {
            u32 const rval = zo_rd32(ip2 - rep1);
            current0 = (u32)(ip0 - base);
            hashTable[h0] = current0;
            if ((zo_rd32(ip2) == rval) & (rep1 > 0)) {
                ip0 = ip2;
                match0 = ip0 - rep1;
                mLength = ip0[-1] == match0[-1];
                ip0 -= mLength; match0 -= mLength;
                offcode = 1;   /* REPCODE1_TO_OFFBASE */
                mLength += 4;
                hashTable[h1] = (u32)(ip1 - base);
                found = 1;
                break;
            }
            if (idx >= prefixStartIndex && zo_rd32(base + idx) == zo_rd32(ip0)) {
                hashTable[h1] = (u32)(ip1 - base);
                found = 2;
                break;
            }
            idx = hashTable[h1];
            h0 = h1;
            h1 = zo_hashPtr(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = (u32)(ip0 - base);
            hashTable[h0] = current0;
            if (idx >= prefixStartIndex && zo_rd32(base + idx) == zo_rd32(ip0)) {
                if (step <= 4) hashTable[h1] = (u32)(ip1 - base);
                found = 2;
                break;
            }
            idx = hashTable[h1];
            h0 = h1;
            h1 = zo_hashPtr(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2;
            ip2 = ip0 + step;
            ip3 = ip1 + step;
            if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
        } while (ip3 < ilimit);
        if (!found) break;

        if (found == 2) {   /* label _offset */
            match0 = base + idx;
            rep2 = rep1;
            rep1 = (u32)(ip0 - match0);
            offcode = rep1 + 3;   /* OFFSET_TO_OFFBASE */
            mLength = 4;
            while (((ip0 > anchor) & (match0 > prefixStart)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }
        }
        /* label _match */
        mLength += zo_count(ip0 + mLength, match0 + mLength, iend);
        zo_storeSeq(ss, (size_t)(ip0 - anchor), anchor, offcode, mLength);
        ip0 += mLength;
        anchor = ip0;
        if (ip0 <= ilimit) {
            hashTable[zo_hashPtr(base + current0 + 2, hlog, mls)] = current0 + 2;
            hashTable[zo_hashPtr(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
            if (rep2 > 0) {
                while ((ip0 <= ilimit) && (zo_rd32(ip0) == zo_rd32(ip0 - rep2))) {
                    size_t const rLength = zo_count(ip0 + 4, ip0 + 4 - rep2, iend) + 4;
                    { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                    hashTable[zo_hashPtr(ip0, hlog, mls)] = (u32)(ip0 - base);
                    ip0 += rLength;
                    zo_storeSeq(ss, 0, anchor, 1, rLength);
                    anchor = ip0;
                }
            }
        }
    }
    /* label _cleanup: zstd_fast.c:330-355 */
    saved2 = ((saved1 != 0) && (rep1 != 0)) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return (size_t)(iend - anchor);
}

#include "zo_dfast.inc"
#include "zo_lazy.inc"

/* ------------------------------------------------------------------ the level-1 parse, attached dictionary */
/* zstd_fast.c:459-656 ZSTD_compressBlock_fast_dictMatchState_generic (hasStep == 0) */
static size_t zo_fast_dictMatchState(zo_matchstate* ms, zo_seqstore* ss, u32 rep[3], const u8* src, size_t srcSize)
{
    u32* const hashTable = ms->hashTable;
    u32 const hlog = ms->cp.hlog, mls = ms->cp.mml;
    u32 const stepSize = ms->cp.tlen + !(ms->cp.tlen);
    const u8* const base = ms->base;
    const u8* const istart = src;
    const u8* ip0 = istart;
    const u8* ip1 = ip0 + stepSize;
    const u8* anchor = istart;
    u32 const prefixStartIndex = ms->dictLimit;
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    u32 offset_1 = rep[0], offset_2 = rep[1];

    const zo_cdict* const dms = ms->dms;
    const u32* const dictHashTable = dms->hashTable;
    u32 const dictStartIndex = 2;                      /* dms->window.dictLimit */
    const u8* const dictBase = dms->content - 2;       /* dms->window.base */
    const u8* const dictStart = dictBase + dictStartIndex;
    const u8* const dictEnd = dms->content + dms->contentSize;   /* dms->window.nextSrc */
    u32 const dictIndexDelta = prefixStartIndex - (u32)(dictEnd - dictBase);
    u32 const dictAndPrefixLength = (u32)(istart - prefixStart + dictEnd - dictStart);
    u32 const dictHBits = dms->cp.hlog + 8;            /* ZSTD_SHORT_CACHE_TAG_BITS */
    const u8* const kEnd = NULL; (void)kEnd;

    ip0 += (dictAndPrefixLength == 0);
    /* zstd_fast.c:518 asserts offsets fit: they always do with an attached dictionary */

    while (ip1 <= ilimit) {   /* outer loop: repeats after every match */
        size_t mLength;
        size_t hash0 = zo_hashPtr(ip0, hlog, mls);
        size_t const dictHashAndTag0 = zo_hashPtr(ip0, dictHBits, mls);
        u32 dictMatchIndexAndTag = dictHashTable[dictHashAndTag0 >> 8];
        int dictTagsMatch = ((u32)dictMatchIndexAndTag & 0xFF) == ((u32)dictHashAndTag0 & 0xFF);
        u32 matchIndex = hashTable[hash0];
        u32 curr = (u32)(ip0 - base);
        size_t step = stepSize;
        size_t const kStepIncr = 1 << 8;
        const u8* nextStep = ip0 + kStepIncr;
        int matched = 0;

        for (;;) {   /* inner search loop */
            const u8* match = base + matchIndex;
            u32 const repIndex = curr + 1 - offset_1;
            const u8* repMatch = (repIndex < prefixStartIndex) ? dictBase + (repIndex - dictIndexDelta) : base + repIndex;
            size_t const hash1 = zo_hashPtr(ip1, hlog, mls);
            size_t const dictHashAndTag1 = zo_hashPtr(ip1, dictHBits, mls);
            hashTable[hash0] = curr;

            if (((u32)((prefixStartIndex - 1) - repIndex) >= 3) && (zo_rd32(repMatch) == zo_rd32(ip0 + 1))) {
                const u8* const repMatchEnd = repIndex < prefixStartIndex ? dictEnd : iend;
                mLength = zo_count_2segments(ip0 + 1 + 4, repMatch + 4, iend, repMatchEnd, prefixStart) + 4;
                ip0++;
                zo_storeSeq(ss, (size_t)(ip0 - anchor), anchor, 1, mLength);
                matched = 1;
                break;
            }
            if (dictTagsMatch) {
                u32 const dictMatchIndex = dictMatchIndexAndTag >> 8;
                const u8* dictMatch = dictBase + dictMatchIndex;
                if (dictMatchIndex > dictStartIndex && zo_rd32(dictMatch) == zo_rd32(ip0)) {
                    if (matchIndex <= prefixStartIndex) {
                        u32 const offset = (u32)(curr - dictMatchIndex - dictIndexDelta);
                        mLength = zo_count_2segments(ip0 + 4, dictMatch + 4, iend, dictEnd, prefixStart) + 4;
                        while (((ip0 > anchor) & (dictMatch > dictStart)) && (ip0[-1] == dictMatch[-1])) { ip0--; dictMatch--; mLength++; }
                        offset_2 = offset_1;
                        offset_1 = offset;
                        zo_storeSeq(ss, (size_t)(ip0 - anchor), anchor, offset + 3, mLength);
                        matched = 1;
                        break;
                    }
                }
            }
            if (matchIndex > prefixStartIndex && zo_rd32(match) == zo_rd32(ip0)) {
                u32 const offset = (u32)(ip0 - match);
                mLength = zo_count(ip0 + 4, match + 4, iend) + 4;
                while (((ip0 > anchor) & (match > prefixStart)) && (ip0[-1] == match[-1])) { ip0--; match--; mLength++; }
                offset_2 = offset_1;
                offset_1 = offset;
                zo_storeSeq(ss, (size_t)(ip0 - anchor), anchor, offset + 3, mLength);
                matched = 1;
                break;
            }
            dictMatchIndexAndTag = dictHashTable[dictHashAndTag1 >> 8];
            dictTagsMatch = ((u32)dictMatchIndexAndTag & 0xFF) == ((u32)dictHashAndTag1 & 0xFF);
            matchIndex = hashTable[hash1];
            if (ip1 >= nextStep) { step++; nextStep += kStepIncr; }
            ip0 = ip1;
            ip1 = ip1 + step;
            if (ip1 > ilimit) goto cleanup;
            curr = (u32)(ip0 - base);
            hash0 = hash1;
        }
        (void)matched;
        ip0 += mLength;
        anchor = ip0;
        if (ip0 <= ilimit) {
            hashTable[zo_hashPtr(base + curr + 2, hlog, mls)] = curr + 2;
            hashTable[zo_hashPtr(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
            while (ip0 <= ilimit) {
                u32 const current2 = (u32)(ip0 - base);
                u32 const repIndex2 = current2 - offset_2;
                const u8* repMatch2 = repIndex2 < prefixStartIndex ? dictBase - dictIndexDelta + repIndex2 : base + repIndex2;
                if (((u32)((prefixStartIndex - 1) - (u32)repIndex2) >= 3) && (zo_rd32(repMatch2) == zo_rd32(ip0))) {
                    const u8* const repEnd2 = repIndex2 < prefixStartIndex ? dictEnd : iend;
                    size_t const repLength2 = zo_count_2segments(ip0 + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    u32 const tmpOffset = offset_2; offset_2 = offset_1; offset_1 = tmpOffset;
                    zo_storeSeq(ss, 0, anchor, 1, repLength2);
                    hashTable[zo_hashPtr(ip0, hlog, mls)] = current2;
                    ip0 += repLength2;
                    anchor = ip0;
                    continue;
                }
                break;
            }
        }
        ip1 = ip0 + stepSize;
    }
cleanup:
    rep[0] = offset_1;
    rep[1] = offset_2;
    return (size_t)(iend - anchor);
}

/* ------------------------------------------------------------------ the level-1 parse, dictionary as extDict */
/* zstd_fast.c:685-936 ZSTD_compressBlock_fast_extDict_generic: the window is two segments, the dictionary
 * content at dictBase + [lowLimit, dictLimit) and the input at base + [dictLimit, ...). */
static size_t zo_fast_extDict(zo_matchstate* ms, zo_seqstore* ss, u32 rep[3], const u8* src, size_t srcSize)
{
    u32* const hashTable = ms->hashTable;
    u32 const hlog = ms->cp.hlog, mls = ms->cp.mml;
    size_t const stepSize = ms->cp.tlen + !(ms->cp.tlen) + 1;
    const u8* const base = ms->base;
    const u8* const dictBase = ms->dictBase;
    const u8* const istart = src;
    const u8* anchor = istart;
    u32 const endIndex = (u32)((size_t)(istart - base) + srcSize);
    u32 const lowLimit = zo_lowestMatchIndex(ms, endIndex, ms->cp.wlog);
    u32 const dictStartIndex = lowLimit;
    const u8* const dictStart = dictBase + dictStartIndex;
    u32 const dictLimit = ms->dictLimit;
    u32 const prefixStartIndex = dictLimit < lowLimit ? lowLimit : dictLimit;
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const dictEnd = dictBase + prefixStartIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    u32 offset_1 = rep[0], offset_2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0;
    const u8* ip0 = istart;
    size_t const kStepIncr = 1 << (8 - 1);

    if (prefixStartIndex == dictStartIndex) return zo_fast_noDict(ms, ss, rep, src, srcSize);   /* :737-738 */
    {   u32 const curr = (u32)(ip0 - base);
        u32 const maxRep = curr - dictStartIndex;
        if (offset_2 >= maxRep) { offsetSaved2 = offset_2; offset_2 = 0; }
        if (offset_1 >= maxRep) { offsetSaved1 = offset_1; offset_1 = 0; }
    }
    for (;;) {   /* label _start */
        size_t step = stepSize;
        const u8* nextStep = ip0 + kStepIncr;
        const u8* ip1 = ip0 + 1;
        const u8* ip2 = ip0 + step;
        const u8* ip3 = ip2 + 1;
        u32 hash0, hash1, idx, current0 = 0, offcode = 0;
        const u8* idxBase;
        const u8* match0 = NULL;
        const u8* matchEnd = NULL;
        size_t mLength = 0;
        int found = 0;   /* 1 repcode at ip2, 2 table match at ip0 */

        if (ip3 >= ilimit) break;
        hash0 = zo_hashPtr(ip0, hlog, mls);
        hash1 = zo_hashPtr(ip1, hlog, mls);
        idx = hashTable[hash0];
        idxBase = idx < prefixStartIndex ? dictBase : base;
        do {
            {   u32 const current2 = (u32)(ip2 - base);
                u32 const repIndex = current2 - offset_1;
                const u8* const repBase = repIndex < prefixStartIndex ? dictBase : base;
                u32 rval;
                if (((u32)(prefixStartIndex - repIndex) >= 4) & (offset_1 > 0)) rval = zo_rd32(repBase + repIndex);
                else rval = zo_rd32(ip2) ^ 1;
                current0 = (u32)(ip0 - base);
                hashTable[hash0] = current0;
                if (zo_rd32(ip2) == rval) {
                    ip0 = ip2;
                    match0 = repBase + repIndex;
                    matchEnd = repIndex < prefixStartIndex ? dictEnd : iend;
                    mLength = ip0[-1] == match0[-1];
                    ip0 -= mLength; match0 -= mLength;
                    offcode = 1;
                    mLength += 4;
                    found = 1;
                    break;
                }
            }
            {   u32 const mval = idx >= dictStartIndex ? zo_rd32(idxBase + idx) : zo_rd32(ip0) ^ 1;
                if (zo_rd32(ip0) == mval) { found = 2; break; }
            }
            idx = hashTable[hash1];
            idxBase = idx < prefixStartIndex ? dictBase : base;
            hash0 = hash1;
            hash1 = zo_hashPtr(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = (u32)(ip0 - base);
            hashTable[hash0] = current0;
            {   u32 const mval = idx >= dictStartIndex ? zo_rd32(idxBase + idx) : zo_rd32(ip0) ^ 1;
                if (zo_rd32(ip0) == mval) { found = 2; break; }
            }
            idx = hashTable[hash1];
            idxBase = idx < prefixStartIndex ? dictBase : base;
            hash0 = hash1;
            hash1 = zo_hashPtr(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2;
            ip2 = ip0 + step;
            ip3 = ip1 + step;
            if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
        } while (ip3 < ilimit);
        if (!found) break;

        if (found == 2) {   /* label _offset */
            u32 const offset = current0 - idx;
            const u8* const lowMatchPtr = idx < prefixStartIndex ? dictStart : prefixStart;
            matchEnd = idx < prefixStartIndex ? dictEnd : iend;
            match0 = idxBase + idx;
            offset_2 = offset_1;
            offset_1 = offset;
            offcode = offset + 3;
            mLength = 4;
            while (((ip0 > anchor) & (match0 > lowMatchPtr)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }
        }
        /* label _match */
        mLength += zo_count_2segments(ip0 + mLength, match0 + mLength, iend, matchEnd, prefixStart);
        zo_storeSeq(ss, (size_t)(ip0 - anchor), anchor, offcode, mLength);
        ip0 += mLength;
        anchor = ip0;
        if (ip1 < ip0) hashTable[hash1] = (u32)(ip1 - base);
        if (ip0 <= ilimit) {
            hashTable[zo_hashPtr(base + current0 + 2, hlog, mls)] = current0 + 2;
            hashTable[zo_hashPtr(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
            while (ip0 <= ilimit) {
                u32 const repIndex2 = (u32)(ip0 - base) - offset_2;
                const u8* const repMatch2 = repIndex2 < prefixStartIndex ? dictBase + repIndex2 : base + repIndex2;
                if ((((u32)((prefixStartIndex - 1) - repIndex2) >= 3) & (offset_2 > 0)) && (zo_rd32(repMatch2) == zo_rd32(ip0))) {
                    const u8* const repEnd2 = repIndex2 < prefixStartIndex ? dictEnd : iend;
                    size_t const repLength2 = zo_count_2segments(ip0 + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    { u32 const t = offset_2; offset_2 = offset_1; offset_1 = t; }
                    zo_storeSeq(ss, 0, anchor, 1, repLength2);
                    hashTable[zo_hashPtr(ip0, hlog, mls)] = (u32)(ip0 - base);
                    ip0 += repLength2;
                    anchor = ip0;
                    continue;
                }
                break;
            }
        }
    }
    /* label _cleanup :860-874 */
    offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    return (size_t)(iend - anchor);
}

/* ------------------------------------------------------------------ frame writer */
/* zstd_compress.c:4474-4520 ZSTD_writeFrameHeader (contentSizeFlag=1, checksumFlag=0, noDictIDFlag=0) */
static size_t zo_writeFrameHeader(u8* op, u32 windowLog, u64 pledgedSrcSize, u32 dictID)
{
    u32 const dictIDSizeCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
    u32 const windowSize = (u32)1 << windowLog;
    u32 const singleSegment = (windowSize >= pledgedSrcSize);
    u8  const windowLogByte = (u8)((windowLog - ZO_WLOG_ABSMIN) << 3);
    u32 const fcsCode = (pledgedSrcSize >= 256) + (pledgedSrcSize >= 65536 + 256) + (pledgedSrcSize >= 0xFFFFFFFFU);
    size_t pos = 0;
    zo_wr32(op, ZO_MAGIC); pos = 4;
    op[pos++] = (u8)(dictIDSizeCode + (singleSegment << 5) + (fcsCode << 6));
    if (!singleSegment) op[pos++] = windowLogByte;
    switch (dictIDSizeCode) {
    case 1: op[pos] = (u8)dictID; pos++; break;
    case 2: zo_wr16(op + pos, dictID); pos += 2; break;
    case 3: zo_wr32(op + pos, dictID); pos += 4; break;
    default: break;
    }
    switch (fcsCode) {
    case 0: if (singleSegment) op[pos++] = (u8)pledgedSrcSize; break;
    case 1: zo_wr16(op + pos, (u32)(pledgedSrcSize - 256)); pos += 2; break;
    case 2: zo_wr32(op + pos, (u32)pledgedSrcSize); pos += 4; break;
    default: zo_wr64(op + pos, pledgedSrcSize); pos += 8; break;
    }
    return pos;
}

/* zstd_compress.c:3413-3433 ZSTD_isRLE */
static int zo_isRLE(const u8* src, size_t length) { size_t i; for (i = 1; i < length; i++) if (src[i] != src[0]) return 0; return 1; }

static void zo_initEntropyNone(zo_cblockstate* bs)
{   /* zstd_compress.c:1863-1872 ZSTD_reset_compressedBlockState */
    memset(bs, 0, sizeof(*bs));
    bs->rep[0] = 1; bs->rep[1] = 4; bs->rep[2] = 8;
    bs->entropy.huf_repeat = zo_rep_none;
    bs->entropy.ll_repeat = bs->entropy.of_repeat = bs->entropy.ml_repeat = zo_rep_none;
}

/* zstd_fast.c:50-82 ZSTD_fillHashTableForCCtx with ZSTD_dtlm_fast (every third position), and :15-48
 * ZSTD_fillHashTableForCDict with ZSTD_dtlm_full (tagged entries, in-between positions only into empty cells) */
static void zo_fillHashTable(u32* hashTable, const zo_cparams* cp, const u8* base, u32 nextToUpdate, const u8* end, int forCDict)
{
    u32 const hBits = cp->hlog + (forCDict ? 8 : 0);
    u32 const mls = cp->mml;
    const u8* ip = base + nextToUpdate;
    const u8* const iend = end - 8;
    for (; ip + 3 < iend + 2; ip += 3) {
        u32 const curr = (u32)(ip - base);
        u32 const h0 = zo_hashPtr(ip, hBits, mls);
        if (!forCDict) { hashTable[h0] = curr; continue; }
        hashTable[h0 >> 8] = (curr << 8) | (h0 & 0xFF);   /* zstd_compress_internal.h ZSTD_writeTaggedIndex */
        {   u32 p;
            for (p = 1; p < 3; ++p) {
                u32 const h = zo_hashPtr(ip + p, hBits, mls);
                if (hashTable[h >> 8] == 0) hashTable[h >> 8] = ((curr + p) << 8) | (h & 0xFF);
            }
        }
    }
}

enum { zo_dm_none = 0, zo_dm_attach = 1, zo_dm_copy = 2, zo_dm_load = 3 };

/* One frame, begin to end: zstd_compress.c:5001-5046 (begin), :4573-4632 (continue), :4392-4471 (frameChunk),
 * :4222-4283 (block), :5107-5144 (epilogue).  `cdict` may be NULL (dictMode zo_dm_none); otherwise dictMode says how
 * ZSTD_compressBegin_internal brought the dictionary in: attached in place (:2270-2329), tables copied
 * (:2347-2426), or content re-loaded into the working tables (:5027-5044). */
static size_t zo_compressFrame(u8* dst, size_t dstCapacity, const u8* src, size_t srcSize,
                               zo_cparams cp, const zo_cdict* cdict, int dictMode, zo_seq_dump_fn dump, void* dumpArg)
{
    u8* op = dst;
    size_t remaining = srcSize;
    const u8* ip = src;
    size_t blockSize;
    zo_matchstate ms;
    zo_cblockstate* bsPrev = (zo_cblockstate*)malloc(sizeof(zo_cblockstate));
    zo_cblockstate* bsNext = (zo_cblockstate*)malloc(sizeof(zo_cblockstate));
    zo_seqstore ss;
    u8* codeWksp = NULL;
    int isFirstBlock = 1;
    size_t result = 0;
    int const literalsDisabled = (cp.strat == 1) && (cp.tlen > 0);   /* zstd_compress_internal.h:605-618 */
    u32 const maxDist = (u32)1 << cp.wlog;

    memset(&ms, 0, sizeof(ms));
    memset(&ss, 0, sizeof(ss));
    if (!bsPrev || !bsNext) { result = ZO_ERR(memory_allocation); goto done; }
    if (dstCapacity < 18) { result = ZO_ERR(dstSize_tooSmall); goto done; }   /* ZSTD_FRAMEHEADERSIZE_MAX */

    /* zstd_compress.c:2045-2225 ZSTD_resetCCtx_internal */
    {   u64 const ws0 = ((u64)1 << cp.wlog) < (u64)srcSize ? ((u64)1 << cp.wlog) : (u64)srcSize;
        size_t const windowSize = ws0 ? (size_t)ws0 : 1;
        blockSize = windowSize < ZO_BLOCKSIZE_MAX ? windowSize : ZO_BLOCKSIZE_MAX;
    }
    ms.cp = cp;
    ms.hashTable = (u32*)calloc((size_t)1 << cp.hlog, sizeof(u32));
    /* zstd_compress.c:236-252 ZSTD_resolveRowMatchFinderMode (SIMD build) and :1977-1996 */
    ms.useRow = (cp.strat >= 3 && cp.strat <= 5) && (cp.wlog > 14);
    if (cdict != NULL && dictMode != zo_dm_load) ms.useRow = cdict->useRow;   /* :2294, :2364 "cdict overrides" */
    if (cp.strat >= 2 && !ms.useRow) {   /* :262-270: a chain table for every strategy but ZSTD_fast and the row finder */
        ms.chainTable = (u32*)calloc((size_t)1 << cp.clog, sizeof(u32));
        if (!ms.chainTable) { result = ZO_ERR(memory_allocation); goto done; }
    }
    if (ms.useRow) {
        u32 const rowLog = cp.slog < 4 ? 4 : (cp.slog > 6 ? 6 : cp.slog);
        ms.tagTable = (u8*)calloc((size_t)1 << cp.hlog, 1);
        if (!ms.tagTable) { result = ZO_ERR(memory_allocation); goto done; }
        ms.rowHashLog = cp.hlog - rowLog;
        ms.hashSalt = zo_freshHashSalt();
    }
    ss.maxNbSeq = (u32)(blockSize / ((cp.mml == 3) ? 3 : 4)) + 1;
    ss.seq = (zo_seqdef*)malloc((ss.maxNbSeq + 1) * sizeof(zo_seqdef));
    ss.lit = (u8*)malloc(blockSize + 32);
    codeWksp = (u8*)malloc(3 * (size_t)ss.maxNbSeq + 8);
    if (!ms.hashTable || !ss.seq || !ss.lit || !codeWksp) { result = ZO_ERR(memory_allocation); goto done; }
    zo_initEntropyNone(bsPrev);

    /* zstd_compress_internal.h:1232-1285 window_init + first window_update: first byte has index 2 */
    ms.base = src - 2; ms.dictBase = src - 2; ms.dictLimit = 2; ms.lowLimit = 2; ms.loadedDictEnd = 0; ms.dms = NULL;
    ms.nextToUpdate = 2;   /* zstd_compress.c:4596-4599 */
    if (cdict != NULL) {
        u32 const cdictEnd = cdict->endIndex;          /* window.nextSrc - window.base of the dictionary's match state */
        u32 const cdictLen = cdictEnd - 2;
        if (dictMode == zo_dm_attach) {
            /* :2301-2320: the working window is cleared at cdictEnd, then ZSTD_window_update (non contiguous,
             * empty former prefix => no extDict) rebases so that the input starts at index cdictEnd */
            if (cdictLen != 0) {
                ms.dms = cdict;
                ms.base = src - cdictEnd; ms.dictBase = ms.base;
                ms.dictLimit = cdictEnd; ms.lowLimit = cdictEnd;
                ms.loadedDictEnd = cdictEnd;
            }
        } else {
            /* zo_dm_copy :2376-2417 (tables untagged, window/loadedDictEnd copied) or zo_dm_load :5030-5044
             * (content indexed again with the working parameters).  Either way the dictionary content is the
             * window's previous segment [2, cdictEnd) and the input a new, non-contiguous one:
             * zstd_compress_internal.h:1262-1274 turns the former into the extDict. */
            size_t i;
            if (dictMode == zo_dm_copy) {
                for (i = 0; i < ((size_t)1 << cp.hlog); i++) ms.hashTable[i] = cdict->hashTable[i] >> 8;
            } else if (cdict->isLoaded && cdictLen > 8) {
                /* zstd_compress.c:4740-4761: only the suffix the tables can index is inserted */
                u32 const maxDictSize = 8U << (cp.hlog > cp.clog ? (cp.hlog < 28 ? cp.hlog : 28) : (cp.clog < 28 ? cp.clog : 28));
                u32 const from = cdictLen > maxDictSize ? cdictEnd - maxDictSize : 2;
                if (cdictEnd - from > 8)
                    zo_fillHashTable(ms.hashTable, &cp, cdict->content - 2, from, cdict->content + cdictLen, 0);
            }
            if (cdictLen != 0) {
                ms.dictBase = cdict->content - 2;
                ms.base = src - cdictEnd;
                ms.lowLimit = 2; ms.dictLimit = cdictEnd;
                if (ms.dictLimit - ms.lowLimit < 8) ms.lowLimit = ms.dictLimit;   /* too small extDict */
            }
            ms.loadedDictEnd = cdict->isLoaded ? cdictEnd : 0;
        }
        *bsPrev = cdict->cbs;   /* :2326 / :2423 copy block state; zo_dm_load parses the same header again (:4942) */
        ms.nextToUpdate = ms.dictLimit;   /* zstd_compress.c:4596-4600 (a new, non-contiguous segment) */
    }

    op += zo_writeFrameHeader(op, cp.wlog, (u64)srcSize, cdict ? cdict->dictID : 0);
    dstCapacity -= (size_t)(op - dst);

    while (remaining) {
        u32 const lastBlock = (blockSize >= remaining);
        size_t cSize;
        if (dstCapacity < 3 + 2 + 1) { result = ZO_ERR(dstSize_tooSmall); goto done; }
        if (remaining < blockSize) blockSize = remaining;

        /* zstd_compress_internal.h:1199-1230 ZSTD_checkDictValidity */
        {   u32 const blockEndIdx = (u32)((ip + blockSize) - ms.base);
            if (blockEndIdx > ms.loadedDictEnd + maxDist || ms.loadedDictEnd != ms.dictLimit) { ms.loadedDictEnd = 0; ms.dms = NULL; }
        }
        /* zstd_compress_internal.h:1153-1190 ZSTD_window_enforceMaxDist(window, ip, ...) */
        {   u32 const blockEndIdx = (u32)(ip - ms.base);
            if (blockEndIdx > maxDist + ms.loadedDictEnd) {
                u32 const newLowLimit = blockEndIdx - maxDist;
                if (ms.lowLimit < newLowLimit) ms.lowLimit = newLowLimit;
                if (ms.dictLimit < ms.lowLimit) ms.dictLimit = ms.lowLimit;
                ms.loadedDictEnd = 0; ms.dms = NULL;
            }
        }
        if (ms.nextToUpdate < ms.lowLimit) ms.nextToUpdate = ms.lowLimit;   /* zstd_compress.c:4427 */

        /* zstd_compress.c:3147-3306 ZSTD_buildSeqStore */
        if (blockSize < 1 + 1 + 3 + 1 + 1) {   /* MIN_CBLOCK_SIZE + blockHeader + 1 + 1 == 7 */
            cSize = 0;
        } else {
            size_t lastLL;
            ss.nbSeq = 0; ss.litSize = 0; ss.longLengthType = 0;
            bsNext->rep[0] = bsPrev->rep[0]; bsNext->rep[1] = bsPrev->rep[1]; bsNext->rep[2] = bsPrev->rep[2];
            /* zstd_compress_internal.h:971-978 ZSTD_matchState_dictMode */
            /* :3174-3181 limited update after a very long match */
            {   u32 const curr = (u32)(ip - ms.base);
                if (curr > ms.nextToUpdate + 384) {
                    u32 const gap = curr - ms.nextToUpdate - 384;
                    ms.nextToUpdate = curr - (gap < 192 ? gap : 192);
                }
            }
            if (cp.strat >= 3)              lastLL = zo_lazy_parse(&ms, &ss, bsNext->rep, ip, blockSize, cp.strat - 3);   /* no dictionary, or an attached one */
            else if (cp.strat == 2 && ms.dms) lastLL = zo_dfast_dictMatchState(&ms, &ss, bsNext->rep, ip, blockSize);
            else if (cp.strat == 2)         lastLL = zo_dfast_noDict(&ms, &ss, bsNext->rep, ip, blockSize);
            else if (ms.lowLimit < ms.dictLimit) lastLL = zo_fast_extDict(&ms, &ss, bsNext->rep, ip, blockSize);
            else if (ms.dms)                lastLL = zo_fast_dictMatchState(&ms, &ss, bsNext->rep, ip, blockSize);
            else                            lastLL = zo_fast_noDict(&ms, &ss, bsNext->rep, ip, blockSize);
            memcpy(ss.lit + ss.litSize, ip + blockSize - lastLL, lastLL);
            ss.litSize += lastLL;
            if (dump) dump(dumpArg, ss.seq, ss.nbSeq, ss.lit, ss.litSize, bsNext->rep);
            cSize = zo_entropyCompressSeqStore(&ss, &bsPrev->entropy, &bsNext->entropy, literalsDisabled,
                                               op + 3, dstCapacity - 3, blockSize, codeWksp, (int)cp.strat);
            if (zo_isErr(cSize)) { result = cSize; goto done; }
            if (!isFirstBlock && cSize < 25 && zo_isRLE(ip, blockSize)) { cSize = 1; op[3] = ip[0]; }
        }
        if (cSize > 1) { zo_cblockstate* t = bsPrev; bsPrev = bsNext; bsNext = t; }   /* :4272-4274 */
        if (bsPrev->entropy.of_repeat == zo_rep_valid) bsPrev->entropy.of_repeat = zo_rep_check;   /* :4279-4280 */

        if (cSize == 0) {   /* zstd_compress_internal.h:569-579 ZSTD_noCompressBlock */
            if (blockSize + 3 > dstCapacity) { result = ZO_ERR(dstSize_tooSmall); goto done; }
            zo_wr24(op, lastBlock + ((u32)zo_bt_raw << 1) + (u32)(blockSize << 3));
            memcpy(op + 3, ip, blockSize);
            cSize = 3 + blockSize;
        } else {
            u32 const hdr = cSize == 1 ? lastBlock + ((u32)zo_bt_rle << 1) + (u32)(blockSize << 3)
                                       : lastBlock + ((u32)zo_bt_compressed << 1) + (u32)(cSize << 3);
            zo_wr24(op, hdr);
            cSize += 3;
        }
        ip += blockSize; remaining -= blockSize;
        op += cSize; dstCapacity -= cSize;
        isFirstBlock = 0;
    }
    if (srcSize == 0) {   /* :5125-5132 one empty raw last block */
        if (dstCapacity < 4) { result = ZO_ERR(dstSize_tooSmall); goto done; }
        zo_wr24(op, 1 + ((u32)zo_bt_raw << 1));
        op += 3;
    }
    result = (size_t)(op - dst);
done:
    free(ms.hashTable); free(ms.chainTable); free(ms.tagTable); free(ss.seq); free(ss.lit); free(codeWksp); free(bsPrev); free(bsNext);
    return result;
}

/* zstd.h ZSTD_COMPRESSBOUND */
size_t zo_compressBound(size_t srcSize)
{
    if (srcSize >= ((sizeof(size_t) == 8) ? 0xFF00FF00FF00FF00ULL : 0xFF00FF00U)) return ZO_ERR(srcSize_wrong);
    return srcSize + (srcSize >> 8) + ((srcSize < (128 << 10)) ? (((128 << 10) - srcSize) >> 11) : 0);
}

/* zstd_compress.c:5235-5277 ZSTD_compress -> ZSTD_compress_usingDict(dict=NULL) */
size_t zo_compress_dump(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level,
                        zo_seq_dump_fn dump, void* dumpArg)
{
    zo_cparams cp;
    static const u8 empty[8] = {0};
    if (zo_getCParams(&cp, level, (u64)srcSize, 0, zo_cpm_noAttachDict))
        return ZO_ERR(parameter_unsupported);   /* oracle scope: the ZSTD_fast .. ZSTD_lazy2 levels */
    if (srcSize > 0xE0000000u) return ZO_ERR(srcSize_wrong);   /* index overflow correction not restated */
    if (src == NULL) src = empty;
    return zo_compressFrame((u8*)dst, dstCapacity, (const u8*)src, srcSize, cp, NULL, zo_dm_none, dump, dumpArg);
}
size_t zo_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level)
{
    return zo_compress_dump(dst, dstCapacity, src, srcSize, level, NULL, NULL);
}

int zo_cparams_query(int level, unsigned long long srcSize, size_t dictSize, int mode, unsigned out[7])
{
    zo_cparams cp;
    if (zo_getCParams(&cp, level, srcSize, dictSize, mode)) return 1;
    out[0] = cp.wlog; out[1] = cp.clog; out[2] = cp.hlog; out[3] = cp.slog; out[4] = cp.mml; out[5] = cp.tlen; out[6] = cp.strat;
    return 0;
}

/* ------------------------------------------------------------------ dictionary (CDict) */
#include "zo_dict.inc"
