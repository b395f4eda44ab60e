// zsa_params.h -- compression-parameter resolution, usable from host and device code.
// Restates ZSTD_getCParams_internal + ZSTD_adjustCParams_internal (zstd_compress.c:6935-6983, 1427-1555) for the
// clevels.h rows whose strategy this engine implements (ZSTD_fast, ZSTD_dfast, ZSTD_greedy, ZSTD_lazy, ZSTD_lazy2).
#pragma once
#include "zsa_common.h"

struct ZsaCParams { u32 wlog, clog, hlog, slog, mml, tlen, strat; };
enum { zsa_cpm_noAttachDict = 0, zsa_cpm_attachDict = 1, zsa_cpm_createCDict = 2, zsa_cpm_unknown = 3 };
#define ZSA_SIZE_UNKNOWN (~(u64)0)

ZSA_HD u32 zsa_hb32(u32 v) { return 31u - (u32)__builtin_clz(v); }

#define ZSA_MAX_LEVEL 12
ZSA_HD ZsaCParams zsa_levelRow(u32 tableID, u32 row)
{   // clevels.h:26-131, rows {negative-level base, 1 .. 12}.  Strategies: 1 fast, 2 dfast, 3 greedy, 4 lazy, 5 lazy2;
    // 6 btlazy2 and 7 btopt rows are outside this engine
    const u8 t[4][ZSA_MAX_LEVEL + 1][7] = {
        {   // > 256 KB
            {19,12,13,1,6,1,1}, {19,13,14,1,7,0,1}, {20,15,16,1,6,0,1}, {21,16,17,1,5,0,2}, {21,18,18,1,5,0,2},
            {21,18,19,3,5,2,3}, {21,18,19,3,5,4,4}, {21,19,20,4,5,8,4}, {21,19,20,4,5,16,5}, {22,20,21,4,5,16,5},
            {22,21,22,5,5,16,5}, {22,21,22,6,5,16,5}, {22,22,23,6,5,32,5} },
        {   // <= 256 KB
            {18,12,13,1,5,1,1}, {18,13,14,1,6,0,1}, {18,14,14,1,5,0,2}, {18,16,16,1,4,0,2}, {18,16,17,3,5,2,3},
            {18,17,18,5,5,2,3}, {18,18,19,3,5,4,4}, {18,18,19,4,4,4,4}, {18,18,19,4,4,8,5}, {18,18,19,5,4,8,5},
            {18,18,19,6,4,8,5}, {18,18,19,5,4,12,6}, {18,19,19,7,4,12,6} },
        {   // <= 128 KB
            {17,12,12,1,5,1,1}, {17,12,13,1,6,0,1}, {17,13,15,1,5,0,1}, {17,15,16,2,5,0,2}, {17,17,17,2,4,0,2},
            {17,16,17,3,4,2,3}, {17,16,17,3,4,4,4}, {17,16,17,3,4,8,5}, {17,16,17,4,4,8,5}, {17,16,17,5,4,8,5},
            {17,16,17,6,4,8,5}, {17,17,17,5,4,8,6}, {17,18,17,7,4,12,6} },
        {   // <= 16 KB
            {14,12,13,1,5,1,1}, {14,14,15,1,5,0,1}, {14,14,15,1,4,0,1}, {14,14,15,2,4,0,2}, {14,14,14,4,4,2,3},
            {14,14,14,3,4,4,4}, {14,14,14,4,4,8,5}, {14,14,14,6,4,8,5}, {14,14,14,8,4,8,5}, {14,15,14,5,4,8,6},
            {14,15,14,9,4,8,6}, {14,15,14,3,4,12,7}, {14,15,14,4,3,24,7} } };
    ZsaCParams c;
    c.wlog = t[tableID][row][0]; c.clog = t[tableID][row][1]; c.hlog = t[tableID][row][2]; c.slog = t[tableID][row][3];
    c.mml = t[tableID][row][4]; c.tlen = t[tableID][row][5]; c.strat = t[tableID][row][6];
    return c;
}
// zstd_compress.c:223-252: greedy/lazy/lazy2 use the row-hash match finder when windowLog > 14 (a SIMD build, as the
// reference's x86-64 cgo build is), else the hash chain
ZSA_HD bool zsa_useRowFinder(u32 strat, u32 wlog) { return strat >= 3 && strat <= 5 && wlog > 14; }
// bytes of what follows the first 4 << hashLog bytes of a frame's table scratch: ZSTD_dfast's short-hash table, the hash chain,
// or -- row finder -- the rest of the row records.  The row finder's table is laid out in RECORDS of 8 bytes per entry, one per
// row: the row's tag bytes (byte 0 = the head) followed by its cells, so that a search finds a row's tags and cells in the same
// 128-byte line (libzstd keeps two arrays, zstd_compress.c:1938-1996; the layout is private to the match finder and does not
// show in its results).  8 << hashLog bytes in all.
ZSA_HD u32 zsa_auxTableBytes(u32 strat, u32 wlog, u32 hlog, u32 clog)
{
    if (strat < 2) return 0;
    if (zsa_useRowFinder(strat, wlog)) return 4u << hlog;
    return 4u << clog;
}
// zstd_compress.c:1427-1555 ZSTD_dictAndWindowLog + ZSTD_adjustCParams_internal (strategies below btlazy2: cycleLog == chainLog;
// the row finder's hashLog cap, 24 + rowLog, is out of reach of these rows)
ZSA_HD ZsaCParams zsa_adjustCParams(ZsaCParams cp, u64 srcSize, u64 dictSize, int mode)
{
    if (mode == zsa_cpm_createCDict) { if (dictSize && srcSize == ZSA_SIZE_UNKNOWN) srcSize = 513; }
    else if (mode == zsa_cpm_attachDict) dictSize = 0;
    u64 const maxWindowResize = (u64)1 << (ZSA_WLOG_MAX - 1);
    if (srcSize <= maxWindowResize && dictSize <= maxWindowResize) {
        u32 const tSize = (u32)(srcSize + dictSize);
        u32 const srcLog = (tSize < (1u << 6)) ? 6 : zsa_hb32(tSize - 1) + 1;
        if (cp.wlog > srcLog) cp.wlog = srcLog;
    }
    if (srcSize != ZSA_SIZE_UNKNOWN) {
        u32 dawLog = cp.wlog;   // ZSTD_dictAndWindowLog
        if (dictSize) {
            u64 const windowSize = (u64)1 << cp.wlog, dws = dictSize + windowSize;
            if (windowSize >= dictSize + srcSize) dawLog = cp.wlog;
            else if (dws >= ((u64)1 << ZSA_WLOG_MAX)) dawLog = ZSA_WLOG_MAX;
            else dawLog = zsa_hb32((u32)dws - 1) + 1;
        }
        if (cp.hlog > dawLog + 1) cp.hlog = dawLog + 1;
        if (cp.clog > dawLog) cp.clog -= (cp.clog - dawLog);
    }
    if (cp.wlog < ZSA_WLOG_ABSMIN) cp.wlog = ZSA_WLOG_ABSMIN;
    if (mode == zsa_cpm_createCDict && (cp.strat == 1 || cp.strat == 2)) {   // ZSTD_CDictIndicesAreTagged
        if (cp.hlog > 24) cp.hlog = 24;
        if (cp.clog > 24) cp.clog = 24;
    }
    return cp;
}
// zstd_compress.c:6935-6983 ZSTD_getCParams_internal.  Returns 0 ok, 1 level outside the table.
ZSA_HD int zsa_resolveCParams(ZsaCParams* out, int level, u64 srcSizeHint, u64 dictSize, int mode)
{
    u64 rSize;
    {   u64 const ds = (mode == zsa_cpm_attachDict) ? 0 : dictSize;
        int const unknown = srcSizeHint == ZSA_SIZE_UNKNOWN;
        u64 const addedSize = unknown && ds > 0 ? 500 : 0;
        rSize = unknown && ds == 0 ? ZSA_SIZE_UNKNOWN : srcSizeHint + ds + addedSize;
    }
    u32 const tableID = (rSize <= 256 * 1024) + (rSize <= 128 * 1024) + (rSize <= 16 * 1024);
    u32 row;
    if (level == 0) row = 3; else if (level < 0) row = 0; else if (level > ZSA_MAX_LEVEL) return 1; else row = (u32)level;
    ZsaCParams cp = zsa_levelRow(tableID, row);
    if (cp.strat > 5) return 1;
    if (level < 0) { int const minLevel = -(1 << 17); int const cl = level < minLevel ? minLevel : level; cp.tlen = (u32)(-cl); }
    *out = zsa_adjustCParams(cp, srcSizeHint, dictSize, mode);
    return 0;
}
// How ZSTD_compress_usingCDict brings a digested dictionary into one frame, and with which parameters:
// zstd_compress.c:5584-5619 (parameters), :5001-5046 (cdict tables vs re-load), :2254-2267 (attach vs copy).
enum { zsa_dm_none = 0, zsa_dm_attach = 1, zsa_dm_copy = 2, zsa_dm_load = 3 };
ZSA_HD int zsa_resolveCDictUse(ZsaCParams* out, int* dictMode, const ZsaCParams& cdictCp, int cdictLevel, u64 dictSize, u64 srcSize)
{
    bool const useCDictParams = (srcSize < 128 * 1024) || (srcSize < dictSize * 6) || (cdictLevel == 0);
    ZsaCParams cp = cdictCp;
    if (!useCDictParams && zsa_resolveCParams(&cp, cdictLevel, srcSize, dictSize, zsa_cpm_unknown)) return 1;
    {   u32 const limitedSrcSize = (u32)(srcSize < (1u << 19) ? srcSize : (1u << 19));
        u32 const limitedSrcLog = limitedSrcSize > 1 ? zsa_hb32(limitedSrcSize - 1) + 1 : 1;
        if (cp.wlog < limitedSrcLog) cp.wlog = limitedSrcLog;
    }
    if (dictSize > 0 && useCDictParams) {
        // ZSTD_shouldAttachDict: the cut-off of the CDict's strategy (zstd_compress.c:2241-2252)
        u64 const cutoff = cdictCp.strat <= 1 ? 8 * 1024 : (cdictCp.strat == 2 ? 16 * 1024 : 32 * 1024);
        if (srcSize <= cutoff) {
            u32 const windowLog = cp.wlog;
            cp = zsa_adjustCParams(cdictCp, srcSize, dictSize, zsa_cpm_attachDict);
            cp.wlog = windowLog;
            *dictMode = zsa_dm_attach;
        } else *dictMode = zsa_dm_copy;
    } else *dictMode = zsa_dm_load;
    *out = cp;
    return 0;
}
ZSA_HD u64 zsa_bound(u64 srcSize)
{   // zstd.h ZSTD_COMPRESSBOUND
    return srcSize + (srcSize >> 8) + ((srcSize < (128u << 10)) ? (((128u << 10) - srcSize) >> 11) : 0);
}
