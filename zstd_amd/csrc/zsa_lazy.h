// zsa_lazy.h -- the greedy / lazy / lazy2 parses without a dictionary (zstd_lazy.c:1493-1757), with the two match finders
// libzstd 1.5.5 pairs them with: hash chain (windowLog <= 14, :617-761) and row hash (windowLog > 14, :764-1313;
// zstd_compress.c:236-252 picks).  These are levels 5..12 where clevels.h has a greedy/lazy/lazy2 row -- level 5 is the
// default of the reference's Compress() (zstd.go:23,90-92).
//
// Tables, all in the workgroup's global scratch.  Hash chain: hashTable (1 << hashLog cells), then the chain table (1 << chainLog
// cells).  Row finder: one RECORD per row at hashTable, 8 bytes per entry -- the row's tag bytes (byte 0 of every row is the
// row's head) followed by its cells -- so that the tags and the cells of a row share a 128-byte line (zsa_row_tags / zsa_row_cells).
//
// The parser runs in wave-uniform control flow: every lane follows the same chain with the same values (loads of one
// address by all lanes are one transaction) and lane 0 alone stores, so that the one step with real parallelism can use
// the whole wave -- entering the positions a match skipped over into the row table (zsa_row_update, 64 positions per
// round trip instead of one), likewise for the hash chain, and measuring the candidates of a search (one per lane).
// Wherever lane 0 stores into a
// table that all lanes read, a wave fence sits between the reads and the store and another one behind the store: free
// on the GPU (the lanes move in lock step anyway), and the rendezvous that keeps the lanes of the CPU emulator
// (tests/emu, which runs the lanes one after the other between rendezvous) seeing the same table.  Two
// things of the reference are not restated because they cannot change a result (argued in oracle/zo_lazy.inc, which is
// pinned to the reference build): the row finder's 8-entry hash cache -- hashes are taken at the point of use; its one
// visible effect, the parser stopping 8 bytes earlier, is kept -- and the SIMD forms of the tag comparison, replaced by a
// SWAR byte-equality mask over the row's 16/32/64 tag bytes.
// Included by zsa_encode.h.
#pragma once

// ZSTD_count by one lane's loads (every lane of the uniform chain makes the same ones), 16 bytes per round trip
ZSA_DEV u32 zsa_lz_count(const u8* pIn, const u8* pMatch, const u8* const pInLimit)
{
    const u8* const pStart = pIn;
    while (pIn + 16 <= pInLimit) {
        u64 const a0 = wv_ldg64(pIn), a1 = wv_ldg64(pIn + 8), b0 = wv_ldg64(pMatch), b1 = wv_ldg64(pMatch + 8);
        u64 const d0 = a0 ^ b0, d1 = a1 ^ b1;
        if (d0) return (u32)(pIn - pStart) + ((u32)__builtin_ctzll(d0) >> 3);
        if (d1) return (u32)(pIn - pStart) + 8 + ((u32)__builtin_ctzll(d1) >> 3);
        pIn += 16; pMatch += 16;
    }
    while (pIn < pInLimit && wv_ldg8(pIn) == wv_ldg8(pMatch)) { pIn++; pMatch++; }
    return (u32)(pIn - pStart);
}

// the same, with the first 32 bytes of both sides requested together: most matches end there, after one round trip
ZSA_DEV u32 zsa_lz_count32(const u8* pIn, const u8* pMatch, const u8* const pInLimit)
{
    if (pIn + 32 > pInLimit) return zsa_lz_count(pIn, pMatch, pInLimit);
    u64 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = wv_ldg64(pIn + 8 * i); b[i] = wv_ldg64(pMatch + 8 * i); }
#pragma unroll
    for (int i = 0; i < 4; i++) { u64 const d = a[i] ^ b[i]; if (d) return 8u * (u32)i + ((u32)__builtin_ctzll(d) >> 3); }
    return 32 + zsa_lz_count(pIn + 32, pMatch + 32, pInLimit);
}

// zstd_compress.c:1916-1927: the hash salt of a fresh context, ZSTD_advanceHashSalt on hashSalt = hashSaltEntropy = 0
constexpr u64 zsa_bitmix_c(u64 val, u64 len)
{
    val ^= ((val >> 49) | (val << 15)) ^ ((val >> 24) | (val << 40));
    val *= 0x9FB21C651E98DF25ULL;
    val ^= (val >> 35) + len;
    val *= 0x9FB21C651E98DF25ULL;
    return val ^ (val >> 28);
}
constexpr u64 ZSA_FRESH_HASH_SALT = zsa_bitmix_c(0, 8) ^ zsa_bitmix_c(0, 4);

// zstd_compress_internal.h:849-863 ZSTD_hashPtrSalted, mls in 4..6; v = MEM_readLE64(p)
ZSA_DEV u32 zsa_hash_salted(u64 v, u32 hBits, u32 mls, u64 salt)
{
    if (mls == 5) return (u32)((((v << (64 - 40)) * 889523592379ULL) ^ salt) >> (64 - hBits));
    if (mls == 6) return (u32)((((v << (64 - 48)) * 227718039650203ULL) ^ salt) >> (64 - hBits));
    return (((u32)v * 2654435761U) ^ (u32)salt) >> (32 - hBits);
}

struct ZsaLazyCtx {
    ZsaEncShared* S;         // phase accounting only (ZSA_PARSE_PROF builds)
    u32* hashTable; u32* chainTable;
    const u8* base;
    u32 hlog, clog, wlog, slog, mls, rowLog, rowHashLog;
    u32 lowLimit;            // window.lowLimit (no dictionary: loadedDictEnd == 0)
    u32 nextToUpdate;        // first index not yet entered into the tables; lives across the blocks of a frame
    u32 lazySkipping;
    bool useRow;
    // row cache (registers): the rows of positions cBase .. cBase + cN - 1 as they are in memory now.  Lane L holds tag
    // byte and cell (L & rowMask) of the row of position cBase + (L >> rowLog), and that row's offset in cRel.  A search
    // that misses fills it for its own and the next positions in the one round trip it needs anyway; entering a position
    // into a row patches the cached copy, a table update after a match drops it.
    u32 cBase, cN;
    u32 cTag, cCell, cRel;
    // attached dictionary (ZSTD_dictMatchState; read by the <true> instantiations only): its digest, and dictionary + payload as
    // one index space (zsa_parse_dict.h) -- a dictionary index d is the joint index d + dmsDelta, right below the payload's
    const ZsaCDictDev* dms;
    ZsaDmsView V;
    u32 dmsDelta;
};

// 16 source bytes at `p`, kept in registers: literal runs are searched position after position, and the bytes a position
// needs (8 at ip for the hash, 4 at ip+1-offset for the repcode check) mostly lie in what the previous position loaded.
// the record of the row whose first entry has number relRow (= row number << rowLog): tags, then cells
ZSA_DEV u8* zsa_row_tags(const ZsaLazyCtx& C, u32 relRow) { return (u8*)C.hashTable + ((size_t)relRow << 3); }
ZSA_DEV u32* zsa_row_cells(const ZsaLazyCtx& C, u32 relRow) { return (u32*)(zsa_row_tags(C, relRow) + ((size_t)1 << C.rowLog)); }

struct ZsaSrcWin { const u8* p; u32 max8, max4; u64 lo, hi; };
ZSA_DEV void zsa_win_fill(ZsaSrcWin& W, const u8* p, const u8* iend)
{
    bool const both = p + 16 <= iend;
    W.p = p; W.lo = wv_ldg64(p); W.hi = both ? wv_ldg64(p + 8) : 0;
    W.max8 = both ? 8u : 0u; W.max4 = both ? 12u : 4u;
}
ZSA_DEV u64 zsa_win_at(const ZsaSrcWin& W, u32 d)   // 8 bytes at offset d <= 8 (bytes past offset 16 read as zero)
{
    return d == 0 ? W.lo : (d >= 8 ? (W.hi >> (8u * (d - 8u))) : ((W.lo >> (8u * d)) | (W.hi << (64u - 8u * d))));
}
ZSA_DEV u64 zsa_win_get64(ZsaSrcWin& W, const u8* p, const u8* iend)
{
    size_t const d = (size_t)(p - W.p);
    if (!(W.p && d <= W.max8)) { zsa_win_fill(W, p, iend); return W.lo; }
    return zsa_win_at(W, (u32)d);
}
ZSA_DEV u32 zsa_win_get32(ZsaSrcWin& W, const u8* p, const u8* iend)
{
    size_t const d = (size_t)(p - W.p);
    if (!(W.p && d <= W.max4)) { zsa_win_fill(W, p, iend); return (u32)W.lo; }
    return (u32)zsa_win_at(W, (u32)d);
}

// ---------------------------------------------------------------- candidate evaluation, one candidate per lane
// Both finders walk their candidates in a fixed order and keep the first one that is strictly longer than everything
// before it (zstd_lazy.c:696-720, :1232-1258).  The reference skips the full comparison of a candidate whose 4 bytes
// at [ml-3, ml+1) differ -- such a candidate cannot be longer than ml, so measuring every candidate in full and taking
// the first maximum is the same selection; its early exit when a match reaches the end of the input changes nothing
// either (nothing can be longer).  Lane k measures candidate k; `ml` and *offBasePtr are updated like the loop would.
ZSA_DEV void zsa_lazy_evaluate(const ZsaLazyCtx& C, const u8* ip, const u8* iLimit, u32 curr, u32 myIndex, u32 nCand,
                               u32* ml, u32* offBasePtr, int lane)
{
    if (nCand == 0) return;
    bool const mine = (u32)lane < nCand;
    u32 const len = mine ? zsa_lz_count(ip, C.base + myIndex, iLimit) : 0u;
    u32 best = *ml, which = ZSA_WAVE;
    for (u32 k = 0; k < nCand; k++) { u32 const l = wv_readlane(len, (int)k); if (l > best) { best = l; which = k; } }   // first maximum
    if (which < ZSA_WAVE) {
        *ml = best;
        *offBasePtr = (curr - wv_readlane(myIndex, (int)which)) + 3;
    }
}

// zstd_compress_internal.h:780-793 ZSTD_count_2segments by one lane's loads: a match that starts in the dictionary [.., mEnd) and
// may run on into the payload from iStart
ZSA_DEV u32 zsa_lz_count_2segments(const u8* ip, const u8* match, const u8* iEnd, const u8* mEnd, const u8* iStart)
{
    const u8* const vEnd = (ip + (mEnd - match) < iEnd) ? ip + (mEnd - match) : iEnd;
    u32 const n = zsa_lz_count(ip, match, vEnd);
    if (match + n != mEnd) return n;
    return n + zsa_lz_count(ip + n, iStart, iEnd);
}
// The same selection over candidates of the attached dictionary (zstd_lazy.c:739-757, :1291-1309): myIndex is an index of the
// dictionary's own window; a candidate counts only if its first 4 bytes agree (a full measurement of one that does not stays
// below 4, never above ml >= 3); the offset is taken in the joint index space.
ZSA_DEV void zsa_lazy_evaluate_dms(const ZsaLazyCtx& C, const u8* ip, const u8* iLimit, u32 curr, u32 myIndex, u32 nCand,
                                   u32* ml, u32* offBasePtr, int lane)
{
    if (nCand == 0) return;
    bool const mine = (u32)lane < nCand;
    const u8* const dmsBase = C.dms->content - 2;
    const u8* const dmsEnd = C.dms->content + C.dms->contentSize;
    u32 const len = mine ? zsa_lz_count_2segments(ip, dmsBase + myIndex, iLimit, dmsEnd, C.base + C.V.prefix) : 0u;
    u32 best = *ml, which = ZSA_WAVE;
    for (u32 k = 0; k < nCand; k++) { u32 const l = wv_readlane(len, (int)k); if (l > best) { best = l; which = k; } }
    if (which < ZSA_WAVE) {
        *ml = best;
        *offBasePtr = (curr - (wv_readlane(myIndex, (int)which) + C.dmsDelta)) + 3;
    }
}

// ---------------------------------------------------------------- hash chain
// zstd_lazy.c:621-646 ZSTD_insertAndFindFirstIndex_internal: positions [nextToUpdate, target) enter the chain, 64 per
// step (lane k takes position b + k).  A position's predecessor is the cell's old content, or the nearest earlier
// position of the same step with the same hash; the last position per hash of a step is what the cell holds afterwards.
ZSA_DEV u32 zsa_hc_insertAndFind(ZsaLazyCtx& C, const u8* ip, u64 vip, int lane)
{
    u32 const chainMask = (1u << C.clog) - 1u;
    u32 const target = (u32)(ip - C.base);
    u32 const from = C.nextToUpdate;
    u32 const to = (C.lazySkipping && from < target) ? from + 1 : target;   // :639-641 one position only while skipping
    for (u32 b = from; b < to; b += ZSA_WAVE) {
        u32 const n = (to - b) < ZSA_WAVE ? (to - b) : ZSA_WAVE;
        bool const act = (u32)lane < n;
        u32 const pos = b + (act ? (u32)lane : 0u);
        u32 const h = zsa_hash(wv_ldg64(C.base + pos), C.hlog, C.mls);
        u32 prev = wv_ldg_cell(C.hashTable + h);
        bool last = true;
        for (u32 s2 = 0; s2 < n; s2++) {
            bool const same = wv_readlane(h, (int)s2) == h;
            if (same && s2 < (u32)lane) prev = b + s2;
            if (same && s2 > (u32)lane) last = false;
        }
        if (act) wv_stg_cell(C.chainTable + (pos & chainMask), prev);
        if (act && last) wv_stg_cell(C.hashTable + h, pos);
        wv_sync();   // the next step and the search read what this one stored
    }
    C.nextToUpdate = target;
    return wv_ldg_cell(C.hashTable + zsa_hash(vip, C.hlog, C.mls));
}
// zstd_lazy.c:655-761 ZSTD_HcFindBestMatch, noDict.  Returns the best length (3 = none), offBase in *offBasePtr.
// The chain is walked first (its course depends on indices only), 64 candidates at a time, then measured by all lanes.
template <bool DMS>
ZSA_DEV u32 zsa_hc_findBestMatch(ZsaLazyCtx& C, const u8* ip, u64 vip, const u8* iLimit, u32* offBasePtr, int lane)
{
    u32 const chainSize = 1u << C.clog, chainMask = chainSize - 1u;
    u32 const curr = (u32)(ip - C.base);
    u32 const maxDistance = 1u << C.wlog;
    u32 const lowLimit = (curr - C.lowLimit > maxDistance) ? curr - maxDistance : C.lowLimit;
    u32 const minChain = curr > chainSize ? curr - chainSize : 0;
    u32 nbAttempts = 1u << C.slog;
    u32 ml = 4 - 1;
    u32 matchIndex = zsa_hc_insertAndFind(C, ip, vip, lane);
    bool more = (matchIndex >= lowLimit) & (nbAttempts > 0);
    while (more) {
        u32 myIndex = 0, nCand = 0;
        while (more && nCand < ZSA_WAVE) {
            if ((u32)lane == nCand) myIndex = matchIndex;
            nCand++;
            if (matchIndex <= minChain) { more = false; break; }   // (leaves the loop without using up an attempt, :718)
            nbAttempts--;
            matchIndex = wv_ldg_cell(C.chainTable + (matchIndex & chainMask));
            more = (matchIndex >= lowLimit) & (nbAttempts > 0);
        }
        zsa_lazy_evaluate(C, ip, iLimit, curr, myIndex, nCand, &ml, offBasePtr, lane);
        if (ip + ml == iLimit) break;   // :715 best possible
    }
    if constexpr (DMS) {   // :726-758 the dictionary's chain, with what is left of the attempts (tables read-only, shared by all frames)
        const u32* const dmsChain = C.dms->chainTable;
        u32 const dmsChainSize = 1u << C.dms->cp.clog, dmsChainMask = dmsChainSize - 1u;
        u32 const dmsSize = 2u + C.dms->contentSize;
        u32 const dmsMinChain = dmsSize > dmsChainSize ? dmsSize - dmsChainSize : 0;
        matchIndex = wv_ldg_cell(C.dms->hashTable + zsa_hash(vip, C.dms->cp.hlog, C.mls));
        more = (matchIndex >= 2u) & (nbAttempts > 0);
        while (more) {
            u32 myIndex = 0, nCand = 0;
            while (more && nCand < ZSA_WAVE) {
                if ((u32)lane == nCand) myIndex = matchIndex;
                nCand++;
                if (matchIndex <= dmsMinChain) { more = false; break; }
                nbAttempts--;
                matchIndex = wv_ldg_cell(dmsChain + (matchIndex & dmsChainMask));
                more = (matchIndex >= 2u) & (nbAttempts > 0);
            }
            zsa_lazy_evaluate_dms(C, ip, iLimit, curr, myIndex, nCand, &ml, offBasePtr, lane);
            if (ip + ml == iLimit) break;
        }
    }
    return ml;
}

// ---------------------------------------------------------------- row hash
// zstd_lazy.c:786-791 ZSTD_row_nextIndex + :1224-1230: one position enters its row (uniform; lane 0 stores)
ZSA_DEV void zsa_row_put(ZsaLazyCtx& C, u32 relRow, u32 tag, u32 idx, u32 rowMask, u32 head, int lane)
{
    u8* const tagRow = zsa_row_tags(C, relRow);
    u32 next = (head - 1u) & rowMask;
    next += (next == 0) ? rowMask : 0;
    if (lane == 0) { wv_stg8(tagRow, next); wv_stg8(tagRow + next, tag); wv_stg_cell(zsa_row_cells(C, relRow) + next, idx); }
    if (C.cRel == relRow && ((u32)lane >> C.rowLog) < C.cN) {   // the cached copies of this row
        if (((u32)lane & rowMask) == next) { C.cTag = tag; C.cCell = idx; }
        if (((u32)lane & rowMask) == 0) C.cTag = next;
    }
}
// zstd_lazy.c:867-890 ZSTD_row_update_internalImpl for positions [from, to), 64 per step: lane k takes position
// from + k.  The reference enters them one after the other; the only coupling between two of them is the head of a row
// both fall into, which cycles through the slots rowMask, rowMask-1, .., 1.  So the r-th position (in order) of this step
// that falls into a row takes the slot r steps behind the row's head, and the last one leaves the head there; a position
// whose slot a later position of the same step takes again (rowEntries-1 or more later ones in its row: runs of equal
// bytes) is not stored at all, which is what overwriting would have left.
ZSA_DEV void zsa_row_insert_range(ZsaLazyCtx& C, u32 from, u32 to, u32 rowMask, int lane, u64 const salt = ZSA_FRESH_HASH_SALT)
{   // (salt: a frame's context has the fresh one, a dictionary's tables are built with 0, zstd_compress.c:1985-1990)
    if (from < to) C.cN = 0;
    for (u32 b = from; b < to; b += ZSA_WAVE) {
        u32 const n = (to - b) < ZSA_WAVE ? (to - b) : ZSA_WAVE;
        bool const act = (u32)lane < n;
        u32 const pos = b + (act ? (u32)lane : 0u);
        u32 const hash = zsa_hash_salted(wv_ldg64(C.base + pos), C.rowHashLog + 8, C.mls, salt);
        u32 const relRow = (hash >> 8) << C.rowLog;
        u8* const tagRow = zsa_row_tags(C, relRow);
        u32 const head0 = wv_ldg8(tagRow) & rowMask;
        u32 rank = 0, count = 0;   // positions of this step in my row: before me / in total
        for (u32 s2 = 0; s2 < n; s2++) {
            bool const same = wv_readlane(relRow, (int)s2) == relRow;
            rank += (same && s2 < (u32)lane);
            count += same;
        }
        u32 const c0 = (head0 ? head0 : 1u) - 1u;                       // head 0 (fresh row) steps to rowMask like head 1
        u32 const slot = ((c0 + 5u * rowMask - (rank + 1u)) % rowMask) + 1u;   // rank + 1 <= 64 <= 5 * rowMask
        if (act && (count - 1u - rank) < rowMask) { wv_stg8(tagRow + slot, hash & 0xFF); wv_stg_cell(zsa_row_cells(C, relRow) + slot, pos); }
        if (act && rank == count - 1u) wv_stg8(tagRow, slot);
        wv_sync();   // the next step, and the search that follows, read what this one stored
    }
}
// zstd_lazy.c:896-923 ZSTD_row_update_internal (useCache == 1): long gaps are indexed at their two ends only
ZSA_DEV void zsa_row_update(ZsaLazyCtx& C, const u8* ip, u32 rowMask, int lane)
{
    u32 idx = C.nextToUpdate;
    u32 const target = (u32)(ip - C.base);
    if (target - idx > 384) {
        zsa_row_insert_range(C, idx, idx + 96, rowMask, lane);
        idx = target - 32;
    }
    zsa_row_insert_range(C, idx, target, rowMask, lane);
    C.nextToUpdate = target;
}
// zstd_lazy.c:1120-1313 ZSTD_RowFindBestMatch, noDict
template <bool DMS>
ZSA_DEV u32 zsa_row_findBestMatch(ZsaLazyCtx& C, const u8* ip, u64 vip, const ZsaSrcWin& W, const u8* iLimit, u32* offBasePtr, int lane)
{
    u32 const curr = (u32)(ip - C.base);
    u32 const maxDistance = 1u << C.wlog;
    u32 const lowLimit = (curr - C.lowLimit > maxDistance) ? curr - maxDistance : C.lowLimit;
    u32 const rowEntries = 1u << C.rowLog, rowMask = rowEntries - 1u;
    u32 nbAttempts = 1u << (C.slog < C.rowLog ? C.slog : C.rowLog);
    u32 ml = 4 - 1;

    ZSA_PPROF(C.S, lane, 1);
    if (!C.lazySkipping) zsa_row_update(C, ip, rowMask, lane);   // :1184-1186
    else C.nextToUpdate = curr;                            // :1187-1193
    ZSA_PPROF(C.S, lane, 8);
    u32 const hash = zsa_hash_salted(vip, C.rowHashLog + 8, C.mls, ZSA_FRESH_HASH_SALT);
    u32 const relRow = (hash >> 8) << C.rowLog;
    u32 const tag = hash & 0xFF;
    // :1040-1102 the match mask and the row's cells in one round trip: lane p reads tag byte p and cell p of the row
    // (rowEntries <= 64), so "bit p set iff tagRow[p] == tag" is one ballot
    if (!(curr - C.cBase < C.cN)) {
        // miss: read this position's row and, in the same round trip, the rows of the next positions whose bytes are at
        // hand (group j of lanes = position curr + j)
        u32 const j = (u32)lane >> C.rowLog;
        size_t const d0 = (size_t)(ip - W.p);
        bool const have = W.p && d0 <= W.max8;                        // ip's bytes came from the window
        u32 const dj = (u32)d0 + j;
        bool const known = j == 0 || (have && dj <= W.max8);
        u64 const vj = j == 0 ? vip : zsa_win_at(W, known ? dj : 0u);
        u32 const hj = zsa_hash_salted(vj, C.rowHashLog + 8, C.mls, ZSA_FRESH_HASH_SALT);
        u32 const relj = known ? ((hj >> 8) << C.rowLog) : relRow;
        C.cTag = wv_ldg8(zsa_row_tags(C, relj) + ((u32)lane & rowMask));
        C.cCell = wv_ldg_cell(zsa_row_cells(C, relj) + ((u32)lane & rowMask));
        C.cRel = relj;
        C.cBase = curr;
        u32 const groups = (u32)ZSA_WAVE >> C.rowLog;
        u32 const avail = have ? (W.max8 - (u32)d0 + 1u) : 1u;
        C.cN = groups < avail ? groups : avail;
    }
    // the group of lanes that holds this position's row; "bit p set iff tagRow[p] == tag" is that group's part of one ballot
    u32 const g0 = (curr - C.cBase) << C.rowLog;
    u64 eqAll = wv_ballot(C.cTag == tag) >> g0;
    if (rowEntries < 64) eqAll &= (((u64)1 << rowEntries) - 1u);
    u64 const eq = eqAll;
    u32 const headByte = wv_readlane(C.cTag, (int)g0), head = headByte & rowMask;
    // :1205-1222: candidates from the newest (head) to the oldest; position 0 holds the head byte, not an entry; the list
    // ends before the first index below lowLimit.  Lane k keeps the k-th of them.
    u32 myIndex = 0, nCand = 0;
    {   u64 rot = head ? ((eq >> head) | (eq << (rowEntries - head))) : eq;
        if (rowEntries < 64) rot &= (((u64)1 << rowEntries) - 1u);
        for (; rot && nCand < nbAttempts; rot &= rot - 1) {
            u32 const matchPos = (head + (u32)__builtin_ctzll(rot)) & rowMask;
            if (matchPos == 0) continue;
            u32 const matchIndex = wv_readlane(C.cCell, (int)(g0 + matchPos));
            if (matchIndex < lowLimit) break;
            if ((u32)lane == nCand) myIndex = matchIndex;
            nCand++;
        }
    }
    wv_lds_fence();
    zsa_row_put(C, relRow, tag, C.nextToUpdate++, rowMask, headByte, lane);   // :1224-1230 the current position enters the row right away
    wv_lds_fence();
    ZSA_PPROF(C.S, lane, 9);
    zsa_lazy_evaluate(C, ip, iLimit, curr, myIndex, nCand, &ml, offBasePtr, lane);   // :1232-1258
    ZSA_PPROF(C.S, lane, 10);
    if constexpr (DMS) {   // :1171-1181, :1265-1311 the dictionary's row (unsalted hash, its own hash width), remaining attempts
        u32 const left = nbAttempts - nCand;
        u32 const dmsHash = zsa_hash(vip, (C.dms->cp.hlog - C.rowLog) + 8, C.mls);
        u32 const dmsRel = (dmsHash >> 8) << C.rowLog;
        const u8* const dTags = (const u8*)C.dms->hashTable + ((size_t)dmsRel << 3);
        const u32* const dCells = (const u32*)(dTags + ((size_t)1 << C.rowLog));
        u32 const p = (u32)lane & rowMask;
        u32 const dTag = wv_ldg8(dTags + p), dCell = wv_ldg_cell(dCells + p);
        u64 deq = wv_ballot(dTag == (dmsHash & 0xFF));
        if (rowEntries < 64) deq &= (((u64)1 << rowEntries) - 1u);
        u32 const dHead = wv_readlane(dTag, 0) & rowMask;
        u32 dIndex = 0, dCand = 0;
        {   u64 rot = dHead ? ((deq >> dHead) | (deq << (rowEntries - dHead))) : deq;
            if (rowEntries < 64) rot &= (((u64)1 << rowEntries) - 1u);
            for (; rot && dCand < left; rot &= rot - 1) {
                u32 const matchPos = (dHead + (u32)__builtin_ctzll(rot)) & rowMask;
                if (matchPos == 0) continue;
                u32 const matchIndex = wv_readlane(dCell, (int)matchPos);
                if (matchIndex < 2u) break;
                if ((u32)lane == dCand) dIndex = matchIndex;
                dCand++;
            }
        }
        zsa_lazy_evaluate_dms(C, ip, iLimit, curr, dIndex, dCand, &ml, offBasePtr, lane);
    }
    return ml;
}

template <bool DMS>
ZSA_DEV u32 zsa_searchMax(ZsaLazyCtx& C, const u8* ip, u64 vip, const ZsaSrcWin& W, const u8* iend, u32* offBasePtr, int lane)
{   // zstd_lazy.c:1466-1487; vip = the 8 bytes at ip, W = the source window they came from
    return C.useRow ? zsa_row_findBestMatch<DMS>(C, ip, vip, W, iend, offBasePtr, lane) : zsa_hc_findBestMatch<DMS>(C, ip, vip, iend, offBasePtr, lane);
}

// zstd_lazy.c:1565-1577 and its repeats (:1618-1632, :1654-1668, :1714-1733): the repcode candidate `offset` behind index `ip`
// with an attached dictionary -- its source may lie in the dictionary, or start there and run on into the payload.  Returns the
// match length, 0 for none.  Uniform arguments, all lanes (the count takes 64 bytes per round trip in the joint index space).
ZSA_DEV u32 zsa_lazy_dmsRep(const ZsaLazyCtx& C, u32 ip, u32 offset, u32 iend, int lane)
{
    u32 const repIndex = ip - offset;
    if ((u32)((C.V.prefix - 1u) - repIndex) < 3u) return 0;   // (intentional wrap: the 4 bytes would straddle the prefix start)
    if (zsa_dv32(C.V, repIndex) != wv_ldg32(C.base + ip)) return 0;
    return zsa_dv_count(C.V, ip + 4, repIndex + 4, iend, lane) + 4;
}
// :1691-1696 catch up in the joint index space: bytes in front of (start, match) that agree, start above anchor, match above
// the start of its segment
ZSA_DEV u32 zsa_lazy_dmsCatchup(const ZsaLazyCtx& C, u32 start, u32 matchIndex, u32 anchor, int lane)
{
    u32 const mStart = matchIndex < C.V.prefix ? C.V.dictLow : C.V.prefix;
    u32 const roomA = start - anchor, roomM = matchIndex - mStart;
    u32 const room = roomA < roomM ? roomA : roomM;
    u32 total = 0;
    while (total < room) {
        u32 const k = total + (u32)lane;
        bool const same = k < room && wv_ldg8(C.base + start - 1 - k) == zsa_dv8(C.V, matchIndex - 1 - k);
        u64 const stop = wv_ballot(!same);
        if (stop) return total + (u32)__builtin_ctzll(stop);
        total += ZSA_WAVE;
    }
    return room;
}

// ---------------------------------------------------------------- greedy parse with the row finder, 64 positions at a time
// (level 5 -- the default of the reference's Compress() -- above 16 KiB; ZSTD_greedy rows of clevels.h with searchLog <= 3)
//
// The greedy parse (zstd_lazy.c:1554-1603, depth 0) visits position after position: repcode check at ip + 1, else a search at
// ip -- which first enters every position below ip into its row, then reads ip's row, then enters ip itself -- and the first
// of the two that yields 4 bytes makes a sequence.  One chain, and every link of it waits for tables that no cache holds.
// But the table a search reads is a pure function of the positions BEFORE it: as long as no earlier position of a stretch
// falls into the row a later one reads, all searches of the stretch can be made against the table as it stood at the
// stretch's start.  So a CHUNK is 64 consecutive positions, lane k <-> position ip0 + k:
//   * every lane hashes its position, reads its row's tags (16 or 32 bytes), takes the first nbAttempts slots whose tag agrees
//     (newest first, from the row's head), reads their cells, drops what lies below the window, and compares 4 bytes of what
//     is left with its own: the candidates that survive are exactly those that can make a match (a candidate whose first 4
//     bytes differ is never longer than 3, :1242-1258) -- four memory round trips for 64 searches instead of one chain of 64;
//   * lanes that share a row with an earlier lane are found by comparing row numbers; the chunk is valid below the first;
//   * the events are then picked on lane masks in the reference's order (repcode at ip + 1 before the search at ip; the
//     immediate repcode after every sequence, :1735-1745): M1 / M2 hold the lanes whose 4 bytes equal the 4 bytes offset_1 /
//     offset_2 behind them, H the lanes whose search finds a match.  Only the winning search is measured (one candidate per
//     lane, first maximum: zsa_lazy_evaluate), caught up backwards, and stored;
//   * nothing enters the table while a chunk is scanned; when it is left, the positions that the reference has entered by
//     then -- everything up to the last search made -- are entered by zsa_row_insert_range, 64 per round trip.
// What the chunk does not handle is left to the serial loop below: searches that start more than 320 positions behind the
// last table update (the reference then enters only the two ends of the gap, :896-923), literal runs of 192 bytes and more
// (the step grows and searches start to skip, :1591-1603), the last 104 bytes of a block, stretches where neighbouring
// positions share rows (runs of equal bytes), rows of 64 entries and searches of more than 8 attempts.
struct ZsaGreedyState {   // what the chunk parser takes over from the serial loop and hands back (a copy: the loop's own state stays in registers)
    u32 ip, anchor, off1, off2, nbSeq, hold, nextToUpdate;
    ZsaEncShared* S; u32* hashTable; const u8* base;
    u32 hlog, wlog, slog, mls, rowLog, rowHashLog, lowLimit;
};

// bit i set iff byte i of w equals the byte that `rep` repeats four times
ZSA_DEV u32 zsa_eq4(u32 w, u32 rep)
{
    u32 const x = w ^ rep;
    u32 t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    t = ~(t | x | 0x7F7F7F7Fu);                          // 0x80 in every byte of x that is zero
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}
// zstd_lazy.c:1685-1698: bytes in front of the match that agree as well (start > anchor, match > prefixLowest)
ZSA_DEV u32 zsa_lazy_catchup(const u8* start, const u8* m, const u8* anchor, const u8* prefixLowest)
{
    u32 const roomA = (u32)(start - anchor), roomM = (u32)(m - prefixLowest);
    u32 const maxBack = roomA < roomM ? roomA : roomM;
    u32 back = 0;
    if (maxBack > 0 && roomM >= 8) {   // 8 bytes before both in one round trip
        u64 const d = wv_ldg64(start - 8) ^ wv_ldg64(m - 8);
        back = d ? ((u32)__builtin_clzll(d) >> 3) : 8;
        if (back > maxBack) back = maxBack;
        if (back == 8) while (back < maxBack && wv_ldg8(start - 1 - back) == wv_ldg8(m - 1 - back)) back++;
    } else {
        while (back < maxBack && wv_ldg8(start - 1 - back) == wv_ldg8(m - 1 - back)) back++;
    }
    return back;
}

// index of the lowest set bit, 0xFFFFFFFF for zero
ZSA_DEV u32 zsa_first_bit64(u64 x) { return x ? (u32)__builtin_ctzll(x) : 0xFFFFFFFFu; }
#define ZSA_GREEDY_CAND 8
#ifndef ZSA_GREEDY_W0
#define ZSA_GREEDY_W0 32u
#endif
ZSA_DEVFN void zsa_greedy_row_chunk(ZsaGreedyState* stp, u64* seqOutIn, u32 iendIdx, u32 prefixLowestIdx)
{
    int const lane = wv_lane();
    ZsaLazyCtx C;
    C.S = (ZsaEncShared*)wv_uniform64((u64)stp->S);
    C.hashTable = (u32*)wv_uniform64((u64)stp->hashTable); C.chainTable = nullptr;
    C.base = (const u8*)wv_uniform64((u64)stp->base);
    C.hlog = wv_uniform(stp->hlog); C.clog = 0; C.wlog = wv_uniform(stp->wlog); C.slog = wv_uniform(stp->slog); C.mls = wv_uniform(stp->mls);
    C.rowLog = wv_uniform(stp->rowLog); C.rowHashLog = wv_uniform(stp->rowHashLog); C.lowLimit = wv_uniform(stp->lowLimit);
    C.nextToUpdate = wv_uniform(stp->nextToUpdate); C.lazySkipping = 0; C.useRow = true;
    C.cBase = 0; C.cN = 0; C.cTag = 0; C.cCell = 0; C.cRel = 0;
    u64* const seqOut = (u64*)wv_uniform64((u64)seqOutIn);
    iendIdx = wv_uniform(iendIdx); prefixLowestIdx = wv_uniform(prefixLowestIdx);
    u32 ip = wv_uniform(stp->ip), anchor = wv_uniform(stp->anchor), off1 = wv_uniform(stp->off1), off2 = wv_uniform(stp->off2);
    u32 nbSeq = wv_uniform(stp->nbSeq), hold = wv_uniform(stp->hold);
    const u8* const base = C.base;
    u32 const rowLog = C.rowLog, rowEntries = 1u << rowLog, rowMask = rowEntries - 1u;
    u32 const entriesMask = rowEntries < 32 ? (1u << rowEntries) - 1u : 0xFFFFFFFFu;
    u32 const nbAttempts = 1u << (C.slog < rowLog ? C.slog : rowLog);
    u32 const maxDistance = 1u << C.wlog;
    const u8* const iend = base + iendIdx;
    // A chunk searches as many positions as the search is likely to last: ZSA_GREEDY_W0 behind a sequence, twice as many every
    // time nothing is found (every searched position costs three scattered table lines, and the tables of the frames in flight
    // are far larger than any cache: 64-wide chunks spent four fifths of that traffic on positions that a match then skipped)
    u32 width = ZSA_GREEDY_W0;

    for (;;) {   // one chunk per iteration
        if (ip < hold || ip - C.nextToUpdate > 320 || ip - anchor >= 192 || ip + 64 + 40 > iendIdx) break;
        u32 const ip0 = ip;
        ZSA_STAT(0);
        ZSA_PPROF(C.S, lane, 1);
        u32 const q = ip0 + (u32)lane;
        const u8* const pq = base + q;
        u64 const v = wv_ldg64(pq);
        u32 const r1 = wv_ldg32(pq - off1), r2 = wv_ldg32(pq - off2);        // (an offset of 0 reads the lane's own bytes: its mask is dropped below)
        u64 M1 = off1 ? wv_ballot(r1 == (u32)v) : 0, M2 = off2 ? wv_ballot(r2 == (u32)v) : 0;
        if ((M1 >> 1) & 1ull) {
            // The repcode at ip0 + 1 hits: no search is made at ip0, so nothing enters the table here -- and what enters it at the
            // next search depends on how far away that one is (:896-923).  The sequence is stored and the chunk starts over behind it.
            u32 const start = ip0 + 1;
            u32 const mlen = zsa_lz_count32(base + start + 4, base + start + 4 - off1, iend) + 4;
            if (lane == 0) seqOut[nbSeq] = zsa_packseq(start - anchor, mlen - ZSA_MINMATCH, 1);
            ZSA_STAT(3);
            nbSeq++;
            ip = anchor = start + mlen;
            for (;;) {   // :1735-1745 immediate repcode
                if (!(((s32)(iendIdx - ip) >= 16) && off2 > 0)) break;
                u32 const o2 = ip - ip0;
                bool const hit = o2 < 64 ? ((M2 >> o2) & 1ull) != 0 : wv_ldg32(base + ip) == wv_ldg32(base + ip - off2);
                if (!hit) break;
                u32 const ml2 = zsa_lz_count32(base + ip + 4, base + ip + 4 - off2, iend) + 4;
                { u32 const t = off2; off2 = off1; off1 = t; }
                { u64 const t = M2; M2 = M1; M1 = t; }
                if (lane == 0) seqOut[nbSeq] = zsa_packseq(0, ml2 - ZSA_MINMATCH, 1);
                nbSeq++;
                ip += ml2; anchor = ip;
            }
            width = ZSA_GREEDY_W0;
            continue;
        }
        zsa_row_insert_range(C, C.nextToUpdate, ip0, rowMask, lane);   // what the search at ip0 enters first (:1184-1186; no more than 320 positions: all of them)
        C.nextToUpdate = ip0;
        ZSA_PPROF(C.S, lane, 8);
        // ---- set-up: 64 searches against the table as it stands
        u32 const hash = zsa_hash_salted(v, C.rowHashLog + 8, C.mls, ZSA_FRESH_HASH_SALT);
        u32 const relRow = (hash >> 8) << rowLog;
        u32 const tagRep = (hash & 0xFFu) * 0x01010101u;
        const u8* const tagRow = zsa_row_tags(C, relRow);
        const u32* const cellRow = zsa_row_cells(C, relRow);
        bool const searched = (u32)lane < width;
        u64 ta = 0, tb = 0, tc = 0, td = 0;
        if (searched) {
            ta = wv_ldg64(tagRow); tb = wv_ldg64(tagRow + 8);
            if (rowLog == 5) { tc = wv_ldg64(tagRow + 16); td = wv_ldg64(tagRow + 24); }   // (uniform)
        }
        u32 eq = zsa_eq4((u32)ta, tagRep) | (zsa_eq4((u32)(ta >> 32), tagRep) << 4) | (zsa_eq4((u32)tb, tagRep) << 8) | (zsa_eq4((u32)(tb >> 32), tagRep) << 12);
        if (rowLog == 5) eq |= (zsa_eq4((u32)tc, tagRep) << 16) | (zsa_eq4((u32)(tc >> 32), tagRep) << 20) | (zsa_eq4((u32)td, tagRep) << 24) | (zsa_eq4((u32)(td >> 32), tagRep) << 28);
        u32 const head = (u32)ta & 0xFFu & rowMask;
        eq &= ~1u;                                                           // slot 0 holds the head, never an entry (:1210)
        if (!searched) eq = 0;
        u32 rot = head ? (((eq >> head) | (eq << (rowEntries - head))) & entriesMask) : eq;   // newest entry first (:1205-1208)
        u32 cell[ZSA_GREEDY_CAND];
        u32 have = 0;
#pragma unroll
        for (u32 i = 0; i < ZSA_GREEDY_CAND; i++) cell[i] = 0;
#pragma unroll
        for (u32 i = 0; i < ZSA_GREEDY_CAND; i++) {
            bool const h = rot != 0 && i < nbAttempts;
            if (!wv_ballot(h)) break;                                        // (uniform: most rows hold one or two entries of a tag)
            u32 const pos = (head + (h ? (u32)__builtin_ctz(rot) : 0u)) & rowMask;
            if (h) { cell[i] = wv_ldg_cell(cellRow + pos); have |= 1u << i; rot &= rot - 1u; }
        }
        // the list ends before the first index below the window (:1213); then the 4 bytes
        u32 const lowLimit = (q - C.lowLimit > maxDistance) ? q - maxDistance : C.lowLimit;
        u32 alive = 0;
        {   bool on = true;
#pragma unroll
            for (u32 i = 0; i < ZSA_GREEDY_CAND; i++) { on = on && ((have >> i) & 1u) && cell[i] >= lowLimit; if (on) alive |= 1u << i; }
        }
        u32 first4[ZSA_GREEDY_CAND];
#pragma unroll
        for (u32 i = 0; i < ZSA_GREEDY_CAND; i++) first4[i] = 0;
#pragma unroll
        for (u32 i = 0; i < ZSA_GREEDY_CAND; i++) {
            bool const a = ((alive >> i) & 1u) != 0;
            if (!wv_ballot(a)) break;
            if (a) first4[i] = wv_ldg32(base + cell[i]);
        }
        u32 pm = 0;                                                          // candidates that can make a match
#pragma unroll
        for (u32 i = 0; i < ZSA_GREEDY_CAND; i++) if (((alive >> i) & 1u) && first4[i] == (u32)v) pm |= 1u << i;
        // lanes whose row another lane of the chunk falls into as well, by a bitmap pass over the row numbers (as zsa_parse.h does
        // for cells: the first lane to reach a row's bit finds it clear, the others find it set and clear it again, which tells the
        // first one).  Below the second-lowest of them no two lanes share a row.  (row numbers modulo 8192: a false share only
        // shortens the chunk; the bitmap is all zero between uses)
        u32 winValid;
        {   u32 const rowNo = hash >> 8;
            u32* const dw = &C.S->dupBits[(rowNo >> 5) & 255u];
            u32 const bit = 1u << (rowNo & 31u);
            u32 const dprev = searched ? wv_atomic_or_lds(dw, bit) : 0u;
            bool const late = (dprev & bit) != 0;
            wv_lds_fence();
            if (searched) (void)wv_atomic_and_lds(dw, late ? ~bit : 0xFFFFFFFFu);
            wv_lds_fence();
            bool const shares = searched && (late | ((*dw & bit) == 0));
            wv_lds_fence();
            if (searched) *dw = 0;
            wv_lds_fence();
            u64 const G = wv_ballot(shares);
            u64 const G2 = G & (G - 1);
            winValid = G2 ? (u32)__builtin_ctzll(G2) : 64u;
        }
        if (winValid < 4) hold = ip0 + 48;                                   // runs of equal bytes: the serial loop takes the next stretch
        u32 lim = winValid < 63 ? winValid : 63u;                            // positions [ip0, ip0 + lim) are decided here
        if (lim > width) lim = width;
        u64 const H = wv_ballot(pm != 0) & (((u64)1 << lim) - 1u);
        u32 ntu = ip0;                                                       // first position the reference has not entered yet
        ZSA_PPROF(C.S, lane, 9);
        // ---- events
        for (;;) {
            u32 const o = ip - ip0;                                          // < lim
            u32 fr = zsa_first_bit64((M1 >> o) >> 1);                              // repcode at (ip + fr) + 1
            u32 const fh = zsa_first_bit64(H >> o);                                // search at ip + fh finds a match
            if (fr >= lim - o) fr = 0xFFFFFFFFu;
            if ((fr & fh) == 0xFFFFFFFFu) { ip = ip0 + lim; ntu = ip; width = width < 32 ? width * 2 : 64u; break; }   // nothing: every position was searched in vain
            u32 start, mlen, offBase;
            if (fr <= fh) {
                u32 const p = ip + fr;
                if (fr) ntu = p;                                             // (positions ip .. p - 1 were searched)
                start = p + 1; offBase = 1;
                mlen = zsa_lz_count32(base + start + 4, base + start + 4 - off1, iend) + 4;
            } else {
                u32 const cur = ip + fh, L = cur - ip0;
                ntu = cur + 1;
                u32 const pmL = wv_readlane(pm, (int)L);
                u32 myIndex = 0, nCand = 0;
#pragma unroll
                for (u32 i = 0; i < ZSA_GREEDY_CAND; i++) {
                    u32 const ci = wv_readlane(cell[i], (int)L);
                    if ((pmL >> i) & 1u) { if ((u32)lane == nCand) myIndex = ci; nCand++; }
                }
                ZSA_PPROF(C.S, lane, 10);
                // one candidate per lane, each measured in full; the first of the longest wins (zsa_lazy_evaluate).  With a single
                // candidate the offset is known beforehand, and what follows the measurement -- the new offset's mask, the bytes in
                // front of the match -- is requested in the same round trip
                u32 const cand0 = wv_readlane(myIndex, 0);
                bool const single = nCand == 1;
                u32 rn = 0; u64 fa = 0, fb = 0;
                u32 const roomM0 = cand0 - prefixLowestIdx;
                bool const front = single && cur > anchor && roomM0 >= 8;
                if (single) rn = wv_ldg32((u32)lane >= L ? pq - (cur - cand0) : pq);
                if (front) { fa = wv_ldg64(base + cur - 8); fb = wv_ldg64(base + cand0 - 8); }
                u32 const len = (u32)lane < nCand ? zsa_lz_count32(base + cur, base + myIndex, iend) : 0u;
                u32 ml = 3, which = 0;
                for (u32 k = 0; k < nCand; k++) { u32 const l = wv_readlane(len, (int)k); if (l > ml) { ml = l; which = k; } }   // first maximum
                u32 const off = cur - wv_readlane(myIndex, (int)which), ofb = off + 3;
                ZSA_PPROF(C.S, lane, 11);
                if (!single) rn = wv_ldg32((u32)lane >= L ? pq - off : pq);  // (lanes behind the match are not looked at again; the offset may reach below the frame from there)
                u32 back;
                if (front) {   // zstd_lazy.c:1685-1698 with the 8 bytes in front of both already here
                    u32 const roomA = cur - anchor, maxBack = roomA < roomM0 ? roomA : roomM0;
                    u64 const d = fa ^ fb;
                    back = d ? ((u32)__builtin_clzll(d) >> 3) : 8;
                    if (back > maxBack) back = maxBack;
                    if (back == 8) while (back < maxBack && wv_ldg8(base + cur - 1 - back) == wv_ldg8(base + cand0 - 1 - back)) back++;
                } else back = zsa_lazy_catchup(base + cur, base + cur - off, base + anchor, base + prefixLowestIdx);
                start = cur - back; mlen = ml + back; offBase = ofb;
                off2 = off1; M2 = M1; off1 = off; M1 = wv_ballot(rn == (u32)v);
            }
            if (lane == 0) seqOut[nbSeq] = zsa_packseq(start - anchor, mlen - ZSA_MINMATCH, offBase);
            ZSA_STAT(3);
            nbSeq++;
            ip = anchor = start + mlen;
            width = ZSA_GREEDY_W0;
            // :1735-1745 immediate repcode
            for (;;) {
                if (!(((s32)(iendIdx - ip) >= 16) && off2 > 0)) break;
                u32 const o2 = ip - ip0;
                bool const hit = o2 < 64 ? ((M2 >> o2) & 1ull) != 0 : wv_ldg32(base + ip) == wv_ldg32(base + ip - off2);
                if (!hit) break;
                u32 const ml2 = zsa_lz_count32(base + ip + 4, base + ip + 4 - off2, iend) + 4;
                { u32 const t = off2; off2 = off1; off1 = t; }
                { u64 const t = M2; M2 = M1; M1 = t; }
                if (lane == 0) seqOut[nbSeq] = zsa_packseq(0, ml2 - ZSA_MINMATCH, 1);
                nbSeq++;
                ip += ml2; anchor = ip;
            }
            if (ip - ip0 >= lim) break;
        }
        ZSA_PPROF(C.S, lane, 10);
        // ---- the positions the reference has entered by now: [ip0, ntu), all below winValid, so each in a row of its own that
        // nothing has touched since the set-up read its head (zstd_lazy.c:786-791 ZSTD_row_nextIndex + :1224-1230)
        wv_lds_fence();
        if ((u32)lane < ntu - ip0) {
            u32 next = (head - 1u) & rowMask;
            next += (next == 0) ? rowMask : 0;
            wv_stg8((u8*)tagRow, next); wv_stg8((u8*)tagRow + next, hash & 0xFFu); wv_stg_cell((u32*)cellRow + next, q);
        }
        wv_sync();   // the next chunk, and the serial loop, read what this one stored
        C.nextToUpdate = ntu;
        ZSA_PPROF(C.S, lane, 7);
    }
    stp->ip = ip; stp->anchor = anchor; stp->off1 = off1; stp->off2 = off2; stp->nbSeq = nbSeq; stp->hold = hold;   // (per-lane copies of uniform values)
    stp->nextToUpdate = C.nextToUpdate;
}

// depth 0 = greedy, 1 = lazy, 2 = lazy2 (zstd_lazy.c:1493-1757).  Appends packed sequences to seqOut, returns their
// number; rep[] and *lastLL are updated.  Called by all lanes with uniform arguments; everything it computes is uniform.
template <bool DMS>
ZSA_DEV u32 zsa_parse_lazy(ZsaLazyCtx& C, u32 dictLimit, const u8* src, u32 srcSize, u32 depth, u32 rep[3], u64* seqOut, u32* lastLL, int lane)
{
    const u8* const istart = src;
    const u8* ip = istart;
    const u8* anchor = istart;
    const u8* const iend = istart + srcSize;
    s32 const margin = C.useRow ? 8 + 8 : 8;             // :1505 (the row finder's hash cache looks 8 positions ahead)
    const u8* const prefixLowest = C.base + dictLimit;
    u32 offset_1 = rep[0], offset_2 = rep[1], saved1 = 0, saved2 = 0, nbSeq = 0;
#define ZSA_BEFORE_LIMIT(p) ((s32)(iend - (p)) > margin)      /* p <  ilimit */
#define ZSA_UPTO_LIMIT(p)   ((s32)(iend - (p)) >= margin)     /* p <= ilimit */
#define ZSA_HB(x) (31u - (u32)__builtin_clz((u32)(x)))

    if constexpr (!DMS) ip += (ip == prefixLowest);   // (:1526-1529; an attached dictionary is never empty)
    if constexpr (!DMS) {   u32 const curr = (u32)(ip - C.base);
        u32 const maxDistance = 1u << C.wlog;
        u32 const windowLow = (curr - dictLimit > maxDistance) ? curr - maxDistance : dictLimit;   // ZSTD_getLowestPrefixIndex
        u32 const maxRep = curr - windowLow;
        if (offset_2 > maxRep) { saved2 = offset_2; offset_2 = 0; }
        if (offset_1 > maxRep) { saved1 = offset_1; offset_1 = 0; }
    }
    C.lazySkipping = 0;
    C.cN = 0; C.cBase = 0;
    ZsaSrcWin W, R;   // source bytes around ip, and around ip - offset_1
    W.p = R.p = nullptr; W.max8 = W.max4 = R.max8 = R.max4 = 0; W.lo = W.hi = R.lo = R.hi = 0;

    // greedy + row finder: whole chunks of positions where the conditions hold (zsa_greedy_row_chunk), single steps elsewhere
#ifdef ZSA_NO_GREEDY_CHUNK
    bool const chunks = false;
#else
    bool const chunks = !DMS && depth == 0 && C.useRow && C.rowLog <= 5 && (C.slog < C.rowLog ? C.slog : C.rowLog) <= 3;
#endif
    u32 const iendIdx = (u32)(iend - C.base);
    u32 hold = 0;
    while (ZSA_BEFORE_LIMIT(ip)) {
        if (chunks && !C.lazySkipping) {
            u32 const ipIdx = (u32)(ip - C.base);
            if (ipIdx >= hold && ipIdx - C.nextToUpdate <= 320 && (u32)(ip - anchor) < 192 && ipIdx + 64 + 40 <= (u32)(iend - C.base)) {
                ZsaGreedyState gs;
                gs.ip = ipIdx; gs.anchor = (u32)(anchor - C.base); gs.off1 = offset_1; gs.off2 = offset_2; gs.nbSeq = nbSeq; gs.hold = hold;
                gs.nextToUpdate = C.nextToUpdate; gs.S = C.S; gs.hashTable = C.hashTable; gs.base = C.base;
                gs.hlog = C.hlog; gs.wlog = C.wlog; gs.slog = C.slog; gs.mls = C.mls; gs.rowLog = C.rowLog; gs.rowHashLog = C.rowHashLog; gs.lowLimit = C.lowLimit;
                zsa_greedy_row_chunk(&gs, seqOut, (u32)(iend - C.base), dictLimit);
                bool const moved = wv_uniform(gs.ip) != ipIdx;
                ip = C.base + wv_uniform(gs.ip); anchor = C.base + wv_uniform(gs.anchor);
                offset_1 = wv_uniform(gs.off1); offset_2 = wv_uniform(gs.off2); nbSeq = wv_uniform(gs.nbSeq); hold = wv_uniform(gs.hold);
                C.nextToUpdate = wv_uniform(gs.nextToUpdate);
                C.cN = 0; W.p = nullptr; R.p = nullptr;
                if (moved) continue;
            }
        }
        u32 matchLength = 0;
        u32 offBase = 1;   // REPCODE1_TO_OFFBASE
        const u8* start = ip + 1;
        bool storeNow = false;

        // the bytes at ip serve the repcode check at ip+1 and the search at ip; both loads share a round trip
        u64 vip = zsa_win_get64(W, ip, iend);
        // :1565-1582 repcode at ip+1
        if constexpr (DMS) {
            u32 const mlRep = zsa_lazy_dmsRep(C, (u32)(ip - C.base) + 1, offset_1, iendIdx, lane);
            if (mlRep) { matchLength = mlRep; if (depth == 0) storeNow = true; }
        } else
        if ((offset_1 > 0) && (zsa_win_get32(R, ip + 1 - offset_1, iend) == (u32)(vip >> 8))) {
            matchLength = zsa_lz_count(ip + 1 + 4, ip + 1 + 4 - offset_1, iend) + 4;
            if (depth == 0) storeNow = true;
        }
        if (!storeNow) {
            {   u32 offbaseFound = 999999999;   // :1585-1589
                u32 const ml2 = zsa_searchMax<DMS>(C, ip, vip, W, iend, &offbaseFound, lane);
                if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = offbaseFound; }
            }
            if (matchLength < 4) {   // :1591-1603
                u32 const step = ((u32)(ip - anchor) >> 8) + 1;   // kSearchStrength
                ip += step;
                C.lazySkipping = step > 8;                        // kLazySkippingStep
                continue;
            }
            if (depth >= 1)
            while (ZSA_BEFORE_LIMIT(ip)) {   // :1606-1678
                ip++;
                vip = zsa_win_get64(W, ip, iend);
                if constexpr (DMS) {   // :1618-1632
                    u32 const mlRep = zsa_lazy_dmsRep(C, (u32)(ip - C.base), offset_1, iendIdx, lane);
                    s32 const gain2 = (s32)(mlRep * 3);
                    s32 const gain1 = (s32)(matchLength * 3 - ZSA_HB(offBase) + 1);
                    if ((mlRep >= 4) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1; start = ip; }
                } else
                if ((offBase) && ((offset_1 > 0) && ((u32)vip == zsa_win_get32(R, ip - offset_1, iend)))) {
                    u32 const mlRep = zsa_lz_count(ip + 4, ip + 4 - offset_1, iend) + 4;
                    s32 const gain2 = (s32)(mlRep * 3);
                    s32 const gain1 = (s32)(matchLength * 3 - ZSA_HB(offBase) + 1);
                    if ((mlRep >= 4) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1; start = ip; }
                }
                {   u32 ofbCandidate = 999999999;
                    u32 const ml2 = zsa_searchMax<DMS>(C, ip, vip, W, iend, &ofbCandidate, lane);
                    s32 const gain2 = (s32)(ml2 * 4 - ZSA_HB(ofbCandidate));
                    s32 const gain1 = (s32)(matchLength * 4 - ZSA_HB(offBase) + 4);
                    if ((ml2 >= 4) && (gain2 > gain1)) { matchLength = ml2; offBase = ofbCandidate; start = ip; continue; }
                }
                if ((depth == 2) && ZSA_BEFORE_LIMIT(ip)) {
                    ip++;
                    vip = zsa_win_get64(W, ip, iend);
                    if constexpr (DMS) {   // :1654-1668
                        u32 const mlRep = zsa_lazy_dmsRep(C, (u32)(ip - C.base), offset_1, iendIdx, lane);
                        s32 const gain2 = (s32)(mlRep * 4);
                        s32 const gain1 = (s32)(matchLength * 4 - ZSA_HB(offBase) + 1);
                        if ((mlRep >= 4) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1; start = ip; }
                    } else
                    if ((offBase) && ((offset_1 > 0) && ((u32)vip == zsa_win_get32(R, ip - offset_1, iend)))) {
                        u32 const mlRep = zsa_lz_count(ip + 4, ip + 4 - offset_1, iend) + 4;
                        s32 const gain2 = (s32)(mlRep * 4);
                        s32 const gain1 = (s32)(matchLength * 4 - ZSA_HB(offBase) + 1);
                        if ((mlRep >= 4) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1; start = ip; }
                    }
                    {   u32 ofbCandidate = 999999999;
                        u32 const ml2 = zsa_searchMax<DMS>(C, ip, vip, W, iend, &ofbCandidate, lane);
                        s32 const gain2 = (s32)(ml2 * 4 - ZSA_HB(ofbCandidate));
                        s32 const gain1 = (s32)(matchLength * 4 - ZSA_HB(offBase) + 7);
                        if ((ml2 >= 4) && (gain2 > gain1)) { matchLength = ml2; offBase = ofbCandidate; start = ip; continue; }
                    }
                }
                break;
            }
            if (offBase > 3) {   // :1685-1698 catch up
                u32 const off = offBase - 3;
                u32 back;
                if constexpr (DMS) back = zsa_lazy_dmsCatchup(C, (u32)(start - C.base), (u32)(start - C.base) - off, (u32)(anchor - C.base), lane);
                else back = zsa_lazy_catchup(start, start - off, anchor, prefixLowest);
                start -= back; matchLength += back;
                offset_2 = offset_1; offset_1 = off;
            }
        }
        // :1700-1704 _storeSequence
        ZSA_STAT(5);
        if (lane == 0) seqOut[nbSeq] = zsa_packseq((u32)(start - anchor), matchLength - ZSA_MINMATCH, offBase);
        nbSeq++;
        anchor = ip = start + matchLength;
        C.lazySkipping = 0;   // :1705-1711
        // :1714-1745 immediate repcode
        if constexpr (DMS) {
            while (ZSA_UPTO_LIMIT(ip)) {
                u32 const mlRep = zsa_lazy_dmsRep(C, (u32)(ip - C.base), offset_2, iendIdx, lane);
                if (!mlRep) break;
                { u32 const t = offset_2; offset_2 = offset_1; offset_1 = t; }
                if (lane == 0) seqOut[nbSeq] = zsa_packseq(0, mlRep - ZSA_MINMATCH, 1);
                nbSeq++;
                ip += mlRep;
                anchor = ip;
            }
        } else
        while ((ZSA_UPTO_LIMIT(ip) & (offset_2 > 0)) && (wv_ldg32(ip) == wv_ldg32(ip - offset_2))) {
            matchLength = zsa_lz_count(ip + 4, ip + 4 - offset_2, iend) + 4;
            { u32 const t = offset_2; offset_2 = offset_1; offset_1 = t; }
            if (lane == 0) seqOut[nbSeq] = zsa_packseq(0, matchLength - ZSA_MINMATCH, 1);
            nbSeq++;
            ip += matchLength;
            anchor = ip;
        }
    }
#undef ZSA_BEFORE_LIMIT
#undef ZSA_UPTO_LIMIT
#undef ZSA_HB
    saved2 = ((saved1 != 0) && (offset_1 != 0)) ? saved1 : saved2;   // :1747-1753
    rep[0] = offset_1 ? offset_1 : saved1;
    rep[1] = offset_2 ? offset_2 : saved2;
    *lastLL = (u32)(iend - anchor);
    return nbSeq;
}
