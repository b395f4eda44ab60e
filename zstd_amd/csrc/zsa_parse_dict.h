// zsa_parse_dict.h -- the ZSTD_fast parse against an ATTACHED dictionary (zstd_fast.c:459-656: payloads of up to 8 KiB with a
// CDict, the BulkProcessor path), executed by all 64 lanes of the wave.
//
// The reference visits positions one by one (step 1, growing by one every 256 bytes without a match) and per position p
// performs, in this order: insert p into the local table; a repcode check at p + 1; a check against the dictionary's
// tagged table (only taken when the local table has nothing valid for p); a check against the local table.  Here a WINDOW
// is up to 64 consecutive positions, lane k <-> position ip0 + k:
//   * every lane hashes its position twice (local table, dictionary table + 8 tag bits), reads both cells and makes its
//     three checks; lane order == position order and rep > dictionary > local inside a lane, so the first lane of one
//     ballot with any hit is the event the reference stops at;
//   * a lane's local probe is exact as long as no earlier lane of the window falls into the same cell; the lanes that share
//     a cell are found with a bitmap pass over the cell numbers and the window is cut below the second-lowest of them;
//   * dictionary and payload form one index space (dictionary bytes sit right below prefixStartIndex), so a match that
//     starts in the dictionary and runs on into the payload -- ZSTD_count_2segments -- is one forward count, 64 bytes per step;
//   * positions up to the event enter the local table in one store (distinct cells), then current + 2 and ip0 - 2.
// Included by zsa_encode.h (uses zsa_hash, zsa_packseq, ZsaEncParams, ZsaCDictDev, ZsaWindow from there).
#pragma once

#ifndef ZSA_DMS_W0
#define ZSA_DMS_W0 16u     // positions in the first window of a search (attached-dictionary parse)
#endif

struct ZsaDmsView {       // one index space over dictionary + payload
    const u8* base;       // index i >= prefix <-> base + i
    const u8* dbase;      // index i <  prefix <-> dbase + i   (dictionary content, shifted by dictIndexDelta)
    u32 prefix;           // prefixStartIndex
    u32 dictLow;          // lowest dictionary index (dictStartIndex + dictIndexDelta)
};
ZSA_DEV u32 zsa_dv8(const ZsaDmsView& V, u32 i) { return i >= V.prefix ? wv_ldg8(V.base + i) : wv_ldg8(V.dbase + i); }
// 4 bytes at index i; the callers guarantee that [i, i + 4) does not straddle the prefix start
ZSA_DEV u32 zsa_dv32(const ZsaDmsView& V, u32 i) { return i >= V.prefix ? wv_ldg32(V.base + i) : wv_ldg32(V.dbase + i); }

// common length of [ip, iend) and [m, ...) in the joint index space, m < ip (ZSTD_count / ZSTD_count_2segments,
// zstd_compress_internal.h:755-793); uniform arguments, all lanes
ZSA_DEV u32 zsa_dv_count(const ZsaDmsView& V, u32 ip, u32 m, u32 iend, int lane)
{
    u32 total = 0;
    for (;;) {
        u32 const a = ip + total + (u32)lane;
        bool const same = a < iend && wv_ldg8(V.base + a) == zsa_dv8(V, m + total + (u32)lane);
        u64 const stop = wv_ballot(!same);
        if (stop) return total + (u32)__builtin_ctzll(stop);
        total += ZSA_WAVE;
    }
}
// a table match measured in both directions -- forwards from (ip + 4, m + 4), backwards from (ip - 1, m - 1), where ip stays
// above `anchor` and m above `low` (zstd_fast.c:569-573, :583-588) -- the first 64 bytes of each direction in one round trip
ZSA_DEV void zsa_dv_extend(const ZsaDmsView& V, u32 ip, u32 m, u32 iend, u32 anchor, u32 low, u32* fwd, u32* bwd, int lane)
{
    u32 const room = (ip - anchor) < (m - low) ? (ip - anchor) : (m - low);
    u32 const a = ip + 4 + (u32)lane;
    bool const fin = a < iend, bin = (u32)lane < room;
    u32 const f0 = wv_ldg8(V.base + (fin ? a : ip)), f1 = zsa_dv8(V, fin ? m + 4 + (u32)lane : m);
    u32 const b0 = wv_ldg8(V.base + (bin ? ip - 1 - (u32)lane : ip)), b1 = zsa_dv8(V, bin ? m - 1 - (u32)lane : m);
    u64 const fstop = wv_ballot(!(fin && f0 == f1)), bstop = wv_ballot(!(bin && b0 == b1));
    *fwd = fstop ? (u32)__builtin_ctzll(fstop) : ZSA_WAVE + zsa_dv_count(V, ip + 4 + ZSA_WAVE, m + 4 + ZSA_WAVE, iend, lane);
    if (bstop) *bwd = (u32)__builtin_ctzll(bstop);              // (a lane at or beyond `room` counts as a mismatch)
    else {   // more than 64 bytes backwards: rare
        u32 total = ZSA_WAVE;
        while (total < room) {
            u32 const k = total + (u32)lane;
            bool const same = k < room && wv_ldg8(V.base + ip - 1 - k) == zsa_dv8(V, m - 1 - k);
            u64 const stop = wv_ballot(!same);
            if (stop) { total += (u32)__builtin_ctzll(stop); break; }
            total += ZSA_WAVE;
        }
        *bwd = total;
    }
}

// Parses one block with the dictionary `Wn.dms` attached.  hashTable: the frame's local table (global scratch, zeroed).
// dup: LDS words, all zero, at least 256 (a bitmap over cell numbers modulo 8192).  Returns the number of sequences
// appended to seqOut; rep[] and *lastLL are updated.  All values are wave-uniform.  (A function of its own, so that its loop
// gets its own register allocation: inlined into the frame encoder it ran out of spilled registers.)
ZSA_DEVFN u32 zsa_parse_dms_wave(u32* hashTable, u32 hlog, u32 mls, u32 tlen, const u8* wbase, u32 wDictLimit, const ZsaCDictDev* dms,
                                 const u8* src, u32 srcSize, u32* rep /* LDS: [0], [1] in and out, [2] receives the last literal run */,
                                 u64* seqOut, u32* dup)
{
    int const lane = wv_lane();
    hlog = wv_uniform(hlog); mls = wv_uniform(mls); srcSize = wv_uniform(srcSize);
    u32 const stepSize = wv_uniform(tlen + !tlen);              // zstd_fast.c:470
    struct { const u8* base; u32 dictLimit; } Wn;
    Wn.base = (const u8*)wv_uniform64((u64)wbase); Wn.dictLimit = wv_uniform(wDictLimit);
    dms = (const ZsaCDictDev*)wv_uniform64((u64)dms);
    seqOut = (u64*)wv_uniform64((u64)seqOut); hashTable = (u32*)wv_uniform64((u64)hashTable);
    src = (const u8*)wv_uniform64((u64)src);
    const u32* const dictTable = dms->hashTable;
    u32 const dictHBits = dms->cp.hlog + 8;                     // 8 tag bits (:21)
    u32 const prefix = Wn.dictLimit;
    u32 const dictEndIndex = 2 + dms->contentSize;              // the dictionary's own indices run from 2 (window start) to its end
    u32 const dictIndexDelta = prefix - dictEndIndex;           // :484
    ZsaDmsView V;
    V.base = Wn.base; V.dbase = (dms->content - 2) - dictIndexDelta; V.prefix = prefix; V.dictLow = 2 + dictIndexDelta;
    u32 const istart = (u32)(src - Wn.base), iend = istart + srcSize, ilimit = iend - 8;
    u32 ip0 = istart, anchor = istart;
    u32 off1 = wv_uniform(rep[0]), off2 = wv_uniform(rep[1]), nbSeq = 0;
    ip0 += ((istart - prefix) + dms->contentSize == 0);         // :497 dictAndPrefixLength == 0
    u32 step = stepSize, nextStep = ip0 + 256;                  // :533, kStepIncr = 1 << 8
    u32 width = ZSA_DMS_W0;                                     // positions per window: narrow after a match, doubling while nothing is found
    if (srcSize >= 8 + stepSize)   // (ilimit does not wrap)
    for (;;) {   // one window per iteration
        // positions: consecutive while the step is 1, up to and including the first one at or beyond nextStep (whose own
        // step is 2); beyond that one position per window
        u32 nAct = 1;
        if (step == 1) { u32 const a = nextStep - ip0 + 1; nAct = a < width ? a : width; }
        u32 const cur = ip0 + (u32)lane;
        u32 const myStep = step + ((lane > 0 && cur >= nextStep) ? 1u : 0u);
        bool const valid = (u32)lane < nAct && cur + myStep <= ilimit;       // the reference reaches it (:530, :618)
        u32 nEff = (u32)__builtin_popcountll(wv_ballot(valid));              // (a prefix of the lanes)
        if (nEff == 0) break;
        bool ends = nEff < nAct;
        u64 v = 0; u32 h = 0, mIdx = 0, dcell = 0, dh = 0;
        if (valid) {
            v = wv_ldg64(V.base + cur);
            h = zsa_hash(v, hlog, mls);
            dh = zsa_hash(v, dictHBits, mls);
            mIdx = wv_ldg_cell(hashTable + h);
            dcell = wv_ldg_cell(dictTable + (dh >> 8));
        }
        {   // lanes that share their local cell with another lane of the window: cut the window below the second-lowest
            u32* const dw = dup + ((h >> 5) & 255u);
            u32 const bit = 1u << (h & 31u);
            bool late = false;
            if (valid) late = (wv_atomic_or_lds(dw, bit) & bit) != 0;
            wv_lds_fence();
            if (late) (void)wv_atomic_and_lds(dw, ~bit);
            wv_lds_fence();
            bool const shares = valid && (late || !(*dw & bit));
            wv_lds_fence();
            if (valid) *dw = 0;
            wv_lds_fence();
            u64 const G = wv_ballot(shares);
            u64 const G2 = G & (G - 1);
            if (G2) { u32 const w = (u32)__builtin_ctzll(G2); if (w < nEff) { nEff = w; ends = false; } }   // (w >= 1)
        }
        bool const mine = (u32)lane < nEff;
        // the three checks of a position (:536-592)
        u32 kind = 0;   // 1: repcode at cur + 1, 2: dictionary, 3: local table
        {   // the bytes of all three candidates in one round trip (a lane without one reads the block's first bytes)
            u32 const repIndex = cur + 1 - off1;
            u32 const w0 = (u32)v, w1 = (u32)(v >> 8);
            u32 const dIdx = dcell >> 8;
            bool const repTry = mine && (u32)((prefix - 1) - repIndex) >= 3;
            bool const dTry = mine && (dcell & 0xFFu) == (dh & 0xFFu) && dIdx > 2 && mIdx <= prefix;
            bool const lTry = mine && mIdx > prefix;
            u32 cR = zsa_dv32(V, repTry ? repIndex : istart);
            u32 cD = wv_ldg32(dTry ? dms->content - 2 + dIdx : V.base + istart);
            u32 cL = wv_ldg32(V.base + (lTry ? mIdx : istart));
            wv_join(cR, cD, cL);
            kind = (repTry && cR == w1) ? 1u : ((dTry && cD == w0) ? 2u : ((lTry && cL == w0) ? 3u : 0u));
        }
        u64 const found = wv_ballot(kind != 0);
        u32 const nIns = found ? (u32)__builtin_ctzll(found) + 1 : nEff;     // positions that enter the local table (:535)
        if ((u32)lane < nIns) wv_stg_cell(hashTable + h, cur);               // (distinct cells: the window was cut)
        wv_lds_fence();
        if (!found) {
            if (ends) break;
            // the step after the last position of the window (:613-618)
            u32 const last = ip0 + nEff - 1;
            u32 const s = step + ((nEff > 1 && last >= nextStep) ? 1u : 0u);   // the step that position was reached with
            if (s != step) { step = s; nextStep += 256; }
            ip0 = last + step;
            if (ip0 >= nextStep) { step++; nextStep += 256; }
            width = width * 2 < ZSA_WAVE ? width * 2 : ZSA_WAVE;
            continue;
        }
        int const e = __builtin_ctzll(found);
        u32 const eKind = wv_readlane(kind, e), curE = ip0 + (u32)e;
        u32 start, mLength, offcode;
        if (eKind == 1) {
            start = curE + 1;
            mLength = zsa_dv_count(V, start + 4, start - off1 + 4, iend, lane) + 4;
            offcode = 1;
        } else {
            u32 const m = eKind == 2 ? (wv_readlane(dcell, e) >> 8) + dictIndexDelta : wv_readlane(mIdx, e);
            u32 const low = eKind == 2 ? V.dictLow : prefix;
            u32 fwd, bwd;
            zsa_dv_extend(V, curE, m, iend, anchor, low, &fwd, &bwd, lane);
            start = curE - bwd;
            mLength = 4 + fwd + bwd;
            off2 = off1; off1 = curE - m;
            offcode = off1 + 3;
        }
        if (lane == 0) wv_stg64(seqOut + nbSeq, zsa_packseq(start - anchor, mLength - ZSA_MINMATCH, offcode));
        nbSeq++;
        ip0 = start + mLength;
        anchor = ip0;
        if (ip0 <= ilimit) {
            {   // :599-601: current + 2, then ip0 - 2
                u32 const pp = lane == 0 ? curE + 2 : ip0 - 2;
                u32 hh = 0;
                if (lane < 2) hh = zsa_hash(wv_ldg64(V.base + pp), hlog, mls);
                if (lane == 0) wv_stg_cell(hashTable + hh, pp);
                wv_lds_fence();
                if (lane == 1) wv_stg_cell(hashTable + hh, pp);
                wv_lds_fence();
            }
            while (ip0 <= ilimit) {   // :603-620: the second offset, immediately and repeatedly
                u32 const repIndex2 = ip0 - off2;
                if (!((u32)((prefix - 1) - repIndex2) >= 3)) break;
                if (wv_uniform(zsa_dv32(V, repIndex2)) != wv_uniform(wv_ldg32(V.base + ip0))) break;
                u32 const len = zsa_dv_count(V, ip0 + 4, repIndex2 + 4, iend, lane) + 4;
                { u32 const t = off2; off2 = off1; off1 = t; }
                if (lane == 0) {
                    wv_stg64(seqOut + nbSeq, zsa_packseq(0, len - ZSA_MINMATCH, 1));
                    wv_stg_cell(hashTable + zsa_hash(wv_ldg64(V.base + ip0), hlog, mls), ip0);
                }
                wv_lds_fence();
                nbSeq++;
                ip0 += len;
                anchor = ip0;
            }
        }
        step = stepSize; nextStep = ip0 + 256;                  // :530-533 a new search
        width = ZSA_DMS_W0;
        if (ip0 + step > ilimit) break;
    }
    if (lane == 0) { rep[0] = off1; rep[1] = off2; rep[2] = iend - anchor; }   // :625-629 (no offset is ever set aside in this variant)
    return wv_uniform(nbSeq);
}

// ------------------------------------------------------------------------------------------------------------------------
// The ZSTD_fast parse with the dictionary as an EXTERNAL segment (zstd_fast.c:685-936: payloads above 8 KiB with a CDict,
// whose table was copied into the frame's), executed by all 64 lanes of the wave.
//
// The reference walks pairs of positions (P, P + 1), the next pair `step` further on (2 at level 1, growing by one every 128
// bytes without a match), and per pair performs, in this order: enter P; repcode check at the NEXT pair's first position;
// table check of P; enter P + 1; table check of P + 1.  While the step is 2 the positions of a search are consecutive, and a
// WINDOW is n pairs = 2n positions plus the position after them, lane k <-> position ip0 + k:
//   * every lane hashes its position and reads its cell; even lanes from 2 on also make the repcode check of the pair before
//     them; the three checks of pair i have ranks 3i (repcode, made by lane 2i + 2), 3i + 1, 3i + 2 and the lowest rank over
//     two ballots is the event the reference stops at;
//   * shared cells cut the window as in the parser above, at pair granularity; the two positions of one pair sharing a cell
//     (runs of equal bytes) is resolved in place: the second one's candidate is the first;
//   * dictionary and payload are one index space (ZsaDmsView), so two-segment match lengths are one forward count.
// Beyond step 2 a window is a single pair.  A window starts narrow (the tables are small here, but every position costs a
// candidate read) and doubles while nothing is found.
// Returns the number of sequences, or 0xFFFFFFFF when the external segment has been invalidated for this block (:737-738):
// the caller then runs the plain parser.  rep: LDS, [0], [1] in and out, [2] receives the last literal run.
ZSA_DEVFN u32 zsa_parse_ext_wave(u32* hashTable, u32 hlog, u32 mls, u32 tlen, u32 wlog, const u8* wbase, const u8* wdictBase, u32 wDictLimit,
                                 u32 wLowLimit, u32 loadedDictEnd, const u8* src, u32 srcSize, u32* rep, u64* seqOut, u32* dup)
{
    int const lane = wv_lane();
    hlog = wv_uniform(hlog); mls = wv_uniform(mls); srcSize = wv_uniform(srcSize); wlog = wv_uniform(wlog);
    hashTable = (u32*)wv_uniform64((u64)hashTable); seqOut = (u64*)wv_uniform64((u64)seqOut);
    src = (const u8*)wv_uniform64((u64)src);
    u32 const stepSize = wv_uniform(tlen + !tlen + 1);          // :692
    ZsaDmsView V;
    V.base = (const u8*)wv_uniform64((u64)wbase); V.dbase = (const u8*)wv_uniform64((u64)wdictBase);
    u32 const istart = (u32)(src - V.base), iend = istart + srcSize, ilimit = iend - 8;
    {   // ZSTD_getLowestMatchIndex (zstd_compress_internal.h:1290-1302)
        u32 const maxDistance = 1u << wlog, lowestValid = wv_uniform(wLowLimit);
        u32 const withinWindow = (iend - lowestValid > maxDistance) ? iend - maxDistance : lowestValid;
        V.dictLow = wv_uniform(loadedDictEnd) != 0 ? lowestValid : withinWindow;
    }
    u32 const dictStart = V.dictLow;
    {   u32 const dl = wv_uniform(wDictLimit); V.prefix = dl < dictStart ? dictStart : dl; }
    u32 const prefix = V.prefix;
    if (prefix == dictStart) return 0xFFFFFFFFu;
    u32 ip0 = istart, anchor = istart;
    u32 off1 = wv_uniform(rep[0]), off2 = wv_uniform(rep[1]), saved1 = 0, saved2 = 0, nbSeq = 0;
    {   u32 const maxRep = ip0 - dictStart;                     // :741-744
        if (off2 >= maxRep) { saved2 = off2; off2 = 0; }
        if (off1 >= maxRep) { saved1 = off1; off1 = 0; }
    }
    // a search: ip0, the distance d to the next pair (fixed one pair ahead), the step that later pairs take, its next raise
    u32 d = stepSize, step = stepSize, nextStep = ip0 + 128, width = 4;
    if (srcSize >= 8 && ip0 + d + 1 < ilimit)
    for (;;) {   // one window per iteration
        bool const fast = d == 2 && step == 2;
        u32 nP = 1;                                             // pairs
        if (fast) {   // no raise of the step inside the window: ip0 + 2 nP + 2 < nextStep
            u32 const room = nextStep > ip0 + 4 ? (nextStep - ip0 - 3) >> 1 : 0;
            nP = room < width ? room : width;
            if (nP == 0) nP = 1;
        }
        bool const wide = fast && (ip0 + 2 * nP + 2 < nextStep);
        if (!wide) nP = 1;
        // lane k: position ip0 + k for k < 2 nP; lane 2 nP: the position after the last pair (repcode check, and its hash for
        // the insert that follows a match)
        u32 const cur = (u32)lane < 2 * nP ? ip0 + (u32)lane : ip0 + 2 * (nP - 1) + d;   // (wide: d == 2)
        u32 const myPair = (u32)lane >> 1;
        bool const pairValid = myPair < nP && (ip0 + 2 * myPair + d + 1 < ilimit);       // the pair is entered: `ip3 < ilimit` (:756, :846)
        u32 nEffP = (u32)__builtin_popcountll(wv_ballot(pairValid)) >> 1;               // (a prefix of the lanes, two per pair)
        if (nEffP == 0) break;
        bool ends = nEffP < nP;
        bool const loads = (u32)lane <= 2 * nEffP;
        u64 v = 0; u32 h = 0, mIdx = 0;
        if (loads) { v = wv_ldg64(V.base + cur); h = zsa_hash(v, hlog, mls); }
        if ((u32)lane < 2 * nEffP) mIdx = wv_ldg_cell(hashTable + h);
        bool pairShares = false;
        {   // lanes that share their cell with another lane of the window: cut below the pair of the second-lowest
            bool const on = (u32)lane < 2 * nEffP;
            u32* const dw = dup + ((h >> 5) & 255u);
            u32 const bit = 1u << (h & 31u);
            bool late = false;
            if (on) late = (wv_atomic_or_lds(dw, bit) & bit) != 0;
            wv_lds_fence();
            if (late) (void)wv_atomic_and_lds(dw, ~bit);
            wv_lds_fence();
            bool const shares = on && (late || !(*dw & bit));
            wv_lds_fence();
            if (on) *dw = 0;
            wv_lds_fence();
            u64 const G = wv_ballot(shares);
            u64 const G2 = G & (G - 1);
            if (G2) {
                u32 const w = (u32)__builtin_ctzll(G2) >> 1;    // the pair that holds the second-lowest sharer
                if (w == 0) { nEffP = 1; ends = false; pairShares = true; }   // (then lanes 0 and 1 are the two lowest sharers)
                else if (w < nEffP) { nEffP = w; ends = false; }
            }
        }
        if (pairShares) {   // one pair: exact by hand -- position ip0 + 1 reads what position ip0 has just entered
            u32 const h0 = wv_readlane(h, 0), h1 = wv_readlane(h, 1);
            pairShares = h0 == h1;                              // (the bitmap folds cell numbers: they may differ after all)
            if (pairShares && lane == 1) mIdx = ip0;
        }
        bool const mine = (u32)lane < 2 * nEffP;
        // repcode check of pair j - 1 by lane 2 j (:761-781); table check of the lane's own position (:785-800, :815-830)
        bool const repLane = !(lane & 1) && lane >= 2 && (u32)lane <= 2 * nEffP;
        u32 const repIndex = cur - off1;
        bool const repTry = repLane && off1 > 0 && (u32)(prefix - repIndex) >= 4;
        bool const tabTry = mine && mIdx >= dictStart;
        u32 cR = zsa_dv32(V, repTry ? repIndex : istart);
        u32 cM = zsa_dv32(V, tabTry ? mIdx : istart);
        { u64 z = 0; wv_join(z, cR, cM); }
        bool const rHit = repTry && cR == (u32)v;
        bool const mHit = tabTry && cM == (u32)v;
        u64 const RB = wv_ballot(rHit), MB = wv_ballot(mHit);
        u32 rankR = 0xFFFFFFFFu, rankM = 0xFFFFFFFFu;
        if (RB) rankR = 3u * (((u32)__builtin_ctzll(RB) >> 1) - 1u);
        if (MB) { u32 const b = (u32)__builtin_ctzll(MB); rankM = 3u * (b >> 1) + 1u + (b & 1u); }
        bool const found = (RB | MB) != 0;
        u32 const rank = rankR < rankM ? rankR : rankM;
        u32 const ePair = found ? rank / 3u : 0, eRole = found ? rank - 3u * ePair : 0;   // role 0: repcode, 1: first, 2: second position
        // positions that enter the table: all of the window, or up to the event's pair -- its second position too, as the
        // reference enters `ip1` after every match (:870-873); when the second check hit, `ip1` is the position after the
        // pair, which enters only if the match ends beyond it (below)
        u32 const nIns = found ? 2 * ePair + 2 : 2 * nEffP;
        {   bool st = (u32)lane < nIns;
            if (pairShares && lane == 0 && nIns >= 2) st = false;   // same cell: the later position stays
            if (st) wv_stg_cell(hashTable + h, cur);
            wv_lds_fence();
        }
        if (!found) {
            if (ends) break;
            if (wide) ip0 += 2 * nEffP;                         // d, step unchanged: no raise inside the window
            else {                                              // :840-846
                ip0 += d; d = step;
                if (ip0 + step >= nextStep) { step++; nextStep += 128; }
            }
            if (!(ip0 + d + 1 < ilimit)) break;
            width = width * 2 < 31 ? width * 2 : 31;
            continue;
        }
        u32 const current0 = ip0 + 2 * ePair + (eRole == 2);
        u32 start, mLength, offcode;
        if (eRole == 0) {
            u32 const ip2 = wv_readlane(cur, 2 * (int)ePair + 2);
            u32 const rIdx = ip2 - off1;
            // :776 `mLength = ip0[-1] == match0[-1]`: the byte before the candidate is read in the candidate's own segment
            u32 const b0 = wv_uniform(wv_ldg8(V.base + ip2 - 1));
            u32 const b1 = wv_uniform(wv_ldg8((rIdx < prefix ? V.dbase : V.base) + rIdx - 1));
            u32 const back = b0 == b1;
            start = ip2 - back;
            mLength = back + 4 + zsa_dv_count(V, ip2 + 4, rIdx + 4, iend, lane);
            offcode = 1;
        } else {
            int const e = 2 * (int)ePair + (eRole == 2);
            u32 const m = wv_readlane(mIdx, e);
            u32 const low = m < prefix ? dictStart : prefix;
            u32 fwd, bwd;
            zsa_dv_extend(V, current0, m, iend, anchor, low, &fwd, &bwd, lane);
            start = current0 - bwd;
            mLength = 4 + fwd + bwd;
            off2 = off1; off1 = current0 - m;
            offcode = off1 + 3;
        }
        if (lane == 0) wv_stg64(seqOut + nbSeq, zsa_packseq(start - anchor, mLength - ZSA_MINMATCH, offcode));
        nbSeq++;
        ip0 = start + mLength;
        anchor = ip0;
        if (eRole == 2) {   // :870-873 `if (ip1 < ip0)`: with a large step the position after the pair can lie beyond the match
            int const x = 2 * (int)ePair + 2;
            if (wv_readlane(cur, x) < ip0 && lane == x) wv_stg_cell(hashTable + h, cur);   // (after the window's own: may share a cell)
            wv_lds_fence();
        }
        if (ip0 <= ilimit) {
            {   // :877-879: current0 + 2, then ip0 - 2
                u32 const pp = lane == 0 ? current0 + 2 : ip0 - 2;
                u32 hh = 0;
                if (lane < 2) hh = zsa_hash(wv_ldg64(V.base + pp), hlog, mls);
                if (lane == 0) wv_stg_cell(hashTable + hh, pp);
                wv_lds_fence();
                if (lane == 1) wv_stg_cell(hashTable + hh, pp);
                wv_lds_fence();
            }
            while (ip0 <= ilimit) {   // :881-897: the second offset, immediately and repeatedly
                u32 const repIndex2 = ip0 - off2;
                if (!((u32)((prefix - 1) - repIndex2) >= 3 && off2 > 0)) break;
                if (wv_uniform(zsa_dv32(V, repIndex2)) != wv_uniform(wv_ldg32(V.base + ip0))) break;
                u32 const len = zsa_dv_count(V, ip0 + 4, repIndex2 + 4, iend, lane) + 4;
                { u32 const t = off2; off2 = off1; off1 = t; }
                if (lane == 0) {
                    wv_stg64(seqOut + nbSeq, zsa_packseq(0, len - ZSA_MINMATCH, 1));
                    wv_stg_cell(hashTable + zsa_hash(wv_ldg64(V.base + ip0), hlog, mls), ip0);
                }
                wv_lds_fence();
                nbSeq++;
                ip0 += len;
                anchor = ip0;
            }
        }
        d = stepSize; step = stepSize; nextStep = ip0 + 128; width = 4;   // :748-753 a new search
        if (!(ip0 + d + 1 < ilimit)) break;
    }
    saved2 = ((saved1 != 0) && (off1 != 0)) ? saved1 : saved2;             // _cleanup (:851-861)
    if (lane == 0) { rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2; rep[2] = iend - anchor; }
    return wv_uniform(nbSeq);
}
