// zsa_dfast.h -- the ZSTD_dfast parse without a dictionary (zstd_double_fast.c:98-305): levels 3-4, level 2 at <= 256 KB,
// executed by all 64 lanes of the wave.
//
// Two tables in the workgroup's global scratch: `hashLong` keyed by 8 bytes (hashLog bits) and `hashSmall` keyed by
// minMatch bytes (chainLog bits).  The reference visits positions one by one (step 1, growing by one every 256 bytes
// without a match) and per position p: enters p into both tables; checks the repcode at p + 1; checks the long table's
// candidate for p (8 equal bytes); checks the short table's candidate (4 equal bytes) and, if that one holds, prefers a long
// match at the NEXT position over it.
//
// Here a WINDOW is up to 63 consecutive positions plus one look-ahead position, lane k <-> position ip0 + k:
//   * every lane reads its 8 source bytes, hashes them twice, reads both cells and then -- in one more round trip -- the
//     bytes of the long candidate, of the short candidate and of the repcode; lane order == position order and
//     rep > long > short inside a lane, so the first lane of ONE ballot with any hit is the event the reference stops at;
//   * the "long match at the next position" of a short hit is the long probe of the next lane, which the window has already
//     made (the lane after the last position probes the long table only);
//   * a lane's probes are exact as long as no earlier lane of the window falls into the same cell of the same table; lanes
//     that share a cell are found with a bitmap pass per table (cell numbers modulo 16384) and the window is cut below the
//     second-lowest of them; a look-ahead probe that may be stale is simply repeated after the window's stores;
//   * positions up to the event enter both tables in one store each (distinct cells), then the reference's
//     complementary positions, then the immediate-repcode loop.
// Match lengths are counted by all lanes, a byte per lane, forwards and backwards in the same round trip.
//
// Cells carry a FINGERPRINT above the position (posBits != 0: positions need posBits bits; the long table's from all 8 bytes,
// the short table's from the 4 bytes its check compares), so equal bytes imply an equal fingerprint and a window decides its
// probable event from the cells alone -- no round trip for the candidates' bytes.  The bytes are compared where the event is
// taken: a long match by the extension that measures it anyway, a short match together with the look-ahead's long candidate.
// A candidate whose fingerprint agreed by accident is struck out and the window's ballot is taken again.
// Included by zsa_encode.h (uses zsa_hash, zsa_packseq from there).
#pragma once

#ifndef ZSA_DFAST_W0
#define ZSA_DFAST_W0 8u    // positions in the first window of a search
#endif

// common length of [ip, iend) and [m, ...), m < ip; uniform arguments, all lanes (ZSTD_count, zstd_compress_internal.h:755-774)
ZSA_DEV u32 zsa_df_count(const u8* base, u32 ip, u32 m, u32 iend, int lane)
{
    u32 total = 0;
    for (;;) {
        u32 const a = ip + total + (u32)lane;
        bool const same = a < iend && wv_ldg8(base + a) == wv_ldg8(base + m + total + (u32)lane);
        u64 const stop = wv_ballot(!same);
        if (stop) return total + (u32)__builtin_ctzll(stop);
        total += ZSA_WAVE;
    }
}
// forward from (ip + skip, m + skip) and backward from (ip - 1, m - 1) -- the "catch up" loops, :193/:245/:253: ip stays
// above `anchor`, m above `low` -- with the first round of both directions in one round trip
ZSA_DEV void zsa_df_extend(const u8* base, u32 ip, u32 m, u32 skip, u32 iend, u32 anchor, u32 low, u32* fwd, u32* bwd, int lane)
{
    u32 const room = (ip - anchor) < (m - low) ? (ip - anchor) : (m - low);
    u32 const a = ip + skip + (u32)lane;
    bool const fin = a < iend, bin = (u32)lane < room;
    u32 const f0 = wv_ldg8(base + (fin ? a : ip)), f1 = wv_ldg8(base + (fin ? m + skip + (u32)lane : ip));
    u32 const b0 = wv_ldg8(base + (bin ? ip - 1 - (u32)lane : ip)), b1 = wv_ldg8(base + (bin ? m - 1 - (u32)lane : ip));
    u64 const fstop = wv_ballot(!(fin && f0 == f1)), bstop = wv_ballot(!(bin && b0 == b1));
    *fwd = fstop ? (u32)__builtin_ctzll(fstop) : ZSA_WAVE + zsa_df_count(base, ip + skip + ZSA_WAVE, m + skip + ZSA_WAVE, iend, lane);
    u32 back = bstop ? (u32)__builtin_ctzll(bstop) : ZSA_WAVE;   // (a lane at or beyond `room` counts as a mismatch: back <= room)
    if (!bstop) while (back < room) {   // more than 64 bytes backwards: rare
        u32 const k = back + (u32)lane;
        bool const same = k < room && wv_ldg8(base + ip - 1 - k) == wv_ldg8(base + m - 1 - k);
        u64 const stop = wv_ballot(!same);
        if (stop) { back += (u32)__builtin_ctzll(stop); break; }
        back += ZSA_WAVE;
    }
    *bwd = back;
}

// Parses one block.  base: window base (index i <-> base + i).  bmL, bmS: LDS, 512 words each, all zero (left all zero).
// rep: LDS, [0] and [1] in and out, [2] receives the last literal run.  Returns the number of sequences appended to seqOut.
// All arguments and the result are wave-uniform.  (A function of its own: its loop gets its own register allocation.)
// fingerprints: equal 8 (4) bytes give equal values; the top 32 - posBits bits are kept
ZSA_DEV u32 zsa_df_fpL(u64 v) { return ((u32)v ^ ((u32)(v >> 32) * 2654435761U)) * 2246822519U; }
ZSA_DEV u32 zsa_df_fpS(u64 v) { return (u32)v * 2246822519U; }
#ifdef ZSA_FP_TEST_BITS   // test builds: few fingerprint bits, so that accidental agreement is common
#define ZSA_DF_FP(h) (((h) >> posBits) & ((1u << ZSA_FP_TEST_BITS) - 1u))
#else
#define ZSA_DF_FP(h) ((h) >> posBits)
#endif
#define ZSA_DF_CELL_L(p, v) (fpOn ? (ZSA_DF_FP(zsa_df_fpL(v)) << posBits) | (p) : (p))
#define ZSA_DF_CELL_S(p, v) (fpOn ? (ZSA_DF_FP(zsa_df_fpS(v)) << posBits) | (p) : (p))

ZSA_DEVFN u32 zsa_parse_dfast_wave(u32* hashLong, u32* hashSmall, u32 hBitsL, u32 hBitsS, u32 mls, u32 wlog, const u8* base, u32 dictLimit,
                                   const u8* src, u32 srcSize, u32* rep, u64* seqOut, u32* bmL, u32* bmS, u32 posBits)
{
    int const lane = wv_lane();
    posBits = wv_uniform(posBits);
    bool const fpOn = posBits != 0;                                   // (0: plain positions in the cells, candidates' bytes read per window)
    u32 const posMask = fpOn ? (1u << posBits) - 1u : 0xFFFFFFFFu;
    hashLong = (u32*)wv_uniform64((u64)hashLong); hashSmall = (u32*)wv_uniform64((u64)hashSmall);
    hBitsL = wv_uniform(hBitsL); hBitsS = wv_uniform(hBitsS); mls = wv_uniform(mls); wlog = wv_uniform(wlog);
    base = (const u8*)wv_uniform64((u64)base); dictLimit = wv_uniform(dictLimit);
    src = (const u8*)wv_uniform64((u64)src); srcSize = wv_uniform(srcSize); seqOut = (u64*)wv_uniform64((u64)seqOut);
    u32 const istart = (u32)(src - base), iend = istart + srcSize, ilimit = iend - 8;
    u32 const maxDistance = 1u << wlog;
    // ZSTD_getLowestPrefixIndex (zstd_compress_internal.h:1307-1318), no loaded dictionary
    u32 const prefixLowest = (iend - dictLimit > maxDistance) ? iend - maxDistance : dictLimit;
    u32 ip0 = istart, anchor = istart;
    u32 off1 = wv_uniform(rep[0]), off2 = wv_uniform(rep[1]), saved1 = 0, saved2 = 0, nbSeq = 0;
    ip0 += (ip0 == prefixLowest);                                     // :132
    {   u32 const windowLow = (ip0 - dictLimit > maxDistance) ? ip0 - maxDistance : dictLimit;   // :133-139
        u32 const maxRep = ip0 - windowLow;
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
    u32 step = 1, nextStep = ip0 + 256;                               // :157-158, kStepIncr = 1 << kSearchStrength
    // Every position of a window costs two scattered cell reads and three candidate reads, and the tables of the frames in
    // flight are far larger than the caches: a window is only as wide as the search is likely to last.  It starts narrow
    // after a match and doubles while nothing is found.
    u32 width = ZSA_DFAST_W0;
    // pv: the 8 source bytes at pvAt + lane for lanes below pvN (where readable): requested along with the last match's inserts,
    // or left over from the previous window of the same search
    u64 pv = 0; u32 pvAt = 0, pvN = 0;
    if (srcSize >= 9 && ip0 + 1 <= ilimit)
    for (;;) {   // one window per iteration
        // positions: consecutive while the step is 1, up to and including the first one at or beyond nextStep (whose own
        // step is 2); beyond that one position per window.  Then one look-ahead position: where the last one steps to.
        u32 nAct = 1;
        if (step == 1) { u32 const a = nextStep - ip0 + 1; nAct = a < width ? a : width; }
        u32 const lastAct = ip0 + nAct - 1;
        u32 const lastStep = step == 1 ? ((nAct > 1 && lastAct >= nextStep) ? 2u : 1u) : step;
        u32 const cur = (u32)lane < nAct ? ip0 + (u32)lane : lastAct + lastStep;
        u32 const myStep = (u32)lane == nAct - 1 ? lastStep : 1u;
        bool const valid = (u32)lane < nAct && cur + myStep <= ilimit;          // the reference reaches it (:164, :223)
        u32 nEff = (u32)__builtin_popcountll(wv_ballot(valid));                 // (a prefix of the lanes)
        if (nEff == 0) break;
        bool ends = nEff < nAct;
        bool const probe = (u32)lane <= nEff;                                   // lane nEff: look-ahead, long table only
        bool const both = (u32)lane < nEff;
        u64 v = 0; u32 hL = 0, hS = 0, rawL = 0, rawS = 0;
        if (pvAt != ip0) { pv = wv_ldg64(base + (ip0 + (u32)lane <= ilimit ? ip0 + (u32)lane : ip0)); pvAt = ip0; pvN = ZSA_WAVE; }
        if (probe) {
            v = (cur == ip0 + (u32)lane && (u32)lane < pvN) ? pv : wv_ldg64(base + cur);
            hL = zsa_hash(v, hBitsL, 8);
            hS = zsa_hash(v, hBitsS, mls);
            rawL = wv_ldg_cell(hashLong + hL);
        }
        if (both) rawS = wv_ldg_cell(hashSmall + hS);
        u32 cR = wv_ldg32(base + (both ? cur + 1 - off1 : istart));             // the repcode's bytes travel with the cells
        wv_join(rawL, rawS, cR);
        u32 const cellL = ZSA_DF_CELL_L(cur, v), cellS = ZSA_DF_CELL_S(cur, v);
        u32 const iL = rawL & posMask, iS = rawS & posMask;
        u64 sharedL;
        {   // lanes that share a cell with another lane of the window, per table: cut the window below the second-lowest
            u32* const dwL = bmL + ((hL >> 5) & 511u); u32 const bitL = 1u << (hL & 31u);
            u32* const dwS = bmS + ((hS >> 5) & 511u); u32 const bitS = 1u << (hS & 31u);
            bool lateL = false, lateS = false;
            if (probe) lateL = (wv_atomic_or_lds(dwL, bitL) & bitL) != 0;
            if (both) lateS = (wv_atomic_or_lds(dwS, bitS) & bitS) != 0;
            wv_lds_fence();
            if (lateL) (void)wv_atomic_and_lds(dwL, ~bitL);
            if (lateS) (void)wv_atomic_and_lds(dwS, ~bitS);
            wv_lds_fence();
            bool const shL = probe && (lateL || !(*dwL & bitL));
            bool const shS = both && (lateS || !(*dwS & bitS));
            wv_lds_fence();
            if (probe) *dwL = 0;
            if (both) *dwS = 0;
            wv_lds_fence();
            sharedL = wv_ballot(shL);
            u64 const sharedS = wv_ballot(shS);
            u64 const GL2 = sharedL & (sharedL - 1), GS2 = sharedS & (sharedS - 1);
            u32 w = ZSA_WAVE;
            if (GL2) w = (u32)__builtin_ctzll(GL2);
            if (GS2) { u32 const ws = (u32)__builtin_ctzll(GS2); w = ws < w ? ws : w; }
            if (w < nEff) { nEff = w; ends = false; }                           // (w >= 1)
        }
        bool const mine = (u32)lane < nEff;
        bool const hasL = probe && iL > prefixLowest, hasS = mine && iS > prefixLowest;
        // what a lane may hold, in the reference's order: 1: repcode at cur + 1 (:179), 2: long match at cur (:188), 4: short
        // match at cur (:201).  With fingerprints 2 and 4 are probable until the bytes are compared below.
        bool longHit; u32 flags = 0;
        if (fpOn) {
            longHit = hasL && (rawL >> posBits) == ZSA_DF_FP(zsa_df_fpL(v));
            if (hasS && (rawS >> posBits) == ZSA_DF_FP(zsa_df_fpS(v))) flags = 4;
        } else {   // candidates' bytes in one more round trip (a lane without a candidate reads its own block's start)
            u64 cL = wv_ldg64(base + (hasL ? iL : istart));
            u32 cS = wv_ldg32(base + (hasS ? iS : istart)), c3 = 0;
            wv_join(cL, cS, c3);
            longHit = hasL && cL == v;
            if (hasS && cS == (u32)v) flags = 4;
        }
        if (longHit) flags |= 2;
        if (off1 > 0 && cR == (u32)(v >> 8)) flags |= 1;
        if (!mine) flags = 0;
        u32 nStored = 0;                                                        // positions of this window already in the tables
      again: ;
        u64 const found = wv_ballot(flags != 0);
        {   u32 const nIns = found ? (u32)__builtin_ctzll(found) + 1 : nEff;    // positions that enter the tables (:174) -- at least these
            if ((u32)lane >= nStored && (u32)lane < nIns) { wv_stg_cell(hashLong + hL, cellL); wv_stg_cell(hashSmall + hS, cellS); }   // (distinct cells: the window was cut)
            wv_lds_fence();
            nStored = nIns;
        }
        if (!found) {
            if (ends) break;
            // the step after the last position of the window (:208-223)
            u32 const last = ip0 + nEff - 1;
            u32 const s = step + ((nEff > 1 && last >= nextStep) ? 1u : 0u);    // the step that position was reached with
            if (s != step) { step = s; nextStep += 256; }
            {   u32 const d = last + step - ip0;       // the source bytes this window fetched beyond its end serve the next one
                if (d < pvN) {
                    pv = (u64)wv_shfl((u32)pv, lane + (int)d) | ((u64)wv_shfl((u32)(pv >> 32), lane + (int)d) << 32);
                    pvN -= d; pvAt = ip0 + d;
                }
            }
            ip0 = last + step;
            if (ip0 >= nextStep) { step++; nextStep += 256; }
            width = width * 2 < ZSA_WAVE - 1 ? width * 2 : ZSA_WAVE - 1;
            continue;
        }
        int const e = __builtin_ctzll(found);
        u32 const eFlags = wv_readlane(flags, e), curE = ip0 + (u32)e;
        u32 const eStep = (u32)e == nAct - 1 ? lastStep : 1u;                   // `step` as the reference holds it at this position
        u32 const ip1 = wv_readlane(cur, e + 1), hl1 = wv_readlane(hL, e + 1);
        u32 start, mLength, offBase;
        if (eFlags & 1) {
            start = curE + 1;
            mLength = zsa_df_count(base, start + 4, start + 4 - off1, iend, lane) + 4;
            offBase = 1;
        } else {
            u32 ipM = curE, m, skip;
            if (eFlags & 2) { m = wv_readlane(iL, e); skip = fpOn ? 0 : 8; }    // (with fingerprints the extension compares the first 8 bytes too)
            else {   // _search_next_long (:237-253): a long match at ip1 beats the short one at ip
                u32 raw1; bool prob1;
                if (!((sharedL >> (e + 1)) & 1)) { raw1 = wv_readlane(rawL, e + 1); prob1 = wv_readlane(longHit ? 1u : 0u, e + 1) != 0; }
                else {   // that lane's probe may predate a store of this window: read the cell again
                    raw1 = wv_uniform(wv_ldg_cell(hashLong + hl1));
                    prob1 = (raw1 & posMask) > prefixLowest && (!fpOn || (raw1 >> posBits) == (wv_readlane(cellL, e + 1) >> posBits));
                }
                u32 const iL1 = raw1 & posMask, mS = wv_readlane(iS, e);
                bool hit1 = prob1, shortOk = true;
                if (fpOn || ((sharedL >> (e + 1)) & 1)) {   // the bytes of the short candidate and of the look-ahead's long one, one round trip
                    u64 const v1 = (u64)wv_readlane((u32)v, e + 1) | ((u64)wv_readlane((u32)(v >> 32), e + 1) << 32);
                    u64 c1 = wv_ldg64(base + (prob1 ? iL1 : istart));
                    u32 c4 = wv_ldg32(base + mS), c3 = 0;
                    wv_join(c1, c4, c3);
                    hit1 = prob1 && wv_uniform64(c1) == v1;
                    shortOk = wv_uniform(c4) == wv_readlane((u32)v, e);
                }
                if (!shortOk) { if (lane == e) flags &= ~4u; goto again; }      // (fingerprints only) not a match: the reference moves on
                if (hit1) { ipM = ip1; m = iL1; skip = 8; } else { m = mS; skip = 4; }
            }
            u32 fwd, bwd;
            zsa_df_extend(base, ipM, m, skip, iend, anchor, prefixLowest, &fwd, &bwd, lane);
            if (skip == 0) {
                if (fwd < 8) { if (lane == e) flags &= ~2u; goto again; }       // the long candidate's fingerprint agreed by accident
            }
            start = ipM - bwd;
            mLength = skip + fwd + bwd;
            off2 = off1; off1 = ipM - m;                                        // _match_found (:257-259)
            u32 const cell1 = wv_readlane(cellL, e + 1);
            if (eStep < 4 && lane == 0) wv_stg_cell(hashLong + hl1, cell1);     // :261-266
            wv_lds_fence();
            offBase = off1 + 3;
        }
        if (lane == 0) wv_stg64(seqOut + nbSeq, zsa_packseq(start - anchor, mLength - ZSA_MINMATCH, offBase));
        nbSeq++;
        ip0 = start + mLength;                                                  // _match_stored (:274-303)
        anchor = ip0;
        if (ip0 <= ilimit) {
            // complementary insertion (:279-287): curr + 2 and ip - 2 into the long table, curr + 2 and ip - 1 into the short
            // one; lane 2 fetches what the immediate-repcode check needs in the same round trip
            u32 const pp = lane == 0 ? curE + 2 : (lane == 1 ? ip0 - 2 : ip0);
            u64 w = 0; u32 r2 = 0, r3 = 0;
            if (lane < 3) w = wv_ldg64(base + pp);
            if (lane == 2) r2 = wv_ldg32(base + ip0 - off2);
            pv = wv_ldg64(base + (ip0 + (u32)lane <= ilimit ? ip0 + (u32)lane : ip0)); pvAt = ip0; pvN = ZSA_WAVE;   // the next window's source bytes
            wv_join(pv, r2, r3);
            u32 const hl = zsa_hash(w, hBitsL, 8);
            u32 const hs = zsa_hash(lane == 1 ? w >> 8 : w, hBitsS, mls);       // lane 1: the bytes at ip - 1 (mls <= 7)
            u32 const cl = ZSA_DF_CELL_L(pp, w);
            if (lane == 0) { wv_stg_cell(hashLong + hl, cl); wv_stg_cell(hashSmall + hs, ZSA_DF_CELL_S(pp, w)); }
            wv_lds_fence();
            if (lane == 1) { wv_stg_cell(hashLong + hl, cl); wv_stg_cell(hashSmall + hs, ZSA_DF_CELL_S(pp + 1, w >> 8)); }
            wv_lds_fence();
            u32 vi = wv_readlane((u32)w, 2), ri = wv_readlane(r2, 2);
            u32 hli = wv_readlane(hl, 2), hsi = wv_readlane(hs, 2);
            u32 cli = wv_readlane(cl, 2), csi = wv_readlane(ZSA_DF_CELL_S(pp, w), 2);
            while ((off2 > 0) & (vi == ri)) {   // immediate repcode (:290-302); ip0 <= ilimit holds
                u32 const rLength = zsa_df_count(base, ip0 + 4, ip0 + 4 - off2, iend, lane) + 4;
                { u32 const t = off2; off2 = off1; off1 = t; }
                if (lane == 0) {
                    wv_stg_cell(hashSmall + hsi, csi); wv_stg_cell(hashLong + hli, cli);
                    wv_stg64(seqOut + nbSeq, zsa_packseq(0, rLength - ZSA_MINMATCH, 1));
                }
                wv_lds_fence();
                nbSeq++;
                ip0 += rLength;
                anchor = ip0;
                if (ip0 > ilimit) break;
                pv = wv_ldg64(base + (ip0 + (u32)lane <= ilimit ? ip0 + (u32)lane : ip0)); pvAt = ip0; pvN = ZSA_WAVE;
                ri = wv_ldg32(base + ip0 - off2);
                wv_join(pv, ri, r3);
                u64 const wi = (u64)wv_readlane((u32)pv, 0) | ((u64)wv_readlane((u32)(pv >> 32), 0) << 32);
                vi = (u32)wi; ri = wv_uniform(ri);
                hli = zsa_hash(wi, hBitsL, 8); hsi = zsa_hash(wi, hBitsS, mls);
                cli = ZSA_DF_CELL_L(ip0, wi); csi = ZSA_DF_CELL_S(ip0, wi);
            }
        }
        step = 1; nextStep = ip0 + 256;                                         // :157-158 a new search
        width = ZSA_DFAST_W0;
        if (ip0 + 1 > ilimit) break;                                            // :164
    }
    saved2 = ((saved1 != 0) && (off1 != 0)) ? saved1 : saved2;                  // _cleanup (:225-235)
    if (lane == 0) { rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2; rep[2] = iend - anchor; }
    return wv_uniform(nbSeq);
}
#undef ZSA_DF_CELL_L
#undef ZSA_DF_FP
#undef ZSA_DF_CELL_S
