// zsa_dfast.h -- the ZSTD_dfast parse without a dictionary (zstd_double_fast.c:98-305): levels 3-4, level 2 at <= 256 KB.
//
// Two tables in the workgroup's global scratch: `hashLong` keyed by 8 bytes (hashLog bits) and `hashSmall` keyed by
// minMatch bytes (chainLog bits).  Per position the candidates are tried in the reference's order: repcode at ip+1,
// long match at ip, then -- only if the short table has a match at ip -- long match at ip+1 before settling for the
// short one.
//
// The chain is walked by one lane (the frames of a launch still run 16 per CU) as a software pipeline,
// because every table probe is a ~1000-cycle access (tools/microbench/loaded_latency.hip: the tables of the frames in
// flight total 1.5 GiB).  Three positions are in flight -- p0 being decided, p1, p2:
//   iteration for p0:   store p0 into both cells
//                       cells of p1 (requested one iteration ago) arrive; a cell that is the one just stored is patched to p0
//                       request: candidate bytes and repcode bytes of p1, both cells of p2
//                       decide p0 with the candidate bytes requested one iteration ago; request the source word of p3
// The three register sets rotate by renaming (the loop body is written out three times): copying a value that is still in
// flight would have to wait for it.  The section's state is seeded with wv_vzero() so that it stays in vector registers.
// What this buys is limited (+10 %) because a search episode of this corpus is only ~3 positions long: filling the
// pipeline, the match extension and the post-match inserts are ~5 round trips per match that no prefetch distance hides --
// evaluating all positions of an episode at once (the cooperative window of zsa_parse.h) is the real next step.
// Included by zsa_encode.h.
#pragma once

// Match extension on one lane: forward from (ip + skip, m + skip) (ZSTD_count, :191/:243/:251) and backward from (ip, m)
// (the "catch up" loops, :193/:245/:253), with all first-round loads -- 16 bytes forward on both sides, 8 bytes backward on
// both sides -- issued together: one memory round trip decides most matches of this corpus.  Returns the forward length
// beyond `skip`; ip and m are moved back by the backward length, which is returned in *bwd.
ZSA_DEV u32 zsa_dfast_extend(const u8*& ip, const u8*& m, u32 skip, const u8* iend, const u8* anchor, const u8* prefixLowest, u32* bwd)
{
    const u8* pIn = ip + skip; const u8* pM = m + skip;
    u32 const maxBack = (u32)((ip - anchor) < (m - prefixLowest) ? (ip - anchor) : (m - prefixLowest));
    bool const wideBack = maxBack > 0 && (m - prefixLowest) >= 8;     // then ip - 8 and m - 8 are readable
    bool const wideFwd = pIn + 16 <= iend;
    u64 a0 = 0, a1 = 0, b0 = 0, b1 = 0, pa = 0, pb = 0;
    if (wideFwd) { a0 = wv_ldg64(pIn); a1 = wv_ldg64(pIn + 8); b0 = wv_ldg64(pM); b1 = wv_ldg64(pM + 8); }
    if (wideBack) { pa = wv_ldg64(ip - 8); pb = wv_ldg64(m - 8); }
    u32 fwd = 0;
    if (wideFwd) {
        u64 const d0 = a0 ^ b0, d1 = a1 ^ b1;
        if (d0) fwd = (u32)__builtin_ctzll(d0) >> 3;
        else if (d1) fwd = 8 + ((u32)__builtin_ctzll(d1) >> 3);
        else {   // long match: keep going 16 bytes per round trip
            fwd = 16;
            for (;;) {
                if (pIn + fwd + 16 <= iend) {
                    u64 const e0 = wv_ldg64(pIn + fwd) ^ wv_ldg64(pM + fwd), e1 = wv_ldg64(pIn + fwd + 8) ^ wv_ldg64(pM + fwd + 8);
                    if (e0) { fwd += (u32)__builtin_ctzll(e0) >> 3; break; }
                    if (e1) { fwd += 8 + ((u32)__builtin_ctzll(e1) >> 3); break; }
                    fwd += 16;
                } else {
                    while (pIn + fwd < iend && wv_ldg8(pIn + fwd) == wv_ldg8(pM + fwd)) fwd++;
                    break;
                }
            }
        }
    } else {
        while (pIn + fwd < iend && wv_ldg8(pIn + fwd) == wv_ldg8(pM + fwd)) fwd++;
    }
    u32 back = 0;
    if (wideBack) {
        u64 const d = pa ^ pb;
        back = d ? ((u32)__builtin_clzll(d) >> 3) : 8;              // equal bytes counted from ip[-1] downwards
        if (back > maxBack) back = maxBack;
        if (back == 8) while (back < maxBack && wv_ldg8(ip - 1 - back) == wv_ldg8(m - 1 - back)) back++;
    } else {
        while (back < maxBack && wv_ldg8(ip - 1 - back) == wv_ldg8(m - 1 - back)) back++;
    }
    ip -= back; m -= back;
    *bwd = back;
    return fwd;
}
// ZSTD_count alone (repcode matches have no backward step at this strategy), 16 bytes per round trip
ZSA_DEV u32 zsa_dfast_count(const u8* pIn, const u8* pMatch, const u8* const pInLimit)
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

// `base`: window base (index i <-> base + i).  Appends packed sequences to seqOut, returns their number; rep[] and
// *lastLL are updated.  Runs on one lane.
ZSA_DEV u32 zsa_parse_dfast(u32* hashLong, u32* hashSmall, const ZsaEncParams& P, const u8* base, u32 dictLimit,
                            const u8* src, u32 srcSize, u32 rep[3], u64* seqOut, u32* lastLL)
{
    u32 const hBitsL = P.hlog, hBitsS = P.clog, mls = P.mml;
    const u8* const istart = src;
    u32 const endIndex = (u32)(istart - base) + srcSize;
    u32 const maxDistance = 1u << P.wlog;
    // ZSTD_getLowestPrefixIndex (zstd_compress_internal.h:1307-1318), no loaded dictionary
    u32 const prefixLowestIndex = (endIndex - dictLimit > maxDistance) ? endIndex - maxDistance : dictLimit;
    const u8* const prefixLowest = base + prefixLowestIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    const u8* anchor = istart;
    const u8* ip = istart + wv_vzero();   // per-lane state: loads stay in flight until they are used (zsa_wave_hip.h)
    u32 off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0, nbSeq = 0;

    ip += (ip == prefixLowest);
    {   u32 const current = (u32)(ip - base);
        u32 const windowLow = (current - dictLimit > maxDistance) ? current - maxDistance : dictLimit;
        u32 const maxRep = current - windowLow;
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
    // candidate bytes: indices at or below the prefix start are never dereferenced by the reference; here they are
    // redirected to the block start (any readable address) and the result is ignored
#define ZSA_CAND(idx) ((idx) > prefixLowestIndex ? base + (idx) : istart)
    u64 pre0 = 0, pre1 = 0, pre2 = 0; bool havePre = false;
    for (;;) {   // one stored match per iteration (:156)
        u32 step = 1;
        const u8* nextStep = ip + 256;     // 1 << kSearchStrength (:125)
        const u8* ip1 = ip + 1;
        u32 curr = 0, offset = 0, mLength = 0, hl1 = 0, skip = 0;
        const u8* mPtr = base;
        int outcome = 0;                   // 0 block exhausted, 1 repcode stored, 2 match at (ip, offset, mLength)
        if (ip1 > ilimit) break;
        // ---- three positions are in flight (p0 being decided, p1, p2); their state lives in three register sets that
        // rotate by renaming (the loop body is written out three times), because a register-to-register copy of a
        // value still in flight would have to wait for it.
        const u8* Pp[3]; u64 Vw[3]; u32 HL[3], HS[3], IL[3], IS[3], MS[3], RR[3]; u64 ML[3];
        Pp[0] = ip; Pp[1] = ip1; Pp[2] = ip1 + 1;   // the step is 1 for the first 256 bytes
        if (havePre) { Vw[0] = pre0; Vw[1] = pre1; Vw[2] = pre2; havePre = false; }   // requested with the post-match loads
        else { Vw[0] = wv_ldg64(Pp[0]); Vw[1] = wv_ldg64(Pp[1]); Vw[2] = wv_ldg64(Pp[2] <= ilimit ? Pp[2] : Pp[1]); }
        HL[0] = zsa_hash(Vw[0], hBitsL, 8); HS[0] = zsa_hash(Vw[0], hBitsS, mls);
        HL[1] = zsa_hash(Vw[1], hBitsL, 8); HS[1] = zsa_hash(Vw[1], hBitsS, mls);
        IL[0] = wv_ldg_cell(hashLong + HL[0]); IS[0] = wv_ldg_cell(hashSmall + HS[0]);
        IL[1] = wv_ldg_cell(hashLong + HL[1]); IS[1] = wv_ldg_cell(hashSmall + HS[1]);   // before the store of p0: patched below
        RR[0] = wv_ldg32(ip + 1 - off1);
        ML[0] = wv_ldg64(ZSA_CAND(IL[0]));
        MS[0] = wv_ldg32(ZSA_CAND(IS[0]));
        HL[2] = HS[2] = IL[2] = IS[2] = MS[2] = RR[1] = RR[2] = MS[1] = 0; ML[1] = ML[2] = 0;
        // one position (:170-223): a = the position decided now, b and c the next two
#define ZSA_DFAST_POS(a, b, c)                                                                                              \
        {   curr = (u32)(Pp[a] - base);                                                                                     \
            wv_stg_cell(hashLong + HL[a], curr); wv_stg_cell(hashSmall + HS[a], curr);                                      \
            if (HL[b] == HL[a]) IL[b] = curr;            /* the cells of p1 were read before this store */                   \
            if (HS[b] == HS[a]) IS[b] = curr;                                                                               \
            /* requests: candidates and repcode bytes of p1, cells of p2 */                                                 \
            ML[b] = wv_ldg64(ZSA_CAND(IL[b]));                                                                              \
            MS[b] = wv_ldg32(ZSA_CAND(IS[b]));                                                                              \
            RR[b] = wv_ldg32(Pp[b] + 1 - off1);                                                                             \
            u32 const stepNext = step + (Pp[b] >= nextStep);             /* :208-213: the step from p1 to p2 */             \
            const u8* const nextStep1 = nextStep + ((Pp[b] >= nextStep) ? 256 : 0);                                         \
            u32 const stepNext2 = stepNext + (Pp[c] >= nextStep1);       /* the step from p2 to p3 */                       \
            const u8* const Pd = Pp[c] + stepNext2;                                                                         \
            bool const has2 = Pp[c] <= ilimit;                                                                              \
            HL[c] = zsa_hash(Vw[c], hBitsL, 8); HS[c] = zsa_hash(Vw[c], hBitsS, mls);                                       \
            IL[c] = wv_ldg_cell(hashLong + (has2 ? HL[c] : HL[b]));                                                         \
            IS[c] = wv_ldg_cell(hashSmall + (has2 ? HS[c] : HS[b]));                                                        \
            /* decide p0 */                                                                                                 \
            if ((off1 > 0) & (RR[a] == (u32)(Vw[a] >> 8))) {   /* :179-184 repcode at ip+1 */                               \
                ip = Pp[a]; ip1 = Pp[b]; hl1 = HL[b];                                                                       \
                outcome = 1;                                                                                                \
                break;                                                                                                      \
            }                                                                                                               \
            if (IL[a] > prefixLowestIndex && ML[a] == Vw[a]) {   /* :188-196 long match at ip */                            \
                ip = Pp[a]; ip1 = Pp[b]; hl1 = HL[b];                                                                       \
                mPtr = base + IL[a]; skip = 8;                                                                              \
                outcome = 2;                                                                                                \
                break;                                                                                                      \
            }                                                                                                               \
            if (IS[a] > prefixLowestIndex && MS[a] == (u32)Vw[a]) {   /* :201-206 -> :237-253 */                            \
                ip = Pp[a]; ip1 = Pp[b]; hl1 = HL[b];                                                                       \
                if (IL[b] > prefixLowestIndex && ML[b] == Vw[b]) { ip = ip1; mPtr = base + IL[b]; skip = 8; }               \
                else { mPtr = base + IS[a]; skip = 4; }                                                                     \
                outcome = 2;                                                                                                \
                break;                                                                                                      \
            }                                                                                                               \
            /* no match here: p3 takes the register set of p0 */                                                            \
            nextStep = nextStep1; step = stepNext;                                                                          \
            Pp[a] = Pd; Vw[a] = wv_ldg64(Pd <= ilimit ? Pd : Pp[b]);                                                        \
            if (!(Pp[c] <= ilimit)) break;                   /* `while (ip1 <= ilimit)` for the new p1 */                   \
        }
        for (;;) {
            ZSA_DFAST_POS(0, 1, 2)
            ZSA_DFAST_POS(1, 2, 0)
            ZSA_DFAST_POS(2, 0, 1)
        }
#undef ZSA_DFAST_POS
        // the match is measured once, outside the position loop
        if (outcome == 1) {
            mLength = zsa_dfast_count(ip + 5, ip + 5 - off1, iend) + 4;
            ip++;
            seqOut[nbSeq++] = zsa_packseq((u32)(ip - anchor), mLength - ZSA_MINMATCH, 1);
        } else if (outcome == 2) {
            u32 bw;
            offset = (u32)(ip - mPtr);
            mLength = zsa_dfast_extend(ip, mPtr, skip, iend, anchor, prefixLowest, &bw) + skip + bw;
        }
        if (outcome == 0) break;
        if (outcome == 2) {   // _match_found (:257-272)
            off2 = off1; off1 = offset;
            if (step < 4) wv_stg_cell(hashLong + hl1, (u32)(ip1 - base));
            seqOut[nbSeq++] = zsa_packseq((u32)(ip - anchor), mLength - ZSA_MINMATCH, offset + 3);
        }
        ip += mLength;   // _match_stored (:274-303)
        anchor = ip;
        if (ip <= ilimit) {
            u32 const indexToInsert = curr + 2;
            u64 const va = wv_ldg64(base + indexToInsert), vb = wv_ldg64(ip - 2);
            u64 vi = wv_ldg64(ip);
            u32 ri = wv_ldg32(ip - off2);
            // the source words the next search starts with, in the same round trip
            pre1 = wv_ldg64(ip + 1 <= ilimit ? ip + 1 : ip); pre2 = wv_ldg64(ip + 2 <= ilimit ? ip + 2 : ip);
            wv_stg_cell(hashLong + zsa_hash(va, hBitsL, 8), indexToInsert);
            wv_stg_cell(hashLong + zsa_hash(vb, hBitsL, 8), (u32)(ip - 2 - base));
            wv_stg_cell(hashSmall + zsa_hash(va, hBitsS, mls), indexToInsert);
            wv_stg_cell(hashSmall + zsa_hash(vb >> 8, hBitsS, mls), (u32)(ip - 1 - base));   // bytes at ip-1 (mls <= 7)
            while ((off2 > 0) & ((u32)vi == ri)) {   // immediate repcode (:290-302); ip <= ilimit holds
                u32 const rLength = zsa_dfast_count(ip + 4, ip + 4 - off2, iend) + 4;
                { u32 const t = off2; off2 = off1; off1 = t; }
                wv_stg_cell(hashSmall + zsa_hash(vi, hBitsS, mls), (u32)(ip - base));
                wv_stg_cell(hashLong + zsa_hash(vi, hBitsL, 8), (u32)(ip - base));
                seqOut[nbSeq++] = zsa_packseq(0, rLength - ZSA_MINMATCH, 1);
                ip += rLength;
                anchor = ip;
                if (ip > ilimit) break;
                vi = wv_ldg64(ip); ri = wv_ldg32(ip - off2);
                pre1 = wv_ldg64(ip + 1 <= ilimit ? ip + 1 : ip); pre2 = wv_ldg64(ip + 2 <= ilimit ? ip + 2 : ip);
            }
            if (ip + 1 <= ilimit) { pre0 = vi; havePre = true; }
        }
    }
#undef ZSA_CAND
    saved2 = ((saved1 != 0) && (off1 != 0)) ? saved1 : saved2;   // _cleanup (:225-235)
    rep[0] = off1 ? off1 : saved1;
    rep[1] = off2 ? off2 : saved2;
    *lastLL = (u32)(iend - anchor);
    return nbSeq;
}
