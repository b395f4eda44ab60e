// zsa_dfast.h -- the ZSTD_dfast parse without a dictionary (zstd_double_fast.c:98-305): levels 3-4, level 2 at <= 256 KB.
//
// Two tables in the workgroup's global scratch: `hashLong` keyed by 8 bytes (hashLog bits) and `hashSmall` keyed by
// minMatch bytes (chainLog bits).  Per position the candidates are tried in the reference's order: repcode at ip+1,
// long match at ip, then -- only if the short table has a match at ip -- long match at ip+1 before settling for the
// short one.  This first version walks the chain on lane 0 (correctness and byte identity first);
// the frames of a launch still run 16 per CU.  Included by zsa_encode.h.
#pragma once

ZSA_DEV u32 zsa_dfast_catchUp(const u8*& ip, const u8*& match, const u8* anchor, const u8* prefixLowest)
{   // :193, :245, :253
    u32 n = 0;
    while (((ip > anchor) & (match > prefixLowest)) && (ip[-1] == match[-1])) { ip--; match--; n++; }
    return n;
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
    const u8* ip = istart;
    u32 off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0, nbSeq = 0;

    ip += (ip == prefixLowest);
    {   u32 const current = (u32)(ip - base);
        u32 const windowLow = (current - dictLimit > maxDistance) ? current - maxDistance : dictLimit;
        u32 const maxRep = current - windowLow;
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
    for (;;) {   // one stored match per iteration (:156)
        u32 step = 1;
        const u8* nextStep = ip + 256;     // 1 << kSearchStrength (:125)
        const u8* ip1 = ip + 1;
        u32 hl0, hl1 = 0, idxl0, curr = 0, offset = 0, mLength = 0;
        int outcome = 0;                   // 0 block exhausted, 1 repcode stored, 2 match at (ip, offset, mLength)
        if (ip1 > ilimit) break;
        u64 v0 = zsa_ld64(ip);
        hl0 = zsa_hash(v0, hBitsL, 8);
        idxl0 = hashLong[hl0];
        do {   // one position per iteration (:170-223)
            u32 const hs0 = zsa_hash(v0, hBitsS, mls);
            u32 const idxs0 = hashSmall[hs0];
            u64 const v1 = zsa_ld64(ip1);
            curr = (u32)(ip - base);
            hashLong[hl0] = curr; hashSmall[hs0] = curr;
            if ((off1 > 0) & (zsa_ld32(ip + 1 - off1) == zsa_ld32(ip + 1))) {   // :179-184
                mLength = zsa_count(ip + 5, ip + 5 - off1, iend) + 4;
                ip++;
                seqOut[nbSeq++] = zsa_packseq((u32)(ip - anchor), mLength - ZSA_MINMATCH, 1);
                outcome = 1;
                break;
            }
            hl1 = zsa_hash(v1, hBitsL, 8);
            if (idxl0 > prefixLowestIndex && zsa_ld64(base + idxl0) == v0) {   // :188-196
                const u8* m = base + idxl0;
                mLength = zsa_count(ip + 8, m + 8, iend) + 8;
                offset = (u32)(ip - m);
                mLength += zsa_dfast_catchUp(ip, m, anchor, prefixLowest);
                outcome = 2;
                break;
            }
            u32 const idxl1 = hashLong[hl1];
            if (idxs0 > prefixLowestIndex && zsa_ld32(base + idxs0) == (u32)v0) {   // :201-206 -> :237-253
                if (idxl1 > prefixLowestIndex && zsa_ld64(base + idxl1) == v1) {
                    const u8* m = base + idxl1;
                    ip = ip1;
                    mLength = zsa_count(ip + 8, m + 8, iend) + 8;
                    offset = (u32)(ip - m);
                    mLength += zsa_dfast_catchUp(ip, m, anchor, prefixLowest);
                } else {
                    const u8* m = base + idxs0;
                    mLength = zsa_count(ip + 4, m + 4, iend) + 4;
                    offset = (u32)(ip - m);
                    mLength += zsa_dfast_catchUp(ip, m, anchor, prefixLowest);
                }
                outcome = 2;
                break;
            }
            if (ip1 >= nextStep) { step++; nextStep += 256; }   // :208-213
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1;
            v0 = v1;                                             // ip is the old ip1
        } while (ip1 <= ilimit);
        if (outcome == 0) break;
        if (outcome == 2) {   // _match_found (:257-272)
            off2 = off1; off1 = offset;
            if (step < 4) hashLong[hl1] = (u32)(ip1 - base);
            seqOut[nbSeq++] = zsa_packseq((u32)(ip - anchor), mLength - ZSA_MINMATCH, offset + 3);
        }
        ip += mLength;   // _match_stored (:274-303)
        anchor = ip;
        if (ip <= ilimit) {
            u32 const indexToInsert = curr + 2;
            u64 const va = zsa_ld64(base + indexToInsert), vb = zsa_ld64(ip - 2);
            hashLong[zsa_hash(va, hBitsL, 8)] = indexToInsert;
            hashLong[zsa_hash(vb, hBitsL, 8)] = (u32)(ip - 2 - base);
            hashSmall[zsa_hash(va, hBitsS, mls)] = indexToInsert;
            hashSmall[zsa_hash(vb >> 8, hBitsS, mls)] = (u32)(ip - 1 - base);   // bytes at ip-1 (mls <= 7)
            while ((ip <= ilimit) && ((off2 > 0) & (zsa_ld32(ip) == zsa_ld32(ip - off2)))) {
                u32 const rLength = zsa_count(ip + 4, ip + 4 - off2, iend) + 4;
                { u32 const t = off2; off2 = off1; off1 = t; }
                u64 const vi = zsa_ld64(ip);
                hashSmall[zsa_hash(vi, hBitsS, mls)] = (u32)(ip - base);
                hashLong[zsa_hash(vi, hBitsL, 8)] = (u32)(ip - base);
                seqOut[nbSeq++] = zsa_packseq(0, rLength - ZSA_MINMATCH, 1);
                ip += rLength;
                anchor = ip;
            }
        }
    }
    saved2 = ((saved1 != 0) && (off1 != 0)) ? saved1 : saved2;   // _cleanup (:225-235)
    rep[0] = off1 ? off1 : saved1;
    rep[1] = off2 ? off2 : saved2;
    *lastLL = (u32)(iend - anchor);
    return nbSeq;
}
