// zsa_dict_hi.h -- the ZSTD_dfast parse against an ATTACHED dictionary (zstd_double_fast.c:309-532
// ZSTD_compressBlock_doubleFast_dictMatchState_generic): payloads of up to 16 KiB compressed with a CDict of level 3 or 4
// (NewBulkProcessor(dict, level), zstd_bulk.go:33-58; attach cut-off zstd_compress.c:2241-2252).
//
// Unlike the parser without a dictionary (zsa_dfast.h), the reference's loop here is the plain one: one position per
// iteration, candidates in the order  repcode(ip+1)  ->  long(ip): the frame's table if its cell lies in the payload, else
// the dictionary's tagged long table  ->  short(ip), likewise; a short hit first tries long(ip+1) (frame, else dictionary)
// and takes the short match only if that fails.  The wave runs the chain in uniform control flow: every lane holds the
// same position, the four cells of a position (frame long/short, dictionary long/short) are requested together, lane 0
// stores, and the one step with real width -- measuring a match forwards and backwards across the dictionary / payload
// seam -- uses all 64 lanes in the joint index space of zsa_parse_dict.h (64 bytes per round trip).
// Included by zsa_encode.h.
#pragma once

// Parses one block.  hashLong / hashSmall: the frame's own tables (global scratch, zeroed).  Returns the number of sequences
// appended to seqOut; rep[0], rep[1] are updated, rep[2] receives the last literal run.  All values are wave-uniform.
ZSA_DEVFN u32 zsa_parse_dfast_dms(u32* hashLong, u32* hashSmall, u32 hBitsL, u32 hBitsS, u32 mls, const u8* wbase, u32 wDictLimit,
                                  const ZsaCDictDev* dms, const u8* src, u32 srcSize, u32* rep, u64* seqOut)
{
    int const lane = wv_lane();
    hBitsL = wv_uniform(hBitsL); hBitsS = wv_uniform(hBitsS); mls = wv_uniform(mls); srcSize = wv_uniform(srcSize);
    const u8* const base = (const u8*)wv_uniform64((u64)wbase);
    u32 const prefix = wv_uniform(wDictLimit);                  // prefixLowestIndex: loadedDictEnd != 0 with an attached dictionary (:325)
    dms = (const ZsaCDictDev*)wv_uniform64((u64)dms);
    seqOut = (u64*)wv_uniform64((u64)seqOut);
    hashLong = (u32*)wv_uniform64((u64)hashLong); hashSmall = (u32*)wv_uniform64((u64)hashSmall);
    src = (const u8*)wv_uniform64((u64)src);
    const u32* const dictLong = dms->hashTable;
    const u32* const dictSmall = dms->chainTable;
    u32 const dictHBitsL = dms->cp.hlog + 8, dictHBitsS = dms->cp.clog + 8;   // 8 tag bits (:340-341)
    u32 const dictEndIndex = 2 + dms->contentSize;
    u32 const delta = prefix - dictEndIndex;                     // dictIndexDelta (:339)
    ZsaDmsView V;
    V.base = base; V.dbase = (dms->content - 2) - delta; V.prefix = prefix; V.dictLow = 2 + delta;
    u32 const istart = (u32)(src - base), iend = istart + srcSize;
    u32 ip = istart, anchor = istart;
    u32 off1 = wv_uniform(rep[0]), off2 = wv_uniform(rep[1]), nbSeq = 0;
    ip += ((istart - prefix) + dms->contentSize == 0);          // :357
    if (srcSize >= 8) {
        u32 const ilimit = iend - 8;
        while (ip < ilimit) {   // :365
            u64 const v = wv_ldg64(base + ip);
            u32 const h2 = zsa_hash(v, hBitsL, 8), h = zsa_hash(v, hBitsS, mls);
            u32 const dhL = zsa_hash(v, dictHBitsL, 8), dhS = zsa_hash(v, dictHBitsS, mls);
            u32 const dcL = wv_ldg_cell(dictLong + (dhL >> 8)), dcS = wv_ldg_cell(dictSmall + (dhS >> 8));
            u32 const matchIndexL = wv_ldg_cell(hashLong + h2);
            u32 matchIndexS = wv_ldg_cell(hashSmall + h);
            u32 const curr = ip;
            u32 const repIndex = curr + 1 - off1;
            wv_wave_fence();
            if (lane == 0) { wv_stg_cell(hashLong + h2, curr); wv_stg_cell(hashSmall + h, curr); }   // :385
            wv_wave_fence();
            u32 mLength = 0, offset = 0, start = ip;
            int found = 0;   // 1: repcode (stored with offBase 1), 2: match at (start, offset, mLength)
            // :388-395 repcode at ip + 1
            if ((u32)((prefix - 1u) - repIndex) >= 3u && zsa_dv32(V, repIndex) == (u32)(v >> 8)) {
                mLength = zsa_dv_count(V, ip + 1 + 4, repIndex + 4, iend, lane) + 4;
                start = ip + 1;
                found = 1;
            }
            u32 mJoint = 0;   // the match's index in the joint space
            if (!found) {
                if (matchIndexL > prefix) {                      // :397-404 long match in the payload
                    if (wv_ldg64(base + matchIndexL) == v) { mJoint = matchIndexL; found = 2; }
                } else if ((dcL & 0xFF) == (dhL & 0xFF)) {       // :405-416 long match in the dictionary
                    u32 const d = dcL >> 8;
                    if (d > 2u && wv_ldg64(dms->content - 2 + d) == v) { mJoint = d + delta; found = 2; }
                }
                if (found) {
                    u32 fwd, bwd;
                    // (zsa_dv_extend measures forwards from +4 of what it is given: called 4 further on, it starts at +8; its
                    // backward part then first sees the 4 bytes in front of ip + 4, which are known to agree)
                    zsa_dv_extend(V, ip + 4, mJoint + 4, iend, anchor, mJoint >= prefix ? prefix : V.dictLow, &fwd, &bwd, lane);
                    mLength = 8 + fwd; offset = ip - mJoint;
                    u32 const back = bwd - 4;
                    start = ip - back; mLength += back;
                }
            }
            if (!found) {
                bool shortFound = false;
                u32 sJoint = 0;
                if (matchIndexS > prefix) {                      // :418-422
                    if (wv_ldg32(base + matchIndexS) == (u32)v) { shortFound = true; sJoint = matchIndexS; }
                } else if ((dcS & 0xFF) == (dhS & 0xFF)) {       // :423-431
                    u32 const d = dcS >> 8;
                    if (d > 2u && wv_ldg32(dms->content - 2 + d) == (u32)v) { shortFound = true; sJoint = d + delta; }
                }
                if (!shortFound) { ip += ((ip - anchor) >> 8) + 1; continue; }   // :433-437
                {   // _search_next_long (:439-468)
                    u64 const v1 = wv_ldg64(base + ip + 1);
                    u32 const hl3 = zsa_hash(v1, hBitsL, 8), dh3 = zsa_hash(v1, dictHBitsL, 8);
                    u32 const matchIndexL3 = wv_ldg_cell(hashLong + hl3);
                    u32 const dc3 = wv_ldg_cell(dictLong + (dh3 >> 8));
                    wv_wave_fence();
                    if (lane == 0) wv_stg_cell(hashLong + hl3, curr + 1);
                    wv_wave_fence();
                    if (matchIndexL3 > prefix) {
                        if (wv_ldg64(base + matchIndexL3) == v1) { mJoint = matchIndexL3; found = 2; }
                    } else if ((dc3 & 0xFF) == (dh3 & 0xFF)) {
                        u32 const d = dc3 >> 8;
                        if (d > 2u && wv_ldg64(dms->content - 2 + d) == v1) { mJoint = d + delta; found = 2; }
                    }
                    if (found) {
                        u32 fwd, bwd;
                        u32 const p1 = ip + 1;
                        zsa_dv_extend(V, p1 + 4, mJoint + 4, iend, anchor, mJoint >= prefix ? prefix : V.dictLow, &fwd, &bwd, lane);
                        mLength = 8 + fwd; offset = p1 - mJoint;
                        u32 const back = bwd - 4;
                        start = p1 - back; mLength += back;
                    }
                }
                if (!found) {   // :470-479 the short match
                    u32 fwd, bwd;
                    zsa_dv_extend(V, ip, sJoint, iend, anchor, sJoint >= prefix ? prefix : V.dictLow, &fwd, &bwd, lane);
                    mLength = 4 + fwd; offset = ip - sJoint;
                    start = ip - bwd; mLength += bwd;
                    found = 2;
                }
            }
            if (found == 2) { off2 = off1; off1 = offset; }   // :481-485
            if (lane == 0) seqOut[nbSeq] = zsa_packseq(start - anchor, mLength - ZSA_MINMATCH, found == 2 ? offset + 3 : 1u);
            nbSeq++;
            // _match_stored (:487-523)
            ip = start + mLength;
            anchor = ip;
            if (ip <= ilimit) {
                u32 const indexToInsert = curr + 2;
                u64 const va = wv_ldg64(base + indexToInsert), vb = wv_ldg64(base + ip - 2), vc = wv_ldg64(base + ip - 1);
                wv_wave_fence();
                if (lane == 0) {   // program order: a later store to the same cell wins, as in the reference
                    wv_stg_cell(hashLong + zsa_hash(va, hBitsL, 8), indexToInsert);
                    wv_stg_cell(hashLong + zsa_hash(vb, hBitsL, 8), ip - 2);
                    wv_stg_cell(hashSmall + zsa_hash(va, hBitsS, mls), indexToInsert);
                    wv_stg_cell(hashSmall + zsa_hash(vc, hBitsS, mls), ip - 1);
                }
                wv_wave_fence();
                while (ip <= ilimit) {   // immediate repcode (:503-522)
                    u32 const repIndex2 = ip - off2;
                    if (!((u32)((prefix - 1u) - repIndex2) >= 3u)) break;
                    u64 const vi = wv_ldg64(base + ip);
                    if (zsa_dv32(V, repIndex2) != (u32)vi) break;
                    u32 const repLength2 = zsa_dv_count(V, ip + 4, repIndex2 + 4, iend, lane) + 4;
                    { u32 const t = off2; off2 = off1; off1 = t; }
                    if (lane == 0) seqOut[nbSeq] = zsa_packseq(0, repLength2 - ZSA_MINMATCH, 1);
                    nbSeq++;
                    wv_wave_fence();
                    if (lane == 0) { wv_stg_cell(hashSmall + zsa_hash(vi, hBitsS, mls), ip); wv_stg_cell(hashLong + zsa_hash(vi, hBitsL, 8), ip); }
                    wv_wave_fence();
                    ip += repLength2;
                    anchor = ip;
                }
            }
        }
    }
    if (lane == 0) { rep[0] = off1; rep[1] = off2; rep[2] = iend - anchor; }
    return nbSeq;
}
