// snappy_compress_mw.h -- the Snappy raw-format encoder of one buffer by one wavefront, "many matches per window" form (round 3): the
// scheme of lz4_compress_mw.h on M/snappy/SnappyRawCompressor.java:47-232.  A window of 64 consecutive positions is loaded once (lane l:
// the 8 bytes at base + l, their hash, the table entry of that hash as the window found it, the 16 bytes at that entry's position) and
// the Java loop is replayed over it with wave-uniform control: the search (:138-162, skip schedule included: the first 33 probes of a
// search are consecutive positions, then every second one), the copy (:186-198), the `input - 1` insert and the re-probe behind it
// (:199-219) -- lane reads of registers; the table is only read at the start of a window and written at its end, which matters most for the
// wavefronts whose table lives in global memory (snappy_compress.hip, two tiers): one round of table loads per window, not per sequence.
// A search that runs through a whole window goes on in the batch-probe step (snappy_compress_body.h's, in its mode 2 form).
#pragma once
#include "snappy_compress_body.h"

namespace achip {

namespace snmw {
__device__ __forceinline__ uint32_t rl32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int src) { return ((uint64_t)rl32((uint32_t)(v >> 32), src) << 32) | rl32((uint32_t)v, src); }
__device__ __forceinline__ uint64_t bits(int lo, int hi)
{
    const uint64_t upTo = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    const uint64_t below = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
    return upTo & ~below;
}
// bits [lo, hi) for WAVE-UNIFORM lo <= hi with hi - lo <= 63, in one scalar instruction (lz4_compress_mw.h: these encoders' time is their scalar instructions)
__device__ __forceinline__ uint64_t sbits(int lo, int hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t m;
    asm("s_bfm_b64 %0, %1, %2" : "=s"(m) : "s"(hi - lo), "s"(lo));
    return m;
#else
    return bits(lo, hi);
#endif
}
// the lanes of a window that a search starting at lane c probes: c .. c + 32 (its first 33 probes advance by one), then every second lane
__device__ __forceinline__ uint64_t probe_lanes(int c)  // (wave-uniform c in 1 .. 64)
{
    c = uni(c);  // (the compiler's analysis does not always see it)
    uint64_t m = sbits(c, c + 33 < 64 ? c + 33 : 64);
    if (c + 34 < 64) {
        // lanes c + 34, c + 36, ...: same parity as c
        const uint64_t parity = (c & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
        m |= parity & sbits(c + 34, 64);
    }
    return m;
}
}  // namespace snmw

// subFirst / subLimit / outputAt (round 6): the buffer's independent 64 KiB sub-blocks (SnappyRawCompressor.java:93-99: the table is cleared for each, positions are
// relative to it, no copy reaches across) from sub-block subFirst up to, not including, subLimit; outputAt < 0: the stream's start (the length preamble is written,
// the sub-blocks follow it), otherwise the sub-blocks' bytes start at out + outputAt and no preamble is written.  The defaults are the whole buffer.
__device__ __forceinline__ void snappy_compress_buffer_mw(uint16_t* table, const uint8_t* __restrict__ in0, int32_t inLen, uint8_t* __restrict__ out, int32_t outCap, int lane,
                                                          int32_t& stOut, int32_t& outputOut, int32_t subFirst = 0, int32_t subLimit = 0x7FFF, int32_t outputAt = -1)
{
    using namespace snc;
    using namespace snmw;
    inLen = uni(inLen);  // (loaded from the batch arrays: wave-uniform, which the compiler cannot know -- achip_device.h, uni())
    outCap = uni(outCap);
    int32_t st = 0;
    int32_t output = 0;
    const int64_t bound = 32 + (int64_t)inLen + inLen / 6;
    if ((int64_t)outCap < bound) {
        st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_MAX_OUTPUT);
    }
    else {
        if (outputAt < 0) {
            uint32_t n = (uint32_t)inLen;
            int32_t nb = n < (1u << 7) ? 1 : (n < (1u << 14) ? 2 : (n < (1u << 21) ? 3 : (n < (1u << 28) ? 4 : 5)));
            if (lane == 0) {
                for (int i = 0; i < nb; i++) {
                    out[i] = (uint8_t)((n >> (7 * i)) | (i + 1 < nb ? 0x80u : 0u));
                }
            }
            output = nb;
        }
        else {
            output = outputAt;
        }
        const int64_t endAddress = (int64_t)subLimit * BLOCK_SIZE < inLen ? (int64_t)subLimit * BLOCK_SIZE : inLen;
        for (int64_t blockAddress = (int64_t)subFirst * BLOCK_SIZE; blockAddress < endAddress; blockAddress += BLOCK_SIZE) {
            const uint8_t* __restrict__ in = in0 + blockAddress;
            const int32_t blockLimit = (int32_t)((inLen - blockAddress) < BLOCK_SIZE ? (inLen - blockAddress) : BLOCK_SIZE);
            int32_t tableSize = blockLimit <= 1 ? 0 : (int32_t)((0x80000000u >> __builtin_clz((uint32_t)(blockLimit - 1))) << 1);
            tableSize = tableSize < 256 ? 256 : (tableSize > MAX_HASH_TABLE_SIZE ? MAX_HASH_TABLE_SIZE : tableSize);
            wave_mem_order();
            for (int i = lane; i < tableSize; i += 64) {
                table[i] = 0;
            }
            wave_mem_order();
            const int hashBits = 31 - __builtin_clz((uint32_t)tableSize);
            const int32_t shift = 32 - hashBits;
            const int32_t fastInputLimit = blockLimit - INPUT_MARGIN_BYTES;

            int32_t nextEmit = 0;
            int32_t input = 0;
            if (input <= fastInputLimit) {
                int mode = 0;           // 0: block start; 1: after a copy that ended at `input`; 2: a search that ran through a window goes on
                int32_t scanStart = 1;  // (mode 2) position of probe 0 of the search
                int32_t k0 = 0;         // (mode 2) its next probe
                for (;;) {
                    if (mode == 2) {
                        // ---- the batch-probe step of snappy_compress_body.h for a search in progress ----
                        const int32_t k = k0 + lane;
                        const int32_t pos = scanStart + snappy_scan_offset(k);
                        const bool valid = pos + ((32 + k) >> 5) <= fastInputLimit;  // the loop condition of :141
                        const unsigned long long invalidMask = __ballot(!valid);
                        const int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
                        const bool active = lane < firstInvalid;
                        const unsigned long long activeMask = __ballot(active);
                        uint32_t x = 0;
                        int32_t h = 0;
                        int32_t cand = 0;
                        if (active) {
                            x = ld4(in + pos);
                            h = snappy_hash(x, shift);
                            cand = (int32_t)table[h];
                        }
                        const unsigned long long same = wave_match_any14((uint32_t)h, hashBits, activeMask);
                        const unsigned long long earlier = same & ((1ull << lane) - 1ull);
                        {
                            const bool fromBatch = active && earlier != 0;
                            const int32_t latest = __shfl(pos, fromBatch ? 63 - __builtin_clzll(earlier) : lane);
                            if (fromBatch) {
                                cand = latest;
                            }
                        }
                        bool hit = false;
                        if (active) {
                            hit = ld4(in + cand) == x;
                        }
                        const unsigned long long hitMask = __ballot(hit);
                        const int winner = hitMask ? __builtin_ctzll(hitMask) : -1;
                        const int lastWriter = winner >= 0 ? winner : firstInvalid - 1;
                        {
                            const unsigned long long upTo = lastWriter >= 63 ? ~0ull : ((1ull << (lastWriter + 1)) - 1ull);
                            const unsigned long long later = same & upTo & ~((2ull << lane) - 1ull);
                            if (active && lane <= lastWriter && later == 0) {
                                table[h] = (uint16_t)pos;
                            }
                        }
                        wave_mem_order();
                        if (winner < 0) {
                            if (firstInvalid < 64) {
                                break;  // the search ran off the end: what is left is a literal (:160-162)
                            }
                            k0 += 64;
                            continue;
                        }
                        input = uni(__shfl(pos, winner));
                        const int32_t candidate = uni(__shfl(cand, winner));
                        const int32_t literalLength = input - nextEmit;  // :169-175
                        output += snappy_literal_header(out, output, literalLength, lane);
                        group_copy<64>(out + output, in + nextEmit, literalLength, lane);
                        output += literalLength;
                        const int32_t matched = 4 + wave_count(in, input + 4, candidate + 4, blockLimit, lane);
                        output = snappy_emit_copy(out, output, input - candidate, matched, lane);
                        input += matched;
                        nextEmit = input;
                        if (input >= fastInputLimit) {
                            break;  // :194-196
                        }
                        mode = 1;
                        continue;
                    }

                    // ---- a window: 64 consecutive positions from `base`.  After a copy: lane 0 = input - 1 (inserted, never probed), lane 1 =
                    // input (the re-probe), the search from lane 2; at the start of a block: lane 0 = position 0 (neither), the search from lane 1 ----
                    const int32_t base = mode == 0 ? 0 : input - 1;
                    const int32_t pos = base + lane;
                    const bool canLoad = pos + 8 <= blockLimit;
                    uint64_t x = 0;
                    int32_t h = 0;
                    int32_t tc = 0;
                    if (canLoad) {
                        x = ld8(in + pos);
                        h = snappy_hash((uint32_t)x, shift);
                        tc = (int32_t)table[h];
                    }
                    const uint32_t x4 = (uint32_t)x;
                    const unsigned long long loadMask = __ballot(canLoad);
                    const unsigned long long same = wave_match_any14((uint32_t)h, hashBits, loadMask) & loadMask;
                    // the 16 bytes at the table entry's position where the block has them (the candidate's own 4 and 12 behind them), else its 4
                    const bool fast = canLoad && tc + 16 <= blockLimit;
                    uint32_t c4 = 0;
                    uint64_t after8 = 0;
                    if (fast) {
                        const u32x4 a = ld16(in + tc);
                        c4 = a.x;
                        after8 = (uint64_t)a.y | ((uint64_t)a.z << 32);
                    }
                    else if (canLoad) {
                        c4 = ld4(in + tc);
                    }

                    // What a copy of this lane against ITS TABLE ENTRY would measure behind the 4-byte hit, by every lane at once (round 6, as lz4_compress_mw.h): this
                    // window's bytes from lane + 4 against the 8 fetched behind the candidate -- bits 0..3 the count (0 .. 8), bit 8 "usable" (the last four lanes
                    // hold only the first 4 of those bytes: a difference among them is still the count).  The replay reads one word with one lane read.
                    uint32_t facts = 0;
                    {
                        const uint32_t upHi = (uint32_t)__shfl((int32_t)(uint32_t)(x >> 32), lane < 60 ? lane + 4 : lane);
                        const uint64_t dF = ((x >> 32) | ((uint64_t)upHi << 32)) ^ after8;
                        const uint32_t fwd = dF == 0 ? 8u : (uint32_t)(__builtin_ctzll(dF) >> 3);
                        facts = fwd | (fast && pos + 12 <= blockLimit && (lane < 60 || fwd < 4) ? 256u : 0u);
                    }
                    // the copy's positions on the vector side (round 6; vec(), achip_device.h): `nextEmit` and `output` travel through the replay in vector registers, a
                    // copy whose count the registers know is measured and emitted with straight-line vector instructions -- the replay was bound by the CU's one
                    // scalar unit; the scalar variables are brought up to date where scalar code needs them
                    int32_t vNextEmit = vec(nextEmit), vOutput = vec(output);
                    const unsigned long long sameBelow = same & ((1ull << lane) - 1ull);
                    unsigned long long M = mode == 0 ? 0ull : 1ull;  // inserted lanes
                    int c = mode == 0 ? 1 : 2;                      // first lane of the search that follows
                    int r = mode == 0 ? -1 : 1;                     // lane of a pending re-probe, -1: none
                    bool blockDone = false, searchGoesOn = false;
                    int32_t probesDone = 0;
                    for (;;) {
                        int wl = -1;
                        int32_t cand = 0;
                        int jl = -1;
                        bool viaReprobe = false;
                        if (r >= 0) {
                            // :207-219 the table lookup at `input` and its insert; the copy loop goes on while the 4 bytes match
                            const unsigned long long elig = rl64(same, r) & M & sbits(0, r);
                            jl = elig != 0 ? 63 - __builtin_clzll(elig) : -1;
                            const uint32_t xr = rl32(x4, r);
                            const uint32_t cr = jl >= 0 ? rl32(x4, jl) : rl32(c4, r);
                            cand = jl >= 0 ? base + jl : (int32_t)rl32((uint32_t)tc, r);
                            M |= 1ull << r;
                            if (cr == xr) {
                                wl = r;
                                viaReprobe = true;
                            }
                            else {
                                nextEmit = base + r;  // :220 (where it already is: the copy before ended here)
                                vNextEmit = vec(nextEmit);
                                c = r + 1;
                            }
                            r = -1;
                        }
                        if (wl < 0) {
                            // the search :138-162 from lane c: probe t of it sits at lane c + t for t <= 32, then at every second lane
                            const unsigned long long probes = probe_lanes(c);
                            const int d = lane - c;
                            const int32_t kk = d <= 32 ? d : 32 + ((d - 32) >> 1);
                            const bool probing = lane >= c && ((probes >> lane) & 1ull) != 0;
                            const bool canProbe = pos + ((32 + kk) >> 5) <= fastInputLimit;
                            // a lane sees the inserts of the replay so far and of the probing lanes before it (`probes` holds no lane below c, sameBelow none from this lane on)
                            const unsigned long long elig = sameBelow & (M | probes);
                            const int j = elig != 0 ? 63 - __builtin_clzll(elig) : -1;
                            const uint32_t cv = __shfl(x4, j >= 0 ? j : lane);
                            const uint32_t cmp = j >= 0 ? cv : c4;
                            const int32_t cp = j >= 0 ? base + j : tc;
                            const bool hit = probing && canProbe && cmp == x4;
                            const unsigned long long hm = __ballot(hit);
                            const unsigned long long im = __ballot(probing && !canProbe);
                            const unsigned long long first = hm | im;
                            if (first == 0) {
                                M |= probes;
                                probesDone = (int32_t)__popcll(probes);
                                searchGoesOn = true;
                                break;
                            }
                            const int w = __builtin_ctzll(first);
                            if (((im >> w) & 1ull) != 0) {  // the loop condition of :141 fails at lane w: the block ends in a literal
                                M |= probes & sbits(c, w);
                                blockDone = true;
                                break;
                            }
                            M |= probes & sbits(c, w + 1);
                            wl = w;
                            cand = (int32_t)rl32((uint32_t)cp, w);
                            jl = (int)rl32((uint32_t)j, w);
                        }
                        // ---- a copy starts at lane wl against `cand` ----
                        input = base + wl;
                        uint32_t factsW = 0;
                        if (jl < 0) {
                            factsW = rl32(facts, wl);
                        }
                        else if (wl < 60 && input + 12 <= blockLimit) {  // a candidate inside the window: the 8 bytes behind either hit are the words of lanes wl + 4 and jl + 4
                            const uint64_t dF = rl64(x, wl + 4) ^ rl64(x, jl + 4);
                            factsW = (dF == 0 ? 8u : (uint32_t)(__builtin_ctzll(dF) >> 3)) | 256u;
                        }
                        if ((factsW & 256u) != 0) {
                            const int32_t vIn = vec(input), vCand = vec(cand);
                            const uint32_t vF = vec(factsW);
                            if (!viaReprobe) {  // the literal before it :169-175: bytes of this window, stored from the lanes' registers
                                const int32_t vLit = vIn - vNextEmit;
                                vOutput += snappy_literal_header(out, vOutput, vLit, lane);
                                if (pos >= vNextEmit && pos < vIn) {
                                    out[vOutput + (pos - vNextEmit)] = (uint8_t)x4;
                                }
                                vOutput += vLit;
                            }
                            const int32_t vLimitLen = blockLimit - (vIn + 4);
                            const int32_t vFwd = (int32_t)(vF & 15u);
                            int32_t vMatched = 4 + (vFwd < vLimitLen ? vFwd : vLimitLen);
                            if (__ballot(vFwd == 8 && vLimitLen > 8) != 0) {  // (uniform) the registers' 8 bytes all match: memory has the rest
                                vMatched = 12 + vec(wave_count(in, vIn + 12, vCand + 12, blockLimit, lane));
                            }
                            vOutput = snappy_emit_copy(out, vOutput, vIn - vCand, vMatched, lane);
                            vNextEmit = vIn + vMatched;
                            input = uni(vNextEmit);
                            if (input >= fastInputLimit) {  // :194-196
                                blockDone = true;
                                break;
                            }
                            const int rr = input - base;
                            if (rr < 64) {
                                M |= 1ull << (rr - 1);  // :203-205 the `input - 1` insert
                                r = rr;
                                continue;
                            }
                            break;  // the copy ends beyond the window: the next one starts at input - 1
                        }
                        // ---- every other copy: scalar code, as before round 6 ----
                        nextEmit = uni(vNextEmit);
                        output = uni(vOutput);
                        if (!viaReprobe) {  // the literal before it :169-175: bytes of this window, stored from the lanes' registers
                            const int32_t literalLength = input - nextEmit;
                            output += snappy_literal_header(out, output, literalLength, lane);
                            if (pos >= nextEmit && pos < input) {
                                out[output + (pos - nextEmit)] = (uint8_t)x4;
                            }
                            output += literalLength;
                        }
                        int32_t matched;
                        {
                            const int32_t a0 = input + 4, b0 = cand + 4;
                            const int32_t limitLen = blockLimit - a0;
                            const int la = a0 - base, lb = b0 - base;
                            const bool okA = la < 64 && a0 + 8 <= blockLimit;
                            const bool okB = jl >= 0 ? (lb < 64 && b0 + 8 <= blockLimit) : rl32((uint32_t)fast, wl) != 0;
                            if (okA && okB) {
                                const uint64_t a8 = rl64(x, la);
                                const uint64_t b8 = jl >= 0 ? rl64(x, lb) : rl64(after8, wl);
                                const uint64_t dd = a8 ^ b8;
                                int32_t eq = dd == 0 ? 8 : (int32_t)(__builtin_ctzll(dd) >> 3);
                                eq = eq < limitLen ? eq : limitLen;
                                matched = 4 + eq;
                                if (eq == 8 && limitLen > 8) {
                                    matched = 12 + wave_count(in, a0 + 8, b0 + 8, blockLimit, lane);
                                }
                            }
                            else {
                                matched = 4 + wave_count(in, a0, b0, blockLimit, lane);
                            }
                        }
                        output = snappy_emit_copy(out, output, input - cand, matched, lane);
                        input += matched;
                        nextEmit = input;
                        vNextEmit = vec(nextEmit);
                        vOutput = vec(output);
                        if (input >= fastInputLimit) {  // :194-196
                            blockDone = true;
                            break;
                        }
                        const int rr = input - base;
                        if (rr < 64) {
                            M |= 1ull << (rr - 1);  // :203-205 the `input - 1` insert
                            r = rr;
                            continue;
                        }
                        break;  // the copy ends beyond the window: the next one starts at input - 1
                    }
                    nextEmit = uni(vNextEmit);
                    output = uni(vOutput);
                    // the table takes the latest inserted lane of every hash
                    {
                        const unsigned long long later = same & M & ~((2ull << lane) - 1ull);
                        if (canLoad && ((M >> lane) & 1ull) != 0 && later == 0) {
                            table[h] = (uint16_t)pos;
                        }
                    }
                    wave_mem_order();
                    if (blockDone) {
                        break;
                    }
                    if (searchGoesOn) {
                        mode = 2;
                        scanStart = base + c;
                        k0 = probesDone;
                        continue;
                    }
                    mode = 1;
                }
            }
            if (nextEmit < blockLimit) {  // :224-229
                const int32_t literalLength = blockLimit - nextEmit;
                output += snappy_literal_header(out, output, literalLength, lane);
                group_copy<64>(out + output, in + nextEmit, literalLength, lane);
                output += literalLength;
            }
        }
    }
    stOut = st;
    outputOut = output;
}

}  // namespace achip
