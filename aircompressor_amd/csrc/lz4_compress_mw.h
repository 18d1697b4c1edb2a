// lz4_compress_mw.h -- the LZ4 block encoder of one wavefront, "many matches per window" form (round 3; lz4_compress.hip has the
// design notes of the family, lz4_compress_body.h the batch-probe encoder this one grew out of and still uses for long searches).
//
// Bit-exactness with M/lz4/Lz4RawCompressor.java:69-192 makes the parse one serial chain per block, and round 2's profile said what a
// link of that chain costs on a wavefront: ~3.2 us per sequence on text -- five DEPENDENT memory round trips (the probes' bytes, the
// candidates' bytes, the catch-up bytes, the bytes of `count`, the literal copy's loads), each waited for in turn, for 12 bytes of
// progress.  The batch-probe encoder evaluates 64 probe positions at once but throws the batch away at its first match, which on
// text is lane 3 .. 6.
//
// Here a WINDOW of 64 consecutive positions is loaded once -- lane l holds the 8 bytes at base + l, their hash, the table entry of
// that hash as it was when the window began, and 16 bytes around that entry's position ([entry - 4, entry + 12): the candidate's own 4
// bytes, 4 before it for the catch-up, 8 behind it for `count`) -- and then the Java loop is REPLAYED over the window with wave-uniform
// control and no memory access at all in the common case:
//   * what the table would say at a lane is known without touching the table: the latest lane of the window with the same hash that
//     the replay has inserted so far (a 64-bit mask per lane from wave_match_any, intersected with the mask of inserted lanes), else
//     the entry read at the window's start; a candidate inside the window is another lane's register;
//   * search (:113-138): every remaining lane evaluates its hit test at once, the first hit is the match; the lanes before it are
//     marked inserted, the lanes inside the match never are;
//   * catch-up (:141-144), count (:240-267) on the first 8 bytes, the re-probe behind the match (:157-184) with its `input - 2`
//     insert: lane reads (readlane) of registers; only a match longer than 12 bytes or a catch-up beyond 4 goes to memory;
//   * literals are stored straight from the lanes' registers (a literal byte of the window is the low byte of its lane's 8), the
//     token / offset / length bytes by lane 0: stores only, nothing to wait for;
//   * at the end of the window the table takes the latest inserted lane of every hash.
// A window therefore costs two memory round trips (its own bytes, the candidates' surroundings) plus one LDS round trip, whatever the
// number of sequences in it -- about five on text.  A search that runs through a whole window without a match (incompressible data:
// the skip schedule takes over after 64 probes) is handed to the batch-probe loop below, which comes back here after its match.
#pragma once
#include "lz4_compress_body.h"

#if defined(ACHIP_HOST_STATS)  // (tools/hostemu, a counting build: which way the replay's sequences go -- tools/hostemu/lz4_paths.py)
extern "C" long long g_zc_stats[32];
#define MWC(k) do { if (lane == 0) g_zc_stats[k]++; } while (0)
#else
#define MWC(k)
#endif
namespace achip {

namespace lz4mw {
__device__ __forceinline__ uint32_t rl32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int src) { return ((uint64_t)rl32((uint32_t)(v >> 32), src) << 32) | rl32((uint32_t)v, src); }
// bits [lo, hi) of a 64-bit mask (0 <= lo, hi <= 64)
__device__ __forceinline__ uint64_t bits(int lo, int hi)
{
    const uint64_t upTo = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    const uint64_t below = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
    return upTo & ~below;
}
// The same for WAVE-UNIFORM lo <= hi with hi - lo <= 63 (every call site below: the masks of the replay), in one scalar instruction: `bits` compiles to a
// dozen (two 64-bit shifts with their >= 64 cases, subtractions, selects), twice per sequence, in a kernel whose time IS its scalar instructions -- 6.0e10
// of them against 2.1e10 vector instructions per launch on the corpus batch, one scalar unit per CU (profiles/r06_counters_lz4_compress_corpus.txt).
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
// 8 bytes starting at byte `off` (0 .. 8) of the 16 bytes (lo, hi)
__device__ __forceinline__ uint64_t ext64(uint64_t lo, uint64_t hi, int off)
{
    return off == 0 ? lo : (off >= 8 ? hi : ((lo >> (8 * off)) | (hi << (64 - 8 * off))));
}
}  // namespace lz4mw

template <typename TableT>
__device__ int32_t lz4_compress_block_mw(const uint8_t* __restrict__ in, int32_t inLen, uint8_t* __restrict__ out, int32_t outCap, TableT* table, int lane, int32_t& stOut)
{
    using namespace lz4c;
    using namespace lz4mw;
    inLen = uni(inLen);  // (loaded from the batch arrays: wave-uniform, which the compiler cannot know -- achip_device.h, uni())
    outCap = uni(outCap);
    int32_t st = 0;
    int32_t output = 0;
    const int64_t bound = (int64_t)inLen + inLen / 255 + 16;
    if ((uint32_t)inLen > 0x7E000000u) {
        st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_LZ4_MAX_INPUT);
    }
    else if ((int64_t)outCap < bound) {
        st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_MAX_OUTPUT);
    }
    else {
        int32_t tableSize = inLen <= 1 ? 0 : (int32_t)((0x80000000u >> __builtin_clz((uint32_t)(inLen - 1))) << 1);
        tableSize = tableSize < MIN_TABLE_SIZE ? MIN_TABLE_SIZE : (tableSize > MAX_TABLE_SIZE ? MAX_TABLE_SIZE : tableSize);
        for (int i = lane; i < tableSize; i += 64) {
            table[i] = 0;
        }
        wave_sync();  // (the accesses of this wavefront to its table, in order; it may share its workgroup with others: lz4_compress_tiers_kernel)
        const int32_t mask = tableSize - 1;
        const int hashBits = 32 - __builtin_clz((uint32_t)mask | 1u);
        const int32_t inputLimit = inLen;
        const int32_t matchFindLimit = inputLimit - MATCH_FIND_LIMIT;
        const int32_t matchLimit = inputLimit - LAST_LITERAL_SIZE;
        int32_t anchor = 0;

        // emitMatch :209-235 behind a literal run already in place; returns the new output position
        auto emit_match = [&](int32_t tokenPos, int32_t literalLength, int32_t offset, int32_t matchLength) {
            if (lane == 0) {
                lz4_write_run_length(out, tokenPos, literalLength, matchLength >= ML_MASK ? ML_MASK : (uint32_t)matchLength);
                out[output] = (uint8_t)offset;
                out[output + 1] = (uint8_t)((uint32_t)offset >> 8);
                if (matchLength >= ML_MASK) {
                    int32_t o = output + 2;
                    int32_t remaining = matchLength - ML_MASK;
                    while (remaining >= 510) {
                        out[o++] = 255;
                        out[o++] = 255;
                        remaining -= 510;
                    }
                    if (remaining >= 255) {
                        out[o++] = 255;
                        remaining -= 255;
                    }
                    out[o++] = (uint8_t)remaining;
                }
            }
            output += 2;
            if (matchLength >= ML_MASK) {
                output += 1 + (matchLength - ML_MASK) / 255;
            }
        };

        // the same with every argument in a vector register (see vec(), achip_device.h): returns the output position behind the match
        auto emit_match_v = [&](int32_t tokenPos, int32_t literalLength, int32_t offset, int32_t matchLength, int32_t at) -> int32_t {
            if (lane == 0) {
                lz4_write_run_length(out, tokenPos, literalLength, matchLength >= ML_MASK ? ML_MASK : (uint32_t)matchLength);
                out[at] = (uint8_t)offset;
                out[at + 1] = (uint8_t)((uint32_t)offset >> 8);
                if (matchLength >= ML_MASK) {
                    int32_t o = at + 2;
                    int32_t remaining = matchLength - ML_MASK;
                    while (remaining >= 510) {
                        out[o++] = 255;
                        out[o++] = 255;
                        remaining -= 510;
                    }
                    if (remaining >= 255) {
                        out[o++] = 255;
                        remaining -= 255;
                    }
                    out[o++] = (uint8_t)remaining;
                }
            }
            return at + 2 + (matchLength >= ML_MASK ? 1 + (int32_t)((uint32_t)(matchLength - ML_MASK) / 255u) : 0);
        };

        if (inLen >= MIN_LENGTH) {
            // mode 0: block start (position 0 is inserted, the search starts at 1); mode 1: after a match that ended at `input`;
            // mode 2: a search that has run through a window goes on (probe k0 of the search that began at scanStart comes next)
            int mode = 0;
            int32_t input = 0;
            int32_t scanStart = 1;
            int32_t k0 = 0;
            for (;;) {
                if (mode == 2) {
                    // ---- the batch-probe step of lz4_compress_body.h for a search in progress: 64 probes at the positions of the skip schedule ----
                    const int32_t k = k0 + lane;
                    const int32_t pos = scanStart + lz4_scan_offset(k);
                    const bool valid = pos + lz4_scan_advance(k) <= matchFindLimit;
                    const unsigned long long invalidMask = __ballot(!valid);
                    const int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
                    const bool active = lane < firstInvalid;
                    const unsigned long long activeMask = __ballot(active);
                    uint64_t x = 0;
                    int32_t h = 0;
                    int32_t cand = 0;
                    if (active) {
                        x = ld8(in + pos);
                        h = lz4_hash(x, mask);
                        cand = (int32_t)table[h];
                    }
                    const unsigned long long same = wave_match_any((uint32_t)h, hashBits, activeMask);
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
                        hit = ld4(in + cand) == (uint32_t)x && cand + MAX_DISTANCE >= pos;
                    }
                    const unsigned long long hitMask = __ballot(hit);
                    const int winner = hitMask ? __builtin_ctzll(hitMask) : -1;
                    const int lastWriter = winner >= 0 ? winner : firstInvalid - 1;
                    {
                        const unsigned long long upTo = lastWriter >= 63 ? ~0ull : ((1ull << (lastWriter + 1)) - 1ull);
                        const unsigned long long later = same & upTo & ~((2ull << lane) - 1ull);
                        if (active && lane <= lastWriter && later == 0) {
                            table[h] = (TableT)pos;
                        }
                    }
                    wave_sync();  // (the accesses of this wavefront to its table, in order; it may share its workgroup with others: lz4_compress_tiers_kernel)
                    if (winner < 0) {
                        if (firstInvalid < 64) {
                            break;  // the search ran off the end: last literals from anchor
                        }
                        k0 += 64;
                        continue;
                    }
                    MWC(25);
                    input = (int32_t)rl32((uint32_t)pos, winner);
                    int32_t matchIndex = (int32_t)rl32((uint32_t)cand, winner);
                    int32_t room = input - anchor < matchIndex ? input - anchor : matchIndex;  // catch up :141-144
                    while (room > 0) {
                        const bool eq = lane < room && in[input - 1 - lane] == in[matchIndex - 1 - lane];
                        const unsigned long long ne = ~__ballot(eq);
                        const int run = ne ? __builtin_ctzll(ne) : 64;
                        input -= run;
                        matchIndex -= run;
                        room -= run;
                        if (run < 64) {
                            break;
                        }
                    }
                    const int32_t literalLength = input - anchor;
                    const int32_t tokenPos = output;
                    const int32_t litPos = tokenPos + lz4_run_length_size(literalLength);
                    group_copy<64>(out + litPos, in + anchor, literalLength, lane);
                    output = litPos + literalLength;
                    const int32_t matchLength = wave_count(in, input + MIN_MATCH, matchIndex + MIN_MATCH, matchLimit, lane);
                    emit_match(tokenPos, literalLength, input - matchIndex, matchLength);
                    input += matchLength + MIN_MATCH;
                    anchor = input;
                    if (input > matchFindLimit) {
                        break;
                    }
                    mode = 1;
                    continue;
                }

                // ---- a window: 64 consecutive positions from `base`; lane 0 (position 0, or input - 2) is inserted, never probed ----
                MWC(26);
                const int32_t base = mode == 0 ? 0 : input - 2;
                const int32_t pos = base + lane;
                const bool canLoad = pos + 8 <= inLen;
                const bool canProbe = pos + 1 <= matchFindLimit;  // a search probe needs its successor inside the limit (:127-129)
                uint64_t x = 0;
                int32_t h = 0;
                int32_t tc = 0;
                if (canLoad) {
                    x = ld8(in + pos);
                    h = lz4_hash(x, mask);
                    tc = (int32_t)table[h];
                }
                const uint32_t x4 = (uint32_t)x;
                const unsigned long long loadMask = __ballot(canLoad);
                const unsigned long long same = wave_match_any((uint32_t)h, hashBits, loadMask) & loadMask;
                // the 16 bytes around the table entry: [tc - 4, tc + 12) where there is room for them, else only the entry's own 4 bytes
                // (fast == false: such a lane's match, if it wins, is measured from memory like the batch-probe encoder does)
                const int32_t tcLo = tc >= 4 ? tc - 4 : 0;
                const int shift = tc - tcLo;
                const bool fast = canLoad && tcLo + 16 <= inLen;
                uint64_t rLo = 0, rHi = 0;
                uint32_t c4 = 0;
                if (fast) {
                    const u32x4 a = ld16(in + tcLo);
                    rLo = (uint64_t)a.x | ((uint64_t)a.y << 32);
                    rHi = (uint64_t)a.z | ((uint64_t)a.w << 32);
                    c4 = (uint32_t)ext64(rLo, rHi, shift);
                }
                else if (canLoad) {
                    c4 = ld4(in + tc);
                }

                // What a match of this lane against ITS TABLE ENTRY would be, measured by every lane at once (round 6; zstd_dfast_mw.h does the same): the
                // equal bytes behind the 4-byte hit as far as the registers go (0 .. 8: this window's bytes from lane + 4 against the 8 fetched behind the
                // candidate) and before it (0 .. 4: lane - 4's bytes against the 4 fetched before the candidate) -- the replay below reads one packed word
                // with one v_readlane where it used to take the 16 candidate bytes apart in scalar code, per sequence.
                // bits 0..3 forward count, 4..6 backward count, 8 the forward count is usable (lane + 4 inside the window, 8 bytes there inside the input),
                // 9 the backward count is usable (lane >= 4)
                uint32_t facts = 0;
                {
                    const uint32_t upHi = (uint32_t)__shfl((int32_t)(uint32_t)(x >> 32), lane < 60 ? lane + 4 : lane);  // bytes [pos + 4, pos + 8) are x's upper half, [pos + 8, pos + 12) lane + 4's
                    const uint64_t my8 = (x >> 32) | ((uint64_t)upHi << 32);                                // [pos + 4, pos + 12)
                    const uint64_t cand8 = ext64(rLo, rHi, shift + 4);                                        // [tc + 4, tc + 12)
                    const uint64_t dF = my8 ^ cand8;
                    const uint32_t fwd = dF == 0 ? 8u : (uint32_t)(__builtin_ctzll(dF) >> 3);
                    // (the window's last four lanes hold only the first 4 of those 8 bytes: a difference among them is still the count)
                    const bool fwdOk = fast && pos + 12 <= inLen && (lane < 60 || fwd < 4);
                    // [pos - 4, pos), last byte in the top bits.  The window's first lanes have only the bytes from `base` on -- which is all a catch-up from there
                    // may use: it stops at `anchor`, and anchor >= base whenever a window is replayed (the unknown low bytes cannot raise a count above that room)
                    const uint32_t x0 = (uint32_t)__shfl((int32_t)x4, 0);
                    const uint32_t farBefore = (uint32_t)__shfl((int32_t)x4, lane >= 4 ? lane - 4 : lane);
                    const uint32_t before = lane >= 4 ? farBefore : (lane == 0 ? 0u : x0 << (8 * (4 - lane)));
                    const uint32_t candBefore = shift == 4 ? (uint32_t)rLo : (uint32_t)((uint32_t)rLo << (8 * (4 - shift)));
                    const uint32_t dB = before ^ candBefore;
                    const uint32_t bwd = dB == 0 ? 4u : (uint32_t)(__builtin_clz(dB) >> 3);
                    const bool bwdOk = fast;
                    facts = fwd | (bwd << 4) | (fwdOk ? 256u : 0u) | (bwdOk ? 512u : 0u);
                }
                const unsigned long long belowMe = (1ull << lane) - 1ull;
                const unsigned long long sameBelow = same & belowMe;

                // the sequence's positions on the vector side (round 6): `anchor` and `output` travel through the replay in vector registers, the common
                // sequence -- its candidate the table's entry, both of its counts in `facts`, nothing for memory to settle -- is measured and emitted with
                // straight-line vector instructions; the scalar variables are brought up to date where scalar code needs them
                int32_t vAnchor = vec(anchor), vOutput = vec(output);
                unsigned long long M = 1ull;     // inserted lanes
                int c = mode == 0 ? 1 : 3;      // first lane of the search that follows
                int r = mode == 0 ? -1 : 2;     // lane of a pending re-probe (the position right behind a match), -1: none
                bool blockDone = false;          // the search ran off the end, or a match ended beyond matchFindLimit
                bool searchGoesOn = false;       // the window is used up in the middle of a search
                for (;;) {
                    // ONE step finds the next match (round 6): the immediate re-probe :171-176 at lane r -- the position right behind a match, the table as the
                    // replay has left it -- is the first probing lane of the search :113-138 over the lanes behind it; its test is the search lanes' test (the
                    // same 4 bytes, the same distance rule), a hit there is the match without literals.  Until round 6 the re-probe was a block of scalar
                    // code of its own, ~65 scalar instructions per sequence in a kernel bound by its scalar instructions.
                    // (Its position is at most matchFindLimit -- the check behind the match before it -- so it always probes; a search lane needs its
                    // successor inside the limit: canProbe.)
                    const int p0 = r >= 0 ? r : c;  // the first probing lane
                    if (r >= 0) {
                        c = r + 1;                  // where the search proper starts (its probe 0: the skip schedule counts from here if it runs through the window)
                    }
                    // a lane sees the inserts of the replay so far and of the probing lanes before it
                    const unsigned long long elig = sameBelow & (M | ~sbits(0, p0));  // (= same & (M | bits(p0, lane)) & bits(0, lane))
                    const int j = elig != 0 ? 63 - __builtin_clzll(elig) : -1;
                    const uint32_t cv = __shfl(x4, j >= 0 ? j : lane);
                    const uint32_t cmp = j >= 0 ? cv : c4;
                    const int32_t cp = j >= 0 ? base + j : tc;
                    const bool probing = lane >= p0;
                    const bool mayProbe = canProbe || lane == r;
                    const bool hit = probing && mayProbe && cmp == x4 && (j >= 0 || cp + MAX_DISTANCE >= pos);
                    const unsigned long long hm = __ballot(hit);
                    const unsigned long long im = __ballot(probing && !mayProbe);
                    const unsigned long long first = hm | im;
                    if (first == 0) {
                        M |= sbits(p0, 64);
                        searchGoesOn = true;
                        break;
                    }
                    const int w = __builtin_ctzll(first);
                    if (((im >> w) & 1ull) != 0) {  // the probe at lane w would step beyond matchFindLimit: the block ends in literals
                        M |= sbits(p0, w);
                        blockDone = true;
                        break;
                    }
                    M |= sbits(p0, w + 1);
                    const int wl = w;                                    // lane where the match starts
                    int32_t cand = (int32_t)rl32((uint32_t)cp, w);      // its candidate's position
                    const int jl = (int)rl32((uint32_t)j, w);           // ... as a lane of this window (-1: the table's entry)
                    const bool zeroLit = w == r;
                    r = -1;
                    // ---- a match starts at lane wl against `cand` (lane jl of the window, or the table's entry of lane wl) ----
                    input = base + wl;
                    // the table's entry as the candidate (the common case): what this lane measured before the replay began, one lane read
                    uint32_t factsW = 0;
                    if (jl < 0) {
                        factsW = rl32(facts, wl);
                    }
                    else if (jl >= 4 && wl < 60 && input + 12 <= inLen) {
                        // a candidate inside the window (another lane's position): the same two counts from the lanes' registers -- the 8 bytes behind either 4-byte
                        // hit are the words of lanes wl + 4 and jl + 4, the 4 before them the low words of lanes wl - 4 and jl - 4 (wl > jl >= 4)
                        const uint64_t dF = rl64(x, wl + 4) ^ rl64(x, jl + 4);
                        const uint32_t dB = rl32(x4, wl - 4) ^ rl32(x4, jl - 4);
                        factsW = (dF == 0 ? 8u : (uint32_t)(__builtin_ctzll(dF) >> 3)) | ((dB == 0 ? 4u : (uint32_t)(__builtin_clz(dB) >> 3)) << 4) | 768u;
                    }
                    const bool quick = (factsW & 768u) == 768u;  // both counts usable (which includes `fast` for a table entry)
                    MWC(20);
                    if (jl >= 0) MWC(23);
                    if (!quick) MWC(24);
                    if (quick) {
                        const int32_t vIn0 = vec(input), vCand0 = vec(cand);
                        const uint32_t vF = vec(factsW);
                        // catch up :141-144 (not behind a re-probe's hit :171-176), then count :240-267 -- `back` bytes of the 4-byte hit itself lie behind the new
                        // start + 4, then what was measured behind the hit
                        const int32_t vRoom = (vIn0 - vAnchor) < vCand0 ? (vIn0 - vAnchor) : vCand0;
                        const int32_t vT = (int32_t)((vF >> 4) & 7u);
                        int32_t vBack = vT < vRoom ? vT : vRoom;
                        vBack = vBack > 0 && !zeroLit ? vBack : 0;
                        const int32_t vInput = vIn0 - vBack, vCand = vCand0 - vBack;
                        const int32_t vLimitLen = matchLimit - (vInput + MIN_MATCH);
                        const int32_t vFwd = (int32_t)(vF & 15u);
                        const int32_t vKnown = vBack + vFwd;
                        // not this way: more than 4 bytes match backwards -- the general way below goes to memory for the rest
                        const bool beyond = vBack == 4 && vRoom > 4;
                        if (__ballot(beyond) != 0) MWC(22);
                        if (__ballot(beyond) == 0) {  // (uniform)
                            MWC(21);
                            int32_t vMl = vKnown < vLimitLen ? vKnown : vLimitLen;
                            if (__ballot(vFwd >= 8 && vKnown < vLimitLen) != 0) {  // (uniform) the registers' 8 bytes behind the hit all match: memory has the rest
                                vMl = vKnown + vec(wave_count(in, vInput + MIN_MATCH + vKnown, vCand + MIN_MATCH + vKnown, matchLimit, lane));
                            }
                            const int32_t vLit = vInput - vAnchor;  // (0 behind a re-probe's hit: the zero-literal token :181-183 is the token of an empty run)
                            const int32_t vTok = vOutput;
                            const int32_t vLitPos = vTok + (vLit >= RUN_MASK ? 2 + (int32_t)((uint32_t)(vLit - RUN_MASK) / 255u) : 1);
                            if (pos >= vAnchor && pos < vInput) {
                                out[vLitPos + (pos - vAnchor)] = (uint8_t)x4;
                            }
                            vOutput = emit_match_v(vTok, vLit, vInput - vCand, vMl, vLitPos + vLit);
                            vAnchor = vInput + vMl + MIN_MATCH;
                            input = uni(vAnchor);
                            if (input > matchFindLimit) {  // :152-155
                                blockDone = true;
                                break;
                            }
                            const int rr = input - base;
                            if (rr < 64) {
                                M |= 1ull << (rr - 2);  // :157-159 the `input - 2` insert
                                r = rr;
                                continue;
                            }
                            break;  // the match ends beyond the window: the next one starts at input - 2
                        }
                    }
                    // ---- every other sequence: scalar code, as before round 6 ----
                    anchor = uni(vAnchor);
                    output = uni(vOutput);
                    int32_t back = 0;
                    int32_t matchLength = -1;
                    if (quick) {
                        bool slow = false;
                        if (!zeroLit) {  // catch up :141-144
                            const int32_t room = input - anchor < cand ? input - anchor : cand;
                            if (room > 0) {
                                const int32_t t = (int32_t)((factsW >> 4) & 7u);
                                back = t < room ? t : room;
                                slow = back == 4 && room > 4;  // more than 4 bytes match backwards: the general way below goes to memory for the rest
                            }
                        }
                        if (!slow) {
                            // count :240-267: `back` bytes of the 4-byte hit itself lie behind the new start + 4, then what was measured behind the hit
                            const int32_t a0 = input - back + MIN_MATCH, b0 = cand - back + MIN_MATCH;
                            const int32_t limitLen = matchLimit - a0;
                            const int32_t fwd = (int32_t)(factsW & 15u);
                            const int32_t known = back + fwd;
                            if (fwd < 8 || known >= limitLen) {
                                matchLength = known < limitLen ? known : limitLen;
                            }
                            else {
                                matchLength = known + wave_count(in, a0 + known, b0 + known, matchLimit, lane);
                            }
                            input -= back;
                            cand -= back;
                        }
                        else {
                            back = 0;
                        }
                    }
                    const bool general = matchLength < 0;
                    const bool winFast = general && (jl >= 0 || rl32((uint32_t)fast, wl) != 0);
                    const int shiftW = general ? (int)rl32((uint32_t)shift, wl) : 0;
                    uint64_t rLoW = 0, rHiW = 0;
                    if (general) {
                        rLoW = rl64(rLo, wl);
                        rHiW = rl64(rHi, wl);
                    }
                    if (general && !zeroLit) {  // catch up :141-144
                        int32_t room = input - anchor < cand ? input - anchor : cand;
                        bool slow = !winFast;
                        if (room > 0 && !slow) {
                            // the 4 bytes before either position, last byte in the top bits
                            uint32_t bi = 0, bc = 0;
                            if (wl >= 4) {
                                bi = rl32(x4, wl - 4);
                            }
                            else {
                                slow = true;
                            }
                            if (jl >= 4) {
                                bc = rl32(x4, jl - 4);
                            }
                            else if (jl >= 0) {
                                slow = true;
                            }
                            else {
                                bc = shiftW == 4 ? (uint32_t)rLoW : (uint32_t)((uint32_t)rLoW << (8 * (4 - shiftW)));  // (only `shift` = cand bytes exist before a candidate below 4: room stops there)
                            }
                            if (!slow) {
                                const uint32_t d = bi ^ bc;
                                const int32_t t = d == 0 ? 4 : (int32_t)(__builtin_clz(d) >> 3);
                                back = t < room ? t : room;
                                if (back == 4 && room > 4) {
                                    slow = true;  // more than 4 bytes match backwards: the rest from memory
                                }
                            }
                        }
                        if (slow && room > 0) {
                            int32_t i2 = input - back, m2 = cand - back, room2 = room - back;
                            while (room2 > 0) {
                                const bool eq = lane < room2 && in[i2 - 1 - lane] == in[m2 - 1 - lane];
                                const unsigned long long ne = ~__ballot(eq);
                                const int run = ne ? __builtin_ctzll(ne) : 64;
                                i2 -= run;
                                m2 -= run;
                                room2 -= run;
                                if (run < 64) {
                                    break;
                                }
                            }
                            back = input - i2;
                        }
                        input -= back;
                        cand -= back;
                    }
                    // literals [anchor, input): bytes of this window (anchor >= base here), stored from the lanes' registers; token byte behind the match length
                    int32_t literalLength = 0;
                    int32_t tokenPos;
                    if (zeroLit) {
                        tokenPos = output++;  // zero-literal token :181-183
                    }
                    else {
                        literalLength = input - anchor;
                        tokenPos = output;
                        const int32_t litPos = tokenPos + lz4_run_length_size(literalLength);
                        if (pos >= anchor && pos < input) {
                            out[litPos + (pos - anchor)] = (uint8_t)x4;
                        }
                        output = litPos + literalLength;
                    }
                    // count :240-267 from input + 4 against cand + 4: the first 8 bytes from registers
                    if (general) {
                        const int32_t a0 = input + MIN_MATCH, b0 = cand + MIN_MATCH;
                        const int32_t limitLen = matchLimit - a0;  // (>= 4: input <= matchFindLimit - 1)
                        const int la = a0 - base;
                        const int lb = b0 - base;
                        // (a catch-up that went to memory may have moved both positions out of what the registers hold)
                        const bool okA = la >= 0 && la < 64 && a0 + 8 <= inLen;
                        const bool okB = jl >= 0 ? (lb >= 0 && lb < 64 && b0 + 8 <= inLen) : (winFast && back <= shiftW + 4);
                        if (okA && okB) {
                            const uint64_t a8 = rl64(x, la);
                            const uint64_t b8 = jl >= 0 ? rl64(x, lb) : ext64(rLoW, rHiW, shiftW + 4 - back);
                            const uint64_t d = a8 ^ b8;
                            int32_t eq = d == 0 ? 8 : (int32_t)(__builtin_ctzll(d) >> 3);
                            eq = eq < limitLen ? eq : limitLen;
                            matchLength = eq;
                            if (eq == 8 && limitLen > 8) {
                                matchLength = 8 + wave_count(in, a0 + 8, b0 + 8, matchLimit, lane);
                            }
                        }
                        else {
                            matchLength = wave_count(in, a0, b0, matchLimit, lane);
                        }
                    }
                    emit_match(tokenPos, literalLength, input - cand, matchLength);
                    input += matchLength + MIN_MATCH;
                    anchor = input;
                    vAnchor = vec(anchor);
                    vOutput = vec(output);
                    if (input > matchFindLimit) {  // :152-155
                        blockDone = true;
                        break;
                    }
                    const int rr = input - base;
                    if (rr < 64) {
                        M |= 1ull << (rr - 2);  // :157-159 the `input - 2` insert
                        r = rr;
                        continue;
                    }
                    break;  // the match ends beyond the window: the next one starts at input - 2
                }
                anchor = uni(vAnchor);
                output = uni(vOutput);
                // the table takes the latest inserted lane of every hash
                {
                    const unsigned long long later = same & M & ~((2ull << lane) - 1ull);
                    if (canLoad && ((M >> lane) & 1ull) != 0 && later == 0) {
                        table[h] = (TableT)pos;
                    }
                }
                wave_sync();  // (the accesses of this wavefront to its table, in order; it may share its workgroup with others: lz4_compress_tiers_kernel)
                if (blockDone) {
                    break;
                }
                if (searchGoesOn) {
                    mode = 2;
                    scanStart = base + c;
                    k0 = 64 - c;
                    continue;
                }
                mode = 1;
            }
        }
        {  // emitLastLiteral :269-280
            const int32_t length = inputLimit - anchor;
            if (lane == 0) {
                lz4_write_run_length(out, output, length, 0);
            }
            output += lz4_run_length_size(length);
            group_copy<64>(out + output, in + anchor, length, lane);
            output += length;
        }
    }
    stOut = st;
    return output;
}

}  // namespace achip
