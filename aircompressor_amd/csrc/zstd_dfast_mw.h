// zstd_dfast_mw.h -- DoubleFastBlockCompressor.compressBlock (M/zstd/DoubleFastBlockCompressor.java:28-180) of one wavefront in the "many
// matches per window" form (round 3; the LZ4 / Snappy encoders' lz4_compress_mw.h has the idea, dfast_compress_block() in
// zstd_compress_body.h the batch-probe form this one replaces as the default).  Included by zstd_compress_body.h INSIDE namespace achip::zc.
//
// On text the batch-probe match finder pays about seven dependent memory round trips per sequence (the probe's bytes, two table entries,
// the candidates' bytes, `count`, the backward extension, the literal copy, the bytes of the `current + 2` / `input - 2` inserts and of
// the repeat-offset loop), each waited for by the whole wavefront, for ~20 bytes of progress.  Here the lanes hold a WINDOW of 64
// consecutive positions -- the 8 bytes at the position, both hashes, both table entries as they were when the window began, 32 bytes
// around the long table's candidate ([entry - 8, entry + 24)), 16 around the short table's ([entry - 4, entry + 12)), and the 4 bytes at
// position + 1 - offset for both repeat offsets -- and the Java loop is replayed over those registers by wave-uniform code:
//   * what a table says at a lane = the latest lane of the window with the same hash that the replay has inserted so far (zc_match_any
//     masks intersected with the masks of inserted lanes, one per table), else the entry read at the window's start;
//   * the three hit tests (:71-73 repeat offset at position + 1, :83 long, :104 short, and :109-116 the long table at position + 1 behind
//     a short hit) are evaluated by all remaining lanes at once, the first hit is the match;
//   * `count` and the backward extension (:136-142) read other lanes' registers (a candidate inside the window: a byte-equality mask at
//     that distance; a repeat offset: the equality mask of the window against itself at that offset; a table candidate: its 32 / 16
//     bytes) and go to memory only for what lies beyond them;
//   * literals are stored from the lanes' registers, the sequence arrays by plain stores: nothing to wait for;
//   * a new offset (every match that is not a repeat) needs the window's bytes at that offset for the repeat tests that follow: ONE
//     load per lane, issued when the offset is known and first needed at the next search -- the only dependent round trip per sequence;
//   * at the end of the window both tables take the latest inserted lane of every hash (positions are inserted in increasing order).
// The window form is used while the search advances by one position per probe (input - anchor < 192: step = ((input - anchor) >> 8) + 1,
// :150); longer literal runs (incompressible data) take the serial step below, which is the Java loop as it stands.
#pragma once

#ifdef ACHIP_HOST_STATS  // (CPU emulator only: how often each part of the replay runs -- tools/hostemu/zc_stats.py)
extern "C" long long g_zc_stats[32];
// (out of line and not instrumented: under the emulator's access-granular lockstep every traced access is an order point of ALL lanes)
static __attribute__((noinline, no_sanitize("coverage"))) void zc_stat_add(int i, long long n) { g_zc_stats[i] += n; }
#define ZC_STAT(i, n) \
    if (lane == 0) zc_stat_add((i), (n))
#else
#define ZC_STAT(i, n)
#endif

namespace dmw {
__device__ __forceinline__ uint32_t rl32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int src) { return ((uint64_t)rl32((uint32_t)(v >> 32), src) << 32) | rl32((uint32_t)v, src); }
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src);
}
// bits [lo, hi) of a 64-bit mask (0 <= lo, hi <= 64; empty when hi <= lo)
__device__ __forceinline__ uint64_t bits(int lo, int hi)
{
    const uint64_t upTo = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    const uint64_t below = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
    return upTo & ~below;
}
// 8 bytes from byte k (0 .. 31) of the 32 bytes r0 .. r3 (what lies beyond them reads as 0)
__device__ __forceinline__ uint64_t ext32(uint64_t r0, uint64_t r1, uint64_t r2, uint64_t r3, int k)
{
    const int w = k >> 3, s = (k & 7) * 8;
    const uint64_t lo = w == 0 ? r0 : (w == 1 ? r1 : (w == 2 ? r2 : r3));
    const uint64_t hi = w == 0 ? r1 : (w == 1 ? r2 : (w == 2 ? r3 : 0ull));
    return s == 0 ? lo : ((lo >> s) | (hi << (64 - s)));
}
// number of consecutive set bits of E from bit k, at most up to bit nv (the number of valid bits)
__device__ __forceinline__ int run_from(uint64_t E, int k, int nv)
{
    if (k >= nv) {
        return 0;
    }
    const uint64_t inv = ~(E >> k);
    const int r = inv != 0 ? __builtin_ctzll(inv) : 64;
    return r < nv - k ? r : nv - k;
}
}  // namespace dmw

__device__ int32_t dfast_compress_block_mw(Ctx& c, int32_t inputAddress, int32_t inputSize)
{
    using namespace dmw;
    if (inputSize < 96) {
        return dfast_compress_block(c, inputAddress, inputSize);  // (the candidates' 32 bytes are taken from inside [0, inputEnd))
    }
    const uint8_t* __restrict__ in = c.in;
    const int lane = c.lane;
    const int32_t windowBase = c.windowBaseOffset;
    int32_t* longTable = c.hashTable;
    int32_t* shortTable = c.chainTable;
    const int32_t longBits = c.hashLog;
    const int32_t shortBits = c.chainLog;
    const int32_t inputEnd = inputAddress + inputSize;
    const int32_t inputLimit = inputEnd - 8;
    int32_t input = inputAddress;
    int32_t anchor = inputAddress;
    int32_t offset1 = c.offset0;
    int32_t offset2 = c.offset1;
    int32_t savedOffset = 0;
    if (input - windowBase == 0) {
        input++;
    }
    const int32_t maxRep = input - windowBase;
    if (offset2 > maxRep) {
        savedOffset = offset2;
        offset2 = 0;
    }
    if (offset1 > maxRep) {
        savedOffset = offset1;
        offset1 = 0;
    }
    int32_t litStored = anchor;  // the literal bytes [anchor, litStored) are in litBuf already (stored from registers as the windows went by)

    auto hash_s = [&](uint64_t v) { return c.searchLength == 5 ? hash5(v, shortBits) : hash4((uint32_t)v, shortBits); };
    // SequenceStore.storeSequence :83-111 without the literal copy
    auto seq = [&](int32_t literalLength, int32_t offsetCode, int32_t matchLengthBase) {
        c.literalsLength += literalLength;
        const int32_t i = c.sequenceCount;
        if (literalLength > 65535) {
            c.longLengthField = 1;
            c.longLengthPosition = i;
        }
        c.seqLitLen[i] = literalLength;
        c.seqOffset[i] = offsetCode + 1;
        if (matchLengthBase > 65535) {
            c.longLengthField = 2;
            c.longLengthPosition = i;
        }
        c.seqMatchLen[i] = matchLengthBase;
        c.sequenceCount = i + 1;
    };
    // :153-177 behind a match that began at `current` and ended at `input`, from memory: the two inserts (unless the window has made them) and the repeat-offset loop
    auto after_match_serial = [&](int32_t current, bool inserts) {
        if (input <= inputLimit) {
            if (inserts) {
                const uint64_t a = ld8(in + current + 2);
                longTable[hash8(a, longBits)] = current + 2;
                shortTable[hash_s(a)] = current + 2;
                const uint64_t b = ld8(in + input - 2);
                longTable[hash8(b, longBits)] = input - 2;
                shortTable[hash_s(b)] = input - 2;
            }
            while (input <= inputLimit && offset2 > 0 && ld4(in + input) == ld4(in + input - offset2)) {
                const int32_t repetitionLength = wave_count(in, input + 4, input + 4 - offset2, inputEnd, lane) + 4;
                const int32_t temp = offset2;
                offset2 = offset1;
                offset1 = temp;
                const uint64_t r = ld8(in + input);
                shortTable[hash_s(r)] = input;
                longTable[hash8(r, longBits)] = input;
                seq(0, 0, repetitionLength - 3);
                input += repetitionLength;
                anchor = input;
            }
        }
        litStored = anchor;
    };

    while (input < inputLimit) {
        if (input - anchor >= 192) {
            ZC_STAT(14, 1);
            // ---- the serial step: the Java loop as it stands (long literal runs: the probes skip ahead) ----
            const uint64_t here = ld8(in + input);
            const int32_t shortHash = hash_s(here);
            int32_t shortMatch = shortTable[shortHash];
            const int32_t longHash = hash8(here, longBits);
            int32_t longMatch = longTable[longHash];
            const bool repHit = offset1 > 0 && ld4(in + input + 1 - offset1) == ld4(in + input + 1);
            bool longHit = false, shortHit = false;
            if (!repHit) {
                longHit = longMatch > windowBase && ld8(in + longMatch) == here;
                if (!longHit) {
                    shortHit = shortMatch > windowBase && ld4(in + shortMatch) == (uint32_t)here;
                }
            }
            const int32_t current = input;
            longTable[longHash] = current;
            shortTable[shortHash] = current;
            int32_t matchLength;
            int32_t offset = 0;
            if (repHit) {
                matchLength = wave_count(in, input + 1 + 4, input + 1 + 4 - offset1, inputEnd, lane) + 4;
                input++;
                store_sequence(c, anchor, input - anchor, 0, matchLength - 3);
            }
            else {
                if (longHit) {
                    matchLength = wave_count(in, input + 8, longMatch + 8, inputEnd, lane) + 8;
                    offset = input - longMatch;
                    while (input > anchor && longMatch > windowBase && in[input - 1] == in[longMatch - 1]) {
                        input--;
                        longMatch--;
                        matchLength++;
                    }
                }
                else if (shortHit) {
                    const uint64_t next = ld8(in + input + 1);
                    const int32_t nextHash = hash8(next, longBits);
                    int32_t nextMatch = longTable[nextHash];
                    longTable[nextHash] = current + 1;
                    if (nextMatch > windowBase && ld8(in + nextMatch) == next) {
                        matchLength = wave_count(in, input + 1 + 8, nextMatch + 8, inputEnd, lane) + 8;
                        input++;
                        offset = input - nextMatch;
                        while (input > anchor && nextMatch > windowBase && in[input - 1] == in[nextMatch - 1]) {
                            input--;
                            nextMatch--;
                            matchLength++;
                        }
                    }
                    else {
                        matchLength = wave_count(in, input + 4, shortMatch + 4, inputEnd, lane) + 4;
                        offset = input - shortMatch;
                        while (input > anchor && shortMatch > windowBase && in[input - 1] == in[shortMatch - 1]) {
                            input--;
                            shortMatch--;
                            matchLength++;
                        }
                    }
                }
                else {
                    input += ((input - anchor) >> 8) + 1;
                    continue;
                }
                offset2 = offset1;
                offset1 = offset;
                store_sequence(c, anchor, input - anchor, offset + 2, matchLength - 3);
            }
            input += matchLength;
            anchor = input;
            after_match_serial(current, true);
            continue;
        }

        ZC_STAT(0, 1);
        // ---- a window: the 64 positions from `input`; lanes 0 .. nact - 1 may be probed, lane 63 only lends its bytes ----
        const int32_t base = input;
        if (litStored < base) {  // (only at the start of a block, whose first position is never probed: :52-54)
            group_copy<64>(c.litBuf + c.literalsLength + (litStored - anchor), in + litStored, base - litStored, lane);
            litStored = base;
        }
        const int32_t p = base + lane;
        const bool ld = p <= inputLimit;  // 8 bytes at p exist
        const int nld = inputLimit - base + 1 < 64 ? inputLimit - base + 1 : 64;
        const int nact = inputLimit - base < 63 ? inputLimit - base : 63;
        uint64_t x = 0;
        uint32_t d1 = 0xFFFFFFFFu, d2 = 0xFFFFFFFFu;  // (bytes at p + 1 .. p + 4) ^ (bytes at p + 1 - offset ..): a repeat match at p + 1 <=> 0
        int32_t sHash = 0, lHash = 0, tS = 0, tL = 0;
        if (ld) {
            x = ld8(in + p);
            const uint32_t xs = (uint32_t)(x >> 8);
            if (offset1 > 0 && p + 1 >= offset1) {
                d1 = ld4(in + p + 1 - offset1) ^ xs;
            }
            if (offset2 > 0 && p + 1 >= offset2) {
                d2 = ld4(in + p + 1 - offset2) ^ xs;
            }
            sHash = hash_s(x);
            lHash = hash8(x, longBits);
            tS = shortTable[sHash];
            tL = longTable[lHash];
        }
        const uint32_t x4 = (uint32_t)x;
        const unsigned long long ldMask = __ballot(ld);
        const unsigned long long sameS = zc_match_any((uint32_t)sHash, shortBits, ldMask) & ldMask;
        const unsigned long long sameL = zc_match_any((uint32_t)lHash, longBits, ldMask) & ldMask;
        // the candidates' surroundings
        uint64_t L0 = 0, L1 = 0, L2 = 0, L3 = 0, S0 = 0, S1 = 0;
        int shL = 0, shS = 0;
        uint64_t cL8 = 0;
        uint32_t cS4 = 0;
        const bool okL = ld && tL > windowBase;
        const bool okS = ld && tS > windowBase;
        if (okL) {
            int32_t s = tL - 8;
            s = s < 0 ? 0 : s;
            s = s > inputEnd - 32 ? inputEnd - 32 : s;
            shL = tL - s;
            const u32x4 a = ld16(in + s);
            const u32x4 b = ld16(in + s + 16);
            L0 = (uint64_t)a.x | ((uint64_t)a.y << 32);
            L1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
            L2 = (uint64_t)b.x | ((uint64_t)b.y << 32);
            L3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
            cL8 = ext32(L0, L1, L2, L3, shL);
        }
        if (okS) {
            int32_t s = tS - 4;
            s = s < 0 ? 0 : s;
            s = s > inputEnd - 16 ? inputEnd - 16 : s;
            shS = tS - s;
            const u32x4 a = ld16(in + s);
            S0 = (uint64_t)a.x | ((uint64_t)a.y << 32);
            S1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
            cS4 = (uint32_t)ext32(S0, S1, 0ull, 0ull, shS);
        }

        unsigned long long ML = 0, MS = 0;  // lanes the replay has inserted into the long / short table
        int cs = 0;                         // first lane of the search that comes next
        int post = 0;                       // the match ended beyond the window: 1 = its inserts and the repeat loop, 2 = the rest of the repeat loop, from memory
        int32_t postCurrent = 0;

        // equal bytes of in[aPos ...] (held by the lanes from la0 on) and bytes k0 ... of the nbytes (r0 .. r3) read around a table candidate (bPos), then memory
        auto count_table = [&](uint64_t r0, uint64_t r1, uint64_t r2, uint64_t r3, int nbytes, int k0, int la0, int32_t aPos, int32_t bPos) -> int32_t {
            int32_t eq = 0;
            bool mismatch = false;
            for (;;) {
                const int la = la0 + eq, k = k0 + eq;
                if (la >= nld || k >= nbytes) {
                    break;
                }
                const int n = nbytes - k > 8 ? 8 : nbytes - k;
                const uint64_t dd = rl64(x, la) ^ ext32(r0, r1, r2, r3, k);
                const int e = dd == 0 ? 8 : (__builtin_ctzll(dd) >> 3);
                if (e < n) {
                    eq += e;
                    mismatch = true;
                    break;
                }
                eq += n;
            }
            if (!mismatch) {
                ZC_STAT(11, 1);
                eq += wave_count(in, aPos + eq, bPos + eq, inputEnd, lane);
            }
            return eq;
        };
        // equal bytes of in[base + k ...] and in[base + k - o ...] for a distance o inside the window, then memory
        auto count_window = [&](int o, int k) -> int32_t {
            const uint32_t prev = (uint32_t)__shfl((int)x4, lane >= o ? lane - o : lane);
            const unsigned long long Eo = __ballot(ld && lane >= o && ((prev ^ x4) & 0xFFu) == 0);
            int32_t eq = run_from(Eo, k, nld);
            if (k + eq >= nld) {
                ZC_STAT(12, 1);
                eq += wave_count(in, base + k + eq, base + k + eq - o, inputEnd, lane);
            }
            return eq;
        };
        // equal bytes of in[base + k + 1 ...] and in[base + k + 1 - offset ...] from the repeat-offset differences d (bit q of its mask: the byte at p_q + 1)
        auto count_repeat = [&](uint32_t d, int32_t offset, int k) -> int32_t {
            const unsigned long long E = __ballot((d & 0xFFu) == 0);
            int32_t eq = run_from(E, k, nld);
            if (k + eq >= nld) {
                ZC_STAT(13, 1);
                eq += wave_count(in, base + k + eq + 1, base + k + eq + 1 - offset, inputEnd, lane);
            }
            return eq;
        };

        while (cs < nact) {
            ZC_STAT(1, 1);
            // ---- the search :57-150 over lanes cs .. nact - 1 at once: a lane sees the replay's inserts and those of the search lanes before it ----
            const unsigned long long below = bits(0, lane);
            const unsigned long long assumed = bits(cs, lane);
            const unsigned long long eL = sameL & (ML | assumed) & below;
            const int jL = eL != 0 ? 63 - __builtin_clzll(eL) : -1;
            const unsigned long long eS = sameS & (MS | assumed) & below;
            const int jS = eS != 0 ? 63 - __builtin_clzll(eS) : -1;
            const uint64_t xjL = shfl64(x, jL >= 0 ? jL : lane);
            const uint32_t xjS = (uint32_t)__shfl((int)x4, jS >= 0 ? jS : lane);
            const int32_t pL = jL >= 0 ? base + jL : tL;
            const int32_t pS = jS >= 0 ? base + jS : tS;
            const bool longH = pL > windowBase && (jL >= 0 ? xjL == x : (okL && cL8 == x));
            const bool shortH = pS > windowBase && (jS >= 0 ? xjS == x4 : (okS && cS4 == x4));
            const bool repH = offset1 > 0 && d1 == 0;
            const bool probing = lane >= cs && lane < nact;
            const unsigned long long hm = __ballot(probing && (repH || longH || shortH));
            if (hm == 0) {
                ZC_STAT(15, 1);
                ML |= bits(cs, nact);
                MS |= bits(cs, nact);
                cs = nact;
                input = base + nact;
                break;
            }
            const int w = __builtin_ctzll(hm);
            ML |= bits(cs, w + 1);
            MS |= bits(cs, w + 1);
            const uint32_t flags = (repH ? 1u : 0u) | (longH ? 2u : 0u) | (shortH ? 4u : 0u);
            const uint32_t fw = rl32(flags, w);
            const int32_t current = base + w;
            int32_t matchLength;
            int32_t offset = 0;
            ZC_STAT(2, 1);
            if ((fw & 1u) != 0) {
                ZC_STAT(3, 1);
                // repeat offset at current + 1 :71-80
                matchLength = 4 + count_repeat(d1, offset1, w + 4);
                input = current + 1;
                if (p >= litStored && p < input) {
                    c.litBuf[c.literalsLength + (p - anchor)] = (uint8_t)x4;
                }
                seq(input - anchor, 0, matchLength - 3);
            }
            else {
                int m = w;        // lane where the match starts
                bool isLong = true;
                if ((fw & 2u) == 0) {
                    // short hit: the long table at current + 1 first :104-123
                    ML |= 1ull << (w + 1);
                    if ((rl32(flags, w + 1) & 2u) != 0) {
                        m = w + 1;
                    }
                    else {
                        isLong = false;
                    }
                }
                const int jc = (int)rl32((uint32_t)(isLong ? jL : jS), m);
                int32_t cand = jc >= 0 ? base + jc : (int32_t)rl32((uint32_t)(isLong ? tL : tS), m);
                input = base + m;
                const int minLen = isLong ? 8 : 4;
                uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
                int shW = 0;
                ZC_STAT(isLong ? 4 : 5, 1);
                if (jc >= 0) {
                    ZC_STAT(6, 1);
                    matchLength = minLen + count_window(m - jc, m + minLen);
                }
                else if (isLong) {
                    r0 = rl64(L0, m);
                    r1 = rl64(L1, m);
                    r2 = rl64(L2, m);
                    r3 = rl64(L3, m);
                    shW = (int)rl32((uint32_t)shL, m);
                    matchLength = 8 + count_table(r0, r1, r2, r3, 32, shW + 8, m + 8, input + 8, cand + 8);
                }
                else {
                    r0 = rl64(S0, m);
                    r1 = rl64(S1, m);
                    shW = (int)rl32((uint32_t)shS, m);
                    matchLength = 4 + count_table(r0, r1, 0ull, 0ull, 16, shW + 4, m + 4, input + 4, cand + 4);
                }
                // backward :136-142 (and its twins :93-99, :117-123): registers first
                int32_t back = 0;
                bool fromMemory = false;
                while (input - back > anchor && cand - back > windowBase) {
                    const int li = input - back - 1 - base;
                    if (li < 0) {
                        fromMemory = true;
                        break;
                    }
                    const uint32_t bi = rl32(x4, li) & 0xFFu;
                    uint32_t bc;
                    if (jc >= 0) {
                        const int lc = jc - back - 1;
                        if (lc < 0) {
                            fromMemory = true;
                            break;
                        }
                        bc = rl32(x4, lc) & 0xFFu;
                    }
                    else {
                        const int kb = shW - back - 1;
                        if (kb < 0) {
                            fromMemory = true;
                            break;
                        }
                        bc = (uint32_t)ext32(r0, r1, r2, r3, kb) & 0xFFu;
                    }
                    if (bi != bc) {
                        break;
                    }
                    back++;
                }
                ZC_STAT(7, back);
                if (fromMemory) {
                    ZC_STAT(8, 1);
                    while (input - back > anchor && cand - back > windowBase && in[input - back - 1] == in[cand - back - 1]) {
                        back++;
                    }
                }
                input -= back;
                cand -= back;
                matchLength += back;
                offset = input - cand;
                if (p >= litStored && p < input) {
                    c.litBuf[c.literalsLength + (p - anchor)] = (uint8_t)x4;
                }
                seq(input - anchor, offset + 2, matchLength - 3);
                offset2 = offset1;
                offset1 = offset;
                d2 = d1;
                d1 = 0xFFFFFFFFu;
                if (ld && p + 1 >= offset) {
                    d1 = ld4(in + p + 1 - offset) ^ (uint32_t)(x >> 8);
                }
            }
            input += matchLength;
            anchor = input;
            litStored = anchor;
            if (input <= inputLimit) {
                if (input - base > 62) {
                    ZC_STAT(9, 1);
                    post = 1;
                    postCurrent = current;
                    break;
                }
                // :155-162 the inserts of current + 2 and input - 2
                ML |= (1ull << (w + 2)) | (1ull << (input - base - 2));
                MS |= (1ull << (w + 2)) | (1ull << (input - base - 2));
                // :164-176 the repeat-offset loop at `input`: d2 of the lane before it
                while (input <= inputLimit && offset2 > 0) {
                    const int q = input - base - 1;
                    if (rl32(d2, q) != 0) {
                        break;
                    }
                    ZC_STAT(10, 1);
                    const int32_t repetitionLength = 4 + count_repeat(d2, offset2, q + 4);
                    const int32_t to = offset2;
                    offset2 = offset1;
                    offset1 = to;
                    const uint32_t td = d2;
                    d2 = d1;
                    d1 = td;
                    ML |= 1ull << (q + 1);
                    MS |= 1ull << (q + 1);
                    seq(0, 0, repetitionLength - 3);
                    input += repetitionLength;
                    anchor = input;
                    litStored = anchor;
                    if (input - base > 62) {
                        post = 2;
                        break;
                    }
                }
                if (post != 0) {
                    break;
                }
            }
            cs = input - base;
        }
        // literal bytes of this window behind the last match: stored now, from registers (the next window's first sequence, or the block's end, owns them)
        if (post == 0 && p >= litStored && p < input) {
            c.litBuf[c.literalsLength + (p - anchor)] = (uint8_t)x4;
        }
        if (post == 0 && input > litStored) {
            litStored = input;
        }
        // both tables take the latest inserted lane of every hash
        {
            const unsigned long long above = lane >= 63 ? 0ull : ~((2ull << lane) - 1ull);
            if (ld && ((ML >> lane) & 1ull) != 0 && (sameL & ML & above) == 0) {
                longTable[lHash] = p;
            }
            if (ld && ((MS >> lane) & 1ull) != 0 && (sameS & MS & above) == 0) {
                shortTable[sHash] = p;
            }
        }
        wave_mem_order();
        if (post != 0) {
            after_match_serial(postCurrent, post == 1);
        }
    }
    c.tempOffset0 = offset1 != 0 ? offset1 : savedOffset;
    c.tempOffset1 = offset2 != 0 ? offset2 : savedOffset;
    return inputEnd - anchor;
}
