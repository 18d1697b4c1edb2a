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
// zc_match_any with fewer ballots: classes by the low 9 bits of the key, then one check that every class is pure (all of its lanes have the full
// key of its first lane) -- which it is unless two different hashes of a window share their low bits: then the full loop
__device__ __forceinline__ unsigned long long match_any_fast(uint32_t key, int keyBits, unsigned long long active, bool isActive)
{
    if (keyBits <= 9) {
        return zc_match_any(key, keyBits, active);
    }
    const unsigned long long eq = zc_match_any(key, 9, active);
    const int leader = eq != 0 ? __builtin_ctzll(eq) : 0;
    const uint32_t leaderKey = (uint32_t)__shfl((int)key, leader);
    if (__ballot(isActive && leaderKey != key) != 0) {
        return zc_match_any(key, keyBits, active);
    }
    return eq;
}
}  // namespace dmw

__device__ int32_t dfast_compress_block_mw(Ctx& c, int32_t inputAddress, int32_t inputSize)
{
    using namespace dmw;
    inputAddress = uni(inputAddress);  // (what the kernel loaded from the batch arrays is wave-uniform, which the compiler cannot know)
    inputSize = uni(inputSize);
    if (inputSize < 96) {
        return dfast_compress_block(c, inputAddress, inputSize);  // (the candidates' 32 bytes are taken from inside [0, inputEnd))
    }
    const uint8_t* __restrict__ in = c.in;
    const int lane = c.lane;
    const int32_t windowBase = uni(c.windowBaseOffset);
    int32_t* longTable = c.hashTable;
    int32_t* shortTable = c.chainTable;
    const int32_t longBits = uni(c.hashLog);
    const int32_t shortBits = uni(c.chainLog);
    const bool fiveByteHash = uni(c.searchLength) == 5;
    const int32_t inputEnd = inputAddress + inputSize;
    const int32_t inputLimit = inputEnd - 8;
    int32_t input = inputAddress;
    int32_t anchor = inputAddress;
    int32_t offset1 = uni(c.offset0);
    int32_t offset2 = uni(c.offset1);
    c.literalsLength = uni(c.literalsLength);
    c.sequenceCount = uni(c.sequenceCount);
    c.longLengthField = uni(c.longLengthField);
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
    const unsigned long long below = bits(0, lane);  // the lanes before this one

    auto hash_s = [&](uint64_t v) { return fiveByteHash ? hash5(v, shortBits) : hash4((uint32_t)v, shortBits); };
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
                const uint64_t a = uni(ld8(in + current + 2));
                longTable[hash8(a, longBits)] = current + 2;
                shortTable[hash_s(a)] = current + 2;
                const uint64_t b = uni(ld8(in + input - 2));
                longTable[hash8(b, longBits)] = input - 2;
                shortTable[hash_s(b)] = input - 2;
            }
            while (input <= inputLimit && offset2 > 0 && uni(ld4(in + input)) == uni(ld4(in + input - offset2))) {
                const int32_t repetitionLength = wave_count(in, input + 4, input + 4 - offset2, inputEnd, lane) + 4;
                const int32_t temp = offset2;
                offset2 = offset1;
                offset1 = temp;
                const uint64_t r = uni(ld8(in + input));
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
            const uint64_t here = uni(ld8(in + input));
            const int32_t shortHash = hash_s(here);
            int32_t shortMatch = uni(shortTable[shortHash]);
            const int32_t longHash = hash8(here, longBits);
            int32_t longMatch = uni(longTable[longHash]);
            const bool repHit = offset1 > 0 && uni(ld4(in + input + 1 - offset1)) == (uint32_t)(here >> 8);
            bool longHit = false, shortHit = false;
            if (!repHit) {
                longHit = longMatch > windowBase && uni(ld8(in + longMatch)) == here;
                if (!longHit) {
                    shortHit = shortMatch > windowBase && uni(ld4(in + shortMatch)) == (uint32_t)here;
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
                    while (input > anchor && longMatch > windowBase && uni((uint32_t)in[input - 1]) == uni((uint32_t)in[longMatch - 1])) {
                        input--;
                        longMatch--;
                        matchLength++;
                    }
                }
                else if (shortHit) {
                    const uint64_t next = uni(ld8(in + input + 1));
                    const int32_t nextHash = hash8(next, longBits);
                    int32_t nextMatch = uni(longTable[nextHash]);
                    longTable[nextHash] = current + 1;
                    if (nextMatch > windowBase && uni(ld8(in + nextMatch)) == next) {
                        matchLength = wave_count(in, input + 1 + 8, nextMatch + 8, inputEnd, lane) + 8;
                        input++;
                        offset = input - nextMatch;
                        while (input > anchor && nextMatch > windowBase && uni((uint32_t)in[input - 1]) == uni((uint32_t)in[nextMatch - 1])) {
                            input--;
                            nextMatch--;
                            matchLength++;
                        }
                    }
                    else {
                        matchLength = wave_count(in, input + 4, shortMatch + 4, inputEnd, lane) + 4;
                        offset = input - shortMatch;
                        while (input > anchor && shortMatch > windowBase && uni((uint32_t)in[input - 1]) == uni((uint32_t)in[shortMatch - 1])) {
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
        const unsigned long long sameS = match_any_fast((uint32_t)sHash, shortBits, ldMask, ld) & ldMask;
        const unsigned long long sameL = match_any_fast((uint32_t)lHash, longBits, ldMask, ld) & ldMask;
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

        // What a match against the TABLE's candidate would be, for every lane at once (the replay then reads one packed word per match instead
        // of walking the candidate's bytes): equal bytes behind the first 8 / 4 as far as the fetched bytes and the window's lanes go, equal
        // bytes before both positions (up to 4), each with a flag "everything known was equal: the rest is in memory".
        uint32_t packed = 0;
        {
            const uint64_t x8 = shfl64(x, lane + 8 < 64 ? lane + 8 : lane);     // the bytes at p + 8 (valid: lane + 8 < nld)
            const uint64_t x16 = shfl64(x, lane + 16 < 64 ? lane + 16 : lane);  // ... at p + 16
            const bool v8 = lane + 8 < nld, v16 = lane + 16 < nld;
            // the 4 bytes before p, the last one in the top byte; lanes 1 .. 3 know 1 .. 3 of them
            const uint32_t prev = (uint32_t)__shfl((int)x4, lane >= 4 ? lane - 4 : 0);
            const uint32_t xm4 = lane >= 4 ? prev : (lane == 0 ? 0u : prev << (8 * (4 - lane)));
            const int knownBack = lane < 4 ? lane : 4;
            if (okL) {
                const int k0 = shL + 8, nAvail = 24 - shL;
                const uint64_t d0 = x8 ^ ext32(L0, L1, L2, L3, k0);
                const uint64_t dn = x16 ^ ext32(L0, L1, L2, L3, k0 + 8);
                const int e0 = d0 == 0 ? 8 : (__builtin_ctzll(d0) >> 3);
                const int e1 = dn == 0 ? 8 : (__builtin_ctzll(dn) >> 3);
                const int cnt = e0 == 8 ? 8 + e1 : e0;
                const int usable = !v8 ? 0 : (nAvail < 8 ? nAvail : (!v16 ? 8 : (nAvail < 16 ? nAvail : 16)));
                const int f = cnt < usable ? cnt : usable;
                const int kb = shL < 4 ? shL : 4;
                const uint32_t before = shL >= 4 ? (uint32_t)ext32(L0, L1, L2, L3, shL - 4) : (shL == 0 ? 0u : (uint32_t)L0 << (8 * (4 - shL)));
                const uint32_t db = xm4 ^ before;
                const int eb = db == 0 ? 4 : (__builtin_clz(db) >> 3);
                const int ub = kb < knownBack ? kb : knownBack;
                const int b = eb < ub ? eb : ub;
                packed |= (uint32_t)f | (f == usable ? 32u : 0u) | ((uint32_t)b << 6) | (b == ub ? 512u : 0u);
            }
            if (okS) {
                const int k0 = shS + 4, nAvail = 12 - shS;
                const uint64_t a = (x >> 32) | (x8 << 32);
                const uint64_t d0 = a ^ ext32(S0, S1, 0ull, 0ull, k0);
                const int cnt = d0 == 0 ? 8 : (__builtin_ctzll(d0) >> 3);
                int usable = v8 ? 8 : 4;
                usable = nAvail < usable ? nAvail : usable;
                const int f = cnt < usable ? cnt : usable;
                const int kb = shS < 4 ? shS : 4;
                const uint32_t before = shS >= 4 ? (uint32_t)ext32(S0, S1, 0ull, 0ull, shS - 4) : (shS == 0 ? 0u : (uint32_t)S0 << (8 * (4 - shS)));
                const uint32_t db = xm4 ^ before;
                const int eb = db == 0 ? 4 : (__builtin_clz(db) >> 3);
                const int ub = kb < knownBack ? kb : knownBack;
                const int b = eb < ub ? eb : ub;
                packed |= ((uint32_t)f << 10) | (f == usable ? (1u << 14) : 0u) | ((uint32_t)b << 15) | (b == ub ? (1u << 18) : 0u);
            }
        }
        // hits against the tables' own entries do not change during the replay; lanes of the window with equal hashes (rare) make a lane's view of a
        // table depend on what the replay has inserted: only windows that have such lanes evaluate that
        const bool tableLongH = okL && cL8 == x;
        const bool tableShortH = okS && cS4 == x4;
        const unsigned long long sameLb = sameL & below, sameSb = sameS & below;
        const bool anySame = __ballot(sameLb != 0 || sameSb != 0) != 0;
        ZC_STAT(16, anySame ? 1 : 0);
        uint8_t* const litLane = c.litBuf + p;  // (a literal byte of lane q goes to litLane[literalsLength - anchor])
        // the window's sequences are kept by the lanes (sequence k by lane k) and stored together at its end
        int32_t qLL = 0, qOF = 0, qML = 0;
        int nq = 0;
        const int32_t seqBase = c.sequenceCount;
        auto seq_w = [&](int32_t literalLength, int32_t offsetCode, int32_t matchLengthBase) {
            c.literalsLength += literalLength;
            if (literalLength > 65535) {
                c.longLengthField = 1;
                c.longLengthPosition = seqBase + nq;
            }
            if (matchLengthBase > 65535) {
                c.longLengthField = 2;
                c.longLengthPosition = seqBase + nq;
            }
            if (lane == nq) {
                qLL = literalLength;
                qOF = offsetCode + 1;
                qML = matchLengthBase;
            }
            nq++;
        };

        unsigned long long ML = 0, MS = 0;  // lanes the replay has inserted into the long / short table
        int cs = 0;                         // first lane of the search that comes next
        int post = 0;                       // the match ended beyond the window: 1 = its inserts and the repeat loop, 2 = the rest of the repeat loop, from memory
        int32_t postCurrent = 0;

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
            int jL = -1, jS = -1;
            bool longH = tableLongH, shortH = tableShortH;
            if (anySame) {
                const unsigned long long seen = ~bits(0, cs);  // (with `below`: the search lanes before this one)
                const unsigned long long eL = sameLb & (ML | seen);
                jL = eL != 0 ? 63 - __builtin_clzll(eL) : -1;
                const unsigned long long eS = sameSb & (MS | seen);
                jS = eS != 0 ? 63 - __builtin_clzll(eS) : -1;
                const uint64_t xjL = shfl64(x, jL >= 0 ? jL : lane);
                const uint32_t xjS = (uint32_t)__shfl((int)x4, jS >= 0 ? jS : lane);
                if (jL >= 0) {
                    longH = base + jL > windowBase && xjL == x;
                }
                if (jS >= 0) {
                    shortH = base + jS > windowBase && xjS == x4;
                }
            }
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
                    litLane[c.literalsLength - anchor] = (uint8_t)x4;
                }
                seq_w(input - anchor, 0, matchLength - 3);
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
                ZC_STAT(isLong ? 4 : 5, 1);
                int32_t back = 0;
                if (jc < 0) {
                    // the table's candidate: what its lane has measured
                    const uint32_t pk = rl32(packed, m) >> (isLong ? 0 : 10);
                    const int f = (int)(pk & (isLong ? 31u : 15u));
                    const bool fAll = ((pk >> (isLong ? 5 : 4)) & 1u) != 0;
                    const int b = (int)((pk >> (isLong ? 6 : 5)) & 7u);
                    const bool bAll = ((pk >> (isLong ? 9 : 8)) & 1u) != 0;
                    matchLength = minLen + f;
                    if (fAll) {
                        ZC_STAT(11, 1);
                        matchLength += wave_count(in, input + minLen + f, cand + minLen + f, inputEnd, lane);
                    }
                    // backward :136-142 (and its twins :93-99, :117-123)
                    const int32_t roomA = input - anchor, roomB = cand - windowBase;
                    const int32_t room = roomA < roomB ? roomA : roomB;
                    back = b < room ? b : room;
                    if (bAll && back < room) {
                        ZC_STAT(8, 1);
                        while (input - back > anchor && cand - back > windowBase && uni((uint32_t)in[input - back - 1]) == uni((uint32_t)in[cand - back - 1])) {
                            back++;
                        }
                    }
                    ZC_STAT(7, back);
                }
                else {
                    // a candidate inside the window (rare): the other lanes' registers
                    ZC_STAT(6, 1);
                    matchLength = minLen + count_window(m - jc, m + minLen);
                    bool fromMemory = false;
                    while (input - back > anchor && cand - back > windowBase) {
                        const int li = input - back - 1 - base;
                        const int lc = jc - back - 1;
                        if (li < 0 || lc < 0) {
                            fromMemory = true;
                            break;
                        }
                        if ((rl32(x4, li) & 0xFFu) != (rl32(x4, lc) & 0xFFu)) {
                            break;
                        }
                        back++;
                    }
                    if (fromMemory) {
                        while (input - back > anchor && cand - back > windowBase && uni((uint32_t)in[input - back - 1]) == uni((uint32_t)in[cand - back - 1])) {
                            back++;
                        }
                    }
                }
                input -= back;
                cand -= back;
                matchLength += back;
                offset = input - cand;
                if (p >= litStored && p < input) {
                    litLane[c.literalsLength - anchor] = (uint8_t)x4;
                }
                seq_w(input - anchor, offset + 2, matchLength - 3);
                offset2 = offset1;
                offset1 = offset;
                d2 = d1;
                d1 = 0xFFFFFFFFu;
                if (ld && p + 1 >= offset && input + matchLength - base <= 62) {  // (a match that ends beyond the window ends the window)
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
                    seq_w(0, 0, repetitionLength - 3);
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
            litLane[c.literalsLength - anchor] = (uint8_t)x4;
        }
        if (lane < nq) {
            c.seqLitLen[seqBase + lane] = qLL;
            c.seqOffset[seqBase + lane] = qOF;
            c.seqMatchLen[seqBase + lane] = qML;
        }
        c.sequenceCount = seqBase + nq;
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
