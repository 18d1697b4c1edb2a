// snappy_decompress_v5.hip -- batched Snappy raw-format decode for gfx950 in two passes: parse to records, then a wavefront per block
// executes them (the Snappy counterpart of lz4_decompress_v7.hip; achip_seqexec.h has the design).
//
// Same contract and Java-order checks as snappy_decompress_v2.hip (M/snappy/SnappyRawDecompressor.java:35-322).  As for LZ4, every check
// of the Java loop depends on lengths, offsets and positions only, so the parse pass decides status, error offset and output length.
// A lane per block; elements become records {literal run <= 16, copy <= 16}: a run and the copy behind it share one, longer runs and
// copies are cut into pieces.  The executor counts compressed positions from the block's first byte: the length preamble is part of the
// first record's `skip`.
#include <type_traits>

#include "achip_seqexec.h"
#include "achip_waveparse.h"

namespace achip {

__device__ __forceinline__ int32_t snappy_op_entry5(int32_t op)  // opLookupTable layout :223-271
{
    const int32_t kind = op & 3;
    const int32_t hi = op >> 2;
    if (kind == 0) {
        return hi < 60 ? hi + 1 : (((hi - 59) << 11) | 1);
    }
    if (kind == 1) {
        return (1 << 11) | ((hi >> 3) << 8) | ((hi & 7) + 4);
    }
    return ((kind == 2 ? 2 : 4) << 11) | (hi + 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The parse pass: the structure of lz4_parse2_kernel (lz4_decompress_v7.hip has the reasons) -- the
// stream through sx::LaneFeed (loads in flight for four trips), exactly one record per lane and trip kept in registers, flags in vector
// registers and a select-only common path, sequences cut into pieces of at most 16 literal + 16 copy bytes for the second executor.
// A trip looks at the element at ip and, when that is a literal run of at most 16 bytes, at the element behind it: a run and the copy
// behind it make one record.  FAST PATH: runs with the length in the tag (<= 60 bytes) and copies with 1- or 2-byte offsets, nothing
// within the last bytes of either buffer, offset inside the output -- for those every Java check is known to pass.  Anything else is
// parsed by snappy_parse_general, one element at a time: uncompressAll's loop body (M/snappy/SnappyRawDecompressor.java:84-216)
// restated check by check, reading the stream directly.
struct SnappyParseState {
    int32_t ip, op, st, eo;
};

// one element, the general way; returns 0 (an error -- S.st set -- or an element of length 0), 1 (a run: rLen bytes at rStart) or 2 (a copy)
__device__ __forceinline__ int snappy_parse_general(const uint8_t* __restrict__ in, SnappyParseState& S, int32_t inLimit, int32_t outLimit, int32_t& rLen, int32_t& rOff, int32_t& rStart)
{
    const int32_t fastOutLimit = outLimit - 8;
    int32_t ip = S.ip;
    const int32_t opc = (int32_t)in[ip];
    ip++;
#define SN_FAIL2(off)                                                      \
    {                                                                      \
        S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_MALFORMED); \
        S.eo = (int32_t)(off);                                             \
        return 0;                                                          \
    }
    const int32_t entry = snappy_op_entry5(opc);
    const int32_t trailerBytes = entry >> 11;
    if (!(ip + 4 < inLimit)) {  // :90-92
        if (ip + trailerBytes > inLimit) SN_FAIL2(ip);
    }
    uint32_t t4 = 0;
    for (int i = 0; i < trailerBytes; i++) {  // (the bytes the masked 4-byte load of :87-106 keeps)
        t4 |= (uint32_t)in[ip + i] << (8 * i);
    }
    const int32_t trailer = (int32_t)t4;
    if (trailer < 0) SN_FAIL2(ip);
    ip += trailerBytes;
    const int32_t length = entry & 0xff;
    if (length == 0) {
        S.ip = ip;
        return 0;
    }
    if ((opc & 3) == 0) {  // literal :116-146
        const int32_t lit = (int32_t)((uint32_t)length + (uint32_t)trailer);
        if (lit < 0) SN_FAIL2(ip);
        const int64_t litOutLimit = (int64_t)S.op + lit;
        if ((litOutLimit > fastOutLimit || (int64_t)ip + lit > inLimit - 8) && (litOutLimit > outLimit || (int64_t)ip + lit > inLimit)) SN_FAIL2(ip);
        rLen = lit;
        rStart = ip;
        S.ip = ip + lit;
        S.op += lit;
        return 1;
    }
    // copy :147-216
    const int32_t matchOffset = (int32_t)((uint32_t)(entry & 0x700) + (uint32_t)trailer);
    if (matchOffset <= 0 || matchOffset > S.op || (int64_t)S.op + length > outLimit) SN_FAIL2(ip);
#undef SN_FAIL2
    rLen = length;
    rOff = matchOffset;
    rStart = ip;
    S.ip = ip;
    S.op += length;
    return 2;
}

__global__ __launch_bounds__(64) void snappy_parse2_kernel(BatchArgs a, sx::ArenaHeader* hdr, sx::BlockMeta* meta, int32_t* only, uint64_t* arena, int32_t maxChunks, const int32_t* stats)
{
    if (stats != nullptr && snappy_pick(stats, a.nBlocks) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    constexpr int NS = 4;
    using Feed = sx::LaneFeed<NS>;
    __shared__ __attribute__((aligned(16))) uint8_t ldsIn[Feed::STRIDE * 64];
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < batch_count(a);
    const uint8_t* in0 = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    const int32_t inLen0 = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    SnappyParseState S;
    S.ip = 0;
    S.op = 0;
    S.st = 0;
    S.eo = 0;
    int32_t finished = have ? 0 : 1;

    // readUncompressedLength :277-321 (at most 5 bytes: read straight from the input buffer)
    uint32_t expected = 0;
    int32_t nread = 0;
    if (have) {
        for (int i = 0; i < 5; i++) {
            if (nread >= inLen0) {
                S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_TRUNCATED);
                S.eo = inLen0 - nread;
                break;
            }
            const uint32_t b = in0[nread++];
            expected |= (b & 0x7f) << (7 * i);
            if ((b & 0x80) == 0) {
                break;
            }
            if (i == 4) {
                S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LEN_HIGH_BIT);
                S.eo = nread;
            }
        }
        if (S.st == 0 && (int32_t)expected < 0) {
            S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_INVALID_LENGTH);
            S.eo = 0;
        }
        if (S.st == 0 && (int64_t)expected > (int64_t)outLimit) {  // :49-50
            S.st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_OUTPUT_TOO_SMALL);
            S.eo = 0;
        }
        if (S.st != 0) {
            finished = 1;
        }
    }

    // uncompressAll :70-220 ; positions relative to the first byte after the varint
    const uint8_t* const in = in0 + (finished != 0 ? 0 : nread);
    const int32_t inLimit = finished != 0 ? 0 : inLen0 - nread;
    Feed L;
    {
        const unsigned long long nonEmpty = __ballot(inLimit > 0);
        const uint8_t* anywhere = (const uint8_t*)hdr;
        if (nonEmpty != 0) {  // (uniform) one idle address per wavefront, as an offset from the batch's base (the loads stay global_load)
            anywhere = a.srcBase + (int64_t)sx::shfl_u64((uint64_t)((in - a.srcBase) - (int64_t)((uintptr_t)in & 31)), __builtin_ctzll(nonEmpty));
        }
        L.init(ldsIn, lane, in, inLimit, anywhere);
    }
    const int32_t B = L.inBase;
    const int32_t fastOutLimit = outLimit - 8;

    // ---- state of the sequence under way ----
    // phase 0: at an element; 1: a short run read (tLit bytes at tStart), at the element behind it (q); 2: its pieces are being emitted
    int32_t phase = 0;
    int32_t tLit = 0, tStart = 0, q = 0;
    int32_t sLit = 0, sLitPos = 0, sMl = 0, sOff = 0, sK = 0;
    int32_t litEndPrev = 0;  // position (counted from in0) behind the previous record's literals
    int32_t fallback = 0;
    uint64_t rec[8];
    int32_t groupAny = 0;
    int32_t firstChunk = -1, chunk = -1, fill = sx::CHUNK_RECS, count = 0;

    auto trip = [&](auto tTag) {
        constexpr int T = decltype(tTag)::value;
        constexpr int SLOT = T % NS;
        L.template land<SLOT>();
        const bool active = finished == 0;
        // ---- phase 0: the element at ip ----
        const int32_t v0 = S.ip + B;
        const uint32_t w = L.rd32(v0);
        const uint32_t tag = w & 0xFF;
        const bool do0 = active && phase == 0 && L.resident(v0, 4);
        const bool isRun = (tag & 3) == 0;
        const int32_t nLit = (int32_t)(tag >> 2) + 1;          // (a run with its length in the tag)
        const int32_t nQ = S.ip + 1 + nLit;
        // a run: fast when the length is in the tag, the four bytes behind the tag are inside the stream (:90), and neither buffer's last
        // bytes are near (:128 -- the first half of that condition false)
        const bool runFast = (tag >> 2) < 60 && S.ip + 5 < inLimit && S.op + nLit <= fastOutLimit && nQ <= inLimit - 8;
        const bool end0 = do0 && S.ip >= inLimit;               // the loop condition :84
        const bool gen0 = do0 && !end0 && isRun && !runFast;
        const bool ok0run = do0 && !end0 && isRun && runFast;
        const bool ok0copy = do0 && !end0 && !isRun;            // a copy at ip: phase 1 looks at it (an empty run before it)
        tLit = ok0run ? nLit : (ok0copy ? 0 : tLit);
        tStart = ok0run ? S.ip + 1 : (ok0copy ? S.ip : tStart);
        q = ok0run ? nQ : (ok0copy ? S.ip : q);
        // a run of more than 16 bytes is emitted on its own (pieces); a shorter one waits for the element behind it
        const bool longRun = ok0run && nLit > 16;
        sLit = longRun ? nLit : sLit;
        sLitPos = longRun ? nread + S.ip + 1 : sLitPos;
        sMl = longRun ? 0 : sMl;
        sOff = longRun ? 0 : sOff;
        sK = longRun ? 0 : sK;
        S.op += ok0run ? nLit : 0;
        S.ip = ok0run ? nQ : S.ip;
        phase = longRun ? 2 : ((ok0run || ok0copy) ? 1 : phase);
        finished |= end0 ? 1 : 0;
        L.restart(nQ + B, ok0run && nQ + B >= L.issueV + 64);
        // ---- phase 1: the element at q, behind a run of tLit <= 16 bytes (S.ip == q) ----
        const int32_t v1 = q + B;
        const uint32_t x = L.rd32(v1);
        const uint32_t tag2 = x & 0xFF;
        const bool do1 = finished == 0 && phase == 1 && L.resident(v1, 4);
        const uint32_t kind = tag2 & 3;
        const int32_t len1 = (int32_t)((tag2 >> 2) & 7) + 4, off1 = (int32_t)(((tag2 >> 5) << 8) | ((x >> 8) & 0xFF));
        const int32_t len2 = (int32_t)(tag2 >> 2) + 1, off2 = (int32_t)((x >> 8) & 0xFFFF);
        const int32_t cLen = kind == 1 ? len1 : len2, cOff = kind == 1 ? off1 : off2;
        const int32_t cNext = q + (kind == 1 ? 2 : 3);
        const bool atEnd = q >= inLimit;
        const bool isCopy = !atEnd && kind != 0;
        const bool copyFast = (kind == 1 || kind == 2) && q + 5 < inLimit && cOff > 0 && cOff <= S.op && S.op + cLen <= outLimit;
        const bool ok1copy = do1 && isCopy && copyFast;
        // not a fast copy behind the run: the run goes alone (tLit > 0), or -- nothing held -- the general path takes the element
        const bool alone = do1 && !ok1copy && tLit > 0;
        const bool gen1 = do1 && !ok1copy && tLit == 0 && !atEnd;
        const bool end1 = do1 && !ok1copy && tLit == 0 && atEnd;
        const bool seq1 = ok1copy || alone;
        sLit = seq1 ? tLit : sLit;
        sLitPos = seq1 ? nread + tStart : sLitPos;
        sMl = ok1copy ? cLen : (alone ? 0 : sMl);
        sOff = ok1copy ? cOff : (alone ? 0 : sOff);
        sK = seq1 ? 0 : sK;
        S.ip = ok1copy ? cNext : S.ip;
        S.op += ok1copy ? cLen : 0;
        phase = seq1 ? 2 : (do1 ? 0 : phase);
        finished |= end1 ? 1 : 0;
        if (gen0 || gen1) {  // (rare) one element, reading the stream directly
            int32_t rLen = 0, rOff = 0, rStart = 0;
            const int kindG = snappy_parse_general(in, S, inLimit, outLimit, rLen, rOff, rStart);
            sLit = kindG == 1 ? rLen : 0;
            sLitPos = nread + rStart;
            sMl = kindG == 2 ? rLen : 0;
            sOff = kindG == 2 ? rOff : 0;
            sK = 0;
            phase = kindG != 0 ? 2 : 0;
            finished |= S.st != 0 ? 1 : 0;
            fallback |= (kindG == 2 && rOff > 0xFFFF) ? 1 : 0;  // an offset beyond the record field: the ring decoder takes the block
            finished |= fallback;
            L.restart(S.ip + B, finished == 0 && S.ip + B >= L.issueV + 64);
        }
        // ---- phase 2: one piece (see lz4_parse2_kernel) ----
        const bool do2 = finished == 0 && phase == 2;
        const int32_t pl = sLit < 16 ? sLit : 16;
        const int32_t pm = sLit > 16 ? 0 : (sMl < 16 ? sMl : 16);
        const int32_t xm = 16 * sK + sOff;
        const int32_t o = sK > 0 ? sx::largest_multiple(sOff > 0 ? sOff : 1, xm < 65535 ? xm : 65535) : sOff;
        const int32_t skip = pl > 0 ? sLitPos - litEndPrev : 0;
        const bool fb = do2 && skip > sx::MAX_SKIP;
        const bool ok2 = do2 && !fb;
        rec[T] = ok2 ? sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, (uint32_t)skip) : 0ull;
        groupAny |= ok2 ? 1 : 0;
        fallback |= fb ? 1 : 0;
        litEndPrev = ok2 && pl > 0 ? sLitPos + pl : litEndPrev;
        sLitPos += ok2 ? pl : 0;
        sLit -= ok2 ? pl : 0;
        sMl -= ok2 ? pm : 0;
        sK += ok2 && pm > 0 ? 1 : 0;
        phase = (ok2 && sLit == 0 && sMl == 0) ? 0 : phase;
        finished |= fb ? 1 : 0;
        L.template issue<SLOT>(S.ip + B, finished == 0);
    };

    while (__ballot(finished == 0) != 0) {  // (uniform)
        groupAny = 0;
        trip(std::integral_constant<int, 0>{});
        trip(std::integral_constant<int, 1>{});
        trip(std::integral_constant<int, 2>{});
        trip(std::integral_constant<int, 3>{});
        trip(std::integral_constant<int, 4>{});
        trip(std::integral_constant<int, 5>{});
        trip(std::integral_constant<int, 6>{});
        trip(std::integral_constant<int, 7>{});
        // ---- the group leaves: a chunk for every lane that needs one (one atomic per wavefront), then one 64-byte piece per lane ----
        const bool flush = groupAny != 0 && fallback == 0;
        const bool need = flush && fill == sx::CHUNK_RECS;
        const unsigned long long nm = __ballot(need);
        if (nm != 0) {  // (uniform)
            int32_t base = 0;
            if (lane == __builtin_ctzll(nm)) {
                base = atomicAdd(&hdr->nextChunk, (int32_t)__popcll(nm));
            }
            base = sx::wave_bcast(base, __builtin_ctzll(nm));
            if (need) {
                const int32_t c = base + (int32_t)__popcll(nm & ((1ull << lane) - 1));
                if (c >= maxChunks) {  // the arena is exhausted: the ring decoder takes the block
                    fallback = 1;
                    finished = 1;
                }
                else {
                    if (chunk >= 0) {
                        arena[(int64_t)chunk * sx::CHUNK_SLOTS + sx::CHUNK_RECS] = (uint64_t)(uint32_t)c;  // link
                    }
                    else {
                        firstChunk = c;
                    }
                    chunk = c;
                    fill = 0;
                }
            }
        }
        if (flush && fallback == 0) {
            uint8_t* const dst = (uint8_t*)(arena + (int64_t)chunk * sx::CHUNK_SLOTS + fill);
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                st16(dst + 8 * k, u32x4{(uint32_t)rec[k], (uint32_t)(rec[k] >> 32), (uint32_t)rec[k + 1], (uint32_t)(rec[k + 1] >> 32)});
            }
            fill += 8;
            count += 8;
        }
    }
    if (!have && block < a.nBlocks) {  // (a batch assembled on the device may hold fewer blocks than the launch was sized for)
        only[block] = 0;
        meta[block].firstChunk = 0;
        meta[block].count = 0;
    }
    if (have) {
        if (fallback != 0) {
            only[block] = 1;
            meta[block].firstChunk = 0;
            meta[block].count = 0;
            atomicAdd(&hdr->fallbackBlocks, 1);
        }
        else {
            if (S.st == 0 && (int64_t)expected != (int64_t)S.op) {  // :61-65
                S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
                S.eo = 0;
            }
            only[block] = 0;
            meta[block].firstChunk = firstChunk < 0 ? 0 : firstChunk;
            meta[block].count = S.st == 0 ? count : 0;
            a.outLen[block] = S.st == 0 ? S.op : 0;
            a.status[block] = S.st;
            a.errOffset[block] = (int64_t)S.eo;
        }
    }
}

// ---- snappy_parse_wave_kernel: the parse pass with a WAVEFRONT per block (round 5), for batches of FEW blocks -- the Snappy counterpart of
// lz4_parse_wave_kernel (lz4_decompress_v7.hip has the reasons: a lane's serial chain costs a 64 KiB text block 7 ms when nothing else runs).
// Lane p of a window of 64 stream positions reads the bytes AS IF an element started at p: a run with its length in the tag and the 1- or 2-byte-offset
// copy behind it (one record, as in the lane parser), a run alone, or a copy alone; `next` is where the element behind that would begin.  The real
// elements are the chain 0 -> next[0] -> ...: a scalar loop of lane reads.  The lanes on the chain get their output positions from a scan, make the
// fast path's checks (those of snappy_parse2_kernel above) and store their records -- up to seven pieces each (60 + 64 bytes).  A run with length
// bytes, a copy with a 4-byte offset, a failing check and the stream's last 168 bytes end the chain and go through snappy_parse_general, the Java loop
// body check by check, the records of a long run written by the whole wavefront.  Same statuses, error offsets and output as the lane parser.
__global__ __launch_bounds__(64) void snappy_parse_wave_kernel(BatchArgs a, sx::ArenaHeader* hdr, sx::BlockMeta* meta, int32_t* only, uint64_t* arena, int32_t maxChunks, const int32_t* stats)
{
    if (stats != nullptr && snappy_pick(stats, a.nBlocks) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    using Stage = WaveStage<wp::SNAPPY_STAGE>;
    static_assert(64 * 7 <= sx::CHUNK_RECS, "a window's records reach into one new chunk at most");
    __shared__ __attribute__((aligned(16))) uint8_t stageLds[Stage::CAP + 16];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
    const int32_t inLen0 = uni(a.srcLen[block]);
    const int32_t outLimit = uni(a.dstCap[block]);
    SnappyParseState S;
    S.ip = 0;
    S.op = 0;
    S.st = 0;
    S.eo = 0;
    // readUncompressedLength :277-321 (uniform: every lane reads the same bytes)
    uint32_t expected = 0;
    int32_t nread = 0;
    for (int i = 0; i < 5; i++) {
        if (nread >= inLen0) {
            S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_TRUNCATED);
            S.eo = inLen0 - nread;
            break;
        }
        const uint32_t b = (uint32_t)uni((int32_t)in0[nread]);
        nread++;
        expected |= (b & 0x7f) << (7 * i);
        if ((b & 0x80) == 0) {
            break;
        }
        if (i == 4) {
            S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LEN_HIGH_BIT);
            S.eo = nread;
        }
    }
    if (S.st == 0 && (int32_t)expected < 0) {
        S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_INVALID_LENGTH);
        S.eo = 0;
    }
    if (S.st == 0 && (int64_t)expected > (int64_t)outLimit) {  // :49-50
        S.st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNAPPY_OUTPUT_TOO_SMALL);
        S.eo = 0;
    }
    bool finished = S.st != 0;  // (uniform)
    // uncompressAll :70-220 ; positions relative to the first byte after the varint
    const uint8_t* const in = in0 + (finished ? 0 : nread);
    const int32_t inLimit = finished ? 0 : inLen0 - nread;
    Stage W;
    W.lds = stageLds;
    W.in = in;
    W.inLimit = inLimit;
    W.b0 = -1;
    W.pend[0] = u32x4{0, 0, 0, 0};
    W.pend[1] = u32x4{0, 0, 0, 0};
    W.lane = lane;
    WaveRecordSink K;
    K.hdr = hdr;
    K.arena = arena;
    K.maxChunks = maxChunks;
    K.firstChunk = -1;
    K.chunk = -1;
    K.fill = sx::CHUNK_RECS;
    K.count = 0;
    K.fallback = false;
    K.fresh = -1;
    const int32_t fastOutLimit = outLimit - 8;
    int32_t litEndPrev = 0;  // (uniform) position (counted from in0) the executor's literal cursor stands at behind the records so far
    bool serial = false;                // (uniform) the elements are long: one pair at a time (lz4_parse_wave_kernel has the reasons)
    while (!finished && !K.fallback) {  // (uniform)
        bool general = true;
        const bool windowable = (int64_t)S.ip + wp::SNAPPY_STAGE + 24 <= (int64_t)inLimit;  // (uniform) nothing a window looks at can reach the stream's last bytes (the margins of lz4_parse_wave_kernel)
        if (windowable && serial) {
            // the element (pair) at the window's first position, read by every lane at once; lane k makes its piece k (at most seven)
            const int32_t base = S.ip;
            const uint8_t* const stage = W.window(base);
            uint32_t x;
            __builtin_memcpy(&x, stage, 4);
            const uint32_t tag = x & 0xFF;
            const bool isRun = (tag & 3) == 0;
            const int32_t nLit = uni(isRun ? (int32_t)(tag >> 2) + 1 : 0);
            const int32_t q = isRun ? 1 + nLit : 0;
            uint32_t y;
            __builtin_memcpy(&y, stage + q, 4);
            const uint32_t tag2 = y & 0xFF, kind2 = tag2 & 3;
            const bool isCopy = kind2 == 1 || kind2 == 2;
            const int32_t len1 = (int32_t)((tag2 >> 2) & 7) + 4, off1 = (int32_t)(((tag2 >> 5) << 8) | ((y >> 8) & 0xFF));
            const int32_t len2 = (int32_t)(tag2 >> 2) + 1, off2 = (int32_t)((y >> 8) & 0xFFFF);
            const int32_t cLen = uni(isCopy ? (kind2 == 1 ? len1 : len2) : 0), cOff = uni(kind2 == 1 ? off1 : off2);
            const int32_t next = q + (isCopy ? (kind2 == 1 ? 2 : 3) : 0);
            const bool stop = isRun ? (tag >> 2) >= 60 : (tag & 3) == 3;
            const int32_t opCopy = S.op + nLit, opEnd = opCopy + cLen;
            const int32_t skip = nread + base + (isRun ? 1 : 0) - litEndPrev;
            const bool ok = !stop && !(isRun && opCopy > fastOutLimit) && !(isCopy && (cOff == 0 || cOff > opCopy || opEnd > outLimit)) && skip <= sx::MAX_SKIP;
            serial = false;
            if (uni(ok ? 1 : 0) != 0) {  // (uniform)
                const int32_t litFull = nLit > 16 ? (nLit + 15) / 16 - 1 : 0;
                const int32_t matchRest = cLen > 16 ? (cLen - 16 + 15) / 16 : 0;
                const int32_t pieces = litFull + 1 + matchRest;
                const int32_t k = lane;
                int32_t pl, pm, o = cOff;
                if (k < litFull) {
                    pl = 16;
                    pm = 0;
                }
                else if (k == litFull) {
                    pl = nLit - 16 * litFull;
                    pm = cLen < 16 ? cLen : 16;
                }
                else {
                    const int32_t m = k - litFull;
                    pl = 0;
                    pm = cLen - 16 * m < 16 ? cLen - 16 * m : 16;
                    const int32_t xm = 16 * m + cOff;
                    o = sx::largest_multiple(cOff > 0 ? cOff : 1, xm < 65535 ? xm : 65535);
                }
                K.put(sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, k == 0 ? (uint32_t)skip : 0u), k < pieces, lane, pieces, lane);
                if (!K.fallback) {
                    S.op = opEnd;
                    litEndPrev = nread + base + q;
                    S.ip = base + next;
                    general = false;
                    serial = next >= 24;
                }
            }
        }
        else if (windowable) {
            const int32_t base = S.ip;
            const uint8_t* const stage = W.window(base);
            // what an element at position `lane` of the window would be
            uint32_t x;
            __builtin_memcpy(&x, stage + lane, 4);
            const uint32_t tag = x & 0xFF;
            const bool isRun = (tag & 3) == 0;
            const int32_t nLit = isRun ? (int32_t)(tag >> 2) + 1 : 0;  // (a run with its length in the tag)
            const int32_t q = lane + (isRun ? 1 + nLit : 0);           // the element behind the run (<= 124); the copy itself when there is no run
            uint32_t y;
            __builtin_memcpy(&y, stage + q, 4);
            const uint32_t tag2 = y & 0xFF, kind2 = tag2 & 3;
            const bool isCopy = kind2 == 1 || kind2 == 2;
            const int32_t len1 = (int32_t)((tag2 >> 2) & 7) + 4, off1 = (int32_t)(((tag2 >> 5) << 8) | ((y >> 8) & 0xFF));
            const int32_t len2 = (int32_t)(tag2 >> 2) + 1, off2 = (int32_t)((y >> 8) & 0xFFFF);
            const int32_t cLen = isCopy ? (kind2 == 1 ? len1 : len2) : 0, cOff = kind2 == 1 ? off1 : off2;
            const int32_t next = q + (isCopy ? (kind2 == 1 ? 2 : 3) : 0);
            // not for the chain: a run with length bytes, a copy with a 4-byte offset
            const bool stop = isRun ? (tag >> 2) >= 60 : (tag & 3) == 3;
            const unsigned long long stopMask = __ballot(stop);
            unsigned long long members = 0;
            int32_t cur = 0;
            wave_chain(next, stop, stopMask, lane, members, cur);
            // the members' places in the output, and the checks that need them (the lane parser's runFast / copyFast; the input-side conditions hold in a window)
            const bool member = ((members >> lane) & 1ull) != 0;
            const int32_t litFull = nLit > 16 ? (nLit + 15) / 16 - 1 : 0;
            const int32_t matchRest = cLen > 16 ? (cLen - 16 + 15) / 16 : 0;
            // one scan for both: output bytes in the low half (at most 64 x 124), pieces in the high half (at most 64 x 7)
            const int32_t scanned = sx::wave_scan_incl(member ? ((nLit + cLen) | ((litFull + 1 + matchRest) << 16)) : 0, lane);
            const int32_t endRel = scanned & 0xFFFF, pieceEnd = scanned >> 16;
            const int32_t opEnd = S.op + endRel, opCopy = opEnd - cLen;
            const bool wrong = member && ((isRun && opCopy > fastOutLimit) || (isCopy && (cOff == 0 || cOff > opCopy || opEnd > outLimit)));
            const unsigned long long wrongMask = __ballot(wrong);
            if (wrongMask != 0) {  // (uniform) the chain ends in front of the first such element
                const int first = __builtin_ctzll(wrongMask);
                members &= (1ull << first) - 1ull;
                cur = first;
            }
            const bool mine = ((members >> lane) & 1ull) != 0;
            if (members != 0) {  // (uniform)
                const unsigned long long below = members & ((1ull << lane) - 1ull);
                const int prevLane = below != 0 ? 63 - __builtin_clzll(below) : 0;
                const int32_t prevQ = __shfl(q, prevLane);
                const int32_t litStart = nread + base + lane + (isRun ? 1 : 0);
                const int32_t skip = litStart - (below != 0 ? nread + base + prevQ : litEndPrev);
                const int last = 63 - __builtin_clzll(members);
                const int32_t pieces = mine ? litFull + 1 + matchRest : 0;
                const int32_t n = sx::wave_bcast(pieceEnd, last);
                if (__ballot(mine && skip > sx::MAX_SKIP) != 0) {  // (a gap beyond the record field) the ring decoder takes the block
                    K.fallback = true;
                }
                else if (K.begin(n, lane)) {
                    for (int32_t k = 0; __ballot(k < pieces) != 0; k++) {  // (uniform) piece k of every element that has one: at most seven rounds, one or two on text
                        int32_t pl, pm, o = cOff;
                        if (k < litFull) {
                            pl = 16;
                            pm = 0;
                        }
                        else if (k == litFull) {
                            pl = nLit - 16 * litFull;
                            pm = cLen < 16 ? cLen : 16;
                        }
                        else {
                            const int32_t m = k - litFull;  // copy pieces before this one
                            pl = 0;
                            pm = cLen - 16 * m < 16 ? cLen - 16 * m : 16;
                            const int32_t xm = 16 * m + cOff;
                            o = sx::largest_multiple(cOff > 0 ? cOff : 1, xm < 65535 ? xm : 65535);
                        }
                        K.store(sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, k == 0 ? (uint32_t)skip : 0u), k < pieces, pieceEnd - pieces + k);
                    }
                    K.end(n);
                    S.op += sx::wave_bcast(endRel, last);
                    litEndPrev = nread + base + sx::wave_bcast(q, last);
                    S.ip = base + cur;
                    general = false;
                    serial = cur >= 48 && __popcll(members) <= 2;
                }
            }
        }
        if (general && !K.fallback) {  // (uniform) one element the Java way, its records by the whole wavefront
            if (S.ip >= inLimit) {  // the loop condition :84
                finished = true;
            }
            else {
                int32_t rLen = 0, rOff = 0, rStart = 0;
                const int kindG = snappy_parse_general(in, S, inLimit, outLimit, rLen, rOff, rStart);
                if (S.st != 0) {
                    finished = true;
                }
                else if (kindG == 2 && rOff > 0xFFFF) {  // an offset beyond the record field: the ring decoder takes the block
                    K.fallback = true;
                }
                else if (kindG != 0) {
                    const int32_t sLit = kindG == 1 ? rLen : 0, sMl = kindG == 2 ? rLen : 0, sOff = kindG == 2 ? rOff : 0;
                    const int32_t litFull = sLit > 16 ? (sLit + 15) / 16 - 1 : 0;
                    const int32_t matchRest = sMl > 16 ? (sMl - 16 + 15) / 16 : 0;
                    const int32_t pieces = litFull + 1 + matchRest;
                    const int32_t skip0 = nread + rStart - litEndPrev;
                    if (skip0 > sx::MAX_SKIP) {
                        K.fallback = true;
                    }
                    for (int32_t k0 = 0; k0 < pieces && !K.fallback; k0 += 64) {  // (uniform)
                        const int32_t k = k0 + lane;
                        int32_t pl, pm, o = sOff;
                        if (k < litFull) {
                            pl = 16;
                            pm = 0;
                        }
                        else if (k == litFull) {
                            pl = sLit - 16 * litFull;
                            pm = sMl < 16 ? sMl : 16;
                        }
                        else {
                            const int32_t m = k - litFull;
                            pl = 0;
                            pm = sMl - 16 * m < 16 ? sMl - 16 * m : 16;
                            const int32_t xm = 16 * m + sOff;
                            o = sx::largest_multiple(sOff > 0 ? sOff : 1, xm < 65535 ? xm : 65535);
                        }
                        const int32_t left = pieces - k0;
                        K.put(sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, k == 0 ? (uint32_t)skip0 : 0u), k < pieces, lane, left < 64 ? left : 64, lane);
                    }
                    litEndPrev = nread + rStart + sLit;
                }
            }
        }
    }
    if (lane == 0) {
        if (K.fallback) {
            only[block] = 1;
            meta[block].firstChunk = 0;
            meta[block].count = 0;
            atomicAdd(&hdr->fallbackBlocks, 1);
        }
        else {
            if (S.st == 0 && (int64_t)expected != (int64_t)S.op) {  // :61-65
                S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LENGTH_MISMATCH);
                S.eo = 0;
            }
            only[block] = 0;
            meta[block].firstChunk = K.firstChunk < 0 ? 0 : K.firstChunk;
            meta[block].count = S.st == 0 ? K.count : 0;
            a.outLen[block] = S.st == 0 ? S.op : 0;
            a.status[block] = S.st;
            a.errOffset[block] = (int64_t)S.eo;
        }
    }
}

// which of the two parsers: 0 = by the batch (a wavefront per block up to 32 768 blocks with a count known to the host, a lane per block above), 1 = a lane per block,
// 2 = a wavefront per block (context option snappy.decompress.parse).  Corpus blocks of 64 KiB, GiB/s lane / wavefront (profiles/r05_parsesweep.txt): 4 096 blocks 27 / 106;
// 8 192: 53 / 129; 16 384: 98 / 156; 32 768: 157 / 173; 65 536: 236 / 182.
constexpr int32_t SNAPPY_WAVE_PARSE_MAX_BLOCKS = 32768;
int g_snappy_parse_mode = 0;

hipError_t launch_seq_execute2(const BatchArgs& a, hipStream_t stream, const sx::BlockMeta* meta, const uint64_t* arena, int execVariant, const int32_t* stats, int32_t shortLimit);
int64_t twopass_scratch_bytes(int32_t nBlocks, int64_t perBlock);
hipError_t launch_snappy_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);

int64_t snappy_twopass_scratch_bytes(int32_t nBlocks) { return twopass_scratch_bytes(nBlocks, 131072); }

hipError_t launch_snappy_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* s = (uint8_t*)scratch;
    sx::ArenaHeader* hdr = (sx::ArenaHeader*)s;
    sx::BlockMeta* meta = (sx::BlockMeta*)(s + 4096);
    int32_t* only = (int32_t*)(s + 4096 + (int64_t)a.nBlocks * 8);
    const int64_t fixed = 4096 + (((int64_t)a.nBlocks * 12 + 4095) & ~4095LL);
    uint64_t* arena = (uint64_t*)(s + fixed);
    const int64_t chunks = (scratchBytes - fixed) / (sx::CHUNK_SLOTS * 8) - 1;  // (one to spare: the second executor's unconditional record loads)
    const int32_t maxChunks = (int32_t)(chunks > 0x7FFFFFFF ? 0x7FFFFFFF : chunks);
    hipError_t e = hipMemsetAsync(hdr, 0, sizeof(sx::ArenaHeader), stream);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((a.nBlocks + 63) / 64)), wg(64);
    {
        const bool wavePerBlock = a.nBlocksDev == nullptr && (g_snappy_parse_mode == 2 || (g_snappy_parse_mode == 0 && a.nBlocks <= SNAPPY_WAVE_PARSE_MAX_BLOCKS));
        if (wavePerBlock) {
            hipLaunchKernelGGL(snappy_parse_wave_kernel, dim3((unsigned)a.nBlocks), wg, 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
        }
        else {
            hipLaunchKernelGGL(snappy_parse2_kernel, grid, wg, 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
        }
        e = launch_seq_execute2(a, stream, meta, arena, execVariant, stats, 6);
        if (e != hipSuccess) return e;
    }
    BatchArgs f = a;
    f.only = only;
    f.onlyStats = stats;
    f.onlyShortLimit = 6;
    e = launch_snappy_decompress_rings(f, stream, groupSize, ringClass, nullptr);
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace achip
