// zstd_decompress.hip -- batched Zstd frame decode for gfx950.
//
// Replaces ZstdFrameDecompressor.decompress and everything under it:
//   frames / blocks ....... M/zstd/ZstdFrameDecompressor.java:135-310,860-962
//   literals .............. M/zstd/ZstdFrameDecompressor.java:708-858, M/zstd/Huffman.java:52-324
//   FSE table descriptions  M/zstd/FseTableReader.java:27-168, M/zstd/FseCompressionTable.java:133-154
//   Huffman weight stream . M/zstd/FiniteStateEntropy.java:38-151
//   backward bit streams .. M/zstd/BitInputStream.java:28-206
//   sequences ............. M/zstd/ZstdFrameDecompressor.java:312-516 (decode), :518-607,678-706 (execute)
//   checksum .............. M/zstd/XxHash64.java:182-291
//
// One wavefront per batch item (an input buffer holding one or more frames).  Everything that is
// serial in the format -- headers, table descriptions, the three-state FSE sequence stream -- is
// executed wave-uniformly (all 64 lanes compute the same values from the same addresses, so there
// is no cross-lane traffic); the four Huffman streams run on four lanes; everything that moves
// bytes (literal / match copies, RLE and raw blocks, the XXH64 stripes) is spread over the lanes.
// Decoding tables live in LDS (Huffman 8 KiB, three FSE tables 6 KiB, a 1024-sequence ring 8 KiB);
// the regenerated literals of the current block live in a per-wave scratch slab in HBM (128 KiB + 64).
// Checks are made in the order the Java code makes them, so status + detail equal the Java exception.
#include "zstd_dec_common.h"

namespace achip {

namespace zd {

struct FrameState {
    int32_t prevOffsets[3];
    int32_t hufTableLog;           // -1 = no table loaded (persists across frames, like the Java Huffman object)
    int32_t curLog[3];             // current table log per stream; -1 = none
    const FseTable* cur[3];
};

// ---- literals sections; return bytes consumed or -1; set litPtr/litSize ----
__device__ int32_t decode_literals(Ctx& c, Shared& sh, FrameState& fs, const FseTable* dflt, int32_t input, int32_t blockSize, const uint8_t** litPtr, int32_t* litSize)
{
    (void)dflt;
    const int32_t inputAddress = input;
    const int32_t inputLimit = input + blockSize;
    const int32_t b0 = (int32_t)rd_le(c, input, 1);
    const int32_t literalsBlockType = b0 & 3;
    const int32_t type = (b0 >> 2) & 3;
    if (literalsBlockType == 0) {  // decodeRawLiterals :812-858
        int32_t literalSize;
        if (type == 0 || type == 2) {
            literalSize = b0 >> 3;
            input += 1;
        }
        else if (type == 1) {
            literalSize = (int32_t)rd_le(c, input, 2) >> 4;
            input += 2;
        }
        else {
            literalSize = (int32_t)rd_le(c, input, 3) >> 4;
            input += 3;
        }
        ZVERIFY(c, input + literalSize <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        *litPtr = c.in + input;
        *litSize = literalSize;
        return input + literalSize - inputAddress;
    }
    if (literalsBlockType == 1) {  // decodeRleLiterals :776-810
        int32_t outputSize;
        if (type == 0 || type == 2) {
            outputSize = b0 >> 3;
            input += 1;
        }
        else if (type == 1) {
            outputSize = (int32_t)rd_le(c, input, 2) >> 4;
            input += 2;
        }
        else {
            ZVERIFY(c, blockSize >= 4, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            outputSize = (int32_t)(rd_le(c, input, 4) & 0xFFFFFF) >> 4;
            input += 3;
        }
        ZVERIFY(c, outputSize <= MAX_BLOCK_SIZE, ACHIP_D_ZSTD_LITERALS_TOO_LARGE, input);
        const int32_t value = (int32_t)rd_le(c, input++, 1);
        wave_fill(c.lit, value, outputSize, c.lane);
        *litPtr = c.lit;
        *litSize = outputSize;
        return input - inputAddress;
    }
    // compressed (2) / treeless (3): decodeCompressedLiterals :708-774
    if (literalsBlockType == 3) {
        ZVERIFY(c, fs.hufTableLog != -1, ACHIP_D_ZSTD_DICT_CORRUPTED, input);
    }
    ZVERIFY(c, blockSize >= 5, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    int32_t compressedSize, uncompressedSize, headerSize;
    bool singleStream = false;
    if (type == 0 || type == 1) {
        singleStream = type == 0;
        const uint32_t header = (uint32_t)rd_le(c, input, 4);
        headerSize = 3;
        uncompressedSize = (int32_t)((header >> 4) & 0x3FF);
        compressedSize = (int32_t)((header >> 14) & 0x3FF);
    }
    else if (type == 2) {
        const uint32_t header = (uint32_t)rd_le(c, input, 4);
        headerSize = 4;
        uncompressedSize = (int32_t)((header >> 4) & 0x3FFF);
        compressedSize = (int32_t)((header >> 18) & 0x3FFF);
    }
    else {
        const uint64_t header = rd_le(c, input, 5);
        headerSize = 5;
        uncompressedSize = (int32_t)((header >> 4) & 0x3FFFF);
        compressedSize = (int32_t)((header >> 22) & 0x3FFFF);
    }
    ZVERIFY(c, uncompressedSize <= MAX_BLOCK_SIZE, ACHIP_D_ZSTD_LITERALS_TOO_LARGE, input);
    ZVERIFY(c, headerSize + compressedSize <= blockSize, ACHIP_D_ZSTD_CORRUPTED, input);
    input += headerSize;
    const int32_t streamsLimit = input + compressedSize;
    if (literalsBlockType != 3) {
        int32_t tl = 0;
        const int32_t n = huf_read_table(c, sh, input, compressedSize, &tl);
        if (n < 0) return -1;
        fs.hufTableLog = tl;
        input += n;
    }
    const int32_t tableLog = fs.hufTableLog;
    *litPtr = c.lit;
    *litSize = uncompressedSize;

    // stream boundaries (decode4Streams :168-176) -- lanes 0..3 own one stream each
    int32_t sStart[4], sEnd[4], oStart[4], oEnd[4];
    int nStreams;
    if (singleStream) {
        nStreams = 1;
        sStart[0] = input;
        sEnd[0] = streamsLimit;
        oStart[0] = 0;
        oEnd[0] = uncompressedSize;
    }
    else {
        nStreams = 4;
        ZVERIFY(c, streamsLimit - input >= 10, ACHIP_D_ZSTD_CORRUPTED, input);
        const int32_t start1 = input + 6;
        const int32_t start2 = start1 + (int32_t)rd_le(c, input, 2);
        const int32_t start3 = start2 + (int32_t)rd_le(c, input + 2, 2);
        const int32_t start4 = start3 + (int32_t)rd_le(c, input + 4, 2);
        ZVERIFY(c, start2 < start3 && start3 < start4 && start4 < streamsLimit, ACHIP_D_ZSTD_CORRUPTED, input);
        const int32_t seg = (uncompressedSize + 3) / 4;
        sStart[0] = start1; sEnd[0] = start2;
        sStart[1] = start2; sEnd[1] = start3;
        sStart[2] = start3; sEnd[2] = start4;
        sStart[3] = start4; sEnd[3] = streamsLimit;
        oStart[0] = 0; oEnd[0] = seg;
        oStart[1] = seg; oEnd[1] = 2 * seg;
        oStart[2] = 2 * seg; oEnd[2] = 3 * seg;
        oStart[3] = 3 * seg; oEnd[3] = uncompressedSize;
    }
    // initialise the bit streams in order (the Java code raises the first failing initialiser)
    Bits mine;
    mine.start = mine.current = 0;
    mine.bits = 0;
    mine.consumed = 0;
    mine.overflow = false;
    int32_t myOutStart = 0, myOutEnd = 0, myStreamStart = 0;
    for (int s = 0; s < nStreams; s++) {
        Bits b;
        int32_t eo = 0;
        const int32_t d = bit_init(c, b, sStart[s], sEnd[s], &eo);
        if (d != 0) ZFAIL(c, d, eo);
        if (c.lane == s) {
            mine = b;
            myOutStart = oStart[s];
            myOutEnd = oEnd[s];
            myStreamStart = sStart[s];
        }
    }
    if (!singleStream) {
        // the lock-step loop's post-condition (:273) can only fail if the last segment is over-long
        ZVERIFY(c, oStart[3] <= uncompressedSize || uncompressedSize < 0, ACHIP_D_ZSTD_CORRUPTED, input);
    }
    int32_t myDetail = 0;
    if (c.lane < nStreams) {
        if (myOutStart <= myOutEnd) {
            myDetail = huf_decode_stream(c, sh.huf, tableLog, mine, c.lit, myOutStart, myOutEnd);
        }
        else {
            myDetail = ACHIP_D_ZSTD_CORRUPTED;
        }
    }
    const unsigned long long bad = __ballot(myDetail != 0);
    if (bad != 0) {
        const int first = __builtin_ctzll(bad);
        const int32_t d = __shfl(myDetail, first);
        const int32_t eo = __shfl(myStreamStart, first);
        ZFAIL(c, d, d == ACHIP_D_ZSTD_CORRUPTED ? input : eo);
    }
    return headerSize + compressedSize;
}

// computeLiteralsTable / computeOffsetsTable / computeMatchLengthTable :609-676 ; returns new input or -1
__device__ int32_t compute_table(Ctx& c, Shared& sh, FrameState& fs, int which, int32_t type, int32_t input, int32_t inputLimit, const FseTable* dflt, int32_t dfltLog,
                                 int32_t maxSymbol, int32_t maxLog)
{
    if (type == 1) {
        ZVERIFY(c, input < inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        const int32_t value = (int8_t)rd_le(c, input++, 1);
        ZVERIFY(c, value <= maxSymbol, ACHIP_D_ZSTD_VALUE_TOO_LARGE, input);
        ZVERIFY(c, value >= 0, ACHIP_D_ZSTD_CORRUPTED, input);  // Java: negative array index later
        __syncthreads();
        if (c.lane == 0) {
            sh.fse[which].e[0] = (uint32_t)value << 16;  // initializeRleTable: newState 0, bits 0
        }
        __syncthreads();
        fs.cur[which] = &sh.fse[which];
        fs.curLog[which] = 0;
    }
    else if (type == 0) {
        fs.cur[which] = dflt;
        fs.curLog[which] = dfltLog;
    }
    else if (type == 3) {
        ZVERIFY(c, fs.curLog[which] >= 0, ACHIP_D_ZSTD_TABLE_MISSING, input);
    }
    else {
        int32_t log = 0;
        const int32_t n = read_fse_table(c, sh, sh.fse[which], input, inputLimit, maxSymbol, maxLog, &log);
        if (n < 0) return -1;
        input += n;
        fs.cur[which] = &sh.fse[which];
        fs.curLog[which] = log;
    }
    return input;
}


// ---- byte-parallel execution of up to 64 decoded sequences (ZstdFrameDecompressor.java:488-509 per sequence) ----
// Lane i first owns sequence i: two wave scans give every sequence its literal and output start, and the Java
// validity checks (:491-496) are evaluated for all of them at once -- the lowest failing lane / first failing
// check is what the sequential Java loop would have thrown.  Then the batch's output span is produced 64 bytes
// per step, one byte per lane: a binary search over the scanned starts tells the lane which sequence its byte
// belongs to and whether it is a literal (source = literal stream) or a match byte (source = output - offset).
// Match bytes whose source lies inside the same 64-byte step are resolved by pointer jumping over the lanes
// (<= 6 rounds, each round doubles the resolved distance) -- no byte waits for another byte.
__device__ int32_t execute_batch(Ctx& c, Shared& sh, int32_t i0, int32_t n, int32_t& output, int32_t& literalsInput, int32_t litSize, int32_t input)
{
    ZRings& R = *c.R;
    const int lane = c.lane;
    const int32_t outputLimit = c.outCap;
    int32_t ll = 0, ml = 0, of = 0;
    if (lane < n) {
        const uint64_t s = sh.seq[i0 + lane];
        ll = (int32_t)(s & 0x1FFFF);
        ml = (int32_t)((s >> 17) & 0x3FFFF);
        of = (int32_t)(s >> 35);  // 29 bits: the largest offset the format can express is 0x1FFFFFFC (code 28)
    }
    // inclusive scans of litLen and litLen + matchLen
    int32_t lsum = ll, osum = ll + ml;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t a = __shfl_up(lsum, d);
        const int32_t b = __shfl_up(osum, d);
        if (lane >= d) {
            lsum += a;
            osum += b;
        }
    }
    const int32_t lp = lsum - ll;          // literals consumed before this sequence (within the batch)
    const int32_t op = osum - (ll + ml);   // output produced before this sequence (within the batch)
    // the three checks, in the Java order, for every sequence at once
    int32_t err = 0;
    if (lane < n) {
        if ((int64_t)output + osum > outputLimit) {
            err = 1;  // "Output buffer too small"
        }
        else if ((int64_t)literalsInput + lsum > litSize) {
            err = 2;  // "Input is corrupted" (literals exhausted)
        }
        else if (of <= 0 || (int64_t)output + op + ll - of < 0) {
            err = 3;  // "Input is corrupted" (match before the start of the call's output)
        }
    }
    const unsigned long long bad = __ballot(err != 0);
    if (bad != 0) {
        const int32_t e = __shfl(err, __builtin_ctzll(bad));
        ZFAIL(c, e == 1 ? ACHIP_D_ZSTD_OUTPUT_TOO_SMALL : ACHIP_D_ZSTD_CORRUPTED, input);
    }
    const int32_t span = __shfl(osum, 63);
    const int32_t litTotal = __shfl(lsum, 63);
    __syncthreads();
    sh.bS[lane] = lane < n ? op : span;
    sh.bLL[lane] = ll;
    sh.bLP[lane] = lp;
    sh.bOF[lane] = of;
    if (lane == 0) {
        sh.bS[64] = span;
    }
    __syncthreads();

    for (int32_t chunk = 0; chunk < span; chunk += 64) {
        const int32_t p = chunk + lane;
        const bool active = p < span;
        // which sequence: largest k with bS[k] <= p
        int32_t k = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            if (sh.bS[k + step] <= p) {
                k += step;
            }
        }
        const int32_t r = p - sh.bS[k];
        const int32_t kll = sh.bLL[k];
        const bool isLit = r < kll;
        const int32_t litBefore = sh.bLP[k] + (isLit ? r : kll);  // literal bytes of the batch consumed before byte p
        // source descriptor: literal -> position in the literal stream; match -> absolute output position
        int32_t src = isLit ? literalsInput + litBefore : (output + p) - sh.bOF[k];
        int32_t kind = isLit ? 0 : 1;
        const int32_t chunkAbs = output + chunk;
        // pointer jumping for match bytes whose source is inside this 64-byte step
        for (;;) {
            const bool pending = active && kind == 1 && src >= chunkAbs && src < chunkAbs + lane;  // a source is always an earlier lane
            if (__ballot(pending) == 0) {
                break;
            }
            const int from = pending ? (src - chunkAbs) : lane;
            const int32_t nsrc = __shfl(src, from);
            const int32_t nkind = __shfl(kind, from);
            if (pending) {
                src = nsrc;
                kind = nkind;
            }
        }
        // literal bytes come from the input ring: make the step's literal window resident (monotone cursor)
        const int32_t litCursor = literalsInput + __shfl(litBefore, 0);
        R.ensure_input(litCursor, 64);
        uint32_t byte = 0;
        if (active) {
            if (kind == 0) {
                byte = R.in_u8(src);
            }
            else if (chunkAbs - src <= ZRings::LDS_REACH) {
                byte = R.out_get(src);
            }
            else {
                byte = R.outAligned[R.outBase + src];  // far back-reference: flushed long ago
            }
        }
        wave_mem_order();
        if (active) {
            R.out_put(output + p, byte);
        }
        const int32_t done = chunk + 64 < span ? chunk + 64 : span;
        R.flush_complete(output + done);
    }
    output += span;
    literalsInput += litTotal;
    return 0;
}

// decompressSequences :312-516 ; returns decoded size or -1
__device__ int32_t decompress_sequences(Ctx& c, Shared& sh, FrameState& fs, const FseTable* dflt, int32_t inputAddress, int32_t inputLimit, int32_t outputAddress,
                                        const uint8_t* litPtr, int32_t litSize)
{
    const int32_t outputLimit = c.outCap;
    int32_t input = inputAddress;
    int32_t output = outputAddress;
    int32_t literalsInput = 0;
    const int32_t size = inputLimit - inputAddress;
    ZVERIFY(c, size >= 1, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);

    c.R->reset_input(litPtr, litSize);
    int32_t sequenceCount = (int32_t)rd_le(c, input++, 1);
    if (sequenceCount != 0) {
        if (sequenceCount == 255) {
            ZVERIFY(c, input + 2 <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            sequenceCount = (int32_t)rd_le(c, input, 2) + 0x7F00;
            input += 2;
        }
        else if (sequenceCount > 127) {
            ZVERIFY(c, input < inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            sequenceCount = ((sequenceCount - 128) << 8) + (int32_t)rd_le(c, input++, 1);
        }
        ZVERIFY(c, input + 4 <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        const int32_t type = (int32_t)rd_le(c, input++, 1);
        input = compute_table(c, sh, fs, 0, type >> 6, input, inputLimit, &dflt[0], 6, 35, 9);
        if (input < 0) return -1;
        input = compute_table(c, sh, fs, 1, (type >> 4) & 3, input, inputLimit, &dflt[1], 5, 28, 8);
        if (input < 0) return -1;
        input = compute_table(c, sh, fs, 2, (type >> 2) & 3, input, inputLimit, &dflt[2], 6, 52, 9);
        if (input < 0) return -1;

        Bits b;
        {
            int32_t eo = 0;
            const int32_t d = bit_init(c, b, input, inputLimit, &eo);
            if (d != 0) ZFAIL(c, d, eo);
        }
        const FseTable* llt = fs.cur[0];
        const FseTable* oft = fs.cur[1];
        const FseTable* mlt = fs.cur[2];
        int32_t llState = (int32_t)peek_bits(b.consumed, b.bits, fs.curLog[0]);
        b.consumed += fs.curLog[0];
        int32_t ofState = (int32_t)peek_bits(b.consumed, b.bits, fs.curLog[1]);
        b.consumed += fs.curLog[1];
        int32_t mlState = (int32_t)peek_bits(b.consumed, b.bits, fs.curLog[2]);
        b.consumed += fs.curLog[2];
        int32_t p0 = fs.prevOffsets[0], p1 = fs.prevOffsets[1], p2 = fs.prevOffsets[2];

        bool streamEnded = false;
        while (sequenceCount > 0 && !streamEnded) {
            // ---- decode up to SEQ_RING sequences (wave-uniform serial) ----
            int32_t nDecoded = 0;
            bool notConsumed = false;
            __syncthreads();
            while (sequenceCount > 0 && nDecoded < SEQ_RING) {
                sequenceCount--;
                b.overflow = false;
                bit_load(c, b);
                if (b.overflow) {
                    if (sequenceCount != 0) {
                        notConsumed = true;
                    }
                    streamEnded = true;
                    break;
                }
                const uint32_t lle = llt->e[llState], mle = mlt->e[mlState], ofe = oft->e[ofState];
                const int32_t llCode = FSE_SYMBOL(lle), mlCode = FSE_SYMBOL(mle), ofCode = FSE_SYMBOL(ofe);
                const int32_t llBits = LL_BITS[llCode], mlBits = ML_BITS[mlCode], ofBits = ofCode;
                int32_t offset = OF_BASE[ofCode];
                if (ofCode > 0) {
                    offset += (int32_t)peek_bits(b.consumed, b.bits, ofBits);
                    b.consumed += ofBits;
                }
                if (ofCode <= 1) {
                    if (llCode == 0) {
                        offset++;
                    }
                    if (offset != 0) {
                        int32_t temp = offset == 3 ? p0 - 1 : (offset == 1 ? p1 : p2);
                        if (temp == 0) {
                            temp = 1;
                        }
                        if (offset != 1) {
                            p2 = p1;
                        }
                        p1 = p0;
                        p0 = temp;
                        offset = temp;
                    }
                    else {
                        offset = p0;
                    }
                }
                else {
                    p2 = p1;
                    p1 = p0;
                    p0 = offset;
                }
                int32_t matchLength = ML_BASE[mlCode];
                if (mlCode > 31) {
                    matchLength += (int32_t)peek_bits(b.consumed, b.bits, mlBits);
                    b.consumed += mlBits;
                }
                int32_t literalsLength = LL_BASE[llCode];
                if (llCode > 15) {
                    literalsLength += (int32_t)peek_bits(b.consumed, b.bits, llBits);
                    b.consumed += llBits;
                }
                if (llBits + mlBits + ofBits > 64 - 7 - (9 + 9 + 8)) {
                    bit_load(c, b);
                }
                int32_t nb = FSE_NBITS(lle);
                llState = FSE_NEWSTATE(lle) + (int32_t)peek_bits(b.consumed, b.bits, nb);
                b.consumed += nb;
                nb = FSE_NBITS(mle);
                mlState = FSE_NEWSTATE(mle) + (int32_t)peek_bits(b.consumed, b.bits, nb);
                b.consumed += nb;
                nb = FSE_NBITS(ofe);
                ofState = FSE_NEWSTATE(ofe) + (int32_t)peek_bits(b.consumed, b.bits, nb);
                b.consumed += nb;
                // an FSE state past the table is only reachable through corrupt tables; clamp so LDS reads stay in range
                llState &= 511;
                mlState &= 511;
                ofState &= 511;
                if (c.lane == 0) {
                    sh.seq[nDecoded] = (uint64_t)((uint32_t)literalsLength & 0x1FFFFu) | ((uint64_t)((uint32_t)matchLength & 0x3FFFFu) << 17) | ((uint64_t)((uint32_t)offset & 0x1FFFFFFFu) << 35);
                }
                nDecoded++;
            }
            __syncthreads();
            // ---- execute them, 64 sequences per batch, one output byte per lane per step ----
            for (int32_t i0 = 0; i0 < nDecoded; i0 += 64) {
                const int32_t nb = nDecoded - i0 < 64 ? nDecoded - i0 : 64;
                if (execute_batch(c, sh, i0, nb, output, literalsInput, litSize, input) < 0) {
                    return -1;
                }
            }
            if (notConsumed) {
                ZFAIL(c, ACHIP_D_ZSTD_SEQUENCES_NOT_CONSUMED, input);
            }
        }
        fs.prevOffsets[0] = p0;
        fs.prevOffsets[1] = p1;
        fs.prevOffsets[2] = p2;
    }
    // copyLastLiteral :518-525
    const int32_t last = litSize - literalsInput;
    ZVERIFY(c, (int64_t)output + last <= outputLimit, ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, input);
    c.R->copy_literals(literalsInput, output, last);
    output += last;
    return output - outputAddress;
}

__device__ int32_t decode_compressed_block(Ctx& c, Shared& sh, FrameState& fs, const FseTable* dflt, int32_t inputAddress, int32_t blockSize, int32_t outputAddress, int32_t windowSize)
{
    int32_t input = inputAddress;
    ZVERIFY(c, blockSize <= MAX_BLOCK_SIZE, ACHIP_D_ZSTD_BLOCK_TOO_LARGE, input);
    ZVERIFY(c, blockSize >= 3, ACHIP_D_ZSTD_BLOCK_TOO_SMALL, input);
    const uint8_t* litPtr = nullptr;
    int32_t litSize = 0;
    const int32_t n = decode_literals(c, sh, fs, dflt, input, blockSize, &litPtr, &litSize);
    if (n < 0) return -1;
    input += n;
    ZVERIFY(c, windowSize <= MAX_WINDOW_SIZE, ACHIP_D_ZSTD_WINDOW_TOO_LARGE, input);
    wave_mem_order();
    return decompress_sequences(c, sh, fs, dflt, input, inputAddress + blockSize, outputAddress, litPtr, litSize);
}

// ZstdFrameDecompressor.decompress :135-210 ; returns bytes written or -1
__device__ int32_t zstd_decompress_item(Ctx& c, Shared& sh, const FseTable* dflt)
{
    if (c.outCap == 0) {
        return 0;
    }
    const int32_t inputLimit = c.inLen;
    int32_t input = 0;
    int32_t output = 0;
    FrameState fs;
    fs.hufTableLog = -1;
    while (input < inputLimit) {
        fs.prevOffsets[0] = 1;  // reset() :212-221
        fs.prevOffsets[1] = 4;
        fs.prevOffsets[2] = 8;
        fs.curLog[0] = fs.curLog[1] = fs.curLog[2] = -1;
        fs.cur[0] = fs.cur[1] = fs.cur[2] = nullptr;
        const int32_t outputStart = output;

        // verifyMagic :949-962
        ZVERIFY(c, inputLimit - input >= 4, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        const uint32_t magic = (uint32_t)rd_le(c, input, 4);
        if (magic != 0xFD2FB528u) {
            ZFAIL(c, magic == 0xFD2FB527u ? ACHIP_D_ZSTD_V07_MAGIC : ACHIP_D_ZSTD_BAD_MAGIC, input);
        }
        input += 4;
        // readFrameHeader :860-940
        const int32_t headerAddress = input;
        ZVERIFY(c, input < inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        const int32_t fhd = (int32_t)rd_le(c, input++, 1);
        const bool singleSegment = (fhd & 0x20) != 0;
        const int32_t dictDesc = fhd & 3;
        const int32_t csDesc = fhd >> 6;
        const int32_t headerSize = 1 + (singleSegment ? 0 : 1) + (dictDesc == 0 ? 0 : (1 << (dictDesc - 1))) + (csDesc == 0 ? (singleSegment ? 1 : 0) : (1 << csDesc));
        ZVERIFY(c, headerSize <= inputLimit - headerAddress, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
        int32_t windowSize = -1;
        if (!singleSegment) {
            const int32_t wd = (int32_t)rd_le(c, input++, 1);
            const uint32_t base = 1u << ((10 + (wd >> 3)) & 31);
            windowSize = (int32_t)(base + (uint32_t)(((int32_t)base / 8) * (wd & 7)));
        }
        if (dictDesc != 0) {
            ZFAIL(c, ACHIP_D_ZSTD_DICTIONARY, input + (1 << (dictDesc - 1)));
        }
        input = headerAddress + headerSize;  // content size field is not needed for decoding
        const bool hasChecksum = (fhd & 4) != 0;

        bool lastBlock;
        do {
            ZVERIFY(c, input + 3 <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            const int32_t header = (int32_t)rd_le(c, input, 3);
            input += 3;
            lastBlock = (header & 1) != 0;
            const int32_t blockType = (header >> 1) & 3;
            const int32_t blockSize = (header >> 3) & 0x1FFFFF;
            int32_t decodedSize;
            if (blockType == 0) {  // decodeRawBlock :223-229
                ZVERIFY(c, (int64_t)input + blockSize <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
                ZVERIFY(c, (int64_t)output + blockSize <= c.outCap, ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, input);
                c.R->reset_input(c.in + input, blockSize);
                c.R->copy_literals(0, output, blockSize);
                decodedSize = blockSize;
                input += blockSize;
            }
            else if (blockType == 1) {  // decodeRleBlock :231-263
                ZVERIFY(c, input + 1 <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
                ZVERIFY(c, (int64_t)output + blockSize <= c.outCap, ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, input);
                c.R->fill(output, c.in[input], blockSize);
                decodedSize = blockSize;
                input += 1;
            }
            else if (blockType == 2) {
                ZVERIFY(c, (int64_t)input + blockSize <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
                decodedSize = decode_compressed_block(c, sh, fs, dflt, input, blockSize, output, windowSize);
                if (decodedSize < 0) return -1;
                input += blockSize;
            }
            else {
                ZFAIL(c, ACHIP_D_ZSTD_INVALID_BLOCK_TYPE, input);
            }
            output += decodedSize;
            wave_mem_order();
        } while (!lastBlock);

        if (hasChecksum) {
            // flush the ring, then all of this frame's stores are visible to the hashing loads: same wave, program order
            c.R->flush_all(output);
            const uint64_t hash = wave_xxh64(c.out + outputStart, output - outputStart, c.lane);
            ZVERIFY(c, input + 4 <= inputLimit, ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            const uint32_t checksum = (uint32_t)rd_le(c, input, 4);
            if (checksum != (uint32_t)hash) {
                ZFAIL(c, ACHIP_D_ZSTD_BAD_CHECKSUM, input);
            }
            input += 4;
        }
    }
    c.R->flush_all(output);
    return output;
}

}  // namespace zd

// builds the three predefined tables once per launch (one wave), into global memory
__global__ __launch_bounds__(64) void zstd_default_tables_kernel(zd::FseTable* dflt)
{
    using namespace zd;
    __shared__ Shared sh;
    Ctx c;
    c.in = nullptr;
    c.inLen = 0;
    c.out = nullptr;
    c.outCap = 0;
    c.lit = nullptr;
    c.R = nullptr;
    c.lane = threadIdx.x;
    c.detail = 0;
    c.errOff = 0;
    const int16_t* norms[3] = {LL_DEFAULT_NORM, OF_DEFAULT_NORM, ML_DEFAULT_NORM};
    const int32_t maxSym[3] = {35, 28, 52};
    const int32_t logs[3] = {6, 5, 6};
    for (int k = 0; k < 3; k++) {
        __syncthreads();
        for (int i = c.lane; i <= maxSym[k]; i += 64) {
            sh.norm[i] = norms[k][i];
        }
        __syncthreads();
        fse_build(c, sh, sh.fse[k], maxSym[k], logs[k], 0);
        __syncthreads();
        for (int i = c.lane; i < 512; i += 64) {
            dflt[k].e[i] = sh.fse[k].e[i];
        }
    }
}

__global__ __launch_bounds__(64) void zstd_decompress_kernel(BatchArgs a, const zd::FseTable* __restrict__ dflt, uint8_t* litSlabs, int32_t* nextItem, const int32_t* __restrict__ list,
                                                              const int32_t* __restrict__ listCount)
{
    using namespace zd;
    __shared__ Shared sh;
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    const int32_t nItems = list != nullptr ? *listCount : a.nBlocks;  // list mode: the pipeline's fallback list
    for (;;) {
        __syncthreads();
        if (lane == 0) {
            item = atomicAdd(nextItem, 1);
        }
        __syncthreads();
        if (item >= nItems) {
            return;
        }
        const int32_t block = list != nullptr ? list[item] : item;
        Ctx c;
        c.in = a.srcBase + a.srcOff[block];
        c.inLen = a.srcLen[block];
        c.out = a.dstBase + a.dstOff[block];
        c.outCap = a.dstCap[block];
        c.lit = litSlabs + (size_t)blockIdx.x * LIT_SLAB;
        c.lane = lane;
        c.detail = 0;
        c.errOff = 0;
        ZRings R;
        R.init(sh.rings, sh.rings + 2048, c.in, 0, c.out, lane);
        c.R = &R;
        const int32_t r = zstd_decompress_item(c, sh, dflt);
        if (lane == 0) {
            a.outLen[block] = r >= 0 ? r : 0;
            a.status[block] = r >= 0 ? 0 : mk_status(ACHIP_CLASS_MALFORMED, c.detail);
            a.errOffset[block] = r >= 0 ? 0 : (int64_t)c.errOff;
        }
    }
}

namespace {
constexpr int ZD_MAX_WAVES = 256 * 8;  // persistent waves: 8 per CU
}

// scratch of the one-kernel decoder: [item counter | predefined tables | one literal slab per persistent wave]
int64_t zstd_decompress_general_scratch_bytes() { return 4096 + (int64_t)sizeof(zd::FseTable) * 3 + (int64_t)ZD_MAX_WAVES * zd::LIT_SLAB; }

int64_t zstd_decompress_pipe_scratch_bytes(int32_t nBlocks, int32_t tileMax);
hipError_t launch_zstd_decompress_pipe(const BatchArgs& a, hipStream_t stream, void* scratch, void* generalScratch, int32_t tileMax, const ZstdMbProvider* mbp);
void* zstd_decompress_pipe_general_scratch(void* scratch, int32_t nBlocks, int32_t tileMax);

int64_t zstd_decompress_scratch_bytes(int32_t nBlocks, int32_t tileMax) { return zstd_decompress_pipe_scratch_bytes(nBlocks, tileMax); }

// resets the item counter and builds the predefined FSE tables; returns them through *dflt
hipError_t launch_zstd_decompress_prepare(hipStream_t stream, void* generalScratch, const zd::FseTable** dflt)
{
    uint8_t* base = (uint8_t*)generalScratch;
    hipError_t e = hipMemsetAsync(base, 0, 64, stream);
    if (e != hipSuccess) return e;
    zd::FseTable* t = (zd::FseTable*)(base + 1024);
    hipLaunchKernelGGL(zstd_default_tables_kernel, dim3(1), dim3(64), 0, stream, t);
    *dflt = t;
    return hipGetLastError();
}

// the one-kernel decoder over a device-resident list of items (after launch_zstd_decompress_prepare on the same stream)
hipError_t launch_zstd_decompress_list(const BatchArgs& a, hipStream_t stream, void* generalScratch, const int32_t* list, const int32_t* listCount)
{
    uint8_t* base = (uint8_t*)generalScratch;
    const unsigned grid = (unsigned)(list != nullptr || a.nBlocks >= ZD_MAX_WAVES ? ZD_MAX_WAVES : a.nBlocks);
    hipLaunchKernelGGL(zstd_decompress_kernel, dim3(grid), dim3(64), 0, stream, a, (const zd::FseTable*)(base + 1024), base + 4096 + sizeof(zd::FseTable) * 3, (int32_t*)base, list, listCount);
    return hipGetLastError();
}

// variant 1 (default): five-stage pipeline + one-kernel decoder for whatever it hands back; variant 0: one-kernel decoder only
hipError_t launch_zstd_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int variant, int32_t tileMax, const ZstdMbProvider* mbp)
{
    (void)scratchBytes;
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    void* general = zstd_decompress_pipe_general_scratch(scratch, a.nBlocks, tileMax);
    if (variant == 0) {
        const zd::FseTable* dflt = nullptr;
        hipError_t e = launch_zstd_decompress_prepare(stream, general, &dflt);
        if (e != hipSuccess) return e;
        return launch_zstd_decompress_list(a, stream, general, nullptr, nullptr);
    }
    return launch_zstd_decompress_pipe(a, stream, scratch, general, tileMax, mbp);
}

}  // namespace achip
