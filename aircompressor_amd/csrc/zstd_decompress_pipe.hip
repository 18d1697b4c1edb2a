// zstd_decompress_pipe.hip -- five-stage Zstd decode pipeline for gfx950 (the default for batches).
//
// The one-kernel decoder (zstd_decompress.hip) runs one wavefront per item and most of a Zstd block
// is a serial chain (FSE states, Huffman bit positions), so 63 of 64 lanes idle there.  This file
// spreads each serial chain over the LANES instead -- one chain per lane, many items per wavefront --
// and gives every stage the launch shape it wants:
//
//   K1 parse      one wavefront per item   frame / block / literals / sequences headers, Huffman and FSE
//                                          tables built in LDS and written to the item's scratch slot
//   K2 literals   16 items per wavefront   the 4 Huffman streams of an item on 4 lanes, 16 x 4 KiB tables in LDS
//   K3 sequences  16 items per wavefront   the three-state FSE stream of an item on one lane, 16 x 5 KiB tables
//                                          in LDS; (litLen, matchLen, offset) records go to the item's arena range
//   K4 execute    64 items per workgroup   4 lanes per item through the LDS rings of achip_rings.h
//                                          (literal stream in, history window out, whole-line HBM stores)
//   K5 checksum   16 items per wavefront   XXH64: the four accumulators of an item on 4 lanes
//
// Scope: items that hold exactly one frame with exactly one compressed block (what ZstdFrameCompressor and
// libzstd emit for inputs <= 128 KiB -- BASELINE configs[3]) with a Huffman table log <= 11 go through K1..K5 as
// described.  Items that hold exactly one frame of SEVERAL blocks, or of a raw / RLE block (what ZstdOutputStream,
// ZstdFrameCompressor and libzstd emit for longer inputs: SURVEY 8f row 3) are collected by K1 and go through the
// multi-block stages at the end of this file, where the slots of K2 / K3 are BLOCKS and K4 / K5 walk a frame's blocks
// in order.  Everything else, and every item in which ANY stage sees ANYTHING irregular, is appended to a fallback
// list and decoded from scratch by the one-kernel decoder afterwards, which reports the Java-exact status / offset.
// The stages therefore only have to DETECT every condition the Java decoder rejects (in any order), never to
// classify it.
//
// Reference (same functions as zstd_decompress.hip): M/zstd/ZstdFrameDecompressor.java:135-607,708-962,
// M/zstd/Huffman.java:52-324, M/zstd/FseTableReader.java:27-168, M/zstd/BitInputStream.java:28-206,
// M/zstd/XxHash64.java:182-291.
#include "zstd_dec_common.h"
#include "achip_seqexec2.h"
#include "zstd_codes.h"
#include "achip_xxhash.h"
#include <cstring>
#include <vector>

namespace achip {

namespace zp {
using namespace zd;

constexpr int FAST_HUF_LOG = 11;
constexpr int HUF_SLOT = 1 << FAST_HUF_LOG;  // u16 entries per item
constexpr int FSE_LL = 0, FSE_OF = 512, FSE_ML = 768;
constexpr int FSE_SLOT = 1280;               // u16 entries per item: LL 512, OF 256, ML 512 states (symbol | nextState << 6)
constexpr int LIT_STRIDE = MAX_BLOCK_SIZE + 64;
constexpr int ITEMS_PER_WAVE = 16;

struct Desc {
    int32_t state;     // 1 = on the fast path, 0 = handed to the fallback list
    int32_t litMode;   // 0 raw (in the source), 1 RLE, 2 Huffman
    int32_t litSize;
    int32_t litSrc;    // raw: offset in the source; RLE: the byte
    int32_t hufLog;
    int32_t nStreams;
    int32_t sStart[4];
    int32_t sEnd[4];
    int32_t nbSeq;
    int32_t seqStart, seqEnd;  // the sequences bit stream
    int32_t log[3];
    uint32_t seqBase;  // first record in the sequence arena
    int32_t nDecoded;  // K3: records written
    int32_t hasChecksum;
    uint32_t checksum;
    int32_t outSize;   // K4
    uint32_t litBase;  // regenerated literals (RLE / Huffman): first 64-byte unit in the literal arena
    int32_t pad[6];
};
static_assert(sizeof(Desc) == 128, "Desc is 128 bytes");

// ---- multi-block frames: one item = one frame; a SLOT of the stages is one of its blocks ----
struct MbItem {            // per item routed to the multi-block stages (index j in K1's list)
    int32_t item;          // index in the caller's batch
    int32_t firstBlock;    // its first block, counted over all listed items (a pass takes a range of items, its slots count from the range's first block)
    int32_t nBlocks;
    int32_t state;         // 1 = on the fast path, 0 = handed to the fallback list
    int32_t hasChecksum;
    uint32_t checksum;
    int32_t outSize;       // execute stage: bytes produced
    uint32_t litUnits;     // what its blocks will take from the literal arena (64-byte units) and from the sequence arena (records), as
    uint32_t seqs;         // their headers announce it: the host cuts the list into passes that fit the arenas
    int32_t pad[3];
};
static_assert(sizeof(MbItem) == 48, "MbItem is 48 bytes");

struct MbBlock {           // per block slot, written by the walk
    int32_t itemSlot;      // j
    int32_t srcPos;        // the block's content (behind its 3-byte header), relative to the item's source
    int32_t size;          // Block_Size
    int32_t kind;          // 0 raw, 1 RLE, 2 compressed; < 0: the slot is not in use
    int32_t hufSlot;       // the slot whose Huffman table the literals use (its own when it defines one; -1: none defined so far)
    int32_t fseSlot[3];    // the same for the literal-length, offset and match-length tables
    int32_t repOut[3];     // K3: the repeat-offset history behind the block (real offsets, or sentinels for "what it was before the block")
    int32_t pad;
};
static_assert(sizeof(MbBlock) == 48, "MbBlock is 48 bytes");

struct Pipe {
    Desc* desc;
    uint16_t* huf;
    uint16_t* fse;
    uint8_t* lit;
    uint64_t* seq;
    uint32_t seqCap;
    uint32_t* seqCursor;
    uint32_t litCap;      // literal arena, in 64-byte units
    uint32_t* litCursor;
    int32_t* fallbackCount;
    int32_t* fallback;
    int32_t* longCount;   // (single-block tiles) items of this tile whose sequences are long (long_sequences): counted by K3, read by both K4 kernels
    int32_t first;  // first item of this tile
    int32_t count;  // items in this tile (multi-block stages: block slots of this pass)
    // multi-block stages (K1 only appends to mbList; nullptr there: multi-block frames go to the fallback list)
    int32_t* mbList;
    int32_t* mbCount;     // (= counters + 40; + 41: blocks of all listed items)
    MbItem* mbItem;
    MbBlock* mb;
    int32_t passFirst;    // this pass: the first block of its first item (slot 0) ...
    int32_t itemFirst, itemEnd;  // ... and its items [itemFirst, itemEnd) of the list
    int32_t mbSlots;      // what one pass can hold: block slots, literal arena units, sequence records (an item beyond that goes to the fallback list)
    uint32_t mbLitCap, mbSeqCap;
    // multi-block stages: the order the sequence stage takes a pass's block slots in -- by sequence count, longest first (`count` entries), and the
    // sort's 2 x ORDER_BUCKETS counters (bucket sizes, bucket fill); nullptr: slot order
    int32_t* order;
    int32_t* orderHist;
};
constexpr int ORDER_BUCKETS = 256;  // by sequence count / 256, descending (a block holds at most 128 KiB / 3 = 43 691 sequences: bucket 170)

// K4's choice per item: at least 80 output bytes per sequence (capacity as the stand-in for the output size) ...
__device__ __forceinline__ bool long_sequences(int32_t capacity, int32_t nSeq) { return (int64_t)capacity >= 80LL * (nSeq > 0 ? nSeq : 1); }
// ... and only in a tile with enough such items to fill the chip with ring groups (round 5).  The rings give an item four lanes: a handful of long-sequence items
// in a tile of text -- the corpus has them: spreadsheets, images, tables of numbers -- made the ring kernel a serial chain of ~6 ms behind the record executor, for
// a twentieth of the data (65 536 corpus-text frames 70.4 -> 63.2 ms with every item on the record executor; fragments frames, all long: the rings win from
// ~25 000 items on -- 16 384: 690 GiB/s on the record executor against 614, 32 768: 736 against 815, 65 536: 762 against 985; tools/zstd_batch_sizes.py)
constexpr int32_t RINGS_MIN_LONG_ITEMS = 24576;
__device__ __forceinline__ bool item_takes_rings(const Pipe& p, int32_t capacity, int32_t nSeq) { return long_sequences(capacity, nSeq) && *p.longCount >= RINGS_MIN_LONG_ITEMS; }

__device__ __forceinline__ void to_fallback(const Pipe& p, int32_t slot, int stage)
{
    p.desc[slot].state = 0;
    atomicAdd(p.fallbackCount + 32 + stage, 1);  // per-stage diagnostics (achip_ctx_get_stat)
    const int32_t k = atomicAdd(p.fallbackCount, 1);
    p.fallback[k] = p.first + slot;
}
// multi-block stages: the ITEM leaves the fast path (once: its blocks fail independently of each other)
__device__ __forceinline__ void mb_to_fallback(const Pipe& p, int32_t j, int stage)
{
    if (atomicExch(&p.mbItem[j].state, 0) == 1) {
        atomicAdd(p.fallbackCount + 32 + stage, 1);
        const int32_t k = atomicAdd(p.fallbackCount, 1);
        p.fallback[k] = p.mbItem[j].item;
    }
}
template <bool MB>
__device__ __forceinline__ void slot_to_fallback(const Pipe& p, int32_t slot, int stage)
{
    if (MB) {
        p.desc[slot].state = 0;
        mb_to_fallback(p, p.mb[slot].itemSlot, stage);
    }
    else {
        to_fallback(p, slot, stage);
    }
}
// the caller's item behind a slot
template <bool MB>
__device__ __forceinline__ int32_t slot_item(const Pipe& p, int32_t slot)
{
    return MB ? p.mbItem[p.mb[slot].itemSlot].item : p.first + slot;
}

// ---- K1 ----
__device__ bool parse_block(Ctx& c, TableShared& sh, const Pipe& p, int32_t slot, const FseTable* dflt, Desc& d, int32_t input, int32_t blockSize, bool hufBefore, int32_t fseBefore);

// returns 1 when the item is on the fast path and d is complete, 2 when it is a candidate for the multi-block stages, 0 otherwise
// (wave-uniform)
__device__ int32_t parse_item(Ctx& c, TableShared& sh, const Pipe& p, int32_t slot, const FseTable* dflt, Desc& d)
{
    if (c.outCap <= 0) {
        return 0;
    }
    const int32_t inputLimit = c.inLen;
    if (inputLimit < 4 + 1 + 3 + 3) {
        return 0;
    }
    int32_t input = 0;
    if ((uint32_t)rd_le(c, input, 4) != 0xFD2FB528u) {
        return 0;
    }
    input += 4;
    const int32_t headerAddress = input;
    const int32_t fhd = (int32_t)rd_le(c, input++, 1);
    const bool singleSegment = (fhd & 0x20) != 0;
    const int32_t dictDesc = fhd & 3;
    const int32_t csDesc = fhd >> 6;
    const int32_t headerSize = 1 + (singleSegment ? 0 : 1) + (dictDesc == 0 ? 0 : (1 << (dictDesc - 1))) + (csDesc == 0 ? (singleSegment ? 1 : 0) : (1 << csDesc));
    if (dictDesc != 0 || headerSize > inputLimit - headerAddress) {
        return 0;
    }
    if (!singleSegment) {
        const int32_t wd = (int32_t)rd_le(c, input++, 1);
        const uint32_t base = 1u << ((10 + (wd >> 3)) & 31);
        const int32_t windowSize = (int32_t)(base + (uint32_t)(((int32_t)base / 8) * (wd & 7)));
        if (windowSize > MAX_WINDOW_SIZE || windowSize < 0) {
            return 0;
        }
    }
    input = headerAddress + headerSize;
    d.hasChecksum = (fhd & 4) != 0 ? 1 : 0;
    if (input + 3 > inputLimit) {
        return 0;
    }
    const int32_t header = (int32_t)rd_le(c, input, 3);
    input += 3;
    const int32_t blockSize = (header >> 3) & 0x1FFFFF;
    if ((header & 1) == 0 || ((header >> 1) & 3) != 2) {
        return p.mbList != nullptr ? 2 : 0;  // more than one block, or a raw / RLE block: the multi-block stages' walk looks at it
    }
    if (blockSize > MAX_BLOCK_SIZE || blockSize < 3) {
        return 0;
    }
    if ((int64_t)input + blockSize + (d.hasChecksum ? 4 : 0) != inputLimit) {
        return 0;  // truncated, or another frame follows
    }
    d.checksum = d.hasChecksum ? (uint32_t)rd_le(c, input + blockSize, 4) : 0u;
    return parse_block(c, sh, p, slot, dflt, d, input, blockSize, false, 0) ? 1 : 0;
}

// One compressed block: `input` is its first byte, d receives everything K2 / K3 / K4 need.  hufBefore: an earlier block of the frame
// left a Huffman table behind (treeless literals may use it: the caller's link says whose); fseBefore bit k: the same for the
// literal-length / offset / match-length table (repeat mode).  Returns false for everything the stages do not take.
__device__ bool parse_block(Ctx& c, TableShared& sh, const Pipe& p, int32_t slot, const FseTable* dflt, Desc& d, int32_t input, int32_t blockSize, bool hufBefore, int32_t fseBefore)
{
    const int32_t blockLimit = input + blockSize;

    // literals section header (decode*Literals, ZstdFrameDecompressor.java:708-858)
    const int32_t b0 = (int32_t)rd_le(c, input, 1);
    const int32_t literalsBlockType = b0 & 3;
    const int32_t type = (b0 >> 2) & 3;
    d.hufLog = 0;
    d.nStreams = 0;
    if (literalsBlockType == 0 || literalsBlockType == 1) {
        int32_t size;
        if (type == 0 || type == 2) {
            size = b0 >> 3;
            input += 1;
        }
        else if (type == 1) {
            size = (int32_t)rd_le(c, input, 2) >> 4;
            input += 2;
        }
        else {
            if (literalsBlockType == 1 && blockSize < 4) {
                return false;
            }
            size = (int32_t)(rd_le(c, input, 3) & 0xFFFFFF) >> 4;
            input += 3;
        }
        if (size > MAX_BLOCK_SIZE) {
            return false;
        }
        d.litSize = size;
        if (literalsBlockType == 0) {
            if (input + size > blockLimit) {
                return false;
            }
            d.litMode = 0;
            d.litSrc = input;
            input += size;
        }
        else {
            if (input + 1 > blockLimit) {
                return false;
            }
            d.litMode = 1;
            d.litSrc = (int32_t)rd_le(c, input++, 1);
        }
    }
    else {
        if ((literalsBlockType == 3 && !hufBefore) || blockSize < 5) {
            return false;  // (nothing to reuse: "Dictionary is corrupted" :292)
        }
        int32_t compressedSize, uncompressedSize, headerBytes;
        bool singleStream = false;
        if (type == 0 || type == 1) {
            singleStream = type == 0;
            const uint32_t h = (uint32_t)rd_le(c, input, 4);
            headerBytes = 3;
            uncompressedSize = (int32_t)((h >> 4) & 0x3FF);
            compressedSize = (int32_t)((h >> 14) & 0x3FF);
        }
        else if (type == 2) {
            const uint32_t h = (uint32_t)rd_le(c, input, 4);
            headerBytes = 4;
            uncompressedSize = (int32_t)((h >> 4) & 0x3FFF);
            compressedSize = (int32_t)((h >> 18) & 0x3FFF);
        }
        else {
            const uint64_t h = rd_le(c, input, 5);
            headerBytes = 5;
            uncompressedSize = (int32_t)((h >> 4) & 0x3FFFF);
            compressedSize = (int32_t)((h >> 22) & 0x3FFFF);
        }
        if (uncompressedSize > MAX_BLOCK_SIZE || headerBytes + compressedSize > blockSize) {
            return false;
        }
        input += headerBytes;
        const int32_t streamsLimit = input + compressedSize;
        int32_t tl = 0;
        if (literalsBlockType == 2) {
            const int32_t n = huf_read_table(c, sh, input, compressedSize, &tl);
            if (n < 0 || tl > FAST_HUF_LOG) {
                return false;
            }
            input += n;
        }
        d.litMode = 2;
        d.litSize = uncompressedSize;
        d.litSrc = 0;
        d.hufLog = tl;
        if (singleStream) {
            d.nStreams = 1;
            d.sStart[0] = input;
            d.sEnd[0] = streamsLimit;
            d.sStart[1] = d.sStart[2] = d.sStart[3] = 0;
            d.sEnd[1] = d.sEnd[2] = d.sEnd[3] = 0;
        }
        else {
            if (streamsLimit - input < 10) {
                return false;
            }
            const int32_t start1 = input + 6;
            const int32_t start2 = start1 + (int32_t)rd_le(c, input, 2);
            const int32_t start3 = start2 + (int32_t)rd_le(c, input + 2, 2);
            const int32_t start4 = start3 + (int32_t)rd_le(c, input + 4, 2);
            if (!(start2 < start3 && start3 < start4 && start4 < streamsLimit)) {
                return false;
            }
            d.nStreams = 4;
            d.sStart[0] = start1; d.sEnd[0] = start2;
            d.sStart[1] = start2; d.sEnd[1] = start3;
            d.sStart[2] = start3; d.sEnd[2] = start4;
            d.sStart[3] = start4; d.sEnd[3] = streamsLimit;
            const int32_t seg = (uncompressedSize + 3) / 4;
            if (3 * seg > uncompressedSize) {
                return false;  // the fourth segment would be negative
            }
        }
        // publish the table (1 << tl entries; the rest of the slot is never indexed); treeless literals: the table, and its log, are those
        // of the slot the block's link names
        if (literalsBlockType == 2) {
            uint16_t* g = p.huf + (size_t)slot * HUF_SLOT;
            for (int32_t i = c.lane * 8; i < (1 << tl); i += 64 * 8) {
                *(u32x4*)(g + i) = *(const u32x4*)(sh.huf + i);
            }
        }
        input = streamsLimit;
    }

    d.litBase = 0;
    if (d.litMode != 0) {
        // room for the regenerated literals (+ 64 bytes of slack for whole-vector stores)
        const uint32_t units = (uint32_t)(d.litSize + 64 + 63) >> 6;
        uint32_t lb = 0;
        if (c.lane == 0) {
            lb = atomicAdd(p.litCursor, units);
        }
        lb = __shfl(lb, 0);
        if ((uint64_t)lb + units > p.litCap) {
            return false;  // arena full: the one-kernel decoder takes it
        }
        d.litBase = lb;
    }

    // sequences section header (decompressSequences :312-376)
    if (blockLimit - input < 1) {
        return false;
    }
    int32_t sequenceCount = (int32_t)rd_le(c, input++, 1);
    d.log[0] = d.log[1] = d.log[2] = 0;
    d.seqBase = 0;
    if (sequenceCount != 0) {
        if (sequenceCount == 255) {
            if (input + 2 > blockLimit) {
                return false;
            }
            sequenceCount = (int32_t)rd_le(c, input, 2) + 0x7F00;
            input += 2;
        }
        else if (sequenceCount > 127) {
            if (input >= blockLimit) {
                return false;
            }
            sequenceCount = ((sequenceCount - 128) << 8) + (int32_t)rd_le(c, input++, 1);
        }
        if (input + 4 > blockLimit) {
            return false;
        }
        const int32_t modes = (int32_t)rd_le(c, input++, 1);
        const int32_t maxSym[3] = {35, 28, 52};
        const int32_t maxLog[3] = {9, 8, 9};
        const int32_t dfltLog[3] = {6, 5, 6};
        const int32_t base[3] = {FSE_LL, FSE_OF, FSE_ML};
        // Published form of a decoding table (half the LDS of the {newState, symbol, bits} form, so K3 keeps 64 items per
        // CU): per state  symbol | nextState << 6  (FseTableReader.java:143-158 walks the states of a symbol in order, handing out nextState
        // = count, count + 1, ...: below 2 << log = 1024, ten bits).  K3 recovers  bits = log - highBit(nextState),  newState =
        // (nextState << bits) - (1 << log).  (Until round 4: symbol | rank << 6 plus a count per symbol, added at every step.)
        uint16_t* g = p.fse + (size_t)slot * FSE_SLOT;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int32_t mode = (modes >> (6 - 2 * k)) & 3;
            int32_t tableLog = 0;
            if (mode == 3) {
                if (((fseBefore >> k) & 1) == 0) {
                    return false;  // nothing to repeat
                }
                // (the table and its log are those of the slot the block's link names)
            }
            else if (mode == 1) {
                if (input >= blockLimit) {
                    return false;
                }
                const int32_t value = (int8_t)rd_le(c, input++, 1);
                if (value > maxSym[k] || value < 0) {
                    return false;
                }
                if (c.lane == 0) {
                    g[base[k]] = (uint16_t)(value | (1 << 6));  // one state: nextState 1
                }
            }
            else if (mode == 0) {
                const int32_t log = dfltLog[k];
                for (int32_t i = c.lane; i < (1 << log); i += 64) {
                    const uint32_t e = dflt[k].e[i];
                    const int32_t sym = FSE_SYMBOL(e);
                    const int32_t next = (FSE_NEWSTATE(e) + (1 << log)) >> FSE_NBITS(e);
                    g[base[k] + i] = (uint16_t)(sym | ((next & 1023) << 6));
                }
                tableLog = log;
            }
            else {
                int32_t log = 0;
                const int32_t n = read_fse_table(c, sh, sh.fse[k], input, blockLimit, maxSym[k], maxLog[k], &log);
                if (n < 0) {
                    return false;
                }
                input += n;
                for (int32_t i = c.lane; i < (1 << log); i += 64) {
                    const uint32_t e = sh.fse[k].e[i];
                    const int32_t sym = FSE_SYMBOL(e);
                    const int32_t next = (FSE_NEWSTATE(e) + (1 << log)) >> FSE_NBITS(e);
                    g[base[k] + i] = (uint16_t)(sym | ((next & 1023) << 6));
                }
                __syncthreads();  // sh.norm is rewritten by the next table
                tableLog = log;
            }
            if (k == 0) {
                d.log[0] = tableLog;
            }
            else if (k == 1) {
                d.log[1] = tableLog;
            }
            else {
                d.log[2] = tableLog;
            }
        }
        uint32_t b = 0;
        if (c.lane == 0) {
            b = atomicAdd(p.seqCursor, (uint32_t)sequenceCount);
        }
        b = __shfl(b, 0);
        if ((uint64_t)b + (uint32_t)sequenceCount > p.seqCap) {
            return false;  // arena full: the one-kernel decoder takes it
        }
        d.seqBase = b;
    }
    d.nbSeq = sequenceCount;
    d.nDecoded = 0;
    d.seqStart = input;
    d.seqEnd = blockLimit;
    d.outSize = 0;
    return true;
}

}  // namespace zp

__global__ __launch_bounds__(64) void zstd_pipe_parse_kernel(BatchArgs a, zp::Pipe p, const zd::FseTable* __restrict__ dflt)
{
    using namespace zp;
    __shared__ TableShared sh;
    const int32_t slot = blockIdx.x;
    const int32_t block = p.first + slot;
    Ctx c;
    c.in = a.srcBase + a.srcOff[block];
    c.inLen = a.srcLen[block];
    c.out = a.dstBase + a.dstOff[block];
    c.outCap = a.dstCap[block];
    c.lit = nullptr;
    c.R = nullptr;
    c.lane = threadIdx.x;
    c.detail = 0;
    c.errOff = 0;
    Desc d;
    d.state = 0;
    d.litMode = 0;
    d.litSize = 0;
    d.litSrc = 0;
    d.hufLog = 0;
    d.nStreams = 0;
    for (int i = 0; i < 4; i++) {
        d.sStart[i] = 0;
        d.sEnd[i] = 0;
    }
    d.nbSeq = 0;
    d.seqStart = d.seqEnd = 0;
    d.log[0] = d.log[1] = d.log[2] = 0;
    d.seqBase = 0;
    d.nDecoded = 0;
    d.hasChecksum = 0;
    d.checksum = 0;
    d.outSize = 0;
    d.litBase = 0;
    for (int i = 0; i < 6; i++) {
        d.pad[i] = 0;
    }
    const int32_t r = parse_item(c, sh, p, slot, dflt, d);
    d.state = r == 1 ? 1 : 0;
    if (c.lane == 0) {
        p.desc[slot] = d;
        if (r == 2) {
            p.mbList[atomicAdd(p.mbCount, 1)] = block;
        }
        else if (r == 0) {
            const int32_t k = atomicAdd(p.fallbackCount, 1);
            p.fallback[k] = block;
            atomicAdd(p.fallbackCount + 32 + 1, 1);
        }
    }
}

// ---- shared by K2 and K3: the backward bit reader, and K3's quad broadcasts ----
// One item per QUAD of lanes: lane 0 owns the literal-length state, lane 1 the match-length state, lane 2 the offset
// state, lane 3 assembles and stores the record.  The three table lookups, the code -> (baseline, extra bits)
// conversions and the six bit-field extractions of a sequence run side by side; the fields' bit positions are a
// prefix over the quad (DPP quad_perm broadcasts, no LDS traffic).  The bit container and the repeat-offset history
// are replicated in the four lanes.  All 64 lanes of the wavefront work: 16 items per wavefront.
struct QuadBits {
    int32_t start, current, consumed, b;
    uint64_t bits, A, B, P;
    const uint8_t* in;

    __device__ __forceinline__ uint64_t word_at(int32_t pos) const { return ld8(in + (pos > 0 ? pos : 0)); }  // a clamped address is never consumed
    // Initializer.initialize :110-130 (end >= 8: a frame header and a block header precede every stream)
    __device__ __forceinline__ bool init(const uint8_t* src, int32_t s, int32_t end)
    {
        in = src;
        const int32_t size = end - s;
        if (size < 1 || end < 8) {
            return false;
        }
        const int32_t last = in[end - 1];
        if (last == 0) {
            return false;
        }
        start = s;
        consumed = 8 - zd::highest_bit((uint32_t)last);
        A = ld8(in + end - 8);
        B = 0;
        if (size >= 8) {
            current = end - 8;
        }
        else {
            current = s;  // the whole stream is in the container; load() never moves
            A >>= 8 * (8 - size);
            consumed += (8 - size) * 8;
        }
        b = current;
        P = word_at(b - 8);
        bits = A;
        return true;
    }
    // Loader.load :171-204 without branches.  The three cases of the Java method move `current` down by the whole
    // bytes consumed but never below `start`, and take 8 bits off `consumed` per byte moved; with consumed > 64 the
    // Java method only raises its overflow flag, which is what this returns.
    __device__ __forceinline__ bool load()
    {
        const bool over = consumed > 64;
        int32_t nc = current - (int32_t)((uint32_t)consumed >> 3);
        nc = nc > start ? nc : start;
        nc = over ? current : nc;
        consumed -= 8 * (current - nc);
        current = nc;
        // slide the 16-byte window (at most one word per load: current moves by <= 8) and request the next lower word
        const bool need = current < b;
        B = need ? A : B;
        A = need ? P : A;
        b = need ? b - 8 : b;
        P = word_at(b - 8);
        const int32_t sh = 8 * (current - b);  // 0..64
        const uint64_t mid = (A >> (sh & 63)) | (B << ((64 - sh) & 63));
        bits = sh == 0 ? A : (sh == 64 ? B : mid);
        return over;
    }
    // Loader.load with its return value (true: nothing more can be loaded -- overflow, already at the start, or the move
    // was cut short at the start of the stream)
    __device__ __forceinline__ bool load_java()
    {
        const bool stop = consumed > 64 || current == start || current - (int32_t)((uint32_t)consumed >> 3) < start;
        load();
        return stop;
    }
    // BitInputStream.peekBits :64-67 for n <= 31, on 32-bit lanes after the one 64-bit alignment shift
    __device__ __forceinline__ int32_t peek(int32_t at, int32_t n) const
    {
        const uint32_t hi = (uint32_t)((bits << (at & 63)) >> 32);
        return (int32_t)((hi >> 1) >> ((31 - n) & 31));
    }
};

// One Huffman stream (Huffman.decodeSingleStream :130-164 body + decodeTail :291-317) decoded by the calling lane through
// the windowed reader: same loads, symbols and end-of-stream test as zd::huf_decode_stream.
// SPLIT (round 6): the table as the literal stage keeps it in LDS since then -- HUF_SLOT symbol bytes, then HUF_SLOT / 2 bytes of code lengths, a nibble each (3 KiB an
// item instead of 4: more items, i.e. more streams, per CU) -- two independent LDS reads and a nibble select per symbol; !SPLIT: K1's u16 entries, `symbol | length << 8`.
// SPLIT == 2: HUF_SLOT symbol bytes, then the code length of each of the 256 symbols (2 304 bytes an item): the length is a second, DEPENDENT LDS read.
template <int SPLIT>
__device__ __forceinline__ int32_t huf_symbol_at(const void* huf, int32_t tableLog, uint64_t bits, int32_t& consumed)
{
    if (SPLIT == 0) {
        return zd::huf_symbol((const uint16_t*)huf, tableLog, bits, consumed);
    }
    const uint8_t* t = (const uint8_t*)huf;
    const int32_t idx = (int32_t)zd::peek_bits_fast(consumed, bits, tableLog);
    const uint32_t sym = t[idx];
    if (SPLIT == 2) {
        consumed += (int32_t)t[zp::HUF_SLOT + sym];
        return (int32_t)sym;
    }
    const uint32_t pair = t[zp::HUF_SLOT + (idx >> 1)];
    consumed += (int32_t)((pair >> ((idx & 1) * 4)) & 15u);
    return (int32_t)sym;
}
template <int SPLIT = 0>
__device__ __forceinline__ bool huf_decode_stream_win(QuadBits& b, const void* huf, int32_t tableLog, uint8_t* out, int32_t output, int32_t outputLimit)
{
    const int32_t fastLimit = outputLimit - 4;
    bool done = false;
    // four trips of the Java loop per 16-byte store (round 4): the stage's lanes each walk their own stream, so every load and store of a
    // wavefront is 64 separate lines for the memory pipeline -- a quarter of the store instructions is a quarter of that work
    while (!done && output + 12 < fastLimit) {
        uint32_t w4[4];
        int n4 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (!done) {
                if (b.load_java()) {
                    done = true;
                }
                else {
                    uint32_t w = (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed);
                    w |= (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed) << 8;
                    w |= (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed) << 16;
                    w |= (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed) << 24;
                    w4[t] = w;
                    n4 = t + 1;
                }
            }
        }
        if (n4 == 4) {
            st16(out + output, u32x4{w4[0], w4[1], w4[2], w4[3]});
        }
        else {
            for (int t = 0; t < n4; t++) {
                st4(out + output + 4 * t, w4[t]);
            }
        }
        output += 4 * n4;
    }
    while (!done && output < fastLimit) {
        if (b.load_java()) {
            done = true;
            break;
        }
        uint32_t w = (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed);
        w |= (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed) << 8;
        w |= (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed) << 16;
        w |= (uint32_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed) << 24;
        st4(out + output, w);
        output += 4;
    }
    if (!done) {
        while (output < outputLimit) {
            if (b.load_java()) {
                break;
            }
            out[output++] = (uint8_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed);
        }
    }
    while (output < outputLimit) {
        out[output++] = (uint8_t)huf_symbol_at<SPLIT>(huf, tableLog, b.bits, b.consumed);
    }
    return b.start == b.current && b.consumed == 64;
}

// ---- K2: literals (MB: the slots are the blocks of multi-block frames; a block with treeless literals uses the table of the slot its
// link names) ----
// ITEMS: items per wavefront (16: every quad has one; 8 -- an experiment for the next round -- leaves half the lanes idle and halves the LDS,
// so that more wavefronts share a CU: the stage waits for memory more than it computes)
template <bool MB, int ITEMS = zp::ITEMS_PER_WAVE, int SPLIT = 0>
__global__ __launch_bounds__(64) void zstd_pipe_literals_kernel(BatchArgs a, zp::Pipe p)
{
    using namespace zp;
    constexpr int SLOT_U16 = SPLIT == 1 ? HUF_SLOT * 3 / 4 : (SPLIT == 2 ? HUF_SLOT / 2 + 128 : HUF_SLOT);  // an item's table in LDS, in u16 units (1: 2 048 symbol bytes + 1 024 bytes of length nibbles; 2: + 256 lengths by symbol)
    __shared__ __attribute__((aligned(16))) uint16_t tables[ITEMS * SLOT_U16];  // 64 KiB at 16 items
    const int lane = threadIdx.x;
    const int q = lane >> 2;  // item of this lane
    const int s = lane & 3;   // stream of this lane
    const int32_t slot = blockIdx.x * ITEMS + q;
    bool valid = q < ITEMS && slot < p.count;
    int32_t tableSlot = slot;
    if (MB && valid) {
        const MbBlock b = p.mb[slot];
        valid = b.kind == 2 && p.mbItem[b.itemSlot].state == 1;
        tableSlot = b.hufSlot;
    }
    Desc d;
    d.state = 0;
    if (valid) {
        d = p.desc[slot];
    }
    const bool live = valid && d.state == 1;
    if (MB && live && d.litMode == 2) {
        d.hufLog = tableSlot >= 0 ? p.desc[tableSlot].hufLog : 0;  // (its own, or that of the block that defined the table)
    }
    // stage the tables (each copy is done by the whole wavefront)
    for (int k = 0; k < ITEMS; k++) {
        const int32_t useHuf = __shfl((live && d.litMode == 2) ? d.hufLog : 0, k * 4);
        const int32_t from = __shfl(tableSlot, k * 4);
        if (useHuf > 0) {
            const uint16_t* g = p.huf + (size_t)from * HUF_SLOT;
            for (int32_t i = lane * 8; i < (1 << useHuf); i += 64 * 8) {
                const u32x4 v = *(const u32x4*)(g + i);
                if (SPLIT == 0) {
                    *(u32x4*)(tables + k * SLOT_U16 + i) = v;
                }
                else {  // eight entries `symbol | length << 8`: eight symbol bytes, eight length nibbles
                    uint8_t* t = (uint8_t*)(tables + k * SLOT_U16);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                    uint32_t symLo = 0, symHi = 0, lens = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t two = (w[j] & 0xFFu) | ((w[j] >> 8) & 0xFF00u);  // the two symbols of the pair
                        const uint32_t nib = ((w[j] >> 8) & 0xFu) | ((w[j] >> 20) & 0xF0u);  // their two lengths
                        if (j < 2) symLo |= two << (16 * j);
                        else symHi |= two << (16 * (j - 2));
                        lens |= nib << (8 * j);
                    }
                    *(uint32_t*)(t + i) = symLo;
                    *(uint32_t*)(t + i + 4) = symHi;
                    if (SPLIT == 1) {
                        *(uint32_t*)(t + HUF_SLOT + i / 2) = lens;
                    }
                    else {  // (every entry of a symbol carries the symbol's length: whichever store lands last, lands the same byte)
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t sy = ((j < 4 ? symLo : symHi) >> (8 * (j & 3))) & 0xFFu;
                            t[HUF_SLOT + sy] = (uint8_t)((lens >> (4 * j)) & 15u);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    int32_t bad = 0;
    if (live && d.litMode != 0) {
        const int32_t block = slot_item<MB>(p, slot);
        uint8_t* lit = p.lit + (size_t)d.litBase * 64;
        if (d.litMode == 1) {
            const uint32_t v = (uint32_t)(d.litSrc & 0xFF) * 0x01010101u;
            const u32x4 vv = {v, v, v, v};
            for (int32_t i = s * 16; i < d.litSize; i += 64) {
                *(u32x4*)(lit + i) = vv;  // the slab has 64 bytes of slack
            }
        }
        else if (s < d.nStreams) {
            Ctx c;
            c.in = a.srcBase + a.srcOff[block];
            c.inLen = a.srcLen[block];
            c.out = nullptr;
            c.outCap = 0;
            c.lit = lit;
            c.R = nullptr;
            c.lane = lane;
            c.detail = 0;
            c.errOff = 0;
            const int32_t seg = (d.litSize + 3) / 4;
            const int32_t oStart = d.nStreams == 1 ? 0 : s * seg;
            const int32_t oEnd = d.nStreams == 1 ? d.litSize : (s == 3 ? d.litSize : (s + 1) * seg);
            const int32_t myStart = s == 0 ? d.sStart[0] : (s == 1 ? d.sStart[1] : (s == 2 ? d.sStart[2] : d.sStart[3]));
            const int32_t myEnd = s == 0 ? d.sEnd[0] : (s == 1 ? d.sEnd[1] : (s == 2 ? d.sEnd[2] : d.sEnd[3]));
            QuadBits b;
            if (!b.init(c.in, myStart, myEnd) || oStart > oEnd) {
                bad = 1;
            }
            else if (!huf_decode_stream_win<SPLIT>(b, tables + q * SLOT_U16, d.hufLog, lit, oStart, oEnd)) {
                bad = 1;
            }
        }
    }
    // any failing stream sends the whole item to the fallback list
    const unsigned long long badMask = __ballot(bad != 0);
    if (live && s == 0 && ((badMask >> (q * 4)) & 0xFull) != 0) {
        slot_to_fallback<MB>(p, slot, 2);
    }
}

// K2's launch.  The stage's wavefronts wait more than they issue (53 % of their cycles on the corpus batch, round 6 counters), so what counts is streams in
// flight per CU, and that is LDS: 4 KiB of table an item.  16 items a wavefront are 64 KiB -- two wavefronts a CU, 128 streams, 32 KiB of the LDS unused;
// `zstd.decompress.lit_items` 10 (40 KiB: four wavefronts, 160 streams) or 8 (32 KiB: five, 160) use all of it, with lanes idle in every wavefront.
// Measured (profiles/r06_notes.md section 10; 32 768 corpus frames): 16 items 6.96 ms, 10 items **4.83**, 8 items 6.96 (the fifth wavefront does not get
// its LDS: four of 32 lanes are the 128 streams of two full ones).  Then the table itself (SPLIT, above): symbol bytes and length nibbles apart are 3 KiB an item --
// 13 items a wavefront, 208 streams a CU: **4.11**; symbols and lengths-by-symbol (2 304 bytes, 16 items, 256 streams, but a dependent second lookup): 4.60.
// 13 is the default.
int g_zstd_lit_items = 13;
template <bool MB>
inline void launch_literals(const BatchArgs& a, const zp::Pipe& p, hipStream_t stream)
{
    if (g_zstd_lit_items == 13) {  // (3 KiB an item: 13 items = 39 KiB, four wavefronts a CU, 208 streams)
        hipLaunchKernelGGL((zstd_pipe_literals_kernel<MB, 13, 1>), dim3((unsigned)((p.count + 12) / 13)), dim3(64), 0, stream, a, p);
    }
    else if (g_zstd_lit_items == 20) {  // (2 304 bytes an item, 16 items = 36 KiB, four wavefronts a CU, 256 streams -- and a dependent second lookup per symbol)
        hipLaunchKernelGGL((zstd_pipe_literals_kernel<MB, 16, 2>), dim3((unsigned)((p.count + 15) / 16)), dim3(64), 0, stream, a, p);
    }
    else if (g_zstd_lit_items == 8) {
        hipLaunchKernelGGL((zstd_pipe_literals_kernel<MB, 8>), dim3((unsigned)((p.count + 7) / 8)), dim3(64), 0, stream, a, p);
    }
    else if (g_zstd_lit_items == 10) {
        hipLaunchKernelGGL((zstd_pipe_literals_kernel<MB, 10>), dim3((unsigned)((p.count + 9) / 10)), dim3(64), 0, stream, a, p);
    }
    else {
        hipLaunchKernelGGL((zstd_pipe_literals_kernel<MB, 16>), dim3((unsigned)((p.count + 15) / 16)), dim3(64), 0, stream, a, p);
    }
}

// ---- multi-block stages: the order K3 takes a pass's block slots in.  A pass is whatever number of blocks its frames hold -- 40 896 for the bench's
// 1 024 corpus streams: 2.5 rounds of 64 x 256 items, i.e. three --, its slots include the frames' raw and RLE blocks (nothing for K3 to do), and the
// last round runs as long as its longest item.  Sorted by sequence count, longest first, the slots without work form whole wavefronts at the end
// and the last round is the shortest items: K3 17.9 -> ~11 ms per launch on those streams.  A counting sort in two launches; one atomic per wavefront
// and bucket (the lanes of a wavefront that share a bucket go together), so a pass whose blocks all fall into one bucket costs a wavefront one
// atomic, and a wavefront's slots stay together.  (The single-block pipeline keeps batch order: 65 536 items are exactly four rounds, measured
// 28.8 against 29.5 ms -- profiles/r04_notes.md.)  The order decides when a slot is decoded, never what it decodes to. ----
__device__ __forceinline__ int32_t mb_order_bucket(const zp::Pipe& p, int32_t slot)
{
    int32_t n = 0;
    if (p.mb[slot].kind == 2) {
        const zp::Desc& d = p.desc[slot];
        n = d.state == 1 ? d.nbSeq : 0;
    }
    n = n < 0 ? 0 : n >> 8;
    return zp::ORDER_BUCKETS - 1 - (n < zp::ORDER_BUCKETS - 1 ? n : zp::ORDER_BUCKETS - 1);
}
// PLACE == false: bucket sizes into orderHist[0 .. B); PLACE == true: every slot takes its place (orderHist[B .. 2B) counts what is taken)
template <bool PLACE>
__global__ __launch_bounds__(64) void zstd_mb_order_kernel(zp::Pipe p)
{
    using namespace zp;
    __shared__ int32_t start[ORDER_BUCKETS];
    const int lane = threadIdx.x;
    if (PLACE) {
        int32_t base = 0;
        for (int b0 = 0; b0 < ORDER_BUCKETS; b0 += 64) {  // (uniform) exclusive scan of the bucket sizes
            const int32_t c = p.orderHist[b0 + lane];
            const int32_t incl = sx::wave_scan_incl(c, lane);
            start[b0 + lane] = base + incl - c;
            base += sx::wave_bcast(incl, 63);
        }
        __syncthreads();
    }
    const int32_t slot = blockIdx.x * 64 + lane;
    const bool have = slot < p.count;
    const int32_t b = have ? mb_order_bucket(p, slot) : -1;
    unsigned long long left = __ballot(have);
    while (left != 0) {  // (uniform) bucket by bucket, in the order the buckets turn up among the lanes
        const int leader = __builtin_ctzll(left);
        const int32_t k = sx::wave_bcast(b, leader);
        const unsigned long long same = __ballot(b == k);
        int32_t at = 0;
        if (lane == leader) {
            at = atomicAdd(p.orderHist + (PLACE ? ORDER_BUCKETS : 0) + k, (int32_t)__popcll(same));
        }
        at = sx::wave_bcast(at, leader);
        if (PLACE && b == k) {
            p.order[start[k] + at + (int32_t)__popcll(same & ((1ull << lane) - 1ull))] = slot;
        }
        left &= ~same;
    }
}

// ---- K3: sequences, a lane per item (MB: the slots are the blocks of multi-block frames -- each of a block's three tables may be that of an earlier
// block (repeat mode), and the repeat-offset history at the block's start is what the block before leaves behind, which is not known
// here: the history starts as three SENTINELS (achip_seqexec2.h REP_SENTINEL), records may hold sentinels, and the history behind the
// block goes to MbBlock::repOut for the execute stage, which walks the blocks in order and knows) ----
// Until round 4 the stage gave an item to a QUAD of lanes (one state each, the fields' positions a prefix over the quad): 16 items and ~216
// instructions per step and wavefront.  What bounds the stage
// is the LDS: 2.5 KB of state tables per item, ~55 items per CU whatever the launch shape, each a serial chain -- so the rate is (items per
// CU) / (time of one step), and a step of a wavefront costs its instruction count whether 16 or 60 of its lanes hold an item.  Here a LANE
// owns an item: its three states, its bit container, its repeat-offset history; the three table lookups of a step are independent loads of
// one lane, nothing is broadcast, the record goes straight to the arena (8 bytes per lane and step: the L2 merges a line's pieces).  60 items
// per wavefront at first; 64 since K1 publishes  symbol | nextState << 6  itself (the staging is a plain copy, no counts to hold) and the code
// tables (128 words) moved from LDS to a constant table in memory (hot in the CU's L1): 64 x 2 560 B = the CU's whole LDS, one wavefront per CU.
namespace zp {
constexpr int SEQL_ITEMS = 64;      // every lane holds an item: 64 x 2 560 B of tables = the CU's whole LDS
constexpr int SEQL_STRIDE = 1280;   // u16 per item in LDS
}
// sequence code -> baseline | extra bits << 24: literal-length codes at 0 .. 35, match-length codes at 64 .. 116 (ZstdFrameDecompressor.java:68-83;
// the arithmetic form in zstd_codes.h is what tests/test_host_logic.py holds against the Java tables, and tests/test_host_logic.py holds THIS
// table against that form)
#define ZC(base, bits) ((uint32_t)(base) | ((uint32_t)(bits) << 24))
__device__ const uint32_t seq_code_table[128] = {
    ZC(0, 0), ZC(1, 0), ZC(2, 0), ZC(3, 0), ZC(4, 0), ZC(5, 0), ZC(6, 0), ZC(7, 0), ZC(8, 0), ZC(9, 0), ZC(10, 0), ZC(11, 0), ZC(12, 0), ZC(13, 0), ZC(14, 0), ZC(15, 0),
    ZC(16, 1), ZC(18, 1), ZC(20, 1), ZC(22, 1), ZC(24, 2), ZC(28, 2), ZC(32, 3), ZC(40, 3), ZC(48, 4), ZC(64, 6), ZC(128, 7), ZC(256, 8), ZC(512, 9), ZC(1024, 10),
    ZC(2048, 11), ZC(4096, 12), ZC(8192, 13), ZC(16384, 14), ZC(32768, 15), ZC(65536, 16),
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    ZC(3, 0), ZC(4, 0), ZC(5, 0), ZC(6, 0), ZC(7, 0), ZC(8, 0), ZC(9, 0), ZC(10, 0), ZC(11, 0), ZC(12, 0), ZC(13, 0), ZC(14, 0), ZC(15, 0), ZC(16, 0), ZC(17, 0), ZC(18, 0),
    ZC(19, 0), ZC(20, 0), ZC(21, 0), ZC(22, 0), ZC(23, 0), ZC(24, 0), ZC(25, 0), ZC(26, 0), ZC(27, 0), ZC(28, 0), ZC(29, 0), ZC(30, 0), ZC(31, 0), ZC(32, 0), ZC(33, 0), ZC(34, 0),
    ZC(35, 1), ZC(37, 1), ZC(39, 1), ZC(41, 1), ZC(43, 2), ZC(47, 2), ZC(51, 3), ZC(59, 3), ZC(67, 4), ZC(83, 4), ZC(99, 5), ZC(131, 7), ZC(259, 8), ZC(515, 9),
    ZC(1027, 10), ZC(2051, 11), ZC(4099, 12), ZC(8195, 13), ZC(16387, 14), ZC(32771, 15), ZC(65539, 16),
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
};
#undef ZC
// where the lanes of the sequence stage that hold no item put the record store every step issues (see the kernel): a slot per lane and workgroup
// (workgroups SEQ_IDLE_GROUPS apart share theirs: nothing reads it)
constexpr int SEQ_IDLE_GROUPS = 1024;
__device__ uint64_t seq_idle_sink[SEQ_IDLE_GROUPS * 256];
// Items per workgroup of the sequence stage: the stage is ONE wavefront per CU (its tables fill the LDS), every lane a serial chain -- 4 096 items at 64 a workgroup
// are 64 workgroups on 64 of the 256 CUs, each as long as a full one takes.  So a tile of fewer than 256 x 64 items is spread: count / 256 items a workgroup
// (the static LDS allocation stays: still one workgroup per CU), down to a single item; a step of a wavefront with fewer lanes on it is also the shorter one.
constexpr int32_t ZSTD_EXEC_ALL_RECORDS_MAX_ITEMS = 16384;
int g_zstd_seq_spread = 256;  // (the CUs a small tile is spread over; tools/hostemu sets 1 to get full workgroups from a handful of items)
inline int32_t seql_items_for(int32_t count)
{
    const int32_t per = (count + g_zstd_seq_spread - 1) / g_zstd_seq_spread;
    return per < 1 ? 1 : (per > zp::SEQL_ITEMS ? zp::SEQL_ITEMS : per);
}
int g_zstd_seq_waves = 1;
// Wavefronts per workgroup of the sequence stage (context option zstd.decompress.seq_waves: 1, 2 or 4).  The LDS holds 64 items' tables however they are
// spread; a wavefront's step costs its ~140 vector instructions whether 16 or 64 of its lanes hold an item, but four wavefronts of 16 items issue theirs on
// four SIMDs side by side where one wavefront of 64 leaves three SIMDs idle.
inline int32_t seql_waves_for(int32_t count)
{
    const int32_t items = seql_items_for(count);
    return items < g_zstd_seq_waves ? 1 : g_zstd_seq_waves;
}
inline int32_t seql_items_per_wave(int32_t count)
{
    const int32_t w = seql_waves_for(count);
    return (seql_items_for(count) + w - 1) / w;
}
template <bool MB>
__global__ __launch_bounds__(256) void zstd_pipe_sequences_lane_kernel(BatchArgs a, zp::Pipe p, int32_t itemsPerGroup, int32_t itemsPerWave)
{
    using namespace zp;
    __shared__ __attribute__((aligned(16))) uint16_t tables[SEQL_ITEMS * SEQL_STRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (itemsPerGroup <= SEQL_ITEMS: a tile of few items is spread over the CUs -- seql_items_for below -- and the lanes beyond sit out; the group's items are
    // dealt to its wavefronts itemsPerWave at a time, an item's place in the group is also its table slot in the LDS)
    const int32_t inGroup = wave * itemsPerWave + lane;
    const int32_t place = blockIdx.x * itemsPerGroup + inGroup;
    bool valid = lane < itemsPerWave && inGroup < itemsPerGroup && place < p.count;
    const int32_t slot = MB && valid && p.order != nullptr ? p.order[place] : place;  // (multi-block passes: longest items first, zstd_mb_order_kernel)
    int32_t from0 = slot, from1 = slot, from2 = slot;  // the slots the literal-length, offset and match-length tables come from
    if (MB && valid) {
        const MbBlock b = p.mb[slot];
        valid = b.kind == 2 && p.mbItem[b.itemSlot].state == 1;
        from0 = b.fseSlot[0];
        from1 = b.fseSlot[1];
        from2 = b.fseSlot[2];
    }
    Desc d;
    d.state = 0;
    d.nbSeq = 0;
    if (valid) {
        d = p.desc[slot];
    }
    const bool live = valid && d.state == 1 && d.nbSeq > 0 && (!MB || (from0 >= 0 && from1 >= 0 && from2 >= 0));
    if (MB && live) {
        d.log[0] = p.desc[from0].log[0];
        d.log[1] = p.desc[from1].log[1];
        d.log[2] = p.desc[from2].log[2];
    }
    // stage the tables, item by item (each copy is done by the whole wavefront), eight items' loads in flight at a time: with a load and its store
    // per trip the staging of 64 items was 192 memory round trips one after the other -- up to 0.4 ms per wavefront, and most of the stage's
    // time on streams of small blocks (a hundred sequences per block).  Round 6: the 24 loads of a trip are UNCONDITIONAL (a slot without an item
    // is filled from the first live item's tables, the lanes beyond a table's last piece copy that piece again) -- under their branches the compiler
    // put a `vmcnt(0)` in front of every one of them, and "eight items in flight" was one.
    const unsigned long long liveMask = __ballot(live);
    if (liveMask == 0) {  // (uniform)
        return;
    }
    const int leader = (int)__builtin_ctzll(liveMask);
    const int32_t lf0 = __shfl(from0, leader), lf1 = __shfl(from1, leader), lf2 = __shfl(from2, leader), leaderSlot = __shfl(slot, leader);
    const int32_t sf0 = live ? from0 : lf0, sf1 = live ? from1 : lf1, sf2 = live ? from2 : lf2;
    constexpr int STAGE_ITEMS = 8;
    for (int k0 = 0; k0 < SEQL_ITEMS; k0 += STAGE_ITEMS) {  // (uniform)
        if (((liveMask >> k0) & ((1ull << STAGE_ITEMS) - 1ull)) == 0) {
            continue;
        }
        u32x4 v[STAGE_ITEMS][3];
#pragma unroll
        for (int u = 0; u < STAGE_ITEMS; u++) {
            const int k = k0 + u;
            const int32_t f0 = __shfl(sf0, k), f1 = __shfl(sf1, k), f2 = __shfl(sf2, k);
            const uint16_t* gLL = p.fse + (size_t)f0 * FSE_SLOT;
            const uint16_t* gOF = p.fse + (size_t)f1 * FSE_SLOT;
            const uint16_t* gML = p.fse + (size_t)f2 * FSE_SLOT;
#pragma unroll
            for (int t = 0; t < 3; t++) {  // 1280 states, 8 per piece: LL 0..511, OF 512..767, ML 768..1279
                const int32_t i0 = (lane + 64 * t) * 8;
                const int32_t i = i0 < FSE_SLOT ? i0 : FSE_SLOT - 8;
                const uint16_t* g = i < FSE_OF ? gLL : (i < FSE_ML ? gOF : gML);
                v[u][t] = ld16_global((const uint8_t*)(g + i));
            }
        }
#pragma unroll
        for (int u = 0; u < STAGE_ITEMS; u++) {
            const int k = k0 + u;
            if (k < itemsPerWave) {  // (uniform; the slots beyond belong to the next wavefront)
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int32_t i0 = (lane + 64 * t) * 8;
                    const int32_t i = i0 < FSE_SLOT ? i0 : FSE_SLOT - 8;
                    *(u32x4*)(tables + (wave * itemsPerWave + k) * SEQL_STRIDE + i) = v[u][t];
                }
            }
        }
    }
    wave_sync();  // (a wavefront reads the tables it staged itself)
    // From here on EVERY lane of the wavefront stays to the end, whether it holds an item or not (round 6).  Two things follow from that, and both are
    // what the stage's time is made of: the code tables can be held by the lanes and read with a lane gather (`ds_bpermute` returns 0 for a lane that
    // has left: all have to be there) instead of two loads from memory in the middle of every step's dependency chain, and every step issues the same
    // vector-memory operations -- one load of stream bytes, one record store -- so that the compiler can count them: the wait for the stream bytes is
    // `vmcnt(1)`, with the step's record store still on its way, where a store under a branch made it `vmcnt(0)` and every step waited for the
    // store before it to reach the L2.  A lane without work (no item, an item that is through or has failed) runs the step on harmless values: it reads
    // its own stream's start (or the first live lane's), looks up its own -- possibly unwritten -- table slot with masked indices, and stores the record
    // it stored last again (a lane that never had one: into a dummy slot of its own).
    const int32_t block = slot_item<MB>(p, live ? slot : leaderSlot);
    const uint8_t* src = a.srcBase + a.srcOff[block];
    const int32_t tableSlot = inGroup < SEQL_ITEMS ? inGroup : SEQL_ITEMS - 1;  // (a lane without an item reads somebody's tables: harmless)
    const uint16_t* tLL = tables + tableSlot * SEQL_STRIDE + FSE_LL;
    const uint16_t* tOF = tables + tableSlot * SEQL_STRIDE + FSE_OF;
    const uint16_t* tML = tables + tableSlot * SEQL_STRIDE + FSE_ML;
    const int32_t logLL = live ? d.log[0] : 0, logOF = live ? d.log[1] : 0, logML = live ? d.log[2] : 0;
    uint64_t* rec = live ? p.seq + d.seqBase : seq_idle_sink + ((blockIdx.x & (SEQ_IDLE_GROUPS - 1)) * 256 + threadIdx.x);
    // literal-length code c -> baseline | extra bits << 24 in lane c, match-length code c likewise (36 and 53 codes: a lane each)
    const uint32_t codeLL = seq_code_table[lane], codeML = seq_code_table[64 + lane];

    // The bit stream, MSB first: `w` holds the stream from the current bit position down (its top `have` bits came from whole bytes above
    // `ptr`; what lies below them in `w` is either zero or the stream's next bits, never anything else), `nextWord` the eight bytes below
    // `ptr`, requested one refill ahead.  A field of n bits at distance `at` from the current position is the top of w << at; consuming is a
    // shift.  This reads exactly the bits BitInputStream.peekBits reads as long as a step stays inside the Java container -- which it does
    // for every stream that is not sent to the fallback list below: offset codes above 24 are (7 + 24 + 16 + 16 bits fit a container).
    // `rem` counts the stream's bits not yet consumed: Loader.load() reports overflow exactly when it is negative at a step's start
    // (bitsConsumed > 64 is only possible with the container at the stream's first byte).  A step whose EXTRA bits run past the stream's start
    // reads what the Java code reads there (round 6, below): zeros behind the stream's last real bits, and from bit 64 on the container's own
    // top bits again -- its shifts wrap.
    bool bad = false;
    uint64_t w = 0, nextWord = 0, startWord = 0;
    int32_t have = 0, ptr = 8, rem = 0;
    int32_t sStart = 8;  // (a lane without work: "the stream" is empty and ends at byte 8 of a source that has at least a frame and a block header)
    if (live) {
        const int32_t end = d.seqEnd, size = end - d.seqStart;
        if (size < 1 || end < 8 || d.seqStart < 8) {  // (Initializer.initialize :110-130; a stream always has a frame and a block header before it)
            bad = true;
        }
        else {
            sStart = d.seqStart;
            const int32_t last = src[end - 1];
            bad = last == 0;
            const int32_t c0 = 8 - zd::highest_bit((uint32_t)(last | 1));
            uint64_t bits = ld8(src + end - 8);
            int32_t consumed = c0;
            if (size >= 8) {
                ptr = end - 8;
            }
            else {
                bits >>= 8 * (8 - size);
                consumed += (8 - size) * 8;
                ptr = sStart;
            }
            w = bits << (consumed & 63);
            have = 64 - consumed;
            rem = 8 * size - c0;
            // the Java reader's container once it sits at the stream's start (BitInputStream.Loader.load :176-204: `bits = getLong(startAddress)`; a stream
            // shorter than 8 bytes: what readTail :39-59 built) -- only read where a sequence's extra bits run past the stream's start (below)
            startWord = size >= 8 ? ld8(src + sStart) : bits;
        }
    }
    nextWord = ld8(src + ptr - 8);
    auto refill = [&]() {
        int32_t k = (64 - have) >> 3;
        const int32_t avail = ptr - sStart;
        k = k < avail ? k : avail;
        k = k < 0 ? 0 : (k > 8 ? 8 : k);
        w |= have >= 64 ? 0ull : (nextWord >> (have & 63));
        have += 8 * k;
        ptr -= k;
        nextWord = ld8(src + ptr - 8);
    };
    auto field = [&](int32_t at, int32_t n) -> int32_t {  // n <= 31
        const uint32_t hi = (uint32_t)((w << (at & 63)) >> 32);
        return (int32_t)((hi >> 1) >> ((31 - n) & 31));
    };
    // the same for a field that ends inside the accumulator's upper half (at + n <= 32): one bit-field extract
    auto field32 = [](uint32_t hi, int32_t end, int32_t n) -> int32_t {  // end = at + n
#if defined(__HIP_DEVICE_COMPILE__)
        return (int32_t)__builtin_amdgcn_ubfe(hi, (uint32_t)(32 - end), (uint32_t)n);
#else
        return n == 0 ? 0 : (int32_t)((hi >> ((32 - end) & 31)) & ((1u << n) - 1u));
#endif
    };
    int32_t nDecoded = 0;
    // the repeat-offset history; MB: "what it was before the block", entries 0 .. 2 (see above)
    int32_t p0 = MB ? sx2::REP_SENTINEL : 1, p1 = MB ? (sx2::REP_SENTINEL | (1 << 16)) : 4, p2 = MB ? (sx2::REP_SENTINEL | (2 << 16)) : 8;
    {
        // initial states in stream order LL, OF, ML (:378-386)
        int32_t sLL = field(0, logLL) & 511;
        int32_t sOF = field(logLL, logOF) & 255;
        int32_t sML = field(logLL + logOF, logML) & 511;
        {
            const int32_t n0 = logLL + logOF + logML;
            w <<= n0;
            have -= n0;
            rem -= n0;
        }
        int32_t sequenceCount = live && !bad ? d.nbSeq : 0;
        uint64_t lastRecord = 0;
        uint32_t tabLL = codeLL, tabML = codeML;
        settle(nextWord);  // (nothing the loop reads is still on its way when it is entered: its waits are then for its own loads alone)
        settle(tabLL);
        settle(tabML);
        // ZstdFrameDecompressor.java:388-486, straight-line: an irregular stream sets `bad` and keeps decoding
        // harmless garbage (every index is masked) until the count runs out
        while (__ballot(sequenceCount > 0) != 0) {  // (uniform)
            const bool act = sequenceCount > 0;  // this lane's item has a sequence to decode in this step
            sequenceCount -= act ? 1 : 0;
            const bool over = rem < 0;                 // Loader.load() :171-175
            bad |= act && over && sequenceCount != 0;  // "Not all sequences were consumed"
            sequenceCount = over ? 0 : sequenceCount;
            refill();
            const uint32_t eLL = tLL[sLL], eML = tML[sML], eOF = tOF[sOF];
            const int32_t cLL = (int32_t)(eLL & 63), cML = (int32_t)(eML & 63), cOF = (int32_t)(eOF & 63);
            const uint32_t tl = (uint32_t)__shfl((int32_t)tabLL, cLL), tm = (uint32_t)__shfl((int32_t)tabML, cML);
            const int32_t xLL = (int32_t)(tl >> 24), xML = (int32_t)(tm >> 24), xOF = cOF & 31;
            // codes beyond the tables are only reachable through a table the Java reader would also have rejected or mis-indexed; offset
            // codes above 24 give offsets no window allows (checked below) and extra-bit counts the shortcut above does not cover
            // (only where the Java loop decodes the sequence at all: at an overflow it leaves before it reads a code -- an item with nothing left to decode is not
            // irregular for the garbage this loop computes in that step.  Until round 6 these checks ran in every step: such items went to the fallback list, and
            // the incremental reader, which has none, refused streams the reference reads: tools/fuzz_zstd_tail.py)
            bad |= act && !over && (cLL > 35 || cML > 52 || cOF > 24);
            // extra bits are read in the order offset, match length, literal length
            int32_t vOF = (cOF < 2 ? cOF : (1 << xOF) - 3) + field(0, xOF);
            int32_t matchLength = (int32_t)(tm & 0xFFFFFF) + field(xOF, xML);
            int32_t literalsLength = (int32_t)(tl & 0xFFFFFF) + field(xOF + xML, xLL);
            const int32_t xsum = xLL + xML + xOF;
            w <<= (xsum & 63);
            have -= xsum;
            rem -= xsum;
            // The extra bits ran past the stream's start: a corrupt stream's last sequence or two.  The Java reader does not notice here -- its next load() reports
            // the overflow, and with no sequence left to decode that is not an error (:395-399) -- so the sequence it executes is made of what peekBits returns:
            // at that point its container sits at the stream's start (every other position of Loader.load leaves more than 56 bits in it, more than three extra
            // fields take), so a field at container bit q reads `((C << (q & 63)) >>> 1) >>> (63 - n)`: real bits, then zeros, and from q = 64 on C's top bits again.
            // (Until round 6 such an item went to the fallback list; the incremental reader has none and refused a stream the reference reads:
            // tools/fuzz_zstd_stream.py, profiles/r06_notes.md.)
            // (The block-slot instantiation -- frames of several blocks and the incremental reader's steps, which have no other decoder behind them -- reads them; the
            // single-block instantiation, whose step this check and its merge made 3.6 % longer on the corpus batch, keeps handing such items to the one-kernel
            // decoder, which reads the same bits its own way.)
            const bool overrun = act && !over && rem < 0;
            if (!MB) {
                bad |= overrun;
            }
            else if (__ballot(overrun) != 0) {  // (uniform, rare)
                if (overrun) {
                    const int32_t q0 = 64 - (rem + xsum);  // Java's bitsConsumed at the step's start: 64 - the stream's remaining bits (in 8 .. 64 here)
                    auto peek = [&](int32_t q, int32_t n) -> int32_t { return n == 0 ? 0 : (int32_t)(((startWord << (q & 63)) >> 1) >> ((63 - n) & 63)); };
                    vOF = (cOF < 2 ? cOF : (1 << xOF) - 3) + peek(q0, xOF);
                    matchLength = (int32_t)(tm & 0xFFFFFF) + peek(q0 + xOF, xML);
                    literalsLength = (int32_t)(tl & 0xFFFFFF) + peek(q0 + xOF + xML, xLL);
                }
            }
            if (xsum > 64 - 7 - (9 + 9 + 8)) {
                refill();
            }
            // state updates in the order LL, ML, OF
            const int32_t nLL = (int32_t)(eLL >> 6), nML = (int32_t)(eML >> 6), nOF = (int32_t)(eOF >> 6);
            const int32_t nbLL = (logLL - (31 - __builtin_clz((uint32_t)nLL | 1u))) & 15;
            const int32_t nbML = (logML - (31 - __builtin_clz((uint32_t)nML | 1u))) & 15;
            const int32_t nbOF = (logOF - (31 - __builtin_clz((uint32_t)nOF | 1u))) & 15;
            const uint32_t hiS = (uint32_t)(w >> 32);  // (the three counts are at most 9 + 9 + 8 bits: all inside the upper half)
            const int32_t endML = nbLL + nbML, nbsum = endML + nbOF;
            sLL = ((nLL << nbLL) - (1 << logLL) + field32(hiS, nbLL, nbLL)) & 511;
            sML = ((nML << nbML) - (1 << logML) + field32(hiS, endML, nbML)) & 511;
            sOF = ((nOF << nbOF) - (1 << logOF) + field32(hiS, nbsum, nbOF)) & 255;
            {
                w <<= nbsum;
                have -= nbsum;
                rem -= nbsum;
            }
            // repeat-offset history, :419-452.  (Round 6 moved it into a kernel of its own -- a lane per item on every SIMD, the records rewritten in place -- to shorten
            // this loop: the loop lost 14 % of its instructions and 4 % of its time (28.3 -> 27.0 ms), the new kernel took 12.6 ms: a lane's serial walk over records
            // in memory is a memory round trip a step.  Back here: profiles/r06_notes.md.)
            const int32_t raw = vOF + ((cOF <= 1 && cLL == 0) ? 1 : 0);
            const bool rep = cOF <= 1;
            // (a sentinel's low bits count the "- 1" steps: at most one per sequence, fewer than 2^16)
            int32_t temp = raw == 3 ? ((MB && p0 >= sx2::REP_SENTINEL) ? p0 + 1 : p0 - 1) : (raw == 1 ? p1 : p2);
            temp = temp == 0 ? 1 : temp;
            // (not at an overflow: the Java loop leaves there before it decodes anything -- the history a block leaves behind is that of its last decoded sequence)
            const bool produce = act && !over;
            const bool shift2 = produce && (rep ? (raw != 0 && raw != 1) : true);  // p2 = p1
            const bool shift1 = produce && (rep ? raw != 0 : true);                // p1 = p0, p0 = new
            const int32_t offset = rep ? (raw != 0 ? temp : p0) : raw;
            p2 = shift2 ? p1 : p2;
            p1 = shift1 ? p0 : p1;
            p0 = shift1 ? offset : p0;
            // an offset beyond 2^24 cannot be a valid back-reference (the window is at most 2^23); keeps the record fields in range and
            // the sentinels apart from real offsets
            bad |= produce && (offset <= 0 || (offset > (1 << 24) && !(MB && rep && offset >= sx2::REP_SENTINEL)));
            // 8 bytes per lane and step, straight to the arena (the L2 merges a line's pieces; staging 16 records per lane in LDS and storing lines --
            // what the quad version did -- measured no faster), UNCONDITIONALLY: a lane that has no record in this step stores its last one again
            const uint64_t record = (uint64_t)(uint32_t)literalsLength | ((uint64_t)(uint32_t)matchLength << 18) | ((uint64_t)(uint32_t)(offset & 0xFFFFFFF) << 36);
            lastRecord = produce ? record : lastRecord;
            const int32_t at = produce ? nDecoded : (nDecoded > 0 ? nDecoded - 1 : 0);
            rec[at] = lastRecord;
            nDecoded += produce ? 1 : 0;
        }
    }
    if (live) {
        if (bad) {
            slot_to_fallback<MB>(p, slot, 3);
        }
        else {
            p.desc[slot].nDecoded = nDecoded;
            if (MB) {
                p.mb[slot].repOut[0] = p0;
                p.mb[slot].repOut[1] = p1;
                p.mb[slot].repOut[2] = p2;
            }
        }
    }
    if (!MB) {  // the tile's long-sequence items, for K4's choice (one atomic per wavefront)
        const bool isLong = live && !bad && long_sequences(a.dstCap[block], nDecoded);
        const unsigned long long lm = __ballot(isLong);
        if (isLong && lane == (int)__builtin_ctzll(lm)) {
            atomicAdd(p.longCount, (int32_t)__popcll(lm));
        }
    }
}

// ---- K4: execute ----
template <int GS, int IN_RING, int OUT_RING>
__global__ __launch_bounds__(256) void zstd_pipe_execute_kernel(BatchArgs a, zp::Pipe p, int32_t mode)
{
    using namespace zp;
    ACHIP_DYNAMIC_LDS(smem);
    static_assert(GS == 4, "the 16-byte far-match prefetch is spread as one dword per lane");
    constexpr int GROUPS_PER_WG = 256 / GS;
    const int g = threadIdx.x & (GS - 1);
    const int grp = threadIdx.x / GS;
    const int32_t slot = blockIdx.x * GROUPS_PER_WG + grp;
    if (slot >= p.count) {
        return;
    }
    const Desc d = p.desc[slot];
    if (d.state != 1) {
        return;
    }
    const int32_t block = p.first + slot;
    if (mode == 2 && !zp::item_takes_rings(p, a.dstCap[block], d.nDecoded)) {
        return;  // (auto: the record executor takes this item)
    }
    const uint8_t* src = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t outLimit = a.dstCap[block];
    const uint8_t* lit = d.litMode == 0 ? src + d.litSrc : p.lit + (size_t)d.litBase * 64;
    const int32_t litSize = d.litSize;
    const uint64_t* rec = p.seq + d.seqBase;
    const int32_t nSeq = d.nDecoded;

    Rings<GS, IN_RING, OUT_RING> R;
    R.init(smem + grp * (IN_RING + OUT_RING + a.ringPad), smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING, lit, litSize, out, g,
           a.ringPad >= 16 * GS ? smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING + OUT_RING : nullptr);

    int32_t output = 0;
    int32_t literalsInput = 0;
    bool bad = false;
    const int laneBase = (int)(threadIdx.x & 63) - g;
    // GS records per step, one per lane; the next step's records are requested before this step runs
    uint64_t next = g < nSeq ? rec[g] : 0ull;
    for (int32_t i0 = 0; i0 < nSeq && !bad; i0 += GS) {
        const uint64_t mine = next;
        next = i0 + GS + g < nSeq ? rec[i0 + GS + g] : 0ull;
        const int32_t n = nSeq - i0 < GS ? nSeq - i0 : GS;
        const int32_t myLL = (int32_t)(mine & 0x3FFFF);
        const int32_t myML = (int32_t)((mine >> 18) & 0x3FFFF);
        const int32_t myOF = (int32_t)(mine >> 36);
        // where this lane's sequence lands: exclusive prefix over the step (lanes past n hold zeros)
        int32_t before = 0;
#pragma unroll
        for (int k = 0; k < GS - 1; k++) {
            const int32_t t = ACHIP_GROUP_SHFL(GS, myLL + myML, laneBase, k);
            before += k < g ? t : 0;
        }
        const int32_t myMatchPos = output + before + myLL;
        // A back-reference that copy_match would serve from HBM (farther back than the LDS window) is requested NOW for
        // all GS sequences at once, when its first 16 source bytes are already flushed: GS loads in flight, not one.
        const int32_t flushedAbs = R.flushedV - R.outBase;
        const bool pre = g < n && myOF > Rings<GS, IN_RING, OUT_RING>::LDS_REACH && myOF >= 16 && myMatchPos - myOF >= 0 && myMatchPos - myOF + 16 <= flushedAbs;
        u32x4 far = {0, 0, 0, 0};
        if (pre) {
            far = ld16(out + (myMatchPos - myOF));
        }
        for (int k = 0; k < n; k++) {
            const int32_t ll = ACHIP_GROUP_SHFL(GS, myLL, laneBase, k);
            const int32_t ml = ACHIP_GROUP_SHFL(GS, myML, laneBase, k);
            const int32_t of = ACHIP_GROUP_SHFL(GS, myOF, laneBase, k);
            const int32_t hasFar = ACHIP_GROUP_SHFL(GS, pre ? 1 : 0, laneBase, k);
            // ZstdFrameDecompressor.java:491-496
            if ((int64_t)output + ll + ml > outLimit || literalsInput + ll > litSize || of > output + ll) {
                bad = true;
                break;
            }
            R.copy_literals(literalsInput, output, ll);
            output += ll;
            literalsInput += ll;
            if (hasFar != 0) {
                // dword g of the prefetched 16 bytes goes to lane g
                uint32_t mineW = 0;
#pragma unroll
                for (int q = 0; q < GS; q++) {
                    const uint32_t t = (uint32_t)ACHIP_GROUP_SHFL(GS, (int)(q == 0 ? far.x : (q == 1 ? far.y : (q == 2 ? far.z : far.w))), laneBase, k);
                    mineW = g == q ? t : mineW;
                }
                const int32_t head = ml < 16 ? ml : 16;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (4 * g + t < head) {
                        R.out_put(output + 4 * g + t, (mineW >> (8 * t)) & 0xFF);
                    }
                }
                R.flush_complete(output + head);
                if (ml > head) {
                    R.copy_match(output + head, of, ml - head);
                }
            }
            else {
                R.copy_match(output, of, ml);
            }
            output += ml;
        }
    }
    if (!bad) {
        const int32_t last = litSize - literalsInput;  // copyLastLiteral :518-525
        if ((int64_t)output + last > outLimit) {
            bad = true;
        }
        else {
            R.copy_literals(literalsInput, output, last);
            output += last;
            R.flush_all(output);
        }
    }
    if (g == 0) {
        if (bad) {
            to_fallback(p, slot, 4);
        }
        else if (d.hasChecksum) {
            p.desc[slot].outSize = output;
        }
        else {
            a.outLen[block] = output;
            a.status[block] = 0;
            a.errOffset[block] = 0;
        }
    }
}

// ---- K4, second version (the default): a WAVEFRONT per item runs the records through the executor of the LZ4 / Snappy two-pass decoders
// (achip_seqexec2.h exec_records: a 4 KiB circular LDS window, every global load one batch ahead, long sequences cut into pieces of
// 16 + 16 bytes on the fly).  The records are checked group by group (ZstdFrameDecompressor.java:491-496); an item with a record that
// runs outside its buffers goes to the fallback list like everything irregular. ----
// Which of the two runs an item is decided per item (mode 2, the default): the ring version is the faster one on long sequences (measured on
// 128 KiB frames: fragments data -- 100 bytes per sequence -- 720 against 600 GiB/s; corpus -- 13 bytes per sequence -- 83 against 105), and the
// item's capacity over its sequence count is what both kernels can see (an upper bound of the bytes per sequence: a caller that hands over far
// more capacity than the frame needs gets the ring version).
int g_zstd_pipe_exec = 2;  // context option zstd.decompress.exec: 2 = per item (default), 1 = this kernel, 0 = the ring version above

template <int WIN = sx2::WIN_DEFAULT>
__global__ __launch_bounds__(64) void zstd_pipe_execute2_kernel(BatchArgs a, zp::Pipe p, int32_t mode)
{
    using namespace zp;
    __shared__ __attribute__((aligned(16))) uint8_t win[WIN + 16];
    const int32_t slot = blockIdx.x;
    if (slot >= p.count) {
        return;
    }
    const Desc d = p.desc[slot];
    if (d.state != 1) {
        return;
    }
    const int lane = threadIdx.x;
    const int32_t block = p.first + slot;
    if (mode == 2 && zp::item_takes_rings(p, a.dstCap[block], d.nDecoded)) {
        return;  // (auto: the ring version takes this item)
    }
    const uint8_t* src = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const uint8_t* lit = d.litMode == 0 ? src + d.litSrc : p.lit + (size_t)d.litBase * 64;
    sx2::RecordSource S{p.seq + d.seqBase, d.nDecoded};
    bool bad = false;
    const int32_t output = sx2::exec_records<WIN>(win, S, lit, d.litSize, out, a.dstCap[block], lane, bad);
    if (lane == 0) {
        if (bad) {
            to_fallback(p, slot, 4);
        }
        else if (d.hasChecksum) {
            p.desc[slot].outSize = output;
        }
        else {
            a.outLen[block] = output;
            a.status[block] = 0;
            a.errOffset[block] = 0;
        }
    }
}

// ---- K5: checksum ----
__global__ __launch_bounds__(64) void zstd_pipe_checksum_kernel(BatchArgs a, zp::Pipe p)
{
    using namespace zp;
    const int lane = threadIdx.x;
    const int q = lane >> 2;
    const int s = lane & 3;
    const int32_t slot = blockIdx.x * ITEMS_PER_WAVE + q;
    if (slot >= p.count) {
        return;
    }
    const Desc d = p.desc[slot];
    if (d.state != 1 || !d.hasChecksum) {
        return;
    }
    const int32_t block = p.first + slot;
    const uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t len = d.outSize;
    const uint64_t hash = quad_xxh64(out, len, 0, s, lane - s);  // XxHash64.java:182-291 (xxhash.hip)
    if (s == 0) {
        if ((uint32_t)hash == d.checksum) {
            a.outLen[block] = len;
            a.status[block] = 0;
            a.errOffset[block] = 0;
        }
        else {
            to_fallback(p, slot, 5);
        }
    }
}

// ================================================================================================================================
// Multi-block frames (what ZstdOutputStream.java:154-221 writes -- one frame, a block per 128 KiB, the window, the
// repeat-offset history and the entropy tables carried from block to block -- and what ZstdFrameCompressor / libzstd write for inputs
// beyond one block).  K1 lists the items whose first block is not a frame's only compressed block; then
//
//   count    a lane per listed item     walks the frame's block headers: one frame that fills the item exactly, block sizes in range;
//                                       adds up what the blocks' headers announce for the literal and the sequence arena
//   scan     one wavefront              gives every item its range of block indices; the host reads the item records (the one
//                                       synchronisation of a decode call), asks for scratch by what they need, and cuts the list into
//                                       PASSES: runs of items whose block slots, literals and sequences fit the scratch
//   fill     a lane per item            walks again: a slot per block with its place, kind and LINKS -- the slot whose Huffman table
//                                       treeless literals reuse, the slots whose FSE tables repeat mode reuses
//   parse    a wavefront per block      K1's block part (parse_block); raw blocks become a literal run, RLE blocks one record
//   K2, K3   16 blocks per wavefront    as above, on block slots, tables fetched through the links; K3 cannot know the repeat-offset
//                                       history at a block's start and leaves sentinels (achip_seqexec2.h REP_SENTINEL)
//   execute  a wavefront per item       the blocks in order through exec_records, ONE LDS window and one output position for the frame
//                                       (matches reach into earlier blocks), the repeat-offset history resolved from block to block
//   checksum a quad per item            XXH64 of the frame's output
//
// Anything irregular in any block sends the ITEM to the fallback list (the one-kernel decoder reports the Java-exact status).
// ================================================================================================================================
namespace zp {

struct MbWalk {
    int32_t n;            // blocks
    int32_t hasChecksum;
    uint32_t checksum;
    uint32_t litUnits;    // literal arena units / sequence records the blocks' headers announce (what parse_block and the RLE blocks will take)
    uint32_t seqs;
    bool ok;
};

// ZstdFrameDecompressor.java:150-214 for an item K1 has looked at (magic, frame header, window size are fine): the block headers up to
// the last block, the checksum, nothing behind it.  FILL: blocks[0 .. n) receive their place, kind and links (slot numbers count from
// firstSlot).  Lane-private.
template <bool FILL>
__device__ MbWalk mb_walk(const Ctx& c, MbBlock* blocks, int32_t firstSlot, int32_t itemSlot, int32_t maxBlocks)
{
    MbWalk w;
    w.n = 0;
    w.hasChecksum = 0;
    w.checksum = 0;
    w.litUnits = 0;
    w.seqs = 0;
    w.ok = false;
    const int64_t inputLimit = c.inLen;
    const int32_t fhd = (int32_t)rd_le(c, 4, 1);
    const bool singleSegment = (fhd & 0x20) != 0;
    const int32_t csDesc = fhd >> 6;
    int64_t input = 4 + 1 + (singleSegment ? 0 : 1) + (csDesc == 0 ? (singleSegment ? 1 : 0) : (1 << csDesc));  // (no dictionary id: K1)
    w.hasChecksum = (fhd & 4) != 0 ? 1 : 0;
    int32_t lastHuf = -1, lastLL = -1, lastOF = -1, lastML = -1;
    for (;;) {
        if (input + 3 > inputLimit || w.n >= maxBlocks) {
            return w;
        }
        const int32_t header = (int32_t)rd_le(c, (int32_t)input, 3);
        input += 3;
        const int32_t type = (header >> 1) & 3;
        const int32_t size = (header >> 3) & 0x1FFFFF;
        int64_t adv = size;
        if (type == 1) {
            adv = 1;
            if (size > 0x40000) {
                return w;  // (the record that stands for the block holds size - 1 in 18 bits)
            }
        }
        else if (type == 2) {
            if (size > MAX_BLOCK_SIZE || size < 3) {
                return w;
            }
        }
        else if (type == 3) {
            return w;
        }
        if (input + adv > inputLimit) {
            return w;
        }
        {
            const int32_t slot = firstSlot + w.n;
            MbBlock b;
            b.itemSlot = itemSlot;
            b.srcPos = (int32_t)input;
            b.size = size;
            b.kind = type;
            b.hufSlot = -1;
            b.fseSlot[0] = b.fseSlot[1] = b.fseSlot[2] = -1;
            b.repOut[0] = sx2::REP_SENTINEL;
            b.repOut[1] = sx2::REP_SENTINEL | (1 << 16);
            b.repOut[2] = sx2::REP_SENTINEL | (2 << 16);
            b.pad = 0;
            if (type == 1) {
                w.seqs += size > 0 ? 1 : 0;  // (the record that stands for the block)
            }
            if (type == 2) {
                // the literals section's header gives its sizes (:708-858), behind it the sequences section's header gives the count and
                // names the table modes (:328-349); the parse stage checks every field again
                const int32_t bs = (int32_t)input, bl = (int32_t)input + size;
                const int32_t b0 = (int32_t)rd_le(c, bs, 1);
                const int32_t lbt = b0 & 3, sf = (b0 >> 2) & 3;
                int32_t section, regenerated;
                if (lbt < 2) {
                    const int32_t hdr = sf == 1 ? 2 : (sf == 3 ? 3 : 1);
                    regenerated = sf == 1 ? (int32_t)rd_le(c, bs, 2) >> 4 : (sf == 3 ? (int32_t)(rd_le(c, bs, 3) & 0xFFFFFF) >> 4 : b0 >> 3);
                    section = hdr + (lbt == 0 ? regenerated : 1);
                }
                else {
                    const int32_t hdr = sf < 2 ? 3 : (sf == 2 ? 4 : 5);
                    const uint64_t h = rd_le(c, bs, 5);
                    const int32_t csize = sf < 2 ? (int32_t)((h >> 14) & 0x3FF) : (sf == 2 ? (int32_t)((h >> 18) & 0x3FFF) : (int32_t)((h >> 22) & 0x3FFFF));
                    regenerated = sf < 2 ? (int32_t)((h >> 4) & 0x3FF) : (sf == 2 ? (int32_t)((h >> 4) & 0x3FFF) : (int32_t)((h >> 4) & 0x3FFFF));
                    section = hdr + csize;
                    if (lbt == 2) {
                        lastHuf = slot;
                    }
                    b.hufSlot = lastHuf;
                }
                if (lbt != 0 && regenerated <= MAX_BLOCK_SIZE) {
                    w.litUnits += (uint32_t)(regenerated + 64 + 63) >> 6;  // (parse_block's request)
                }
                const int32_t seqPos = bs + section;
                if (seqPos < bl) {
                    const int32_t cnt = (int32_t)rd_le(c, seqPos, 1);
                    const int32_t modesPos = seqPos + 1 + (cnt == 255 ? 2 : (cnt > 127 ? 1 : 0));
                    if (cnt != 0 && modesPos < bl) {
                        w.seqs += cnt == 255 ? (uint32_t)rd_le(c, seqPos + 1, 2) + 0x7F00u : (cnt > 127 ? (uint32_t)(((cnt - 128) << 8) + (int32_t)rd_le(c, seqPos + 1, 1)) : (uint32_t)cnt);
                        const int32_t modes = (int32_t)rd_le(c, modesPos, 1);
                        lastLL = ((modes >> 6) & 3) != 3 ? slot : lastLL;
                        lastOF = ((modes >> 4) & 3) != 3 ? slot : lastOF;
                        lastML = ((modes >> 2) & 3) != 3 ? slot : lastML;
                        b.fseSlot[0] = lastLL;
                        b.fseSlot[1] = lastOF;
                        b.fseSlot[2] = lastML;
                    }
                }
            }
            if (FILL) {
                blocks[w.n] = b;
            }
        }
        w.n++;
        input += adv;
        if ((header & 1) != 0) {
            break;
        }
    }
    if (w.hasChecksum) {
        if (input + 4 > inputLimit) {
            return w;
        }
        w.checksum = (uint32_t)rd_le(c, (int32_t)input, 4);
        input += 4;
    }
    w.ok = input == inputLimit;  // (anything behind the frame -- another frame, garbage -- is the one-kernel decoder's)
    return w;
}

__device__ __forceinline__ Ctx mb_source(const BatchArgs& a, int32_t item, int lane)
{
    Ctx c;
    c.in = a.srcBase + a.srcOff[item];
    c.inLen = a.srcLen[item];
    c.out = nullptr;
    c.outCap = 0;
    c.lit = nullptr;
    c.R = nullptr;
    c.lane = lane;
    c.detail = 0;
    c.errOff = 0;
    return c;
}

__device__ __forceinline__ bool mb_in_pass(const Pipe& p, const MbItem& it, int32_t j) { return it.state == 1 && j >= p.itemFirst && j < p.itemEnd; }

}  // namespace zp

__global__ __launch_bounds__(64) void zstd_mb_count_kernel(BatchArgs a, zp::Pipe p)
{
    using namespace zp;
    const int32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j >= *p.mbCount) {
        return;
    }
    const int32_t item = p.mbList[j];
    const Ctx c = mb_source(a, item, threadIdx.x);
    MbWalk w = mb_walk<false>(c, nullptr, 0, j, p.mbSlots);
    w.ok = w.ok && w.litUnits <= p.mbLitCap && w.seqs <= p.mbSeqCap;  // (it must fit a pass of its own)
    MbItem it;
    it.item = item;
    it.firstBlock = 0;
    it.nBlocks = w.ok ? w.n : 0;
    it.state = w.ok ? 1 : 0;
    it.hasChecksum = w.hasChecksum;
    it.checksum = w.checksum;
    it.outSize = 0;
    it.litUnits = w.ok ? w.litUnits : 0;
    it.seqs = w.ok ? w.seqs : 0;
    it.pad[0] = it.pad[1] = it.pad[2] = 0;
    p.mbItem[j] = it;
    if (!w.ok) {
        atomicAdd(p.fallbackCount + 32 + 6, 1);
        p.fallback[atomicAdd(p.fallbackCount, 1)] = item;
    }
}

// one wavefront: exclusive scan of the items' block counts
__global__ __launch_bounds__(64) void zstd_mb_scan_kernel(zp::Pipe p)
{
    using namespace zp;
    const int lane = threadIdx.x;
    const int32_t n = *p.mbCount;
    int32_t base = 0;
    for (int32_t j0 = 0; j0 < n; j0 += 64) {  // (uniform)
        const int32_t j = j0 + lane;
        const int32_t nb = j < n ? p.mbItem[j].nBlocks : 0;
        const int32_t incl = sx::wave_scan_incl(nb, lane);
        if (j < n) {
            p.mbItem[j].firstBlock = base + incl - nb;
        }
        base += sx::wave_bcast(incl, 63);
    }
    if (lane == 0) {
        p.mbCount[1] = base;
    }
}

// no scratch for the multi-block stages (or less than the item needs): items [first, end) of the list go to the fallback list
__global__ __launch_bounds__(64) void zstd_mb_release_kernel(zp::Pipe p, int32_t first, int32_t end)
{
    using namespace zp;
    const int32_t j = first + blockIdx.x * 64 + threadIdx.x;
    if (j < end && j < *p.mbCount) {
        mb_to_fallback(p, j, 6);
    }
}

__global__ __launch_bounds__(64) void zstd_mb_fill_kernel(BatchArgs a, zp::Pipe p)
{
    using namespace zp;
    const int32_t j = p.itemFirst + blockIdx.x * 64 + threadIdx.x;
    if (j >= p.itemEnd) {
        return;
    }
    const MbItem it = p.mbItem[j];
    if (!mb_in_pass(p, it, j)) {
        return;
    }
    const Ctx c = mb_source(a, it.item, threadIdx.x);
    const int32_t firstSlot = it.firstBlock - p.passFirst;
    mb_walk<true>(c, p.mb + firstSlot, firstSlot, j, p.mbSlots);
}

// a wavefront per block slot: the block's Desc (tables to the slot's table space, arena ranges)
__global__ __launch_bounds__(64) void zstd_mb_parse_kernel(BatchArgs a, zp::Pipe p, const zd::FseTable* __restrict__ dflt)
{
    using namespace zp;
    __shared__ TableShared sh;
    const int32_t slot = blockIdx.x;
    const MbBlock b = p.mb[slot];
    if (b.kind < 0) {
        return;
    }
    const int lane = threadIdx.x;
    const MbItem it = p.mbItem[b.itemSlot];
    Ctx c = mb_source(a, it.item, lane);
    Desc d;
    d.state = 0;
    d.litMode = 0;
    d.litSize = 0;
    d.litSrc = 0;
    d.hufLog = 0;
    d.nStreams = 0;
    for (int i = 0; i < 4; i++) {
        d.sStart[i] = 0;
        d.sEnd[i] = 0;
    }
    d.nbSeq = 0;
    d.seqStart = d.seqEnd = 0;
    d.log[0] = d.log[1] = d.log[2] = 0;
    d.seqBase = 0;
    d.nDecoded = 0;
    d.hasChecksum = 0;
    d.checksum = 0;
    d.outSize = 0;
    d.litBase = 0;
    for (int i = 0; i < 6; i++) {
        d.pad[i] = 0;
    }
    bool ok = it.state == 1;
    if (ok && b.kind == 0) {
        // decodeRawBlock :211-217: the block is one run of literals that lie in the source
        d.litSrc = b.srcPos;
        d.litSize = b.size;
    }
    else if (ok && b.kind == 1) {
        // decodeRleBlock :219-250: one literal (the byte, in the source) and a match of offset 1 over the rest
        d.litSrc = b.srcPos;
        d.litSize = b.size > 0 ? 1 : 0;
        if (b.size > 0) {
            uint32_t at = 0;
            if (lane == 0) {
                at = atomicAdd(p.seqCursor, 1u);
            }
            at = __shfl(at, 0);
            ok = (uint64_t)at + 1 <= p.seqCap;
            if (ok && lane == 0) {
                p.seq[at] = 1ull | ((uint64_t)(uint32_t)(b.size - 1) << 18) | (1ull << 36);
            }
            d.seqBase = at;
            d.nDecoded = ok ? 1 : 0;
        }
    }
    else if (ok) {
        const int32_t fseBefore = ((b.fseSlot[0] >= 0 && b.fseSlot[0] != slot) ? 1 : 0) | ((b.fseSlot[1] >= 0 && b.fseSlot[1] != slot) ? 2 : 0) | ((b.fseSlot[2] >= 0 && b.fseSlot[2] != slot) ? 4 : 0);
        ok = parse_block(c, sh, p, slot, dflt, d, b.srcPos, b.size, b.hufSlot >= 0 && b.hufSlot != slot, fseBefore);
    }
    d.state = ok ? 1 : 0;
    if (lane == 0) {
        p.desc[slot] = d;
        if (!ok) {
            mb_to_fallback(p, b.itemSlot, 1);
        }
    }
}

// K4 for multi-block frames: a wavefront per item
template <int WIN = sx2::WIN_DEFAULT>
__global__ __launch_bounds__(64) void zstd_mb_execute_kernel(BatchArgs a, zp::Pipe p)
{
    using namespace zp;
    __shared__ __attribute__((aligned(16))) uint8_t win[WIN + 16];
    const int32_t j = p.itemFirst + blockIdx.x;
    const MbItem it = p.mbItem[j];
    if (!mb_in_pass(p, it, j)) {
        return;
    }
    const int lane = threadIdx.x;
    const uint8_t* src = a.srcBase + a.srcOff[it.item];
    uint8_t* out = a.dstBase + a.dstOff[it.item];
    const int32_t outLimit = a.dstCap[it.item];
    int32_t rep0 = 1, rep1 = 4, rep2 = 8;  // reset() :199-203
    int32_t output = 0;
    bool bad = false;
    for (int32_t i = 0; i < it.nBlocks && !bad; i++) {  // (uniform)
        const int32_t slot = it.firstBlock - p.passFirst + i;
        const Desc d = p.desc[slot];
        const MbBlock b = p.mb[slot];
        if (d.state != 1) {
            bad = true;
            break;
        }
        const uint8_t* lit = d.litMode == 0 ? src + d.litSrc : p.lit + (size_t)d.litBase * 64;
        sx2::RecordSource S{p.seq + d.seqBase, d.nDecoded, rep0, rep1, rep2};
        output = sx2::exec_records<WIN>(win, S, lit, d.litSize, out, outLimit, lane, bad, output);
        const int32_t n0 = sx2::rep_resolve(b.repOut[0], rep0, rep1, rep2), n1 = sx2::rep_resolve(b.repOut[1], rep0, rep1, rep2), n2 = sx2::rep_resolve(b.repOut[2], rep0, rep1, rep2);
        rep0 = n0;
        rep1 = n1;
        rep2 = n2;
        wave_sync();  // (the block's last bytes have left the window before the next block writes it)
    }
    if (lane == 0) {
        if (bad) {
            mb_to_fallback(p, j, 4);
        }
        else if (it.hasChecksum) {
            p.mbItem[j].outSize = output;
        }
        else {
            a.outLen[it.item] = output;
            a.status[it.item] = 0;
            a.errOffset[it.item] = 0;
            atomicAdd(p.mbCount + 2, 1);
        }
    }
}

// K5 for multi-block frames: a quad per item
__global__ __launch_bounds__(64) void zstd_mb_checksum_kernel(BatchArgs a, zp::Pipe p)
{
    using namespace zp;
    const int lane = threadIdx.x;
    const int q = lane >> 2;
    const int s = lane & 3;
    const int32_t j = p.itemFirst + blockIdx.x * ITEMS_PER_WAVE + q;
    if (j >= p.itemEnd) {
        return;
    }
    const MbItem it = p.mbItem[j];
    if (!mb_in_pass(p, it, j) || !it.hasChecksum) {
        return;
    }
    const uint64_t hash = quad_xxh64(a.dstBase + a.dstOff[it.item], it.outSize, 0, s, lane - s);  // :192-204
    if (s == 0) {
        if ((uint32_t)hash == it.checksum) {
            a.outLen[it.item] = it.outSize;
            a.status[it.item] = 0;
            a.errOffset[it.item] = 0;
            atomicAdd(p.mbCount + 2, 1);
        }
        else {
            mb_to_fallback(p, j, 5);
        }
    }
}

// the one-kernel decoder, run over a list of items (zstd_decompress.hip)
hipError_t launch_zstd_decompress_prepare(hipStream_t stream, void* generalScratch, const zd::FseTable** dflt);
hipError_t launch_zstd_decompress_list(const BatchArgs& a, hipStream_t stream, void* generalScratch, const int32_t* list, const int32_t* listCount);
int64_t zstd_decompress_general_scratch_bytes();

namespace {
constexpr int32_t PIPE_TILE_DEFAULT = 65536;              // items per pass through the five stages (K4 wants >= 64 Ki items in flight: 16 per wavefront)
constexpr uint32_t PIPE_LIT_PER_ITEM = 80 * 1024 / 64;   // literal arena: average 64-byte units per item ...
constexpr uint32_t PIPE_LIT_FLOOR = 16 * (zp::LIT_STRIDE / 64 + 1);  // ... plus 16 blocks of the maximum size
constexpr uint32_t PIPE_SEQ_PER_ITEM = 20480;     // sequence arena: average records per item (text: 10-16 K per 128 KiB block) ...
constexpr uint32_t PIPE_SEQ_FLOOR = 16 * 43691;   // ... plus room for 16 blocks of the maximum count (128 KiB / 3), so small batches always fit
struct PipeLayout {
    int64_t counters, fallback, mbList, mbItem, desc, huf, fse, lit, seq, general, total;
    int32_t tile;
};
PipeLayout pipe_layout(int32_t nBlocks, int32_t tileMax)
{
    const int32_t PIPE_TILE = tileMax > 0 ? tileMax : PIPE_TILE_DEFAULT;
    PipeLayout L;
    L.tile = nBlocks < PIPE_TILE ? nBlocks : PIPE_TILE;
    auto up = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    int64_t o = 0;
    L.counters = o;
    o = up(o + 256);
    L.fallback = o;
    o = up(o + (int64_t)nBlocks * 4);
    L.mbList = o;
    o = up(o + (int64_t)nBlocks * 4);
    L.mbItem = o;
    o = up(o + (int64_t)nBlocks * sizeof(zp::MbItem));
    L.desc = o;
    o = up(o + (int64_t)L.tile * sizeof(zp::Desc));
    L.huf = o;
    o = up(o + (int64_t)L.tile * zp::HUF_SLOT * 2);
    L.fse = o;
    o = up(o + (int64_t)L.tile * zp::FSE_SLOT * 2);
    L.lit = o;
    o = up(o + ((int64_t)L.tile * PIPE_LIT_PER_ITEM + PIPE_LIT_FLOOR) * 64);
    L.seq = o;
    o = up(o + ((int64_t)L.tile * PIPE_SEQ_PER_ITEM + PIPE_SEQ_FLOOR) * 8);
    L.general = o;
    o = up(o + zstd_decompress_general_scratch_bytes());
    L.total = o;
    return L;
}

// scratch of the multi-block stages for passes "of passBlocks blocks of 128 KiB": the arenas are sized as the single-block pipeline sizes
// them for a tile of that many items, the block slots (7 KiB of table space each) for blocks an eighth of that size -- encoders cut
// blocks short where the statistics change (libzstd on mixed data: ~9 KiB blocks).  The host cuts the list of items into passes whose
// slots, literals and sequences fit (launch_zstd_mb_stages).  The execute stage is one wavefront per FRAME, whatever the frame's size
// (its blocks depend on each other through the window), so what a pass takes is about what its longest frame takes: the default
// (65 536: ~20 GB, asked for when a batch first holds such frames) is meant to hold a batch of some 8 GiB in one pass.
constexpr uint32_t MB_LIT_FLOOR = 3 * PIPE_LIT_FLOOR, MB_SEQ_FLOOR = 3 * PIPE_SEQ_FLOOR;  // room for 48 blocks of the maximum size / count
struct MbLayout {
    int64_t counters, mb, desc, huf, fse, lit, seq, order, total;
    int32_t slots;
    uint32_t litCap, seqCap;  // literal arena (64-byte units), sequence arena (records)
};
// what a pass may hold at most with the option at passBlocks
struct MbCaps {
    int32_t slots;
    uint32_t lit, seq;
};
MbCaps mb_caps(int32_t passBlocks)
{
    return MbCaps{8 * passBlocks, (uint32_t)passBlocks * PIPE_LIT_PER_ITEM + MB_LIT_FLOOR, (uint32_t)passBlocks * PIPE_SEQ_PER_ITEM + MB_SEQ_FLOOR};
}
MbLayout mb_layout(const MbCaps& caps)
{
    MbLayout L;
    L.slots = caps.slots;
    L.litCap = caps.lit;
    L.seqCap = caps.seq;
    auto up = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    int64_t o = 0;
    L.counters = o;
    o = up(o + 256);
    L.mb = o;
    o = up(o + (int64_t)L.slots * sizeof(zp::MbBlock));
    L.desc = o;
    o = up(o + (int64_t)L.slots * sizeof(zp::Desc));
    L.huf = o;
    o = up(o + (int64_t)L.slots * zp::HUF_SLOT * 2);
    L.fse = o;
    o = up(o + (int64_t)L.slots * zp::FSE_SLOT * 2);
    L.lit = o;
    o = up(o + (int64_t)L.litCap * 64 + 4096);  // (+ a chunk to spare: whole-vector stores behind the last literal)
    L.seq = o;
    o = up(o + (int64_t)L.seqCap * 8);
    L.order = o;  // [bucket sizes, bucket fill][the sequence stage's order]
    o = up(o + 2 * zp::ORDER_BUCKETS * 4 + (int64_t)L.slots * 4);
    L.total = o;
    return L;
}
}  // namespace

int64_t zstd_decompress_mb_scratch_bytes(int32_t passBlocks) { return mb_layout(mb_caps(passBlocks < 16 ? 16 : passBlocks)).total; }

int64_t zstd_decompress_pipe_scratch_bytes(int32_t nBlocks, int32_t tileMax) { return pipe_layout(nBlocks, tileMax).total; }

namespace {
// The multi-block stages over the items K1 listed (p: the pipeline's Pipe after its tiles).  The one host synchronisation of a decode
// call is here: the number of listed items and of their blocks decides whether there is anything to do, how much scratch to ask the
// caller for, and how many passes to run.
hipError_t launch_zstd_mb_stages(const BatchArgs& a, hipStream_t stream, zp::Pipe p, const zd::FseTable* dflt, const ZstdMbProvider* mbp)
{
    const MbCaps top = mb_caps(mbp->passBlocks);
    p.mbSlots = top.slots;
    p.mbLitCap = top.lit;
    p.mbSeqCap = top.seq;
    const unsigned perLane = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL(zstd_mb_count_kernel, dim3(perLane), dim3(64), 0, stream, a, p);
    hipLaunchKernelGGL(zstd_mb_scan_kernel, dim3(1), dim3(64), 0, stream, p);
    int32_t totals[2] = {0, 0};  // listed items, their blocks
    hipError_t e = hipMemcpyAsync(totals, p.mbCount, sizeof(totals), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (totals[0] <= 0) {
        return hipSuccess;
    }
    const unsigned items64 = (unsigned)((totals[0] + 63) / 64);
    // what the items need: for the size of the scratch and for the cut into passes
    std::vector<zp::MbItem> items((size_t)totals[0]);
    e = hipMemcpyAsync(items.data(), p.mbItem, items.size() * sizeof(zp::MbItem), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    int64_t needLit = 0, needSeq = 0;
    for (const zp::MbItem& it : items) {
        needLit += it.litUnits;
        needSeq += it.seqs;
    }
    // The scratch: what the batch needs, at most what the option allows -- or, when the device cannot give that, what half the option
    // allows, a quarter ... (more passes).  A call with one 300 KiB frame asks for a megabyte, not for the 20 GB a full pass takes.
    uint8_t* mbase = nullptr;
    MbLayout M = mb_layout(top);
    for (int32_t pb = mbp->passBlocks; totals[1] > 0 && mbase == nullptr && pb >= 16; pb = pb >= 32 ? pb / 2 : 0) {
        const MbCaps c = mb_caps(pb);
        MbCaps want;
        want.slots = (int32_t)(totals[1] < c.slots ? (totals[1] < 64 ? 64 : totals[1]) : c.slots);
        want.lit = (uint32_t)(needLit + 64 < (int64_t)c.lit ? needLit + 64 : (int64_t)c.lit);
        want.seq = (uint32_t)(needSeq + 64 < (int64_t)c.seq ? needSeq + 64 : (int64_t)c.seq);
        M = mb_layout(want);
        mbase = (uint8_t*)mbp->get(mbp->user, M.total);
    }
    if (mbase == nullptr) {
        hipLaunchKernelGGL(zstd_mb_release_kernel, dim3(items64), dim3(64), 0, stream, p, 0, totals[0]);  // (no blocks: every listed item failed its walk already)
        return hipGetLastError();
    }
    const uint32_t litCap = M.litCap, seqCap = M.seqCap;
    p.mbSlots = M.slots;
    p.mb = (zp::MbBlock*)(mbase + M.mb);
    p.desc = (zp::Desc*)(mbase + M.desc);
    p.huf = (uint16_t*)(mbase + M.huf);
    p.fse = (uint16_t*)(mbase + M.fse);
    p.lit = mbase + M.lit;
    p.seq = (uint64_t*)(mbase + M.seq);
    p.seqCap = seqCap;
    p.litCap = litCap;
    p.seqCursor = (uint32_t*)(mbase + M.counters);
    p.litCursor = (uint32_t*)(mbase + M.counters + 4);
    p.first = 0;
    for (int32_t j0 = 0; j0 < totals[0];) {
        // the longest run of items whose blocks, literals and sequences fit one pass (every item fits one alone: the count kernel saw to that)
        int64_t blocks = 0, lit = 0, seq = 0;
        int32_t j1 = j0;
        while (j1 < totals[0] && blocks + items[j1].nBlocks <= M.slots && lit + items[j1].litUnits <= litCap && seq + items[j1].seqs <= seqCap) {
            blocks += items[j1].nBlocks;
            lit += items[j1].litUnits;
            seq += items[j1].seqs;
            j1++;
        }
        if (j1 == j0) {
            // more than a pass holds (the scratch came out smaller than the count kernel assumed): the one-kernel decoder takes the item
            hipLaunchKernelGGL(zstd_mb_release_kernel, dim3(1), dim3(64), 0, stream, p, j0, j0 + 1);
            j0++;
            continue;
        }
        p.itemFirst = j0;
        p.itemEnd = j1;
        p.passFirst = items[j0].firstBlock;
        p.count = (int32_t)blocks;
        j0 = j1;
        if (blocks == 0) {
            continue;  // (items that failed their walk)
        }
        e = hipMemsetAsync(mbase + M.counters, 0, 256, stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(p.mb, 0xFF, (size_t)p.count * sizeof(zp::MbBlock), stream);
        if (e != hipSuccess) return e;
        const unsigned nItems = (unsigned)(p.itemEnd - p.itemFirst);
        hipLaunchKernelGGL(zstd_mb_fill_kernel, dim3((nItems + 63) / 64), dim3(64), 0, stream, a, p);
        hipLaunchKernelGGL(zstd_mb_parse_kernel, dim3((unsigned)p.count), dim3(64), 0, stream, a, p, dflt);
        launch_literals<true>(a, p, stream);
        p.order = nullptr;
        if (p.count > zp::SEQL_ITEMS) {  // (more than one wavefront of slots)
            p.orderHist = (int32_t*)(mbase + M.order);
            p.order = p.orderHist + 2 * zp::ORDER_BUCKETS;
            e = hipMemsetAsync(p.orderHist, 0, 2 * zp::ORDER_BUCKETS * 4, stream);
            if (e != hipSuccess) return e;
            const unsigned g64 = (unsigned)((p.count + 63) / 64);
            hipLaunchKernelGGL(zstd_mb_order_kernel<false>, dim3(g64), dim3(64), 0, stream, p);
            hipLaunchKernelGGL(zstd_mb_order_kernel<true>, dim3(g64), dim3(64), 0, stream, p);
        }
        hipLaunchKernelGGL(zstd_pipe_sequences_lane_kernel<true>, dim3((unsigned)((p.count + seql_items_for(p.count) - 1) / seql_items_for(p.count))), dim3(64 * seql_waves_for(p.count)), 0, stream, a, p, seql_items_for(p.count), seql_items_per_wave(p.count));
        // A frame is one wavefront's work whatever its size: a pass of few frames leaves the LDS idle, and a window of 32 KiB instead of 4 turns most
        // of a text frame's far matches (offsets beyond the window: 64-byte sectors re-read through the L2) into LDS reads
        if (nItems <= 1024) {  // (four wavefronts per CU on 256 CUs: what 32 KiB windows leave room for)
            hipLaunchKernelGGL(zstd_mb_execute_kernel<32768>, dim3(nItems), dim3(64), 0, stream, a, p);
        }
        else {
            hipLaunchKernelGGL(zstd_mb_execute_kernel<>, dim3(nItems), dim3(64), 0, stream, a, p);
        }
        hipLaunchKernelGGL(zstd_mb_checksum_kernel, dim3((nItems + zp::ITEMS_PER_WAVE - 1) / zp::ITEMS_PER_WAVE), dim3(64), 0, stream, a, p);
    }
    return hipGetLastError();
}
}  // namespace
void* zstd_decompress_pipe_general_scratch(void* scratch, int32_t nBlocks, int32_t tileMax) { return (uint8_t*)scratch + pipe_layout(nBlocks, tileMax).general; }

hipError_t launch_zstd_decompress_pipe(const BatchArgs& a, hipStream_t stream, void* scratch, void* generalScratch, int32_t tileMax, const ZstdMbProvider* mbp)
{
    const PipeLayout L = pipe_layout(a.nBlocks, tileMax);
    uint8_t* base = (uint8_t*)scratch;
    const zd::FseTable* dflt = nullptr;
    {
        hipError_t e0 = launch_zstd_decompress_prepare(stream, generalScratch, &dflt);
        if (e0 != hipSuccess) return e0;
    }
    zp::Pipe p;
    p.desc = (zp::Desc*)(base + L.desc);
    p.huf = (uint16_t*)(base + L.huf);
    p.fse = (uint16_t*)(base + L.fse);
    p.lit = base + L.lit;
    p.seq = (uint64_t*)(base + L.seq);
    p.seqCap = (uint32_t)L.tile * PIPE_SEQ_PER_ITEM + PIPE_SEQ_FLOOR;
    p.fallbackCount = (int32_t*)(base + L.counters);
    p.seqCursor = (uint32_t*)(base + L.counters + 64);
    p.litCursor = (uint32_t*)(base + L.counters + 68);
    p.longCount = (int32_t*)(base + L.counters + 72);
    p.litCap = (uint32_t)L.tile * PIPE_LIT_PER_ITEM + PIPE_LIT_FLOOR;
    p.fallback = (int32_t*)(base + L.fallback);
    const bool mbOn = mbp != nullptr && mbp->get != nullptr && mbp->passBlocks >= 16;
    p.mbList = mbOn ? (int32_t*)(base + L.mbList) : nullptr;
    p.mbCount = (int32_t*)(base + L.counters) + 40;
    p.mbItem = (zp::MbItem*)(base + L.mbItem);
    p.mb = nullptr;
    p.passFirst = 0;
    p.itemFirst = p.itemEnd = 0;
    p.mbSlots = 0;
    p.mbLitCap = p.mbSeqCap = 0;
    p.order = nullptr;
    p.orderHist = nullptr;
    hipError_t e = hipMemsetAsync(base + L.counters, 0, 256, stream);
    if (e != hipSuccess) return e;
    for (int32_t first = 0; first < a.nBlocks; first += L.tile) {
        p.first = first;
        p.count = a.nBlocks - first < L.tile ? a.nBlocks - first : L.tile;
        if (first != 0) {
            e = hipMemsetAsync(p.seqCursor, 0, 12, stream);  // sequence and literal cursors, the count of long-sequence items
            if (e != hipSuccess) return e;
        }
        const unsigned w16 = (unsigned)((p.count + zp::ITEMS_PER_WAVE - 1) / zp::ITEMS_PER_WAVE);
        hipLaunchKernelGGL(zstd_pipe_parse_kernel, dim3((unsigned)p.count), dim3(64), 0, stream, a, p, dflt);
        // (items per wavefront in K2 / K3 of 8 instead of 16, and an 8 KiB window for the record executor, were round-2 experiments: measured in
        // round 3 within noise of the defaults on all three data sets -- profiles/r03_notes.md -- and removed)
        launch_literals<false>(a, p, stream);
        hipLaunchKernelGGL(zstd_pipe_sequences_lane_kernel<false>, dim3((unsigned)((p.count + seql_items_for(p.count) - 1) / seql_items_for(p.count))), dim3(64 * seql_waves_for(p.count)), 0, stream, a, p, seql_items_for(p.count), seql_items_per_wave(p.count));
        constexpr int GS = 4, IN_RING = 128, OUT_RING = 256;
        // (a tile of few items: every item to the record executor, a wavefront each -- the rings give an item four lanes, and an item of long sequences is then a
        // serial chain of 5.4 ms whatever else the chip does: 64 frames 15.2 ms a call, 6.8 of it the sequence stage's own chain, 5.4 this one)
        const int32_t execMode = g_zstd_pipe_exec == 2 && p.count <= ZSTD_EXEC_ALL_RECORDS_MAX_ITEMS ? 1 : (int32_t)g_zstd_pipe_exec;
        if (execMode != 0) {
            hipLaunchKernelGGL(zstd_pipe_execute2_kernel<>, dim3((unsigned)p.count), dim3(64), 0, stream, a, p, execMode);
        }
        if (execMode != 1) {
            hipLaunchKernelGGL((zstd_pipe_execute_kernel<GS, IN_RING, OUT_RING>), dim3((unsigned)((p.count + 256 / GS - 1) / (256 / GS))), dim3(256), (size_t)(256 / GS) * (IN_RING + OUT_RING + a.ringPad), stream, a, p, execMode);
        }
        hipLaunchKernelGGL(zstd_pipe_checksum_kernel, dim3(w16), dim3(64), 0, stream, a, p);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (mbOn) {
        e = launch_zstd_mb_stages(a, stream, p, dflt, mbp);
        if (e != hipSuccess) return e;
    }
    return launch_zstd_decompress_list(a, stream, generalScratch, p.fallback, p.fallbackCount);
}

// ================================================================================================================================
// ONE LONG STREAM, A STEP AT A TIME (what ZstdInputStream does over ZstdIncrementalFrameDecompressor.java:44-72,216-234).
// The batched calls above want whole frames resident, input and output; a frame of gigabytes (ZstdOutputStream writes ONE frame however long
// the stream) then needs gigabytes.  Here the host (achip_abi.cpp: achip_zstdstream_decompress_*) cuts a frame into STEPS of whole blocks and
// the multi-block stages run on a step as if it were a frame -- the host puts a six-byte frame header in front of the step's blocks and
// marks its last block "last" -- with what a block may inherit from the blocks before it carried from step to step in a device-side record:
//   tables     a ghost slot 0 in front of the step's block slots receives the carried Huffman table and the three FSE tables; blocks
//              whose links the walk left open (treeless literals / repeat mode with no definition inside the step) are linked to it
//   history    the step's output goes behind the frame's last `window` bytes in one buffer (positions count from the oldest byte kept:
//              a match that reaches further back than the window is malformed, as in the Java decoder), the executor's LDS window is
//              filled from there (exec_records' warm start), the repeat offsets continue
//   checksum   a running XXH64 (the four accumulators, the length, the pending stripe), closed and compared at the frame's last block
// A block that any stage finds irregular ends the step IN FRONT of it: the bytes of the blocks before it are delivered, the stream fails at
// the read that reaches the block -- where ZstdInputStream throws.
// ================================================================================================================================
struct ZstdStreamCarry {
    int32_t rep[3];        // the repeat-offset history (reset to 1, 4, 8 at a frame's start by the host)
    int32_t hufLog;        // the carried Huffman table's log; < 0: none
    int32_t fseLog[3];     // ... the three FSE tables' logs; < 0: none
    int32_t goodBlocks;    // (out) blocks of the last step that decoded
    int32_t produced;      // (out) their bytes
    int32_t checksumOk;    // (out) the frame's checksum matched (only written by a step that closes a frame with one)
    int32_t tailLen;       // XXH64: bytes of the pending stripe
    int32_t pad;
    uint64_t total;        // XXH64: bytes hashed so far
    uint64_t v[4];         // XXH64: the accumulators
    uint8_t tail[32];
    uint16_t huf[zp::HUF_SLOT];
    uint16_t fse[zp::FSE_SLOT];
};
int64_t zstd_stream_carry_bytes() { return (int64_t)sizeof(ZstdStreamCarry); }
// a frame's first carry (host memory, zstd_stream_carry_bytes() of it): ZstdFrameDecompressor.reset() :199-203, no tables, a fresh XxHash64 (seed 0)
void zstd_stream_carry_init(void* hostCarry)
{
    ZstdStreamCarry* c = (ZstdStreamCarry*)hostCarry;
    memset((void*)c, 0, sizeof(*c));
    c->rep[0] = 1;
    c->rep[1] = 4;
    c->rep[2] = 8;
    c->hufLog = -1;
    c->fseLog[0] = c->fseLog[1] = c->fseLog[2] = -1;
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL;
    c->v[0] = P1 + P2;
    c->v[1] = P2;
    c->v[2] = 0;
    c->v[3] = 0 - P1;
}

// the ghost slot's tables from the carry, and the open links of the step's blocks to the ghost slot
__global__ __launch_bounds__(64) void zstd_ss_ghost_kernel(zp::Pipe p, const ZstdStreamCarry* carry)
{
    using namespace zp;
    const int lane = threadIdx.x;
    for (int32_t i = lane * 8; i < HUF_SLOT; i += 64 * 8) {
        *(u32x4*)(p.huf + i) = *(const u32x4*)(carry->huf + i);
    }
    for (int32_t i = lane * 8; i < FSE_SLOT; i += 64 * 8) {
        *(u32x4*)(p.fse + i) = *(const u32x4*)(carry->fse + i);
    }
    if (lane == 0) {
        Desc d = Desc();
        d.hufLog = carry->hufLog > 0 ? carry->hufLog : 0;
        d.log[0] = carry->fseLog[0] > 0 ? carry->fseLog[0] : 0;
        d.log[1] = carry->fseLog[1] > 0 ? carry->fseLog[1] : 0;
        d.log[2] = carry->fseLog[2] > 0 ? carry->fseLog[2] : 0;
        p.desc[0] = d;
    }
    // every block its own item record (a copy of the step's): a block that a stage finds irregular leaves the fast path alone -- the blocks in
    // front of it are still to be delivered
    const MbItem whole = p.mbItem[0];
    for (int32_t slot = 1 + lane; slot < p.count; slot += 64) {
        MbBlock& b = p.mb[slot];
        p.mbItem[slot] = whole;
        b.itemSlot = slot;
        if (b.kind != 2) {
            continue;
        }
        if (b.hufSlot < 0 && carry->hufLog >= 0) {
            b.hufSlot = 0;
        }
        for (int k = 0; k < 3; k++) {
            if (b.fseSlot[k] < 0 && carry->fseLog[k] >= 0) {
                b.fseSlot[k] = 0;
            }
        }
    }
}

// K4 for a step: the step's blocks in order behind the history (positions count from the oldest byte kept)
template <int WIN>
__global__ __launch_bounds__(64) void zstd_ss_execute_kernel(BatchArgs a, zp::Pipe p, ZstdStreamCarry* carry, int32_t startPos)
{
    using namespace zp;
    __shared__ __attribute__((aligned(16))) uint8_t win[WIN + 16];
    const int lane = threadIdx.x;
    const MbItem it = p.mbItem[0];
    const uint8_t* src = a.srcBase + a.srcOff[0];
    uint8_t* out = a.dstBase + a.dstOff[0];
    const int32_t outLimit = a.dstCap[0];
    int32_t rep0 = carry->rep[0], rep1 = carry->rep[1], rep2 = carry->rep[2];
    int32_t output = startPos;
    int32_t good = 0;
    bool bad = it.state != 1;
    for (int32_t i = 0; i < it.nBlocks && !bad; i++) {  // (uniform)
        const int32_t slot = 1 + i;
        const Desc d = p.desc[slot];
        const MbBlock b = p.mb[slot];
        if (d.state != 1 || p.mbItem[slot].state != 1) {
            break;
        }
        const uint8_t* lit = d.litMode == 0 ? src + d.litSrc : p.lit + (size_t)d.litBase * 64;
        sx2::RecordSource S{p.seq + d.seqBase, d.nDecoded, rep0, rep1, rep2};
        const int32_t before = output;
        output = sx2::exec_records<WIN>(win, S, lit, d.litSize, out, outLimit, lane, bad, output, i == 0);
        if (bad) {
            output = before;  // (what the block wrote in front of its damage is not delivered)
            break;
        }
        const int32_t n0 = sx2::rep_resolve(b.repOut[0], rep0, rep1, rep2), n1 = sx2::rep_resolve(b.repOut[1], rep0, rep1, rep2), n2 = sx2::rep_resolve(b.repOut[2], rep0, rep1, rep2);
        rep0 = n0;
        rep1 = n1;
        rep2 = n2;
        good = i + 1;
        wave_sync();
    }
    if (lane == 0) {
        carry->rep[0] = rep0;
        carry->rep[1] = rep1;
        carry->rep[2] = rep2;
        carry->goodBlocks = good;
        carry->produced = output - startPos;
    }
}

// the tables the blocks after this step may inherit: for each kind the last good block that defined its own (the walk's links say who)
__global__ __launch_bounds__(64) void zstd_ss_carry_kernel(zp::Pipe p, ZstdStreamCarry* carry)
{
    using namespace zp;
    const int lane = threadIdx.x;
    const int32_t good = carry->goodBlocks;
    int32_t lastHuf = 0, lastFse[3] = {0, 0, 0};  // (0: the ghost slot -- nothing new)
    for (int32_t slot = 1; slot <= good; slot++) {  // (uniform; the links are a few words per block)
        const MbBlock b = p.mb[slot];
        if (b.kind != 2) {
            continue;
        }
        lastHuf = b.hufSlot == slot ? slot : lastHuf;
        for (int k = 0; k < 3; k++) {
            lastFse[k] = b.fseSlot[k] == slot ? slot : lastFse[k];
        }
    }
    if (lastHuf > 0) {
        for (int32_t i = lane * 8; i < HUF_SLOT; i += 64 * 8) {
            *(u32x4*)(carry->huf + i) = *(const u32x4*)(p.huf + (size_t)lastHuf * HUF_SLOT + i);
        }
        if (lane == 0) {
            carry->hufLog = p.desc[lastHuf].hufLog;
        }
    }
    constexpr int32_t first[4] = {FSE_LL, FSE_OF, FSE_ML, FSE_SLOT};
    for (int k = 0; k < 3; k++) {
        if (lastFse[k] > 0) {
            const int32_t begin = k == 0 ? FSE_LL : (k == 1 ? FSE_OF : FSE_ML), end = k == 0 ? FSE_OF : (k == 1 ? FSE_ML : FSE_SLOT);
            for (int32_t i = begin + lane * 8; i < end; i += 64 * 8) {
                *(u32x4*)(carry->fse + i) = *(const u32x4*)(p.fse + (size_t)lastFse[k] * FSE_SLOT + i);
            }
            if (lane == 0) {
                carry->fseLog[k] = p.desc[lastFse[k]].log[k];
            }
        }
    }
    (void)first;
}

// the running XXH64 over the step's output (XxHash64.java:182-291 cut at stripe boundaries); closing: the frame's checksum word against the low 32 bits
__global__ __launch_bounds__(64) void zstd_ss_checksum_kernel(const uint8_t* __restrict__ data, ZstdStreamCarry* carry, int32_t closing, uint32_t expected)
{
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto mix = [&](uint64_t cur, uint64_t v) { return rotl(cur + v * P2, 31) * P1; };
    const int s = threadIdx.x;  // accumulator of this lane (the first four lanes work)
    if (s >= 4) {
        return;
    }
    const int32_t n = carry->produced;
    int32_t tailLen = carry->tailLen;
    uint64_t v = carry->v[s];
    int32_t at = 0;
    if (tailLen > 0) {  // complete the pending stripe first
        const int32_t take = 32 - tailLen < n ? 32 - tailLen : n;
        for (int32_t i = s; i < take; i += 4) {
            carry->tail[tailLen + i] = data[i];
        }
        wave_sync();
        tailLen += take;
        at = take;
        if (tailLen == 32) {
            v = mix(v, ld8(carry->tail + 8 * s));
            tailLen = 0;
        }
    }
    const int32_t stripes = (n - at) >> 5;
    const uint8_t* q = data + at + 8 * s;
    for (int32_t k = 0; k < stripes; k++) {
        v = mix(v, ld8(q + (int64_t)k * 32));
    }
    at += stripes * 32;
    wave_sync();
    if (at < n) {  // (the pending stripe was consumed or there was none: the rest starts a new one)
        for (int32_t i = s; i < n - at; i += 4) {
            carry->tail[i] = data[at + i];
        }
        tailLen = n - at;
    }
    wave_sync();
    const uint64_t total = carry->total + (uint64_t)(uint32_t)n;
    carry->v[s] = v;
    if (s == 0) {
        carry->tailLen = tailLen;
        carry->total = total;
    }
    if (closing) {
        uint64_t hash;
        if (total >= 32) {
            const uint64_t v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
            hash = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
            hash = (hash ^ mix(0, v1)) * P1 + P4;
            hash = (hash ^ mix(0, v2)) * P1 + P4;
            hash = (hash ^ mix(0, v3)) * P1 + P4;
            hash = (hash ^ mix(0, v4)) * P1 + P4;
        }
        else {
            hash = P5;  // (seed 0)
        }
        hash += total;
        const uint8_t* t = carry->tail;
        int32_t index = 0;
        while (index <= tailLen - 8) {
            hash = rotl(hash ^ mix(0, ld8(t + index)), 27) * P1 + P4;
            index += 8;
        }
        if (index <= tailLen - 4) {
            hash = rotl(hash ^ ((uint64_t)ld4(t + index) * P1), 23) * P2 + P3;
            index += 4;
        }
        while (index < tailLen) {
            hash = rotl(hash ^ ((uint64_t)t[index] * P5), 11) * P1;
            index++;
        }
        hash ^= hash >> 33;
        hash *= P2;
        hash ^= hash >> 29;
        hash *= P3;
        hash ^= hash >> 32;
        if (s == 0) {
            carry->checksumOk = (uint32_t)hash == expected ? 1 : 0;
        }
    }
}

namespace {
struct StepLayout {
    int64_t counters, fallback, mbList, mbItem, args, mb, desc, huf, fse, lit, seq, order, general, total;
    int32_t slots;
    uint32_t litCap, seqCap;
};
StepLayout step_layout(int32_t slots, uint32_t litUnits, uint32_t seqs)
{
    StepLayout L;
    L.slots = slots < 64 ? 64 : slots;
    L.litCap = litUnits + 64;
    L.seqCap = seqs + 64;
    auto up = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    int64_t o = 0;
    L.counters = o; o = up(o + 512);           // [0, 256) the pipeline's counters (fallbackCount, mbCount at + 40 words); [256, 512) the pass's cursors
    L.fallback = o; o = up(o + (int64_t)L.slots * 4 + 64);
    L.mbList = o; o = up(o + 64);
    L.mbItem = o; o = up(o + (int64_t)L.slots * sizeof(zp::MbItem));
    L.args = o; o = up(o + 64);                // the one item's srcOff, dstOff, errOffset (8 bytes each), srcLen, dstCap, outLen, status
    L.mb = o; o = up(o + (int64_t)L.slots * sizeof(zp::MbBlock));
    L.desc = o; o = up(o + (int64_t)L.slots * sizeof(zp::Desc));
    L.huf = o; o = up(o + (int64_t)L.slots * zp::HUF_SLOT * 2);
    L.fse = o; o = up(o + (int64_t)L.slots * zp::FSE_SLOT * 2);
    L.lit = o; o = up(o + (int64_t)L.litCap * 64 + 4096);
    L.seq = o; o = up(o + (int64_t)L.seqCap * 8 + 4096);
    L.order = o; o = up(o + 2 * zp::ORDER_BUCKETS * 4 + (int64_t)L.slots * 4);
    L.general = o; o = up(o + 4096 + (int64_t)sizeof(zd::FseTable) * 3);  // (the predefined tables only: launch_zstd_decompress_prepare)
    L.total = o;
    return L;
}
}  // namespace

// What a step of `blocks` blocks needs at most: slots for the blocks and the ghost, a literal arena and a sequence arena for blocks of the maximum size.
int64_t zstd_stream_step_scratch_bytes(int32_t blocks)
{
    return step_layout(blocks + 1, (uint32_t)blocks * (uint32_t)((zp::LIT_STRIDE + 63) / 64 + 1), (uint32_t)blocks * 43691u + 64u).total;
}

// One step.  dSrc: the six-byte header + the step's blocks (the last marked last), srcLen bytes; dOut: where position 0 -- the oldest history byte
// kept -- lies; the step's output starts at startPos and may reach outLimit; closing: the step ends the frame, with a checksum word to verify if
// hasChecksum.  result[0..2] (host): blocks decoded, bytes produced, checksum verdict (1 ok, 0 mismatch, -1 none checked).  Synchronous.
hipError_t launch_zstd_stream_step(hipStream_t stream, void* scratch, int64_t scratchBytes, void* carryDev, const uint8_t* dSrc, int32_t srcLen, int32_t blocks, uint8_t* dOut,
                                   int32_t startPos, int32_t outLimit, int32_t closing, int32_t hasChecksum, uint32_t expected, int32_t* result)
{
    ZstdStreamCarry* carry = (ZstdStreamCarry*)carryDev;
    const StepLayout L = step_layout(blocks + 1, (uint32_t)blocks * (uint32_t)((zp::LIT_STRIDE + 63) / 64 + 1), (uint32_t)blocks * 43691u + 64u);
    if (L.total > scratchBytes) {
        return hipErrorUnknown;  // (the caller sized the scratch with zstd_stream_step_scratch_bytes for fewer blocks)
    }
    uint8_t* base = (uint8_t*)scratch;
    const zd::FseTable* dflt = nullptr;
    hipError_t e = launch_zstd_decompress_prepare(stream, base + L.general, &dflt);
    if (e != hipSuccess) return e;
    // the one item
    struct {
        int64_t srcOff, dstOff, err;
        int32_t srcLen, dstCap, outLen, status;
    } args = {0, 0, 0, srcLen, outLimit, 0, 0};
    int32_t one[2] = {0, 0};
    e = hipMemsetAsync(base + L.counters, 0, 512, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(base + L.args, &args, sizeof(args), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(base + L.mbList, one, 4, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    one[0] = 1;
    e = hipMemcpyAsync(base + L.counters + 160, one, 4, hipMemcpyHostToDevice, stream);  // mbCount[0] = 1 listed item
    if (e != hipSuccess) return e;
    BatchArgs a = BatchArgs();
    a.srcBase = dSrc;
    a.srcOff = (const int64_t*)(base + L.args);
    a.dstOff = (const int64_t*)(base + L.args + 8);
    a.errOffset = (int64_t*)(base + L.args + 16);
    a.srcLen = (const int32_t*)(base + L.args + 24);
    a.dstCap = (const int32_t*)(base + L.args + 28);
    a.outLen = (int32_t*)(base + L.args + 32);
    a.status = (int32_t*)(base + L.args + 36);
    a.dstBase = dOut;
    a.nBlocks = 1;
    zp::Pipe p = zp::Pipe();
    p.fallbackCount = (int32_t*)(base + L.counters);
    p.fallback = (int32_t*)(base + L.fallback);
    p.mbList = (int32_t*)(base + L.mbList);
    p.mbCount = (int32_t*)(base + L.counters) + 40;
    p.mbItem = (zp::MbItem*)(base + L.mbItem);
    p.mb = (zp::MbBlock*)(base + L.mb);
    p.desc = (zp::Desc*)(base + L.desc);
    p.huf = (uint16_t*)(base + L.huf);
    p.fse = (uint16_t*)(base + L.fse);
    p.lit = base + L.lit;
    p.seq = (uint64_t*)(base + L.seq);
    p.seqCap = L.seqCap;
    p.litCap = L.litCap;
    p.seqCursor = (uint32_t*)(base + L.counters + 256);
    p.litCursor = (uint32_t*)(base + L.counters + 260);
    p.first = 0;
    p.count = blocks + 1;
    p.passFirst = -1;  // (block i of the item sits in slot 1 + i: slot 0 is the ghost)
    p.itemFirst = 0;
    p.itemEnd = 1;
    p.mbSlots = L.slots;
    p.mbLitCap = L.litCap;
    p.mbSeqCap = L.seqCap;
    p.order = nullptr;
    p.orderHist = nullptr;
    hipLaunchKernelGGL(zstd_mb_count_kernel, dim3(1), dim3(64), 0, stream, a, p);
    hipLaunchKernelGGL(zstd_mb_scan_kernel, dim3(1), dim3(64), 0, stream, p);
    e = hipMemsetAsync(p.mb, 0xFF, (size_t)p.count * sizeof(zp::MbBlock), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(zstd_mb_fill_kernel, dim3(1), dim3(64), 0, stream, a, p);
    hipLaunchKernelGGL(zstd_ss_ghost_kernel, dim3(1), dim3(64), 0, stream, p, carry);
    hipLaunchKernelGGL(zstd_mb_parse_kernel, dim3((unsigned)p.count), dim3(64), 0, stream, a, p, dflt);
    launch_literals<true>(a, p, stream);
    hipLaunchKernelGGL(zstd_pipe_sequences_lane_kernel<true>, dim3((unsigned)((p.count + seql_items_for(p.count) - 1) / seql_items_for(p.count))), dim3(64 * seql_waves_for(p.count)), 0, stream, a, p, seql_items_for(p.count), seql_items_per_wave(p.count));
    hipLaunchKernelGGL(zstd_ss_execute_kernel<32768>, dim3(1), dim3(64), 0, stream, a, p, carry, startPos);
    hipLaunchKernelGGL(zstd_ss_carry_kernel, dim3(1), dim3(64), 0, stream, p, carry);
    hipLaunchKernelGGL(zstd_ss_checksum_kernel, dim3(1), dim3(64), 0, stream, dOut + startPos, carry, closing != 0 && hasChecksum != 0 ? 1 : 0, expected);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    struct {
        int32_t goodBlocks, produced, checksumOk;
    } r = {0, 0, 0};
    e = hipMemcpyAsync(&r, &carry->goodBlocks, sizeof(r), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    result[0] = r.goodBlocks;
    result[1] = r.produced;
    result[2] = (closing != 0 && hasChecksum != 0 && r.goodBlocks == blocks) ? r.checksumOk : -1;
    return hipSuccess;
}

}  // namespace achip
