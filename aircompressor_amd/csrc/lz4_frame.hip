// lz4_frame.hip -- batched LZ4 frame container decode for gfx950.
//
// Replaces Lz4FrameCompression.decompress / decompressFrame / skipFrame (M/lz4/Lz4FrameCompression.java:145-343) over the
// HIP block decoder.  One wavefront per item (a buffer of concatenated frames), persistent grid: the frame walk is a
// serial chain (every block's output position and remaining capacity depend on the blocks before it), so the wavefront
// runs the Java loop itself -- same checks, same order, so status / detail / offset equal the Java exception -- and uses
// its 64 lanes inside each step: the block decoder of lz4_decode_body.h over LDS rings (64 lanes per block), 64 x 16-byte
// copies for stored blocks, XXH32 (achip_xxhash.h) for the header / block / content checksums.  Throughput comes from
// many frames per batch; a 4 MiB block is one wavefront's work.
#include "lz4_decode_body.h"
#include "achip_xxhash.h"

namespace achip {

namespace lz4f {
constexpr uint32_t MAGIC = 0x184D2204u, SKIPPABLE_MAGIC = 0x184D2A50u, SKIPPABLE_MASK = 0xFFFFFFF0u;
constexpr int FLG_BLOCK_INDEPENDENCE = 1 << 5, FLG_BLOCK_CHECKSUM = 1 << 4, FLG_CONTENT_SIZE = 1 << 3, FLG_CONTENT_CHECKSUM = 1 << 2, FLG_DICTIONARY_ID = 1;
constexpr int FLG_RESERVED_MASK = 0x02, BD_RESERVED_MASK = 0x8F;
constexpr int HEADER_SIZE = 7;
constexpr uint32_t UNCOMPRESSED_FLAG = 0x80000000u;
constexpr int IN_RING = 2048, OUT_RING = 4096;
using FR = Rings<64, IN_RING, OUT_RING, 1>;

// XXH32 (seed 0) of fewer than 16 bytes: the frame descriptor (XxHash32JavaHasher.java:92-110)
__device__ __forceinline__ uint32_t xxh32_short(const uint8_t* p, int32_t len)
{
    constexpr uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
    auto rotl = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
    uint32_t h = P5 + (uint32_t)len;
    int32_t i = 0;
    for (; i + 4 <= len; i += 4) {
        h = rotl(h + ld4(p + i) * P3, 17) * P4;
    }
    for (; i < len; i++) {
        h = rotl(h + (uint32_t)p[i] * P5, 11) * P1;
    }
    h ^= h >> 15;
    h *= P2;
    h ^= h >> 13;
    h *= P3;
    h ^= h >> 16;
    return h;
}

#define LZ4F_FAIL(detail, off)                            \
    {                                                     \
        eo = (int64_t)(off);                              \
        return mk_status(ACHIP_CLASS_MALFORMED, detail);  \
    }

// decompressFrame :184-322 ; wave-uniform.  Returns 0 or the status; pos / op advance.
__device__ int32_t decompress_frame(const uint8_t* __restrict__ in, int32_t inLen, uint8_t* out, int32_t outCap, uint8_t* lds, int lane, int32_t& pos, int32_t& op, int64_t& eo)
{
    const int32_t frameStart = pos;
    const int32_t outStart = op;
    const int32_t dstart = frameStart + 4;
    if (dstart + 2 > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_TRUNC_HEADER, dstart);
    const int flg = in[dstart], bd = in[dstart + 1];
    const int version = (flg >> 6) & 3;
    if (version != 1) LZ4F_FAIL(version == 0 ? ACHIP_D_LZ4F_VERSION_0 : (version == 2 ? ACHIP_D_LZ4F_VERSION_2 : ACHIP_D_LZ4F_VERSION_3), dstart);
    if ((flg & FLG_RESERVED_MASK) != 0 || (bd & BD_RESERVED_MASK) != 0) LZ4F_FAIL(ACHIP_D_LZ4F_RESERVED_BITS, dstart);
    const bool blockChecksum = (flg & FLG_BLOCK_CHECKSUM) != 0, contentSize = (flg & FLG_CONTENT_SIZE) != 0, contentChecksum = (flg & FLG_CONTENT_CHECKSUM) != 0;
    if ((flg & FLG_BLOCK_INDEPENDENCE) == 0) LZ4F_FAIL(ACHIP_D_LZ4F_LINKED_BLOCKS, dstart);
    if ((flg & FLG_DICTIONARY_ID) != 0) LZ4F_FAIL(ACHIP_D_LZ4F_DICTIONARY, dstart);
    const int sizeId = (bd >> 4) & 7;
    if (sizeId < 4) LZ4F_FAIL(ACHIP_D_LZ4F_BLOCK_MAX_SIZE, dstart + 1);
    const int32_t blockMax = 1 << (8 + 2 * sizeId);  // 64 KiB, 256 KiB, 1 MiB, 4 MiB (Lz4FrameFormat.java:58-67)
    int32_t p = dstart + 2;
    if ((int64_t)p + (contentSize ? 8 : 0) + 1 > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_TRUNC_HEADER, p);
    int64_t expectedSize = -1;
    if (contentSize) {
        expectedSize = (int64_t)ld8(in + p);
        p += 8;
    }
    const int expectedHc = in[p];
    const int actualHc = (int)((xxh32_short(in + dstart, p - dstart) >> 8) & 0xFF);
    if (expectedHc != actualHc) LZ4F_FAIL(ACHIP_D_LZ4F_HEADER_CHECKSUM, p);
    p++;

    int32_t o = outStart;
    for (;;) {
        if ((int64_t)p + 4 > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_MISSING_BLOCK_SIZE, p);
        const uint32_t header = ld4(in + p);
        p += 4;
        if (header == 0) {
            break;
        }
        const bool stored = (header & UNCOMPRESSED_FLAG) != 0;
        const int64_t blockLen = header & 0x7FFFFFFFu;
        if (blockLen > blockMax || (int64_t)p + blockLen > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_BLOCK_PAST_END, p);
        if (stored) {
            if ((int64_t)o + blockLen > outCap) LZ4F_FAIL(ACHIP_D_LZ4F_OUTPUT_TOO_SMALL, o);
            wave_mem_order();
            group_copy<64>(out + o, in + p, (int32_t)blockLen, lane);
            wave_mem_order();
            o += (int32_t)blockLen;
        }
        else {
            FR R;
            R.init(lds, lds + IN_RING, in + p, (int32_t)blockLen, out + o, lane);
            int32_t bst = 0, beo = 0, bop = 0;
            lz4_block_decode<64, IN_RING, OUT_RING, 1>(R, in + p, (int32_t)blockLen, outCap - o, bst, beo, bop);
            wave_mem_order();
            if (bst != 0) {
                if (bst == mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT)) {
                    LZ4F_FAIL(ACHIP_D_LZ4F_OUTPUT_TOO_SMALL, o);  // the block codec returned -1 (:275-277)
                }
                eo = (int64_t)beo;  // the block codec's exception propagates; its offset is relative to the block
                return bst;
            }
            if (bop > blockMax) LZ4F_FAIL(ACHIP_D_LZ4F_BLOCK_EXCEEDS_MAX, p);
            o += bop;
        }
        if (blockChecksum) {
            const int64_t cpos = (int64_t)p + blockLen;
            if (cpos + 4 > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_MISSING_BLOCK_CHECKSUM, cpos);
            const uint32_t actual = quad_xxh32(in + p, (int32_t)blockLen, 0u, lane & 3, lane - (lane & 3));
            if (ld4(in + cpos) != actual) LZ4F_FAIL(ACHIP_D_LZ4F_BLOCK_CHECKSUM, cpos);
        }
        p += (int32_t)blockLen;
        if (blockChecksum) {
            p += 4;
        }
    }
    const int32_t contentLen = o - outStart;
    if (contentChecksum) {
        if ((int64_t)p + 4 > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_MISSING_CONTENT_CHECKSUM, p);
        const uint32_t actual = quad_xxh32(out + outStart, contentLen, 0u, lane & 3, lane - (lane & 3));
        if (ld4(in + p) != actual) LZ4F_FAIL(ACHIP_D_LZ4F_CONTENT_CHECKSUM, p);
        p += 4;
    }
    if (contentSize && (int64_t)contentLen != expectedSize) LZ4F_FAIL(ACHIP_D_LZ4F_CONTENT_SIZE, p);
    pos = p;
    op = o;
    return 0;
}

// decompress :145-177
__device__ int32_t decompress_item(const uint8_t* __restrict__ in, int32_t inLen, uint8_t* out, int32_t outCap, uint8_t* lds, int lane, int32_t& opOut, int64_t& eo)
{
    eo = 0;
    opOut = 0;
    if (inLen < HEADER_SIZE) LZ4F_FAIL(ACHIP_D_LZ4F_TOO_SHORT, 0);
    int32_t pos = 0, op = 0;
    while (pos < inLen) {
        if ((int64_t)pos + 4 > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_TRUNC_MAGIC, pos);
        const uint32_t magic = ld4(in + pos);
        if (magic == MAGIC) {
            const int32_t r = decompress_frame(in, inLen, out, outCap, lds, lane, pos, op, eo);
            if (r != 0) {
                return r;
            }
        }
        else if ((magic & SKIPPABLE_MASK) == SKIPPABLE_MAGIC) {  // skipFrame :327-343
            const int64_t spos = (int64_t)pos + 4;
            if (spos + 4 > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_TRUNC_SKIP_SIZE, spos);
            const int64_t frameEnd = spos + 4 + (int64_t)ld4(in + spos);
            if (frameEnd > inLen) LZ4F_FAIL(ACHIP_D_LZ4F_TRUNC_SKIP, spos);
            pos = (int32_t)frameEnd;
        }
        else {
            LZ4F_FAIL(ACHIP_D_LZ4F_BAD_MAGIC, pos);
        }
    }
    opOut = op;
    return 0;
}
#undef LZ4F_FAIL

}  // namespace lz4f

// ---- reader variant 1 (round 2; written without a GPU at hand, not the default until measured): the blocks of the regular frames as ONE
// batch for the two-pass block decoder, as hadoop_streams.hip does it for Hadoop streams.
//   walk   a lane per item: the item is exactly one frame, its header is in order, no block checksums; every compressed block becomes an
//          entry of a device-side batch, block k at output position k x blockMaxSize with at most a block's room.  That position is a GUESS
//          -- a block's length is not in the frame -- which holds for what every writer produces (all blocks but the last are full);
//   decode the batch through the two-pass decoder; its record arena is sized after the host has read back the number of blocks and their room;
//   fold   a wavefront per item walks the frame again: compressed blocks must have decoded, all but the last to a full block; stored blocks
//          are copied now; content checksum and content size are checked; anything else flags the item;
//   then the kernel above runs the flagged items, and every item of any other shape, from scratch (the Java loop: same status / offset).
namespace lz4f {
constexpr int32_t MAX_LIST_BLOCKS = 1 << 20;

struct BlockList {
    int32_t* counters;   // [0] entries allocated, [1] entries in use (sealed), [2..3] the entries' room (64 bits)
    int32_t* sFirst;     // per item: first entry, -1 = not on this path
    int32_t* sSerial;    // per item: 1 = the wavefront-per-item kernel decodes it
    int64_t* cSrcOff;
    int64_t* cDstOff;
    int64_t* cErrOff;
    int32_t* cSrcLen;
    int32_t* cDstCap;
    int32_t* cOutLen;
    int32_t* cStatus;
};

struct FrameShape {
    bool ok;
    int32_t firstBlock;      // position of the first block header
    int32_t blockMax;
    int32_t compressedBlocks;
    bool contentChecksum;
    int64_t expectedSize;    // -1: not announced
};

// the header of an item that holds exactly one frame of the shape the list path takes (decompressFrame :184-240 without its verdicts)
__device__ __forceinline__ FrameShape frame_shape(const uint8_t* __restrict__ in, int32_t inLen)
{
    FrameShape f;
    f.ok = false;
    f.firstBlock = 0;
    f.blockMax = 0;
    f.compressedBlocks = 0;
    f.contentChecksum = false;
    f.expectedSize = -1;
    if (inLen < HEADER_SIZE + 4 || ld4(in) != MAGIC) {
        return f;
    }
    const int flg = in[4], bd = in[5];
    if (((flg >> 6) & 3) != 1 || (flg & FLG_RESERVED_MASK) != 0 || (bd & BD_RESERVED_MASK) != 0 || (flg & FLG_BLOCK_INDEPENDENCE) == 0 || (flg & FLG_DICTIONARY_ID) != 0 ||
        (flg & FLG_BLOCK_CHECKSUM) != 0) {
        return f;
    }
    const int sizeId = (bd >> 4) & 7;
    if (sizeId < 4) {
        return f;
    }
    f.blockMax = 1 << (8 + 2 * sizeId);
    f.contentChecksum = (flg & FLG_CONTENT_CHECKSUM) != 0;
    int32_t p = 6;
    if ((flg & FLG_CONTENT_SIZE) != 0) {
        if (p + 8 + 1 > inLen) {
            return f;
        }
        f.expectedSize = (int64_t)ld8(in + p);
        p += 8;
    }
    if (p + 1 > inLen || in[p] != (int)((xxh32_short(in + 4, p - 4) >> 8) & 0xFF)) {
        return f;
    }
    f.firstBlock = p + 1;
    f.ok = true;
    return f;
}

// the blocks of such a frame: FILL = 0 counts the compressed ones and checks that the frame ends the item; FILL = 1 writes their entries
template <bool FILL>
__device__ __forceinline__ bool frame_blocks(const BatchArgs& a, int32_t item, FrameShape& f, const BlockList& L, int32_t firstEntry)
{
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[item];
    const int32_t inLen = a.srcLen[item];
    const int64_t outCap = a.dstCap[item];
    int64_t p = f.firstBlock;
    int64_t o = 0;
    int32_t n = 0;
    for (;;) {
        if (p + 4 > inLen) {
            return false;
        }
        const uint32_t header = ld4(in + p);
        p += 4;
        if (header == 0) {
            break;
        }
        const int64_t blockLen = header & 0x7FFFFFFFu;
        if (blockLen > f.blockMax || p + blockLen > inLen || o >= outCap) {
            return false;
        }
        if ((header & UNCOMPRESSED_FLAG) != 0) {
            if (o + blockLen > outCap) {
                return false;
            }
            // (a stored block shorter than a full one must be the last: fold sees to that through the positions)
            o += blockLen < f.blockMax ? blockLen : f.blockMax;
        }
        else {
            if (FILL) {
                const int32_t e = firstEntry + n;
                L.cSrcOff[e] = a.srcOff[item] + p;
                L.cSrcLen[e] = (int32_t)blockLen;
                L.cDstOff[e] = a.dstOff[item] + o;
                L.cDstCap[e] = (int32_t)(outCap - o < f.blockMax ? outCap - o : f.blockMax);
                L.cOutLen[e] = 0;
                L.cStatus[e] = -1;
                L.cErrOff[e] = 0;
            }
            n++;
            o += f.blockMax;  // the guess: this block is full (the last one may fall short: nothing follows it)
        }
        p += blockLen;
    }
    if (f.contentChecksum) {
        if (p + 4 > inLen) {
            return false;
        }
        p += 4;
    }
    f.compressedBlocks = n;
    return p == inLen;  // (frames or skippable frames behind it: the other kernel's)
}
}  // namespace lz4f

__global__ __launch_bounds__(64) void lz4frame_walk_kernel(BatchArgs a, lz4f::BlockList L)
{
    using namespace lz4f;
    const int32_t item = blockIdx.x * 64 + threadIdx.x;
    if (item >= a.nBlocks) {
        return;
    }
    L.sFirst[item] = -1;
    L.sSerial[item] = 1;
    FrameShape f = frame_shape(a.srcBase + a.srcOff[item], a.srcLen[item]);
    if (!f.ok || a.dstCap[item] <= 0 || !frame_blocks<false>(a, item, f, L, 0)) {
        return;
    }
    const int32_t first = atomicAdd(L.counters, f.compressedBlocks);
    if ((int64_t)first + f.compressedBlocks > MAX_LIST_BLOCKS) {
        return;
    }
    frame_blocks<true>(a, item, f, L, first);
    long long room = 0;
    for (int32_t k = 0; k < f.compressedBlocks; k++) {
        room += L.cDstCap[first + k];
    }
    atomicAdd((unsigned long long*)(L.counters + 2), (unsigned long long)room);
    L.sFirst[item] = first;
    L.sSerial[item] = 0;
}

__global__ void lz4frame_seal_kernel(lz4f::BlockList L)
{
    const int32_t allocated = L.counters[0];
    L.counters[1] = allocated < lz4f::MAX_LIST_BLOCKS ? allocated : lz4f::MAX_LIST_BLOCKS;
}

// a wavefront per item on the list path
__global__ __launch_bounds__(64) void lz4frame_fold_kernel(BatchArgs a, lz4f::BlockList L, int32_t decoded)
{
    using namespace lz4f;
    const int32_t item = blockIdx.x;
    if (L.sSerial[item] != 0) {
        return;
    }
    const int lane = threadIdx.x;
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[item];
    uint8_t* out = a.dstBase + a.dstOff[item];
    const FrameShape f = frame_shape(in, a.srcLen[item]);  // (the walk's findings again: wave-uniform)
    bool ok = decoded != 0;
    int64_t p = f.firstBlock, o = 0;
    int32_t e = L.sFirst[item];
    bool shortSeen = false;  // a block that fell short of a full one: nothing may follow it
    while (ok) {  // (uniform)
        const uint32_t header = ld4(in + p);
        p += 4;
        if (header == 0) {
            break;
        }
        const int64_t blockLen = header & 0x7FFFFFFFu;
        ok = !shortSeen;
        if (!ok) {
            break;
        }
        if ((header & UNCOMPRESSED_FLAG) != 0) {
            wave_sync();
            group_copy<64>(out + o, in + p, (int32_t)blockLen, lane);
            wave_sync();
            shortSeen = blockLen < f.blockMax;
            o += blockLen;
        }
        else {
            const int32_t got = L.cOutLen[e];
            ok = L.cStatus[e] == 0;
            shortSeen = got < f.blockMax;
            o += got;
            e++;
        }
        p += blockLen;
    }
    if (ok && f.contentChecksum) {
        wave_sync();
        const uint32_t actual = quad_xxh32(out, (int32_t)o, 0u, lane & 3, lane - (lane & 3));
        ok = ld4(in + p) == actual;
    }
    ok = ok && (f.expectedSize < 0 || o == f.expectedSize);
    if (lane == 0) {
        if (ok) {
            a.outLen[item] = (int32_t)o;
            a.status[item] = 0;
            a.errOffset[item] = 0;
        }
        else {
            L.sSerial[item] = 1;  // the Java loop decides what it means
        }
    }
}

// LISTED: only the items `list` flags (the list path's leftovers)
template <bool LISTED>
__global__ __launch_bounds__(64) void lz4frame_decompress_kernel(BatchArgs a, int32_t* nextItem, const int32_t* __restrict__ list)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[lz4f::IN_RING + lz4f::OUT_RING];
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    for (;;) {
        __syncthreads();
        if (lane == 0) {
            item = atomicAdd(nextItem, 1);
        }
        __syncthreads();
        const int32_t block = item;
        if (block >= a.nBlocks) {
            return;
        }
        if (LISTED && list[block] == 0) {
            continue;
        }
        int32_t op = 0;
        int64_t eo = 0;
        const int32_t st = lz4f::decompress_item(a.srcBase + a.srcOff[block], a.srcLen[block], a.dstBase + a.dstOff[block], a.dstCap[block], lds, lane, op, eo);
        if (lane == 0) {
            a.outLen[block] = st == 0 ? op : 0;
            a.status[block] = st;
            a.errOffset[block] = st == 0 ? 0 : eo;
        }
    }
}

hipError_t launch_lz4_sequence_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit);
hipError_t launch_lz4_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats);
hipError_t launch_lz4_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);
int lz4_ring_group_for(int32_t nBlocks);
int64_t twopass_scratch_bytes(int32_t nBlocks, int64_t perBlock);

int64_t lz4frame_decompress_scratch_bytes(int32_t nItems, int variant)
{
    if (variant != 1 && variant != 2) {
        return 4096;
    }
    const int64_t n = nItems < 1 ? 1 : nItems;
    return 4096 + n * 8 + 64 + (int64_t)lz4f::MAX_LIST_BLOCKS * (8 * 3 + 4 * 4) + 4096;
}

// variant 0 (default): a wavefront per item; variant 1: the block list (above)
hipError_t launch_lz4frame_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, int variant, const AuxScratch* aux)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* base = (uint8_t*)scratch;
    int32_t* counter = (int32_t*)base;
    hipError_t e = hipMemsetAsync(counter, 0, 4096, stream);
    if (e != hipSuccess) return e;
    const int32_t maxWaves = 256 * 16;
    const unsigned grid = (unsigned)(a.nBlocks < maxWaves ? a.nBlocks : maxWaves);
    if ((variant != 1 && variant != 2) || aux == nullptr || aux->get == nullptr) {
        hipLaunchKernelGGL(lz4frame_decompress_kernel<false>, dim3(grid), dim3(64), 0, stream, a, counter, (const int32_t*)nullptr);
        return hipGetLastError();
    }
    lz4f::BlockList L;
    uint8_t* p = base + 4096;
    auto take = [&](int64_t bytes) {
        uint8_t* r = p;
        p += (bytes + 15) & ~(int64_t)15;
        return r;
    };
    const int64_t n = a.nBlocks, C = lz4f::MAX_LIST_BLOCKS;
    L.counters = counter + 16;
    L.sFirst = (int32_t*)take(4 * n);
    L.sSerial = (int32_t*)take(4 * n);
    L.cSrcOff = (int64_t*)take(8 * C);
    L.cDstOff = (int64_t*)take(8 * C);
    L.cErrOff = (int64_t*)take(8 * C);
    L.cSrcLen = (int32_t*)take(4 * C);
    L.cDstCap = (int32_t*)take(4 * C);
    L.cOutLen = (int32_t*)take(4 * C);
    L.cStatus = (int32_t*)take(4 * C);
    hipLaunchKernelGGL(lz4frame_walk_kernel, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, stream, a, L);
    hipLaunchKernelGGL(lz4frame_seal_kernel, dim3(1), dim3(1), 0, stream, L);
    // variant 2 (the default since round 3): the sequence-length probe of the block API's auto mode runs on the listed blocks before the
    // synchronisation; long sequences (a lane walking them pays per sequence, the wavefront-per-item kernel moves 64 bytes per step: 75
    // against 22 GiB/s on the fragments frames) leave every item to the wavefront-per-item kernel, short ones (text: 7.8 -> 13.4 GiB/s) take the list
    int32_t* stats = counter + 24;
    if (variant == 2) {
        BatchArgs c = a;
        c.srcOff = L.cSrcOff;
        c.srcLen = L.cSrcLen;
        c.nBlocks = (int32_t)C;
        c.nBlocksDev = L.counters + 1;
        e = launch_lz4_sequence_sample(c, stream, stats, 0, 0);
        if (e != hipSuccess) return e;
    }
    // the number of listed blocks and their room decide the record arena: the one synchronisation of the call
    int32_t counts[12] = {0};
    e = hipMemcpyAsync(counts, L.counters, sizeof(counts), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    const int32_t nListed = counts[1];
    const bool isShort = counts[8 + 1] > 0 && (int64_t)counts[8 + 2] < 12LL * (int64_t)counts[8 + 1];
    int32_t decoded = 0;
    BatchArgs c = a;
    c.srcOff = L.cSrcOff;
    c.srcLen = L.cSrcLen;
    c.dstOff = L.cDstOff;
    c.dstCap = L.cDstCap;
    c.outLen = L.cOutLen;
    c.status = L.cStatus;
    c.errOffset = L.cErrOff;
    c.nBlocks = nListed;
    c.nBlocksDev = nullptr;
    c.only = nullptr;
    c.onlyStats = nullptr;
    if (nListed > 0 && (variant == 1 || isShort)) {
        long long room = 0;
        __builtin_memcpy(&room, counts + 2, 8);
        // records per block as for the block codec (96 KiB of arena per 64 KiB of output: lz4_decompress_v7.hip), by the blocks' room
        const int64_t perBlock = ((room + nListed - 1) / nListed * 3 / 2 + 4095) & ~4095LL;
        const int64_t bytes = twopass_scratch_bytes(nListed, perBlock < 98304 ? 98304 : perBlock);
        void* arena = aux->get(aux->user, bytes);
        if (arena != nullptr) {
            e = launch_lz4_decompress_twopass(c, stream, arena, bytes, 16, 0, 2, nullptr);
            if (e != hipSuccess) return e;
            decoded = 1;
        }
    }
    else if (nListed > 0) {
        // long sequences (round 5): the listed blocks through the ring decoders, the lanes per block by how many blocks there are (profiles/r05_groupsweep.txt: 16 384
        // blocks of 256 KiB 890 GiB/s at 16 lanes, 1 024 of 4 MiB 85 at 64) -- the wavefront-per-item kernel below, where these frames went until now, is a serial
        // reader that was never the fast one: 149 GiB/s on 16 384 frames of 256 KiB
        e = launch_lz4_decompress_rings(c, stream, lz4_ring_group_for(nListed), 0, nullptr);
        if (e != hipSuccess) return e;
        decoded = 1;
    }
    hipLaunchKernelGGL(lz4frame_fold_kernel, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, L, decoded);
    hipLaunchKernelGGL(lz4frame_decompress_kernel<true>, dim3(grid), dim3(64), 0, stream, a, counter, (const int32_t*)L.sSerial);
    return hipGetLastError();
}

}  // namespace achip
