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

__global__ __launch_bounds__(64) void lz4frame_decompress_kernel(BatchArgs a, int32_t* nextItem)
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

hipError_t launch_lz4frame_decompress(const BatchArgs& a, hipStream_t stream, void* scratch)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    int32_t* counter = (int32_t*)scratch;
    hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
    if (e != hipSuccess) return e;
    const int32_t maxWaves = 256 * 16;
    const unsigned grid = (unsigned)(a.nBlocks < maxWaves ? a.nBlocks : maxWaves);
    hipLaunchKernelGGL(lz4frame_decompress_kernel, dim3(grid), dim3(64), 0, stream, a, counter);
    return hipGetLastError();
}

}  // namespace achip
