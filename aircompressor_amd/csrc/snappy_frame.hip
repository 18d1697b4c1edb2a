// snappy_frame.hip -- batched x-snappy-framed stream decode / encode for gfx950.
//
// Replaces the read-to-the-end use of SnappyFramedInputStream (M/snappy/SnappyFramedInputStream.java:52-73 stream header,
// :135-214 ensureBuffer, :226-305 chunk headers) and the write-all-then-close use of SnappyFramedOutputStream
// (M/snappy/SnappyFramedOutputStream.java:73-96, :113-145, :200-255) over the HIP block codec.  An item of the batch is
// a whole stream.  One wavefront per item, persistent grid: the chunk walk is a serial chain (a chunk's position is known
// only after the one before it), so the wavefront runs the Java loop itself -- same checks in the same order -- and puts
// its 64 lanes into each step: the ring block decoder of snappy_decode_body.h (64 lanes per chunk), 64 x 16-byte copies
// for stored chunks, the row-parallel CRC-32C of achip_crc32c.h for the masked checksums, the batch-probing block encoder
// of snappy_compress_body.h.  Throughput comes from many streams per batch.
// What the one-shot form adds to the Java classes: the destination capacity (ACHIP_D_SNF_OUTPUT_TOO_SMALL / _MAX_OUTPUT); the
// offset reported with the stream-level IOExceptions, which carry none (the position of the chunk header); the capacity handed
// to the block decoder (the Java reader's buffer of max(65541, everything seen so far) bytes, limited by what the destination
// has left); a chunk that ends inside its length preamble is ACHIP_D_SNAPPY_TRUNCATED (Java would read stale buffer bytes).
#include "snappy_decode_body.h"
#include "snappy_compress_mw.h"
#include "achip_crc32c.h"

namespace achip {

namespace snf {
constexpr int COMPRESSED_DATA_FLAG = 0x00, UNCOMPRESSED_DATA_FLAG = 0x01, STREAM_IDENTIFIER_FLAG = 0xff;
constexpr int MAX_BLOCK_SIZE = 65536;
constexpr int IN_RING = 2048, OUT_RING = 4096;

#define SNF_FAIL(detail, off)                          \
    {                                                  \
        eo = (int64_t)(off);                           \
        return mk_status(ACHIP_CLASS_MALFORMED, detail); \
    }

// readUncompressedLength (M/snappy/SnappyRawDecompressor.java:277-321) of a chunk's data
__device__ __forceinline__ int32_t read_uncompressed_length(const uint8_t* in, int32_t len, int32_t& expectedOut, int32_t& eoOut)
{
    uint32_t expected = 0;
    int32_t nread = 0;
    for (int i = 0; i < 5; i++) {
        if (nread >= len) {
            eoOut = len - nread;
            return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_TRUNCATED);
        }
        const uint32_t b = in[nread++];
        expected |= (b & 0x7f) << (7 * i);
        if ((b & 0x80) == 0) {
            break;
        }
        if (i == 4) {
            eoOut = nread;
            return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LEN_HIGH_BIT);
        }
    }
    if ((int32_t)expected < 0) {
        eoOut = 0;
        return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_INVALID_LENGTH);
    }
    expectedOut = (int32_t)expected;
    return 0;
}

__device__ int32_t decompress_item(const Crc32cTables& tables, const uint8_t* __restrict__ in, int32_t inLen, uint8_t* out, int32_t outCap, uint8_t* lds, int lane,
                                   int32_t& opOut, int64_t& eo)
{
    eo = 0;
    opOut = 0;
    if (inLen < 10) SNF_FAIL(ACHIP_D_SNF_EOF_STREAM_HEADER, 0);  // :66-69
    if (ld8(in) != 0x50614E73000006FFull || in[8] != 0x70 || in[9] != 0x59) SNF_FAIL(ACHIP_D_SNF_BAD_STREAM_HEADER, 0);  // ff 06 00 00 "sNaPpY" :70-72
    int32_t pos = 10;
    int32_t o = 0;
    int32_t javaInput = MAX_BLOCK_SIZE + 5, javaUncompressed = MAX_BLOCK_SIZE + 5;  // allocateBuffersBasedOnSize(MAX_BLOCK_SIZE + 5) :61
    for (;;) {
        const int32_t chunk = pos;
        if (pos == inLen) {
            break;  // readBlockHeader: end of stream :295-297
        }
        if (inLen - pos < 4) SNF_FAIL(ACHIP_D_SNF_EOF_BLOCK_HEADER, chunk);  // :299-301
        const uint32_t header = ld4(in + pos);
        const int flag = (int)(header & 0xFF);
        const int32_t length = (int32_t)(header >> 8);
        pos += 4;
        bool skip = false;
        int32_t minLength;
        if (flag == COMPRESSED_DATA_FLAG || flag == UNCOMPRESSED_DATA_FLAG) {  // getFrameMetaData :234-277
            minLength = 5;
        }
        else if (flag == STREAM_IDENTIFIER_FLAG) {
            if (length != 6) SNF_FAIL(ACHIP_D_SNF_STREAM_ID_LENGTH, chunk);
            skip = true;
            minLength = 6;
        }
        else {
            if (flag <= 0x7f) SNF_FAIL(ACHIP_D_SNF_UNSKIPPABLE, chunk);
            skip = true;
            minLength = 0;
        }
        if (length < minLength) SNF_FAIL(ACHIP_D_SNF_INVALID_LENGTH, chunk);
        if (skip) {  // :151-154; skip stops quietly at the end of the stream
            pos += length < inLen - pos ? length : inLen - pos;
            continue;
        }
        if (length > javaInput) {  // :156-158
            javaInput = length;
            javaUncompressed = javaUncompressed < length ? length : javaUncompressed;
        }
        if (inLen - pos < length) SNF_FAIL(ACHIP_D_SNF_EOF_FRAME, chunk);  // :160-163
        const uint32_t stored = ld4(in + pos);                             // getFrameData :279-288
        const uint8_t* data = in + pos + 4;
        const int32_t dlen = length - 4;
        int32_t produced;
        if (flag == COMPRESSED_DATA_FLAG) {  // :167-177
            int32_t ulen = 0, beo = 0;
            const int32_t pst = read_uncompressed_length(data, dlen, ulen, beo);
            if (pst != 0) {
                eo = (int64_t)beo;
                return pst;
            }
            javaUncompressed = javaUncompressed < ulen ? ulen : javaUncompressed;
            if (ulen > outCap - o) {
                eo = (int64_t)chunk;
                return mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_OUTPUT_TOO_SMALL);
            }
            const int32_t limit = javaUncompressed < outCap - o ? javaUncompressed : outCap - o;
            int32_t bst = 0, bop = 0;
            wave_mem_order();
            snappy_buffer_decode<64, IN_RING, OUT_RING, 1>(lds, lds + IN_RING, nullptr, data, dlen, out + o, limit, lane, bst, beo, bop);
            wave_mem_order();
            if (bst != 0) {
                eo = (int64_t)beo;  // the block codec's exception propagates with its own offset
                return bst;
            }
            produced = bop;
        }
        else {  // raw :178-186
            if (dlen > outCap - o) {
                eo = (int64_t)chunk;
                return mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_OUTPUT_TOO_SMALL);
            }
            wave_mem_order();
            group_copy<64>(out + o, data, dlen, lane);
            wave_mem_order();
            produced = dlen;
        }
        if (stored != crc32c_mask(wave_crc32c(tables, out + o, produced, lane))) SNF_FAIL(ACHIP_D_SNF_CHECKSUM, chunk);  // :188-193
        o += produced;
        pos += length;
    }
    opOut = o;
    return 0;
}
#undef SNF_FAIL

}  // namespace snf

__global__ __launch_bounds__(64) void snappyframed_decompress_kernel(BatchArgs a, int32_t* nextItem, const int32_t* only)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[snf::IN_RING + snf::OUT_RING];
    __shared__ Crc32cTables tables;
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    crc32c_tables_init(tables, lane);
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
        if (only != nullptr && only[block] == 0) {
            continue;  // done by the chunk-parallel path
        }
        int32_t op = 0;
        int64_t eo = 0;
        const int32_t st = snf::decompress_item(tables, a.srcBase + a.srcOff[block], a.srcLen[block], a.dstBase + a.dstOff[block], a.dstCap[block], lds, lane, op, eo);
        if (lane == 0) {
            a.outLen[block] = st == 0 ? op : 0;
            a.status[block] = st;
            a.errOffset[block] = st == 0 ? 0 : eo;
        }
    }
}

namespace snf {
constexpr int64_t SLAB_BYTES = (32 + MAX_BLOCK_SIZE + MAX_BLOCK_SIZE / 6 + 63) / 64 * 64;  // maxCompressedLength(65536), rounded

// new SnappyFramedOutputStream(c, out); write(all); close()  -- M/snappy/SnappyFramedOutputStream.java:73-96, 113-145, 158-171
__device__ int32_t compress_item(const Crc32cTables& tables, uint16_t* table, const uint8_t* __restrict__ in, int32_t inLen, uint8_t* out, int32_t outCap, uint8_t* slab,
                                 int lane, int32_t& opOut)
{
    opOut = 0;
    if (inLen < 0) {
        return mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    }
    const int64_t blocks = ((int64_t)inLen + MAX_BLOCK_SIZE - 1) / MAX_BLOCK_SIZE;
    const int64_t bound = 10 + 8 * blocks + (int64_t)inLen;  // achip_snappyframed_max_compressed_length
    if (bound > 0x7FFFFFFF) {
        return mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    }
    if ((int64_t)outCap < bound) {
        return mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_MAX_OUTPUT);
    }
    if (lane == 0) {  // constructor :94
        st8(out, 0x50614E73000006FFull);
        out[8] = 0x70;
        out[9] = 0x59;
    }
    int32_t o = 10;
    for (int64_t pos = 0; pos < inLen; pos += MAX_BLOCK_SIZE) {  // full blocks straight from the input, the rest through the buffer :126-145, :186-192
        const int32_t length = (int32_t)(inLen - pos < MAX_BLOCK_SIZE ? inLen - pos : MAX_BLOCK_SIZE);
        const uint8_t* block = in + pos;
        const uint32_t crc = crc32c_mask(wave_crc32c(tables, block, length, lane));  // writeCompressed :204
        int32_t cst = 0, compressed = 0;
        snappy_compress_buffer_mw(table, block, length, slab, (int32_t)SLAB_BYTES, lane, cst, compressed);  // :206-211
        wave_mem_order();
        if (cst != 0) {
            return cst;
        }
        const bool keep = ((double)compressed / (double)length) <= 0.85;  // :214
        const uint8_t* data = keep ? slab : block;
        const int32_t dlen = keep ? compressed : length;
        if (lane == 0) {  // writeBlock :241-254
            const uint32_t headerLength = (uint32_t)dlen + 4u;
            st4(out + o, (keep ? (uint32_t)COMPRESSED_DATA_FLAG : (uint32_t)UNCOMPRESSED_DATA_FLAG) | (headerLength << 8));
            st4(out + o + 4, crc);
        }
        group_copy<64>(out + o + 8, data, dlen, lane);
        wave_mem_order();
        o += 8 + dlen;
    }
    opOut = o;
    return 0;
}
}  // namespace snf

__global__ __launch_bounds__(64) void snappyframed_compress_kernel(BatchArgs a, uint8_t* slabs, int32_t* nextItem, const int32_t* only)
{
    __shared__ uint16_t table[snc::MAX_HASH_TABLE_SIZE];
    __shared__ Crc32cTables tables;
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    crc32c_tables_init(tables, lane);
    uint8_t* slab = slabs + (size_t)blockIdx.x * snf::SLAB_BYTES;
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
        if (only != nullptr && only[block] == 0) {
            continue;  // done by the block-parallel path
        }
        int32_t op = 0;
        const int32_t st = snf::compress_item(tables, table, a.srcBase + a.srcOff[block], a.srcLen[block], a.dstBase + a.dstOff[block], a.dstCap[block], slab, lane, op);
        if (lane == 0) {
            a.outLen[block] = st == 0 ? op : 0;
            a.status[block] = st;
            a.errOffset[block] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Block-parallel writer (the default).  The blocks of a stream are independent; only a chunk's position depends on the
// sizes of the chunks before it.  But every block except a stream's last is full, and a chunk is never larger than its block
// stored raw, so block k can be written at its WORST-CASE position 10 + k * (8 + 65536) without knowing anything else:
//   plan     one lane per stream checks the arguments and appends its blocks to one list;
//   encode   persistent wavefronts (two tiers, as in snappy_compress.hip) take blocks from the list: masked CRC-32C, the block
//            encoder into a private slab, the 0.85 rule, chunk header + body to the worst-case position, the chunk's size noted;
//   compact  one wavefront per stream writes the stream header and moves the chunks left into place, in order (a move to the
//            left with all loads of a round before its stores is safe at any distance).
// Streams whose blocks do not fit into the list go to the one-wavefront-per-stream kernel above.
namespace snf {
constexpr int32_t MAX_BLOCKS = 1 << 20;
constexpr int32_t WORST_CHUNK = 8 + MAX_BLOCK_SIZE;

struct BlockList {
    int32_t* sFirst;   // per stream
    int32_t* sCount;
    int32_t* sStatus;
    int32_t* sSerial;
    int32_t* bStream;  // per block
    int32_t* bIndex;
    int32_t* bSize;    // chunk bytes written at the worst-case position (header included)
    int32_t* counters; // [0] blocks allocated, [1] blocks in the list, [2] encode cursor, [3] compact cursor, [32] serial cursor
};

__global__ __launch_bounds__(64) void snappyframed_plan_kernel(BatchArgs a, BlockList L)
{
    const int32_t stream = blockIdx.x * 64 + threadIdx.x;
    if (stream >= a.nBlocks) {
        return;
    }
    const int32_t inLen = a.srcLen[stream];
    int32_t st = 0;
    int64_t blocks = 0;
    if (inLen < 0) {
        st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    }
    else {
        blocks = ((int64_t)inLen + MAX_BLOCK_SIZE - 1) / MAX_BLOCK_SIZE;
        const int64_t bound = 10 + 8 * blocks + (int64_t)inLen;  // achip_snappyframed_max_compressed_length
        if (bound > 0x7FFFFFFF) {
            st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
        }
        else if ((int64_t)a.dstCap[stream] < bound) {
            st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_MAX_OUTPUT);
        }
    }
    const int32_t n = st == 0 ? (int32_t)blocks : 0;
    const int32_t first = n > 0 ? atomicAdd(L.counters, n) : 0;
    const bool fits = (int64_t)first + n <= MAX_BLOCKS;
    L.sFirst[stream] = first;
    L.sCount[stream] = fits ? n : 0;
    L.sStatus[stream] = st;
    L.sSerial[stream] = fits ? 0 : 1;
    for (int64_t k = 0; k < n && first + k < MAX_BLOCKS; k++) {
        L.bStream[first + k] = fits ? stream : -1;  // -1: a hole (its stream goes the other way)
        L.bIndex[first + k] = (int32_t)k;
    }
}

__global__ void snappyframed_seal_blocks_kernel(BlockList L)
{
    const int32_t allocated = L.counters[0];
    L.counters[1] = allocated < MAX_BLOCKS ? allocated : MAX_BLOCKS;
}

__global__ __launch_bounds__(256) void snappyframed_encode_kernel(BatchArgs a, BlockList L, uint8_t* outSlabs, uint16_t* tableSlabs)
{
    __shared__ uint16_t ldsTable[snc::MAX_HASH_TABLE_SIZE];
    __shared__ Crc32cTables tables;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    crc32c_tables_init(tables, lane);  // (all four wavefronts write the same values between the same barriers)
    uint8_t* const slab = outSlabs + ((size_t)blockIdx.x * 4 + wave) * SLAB_BYTES;
    uint16_t* const tableSlab = tableSlabs + ((size_t)blockIdx.x * 3 + (wave > 0 ? wave - 1 : 0)) * snc::MAX_HASH_TABLE_SIZE;
    const int32_t total = L.counters[1];
    for (;;) {
        int32_t b = 0;
        if (lane == 0) {
            b = atomicAdd(L.counters + 2, 1);
        }
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= total) {
            return;
        }
        const int32_t stream = L.bStream[b];
        if (stream < 0) {
            continue;
        }
        const int32_t k = L.bIndex[b];
        const int64_t pos = (int64_t)k * MAX_BLOCK_SIZE;
        const int32_t inLen = a.srcLen[stream];
        const int32_t length = (int32_t)(inLen - pos < MAX_BLOCK_SIZE ? inLen - pos : MAX_BLOCK_SIZE);
        const uint8_t* block = a.srcBase + a.srcOff[stream] + pos;
        uint8_t* out = a.dstBase + a.dstOff[stream] + 10 + (int64_t)k * WORST_CHUNK;
        const uint32_t crc = crc32c_mask(wave_crc32c(tables, block, length, lane));  // writeCompressed :204
        int32_t cst = 0, compressed = 0;
        if (wave == 0) {
            snappy_compress_buffer_mw(ldsTable, block, length, slab, (int32_t)SLAB_BYTES, lane, cst, compressed);  // :206-211
        }
        else {
            snappy_compress_buffer_mw(tableSlab, block, length, slab, (int32_t)SLAB_BYTES, lane, cst, compressed);
        }
        wave_mem_order();
        const bool keep = ((double)compressed / (double)length) <= 0.85;  // :214
        const uint8_t* data = keep ? slab : block;
        const int32_t dlen = keep ? compressed : length;
        if (lane == 0) {  // writeBlock :241-254
            const uint32_t headerLength = (uint32_t)dlen + 4u;
            st4(out, (keep ? (uint32_t)COMPRESSED_DATA_FLAG : (uint32_t)UNCOMPRESSED_DATA_FLAG) | (headerLength << 8));
            st4(out + 4, crc);
            L.bSize[b] = 8 + dlen;
        }
        group_copy<64>(out + 8, data, dlen, lane);
        wave_mem_order();
    }
}

__global__ __launch_bounds__(64) void snappyframed_compact_kernel(BatchArgs a, BlockList L)
{
    const int lane = threadIdx.x;
    for (;;) {
        int32_t stream = 0;
        if (lane == 0) {
            stream = atomicAdd(L.counters + 3, 1);
        }
        stream = __builtin_amdgcn_readfirstlane(stream);
        if (stream >= a.nBlocks) {
            return;
        }
        if (L.sSerial[stream] != 0) {
            continue;
        }
        const int32_t st = L.sStatus[stream];
        int32_t o = 0;
        if (st == 0) {
            uint8_t* out = a.dstBase + a.dstOff[stream];
            if (lane == 0) {  // constructor :94
                st8(out, 0x50614E73000006FFull);
                out[8] = 0x70;
                out[9] = 0x59;
            }
            o = 10;
            const int32_t first = L.sFirst[stream], n = L.sCount[stream];
            for (int32_t k = 0; k < n; k++) {
                const int32_t size = L.bSize[first + k];
                const int64_t from = 10 + (int64_t)k * WORST_CHUNK;
                if (from != o) {
                    wave_mem_order();
                    group_copy<64>(out + o, out + from, size, lane);  // to the left; every lane loads before it stores
                    wave_mem_order();
                }
                o += size;
            }
        }
        if (lane == 0) {
            a.outLen[stream] = st == 0 ? o : 0;
            a.status[stream] = st;
            a.errOffset[stream] = 0;
        }
    }
}
}  // namespace snf

namespace {
constexpr int SNF_COMPRESS_WAVES = 256 * 3;    // one wavefront per stream: 44 KB of LDS each
constexpr int SNF_ENCODE_WORKGROUPS = 256 * 3; // four wavefronts around one LDS table + the CRC tables
}
int64_t snappyframed_compress_scratch_bytes(int32_t nStreams)
{
    const int64_t n = nStreams < 1 ? 1 : nStreams;
    return 4096 + (int64_t)SNF_COMPRESS_WAVES * snf::SLAB_BYTES + (int64_t)SNF_ENCODE_WORKGROUPS * (4 * snf::SLAB_BYTES + 3 * snc::MAX_HASH_TABLE_SIZE * 2) +
           n * 16 + (int64_t)snf::MAX_BLOCKS * 12 + 4096;
}

hipError_t launch_snappyframed_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int variant)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* base = (uint8_t*)scratch;
    int32_t* counters = (int32_t*)base;
    hipError_t e = hipMemsetAsync(counters, 0, 4096, stream);
    if (e != hipSuccess) return e;
    uint8_t* serialSlabs = base + 4096;
    const unsigned serialGrid = (unsigned)(a.nBlocks < SNF_COMPRESS_WAVES ? a.nBlocks : SNF_COMPRESS_WAVES);
    if (variant == 0) {  // one wavefront per stream
        hipLaunchKernelGGL(snappyframed_compress_kernel, dim3(serialGrid), dim3(64), 0, stream, a, serialSlabs, counters + 32, (const int32_t*)nullptr);
        return hipGetLastError();
    }
    uint8_t* p = serialSlabs + (int64_t)SNF_COMPRESS_WAVES * snf::SLAB_BYTES;
    auto take = [&](int64_t bytes) {
        uint8_t* r = p;
        p += (bytes + 63) & ~(int64_t)63;
        return r;
    };
    uint8_t* outSlabs = take((int64_t)SNF_ENCODE_WORKGROUPS * 4 * snf::SLAB_BYTES);
    uint16_t* tableSlabs = (uint16_t*)take((int64_t)SNF_ENCODE_WORKGROUPS * 3 * snc::MAX_HASH_TABLE_SIZE * 2);
    snf::BlockList L;
    const int64_t n = a.nBlocks;
    L.counters = counters;
    L.sFirst = (int32_t*)take(4 * n);
    L.sCount = (int32_t*)take(4 * n);
    L.sStatus = (int32_t*)take(4 * n);
    L.sSerial = (int32_t*)take(4 * n);
    L.bStream = (int32_t*)take(4 * (int64_t)snf::MAX_BLOCKS);
    L.bIndex = (int32_t*)take(4 * (int64_t)snf::MAX_BLOCKS);
    L.bSize = (int32_t*)take(4 * (int64_t)snf::MAX_BLOCKS);
    const unsigned perStream = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL(snf::snappyframed_plan_kernel, dim3(perStream), dim3(64), 0, stream, a, L);
    hipLaunchKernelGGL(snf::snappyframed_seal_blocks_kernel, dim3(1), dim3(1), 0, stream, L);
    hipLaunchKernelGGL(snf::snappyframed_encode_kernel, dim3(SNF_ENCODE_WORKGROUPS), dim3(256), 0, stream, a, L, outSlabs, tableSlabs);
    hipLaunchKernelGGL(snf::snappyframed_compact_kernel, dim3((unsigned)(a.nBlocks < 2048 ? a.nBlocks : 2048)), dim3(64), 0, stream, a, L);
    hipLaunchKernelGGL(snappyframed_compress_kernel, dim3(serialGrid), dim3(64), 0, stream, a, serialSlabs, counters + 32, (const int32_t*)L.sSerial);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Chunk-parallel reader (the default).  A stream's chunks are independent of each other except for where they start and
// where their plaintext goes, and both follow from the chunk headers alone (a compressed chunk announces its plaintext
// length in its preamble).  So:
//   walk    one LANE per stream runs the Java loop over the chunk headers only (same checks, same order, everything that
//           does not need the chunk's body) and writes one descriptor per data chunk -- source, destination, the capacity
//           the Java reader would hand its block decoder -- into a batch assembled on the device;
//   decode  that batch goes through the batched Snappy block decoders (rings / lane-per-block, chosen on the device as for
//           any other batch): a 64 KiB chunk is exactly the unit those kernels are built for;
//   verify  one wavefront per chunk copies a stored chunk or takes the decoded one and checks the masked CRC-32C;
//   fold    one lane per stream looks at its chunks in stream order: the first chunk that failed (block codec error before
//           checksum error) gives the stream's status, else the error the walk stopped at, else the length.
// A stream whose chunks do not fit into the descriptor arrays is left to the one-wavefront-per-stream kernel below.
namespace snf {
constexpr int32_t MAX_CHUNKS = 1 << 20;

struct ChunkList {
    // per stream
    int32_t* sFirst;
    int32_t* sCount;
    int32_t* sStatus;   // what the walk stopped at (0: end of stream)
    int64_t* sErrOff;
    int32_t* sOut;      // plaintext bytes if every chunk decodes
    int32_t* sSerial;   // 1: handled by the serial kernel
    // per chunk: a batch for the block decoders ...
    int64_t* cSrcOff;
    int32_t* cSrcLen;
    int64_t* cDstOff;
    int32_t* cDstCap;
    int32_t* cOutLen;
    int32_t* cStatus;
    int64_t* cErrOff;
    // ... and what the verify / fold steps need
    uint32_t* cCrc;
    int32_t* cRawLen;   // >= 0: a stored chunk of this many bytes (the decoders see an empty input and are overruled); -1: compressed
    int32_t* cPos;      // position of the chunk header in its stream
    int32_t* counters;  // [0] chunks allocated, [1] chunks in the batch (= min(allocated, MAX_CHUNKS)), [2] verify cursor, [16] mixed groups
};

__device__ __forceinline__ uint32_t rd_bytes(const uint8_t* p, int n)  // little-endian, n <= 4 (cold: byte loads)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        v |= (uint32_t)p[i] << (8 * i);
    }
    return v;
}

// The Java loop over one stream without the chunk bodies.  FILL = false: count the data chunks; true: write their descriptors.
template <bool FILL>
__device__ void walk_stream(const BatchArgs& a, const ChunkList& L, int32_t stream, int32_t first, int32_t& countOut, int32_t& stOut, int64_t& eoOut, int32_t& outOut)
{
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[stream];
    const int32_t inLen = a.srcLen[stream];
    const int32_t outCap = a.dstCap[stream];
    countOut = 0;
    outOut = 0;
    stOut = 0;
    eoOut = 0;
#define WALK_FAIL(detail, off)                              \
    {                                                       \
        stOut = mk_status(ACHIP_CLASS_MALFORMED, detail);   \
        eoOut = (int64_t)(off);                             \
        outOut = o;                                         \
        countOut = n;                                       \
        return;                                             \
    }
    int32_t o = 0;
    int32_t n = 0;
    if (inLen < 10) WALK_FAIL(ACHIP_D_SNF_EOF_STREAM_HEADER, 0);
    if (rd_bytes(in, 4) != 0x000006FFu || rd_bytes(in + 4, 4) != 0x50614E73u || in[8] != 0x70 || in[9] != 0x59) WALK_FAIL(ACHIP_D_SNF_BAD_STREAM_HEADER, 0);
    int32_t pos = 10;
    int32_t javaInput = MAX_BLOCK_SIZE + 5, javaUncompressed = MAX_BLOCK_SIZE + 5;
    for (;;) {
        const int32_t chunk = pos;
        if (pos == inLen) {
            break;
        }
        if (inLen - pos < 4) WALK_FAIL(ACHIP_D_SNF_EOF_BLOCK_HEADER, chunk);
        const uint32_t header = rd_bytes(in + pos, 4);
        const int flag = (int)(header & 0xFF);
        const int32_t length = (int32_t)(header >> 8);
        pos += 4;
        bool skip = false;
        int32_t minLength;
        if (flag == COMPRESSED_DATA_FLAG || flag == UNCOMPRESSED_DATA_FLAG) {
            minLength = 5;
        }
        else if (flag == STREAM_IDENTIFIER_FLAG) {
            if (length != 6) WALK_FAIL(ACHIP_D_SNF_STREAM_ID_LENGTH, chunk);
            skip = true;
            minLength = 6;
        }
        else {
            if (flag <= 0x7f) WALK_FAIL(ACHIP_D_SNF_UNSKIPPABLE, chunk);
            skip = true;
            minLength = 0;
        }
        if (length < minLength) WALK_FAIL(ACHIP_D_SNF_INVALID_LENGTH, chunk);
        if (skip) {
            pos += length < inLen - pos ? length : inLen - pos;
            continue;
        }
        if (length > javaInput) {
            javaInput = length;
            javaUncompressed = javaUncompressed < length ? length : javaUncompressed;
        }
        if (inLen - pos < length) WALK_FAIL(ACHIP_D_SNF_EOF_FRAME, chunk);
        const uint32_t stored = rd_bytes(in + pos, 4);
        const int32_t dlen = length - 4;
        int32_t produced, limit, rawLen;
        if (flag == COMPRESSED_DATA_FLAG) {
            int32_t ulen = 0, beo = 0;
            const int32_t pst = read_uncompressed_length(in + pos + 4, dlen, ulen, beo);
            if (pst != 0) {  // the block codec's own exception, before any byte is decoded
                stOut = pst;
                eoOut = (int64_t)beo;
                outOut = o;
                countOut = n;
                return;
            }
            javaUncompressed = javaUncompressed < ulen ? ulen : javaUncompressed;
            if (ulen > outCap - o) {
                stOut = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_OUTPUT_TOO_SMALL);
                eoOut = (int64_t)chunk;
                outOut = o;
                countOut = n;
                return;
            }
            limit = javaUncompressed < outCap - o ? javaUncompressed : outCap - o;
            produced = ulen;
            rawLen = -1;
        }
        else {
            if (dlen > outCap - o) {
                stOut = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_SNF_OUTPUT_TOO_SMALL);
                eoOut = (int64_t)chunk;
                outOut = o;
                countOut = n;
                return;
            }
            limit = 0;
            produced = dlen;
            rawLen = dlen;
        }
        if (FILL) {
            const int32_t c = first + n;
            L.cSrcOff[c] = a.srcOff[stream] + pos + 4;
            L.cSrcLen[c] = rawLen >= 0 ? 0 : dlen;
            L.cDstOff[c] = a.dstOff[stream] + o;
            L.cDstCap[c] = limit;
            L.cCrc[c] = stored;
            L.cRawLen[c] = rawLen;
            L.cPos[c] = chunk;
        }
        n++;
        o += produced;
        pos += length;
    }
#undef WALK_FAIL
    countOut = n;
    outOut = o;
}

__global__ __launch_bounds__(64) void snappyframed_walk_kernel(BatchArgs a, ChunkList L)
{
    const int32_t stream = blockIdx.x * 64 + threadIdx.x;
    if (stream >= a.nBlocks) {
        return;
    }
    int32_t n = 0, st = 0, out = 0;
    int64_t eo = 0;
    walk_stream<false>(a, L, stream, 0, n, st, eo, out);
    const int32_t first = n > 0 ? atomicAdd(L.counters, n) : 0;
    const bool fits = (int64_t)first + n <= MAX_CHUNKS;
    L.sFirst[stream] = first;
    L.sCount[stream] = fits ? n : 0;
    L.sStatus[stream] = st;
    L.sErrOff[stream] = eo;
    L.sOut[stream] = out;
    L.sSerial[stream] = fits ? 0 : 1;
    if (fits && n > 0) {
        walk_stream<true>(a, L, stream, first, n, st, eo, out);
    }
    else if (!fits) {  // the part of this stream's range that lies inside the arrays: empty blocks nobody looks at
        for (int64_t c = first; c < (int64_t)first + n && c < MAX_CHUNKS; c++) {
            L.cSrcOff[c] = 0;
            L.cSrcLen[c] = 0;
            L.cDstOff[c] = 0;
            L.cDstCap[c] = 0;
            L.cCrc[c] = 0;
            L.cRawLen[c] = -1;
            L.cPos[c] = 0;
        }
    }
}

// the batch holds the chunks that fit: a stream that does not fit leaves a hole of descriptors nobody reads (src length 0 below)
__global__ void snappyframed_seal_kernel(ChunkList L)
{
    const int32_t allocated = L.counters[0];
    L.counters[1] = allocated < MAX_CHUNKS ? allocated : MAX_CHUNKS;
}

__global__ __launch_bounds__(64) void snappyframed_verify_kernel(BatchArgs a, ChunkList L)
{
    __shared__ Crc32cTables tables;
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    crc32c_tables_init(tables, lane);
    const int32_t total = L.counters[1];
    for (;;) {
        __syncthreads();
        if (lane == 0) {
            item = atomicAdd(L.counters + 2, 1);
        }
        __syncthreads();
        const int32_t c = item;
        if (c >= total) {
            return;
        }
        const int32_t rawLen = L.cRawLen[c];
        uint8_t* dst = a.dstBase + L.cDstOff[c];
        int32_t produced;
        if (rawLen >= 0) {
            group_copy<64>(dst, a.srcBase + L.cSrcOff[c], rawLen, lane);
            wave_sync();  // (the checksum below reads what the other lanes copied)
            produced = rawLen;
        }
        else {
            if (L.cStatus[c] != 0) {
                continue;  // the block decoder's verdict stands
            }
            produced = L.cOutLen[c];
        }
        const bool ok = L.cCrc[c] == crc32c_mask(wave_crc32c(tables, dst, produced, lane));
        if (lane == 0) {
            L.cStatus[c] = ok ? 0 : mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_SNF_CHECKSUM);
            L.cErrOff[c] = ok ? 0 : (int64_t)L.cPos[c];
        }
    }
}

__global__ __launch_bounds__(64) void snappyframed_fold_kernel(BatchArgs a, ChunkList L)
{
    const int32_t stream = blockIdx.x * 64 + threadIdx.x;
    if (stream >= a.nBlocks || L.sSerial[stream] != 0) {
        return;
    }
    int32_t st = L.sStatus[stream];
    int64_t eo = L.sErrOff[stream];
    const int32_t first = L.sFirst[stream], n = L.sCount[stream];
    for (int32_t k = 0; k < n; k++) {
        const int32_t cs = L.cStatus[first + k];
        if (cs != 0) {
            st = cs;
            eo = L.cErrOff[first + k];
            break;
        }
    }
    a.outLen[stream] = st == 0 ? L.sOut[stream] : 0;
    a.status[stream] = st;
    a.errOffset[stream] = st == 0 ? 0 : eo;
}
}  // namespace snf

hipError_t launch_snappy_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);
int snappy_ring_group_for(int32_t nBlocks);
hipError_t launch_snappy_element_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit);
hipError_t launch_lz4_mixed_groups(const BatchArgs& a, hipStream_t stream, int32_t* mixedGroups, int32_t minBlocks);

int64_t snappyframed_decompress_scratch_bytes(int32_t nStreams)
{
    const int64_t n = nStreams < 1 ? 1 : nStreams;
    return 4096 + n * (4 * 5 + 8) + (int64_t)snf::MAX_CHUNKS * (8 * 3 + 4 * 7) + 4096;
}

hipError_t launch_snappy_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats);
int64_t twopass_scratch_bytes(int32_t nBlocks, int64_t perBlock);

// variant 1 (default): the chunks through the ring decoders (with the probes' other choices behind them); variant 2 (round 2, written without
// a GPU at hand: not the default until measured): through the two-pass decoder -- the host reads the chunk count back (one
// synchronisation) and asks `aux` for the record arena; chunks whose records do not fit take the rings as in variant 1; 0: a wavefront per stream
hipError_t launch_snappyframed_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, int variant, const AuxScratch* aux)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* base = (uint8_t*)scratch;
    int32_t* counters = (int32_t*)base;
    hipError_t e = hipMemsetAsync(counters, 0, 4096, stream);
    if (e != hipSuccess) return e;
    const int32_t maxWaves = 256 * 8;
    if (variant == 0) {  // one wavefront per stream
        const unsigned grid = (unsigned)(a.nBlocks < maxWaves ? a.nBlocks : maxWaves);
        hipLaunchKernelGGL(snappyframed_decompress_kernel, dim3(grid), dim3(64), 0, stream, a, counters + 32, (const int32_t*)nullptr);
        return hipGetLastError();
    }
    // carve the lists out of the scratch
    snf::ChunkList L;
    uint8_t* p = base + 4096;
    const int64_t n = a.nBlocks;
    auto take = [&](int64_t bytes) {
        uint8_t* r = p;
        p += (bytes + 15) & ~(int64_t)15;
        return r;
    };
    L.counters = counters;
    L.sErrOff = (int64_t*)take(8 * n);
    L.sFirst = (int32_t*)take(4 * n);
    L.sCount = (int32_t*)take(4 * n);
    L.sStatus = (int32_t*)take(4 * n);
    L.sOut = (int32_t*)take(4 * n);
    L.sSerial = (int32_t*)take(4 * n);
    const int64_t C = snf::MAX_CHUNKS;
    L.cSrcOff = (int64_t*)take(8 * C);
    L.cDstOff = (int64_t*)take(8 * C);
    L.cErrOff = (int64_t*)take(8 * C);
    L.cSrcLen = (int32_t*)take(4 * C);
    L.cDstCap = (int32_t*)take(4 * C);
    L.cOutLen = (int32_t*)take(4 * C);
    L.cStatus = (int32_t*)take(4 * C);
    L.cCrc = (uint32_t*)take(4 * C);
    L.cRawLen = (int32_t*)take(4 * C);
    L.cPos = (int32_t*)take(4 * C);
    const unsigned perStream = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL(snf::snappyframed_walk_kernel, dim3(perStream), dim3(64), 0, stream, a, L);
    hipLaunchKernelGGL(snf::snappyframed_seal_kernel, dim3(1), dim3(1), 0, stream, L);
    // the chunks as a batch of Snappy blocks whose size is known on the device only: launches are sized for the arrays
    BatchArgs c = a;  // (every field the chunk batch does not set keeps the caller's value: no filter, no device count yet)
    c.srcBase = a.srcBase;
    c.srcOff = L.cSrcOff;
    c.srcLen = L.cSrcLen;
    c.dstBase = a.dstBase;
    c.dstOff = L.cDstOff;
    c.dstCap = L.cDstCap;
    c.outLen = L.cOutLen;
    c.status = L.cStatus;
    c.errOffset = L.cErrOff;
    c.nBlocks = snf::MAX_CHUNKS;
    c.ringPad = a.ringPad;
    c.nBlocksDev = counters + 1;
    c.only = nullptr;
    c.onlyStats = nullptr;
    int32_t* mixedGroups = counters + 16;
    bool viaTwoPass = false;
    int32_t nChunksHost = -1;  // the chunk count once the host has read it (variants 2 and 3)
    if ((variant == 2 || variant == 3) && aux != nullptr && aux->get != nullptr) {
        // variant 3 (the default since round 3): the element-length probe of the block API's auto mode runs on the chunk list before the
        // one synchronisation and its verdict comes back with the chunk count: short elements (text) -> the two-pass decoder, long
        // copies -> the rings (measured, 1024 streams x 4 MiB: rings 804 / 116 GiB/s fragments / corpus, two-pass 456 / 236)
        int32_t head[20] = {0};
        if (variant == 3) {
            e = hipMemsetAsync(mixedGroups, 0, 4 * sizeof(int32_t), stream);
            if (e == hipSuccess) e = launch_snappy_element_sample(c, stream, mixedGroups, 0, 0);
            if (e != hipSuccess) return e;
        }
        e = hipMemcpyAsync(head, counters, sizeof(head), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        const int32_t nChunks = head[1];
        nChunksHost = nChunks;
        const int32_t* v = head + 16;
        const bool wantTwoPass = variant == 2 || (v[1] > 0 && (int64_t)v[2] < 6LL * (int64_t)v[1]);
        viaTwoPass = nChunks == 0;
        if (nChunks > 0 && wantTwoPass) {
            const int64_t bytes = twopass_scratch_bytes(nChunks, 131072);  // (chunks hold at most 64 KiB: the block codec's arena per block)
            void* arena = aux->get(aux->user, bytes);
            if (arena != nullptr) {
                BatchArgs t = c;
                t.nBlocks = nChunks;
                t.nBlocksDev = nullptr;
                e = launch_snappy_decompress_twopass(t, stream, arena, bytes, 4, 0, 2, nullptr);
                if (e != hipSuccess) return e;
                viaTwoPass = true;
            }
        }
    }
    if (!viaTwoPass && nChunksHost > 0) {  // the host knows the count (variant 3 chose the rings): one launch of that size, no probes
        BatchArgs t = c;
        t.nBlocks = nChunksHost;
        t.nBlocksDev = nullptr;
        e = launch_snappy_decompress_rings(t, stream, snappy_ring_group_for(nChunksHost), 0, nullptr);
        if (e != hipSuccess) return e;
    }
    else if (!viaTwoPass) {
        e = launch_snappy_decompress_rings(c, stream, 4, 0, nullptr);  // (the chunk count is known on the device only: the launch is sized for the most there can be)
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(snf::snappyframed_verify_kernel, dim3(maxWaves), dim3(64), 0, stream, a, L);
    hipLaunchKernelGGL(snf::snappyframed_fold_kernel, dim3(perStream), dim3(64), 0, stream, a, L);
    // streams that did not fit into the lists
    {
        const unsigned grid = (unsigned)(a.nBlocks < maxWaves ? a.nBlocks : maxWaves);
        hipLaunchKernelGGL(snappyframed_decompress_kernel, dim3(grid), dim3(64), 0, stream, a, counters + 32, (const int32_t*)L.sSerial);
    }
    return hipGetLastError();
}

}  // namespace achip
